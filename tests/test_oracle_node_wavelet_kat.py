"""Known-answer tests of the vertex model's wavelet filter in the oracle (oracle/qgnode_oracle.c: wv_setup, wv_masked_apply,
orn_wavelet_filter) -- wavelet_filter qg-node/qg_baroclinic_ms.h:346-400, sig_lev / mask_c :525-578, wavelet_mask /
inverse_wavelet_mask qg-node/wavelet_vertex.h:10-46 -- against an independent numpy restatement of that text and against
the limits it must have (all-pass with an open mask = identity, all-stop = nothing removed)."""
import numpy as np
import pytest

import orn


def params(N, nl, Lfmax, Lfmin=None, fac=0.0, extra=""):
    return orn.node_params(N, nl, extra=f"Lfmax = {Lfmax}\nLfmin = {Lfmax if Lfmin is None else Lfmin}\nfac_filt_Rd = {fac}\n" + extra)


def island(N):
    mk = np.ones((1, N + 1, N + 1)); mk[0, N // 4: N // 4 + N // 8 + 1, N // 2: N // 2 + N // 8] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    return mk


def np_ghost(f):  # dirichlet(0) ghost ring, x sides first
    n = f.shape[0]
    g = np.zeros((n + 2, n + 2)); g[1:-1, 1:-1] = f
    g[1:-1, 0] = -g[1:-1, 1]; g[1:-1, -1] = -g[1:-1, -2]
    g[0, :] = -g[1, :]; g[-1, :] = -g[-2, :]
    return g


def np_bilinear(c):  # coarse [n][n] with ghosts -> fine [2n][2n]
    n = c.shape[0]
    g = np_ghost(c)
    f = np.empty((2 * n, 2 * n))
    for b in (0, 1):
        for a in (0, 1):
            cx, cy = (1 if a else -1), (1 if b else -1)
            C = g[1:-1, 1:-1]; X = g[1:-1, 1 + cx: n + 1 + cx]; Y = g[1 + cy: n + 1 + cy, 1:-1]; XY = g[1 + cy: n + 1 + cy, 1 + cx: n + 1 + cx]
            f[b::2, a::2] = (9 * C + 3 * (X + Y) + XY) / 16
    return f


def np_masked_filter(s0, sig, mc):
    """wavelet_vertex.h:10-46 with the scaling by sig_lev in between; lists are level 0 (finest) ... K-1 (1 x 1)"""
    K = len(sig)
    s = [s0]
    for k in range(1, K):
        f = s[-1]
        s.append(0.25 * (f[0::2, 0::2] + f[1::2, 0::2] + f[0::2, 1::2] + f[1::2, 1::2]))
    w = [(s[k] - np_bilinear(s[k + 1])) * mc[k] * sig[k] for k in range(K - 1)] + [s[K - 1] * mc[K - 1] * sig[K - 1]]
    r = w[K - 1] * mc[K - 1]
    for k in range(K - 2, -1, -1):
        r = (np_bilinear(r) + w[k]) * mc[k]
    return r


@pytest.mark.parametrize("N,nl,Lfmax,Lfmin", [(32, 2, 12.0, 3.0), (64, 3, 5.0, 5.0), (16, 1, 30.0, 2.0)])
def test_masked_transform_against_numpy(N, nl, Lfmax, Lfmin):
    o = orn.NodeOracle(params(N, nl, Lfmax, Lfmin), quiet=1)
    o.set(orn.MASK, island(N)); o.set(orn.PSI, orn.node_psi(nl, N) * island(N)); o.set_const()
    K = o.cell_levels()
    sig, mc = [o.wv_get(0, k) for k in range(K)], [o.wv_get(1, k) for k in range(K)]
    # coefficients: 0 where L_filt(y) > 2 Delta, 1 where <= Delta or any child is 1 (:527-552); mask_c = cell mean of the vertex mask
    for k in range(K):
        n = N >> k; D = 100.0 / n
        y = np.arange(n) * D
        L2 = Lfmax + (y / 100.0) * (Lfmin - Lfmax)
        base = np.where(L2 > 2 * D, 0.0, np.where(L2 > D, 1 - (L2 - D) / D, 1.0))[:, None] * np.ones((1, n))
        if k > 0:
            ch = sig[k - 1]
            anyc = (ch[0::2, 0::2] + ch[1::2, 0::2] + ch[0::2, 1::2] + ch[1::2, 1::2]) > 0
            base = np.where(anyc, 1.0, base)
        assert np.allclose(sig[k], base, atol=1e-14), k
    m = island(N)[0]
    assert np.allclose(mc[0], 0.25 * (m[:-1, :-1] + m[:-1, 1:] + m[1:, :-1] + m[1:, 1:]), atol=0)
    rng = np.random.default_rng(N)
    cells = rng.standard_normal((nl, N, N))
    got = o.wv_apply(cells)
    for l in range(nl):
        want = np_masked_filter(cells[l], sig, mc)
        assert np.abs(got[l] - want).max() <= 1e-13 * np.abs(cells).max(), l


def test_limits_all_stop_and_linearity():
    """(the all-pass limit is NOT the identity here: the boundary vertices always carry mask = 0, so mask_c < 1 next to the
    walls on every level and the coarse levels spread that inwards -- the numpy comparison above covers sig_lev = 1)"""
    N, nl = 32, 2
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((nl, N, N)), rng.standard_normal((nl, N, N))
    o = orn.NodeOracle(params(N, nl, 1e29), quiet=1)        # longer than the domain: sig_lev = 0 everywhere
    o.set_const()
    assert np.abs(o.wv_apply(a)).max() == 0.0
    o = orn.NodeOracle(params(N, nl, 9.0, 2.0), quiet=1)
    o.set(orn.MASK, island(N)); o.set_const()
    assert np.abs(o.wv_apply(a + 2 * b) - (o.wv_apply(a) + 2 * o.wv_apply(b))).max() <= 1e-13 * 4


def test_filter_event_removes_the_large_scales_and_keeps_the_running_mean():
    """psi_loc = vertex average of the filtered cell field; psi -= psi_loc (times mask), psi_f = running mean of psi_loc/dtflt
    over the calls (nbar), q recomputed from the new psi (:381-392)"""
    N, nl, dtflt = 32, 3, 0.5
    mk = island(N)
    o = orn.NodeOracle(params(N, nl, 40.0, 40.0), quiet=1, TOLERANCE=1e-11)
    o.set(orn.MASK, mk); o.set(orn.PSI, orn.node_psi(nl, N) * mk); o.set_const()
    psi0 = o.get(orn.PSI)
    o.wavelet_filter(dtflt)
    psi1, pf1 = o.get(orn.PSI), o.get(orn.PSIF)
    removed = psi0 - psi1
    inner = mk[0] == 1
    # the elliptic solve inside the filter reproduces psi0 (q was computed from it) to the solver tolerance
    assert np.abs(pf1 * dtflt - removed)[:, inner].max() <= 1e-8 * np.abs(psi0).max()
    assert np.abs(removed).max() > 0.1 * np.abs(psi0).max()          # 40 > 2 Delta on the fine levels: large scales go
    assert np.all(psi1[:, ~inner] == 0)
    q1 = o.get(orn.Q)
    o.wavelet_filter(dtflt)
    pf2 = o.get(orn.PSIF)
    removed2 = psi1 - o.get(orn.PSI)
    assert np.abs(pf2 - (pf1 + removed2 / dtflt) / 2)[:, inner].max() <= 1e-8 * np.abs(pf1).max()
    assert np.isfinite(q1).all()


def test_default_lfmax_is_huge_and_leaves_psi_alone():
    N, nl = 16, 2
    o = orn.NodeOracle(orn.node_params(N, nl), quiet=1, TOLERANCE=1e-12)
    o.set(orn.PSI, orn.node_psi(nl, N)); o.set_const()
    psi0 = o.get(orn.PSI)
    o.wavelet_filter(1.0)
    assert np.abs(o.get(orn.PSI) - psi0).max() <= 1e-9 * np.abs(psi0).max() and np.abs(o.get(orn.PSIF)).max() == 0
