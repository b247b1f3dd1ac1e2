"""Known-answer tests of the oracle's wavelet scale filter (msqg/qg.h:509-560, coefficients
qg.h:1059-1090; Basilisk's wavelet()/inverse_wavelet() restated from the published source,
[BASILISK RULE], parity unpinned)."""
import numpy as np

import orc


def make(N, nl, extra, psi=None):
    o = orc.Oracle(orc.double_gyre_params(N, nl, extra=extra), smoother=orc.GS_RB, quiet=1, TOLERANCE=1e-11)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N) if psi is None else psi)
    o.set_const()
    return o


def test_siglev_limits_and_transition():
    N = 32                                   # L0 = 80: Delta_k = 2.5 * 2^k
    o = make(N, 2, "afilt = 1\nLfmax = 0.1\n")            # sig_filt = 0.1 <= Delta everywhere
    K = o.wavelet_levels()
    assert K == 6 and o.siglev(K - 1).shape == (1, 1, 1)
    for k in range(K):                       # low pass 1 everywhere -> high pass 0: everything is "large scale"
        assert np.all(o.siglev(k) == 0)
    o = make(N, 2, "afilt = 1000\n")         # sig_filt = 1000 > 2 * L0: low pass 0 -> high pass 1
    for k in range(K):
        assert np.all(o.siglev(k) == 1)
    # sig_filt = 7: Delta_1 = 5 < 7 <= 10 = 2 Delta_1 -> low pass 1 - (7-5)/5 = 0.6 on level 1, 0 on level 0
    # (7 > 2 * 2.5), 1 on coarser levels (children > 0) -> high pass 1, 0.4, 0, 0 ...
    o = make(N, 2, "afilt = 7\n")
    assert np.all(o.siglev(0) == 1) and np.allclose(o.siglev(1), 0.4, rtol=1e-15)
    for k in range(2, K):
        assert np.all(o.siglev(k) == 0)
    # space-dependent Rd: the switch follows the local value
    Rd = np.ones((1, N, N)); Rd[0, :, N // 2:] = 100.0
    o = orc.Oracle(orc.double_gyre_params(N, 2, extra="afilt = 1\n"), quiet=1)
    o.set(orc.RD, Rd)
    o.set_const()
    s0 = o.siglev(0)
    assert np.all(s0[0, :, : N // 2] == 0) and np.all(s0[0, :, N // 2:] == 1)


def test_perfect_reconstruction_and_annihilation():
    N, nl = 32, 3
    o = make(N, nl, "afilt = 1000\n")        # all coefficients kept
    psi = o.get(orc.PSI)
    o.wavelet_apply(orc.PSI)
    assert np.abs(o.get(orc.PSI) - psi).max() <= 4e-16 * np.abs(psi).max()
    o = make(N, nl, "afilt = 1\nLfmax = 0.1\n")   # all coefficients removed
    o.wavelet_apply(orc.PSI)
    assert np.all(o.get(orc.PSI) == 0)


def np_wavelet_filter(f, sig):
    """independent numpy restatement: restriction = 2x2 mean, Dirichlet ghosts (-edge, corners +),
    bilinear prolongation 9/3/3/1, details scaled by sig[k] (level 0 = finest)"""
    def ghost(a):
        g = np.pad(a, ((0, 0), (1, 1), (1, 1)))
        g[:, 1:-1, 0], g[:, 1:-1, -1] = -a[:, :, 0], -a[:, :, -1]
        g[:, 0, :], g[:, -1, :] = -g[:, 1, :], -g[:, -2, :]
        return g

    def prolong(c):
        g = ghost(c)
        n = c.shape[1]
        out = np.empty((c.shape[0], 2 * n, 2 * n))
        for dj, sj in ((0, -1), (1, 1)):
            for di, si in ((0, -1), (1, 1)):
                C0 = g[:, 1:-1, 1:-1]
                Cx = g[:, 1:-1, 1 + si: 1 + si + n]
                Cy = g[:, 1 + sj: 1 + sj + n, 1:-1]
                Cxy = g[:, 1 + sj: 1 + sj + n, 1 + si: 1 + si + n]
                out[:, dj::2, di::2] = (9 * C0 + 3 * (Cx + Cy) + Cxy) / 16
        return out

    s = [f]
    while s[-1].shape[1] > 1:
        a = s[-1]
        n = a.shape[1] // 2
        s.append(a.reshape(a.shape[0], n, 2, n, 2).mean(axis=(2, 4)))
    K = len(s)
    w = [(s[k] - prolong(s[k + 1])) * sig[k] for k in range(K - 1)] + [s[K - 1] * sig[K - 1]]
    r = w[K - 1]
    for k in range(K - 2, -1, -1):
        r = prolong(r) + w[k]
    return r


def test_against_numpy_restatement():
    N, nl = 32, 3
    rng = np.random.default_rng(3)
    Rd = 0.5 + 2.5 * rng.random((1, N, N))          # sig_filt = 3 Rd in [1.5, 9]: mixed coefficients on levels 0..2
    o = orc.Oracle(orc.double_gyre_params(N, nl, extra="afilt = 3\n"), quiet=1)
    psi = rng.standard_normal((nl, N, N))
    o.set(orc.RD, Rd)
    o.set(orc.PSI, psi)
    o.set_const()
    sig = [o.siglev(k) for k in range(o.wavelet_levels())]
    assert 0 < sig[0].mean() < 1 and any(((x > 0) & (x < 1)).any() for x in sig)
    o.wavelet_apply(orc.PSI)
    ref = np_wavelet_filter(psi, sig)
    assert np.abs(o.get(orc.PSI) - ref).max() <= 1e-14 * np.abs(psi).max()


def test_wavelet_filter_bookkeeping():
    N, nl = 32, 2
    o = make(N, nl, "afilt = 7\n")
    q0 = o.get(orc.Q)
    o.wavelet_filter(0.5)
    q1, qof = o.get(orc.Q), o.get(orc.QOF)
    assert np.abs(qof - (q0 - q1) / 0.5).max() <= 1e-13 * np.abs(q0).max() / 0.5
    assert np.array_equal(o.get(orc.TMP), q0)
    # the filtered q is comp_q of the filtered psi
    assert np.array_equal(o.pyp2q(o.get(orc.PSI)), q1)
    # dtflt < 0 (energy diagnostics): q restored, qof = (q0 - q1) / dtflt
    o = make(N, nl, "afilt = 7\n")
    o.wavelet_filter(-0.5)
    assert np.array_equal(o.get(orc.Q), q0)
    assert np.abs(o.get(orc.QOF) + (q0 - q1) / 0.5).max() <= 1e-12 * np.abs(q0).max()
    # second call: nbar is passed by value in the reference, so no running mean
    o.wavelet_filter(-0.5)
    assert np.abs(o.get(orc.QOF) + (q0 - q1) / 0.5).max() <= 1e-6 * np.abs(q0).max()   # re-solve from the filtered psi: solver tolerance
