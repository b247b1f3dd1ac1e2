"""Known-answer tests of the surface-QG option of the vertex-grid oracle (params key sqg = 1; oracle/qgnode_oracle.c
comp_stretch_sqg, lap_bs, the surface term in comp_q / invert_q).  What is restated is the finished part of
qg-node/sqg_baroclinic_ms.h: comp_stretch with the surface buoyancy :77-98, idh0[0] = 1/dh[0] :502, S2 of the surface
= f/N2[0] :545, the tmp boundary rule :64-67 and laplacian(bs) in both dissipation operators :160-201; that file stops
at "TODO: STOPPED HERE" (:222) and does not compile, so bs is a prescribed field here.  PARITY UNPINNED like the rest of
the vertex oracle: these checks pin the restatement against closed forms and against the baroclinic model it extends."""
import numpy as np
import pytest

import orn

N2S = 300.0   # N^2 of the surface "layer" (first entry of N2 when sqg = 1)


def sqg_params(N, nl, extra="", **kw):
    dh, N2 = orn.NODE_LAYERS[nl]
    base = orn.node_params(N, nl, extra=extra, **kw)
    return base.replace(f"N2   = {N2}", f"N2   = [{N2S}," + N2[1:]) + "sqg = 1\n"


def bs_field(N):
    x = np.arange(N + 1) / N
    return 0.3 * np.outer(np.sin(np.pi * x), np.sin(2 * np.pi * x))[None] + 0.05


def make(N, nl, sqg, bs=None, mask=None, **kw):
    txt = sqg_params(N, nl, **kw) if sqg else orn.node_params(N, nl, **kw)
    o = orn.NodeOracle(txt, smoother=orn.GS_RB, quiet=1, TOLERANCE=1e-12)
    if mask is not None:
        o.set(orn.MASK, mask)
    if sqg and bs is not None:
        o.set(orn.BS, bs)
    psi = orn.node_psi(nl, N) * (mask if mask is not None else 1.0)
    o.set(orn.PSI, psi)
    o.set_const()
    return o


@pytest.mark.parametrize("nl", [2, 3, 4])
def test_without_surface_buoyancy_it_is_the_baroclinic_model(nl):
    """bs = 0: S2 layers 1..nl-1 of the sqg file are the interfaces of the baroclinic model and the stretching terms are
    the same numbers written with the opposite sign convention => same q, same steps up to the association of products"""
    N = 32
    a, b = make(N, nl, True, bs=np.zeros((1, N + 1, N + 1)), nu4=1.0), make(N, nl, False, nu4=1.0)
    assert a.param("sqg") == 1 and b.param("sqg") == 0
    assert a.param("idh0_0") == 1.0 / float(orn.NODE_LAYERS[nl][0].strip("[]").split(",")[0]) and b.param("idh0_0") == 0.0
    assert np.array_equal(a.get(orn.S2), b.get(orn.S2))
    assert np.allclose(a.get(orn.Q), b.get(orn.Q), rtol=0, atol=1e-13 * np.abs(b.get(orn.Q)).max())
    for _ in range(3):
        a.step(True); b.step(True)
    # the tmp boundary rule differs on purpose (:64-67 subtract psi_bc instead of zeta at the wall), which feeds the
    # biharmonic term next to the walls only: compare away from them
    qa, qb = a.get(orn.Q), b.get(orn.Q)
    assert np.abs(qa - qb)[:, 3:-3, 3:-3].max() <= 1e-9 * np.abs(qb).max()


def test_surface_term_of_comp_q_and_its_inverse():
    """q_0 gains exactly S2S * bs / dh[0] with S2S = f0 / N2[0] (f, not f^2, :545); invert_q takes it out again"""
    N, nl = 32, 3
    bs = bs_field(N)
    a, b = make(N, nl, True, bs=bs), make(N, nl, True, bs=np.zeros_like(bs))
    s2s = a.get(orn.S2S)
    assert np.allclose(s2s, 46.5 / N2S, rtol=1e-15)
    dq0 = a.get(orn.Q) - b.get(orn.Q)
    inner = (slice(None), slice(1, -1), slice(1, -1))
    want = s2s * bs / 0.1
    assert np.allclose(dq0[0][inner[1:]], want[0][inner[1:]], rtol=1e-10, atol=1e-12)
    assert np.abs(dq0[1:]).max() == 0.0
    psi0 = a.get(orn.PSI)
    q = a.get(orn.Q)
    a.set(orn.PSI, np.zeros_like(psi0))
    st = a.invert_q()
    assert st.resa < 1e-12
    assert np.abs(a.get(orn.PSI) - psi0).max() <= st.resa * 0.0737 * 100.0**2   # residual x |A^-1|_inf of the Poisson problem
    assert np.array_equal(a.get(orn.Q), q)     # invert_q leaves q (and its boundary rule) alone


def test_dissipation_carries_the_laplacian_of_bs():
    """rhs(bs) - rhs(0) in the top layer = (nu - nu4) * S2S * laplacian(bs) / dh[0] on the inner vertices: both
    dissipation operators add comp_stretch(., laplacian(bs), ...) (:170-201; del4_bs is laplacian(bs) again)"""
    N, nl, nu, nu4 = 32, 3, 5.0, 1.5
    bs = bs_field(N)
    a, b = make(N, nl, True, bs=bs, nu=nu, nu4=nu4), make(N, nl, True, bs=np.zeros_like(bs), nu=nu, nu4=nu4)
    # same psi on both sides: the tendency is evaluated at fixed psi (rhs_pv does not invert)
    a.rhs_pv(); b.rhs_pv()
    d = a.get(orn.DQ) - b.get(orn.DQ)
    D = 100.0 / N
    lap = (bs[0, 1:-1, 2:] + bs[0, 1:-1, :-2] + bs[0, 2:, 1:-1] + bs[0, :-2, 1:-1] - 4 * bs[0, 1:-1, 1:-1]) / D**2
    want = (nu - nu4) * (46.5 / N2S) * lap / 0.1
    assert np.allclose(d[0, 1:-1, 1:-1], want, rtol=1e-9, atol=1e-9 * np.abs(want).max())
    assert np.abs(d[1:]).max() <= 1e-12 * np.abs(a.get(orn.DQ)).max()


def test_masked_steps_stay_finite_and_bs_matters():
    N, nl = 32, 3
    mk = np.ones((1, N + 1, N + 1)); mk[0, 8:13, 16:21] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    a, b = make(N, nl, True, bs=bs_field(N), mask=mk), make(N, nl, True, bs=np.zeros((1, N + 1, N + 1)), mask=mk)
    for _ in range(4):
        a.step(True); b.step(True)
    assert np.isfinite(a.get(orn.Q)).all() and np.abs(a.get(orn.PSI) - b.get(orn.PSI)).max() > 0
    assert np.all(a.get(orn.PSI)[:, mk[0] == 0] == 0)


def test_sqg_needs_two_layers():
    with pytest.raises(ValueError):
        orn.NodeOracle(orn.node_params(16, 1) + "sqg = 1\n")
