"""GPU parity of the surface-QG option of the vertex model (params key sqg = 1: msomn_* with MSOMN_BS / MSOMN_S2S,
kernels k_n_stretch_sqg, k_n_lap_bs, k_n_sqg_rhs) against oracle/qgnode_oracle.c (pinned by tests/test_oracle_sqg_kat.py).
strict build: bit-exact; product build: relative tolerance at the assertion."""
import numpy as np
import pytest

import orn
from msom_amd import MsomError, NodeQG
from test_oracle_sqg_kat import bs_field, sqg_params

pytestmark = pytest.mark.gpu


def pair(N, nl, strict, mask=True, **kw):
    txt = sqg_params(N, nl, extra="tau1 = 5e-4\ntf1 = 0.3\ntf2 = 0.7\n", **kw)
    o = orn.NodeOracle(txt, smoother=orn.GS_RB, quiet=1, TOLERANCE=1e-9)
    g = NodeQG(txt, strict=strict)
    g.set_option("quiet", 1); g.set_option("TOLERANCE", 1e-9)
    mk = np.ones((1, N + 1, N + 1))
    if mask:
        mk[0, N // 4: N // 4 + N // 8 + 1, N // 2: N // 2 + N // 8] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    psi = orn.node_psi(nl, N) * mk
    bs = bs_field(N)
    for m_, set_ in ((o, lambda f, a: o.set(getattr(orn, f), a)), (g, lambda f, a: g.set(f, a))):
        set_("MASK", mk); set_("BS", bs); set_("PSI", psi)
        m_.set_const()
    return o, g


def same(a, b, strict, rtol):
    if strict:
        assert np.array_equal(a, b), f"max diff {np.abs(a - b).max():g}"
    else:
        assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,bc_fac", [(2, 0.0), (3, 1.0), (4, 0.5)])
def test_sqg_operators_and_steps(nl, bc_fac, strict):
    o, g = pair(32, nl, strict, bc_fac=bc_fac, nu4=1.5)
    assert g.param("sqg") == 1 and g.param("idh0_0") == o.param("idh0_0") != 0
    same(g.get("S2S"), o.get(orn.S2S), True, 0)
    same(g.get("S2"), o.get(orn.S2), True, 0)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-12)            # comp_q with the surface term
    so, sg = o.invert_q(), g.invert_q()
    if strict:
        assert (sg.i, sg.resb, sg.resa) == (so.i, so.resb, so.resa)
    same(g.get("PSI"), o.get(orn.PSI), strict, 1e-7)
    o.rhs_pv(); g.rhs_pv()                                    # laplacian(bs) in both dissipation operators, tmp rule
    same(g.get("DQ"), o.get(orn.DQ), strict, 1e-6)
    o.set_tnext(0.11); g.set_tnext(0.11)
    for _ in range(5):
        o.step(True); g.step(True)
        if strict:
            assert (g.t, g.dt) == (o.t, o.dt)
    same(g.get("PSI"), o.get(orn.PSI), strict, 1e-6)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-6)


def test_sqg_full_size_round_trip():
    """BASELINE config 5 size (2048^2, 3 layers) with an island: q -> psi -> q closes with the surface term in place"""
    N, nl = 2048, 3
    g = NodeQG(sqg_params(N, nl, bc_fac=1.0))
    g.set_option("quiet", 1); g.set_option("TOLERANCE", 1e-7)
    mk = np.ones((1, N + 1, N + 1)); mk[0, 500:700, 900:1200] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    g.set("MASK", mk); g.set("BS", bs_field(N)); g.set("PSI", orn.node_psi(nl, N) * mk)
    g.set_const()
    q0 = g.get("Q")
    g.set("PSI", np.zeros((nl, N + 1, N + 1)))
    st = g.invert_q()
    assert st.resa < 1e-7 and st.i < 40
    g.comp_q()
    inner = mk[0] == 1
    assert np.abs((g.get("Q") - q0)[:, inner]).max() <= 2e-7
    for _ in range(2):
        g.step(True)
    assert np.isfinite(g.ke())


def test_sqg_needs_two_layers():
    with pytest.raises(MsomError, match="nl >= 2"):
        NodeQG(orn.node_params(16, 1) + "sqg = 1\n")
