"""GPU parity of the vertex model's wavelet filter (msomn_wavelet_filter, k_wv_recon_m / k_wv_root_m / k_wv_vert2cell /
k_wv_vertex_update) against oracle/qgnode_oracle.c (pinned by tests/test_oracle_node_wavelet_kat.py).
strict build: bit-exact; product build: relative tolerance at the assertion."""
import numpy as np
import pytest

import orn
from msom_amd import NodeQG
from test_oracle_node_wavelet_kat import island, params

pytestmark = pytest.mark.gpu


def pair(N, nl, strict, Lfmax, Lfmin=None, fac=0.0):
    txt = params(N, nl, Lfmax, Lfmin, fac)
    o = orn.NodeOracle(txt, smoother=orn.GS_RB, quiet=1, TOLERANCE=1e-10)
    g = NodeQG(txt, strict=strict)
    g.set_option("quiet", 1); g.set_option("TOLERANCE", 1e-10)
    mk = island(N)
    psi = orn.node_psi(nl, N) * mk
    for m_, set_ in ((o, lambda f, a: o.set(getattr(orn, f), a)), (g, lambda f, a: g.set(f, a))):
        set_("MASK", mk); set_("PSI", psi)
        m_.set_const()
    return o, g


def same(a, b, strict, rtol):
    if strict:
        assert np.array_equal(a, b), f"max diff {np.abs(a - b).max():g}"
    else:
        assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("N,nl,Lfmax,Lfmin,fac", [(32, 2, 12.0, 3.0, 0.0), (64, 3, 25.0, 25.0, 0.0), (32, 3, 30.0, 30.0, 40.0), (16, 1, 30.0, 2.0, 0.0)])
def test_coefficients_transform_and_filter_event(N, nl, Lfmax, Lfmin, fac, strict):
    o, g = pair(N, nl, strict, Lfmax, Lfmin, fac)
    for k in range(o.cell_levels()):
        assert np.array_equal(g.wv_get(0, k), o.wv_get(0, k)), k      # sig_lev (fac_filt_Rd > 0: from S2 of layer 0)
        same(g.wv_get(1, k), o.wv_get(1, k), True, 0)                  # mask_c
    cells = np.random.default_rng(N + nl).standard_normal((nl, N, N))
    same(g.wv_apply(cells), o.wv_apply(cells), strict, 1e-13)
    for _ in range(2):
        o.wavelet_filter(0.5); g.wavelet_filter(0.5)
        for gf, of in (("PSI", orn.PSI), ("PSIF", orn.PSIF), ("Q", orn.Q)):
            same(g.get(gf), o.get(of), strict, 1e-7)
    for _ in range(2):                                                  # the model keeps running on the filtered state
        o.step(True); g.step(True)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-6)


def test_filter_event_in_the_driver_loop(tmp_path):
    """msomn_run with dtflt > 0: event filter (t = dtflt; t += dtflt) shortens the steps like the output events do"""
    N, nl = 32, 2
    txt = params(N, nl, 30.0, 30.0, extra="noise_init = 1e-3\ntend = 0.2\ndtout = 0.1\ndtflt = 0.05\n")
    g = NodeQG(txt)
    g.set_option("quiet", 1)
    g.run(str(tmp_path))
    assert abs(g.t - 0.2) < 1e-12 and np.abs(g.get("PSIF")).max() > 0


def test_full_size_filter():
    N, nl = 2048, 3
    g = NodeQG(params(N, nl, 5.0, 5.0))
    g.set_option("quiet", 1); g.set_option("TOLERANCE", 1e-7)
    mk = island(N)
    g.set("MASK", mk); g.set("PSI", orn.node_psi(nl, N) * mk); g.set_const()
    psi0 = g.get("PSI")
    g.wavelet_filter(1.0)
    psi1 = g.get("PSI")
    assert np.isfinite(psi1).all() and np.abs(psi1).max() < np.abs(psi0).max() and np.all(psi1[:, mk[0] == 0] == 0)
