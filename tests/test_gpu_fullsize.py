"""Differential tests of the PRODUCT PATH at BASELINE.json's sizes (C3 2048^2 x 3, C4 4096^2 x 6), where the CPU oracle is too
slow: the default kernels -- chained smoother with its lean interior body, prolongation and correction riders, one-launch
coarse levels, fused residual passes, one-layer-per-wavefront tendency kernel with the advance folded in -- against the SAME
library driven through the kernel-per-reference-loop chain (march = 0, block8 = 0, fused = 0, mg_fused = 0, mg_coarse = 0, prolong_fused = 0:
the path the small-grid tests hold to the oracle bit for bit).  Strict build: bit-identical; product build: <= 1e-10 relative.
The strip / chunk logic of the marching kernels only meets many interior chunks, both marching directions and several
workgroup rounds at these sizes (a hazard of the lean smoother body showed at 2048^2 x 6 and nowhere below)."""
import numpy as np
import pytest

import orc
from msom_amd import QG, FIELDS as F
from test_gpu_parity import rel

pytestmark = pytest.mark.gpu

REFERENCE_CHAIN = dict(march=0, block8=0, fused=0, mg_fused=0, mg_coarse=0, prolong_fused=0, adv_fused=0)


def run(N, nl, strict, opts, steps, tol):
    txt = orc.double_gyre_params(N, nl)
    g = QG(txt, strict=strict)
    g.option("quiet", 1)
    g.option("TOLERANCE", tol)
    g.set(F["PSI"], orc.synthetic_psi(nl, N, N))
    g.set_const()
    if strict:
        g.option("uniform_S", 1)   # the chained smoother exists for the uniform-S column solver (opt-in in the strict build)
    for k, v in opts.items():
        g.option(k, v)
    g.set_tnext(float("inf"))
    dq, dtmax = g.update()         # one RHS evaluation (update_qg) ...
    dts = [g.step() for _ in range(steps)]   # ... and whole RK2 steps
    st = g.mgstats()
    out = dict(dq=dq, dtmax=dtmax, dts=dts, q=g.get(F["Q"]), psi=g.get(F["PSI"]), st=(st.i, st.resa, st.resb))
    g.close()
    return out


@pytest.mark.parametrize("N,nl", [(2048, 3), (4096, 6)])
@pytest.mark.parametrize("strict", [True, False])
def test_default_kernels_equal_the_reference_loop_chain_at_baseline_sizes(N, nl, strict):
    # TOLERANCE 1e-7: several multigrid cycles per solve, so nrelax adapts and passes of K = 2, 3, 4 half-sweeps all run
    a = run(N, nl, strict, {}, steps=1, tol=1e-7)
    b = run(N, nl, strict, dict(REFERENCE_CHAIN, uniform_S=1) if strict else REFERENCE_CHAIN, steps=1, tol=1e-7)
    if strict:
        assert a["dtmax"] == b["dtmax"] and a["dts"] == b["dts"] and a["st"] == b["st"]
        for k in ("dq", "q", "psi"):
            assert np.array_equal(a[k], b[k]), k
    else:
        assert a["dts"] == pytest.approx(b["dts"], rel=1e-12) and a["st"][0] == b["st"][0]
        for k in ("dq", "q", "psi"):
            assert rel(a[k], b[k]) <= 1e-10, k


@pytest.mark.parametrize("N,nl", [(2048, 6), (4096, 6)])
def test_lean_body_of_the_chained_smoother_equals_the_general_body(N, nl):
    """product build, defaults, reference tolerance: interior chunks through march_lean (requests two steps ahead,
    counted waits, inline-assembly stores) against the same pass with every chunk in the general body, bit for bit --
    the two bodies share every expression"""
    a = run(N, nl, False, dict(march_lean=2), steps=2, tol=1e-3)
    b = run(N, nl, False, dict(march_lean=0), steps=2, tol=1e-3)
    c = run(N, nl, False, dict(march_lean=1), steps=2, tol=1e-3)
    for k in ("dq", "q", "psi"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(c[k], b[k]), k


def test_tendency_kernel_instantiations_round_alike_at_c4():
    """k_rhs_lpw with and without the ghost-line code (lpw_dbg = 4: every wavefront takes the former), product build"""
    a = run(4096, 6, False, dict(lpw_dbg=0), steps=1, tol=1e-3)
    b = run(4096, 6, False, dict(lpw_dbg=4), steps=1, tol=1e-3)
    for k in ("dq", "q", "psi"):
        assert np.array_equal(a[k], b[k]), k


def test_device_noise_at_c5_size_moments_and_tiling_independence():
    """C5, first half (msqg/qg_stochastic.h with the Philox device generator) at 2048^2 x 3: the moments of the noise field
    and its independence of the tiling (2 x 1 tiles through the in-process transport), then the stochastic step itself on
    tiles against the single tile, bit for bit (product build)"""
    from test_gpu_tiled import run_tiled, assemble
    N, nl = 2048, 3
    ex = "tr_stoch = 50\namp_stoch = 2.0\nMGLEVELS = 10\n"
    txt = orc.double_gyre_params(N, nl, extra=ex)
    psi = orc.synthetic_psi(nl, N, N)
    g = QG(txt)
    g.option("quiet", 1); g.option("stochastic", 1); g.option("noise_mode", 1); g.option("seed", 11)
    g.set(F["PSI"], psi); g.set_const()
    g.set(F["SIGMA"], np.full((nl, N, N), 0.5))
    g.set_tnext(float("inf"))
    for _ in range(2):
        g.step()
    noise, q1 = g.get(F["NOISE"]), g.get(F["Q"])
    g.close()
    n = noise / (2.0 * 0.5)
    M = n.size
    assert abs(n.mean()) < 5 / np.sqrt(M) and abs(n.var() - 1) < 5 * np.sqrt(2 / M)
    assert abs(np.mean(n**3)) < 5 * np.sqrt(15 / M) and abs(np.mean(n**4) - 3) < 5 * np.sqrt(96 / M)
    for a, b in ((n[:, :, 1:], n[:, :, :-1]), (n[:, 1:], n[:, :-1]), (n[1:], n[:-1])):
        assert abs(np.mean(a * b)) < 5 / np.sqrt(a.size)
    out = run_tiled(txt, 2, 1, psi, nsteps=2, strict=False, opts={"stochastic": 1, "noise_mode": 1, "seed": 11},
                    fn=lambda g_, r: g_.get(F["NOISE"]), pre=lambda g_, r: g_.set(F["SIGMA"], np.full((nl, N, N // 2), 0.5)))
    got = np.concatenate([out[ix]["extra"] for ix in range(2)], axis=2)
    assert np.array_equal(got, noise)
    assert np.array_equal(assemble(out, "q", 2, 1), q1)
