"""GPU parity of the chained half-sweep smoother (msom_amd/csrc/kernels_march.hip, option march): K = 2..4 red-black
half-sweeps of relax_layer per pass, values of the intermediate half-sweeps only in registers.  Same per-cell arithmetic
as the half-sweep-per-launch path: BIT-EXACT in the strict build (against that path and against the CPU oracle); the
product build differs by fp64 round-off only (FMA contraction is chosen per kernel), tolerance at the assertion.
The pass exists for the uniform-S column solver, which the strict build uses only on request (uniform_S = 1; its default
is the general solver that the oracle follows), so the strict runs below switch it on on both sides."""
import numpy as np
import pytest

import orc
from msom_amd import QG, FIELDS as F
from test_gpu_parity import make_pair, rel

pytestmark = pytest.mark.gpu

SLIP = "sbc = 1.5\nRe = 300\nEks = 0.001\n"


def run(txt, strict, nl, ny, nx, steps=2, **opts):
    g = QG(txt, strict=strict)
    g.option("quiet", 1); g.option("TOLERANCE", 1e-8)
    g.set(F["PSI"], orc.synthetic_psi(nl, ny, nx))
    g.set_const()
    opts.setdefault("uniform_S", 1)
    if opts.get("march") == 0:
        opts.setdefault("block8", 0)      # the reference side of a comparison: one launch per colour on every level
    for k, v in opts.items():
        g.option(k, v)
    if opts.get("march") and opts["uniform_S"] == 1 and nl > 1:
        assert g.param("uniform_S") == 1.0
    if opts.get("march") and opts["uniform_S"] == 1:
        assert g.param("march_levels") >= 1     # the pass really runs (an ignored option would compare a path with itself)
    for _ in range(steps):
        g.step()
    st = g.mgstats()
    out = (g.get(F["PSI"]), g.get(F["Q"]), (st.i, st.resa))
    g.close()
    return out


@pytest.mark.parametrize("nx,ny,nl", [(512, 64, 3), (1024, 128, 6), (512, 512, 2), (2048, 64, 4), (512, 128, 8), (1024, 64, 5), (512, 64, 7)])
@pytest.mark.parametrize("strict", [True, False])
def test_march_equals_half_sweep_per_launch(nx, ny, nl, strict):
    """two RK2 steps (four inversions, TOLERANCE 1e-8 => several cycles): psi, q, cycle count and final residual"""
    txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else ""))
    a = run(txt, strict, nl, ny, nx, march=0)
    b = run(txt, strict, nl, ny, nx, march=2)
    if strict:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    else:
        assert rel(b[0], a[0]) <= 1e-10 and a[2][0] == b[2][0]


@pytest.mark.parametrize("k,rows", [(2, 0), (3, 0), (4, 0), (4, 32), (3, 48), (4, 16), (2, 80)])
@pytest.mark.parametrize("prolong", [0, 1, 2])
def test_march_pass_lengths_and_chunk_heights(k, rows, prolong):
    """the result depends neither on how the 2 nrelax half-sweeps are cut into passes, nor on whether the prolongation
    rides in the first pass (march_prolong) or in a red half-sweep of its own, nor on the chunk height"""
    nx, ny, nl = 512, 128, 3
    txt = orc.double_gyre_params(nx, nl, extra=f"Ny = {ny}\n" + SLIP)
    a = run(txt, True, nl, ny, nx, march=0)
    b = run(txt, True, nl, ny, nx, march=2, march_k=k, march_rows=rows, march_prolong=prolong & 1, march_correct=prolong >> 1)   # 2: correction folded into the last pass
    assert np.array_equal(a[0], b[0]) and a[2] == b[2]


def test_march_against_oracle():
    """product build with the chained smoother against the CPU oracle (general column solver, red-black): 5 steps at
    TOLERANCE 1e-12, <= 1e-10 relative on psi and q -- the bound of test_fast_ten_steps_tight_tolerance"""
    nx, ny, nl = 512, 64, 3
    o, g = make_pair(nx, ny, nl, strict=False, TOLERANCE=1e-12)
    g.option("march", 2)
    for _ in range(5):
        o.step(); g.step()
    assert g.t == pytest.approx(o.t, rel=1e-12)
    assert rel(g.get(F["Q"]), o.get(orc.Q)) <= 1e-10
    assert rel(g.get(F["PSI"]), o.get(orc.PSI)) <= 1e-10


PERIODIC = "sbc = -1\ntau0 = 0\n"   # a periodic box cannot absorb the mean of the double-gyre wind curl


@pytest.mark.parametrize("nx,ny,nl", [(512, 64, 2), (1024, 128, 6), (512, 512, 3), (2048, 64, 4)])
@pytest.mark.parametrize("strict", [True, False])
def test_march_on_the_periodic_domain(nx, ny, nl, strict):
    """sbc = -1 on one tile (round 3): the pass sees a tile without walls, its deep halo is the field's own other side
    (launch_split_wrap); prolongation and correction riders included.  Equal to the half-sweep-per-launch path, whose
    ghost cells are wrapped copies refreshed after every colour"""
    txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + PERIODIC)
    a = run(txt, strict, nl, ny, nx, march=0)
    b = run(txt, strict, nl, ny, nx, march=2)
    if strict:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    else:
        assert rel(b[0], a[0]) <= 1e-10 and a[2][0] == b[2][0]


@pytest.mark.parametrize("k,prolong", [(2, 0), (3, 1), (4, 2), (4, 3), (3, 3)])
def test_march_periodic_pass_lengths_and_riders(k, prolong):
    nx, ny, nl = 512, 128, 3
    txt = orc.double_gyre_params(nx, nl, extra=f"Ny = {ny}\n" + PERIODIC)
    a = run(txt, True, nl, ny, nx, march=0)
    b = run(txt, True, nl, ny, nx, march=2, march_k=k, march_prolong=prolong & 1, march_correct=prolong >> 1)
    assert np.array_equal(a[0], b[0]) and a[2] == b[2]


def test_march_periodic_against_oracle():
    """product build, periodic box, chained smoother forced on, against the CPU oracle (red-black): 4 steps at TOLERANCE 1e-12"""
    nx, ny, nl = 512, 64, 3
    o, g = make_pair(nx, ny, nl, strict=False, extra=PERIODIC, TOLERANCE=1e-12)
    g.option("march", 2)
    for _ in range(4):
        o.step(); g.step()
    assert rel(g.get(F["Q"]), o.get(orc.Q)) <= 1e-10
    assert rel(g.get(F["PSI"]), o.get(orc.PSI)) <= 1e-10


@pytest.mark.parametrize("nx,ny,extra", [(512, 64, ""), (1024, 128, ""), (512, 128, PERIODIC), (2048, 64, SLIP)])
@pytest.mark.parametrize("strict", [True, False])
def test_march_with_one_layer(nx, ny, extra, strict):
    """nl = 1 (round 3): no vertical coupling, the column system is x = rhs / 4; walls and the periodic box"""
    txt = orc.double_gyre_params(nx, 1, extra=f"Ny = {ny}\n" + extra)
    a = run(txt, strict, 1, ny, nx, march=0)
    b = run(txt, strict, 1, ny, nx, march=2)
    if strict:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    else:
        assert rel(b[0], a[0]) <= 1e-10 and a[2][0] == b[2][0]


def test_march_off_where_it_does_not_apply():
    """general S field: the option is ignored (no error, same result as march = 0)"""
    nx, ny, nl = 512, 64, 2
    txt = orc.double_gyre_params(nx, nl, extra=f"Ny = {ny}\n")
    a = run(txt, True, nl, ny, nx, march=0, uniform_S=0)
    b = run(txt, True, nl, ny, nx, march=2, uniform_S=0)
    assert np.array_equal(a[0], b[0])


@pytest.mark.parametrize("seed", range(8))
def test_randomised_product_build_vs_oracle(seed):
    """differential test of the product build's default kernels (chained smoother forced on, one-layer-per-wavefront
    tendency kernel) against the CPU oracle over random parameter combinations (layers, aspect ratio, partial slip, both
    viscosities, drag, 3-D forcing): two RK2 steps at TOLERANCE 1e-11, <= 1e-9 relative on psi and q"""
    rng = np.random.default_rng(4000 + seed)
    nl = int(rng.choice([2, 3, 4, 6]))
    nx, ny = [(512, 64), (512, 128), (1024, 64)][int(rng.integers(3))]
    extra = f"Ny = {ny}\n"
    if rng.random() < 0.5:
        extra += f"sbc = {rng.choice([0.5, 2.0, 100.0])}\n"
    if rng.random() < 0.5:
        extra += f"Re = {rng.choice([200.0, 1500.0])}\n"
    if rng.random() < 0.3:
        extra += "Re4 = 0\n"
    if rng.random() < 0.5:
        extra += f"Eks = {rng.choice([0.001, 0.01])}\n"
    txt = orc.double_gyre_params(nx, nl, extra=extra)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
    g = QG(txt)
    g.option("quiet", 1); g.option("march", 2)
    for h in (o, g):
        h.option("TOLERANCE", 1e-11)
    p0 = orc.synthetic_psi(nl, ny, nx)
    o.set(orc.PSI, p0); g.set(F["PSI"], p0)
    if rng.random() < 0.5:
        qf = 1e-6 * rng.standard_normal((nl, ny, nx))
        o.set(orc.QFORC, qf); g.set(F["QFORC"], qf)
    o.set_const(); g.set_const()
    assert g.param("uniform_S") == 1.0
    for _ in range(2):
        o.step(); g.step()
    desc = f"nl={nl} {nx}x{ny} {extra!r}"
    assert g.t == pytest.approx(o.t, rel=1e-12), desc
    assert rel(g.get(F["PSI"]), o.get(orc.PSI)) <= 1e-9, desc
    assert rel(g.get(F["Q"]), o.get(orc.Q)) <= 1e-9, desc


@pytest.mark.parametrize("nx,ny,nl", [(64, 64, 3), (128, 64, 2), (256, 256, 6), (512, 128, 1), (256, 64, 4), (1024, 128, 5), (128, 128, 6), (64, 16, 3), (1024, 16, 2), (32, 128, 4), (256, 128, 8), (128, 128, 7)])
@pytest.mark.parametrize("strict", [True, False])
def test_tiled_level_visit_equals_half_sweep_per_launch(nx, ny, nl, strict):
    """option block8 (default on, round 3): prolongation + all (up to 8) half-sweeps of a visit of a launch-bound level in ONE launch of
    the LDS-tiled smoother with a halo of 8 (k_relax_block<.., 8>; more half-sweeps: further passes, a single left-over one as a colour
    pass).  TOLERANCE 1e-8: several cycles per solve, nrelax adapts (4, 5, ... sweeps => 8, 8 + 2, ... half-sweeps).  Against one launch
    per colour: bit for bit in the strict build, round-off in the product build; every tile shape (block_variant)"""
    txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + SLIP)
    a = run(txt, strict, nl, ny, nx, march=0, block8=0)
    for variant in (0, 1, 3, 6):
        b = run(txt, strict, nl, ny, nx, march=0, block8=1, block_variant=variant)
        if strict:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], variant
        else:
            assert rel(b[0], a[0]) <= 1e-10 and a[2][0] == b[2][0], variant
    g = QG(txt); g.option("block_variant", 0); g.close()      # the knob is a global of the library: back to the default


@pytest.mark.parametrize("nx,ny,nl", [(64, 64, 3), (256, 128, 2), (512, 512, 6), (128, 64, 1)])
@pytest.mark.parametrize("strict", [True, False])
def test_tiled_level_visit_on_the_periodic_domain(nx, ny, nl, strict):
    """sbc = -1 (doubly periodic single tile): the region of a tile beyond the domain holds periodic images loaded from their wrapped
    positions; the ghost lines are refreshed after each pass (launch_split_wrap).  Levels of fewer than 64 rows keep the colour launches"""
    txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + "sbc = -1\ntau0 = 0\n")
    a = run(txt, strict, nl, ny, nx, march=0, block8=0)
    for variant in (0, 6):
        b = run(txt, strict, nl, ny, nx, march=0, block8=1, block_variant=variant)
        if strict:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], variant
        else:
            assert rel(b[0], a[0]) <= 1e-10 and a[2][0] == b[2][0], variant
    g = QG(txt); g.option("block_variant", 0); g.close()


def test_tiled_level_visit_against_oracle():
    """product build, defaults, against the CPU oracle (general column solver, red-black) at 256^2 x 3 (levels 64 .. 256 in the
    tiled pass, <= 32 in the one-launch coarse kernel): 5 steps at TOLERANCE 1e-12, <= 1e-10 relative on psi and q -- the bound of
    test_fast_ten_steps_tight_tolerance"""
    nx, ny, nl = 256, 256, 3
    o, g = make_pair(nx, ny, nl, strict=False, TOLERANCE=1e-12)
    assert g.param("uniform_S") == 1.0
    for _ in range(5):
        o.step(); g.step()
    assert g.t == pytest.approx(o.t, rel=1e-12)
    assert rel(g.get(F["Q"]), o.get(orc.Q)) <= 1e-10
    assert rel(g.get(F["PSI"]), o.get(orc.PSI)) <= 1e-10


@pytest.mark.parametrize("opts", [dict(prolong_fused=0), dict(mg_fused=0), dict(mg_coarse=0), dict(mg_coarse=2, restrict2=0), dict(async_solve=0, fused=0),
                                  dict(block8_max=128), dict(mg_coarse_dim=16), dict(graph=1)], ids=str)
def test_tiled_level_visit_beside_the_other_switches(opts):
    """the round-3 defaults (block8, restrict2, lean coarse kernel) with one older switch flipped at a time, against the kernel-per-loop
    chain: validation build with uniform S, bit for bit; 256 x 128 x 3 and 512 x 512 x 2, several cycles per solve"""
    for nx, ny, nl in ((256, 128, 3), (512, 512, 2)):
        txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + SLIP)
        a = run(txt, True, nl, ny, nx, march=0, block8=0, mg_coarse=0, mg_fused=0, prolong_fused=0, fused=0, adv_fused=0)
        b = run(txt, True, nl, ny, nx, **opts)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (nx, opts)
