"""GPU parity of the vertex-grid variant (msomn_* C ABI, kernels_node.hip) against the CPU oracle
oracle/qgnode_oracle.c run with the same red-black smoother.

strict build (-ffp-contract=off, reference expression order): bit-exact (np.array_equal).
fast build (FMA contraction, reciprocal multiplies): relative tolerance 1e-9 on the fields after
the elliptic solve converged to the same TOLERANCE (stated per test)."""
import numpy as np
import pytest

import orn
from msom_amd import NodeQG

pytestmark = pytest.mark.gpu


def land_mask(N):
    """an island and a ragged coast: mask = 0 on land vertices and on the boundary vertices"""
    m = np.ones((1, N + 1, N + 1))
    m[0, N // 4: N // 4 + N // 8 + 1, N // 2: N // 2 + N // 8] = 0
    m[0, : N // 6, : N // 5] = 0
    m[0, 0, :] = m[0, -1, :] = m[0, :, 0] = m[0, :, -1] = 0
    return m


def topo_field(N):
    x = np.arange(N + 1) / N
    return 0.05 * np.outer(np.cos(2 * np.pi * x), np.sin(np.pi * x))[None]


def make_pair(N, nl, strict, bc_fac=0.0, nu4=0.0, mask=False, topo=False, pg=False, extra="", s2x=False, **opts):
    par = orn.node_params(N, nl, bc_fac=bc_fac, nu4=nu4, extra=extra)
    o = orn.NodeOracle(par, smoother=orn.GS_RB, quiet=1, **opts)
    g = NodeQG(par, strict=strict)
    g.set_option("quiet", 1)
    for k, v in opts.items():
        g.set_option(k, v)
    psi = orn.node_psi(nl, N)
    if mask:
        mk = land_mask(N)
        psi = psi * mk
        o.set(orn.MASK, mk); g.set("MASK", mk)
    if topo and nl > 1:
        tp = topo_field(N)
        o.set(orn.TOPO, tp); g.set("TOPO", tp)
    if pg and nl > 1:
        pgf = 0.3 * orn.node_psi(nl, N)[::-1].copy()
        o.set(orn.PSIPG, pgf); g.set("PSIPG", pgf)
    if s2x and nl > 1:      # N2 that varies in x (the reference's is one number per interface): no row tables
        n2 = o.get(orn.S2)
        n2 *= 1.0 + 0.3 * np.cos(np.linspace(0, 5, N + 1))[None, None, :]
        o.set(orn.S2, n2); g.set("S2", n2)
    o.set(orn.PSI, psi); g.set("PSI", psi)
    o.set_const(); g.set_const()
    return o, g


def same(a, b, strict, rtol=1e-9):
    if strict:
        assert np.array_equal(a, b), f"max diff {np.abs(a - b).max():g}"
    else:
        assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-300), f"rel diff {np.abs(a - b).max() / np.abs(b).max():g}"


FIELDS = [("PSI", orn.PSI), ("Q", orn.Q)]


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl", [1, 2, 3, 4])
def test_set_const_and_comp_q(nl, strict):
    o, g = make_pair(32, nl, strict, bc_fac=0.5, mask=True, extra="flag_ms = 1\ngp_low = 0.02\n")
    assert g.param("DT") == o.param("DT")
    assert g.param("iRd2_low") == o.param("iRd2_low")
    for l in range(nl):
        assert g.param(f"idh0_{l}") == o.param(f"idh0_{l}") and g.param(f"idh1_{l}") == o.param(f"idh1_{l}")
    if nl > 1:
        same(g.get("S2"), o.get(orn.S2), True)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-12)
    for k in range(g.nlevels):
        assert np.array_equal(g.dbg_level_mask(k), o.level_mask(k))


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl", [1, 3])
def test_multigrid_pieces(nl, strict):
    N = 32
    o, g = make_pair(N, nl, strict, mask=True, extra="gp_low = 0.02\n")
    rng = np.random.default_rng(5)
    for k in range(g.nlevels):
        n1 = (N >> k) + 1
        da, res = rng.standard_normal((nl, n1, n1)), rng.standard_normal((nl, n1, n1))
        same(g.dbg_relax(k, da, res, 2), o.relax(k, da, res, 2), strict, 1e-11)
        if k + 1 < g.nlevels:
            same(g.dbg_restrict(k, res), o.restrict(k, res), strict, 1e-13)
        if k > 0:
            same(g.dbg_prolong(k, da), o.prolong(k, da), strict, 1e-13)
    a, b = rng.standard_normal((nl, N + 1, N + 1)), rng.standard_normal((nl, N + 1, N + 1))
    rg, mg = g.dbg_residual(a, b)
    ro, mo = o.residual(a, b)
    same(rg, ro, strict, 1e-12)
    assert mg == np.abs(rg).max()
    if strict:
        assert mg == mo


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N", [(1, 128), (3, 256), (6, 128), (8, 64)])
def test_tiled_smoother_passes(nl, N, strict):
    """k_n_relax_tile (1 or 2 red-black sweeps per pass through an LDS tile, out of place) on the wide levels:
    1, 2, 3 and 5 sweeps against the oracle's colour-by-colour sweeps, and against the per-colour kernels"""
    if nl not in orn.NODE_LAYERS:
        orn.NODE_LAYERS[nl] = ("[" + ",".join(["%.3f" % (1.0 / nl)] * nl) + "]", "[" + ",".join(["%d." % (9000 - 900 * l) for l in range(nl - 1)]) + "]")
    o, g = make_pair(N, nl, strict, mask=True, extra="gp_low = 0.02\n")
    g.set_option("node_split", 0)       # the split layout has its own colour passes
    g.set_option("tiled_relax", 1)
    rng = np.random.default_rng(11)
    for k in (0, 1):
        n1 = (N >> k) + 1
        da, res = rng.standard_normal((nl, n1, n1)), rng.standard_normal((nl, n1, n1))
        for ns in (1, 2, 3, 5):
            got, ref = g.dbg_relax(k, da, res, ns), o.relax(k, da, res, ns)
            same(got, ref, strict, 1e-10)
            g.set_option("tiled_relax", 0)
            plain = g.dbg_relax(k, da, res, ns)
            g.set_option("tiled_relax", 1)
            assert np.array_equal(got, plain) or not strict


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,bc_fac,mask", [(1, 0.0, False), (1, 1.0, True), (2, 0.0, True), (3, 1.0, False), (4, 0.5, True)])
def test_invert_and_rhs(nl, bc_fac, mask, strict):
    o, g = make_pair(32, nl, strict, bc_fac=bc_fac, nu4=2.0, mask=mask, topo=True, pg=True, extra="gp_low = 0.02\n", TOLERANCE=1e-9)
    o.forcing(); g.forcing()
    same(g.get("QFORC"), o.get(orn.QFORC), True)
    so, sg = o.invert_q(), g.invert_q()
    if strict:
        assert (sg.i, sg.resb, sg.resa) == (so.i, so.resb, so.resa)
    else:
        assert abs(sg.i - so.i) <= 1 and sg.resa < 1e-9
    same(g.get("PSI"), o.get(orn.PSI), strict, 1e-7)
    o.rhs_pv(); g.rhs_pv()
    if nl > 1:
        same(g.get("ZETA"), o.get(orn.ZETA), strict, 1e-7)
    same(g.get("DQ"), o.get(orn.DQ), strict, 1e-6)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N", [(1, 64), (3, 32), (4, 64)])
def test_time_steps(nl, N, strict):
    o, g = make_pair(N, nl, strict, bc_fac=1.0, nu4=1.0, mask=True, topo=True, extra="gp_low = 0.02\ntau1 = 5e-4\ntf1 = 0.3\ntf2 = 0.7\n", TOLERANCE=1e-8)
    o.set_tnext(0.11); g.set_tnext(0.11)
    for it in range(6):
        o.step(True); g.step(True)
        if strict:
            assert (g.t, g.dt) == (o.t, o.dt)
            assert g.mgstats().i == o.mgstats().i
        else:
            assert abs(g.t - o.t) <= 1e-12 * o.t and abs(g.dt - o.dt) <= 1e-9 * o.dt
    for gf, of in FIELDS:
        same(g.get(gf), o.get(of), strict, 1e-6)
    ke_o = o.ke()
    assert abs(g.ke() - ke_o) <= 1e-10 * abs(ke_o)
    d_o, d_g = o.diag1d(), g.diag1d()                # event write_1d_diag: ke, dissipation, forcing
    assert np.all(np.abs(d_g - d_o) <= 1e-10 * np.abs(d_o).max()) and np.all(d_o != 0)
    assert g.iter == 6


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("N,L_filt", [(64, 8.0), (128, 1e6), (32, 20.0)])
def test_stochastic_forcing(N, L_filt, strict):
    """-D_STOCHASTIC of the vertex model (qg-node/qg_stochastic.h, qg.h:306-320): wavelet coefficients, the filtered
    cell noise, and four RK2 steps driven by the reference-exact serial rand() stream (same seed on both sides)"""
    import ctypes
    libc = ctypes.CDLL(None)
    par = orn.node_params(N, 1, bc_fac=1.0, extra=f"gp_low = 0.02\namp_stoch = 0.3\nL_filt = {L_filt}\n")
    o = orn.NodeOracle(par, smoother=orn.GS_RB, quiet=1, stochastic=1, TOLERANCE=1e-9)
    g = NodeQG(par, strict=strict)
    for k, v in (("quiet", 1), ("stochastic", 1), ("TOLERANCE", 1e-9)):
        g.set_option(k, v)
    psi = orn.node_psi(1, N)
    o.set(orn.PSI, psi); g.set("PSI", psi)
    o.set_const(); g.set_const()
    for k in range(o.cell_levels()):
        assert np.array_equal(g.csig(k), o.csig(k))
    n0 = np.random.default_rng(3).standard_normal((N, N))
    o.set_noise(n0); o.filter_noise()
    same(g.noise(set=n0, filter=True), o.noise(), strict, 1e-13)
    for m in (o, g):
        libc.srand(11)
        for _ in range(4):
            m.step(True)
    same(g.noise(), o.noise(), strict, 1e-13)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-7)
    same(g.get("PSI"), o.get(orn.PSI), strict, 1e-7)
    assert np.abs(g.noise()).max() > 0


@pytest.mark.parametrize("strict", [True, False])
def test_forcing_3d(strict):
    """-DFORCING_3D (qg_baroclinic_ms.h:179-185): q_forcing_3d added to every layer of the tendency before the mask"""
    N, nl = 32, 3
    o, g = make_pair(N, nl, strict, mask=True, TOLERANCE=1e-9)
    f3 = 1e-3 * np.random.default_rng(8).standard_normal((nl, N + 1, N + 1))
    o.option("forcing_3d", 1); o.set(orn.QFORC3D, f3); g.set("QFORC3D", f3)
    ref = orn.NodeOracle(orn.node_params(N, nl), smoother=orn.GS_RB, quiet=1, TOLERANCE=1e-9)
    for _ in range(2):
        o.step(True); g.step(True)
    same(g.get("Q"), o.get(orn.Q), strict, 1e-7)
    o.rhs_pv(); g.rhs_pv()
    same(g.get("DQ"), o.get(orn.DQ), strict, 1e-7)
    assert np.abs(o.get(orn.DQ)).max() > 1e-4


def test_full_size_properties():
    """2048^2 x 3 vertex grid with an island: the elliptic solve converges, q -> psi -> q closes, KE finite"""
    N, nl = 2048, 3
    par = orn.node_params(N, nl, bc_fac=1.0)
    g = NodeQG(par)
    g.set_option("quiet", 1)
    g.set_option("TOLERANCE", 1e-7)
    mk = land_mask(N)
    psi = orn.node_psi(nl, N) * mk
    g.set("MASK", mk)
    g.set("PSI", psi)
    g.set_const()
    q0 = g.get("Q")
    g.set("PSI", np.zeros_like(psi))
    st = g.invert_q()
    assert st.resa < 1e-7 and st.i < 30
    inner = mk[0] == 1
    p1 = g.get("PSI")
    assert np.all(p1[:, ~inner] == 0)
    g.comp_q()
    q1 = g.get("Q")
    assert np.abs((q1 - q0)[:, inner]).max() <= 2e-7
    for _ in range(2):
        g.step(True)
    assert np.isfinite(g.ke()) and g.t > 0


def test_netcdf_round_trip_and_driver(tmp_path):
    from scipy.io import netcdf_file
    N, nl = 32, 2
    par = orn.node_params(N, nl, extra="noise_init = 1e-3\ntend = 0.2\ndtout = 0.1\ndtdiag = 0.05\n")
    g = NodeQG(par)
    g.set_option("quiet", 1)
    n = g.run(str(tmp_path))
    assert n > 0 and abs(g.t - 0.2) < 1e-12
    out = tmp_path / "outdir_0001"
    assert (out / "params.in").read_text() == par
    rows = (out / "diag_1d.dat").read_text().splitlines()
    assert rows[0].startswith("# time, ke") and len(rows) == 5 and abs(float(rows[-1].split(",")[0]) - 0.2) < 1e-9
    with netcdf_file(str(out / "vars.nc"), "r", mmap=False) as nc:
        assert nc.variables["psi"].shape == (3, nl, N + 1, N + 1)
        assert np.allclose(nc.variables["time"][:], [0, 0.1, 0.2])
        assert np.allclose(nc.variables["x"][:], np.arange(N + 1) * 100.0 / N)
        last = np.array(nc.variables["psi"][-1], dtype=np.float64)
    assert np.allclose(last, g.get("PSI"), rtol=1e-6, atol=1e-9)
    g2 = NodeQG(par)
    g2.read_nc("PSI", str(out / "vars.nc"), "psi")
    assert np.allclose(g2.get("PSI"), last)


def test_error_convention():
    from msom_amd import MsomError
    with pytest.raises(MsomError, match="power of two"):
        NodeQG(orn.node_params(48, 2))
    with pytest.raises(MsomError, match="periodic"):
        NodeQG(orn.node_params(32, 2, bc_fac=-1))
    g = NodeQG(orn.node_params(16, 2))
    with pytest.raises(MsomError, match="set_const"):
        g.step()


@pytest.mark.parametrize("seed", range(8))
def test_randomised_configurations_strict_vs_oracle(seed):
    """vertex model: random layer counts, sizes, slip factors, viscosities, ragged land masks, topography, background
    flow, beta-plane S2 (flag_ms), time-dependent forcing; three RK2 steps, strict build bit-exact against the oracle"""
    rng = np.random.default_rng(500 + seed)
    nl = int(rng.choice([1, 2, 3, 4]))
    N = int(rng.choice([16, 32, 64, 128]))
    extra = f"gp_low = {rng.choice([0.0, 0.02])}\ntau1 = {rng.choice([0.0, 5e-4])}\ntf1 = 0.4\ntf2 = 0.9\nflag_ms = {int(rng.integers(0, 2))}\n"
    par = orn.node_params(N, nl, bc_fac=float(rng.choice([0.0, 0.5, 1.0])), nu=float(rng.choice([2.0, 5.0])), nu4=float(rng.choice([0.0, 1.5])), extra=extra)
    tol = float(rng.choice([1e-5, 1e-9]))
    o = orn.NodeOracle(par, smoother=orn.GS_RB, quiet=1, TOLERANCE=tol)
    g = NodeQG(par, strict=True)
    g.set_option("quiet", 1); g.set_option("TOLERANCE", tol)
    mk = np.ones((1, N + 1, N + 1))
    for _ in range(int(rng.integers(0, 4))):       # random rectangular islands
        i0, j0 = rng.integers(1, N - 3, 2)
        mk[0, j0: j0 + rng.integers(1, N // 4 + 1), i0: i0 + rng.integers(1, N // 4 + 1)] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    psi = orn.node_psi(nl, N) * mk
    for m_, set_ in ((o, lambda f, a: o.set(getattr(orn, f), a)), (g, lambda f, a: g.set(f, a))):
        set_("MASK", mk)
        if nl > 1 and seed % 2 == 0:
            set_("TOPO", topo_field(N))
            set_("PSIPG", 0.3 * orn.node_psi(nl, N)[::-1].copy())
        set_("PSI", psi)
        m_.set_const()
    for _ in range(3):
        o.step(True); g.step(True)
    desc = f"nl={nl} N={N} {par[-120:]!r}"
    assert (g.t, g.dt) == (o.t, o.dt), desc
    assert np.array_equal(g.get("PSI"), o.get(orn.PSI)) and np.array_equal(g.get("Q"), o.get(orn.Q)), desc


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N", [(1, 128), (3, 256), (6, 128), (8, 64), (2, 512)])
def test_chained_half_sweep_smoother(nl, N, strict):
    """k_n_relax_march (K = 2..4 colour half-sweeps per pass, register windows, out of place) on the wide levels: 1, 2, 3 and
    5 sweeps (2, 4, 4 + 2, 4 + 4 + 2 half-sweeps) against the oracle's colour-by-colour sweeps and against the per-colour kernels"""
    if nl not in orn.NODE_LAYERS:
        orn.NODE_LAYERS[nl] = ("[" + ",".join(["%.3f" % (1.0 / nl)] * nl) + "]", "[" + ",".join(["%d." % (9000 - 900 * l) for l in range(nl - 1)]) + "]")
    o, g = make_pair(N, nl, strict, mask=True, extra="gp_low = 0.02\n")
    g.set_option("node_split", 0)
    rng = np.random.default_rng(12)
    for k in (0, 1):
        n1 = (N >> k) + 1
        if n1 < 64:
            continue
        da, res = rng.standard_normal((nl, n1, n1)), rng.standard_normal((nl, n1, n1))
        for ns in (1, 2, 3, 5):
            g.set_option("node_march", 64)
            got, ref = g.dbg_relax(k, da, res, ns), o.relax(k, da, res, ns)
            same(got, ref, strict, 1e-10)
            g.set_option("node_march", 0)
            plain = g.dbg_relax(k, da, res, ns)
            assert np.array_equal(got, plain) or not strict


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N,split,s2x", [(1, 128, 65, False), (3, 256, 65, False), (6, 128, 129, False), (2, 512, 129, True), (3, 64, 65, False),
                                             (3, 128, 65, True), (4, 128, 0, True)])
def test_split_layout_levels(nl, N, split, s2x, strict):
    """Wide levels keep da / res / mask / S2 in the x-parity split layout (option node_split): every multigrid piece on every
    level against the oracle, then inversion and RK2 steps against the oracle and against the natural layout (identical:
    the layout changes no arithmetic)."""
    if nl not in orn.NODE_LAYERS:
        orn.NODE_LAYERS[nl] = ("[" + ",".join(["%.3f" % (1.0 / nl)] * nl) + "]", "[" + ",".join(["%d." % (9000 - 900 * l) for l in range(nl - 1)]) + "]")
    o, g = make_pair(N, nl, strict, mask=True, bc_fac=0.5, extra="gp_low = 0.02\n", s2x=s2x)
    assert g.param("s2_xuniform") == (0.0 if s2x or nl == 1 else 1.0)
    g.set_option("node_split", split)
    rng = np.random.default_rng(21)
    for k in range(g.nlevels):
        n1 = (N >> k) + 1
        da, res = rng.standard_normal((nl, n1, n1)), rng.standard_normal((nl, n1, n1))
        same(g.dbg_relax(k, da, res, 2), o.relax(k, da, res, 2), strict, 1e-11)
        if k + 1 < g.nlevels:
            same(g.dbg_restrict(k, res), o.restrict(k, res), strict, 1e-13)
        if k > 0:
            same(g.dbg_prolong(k, da), o.prolong(k, da), strict, 1e-13)
    a, b = rng.standard_normal((nl, N + 1, N + 1)), rng.standard_normal((nl, N + 1, N + 1))
    rg, mg = g.dbg_residual(a, b)
    ro, mo = o.residual(a, b)
    same(rg, ro, strict, 1e-12)
    assert mg == np.abs(rg).max()
    o.set_tnext(float("inf")); g.set_tnext(float("inf"))
    for _ in range(3):
        o.step(True); g.step(True)
    for name, idx in FIELDS:
        same(g.get(name), o.get(idx), strict, 1e-6)
    # natural layout from the same start: the same numbers in both builds
    o2, g2 = make_pair(N, nl, strict, mask=True, bc_fac=0.5, extra="gp_low = 0.02\n", s2x=s2x)
    g2.set_option("node_split", 0 if nl != 3 else split)      # nl = 3: split layout again, but with the prolongation as its own launch
    g2.set_option("node_pfused", 0)
    g2.set_option("s2_rows", 0)
    g2.set_tnext(float("inf"))
    for _ in range(3):
        g2.step(True)
    for name, _ in FIELDS:
        assert np.array_equal(g2.get(name), g.get(name))
    assert (g2.mgstats().i, g2.mgstats().resa) == (g.mgstats().i, g.mgstats().resa)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N", [(1, 128), (3, 256), (6, 128), (2, 512), (3, 1024)])
def test_chained_half_sweeps_in_the_split_layout(nl, N, strict):
    """k_n_relax_march_s (round 3): K = 2..4 colour half-sweeps of a split level per pass, lane = vertex pair, residual and mask in
    register delay lines, S2 by row tables.  1, 2, 3 and 5 sweeps against the oracle's colour-by-colour sweeps and against the
    colour-per-launch kernels, level by level; then whole RK2 steps (prolongation folded into the first colour, then passes of
    4 + 4 + one single colour) against the oracle and against the run without the pass"""
    if nl not in orn.NODE_LAYERS:
        orn.NODE_LAYERS[nl] = ("[" + ",".join(["%.3f" % (1.0 / nl)] * nl) + "]", "[" + ",".join(["%d." % (9000 - 900 * l) for l in range(nl - 1)]) + "]")
    o, g = make_pair(N, nl, strict, mask=True, bc_fac=0.5, extra="gp_low = 0.02\n")
    g.set_option("node_split", 65)
    rng = np.random.default_rng(33)
    for k in (0, 1):
        n1 = (N >> k) + 1
        if n1 < 65:
            continue
        da, res = rng.standard_normal((nl, n1, n1)), rng.standard_normal((nl, n1, n1))
        da[:, 0, :] = da[:, -1, :] = da[:, :, 0] = da[:, :, -1] = 0      # the correction's boundary vertices are 0 (boundary_level)
        for ns in (1, 2, 3, 5):
            g.set_option("node_march_s", 65)
            got, ref = g.dbg_relax(k, da, res, ns), o.relax(k, da, res, ns)
            same(got, ref, strict, 1e-10)
            g.set_option("node_march_s", 0)         # -> LDS-tiled passes of up to 4 half-sweeps (k_n_relax_tile_s), nl <= 4
            tiled = g.dbg_relax(k, da, res, ns)
            same(tiled, ref, strict, 1e-10)
            g.set_option("node_tile_s", 0)          # -> one launch per colour
            plain = g.dbg_relax(k, da, res, ns)
            g.set_option("node_tile_s", 65)
            assert (np.array_equal(got, plain) and np.array_equal(tiled, plain)) or not strict
    g.set_option("node_march_s", 65)
    o.set_tnext(float("inf")); g.set_tnext(float("inf"))
    for _ in range(2):
        o.step(True); g.step(True)
    for name, idx in FIELDS:
        same(g.get(name), o.get(idx), strict, 1e-6)
    o3, g3 = make_pair(N, nl, strict, mask=True, bc_fac=0.5, extra="gp_low = 0.02\n")     # the other way of cutting 9 half-sweeps into passes: 4 + 3 + 2
    g3.set_option("node_split", 65); g3.set_option("node_march_s", 65); g3.set_option("node_march_tail1", 0)
    g3.set_tnext(float("inf"))
    for _ in range(2):
        g3.step(True)
    for name, _ in FIELDS:
        assert np.array_equal(g3.get(name), g.get(name)) or not strict
    for tile in (65, 0):        # without the chained pass: the tiled passes on every split level, then one launch per colour
        o2, g2 = make_pair(N, nl, strict, mask=True, bc_fac=0.5, extra="gp_low = 0.02\n")
        g2.set_option("node_split", 65)
        g2.set_option("node_march_s", 1 << 20)
        g2.set_option("node_tile_s", tile)
        g2.set_tnext(float("inf"))
        for _ in range(2):
            g2.step(True)
        for name, _ in FIELDS:
            assert np.array_equal(g2.get(name), g.get(name)) or not strict
        assert g2.mgstats().i == g.mgstats().i


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,N,split,nitermax", [(1, 256, 65, 100), (3, 128, 65, 100), (3, 256, 65, 3), (4, 64, 0, 2), (6, 128, 65, 100)])
def test_correction_riding_in_the_next_residual_pass(nl, N, split, nitermax, strict):
    """k_n_correct_residual (round 3): a += da of cycle i is applied inside the residual pass of cycle i + 1 (second psi buffer,
    swapped); the last correction of a solve that ends on the iteration count is the plain kernel.  TOLERANCE 1e-10: several cycles
    per solve.  Against the oracle and (strict build: bit for bit) against the run with separate passes"""
    if nl not in orn.NODE_LAYERS:
        orn.NODE_LAYERS[nl] = ("[" + ",".join(["%.3f" % (1.0 / nl)] * nl) + "]", "[" + ",".join(["%d." % (9000 - 900 * l) for l in range(nl - 1)]) + "]")
    kw = dict(mask=True, bc_fac=0.5, topo=True, extra="gp_low = 0.02\n", TOLERANCE=1e-10, NITERMAX=nitermax)
    o, g = make_pair(N, nl, strict, **kw)
    _, g2 = make_pair(N, nl, strict, **kw)
    _, g3 = make_pair(N, nl, strict, **kw)
    for h, on in ((g, 2), (g2, 0), (g3, 1)):       # 2 (default): the row-marching kernel, 1: one thread per vertex, 0: separate passes
        h.set_option("node_split", split if split else 1 << 20)
        h.set_option("node_corr_fused", on)
        h.set_option("node_rhs_fused", on)         # and the tendency in three passes against the twelve loops
        h.set_tnext(float("inf"))
    o.set_tnext(float("inf"))
    cyc = 0
    for _ in range(3):
        o.step(True); g.step(True); g2.step(True); g3.step(True)
        assert g.mgstats().i == g2.mgstats().i == g3.mgstats().i
        assert g.mgstats().i == o.mgstats().i or not strict
        cyc = max(cyc, g.mgstats().i)
    assert cyc >= 2 or nl == 1          # the fused pass ran (the one-layer Helmholtz problem converges in one cycle)
    for name, idx in FIELDS:
        same(g.get(name), g2.get(name), strict, 1e-9)       # product build: the compiler contracts the merged expressions differently
        same(g3.get(name), g2.get(name), strict, 1e-9)
        same(g.get(name), o.get(idx), strict, 1e-6)
