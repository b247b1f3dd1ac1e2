"""Known-answer tests that pin the CPU oracle (oracle/qg_oracle.c).

The reference ships no golden vectors for this path ("parity unpinned"), so the
oracle is pinned against analytic answers of the discrete operators it restates:
eigenfunctions of the 5-point Laplacian with the wall on the cell face (msqg/layer.h:13-21),
vertical eigenvectors of the stretching operator (msqg/qg.h:203-246), Arakawa's conservation
identities (msqg/qg.h:252-262), dense-matrix solves of the layered elliptic problem
(msqg/poisson_layer.h), and the dt-limiter recurrence (newqg/qg.h:202-219).
"""
import numpy as np
import pytest

import orc


def params(N=16, nl=3, extra="", **kw):
    base = dict(L0=80, Rom=0.025, Ekb=0.002, tau0=1e-4, beta=0.5, DT=5e-2, CFL=0.6)
    base.update(kw)
    Fr, dh = orc.LAYERS[nl]
    txt = f"N = {N}\nnl = {nl}\nFr = {Fr}\ndh = {dh}\n"
    txt += "".join(f"{k} = {v}\n" for k, v in base.items())
    return txt + extra


def gamma_matrix(o):
    """Dense nl x nl matrix of the stretching operator for uniform S (msqg/qg.h:216-237)."""
    nl = o.nl
    S = [(o.param(f"Fr_{l}") / o.param("Rom")) ** 2 for l in range(nl - 1)]
    G = np.zeros((nl, nl))
    for l in range(nl):
        if l > 0:
            c = S[l - 1] * o.param(f"idh0_{l}")
            G[l, l - 1] += c
            G[l, l] -= c
        if l < nl - 1:
            c = S[l] * o.param(f"idh1_{l}")
            G[l, l + 1] += c
            G[l, l] -= c
    return G


def test_params_derived_values():
    # msqg/test/params.double_gyre.in + msqg/qg.h:739-746: DT = 0.5*min(DT, D^4 Re4/32)
    o = orc.Oracle(orc.double_gyre_params(256, 3))
    assert o.nx == 256 and o.nl == 3
    D = 80.0 / 256
    assert o.param("DT") == pytest.approx(0.5 * min(5e-2, D**4 * 1563 / 32), rel=1e-15)
    assert o.param("DT") == pytest.approx(0.025)
    assert o.param("iRe4") == pytest.approx(-1 / 1563.0)
    assert o.param("iRe") == 0.0
    o.set_const()
    # layer metrics msqg/qg.h:1017-1027
    dh = [0.06, 0.14, 0.8]
    dhc = [0.5 * (dh[0] + dh[1]), 0.5 * (dh[1] + dh[2])]
    assert o.param("idh0_0") == 0.0
    assert o.param("idh1_0") == pytest.approx(1 / (dhc[0] * dh[0]), rel=1e-15)
    assert o.param("idh0_1") == pytest.approx(1 / (dhc[0] * dh[1]), rel=1e-15)
    assert o.param("idh1_1") == pytest.approx(1 / (dhc[1] * dh[1]), rel=1e-15)
    assert o.param("idh0_2") == pytest.approx(1 / (dhc[1] * dh[2]), rel=1e-15)
    assert o.param("idh1_2") == 0.0
    S = o.get(orc.S)
    assert S.shape == (2, 256, 256)
    assert np.allclose(S[0], (0.0023669 / 0.025) ** 2, rtol=1e-15)
    assert np.allclose(S[1], (0.0076173 / 0.025) ** 2, rtol=1e-15)


def test_parser_ignores_comments_and_unknown_keys():
    txt = "#!sh\n# comment = 3\n N = 32 \nnl=2\nbogus = 7\nFr = [ 0.1 ]\ndh = [0.5, 0.5]\nRom = 0.1\nL0 = 2\n"
    o = orc.Oracle(txt)
    assert (o.nx, o.ny, o.nl) == (32, 32, 2)
    assert o.param("L0") == 2.0 and o.param("Fr_0") == 0.1 and o.param("dh_1") == 0.5
    assert o.param("DT") == 1e10  # Basilisk default, no viscous clamp


@pytest.mark.parametrize("nl", [2, 3, 6])
def test_laplacian_and_stretching_eigenfunctions(nl):
    N = 32
    o = orc.Oracle(params(N, nl))
    o.set_const()
    G = gamma_matrix(o)
    gam, vec = np.linalg.eig(G)
    D = 80.0 / N
    x = (np.arange(N) + 0.5) / N
    for (k, m, iv) in [(1, 1, 0), (3, 2, 1), (N, N, nl - 1), (5, 7, nl // 2)]:
        h = np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
        psi = vec[:, iv].real[:, None, None] * h[None]
        lam = -(4 / D**2) * (np.sin(k * np.pi / (2 * N)) ** 2 + np.sin(m * np.pi / (2 * N)) ** 2)
        q = o.pyp2q(psi)
        scale = np.abs(psi).max() * (abs(lam) + abs(gam[iv].real))
        assert np.abs(q - (lam + gam[iv].real) * psi).max() <= 1e-12 * max(scale, 1e-300)
        # comp_del2 alone
        o.set(orc.PSI, psi)
        o.comp_del2(orc.PSI, orc.ZETA, 0.0, 1.0)
        assert np.abs(o.get(orc.ZETA) - lam * psi).max() <= 1e-12 * np.abs(lam * psi).max()


def test_invertq_recovers_eigenfunction():
    N, nl = 32, 3
    o = orc.Oracle(params(N, nl), smoother=orc.GS_LEX, TOLERANCE=1e-13, quiet=1)
    o.set_const()
    G = gamma_matrix(o)
    gam, vec = np.linalg.eig(G)
    D = 80.0 / N
    x = (np.arange(N) + 0.5) / N
    k, m, iv = 2, 3, 1
    h = np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
    psi = vec[:, iv].real[:, None, None] * h[None]
    lam = -(4 / D**2) * (np.sin(k * np.pi / (2 * N)) ** 2 + np.sin(m * np.pi / (2 * N)) ** 2)
    q = (lam + gam[iv].real) * psi
    p = o.pyq2p(q)
    st = o.mgstats()
    assert st.resa <= 1e-13 and st.i < 40
    assert np.abs(p - psi).max() <= 1e-10 * np.abs(psi).max()


def dense_operator(o, N):
    """Dense matrix of lap + Gamma with Dirichlet-at-face walls, unknown index (l, j, i)."""
    nl = o.nl
    D = o.param("L0") / N
    G = gamma_matrix(o)
    n = nl * N * N
    A = np.zeros((n, n))

    def idx(l, j, i):
        return (l * N + j) * N + i

    for l in range(nl):
        for j in range(N):
            for i in range(N):
                r = idx(l, j, i)
                A[r, r] += -4 / D**2
                for (di, dj) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    ii, jj = i + di, j + dj
                    if 0 <= ii < N and 0 <= jj < N:
                        A[r, idx(l, jj, ii)] += 1 / D**2
                    else:
                        A[r, r] += -1 / D**2  # ghost = -interior
                for l2 in range(nl):
                    A[r, idx(l2, j, i)] += G[l, l2]
    return A


@pytest.mark.parametrize("smoother", [orc.GS_LEX, orc.GS_RB])
def test_multigrid_matches_dense_solve(smoother):
    N, nl = 8, 3
    o = orc.Oracle(params(N, nl), smoother=smoother, TOLERANCE=1e-13, quiet=1)
    o.set_const()
    A = dense_operator(o, N)
    rng = np.random.default_rng(1)
    q = rng.standard_normal((nl, N, N))
    ref = np.linalg.solve(A, q.ravel()).reshape(nl, N, N)
    p = o.pyq2p(q)
    assert np.abs(p - ref).max() <= 1e-9 * np.abs(ref).max()
    # residual_layer is b - A a
    res, m = o.residual(ref, q)
    assert m <= 1e-10 * np.abs(q).max()
    a = rng.standard_normal((nl, N, N))
    res, m = o.residual(a, q)
    assert np.abs(res - (q - (A @ a.ravel()).reshape(nl, N, N))).max() <= 1e-12 * np.abs(res).max()
    assert m == pytest.approx(np.abs(res).max(), rel=1e-15)


@pytest.mark.parametrize("smoother", [orc.GS_LEX, orc.GS_RB])
def test_relaxation_fixed_point_is_dense_solution(smoother):
    """relax_layer sweeps (Thomas in the vertical) converge to the solution of A a = b."""
    N, nl = 8, 3
    o = orc.Oracle(params(N, nl), smoother=smoother)
    o.set_const()
    A = dense_operator(o, N)
    rng = np.random.default_rng(2)
    b = rng.standard_normal((nl, N, N))
    ref = np.linalg.solve(A, b.ravel()).reshape(nl, N, N)
    a = o.relax(0, np.zeros_like(b), b, nsweeps=800)
    assert np.abs(a - ref).max() <= 1e-10 * np.abs(ref).max()
    # one sweep from the exact solution leaves it unchanged
    a1 = o.relax(0, ref, b, nsweeps=1)
    assert np.abs(a1 - ref).max() <= 1e-12 * np.abs(ref).max()


def test_red_black_sweep_is_two_jacobi_half_sweeps():
    """Definition check of the RB ordering used by the GPU: red = (i+j) even first."""
    N, nl = 8, 2
    o = orc.Oracle(params(N, nl), smoother=orc.GS_RB)
    o.set_const()
    A = dense_operator(o, N)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((nl, N, N))
    a0 = rng.standard_normal((nl, N, N))
    a1 = o.relax(0, a0, b, nsweeps=1)
    # independent restatement: for each colour, solve the column system with the other
    # colour frozen; the wall ghost uses the cell's own value before the sweep (lagged ghost).
    D = o.param("L0") / N
    G = gamma_matrix(o)
    a = a0.copy()
    for c in (0, 1):
        new = a.copy()
        for j in range(N):
            for i in range(N):
                if (i + j) % 2 != c:
                    continue
                nb = np.zeros(nl)
                for (di, dj) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    ii, jj = i + di, j + dj
                    nb += a[:, jj, ii] if (0 <= ii < N and 0 <= jj < N) else -a0[:, j, i]
                M = G - 4 / D**2 * np.eye(nl)
                new[:, j, i] = np.linalg.solve(M, b[:, j, i] - nb / D**2)
        a = new
    assert np.abs(a1 - a).max() <= 1e-12 * np.abs(a).max()


def test_arakawa_identities_periodic():
    """sum J = sum psi J = sum zeta J = 0 on a periodic box (Arakawa 1966), qg.h:252-262."""
    N, nl = 32, 2
    txt = f"N = {N}\nnl = {nl}\nL0 = 1\nRom = 1\nbeta = 0\nsbc = -1\nFr = [0]\ndh = [0.5,0.5]\n"
    o = orc.Oracle(txt)
    o.set_const()
    rng = np.random.default_rng(4)
    psi = rng.standard_normal((nl, N, N))
    zeta = rng.standard_normal((nl, N, N))
    o.set(orc.PSI, psi)
    o.set(orc.ZETA, zeta)
    o.set(orc.DQ, np.zeros_like(psi))
    o.advection_pv(orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1.0)
    J = o.get(orc.DQ)
    scale = np.abs(J).sum()
    assert abs(J.sum()) <= 1e-13 * scale
    assert abs((J * psi).sum()) <= 1e-13 * scale * np.abs(psi).max()
    assert abs((J * zeta).sum()) <= 1e-13 * scale * np.abs(zeta).max()
    # antisymmetry J(a,b) = -J(b,a)
    o.set(orc.PSI, zeta)
    o.set(orc.ZETA, psi)
    o.set(orc.DQ, np.zeros_like(psi))
    o.advection_pv(orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1.0)
    assert np.abs(o.get(orc.DQ) + J).max() <= 1e-12 * np.abs(J).max()


def test_jacobian_analytic_second_order():
    """-J(psi, zeta) of smooth fields converges at 2nd order to psi_y zeta_x - psi_x zeta_y."""
    errs = []
    for N in (32, 64):
        txt = f"N = {N}\nnl = 2\nL0 = 1\nRom = 1\nbeta = 0\nsbc = -1\nFr = [0]\ndh = [0.5,0.5]\n"
        o = orc.Oracle(txt)
        o.set_const()
        x = (np.arange(N) + 0.5) / N
        X, Y = np.meshgrid(x, x)  # [y][x]
        psi = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
        zeta = np.cos(4 * np.pi * X) * np.sin(2 * np.pi * Y)
        px = 2 * np.pi * np.cos(2 * np.pi * X) * np.cos(2 * np.pi * Y)
        py = -2 * np.pi * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y)
        zx = -4 * np.pi * np.sin(4 * np.pi * X) * np.sin(2 * np.pi * Y)
        zy = 2 * np.pi * np.cos(4 * np.pi * X) * np.cos(2 * np.pi * Y)
        exact = -(px * zy - py * zx)
        o.set(orc.PSI, np.stack([psi, psi]))
        o.set(orc.ZETA, np.stack([zeta, zeta]))
        o.set(orc.DQ, np.zeros((2, N, N)))
        o.advection_pv(orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1.0)
        errs.append(np.abs(o.get(orc.DQ)[0] - exact).max())
    assert errs[1] < errs[0] / 3.5


def test_beta_and_stretch_coupling_terms():
    """dq_l = -J(psi_l, zeta_l) - beta v_l + S-coupled cross-layer Jacobians (qg.h:314-369)."""
    N, nl = 16, 3
    o = orc.Oracle(params(N, nl, sbc=-1))
    o.set_const()
    rng = np.random.default_rng(5)
    psi = rng.standard_normal((nl, N, N))
    z = np.zeros_like(psi)
    D = 80.0 / N
    o.set(orc.PSI, psi)
    o.set(orc.ZETA, z)
    o.set(orc.DQ, z)
    o.advection_pv(orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1.0)
    got = o.get(orc.DQ)

    def mjac(p, q):
        r = lambda a, dx, dy: np.roll(np.roll(a, -dx, axis=1), -dy, axis=0)
        P = lambda dx, dy: r(p, dx, dy)
        Q = lambda dx, dy: r(q, dx, dy)
        return ((Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0))
                + Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1)) - Q(0, 1) * (P(1, 1) - P(-1, 1))
                + Q(0, -1) * (P(1, -1) - P(-1, -1)) + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1))
                - P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1))) / (12 * D * D)

    S = [(o.param(f"Fr_{l}") / 0.025) ** 2 for l in range(nl - 1)]
    exp = np.zeros_like(psi)
    jd = [mjac(psi[l], psi[l + 1]) for l in range(nl - 1)]
    for l in range(nl):
        exp[l] = 0.5 * (np.roll(psi[l], 1, axis=1) - np.roll(psi[l], -1, axis=1)) / (2 * D)
        if l > 0:
            exp[l] += S[l - 1] * (-jd[l - 1]) * o.param(f"idh0_{l}")
        if l < nl - 1:
            exp[l] += S[l] * jd[l] * o.param(f"idh1_{l}")
    assert np.abs(got - exp).max() <= 1e-12 * np.abs(exp).max()


def test_dt_limiter_sequence():
    """timestep(): d = CFL*min(dtmax/CFL, dtmin); if d > previous: d = (previous + 0.1 d)/1.1."""
    o = orc.Oracle(params(16, 2, CFL=0.6))
    prev = 0.0
    for dtmin, dtmax in [(1.0, 0.05), (0.02, 0.05), (0.5, 0.05), (0.5, 0.05), (1e-3, 0.05), (10.0, 10.0)]:
        d = min(dtmax / 0.6, dtmin) * 0.6
        if d > prev:
            d = (prev + 0.1 * d) / 1.1
        prev = d
        assert o.limiter(dtmin, dtmax) == pytest.approx(d, rel=1e-15)


def test_cfl_dt_from_uniform_shear():
    """psi = -U y  ->  u = U on x-faces; dt = CFL*D/|U| before smoothing (qg.h:276-283,383-391)."""
    N, nl = 16, 2
    o = orc.Oracle(params(N, nl, sbc=-1, CFL=0.5))
    o.set_const()
    D = 80.0 / N
    y = (np.arange(N) + 0.5) * D
    U = 3.0
    psi = np.repeat((-U * y)[None, :, None], N, axis=2).repeat(nl, axis=0)
    # periodic wrap makes a jump at the seam; use a sine in y instead: u = -dpsi/dy
    psi = np.repeat((np.sin(2 * np.pi * y / 80.0))[None, :, None], N, axis=2).repeat(nl, axis=0)
    o.set(orc.PSI, psi)
    o.set(orc.ZETA, np.zeros_like(psi))
    o.set(orc.DQ, np.zeros_like(psi))
    o.reset_limiter()
    dt = o.advection_pv(orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1e10)
    # face velocity u = -(psi[j+1]-psi[j-1])/(2D); max |u| over faces
    umax = np.abs((np.roll(psi[0, :, 0], -1) - np.roll(psi[0, :, 0], 1)) / (2 * D)).max()
    d = 0.5 * D / umax
    # 2*nl limiter calls, the first from previous = 0: d1 = 0.1 d/1.1, then relaxing up
    prev = 0.0
    cur = 1e10
    for call in range(2 * nl):
        dm = D / umax if call % 2 == 0 else np.inf  # psipg = 0 -> no constraint
        cur = min(cur / 0.5, dm) * 0.5
        if cur > prev:
            cur = (prev + 0.1 * cur) / 1.1
        prev = cur
    assert dt == pytest.approx(cur, rel=1e-13)
    assert dt < d


def test_restriction_prolongation_rules():
    N, nl = 16, 2
    o = orc.Oracle(params(N, nl))
    o.set_const()
    rng = np.random.default_rng(6)
    f = rng.standard_normal((nl, N, N))
    c = o.restrict(0, f)
    exp = 0.25 * (f[:, 0::2, 0::2] + f[:, 1::2, 0::2] + f[:, 0::2, 1::2] + f[:, 1::2, 1::2])
    assert np.abs(c - exp).max() <= 1e-15
    # bilinear prolongation of a field that is linear in x and y is exact away from the walls
    xc = (np.arange(N // 2) + 0.5) * 2
    lin = 0.3 * xc[None, None, :] + 0.7 * xc[None, :, None] + np.zeros((nl, 1, 1))
    fine = o.prolong(1, lin)
    xf = np.arange(N) + 0.5
    expf = 0.3 * xf[None, None, :] + 0.7 * xf[None, :, None] + np.zeros((nl, 1, 1))
    assert np.abs(fine[:, 1:-1, 1:-1] - expf[:, 1:-1, 1:-1]).max() <= 1e-13
    # at the wall the coarse ghost is -interior (homogeneous Dirichlet on the correction),
    # corner ghost = +interior (y-BC applied to the x-ghost)
    cc = rng.standard_normal((nl, N // 2, N // 2))
    ff = o.prolong(1, cc)
    assert ff[0, 0, 0] == pytest.approx((9 * cc[0, 0, 0] - 3 * cc[0, 0, 0] - 3 * cc[0, 0, 0] + cc[0, 0, 0]) / 16, rel=1e-14)
    assert ff[0, 0, 3] == pytest.approx((9 * cc[0, 0, 1] + 3 * cc[0, 0, 2] - 3 * cc[0, 0, 1] - cc[0, 0, 2]) / 16, rel=1e-14)


def test_mg_cycle_stats_and_warm_start():
    N, nl = 64, 3
    o = orc.Oracle(orc.double_gyre_params(N, nl), smoother=orc.GS_LEX)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    o.set_const()
    q = o.get(orc.Q)
    psi0 = o.get(orc.PSI)
    # cold start needs more than one cycle at a tight tolerance, converges monotonically
    o.option("TOLERANCE", 1e-10)
    o.set(orc.PSI, np.zeros_like(q))
    st = o.invertq()
    assert 1 < st.i < 30 and st.resa <= 1e-10 < st.resb
    assert np.abs(o.get(orc.PSI) - psi0).max() <= 1e-5 * np.abs(psi0).max()  # |A^-1| ~ (L0/pi)^2
    # warm start at reference tolerance: NITERMIN = 1 cycle is always done
    o.option("TOLERANCE", 1e-3)
    st = o.invertq()
    assert st.i == 1 and st.nrelax == 4


def test_lex_and_rb_agree_at_tight_tolerance():
    N, nl = 32, 3
    res = []
    for sm in (orc.GS_LEX, orc.GS_RB):
        o = orc.Oracle(orc.double_gyre_params(N, nl), smoother=sm, TOLERANCE=1e-13, quiet=1)
        o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
        o.set_const()
        q = o.get(orc.Q)
        res.append(o.pyq2p(q))
    assert np.abs(res[0] - res[1]).max() <= 1e-7 * np.abs(res[0]).max()  # tol 1e-13 x |A^-1| ~ (L0/pi)^2


def test_step_sequence_and_dtnext():
    N, nl = 32, 3
    o = orc.Oracle(orc.double_gyre_params(N, nl), smoother=orc.GS_RB)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    o.set_const()
    o.set_tnext(1.0)
    DT = o.param("DT")
    q0 = o.get(orc.Q)
    dts = []
    for _ in range(3):
        o.step()
        dts.append(o.dt)
    # limiter ramps dt up from previous = 0: dt never exceeds DT, increases monotonically
    assert all(0 < d <= DT for d in dts) and dts[0] < dts[1] < dts[2]
    assert o.t == pytest.approx(sum(dts), rel=1e-14)
    assert o.iter == 3
    q = o.get(orc.Q)
    assert np.isfinite(q).all() and np.abs(q - q0).max() > 0
    # dtnext lands exactly on the event time
    o.set_tnext(o.t + 0.3 * dts[-1])
    t_target = o.t + 0.3 * dts[-1]
    o.step()
    assert o.t == pytest.approx(t_target, rel=1e-15)


def test_bas_roundtrip_matches_reference_reader(tmp_path):
    """Layout read back exactly as msqg/scripts/read_data.py:44-46 does."""
    N, nl = 16, 3
    o = orc.Oracle(params(N, nl))
    rng = np.random.default_rng(7)
    f = rng.standard_normal((nl, N, N)).astype(np.float32).astype(np.float64)
    o.set(orc.PSI, f)
    path = str(tmp_path / "po.bas")
    assert o.write_bas(orc.PSI, path) == 0
    raw = np.fromfile(path, "f4")
    assert raw.size == nl * (N + 1) ** 2 and int(raw[0]) == N
    p = raw.reshape(nl, N + 1, N + 1).transpose(0, 2, 1)[:, 1:, 1:]
    assert np.array_equal(p.astype(np.float64), f)
    D = 80.0 / N
    assert np.allclose(raw[1:N + 1], (np.arange(N) + 0.5) * D, rtol=1e-6)
    o2 = orc.Oracle(params(N, nl))
    assert o2.read_bas(orc.PSI, path) == 0
    assert np.array_equal(o2.get(orc.PSI), f)
    # different resolution on file: nearest-cell sampling (auxiliar_input.h:44-47)
    o3 = orc.Oracle(params(2 * N, nl))
    assert o3.read_bas(orc.PSI, path) == 0
    assert np.array_equal(o3.get(orc.PSI), np.repeat(np.repeat(f, 2, axis=1), 2, axis=2))


def test_pystep_bfn_direction_flips_dissipation_sign():
    """msqg/qg_bfn.h:34-44: backward integration flips iRe, iRe4, Eks, Ekb."""
    N, nl = 32, 3
    txt = orc.double_gyre_params(N, nl)
    o = orc.Oracle(txt, smoother=orc.GS_RB, TOLERANCE=1e-12, quiet=1)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    o.set_const()
    q = o.get(orc.Q)
    f = o.pystep_bfn(q, +1.0)
    b = o.pystep_bfn(q, -1.0)
    # the non-dissipative part is common, the dissipative part changes sign
    o2 = orc.Oracle(txt.replace("Ekb   = 0.002", "Ekb = 0").replace(f"Re4   = {1563.0 * (N / 256.0) ** 4}", "Re4 = 0"),
                    smoother=orc.GS_RB, TOLERANCE=1e-12, quiet=1)
    o2.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    o2.set_const()
    inviscid = o2.pystep_bfn(q, +1.0)
    assert np.abs(0.5 * (f + b) - inviscid).max() <= 1e-9 * np.abs(inviscid).max()
    assert np.abs(f - b).max() > 0
