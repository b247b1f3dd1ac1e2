"""Known-answer tests of the vertex-grid oracle (oracle/qgnode_oracle.c, restating qg-node/):
eigenfunctions of the nodal 5-point Laplacian with psi = 0 ON the boundary vertices, dense
solves of the masked layered elliptic problem, transfer operators of qg-node/my_vertex.h,
wall boundary values of q / zeta (free slip ... no slip), dt limiter."""
import numpy as np
import pytest

import orn


def gamma_matrix(o, f=46.5):
    nl = o.nl
    _, N2s = orn.NODE_LAYERS[nl]
    N2 = [float(v) for v in N2s.strip("[]").split(",")]
    S = [f * f / n for n in N2[:nl - 1]]
    G = np.zeros((nl, nl))
    for l in range(nl):
        if l > 0:
            c = S[l - 1] * o.param(f"idh0_{l}"); G[l, l - 1] += c; G[l, l] -= c
        if l < nl - 1:
            c = S[l] * o.param(f"idh1_{l}"); G[l, l + 1] += c; G[l, l] -= c
    return G


@pytest.mark.parametrize("nl", [2, 3])
def test_nodal_eigenfunctions_and_inverse(nl):
    N = 32
    o = orn.NodeOracle(orn.node_params(N, nl), smoother=orn.GS_LEX, TOLERANCE=1e-11, quiet=1)
    o.set_const()
    G = gamma_matrix(o)
    gam, vec = np.linalg.eig(G)
    D = 100.0 / N
    x = np.arange(N + 1) / N
    k, m, iv = 3, 2, 1
    h = np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
    psi = vec[:, iv].real[:, None, None] * h[None]
    lam = -(4 / D**2) * (np.sin(k * np.pi / (2 * N)) ** 2 + np.sin(m * np.pi / (2 * N)) ** 2)
    o.set(orn.PSI, psi)
    o.comp_q()
    q = o.get(orn.Q)
    inner = (slice(None), slice(1, -1), slice(1, -1))
    assert np.abs(q[inner] - ((lam + gam[iv].real) * psi)[inner]).max() <= 1e-11 * np.abs(q).max()
    assert np.all(q[:, 0, :] == 0) and np.all(q[:, :, -1] == 0)       # bc_fac = 0: free slip, q = 0 on the walls
    o.set(orn.PSI, np.zeros_like(psi))
    st = o.invert_q()
    assert st.resa < 1e-11 and st.i < 40
    assert np.abs(o.get(orn.PSI) - psi).max() <= 1e-9 * np.abs(psi).max()


def test_no_slip_wall_vorticity():
    """q, zeta on a boundary vertex = 2 bc_fac / D^2 (psi_first_interior - 0), qg-node/qg.h:206-214"""
    N, nl = 16, 2
    o = orn.NodeOracle(orn.node_params(N, nl, bc_fac=1.0))
    o.set_const()
    psi = orn.node_psi(nl, N)
    o.set(orn.PSI, psi)
    o.comp_q()
    q = o.get(orn.Q)
    D = 100.0 / N
    assert np.allclose(q[:, 1:-1, 0], 2 / D**2 * psi[:, 1:-1, 1], rtol=1e-14)
    assert np.allclose(q[:, 1:-1, -1], 2 / D**2 * psi[:, 1:-1, -2], rtol=1e-14)
    assert np.allclose(q[:, 0, 1:-1], 2 / D**2 * psi[:, 1, 1:-1], rtol=1e-14)
    assert np.allclose(q[:, -1, 1:-1], 2 / D**2 * psi[:, -2, 1:-1], rtol=1e-14)


def dense(o, N, nl, mask=None):
    D = 100.0 / N
    G = gamma_matrix(o) if nl > 1 else np.zeros((1, 1))
    n1 = N - 1
    A = np.zeros((nl * n1 * n1, nl * n1 * n1))
    idx = lambda l, j, i: (l * n1 + (j - 1)) * n1 + (i - 1)
    for l in range(nl):
        for j in range(1, N):
            for i in range(1, N):
                r = idx(l, j, i)
                A[r, r] -= 4 / D**2
                for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    ii, jj = i + di, j + dj
                    if 1 <= ii < N and 1 <= jj < N:
                        A[r, idx(l, jj, ii)] += 1 / D**2
                for l2 in range(nl):
                    A[r, idx(l2, j, i)] += G[l, l2]
    return A


@pytest.mark.parametrize("smoother", [orn.GS_LEX, orn.GS_RB])
@pytest.mark.parametrize("nl", [1, 3])
def test_vertex_multigrid_matches_dense_solve(smoother, nl):
    N = 8
    o = orn.NodeOracle(orn.node_params(N, nl), smoother=smoother, TOLERANCE=1e-12, quiet=1)
    o.set_const()
    A = dense(o, N, nl)
    rng = np.random.default_rng(0)
    q = np.zeros((nl, N + 1, N + 1))
    q[:, 1:-1, 1:-1] = rng.standard_normal((nl, N - 1, N - 1))
    ref = np.linalg.solve(A, q[:, 1:-1, 1:-1].ravel()).reshape(nl, N - 1, N - 1)
    o.set(orn.Q, q)
    st = o.invert_q()
    assert st.resa < 1e-12
    assert np.abs(o.get(orn.PSI)[:, 1:-1, 1:-1] - ref).max() <= 1e-9 * np.abs(ref).max()
    a = np.zeros_like(q); a[:, 1:-1, 1:-1] = rng.standard_normal((nl, N - 1, N - 1))
    res, m = o.residual(a, q)
    exp = q[:, 1:-1, 1:-1] - (A @ a[:, 1:-1, 1:-1].ravel()).reshape(nl, N - 1, N - 1)
    assert np.abs(res[:, 1:-1, 1:-1] - exp).max() <= 1e-12 * np.abs(exp).max()
    assert np.all(res[:, 0, :] == 0) and m == pytest.approx(np.abs(res).max())
    # relaxation fixed point
    x = o.relax(0, np.zeros_like(q), q, 600)
    assert np.abs(x[:, 1:-1, 1:-1] - ref).max() <= 1e-9 * np.abs(ref).max()


def test_vertex_transfer_operators():
    N, nl = 16, 2
    o = orn.NodeOracle(orn.node_params(N, nl))
    o.set_const()
    rng = np.random.default_rng(1)
    f = rng.standard_normal((nl, N + 1, N + 1))
    c = o.restrict(0, f)
    f0 = f.copy(); f0[:, 0, :] = f0[:, -1, :] = 0; f0[:, :, 0] = f0[:, :, -1] = 0     # boundary(res) = 0 first
    fi = 2 * np.arange(1, N // 2)
    J, I = np.meshgrid(fi, fi, indexing="ij")
    exp = (f0[:, J, I + 1] + 2 * f0[:, J, I] + f0[:, J, I - 1] + f0[:, J + 1, I] + f0[:, J - 1, I]) / 6
    assert np.allclose(c[:, 1:-1, 1:-1], exp, rtol=1e-14) and np.all(c[:, 0, :] == 0)
    cc = rng.standard_normal((nl, N // 2 + 1, N // 2 + 1))
    cc[:, 0, :] = cc[:, -1, :] = 0; cc[:, :, 0] = cc[:, :, -1] = 0
    ff = o.prolong(1, cc)
    assert np.array_equal(ff[:, 2:-1:2, 2:-1:2], cc[:, 1:-1, 1:-1])                   # injection
    assert np.allclose(ff[:, 2:-1:2, 1::2], 0.5 * (cc[:, 1:-1, :-1] + cc[:, 1:-1, 1:]))  # edge midpoints
    assert np.allclose(ff[:, 1::2, 1::2], 0.25 * (cc[:, :-1, :-1] + cc[:, :-1, 1:] + cc[:, 1:, :-1] + cc[:, 1:, 1:]))
    assert np.all(ff[:, 0, :] == 0)
    # level masks: 9-point full weighting, then 0 on the walls
    m1 = o.level_mask(1)
    assert np.all(m1[0, 1:-1, 1:-1] == 1) and np.all(m1[0, 0, :] == 0)


def test_time_steps_and_limiter():
    N, nl = 32, 3
    o = orn.NodeOracle(orn.node_params(N, nl, bc_fac=0.5, nu4=1.0))
    o.set(orn.PSI, orn.node_psi(nl, N))
    o.set_const()
    D = 100.0 / N
    DT = o.param("DT")
    assert DT == pytest.approx(min(0.5 * min(5e-2, D * D / 5 / 4), 1 / (2 * 0.5 * 100)))   # qg-node/qg.h:511-512
    o.set_tnext(float("inf"))
    dts = []
    for _ in range(3):
        o.step()
        dts.append(o.dt)
    assert all(0 < d <= DT for d in dts) and dts[0] < dts[1] < dts[2]
    q = o.get(orn.Q)
    assert np.isfinite(q).all() and np.all(q[:, 1:-1, 1:-1] != 0)
    assert o.ke() > 0


def test_stochastic_forcing_pieces():
    """qg-node/qg_stochastic.h + qg.h:306-320: wavelet coefficients of the uniform filter length, all-pass /
    all-stop limits of the filtered noise, sqrt(dt) weights of the predictor / corrector advance"""
    N = 32                                   # L0 = 100: Delta_k = 3.125 * 2^k
    o = orn.NodeOracle(orn.node_params(N, 1, extra="amp_stoch = 0.5\nL_filt = 1e6\n"), stochastic=1, seed=7)
    o.set_const()
    K = o.cell_levels()
    assert K == 6 and all(np.all(o.csig(k) == 1) for k in range(K))        # low pass 0 everywhere -> high pass 1
    rng = np.random.default_rng(0)
    n0 = rng.standard_normal((N, N))
    o.set_noise(n0); o.filter_noise()
    assert np.abs(o.noise() - n0).max() <= 4e-16 * np.abs(n0).max() * 4
    o2 = orn.NodeOracle(orn.node_params(N, 1, extra="amp_stoch = 0.5\nL_filt = 0.1\n"), stochastic=1)
    o2.set_const()
    assert all(np.all(o2.csig(k) == 0) for k in range(K))
    o2.set_noise(n0); o2.filter_noise()
    assert np.all(o2.noise() == 0)
    o3 = orn.NodeOracle(orn.node_params(N, 1, extra="amp_stoch = 0.5\nL_filt = 8\n"), stochastic=1)
    o3.set_const()                           # Delta_1 = 6.25 < 8 <= 12.5: low pass 1 - (8 - 6.25)/6.25 = 0.72 on level 1
    assert np.all(o3.csig(0) == 1) and np.allclose(o3.csig(1), 0.28, rtol=1e-14) and np.all(o3.csig(2) == 0)
    # advance: predictor draws the noise and adds n sqrt(dt/2) / ... , corrector re-uses it with sqrt(dt)
    z = np.zeros((1, N + 1, N + 1))
    o.set(orn.Q, z); o.set(orn.DQ, z)
    dt = 0.04
    o.advance(orn.QPRED, orn.Q, orn.DQ, dt / 2)
    n = o.noise()
    assert 0.3 < n.std() / 0.5 < 1.7 and abs(n.mean()) < 0.2             # amp_stoch * N(0,1), all-pass filter
    qp = o.get(orn.QPRED)[0]
    assert np.allclose(qp[:N, :N], n * np.sqrt(dt / 2) / np.sqrt(2), rtol=1e-14)
    assert np.array_equal(qp[N, :N], qp[N - 1, :N]) and np.array_equal(qp[:N, N], qp[:N, N - 1])   # ghost cells of the cell field
    o.advance(orn.Q, orn.Q, orn.DQ, dt)
    assert np.array_equal(o.noise(), n)
    assert np.allclose(o.get(orn.Q)[0][:N, :N], n * np.sqrt(dt), rtol=1e-14)
