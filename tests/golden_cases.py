"""Golden-vector cases of the hot path, shared by the generator (tools/make_golden.py), the CPU test that pins the oracle
to the committed files (tests/test_golden_oracle.py) and the GPU tests that compare the HIP library with the same files
(tests/test_gpu_golden.py).

A case is a function `case(make, inputs)` -> dict of named results.  `make(params_text, **options)` builds a model behind
one of two adapters with the same method names: `OracleModel` (oracle/qg_oracle.c through tests/orc.py) or `GpuModel`
(libmsomhip through the C ABI, msom_amd.QG).  `inputs` are the arrays stored in the fixture (`in_*` keys): the generator
creates them, the tests read them back from the file, so no test depends on a random-number stream.

What the files pin: the per-operator vectors of msqg/qg.h:172-488 and
msqg/poisson_layer.h:48-258, the elliptic inversion incl. cycle counts, the 10-step double-gyre run of msqg/qg.c:53-109
from a fixed float32 `p0.bas`, BASELINE configs C1 (128^2 x 1) and C2 (512^2 x 3) at their stated sizes, the stochastic
srand(7) run of msqg/qg_stochastic.h, passive tracers, the wavelet filter, and the vertex-grid variant with an island.
The reference ships no vectors of its own and cannot be built here: the files are produced by the CPU restatement
(PARITY UNPINNED against the reference itself, see oracle/qg_oracle.h); what they buy is that oracle and kernels can no
longer move together unnoticed between rounds.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, ROOT)
from msom_amd import workloads as wl  # noqa: E402

NAMES = ["PSI", "Q", "ZETA", "PSIPG", "ZETAPG", "QFORC", "TMP", "FR", "S", "DQ", "RO", "TOPO", "QPRED", "NOISE", "SIGMA",
         "PTR", "PTR_RELAX", "DPTR", "PTR_PRED", "RD", "QOF"]


def _stats(st):
    return np.array([st.i, st.resb, st.resa, st.sum, st.nrelax], dtype=np.float64)


class OracleModel:
    """adapter over tests/orc.py (red-black smoother unless smoother=0 is asked for)"""
    kind = "oracle"

    def __init__(self, txt, smoother=1, **opts):
        import orc
        self.orc = orc
        self.m = orc.Oracle(txt, smoother=smoother, quiet=1)
        for k, v in opts.items():
            self.m.option(k, v)
        self.nx, self.ny, self.nl = self.m.nx, self.m.ny, self.m.nl

    def fid(self, name):
        return getattr(self.orc, name)

    def option(self, k, v): self.m.option(k, v)
    def param(self, k): return self.m.param(k)
    def set(self, name, a): self.m.set(self.fid(name), a)
    def get(self, name): return self.m.get(self.fid(name))
    def set_const(self): self.m.set_const()
    def pyp2q(self, psi): return self.m.pyp2q(psi)

    def pyq2p(self, q):
        p = self.m.pyq2p(q)
        return p, _stats(self.m.mgstats())

    def del2(self, fin, fout, add, fac): self.m.comp_del2(self.fid(fin), self.fid(fout), add, fac)
    def stretch(self, fin, fout, add, fac): self.m.comp_stretch(self.fid(fin), self.fid(fout), add, fac)
    def advection(self, zeta, dq): self.m.advection_pv(self.fid(zeta), self.orc.Q, self.orc.PSI, self.fid(dq), 1e10)

    def update(self):
        d = self.m.update()
        return self.m.get(self.orc.DQ), d

    def pystep_bfn(self, q, direction): return self.m.pystep_bfn(q, direction)
    def nlevels(self): return self.m.nlevels()
    def level_dims(self, lev): return self.m.level_dims(lev)
    def relax(self, lev, da, res, ns): return self.m.relax(lev, da, res, ns)
    def residual(self, a, b): return self.m.residual(a, b)
    def restrict(self, lev, f): return self.m.restrict(lev, f)
    def prolong(self, lev, c): return self.m.prolong(lev, c)

    def step(self):
        self.m.step()
        return self.m.dt

    def set_tnext(self, t): self.m.set_tnext(t)
    @property
    def t(self): return self.m.t
    def ke(self): return self.m.ke()
    def mgstats(self): return _stats(self.m.mgstats())
    def wavelet_filter(self, dtflt): self.m.wavelet_filter(dtflt)
    def read_bas(self, name, path): assert self.m.read_bas(self.fid(name), path) == 0
    def close(self): self.m = None


class GpuModel:
    """adapter over msom_amd.QG (the C ABI of include/msom.h); strict selects libmsomhip_strict.so"""
    kind = "gpu"

    def __init__(self, txt, strict=True, **opts):
        from msom_amd import FIELDS, QG
        self.F = FIELDS
        self.m = QG(txt, strict=strict)
        self.m.option("quiet", 1)
        for k, v in opts.items():
            self.m.option(k, v)
        self.nx, self.ny, self.nl = self.m.nx, self.m.ny, self.m.nl

    def fid(self, name): return self.F[name]
    def option(self, k, v): self.m.option(k, v)
    def param(self, k): return self.m.param(k)
    def set(self, name, a): self.m.set(self.fid(name), a)
    def get(self, name): return self.m.get(self.fid(name))
    def set_const(self): self.m.set_const()

    def pyp2q(self, psi):
        q = np.empty_like(psi)
        self.m.pyp2q(psi, q)
        return q

    def pyq2p(self, q):
        p = np.empty_like(q)
        self.m.pyq2p(p, q)
        return p, _stats(self.m.mgstats())

    def del2(self, fin, fout, add, fac): self.m.op("del2", self.fid(fin), self.fid(fout), add, fac)
    def stretch(self, fin, fout, add, fac): self.m.op("stretch", self.fid(fin), self.fid(fout), add, fac)
    def advection(self, zeta, dq): self.m.op("advection", self.fid(zeta), self.fid(dq))
    def update(self): return self.m.update(want=True)

    def pystep_bfn(self, q, direction):
        tend = np.empty_like(q)
        self.m.pystep_bfn(q, tend, direction, 1)
        return tend

    def nlevels(self): return self.m.nlevels()
    def level_dims(self, lev): return self.m.level_dims(lev)
    def relax(self, lev, da, res, ns): return self.m.relax(lev, da, res, ns)
    def residual(self, a, b): return self.m.residual(a, b)
    def restrict(self, lev, f): return self.m.restrict(lev, f)
    def prolong(self, lev, c): return self.m.prolong(lev, c)
    def step(self): return self.m.step()
    def set_tnext(self, t): self.m.set_tnext(t)
    @property
    def t(self): return self.m.t
    def ke(self): return self.m.ke()
    def mgstats(self): return _stats(self.m.mgstats())
    def wavelet_filter(self, dtflt): self.m.wavelet_filter(dtflt)
    def read_bas(self, name, path): self.m.read_bas(self.fid(name), path)
    def close(self): self.m.close()


# ---------------------------------------------------------------------------------------------------------------------
# input builders (generator side only: their results are stored in the fixture as in_*)

def _rng_inputs(seed, shapes):
    rng = np.random.default_rng(seed)
    return {k: s * rng.standard_normal(shape) for k, (shape, s) in shapes.items()}


def ops_inputs(nx, ny, nl, seed):
    sh = (nl, ny, nx)
    return _rng_inputs(seed, {"in_psi": (sh, 1.0), "in_zeta": (sh, 1.0), "in_dq": (sh, 0.1), "in_a": (sh, 1.0), "in_b": (sh, 1.0),
                              "in_qpert": (sh, 1e-6)})


# ---------------------------------------------------------------------------------------------------------------------
# cases

def case_ops(make, inp, nx, ny, nl, extra=""):
    """every operator of the path on stored random fields; msqg/qg.h:172-246 (del2, stretch), :288-393 (advection_pv),
    :397-403 (comp_q), :609-650 (update_qg), msqg/poisson_layer.h:48-258 (relax, residual), Basilisk restriction /
    bilinear prolongation, msqg/qg.h:114-163 (invertq), msqg/qg_bfn.h:21-103 (pystep_bfn)"""
    out = {}
    txt = wl.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + extra)
    m = make(txt)
    m.set("PSI", wl.synthetic_psi(nl, ny, nx))
    m.set_const()
    out["q_of_synthetic_psi"] = m.get("Q")
    if nl > 1:
        out["S"] = m.get("S")
    psi, zeta = inp["in_psi"], inp["in_zeta"]
    out["pyp2q"] = m.pyp2q(psi)
    m.set("PSI", psi)
    m.set("ZETA", psi[::-1].copy())
    m.del2("PSI", "ZETA", 0.5, -2.0)
    out["del2_add"] = m.get("ZETA")
    m.stretch("PSI", "ZETA", 1.0, 0.3)
    out["stretch_add"] = m.get("ZETA")
    m.set("ZETA", zeta)
    m.set("DQ", inp["in_dq"])
    m.advection("ZETA", "DQ")
    out["advection"] = m.get("DQ")
    res, mx = m.residual(inp["in_a"], inp["in_b"])
    out["residual"], out["residual_max"] = res, np.float64(mx)
    # level inputs are derived from the stored arrays by restriction
    da, rs = inp["in_a"], inp["in_b"]
    for lev in range(m.nlevels()):
        lx, ly = m.level_dims(lev)
        assert da.shape == (nl, ly, lx)
        out[f"relax1_l{lev}"] = m.relax(lev, da, rs, 1)
        out[f"relax3_l{lev}"] = m.relax(lev, da, rs, 3)
        if lev >= 1:
            out[f"prolong_l{lev}"] = m.prolong(lev, da)
        if lev + 1 < m.nlevels():
            da_c, rs_c = m.restrict(lev, da), m.restrict(lev, rs)
            out[f"restrict_l{lev}"] = rs_c
            da, rs = da_c, rs_c
    m.close()
    # elliptic inversion at the reference tolerance and a tight one: psi, (cycles, resb, resa, sum, nrelax)
    for tag, tol in (("tol1e-3", 1e-3), ("tol1e-11", 1e-11)):
        m = make(txt, TOLERANCE=tol)
        m.set("PSI", wl.synthetic_psi(nl, ny, nx))
        m.set_const()
        q = m.get("Q") + inp["in_qpert"]
        p, st = m.pyq2p(q)
        out[f"pyq2p_{tag}"], out[f"pyq2p_{tag}_stats"] = p, st
        if tol < 1e-6:
            dq, dt = m.update()
            out["update_dq"], out["update_dt"] = dq, np.float64(dt)
            for k, d in enumerate((1.0, -1.0, 1.0)):
                out[f"pystep_bfn_{k}"] = m.pystep_bfn(q, d)
        m.close()
    return out


def forcing_inputs(nx, ny, nl, seed):
    x = (np.arange(nx) + 0.5) / nx
    y = (np.arange(ny) + 0.5) / ny
    topo = 0.01 * np.exp(-((x[None, :] - 0.5) ** 2 + (y[:, None] - 0.5) ** 2) * 30)[None]
    d = _rng_inputs(seed, {"in_qforc": ((nl, ny, nx), 1e-7)})
    d["in_topo"] = topo
    return d


def case_forcing(make, inp, nx, ny, nl):
    """update_qg with every optional term on: partial slip (msqg/qg.h:185-198), Laplacian + biharmonic dissipation
    (:407-422), surface and bottom Ekman drag (:429-445), wind (:447-464), q forcing (:466-479), topography (:481-488),
    background flow with flsrv (:310-380)"""
    out = {}
    ex = "sbc = 2.0\nEks = 0.001\nRe = 500\nupg = [0.3,0.1,0.0]\nvpg = [0.0,-0.2,0.05]\nflsrv = 1\n"
    m = make(wl.double_gyre_params(nx, nl, extra=ex), TOLERANCE=1e-9)
    m.set("PSI", wl.synthetic_psi(nl, ny, nx))
    m.set("TOPO", inp["in_topo"])
    m.option("flag_topo", 1)
    m.set("QFORC", inp["in_qforc"])
    m.set_const()
    out["psipg"], out["zetapg"] = m.get("PSIPG"), m.get("ZETAPG")
    dq, dt = m.update()
    out["update_dq"], out["update_dt"] = dq, np.float64(dt)
    m.set_tnext(float("inf"))
    out["dt"] = np.array([m.step() for _ in range(3)])
    out["q_end"], out["psi_end"] = m.get("Q"), m.get("PSI")
    m.close()
    return out


def case_run(make, inp, nx, nl, nsteps=10, snap=(1, 5, 10), stride=1, extra="", p0bas=None, **opts):
    """the time loop of msqg/qg.c:53-109 + Basilisk run(): dt, t, KE line, mgstats of the last solve for every step;
    psi and q at the steps in `snap` (every `stride`-th point when the grid is large)"""
    m = make(wl.double_gyre_params(nx, nl, extra=extra), **opts)
    if p0bas is not None:
        m.read_bas("PSI", p0bas)          # float32 file, as `p0.bas` of msqg/qg.h:940-950
    else:
        m.set("PSI", wl.synthetic_psi(nl, nx, nx))
    m.set_const()
    m.set_tnext(1.0)                      # dtout = 1: the dtnext() shortening of the step is part of the sequence
    out = {"psi_0": m.get("PSI")[:, ::stride, ::stride], "q_0": m.get("Q")[:, ::stride, ::stride]}
    dts, ts, kes, sts = [], [], [], []
    for k in range(1, nsteps + 1):
        dts.append(m.step())
        ts.append(m.t)
        kes.append(m.ke())
        sts.append(m.mgstats())
        if k in snap:
            psi, q = m.get("PSI"), m.get("Q")
            out[f"psi_{k}"], out[f"q_{k}"] = psi[:, ::stride, ::stride], q[:, ::stride, ::stride]
            if stride > 1:                # whole-field functionals so that the un-sampled points are covered too
                out[f"psi_{k}_absmax"] = np.abs(psi).max(axis=(1, 2))
                out[f"q_{k}_absmax"] = np.abs(q).max(axis=(1, 2))
                out[f"psi_{k}_rowsum"] = psi.sum(axis=2)
                out[f"q_{k}_rowsum"] = q.sum(axis=2)
    out["dt"], out["t"], out["ke"], out["mgstats"] = np.array(dts), np.array(ts), np.array(kes), np.array(sts)
    m.close()
    return out


def case_stochastic(make, inp, nx, nl):
    """-D_STOCHASTIC variant (msqg/qg_stochastic.h:17-149) with the reference's serial rand() stream, srand(7)"""
    m = make(wl.double_gyre_params(nx, nl, extra="tr_stoch = 50\namp_stoch = 1e-5\n"), stochastic=1)
    m.set("PSI", wl.synthetic_psi(nl, nx, nx))
    m.set("SIGMA", inp["in_sigma"])
    m.set_const()
    ctypes.CDLL(None).srand(7)
    m.set_tnext(float("inf"))
    out = {"dt": np.array([m.step() for _ in range(3)])}
    out["q_end"], out["psi_end"], out["noise_end"] = m.get("Q"), m.get("PSI"), m.get("NOISE")
    m.close()
    return out


def case_tracers(make, inp, nx, nl, nptr):
    """passive tracers (msqg/qg.h:574-588, 634-647)"""
    ex = f"nptr = {nptr}\nptr_r = [{','.join(['10', '0', '3.5'][:nptr])}]\nPe = [{','.join(['200', '50', '0'][:nptr])}]\n"
    m = make(wl.double_gyre_params(nx, nl, extra=ex))
    m.set("PSI", wl.synthetic_psi(nl, nx, nx))
    m.set("PTR", inp["in_c0"])
    m.set("PTR_RELAX", inp["in_relax"])
    m.set_const()
    out = {}
    dq, dt = m.update()
    out["update_dq"], out["update_dptr"] = dq, m.get("DPTR")
    m.set_tnext(float("inf"))
    for _ in range(4):
        m.step()
    out["ptr_end"], out["q_end"] = m.get("PTR"), m.get("Q")
    m.close()
    return out


def case_wavelet(make, inp, N, nl):
    """wavelet scale filter (msqg/qg.h:509-560, coefficients :1059-1090), then two more steps"""
    m = make(wl.double_gyre_params(N, nl, extra="afilt = 4\n"), TOLERANCE=1e-11)
    m.set("RD", inp["in_rd"])
    m.set("PSI", wl.synthetic_psi(nl, N, N))
    m.set_const()
    m.wavelet_filter(0.5)
    out = {k.lower() + "_filtered": m.get(k) for k in ("PSI", "Q", "QOF")}
    for _ in range(2):
        m.step()
    out["q_end"] = m.get("Q")
    m.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# vertex-grid variant (qg-node): adapters + cases

class NodeOracleModel:
    kind = "oracle"

    def __init__(self, txt, **opts):
        import orn
        self.orn = orn
        self.m = orn.NodeOracle(txt, smoother=orn.GS_RB, quiet=1, **opts)
        self.N, self.nl = self.m.N, self.m.nl

    def set(self, name, a): self.m.set(getattr(self.orn, name), a)
    def get(self, name): return self.m.get(getattr(self.orn, name))
    def set_const(self): self.m.set_const()
    def nlevels(self): return self.m.nlevels()
    def relax(self, k, da, res, ns): return self.m.relax(k, da, res, ns)
    def residual(self, a, b): return self.m.residual(a, b)
    def restrict(self, k, f): return self.m.restrict(k, f)
    def prolong(self, k, c): return self.m.prolong(k, c)
    def level_mask(self, k): return self.m.level_mask(k)
    def forcing(self): self.m.forcing()
    def invert_q(self): return _stats(self.m.invert_q())
    def rhs_pv(self): self.m.rhs_pv()
    def step(self): self.m.step(True); return self.m.dt
    def set_tnext(self, t): self.m.set_tnext(t)
    @property
    def t(self): return self.m.t
    def ke(self): return self.m.ke()
    def mgstats(self): return _stats(self.m.mgstats())
    def diag1d(self): return self.m.diag1d()
    def noise(self): return self.m.noise()
    def wavelet_filter(self, dtflt): self.m.wavelet_filter(dtflt)
    def wv_apply(self, c): return self.m.wv_apply(c)
    def close(self): self.m = None


class NodeGpuModel:
    kind = "gpu"

    def __init__(self, txt, strict=True, **opts):
        from msom_amd import NodeQG
        self.m = NodeQG(txt, strict=strict)
        self.m.set_option("quiet", 1)
        for k, v in opts.items():
            self.m.set_option(k, v)
        self.nl = int(self.m.param("nl"))
        self.N = int(self.m.param("N"))

    def set(self, name, a): self.m.set(name, a)
    def get(self, name): return self.m.get(name)
    def set_const(self): self.m.set_const()
    def nlevels(self): return self.m.nlevels
    def relax(self, k, da, res, ns): return self.m.dbg_relax(k, da, res, ns)
    def residual(self, a, b): return self.m.dbg_residual(a, b)
    def restrict(self, k, f): return self.m.dbg_restrict(k, f)
    def prolong(self, k, c): return self.m.dbg_prolong(k, c)
    def level_mask(self, k): return self.m.dbg_level_mask(k)
    def forcing(self): self.m.forcing()
    def invert_q(self): return _stats(self.m.invert_q())
    def rhs_pv(self): self.m.rhs_pv()
    def step(self): self.m.step(True); return self.m.dt
    def set_tnext(self, t): self.m.set_tnext(t)
    @property
    def t(self): return self.m.t
    def ke(self): return self.m.ke()
    def mgstats(self): return _stats(self.m.mgstats())
    def diag1d(self): return self.m.diag1d()
    def noise(self): return self.m.noise()
    def wavelet_filter(self, dtflt): self.m.wavelet_filter(dtflt)
    def wv_apply(self, c): return self.m.wv_apply(c)
    def close(self): self.m.close()


NODE_PARAMS = """#!sh
N  = {N}
nl = {nl}
L0 = 100
f0 = 46.5
hEkb  = 0.01
tau0 = 1e-3
nu = 5.0
nu4 = 1.0
beta = 0.5
bc_fac = {bc_fac}
dh   = {dh}
N2   = {N2}
DT    = 5.e-2
tend  = 100.
dtout = 1
CFL   = 0.2
TOLERANCE = 1e-5
gp_low = 0.02
"""
NODE_LAYERS = {1: ("[1.0]", "[1.0]"), 3: ("[0.1,0.3,0.6]", "[9000.,3000.]")}


def node_inputs(N, nl, seed):
    """island + ragged coast mask (0 on land and on the boundary vertices), topography, background flow, smooth psi"""
    mk = np.ones((1, N + 1, N + 1))
    mk[0, N // 4: N // 4 + N // 8 + 1, N // 2: N // 2 + N // 8] = 0
    mk[0, : N // 6, : N // 5] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    x = np.arange(N + 1) / N
    psi = np.zeros((nl, N + 1, N + 1))
    for l in range(nl):
        for k in range(1, 4):
            for m in range(1, 4):
                psi[l] += np.sin(1.3 * k + 2.1 * m + 0.7 * l) / (k * m) * np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
        psi[l] *= 1e-2 * (1 - 0.2 * l)
    d = {"in_mask": mk, "in_psi": psi * mk, "in_topo": 0.05 * np.outer(np.cos(2 * np.pi * x), np.sin(np.pi * x))[None],
         "in_psipg": 0.3 * psi[::-1].copy()}
    d.update(_rng_inputs(seed, {"in_a": ((nl, N + 1, N + 1), 1.0), "in_b": ((nl, N + 1, N + 1), 1.0)}))
    return d


def case_node(make, inp, N, nl, bc_fac):
    """vertex model with a land mask: coarse masks, smoother / residual / transfer operators
    (qg-node/qg_baroclinic_ms.h:228-339, my_vertex.h:49-105), vpoisson (nodal-poisson.h:19-143), rhs_pv_baroclinic
    (qg_baroclinic_ms.h:104-196), 6 RK2 steps with the time-dependent wind event (qg-node/qg.h:258-354)"""
    dh, N2 = NODE_LAYERS[nl]
    txt = NODE_PARAMS.format(N=N, nl=nl, bc_fac=bc_fac, dh=dh, N2=N2) + "tau1 = 5e-4\ntf1 = 0.3\ntf2 = 0.7\n"
    m = make(txt, TOLERANCE=1e-8)
    m.set("MASK", inp["in_mask"])
    if nl > 1:
        m.set("TOPO", inp["in_topo"])
        m.set("PSIPG", inp["in_psipg"])
    m.set("PSI", inp["in_psi"])
    m.set_const()
    out = {"q_0": m.get("Q")}
    if nl > 1:
        out["S2"] = m.get("S2")
    da, rs = inp["in_a"], inp["in_b"]
    r, mx = m.residual(da, rs)
    out["residual"], out["residual_max"] = r, np.float64(mx)
    for k in range(m.nlevels()):
        out[f"mask_l{k}"] = m.level_mask(k)
        out[f"relax2_l{k}"] = m.relax(k, da, rs, 2)
        if k > 0:
            out[f"prolong_l{k}"] = m.prolong(k, da)
        if k + 1 < m.nlevels():
            da, rs = m.restrict(k, da), m.restrict(k, rs)
            out[f"restrict_l{k}"] = rs
    m.forcing()
    out["qforc"] = m.get("QFORC")
    out["invert_stats"] = m.invert_q()
    out["psi_inverted"] = m.get("PSI")
    m.rhs_pv()
    out["rhs_dq"] = m.get("DQ")
    m.set_tnext(0.11)
    dts, ts, sts = [], [], []
    for _ in range(6):
        dts.append(m.step()); ts.append(m.t); sts.append(m.mgstats())
    out["dt"], out["t"], out["mgstats"] = np.array(dts), np.array(ts), np.array(sts)
    out["psi_end"], out["q_end"] = m.get("PSI"), m.get("Q")
    out["ke"] = np.float64(m.ke())
    out["diag1d_rowsum"] = m.diag1d()
    m.close()
    return out


def case_node_sqg(make, inp, N, nl):
    """surface-QG option of the vertex model (params key sqg = 1): the finished parts of qg-node/sqg_baroclinic_ms.h
    (:64-67, 77-98, 160-201, 502, 545) with a prescribed surface buoyancy"""
    dh, N2 = NODE_LAYERS[nl]
    txt = NODE_PARAMS.format(N=N, nl=nl, bc_fac=1.0, dh=dh, N2="[300.," + N2[1:]) + "sqg = 1\ntau1 = 5e-4\ntf1 = 0.3\ntf2 = 0.7\n"
    m = make(txt, TOLERANCE=1e-9)
    m.set("MASK", inp["in_mask"])
    m.set("BS", inp["in_bs"])
    m.set("PSI", inp["in_psi"])
    m.set_const()
    out = {"q_0": m.get("Q"), "S2S": m.get("S2S")}
    out["invert_stats"] = m.invert_q()
    out["psi_inverted"] = m.get("PSI")
    m.rhs_pv()
    out["rhs_dq"] = m.get("DQ")
    m.set_tnext(0.11)
    out["dt"] = np.array([m.step() for _ in range(5)])
    out["psi_end"], out["q_end"] = m.get("PSI"), m.get("Q")
    m.close()
    return out


def case_node_wavelet(make, inp, N, nl):
    """wavelet filter of the vertex model (qg_baroclinic_ms.h:346-400, wavelet_vertex.h:10-46): two filter events, then two steps"""
    dh, N2 = NODE_LAYERS[nl]
    txt = NODE_PARAMS.format(N=N, nl=nl, bc_fac=1.0, dh=dh, N2=N2) + "Lfmax = 12\nLfmin = 3\n"
    m = make(txt, TOLERANCE=1e-10)
    m.set("MASK", inp["in_mask"])
    m.set("PSI", inp["in_psi"])
    m.set_const()
    out = {"filtered_cells": m.wv_apply(inp["in_cells"])}
    for k in (1, 2):
        m.wavelet_filter(0.5)
        out[f"psi_f{k}"], out[f"psif_f{k}"], out[f"q_f{k}"] = m.get("PSI"), m.get("PSIF"), m.get("Q")
    for _ in range(2):
        m.step()
    out["q_end"] = m.get("Q")
    m.close()
    return out


def case_node_stochastic(make, inp, N):
    """-D_STOCHASTIC of the vertex model (qg-node/qg_stochastic.h:15-65, qg.h:306-320), serial rand() stream, srand(11)"""
    txt = NODE_PARAMS.format(N=N, nl=1, bc_fac=1.0, dh=NODE_LAYERS[1][0], N2=NODE_LAYERS[1][1]) + "amp_stoch = 0.3\nL_filt = 8.0\n"
    m = make(txt, stochastic=1, TOLERANCE=1e-9)
    m.set("PSI", inp["in_psi"])
    m.set_const()
    ctypes.CDLL(None).srand(11)
    for _ in range(4):
        m.step()
    out = {"noise_end": m.noise(), "q_end": m.get("Q"), "psi_end": m.get("PSI")}
    m.close()
    return out


def sqg_inputs(N, nl, seed):
    d = node_inputs(N, nl, seed)
    x = np.arange(N + 1) / N
    return {"in_mask": d["in_mask"], "in_psi": d["in_psi"], "in_bs": 0.3 * np.outer(np.sin(np.pi * x), np.sin(2 * np.pi * x))[None] + 0.05}


def wavelet_inputs(N, nl, seed):
    d = node_inputs(N, nl, seed)
    return {"in_mask": d["in_mask"], "in_psi": d["in_psi"], "in_cells": np.random.default_rng(seed).standard_normal((nl, N, N))}


NODE_CASES = {
    "node_wavelet_32x3": (case_node_wavelet, dict(N=32, nl=3), lambda: wavelet_inputs(32, 3, 205)),
    "node_sqg_32x3": (case_node_sqg, dict(N=32, nl=3), lambda: sqg_inputs(32, 3, 204)),
    "node_island_32x3": (case_node, dict(N=32, nl=3, bc_fac=1.0), lambda: node_inputs(32, 3, 201)),
    "node_island_64x1": (case_node, dict(N=64, nl=1, bc_fac=0.5), lambda: node_inputs(64, 1, 202)),
    "node_stochastic_32x1": (case_node_stochastic, dict(N=32), lambda: {"in_psi": node_inputs(32, 1, 203)["in_psi"]}),
}


# name -> (case function, kwargs, input builder or None).  Sizes: C1 = 128^2 x 1, C2 = 512^2 x 3 (BASELINE.json configs).
P0BAS = os.path.join(GOLDEN, "p0_32x3.bas")
CASES = {
    "ops_32x32x3": (case_ops, dict(nx=32, ny=32, nl=3), lambda: ops_inputs(32, 32, 3, 101)),
    "ops_64x32x2": (case_ops, dict(nx=64, ny=32, nl=2), lambda: ops_inputs(64, 32, 2, 102)),
    "ops_16x16x6": (case_ops, dict(nx=16, ny=16, nl=6), lambda: ops_inputs(16, 16, 6, 103)),
    "ops_32x32x1": (case_ops, dict(nx=32, ny=32, nl=1), lambda: ops_inputs(32, 32, 1, 104)),
    "forcing_32x32x3": (case_forcing, dict(nx=32, ny=32, nl=3), lambda: forcing_inputs(32, 32, 3, 105)),
    "run_p0bas_32x32x3": (case_run, dict(nx=32, nl=3, p0bas=P0BAS), None),
    "run_p0bas_32x32x3_tol1e-12": (case_run, dict(nx=32, nl=3, p0bas=P0BAS, TOLERANCE=1e-12), None),
    "run_C1_128x128x1": (case_run, dict(nx=128, nl=1, snap=(10,)), None),
    "run_C2_512x512x3": (case_run, dict(nx=512, nl=3, snap=(10,), stride=8), None),
    "stochastic_srand7_16x16x3": (case_stochastic, dict(nx=16, nl=3),
                                  lambda: {"in_sigma": np.abs(_rng_inputs(106, {"s": ((3, 16, 16), 1.0)})["s"])}),
    "tracers_32x32x3": (case_tracers, dict(nx=32, nl=3, nptr=2),
                        lambda: _rng_inputs(107, {"in_c0": ((6, 32, 32), 1e-3), "in_relax": ((6, 32, 32), 1e-3)})),
    "wavelet_64x64x3": (case_wavelet, dict(N=64, nl=3),
                        lambda: {"in_rd": np.concatenate([np.ones((1, 64, 32)), 3.0 * np.ones((1, 64, 32))], axis=2)}),
}
# the oracle-only variant with the reference's lexicographic sweep order (not reproducible on the GPU: order-dependent)
LEX_CASES = {"run_p0bas_32x32x3_lexicographic": (case_run, dict(nx=32, nl=3, p0bas=P0BAS, smoother=0), None),
             "run_p0bas_32x32x3_lexicographic_tol1e-12": (case_run, dict(nx=32, nl=3, p0bas=P0BAS, smoother=0, TOLERANCE=1e-12), None)}

# results that are sums over the grid: the summation order differs between OpenMP teams and the GPU reductions
SUM_KEYS = ("ke", "mgstats", "_stats", "_rowsum")


def is_sum_key(k):
    return any(s in k for s in SUM_KEYS)


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def run_case(name, make, table=None):
    fn, kw, _ = (table or CASES)[name]
    gold = load(name)
    inp = {k: v for k, v in gold.items() if k.startswith("in_")}
    exp = {k: v for k, v in gold.items() if not k.startswith("in_")}
    return fn(make, inp, **kw), exp


def compare(got, exp, exact=True, rtol=0.0, sum_rtol=1e-12):
    """exact: bit-for-bit on fields (np.array_equal), sum_rtol on grid sums; otherwise rtol relative to max|expected|"""
    assert set(got) == set(exp), (sorted(set(got) ^ set(exp)))
    for k in sorted(exp):
        g, e = np.asarray(got[k]), exp[k]
        assert g.shape == e.shape, (k, g.shape, e.shape)
        if is_sum_key(k):
            if k.endswith("stats") or k == "mgstats":   # columns i, resb, resa, sum, nrelax: only `sum` is a sum
                gi, ei = np.atleast_2d(g), np.atleast_2d(e)
                if exact:
                    assert np.array_equal(gi[:, [0, 1, 2, 4]], ei[:, [0, 1, 2, 4]]), k
                else:
                    assert np.array_equal(gi[:, [0, 4]], ei[:, [0, 4]]), k
                    # product build: residual before to rounding; the residual after is cancellation noise once the
                    # solve has converged far below resb, so it is compared relative to resb
                    assert np.allclose(gi[:, 1], ei[:, 1], rtol=max(rtol, 1e-6), atol=1e-15), k
                    # (1e-12: a solve that starts converged, residuals of 1e-13 before and after, is rounding of the operands alone)
                    assert np.all(np.abs(gi[:, 2] - ei[:, 2]) <= 1e-3 * np.abs(ei[:, 2]) + 1e-8 * np.abs(ei[:, 1]) + 1e-12), k
                assert np.allclose(gi[:, 3], ei[:, 3], rtol=sum_rtol, atol=1e-16), k
            else:
                tol = sum_rtol if exact else max(rtol, sum_rtol)
                assert np.abs(g - e).max() <= tol * max(np.abs(e).max(), 1e-300), (k, np.abs(g - e).max())
        elif exact:
            assert np.array_equal(g, e), (k, float(np.abs(g - e).max()))
        else:
            assert np.abs(g - e).max() <= rtol * max(np.abs(e).max(), 1e-300), (k, float(np.abs(g - e).max() / max(np.abs(e).max(), 1e-300)))
