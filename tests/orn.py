"""ctypes binding of the vertex-grid oracle (oracle/qgnode_oracle.c).  Test infrastructure only."""
import ctypes as C

import numpy as np

import orc

PSI, Q, ZETA, TMP, PSIPG, S2, TOPO, QFORC, MASK, DQ, QPRED, QFORC3D, BS, S2S, PSIF = range(15)
GS_LEX, GS_RB = 0, 1


class MGStats(C.Structure):
    _fields_ = [("i", C.c_int), ("resb", C.c_double), ("resa", C.c_double), ("sum", C.c_double), ("nrelax", C.c_int)]


_done = False


def lib():
    global _done
    L = orc.lib()
    if not _done:
        vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
        dp = C.POINTER(C.c_double)
        for name, res, args in [
            ("orn_create_str", vp, [cs]), ("orn_destroy", None, [vp]), ("orn_set_option", ci, [vp, cs, cd]),
            ("orn_get_param", cd, [vp, cs]), ("orn_nlayers_of", ci, [vp, ci]), ("orn_set_field", None, [vp, ci, dp]),
            ("orn_get_field", None, [vp, ci, dp]), ("orn_set_const", None, [vp]), ("orn_update", cd, [vp, ci, ci, cd]),
            ("orn_advance", None, [vp, ci, ci, ci, cd]), ("orn_forcing", None, [vp]), ("orn_step", ci, [vp, ci]),
            ("orn_ke", cd, [vp]), ("orn_time", cd, [vp]), ("orn_dt", cd, [vp]), ("orn_set_tnext", None, [vp, cd]),
            ("orn_last_mgstats", MGStats, [vp]), ("orn_invert_q", MGStats, [vp, ci]), ("orn_comp_q", None, [vp, ci, ci]),
            ("orn_rhs_pv", None, [vp, ci, ci]), ("orn_comp_del2_zeta", None, [vp]), ("orn_relax_raw", None, [vp, ci, dp, dp, ci]),
            ("orn_residual_raw", cd, [vp, dp, dp, dp]), ("orn_restrict_raw", None, [vp, ci, dp, dp]),
            ("orn_prolong_raw", None, [vp, ci, dp, dp]), ("orn_get_level_mask", None, [vp, ci, dp]),
            ("orn_diag1d", None, [vp, dp]), ("orn_get_noise", None, [vp, dp]), ("orn_set_noise", None, [vp, dp]), ("orn_filter_noise", None, [vp]),
            ("orn_get_csig", None, [vp, ci, dp]), ("orn_cell_levels", ci, [vp]),
            ("orn_wavelet_filter", None, [vp, cd]), ("orn_wv_get", None, [vp, ci, ci, dp]), ("orn_wv_apply", None, [vp, dp, dp]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _done = True
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class NodeOracle:
    def __init__(self, params_text, smoother=GS_RB, **options):
        self.L = lib()
        self.h = self.L.orn_create_str(params_text.encode())
        if not self.h:
            raise ValueError("bad params")
        self.N = int(self.param("N"))
        self.nl = int(self.param("nl"))
        self.option("smoother", smoother)
        for k, v in options.items():
            self.option(k, v)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orn_destroy(self.h)
            self.h = None

    def option(self, k, v):
        if self.L.orn_set_option(self.h, k.encode(), float(v)) != 0:
            raise KeyError(k)

    def param(self, k):
        return self.L.orn_get_param(self.h, k.encode())

    def shape(self, f):
        return (self.L.orn_nlayers_of(self.h, f), self.N + 1, self.N + 1)

    def set(self, f, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == self.shape(f), (a.shape, self.shape(f))
        self.L.orn_set_field(self.h, f, _p(a))

    def get(self, f):
        a = np.empty(self.shape(f))
        self.L.orn_get_field(self.h, f, _p(a))
        return a

    def set_const(self):
        self.L.orn_set_const(self.h)

    def update(self, q=Q, dq=DQ, dtmax=None):
        return self.L.orn_update(self.h, q, dq, self.param("DT") if dtmax is None else dtmax)

    def advance(self, out, inp, dq, dt):
        self.L.orn_advance(self.h, out, inp, dq, dt)

    def forcing(self):
        self.L.orn_forcing(self.h)

    def step(self, forcing_event=True):
        self.L.orn_step(self.h, int(forcing_event))

    def ke(self):
        return self.L.orn_ke(self.h)

    @property
    def t(self):
        return self.L.orn_time(self.h)

    @property
    def dt(self):
        return self.L.orn_dt(self.h)

    def set_tnext(self, t):
        self.L.orn_set_tnext(self.h, t)

    def mgstats(self):
        return self.L.orn_last_mgstats(self.h)

    def invert_q(self, q=Q):
        return self.L.orn_invert_q(self.h, q)

    def comp_q(self, psi=PSI, q=Q):
        self.L.orn_comp_q(self.h, psi, q)

    def rhs_pv(self, q=Q, dq=DQ):
        self.L.orn_rhs_pv(self.h, q, dq)

    def comp_zeta(self):
        self.L.orn_comp_del2_zeta(self.h)

    def nlevels(self):
        return int(self.param("nlevels"))

    def relax(self, lev, da, res, nsweeps=1):
        da = np.array(da, dtype=np.float64, order="C")
        res = np.ascontiguousarray(res, dtype=np.float64)
        self.L.orn_relax_raw(self.h, lev, _p(da), _p(res), nsweeps)
        return da

    def residual(self, a, b):
        a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
        r = np.empty_like(a)
        m = self.L.orn_residual_raw(self.h, _p(a), _p(b), _p(r))
        return r, m

    def restrict(self, lev, fine):
        fine = np.ascontiguousarray(fine, dtype=np.float64)
        nl, n1, _ = fine.shape
        c = np.empty((nl, (n1 - 1) // 2 + 1, (n1 - 1) // 2 + 1))
        self.L.orn_restrict_raw(self.h, lev, _p(fine), _p(c))
        return c

    def prolong(self, lev, coarse):
        coarse = np.ascontiguousarray(coarse, dtype=np.float64)
        nl, n1, _ = coarse.shape
        f = np.empty((nl, 2 * (n1 - 1) + 1, 2 * (n1 - 1) + 1))
        self.L.orn_prolong_raw(self.h, lev, _p(coarse), _p(f))
        return f

    def diag1d(self):
        a = np.empty(3)
        self.L.orn_diag1d(self.h, _p(a))
        return a

    # stochastic forcing (cell scalars n_stoch, sig_lev)
    def noise(self):
        a = np.empty((self.N, self.N))
        self.L.orn_get_noise(self.h, _p(a))
        return a

    def set_noise(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.L.orn_set_noise(self.h, _p(a))

    def filter_noise(self):
        self.L.orn_filter_noise(self.h)

    def csig(self, k):
        n = self.N >> k
        a = np.empty((n, n))
        self.L.orn_get_csig(self.h, k, _p(a))
        return a

    def cell_levels(self):
        return self.L.orn_cell_levels(self.h)

    # wavelet filter of the vertex model (qg_baroclinic_ms.h:346-400)
    def wavelet_filter(self, dtflt):
        self.L.orn_wavelet_filter(self.h, dtflt)

    def wv_get(self, what, k):
        n = self.N >> k
        a = np.empty((n, n))
        self.L.orn_wv_get(self.h, what, k, _p(a))
        return a

    def wv_apply(self, cells):
        cells = np.ascontiguousarray(cells, dtype=np.float64)
        out = np.empty_like(cells)
        self.L.orn_wv_apply(self.h, _p(cells), _p(out))
        return out

    def level_mask(self, lev):
        n1 = (self.N >> lev) + 1
        a = np.empty((1, n1, n1))
        self.L.orn_get_level_mask(self.h, lev, _p(a))
        return a


NODE_PARAMS = """#!sh
N  = {N}
nl = {nl}
L0 = 100
f0 = 46.5
hEkb  = 0.01
tau0 = 1e-3
nu = {nu}
nu4 = {nu4}
beta = 0.5
bc_fac = {bc_fac}
dh   = {dh}
N2   = {N2}
DT    = 5.e-2
tend  = 100.
dtout = 1
CFL   = 0.2
TOLERANCE = 1e-5
"""
NODE_LAYERS = {1: ("[1.0]", "[1.0]"), 2: ("[0.3,0.7]", "[4000.]"), 3: ("[0.1,0.3,0.6]", "[9000.,3000.]"),
               4: ("[0.1,0.2,0.3,0.4]", "[9000.,5000.,2000.]")}


def node_params(N, nl, bc_fac=0.0, nu=5.0, nu4=0.0, extra=""):
    dh, N2 = NODE_LAYERS[nl]
    return NODE_PARAMS.format(N=N, nl=nl, nu=nu, nu4=nu4, bc_fac=bc_fac, dh=dh, N2=N2) + extra


def node_psi(nl, N, amp=1e-2):
    """smooth seed-free psi on the (N+1)^2 vertices, zero on the boundary vertices"""
    x = np.arange(N + 1) / N
    psi = np.zeros((nl, N + 1, N + 1))
    for l in range(nl):
        for k in range(1, 4):
            for m in range(1, 4):
                psi[l] += np.sin(1.3 * k + 2.1 * m + 0.7 * l) / (k * m) * np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
        psi[l] *= amp * (1 - 0.2 * l)
    psi[:, 0, :] = psi[:, -1, :] = 0
    psi[:, :, 0] = psi[:, :, -1] = 0
    return psi
