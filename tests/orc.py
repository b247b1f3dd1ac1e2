"""ctypes binding of the CPU oracle (oracle/qg_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_ROOT, "oracle", "_build", "liborc.so")

PSI, Q, ZETA, PSIPG, ZETAPG, QFORC, TMP, FR, S, DQ, RO, TOPO, QPRED, NOISE, SIGMA, PTR, PTR_RELAX, DPTR, PTR_PRED = range(19)
RD, QOF = 19, 20
DE_BF, DE_VD, DE_J1, DE_J2, DE_J3, DE_FT, TMP2, PO_MFT = range(21, 29)
GS_LEX, GS_RB = 0, 1


class MGStats(C.Structure):
    _fields_ = [("i", C.c_int), ("resb", C.c_double), ("resa", C.c_double), ("sum", C.c_double), ("nrelax", C.c_int)]


def build():
    srcs = [os.path.join(_ROOT, "oracle", f) for f in ("qg_oracle.c", "qgnode_oracle.c")]
    if (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def _limit_threads():
    # GPU boxes expose every host core but give the job a small CPU share: an unbounded
    # OpenMP team there spends its time spinning.  Must be set before libgomp starts.
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, n))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def lib():
    global _lib
    if _lib is None:
        _limit_threads()
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L = _lib
        L.orc_create_str.restype = C.c_void_p
        L.orc_create_str.argtypes = [C.c_char_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.orc_get_param.restype = C.c_double
        L.orc_get_param.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_set_const.argtypes = [C.c_void_p]
        L.orc_nlayers_of.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_field.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_get_field.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_remove_mean.argtypes = [C.c_void_p, C.c_int]
        L.orc_comp_del2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_comp_stretch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_comp_q.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_advection_pv.restype = C.c_double
        L.orc_advection_pv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_dissip.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_forcing_terms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_invertq.restype = MGStats
        L.orc_invertq.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_update.restype = C.c_double
        L.orc_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.orc_advance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_ke.restype = C.c_double
        L.orc_ke.argtypes = [C.c_void_p]
        L.orc_timestep_limiter.restype = C.c_double
        L.orc_timestep_limiter.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_reset_limiter.argtypes = [C.c_void_p]
        L.orc_nlevels.argtypes = [C.c_void_p]
        L.orc_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_relax_raw.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int]
        L.orc_residual_raw.restype = C.c_double
        L.orc_residual_raw.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_restrict_raw.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int]
        L.orc_prolong_raw.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_step.argtypes = [C.c_void_p]
        L.orc_time.restype = C.c_double
        L.orc_time.argtypes = [C.c_void_p]
        L.orc_dt.restype = C.c_double
        L.orc_dt.argtypes = [C.c_void_p]
        L.orc_iter.argtypes = [C.c_void_p]
        L.orc_last_mgstats.restype = MGStats
        L.orc_last_mgstats.argtypes = [C.c_void_p]
        L.orc_set_tnext.argtypes = [C.c_void_p, C.c_double]
        L.orc_pystep_bfn.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_int]
        L.orc_pyq2p.argtypes = [C.c_void_p, dp, dp]
        L.orc_pyp2q.argtypes = [C.c_void_p, dp, dp]
        L.orc_write_bas.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_read_bas.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_wavelet_filter.argtypes = [C.c_void_p, C.c_double]
        L.orc_energy_tend.argtypes = [C.c_void_p, C.c_double]
        L.orc_filter_de.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_reset_de.argtypes = [C.c_void_p]
        L.orc_pystep_de.argtypes = [C.c_void_p] + [dp] * 7 + [C.c_int]
        L.orc_wavelet_levels.argtypes = [C.c_void_p]
        L.orc_get_siglev.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_wavelet_apply.argtypes = [C.c_void_p, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One oracle instance = one `qg.e` process of the reference (msqg formulation)."""

    def __init__(self, params_text, smoother=GS_RB, **options):
        self.L = lib()
        self.h = self.L.orc_create_str(params_text.encode())
        if not self.h:
            raise ValueError("bad params")
        self.nx = int(self.param("nx"))
        self.ny = int(self.param("ny"))
        self.nl = int(self.param("nl"))
        self.option("smoother", smoother)
        for k, v in options.items():
            self.option(k, v)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def option(self, key, v):
        if self.L.orc_set_option(self.h, key.encode(), float(v)) != 0:
            raise KeyError(key)

    def param(self, key):
        return self.L.orc_get_param(self.h, key.encode())

    def set_const(self):
        self.L.orc_set_const(self.h)

    def shape(self, field):
        return (self.L.orc_nlayers_of(self.h, field), self.ny, self.nx)

    def set(self, field, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == self.shape(field), (a.shape, self.shape(field))
        self.L.orc_set_field(self.h, field, _p(a))

    def get(self, field):
        a = np.empty(self.shape(field), dtype=np.float64)
        self.L.orc_get_field(self.h, field, _p(a))
        return a

    def remove_mean(self, field):
        self.L.orc_remove_mean(self.h, field)

    def comp_del2(self, fin, fout, add, fac):
        self.L.orc_comp_del2(self.h, fin, fout, add, fac)

    def comp_stretch(self, fin, fout, add, fac):
        self.L.orc_comp_stretch(self.h, fin, fout, add, fac)

    def comp_q(self, psi=PSI, q=Q):
        self.L.orc_comp_q(self.h, psi, q)

    def advection_pv(self, zeta, q, psi, dq, dtmax):
        return self.L.orc_advection_pv(self.h, zeta, q, psi, dq, dtmax)

    def dissip(self, zeta=ZETA, dq=DQ):
        self.L.orc_dissip(self.h, zeta, dq)

    def forcing_terms(self, zeta=ZETA, psi=PSI, dq=DQ):
        self.L.orc_forcing_terms(self.h, zeta, psi, dq)

    def invertq(self, psi=PSI, q=Q):
        return self.L.orc_invertq(self.h, psi, q)

    def update(self, q=Q, dq=DQ, dtmax=None):
        return self.L.orc_update(self.h, q, dq, self.param("DT") if dtmax is None else dtmax)

    def advance(self, out, inp, dq, dt):
        self.L.orc_advance(self.h, out, inp, dq, dt)

    def ke(self):
        return self.L.orc_ke(self.h)

    def limiter(self, dtmin_faces, dtmax):
        return self.L.orc_timestep_limiter(self.h, dtmin_faces, dtmax)

    def reset_limiter(self):
        self.L.orc_reset_limiter(self.h)

    def nlevels(self):
        return self.L.orc_nlevels(self.h)

    def level_dims(self, lev):
        nx, ny = C.c_int(), C.c_int()
        self.L.orc_level_dims(self.h, lev, C.byref(nx), C.byref(ny))
        return nx.value, ny.value

    def relax(self, lev, da, res, nsweeps=1):
        da = np.array(da, dtype=np.float64, order="C")
        res = np.ascontiguousarray(res, dtype=np.float64)
        self.L.orc_relax_raw(self.h, lev, _p(da), _p(res), nsweeps)
        return da

    def residual(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        res = np.empty_like(a)
        m = self.L.orc_residual_raw(self.h, _p(a), _p(b), _p(res))
        return res, m

    def restrict(self, lev_fine, fine):
        fine = np.ascontiguousarray(fine, dtype=np.float64)
        nlay, ny, nx = fine.shape
        coarse = np.empty((nlay, ny // 2, nx // 2))
        self.L.orc_restrict_raw(self.h, lev_fine, _p(fine), _p(coarse), nlay)
        return coarse

    def prolong(self, lev_coarse, coarse):
        coarse = np.ascontiguousarray(coarse, dtype=np.float64)
        nlay, ny, nx = coarse.shape
        fine = np.empty((nlay, ny * 2, nx * 2))
        self.L.orc_prolong_raw(self.h, lev_coarse, _p(coarse), _p(fine))
        return fine

    def step(self):
        self.L.orc_step(self.h)

    @property
    def t(self):
        return self.L.orc_time(self.h)

    @property
    def dt(self):
        return self.L.orc_dt(self.h)

    @property
    def iter(self):
        return self.L.orc_iter(self.h)

    def mgstats(self):
        return self.L.orc_last_mgstats(self.h)

    def set_tnext(self, tnext):
        self.L.orc_set_tnext(self.h, tnext)

    def pystep_bfn(self, q, direction=1.0, vartype=1):
        q = np.ascontiguousarray(q, dtype=np.float64)
        tend = np.empty_like(q)
        self.L.orc_pystep_bfn(self.h, _p(q), _p(tend), direction, vartype)
        return tend

    def pyq2p(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        p = np.empty_like(q)
        self.L.orc_pyq2p(self.h, _p(p), _p(q))
        return p

    def pyp2q(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        q = np.empty_like(p)
        self.L.orc_pyp2q(self.h, _p(p), _p(q))
        return q

    def wavelet_filter(self, dtflt):
        self.L.orc_wavelet_filter(self.h, dtflt)

    def wavelet_levels(self):
        return self.L.orc_wavelet_levels(self.h)

    def energy_tend(self, dt):
        self.L.orc_energy_tend(self.h, dt)

    def filter_de(self, pm_field, dtflt):
        self.L.orc_filter_de(self.h, pm_field, dtflt)

    def reset_de(self):
        self.L.orc_reset_de(self.h)

    def pystep_de(self, po, onlyKE=0):
        po = np.ascontiguousarray(po, dtype=np.float64)
        outs = [np.empty_like(po) for _ in range(6)]
        self.L.orc_pystep_de(self.h, _p(po), *[_p(a) for a in outs], onlyKE)
        return outs

    def siglev(self, lev):
        a = np.empty((1, self.ny >> lev, self.nx >> lev))
        self.L.orc_get_siglev(self.h, lev, _p(a))
        return a

    def wavelet_apply(self, field):
        self.L.orc_wavelet_apply(self.h, field)

    def write_bas(self, field, path):
        return self.L.orc_write_bas(self.h, field, path.encode())

    def read_bas(self, field, path):
        return self.L.orc_read_bas(self.h, field, path.encode())


# canonical synthetic inputs: defined in msom_amd/workloads.py (no oracle there), re-exported for the tests
import sys as _sys

_sys.path.insert(0, _ROOT)
from msom_amd.workloads import DOUBLE_GYRE, LAYERS, double_gyre_params, synthetic_psi  # noqa: E402,F401
