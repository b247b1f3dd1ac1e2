"""ctypes binding of libmsomhip (include/msom.h) and the host mirror of the reference's
Python surface (msqg/qg.i:29-36, msqg/qg_bfn.i, usage in msqg/qg_bfn.py:33-80)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")

FIELDS = dict(PSI=0, Q=1, ZETA=2, PSIPG=3, ZETAPG=4, QFORC=5, TMP=6, FR=7, S=8, DQ=9, RO=10, TOPO=11,
              QPRED=12, NOISE=13, SIGMA=14, PTR=15, PTR_RELAX=16, DPTR=17, PTR_PRED=18, RD=19, QOF=20,
              DE_BF=21, DE_VD=22, DE_J1=23, DE_J2=24, DE_J3=25, DE_FT=26, TMP2=27, PO_MFT=28)


class MsomError(RuntimeError):
    pass


class MGStats(C.Structure):
    """mgstats of Basilisk (mspg/elliptic.h:118-123)."""
    _fields_ = [("i", C.c_int), ("resb", C.c_double), ("resa", C.c_double), ("sum", C.c_double), ("nrelax", C.c_int)]

    def __repr__(self):
        return f"MGStats(i={self.i}, resb={self.resb:g}, resa={self.resa:g}, sum={self.sum:g}, nrelax={self.nrelax})"


_libs = {}
_dp = C.POINTER(C.c_double)


def load_library(strict=False):
    """Load the HIP library.  strict=True loads the validation build (-ffp-contract=off,
    reference expression order) that is bit-exact against the CPU oracle."""
    name = "libmsomhip_strict.so" if strict else "libmsomhip.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise MsomError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(make -C msom_amd/csrc).  There is no CPU fallback.")
    L = C.CDLL(path)
    vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
    sig = {
        "msom_last_error": (cs, []),
        "msom_version": (cs, []),
        "msom_create": (vp, [cs]),
        "msom_create_str": (vp, [cs]),
        "msom_destroy": (ci, [vp]),
        "msom_set_option": (ci, [vp, cs, cd]),
        "msom_get_param": (cd, [vp, cs]),
        "msom_set_field": (ci, [vp, ci, vp]),
        "msom_get_field": (ci, [vp, ci, vp]),
        "msom_field_layers": (ci, [vp, ci]),
        "msom_remove_mean": (ci, [vp, ci]),
        "msom_set_const": (ci, [vp]),
        "msom_read_inputs": (ci, [vp, cs]),
        "msom_update": (cd, [vp, vp, vp, cd]),
        "msom_advance": (ci, [vp, vp, vp, vp, cd]),
        "msom_invertq": (ci, [vp, vp, vp, C.POINTER(MGStats)]),
        "msom_comp_q": (ci, [vp, vp, vp]),
        "pystep_bfn": (ci, [vp, vp, ci, ci, ci, vp, ci, ci, ci, cd, ci]),
        "pyq2p": (ci, [vp, vp, ci, ci, ci, vp, ci, ci, ci]),
        "pyp2q": (ci, [vp, vp, ci, ci, ci, vp, ci, ci, ci]),
        "msom_step": (ci, [vp, _dp]),
        "msom_set_tnext": (ci, [vp, cd]),
        "msom_time": (cd, [vp]),
        "msom_iter": (ci, [vp]),
        "msom_ke": (cd, [vp]),
        "msom_last_mgstats": (ci, [vp, C.POINTER(MGStats)]),
        "msom_run": (ci, [vp, cs, C.c_long]),
        "msom_write_bas": (ci, [vp, ci, cs]),
        "msom_read_bas": (ci, [vp, ci, cs]),
        "msom_write_nc": (ci, [vp, cs]),
        "msom_read_nc": (ci, [vp, ci, cs, cs, ci]),
        "msom_set_device": (ci, [ci]),
        "msom_comm_unique_id": (ci, [vp]),
        "msom_create_tiled": (vp, [cs, ci, ci, ci, vp]),
        "msom_tile_info": (ci, [vp] + [C.POINTER(ci)] * 6),
        "msom_dbg_nlevels": (ci, [vp]),
        "msom_dbg_level_dims": (ci, [vp, ci, C.POINTER(ci), C.POINTER(ci)]),
        "msom_dbg_relax": (ci, [vp, ci, vp, vp, ci]),
        "msom_dbg_residual": (ci, [vp, vp, vp, vp, _dp]),
        "msom_dbg_restrict": (ci, [vp, ci, vp, vp]),
        "msom_dbg_prolong": (ci, [vp, ci, vp, vp]),
        "msom_dbg_op": (ci, [vp, cs, ci, ci, cd, cd]),
        "msom_profile_read": (ci, [vp, cs, _dp, C.POINTER(C.c_long)]),
        "msom_profile_reset": (ci, [vp]),
        "msom_sync": (ci, [vp]),
        "msom_bench_kernel": (ci, [vp, cs, ci, _dp]),
        "msom_dbg_rccl_selftest": (ci, []),
        "msom_wavelet_filter": (ci, [vp, cd]),
        "msom_energy_tend": (ci, [vp, cd]),
        "msom_filter_de": (ci, [vp, ci, cd]),
        "msom_reset_de": (ci, [vp]),
        "pystep_de": (ci, [vp] + [vp, ci, ci, ci] * 7 + [ci]),
        "msom_dbg_wavelet_levels": (ci, [vp]),
        "msom_dbg_siglev": (ci, [vp, ci, vp]),
        "msom_dbg_wavelet_apply": (ci, [vp, ci]),
        # vertex-grid variant (qg-node/)
        "msomn_create": (vp, [cs]),
        "msomn_create_str": (vp, [cs]),
        "msomn_destroy": (None, [vp]),
        "msomn_set_option": (ci, [vp, cs, cd]),
        "msomn_get_param": (cd, [vp, cs]),
        "msomn_field_layers": (ci, [vp, ci]),
        "msomn_set_field": (ci, [vp, ci, vp]),
        "msomn_get_field": (ci, [vp, ci, vp]),
        "msomn_set_const": (ci, [vp]),
        "msomn_update": (ci, [vp, ci, ci, cd, _dp]),
        "msomn_advance": (ci, [vp, ci, ci, ci, cd]),
        "msomn_invert_q": (ci, [vp, ci, C.POINTER(MGStats)]),
        "msomn_comp_q": (ci, [vp, ci, ci]),
        "msomn_rhs_pv": (ci, [vp, ci, ci]),
        "msomn_forcing": (ci, [vp]),
        "msomn_step": (ci, [vp, ci]),
        "msomn_set_tnext": (ci, [vp, cd]),
        "msomn_time": (cd, [vp]),
        "msomn_dt": (cd, [vp]),
        "msomn_iter": (ci, [vp]),
        "msomn_ke": (ci, [vp, _dp]),
        "msomn_last_mgstats": (ci, [vp, C.POINTER(MGStats)]),
        "msomn_profile_read": (ci, [vp, cs, _dp, C.POINTER(C.c_long)]),
        "msomn_profile_reset": (ci, [vp]),
        "msomn_diag1d": (ci, [vp, _dp]),
        "msomn_write_nc": (ci, [vp, cs]),
        "msomn_read_nc": (ci, [vp, ci, cs, cs, ci]),
        "msomn_run": (ci, [vp, cs, C.c_long]),
        "msomn_dbg_relax": (ci, [vp, ci, vp, vp, ci]),
        "msomn_dbg_residual": (ci, [vp, vp, vp, vp, _dp]),
        "msomn_dbg_restrict": (ci, [vp, ci, vp, vp]),
        "msomn_dbg_prolong": (ci, [vp, ci, vp, vp]),
        "msomn_dbg_level_mask": (ci, [vp, ci, vp]),
        "msomn_dbg_del2_zeta": (ci, [vp]),
        "msomn_dbg_noise": (ci, [vp, vp, ci, vp]),
        "msomn_dbg_csig": (ci, [vp, ci, vp]),
        "msomn_wavelet_filter": (ci, [vp, cd]),
        "msomn_dbg_wv_get": (ci, [vp, ci, ci, vp]),
        "msomn_dbg_wv_apply": (ci, [vp, vp, vp]),
    }
    for fn, (res, args) in sig.items():
        f = getattr(L, fn)  # AttributeError if the library does not export a declared symbol
        f.restype, f.argtypes = res, args
    _libs[name] = L
    return L


def _ptr(a):
    """numpy array (host) or an object with .data_ptr() (torch tensor, host or device)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _f64(a, shape=None):
    if hasattr(a, "data_ptr"):
        return a
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"array shape {a.shape} != {tuple(shape)}")
    return a


class QG:
    """One model instance = one `qg.e` process / one `import qg` of the reference."""

    def __init__(self, params=None, path=None, strict=False, tiled=None):
        self.L = load_library(strict)
        self.h = None
        if tiled is not None:
            px, py, rank, uid = tiled
            self.h = self.L.msom_create_tiled(params.encode(), px, py, rank, uid)
        elif path is not None:
            self.h = self.L.msom_create(path.encode())
        else:
            self.h = self.L.msom_create_str(params.encode())
        if not self.h:
            raise MsomError(self.L.msom_last_error().decode())
        self.nl = int(self.param("nl"))
        px_, py_, ix, iy, nx, ny = (C.c_int() for _ in range(6))
        self.L.msom_tile_info(self.h, *(C.byref(v) for v in (px_, py_, ix, iy, nx, ny)))
        self.nx, self.ny = nx.value, ny.value
        self.tile = (px_.value, py_.value, ix.value, iy.value)

    def close(self):
        if self.h:
            self.L.msom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise MsomError(f"error {r}: {self.L.msom_last_error().decode()}")

    # -- parameters / options
    def param(self, key):
        return self.L.msom_get_param(self.h, key.encode())

    def option(self, key, value):
        self._chk(self.L.msom_set_option(self.h, key.encode(), float(value)))

    # -- fields (pyset_field / pyget_field, msqg/qg.h:1164-1188)
    def shape(self, field):
        return (self.L.msom_field_layers(self.h, field), self.ny, self.nx)

    def set(self, field, a):
        a = _f64(a, self.shape(field))
        self._chk(self.L.msom_set_field(self.h, field, _ptr(a)))

    def get(self, field):
        a = np.empty(self.shape(field))
        self._chk(self.L.msom_get_field(self.h, field, _ptr(a)))
        return a

    def remove_mean(self, field):
        self._chk(self.L.msom_remove_mean(self.h, field))

    def set_const(self):
        self._chk(self.L.msom_set_const(self.h))

    def read_inputs(self, directory="."):
        self._chk(self.L.msom_read_inputs(self.h, directory.encode()))

    # -- hooks
    def update(self, q=None, dtmax=None, want=True):
        """update_qg: returns (dqdt, dtmax)."""
        q = None if q is None else _f64(q, self.shape(1))
        out = np.empty(self.shape(1)) if want else None
        d = self.L.msom_update(self.h, _ptr(q), _ptr(out), self.param("DT") if dtmax is None else dtmax)
        if d < 0:
            raise MsomError(self.L.msom_last_error().decode())
        return out, d

    def advance(self, qin, dqdt, dt):
        qin, dqdt = _f64(qin, self.shape(1)), _f64(dqdt, self.shape(1))
        out = np.empty(self.shape(1))
        self._chk(self.L.msom_advance(self.h, _ptr(out), _ptr(qin), _ptr(dqdt), dt))
        return out

    def invertq(self, q=None, psi0=None):
        q = None if q is None else _f64(q, self.shape(1))
        psi = np.zeros(self.shape(0)) if psi0 is None else np.array(psi0, dtype=np.float64, order="C")
        st = MGStats()
        self._chk(self.L.msom_invertq(self.h, _ptr(q), _ptr(psi), C.byref(st)))
        return psi, st

    def comp_q(self, psi):
        psi = _f64(psi, self.shape(0))
        q = np.empty(self.shape(1))
        self._chk(self.L.msom_comp_q(self.h, _ptr(psi), _ptr(q)))
        return q

    # -- msqg/qg_bfn.h
    def pystep_bfn(self, var, tend, direction=1.0, vartype=1):
        var = _f64(var)
        assert tend.dtype == np.float64 and tend.flags.c_contiguous
        self._chk(self.L.pystep_bfn(self.h, _ptr(var), *var.shape, _ptr(tend), *tend.shape, direction, vartype))

    def pyq2p(self, p, q):
        q = _f64(q)
        assert p.dtype == np.float64 and p.flags.c_contiguous
        self._chk(self.L.pyq2p(self.h, _ptr(p), *p.shape, _ptr(q), *q.shape))

    def pyp2q(self, p, q):
        p = _f64(p)
        assert q.dtype == np.float64 and q.flags.c_contiguous
        self._chk(self.L.pyp2q(self.h, _ptr(p), *p.shape, _ptr(q), *q.shape))

    # -- time loop
    def step(self):
        dt = C.c_double()
        self._chk(self.L.msom_step(self.h, C.byref(dt)))
        return dt.value

    def set_tnext(self, tnext):
        self._chk(self.L.msom_set_tnext(self.h, tnext))

    @property
    def t(self):
        return self.L.msom_time(self.h)

    @property
    def iter(self):
        return self.L.msom_iter(self.h)

    def ke(self):
        return self.L.msom_ke(self.h)

    def mgstats(self):
        st = MGStats()
        self._chk(self.L.msom_last_mgstats(self.h, C.byref(st)))
        return st

    def run(self, workdir=".", nsteps_max=-1):
        self._chk(self.L.msom_run(self.h, workdir.encode(), nsteps_max))

    # wavelet scale filter (msqg/qg.h:509-560)
    def wavelet_filter(self, dtflt):
        self._chk(self.L.msom_wavelet_filter(self.h, dtflt))

    # energy / PV budgets (msqg/qg_energy.h)
    def energy_tend(self, dt):
        self._chk(self.L.msom_energy_tend(self.h, dt))

    def filter_de(self, pm_field, dtflt):
        self._chk(self.L.msom_filter_de(self.h, pm_field, dtflt))

    def reset_de(self):
        self._chk(self.L.msom_reset_de(self.h))

    def pystep_de(self, po, bf, vd, j1, j2, j3, ft, onlyKE=0):
        """bas.pystep_de(p, bf, vd, j1, j2, j3, ft, flag_keonly) of msqg/scripts/energy_offline.py:113"""
        arrs = [_f64(po, (self.nl, self.ny, self.nx))] + [a for a in (bf, vd, j1, j2, j3, ft)]
        args = []
        for a in arrs:
            args += [_ptr(a), self.nl, self.ny, self.nx]
        self._chk(self.L.pystep_de(self.h, *args, int(onlyKE)))

    def wavelet_levels(self):
        n = self.L.msom_dbg_wavelet_levels(self.h)
        if n < 0:
            self._chk(n)
        return n

    def siglev(self, level):
        a = np.empty((1, self.ny >> level, self.nx >> level))
        self._chk(self.L.msom_dbg_siglev(self.h, level, _ptr(a)))
        return a

    def wavelet_apply(self, field):
        self._chk(self.L.msom_dbg_wavelet_apply(self.h, field))

    def write_bas(self, field, path):
        self._chk(self.L.msom_write_bas(self.h, field, path.encode()))

    def read_bas(self, field, path):
        self._chk(self.L.msom_read_bas(self.h, field, path.encode()))

    def write_nc(self, path):
        self._chk(self.L.msom_write_nc(self.h, path.encode()))

    def read_nc(self, field, path, varname, record=-1):
        self._chk(self.L.msom_read_nc(self.h, field, path.encode(), varname.encode(), record))

    # -- debug hooks
    def nlevels(self):
        return self.L.msom_dbg_nlevels(self.h)

    def level_dims(self, lev):
        nx, ny = C.c_int(), C.c_int()
        self._chk(self.L.msom_dbg_level_dims(self.h, lev, C.byref(nx), C.byref(ny)))
        return nx.value, ny.value

    def relax(self, lev, da, res, nsweeps=1):
        da = np.array(da, dtype=np.float64, order="C")
        res = _f64(res)
        self._chk(self.L.msom_dbg_relax(self.h, lev, _ptr(da), _ptr(res), nsweeps))
        return da

    def residual(self, a, b):
        a, b = _f64(a), _f64(b)
        res = np.empty_like(a)
        m = C.c_double()
        self._chk(self.L.msom_dbg_residual(self.h, _ptr(a), _ptr(b), _ptr(res), C.byref(m)))
        return res, m.value

    def restrict(self, lev_fine, fine):
        fine = _f64(fine)
        nl, ny, nx = fine.shape
        coarse = np.empty((nl, ny // 2, nx // 2))
        self._chk(self.L.msom_dbg_restrict(self.h, lev_fine, _ptr(fine), _ptr(coarse)))
        return coarse

    def prolong(self, lev_coarse, coarse):
        coarse = _f64(coarse)
        nl, ny, nx = coarse.shape
        fine = np.empty((nl, ny * 2, nx * 2))
        self._chk(self.L.msom_dbg_prolong(self.h, lev_coarse, _ptr(coarse), _ptr(fine)))
        return fine

    def op(self, name, f_in, f_out, add=0.0, fac=1.0):
        self._chk(self.L.msom_dbg_op(self.h, name.encode(), f_in, f_out, add, fac))

    # -- measurement
    def profile_read(self, kernel):
        ms, n = C.c_double(), C.c_long()
        self._chk(self.L.msom_profile_read(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        self._chk(self.L.msom_profile_reset(self.h))

    def sync(self):
        """wait for the work queued on the handle's stream (msom_step may return before its last kernel has finished)"""
        self._chk(self.L.msom_sync(self.h))

    def bench_kernel(self, kernel, reps=20):
        ms = C.c_double()
        self._chk(self.L.msom_bench_kernel(self.h, kernel.encode(), reps, C.byref(ms)))
        return ms.value


# ---------------------------------------------------------------------------
# module-level mirror of the reference's single-instance `import qg` surface
# (msqg/qg_bfn.py:33-37: read_params -> init_grid -> set_vars -> set_vars_bfn -> set_const)

_state = {"path": None, "model": None}


def read_params(path2file):
    _state["path"] = path2file


def init_grid(N):  # grid size comes from params.in; kept for call-compatibility
    return None


def set_vars():
    if _state["path"] is None:
        raise MsomError("read_params() must be called first")
    _state["model"] = QG(path=_state["path"])
    d = os.path.dirname(os.path.abspath(_state["path"]))
    _state["model"].option("quiet", 1)
    _state["model"].read_inputs(d)


def set_vars_bfn():
    return None


def set_const():
    _state["model"].set_const()


def pystep_bfn(var, tend, direction, vartype):
    _state["model"].pystep_bfn(var, tend, direction, vartype)


def pyq2p(p, q):
    _state["model"].pyq2p(p, q)


def pyp2q(p, q):
    _state["model"].pyp2q(p, q)


def pystep_de(p, bf, vd, j1, j2, j3, ft, onlyKE=0):
    """msqg/qg_energy.i:31, called as bas.pystep_de(p,bf,vd,j1,j2,j3,ft,flag_keonly) (msqg/scripts/energy_offline.py:113)"""
    _state["model"].pystep_de(p, bf, vd, j1, j2, j3, ft, onlyKE)


def trash_vars():
    if _state["model"] is not None:
        _state["model"].close()
        _state["model"] = None


def trash_vars_bfn():
    return None


NODE_FIELDS = dict(PSI=0, Q=1, ZETA=2, TMP=3, PSIPG=4, S2=5, TOPO=6, QFORC=7, MASK=8, DQ=9, QPRED=10, QFORC3D=11, BS=12, S2S=13, PSIF=14)


class NodeQG:
    """Vertex-grid (masked-domain) model: one `qg.e` process of qg-node/qg.c.  Field arrays are
    [layers][N+1][N+1] (qg-node/netcdf_vertex_bas.h:253)."""

    def __init__(self, params=None, path=None, strict=False):
        self.L = load_library(strict)
        self.h = self.L.msomn_create(path.encode()) if path is not None else self.L.msomn_create_str(params.encode())
        if not self.h:
            raise MsomError(self.L.msom_last_error().decode())
        self.N, self.nl = int(self.param("N")), int(self.param("nl"))
        self.nlevels = int(self.param("nlevels"))

    def close(self):
        if self.h:
            self.L.msomn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise MsomError(f"error {r}: {self.L.msom_last_error().decode()}")

    def _fid(self, f):
        return NODE_FIELDS[f] if isinstance(f, str) else int(f)

    def param(self, key):
        return self.L.msomn_get_param(self.h, key.encode())

    def set_option(self, key, v):
        self._chk(self.L.msomn_set_option(self.h, key.encode(), float(v)))

    def profile_read(self, slot):
        ms, n = C.c_double(), C.c_long()
        self._chk(self.L.msomn_profile_read(self.h, slot.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        self._chk(self.L.msomn_profile_reset(self.h))

    def layers(self, f):
        return self.L.msomn_field_layers(self.h, self._fid(f))

    def shape(self, f):
        return (self.layers(f), self.N + 1, self.N + 1)

    def set(self, f, a):
        a = _f64(a, self.shape(f))
        self._chk(self.L.msomn_set_field(self.h, self._fid(f), _ptr(a)))

    def get(self, f):
        a = np.empty(self.shape(f))
        self._chk(self.L.msomn_get_field(self.h, self._fid(f), _ptr(a)))
        return a

    def set_const(self):
        self._chk(self.L.msomn_set_const(self.h))

    def update(self, q="Q", dq="DQ", dtmax=None):
        dt = C.c_double()
        self._chk(self.L.msomn_update(self.h, self._fid(q), self._fid(dq), self.param("DT") if dtmax is None else dtmax, C.byref(dt)))
        return dt.value

    def advance(self, out, inp, dq, dt):
        self._chk(self.L.msomn_advance(self.h, self._fid(out), self._fid(inp), self._fid(dq), dt))

    def invert_q(self, q="Q"):
        st = MGStats()
        self._chk(self.L.msomn_invert_q(self.h, self._fid(q), C.byref(st)))
        return st

    def comp_q(self, psi="PSI", q="Q"):
        self._chk(self.L.msomn_comp_q(self.h, self._fid(psi), self._fid(q)))

    def rhs_pv(self, q="Q", dq="DQ"):
        self._chk(self.L.msomn_rhs_pv(self.h, self._fid(q), self._fid(dq)))

    def forcing(self):
        self._chk(self.L.msomn_forcing(self.h))

    def step(self, with_forcing_event=False):
        self._chk(self.L.msomn_step(self.h, int(with_forcing_event)))

    def set_tnext(self, t):
        self._chk(self.L.msomn_set_tnext(self.h, t))

    @property
    def t(self):
        return self.L.msomn_time(self.h)

    @property
    def dt(self):
        return self.L.msomn_dt(self.h)

    @property
    def iter(self):
        return self.L.msomn_iter(self.h)

    def ke(self):
        v = C.c_double()
        self._chk(self.L.msomn_ke(self.h, C.byref(v)))
        return v.value

    def mgstats(self):
        st = MGStats()
        self._chk(self.L.msomn_last_mgstats(self.h, C.byref(st)))
        return st

    def diag1d(self):
        out = (C.c_double * 3)()
        self._chk(self.L.msomn_diag1d(self.h, out))
        return np.array(out[:])

    def write_nc(self, path):
        self._chk(self.L.msomn_write_nc(self.h, path.encode()))

    def read_nc(self, f, path, var, record=-1):
        self._chk(self.L.msomn_read_nc(self.h, self._fid(f), path.encode(), var.encode(), record))

    def run(self, workdir=".", nsteps_max=-1):
        r = self.L.msomn_run(self.h, workdir.encode(), nsteps_max)
        if r < 0:
            self._chk(r)
        return r

    # multigrid pieces (parity tests); level k has (N >> k) + 1 vertices per side
    def _lshape(self, k, nl=None):
        n1 = (self.N >> k) + 1
        return (self.nl if nl is None else nl, n1, n1)

    def dbg_relax(self, k, da, res, nsweeps):
        da = np.array(_f64(da, self._lshape(k)))
        res = _f64(res, self._lshape(k))
        self._chk(self.L.msomn_dbg_relax(self.h, k, _ptr(da), _ptr(res), nsweeps))
        return da

    def dbg_residual(self, a, b):
        a, b = _f64(a, self._lshape(0)), _f64(b, self._lshape(0))
        res = np.empty(self._lshape(0))
        mx = C.c_double()
        self._chk(self.L.msomn_dbg_residual(self.h, _ptr(a), _ptr(b), _ptr(res), C.byref(mx)))
        return res, mx.value

    def dbg_restrict(self, k, fine):
        fine = _f64(fine, self._lshape(k))
        out = np.empty(self._lshape(k + 1))
        self._chk(self.L.msomn_dbg_restrict(self.h, k, _ptr(fine), _ptr(out)))
        return out

    def dbg_prolong(self, k, coarse):
        coarse = _f64(coarse, self._lshape(k))
        out = np.empty(self._lshape(k - 1))
        self._chk(self.L.msomn_dbg_prolong(self.h, k, _ptr(coarse), _ptr(out)))
        return out

    def dbg_level_mask(self, k):
        out = np.empty(self._lshape(k, 1))
        self._chk(self.L.msomn_dbg_level_mask(self.h, k, _ptr(out)))
        return out

    def dbg_del2_zeta(self):
        self._chk(self.L.msomn_dbg_del2_zeta(self.h))

    # stochastic forcing (cell scalars)
    def noise(self, set=None, filter=False):
        out = np.empty((self.N, self.N))
        a = None if set is None else _f64(set, (self.N, self.N))
        self._chk(self.L.msomn_dbg_noise(self.h, _ptr(a), int(filter), _ptr(out)))
        return out

    def csig(self, k):
        n = self.N >> k
        out = np.empty((n, n))
        self._chk(self.L.msomn_dbg_csig(self.h, k, _ptr(out)))
        return out

    # wavelet filter of the vertex model (qg_baroclinic_ms.h:346-400)
    def wavelet_filter(self, dtflt):
        self._chk(self.L.msomn_wavelet_filter(self.h, dtflt))

    def wv_get(self, what, k):
        n = self.N >> k
        out = np.empty((n, n))
        self._chk(self.L.msomn_dbg_wv_get(self.h, what, k, _ptr(out)))
        return out

    def wv_apply(self, cells):
        cells = _f64(cells, (self.nl, self.N, self.N))
        out = np.empty_like(cells)
        self._chk(self.L.msomn_dbg_wv_apply(self.h, _ptr(cells), _ptr(out)))
        return out
