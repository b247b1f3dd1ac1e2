"""CPU-side tests of the product's host logic and C ABI (no GPU, no compute calls):
the shared library loads and exports every symbol include/msom.h declares, the params.in
parser and the .bas IO of the library agree with the oracle's restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc
import msom_amd
from msom_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXARR = 64


class Params(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("N", "Ny", "nl", "ediag", "varRo", "nptr", "flsrv")]
                + [(k, C.c_double) for k in ("L0", "Rom", "Ekb", "Eks", "tau0", "Re", "Re4", "iRe", "iRe4", "sbc", "beta",
                                              "afilt", "Lfmax", "DT", "tend", "dtout", "dtflt", "CFL")]
                + [(k, C.c_double * MAXARR) for k in ("Frm", "dhu", "upg", "vpg", "ptr_r", "ptr_ir", "Pe", "iPe")]
                + [(k, C.c_double) for k in ("tr_stoch", "itr_stoch", "amp_stoch", "tolerance")]
                + [("nitermax", C.c_int), ("nitermin", C.c_int), ("mglevels", C.c_int)])


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "msom.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:msom_|py)[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("strict", [False, True])
def test_library_exports_every_declared_symbol(strict):
    L = api.load_library(strict=strict)
    names = declared_symbols()
    assert len(names) > 35 and "pystep_bfn" in names and "msom_update" in names
    for n in names:
        assert hasattr(L, n), n
    assert b"msomhip" in L.msom_version()


def test_no_device_fails_loudly_without_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(msom_amd.MsomError, match="no HIP device"):
        msom_amd.QG("N = 16\nnl = 2\n")


def parse(text):
    L = api.load_library()
    p = Params()
    L.msom_params_defaults(C.byref(p))
    L.msom_params_parse_text(C.byref(p), text.encode())
    if p.Ny <= 0:
        p.Ny = p.N
    L.msom_params_derive(C.byref(p))
    return p


def test_params_parser_matches_reference_semantics():
    p = parse(orc.double_gyre_params(256, 3))
    assert (p.N, p.nl, p.L0) == (256, 3, 80.0)
    assert p.DT == pytest.approx(0.025) and p.iRe4 == -1 / 1563.0 and p.iRe == 0.0
    assert list(p.Frm[:2]) == [0.0023669, 0.0076173] and list(p.dhu[:3]) == [0.06, 0.14, 0.8]
    assert (p.CFL, p.tend, p.dtout, p.beta, p.Ekb, p.tau0, p.Rom) == (0.6, 500.0, 1.0, 0.5, 0.002, 1e-4, 0.025)
    # blanks anywhere, comments, shebang, unknown keys, second '=' (msqg/qg.h:668-697)
    p = parse("#!sh\n# N = 5\n  N   =  32 \nnl=2\nbogus = 7\nFr = [ 0.1 ]\ndh = [0.5 , 0.5]\nRom = 0.1 = 3\nL0 = 2\n\n")
    assert (p.N, p.Ny, p.nl, p.L0, p.Rom) == (32, 32, 2, 2.0, 0.1)
    assert p.DT == 1e10 and p.Frm[0] == 0.1 and list(p.dhu[:2]) == [0.5, 0.5]
    # defaults of msqg/qg.h:63-106 + Basilisk globals
    p = parse("")
    assert (p.N, p.nl, p.L0, p.CFL, p.DT, p.beta, p.tend, p.dtout, p.dtflt, p.ediag) == (64, 1, 1.0, 0.5, 1e10, 0.5, 1.0, 1.0, -1.0, -1)
    assert (p.tolerance, p.nitermax, p.nitermin) == (1e-3, 100, 1)
    # viscous clamps msqg/qg.h:745-746
    p = parse("N = 64\nL0 = 1\nRe = 100\nRe4 = 1e9\nDT = 1\n")
    D = 1.0 / 64
    dt = 0.5 * min(1.0, D * D * 100 / 4)
    dt = 0.5 * min(dt, D**4 * 1e9 / 32)
    assert p.DT == dt and p.iRe == 0.01 and p.iRe4 == -1e-9


@pytest.mark.parametrize("txt", [orc.double_gyre_params(128, 6), "N = 32\nnl = 2\nRe = 40\nFr=[0.2]\ndh=[0.3,0.7]\nRom=0.05\nL0=7\n"])
def test_params_parser_agrees_with_oracle(txt):
    p = parse(txt)
    o = orc.Oracle(txt)
    for k in ("L0", "DT", "iRe", "iRe4", "CFL", "Rom", "tend", "dtout", "beta", "tau0", "Ekb"):
        assert getattr(p, k) == o.param(k), k
    assert (p.N, p.nl) == (o.nx, o.nl)
    for l in range(p.nl):
        assert p.dhu[l] == o.param(f"dh_{l}")
    for l in range(p.nl - 1):
        assert p.Frm[l] == o.param(f"Fr_{l}")


def test_bas_io_matches_oracle_bytes(tmp_path):
    L = api.load_library()
    L.msom_bas_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
    L.msom_bas_read.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
    N, nl = 16, 3
    a = np.random.default_rng(0).standard_normal((nl, N, N))
    txt = f"N = {N}\nnl = {nl}\nL0 = 80\n"
    o = orc.Oracle(txt)
    o.set(orc.PSI, a)
    po, pl = str(tmp_path / "o.bas"), str(tmp_path / "l.bas")
    assert o.write_bas(orc.PSI, po) == 0
    assert L.msom_bas_write(pl.encode(), a.ctypes.data, nl, N, 80.0) == 0
    assert open(po, "rb").read() == open(pl, "rb").read()
    # the layout the reference's post-processing reads (msqg/scripts/read_data.py:44-46)
    raw = np.fromfile(pl, "f4").reshape(nl, N + 1, N + 1).transpose(0, 2, 1)[:, 1:, 1:]
    assert np.array_equal(raw, a.astype("f4"))
    b = np.empty_like(a)
    assert L.msom_bas_read(pl.encode(), b.ctypes.data, nl, N, 80.0) == 0
    assert np.array_equal(b, a.astype("f4").astype("f8"))
    # resampling onto a finer model grid (auxiliar_input.h:44-56)
    c = np.empty((nl, 2 * N, 2 * N))
    assert L.msom_bas_read(pl.encode(), c.ctypes.data, nl, 2 * N, 80.0) == 0
    o2 = orc.Oracle(f"N = {2 * N}\nnl = {nl}\nL0 = 80\n")
    assert o2.read_bas(orc.PSI, po) == 0
    assert np.array_equal(c, o2.get(orc.PSI))
    assert L.msom_bas_read(b"/nonexistent.bas", b.ctypes.data, nl, N, 80.0) != 0
    assert b"not found" in L.msom_last_error()


def test_netcdf3_writer_is_read_by_scipy(tmp_path):
    """The libnetcdf-free classic writer: dims/vars/coordinates as newqg/netcdf_bas.h:42-134,
    one record per write_nc call (:144-244); checked with an independent reader."""
    from scipy.io import netcdf_file
    L = api.load_library()
    cpp = C.POINTER(C.c_char_p)
    L.msom_nc_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, cpp]
    L.msom_nc_append.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, cpp, C.c_double, C.POINTER(C.c_void_p)]
    L.msom_nc_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
    nl, ny, nx, L0 = 3, 8, 16, 80.0
    names = (C.c_char_p * 2)(b"psi", b"q")
    path = str(tmp_path / "vars.nc").encode()
    assert L.msom_nc_create(path, nl, ny, nx, L0, 2, names) == 0
    rng = np.random.default_rng(0)
    recs = []
    for k in range(3):
        a, b = rng.standard_normal((nl, ny, nx)), rng.standard_normal((nl, ny, nx))
        ptrs = (C.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
        assert L.msom_nc_append(path, nl, ny, nx, 2, names, 0.5 * k, ptrs) == k
        recs.append((a, b))
    f = netcdf_file(path.decode(), "r", mmap=False)
    assert list(f.dimensions) == ["level", "y", "x", "time"] and f.dimensions["time"] is None
    assert (f.dimensions["level"], f.dimensions["y"], f.dimensions["x"]) == (nl, ny, nx)
    assert f.variables["psi"].dimensions == ("time", "level", "y", "x") and f.variables["psi"].data.dtype == np.dtype(">f4")
    D = L0 / nx
    assert np.allclose(f.variables["x"][:], (np.arange(nx) + 0.5) * D) and np.allclose(f.variables["y"][:], (np.arange(ny) + 0.5) * D)
    assert np.allclose(f.variables["time"][:], [0.0, 0.5, 1.0])
    for k, (a, b) in enumerate(recs):
        assert np.array_equal(f.variables["psi"][k], a.astype("f4")) and np.array_equal(f.variables["q"][k], b.astype("f4"))
    f.close()
    # restart path: read_nc by name, last record
    out = np.empty((nl, ny, nx)); t = C.c_double()
    assert L.msom_nc_read(path, b"q", -1, nl, ny, nx, out.ctypes.data, C.byref(t)) == 0
    assert np.array_equal(out, recs[-1][1].astype("f4").astype("f8")) and t.value == 1.0
    assert L.msom_nc_read(path, b"nope", 0, nl, ny, nx, out.ctypes.data, None) != 0
    # a file written by another NetCDF-3 writer (with attributes) is readable too
    p2 = str(tmp_path / "restart.nc")
    g = netcdf_file(p2, "w")
    g.history = "made by scipy"
    for n_, v_ in (("time", None), ("level", nl), ("y", ny), ("x", nx)):
        g.createDimension(n_, v_)
    tv = g.createVariable("time", "f4", ("time",)); tv.units = "s"
    pv = g.createVariable("psi", "f8", ("time", "level", "y", "x")); pv.long_name = "stream function"
    tv[0] = 3.0; pv[0] = recs[0][0]
    g.close()
    assert L.msom_nc_read(p2.encode(), b"psi", 0, nl, ny, nx, out.ctypes.data, C.byref(t)) == 0, L.msom_last_error()
    assert np.array_equal(out, recs[0][0]) and t.value == 3.0
