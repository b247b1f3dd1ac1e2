"""GPU parity of the energy / PV budgets (msom_energy_tend, msom_filter_de, pystep_de; msqg/qg_energy.h)
against the oracle (pinned by tests/test_oracle_energy_kat.py).  strict build: bit-exact."""
import os
import subprocess

import numpy as np
import pytest

import orc
from msom_amd import FIELDS as F
from msom_amd import QG

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "msom_amd", "lib", "msom_qg")
DE = ("DE_BF", "DE_VD", "DE_J1", "DE_J2", "DE_J3", "DE_FT", "PO_MFT")


def pair(N, nl, extra, strict, pg=False, tol=1e-11):
    txt = orc.double_gyre_params(N, nl, extra=extra)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1, TOLERANCE=tol)
    g = QG(txt, strict=strict)
    g.option("quiet", 1); g.option("TOLERANCE", tol)
    psi = orc.synthetic_psi(nl, N, N)
    if pg:
        pgf = 0.3 * psi[::-1].copy()
        o.set(orc.PSIPG, pgf); g.set(F["PSIPG"], pgf)
    o.set(orc.PSI, psi); g.set(F["PSI"], psi)
    o.set_const(); g.set_const()
    return o, g


def same(a, b, strict, rtol, name=""):
    if strict:
        assert np.array_equal(a, b), f"{name}: max diff {np.abs(a - b).max():g}"
    else:
        assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-300), name


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nl,ediag,pg,extra", [(3, 0, False, ""), (3, 1, True, "Re = 800\nEks = 0.003\nflsrv = 1\n"), (2, 0, True, "Re = 500\n"),
                                               (1, 1, False, ""), (6, 0, False, "Eks = 0.001\n")])
def test_energy_tend_through_time_steps(nl, ediag, pg, extra, strict):
    o, g = pair(32, nl, f"ediag = {ediag}\n" + extra, strict, pg=pg, tol=1e-9)
    for it in range(3):
        o.energy_tend(o.dt); g.energy_tend(o.dt)      # event comp_diag, then the step
        o.step(); g.step()
    for k in DE:
        same(g.get(F[k]), o.get(getattr(orc, k)), strict, 1e-7, k)
    assert np.abs(g.get(F["DE_J1"])).max() > 0 or nl == 1
    g.reset_de(); o.reset_de()
    assert all(np.all(g.get(F[k]) == 0) for k in DE[:6])
    same(g.get(F["PO_MFT"]), o.get(orc.PO_MFT), strict, 1e-7)


@pytest.mark.parametrize("strict", [True, False])
def test_filter_de(strict):
    o, g = pair(64, 3, "ediag = 0\nafilt = 4\ndtflt = 0.25\n", strict)
    q0 = g.get(F["Q"])
    for m_ in (o, g):
        m_.energy_tend(0.5)
    o.filter_de(orc.PO_MFT, 0.25); g.filter_de(F["PO_MFT"], 0.25)
    for k in DE:
        same(g.get(F[k]), o.get(getattr(orc, k)), strict, 1e-8, k)
    assert np.array_equal(g.get(F["Q"]), q0) and np.all(g.get(F["PO_MFT"]) == 0)
    assert np.abs(g.get(F["DE_FT"])).max() > 0


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("onlyKE", [0, 1])
def test_pystep_de(onlyKE, strict):
    N, nl = 32, 3
    o, g = pair(N, nl, "afilt = 4\ndtflt = 0.25\nRe = 800\n", strict)
    psi = 1.3 * orc.synthetic_psi(nl, N, N)
    ref = o.pystep_de(psi, onlyKE)
    outs = [np.empty((nl, N, N)) for _ in range(6)]
    g.pystep_de(psi, *outs, onlyKE)
    for name, a, b in zip(DE, outs, ref):
        same(a, b, strict, 1e-8, name)
    assert np.all(g.get(F["PSI"]) == 0)
    if onlyKE:
        assert np.all(g.get(F["S"]) == 0)
        g.set_const()
        assert np.abs(g.get(F["S"])).max() > 0     # set_const restores S = (Fr/Ro)^2
    # module-level mirror used by msqg/scripts/energy_offline.py
    import msom_amd as bas
    assert callable(bas.pystep_de)


def read_bas(path, nl, n):
    return np.fromfile(path, "f4").reshape(nl, n + 1, n + 1).transpose(0, 2, 1)[:, 1:, 1:]


def test_driver_writes_budgets(tmp_path):
    """msom_qg with ediag = 0 and dtflt > 0: comp_diag every iteration, filter_de before the filter, de_*%09d.bas
    = budget / dtout at every output, then reset (msqg/qg.c:131-160)"""
    N, nl = 32, 3
    txt = orc.double_gyre_params(N, nl, extra="ediag = 0\nafilt = 4\ndtflt = 0.03\nTOLERANCE = 1e-10\n").replace("tend  = 500.", "tend = 0.06").replace("dtout = 1.", "dtout = 0.02")
    (tmp_path / "params.in").write_text(txt)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1, TOLERANCE=1e-10)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    assert o.write_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    res = subprocess.run([EXE, "params.in"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1, TOLERANCE=1e-10)
    assert o.read_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    o.remove_mean(orc.PSI)
    o.set_const()
    tend, dtout, dtflt, tout, tflt = 0.06, 0.02, 0.03, 0.0, 0.03
    budgets = {}
    while True:
        if tflt <= tend + 1e-10 and o.t >= tflt - 1e-12:
            o.filter_de(orc.PO_MFT, dtflt); o.wavelet_filter(dtflt); tflt += dtflt
        o.energy_tend(o.dt)
        pending = tout <= tend + 1e-10
        if pending and o.t >= tout - 1e-12 * max(1.0, abs(tout)):
            o.invertq()
            budgets[o.iter] = {k: o.get(getattr(orc, k)) / dtout for k in DE[:6]}
            o.reset_de()
            tout += dtout
            pending = tout <= tend + 1e-10
        if not pending:
            break
        o.set_tnext(min(tout, tflt) if tflt <= tend + 1e-10 else tout)
        o.step()
    od = tmp_path / "outdir_0001"
    assert len(budgets) == 4
    for it, b in budgets.items():
        for k, ref in b.items():
            a = read_bas(od / f"{k.lower()}{it:09d}.bas", nl, N)
            scale = max(np.abs(v[k]).max() for v in budgets.values())
            assert np.allclose(a, ref.astype("f4"), rtol=1e-4, atol=1e-6 * scale), (k, it)
    assert max(np.abs(b["DE_FT"]).max() for b in budgets.values()) > 0
