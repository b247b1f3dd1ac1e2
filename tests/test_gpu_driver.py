"""The C host driver msom_qg (counterpart of msqg/qg.c main) on the GPU box: same params.in,
same p0.bas restart file, same outdir_%04d/po%09d.bas / qo%09d.bas outputs, same per-step
stdout line (msqg/qg.c:101-122), checked against the oracle driven through the same event
schedule (Basilisk run(): events first, then one predictor-corrector step)."""
import os
import re
import subprocess

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "msom_amd", "lib", "msom_qg")


def read_bas(path, nl, n):
    return np.fromfile(path, "f4").reshape(nl, n + 1, n + 1).transpose(0, 2, 1)[:, 1:, 1:]


def test_msom_qg_reproduces_reference_driver_outputs(tmp_path):
    N, nl = 32, 3
    txt = orc.double_gyre_params(N, nl).replace("tend  = 500.", "tend = 0.06").replace("dtout = 1.", "dtout = 0.02")
    (tmp_path / "params.in").write_text(txt)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N) + 2e-4)       # non-zero mean: exercises the mean removal
    assert o.write_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    res = subprocess.run([EXE, "params.in"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    out = res.stdout
    assert f"Config: N = {N}, nl = {nl}, L0 = 80" in out
    assert "p0.bas .. ok" in out and "Backup config" in out and "Writing output in" in out

    # the same schedule on the oracle
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
    assert o.read_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    o.remove_mean(orc.PSI)
    o.set_const()
    tend, dtout, tout = o.param("tend"), o.param("dtout"), 0.0
    lines, outputs = [], {}
    while True:
        lines.append((o.iter, o.dt, o.t, o.ke()))
        pending = tout <= tend + 1e-10
        if pending and o.t >= tout - 1e-12 * max(1.0, abs(tout)):
            o.invertq()
            outputs[o.iter] = (o.get(orc.PSI), o.get(orc.Q))
            tout += dtout
            pending = tout <= tend + 1e-10
        if not pending:
            break
        o.set_tnext(tout)
        o.step()

    got = re.findall(r"i = (\d+), dt = (\S+), t = (\S+), ke_1 = (\S+)", out)
    assert len(got) == len(lines) and out.count("write file") == len(outputs) == 4
    for (i, dt, t, ke), (oi, odt, ot, oke) in zip(got, lines):
        assert int(i) == oi
        assert float(dt) == pytest.approx(odt, rel=2e-5) and float(t) == pytest.approx(ot, rel=2e-5, abs=1e-12)
        assert float(ke) == pytest.approx(oke, rel=2e-5)
    od = tmp_path / "outdir_0001"
    for it, (p, q) in outputs.items():
        for name, ref in ((f"po{it:09d}.bas", p), (f"qo{it:09d}.bas", q)):
            a = read_bas(od / name, nl, N)
            assert np.allclose(a, ref.astype("f4"), rtol=2e-6, atol=1e-30), name
    # backup_config, msqg/qg.h:782-835
    assert (od / "params.in").read_text() == txt
    for f in (f"psipg_{nl}l_N{N}.bas", f"frpg_{nl}l_N{N}.bas", f"qforc_{nl}l_N{N}.bas", f"rdpg_{nl}l_N{N}.bas", "sig_filt.bas", f"dh_{nl}l.bin"):
        assert (od / f).exists(), f
    fr = read_bas(od / f"frpg_{nl}l_N{N}.bas", nl, N)
    assert np.allclose(fr[0], 0.0023669) and np.allclose(fr[1], 0.0076173) and np.all(fr[2] == 0)
    assert np.allclose(np.fromfile(od / f"dh_{nl}l.bin", "f4"), [0.06, 0.14, 0.8])
    # a second run picks the next free output directory (create_outdir, msqg/qg.h:766-776)
    res = subprocess.run([EXE, "params.in", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and (tmp_path / "outdir_0002").is_dir()


def test_msom_qg_missing_params_file(tmp_path):
    res = subprocess.run([EXE, "nope.in"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert res.returncode != 0 and "file nope.in not found" in res.stdout   # reference message, msqg/qg.h:736
