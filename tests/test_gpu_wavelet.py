"""GPU parity of the wavelet scale filter (msom_wavelet_filter, kernels_wavelet.hip) against the
oracle (orc_wavelet_filter; itself pinned by tests/test_oracle_wavelet_kat.py).
strict build: bit-exact; product build: relative tolerance stated per check."""
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from msom_amd import FIELDS as F
from msom_amd import QG

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "msom_amd", "lib", "msom_qg")


def pair(N, nl, extra, strict, Rd=None, psi=None, tol=1e-11):
    txt = orc.double_gyre_params(N, nl, extra=extra)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1, TOLERANCE=tol)
    g = QG(txt, strict=strict)
    g.option("quiet", 1); g.option("TOLERANCE", tol)
    if Rd is not None:
        o.set(orc.RD, Rd); g.set(F["RD"], Rd)
    psi = orc.synthetic_psi(nl, N, N) if psi is None else psi
    o.set(orc.PSI, psi); g.set(F["PSI"], psi)
    o.set_const(); g.set_const()
    return o, g


def same(a, b, strict, rtol):
    if strict:
        assert np.array_equal(a, b), f"max diff {np.abs(a - b).max():g}"
    else:
        assert np.abs(a - b).max() <= rtol * np.abs(b).max()


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("N,nl,afilt", [(32, 1, 7.0), (64, 3, 3.0), (128, 6, 12.0)])
def test_siglev_and_transform(N, nl, afilt, strict):
    rng = np.random.default_rng(N)
    Rd = 0.5 + 2.5 * rng.random((1, N, N))
    psi = rng.standard_normal((nl, N, N))
    o, g = pair(N, nl, f"afilt = {afilt}\n", strict, Rd=Rd, psi=psi)
    K = g.wavelet_levels()
    assert K == o.wavelet_levels() == int(np.log2(N)) + 1
    for k in range(K):
        assert np.array_equal(g.siglev(k), o.siglev(k))
    o.wavelet_apply(orc.PSI); g.wavelet_apply(F["PSI"])
    same(g.get(F["PSI"]), o.get(orc.PSI), strict, 1e-14)


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("dtflt", [0.5, -0.5])
def test_wavelet_filter(dtflt, strict):
    N, nl = 64, 3
    Rd = np.ones((1, N, N)); Rd[0, :, N // 2:] = 3.0
    o, g = pair(N, nl, "afilt = 4\n", strict, Rd=Rd)
    q0 = g.get(F["Q"])
    o.wavelet_filter(dtflt); g.wavelet_filter(dtflt)
    for name, gf, of in (("psi", F["PSI"], orc.PSI), ("q", F["Q"], orc.Q), ("qof", F["QOF"], orc.QOF), ("tmp", F["TMP"], orc.TMP)):
        same(g.get(gf), o.get(of), strict, 1e-8)
    if dtflt < 0:
        assert np.array_equal(g.get(F["Q"]), q0)
    # the model keeps running on the filtered state
    for _ in range(2):
        o.step(); g.step()
    same(g.get(F["Q"]), o.get(orc.Q), strict, 1e-7)


def test_periodic_domain():
    N, nl = 32, 2
    txt = orc.double_gyre_params(N, nl, extra="sbc = -1\nafilt = 4\n")
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
    g = QG(txt, strict=True)
    psi = np.random.default_rng(2).standard_normal((nl, N, N))
    o.set(orc.PSI, psi); g.set(F["PSI"], psi)
    o.set_const(); g.set_const()
    o.wavelet_apply(orc.PSI); g.wavelet_apply(F["PSI"])
    assert np.array_equal(g.get(F["PSI"]), o.get(orc.PSI))


def test_full_size_properties():
    """4096^2 x 3: all-pass coefficients reproduce psi to rounding, all-stop coefficients give 0, the
    filter is linear, and the high pass removes a basin-scale field"""
    N, nl = 4096, 3
    psi = orc.synthetic_psi(nl, N, N)
    g = QG(orc.double_gyre_params(N, nl, extra="afilt = 1000\n"))
    g.set(F["PSI"], psi); g.set_const()
    g.wavelet_apply(F["PSI"])
    assert np.abs(g.get(F["PSI"]) - psi).max() <= 1e-15 * np.abs(psi).max() * 8
    g.close()
    g = QG(orc.double_gyre_params(N, nl, extra="afilt = 1\nLfmax = 1e-3\n"))
    g.set(F["PSI"], psi); g.set_const()
    g.wavelet_apply(F["PSI"])
    assert np.all(g.get(F["PSI"]) == 0)
    g.close()
    g = QG(orc.double_gyre_params(N, nl, extra="afilt = 0.6\n"))     # sig_filt = 0.6: levels 0..4 (Delta < 0.3125) kept
    g.set(F["PSI"], psi); g.set_const()
    sig = [float(g.siglev(k).mean()) for k in range(g.wavelet_levels())]
    assert sig[0] == 1 and sig[-1] == 0 and 0 < sum(sig) < len(sig)
    g.wavelet_apply(F["PSI"]); a = g.get(F["PSI"])
    g.set(F["PSI"], 2.5 * psi); g.wavelet_apply(F["PSI"]); b = g.get(F["PSI"])
    assert np.abs(b - 2.5 * a).max() <= 1e-14 * np.abs(psi).max()     # rounding scales with the unfiltered field
    # psi is made of basin-scale modes: the high pass removes nearly all of it
    assert np.abs(a).max() <= 1e-2 * np.abs(psi).max() and abs(a.mean()) <= 1e-4 * np.abs(psi).max()


def read_bas(path, nl, n):
    return np.fromfile(path, "f4").reshape(nl, n + 1, n + 1).transpose(0, 2, 1)[:, 1:, 1:]


def test_driver_filter_event(tmp_path):
    """msom_qg with dtflt > 0: `filter` events at t = dtflt, 2 dtflt, ... (msqg/qg.h:655-658) before the
    stdout line and the outputs; pf%09d.bas = invertq(qof) (msqg/qg.c:124-129).  The first guess of that
    solve is whatever `tmpl` holds (reference: lap(zeta) left by dissip; here: the filter's saved q, the
    fused tendency kernel never materialises tmp), so the comparison runs at a tight TOLERANCE where the
    first guess does not matter."""
    N, nl = 32, 3
    txt = orc.double_gyre_params(N, nl, extra="afilt = 4\ndtflt = 0.03\nTOLERANCE = 1e-10\n").replace("tend  = 500.", "tend = 0.06").replace("dtout = 1.", "dtout = 0.02")
    (tmp_path / "params.in").write_text(txt)
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N))
    assert o.write_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    res = subprocess.run([EXE, "params.in"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1, TOLERANCE=1e-10)
    assert o.read_bas(orc.PSI, str(tmp_path / "p0.bas")) == 0
    o.remove_mean(orc.PSI)
    o.set_const()
    tend, dtout, dtflt, tout, tflt = 0.06, 0.02, 0.03, 0.0, 0.03
    lines, outputs, nfilt = [], {}, 0
    while True:
        if tflt <= tend + 1e-10 and o.t >= tflt - 1e-12:
            o.wavelet_filter(dtflt); nfilt += 1; tflt += dtflt
        lines.append((o.iter, o.dt, o.t, o.ke()))
        pending = tout <= tend + 1e-10
        if pending and o.t >= tout - 1e-12 * max(1.0, abs(tout)):
            o.invertq()
            o.invertq(orc.TMP, orc.QOF)
            outputs[o.iter] = (o.get(orc.PSI), o.get(orc.Q), o.get(orc.TMP))
            tout += dtout
            pending = tout <= tend + 1e-10
        if not pending:
            break
        o.set_tnext(min(tout, tflt) if tflt <= tend + 1e-10 else tout)
        o.step()
    out = res.stdout
    assert out.count("Filter solution") == nfilt == 2
    got = re.findall(r"i = (\d+), dt = (\S+), t = (\S+), ke_1 = (\S+)", out)
    assert len(got) == len(lines)
    for (i, dt, t, ke), (oi, odt, ot, oke) in zip(got, lines):
        assert int(i) == oi and float(t) == pytest.approx(ot, rel=2e-5, abs=1e-12) and float(ke) == pytest.approx(oke, rel=2e-5)
    od = tmp_path / "outdir_0001"
    scale = [max(np.abs(v[k]).max() for v in outputs.values()) for k in range(3)]
    assert scale[2] > 0.1                     # pf = psi of the filter mean is O(q_large_scale / dtflt)
    for it, (p, q, pf) in outputs.items():
        for k, (name, ref) in enumerate(((f"po{it:09d}.bas", p), (f"qo{it:09d}.bas", q), (f"pf{it:09d}.bas", pf))):
            a = read_bas(od / name, nl, N)
            assert np.allclose(a, ref.astype("f4"), rtol=1e-4, atol=1e-6 * scale[k]), name
