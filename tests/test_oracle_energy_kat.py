"""Known-answer tests of the oracle's energy / PV budgets (msqg/qg_energy.h): the PV budget (ediag = 1)
closes against the tendency operators term by term, the energy budget (ediag = 0) of the Jacobian terms
vanishes in a periodic box (Arakawa), bookkeeping of po_mft / filter_de / pystep_de."""
import numpy as np

import orc

DH = {3: [0.06, 0.14, 0.8], 2: [0.2, 0.8]}


def make(N, nl, extra, psi=None, pg=False, **opt):
    o = orc.Oracle(orc.double_gyre_params(N, nl, extra=extra), smoother=orc.GS_RB, quiet=1, **opt)
    o.set(orc.PSI, orc.synthetic_psi(nl, N, N) if psi is None else psi)
    if pg:
        o.set(orc.PSIPG, 0.3 * orc.synthetic_psi(nl, N, N)[::-1].copy())
    o.set_const()
    return o


def test_pv_budget_closes_against_the_tendency():
    N, nl, dt = 32, 3, 0.37
    o = make(N, nl, "ediag = 1\ntau0 = 0\nRe = 800\nEks = 0.003\nflsrv = 1\n", pg=True)
    o.energy_tend(dt)
    de = {k: o.get(getattr(orc, k)) for k in ("DE_BF", "DE_VD", "DE_J1", "DE_J2", "DE_J3")}
    # the same terms through the model's own operators
    o.L.orc_comp_del2(o.h, orc.PSI, orc.ZETA, 0.0, 1.0)
    zero = np.zeros((nl, N, N))
    o.set(orc.DQ, zero); o.L.orc_advection_pv(o.h, orc.ZETA, orc.Q, orc.PSI, orc.DQ, 1.0); adv = o.get(orc.DQ)
    o.set(orc.DQ, zero); o.L.orc_dissip(o.h, orc.ZETA, orc.DQ); dis = o.get(orc.DQ)
    o.set(orc.DQ, zero); o.L.orc_forcing_terms(o.h, orc.ZETA, orc.PSI, orc.DQ); frc = o.get(orc.DQ)   # tau0 = 0: Ekman only
    scale = np.abs(adv).max()
    assert np.abs((de["DE_J1"] + de["DE_J2"] + de["DE_J3"]) / dt - adv).max() <= 1e-12 * scale
    assert np.abs(de["DE_VD"] / dt - dis).max() <= 1e-12 * np.abs(dis).max()
    assert np.abs(de["DE_BF"] / dt - frc).max() <= 1e-12 * np.abs(frc).max()
    assert all(np.abs(v).max() > 0 for v in de.values())
    assert np.array_equal(o.get(orc.PO_MFT), o.get(orc.PSI))
    o.energy_tend(dt)                                   # budgets accumulate, the psi mean stays psi
    assert np.abs(o.get(orc.DE_VD) - 2 * de["DE_VD"]).max() <= 1e-15 * np.abs(de["DE_VD"]).max() * 4
    assert np.abs(o.get(orc.PO_MFT) - o.get(orc.PSI)).max() <= 1e-18


def test_energy_budget_of_the_jacobians_vanishes_in_a_periodic_box():
    N, nl = 32, 3
    rng = np.random.default_rng(4)
    psi = np.zeros((nl, N, N))
    x = np.arange(N) / N
    for l in range(nl):
        for k in range(1, 4):
            for m in range(1, 4):
                psi[l] += rng.standard_normal() * np.outer(np.sin(2 * np.pi * (m * x + rng.random())), np.cos(2 * np.pi * (k * x + rng.random())))
    o = make(N, nl, "ediag = 0\nsbc = -1\nbeta = 0\n", psi=1e-2 * psi)
    o.energy_tend(1.0)
    j1 = o.get(orc.DE_J1)
    dh = np.array(DH[nl])[:, None, None]
    assert abs((dh * j1).sum()) <= 1e-12 * np.abs(dh * j1).sum()
    assert np.abs(j1).max() > 0


def test_filter_de_and_pystep_de():
    N, nl, dtflt = 32, 2, 0.25
    o = make(N, nl, f"ediag = 1\nafilt = 4\ndtflt = {dtflt}\n", TOLERANCE=1e-11)
    q0 = o.get(orc.Q)
    o.energy_tend(1.0)
    o.filter_de(orc.PO_MFT, dtflt)
    ref = make(N, nl, f"ediag = 1\nafilt = 4\ndtflt = {dtflt}\n", TOLERANCE=1e-11)
    ref.wavelet_filter(dtflt)
    large = q0 - ref.get(orc.Q)                         # what the filter removes
    # de_ft = tmp2 * dtflt * 1 with tmp2 = (q0 - q_filtered) / (-dtflt)
    assert np.abs(o.get(orc.DE_FT) + large).max() <= 1e-9 * np.abs(large).max()
    assert np.array_equal(o.get(orc.Q), q0)            # negative dtflt restores q
    assert np.all(o.get(orc.PO_MFT) == 0)
    # pystep_de: ediag = 1, dt = 1, budgets of the psi passed in; psi itself is zeroed by filter_de(po_mft = pol)
    o2 = make(N, nl, f"afilt = 4\ndtflt = {dtflt}\nRe = 800\n", TOLERANCE=1e-11)
    psi = orc.synthetic_psi(nl, N, N) * 1.7
    bf, vd, j1, j2, j3, ft = o2.pystep_de(psi)
    o3 = make(N, nl, f"ediag = 1\nafilt = 4\ndtflt = {dtflt}\nRe = 800\n", psi=psi, TOLERANCE=1e-11)
    o3.energy_tend(1.0)
    assert np.array_equal(j1, o3.get(orc.DE_J1)) and np.array_equal(vd, o3.get(orc.DE_VD)) and np.array_equal(bf, o3.get(orc.DE_BF))
    assert np.all(j2 == 0) and np.abs(ft).max() > 0 and np.all(o2.get(orc.PSI) == 0)
    ke = o2.pystep_de(psi, onlyKE=1)
    assert np.all(o2.get(orc.S) == 0) and np.abs(ke[2] - j1).max() > 0
