"""The CPU oracle against the committed golden files (tests/golden/*.npz, written by tools/make_golden.py): fields bit for
bit, grid sums (KE line, mgstats.sum, row sums) to 1e-12 relative because the OpenMP team size changes their summation
order.  This is what keeps the oracle from drifting between rounds; the reference itself has no vectors for this path
(PARITY UNPINNED).  Also pins the .bas restart file against the layout msqg/scripts/read_data.py reads."""
import numpy as np
import pytest

import golden_cases as gc


@pytest.mark.parametrize("name", list(gc.CASES))
def test_oracle_reproduces_golden(name):
    got, exp = gc.run_case(name, gc.OracleModel)
    gc.compare(got, exp, exact=True)


@pytest.mark.parametrize("name", list(gc.LEX_CASES))
def test_oracle_lexicographic_reproduces_golden(name):
    """reference sweep order (msqg/poisson_layer.h:75-149 in place, x outer / y inner), single-thread order"""
    fn, kw, _ = gc.LEX_CASES[name]
    kw = dict(kw)
    sm = kw.pop("smoother")
    got = fn(lambda txt, **o: gc.OracleModel(txt, smoother=sm, **o), {}, **kw)
    gc.compare(got, gc.load(name), exact=True)


@pytest.mark.parametrize("name", list(gc.NODE_CASES))
def test_node_oracle_reproduces_golden(name):
    got, exp = gc.run_case(name, gc.NodeOracleModel, gc.NODE_CASES)
    gc.compare(got, exp, exact=True)


def test_lexicographic_and_red_black_runs_differ_by_the_solver_tolerance_only():
    """the same 10 steps with the reference's lexicographic Gauss-Seidel (msqg/poisson_layer.h:75-149) and with the
    red-black order the GPU uses.  Each solve stops with max|res| = resa (mgstats), and an iterate with residual r is
    within |A^-1|_inf r <= 0.0737 L0^2 r of the discrete solution, so after the FIRST step the two psi differ by at most
    (resa_rb + resa_lex) 0.0737 L0^2 -- checked at TOLERANCE 1e-12 and at the reference's 1e-3 (msqg/qg.h:159).
    Measured on the files (relative to max|psi|): 8.7e-8 / 2.8e-9 after 1 / 10 steps at 1e-12, 6.8e-3 / 2.5e-3 at 1e-3
    (the reference's own tolerance leaves psi uncertain to that level whatever the sweep order)."""
    for tag in ("_tol1e-12", ""):
        rb, lex = gc.load("run_p0bas_32x32x3" + tag), gc.load("run_p0bas_32x32x3_lexicographic" + tag)
        resa = rb["mgstats"][0, 2] + lex["mgstats"][0, 2]
        d1 = np.abs(rb["psi_1"] - lex["psi_1"]).max()
        assert d1 <= resa * 0.0737 * 80.0 ** 2, (tag, d1, resa)
        d10 = np.abs(rb["psi_10"] - lex["psi_10"]).max() / np.abs(lex["psi_10"]).max()
        assert d10 <= (1e-8 if tag else 5e-3), (tag, d10)      # measured 2.8e-9 / 2.5e-3
        assert np.array_equal(rb["mgstats"][:, 0], lex["mgstats"][:, 0])   # same number of cycles per solve


def test_p0bas_layout_is_what_read_data_py_reads():
    """msqg/scripts/read_data.py:44-46: float32, (N+1)^2 frame per layer, [layer][x][y] -> psi[l, 1:, 1:].T"""
    N, nl = 32, 3
    raw = np.fromfile(gc.P0BAS, dtype=np.float32).reshape(nl, N + 1, N + 1)
    assert raw[0, 0, 0] == N
    psi = raw[:, 1:, 1:].transpose(0, 2, 1).astype(np.float64)
    want = gc.wl.synthetic_psi(nl, N, N).astype(np.float32).astype(np.float64)
    assert np.array_equal(psi, want)
    assert np.array_equal(gc.load("run_p0bas_32x32x3")["psi_0"], want)
