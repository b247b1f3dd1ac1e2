"""GPU parity of the one-layer-per-wavefront tendency kernel (msom_amd/csrc/kernels_lpw.hip, option rhs_variant = 6):
register windows + whole-wave DPP shifts instead of LDS tiles.  Strict build: BIT-EXACT against the CPU oracle (and
hence against the LDS-tile kernel and the kernel-per-reference-loop chain); product build: fp64 round-off of the
re-associated vertical coupling, tolerance stated at the assertion."""
import numpy as np
import pytest

import orc
from msom_amd import QG, FIELDS as F
from test_gpu_parity import make_pair, rand_field, rel

pytestmark = pytest.mark.gpu

PER = "sbc = -1\n"
SLIP = "sbc = 1.5\nRe = 300\nEks = 0.001\n"


@pytest.mark.parametrize("nx,ny,nl", [(64, 64, 3), (128, 32, 6), (32, 32, 1), (16, 16, 2), (256, 128, 4), (512, 16, 3), (64, 256, 2), (128, 64, 8),
                                      (256, 64, 5), (1024, 32, 7)])
@pytest.mark.parametrize("extra", ["", SLIP, PER, "Re4 = 0\nRe = 500\n"])
def test_lpw_tendency_bit_exact(nx, ny, nl, extra):
    """update_qg: dq of the strict build equals the oracle's bit for bit (several strips and chunks;
    walls, partial slip, periodic; with 3-D forcing)"""
    o, g = make_pair(nx, ny, nl, strict=True, extra=extra, TOLERANCE=1e-9)
    qf = rand_field(31, (nl, ny, nx), 1e-7)
    o.set(orc.QFORC, qf); g.set(F["QFORC"], qf)
    d_o = o.update()
    g.option("rhs_variant", 6)
    dq, d = g.update()
    assert d == d_o
    assert np.array_equal(dq, o.get(orc.DQ))


@pytest.mark.parametrize("rows", [8, 16, 24, 64])
def test_lpw_chunk_heights(rows):
    """the result does not depend on the chunk height (warm-up rows, ragged last chunk, ghost rows inside a chunk)"""
    nx, ny, nl = 256, 128, 3
    o, g = make_pair(nx, ny, nl, strict=True, extra=SLIP, TOLERANCE=1e-9)
    o.update()
    g.option("rhs_variant", 6); g.option("rhs_dbg", rows << 8)
    dq, _ = g.update()
    g.option("rhs_dbg", 0)
    assert np.array_equal(dq, o.get(orc.DQ))


@pytest.mark.parametrize("nx,ny,nl,extra", [(64, 64, 3, ""), (128, 64, 6, SLIP), (32, 32, 1, ""), (64, 32, 2, PER + "tau0 = 0\n"), (256, 64, 4, "Eks = 0.01\n")])
def test_lpw_three_steps_bit_exact(nx, ny, nl, extra):
    """RK2 steps with the corrector's advance fused in the pass (q_out = q_in + dt dq)"""
    o, g = make_pair(nx, ny, nl, strict=True, extra=extra, TOLERANCE=1e-8)
    g.option("rhs_variant", 6)
    for _ in range(3):
        o.step(); g.step()
    assert g.t == o.t
    assert np.array_equal(g.get(F["PSI"]), o.get(orc.PSI))
    assert np.array_equal(g.get(F["Q"]), o.get(orc.Q))


def test_lpw_general_S_field():
    """non-uniform Froude field: S read per cell in the finalisation"""
    nx, ny, nl = 128, 64, 4
    o, g = make_pair(nx, ny, nl, strict=True, TOLERANCE=1e-8)
    x = (np.arange(nx) + 0.5) / nx
    fr = np.stack([o.param(f"Fr_{l}") * (1 + 0.3 * np.sin(2 * np.pi * (l + 1) * x))[None, :] * np.ones((ny, 1)) for l in range(nl - 1)])
    o.set(orc.FR, fr); g.set(F["FR"], fr)
    o.set_const(); g.set_const()
    g.option("rhs_variant", 6)
    for _ in range(2):
        o.step(); g.step()
    assert np.array_equal(g.get(F["Q"]), o.get(orc.Q))


@pytest.mark.parametrize("nx,ny,nl,extra", [(256, 128, 6, ""), (512, 64, 3, SLIP), (64, 64, 1, ""), (128, 128, 4, PER)])
def test_lpw_fast_build_matches_lds_tile_kernel(nx, ny, nl, extra):
    """product build: both kernels evaluate the same terms; the vertical coupling is summed in a different order.
    Tolerance: 1e-12 of max|dq| (fp64 round-off of ~10 additions)"""
    out = []
    for variant in (1, 6):
        txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny != nx else "") + extra)
        g = QG(txt); g.option("quiet", 1); g.option("TOLERANCE", 1e-9); g.option("rhs_variant", variant)
        g.set(F["PSI"], orc.synthetic_psi(nl, ny, nx)); g.set_const()
        dq, d = g.update()
        for _ in range(2):
            g.step()
        out.append((dq, d, g.get(F["Q"])))
    assert out[0][1] == out[1][1]
    assert rel(out[1][0], out[0][0]) < 1e-12
    assert rel(out[1][2], out[0][2]) < 1e-11
