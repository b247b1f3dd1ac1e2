"""The hook-level boundary driven from the CALLER's side, exactly as Basilisk's run() of predictor-corrector.h drives the
reference's plug-ins (msqg/qg.h:922-923: `update = update_qg; advance = advance_qg;`):

    dt = dtnext(update(evolving, updates, DT));  advance(predictor, evolving, updates, dt / 2);
    update(predictor, updates, dt);              advance(evolving, evolving, updates, dt);

with the caller holding q, dq and the predictor (host arrays, then device arrays), the library holding only psi (the warm
start of the next inversion, like the reference's `pol`).  The composition must equal `msom_step` and the CPU oracle bit
for bit in the strict build -- deterministic (msqg/qg.h:594-650) and stochastic (msqg/qg_stochastic.h:128-149: every other
advance draws new noise, the predictor's weight is sqrt(dt)/sqrt(2)) -- and the product build to round-off.  Also the
elliptic plug-in below the path called directly: msom_invertq (invertq, msqg/qg.h:114-163) and msom_comp_q (:397-403)."""
import ctypes

import numpy as np
import pytest

import orc
from msom_amd import QG, FIELDS as F
from test_gpu_parity import make_pair, rand_field, rel

pytestmark = pytest.mark.gpu


def caller_side_step(g, q, DT):
    """one RK2 step of run() through msom_update / msom_advance; returns (q_new, dt)"""
    dq, dtmax = g.update(q, DT)
    dt = dtmax                      # dtnext() with no scheduled event ahead (tnext = inf)
    qp = g.advance(q, dq, dt / 2)   # predictor
    dq2, _ = g.update(qp, dt)       # return value discarded, as run() does
    return g.advance(q, dq2, dt), dt


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nx,ny,nl,extra", [(64, 64, 3, ""), (128, 32, 6, "sbc = 1.5\nRe = 300\nEks = 0.001\n"), (32, 32, 1, ""), (64, 64, 2, "sbc = -1\ntau0 = 0\n")])
def test_rk2_from_the_caller_side_equals_msom_step_and_oracle(nx, ny, nl, extra, strict):
    o, a = make_pair(nx, ny, nl, strict=strict, extra=extra, TOLERANCE=1e-9 if strict else 1e-12)
    _, b = make_pair(nx, ny, nl, strict=strict, extra=extra, TOLERANCE=1e-9 if strict else 1e-12)
    for m in (o, a):
        m.set_tnext(float("inf"))
    DT = b.param("DT")
    q = b.get(F["Q"])
    for _ in range(3):
        o.step()
        dt_a = a.step()
        q, dt_b = caller_side_step(b, q, DT)
        if strict:
            assert dt_b == dt_a == o.dt
            assert np.array_equal(q, a.get(F["Q"]))
            assert np.array_equal(q, o.get(orc.Q))
            assert np.array_equal(b.get(F["PSI"]), a.get(F["PSI"]))   # the warm start the library keeps
        else:
            assert dt_b == pytest.approx(dt_a, rel=1e-12)
            assert rel(q, a.get(F["Q"])) <= 1e-12      # fused advance (msom_step) against the separate advance kernel
            assert rel(q, o.get(orc.Q)) <= 1e-10
    a.close(); b.close()


class DevBuf:
    """fp64 array in HBM through the HIP runtime the library itself links (plain hipMalloc / hipMemcpy via ctypes)"""
    hip = None

    def __init__(self, a):
        if DevBuf.hip is None:
            DevBuf.hip = ctypes.CDLL("libamdhip64.so")
        self.shape, self.nbytes = a.shape, a.nbytes
        self.ptr = ctypes.c_void_p()
        assert DevBuf.hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(a.nbytes)) == 0
        assert DevBuf.hip.hipMemcpy(self.ptr, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1) == 0   # host -> device

    def host(self):
        out = np.empty(self.shape)
        assert DevBuf.hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), self.ptr, ctypes.c_size_t(self.nbytes), 2) == 0  # device -> host
        return out

    def free(self):
        DevBuf.hip.hipFree(self.ptr)


def test_hooks_take_device_pointers():
    """the same composition with every array in HBM: the C ABI accepts host or device pointers"""
    nx = ny = 64; nl = 3
    _, a = make_pair(nx, ny, nl, strict=True)
    _, b = make_pair(nx, ny, nl, strict=True)
    a.set_tnext(float("inf"))
    q0 = b.get(F["Q"])
    q, dq, qp = DevBuf(q0), DevBuf(np.zeros_like(q0)), DevBuf(np.zeros_like(q0))
    L, h = b.L, b.h
    DT = b.param("DT")
    for _ in range(2):
        dt_a = a.step()
        dt = L.msom_update(h, q.ptr, dq.ptr, DT)
        assert dt == dt_a
        assert L.msom_advance(h, qp.ptr, q.ptr, dq.ptr, dt / 2) == 0
        assert L.msom_update(h, qp.ptr, dq.ptr, dt) > 0
        assert L.msom_advance(h, q.ptr, q.ptr, dq.ptr, dt) == 0      # in place, as advance(evolving, evolving, ...)
        assert np.array_equal(q.host(), a.get(F["Q"]))
    for d in (q, dq, qp):
        d.free()
    a.close(); b.close()


@pytest.mark.parametrize("strict", [True, False])
def test_stochastic_rk2_from_the_caller_side(strict):
    """-D_STOCHASTIC: advance_qg toggles predictor / corrector itself (msqg/qg_stochastic.h:132-137), so the caller's
    sequence update, advance(dt/2), update, advance(dt) reproduces the noise schedule of msom_step and of the oracle on
    the same serial rand() stream"""
    nx = ny = 16; nl = 3
    ex = "tr_stoch = 50\namp_stoch = 1e-5\n"
    sig = np.abs(rand_field(12, (nl, ny, nx)))
    libc = ctypes.CDLL(None)
    o, a = make_pair(nx, ny, nl, strict=strict, extra=ex, stochastic=1, TOLERANCE=1e-9 if strict else 1e-12)
    _, b = make_pair(nx, ny, nl, strict=strict, extra=ex, stochastic=1, TOLERANCE=1e-9 if strict else 1e-12)
    for m, f in ((o, orc.SIGMA), (a, F["SIGMA"]), (b, F["SIGMA"])):
        m.set(f, sig)
    outs = {}
    for name, m in (("o", o), ("a", a)):
        libc.srand(7)
        m.set_tnext(float("inf"))
        for _ in range(3):
            m.step()
        outs[name] = m.get(orc.Q if m is o else F["Q"])
    libc.srand(7)
    q = b.get(F["Q"])
    DT = b.param("DT")
    for _ in range(3):
        q, _ = caller_side_step(b, q, DT)
    if strict:
        assert np.array_equal(q, outs["a"]) and np.array_equal(q, outs["o"])
    else:
        assert rel(q, outs["a"]) <= 1e-11 and rel(q, outs["o"]) <= 1e-9
    a.close(); b.close()


@pytest.mark.parametrize("nx,ny,nl", [(64, 64, 3), (128, 64, 6), (32, 32, 1)])
def test_invertq_and_comp_q_called_directly(nx, ny, nl):
    """msom_comp_q(psi) = lap(psi) + Gamma(psi) and msom_invertq(q, psi_first_guess) -> (psi, mgstats): equal to the oracle's
    comp_q / invertq (cycle count, residuals) and to the array-level entry points pyp2q / pyq2p, bit for bit (strict)"""
    o, g = make_pair(nx, ny, nl, strict=True, TOLERANCE=1e-10)
    psi = rand_field(5, (nl, ny, nx), 1e-3)
    q = g.comp_q(psi)
    o.set(orc.PSI, psi); o.comp_q()
    assert np.array_equal(q, o.get(orc.Q))
    q2 = np.empty_like(q); g.pyp2q(psi, q2)
    assert np.array_equal(q, q2)
    # inversion from a zero first guess
    p, st = g.invertq(q, psi0=np.zeros_like(psi))
    o.set(orc.PSI, np.zeros_like(psi)); o.set(orc.Q, q); o.invertq()
    so = o.mgstats()
    assert (st.i, st.resb, st.resa, st.nrelax) == (so.i, so.resb, so.resa, so.nrelax)
    assert np.array_equal(p, o.get(orc.PSI))
    p2 = np.empty_like(p); g.pyq2p(p2, q)       # pyq2p starts from the psi the library holds: the converged one
    assert rel(p2, psi) <= 1e-4 and rel(p, psi) <= 1e-4     # TOLERANCE 1e-10 on max |res|; |q| ~ 1 here
    g.close()
