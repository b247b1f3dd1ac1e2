"""Canonical synthetic workloads of the hot path: the Verron double-gyre parameter set of
msqg/test/params.double_gyre.in with N / nl overridden, and the seed-free initial stream function.  Shared by the
tests, the golden-fixture generator and bench.py; knows nothing about the CPU oracle."""
import numpy as np

DOUBLE_GYRE = """#!sh
# Double gyre configuration of Verron 1992 (values of msqg/test/params.double_gyre.in)
N  = {N}
nl = {nl}
L0 = 80
Rom   = 0.025
Ekb   = 0.002
tau0  = 0.0001
Re4   = {Re4}
beta  = 0.5
Fr = {Fr}
dh = {dh}
DT    = 5.e-2
tend  = 500.
dtout = 1.
CFL   = 0.6
"""

LAYERS = {
    1: ("[0.0023669]", "[1.0]"),
    2: ("[0.0023669]", "[0.2,0.8]"),
    3: ("[0.0023669,0.0076173]", "[0.06,0.14,0.8]"),
    4: ("[0.0023669,0.0076173,0.0076173]", "[0.06,0.14,0.4,0.4]"),
    5: ("[0.0023669,0.0023669,0.0076173,0.0076173]", "[0.03,0.03,0.14,0.4,0.4]"),
    6: ("[0.0023669,0.0023669,0.0076173,0.0076173,0.0076173]", "[0.03,0.03,0.07,0.07,0.4,0.4]"),
    7: ("[0.0023669,0.0023669,0.0076173,0.0076173,0.0076173,0.0076173]", "[0.03,0.03,0.07,0.07,0.2,0.2,0.4]"),
    8: ("[0.0023669,0.0023669,0.0076173,0.0076173,0.0076173,0.0076173,0.0076173]", "[0.03,0.03,0.07,0.07,0.2,0.2,0.2,0.2]"),
}


def _split_layers(nl):
    """nl > 8 (build-defined, like the nl = 6 split of BASELINE.md): the three Verron layers cut into equal sub-layers, the
    interface Froude numbers of the layer the upper neighbour belongs to"""
    cnt = [max(1, round(nl * f)) for f in (0.25, 0.25)]
    cnt.append(nl - sum(cnt))
    dh, fr = [], []
    for h, fr_below, c in zip((0.06, 0.14, 0.8), (0.0023669, 0.0076173, 0.0076173), cnt):
        dh += [h / c] * c
        fr += [fr_below] * c
    return "[" + ",".join(f"{v:.7g}" for v in fr[:nl - 1]) + "]", "[" + ",".join(f"{v:.7g}" for v in dh) + "]"


for _nl in range(9, 17):
    LAYERS[_nl] = _split_layers(_nl)


def double_gyre_params(N, nl, extra="", L0=80.0):
    """Verron double gyre with N, nl overridden; Re4 ~ Delta^-4 keeps the viscous
    clamp of msqg/qg.h:746 at DT = 0.025 for every resolution."""
    Fr, dh = LAYERS[nl]
    delta_ratio = (80.0 / 256.0) / (L0 / N)
    return DOUBLE_GYRE.format(N=N, nl=nl, Re4=1563.0 * delta_ratio ** 4, Fr=Fr, dh=dh).replace("L0 = 80", f"L0 = {L0}") + extra


def synthetic_psi(nl, ny, nx, amp=1e-3):
    """Seed-free IC of SURVEY 8d: 16 sine modes per layer, zero on the walls."""
    x = (np.arange(nx) + 0.5) / nx
    y = (np.arange(ny) + 0.5) / ny
    psi = np.zeros((nl, ny, nx))
    for l in range(nl):
        for k in range(1, 5):
            for m in range(1, 5):
                c = np.sin(1.7 * k + 2.3 * m + 0.9 * l) / (k * m)
                psi[l] += c * np.outer(np.sin(m * np.pi * y), np.sin(k * np.pi * x))
        psi[l] *= amp * (1.0 - 0.15 * l)
    return psi
