#!/usr/bin/env python
"""bench.py -- timesteps/s and grid-point-updates/s of the multi-layer QG hot path.

One "step" = one full predictor-corrector (RK2) time step of msqg: 2 elliptic inversions q -> psi (multigrid,
TOLERANCE 1e-3 as msqg/qg.h:159), 2 PV-tendency evaluations, 2 advances.

Workloads (BASELINE.json configs; Verron double-gyre parameters of msqg/test/params.double_gyre.in with N / nl
overridden, synthetic seed-free initial stream function, fp64):
  C4 (default)  4096 x 4096 x 6   -- the configuration the metric is quoted on; fits one MI355X (14 of 288 GB)
  C3            2048 x 2048 x 3
  C2             512 x  512 x 3
  C1             128 x  128 x 1   -- BASELINE configs[0]: the 1-layer barotropic double gyre the CPU reference runs
  C5            2048 x 2048 x 3, stochastic variant (msqg/qg_stochastic.h; noise from the device Philox generator)
N = 1: the whole grid on one GPU.  N > 1 (one process per GPU, RCCL): by default WEAK scaling -- every rank owns one
tile of the configuration's N = 1 size (`--tile NXxNY` overrides it), the domain grows with the tile grid and the
grid spacing stays fixed; `--split` instead splits the configuration's own grid over the ranks (BASELINE C4 as
written: 4096^2 x 6 on 2 x 4 tiles of 2048 x 1024 at N = 8).  After the timed weak-scaling leg a multi-rank run also
times that split layout and reports it as `split_global_grid` in the same line.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel of the step (the finest-level smoother pass);
its duration is the average of HIP-event pairs recorded on the library's own stream around every such launch INSIDE
the timed steps (option "profile"), `frac` = the pass's own compulsory HBM bytes / that duration / 8 TB/s.  The other
finest-level kernels (PV tendency, residual passes, red + prolongation) are timed the same way and listed under
`roofline.kernels`.  `cpu_baseline` is the CPU oracle (oracle/, OpenMP) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md

TRAFFIC_PROFILE = "r03_pmc_traffic_march.json"   # FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh, summarised by tools/pmc_summary.py


def kernel_source_sha1():
    """fingerprint of the chained-smoother source the PMC traffic figure belongs to"""
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(ROOT, "msom_amd", "csrc", "kernels_march.hip"), "rb").read()).hexdigest()
    except OSError:
        return None


CONFIGS = {
    "C1": dict(N=128, nl=1, stochastic=False),
    "C2": dict(N=512, nl=3, stochastic=False),
    "C3": dict(N=2048, nl=3, stochastic=False),
    "C4": dict(N=4096, nl=6, stochastic=False),
    "C5": dict(N=2048, nl=3, stochastic=True),
}


def cpu_baseline(nl, n_cpu=4096, steps=2):
    """CPU oracle (plain C + OpenMP, red-black smoother, all host cores of this process) on a bounded sample of the SAME
    workload: 1 warm-up + `steps` timed RK2 steps at the metric configuration's own size (4096^2 x 6 by default: ~3 s per
    step on the GPU box's 16-thread share, ~1-2 min for the leg with the 14 GB of first touch).  The only place bench.py
    touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from msom_amd import workloads as wl

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = cores
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    threads = max(1, min(cores, quota, 16))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    o = orc.Oracle(wl.double_gyre_params(n_cpu, nl), smoother=orc.GS_RB, quiet=1)
    o.set(orc.PSI, wl.synthetic_psi(nl, n_cpu, n_cpu))
    o.set_const()
    o.step()  # warm-up (first touch, limiter start)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step()
    dt = time.perf_counter() - t0
    return {
        "value": n_cpu * n_cpu * nl * steps / dt,
        "unit": "grid-point-updates/s",
        "cores": orc.lib().orc_num_threads(),
        "kind": "port",
        "sample": f"{steps} RK2 steps at {n_cpu}x{n_cpu}x{nl} fp64 (same params.in, same initial psi, TOLERANCE 1e-3, red-black smoother, "
                  f"1 warm-up step), {dt / steps * 1e3:.0f} ms/step on {orc.lib().orc_num_threads()} OpenMP threads",
        "steps_per_s": steps / dt,
    }


_VARIANT_SNIPPET = """
import sys, time
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
import orc
from msom_amd import workloads as wl
n, nl, steps, sm = {n}, {nl}, {steps}, {sm}
o = orc.Oracle(wl.double_gyre_params(n, nl), smoother=sm, quiet=1)
o.set(orc.PSI, wl.synthetic_psi(nl, n, n)); o.set_const(); o.step()
t0 = time.perf_counter()
for _ in range(steps): o.step()
print(n * n * nl * steps / (time.perf_counter() - t0))
"""


def cpu_variants(nl, n=512, steps=2):
    """SURVEY 8d asks for the lexicographic (reference order) and single-thread CPU timings next to the OpenMP red-black
    one: each in its own process (OMP_NUM_THREADS is read once), on a smaller sample."""
    import subprocess

    out = {}
    for name, sm, threads in (("red_black_1_thread", 1, 1), ("lexicographic_reference_order_1_thread", 0, 1)):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        code = _VARIANT_SNIPPET.format(tests=os.path.join(ROOT, "tests"), root=ROOT, n=n, nl=nl, steps=steps, sm=sm)
        try:
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
            out[name] = {"value": float(r.stdout.strip().splitlines()[-1]), "unit": "grid-point-updates/s", "cores": threads,
                         "sample": f"{steps} RK2 steps at {n}x{n}x{nl}"}
        except Exception as e:  # the headline baseline above does not depend on these
            out[name] = {"error": str(e)[:200]}
    return out


def psi_fn(l, y, x):
    """synthetic seed-free IC: global sine modes sampled on a tile (same formula as workloads.synthetic_psi)"""
    f = np.zeros((y.size, x.size))
    for k in range(1, 5):
        for m in range(1, 5):
            f += np.sin(1.7 * k + 2.3 * m + 0.9 * l) / (k * m) * np.outer(np.sin(m * np.pi * y), np.sin(k * np.pi * x))
    return 1e-3 * (1.0 - 0.15 * l) * f


class Leg:
    """one model instance on this rank's tile of a (tx px) x (ty py) domain + its timed loop"""

    def __init__(self, tx, ty, nl, px, py, rank, dist, cfg_n, stochastic=False, local_rank=0, extra_params="", fr_field=False):
        from msom_amd import FIELDS as F
        from msom_amd import QG, load_library, tiling
        from msom_amd import workloads as wl

        self.tx, self.ty, self.nl, self.px, self.py, self.dist = tx, ty, nl, px, py, dist
        self.gnx, self.gny = tx * px, ty * py
        # the grid spacing of the configuration (Delta = 80 / N_config) is the same on every leg: L0 follows the domain width
        extra = (f"Ny = {self.gny}\n" if self.gny != self.gnx else "")
        if stochastic:
            extra += "tr_stoch = 50\namp_stoch = 1e-5\n"
        extra += extra_params
        params = wl.double_gyre_params(self.gnx, nl, extra=extra, L0=80.0 * self.gnx / cfg_n)
        self.params = params
        lib = load_library()
        if lib.msom_set_device(local_rank) != 0:
            raise SystemExit(lib.msom_last_error().decode())
        if px * py > 1:
            uid = tiling.broadcast_unique_id(dist, lambda: tiling.rccl_unique_id(lib), device="cuda")
            g = QG(params, tiled=(px, py, rank, uid))
        else:
            g = QG(params)
        g.option("quiet", 1)
        if stochastic:
            g.option("stochastic", 1)
            g.option("noise_mode", 1)   # counter-based Philox on the device (the reference's serial rand() is a host loop)
            g.option("seed", 7)
        g.set(F["PSI"], tiling.synthetic_tile(psi_fn, rank, px, py, nl, tx, ty))
        if stochastic:
            g.set(F["SIGMA"], np.ones((nl, ty, tx)))
        if fr_field:
            # a spatially varying Froude field, what frpg_<nl>l_N<N>.bas supplies (msqg/qg.h:940-984): S = (Fr / Ro)^2 per cell
            ix, iy = rank % px, rank // px
            x = (ix * tx + np.arange(tx) + 0.5) / self.gnx
            y = (iy * ty + np.arange(ty) + 0.5) / self.gny
            shape = 1.0 + 0.3 * np.outer(np.sin(2 * np.pi * y), np.cos(2 * np.pi * x))
            if nl > 1:
                g.set(F["FR"], np.stack([g.param(f"Fr_{l}") * shape for l in range(g.shape(F["FR"])[0])]))
        g.set_const()
        g.set_tnext(float("inf"))
        self.g = g

    def barrier(self):
        self.g.sync()     # the library's own stream (msom_step may return with its last tendency pass still running)
        if self.dist is not None:
            import torch

            self.dist.barrier()
            torch.cuda.synchronize()

    PROF_KEYS = ("march4", "march_pl", "march_corr", "march3", "march2", "sweep", "red_prolong", "resid_restrict", "resid_correct", "resid_max", "rhs")

    def run(self, steps, warmup):
        """W warm-up steps, then EXACTLY `steps` timed steps between barriers.  Inside the timed steps only the chained smoother
        passes (the dominant kernel of the roofline entry) are bracketed by HIP events (option profile = 2): an event pair costs
        ~10 us of stream time, and with every kernel bracketed the step is 1.4 % (4096^2 x 6) to 20 % (512^2 x 3) slower.  The
        other kernels are timed the same way in a second, untimed pass of up to 20 steps right after."""
        g = self.g
        for _ in range(warmup):
            g.step()
        g.option("profile", 2)
        g.profile_reset()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        g.option("profile", 0)
        self.prof = {k: g.profile_read(k) for k in self.PROF_KEYS}
        g.profile_reset()
        g.option("profile", 1)
        for _ in range(min(steps, 20)):
            g.step()
        g.option("profile", 0)
        for k in self.PROF_KEYS:
            if self.prof[k][1] == 0:
                self.prof[k] = g.profile_read(k)
        if self.dist is not None:
            import torch

            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def summary(self, steps, elapsed):
        st = self.g.mgstats()
        return {
            "grid": f"{self.gnx}x{self.gny}x{self.nl}", "tiles": f"{self.px}x{self.py} of {self.tx}x{self.ty}",
            "value": self.gnx * self.gny * self.nl * steps / elapsed, "unit": "grid-point-updates/s",
            "timesteps_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "mg_cycles_per_solve": st.i, "mg_nrelax": st.nrelax, "mg_resa": st.resa, "ke_1": self.g.ke(),
        }

    def kernels(self):
        """finest-level kernels timed with HIP events inside the timed steps; bytes = compulsory HBM bytes of the launch"""
        g, nl = self.g, self.nl
        w = 8.0 * self.tx * self.ty * nl
        uniform = g.param("uniform_S") == 1.0
        sigma = 0.0 if uniform else (nl - 1) / nl
        plain_bytes = (3.0 + sigma) * w / 2.0
        lean = "; interior chunks in the lean body march_lean" if g.param("march_lean") >= 1.0 else ""
        spec = [
            ("march4", "k_relax_march_dma<nl,4> (4 chained red-black half-sweeps per pass, rows by LDS-DMA" + lean + ")", 2.5 * w, "other colour in w/2 + residual w + both colours out w"),
            ("march_pl", "k_relax_march_dma<nl,4,PL> (bilinear prolongation + 4 chained half-sweeps: first pass of a level visit" + lean + ")", 1.75 * w,
             "coarse correction w/4 + residual w + last colour out w/2 (the other colour is recomputed by the next pass before anything reads it)"),
            ("march_corr", "k_relax_march_dma<nl,4,CORR> (4 chained half-sweeps + correction psi += da: last pass of the cycle" + lean + ")", 3.5 * w,
             "other colour in w/2 + residual w + psi in w + psi out w"),
            ("march3", "k_relax_march<nl,3> (3 chained half-sweeps per pass)", 2.5 * w, "as march4"),
            ("march2", "k_relax_march<nl,2>", 2.5 * w, "as march4"),
            ("sweep", "k_relax_color_x2 red + black (two launches)", 2.0 * plain_bytes, "per colour: other colour in, own residual in [, S], own colour out"),
            ("red_prolong", "k_relax_red_prolong3 (first red half-sweep + bilinear prolongation)", 1.25 * w, "residual w/2 + coarse w/4 + red out w/2"),
            ("resid_restrict", "k_residual2<write+restrict> (pre-cycle residual + the first two restrictions)", (3.25 + 1.0 / 16.0) * w, "psi, q in, residual out, level-1 and level-2 residuals out w/4 + w/16")
            if g.param("restrict2") == 1.0 else
            ("resid_restrict", "k_residual2<write+restrict> (pre-cycle residual + first restriction)", 3.25 * w, "psi, q in, residual out, level-1 residual out w/4"),
            ("resid_correct", "k_correct_residual (psi += da, residual max, max|u|)", 4.0 * w, "psi, da, q in, psi out"),
            ("resid_max", "k_resmax_march<nl> (marching max|res|, max|u| of the corrected psi)" if g.param("resmax_marching") == 1.0 else
             "k_correct_residual<max only> (LDS-tiled max|res|, max|u| of the corrected psi)", 2.0 * w, "psi, q in"),
            ("rhs", "k_rhs_lpw (Arakawa Jacobians + beta + dissipation + drag + forcing + advance)", 3.0 * w, "psi, q_in in, q_out out"),
        ]
        out = {}
        for key, name, nbytes, what in spec:
            ms, n = self.prof[key]
            if n > 0 and ms > 0:
                out[key] = {"kernel": name, "avg_launch_ms": ms, "launches_timed": n, "compulsory_bytes_per_launch": nbytes, "bytes_are": what,
                            "achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "frac_hbm": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if "rhs" in out:
            # SURVEY 8(d) prices the tendency as three separate passes -- Jacobians (3 + sigma) w, dissipation (3 + sigma) w, advance 3 w;
            # the fused pass does their work on 3 w, so this figure is an EFFECTIVE rate (it may exceed the HBM peak), never `frac`
            sv = (9.0 + 2.0 * (nl - 1) / nl) * w
            out["rhs"]["effective_bw_in_survey_units"] = {"GBs": sv / (out["rhs"]["avg_launch_ms"] * 1e-3) / 1e9, "bytes_if_run_as_separate_passes": sv,
                                                          "note": "K1+K3 (3+sigma) w + K5 (3+sigma) w + K7 3 w, SURVEY 8(d)"}
            out["rhs"]["fp64_flop_per_point_layer"] = 215
            out["rhs"]["achieved_fp64_TFLOPs"] = 215.0 * self.tx * self.ty * nl / (out["rhs"]["avg_launch_ms"] * 1e-3) / 1e12
        return out, w, plain_bytes, uniform


def vertex_sqg_leg(N=2048, nl=3, steps=6):
    """BASELINE config 5, second half: the vertex-grid model (qg-node) with the surface-QG option and an island mask,
    (N + 1)^2 x nl vertices, TOLERANCE 1e-5 (qg-node/params.in:22); a short leg, the vertex path is correct-first code"""
    from msom_amd import NodeQG

    dh, n2 = "[0.1,0.3,0.6]", "[300.,9000.,3000.]"
    txt = (f"N = {N}\nnl = {nl}\nL0 = 100\nf0 = 46.5\nhEkb = 0.01\ntau0 = 1e-3\nnu = 5.0\nnu4 = 0.0\nbeta = 0.5\nbc_fac = 1.0\n"
           f"dh = {dh}\nN2 = {n2}\nDT = 5.e-2\ntend = 100.\ndtout = 1\nCFL = 0.2\nTOLERANCE = 1e-5\nsqg = 1\n")
    g = NodeQG(txt)
    g.set_option("quiet", 1)
    x = np.arange(N + 1) / N
    mk = np.ones((1, N + 1, N + 1))
    mk[0, N // 4: N // 4 + N // 8, N // 2: N // 2 + N // 8] = 0
    mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
    psi = np.stack([1e-2 * (1 - 0.2 * l) * sum(np.sin(1.3 * k + 2.1 * m + 0.7 * l) / (k * m) * np.outer(np.sin(m * np.pi * x), np.sin(k * np.pi * x))
                                              for k in range(1, 4) for m in range(1, 4)) for l in range(nl)]) * mk
    g.set("MASK", mk)
    g.set("BS", 0.3 * np.outer(np.sin(np.pi * x), np.sin(2 * np.pi * x))[None] + 0.05)
    g.set("PSI", psi)
    g.set_const()
    for _ in range(3):
        g.step(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        g.step(True)
    el = time.perf_counter() - t0
    out = {"grid": f"{N + 1}x{N + 1}x{nl} vertices", "value": (N + 1) ** 2 * nl * steps / el, "unit": "vertex-updates/s", "ms_per_step": el / steps * 1e3,
           "steps": steps, "mg_cycles_per_solve": g.mgstats().i,
           "variant": "qg-node vertex model, sqg = 1 (surface buoyancy prescribed), island mask, no-slip, 5 sweeps per level and cycle"}
    # finest-level launches, HIP-event timed in two more (untimed) steps; bytes = compulsory HBM bytes of the launch with
    # wv = 8 (N + 1)^2 nl: each distinct array once per read and once per write (mask: one layer; S2: row tables, no field read)
    g.set_option("profile", 1); g.profile_reset()
    for _ in range(2):
        g.step(True)
    g.set_option("profile", 0)
    wv = 8.0 * (N + 1) ** 2 * nl
    spec = {"march_fine": ("k_n_relax_march_s<nl,K> (K = 2..4 chained colour half-sweeps, split layout; the figure is an average over the K of the cycle)", (1.5 + 0.5 / nl) * wv,
                           "a pass of K half-sweeps: other colour w/2 + residual and mask of the colours it updates (counted once: w/2 + w/(2 nl)) -> last colour w/2; the same bytes as ONE colour pass"),
            "relax_fine": ("k_n_relax_s<nl> (one colour half-sweep, split layout, S2 row tables)", (1.5 + 0.5 / nl) * wv, "other colour w/2 + own residual w/2 + own mask w/(2 nl) -> own colour w/2"),
            "relax_prolong_fine": ("k_n_relax_prolong_s<nl> (prolongation + first colour half-sweep)", (1.75 + 0.5 / nl) * wv, "coarse w/4 + residual w/2 + mask -> both colours w"),
            "residual": ("k_n_residual (first residual of a solve, to the split layout + max)", (3.0 + 1.0 / nl) * wv, "psi, q, mask in; residual out"),
            "correct_residual": ("k_n_correct_residual_m<nl> (correction of cycle i + residual of cycle i + 1, rows marched)", (5.0 + 1.0 / nl) * wv, "psi, da, q, mask in; psi (second buffer), residual out"),
            "correct": ("k_n_correct (psi += da, boundary value: last cycle of a solve)", 3.0 * wv, "psi, da in; psi out"),
            "rhs": ("rhs_pv in three passes: k_n_rhs_pre (mask, zeta), k_n_del2_bnd (tmp), k_n_rhs_all (Jacobians, beta, drag, stretch / del2 pairs, forcing, mask) [+ k_n_lap_bs with sqg]",
                    (5.0 + 1.0 / nl + 2.0 + 4.0 + (nl - 1.0) / nl + 3.0 / nl) * wv, "q in/out, psi in, psi out, zeta out, mask | zeta in, tmp out | psi, zeta, tmp, S2 in, dq out, 2-d fields"),
            "coarse": ("k_n_mg_coarse<nl> (levels of <= 33^2 vertices in one launch)", None, "launch-latency bound")}
    ks = {}
    for slot, (name, nbytes, what) in spec.items():
        ms, n = g.profile_read(slot)
        if n:
            ks[slot] = {"kernel": name, "avg_launch_ms": ms, "launches_per_step": n / 2.0, "bytes_are": what}
            if nbytes:
                ks[slot].update({"compulsory_bytes_per_launch": nbytes, "achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "frac_hbm": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    out["kernels"] = ks
    out["finest_level_ms_per_step"] = sum(v["avg_launch_ms"] * v["launches_per_step"] for v in ks.values())
    g.close()
    return out


def roofline(leg, world, leg_steps=1):
    ks, w, plain_bytes, uniform = leg.kernels()
    nl = leg.nl
    dom = next((k for k in ("march_corr", "march4", "march3", "march2") if k in ks), "sweep" if "sweep" in ks else None)
    if dom is None:
        return {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None, "kernels": ks}
    d = ks[dom]
    K = 4 if dom == "march_corr" else (int(dom[-1]) if dom.startswith("march") else 2)
    nbytes, ms = d["compulsory_bytes_per_launch"], d["avg_launch_ms"]
    if dom == "sweep":      # two launches per timed pair: report one colour half-sweep launch
        nbytes, ms, K = nbytes / 2.0, ms / 2.0, 1
    achieved = nbytes / (ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, collected with rocprofv3 --pmc in separate
    # runs and stored under profiles/); only quoted for the exact configuration it was measured on
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
        if (leg.tx, leg.ty, nl, world) == (4096, 4096, 6, 1) and uniform and dom.startswith("march"):
            traffic = pmc.get(f"traffic_bytes_per_launch_{dom}")
            # NOT measured by this run: PMC counters need their own rocprofv3 passes (no --pmc beside tracing on this pool)
            traffic_source = {"file": "profiles/" + TRAFFIC_PROFILE, "measured_at_commit": pmc.get("commit"), "kernel_source_sha1_then": pmc.get("kernel_source_sha1"),
                              "kernel_source_sha1_now": kernel_source_sha1(), "how": pmc.get("what")}
            if traffic_source["kernel_source_sha1_then"] != traffic_source["kernel_source_sha1_now"]:
                traffic = None   # the kernel changed since the counters were read: do not quote a stale figure
                traffic_source["stale"] = True
    except Exception:
        pass
    # SURVEY 8(d)'s accounting unit for the smoother is a red+black SWEEP = (3 + sigma) w [R a, R b, W a]; a pass of K
    # chained half-sweeps does the work of K/2 such sweeps while moving only 2.5 w: effective bandwidth, NOT a fraction
    eff = 2.0 * plain_bytes * (K / 2.0) / (ms * 1e-3) / 1e9
    return {
        "bound": "hbm",
        "kernel": d["kernel"] if dom != "sweep" else f"k_relax_color_x2<{nl}> (one colour half-sweep per launch)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "launches_timed": d["launches_timed"],
        "half_sweeps_per_launch": K,
        "traffic_over_algorithmic": (traffic / nbytes) if traffic else None,
        "effective_bw_in_survey_sweep_units": {"GBs": eff, "per_unit_bytes": 2.0 * plain_bytes, "units_per_launch": K / 2.0,
                                               "note": "K half-sweeps = K/2 sweeps of (3+sigma) w each if run one by one; values between chained half-sweeps stay in registers"},
        "smoother_that_ran": dom,
        # all finest-level kernels together: compulsory bytes of every timed launch / their summed durations
        "finest_level_kernels_together": (lambda b, t: {"GBs": b / t / 1e9 if t > 0 else 0.0, "frac_hbm": b / t / 1e9 / HBM_PEAK_GBS if t > 0 else 0.0,
                                                       "compulsory_GB_per_step": b / 1e9 / max(1, leg_steps), "kernel_ms_per_step": t * 1e3 / max(1, leg_steps)})(
            sum(v["compulsory_bytes_per_launch"] * v["launches_timed"] for v in ks.values()),
            sum(v["avg_launch_ms"] * 1e-3 * v["launches_timed"] for v in ks.values())),
        "kernels": ks,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C4")
    ap.add_argument("--tile", default=None, help="NXxNY: tile per rank (weak scaling), overrides the configuration's grid")
    ap.add_argument("--split", action="store_true", help="split the configuration's own grid over the ranks (strong)")
    ap.add_argument("--nl", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (other configs at N = 1, split grid at N > 1)")
    ap.add_argument("--cpu-n", type=int, default=0, help="grid of the cpu_baseline leg (0: the configuration's own N)")
    ap.add_argument("--opt", action="append", default=[], help="key=value library option (tuning A/B), repeatable")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from msom_amd import tiling

    cfg = CONFIGS[args.config]
    nl = args.nl or cfg["nl"]
    px, py = tiling.tile_grid(world)
    if args.tile:
        tx, ty = (int(v) for v in args.tile.lower().split("x"))
    elif args.split:
        tx, ty = cfg["N"] // px, cfg["N"] // py
    else:
        tx = ty = cfg["N"]
    try:
        leg = Leg(tx, ty, nl, px, py, rank, dist, cfg["N"], stochastic=cfg["stochastic"], local_rank=local_rank)
        for kv in args.opt:
            k, v = kv.split("=")
            leg.g.option(k, float(v))
        elapsed = leg.run(args.steps, args.warmup)
    except Exception as e:  # noqa: BLE001  -- say what failed in the one line the driver reads, then fail
        if rank == 0:
            print(json.dumps({"metric": "grid-point-updates/s", "value": 0.0, "unit": "grid-point-updates/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "higher_is_better": True, "error": repr(e)}), flush=True)
        raise
    main_sum = leg.summary(args.steps, elapsed)
    roof = roofline(leg, world, args.steps) if rank == 0 else None
    uniform = leg.g.param("uniform_S") == 1.0
    leg.g.close()

    extra = {}
    if not args.no_extra:
        if world == 1 and args.config == "C4" and not args.tile:
            # the other BASELINE configurations on one GPU, short legs (their kernels are timed the same way)
            # (a leg that fails is recorded as such: the headline line must still be printed)
            for name in ("C1", "C2", "C3", "C5"):
                c = CONFIGS[name]
                try:
                    lg = Leg(c["N"], c["N"], c["nl"], 1, 1, 0, None, c["N"], stochastic=c["stochastic"], local_rank=local_rank)
                    steps = 100 if c["N"] <= 512 else 40
                    el = lg.run(steps, 10)
                    s = lg.summary(steps, el)
                    ks, _, _, _ = lg.kernels()
                    s["kernels"] = {k: {"avg_launch_ms": v["avg_launch_ms"], "frac_hbm": v["frac_hbm"]} for k, v in ks.items()}
                    if c["stochastic"]:
                        s["variant"] = "msqg/qg_stochastic.h, device Philox noise (noise_mode 1), tr_stoch 50, amp_stoch 1e-5, sigma = 1"
                    extra[name] = s
                    lg.g.close()
                except Exception as e:  # noqa: BLE001
                    extra[name] = {"error": repr(e)}
            # the metric configuration off the uniform-S / walls fast path: the doubly periodic box
            # (sbc = -1, msqg/qg.h:842-846; tau0 = 0: a periodic box cannot absorb the mean of the wind curl) and a
            # spatially varying Froude field (general column solver: every cell factorises its own tridiagonal system)
            for name, kw, what in (("C4_periodic", dict(extra_params="sbc = -1\ntau0 = 0\n"), "4096x4096x6, sbc = -1 (doubly periodic), tau0 = 0"),
                                   ("C4_general_S", dict(fr_field=True), "4096x4096x6, Fr(x, y) = Frm (1 + 0.3 sin 2 pi y cos 2 pi x): general column solver")):
                try:
                    c = CONFIGS["C4"]
                    lg = Leg(c["N"], c["N"], c["nl"], 1, 1, 0, None, c["N"], local_rank=local_rank, **kw)
                    el = lg.run(10, 3)
                    s_ = lg.summary(10, el)
                    ks, _, _, _ = lg.kernels()
                    s_["kernels"] = {k: {"avg_launch_ms": v["avg_launch_ms"], "frac_hbm": v["frac_hbm"]} for k, v in ks.items()}
                    s_["smoother_that_ran"] = next((k for k in ("march_corr", "march4", "march3", "march2", "sweep") if k in ks), None)
                    s_["uniform_S_fast_path"] = bool(lg.g.param("uniform_S") == 1.0)
                    s_["variant"] = what
                    extra[name] = s_
                    lg.g.close()
                except Exception as e:  # noqa: BLE001
                    extra[name] = {"error": repr(e)}
            try:
                extra["C5_vertex_sqg"] = vertex_sqg_leg()
            except Exception as e:  # noqa: BLE001
                extra["C5_vertex_sqg"] = {"error": repr(e)}
        elif world > 1 and not args.split and not args.tile:
            # BASELINE C4 as written: the configuration's own grid split over the ranks (2 x 4 tiles of 2048 x 1024 at N = 8)
            try:
                lg, err = None, None
                try:
                    lg = Leg(cfg["N"] // px, cfg["N"] // py, nl, px, py, rank, dist, cfg["N"], stochastic=cfg["stochastic"], local_rank=local_rank)
                except Exception as e:  # noqa: BLE001
                    err = e
                # a rank that failed to create its tile must not leave the others inside the first collective of the leg:
                # every rank learns whether all tiles exist before any of them steps
                import torch
                bad = torch.tensor([1.0 if err is not None else 0.0], device="cuda")
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if float(bad.item()) > 0:
                    raise err if err is not None else RuntimeError("another rank could not create its tile of the split layout")
                el = lg.run(args.steps, args.warmup)
                extra["split_global_grid"] = lg.summary(args.steps, el)
                if rank == 0:
                    ks, _, _, _ = lg.kernels()
                    extra["split_global_grid"]["smoother_that_ran"] = next((k for k in ("march_corr", "march4", "march3", "march2", "sweep") if k in ks), None)
                lg.g.close()
            except Exception as e:  # noqa: BLE001
                extra["split_global_grid"] = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": f"grid-point-updates/s (timesteps/s x N^2 x nl), multi-layer QG RK2 step at {tx}x{ty}x{nl} per GPU",
            "value": main_sum["value"],
            "unit": "grid-point-updates/s",
            "timesteps_per_s": main_sum["timesteps_per_s"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_sum["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if args.split else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: msqg double gyre (Verron 1992 params){' + stochastic forcing' if cfg['stochastic'] else ''}, "
                            f"{leg.gnx}x{leg.gny}x{nl} fp64, tiles {px}x{py} of {tx}x{ty}, "
                            f"TOLERANCE 1e-3, RK2 step = 2 inversions + 2 tendencies + 2 advances",
                "mg_cycles_per_solve": main_sum["mg_cycles_per_solve"], "mg_nrelax": main_sum["mg_nrelax"], "mg_resa": main_sum["mg_resa"],
                "ke_1": main_sum["ke_1"], "uniform_S_fast_path": bool(uniform),
                **({"options": args.opt} if args.opt else {}),
            },
            "roofline": roof,
        }
        if extra:
            out["other_legs"] = extra
        if not args.no_cpu and world == 1:   # CPU baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(nl, n_cpu=args.cpu_n or cfg["N"])
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["variants"] = cpu_variants(nl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
