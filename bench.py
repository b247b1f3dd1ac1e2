#!/usr/bin/env python
"""bench.py -- timesteps/s and grid-point-updates/s of the multi-layer QG hot path.

One "step" = one full predictor-corrector (RK2) time step of msqg: 2 elliptic inversions
q -> psi (multigrid, TOLERANCE 1e-3 as msqg/qg.h:159), 2 PV-tendency evaluations, 2 advances
(SURVEY 8d).  Workload at N = 1: BASELINE.json's metric configuration, 4096 x 4096 x 6 layers,
fp64, Verron double-gyre parameters (msqg/test/params.double_gyre.in) with nl = 6, synthetic
seed-free initial stream function.  N > 1: one process per GPU, weak scaling -- every rank
owns one 4096 x 4096 x 6 tile of a (4096 px) x (4096 py) domain.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the
library's stream around the dominant kernel (finest-level red-black smoother colour sweep);
`cpu_baseline` is the CPU oracle (oracle/, OpenMP) timed on the host cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md


def cpu_baseline(nl, n_cpu=1024, steps=2):
    """CPU oracle (plain C + OpenMP, red-black smoother) on a bounded sample: `steps` RK2 steps
    of the same parameter set on an n_cpu^2 x nl grid; grid-point-updates/s is size-normalised."""
    import orc

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
    o = orc.Oracle(orc.double_gyre_params(n_cpu, nl), smoother=orc.GS_RB, quiet=1)
    o.set(orc.PSI, orc.synthetic_psi(nl, n_cpu, n_cpu))
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
        "sample": f"{steps} RK2 steps at {n_cpu}x{n_cpu}x{nl} fp64 (same params, red-black smoother, 1 warm-up step), "
                  f"{dt / steps * 1e3:.0f} ms/step",
        "steps_per_s": steps / dt,
    }


_VARIANT_SNIPPET = """
import sys, time
sys.path.insert(0, {tests!r})
import orc
n, nl, steps, sm = {n}, {nl}, {steps}, {sm}
o = orc.Oracle(orc.double_gyre_params(n, nl), smoother=sm, quiet=1)
o.set(orc.PSI, orc.synthetic_psi(nl, n, n)); o.set_const(); o.step()
t0 = time.perf_counter()
for _ in range(steps): o.step()
print(n * n * nl * steps / (time.perf_counter() - t0))
"""


def cpu_variants(nl, n=512, steps=2):
    """SURVEY 8d asks for the lexicographic (reference order) and single-thread CPU timings next to the
    OpenMP red-black one: each in its own process (OMP_NUM_THREADS is read once), on a smaller sample."""
    import subprocess

    out = {}
    for name, sm, threads in (("red_black_1_thread", 1, 1), ("lexicographic_reference_order_1_thread", 0, 1)):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        code = _VARIANT_SNIPPET.format(tests=os.path.join(ROOT, "tests"), n=n, nl=nl, steps=steps, sm=sm)
        try:
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
            out[name] = {"value": float(r.stdout.strip().splitlines()[-1]), "unit": "grid-point-updates/s", "cores": threads,
                         "sample": f"{steps} RK2 steps at {n}x{n}x{nl}"}
        except Exception as e:  # the headline baseline above does not depend on these
            out[name] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=4096, help="tile edge (cells)")
    ap.add_argument("--nl", type=int, default=6)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-n", type=int, default=1024)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import orc
    from msom_amd import FIELDS as F
    from msom_amd import QG

    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from msom_amd import load_library, tiling

    px, py = tiling.tile_grid(world)
    N, nl = args.N, args.nl
    gnx, gny = N * px, N * py
    # weak scaling: the tile (N x N x nl, Delta = 80/N) is fixed, the domain grows with the tile grid
    params = orc.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else ""), L0=80.0 * px)
    lib = load_library()
    if lib.msom_set_device(local_rank) != 0:
        raise SystemExit(lib.msom_last_error().decode())
    if world > 1:
        uid = tiling.broadcast_unique_id(dist, lambda: tiling.rccl_unique_id(lib), device="cuda")
        g = QG(params, tiled=(px, py, rank, uid))
    else:
        g = QG(params)
    g.option("quiet", 1)

    # synthetic seed-free IC: global sine modes sampled on this rank's tile
    def psi_fn(l, y, x):
        f = np.zeros((y.size, x.size))
        for k in range(1, 5):
            for m in range(1, 5):
                f += np.sin(1.7 * k + 2.3 * m + 0.9 * l) / (k * m) * np.outer(np.sin(m * np.pi * y), np.sin(k * np.pi * x))
        return 1e-3 * (1.0 - 0.15 * l) * f

    g.set(F["PSI"], tiling.synthetic_tile(psi_fn, rank, px, py, nl, N, N))
    g.set_const()
    g.set_tnext(float("inf"))

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        g.step()
    g.option("profile", 1)
    g.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.step()   # msom_step synchronises the library's stream before returning
    barrier()
    elapsed = time.perf_counter() - t0
    g.option("profile", 0)
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    sweep_ms, sweep_n = g.profile_read("sweep")      # one sweep = red + black launch (half-sweep-per-launch path)
    m4_ms, m4_n = g.profile_read("march4")           # finest-level passes of 4 / 3 chained half-sweeps (one GPU)
    m3_ms, m3_n = g.profile_read("march3")
    resid_ms, resid_n = g.profile_read("residual")
    st = g.mgstats()
    ke = g.ke()
    # second kernel of the step (the Jacobian / PV-tendency pass): a short back-to-back microbenchmark after the timed
    # region (HIP events on the library's stream); every rank takes part (tiles exchange the psi halo)
    rhs_ms = g.bench_kernel("rhs_adv", 5)
    plain_ms = g.bench_kernel("sweep", 5) / 2.0     # one plain colour half-sweep launch, for comparison

    if rank == 0:
        w = 8.0 * N * N * nl                       # bytes of one layered fp64 field of the tile
        uniform = g.param("uniform_S") == 1.0
        sigma = 0.0 if uniform else (nl - 1) / nl
        # algorithmic bytes of ONE colour half-sweep launch: read the other colour's da (w/2),
        # read own-colour res (w/2) [+ own-colour S], write own-colour da (w/2)
        plain_bytes = (3.0 + sigma) * w / 2.0
        marched = m4_n + m3_n > 0
        if marched:
            # the smoother's finest-level pass: K chained half-sweeps per launch (kernels_march.hip).
            # Contract figure: SURVEY 8(d)'s per-unit bytes (one red+black sweep = (3 + sigma) w: R a, R b, W a) x the
            # units one launch processes (K / 2 sweeps).  The pass itself needs less -- the other colour's da in (w/2),
            # the residual of both colours (w), both colours of da out (w) = 2.5 w whatever K -- because the values
            # between the chained half-sweeps never leave the registers; that figure and the measured HBM traffic are
            # reported next to the contract one (so `frac` can exceed 1: it is an effective bandwidth)
            K, launch_ms, launches = (4, m4_ms, m4_n) if m4_n > 0 else (3, m3_ms, m3_n)   # the longer pass where both run
            launch_bytes = plain_bytes * K
            pass_bytes = 2.5 * w
            kernel = f"k_relax_march<{nl}, {K}, false, false> (finest level: {K} chained red-black colour half-sweeps = {K / 2:g} sweeps per pass, intermediate values in registers)"
            pmc_file = "r01_pmc_traffic_march.json"
        else:
            K, launch_bytes, launch_ms, launches = 1, plain_bytes, sweep_ms / 2.0, 2 * sweep_n
            pass_bytes = plain_bytes
            kernel = f"k_relax_color_x2<{nl}, {'true' if uniform else 'false'}, true> (finest-level red-black colour half-sweep)"
            pmc_file = "r01_pmc_traffic_relax_fine.json"
        achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
        # collected separately with rocprofv3 --pmc and stored under profiles/
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            if N == 4096 and nl == 6 and uniform and world == 1:
                traffic = pmc["traffic_bytes_per_launch"] if K == 1 else pmc[f"traffic_bytes_per_launch_K{K}"]
        except Exception:
            pass
        out = {
            "metric": "grid-point-updates/s (timesteps/s x N^2 x nl), multi-layer QG RK2 step at 4096^2 x 6L per GPU",
            "value": gnx * gny * nl * args.steps / elapsed,
            "unit": "grid-point-updates/s",
            "timesteps_per_s": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"msqg double gyre (Verron 1992 params), {gnx}x{gny}x{nl} fp64, tiles {px}x{py} of {N}x{N}, "
                            f"TOLERANCE 1e-3, RK2 step = 2 inversions + 2 tendencies + 2 advances",
                "mg_cycles_per_solve": st.i, "mg_nrelax": st.nrelax, "mg_resa": st.resa, "ke_1": ke,
                "uniform_S_fast_path": bool(uniform),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": launch_bytes,
                "avg_launch_ms": launch_ms,
                "launches_timed": launches,
                "half_sweeps_per_launch": K,
                "other_pass": ({"half_sweeps_per_launch": 3, "avg_launch_ms": m3_ms, "launches_timed": m3_n} if marched and K == 4 and m3_n > 0 else None),
                "per_unit_bytes": 2.0 * plain_bytes, "units_per_launch": K / 2.0, "unit_name": "red+black sweep (SURVEY 8d: R a, R b, W a)",
                "pass_compulsory_bytes": pass_bytes,
                "achieved_vs_pass_bytes": pass_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0,
                "frac_vs_pass_bytes": pass_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if launch_ms > 0 else 0.0,
                "plain_half_sweep_kernel": {
                    "kernel": f"k_relax_color_x2<{nl}, {'true' if uniform else 'false'}, true> (one colour half-sweep per launch; used on tiles and small levels)",
                    "avg_launch_ms": plain_ms, "algorithmic_bytes_per_launch": plain_bytes,
                    "achieved_GBs": plain_bytes / (plain_ms * 1e-3) / 1e9 if plain_ms > 0 else 0.0,
                    "note": "K of these move 1.5 K w; the chained pass moves 2.5 w for the same K half-sweeps",
                },
                "tendency_kernel": {
                    "kernel": "k_rhs_lpw<4, true, false, true> (Arakawa Jacobians + beta + dissipation + drag + forcing + advance, one pass over psi)",
                    "avg_launch_ms": rhs_ms,
                    "algorithmic_bytes_per_launch": 3.0 * w,
                    "achieved_GBs": 3.0 * w / (rhs_ms * 1e-3) / 1e9 if rhs_ms > 0 else 0.0,
                    "frac_hbm": 3.0 * w / (rhs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rhs_ms > 0 else 0.0,
                    "fp64_flop_per_point_layer": 215,
                    "achieved_fp64_TFLOPs": 215.0 * N * N * nl / (rhs_ms * 1e-3) / 1e12 if rhs_ms > 0 else 0.0,
                    "note": "fused: reads psi and q_in once, writes q_out once (the reference's loop chain moves ~25 w); "
                            "one layer per wavefront, stencils from register windows + whole-wave DPP shifts; fp64-issue / wait bound (3 waves per SIMD), not HBM bound",
                },
                "residual_kernels": {
                    "avg_launch_ms": resid_ms, "launches_timed": resid_n,
                    "note": "k_residual2<write+restrict> (3.25 w) and k_residual2<correct> (4 w) alternate",
                    "achieved_GBs": (3.625 * w / (resid_ms * 1e-3) / 1e9) if resid_ms > 0 else 0.0,
                },
            },
        }
        if not args.no_cpu and world == 1:   # CPU baseline: rank 0 at N = 1 only
            g.close()
            out["cpu_baseline"] = cpu_baseline(nl, n_cpu=args.cpu_n)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["variants"] = cpu_variants(nl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
