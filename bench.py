#!/usr/bin/env python3
"""bench.py -- DFSPH dam-break throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--mode faithful]

A "step" is one pass of the reference main-loop body (ref: dfsph.py:606-617: grid update, density,
alpha, divergence solve, viscosity PCG + vorticity, CFL dt, velocity update, pressure solve, position
update) over the synthetic dam break of SURVEY 8(d).  N=1 runs config C3 (126^3 = 2 000 376 liquid
particles + ~1.0 M boundary particles), the per-GPU load of the 8-GPU config C4 (weak scaling).
Rank 0 prints ONE JSON line: M particle-steps/s for the whole job, plus
  "roofline":     the north-star kernel pair `pressure_iter` (ref: dfsph.py:518-547) timed live with
                  HIP events on the context's stream, algorithmic bytes (84 + 24 s) B per liquid particle
                   over the 8 TB/s HBM peak,
  "cpu_baseline": the reference-algorithm CPU restatement (oracle/, faithful mode, OpenMP) timed on this
                  box's host cores on a bounded C2-size sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_sim(n, mode, **attrs):
    from wcsph_amd import dfsph
    t0 = time.time()
    pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode, **attrs)
    dfsph.reset_param()
    pd.sync()
    return dfsph, pd, time.time() - t0


def cpu_baseline(n, steps):
    """Reference-algorithm CPU restatement (oracle/wcsph_ref.c, faithful mode) on the host cores: all cores, and 1 thread."""
    import oracle
    from oracle import Oracle
    from wcsph_amd import scene
    liquid, solid = scene.dam_break_dfsph(n)
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.025, mode="faithful")
    o.reset_param()
    o.dfsph_step()  # first step pays the page faults of the 8 KiB/particle neighbour table
    maxt = oracle.get_max_threads()
    # the loops of a C2-size scene are short: try a few team sizes, report the best one (and which)
    sweep = {}
    for th in sorted({maxt, 64, 32, 16} & set(range(1, maxt + 1)), reverse=True):
        oracle.set_num_threads(th)
        t0 = time.perf_counter()
        o.dfsph_step()
        sweep[th] = time.perf_counter() - t0
    cores = min(sweep, key=sweep.get)
    oracle.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.dfsph_step()
    dt = time.perf_counter() - t0
    oracle.set_num_threads(1)
    t1 = time.perf_counter()
    o.dfsph_step()
    dt1 = time.perf_counter() - t1
    oracle.set_num_threads(maxt)
    val = len(liquid) * steps / dt / 1e6
    o.close()
    return {"value": round(val, 4), "unit": "M particle-steps/s", "cores": cores, "kind": "port",
            "sample": "DFSPH dam break %d^3 = %d liquid + %d boundary particles (config C2 size), %d steps after 1 "
                      "warm-up step, faithful neighbour mode, gcc -O2 -fopenmp" % (n, len(liquid), len(solid), steps),
            "ms_per_step": round(dt / steps * 1e3, 2), "host_threads_available": maxt,
            "thread_sweep_ms": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "single_thread": {"value": round(len(liquid) / dt1 / 1e6, 4), "ms_per_step": round(dt1 * 1e3, 1), "steps": 1}}


def main():
    # the reference's classes print scene statistics (ParticleData.py:185, HashGrid.py:54); stdout carries the JSON line only
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        out = _main()
    if out is not None:
        print(json.dumps(out), flush=True)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3", help="C2 | C3 | default | an integer particleDim")
    ap.add_argument("--mode", default="faithful", choices=["faithful", "clean"])
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-n", type=int, default=46)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-clean", action="store_true", help="skip the clean-mode companion run")
    ap.add_argument("--kernel-reps", type=int, default=20)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        from wcsph_amd import slab
        return slab.bench_main(args)

    import __graft_entry__ as g
    g.build()
    from wcsph_amd import scene, _lib
    import ctypes as C
    n = scene.CONFIGS.get(args.config) or int(args.config)
    dfsph, pd, t_setup = build_sim(n, args.mode)
    ctx = pd._need_ctx()
    lib = _lib.load()
    try:
        import torch
        tsync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    except Exception:
        tsync = lambda: None  # noqa: E731

    iters = []
    for _ in range(args.warmup):
        dfsph.step()
    pd.sync(); tsync()
    # every pressure_iter launch group of the timed region is bracketed by HIP events on the context's stream
    _lib.check(ctx, lib.wcsph_profile_begin(ctx, 16 * args.steps + 16))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dfsph.step()
        iters.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter))
    pd.sync(); tsync()
    elapsed = time.perf_counter() - t0
    ms, n_launch = C.c_float(), C.c_int32()
    _lib.check(ctx, lib.wcsph_profile_end(ctx, C.byref(ms), C.byref(n_launch)))
    nl, ns = pd.liquid_count, pd.solid_count
    value = nl * args.steps / elapsed / 1e6

    # the same kernel pair replayed back to back after the run (a replay finds more of its stream in the Infinity Cache)
    ms_replay = C.c_float()
    _lib.check(ctx, lib.wcsph_time_kernel(ctx, b"pressure_iter", args.kernel_reps, C.byref(ms_replay)))
    s = ns / nl
    alg_bytes = (84.0 + 24.0 * s) * nl
    achieved = alg_bytes / (ms.value * 1e-3) / 1e9
    extra = {}
    for name in ("pressure_vel_loop", "pressure_drho_loop", "divergence_iter", "density_coff", "grid_update", "visc_iter"):
        m2 = C.c_float()
        _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), max(args.kernel_reps // 2, 2), C.byref(m2)))
        extra[name + "_us"] = round(m2.value * 1e3, 1)
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf))
            if t.get("n_liquid") == nl:
                traffic = t.get("pressure_iter_bytes_per_launch")
        except Exception:
            pass
    it = np.array(iters, dtype=np.float64)
    out = {
        "metric": "M particle-steps/sec, DFSPH dam-break", "value": round(value, 3), "unit": "M particle-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "DFSPH 3D dam-break %s: %d^3 = %d liquid + %d boundary particles, CubicKernel, CFL dt, "
                               "viscosity PCG + vorticity on (reference defaults)" % (args.config, n, nl, ns),
                   "neighbor_mode": args.mode, "n_liquid": nl, "n_solid": ns,
                   "mean_iters": {"vs": round(it[:, 0].mean(), 2), "dv": round(it[:, 1].mean(), 2), "pr": round(it[:, 2].mean(), 2)},
                   "dt_end": float(dfsph.deltaT[0]), "setup_s": round(t_setup, 2)},
        "roofline": {"bound": "hbm", "kernel": "pressure_iter as the fused solver launches it: k_vel_from_k (kappa += k) + k_drho_press<ITER> (k epilogue), ref dfsph.py:518-547",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "avg_launch_us": round(ms.value * 1e3, 1), "launches_timed": n_launch.value,
                     "timing": "HIP events around every launch group inside the timed steps", "replay_us": round(ms_replay.value * 1e3, 1),
                     "algorithmic_bytes_per_launch": int(alg_bytes), "other_kernels": extra},
    }
    pd.close()
    if args.mode == "faithful" and not args.no_clean:
        # the N > 1 slab runs use clean neighbour semantics: same workload, same step count, for a
        # like-for-like reading of the multi-GPU numbers
        dfsph2, pd2, _ = build_sim(n, "clean")
        for _ in range(args.warmup):
            dfsph2.step()
        pd2.sync(); tsync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dfsph2.step()
        pd2.sync(); tsync()
        el2 = time.perf_counter() - t0
        out["config"]["clean_mode"] = {"value": round(nl * args.steps / el2 / 1e6, 3), "ms_per_step": round(el2 / args.steps * 1e3, 3)}
        pd2.close()
    if not args.no_clean:
        # equal-work micro-benchmark: fixed dv = 2, pr = 4, vs = 4 iterations per step
        d4, p4, _ = build_sim(n, args.mode)
        for _ in range(max(args.warmup // 2, 2)):
            d4.step_fixed(2, 4, 4)
        p4.sync(); tsync()
        t0 = time.perf_counter()
        nfix = max(args.steps // 2, 5)
        for _ in range(nfix):
            d4.step_fixed(2, 4, 4)
        p4.sync(); tsync()
        el4 = time.perf_counter() - t0
        out["config"]["fixed_iters_dv2_pr4_vs4"] = {"value": round(nl * nfix / el4 / 1e6, 3), "ms_per_step": round(el4 / nfix * 1e3, 3), "steps": nfix}
        p4.close()
    if args.config == "C3" and not args.no_clean:
        # BASELINE.json configs[1] (C2, ~100 k particles) for reference: too small to fill 256 CUs, launch-bound
        d3, p3, _ = build_sim(scene.CONFIGS["C2"], args.mode)
        for _ in range(args.warmup):
            d3.step()
        p3.sync(); tsync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d3.step()
        p3.sync(); tsync()
        el3 = time.perf_counter() - t0
        out["config"]["C2_100k"] = {"value": round(p3.liquid_count * args.steps / el3 / 1e6, 3), "ms_per_step": round(el3 / args.steps * 1e3, 3),
                                    "n_liquid": p3.liquid_count}
        p3.close()
    if args.config == "C3" and not args.no_clean:
        # BASELINE.json configs[4] (C5): 1 M particles with the Akinci surface tension switched on as well (the reference's
        # default coefficients are 0, which skips both tension passes)
        d5, p5, _ = build_sim(scene.CONFIGS["C5"], args.mode, tension_coff=0.05, tension_coff_b=0.05)
        for _ in range(args.warmup):
            d5.step()
        p5.sync(); tsync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d5.step()
        p5.sync(); tsync()
        el5 = time.perf_counter() - t0
        out["config"]["C5_1M_tension"] = {"value": round(p5.liquid_count * args.steps / el5 / 1e6, 3), "ms_per_step": round(el5 / args.steps * 1e3, 3),
                                          "n_liquid": p5.liquid_count, "tension_coff": 0.05, "tension_coff_b": 0.05}
        p5.close()
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.cpu_steps)
    return out


if __name__ == "__main__":
    main()
