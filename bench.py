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
  "cpu_baseline": the reference-algorithm CPU restatement (oracle/, faithful mode, OpenMP, gcc -O3
                  -march=native) timed on this box's host cores on the SAME configuration (C3), 5 steps.
`value` is the configuration's first steps (the column still at rest: the cheapest state of a dam break);
config.developed_flow reports the same measurement after 1200 steps (collapsed, splashing).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL and device-tensor sharing fail with hipIpcGetMemHandle: invalid
# argument otherwise); the driver's environment exports it already -- keep it for every other launcher, before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# a rank of a decomposition drives five streams (compute, the communicator's halo and side streams, torch's own, the null stream); HIP
# maps a process's streams onto FOUR hardware queues by default, and two dependent streams that share one pay ~260 us per cross-stream
# event wait (profiles/r06_overlap.md, "a hazard found ...").  Eight queues for the multi-rank launch only: the 1-GPU run has one stream.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

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


def kernel_source_sha16():
    """hash of the sources that define the pressure_iter kernels: a recorded PMC traffic figure is only quoted for them"""
    h = hashlib.sha256()
    for f in ("device.h", "dfsph_kernels.h", "grid.h"):
        h.update(open(os.path.join(ROOT, "wcsph_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# ---- HBM traffic of the north-star kernel pair, measured in THIS run -------------------------------------------------------
def _pmc_passes(n, mode, counters, timeout=240):
    """one child `rocprofv3 --pmc <counter> --kernel-trace` run of tools/run_kernel_group.py per counter (the profiler cannot attach
    to this process; separate passes and no other trace domain, as MI355X_MICROARCH.md "HBM" / "rocprofv3" prescribes).
    Returns ({counter: {kernel: average value per dispatch}}, None) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="wcsph_pmc_", dir="/tmp")
    vals = {}
    try:
        for group in counters:          # one rocprofv3 run per group (counters of one group share a pass)
            out = os.path.join(tmp, group.split()[0])
            cmd = [rocprof, "--pmc"] + group.split() + ["--kernel-trace", "--kernel-include-regex", "k_vel_from_k|k_drho_press", "-d", out, "-o", "pmc",
                   "--", sys.executable, os.path.join(ROOT, "tools", "run_kernel_group.py"), "pressure_iter", "10", str(n), mode]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out" % group
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (group, r.returncode)
            for counter in group.split():
                vals[counter] = {}
                for kern in ("k_vel_from_k", "k_drho_press"):
                    rows = list(sqlite3.connect(dbs[0]).execute(
                        "select avg(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + kern + "%", counter)))
                    if not rows or rows[0][0] is None or rows[0][1] < 4:
                        return None, "no %s samples for %s" % (counter, kern)
                    vals[counter][kern] = float(rows[0][0])
    except Exception as e:   # profiling is best effort: the bench line must come out
        return None, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return vals, None


TRAFFIC_NOTE = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over k_vel_from_k + "
                "k_drho_press<ITER>; 2 x FETCH + WRITE, KB per dispatch.  The guide's x2 FETCH_SIZE correction is calibrated for wide "
                "coalesced streams; for 16-byte GATHERS it is uncalibrated, so this is an UPPER bound (x1 would give ~0.6 of it), and "
                "hits in the 256 MB Infinity Cache are counted: traffic_frac bounds the HBM-side rate from above, it does not measure it")


def measure_traffic_live(n, mode, timeout=240):
    """FETCH_SIZE and WRITE_SIZE of k_vel_from_k + k_drho_press<ITER> per launch.  Bytes per pressure_iter launch =
    2 x FETCH_SIZE (gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads; WRITE_SIZE is taken as is) in KB per
    dispatch.  Returns (bytes, note) or (None, reason)."""
    vals, why = _pmc_passes(n, mode, ("FETCH_SIZE", "WRITE_SIZE"), timeout)
    if vals is None:
        return None, why
    kb = {c: sum(vals[c].values()) for c in vals}
    return int((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024), TRAFFIC_NOTE


# measured on this architecture with tools/micro/valu_rate.hip and tools/micro/gather_rate.hip (profiles/r03i_gather_rate.txt)
VALU_RATE_G = 1035.0          # G wave-instructions / s, whole chip
GATHER_CYCLES = 16.2          # CU-cycles per 64-lane gather of 8..16-byte records that all hit the L1
N_CU, CLOCK_GHZ = 256, 2.4


def measure_pipes_live(n, mode, timeout=240):
    """What the pair is bound by, next to the HBM fraction: VALU time from SQ_INSTS_VALU at the measured issue
    rate, the vector memory pipeline's floor from the gather instruction count (16.2 CU-cycles each whatever the lanes do), and the
    vector-L1 tag look-ups per gather instruction (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD: 16 would be ideal for 64 x 16 B)."""
    vals, why = _pmc_passes(n, mode, ("SQ_INSTS_VALU SQ_INSTS_VMEM_RD", "TCP_TOTAL_CACHE_ACCESSES"), timeout)
    if vals is None:
        return {"note": why}
    valu = sum(vals["SQ_INSTS_VALU"].values())
    vmem = sum(vals["SQ_INSTS_VMEM_RD"].values())
    tcp = sum(vals["TCP_TOTAL_CACHE_ACCESSES"].values())
    return {"valu_us": round(valu / (VALU_RATE_G * 1e9) * 1e6, 1), "valu_wave_instructions": int(valu),
            "gather_floor_us": round(vmem * GATHER_CYCLES / (N_CU * CLOCK_GHZ * 1e9) * 1e6, 1), "vmem_read_wave_instructions": int(vmem),
            "l1_lines_per_gather": round(tcp / max(vmem, 1.0), 1), "l1_lookup_us": round(tcp / (N_CU * CLOCK_GHZ * 1e9) * 1e6, 1),
            "note": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_VMEM_RD / TCP_TOTAL_CACHE_ACCESSES, separate passes, per pressure_iter launch (both kernels); "
                    "valu_us at %.0f G wave-instr/s (tools/micro/valu_rate.hip), gather_floor_us = VMEM reads x %.1f CU-cycles / (%d CUs x %.1f GHz), "
                    "l1_lookup_us = one tag look-up per clock and CU" % (VALU_RATE_G, GATHER_CYCLES, N_CU, CLOCK_GHZ)}


# ---- CPU baseline (child process: the baseline build of the oracle, its own 17 GB of tables) -------------------------
def _cpu_child(n, steps, sweep, single):
    import oracle
    from oracle import Oracle
    from wcsph_amd import scene
    liquid, solid = scene.dam_break_dfsph(n)
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.025, mode="faithful")
    o.reset_param()
    o.dfsph_step()  # first step pays the page faults of the 8 KiB/particle neighbour table
    maxt = oracle.get_max_threads()
    times = {}
    for th in sorted({t for t in sweep if 1 <= t <= maxt} | {maxt}, reverse=True):
        oracle.set_num_threads(th)
        t0 = time.perf_counter()
        o.dfsph_step()
        times[th] = time.perf_counter() - t0
    cores = min(times, key=times.get)
    oracle.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.dfsph_step()
    dt = time.perf_counter() - t0
    out = {"value": round(len(liquid) * steps / dt / 1e6, 4), "ms_per_step": round(dt / steps * 1e3, 1), "cores": cores,
           "n_liquid": len(liquid), "n_solid": len(solid), "steps": steps, "host_threads_available": maxt,
           "thread_sweep_ms": {str(k): round(v * 1e3, 1) for k, v in times.items()},
           "iters_last_step": [o.vs_iter, o.dv_iter, o.pr_iter]}
    if single:
        oracle.set_num_threads(1)
        t1 = time.perf_counter()
        o.dfsph_step()
        d1 = time.perf_counter() - t1
        out["single_thread"] = {"value": round(len(liquid) / d1 / 1e6, 4), "ms_per_step": round(d1 * 1e3, 1), "steps": 1}
    o.close()
    print("CPU_BASELINE_JSON " + json.dumps(out), flush=True)


def _run_cpu_child(n, steps, sweep, single, timeout):
    env = dict(os.environ, WCSPH_ORACLE_BASELINE="1")
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its workers when the caller set none: the baseline child runs after
    # the ranks have left, alone on the host, and sweeps the thread counts itself
    if "TORCHELASTIC_RUN_ID" in env or int(env.get("WORLD_SIZE", "1")) > 1:
        env.pop("OMP_NUM_THREADS", None)
    code = "import bench; bench._cpu_child(%d, %d, %r, %r)" % (n, steps, list(sweep), bool(single))
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout}
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            return json.loads(line[len("CPU_BASELINE_JSON "):])
    return {"error": (r.stderr or r.stdout)[-400:]}


def cpu_baseline(n_main, n_small):
    """Reference-algorithm CPU restatement (oracle/wcsph_ref.c, faithful mode, gcc -O3 -march=native -fopenmp, built on
    this host) on the headline configuration; the C2-size sample of round 1 and a 1-thread figure ride along."""
    nthreads = os.cpu_count() or 1
    main = _run_cpu_child(n_main, 5, [nthreads, nthreads // 2, nthreads // 4], False, 1200)
    small = _run_cpu_child(n_small, 10, [nthreads, 64, 32, 16], True, 300)
    if "error" in main:
        return {"value": None, "unit": "M particle-steps/s", "cores": 0, "kind": "port", "sample": "failed: " + main["error"], "c2_sample": small}
    return {"value": main["value"], "unit": "M particle-steps/s", "cores": main["cores"], "kind": "port",
            "sample": "the headline workload itself: DFSPH dam break %d^3 = %d liquid + %d boundary particles (config C3), %d steps after 1 "
                      "warm-up step and a %d-step thread sweep, faithful neighbour mode, gcc -O3 -march=native -fopenmp built on this host"
                      % (n_main, main["n_liquid"], main["n_solid"], main["steps"], len(main["thread_sweep_ms"])),
            "ms_per_step": main["ms_per_step"], "host_threads_available": main["host_threads_available"],
            "thread_sweep_ms": main["thread_sweep_ms"], "iters_last_step": main["iters_last_step"], "c2_sample": small}


# ---- timed legs ---------------------------------------------------------------------------------------------------
def timed_steps(dfsph, pd, warmup, steps, tsync, fixed=None, profile=False):
    from wcsph_amd import _lib
    import ctypes as C
    ctx, lib = pd._need_ctx(), _lib.load()
    step = (lambda: dfsph.step_fixed(*fixed)) if fixed else dfsph.step
    for _ in range(warmup):
        step()
    pd.sync(); tsync()
    if profile:   # every pressure_iter launch group of the timed region is bracketed by HIP events on the context's stream
        _lib.check(ctx, lib.wcsph_profile_begin(ctx, 16 * steps + 16))
    iters = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        iters.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter))
    pd.sync(); tsync()
    elapsed = time.perf_counter() - t0
    out = {"elapsed": elapsed, "iters": np.array(iters, dtype=np.float64)}
    if profile:
        ms, n_launch = C.c_float(), C.c_int32()
        _lib.check(ctx, lib.wcsph_profile_end(ctx, C.byref(ms), C.byref(n_launch)))
        out["launch_ms"], out["launches"] = ms.value, n_launch.value
    return out


# ---- per-group breakdown of a step, in situ -------------------------------------------------------------
def lap_group(tag):
    """kernel tag of a lap (csrc PROF_LAP) -> the group names the `rest` leg's other_kernels use"""
    if tag.startswith("grid:"):
        return "grid_update"
    if tag == "k_build_lists":
        return "build_lists"
    if tag.startswith("k_density"):
        return "density_coff"
    if tag in ("divergence_iter", "pressure_iter", "step_boundary"):
        return tag
    if "k_drho_div<DRHO_BEGIN" in tag or tag.startswith("k_vel_from_k<1>"):
        return "divergence_open"            # warm-start velocities + begin_divergence_iter (dfsph.py:415-446)
    if "k_drho_press<DRHO_BEGIN" in tag or "k_warm_press" in tag:
        return "pressure_open"              # warmstart_pressure + begin_pressure_iter (dfsph.py:487-516)
    if tag in ("k_visc_minv_residual", "k_visc_minv_residual_vort", "k_visc_Ad", "k_visc_update", "k_visc_dir", "k_vorticity"):
        return "visc_iter"                  # the PCG (its opening walk + every iteration) and the vorticity sums that ride on it
    return "elementwise"                    # k_end_div_begin_nonpressure, k_finish_nonpressure (+ CFL tail), k_update_vel_warm_scale, k_end_press_update_pos ...


def lap_breakdown(dfsph, pd, steps):
    """`steps` more steps with a HIP event behind every tagged launch (wcsph_profile_laps_*): us per step by group.  The laps tile
    the stream's time line, so the groups sum to the profiled steps' own duration; `after_host_wait_us` is the part of it
    in laps that began behind a host wait (an upper bound of the GPU's idle time there)."""
    from wcsph_amd import _lib
    import ctypes as C
    ctx, lib = pd._need_ctx(), _lib.load()
    pd.sync()
    _lib.check(ctx, lib.wcsph_profile_laps_begin(ctx, 160 * steps + 16))
    t0 = time.perf_counter()
    iters = []
    for _ in range(steps):
        dfsph.step()
        iters.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter))
    pd.sync()
    wall = time.perf_counter() - t0
    buf = C.create_string_buffer(1 << 16)
    _lib.check(ctx, lib.wcsph_profile_laps_end(ctx, buf, len(buf)))
    laps = json.loads(buf.value.decode())
    groups, waits, by_kernel = {}, 0.0, {}
    for tag, (ms, n, ms_wait) in laps.items():
        g = lap_group(tag)
        groups[g] = groups.get(g, 0.0) + ms
        waits += ms_wait
        by_kernel[tag] = [round(ms / steps * 1e3, 1), round(n / steps, 2)]
    tot = sum(groups.values())
    it = np.array(iters, dtype=np.float64)
    return {"steps": steps, "ms_per_step_profiled": round(wall / steps * 1e3, 3),
            "groups_us_per_step": {k: round(v / steps * 1e3, 1) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
            "groups_sum_ms_per_step": round(tot / steps, 3), "after_host_wait_us_per_step": round(waits / steps * 1e3, 1),
            "mean_iters": {"vs": round(it[:, 0].mean(), 2), "dv": round(it[:, 1].mean(), 2), "pr": round(it[:, 2].mean(), 2)},
            "laps_us_per_step_and_count": by_kernel,
            "note": "one HIP event behind every tagged launch; a lap = the launch + the idle time before it; the events cost a few per cent"}


def state_facts(pd):
    """what the neighbour passes walk in this state: mean compact-list length (8-slot blocks), liquid bounding box"""
    from wcsph_amd import _lib
    import ctypes as C
    ctx, lib = pd._need_ctx(), _lib.load()
    hist = ((C.c_int64 * 64) * 5)()
    _lib.check(ctx, lib.wcsph_list_stats(ctx, hist))
    h = np.array([[hist[a][b] for b in range(64)] for a in range(5)], dtype=np.float64)
    k = np.arange(64)
    pos = pd.pos.to_numpy()[:pd.liquid_count]
    return {"mean_liquid_list_blocks": round(float((h[0] * k).sum() / max(h[0].sum(), 1)), 2), "mean_solid_list_blocks": round(float((h[1] * k).sum() / max(h[1].sum(), 1)), 3),
            "mean_wave_trips": round(float((h[4] * k).sum() / max(h[4].sum(), 1)), 2),
            "liquid_bbox": [[round(float(v), 3) for v in pos.min(0)], [round(float(v), 3) for v in pos.max(0)]]}


def leg(value_n, r, steps):
    return {"value": round(value_n * steps / r["elapsed"] / 1e6, 3), "ms_per_step": round(r["elapsed"] / steps * 1e3, 3)}


def main():
    # stdout carries the ONE JSON line and nothing else: the reference's classes print scene statistics (ParticleData.py:185,
    # HashGrid.py:54), and C++ layers under torch.distributed write to file descriptor 1 directly (gloo's "[Gloo] Rank r is
    # connected to ...", RCCL's version banner at the first communicator) -- for the whole run fd 1 points at stderr, and the line
    # goes to the real stdout at the end
    import contextlib
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    try:
        with contextlib.redirect_stdout(sys.stderr):
            out = _main()
    finally:
        sys.stdout.flush()
        os.dup2(real, 1)
        os.close(real)
    if out is not None:
        print(json.dumps(out), flush=True)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="C2 | C3 | default | an integer particleDim")
    ap.add_argument("--mode", default="faithful", choices=["faithful", "clean"])
    ap.add_argument("--cpu-n", type=int, default=46, help="particleDim of the small CPU sample that rides along")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-clean", action="store_true", help="skip the companion legs (clean mode, fixed iterations, C2, C5)")
    ap.add_argument("--no-developed", action="store_true", help="skip the developed-flow leg (1200 untimed steps)")
    ap.add_argument("--developed-after", type=int, default=1200)
    ap.add_argument("--kernel-reps", type=int, default=20)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        from wcsph_amd import slab
        out = slab.bench_main(args)
        if out is not None:                         # rank 0, after the process group is gone
            from wcsph_amd import scene
            n_base = scene.CONFIGS.get(args.config) or int(args.config)
            if not args.no_pmc:
                # HBM-side traffic of the kernel pair on ONE GPU's load (the per-GPU configuration of this weak-scaling run as a
                # single context on rank 0's GPU; the profiler cannot attach to the ranks)
                traffic, note = measure_traffic_live(n_base, args.mode)
                out["roofline"]["traffic"] = traffic
                out["roofline"]["traffic_source"] = note + " -- taken on one GPU's load (%d^3 particles as a single context), not inside the %d-rank run" % (n_base, world)
            if not args.no_cpu:                     # the same CPU leg as at N = 1
                out["cpu_baseline"] = cpu_baseline(n_base, args.cpu_n)
        return out

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

    r = timed_steps(dfsph, pd, args.warmup, args.steps, tsync, profile=True)
    nl, ns = pd.liquid_count, pd.solid_count
    elapsed = r["elapsed"]
    value = nl * args.steps / elapsed / 1e6
    rest_breakdown = lap_breakdown(dfsph, pd, 10)      # the ten steps behind the timed window, before the replays disturb the state
    rest_state = state_facts(pd)

    # the same kernel pair replayed back to back after the run (a replay finds more of its stream in the Infinity Cache)
    ms_replay = C.c_float()
    _lib.check(ctx, lib.wcsph_time_kernel(ctx, b"pressure_iter", args.kernel_reps, C.byref(ms_replay)))
    s = ns / nl
    alg_bytes = (84.0 + 24.0 * s) * nl
    achieved = alg_bytes / (r["launch_ms"] * 1e-3) / 1e9
    extra = {}
    for name in ("pressure_vel_loop", "pressure_drho_loop", "divergence_iter", "density_coff", "grid_update", "build_lists", "visc_iter"):
        m2 = C.c_float()
        _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), max(args.kernel_reps // 2, 2), C.byref(m2)))
        extra[name + "_us"] = round(m2.value * 1e3, 1)
    # HBM traffic of the pair from the PMC counters, measured in THIS run by child rocprofv3 passes (null when they fail: a recorded
    # figure from another source tree is never substituted)
    traffic, traffic_note = (None, "skipped (--no-pmc)") if args.no_pmc else measure_traffic_live(n, args.mode)
    pipes = {"note": "skipped (--no-pmc)"} if args.no_pmc else measure_pipes_live(n, args.mode)
    it = r["iters"]
    out = {
        "metric": "M particle-steps/sec, DFSPH dam-break", "value": round(value, 3), "unit": "M particle-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "DFSPH 3D dam-break %s: %d^3 = %d liquid + %d boundary particles, CubicKernel, CFL dt, "
                               "viscosity PCG + vorticity on (reference defaults)" % (args.config, n, nl, ns),
                   "state": "steps %d..%d from rest (column not yet collapsed: minimum iteration counts)" % (args.warmup + 1, args.warmup + args.steps),
                   "neighbor_mode": args.mode, "n_liquid": nl, "n_solid": ns,
                   "mean_iters": {"vs": round(it[:, 0].mean(), 2), "dv": round(it[:, 1].mean(), 2), "pr": round(it[:, 2].mean(), 2)},
                   "dt_end": float(dfsph.deltaT[0]), "setup_s": round(t_setup, 2),
                   "breakdown": rest_breakdown, "state_facts": rest_state},
        "roofline": {"bound": "hbm", "kernel": "pressure_iter as the fused solver launches it: k_vel_from_k (kappa += k) + k_drho_press<ITER> (k epilogue), ref dfsph.py:518-547",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_is": "ALGORITHMIC bytes (84 + 24 s) B x n_liquid per launch / measured launch time / 8 TB/s",
                     "traffic": traffic, "traffic_source": traffic_note,
                     "traffic_frac": (round(traffic / (r["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                     "avg_launch_us": round(r["launch_ms"] * 1e3, 1), "launches_timed": r["launches"],
                     "timing": "HIP events around every launch group inside the timed steps", "replay_us": round(ms_replay.value * 1e3, 1),
                     "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_source_sha16": kernel_source_sha16(), "other_kernels": extra,
                     # what the pair actually runs at (it is not HBM-bound: DESIGN.md 4): VALU time, the vector memory pipeline's
                     # instruction floor and its L1 tag look-ups, per launch, measured in this run
                     "valu_us": pipes.get("valu_us"), "gather_floor_us": pipes.get("gather_floor_us"), "l1_lines_per_gather": pipes.get("l1_lines_per_gather"),
                     "l1_lookup_us": pipes.get("l1_lookup_us"), "pipes": pipes},
    }
    pd.close()
    if (args.warmup, args.steps) != (20, 100):
        # SURVEY 8(d)'s own window -- 20 warm-up + 100 timed steps from rest -- beside whatever window this run was asked for: the
        # column starts to collapse inside it (lists grow, neighbourhoods lose their lattice order)
        dfw, pdw, _ = build_sim(n, args.mode)
        rw = timed_steps(dfw, pdw, 20, 100, tsync, profile=True)
        out["config"]["default_window"] = dict(leg(nl, rw, 100), warmup=20, steps=100, state="steps 21..120 from rest",
                                               mean_iters={"vs": round(rw["iters"][:, 0].mean(), 2), "dv": round(rw["iters"][:, 1].mean(), 2), "pr": round(rw["iters"][:, 2].mean(), 2)},
                                               pressure_iter_us=round(rw["launch_ms"] * 1e3, 1),
                                               pressure_iter_frac=round(alg_bytes / (rw["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        out["config"]["default_window"]["breakdown"] = lap_breakdown(dfw, pdw, 20)      # steps 121..140
        out["config"]["default_window"]["state"] = state_facts(pdw)
        pdw.close()
    if not args.no_developed:
        # the same workload after the column has collapsed (t ~ 1.1 s): longer lists, disordered neighbourhoods, a grid
        # box that spans the whole domain -- what a production run spends its time in
        dfd, pdd, _ = build_sim(n, args.mode)
        for _ in range(args.developed_after):
            dfd.step()
        rd = timed_steps(dfd, pdd, 0, 20, tsync, profile=True)
        out["config"]["developed_flow"] = dict(
            leg(nl, rd, 20), after_steps=args.developed_after, t_sim=round(float(dfd.current_time), 3),
            ratio_to_rest=round((rd["elapsed"] / 20) / (elapsed / args.steps), 2),
            mean_iters={"vs": round(rd["iters"][:, 0].mean(), 2), "dv": round(rd["iters"][:, 1].mean(), 2), "pr": round(rd["iters"][:, 2].mean(), 2)},
            pressure_iter_us=round(rd["launch_ms"] * 1e3, 1),
            pressure_iter_frac=round(alg_bytes / (rd["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        # a second window right behind the first (how far two windows of ONE run differ), then 20 steps with the per-group laps
        # (what the step is made of in this state), then the state's own facts
        rd2 = timed_steps(dfd, pdd, 0, 20, tsync, profile=True)
        dev = out["config"]["developed_flow"]
        dev["window_2"] = dict(leg(nl, rd2, 20), steps="%d..%d" % (args.developed_after + 21, args.developed_after + 40),
                               mean_iters={"vs": round(rd2["iters"][:, 0].mean(), 2), "dv": round(rd2["iters"][:, 1].mean(), 2), "pr": round(rd2["iters"][:, 2].mean(), 2)},
                               pressure_iter_us=round(rd2["launch_ms"] * 1e3, 1))
        dev["breakdown"] = lap_breakdown(dfd, pdd, 20)
        dev["state"] = state_facts(pdd)
        pdd.close()
    if args.mode == "faithful" and not args.no_clean:
        # the N > 1 slab runs can use either neighbour semantics; clean mode for a like-for-like reading
        dfsph2, pd2, _ = build_sim(n, "clean")
        out["config"]["clean_mode"] = leg(nl, timed_steps(dfsph2, pd2, args.warmup, args.steps, tsync), args.steps)
        pd2.close()
    if not args.no_clean:
        # equal-work micro-benchmark: fixed dv = 2, pr = 4, vs = 4 iterations per step
        d4, p4, _ = build_sim(n, args.mode)
        nfix = max(args.steps // 2, 5)
        out["config"]["fixed_iters_dv2_pr4_vs4"] = dict(leg(nl, timed_steps(d4, p4, max(args.warmup // 2, 2), nfix, tsync, fixed=(2, 4, 4)), nfix), steps=nfix)
        p4.close()
    if args.config == "C3" and not args.no_clean:
        # BASELINE.json configs[1] (C2, ~100 k particles) for reference: too small to fill 256 CUs, launch-bound
        d3, p3, _ = build_sim(scene.CONFIGS["C2"], args.mode)
        out["config"]["C2_100k"] = dict(leg(p3.liquid_count, timed_steps(d3, p3, args.warmup, args.steps, tsync), args.steps), n_liquid=p3.liquid_count)
        p3.close()
        # BASELINE.json configs[4] (C5): 1 M particles with the Akinci surface tension on (the reference ships 0.0 for both
        # coefficients, which skips the passes) and the adhesion ball of the reference's sphere scene inside the liquid,
        # so that the boundary term (dfsph.py:297-304, gated to that ball) does work
        liq5, sol5 = scene.dam_break_with_ball(scene.CONFIGS["C5"])
        # two semantics of compute_tension (dfsph.py:264-305), both parity-tested against the oracle at this size
        # (tests/test_gpu_configs.py::test_C5_1M_tension_one_step): "literal" = what the reference computes (raw bucket rows,
        # the running normal scaled by h once per entry, Q8) -- the default of faithful mode; "fixed" = the repaired normals
        for key, fixed in (("C5_1M_tension", 0 if args.mode == "faithful" else 1), ("C5_1M_tension_fixed", 1)):
            d5, p5, _ = build_sim(scene.CONFIGS["C5"], args.mode, liquid=liq5, boundary=sol5, tension_coff=0.05, tension_coff_b=0.05, tension_fixed=fixed)
            out["config"][key] = dict(leg(p5.liquid_count, timed_steps(d5, p5, args.warmup, args.steps, tsync), args.steps),
                                      n_liquid=p5.liquid_count, n_solid=p5.solid_count, tension_coff=0.05, tension_coff_b=0.05,
                                      tension_semantics="reference-literal loop over the raw bucket rows (tension_fixed=0)" if fixed == 0
                                      else "repaired normals over the compact lists (tension_fixed=1)")
            p5.close()
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(n, args.cpu_n)
    return out


if __name__ == "__main__":
    main()
