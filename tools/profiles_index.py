#!/usr/bin/env python3
"""Regenerates profiles/README.md: a hand-written "read first" block + every file grouped by round and purpose (by name pattern).
   python tools/profiles_index.py"""
import collections, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
files = sorted(f for f in os.listdir(P) if f != "README.md")
KINDS = [
    (r"_experiments?\.md$", "experiment write-up (what was tried, counters, verdict)"),
    (r"bench.*under_rocprof|bench_rocprof_stats", "the bench command itself under `rocprofv3 --kernel-trace --stats`"),
    (r"bench_two_ranks", "`bench.py --gpus 2`, both ranks on ONE GPU, halos host-staged over gloo (plumbing, not a scaling number)"),
    (r"slab_nccl_world1", "the slab bench path on the real RCCL backend with one rank"),
    (r"bench", "bench line (`python bench.py ...`), `roofline` + `cpu_baseline` inside"),
    (r"kernel_stats", "per-kernel in-situ durations (`rocprofv3 --kernel-trace`, `tools/kstats.sh` / `tools/rocpd_stats.py`)"),
    (r"pmc_traffic|pmc_grid_traffic", "HBM-side traffic passes (FETCH_SIZE / WRITE_SIZE, own `--pmc` runs, `tools/pmc_traffic.sh`)"),
    (r"pmc", "PMC counter passes (`tools/pmc_passes.sh`, `tools/pmc_tile.sh`, `tools/pmc_build_lists.sh`)"),
    (r"gputests", "`pytest -m gpu` log"),
    (r"slab_overhead|slab_world1|redistribute", "slab (multi-GPU) driver on one GPU: overhead over the fused step"),
    (r"^r\d+[a-z]*_ab_", "A/B of two builds on one box (`tools/ab_libs.sh`)"),
    (r"long_run|soak", "long-run / soak test"),
    (r"size_sweep|sweep", "parameter sweep"),
    (r"gather_rate|launch_rate|mall_policy|busy|idle", "micro-benchmark (`tools/micro/*`) or GPU-idle analysis"),
    (r"gather_model|list_balance|stress_lists|wall_density|developed_parity|ablation", "model / diagnostic output"),
    (r"c5", "config C5 (tension on) kernel statistics"),
    (r"tile", "LDS tile walk experiment (round 5)"),
    (r"r06_(overlap|cluster_pair|developed)\.md$", "round-6 write-up (what was built, measured, concluded)"),
    (r"hw_queues", "stream / hardware-queue hazard of the sixth context of a process (`r06_overlap.md`)"),
    (r"step_boundary", "kernel trace of the step boundary of the periodic ring (`tools/r06_step_boundary.sh`, `rocpd_step_boundary.py`)"),
    (r"smoke", "`__graft_entry__.smoke()` log"),
    (r"migrate_plan", "`pytest -m gpu tests/test_gpu_migrate_plan.py` log"),
    (r"overlap|ring_gaps|ring", "halo / compute overlap on the loop-back ring of C4's per-rank shape (`tools/overlap_ring.py`, `rocpd_overlap.py`, `rocpd_ring_steps.py`)"),
    (r"prio_probe", "micro-benchmark (`tools/micro/*`) or GPU-idle analysis"),
    (r"cpair", "cluster-pair walk experiment (round 6): PMC passes"),
    (r"live_parity", "live-state parity numbers (`pytest -s`: flip candidates, quantiles, slabs against one context)"),
]


def kind(f):
    for rx, k in KINDS:
        if re.search(rx, f):
            return k
    return "other"


HEAD = """# profiles/ -- index by round and purpose

Every file is the output of a command run on an MI355X box of the pool (`gpurun`), copied here from the scratch directory
`gpurun_out/`.  Names: `r<round><campaign letter>_<what>`; a later letter is a later campaign of the round, the LAST letter of a
round ran on (or nearest to) its final source.  Boxes of the pool differ by +-3 % on the same code, so A/B rows always compare
libraries within ONE `gpurun` call.  C3 = 126^3 = 2 000 376 liquid + 1 008 602 boundary particles, faithful mode, unless a file
says otherwise.  (Regenerate with `python tools/profiles_index.py`.)

## Read first (round 6)

| file | what it shows |
|---|---|
| `r06_cluster_pair.md` (+ `r06_pmc_cpair.txt`) | the list-free cluster-pair walk VERDICT r05 asked for -- clusters of 4 / 2 consecutive sorted particles, j-windows, DPP sums -- built equal to rounding and measured with counters: 132 vs 86 us at rest, 256 vs 204 us developed (gate: 155); where it loses |
| `r06_overlap.md` (+ `r06z_overlap.md`, `r06z_overlap_ring.json`, `r06z_ring_gaps_*.txt`, `r06z_overlap_*_kernel_stats.md`, `r06z_prio_probe.txt`) | halo refreshes of the native RCCL plane on their own stream: kernel trace of a loop-back ring of C4's per-rank shape (95 % of the RCCL time inside a compute kernel), the price list (split +253 us, event hop 12 us, RCCL launch 16 us, RCCL's kernel beside a saturating grid), every schedule timed, why the choice is measured at run time |
| `r06_developed.md` (+ `r06z_bench_20_5.json`: `config.developed_flow.breakdown / window_2 / state`) | the developed-flow leg made explainable: two windows, per-group in-situ breakdown (lap events), state facts; what can and cannot explain round 5's 553 M |
| `r06z_bench_20_5.json`, `r06z_bench_default.json`, `r06z_rest_kernel_stats.md`, `r06z_dev_kernel_stats.md`, `r06z_bench_rocprof_stats.md`, `r06z_pmc_*` | the campaign on the final source (`tools/campaign_r06.sh`): 1254 M particle-steps/s (1.595 ms/step), 1016-1021 M over the default window, 605 M developed; `pressure_iter` 136.6 us = 0.176 |
| `r06z_slab_overhead.txt`, `r06z_slab_nccl_world1.json`, `r06z_bench_two_ranks_one_gpu_gloo.json` | slab driver on one GPU: +0.010 ... 0.014 ms (clean) over the fused step on either plane; the bench path on the real RCCL backend with one rank incl. the schedule tuning; two ranks on one GPU over gloo (2.2 host waits per step, stdout = the one JSON line) |
| `r06z_live_parity.log`, `r06z_gputests.log` | the live-state parity numbers themselves (flip candidates, quantiles, 3 slabs and C4's 8 slabs against one context, native plane schedules against the callback plane); `pytest -m gpu` on the final source: 136 passed, 1 skipped |
| `r06f_*` (same file names as `r06z_*`), `r06f_ring_real.json`, `r06f_ring_real_nosync.json`, `r06f_migrate_plan_tests.log` | the campaign repeated on the round's FINAL source (migration + ghost rebuild inside the library, `csrc/api_migrate.h`): 1242 M / 1012-1018 M / 605 M, `pressure_iter` 137.9 us = 0.174, 145 GPU tests passed; the periodic ring driven by the real migration code: host waits per step 2.0 -> 1.0, redistribute 0.25 ms (torch plan) -> 0.145 (library plan behind the caller's exchange) -> 0.107 (`wcsph_slab_migrate` over RCCL), every variant bit-identical; `_nosync`: the same steps without the measurement's synchronisations (2.87 -> 2.67 ms per step) |
| `r06h_*` | the last campaign of the round (`tools/campaign_r06_final.sh`) after the step boundary's clean-up (lazy halo-list views, sorted-order classification, no fill launch): 1257 M / 1017 M / 609 M, `pytest -m gpu` twice 146 passed; ring with the real migration 2.62 ms per step on the native inline plane against 2.84 (torch plan, callbacks); `r06h_step_boundary.txt`, `r06h_sixth_context_hw_queues.json` (a stream / hardware-queue hazard of processes that create six contexts, see `r06_overlap.md`) |
| `r06_compiler_sched_flags.txt` | seven LLVM scheduling strategies / biases on the whole library: none helps |
| `r05_lds_tile_experiment.md`, `r04_experiments.md`, `r03_experiments.md`, `r02b_wave_brick_experiment.md` | every earlier measured-and-rejected idea with its counters |

## By round
"""
by_round = collections.OrderedDict()
for f in files:
    m = re.match(r"r(\d+)", f)
    r = int(m.group(1)) if m else 0
    by_round.setdefault(r, collections.OrderedDict()).setdefault(kind(f), []).append(f)
out = [HEAD]
for r, kd in by_round.items():
    out.append("### %s\n" % ("unnumbered" if r == 0 else "round %d" % r))
    out.append("| purpose | files |\n|---|---|")
    for k, fs in kd.items():
        out.append("| %s | %s |" % (k, " ".join("`%s`" % f for f in fs)))
    out.append("")
open(os.path.join(P, "README.md"), "w").write("\n".join(out) + "\n")
print("profiles/README.md: %d files" % len(files))
