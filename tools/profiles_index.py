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

## Read first (round 5)

| file | what it shows |
|---|---|
| `r05_lds_tile_experiment.md` (+ `r05_tile_*`, `r05_pmc_build_lists.txt`, `r05_ab_flat_list_build.txt`) | (1) the LDS design VERDICT r04 asked for -- compact tiles, LDS-DMA staging, 16-bit LDS-slot lists -- built bit-exact and measured with counters: 191 vs 135 us at rest, 417 vs 326 us developed; where it loses.  (2) the list build in the developed flow: divergence-bound (34 of 64 lanes active), a flattened candidate loop built and 26-32 % slower |
| `r05z_bench_20_5.json`, `r05z_bench_default.json`, `r05z_rest_kernel_stats.md`, `r05z_dev_kernel_stats.md`, `r05z_bench_rocprof_stats.md`, `r05z_pmc_*` | the campaign on the final source (`tools/campaign_r05.sh`): 1237-1248 M particle-steps/s (1.60-1.62 ms/step), 1018-1021 M over the default window, 607-610 M developed; DESIGN 4's table is generated from the ONE kernel trace here |
| `r05_slab_overhead.txt`, `r05z_slab_overhead.txt`, `r05z_slab_nccl_world1.json` | slab driver on one GPU: +0.115 ms over the fused step (state store 0.052 + load 0.047) -> **+0.013 ... 0.018 ms** with the state resident in the library; the communicator called at every exchange point (Python callbacks or the native RCCL plane) adds 0.00-0.04 ms; the bench path on the real RCCL backend with one rank and the native data plane: 1184-1191 M |
| `r05_c5.md` (+ `r05_c5_*_kernel_stats.md`) | C5 literal / repaired tension legs on the final source |
| `r05z_gputests.log` | `pytest -m gpu` on the final source: 130 passed, 1 skipped |
| `r04_experiments.md`, `r03_experiments.md`, `r02b_wave_brick_experiment.md` | every earlier measured-and-rejected idea with its counters (the stop list of VERDICT r04 comes from these) |

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
