#!/bin/bash
# Round-6 measurement campaign on the GPU box: the driver's bench line (with the per-group breakdown of the rest / default-window /
# developed legs and its two developed windows), ONE in-situ kernel trace at rest and in the developed flow, PMC passes of the
# pressure pair, the loop-back ring of C4's per-rank shape in every halo schedule (timings, overlap trace, idle gaps), the slab
# driver at world = 1, two ranks on the one GPU over gloo, the real RCCL backend with one rank, the bench command itself under
# rocprofv3 --stats, the GPU test suite.
# usage: tools/campaign_r06.sh <tag>      (outputs under gpurun_out/<tag>_*)
TAG=${1:-r06z}; R=/root/repo; O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_20_5.json 2> $O/${TAG}_bench_20_5.err
python bench.py --no-clean --no-developed > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
for m in clean faithful; do python tools/slab_comm_overhead.py 126 $m 2>/dev/null | grep "^{"; done > $O/${TAG}_slab_overhead.txt
WCSPH_BENCH_COMM=gloo MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 10 --warmup 3 --config 80 --no-cpu --no-pmc > $O/${TAG}_bench_two_ranks_one_gpu_gloo.json 2> $O/${TAG}_bench_two_ranks.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/slab_nccl_world1.py 126 20 2>/dev/null | grep "^{" > $O/${TAG}_slab_nccl_world1.json
# the ring: every schedule, both neighbour modes; then the trace of the default schedule and the idle gaps of four of them
( cd /tmp && export TMPDIR=/tmp
  for m in clean faithful; do python $R/tools/overlap_ring.py 32 252 252 $m 20 2>/dev/null | grep "^{"; done > $O/${TAG}_overlap_ring.json )
( cd /tmp && export TMPDIR=/tmp
  python $R/tools/ring_real.py 32 252 252 20 callbacks_torchplan,callbacks,callbacks_nosplit,native0_hdr,native0,native1,native2 2.0 2>/dev/null | grep "^{" > $O/${TAG}_ring_real.json )   # (clean mode: the faithful mode's global bucket table needs real ranks)
bash $R/tools/r06_overlap.sh $TAG clean
bash $R/tools/r06_overlap_gaps.sh $TAG clean
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace -d /tmp/prof -o rest -- python $R/tools/run_steps.py 45 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/prof/rest_results.db 0.6 > $O/${TAG}_rest_kernel_stats.md
rocprofv3 --kernel-trace -d /tmp/prof -o dev -- python $R/tools/run_steps.py 1240 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/prof/dev_results.db 0.985 > $O/${TAG}_dev_kernel_stats.md
rm -rf /tmp/prof
bash $R/tools/pmc_traffic.sh pressure_iter "k_vel_from_k|k_drho_press" $O/${TAG}_pmc_traffic.txt
bash $R/tools/pmc_passes.sh pressure_iter "k_vel_from_k|k_drho_press" $O/${TAG}_pmc_pressure_iter.txt
rocprofv3 --kernel-trace --stats -d /tmp/profb -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-clean --no-developed > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py /tmp/profb/bench_results.db 0.0 > $O/${TAG}_bench_rocprof_stats.md 2>/dev/null
rm -rf /tmp/profb
$R/tools/micro/prio_probe f > $O/${TAG}_prio_probe.txt 2>&1
cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_gputests.log 2>&1
# the live-state parity numbers themselves (flip candidates, quantiles, slabs against one context)
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_slab_live.py tests/test_gpu_slab_c4.py -m gpu -q -s -k "live or perturbed" 2>&1 | grep -i "live\|step \|slabs vs\|vs callbacks\|passed\|failed\|flip" > $O/${TAG}_live_parity.log
