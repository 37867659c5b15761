TAG=r06h; R=/root/repo; O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_20_5.json 2> $O/${TAG}_bench_20_5.err
for m in clean faithful; do python tools/slab_comm_overhead.py 126 $m 2>/dev/null | grep "^{"; done > $O/${TAG}_slab_overhead.txt
WCSPH_BENCH_COMM=gloo MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 10 --warmup 3 --config 80 --no-cpu --no-pmc > $O/${TAG}_bench_two_ranks_one_gpu_gloo.json 2> $O/${TAG}_bench_two_ranks.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/slab_nccl_world1.py 126 20 2>/dev/null | grep "^{" > $O/${TAG}_slab_nccl_world1.json
( cd /tmp && export TMPDIR=/tmp
  python $R/tools/ring_real.py 32 252 252 20 callbacks_torchplan,callbacks,callbacks_nosplit,native0_hdr,native0,native1,native2 2.0 2>/dev/null | grep "^{" > $O/${TAG}_ring_real.json
  RING_NOSYNC=1 python $R/tools/ring_real.py 32 252 252 20 callbacks_torchplan,callbacks_nosplit,native0_hdr,native0,native1 2.0 2>/dev/null | grep "^{" > $O/${TAG}_ring_real_nosync.json
  for m in clean; do python $R/tools/overlap_ring.py 32 252 252 $m 20 2>/dev/null | grep "^{"; done > $O/${TAG}_overlap_ring.json )
rm -f $O/${TAG}_step_boundary.txt; bash tools/r06_step_boundary.sh $TAG
cd $R
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^serach\|^liqiud" | tail -12 > $O/${TAG}_gputests_$i.log; done
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
