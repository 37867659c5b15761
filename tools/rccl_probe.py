#!/usr/bin/env python3
"""Can two ranks of an RCCL (backend "nccl") process group share ONE GPU on this pool?  (all_reduce + grouped send/recv)
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_probe.py
Answer on the single-GPU boxes of the pool (round 3, RCCL 2.26.6): NO -- ncclInvalidUsage, "Duplicate GPU detected : rank 0 and
rank 1 both on CUDA device 23000" at init_process_group.  Hence the two-rank GPU tests run over gloo with the halos staged through
the host (tests/test_gpu_slab.py, WCSPH_BENCH_COMM=gloo), and the nccl branch has executed with one rank only."""
import os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.full((4,), float(rank + 1), device=dev)
dist.all_reduce(t)
print("rank", rank, "all_reduce", t.tolist(), flush=True)
a = torch.full((8, 3), float(rank), device=dev); b = torch.empty((8, 3), device=dev)
peer = 1 - rank
for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, peer)]):
    r.wait()
torch.cuda.synchronize()
print("rank", rank, "recv", b[0].tolist(), flush=True)
dist.barrier(); dist.destroy_process_group()
