#!/usr/bin/env python3
"""The slab bench path with the REAL backend string ("nccl" = RCCL) on a one-rank process group: exercises communicator
set-up, device-tensor all-reduces and the stream hand-over of SlabSim on a single GPU (no neighbour to talk to).
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/slab_nccl_world1.py [n] [steps]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import slab
a = argparse.Namespace(gpus=1, steps=int(sys.argv[2]) if len(sys.argv) > 2 else 10, warmup=3, config=sys.argv[1] if len(sys.argv) > 1 else "60", mode="faithful")
os.environ.setdefault("WCSPH_BENCH_COMM", "nccl")
out = slab.bench_main(a)
if out is not None:
    print(json.dumps(out))
