"""bench.py prints exactly one JSON line with the contract's keys (driver contract) -- run on a tiny scene."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "12", "--steps", "3", "--warmup", "1",
                        "--cpu-n", "8", "--developed-after", "30", "--kernel-reps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "M particle-steps/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 0
    assert rf["launches_timed"] >= 2 * d["steps"] and rf["avg_launch_us"] > 0       # in-situ events: >= 2 pressure iterations per step
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert "developed_flow" in d["config"] and d["config"]["developed_flow"]["value"] > 0
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert d["value"] > 0 and abs(d["ms_per_step"] - d["config"]["n_liquid"] / d["value"] / 1e3) / d["ms_per_step"] < 0.02


def test_two_rank_bench_line_over_gloo():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on the
    one GPU of the test box and the halos staged through the host (WCSPH_BENCH_COMM=gloo): rank 0 prints ONE JSON line
    for the whole job, weak scaling, faithful neighbour semantics across the ranks."""
    env = dict(os.environ, WCSPH_BENCH_COMM="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "24",
                        "--cpu-n", "8"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout of the whole job is the ONE line: nothing of gloo's / RCCL's own chatter (they write to fd 1 directly) may reach it
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["neighbor_mode"] == "faithful" and sum(d["config"]["owned_per_rank"]) == d["config"]["n_liquid"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    assert d["roofline"]["launches_timed"] >= 2 * d["steps"]            # HIP events around every pressure_iter of the timed steps
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0     # measured once, on rank 0's host
    # 4 decision points inside the step (divergence, viscosity, CFL, pressure) + 2 in the migration / ghost rebuild (counts)
    assert d["config"]["host_waits_per_step"] <= 6.5 and d["config"]["step_driver"].startswith("wcsph_slab_step")


def test_two_rank_bench_survives_an_rccl_that_refuses():
    """The default launch (RCCL data plane) with both ranks on the ONE GPU of the test box: RCCL refuses that at group creation
    ("Duplicate GPU detected") on every rank -- a real RCCL failure, not a simulated one.  The ranks agree over the gloo
    control plane, the same HIP computation runs with the halos staged through the host, and rank 0 still prints ONE JSON
    line that says which transport carried the halos."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", WCSPH_BENCH_LOCAL_GPU="0")
    env.pop("WCSPH_BENCH_COMM", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "24",
                        "--cpu-n", "8"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and sum(d["config"]["owned_per_rank"]) == d["config"]["n_liquid"]
    assert d["config"]["comm"].startswith("gloo fallback") and "RCCL probe failed" in r.stderr
