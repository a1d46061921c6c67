"""The N>1 path of bench.py (one process per GPU, frames sharded over ranks, barrier +
max-over-ranks timing, no data-path collective) rehearsed on CPU: world_size 2, gloo, with
the oracle as the worker on tiny frames (bench.py --dry-run-cpu, test-only)."""
import json
import os
import subprocess
import sys

import oracle_lib as ol
from conftest import ROOT


def run_bench(nproc, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "6", "--warmup", "2", "--ring", "3", "--dry-run-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_shard_frames_without_overlap():
    r = run_bench(2, 29731)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 6
    assert r["seed0_per_rank"] == [1, 4]          # rank 0: seeds 1..3, rank 1: seeds 4..6
    # every rank really encoded its own frames: bit counts equal the oracle's for those seeds
    for rank, bits in enumerate(r["bits_per_rank"]):
        for k, b in enumerate(bits):
            assert b == ol.oracle_encode(ol.lcg_frame(64, 48, 1 + rank * 3 + k)).n_bits
    assert r["value"] > 0


def test_single_rank_same_code_path():
    r = run_bench(1, 29732)
    assert r["n_gpus"] == 1 and r["seed0_per_rank"] == [1]
