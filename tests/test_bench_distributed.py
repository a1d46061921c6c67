"""The N>1 path of bench.py (one process per GPU, frames sharded over ranks, barrier +
max-over-ranks timing, no data-path collective) rehearsed on CPU: world_size 2, gloo, with
the oracle as the worker on tiny frames (bench.py --dry-run-cpu, test-only).

Two ways in, same worker code: `python bench.py --gpus N` alone (bench.py starts its own N worker
processes before anything touches a GPU -- the shape of the command the driver runs) and under
torchrun (RANK/WORLD_SIZE already in the environment)."""
import json
import os
import subprocess
import sys

import oracle_lib as ol
from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
ARGS = ["--steps", "3", "--warmup", "1", "--frames-per-step", "3", "--dry-run-cpu"]


def parse(out):
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def run_launcher_free(nproc):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, BENCH, "--gpus", str(nproc)] + ARGS
    return parse(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300))


def run_torchrun(nproc, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", str(nproc)] + ARGS
    return parse(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300))


def check_ranks(r, n):
    assert r["n_gpus"] == n and r["ranks_seen"] == n and r["scaling"] == "weak" and r["steps"] == 3
    assert r["seed0_per_rank"] == [1 + 3 * k for k in range(n)]   # rank 0: seeds 1..3, rank 1: seeds 4..6, ...
    assert len(r["per_rank_mpixel_s"]) == n and all(v > 0 for v in r["per_rank_mpixel_s"])
    # every rank really encoded its own frames: bit counts equal the oracle's for those seeds
    for rank, bits in enumerate(r["bits_per_rank"]):
        for k, b in enumerate(bits):
            assert b == ol.oracle_encode(ol.lcg_frame(64, 48, 1 + rank * 3 + k)).n_bits
    assert r["value"] > 0
    # the N > 1 line is complete (cpu_baseline and the PCIe-inclusive leg were emitted at N = 1 only)
    assert r["cpu_baseline"]["kind"] in ("port", "reference") and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["value"] > 0
    assert r["end_to_end"]["gpus"] == n


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2`: bench.py spawns its two workers itself."""
    r = run_launcher_free(2)
    assert r["launcher"] == "bench.py"
    check_ranks(r, 2)


def test_two_ranks_under_torchrun():
    r = run_torchrun(2, 29731)
    assert r["launcher"] == "external"
    check_ranks(r, 2)


def test_eight_ranks_without_a_launcher():
    """The shape of the driver's 8-GPU run: eight worker processes, frames sharded 8 ways, the
    barrier and the max over eight ranks, one line with ranks_seen == 8."""
    r = run_launcher_free(8)
    assert r["launcher"] == "bench.py"
    check_ranks(r, 8)


def test_eight_ranks_under_torchrun():
    r = run_torchrun(8, 29741)
    assert r["launcher"] == "external"
    check_ranks(r, 8)


def test_single_rank_same_code_path():
    r = run_launcher_free(1)
    assert r["n_gpus"] == 1 and r["seed0_per_rank"] == [1] and r["ranks_seen"] == 1


def test_a_dead_rank_fails_the_run():
    """A worker that exits non-zero makes `bench.py --gpus N` exit non-zero (and not hang)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["MI355_BENCH_TEST_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dead-rank-grace", "5"] + ARGS, env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])


def test_secondary_legs_are_emitted_at_every_n():
    """cpu_baseline and end_to_end belong to the line at every N (the dry run emits the same keys through the same
    rank-0 code path; the GPU worker's N > 1 form of end_to_end -- rank 0 driving every GPU of the run through one pool
    while the other ranks wait on the gloo tail group -- has not run on hardware yet: DESIGN.md §5 says so)."""
    for n in (1, 2):
        r = run_launcher_free(n)
        assert "cpu_baseline" in r and "end_to_end" in r and r["end_to_end"]["gpus"] == n
