#!/bin/bash
# GPU box helper: usage tools/gpu_round.sh <tag> [steps...]; steps: test bench prof cfg
# Logs under gpurun_out/<tag>/.  Steps are joined with && (a failed GPU step stops the call).
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for STEP in "$@"; do
  case $STEP in
    test)  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }; tail -3 "$OUT/pytest_gpu.log" ;;
    bench) timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }; cat "$OUT/bench.json" ;;
    quick) timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --quick --no-cpu-baseline > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err" || { tail -30 "$OUT/bench_quick.err"; exit 1; }; cat "$OUT/bench_quick.json" ;;
    prof)  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/prof_bench.json" 2> "$OUT/prof.err" || { tail -30 "$OUT/prof.err"; exit 1; }
           find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"; head -12 "$OUT/kernel_stats.csv"
           find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete ;;
    cfg)   timeout -k 10 600 python tools/config_bench.py > "$OUT/config_bench.log" 2>&1 || { tail -30 "$OUT/config_bench.log"; exit 1; }; cat "$OUT/config_bench.log" ;;
    smoke) timeout -k 10 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1 || { tail -30 "$OUT/smoke.log"; exit 1; }; tail -2 "$OUT/smoke.log" ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
done
