#!/bin/bash
# GPU box helper (round 3): full GPU suite, the driver's bench command, 2-rank rehearsals (RCCL first, gloo as the fallback)
set -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
# RCCL refuses two ranks on one GPU ("Duplicate GPU detected"): its path (init with device_id, barrier, all_gather, gloo tail group) runs with ONE rank
timeout -k 10 300 python bench.py --gpus 1 --rehearse-dist --backend nccl --steps 5 --warmup 2 --quick --no-cpu-baseline > "$OUT/bench_1rank_rccl.json" 2> "$OUT/bench_1rank_rccl.err"
echo "rccl 1-rank rc=$?"; tail -c 600 "$OUT/bench_1rank_rccl.json"; tail -3 "$OUT/bench_1rank_rccl.err"
timeout -k 10 300 python bench.py --gpus 2 --share-device --backend gloo --steps 5 --warmup 2 --frames-per-step 32 > "$OUT/bench_2ranks_gloo.json" 2> "$OUT/bench_2ranks_gloo.err"
echo "gloo 2-rank rc=$?"; tail -c 3000 "$OUT/bench_2ranks_gloo.json"
