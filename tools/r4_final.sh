#!/bin/bash
# GPU box helper (round 4): the round's evidence in one call.  usage: tools/r4_final.sh <tag> [steps...]
# (the pmc, cfgpmc and overlap steps run with MI355_JPEG_TAPER=0: equal parts of 16 frames, which their summaries assume)
# steps: test bench prof pmc cfgpmc overlap stamps   (default: all but stamps, which needs libmi355jpeg_stamps.so = a `make STAMPS=1` build).  Steps are joined with && semantics (set -e).
set -e -o pipefail
TAG=$1; shift
STEPS=${@:-test bench prof pmc cfgpmc overlap}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for STEP in $STEPS; do
  case $STEP in
    test)  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }; tail -2 "$OUT/pytest_gpu.log" ;;
    bench) timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }; cut -c1-600 "$OUT/bench.json" ;;
    prof)  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/prof_bench.json" 2> "$OUT/prof.err" || { tail -30 "$OUT/prof.err"; exit 1; }
           find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"; head -8 "$OUT/kernel_stats.csv"
           python3 tools/profile_summary.py "$OUT" "$OUT/driver_cmd_summary.json" | tail -5
           find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc)   tools/pmc_run.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1 || { tail -20 "$OUT/pmc.log"; exit 1; }; grep -A30 "^k_screen_encode" "$OUT/pmc/summary.txt" | head -32
           find "$OUT/pmc" -name "*.csv" -size +8M -delete ;;
    cfgpmc)
      for C in c2 c4; do
        i=0
        for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
                 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "FETCH_SIZE" "WRITE_SIZE"; do
          MI355_JPEG_TAPER=0 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc_$C/pass$i" -- python3 tools/config_bench.py $C > "$OUT/pmc_$C.pass$i.log" 2>&1
          i=$((i+1))
        done
        python3 tools/pmc_summary.py "$OUT/pmc_$C" > "$OUT/pmc_$C.summary.txt"
        echo "== $C"; grep -A22 "^k_screen_encode" "$OUT/pmc_$C.summary.txt" | head -24
        find "$OUT/pmc_$C" -name "*.csv" -size +8M -delete
      done ;;
    overlap)
      MI355_JPEG_TAPER=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/overlap" -- python3 tools/overlap_probe.py > "$OUT/overlap.log" 2>&1 || { tail -20 "$OUT/overlap.log"; exit 1; }
      python3 tools/overlap_summary.py "$OUT/overlap" "$OUT/overlap_summary.json" ;;
    stamps)
      MI355_JPEG_LIB=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd/libmi355jpeg_stamps.so python3 tools/stamps.py > "$OUT/stamps.txt" 2>&1 || { tail -20 "$OUT/stamps.txt"; exit 1; }; tail -8 "$OUT/stamps.txt" ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
done
