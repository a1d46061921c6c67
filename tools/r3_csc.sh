#!/bin/bash
# GPU box helper (round 3): standard mode with its colour conversion on the matrix units (libmi355jpeg.so) against the
# VALU build of the same definition (libmi355jpeg_valu.so = make EXTRA=-DMI355_STD_CSC_VALU): parity suite of the mode
# for both, then interleaved timing (4:2:0 and 4:4:4, 128 and 1 frames per call).
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
timeout -k 10 900 python -m pytest tests/test_standard_mode.py -m gpu -x -q > "$OUT/pytest_std_mfma.log" 2>&1 || { tail -40 "$OUT/pytest_std_mfma.log"; exit 1; }
tail -2 "$OUT/pytest_std_mfma.log"
MI355_JPEG_LIB=$P/libmi355jpeg_valu.so timeout -k 10 900 python -m pytest tests/test_standard_mode.py -m gpu -x -q > "$OUT/pytest_std_valu.log" 2>&1 || { tail -40 "$OUT/pytest_std_valu.log"; exit 1; }
tail -2 "$OUT/pytest_std_valu.log"
for r in 1 2; do
  for FL in 6 2; do
    MI355_DIAG_CAP_MB=16 MI355_DIAG_FLAGS=$FL MI355_JPEG_LIB=$P/libmi355jpeg_valu.so python tools/pipeline_diag.py 128 | sed "s/^/valu /" | tee -a "$OUT/csc.log"
    MI355_DIAG_CAP_MB=16 MI355_DIAG_FLAGS=$FL python tools/pipeline_diag.py 128 | sed "s/^/mfma /" | tee -a "$OUT/csc.log"
  done
done
