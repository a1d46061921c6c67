#!/bin/bash
# GPU box helper: phase stamps of standard 4:2:0 / 4:4:4, colour conversion on the matrix units vs VALU (STAMPS=1 builds)
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for FL in 6 2; do
  echo "== flags $FL valu" >> "$OUT/stamps.log"
  MI355_DIAG_FLAGS=$FL MI355_JPEG_LIB=$P/libmi355jpeg_stamps_valu.so python tools/stamps.py >> "$OUT/stamps.log" 2>&1
  echo "== flags $FL mfma" >> "$OUT/stamps.log"
  MI355_DIAG_FLAGS=$FL MI355_JPEG_LIB=$P/libmi355jpeg_stamps.so python tools/stamps.py >> "$OUT/stamps.log" 2>&1
done
grep -E "==|cycles per wave" "$OUT/stamps.log"
