#!/bin/bash
# GPU box helper (round 3): phase stamps of k_encode_tile (diagnostic build libmi355jpeg_stamps.so), one frame and a batch
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for n in 1 128; do
MI355_JPEG_LIB=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd/libmi355jpeg_stamps.so MI355_JPEG_DUMP_STAMPS=1 timeout -k 10 300 python - $n <<'PY' 2>&1 | tee -a "$OUT/stamps.log"
import importlib, sys, torch
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
n = int(sys.argv[1]); W, H = 3840, 2160
enc = jpeg.Encoder(0); dev = torch.device("cuda", 0)
d_rgb = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, n, 1)
cap = 8 << 20
d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev); d_bits = torch.zeros(n, dtype=torch.int64, device=dev)
for _ in range(3):
    enc.encode_scan_device(d_rgb.data_ptr(), W, H, n, d_out.data_ptr(), cap, d_bits.data_ptr())
    torch.cuda.synchronize()
print("frames", n, flush=True)
enc.sync()
PY
done
