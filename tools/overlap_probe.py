#!/usr/bin/env python3
"""GPU box helper: the block-encode kernel ALONE against the same kernel BESIDE the tail kernels of the part in front.
Phase A: ten calls of 16 4K frames (one part: its tail kernels run after it, nothing beside the encode kernel).
Phase B: four calls of 128 frames (eight parts: every encode launch but the first runs beside the previous part's
k_dc_heads / k_tile_scan / k_merge on the library's side stream).  Run under `rocprofv3 --kernel-trace`; the summary
(tools/overlap_summary.py) splits the k_screen_encode launches of the trace by phase: 2 + 10 launches of phase A first,
then 8 x (1 + 4) of phase B."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch  # noqa: E402

W, H = 3840, 2160
enc = jpeg.Encoder(0)
enc.set_quality(50)
dev = torch.device("cuda", 0)
d_rgb = torch.empty((128, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, 128, 1)
cap = 8 << 20
d_out = torch.zeros((128, cap), dtype=torch.uint8, device=dev)
d_bits = torch.zeros(128, dtype=torch.int64, device=dev)
for n, reps in ((16, 12), (128, 5)):
    for _ in range(reps):
        enc.encode_scan_device(d_rgb.data_ptr(), W, H, n, d_out.data_ptr(), cap, d_bits.data_ptr())
    enc.sync()
    torch.cuda.synchronize()
print("bits of frame 0:", int(d_bits[0]))
enc.close()
