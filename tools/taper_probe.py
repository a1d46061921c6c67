#!/usr/bin/env python3
"""Development helper (GPU box): 128-frame 4K batches at several qualities and modes, Gpixel/s each (run under different
MI355_JPEG_TAPER values: how far from the cliff is a taper ratio when the merge / encode time ratio changes?)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch  # noqa: E402

W, H, N = 3840, 2160, 128
dev = torch.device("cuda", 0)
enc = jpeg.Encoder(0)
d = torch.empty((N, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d.data_ptr(), W * H * 3, N, 1)
cap = 16 << 20
o = torch.zeros((N, cap), dtype=torch.uint8, device=dev)
b = torch.zeros(N, dtype=torch.int64, device=dev)
res = []
for name, q, flags in (("strict q50", 50, jpeg.F_DEFAULT), ("strict q75", 75, jpeg.F_DEFAULT), ("strict q90", 90, jpeg.F_DEFAULT),
                       ("strict q25", 25, jpeg.F_DEFAULT), ("std444 q50", 50, jpeg.F_STANDARD), ("std420 q50", 50, jpeg.F_STANDARD | jpeg.F_420),
                       ("std420 q90", 90, jpeg.F_STANDARD | jpeg.F_420)):
    enc.set_quality(q)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        enc.encode_scan_device(d.data_ptr(), W, H, N, o.data_ptr(), cap, b.data_ptr(), flags=flags)
        enc.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        enc.encode_scan_device(d.data_ptr(), W, H, N, o.data_ptr(), cap, b.data_ptr(), flags=flags)
    enc.sync()
    dt = (time.perf_counter() - t0) / reps
    res.append("%s %.1f" % (name, N * W * H / dt / 1e9))
print("TAPER=%-3s " % os.environ.get("MI355_JPEG_TAPER", "-") + " | ".join(res), flush=True)
