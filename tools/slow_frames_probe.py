#!/usr/bin/env python3
"""Development helper (GPU box): are some frames of bench.py's resident batch slower to encode than others?  (The kernel trace of
the driver's command shows whichever launch covers frames 110-111 taking ~35 us longer than its size explains, on every box.)
8-frame calls over windows of the 128-frame input, with the output window following the input's or held fixed, and with the
input window held fixed while the output's moves."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch  # noqa: E402

W, H, N = 3840, 2160, 128
FB = W * H * 3
dev = torch.device("cuda", 0)
enc = jpeg.Encoder(0)
enc.set_quality(50)
d = torch.empty((N, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d.data_ptr(), FB, N, 1)
cap = 8 << 20
o = torch.zeros((N, cap), dtype=torch.uint8, device=dev)
b = torch.zeros(N, dtype=torch.int64, device=dev)
print("d_rgb %x  d_out %x" % (d.data_ptr(), o.data_ptr()))


def t(fin, fout, n=8, reps=30):
    for _ in range(3):
        enc.encode_scan_device(d.data_ptr() + fin * FB, W, H, n, o.data_ptr() + fout * cap, cap, b.data_ptr() + 8 * fout)
    enc.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode_scan_device(d.data_ptr() + fin * FB, W, H, n, o.data_ptr() + fout * cap, cap, b.data_ptr() + 8 * fout)
    enc.sync()
    return (time.perf_counter() - t0) / reps * 1e6


print("input and output window together:", " ".join("%d:%.0f" % (k, t(k, k)) for k in range(0, N, 8)))
print("input window moves, output at 0:  ", " ".join("%d:%.0f" % (k, t(k, 0)) for k in range(0, N, 8)))
print("input at 0, output window moves:  ", " ".join("%d:%.0f" % (k, t(0, k)) for k in range(0, N, 8)))
print("single frames 104..119 (in, out): ", " ".join("%d:%.0f" % (k, t(k, k, 1, 60)) for k in range(104, 120)))
