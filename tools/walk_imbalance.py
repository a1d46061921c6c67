#!/usr/bin/env python3
"""Development helper (GPU box): how much of the entropy walk is spent on the wave's LARGEST unit?
The walk of k_screen_encode runs ceil(max symbols / 2) trips for the 64 units of a (tile, channel) pass.  From the quantised
coefficients of one 4K LCG frame (mi355_jpeg_probe_coefficients, strict q50, chroma averaging on) this prints, per channel:
the mean and the wave maximum of the symbol counts (non-zero AC coefficients + ZRLs), the trips the kernel runs, the trips
a perfectly balanced walk would run, and what a split at a symbol budget T would run (main loop T/2 trips + the leftover
symbols of the long units spread over all 64 lanes)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")

W, H = 3840, 2160
n = W * H * 3
import torch
enc = jpeg.Encoder(0)
d = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda:0")
enc.synth_lcg_device(d.data_ptr(), n, 1, 1)
rgb = d[0].cpu().numpy()
cf = enc.probe_coefficients(rgb, jpeg.F_CDS).astype(np.int32)  # [3N, 64], rows chan*N + block, zig-zag order
N = cf.shape[0] // 3
ac = cf[:, 1:] != 0
pos = np.where(ac, np.arange(1, 64)[None, :], 0)
cnt = ac.sum(1)
# ZRLs: for every non-zero, floor(run / 16) with run = zeros since the previous non-zero
zrl = np.zeros(cf.shape[0], np.int64)
prev = np.zeros(cf.shape[0], np.int64)
for k in range(1, 64):
    nz = cf[:, k] != 0
    run = k - prev - 1
    zrl += np.where(nz, run // 16, 0)
    prev = np.where(nz, k, prev)
sym = cnt + zrl
for c, name in enumerate(("Y", "Cb", "Cr")):
    u = sym[c * N:(c + 1) * N]
    pad = (-len(u)) % 64
    w = np.concatenate([u, np.zeros(pad, u.dtype)]).reshape(-1, 64)  # one row = one wave pass (tile = 64 consecutive blocks)
    mx, mean = w.max(1), w.mean(1)
    trips = np.ceil(mx / 2).sum()
    ideal = np.ceil(w.sum(1) / 64 / 2).sum()
    line = "%-2s symbols/unit mean %.2f, wave max mean %.2f (mean/max %.3f); trips now %d, balanced %d (%.1f %%)" % (
        name, u.mean(), mx.mean(), u.mean() / mx.mean(), trips, ideal, 100 * ideal / trips)
    best = None
    for T in range(8, 64, 2):
        left = np.maximum(w - T, 0).sum(1)
        t = (np.minimum(mx, T) / 2).round().astype(np.int64) * 0 + np.ceil(np.minimum(mx, T) / 2) + 2 * np.ceil(left / 64)  # a leftover batch of 64 symbols costs ~2 trips' worth
        tot = t.sum()
        if best is None or tot < best[1]:
            best = (T, tot)
    print(line + "; best split T=%d: %d (%.1f %%)" % (best[0], best[1], 100 * best[1] / trips))
