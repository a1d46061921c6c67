#!/usr/bin/env python3
"""PCIe-inclusive throughput of the multi-GPU batch driver (host frames in, host scans out).
Secondary number for DESIGN.md; the headline `value` of bench.py is device-resident."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = bench.lcg_frames(8, 1, W, H)
frames = np.concatenate([base] * (n // 8))
ndev = jpeg.device_count()
for ids in ([0], [0, 0], list(range(ndev))):
    pool = jpeg.Pool(ids)
    pool.set_quality(50)
    best = 1e9
    for it in range(3):
        out, bits, secs = pool.encode(frames, cap=6 << 20)
        best = min(best, secs)
    assert bits[0] == 38227880
    pool.close()
    print(json.dumps({"workers": ids, "frames": n, "seconds": round(best, 4),
                      "Mpixel_per_s_pcie_inclusive": round(n * W * H / best / 1e6, 1),
                      "GB_per_s_h2d": round(n * W * H * 3 / best / 1e9, 2)}), flush=True)
