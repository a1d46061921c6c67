#!/usr/bin/env python3
"""Development helper (GPU box): per-stage device times of the 4K encode (HIP events);
with a `make STAMPS=1` build and MI355_JPEG_DUMP_STAMPS=1 also the in-kernel phase shares."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch
W, H = 3840, 2160
rgb = ol.lcg_frame(W, H, 1)
enc = jpeg.Encoder(0)
d_rgb = torch.from_numpy(rgb).cuda()
cap = 16 << 20
d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
d_bits = torch.zeros(1, dtype=torch.int64, device="cuda")
enc.set_profiling(1)
for flags in [1]:
    ts = []
    for it in range(8):
        enc.encode_scan_device(d_rgb.data_ptr(), W, H, 1, d_out.data_ptr(), cap, d_bits.data_ptr(), flags=flags)
        try:
            enc.sync()
        except Exception as e:
            pass
        ts.append(enc.last_timings()["transform_ms"])
    print("flags %#x transform_ms min %.4f med %.4f" % (flags, min(ts), sorted(ts)[len(ts) // 2]), flush=True)
