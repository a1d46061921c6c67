#!/usr/bin/env python3
"""Development helper (GPU box): block-encode kernel time per frame (16-frame launches, HIP events) under the diagnostic
bits of MI355_JPEG_STAGGER (256: no colour conversion, 512: no entropy walk) -- outputs are garbage, timing only."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch
W, H, n = 3840, 2160, 64
enc = jpeg.Encoder(0)
dev = torch.device("cuda", 0)
d_rgb = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, n, 1)
cap = 8 << 20
d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
d_bits = torch.zeros(n, dtype=torch.int64, device=dev)
def go():
    enc.encode_scan_device(d_rgb.data_ptr(), W, H, n, d_out.data_ptr(), cap, d_bits.data_ptr())
go()
try: enc.sync()
except Exception: pass
enc.set_profiling(2)
for _ in range(5): go()
try: enc.sync()
except Exception: pass
prof, calls = enc.profile_summary()
print(json.dumps({"stagger": os.environ.get("MI355_JPEG_STAGGER"), "kernel_us_per_frame": round(prof["transform_ms"] / calls / n * 1e3, 2)}))
