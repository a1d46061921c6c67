#!/usr/bin/env python3
"""Development helper (GPU box): timing of the single-launch pipeline under diagnostic knobs.
usage: fused_diag.py [frames] ; env MI355_JPEG_PIPELINE are read by the library."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch
W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc = jpeg.Encoder(0)
dev = torch.device("cuda", 0)
d_rgb = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, n, 1)
cap = int(os.environ.get("MI355_DIAG_CAP_MB", "8")) << 20
d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
d_bits = torch.zeros(n, dtype=torch.int64, device=dev)
FLAGS = int(os.environ.get("MI355_DIAG_FLAGS", str(jpeg.F_DEFAULT)))  # e.g. 6 = MI355_F_STANDARD | MI355_F_420
enc.set_quality(int(os.environ.get("MI355_DIAG_QUALITY", "50")))
def go(k):
    enc.encode_scan_device(d_rgb.data_ptr(), W, H, k, d_out.data_ptr(), cap, d_bits.data_ptr(), flags=FLAGS)
res = {}
for k in (1, n):
    go(k)
    try: enc.sync()
    except Exception as e: res["err%d" % k] = str(e)
    torch.cuda.synchronize()
    reps = 20 if k == 1 else 5
    t0 = time.perf_counter()
    for _ in range(reps): go(k)
    try: enc.sync()
    except Exception as e: pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res["frames%d_us_per_frame" % k] = round(dt / k * 1e6, 2)
print(json.dumps({"pipeline": os.environ.get("MI355_JPEG_PIPELINE"), "flags": FLAGS, **res}))
