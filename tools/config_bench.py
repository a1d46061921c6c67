#!/usr/bin/env python3
"""Development helper (GPU box): device-resident throughput of the BASELINE.json configs that are
not the bench line -- configs[2] (batch of 1080p frames, q75), configs[4] (16384x16384, q90, no
chroma averaging) -- plus standard mode on the 4K frame.  Prints one JSON line per case."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch  # noqa: E402


def run(name, W, H, n, quality, flags, cap_per_frame, reps):
    enc = jpeg.Encoder(0)
    enc.set_quality(quality)
    dev = torch.device("cuda", 0)
    d_rgb = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, n, 1)
    d_out = torch.zeros((n, cap_per_frame), dtype=torch.uint8, device=dev)
    d_bits = torch.zeros(n, dtype=torch.int64, device=dev)

    def go():
        enc.encode_scan_device(d_rgb.data_ptr(), W, H, n, d_out.data_ptr(), cap_per_frame, d_bits.data_ptr(), flags=flags)

    go()
    enc.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    enc.sync()
    dt = (time.perf_counter() - t0) / reps
    enc.set_profiling(1)  # stage times from one extra call (per-stage profiling runs a batch as one part)
    go()
    enc.sync()
    t = enc.last_timings()
    bits = d_bits.cpu().numpy()
    print(json.dumps({"case": name, "frames": n, "W": W, "H": H, "quality": quality, "flags": flags,
                      "ms_per_call": round(dt * 1e3, 4), "Gpixel_per_s": round(n * W * H / dt / 1e9, 2),
                      "bits_per_pixel": round(float(bits.sum()) / (n * W * H), 3),
                      "stage_ms": {k: round(v, 4) for k, v in t.items()}}), flush=True)
    enc.close()
    del d_rgb, d_out, d_bits
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c2", "c4", "std"]
    if "c1" in which:
        run("configs[1] 4K q50 strict", 3840, 2160, 1, 50, jpeg.F_CDS, 8 << 20, 50)
        run("configs[1] x16 frames per call", 3840, 2160, 16, 50, jpeg.F_CDS, 8 << 20, 10)
    if "c2" in which:
        run("configs[2] 256 x 1080p q75 strict (of 1024)", 1920, 1080, 256, 75, jpeg.F_CDS, 3 << 20, 4)
    if "c4" in which:
        run("configs[4] 16384^2 q90 no-CDS strict", 16384, 16384, 1, 90, 0, 600 << 20, 3)
    if "std" in which:
        run("standard mode 4K q50", 3840, 2160, 1, 50, jpeg.F_STANDARD, 8 << 20, 50)
        run("standard mode 4K q90", 3840, 2160, 1, 90, jpeg.F_STANDARD, 16 << 20, 20)
        run("standard 4:2:0 4K q50", 3840, 2160, 1, 50, jpeg.F_STANDARD | jpeg.F_420, 8 << 20, 50)
        run("standard 4:2:0 4K q50 x16 frames per call", 3840, 2160, 16, 50, jpeg.F_STANDARD | jpeg.F_420, 8 << 20, 10)
