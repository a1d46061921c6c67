#!/usr/bin/env python3
"""Development helper (GPU box): ms per call of N-frame 4K batches in one context, fresh and after other shapes have
grown the workspace (why did bench.py's 100-frames-per-call case read 234 Gpixel/s where the sweep read 255?)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch  # noqa: E402

W, H = 3840, 2160
dev = torch.device("cuda", 0)
enc = jpeg.Encoder(0)
enc.set_quality(50)


def timed(n, w=W, h=H, cap=8 << 20, reps=6, warm=1, q=50, flags=jpeg.F_DEFAULT, tag="", prof=0, stream=None, one_sync=False):
    enc.set_quality(q)
    kw = {} if stream is None else {"stream": stream}
    d = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    enc.synth_lcg_device(d.data_ptr(), w * h * 3, n, 1)
    o = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
    b = torch.zeros(n, dtype=torch.int64, device=dev)
    for _ in range(warm):
        enc.encode_scan_device(d.data_ptr(), w, h, n, o.data_ptr(), cap, b.data_ptr(), flags=flags, **kw)
    enc.sync()
    torch.cuda.synchronize()
    ts = []
    enc.set_profiling(prof)
    if one_sync:  # bench.py's way: all calls queued, one sync
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.encode_scan_device(d.data_ptr(), w, h, n, o.data_ptr(), cap, b.data_ptr(), flags=flags, **kw)
        enc.sync(*([stream] if stream is not None else []))
        torch.cuda.synchronize()
        ts = [(time.perf_counter() - t0) * 1e3 / reps]
        reps = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        enc.encode_scan_device(d.data_ptr(), w, h, n, o.data_ptr(), cap, b.data_ptr(), flags=flags, **kw)
        enc.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    enc.set_profiling(0)
    print("%-34s n=%4d  %s  -> %.1f Gpixel/s (best %.1f)" % (tag, n, " ".join("%.3f" % t for t in ts), n * w * h / (sum(ts) / len(ts)) / 1e6,
                                                           n * w * h / min(ts) / 1e6), flush=True)
    del d, o, b
    torch.cuda.empty_cache()


st = torch.cuda.current_stream().cuda_stream
timed(100, warm=3, tag="plain")
timed(100, warm=1, prof=2, tag="profiling 2")
timed(100, warm=1, stream=st, tag="torch stream")
timed(100, warm=1, prof=2, stream=st, tag="profiling 2 + torch stream")
timed(100, warm=1, prof=2, stream=st, one_sync=True, tag="... + one sync for 6 calls")
timed(100, warm=1, one_sync=True, tag="one sync for 6 calls, plain")
timed(128, warm=1, prof=2, stream=st, one_sync=True, tag="128: prof 2 + stream + one sync")
timed(128, warm=1, one_sync=True, tag="128: one sync, plain")
timed(256, w=1920, h=1080, cap=3 << 20, q=75, reps=4, prof=2, stream=st, one_sync=True, tag="1080p x 256 q75, bench way")
timed(256, w=1920, h=1080, cap=3 << 20, q=75, reps=4, tag="1080p x 256 q75, plain")
