#!/usr/bin/env python3
"""Development helper (GPU box): quick parity check against the oracle and a raw
timing of the pipeline stages.  Not part of the test-suite."""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

jpeg = importlib.import_module("jpeg-encoder-opencl_amd")


def check(enc, name, rgb, q=50, cds=True):
    ql, qc = ol.quant_tables(q)
    enc.set_quant(ql, qc)
    flags = jpeg.F_CDS if cds else 0
    o = ol.oracle_encode(rgb, ql, qc, cds, ol.KEEP_ZIGZAG | ol.KEEP_U8_STAGES | ol.KEEP_UNIT_BITS)
    smp = enc.probe_samples(rgb, flags)
    ok_s = np.array_equal(smp, o.padded)
    cf = enc.probe_coefficients(rgb, flags)
    ok_c = np.array_equal(cf.astype(np.int32), o.zigzag)
    ub = enc.probe_unit_bits(rgb, flags)
    ok_u = np.array_equal(ub, o.unit_bits)
    bits, nb = enc.encode_scan(rgb, flags)
    ok_b = nb[0] == o.n_bits and np.array_equal(bits[0], o.bits)
    print("%-28s samples %s coefs %s unit_bits %s scan %s (%d bits)" % (name, ok_s, ok_c, ok_u, ok_b, nb[0]),
          flush=True)
    if not ok_c:
        bad = np.argwhere(cf.astype(np.int32) != o.zigzag)
        print("   first coef mismatches:", bad[:5].tolist(), "of", len(bad))
    if not ok_u and ok_c:
        bad = np.argwhere(ub != o.unit_bits)
        print("   first unit_bits mismatches:", bad[:5].tolist(), ub[bad[:5, 0]], o.unit_bits[bad[:5, 0]])
    if not ok_b and ok_u:
        a = np.unpackbits(bits[0])[:nb[0]]
        b = np.unpackbits(o.bits)[:o.n_bits]
        n = min(len(a), len(b))
        d = np.argwhere(a[:n] != b[:n])
        print("   first bit mismatch at", d[:3].tolist(), "of", len(d))
    return ok_s and ok_c and ok_u and ok_b


def main():
    print("devices:", jpeg.device_count(), flush=True)
    enc = jpeg.Encoder(0)
    allok = True
    fruit = ol.read_ppm(os.path.join(ROOT, "tests", "golden", "fruit.ppm"))
    allok &= check(enc, "fruit q50 cds", fruit)
    allok &= check(enc, "fruit q90 nocds", fruit, 90, False)
    rng = np.random.default_rng(1)
    for (W, H) in [(8, 8), (64, 48), (512, 256), (100, 37), (37, 100), (1024, 64), (520, 8), (4096, 8)]:
        allok &= check(enc, "rand %dx%d" % (W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8), 50, True)
    allok &= check(enc, "lcg 640x480 q75", ol.lcg_frame(640, 480, 1), 75, True)
    print("ALL OK" if allok else "MISMATCH", flush=True)

    # timing, 4K LCG frame, device resident
    import torch
    W, H = 3840, 2160
    rgb = ol.lcg_frame(W, H, 1)
    ql, qc = ol.quant_tables(50)
    for mode in (2, 0):
        os.environ["MI355_JPEG_TRANSFORM_MODE"] = str(mode)
        e2 = jpeg.Encoder(0)
        e2.set_quant(ql, qc)
        d_rgb = torch.from_numpy(rgb).cuda()
        cap = 16 << 20
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        d_bits = torch.zeros(1, dtype=torch.int64, device="cuda")
        e2.set_profiling(True)
        for it in range(6):
            e2.encode_scan_device(d_rgb.data_ptr(), W, H, 1, d_out.data_ptr(), cap, d_bits.data_ptr())
            e2.sync()
            t = e2.last_timings()
            print("mode", mode, "iter", it, json.dumps({k: round(v, 4) for k, v in t.items()}), flush=True)
        nb = int(d_bits.item())
        packed = d_out[:(nb + 7) // 8].cpu().numpy()
        ascii_bits = (np.unpackbits(packed)[:nb] + ord("0")).astype(np.uint8)
        print("mode", mode, "4K bits", nb, "sha256", hashlib.sha256(ascii_bits.tobytes()).hexdigest(), flush=True)
        # back-to-back throughput without per-stage events
        e2.set_profiling(False)
        torch.cuda.synchronize()
        t0 = time.time()
        K = 20
        for it in range(K):
            e2.encode_scan_device(d_rgb.data_ptr(), W, H, 1, d_out.data_ptr(), cap, d_bits.data_ptr())
        e2.sync()
        dt = (time.time() - t0) / K
        print("mode", mode, "ms/frame %.4f  Mpx/s %.1f" % (dt * 1e3, W * H / dt / 1e6), flush=True)
        e2.close()


if __name__ == "__main__":
    main()
