#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/libjpegref.so, built
by oracle/Makefile from /root/reference/src/utils.cpp in place).

Runs only in the build container (the reference does not travel to the GPU box).
Fixtures are data only: inputs and expected outputs.

    python tools/make_golden.py            # small + medium cases (~1 min)
    python tools/make_golden.py --big      # also 4K / 1080p / 2048^2 hashes (~1 min more)
"""
import argparse
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FNV_OFF, FNV_PRIME = 1469598103934665603, 1099511628211


def fnv1a64(buf):
    """FNV-1a 64 over raw bytes (SURVEY Appendix B's stage hash).  Vectorised by
    nothing: byte-serial, so only used on small buffers."""
    h = FNV_OFF
    for b in bytes(buf):
        h = ((h ^ b) * FNV_PRIME) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def sha_bits(r):
    return hashlib.sha256(r.bit_chars.tobytes()).hexdigest()


def case_record(name, rgb, quality, cds_on, stages=False):
    ql, qc = ol.quant_tables(quality)
    keep = ol.KEEP_ZIGZAG | (ol.KEEP_U8_STAGES | ol.KEEP_DCT if stages else 0)
    r = ol.ref_encode(rgb, ql, qc, cds_on, keep)
    rec = {
        "name": name, "W": int(rgb.shape[1]), "H": int(rgb.shape[0]), "quality": quality,
        "cds_on": bool(cds_on), "n_bits": int(r.n_bits), "sha256_ascii_bits": sha_bits(r),
        "sha256_packed_bits": hashlib.sha256(r.bits.tobytes()).hexdigest(),
        "sha256_zigzag_i32": hashlib.sha256(r.zigzag.astype("<i4").tobytes()).hexdigest(),
        "max_abs_q": int(np.abs(r.zigzag).max()),
        "units_c63_nonzero": int((r.zigzag[:, 63] != 0).sum()),
    }
    if stages:
        rec["sha256_csc"] = hashlib.sha256(r.csc.tobytes()).hexdigest()
        rec["sha256_cds"] = hashlib.sha256(r.cds.tobytes()).hexdigest()
        rec["sha256_padded"] = hashlib.sha256(r.padded.tobytes()).hexdigest()
        rec["sha256_dct_f64"] = hashlib.sha256(r.dct.astype("<f8").tobytes()).hexdigest()
    return rec, r


def tiled_fruit(W=3840, H=2160):
    """src[(y mod 254) * 253 + (x mod 253)] of data/fruit.ppm."""
    fruit = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    yy, xx = np.arange(H) % fruit.shape[0], np.arange(W) % fruit.shape[1]
    return np.ascontiguousarray(fruit[yy][:, xx])


def add_tiled_fruit():
    assert ol.ref() is not None, "build oracle/_ref first (make -C oracle)"
    path = os.path.join(GOLD, "cases.json")
    cases = [c for c in json.load(open(path)) if not c["name"].startswith(("fruit_tiled_", "std420_fruit_tiled_"))]
    rgb = tiled_fruit()
    rec = case_record("fruit_tiled_3840x2160_q50_cds", rgb, 50, True)[0]
    rec["big"] = True
    cases.append(rec)
    print(rec["name"], rec["n_bits"], rec["sha256_ascii_bits"])
    ql, qc = ol.quant_tables(50)
    o = ol.oracle_std_encode(rgb, ql, qc, 0, 1)
    rec2 = {"name": "std420_fruit_tiled_3840x2160_q50", "W": 3840, "H": 2160, "quality": 50, "flags": "MI355_F_STANDARD|MI355_F_420",
            "n_bits": int(o.n_bits), "sha256_packed_bits": hashlib.sha256(o.bits[:(o.n_bits + 7) // 8].tobytes()).hexdigest(),
            "source": "oracle/jpeg_oracle.c orc_std_encode (the checker of standard mode: parity unpinned by the reference)",
            "big": True, "standard": True}
    cases.append(rec2)
    print(rec2["name"], rec2["n_bits"], rec2["sha256_packed_bits"])
    with open(path, "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote", len(cases), "cases")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--tiled-fruit", action="store_true",
                    help="only ADD the natural-statistics cases of SURVEY §8(d) to the existing cases.json: fruit.ppm tiled to "
                         "3840x2160, strict q50 from the reference build (SURVEY Appendix B: 39 146 255 bits, a23fc925...) and "
                         "standard 4:2:0 q50 from the checker (that mode is not a behaviour of the reference)")
    args = ap.parse_args()
    if args.tiled_fruit:
        return add_tiled_fruit()
    assert ol.ref() is not None, "build oracle/_ref first (make -C oracle)"
    os.makedirs(GOLD, exist_ok=True)
    L = ol.ref()

    # --- constant tables as the reference build produces them
    import ctypes as C
    cos = [[float.hex(L.ref_cos(a, k)) for k in range(8)] for a in range(8)]
    scale = {"00": float.hex(L.ref_scale(0, 0)), "0x": float.hex(L.ref_scale(0, 1)),
             "x0": float.hex(L.ref_scale(1, 0)), "xx": float.hex(L.ref_scale(1, 1))}
    buf = C.create_string_buffer(64)
    huff = {}
    for t, tname in enumerate(["dc_luma", "dc_chroma", "ac_luma", "ac_chroma"]):
        rows = []
        for run in range(16 if t >= 2 else 1):
            row = []
            for size in range(11 if t >= 2 else 12):
                n = L.ref_huff_code(t, run, size, buf)
                s = buf.value.decode()
                row.append(None if (n < 0 or s == "NULL") else s)
            rows.append(row)
        huff[tname] = rows
    ql = np.zeros(64, np.uint32)
    qc = np.zeros(64, np.uint32)
    L.ref_quant_tables(ql.ctypes.data, qc.ctypes.data)
    # exhaustive colour conversion: pixel index i = r<<16 | g<<8 | b through performCSC
    i = np.arange(1 << 24, dtype=np.uint32)
    allrgb = np.stack([(i >> 16) & 255, (i >> 8) & 255, i & 255], -1).astype(np.uint8)
    L.ref_csc_only(allrgb.ctypes.data, 1 << 24)
    csc_sha = hashlib.sha256(allrgb.tobytes()).hexdigest()
    with open(os.path.join(GOLD, "tables.json"), "w") as f:
        json.dump({"cos": cos, "scale": scale, "huffman": huff, "quant_lum": ql.tolist(),
                   "quant_chrom": qc.tolist(), "csc_exhaustive_sha256": csc_sha}, f, indent=1)

    # --- fruit.ppm: the reference's own sample input (data file, not source)
    src = "/root/reference/data/fruit.ppm"
    shutil.copyfile(src, os.path.join(GOLD, "fruit.ppm"))
    fruit = ol.read_ppm(src)
    cases = []
    rec, r = case_record("fruit_q50_cds", fruit, 50, True, stages=True)
    rec["fnv_csc"] = fnv1a64(r.csc.tobytes())
    rec["fnv_zigzag"] = fnv1a64(r.zigzag.astype("<i4").tobytes())
    rec["first64"] = "".join(chr(c) for c in r.bit_chars[:64])
    cases.append(rec)
    r.bits.tofile(os.path.join(GOLD, "fruit_q50_cds.scanbits"))
    r.zigzag.astype("<i2").tofile(os.path.join(GOLD, "fruit_q50_cds.zigzag_i16"))
    for q, cds in [(50, False), (75, True), (90, False)]:
        cases.append(case_record("fruit_q%d_%s" % (q, "cds" if cds else "nocds"), fruit, q, cds)[0])

    # --- pinned LCG frames, small enough for the CPU suite
    for (W, H, seed, q, cds) in [(64, 48, 1, 50, True), (253, 254, 3, 50, True), (100, 37, 2, 75, True),
                                 (37, 100, 5, 90, False), (256, 256, 1, 50, True), (8, 8, 7, 50, True),
                                 (16, 8, 9, 100, False), (512, 256, 4, 10, True)]:
        rgb = ol.lcg_frame(W, H, seed)
        cases.append(case_record("lcg_%dx%d_s%d_q%d_%s" % (W, H, seed, q, "cds" if cds else "nocds"),
                                 rgb, q, cds, stages=True)[0])
        cases[-1]["seed"] = seed
    # structured inputs: flat, extremes (exercise large DC categories), gradients
    H, W = 40, 72
    yy, xx = np.mgrid[0:H, 0:W]
    structured = {
        "flat0": np.zeros((H, W, 3), np.uint8), "flat255": np.full((H, W, 3), 255, np.uint8),
        "checker": (((yy // 8 + xx // 8) % 2) * 255).astype(np.uint8)[..., None].repeat(3, 2),
        "pixchecker": (((yy + xx) % 2) * 255).astype(np.uint8)[..., None].repeat(3, 2),
        "gradient": np.stack([(xx * 255 // (W - 1)), (yy * 255 // (H - 1)), ((xx + yy) % 256)], -1).astype(np.uint8),
    }
    np.savez_compressed(os.path.join(GOLD, "structured_inputs.npz"), **structured)
    for name, rgb in structured.items():
        for q in (50, 100):
            cases.append(case_record("%s_q%d" % (name, q), rgb, q, True)[0])

    if args.big:
        # bench.py re-hashes the first frames of every rank's batch (rank r owns seeds 1 + 128 r ...) and the
        # configs[2] test samples its 1024 frames: more pinned frames of those two shapes
        extra = [(3840, 2160, s, 50, True) for s in (2, 3, 4)] + \
                [(3840, 2160, 1 + 128 * r, 50, True) for r in range(1, 8)] + \
                [(1920, 1080, s, 75, True) for s in (2, 513, 1024)]
        for (W, H, seed, q, cds) in [(3840, 2160, 1, 50, True), (1920, 1080, 1, 75, True),
                                     (2048, 2048, 1, 90, False)] + extra:
            rgb = ol.lcg_frame(W, H, seed)
            rec = case_record("lcg_%dx%d_s%d_q%d_%s" % (W, H, seed, q, "cds" if cds else "nocds"),
                              rgb, q, cds)[0]
            rec["seed"] = seed
            rec["big"] = True
            cases.append(rec)
            print(rec["name"], rec["n_bits"], rec["sha256_ascii_bits"])
    else:
        # keep previously generated big records
        old = os.path.join(GOLD, "cases.json")
        if os.path.exists(old):
            cases += [c for c in json.load(open(old)) if c.get("big")]

    with open(os.path.join(GOLD, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
