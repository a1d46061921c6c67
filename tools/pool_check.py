#!/usr/bin/env python3
"""Development helper (GPU box): repeated pool / batched encodes of 4K LCG frames, every frame's bit
count and SHA-256 checked against the first occurrence of its seed (and seed 1 against the golden)."""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
frames = np.stack([ol.lcg_frame(3840, 2160, 1 + (f % 3)) for f in range(12)])
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    pool = jpeg.Pool([0])
    out, bits, secs = pool.encode(frames, cap=6 << 20)
    pool.close()
    ref = {}
    for f in range(12):
        h = hashlib.sha256(out[f, :(bits[f] + 7) // 8].tobytes()).hexdigest()
        key = f % 3
        if key not in ref:
            ref[key] = (bits[f], h)
        if (bits[f], h) != ref[key] or (key == 0 and bits[f] != 38227880):
            bad += 1
            print("rep", rep, "frame", f, "bits", bits[f], "expected", ref[key][0])
print("batch parts limit", os.environ.get("MI355_JPEG_BATCH_PARTS", "default"), "- bad frames:", bad)
