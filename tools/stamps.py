#!/usr/bin/env python3
"""Development helper (GPU box): one 4K frame through a STAMPS=1 build (MI355_JPEG_LIB), phase cycles on stderr.
MI355_DIAG_FLAGS selects the mode (default strict; 6 = standard 4:2:0)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI355_JPEG_DUMP_STAMPS"] = "1"
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch
W, H = 3840, 2160
enc = jpeg.Encoder(0)
dev = torch.device("cuda", 0)
d_rgb = torch.empty((1, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, 1, 1)
d_out = torch.zeros((1, 8 << 20), dtype=torch.uint8, device=dev)
d_bits = torch.zeros(1, dtype=torch.int64, device=dev)
FLAGS = int(os.environ.get("MI355_DIAG_FLAGS", str(jpeg.F_DEFAULT)))
for i in range(3):
    enc.encode_scan_device(d_rgb.data_ptr(), W, H, 1, d_out.data_ptr(), 8 << 20, d_bits.data_ptr(), flags=FLAGS)
    enc.sync()
print("bits", int(d_bits[0]))
