import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
import torch
W, H = 3840, 2160
enc = jpeg.Encoder(0)
dev = torch.device("cuda", 0)
d_rgb = torch.empty((1, H, W, 3), dtype=torch.uint8, device=dev)
enc.synth_lcg_device(d_rgb.data_ptr(), W * H * 3, 1, 1)
d_out = torch.zeros((1, 8 << 20), dtype=torch.uint8, device=dev)
d_bits = torch.zeros(1, dtype=torch.int64, device=dev)
enc.screen_stats(reset=True)
enc.encode_scan_device(d_rgb.data_ptr(), W, H, 1, d_out.data_ptr(), 8 << 20, d_bits.data_ptr(), flags=jpeg.F_CDS)
enc.sync()
print("looks, exact:", enc.screen_stats(), "calls per frame:", (W//8)*(H//8)//64*3*16)
