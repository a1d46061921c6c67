"""profiles/traffic.json carries what bench.py reads for roofline.traffic (and what make_traffic.py writes)."""
import json
import os

from conftest import ROOT


def test_traffic_json_format():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("kernel", "kernel_sources_sha", "source", "k_screen_encode_hbm_bytes_per_frame",
                "k_screen_encode_valu_insts_per_frame", "breakdown_bytes_per_frame"):
        assert key in t, key
    assert t["kernel"] == "k_screen_encode" and len(t["kernel_sources_sha"]) == 16
    rgb = 3840 * 2160 * 3
    rd, wr = t["breakdown_bytes_per_frame"]["read_corrected"], t["breakdown_bytes_per_frame"]["written"]
    # the frame's RGB bytes are fetched at least once and (nearly) only once; nothing is written twice
    assert rgb <= rd < 1.1 * rgb and 0 < wr < rgb
    assert t["k_screen_encode_hbm_bytes_per_frame"] == rd + wr
    assert os.path.exists(os.path.join(ROOT, t["source"]))


def test_bench_hashes_the_kernel_sources():
    import bench
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and int(sha, 16) >= 0
