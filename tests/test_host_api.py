"""The utils.hpp-shaped C++ host API and the command line tool: built with
g++ against libmi355jpeg.so; on the GPU box driven like the reference's JpegEncoderHost
and compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLD, ROOT

PKG = os.path.join(ROOT, "jpeg-encoder-opencl_amd")
CLI = os.path.join(PKG, "host", "mi355-jpeg")


def build_host(tmp_path):
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    exe = str(tmp_path / "host_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host_driver.cpp"),
                           os.path.join(PKG, "host", "libmi355host.a"), "-L" + PKG, "-lmi355jpeg",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_host_api_builds_and_tables_match(tmp_path, jpeg):
    """No GPU needed: the shim compiles, the CLI parses arguments, and the four code
    tables it publishes as huffman.hpp-style strings equal the reference dump."""
    import json
    exe = build_host(tmp_path)
    assert subprocess.call([CLI, "--help"]) == 0
    prefix = str(tmp_path / "o")
    rc = subprocess.call([exe, os.path.join(GOLD, "fruit.ppm"), prefix])
    if jpeg.device_count() == 0:
        assert rc == 1  # fails loudly: no CPU path
    lines = open(prefix + ".tables").read().split("\n")
    with open(os.path.join(GOLD, "tables.json")) as f:
        t = json.load(f)["huffman"]
    want = [s or "NULL" for s in t["dc_luma"][0]] + [s or "NULL" for s in t["dc_chroma"][0]]
    for name in ("ac_luma", "ac_chroma"):
        for row in t[name]:
            want += [s or "NULL" for s in row]
    assert lines[:len(want)] == want
    assert lines[len(want)] == "99 18"


@pytest.mark.gpu
def test_host_driver_matches_oracle(tmp_path):
    exe = build_host(tmp_path)
    for (q, cds) in [(50, True), (90, False)]:
        prefix = str(tmp_path / ("o%d" % q))
        subprocess.check_call([exe, os.path.join(GOLD, "fruit.ppm"), prefix, str(q), "1" if cds else "0"])
        rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
        ql, qc = ol.quant_tables(q)
        o = ol.oracle_encode(rgb, ql, qc, cds)
        want = "".join(str(b) for b in np.unpackbits(o.bits)[:o.n_bits])
        assert open(prefix + ".bits").read() == want
        assert open(prefix + ".bits2").read() == want
        assert open(prefix + ".jpg", "rb").read() == ol.jfif_frame(o.bits, o.n_bits, rgb.shape[1], rgb.shape[0], ql, qc)


@pytest.mark.gpu
def test_cli_ppm_to_jpg(tmp_path):
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    out = str(tmp_path / "fruit.jpg")
    bits = str(tmp_path / "fruit.bits")
    subprocess.check_call([CLI, os.path.join(GOLD, "fruit.ppm"), out, "--bits", bits, "--repeat", "2"])
    rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    ql, qc = ol.quant_tables(50)
    o = ol.oracle_encode(rgb)
    assert open(out, "rb").read() == ol.jfif_frame(o.bits, o.n_bits, rgb.shape[1], rgb.shape[0], ql, qc)
    assert len(open(bits).read()) == 307829
    assert subprocess.call([CLI, "/nonexistent.ppm", out]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("sub", ["420", "444"])
def test_cli_standard_mode_writes_a_decodable_jpeg(tmp_path, sub):
    """--mode standard: the file equals the checker's framing of the checker's bits and decodes in
    Pillow (when importable) to the source picture."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    out = str(tmp_path / "fruit.jpg")
    subprocess.check_call([CLI, os.path.join(GOLD, "fruit.ppm"), out, "--mode", "standard", "--subsample", sub,
                           "-q", "90"])
    rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    ql, qc = ol.quant_tables(90)
    ss = 1 if sub == "420" else 0
    o = ol.oracle_std_encode(rgb, ql, qc, subsample=ss)
    data = open(out, "rb").read()
    assert data == ol.jfif_frame(o.bits, o.n_bits, rgb.shape[1], rgb.shape[0], ql, qc, ss)
    try:
        from PIL import Image
    except ImportError:
        return
    dec = np.asarray(Image.open(out).convert("RGB")).astype(np.float64)
    assert dec.shape == rgb.shape
    mse = ((dec - rgb) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / mse) > (19.5 if ss else 29.0)  # fruit.ppm is close to noise
    assert subprocess.call([CLI, os.path.join(GOLD, "fruit.ppm"), out, "--mode", "standard", "--subsample", "ref420"]) == 2


def write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "standard"])
def test_cli_batch_directory_mode(tmp_path, mode):
    """--batch: a directory of PPMs of two sizes (+ one broken file) through the multi-GPU pool; every
    output equals the checker's framing of the checker's bits, i.e. the one-file form's bytes."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = {"a0": ol.lcg_frame(96, 64, 1), "a1": ol.lcg_frame(96, 64, 2), "a2": ol.lcg_frame(96, 64, 3),
              "b0": ol.lcg_frame(50, 33, 4), "b1": ol.lcg_frame(50, 33, 5)}
    for k, v in frames.items():
        write_ppm(src / (k + ".ppm"), v)
    (src / "broken.ppm").write_bytes(b"P5\n1 1\n255\n\0")
    args = [CLI, "--batch", str(src), str(dst), "-q", "80"]
    if mode == "standard":
        args += ["--mode", "standard", "--subsample", "420"]
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    ql, qc = ol.quant_tables(80)
    for k, rgb in frames.items():
        if mode == "standard":
            o = ol.oracle_std_encode(rgb, ql, qc, subsample=1)
            want = ol.jfif_frame(o.bits, o.n_bits, rgb.shape[1], rgb.shape[0], ql, qc, 1)
        else:
            o = ol.oracle_encode(rgb, ql, qc, True)
            want = ol.jfif_frame(o.bits, o.n_bits, rgb.shape[1], rgb.shape[0], ql, qc)
        assert (dst / (k + ".jpg")).read_bytes() == want, k
    assert not (dst / "broken.jpg").exists()


def test_ppm_reader_accepts_wellformed_headers(tmp_path):
    """No GPU needed: the host reader takes the reference's strict 3-line form and every other
    well-formed P6/255 header (comments, arbitrary whitespace), and rejects bad input with -1
    (exit code 1 of the CLI) instead of reading garbage."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    raster = bytes(range(48)) * 4  # 8x8x3
    good = [b"P6\n8 8\n255\n", b"P6\n# a comment\n8 8\n255\n", b"P6 8 8 255\n", b"P6\n8\n8\n255\n",
            b"P6\t8  8 # trailing comment\n255\n", b"P6\r\n8 8\r\n255\n"]
    bad = [b"P5\n8 8\n255\n", b"P6\n8 8\n65535\n", b"P6\n8 x\n255\n", b"P6\n0 8\n255\n", b"P6\n8 8\n"]
    for i, hdr in enumerate(good + bad):
        path = tmp_path / ("t%d.ppm" % i)
        path.write_bytes(hdr + raster)
        out = subprocess.run([CLI, str(path), str(tmp_path / "o.jpg")], capture_output=True, text=True)
        if i < len(good):
            assert "Image %s: 8 x 8" % path in out.stdout, (hdr, out.stdout)
        else:
            assert out.returncode == 1 and "Image" not in out.stdout, (hdr, out.stdout)
    short = tmp_path / "short.ppm"
    short.write_bytes(b"P6\n8 8\n255\n" + raster[:100])
    out = subprocess.run([CLI, str(short), str(tmp_path / "o.jpg")], capture_output=True, text=True)
    assert out.returncode == 1 and "Error reading the file" in out.stdout
