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


# ---- the reference's own interface, name for name (SURVEY §8b "Caller") ---------------------------------
REF_HDR = "/root/reference/src/utils.hpp"
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_api_driver")
STAGE_NAMES = ["readPPMImage", "writePPMImage", "removeRedChannel", "performCSC", "performCDS", "getPixelPtr", "getPixel",
               "getPixelR", "getPixelG", "getPixelB", "setPixelR", "setPixelG", "setPixelB", "copyUIntToDoubleImage",
               "copyToLargerImage", "getNearest8x8ImageSize", "addReversedPadding", "substractfromAll", "performDCT",
               "performQuantization", "everyMCUisnow2DArray", "performZigZag", "performRLE", "HuffmanEncoder",
               "JpegEncoderHost"]


def test_host_library_exports_the_reference_stage_functions():
    """libmi355host.a defines every function of the reference's path under the reference's name (the demangled
    signatures of the ones with distinctive arguments are checked literally, utils.hpp:77-137)."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    syms = subprocess.check_output(["nm", "-C", "--defined-only", os.path.join(PKG, "host", "libmi355host.a")], text=True)
    syms = syms.replace("[abi:cxx11]", "")
    for name in STAGE_NAMES:
        assert (" T %s(" % name) in syms, name
    for sig in ["performCSC(PPMimage*)", "performDCT(PPMimage_d*)", "substractfromAll(PPMimage_d*, double)",
                "performQuantization(PPMimage_d*, unsigned int const (*) [8], unsigned int const (*) [8])",
                "everyMCUisnow2DArray(PPMimage_d*, int (*) [64])", "performZigZag(int (*) [64], int (*) [64], int)",
                "performRLE(int (*) [64], std::vector<std::vector<int",
                "JpegEncoderHost(PPMimage, CPUTelemetry*)", "addReversedPadding(PPMimage*, unsigned long, unsigned long)"]:
        assert sig in syms, sig
    # HuffmanEncoder with the reference's three arguments (utils.hpp:137), next to the two-argument short form
    assert any("HuffmanEncoder(int (*) [64], std::vector<std::vector<int" in l and l.rstrip().endswith("int)")
               for l in syms.splitlines())
    # JpegEncoderHost sits in an archive member of its own: a program that defines it itself (the reference's main file) links
    members = subprocess.check_output(["nm", "-C", "--defined-only", "-A", os.path.join(PKG, "host", "libmi355host.a")], text=True)
    members = members.replace("[abi:cxx11]", "")
    owner = [l.split(":")[1] for l in members.splitlines() if " T JpegEncoderHost(" in l]
    assert owner == ["mi355_driver.o"]
    assert not any(" T performCSC(" in l and "mi355_driver.o" in l for l in members.splitlines())


@pytest.mark.skipif(not os.path.exists(REF_HDR), reason="the reference sources are not present on this machine")
def test_reference_header_links_against_the_host_library(tmp_path):
    """A translation unit that #includes the REFERENCE'S utils.hpp in place and calls its stage functions in
    JpegEncoderHost's order links against libmi355host.a + libmi355jpeg.so (link check, nothing runs)."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    exe = str(tmp_path / "ref_api_driver")
    subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-w", "-DMI355_USE_REFERENCE_HEADER", "-I/root/reference/src", "-o", exe,
                           os.path.join(ROOT, "tests", "ref_api_driver.cpp"), os.path.join(PKG, "host", "libmi355host.a"),
                           "-L" + PKG, "-lmi355jpeg", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    und = subprocess.check_output(["nm", "-C", "--undefined-only", exe], text=True)
    assert "performCSC" not in und and "HuffmanEncoder" not in und  # resolved from the archive, not left dangling
    # the same TU against this repo's own header (the form that is built where the reference is absent)
    subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-w", "-I" + os.path.join(PKG, "host"), "-o", exe + "2",
                           os.path.join(ROOT, "tests", "ref_api_driver.cpp"), os.path.join(PKG, "host", "libmi355host.a"),
                           "-L" + PKG, "-lmi355jpeg", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])


def _write_p6(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["fruit", "lcg_ragged", "lcg_aligned"])
def test_reference_stage_sequence_on_the_gpu(tmp_path, which):
    """The driver written against the reference's header (oracle/_ref/ref_api_driver: built in the build container
    with /root/reference/src/utils.hpp included in place, linked against libmi355host.a) runs the reference's stage
    functions one by one on the GPU in JpegEncoderHost's order; every intermediate it dumps equals the oracle's:
    performCSC, performCDS, padding, performDCT (fp64, bit for bit -- SURVEY §8 a2/a11), performQuantization,
    zig-zag rows, the RLE pair lists, the scan string; JpegEncoderHost fills all nine CPUTelemetry fields."""
    exe = REF_DRIVER
    if not os.path.exists(exe):  # no prebuilt driver travelled: build the form against this repo's header
        exe = str(tmp_path / "ref_api_driver")
        subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-w", "-I" + os.path.join(PKG, "host"), "-o", exe,
                               os.path.join(ROOT, "tests", "ref_api_driver.cpp"), os.path.join(PKG, "host", "libmi355host.a"),
                               "-L" + PKG, "-lmi355jpeg", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    rgb = {"fruit": lambda: ol.read_ppm(os.path.join(GOLD, "fruit.ppm")), "lcg_ragged": lambda: ol.lcg_frame(100, 37, 2),
           "lcg_aligned": lambda: ol.lcg_frame(256, 64, 5)}[which]()
    H, W, _ = rgb.shape
    src = str(tmp_path / "in.ppm")
    _write_p6(src, rgb)
    pre = str(tmp_path / "o")
    out = subprocess.run([exe, src, pre], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    o = ol.oracle_encode(rgb, None, None, True, ol.KEEP_ZIGZAG | ol.KEEP_U8_STAGES | ol.KEEP_DCT)
    H8, W8 = o.H8, o.W8
    assert np.array_equal(np.fromfile(pre + ".csc", np.uint8).reshape(H, W, 3), o.csc)
    assert np.array_equal(np.fromfile(pre + ".cds", np.uint8).reshape(H, W, 3), o.cds)
    assert np.array_equal(np.fromfile(pre + ".pad", np.uint8).reshape(H8, W8, 3), o.padded)
    dct = np.fromfile(pre + ".dct", np.float64).reshape(H8, W8, 3)
    assert np.array_equal(dct.view(np.uint64), o.dct.view(np.uint64)), "performDCT differs in some bit of some double"
    zz = np.fromfile(pre + ".zigzag", np.int32).reshape(-1, 64)
    assert np.array_equal(zz, o.zigzag)
    # performQuantization's image, gathered block-wise in natural order, is the zig-zag array un-zig-zagged
    quant = np.fromfile(pre + ".quant", np.float64).reshape(H8 // 8, 8, W8 // 8, 8, 3)
    nat = quant.transpose(4, 0, 2, 1, 3).reshape(-1, 64)            # [chan*N + block][v*8+u]
    assert np.array_equal(nat[:, ol.zigzag_order()].astype(np.int32), o.zigzag)
    # RLE pair lists: expand them and compare with the rows (RLEBlockAC, utils.cpp:572-609: always a final (0,0))
    flat = np.fromfile(pre + ".rle", np.int32)
    pos = 0
    for r in range(zz.shape[0]):
        n = int(flat[pos]); pairs = flat[pos + 1:pos + 1 + n].reshape(-1, 2); pos += 1 + n
        assert tuple(pairs[-1]) == (0, 0)
        k = 1
        row = np.zeros(64, np.int32)
        for run, val in pairs[:-1]:
            k += int(run)
            if val != 0:
                row[k] = val
            k += 1
        assert np.array_equal(row[1:], zz[r, 1:]), r
    assert pos == flat.size
    want = "".join(str(b) for b in np.unpackbits(o.bits)[:o.n_bits])
    assert open(pre + ".scan").read() == want
    tel = np.fromfile(pre + ".tel", np.float64)
    assert tel.shape == (9,) and (tel > 0).all()


@pytest.mark.gpu
def test_cli_prints_the_reference_speedup_table(tmp_path):
    """mi355-jpeg --stages --cpu-telemetry: the reference's '## Speedups: ##' block (OpenCLProject_JpegEncoder.cpp:622-629),
    CPU stage time / GPU stage time, with the CPU times of the reference path (here: the checker's run of it)."""
    subprocess.check_call(["make", "-s", "-C", PKG, "all", "host"])
    rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    r = ol.ref_encode(rgb) if ol.ref() is not None else ol.oracle_encode(rgb)
    us = r.stage_us  # CSC, CDS, copy, shift, DCT, quant, zigzag, RLE, huffman
    cpu = [us[0], us[1], us[3], us[4], us[5], us[2], us[6], us[7], us[8]]  # CPUTelemetry's declaration order
    tel = str(tmp_path / "cpu.txt")
    open(tel, "w").write(" ".join("%.3f" % max(v, 0.001) for v in cpu))
    out = subprocess.run([CLI, os.path.join(GOLD, "fruit.ppm"), str(tmp_path / "f.jpg"), "--stages", "--cpu-telemetry", tel],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    txt = out.stdout
    assert "Stage-by-stage scan equals the fused path's scan (307829 bits)" in txt
    block = txt[txt.index("## Speedups: ##"):]
    for label in ["Color conversion: ", "Chroma subsampling: ", "Level shifting: ", "DCT: ", "Quantization: ", "ZigZag: ", "RLE: "]:
        line = [l for l in block.splitlines() if l.startswith(label)]
        assert len(line) == 1 and float(line[0][len(label):]) > 0, label
    for label in ["CSC Time MI355X", "CDS Time MI355X", "Total Copy Time MI355X", "Level Shifting Time MI355X", "DCT Time MI355X",
                  "Quantization Time MI355X", "ZigZag Time MI355X", "RLE Time MI355X", "Huffman Time MI355X", "Total Time MI355X"]:
        assert label in txt, label
