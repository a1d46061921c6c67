"""Standard mode: a decodable baseline JPEG.  NOT a behaviour of the
reference -- parity unpinned by it.  The mode is defined by its arithmetic (the true DCT-II as a fixed-point
map: tests/golden/std_dct_q39.i64, of which the top three base-256 digits = 23 fractional bits are used,
evaluated exactly in integers; the quotient by Q in single precision, nearest integer with ties to even;
colour conversion in 15-bit fixed point, libjpeg's form; 4:2:0 chroma as the box filter of that linear form, rounded
once), so the checker in oracle/ and the HIP path must agree bit for bit;
what ties it to the outside world is that libjpeg (PIL) decodes the files to the picture an independent
JPEG encoder produces, and that the quantised values are within one of the exact quotient's rounding."""
import io
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLD

PIL = pytest.importorskip("PIL.Image")

KEEP = ol.KEEP_ZIGZAG | ol.KEEP_UNIT_BITS


def psnr(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / mse)


def smooth_frame(W, H, seed=0):
    """A picture with natural-ish statistics (smooth gradients + a little texture)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 120 * np.sin(x / 37.0 + seed) * np.cos(y / 23.0),
                    127 + 100 * np.cos(x / 51.0) * np.cos(y / 17.0 + 1),
                    (x * 255 // max(W - 1, 1) + y * 255 // max(H - 1, 1)) / 2], -1)
    img += rng.normal(0, 3, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def pil_decode(data):
    return np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))


def pil_encode(rgb, ql, qc, subsample=0):
    buf = io.BytesIO()  # Pillow takes the tables in natural order and writes them zig-zagged
    PIL.fromarray(rgb).save(buf, "JPEG", subsampling=2 if subsample else 0,
                            qtables=[[int(v) for v in ql.reshape(64)], [int(v) for v in qc.reshape(64)]])
    return buf.getvalue()


def test_dct_table_is_the_true_dct():
    """The defining table equals 1/4 a(u) a(v) cos cos * 2^39 (zig-zag row order) to the unit."""
    T = ol.std_dct_table()
    zz = ol.zigzag_order()
    a = np.array([np.sqrt(0.5)] + [1.0] * 7)
    k = np.arange(8)
    C = np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16)  # [x][u]
    for R in range(64):
        v, u = divmod(int(zz[R]), 8)
        want = (a[u] * a[v] / 4) * np.outer(C[:, v], C[:, u]).reshape(64) * 2.0 ** 39
        assert np.abs(T[R] - want).max() <= 1.0
    assert (T[0] == 1 << 36).all()


@pytest.mark.parametrize("quality", [50, 75, 90])
def test_oracle_standard_files_decode_like_an_independent_encoder(quality):
    rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    ql, qc = ol.quant_tables(quality)
    o = ol.oracle_std_encode(rgb, ql, qc)
    H, W, _ = rgb.shape
    f = ol.jfif_frame(o.bits, o.n_bits, W, H, ql, qc)
    ours = pil_decode(f)
    theirs = pil_decode(pil_encode(rgb, ql, qc))
    assert ours.shape == rgb.shape
    # same tables, same sampling: both decode to nearly the same picture (libjpeg's integer DCT and
    # colour conversion round differently; fruit.ppm is close to noise, so that is the worst case) and
    # both are equally far from the source
    assert psnr(ours, theirs) > 30.0
    assert abs(psnr(ours, rgb) - psnr(theirs, rgb)) < 0.1


@pytest.mark.parametrize("kind,quality", [("fruit", 50), ("fruit", 90), ("smooth", 50), ("smooth", 92)])
def test_oracle_420_files_decode_like_an_independent_encoder(kind, quality):
    """4:2:0: 16x16 MCUs, sampling factors 2x2 / 1x1 / 1x1 -- decoded by libjpeg, compared with
    Pillow's own 4:2:0 encoder on the same tables (odd sizes: mirror padding to multiples of 16)."""
    rgb = ol.read_ppm(os.path.join(GOLD, "fruit.ppm")) if kind == "fruit" else smooth_frame(333, 201, 4)
    ql, qc = ol.quant_tables(quality)
    o = ol.oracle_std_encode(rgb, ql, qc, KEEP, subsample=1)
    H, W, _ = rgb.shape
    M = ((W + 15) // 16) * ((H + 15) // 16)
    assert o.n_blocks == M and o.zigzag.shape == (6 * M, 64) and o.n_bits == int(o.unit_bits.sum())
    f = ol.jfif_frame(o.bits, o.n_bits, W, H, ql, qc, 1)
    sof = f.index(b"\xff\xc0")
    assert f[sof + 10:sof + 19] == bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    ours = pil_decode(f)
    theirs = pil_decode(pil_encode(rgb, ql, qc, 1))
    assert ours.shape == rgb.shape
    assert psnr(ours, theirs) > (35.0 if kind == "fruit" else 44.0)
    assert abs(psnr(ours, rgb) - psnr(theirs, rgb)) < 0.1


def test_standard_colour_conversion_definition():
    """The fixed-point colour conversion that defines standard mode: the C checker equals the numpy restatement on all
    2^24 inputs, every value lies in 0..255 without clamping, greys stay grey, and it is within 1 of the correctly
    rounded real-valued conversion everywhere (equal to it for >= 99 % of the inputs)."""
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([v >> 16, (v >> 8) & 255, v & 255], -1).astype(np.uint8)
    want = ol.std_csc(rgb)
    assert want.min() >= 0 and want.max() <= 255
    assert np.array_equal(ol.oracle_std_csc(rgb), want.astype(np.uint8))
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    assert np.array_equal(ol.std_csc(grey), np.stack([grey[:, 0], np.full(256, 128), np.full(256, 128)], -1))
    r, g, b = (rgb[..., i].astype(np.float64) for i in range(3))
    real = np.stack([0.299 * r + 0.587 * g + 0.114 * b, 128 - 0.168736 * r - 0.331264 * g + 0.5 * b,
                     128 + 0.5 * r - 0.418688 * g - 0.081312 * b], -1)
    diff = np.abs(want - np.minimum(np.floor(real + 0.5), 255))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.99


def test_standard_420_chroma_definition():
    """4:2:0 chroma = the linear form box-filtered over the quad and rounded once: in 0..255 without clamping, and within
    1 of the rounded mean of the four per-pixel chroma values."""
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    rgb[:8, :8] = 0, 0, 255      # extremes of Cb ...
    rgb[8:16, :8] = 255, 255, 0
    rgb[16:24, :8] = 255, 0, 0   # ... and Cr
    rgb[24:32, :8] = 0, 255, 255
    got = ol.std_chroma420(rgb)
    assert got.min() == 0 and got.max() == 255
    pp = ol.std_csc(rgb)[..., 1:]
    mean = (pp[0::2, 0::2] + pp[0::2, 1::2] + pp[1::2, 0::2] + pp[1::2, 1::2] + 2) >> 2
    assert np.abs(got - mean).max() <= 1


def test_chroma_numerators_divide_by_32_for_all_inputs():
    """The HIP colour conversion forms the chroma numerators divided by 32 (constants that fit the
    16-bit lanes of v_dot2): floor(x / 1e6) == floor((x / 32) / 31250) and the rounded variants, for
    all 2^24 inputs -- the algebra behind csc_packed() in jpeg_screen_kernels.hip."""
    v = np.arange(256, dtype=np.int64)
    r, g, b = v[:, None, None], v[None, :, None], v[None, None, :]
    for x, y in ((128000000 + 500000 * b - 168736 * r - 331264 * g, 4000000 + 15625 * b - 5273 * r - 10352 * g),
                 (128000000 + 500000 * r - 418688 * g - 81312 * b, 4000000 + 15625 * r - 13084 * g - 2541 * b)):
        assert (x == 32 * y).all() and (y > 0).all() and (y < 2 ** 31).all()
        assert (x // 1000000 == y // 31250).all()
        assert ((x + 500000) // 1000000 == (y + 15625) // 31250).all()


def test_division_constants():
    """The 24-bit multiply-high divisions of the HIP colour conversion (div1000 / div31250 in
    jpeg_screen_kernels.hip) are exact over the ranges they are used on."""
    s = np.arange(0, 256000, dtype=np.uint64)
    assert (((s * 8589935) >> 33) == s // 1000).all() and 8589935 < 2 ** 24
    s = np.arange(0, 8100000, dtype=np.uint64)
    assert (((s * 8796094) >> 38) == s // 31250).all() and 8796094 < 2 ** 24


@pytest.mark.parametrize("ss", [0, 1])
def test_oracle_restart_files_decode_to_the_same_pixels(ss):
    """DRI/RSTm: every 64-MCU interval starts from zero predictors and ends on a byte boundary; a
    decoder must produce exactly the pixels of the file without restart markers."""
    rgb = smooth_frame(333, 201, 4)
    ql, qc = ol.quant_tables(85)
    f = ol.oracle_std_jfif_restart(rgb, ql, qc, ss, 64)
    o = ol.oracle_std_encode(rgb, ql, qc, subsample=ss)
    plain = ol.jfif_frame(o.bits, o.n_bits, 333, 201, ql, qc, ss)
    assert b"\xff\xdd\x00\x04\x00\x40" in f and b"\xff\xdd" not in plain
    n_mcu = ((333 + 15) // 16) * ((201 + 15) // 16) if ss else ((333 + 7) // 8) * ((201 + 7) // 8)
    sos = f.index(b"\xff\xda")
    markers = [f[i + 1] for i in range(sos + 14, len(f) - 2) if f[i] == 0xFF and 0xD0 <= f[i + 1] <= 0xD7]
    assert markers == [0xD0 + (i & 7) for i in range((n_mcu + 63) // 64 - 1)]
    assert np.array_equal(pil_decode(f), pil_decode(plain))


def test_oracle_standard_coefficients_against_float_dct():
    """The mode's CONTRACT (include/mi355_jpeg.h, MI355_F_STANDARD), against independent arithmetic: scipy's orthonormal
    fp64 DCT-II of the same samples, divided by Q and rounded half away.  The defined value (23-bit fixed-point DCT, fp32
    quotient) equals that integer except where the quotient lies within 2e-3 of a rounding tie, and there it is off by at
    most one.  Inputs reach the extremes: two-level blocks at q100 give quotients up to ~1000, where the fp32 quotient's
    absolute error is largest."""
    import scipy.fft
    rgb = smooth_frame(96, 64, 9)
    rgb[:32] = ol.lcg_frame(96, 32, 5)               # noise blocks
    rng = np.random.default_rng(11)
    rgb[32:48] = np.where(rng.random((16, 96, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, 2)   # two-level greys: |coef| up to ~1000
    rgb[48:56, :48] = 255
    rgb[48:56, 48:] = 0
    ycc = ol.std_csc(rgb)
    zz = ol.zigzag_order()
    seen_large, seen_near = 0, 0
    for q in (50, 90, 100):
        ql, qc = ol.quant_tables(q)
        o = ol.oracle_std_encode(rgb, ql, qc, KEEP)
        N = o.n_blocks
        for c in range(3):
            blocks = (ycc[..., c] - 128.0).reshape(8, 8, 12, 8).transpose(0, 2, 1, 3).reshape(N, 8, 8)
            coef = scipy.fft.dctn(blocks, type=2, norm="ortho", axes=(1, 2)).reshape(N, 64)[:, zz]
            z = coef / (ql if c == 0 else qc).reshape(64)[zz].astype(np.float64)
            want = np.sign(z) * np.floor(np.abs(z) + 0.5)
            got = o.zigzag[c * N:(c + 1) * N]
            bad = got != want
            dist = np.abs(np.abs(z) % 1.0 - 0.5)                       # distance of the quotient from a rounding tie
            assert np.all(np.abs(got - want)[bad] == 1), (q, c)
            assert np.all(dist[bad] < 2e-3), (q, c, int(bad.sum()), float(dist[bad].max()))
            seen_large = max(seen_large, float(np.abs(z[:, 1:]).max()))
            seen_near += int((dist[:, 1:] < 2e-3).sum())
    assert seen_large > 400 and seen_near > 20   # the inputs do reach large quotients and do come near ties


def test_oracle_standard_eob_and_tables():
    """EOB is omitted after a non-zero coefficient 63; run 3 / size 4 uses the 16-bit Annex K code."""
    rgb = ol.lcg_frame(64, 64, 3)
    ql, qc = ol.quant_tables(95)
    o = ol.oracle_std_encode(rgb, ql, qc, KEEP)
    strict_like = ol.oracle_entropy(o.zigzag)  # the reference's entropy rules on the same coefficients
    n63 = int((o.zigzag[:, 63] != 0).sum())
    assert n63 > 0
    # every unit with c63 != 0 saves its 4-bit EOB (luma 1010 / chroma 00 -> 4 or 2 bits)
    N = o.n_blocks
    saved = 4 * int((o.zigzag[:N, 63] != 0).sum()) + 2 * int((o.zigzag[N:, 63] != 0).sum())
    assert o.n_bits <= strict_like[1] - saved  # further savings: one bit per 17-bit typo hit
    assert o.n_bits == int(o.unit_bits.sum())


# ------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("W,H,q,kind", [(253, 254, 50, "fruit"), (640, 360, 50, "lcg"), (100, 37, 90, "lcg"),
                                        (1920, 1080, 75, "smooth"), (8, 8, 100, "lcg"), (333, 65, 25, "smooth"),
                                        (512, 512, 100, "lcg"), (3840, 2160, 50, "lcg"), (2048, 72, 90, "extremes")])
def test_gpu_standard_mode_equals_checker(jpeg, enc, W, H, q, kind):
    rgb = (ol.read_ppm(os.path.join(GOLD, "fruit.ppm")) if kind == "fruit"
           else ol.lcg_frame(W, H, 7) if kind == "lcg" else smooth_frame(W, H, 2))
    if kind == "extremes":
        # whole tiles inside the image (the colour conversion runs on the matrix units there) filled with the corner colours
        # of the RGB cube and their neighbours: every extreme of Y, Cb and Cr, rounding at both ends of the range
        corners = np.array([[r, g, b] for r in (0, 1, 254, 255) for g in (0, 1, 254, 255) for b in (0, 1, 254, 255)], np.uint8)
        rgb = corners[np.random.default_rng(5).integers(0, len(corners), (H, W))]
    ql, qc = ol.quant_tables(q)
    enc.set_quant(ql, qc)
    o = ol.oracle_std_encode(rgb, ql, qc, KEEP)
    cf = enc.probe_coefficients(rgb, jpeg.F_STANDARD)
    assert np.array_equal(cf.astype(np.int32), o.zigzag)
    bits, nb = enc.encode_scan(rgb, jpeg.F_STANDARD)
    assert nb[0] == o.n_bits
    assert np.array_equal(bits[0], o.bits)
    # MI355_F_CDS is ignored in standard mode (4:4:4)
    bits2, nb2 = enc.encode_scan(rgb, jpeg.F_STANDARD | jpeg.F_CDS)
    assert nb2[0] == nb[0] and np.array_equal(bits2[0], bits[0])
    enc.set_quality(50)


@pytest.mark.gpu
def test_gpu_standard_jfif_decodes(jpeg, enc):
    rgb = smooth_frame(640, 480, 5)
    ql, qc = ol.quant_tables(85)
    enc.set_quant(ql, qc)
    got = enc.encode_jfif(rgb, jpeg.F_STANDARD)
    o = ol.oracle_std_encode(rgb, ql, qc)
    assert got == ol.jfif_frame(o.bits, o.n_bits, 640, 480, ql, qc)
    dec = pil_decode(got)
    assert psnr(dec, rgb) > 35.0
    assert psnr(dec, pil_decode(pil_encode(rgb, ql, qc))) > 45.0
    enc.set_quality(50)


@pytest.mark.gpu
def test_gpu_standard_mode_has_nothing_to_verify(jpeg, monkeypatch):
    """Standard mode is DEFINED by its arithmetic (3-digit fixed-point DCT on the matrix units, fp32 quotient): the
    accept margins of the strict screen play no part in it -- with the margins blown up (debug factor: every strict
    coefficient would go to the second look) the standard results do not change and neither counter moves."""
    monkeypatch.setenv("MI355_JPEG_SCREEN_TAU_SCALE", "1e9")
    e2 = jpeg.Encoder(0)
    e2.screen_stats(reset=True)
    for (W, H, q) in [(320, 200, 50), (100, 37, 92)]:
        rgb = ol.lcg_frame(W, H, 11)
        ql, qc = ol.quant_tables(q)
        e2.set_quant(ql, qc)
        o = ol.oracle_std_encode(rgb, ql, qc, KEEP)
        assert np.array_equal(e2.probe_coefficients(rgb, jpeg.F_STANDARD).astype(np.int32), o.zigzag)
        bits, nb = e2.encode_scan(rgb, jpeg.F_STANDARD)
        assert nb[0] == o.n_bits and np.array_equal(bits[0], o.bits)
    assert e2.screen_stats() == (0, 0)
    e2.close()


def test_checker_quantiser_is_within_one_of_the_exact_quotient():
    """What the fp32 definition costs in accuracy: against round-half-away of the exact 2^-39 fixed-point DCT divided by Q
    the defined value differs by at most 1, and only when the exact quotient sits within 2e-3 of a half-integer."""
    rng = np.random.default_rng(3)
    T = ol.std_dct_table().astype(object)
    for q in (50, 90, 100):
        ql, qc = ol.quant_tables(q)
        rgb = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
        o = ol.oracle_std_encode(rgb, ql, qc, KEEP)
        # luma plane as the checker converts it
        y = ol.std_csc(rgb)[..., 0] - 128
        zz = ol.zigzag_order()
        bad = 0
        for blk in range(64):
            by, bx = divmod(blk, 8)
            p = y[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].reshape(64).astype(object)
            for R in range(1, 64):
                Y = int(sum(int(T[R][s]) * int(p[s]) for s in range(64)))
                D = int(ql.reshape(64)[zz[R]]) << 39
                exact = (2 * abs(Y) + D) // (2 * D) * (1 if Y >= 0 else -1)
                got = int(o.zigzag[blk, R])
                if got != exact:
                    bad += 1
                    frac = abs(abs(Y) / D - np.floor(abs(Y) / D) - 0.5)
                    assert abs(got - exact) == 1 and frac < 2e-3, (q, blk, R, got, exact, frac)
        assert bad < 64 * 63 // 100


@pytest.mark.gpu
def test_gpu_standard_batch_and_strict_interleaved(jpeg, enc):
    """Standard and strict calls on one context do not disturb each other; batches equal single frames."""
    enc.set_quality(50)
    ql, qc = ol.quant_tables(50)
    frames = np.stack([ol.lcg_frame(256, 128, s) for s in (1, 2, 3)])
    want_std = [ol.oracle_std_encode(f, ql, qc) for f in frames]
    want_strict = [ol.oracle_encode(f) for f in frames]
    for _ in range(2):
        bits, nb = enc.encode_scan(frames, jpeg.F_STANDARD)
        for f in range(3):
            assert nb[f] == want_std[f].n_bits and np.array_equal(bits[f], want_std[f].bits)
        bits, nb = enc.encode_scan(frames, jpeg.F_CDS)
        for f in range(3):
            assert nb[f] == want_strict[f].n_bits and np.array_equal(bits[f], want_strict[f].bits)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,q,kind", [(253, 254, 50, "fruit"), (640, 368, 50, "lcg"), (100, 37, 90, "lcg"),
                                        (1920, 1080, 75, "smooth"), (16, 16, 100, "lcg"), (8, 9, 60, "lcg"),
                                        (333, 65, 25, "smooth"), (1024, 1024, 95, "lcg"), (1040, 16, 80, "lcg"),
                                        (3840, 2160, 50, "lcg")])
def test_gpu_420_equals_checker(jpeg, enc, W, H, q, kind):
    """Real 4:2:0 (MI355_F_STANDARD | MI355_F_420): scan bits and coefficients equal the checker's,
    incl. sizes that need mirror padding to 16, a last tile with empty luma quarter-tiles, one-MCU images."""
    rgb = (ol.read_ppm(os.path.join(GOLD, "fruit.ppm")) if kind == "fruit"
           else ol.lcg_frame(W, H, 7) if kind == "lcg" else smooth_frame(W, H, 2))
    ql, qc = ol.quant_tables(q)
    enc.set_quant(ql, qc)
    flags = jpeg.F_STANDARD | jpeg.F_420
    o = ol.oracle_std_encode(rgb, ql, qc, KEEP, subsample=1)
    cf = enc.probe_coefficients(rgb, flags)
    assert cf.shape == o.zigzag.shape
    assert np.array_equal(cf.astype(np.int32), o.zigzag)
    bits, nb = enc.encode_scan(rgb, flags)
    assert nb[0] == o.n_bits
    assert np.array_equal(bits[0], o.bits)
    enc.set_quality(50)


@pytest.mark.gpu
def test_gpu_420_jfif_decodes_and_batches(jpeg, enc):
    ql, qc = ol.quant_tables(85)
    enc.set_quant(ql, qc)
    flags = jpeg.F_STANDARD | jpeg.F_420
    rgb = smooth_frame(650, 490, 5)
    got = enc.encode_jfif(rgb, flags)
    o = ol.oracle_std_encode(rgb, ql, qc, subsample=1)
    assert got == ol.jfif_frame(o.bits, o.n_bits, 650, 490, ql, qc, 1)
    dec = pil_decode(got)
    assert dec.shape == rgb.shape and psnr(dec, rgb) > 33.0
    assert psnr(dec, pil_decode(pil_encode(rgb, ql, qc, 1))) > 44.0
    frames = np.stack([ol.lcg_frame(272, 144, s) for s in (1, 2, 3, 4, 5)])
    bits, nb = enc.encode_scan(frames, flags)
    for f in range(5):
        w = ol.oracle_std_encode(frames[f], ql, qc, subsample=1)
        assert nb[f] == w.n_bits and np.array_equal(bits[f], w.bits)
    with pytest.raises(jpeg.JpegError):  # 4:2:0 MCUs exist in standard mode only
        enc.encode_scan(rgb, jpeg.F_420)
    enc.set_quality(50)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,q,ss,kind", [(333, 201, 85, 0, "smooth"), (333, 201, 85, 1, "smooth"), (1920, 1080, 75, 1, "lcg"),
                                          (640, 368, 50, 0, "lcg"), (16, 16, 90, 1, "lcg"), (1024, 1024, 100, 0, "lcg"),
                                          (3840, 2160, 50, 1, "lcg")])
def test_gpu_restart_intervals_equal_checker(jpeg, enc, W, H, q, ss, kind):
    """MI355_F_RESTART: the device writes the same file as the checker (DRI, aligned intervals, RSTm
    inserted together with the byte stuffing); Pillow decodes it to the pixels of the plain file."""
    rgb = ol.lcg_frame(W, H, 3) if kind == "lcg" else smooth_frame(W, H, 4)
    ql, qc = ol.quant_tables(q)
    enc.set_quant(ql, qc)
    flags = jpeg.F_STANDARD | (jpeg.F_420 if ss else 0)
    got = enc.encode_jfif(rgb, flags | jpeg.F_RESTART)
    assert got == ol.oracle_std_jfif_restart(rgb, ql, qc, ss, 64)
    if W * H <= 1920 * 1080:
        assert np.array_equal(pil_decode(got), pil_decode(enc.encode_jfif(rgb, flags)))
    with pytest.raises(jpeg.JpegError):  # restart intervals exist in standard mode only
        enc.encode_jfif(rgb, jpeg.F_RESTART)
    enc.set_quality(50)


@pytest.mark.gpu
def test_gpu_standard_randomised_sweep(jpeg, enc):
    """60 random cases over size (ragged, tiny, wide), quality, sampling, restart intervals and content."""
    rng = np.random.default_rng(4202)
    for it in range(60):
        W = int(rng.choice([8, 16, 24, 40, 64, 100, 127, 255, 256, 320, 511, 640, 1000]))
        H = int(rng.choice([8, 9, 16, 31, 48, 64, 100, 129, 240]))
        ss = int(rng.integers(0, 2))
        A = 16 if ss else 8
        if (W + A - 1) // A * A - W > W or (H + A - 1) // A * A - H > H:
            continue
        kind = it % 4
        if kind == 0:
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == 1:
            rgb = smooth_frame(W, H, it)
        elif kind == 2:
            rgb = np.full((H, W, 3), rng.integers(0, 256, 3), np.uint8)
        else:
            rgb = (rng.integers(0, 2, (H, W, 1)) * rng.integers(100, 256)).astype(np.uint8).repeat(3, 2)
        q = int(rng.choice([5, 25, 50, 75, 90, 97, 100]))
        ql, qc = ol.quant_tables(q)
        enc.set_quant(ql, qc)
        flags = jpeg.F_STANDARD | (jpeg.F_420 if ss else 0)
        try:
            o = ol.oracle_std_encode(rgb, ql, qc, 0, subsample=ss)
        except RuntimeError:  # a size category outside the tables (binary images at q100): both refuse
            with pytest.raises(jpeg.JpegError):
                enc.encode_scan(rgb, flags)
            continue
        bits, nb = enc.encode_scan(rgb, flags)
        assert nb[0] == o.n_bits and np.array_equal(bits[0], o.bits), (it, W, H, q, ss, kind)
        if it % 3 == 0:
            assert enc.encode_jfif(rgb, flags | jpeg.F_RESTART) == ol.oracle_std_jfif_restart(rgb, ql, qc, ss, 64), \
                (it, W, H, q, ss, kind)
    enc.set_quality(50)


@pytest.mark.gpu
@pytest.mark.parametrize("flags_extra", [0, 8], ids=["plain", "restart"])
def test_gpu_tiny_420_noise_at_q100_fits_the_flag_aware_bound(jpeg, enc, flags_extra):
    """an 8x8 image in 4:2:0 is one MCU of SIX units (mi355_jpeg_scan_bound counted three); the
    one-call file entry point must not run out of capacity on tiny noisy images at high quality."""
    flags = jpeg.F_STANDARD | jpeg.F_420 | flags_extra
    assert jpeg.scan_bound(8, 8, flags) >= 2 * jpeg.scan_bound(8, 8, 0) - 16
    rng = np.random.default_rng(5)
    ql, qc = ol.quant_tables(100)
    enc.set_quant(ql, qc)
    for W, H in [(8, 8), (9, 16), (24, 8)]:
        for _ in range(4):
            rgb = (rng.integers(0, 2, (H, W, 3)) * 255).astype(np.uint8)
            got = enc.encode_jfif(rgb, flags)
            if flags_extra:
                assert got == ol.oracle_std_jfif_restart(rgb, ql, qc, subsample=1)
            else:
                o = ol.oracle_std_encode(rgb, ql, qc, subsample=1)
                assert got == ol.jfif_frame(o.bits, o.n_bits, W, H, ql, qc, 1)
            assert pil_decode(got).shape == rgb.shape
    enc.set_quality(50)


@pytest.mark.gpu
def test_gpu_standard_420_tiled_fruit_golden(jpeg, enc):
    """SURVEY §8(d)'s natural-statistics input in the decodable mode: fruit.ppm tiled to 3840x2160, 4:2:0, q50, against the
    fixture the checker wrote (tests/golden/cases.json, tools/make_golden.py --tiled-fruit) -- the same record bench.py
    gates its `other_configs` leg on -- and through Pillow: the file decodes, as far from the source as Pillow's own
    encode of the same pixels with the same tables (within 0.15 dB), and within 40 dB of that file's decode (measured:
    43.0 dB; fruit.ppm is close to noise, the worst case for two encoders whose DCT and colour conversion round
    differently -- the smooth frames of the tests above reach 44-45 dB)."""
    import hashlib
    import json
    from conftest import case_input
    case = [c for c in json.load(open(os.path.join(GOLD, "cases.json"))) if c["name"] == "std420_fruit_tiled_3840x2160_q50"][0]
    rgb = case_input(case)
    ql, qc = ol.quant_tables(case["quality"])
    enc.set_quant(ql, qc)
    flags = jpeg.F_STANDARD | jpeg.F_420
    bits, nb = enc.encode_scan(rgb, flags, cap=(case["n_bits"] + 7) // 8 + 64)
    assert nb[0] == case["n_bits"]
    assert hashlib.sha256(bits[0][:(nb[0] + 7) // 8].tobytes()).hexdigest() == case["sha256_packed_bits"]
    ours = pil_decode(enc.encode_jfif(rgb, flags)).astype(np.float64)
    theirs = pil_decode(pil_encode(rgb, ql, qc, subsample=1)).astype(np.float64)
    src = rgb.astype(np.float64)
    psnr = lambda a, b: 10 * np.log10(255.0 ** 2 / np.mean((a - b) ** 2))
    # both files are the same picture at the same quality: equally far from the source, and close to each other
    assert abs(psnr(ours, src) - psnr(theirs, src)) < 0.15, (psnr(ours, src), psnr(theirs, src))
    assert psnr(ours, theirs) >= 40.0, psnr(ours, theirs)
