"""ctypes bindings for the test oracle (oracle/liboracle.so) and, where it has been
built (this container only), the real reference library (oracle/_ref/libjpegref.so).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

KEEP_ZIGZAG, KEEP_UNIT_BITS, KEEP_U8_STAGES, KEEP_DCT = 1, 2, 4, 8


class OrcResult(C.Structure):
    _fields_ = [
        ("W8", C.c_size_t), ("H8", C.c_size_t), ("n_blocks", C.c_size_t),
        ("n_bits", C.c_uint64), ("bits", C.POINTER(C.c_uint8)), ("bits_bytes", C.c_size_t),
        ("zigzag", C.POINTER(C.c_int32)), ("unit_bits", C.POINTER(C.c_uint32)),
        ("csc", C.POINTER(C.c_uint8)), ("cds", C.POINTER(C.c_uint8)),
        ("padded", C.POINTER(C.c_uint8)), ("dct", C.POINTER(C.c_double)),
        ("stage_us", C.c_double * 9),
    ]


_oracle = None
_ref = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_cos.restype = C.c_double
        L.orc_cos.argtypes = [C.c_int, C.c_int]
        L.orc_scale.restype = C.c_double
        L.orc_scale.argtypes = [C.c_int, C.c_int]
        L.orc_encode.restype = C.c_int
        L.orc_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.POINTER(OrcResult)]
        L.orc_result_free.argtypes = [C.POINTER(OrcResult)]
        L.orc_huff_code.restype = C.c_int
        L.orc_huff_code.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_entropy.restype = C.c_int
        L.orc_entropy.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
        L.orc_lcg_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.orc_std_encode.restype = C.c_int
        L.orc_std_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.POINTER(OrcResult)]
        L.orc_std_csc.restype = None
        L.orc_std_csc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_std_jfif_restart.restype = C.c_long
        L.orc_std_jfif_restart.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_uint, C.c_void_p, C.c_size_t]
        L.orc_jfif_frame_s.restype = C.c_long
        L.orc_jfif_frame_s.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_jfif_frame.restype = C.c_long
        L.orc_jfif_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_dct_block.argtypes = [C.c_void_p]
        L.orc_dct_blocks.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_quant_block.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_csc.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_cds.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.orc_zigzag_order.argtypes = [C.c_void_p]
        L.orc_quant_q50.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_quant_ijg.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _oracle = L
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libjpegref.so"))


def ref():
    """The real reference stage library (built in this container from
    /root/reference/src/utils.cpp in place).  None when not built/loadable."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libjpegref.so")
        if not os.path.exists(path):
            return None
        try:
            L = C.CDLL(path)
        except OSError:
            return None
        L.ref_run.restype = C.c_void_p
        L.ref_run.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_dims.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        for name, rt in [("ref_csc", C.POINTER(C.c_uint8)), ("ref_cds", C.POINTER(C.c_uint8)),
                         ("ref_padded", C.POINTER(C.c_uint8)), ("ref_dct", C.POINTER(C.c_double)),
                         ("ref_zigzag", C.POINTER(C.c_int32)), ("ref_nbits", C.c_uint64),
                         ("ref_bits", C.POINTER(C.c_char)),
                         ("ref_stage_us", C.POINTER(C.c_double))]:
            f = getattr(L, name)
            f.restype = rt
            f.argtypes = [C.c_void_p]
        L.ref_csc_only.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_quant_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_huff_code.restype = C.c_int
        L.ref_huff_code.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.ref_cos.restype = C.c_double
        L.ref_cos.argtypes = [C.c_size_t, C.c_size_t]
        L.ref_scale.restype = C.c_double
        L.ref_scale.argtypes = [C.c_size_t, C.c_size_t]
        _ref = L
    return _ref


# ---------------------------------------------------------------- helpers

def quant_tables(quality=50):
    """(qlum, qchrom) uint32[64] row-major [v][u].  quality 50 = the reference's
    tables (utils.hpp:42-62); anything else = IJG scaling (build convention)."""
    ql = np.zeros(64, np.uint32)
    qc = np.zeros(64, np.uint32)
    if quality == 50:
        oracle().orc_quant_q50(ql.ctypes.data, qc.ctypes.data)
    else:
        oracle().orc_quant_ijg(quality, ql.ctypes.data, qc.ctypes.data)
    return ql, qc


def lcg_frame(W, H, seed=1):
    """Pinned synthetic frame of SURVEY §8d: uint8 [H, W, 3]."""
    buf = np.empty(W * H * 3, np.uint8)
    oracle().orc_lcg_fill(buf.ctypes.data, buf.size, seed)
    return buf.reshape(H, W, 3)


def read_ppm(path):
    """Minimal P6 reader for the reference's 3-line header form (utils.cpp:11-65)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:3] == b"P6\n"
    pos = 3
    while data[pos:pos + 1] == b"#":
        pos = data.index(b"\n", pos) + 1
    end = data.index(b"\n", pos)
    W, H = (int(t) for t in data[pos:end].split())
    pos = end + 1
    end = data.index(b"\n", pos)
    assert int(data[pos:end]) == 255
    pos = end + 1
    return np.frombuffer(data, np.uint8, W * H * 3, pos).reshape(H, W, 3).copy()


class Encoded:
    """Result of one CPU encode, as numpy arrays."""
    pass


def oracle_encode(rgb, qlum=None, qchrom=None, cds_on=True, keep=0):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W, _ = rgb.shape
    if qlum is None:
        qlum, qchrom = quant_tables(50)
    qlum = np.ascontiguousarray(qlum, np.uint32)
    qchrom = np.ascontiguousarray(qchrom, np.uint32)
    res = OrcResult()
    rc = oracle().orc_encode(rgb.ctypes.data, W, H, qlum.ctypes.data, qchrom.ctypes.data,
                             int(cds_on), keep, C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_encode failed: %d" % rc)
    out = Encoded()
    out.W8, out.H8, out.n_blocks, out.n_bits = res.W8, res.H8, res.n_blocks, res.n_bits
    out.bits = np.ctypeslib.as_array(res.bits, (res.bits_bytes,)).copy()
    N = res.n_blocks
    out.zigzag = np.ctypeslib.as_array(res.zigzag, (3 * N, 64)).copy() if res.zigzag else None
    out.unit_bits = np.ctypeslib.as_array(res.unit_bits, (3 * N,)).copy() if res.unit_bits else None
    out.csc = np.ctypeslib.as_array(res.csc, (H, W, 3)).copy() if res.csc else None
    out.cds = np.ctypeslib.as_array(res.cds, (H, W, 3)).copy() if res.cds else None
    out.padded = np.ctypeslib.as_array(res.padded, (res.H8, res.W8, 3)).copy() if res.padded else None
    out.dct = np.ctypeslib.as_array(res.dct, (res.H8, res.W8, 3)).copy() if res.dct else None
    out.stage_us = list(res.stage_us)
    oracle().orc_result_free(C.byref(res))
    return out


def zigzag_order():
    """zz[k] = natural index (v*8+u) of zig-zag position k."""
    zz = np.zeros(64, np.uint8)
    oracle().orc_zigzag_order(zz.ctypes.data)
    return zz


def oracle_entropy(zigzag):
    """performRLE + HuffmanEncoder (the reference's rules) on int32 rows [3N][64]: (packed bits, n_bits)."""
    z = np.ascontiguousarray(zigzag, np.int32)
    N = z.shape[0] // 3
    out, nb, n = C.POINTER(C.c_uint8)(), C.c_size_t(), C.c_uint64()
    rc = oracle().orc_entropy(z.ctypes.data, N, C.byref(out), C.byref(nb), C.byref(n), None)
    if rc != 0:
        raise RuntimeError("orc_entropy failed: %d" % rc)
    return np.ctypeslib.as_array(out, (nb.value,)).copy(), int(n.value)


def std_dct_table():
    """The fixed-point true-DCT table that DEFINES standard mode (tests/golden/std_dct_q39.i64)."""
    return np.fromfile(os.path.join(ROOT, "tests", "golden", "std_dct_q39.i64"), "<i8").reshape(64, 64)


STD_Y = (9798, 19235, 3735)                                     # x 2^-15, sum 2^15
STD_C = ((-5529, -10855, 16384), (16384, -13720, -2664))        # Cb, Cr x 2^-15, each row sums to 0
STD_C420 = ((-2765, -5427, 8192), (8192, -6860, -1332))         # the same / 4 at 16 bits (4:2:0 box filter)


def std_csc(rgb):
    """numpy restatement of standard mode's per-pixel colour conversion (15-bit fixed point, libjpeg's form)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    y = (STD_Y[0] * r + STD_Y[1] * g + STD_Y[2] * b + 16384) >> 15
    cc = [((c[0] * r + c[1] * g + c[2] * b + 16383) >> 15) + 128 for c in STD_C]
    return np.stack([y] + cc, -1)


def std_chroma420(rgb):
    """4:2:0 chroma of standard mode for an image whose sides are even: the box filter of the linear form over
    each 2x2 quad, rounded once.  Returns (H/2, W/2, 2)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    out = []
    for c in STD_C420:
        lin = c[0] * r + c[1] * g + c[2] * b
        quad = lin[0::2, 0::2] + lin[0::2, 1::2] + lin[1::2, 0::2] + lin[1::2, 1::2]
        out.append(((quad + 32767) >> 16) + 128)
    return np.stack(out, -1)


def oracle_std_csc(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.empty_like(rgb)
    oracle().orc_std_csc(rgb.ctypes.data, rgb.size // 3, out.ctypes.data)
    return out


def oracle_std_encode(rgb, qlum, qchrom, keep=0, subsample=0):
    """Standard (decodable) mode of the test oracle; subsample 0 = 4:4:4, 1 = 4:2:0."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W, _ = rgb.shape
    qlum = np.ascontiguousarray(qlum, np.uint32)
    qchrom = np.ascontiguousarray(qchrom, np.uint32)
    dct = np.ascontiguousarray(std_dct_table(), np.int64)
    res = OrcResult()
    rc = oracle().orc_std_encode(rgb.ctypes.data, W, H, qlum.ctypes.data, qchrom.ctypes.data, dct.ctypes.data,
                                 subsample, keep, C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_std_encode failed: %d" % rc)
    out = Encoded()
    out.W8, out.H8, out.n_blocks, out.n_bits = res.W8, res.H8, res.n_blocks, res.n_bits
    out.bits = np.ctypeslib.as_array(res.bits, (res.bits_bytes,)).copy()
    units = (6 if subsample else 3) * res.n_blocks
    out.zigzag = np.ctypeslib.as_array(res.zigzag, (units, 64)).copy() if res.zigzag else None
    out.unit_bits = np.ctypeslib.as_array(res.unit_bits, (units,)).copy() if res.unit_bits else None
    oracle().orc_result_free(C.byref(res))
    return out


def oracle_std_jfif_restart(rgb, qlum, qchrom, subsample=0, interval=64):
    """Whole file of standard mode with restart markers every `interval` MCUs."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W, _ = rgb.shape
    qlum = np.ascontiguousarray(qlum, np.uint32)
    qchrom = np.ascontiguousarray(qchrom, np.uint32)
    dct = np.ascontiguousarray(std_dct_table(), np.int64)
    cap = 4 * W * H * 3 + (1 << 16)
    out = np.empty(cap, np.uint8)
    n = oracle().orc_std_jfif_restart(rgb.ctypes.data, W, H, qlum.ctypes.data, qchrom.ctypes.data, dct.ctypes.data,
                                      subsample, interval, out.ctypes.data, cap)
    if n <= 0:
        raise RuntimeError("orc_std_jfif_restart failed: %d" % n)
    return out[:n].tobytes()


def ref_encode(rgb, qlum=None, qchrom=None, cds_on=True, keep=0):
    """Run the REAL reference CPU path (only where oracle/_ref is built)."""
    L = ref()
    assert L is not None, "oracle/_ref/libjpegref.so not available"
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W, _ = rgb.shape
    if qlum is None:
        qlum = np.zeros(64, np.uint32)
        qchrom = np.zeros(64, np.uint32)
        L.ref_quant_tables(qlum.ctypes.data, qchrom.ctypes.data)
    qlum = np.ascontiguousarray(qlum, np.uint32)
    qchrom = np.ascontiguousarray(qchrom, np.uint32)
    k = (1 if keep & KEEP_U8_STAGES else 0) | (2 if keep & KEEP_DCT else 0)
    h = L.ref_run(rgb.ctypes.data, W, H, qlum.ctypes.data, qchrom.ctypes.data, int(cds_on), k)
    assert h
    out = Encoded()
    w8, h8 = C.c_size_t(), C.c_size_t()
    L.ref_dims(h, C.byref(w8), C.byref(h8))
    out.W8, out.H8 = w8.value, h8.value
    N = out.W8 * out.H8 // 64
    out.n_blocks = N
    out.n_bits = L.ref_nbits(h)
    chars = np.ctypeslib.as_array(C.cast(L.ref_bits(h), C.POINTER(C.c_uint8)), (max(out.n_bits, 1),))
    out.bit_chars = chars[:out.n_bits].copy()
    out.bits = np.packbits(out.bit_chars - ord("0"))
    out.zigzag = np.ctypeslib.as_array(L.ref_zigzag(h), (3 * N, 64)).copy()
    if keep & KEEP_U8_STAGES:
        out.csc = np.ctypeslib.as_array(L.ref_csc(h), (H, W, 3)).copy()
        out.cds = np.ctypeslib.as_array(L.ref_cds(h), (H, W, 3)).copy()
        out.padded = np.ctypeslib.as_array(L.ref_padded(h), (out.H8, out.W8, 3)).copy()
    if keep & KEEP_DCT:
        out.dct = np.ctypeslib.as_array(L.ref_dct(h), (out.H8, out.W8, 3)).copy()
    out.stage_us = [L.ref_stage_us(h)[i] for i in range(9)]
    L.ref_free(h)
    return out


def unpack_bits(packed, n_bits):
    return np.unpackbits(np.asarray(packed, np.uint8))[:n_bits]


def jfif_frame(bits, n_bits, W, H, qlum, qchrom, subsample=0):
    bits = np.ascontiguousarray(bits, np.uint8)
    cap = 2 * bits.size + 2048
    out = np.empty(cap, np.uint8)
    qlum = np.ascontiguousarray(qlum, np.uint32)
    qchrom = np.ascontiguousarray(qchrom, np.uint32)
    n = oracle().orc_jfif_frame_s(bits.ctypes.data, n_bits, W, H, qlum.ctypes.data,
                                  qchrom.ctypes.data, subsample, out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()
