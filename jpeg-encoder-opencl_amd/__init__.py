"""MI355X-native strict JPEG encode path -- Python binding of the C ABI.

The product is ``libmi355jpeg.so`` (hand-written HIP kernels for gfx950 behind
``include/mi355_jpeg.h``).  This module is the thin ctypes layer that tests and
``bench.py`` use; it holds no compute of its own and there is NO CPU fallback:
if the library is missing or no gfx950 device is usable, calls raise.

The directory name contains a hyphen, so import it with::

    import importlib; jpeg = importlib.import_module("jpeg-encoder-opencl_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_JPEG_LIB") or os.path.join(_HERE, "libmi355jpeg.so")  # (override: A/B runs of two builds)

F_CDS = 1
F_STANDARD = 2  # decodable baseline JPEG (not a behaviour of the reference), see include/mi355_jpeg.h
F_420 = 4       # with F_STANDARD: real 4:2:0 MCUs
F_RESTART = 8   # with F_STANDARD: restart intervals of 64 MCUs (DRI/RSTm written by encode_jfif)
F_DEFAULT = F_CDS

OK, E_ARG, E_NO_DEVICE, E_CAPACITY, E_CATEGORY, E_ALLOC, E_TABLE, E_INTERNAL, E_NOT_ENCODED, E_HIP = 0, -1, -2, -3, -4, -5, -6, -7, -8, -100
# per-frame verdicts in the bit-count array of a batched call (include/mi355_jpeg.h)
BITS_CAPACITY, BITS_CATEGORY, BITS_FLAGGED = 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFE


class HuffTable(C.Structure):
    _fields_ = [("code", C.c_uint32 * 256), ("len", C.c_uint8 * 256)]


class Timings(C.Structure):
    _fields_ = [("transform_ms", C.c_float), ("size_ms", C.c_float), ("scan_ms", C.c_float),
                ("emit_ms", C.c_float), ("total_ms", C.c_float)]


class ScreenStats(C.Structure):
    _fields_ = [("second_looks", C.c_uint64), ("exact_units", C.c_uint64), ("rewalked_units", C.c_uint64),
                ("general_passes", C.c_uint64)]


class JpegError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("mi355_jpeg: %s (status %d)" % (msg, status))
        self.status = status


# Every symbol include/mi355_jpeg.h declares (tests check the library exports all).
ABI_SYMBOLS = [
    "mi355_jpeg_abi_version", "mi355_jpeg_strerror", "mi355_jpeg_device_count", "mi355_jpeg_create",
    "mi355_jpeg_destroy", "mi355_jpeg_set_quant", "mi355_jpeg_set_quality", "mi355_jpeg_set_huffman",
    "mi355_jpeg_get_quant", "mi355_jpeg_get_huffman", "mi355_jpeg_reference_huffman", "mi355_jpeg_padded_size", "mi355_jpeg_scan_bound",
    "mi355_jpeg_encode_scan", "mi355_jpeg_encode_scan_device", "mi355_jpeg_sync", "mi355_jpeg_encode_jfif",
    "mi355_jpeg_probe_samples", "mi355_jpeg_probe_coefficients", "mi355_jpeg_probe_unit_bits",
    "mi355_jpeg_entropy_only", "mi355_jpeg_set_profiling", "mi355_jpeg_last_timings",
    "mi355_jpeg_profile_summary", "mi355_jpeg_synth_lcg_device", "mi355_jpeg_stuff_device", "mi355_jpeg_pool_create", "mi355_jpeg_pool_destroy", "mi355_jpeg_pool_workers",
    "mi355_jpeg_pool_set_quant", "mi355_jpeg_pool_set_quality", "mi355_jpeg_pool_set_huffman", "mi355_jpeg_pool_encode", "mi355_jpeg_pool_encode_ex",
    "mi355_jpeg_pool_register", "mi355_jpeg_pool_unregister", "mi355_jpeg_pool_debug_counts",
    "mi355_jpeg_set_encode_waves", "mi355_jpeg_wrap_jfif", "mi355_jpeg_scan_bound_flags",
    "mi355_jpeg_last_call_launches", "mi355_jpeg_screen_stats",
    # the reference's stage functions one by one (host/mi355_stage_api.cpp wraps them in the reference's signatures)
    "mi355_jpeg_stage_csc", "mi355_jpeg_stage_cds", "mi355_jpeg_stage_copy_larger", "mi355_jpeg_stage_mirror_pad",
    "mi355_jpeg_stage_to_double", "mi355_jpeg_stage_subtract", "mi355_jpeg_stage_dct", "mi355_jpeg_stage_quantize",
    "mi355_jpeg_stage_blocks", "mi355_jpeg_stage_zigzag", "mi355_jpeg_stage_rle", "mi355_jpeg_stage_huffman",
]

_lib = None


def build():
    """Compile the HIP library in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    """The loaded C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JpegError(E_NO_DEVICE, "libmi355jpeg.so not built (run `make -C %s`)" % _HERE)
        # PyTorch-ROCm ships its own libamdhip64.so.7; two HIP runtimes in one process
        # cannot both own the GPU.  Loading torch first makes this library bind to the
        # runtime torch uses (same SONAME), so torch tensors and our kernels share it.
        if os.environ.get("MI355_JPEG_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        vp, u32, u64p, sz = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_size_t
        L.mi355_jpeg_abi_version.restype = C.c_int
        L.mi355_jpeg_strerror.restype = C.c_char_p
        L.mi355_jpeg_strerror.argtypes = [C.c_int]
        L.mi355_jpeg_device_count.restype = C.c_int
        L.mi355_jpeg_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.mi355_jpeg_destroy.argtypes = [vp]
        L.mi355_jpeg_destroy.restype = None
        L.mi355_jpeg_set_quant.argtypes = [vp, vp, vp]
        L.mi355_jpeg_set_quality.argtypes = [vp, C.c_int]
        L.mi355_jpeg_set_huffman.argtypes = [vp, C.c_int, C.POINTER(HuffTable)]
        L.mi355_jpeg_set_encode_waves.argtypes = [vp, u32]
        L.mi355_jpeg_get_quant.argtypes = [vp, vp, vp]
        L.mi355_jpeg_get_huffman.argtypes = [vp, C.c_int, C.POINTER(HuffTable)]
        L.mi355_jpeg_reference_huffman.argtypes = [C.c_int, C.POINTER(HuffTable)]
        L.mi355_jpeg_padded_size.argtypes = [u32, u32, C.POINTER(u32), C.POINTER(u32)]
        L.mi355_jpeg_padded_size.restype = None
        L.mi355_jpeg_scan_bound.argtypes = [u32, u32]
        L.mi355_jpeg_scan_bound.restype = sz
        L.mi355_jpeg_scan_bound_flags.argtypes = [u32, u32, u32]
        L.mi355_jpeg_scan_bound_flags.restype = sz
        L.mi355_jpeg_last_call_launches.argtypes = [vp, C.POINTER(u32)]
        L.mi355_jpeg_screen_stats.argtypes = [vp, vp, C.POINTER(ScreenStats), C.c_int]
        L.mi355_jpeg_encode_scan.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, u64p]
        L.mi355_jpeg_encode_scan_device.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, vp, vp]
        L.mi355_jpeg_sync.argtypes = [vp, vp]
        L.mi355_jpeg_encode_jfif.argtypes = [vp, vp, u32, u32, u32, vp, sz, C.POINTER(sz)]
        L.mi355_jpeg_wrap_jfif.argtypes = [vp, vp, C.c_uint64, u32, u32, u32, vp, sz, C.POINTER(sz)]
        L.mi355_jpeg_probe_samples.argtypes = [vp, vp, u32, u32, u32, vp]
        L.mi355_jpeg_probe_coefficients.argtypes = [vp, vp, u32, u32, u32, vp]
        L.mi355_jpeg_probe_unit_bits.argtypes = [vp, vp, u32, u32, u32, vp]
        L.mi355_jpeg_entropy_only.argtypes = [vp, vp, u32, vp, sz, u64p]
        L.mi355_jpeg_set_profiling.argtypes = [vp, C.c_int]
        L.mi355_jpeg_last_timings.argtypes = [vp, C.POINTER(Timings)]
        L.mi355_jpeg_profile_summary.argtypes = [vp, C.POINTER(Timings), C.POINTER(u32)]
        L.mi355_jpeg_synth_lcg_device.argtypes = [vp, vp, sz, u32, u32, vp]
        L.mi355_jpeg_stuff_device.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
        L.mi355_jpeg_pool_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.mi355_jpeg_pool_destroy.argtypes = [vp]
        L.mi355_jpeg_pool_destroy.restype = None
        L.mi355_jpeg_pool_workers.argtypes = [vp]
        L.mi355_jpeg_pool_set_quant.argtypes = [vp, vp, vp]
        L.mi355_jpeg_pool_set_quality.argtypes = [vp, C.c_int]
        if hasattr(L, "mi355_jpeg_pool_set_huffman"):  # (absent from an ABI 3 build loaded through MI355_JPEG_LIB for an A/B run)
            L.mi355_jpeg_pool_set_huffman.argtypes = [vp, C.c_int, C.POINTER(HuffTable)]
        L.mi355_jpeg_pool_encode.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, u64p, C.POINTER(C.c_double)]
        L.mi355_jpeg_pool_encode_ex.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, u64p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.mi355_jpeg_pool_register.argtypes = [vp, vp, sz]
        L.mi355_jpeg_pool_unregister.argtypes = [vp, vp]
        L.mi355_jpeg_pool_debug_counts.argtypes = [vp, u64p]
        for name in ABI_SYMBOLS:  # fail at load time, not at first use, if a symbol is missing
            if os.environ.get("MI355_JPEG_LIB") and name == "mi355_jpeg_pool_set_huffman":
                continue  # an older build in an A/B run
            getattr(L, name)
        _lib = L
    return _lib


def _check(status):
    if status != OK:
        raise JpegError(status, lib().mi355_jpeg_strerror(status).decode())


def device_count():
    return lib().mi355_jpeg_device_count()


def reference_huffman(table):
    """(code, len) arrays of the reference's table (host only)."""
    t = HuffTable()
    _check(lib().mi355_jpeg_reference_huffman(table, C.byref(t)))
    return np.array(t.code, np.uint32), np.array(t.len, np.uint8)


def scan_bound(W, H, flags=0):
    return lib().mi355_jpeg_scan_bound_flags(W, H, flags)


class Encoder:
    """One encode context on one GPU (single owner, like the C context)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().mi355_jpeg_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().mi355_jpeg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables
    def set_quant(self, qlum, qchrom):
        qlum = np.ascontiguousarray(qlum, np.uint32).reshape(64)
        qchrom = np.ascontiguousarray(qchrom, np.uint32).reshape(64)
        _check(lib().mi355_jpeg_set_quant(self._h, qlum.ctypes.data, qchrom.ctypes.data))

    def set_quality(self, quality):
        _check(lib().mi355_jpeg_set_quality(self._h, quality))

    def set_encode_waves(self, waves):
        """Persistent waves of the block-encode kernel per call (0 = default: fill the device)."""
        _check(lib().mi355_jpeg_set_encode_waves(self._h, int(waves)))

    def get_quant(self):
        ql, qc = np.zeros(64, np.uint32), np.zeros(64, np.uint32)
        _check(lib().mi355_jpeg_get_quant(self._h, ql.ctypes.data, qc.ctypes.data))
        return ql, qc

    def get_huffman(self, table):
        t = HuffTable()
        _check(lib().mi355_jpeg_get_huffman(self._h, table, C.byref(t)))
        return np.array(t.code, np.uint32), np.array(t.len, np.uint8)

    def set_huffman(self, table, code=None, length=None):
        if code is None:
            _check(lib().mi355_jpeg_set_huffman(self._h, table, None))
            return
        t = HuffTable()
        for i in range(256):
            t.code[i] = int(code[i])
            t.len[i] = int(length[i])
        _check(lib().mi355_jpeg_set_huffman(self._h, table, C.byref(t)))

    # ---- hot path, host buffers
    def encode_scan(self, rgb, flags=F_DEFAULT, cap=None):
        """rgb: uint8 [H,W,3] or [n,H,W,3].  Returns (list of packed-bit arrays, list of bit counts)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        if rgb.ndim == 3:
            rgb = rgb[None]
        n, H, W, _ = rgb.shape
        if cap is None:
            cap = scan_bound(W, H, flags)
        out = np.zeros((n, cap), np.uint8)
        bits = (C.c_uint64 * n)()
        _check(lib().mi355_jpeg_encode_scan(self._h, rgb.ctypes.data, W, H, n, flags, out.ctypes.data,
                                            cap, bits))
        return [out[f, :(bits[f] + 7) // 8].copy() for f in range(n)], [int(b) for b in bits]

    def encode_jfif(self, rgb, flags=F_DEFAULT):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        cap = 2 * scan_bound(W, H, flags) + 4096
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        _check(lib().mi355_jpeg_encode_jfif(self._h, rgb.ctypes.data, W, H, flags, out.ctypes.data, cap,
                                            C.byref(n)))
        return out[:n.value].tobytes()

    def wrap_jfif(self, scan, n_bits, W, H, flags=F_DEFAULT):
        """Container around a host-resident scan (header, stuffed bytes, EOI)."""
        scan = np.ascontiguousarray(scan, np.uint8)
        cap = 2 * scan.size + 4096
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        _check(lib().mi355_jpeg_wrap_jfif(self._h, scan.ctypes.data, n_bits, W, H, flags, out.ctypes.data, cap,
                                          C.byref(n)))
        return out[:n.value].tobytes()

    # ---- hot path, device buffers (raw pointers, e.g. torch tensor .data_ptr())
    def encode_scan_device(self, d_rgb, W, H, n_frames, d_out, out_stride, d_bits, flags=F_DEFAULT, stream=0):
        _check(lib().mi355_jpeg_encode_scan_device(self._h, d_rgb, W, H, n_frames, flags, d_out, out_stride,
                                                   d_bits, stream))

    def sync(self, stream=0):
        _check(lib().mi355_jpeg_sync(self._h, stream))

    # ---- either side of the path
    def synth_lcg_device(self, d_dst, frame_bytes, n_frames, seed0, stream=0):
        """Fill device memory with the pinned LCG frames (seed = seed0 + frame)."""
        _check(lib().mi355_jpeg_synth_lcg_device(self._h, d_dst, frame_bytes, n_frames, seed0, stream))

    def stuff_device(self, d_scan, d_bits, max_scan_bytes, d_out, cap, d_out_len, stream=0):
        _check(lib().mi355_jpeg_stuff_device(self._h, d_scan, d_bits, max_scan_bytes, d_out, cap, d_out_len, stream))

    # ---- stage probes
    def probe_samples(self, rgb, flags=F_DEFAULT):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        W8, H8 = (W + 7) // 8 * 8, (H + 7) // 8 * 8
        out = np.empty((H8, W8, 3), np.uint8)
        _check(lib().mi355_jpeg_probe_samples(self._h, rgb.ctypes.data, W, H, flags, out.ctypes.data))
        return out

    def probe_coefficients(self, rgb, flags=F_DEFAULT):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        if flags & F_420:
            units = 6 * ((W + 15) // 16) * ((H + 15) // 16)
        else:
            units = 3 * ((W + 7) // 8) * ((H + 7) // 8)
        out = np.empty((units, 64), np.int16)
        _check(lib().mi355_jpeg_probe_coefficients(self._h, rgb.ctypes.data, W, H, flags, out.ctypes.data))
        return out

    def probe_unit_bits(self, rgb, flags=F_DEFAULT):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W, _ = rgb.shape
        N = ((W + 7) // 8) * ((H + 7) // 8)
        out = np.empty(3 * N, np.uint32)
        _check(lib().mi355_jpeg_probe_unit_bits(self._h, rgb.ctypes.data, W, H, flags, out.ctypes.data))
        return out

    def entropy_only(self, zigzag, cap=None):
        zigzag = np.ascontiguousarray(zigzag, np.int16)
        N = zigzag.shape[0] // 3
        if cap is None:
            cap = (3 * N * 1727 + 7) // 8 + 8
        out = np.zeros(cap, np.uint8)
        bits = C.c_uint64()
        _check(lib().mi355_jpeg_entropy_only(self._h, zigzag.ctypes.data, N, out.ctypes.data, cap,
                                             C.byref(bits)))
        return out[:(bits.value + 7) // 8].copy(), bits.value

    # ---- measurement
    def set_profiling(self, mode=1):
        """0 off, 1 events around every stage, 2 around the transform kernel only."""
        _check(lib().mi355_jpeg_set_profiling(self._h, int(mode)))

    def profile_summary(self):
        """(sums in ms over the profiled calls, number of calls)."""
        t, n = Timings(), C.c_uint32()
        _check(lib().mi355_jpeg_profile_summary(self._h, C.byref(t), C.byref(n)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}, n.value

    def last_call_parts(self):
        """Launches of the block-encode kernel the last encode call was split into."""
        n = C.c_uint32()
        _check(lib().mi355_jpeg_last_call_launches(self._h, C.byref(n)))
        return n.value

    def screen_stats(self, reset=False, stream=0):
        """(second looks, units recomputed by the exact fp64 chain) since creation / the last reset."""
        st = ScreenStats()
        _check(lib().mi355_jpeg_screen_stats(self._h, stream, C.byref(st), int(reset)))
        return st.second_looks, st.exact_units

    def walk_stats(self, reset=False, stream=0):
        """(units whose AC string was coded a second time straight into device memory, passes in the general walk
        loop) since creation / the last reset."""
        st = ScreenStats()
        _check(lib().mi355_jpeg_screen_stats(self._h, stream, C.byref(st), int(reset)))
        return st.rewalked_units, st.general_passes

    def last_timings(self):
        t = Timings()
        _check(lib().mi355_jpeg_last_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}


class Pool:
    """Multi-GPU batch driver: host frames in, host scans out, frames sharded over the
    workers (one context + thread per entry of device_ids), no collective."""

    def __init__(self, device_ids=None):
        self._h = C.c_void_p()
        if device_ids is None:
            _check(lib().mi355_jpeg_pool_create(None, 0, C.byref(self._h)))
        else:
            arr = (C.c_int * len(device_ids))(*device_ids)
            _check(lib().mi355_jpeg_pool_create(arr, len(device_ids), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mi355_jpeg_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workers(self):
        return lib().mi355_jpeg_pool_workers(self._h)

    def set_quality(self, quality):
        _check(lib().mi355_jpeg_pool_set_quality(self._h, quality))

    def set_quant(self, qlum, qchrom):
        qlum = np.ascontiguousarray(qlum, np.uint32).reshape(64)
        qchrom = np.ascontiguousarray(qchrom, np.uint32).reshape(64)
        _check(lib().mi355_jpeg_pool_set_quant(self._h, qlum.ctypes.data, qchrom.ctypes.data))

    def register(self, array):
        """Pin a numpy array for the pool's DMA until unregister() / close() (streaming callers: buffers they reuse)."""
        _check(lib().mi355_jpeg_pool_register(self._h, array.ctypes.data, array.nbytes))

    def unregister(self, array):
        _check(lib().mi355_jpeg_pool_unregister(self._h, array.ctypes.data))

    def debug_counts(self):
        """(device/pinned allocations, host registrations, streams + events, encode calls) since the pool exists."""
        c = (C.c_uint64 * 4)()
        _check(lib().mi355_jpeg_pool_debug_counts(self._h, c))
        return tuple(int(x) for x in c)

    def encode_into(self, frames, out, flags=F_DEFAULT):
        """frames [n,H,W,3] uint8 -> out [n,cap] uint8 (both caller-owned, e.g. registered).  Returns
        (bits list, per-frame status list, seconds, return code) without raising on per-frame errors."""
        n, H, W, _ = frames.shape
        bits = (C.c_uint64 * n)()
        st = (C.c_int * n)()
        secs = C.c_double()
        rc = lib().mi355_jpeg_pool_encode_ex(self._h, frames.ctypes.data, W, H, n, flags, out.ctypes.data, out.shape[1],
                                             bits, st, C.byref(secs))
        return [int(b) for b in bits], [int(x) for x in st], secs.value, rc

    def encode(self, frames, flags=F_DEFAULT, cap=None):
        """frames: uint8 [n,H,W,3].  Returns (out [n,cap] uint8, bits list, seconds)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, H, W, _ = frames.shape
        if cap is None:
            cap = scan_bound(W, H)
        out = np.zeros((n, cap), np.uint8)
        bits = (C.c_uint64 * n)()
        secs = C.c_double()
        _check(lib().mi355_jpeg_pool_encode(self._h, frames.ctypes.data, W, H, n, flags, out.ctypes.data, cap,
                                            bits, C.byref(secs)))
        return out, [int(b) for b in bits], secs.value
