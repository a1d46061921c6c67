/* mi355_jpeg.h -- C ABI of the MI355X-native strict JPEG encode path.
 *
 * Drop-in boundary for the CPU encode path of rusty-electron/jpeg-encoder-opencl:
 * the work JpegEncoderHost does between reading the PPM and discarding the scan
 * bit string (reference src/OpenCLProject_JpegEncoder.cpp:59-225, stage library
 * src/utils.cpp) runs as hand-written HIP kernels for gfx950 behind these entry
 * points.  The reference has no FFI of its own -- its interface is the header
 * pair src/utils.hpp + src/huffman.hpp -- so each entry point names the
 * reference functions it replaces.  Plain pointers and sizes only; no
 * allocation crosses the boundary; errors are negative ints (the reference
 * prints to stdout and returns -1, utils.cpp:17-63, or is UB).
 *
 * Results are bit-identical to the reference CPU path (including its quirks,
 * SURVEY.md Appendix A); there is NO CPU fallback: every compute entry point
 * fails with MI355_E_NO_DEVICE / a HIP error when no gfx950 device is usable.
 *
 * Threading: a context is single-owner.  One context per GPU per host thread;
 * contexts are independent.  The device workspace, the status word and the hand-off
 * records of a context are per context, not per stream: issue the encode calls of ONE
 * context in stream order on ONE stream at a time (sync, or order the streams with an
 * event, before switching to another stream).  Callers that keep several calls in
 * flight use one context per stream (mi355_jpeg_pool does: one context per worker).
 */
#ifndef MI355_JPEG_H
#define MI355_JPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_JPEG_ABI_VERSION 4

typedef struct mi355_jpeg_ctx mi355_jpeg_ctx;

/* No C++ exception crosses this boundary: every entry point catches what the host side may throw (std::bad_alloc and
 * a std::thread that cannot start -> MI355_E_ALLOC, anything else -> MI355_E_INTERNAL).  The reference's counterpart
 * is -1 plus a message on stdout (utils.cpp:17-63). */

enum mi355_jpeg_status {
    MI355_OK = 0,
    MI355_E_ARG = -1,        /* null pointer, zero size, pad wider than the image (reference UB, utils.cpp:215,226) */
    MI355_E_NO_DEVICE = -2,  /* no usable HIP device / kernels not loadable */
    MI355_E_CAPACITY = -3,   /* caller's output buffer too small */
    MI355_E_CATEGORY = -4,   /* a DC size > 11 or AC size > 10: the reference reads past its tables
                                (huffman.hpp:9-23,45-57) -- defined here as an error */
    MI355_E_ALLOC = -5,      /* device or host allocation failed */
    MI355_E_TABLE = -6,      /* malformed quantisation / Huffman table; a container asked for with quantiser
                                entries > 255 (an 8-bit DQT cannot hold them) */
    MI355_E_INTERNAL = -7,   /* an internal invariant failed, or an exception other than an allocation failure was stopped at
                                this boundary (should never happen; results of the call are invalid) */
    MI355_E_NOT_ENCODED = -8, /* per-frame status only (mi355_jpeg_pool_encode_ex): the call ended -- with the error it
                                returns -- before this frame was delivered */
    MI355_E_HIP = -100       /* MI355_E_HIP - hipError_t */
};

/* flags */
#define MI355_F_CDS 1u       /* run the 2x2 chroma averaging (performCDS, utils.cpp:113-141).
                                Set = reference behaviour; clear = "4:4:4 (no subsample)" build convention */
#define MI355_F_STANDARD 2u  /* SURVEY §8 f1 -- NOT a behaviour of the reference: a decodable baseline JPEG.
                                True 8x8 DCT-II (as a fixed-point map of 23 fractional bits evaluated exactly on the
                                matrix units, quotient by Q in fp32: the mode is defined by that arithmetic), colour conversion in
                                15-bit fixed point (libjpeg's form: (c.RGB + half) >> 15, rows summing exactly), 4:4:4 unless MI355_F_420 (MI355_F_CDS is ignored), Annex K code tables proper (without
                                the seven 17-bit entries of huffman.hpp:92-98), EOB omitted after a non-zero
                                coefficient 63.  Files from mi355_jpeg_encode_jfif decode in libjpeg/PIL.
                                CONTRACT of the quantised values (the one tolerance of this mode; tests/test_standard_mode.py
                                asserts exactly this against an independent fp64 DCT): a coefficient equals
                                round-half-away(DCT-II(samples - 128) / Q) computed in exact arithmetic, EXCEPT where
                                that quotient lies within 2e-3 of a rounding tie (k + 1/2), where it may be the other
                                neighbour (off by one).  Within that contract the value is DEFINED by the arithmetic above
                                and is bit-identical across calls, batch shapes and library versions of the same ABI
                                (ABI 3 and later; ABI 2 bitstreams of this mode differ). */
#define MI355_F_420 4u       /* with MI355_F_STANDARD only (MI355_E_ARG otherwise): real 4:2:0 -- 16x16 MCUs of four
                                luma blocks + one Cb + one Cr block (sampling 2x2,1x1,1x1 in SOF0), chroma = the conversion's
                                linear form box-filtered over the 2x2 quad, rounded once; image mirror-padded to multiples of 16.
                                Coefficient probe row order: luma 4*mcu + k (scan order), Cb at 4M + mcu, Cr at
                                5M + mcu, M = MCUs.  (The reference's "subsampling" keeps full-resolution planes and
                                is MI355_F_CDS.) */
#define MI355_F_RESTART 8u   /* with MI355_F_STANDARD only: restart intervals of 64 MCUs (one tile of the device
                                pipeline).  DC predictors start from 0 in every interval and every interval is
                                padded to a byte boundary with 1s; mi355_jpeg_encode_jfif writes DRI and the RSTm
                                markers between the intervals (a decoder can then resynchronise / decode intervals
                                in parallel).  The scan entry points return the aligned intervals WITHOUT markers
                                (markers must not be byte-stuffed, so they are added together with the stuffing);
                                mi355_jpeg_wrap_jfif cannot add them afterwards and refuses the flag. */
#define MI355_F_DEFAULT MI355_F_CDS

/* One Huffman table in the form the kernels consume: index (run<<4)|size,
 * len 0 = no code (the reference's "NULL" strings), code right-aligned, len <= 17
 * (huffman.hpp:92-98 holds seven 17-bit codes). */
typedef struct mi355_huff_table {
    uint32_t code[256];
    uint8_t len[256];
} mi355_huff_table;

/* Per-stage device time of the last encode call, in milliseconds (HIP events on
 * the call's stream).  Mirrors the role of CPUTelemetry (utils.hpp:65-75). */
typedef struct mi355_jpeg_timings {
    /* default (screened) pipeline / exact pipeline (MI355_JPEG_TRANSFORM_MODE=0,1): */
    float transform_ms; /* k_screen_encode: CSC..zig-zag + per-unit RLE/Huffman strings (incl. the rare exact
                           fp64 recomputation) / k_transform (utils.cpp:92-558) */
    float size_ms;      /* k_dc_heads: DC symbols at tile heads / k_unit_sizes (utils.cpp:572-653) */
    float scan_ms;      /* k_tile_scan: exclusive prefix sum of tile bit counts */
    float emit_ms;      /* k_merge / k_emit: final bit string (HuffmanEncoder, utils.cpp:656-698) */
    float total_ms;
} mi355_jpeg_timings;

/* ---- lifetime --------------------------------------------------------- */
int mi355_jpeg_abi_version(void);
const char *mi355_jpeg_strerror(int status);
/* Number of usable devices (0 when there is none; never fails). */
int mi355_jpeg_device_count(void);
/* Creates a context on HIP device `device_id` with the reference's tables:
 * quant_mat_lum/quant_mat_chrom (utils.hpp:42-62) and the four code tables of
 * huffman.hpp (including the 17-bit entries). */
int mi355_jpeg_create(int device_id, mi355_jpeg_ctx **ctx);
void mi355_jpeg_destroy(mi355_jpeg_ctx *ctx);

/* ---- tables (utils.hpp:42-62, huffman.hpp) ----------------------------
 * The setters wait for the whole device (hipDeviceSynchronize) before they rewrite the device-resident
 * tables, so encode calls still in flight on any stream finish with the tables they were issued under;
 * calls issued afterwards see the new set.  (A context is single-owner: do not call a setter from one
 * thread while another thread issues encode calls on the same context.) */
/* qlum/qchrom: 64 entries, row-major [v][u] exactly like quant_mat_lum[8][8];
 * values 1..65535.  Replaces the table arguments of performQuantization
 * (utils.cpp:454). */
int mi355_jpeg_set_quant(mi355_jpeg_ctx *ctx, const uint32_t qlum[64], const uint32_t qchrom[64]);
/* Build convention (not in the reference): IJG quality scaling of the q=50
 * tables, quality 1..100; 50 restores the reference tables. */
int mi355_jpeg_set_quality(mi355_jpeg_ctx *ctx, int quality);
/* table: 0 DC luma, 1 DC chroma, 2 AC luma, 3 AC chroma (DC_LUMA_HUFF_CODES ...
 * AC_CHROMA_HUFF_CODES).  NULL restores the reference table. */
int mi355_jpeg_set_huffman(mi355_jpeg_ctx *ctx, int table, const mi355_huff_table *t);
/* The reference's own code table `table` (huffman.hpp, including the seven 17-bit
 * AC-luma entries); host only, needs no device. */
int mi355_jpeg_reference_huffman(int table, mi355_huff_table *t);
int mi355_jpeg_get_quant(mi355_jpeg_ctx *ctx, uint32_t qlum[64], uint32_t qchrom[64]);
int mi355_jpeg_get_huffman(mi355_jpeg_ctx *ctx, int table, mi355_huff_table *t);

/* ---- tuning --------------------------------------------------------------
 * Persistent waves of the block-encode kernel per call: 0 = default (fill the device: lowest
 * latency for one call at a time), or a multiple of 32 in [32, 8192].  Callers that keep several
 * encode calls in flight on one device (one context + stream each, like bench.py) do better with
 * half the device per call (1024 on MI355X): two calls are then resident side by side and each
 * wave amortises its table set-up over twice as many tiles.  Speed only; results are identical. */
int mi355_jpeg_set_encode_waves(mi355_jpeg_ctx *ctx, uint32_t waves);

/* ---- geometry helpers (getNearest8x8ImageSize, utils.cpp:184-187) ------ */
void mi355_jpeg_padded_size(uint32_t W, uint32_t H, uint32_t *W8, uint32_t *H8);
/* Upper bound in bytes of one frame's packed scan bits (what to size `out` as) in strict mode and
 * standard 4:4:4 (flags without MI355_F_420 / MI355_F_RESTART). */
size_t mi355_jpeg_scan_bound(uint32_t W, uint32_t H);
/* The same for any flags: 4:2:0 pads to 16x16 MCUs of six units, restart intervals pad every interval
 * to a byte boundary. */
size_t mi355_jpeg_scan_bound_flags(uint32_t W, uint32_t H, uint32_t flags);

/* ---- the hot path ------------------------------------------------------
 * rgb: interleaved 8-bit RGB, W*H*3 bytes per frame, row-major, frames
 * contiguous (what readPPMImage returns, utils.cpp:11-65).
 * out: frame f's scan bits are written MSB-first from out + f*out_stride; the
 * last partial byte is zero padded.  bits[f] receives the exact bit count (the
 * length of the string HuffmanEncoder returns, utils.cpp:697).
 *
 * Replaces, per frame: performCSC, performCDS, copyToLargerImage,
 * addReversedPadding, copyUIntToDoubleImage, substractfromAll, performDCT,
 * performQuantization, everyMCUisnow2DArray, performZigZag, performRLE,
 * HuffmanEncoder (src/OpenCLProject_JpegEncoder.cpp:59-225). */

/* Host buffers in, host buffers out (PCIe both ways, synchronous). */
int mi355_jpeg_encode_scan(mi355_jpeg_ctx *ctx, const uint8_t *rgb, uint32_t W, uint32_t H,
                           uint32_t n_frames, uint32_t flags, uint8_t *out, size_t out_stride,
                           uint64_t *bits);

/* Device buffers in, device buffers out, asynchronous on `stream` (a
 * hipStream_t, NULL = default stream).  d_bits: device array of n_frames
 * uint64.  Errors detected on the device (capacity, category) are reported by
 * the next mi355_jpeg_sync(), and PER FRAME in d_bits, with the cause: a frame that holds a coefficient without a
 * code gets d_bits[f] = MI355_BITS_CATEGORY, one whose scan does not fit out_stride MI355_BITS_CAPACITY (its output
 * bytes are undefined; d_bits[f] >= MI355_BITS_FLAGGED tests for either); every other frame of the call is complete
 * and valid, whatever its neighbours hold -- the device workspace of a frame cannot be used up by another one.
 *
 * n_frames: 1 .. 65535, any out_stride >= 8 (a multiple of 4); a batch is never refused for its
 * size: it is cut into parts of ~16 4K frames' worth of pixels, each with 32-bit offsets of its own.
 * Device memory the library allocates for a call (kept for later calls, grown on demand): per frame
 * of a part 4 bytes per unit (units = 8x8 blocks x 3) plus 2 x min(9/16 x out_stride + 4 x units,
 * 216 x units) + 0.25 MiB of string arena; one such set per part while the sets fit a quarter of the
 * free device memory (at most 32 GB), otherwise as many as fit and parts take turns (a little slower:
 * a part then waits for the tail kernels of the part whose set it reuses); plus 12 bytes per tile (64
 * blocks) for every frame.  E.g. 128 4K frames at out_stride = 8 MiB: 8 parts, 5.7 GB; at out_stride =
 * mi355_jpeg_scan_bound (84 MB): 22 GB. */
#define MI355_BITS_CAPACITY UINT64_MAX        /* = MI355_E_CAPACITY for this frame */
#define MI355_BITS_CATEGORY (UINT64_MAX - 1)  /* = MI355_E_CATEGORY for this frame (ABI 4; ABI 3 wrote UINT64_MAX for both) */
#define MI355_BITS_FLAGGED (UINT64_MAX - 1)
int mi355_jpeg_encode_scan_device(mi355_jpeg_ctx *ctx, const void *d_rgb, uint32_t W, uint32_t H,
                                  uint32_t n_frames, uint32_t flags, void *d_out,
                                  size_t out_stride, uint64_t *d_bits, void *stream);
/* Waits for `stream`, then returns the first device-side error of the calls
 * issued since the previous sync (MI355_OK if none).  After MI355_E_CAPACITY / MI355_E_CATEGORY the
 * frames at fault are the ones whose d_bits entry is >= MI355_BITS_FLAGGED; all other frames of those calls are
 * good.  After MI355_E_INTERNAL everything issued since the previous sync must be discarded. */
int mi355_jpeg_sync(mi355_jpeg_ctx *ctx, void *stream);

/* Whole file: build-defined JFIF framing (the reference writes no container,
 * SURVEY.md Appendix C) around the scan of ONE frame: SOI, APP0, DQT, SOF0
 * (H1V1 x3), DHT, SOS, stuffed entropy bytes padded with 1s, EOI. */
int mi355_jpeg_encode_jfif(mi355_jpeg_ctx *ctx, const uint8_t *rgb, uint32_t W, uint32_t H,
                           uint32_t flags, uint8_t *out, size_t cap, size_t *out_len);

/* The same container around a scan that is already in host memory (e.g. one frame of
 * mi355_jpeg_pool_encode): header, byte-stuffed entropy bytes, EOI.  Host only -- framing is data
 * format, not the hot path; `flags` must be the flags the scan was encoded with, the context must
 * hold the same tables.  MI355_E_CAPACITY when cap is too small (*out_len = the size needed). */
int mi355_jpeg_wrap_jfif(mi355_jpeg_ctx *ctx, const uint8_t *scan, uint64_t n_bits, uint32_t W, uint32_t H,
                         uint32_t flags, uint8_t *out, size_t cap, size_t *out_len);

/* ---- stage probes (host buffers; one frame) ----------------------------
 * Let every stage be parity-checked like the reference's per-stage dumps. */
/* Samples entering the transform: after performCSC, performCDS and padding
 * (utils.cpp:92-141,199-233); out: W8*H8*3 bytes interleaved like the
 * reference's padded ppm_t. */
int mi355_jpeg_probe_samples(mi355_jpeg_ctx *ctx, const uint8_t *rgb, uint32_t W, uint32_t H,
                             uint32_t flags, uint8_t *out);
/* Quantised zig-zag coefficients in the reference's row order
 * (zigzag_arr[chan*N + block][64], utils.cpp:482-558); out: 3*N*64 int16. */
int mi355_jpeg_probe_coefficients(mi355_jpeg_ctx *ctx, const uint8_t *rgb, uint32_t W, uint32_t H,
                                  uint32_t flags, int16_t *out);
/* Bits each unit contributes, scan order 3*block+chan; out: 3*N uint32.  Strict mode only
 * (MI355_E_ARG with MI355_F_STANDARD). */
int mi355_jpeg_probe_unit_bits(mi355_jpeg_ctx *ctx, const uint8_t *rgb, uint32_t W, uint32_t H,
                               uint32_t flags, uint32_t *out);
/* Entropy-code caller-supplied coefficients (reference row order, int16) --
 * performRLE + HuffmanEncoder alone (utils.cpp:572-698). */
int mi355_jpeg_entropy_only(mi355_jpeg_ctx *ctx, const int16_t *zigzag, uint32_t n_blocks,
                            uint8_t *out, size_t cap, uint64_t *bits);

/* ---- the reference's stage functions, one by one (utils.hpp:77-137) -----------------------------------
 * Host images in, host images out, each call = one stage kernel on the GPU between an upload and a download.
 * NOT the fast path -- mi355_jpeg_encode_scan* fuses all of it and never materialises these intermediates --
 * but the drop-in for a driver that is written stage by stage like JpegEncoderHost
 * (OpenCLProject_JpegEncoder.cpp:59-225): same in-place semantics, every intermediate bit-identical to the
 * reference's.  host/mi355_stage_api.cpp wraps them in the reference's own signatures.
 * Images: interleaved 3 x u8 per pixel (ppm_t.data) or 3 x double per pixel (ppm_d_t.data), row-major. */
/* performCSC (utils.cpp:92-110), in place. */
int mi355_jpeg_stage_csc(mi355_jpeg_ctx *ctx, uint8_t *img, uint32_t W, uint32_t H);
/* performCDS (utils.cpp:113-141), in place on the UNPADDED image. */
int mi355_jpeg_stage_cds(mi355_jpeg_ctx *ctx, uint8_t *img, uint32_t W, uint32_t H);
/* copyToLargerImage (utils.cpp:199-208): src (W x H) into the top-left corner of dst (W8 x H8); the rest of
 * dst is left as it is, like in the reference. */
int mi355_jpeg_stage_copy_larger(mi355_jpeg_ctx *ctx, const uint8_t *src, uint32_t W, uint32_t H, uint8_t *dst,
                                 uint32_t W8, uint32_t H8);
/* addReversedPadding (utils.cpp:211-233), in place: mirror pad right, then bottom.  MI355_E_ARG when a pad is
 * wider than the image (the reference indexes out of bounds there). */
int mi355_jpeg_stage_mirror_pad(mi355_jpeg_ctx *ctx, uint8_t *img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH);
/* copyUIntToDoubleImage (utils.cpp:236-246). */
int mi355_jpeg_stage_to_double(mi355_jpeg_ctx *ctx, const uint8_t *src, double *dst, uint32_t W, uint32_t H);
/* substractfromAll (utils.cpp:190-196), in place. */
int mi355_jpeg_stage_subtract(mi355_jpeg_ctx *ctx, double *img, uint32_t W, uint32_t H, double value);
/* performDCT (utils.cpp:262-270, 314-348), in place: the ordered in-place fp64 chain per 8x8 block and channel, on
 * whatever doubles the image holds.  W8, H8 multiples of 8. */
int mi355_jpeg_stage_dct(mi355_jpeg_ctx *ctx, double *img, uint32_t W8, uint32_t H8);
/* performQuantization (utils.cpp:454-467), in place, with the tables given ([v][u] like quant_mat_lum). */
int mi355_jpeg_stage_quantize(mi355_jpeg_ctx *ctx, double *img, uint32_t W8, uint32_t H8, const uint32_t qlum[64],
                              const uint32_t qchrom[64]);
/* everyMCUisnow2DArray (utils.cpp:482-498): out = int[3 * N][64], N = W8 * H8 / 64. */
int mi355_jpeg_stage_blocks(mi355_jpeg_ctx *ctx, const double *img, uint32_t W8, uint32_t H8, int32_t *linear);
/* performZigZag (utils.cpp:554-558). */
int mi355_jpeg_stage_zigzag(mi355_jpeg_ctx *ctx, const int32_t *linear, int32_t *zigzag, uint32_t rows);
/* performRLE (utils.cpp:612-620): row r's flat (run, value) pairs at pairs + 128 * r, counts[r] ints of them
 * (an even number <= 128, the final (0,0) included). */
int mi355_jpeg_stage_rle(mi355_jpeg_ctx *ctx, const int32_t *zigzag, uint32_t rows, int32_t *pairs, uint32_t *counts);
/* HuffmanEncoder (utils.cpp:656-698): DC differences from zigzag[row][0], AC symbols from the pair lists AS
 * GIVEN (same layout as mi355_jpeg_stage_rle writes).  Packed bits MSB-first; *bits = the length of the string the
 * reference returns.  A (run, size) the reference has no code for is MI355_E_CATEGORY. */
int mi355_jpeg_stage_huffman(mi355_jpeg_ctx *ctx, const int32_t *zigzag, const int32_t *pairs, const uint32_t *counts,
                             uint32_t rows_per_channel, uint8_t *out, size_t cap, uint64_t *bits);

/* ---- either side of the path ---------------------------------
 * Pinned synthetic input, generated in place on the device: frame f gets the
 * bytes s_{k+1} >> 24 of the LCG s <- s*1664525 + 1013904223 with s_0 = seed0 + f. */
int mi355_jpeg_synth_lcg_device(mi355_jpeg_ctx *ctx, void *d_dst, size_t frame_bytes, uint32_t n_frames,
                                uint32_t seed0, void *stream);
/* JFIF byte stuffing of ONE frame's scan on the device: d_scan / d_bits as written by
 * mi355_jpeg_encode_scan_device; the last partial byte is padded with 1s and every 0xFF is
 * followed by 0x00.  d_out_len (device uint64) receives the stuffed length.  max_scan_bytes
 * bounds ceil(bits/8) (e.g. the scan buffer's stride). */
int mi355_jpeg_stuff_device(mi355_jpeg_ctx *ctx, const void *d_scan, const uint64_t *d_bits, size_t max_scan_bytes,
                            void *d_out, size_t cap, uint64_t *d_out_len, void *stream);

/* ---- multi-GPU batch driver (host frames in, host scans out) -------------
 * The production shape of BASELINE configs[3]: a batch of independent frames sharded
 * across the GPUs of one node, no collective.  One worker thread + one context per
 * entry of device_ids (an id may repeat: several workers on one GPU); each worker pulls
 * chunks of frames from the call's shared cursor and streams them through its device with
 * H2D(k+1) || encode(k) || D2H(k-1) on separate HIP streams.  Same results as
 * mi355_jpeg_encode_scan frame by frame.
 *
 * The pool is persistent: worker threads (pinned to the CPUs of their GPU's NUMA node where
 * /sys/bus/pci/devices/<bus id>/numa_node says which), contexts, streams, events and device
 * buffers are made once and reused by every call; device buffers only grow.  A pool is driven
 * from one host thread at a time. */
typedef struct mi355_jpeg_pool mi355_jpeg_pool;
/* device_ids NULL: every visible device once. */
int mi355_jpeg_pool_create(const int *device_ids, int n_workers, mi355_jpeg_pool **pool);
void mi355_jpeg_pool_destroy(mi355_jpeg_pool *pool);
int mi355_jpeg_pool_workers(mi355_jpeg_pool *pool);
int mi355_jpeg_pool_set_quant(mi355_jpeg_pool *pool, const uint32_t qlum[64], const uint32_t qchrom[64]);
int mi355_jpeg_pool_set_quality(mi355_jpeg_pool *pool, int quality);
/* mi355_jpeg_set_huffman on every context of the pool (ABI 4). */
int mi355_jpeg_pool_set_huffman(mi355_jpeg_pool *pool, int table, const mi355_huff_table *t);
/* Streaming callers: register the host buffers they reuse (frame rings, output slabs) once; encode calls whose
 * rgb / out lie inside a registered range do no registration work of their own.  Memory that is not registered
 * this way is registered for the duration of each call (each buffer as one range, on the calling thread).  The
 * memory must stay allocated until it is unregistered or the pool is destroyed. */
int mi355_jpeg_pool_register(mi355_jpeg_pool *pool, void *ptr, size_t bytes);
int mi355_jpeg_pool_unregister(mi355_jpeg_pool *pool, void *ptr);
/* out: frame f at out + f*out_stride; bits[f] its bit count.  seconds (may be NULL)
 * receives the wall time of the call (PCIe-inclusive). */
int mi355_jpeg_pool_encode(mi355_jpeg_pool *pool, const uint8_t *rgb, uint32_t W, uint32_t H,
                           uint32_t n_frames, uint32_t flags, uint8_t *out, size_t out_stride,
                           uint64_t *bits, double *seconds);
/* The same with a per-frame status array (may be NULL): frame_status[f] = MI355_OK, or MI355_E_CAPACITY /
 * MI355_E_CATEGORY -- each frame's OWN cause -- for a frame that did not fit out_stride / holds a coefficient without
 * a code (bits[f] = MI355_BITS_CAPACITY / MI355_BITS_CATEGORY, out bytes undefined).  Every other frame of the batch
 * is returned in full, whichever worker took it.  The call returns MI355_E_CATEGORY if any frame had that, else
 * MI355_E_CAPACITY if any had that, else MI355_OK -- unless a worker ended on a hard error (HIP, allocation): then
 * that error is returned and the frames it never delivered read MI355_E_NOT_ENCODED / MI355_BITS_CAPACITY (both arrays
 * are initialised to that before the workers start).
 * Frames are handed out to the workers in chunks from a shared cursor (about 256 MB of input, smaller for small
 * batches): a slower GPU takes fewer chunks instead of setting the time of the call. */
int mi355_jpeg_pool_encode_ex(mi355_jpeg_pool *pool, const uint8_t *rgb, uint32_t W, uint32_t H,
                              uint32_t n_frames, uint32_t flags, uint8_t *out, size_t out_stride,
                              uint64_t *bits, int *frame_status, double *seconds);
/* Test hook: what the pool has created since it exists -- counts[0] device / pinned allocations, [1] host-memory
 * registrations, [2] streams + events, [3] encode calls.  A second call of the same shape on a pool whose buffers
 * are registered adds to [3] only. */
int mi355_jpeg_pool_debug_counts(mi355_jpeg_pool *pool, uint64_t counts[4]);

/* ---- measurement ------------------------------------------------------- */
/* mode 0: off (default).  mode 1: HIP events around every stage of each encode
 * call, recorded on the call's stream.  mode 2: events around the transform
 * kernel only (two records per call; what bench.py keeps on during the timed
 * region).  Enabling resets the accumulated sums. */
int mi355_jpeg_set_profiling(mi355_jpeg_ctx *ctx, int mode);
/* Stage times of the last *_device or host encode call (waits for its events). */
int mi355_jpeg_last_timings(mi355_jpeg_ctx *ctx, mi355_jpeg_timings *t);
/* Sums over all encode calls since profiling was enabled (waits for their
 * events); *calls receives the number of calls summed. */
int mi355_jpeg_profile_summary(mi355_jpeg_ctx *ctx, mi355_jpeg_timings *sum, uint32_t *calls);

/* Launch shape of the last encode call: how many launches of the block-encode kernel (the dominant
 * kernel) it was split into.  bench.py divides the event bracket of a call by this. */
int mi355_jpeg_last_call_launches(mi355_jpeg_ctx *ctx, uint32_t *block_encode_launches);

/* How often the screened transform had to look twice, since the context was created
 * or the counters were last reset.  Waits for `stream`.  second_looks: wave-level groups (256
 * coefficients of 16 units) in which at least one coefficient was not decided by the fp32 first look and
 * all five digits were consulted; exact_units: units left undecided by both looks and recomputed with the
 * reference's ordered fp64 chain (the arbiter).  On noise exact_units / units is ~1e-7; a regression of
 * the accept thresholds to "accept everything" shows up as both counters stuck at zero on inputs built
 * to sit on rounding boundaries (tests/test_screen_pinning.py).
 * ABI 4 adds the two counters of the entropy walk (HuffmanEncoder, utils.cpp:656-698): rewalked_units: units whose AC
 * string was longer than its 24-word slot in on-chip memory and was therefore coded a second time straight into device
 * memory (noise at q >= 90: most luma units; q50: none); general_passes: (tile, channel) passes coded by the general
 * walk loop (values beyond +-31 or a table with holes) instead of the branch-free one. */
typedef struct mi355_jpeg_screen_counts {
    uint64_t second_looks;
    uint64_t exact_units;
    uint64_t rewalked_units;
    uint64_t general_passes;
} mi355_jpeg_screen_counts;
int mi355_jpeg_screen_stats(mi355_jpeg_ctx *ctx, void *stream, mi355_jpeg_screen_counts *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* MI355_JPEG_H */
