// Internal interface between the C ABI (mi355_jpeg.cpp) and the HIP kernels
// (jpeg_kernels.hip).  Not installed; the public boundary is include/mi355_jpeg.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

struct Geom {
    uint32_t W, H;          // image size as given
    uint32_t W8, H8;        // padded to multiples of 8 (getNearest8x8ImageSize, utils.cpp:184-187)
    uint32_t nbx;           // blocks per row = W8/8
    uint32_t N;             // blocks per frame
    uint32_t tiles;         // ceil(N/64)
    uint32_t flags;         // MI355_F_*
    uint32_t fast_rows;     // 1: W % 8 == 0 and the frame base is 8-byte aligned
    uint32_t passes;        // wave passes per tile: 3 (one per channel); 6 in 4:2:0 standard mode
                            // (four luma quarter-tiles of 16 MCUs, then Cb, then Cr)
    uint32_t nmx;           // 4:2:0 only: MCUs (16x16) per row = W8/16; there N = MCUs per frame,
                            // W8/H8 are multiples of 16 and a tile is 64 MCUs = 384 units of the scan
    uint64_t frame_stride;  // bytes between frames = W*H*3
};
inline bool is420(const Geom& g) { return g.passes == 6; }
// Per-frame verdicts in the bit-count array of a call (include/mi355_jpeg.h: MI355_BITS_*): no output for the frame.
constexpr uint64_t kBitsCapacity = ~0ull;  // the scan does not fit the frame's output slot
constexpr uint64_t kBitsCategory = ~1ull;  // a coefficient without a code in the tables
constexpr uint64_t kBitsFlagged = ~1ull;   // >= this: one of the above
// units (8x8 blocks in the scan) per frame
inline size_t unit_count(const Geom& g) { return (size_t)g.N * g.passes; }

// Sizes of the device workspace per frame, in elements.
inline size_t coef_dwords(const Geom& g) { return (size_t)g.tiles * g.passes * 2048; }
inline size_t unit_off_words(const Geom& g) { return (size_t)g.tiles * 192; }

// Device pointers of the screened (integer-MFMA) pipeline, jpeg_screen_kernels.hip.
struct ScreenParams {
    const uint4* afrag;     // [4 row tiles][5 digits][64 lanes] 16 int8: MFMA A fragments of the fixed-point map
    const uint4* csc_frag;  // [kCscSets standard-mode sets + kCscStrictSets strict-mode sets][64 lanes] 16 int8: MFMA A fragments of the colour conversion (jpeg_tables.h)
    const double* qconst;   // [2 channel types][64 zig-zag positions][4] = {s1, thr1, s2, thr2}
    const float* qconst_f;  // [2][16 groups of 4 positions][8] = {2^-23/Q x4, first-look threshold x4}
    const double* qd;       // [2][64] quantiser divisors as doubles, natural order
    const uint32_t* qnat_zz; // [2][64] quantiser divisors as integers, ZIG-ZAG order (standard mode's exact decision)
    const uint32_t* lut;    // [4][256] Huffman LUTs (code << 5 | len)
    const uint32_t* lut2;   // [2 channel types][66 rows: value + 32][16 columns: run] whole AC symbols, left-aligned | length
    uint32_t* meta;         // [frame][tile][pass][64] aclen << 16 | (uint16)dc
    uint32_t* pass_off;     // [frame][tile][pass] arena word offset of the pass's first string (the others follow in lane order)
    uint32_t* arena;        // AC bit strings, one word-aligned blob per unit
    uint32_t arena_words;   // = grid * region_words + overflow pool
    uint32_t region_words;  // private region of each persistent wave (bump-allocated without atomics)
    uint32_t overflow_base; // first word of the shared overflow pool (= grid * region_words)
    uint32_t* counters;     // [0] overflow-pool words used
    unsigned long long* stats;  // [0] second looks (wave-level groups), [1] units recomputed by the exact chain, [2] units walked twice, [3] general-loop passes
    uint32_t prio_from_wg;  // k_screen_encode: workgroups >= this raise their issue priority (0xFFFFFFFF: none)
    uint32_t stagger;       // k_screen_encode: those workgroups start this many s_sleep(127) (~3.4 us each) late (0: none)
    uint32_t* status;
    uint32_t* tile_bits;    // [frame][tile] bit totals, accumulated with atomics (zero on entry); bit 31: a unit of the tile had an error
    uint32_t* coefs;        // probe output (tiled coefficient layout) or nullptr
    uint8_t* samples;       // probe output (padded YCbCr image, interleaved) or nullptr
    unsigned long long* stamps;  // diagnostic build only (-DMI355_STAMPS): [wave][8] phase cycle sums
};

// number of persistent single-wave workgroups launch_screen_encode will use
uint32_t screen_grid(const Geom& g, uint32_t n_frames, uint32_t max_waves);
hipError_t launch_screen_encode(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const ScreenParams& sp,
                                bool probe, uint32_t grid_waves, hipStream_t s);
hipError_t launch_dc_heads(const Geom& g, uint32_t n_frames, const ScreenParams& sp, hipStream_t s);
hipError_t launch_merge(const Geom& g, uint32_t n_frames, const uint32_t* meta, const uint32_t* pass_off, const uint32_t* arena,
                        const uint32_t* lut, const uint64_t* tile_off,
                        uint8_t* out, uint64_t out_stride, const uint64_t* frame_bits /* ~0: the frame is skipped */,
                        uint32_t lds_words_limit, bool small_window /* two workgroups per CU beside the block encode */, hipStream_t s);

// auxiliary kernels (jpeg_aux_kernels.hip)
hipError_t launch_lcg_fill(uint8_t* dst, uint64_t frame_bytes, uint32_t n_frames, uint32_t seed0, hipStream_t s);
// stuffs ceil(*d_nbits/8) bytes (<= max_bytes) of `in` into `out`; *d_total = stuffed length in bytes
hipError_t launch_stuff(const uint8_t* in, const uint64_t* d_nbits, uint64_t max_bytes, uint32_t* counts,
                        uint64_t* offs, uint64_t* d_total, uint8_t* out, uint64_t cap, uint32_t* status,
                        const uint64_t* tile_off /* restart intervals: [tiles] byte-aligned bit offsets, or nullptr */,
                        uint32_t tiles, hipStream_t s);

hipError_t launch_transform(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const double* qd,
                            uint32_t* coefs, int mode, hipStream_t s);
hipError_t launch_probe_samples(const Geom& g, const uint8_t* rgb, uint8_t* samples, hipStream_t s);
hipError_t launch_unit_sizes(const Geom& g, uint32_t n_frames, const uint32_t* coefs,
                             const uint32_t* lut, uint32_t* unit_off, uint32_t* tile_bits,
                             uint32_t* status, hipStream_t s);
hipError_t launch_tile_scan(const Geom& g, uint32_t n_frames, uint32_t* tile_bits,
                            uint64_t* tile_off, uint8_t* out, uint64_t out_stride,
                            uint64_t* frame_bits, uint32_t* status, uint32_t* reset_counters,
                            bool rearm_tiles, uint64_t* chunk_tot,
                            bool flag_frames /* a frame over capacity or with a poisoned tile total (bit 31) gets bits = ~0 */,
                            hipStream_t s);
// Frames above 8192 tiles are scanned by scan_chunks(g) workgroups each (4096 tiles per chunk); their totals need
// scan_chunks(g) words of scratch per frame (`chunk_tot`; nullptr: one workgroup per frame).
inline uint32_t scan_chunks(const Geom& g) { return g.tiles > 8192 ? (g.tiles + 4095) / 4096 : 0; }
// entries of the tile-offset workspace of a batch: [frame][tiles + 1] offsets, then the scan scratch
inline size_t tile_off_entries(const Geom& g, uint32_t n_frames) {
    return ((size_t)g.tiles + 1 + scan_chunks(g)) * n_frames;
}
hipError_t launch_emit(const Geom& g, uint32_t n_frames, const uint32_t* coefs, const uint32_t* lut,
                       const uint32_t* unit_off, const uint64_t* tile_off, uint8_t* out,
                       uint64_t out_stride, const uint32_t* status, uint32_t lds_words_limit,
                       hipStream_t s);
hipError_t launch_coefs_to_rows(const Geom& g, const uint32_t* coefs, int16_t* rows, hipStream_t s);
hipError_t launch_rows_to_coefs(const Geom& g, const int16_t* rows, uint32_t* coefs, hipStream_t s);
hipError_t launch_unit_bits(const Geom& g, const uint32_t* unit_off, const uint32_t* tile_bits,
                            uint32_t* out, hipStream_t s);

// stage-by-stage kernels (jpeg_stage_kernels.hip): one per stage function of the reference's interface
hipError_t launch_stage_csc(uint8_t* img, uint64_t n_px, hipStream_t s);
hipError_t launch_stage_cds(uint8_t* img, uint32_t W, uint32_t H, hipStream_t s);
hipError_t launch_stage_copy_larger(const uint8_t* src, uint32_t W, uint32_t H, uint8_t* dst, uint32_t W8, hipStream_t s);
hipError_t launch_stage_mirror(uint8_t* img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH, hipStream_t s);
hipError_t launch_stage_u8_to_f64(const uint8_t* src, double* dst, uint64_t n, hipStream_t s);
hipError_t launch_stage_sub(double* img, uint64_t n, double val, hipStream_t s);
hipError_t launch_stage_dct(double* img, uint32_t W8, uint32_t H8, hipStream_t s);
hipError_t launch_stage_quant(double* img, uint32_t W8, uint32_t H8, const double* q, hipStream_t s);
hipError_t launch_stage_blocks(const double* img, uint32_t W8, uint32_t H8, int* lin, hipStream_t s);
hipError_t launch_stage_zigzag(const int* lin, int* zz, uint64_t rows, hipStream_t s);
hipError_t launch_stage_rle(const int* zz, uint64_t rows, int* pairs, uint32_t* counts, hipStream_t s);
hipError_t launch_stage_huffman(const int* zz, const int* pairs, const uint32_t* counts, uint64_t N, const uint32_t* lut,
                                uint32_t* ubits, uint32_t* inchunk, uint64_t* chunk_sum, uint64_t* total, uint32_t* outw,
                                uint64_t cap_bits, uint32_t* status, hipStream_t s);

}  // namespace mi355
