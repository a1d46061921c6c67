// Device-side building blocks of the screened (integer-MFMA) pipeline (jpeg_screen_kernels.hip).  See that
// file's header comment for why the screen is bit-exact.
#pragma once
#include "jpeg_devfn.h"
#include "jpeg_screen_tables.h"  // kScreenLimbs, kScreenFracBits (the tables themselves are uploaded by the host)

namespace mi355 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// "does any active lane say yes": the ballot's SGPR pair compared on the scalar unit (HIP's __any goes through a
// v_cndmask + v_cmp pair first)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// the zig-zag rows are written as packed uint32 pairs and read back as int16: tell TBAA
typedef int16_t __attribute__((may_alias)) i16a;

#ifndef MI355_WALK_UNROLL2
#define MI355_WALK_UNROLL2 1
#endif

constexpr uint32_t kSlotWordsFull = 54;  // worst case 63*(17+10)+4 = 1705 bits
#ifndef MI355_SLOT_ROWS
#define MI355_SLOT_ROWS 24
#endif

// Zig-zag rows of a wave's 64 units in LDS: int16, laid out [position 0..64][unit].  Unit u's
// coefficient at a position sits at int16 index pos * 64 + row_unit_off(u): units 0..31 use the low
// halves of 32 consecutive dwords, units 32..63 the high halves.  The entropy walk reads "my unit's
// coefficient at MY next non-zero position" -- 64 data-dependent positions per wave instruction --
// and with this layout the 32 lanes of each LDS lane group always hit 32 different banks (the
// stride-33 unit-major rows of round 1 collided at random: 38 % of the LDS-active cycles were bank
// conflicts).  Position 64 is a sentinel row holding kRowSentinel (read by lanes that have run out of symbols).
constexpr uint32_t kRowWords = 65 * 32;  // dwords per wave
__device__ __forceinline__ uint32_t row_unit_off(uint32_t u) { return ((u & 31u) << 1) | (u >> 5); }

// ----------------------------------------------------------------------------
// integer-exact colour conversion (performCSC, utils.cpp:92-110)
//
// Y  = (uint8)(0.299 R + 0.587 G + 0.114 B) evaluated in fp64 differs from the
// decimal value (299R+587G+114B)/1000 by < 1e-13, so its truncation equals the
// integer quotient unless the decimal value is itself an integer (remainder 0),
// where the fp64 sum may land just below it: that case (1 pixel in 1000) is
// evaluated in fp64.  For Cb/Cr, (c0 R + c1 G + c2 B)/1e6 + 128, the integer
// quotient is exact for all 2^24 inputs.  Both statements are checked
// exhaustively (tests: exhaustive colour conversion on the GPU path).
// ----------------------------------------------------------------------------
// STD = standard mode (not a behaviour of the reference): round to nearest
// instead of truncating, clamp to 255; pure integer arithmetic.
// (a * m) >> 32 for a, m < 2^24 on the full-rate 24-bit multiplier (v_mul_hi_u32 is quarter rate,
// and the compiler cannot see the operand ranges behind the dot product).
__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t m) {
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(m), "v"(a));
    return r;
}
// s / 1000 for s <= 255500 and s / 31250 for s < 8.1e6 (both checked exhaustively,
// tests/test_standard_mode.py::test_division_constants)
__device__ __forceinline__ uint32_t div1000(uint32_t s) { return mulhi24(s, 8589935u) >> 1; }
__device__ __forceinline__ uint32_t div31250(uint32_t s) { return mulhi24(s, 8796094u) >> 6; }

template <bool STD>
__device__ __forceinline__ uint32_t csc_int(int chan, uint32_t r, uint32_t g, uint32_t b) {
    if constexpr (STD) {  // 15-bit fixed point (jpeg_tables.h); + 128 << 15 keeps the chroma numerator positive
        const int s = kStdCsc[chan][0] * (int)r + kStdCsc[chan][1] * (int)g + kStdCsc[chan][2] * (int)b;
        return (uint32_t)(s + (chan == 0 ? 16384 : 16383 + (128 << 15))) >> 15;
    }
    if (chan == 0) {
        uint32_t s = 299u * r + 587u * g + 114u * b;  // <= 255000
        uint32_t y = div1000(s);
        if (s == __umul24(y, 1000u)) y = csc1(r, g, b, 0.299, 0.587, 0.114, 0.0);
        return y;
    } else {
        // numerators divided by 32 (exact: every constant and 128e6 are multiples of 32)
        const uint32_t s = chan == 1 ? 4000000u + 15625u * b - 5273u * r - 10352u * g
                                     : 4000000u + 15625u * r - 13084u * g - 2541u * b;
        return div31250(s);
    }
}

template <bool STD>
__device__ __forceinline__ uint32_t csc_int_at(const uint8_t* __restrict__ f, uint32_t W, uint32_t x,
                                               uint32_t y, int chan) {
    const uint8_t* p = f + ((size_t)y * W + x) * 3;
    return csc_int<STD>(chan, p[0], p[1], p[2]);
}

// padded pixel (px,py) of channel chan, generic path (see sample_generic)
template <bool STD>
__device__ __forceinline__ uint32_t sample_generic_int(const uint8_t* __restrict__ f, const Geom& g,
                                                       int chan, bool avg, uint32_t px, uint32_t py) {
    uint32_t mx = px < g.W ? px : 2 * g.W - 1 - px;
    uint32_t my = py < g.H ? py : 2 * g.H - 1 - py;
    if (avg) {
        uint32_t qx = mx & ~1u, qy = my & ~1u;
        if (qx + 1 < g.W && qy + 1 < g.H) {
            uint32_t s = csc_int_at<STD>(f, g.W, qx, qy, chan) + csc_int_at<STD>(f, g.W, qx + 1, qy, chan) +
                         csc_int_at<STD>(f, g.W, qx, qy + 1, chan) + csc_int_at<STD>(f, g.W, qx + 1, qy + 1, chan);
            return s >> 2;
        }
    }
    return csc_int_at<STD>(f, g.W, mx, my, chan);
}

// Raw RGB of rows 2*gq, 2*gq+1 of block (bx,by): 2 x 24 bytes as six 8-byte loads (fast
// path: the block lies inside the image and rows are 8-byte aligned).
__device__ __forceinline__ void load_raw_rowpair(const uint8_t* __restrict__ f, const Geom& g, uint32_t bx,
                                                 uint32_t by, uint32_t gq, uint32_t (&w)[12]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        // 32-bit byte offset from the (wave-uniform) frame base: the loads take the base from SGPRs and the address
        // costs two 32-bit instructions instead of a chain of quarter-rate v_mad_u64_u32 (fast_rows guarantees
        // W * H * 3 < 2^32)
        const uint32_t off = (__umul24(by * 8 + gq * 2 + r, g.W) + bx * 8) * 3u;  // rows and widths are below 2^24
        const uint2* p = reinterpret_cast<const uint2*>(f + off);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint2 v = p[j];
            w[r * 6 + 2 * j] = v.x;
            w[r * 6 + 2 * j + 1] = v.y;
        }
    }
}

// The same conversion on a pixel fetched as rg = R | G << 16 (one v_perm_b32 on the raw dwords) and b:
// the R and G terms are one v_dot2 (16-bit lanes), so four instructions form the weighted sum
// instead of six.  The chroma numerators are divisible by 32 for every input (all six constants
// and 128e6 are), so floor(x / 1e6) == floor((x / 32) / 31250) with constants that fit 16 bits:
// 168736/32 = 5273, 331264/32 = 10352, 500000/32 = 15625, 418688/32 = 13084, 81312/32 = 2541.
// Same results as csc_int for all 2^24 inputs (the exhaustive colour conversion tests run this one).
template <bool STD>
__device__ __forceinline__ uint32_t csc_packed(int chan, uint32_t rg, uint32_t b) {
    typedef short v2s __attribute__((ext_vector_type(2)));
    const v2s RG = __builtin_bit_cast(v2s, rg);
    if constexpr (STD) {  // the same integers as csc_int<true>
        const v2s K = chan == 0 ? v2s{(short)kStdCsc[0][0], (short)kStdCsc[0][1]}
                                : (chan == 1 ? v2s{(short)kStdCsc[1][0], (short)kStdCsc[1][1]} : v2s{(short)kStdCsc[2][0], (short)kStdCsc[2][1]});
        const int kb = chan == 0 ? kStdCsc[0][2] : (chan == 1 ? kStdCsc[1][2] : kStdCsc[2][2]);
        const int s = __builtin_amdgcn_sdot2(RG, K, kb * (int)b + (chan == 0 ? 16384 : 16383 + (128 << 15)), false);
        return (uint32_t)s >> 15;
    }
    if (chan == 0) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_sdot2(RG, v2s{299, 587}, (int)(114u * b), false);
        uint32_t y = div1000(s);
        if (s == __umul24(y, 1000u)) y = csc1(rg & 0xffffu, rg >> 16, b, 0.299, 0.587, 0.114, 0.0);
        return y;
    } else {
        const int kb = chan == 1 ? 15625 : -2541;
        const v2s K = chan == 1 ? v2s{-5273, -10352} : v2s{15625, -13084};
        const uint32_t s = (uint32_t)__builtin_amdgcn_sdot2(RG, K, 4000000 + kb * (int)b, false);  // (128e6 + ...) / 32
        return div31250(s);
    }
}

// 16 samples of channel CHAN from the raw row pair, packed 4 per dword in sample order
// (y*8+x), as unsigned bytes; chroma averaging over the 2x2 quads of the row pair.
// Strict luma: the integer quotient is the reference's value unless the decimal value is an integer (remainder 0, one
// pixel in a thousand), where the fp64 sum has to be evaluated.  Testing every pixel with an exec-masked branch cost
// 7 % of the kernel (the branch machinery, not the rare fp64 code): so four pixels are converted branch-free with their
// remainders (one v_mad_i32_i24 each), and ONE wave-uniform test per four pixels (taken 23 % of the time) guards the
// per-pixel fix-ups.
template <int CHAN, bool STD>
__device__ __forceinline__ void convert_rowpair(const uint32_t (&w)[12], bool avg, uint32_t (&pk)[4]) {
    uint32_t val[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t rgs[4], bs[4], rem[4];
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const int x = 4 * h + xx;
                // pixel x = bytes 3x, 3x+1, 3x+2 of the row: R and G through a byte permute of the two
                // dwords around them (selector 0..3 = low dword, 4..7 = high dword, 0x0c = zero)
                constexpr uint32_t kZ = 0x0cu;
                const int o = 3 * x, i = o >> 2, k = o & 3;
                const uint32_t lo = w[r * 6 + i], hi = w[r * 6 + (i < 5 ? i + 1 : i)];
                const uint32_t rg = __builtin_amdgcn_perm(hi, lo, (uint32_t)k | (kZ << 8) | ((uint32_t)(k + 1) << 16) | (kZ << 24));
                const uint32_t b = (w[r * 6 + ((o + 2) >> 2)] >> (8 * ((o + 2) & 3))) & 255u;
                if constexpr (CHAN == 0 && !STD) {
                    typedef short v2s __attribute__((ext_vector_type(2)));
                    const uint32_t s = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, rg), v2s{299, 587}, (int)(114u * b), false);
                    const uint32_t y = div1000(s);
                    val[r][x] = y;
                    rgs[xx] = rg, bs[xx] = b;
                    // s - 1000 y as ONE 24-bit multiply-add (left to itself the compiler forms it with a quarter-rate
                    // v_mad_u64_u32 and a v_bfe in front)
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(rem[xx]) : "v"(y), "s"(-1000), "v"(s));
                } else {
                    val[r][x] = csc_packed<STD>(CHAN, rg, b);
                }
            }
            if constexpr (CHAN == 0 && !STD) {
                const uint32_t m01 = rem[0] < rem[1] ? rem[0] : rem[1], m23 = rem[2] < rem[3] ? rem[2] : rem[3];
                if (wave_any((m01 < m23 ? m01 : m23) == 0u)) {
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx)
                        if (rem[xx] == 0u) val[r][4 * h + xx] = csc1(rgs[xx] & 0xffffu, rgs[xx] >> 16, bs[xx], 0.299, 0.587, 0.114, 0.0);
                }
            }
        }
    if (avg) {
        // every 2x2 quad becomes its truncated mean: both rows are m0 m0 m1 m1 | m2 m2 m3 m3 (one byte permute each)
        uint32_t m[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) m[x] = (val[0][2 * x] + val[0][2 * x + 1] + val[1][2 * x] + val[1][2 * x + 1]) >> 2;
        pk[0] = __builtin_amdgcn_perm(m[1], m[0], 0x04040000u);
        pk[1] = __builtin_amdgcn_perm(m[3], m[2], 0x04040000u);
        pk[2] = pk[0];
        pk[3] = pk[1];
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                pk[r * 2 + h] = val[r][4 * h] | (val[r][4 * h + 1] << 8) | (val[r][4 * h + 2] << 16) |
                                (val[r][4 * h + 3] << 24);
    }
}

// Edge / unaligned tiles: one sample at a time with mirroring.
template <int CHAN, bool STD>
__device__ __forceinline__ void generic_rowpair(const uint8_t* __restrict__ f, const Geom& g, bool avg,
                                                uint32_t bx, uint32_t by, uint32_t gq, uint32_t (&pk)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = 0;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            int s = i * 4 + j;  // 0..15 within the row pair
            uint32_t smp = sample_generic_int<STD>(f, g, CHAN, avg, bx * 8 + (s & 7), by * 8 + gq * 2 + (s >> 3));
            v |= smp << (8 * j);
        }
        pk[i] = v;
    }
}

// ---- 4:2:0 standard mode: one chroma sample = the linear form box-filtered over the 2x2 quad, rounded once ----
// (jpeg_tables.h: kStdCsc420).
// Raw RGB of pixel rows row0, row0+1 of MCU (mx,my): 2 x 48 bytes as twelve 8-byte loads (fast
// path: every MCU of the wave interior, W % 8 == 0, base 8-aligned).
__device__ __forceinline__ void load_raw_mcu_rows(const uint8_t* __restrict__ f, const Geom& g, uint32_t mx,
                                                  uint32_t my, uint32_t row0, uint32_t (&w)[24]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t off = (__umul24(my * 16 + row0 + r, g.W) + mx * 16) * 3u;  // see load_raw_rowpair
        const uint2* p = reinterpret_cast<const uint2*>(f + off);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            uint2 v = p[i];
            w[r * 12 + 2 * i] = v.x;
            w[r * 12 + 2 * i + 1] = v.y;
        }
    }
}
__device__ __forceinline__ int std_lin420(int cc, uint32_t r, uint32_t g, uint32_t b) {
    return kStdCsc420[cc][0] * (int)r + kStdCsc420[cc][1] * (int)g + kStdCsc420[cc][2] * (int)b;
}
// One row of 8 chroma samples of channel CHAN from two pixel rows of 16, packed 4 per dword (unsigned bytes).
template <int CHAN>
__device__ __forceinline__ void convert_chroma420_row(const uint32_t (&w)[24], uint32_t (&pk2)[2]) {
    typedef short v2s __attribute__((ext_vector_type(2)));
    uint32_t m[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        int sum = 32767 + (128 << 16);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr uint32_t kZ = 0x0cu;
                const int o = 3 * (2 * x + h), i = o >> 2, k = o & 3;
                const uint32_t lo = w[r * 12 + i], hi = w[r * 12 + (i < 11 ? i + 1 : i)];
                const uint32_t rg = __builtin_amdgcn_perm(hi, lo, (uint32_t)k | (kZ << 8) | ((uint32_t)(k + 1) << 16) | (kZ << 24));
                const uint32_t b = (w[r * 12 + ((o + 2) >> 2)] >> (8 * ((o + 2) & 3))) & 255u;
                sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, rg), v2s{(short)kStdCsc420[CHAN - 1][0], (short)kStdCsc420[CHAN - 1][1]},
                                             kStdCsc420[CHAN - 1][2] * (int)b + sum, false);
            }
        m[x] = (uint32_t)sum >> 16;
    }
    pk2[0] = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
    pk2[1] = m[4] | (m[5] << 8) | (m[6] << 16) | (m[7] << 24);
}
// Edge MCUs: one chroma sample at a time, every contributing pixel mirrored on its own (the checker pads the RGB image,
// then filters).
__device__ __forceinline__ void generic_chroma420(const uint8_t* __restrict__ f, const Geom& g, int chan,
                                                  uint32_t mx, uint32_t my, uint32_t gq, uint32_t (&pk)[4]) {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        uint32_t v = 0;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int sidx = i * 4 + j;  // 0..15 within the chroma row pair 2gq, 2gq+1
            const uint32_t px = mx * 16 + 2 * (sidx & 7), py = my * 16 + 2 * (gq * 2 + (sidx >> 3));
            int sum = 32767 + (128 << 16);
#pragma unroll 1
            for (int d = 0; d < 4; ++d) {
                const uint32_t x = px + (d & 1), y = py + (d >> 1);
                const uint32_t sx = x < g.W ? x : 2 * g.W - 1 - x, sy = y < g.H ? y : 2 * g.H - 1 - y;
                const uint8_t* p = f + ((size_t)sy * g.W + sx) * 3;
                sum += std_lin420(chan - 1, p[0], p[1], p[2]);
            }
            v |= ((uint32_t)sum >> 16) << (8 * j);
        }
        pk[i] = v;
    }
}

// ---- standard 4:4:4: the colour conversion on the matrix units (whole tiles inside the image) ----
// B operand = raw RGB bytes ^ 0x80 (x - 128 as int8), 16 consecutive bytes per lane; A = the block-diagonal fragment sets
// of jpeg_tables.h / mi355_jpeg.cpp (upload_csc_frag): lane (n, gq) receives four outputs computed from its own bytes.
// Coefficients are two balanced base-256 digits, so an output is (acc1 << 8) + acc0, exact in int32.  The rounding constant
// travels in the accumulator input; because every coefficient row sums to a power of two (luma) or to zero (chroma), x - 128
// in place of x changes nothing but the level shift the transform wants anyway: the result byte IS sample - 128.
struct __attribute__((packed, aligned(8))) RawChunk {
    uint32_t w[4];
};
__device__ __forceinline__ v4i chunk_b(const RawChunk& c) {
    return v4i{(int)(c.w[0] ^ 0x80808080u), (int)(c.w[1] ^ 0x80808080u), (int)(c.w[2] ^ 0x80808080u), (int)(c.w[3] ^ 0x80808080u)};
}
__device__ __forceinline__ v4i splat4(int v) { return v4i{v, v, v, v}; }
// byte 2 of each of four int32 -> one dword, in order
__device__ __forceinline__ uint32_t pack_byte2(const v4i& s) {
    const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)s[1], (uint32_t)s[0], 0x0c0c0602u);
    const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)s[3], (uint32_t)s[2], 0x06020c0cu);
    return p01 | p23;
}
__device__ __forceinline__ void load_csc_fragments(const ScreenParams& sp, uint32_t lane, int first_set, v4i (&F)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 t = sp.csc_frag[(first_set + i) * 64 + lane];
        F[i] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
    }
}
// Rows 2gq, 2gq+1 of block (bx, by) as four chunks: [2r + h] = bytes 8h .. 8h+15 of row r (the two chunks of a row overlap
// by 8 bytes; the fragments of half 1 ignore the first four bytes of theirs).
__device__ __forceinline__ void load_std_rowpair(const uint8_t* __restrict__ f, const Geom& g, uint32_t bx, uint32_t by,
                                                 uint32_t gq, RawChunk (&X)[4]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t off = (__umul24(by * 8 + gq * 2 + r, g.W) + bx * 8) * 3u;  // see load_raw_rowpair
        X[2 * r] = *reinterpret_cast<const RawChunk*>(f + off);
        X[2 * r + 1] = *reinterpret_cast<const RawChunk*>(f + off + 8);
    }
}
// 16 samples - 128 of one channel (the fragments say which) from the row pair, packed 4 per dword in sample order.
// Scale: coefficients x 2^15; the low digit is accumulated TWICE and the high one shifted by 9, which puts the binary point
// at bit 16 -- the sample is byte 2 -- for two idle matrix instructions instead of sixteen shifts.
// F = [half][digit].  The rounding constant -- 32768 for luma (half), 32766 for chroma (2 * (half - 1)) -- enters as
// INLINE constants of the accumulator inputs (no registers): 64 << 9 through the high digit, 0 or -2 through the low one.
template <bool CHROMA>
__device__ __forceinline__ void std_rowpair_mfma(const RawChunk (&X)[4], const v4i (&F)[4], uint32_t (&pk)[4]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v4i b = chunk_b(X[2 * r + h]);
            v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[2 * h], b, splat4(CHROMA ? -2 : 0), 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[2 * h], b, a0, 0, 0, 0);
            const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[2 * h + 1], b, splat4(64), 0, 0, 0);
            v4i sv;
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = (a1[i] << 9) + a0[i];
            pk[2 * r + h] = pack_byte2(sv);
        }
}
// ---- strict mode: the reference's conversion on the matrix units (whole tiles inside the image) ----
// The integer numerators of csc_int -- 299 R + 587 G + 114 B, and (128e6 + ...) / 32 for chroma -- formed exactly by two
// matrix instructions per four pixels (two balanced base-256 digits of the coefficients, fragment sets kCscSets.. of
// jpeg_tables.h) instead of four vector instructions per pixel (byte permute, byte extract, multiply-add, dot product);
// the division, the remainder test of luma and its rare fp64 evaluation are those of convert_rowpair.  w = the raw row
// pair (2 x 24 bytes); each half of a row is one operand {three dwords = four pixels, a fourth the fragments ignore}: the
// fourth is left undefined on purpose, so that the register allocator can take whatever lies behind the three (a defined
// filler cost eleven moves per row pair).  The operand bytes are x - 128 = x ^ 0x80; the constant term, and what the - 128
// takes away, enter through the accumulator input of the low digit (F = {digit 0, digit 1}, cst = kCscStrictC splat).
template <int CHAN>
__device__ __forceinline__ void strict_rowpair_mfma(const uint32_t (&w)[12], const v4i (&F)[2], const v4i& cst, bool avg, uint32_t (&pk)[4]) {
    uint32_t val[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int dontcare;
            asm volatile("" : "=v"(dontcare));  // no instruction: an undefined register (volatile: one per operand, not one shared by copies)
            const v4i b = v4i{(int)(w[r * 6 + 3 * h] ^ 0x80808080u), (int)(w[r * 6 + 3 * h + 1] ^ 0x80808080u),
                              (int)(w[r * 6 + 3 * h + 2] ^ 0x80808080u), dontcare};
            const v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[0], b, cst, 0, 0, 0);
            const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[1], b, v4i{0, 0, 0, 0}, 0, 0, 0);
            uint32_t rem[4];
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const uint32_t s = (uint32_t)((a1[xx] << 8) + a0[xx]);
                if constexpr (CHAN == 0) {
                    const uint32_t y = div1000(s);
                    val[r][4 * h + xx] = y;
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(rem[xx]) : "v"(y), "s"(-1000), "v"(s));
                } else {
                    val[r][4 * h + xx] = div31250(s);
                }
            }
            if constexpr (CHAN == 0) {
                const uint32_t m01 = rem[0] < rem[1] ? rem[0] : rem[1], m23 = rem[2] < rem[3] ? rem[2] : rem[3];
                if (wave_any((m01 < m23 ? m01 : m23) == 0u)) {
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx)
                        if (rem[xx] == 0u) {
                            const int o = 3 * (4 * h + xx);
                            const uint32_t R = (w[r * 6 + (o >> 2)] >> (8 * (o & 3))) & 255u, G = (w[r * 6 + ((o + 1) >> 2)] >> (8 * ((o + 1) & 3))) & 255u,
                                           B = (w[r * 6 + ((o + 2) >> 2)] >> (8 * ((o + 2) & 3))) & 255u;
                            val[r][4 * h + xx] = csc1(R, G, B, 0.299, 0.587, 0.114, 0.0);
                        }
                }
            }
        }
    if (avg) {
        uint32_t m[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) m[x] = (val[0][2 * x] + val[0][2 * x + 1] + val[1][2 * x] + val[1][2 * x + 1]) >> 2;
        pk[0] = __builtin_amdgcn_perm(m[1], m[0], 0x04040000u);
        pk[1] = __builtin_amdgcn_perm(m[3], m[2], 0x04040000u);
        pk[2] = pk[0];
        pk[3] = pk[1];
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                pk[r * 2 + h] = val[r][4 * h] | (val[r][4 * h + 1] << 8) | (val[r][4 * h + 2] << 16) | (val[r][4 * h + 3] << 24);
    }
}

// ----------------------------------------------------------------------------
// device-side parameter block
// ----------------------------------------------------------------------------
// (ScreenParams is declared in jpeg_device.h)

__device__ __forceinline__ int meta_dc(uint32_t w) { return (int)(int16_t)(w & 0xffffu); }

// Arena allocation for one wave.  Every persistent wave owns a private region and bumps a
// private pointer (no atomics: a returning atomic on one address saturates at ~88 per
// microsecond chip-wide, far below the wave-tile rate).  A wave whose region is full takes
// chunks from the shared overflow pool with one atomic per chunk.
constexpr uint32_t kOverflowChunk = 1024;  // words
struct WaveArena {
    uint32_t ptr, left;
    // returns the base word of `need` words, or 0xFFFFFFFF if the arena is exhausted
    __device__ __forceinline__ uint32_t take(const ScreenParams& sp, uint32_t need, uint32_t lane) {
        if (need > left) {
            uint32_t grab = need > kOverflowChunk ? need : kOverflowChunk;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(sp.counters, grab);
            base = __builtin_amdgcn_readfirstlane(base);
            if ((uint64_t)sp.overflow_base + base + grab > sp.arena_words) return 0xFFFFFFFFu;
            ptr = sp.overflow_base + base;
            left = grab;
        }
        uint32_t r = ptr;
        ptr += need;
        left -= need;
        return r;
    }
};

// ----------------------------------------------------------------------------
// Unit walk as a loop over NON-ZERO coefficients (the work is proportional to what
// actually gets coded).  The zig-zag row of the unit lives in LDS; a 64-bit
// non-zero mask built during quantisation gives the positions (lowest set bit =
// next coefficient, run = gap to the previous one).  A combined (run, value) LUT
// returns the complete symbol -- Huffman code and value bits, left-aligned, length
// in the low five bits -- for |value| <= 31; larger values assemble it from the
// (run,size) table.  Bits are packed with 32-bit funnel shifts.  LDS reads are
// software-pipelined two symbols ahead.  Same bits as walk_ac().
// ----------------------------------------------------------------------------
// The kernel's LDS is kept at 72,576 B per workgroup (24 slot rows, no fp64 threshold table), so that next to the two
// resident workgroups of a CU one workgroup of k_merge with the full window (17,280 B) or two with the half window that
// batches run (2 x 8,896 B), or k_dc_heads, still fit: another stream's tail kernels then run under this kernel instead of
// waiting for it (tests/test_kernel_budget.py checks the budget on the built library).
#ifndef MI355_SLOT_ROWS
#define MI355_SLOT_ROWS 24
#endif
constexpr uint32_t kSlotRows = MI355_SLOT_ROWS;  // words per unit in the LDS slot; larger strings re-walk into global memory
// Symbol table layout [value + 32][run] (index (v + 32) * 16 + run), values clamped to [-32, 32]: 65 rows of 16
// entries, and a 66th row.  The rows of -32 and +32 are kLut2Miss in every column ("no whole-symbol entry": larger
// values are clamped onto them and take the slow path).  The value-0 row holds ZRL in column 15 and nothing else: the
// walk visits a zero coefficient only at the positions mark_zero_runs adds to its mask -- always with run 15 -- so ZRL
// needs no marker value, no compare and no select.  A lane that has run out of symbols reads the sentinel position 64
// of its row buffer, which holds kRowSentinel = 29, with "run" kRunNone = 64: index (29 + 32) * 16 + 64 = entry 0 of
// the 66th row, which is zero -- a no-op put.  EOB is entry 1 of the 66th row.
// Run-minor on purpose: an LDS bank is (16 (v & 1) + run) mod 32, so lanes that code the same value (on noise 70 %
// of the symbols are +-1) after different runs read different banks.  The run-major layout of round 1 put every
// lane with the same value on ONE bank at up to 16 different addresses: that was most of the kernel's LDS bank
// conflicts.  (A 17-column layout with a zero column for the exhausted lanes was 528 bytes larger: k_merge's
// workgroup no longer fitted next to two of the block-encode kernel's on a CU, and batched calls lost 20 %.)
constexpr uint32_t kLut2Cols = 16, kLut2Rows = 66, kLut2Words = kLut2Rows * kLut2Cols;  // per channel type
constexpr uint32_t kLut2Zrl = 32 * kLut2Cols + 15;
constexpr uint32_t kLut2Eob = 65 * kLut2Cols + 1;
constexpr uint32_t kRunNone = 64;      // "run" of a lane without symbols left (v_ffbl_b32 of 0, clamped)
constexpr uint32_t kRowSentinel = 29;  // value at position 64 of every unit's row
static_assert((kRowSentinel + 32) * kLut2Cols + kRunNone == 65 * kLut2Cols, "exhausted lanes must land on the zero entry of the 66th row");
// "No whole-symbol entry" (a value outside [-31, 31], or a hole in the caller's Huffman table): length field 31, which
// no symbol has (<= 27).  The branch-free walk puts it like any entry -- the pass is walked again anyway -- and
// notices through the running maximum of the length fields: one instruction per symbol.
constexpr uint32_t kLut2Miss = 31u;

// Left-aligned 32-bit bit packer.  e = symbol bits left-aligned | length (<= 27) in bits 4..0.
// Every put stores the word being filled (a later put to the same word overwrites it).  State: the pending bits, the
// TOTAL number of bits put so far (the shifts only look at its low five bits, so the count within the word is never
// masked out), and the position of the word being filled in the store's own units (Store::at(word): an LDS byte
// address for StoreLds, a word index for StoreGlobal) -- one shift, one shift-add and one compare per symbol for the
// bookkeeping (round 3: add, compare, shift, and, add, and).
template <typename Store>
struct Packer32 {
    uint32_t acc = 0;  // pending bits, left-aligned
    uint32_t nt = 0;   // bits put so far
    uint32_t pos;      // position of the word being filled
    Store st;
    __device__ __forceinline__ explicit Packer32(Store s) : pos(s.at(0u)), st(s) {}
    __device__ __forceinline__ void put(uint32_t e) {
        const uint32_t ml = e & ~31u, t = e & 31u;
        const uint32_t hi = acc | (ml >> (nt & 31u));                      // (the shift instruction ignores the upper bits itself)
        const uint32_t lo = __builtin_amdgcn_alignbit(ml, 0u, nt & 31u);   // ml << (32 - n), and 0 when n == 0
        st(pos, hi);
        nt += t;
        const uint32_t pos2 = st.at(nt >> 5);
        acc = pos2 != pos ? lo : hi;
        pos = pos2;
    }
    __device__ __forceinline__ void finish() { st(pos, acc); }
    __device__ __forceinline__ void reset() { acc = 0, nt = 0, pos = st.at(0u); }
    __device__ __forceinline__ uint32_t bits() const { return nt; }
    __device__ __forceinline__ uint32_t words() const { return (nt + 31u) >> 5; }
};

typedef __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
struct StoreLds {  // [word][lane]: kSlotRows words per unit are kept; later words of an oversized string land in the dump row
    static constexpr uint32_t kStep = 256;  // LDS bytes from one word of a unit to the next (64 lanes)
    uint32_t base;                          // LDS byte address of this lane's column
    __device__ __forceinline__ explicit StoreLds(uint32_t* col) : base((uint32_t)(uintptr_t)(lds_u32_ptr)col) {}
    __device__ __forceinline__ uint32_t at(uint32_t word) const {
        uint32_t a;  // base + word * 256 as ONE shift-add (the compiler folds the caller's nt >> 5 into shift + mask + add)
        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a) : "v"(word), "v"(base));
        return a;
    }
    __device__ __forceinline__ void operator()(uint32_t pos, uint32_t v) const {
        const uint32_t lim = base + MI355_SLOT_ROWS * kStep;  // the dump row
        *(lds_u32_ptr)(uintptr_t)(pos < lim ? pos : lim) = v;
    }
};
struct StoreGlobal {  // lane-private run of kSlotWordsFull words
    static constexpr uint32_t kStep = 1;
    uint32_t* dst;
    __device__ __forceinline__ uint32_t at(uint32_t word) const { return word; }
    __device__ __forceinline__ void operator()(uint32_t w, uint32_t v) const {
        dst[w < kSlotWordsFull - 1 ? w : kSlotWordsFull - 1] = v;
    }
};

// Symbol for (run r < 16, value v != 0): LUT2 hit for |v| <= 31, else from the (run,size) table.
// Returns 0 when the reference has no code for it (quirk Q13).
__device__ __forceinline__ uint32_t symbol_slow(int v, uint32_t r, const uint32_t* __restrict__ act) {
    const int size = bit_size(v);
    if (size > 10) return 0u;
    const uint32_t a = act[(r << 4) | (uint32_t)size];
    const uint32_t len = lut_len(a);
    if (len == 0u) return 0u;
    const uint32_t t = len + (uint32_t)size;  // <= 27
    const uint32_t m = (lut_code(a) << size) | value_bits(v, size);
    return (m << (32u - t)) | t;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {  // uniform result
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {  // uniform result
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, dpp_src<0x111, 0xf>(v));
    v = mx(v, dpp_src<0x112, 0xf>(v));
    v = mx(v, dpp_src<0x114, 0xf>(v));
    v = mx(v, dpp_src<0x118, 0xf>(v));
    v = mx(v, dpp_src<0x142, 0xa>(v));
    v = mx(v, dpp_src<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct WalkA {  // stage A result: position of a symbol + its value read in flight
    uint32_t run;  // zeros between the previous symbol and this one
    uint32_t u;    // the coefficient as stored: 16 bits, zero-extended
};
struct WalkB {  // stage B result: symbol entry read in flight
    uint32_t e_fast;
    uint32_t r;
    uint32_t u;
};
typedef uint16_t __attribute__((may_alias)) u16a;

// ZRLs without a branch in the walk.  The reference emits (15,0) at every 16th zero of a run that ends in a non-zero
// (RLEBlockAC, utils.cpp:586-596).  Those positions -- prev + 16, prev + 32, ... below the next non-zero -- are made
// "virtual non-zeros" before the walk: their bit is set in the walk mask, so the walk visits them like any symbol --
// a ZERO coefficient after a run of 15, which is exactly where the symbol table keeps ZRL -- and every run it ever
// sees is below 16.  Returns the walk mask (bit 0 clear).
__device__ __forceinline__ uint64_t mark_zero_runs(uint64_t mask) {
    // Closed form (round 4; rounds 2-3 looped once per ZRL of the unit that has the most, with a wave-wide test per round).
    // x: the non-zeros, position 0 counting as the start of the first run.  For a non-zero at p the reference emits ZRLs at
    // p + 16, p + 32, p + 48 as far as they lie in front of the NEXT non-zero, i.e. candidate c = p + 16 k is one iff x has
    // no bit in (p, c].  y = x smeared upwards by 15: bit q of y says "x has a bit in [q - 15, q]", so (p, p + 16] is
    // clear iff bit c of y is, (p, p + 32] iff bits c and c - 16 of y are, and so on.  Zeros after the unit's LAST
    // non-zero end in EOB, not in ZRLs: candidates at or above it are dropped.
    const uint64_t x = mask | 1ull;
    uint64_t y = x;
    y |= y << 1;
    y |= y << 2;
    y |= y << 4;
    y |= y << 8;
    const uint64_t n1 = ~y, n2 = n1 & ~(y << 16), n3 = n2 & ~(y << 32);
    const uint64_t z = ((x << 16) & n1) | ((x << 32) & n2) | ((x << 48) & n3);
    const uint64_t below_last = (1ull << (63u - (uint32_t)__builtin_clzll(x))) - 1ull;  // x != 0
    return (mask | (z & below_last)) & ~1ull;
}

// row: this lane's unit in the [position][unit] row buffer (row[pos * 64] = coefficient at zig-zag
// position pos; position 64 = kRowSentinel); mask: the walk mask from mark_zero_runs (non-zero positions 1..63 plus
// the ZRL positions).  maxcnt: wave-uniform upper bound of the lanes' mask populations (wave_max of popcount(mask),
// formed by the caller with all lanes active: DPP reductions need the full wave).
// Per symbol: one v_ffbl_b32 on the shifted mask, one 16-bit LDS read of the value, one table read at
// [clamped value][run], one put -- and NO branch and no select: in-kernel experiments showed the walk bound by its
// exec-masked branches per symbol (rare-path tests), then by its instruction count, not by its LDS round trips.  A
// table miss on a real value (|v| > 31, or a table with holes) is only recorded; if any lane had one, the wave walks
// the pass again with the general loop, and `general` (wave-uniform, kept by the caller per channel type) makes it
// start there next time -- noise at q = 90 lives in the general loop, q = 50 never sees it.
// One pass over the walk mask.  GENERAL = false: the branch-free loop (table misses on real values are recorded in
// `miss`); GENERAL = true: misses assemble their symbol from the (run, size) table behind an exec-masked branch.  Two
// instantiations on purpose: left to itself the compiler keeps ONE loop and guards the general part with exec-mask
// juggling and a branch per symbol -- the very cost this structure removes.
// lut2 (and row) MUST live in LDS: the table read forms its LDS byte address by hand.
template <bool GENERAL, typename Store>
__device__ __forceinline__ void walk_loop(const u16a* rowu, uint64_t mask, const uint32_t* __restrict__ lut2,
                                          const uint32_t* __restrict__ act, Packer32<Store>& pk,
                                          const uint32_t maxcnt, uint32_t& miss, bool& bad) {
    // The mask is kept SHIFTED: bit 0 of m = the position after the last symbol taken.  After mark_zero_runs the next
    // symbol is never more than 16 positions away, so it is always found in the low word (one v_ffbl_b32, no 64-bit
    // search, no clearing of the bit found: the shift drops it).  A lane that has run out (m == 0: ffbl = 0xFFFFFFFF)
    // gets "run" kRunNone = 64, which takes it straight to the sentinel position 64, where it stays; the table entry
    // [kRowSentinel][64] is entry 0 of the 66th row: zero.
    uint64_t m = mask >> 1;
    uint32_t prev = 0;  // position of the last symbol handed out by stage A
    auto stageA = [&]() -> WalkA {
        WalkA a;
        uint32_t t;  // v_ffbl_b32 of 0 is 0xFFFFFFFF (__ffs costs four instructions more)
        asm("v_ffbl_b32 %0, %1" : "=v"(t) : "v"((uint32_t)m));
        t = t < kRunNone ? t : kRunNone;  // zeros in front of the symbol (< 16); kRunNone = none left
        const uint32_t nx = prev + t + 1u;
        const uint32_t pos = nx < 64u ? nx : 64u;
        a.run = t;
        m >>= (t + 1u) & 63u;
        prev = pos;
        a.u = rowu[pos * 64u];
        return a;
    };
    auto stageB = [&](const WalkA& a) -> WalkB {
        WalkB b;
        b.u = a.u;
        b.r = a.run;
        // table row = clamp(v, -32, 32) + 32 without sign extension: (u + 32) mod 2^16 is v + 32 for v in [-32, 32] and
        // something above 64 for every other value, which the minimum sends to row 64 (= +32: kLut2Miss in every column).
        // (A sign-extending read + three-input median + signed multiply-add is one instruction less and was NOT faster:
        // 253.0 against 254.7 Gpixel/s over three 100-step rounds, gpurun r4q.)
        const uint16_t t = (uint16_t)(a.u + 32u);
        const uint32_t trow = t < 64 ? t : 64;
        // LDS byte address by hand -- one shift-add, one 24-bit multiply-add (the compiler prefers three instructions)
        typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
        const uint32_t col = (uint32_t)(uintptr_t)(lds_u32)lut2 + (b.r << 2);
        uint32_t addr;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(trow), "s"(kLut2Cols * 4u), "v"(col));
        b.e_fast = *(lds_u32)(uintptr_t)addr;
        return b;
    };
    auto stageC = [&](const WalkB& b) {
        uint32_t e = b.e_fast;
        if constexpr (GENERAL) {
            if ((e & 31u) == kLut2Miss) {  // no whole-symbol entry: a large value (or a table with holes)
                e = symbol_slow((int)(int16_t)b.u, b.r, act);
                bad = bad || e == 0u;
            }
        } else {
            const uint32_t len = e & 31u;
            miss = miss > len ? miss : len;
        }
        pk.put(e);
    };
    WalkA a1 = stageA();
    WalkA a2 = stageA();
    WalkB b1 = stageB(a1);
    // two symbols per trip: the pipeline registers rotate by renaming instead of by moves (an odd
    // count runs one extra step on the sentinel, a no-op for every lane)
    for (uint32_t i = 0; i < maxcnt; i += 2) {
        WalkA a3 = stageA();
        WalkB b2 = stageB(a2);
        stageC(b1);
        WalkA a4 = stageA();
        WalkB b3 = stageB(a3);
        stageC(b2);
        a2 = a4;
        b1 = b3;
    }
}

template <bool STD, typename Store>
__device__ __forceinline__ bool walk_nonzeros(const i16a* row, uint64_t mask, const uint32_t* __restrict__ lut2,
                                              const uint32_t* __restrict__ act, Packer32<Store>& pk,
                                              const uint32_t maxcnt, bool& general) {
    const u16a* const rowu = reinterpret_cast<const u16a*>(row);
    bool bad = false;  // a non-zero coefficient without a code (quirk Q13)
    uint32_t miss = 0;
    if (!general) {
        walk_loop<false>(rowu, mask, lut2, act, pk, maxcnt, miss, bad);
        if (wave_any(miss == kLut2Miss)) {  // walk this pass again, the careful way; and start there next time
            general = true;
            pk.reset();
        }
    }
    if (general) walk_loop<true>(rowu, mask, lut2, act, pk, maxcnt, miss, bad);
    // the reference appends EOB ALWAYS (quirk Q8); a standard encoder omits it after coefficient 63
    if (!(STD && (mask >> 63))) pk.put(lut2[kLut2Eob]);
    pk.finish();
    return !bad;  // false: a size category without a code (quirk Q13)
}

// ----------------------------------------------------------------------------
// k_screen_encode: 256-thread workgroups = 4 independent persistent waves that share
// the constant tables in LDS.  One (tile, channel) per wave iteration.  Lane roles:
// MFMA phase lane = (n = lane&15: block within a group of 16, gq = lane>>4: row pair
// of the block / row group of the accumulator); walk phase lane = block.
// ----------------------------------------------------------------------------
// ----------------------------------------------------------------------------
// Quantise + verify one 16x16x64 group of the map (positions 16 mt + 4 gq + r, r = 0..3, of unit
// 16 j + n; B = the 16 units' level-shifted samples).
//
// First look: the top THREE of the five base-256 digits of Lt (A fragments resident in 12 VGPRs per row
// tile).  Y' = acc4 * 2^16 + acc3 * 2^8 + acc2 is Lt p / 2^16 without the two low digits, so
//     c/Q = Y' * 2^-23 / Q + e,   |e| <= (E1_R + delta_R) / Q,
// E1_R = 128 * (256 * sum_i |digit1[R][i]| + sum_i |digit0[R][i]|) * 2^-39 (exact worst case of the dropped
// digits, <= 2^-11; computed per row on the host), delta_R = eps_R + 2^-27 (chain rounding + map error).
// In fp32: t = acc3 * 256 + acc2 (|t| < 2^28.1, its conversion is off by <= 2^4.1), acc4 converts exactly,
// fv = fma(acc4, 2^16, fl(t)), zf = fl(fv * fl(2^-23/Q)): three roundings, |zf - Y' 2^-23/Q| <= |z| 2^-22 +
// 2^-18/Q.  rn = nearest integer of zf by the 1.5 * 2^23 trick, d = zf - rn exact.  The quantised value is
// rn whenever |d| + |zf| 2^-21 < 0.5 - (E1_R + 2^-18 + delta_R)/Q - 2^-22; the kernel tests the stronger
// d^2 < thr_R^2 with thr_R = 0.5 - (E1_R + 2^-18 + delta_R)/Q - 2^-22 - 2^-21 max|zf| over the row's inputs
// (max|zf| <= 128 sum_i |Lt3[R][i]| / Q (1 + 2^-20)), thr_R^2 rounded down on the host.
// Second look (wave-uniform, 0.2 % of the groups on q50 noise): the two low digits are fetched and all five
// give y2 = Lt p exactly in fp64; threshold 0.5 - delta_R/Q (1 + 1e-6) - 2^-38.  Standard mode is DEFINED by
// its integer map and decides exactly there.  What is still undecided sets `amb`: the exact ordered fp64
// chain then recomputes the unit (exact_unit_wave) -- that chain remains the arbiter.
// Position 0 (mt == 0, gq == 0, r == 0) is never judged here: coefficient 0 is formed exactly elsewhere.
// ----------------------------------------------------------------------------
constexpr int kLookDigits = 3;  // digits 2, 3, 4

__device__ __forceinline__ void load_look_fragments(const ScreenParams& sp, uint32_t lane, v4i (&A)[4][kLookDigits]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int l = 0; l < kLookDigits; ++l) {
            const uint4 t = sp.afrag[(mt * kScreenLimbs + (kScreenLimbs - kLookDigits) + l) * 64 + lane];
            A[mt][l] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
        }
}

// The second look of screen_quantise (rare, wave-uniform): all five digits, fp64.  tt = the first look's thr^2 - d^2.
__device__ __forceinline__ void second_look(const v4i (&acc)[kLookDigits], const int (&t)[4], const float (&tt)[4], const v4i& B,
                                            const ScreenParams& sp, uint32_t ct, int mt, uint32_t gq, uint32_t lane,
                                            uint32_t (&qb)[4], bool& amb) {
    bool a1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a1[r] = !(tt[r] > 0.0f);
    if (lane == 0) atomicAdd(&sp.stats[0], 1ull);
    const uint4 t1 = sp.afrag[(mt * kScreenLimbs + 1) * 64 + lane], t0 = sp.afrag[(mt * kScreenLimbs) * 64 + lane];
    const v4i acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)t1.x, (int)t1.y, (int)t1.z, (int)t1.w}, B,
                                                           v4i{0, 0, 0, 0}, 0, 0, 0);
    const v4i acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)t0.x, (int)t0.y, (int)t0.z, (int)t0.w}, B,
                                                           v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (a1[r]) {
            // {s1, thr1, s2, thr2} of this position: read from memory, the second look is rare
            const double* qc = sp.qconst + ((size_t)ct * 64 + 16 * mt + 4 * gq + r) * 4;
            const double y1 = (double)acc[2][r] * 65536.0 + (double)t[r];                 // exact (< 2^37)
            const double y2 = y1 * 65536.0 + (double)(acc1[r] * 256 + acc0[r]);          // exact (< 2^53)
            const double z = y2 * qc[2];
            const double tz = __builtin_fabs(z) + 0.5;
            const double fr = tz - __builtin_floor(tz);
            const int nn = (int)tz;
            qb[r] = (uint32_t)(z < 0.0 ? -nn : nn);
            amb = amb || !(__builtin_fabs(fr - 0.5) < qc[3]);
        }
    }
}

template <bool STD>
__device__ __forceinline__ void screen_quantise(const v4i (&A)[kLookDigits], const v4i& B, const ScreenParams& sp,
                                                const float* __restrict__ qf /* LDS: sf[4], thr[4] of these positions */,
                                                uint32_t ct, int mt, uint32_t gq, uint32_t lane, uint32_t (&qb)[4], bool& amb) {
    v4i acc[kLookDigits];
#pragma unroll
    for (int l = 0; l < kLookDigits; ++l) acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[l], B, v4i{0, 0, 0, 0}, 0, 0, 0);
    int t[4];
    v2f fA, fB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t[r] = acc[1][r] * 256 + acc[0][r];
        const float fv = __builtin_fmaf((float)acc[2][r], 65536.0f, (float)t[r]);
        if (r < 2) fA[r] = fv;
        else fB[r - 2] = fv;
    }
    const v2f sA = {qf[0], qf[1]}, sB = {qf[2], qf[3]};
    const v2f M2 = {12582912.0f, 12582912.0f};
    if constexpr (!STD) {
        // Strict mode, fused: a = fl(fv * sf + 1.5 * 2^23) is the magic number plus the nearest integer rn of the EXACT
        // product, d = fl(fv * sf - rn) that product's distance from it (one rounding each, where the unfused form rounds
        // the product first and measures the distance of ITS nearest integer): every error bound of the text above holds
        // with room to spare, and each pair of positions costs four packed instructions instead of five.
        const v2f aA = __builtin_elementwise_fma(fA, sA, M2), aB = __builtin_elementwise_fma(fB, sB, M2);
        const v2f rA = aA - M2, rB = aB - M2;
        const v2f dA = __builtin_elementwise_fma(fA, sA, -rA), dB = __builtin_elementwise_fma(fB, sB, -rB);
        const v2f hA = {qf[4], qf[5]}, hB = {qf[6], qf[7]};
        const v2f tA = __builtin_elementwise_fma(-dA, dA, hA), tB = __builtin_elementwise_fma(-dB, dB, hB);
        float tt[4] = {tA[0], tA[1], tB[0], tB[1]};
        if (mt == 0 && gq == 0) tt[0] = 1.0f;  // coefficient 0: overwritten by the caller, never judged here
        qb[0] = __float_as_uint(aA[0]), qb[1] = __float_as_uint(aA[1]), qb[2] = __float_as_uint(aB[0]), qb[3] = __float_as_uint(aB[1]);  // 0x4B400000 + q
        const float tmin = __builtin_fminf(__builtin_fminf(__builtin_fminf(tt[0], tt[1]), tt[2]), tt[3]);
        if (wave_any(!(tmin > 0.0f))) second_look(acc, t, tt, B, sp, ct, mt, gq, lane, qb, amb);
        return;
    }
    const v2f zA = fA * sA, zB = fB * sB;
    const v2f aA = zA + M2, aB = zB + M2;
    const float aa[4] = {aA[0], aA[1], aB[0], aB[1]};
    if constexpr (STD) {
        // Standard mode IS this arithmetic (it is not a behaviour of the reference, so there is nothing to verify it
        // against): the DCT-II as a fixed-point map of 23 fractional bits, evaluated exactly on the matrix units; the
        // quotient by Q in fp32 -- two integer -> float conversions, one fma, one product, each rounded to nearest even --
        // and the nearest integer of that by the 1.5 * 2^23 addition (ties to even).  The test suite's checker restates
        // the same operations in the same order.
#pragma unroll
        for (int r = 0; r < 4; ++r) qb[r] = __float_as_uint(aa[r]);  // 0x4B400000 + q
        return;
    }
}

// ----------------------------------------------------------------------------
// The arbiter: the reference's ordered in-place fp64 chain (utils.cpp:314-348) for ONE unit, run by
// the whole wave for a unit whose screened transform left a coefficient undecided (rare: ~1e-7 of
// the units).  Lane l owns sample / coefficient l = y*8 + x.  Each of the 64 dependent steps (u
// outer, v inner) forms its 64 terms (P[y][x]*C[x][u])*C[y][v] in parallel -- multiplying by
// C[.][0] == 1.0 is the identity, so always multiplying changes nothing -- and lane 0 adds them in
// the reference's order (y outer, x inner, from 0.0), scales, and publishes P[v][u].  About 25 us
// per unit, but only a handful of registers and 520 bytes of LDS: no second kernel, no scratch.
// Rewrites the unit's zig-zag row and non-zero mask in LDS; the caller then carries on as if the
// screen had produced them.  All lanes must be active.
// ----------------------------------------------------------------------------
__device__ __forceinline__ void exact_unit_wave(const uint8_t* __restrict__ f, const Geom& g, uint32_t chan, uint32_t bx,
                                                uint32_t by, const double* __restrict__ qd, double* lds /* 65 doubles + the 64 of the cosine table from lds[72] */,
                                                i16a* row /* the unit's column of the row buffer */, uint32_t* mlo,
                                                uint32_t* mhi, uint32_t lane) {
    static constexpr double kCos[8][8] = MI355_COS_TABLE;
    static constexpr uint8_t kZz[64] = MI355_ZIGZAG_TABLE;  // zig-zag position -> natural index
    const uint32_t y = lane >> 3, x = lane & 7;
    const bool avg = (chan != 0) && (g.flags & 1u);
    const uint32_t smp = sample_generic(f, g, avg, bx * 8 + x, by * 8 + y, csc_k(chan, 0), csc_k(chan, 1), csc_k(chan, 2),
                                        csc_k(chan, 3));
    double p = (double)((int)smp - 128);  // quirk Q4
    // the cosine table through LDS (behind the 65 doubles of the steps): read from memory inside the loops, every one of the 64
    // dependent steps waited for a load -- and for every store the wave had in flight -- before its first multiply: 39 us per
    // unit, of which the sums are the smaller part (round 4, tools/slow_frames_probe.py)
    double* const tab = lds + 72;
    tab[lane] = kCos[y][x];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (uint32_t u = 0; u < 8; ++u) {
        const double cxu = tab[x * 8 + u];
#pragma unroll 1
        for (uint32_t v = 0; v < 8; ++v) {
            lds[lane] = (p * cxu) * tab[y * 8 + v];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                double sum = 0.0;
#pragma unroll 8
                for (int i = 0; i < 64; ++i) sum += lds[i];
                sum *= (u == 0 && v == 0) ? kScale00 : ((u == 0 || v == 0) ? kScale0X : kScaleXX);
                lds[64] = sum;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == v * 8 + u) p = lds[64];  // P[v][u] = s, before the next step (quirk Q5)
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int q = (int)__builtin_round(p / qd[(chan ? 64 : 0) + lane]);  // natural index v*8+u == lane (quirk Q6)
    // natural order -> zig-zag order through LDS, then the non-zero mask by ballot
    int* qn = reinterpret_cast<int*>(lds);
    __builtin_amdgcn_wave_barrier();
    qn[lane] = q;
    __builtin_amdgcn_wave_barrier();
    const int qz = qn[kZz[lane]];  // coefficient at zig-zag position `lane`
    const uint64_t nz = __ballot(qz != 0);
    row[lane * 64u] = (int16_t)qz;
    if (lane == 0) {
        *mlo = (uint32_t)nz & ~1u;  // bit 0 = "undecided" flag: cleared; coefficient 0 is not walked
        *mhi = (uint32_t)(nz >> 32);
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace mi355
