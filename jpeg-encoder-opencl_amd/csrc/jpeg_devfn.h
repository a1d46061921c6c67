// Device-side building blocks shared by the HIP translation units of the strict JPEG
// encode path (jpeg_kernels.hip: exact fp64 pipeline; jpeg_screen_kernels.hip: the
// integer-MFMA screened pipeline).  Everything here must reproduce the reference
// bit for bit; both files are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpeg_device.h"
#include "jpeg_tables.h"

#pragma clang fp contract(off)

namespace mi355 {

// ----------------------------------------------------------------------------
// constant tables
// ----------------------------------------------------------------------------
__device__ __forceinline__ constexpr double cos_tab(int a, int k) {
    constexpr double T[8][8] = MI355_COS_TABLE;
    return T[a][k];
}
__device__ __forceinline__ constexpr int zigzag_nat(int k) {
    constexpr uint8_t Z[64] = MI355_ZIGZAG_TABLE;
    return Z[k];
}
__device__ __forceinline__ constexpr double csc_k(int chan, int i) {
    constexpr double K[3][4] = MI355_CSC_TABLE;
    return K[chan][i];
}

// ----------------------------------------------------------------------------
// sample stage: performCSC (utils.cpp:92-110), performCDS (:113-141), mirror
// padding (:199-233), for ONE output channel.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t csc1(uint32_t r, uint32_t g, uint32_t b, double k0, double k1,
                                         double k2, double k3) {
    // doubles, left to right, truncating cast (quirk Q1)
    double v = (((k0 * (double)r) + (k1 * (double)g)) + (k2 * (double)b)) + k3;
    return (uint32_t)(int)v;
}

__device__ __forceinline__ uint32_t csc_at(const uint8_t* __restrict__ f, uint32_t W, uint32_t x,
                                           uint32_t y, double k0, double k1, double k2, double k3) {
    const uint8_t* p = f + ((size_t)y * W + x) * 3;
    return csc1(p[0], p[1], p[2], k0, k1, k2, k3);
}

// Value of padded pixel (px,py) of channel `chan` exactly as the reference forms
// it: CDS runs on the UNPADDED image over complete 2x2 quads only (quirk Q2),
// then the canvas is mirrored, right first, then bottom (quirk Q3).
__device__ __forceinline__ uint32_t sample_generic(const uint8_t* __restrict__ f, const Geom& g,
                                                   bool avg, uint32_t px, uint32_t py, double k0,
                                                   double k1, double k2, double k3) {
    uint32_t mx = px < g.W ? px : 2 * g.W - 1 - px;
    uint32_t my = py < g.H ? py : 2 * g.H - 1 - py;
    if (avg) {
        uint32_t qx = mx & ~1u, qy = my & ~1u;
        if (qx + 1 < g.W && qy + 1 < g.H) {
            uint32_t s = csc_at(f, g.W, qx, qy, k0, k1, k2, k3) +
                         csc_at(f, g.W, qx + 1, qy, k0, k1, k2, k3) +
                         csc_at(f, g.W, qx, qy + 1, k0, k1, k2, k3) +
                         csc_at(f, g.W, qx + 1, qy + 1, k0, k1, k2, k3);
            return s >> 2;  // (uint8_t)(sum / 4.0)
        }
    }
    return csc_at(f, g.W, mx, my, k0, k1, k2, k3);
}

// Loads the 64 samples of block (bx,by), channel `chan`, packed 4 per dword
// (sample y*8+x in byte (y*8+x)&3 of pk[(y*8+x)>>2]).
// FAST: every block of this wave lies inside the image and rows are 8-byte
// aligned (W % 8 == 0, base 8-aligned): 24 coalescing-friendly 8-byte loads.
template <bool FAST>
__device__ __forceinline__ void load_samples(const uint8_t* __restrict__ f, const Geom& g, int chan,
                                             bool avg, uint32_t bx, uint32_t by, uint32_t lane,
                                             uint32_t* lds, uint32_t (&pk)[16]) {
    const double k0 = csc_k(chan, 0), k1 = csc_k(chan, 1), k2 = csc_k(chan, 2), k3 = csc_k(chan, 3);
    if constexpr (FAST) {
#pragma unroll
        for (int yp = 0; yp < 4; ++yp) {  // row pairs
            uint32_t w[2][6];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint2* p = reinterpret_cast<const uint2*>(
                    f + ((size_t)(by * 8 + yp * 2 + r) * g.W + bx * 8) * 3);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    uint2 v = p[j];
                    w[r][2 * j] = v.x;
                    w[r][2 * j + 1] = v.y;
                }
            }
            uint32_t val[2][8];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    uint32_t c[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        int byte = 3 * x + k;
                        c[k] = (w[r][byte >> 2] >> (8 * (byte & 3))) & 255u;
                    }
                    val[r][x] = csc1(c[0], c[1], c[2], k0, k1, k2, k3);
                }
            if (avg) {
#pragma unroll
                for (int x = 0; x < 8; x += 2) {
                    uint32_t m = (val[0][x] + val[0][x + 1] + val[1][x] + val[1][x + 1]) >> 2;
                    val[0][x] = val[0][x + 1] = val[1][x] = val[1][x + 1] = m;
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    pk[(yp * 2 + r) * 2 + h] = val[r][4 * h] | (val[r][4 * h + 1] << 8) |
                                               (val[r][4 * h + 2] << 16) | (val[r][4 * h + 3] << 24);
        }
    } else {
        // Edge / unaligned waves: one sample at a time through LDS so that the
        // register array is only ever indexed statically.
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            uint32_t v = 0;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                int s = i * 4 + j;
                uint32_t smp = sample_generic(f, g, avg, bx * 8 + (s & 7), by * 8 + (s >> 3), k0, k1,
                                              k2, k3);
                v |= smp << (8 * j);
            }
            lds[i * 64 + lane] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) pk[i] = lds[i * 64 + lane];
    }
}

// ----------------------------------------------------------------------------
// The reference's in-place "DCT" (utils.cpp:314-348, quirk Q5), one lane per
// unit, 64 doubles in registers.
//
// For u = 0..7, v = 0..7:   s = sum_{y,x in that order, from 0.0} (P[y][x]*C[x][u])*C[y][v]
//                           s *= scale(u,v);  P[v][u] = s   (stored BEFORE the next step)
//
// Bit-exact savings used here (nothing is re-associated):
//  * t[y][x] = P[y][x]*C[x][u] is formed once per u; inside the v loop only
//    P[v][u] changes, so only t[v][u] is refreshed (the same product the
//    reference forms again and again).
//  * C[.][0] == 1.0 exactly, so products with it are the identity and skipped.
// ----------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void chain_u_static(double (&P)[64]) {
    double t[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) t[i] = (U == 0) ? P[i] : P[i] * cos_tab(i & 7, U);
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        double s = 0.0;
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int x = 0; x < 8; ++x)
                s += (v == 0) ? t[y * 8 + x] : t[y * 8 + x] * cos_tab(y, v);
        s *= (U == 0 && v == 0) ? kScale00 : ((U == 0 || v == 0) ? kScale0X : kScaleXX);
        P[v * 8 + U] = s;
        t[v * 8 + U] = (U == 0) ? s : s * cos_tab(U, U);
    }
}

// Same iteration with a run-time (wave-uniform) u >= 1: keeps the code small
// (one copy of the 8 unrolled v steps) at the price of a uniform branch per step
// to pick the destination register.
__device__ __forceinline__ void chain_u_dynamic(double (&P)[64], int u) {
    double cx[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) cx[x] = cos_tab(x, u);
    const double cuu = cos_tab(u, u);
    double t[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) t[i] = P[i] * cx[i & 7];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        double s = 0.0;
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int x = 0; x < 8; ++x)
                s += (v == 0) ? t[y * 8 + x] : t[y * 8 + x] * cos_tab(y, v);
        s *= (v == 0) ? kScale0X : kScaleXX;
        const double tu = s * cuu;
        switch (u) {
#define MI355_CASE(c)     \
    case c:               \
        P[v * 8 + c] = s; \
        t[v * 8 + c] = tu; \
        break;
            MI355_CASE(1)
            MI355_CASE(2)
            MI355_CASE(3)
            MI355_CASE(4)
            MI355_CASE(5)
            MI355_CASE(6)
            default:
                P[v * 8 + 7] = s;
                t[v * 8 + 7] = tu;
                break;
#undef MI355_CASE
        }
    }
}

template <int MODE>
__device__ __forceinline__ void chain(double (&P)[64]) {
    chain_u_static<0>(P);
    if constexpr (MODE == 0) {
        chain_u_static<1>(P);
        chain_u_static<2>(P);
        chain_u_static<3>(P);
        chain_u_static<4>(P);
        chain_u_static<5>(P);
        chain_u_static<6>(P);
        chain_u_static<7>(P);
    } else {
#pragma unroll 1
        for (int u = 1; u < 8; ++u) chain_u_dynamic(P, u);
    }
}


// ----------------------------------------------------------------------------
// entropy helpers
// ----------------------------------------------------------------------------
// Size category = bit length of |v| with v narrowed to int16 (getValueCategory,
// utils.cpp:623-627).
__device__ __forceinline__ int bit_size(int v) {
    int a = v < 0 ? -v : v;
    return 32 - __clz(a);  // __clz(0) == 32
}
// Value bits (valueToBitString, utils.cpp:630-653): v, or v + 2^size - 1 for v < 0.
__device__ __forceinline__ uint32_t value_bits(int v, int size) {
    return (uint32_t)(v < 0 ? v + (1 << size) - 1 : v);
}

// Device Huffman LUT entry: code << 5 | len  (len <= 17, code < 2^17).
// lut layout: [table 0..3][256], index (run << 4) | size.
__device__ __forceinline__ uint32_t lut_len(uint32_t e) { return e & 31u; }
__device__ __forceinline__ uint32_t lut_code(uint32_t e) { return e >> 5; }

// Walks one unit's 63 AC coefficients (RLEBlockAC, utils.cpp:572-609 fused with
// the AC loop of HuffmanEncoder, :683-694) and calls put(code, len) for every
// symbol incl. value bits.  `c` = 32 packed coefficient pairs.
// Returns false if a size category has no code (quirk Q13 -> error).
template <typename Put>
__device__ __forceinline__ bool walk_ac(const uint32_t (&c)[32], const uint32_t* __restrict__ act,
                                        Put&& put) {
    bool ok = true;
    int run = 0;
    const uint32_t zrl = act[0xF0], eob = act[0x00];
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        int v = (int)(int16_t)((k & 1) ? (c[k >> 1] >> 16) : (c[k >> 1] & 0xffffu));
        if (v == 0) {
            ++run;
        } else {
            while (run >= 16) {  // the reference emits (15,0) at every 16th zero before a later non-zero
                put(lut_code(zrl), lut_len(zrl));
                run -= 16;
            }
            int size = bit_size(v);
            uint32_t e = (size <= 10) ? act[(run << 4) | size] : 0u;
            if (lut_len(e) == 0) {
                ok = false;
            } else {
                put((lut_code(e) << size) | value_bits(v, size), lut_len(e) + size);
            }
            run = 0;
        }
    }
    put(lut_code(eob), lut_len(eob));  // ALWAYS (quirk Q8)
    return ok;
}

// DC symbol (utils.cpp:665-680).
template <typename Put>
__device__ __forceinline__ bool put_dc(int diff, const uint32_t* __restrict__ dct, Put&& put) {
    int d = (int)(int16_t)diff;  // argument narrowed to int16_t
    int size = bit_size(d);
    uint32_t e = (size <= 11) ? dct[size] : 0u;
    if (lut_len(e) == 0) return false;
    put((lut_code(e) << size) | value_bits(d, size), lut_len(e) + size);
    return true;
}

// Wave-wide inclusive prefix sum / maximum with DPP row shifts and row broadcasts (6 VALU
// instructions, no LDS round trips like __shfl).  Lane 63 ends up with the reduction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_src(uint32_t v) {
    // lanes without a valid source (or outside ROW_MASK) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t /*lane*/) {
    v += dpp_src<0x111, 0xf>(v);  // row_shr:1
    v += dpp_src<0x112, 0xf>(v);  // row_shr:2
    v += dpp_src<0x114, 0xf>(v);  // row_shr:4
    v += dpp_src<0x118, 0xf>(v);  // row_shr:8
    v += dpp_src<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
    v += dpp_src<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
    return v;
}
// The same scan with the DPP operand inside the add (v_add_u32_dpp: six instructions).  Left to itself the compiler
// combines `v += dpp_src(v)` that way in some kernels and not in others (k_merge: six moves of zero, six v_mov_b32_dpp,
// six adds); bound_ctrl:1 makes lanes without a source read 0, lanes outside the row mask keep their value.  The s_nop
// between the steps is the two wait states a DPP read needs behind the write of its operand.  All lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    asm volatile(
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ void load_unit(const uint32_t* __restrict__ src, uint32_t (&c)[32]) {
#pragma unroll
    for (int p = 0; p < 32; ++p) c[p] = src[p * 64];
}

// DC predictor of lane's block = DC of the previous block of the same channel
// (never reset inside a frame, utils.cpp:665-671); 0 for the first block.
__device__ __forceinline__ int dc_predictor(const uint32_t* __restrict__ coefs_frame, uint32_t tile,
                                            uint32_t chan, uint32_t lane, int own_dc) {
    int prev = __shfl_up(own_dc, 1);
    if (lane == 0) {
        prev = 0;
        if (tile > 0)
            prev = (int)(int16_t)(coefs_frame[((size_t)(tile - 1) * 3 + chan) * 2048 + 63] & 0xffffu);
    }
    return prev;
}


#ifndef MI355_EMIT_LDS_WORDS
#define MI355_EMIT_LDS_WORDS 4096
#endif
// Bit-assembly window of k_emit / k_merge: 4096 words = 16 KiB = 131072 bits = 682 bits per unit on
// average (tiles with more bits take the direct-to-memory path).  With its 17,280 B of LDS one merge
// workgroup fits on a CU next to two workgroups of k_screen_encode (2 x 72,576 B of 160 KiB, in 1280-byte granules).
constexpr uint32_t kEmitLdsWords = MI355_EMIT_LDS_WORDS;
// k_merge's half-size window (two workgroups per CU beside the block encode: 2 x 8,896 B within the same 17,920 B)
constexpr uint32_t kEmitLdsWordsSmall = 2000;

struct BitWriterLds {
    uint32_t* words;  // LDS
    uint64_t acc;     // right-aligned pending bits
    uint32_t n;       // pending bit count (< 32 after every put)
    uint32_t w;       // next word index
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {
        acc = (acc << len) | code;
        n += len;
        if (n >= 32) {
            n -= 32;
            atomicOr(&words[w++], (uint32_t)(acc >> n));
            acc &= (1ull << n) - 1;
        }
    }
    __device__ __forceinline__ void flush() {
        if (n) atomicOr(&words[w], (uint32_t)(acc << (32 - n)));
    }
};

struct BitWriterGlobal {
    uint32_t* words;  // global, word index relative to the frame's output
    uint64_t acc;
    uint32_t n;
    uint64_t w;
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {
        acc = (acc << len) | code;
        n += len;
        if (n >= 32) {
            n -= 32;
            atomicOr(&words[w++], __builtin_bswap32((uint32_t)(acc >> n)));
            acc &= (1ull << n) - 1;
        }
    }
    __device__ __forceinline__ void flush() {
        if (n) atomicOr(&words[w], __builtin_bswap32((uint32_t)(acc << (32 - n))));
    }
};


}  // namespace mi355
