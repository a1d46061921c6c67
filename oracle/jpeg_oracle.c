/* TEST INFRASTRUCTURE ONLY -- see jpeg_oracle.h.
 *
 * CPU restatement of the reference's strict encode path.  Every function names
 * the reference lines it follows (paths relative to /root/reference).  The
 * behaviours that make the output differ from a standard JPEG encoder (SURVEY.md
 * Appendix A, Q1-Q13) are reproduced on purpose and marked "quirk".
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).  FMA contraction or
 * -ffast-math would change the fp64 transform chain.
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ tables */

/* cos((2a+1)*k*M_PI/16.0) as glibc 2.35 returns it for the reference's argument
 * expression (utils.cpp:330).  Captured from oracle/_ref (tools/make_golden.py
 * re-checks it); shipped as data because the table is not sign-symmetric to
 * the last bit and a different libm could round differently. */
static const double COS_TAB[8][8] = {
    {0x1.0000000000000p+0, 0x1.f6297cff75cb0p-1, 0x1.d906bcf328d46p-1, 0x1.a9b66290ea1a3p-1, 0x1.6a09e667f3bcdp-1, 0x1.1c73b39ae68c9p-1, 0x1.87de2a6aea964p-2, 0x1.8f8b83c69a60dp-3},
    {0x1.0000000000000p+0, 0x1.a9b66290ea1a3p-1, 0x1.87de2a6aea964p-2, -0x1.8f8b83c69a608p-3, -0x1.6a09e667f3bccp-1, -0x1.f6297cff75cb0p-1, -0x1.d906bcf328d47p-1, -0x1.1c73b39ae68c8p-1},
    {0x1.0000000000000p+0, 0x1.1c73b39ae68c9p-1, -0x1.87de2a6aea962p-2, -0x1.f6297cff75cb0p-1, -0x1.6a09e667f3bcep-1, 0x1.8f8b83c69a60cp-3, 0x1.d906bcf328d44p-1, 0x1.a9b66290ea1a5p-1},
    {0x1.0000000000000p+0, 0x1.8f8b83c69a60dp-3, -0x1.d906bcf328d46p-1, -0x1.1c73b39ae68c8p-1, 0x1.6a09e667f3bcbp-1, 0x1.a9b66290ea1a5p-1, -0x1.87de2a6aea965p-2, -0x1.f6297cff75cb2p-1},
    {0x1.0000000000000p+0, -0x1.8f8b83c69a608p-3, -0x1.d906bcf328d47p-1, 0x1.1c73b39ae68c5p-1, 0x1.6a09e667f3bcep-1, -0x1.a9b66290ea1a2p-1, -0x1.87de2a6aea971p-2, 0x1.f6297cff75cb0p-1},
    {0x1.0000000000000p+0, -0x1.1c73b39ae68c6p-1, -0x1.87de2a6aea96dp-2, 0x1.f6297cff75cb0p-1, -0x1.6a09e667f3bc5p-1, -0x1.8f8b83c69a602p-3, 0x1.d906bcf328d46p-1, -0x1.a9b66290ea1a1p-1},
    {0x1.0000000000000p+0, -0x1.a9b66290ea1a4p-1, 0x1.87de2a6aea967p-2, 0x1.8f8b83c69a61dp-3, -0x1.6a09e667f3bc9p-1, 0x1.f6297cff75cb2p-1, -0x1.d906bcf328d43p-1, 0x1.1c73b39ae68c2p-1},
    {0x1.0000000000000p+0, -0x1.f6297cff75cb0p-1, 0x1.d906bcf328d44p-1, -0x1.a9b66290ea1a2p-1, 0x1.6a09e667f3bc4p-1, -0x1.1c73b39ae68c2p-1, 0x1.87de2a6aea95fp-2, -0x1.8f8b83c69a616p-3},
};
/* alpha_u*alpha_v/4.0 with alpha_0 = 1.0/sqrt(2) (utils.cpp:318-319,336). */
static const double SCALE_00 = 0x1.ffffffffffffep-4; /* u=0,v=0 */
static const double SCALE_0X = 0x1.6a09e667f3bccp-3; /* exactly one of u,v is 0 */
static const double SCALE_XX = 0x1p-2;               /* neither */

double orc_cos(int a, int k) { return COS_TAB[a][k]; }
double orc_scale(int u, int v) {
    if (u == 0 && v == 0) return SCALE_00;
    if (u == 0 || v == 0) return SCALE_0X;
    return SCALE_XX;
}

/* Zig-zag by walking the 15 anti-diagonals, alternating direction
 * (utils.cpp:539-551): even diagonals run bottom-left -> top-right. */
void orc_zigzag_order(uint8_t zz[64]) {
    int k = 0;
    for (int d = 0; d < 15; ++d) {
        int lo = d < 8 ? 0 : d - 7, hi = d < 8 ? d : 7;
        for (int i = lo; i <= hi; ++i) {
            int row = (d & 1) ? i : d - i;
            int col = d - row;
            zz[k++] = (uint8_t)(row * 8 + col);
        }
    }
}

/* ITU-T T.81 Annex K.1 tables (what utils.hpp:42-62 holds). */
static const uint8_t Q50_LUM[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t Q50_CHR[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void orc_quant_q50(uint32_t qlum[64], uint32_t qchrom[64]) {
    for (int i = 0; i < 64; ++i) {
        qlum[i] = Q50_LUM[i];
        qchrom[i] = Q50_CHR[i];
    }
}

/* Build convention (the reference has only q=50): IJG quality scaling,
 * clamped to 1..255 (baseline). */
void orc_quant_ijg(int quality, uint32_t qlum[64], uint32_t qchrom[64]) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        long a = ((long)Q50_LUM[i] * scale + 50) / 100;
        long b = ((long)Q50_CHR[i] * scale + 50) / 100;
        qlum[i] = (uint32_t)(a < 1 ? 1 : a > 255 ? 255 : a);
        qchrom[i] = (uint32_t)(b < 1 ? 1 : b > 255 ? 255 : b);
    }
}

/* Annex K.3 Huffman specifications (BITS / HUFFVAL).  huffman.hpp spells the
 * resulting canonical codes out as '0'/'1' strings. */
static const uint8_t BITS_DC_L[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t BITS_DC_C[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t VAL_DC[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t BITS_AC_L[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t VAL_AC_L[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};
static const uint8_t BITS_AC_C[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t VAL_AC_C[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};

typedef struct {
    uint32_t code[256];
    uint8_t len[256]; /* 0 = no code */
} huff_tab;

static huff_tab HT[4];  /* DC luma, DC chroma, AC luma, AC chroma; index = run<<4 | size */
static huff_tab HTS[4]; /* the same without the reference's 17-bit typos (standard mode) */
static int HT_ready;

static void canon(const uint8_t bits[16], const uint8_t *val, huff_tab *t) {
    memset(t, 0, sizeof *t);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i) {
            t->code[val[k]] = code++;
            t->len[val[k]] = (uint8_t)l;
            ++k;
        }
        code <<= 1;
    }
}

static void build_tables(void) {
    if (HT_ready) return;
    canon(BITS_DC_L, VAL_DC, &HT[0]);
    canon(BITS_DC_C, VAL_DC, &HT[1]);
    canon(BITS_AC_L, VAL_AC_L, &HT[2]);
    canon(BITS_AC_C, VAL_AC_C, &HT[3]);
    memcpy(HTS, HT, sizeof HT);
    /* quirk Q11 (huffman.hpp:92-98): the AC-luma codes for run 3, sizes 4..10
     * carry one extra leading '1' and are 17 bits long. */
    for (int s = 4; s <= 10; ++s) {
        int rs = (3 << 4) | s;
        HT[2].code[rs] |= 1u << HT[2].len[rs];
        HT[2].len[rs] += 1;
    }
    HT_ready = 1;
}

int orc_huff_code(int table, int run, int size, uint32_t *code) {
    build_tables();
    if (table < 0 || table > 3 || run < 0 || size < 0) return -1;
    if (table < 2) { /* DC vectors have 12 entries (huffman.hpp:9-23,26-40) */
        if (run != 0 || size > 11) return -1;
    } else { /* AC: 16 runs x 11 sizes (huffman.hpp:43-247) */
        if (run > 15 || size > 10) return -1;
    }
    int rs = (run << 4) | size;
    if (!HT[table].len[rs]) return -1; /* the "NULL" strings */
    if (code) *code = HT[table].code[rs];
    return HT[table].len[rs];
}

/* ------------------------------------------------------- pixel-domain stages */

/* quirk Q1 (utils.cpp:106-108): doubles, left-to-right, truncating cast. */
void orc_csc(uint8_t *px, size_t npix) {
    for (size_t i = 0; i < npix; ++i) {
        uint8_t r = px[3 * i], g = px[3 * i + 1], b = px[3 * i + 2];
        px[3 * i + 0] = (uint8_t)(0.299 * r + 0.587 * g + 0.114 * b);
        px[3 * i + 1] = (uint8_t)(-0.168736 * r - 0.331264 * g + 0.5 * b + 128);
        px[3 * i + 2] = (uint8_t)(0.5 * r - 0.418688 * g - 0.081312 * b + 128);
    }
}

/* quirk Q2 (utils.cpp:120-138): complete 2x2 quads only, mean replicated into
 * all four pixels (planes stay full resolution), Y untouched.  The sum of four
 * bytes divided by 4.0 and truncated is sum>>2. */
void orc_cds(uint8_t *px, size_t W, size_t H) {
    if (W < 2 || H < 2) return; /* size_t loop bounds W-1, H-1 admit no quad */
    for (size_t y = 0; y + 1 < H; y += 2)
        for (size_t x = 0; x + 1 < W; x += 2) {
            uint8_t *p[4] = {px + 3 * (y * W + x), px + 3 * (y * W + x + 1),
                             px + 3 * ((y + 1) * W + x), px + 3 * ((y + 1) * W + x + 1)};
            for (int c = 1; c <= 2; ++c) {
                unsigned s = p[0][c] + p[1][c] + p[2][c] + p[3][c];
                uint8_t m = (uint8_t)(s >> 2);
                p[0][c] = p[1][c] = p[2][c] = p[3][c] = m;
            }
        }
}

void orc_padded_size(size_t W, size_t H, size_t *W8, size_t *H8) {
    *W8 = (W + 7) / 8 * 8; /* ceil(w/8.0)*8, utils.cpp:185-186 */
    *H8 = (H + 7) / 8 * 8;
}

/* quirk Q3 (utils.cpp:199-233): mirror including the edge pixel, right first,
 * then bottom over the full padded width. */
void orc_pad(const uint8_t *src, size_t W, size_t H, uint8_t *dst, size_t W8, size_t H8) {
    for (size_t y = 0; y < H8; ++y) {
        size_t sy = y < H ? y : H - 1 - (y - H);
        for (size_t x = 0; x < W8; ++x) {
            size_t sx = x < W ? x : W - 1 - (x - W);
            memcpy(dst + 3 * (y * W8 + x), src + 3 * (sy * W + sx), 3);
        }
    }
}

/* quirk Q5 (utils.cpp:314-348): the textbook double sum evaluated IN PLACE.
 * Output (u,v) is stored to P[v][u] before the next output is formed, so later
 * outputs read earlier outputs.  u outer, v inner; sum y outer, x inner,
 * starting from 0.0; each term (P*cos_x)*cos_y. */
void orc_dct_block(double P[64]) {
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            double s = 0.0;
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) s += P[y * 8 + x] * COS_TAB[x][u] * COS_TAB[y][v];
            s *= orc_scale(u, v);
            P[v * 8 + u] = s;
        }
}

/* the same chain on n blocks of 64 doubles, in place (for the tests that sweep millions of blocks) */
void orc_dct_blocks(double *P, size_t n) {
    for (size_t i = 0; i < n; ++i) orc_dct_block(P + 64 * i);
}

/* quirk Q6 (utils.cpp:457-463): correctly rounded divide, then round half away
 * from zero. */
void orc_quant_block(double P[64], const uint32_t q[64]) {
    for (int i = 0; i < 64; ++i) P[i] = round(P[i] / q[i]);
}

/* ------------------------------------------------------------- entropy stage */

struct orc_bitsink {
    uint8_t *buf;
    size_t cap;    /* bytes */
    uint64_t nbit; /* bits written */
};

static int sink_put(struct orc_bitsink *s, uint32_t code, int len) {
    for (int i = len - 1; i >= 0; --i) {
        size_t byte = (size_t)(s->nbit >> 3);
        if (byte >= s->cap) {
            size_t ncap = s->cap ? s->cap * 2 : 4096;
            uint8_t *nb = (uint8_t *)realloc(s->buf, ncap);
            if (!nb) return ORC_E_CAPACITY;
            memset(nb + s->cap, 0, ncap - s->cap);
            s->buf = nb;
            s->cap = ncap;
        }
        if ((code >> i) & 1) s->buf[byte] |= (uint8_t)(0x80u >> (s->nbit & 7));
        s->nbit++;
    }
    return 0;
}

/* quirk Q9 (utils.cpp:623-653): size = bit length of |v| (argument narrowed to
 * int16_t), negative values as v + 2^size - 1. */
static int bit_size(int32_t v) {
    int16_t w = (int16_t)v;
    int a = w < 0 ? -w : w, n = 0;
    while (a) {
        ++n;
        a >>= 1;
    }
    return n;
}
static uint32_t value_bits(int32_t v, int size) {
    int16_t w = (int16_t)v;
    return (uint32_t)(w >= 0 ? w : w + (1 << size) - 1) & ((1u << size) - 1);
}

static int put_symbol(struct orc_bitsink *sink, int table, int run, int32_t v, int *nbits) {
    int size = bit_size(v);
    uint32_t code;
    int len = orc_huff_code(table, run, size, &code);
    if (len < 0) return ORC_E_CATEGORY; /* quirk Q13: the reference would read out of bounds */
    *nbits += len + size;
    if (sink) {
        int e = sink_put(sink, code, len);
        if (e) return e;
        if (size) e = sink_put(sink, value_bits(v, size), size);
        if (e) return e;
    }
    return 0;
}

/* One unit = one 8x8 block of one channel, already zig-zagged.
 * DC: utils.cpp:665-680.  AC: RLE of utils.cpp:572-609 (quirk Q8: `last` is
 * searched down to index 0, ZRL on the 16th zero, and (0,0) is appended
 * ALWAYS, even when coefficient 63 is non-zero) fed to utils.cpp:683-694. */
int orc_unit_bits(const int32_t zz[64], int32_t dc_diff, int chroma, struct orc_bitsink *sink) {
    int nbits = 0, e;
    if ((e = put_symbol(sink, chroma ? 1 : 0, 0, dc_diff, &nbits))) return e;
    int last = 0;
    for (int i = 63; i >= 0; --i)
        if (zz[i] != 0) {
            last = i;
            break;
        }
    int run = 0, tab = chroma ? 3 : 2;
    for (int i = 1; i <= last; ++i) {
        if (zz[i] == 0) {
            if (run == 15) {
                if ((e = put_symbol(sink, tab, 15, 0, &nbits))) return e;
                run = 0;
            } else {
                ++run;
            }
        } else {
            if ((e = put_symbol(sink, tab, run, zz[i], &nbits))) return e;
            run = 0;
        }
    }
    if ((e = put_symbol(sink, tab, 0, 0, &nbits))) return e;
    return nbits;
}

/* quirk Q10 (utils.cpp:665-695): block-major, Y then Cb then Cr per block, one
 * DC predictor per channel starting at 0 and never reset. */
int orc_entropy(const int32_t *zigzag, size_t N, uint8_t **bits, size_t *bits_bytes,
                uint64_t *n_bits, uint32_t *unit_bits) {
    struct orc_bitsink sink = {0, 0, 0};
    int32_t pred[3] = {0, 0, 0};
    for (size_t i = 0; i < N; ++i)
        for (int c = 0; c < 3; ++c) {
            const int32_t *zz = zigzag + (i + N * (size_t)c) * 64;
            int32_t diff = zz[0] - pred[c];
            pred[c] = zz[0];
            int n = orc_unit_bits(zz, diff, c != 0, bits ? &sink : 0);
            if (n < 0) {
                free(sink.buf);
                return n;
            }
            if (!bits) sink.nbit += (uint64_t)n;
            if (unit_bits) unit_bits[3 * i + c] = (uint32_t)n;
        }
    if (bits) {
        *bits = sink.buf;
        *bits_bytes = (size_t)((sink.nbit + 7) / 8);
    }
    *n_bits = sink.nbit;
    return 0;
}

/* --------------------------------------------------------------- whole path */

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int orc_encode(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
               const uint32_t qchrom[64], int cds_on, int keep, orc_result *out) {
    if (!rgb || !out || W == 0 || H == 0) return ORC_E_ARG;
    memset(out, 0, sizeof *out);
    size_t W8, H8;
    orc_padded_size(W, H, &W8, &H8);
    /* The reference mirrors with `oldWidth - diff` in size_t (utils.cpp:215,226):
     * a pad wider than the image underflows (UB).  Refuse those sizes. */
    if (W8 - W > W || H8 - H > H) return ORC_E_ARG;
    const size_t N = W8 * H8 / 64;
    out->W8 = W8;
    out->H8 = H8;
    out->n_blocks = N;

    uint8_t *img = (uint8_t *)malloc(W * H * 3);
    uint8_t *pad = (uint8_t *)malloc(W8 * H8 * 3);
    int32_t *zig = (int32_t *)malloc(N * 3 * 64 * sizeof(int32_t));
    if (!img || !pad || !zig) {
        free(img), free(pad), free(zig);
        return ORC_E_CAPACITY;
    }
    memcpy(img, rgb, W * H * 3);

    double t0 = now_us();
    orc_csc(img, W * H);
    out->stage_us[0] = now_us() - t0;
    if (keep & ORC_KEEP_U8_STAGES) {
        out->csc = (uint8_t *)malloc(W * H * 3);
        memcpy(out->csc, img, W * H * 3);
    }
    t0 = now_us();
    if (cds_on) orc_cds(img, W, H);
    out->stage_us[1] = now_us() - t0;
    if (keep & ORC_KEEP_U8_STAGES) {
        out->cds = (uint8_t *)malloc(W * H * 3);
        memcpy(out->cds, img, W * H * 3);
    }
    t0 = now_us();
    orc_pad(img, W, H, pad, W8, H8);
    out->stage_us[2] = now_us() - t0;
    if (keep & ORC_KEEP_U8_STAGES) {
        out->padded = (uint8_t *)malloc(W8 * H8 * 3);
        memcpy(out->padded, pad, W8 * H8 * 3);
    }
    free(img);
    if (keep & ORC_KEEP_DCT) out->dct = (double *)malloc(W8 * H8 * 3 * sizeof(double));

    /* Per block and channel: level shift (quirk Q4: -128 on all three channels,
     * utils.cpp:190-196), transform, quantise, linearise + zig-zag (quirk Q7:
     * row = chan*N + by*(W8/8)+bx, utils.cpp:485-493). */
    uint8_t zz[64];
    orc_zigzag_order(zz);
    const size_t bw = W8 / 8;
    double t_dct = 0, t_q = 0, t_zz = 0;
    for (size_t by = 0; by < H8 / 8; ++by)
        for (size_t bx = 0; bx < bw; ++bx)
            for (int c = 0; c < 3; ++c) {
                double P[64];
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x)
                        P[y * 8 + x] =
                            (double)pad[3 * ((by * 8 + y) * W8 + bx * 8 + x) + c] - 128.0;
                t0 = now_us();
                orc_dct_block(P);
                t_dct += now_us() - t0;
                if (out->dct)
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x)
                            out->dct[3 * ((by * 8 + y) * W8 + bx * 8 + x) + c] = P[y * 8 + x];
                t0 = now_us();
                orc_quant_block(P, c == 0 ? qlum : qchrom);
                t_q += now_us() - t0;
                t0 = now_us();
                int32_t *row = zig + ((size_t)c * N + by * bw + bx) * 64;
                for (int k = 0; k < 64; ++k) row[k] = (int32_t)P[zz[k]];
                t_zz += now_us() - t0;
            }
    out->stage_us[4] = t_dct;
    out->stage_us[5] = t_q;
    out->stage_us[6] = t_zz;
    free(pad);

    if (keep & ORC_KEEP_UNIT_BITS) out->unit_bits = (uint32_t *)malloc(N * 3 * sizeof(uint32_t));
    t0 = now_us();
    int e = orc_entropy(zig, N, &out->bits, &out->bits_bytes, &out->n_bits, out->unit_bits);
    out->stage_us[8] = now_us() - t0;
    if (keep & ORC_KEEP_ZIGZAG)
        out->zigzag = zig;
    else
        free(zig);
    if (e) {
        if (keep & ORC_KEEP_ZIGZAG) out->zigzag = 0, free(zig);
        orc_result_free(out);
        return e;
    }
    return ORC_OK;
}

void orc_result_free(orc_result *r) {
    if (!r) return;
    free(r->bits), free(r->zigzag), free(r->unit_bits);
    free(r->csc), free(r->cds), free(r->padded), free(r->dct);
    memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------ standard mode */

static int std_code(int table, int run, int size, uint32_t *code) {
    build_tables();
    if (table < 2 ? (run != 0 || size > 11) : (run > 15 || size > 10)) return -1;
    int rs = (run << 4) | size;
    if (!HTS[table].len[rs]) return -1;
    *code = HTS[table].code[rs];
    return HTS[table].len[rs];
}

static int std_symbol(struct orc_bitsink *sink, int table, int run, int32_t v, int *nbits) {
    int size = bit_size(v);
    uint32_t code;
    int len = std_code(table, run, size, &code);
    if (len < 0) return ORC_E_CATEGORY;
    *nbits += len + size;
    if (sink) {
        int e = sink_put(sink, code, len);
        if (!e && size) e = sink_put(sink, value_bits(v, size), size);
        if (e) return e;
    }
    return 0;
}

static int std_unit(const int32_t zz[64], int32_t dc_diff, int chroma, struct orc_bitsink *sink) {
    int nbits = 0, e;
    if ((e = std_symbol(sink, chroma ? 1 : 0, 0, dc_diff, &nbits))) return e;
    int run = 0, tab = chroma ? 3 : 2;
    for (int i = 1; i < 64; ++i) {
        if (zz[i] == 0) {
            ++run;
            continue;
        }
        while (run >= 16) {
            if ((e = std_symbol(sink, tab, 15, 0, &nbits))) return e;
            run -= 16;
        }
        if ((e = std_symbol(sink, tab, run, zz[i], &nbits))) return e;
        run = 0;
    }
    if (run > 0 && (e = std_symbol(sink, tab, 0, 0, &nbits))) return e; /* EOB only if zeros remain */
    return nbits;
}

/* One 8x8 block of level-shifted samples -> quantised zig-zag row.  Standard mode is DEFINED by this arithmetic (it is
 * not a behaviour of the reference): the DCT-II as a fixed-point map -- `dct` = round(D * 2^39), of which the top three
 * balanced base-256 digits are kept, i.e. 23 fractional bits -- evaluated exactly in integers; then the quotient by Q in
 * single precision: acc4 and t = 256 acc3 + acc2 converted to float (round to nearest even), fv = fmaf(acc4, 2^16, t),
 * zf = fv * (float)(2^-23 / Q), and the nearest integer of zf through the 1.5 * 2^23 addition (ties to even).  Row 0 is
 * exactly sum / 8: q0 = round-half-away(sum / (8 Q)) in integers.  The HIP path (screen_quantise<STD>,
 * jpeg_screen_devfn.h) performs the same operations in the same order, so the two agree bit for bit. */
static void std_digits(int64_t x, int d[5]) { /* balanced base-256 digits, least significant first (tools/gen_screen_tables.py) */
    for (int l = 0; l < 5; ++l) {
        int64_t m = (x + 128) % 256;
        if (m < 0) m += 256;
        const int64_t r = m - 128;
        d[l] = (int)r;
        x = (x - r) / 256;
    }
}
/* dig: [3][64][64] = digits 4, 3, 2 of the table (std_digit_table) */
static void std_digit_table(const int64_t *dct, int32_t *dig) {
    for (int i = 0; i < 4096; ++i) {
        int d[5];
        std_digits(dct[i], d);
        dig[i] = d[4], dig[4096 + i] = d[3], dig[8192 + i] = d[2];
    }
}
static void std_block(const int32_t p[64], const uint32_t *q, const uint8_t zz[64], const int32_t *dig,
                      int32_t *row) {
    {
        int64_t sum = 0;
        for (int s = 0; s < 64; ++s) sum += p[s];
        const int64_t Q0 = q[zz[0]], a = sum < 0 ? -sum : sum, n = (a + 4 * Q0) / (8 * Q0);
        row[0] = (int32_t)(sum < 0 ? -n : n);
    }
    for (int R = 1; R < 64; ++R) {
        int32_t acc4 = 0, acc3 = 0, acc2 = 0;
        for (int s = 0; s < 64; ++s)
            acc4 += dig[R * 64 + s] * p[s], acc3 += dig[4096 + R * 64 + s] * p[s], acc2 += dig[8192 + R * 64 + s] * p[s];
        const int32_t t = acc3 * 256 + acc2;
        const float fv = fmaf((float)acc4, 65536.0f, (float)t);
        const float sf = (float)(ldexp(1.0, 16 - 39) / (double)q[zz[R]]);
        const float zf = fv * sf;
        volatile float a = zf + 12582912.0f; /* one rounding, in single precision */
        union {
            float f;
            uint32_t u;
        } b;
        b.f = a;
        row[R] = (int32_t)(int16_t)(b.u & 0xffffu); /* the float's bits are 0x4B400000 + q */
    }
}

/* Standard-mode colour conversion: DEFINED as 15-bit fixed point, the form libjpeg's jccolor.c uses with 16 bits
 * (Y = (c.RGB + half) >> bits; chroma with the "half - 1" constant so that 255.5 cannot round to 256; every row of
 * coefficients sums to its exact total, so a grey pixel stays grey):
 *     Y  = ( 9798 R + 19235 G +  3735 B + 16384) >> 15
 *     Cb = ((-5529 R - 10855 G + 16384 B + 16383) >> 15) + 128         (arithmetic shift = floor)
 *     Cr = ((16384 R - 13720 G -  2664 B + 16383) >> 15) + 128
 * and 4:2:0 chroma as the box filter of the same linear form over the 2x2 quad, rounded ONCE (coefficients c/4 at 16 bits):
 *     Cb = ((sum_quad(-2765 R - 5427 G + 8192 B) + 32767) >> 16) + 128
 *     Cr = ((sum_quad( 8192 R - 6860 G - 1332 B) + 32767) >> 16) + 128
 * The HIP path evaluates exactly these integers (on the matrix units where a tile lies inside the image). */
static const int kStdY[3] = {9798, 19235, 3735};
static const int kStdC[2][3] = {{-5529, -10855, 16384}, {16384, -13720, -2664}};
static const int kStdC420[2][3] = {{-2765, -5427, 8192}, {8192, -6860, -1332}};
static int32_t std_floor_shift(int64_t v, int bits) { /* floor(v / 2^bits) without relying on >> of negatives */
    const int64_t d = (int64_t)1 << bits;
    return (int32_t)(v >= 0 ? v / d : -((-v + d - 1) / d));
}
void orc_std_csc(const uint8_t *rgb, size_t n, uint8_t *ycc) { /* per-pixel form (4:4:4 and every luma sample) */
    for (size_t i = 0; i < n; ++i) {
        const int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        ycc[3 * i] = (uint8_t)std_floor_shift((int64_t)kStdY[0] * r + kStdY[1] * g + kStdY[2] * b + 16384, 15);
        for (int c = 0; c < 2; ++c)
            ycc[3 * i + 1 + c] =
                (uint8_t)(std_floor_shift((int64_t)kStdC[c][0] * r + kStdC[c][1] * g + kStdC[c][2] * b + 16383, 15) + 128);
    }
}

/* mirror padding of the RGB image, colour conversion (4:2:0: chroma straight from the padded RGB quads), fixed-point DCT +
 * quantisation.  Returns the malloc'd zig-zag rows (layout: see orc_std_encode) or NULL for a refused size. */
static int32_t *std_transform(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
                              const uint32_t qchrom[64], const int64_t *dct, int subsample, size_t *Wp_out,
                              size_t *Hp_out, size_t *M_out) {
    const size_t A = subsample ? 16 : 8; /* MCU edge in pixels */
    const size_t Wp = (W + A - 1) / A * A, Hp = (H + A - 1) / A * A;
    if (Wp - W > W || Hp - H > H) return NULL;
    const size_t M = (Wp / A) * (Hp / A);            /* MCUs */
    const size_t units = subsample ? 6 * M : 3 * M;  /* 8x8 blocks in the scan */
    *Wp_out = Wp, *Hp_out = Hp, *M_out = M;
    uint8_t *prgb = (uint8_t *)malloc(Wp * Hp * 3), *pad = (uint8_t *)malloc(Wp * Hp * 3);
    int32_t *zig = (int32_t *)malloc(units * 64 * sizeof(int32_t));
    orc_pad(rgb, W, H, prgb, Wp, Hp); /* pixel-wise conversion commutes with mirroring */
    orc_std_csc(prgb, Wp * Hp, pad);
    uint8_t zz[64];
    orc_zigzag_order(zz);
    int32_t *dig = (int32_t *)malloc(3 * 4096 * sizeof(int32_t));
    std_digit_table(dct, dig);
    /* Row order of `zig`.  4:4:4: chan*M + block (like the strict path).  4:2:0: the luma blocks in
     * scan order (4*mcu + k, k = 2*(row in MCU) + (column in MCU)), then Cb (4M + mcu), then Cr (5M + mcu). */
    if (!subsample) {
        const size_t bw = Wp / 8;
        for (size_t by = 0; by < Hp / 8; ++by)
            for (size_t bx = 0; bx < bw; ++bx)
                for (int c = 0; c < 3; ++c) {
                    int32_t p[64];
                    for (int s = 0; s < 64; ++s)
                        p[s] = (int32_t)pad[3 * ((by * 8 + s / 8) * Wp + bx * 8 + s % 8) + c] - 128;
                    std_block(p, c == 0 ? qlum : qchrom, zz, dig, zig + ((size_t)c * M + by * bw + bx) * 64);
                }
    } else {
        const size_t mw = Wp / 16;
        for (size_t my = 0; my < Hp / 16; ++my)
            for (size_t mx = 0; mx < mw; ++mx) {
                const size_t mcu = my * mw + mx;
                int32_t p[64];
                for (int k = 0; k < 4; ++k) {
                    const size_t x0 = mx * 16 + (k & 1) * 8, y0 = my * 16 + (k >> 1) * 8;
                    for (int s = 0; s < 64; ++s) p[s] = (int32_t)pad[3 * ((y0 + s / 8) * Wp + x0 + s % 8)] - 128;
                    std_block(p, qlum, zz, dig, zig + (4 * mcu + k) * 64);
                }
                for (int c = 1; c < 3; ++c) {
                    /* box filter of the linear form over the 2x2 quad of (padded) RGB pixels, rounded once */
                    for (int s = 0; s < 64; ++s) {
                        const size_t x = mx * 16 + 2 * (s % 8), y = my * 16 + 2 * (s / 8);
                        int64_t acc = 32767;
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                const uint8_t *px = prgb + 3 * ((y + dy) * Wp + x + dx);
                                acc += (int64_t)kStdC420[c - 1][0] * px[0] + kStdC420[c - 1][1] * px[1] + kStdC420[c - 1][2] * px[2];
                            }
                        p[s] = std_floor_shift(acc, 16); /* = Cb - 128 */
                    }
                    std_block(p, qchrom, zz, dig, zig + ((size_t)(3 + c) * M + mcu) * 64);
                }
            }
    }
    free(pad);
    free(prgb);
    free(dig);
    return zig;
}

/* one unit's place in the row array for MCU i, block k of the MCU (see orc_std_encode) */
static const int32_t *std_row(const int32_t *zig, size_t M, int subsample, size_t i, int k, int *comp) {
    const int c = subsample ? (k < 4 ? 0 : k - 3) : k;
    *comp = c;
    if (subsample) return k < 4 ? zig + (4 * i + k) * 64 : zig + ((size_t)(3 + c) * M + i) * 64;
    return zig + (i + M * (size_t)c) * 64;
}

int orc_std_encode(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
                   const uint32_t qchrom[64], const int64_t *dct, int subsample, int keep, orc_result *out) {
    if (!rgb || !out || !dct || W == 0 || H == 0 || subsample < 0 || subsample > 1) return ORC_E_ARG;
    memset(out, 0, sizeof *out);
    size_t Wp, Hp, M;
    int32_t *zig = std_transform(rgb, W, H, qlum, qchrom, dct, subsample, &Wp, &Hp, &M);
    if (!zig) return ORC_E_ARG;
    const size_t units = subsample ? 6 * M : 3 * M;
    out->W8 = Wp, out->H8 = Hp, out->n_blocks = M;
    if (keep & ORC_KEEP_UNIT_BITS) out->unit_bits = (uint32_t *)malloc(units * sizeof(uint32_t));
    struct orc_bitsink sink = {0, 0, 0};
    int32_t pred[3] = {0, 0, 0};
    int err = 0;
    size_t u = 0; /* unit index in scan order */
    const int per_mcu = subsample ? 6 : 3;
    for (size_t i = 0; i < M && !err; ++i)
        for (int k = 0; k < per_mcu; ++k, ++u) {
            int c;
            const int32_t *r = std_row(zig, M, subsample, i, k, &c);
            int32_t diff = r[0] - pred[c];
            pred[c] = r[0];
            int n = std_unit(r, diff, c != 0, &sink);
            if (n < 0) {
                err = n;
                break;
            }
            if (out->unit_bits) out->unit_bits[u] = (uint32_t)n;
        }
    if (err) {
        free(sink.buf), free(zig);
        orc_result_free(out);
        return err;
    }
    out->bits = sink.buf;
    out->n_bits = sink.nbit;
    out->bits_bytes = (size_t)((sink.nbit + 7) / 8);
    if (keep & ORC_KEEP_ZIGZAG)
        out->zigzag = zig;
    else
        free(zig);
    return ORC_OK;
}

void orc_lcg_fill(uint8_t *dst, size_t nbytes, uint32_t seed) {
    uint32_t s = seed;
    for (size_t k = 0; k < nbytes; ++k) {
        s = s * 1664525u + 1013904223u;
        dst[k] = (uint8_t)(s >> 24);
    }
}

/* ------------------------------------------------------------ JFIF framing */
/* Build-defined: the reference writes no container. */

typedef struct {
    uint8_t *p;
    size_t n, cap;
    int ovf;
} wr;
static void w8(wr *w, unsigned v) {
    if (w->n < w->cap)
        w->p[w->n] = (uint8_t)v;
    else
        w->ovf = 1;
    w->n++;
}
static void w16(wr *w, unsigned v) {
    w8(w, v >> 8);
    w8(w, v & 255);
}
static void dht(wr *w, int cls_id, const uint8_t bits[16], const uint8_t *val, int nval) {
    w16(w, 0xFFC4);
    w16(w, 2 + 1 + 16 + nval);
    w8(w, cls_id);
    for (int i = 0; i < 16; ++i) w8(w, bits[i]);
    for (int i = 0; i < nval; ++i) w8(w, val[i]);
}

long orc_jfif_frame(const uint8_t *bits, uint64_t n_bits, size_t W, size_t H,
                    const uint32_t qlum[64], const uint32_t qchrom[64], uint8_t *out, size_t cap) {
    return orc_jfif_frame_s(bits, n_bits, W, H, qlum, qchrom, 0, out, cap);
}

/* subsample 1: luma sampling factors 2x2 (4:2:0 MCUs), else 1x1 */
static void jfif_header(wr *wp, size_t W, size_t H, const uint32_t qlum[64], const uint32_t qchrom[64],
                        int subsample, unsigned restart_interval);

long orc_jfif_frame_s(const uint8_t *bits, uint64_t n_bits, size_t W, size_t H, const uint32_t qlum[64],
                      const uint32_t qchrom[64], int subsample, uint8_t *out, size_t cap) {
    wr w = {out, 0, cap, 0};
    jfif_header(&w, W, H, qlum, qchrom, subsample, 0);
    size_t nb = (size_t)((n_bits + 7) / 8);
    for (size_t i = 0; i < nb; ++i) {
        unsigned b = bits[i];
        if (i == nb - 1 && (n_bits & 7)) b |= 0xFFu >> (n_bits & 7); /* pad with 1s */
        w8(&w, b);
        if (b == 0xFF) w8(&w, 0);
    }
    w16(&w, 0xFFD9);
    return w.ovf ? ORC_E_CAPACITY : (long)w.n;
}

/* Standard mode with restart markers: DRI = `interval` MCUs; every interval starts with zero DC
 * predictors, is padded to a byte with 1s and (except the last) followed by RSTm, m = index mod 8.
 * Writes the whole file. */
long orc_std_jfif_restart(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
                          const uint32_t qchrom[64], const int64_t *dct, int subsample, unsigned interval,
                          uint8_t *out, size_t cap) {
    if (!rgb || !dct || !out || W == 0 || H == 0 || interval == 0 || interval > 65535) return ORC_E_ARG;
    size_t Wp, Hp, M;
    int32_t *zig = std_transform(rgb, W, H, qlum, qchrom, dct, subsample, &Wp, &Hp, &M);
    if (!zig) return ORC_E_ARG;
    wr w = {out, 0, cap, 0};
    jfif_header(&w, W, H, qlum, qchrom, subsample, interval);
    const int per_mcu = subsample ? 6 : 3;
    int err = 0;
    for (size_t i0 = 0, idx = 0; i0 < M && !err; i0 += interval, ++idx) {
        struct orc_bitsink sink = {0, 0, 0};
        int32_t pred[3] = {0, 0, 0};
        const size_t i1 = i0 + interval < M ? i0 + interval : M;
        for (size_t i = i0; i < i1 && !err; ++i)
            for (int k = 0; k < per_mcu; ++k) {
                int c;
                const int32_t *r = std_row(zig, M, subsample, i, k, &c);
                int32_t diff = r[0] - pred[c];
                pred[c] = r[0];
                if (std_unit(r, diff, c != 0, &sink) < 0) {
                    err = ORC_E_CATEGORY;
                    break;
                }
            }
        const size_t nb = (size_t)((sink.nbit + 7) / 8);
        for (size_t i = 0; i < nb; ++i) {
            unsigned b = sink.buf[i];
            if (i == nb - 1 && (sink.nbit & 7)) b |= 0xFFu >> (sink.nbit & 7);
            w8(&w, b);
            if (b == 0xFF) w8(&w, 0);
        }
        free(sink.buf);
        if (i1 < M) w16(&w, 0xFFD0 + (unsigned)(idx & 7));
    }
    free(zig);
    w16(&w, 0xFFD9);
    if (err) return err;
    return w.ovf ? ORC_E_CAPACITY : (long)w.n;
}

static void jfif_header(wr *wp, size_t W, size_t H, const uint32_t qlum[64], const uint32_t qchrom[64],
                        int subsample, unsigned restart_interval) {
    wr w = *wp;
    uint8_t zz[64];
    orc_zigzag_order(zz);
    w16(&w, 0xFFD8);
    w16(&w, 0xFFE0), w16(&w, 16);
    w8(&w, 'J'), w8(&w, 'F'), w8(&w, 'I'), w8(&w, 'F'), w8(&w, 0);
    w16(&w, 0x0101), w8(&w, 0), w16(&w, 1), w16(&w, 1), w8(&w, 0), w8(&w, 0);
    for (int t = 0; t < 2; ++t) {
        const uint32_t *q = t ? qchrom : qlum;
        w16(&w, 0xFFDB), w16(&w, 67), w8(&w, t);
        for (int k = 0; k < 64; ++k) w8(&w, q[zz[k]] > 255 ? 255 : q[zz[k]]);
    }
    w16(&w, 0xFFC0), w16(&w, 17), w8(&w, 8), w16(&w, (unsigned)H), w16(&w, (unsigned)W), w8(&w, 3);
    w8(&w, 1), w8(&w, subsample ? 0x22 : 0x11), w8(&w, 0);
    w8(&w, 2), w8(&w, 0x11), w8(&w, 1);
    w8(&w, 3), w8(&w, 0x11), w8(&w, 1);
    dht(&w, 0x00, BITS_DC_L, VAL_DC, 12);
    dht(&w, 0x10, BITS_AC_L, VAL_AC_L, 162);
    dht(&w, 0x01, BITS_DC_C, VAL_DC, 12);
    dht(&w, 0x11, BITS_AC_C, VAL_AC_C, 162);
    if (restart_interval) w16(&w, 0xFFDD), w16(&w, 4), w16(&w, restart_interval);
    w16(&w, 0xFFDA), w16(&w, 12), w8(&w, 3);
    w8(&w, 1), w8(&w, 0x00), w8(&w, 2), w8(&w, 0x11), w8(&w, 3), w8(&w, 0x11);
    w8(&w, 0), w8(&w, 63), w8(&w, 0);
    *wp = w;
}
