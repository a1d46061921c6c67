/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the strict JPEG encode path of
 * rusty-electron/jpeg-encoder-opencl, i.e. of the call sequence of
 * JpegEncoderHost (reference src/OpenCLProject_JpegEncoder.cpp:59-225) over the
 * stage library src/utils.cpp.  Plain C, one thread, written from the behaviour
 * spec in SURVEY.md Appendix A -- not a transliteration.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library, and only as the checker.  The product path (the HIP library
 * behind include/mi355_jpeg.h) never links or calls it.
 *
 * Parity status: PINNED.  The restatement is checked in this repository's
 * container against the real reference (oracle/_ref/libjpegref.so, built from
 * /root/reference/src/utils.cpp in place) stage by stage, and against the
 * golden vectors under tests/golden/ that were generated from that build
 * (tools/make_golden.py).  The reference itself holds no tests or golden
 * vectors for this path.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_OK = 0,
    ORC_E_ARG = -1,
    ORC_E_CAPACITY = -2,
    ORC_E_CATEGORY = -3 /* DC size > 11 or AC size > 10: the reference indexes out of
                           its tables here (huffman.hpp:9-23,45-57; UB) */
};

/* ---- tables --------------------------------------------------------- */
/* C[a][k] = cos((2a+1)k*pi/16) as glibc 2.35 rounds it (utils.cpp:330);
 * scale(u,v) = alpha_u*alpha_v/4.0 (utils.cpp:318-319,336). */
double orc_cos(int a, int k);
double orc_scale(int u, int v);
/* zz[k] = natural index (v*8+u) of the k-th zig-zag coefficient
 * (utils.cpp:539-551). */
void orc_zigzag_order(uint8_t zz[64]);
/* q=50 Annex-K tables, row-major [v][u] (utils.hpp:42-62). */
void orc_quant_q50(uint32_t qlum[64], uint32_t qchrom[64]);
/* Build convention (not pinned by the reference): IJG scaling. */
void orc_quant_ijg(int quality, uint32_t qlum[64], uint32_t qchrom[64]);
/* table: 0 DC luma, 1 DC chroma, 2 AC luma, 3 AC chroma (huffman.hpp).
 * Returns the code length (<= 17), or -1 if the reference has no code
 * ("NULL" string or index out of range).  *code gets the bits, right aligned. */
int orc_huff_code(int table, int run, int size, uint32_t *code);

/* ---- per-stage functions ------------------------------------------- */
/* utils.cpp:92-110, in place on interleaved 3 B/px. */
void orc_csc(uint8_t *px, size_t npix);
/* utils.cpp:113-141, in place, on the UNPADDED image. */
void orc_cds(uint8_t *px, size_t W, size_t H);
/* utils.cpp:184-187 + OpenCLProject_JpegEncoder.cpp:93-98. */
void orc_padded_size(size_t W, size_t H, size_t *W8, size_t *H8);
/* utils.cpp:199-233: copy into the (W8,H8) canvas + mirror pad. */
void orc_pad(const uint8_t *src, size_t W, size_t H, uint8_t *dst, size_t W8, size_t H8);
/* utils.cpp:314-348 for ONE channel of one block: 64 doubles [y*8+x], in place. */
void orc_dct_block(double P[64]);
void orc_dct_blocks(double *P, size_t n);
/* utils.cpp:457-463 for one block: P[v*8+u] = round(P/q[v*8+u]). */
void orc_quant_block(double P[64], const uint32_t q[64]);
/* utils.cpp:572-609 + 656-698 for one unit.  dc_diff already formed.
 * Returns the number of bits or ORC_E_CATEGORY; if sink != NULL the bits are
 * appended through it. */
struct orc_bitsink;
int orc_unit_bits(const int32_t zz[64], int32_t dc_diff, int chroma, struct orc_bitsink *sink);

/* ---- whole path ------------------------------------------------------ */
typedef struct orc_result {
    size_t W8, H8;        /* padded size */
    size_t n_blocks;      /* N = W8*H8/64; units = 3N */
    uint64_t n_bits;      /* scan bits */
    uint8_t *bits;        /* packed MSB-first, zero padded to a byte; malloc'd */
    size_t bits_bytes;
    int32_t *zigzag;      /* [3N][64] in the reference's row order chan*N+block; malloc'd, or NULL */
    uint32_t *unit_bits;  /* [3N] bit count per unit, scan order 3*block+chan; malloc'd, or NULL */
    uint8_t *csc, *cds, *padded; /* stage snapshots (3 B/px) if requested, else NULL */
    double *dct;          /* padded 3 doubles/px after the transform, if requested */
    double stage_us[9];   /* CSC, CDS, copy, shift, DCT, quant, zigzag, RLE(0), huffman */
} orc_result;

enum { ORC_KEEP_ZIGZAG = 1, ORC_KEEP_UNIT_BITS = 2, ORC_KEEP_U8_STAGES = 4, ORC_KEEP_DCT = 8 };

/* rgb: interleaved W*H*3.  qlum/qchrom row-major [v][u].  cds_on: run the
 * chroma averaging (reference behaviour) or skip it (build convention). */
int orc_encode(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
               const uint32_t qchrom[64], int cds_on, int keep, orc_result *out);
void orc_result_free(orc_result *r);

/* Entropy-code an existing coefficient array (reference row order) only. */
int orc_entropy(const int32_t *zigzag, size_t n_blocks, uint8_t **bits, size_t *bits_bytes,
                uint64_t *n_bits, uint32_t *unit_bits /* [3N] or NULL */);

/* ---- standard mode --------------------------------------------
 * NOT a behaviour of the reference (parity unpinned by it): a decodable baseline
 * JPEG, defined by a fixed sequence of integer and fp32 operations so that the GPU path and
 * this checker agree bit for bit:
 *   samples  15-bit fixed-point colour conversion in libjpeg's form (orc_std_csc:
 *            Y = (9798R+19235G+3735B+16384)>>15, Cb/Cr likewise with the "half - 1"
 *            constant; rows sum exactly, nothing to clamp), mirror padding as in strict mode;
 *   DCT      the mode is DEFINED by its arithmetic (round 3), restated here operation for operation (std_block):
 *            the top three base-256 digits of dct[R][s] (= the true DCT-II with 23 fractional bits; dct = the DCT-II
 *            rounded to 2^-39, rows in zig-zag order: tests/golden/std_dct_q39.i64, tools/gen_screen_tables.py) applied
 *            exactly in integers (acc4, acc3, acc2), t = 256 acc3 + acc2 and acc4 converted to float, fv = fmaf(acc4,
 *            2^16, t), zf = fv * (float)(2^-23/Q), nearest integer of zf with ties to even; coefficient 0 is
 *            round-half-away(sum / (8 Q0)) in integers.  Against round-half-away of the exact 2^-39 quotient this differs by
 *            at most one, and only within 2e-3 of a tie (the contract in include/mi355_jpeg.h);
 *   entropy  Annex-K tables without the reference's seven 17-bit typos, EOB
 *            omitted when coefficient 63 is non-zero.
 * keep: ORC_KEEP_ZIGZAG / ORC_KEEP_UNIT_BITS.
 *   subsample 0: 4:4:4, one Y, Cb, Cr block per 8x8 MCU; zigzag rows chan*M + block.
 *   subsample 1: 4:2:0, 16x16 MCUs (Y00 Y01 Y10 Y11 Cb Cr), image mirror-padded to multiples of 16,
 *            chroma = the same linear form box-filtered over the 2x2 quad of padded RGB pixels and rounded once
 *            (coefficients c/4 at 16 bits, see orc_std_csc's comment); zigzag rows: luma 4*mcu + k,
 *            then Cb at 4M + mcu, Cr at 5M + mcu; n_blocks = M (MCUs); unit_bits in scan order. */
int orc_std_encode(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
                   const uint32_t qchrom[64], const int64_t *dct, int subsample, int keep, orc_result *out);
/* the per-pixel colour conversion of standard mode: n RGB pixels -> n YCbCr pixels */
void orc_std_csc(const uint8_t *rgb, size_t n, uint8_t *ycc);
/* The whole file of standard mode WITH restart markers (DRI = interval MCUs): every interval starts
 * from zero DC predictors, is padded to a byte with 1s and, except the last, followed by RSTm. */
long orc_std_jfif_restart(const uint8_t *rgb, size_t W, size_t H, const uint32_t qlum[64],
                          const uint32_t qchrom[64], const int64_t *dct, int subsample, unsigned interval,
                          uint8_t *out, size_t cap);
long orc_jfif_frame_s(const uint8_t *bits, uint64_t n_bits, size_t W, size_t H, const uint32_t qlum[64],
                      const uint32_t qchrom[64], int subsample, uint8_t *out, size_t cap);

/* Pinned synthetic input of SURVEY §8d: LCG s <- s*1664525+1013904223,
 * byte = s>>24, seed = 1+frame. */
void orc_lcg_fill(uint8_t *dst, size_t nbytes, uint32_t seed);

/* Build-defined JFIF framing; returns bytes written or
 * ORC_E_CAPACITY.  The reference has no container; this is applied identically
 * to oracle bits and to GPU bits. */
long orc_jfif_frame(const uint8_t *bits, uint64_t n_bits, size_t W, size_t H,
                    const uint32_t qlum[64], const uint32_t qchrom[64], uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
