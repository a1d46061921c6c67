// Constant tables of the strict encode path, shared by host and device code of
// the product library.  (The test oracle keeps its own copy on purpose.)
//
//  * kCosBits: cos((2a+1)*k*pi/16) exactly as glibc 2.35 rounds it for the
//    reference's argument expression (reference src/utils.cpp:330).  The table is
//    NOT sign-symmetric to the last bit, so it is shipped as data (SURVEY.md
//    Appendix B) and never recomputed on the device.
//  * scale constants alpha_u*alpha_v/4.0 (utils.cpp:318-319,336).
//  * zig-zag order (utils.cpp:539-551 walks the 15 anti-diagonals).
//  * ITU-T T.81 Annex K quantisation / Huffman specifications, which are what
//    utils.hpp:42-62 and huffman.hpp spell out.
#pragma once
#include <stdint.h>

namespace mi355 {

#define MI355_COS_TABLE                                                                          \
    {                                                                                            \
        {0x1.0000000000000p+0, 0x1.f6297cff75cb0p-1, 0x1.d906bcf328d46p-1, 0x1.a9b66290ea1a3p-1, \
         0x1.6a09e667f3bcdp-1, 0x1.1c73b39ae68c9p-1, 0x1.87de2a6aea964p-2, 0x1.8f8b83c69a60dp-3}, \
        {0x1.0000000000000p+0, 0x1.a9b66290ea1a3p-1, 0x1.87de2a6aea964p-2, -0x1.8f8b83c69a608p-3, \
         -0x1.6a09e667f3bccp-1, -0x1.f6297cff75cb0p-1, -0x1.d906bcf328d47p-1,                     \
         -0x1.1c73b39ae68c8p-1},                                                                 \
        {0x1.0000000000000p+0, 0x1.1c73b39ae68c9p-1, -0x1.87de2a6aea962p-2, -0x1.f6297cff75cb0p-1, \
         -0x1.6a09e667f3bcep-1, 0x1.8f8b83c69a60cp-3, 0x1.d906bcf328d44p-1, 0x1.a9b66290ea1a5p-1}, \
        {0x1.0000000000000p+0, 0x1.8f8b83c69a60dp-3, -0x1.d906bcf328d46p-1, -0x1.1c73b39ae68c8p-1, \
         0x1.6a09e667f3bcbp-1, 0x1.a9b66290ea1a5p-1, -0x1.87de2a6aea965p-2, -0x1.f6297cff75cb2p-1}, \
        {0x1.0000000000000p+0, -0x1.8f8b83c69a608p-3, -0x1.d906bcf328d47p-1, 0x1.1c73b39ae68c5p-1, \
         0x1.6a09e667f3bcep-1, -0x1.a9b66290ea1a2p-1, -0x1.87de2a6aea971p-2, 0x1.f6297cff75cb0p-1}, \
        {0x1.0000000000000p+0, -0x1.1c73b39ae68c6p-1, -0x1.87de2a6aea96dp-2, 0x1.f6297cff75cb0p-1, \
         -0x1.6a09e667f3bc5p-1, -0x1.8f8b83c69a602p-3, 0x1.d906bcf328d46p-1,                      \
         -0x1.a9b66290ea1a1p-1},                                                                 \
        {0x1.0000000000000p+0, -0x1.a9b66290ea1a4p-1, 0x1.87de2a6aea967p-2, 0x1.8f8b83c69a61dp-3, \
         -0x1.6a09e667f3bc9p-1, 0x1.f6297cff75cb2p-1, -0x1.d906bcf328d43p-1, 0x1.1c73b39ae68c2p-1}, \
        {0x1.0000000000000p+0, -0x1.f6297cff75cb0p-1, 0x1.d906bcf328d44p-1, -0x1.a9b66290ea1a2p-1, \
         0x1.6a09e667f3bc4p-1, -0x1.1c73b39ae68c2p-1, 0x1.87de2a6aea95fp-2,                       \
         -0x1.8f8b83c69a616p-3},                                                                 \
    }

constexpr double kScale00 = 0x1.ffffffffffffep-4;  // u == 0 && v == 0
constexpr double kScale0X = 0x1.6a09e667f3bccp-3;  // exactly one of u, v is 0
constexpr double kScaleXX = 0x1p-2;                // neither

// Natural index (v*8+u) of the k-th zig-zag coefficient.
#define MI355_ZIGZAG_TABLE                                                                      \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,    \
     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,    \
     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63}

// Colour conversion constants (utils.cpp:106-108), per output channel:
// value = (uint8_t)(((k0*R + k1*G) + k2*B) + k3) in doubles, left to right.
// "a - c*G" in the source equals "a + (-c)*G" bit for bit, and Y's missing
// "+128" is written "+0.0" (Y >= +0, so the sum is unchanged).
#define MI355_CSC_TABLE                              \
    {                                                \
        {0.299, 0.587, 0.114, 0.0},                  \
        {-0.168736, -0.331264, 0.5, 128.0},          \
        {0.5, -0.418688, -0.081312, 128.0},          \
    }

// Standard mode (not a behaviour of the reference): colour conversion DEFINED in 15-bit fixed point, libjpeg's form --
//     Y  = ( 9798 R + 19235 G +  3735 B + 16384) >> 15
//     Cb = ((-5529 R - 10855 G + 16384 B + 16383) >> 15) + 128      (arithmetic shift; "half - 1": 255.5 cannot become 256)
//     Cr = ((16384 R - 13720 G -  2664 B + 16383) >> 15) + 128
// every row sums exactly (2^15, 0, 0): nothing to clamp, greys stay grey, and with x' = x - 128 the same formulas give
// the level-shifted samples without an offset term.  4:2:0 chroma: the same linear form box-filtered over the 2x2 quad
// and rounded ONCE, coefficients c / 4 at 16 bits --
//     Cb = ((sum_quad(-2765 R - 5427 G + 8192 B) + 32767) >> 16) + 128,  Cr = ((sum_quad(8192 R - 6860 G - 1332 B) + 32767) >> 16) + 128.
constexpr int kStdCsc[3][3] = {{9798, 19235, 3735}, {-5529, -10855, 16384}, {16384, -13720, -2664}};
constexpr int kStdCsc420[2][3] = {{-2765, -5427, 8192}, {8192, -6860, -1332}};
// Fragment sets of the per-pixel conversion for the matrix units (built by the host, mi355_jpeg.cpp: upload_csc_frag;
// consumed by jpeg_screen_devfn.h: std_rowpair_mfma): set ((chan * 2 + half) * 2 + digit), 12 sets.
constexpr int kCscSets = 12;
// Strict mode (the reference's conversion, utils.cpp:92-110) on the matrix units: the integer numerators of
// jpeg_screen_devfn.h's csc_int -- Y: 299 R + 587 G + 114 B; Cb, Cr: the numerators over 1e6 divided by 32 -- as 6 more
// fragment sets of the same shape, sets kCscSets + chan * 2 + digit: four pixels at chunk bytes 3 r .. 3 r + 2 (both halves
// of a block row are presented that way: {w0, w1, w2, -} and {w3, w4, w5, -}, the fourth dword has no coefficients).  The
// constant term -- 4000000 for chroma, and the 128 * sum of the coefficients that the x - 128 operand bytes take away:
// 128000 for luma, 0 for chroma, whose rows sum to zero -- enters through the accumulator input (kCscStrictC).
constexpr int kCscStrict[3][3] = {{299, 587, 114}, {-5273, -10352, 15625}, {15625, -13084, -2541}};
constexpr int kCscStrictC[3] = {128000, 4000000, 4000000};
constexpr int kCscStrictSets = 6;

}  // namespace mi355
