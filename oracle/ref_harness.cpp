// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Harness around the *real* reference stage library.  It is compiled together
// with /root/reference/src/utils.cpp (in place, unmodified, never copied) into
// oracle/_ref/libjpegref.so by oracle/Makefile.  It replays the call order of
// JpegEncoderHost (reference src/OpenCLProject_JpegEncoder.cpp:59-225) with heap
// buffers instead of the reference's stack VLAs (:190-191) and without the
// debug PPM dumps, and hands every intermediate artefact back to the caller
// so that oracle/jpeg_oracle.c (the restatement) and the HIP path can be
// checked stage by stage.
//
// Everything below is this repository's own code; the reference supplies only
// the functions it calls (performCSC ... HuffmanEncoder) and the tables in
// utils.hpp / huffman.hpp.
#include <pthread.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <OpenCL/cl-patched.hpp>  // the reference's own header (utils.hpp needs cl_uint)
#include "huffman.hpp"
#include "utils.hpp"

// Declared in the reference's utils.cpp but with a mismatching prototype in
// utils.hpp (utils.hpp:128 vs utils.cpp:572) -- we only go through performRLE.

namespace {

struct RefRun {
    size_t W = 0, H = 0, W8 = 0, H8 = 0;
    std::vector<uint8_t> csc, cds, padded;  // interleaved 3 B/px snapshots
    std::vector<double> dct;                // padded, 3 doubles/px, after performDCT
    std::vector<int32_t> zigzag;            // [3*N][64]
    std::string bits;                       // '0'/'1'
    double us[9] = {0};                     // CSC, CDS, copy, shift, DCT, quant, zigzag, RLE, huffman
};

struct Job {
    const uint8_t *rgb;
    size_t W, H;
    const uint32_t *qlum, *qchrom;
    int cds_on;
    int keep;  // bit0: keep u8 stage snapshots, bit1: keep DCT doubles
    RefRun *out;
};

double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

void *run_job(void *arg) {
    Job *j = static_cast<Job *>(arg);
    RefRun *r = j->out;
    const size_t W = j->W, H = j->H;
    r->W = W;
    r->H = H;

    unsigned int ql[8][8], qc[8][8];
    for (int i = 0; i < 64; ++i) {
        ql[i / 8][i % 8] = j->qlum[i];
        qc[i / 8][i % 8] = j->qchrom[i];
    }

    ppm_t img;
    img.width = W;
    img.height = H;
    img.data = (rgb_pixel_t *)malloc(W * H * sizeof(rgb_pixel_t));
    memcpy(img.data, j->rgb, W * H * 3);

    double t0 = now_us();
    performCSC(&img);
    r->us[0] = now_us() - t0;
    if (j->keep & 1) r->csc.assign((uint8_t *)img.data, (uint8_t *)img.data + W * H * 3);

    t0 = now_us();
    if (j->cds_on) performCDS(&img);
    r->us[1] = now_us() - t0;
    if (j->keep & 1) r->cds.assign((uint8_t *)img.data, (uint8_t *)img.data + W * H * 3);

    size_t W8, H8;
    if (W % 8 == 0 && H % 8 == 0) {
        W8 = W;
        H8 = H;
    } else {
        getNearest8x8ImageSize(W, H, &W8, &H8);
    }
    r->W8 = W8;
    r->H8 = H8;

    ppm_t big;
    big.width = W8;
    big.height = H8;
    big.data = (rgb_pixel_t *)malloc(W8 * H8 * sizeof(rgb_pixel_t));
    t0 = now_us();
    copyToLargerImage(&img, &big);
    r->us[2] = now_us() - t0;
    addReversedPadding(&big, W, H);
    if (j->keep & 1) r->padded.assign((uint8_t *)big.data, (uint8_t *)big.data + W8 * H8 * 3);

    ppm_d_t d;
    d.width = W8;
    d.height = H8;
    d.data = (rgb_pixel_d_t *)malloc(W8 * H8 * sizeof(rgb_pixel_d_t));
    t0 = now_us();
    copyUIntToDoubleImage(&big, &d);
    r->us[2] += now_us() - t0;
    free(img.data);
    free(big.data);

    t0 = now_us();
    substractfromAll(&d, 128.0);
    r->us[3] = now_us() - t0;

    t0 = now_us();
    performDCT(&d);
    r->us[4] = now_us() - t0;
    if (j->keep & 2) r->dct.assign((double *)d.data, (double *)d.data + W8 * H8 * 3);

    t0 = now_us();
    performQuantization(&d, ql, qc);
    r->us[5] = now_us() - t0;

    const size_t N = W8 * H8 / 64;
    const size_t rows = N * 3;
    int(*linear)[64] = (int(*)[64])malloc(rows * 64 * sizeof(int));
    r->zigzag.resize(rows * 64);
    int(*zz)[64] = (int(*)[64])r->zigzag.data();
    t0 = now_us();
    everyMCUisnow2DArray(&d, linear);
    performZigZag(linear, zz, (int)rows);
    r->us[6] = now_us() - t0;
    free(linear);
    free(d.data);

    std::vector<std::vector<int>> rle;
    t0 = now_us();
    performRLE(zz, rle, (int)rows);
    r->us[7] = now_us() - t0;

    t0 = now_us();
    r->bits = HuffmanEncoder(zz, rle, (int)N);
    r->us[8] = now_us() - t0;
    return nullptr;
}

}  // namespace

extern "C" {

// Runs the reference CPU path.  qlum/qchrom: 64 entries, row-major [v][u] as
// utils.hpp:42-62.  Returns an opaque handle (free with ref_free).
void *ref_run(const uint8_t *rgb, size_t W, size_t H, const uint32_t *qlum, const uint32_t *qchrom,
              int cds_on, int keep) {
    RefRun *r = new RefRun();
    Job j{rgb, W, H, qlum, qchrom, cds_on, keep, r};
    // HuffmanEncoder keeps a VLA of N*3 ints on the stack (utils.cpp:661): give
    // the worker a stack that fits the largest config.
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    size_t need = ((W + 7) / 8) * ((H + 7) / 8) * 3 * sizeof(int) + (64u << 20);
    pthread_attr_setstacksize(&attr, need);
    pthread_t th;
    if (pthread_create(&th, &attr, run_job, &j) != 0) {
        delete r;
        return nullptr;
    }
    pthread_join(th, nullptr);
    pthread_attr_destroy(&attr);
    return r;
}

void ref_free(void *h) { delete static_cast<RefRun *>(h); }

void ref_dims(void *h, size_t *W8, size_t *H8) {
    RefRun *r = static_cast<RefRun *>(h);
    *W8 = r->W8;
    *H8 = r->H8;
}
const uint8_t *ref_csc(void *h) { return static_cast<RefRun *>(h)->csc.data(); }
const uint8_t *ref_cds(void *h) { return static_cast<RefRun *>(h)->cds.data(); }
const uint8_t *ref_padded(void *h) { return static_cast<RefRun *>(h)->padded.data(); }
const double *ref_dct(void *h) { return static_cast<RefRun *>(h)->dct.data(); }
const int32_t *ref_zigzag(void *h) { return static_cast<RefRun *>(h)->zigzag.data(); }
uint64_t ref_nbits(void *h) { return static_cast<RefRun *>(h)->bits.size(); }
const char *ref_bits(void *h) { return static_cast<RefRun *>(h)->bits.data(); }
const double *ref_stage_us(void *h) { return static_cast<RefRun *>(h)->us; }

// performCSC alone (utils.cpp:92-110), in place on npix interleaved pixels: used
// for the exhaustive 2^24 colour-conversion check.
void ref_csc_only(uint8_t *px, size_t npix) {
    ppm_t img;
    img.width = npix;
    img.height = 1;
    img.data = (rgb_pixel_t *)px;
    performCSC(&img);
}

// The reference's constant tables, for pinning the restatement's tables.
void ref_quant_tables(uint32_t *qlum, uint32_t *qchrom) {
    for (int i = 0; i < 64; ++i) {
        qlum[i] = quant_mat_lum[i / 8][i % 8];
        qchrom[i] = quant_mat_chrom[i / 8][i % 8];
    }
}

// table: 0 DC luma, 1 DC chroma, 2 AC luma, 3 AC chroma.  Copies the code
// string (NUL terminated, <= 31 chars) for [run][size]; returns its length or
// -1 when out of range.
int ref_huff_code(int table, int run, int size, char *out) {
    const std::string *s = nullptr;
    if (table == 0 && run == 0 && size >= 0 && size < (int)DC_LUMA_HUFF_CODES.size())
        s = &DC_LUMA_HUFF_CODES[size];
    if (table == 1 && run == 0 && size >= 0 && size < (int)DC_CHROMA_HUFF_CODES.size())
        s = &DC_CHROMA_HUFF_CODES[size];
    if (table == 2 && run >= 0 && run < (int)AC_LUMA_HUFF_CODES.size() && size >= 0 &&
        size < (int)AC_LUMA_HUFF_CODES[run].size())
        s = &AC_LUMA_HUFF_CODES[run][size];
    if (table == 3 && run >= 0 && run < (int)AC_CHROMA_HUFF_CODES.size() && size >= 0 &&
        size < (int)AC_CHROMA_HUFF_CODES[run].size())
        s = &AC_CHROMA_HUFF_CODES[run][size];
    if (!s || s->size() > 31) return -1;
    memcpy(out, s->c_str(), s->size() + 1);
    return (int)s->size();
}

// std::cos((2a+1)*k*M_PI/16.0) exactly as utils.cpp:330 forms it, and the
// three scale constants of utils.cpp:318-319,336.
double ref_cos(size_t a, size_t k) { return std::cos((2 * a + 1) * k * M_PI / 16.0); }
double ref_scale(size_t u, size_t v) {
    double alphaU = (u == 0) ? 1.0 / std::sqrt(2) : 1.0;
    double alphaV = (v == 0) ? 1.0 / std::sqrt(2) : 1.0;
    return (alphaU * alphaV / 4.0);
}
}
