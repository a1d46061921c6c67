// Implementation of the utils.hpp-shaped host API on top of the C ABI (no CPU compute).
#include "mi355_utils.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

// ---- tables ------------------------------------------------------------------------
// ITU-T T.81 Annex K.1 (what utils.hpp:42-62 holds).
const unsigned int quant_mat_lum[8][8] = {{16, 11, 10, 16, 24, 40, 51, 61},     {12, 12, 14, 19, 26, 58, 60, 55},
                                          {14, 13, 16, 24, 40, 57, 69, 56},     {14, 17, 22, 29, 51, 87, 80, 62},
                                          {18, 22, 37, 56, 68, 109, 103, 77},   {24, 35, 55, 64, 81, 104, 113, 92},
                                          {49, 64, 78, 87, 103, 121, 120, 101}, {72, 92, 95, 98, 112, 100, 103, 99}};
const unsigned int quant_mat_chrom[8][8] = {{17, 18, 24, 47, 99, 99, 99, 99}, {18, 21, 26, 66, 99, 99, 99, 99},
                                            {24, 26, 56, 99, 99, 99, 99, 99}, {47, 66, 99, 99, 99, 99, 99, 99},
                                            {99, 99, 99, 99, 99, 99, 99, 99}, {99, 99, 99, 99, 99, 99, 99, 99},
                                            {99, 99, 99, 99, 99, 99, 99, 99}, {99, 99, 99, 99, 99, 99, 99, 99}};

namespace {

mi355_jpeg_ctx* g_ctx = nullptr;
int g_device = 0, g_quality = 50;
unsigned g_mode = 0;  // MI355_F_STANDARD [| MI355_F_420], or 0 = strict

unsigned path_flags(bool cds) { return g_mode ? g_mode : (cds ? MI355_F_CDS : 0u); }

int fail(int rc) {
    std::cout << "mi355-jpeg: " << mi355_jpeg_strerror(rc) << std::endl;
    return 1;
}

int ensure_ctx() {
    if (g_ctx) return 0;
    int rc = mi355_jpeg_create(g_device, &g_ctx);
    if (rc) return fail(rc);
    if (g_quality != 50 && (rc = mi355_jpeg_set_quality(g_ctx, g_quality))) return fail(rc);
    return 0;
}

// The code tables as the strings huffman.hpp spells out, generated from the library's
// reference tables (which include the seven 17-bit entries).
std::string bits_of(uint32_t code, int len) {
    std::string s((size_t)len, '0');
    for (int i = 0; i < len; ++i)
        if ((code >> (len - 1 - i)) & 1u) s[(size_t)i] = '1';
    return s;
}
mi355_huff_table ref_table(int table) {
    mi355_huff_table t;
    memset(&t, 0, sizeof t);
    mi355_jpeg_reference_huffman(table, &t);  // host only: no device needed
    return t;
}
std::vector<std::string> dc_strings(int table) {
    mi355_huff_table t = ref_table(table);
    std::vector<std::string> v;
    for (int s = 0; s < 12; ++s) v.push_back(t.len[s] ? bits_of(t.code[s], t.len[s]) : std::string("NULL"));
    return v;
}
std::vector<std::vector<std::string>> ac_strings(int table) {
    mi355_huff_table t = ref_table(table);
    std::vector<std::vector<std::string>> v(16);
    for (int r = 0; r < 16; ++r)
        for (int s = 0; s < 11; ++s) {
            int rs = (r << 4) | s;
            v[(size_t)r].push_back(t.len[rs] ? bits_of(t.code[rs], t.len[rs]) : std::string("NULL"));
        }
    return v;
}

}  // namespace

const std::vector<std::string> DC_LUMA_HUFF_CODES = dc_strings(0);
const std::vector<std::string> DC_CHROMA_HUFF_CODES = dc_strings(1);
const std::vector<std::vector<std::string>> AC_LUMA_HUFF_CODES = ac_strings(2);
const std::vector<std::vector<std::string>> AC_CHROMA_HUFF_CODES = ac_strings(3);

// ---- PPM I/O --------------------------------------------------------------------------
// The reference reader (utils.cpp:11-65) takes exactly "P6\n", optional '#' lines, "<w> <h>\n",
// "255\n", raster.  This one accepts every well-formed binary PPM header of maxval 255 (any
// whitespace between tokens, '#' comments anywhere in the header, one whitespace byte before
// the raster) and checks the raster length; same return convention: 0, or -1 + message on
// stdout.  Reentrant (no strtok).
namespace {
// next header token; skips whitespace and comments.  Returns false at EOF.
bool ppm_token(FILE* fp, char* tok, size_t cap) {
    int ch = fgetc(fp);
    for (;;) {
        while (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\f' || ch == '\v') ch = fgetc(fp);
        if (ch != '#') break;
        while (ch != '\n' && ch != EOF) ch = fgetc(fp);
    }
    size_t n = 0;
    while (ch != EOF && ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r' && ch != '\f' && ch != '\v' && ch != '#') {
        if (n + 1 < cap) tok[n++] = (char)ch;
        ch = fgetc(fp);
    }
    tok[n] = 0;
    if (ch == '#') {  // comment glued to a token: drop it up to the end of the line
        while (ch != '\n' && ch != EOF) ch = fgetc(fp);
    }
    // the whitespace byte that ended the token is consumed (for the last token it is THE separator)
    return n > 0;
}
}  // namespace

int readPPMImage(const char* path, size_t* width, size_t* height, rgb_pixel_t** img) {
    FILE* fp = fopen(path, "rb");
    if (!fp) {
        std::cout << "Error opening the file" << std::endl;
        return -1;
    }
    char tok[32];
    if (!ppm_token(fp, tok, sizeof tok)) {
        std::cout << "Error reading the file" << std::endl;
        fclose(fp);
        return -1;
    }
    if (strcmp(tok, "P6") != 0) {
        std::cout << "Invalid file format" << std::endl;
        fclose(fp);
        return -1;
    }
    long vals[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        char* end = nullptr;
        if (!ppm_token(fp, tok, sizeof tok) || (vals[i] = strtol(tok, &end, 10), *end != 0) || vals[i] <= 0) {
            std::cout << "Invalid file format" << std::endl;
            fclose(fp);
            return -1;
        }
    }
    if (vals[2] != 255) {
        std::cout << "Invalid maximum value" << std::endl;
        fclose(fp);
        return -1;
    }
    const long w = vals[0], h = vals[1];
    if (w > 65535 || h > 65535) {
        std::cout << "Invalid file format" << std::endl;
        fclose(fp);
        return -1;
    }
    size_t n = (size_t)w * (size_t)h;
    rgb_pixel_t* p = (rgb_pixel_t*)malloc(n * sizeof(rgb_pixel_t));
    if (!p) {
        std::cout << "Error allocating memory" << std::endl;
        fclose(fp);
        return -1;
    }
    if (fread(p, sizeof(rgb_pixel_t), n, fp) != n) {
        std::cout << "Error reading the file" << std::endl;
        free(p);
        fclose(fp);
        return -1;
    }
    fclose(fp);
    *width = (size_t)w;
    *height = (size_t)h;
    *img = p;
    return 0;
}

int writePPMImage(const char* path, size_t width, size_t height, rgb_pixel_t* img) {
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        std::cout << "Error opening the file" << std::endl;
        return -1;
    }
    fprintf(fp, "P6\n%zu %zu\n255\n", width, height);
    fwrite(img, sizeof(rgb_pixel_t), width * height, fp);
    fclose(fp);
    return 0;
}

void getNearest8x8ImageSize(size_t width, size_t height, size_t* newWidth, size_t* newHeight) {
    uint32_t w8, h8;
    mi355_jpeg_padded_size((uint32_t)width, (uint32_t)height, &w8, &h8);
    *newWidth = w8;
    *newHeight = h8;
}

mi355_jpeg_ctx* mi355_context() { return ensure_ctx() ? nullptr : g_ctx; }

// ---- encode path -------------------------------------------------------------------------
int mi355_select(int device, int quality) {
    if (g_ctx) {
        mi355_jpeg_destroy(g_ctx);
        g_ctx = nullptr;
    }
    g_device = device;
    g_quality = quality;
    return ensure_ctx();
}

void mi355_set_mode(unsigned mode_flags) { g_mode = mode_flags & (MI355_F_STANDARD | MI355_F_420 | MI355_F_RESTART); }

int JpegEncoderDevice(ppm_t img, GPUTelemetry* tel, std::string* scanData, bool cds) {
    if (ensure_ctx()) return 1;
    std::vector<uint8_t> scan(mi355_jpeg_scan_bound((uint32_t)img.width, (uint32_t)img.height));
    uint64_t nbits = 0;
    mi355_jpeg_set_profiling(g_ctx, 1);
    auto t0 = std::chrono::steady_clock::now();
    int rc = mi355_jpeg_encode_scan(g_ctx, (const uint8_t*)img.data, (uint32_t)img.width, (uint32_t)img.height, 1,
                                    path_flags(cds), scan.data(), scan.size(), &nbits);
    auto t1 = std::chrono::steady_clock::now();
    if (rc) return fail(rc);
    if (tel) {
        mi355_jpeg_timings t;
        mi355_jpeg_last_timings(g_ctx, &t);
        tel->blockEncodeTime = t.transform_ms * 1e3;
        tel->fixupTime = t.size_ms * 1e3;
        tel->scanTime = t.scan_ms * 1e3;
        tel->emitTime = t.emit_ms * 1e3;
        tel->totalTime = t.total_ms * 1e3;
        tel->wallTime = std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    mi355_jpeg_set_profiling(g_ctx, 0);
    if (scanData) {
        scanData->assign((size_t)nbits, '0');
        for (uint64_t i = 0; i < nbits; ++i)
            if (scan[(size_t)(i >> 3)] & (0x80u >> (i & 7))) (*scanData)[(size_t)i] = '1';
    }
    return 0;
}

int transformToZigZag(ppm_t img, int zigzag_arr[][64], bool cds) {
    if (ensure_ctx()) return 1;
    size_t w8, h8;
    getNearest8x8ImageSize(img.width, img.height, &w8, &h8);
    size_t rows = w8 * h8 / 64 * 3;
    std::vector<int16_t> tmp(rows * 64);
    int rc = mi355_jpeg_probe_coefficients(g_ctx, (const uint8_t*)img.data, (uint32_t)img.width,
                                           (uint32_t)img.height, cds ? MI355_F_CDS : 0u, tmp.data());
    if (rc) return fail(rc);
    for (size_t i = 0; i < rows; ++i)
        for (int k = 0; k < 64; ++k) zigzag_arr[i][k] = tmp[i * 64 + (size_t)k];
    return 0;
}

std::string HuffmanEncoder(int zigzag_arr[][64], int numRowsPerChannel) {
    std::string out;
    if (ensure_ctx() || numRowsPerChannel <= 0) return out;
    size_t rows = (size_t)numRowsPerChannel * 3;
    std::vector<int16_t> tmp(rows * 64);
    for (size_t i = 0; i < rows; ++i)
        for (int k = 0; k < 64; ++k) tmp[i * 64 + (size_t)k] = (int16_t)zigzag_arr[i][k];
    std::vector<uint8_t> scan((rows * 1727 + 7) / 8 + 8);
    uint64_t nbits = 0;
    int rc = mi355_jpeg_entropy_only(g_ctx, tmp.data(), (uint32_t)numRowsPerChannel, scan.data(), scan.size(), &nbits);
    if (rc) {
        fail(rc);
        return out;
    }
    out.assign((size_t)nbits, '0');
    for (uint64_t i = 0; i < nbits; ++i)
        if (scan[(size_t)(i >> 3)] & (0x80u >> (i & 7))) out[(size_t)i] = '1';
    return out;
}

int writeJpegFile(const char* path, ppm_t img, bool cds) {
    if (ensure_ctx()) return 1;
    size_t cap = 2 * mi355_jpeg_scan_bound((uint32_t)img.width, (uint32_t)img.height) + 4096, len = 0;
    std::vector<uint8_t> buf(cap);
    int rc = mi355_jpeg_encode_jfif(g_ctx, (const uint8_t*)img.data, (uint32_t)img.width, (uint32_t)img.height,
                                    path_flags(cds), buf.data(), cap, &len);
    if (rc) return fail(rc);
    FILE* fp = fopen(path, "wb");
    if (!fp) {
        std::cout << "Error opening the file" << std::endl;
        return 1;
    }
    fwrite(buf.data(), 1, len, fp);
    fclose(fp);
    return 0;
}
