// The reference's stage functions with the reference's own signatures (src/utils.hpp:77-137), each running
// its stage on the GPU through the C ABI (mi355_jpeg_stage_*).  No stage arithmetic happens on the host: this
// file only marshals the reference's data structures (ppm_t, ppm_d_t, int[][64], vector<vector<int>>) into
// flat buffers and back.
#include "mi355_utils.hpp"

#include <cstring>
#include <iostream>
#include <stdexcept>

namespace {
mi355_jpeg_ctx* ctx() {
    mi355_jpeg_ctx* c = mi355_context();
    if (!c) throw std::runtime_error("mi355-jpeg: no usable device");
    return c;
}
void check(int rc) {
    if (rc == MI355_OK) return;
    std::cout << "mi355-jpeg: " << mi355_jpeg_strerror(rc) << std::endl;  // stdout, like the reference's messages
    throw std::runtime_error(mi355_jpeg_strerror(rc));
}
}  // namespace

// indexing sugar of the reference (utils.cpp:144-181); no arithmetic
rgb_pixel_t* getPixelPtr(ppm_t* img, size_t x, size_t y) { return &img->data[y * img->width + x]; }
rgb_pixel_t getPixel(ppm_t* img, size_t x, size_t y) { return img->data[y * img->width + x]; }
uint8_t getPixelR(ppm_t* img, size_t x, size_t y) { return img->data[y * img->width + x].r; }
uint8_t getPixelG(ppm_t* img, size_t x, size_t y) { return img->data[y * img->width + x].g; }
uint8_t getPixelB(ppm_t* img, size_t x, size_t y) { return img->data[y * img->width + x].b; }
void setPixelR(ppm_t* img, size_t x, size_t y, uint8_t v) { img->data[y * img->width + x].r = v; }
void setPixelG(ppm_t* img, size_t x, size_t y, uint8_t v) { img->data[y * img->width + x].g = v; }
void setPixelB(ppm_t* img, size_t x, size_t y, uint8_t v) { img->data[y * img->width + x].b = v; }

// the reference's debug helper (utils.cpp:84-89), not on the path
void removeRedChannel(ppm_t* img) {
    for (size_t i = 0; i < img->width * img->height; ++i) img->data[i].r = 0;
}

void performCSC(ppm_t* img) { check(mi355_jpeg_stage_csc(ctx(), (uint8_t*)img->data, (uint32_t)img->width, (uint32_t)img->height)); }
void performCDS(ppm_t* img) { check(mi355_jpeg_stage_cds(ctx(), (uint8_t*)img->data, (uint32_t)img->width, (uint32_t)img->height)); }

void copyToLargerImage(ppm_t* img, ppm_t* newImg) {
    check(mi355_jpeg_stage_copy_larger(ctx(), (const uint8_t*)img->data, (uint32_t)img->width, (uint32_t)img->height,
                                       (uint8_t*)newImg->data, (uint32_t)newImg->width, (uint32_t)newImg->height));
}
void addReversedPadding(ppm_t* img, size_t oldWidth, size_t oldHeight) {
    check(mi355_jpeg_stage_mirror_pad(ctx(), (uint8_t*)img->data, (uint32_t)img->width, (uint32_t)img->height, (uint32_t)oldWidth,
                                      (uint32_t)oldHeight));
}
void copyUIntToDoubleImage(ppm_t* img, ppm_d_t* newImg) {
    check(mi355_jpeg_stage_to_double(ctx(), (const uint8_t*)img->data, (double*)newImg->data, (uint32_t)img->width,
                                     (uint32_t)img->height));
}
void substractfromAll(ppm_d_t* img, double val) {
    check(mi355_jpeg_stage_subtract(ctx(), (double*)img->data, (uint32_t)img->width, (uint32_t)img->height, val));
}
void performDCT(ppm_d_t* img) { check(mi355_jpeg_stage_dct(ctx(), (double*)img->data, (uint32_t)img->width, (uint32_t)img->height)); }

void performQuantization(ppm_d_t* img, const unsigned int qlum[][8], const unsigned int qchrom[][8]) {
    uint32_t ql[64], qc[64];
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) ql[v * 8 + u] = qlum[v][u], qc[v * 8 + u] = qchrom[v][u];
    check(mi355_jpeg_stage_quantize(ctx(), (double*)img->data, (uint32_t)img->width, (uint32_t)img->height, ql, qc));
}
void everyMCUisnow2DArray(ppm_d_t* img, int linear_arr[][64]) {
    check(mi355_jpeg_stage_blocks(ctx(), (const double*)img->data, (uint32_t)img->width, (uint32_t)img->height,
                                  (int32_t*)&linear_arr[0][0]));
}
void performZigZag(int linear_arr[][64], int zigzag_arr[][64], int rows) {
    if (rows <= 0) return;
    check(mi355_jpeg_stage_zigzag(ctx(), (const int32_t*)&linear_arr[0][0], (int32_t*)&zigzag_arr[0][0], (uint32_t)rows));
}

void performRLE(int zigzag_array[][64], std::vector<std::vector<int>>& rle_vector, int rows) {
    if (rows <= 0) return;
    std::vector<int32_t> pairs((size_t)rows * 128);
    std::vector<uint32_t> counts((size_t)rows);
    check(mi355_jpeg_stage_rle(ctx(), (const int32_t*)&zigzag_array[0][0], (uint32_t)rows, pairs.data(), counts.data()));
    for (int r = 0; r < rows; ++r)  // the reference appends one vector per row (utils.cpp:616-618)
        rle_vector.emplace_back(pairs.begin() + (size_t)r * 128, pairs.begin() + (size_t)r * 128 + counts[(size_t)r]);
}

std::string HuffmanEncoder(int zigzag_array[][64], std::vector<std::vector<int>>& rle_vector, int numRowsPerChannel) {
    std::string out;
    if (numRowsPerChannel <= 0) return out;
    const size_t rows = (size_t)numRowsPerChannel * 3;
    if (rle_vector.size() < rows) check(MI355_E_ARG);
    std::vector<int32_t> pairs(rows * 128, 0);
    std::vector<uint32_t> counts(rows);
    size_t sym = 0;
    for (size_t r = 0; r < rows; ++r) {
        const std::vector<int>& v = rle_vector[r];
        if (v.size() > 128) check(MI355_E_ARG);  // more pairs than a block has coefficients
        counts[r] = (uint32_t)(v.size() & ~(size_t)1);
        memcpy(&pairs[r * 128], v.data(), counts[r] * sizeof(int));
        sym += counts[r] / 2 + 1;
    }
    const size_t cap = (sym * 27 + 7) / 8 + 16;  // a symbol is at most 17 + 10 bits
    std::vector<uint8_t> scan(cap);
    uint64_t nbits = 0;
    check(mi355_jpeg_stage_huffman(ctx(), (const int32_t*)&zigzag_array[0][0], pairs.data(), counts.data(), (uint32_t)numRowsPerChannel,
                                   scan.data(), cap, &nbits));
    out.assign((size_t)nbits, '0');
    for (uint64_t i = 0; i < nbits; ++i)
        if (scan[(size_t)(i >> 3)] & (0x80u >> (i & 7))) out[(size_t)i] = '1';
    return out;
}
