// JpegEncoderHost with the reference's signature (src/OpenCLProject_JpegEncoder.cpp:28): the reference's stage
// sequence (:59-225) through the stage functions of mi355_stage_api.cpp, each timed into the reference's
// CPUTelemetry.  Own object file: a program that brings its own JpegEncoderHost (the reference's main file does)
// never pulls this one out of libmi355host.a.
#include "mi355_utils.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

namespace {
std::string g_last_scan;
struct Clock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {  // microseconds since the previous lap
        const auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
        return us;
    }
};
}  // namespace

const std::string& mi355_last_scan() { return g_last_scan; }

int JpegEncoderHost(ppm_t imgCPU, CPUTelemetry* cpu_telemetry) {
    std::cout << "\n### MI355X implementation, stage by stage ###" << std::endl;
    CPUTelemetry t;
    memset(&t, 0, sizeof t);
    ppm_t img3 = {0, 0, nullptr};
    ppm_d_t imgd = {0, 0, nullptr};
    int (*linear)[64] = nullptr, (*zigzag)[64] = nullptr;
    int rc = 0;
    try {
        Clock clk;
        performCSC(&imgCPU);
        t.CSCTime = clk.lap();
        std::cout << "CSC Time MI355X: " << t.CSCTime << " us" << std::endl;
        performCDS(&imgCPU);
        t.CDSTime = clk.lap();
        std::cout << "CDS Time MI355X: " << t.CDSTime << " us" << std::endl;

        size_t newWidth, newHeight;
        if (imgCPU.width % 8 == 0 && imgCPU.height % 8 == 0) newWidth = imgCPU.width, newHeight = imgCPU.height;
        else getNearest8x8ImageSize(imgCPU.width, imgCPU.height, &newWidth, &newHeight);
        img3.width = newWidth, img3.height = newHeight;
        img3.data = (rgb_pixel_t*)malloc(newWidth * newHeight * sizeof(rgb_pixel_t));
        imgd.width = newWidth, imgd.height = newHeight;
        imgd.data = (rgb_pixel_d_t*)malloc(newWidth * newHeight * sizeof(rgb_pixel_d_t));
        const size_t rows = newWidth * newHeight / 64 * 3, rowsperchannel = newWidth * newHeight / 64;
        linear = (int(*)[64])malloc(rows * 64 * sizeof(int));
        zigzag = (int(*)[64])malloc(rows * 64 * sizeof(int));
        if (!img3.data || !imgd.data || !linear || !zigzag) throw std::runtime_error("out of memory");
        memset(img3.data, 0, newWidth * newHeight * sizeof(rgb_pixel_t));

        clk.lap();
        copyToLargerImage(&imgCPU, &img3);
        const double copy1 = clk.lap();
        addReversedPadding(&img3, imgCPU.width, imgCPU.height);  // untimed in the reference as well (:120)
        clk.lap();
        copyUIntToDoubleImage(&img3, &imgd);
        t.TotalCopyTime = copy1 + clk.lap();
        std::cout << "Total Copy Time MI355X: " << t.TotalCopyTime << " us" << std::endl;
        substractfromAll(&imgd, 128.0);
        t.levelShiftTime = clk.lap();
        std::cout << "Level Shifting Time MI355X: " << t.levelShiftTime << " us" << std::endl;
        performDCT(&imgd);
        t.DCTTime = clk.lap();
        std::cout << "DCT Time MI355X: " << t.DCTTime << " us" << std::endl;
        performQuantization(&imgd, quant_mat_lum, quant_mat_chrom);
        t.QuantTime = clk.lap();
        std::cout << "Quantization Time MI355X: " << t.QuantTime << " us" << std::endl;
        everyMCUisnow2DArray(&imgd, linear);
        performZigZag(linear, zigzag, (int)rows);
        t.zigZagTime = clk.lap();
        std::cout << "ZigZag Time MI355X: " << t.zigZagTime << " us" << std::endl;
        std::vector<std::vector<int>> rle;
        performRLE(zigzag, rle, (int)rows);
        t.RLETime = clk.lap();
        std::cout << "RLE Time MI355X: " << t.RLETime << " us" << std::endl;
        g_last_scan = HuffmanEncoder(zigzag, rle, (int)rowsperchannel);
        t.HuffmanTime = clk.lap();
        std::cout << "Huffman Time MI355X: " << t.HuffmanTime << " us" << std::endl;
        std::cout << "Total Time MI355X: "
                  << (t.CSCTime + t.CDSTime + t.TotalCopyTime + t.levelShiftTime + t.DCTTime + t.QuantTime + t.zigZagTime +
                      t.RLETime + t.HuffmanTime)
                  << " us" << std::endl;
    } catch (const std::exception& ex) {
        std::cout << "Error in the encode path: " << ex.what() << std::endl;
        rc = 1;
    }
    free(img3.data);
    free(imgd.data);
    free(linear);
    free(zigzag);
    if (cpu_telemetry && !rc) *cpu_telemetry = t;
    return rc;
}
