// Test driver written the way the reference's JpegEncoderHost drives its stage library
// (src/OpenCLProject_JpegEncoder.cpp:28-250), against the utils.hpp-shaped host API:
//   host_driver in.ppm out_prefix [quality] [cds 0/1]
// writes <prefix>.bits ('0'/'1' string of the whole path), <prefix>.bits2 (the same string
// produced stage-wise: transformToZigZag + HuffmanEncoder), <prefix>.tables (the code
// tables as strings) and <prefix>.jpg.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../jpeg-encoder-opencl_amd/host/mi355_utils.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string prefix = argv[2];
    const int quality = argc > 3 ? atoi(argv[3]) : 50;
    const bool cds = argc > 4 ? atoi(argv[4]) != 0 : true;

    {   // table strings need no GPU
        std::ofstream t(prefix + ".tables");
        for (auto& s : DC_LUMA_HUFF_CODES) t << s << "\n";
        for (auto& s : DC_CHROMA_HUFF_CODES) t << s << "\n";
        for (auto& row : AC_LUMA_HUFF_CODES)
            for (auto& s : row) t << s << "\n";
        for (auto& row : AC_CHROMA_HUFF_CODES)
            for (auto& s : row) t << s << "\n";
        t << quant_mat_lum[7][7] << " " << quant_mat_chrom[0][1] << "\n";
    }

    ppm_t imgCPU;
    if (readPPMImage(argv[1], &imgCPU.width, &imgCPU.height, &imgCPU.data) == -1) return 1;
    if (mi355_select(0, quality)) return 1;

    GPUTelemetry tel;
    std::string scanData;
    if (JpegEncoderDevice(imgCPU, &tel, &scanData, cds)) return 1;
    std::ofstream(prefix + ".bits") << scanData;
    std::cout << "Total Time GPU: " << tel.totalTime << " us, scan bits " << scanData.size() << std::endl;

    size_t newWidth, newHeight;
    getNearest8x8ImageSize(imgCPU.width, imgCPU.height, &newWidth, &newHeight);
    unsigned int rowsperchannel = (unsigned int)(newWidth * newHeight / 64);
    unsigned int rows = rowsperchannel * 3;
    int(*zigzag_arr)[64] = new int[rows][64];
    if (transformToZigZag(imgCPU, zigzag_arr, cds)) return 1;
    std::string scanData2 = HuffmanEncoder(zigzag_arr, (int)rowsperchannel);
    std::ofstream(prefix + ".bits2") << scanData2;
    delete[] zigzag_arr;

    if (writeJpegFile((prefix + ".jpg").c_str(), imgCPU, cds)) return 1;
    free(imgCPU.data);
    return scanData == scanData2 ? 0 : 3;
}
