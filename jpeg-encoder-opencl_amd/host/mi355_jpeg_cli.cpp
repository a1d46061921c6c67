// mi355-jpeg: PPM in, JFIF out, through the MI355X encode path.
//
//   mi355-jpeg in.ppm out.jpg [-q N] [--mode strict|standard] [--subsample ref420|none|420|444]
//              [--no-cds] [--device K] [--repeat R] [--bits out.bits]
//
// --mode strict (default): the reference's arithmetic; --subsample ref420 (default) = its performCDS
// (2x2 chroma means written back at full resolution), none (= --no-cds) skips it.
// --mode standard: a decodable baseline JPEG (not a behaviour of the reference); --subsample 420
// (default there) = real 16x16 MCUs, 444 = one block per component.
//
// With no arguments, run from a build/ directory like the reference
// (README.md:43, src/OpenCLProject_JpegEncoder.cpp:320): read ../data/fruit.ppm and
// report the stage times; the reference writes no output file, this tool writes
// ../data/fruit.jpg.  Strict mode reproduces the reference's arithmetic, so the file's
// pixels are not a meaningful picture; its scan bits are the parity artefact.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "mi355_utils.hpp"

static void usage() {
    std::cout << "usage: mi355-jpeg in.ppm out.jpg [-q 1..100] [--mode strict|standard] [--subsample ref420|none|420|444]\n"
                 "                  [--no-cds] [--device K] [--repeat R] [--bits file]\n"
                 "       (no arguments: ../data/fruit.ppm -> ../data/fruit.jpg, like the reference's fixed paths)\n";
}

int main(int argc, char** argv) {
    std::string in = "../data/fruit.ppm", out = "../data/fruit.jpg", bits_path;
    int quality = 50, device = 0, repeat = 1, pos = 0;
    bool cds = true;
    std::string mode = "strict", subsample;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else if (a == "-q" && i + 1 < argc) {
            quality = atoi(argv[++i]);
        } else if (a == "--no-cds") {
            cds = false;
        } else if (a == "--mode" && i + 1 < argc) {
            mode = argv[++i];
        } else if (a == "--subsample" && i + 1 < argc) {
            subsample = argv[++i];
        } else if (a == "--device" && i + 1 < argc) {
            device = atoi(argv[++i]);
        } else if (a == "--repeat" && i + 1 < argc) {
            repeat = atoi(argv[++i]);
        } else if (a == "--bits" && i + 1 < argc) {
            bits_path = argv[++i];
        } else if (a[0] != '-' && pos == 0) {
            in = a;
            ++pos;
        } else if (a[0] != '-' && pos == 1) {
            out = a;
            ++pos;
        } else {
            usage();
            return 2;
        }
    }
    if (quality < 1 || quality > 100 || repeat < 1) {
        usage();
        return 2;
    }
    unsigned mode_flags = 0;
    if (mode == "strict") {
        if (subsample == "none") cds = false;
        else if (!subsample.empty() && subsample != "ref420") {
            usage();
            return 2;
        }
    } else if (mode == "standard") {
        if (subsample.empty() || subsample == "420") mode_flags = MI355_F_STANDARD | MI355_F_420;
        else if (subsample == "444") mode_flags = MI355_F_STANDARD;
        else {
            usage();
            return 2;
        }
    } else {
        usage();
        return 2;
    }
    ppm_t img;
    if (readPPMImage(in.c_str(), &img.width, &img.height, &img.data) == -1) return 1;
    std::cout << "Image " << in << ": " << img.width << " x " << img.height << std::endl;
    if (mi355_select(device, quality)) return 1;
    mi355_set_mode(mode_flags);

    std::cout << "\n### MI355X Implementation ###" << std::endl;
    GPUTelemetry tel;
    std::string scan;
    for (int r = 0; r < repeat; ++r)
        if (JpegEncoderDevice(img, &tel, r == repeat - 1 ? &scan : NULL, cds)) return 1;
    std::cout << "Block encode (CSC..RLE/Huffman strings) Time GPU: " << tel.blockEncodeTime << " us\n"
              << "DC heads Time GPU: " << tel.fixupTime << " us\n"
              << "Prefix scan Time GPU: " << tel.scanTime << " us\n"
              << "Bit string emit Time GPU: " << tel.emitTime << " us\n"
              << "Total Time GPU: " << tel.totalTime << " us (wall incl. transfers " << tel.wallTime << " us)\n"
              << "Scan bits: " << scan.size() << std::endl;
    if (!bits_path.empty()) {
        FILE* fp = fopen(bits_path.c_str(), "wb");
        if (fp) {
            fwrite(scan.data(), 1, scan.size(), fp);
            fclose(fp);
        }
    }
    if (writeJpegFile(out.c_str(), img, cds)) return 1;
    std::cout << "Wrote " << out << std::endl;
    free(img.data);
    return 0;
}
