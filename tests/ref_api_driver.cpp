// TEST INFRASTRUCTURE.  A driver written against the reference's stage-function interface, stage by stage in
// the order of JpegEncoderHost (src/OpenCLProject_JpegEncoder.cpp:59-225), that dumps every intermediate.
//
// Built by oracle/Makefile into oracle/_ref/ref_api_driver.  Where /root/reference exists it is compiled
// against the REFERENCE'S OWN header, included in place (-DMI355_USE_REFERENCE_HEADER -I/root/reference/src),
// and linked against libmi355host.a: the reference's declarations (utils.hpp:77-137) resolve, name for name and
// signature for signature, to this repo's GPU implementations.  Elsewhere it is compiled against this
// repo's host/mi355_utils.hpp (same declarations).
//
//   ref_api_driver in.ppm out_prefix     -> out_prefix.{csc,cds,pad,dct,quant,zigzag,rle,scan,tel}
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifdef MI355_USE_REFERENCE_HEADER
#include <CL/cl_platform.h>  // cl_uint, which the reference header uses without including it
#include "utils.hpp"         // /root/reference/src/utils.hpp, in place
#else
#include "mi355_utils.hpp"
#endif

// the two entry points of the drop-in that are not in the reference's utils.hpp: its driver function lives in its
// main file (OpenCLProject_JpegEncoder.cpp:28)
int JpegEncoderHost(ppm_t imgCPU, CPUTelemetry* cpu_telemetry);
const std::string& mi355_last_scan();

static void dump(const std::string& path, const void* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) {
        printf("cannot write %s\n", path.c_str());
        exit(1);
    }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string pre = argv[2];
    ppm_t img;
    if (readPPMImage(argv[1], &img.width, &img.height, &img.data) == -1) return 1;
    const size_t W = img.width, H = img.height;
    // a second copy for the driver function at the end (the stages work in place)
    ppm_t img2 = img;
    img2.data = (rgb_pixel_t*)malloc(W * H * sizeof(rgb_pixel_t));
    memcpy(img2.data, img.data, W * H * sizeof(rgb_pixel_t));

    performCSC(&img);
    dump(pre + ".csc", img.data, W * H * 3);
    performCDS(&img);
    dump(pre + ".cds", img.data, W * H * 3);
    size_t W8, H8;
    if (W % 8 == 0 && H % 8 == 0) W8 = W, H8 = H;
    else getNearest8x8ImageSize(W, H, &W8, &H8);
    ppm_t big = {W8, H8, (rgb_pixel_t*)calloc(W8 * H8, sizeof(rgb_pixel_t))};
    copyToLargerImage(&img, &big);
    addReversedPadding(&big, W, H);
    dump(pre + ".pad", big.data, W8 * H8 * 3);
    ppm_d_t d = {W8, H8, (rgb_pixel_d_t*)malloc(W8 * H8 * sizeof(rgb_pixel_d_t))};
    copyUIntToDoubleImage(&big, &d);
    substractfromAll(&d, 128.0);
    performDCT(&d);
    dump(pre + ".dct", d.data, W8 * H8 * 3 * sizeof(double));
    performQuantization(&d, quant_mat_lum, quant_mat_chrom);
    dump(pre + ".quant", d.data, W8 * H8 * 3 * sizeof(double));
    const size_t rows = W8 * H8 / 64 * 3, per = W8 * H8 / 64;
    int(*lin)[64] = (int(*)[64])malloc(rows * 64 * sizeof(int));
    int(*zz)[64] = (int(*)[64])malloc(rows * 64 * sizeof(int));
    everyMCUisnow2DArray(&d, lin);
    performZigZag(lin, zz, (int)rows);
    dump(pre + ".zigzag", zz, rows * 64 * sizeof(int));
    std::vector<std::vector<int>> rle;
    performRLE(zz, rle, (int)rows);
    {
        std::vector<int> flat;  // per row: count, then the pairs
        for (const auto& v : rle) {
            flat.push_back((int)v.size());
            flat.insert(flat.end(), v.begin(), v.end());
        }
        dump(pre + ".rle", flat.data(), flat.size() * sizeof(int));
    }
    const std::string scan = HuffmanEncoder(zz, rle, (int)per);
    dump(pre + ".scan", scan.data(), scan.size());

    // the driver function, on the untouched copy: same scan, nine telemetry fields filled
    CPUTelemetry t;
    memset(&t, 0, sizeof t);
    if (JpegEncoderHost(img2, &t)) return 1;
    if (mi355_last_scan() != scan) {
        printf("JpegEncoderHost's scan differs from the stage-by-stage one\n");
        return 1;
    }
    const double tel[9] = {t.CSCTime, t.CDSTime, t.levelShiftTime, t.DCTTime, t.QuantTime, t.TotalCopyTime, t.zigZagTime, t.RLETime, t.HuffmanTime};
    dump(pre + ".tel", tel, sizeof tel);
    printf("ok %zu x %zu -> %zu x %zu, %zu scan bits\n", W, H, W8, H8, scan.size());
    return 0;
}
