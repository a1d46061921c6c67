// Host-side C++ face of the MI355X encode path, shaped like the reference's interface
// (src/utils.hpp + src/huffman.hpp) so that a JpegEncoderHost-style driver
// (src/OpenCLProject_JpegEncoder.cpp:28-250) can be written against it.
//
// What is mirrored, and why only this much: the reference's interface is a bag of
// in-place stage functions over host images (performCSC(ppm_t*), performCDS(ppm_t*), ...
// utils.hpp:77-137).  The GPU path is fused -- those intermediates never exist in memory
// -- so the shim exposes the reference's data types, its constant tables, its PPM I/O
// helpers, and the path at the three granularities the library has:
//     JpegEncoderDevice()        = the whole of JpegEncoderHost's stage sequence (:59-225)
//     transformToZigZag()        = performCSC ... performZigZag            (:59-197)
//     HuffmanEncoder(zigzag, n)  = performRLE + HuffmanEncoder             (:213-225)
// All compute happens in libmi355jpeg.so (include/mi355_jpeg.h); nothing here falls back
// to the CPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi355_jpeg.h"

// ---- data types (utils.hpp:7-39) -------------------------------------------------
struct rgb_pixel {
    uint8_t r, g, b;
};
typedef struct rgb_pixel rgb_pixel_t;
struct PPMimage {
    size_t width, height;
    rgb_pixel_t* data;
};
typedef struct PPMimage ppm_t;

// data structures for the intermediate steps (utils.hpp:25-39): the same image with double fields
struct rgb_pixel_d {
    double r, g, b;
};
typedef struct rgb_pixel_d rgb_pixel_d_t;
struct PPMimage_d {
    size_t width, height;
    rgb_pixel_d_t* data;
};
typedef struct PPMimage_d ppm_d_t;

// ---- constant tables (utils.hpp:41-62, huffman.hpp) -------------------------------
// Same names, types and indexing as the reference: [v][u] for the quantisation tables,
// [size] / [run][size] '0'/'1' strings for the code tables ("NULL" where the reference
// has no code; AC_LUMA_HUFF_CODES[3][4..10] are the 17-character entries).
extern const unsigned int quant_mat_lum[8][8];
extern const unsigned int quant_mat_chrom[8][8];
extern const std::vector<std::string> DC_LUMA_HUFF_CODES;
extern const std::vector<std::string> DC_CHROMA_HUFF_CODES;
extern const std::vector<std::vector<std::string>> AC_LUMA_HUFF_CODES;
extern const std::vector<std::vector<std::string>> AC_CHROMA_HUFF_CODES;

// ---- telemetry (utils.hpp:65-75): the reference's struct, field for field, microseconds --------
struct CPUTelemetry {
    double CSCTime;
    double CDSTime;
    double levelShiftTime;
    double DCTTime;
    double QuantTime;
    double TotalCopyTime;
    double zigZagTime;
    double RLETime;
    double HuffmanTime;
};

// ---- telemetry of the fused path, microseconds ---------------
struct GPUTelemetry {
    double blockEncodeTime;  // fused CSC .. per-unit RLE/Huffman strings (k_screen_encode)
    double fixupTime;        // DC symbols at tile heads (k_dc_heads; the name dates from a separate fix-up kernel)
    double scanTime;         // prefix sum of tile bit counts
    double emitTime;         // final bit string
    double totalTime;        // device time of the whole path
    double wallTime;         // host wall clock incl. PCIe transfers
};

// ---- PPM I/O (utils.cpp:11-82): same contract: 0 ok, -1 + message on stdout ------------
int readPPMImage(const char* path, size_t* width, size_t* height, rgb_pixel_t** img);
int writePPMImage(const char* path, size_t width, size_t height, rgb_pixel_t* img);
void getNearest8x8ImageSize(size_t width, size_t height, size_t* newWidth, size_t* newHeight);

// ---- the reference's stage functions, with the reference's signatures (utils.hpp:77-137) --------------
// Each one runs its stage as a HIP kernel on the host image it is given (upload, kernel, download:
// include/mi355_jpeg.h, mi355_jpeg_stage_*), in place like the reference, with bit-identical results.  A driver
// written stage by stage against the reference's header -- JpegEncoderHost, OpenCLProject_JpegEncoder.cpp:28-250
// -- links against libmi355host.a unmodified.  They are the compatibility path: JpegEncoderDevice below runs
// the same stage sequence fused, ~10^4 times faster.  Errors (no device, ...) cannot be returned through
// void: they are printed to stdout like the reference's messages and thrown as std::runtime_error.
void removeRedChannel(ppm_t*);  // the reference's "TEST FUNCTION" (utils.cpp:84-89), host side
void performCSC(ppm_t*);
void performCDS(ppm_t*);
rgb_pixel_t* getPixelPtr(ppm_t*, size_t, size_t);
rgb_pixel_t getPixel(ppm_t*, size_t, size_t);
uint8_t getPixelR(ppm_t*, size_t, size_t);
uint8_t getPixelG(ppm_t*, size_t, size_t);
uint8_t getPixelB(ppm_t*, size_t, size_t);
void setPixelR(ppm_t*, size_t, size_t, uint8_t);
void setPixelG(ppm_t*, size_t, size_t, uint8_t);
void setPixelB(ppm_t*, size_t, size_t, uint8_t);
void copyUIntToDoubleImage(ppm_t*, ppm_d_t*);
void copyToLargerImage(ppm_t*, ppm_t*);
void addReversedPadding(ppm_t*, size_t, size_t);
void substractfromAll(ppm_d_t*, double);
void performDCT(ppm_d_t*);
void performQuantization(ppm_d_t*, const unsigned int[][8], const unsigned int[][8]);
void everyMCUisnow2DArray(ppm_d_t*, int[][64]);
void performZigZag(int[][64], int[][64], int);
void performRLE(int[][64], std::vector<std::vector<int>>&, int);
std::string HuffmanEncoder(int[][64], std::vector<std::vector<int>>&, int);
// The reference's driver (OpenCLProject_JpegEncoder.cpp:28-250): the stage functions above in the reference's
// order, each timed with the wall clock into the reference's nine CPUTelemetry fields (upload and download
// included: that is what a stage-by-stage drop-in costs).  Heap arrays instead of the reference's stack VLAs
// (:190-191), no debug PPM dumps.  Returns 0, or 1 after an error message.  The scan string it ends with
// (which the reference discards, :225) stays available through mi355_last_scan().  Lives in its own object
// file of libmi355host.a, so a program that defines JpegEncoderHost itself (the reference's main file) links too.
int JpegEncoderHost(ppm_t imgCPU, CPUTelemetry* cpu_telemetry);
const std::string& mi355_last_scan();
// the process-wide context the functions of this header share (created on first use; NULL after a failure)
mi355_jpeg_ctx* mi355_context();

// ---- the encode path -----------------------------------------------------------------
// One process-wide context on `device` (created on first use); quality 50 = the
// reference's tables.  All return 0 on success, 1 on error (message on stdout), like
// JpegEncoderHost.
int mi355_select(int device, int quality);
// Encode mode of the calls below.  0 (default) = strict: the reference's arithmetic, scan bits identical
// to its CPU path.  MI355_F_STANDARD [| MI355_F_420] [| MI355_F_RESTART] = decodable baseline JPEG, 4:4:4 or 4:2:0 (not a
// behaviour of the reference; chroma_downsample is ignored there).  See include/mi355_jpeg.h.
void mi355_set_mode(unsigned mode_flags);

// The whole stage sequence of JpegEncoderHost; *scanData receives the '0'/'1' string
// that the reference's HuffmanEncoder returns (utils.cpp:697).
int JpegEncoderDevice(ppm_t img, GPUTelemetry* telemetry = NULL, std::string* scanData = NULL,
                      bool chroma_downsample = true);

// performCSC .. performZigZag: zigzag_arr must hold 3*N rows (N = padded blocks), row
// order chan*N + block as everyMCUisnow2DArray lays it out (utils.cpp:482-498).
int transformToZigZag(ppm_t img, int zigzag_arr[][64], bool chroma_downsample = true);

// performRLE + HuffmanEncoder on a coefficient array in that row order.
std::string HuffmanEncoder(int zigzag_arr[][64], int numRowsPerChannel);

// Build-defined JFIF file around the scan (the reference writes none).
int writeJpegFile(const char* path, ppm_t img, bool chroma_downsample = true);
