// Internal interface between the C ABI (mi355_jpeg.cpp) and the HIP kernels
// (jpeg_kernels.hip).  Not installed; the public boundary is include/mi355_jpeg.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

struct Geom {
    uint32_t W, H;          // image size as given
    uint32_t W8, H8;        // padded to multiples of 8 (getNearest8x8ImageSize, utils.cpp:184-187)
    uint32_t nbx;           // blocks per row = W8/8
    uint32_t N;             // blocks per frame
    uint32_t tiles;         // ceil(N/64)
    uint32_t flags;         // MI355_F_*
    uint32_t fast_rows;     // 1: W % 8 == 0 and the frame base is 8-byte aligned
    uint64_t frame_stride;  // bytes between frames = W*H*3
};

// Sizes of the device workspace per frame, in elements.
inline size_t coef_dwords(const Geom& g) { return (size_t)g.tiles * 3 * 2048; }
inline size_t unit_off_words(const Geom& g) { return (size_t)g.tiles * 192; }

hipError_t launch_transform(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const double* qd,
                            uint32_t* coefs, int mode, hipStream_t s);
hipError_t launch_probe_samples(const Geom& g, const uint8_t* rgb, uint8_t* samples, hipStream_t s);
hipError_t launch_unit_sizes(const Geom& g, uint32_t n_frames, const uint32_t* coefs,
                             const uint32_t* lut, uint32_t* unit_off, uint32_t* tile_bits,
                             uint32_t* status, hipStream_t s);
hipError_t launch_tile_scan(const Geom& g, uint32_t n_frames, const uint32_t* tile_bits,
                            uint64_t* tile_off, uint8_t* out, uint64_t out_stride,
                            uint64_t* frame_bits, uint32_t* status, hipStream_t s);
hipError_t launch_emit(const Geom& g, uint32_t n_frames, const uint32_t* coefs, const uint32_t* lut,
                       const uint32_t* unit_off, const uint64_t* tile_off, uint8_t* out,
                       uint64_t out_stride, const uint32_t* status, uint32_t lds_words_limit,
                       hipStream_t s);
hipError_t launch_coefs_to_rows(const Geom& g, const uint32_t* coefs, int16_t* rows, hipStream_t s);
hipError_t launch_rows_to_coefs(const Geom& g, const int16_t* rows, uint32_t* coefs, hipStream_t s);
hipError_t launch_unit_bits(const Geom& g, const uint32_t* unit_off, const uint32_t* tile_bits,
                            uint32_t* out, hipStream_t s);

}  // namespace mi355
