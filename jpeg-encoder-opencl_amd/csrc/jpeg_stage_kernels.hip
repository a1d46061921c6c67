// Stage-by-stage kernels: one HIP kernel per stage function of the reference's interface
// (src/utils.hpp:77-137), operating on the reference's own intermediate images -- interleaved u8 pixels
// (ppm_t), interleaved fp64 pixels (ppm_d_t), int[rows][64] block arrays, flat (run, value) pair lists.
//
// These are NOT the fast path (that is the fused pipeline of jpeg_screen_kernels.hip, which never
// materialises any of these intermediates): they exist so that a driver written against the reference's
// header -- JpegEncoderHost's stage sequence, src/OpenCLProject_JpegEncoder.cpp:59-225 -- runs on the GPU
// unmodified, stage by stage, with every intermediate bit-identical to the reference's (the per-stage
// parity tests compare them with the oracle).  Same arithmetic as the reference, operation for operation:
// fp64, unfused (-ffp-contract=off), the glibc cosine table, round-half-away, truncating casts.
#include "jpeg_devfn.h"

namespace mi355 {

// performCSC (utils.cpp:92-110): in place per pixel, doubles, left to right, truncating cast.
__global__ void __launch_bounds__(256) k_stage_csc(uint8_t* __restrict__ img, uint64_t n_px) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_px; i += (uint64_t)gridDim.x * 256) {
        uint8_t* p = img + i * 3;
        const uint32_t r = p[0], g = p[1], b = p[2];
        p[0] = (uint8_t)csc1(r, g, b, csc_k(0, 0), csc_k(0, 1), csc_k(0, 2), csc_k(0, 3));
        p[1] = (uint8_t)csc1(r, g, b, csc_k(1, 0), csc_k(1, 1), csc_k(1, 2), csc_k(1, 3));
        p[2] = (uint8_t)csc1(r, g, b, csc_k(2, 0), csc_k(2, 1), csc_k(2, 2), csc_k(2, 3));
    }
}

// performCDS (utils.cpp:113-141): complete 2x2 quads only (x < W-1, y < H-1), Cb and Cr := floor(sum/4),
// written back to all four pixels; Y untouched.
__global__ void __launch_bounds__(256) k_stage_cds(uint8_t* __restrict__ img, uint32_t W, uint32_t H) {
    const uint32_t qw = W / 2, qh = H / 2;  // quads with x = 2 qx < W - 1  <=>  qx < W / 2
    const uint64_t nq = (uint64_t)qw * qh;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (uint64_t)gridDim.x * 256) {
        const uint32_t qy = (uint32_t)(i / qw), qx = (uint32_t)(i - (uint64_t)qy * qw);
        uint8_t* p00 = img + ((size_t)(2 * qy) * W + 2 * qx) * 3;
        uint8_t* p10 = p00 + 3;
        uint8_t* p01 = p00 + (size_t)W * 3;
        uint8_t* p11 = p01 + 3;
#pragma unroll
        for (int c = 1; c < 3; ++c) {
            const uint8_t m = (uint8_t)((p00[c] + p10[c] + p01[c] + p11[c]) / 4.0);
            p00[c] = p10[c] = p01[c] = p11[c] = m;
        }
    }
}

// copyToLargerImage (utils.cpp:199-208): the W x H image into the top-left corner of a W8 x H8 canvas.
__global__ void __launch_bounds__(256)
    k_stage_copy_larger(const uint8_t* __restrict__ src, uint32_t W, uint32_t H, uint8_t* __restrict__ dst, uint32_t W8) {
    const uint64_t n = (uint64_t)W * H;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i - (uint64_t)y * W);
        const uint8_t* s = src + i * 3;
        uint8_t* d = dst + ((size_t)y * W8 + x) * 3;
        d[0] = s[0], d[1] = s[1], d[2] = s[2];
    }
}

// addReversedPadding (utils.cpp:211-233): right pad x >= oldW from column oldW - 1 - (x - oldW), then bottom
// pad y >= oldH from row oldH - 1 - (y - oldH) over the full padded width (the bottom rows mirror the
// already padded rows): every pad pixel is the original pixel at the mirrored coordinates.
__global__ void __launch_bounds__(256)
    k_stage_mirror(uint8_t* __restrict__ img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH) {
    const uint64_t n = (uint64_t)W8 * H8;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t y = (uint32_t)(i / W8), x = (uint32_t)(i - (uint64_t)y * W8);
        if (x < oldW && y < oldH) continue;
        const uint32_t mx = x < oldW ? x : 2 * oldW - 1 - x, my = y < oldH ? y : 2 * oldH - 1 - y;
        const uint8_t* s = img + ((size_t)my * W8 + mx) * 3;
        uint8_t* d = img + i * 3;
        d[0] = s[0], d[1] = s[1], d[2] = s[2];
    }
}

// copyUIntToDoubleImage (utils.cpp:236-246) and substractfromAll (utils.cpp:190-196), element-wise.
__global__ void __launch_bounds__(256) k_stage_u8_to_f64(const uint8_t* __restrict__ src, double* __restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = (double)src[i];
}
__global__ void __launch_bounds__(256) k_stage_sub(double* __restrict__ img, uint64_t n, double val) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) img[i] -= val;
}

// performDCT (utils.cpp:262-270 -> performDCTBlock :314-348): the in-place ordered fp64 chain on whatever
// doubles the image holds, one lane per (block, channel), 64 doubles in registers (chain<1>, jpeg_devfn.h).
__global__ void __launch_bounds__(64) k_stage_dct(double* __restrict__ img, uint32_t W8, uint32_t H8) {
    const uint32_t nbx = W8 / 8;
    const uint64_t units = (uint64_t)nbx * (H8 / 8) * 3;
    const uint64_t u = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (u >= units) return;
    const uint32_t chan = (uint32_t)(u % 3);
    const uint64_t b = u / 3;
    const uint32_t by = (uint32_t)(b / nbx), bx = (uint32_t)(b - (uint64_t)by * nbx);
    double* base = img + ((size_t)(by * 8) * W8 + bx * 8) * 3 + chan;
    double P[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) P[i] = base[((size_t)(i >> 3) * W8 + (i & 7)) * 3];
    chain<1>(P);
#pragma unroll
    for (int i = 0; i < 64; ++i) base[((size_t)(i >> 3) * W8 + (i & 7)) * 3] = P[i];
}

// performQuantization (utils.cpp:454-467): P = round(P / (double)q[v][u]), half away from zero; luma table for
// the first channel, chroma for the other two.  q: [2][64] doubles, natural order.
__global__ void __launch_bounds__(256)
    k_stage_quant(double* __restrict__ img, uint32_t W8, uint32_t H8, const double* __restrict__ q) {
    const uint64_t n = (uint64_t)W8 * H8 * 3;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t px = i / 3;
        const uint32_t c = (uint32_t)(i - px * 3);
        const uint32_t y = (uint32_t)(px / W8), x = (uint32_t)(px - (uint64_t)y * W8);
        img[i] = __builtin_round(img[i] / q[(c ? 64 : 0) + (y & 7) * 8 + (x & 7)]);
    }
}

// everyMCUisnow2DArray (utils.cpp:482-498): row by * (W8/8) + bx + chan * N, column v * 8 + u, double -> int.
__global__ void __launch_bounds__(256)
    k_stage_blocks(const double* __restrict__ img, uint32_t W8, uint32_t H8, int* __restrict__ lin) {
    const uint64_t n = (uint64_t)W8 * H8 * 3;
    const uint32_t nbx = W8 / 8;
    const uint64_t N = (uint64_t)nbx * (H8 / 8);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t px = i / 3;
        const uint32_t c = (uint32_t)(i - px * 3);
        const uint32_t y = (uint32_t)(px / W8), x = (uint32_t)(px - (uint64_t)y * W8);
        const uint64_t row = (uint64_t)(y / 8) * nbx + x / 8 + (uint64_t)c * N;
        lin[row * 64 + (y & 7) * 8 + (x & 7)] = (int)img[i];
    }
}

// performZigZag (utils.cpp:554-558 -> diagonalZigZagBlockLinear :539-551): the standard zig-zag order.
__global__ void __launch_bounds__(256) k_stage_zigzag(const int* __restrict__ lin, int* __restrict__ zz, uint64_t rows) {
    static constexpr uint8_t kZz[64] = MI355_ZIGZAG_TABLE;
    const uint64_t n = rows * 64;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        zz[i] = lin[(i & ~63ull) + kZz[i & 63]];
}

// performRLE (utils.cpp:612-620 -> RLEBlockAC :572-609): per row, last = highest index with a non-zero
// (search includes index 0); for i = 1..last: zero -> (15,0) on the 16th, else count; non-zero -> (count, value);
// ALWAYS a final (0,0).  pairs: [rows][128] ints (at most 64 pairs), counts: ints used per row.
__global__ void __launch_bounds__(64)
    k_stage_rle(const int* __restrict__ zz, uint64_t rows, int* __restrict__ pairs, uint32_t* __restrict__ counts) {
    const uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    const int* z = zz + r * 64;
    int* out = pairs + r * 128;
    int last = 0;
    for (int i = 63; i >= 0; --i)
        if (z[i] != 0) {
            last = i;
            break;
        }
    uint32_t n = 0;
    int count = 0;
    for (int i = 1; i <= last; ++i) {
        const int v = z[i];
        if (v == 0) {
            if (count == 15) {
                out[n++] = 15, out[n++] = 0;
                count = 0;
            } else {
                ++count;
            }
        } else {
            out[n++] = count, out[n++] = v;
            count = 0;
        }
    }
    out[n++] = 0, out[n++] = 0;
    counts[r] = n;
}

// ---- HuffmanEncoder (utils.cpp:656-698) from the (run, value) pair lists, literally --------------------------
// unit u = 3 * block + chan (the scan order) lives in row chan * N + block.  DC: difference with the previous
// block's zz[row][0] of the same channel (predictors start at 0), narrowed to int16; AC: every pair as given.
template <typename Put>
__device__ __forceinline__ bool stage_unit_symbols(const int* __restrict__ zz, const int* __restrict__ pairs,
                                                   const uint32_t* __restrict__ counts, uint64_t N, uint64_t u,
                                                   const uint32_t* __restrict__ lut, Put&& put) {
    const uint32_t chan = (uint32_t)(u % 3);
    const uint64_t blk = u / 3, row = (uint64_t)chan * N + blk;
    const int dc = zz[row * 64], prev = blk ? zz[(row - 1) * 64] : 0;
    bool ok = put_dc(dc - prev, lut + (chan ? 256 : 0), put);
    const uint32_t* act = lut + (chan ? 768 : 512);
    const int* p = pairs + row * 128;
    const uint32_t n = counts[row];
    for (uint32_t j = 0; j + 1 < n; j += 2) {
        const int run = p[j], v = (int)(int16_t)p[j + 1];
        const int size = bit_size(v);
        const uint32_t e = (run >= 0 && run < 16 && size <= 10) ? act[(run << 4) | size] : 0u;
        if (lut_len(e) == 0) ok = false;  // the reference reads past its tables / appends the text "NULL": an error here
        else put((lut_code(e) << size) | value_bits(v, size), lut_len(e) + size);
    }
    return ok;
}

__global__ void __launch_bounds__(256)
    k_stage_huff_size(const int* __restrict__ zz, const int* __restrict__ pairs, const uint32_t* __restrict__ counts,
                      uint64_t N, const uint32_t* __restrict__ lut, uint32_t* __restrict__ ubits, uint32_t* __restrict__ status) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= 3 * N) return;
    uint32_t bits = 0;
    auto count = [&](uint32_t, uint32_t len) { bits += len; };
    if (!stage_unit_symbols(zz, pairs, counts, N, u, lut, count)) atomicOr(status, 1u);  // MI355_E_CATEGORY
    ubits[u] = bits;
}

// exclusive scan of u32 -> u64 in two levels: chunks of 1024 units
__global__ void __launch_bounds__(1024)
    k_stage_scan_chunks(const uint32_t* __restrict__ ubits, uint64_t n, uint32_t* __restrict__ inchunk, uint64_t* __restrict__ chunk_sum) {
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + tid;
    const uint32_t v = i < n ? ubits[i] : 0u;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += s_wave[w];
    if (i < n) inchunk[i] = pre + incl - v;
    if (tid == 1023) chunk_sum[blockIdx.x] = (uint64_t)pre + incl;
}
__global__ void __launch_bounds__(1024)
    k_stage_scan_tops(uint64_t* __restrict__ chunk_sum, uint64_t chunks, uint64_t* __restrict__ total) {
    __shared__ uint64_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t per = (chunks + 1023) / 1024;
    const uint64_t lo = tid * per < chunks ? tid * per : chunks, hi = lo + per < chunks ? lo + per : chunks;
    uint64_t sum = 0;
    for (uint64_t i = lo; i < hi; ++i) sum += chunk_sum[i];
    uint64_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += s_wave[w];
    uint64_t run = pre + incl - sum;
    for (uint64_t i = lo; i < hi; ++i) {
        const uint64_t c = chunk_sum[i];
        chunk_sum[i] = run;  // exclusive offset of the chunk
        run += c;
    }
    if (tid == 1023) *total = pre + incl;
}
__global__ void __launch_bounds__(256)
    k_stage_huff_emit(const int* __restrict__ zz, const int* __restrict__ pairs, const uint32_t* __restrict__ counts, uint64_t N,
                      const uint32_t* __restrict__ lut, const uint32_t* __restrict__ inchunk, const uint64_t* __restrict__ chunk_off,
                      uint32_t* __restrict__ outw /* zeroed */, uint64_t cap_bits, const uint32_t* __restrict__ status) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= 3 * N || *status) return;
    const uint64_t pos = chunk_off[u >> 10] + inchunk[u];
    BitWriterGlobal bw{outw, 0, (uint32_t)(pos & 31), pos >> 5};
    uint64_t end = pos;
    auto put = [&](uint32_t code, uint32_t len) {
        end += len;
        if (end <= cap_bits) bw.put(code, len);
    };
    (void)stage_unit_symbols(zz, pairs, counts, N, u, lut, put);
    if (end <= cap_bits) bw.flush();
}

// ---- launchers -------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t n, uint32_t block) {
    const uint64_t g = (n + block - 1) / block;
    return (uint32_t)(g < 1 ? 1 : (g > 65535u * 16u ? 65535u * 16u : g));
}
hipError_t launch_stage_csc(uint8_t* img, uint64_t n_px, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_csc, dim3(grid_for(n_px, 256)), dim3(256), 0, s, img, n_px);
    return hipGetLastError();
}
hipError_t launch_stage_cds(uint8_t* img, uint32_t W, uint32_t H, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_cds, dim3(grid_for((uint64_t)(W / 2) * (H / 2), 256)), dim3(256), 0, s, img, W, H);
    return hipGetLastError();
}
hipError_t launch_stage_copy_larger(const uint8_t* src, uint32_t W, uint32_t H, uint8_t* dst, uint32_t W8, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_copy_larger, dim3(grid_for((uint64_t)W * H, 256)), dim3(256), 0, s, src, W, H, dst, W8);
    return hipGetLastError();
}
hipError_t launch_stage_mirror(uint8_t* img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_mirror, dim3(grid_for((uint64_t)W8 * H8, 256)), dim3(256), 0, s, img, W8, H8, oldW, oldH);
    return hipGetLastError();
}
hipError_t launch_stage_u8_to_f64(const uint8_t* src, double* dst, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_u8_to_f64, dim3(grid_for(n, 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_stage_sub(double* img, uint64_t n, double val, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_sub, dim3(grid_for(n, 256)), dim3(256), 0, s, img, n, val);
    return hipGetLastError();
}
hipError_t launch_stage_dct(double* img, uint32_t W8, uint32_t H8, hipStream_t s) {
    const uint64_t units = (uint64_t)(W8 / 8) * (H8 / 8) * 3;
    hipLaunchKernelGGL(k_stage_dct, dim3((uint32_t)((units + 63) / 64)), dim3(64), 0, s, img, W8, H8);
    return hipGetLastError();
}
hipError_t launch_stage_quant(double* img, uint32_t W8, uint32_t H8, const double* q, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_quant, dim3(grid_for((uint64_t)W8 * H8 * 3, 256)), dim3(256), 0, s, img, W8, H8, q);
    return hipGetLastError();
}
hipError_t launch_stage_blocks(const double* img, uint32_t W8, uint32_t H8, int* lin, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_blocks, dim3(grid_for((uint64_t)W8 * H8 * 3, 256)), dim3(256), 0, s, img, W8, H8, lin);
    return hipGetLastError();
}
hipError_t launch_stage_zigzag(const int* lin, int* zz, uint64_t rows, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_zigzag, dim3(grid_for(rows * 64, 256)), dim3(256), 0, s, lin, zz, rows);
    return hipGetLastError();
}
hipError_t launch_stage_rle(const int* zz, uint64_t rows, int* pairs, uint32_t* counts, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_rle, dim3((uint32_t)((rows + 63) / 64)), dim3(64), 0, s, zz, rows, pairs, counts);
    return hipGetLastError();
}
hipError_t launch_stage_huffman(const int* zz, const int* pairs, const uint32_t* counts, uint64_t N, const uint32_t* lut,
                                uint32_t* ubits, uint32_t* inchunk, uint64_t* chunk_sum, uint64_t* total, uint32_t* outw,
                                uint64_t cap_bits, uint32_t* status, hipStream_t s) {
    const uint64_t units = 3 * N, chunks = (units + 1023) / 1024;
    hipLaunchKernelGGL(k_stage_huff_size, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, zz, pairs, counts, N, lut, ubits,
                       status);
    hipLaunchKernelGGL(k_stage_scan_chunks, dim3((uint32_t)chunks), dim3(1024), 0, s, ubits, units, inchunk, chunk_sum);
    hipLaunchKernelGGL(k_stage_scan_tops, dim3(1), dim3(1024), 0, s, chunk_sum, chunks, total);
    hipLaunchKernelGGL(k_stage_huff_emit, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, zz, pairs, counts, N, lut, inchunk,
                       chunk_sum, outw, cap_bits, status);
    return hipGetLastError();
}

}  // namespace mi355
