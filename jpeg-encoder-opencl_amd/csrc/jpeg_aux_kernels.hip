// Auxiliary kernels either side of the hot path: the pinned synthetic
// input generator and JFIF byte stuffing of a finished scan.  Not on the timed path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpeg_device.h"

namespace mi355 {

// ----------------------------------------------------------------------------
// Synthetic frames of SURVEY §8d: bytes d[k] = s_{k+1} >> 24 of the 32-bit LCG
// s <- s*1664525 + 1013904223, s_0 = seed, seed = seed0 + frame.  Each thread jumps
// ahead to its 16-byte group by composing the affine map with itself (square and
// multiply over the bits of the index), then steps 16 times.
// ----------------------------------------------------------------------------
__global__ void k_lcg_fill(uint8_t* __restrict__ dst, uint64_t frame_bytes, uint32_t seed0) {
    const uint64_t group = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k0 = group * 16;
    if (k0 >= frame_bytes) return;
    const uint32_t frame = blockIdx.y;
    uint32_t s = seed0 + frame;
    uint32_t a = 1664525u, c = 1013904223u;  // the map applied 2^i times
    for (uint64_t n = k0; n; n >>= 1) {
        if (n & 1) s = a * s + c;
        c = c * (a + 1u);
        a = a * a;
    }
    uint8_t* out = dst + (uint64_t)frame * frame_bytes + k0;
    const uint32_t cnt = frame_bytes - k0 < 16 ? (uint32_t)(frame_bytes - k0) : 16u;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s = s * 1664525u + 1013904223u;
        w[i >> 2] |= (s >> 24) << (8 * (i & 3));
    }
    if (cnt == 16 && (((uintptr_t)out) & 15u) == 0) {
        *reinterpret_cast<uint4*>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint32_t i = 0; i < cnt; ++i) out[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
}

hipError_t launch_lcg_fill(uint8_t* dst, uint64_t frame_bytes, uint32_t n_frames, uint32_t seed0, hipStream_t s) {
    const uint64_t groups = (frame_bytes + 15) / 16;
    dim3 grid((unsigned)((groups + 255) / 256), n_frames);
    hipLaunchKernelGGL(k_lcg_fill, grid, dim3(256), 0, s, dst, frame_bytes, seed0);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------
// Byte stuffing of one frame's scan: final partial byte padded with
// 1s, every 0xFF followed by 0x00.  Three launches: count 0xFF per 4096-byte chunk, scan
// the chunk counts (one workgroup), scatter.
// ----------------------------------------------------------------------------
constexpr uint32_t kStuffChunk = 4096;  // bytes per workgroup (256 threads x 16)

__device__ __forceinline__ uint32_t scan_byte(const uint8_t* __restrict__ in, uint64_t i, uint64_t nbytes, uint64_t nbits) {
    uint32_t b = in[i];
    if (i == nbytes - 1 && (nbits & 7)) b |= 0xFFu >> (nbits & 7);
    return b;
}

// Restart intervals (MI355_F_RESTART): tile t >= 1 starts at bit tile_off[t], a multiple of 8, and the marker
// RST((t-1) & 7) goes in front of its first byte.  First tile in [1, tiles) that starts at or after `bit`.
__device__ __forceinline__ uint32_t first_tile_from(const uint64_t* __restrict__ tile_off, uint32_t tiles, uint64_t bit) {
    uint32_t lo = 1, hi = tiles;  // answer in [lo, hi]
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_off[mid] >= bit) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
    k_stuff_count(const uint8_t* __restrict__ in, const uint64_t* __restrict__ nbits_p, uint32_t* __restrict__ counts,
                  const uint64_t* __restrict__ tile_off, uint32_t tiles) {
    __shared__ uint32_t s_sum[4];
    const uint64_t nbits = *nbits_p, nbytes = (nbits + 7) / 8;
    const uint64_t i0 = (uint64_t)blockIdx.x * kStuffChunk + threadIdx.x * 16;
    uint32_t n = 0;
    for (uint32_t j = 0; j < 16; ++j)
        if (i0 + j < nbytes && scan_byte(in, i0 + j, nbytes, nbits) == 0xFFu) ++n;
    if (tile_off && i0 < nbytes) {  // two marker bytes per tile that starts among this thread's bytes
        const uint64_t i1 = i0 + 16 < nbytes ? i0 + 16 : nbytes;
        n += 2u * (first_tile_from(tile_off, tiles, 8 * i1) - first_tile_from(tile_off, tiles, 8 * i0));
    }
    for (int d = 32; d; d >>= 1) n += __shfl_xor(n, d);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of `n` chunk counts (u32 -> u64 offsets); *total = stuffed length in bytes
__global__ void __launch_bounds__(1024)
    k_stuff_scan(const uint32_t* __restrict__ counts, uint64_t* __restrict__ offs, uint32_t n,
                 uint64_t* __restrict__ total, const uint64_t* __restrict__ nbits_p) {
    __shared__ uint64_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += counts[i];
    uint64_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += s_wave[w];
    uint64_t run = pre + incl - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        offs[i] = run;
        run += counts[i];
    }
    if (tid == 1023) *total = pre + incl + (*nbits_p + 7) / 8;
}

__global__ void __launch_bounds__(256)
    k_stuff_write(const uint8_t* __restrict__ in, const uint64_t* __restrict__ nbits_p, const uint64_t* __restrict__ offs,
                  uint8_t* __restrict__ out, uint64_t cap, uint32_t* __restrict__ status,
                  const uint64_t* __restrict__ tile_off, uint32_t tiles) {
    __shared__ uint32_t s_wave[4];
    const uint64_t nbits = *nbits_p, nbytes = (nbits + 7) / 8;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kStuffChunk + tid * 16;
    uint32_t b[16], n = 0;
    for (uint32_t j = 0; j < 16; ++j) {
        b[j] = i0 + j < nbytes ? scan_byte(in, i0 + j, nbytes, nbits) : 0u;
        n += (i0 + j < nbytes && b[j] == 0xFFu) ? 1u : 0u;
    }
    uint32_t next_tile = tiles;  // next tile whose start is at or after this thread's first byte
    if (tile_off && i0 < nbytes) {
        const uint64_t i1 = i0 + 16 < nbytes ? i0 + 16 : nbytes;
        next_tile = first_tile_from(tile_off, tiles, 8 * i0);
        n += 2u * (first_tile_from(tile_off, tiles, 8 * i1) - next_tile);
    }
    uint32_t incl = n;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += s_wave[w];
    uint64_t o = i0 + offs[blockIdx.x] + pre + incl - n;  // output position of this thread's first byte
    for (uint32_t j = 0; j < 16; ++j) {
        if (i0 + j >= nbytes) break;
        if (o + 4 > cap) {
            atomicOr(status, 2u);  // MI355_E_CAPACITY
            return;
        }
        if (next_tile < tiles && tile_off[next_tile] == 8 * (i0 + j)) {  // RSTm in front of the tile's first byte
            out[o++] = 0xFF;
            out[o++] = (uint8_t)(0xD0u + ((next_tile - 1u) & 7u));
            ++next_tile;
        }
        out[o++] = (uint8_t)b[j];
        if (b[j] == 0xFFu) out[o++] = 0;
    }
}

hipError_t launch_stuff(const uint8_t* in, const uint64_t* d_nbits, uint64_t max_bytes, uint32_t* counts,
                        uint64_t* offs, uint64_t* d_total, uint8_t* out, uint64_t cap, uint32_t* status,
                        const uint64_t* tile_off, uint32_t tiles, hipStream_t s) {
    const uint32_t chunks = (uint32_t)((max_bytes + kStuffChunk - 1) / kStuffChunk);
    if (chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stuff_count, dim3(chunks), dim3(256), 0, s, in, d_nbits, counts, tile_off, tiles);
    hipLaunchKernelGGL(k_stuff_scan, dim3(1), dim3(1024), 0, s, counts, offs, chunks, d_total, d_nbits);
    hipLaunchKernelGGL(k_stuff_write, dim3(chunks), dim3(256), 0, s, in, d_nbits, offs, out, cap, status, tile_off,
                       tiles);
    return hipGetLastError();
}

}  // namespace mi355
