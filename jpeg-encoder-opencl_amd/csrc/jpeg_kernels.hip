// Hand-written HIP kernels (gfx950 / CDNA4) for the strict JPEG encode path.
//
// Pipeline per batch of frames (all device resident):
//   k_transform  RGB u8 -> quantised zig-zag int16 coefficients          (fp64-VALU bound)
//                fuses performCSC, performCDS, copyToLargerImage/addReversedPadding,
//                copyUIntToDoubleImage, substractfromAll, performDCT, performQuantization,
//                everyMCUisnow2DArray, performZigZag        (reference src/utils.cpp:92-558)
//   k_unit_sizes per-unit bit counts + tile-local exclusive offsets     (performRLE + code lengths)
//   k_tile_scan  exclusive scan of the tile sums per frame (64-bit)
//   k_emit       bit-string emission at the scanned offsets              (HuffmanEncoder)
//
// Vocabulary: a *block* is one 8x8 pixel block; a *unit* is one (block, channel)
// pair = one row of the reference's zigzag_arr; a *tile* is 64 consecutive blocks
// of a frame in raster order (192 units, contiguous in the scan order
// 3*block+chan of utils.cpp:665-695).
//
// Floating point: the transform must reproduce the reference's unfused, strictly
// ordered fp64 arithmetic bit for bit => this file is compiled with
// -ffp-contract=off and never uses fma() outside the compiler's own correctly
// rounded division.
#include "jpeg_devfn.h"

namespace mi355 {

// ----------------------------------------------------------------------------
// k_transform: one wave per (tile, channel); lane = block.
// grid.x = tiles*3 (XCD-aware decode below), grid.y = frame.
// coefs layout: [frame][tile][chan][32 coefficient pairs][64 lanes] uint32,
//   low half = zig-zag coefficient 2p, high half = 2p+1 (int16 each), so that
//   the wave's stores and the entropy kernels' loads are 256-byte coalesced rows.
// ----------------------------------------------------------------------------
template <int MODE, bool PROBE>
__global__ void __launch_bounds__(64)
    k_transform(Geom g, const uint8_t* __restrict__ rgb, const double* __restrict__ qd,
                uint32_t* __restrict__ coefs, uint8_t* __restrict__ probe_samples) {
    __shared__ uint32_t lds[16 * 64];
    const uint32_t lane = threadIdx.x;
    // XCD-aware decode: consecutive workgroup ids are dealt round-robin over the 8
    // XCDs, so ids i, i+8, i+16 (same XCD) get the three channels of one tile and
    // share its RGB bytes in that XCD's L2.  Speed only; any mapping is correct.
    uint32_t id = blockIdx.x, tile, chan;
    {
        uint32_t full = (g.tiles / 8) * 24;  // ids covered by complete groups of 8 tiles
        if (id < full) {
            uint32_t grp = id / 24, w = id % 24;
            tile = grp * 8 + (w & 7);
            chan = w >> 3;
        } else {
            uint32_t r = id - full;
            tile = (g.tiles / 8) * 8 + r / 3;
            chan = r % 3;
        }
    }
    const uint32_t frame = blockIdx.y;
    const uint8_t* f = rgb + (size_t)frame * g.frame_stride;
    const uint32_t b = tile * 64 + lane;
    const bool active = b < g.N;
    const uint32_t bc = active ? b : g.N - 1;
    const uint32_t by = bc / g.nbx, bx = bc - by * g.nbx;
    const bool avg = (chan != 0) && (g.flags & 1u);

    // wave-uniform choice of the fast loader
    bool interior = (bx * 8 + 8 <= g.W) && (by * 8 + 8 <= g.H);
    const bool fast = g.fast_rows && __all(interior);

    uint32_t pk[16];
    if (chan == 0) {
        if (fast)
            load_samples<true>(f, g, 0, false, bx, by, lane, lds, pk);
        else
            load_samples<false>(f, g, 0, false, bx, by, lane, lds, pk);
    } else if (chan == 1) {
        if (fast)
            load_samples<true>(f, g, 1, avg, bx, by, lane, lds, pk);
        else
            load_samples<false>(f, g, 1, avg, bx, by, lane, lds, pk);
    } else {
        if (fast)
            load_samples<true>(f, g, 2, avg, bx, by, lane, lds, pk);
        else
            load_samples<false>(f, g, 2, avg, bx, by, lane, lds, pk);
    }

    if constexpr (PROBE) {
        // stage probe: the padded YCbCr image, interleaved like the reference's ppm_t
        if (active) {
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                uint32_t v = (pk[s >> 2] >> (8 * (s & 3))) & 255u;
                size_t px = (size_t)(by * 8 + (s >> 3)) * g.W8 + bx * 8 + (s & 7);
                probe_samples[((size_t)frame * g.W8 * g.H8 + px) * 3 + chan] = (uint8_t)v;
            }
        }
        return;
    }

    // level shift (substractfromAll(...,128.0), all three channels, quirk Q4)
    double P[64];
#pragma unroll
    for (int s = 0; s < 64; ++s)
        P[s] = (double)((int)((pk[s >> 2] >> (8 * (s & 3))) & 255u) - 128);

    chain<MODE>(P);

    // performQuantization (utils.cpp:457-463): correctly rounded divide, round half
    // away from zero; luma table for channel 0, chroma for 1 and 2.
    const double* q = qd + (chan ? 64 : 0);
    int qi[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) qi[i] = (int)__builtin_round(P[i] / q[i]);

    uint32_t* dst = coefs + (((size_t)frame * g.tiles + tile) * 3 + chan) * 2048 + lane;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        uint32_t lo = (uint32_t)qi[zigzag_nat(2 * p)] & 0xffffu;
        uint32_t hi = (uint32_t)qi[zigzag_nat(2 * p + 1)] << 16;
        dst[p * 64] = active ? (lo | hi) : 0u;
    }
}

// ----------------------------------------------------------------------------
// k_unit_sizes: workgroup = one tile = 3 waves (wave = channel), lane = block.
// Writes unit_off[frame][tile][192] (exclusive, tile-local, scan order
// 3*block+chan) and tile_bits[frame][tile].
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(192)
    k_unit_sizes(Geom g, const uint32_t* __restrict__ coefs, const uint32_t* __restrict__ lut,
                 uint32_t* __restrict__ unit_off, uint32_t* __restrict__ tile_bits,
                 uint32_t* __restrict__ status) {
    __shared__ uint32_t s_lut[2][256];
    __shared__ uint32_t s_dc[2][16];
    __shared__ uint32_t s_bits[192];
    const uint32_t tid = threadIdx.x, lane = tid & 63, chan = tid >> 6;
    const uint32_t tile = blockIdx.x, frame = blockIdx.y;
    for (uint32_t i = tid; i < 512; i += 192) s_lut[i >> 8][i & 255] = lut[512 + i];  // AC luma, AC chroma
    if (tid < 32) s_dc[tid >> 4][tid & 15] = lut[(tid >> 4) * 256 + (tid & 15)];       // DC luma, DC chroma
    __syncthreads();

    const uint32_t* cf = coefs + (size_t)frame * g.tiles * 3 * 2048;
    const uint32_t b = tile * 64 + lane;
    const bool active = b < g.N;
    uint32_t c[32];
    load_unit(cf + ((size_t)tile * 3 + chan) * 2048 + lane, c);
    const int dc = (int)(int16_t)(c[0] & 0xffffu);
    const int pred = dc_predictor(cf, tile, chan, lane, dc);

    uint32_t bits = 0;
    auto count = [&](uint32_t, uint32_t len) { bits += len; };
    bool ok = put_dc(dc - pred, s_dc[chan ? 1 : 0], count);
    ok &= walk_ac(c, s_lut[chan ? 1 : 0], count);
    if (!active) {
        bits = 0;
        ok = true;
    }
    if (!ok) atomicOr(status, 1u);  // MI355_E_CATEGORY

    s_bits[lane * 3 + chan] = bits;
    __syncthreads();
    if (tid < 64) {
        uint32_t a0 = s_bits[tid * 3], a1 = s_bits[tid * 3 + 1], a2 = s_bits[tid * 3 + 2];
        uint32_t blk = a0 + a1 + a2, incl = blk;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t n = __shfl_up(incl, d);
            if ((int)tid >= d) incl += n;
        }
        uint32_t excl = incl - blk;
        uint32_t* uo = unit_off + ((size_t)frame * g.tiles + tile) * 192 + tid * 3;
        uo[0] = excl;
        uo[1] = excl + a0;
        uo[2] = excl + a0 + a1;
        if (tid == 63) tile_bits[(size_t)frame * g.tiles + tile] = incl;
    }
}

// ----------------------------------------------------------------------------
// k_tile_scan: one workgroup per frame (NT = 256 threads, 1024 for frames with many tiles);
// exclusive 64-bit scan of the tile sums, 4 NT tiles at a time: thread t takes tiles base + 4t .. + 3, so the sums are
// read and the offsets written with (nearly) coalesced accesses (the next chunk's sums are requested before the current
// chunk is scanned), a 32-bit DPP scan inside each wave, the wave totals through LDS, a 64-bit carry from chunk to chunk.
// (Round 1 gave every thread a contiguous run of tiles: 64 dependent strided loads per thread on a 16384 x 16384 frame,
// 0.19 ms = 8 % of that call.)  Also zeroes every output word that two tiles share (the emit/merge kernels OR into
// those), writes the frame's bit count, checks the caller's capacity and -- for the screened pipeline, whose encode
// kernel accumulates the tile sums with atomics -- re-arms the sums and counters.  The 256-thread form is
// one wave per SIMD with few registers: it fits on a CU next to two resident workgroups of
// k_screen_encode of another stream (a 1024-thread workgroup has to wait for a free slot).
// A chunk's sums fit 32 bits: a tile is at most 384 units of at most 64 symbols of at most 27 bits (< 2^20 bits), a chunk 4096 tiles.
// ----------------------------------------------------------------------------
template <uint32_t NT>
__global__ void __launch_bounds__(NT)
    k_tile_scan(Geom g, uint32_t* __restrict__ tile_bits, uint64_t* __restrict__ tile_off,
                uint8_t* __restrict__ out, uint64_t out_stride, uint64_t* __restrict__ frame_bits,
                uint32_t* __restrict__ status, uint32_t* __restrict__ reset_counters, uint32_t rearm_tiles,
                uint32_t flag_frames) {
    constexpr uint32_t kPer = 4;  // consecutive tiles per thread and chunk
    __shared__ uint32_t s_wave[NT / 64];
    __shared__ uint32_t s_poison;
    if (threadIdx.x == 0) s_poison = 0;
    uint32_t poison = 0;  // bit 31 of a tile total: a unit of that tile had an error (k_screen_encode, k_dc_heads)
    __builtin_amdgcn_s_setprio(3);  // short, on the stream's critical path, resident next to encode kernels
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t frame = blockIdx.x;
    uint32_t* tb = tile_bits + (size_t)frame * g.tiles;
    uint64_t* to = tile_off + (size_t)frame * (g.tiles + 1);
    uint32_t* outw = reinterpret_cast<uint32_t*>(out + (size_t)frame * out_stride);
    // restart intervals (standard mode, MI355_F_RESTART): every tile starts on a byte boundary
    const uint32_t pad = (g.flags & 8u) ? 7u : 0u;
    uint64_t carry = 0;  // bits of all tiles before this chunk (the same in every thread)
    uint32_t next[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) next[k] = tid * kPer + k < g.tiles ? tb[tid * kPer + k] : 0u;
    for (uint32_t base = 0; base < g.tiles; base += NT * kPer) {  // uniform trip count: DPP scans need every lane
        const uint32_t i0 = base + tid * kPer;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            poison |= next[k];
            v[k] = i0 + k < g.tiles ? (((next[k] & 0x7fffffffu) + pad) & ~pad) : 0u;
            sum += v[k];
            const uint32_t in = i0 + NT * kPer + k;
            next[k] = in < g.tiles ? tb[in] : 0u;
        }
        const uint32_t incl = wave_incl_scan(sum, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t pre = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < NT / 64; ++w) {
            const uint32_t t = s_wave[w];
            total += t;
            if (w < wave) pre += t;
        }
        __syncthreads();  // s_wave is rewritten by the next chunk
        uint64_t run = carry + pre + (incl - sum);  // exclusive offset of tile i0
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t i = i0 + k;
            if (i < g.tiles) {
                to[i] = run;
                // word shared with the previous tile: both sides OR into it
                if (i > 0 && (run & 31) && (run >> 5) * 4 + 4 <= out_stride) outw[run >> 5] = 0;
                if (rearm_tiles) tb[i] = 0;
            }
            run += v[k];
        }
        carry += total;
    }
    if (poison & 0x80000000u) atomicOr(&s_poison, 1u);
    __syncthreads();
    if (tid == 0) {
        to[g.tiles] = carry;
        const bool over = ((carry + 31) >> 5) * 4 > out_stride;
        if (over) atomicOr(status, 2u);  // MI355_E_CAPACITY
        // Per-frame verdict: a frame with a poisoned tile total (a coefficient without a code) gets the bit count
        // kBitsCategory, one that does not fit its output slot kBitsCapacity; both are skipped by the merge, the other
        // frames of the call are complete and valid.
        frame_bits[frame] = !flag_frames ? carry : (s_poison ? kBitsCategory : (over ? kBitsCapacity : carry));
        // the screened pipeline's arena counter is consumed by now: re-arm it
        if (reset_counters && frame == 0) reset_counters[0] = 0;
    }
}

// ----------------------------------------------------------------------------
// Frames above 8192 tiles: the same scan by several workgroups per frame, in two launches.  k_tile_chunks: workgroup
// (c, frame) scans chunk c (4096 tiles: 1024 threads x 4) on its own -- offsets relative to the chunk into `tile_off`,
// the chunk's total into `chunk_tot` -- and re-arms the sums; k_tile_fix adds the totals of the chunks in front (at
// most 16 of them), zeroes the shared words, and the last chunk's first thread closes the frame.  One workgroup for
// all 65,536 tiles of a 16384 x 16384 frame is bound by its own stores: 0.084 ms; this pair: see DESIGN.md.
// ----------------------------------------------------------------------------
constexpr uint32_t kChunkTiles = 4096;
__global__ void __launch_bounds__(1024)
    k_tile_chunks(Geom g, uint32_t* __restrict__ tile_bits, uint64_t* __restrict__ tile_off,
                  uint64_t* __restrict__ chunk_tot, uint32_t chunks, uint32_t rearm_tiles) {
    __shared__ uint32_t s_wave[16];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = blockIdx.x, frame = blockIdx.y;
    uint32_t* tb = tile_bits + (size_t)frame * g.tiles;
    uint64_t* to = tile_off + (size_t)frame * (g.tiles + 1);
    const uint32_t pad = (g.flags & 8u) ? 7u : 0u;
    const uint32_t i0 = c * kChunkTiles + tid * 4;
    uint32_t v[4], sum = 0;
    __shared__ uint32_t s_poison;
    if (tid == 0) s_poison = 0;
    uint32_t poison = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t raw = i0 + k < g.tiles ? tb[i0 + k] : 0u;
        poison |= raw;
        v[k] = i0 + k < g.tiles ? (((raw & 0x7fffffffu) + pad) & ~pad) : 0u;
        sum += v[k];
    }
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (poison & 0x80000000u) atomicOr(&s_poison, 1u);
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
        const uint32_t t = s_wave[w];
        total += t;
        if (w < wave) pre += t;
    }
    uint32_t run = pre + (incl - sum);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (i0 + k < g.tiles) {
            to[i0 + k] = run;  // relative to the chunk; k_tile_fix makes it absolute
            if (rearm_tiles) tb[i0 + k] = 0;
        }
        run += v[k];
    }
    __syncthreads();
    if (tid == 0) chunk_tot[(size_t)frame * chunks + c] = total | ((uint64_t)s_poison << 63);  // bit 63: a poisoned tile in the chunk
}
__global__ void __launch_bounds__(1024)
    k_tile_fix(Geom g, uint64_t* __restrict__ tile_off, const uint64_t* __restrict__ chunk_tot, uint32_t chunks,
               uint8_t* __restrict__ out, uint64_t out_stride, uint64_t* __restrict__ frame_bits,
               uint32_t* __restrict__ status, uint32_t* __restrict__ reset_counters, uint32_t flag_frames) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x, frame = blockIdx.y;
    uint64_t* to = tile_off + (size_t)frame * (g.tiles + 1);
    uint32_t* outw = reinterpret_cast<uint32_t*>(out + (size_t)frame * out_stride);
    const uint64_t* ct = chunk_tot + (size_t)frame * chunks;
    constexpr uint64_t kTot = ~(1ull << 63);
    uint64_t before = 0;
    for (uint32_t k = 0; k < c; ++k) before += ct[k] & kTot;
    const uint32_t i0 = c * kChunkTiles + tid * 4;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = i0 + k;
        if (i < g.tiles) {
            const uint64_t run = before + to[i];
            to[i] = run;
            // word shared with the previous tile: both sides OR into it
            if (i > 0 && (run & 31) && (run >> 5) * 4 + 4 <= out_stride) outw[run >> 5] = 0;
        }
    }
    if (c + 1 == chunks && tid == 0) {
        const uint64_t total = before + (ct[c] & kTot);
        to[g.tiles] = total;
        const bool over = ((total + 31) >> 5) * 4 > out_stride;
        if (over) atomicOr(status, 2u);  // MI355_E_CAPACITY
        uint64_t poison = 0;  // see k_tile_scan
        for (uint32_t k = 0; k < chunks; ++k) poison |= ct[k];
        frame_bits[frame] = !flag_frames ? total : ((poison >> 63) ? kBitsCategory : (over ? kBitsCapacity : total));
        if (reset_counters && frame == 0) reset_counters[0] = 0;
    }
}

// ----------------------------------------------------------------------------
// k_emit: workgroup = one tile (3 waves = 3 channels, lane = block).  The tile's
// 192 units are contiguous in the scan, so its bits are assembled in LDS
// (big-endian 32-bit words, LDS atomic OR) and written out as coalesced words;
// only the first / last word can be shared with a neighbour tile (global atomic
// OR into words zeroed by k_tile_scan).  A tile whose bits do not fit the LDS
// window takes the direct-to-global path.
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(192)
    k_emit(Geom g, const uint32_t* __restrict__ coefs, const uint32_t* __restrict__ lut,
           const uint32_t* __restrict__ unit_off, const uint64_t* __restrict__ tile_off,
           uint8_t* __restrict__ out, uint64_t out_stride, const uint32_t* __restrict__ status,
           uint32_t lds_words_limit) {
    __shared__ uint32_t s_lut[2][256];
    __shared__ uint32_t s_dc[2][16];
    __shared__ uint32_t s_words[kEmitLdsWords];
    const uint32_t tid = threadIdx.x, lane = tid & 63, chan = tid >> 6;
    const uint32_t tile = blockIdx.x, frame = blockIdx.y;
    if (*status) return;  // a size had no code / capacity exceeded: nothing is emitted

    const uint64_t* to = tile_off + (size_t)frame * (g.tiles + 1);
    const uint64_t start = to[tile], end = to[tile + 1];
    const uint64_t w0 = start >> 5;                  // first output word of this tile
    const uint32_t nw = (uint32_t)(((end + 31) >> 5) - w0);  // words touched
    const bool use_lds = nw <= lds_words_limit;
    uint32_t* outw = reinterpret_cast<uint32_t*>(out + (size_t)frame * out_stride);
    const bool last_tile = tile + 1 == g.tiles;

    for (uint32_t i = tid; i < 512; i += 192) s_lut[i >> 8][i & 255] = lut[512 + i];
    if (tid < 32) s_dc[tid >> 4][tid & 15] = lut[(tid >> 4) * 256 + (tid & 15)];
    if (use_lds) {
        for (uint32_t i = tid; i < nw; i += 192) s_words[i] = 0;
    } else {
        // direct path: every word of the tile is ORed into, so zero the words this
        // tile owns exclusively (shared boundary words were zeroed by k_tile_scan)
        for (uint32_t i = tid; i < nw; i += 192) {
            bool shared = (i == 0 && (start & 31)) || (i == nw - 1 && (end & 31) && !last_tile);
            if (!shared) __hip_atomic_store(&outw[w0 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();

    const uint32_t* cf = coefs + (size_t)frame * g.tiles * 3 * 2048;
    const uint32_t b = tile * 64 + lane;
    if (b < g.N) {
        uint32_t c[32];
        load_unit(cf + ((size_t)tile * 3 + chan) * 2048 + lane, c);
        const int dc = (int)(int16_t)(c[0] & 0xffffu);
        const int pred = dc_predictor(cf, tile, chan, lane, dc);
        const uint32_t off = unit_off[((size_t)frame * g.tiles + tile) * 192 + lane * 3 + chan];
        const uint64_t pos = (start & 31) + off;  // bit position relative to word w0
        if (use_lds) {
            BitWriterLds bw{s_words, 0, (uint32_t)(pos & 31), (uint32_t)(pos >> 5)};
            auto put = [&](uint32_t code, uint32_t len) { bw.put(code, len); };
            put_dc(dc - pred, s_dc[chan ? 1 : 0], put);
            walk_ac(c, s_lut[chan ? 1 : 0], put);
            bw.flush();
        } else {
            BitWriterGlobal bw{outw, 0, (uint32_t)(pos & 31), w0 + (pos >> 5)};
            auto put = [&](uint32_t code, uint32_t len) { bw.put(code, len); };
            put_dc(dc - pred, s_dc[chan ? 1 : 0], put);
            walk_ac(c, s_lut[chan ? 1 : 0], put);
            bw.flush();
        }
    }
    if (!use_lds) return;
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += 192) {
        uint32_t v = __builtin_bswap32(s_words[i]);
        bool shared = (i == 0 && (start & 31)) || (i == nw - 1 && (end & 31) && !last_tile);
        if (shared) {
            if (v) atomicOr(&outw[w0 + i], v);
        } else {
            outw[w0 + i] = v;
        }
    }
}

// ----------------------------------------------------------------------------
// layout converters for the stage probes (not on the hot path)
// ----------------------------------------------------------------------------
// tiled coefs -> reference row order int16 [chan*N + block][64]
__global__ void k_coefs_to_rows(Geom g, const uint32_t* __restrict__ coefs, int16_t* __restrict__ rows) {
    uint32_t unit = blockIdx.x * 4 + (threadIdx.x >> 6);  // chan*N + block
    uint32_t k = threadIdx.x & 63;
    if (unit >= g.passes * g.N) return;
    uint32_t tile, pass, lane;
    if (g.passes == 6) {
        // 4:2:0 row order: luma 4*mcu + k, then Cb at 4M + mcu, Cr at 5M + mcu; in the tiled workspace a
        // tile's 256 luma units are passes 0..3 back to back in that same order
        if (unit < 4 * g.N) {
            tile = unit >> 8, pass = (unit >> 6) & 3, lane = unit & 63;
        } else {
            uint32_t c = unit / g.N, m = unit - c * g.N;  // c = 4 (Cb), 5 (Cr)
            tile = m >> 6, pass = c, lane = m & 63;
        }
    } else {
        uint32_t b = unit % g.N;
        pass = unit / g.N, tile = b >> 6, lane = b & 63;
    }
    uint32_t w = coefs[((size_t)tile * g.passes + pass) * 2048 + (k >> 1) * 64 + lane];
    rows[(size_t)unit * 64 + k] = (int16_t)((k & 1) ? (w >> 16) : (w & 0xffffu));
}
// reference row order int16 -> tiled coefs
__global__ void k_rows_to_coefs(Geom g, const int16_t* __restrict__ rows, uint32_t* __restrict__ coefs) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // over tiles*3*2048 dwords
    if (idx >= g.tiles * 3 * 2048) return;
    uint32_t lane = idx & 63, p = (idx >> 6) & 31, tc = idx >> 11;
    uint32_t tile = tc / 3, chan = tc - tile * 3;
    uint32_t b = tile * 64 + lane;
    uint32_t v = 0;
    if (b < g.N) {
        const int16_t* r = rows + ((size_t)chan * g.N + b) * 64;
        v = ((uint32_t)(uint16_t)r[2 * p]) | ((uint32_t)(uint16_t)r[2 * p + 1] << 16);
    }
    coefs[idx] = v;
}
// unit_off (tile-local exclusive) + tile_bits -> per-unit bit counts, scan order
__global__ void k_unit_bits(Geom g, const uint32_t* __restrict__ unit_off,
                            const uint32_t* __restrict__ tile_bits, uint32_t* __restrict__ out) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * g.N) return;
    uint32_t tile = idx / 192, j = idx - tile * 192;
    uint32_t last = (tile + 1 == g.tiles) ? (3 * g.N - tile * 192) : 192;
    uint32_t nxt = (j + 1 < last) ? unit_off[(size_t)tile * 192 + j + 1] : tile_bits[tile];
    out[idx] = nxt - unit_off[(size_t)tile * 192 + j];
}

// ----------------------------------------------------------------------------
// launchers (called from mi355_jpeg.cpp)
// ----------------------------------------------------------------------------
hipError_t launch_transform(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const double* qd,
                            uint32_t* coefs, int mode, hipStream_t s) {
    dim3 grid(g.tiles * 3, n_frames);
    if (mode == 0)
        hipLaunchKernelGGL((k_transform<0, false>), grid, dim3(64), 0, s, g, rgb, qd, coefs, nullptr);
    else
        hipLaunchKernelGGL((k_transform<1, false>), grid, dim3(64), 0, s, g, rgb, qd, coefs, nullptr);
    return hipGetLastError();
}
hipError_t launch_probe_samples(const Geom& g, const uint8_t* rgb, uint8_t* samples, hipStream_t s) {
    dim3 grid(g.tiles * 3, 1);
    hipLaunchKernelGGL((k_transform<1, true>), grid, dim3(64), 0, s, g, rgb, nullptr, nullptr, samples);
    return hipGetLastError();
}
hipError_t launch_unit_sizes(const Geom& g, uint32_t n_frames, const uint32_t* coefs,
                             const uint32_t* lut, uint32_t* unit_off, uint32_t* tile_bits,
                             uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(k_unit_sizes, dim3(g.tiles, n_frames), dim3(192), 0, s, g, coefs, lut, unit_off,
                       tile_bits, status);
    return hipGetLastError();
}
hipError_t launch_tile_scan(const Geom& g, uint32_t n_frames, uint32_t* tile_bits,
                            uint64_t* tile_off, uint8_t* out, uint64_t out_stride,
                            uint64_t* frame_bits, uint32_t* status, uint32_t* reset_counters,
                            bool rearm_tiles, uint64_t* chunk_tot, bool flag_frames, hipStream_t s) {
    if (g.tiles <= 8192) {
        hipLaunchKernelGGL(k_tile_scan<256>, dim3(n_frames), dim3(256), 0, s, g, tile_bits, tile_off, out,
                           out_stride, frame_bits, status, reset_counters, rearm_tiles ? 1u : 0u, flag_frames ? 1u : 0u);
    } else if (chunk_tot) {  // several workgroups per frame, two launches
        const uint32_t chunks = scan_chunks(g);
        hipLaunchKernelGGL(k_tile_chunks, dim3(chunks, n_frames), dim3(1024), 0, s, g, tile_bits, tile_off, chunk_tot,
                           chunks, rearm_tiles ? 1u : 0u);
        hipLaunchKernelGGL(k_tile_fix, dim3(chunks, n_frames), dim3(1024), 0, s, g, tile_off, chunk_tot, chunks, out,
                           out_stride, frame_bits, status, reset_counters, flag_frames ? 1u : 0u);
    } else {
        hipLaunchKernelGGL(k_tile_scan<1024>, dim3(n_frames), dim3(1024), 0, s, g, tile_bits, tile_off, out,
                           out_stride, frame_bits, status, reset_counters, rearm_tiles ? 1u : 0u, flag_frames ? 1u : 0u);
    }
    return hipGetLastError();
}
hipError_t launch_emit(const Geom& g, uint32_t n_frames, const uint32_t* coefs, const uint32_t* lut,
                       const uint32_t* unit_off, const uint64_t* tile_off, uint8_t* out,
                       uint64_t out_stride, const uint32_t* status, uint32_t lds_words_limit,
                       hipStream_t s) {
    if (lds_words_limit > kEmitLdsWords) lds_words_limit = kEmitLdsWords;
    hipLaunchKernelGGL(k_emit, dim3(g.tiles, n_frames), dim3(192), 0, s, g, coefs, lut, unit_off,
                       tile_off, out, out_stride, status, lds_words_limit);
    return hipGetLastError();
}
hipError_t launch_coefs_to_rows(const Geom& g, const uint32_t* coefs, int16_t* rows, hipStream_t s) {
    hipLaunchKernelGGL(k_coefs_to_rows, dim3((g.passes * g.N + 3) / 4), dim3(256), 0, s, g, coefs, rows);
    return hipGetLastError();
}
hipError_t launch_rows_to_coefs(const Geom& g, const int16_t* rows, uint32_t* coefs, hipStream_t s) {
    uint32_t n = g.tiles * 3 * 2048;
    hipLaunchKernelGGL(k_rows_to_coefs, dim3((n + 255) / 256), dim3(256), 0, s, g, rows, coefs);
    return hipGetLastError();
}
hipError_t launch_unit_bits(const Geom& g, const uint32_t* unit_off, const uint32_t* tile_bits,
                            uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_unit_bits, dim3((3 * g.N + 255) / 256), dim3(256), 0, s, g, unit_off,
                       tile_bits, out);
    return hipGetLastError();
}

}  // namespace mi355
