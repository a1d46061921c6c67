// k_screen_encode_wide: the block-encode kernel of the four-launch pipeline at THREE waves per SIMD.
//
// Same work per wave as k_screen_encode (jpeg_screen_kernels.hip): one (tile, channel) pass per iteration -- samples,
// fixed-point map on the MFMA units, quantise + verify, zig-zag rows, RLE/Huffman walk into an LDS slot, AC string ->
// arena, {DC, length} -> meta, unit totals -> tile_bits -- and the same tail kernels behind it.  What differs is the shape
// on the CU:
//
//   k_screen_encode        2 workgroups x 4 waves per CU: 199 VGPRs (48 of them the map's A fragments), 72.6 KB of LDS
//                          per workgroup (each with its own copy of the tables) -> two waves per SIMD
//   k_screen_encode_wide   1 workgroup x 11 waves per CU: the A fragments are loaded pair by pair (jpeg_transform_core.inc)
//                          -> <= 168 VGPRs, ONE copy of the tables, 11.9 KB of LDS per wave (12-word string slots instead of
//                          24) -> 142.6 KB: three waves on three SIMDs, two on the fourth -- which keeps 176 registers and,
//                          with the 17.3 KB of LDS left, the room the tail kernels (k_merge: 32 registers, 17.3 KB) need
//                          to run beside the next part's encode, as they do today
//
// The entropy walk and the quantiser spend most of their time waiting (dependent instruction chains, LDS round trips):
// measured in k_encode_tile, twelve waves per CU run the same pass 22-28 % faster per CU than eight.
// Strings longer than the slot (352 bits; q50 luma averages ~175) take the re-walk-to-memory path as in k_screen_encode.
// Strict mode and standard 4:4:4; stage probes and 4:2:0 stay with k_screen_encode.
#include "jpeg_screen_devfn.h"

namespace mi355 {

constexpr uint32_t kWideWaves = 11;
constexpr uint32_t kWideThreads = kWideWaves * 64;
constexpr uint32_t kWideSlotRows = 11;  // words per unit kept in LDS (+ one dump row)

// A fragments of row tiles 2 mtp, 2 mtp + 1 (top three digits each).  `lane16` = lane * 16 made opaque by the caller
// inside the pass loop: left to itself the compiler hoists the twelve 64-bit fragment addresses out of the loop and then
// spills them; with an opaque 32-bit offset the loads take the table's base from SGPRs and cost one add each.
__device__ __forceinline__ void load_pair_fragments(const ScreenParams& sp, uint32_t lane16, int mtp, v4i (&A)[2][kLookDigits]) {
    const char* const base = reinterpret_cast<const char*>(sp.afrag);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int l = 0; l < kLookDigits; ++l) {
            const uint32_t off = (uint32_t)(((2 * mtp + h) * kScreenLimbs + (kScreenLimbs - kLookDigits) + l) * 64 * 16) + lane16;
            const uint4 t = *reinterpret_cast<const uint4*>(base + off);
            A[h][l] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
        }
}

#define TSTAMP(i) do { } while (0)

struct WideWaveLds {
    uint32_t rows[kRowWords];                              // zig-zag rows, int16 [position][unit]
    alignas(16) uint32_t slot[(kWideSlotRows + 1) * 64];   // AC strings [word][lane] + dump row
    uint32_t mask[2][64];                                  // non-zero masks (lo, hi)
};

// MODE 0: strict (the reference's arithmetic); 1: standard 4:4:4.
template <int MODE>
__global__ void __launch_bounds__(kWideThreads)
    k_screen_encode_wide(Geom g, uint32_t n_frames, const uint8_t* __restrict__ rgb, ScreenParams sp) {
    constexpr bool STD = MODE != 0;
    constexpr uint32_t kPasses = 3u;
    __shared__ WideWaveLds s_wave[kWideWaves];
    __shared__ float s_qf[2][16][8];            // per group of 4 positions: 2^-23/Q x4, first-look thresholds x4
    __shared__ uint32_t s_act[2][256];          // (run,size) AC tables
    __shared__ uint32_t s_lut2[2][kLut2Words];  // (value,run) symbol tables
    __shared__ uint32_t s_dc[2][16];            // DC tables

    const uint32_t tid = threadIdx.x;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // lane / n / gq are made opaque at the phase boundaries of a pass (OPAQUE_LANE): everything derived from them -- a
    // few dozen LDS addresses and offsets -- is then recomputed per phase (a handful of instructions) instead of being
    // hoisted out of the pass loop into registers that live for the whole kernel and end up spilled
    uint32_t lane = tid & 63, n = lane & 15, gq = lane >> 4;
#define OPAQUE_LANE() do { asm volatile("" : "+v"(lane)); n = lane & 15u; gq = lane >> 4; } while (0)
    WideWaveLds& wl = s_wave[wv];
    uint32_t* const s_rows = wl.rows;
    uint32_t* const slot = wl.slot;
    uint32_t* const s_mlo = wl.mask[0];
    uint32_t* const s_mhi = wl.mask[1];
    i16a* const tb16 = reinterpret_cast<i16a*>(s_rows);
    for (uint32_t i = tid; i < 512; i += kWideThreads) (&s_act[0][0])[i] = sp.lut[512 + i];
    for (uint32_t i = tid; i < 2 * kLut2Words; i += kWideThreads) (&s_lut2[0][0])[i] = sp.lut2[i];
    if (tid < 256) (&s_qf[0][0][0])[tid] = sp.qconst_f[tid];
    if (tid < 32) s_dc[tid >> 4][tid & 15] = sp.lut[(tid >> 4) * 256 + (tid & 15)];
    if (lane < 32) s_rows[64 * 32 + lane] = kRowSentinel * 0x00010001u;  // sentinel row after zig-zag position 63 (never written again)
    const double q0_luma = sp.qd[0], q0_chroma = sp.qd[64];  // quantiser divisors of coefficient 0
    __syncthreads();

    // Work distribution as in k_screen_encode: with a grid that is a multiple of 8 workgroups, the waves of XCD x
    // (workgroups x, x+8, ...) take the tiles congruent to x mod 8, channel by channel, so that the three channels of a
    // tile are processed side by side in one XCD and share its RGB bytes in that L2.  Speed only: any mapping is correct.
    const uint32_t gwave = blockIdx.x * kWideWaves + wv;
    const uint32_t per_frame = g.tiles * kPasses;
    const bool xcd_map = (gridDim.x % 8u) == 0u;
    const uint32_t xcd = blockIdx.x % 8u;
    const uint32_t local = (blockIdx.x / 8u) * kWideWaves + wv;       // index of this wave inside its XCD
    const uint32_t local_n = (gridDim.x / 8u) * kWideWaves;           // waves per XCD
    const uint32_t tiles_x = xcd_map ? (g.tiles + 7u - xcd) / 8u : 0u;  // tiles this XCD owns per frame
    const uint32_t pairs_total = xcd_map ? tiles_x * kPasses * n_frames : per_frame * n_frames;
    const uint32_t pstart = xcd_map ? local : gwave;
    const uint32_t pstep = xcd_map ? local_n : gridDim.x * kWideWaves;

    WaveArena wa{gwave * sp.region_words, sp.region_words};
    bool walk_general[2] = {false, false};  // per channel type: the last pass had a symbol-table miss (walk_nonzeros)
    for (uint32_t p = pstart; p < pairs_total; p += pstep) {
        OPAQUE_LANE();
        uint32_t frame, tile, chan;
        if (xcd_map) {
            const uint32_t pf = tiles_x * kPasses;
            frame = p / pf;
            const uint32_t q = p - frame * pf;
            tile = (q / kPasses) * 8u + xcd;
            chan = q % kPasses;
        } else {
            frame = p / per_frame;
            const uint32_t q = p - frame * per_frame;
            tile = q / kPasses;
            chan = q % kPasses;
        }
        const uint32_t ct = chan ? 1u : 0u;
        const double q0d = ct ? q0_chroma : q0_luma;
        const bool avg = !STD && (chan != 0) && (g.flags & 1u);  // standard mode never replicates chroma means
        const uint8_t* const f = rgb + (size_t)frame * g.frame_stride;
        const uint32_t nblk = g.N - tile * 64 < 64u ? g.N - tile * 64 : 64u;  // active blocks of the tile
        const bool active = lane < nblk;
        v4i B[4];
        uint32_t dcsum = 0;  // sample sum of the block whose coefficient 0 this lane will form
#include "jpeg_transform_core.inc"
        // ---- walk phase: lane = block
        OPAQUE_LANE();
        if constexpr (!STD) {
            // Units with a coefficient the screen could not decide: the exact chain is the arbiter.
            const bool undecided = active && (s_mlo[lane] & 1u) != 0;
            uint64_t todo = __ballot(undecided);
            while (todo) {  // wave-uniform: one unit at a time, the whole wave on it
                const uint32_t ul = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                if (lane == 0) atomicAdd(&sp.stats[1], 1ull);
                const uint32_t ub = tile * 64 + ul, uby = ub / g.nbx, ubx = ub - uby * g.nbx;
                exact_unit_wave(f, g, chan, ubx, uby, sp.qd, reinterpret_cast<double*>(slot), tb16 + row_unit_off(ul), &s_mlo[ul],
                                &s_mhi[ul], lane);
            }
        }
        i16a* const row16 = tb16 + row_unit_off(lane);
        uint64_t mask = ((uint64_t)s_mhi[lane] << 32 | s_mlo[lane]) & ~1ull;
        const int dc = (int)row16[0];

        Packer32<StoreLds> pkr(StoreLds{slot + lane, kWideSlotRows, kWideSlotRows * 64u});
        mask = mark_zero_runs(mask);  // ZRL positions become virtual non-zeros
        const uint32_t maxcnt = wave_max((uint32_t)__popcll(mask));
        const bool ok = walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pkr, maxcnt, walk_general[ct]);
        const uint32_t aclen = pkr.bits();
        uint32_t nw = pkr.words();
        const bool oversize = nw > kWideSlotRows;
        OPAQUE_LANE();
        // (an error also poisons the tile's bit total -- bit 31, never reached by the sums -- which is how k_tile_scan
        // learns WHICH frame failed)
        if (!ok && active) atomicOr(sp.status, 1u), atomicOr(&sp.tile_bits[(size_t)frame * g.tiles + tile], 0x80000000u);  // MI355_E_CATEGORY
        if (!active) nw = 0;

        // Total bits of the unit = DC symbol + AC string.  The DC difference needs the previous block of the same channel:
        // the neighbouring lane.  Lane 0's predecessor is the last block of the previous tile, which another wave owns:
        // its DC symbol is left out here and added by k_dc_heads from the DCs in `meta`.
        {
            const int pred = __shfl_up(dc, 1);
            uint32_t ubits = aclen;
            auto count = [&](uint32_t, uint32_t len) { ubits += len; };
            const bool dc_ok = lane == 0 || put_dc(dc - pred, s_dc[ct], count);
            if (!dc_ok && active) atomicOr(sp.status, 1u), atomicOr(&sp.tile_bits[(size_t)frame * g.tiles + tile], 0x80000000u);  // MI355_E_CATEGORY
            if (!active) ubits = 0;
            ubits = wave_sum(ubits);
            if (lane == 0 && ubits) atomicAdd(&sp.tile_bits[(size_t)frame * g.tiles + tile], ubits);
        }

        // arena space: regular strings back to back; oversized ones get a full-size private run
        const uint32_t need = oversize && nw ? kSlotWordsFull : nw;
        const uint32_t incl = wave_incl_scan(need, lane);
        const uint32_t base = wa.take(sp, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), lane);
        const uint32_t off = base + incl - need;
        const bool fits = base != 0xFFFFFFFFu;
        if (!fits) {
            // MI355_E_CAPACITY (strings beyond 9/4 of the output capacity: the output could not hold them either)
            if (lane == 0) atomicOr(sp.status, 2u), atomicOr(&sp.tile_bits[(size_t)frame * g.tiles + tile], 0x80000000u);
        } else {
            const uint32_t ncopy = oversize ? 0u : nw;
#pragma unroll
            for (uint32_t w = 0; w < 8; ++w)
                if (w < ncopy) sp.arena[off + w] = slot[w * 64 + lane];
            for (uint32_t w = 8; __any(w < ncopy); ++w)
                if (w < ncopy) sp.arena[off + w] = slot[w * 64 + lane];
            if (__any(oversize && nw)) {  // a string longer than the LDS slot: walk again, straight to memory
                if (oversize && nw) {
                    Packer32<StoreGlobal> pg(StoreGlobal{sp.arena + off});
                    bool gen = true;
                    (void)walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pg, maxcnt, gen);
                }
            }
        }
        const size_t us_base = (((size_t)frame * g.tiles + tile) * kPasses + chan) * 64;
        sp.meta[us_base + lane] = make_uint2(off, active ? ((aclen << 16) | ((uint32_t)dc & 0xffffu)) : 0u);
        __builtin_amdgcn_wave_barrier();
    }
#undef OPAQUE_LANE
}
#undef TSTAMP

uint32_t wide_grid_waves(const Geom& g, uint32_t n_frames, uint32_t max_wgs) {
    const uint64_t total = (uint64_t)g.tiles * 3 * n_frames;
    uint64_t wgs = (total + kWideWaves - 1) / kWideWaves;
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    if (wgs >= 8) wgs &= ~7ull;
    return (uint32_t)wgs * kWideWaves;
}

hipError_t launch_screen_encode_wide(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const ScreenParams& sp, uint32_t grid_waves,
                                     hipStream_t s) {
    const uint32_t grid = grid_waves / kWideWaves;
    if (g.flags & 2u)  // MI355_F_STANDARD
        hipLaunchKernelGGL((k_screen_encode_wide<1>), dim3(grid), dim3(kWideThreads), 0, s, g, n_frames, rgb, sp);
    else
        hipLaunchKernelGGL((k_screen_encode_wide<0>), dim3(grid), dim3(kWideThreads), 0, s, g, n_frames, rgb, sp);
    return hipGetLastError();
}

}  // namespace mi355
