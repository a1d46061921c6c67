// Screened transform pipeline (gfx950 / CDNA4): the reference's in-place fp64 chain
// evaluated as an exact fixed-point linear map on the int8 MFMA units, with a
// rigorous accept/recompute test so that the results stay bit-identical.
//
// Why this is legal: in exact arithmetic the chain of
// utils.cpp:314-348 is a fixed 64x64 linear map L of the level-shifted samples
// p in [-128,127]^64.  The fp64 chain c the reference runs differs from L p only by
// rounding, |c_k - (L p)_k| <= eps_k (forward error analysis,
// tools/gen_screen_tables.py).  The kernel computes Lt p exactly, Lt = round(L*2^39)
// split into five balanced int8 digits (|Lt - L| p <= 2^-27), so
// |c_k/Q_k - z_k| <= tau_k with z_k = (Lt p)_k / Q_k.  The reference's quantiser
// round(fl(c/Q)) equals round-half-away of the exact quotient, so
// whenever z_k is farther than tau_k from every half-integer the quantised value
// is decided.  Units with an undecided coefficient are recomputed (inside the same kernel,
// exact_unit_wave) with the exact ordered fp64 chain -- that chain remains the arbiter.  Coefficient 0 is
// always exact: row 0 of L is SCALE_00 * ones, so c_0 = fl(sum(p) * SCALE_00) is
// formed directly.
//
// Pipeline per batch:
//   k_screen_encode  RGB -> per-unit {DC, AC bit string (word-aligned blob in an arena)}
//                    samples (integer-exact CSC), MFMA map, quantise+verify, LDS transpose to
//                    zig-zag rows, per-unit RLE/Huffman walk into an LDS slot, blob store
//                    (+ the exact fp64 chain for the rare undecided units)
//   k_dc_heads       DC symbol of each tile's first unit per pass -> tile sums
//   k_tile_scan      (jpeg_kernels.hip) 64-bit scan of tile sums
//   k_merge          DC symbols + AC blobs -> final bit string (LDS window per tile)
#include "jpeg_screen_devfn.h"

namespace mi355 {

constexpr uint32_t kEncWaves = 4;

// An error in a unit also poisons its tile's bit total (bit 31, never reached by the sums): k_tile_scan then knows WHICH
// frame failed.
#define POISON_TILE() atomicOr(&sp.tile_bits[(size_t)frame * g.tiles + tile], 0x80000000u)
#define POISON_FT() atomicOr(&sp.tile_bits[ft], 0x80000000u)

// Diagnostic build (make STAMPS=1): s_memtime stamps at the phase boundaries of a wave
// iteration, summed per wave and written to sp.stamps.  Never in the shipped kernel.
#ifdef MI355_STAMPS
#define STAMP(i)                                                                         \
    do {                                                                                 \
        unsigned long long _t;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        stamp_sum[i] += _t - stamp_prev;                                                 \
        stamp_prev = _t;                                                                 \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// MODE 0: strict (the reference's arithmetic); 1: standard 4:4:4; 2: standard 4:2:0 (tile = 64 MCUs,
// six passes: luma quarter-tiles 0..3 -- unit u of pass s is block k = u & 3 of MCU 16 s + u / 4, i.e.
// the scan order of the luma blocks -- then Cb, Cr with one block per MCU).
// At most 224 registers (the attribute counts in units of two on gfx90a and later: 112): two workgroups of this kernel
// per CU leave 64 per SIMD lane, which is what the tail kernels of the part in front need to run BESIDE it (k_merge: two
// waves, 32 allocated each; one register granule more and batched calls lose a fifth to a quarter: DESIGN.md §4.5,
// tests/test_kernel_budget.py).
template <bool PROBE, int MODE>
__attribute__((amdgpu_num_vgpr(112))) __global__ void __launch_bounds__(256, 2)
    k_screen_encode(Geom g, uint32_t n_frames, const uint8_t* __restrict__ rgb, ScreenParams sp) {
    constexpr bool STD = MODE != 0, S420 = MODE == 2;
    constexpr uint32_t kPasses = S420 ? 6u : 3u;
    // Standard 4:4:4 converts whole tiles on the matrix units (jpeg_screen_devfn.h: +4 %).  4:2:0 does not: the same
    // scheme for its chroma passes was built and measured at -1.5 % (profiles/r03_f_*) -- a matrix
    // instruction costs the issuing wave what 2.5 plain VALU instructions cost, the fixed-point form needs only four of
    // those per pixel, and the fragments' registers push the kernel to the limit beyond which the tail kernels stop
    // running beside it.
    constexpr bool kCscMfma = MODE == 1;
    // Strict mode forms the reference's integer numerators on the matrix units too (strict_rowpair_mfma); division, luma's
    // remainder test and the chroma means stay on the vector units.  Bit-exact either way; which is faster depends on the
    // instruction scheduling: under the backend's default strategy the matrix form lost 9 % (gpurun r4cu: its results arrive
    // late in a phase with nothing else to issue), under iterative-ilp it wins 1.2 % (269.8 against 266.5, gpurun r4cs2).
    constexpr bool kCscMfmaStrict = MODE == 0;
    __shared__ uint32_t s_tbuf_all[kEncWaves][kRowWords];          // zig-zag rows, int16 [position][unit] (jpeg_screen_devfn.h)
    __shared__ alignas(16) uint32_t s_slot_all[kEncWaves][(kSlotRows + 1) * 64];  // AC strings [word][lane] + dump row
    __shared__ uint32_t s_mask_all[kEncWaves][2][64];              // non-zero masks (lo, hi)
    __shared__ float s_qf[2][16][8];    // per group of 4 positions: 2^-23/Q x4 (first look: top three digits), its thresholds x4
    __shared__ uint32_t s_act[2][256];  // (run,size) AC tables
    __shared__ uint32_t s_lut2[2][kLut2Words];  // (value,run) symbol tables
    __shared__ uint32_t s_dc[2][16];      // DC tables

    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
    uint32_t* s_tbuf = s_tbuf_all[wv];
    uint32_t* s_slot = s_slot_all[wv];
    uint32_t* s_mlo = s_mask_all[wv][0];
    uint32_t* s_mhi = s_mask_all[wv][1];
    for (uint32_t i = tid; i < 512; i += 256) {
        (&s_act[0][0])[i] = sp.lut[512 + i];
    }
    for (uint32_t i = tid; i < 2 * kLut2Words; i += 256) (&s_lut2[0][0])[i] = sp.lut2[i];
    (&s_qf[0][0][0])[tid] = sp.qconst_f[tid];
    if (tid < 32) s_dc[tid >> 4][tid & 15] = sp.lut[(tid >> 4) * 256 + (tid & 15)];
    if (lane < 32) s_tbuf[64 * 32 + lane] = kRowSentinel * 0x00010001u;  // sentinel row after zig-zag position 63 (never written again)
    i16a* const tb16 = reinterpret_cast<i16a*>(s_tbuf);
    // A fragments of the top three digits stay in registers; the two low digits only matter for the
    // (rare) second look and are fetched on demand.
    v4i A[4][kLookDigits];
    load_look_fragments(sp, lane, A);
    // quantiser divisors of coefficient 0, read once: a load per pass would sit behind everything the wave has in flight
    // (vmcnt retires in issue order), the next pass's first rows included
    const double q0_luma = sp.qd[0], q0_chroma = sp.qd[64];
    __syncthreads();

    // Work distribution.  With a grid that is a multiple of 8 workgroups, the waves of XCD x
    // (workgroups x, x+8, ...) take the tiles congruent to x mod 8, channel by channel, so
    // that the three channels of a tile are processed side by side in one XCD and share its
    // RGB bytes in that L2.  Speed only: any mapping is correct.
    const uint32_t gwave = blockIdx.x * kEncWaves + wv;   // global wave id
    const uint32_t per_frame = g.tiles * kPasses;
    const bool xcd_map = (gridDim.x % 8u) == 0u;
    const uint32_t xcd = blockIdx.x % 8u;
    const uint32_t local = (blockIdx.x / 8u) * kEncWaves + wv;       // index of this wave inside its XCD
    const uint32_t local_n = (gridDim.x / 8u) * kEncWaves;           // waves per XCD
    const uint32_t tiles_x = xcd_map ? (g.tiles + 7u - xcd) / 8u : 0u;  // tiles this XCD owns per frame
    const uint32_t pairs_total = xcd_map ? tiles_x * kPasses * n_frames : per_frame * n_frames;
    const uint32_t pstart = xcd_map ? local : gwave;
    const uint32_t pstep = xcd_map ? local_n : gridDim.x * kEncWaves;

    // Stagger (MI355X_MICROARCH.md, two waves per SIMD, item 9): the two workgroups of a CU otherwise run in lockstep --
    // both waves of a SIMD in the issue-heavy transform, then both in the latency-bound walk.  Starting the
    // later-dispatched workgroup about half a pass late puts one wave's walk beside the other's transform.
    if (blockIdx.x >= sp.prio_from_wg)
        for (uint32_t i = 0; i < sp.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    WaveArena wa{gwave * sp.region_words, sp.region_words};
#ifdef MI355_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev, wave_t0, wave_t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wave_t0)::"memory");  // 100 MHz wall clock
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    bool walk_general[2] = {false, false};  // per channel type: the last pass had a symbol-table miss (walk_nonzeros)
    uint32_t n_rewalked = 0, n_general = 0;  // mi355_jpeg_screen_stats: summed per wave, added once at the end (one
                                             // atomic per pass on one address would serialise the whole device at q90)
    // One pass = one (frame, tile, pass) of this wave's share.  Its coordinates are worked out one pass AHEAD, and the
    // first row pair of the next pass is requested before the entropy walk of the current one: the load (an HBM miss for
    // the first of a tile's three channel waves) lands during the walk instead of stalling the next pass at its first
    // instruction, and because vmcnt retires in issue order it does not wait behind this pass's scattered string stores.
    struct Pass {
        uint32_t frame, tile, chan;
        // block (or MCU) coordinates of this lane's four units: x | y << 16 (packed: registers).  Four scalars, not an
        // array: an array member kept the struct in memory (the optimiser's 20-byte alloca was then promoted to LDS --
        // 5 KB per workgroup, enough to push the tail kernels off the CU: -25 % on batches)
        uint32_t b0, b1, b2, b3;
        bool fast;
        __device__ __forceinline__ uint32_t bxy(int j) const { return j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3)); }
        __device__ __forceinline__ void set_bxy(int j, uint32_t v) {
            if (j == 0) b0 = v;
            else if (j == 1) b1 = v;
            else if (j == 2) b2 = v;
            else b3 = v;
        }
    };
    auto locate = [&](uint32_t p) -> Pass {
        Pass ps;
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
        const bool luma420 = S420 && chan < 4u;

        // block coordinates of this lane's four blocks (16j + n), and whether the whole tile
        // lies inside the image (no mirror padding)
        bool interior = true;
        if constexpr (S420) {
            // MCU of this lane's unit in sub-tile j: luma pass s: 16 s + 4 j + n / 4 (block k = n & 3 of it),
            // chroma pass: 16 j + n.  For chroma passes bxy holds MCU coordinates.
            const uint32_t step = luma420 ? 4u : 16u;
            uint32_t m = tile * 64 + (luma420 ? 16 * chan + (n >> 2) : n);
            uint32_t my = m / g.nmx, mx = m - my * g.nmx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (m >= g.N) {  // past the last MCU: any valid one will do, the lane is masked later
                    mx = g.nmx - 1;
                    my = g.N / g.nmx - 1;
                }
                if (luma420) {
                    const uint32_t lx = 2 * mx + (n & 1), ly = 2 * my + ((n >> 1) & 1);
                    ps.set_bxy(j, lx | (ly << 16));
                    interior = interior && (lx * 8 + 8 <= g.W) && (ly * 8 + 8 <= g.H);
                } else {
                    ps.set_bxy(j, mx | (my << 16));
                    interior = interior && (mx * 16 + 16 <= g.W) && (my * 16 + 16 <= g.H);
                }
                m += step;
                mx += step;
                while (mx >= g.nmx) {
                    mx -= g.nmx;
                    ++my;
                }
            }
        } else {
            uint32_t b = tile * 64 + n;
            uint32_t by = b / g.nbx, bx = b - by * g.nbx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t bb = tile * 64 + 16 * j + n;
                if (bb >= g.N) {  // past the last block: any valid block will do, the lane is masked later
                    bx = g.nbx - 1;
                    by = g.N / g.nbx - 1;
                }
                ps.set_bxy(j, bx | (by << 16));
                interior = interior && (bx * 8 + 8 <= g.W) && (by * 8 + 8 <= g.H);
                bx += 16;
                while (bx >= g.nbx) {
                    bx -= g.nbx;
                    ++by;
                }
            }
        }
        ps.fast = g.fast_rows && !wave_any(!interior);
        ps.frame = frame, ps.tile = tile, ps.chan = chan;
        return ps;
    };
    uint32_t raw[12];  // raw RGB of the row pair to convert next (fast path)
    RawChunk Xn[4];    // the same for the matrix-unit conversion (standard 4:4:4): the next row pair's four chunks
    v4i F[4];          // ... and the colour-conversion fragments of the pass they belong to
    auto request_first_rows = [&](const Pass& ps) {
        const bool chroma420 = S420 && ps.chan >= 4u;
        if (!ps.fast) return;
        const uint8_t* pf = rgb + (size_t)ps.frame * g.frame_stride;
        if constexpr (kCscMfma) {
            // the pass's colour-conversion fragments travel with its first rows: requested before the walk of the pass in
            // front, they do not queue behind that pass's string stores (vmcnt retires in issue order)
            load_csc_fragments(sp, lane, (int)ps.chan * 4, F);
            load_std_rowpair(pf, g, ps.b0 & 0xffffu, ps.b0 >> 16, gq, Xn);
        } else {
            if constexpr (kCscMfmaStrict) {  // the pass's two digit sets (F[0], F[1]) travel with its first rows
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 t = sp.csc_frag[(kCscSets + (int)ps.chan * 2 + i) * 64 + lane];
                    F[i] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
                }
            }
            if (!chroma420) load_raw_rowpair(pf, g, ps.b0 & 0xffffu, ps.b0 >> 16, gq, raw);
        }
    };
    Pass cur{}, nxt{};
    if (pstart < pairs_total) {
        cur = locate(pstart);
        request_first_rows(cur);
    }
    for (uint32_t p = pstart; p < pairs_total; p += pstep) {
        STAMP(7);
        const uint32_t frame = cur.frame, tile = cur.tile, chan = cur.chan;
        // `chan` is the pass; the colour component differs from it only in 4:2:0 (passes 0..3 = luma)
        const uint32_t comp = S420 ? (chan < 4u ? 0u : chan - 3u) : chan;
        const bool luma420 = S420 && chan < 4u, chroma420 = S420 && chan >= 4u;
        const uint32_t ct = comp ? 1u : 0u;
        const uint8_t* f = rgb + (size_t)frame * g.frame_stride;
        const bool avg = !STD && (comp != 0) && (g.flags & 1u);  // standard mode never replicates chroma means
        const size_t us_base = (((size_t)frame * g.tiles + tile) * kPasses + chan) * 64;
        const bool fast = cur.fast;

        STAMP(0);
        // Issue arbitration is oldest-first, and the two workgroups of a CU are dispatched in grid
        // order: without help the waves of the later-dispatched half of the grid get the leftover
        // issue slots and finish ~20 % later than the others (measured with in-kernel timestamps:
        // mean wave end 41.5 vs 49.3 us), leaving the SIMDs half empty at the end.  Raising their
        // priority outside the (LDS-latency-bound) entropy walk equalises the two halves
        // (45.9 vs 45.4 us) and shortens the kernel by 8 %.  Only when this launch fills the device with
        // exactly two workgroups per CU (sp.prio_from_wg = number of CUs, else none): half-device launches
        // of pipelined callers share each CU with another stream's kernel and do better without it (+3 %).
        // Speed only.
        if (blockIdx.x >= sp.prio_from_wg) __builtin_amdgcn_s_setprio(1);

        s_mlo[lane] = 0;
        s_mhi[lane] = 0;
        __builtin_amdgcn_wave_barrier();

        const bool on_mfma = kCscMfma && fast;
        // raw RGB of unit-tile j+1 is fetched while unit-tile j is processed (that of unit-tile 0 was requested a pass ago)
        uint32_t dcsum = 0;  // sample sum of the block whose coefficient 0 this lane will form
        // scale factors and accept thresholds of the NEXT quantiser group, requested one group ahead (see the loop below)
        v4f qf_s = *reinterpret_cast<const v4f*>(&s_qf[ct][gq][0]), qf_h = *reinterpret_cast<const v4f*>(&s_qf[ct][gq][4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bx = cur.bxy(j) & 0xffffu, by = cur.bxy(j) >> 16;
            uint32_t pk[4];  // 16 samples; on_mfma: as sample - 128 (int8), else unsigned
            if (on_mfma) {
                if constexpr (kCscMfma) {
                    // (one buffer: the next row pair lands during the quantiser)
                    if (comp) std_rowpair_mfma<true>(Xn, F, pk);
                    else std_rowpair_mfma<false>(Xn, F, pk);
                    if (j < 3) load_std_rowpair(f, g, cur.bxy(j + 1) & 0xffffu, cur.bxy(j + 1) >> 16, gq, Xn);
                }
            } else if (chroma420) {
                if constexpr (S420) {  // rows 2gq, 2gq+1 of the MCU's 8x8 chroma block <- pixel rows 4gq .. 4gq+3
                    if (fast) {
#pragma unroll 1
                        for (uint32_t half = 0; half < 2; ++half) {  // one chroma row at a time: 24 live dwords
                            uint32_t w24[24], o[2];
                            load_raw_mcu_rows(f, g, bx, by, 4 * gq + 2 * half, w24);
                            if (comp == 1) convert_chroma420_row<1>(w24, o);
                            else convert_chroma420_row<2>(w24, o);
                            if (half == 0) pk[0] = o[0], pk[1] = o[1];
                            else pk[2] = o[0], pk[3] = o[1];
                        }
                    } else {
                        generic_chroma420(f, g, (int)comp, bx, by, gq, pk);
                    }
                }
            } else if (fast) {
                uint32_t rp[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) rp[i] = raw[i];
                if (j < 3) load_raw_rowpair(f, g, cur.bxy(j + 1) & 0xffffu, cur.bxy(j + 1) >> 16, gq, raw);  // (two pairs in flight: -1 %)
                if constexpr (kCscMfmaStrict) {
                    const v4i (&F2)[2] = reinterpret_cast<const v4i (&)[2]>(F);
                    if (comp == 0) strict_rowpair_mfma<0>(rp, F2, splat4(kCscStrictC[0]), false, pk);
                    else strict_rowpair_mfma<1>(rp, F2, splat4(kCscStrictC[1]), avg, pk);  // (the fragments say which chroma channel)
                } else {
                    if (comp == 0) convert_rowpair<0, STD>(rp, false, pk);
                    else if (comp == 1) convert_rowpair<1, STD>(rp, avg, pk);
                    else convert_rowpair<2, STD>(rp, avg, pk);
                }
            } else {
                if (comp == 0) generic_rowpair<0, STD>(f, g, false, bx, by, gq, pk);
                else if (comp == 1) generic_rowpair<1, STD>(f, g, avg, bx, by, gq, pk);
                else generic_rowpair<2, STD>(f, g, avg, bx, by, gq, pk);
            }
            if constexpr (PROBE && !S420) {
                if (sp.samples && tile * 64 + 16 * j + n < g.N) {
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) {
                        uint32_t v = ((pk[sidx >> 2] ^ (on_mfma ? 0x80808080u : 0u)) >> (8 * (sidx & 3))) & 255u;
                        size_t px = (size_t)(by * 8 + gq * 2 + (sidx >> 3)) * g.W8 + bx * 8 + (sidx & 7);
                        sp.samples[((size_t)frame * g.W8 * g.H8 + px) * 3 + chan] = (uint8_t)v;
                    }
                }
            }
            STAMP(5);
            // sum of the block's 64 samples (for the exact DC): 16 in this lane, then over the 4 row-pair lanes
            uint32_t ssum = 0;
            v4i B;
            if (on_mfma) {  // signed bytes already: the sum of the unsigned samples is 16 * 128 more
                int sg = 2048;
#pragma unroll
                for (int i = 0; i < 4; ++i) sg = __builtin_amdgcn_sdot4((int)pk[i], 0x01010101, sg, false);
                ssum = (uint32_t)sg;
                B = v4i{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ssum = __builtin_amdgcn_sad_u8(pk[i], 0u, ssum);
                // level shift: sample - 128 as int8 == sample ^ 0x80
                B = v4i{(int)(pk[0] ^ 0x80808080u), (int)(pk[1] ^ 0x80808080u), (int)(pk[2] ^ 0x80808080u),
                        (int)(pk[3] ^ 0x80808080u)};
            }
            // the sum over the four row-pair lanes of a unit (lanes n, n + 16, n + 32, n + 48) without a trip through LDS:
            // v_permlane16_swap / v_permlane32_swap exchange rows of 16 / halves of 32 between two copies of the value
            {
                const auto r16 = __builtin_amdgcn_permlane16_swap(ssum, ssum, false, false);
                ssum = r16[0] + r16[1];
                const auto r32 = __builtin_amdgcn_permlane32_swap(ssum, ssum, false, false);
                ssum = r32[0] + r32[1];
            }

            // coefficient 0 is formed exactly after this loop, by the lane (n, gq == j) for unit 16j+n
            if (gq == (uint32_t)j) dcsum = ssum;

            bool amb = false;
            uint32_t nzlo = 0, nzhi = 0;  // this lane's part of the unit's non-zero mask
            uint32_t qprev[4];            // values of the even row tile, paired with the odd one for the mask
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                uint32_t qb[4];  // low 16 bits = quantised value
                // the scale factors and thresholds of this group were requested one group ago (they depend on the row tile only:
                // the four sets go round); read where they are used, the two LDS reads sit two instructions in front of their
                // first use and the wave waits out the LDS latency sixteen times per pass
                float qfr[8];
                {
                    const v4f qs = qf_s, qh = qf_h;
                    const float* nq = &s_qf[ct][4 * ((mt + 1) & 3) + gq][0];
                    qf_s = *reinterpret_cast<const v4f*>(nq);
                    qf_h = *reinterpret_cast<const v4f*>(nq + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) qfr[i] = qs[i], qfr[4 + i] = qh[i];
                }
                screen_quantise<STD>(A[mt], B, sp, qfr, ct, mt, gq, lane, qb, amb);
                // zig-zag positions 16mt+4gq .. +3 of unit 16j+n -> transpose buffer + non-zero bits
                i16a* row = tb16 + (16 * mt + 4 * gq) * 64 + row_unit_off(16 * j + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) row[r * 64] = (int16_t)qb[r];
                // Non-zero bits, two values per instruction: the 16-bit values of row tiles mt - 1 and mt side by side
                // (one v_perm_b32), min(value, 1) on both halves (one v_pk_min_u16) = the flags at bits 0 and 16 --
                // exactly where positions 16 (mt - 1) + r and 16 mt + r sit in the mask word -- shifted in by r.
                if (mt & 1) {
                    uint32_t w = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t pr = __builtin_amdgcn_perm(qb[r], qprev[r], 0x05040100u);  // qprev.lo16 | qb.lo16 << 16
                        uint32_t f;  // (the compiler turns min(x, 1) into two compares and two selects)
                        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(f) : "v"(pr));
                        w |= f << r;
                    }
                    if (mt == 1) nzlo = w;
                    else nzhi = w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) qprev[r] = qb[r];
                }
            }
            atomicOr(&s_mlo[16 * j + n], (nzlo << (4 * gq)) & ~1u);
            atomicOr(&s_mhi[16 * j + n], nzhi << (4 * gq));
            if (amb) atomicOr(&s_mlo[16 * j + n], 1u);  // bit 0 (coefficient 0 is never walked) = "undecided unit"
            STAMP(6);
        }
        if (p + pstep < pairs_total) {  // wave-uniform
            nxt = locate(p + pstep);
            request_first_rows(nxt);
        }
        {
            // exact coefficient 0 of unit 16*gq + n.  Strict: c0 = fl(sum * SCALE_00), q0 = round(c0 / Q0)
            // (utils.cpp:336,459).  Standard: row 0 of the true DCT is exactly 1/8,
            // q0 = round-half-away(sum / (8 Q0)) in integers.
            int q0;
            if constexpr (STD) {
                const int sl = (int)dcsum - 8192;
                const uint32_t Q0 = (uint32_t)(ct ? q0_chroma : q0_luma), a0 = (uint32_t)(sl < 0 ? -sl : sl);
                const int n0 = (int)((a0 + 4u * Q0) / (8u * Q0));
                q0 = sl < 0 ? -n0 : n0;
            } else {
                // (fp64, ~30 double-rate instructions per lane and pass -- and still faster than reading the 16321 possible
                // results from a table in memory: 251.8 against 256.9 Gpixel/s with the table, even with its request issued
                // in front of the next pass's rows, gpurun r4r / r4s)
                const double c0 = (double)((int)dcsum - 8192) * kScale00;
                q0 = (int)__builtin_round(c0 / (ct ? q0_chroma : q0_luma));
            }
            tb16[row_unit_off(16 * gq + n)] = (int16_t)q0;
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(1);

        __builtin_amdgcn_s_setprio(0);
        // ---- walk phase: lane = block
        const uint32_t b = luma420 ? tile * 64 + 16 * chan + (lane >> 2) : tile * 64 + lane;  // block, or MCU in 4:2:0
        const bool active = b < g.N;
        if constexpr (!STD) {
            // Units with a coefficient the screen could not decide: the exact chain is the arbiter.
            const bool undecided = active && (s_mlo[lane] & 1u) != 0;
            uint64_t todo = __ballot(undecided);
            while (todo) {  // wave-uniform: one unit at a time, the whole wave on it
                const uint32_t ul = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                if (lane == 0) atomicAdd(&sp.stats[1], 1ull);
                const uint32_t ub = tile * 64 + ul, uby = ub / g.nbx, ubx = ub - uby * g.nbx;
                exact_unit_wave(f, g, chan, ubx, uby, sp.qd, reinterpret_cast<double*>(s_slot), tb16 + row_unit_off(ul), &s_mlo[ul],
                                &s_mhi[ul], lane);
            }
        }
        i16a* const row16 = tb16 + row_unit_off(lane);
        uint64_t mask = ((uint64_t)s_mhi[lane] << 32 | s_mlo[lane]) & ~1ull;
        const int dc = (int)row16[0];

        if constexpr (PROBE) {
            uint32_t* dst = sp.coefs + us_base / 64 * 2048 + lane;
#pragma unroll
            for (int pp = 0; pp < 32; ++pp)
                dst[pp * 64] = active ? (((uint32_t)(uint16_t)row16[2 * pp * 64]) | ((uint32_t)(uint16_t)row16[(2 * pp + 1) * 64] << 16)) : 0u;
        }

        Packer32<StoreLds> pkr(StoreLds{&s_slot[lane]});
        mask = mark_zero_runs(mask);  // ZRL positions become virtual non-zeros (after the probe dump above)
        const uint32_t maxcnt = wave_max((uint32_t)__popcll(mask));
        bool ok = walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pkr, maxcnt, walk_general[ct]);
        n_general += walk_general[ct] ? 1u : 0u;
        const uint32_t aclen = pkr.bits();
        uint32_t nw = pkr.words();
        STAMP(2);
        const bool oversize = nw > kSlotRows;
        // (an error also poisons the tile's bit total -- bit 31, never reached by the sums -- which is how k_tile_scan
        // learns WHICH frame failed without this kernel carrying a per-frame flag array)
        if (!ok && active) atomicOr(sp.status, 1u), POISON_TILE();  // MI355_E_CATEGORY
        if (!active) nw = 0;

        // Total bits of the unit = DC symbol + AC string.  The DC difference needs the previous
        // block of the same channel: the neighbouring lane.  Lane 0's predecessor is the last block
        // of the previous tile, which another wave owns: its DC symbol is left out here and added
        // by k_dc_heads from the DCs in `meta`.  Tile sums are accumulated with one atomic per wave.
        uint32_t ubits = aclen;
        {
            const int pred = __builtin_amdgcn_update_dpp(0, dc, 0x138, 0xf, 0xf, false);  // wave_shr:1 -- the previous lane's DC, no trip through LDS
            auto count = [&](uint32_t, uint32_t len) { ubits += len; };
            const bool dc_ok = lane == 0 || put_dc(dc - pred, s_dc[ct], count);
            if (!dc_ok && active) atomicOr(sp.status, 1u), POISON_TILE();  // MI355_E_CATEGORY
            if (!active) ubits = 0;
        }
        STAMP(3);
        // arena space: regular strings back to back; oversized ones get a full-size private run.  ONE wave scan carries
        // both sums: the units' bits (< 2^11 each) in the low 20 bits, the words needed (<= 54 each) above them.
        const uint32_t need = oversize && nw ? kSlotWordsFull : nw;
        const uint32_t both = wave_incl_scan((need << 20) | ubits, lane);
        const uint32_t both_all = (uint32_t)__builtin_amdgcn_readlane((int)both, 63);
        if (lane == 0 && (both_all & 0xFFFFFu)) atomicAdd(&sp.tile_bits[(size_t)frame * g.tiles + tile], both_all & 0xFFFFFu);
        const uint32_t incl = both >> 20;
        const uint32_t base = wa.take(sp, both_all >> 20, lane);
        const uint32_t off = base + incl - need;
        const bool fits = base != 0xFFFFFFFFu;
        if (!fits) {
            // cannot happen: the overflow pool holds the worst case of every unit of the part (run_screened).  MI355_E_INTERNAL.
            if (lane == 0) atomicOr(sp.status, 4u), POISON_TILE();
        } else {
            const uint32_t ncopy = oversize ? 0u : nw;
            // the first eight words of every string are read from the slot unconditionally, back to back (the slot has 25
            // rows: always in bounds), and only the stores are predicated: read under its predicate, each word costs a full
            // LDS round trip in front of its store
            uint32_t sw[8];
#pragma unroll
            for (uint32_t w = 0; w < 8; ++w) sw[w] = s_slot[w * 64 + lane];
#pragma unroll
            for (uint32_t w = 0; w < 8; ++w)
                if (w < ncopy) sp.arena[off + w] = sw[w];
            for (uint32_t w = 8; wave_any(w < ncopy); w += 4) {  // (24 slot rows + the dump row: rows w .. w + 3 exist for w <= 20)
                uint32_t s4[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) s4[i] = s_slot[(w + i) * 64 + lane];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i)
                    if (w + i < ncopy) sp.arena[off + w + i] = s4[i];
            }
            const uint64_t again = __ballot(oversize && nw);
            if (again) {  // string longer than the LDS slot (q50: never; noise at q90: most luma units): walk again, straight to memory
                n_rewalked += (uint32_t)__popcll(again);
                if (oversize && nw) {
                    Packer32<StoreGlobal> pg(StoreGlobal{sp.arena + off});
                    bool gen = true;
                    (void)walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pg, maxcnt, gen);
                }
            }
        }
        // 4 bytes per unit: the arena offset is not stored, k_merge forms it from the pass's base and a scan of the lengths
        sp.meta[us_base + lane] = active ? ((aclen << 16) | ((uint32_t)dc & 0xffffu)) : 0u;
        if (lane == 0) sp.pass_off[us_base >> 6] = base;
        __builtin_amdgcn_wave_barrier();
        STAMP(4);
        cur = nxt;
    }
    if (lane == 0 && n_rewalked) atomicAdd(&sp.stats[2], (unsigned long long)n_rewalked);
    if (lane == 0 && n_general) atomicAdd(&sp.stats[3], (unsigned long long)n_general);
#ifdef MI355_STAMPS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wave_t1)::"memory");
    if (sp.stamps && lane == 0) {
        for (int i = 0; i < 8; ++i) sp.stamps[(size_t)gwave * 8 + i] = stamp_sum[i];
        sp.stamps[(size_t)(2048 + gwave) * 8] = wave_t0;  // start / end of the wave, 10 ns ticks
        sp.stamps[(size_t)(2048 + gwave) * 8 + 1] = wave_t1;
    }
#endif
}

// ----------------------------------------------------------------------------
// k_dc_heads: DC symbol of the first unit of every (tile, pass).  Its predecessor is the last
// block of the previous tile (or luma quarter-tile), encoded by another wave of k_screen_encode,
// which therefore left the symbol out of the tile sum.  Every DC in `meta` is exact already
// (the exact recomputation keeps coefficient 0).  Light on purpose (few registers, 128 B of LDS): it runs every
// frame next to another stream's k_screen_encode.
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
    k_dc_heads(Geom g, uint32_t n_frames, ScreenParams sp) {
    __shared__ uint32_t s_dcf[2][16];
    // short kernel on its stream's critical path, usually resident next to another stream's
    // k_screen_encode: do not let it starve behind those (older) waves
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x;
    if (lane < 32) s_dcf[lane >> 4][lane & 15] = sp.lut[(lane >> 4) * 256 + (lane & 15)];
    __syncthreads();
    const uint32_t P = g.passes, heads = n_frames * g.tiles * P;
    const bool restart = (g.flags & 8u) != 0;  // MI355_F_RESTART: DC predictors start from 0 in every tile
    for (uint32_t p = blockIdx.x * 64 + lane; p < heads; p += gridDim.x * 64) {
        const uint32_t ft = p / P, c = p - ft * P, tile = restart ? 0u : ft % g.tiles;
        const size_t u0 = (size_t)p * 64;  // ((frame * tiles + tile) * passes + c) * 64
        int pred = 0;
        bool luma = c == 0;
        if (P == 6) {
            // 4:2:0: luma quarter-tile c follows quarter-tile c - 1 (or the previous tile's quarter-tile 3);
            // a quarter-tile past the last MCU has no units at all
            luma = c < 4;
            if (luma && (ft % g.tiles) * 64 + 16 * c >= g.N) continue;
            if (luma && c > 0) pred = meta_dc(sp.meta[u0 - 64 + 63]);
            else if (tile > 0) pred = meta_dc(sp.meta[u0 - 6 * 64 + (luma ? 3 * 64 : 0) + 63]);
        } else if (tile > 0) {
            pred = meta_dc(sp.meta[u0 - 192 + 63]);
        }
        const int dc = meta_dc(sp.meta[u0]);
        uint32_t len = 0;
        auto count = [&](uint32_t, uint32_t l) { len += l; };
        if (!put_dc(dc - pred, s_dcf[luma ? 0 : 1], count)) atomicOr(sp.status, 1u), POISON_FT();  // MI355_E_CATEGORY
        atomicAdd(&sp.tile_bits[ft], len);
    }
}

// ----------------------------------------------------------------------------
// DC predictor of a unit from `meta`: the previous lane, or the last block of the previous tile.
// ----------------------------------------------------------------------------
__device__ __forceinline__ int meta_pred(const uint32_t* __restrict__ meta, uint32_t frame_tile0, uint32_t tile,
                                         uint32_t chan, uint32_t lane, int own_dc) {
    int prev = __shfl_up(own_dc, 1);
    if (lane == 0) {
        prev = 0;
        if (tile > 0) prev = meta_dc(meta[((frame_tile0 + tile - 1) * 3 + chan) * 64 + 63]);
    }
    return prev;
}

// ----------------------------------------------------------------------------
// k_merge: like k_emit, but the AC bits come ready-made from the arena.
// ----------------------------------------------------------------------------
// S420: the tile is 64 MCUs = 384 units; thread t = 6 * mcu + k is the unit at position t of the
// tile's scan (k < 4: luma block k of the MCU = unit 4 mcu + k of the tile's 256 luma units, which the
// encode kernel stored as pass (4 mcu + k) / 64, lane (4 mcu + k) % 64; k = 4, 5: Cb, Cr).
// Launch bounds of 256 for the 192-thread form on purpose: for a three-wave workgroup with this much LDS the compiler
// works out that at most seven waves fit a SIMD and then RAISES the kernel's register allocation to the most seven waves
// allow -- 72 instead of the 32 it uses (.amdhsa_next_free_vgpr 65) -- and with 72 a k_merge wave only fits beside two
// k_screen_encode waves of at most 216 registers (found in round 4 when the encode kernel went to 221 and batched calls
// lost a quarter; tests/test_kernel_budget.py reads the allocation from the kernel descriptors now).
// SMALL: the bit-assembly window at half size (kEmitLdsWordsSmall), so that TWO workgroups of this kernel fit a CU beside
// two of k_screen_encode: batches run this form.  With one workgroup per CU the merge of a part takes about as long as
// the block encode it runs beside, and whenever it takes longer -- parts of unequal size, say 124 frames per call -- its
// workgroups are still streaming through the CUs when the NEXT launch's persistent workgroups arrive: a CU that holds
// two of them has no room for its second encode workgroup (72 KB of LDS), and that workgroup stays out until the
// following k_merge is through as well (rocprofv3 timeline, gpurun r4tl: launches of 540-830 us instead of 500; 232
// instead of 256 Gpixel/s at 100, 124, 132 frames per call).  At two per CU the merge is done well before the launch it
// runs beside (gpurun r4w2: 256 Gpixel/s at every batch size tried).  Tiles beyond 64 000 bits (15.6 bit per pixel)
// assemble their bits in device memory instead.
template <bool S420, bool SMALL>
__global__ void __launch_bounds__(S420 ? 384 : 256)
    k_merge(Geom g, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ pass_off, const uint32_t* __restrict__ arena,
            const uint32_t* __restrict__ lut, const uint64_t* __restrict__ tile_off,
            uint8_t* __restrict__ out, uint64_t out_stride, const uint64_t* __restrict__ frame_bits,
            uint32_t lds_words_limit) {
    constexpr uint32_t NT = S420 ? 384 : 192, UPB = S420 ? 6 : 3;  // threads, units per scan step (block / MCU)
    __builtin_amdgcn_s_setprio(3);  // see k_dc_heads
    // the 4:2:0 form gives up 256 window words for its longer offset array, so that both forms stay
    // within the 17.9 KiB a CU has left next to two resident workgroups of k_screen_encode
    constexpr uint32_t kWindow = (SMALL ? kEmitLdsWordsSmall : kEmitLdsWords) - (S420 ? 256 : 0);
    __shared__ uint32_t s_dc[2][16];
    __shared__ uint32_t s_bits[NT];
    __shared__ alignas(16) uint32_t s_words[kWindow];
    const uint32_t tid = threadIdx.x, lane = tid & 63, chan = tid >> 6;
    const uint32_t tile = blockIdx.x, frame = blockIdx.y;
    // a frame with an error (k_tile_scan wrote its verdict in place of the bit count: over capacity, a size without a
    // code) is skipped as a whole; the other frames of the call are written in full
    if (frame_bits[frame] >= kBitsFlagged) return;
    if (lds_words_limit > kWindow) lds_words_limit = kWindow;
    const uint32_t ft0 = frame * g.tiles;  // (slot indices in 32 bits: a part has fewer unit slots than arena words, and those are below 2^32)
    const uint64_t* to = tile_off + (size_t)frame * (g.tiles + 1);
    const uint64_t start = to[tile], end = to[tile + 1];
    const uint64_t w0 = start >> 5;
    const uint32_t nw = (uint32_t)(((end + 31) >> 5) - w0);
    const bool use_lds = nw <= lds_words_limit;
    uint32_t* outw = reinterpret_cast<uint32_t*>(out + (size_t)frame * out_stride);
    const bool last_tile = tile + 1 == g.tiles;
    const bool restart = (g.flags & 8u) != 0;          // MI355_F_RESTART
    const uint32_t ptile = restart ? 0u : tile;        // "no previous tile" for the DC predictors
    const uint32_t last_blk = g.N - 1 - tile * 64 < 63 ? g.N - 1 - tile * 64 : 63;  // last active block / MCU
    if (tid < 32) s_dc[tid >> 4][tid & 15] = lut[(tid >> 4) * 256 + (tid & 15)];
    if (use_lds) {  // (16 bytes per store; up to three words beyond nw: the window's size is a multiple of four)
        static_assert(kWindow % 4 == 0, "the window is zeroed and written out four words at a time");
        for (uint32_t i = tid * 4; i < nw; i += NT * 4) *reinterpret_cast<uint4*>(&s_words[i]) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (uint32_t i = tid; i < nw; i += NT) {
            bool shared = (i == 0 && (start & 31)) || (i == nw - 1 && (end & 31) && !last_tile);
            if (!shared) __hip_atomic_store(&outw[w0 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // this unit: DC, AC length, arena offset; the first words of its AC string are fetched now
    bool active, chroma, tile_end;  // tile_end: the last unit of the tile's scan
    uint32_t spos;  // position of the unit in the tile's scan
    uint32_t mw, moff;  // the unit's metadata word (aclen << 16 | dc) and the arena offset of its AC string
    int dc, pred;
    // Arena offsets are not stored per unit: a pass's strings lie back to back from the pass's base in lane order, so a
    // unit's offset is the base plus the words of the lanes in front of it -- one wave scan over the pass's 64 metadata words.
    auto words_of = [](uint32_t w) {
        const uint32_t nwz = ((w >> 16) + 31u) >> 5;
        return nwz > kSlotRows ? kSlotWordsFull : nwz;  // oversized strings own a full-size run (k_screen_encode)
    };
    if constexpr (S420) {
        const uint32_t mcu = tid / 6, k = tid - mcu * 6;
        const uint32_t t0 = (ft0 + tile) * 6 * 64;
        {   // wave w = pass w of the tile: offsets in pass order through s_bits, picked up in scan order below
            const uint32_t need = words_of(meta[t0 + tid]);
            s_bits[tid] = pass_off[(ft0 + tile) * 6 + (tid >> 6)] + wave_incl_scan_dpp(need) - need;
            __syncthreads();
        }
        spos = tid;
        chroma = k >= 4;
        active = tile * 64 + mcu < g.N;
        tile_end = mcu == last_blk && k == 5;
        pred = 0;
        if (!chroma) {
            const uint32_t L = 4 * mcu + k;  // pass L >> 6, lane L & 63: slot t0 + L
            mw = meta[t0 + L];
            moff = s_bits[L];
            if (L > 0) pred = meta_dc(meta[t0 + L - 1]);
            else if (ptile > 0) pred = meta_dc(meta[t0 - 6 * 64 + 3 * 64 + 63]);
        } else {
            const uint32_t slot = t0 + k * 64 + mcu;
            mw = meta[slot];
            moff = s_bits[k * 64 + mcu];
            if (mcu > 0) pred = meta_dc(meta[slot - 1]);
            else if (ptile > 0) pred = meta_dc(meta[slot - 6 * 64 + 63]);
        }
        dc = meta_dc(mw);
    } else {
        spos = lane * 3 + chan;
        chroma = chan != 0;
        active = tile * 64 + lane < g.N;
        tile_end = lane == last_blk && chan == 2;
        mw = meta[((ft0 + tile) * 3 + chan) * 64 + lane];
        const uint32_t need = words_of(mw);
        moff = pass_off[(ft0 + tile) * 3 + chan] + wave_incl_scan_dpp(need) - need;
        dc = meta_dc(mw);
        pred = meta_pred(meta, ft0, ptile, chan, lane, dc);
        if (restart && lane == 0) pred = 0;
    }
    const uint32_t aclen = active ? (mw >> 16) : 0u;
    // the first four words of the string in ONE load (dword-aligned; the words behind a shorter string are read and not used:
    // the arena ends in more than a kilobyte of slack per wave region)
    struct __attribute__((packed, aligned(4))) Words4 {
        uint32_t w[4];
    };
    uint32_t pre[4] = {0u, 0u, 0u, 0u};
    if (aclen) {
        const Words4 p4 = *reinterpret_cast<const Words4*>(arena + moff);
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = p4.w[i];
    }
    __syncthreads();
    // tile-local exclusive offsets in scan order; the DC symbol is formed once and kept (code | value bits, right-aligned)
    uint32_t dsym = 0, dcl = 0;
    {
        auto keep = [&](uint32_t code, uint32_t len) { dsym = code, dcl = len; };
        put_dc(dc - pred, s_dc[chroma ? 1 : 0], keep);
    }
    s_bits[spos] = active ? dcl + aclen : 0u;
    __syncthreads();
    if (tid < 64) {
        uint32_t a[UPB], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < UPB; ++i) a[i] = s_bits[tid * UPB + i], sum += a[i];
        uint32_t run = wave_incl_scan_dpp(sum) - sum;
#pragma unroll
        for (uint32_t i = 0; i < UPB; ++i) {
            s_bits[tid * UPB + i] = run;
            run += a[i];
        }
    }
    __syncthreads();
    if (active) {
        // The unit's bits go into the (zeroed) window by OR, a word at a time: the DC symbol at its bit position, then the
        // AC string -- whole words as they lie in the arena (left-aligned, zero beyond the string's end), each funnelled
        // with its predecessor to the string's bit phase: one v_alignbit_b32 and one LDS OR per word, no 64-bit
        // accumulator, no length bookkeeping.  (This kernel's instructions are issued on the SIMDs the block encode of the
        // next part runs on: until round 4 it took 270 vector and 200 scalar instructions per wave of 64 units, an eighth of
        // the encode kernel's own.)
        const uint32_t pos = (uint32_t)(start & 31) + s_bits[spos];  // bits from the first word of the tile: below 2^21
        // restart intervals end on a byte boundary, filled with 1s (their start is aligned)
        const uint32_t fill = restart && tile_end ? (8u - ((pos + dcl + aclen) & 7u)) & 7u : 0u;
        auto body = [&](auto&& orw) {
            if (dcl) {
                const uint32_t d = dsym << (32u - dcl), sh = pos & 31u;  // left-aligned; dcl <= 27
                orw(pos >> 5, d >> sh);
                if (sh + dcl > 32u) orw((pos >> 5) + 1u, d << (32u - sh));
            }
            const uint32_t pa = pos + dcl, sa = pa & 31u, ja = pa >> 5;
            uint32_t prev = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                if (i * 32u < aclen) {
                    orw(ja + i, __builtin_amdgcn_alignbit(prev, pre[i], sa));  // (prev : word) >> sa
                    prev = pre[i];
                }
            }
            for (uint32_t k = 4; k * 32u < aclen; ++k) {
                const uint32_t w = arena[moff + k];
                orw(ja + k, __builtin_amdgcn_alignbit(prev, w, sa));
                prev = w;
            }
            // what the shift pushed out of the string's last word (its ((aclen - 1) & 31) + 1 valid bits reach beyond bit 31)
            if (aclen && ((aclen - 1u) & 31u) + sa >= 32u) orw(ja + ((aclen + 31u) >> 5), prev << (32u - sa));
            if (fill) {
                const uint32_t e = pa + aclen;
                orw(e >> 5, (((1u << fill) - 1u) << (32u - fill)) >> (e & 31u));
            }
        };
        if (use_lds) body([&](uint32_t j, uint32_t v) { atomicOr(&s_words[j], v); });
        else body([&](uint32_t j, uint32_t v) { atomicOr(&outw[w0 + j], __builtin_bswap32(v)); });
    }
    if (!use_lds) return;
    __syncthreads();
    // write-out, four words per thread and trip: one 16-byte LDS read, one 16-byte store where all four words are the
    // tile's own (every group but the first and the last); the words a tile shares with its neighbours go by atomic OR
    struct __attribute__((packed, aligned(4))) Out4 {
        uint32_t w[4];
    };
    const bool share_first = (start & 31) != 0, share_last = (end & 31) != 0 && !last_tile;
    for (uint32_t i = tid * 4; i < nw; i += NT * 4) {
        const uint4 r = *reinterpret_cast<const uint4*>(&s_words[i]);
        const uint32_t v[4] = {__builtin_bswap32(r.x), __builtin_bswap32(r.y), __builtin_bswap32(r.z), __builtin_bswap32(r.w)};
        if (i > 0 && i + 4 < nw) {
            *reinterpret_cast<Out4*>(&outw[w0 + i]) = Out4{{v[0], v[1], v[2], v[3]}};
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t w = i + k;
                if (w >= nw) break;
                const bool shared = (w == 0 && share_first) || (w == nw - 1 && share_last);
                if (shared) {
                    if (v[k]) atomicOr(&outw[w0 + w], v[k]);
                } else {
                    outw[w0 + w] = v[k];
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------
// Number of persistent WAVES (4 per workgroup).  Full groups of 8 workgroups whenever the
// work allows, so that the XCD-aware tile mapping applies.
uint32_t screen_grid(const Geom& g, uint32_t n_frames, uint32_t max_waves) {
    uint32_t total = g.tiles * g.passes * n_frames;
    uint32_t wgs = (total + kEncWaves - 1) / kEncWaves;
    uint32_t max_wgs = max_waves / kEncWaves ? max_waves / kEncWaves : 1;
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs >= 8) wgs &= ~7u;
    return wgs * kEncWaves;
}
hipError_t launch_screen_encode(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const ScreenParams& sp,
                                bool probe, uint32_t grid_waves, hipStream_t s) {
    uint32_t grid = screen_grid(g, n_frames, grid_waves) / kEncWaves;
    const int mode = is420(g) ? 2 : ((g.flags & 2u) ? 1 : 0);  // MI355_F_STANDARD, 4:2:0
#define MI355_LAUNCH_ENC(PR, MD) \
    hipLaunchKernelGGL((k_screen_encode<PR, MD>), dim3(grid), dim3(256), 0, s, g, n_frames, rgb, sp)
    if (probe) {
        if (mode == 2) MI355_LAUNCH_ENC(true, 2);
        else if (mode == 1) MI355_LAUNCH_ENC(true, 1);
        else MI355_LAUNCH_ENC(true, 0);
    } else {
        if (mode == 2) MI355_LAUNCH_ENC(false, 2);
        else if (mode == 1) MI355_LAUNCH_ENC(false, 1);
        else MI355_LAUNCH_ENC(false, 0);
    }
#undef MI355_LAUNCH_ENC
    return hipGetLastError();
}
hipError_t launch_dc_heads(const Geom& g, uint32_t n_frames, const ScreenParams& sp, hipStream_t s) {
    // one lane per (tile, pass) head when the batch is small, a few per lane when it is large
    const uint64_t heads = (uint64_t)n_frames * g.tiles * g.passes;
    const uint32_t head_waves = (uint32_t)((heads + 63) / 64 < 4096 ? (heads + 63) / 64 : 4096);
    hipLaunchKernelGGL(k_dc_heads, dim3(head_waves), dim3(64), 0, s, g, n_frames, sp);
    return hipGetLastError();
}
hipError_t launch_merge(const Geom& g, uint32_t n_frames, const uint32_t* meta, const uint32_t* pass_off, const uint32_t* arena,
                        const uint32_t* lut, const uint64_t* tile_off,
                        uint8_t* out, uint64_t out_stride, const uint64_t* frame_bits, uint32_t lds_words_limit,
                        bool small_window, hipStream_t s) {
    if (lds_words_limit > kEmitLdsWords) lds_words_limit = kEmitLdsWords;
#define MI355_LAUNCH_MERGE(S4, SM, NT) \
    hipLaunchKernelGGL((k_merge<S4, SM>), dim3(g.tiles, n_frames), dim3(NT), 0, s, g, meta, pass_off, arena, lut, tile_off, out, \
                       out_stride, frame_bits, lds_words_limit)
    if (is420(g)) {
        if (small_window) MI355_LAUNCH_MERGE(true, true, 384);
        else MI355_LAUNCH_MERGE(true, false, 384);
    } else {
        if (small_window) MI355_LAUNCH_MERGE(false, true, 192);
        else MI355_LAUNCH_MERGE(false, false, 192);
    }
#undef MI355_LAUNCH_MERGE
    return hipGetLastError();
}

}  // namespace mi355
