// Single-launch strict encode pipeline (gfx950 / CDNA4): RGB frames in, final scan bits out, ONE kernel.
//
// k_encode_fused does, per tile of 64 blocks (= 192 consecutive units of the reference's scan order
// 3*block + chan, utils.cpp:665-695), everything the four-launch pipeline of jpeg_screen_kernels.hip
// spreads over k_screen_encode / k_dc_heads / k_tile_scan / k_merge:
//
//   per channel pass (Y, Cb, Cr), one wave, lane roles as in k_screen_encode:
//       integer-exact colour conversion (+ chroma mean, mirror padding)           performCSC/CDS/pad
//       int8-MFMA fixed-point map, three-digit first look, rare second look        performDCT+Quantization
//       transpose to zig-zag rows in LDS, exact fp64 chain for undecided units     (the arbiter)
//       per-unit RLE/Huffman walk over the non-zeros into an LDS string slot       performRLE/HuffmanEncoder
//   then, still in the same wave:
//       DC symbols (predecessor = previous lane; lane 0: the previous tile's last DCs, handed over by
//       an 8-byte granule), unit and tile bit totals, decoupled look-back over the earlier tiles of the
//       frame for the tile's bit offset, merge of the 192 strings in scan order into an LDS window, coalesced
//       big-endian write-out; the word a tile shares with its successor travels as a "carry" granule, so
//       no output word is ever written by two waves and nothing has to be zeroed beforehand.
//
// No arena, no per-unit metadata, no scan or merge kernel: HBM traffic is the RGB read plus the stream
// written (plus 32 bytes of hand-off record per tile).
//
// Work distribution and forward progress: the waves are dealt round-robin into G = min(frames, waves)
// groups; group k takes the frames k, k + G, ... one after the other, and within a frame its waves draw
// tiles from that frame's ticket counter in scan order.  A wave only ever waits for tiles of the SAME frame
// with a LOWER ticket, and every ticket that was handed out is held by a wave that is running, so the
// lowest unfinished tile of every frame can always finish -- whatever the dispatch order, the number of
// resident workgroups or the other kernels on the device (MI355X_MICROARCH.md: nothing may depend on
// dispatch order or co-residency).  Per-frame counters keep the pullers per counter few (one counter
// saturates at ~88 returning atomics per microsecond) and the look-back chains short: a batch of 128
// frames has 16 waves per frame in flight, so a wave waits for the slowest of 16, not of 2048.  Every spin is bounded and
// gives up with MI355_E_INTERNAL.  Cross-workgroup data are 8-byte {epoch, payload} granules written by one
// sc1 store and polled with sc1 loads (per-XCD L2s are not coherent; cdna_hip_programming.md Guideline 16,
// form R2): the epoch changes with every launch, so stale records of earlier launches never match.
#include "jpeg_screen_devfn.h"

namespace mi355 {

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr uint32_t kFusedWaves = 8;              // waves per workgroup; one workgroup per CU (LDS-bound)
constexpr uint32_t kRowsY = 20, kRowsC = 9;      // words per unit kept in LDS: 640 / 288 bits
constexpr uint32_t kSlotWave = (kRowsY + 2 * kRowsC + 1) * 64;  // three string slots + one dump row, words
constexpr uint32_t kWinWords = 64 * 32;          // merge window = the row buffer without its sentinel row
constexpr uint32_t kSpinLimit = 1u << 20;        // polls (>= ~0.5 us each) before a wait gives up

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits until the granule at p (wave-uniform address) carries `epoch`; returns it.  On give-up sets
// status bit 2 (MI355_E_INTERNAL), returns a granule with payload 0 and zeroes `limit`: after the first
// give-up anywhere in the launch (every wave re-reads the status word per tile) no wait spins any more, so a
// broken launch drains in milliseconds instead of timing out tile after tile.
__device__ __forceinline__ unsigned long long wait_granule(const unsigned long long* p, uint32_t epoch, uint32_t* status,
                                                           uint32_t& limit) {
    unsigned long long v = 0;
    for (uint32_t spins = 0;; ++spins) {
        v = granule_load(p);
        if ((uint32_t)(v >> 48) == epoch) break;
        if (spins >= limit) {
            if (limit) atomicOr(status, 4u);
            limit = 0;
            v = (unsigned long long)epoch << 48;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Exclusive bit offset of `tile` inside its frame: sum of the totals of tiles 0 .. tile-1, by decoupled
// look-back over their scan granules ({epoch:16, state:2 (1 = total, 2 = inclusive prefix), value:46}).
// All 64 lanes take part, one predecessor each per round.
__device__ __forceinline__ unsigned long long tile_lookback(const unsigned long long* rec /* frame's records */, uint32_t tile,
                                                            uint32_t epoch, uint32_t lane, uint32_t* status, uint32_t& limit) {
    constexpr unsigned long long kVal = (1ull << 46) - 1;
    unsigned long long excl = 0;
    int top = (int)tile - 1;  // nearest predecessor not yet accounted for
    while (top >= 0) {
        const int idx = top - (int)lane;
        const bool valid = idx >= 0;
        const uint32_t nvalid = top + 1 < 64 ? (uint32_t)(top + 1) : 64u;
        unsigned long long v = 0;
        uint32_t firstP = 64;
        for (uint32_t spins = 0;; ++spins) {
            if (valid) v = granule_load(rec + (size_t)idx * 4 + 1);
            const bool pub = valid && (uint32_t)(v >> 48) == epoch;
            const bool isP = pub && ((uint32_t)(v >> 46) & 3u) == 2u;
            const unsigned long long bpub = __ballot(pub), bP = __ballot(isP);
            firstP = bP ? (uint32_t)__builtin_ctzll(bP) : 64u;
            const uint32_t upto = firstP < nvalid - 1 ? firstP : nvalid - 1;  // last lane that must have published
            const unsigned long long need = upto >= 63 ? ~0ull : ((2ull << upto) - 1ull);
            if ((bpub & need) == need) break;
            if (spins >= limit) {
                if (limit && lane == 0) atomicOr(status, 4u);
                limit = 0;
                return excl;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        // totals of the lanes nearer than the first prefix (each < 2^19: the sum fits 32 bits)
        const uint32_t agg = (valid && lane < firstP) ? (uint32_t)(v & kVal) : 0u;
        excl += wave_sum(agg);
        if (firstP < 64) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, firstP);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), firstP);
            excl += (((unsigned long long)hi << 32) | lo) & kVal;
            break;
        }
        top -= 64;
    }
    return excl;
}

// Appends bits to the tile's merge window (LDS, big-endian words).  A lane's region is contiguous and only
// its first and last word can be shared with the neighbouring lanes: those are OR-ed, the others stored.
// Words outside [wbase, wbase + kWinWords) belong to another round of an oversized tile and are skipped.
struct WindowAppender {
    uint32_t* win;
    uint32_t wbase;  // first word of this round
    uint32_t w;      // current word (tile-relative)
    uint32_t acc;    // bits already placed in the current word, left-aligned
    uint32_t n;      // their count, 0..31
    bool shared;     // the current word may hold a neighbour's bits
    __device__ __forceinline__ void emit(uint32_t v, bool or_it) {
        const uint32_t i = w - wbase;
        if (i < kWinWords) {
            if (or_it) atomicOr(&win[i], v);
            else win[i] = v;
        }
    }
    // ml: bits left-aligned, t: their count (1..32)
    __device__ __forceinline__ void put(uint32_t ml, uint32_t t) {
        const uint32_t hi = acc | (ml >> n);
        const uint32_t n2 = n + t;
        if (n2 >= 32u) {
            emit(hi, shared);
            shared = false;
            acc = __builtin_amdgcn_alignbit(ml, 0u, n);  // ml << (32 - n), 0 when n == 0
            ++w;
            n = n2 - 32u;
        } else {
            acc = hi;
            n = n2;
        }
    }
    __device__ __forceinline__ void finish() {
        if (n) emit(acc, true);
    }
};

struct StoreOvf {  // oversized strings: [word][lane] in the wave's private overflow area (global memory)
    uint32_t* dst;
    __device__ __forceinline__ void operator()(uint32_t w, uint32_t v) const {
        dst[(w < kSlotWordsFull - 1 ? w : kSlotWordsFull - 1) * 64u] = v;
    }
};

// MODE 0: strict (the reference's arithmetic); 1: standard 4:4:4.
template <bool PROBE, int MODE>
__global__ void __launch_bounds__(512, 2)
    k_encode_fused(Geom g, uint32_t n_frames, const uint8_t* __restrict__ rgb, FusedParams fp) {
    constexpr bool STD = MODE != 0;
    __shared__ uint32_t s_rows_all[kFusedWaves][kRowWords];            // zig-zag rows [position][unit]; later the merge window
    __shared__ alignas(16) uint32_t s_slot_all[kFusedWaves][kSlotWave];  // AC strings of the three passes, [word][lane]
    __shared__ uint32_t s_mask_all[kFusedWaves][2][64];                // non-zero masks (lo, hi)
    __shared__ float s_qf[2][16][8];      // per group of 4 positions: 2^-23/Q x4, first-look thresholds x4
    __shared__ uint32_t s_act[2][256];    // (run,size) AC tables
    __shared__ uint32_t s_lut2[2][kLut2Words];  // (value,run) symbol tables
    __shared__ uint32_t s_dc[2][16];      // DC tables

    const ScreenParams& sp = fp.sp;
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
    uint32_t* s_rows = s_rows_all[wv];
    uint32_t* s_slot = s_slot_all[wv];
    uint32_t* s_mlo = s_mask_all[wv][0];
    uint32_t* s_mhi = s_mask_all[wv][1];
    i16a* const tb16 = reinterpret_cast<i16a*>(s_rows);
    if (tid < 512) (&s_act[0][0])[tid] = sp.lut[512 + tid];
    for (uint32_t i = tid; i < 2 * kLut2Words; i += 512) (&s_lut2[0][0])[i] = sp.lut2[i];
    if (tid < 256) (&s_qf[0][0][0])[tid] = sp.qconst_f[tid];
    if (tid < 32) s_dc[tid >> 4][tid & 15] = sp.lut[(tid >> 4) * 256 + (tid & 15)];
    if (lane < 32) s_rows[64 * 32 + lane] = kRowSentinel * 0x00010001u;  // sentinel row after zig-zag position 63 (never written again)
    v4i A[4][kLookDigits];
    load_look_fragments(sp, lane, A);
    __syncthreads();

    const uint32_t gwave = blockIdx.x * kFusedWaves + wv;
    const uint32_t nwaves = gridDim.x * kFusedWaves;
    const uint32_t groups = n_frames < nwaves ? n_frames : nwaves;
    const uint32_t group = gwave % groups;
    // One frame only: every wave pulls from the same counter.  The workgroup draws its first eight tickets
    // with one atomic, and when the frame has no more tiles than the grid has waves nobody draws again.
    const bool wg_draw = groups == 1;
    const bool single_round = wg_draw && g.tiles <= nwaves;
    __shared__ uint32_t s_first;
    const uint32_t epoch = fp.epoch;
    uint32_t* const ovf = fp.ovf + (size_t)gwave * 3 * kSlotWordsFull * 64;  // this wave's overflow area

    uint32_t spin_limit = kSpinLimit;
    bool walk_general[2] = {false, false};  // per channel type: the last pass had a symbol-table miss (walk_nonzeros)
    bool first_draw = true;
    for (uint32_t frame = group; frame < n_frames; frame += groups) {
    // ticket = tile index inside the frame, in scan order
    uint32_t t;
    if (wg_draw && first_draw) {
        if (tid == 0) s_first = atomicAdd(&fp.ticket[frame], kFusedWaves);
        __syncthreads();
        t = s_first + wv;
    } else {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&fp.ticket[frame], 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
    first_draw = false;
    while (t < g.tiles) {
        const uint32_t tile = t;
        // a wait that gave up anywhere poisons the launch: stop waiting (the value is not needed before the merge)
        const uint32_t poisoned = __hip_atomic_load(sp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u;
        const uint8_t* f = rgb + (size_t)frame * g.frame_stride;
        unsigned long long* const rec = fp.rec + (size_t)frame * g.tiles * 4;  // this frame's records
        const uint32_t nblk = g.N - tile * 64 < 64u ? g.N - tile * 64 : 64u;   // active blocks of the tile
        const bool active = lane < nblk;

        // block coordinates of this lane's four blocks (16j + n), and whether the whole tile lies inside the image
        uint32_t bxs[4], bys[4];
        bool interior = true;
        {
            uint32_t b = tile * 64 + n;
            uint32_t by = b / g.nbx, bx = b - by * g.nbx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bb = tile * 64 + 16 * j + n;
                if (bb >= g.N) {  // past the last block: any valid block will do, the lane is masked later
                    bx = g.nbx - 1;
                    by = g.N / g.nbx - 1;
                }
                bxs[j] = bx;
                bys[j] = by;
                interior = interior && (bx * 8 + 8 <= g.W) && (by * 8 + 8 <= g.H);
                bx += 16;
                while (bx >= g.nbx) {
                    bx -= g.nbx;
                    ++by;
                }
            }
        }
        const bool fast = g.fast_rows && __all(interior);

        // per-channel results kept for the merge: AC length << 16 | (uint16) DC, per lane; which strings are oversized
        uint32_t st0 = 0, st1 = 0, st2 = 0;
        unsigned long long ovmask0 = 0, ovmask1 = 0, ovmask2 = 0;
        uint32_t next_ticket = 0;

#pragma unroll 1
        for (uint32_t chan = 0; chan < 3; ++chan) {
            const uint32_t ct = chan ? 1u : 0u;
            const bool avg = !STD && (chan != 0) && (g.flags & 1u);  // standard mode never replicates chroma means
            s_mlo[lane] = 0;
            s_mhi[lane] = 0;
            __builtin_amdgcn_wave_barrier();

            // raw RGB of unit-tile j+1 is fetched while unit-tile j is processed
            uint32_t raw[12];
            uint32_t dcsum = 0;  // sample sum of the block whose coefficient 0 this lane will form
            if (fast) load_raw_rowpair(f, g, bxs[0], bys[0], gq, raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bx = bxs[j], by = bys[j];
                uint32_t pk[4];
                if (fast) {
                    uint32_t cur[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) cur[i] = raw[i];
                    if (j < 3) load_raw_rowpair(f, g, bxs[j + 1], bys[j + 1], gq, raw);
                    if (chan == 0) convert_rowpair<0, STD>(cur, false, pk);
                    else if (chan == 1) convert_rowpair<1, STD>(cur, avg, pk);
                    else convert_rowpair<2, STD>(cur, avg, pk);
                } else {
                    if (chan == 0) generic_rowpair<0, STD>(f, g, false, bx, by, gq, pk);
                    else if (chan == 1) generic_rowpair<1, STD>(f, g, avg, bx, by, gq, pk);
                    else generic_rowpair<2, STD>(f, g, avg, bx, by, gq, pk);
                }
                if constexpr (PROBE) {
                    if (sp.samples && tile * 64 + 16 * j + n < g.N) {
#pragma unroll
                        for (int sidx = 0; sidx < 16; ++sidx) {
                            const uint32_t v = (pk[sidx >> 2] >> (8 * (sidx & 3))) & 255u;
                            const size_t px = (size_t)(by * 8 + gq * 2 + (sidx >> 3)) * g.W8 + bx * 8 + (sidx & 7);
                            sp.samples[((size_t)frame * g.W8 * g.H8 + px) * 3 + chan] = (uint8_t)v;
                        }
                    }
                }
                // sum of the block's 64 samples (for the exact DC): 16 in this lane, then over the 4 row-pair lanes
                uint32_t ssum = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) ssum = __builtin_amdgcn_sad_u8(pk[i], 0u, ssum);
                ssum += __shfl_xor(ssum, 16);
                ssum += __shfl_xor(ssum, 32);
                // level shift: sample - 128 as int8 == sample ^ 0x80
                const v4i B = v4i{(int)(pk[0] ^ 0x80808080u), (int)(pk[1] ^ 0x80808080u),
                                  (int)(pk[2] ^ 0x80808080u), (int)(pk[3] ^ 0x80808080u)};
                // coefficient 0 is formed exactly after this loop, by the lane (n, gq == j) for unit 16j+n
                if (gq == (uint32_t)j) dcsum = ssum;

                bool amb = false;
                uint32_t nzlo = 0, nzhi = 0;  // this lane's part of the unit's non-zero mask
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    uint32_t qb[4];  // low 16 bits = quantised value
                    screen_quantise<STD>(A[mt], B, sp, &s_qf[ct][4 * mt + gq][0], ct, mt, gq, lane, qb, amb);
                    // zig-zag positions 16mt+4gq .. +3 of unit 16j+n -> row buffer + non-zero bits
                    i16a* row = tb16 + (16 * mt + 4 * gq) * 64 + row_unit_off(16 * j + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[r * 64] = (int16_t)qb[r];
                    const uint32_t nib = ((qb[0] & 0xffffu) ? 1u : 0u) | ((qb[1] & 0xffffu) ? 2u : 0u) |
                                         ((qb[2] & 0xffffu) ? 4u : 0u) | ((qb[3] & 0xffffu) ? 8u : 0u);
                    if (mt < 2) nzlo |= nib << (16 * mt);
                    else nzhi |= nib << (16 * (mt - 2));
                }
                atomicOr(&s_mlo[16 * j + n], (nzlo << (4 * gq)) & ~1u);
                atomicOr(&s_mhi[16 * j + n], nzhi << (4 * gq));
                if (amb) atomicOr(&s_mlo[16 * j + n], 1u);  // bit 0 (coefficient 0 is never walked) = "undecided unit"
            }
            {
                // exact coefficient 0 of unit 16*gq + n.  Strict: c0 = fl(sum * SCALE_00), q0 = round(c0 / Q0)
                // (utils.cpp:336,459).  Standard: row 0 of the true DCT is exactly 1/8,
                // q0 = round-half-away(sum / (8 Q0)) in integers.
                int q0;
                if constexpr (STD) {
                    const int sl = (int)dcsum - 8192;
                    const uint32_t Q0 = (uint32_t)sp.qd[ct * 64], a0 = (uint32_t)(sl < 0 ? -sl : sl);
                    const int n0 = (int)((a0 + 4u * Q0) / (8u * Q0));
                    q0 = sl < 0 ? -n0 : n0;
                } else {
                    const double c0 = (double)((int)dcsum - 8192) * kScale00;
                    q0 = (int)__builtin_round(c0 / sp.qd[ct * 64]);
                }
                tb16[row_unit_off(16 * gq + n)] = (int16_t)q0;
            }
            __builtin_amdgcn_wave_barrier();

            // ---- walk phase: lane = block.  This pass's string slot (still unused) doubles as the scratch of
            // the exact recomputation.
            const uint32_t slot_off = chan == 0 ? 0u : (chan == 1 ? kRowsY * 64u : (kRowsY + kRowsC) * 64u);
            const uint32_t slot_rows = chan == 0 ? kRowsY : kRowsC;
            uint32_t* const slot = s_slot + slot_off;
            if constexpr (!STD) {
                // Units with a coefficient the screen could not decide: the exact chain is the arbiter.
                const bool undecided = active && (s_mlo[lane] & 1u) != 0;
                unsigned long long todo = __ballot(undecided);
                while (todo) {  // wave-uniform: one unit at a time, the whole wave on it
                    const uint32_t ul = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    if (lane == 0) atomicAdd(&sp.stats[1], 1ull);
                    const uint32_t ub = tile * 64 + ul, uby = ub / g.nbx, ubx = ub - uby * g.nbx;
                    exact_unit_wave(f, g, chan, ubx, uby, sp.qd, reinterpret_cast<double*>(slot), tb16 + row_unit_off(ul),
                                    &s_mlo[ul], &s_mhi[ul], lane);
                }
            }
            i16a* const row16 = tb16 + row_unit_off(lane);
            unsigned long long mask = ((unsigned long long)s_mhi[lane] << 32 | s_mlo[lane]) & ~1ull;
            const int dc = (int)row16[0];

            if constexpr (PROBE) {
                if (sp.coefs) {
                    uint32_t* dst = sp.coefs + (((size_t)frame * g.tiles + tile) * 3 + chan) * 2048 + lane;
#pragma unroll
                    for (int pp = 0; pp < 32; ++pp)
                        dst[pp * 64] = active ? (((uint32_t)(uint16_t)row16[2 * pp * 64]) |
                                                 ((uint32_t)(uint16_t)row16[(2 * pp + 1) * 64] << 16))
                                              : 0u;
                }
            }
            if (chan == 2) {
                // the tile's last DCs for the successor's first DC differences; and the next ticket, so that its
                // latency hides behind the walk
                const uint32_t last = nblk - 1;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)st0, last) & 0xffffu;
                const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)st1, last) & 0xffffu;
                const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane(dc, last) & 0xffffu;
                if (lane == 0) {
                    granule_store(rec + (size_t)tile * 4,
                                  ((unsigned long long)epoch << 48) | ((unsigned long long)d2 << 32) | (d1 << 16) | d0);
                    next_ticket = single_round ? 0xFFFFFFFFu : atomicAdd(&fp.ticket[frame], 1u);
                }
            }

            Packer32<StoreLds> pkr(StoreLds{slot + lane, slot_rows, (kRowsY + 2 * kRowsC) * 64u - slot_off});
            mask = mark_zero_runs(mask);  // ZRL positions become virtual non-zeros (after the probe dump above)
            const uint32_t maxcnt = wave_max((uint32_t)__popcll(mask));
            const bool ok = walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pkr, maxcnt, walk_general[ct]);
            const uint32_t aclen = pkr.bits();
            const bool oversize = active && pkr.words() > slot_rows;
            if (!ok && active) atomicOr(sp.status, 1u);  // MI355_E_CATEGORY
            const unsigned long long ovm = __ballot(oversize);
            if (ovm) {  // rare: a string longer than its slot: walk again, straight to this wave's overflow area
                if (oversize) {
                    Packer32<StoreOvf> pg(StoreOvf{ovf + (size_t)chan * kSlotWordsFull * 64 + lane});
                    bool gen = true;
                    (void)walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pg, maxcnt, gen);
                }
            }
            const uint32_t stv = active ? ((aclen << 16) | ((uint32_t)dc & 0xffffu)) : 0u;
            if (chan == 0) st0 = stv, ovmask0 = ovm;
            else if (chan == 1) st1 = stv, ovmask1 = ovm;
            else st2 = stv, ovmask2 = ovm;
            __builtin_amdgcn_wave_barrier();
        }

        // =================== the tile's 192 units -> final bits ===================
        const bool restart = STD && (g.flags & 8u) != 0;  // MI355_F_RESTART: every tile is its own interval
        // DC predictors: the previous lane; lane 0: the previous tile's last block (0 at the start of a frame / interval)
        int dcv[3] = {meta_dc(st0), meta_dc(st1), meta_dc(st2)};
        int pred[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pred[c] = __shfl_up(dcv[c], 1);
        if (poisoned || (fp.debug & 1u)) spin_limit = 0;  // debug bit 0: never wait (timing experiments; output is garbage)
        if (tile > 0 && !restart) {
            const unsigned long long gdc = wait_granule(rec + (size_t)(tile - 1) * 4, epoch, sp.status, spin_limit);
            if (lane == 0) {
                pred[0] = (int)(int16_t)(gdc & 0xffffu);
                pred[1] = (int)(int16_t)((gdc >> 16) & 0xffffu);
                pred[2] = (int)(int16_t)((gdc >> 32) & 0xffffu);
            }
        } else if (lane == 0) {
            pred[0] = pred[1] = pred[2] = 0;
        }
        // DC symbols, left-aligned | length (<= 20 bits)
        uint32_t dcsym[3];
        const uint32_t aclen[3] = {st0 >> 16, st1 >> 16, st2 >> 16};
        uint32_t blk = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t e = 0;
            auto mk = [&](uint32_t code, uint32_t len) { e = (code << (32u - len)) | len; };
            const bool dc_ok = put_dc(dcv[c] - pred[c], s_dc[c ? 1 : 0], mk);
            if (!dc_ok && active) atomicOr(sp.status, 1u);  // MI355_E_CATEGORY
            dcsym[c] = e;
            blk += (e & 31u) + aclen[c];
        }
        if (!active) blk = 0;
        const uint32_t incl = wave_incl_scan(blk, lane);
        uint32_t tbits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t fill = restart ? ((8u - (tbits & 7u)) & 7u) : 0u;  // the interval ends on a byte boundary, filled with 1s
        tbits += fill;

        // publish the total, find the offset, publish the prefix
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) granule_store(rec + 1, ((unsigned long long)epoch << 48) | (2ull << 46) | tbits);
        } else {
            if (lane == 0) granule_store(rec + (size_t)tile * 4 + 1, ((unsigned long long)epoch << 48) | (1ull << 46) | tbits);
            excl = tile_lookback(rec, tile, epoch, lane, sp.status, spin_limit);
            if (lane == 0)
                granule_store(rec + (size_t)tile * 4 + 1, ((unsigned long long)epoch << 48) | (2ull << 46) | (excl + tbits));
        }
        const unsigned long long endbit = excl + tbits;
        const bool last_tile = tile + 1 == g.tiles;
        if (lane == 0) {
            fp.tile_off[(size_t)frame * (g.tiles + 1) + tile] = excl;
            if (last_tile) {
                fp.tile_off[(size_t)frame * (g.tiles + 1) + g.tiles] = endbit;
                fp.frame_bits[frame] = endbit;
            }
        }
        const bool room = ((endbit + 31) >> 5) * 4 <= fp.out_stride;
        const bool fits = room && fp.out != nullptr;  // no output buffer: stage probes
        if (!room && fp.out && lane == 0) atomicOr(sp.status, 2u);  // MI355_E_CAPACITY

        const uint32_t sb = (uint32_t)(excl & 31);                 // bit offset inside the tile's first word
        const unsigned long long w0 = excl >> 5;                   // the tile's first word in the frame's output
        const uint32_t nwords = (sb + tbits + 31) >> 5;            // words the tile touches
        const bool tail_shared = !last_tile && ((sb + tbits) & 31u) != 0;  // the last word continues in the next tile
        uint32_t* const outw = reinterpret_cast<uint32_t*>(fp.out + (size_t)frame * fp.out_stride);
        const uint32_t pos0 = sb + incl - blk;                     // this lane's first bit, tile-relative (from word w0)

        for (uint32_t wbase = 0; wbase < nwords; wbase += kWinWords) {  // one round unless the tile is huge
            const uint32_t cnt = nwords - wbase < kWinWords ? nwords - wbase : kWinWords;
            for (uint32_t i = lane; i < cnt; i += 64) s_rows[i] = 0;
            __builtin_amdgcn_wave_barrier();
            if (active) {
                WindowAppender ap{s_rows, wbase, pos0 >> 5, 0u, pos0 & 31u, true};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    ap.put(dcsym[c] & ~31u, dcsym[c] & 31u);
                    const uint32_t al = aclen[c];
                    const unsigned long long ovm = c == 0 ? ovmask0 : (c == 1 ? ovmask1 : ovmask2);
                    const bool ov = (ovm >> lane) & 1ull;
                    const uint32_t* src = ov ? ovf + (size_t)c * kSlotWordsFull * 64 + lane
                                             : s_slot + (c == 0 ? 0u : (c == 1 ? kRowsY * 64u : (kRowsY + kRowsC) * 64u)) + lane;
                    for (uint32_t done = 0; done < al; done += 32) {
                        const uint32_t word = src[(done >> 5) * 64u];
                        const uint32_t len = al - done < 32u ? al - done : 32u;
                        ap.put(word, len);
                    }
                }
                if (fill && lane == nblk - 1) ap.put(0xFFFFFFFFu << (32u - fill), fill);
                ap.finish();
            }
            __builtin_amdgcn_wave_barrier();
            // The word this tile shares with its successor leaves as a carry granule -- BEFORE this tile waits for
            // its own predecessor's carry (otherwise the carries would form a chain through the whole frame).
            // It does not depend on the incoming carry unless the whole tile lies inside one word.
            const bool last_round = wbase + cnt == nwords;
            const bool carry_late = nwords == 1 && sb != 0;
            if (last_round && tail_shared && !carry_late && lane == 0)
                granule_store(rec + (size_t)tile * 4 + 2, ((unsigned long long)epoch << 48) | s_rows[cnt - 1]);
            // the word shared with the previous tile arrives as its carry granule
            if (wbase == 0 && sb != 0) {
                const unsigned long long gc = wait_granule(rec + (size_t)(tile - 1) * 4 + 2, epoch, sp.status, spin_limit);
                if (lane == 0) s_rows[0] |= (uint32_t)gc;
                __builtin_amdgcn_wave_barrier();
            }
            if (last_round && tail_shared && carry_late && lane == 0)
                granule_store(rec + (size_t)tile * 4 + 2, ((unsigned long long)epoch << 48) | s_rows[0]);
            const uint32_t nstore = (last_round && tail_shared) ? cnt - 1 : cnt;
            if (fits)
                for (uint32_t i = lane; i < nstore; i += 64) outw[w0 + wbase + i] = __builtin_bswap32(s_rows[i]);
            __builtin_amdgcn_wave_barrier();
        }
        // the row buffer was the window: restore its sentinel row if a (huge) tile reached into it -- it cannot:
        // kWinWords stops short of it -- and go on with the next tile
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_ticket);
    }
    }
}

// ----------------------------------------------------------------------------
// launcher
// ----------------------------------------------------------------------------
uint32_t fused_grid(const Geom& g, uint32_t n_frames, uint32_t max_wgs) {
    const uint64_t tiles = (uint64_t)g.tiles * n_frames;
    uint64_t wgs = (tiles + kFusedWaves - 1) / kFusedWaves;
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    return (uint32_t)wgs;
}
size_t fused_ovf_words(uint32_t wgs) { return (size_t)wgs * kFusedWaves * 3 * kSlotWordsFull * 64; }

hipError_t launch_encode_fused(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const FusedParams& fp, bool probe,
                               uint32_t wgs, hipStream_t s) {
    const int mode = (g.flags & 2u) ? 1 : 0;  // MI355_F_STANDARD
#define MI355_LAUNCH_FUSED(PR, MD) \
    hipLaunchKernelGGL((k_encode_fused<PR, MD>), dim3(wgs), dim3(512), 0, s, g, n_frames, rgb, fp)
    if (probe) {
        if (mode == 1) MI355_LAUNCH_FUSED(true, 1);
        else MI355_LAUNCH_FUSED(true, 0);
    } else {
        if (mode == 1) MI355_LAUNCH_FUSED(false, 1);
        else MI355_LAUNCH_FUSED(false, 0);
    }
#undef MI355_LAUNCH_FUSED
    return hipGetLastError();
}

}  // namespace mi355
