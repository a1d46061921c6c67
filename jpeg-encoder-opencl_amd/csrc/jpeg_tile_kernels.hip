// Single-launch encode pipeline (gfx950 / CDNA4): RGB frames in, final scan bits out, ONE kernel, no workspace
// traffic.  Strict mode (the reference's arithmetic) and standard 4:4:4.
//
// A TEAM of three waves owns a tile of 64 blocks (= 192 consecutive units of the reference's scan order
// 3*block + chan, utils.cpp:665-695): wave c of the team runs channel c -- Y, Cb, Cr side by side, so the three
// passes over the tile's RGB bytes happen at the same time on the same CU (fetched from HBM once) -- and the team then
// merges its 192 strings itself:
//
//   per wave (channel), lane roles as in k_screen_encode:
//       integer-exact colour conversion (+ chroma mean, mirror padding)           performCSC/CDS/pad
//       int8-MFMA fixed-point map, three-digit first look, rare second look        performDCT+Quantization
//       transpose to zig-zag rows in LDS, exact fp64 chain for undecided units     (the arbiter)
//       per-unit RLE/Huffman walk over the non-zeros into an LDS string slot       performRLE/HuffmanEncoder
//       DC symbol (predecessor = previous lane; lane 0: the previous tile's last DC of this channel, handed over
//       by an 8-byte granule), unit bit totals -> LDS
//   team sync, then every wave: scan of the 64 block totals (all three waves form the same scan), its own units'
//   bit positions, and its strings appended to the tile's bit window in LDS (the zig-zag row buffers, dead by then);
//   meanwhile the Cr wave finds the tile's bit offset in the frame by decoupled look-back over the earlier tiles;
//   team sync, then the window leaves for memory shifted into place (one v_alignbit per word), coalesced; the word
//   a tile shares with its successor travels as a "carry" granule, so no output word is written by two teams and
//   nothing is zeroed beforehand.
//
// Why three waves per tile instead of one wave running the three passes in turn (the first single-launch kernel,
// round 2): that one re-read its tile's RGB 13 us apart (2.0 x the HBM fetch), and it needed 256 VGPRs.  Here the A
// fragments of the map are NOT resident: the quantiser runs row-tile pair by row-tile pair over the four B operands of
// the wave's 64 units (16 VGPRs), so a pair's fragments (24 VGPRs, 12 KB shared by every wave of the CU: L1/L2 hits)
// are loaded twice per tile instead of living in 48 registers.  That brings the kernel under 168 VGPRs, and the LDS
// image of a team (37.8 KB) lets FOUR teams = twelve waves share a CU and one set of tables: three waves per SIMD
// instead of the two of k_screen_encode.
//
// Work distribution and forward progress: teams are dealt round-robin into G = min(frames, teams) groups; group k
// takes the frames k, k + G, ... one after the other, and within a frame its teams draw tiles from that frame's
// ticket counter in scan order.  A team only ever waits for tiles of the SAME frame with a LOWER ticket, and every
// ticket that was handed out is held by a team that is running, so the lowest unfinished tile of every frame can
// always finish -- whatever the dispatch order, the number of resident workgroups or the other kernels on the device
// (MI355X_MICROARCH.md: nothing may depend on dispatch order or co-residency).  The three waves of a team belong to
// one workgroup (co-resident by construction) and meet at counters in LDS.  Every spin on another team is bounded and
// gives up with MI355_E_INTERNAL.  Cross-workgroup data are 8-byte {epoch, payload} granules written by one sc1 store
// and polled with sc1 loads (per-XCD L2s are not coherent; cdna_hip_programming.md Guideline 16, form R2): the epoch
// changes with every launch, so stale records of earlier launches never match.
#include "jpeg_screen_devfn.h"

namespace mi355 {

typedef __attribute__((address_space(1))) unsigned long long gu64;

#ifndef MI355_TILE_TEAMS
#define MI355_TILE_TEAMS 4
#endif
constexpr uint32_t kTeams = MI355_TILE_TEAMS;    // teams per workgroup; one workgroup per CU (LDS-bound)
constexpr uint32_t kTileThreads = kTeams * 3 * 64;
constexpr uint32_t kRowsY = 20, kRowsC = 9;      // words per unit kept in LDS: 640 / 288 bits
constexpr uint32_t kRegionWords = 64 * 32;       // a wave's row buffer without its sentinel row = its part of the window
constexpr uint32_t kWinWords = 3 * kRegionWords; // the team's bit window
constexpr uint32_t kWinStep = kWinWords - 1;     // rounds of an oversized tile overlap by one word (see the write-out)
constexpr uint32_t kSpinLimit = 1u << 20;        // polls (>= ~0.5 us each) before a wait on another team gives up
constexpr uint32_t kRecGranules = 8;             // per tile: [0..2] last DC per channel, [3] scan state, [4] carry word

struct TeamLds {
    uint32_t rows[3][kRowWords];                 // zig-zag rows [position][unit] per wave; later the bit window
    alignas(16) uint32_t slot_y[(kRowsY + 1) * 64];      // AC strings [word][lane] + dump row
    alignas(16) uint32_t slot_c[2][(kRowsC + 1) * 64];
    uint32_t mask[3][2][64];                     // non-zero masks (lo, hi) per wave
    uint32_t ubits[3][64];                       // bits of every unit (DC symbol + AC string), per wave
    uint32_t arrive;                             // team sync counter (monotonic)
    uint32_t next_ticket;                        // the team's next tile (told at the first meeting of a tile)
    uint32_t first_ticket[2];                    // the team's first tile of a frame (alternating: a frame may hold no tile for the team)
    uint32_t excl_lo, excl_hi;                   // the tile's bit offset in its frame
    uint32_t err;                                // category error seen by a wave of the team in this tile
    uint32_t err_before;                         // an error in an earlier tile of the frame (from the look-back)
    uint32_t pad[8];
};

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The three waves of a team meet here.  `cnt` lives in LDS and only ever grows; `phase` is this wave's count of the
// meetings so far (x 3).  LDS operations of one wave execute in order, so everything the wave wrote to LDS before its
// arrival is visible to whoever sees the counter reach the phase.
// Bounded like every other wait (the partners run the same loop, so the bound is only ever reached through a defect):
// on give-up status bit 2 (MI355_E_INTERNAL) is set and `limit` zeroed, after which the wave no longer waits anywhere.
struct TeamSync {
    uint32_t* cnt;
    uint32_t* status;
    uint32_t phase = 0;
    uint32_t limit = 1u << 22;
    __device__ __forceinline__ void meet(uint32_t lane) {
        phase += 3;
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (uint32_t spins = 0; (int)(__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - phase) < 0; ++spins) {
            if (spins >= limit) {
                if (limit && lane == 0) atomicOr(status, 4u);
                limit = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
};

// Waits until the granule at p (wave-uniform address) carries `epoch`; returns it.  On give-up sets status bit 2
// (MI355_E_INTERNAL), returns a granule with payload 0 and zeroes `limit`: after the first give-up anywhere in the
// launch (every team re-reads the status word per tile) no wait spins any more, so a broken launch drains in
// milliseconds instead of timing out tile after tile.
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

// Scan granule: {epoch:16, state:2 (1 = the tile's total, 2 = inclusive prefix), err:1, value:45}.  err of a total:
// this tile saw an error; err of a prefix: this tile or an earlier one of the frame did.
constexpr unsigned long long kScanVal = (1ull << 45) - 1;
__device__ __forceinline__ unsigned long long scan_granule(uint32_t epoch, uint32_t state, bool err, unsigned long long v) {
    return ((unsigned long long)epoch << 48) | ((unsigned long long)state << 46) | ((unsigned long long)(err ? 1u : 0u) << 45) | v;
}

// Exclusive bit offset of `tile` inside its frame: sum of the totals of tiles 0 .. tile-1, by decoupled look-back
// over their scan granules.  All 64 lanes take part, FOUR predecessors each per round (a round covers 256 tiles: with
// every tile of a frame in flight at the same moment the prefixes travel 256 tiles per polling round).  `err` collects
// the error flags met on the way.
__device__ __forceinline__ unsigned long long tile_lookback(const unsigned long long* rec /* frame's records */, uint32_t tile,
                                                            uint32_t epoch, uint32_t lane, uint32_t* status, uint32_t& limit,
                                                            bool& err) {
    unsigned long long excl = 0;
    int top = (int)tile - 1;  // nearest predecessor not yet accounted for
    while (top >= 0) {
        // slot q = 4 * lane + i is predecessor top - q: nearer tiles in lower slots
        unsigned long long v[4];
        uint32_t first = 256;  // lowest slot holding a prefix
        const uint32_t nvalid = top + 1 < 256 ? (uint32_t)(top + 1) : 256u;
        for (uint32_t spins = 0;; ++spins) {
            uint32_t pubm = 0, prem = 0;  // this lane's slots: published / prefix
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = top - (int)(4 * lane + i);
                v[i] = idx >= 0 ? granule_load(rec + (size_t)idx * kRecGranules + 3) : 0ull;
                const bool pub = idx >= 0 && (uint32_t)(v[i] >> 48) == epoch;
                if (pub) pubm |= 1u << i;
                if (pub && ((uint32_t)(v[i] >> 46) & 3u) == 2u) prem |= 1u << i;
            }
            // first prefix slot over the wave
            const uint32_t myfirst = prem ? 4 * lane + (uint32_t)__builtin_ctz(prem) : 256u;
            // lanes hold increasing slots: the first lane with a prefix wins
            const unsigned long long bp = __ballot(prem != 0);
            first = bp ? (uint32_t)__builtin_amdgcn_readlane((int)myfirst, (int)__builtin_ctzll(bp)) : 256u;
            const uint32_t upto = first < nvalid - 1 ? first : nvalid - 1;  // last slot that must have published
            // every slot <= upto published?
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t q = 4 * lane + i;
                if (q <= upto && !((pubm >> i) & 1u)) ok = false;
            }
            if (__all(ok)) break;
            if (spins >= limit) {
                if (limit && lane == 0) atomicOr(status, 4u);
                limit = 0;
                return excl;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        // totals of the slots nearer than the first prefix (each < 2^19: a round's sum fits 32 bits), the prefix itself
        uint32_t agg = 0;
        uint32_t e = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t q = 4 * lane + i;
            if (q < first && q < nvalid) agg += (uint32_t)(v[i] & kScanVal), e |= (uint32_t)(v[i] >> 45) & 1u;
        }
        excl += wave_sum(agg);
        if (first < 256) {
            const uint32_t fl = first >> 2, fi = first & 3;
            const unsigned long long pv = fi == 0 ? v[0] : (fi == 1 ? v[1] : (fi == 2 ? v[2] : v[3]));
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv, (int)fl);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pv >> 32), (int)fl);
            const unsigned long long p = ((unsigned long long)hi << 32) | lo;
            excl += p & kScanVal;
            if (lane == fl) e |= (uint32_t)(p >> 45) & 1u;
        }
        err = err || wave_any(e != 0);
        if (first < 256) break;
        top -= 256;
    }
    return excl;
}

// Address of word i of the team's bit window: the window runs through the row buffers of waves 2, 1, 0 (the Cr wave
// has the least entropy work and a tile's bits rarely leave its part), 2048 words each, skipping the sentinel rows.
__device__ __forceinline__ uint32_t* win_word(TeamLds& tl, uint32_t i) { return &tl.rows[2u - (i >> 11)][i & 2047u]; }

// Appends bits to the window (big-endian words).  A unit's region is contiguous and only its first and last word can
// be shared with the neighbouring units (other waves): those are OR-ed, the others stored.  Words outside
// [wstart, wstart + kWinWords) belong to another round of an oversized tile and are skipped.
struct WindowAppender {
    TeamLds& tl;
    uint32_t wstart;  // first tile word of this round
    uint32_t w;       // current word (tile-relative)
    uint32_t acc;     // bits already placed in the current word, left-aligned
    uint32_t n;       // their count, 0..31
    bool shared;      // the current word may hold a neighbour's bits
    __device__ __forceinline__ void emit(uint32_t v, bool or_it) {
        const uint32_t i = w - wstart;
        if (i < kWinWords) {
            if (or_it) atomicOr(win_word(tl, i), v);
            else *win_word(tl, i) = v;
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

// A fragments of row tiles 2 mtp, 2 mtp + 1 (top three digits each).  `lane16` = lane * 16 made opaque by the caller
// inside the tile loop: left to itself the compiler hoists the twelve 64-bit fragment addresses out of the loop and then
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

// MODE 0: strict (the reference's arithmetic); 1: standard 4:4:4.
template <int MODE>
__global__ void __launch_bounds__(kTileThreads)
    k_encode_tile(Geom g, uint32_t n_frames, const uint8_t* __restrict__ rgb, TileParams tp) {
    constexpr bool STD = MODE != 0;
    __shared__ TeamLds s_team[kTeams];
    __shared__ float s_qf[2][16][8];            // per group of 4 positions: 2^-23/Q x4, first-look thresholds x4
    __shared__ uint32_t s_act[2][256];          // (run,size) AC tables
    __shared__ uint32_t s_lut2[2][kLut2Words];  // (value,run) symbol tables
    __shared__ uint32_t s_dc[2][16];            // DC tables

    const ScreenParams& sp = tp.sp;
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
    const uint32_t team = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wv / 3u));
    const uint32_t chan = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wv - 3u * (wv / 3u)));
    TeamLds& tl = s_team[team];
    uint32_t* const s_rows = tl.rows[chan];
    uint32_t* const slot = chan == 0 ? tl.slot_y : tl.slot_c[chan - 1];
    const uint32_t slot_rows = chan == 0 ? kRowsY : kRowsC;
    uint32_t* const s_mlo = tl.mask[chan][0];
    uint32_t* const s_mhi = tl.mask[chan][1];
    i16a* const tb16 = reinterpret_cast<i16a*>(s_rows);
    for (uint32_t i = tid; i < 512; i += kTileThreads) (&s_act[0][0])[i] = sp.lut[512 + i];
    for (uint32_t i = tid; i < 2 * kLut2Words; i += kTileThreads) (&s_lut2[0][0])[i] = sp.lut2[i];
    if (tid < 256) (&s_qf[0][0][0])[tid] = sp.qconst_f[tid];
    if (tid < 32) s_dc[tid >> 4][tid & 15] = sp.lut[(tid >> 4) * 256 + (tid & 15)];
    if (lane < 32) s_rows[64 * 32 + lane] = kRowSentinel * 0x00010001u;  // sentinel row after zig-zag position 63 (never written again)
    if (lane == 0 && chan == 0) tl.arrive = 0, tl.err = 0;
    const uint32_t ct = chan ? 1u : 0u;
    const double q0d = sp.qd[ct * 64];  // quantiser divisor of coefficient 0
    __syncthreads();

    const uint32_t gteam = blockIdx.x * kTeams + team;
    const uint32_t nteams = gridDim.x * kTeams;
    const uint32_t groups = n_frames < nteams ? n_frames : nteams;
    const uint32_t group = gteam % groups;
    const uint32_t epoch = tp.epoch;
    const uint32_t gwave = gteam * 3 + chan;
    uint32_t* const ovf = tp.ovf + (size_t)gwave * kSlotWordsFull * 64;  // this wave's overflow area
    const bool avg = !STD && (chan != 0) && (g.flags & 1u);  // standard mode never replicates chroma means
    const bool restart = STD && (g.flags & 8u) != 0;         // MI355_F_RESTART: every tile is its own interval

    const bool wg_draw = n_frames == 1;
    __shared__ uint32_t s_first;
    uint32_t fpar = 0;
    TeamSync ts{&tl.arrive, sp.status};
    uint32_t spin_limit = kSpinLimit;
    bool walk_general = false;  // the last tile had a symbol-table miss (walk_nonzeros)
    for (uint32_t frame = group; frame < n_frames; frame += groups) {
        const uint8_t* const f = rgb + (size_t)frame * g.frame_stride;
        unsigned long long* const rec = tp.rec + (size_t)frame * g.tiles * kRecGranules;  // this frame's records
        // first ticket of the frame: drawn by the team's first wave, told at a meeting.  One frame only: every team pulls
        // from the same counter (it saturates at ~88 returning atomics per microsecond), so the workgroup draws its
        // first tickets with one atomic.
        if (wg_draw) {
            if (tid == 0) s_first = atomicAdd(&tp.ticket[frame], kTeams);
            __syncthreads();  // reached once by every wave: n_frames == 1
            if (chan == 0 && lane == 0) tl.first_ticket[fpar] = s_first + team;
        } else if (chan == 0 && lane == 0) {
            tl.first_ticket[fpar] = atomicAdd(&tp.ticket[frame], 1u);
        }
        ts.meet(lane);
        uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.first_ticket[fpar]);
        fpar ^= 1u;
        while (t < g.tiles) {
            const uint32_t tile = t;
            // a wait that gave up anywhere poisons the launch: stop waiting
            if (__hip_atomic_load(sp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u) spin_limit = 0, ts.limit = 0;
            const uint32_t nblk = g.N - tile * 64 < 64u ? g.N - tile * 64 : 64u;  // active blocks of the tile
            const bool active = lane < nblk;
            uint32_t next_ticket = 0;
            if (chan == 0 && lane == 0) next_ticket = atomicAdd(&tp.ticket[frame], 1u);  // told at the first meeting below

            // ---------------- samples: the wave's 64 blocks as four B operands ----------------
            v4i B[4];
            uint32_t dcsum = 0;  // sample sum of the block whose coefficient 0 this lane will form
            v4i A[2][kLookDigits];
            uint32_t lane16 = lane * 16u;
            asm volatile("" : "+v"(lane16));  // see load_pair_fragments
            {
                // Block coordinates of this lane's four blocks (16j + n) are walked, not stored; a first walk finds out
                // whether the whole tile lies inside the image.
                uint32_t bx0, by0;
                {
                    uint32_t b = tile * 64 + n;
                    if (b >= g.N) b = g.N - 1;  // past the last block: any valid block will do, the lane is masked later
                    by0 = b / g.nbx, bx0 = b - by0 * g.nbx;
                }
                auto next_block = [&](uint32_t& bx, uint32_t& by, int j /* the block reached */) {
                    bx += 16;
                    while (bx >= g.nbx) {
                        bx -= g.nbx;
                        ++by;
                    }
                    if (tile * 64 + 16 * j + n >= g.N) {
                        bx = g.nbx - 1;
                        by = g.N / g.nbx - 1;
                    }
                };
                bool interior = true;
                {
                    uint32_t bx = bx0, by = by0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        interior = interior && (bx * 8 + 8 <= g.W) && (by * 8 + 8 <= g.H);
                        if (j < 3) next_block(bx, by, j + 1);
                    }
                }
                const bool fast = g.fast_rows && __all(interior);
                uint32_t raw[12];  // raw RGB of unit-tile j+1 is fetched while unit-tile j is converted
                uint32_t bx = bx0, by = by0;
                if (fast) load_raw_rowpair(f, g, bx, by, gq, raw);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t pk[4];
                    if (fast) {
                        uint32_t cur[12];
#pragma unroll
                        for (int i = 0; i < 12; ++i) cur[i] = raw[i];
                        if (j < 3) {
                            next_block(bx, by, j + 1);
                            load_raw_rowpair(f, g, bx, by, gq, raw);
                        } else {
                            load_pair_fragments(sp, lane16, 0, A);  // land while the last unit-tile is converted
                        }
                        if (chan == 0) convert_rowpair<0, STD>(cur, false, pk);
                        else if (chan == 1) convert_rowpair<1, STD>(cur, avg, pk);
                        else convert_rowpair<2, STD>(cur, avg, pk);
                    } else {
                        if (chan == 0) generic_rowpair<0, STD>(f, g, false, bx, by, gq, pk);
                        else if (chan == 1) generic_rowpair<1, STD>(f, g, avg, bx, by, gq, pk);
                        else generic_rowpair<2, STD>(f, g, avg, bx, by, gq, pk);
                        if (j < 3) next_block(bx, by, j + 1);
                        else load_pair_fragments(sp, lane16, 0, A);
                    }
                    // sum of the block's 64 samples (for the exact DC): 16 in this lane, then over the 4 row-pair lanes
                    uint32_t ssum = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ssum = __builtin_amdgcn_sad_u8(pk[i], 0u, ssum);
                    ssum += __shfl_xor(ssum, 16);
                    ssum += __shfl_xor(ssum, 32);
                    if (gq == (uint32_t)j) dcsum = ssum;  // coefficient 0 of unit 16j+n is formed by the lane (n, gq == j)
                    // level shift: sample - 128 as int8 == sample ^ 0x80
                    B[j] = v4i{(int)(pk[0] ^ 0x80808080u), (int)(pk[1] ^ 0x80808080u), (int)(pk[2] ^ 0x80808080u),
                               (int)(pk[3] ^ 0x80808080u)};
                }
            }

            // ---------------- map + quantise + verify, row-tile pair by row-tile pair ----------------
            s_mlo[lane] = 0;
            s_mhi[lane] = 0;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mtp = 0; mtp < 2; ++mtp) {
                if (mtp == 1) load_pair_fragments(sp, lane16, 1, A);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool amb = false;
                    uint32_t qa[4], qb[4];  // low 16 bits = quantised value
                    screen_quantise<STD>(A[0], B[j], sp, &s_qf[ct][4 * (2 * mtp) + gq][0], ct, 2 * mtp, gq, lane, qa, amb);
                    screen_quantise<STD>(A[1], B[j], sp, &s_qf[ct][4 * (2 * mtp + 1) + gq][0], ct, 2 * mtp + 1, gq, lane, qb, amb);
                    // zig-zag positions 16mt+4gq .. +3 of unit 16j+n -> row buffer
                    i16a* const row = tb16 + (32 * mtp + 4 * gq) * 64 + row_unit_off(16 * j + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) row[r * 64] = (int16_t)qa[r], row[(16 + r) * 64] = (int16_t)qb[r];
                    // non-zero bits, two values per instruction (see k_screen_encode): flags of positions 32mtp + 4gq + r
                    // and 32mtp + 16 + 4gq + r at bits r and 16 + r, shifted in by 4gq
                    uint32_t w = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t pr = __builtin_amdgcn_perm(qb[r], qa[r], 0x05040100u);  // qa.lo16 | qb.lo16 << 16
                        uint32_t fl;
                        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(fl) : "v"(pr));
                        w |= fl << r;
                    }
                    if (mtp == 0) atomicOr(&s_mlo[16 * j + n], (w << (4 * gq)) & ~1u);
                    else atomicOr(&s_mhi[16 * j + n], w << (4 * gq));
                    if (amb) atomicOr(&s_mlo[16 * j + n], 1u);  // bit 0 (coefficient 0 is never walked) = "undecided unit"
                }
            }
            {
                // exact coefficient 0 of unit 16*gq + n.  Strict: c0 = fl(sum * SCALE_00), q0 = round(c0 / Q0)
                // (utils.cpp:336,459).  Standard: row 0 of the true DCT is exactly 1/8,
                // q0 = round-half-away(sum / (8 Q0)) in integers.
                int q0;
                if constexpr (STD) {
                    const int sl = (int)dcsum - 8192;
                    const uint32_t Q0 = (uint32_t)q0d, a0 = (uint32_t)(sl < 0 ? -sl : sl);
                    const int n0 = (int)((a0 + 4u * Q0) / (8u * Q0));
                    q0 = sl < 0 ? -n0 : n0;
                } else {
                    const double c0 = (double)((int)dcsum - 8192) * kScale00;
                    q0 = (int)__builtin_round(c0 / q0d);
                }
                tb16[row_unit_off(16 * gq + n)] = (int16_t)q0;
            }
            __builtin_amdgcn_wave_barrier();

            // ---------------- walk phase: lane = block ----------------
            if constexpr (!STD) {
                // Units with a coefficient the screen could not decide: the exact chain is the arbiter (scratch: this
                // wave's string slot, still unused).
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
            // the tile's last DC of this channel, for the successor's first DC difference
            {
                const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane(dc, (int)(nblk - 1)) & 0xffffu;
                if (lane == 0) granule_store(rec + (size_t)tile * kRecGranules + chan, ((unsigned long long)epoch << 48) | dl);
            }

            Packer32<StoreLds> pkr(StoreLds{slot + lane, slot_rows, slot_rows * 64u});
            mask = mark_zero_runs(mask);  // ZRL positions become virtual non-zeros
            const uint32_t maxcnt = wave_max((uint32_t)__popcll(mask));
            const bool ok = walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pkr, maxcnt, walk_general);
            const uint32_t aclen = pkr.bits();
            const bool oversize = active && pkr.words() > slot_rows;
            bool bad = !ok && active;  // MI355_E_CATEGORY
            const unsigned long long ovm = __ballot(oversize);
            if (ovm) {  // rare: a string longer than its slot: walk again, straight to this wave's overflow area
                if (oversize) {
                    Packer32<StoreOvf> pg(StoreOvf{ovf + lane});
                    bool gen = true;
                    (void)walk_nonzeros<STD>(row16, mask, s_lut2[ct], s_act[ct], pg, maxcnt, gen);
                }
            }

            // ---------------- DC symbol, unit totals ----------------
            int pred = __shfl_up(dc, 1);
            if (tile > 0 && !restart) {
                const unsigned long long gdc = wait_granule(rec + (size_t)(tile - 1) * kRecGranules + chan, epoch, sp.status, spin_limit);
                if (lane == 0) pred = (int)(int16_t)(gdc & 0xffffu);
            } else if (lane == 0) {
                pred = 0;
            }
            uint32_t dcsym = 0;  // left-aligned | length (<= 22 bits)
            {
                auto mk = [&](uint32_t code, uint32_t len) { dcsym = (code << (32u - len)) | len; };
                const bool dc_ok = put_dc(dc - pred, s_dc[ct], mk);
                bad = bad || (!dc_ok && active);
            }
            if (wave_any(bad)) {
                if (lane == 0) {
                    atomicOr(sp.status, 1u);  // MI355_E_CATEGORY
                    atomicOr(&tl.err, 1u);
                }
            }
            tl.ubits[chan][lane] = active ? (dcsym & 31u) + aclen : 0u;
            // this wave's rows are dead: its part of the window starts out zero
            {
                uint4* const z = reinterpret_cast<uint4*>(s_rows);
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            if (chan == 0 && lane == 0) tl.next_ticket = next_ticket;
            ts.meet(lane);  // ---- meeting 1: totals, zeroed window, next ticket

            const uint32_t u0 = tl.ubits[0][lane], u1 = tl.ubits[1][lane], u2 = tl.ubits[2][lane];
            const uint32_t blk = u0 + u1 + u2;
            const uint32_t incl = wave_incl_scan(blk, lane);
            uint32_t tbits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t fill = restart ? ((8u - (tbits & 7u)) & 7u) : 0u;  // the interval ends on a byte boundary, filled with 1s
            tbits += fill;
            const uint32_t pos0 = incl - blk + (chan >= 1 ? u0 : 0u) + (chan >= 2 ? u1 : 0u);  // this lane's first bit, tile-relative
            const bool tile_err = tl.err != 0;
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.next_ticket);
            const bool last_tile = tile + 1 == g.tiles;
            const uint32_t twords = (tbits + 31) >> 5;  // words of the tile-relative bit string

            if (chan == 2 && lane == 0 && tile > 0)
                granule_store(rec + (size_t)tile * kRecGranules + 3, scan_granule(epoch, 1u, tile_err, tbits));

            const bool ov = (ovm >> lane) & 1ull;
            const uint32_t* const src = ov ? ovf + lane : slot + lane;
            unsigned long long excl = 0;
            bool err_before = false;
            for (uint32_t wstart = 0;; wstart += kWinStep) {  // one round unless the tile is huge
                if (wstart) {  // later rounds: everybody is done with the window, then it starts out zero again
                    ts.meet(lane);
                    uint4* const z = reinterpret_cast<uint4*>(s_rows);
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                    ts.meet(lane);
                }
                if (active) {
                    WindowAppender ap{tl, wstart, pos0 >> 5, 0u, pos0 & 31u, true};
                    ap.put(dcsym & ~31u, dcsym & 31u);
                    for (uint32_t done = 0; done < aclen; done += 32) {
                        const uint32_t word = src[(done >> 5) * 64u];
                        const uint32_t len = aclen - done < 32u ? aclen - done : 32u;
                        ap.put(word, len);
                    }
                    if (fill && chan == 2 && lane == nblk - 1) ap.put(0xFFFFFFFFu << (32u - fill), fill);
                    ap.finish();
                }
                if (wstart == 0 && chan == 2) {
                    // the tile's place in the frame (the predecessors published their totals at THEIR first meeting)
                    if (tile > 0) {
                        excl = tile_lookback(rec, tile, epoch, lane, sp.status, spin_limit, err_before);
                        if (lane == 0)
                            granule_store(rec + (size_t)tile * kRecGranules + 3,
                                          scan_granule(epoch, 2u, tile_err || err_before, excl + tbits));
                    } else if (lane == 0) {
                        granule_store(rec + 3, scan_granule(epoch, 2u, tile_err, tbits));
                    }
                    if (lane == 0) {
                        tl.excl_lo = (uint32_t)excl, tl.excl_hi = (uint32_t)(excl >> 32);
                        tl.err_before = err_before ? 1u : 0u;
                        tl.err = 0;  // re-armed for the next tile (every wave has read it)
                    }
                }
                ts.meet(lane);  // ---- meeting 2: the window is complete, the offset known
                if (wstart == 0) {
                    excl = ((unsigned long long)tl.excl_hi << 32) | tl.excl_lo;
                    err_before = tl.err_before != 0;
                }
                const unsigned long long endbit = excl + tbits;
                const bool room = ((endbit + 31) >> 5) * 4 <= tp.out_stride;
                const bool fits = room && tp.out != nullptr;
                const uint32_t sb = (uint32_t)(excl & 31);                  // bit offset inside the tile's first output word
                const unsigned long long w0 = excl >> 5;                    // the tile's first word in the frame's output
                const uint32_t nwords = (sb + tbits + 31) >> 5;             // output words the tile touches
                const bool tail_shared = !last_tile && ((sb + tbits) & 31u) != 0;  // the last word continues in the next tile
                uint32_t* const outw = reinterpret_cast<uint32_t*>(tp.out + (size_t)frame * tp.out_stride);
                if (wstart == 0 && chan == 2 && lane == 0) {
                    if (!room && tp.out) atomicOr(sp.status, 2u);  // MI355_E_CAPACITY
                    tp.tile_off[(size_t)frame * (g.tiles + 1) + tile] = excl;
                    if (last_tile) {
                        tp.tile_off[(size_t)frame * (g.tiles + 1) + g.tiles] = endbit;
                        // a frame with an error anywhere says so in its bit count; the other frames of the batch are good
                        tp.frame_bits[frame] = (tile_err || err_before || (!room && tp.out)) ? ~0ull : endbit;
                    }
                }
                // Output word k = bits [32k - sb, 32k - sb + 32) of the tile string = the window words k-1 and k funnelled
                // by sb.  This round holds the tile words [wstart, wstart + kWinWords): it can form the output words
                // k = (wstart ? wstart + 1 : 0) .. wstart + kWinWords - 1 (rounds overlap by one word for that).  Wave c
                // forms the words whose window word k lies in its own rows; a word at the start of a part also needs the
                // last word of the neighbouring part, so a tile that leaves the first part ends with a meeting.
                const uint32_t k_lo = wstart ? wstart + 1 : 0u;
                const uint32_t k_hi = nwords < wstart + kWinWords ? nwords : wstart + kWinWords;  // exclusive
                const uint32_t nstore = tail_shared ? nwords - 1 : nwords;  // the shared last word leaves as a carry
                {
                    const uint32_t part = 2u - chan;  // window words [2048 part, 2048 part + 2048) of this round
                    uint32_t a = wstart + part * kRegionWords, b = a + kRegionWords;
                    a = a < k_lo ? k_lo : a;
                    b = b > k_hi ? k_hi : b;
                    for (uint32_t k = a + lane; k < b; k += 64) {
                        if (k == 0) continue;  // the frame-facing first word: below
                        const uint32_t i = k - wstart;
                        const uint32_t hi = *win_word(tl, i - 1);
                        const uint32_t lo = (k < twords) ? *win_word(tl, i) : 0u;
                        const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, sb);
                        if (k < nstore) {
                            if (fits) outw[w0 + k] = __builtin_bswap32(v);
                        } else {  // k == nwords - 1, shared with the next tile: it depends on no other tile (k > 0)
                            granule_store(rec + (size_t)tile * kRecGranules + 4, ((unsigned long long)epoch << 48) | v);
                        }
                    }
                }
                if (wstart == 0 && chan == 2) {
                    // word 0: the bits of the previous tile in it arrive as that tile's carry granule
                    uint32_t v0 = twords ? (*win_word(tl, 0) >> sb) : 0u;
                    if (sb != 0) {
                        const unsigned long long gc = wait_granule(rec + (size_t)(tile - 1) * kRecGranules + 4, epoch, sp.status, spin_limit);
                        v0 |= (uint32_t)gc;
                    }
                    if (lane == 0) {
                        if (nstore > 0) {
                            if (fits) outw[w0] = __builtin_bswap32(v0);
                        } else if (nwords == 1) {  // the whole tile inside one shared word: carried on, with what came in
                            granule_store(rec + (size_t)tile * kRecGranules + 4, ((unsigned long long)epoch << 48) | v0);
                        }
                    }
                }
                if (k_hi >= nwords) {
                    // a tile that reached beyond the first part of the window: nobody may rewrite rows that a neighbour
                    // still reads
                    if (nwords > kRegionWords) ts.meet(lane);
                    break;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------
// launcher
// ----------------------------------------------------------------------------
uint32_t tile_grid(const Geom& g, uint32_t n_frames, uint32_t max_wgs) {
    const uint64_t tiles = (uint64_t)g.tiles * n_frames;
    uint64_t wgs = (tiles + kTeams - 1) / kTeams;
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    return (uint32_t)wgs;
}
size_t tile_ovf_words(uint32_t wgs) { return (size_t)wgs * kTeams * 3 * kSlotWordsFull * 64; }
size_t tile_rec_granules(const Geom& g, uint32_t n_frames) { return (size_t)g.tiles * n_frames * kRecGranules; }

hipError_t launch_encode_tile(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const TileParams& tp, uint32_t wgs,
                              hipStream_t s) {
    if (g.flags & 2u)  // MI355_F_STANDARD
        hipLaunchKernelGGL((k_encode_tile<1>), dim3(wgs), dim3(kTileThreads), 0, s, g, n_frames, rgb, tp);
    else
        hipLaunchKernelGGL((k_encode_tile<0>), dim3(wgs), dim3(kTileThreads), 0, s, g, n_frames, rgb, tp);
    return hipGetLastError();
}

}  // namespace mi355
