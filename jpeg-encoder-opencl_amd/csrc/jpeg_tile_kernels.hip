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
//   the three waves meet; the luma wave (more symbols to code, ties to check in its colour conversion: the longest
//   pass) goes straight on to its next tile, and the two chroma waves merge: both scan the 64 block totals, Cb
//   appends its strings to the tile's bit window in LDS (the chroma waves' zig-zag row buffers, dead by then), Cr
//   appends its own AND the luma strings (two independent append chains per lane), while Cb finds the tile's bit
//   offset in the frame by decoupled look-back over the earlier tiles (loads issued before its append, read after);
//   the two meet, and the window leaves for memory shifted into place (one v_alignbit per word), coalesced; the
//   word a tile shares with its successor travels as a "carry" granule, so no output word is written by two teams
//   and nothing is zeroed beforehand.
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
// one workgroup (co-resident by construction) and meet at counters in LDS.  Every spin is bounded and gives up with
// MI355_E_INTERNAL.  Cross-workgroup data are 8-byte {epoch, payload} granules written by one sc1 store and polled
// with sc1 loads (per-XCD L2s are not coherent; cdna_hip_programming.md Guideline 16, form R2): the epoch changes
// with every launch, so stale records of earlier launches never match.  An sc1 load costs 1-1.5 us under load
// (MI355X_MICROARCH.md, handoff-1to1): every granule a tile needs is requested well before it is read.
#include <type_traits>

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
constexpr uint32_t kWinWords = 2 * kRegionWords; // the team's bit window: the row buffers of the Cb and the Cr wave
constexpr uint32_t kWinStep = kWinWords - 1;     // rounds of an oversized tile overlap by one word (see the write-out)
constexpr uint32_t kSpinLimit = 1u << 20;        // polls (>= ~0.5 us each) before a wait on another team gives up
constexpr uint32_t kRecGranules = 4;             // per tile: [0..2] last DC per channel, [3] carry word; the scan granules are an
                                                 // array of their own (8 bytes per tile: a look-back round reads 2 KB, not 256 lines)

struct TeamLds {
    uint32_t rows[3][kRowWords];                 // zig-zag rows [position][unit] per wave; [1], [2] later the bit window
    alignas(16) uint32_t slot_y[(kRowsY + 1) * 64];      // AC strings [word][lane] + dump row (after the walk: the DC symbols)
    alignas(16) uint32_t slot_c[2][(kRowsC + 1) * 64];
    uint32_t mask[3][2][64];                     // non-zero masks (lo, hi) per wave
    uint32_t ubits[3][64];                       // bits of every unit (DC symbol + AC string), per wave
    uint32_t arrive;                             // meeting counter of the three waves (monotonic)
    uint32_t arrive2;                            // meeting counter of the two chroma waves
    uint32_t merged;                             // +1 by Cb and by Cr per tile, once they read nothing more of what the Y wave wrote
    uint32_t next_ticket;                        // the team's next tile (told at the meeting of a tile)
    uint32_t first_ticket[2];                    // the team's first tile of a frame (alternating: a frame may hold no tile for the team)
    uint32_t first_frame[2];                     // and that frame (0xFFFFFFFF: no work left anywhere)
    uint32_t excl_lo, excl_hi;                   // the tile's bit offset in its frame
    uint32_t err[2];                             // category error seen by a wave of the team in this tile; and
    uint32_t big[2];                             // a wave of the team holds a string in its overflow area.  Indexed by the parity
                                                 // of the team's tile count: the waves run up to a tile apart (see the kernel)
    uint32_t err_before;                         // an error in an earlier tile of the frame (from the look-back)
    uint32_t ovm_y[2];                           // which luma strings are in the Y wave's overflow area
    uint32_t pend[6];                            // Cb wave only: the first output word of its previous tile, still waiting for that
                                                 // tile's carry-in {1 | fits << 1, frame, tile, word index lo, hi, bits}
};

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Spin bookkeeping shared by every wait of a wave.  A wait gives up after `limit` polls, sets status bit 2
// (MI355_E_INTERNAL) and zeroes the limit, after which the wave no longer waits anywhere; every 1024 polls a waiting
// wave also looks at the status word, so that the first give-up anywhere makes a broken launch drain in milliseconds
// instead of timing out wait after wait.
struct Spin {
    uint32_t* status;
    uint32_t limit;
    __device__ __forceinline__ bool give_up(uint32_t spins, uint32_t lane) {  // true: stop waiting
        if (spins >= limit) {
            if (limit && lane == 0) atomicOr(status, 4u);
            limit = 0;
            return true;
        }
        if ((spins & 1023u) == 1023u && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u)) {
            limit = 0;
            return true;
        }
        return false;
    }
};

// Waves of a team meet at a counter in LDS that only ever grows; `phase` is this wave's count of the meetings so far
// (x the number of parties).  LDS operations of one wave execute in order, so everything the wave wrote to LDS before
// its arrival is visible to whoever sees the counter reach the phase.  (The partners run the same loop, so the bound
// is only ever reached through a defect.)
struct Meet {
    uint32_t* cnt;
    uint32_t parties;
    uint32_t phase = 0;
    // arrive without waiting (the partner waits; this wave needs nothing from it)
    __device__ __forceinline__ void signal(uint32_t lane) {
        phase += parties;
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void meet(uint32_t lane, Spin& sp) {
        phase += parties;
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (uint32_t spins = 0; (int)(__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - phase) < 0; ++spins) {
            if (sp.give_up(spins >> 2, lane)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
};
// one-way: wait until *cnt has reached `target`
__device__ __forceinline__ void wait_counter(uint32_t* cnt, uint32_t target, uint32_t lane, Spin& sp) {
    for (uint32_t spins = 0; (int)(__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0; ++spins) {
        if (sp.give_up(spins >> 2, lane)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// Waits until the granule at p (wave-uniform address) carries `epoch`; `v` = a copy requested earlier (tried first).
// On give-up returns a granule with payload 0.
__device__ __forceinline__ unsigned long long wait_granule(const unsigned long long* p, unsigned long long v, uint32_t epoch,
                                                           uint32_t lane, Spin& sp) {
    for (uint32_t spins = 0; (uint32_t)(v >> 48) != epoch; ++spins) {
        if (sp.give_up(spins, lane)) {
            v = (unsigned long long)epoch << 48;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
        v = granule_load(p);
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

// Decoupled look-back over the scan granules of the earlier tiles of the frame.  All 64 lanes take part, FOUR
// predecessors each per round (a round covers 256 tiles: with every tile of a frame in flight at the same moment the
// prefixes travel 256 tiles per polling round).  request() issues the loads of a round; finish() returns the
// exclusive bit offset of `tile` = sum of the totals of tiles 0 .. tile-1, polling where a granule was not there yet.
// `err` collects the error flags met on the way.
struct LookBack {
    unsigned long long v[4];
    __device__ __forceinline__ void request(const unsigned long long* scan /* the frame's scan granules */, int top, uint32_t lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // slot q = 4 * lane + i is predecessor top - q: nearer tiles in lower slots
            const int idx = top - (int)(4 * lane + i);
            v[i] = idx >= 0 ? granule_load(scan + idx) : 0ull;
        }
    }
    __device__ __forceinline__ unsigned long long finish(const unsigned long long* scan, uint32_t tile,
                                                         uint32_t epoch, uint32_t lane, Spin& sp, bool& err) {
        unsigned long long excl = 0;
        int top = (int)tile - 1;  // nearest predecessor not yet accounted for; its round is already requested
        while (top >= 0) {
            uint32_t first = 256;  // lowest slot holding a prefix
            const uint32_t nvalid = top + 1 < 256 ? (uint32_t)(top + 1) : 256u;
            for (uint32_t spins = 0;; ++spins) {
                uint32_t pubm = 0, prem = 0;  // this lane's slots: published / prefix
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool pub = top - (int)(4 * lane + i) >= 0 && (uint32_t)(v[i] >> 48) == epoch;
                    if (pub) pubm |= 1u << i;
                    if (pub && ((uint32_t)(v[i] >> 46) & 3u) == 2u) prem |= 1u << i;
                }
                // lanes hold increasing slots: the first lane with a prefix wins
                const uint32_t myfirst = prem ? 4 * lane + (uint32_t)__builtin_ctz(prem) : 256u;
                const unsigned long long bp = __ballot(prem != 0);
                first = bp ? (uint32_t)__builtin_amdgcn_readlane((int)myfirst, (int)__builtin_ctzll(bp)) : 256u;
                const uint32_t upto = first < nvalid - 1 ? first : nvalid - 1;  // last slot that must have published
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * lane + i <= upto && !((pubm >> i) & 1u)) ok = false;
                if (__all(ok)) break;
                if (sp.give_up(spins, lane)) return excl;
                __builtin_amdgcn_s_sleep(4);
                request(scan, top, lane);
            }
            // totals of the slots nearer than the first prefix (each < 2^19: a round's sum fits 32 bits), the prefix itself
            uint32_t agg = 0, e = 0;
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
                e |= (uint32_t)(p >> 45) & 1u;
            }
            err = err || wave_any(e != 0);
            if (first < 256) break;
            top -= 256;
            if (top >= 0) request(scan, top, lane);
        }
        return excl;
    }
};

// ---- the usual tile: its whole bit string fits the Cb wave's row buffer (2048 words, contiguous, zeroed) ----
// ORs one unit -- the DC symbol `sym` (left-aligned | length), then `nw` words of AC bits from `src` (LDS, [word][lane],
// zeros below the last valid bit) -- into `win` at bit `pos`.  Every word of the AC string is split between two output
// words by the same shift, so an output word is one funnel shift of two neighbouring string words: no serial bit
// bookkeeping.  Words are OR-ed (the window starts out zero; units share words at their ends), indices are clamped
// to the buffer (a lane past its own string ORs zeros), and the loop runs the wave's longest string, branch-free.
struct LeanStream {
    uint32_t w, n, acc, nw;
    __device__ __forceinline__ void start(uint32_t* win, uint32_t pos, uint32_t sym, uint32_t aclen) {
        w = pos >> 5, n = pos & 31u;
        const uint32_t ml = sym & ~31u, n2 = n + (sym & 31u);
        atomicOr(&win[w], ml >> n);
        const bool adv = n2 >= 32u;
        acc = adv ? __builtin_amdgcn_alignbit(ml, 0u, n) : 0u;  // ml << (32 - n), 0 when n == 0
        w += adv ? 1u : 0u;
        n = n2 & 31u;
        nw = (aclen + 31u) >> 5;
    }
    __device__ __forceinline__ void step(uint32_t* win, const uint32_t* src, uint32_t j, uint32_t src_rows) {
        const uint32_t raw = src[(j < src_rows ? j : src_rows - 1u) * 64u];
        const uint32_t cur = j < nw ? raw : 0u;
        const uint32_t i = w + j;
        atomicOr(&win[i < kRegionWords - 1u ? i : kRegionWords - 1u], acc | (cur >> n));
        acc = __builtin_amdgcn_alignbit(cur, 0u, n);
    }
    __device__ __forceinline__ void flush(uint32_t* win, uint32_t trips) {
        const uint32_t i = w + trips;
        atomicOr(&win[i < kRegionWords - 1u ? i : kRegionWords - 1u], acc);
    }
};

// Address of word i of the team's bit window: it runs through the row buffers of the Cb wave, then the Cr wave,
// 2048 words each, skipping the sentinel rows.
__device__ __forceinline__ uint32_t* win_word(TeamLds& tl, uint32_t i) { return &tl.rows[1u + (i >> 11)][i & 2047u]; }

// Appends bits to the window (big-endian words).  A unit's region is contiguous and only its first and last word can
// be shared with the neighbouring units (other lanes, the other wave): those are OR-ed, the others stored.  Words
// outside [wstart, wstart + kWinWords) belong to another round of an oversized tile and are skipped.
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

// Diagnostic build (make STAMPS=1): s_memtime stamps at the phase boundaries of a tile, summed per wave and written
// to sp.stamps ([wave][16]: 0 loop head + ticket, 1 samples, 2 map + quantise, 3 walk (Y: incl. the wait for the
// mergers), 4 DC wait + symbol + totals, 5 meeting, 6 scan + append, 7 look-back (Cb wave), 8 meeting of the chroma
// waves, 9 write-out, 10 first word / carry wait (Cb wave); 13/15 wall ticks, 14 tiles).  Never in the shipped kernel.
#ifdef MI355_STAMPS
#define TSTAMP(i)                                                                    \
    do {                                                                             \
        unsigned long long _t;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                           \
        stamp_sum[i] += _t - stamp_prev;                                             \
        stamp_prev = _t;                                                             \
    } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

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
    __shared__ uint32_t s_first;

    const ScreenParams& sp = tp.sp;
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    // lane / n / gq are made opaque at the phase boundaries of a tile (OPAQUE_LANE): everything derived from them -- a
    // few dozen LDS addresses and offsets -- is then recomputed per phase (a handful of instructions) instead of being
    // hoisted out of the tile loop into registers that live for the whole kernel and end up spilled
    uint32_t lane = tid & 63, n = lane & 15, gq = lane >> 4;
#define OPAQUE_LANE() do { asm volatile("" : "+v"(lane)); n = lane & 15u; gq = lane >> 4; } while (0)
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
    if (lane == 0 && chan == 0) tl.arrive = 0, tl.arrive2 = 0, tl.merged = 0, tl.err[0] = tl.err[1] = 0, tl.big[0] = tl.big[1] = 0, tl.pend[0] = 0;
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
    uint32_t fpar = 0;
    Spin spin{sp.status, kSpinLimit};
    Meet m3{&tl.arrive, 3u};   // Y, Cb, Cr
    Meet m2{&tl.arrive2, 2u};  // Cb, Cr
    uint32_t tiles_done = 0;   // tiles this team has been through
    bool walk_general = false;  // the last tile had a symbol-table miss (walk_nonzeros)
#ifdef MI355_STAMPS
    unsigned long long stamp_sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev, wave_t0, wave_t1, ntiles = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wave_t0)::"memory");  // 100 MHz wall clock
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    // The first output word of a tile also holds the last bits of the previous tile, which arrive as that tile's carry
    // granule -- published about when this tile asks for it.  Instead of waiting, the Cb wave parks the word (tl.pend) and
    // completes it one tile later: `gp` = the carry granule, requested at the start of that later merge.
    auto complete_parked_word = [&](unsigned long long gp) {
        const uint32_t pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[0]);
        if (!(pf & 1u)) return;
        const uint32_t pframe = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[1]);
        const uint32_t ptile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[2]);
        gp = wait_granule(tp.rec + ((size_t)pframe * g.tiles + (ptile - 1)) * kRecGranules + 3, gp, epoch, lane, spin);
        if (lane == 0) {
            if (pf & 2u) {
                const unsigned long long pw = ((unsigned long long)tl.pend[4] << 32) | tl.pend[3];
                reinterpret_cast<uint32_t*>(tp.out + (size_t)pframe * tp.out_stride)[pw] = __builtin_bswap32(tl.pend[5] | (uint32_t)gp);
            }
            tl.pend[0] = 0;
        }
    };
    auto parked_word_granule = [&]() -> unsigned long long {  // request for complete_parked_word
        const uint32_t pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[0]);
        if (!(pf & 1u)) return 0ull;
        const uint32_t pframe = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[1]);
        const uint32_t ptile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.pend[2]);
        return granule_load(tp.rec + ((size_t)pframe * g.tiles + (ptile - 1)) * kRecGranules + 3);
    };
    // Frames: a team first works through its own list (group, group + G, ...), then HELPS: it looks for frames whose
    // ticket counter has not run out and draws from them.  Without that the launch ends when the slowest frame's teams
    // end -- a 128-frame launch lasted 12 % longer than its average wave.  (Any team may take any tile: tickets and
    // look-back do not care who holds them.)
    uint32_t own_next = group, helped = 0, predrawn = 0xFFFFFFFFu;
    if (wg_draw) {  // one frame: the workgroup draws its teams' first tickets with one atomic (see above)
        if (tid == 0) s_first = atomicAdd(&tp.ticket[0], kTeams);
        __syncthreads();
        predrawn = s_first + team;
    }
    for (;;) {
        // (the frame's pointers -- pixels, hand-off records, scan granules -- are formed from an opaque copy of `frame` in
        // the phase that uses them: as loop invariants of the tile loop they end up in VGPR pairs, spilled to scratch, and
        // every reload is an s_waitcnt vmcnt(0) that also waits for the sc1 loads in flight)
        // Next frame and first ticket: chosen by the team's first wave, told at a meeting.
        if (chan == 0) {
            uint32_t nf = 0xFFFFFFFFu, nt = 0;
            if (lane == 0) {
                while (own_next < n_frames) {
                    const uint32_t cand = own_next;
                    own_next += groups;
                    nt = predrawn != 0xFFFFFFFFu ? predrawn : atomicAdd(&tp.ticket[cand], 1u);
                    predrawn = 0xFFFFFFFFu;
                    if (nt < g.tiles) {
                        nf = cand;
                        break;
                    }
                }
                while (nf == 0xFFFFFFFFu && helped + 1 < n_frames) {  // every other frame once, starting behind the own one
                    ++helped;
                    const uint32_t cand = (group + helped) % n_frames;
                    if (__hip_atomic_load(&tp.ticket[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.tiles) {
                        nt = atomicAdd(&tp.ticket[cand], 1u);
                        if (nt < g.tiles) nf = cand;
                        else continue;
                        --helped;  // look at this frame again when these tiles are done: it may still have some
                    }
                }
                tl.first_ticket[fpar] = nt;
                tl.first_frame[fpar] = nf;
            }
        }
        m3.meet(lane, spin);
        const uint32_t frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.first_frame[fpar]);
        uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.first_ticket[fpar]);
        fpar ^= 1u;
        if (frame == 0xFFFFFFFFu) break;
        while (t < g.tiles) {
            OPAQUE_LANE();
            uint32_t frame_s = frame;
            asm volatile("" : "+s"(frame_s));
            const uint8_t* const f = rgb + (size_t)frame_s * g.frame_stride;
            const uint32_t tile = t;
            const uint32_t par = tiles_done & 1u;  // slot of the team's per-tile flags
            const uint32_t nblk = g.N - tile * 64 < 64u ? g.N - tile * 64 : 64u;  // active blocks of the tile
            const bool active = lane < nblk;
            uint32_t next_ticket = 0;
            if (chan == 0 && lane == 0) next_ticket = atomicAdd(&tp.ticket[frame], 1u);  // told at the meeting below
#ifdef MI355_STAMPS
            ++ntiles;
#endif
            v4i B[4];
            uint32_t dcsum = 0;  // sample sum of the block whose coefficient 0 this lane will form

            TSTAMP(0);
            // ---------------- samples: the wave's 64 blocks as four B operands ----------------
#include "jpeg_transform_core.inc"
            // ---------------- walk phase: lane = block ----------------
            OPAQUE_LANE();
            if constexpr (!STD) {
                // Units with a coefficient the screen could not decide: the exact chain is the arbiter (scratch: this
                // wave's string slot; for the Y wave that needs the mergers to be done with the previous tile's strings)
                const bool undecided = active && (s_mlo[lane] & 1u) != 0;
                unsigned long long todo = __ballot(undecided);
                if (todo && chan == 0) wait_counter(&tl.merged, 2u * tiles_done, lane, spin);
                while (todo) {  // wave-uniform: one unit at a time, the whole wave on it
                    const uint32_t ul = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    if (lane == 0) atomicAdd(&sp.stats[1], 1ull);
                    const uint32_t ub = tile * 64 + ul, uby = ub / g.nbx, ubx = ub - uby * g.nbx;
                    exact_unit_wave(rgb + (size_t)frame * g.frame_stride, g, chan, ubx, uby, sp.qd, reinterpret_cast<double*>(slot), tb16 + row_unit_off(ul),
                                    &s_mlo[ul], &s_mhi[ul], lane);
                }
            }
            i16a* const row16 = tb16 + row_unit_off(lane);
            unsigned long long mask = ((unsigned long long)s_mhi[lane] << 32 | s_mlo[lane]) & ~1ull;
            const int dc = (int)row16[0];
            // the tile's last DC of this channel, for the successor's first DC difference; and the predecessor's, asked for
            // now and read after the walk
            uint32_t frame_d = frame;
            asm volatile("" : "+s"(frame_d));
            unsigned long long* const rec = tp.rec + (size_t)frame_d * g.tiles * kRecGranules;  // this frame's records
            {
                const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane(dc, (int)(nblk - 1)) & 0xffffu;
                if (lane == 0) granule_store(rec + (size_t)tile * kRecGranules + chan, ((unsigned long long)epoch << 48) | dl);
            }
            const bool has_pred = tile > 0 && !restart;
            unsigned long long gdc = 0;
            if (has_pred) gdc = granule_load(rec + (size_t)(tile - 1) * kRecGranules + chan);
            // the Y wave runs ahead of the mergers: its slot, totals and ticket word may be rewritten only when both
            // chroma waves are done reading the previous tile's
            if (chan == 0) wait_counter(&tl.merged, 2u * tiles_done, lane, spin);

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
            TSTAMP(3);

            // ---------------- DC symbol, unit totals ----------------
            OPAQUE_LANE();
            int pred = __shfl_up(dc, 1);
            if (has_pred) {
                uint32_t frame_w = frame;
                asm volatile("" : "+s"(frame_w));
                gdc = wait_granule(tp.rec + ((size_t)frame_w * g.tiles + (tile - 1)) * kRecGranules + chan, gdc, epoch, lane, spin);
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
                    atomicOr(&tl.err[par], 1u);
                }
            }
            tl.ubits[chan][lane] = active ? (dcsym & 31u) + aclen : 0u;
            if (ovm && lane == 0) atomicOr(&tl.big[par], 1u);
            if (chan == 0) {
                tl.slot_y[kRowsY * 64 + lane] = dcsym;  // the dump row is dead after the walk
                if (lane == 0) {
                    tl.ovm_y[0] = (uint32_t)ovm, tl.ovm_y[1] = (uint32_t)(ovm >> 32);
                    tl.next_ticket = next_ticket;
                }
            } else {
                // this wave's rows are dead: its part of the window starts out zero
                uint4* const z = reinterpret_cast<uint4*>(s_rows);
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            TSTAMP(4);
            m3.meet(lane, spin);  // ---- the meeting: strings, totals, zeroed window, next ticket
            TSTAMP(5);
            ++tiles_done;
            if (chan == 0) {  // the Y wave is done with this tile
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_ticket);
                continue;
            }
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl.next_ticket);

            // ================= the two chroma waves merge the tile's 192 strings =================
            OPAQUE_LANE();
            uint32_t frame_m = frame;
            asm volatile("" : "+s"(frame_m));
            unsigned long long* const recm = tp.rec + (size_t)frame_m * g.tiles * kRecGranules;
            unsigned long long* const scan = tp.rec + (size_t)n_frames * g.tiles * kRecGranules + (size_t)frame_m * g.tiles;
            const uint32_t u0 = tl.ubits[0][lane], u1 = tl.ubits[1][lane], u2 = tl.ubits[2][lane];
            const uint32_t blk = u0 + u1 + u2;
            const uint32_t incl = wave_incl_scan(blk, lane);
            uint32_t tbits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t fill = restart ? ((8u - (tbits & 7u)) & 7u) : 0u;  // the interval ends on a byte boundary, filled with 1s
            tbits += fill;
            const uint32_t posy = incl - blk;                                 // first bit of this lane's block, tile-relative
            const uint32_t pos0 = posy + u0 + (chan == 2 ? u1 : 0u);          // of this lane's own unit
            const bool last_tile = tile + 1 == g.tiles;
            const uint32_t twords = (tbits + 31) >> 5;  // words of the tile-relative bit string
            bool tile_err = false;
            LookBack lb;
            unsigned long long gcarry = 0;
            if (chan == 1) {
                tile_err = tl.err[par] != 0;
                if (lane == 0) {
                    if (tile > 0) granule_store(scan + tile, scan_granule(epoch, 1u, tile_err, tbits));
                    __hip_atomic_fetch_add(&tl.merged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // done with what the Y wave wrote
                }
                if (tile > 0) lb.request(scan, (int)tile - 1, lane);  // asked for now, read after the append
                gcarry = parked_word_granule();                        // for the word parked a tile ago
            }
            // Cr appends the luma strings along with its own.  String words come from the LDS slots; only a wave that holds
            // an oversized string (rare) takes the loop that can also read the overflow areas in memory -- kept apart so
            // that the usual loop has no flat loads (they would wait for every vector-memory operation in flight, the
            // look-back requests above included).
            const uint32_t ysym = chan == 2 ? tl.slot_y[kRowsY * 64 + lane] : 0u;
            const uint32_t ylen = chan == 2 && active ? u0 - (ysym & 31u) : 0u;
            const bool ovy = chan == 2 && (((lane < 32 ? tl.ovm_y[0] : tl.ovm_y[1]) >> (lane & 31)) & 1u) != 0;
            const bool ov = (ovm >> lane) & 1ull;
            const bool any_ov = tl.big[par] != 0;  // the same answer in both chroma waves
            // The flags of the NEXT tile are cleared by the Cr wave before it lets go of this one: the last tile that used that
            // slot is behind every wave, and nobody can set it for the next one before Cr's signals below (the Y wave waits
            // for `merged`, the Cb wave for `arrive2`).
            if (chan == 2 && lane == 0) tl.err[par ^ 1u] = 0, tl.big[par ^ 1u] = 0;
            const uint32_t* const ovf_y = tp.ovf + (size_t)(gteam * 3) * kSlotWordsFull * 64 + lane;
            unsigned long long excl = 0;
            bool err_before = false;
            if (twords < kRegionWords && !any_ov) {
                // ---------- the usual tile: one round, the whole bit string in the Cb wave's rows (contiguous) ----------
                uint32_t* const win = tl.rows[1];
                {
                    LeanStream so, sy;
                    so.start(win, active ? pos0 : 0u, active ? dcsym : 0u, active ? aclen : 0u);
                    uint32_t trips = so.nw;
                    if (chan == 2) {
                        sy.start(win, active ? posy : 0u, active ? ysym : 0u, ylen);
                        trips = trips > sy.nw ? trips : sy.nw;
                    }
                    trips = wave_max(trips);
                    if (chan == 2) {  // two independent streams per lane: its own string and the luma string of its block
                        for (uint32_t j = 0; j < trips; ++j) {
                            so.step(win, slot + lane, j, kRowsC);
                            sy.step(win, tl.slot_y + lane, j, kRowsY);
                        }
                        sy.flush(win, trips);
                        if (fill && lane == nblk - 1) {  // restart interval: 1s up to the byte boundary, after the block's last unit
                            const uint32_t pe = posy + blk;
                            atomicOr(&win[pe >> 5], ((1u << fill) - 1u) << (32u - (pe & 31u) - fill));
                        }
                    } else {
                        for (uint32_t j = 0; j < trips; ++j) so.step(win, slot + lane, j, kRowsC);
                    }
                    so.flush(win, trips);
                }
                TSTAMP(6);
                if (chan == 2) {  // Cr is done: Cb waits for this signal, nothing here waits for Cb
                    m2.signal(lane);
                    if (lane == 0) __hip_atomic_fetch_add(&tl.merged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                // the tile's place in the frame (the predecessors published their totals after THEIR meeting)
                if (tile > 0) {
                    excl = lb.finish(scan, tile, epoch, lane, spin, err_before);
                    if (lane == 0) granule_store(scan + tile, scan_granule(epoch, 2u, tile_err || err_before, excl + tbits));
                } else if (lane == 0) {
                    granule_store(scan, scan_granule(epoch, 2u, tile_err, tbits));
                }
                TSTAMP(7);
                m2.meet(lane, spin);  // Cr's strings are in
                TSTAMP(8);
                const unsigned long long endbit = excl + tbits;
                const bool room = ((endbit + 31) >> 5) * 4 <= tp.out_stride;
                const bool fits = room && tp.out != nullptr;
                const uint32_t sb = (uint32_t)(excl & 31);                  // bit offset inside the tile's first output word
                const uint32_t nwords = (sb + tbits + 31) >> 5;             // output words the tile touches (<= 2048)
                const bool tail_shared = !last_tile && ((sb + tbits) & 31u) != 0;  // the last word continues in the next tile
                const uint32_t nstore = tail_shared ? nwords - 1 : nwords;  // the shared last word leaves as a carry
                uint32_t* const outw = reinterpret_cast<uint32_t*>(tp.out + (size_t)frame_m * tp.out_stride) + (excl >> 5);
                if (lane == 0) {
                    if (!room && tp.out) atomicOr(sp.status, 2u);  // MI355_E_CAPACITY
                    tp.tile_off[(size_t)frame_m * (g.tiles + 1) + tile] = excl;
                    if (last_tile) {
                        tp.tile_off[(size_t)frame_m * (g.tiles + 1) + g.tiles] = endbit;
                        // a frame with an error anywhere says so in its bit count; the other frames of the batch are good
                        tp.frame_bits[frame_m] = (tile_err || err_before || (!room && tp.out)) ? ~0ull : endbit;
                    }
                }
                // output word k = the window words k-1 and k funnelled by sb (the window is zero beyond the string); word 0 and
                // the shared last word are dealt with below
                for (uint32_t base = 1; base < nstore; base += 256) {  // four words per lane in flight
                    uint32_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = base + 64u * u + lane, kk = k < kRegionWords - 1u ? k : kRegionWords - 1u;
                        v[u] = __builtin_amdgcn_alignbit(win[kk - 1], win[kk], sb);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = base + 64u * u + lane;
                        if (k < nstore && fits) outw[k] = __builtin_bswap32(v[u]);
                    }
                }
                if (tail_shared && nwords > 1) {  // the shared last word depends on no other tile: on its way before this tile waits
                    const uint32_t k = nwords - 1;
                    const uint32_t v = __builtin_amdgcn_alignbit(win[k - 1], win[k], sb);
                    if (lane == 0) granule_store(recm + (size_t)tile * kRecGranules + 3, ((unsigned long long)epoch << 48) | v);
                }
                TSTAMP(9);
                {
                    complete_parked_word(gcarry);
                    // word 0 holds the last bits of the previous tile too (sb of them)
                    uint32_t v0 = win[0] >> sb;
                    if (nstore == 0) {  // nwords == 1: the whole tile inside one shared word: carried on, with what came in
                        if (sb != 0) v0 |= (uint32_t)wait_granule(recm + (size_t)(tile - 1) * kRecGranules + 3, 0ull, epoch, lane, spin);
                        if (lane == 0) granule_store(recm + (size_t)tile * kRecGranules + 3, ((unsigned long long)epoch << 48) | v0);
                    } else if (lane == 0) {
                        if (sb == 0) {
                            if (fits) outw[0] = __builtin_bswap32(v0);
                        } else {  // parked until the next tile
                            const unsigned long long w0 = excl >> 5;
                            tl.pend[1] = frame_m, tl.pend[2] = tile, tl.pend[3] = (uint32_t)w0, tl.pend[4] = (uint32_t)(w0 >> 32);
                            tl.pend[5] = v0;
                            tl.pend[0] = 1u | (fits ? 2u : 0u);
                        }
                    }
                }
                TSTAMP(10);
                continue;
            }
            // ---------- a tile beyond 2047 words, or one with strings in the overflow areas: the general form ----------
            for (uint32_t wstart = 0;; wstart += kWinStep) {  // one round unless the tile is huge
                if (wstart) {  // later rounds: both are done with the window, then it starts out zero again
                    m2.meet(lane, spin);
                    uint4* const z = reinterpret_cast<uint4*>(s_rows);
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                    m2.meet(lane, spin);
                }
                if (active) {
                    WindowAppender ap{tl, wstart, pos0 >> 5, 0u, pos0 & 31u, true};
                    ap.put(dcsym & ~31u, dcsym & 31u);
                    auto own_word = [&](uint32_t done, bool slow) -> uint32_t {
                        if (slow && ov) return ovf[(done >> 5) * 64u + lane];
                        return slot[(done >> 5) * 64u + lane];
                    };
                    auto luma_word = [&](uint32_t done, bool slow) -> uint32_t {
                        if (slow && ovy) return ovf_y[(done >> 5) * 64u];
                        return tl.slot_y[(done >> 5) * 64u + lane];
                    };
                    auto append_all = [&](auto slow_c) {
                        constexpr bool slow = decltype(slow_c)::value;
                        if (chan == 2) {  // two independent chains per lane
                            WindowAppender ay{tl, wstart, posy >> 5, 0u, posy & 31u, true};
                            ay.put(ysym & ~31u, ysym & 31u);
                            const uint32_t mx = aclen > ylen ? aclen : ylen;
                            for (uint32_t done = 0; done < mx; done += 32) {
                                if (done < aclen) ap.put(own_word(done, slow), aclen - done < 32u ? aclen - done : 32u);
                                if (done < ylen) ay.put(luma_word(done, slow), ylen - done < 32u ? ylen - done : 32u);
                            }
                            ay.finish();
                            if (fill && lane == nblk - 1) ap.put(0xFFFFFFFFu << ((32u - fill) & 31u), fill);
                        } else {
                            for (uint32_t done = 0; done < aclen; done += 32)
                                ap.put(own_word(done, slow), aclen - done < 32u ? aclen - done : 32u);
                        }
                    };
                    if (any_ov) append_all(std::true_type{});
                    else append_all(std::false_type{});
                    ap.finish();
                }
                TSTAMP(6);
                if (wstart == 0 && chan == 1) {
                    // the tile's place in the frame (the predecessors published their totals after THEIR meeting)
                    if (tile > 0) {
                        excl = lb.finish(scan, tile, epoch, lane, spin, err_before);
                        if (lane == 0)
                            granule_store(scan + tile, scan_granule(epoch, 2u, tile_err || err_before, excl + tbits));
                    } else if (lane == 0) {
                        granule_store(scan, scan_granule(epoch, 2u, tile_err, tbits));
                    }
                    if (lane == 0) {
                        tl.excl_lo = (uint32_t)excl, tl.excl_hi = (uint32_t)(excl >> 32);
                        tl.err_before = err_before ? 1u : 0u;
                    }
                }
                TSTAMP(7);
                m2.meet(lane, spin);  // ---- the window is complete, the offset known
                TSTAMP(8);
                if (wstart == 0 && chan == 2) excl = ((unsigned long long)tl.excl_hi << 32) | tl.excl_lo;
                const unsigned long long endbit = excl + tbits;
                const bool room = ((endbit + 31) >> 5) * 4 <= tp.out_stride;
                const bool fits = room && tp.out != nullptr;
                const uint32_t sb = (uint32_t)(excl & 31);                  // bit offset inside the tile's first output word
                const unsigned long long w0 = excl >> 5;                    // the tile's first word in the frame's output
                const uint32_t nwords = (sb + tbits + 31) >> 5;             // output words the tile touches
                const bool tail_shared = !last_tile && ((sb + tbits) & 31u) != 0;  // the last word continues in the next tile
                uint32_t* const outw = reinterpret_cast<uint32_t*>(tp.out + (size_t)frame_m * tp.out_stride);
                if (wstart == 0 && chan == 1 && lane == 0) {
                    if (!room && tp.out) atomicOr(sp.status, 2u);  // MI355_E_CAPACITY
                    tp.tile_off[(size_t)frame_m * (g.tiles + 1) + tile] = excl;
                    if (last_tile) {
                        tp.tile_off[(size_t)frame_m * (g.tiles + 1) + g.tiles] = endbit;
                        // a frame with an error anywhere says so in its bit count; the other frames of the batch are good
                        tp.frame_bits[frame_m] = (tile_err || err_before || (!room && tp.out)) ? ~0ull : endbit;
                    }
                }
                // Output word k = bits [32k - sb, 32k - sb + 32) of the tile string = the window words k-1 and k funnelled
                // by sb.  This round holds the tile words [wstart, wstart + kWinWords): it can form the output words
                // k = (wstart ? wstart + 1 : 0) .. wstart + kWinWords - 1 (rounds overlap by one word for that).  Each
                // chroma wave forms the words whose window word k lies in its own rows; the first word of the Cr part also
                // needs the last word of the Cb part, so a tile that reaches the Cr part ends with a meeting.
                const uint32_t k_lo = wstart ? wstart + 1 : 0u;
                const uint32_t k_hi = nwords < wstart + kWinWords ? nwords : wstart + kWinWords;  // exclusive
                const uint32_t nstore = tail_shared ? nwords - 1 : nwords;  // the shared last word leaves as a carry
                {
                    uint32_t a = wstart + (chan - 1u) * kRegionWords, b = a + kRegionWords;
                    a = a < k_lo ? k_lo : a;
                    b = b > k_hi ? k_hi : b;
                    for (uint32_t base = a; base < b; base += 256) {  // four words per lane in flight
                        uint32_t hi[4], lo[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t k = base + 64u * u + lane, i = k - wstart;
                            const bool valid = k < b && k != 0;  // word 0, the frame-facing first word: below
                            // unconditional reads at clamped indices, then selects: predicated reads would be eight
                            // LDS round trips one after the other
                            const uint32_t ih = valid ? i - 1 : 0u, il = valid && k < twords ? i : 0u;
                            const uint32_t rh = *win_word(tl, ih), rl = *win_word(tl, il);
                            hi[u] = valid ? rh : 0u;
                            lo[u] = valid && k < twords ? rl : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t k = base + 64u * u + lane;
                            const uint32_t v = __builtin_amdgcn_alignbit(hi[u], lo[u], sb);
                            if (k < b && k != 0) {
                                if (k < nstore) {
                                    if (fits) outw[w0 + k] = __builtin_bswap32(v);
                                } else {  // k == nwords - 1, shared with the next tile: it depends on no other tile (k > 0)
                                    granule_store(recm + (size_t)tile * kRecGranules + 3, ((unsigned long long)epoch << 48) | v);
                                }
                            }
                        }
                    }
                }
                TSTAMP(9);
                if (wstart == 0 && chan == 1) {
                    complete_parked_word(gcarry);
                    // word 0: the bits of the previous tile in it arrive as that tile's carry granule
                    uint32_t v0 = twords ? (*win_word(tl, 0) >> sb) : 0u;
                    if (sb != 0) v0 |= (uint32_t)wait_granule(recm + (size_t)(tile - 1) * kRecGranules + 3, 0ull, epoch, lane, spin);
                    if (lane == 0) {
                        if (nstore > 0) {
                            if (fits) outw[w0] = __builtin_bswap32(v0);
                        } else if (nwords == 1) {  // the whole tile inside one shared word: carried on, with what came in
                            granule_store(recm + (size_t)tile * kRecGranules + 3, ((unsigned long long)epoch << 48) | v0);
                        }
                    }
                }
                TSTAMP(10);
                if (k_hi >= nwords) {
                    // a round that reached the Cr part: Cb may not rewrite its rows while Cr still reads their last word
                    if (k_hi - wstart > kRegionWords) m2.meet(lane, spin);
                    break;
                }
            }
            // Cr is done with the luma strings
            if (chan == 2 && lane == 0) __hip_atomic_fetch_add(&tl.merged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (chan == 1) complete_parked_word(0ull);  // the last tile's first word
#ifdef MI355_STAMPS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wave_t1)::"memory");
    if (sp.stamps && lane == 0) {
        for (int i = 0; i < 11; ++i) sp.stamps[(size_t)gwave * 16 + i] = stamp_sum[i];
        sp.stamps[(size_t)gwave * 16 + 13] = wave_t0;
        sp.stamps[(size_t)gwave * 16 + 14] = ntiles;
        sp.stamps[(size_t)gwave * 16 + 15] = wave_t1;
    }
#endif
#undef OPAQUE_LANE
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
size_t tile_rec_granules(const Geom& g, uint32_t n_frames) { return (size_t)g.tiles * n_frames * (kRecGranules + 1); }

hipError_t launch_encode_tile(const Geom& g, uint32_t n_frames, const uint8_t* rgb, const TileParams& tp, uint32_t wgs,
                              hipStream_t s) {
    if (g.flags & 2u)  // MI355_F_STANDARD
        hipLaunchKernelGGL((k_encode_tile<1>), dim3(wgs), dim3(kTileThreads), 0, s, g, n_frames, rgb, tp);
    else
        hipLaunchKernelGGL((k_encode_tile<0>), dim3(wgs), dim3(kTileThreads), 0, s, g, n_frames, rgb, tp);
    return hipGetLastError();
}

}  // namespace mi355
