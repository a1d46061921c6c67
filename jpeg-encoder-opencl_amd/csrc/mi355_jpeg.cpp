// C ABI of the MI355X strict JPEG encode path (include/mi355_jpeg.h).
//
// Host-side runtime around the HIP kernels of jpeg_kernels.hip: context,
// device workspace, table upload, launch sequencing, stage probes and the
// build-defined JFIF framer.  There is no CPU compute path in this library: every
// compute entry point needs a usable gfx950 device and fails loudly otherwise.
#include "../../include/mi355_jpeg.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <system_error>
#include <vector>

#include <cmath>

#include "jpeg_device.h"
#include "jpeg_screen_tables.h"
#include "jpeg_tables.h"

using namespace mi355;

namespace {

// ---- Annex K tables (what utils.hpp:42-62 / huffman.hpp hold) --------------
const uint8_t kQ50Lum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                             14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                             18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                             49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQ50Chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                             24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kBitsDcL[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kBitsDcC[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kValDc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kBitsAcL[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kValAcL[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};
const uint8_t kBitsAcC[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kValAcC[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};

void canonical(const uint8_t bits[16], const uint8_t* val, mi355_huff_table* t) {
    memset(t, 0, sizeof *t);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) {
            t->code[val[k]] = code++;
            t->len[val[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

// The reference's four code tables.  DC tables have 12 entries (sizes 0..11), AC
// tables 16 runs x 11 sizes (0..10); everything else has no code.  huffman.hpp:92-98
// spells AC-luma run 3 / sizes 4..10 with one extra leading '1' (17 bits): kept.
// typos: true = the reference's tables (huffman.hpp incl. the seven 17-bit entries);
// false = the Annex K codes proper (standard mode).
void reference_huffman(int table, mi355_huff_table* t, bool typos = true) {
    switch (table) {
        case 0: canonical(kBitsDcL, kValDc, t); break;
        case 1: canonical(kBitsDcC, kValDc, t); break;
        case 2: canonical(kBitsAcL, kValAcL, t); break;
        default: canonical(kBitsAcC, kValAcC, t); break;
    }
    for (int rs = 0; rs < 256; ++rs) {
        int run = rs >> 4, size = rs & 15;
        bool has = table < 2 ? (run == 0 && size <= 11) : (size <= 10 && (size > 0 || run == 0 || run == 15));
        if (!has) t->len[rs] = 0, t->code[rs] = 0;
    }
    if (table == 2 && typos)
        for (int s = 4; s <= 10; ++s) {
            int rs = (3 << 4) | s;
            t->code[rs] |= 1u << t->len[rs];
            t->len[rs] += 1;
        }
}

// One set of stage-boundary events per profiled encode call (slots 0..4:
// start, after transform, after sizes, after scan, after emit).
struct EventSet {
    hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct mi355_jpeg_ctx {
    int device = 0;
    int n_cus = 0;  // compute units of the device
    // batches are encoded as two halves: the tail kernels of the first half run on `side` under the
    // block-encode kernel of the second half
    hipStream_t side = nullptr;
    hipEvent_t ev_half = nullptr, ev_side = nullptr;
    std::vector<hipEvent_t> ev_set;  // per workspace set: its last part's tail kernels are done
    uint32_t batch_parts = 0xFFFFu;  // upper limit of the parts of a batch (default: none; MI355_JPEG_BATCH_PARTS sets 1..8, 1 = one part)

    uint32_t qlum[64], qchrom[64];
    mi355_huff_table huff[4];
    mi355_huff_table huff_std[4];  // standard mode (MI355_F_STANDARD): Annex K proper unless the caller set a table
    // device-resident tables
    double* d_q = nullptr;       // [2][64] doubles, natural order
    uint32_t* d_qzz = nullptr;   // [2][64] integers, zig-zag order (standard mode)
    uint32_t* d_lut = nullptr;   // [2 modes][4][256] code<<5|len (second set: standard mode)
    uint32_t* d_status = nullptr;
    // workspace (grown on demand, never shrunk)
    uint32_t* d_coefs = nullptr;
    size_t coefs_cap = 0;  // dwords
    uint32_t* d_unit_off = nullptr;
    size_t unit_off_cap = 0;
    uint32_t* d_tile_bits = nullptr;
    uint64_t* d_tile_off = nullptr;
    size_t tiles_cap = 0;  // entries of d_tile_bits; d_tile_off has tiles_cap + frames_cap
    size_t tile_off_cap = 0;
    // staging for the host-buffer entry points
    uint8_t* d_in = nullptr;
    size_t in_cap = 0;
    uint8_t* d_out = nullptr;
    size_t out_cap = 0;
    uint64_t* d_bits = nullptr;
    size_t bits_cap = 0;
    // screened (integer-MFMA) pipeline
    uint4* d_afrag = nullptr;       // MFMA A fragments of the fixed-point maps (static): [strict, standard]
    double* d_qconst = nullptr;     // [2][64][4] accept thresholds for the current tables
    float* d_qconst_f = nullptr;    // [2 maps][2][16][8] fp32 first-look scale factors and thresholds
    uint32_t* d_lut2 = nullptr;     // [2 modes][2][66][16] whole AC symbols for |value| <= 31
    uint32_t* d_counters = nullptr; // [64] arena overflow-pool words, one per part in flight
    uint32_t max_sets = 0;          // MI355_JPEG_MAX_SETS (tests): upper limit of the workspace sets of a batch (0 = none)
    int taper = -1;                 // MI355_JPEG_TAPER: size of a part in per cent of the part in front (0 = equal parts; -1: the default of the mode)
    uint32_t stagger = 0;           // MI355_JPEG_STAGGER (timing experiment, 0..64): later-dispatched workgroups start this many sleeps late
    unsigned long long* d_stats = nullptr;  // [0] second looks, [1] exact units, [2] rewalked units, [3] general-loop passes (mi355_jpeg_screen_stats)
    uint32_t last_launches = 0;     // block-encode launches of the last encode call
    uint8_t* d_stage[4] = {nullptr, nullptr, nullptr, nullptr};  // scratch of the stage-by-stage entry points
    size_t stage_cap[4] = {0, 0, 0, 0};
    uint32_t* d_meta = nullptr;  // per workspace set: one word per unit, then one per pass (meta_words)
    size_t meta_cap = 0;
    uint32_t* d_arena = nullptr;
    size_t arena_cap = 0;           // words
    double tau_scale = 1.0;         // debug: widen the accept margins to force the exact recomputation
    uint32_t screen_waves = 2048;   // persistent single-wave workgroups of k_screen_encode
    int transform_mode = 2;         // 0 exact fp64 chain (unrolled), 1 exact (looped), 2 screened MFMA + exact recomputation of undecided units
    uint32_t emit_lds_words = 4096;
    uint32_t* d_stuff_counts = nullptr;  // byte stuffing scratch
    size_t stuff_cap = 0;
    uint64_t* d_stuff_offs = nullptr;
    size_t stuff_offs_cap = 0;
    unsigned long long* d_stamps = nullptr;  // diagnostic build only
    int profiling = 0;              // 0 off, 1 all stages, 2 transform only
    std::vector<EventSet> ev_pool;  // grown on demand, reused after a reset
    size_t ev_used = 0;             // sets recorded since profiling was enabled
    bool ev_open = false;           // between slot 0 and the last slot of one encode call
};

namespace {

// d_tile_off: restart intervals (the frame's [tiles] byte-aligned tile offsets) or nullptr
int stuff_scan(mi355_jpeg_ctx* c, const void* d_scan, const uint64_t* d_bits, size_t max_scan_bytes, void* d_out,
               size_t cap, uint64_t* d_out_len, const uint64_t* d_tile_off, uint32_t tiles, void* stream);

inline int hip_err(hipError_t e) { return e == hipSuccess ? MI355_OK : MI355_E_HIP - (int)e; }
#define HIP_TRY(x)                             \
    do {                                       \
        hipError_t _e = (x);                   \
        if (_e != hipSuccess) return hip_err(_e); \
    } while (0)

template <typename T>
int ensure(T*& p, size_t& cap, size_t need, bool zero = false) {
    if (need <= cap && p) return MI355_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc((void**)&p, need * sizeof(T)) != hipSuccess) return MI355_E_ALLOC;
    // The fill runs on the null stream and may still be in flight when hipMemset returns; the caller's
    // stream is usually non-blocking (no implicit ordering with the null stream): wait for it here, or
    // the first kernel could add into accumulators that are zeroed afterwards.
    if (zero && (hipMemset(p, 0, need * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess))
        return MI355_E_ALLOC;
    cap = need;
    return MI355_OK;
}

int upload_tables(mi355_jpeg_ctx* c) {
    // The copies below run on the null stream; encode calls run on the caller's (usually non-blocking)
    // streams, which do not order themselves against it in either direction.  Wait for everything in
    // flight first, or kernels of earlier calls could read a half-updated table set (quantiser
    // divisors, accept thresholds and Huffman LUTs that do not belong together).
    HIP_TRY(hipDeviceSynchronize());
    double q[128];
    for (int i = 0; i < 64; ++i) {
        q[i] = (double)c->qlum[i];
        q[64 + i] = (double)c->qchrom[i];
    }
    static const uint8_t zz[64] = MI355_ZIGZAG_TABLE;
    uint32_t lut[2 * 4 * 256], qzz[128];
    for (int m = 0; m < 2; ++m)
        for (int t = 0; t < 4; ++t) {
            const mi355_huff_table& h = m ? c->huff_std[t] : c->huff[t];
            for (int i = 0; i < 256; ++i) lut[(m * 4 + t) * 256 + i] = h.len[i] ? ((h.code[i] << 5) | h.len[i]) : 0u;
        }
    for (int R = 0; R < 64; ++R) qzz[R] = c->qlum[zz[R]], qzz[64 + R] = c->qchrom[zz[R]];
    HIP_TRY(hipMemcpy(c->d_qzz, qzz, sizeof qzz, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_q, q, sizeof q, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    // Accept thresholds of the screened transform (jpeg_screen_kernels.hip), per channel type
    // and zig-zag position R, for z = (fixed-point map)/Q:
    //   first look  (top four digits):  |c/Q - z1| <= (2^-19 + eps_R + fixerr)/Q + fp slop
    //   second look (all five digits):  |c/Q - z2| <= (eps_R + fixerr)/Q + fp slop
    // fp slop: z = y*s (relative 2^-52 of |z| < 2^12) and |z|+0.5 (2^-41): 2^-38 covers both.
    // Standard mode shares the first-look constants: its map is the definition (delta = 0), so the
    // strict margins are merely wider than needed; its second look decides exactly in integers.
    double qc[2][64][4];
    for (int ct = 0; ct < 2; ++ct)
        for (int R = 0; R < 64; ++R) {
            const double Q = (double)(ct ? c->qchrom[zz[R]] : c->qlum[zz[R]]);
            const double delta = kScreenEps[R] + kScreenFixErr;
            const double slop = std::ldexp(1.0, -38);
            double tau1 = ((std::ldexp(1.0, -19) + delta) / Q * 1.000001 + slop) * c->tau_scale;
            double tau2 = (delta / Q * 1.000001 + slop) * c->tau_scale;
            qc[ct][R][0] = std::ldexp(1.0, 8 - kScreenFracBits) / Q;
            qc[ct][R][1] = tau1 < 0.5 ? 0.5 - tau1 : -1.0;  // -1: never accepted
            qc[ct][R][2] = std::ldexp(1.0, -kScreenFracBits) / Q;
            qc[ct][R][3] = tau2 < 0.5 ? 0.5 - tau2 : -1.0;
        }
    HIP_TRY(hipMemcpy(c->d_qconst, qc, sizeof qc, hipMemcpyHostToDevice));
    // First look in fp32 (screen_quantise, jpeg_screen_devfn.h): the top three digits give Y' = Lt p / 2^16 without
    // the two low digits; zf = fl(fma(acc4, 2^16, fl(acc3 * 256 + acc2)) * sf), sf = fl(2^-23/Q).
    //   |c/Q - Y' 2^-23/Q| <= (E1_R + delta_R)/Q   (dropped digits: exact worst case per row; map + chain error)
    //   |zf - Y' 2^-23/Q|  <= |z| 2^-22 + 2^-18/Q   (three fp32 roundings; the conversion of acc3 * 256 + acc2)
    // the first term of the second line is covered by 2^-21 max|zf| (the row's largest |zf| over all inputs, from the
    // table); thr = 0.5 - tau - 2^-22 - 2^-21 max|zf|, and the kernel tests d^2 < thr^2 (square rounded DOWN to float).
    // One set per map (strict / standard).
    float qf[2][2][16][8];
    for (int m = 0; m < 2; ++m)
        for (int ct = 0; ct < 2; ++ct)
            for (int grp = 0; grp < 16; ++grp)
                for (int r = 0; r < 4; ++r) {
                    const int R = 16 * (grp >> 2) + 4 * (grp & 3) + r;
                    const double Q = (double)(ct ? c->qchrom[zz[R]] : c->qlum[zz[R]]);
                    const auto& limb = m ? kStdLimb : kScreenLimb;
                    const double delta = kScreenEps[R] + kScreenFixErr;
                    long s1 = 0, s0 = 0;
                    for (int i = 0; i < 64; ++i) s1 += std::abs((int)limb[1][R][i]), s0 += std::abs((int)limb[0][R][i]);
                    const double e1 = std::ldexp(128.0 * (256.0 * (double)s1 + (double)s0), -kScreenFracBits);
                    // the |zf| 2^-21 term at the row's largest |zf|: |Y' 2^-23| <= 128 sum_i |top three digits of row R|
                    double s3 = 0;
                    for (int i = 0; i < 64; ++i)
                        s3 += std::fabs(65536.0 * limb[4][R][i] + 256.0 * limb[3][R][i] + (double)limb[2][R][i]);
                    const double zmax = std::ldexp(128.0 * s3, 16 - kScreenFracBits) / Q * 1.000002;
                    qf[m][ct][grp][r] = (float)(std::ldexp(1.0, 16 - kScreenFracBits) / Q);
                    const double tau = ((e1 + std::ldexp(1.0, -18) + delta) / Q * 1.000001) * c->tau_scale + std::ldexp(1.0, -22);
                    const double thr = 0.5 - tau - std::ldexp(zmax, -21);
                    // the kernel tests d^2 < thr^2: the square, rounded DOWN to float (negative: never accepted)
                    float th2 = -1.0f;
                    if (thr > 0.0) {
                        th2 = (float)(thr * thr);
                        if ((double)th2 > thr * thr) th2 = std::nextafterf(th2, -1.0f);
                    }
                    qf[m][ct][grp][4 + r] = th2;
                }
    HIP_TRY(hipMemcpy(c->d_qconst_f, qf, sizeof qf, hipMemcpyHostToDevice));
    // Whole-symbol tables of the screened pipeline's unit walk: for run r and value v (|v| <= 31)
    // the Huffman code of (r, size(v)) followed by v's value bits, left-aligned in 32 bits, with
    // the total length in bits 4..0; 0 = the reference has no code.  Layout [v + 32][run]; the value-0 row holds ZRL
    // at run 15 and nothing else; the 66th row = {0 (lanes that ran out of symbols land here), EOB, 0, ...}
    // (jpeg_screen_devfn.h: kLut2Cols, kLut2Zrl, kLut2Eob).
    constexpr size_t kL2C = 16, kL2 = 66 * kL2C;  // kLut2Cols, kLut2Words
    std::vector<uint32_t> lut2(2 * 2 * kL2, 0u);
    for (int m = 0; m < 2; ++m)
        for (int ct = 0; ct < 2; ++ct) {
            const mi355_huff_table& t = m ? c->huff_std[2 + ct] : c->huff[2 + ct];
            uint32_t* L = &lut2[(size_t)(m * 2 + ct) * kL2];
            auto entry = [&](uint32_t bits, int len) -> uint32_t { return len ? ((bits << (32 - len)) | (uint32_t)len) : 0u; };
            for (int r = 0; r < 16; ++r)
                for (int v = -31; v <= 31; ++v) {
                    if (v == 0) continue;
                    int a = v < 0 ? -v : v, size = 0;
                    while (a) ++size, a >>= 1;
                    int rs = (r << 4) | size;
                    if (!t.len[rs]) {  // a hole in the table: kLut2Miss, the general loop reports it (quirk Q13)
                        L[(v + 32) * kL2C + r] = 31u;
                        continue;
                    }
                    uint32_t vb = (uint32_t)(v < 0 ? v + (1 << size) - 1 : v);
                    L[(v + 32) * kL2C + r] = entry((t.code[rs] << size) | vb, t.len[rs] + size);
                }
            // rows 0 (-32) and 64 (+32): kLut2Miss in every run column (larger values are clamped onto them)
            for (int r = 0; r < 16; ++r) L[r] = L[64 * kL2C + r] = 31u;
            L[32 * kL2C + 15] = entry(t.code[0xF0], t.len[0xF0]);
            L[65 * kL2C + 1] = entry(t.code[0x00], t.len[0x00]);
        }
    HIP_TRY(hipMemcpy(c->d_lut2, lut2.data(), lut2.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // copies from pageable memory may return before the DMA has landed; encode calls run on
    // non-blocking streams that do not order themselves after the null stream
    HIP_TRY(hipStreamSynchronize(nullptr));
    return MI355_OK;
}

// MFMA A fragments of the fixed-point map: fragment (mt, digit), lane l = (m = l & 15, g = l >> 4)
// holds row 16*mt + m (a zig-zag position), input samples 16*g .. 16*g+15.
constexpr size_t kAfragBytes = (size_t)4 * kScreenLimbs * 64 * 16;  // one map
int upload_afrag(mi355_jpeg_ctx* c) {
    std::vector<int8_t> h(2 * kAfragBytes);
    for (int m = 0; m < 2; ++m)  // 0: the reference's chain as a linear map; 1: the true DCT-II (standard mode)
        for (int mt = 0; mt < 4; ++mt)
            for (int l = 0; l < kScreenLimbs; ++l)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 16; ++i)
                        h[m * kAfragBytes + (((size_t)mt * kScreenLimbs + l) * 64 + lane) * 16 + i] =
                            (m ? kStdLimb : kScreenLimb)[l][16 * mt + (lane & 15)][16 * (lane >> 4) + i];
    HIP_TRY(hipMemcpy(c->d_afrag, h.data(), h.size(), hipMemcpyHostToDevice));
    return MI355_OK;
}

// MFMA A fragments of standard mode's per-pixel colour conversion (jpeg_tables.h; used by the 4:4:4 kernel).  The B operand of v_mfma_i32_16x16x64_i8 is raw
// RGB: lane (n, g) supplies 16 consecutive bytes (XOR 0x80 = x - 128) of ITS OWN rows as K chunk g, so the A matrix is
// block diagonal -- output row 4g + r only reads K chunk g -- and lane (n, g) receives the four outputs r = 0..3 computed
// from its own bytes: no data crosses lanes.  A set is the 4 x 16 matrix W[r][t] (digit of the coefficient that byte t of
// the chunk contributes to output r), laid out as fragment lane (m, kq) = W[m & 3][.] if (m >> 2) == kq, else zero.
// Coefficients are split into two balanced base-256 digits: c = 256 d1 + d0, d0 in -128..127.
int upload_csc_frag(mi355_jpeg_ctx* c) {
    std::vector<int8_t> h((size_t)(kCscSets + kCscStrictSets) * 64 * 16, 0);
    auto put = [&](int set, auto&& coef /* (r, t) -> coefficient or 0 */) {
        for (int digit = 0; digit < 2; ++digit)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, kq = lane >> 4;
                if ((m >> 2) != kq) continue;
                for (int t = 0; t < 16; ++t) {
                    const int cf = coef(m & 3, t);
                    const int d0 = ((cf + 128) & 255) - 128, d1 = (cf - d0) / 256;
                    h[((size_t)(set + digit) * 64 + lane) * 16 + t] = (int8_t)(digit ? d1 : d0);
                }
            }
    };
    // row pairs: the chunk of half 0 is bytes 0..15 of a block row (pixels 0..3 at chunk bytes 3r..3r+2), that of half 1 is
    // bytes 8..23 (pixels 4..7 at chunk bytes 4 + 3r ..)
    for (int chan = 0; chan < 3; ++chan)
        for (int half = 0; half < 2; ++half)
            put((chan * 2 + half) * 2, [&](int r, int t) {
                const int o = t - (half ? 4 : 0) - 3 * r;
                return o >= 0 && o < 3 ? kStdCsc[chan][o] : 0;
            });
    // strict mode: the reference's integer numerators, one pair of digit sets per channel (jpeg_tables.h)
    for (int chan = 0; chan < 3; ++chan)
        put(kCscSets + chan * 2, [&](int r, int t) {
            const int o = t - 3 * r;
            return o >= 0 && o < 3 ? kCscStrict[chan][o] : 0;
        });
    HIP_TRY(hipMemcpy(c->d_afrag + 2 * kAfragBytes / sizeof(uint4), h.data(), h.size(), hipMemcpyHostToDevice));
    return MI355_OK;
}

int make_geom(uint32_t W, uint32_t H, uint32_t flags, const void* base, Geom* g) {
    if (W == 0 || H == 0 || W > 65535u || H > 65535u) return MI355_E_ARG;
    if ((flags & MI355_F_420) && !(flags & MI355_F_STANDARD)) return MI355_E_ARG;  // real 4:2:0 MCUs: standard mode only
    if ((flags & MI355_F_RESTART) && !(flags & MI355_F_STANDARD)) return MI355_E_ARG;
    const bool s420 = (flags & MI355_F_420) != 0;
    const uint32_t A = s420 ? 16 : 8;  // MCU edge
    uint32_t W8 = (W + A - 1) / A * A, H8 = (H + A - 1) / A * A;
    // the reference mirrors with `oldWidth - diff` in size_t (utils.cpp:215,226):
    // a pad wider than the image underflows there (UB) -> refused here
    if (W8 - W > W || H8 - H > H) return MI355_E_ARG;
    g->W = W;
    g->H = H;
    g->W8 = W8;
    g->H8 = H8;
    g->nbx = W8 / 8;
    g->nmx = W8 / 16;
    g->passes = s420 ? 6 : 3;
    g->N = (W8 / A) * (H8 / A);  // 8x8 blocks per channel, or 16x16 MCUs in 4:2:0
    g->tiles = (g->N + 63) / 64;
    g->flags = flags;
    g->frame_stride = (uint64_t)W * H * 3;
    // the fast row loads use 32-bit byte offsets inside a frame (load_raw_rowpair)
    g->fast_rows = (W % 8 == 0) && (((uintptr_t)base & 7u) == 0) && ((uint64_t)W * H * 3u < (1ull << 32));
    return MI355_OK;
}

int ensure_workspace(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames) {
    int e;
    if ((e = ensure(c->d_coefs, c->coefs_cap, coef_dwords(g) * n_frames))) return e;
    if ((e = ensure(c->d_unit_off, c->unit_off_cap, unit_off_words(g) * n_frames))) return e;
    if ((e = ensure(c->d_tile_bits, c->tiles_cap, (size_t)g.tiles * n_frames, true))) return e;
    if ((e = ensure(c->d_tile_off, c->tile_off_cap, tile_off_entries(g, n_frames)))) return e;
    return MI355_OK;
}

// Per-unit metadata of the screened pipeline, in words, for n_frames frames: one word per unit slot, then one per pass
// (the arena offset of the pass's first string).
inline size_t meta_slots(const Geom& g, uint32_t n_frames) { return (size_t)g.tiles * g.passes * 64 * n_frames; }
inline size_t meta_words(const Geom& g, uint32_t n_frames) { return meta_slots(g, n_frames) / 64 * 65; }

constexpr size_t kMaxEventSets = 1u << 16;
constexpr unsigned long kEmitWordsMax = 4096;  // kEmitLdsWords (jpeg_devfn.h): the bit-assembly window of k_emit / k_merge

// slot 0 opens a new event set for this call
// workspace of the screened pipeline; arena_words: capacity for the AC blobs
int ensure_screen_workspace(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames, size_t arena_words) {
    int e;
    if ((e = ensure(c->d_meta, c->meta_cap, meta_words(g, n_frames)))) return e;
    if ((e = ensure(c->d_arena, c->arena_cap, arena_words))) return e;
    return MI355_OK;
}

// Arena geometry for a payload bound of `need_words`: one private region per persistent wave
// plus an overflow pool that alone could hold everything (chunked, hence the extra slack).
struct ArenaPlan {
    uint32_t grid, region_words;
    size_t total_words;
};
// need_words sizes the waves' private regions (what the frames need if they fit their output); pool_words the shared
// overflow pool behind them.  The pool is sized for the WORST case of every unit of the part (54 words each), so that
// it cannot run dry whatever the frames hold: a frame far over its output capacity is then flagged by the capacity
// check of its own tile scan and by nothing else, and cannot take arena space away from the frames beside it.  (The
// pool is only touched where a wave's private region has filled up: the pages are reserved, not used.)
ArenaPlan plan_arena(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames, size_t need_words, size_t pool_words) {
    ArenaPlan p;
    p.grid = screen_grid(g, n_frames, c->screen_waves);
    p.region_words = (uint32_t)((need_words + p.grid - 1) / p.grid);
    p.total_words = (size_t)p.grid * p.region_words + pool_words + (size_t)p.grid * 1024 + 64;
    return p;
}

ScreenParams screen_params(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames, const ArenaPlan& plan,
                           uint32_t* coefs) {
    ScreenParams sp;
    const size_t arena_words = plan.total_words;
    // the later-dispatched half of a launch that fills the device two workgroups per CU (see the kernel)
    sp.prio_from_wg = (c->n_cus > 0 && plan.grid / 4 == 2u * (uint32_t)c->n_cus) ? (uint32_t)c->n_cus : 0xFFFFFFFFu;
    sp.stagger = c->stagger;
    sp.region_words = plan.region_words;
    sp.overflow_base = plan.grid * plan.region_words;
    const bool stdm = (g.flags & MI355_F_STANDARD) != 0;
    sp.afrag = c->d_afrag + (stdm ? kAfragBytes / sizeof(uint4) : 0);
    sp.csc_frag = c->d_afrag + 2 * kAfragBytes / sizeof(uint4);
    sp.qconst = c->d_qconst;
    sp.qconst_f = c->d_qconst_f + (stdm ? 256 : 0);
    sp.qd = c->d_q;
    sp.qnat_zz = c->d_qzz;
    sp.lut = c->d_lut + (stdm ? 1024 : 0);
    sp.lut2 = c->d_lut2 + (stdm ? 2 * 66 * 16 : 0);
    sp.meta = c->d_meta;
    sp.pass_off = c->d_meta + meta_slots(g, n_frames);
    sp.arena = c->d_arena;
    sp.arena_words = (uint32_t)(arena_words > 0xFFFFFFFFull ? 0xFFFFFFFFull : arena_words);
    sp.counters = c->d_counters;
    sp.stats = c->d_stats;
    sp.status = c->d_status;
    sp.tile_bits = c->d_tile_bits;
    sp.coefs = coefs;
    sp.samples = nullptr;
    sp.stamps = nullptr;
#ifdef MI355_STAMPS
    {   // diagnostic build: one lazily allocated buffer, dumped by mi355_jpeg_sync
        static unsigned long long* d_st = nullptr;
        if (!d_st) (void)hipMalloc((void**)&d_st, 8192 * 8 * sizeof(unsigned long long));  // [0,2048): phase sums, [2048,4096): start/end
        sp.stamps = d_st;
        c->d_stamps = d_st;
    }
#endif
    return sp;
}

void record(mi355_jpeg_ctx* c, int i, hipStream_t s) {
    if (!c->profiling) return;
    if (c->profiling == 2 && i > 1) return;
    if (i == 0) {
        if (c->ev_used >= kMaxEventSets) c->ev_used = 0;  // wrap: keep the most recent calls
        if (c->ev_used == c->ev_pool.size()) c->ev_pool.emplace_back();
        ++c->ev_used;
        c->ev_open = true;
    }
    if (!c->ev_open) return;  // stage probes / entropy-only calls are not profiled
    if (i == (c->profiling == 2 ? 1 : 4)) c->ev_open = false;
    EventSet& es = c->ev_pool[c->ev_used - 1];
    if (!es.e[i] && hipEventCreate(&es.e[i]) != hipSuccess) return;
    (void)hipEventRecord(es.e[i], s);
}

int elapsed(const EventSet& es, int a, int b, float* ms) {
    *ms = 0.f;
    if (!es.e[a] || !es.e[b]) return MI355_OK;
    hipError_t e = hipEventSynchronize(es.e[b]);
    if (e != hipSuccess) return MI355_E_HIP - (int)e;
    e = hipEventElapsedTime(ms, es.e[a], es.e[b]);
    return e == hipSuccess ? MI355_OK : MI355_E_HIP - (int)e;
}

int timings_of(mi355_jpeg_ctx* c, const EventSet& es, mi355_jpeg_timings* t) {
    int e;
    memset(t, 0, sizeof *t);
    if ((e = elapsed(es, 0, 1, &t->transform_ms))) return e;
    if (c->profiling == 2) {
        t->total_ms = t->transform_ms;
        return MI355_OK;
    }
    if ((e = elapsed(es, 1, 2, &t->size_ms))) return e;
    if ((e = elapsed(es, 2, 3, &t->scan_ms))) return e;
    if ((e = elapsed(es, 3, 4, &t->emit_ms))) return e;
    return elapsed(es, 0, 4, &t->total_ms);
}

// entropy stages on coefficients already in the workspace
int run_entropy(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames, uint8_t* d_out, size_t out_stride,
                uint64_t* d_bits, hipStream_t s) {
    HIP_TRY(launch_unit_sizes(g, n_frames, c->d_coefs, c->d_lut, c->d_unit_off, c->d_tile_bits,
                              c->d_status, s));
    record(c, 2, s);
    // invariant: d_tile_bits is all zero between API calls (the screened pipeline accumulates into it)
    HIP_TRY(launch_tile_scan(g, n_frames, c->d_tile_bits, c->d_tile_off, d_out, out_stride, d_bits,
                             c->d_status, nullptr, true,
                             scan_chunks(g) ? c->d_tile_off + ((size_t)g.tiles + 1) * n_frames : nullptr, false, s));
    record(c, 3, s);
    HIP_TRY(launch_emit(g, n_frames, c->d_coefs, c->d_lut, c->d_unit_off, c->d_tile_off, d_out,
                        out_stride, c->d_status, c->emit_lds_words, s));
    record(c, 4, s);
    return MI355_OK;
}

// One part of a batch: frames [f0, f0 + nf).  Parts alternate between two sets of {meta, arena, overflow counter}:
// part i + 2 reuses the set of part i once that part's tail kernels are done.
constexpr uint32_t kCounters = 64;  // overflow counters: one per workspace set (re-armed by the tile scan of the part that used the set)
struct BatchPart {
    uint32_t f0, nf;
    ArenaPlan plan;
    uint32_t set;      // workspace set of this part
    uint32_t counter;  // its overflow-pool counter (re-armed by the part's tile scan)
    size_t meta_off, arena_off;  // where the set lies in d_meta / d_arena (words)
    uint32_t set_nf;             // frames the set's metadata was laid out for
};

ScreenParams part_params(mi355_jpeg_ctx* c, const Geom& g, const BatchPart& p) {
    ScreenParams sp = screen_params(c, g, p.nf, p.plan, nullptr);
    sp.meta = c->d_meta + p.meta_off;
    sp.pass_off = sp.meta + meta_slots(g, p.set_nf);  // behind the unit words of the largest part the set holds
    sp.arena = c->d_arena + p.arena_off;
    sp.counters = c->d_counters + p.counter;
    sp.tile_bits = c->d_tile_bits + (size_t)p.f0 * g.tiles;
    return sp;
}

int launch_tails(mi355_jpeg_ctx* c, const Geom& g, const BatchPart& p, const ScreenParams& sp, uint8_t* d_out,
                 size_t out_stride, uint64_t* d_bits, hipStream_t s, bool rec, uint32_t batch_frames) {
    HIP_TRY(launch_dc_heads(g, p.nf, sp, s));
    if (rec) record(c, 2, s);  // slot [1,2] = DC heads (the other tile sums are accumulated by the encode kernel itself)
    HIP_TRY(launch_tile_scan(g, p.nf, sp.tile_bits, c->d_tile_off + (size_t)p.f0 * (g.tiles + 1),
                             d_out + (size_t)p.f0 * out_stride, out_stride, d_bits + p.f0, c->d_status, sp.counters,
                             true, scan_chunks(g) ? c->d_tile_off + ((size_t)g.tiles + 1) * batch_frames + (size_t)p.f0 * scan_chunks(g) : nullptr,
                             true, s));
    if (rec) record(c, 3, s);
    HIP_TRY(launch_merge(g, p.nf, sp.meta, sp.pass_off, sp.arena, sp.lut, c->d_tile_off + (size_t)p.f0 * (g.tiles + 1),
                         d_out + (size_t)p.f0 * out_stride, out_stride, d_bits + p.f0, c->emit_lds_words, batch_frames > p.nf, s));
    if (rec) record(c, 4, s);
    return MI355_OK;
}

// Screened pipeline: k_screen_encode -> k_dc_heads -> k_tile_scan -> k_merge.
// Event slots: [0,1] fused block encode (transform_ms), [1,2] DC heads (size_ms), [2,3] scan, [3,4] merge (emit_ms).
// Batches of four or more frames go in parts of ~128 Mpixel (16 4K frames: ~0.5 ms of block encode; the kernel's time
// per frame is best between 12 and 20 4K frames per launch): the tail kernels of a part run on a side
// stream under the block-encode kernel of the next part (they fit next to its resident workgroups);
// with per-stage profiling on, one part, so that the stage times stay meaningful.
//
// Workspace (device memory the library allocates; grown on demand, never shrunk): TWO sets of {per-unit metadata,
// string arena}, each sized for ONE part, whatever the batch -- parts alternate between them.  Per frame of a part:
// 4 bytes per unit (+ 4 per pass) of metadata, and an arena of 2 x min(9/16 x out_stride + 4 x units, 216 x units) + 0.25 MiB bytes
// (units = blocks x 3; 216 bytes = the longest possible AC string; the factor 2 = every wave's private region + an
// overflow pool that could hold everything).  A part's arena offsets are 32-bit words relative to the part, so the
// size of a batch is not limited by them; parts shrink where a frame is so large that 16 of them would not fit.
int run_screened(mi355_jpeg_ctx* c, const Geom& g, uint32_t n_frames, const uint8_t* d_rgb, uint8_t* d_out,
                 size_t out_stride, uint64_t* d_bits, hipStream_t s) {
    // AC strings are word aligned per unit (<= bits/32 + 1 words); strings longer than the LDS slot (24 words) get a
    // full 54-word run, i.e. at most 54/24 of their own size.  A frame whose strings exceed 9/4 of the output capacity
    // (+ one word per unit) cannot fit the output either; and no frame needs more than 54 words per unit.
    const size_t by_capacity = out_stride / 4 * 9 / 4 + unit_count(g) + 64, by_units = unit_count(g) * 54 + 64;
    const size_t frame_words = by_capacity < by_units ? by_capacity : by_units;
    uint32_t nparts = 1;
    if (n_frames >= 4 && c->profiling != 1) {
        const uint64_t px = (uint64_t)n_frames * g.W * g.H;
        nparts = (uint32_t)((px + (1ull << 26)) >> 27);
        if (nparts > n_frames / 2) nparts = n_frames / 2;
        if (nparts > c->batch_parts) nparts = c->batch_parts;
        if (nparts < 1) nparts = 1;
    }
    // 32-bit word offsets inside a part: (frame_words + by_units) x frames + slack must stay below 2^32
    {
        const uint64_t max_pf = frame_words + by_units + (1u << 16) < (1ull << 32) ? ((1ull << 32) - (1ull << 24)) / (frame_words + by_units + (1u << 16)) : 0;
        if (max_pf == 0) return MI355_E_ARG;  // one frame alone beyond 2^32 arena words (> 39 M blocks at the capacity given)
        const uint32_t need_parts = (uint32_t)((n_frames + max_pf - 1) / max_pf);
        if (nparts < need_parts) nparts = need_parts;
    }
    // Part boundaries: part i = frames [cut[i], cut[i + 1]).
    //
    // Tapered (the ratio of a part to the part in front, in per cent; batches of four parts or more): every part's tail
    // kernels run under the NEXT part's block encode -- except the last part's, which nothing hides -- so the batch should
    // end on a short part; but a part's merge, beside an encode launch, takes 0.6 of the time that launch needs for the
    // same number of frames (strict mode, q50: 19 against 31 us per 4K frame), so a part may not be much shorter than that
    // fraction of the part in front or THAT part's tails stick out instead (which is why a short last part behind equal
    // parts gained nothing, gpurun r4tp).  Sizes therefore fall geometrically to a last part of ~32 Mpixel; the first part
    // takes what is left (nothing runs beside it).  The ratio that pays depends on merge time over encode time, i.e. on
    // mode and bit rate (tools/taper_probe.py, gpurun r4tj; 128 4K frames, Gpixel/s at ratio 0 / 55 / 62 / 70): strict
    // q25 286 / 292 / 290 / 290, q50 260 / 264 / 262 / 262, q75 216 / 207 / 219 / 221, q90 164 / 155 / 167 / 167; standard
    // 4:4:4 q50 282 / 258 / 250 / 274, 4:2:0 q50 450 / 425 / 404 / 444, 4:2:0 q90 117 / 123 / 128 / 124 -- so strict mode
    // runs at 70 (never behind equal parts) and the standard modes, whose block encode is twice as fast for the same
    // merge, with equal parts.  Tapered parts get workspace sets of their own sizes; when those do not fit the memory
    // budget the batch falls back to equal parts that share equal sets.
    const uint32_t taper = c->taper >= 0 ? (uint32_t)c->taper : ((g.flags & MI355_F_STANDARD) ? 0u : 70u);
    std::vector<uint32_t> cut;
    auto equal_cut = [&]() {
        cut.assign(1, 0u);
        for (uint32_t i = 1; i <= nparts; ++i) cut.push_back((uint32_t)(((uint64_t)n_frames * i) / nparts));
    };
    const uint64_t frame_px = (uint64_t)g.W * g.H;
    bool tapered = false;
    const uint32_t nparts_equal = nparts;
    if (taper && nparts >= 4) {
        std::vector<uint32_t> rev;  // sizes from the last part backwards
        double x = (double)(((1ull << 25) + frame_px / 2) / frame_px);
        if (x < 1.0) x = 1.0;
        uint32_t sum = 0;
        while (sum + (uint32_t)(x + 0.5) < n_frames) {
            rev.push_back((uint32_t)(x + 0.5));
            sum += rev.back();
            x = x * 100.0 / taper;
        }
        rev.push_back(n_frames - sum);  // the first part: at most the next size of the series
        bool fits32 = true;
        for (uint32_t nf : rev) fits32 = fits32 && (uint64_t)nf * (frame_words + by_units + (1u << 16)) < (1ull << 32) - (1ull << 24);
        if (fits32 && rev.size() >= 3 && rev.size() <= kCounters) {
            cut.assign(1, 0u);
            for (size_t k = rev.size(); k-- > 0;) cut.push_back(cut.back() + rev[k]);
            nparts = (uint32_t)rev.size();
            tapered = true;
        }
    }
    if (!tapered) equal_cut();
    // Sets of {metadata, arena}: one per part while they fit a budget (a quarter of the free device memory, at most
    // 32 GB), otherwise as many as fit (at least 2) and part i reuses the set of part i - nsets once that part's tail
    // kernels are done.  (Reuse costs: the wait for the side stream between two block-encode launches keeps them from
    // running back to back -- 8 % on the bench with two alternating sets -- so sets are only shared when they must be.)
    size_t budget = (size_t)32 << 30;
    if (nparts > 2) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0, (void)hipGetLastError();
        free_b += c->meta_cap * sizeof(uint32_t) + c->arena_cap * sizeof(uint32_t);  // what this context holds already counts as free
        if (free_b / 4 < budget) budget = free_b / 4;
    }
    auto arena_of = [&](uint32_t nf) {
        const ArenaPlan pl = plan_arena(c, g, nf, (size_t)nf * frame_words, (size_t)nf * by_units);
        return (pl.total_words + 63) & ~(size_t)63;
    };
    std::vector<size_t> meta_off, arena_off;  // per SET, in words; one entry more: the totals
    std::vector<uint32_t> set_nf;
    uint32_t nsets = nparts;
    for (;;) {
        meta_off.assign(1, 0), arena_off.assign(1, 0), set_nf.clear();
        if (tapered) {  // a set per part, each of its part's size
            for (uint32_t i = 0; i < nparts; ++i) {
                set_nf.push_back(cut[i + 1] - cut[i]);
                meta_off.push_back(meta_off.back() + meta_words(g, set_nf.back()));
                arena_off.push_back(arena_off.back() + arena_of(set_nf.back()));
            }
            if (c->max_sets && nparts > c->max_sets) tapered = false;  // tests: shared sets
            else if ((meta_off.back() + arena_off.back()) * sizeof(uint32_t) <= budget || nparts <= 2) break;
            tapered = false;  // does not fit: equal parts, shared sets
            nparts = nparts_equal;
            equal_cut();
            continue;
        }
        uint32_t max_nf = 0;
        for (uint32_t i = 0; i < nparts; ++i) max_nf = cut[i + 1] - cut[i] > max_nf ? cut[i + 1] - cut[i] : max_nf;
        const size_t set_meta = meta_words(g, max_nf), set_arena = arena_of(max_nf);
        nsets = nparts;
        if (nparts > 2) {
            const size_t set_bytes = (set_meta + set_arena) * sizeof(uint32_t);
            const size_t fit = set_bytes ? budget / set_bytes : nparts;
            nsets = (uint32_t)(fit < 2 ? 2 : (fit < nparts ? fit : nparts));
            if (c->max_sets && nsets > c->max_sets) nsets = c->max_sets;  // tests
        }
        // the overflow counter belongs to the set: the wait that guards a set's reuse then also orders the counter's re-arm
        // (part i's tile scan) before part i + nsets allocates from it
        if (nsets > kCounters) nsets = kCounters;
        if (set_arena > 0xFFFFFFFFull) return MI355_E_ARG;  // (cannot happen: parts were sized for it)
        for (uint32_t k = 0; k < nsets; ++k) {
            set_nf.push_back(max_nf);
            meta_off.push_back(meta_off.back() + set_meta);
            arena_off.push_back(arena_off.back() + set_arena);
        }
        break;
    }
    if (tapered) nsets = nparts;
    int e;
    if ((e = ensure(c->d_meta, c->meta_cap, meta_off.back()))) return e;
    if ((e = ensure(c->d_arena, c->arena_cap, arena_off.back()))) return e;
    if (nparts > 1 && !c->side) {
        // The side stream at the highest priority.  Not for the order of dispatch (its kernels are small and take what they
        // get) but for the hardware queue: HIP hands its (four) hardware queues to streams in turn, and in a process that
        // holds more streams than that -- RCCL creates several when a process group is set up -- a default-priority side
        // stream can land on the queue of the caller's stream, where the tail kernels run BETWEEN the block-encode launches
        // instead of beside them: the N > 1 path of bench.py ran 13 % below the single process for that reason (229 against
        // 261 Gpixel/s per GPU, gpurun reh2).  Priority streams have queues of their own.
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // hi = the numerically smallest = highest priority
        HIP_TRY(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_hi));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_half, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    }
    while (nsets < nparts && c->ev_set.size() < nsets) {  // only batches that share sets need them
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->ev_set.push_back(ev);
    }
    record(c, 0, s);
    c->last_launches = nparts;
    for (uint32_t i = 0; i < nparts; ++i) {
        BatchPart part;
        part.f0 = cut[i];
        part.nf = cut[i + 1] - cut[i];
        part.plan = plan_arena(c, g, part.nf, (size_t)part.nf * frame_words, (size_t)part.nf * by_units);
        part.set = i % nsets;
        part.counter = part.set;
        part.meta_off = meta_off[part.set], part.arena_off = arena_off[part.set], part.set_nf = set_nf[part.set];
        const ScreenParams sp = part_params(c, g, part);
        if (i >= nsets) HIP_TRY(hipStreamWaitEvent(s, c->ev_set[part.set], 0));  // the tails of part i - nsets are done with this set
        // (Making this launch wait for the tails of part i - 2 -- they normally end a few per cent before block encode
        // i - 1 does -- costs 8 %: the wait keeps two block-encode launches from running back to back.  gpurun r4f.)
        HIP_TRY(launch_screen_encode(g, part.nf, d_rgb + (size_t)part.f0 * g.frame_stride, sp, false, c->screen_waves, s));
        if (i + 1 == nparts) {  // the last part's tails stay on the caller's stream
            record(c, 1, s);
            if ((e = launch_tails(c, g, part, sp, d_out, out_stride, d_bits, s, nparts == 1, n_frames))) return e;
        } else {  // tails on the side stream, under the next part's block encode
            HIP_TRY(hipEventRecord(c->ev_half, s));
            HIP_TRY(hipStreamWaitEvent(c->side, c->ev_half, 0));
            if ((e = launch_tails(c, g, part, sp, d_out, out_stride, d_bits, c->side, false, n_frames))) return e;
            if (i + nsets < nparts) HIP_TRY(hipEventRecord(c->ev_set[part.set], c->side));
        }
    }
    if (nparts > 1) {
        HIP_TRY(hipEventRecord(c->ev_side, c->side));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_side, 0));
        record(c, 2, s);
        record(c, 3, s);
        record(c, 4, s);
    }
    return MI355_OK;
}

// Screened transform only (stage probes): coefficients into the tiled workspace layout.
int run_screened_probe(mi355_jpeg_ctx* c, const Geom& g, const uint8_t* d_rgb, uint8_t* d_samples, hipStream_t s) {
    ArenaPlan plan = plan_arena(c, g, 1, unit_count(g) * 54, unit_count(g) * 54 + 64);
    if (plan.total_words > 0xFFFFFFFFull) return MI355_E_ARG;
    int e;
    if ((e = ensure_screen_workspace(c, g, 1, plan.total_words))) return e;
    ScreenParams sp = screen_params(c, g, 1, plan, c->d_coefs);
    sp.samples = d_samples;
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(c->d_tile_bits, 0, (size_t)g.tiles * sizeof(uint32_t), s));
    HIP_TRY(launch_screen_encode(g, 1, d_rgb, sp, true, c->screen_waves, s));
    HIP_TRY(launch_dc_heads(g, 1, sp, s));
    // leave the accumulators re-armed for the next encode call
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(c->d_tile_bits, 0, (size_t)g.tiles * sizeof(uint32_t), s));
    return MI355_OK;
}


// Environment knobs (development / A-B switches).  Every one of them is validated: a value that does not parse, or
// that could change RESULTS (accept margins narrower than proven, an unknown transform mode), fails the creation of
// a context with MI355_E_ARG instead of being taken as 0.
struct Knobs {
    int transform_mode = 2;
    uint32_t emit_lds_words = 4096;
    double tau_scale = 1.0;
    uint32_t batch_parts = 0xFFFFu;
    uint32_t max_sets = 0;
    int taper = -1;
    uint32_t screen_waves = 0;  // 0: the device's default
    uint32_t stagger = 0;
};
bool read_knobs(Knobs* k) {
    bool bad = false;
    auto knob_uint = [&](const char* name, unsigned long lo, unsigned long hi, unsigned long* out) {
        const char* v = getenv(name);
        if (!v) return false;
        char* end = nullptr;
        const unsigned long x = strtoul(v, &end, 10);
        if (!*v || *end || v[0] == '-' || v[0] == '+' || v[0] == ' ' || x < lo || x > hi) {
            bad = true;
            return false;
        }
        *out = x;
        return true;
    };
    unsigned long kv = 0;
    if (knob_uint("MI355_JPEG_TRANSFORM_MODE", 0, 2, &kv)) k->transform_mode = (int)kv;
    if (knob_uint("MI355_JPEG_EMIT_LDS_WORDS", 0, kEmitWordsMax, &kv)) k->emit_lds_words = (uint32_t)kv;
    if (const char* ts = getenv("MI355_JPEG_SCREEN_TAU_SCALE")) {
        // widens the accept margins of the screened transform (tests force the second look and the exact chain with
        // it).  Below 1 the margins would be narrower than the error bounds they stand for: refused, like anything
        // that is not a finite number.
        char* end = nullptr;
        const double x = strtod(ts, &end);
        if (!*ts || *end || !(x >= 1.0) || !std::isfinite(x)) bad = true;
        else k->tau_scale = x;
    }
    if (knob_uint("MI355_JPEG_BATCH_PARTS", 1, 8, &kv)) k->batch_parts = (uint32_t)kv;
    if (knob_uint("MI355_JPEG_MAX_SETS", 2, 64, &kv)) k->max_sets = (uint32_t)kv;
    if (knob_uint("MI355_JPEG_TAPER", 0, 95, &kv)) {  // scheduling only; 0 = equal parts
        if (kv && kv < 40) bad = true;
        else k->taper = (int)kv;
    }
    if (knob_uint("MI355_JPEG_SCREEN_WAVES", 32, 8192, &kv)) {
        if (kv & 31) bad = true;
        else k->screen_waves = (uint32_t)kv;
    }
    if (knob_uint("MI355_JPEG_STAGGER", 0, 64, &kv)) k->stagger = (uint32_t)kv;  // timing experiment: bounded, a sleep loop in the kernel
    return !bad;
}

int status_to_error(uint32_t st) {
    if (st & 4u) return MI355_E_INTERNAL;
    if (st & 1u) return MI355_E_CATEGORY;
    if (st & 2u) return MI355_E_CAPACITY;
    return MI355_OK;
}

// No exception crosses the C ABI (SURVEY §8 b; the reference's counterpart is -1 + a message, utils.cpp:17-63): every
// entry point that can allocate on the host (std::vector, std::thread, push_back) runs inside this barrier.
template <typename F>
int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return MI355_E_ALLOC;
    } catch (const std::system_error&) {  // a worker thread could not be started
        return MI355_E_ALLOC;
    } catch (...) {
        return MI355_E_INTERNAL;
    }
}

}  // namespace

extern "C" {

int mi355_jpeg_abi_version(void) { return MI355_JPEG_ABI_VERSION; }

const char* mi355_jpeg_strerror(int status) {
    switch (status) {
        case MI355_OK: return "ok";
        case MI355_E_ARG: return "invalid argument";
        case MI355_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case MI355_E_CAPACITY: return "output buffer too small";
        case MI355_E_CATEGORY: return "coefficient size category outside the Huffman tables";
        case MI355_E_ALLOC: return "allocation failed";
        case MI355_E_TABLE: return "malformed table (or quantiser entries > 255 in a JFIF container)";
        case MI355_E_INTERNAL: return "internal error: a device-side wait gave up";
        default: break;
    }
    if (status <= MI355_E_HIP) return hipGetErrorString((hipError_t)(MI355_E_HIP - status));
    return "unknown error";
}

int mi355_jpeg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_body(int device_id, mi355_jpeg_ctx** out){
    if (!out) return MI355_E_ARG;
    *out = nullptr;
    Knobs kn;
    if (!read_knobs(&kn)) return MI355_E_ARG;  // before anything touches a device: testable anywhere
    int n = mi355_jpeg_device_count();
    if (n <= 0 || device_id < 0 || device_id >= n) return MI355_E_NO_DEVICE;
    HIP_TRY(hipSetDevice(device_id));
    mi355_jpeg_ctx* c = new (std::nothrow) mi355_jpeg_ctx();
    if (!c) return MI355_E_ALLOC;
    c->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess) c->n_cus = cus;
        if (c->n_cus > 0) c->screen_waves = 8u * (uint32_t)c->n_cus;  // two 4-wave workgroups per CU (2048 on MI355X)
    }
    for (int i = 0; i < 64; ++i) c->qlum[i] = kQ50Lum[i], c->qchrom[i] = kQ50Chr[i];
    for (int t = 0; t < 4; ++t) reference_huffman(t, &c->huff[t]), reference_huffman(t, &c->huff_std[t], false);
    c->transform_mode = kn.transform_mode;
    c->emit_lds_words = kn.emit_lds_words;
    c->tau_scale = kn.tau_scale;
    c->batch_parts = kn.batch_parts;
    c->max_sets = kn.max_sets;
    c->taper = kn.taper;
    c->stagger = kn.stagger;
    if (kn.screen_waves) (void)mi355_jpeg_set_encode_waves(c, kn.screen_waves);  // one place derives every grid from it
    int e = MI355_OK;
    if (hipMalloc((void**)&c->d_q, 128 * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->d_qzz, 128 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_lut, 2048 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_status, sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_afrag, 2 * kAfragBytes + (size_t)(kCscSets + kCscStrictSets) * 1024) != hipSuccess ||
        hipMalloc((void**)&c->d_qconst, 512 * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->d_qconst_f, 512 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->d_lut2, 4 * 66 * 16 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_counters, 64 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_stats, 4 * sizeof(unsigned long long)) != hipSuccess)
        e = MI355_E_ALLOC;
    if (!e) e = hip_err(hipMemset(c->d_stats, 0, 4 * sizeof(unsigned long long)));
    if (!e) e = hip_err(hipMemset(c->d_status, 0, sizeof(uint32_t)));
    if (!e) e = hip_err(hipMemset(c->d_counters, 0, 64 * sizeof(uint32_t)));
    if (!e) e = upload_afrag(c);
    if (!e) e = upload_csc_frag(c);
    if (!e) e = upload_tables(c);
    if (!e) e = hip_err(hipDeviceSynchronize());  // every fill and table copy above has landed
    if (e) {
        mi355_jpeg_destroy(c);
        return e;
    }
    *out = c;
    return MI355_OK;
}
int mi355_jpeg_create(int device_id, mi355_jpeg_ctx** out) {
    return guarded([&]() -> int { return create_body(device_id, out); });
}

void mi355_jpeg_destroy(mi355_jpeg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_half) (void)hipEventDestroy(c->ev_half);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    for (hipEvent_t ev : c->ev_set)
        if (ev) (void)hipEventDestroy(ev);
    void* ptrs[] = {c->d_q,        c->d_lut,      c->d_status, c->d_coefs,  c->d_unit_off, c->d_tile_bits,
                    c->d_tile_off, c->d_in,       c->d_out,    c->d_bits,   c->d_afrag,    c->d_qconst,
                    c->d_counters, c->d_meta,     c->d_arena,  c->d_lut2,     c->d_qconst_f,
                    c->d_stuff_counts, c->d_stuff_offs, c->d_qzz, c->d_stats};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (void* p : c->d_stage)
        if (p) (void)hipFree(p);
    for (auto& es : c->ev_pool)
        for (auto& e : es.e)
            if (e) (void)hipEventDestroy(e);
    delete c;
}

static int set_quant_body(mi355_jpeg_ctx* c, const uint32_t qlum[64], const uint32_t qchrom[64]){
    if (!c || !qlum || !qchrom) return MI355_E_ARG;
    for (int i = 0; i < 64; ++i)
        if (qlum[i] < 1 || qlum[i] > 65535 || qchrom[i] < 1 || qchrom[i] > 65535) return MI355_E_TABLE;
    memcpy(c->qlum, qlum, sizeof c->qlum);
    memcpy(c->qchrom, qchrom, sizeof c->qchrom);
    HIP_TRY(hipSetDevice(c->device));
    return upload_tables(c);
}
int mi355_jpeg_set_quant(mi355_jpeg_ctx* c, const uint32_t qlum[64], const uint32_t qchrom[64]) {
    return guarded([&]() -> int { return set_quant_body(c, qlum, qchrom); });
}

static int set_quality_body(mi355_jpeg_ctx* c, int quality){
    if (!c || quality < 1 || quality > 100) return MI355_E_ARG;
    uint32_t ql[64], qc[64];
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        long a = ((long)kQ50Lum[i] * scale + 50) / 100, b = ((long)kQ50Chr[i] * scale + 50) / 100;
        ql[i] = (uint32_t)(a < 1 ? 1 : a > 255 ? 255 : a);
        qc[i] = (uint32_t)(b < 1 ? 1 : b > 255 ? 255 : b);
    }
    return mi355_jpeg_set_quant(c, ql, qc);
}
int mi355_jpeg_set_quality(mi355_jpeg_ctx* c, int quality) {
    return guarded([&]() -> int { return set_quality_body(c, quality); });
}

static int set_huffman_body(mi355_jpeg_ctx* c, int table, const mi355_huff_table* t){
    if (!c || table < 0 || table > 3) return MI355_E_ARG;
    mi355_huff_table nt;
    if (t) {
        nt = *t;
        for (int i = 0; i < 256; ++i) {
            if (nt.len[i] > 17) return MI355_E_TABLE;
            if (nt.len[i] && (nt.code[i] >> nt.len[i])) return MI355_E_TABLE;
        }
        if (table >= 2 && (!nt.len[0x00] || !nt.len[0xF0])) return MI355_E_TABLE;  // EOB and ZRL are always needed
        c->huff_std[table] = nt;
    } else {
        reference_huffman(table, &nt);
        reference_huffman(table, &c->huff_std[table], false);
    }
    c->huff[table] = nt;
    HIP_TRY(hipSetDevice(c->device));
    return upload_tables(c);
}
int mi355_jpeg_set_huffman(mi355_jpeg_ctx* c, int table, const mi355_huff_table* t) {
    return guarded([&]() -> int { return set_huffman_body(c, table, t); });
}

static int set_encode_waves_body(mi355_jpeg_ctx* c, uint32_t waves){
    if (!c || (waves != 0 && (waves < 32 || waves > 8192 || (waves & 31)))) return MI355_E_ARG;
    c->screen_waves = waves ? waves : (c->n_cus > 0 ? 8u * (uint32_t)c->n_cus : 2048u);
    return MI355_OK;
}
int mi355_jpeg_set_encode_waves(mi355_jpeg_ctx* c, uint32_t waves) {
    return guarded([&]() -> int { return set_encode_waves_body(c, waves); });
}

static int reference_huffman_body(int table, mi355_huff_table* t){
    if (!t || table < 0 || table > 3) return MI355_E_ARG;
    reference_huffman(table, t);
    return MI355_OK;
}
int mi355_jpeg_reference_huffman(int table, mi355_huff_table* t) {
    return guarded([&]() -> int { return reference_huffman_body(table, t); });
}

static int get_quant_body(mi355_jpeg_ctx* c, uint32_t qlum[64], uint32_t qchrom[64]){
    if (!c || !qlum || !qchrom) return MI355_E_ARG;
    memcpy(qlum, c->qlum, sizeof c->qlum);
    memcpy(qchrom, c->qchrom, sizeof c->qchrom);
    return MI355_OK;
}
int mi355_jpeg_get_quant(mi355_jpeg_ctx* c, uint32_t qlum[64], uint32_t qchrom[64]) {
    return guarded([&]() -> int { return get_quant_body(c, qlum, qchrom); });
}

static int get_huffman_body(mi355_jpeg_ctx* c, int table, mi355_huff_table* t){
    if (!c || !t || table < 0 || table > 3) return MI355_E_ARG;
    *t = c->huff[table];
    return MI355_OK;
}
int mi355_jpeg_get_huffman(mi355_jpeg_ctx* c, int table, mi355_huff_table* t) {
    return guarded([&]() -> int { return get_huffman_body(c, table, t); });
}

void mi355_jpeg_padded_size(uint32_t W, uint32_t H, uint32_t* W8, uint32_t* H8) {
    if (W8) *W8 = (W + 7) / 8 * 8;
    if (H8) *H8 = (H + 7) / 8 * 8;
}

size_t mi355_jpeg_scan_bound(uint32_t W, uint32_t H) { return mi355_jpeg_scan_bound_flags(W, H, 0); }

size_t mi355_jpeg_scan_bound_flags(uint32_t W, uint32_t H, uint32_t flags) {
    // per unit at most: DC 11+11 bits, 63 x (17+10) AC bits, EOB 4 -> 1727 bits
    const bool s420 = (flags & MI355_F_420) != 0;
    const uint32_t A = s420 ? 16 : 8;
    const size_t mcus = (size_t)((W + A - 1) / A) * ((H + A - 1) / A);
    const size_t units = mcus * (s420 ? 6 : 3);
    size_t bits = units * 1727;
    if (flags & MI355_F_RESTART) bits += ((mcus + 63) / 64) * 7;  // every interval (64 MCUs) padded to a byte
    return (bits + 7) / 8 + 8;
}

static int encode_scan_device_body(mi355_jpeg_ctx* c, const void* d_rgb, uint32_t W, uint32_t H,
                                  uint32_t n_frames, uint32_t flags, void* d_out, size_t out_stride,
                                  uint64_t* d_bits, void* stream){
    if (!c || !d_rgb || !d_out || !d_bits || n_frames == 0 || out_stride < 8 || (out_stride & 3) ||
        ((uintptr_t)d_out & 3))
        return MI355_E_ARG;
    Geom g;
    int e = make_geom(W, H, flags, d_rgb, &g);
    if (e) return e;
    if (n_frames > 65535u) return MI355_E_ARG;
    if ((flags & MI355_F_STANDARD) && c->transform_mode != 2) return MI355_E_ARG;  // the exact pipeline is strict only
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    if (c->transform_mode == 2) {
        // only unit_off / tile arrays of the classic workspace are needed
        if ((e = ensure(c->d_unit_off, c->unit_off_cap, unit_off_words(g) * n_frames))) return e;
        if ((e = ensure(c->d_tile_bits, c->tiles_cap, (size_t)g.tiles * n_frames, true))) return e;
        if ((e = ensure(c->d_tile_off, c->tile_off_cap, tile_off_entries(g, n_frames)))) return e;
        return run_screened(c, g, n_frames, (const uint8_t*)d_rgb, (uint8_t*)d_out, out_stride, d_bits, s);
    }
    if ((e = ensure_workspace(c, g, n_frames))) return e;
    record(c, 0, s);
    c->last_launches = 1;
    HIP_TRY(launch_transform(g, n_frames, (const uint8_t*)d_rgb, c->d_q, c->d_coefs, c->transform_mode, s));
    record(c, 1, s);
    return run_entropy(c, g, n_frames, (uint8_t*)d_out, out_stride, d_bits, s);
}
int mi355_jpeg_encode_scan_device(mi355_jpeg_ctx* c, const void* d_rgb, uint32_t W, uint32_t H,
                                  uint32_t n_frames, uint32_t flags, void* d_out, size_t out_stride,
                                  uint64_t* d_bits, void* stream) {
    return guarded([&]() -> int { return encode_scan_device_body(c, d_rgb, W, H, n_frames, flags, d_out, out_stride, d_bits, stream); });
}

static int sync_body(mi355_jpeg_ctx* c, void* stream){
    if (!c) return MI355_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
#ifdef MI355_STAMPS
    if (c->d_stamps && getenv("MI355_JPEG_DUMP_STAMPS")) {
        std::vector<unsigned long long> h(4096 * 8);
        (void)hipMemcpy(h.data(), c->d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        for (int w = 0; w < 2048; ++w)
            for (int i = 0; i < 8; ++i) sum[i] += (double)h[(size_t)w * 8 + i];
        fprintf(stderr, "[stamps] cycles per wave: setup %.0f | samples %.0f | mfma+quantise %.0f | dc0+barrier %.0f | walk %.0f | dc/sizes %.0f | arena+meta %.0f | loop %.0f\n",
                sum[0] / 2048, sum[5] / 2048, sum[6] / 2048, sum[1] / 2048, sum[2] / 2048, sum[3] / 2048, sum[4] / 2048, sum[7] / 2048);
        // wave start / end distribution (10 ns ticks of s_memrealtime) relative to the first start
        std::vector<double> t0s, t1s;
        unsigned long long first = ~0ull;
        for (int w = 0; w < 2048; ++w) first = h[(size_t)(2048 + w) * 8] < first ? h[(size_t)(2048 + w) * 8] : first;
        for (int w = 0; w < 2048; ++w) {
            t0s.push_back((double)(h[(size_t)(2048 + w) * 8] - first) * 0.01);
            t1s.push_back((double)(h[(size_t)(2048 + w) * 8 + 1] - first) * 0.01);
        }
        {   // mean wave end per XCD (workgroup id % 8) and per wave slot of the workgroup
            double ex[8] = {0}, ew[4] = {0}, dur[8] = {0};
            for (int w = 0; w < 2048; ++w) {
                ex[(w / 4) % 8] += t1s[(size_t)w] / 256.0;
                dur[(w / 4) % 8] += (t1s[(size_t)w] - t0s[(size_t)w]) / 256.0;
                ew[w % 4] += t1s[(size_t)w] / 512.0;
            }
            fprintf(stderr, "[stamps] mean end per XCD: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f | per wave slot: %.1f %.1f %.1f %.1f\n",
                    ex[0], ex[1], ex[2], ex[3], ex[4], ex[5], ex[6], ex[7], ew[0], ew[1], ew[2], ew[3]);
            {   // by first channel of the wave (local % 3), by workgroup half, by workgroup index within the XCD
                double e3[3] = {0}, n3[3] = {0}, eh[2] = {0}, nh[2] = {0};
                int late_by_wgx[64] = {0};
                for (int w = 0; w < 2048; ++w) {
                    const int wg = w / 4, local = (wg / 8) * 4 + w % 4;
                    e3[local % 3] += t1s[(size_t)w], n3[local % 3] += 1;
                    eh[wg >= 256] += t1s[(size_t)w], nh[wg >= 256] += 1;
                    if (t1s[(size_t)w] > 46.0) late_by_wgx[wg / 8]++;
                }
                fprintf(stderr, "[stamps] mean end by first channel: %.1f %.1f %.1f | by workgroup half: %.1f %.1f\n[stamps] late waves (>46 us) by workgroup index within XCD (of 32 each):",
                        e3[0] / n3[0], e3[1] / n3[1], e3[2] / n3[2], eh[0] / nh[0], eh[1] / nh[1]);
                for (int i = 0; i < 64; ++i) fprintf(stderr, " %d", late_by_wgx[i]);
                fprintf(stderr, "\n");
            }
            // histogram of end times, 2 us bins from 28 us
            int hist[16] = {0};
            for (double e : t1s) {
                int b = (int)((e - 28.0) / 2.0);
                hist[b < 0 ? 0 : b > 15 ? 15 : b]++;
            }
            fprintf(stderr, "[stamps] end histogram (2 us bins from 28 us):");
            for (int b = 0; b < 16; ++b) fprintf(stderr, " %d", hist[b]);
            fprintf(stderr, "\n");
        }
        std::sort(t0s.begin(), t0s.end());
        std::sort(t1s.begin(), t1s.end());
        fprintf(stderr, "[stamps] wave start us: p50 %.2f p90 %.2f max %.2f | wave end us: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                t0s[1024], t0s[1843], t0s[2047], t1s[0], t1s[204], t1s[1024], t1s[1843], t1s[2047]);
    }
#endif
    uint32_t st = 0;
    HIP_TRY(hipMemcpy(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        HIP_TRY(hipMemset(c->d_status, 0, sizeof st));
        HIP_TRY(hipStreamSynchronize(nullptr));  // the clear has landed before the next call's kernels look
    }
    return status_to_error(st);
}
int mi355_jpeg_sync(mi355_jpeg_ctx* c, void* stream) {
    return guarded([&]() -> int { return sync_body(c, stream); });
}

static int encode_scan_body(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H,
                           uint32_t n_frames, uint32_t flags, uint8_t* out, size_t out_stride,
                           uint64_t* bits){
    if (!c || !rgb || !out || !bits || n_frames == 0) return MI355_E_ARG;
    Geom g;
    int e = make_geom(W, H, flags, nullptr, &g);
    if (e) return e;
    HIP_TRY(hipSetDevice(c->device));
    size_t in_bytes = (size_t)g.frame_stride * n_frames;
    size_t dstride = (out_stride + 3) & ~(size_t)3;
    if (dstride < 8) dstride = 8;
    if ((e = ensure(c->d_in, c->in_cap, in_bytes))) return e;
    if ((e = ensure(c->d_out, c->out_cap, dstride * n_frames))) return e;
    if ((e = ensure(c->d_bits, c->bits_cap, (size_t)n_frames))) return e;
    HIP_TRY(hipMemcpyAsync(c->d_in, rgb, in_bytes, hipMemcpyHostToDevice, nullptr));
    // the device-side capacity check works on whole words; give it the caller's real limit
    e = mi355_jpeg_encode_scan_device(c, c->d_in, W, H, n_frames, flags, c->d_out, dstride, c->d_bits, nullptr);
    if (e) return e;
    if ((e = mi355_jpeg_sync(c, nullptr))) return e;
    HIP_TRY(hipMemcpy(bits, c->d_bits, sizeof(uint64_t) * n_frames, hipMemcpyDeviceToHost));
    for (uint32_t f = 0; f < n_frames; ++f) {
        size_t nb = (size_t)((bits[f] + 7) / 8);
        if (nb > out_stride) return MI355_E_CAPACITY;
        HIP_TRY(hipMemcpy(out + (size_t)f * out_stride, c->d_out + (size_t)f * dstride, nb, hipMemcpyDeviceToHost));
    }
    return MI355_OK;
}
int mi355_jpeg_encode_scan(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H,
                           uint32_t n_frames, uint32_t flags, uint8_t* out, size_t out_stride,
                           uint64_t* bits) {
    return guarded([&]() -> int { return encode_scan_body(c, rgb, W, H, n_frames, flags, out, out_stride, bits); });
}

// ---- stage probes -----------------------------------------------------------

static int probe_samples_body(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                             uint8_t* out){
    if (!c || !rgb || !out) return MI355_E_ARG;
    Geom g;
    int e = make_geom(W, H, flags, nullptr, &g);
    if (e) return e;
    HIP_TRY(hipSetDevice(c->device));
    size_t ob = (size_t)g.W8 * g.H8 * 3;
    if ((e = ensure(c->d_in, c->in_cap, (size_t)g.frame_stride))) return e;
    if ((e = ensure(c->d_out, c->out_cap, ob))) return e;
    HIP_TRY(hipMemcpy(c->d_in, rgb, g.frame_stride, hipMemcpyHostToDevice));
    if ((flags & MI355_F_STANDARD) && c->transform_mode != 2) return MI355_E_ARG;
    if (flags & MI355_F_420) return MI355_E_ARG;  // no full-resolution chroma planes exist in 4:2:0
    if (c->transform_mode == 2) {
        // the screened pipeline has its own (integer-exact) sample stage: probe that one
        if ((e = ensure_workspace(c, g, 1))) return e;
        if ((e = run_screened_probe(c, g, c->d_in, c->d_out, nullptr))) return e;
    } else {
        HIP_TRY(launch_probe_samples(g, c->d_in, c->d_out, nullptr));
    }
    HIP_TRY(hipMemcpy(out, c->d_out, ob, hipMemcpyDeviceToHost));
    return MI355_OK;
}
int mi355_jpeg_probe_samples(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                             uint8_t* out) {
    return guarded([&]() -> int { return probe_samples_body(c, rgb, W, H, flags, out); });
}

static int transform_to_workspace(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H,
                                  uint32_t flags, Geom* g) {
    int e = make_geom(W, H, flags, nullptr, g);
    if (e) return e;
    if ((flags & MI355_F_STANDARD) && c->transform_mode != 2) return MI355_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if ((e = ensure(c->d_in, c->in_cap, (size_t)g->frame_stride))) return e;
    if ((e = ensure_workspace(c, *g, 1))) return e;
    HIP_TRY(hipMemcpy(c->d_in, rgb, g->frame_stride, hipMemcpyHostToDevice));
    if (c->transform_mode == 2) return run_screened_probe(c, *g, c->d_in, nullptr, nullptr);
    HIP_TRY(launch_transform(*g, 1, c->d_in, c->d_q, c->d_coefs, c->transform_mode, nullptr));
    return MI355_OK;
}

static int probe_coefficients_body(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H,
                                  uint32_t flags, int16_t* out){
    if (!c || !rgb || !out) return MI355_E_ARG;
    Geom g;
    int e = transform_to_workspace(c, rgb, W, H, flags, &g);
    if (e) return e;
    size_t ob = unit_count(g) * 64 * sizeof(int16_t);
    if ((e = ensure(c->d_out, c->out_cap, ob))) return e;
    HIP_TRY(launch_coefs_to_rows(g, c->d_coefs, (int16_t*)c->d_out, nullptr));
    HIP_TRY(hipMemcpy(out, c->d_out, ob, hipMemcpyDeviceToHost));
    return MI355_OK;
}
int mi355_jpeg_probe_coefficients(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H,
                                  uint32_t flags, int16_t* out) {
    return guarded([&]() -> int { return probe_coefficients_body(c, rgb, W, H, flags, out); });
}

static int probe_unit_bits_body(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                               uint32_t* out){
    if (!c || !rgb || !out) return MI355_E_ARG;
    if (flags & MI355_F_STANDARD) return MI355_E_ARG;  // the per-unit size kernel codes the reference's rules only
    Geom g;
    int e = transform_to_workspace(c, rgb, W, H, flags, &g);
    if (e) return e;
    size_t ob = (size_t)g.N * 3 * sizeof(uint32_t);
    if ((e = ensure(c->d_out, c->out_cap, ob))) return e;
    HIP_TRY(launch_unit_sizes(g, 1, c->d_coefs, c->d_lut, c->d_unit_off, c->d_tile_bits, c->d_status, nullptr));
    HIP_TRY(launch_unit_bits(g, c->d_unit_off, c->d_tile_bits, (uint32_t*)c->d_out, nullptr));
    HIP_TRY(hipMemsetAsync(c->d_tile_bits, 0, (size_t)g.tiles * sizeof(uint32_t), nullptr));
    if ((e = mi355_jpeg_sync(c, nullptr))) return e;
    HIP_TRY(hipMemcpy(out, c->d_out, ob, hipMemcpyDeviceToHost));
    return MI355_OK;
}
int mi355_jpeg_probe_unit_bits(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                               uint32_t* out) {
    return guarded([&]() -> int { return probe_unit_bits_body(c, rgb, W, H, flags, out); });
}

static int entropy_only_body(mi355_jpeg_ctx* c, const int16_t* zigzag, uint32_t n_blocks, uint8_t* out,
                            size_t cap, uint64_t* bits){
    if (!c || !zigzag || !out || !bits || n_blocks == 0) return MI355_E_ARG;
    Geom g;
    memset(&g, 0, sizeof g);
    g.N = n_blocks;
    g.passes = 3;
    g.tiles = (n_blocks + 63) / 64;
    HIP_TRY(hipSetDevice(c->device));
    int e;
    size_t ib = (size_t)n_blocks * 3 * 64 * sizeof(int16_t);
    size_t dstride = (cap + 3) & ~(size_t)3;
    if (dstride < 8) dstride = 8;
    if ((e = ensure(c->d_in, c->in_cap, ib))) return e;
    if ((e = ensure(c->d_out, c->out_cap, dstride))) return e;
    if ((e = ensure(c->d_bits, c->bits_cap, (size_t)1))) return e;
    if ((e = ensure_workspace(c, g, 1))) return e;
    HIP_TRY(hipMemcpy(c->d_in, zigzag, ib, hipMemcpyHostToDevice));
    HIP_TRY(launch_rows_to_coefs(g, (const int16_t*)c->d_in, c->d_coefs, nullptr));
    if ((e = run_entropy(c, g, 1, c->d_out, dstride, c->d_bits, nullptr))) return e;
    if ((e = mi355_jpeg_sync(c, nullptr))) return e;
    HIP_TRY(hipMemcpy(bits, c->d_bits, sizeof(uint64_t), hipMemcpyDeviceToHost));
    size_t nb = (size_t)((*bits + 7) / 8);
    if (nb > cap) return MI355_E_CAPACITY;
    HIP_TRY(hipMemcpy(out, c->d_out, nb, hipMemcpyDeviceToHost));
    return MI355_OK;
}
int mi355_jpeg_entropy_only(mi355_jpeg_ctx* c, const int16_t* zigzag, uint32_t n_blocks, uint8_t* out,
                            size_t cap, uint64_t* bits) {
    return guarded([&]() -> int { return entropy_only_body(c, zigzag, n_blocks, out, cap, bits); });
}

// ---- JFIF framing (build-defined) ----------------------

namespace {
struct Writer {
    uint8_t* p;
    size_t n, cap;
    void b(unsigned v) {
        if (n < cap) p[n] = (uint8_t)v;
        ++n;
    }
    void w(unsigned v) {
        b(v >> 8);
        b(v & 255);
    }
};
// BITS/HUFFVAL of a table, by sorting its codes by (length, code).  Codes longer
// than 16 bits (the reference's seven 17-bit entries) cannot be expressed in a DHT
// segment: they are listed at 16 bits, so strict-mode files are parity artefacts,
// not guaranteed-decodable pictures.
void dht_segment(Writer& w, int cls_id, const mi355_huff_table& t) {
    uint8_t bits[16] = {0};
    std::vector<uint8_t> vals;
    for (int l = 1; l <= 16; ++l) {
        std::vector<std::pair<uint32_t, int>> at;
        for (int i = 0; i < 256; ++i) {
            int li = t.len[i];
            uint32_t ci = t.code[i];
            if (li == 17) li = 16, ci &= 0xFFFFu;  // listed without the extra leading '1'
            if (li == l) at.push_back({ci, i});
        }
        for (size_t a = 0; a < at.size(); ++a)
            for (size_t b2 = a + 1; b2 < at.size(); ++b2)
                if (at[b2].first < at[a].first) std::swap(at[a], at[b2]);
        for (auto& pr : at) {
            bits[l - 1]++;
            vals.push_back((uint8_t)pr.second);
        }
    }
    w.w(0xFFC4);
    w.w((unsigned)(2 + 1 + 16 + vals.size()));
    w.b(cls_id);
    for (int i = 0; i < 16; ++i) w.b(bits[i]);
    for (uint8_t v : vals) w.b(v);
}
// the container writes 8-bit DQT segments: quantiser entries above 255 (set_quant takes up to 65535 for the
// scan entry points) cannot be expressed and would silently decode with other tables
bool dqt_fits(const mi355_jpeg_ctx* c) {
    for (int i = 0; i < 64; ++i)
        if (c->qlum[i] > 255 || c->qchrom[i] > 255) return false;
    return true;
}
// header of the build-defined container: SOI, APP0, DQT x2, SOF0, DHT x4, SOS; returns its length
size_t jfif_header(const mi355_jpeg_ctx* c, uint32_t W, uint32_t H, uint32_t flags, uint8_t* dst, size_t cap) {
    static const uint8_t zz[64] = MI355_ZIGZAG_TABLE;
    Writer w{dst, 0, cap};
    w.w(0xFFD8);
    w.w(0xFFE0), w.w(16);
    w.b('J'), w.b('F'), w.b('I'), w.b('F'), w.b(0);
    w.w(0x0101), w.b(0), w.w(1), w.w(1), w.b(0), w.b(0);
    for (int t = 0; t < 2; ++t) {
        const uint32_t* q = t ? c->qchrom : c->qlum;
        w.w(0xFFDB), w.w(67), w.b(t);
        for (int k = 0; k < 64; ++k) w.b(q[zz[k]] > 255 ? 255 : q[zz[k]]);
    }
    w.w(0xFFC0), w.w(17), w.b(8), w.w(H), w.w(W), w.b(3);
    w.b(1), w.b((flags & MI355_F_420) ? 0x22 : 0x11), w.b(0);
    w.b(2), w.b(0x11), w.b(1);
    w.b(3), w.b(0x11), w.b(1);
    const mi355_huff_table* ht = (flags & MI355_F_STANDARD) ? c->huff_std : c->huff;
    dht_segment(w, 0x00, ht[0]);
    dht_segment(w, 0x10, ht[2]);
    dht_segment(w, 0x01, ht[1]);
    dht_segment(w, 0x11, ht[3]);
    if (flags & MI355_F_RESTART) w.w(0xFFDD), w.w(4), w.w(64);  // DRI: one interval = one 64-MCU tile
    w.w(0xFFDA), w.w(12), w.b(3);
    w.b(1), w.b(0x00), w.b(2), w.b(0x11), w.b(3), w.b(0x11);
    w.b(0), w.b(63), w.b(0);
    return w.n;
}
}  // namespace

static int wrap_jfif_body(mi355_jpeg_ctx* c, const uint8_t* scan, uint64_t n_bits, uint32_t W, uint32_t H, uint32_t flags,
                         uint8_t* out, size_t cap, size_t* out_len){
    if (!c || !scan || !out || !out_len || W == 0 || H == 0 || W > 65535u || H > 65535u) return MI355_E_ARG;
    if ((flags & MI355_F_420) && !(flags & MI355_F_STANDARD)) return MI355_E_ARG;
    if (flags & MI355_F_RESTART) return MI355_E_ARG;  // the markers go in with the stuffing, on the device
    if (!dqt_fits(c)) return MI355_E_TABLE;
    Writer w{out, jfif_header(c, W, H, flags, out, cap), cap};
    const size_t nb = (size_t)((n_bits + 7) / 8);
    for (size_t i = 0; i < nb; ++i) {  // entropy bytes: last partial byte padded with 1s, 0xFF -> 0xFF 0x00
        unsigned b = scan[i];
        if (i == nb - 1 && (n_bits & 7)) b |= 0xFFu >> (n_bits & 7);
        w.b(b);
        if (b == 0xFF) w.b(0);
    }
    w.w(0xFFD9);
    *out_len = w.n;
    return w.n > cap ? MI355_E_CAPACITY : MI355_OK;
}
int mi355_jpeg_wrap_jfif(mi355_jpeg_ctx* c, const uint8_t* scan, uint64_t n_bits, uint32_t W, uint32_t H, uint32_t flags,
                         uint8_t* out, size_t cap, size_t* out_len) {
    return guarded([&]() -> int { return wrap_jfif_body(c, scan, n_bits, W, H, flags, out, cap, out_len); });
}

static int encode_jfif_body(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                           uint8_t* out, size_t cap, size_t* out_len){
    if (!c || !rgb || !out || !out_len) return MI355_E_ARG;
    Geom g;
    int e = make_geom(W, H, flags, nullptr, &g);
    if (e) return e;
    if (!dqt_fits(c)) return MI355_E_TABLE;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint8_t> hdr(1024);
    const size_t hlen = jfif_header(c, W, H, flags, hdr.data(), hdr.size());
    if (hlen + 2 > cap) return MI355_E_CAPACITY;

    // scan on the device, stuffed on the device (k_stuff_*), one copy back
    size_t scan_cap = (size_t)W * H + 4096;  // 8 bits per pixel first; the worst-case bound on overflow
    const size_t bound = mi355_jpeg_scan_bound_flags(W, H, flags);
    if (scan_cap > bound) scan_cap = bound;
    if ((e = ensure(c->d_in, c->in_cap, (size_t)g.frame_stride))) return e;
    if ((e = ensure(c->d_bits, c->bits_cap, (size_t)2))) return e;
    HIP_TRY(hipMemcpy(c->d_in, rgb, g.frame_stride, hipMemcpyHostToDevice));
    for (int attempt = 0; attempt < 2; ++attempt) {
        scan_cap = (scan_cap + 3) & ~(size_t)3;
        const size_t stuffed_cap = 2 * scan_cap + 16;
        if ((e = ensure(c->d_out, c->out_cap, scan_cap + stuffed_cap))) return e;
        e = mi355_jpeg_encode_scan_device(c, c->d_in, W, H, 1, flags, c->d_out, scan_cap, c->d_bits, nullptr);
        if (!e) e = stuff_scan(c, c->d_out, c->d_bits, scan_cap, c->d_out + scan_cap, stuffed_cap, c->d_bits + 1,
                               (flags & MI355_F_RESTART) ? c->d_tile_off : nullptr, g.tiles, nullptr);
        if (!e) e = mi355_jpeg_sync(c, nullptr);
        if (e == MI355_E_CAPACITY && attempt == 0 && scan_cap < bound) {
            scan_cap = bound;
            continue;
        }
        if (e) return e;
        uint64_t slen = 0;
        HIP_TRY(hipMemcpy(&slen, c->d_bits + 1, sizeof slen, hipMemcpyDeviceToHost));
        *out_len = hlen + (size_t)slen + 2;
        if (*out_len > cap) return MI355_E_CAPACITY;
        memcpy(out, hdr.data(), hlen);
        HIP_TRY(hipMemcpy(out + hlen, c->d_out + scan_cap, (size_t)slen, hipMemcpyDeviceToHost));
        out[hlen + slen] = 0xFF;
        out[hlen + slen + 1] = 0xD9;
        return MI355_OK;
    }
    return MI355_E_CAPACITY;
}
int mi355_jpeg_encode_jfif(mi355_jpeg_ctx* c, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t flags,
                           uint8_t* out, size_t cap, size_t* out_len) {
    return guarded([&]() -> int { return encode_jfif_body(c, rgb, W, H, flags, out, cap, out_len); });
}

// ---- either side of the path ------------------------------------------------------------

static int synth_lcg_device_body(mi355_jpeg_ctx* c, void* d_dst, size_t frame_bytes, uint32_t n_frames, uint32_t seed0,
                                void* stream){
    if (!c || !d_dst || frame_bytes == 0 || n_frames == 0 || n_frames > 65535u) return MI355_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_lcg_fill((uint8_t*)d_dst, frame_bytes, n_frames, seed0, (hipStream_t)stream));
    return MI355_OK;
}
int mi355_jpeg_synth_lcg_device(mi355_jpeg_ctx* c, void* d_dst, size_t frame_bytes, uint32_t n_frames, uint32_t seed0,
                                void* stream) {
    return guarded([&]() -> int { return synth_lcg_device_body(c, d_dst, frame_bytes, n_frames, seed0, stream); });
}

static int stuff_device_body(mi355_jpeg_ctx* c, const void* d_scan, const uint64_t* d_bits, size_t max_scan_bytes,
                            void* d_out, size_t cap, uint64_t* d_out_len, void* stream){
    return stuff_scan(c, d_scan, d_bits, max_scan_bytes, d_out, cap, d_out_len, nullptr, 0, stream);
}
int mi355_jpeg_stuff_device(mi355_jpeg_ctx* c, const void* d_scan, const uint64_t* d_bits, size_t max_scan_bytes,
                            void* d_out, size_t cap, uint64_t* d_out_len, void* stream) {
    return guarded([&]() -> int { return stuff_device_body(c, d_scan, d_bits, max_scan_bytes, d_out, cap, d_out_len, stream); });
}

}  // extern "C"
namespace {
int stuff_scan(mi355_jpeg_ctx* c, const void* d_scan, const uint64_t* d_bits, size_t max_scan_bytes, void* d_out,
               size_t cap, uint64_t* d_out_len, const uint64_t* d_tile_off, uint32_t tiles, void* stream) {
    if (!c || !d_scan || !d_bits || !d_out || !d_out_len || max_scan_bytes == 0) return MI355_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const size_t chunks = (max_scan_bytes + 4095) / 4096;
    int e;
    if ((e = ensure(c->d_stuff_counts, c->stuff_cap, chunks))) return e;
    if ((e = ensure(c->d_stuff_offs, c->stuff_offs_cap, chunks + 1))) return e;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_stuff((const uint8_t*)d_scan, d_bits, max_scan_bytes, c->d_stuff_counts, c->d_stuff_offs,
                         c->d_stuff_offs + chunks, (uint8_t*)d_out, cap, c->d_status, d_tile_off, tiles, s));
    HIP_TRY(hipMemcpyAsync(d_out_len, c->d_stuff_offs + chunks, sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    return MI355_OK;
}
}  // namespace
extern "C" {

// ---- multi-GPU batch driver ----------------------------------------------------------
//
// Frames are independent (DC predictors start at 0 per frame, utils.cpp:665), so a batch
// shards over GPUs with no exchange at all.  Worker w owns frames [lo_w, hi_w) and moves them
// through its GPU in chunks: three streams, double-buffered device memory, host memory registered
// with HIP so that H2D / D2H are true DMA.  Everything a worker needs lives in the pool and is made
// once: its thread (pinned to the CPUs of the GPU's NUMA node), context, streams, events and device
// buffers (grown on demand, never shrunk).  Host ranges the caller registers with
// mi355_jpeg_pool_register stay registered across calls; memory that is not is registered for the
// duration of one call, like before.

}  // extern "C"

#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

namespace {

struct PoolJob {
    const uint8_t* rgb = nullptr;
    uint32_t W = 0, H = 0, n = 0, chunk = 1, flags = 0;
    std::atomic<uint32_t>* cursor = nullptr;  // next frame nobody has taken yet (shared by the workers of one call)
    uint8_t* out = nullptr;
    size_t out_stride = 0;
    uint64_t* bits = nullptr;
    int* frame_status = nullptr;
    int rc = MI355_OK;
};

struct PoolWorker {
    mi355_jpeg_pool* pool = nullptr;
    mi355_jpeg_ctx* c = nullptr;
    int numa_node = -1;
    // persistent device side
    uint8_t *d_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    uint64_t *d_bits[2] = {nullptr, nullptr}, *h_bits = nullptr;
    size_t in_cap[2] = {0, 0}, out_cap[2] = {0, 0}, bits_cap[2] = {0, 0}, hbits_cap = 0;
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // the worker thread and its mailbox
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    PoolJob* job = nullptr;  // set by pool_encode, cleared by the worker when done
    bool quit = false;
};

}  // namespace

struct mi355_jpeg_pool {
    std::vector<PoolWorker*> w;
    std::mutex reg_mu;
    std::vector<std::pair<uintptr_t, size_t>> registered;  // host ranges registered through the pool
    // what the pool has created since it exists (mi355_jpeg_pool_debug_counts): device allocations, host registrations,
    // streams + events, encode calls
    std::atomic<unsigned long long> n_alloc{0}, n_register{0}, n_objects{0}, n_calls{0};
};

namespace {

// CPUs of the NUMA node a GPU hangs off, from sysfs (best effort: -1 / empty when the platform does not say)
int gpu_numa_node(int device) {
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return -1;
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}
bool pin_to_numa_node(int node) {
    if (node < 0) return false;
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int any = 0;
    for (char* p = buf; *p && *p != '\n';) {  // "0-15,32-47"
        char* end = nullptr;
        long a = strtol(p, &end, 10), b2 = a;
        if (end == p) break;
        p = end;
        if (*p == '-') {
            b2 = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long k = a; k <= b2 && k < CPU_SETSIZE; ++k) CPU_SET((int)k, &set), ++any;
        if (*p == ',') ++p;
    }
    return any && sched_setaffinity(0, sizeof set, &set) == 0;
}

template <typename T>
bool pool_ensure(mi355_jpeg_pool* p, T*& ptr, size_t& cap, size_t need) {
    if (ptr && need <= cap) return true;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    if (hipMalloc((void**)&ptr, need * sizeof(T)) != hipSuccess) return false;
    p->n_alloc++;
    cap = need;
    return true;
}

void pool_run(PoolWorker* w, PoolJob* j) {
    j->rc = MI355_OK;
    mi355_jpeg_ctx* c = w->c;
    mi355_jpeg_pool* pool = w->pool;
    if (hipSetDevice(c->device) != hipSuccess) {
        j->rc = MI355_E_NO_DEVICE;
        return;
    }
    const size_t fbytes = (size_t)j->W * j->H * 3;
    size_t dstride = (j->out_stride + 3) & ~(size_t)3;
    if (dstride < 8) dstride = 8;
    // Work is handed out in chunks from a cursor all workers of the call share (about 256 MB of input each, less for
    // small batches so that every worker gets some): a GPU that is slower today simply takes fewer chunks.
    const uint32_t chunk = j->chunk;
    int rc = MI355_OK;
    auto H = [&](hipError_t e) {
        if (e != hipSuccess && rc == MI355_OK) rc = MI355_E_HIP - (int)e;
        return e == hipSuccess;
    };
    if (!w->s_in) {  // first job of this worker: streams and events, once
        H(hipStreamCreateWithFlags(&w->s_in, hipStreamNonBlocking));
        H(hipStreamCreateWithFlags(&w->s_cmp, hipStreamNonBlocking));
        H(hipStreamCreateWithFlags(&w->s_out, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            H(hipEventCreateWithFlags(&w->ev_in[b], hipEventDisableTiming));
            H(hipEventCreateWithFlags(&w->ev_cmp[b], hipEventDisableTiming));
            H(hipEventCreateWithFlags(&w->ev_out[b], hipEventDisableTiming));
        }
        pool->n_objects += 9;
    }
    for (int b = 0; b < 2 && rc == MI355_OK; ++b)
        if (!pool_ensure(pool, w->d_in[b], w->in_cap[b], fbytes * chunk) || !pool_ensure(pool, w->d_out[b], w->out_cap[b], dstride * chunk) ||
            !pool_ensure(pool, w->d_bits[b], w->bits_cap[b], (size_t)chunk))
            rc = MI355_E_ALLOC;
    if (rc == MI355_OK && (!w->h_bits || w->hbits_cap < (size_t)chunk * 2)) {
        if (w->h_bits) (void)hipHostFree(w->h_bits);
        w->h_bits = nullptr;
        if (H(hipHostMalloc((void**)&w->h_bits, sizeof(uint64_t) * chunk * 2, hipHostMallocDefault))) {
            w->hbits_cap = (size_t)chunk * 2;
            pool->n_alloc++;
        }
    }
    uint8_t *const *d_in = w->d_in, *const *d_out = w->d_out;
    uint64_t *const *d_bits = w->d_bits, *h_bits = w->h_bits;
    hipStream_t s_in = w->s_in, s_cmp = w->s_cmp, s_out = w->s_out;
    hipEvent_t *ev_in = w->ev_in, *ev_cmp = w->ev_cmp, *ev_out = w->ev_out;
    // (caller memory that is not registered through the pool was registered for this call by pool_encode, as ONE range
    // per buffer: slabs registered worker by worker share pages at their ends, and a page unpinned by the worker that
    // finishes first is a page the other worker's DMA still targets)
    uint32_t first_of[2] = {0, 0}, frames_of[2] = {0, 0};  // the chunk in flight in each of the two buffer sets
    int frame_rc = MI355_OK;  // a frame that does not fit / has no code: the others still come back
    auto drain = [&](int b) {  // the chunk in buffer set b is encoded: fetch its bit counts, then its bytes
        const uint32_t a = first_of[b], nf = frames_of[b];
        H(hipStreamWaitEvent(s_out, ev_cmp[b], 0));
        H(hipMemcpyAsync(h_bits + (size_t)b * chunk, d_bits[b], sizeof(uint64_t) * nf, hipMemcpyDeviceToHost, s_out));
        H(hipStreamSynchronize(s_out));
        for (uint32_t f = 0; f < nf && rc == MI355_OK; ++f) {
            const uint64_t nb = h_bits[(size_t)b * chunk + f];
            int st = MI355_OK;
            if (nb >= kBitsFlagged) {  // the device's verdict for this frame, with its cause
                st = nb == kBitsCategory ? MI355_E_CATEGORY : MI355_E_CAPACITY;
            } else if ((size_t)((nb + 7) / 8) > j->out_stride) {  // fits the 4-byte-rounded device slot, not the caller's
                st = MI355_E_CAPACITY;
            }
            j->bits[a + f] = st == MI355_OK ? nb : (st == MI355_E_CATEGORY ? kBitsCategory : kBitsCapacity);
            if (j->frame_status) j->frame_status[a + f] = st;
            if (st != MI355_OK) {
                if (frame_rc == MI355_OK || st == MI355_E_CATEGORY) frame_rc = st;  // (category first, like mi355_jpeg_sync)
                continue;
            }
            H(hipMemcpyAsync(j->out + (size_t)(a + f) * j->out_stride, d_out[b] + (size_t)f * dstride, (size_t)((nb + 7) / 8),
                             hipMemcpyDeviceToHost, s_out));
        }
        H(hipEventRecord(ev_out[b], s_out));
    };
    uint32_t k = 0;
    for (; rc == MI355_OK; ++k) {
        const uint32_t a = j->cursor->fetch_add(chunk);
        if (a >= j->n) break;
        const int b = (int)(k & 1);
        const uint32_t nf = a + chunk <= j->n ? chunk : j->n - a;
        if (k >= 2) {  // buffers of chunk k-2 must be fully drained before reuse
            H(hipStreamWaitEvent(s_in, ev_out[b], 0));
            H(hipStreamWaitEvent(s_cmp, ev_out[b], 0));
        }
        first_of[b] = a, frames_of[b] = nf;
        H(hipMemcpyAsync(d_in[b], j->rgb + (size_t)a * fbytes, fbytes * nf, hipMemcpyHostToDevice, s_in));
        H(hipEventRecord(ev_in[b], s_in));
        H(hipStreamWaitEvent(s_cmp, ev_in[b], 0));
        if (rc == MI355_OK) {
            int e = mi355_jpeg_encode_scan_device(c, d_in[b], j->W, j->H, nf, j->flags, d_out[b], dstride, d_bits[b], s_cmp);
            if (e) rc = e;
        }
        H(hipEventRecord(ev_cmp[b], s_cmp));
        if (k >= 1 && rc == MI355_OK) drain((int)((k - 1) & 1));  // overlaps the encode of chunk k
    }
    if (k >= 1 && rc == MI355_OK) drain((int)((k - 1) & 1));
    if (s_out) (void)hipStreamSynchronize(s_out);
    if (s_in) (void)hipStreamSynchronize(s_in);
    if (s_cmp) {
        // also clears the device-side status for the next call.  Capacity / category are reported per frame above; what
        // is left for the call's return value is anything else (a HIP error, MI355_E_INTERNAL).
        const int e = mi355_jpeg_sync(c, s_cmp);
        if (rc == MI355_OK && e && e != MI355_E_CATEGORY && e != MI355_E_CAPACITY) rc = e;
    }
    if (rc == MI355_OK) rc = frame_rc;
    j->rc = rc;
}

void pool_thread(PoolWorker* w) {
    (void)hipSetDevice(w->c->device);
    (void)pin_to_numa_node(w->numa_node);  // best effort: the copies' staging and the launches stay next to the GPU
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->job != nullptr || w->quit; });
        if (w->quit) break;
        PoolJob* j = w->job;
        lk.unlock();
        pool_run(w, j);
        lk.lock();
        w->job = nullptr;
        w->cv.notify_all();
    }
}

bool pool_covers(mi355_jpeg_pool* p, const void* ptr, size_t bytes) {
    std::lock_guard<std::mutex> g(p->reg_mu);
    const uintptr_t a = (uintptr_t)ptr;
    for (auto& r : p->registered)
        if (a >= r.first && a + bytes <= r.first + r.second) return true;
    return false;
}

}  // namespace

extern "C" {

static int pool_create_body(const int* device_ids, int n_workers, mi355_jpeg_pool** out){
    if (!out) return MI355_E_ARG;
    *out = nullptr;
    if (device_ids && n_workers <= 0) return MI355_E_ARG;
    std::vector<int> ids;
    if (device_ids) ids.assign(device_ids, device_ids + n_workers);  // (may throw std::bad_alloc: stopped at the boundary)
    int ndev = mi355_jpeg_device_count();
    if (ndev <= 0) return MI355_E_NO_DEVICE;
    if (!device_ids)
        for (int d = 0; d < ndev; ++d) ids.push_back(d);
    mi355_jpeg_pool* p = new (std::nothrow) mi355_jpeg_pool();
    if (!p) return MI355_E_ALLOC;
    for (int id : ids) {
        mi355_jpeg_ctx* c = nullptr;
        int e = mi355_jpeg_create(id, &c);
        if (e) {
            mi355_jpeg_pool_destroy(p);
            return e;
        }
        PoolWorker* w = new (std::nothrow) PoolWorker();
        if (!w) {
            mi355_jpeg_destroy(c);
            mi355_jpeg_pool_destroy(p);
            return MI355_E_ALLOC;
        }
        w->pool = p;
        w->c = c;
        w->numa_node = gpu_numa_node(id);
        // Whatever throws from here on (the worker list growing, a thread that cannot start: std::system_error with
        // EAGAIN under a process limit), the pool made so far is torn down -- threads joined, contexts destroyed -- and
        // the caller gets MI355_E_ALLOC.
        try {
            p->w.push_back(w);
        } catch (...) {
            mi355_jpeg_destroy(c);
            delete w;
            mi355_jpeg_pool_destroy(p);
            return MI355_E_ALLOC;
        }
        try {
            w->th = std::thread(pool_thread, w);
        } catch (...) {
            mi355_jpeg_pool_destroy(p);
            return MI355_E_ALLOC;
        }
    }
    *out = p;
    return MI355_OK;
}
int mi355_jpeg_pool_create(const int* device_ids, int n_workers, mi355_jpeg_pool** out) {
    return guarded([&]() -> int { return pool_create_body(device_ids, n_workers, out); });
}

void mi355_jpeg_pool_destroy(mi355_jpeg_pool* p) {
    if (!p) return;
    for (PoolWorker* w : p->w) {
        {
            std::lock_guard<std::mutex> g(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        (void)hipSetDevice(w->c->device);
        for (int b = 0; b < 2; ++b) {
            if (w->d_in[b]) (void)hipFree(w->d_in[b]);
            if (w->d_out[b]) (void)hipFree(w->d_out[b]);
            if (w->d_bits[b]) (void)hipFree(w->d_bits[b]);
            if (w->ev_in[b]) (void)hipEventDestroy(w->ev_in[b]);
            if (w->ev_cmp[b]) (void)hipEventDestroy(w->ev_cmp[b]);
            if (w->ev_out[b]) (void)hipEventDestroy(w->ev_out[b]);
        }
        if (w->h_bits) (void)hipHostFree(w->h_bits);
        if (w->s_in) (void)hipStreamDestroy(w->s_in);
        if (w->s_cmp) (void)hipStreamDestroy(w->s_cmp);
        if (w->s_out) (void)hipStreamDestroy(w->s_out);
        mi355_jpeg_destroy(w->c);
        delete w;
    }
    for (auto& r : p->registered) (void)hipHostUnregister((void*)r.first);
    delete p;
}

int mi355_jpeg_pool_workers(mi355_jpeg_pool* p) { return p ? (int)p->w.size() : 0; }

static int pool_set_quant_body(mi355_jpeg_pool* p, const uint32_t qlum[64], const uint32_t qchrom[64]){
    if (!p) return MI355_E_ARG;
    for (auto* w : p->w) {
        int e = mi355_jpeg_set_quant(w->c, qlum, qchrom);
        if (e) return e;
    }
    return MI355_OK;
}
int mi355_jpeg_pool_set_quant(mi355_jpeg_pool* p, const uint32_t qlum[64], const uint32_t qchrom[64]) {
    return guarded([&]() -> int { return pool_set_quant_body(p, qlum, qchrom); });
}

static int pool_set_quality_body(mi355_jpeg_pool* p, int quality){
    if (!p) return MI355_E_ARG;
    for (auto* w : p->w) {
        int e = mi355_jpeg_set_quality(w->c, quality);
        if (e) return e;
    }
    return MI355_OK;
}
int mi355_jpeg_pool_set_quality(mi355_jpeg_pool* p, int quality) {
    return guarded([&]() -> int { return pool_set_quality_body(p, quality); });
}

static int pool_set_huffman_body(mi355_jpeg_pool* p, int table, const mi355_huff_table* t) {
    if (!p) return MI355_E_ARG;
    for (auto* w : p->w) {
        int e = mi355_jpeg_set_huffman(w->c, table, t);
        if (e) return e;
    }
    return MI355_OK;
}
int mi355_jpeg_pool_set_huffman(mi355_jpeg_pool* p, int table, const mi355_huff_table* t) {
    return guarded([&]() -> int { return pool_set_huffman_body(p, table, t); });
}

static int pool_register_body(mi355_jpeg_pool* p, void* ptr, size_t bytes){
    if (!p || !ptr || !bytes) return MI355_E_ARG;
    if (pool_covers(p, ptr, bytes)) return MI355_OK;
    if (!p->w.empty()) HIP_TRY(hipSetDevice(p->w[0]->c->device));
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));  // pinned for every device of the pool
    p->n_register++;
    std::lock_guard<std::mutex> g(p->reg_mu);
    p->registered.push_back({(uintptr_t)ptr, bytes});
    return MI355_OK;
}
int mi355_jpeg_pool_register(mi355_jpeg_pool* p, void* ptr, size_t bytes) {
    return guarded([&]() -> int { return pool_register_body(p, ptr, bytes); });
}

static int pool_unregister_body(mi355_jpeg_pool* p, void* ptr){
    if (!p || !ptr) return MI355_E_ARG;
    std::lock_guard<std::mutex> g(p->reg_mu);
    for (size_t i = 0; i < p->registered.size(); ++i)
        if (p->registered[i].first == (uintptr_t)ptr) {
            p->registered.erase(p->registered.begin() + (long)i);
            HIP_TRY(hipHostUnregister(ptr));
            return MI355_OK;
        }
    return MI355_E_ARG;
}
int mi355_jpeg_pool_unregister(mi355_jpeg_pool* p, void* ptr) {
    return guarded([&]() -> int { return pool_unregister_body(p, ptr); });
}

static int pool_debug_counts_body(mi355_jpeg_pool* p, uint64_t counts[4]){
    if (!p || !counts) return MI355_E_ARG;
    counts[0] = p->n_alloc, counts[1] = p->n_register, counts[2] = p->n_objects, counts[3] = p->n_calls;
    return MI355_OK;
}
int mi355_jpeg_pool_debug_counts(mi355_jpeg_pool* p, uint64_t counts[4]) {
    return guarded([&]() -> int { return pool_debug_counts_body(p, counts); });
}

static int pool_encode_ex_body(mi355_jpeg_pool* p, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t n_frames,
                              uint32_t flags, uint8_t* out, size_t out_stride, uint64_t* bits, int* frame_status,
                              double* seconds){
    if (!p || p->w.empty() || !rgb || !out || !bits || n_frames == 0) return MI355_E_ARG;
    Geom g;
    int e = make_geom(W, H, flags, nullptr, &g);
    if (e) return e;
    const auto t0 = std::chrono::steady_clock::now();
    p->n_calls++;
    const uint32_t nw = (uint32_t)p->w.size();
    // Host memory the caller has not registered through the pool is registered here for the duration of the call --
    // each buffer as ONE range, before any worker starts, released after the last one has finished (best effort:
    // memory that is pinned already, e.g. by the caller's allocator, refuses and is fine as it is).
    bool in_reg = pool_covers(p, rgb, (size_t)g.frame_stride * n_frames);
    bool out_reg = pool_covers(p, out, out_stride * n_frames);
    bool tmp_in = false, tmp_out = false;
    (void)hipSetDevice(p->w[0]->c->device);
    if (!in_reg) {
        tmp_in = hipHostRegister((void*)rgb, (size_t)g.frame_stride * n_frames, hipHostRegisterPortable) == hipSuccess;
        if (tmp_in) p->n_register++;
    }
    if (!out_reg) {
        tmp_out = hipHostRegister((void*)out, out_stride * n_frames, hipHostRegisterPortable) == hipSuccess;
        if (tmp_out) p->n_register++;
    }
    (void)hipGetLastError();
    // Every frame starts as "not delivered": whatever ends a worker early (a HIP error, an allocation that fails), the
    // caller finds a defined verdict in bits[] / frame_status[] for the frames that worker never reached.
    for (uint32_t f = 0; f < n_frames; ++f) {
        bits[f] = kBitsCapacity;
        if (frame_status) frame_status[f] = MI355_E_NOT_ENCODED;
    }
    std::atomic<uint32_t> cursor{0};
    uint32_t chunk = (uint32_t)((256u << 20) / ((size_t)g.frame_stride ? (size_t)g.frame_stride : 1));
    const uint32_t share = (n_frames + nw * 4 - 1) / (nw * 4);  // small batches: at least four chunks per worker
    if (chunk > share) chunk = share;
    if (chunk < 1) chunk = 1;
    std::vector<PoolJob> jobs(nw);
    for (uint32_t w = 0; w < nw; ++w) {
        PoolJob& jb = jobs[w];
        jb.rgb = rgb, jb.W = W, jb.H = H, jb.flags = flags, jb.out = out, jb.out_stride = out_stride, jb.bits = bits;
        jb.frame_status = frame_status;
        jb.n = n_frames, jb.chunk = chunk, jb.cursor = &cursor;
        PoolWorker* pw = p->w[w];
        {
            std::lock_guard<std::mutex> gk(pw->mu);
            pw->job = &jb;
        }
        pw->cv.notify_all();
    }
    for (uint32_t w = 0; w < nw; ++w) {
        PoolWorker* pw = p->w[w];
        std::unique_lock<std::mutex> lk(pw->mu);
        pw->cv.wait(lk, [&] { return pw->job == nullptr; });
    }
    if (tmp_in) (void)hipHostUnregister((void*)rgb);
    if (tmp_out) (void)hipHostUnregister((void*)out);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int rc = MI355_OK;  // hard errors first, then category, then capacity
    for (auto& jb : jobs) {
        const int e = jb.rc;
        if (!e) continue;
        const bool soft_e = e == MI355_E_CATEGORY || e == MI355_E_CAPACITY, soft_rc = rc == MI355_E_CATEGORY || rc == MI355_E_CAPACITY;
        if (rc == MI355_OK || (soft_rc && !soft_e) || (rc == MI355_E_CAPACITY && e == MI355_E_CATEGORY)) rc = e;
    }
    return rc;
}
int mi355_jpeg_pool_encode_ex(mi355_jpeg_pool* p, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t n_frames,
                              uint32_t flags, uint8_t* out, size_t out_stride, uint64_t* bits, int* frame_status,
                              double* seconds) {
    return guarded([&]() -> int { return pool_encode_ex_body(p, rgb, W, H, n_frames, flags, out, out_stride, bits, frame_status, seconds); });
}

static int pool_encode_body(mi355_jpeg_pool* p, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t n_frames,
                           uint32_t flags, uint8_t* out, size_t out_stride, uint64_t* bits, double* seconds){
    return mi355_jpeg_pool_encode_ex(p, rgb, W, H, n_frames, flags, out, out_stride, bits, nullptr, seconds);
}
int mi355_jpeg_pool_encode(mi355_jpeg_pool* p, const uint8_t* rgb, uint32_t W, uint32_t H, uint32_t n_frames,
                           uint32_t flags, uint8_t* out, size_t out_stride, uint64_t* bits, double* seconds) {
    return guarded([&]() -> int { return pool_encode_body(p, rgb, W, H, n_frames, flags, out, out_stride, bits, seconds); });
}

// ---- the reference's stage functions, one by one ---------------------------------------
// Host image in, one stage kernel, host image out.  Scratch: two device buffers grown on demand.
}  // extern "C"
namespace {
struct Stage {
    mi355_jpeg_ctx* c;
    int e = MI355_OK;
    explicit Stage(mi355_jpeg_ctx* ctx) : c(ctx) {
        if (!c) e = MI355_E_ARG;
        else if (hipSetDevice(c->device) != hipSuccess) e = MI355_E_NO_DEVICE;
    }
    void* buf(int which, size_t bytes) {  // device scratch `which` (0..3) of at least `bytes`
        if (e) return nullptr;
        if (ensure(c->d_stage[which], c->stage_cap[which], bytes ? bytes : 1)) e = MI355_E_ALLOC;
        return e ? nullptr : c->d_stage[which];
    }
    void up(void* d, const void* h, size_t n) {
        if (!e) e = hip_err(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    }
    void down(void* h, const void* d, size_t n) {
        if (!e) e = hip_err(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    }
    void zero(void* d, size_t n) {
        if (!e) e = hip_err(hipMemset(d, 0, n));
    }
    void run(hipError_t le) {
        if (!e && le != hipSuccess) e = MI355_E_HIP - (int)le;
    }
};
}  // namespace
extern "C" {

static int stage_csc_body(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W, uint32_t H){
    if (!img || !W || !H) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W * H * 3;
    uint8_t* d = (uint8_t*)st.buf(0, n);
    st.up(d, img, n);
    if (!st.e) st.run(launch_stage_csc(d, (uint64_t)W * H, nullptr));
    st.down(img, d, n);
    return st.e;
}
int mi355_jpeg_stage_csc(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W, uint32_t H) {
    return guarded([&]() -> int { return stage_csc_body(c, img, W, H); });
}

static int stage_cds_body(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W, uint32_t H){
    if (!img || !W || !H) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W * H * 3;
    uint8_t* d = (uint8_t*)st.buf(0, n);
    st.up(d, img, n);
    if (!st.e && W >= 2 && H >= 2) st.run(launch_stage_cds(d, W, H, nullptr));
    st.down(img, d, n);
    return st.e;
}
int mi355_jpeg_stage_cds(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W, uint32_t H) {
    return guarded([&]() -> int { return stage_cds_body(c, img, W, H); });
}

static int stage_copy_larger_body(mi355_jpeg_ctx* c, const uint8_t* src, uint32_t W, uint32_t H, uint8_t* dst, uint32_t W8,
                                 uint32_t H8){
    if (!src || !dst || !W || !H || W8 < W || H8 < H) return MI355_E_ARG;
    Stage st(c);
    const size_t ns = (size_t)W * H * 3, nd = (size_t)W8 * H8 * 3;
    uint8_t* ds = (uint8_t*)st.buf(0, ns);
    uint8_t* dd = (uint8_t*)st.buf(1, nd);
    st.up(ds, src, ns);
    st.up(dd, dst, nd);  // what lies outside the copied corner stays as the caller had it
    if (!st.e) st.run(launch_stage_copy_larger(ds, W, H, dd, W8, nullptr));
    st.down(dst, dd, nd);
    return st.e;
}
int mi355_jpeg_stage_copy_larger(mi355_jpeg_ctx* c, const uint8_t* src, uint32_t W, uint32_t H, uint8_t* dst, uint32_t W8,
                                 uint32_t H8) {
    return guarded([&]() -> int { return stage_copy_larger_body(c, src, W, H, dst, W8, H8); });
}

static int stage_mirror_pad_body(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH){
    if (!img || !oldW || !oldH || W8 < oldW || H8 < oldH) return MI355_E_ARG;
    if (W8 - oldW > oldW || H8 - oldH > oldH) return MI355_E_ARG;  // the reference underflows `oldWidth - diff` there
    Stage st(c);
    const size_t n = (size_t)W8 * H8 * 3;
    uint8_t* d = (uint8_t*)st.buf(0, n);
    st.up(d, img, n);
    if (!st.e) st.run(launch_stage_mirror(d, W8, H8, oldW, oldH, nullptr));
    st.down(img, d, n);
    return st.e;
}
int mi355_jpeg_stage_mirror_pad(mi355_jpeg_ctx* c, uint8_t* img, uint32_t W8, uint32_t H8, uint32_t oldW, uint32_t oldH) {
    return guarded([&]() -> int { return stage_mirror_pad_body(c, img, W8, H8, oldW, oldH); });
}

static int stage_to_double_body(mi355_jpeg_ctx* c, const uint8_t* src, double* dst, uint32_t W, uint32_t H){
    if (!src || !dst || !W || !H) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W * H * 3;
    uint8_t* ds = (uint8_t*)st.buf(0, n);
    double* dd = (double*)st.buf(1, n * sizeof(double));
    st.up(ds, src, n);
    if (!st.e) st.run(launch_stage_u8_to_f64(ds, dd, n, nullptr));
    st.down(dst, dd, n * sizeof(double));
    return st.e;
}
int mi355_jpeg_stage_to_double(mi355_jpeg_ctx* c, const uint8_t* src, double* dst, uint32_t W, uint32_t H) {
    return guarded([&]() -> int { return stage_to_double_body(c, src, dst, W, H); });
}

static int stage_subtract_body(mi355_jpeg_ctx* c, double* img, uint32_t W, uint32_t H, double value){
    if (!img || !W || !H) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W * H * 3;
    double* d = (double*)st.buf(1, n * sizeof(double));
    st.up(d, img, n * sizeof(double));
    if (!st.e) st.run(launch_stage_sub(d, n, value, nullptr));
    st.down(img, d, n * sizeof(double));
    return st.e;
}
int mi355_jpeg_stage_subtract(mi355_jpeg_ctx* c, double* img, uint32_t W, uint32_t H, double value) {
    return guarded([&]() -> int { return stage_subtract_body(c, img, W, H, value); });
}

static int stage_dct_body(mi355_jpeg_ctx* c, double* img, uint32_t W8, uint32_t H8){
    if (!img || !W8 || !H8 || (W8 & 7) || (H8 & 7)) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W8 * H8 * 3;
    double* d = (double*)st.buf(1, n * sizeof(double));
    st.up(d, img, n * sizeof(double));
    if (!st.e) st.run(launch_stage_dct(d, W8, H8, nullptr));
    st.down(img, d, n * sizeof(double));
    return st.e;
}
int mi355_jpeg_stage_dct(mi355_jpeg_ctx* c, double* img, uint32_t W8, uint32_t H8) {
    return guarded([&]() -> int { return stage_dct_body(c, img, W8, H8); });
}

static int stage_quantize_body(mi355_jpeg_ctx* c, double* img, uint32_t W8, uint32_t H8, const uint32_t qlum[64],
                              const uint32_t qchrom[64]){
    if (!img || !qlum || !qchrom || !W8 || !H8 || (W8 & 7) || (H8 & 7)) return MI355_E_ARG;
    double q[128];
    for (int i = 0; i < 64; ++i) {
        if (!qlum[i] || !qchrom[i]) return MI355_E_TABLE;
        q[i] = (double)qlum[i], q[64 + i] = (double)qchrom[i];
    }
    Stage st(c);
    const size_t n = (size_t)W8 * H8 * 3;
    double* d = (double*)st.buf(1, n * sizeof(double));
    double* dq = (double*)st.buf(2, sizeof q);
    st.up(d, img, n * sizeof(double));
    st.up(dq, q, sizeof q);
    if (!st.e) st.run(launch_stage_quant(d, W8, H8, dq, nullptr));
    st.down(img, d, n * sizeof(double));
    return st.e;
}
int mi355_jpeg_stage_quantize(mi355_jpeg_ctx* c, double* img, uint32_t W8, uint32_t H8, const uint32_t qlum[64],
                              const uint32_t qchrom[64]) {
    return guarded([&]() -> int { return stage_quantize_body(c, img, W8, H8, qlum, qchrom); });
}

static int stage_blocks_body(mi355_jpeg_ctx* c, const double* img, uint32_t W8, uint32_t H8, int32_t* linear){
    if (!img || !linear || !W8 || !H8 || (W8 & 7) || (H8 & 7)) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)W8 * H8 * 3;
    double* d = (double*)st.buf(1, n * sizeof(double));
    int* dl = (int*)st.buf(0, n * sizeof(int));
    st.up(d, img, n * sizeof(double));
    if (!st.e) st.run(launch_stage_blocks(d, W8, H8, dl, nullptr));
    st.down(linear, dl, n * sizeof(int));
    return st.e;
}
int mi355_jpeg_stage_blocks(mi355_jpeg_ctx* c, const double* img, uint32_t W8, uint32_t H8, int32_t* linear) {
    return guarded([&]() -> int { return stage_blocks_body(c, img, W8, H8, linear); });
}

static int stage_zigzag_body(mi355_jpeg_ctx* c, const int32_t* linear, int32_t* zigzag, uint32_t rows){
    if (!linear || !zigzag || !rows) return MI355_E_ARG;
    Stage st(c);
    const size_t n = (size_t)rows * 64 * sizeof(int);
    int* dl = (int*)st.buf(0, n);
    int* dz = (int*)st.buf(1, n);
    st.up(dl, linear, n);
    if (!st.e) st.run(launch_stage_zigzag(dl, dz, rows, nullptr));
    st.down(zigzag, dz, n);
    return st.e;
}
int mi355_jpeg_stage_zigzag(mi355_jpeg_ctx* c, const int32_t* linear, int32_t* zigzag, uint32_t rows) {
    return guarded([&]() -> int { return stage_zigzag_body(c, linear, zigzag, rows); });
}

static int stage_rle_body(mi355_jpeg_ctx* c, const int32_t* zigzag, uint32_t rows, int32_t* pairs, uint32_t* counts){
    if (!zigzag || !pairs || !counts || !rows) return MI355_E_ARG;
    Stage st(c);
    int* dz = (int*)st.buf(0, (size_t)rows * 64 * sizeof(int));
    int* dp = (int*)st.buf(1, (size_t)rows * 128 * sizeof(int));
    uint32_t* dc = (uint32_t*)st.buf(2, (size_t)rows * sizeof(uint32_t));
    st.up(dz, zigzag, (size_t)rows * 64 * sizeof(int));
    if (!st.e) st.run(launch_stage_rle(dz, rows, dp, dc, nullptr));
    st.down(counts, dc, (size_t)rows * sizeof(uint32_t));
    st.down(pairs, dp, (size_t)rows * 128 * sizeof(int));
    return st.e;
}
int mi355_jpeg_stage_rle(mi355_jpeg_ctx* c, const int32_t* zigzag, uint32_t rows, int32_t* pairs, uint32_t* counts) {
    return guarded([&]() -> int { return stage_rle_body(c, zigzag, rows, pairs, counts); });
}

static int stage_huffman_body(mi355_jpeg_ctx* c, const int32_t* zigzag, const int32_t* pairs, const uint32_t* counts,
                             uint32_t N, uint8_t* out, size_t cap, uint64_t* bits){
    if (!zigzag || !pairs || !counts || !out || !bits || !N) return MI355_E_ARG;
    const size_t rows = (size_t)N * 3;
    for (size_t r = 0; r < rows; ++r)
        if (counts[r] > 128 || (counts[r] & 1)) return MI355_E_ARG;
    Stage st(c);
    const size_t chunks = (rows + 1023) / 1024;
    const size_t out_words = (cap + 3) / 4;
    int* dz = (int*)st.buf(0, rows * 64 * sizeof(int));
    int* dp = (int*)st.buf(1, rows * 128 * sizeof(int));
    // counts | unit bits | in-chunk offsets | chunk sums (u64) | total (u64) | status word of THIS call, then the output
    // words.  The call has a status word of its own: the context-wide one belongs to the asynchronous encode calls (a
    // pending error of one of them must stay there for the caller's next mi355_jpeg_sync, not surface here).
    const size_t off_ub = rows * 4, off_ic = off_ub + rows * 4, off_cs = (off_ic + rows * 4 + 7) & ~(size_t)7;
    const size_t off_tot = off_cs + chunks * 8, off_st = off_tot + 8, off_out = off_st + 8;
    uint8_t* dm = (uint8_t*)st.buf(2, off_out + out_words * 4);
    st.up(dz, zigzag, rows * 64 * sizeof(int));
    st.up(dp, pairs, rows * 128 * sizeof(int));
    st.up(dm, counts, rows * 4);
    st.zero(dm + off_st, 8 + out_words * 4);
    if (!st.e)
        st.run(launch_stage_huffman(dz, dp, (const uint32_t*)dm, N, c->d_lut, (uint32_t*)(dm + off_ub), (uint32_t*)(dm + off_ic),
                                    (uint64_t*)(dm + off_cs), (uint64_t*)(dm + off_tot), (uint32_t*)(dm + off_out),
                                    (uint64_t)out_words * 32, (uint32_t*)(dm + off_st), nullptr));
    uint32_t stw = 0;
    st.down(&stw, dm + off_st, sizeof stw);  // (the copy waits for the null stream's kernels)
    if (!st.e) st.e = status_to_error(stw);
    st.down(bits, dm + off_tot, sizeof(uint64_t));
    if (st.e) return st.e;
    const size_t nb = (size_t)((*bits + 7) / 8);
    if (nb > cap) return MI355_E_CAPACITY;
    st.down(out, dm + off_out, nb);
    return st.e;
}
int mi355_jpeg_stage_huffman(mi355_jpeg_ctx* c, const int32_t* zigzag, const int32_t* pairs, const uint32_t* counts,
                             uint32_t N, uint8_t* out, size_t cap, uint64_t* bits) {
    return guarded([&]() -> int { return stage_huffman_body(c, zigzag, pairs, counts, N, out, cap, bits); });
}

// ---- measurement ---------------------------------------------------------------

static int last_call_launches_body(mi355_jpeg_ctx* c, uint32_t* n){
    if (!c || !n) return MI355_E_ARG;
    *n = c->last_launches;
    return MI355_OK;
}
int mi355_jpeg_last_call_launches(mi355_jpeg_ctx* c, uint32_t* n) {
    return guarded([&]() -> int { return last_call_launches_body(c, n); });
}

static int screen_stats_body(mi355_jpeg_ctx* c, void* stream, mi355_jpeg_screen_counts* out, int reset){
    if (!c || !out) return MI355_E_ARG;
    HIP_TRY(hipSetDevice(c->device));
    (void)stream;
    // every stream of the device, the library's side stream included: kernels still running anywhere would keep adding
    // to the counters while they are read or reset
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(h, c->d_stats, sizeof h, hipMemcpyDeviceToHost));
    out->second_looks = h[0];
    out->exact_units = h[1];
    out->rewalked_units = h[2];
    out->general_passes = h[3];
    if (reset) {
        HIP_TRY(hipMemset(c->d_stats, 0, sizeof h));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return MI355_OK;
}
int mi355_jpeg_screen_stats(mi355_jpeg_ctx* c, void* stream, mi355_jpeg_screen_counts* out, int reset) {
    return guarded([&]() -> int { return screen_stats_body(c, stream, out, reset); });
}

static int set_profiling_body(mi355_jpeg_ctx* c, int mode){
    if (!c || mode < 0 || mode > 2) return MI355_E_ARG;
    // stage probes and the entropy-only entry point do not open event sets: drop
    // half-recorded state by starting over
    c->profiling = mode;
    c->ev_used = 0;
    c->ev_open = false;
    return MI355_OK;
}
int mi355_jpeg_set_profiling(mi355_jpeg_ctx* c, int mode) {
    return guarded([&]() -> int { return set_profiling_body(c, mode); });
}

static int last_timings_body(mi355_jpeg_ctx* c, mi355_jpeg_timings* t){
    if (!c || !t) return MI355_E_ARG;
    memset(t, 0, sizeof *t);
    if (!c->profiling || c->ev_used == 0) return MI355_E_ARG;
    return timings_of(c, c->ev_pool[c->ev_used - 1], t);
}
int mi355_jpeg_last_timings(mi355_jpeg_ctx* c, mi355_jpeg_timings* t) {
    return guarded([&]() -> int { return last_timings_body(c, t); });
}

static int profile_summary_body(mi355_jpeg_ctx* c, mi355_jpeg_timings* sum, uint32_t* calls){
    if (!c || !sum || !calls) return MI355_E_ARG;
    memset(sum, 0, sizeof *sum);
    *calls = 0;
    if (!c->profiling) return MI355_E_ARG;
    double acc[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < c->ev_used; ++i) {
        mi355_jpeg_timings t;
        int e = timings_of(c, c->ev_pool[i], &t);
        if (e) return e;
        acc[0] += t.transform_ms, acc[1] += t.size_ms, acc[2] += t.scan_ms, acc[3] += t.emit_ms,
            acc[4] += t.total_ms;
    }
    sum->transform_ms = (float)acc[0], sum->size_ms = (float)acc[1], sum->scan_ms = (float)acc[2];
    sum->emit_ms = (float)acc[3], sum->total_ms = (float)acc[4];
    *calls = (uint32_t)c->ev_used;
    return MI355_OK;
}
int mi355_jpeg_profile_summary(mi355_jpeg_ctx* c, mi355_jpeg_timings* sum, uint32_t* calls) {
    return guarded([&]() -> int { return profile_summary_body(c, sum, calls); });
}

}  // extern "C"
