// A whole MBConv block of the 7 x 7 stage (blocks 13-16) as ONE kernel, f16 handles, round 6: expand 1x1 + BN + Swish ->
// depthwise k x k + BN + Swish -> squeeze-excite (mean, reduce, Swish, excite, sigmoid) -> gate -> project 1x1 + BN (+ skip).
// One workgroup (8 waves) per crop; the expanded tensor, the depthwise output, the channel means and the gate never leave the CU.
//
// Reference: efficientnet 0.0.4 MBConvBlock / SEBlock as instantiated by /root/reference/whenet.py:8, blocks
// 13..16: Conv2D(192 -> 1152, 1x1) -> BN -> Swish -> DepthwiseConv2D(k = 5 | 3, 'same') -> BN -> Swish -> SEBlock(48) ->
// Conv2D(1152 -> 192 | 320, 1x1) -> BN (-> + input when the shapes agree).
//
// Why (round-5 review, items 3-5).  As three launches per block (front7.hip, se.hip, pw.hip) the stage costs 37 us per block and
// 64-crop chain -- 4 x 3 launches each paying a dependent boundary, ~147-288 workgroups that each pull the layer's weights again
// (2.2-7.8 x the algorithmic traffic), and the depthwise output (7.2 MB per 64 crops) written and re-read.  One crop's depthwise output is
// 49 x 1152 x 2 B = 113 KB: it fits the CU's 160 KB of LDS, and with it the whole block does:
//   phase 1  (per wave, 32-channel tiles t = wave, wave + 8, ..)  expand on v_mfma_f32_32x32x16_f16 -- pixels as MFMA rows (the crop's
//            7 x 8 pixel slots = 2 strips, fragments in registers for the whole phase), the tile's weight fragments straight from
//            the packed image (every fragment has exactly one consumer wave: no LDS staging), the next tile's in flight;
//            BN + Swish -> the wave's PRIVATE channel-major tile E[32 ch][8 rows][8 slots] (row 7 and slot 7 are the 'SAME' zeros);
//   phase 2  (same wave, no workgroup barrier)  depthwise taps as per-channel Toeplitz products on v_mfma_f32_4x4x4_16B_f16 as
//            front2.hip / front7.hip, but with ONE crop the four columns of a block are four pairs of OUTPUT ROWS (column j: rows
//            2j, 2j + 1), each lane reading its own input rows; the Toeplitz operand is not fetched (480 B per channel: more than the
//            expand weights) but assembled from two 16-byte tap sequences per (channel, tap row) with four selects;
//            BN + Swish -> D[pixel][channel] (f16, the project's operand layout) + the channel sums;
//   phase 3  squeeze-excite in the workgroup: reduce conv (binary16 weights, f32 arithmetic, fixed order), Swish, excite, sigmoid;
//   phase 4  project, transposed as pw.hip (weights = MFMA rows): K split over 4 wave pairs, each wave 2 pixel strips x half the
//            output tiles; the gate multiplies the operand fragment as it leaves LDS (packed f16, as pw.hip);
//   phase 5  fixed-order combine (k0 + k1) + (k2 + k3) through LDS, + BN bias, + skip, 8-byte NHWC stores.
// Every order of summation depends on the layer only; a crop never meets another crop's data: results are bitwise independent of
// the batch, of the crop's position and of the launch.
//
// Bytes a workgroup pulls per block (L2 hits after the first workgroup of an XCD): expand 442 KB + taps 184 KB + squeeze-excite
// 221 KB + project 442 KB (block 16: 737 KB) = 1.3 MB; HBM traffic per crop: 49 x 192 x 2 read (twice with the skip) + 49 x Cout x 2
// written -- the algorithmic bytes of the block without any intermediate tensor.
//
// MEASURED (profiles/r06/mb7_probe_timeline.txt, pmc_f16_b64_c64_mb7_by_kernel.txt, ab_mb7*.txt; docs/experiments.md 12.3): 29.6 us per
// launch at one crop, 36.7 us at 256 -- against 3 x 8 us at one crop and 67 us at 256 for front7 + se + project.  Workgroup life 28.6 us =
// stage input 2.1 | expand + taps 13.1 (+ 2.4 waiting for the slowest wave) | squeeze-excite 3.6 | project 4.6 | combine + store 2.8: the
// tile phase is issue-bound at two waves per SIMD (423 VALU instructions per wave and tile beside 104 MFMAs; the 96 registers of the
// crop's expand fragments allow no third wave).  Batch 512 +3 %, 64 crops x 3 in flight +0.5-0.9 %, one forward at a time -4 %, batch 1
// +61 us: engine option "mb7", OFF by default (a schedule that pays only when every CU holds a crop cannot be chosen by batch size
// without giving up the bitwise batch invariance).
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#if defined(WHENET_STAMPS) && defined(WHENET_MB7_TILE_STAMPS)     // probe: the stamps time the phases of wave 0's second tile instead
#undef STAMP
#define STAMP(i)
#define TSTAMP(i)                                                                                          \
    do {                                                                                                   \
        if (threadIdx.x == 0 && t == 8 && ::whenet::whenet_stamps) ::whenet::whenet_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define TSTAMP(i)
#endif

namespace whenet {

namespace {

constexpr int CIN = 192, CEXP = 1152, RSE = 48, PX = 49;
constexpr int KSE = CIN / 16;              // k-steps of the expand contraction
constexpr int NTE = CEXP / 32;             // 32-channel tiles of the expanded tensor
constexpr int KSP = CEXP / 16;             // k-steps of the project contraction
constexpr int DP = CEXP * 2 + 16;          // pixel pitch of D in bytes (2320: consecutive pixels 4 banks apart)
constexpr int XP = CIN * 2 + 16;           // pixel pitch of the staged input (400)
constexpr int EP = 8 * 16 + 16;            // channel pitch of a wave's E tile: 8 rows x 8 slots f16 + 16
constexpr int EW = 32 * EP;                // a wave's E tile (4,608 B)
constexpr int NW = 8, NTHR = NW * 64;
constexpr int OFF_D = 0;                   // D [49][DP]; the staged input [49][XP] and the combine buffers alias it
constexpr int OFF_E = PX * DP;             // 113,680: E tiles of the 8 waves; the reduce partials alias them
constexpr int OFF_SUM = OFF_E + NW * EW;   // 150,544: channel sums f32 [1152]
constexpr int OFF_GATE = OFF_SUM + CEXP * 4;   // 155,152: gate f16 [1152]
constexpr int OFF_R = OFF_GATE + CEXP * 2;     // 157,456: r f32 [48]
constexpr int LDS_BYTES = OFF_R + RSE * 4;     // 157,648
constexpr int TPR = 3;                     // project tiles per wave and combine round (8 waves x 3 x 4 KB = 96 KB)
static_assert(NW * TPR * 4096 <= OFF_SUM, "combine buffer overlaps live data");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

struct Mb7Params {
    const half_t* x;        // [n][49][192]
    const half8* wep;       // expand image [12][36][64] half8 (snapshot.h fragment order)
    const float* be;        // [1152]
    const uint4* wds;       // tap sequences [72][5][16][2] x 16 B (pack_mb7_taps)
    const float* bd;        // [1152]
    const half8* w1p;       // reduce kernel [1152][6] half8 (pack_mb7_se)
    const float* b1;        // [48]
    const half8* w2p;       // excite kernel [6][1152] half8
    const float* b2;        // [1152]
    const half8* wpp;       // project image [72][NTP][64] half8
    const float* bp;        // [Cout]
    half_t* out;            // [n][49][Cout]
    half_t* dbg_dw;         // [n][49][1152] or nullptr (single-stage calls: tests)
    half_t* dbg_gate;       // [n][1152] or nullptr
    float inv_hw;
};

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half4 as_half4(unsigned lo, unsigned hi) { return __builtin_bit_cast(half4, uint2v{lo, hi}); }

// K: depthwise kernel size (5 | 3: embedded in a 5 x 5 kernel, only its non-zero tap rows are multiplied); NTP: 32-channel tiles of the
// project's output (6 | 10); RES: the block adds its input.
template <int K, int NTP, bool RES>
__global__ __launch_bounds__(NTHR) void whenet_mb7_kernel(const Mb7Params p) {
    constexpr int KY0 = (5 - K) / 2, KY1 = KY0 + K;          // tap rows of the embedded kernel that exist
    constexpr int NTW = NTP / 2;                             // project tiles per wave
    constexpr int COUT = NTP * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    const int crop = blockIdx.x;
    unsigned char* const Dl = smem + OFF_D;
    unsigned char* const Ew = smem + OFF_E + wave * EW;
    float* const s_sum = reinterpret_cast<float*>(smem + OFF_SUM);
    half_t* const s_gate = reinterpret_cast<half_t*>(smem + OFF_GATE);
    float* const s_r = reinterpret_cast<float*>(smem + OFF_R);
    float* const s_rp = reinterpret_cast<float*>(smem + OFF_E);          // [8 waves][48]

    STAMP(0);
    // ---- phase 0: the crop's 49 x 192 input -> LDS (coalesced), the first tile's weight fragments on their way ----------------
    half8 w[KSE];
    int t = wave;
#pragma unroll
    for (int ks = 0; ks < KSE; ++ks) w[ks] = p.wep[(ks * NTE + t) * 64 + lane];
    {
        const half8* src = reinterpret_cast<const half8*>(p.x + size_t(crop) * PX * CIN);
        constexpr int NV = PX * CIN / 8;                     // 1176 vectors of 16 B
        half8 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + i * NTHR;
            v[i] = src[idx < NV ? idx : NV - 1];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + i * NTHR;
            if (idx < NV) *reinterpret_cast<half8*>(Dl + (idx / 24) * XP + (idx % 24) * 16) = v[i];
        }
        if (lane < 32) *reinterpret_cast<half8*>(Ew + lane * EP + 7 * 16) = half8{0, 0, 0, 0, 0, 0, 0, 0};    // row 7 of E: zeros for good
    }
    lds_barrier();
    // operand side of the expand: MFMA row lm of strip s = image row 4 s + (lm >> 3), pixel slot lm & 7 (row 7 / slot 7 do not exist:
    // any valid pixel -- their results are never stored / stored as zero)
    half8 a[2][KSE];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int row = 4 * s + (lm >> 3), px = lm & 7;
        row = row < 7 ? row : 6;
        px = px < 7 ? px : 6;
        const unsigned char* ap = Dl + (row * 7 + px) * XP + g * 16;
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks) a[s][ks] = *reinterpret_cast<const half8*>(ap + ks * 32);
    }
    lds_barrier();                                           // the staged input is consumed: D may be written
    STAMP(1);

    // ---- phases 1 + 2: per wave, 32-channel tiles -----------------------------------------------------------------------------
    // A wave issues in order: a run of MFMAs blocks it for the matrix pipe's time, a run of Swish arithmetic leaves the pipe idle
    // (tools/probes/mb7_probe.hip, tile stamps: 1.5 us of MFMA runs + 2.3 us of VALU runs per tile).  So the tile is software-pipelined
    // by hand, in program order (sched_barrier keeps the compiler from regrouping):
    //   Swish + E of this tile | taps of block 0 | epilogue of block 0 | taps of block 1 | epilogue of block 1 BESIDE the expand MFMAs
    //   of the wave's NEXT tile
    // -- that epilogue is cut into 8 slots (one packed Swish each), the 24 MFMAs are dealt out over the slots.  (The epilogue of block 0
    // beside the taps of block 1 as well needs both blocks' accumulators, rows and sequences at once: 41 registers spilled.)
    const int cl = lane >> 2, j = lane & 3;                  // taps: lane = channel cl of a 16-channel block x column j (output rows 2j, 2j+1)
    unsigned eoff[6];                                        // byte offsets of input rows 2j - 2 + u in a channel's E rows (row 7 = zeros)
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int er = 2 * j - 2 + u;
        eoff[u] = unsigned((er < 0 || er > 6) ? 7 : er) * 16u;
    }
    const bool row1_ok = 2 * j + 1 < 7;                      // (column 3's second row is row 7)
    const bool hi_i = (lane & 2) != 0;                       // i >> 1 of this lane's Toeplitz row i = lane & 3
    // tap sequences and biases are independent of the data: they travel one stage ahead of their use
    uint4v sq[K];
    auto load_seq = [&](int cb16) {                          // cb16: index of the 16-channel block
#pragma unroll
        for (int ky = KY0; ky < KY1; ++ky)
            sq[ky - KY0] = *reinterpret_cast<const uint4v*>(p.wds + ((size_t(cb16) * 5 + ky) * 16 + cl) * 2 + (lane & 1));
    };
    load_seq(t * 2);
    float bias_e = p.be[t * 32 + lm], bd0 = p.bd[t * 32 + cl], bd1 = p.bd[t * 32 + 16 + cl];
    float16v acc0, acc1;
    // one expand MFMA of the tile whose weights are in w: m = 2 ks + strip
    auto expand_mfma = [&](int m) {
        const int ks = m >> 1;
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][ks], w[ks], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ks], w[ks], acc0, 0, 0, 0);
    };
    auto expand_init = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bias_e;             // (BN bias as the accumulators' initial value)
    };
    expand_init();
#pragma unroll
    for (int m = 0; m < 2 * KSE; ++m) expand_mfma(m);
    for (; t < NTE; t += NW) {
        const int ch0 = t * 32;
        const bool more = t + NW < NTE;                      // (uniform)
        const int tn = more ? t + NW : t;                    // (the last tile prefetches itself: harmless)
        TSTAMP(0);
        // BN + Swish -> E[channel lm][image row][slots 4g..4g+3]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                if (4 * s + qq < 7) {
                    const float16v& acc = s ? acc1 : acc0;
                    const float2v y0 = swish2(float2v{acc[4 * qq], acc[4 * qq + 1]});
                    const float2v y1 = swish2(float2v{acc[4 * qq + 2], acc[4 * qq + 3]});
                    half4 o;
                    o[0] = half_t(y0[0]);
                    o[1] = half_t(y0[1]);
                    o[2] = half_t(y1[0]);
                    o[3] = g ? half_t(0) : half_t(y1[1]);    // pixel slot 7 is 'SAME' padding of the EXPANDED tensor
                    *reinterpret_cast<half4*>(Ew + lm * EP + (4 * s + qq) * 16 + g * 8) = o;
                }
            }
        // the next tile's weights and biases
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KSE; ++ks) w[ks] = p.wep[(ks * NTE + tn) * 64 + lane];
        }
        const float bdv[2] = {bd0, bd1};
        bias_e = p.be[tn * 32 + lm];
        bd0 = p.bd[tn * 32 + cl];
        bd1 = p.bd[tn * 32 + 16 + cl];
        wave_lds_sync();
        TSTAMP(1);

        half8 v[K + 1];
        float4v dacc[2][2];                                  // [output row of the pair][x-group]
        auto load_rows = [&](int cb) {
            const unsigned char* ep = Ew + (cb * 16 + cl) * EP;
#pragma unroll
            for (int u = KY0; u <= KY1; ++u) v[u - KY0] = *reinterpret_cast<const half8*>(ep + eoff[u]);
#pragma unroll
            for (int dl = 0; dl < 2; ++dl)
#pragma unroll
                for (int xg = 0; xg < 2; ++xg) dacc[dl][xg] = float4v{bdv[cb], bdv[cb], bdv[cb], bdv[cb]};
        };
        // tap MFMA m of the block whose rows and sequences are loaded: m = (ky * 2 + dl) * 4 + q
        auto tap_mfma = [&](int m) {
            const int kyi = m >> 3, dl = (m >> 2) & 1, q = m & 3;
            // Toeplitz rows of this lane (output pixel i = lane & 3 of a group) for input chunks rel = 0, 1, 2 (pixels 4 (xg + rel - 1) ..):
            // dwords V[0..5] = the lane's tap sequence S placed at dword 1 + (i >> 1)
            const uint4v S = sq[kyi];
            const unsigned V1 = hi_i ? 0u : S[0], V2 = hi_i ? S[0] : S[1], V3 = hi_i ? S[1] : S[2], V4 = hi_i ? S[2] : 0u;
            const half4 A0 = as_half4(0u, V1), A1 = as_half4(V2, V3), A2 = as_half4(V4, 0u);
            const half8 r = v[kyi + dl];
            const half4 b0 = {r[0], r[1], r[2], r[3]}, b1 = {r[4], r[5], r[6], r[7]};
            if (q == 0) dacc[dl][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, b0, dacc[dl][0], 0, 0, 0);
            if (q == 1) dacc[dl][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(A0, b0, dacc[dl][1], 0, 0, 0);
            if (q == 2) dacc[dl][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, b1, dacc[dl][0], 0, 0, 0);
            if (q == 3) dacc[dl][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, b1, dacc[dl][1], 0, 0, 0);
        };
        constexpr int NTAP = K * 8;                          // tap MFMAs per block
        // epilogue slot k of block cb: BN + Swish of two values -> D[pixel][channel], channel sums (f32, before the binary16 rounding)
        float sumr[2];
        auto epi_slot = [&](int cb, int k) {
            const int dl = k >> 2, xg = (k >> 1) & 1, hf = k & 1;
            const int ch = ch0 + cb * 16 + cl;
            const float2v y = swish2(float2v{dacc[dl][xg][2 * hf], dacc[dl][xg][2 * hf + 1]});
            if ((k & 3) == 0) sumr[dl] = 0.f;
            unsigned char* dp = Dl + ((2 * j + dl) * 7 + 4 * xg + 2 * hf) * DP + ch * 2;
            const bool ok = dl == 0 || row1_ok;
            sumr[dl] += y[0];
            if (ok) *reinterpret_cast<half_t*>(dp) = half_t(y[0]);
            if (4 * xg + 2 * hf + 1 < 7) {                   // (pixel 7 does not exist)
                sumr[dl] += y[1];
                if (ok) *reinterpret_cast<half_t*>(dp + DP) = half_t(y[1]);
            }
        };
        auto epi_finish = [&](int cb) {
            float sum = sumr[0] + (row1_ok ? sumr[1] : 0.f);
            sum += quad_xor1(sum);                           // the 4 columns of the block: (s0 + s1) + (s2 + s3)
            sum += quad_xor2(sum);
            if (j == 0) s_sum[ch0 + cb * 16 + cl] = sum;
        };

        load_rows(0);
#pragma unroll
        for (int m = 0; m < NTAP; ++m) tap_mfma(m);
        TSTAMP(2);
        load_seq(t * 2 + 1);                                 // block 1's sequences travel during block 0's epilogue
#pragma unroll
        for (int k = 0; k < 8; ++k) epi_slot(0, k);
        epi_finish(0);
        load_rows(1);
#pragma unroll
        for (int m = 0; m < NTAP; ++m) tap_mfma(m);
        load_seq(tn * 2);                                    // ... and the next tile's during block 1's
        TSTAMP(3);
        if (more) {
            expand_init();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int m = 0; m < 2 * KSE; ++m)
                    if (m * 8 / (2 * KSE) == k) expand_mfma(m);
                epi_slot(1, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) epi_slot(1, k);
        }
        epi_finish(1);
        TSTAMP(4);
    }
    STAMP(2);

    // ---- phase 3: squeeze-excite ------------------------------------------------------------------------------------------------
    // reduce: wave w takes channels 144 w ..; lane (ci = lane & 7, jg = lane >> 3 < 6) multiplies channels 8 i + ci by outputs 8 jg ..
    const int ci = lane & 7, jg = lane >> 3;
    half8 w1v[18];
    {
        const int jgc = jg < 6 ? jg : 5;
#pragma unroll
        for (int i = 0; i < 18; ++i) w1v[i] = p.w1p[size_t(wave * 144 + i * 8 + ci) * 6 + jgc];
    }
    lds_barrier();                                           // D and the channel sums are complete
    STAMP(3);
    if (p.dbg_dw != nullptr) {
        half_t* dst = p.dbg_dw + size_t(crop) * PX * CEXP;
        for (int i = tid; i < PX * CEXP / 8; i += NTHR) {
            const int px = i / (CEXP / 8), c8 = i % (CEXP / 8);
            reinterpret_cast<half8*>(dst)[i] = *reinterpret_cast<const half8*>(Dl + px * DP + c8 * 16);
        }
    }
    {
        float r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const float m = s_sum[wave * 144 + i * 8 + ci];
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] = fmaf(m, float(w1v[i][e]), r8[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = r8[e];
            v += quad_xor1(v);
            v += quad_xor2(v);
            v += __shfl_xor(v, 4, 64);
            r8[e] = v;
        }
        if (ci == 0 && jg < 6) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_rp[wave * RSE + jg * 8 + e] = r8[e];
        }
    }
    // the excite rows and biases of this lane's channels (independent of the data): channels tid, tid + 512, tid + 1024
    half8 w2v[3][6];
    float b2v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = tid + q * NTHR;
        if (c < CEXP) {
#pragma unroll
            for (int u = 0; u < 6; ++u) w2v[q][u] = p.w2p[size_t(u) * CEXP + c];
            b2v[q] = p.b2[c];
        }
    }
    const float b1v = p.b1[tid < RSE ? tid : 0];
    // ... and the project's first weight fragments (phase 4): K split over the wave pairs (kq), output tiles over the two waves of a pair (nh)
    const int kq = wave >> 1, nh = wave & 1;
    constexpr int NKS = KSP / 4;                             // 18 k-steps per wave
    constexpr int PD = NTW == 3 ? 4 : 3;                     // k-steps of weight fragments in flight (12 / 15 KB per wave)
    const half8* wsrc = p.wpp + (size_t(kq) * NKS * NTP + nh * NTW) * 64 + lane;
    half8 wq[PD][NTW];
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) wq[d][tt] = wsrc[(size_t(d) * NTP + tt) * 64];
    lds_barrier();
    if (tid < RSE) {
        const float* rp = s_rp + tid;
        const float tot = ((rp[0] + rp[RSE]) + (rp[2 * RSE] + rp[3 * RSE])) + ((rp[4 * RSE] + rp[5 * RSE]) + (rp[6 * RSE] + rp[7 * RSE]));
        s_r[tid] = swish_f<true>(__fadd_rn(__fmul_rn(tot, p.inv_hw), b1v));
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = tid + q * NTHR;
        if (c < CEXP) {
            float t0 = b2v[q], t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                t0 = fmaf(s_r[u * 8 + 0], float(w2v[q][u][0]), t0);
                t1 = fmaf(s_r[u * 8 + 1], float(w2v[q][u][1]), t1);
                t2 = fmaf(s_r[u * 8 + 2], float(w2v[q][u][2]), t2);
                t3 = fmaf(s_r[u * 8 + 3], float(w2v[q][u][3]), t3);
                t0 = fmaf(s_r[u * 8 + 4], float(w2v[q][u][4]), t0);
                t1 = fmaf(s_r[u * 8 + 5], float(w2v[q][u][5]), t1);
                t2 = fmaf(s_r[u * 8 + 6], float(w2v[q][u][6]), t2);
                t3 = fmaf(s_r[u * 8 + 7], float(w2v[q][u][7]), t3);
            }
            const half_t gt = half_t(sigmoid_f<true>((t0 + t1) + (t2 + t3)));
            s_gate[c] = gt;
            if (p.dbg_gate != nullptr) p.dbg_gate[size_t(crop) * CEXP + c] = gt;
        }
    }
    lds_barrier();
    STAMP(4);

    // ---- phase 4: project -----------------------------------------------------------------------------------------------------------
    float16v pacc[2][NTW];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[s][tt][r] = 0.f;
    {
        const int m1 = 32 + lm < PX ? 32 + lm : PX - 1;      // strip 1: pixels 32..48 (columns past 48: any pixel, never stored)
        const unsigned char* d0 = Dl + lm * DP + (kq * NKS) * 32 + g * 16;
        const unsigned char* d1 = Dl + m1 * DP + (kq * NKS) * 32 + g * 16;
        const unsigned char* gp = smem + OFF_GATE + (kq * NKS) * 32 + g * 16;
#pragma unroll
        for (int k = 0; k < NKS; ++k) {
            const half8 gt = *reinterpret_cast<const half8*>(gp + k * 32);
            const half8 b0 = *reinterpret_cast<const half8*>(d0 + k * 32) * gt;
            const half8 b1 = *reinterpret_cast<const half8*>(d1 + k * 32) * gt;
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) {
                const half8 wf = wq[k % PD][tt];
                pacc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, b0, pacc[0][tt], 0, 0, 0);
                pacc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, b1, pacc[1][tt], 0, 0, 0);
            }
            if (k + PD < NKS) {
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) wq[k % PD][tt] = wsrc[(size_t(k + PD) * NTP + tt) * 64];
            }
        }
    }
    lds_barrier();                                           // D is dead: its region takes the partial accumulators
    STAMP(5);

    // ---- phase 5: combine (k0 + k1) + (k2 + k3), + BN bias, + skip, store -- rounds of (strip, <= 3 tiles per wave) ----------------------
    // A round's pieces (nh, tile, qq, lane: 4 channels x 1 pixel) are an exact multiple of the workgroup: every lane takes `per` of them;
    // their bias and skip operands are requested before the round's barrier.
    float4v* const cb4 = reinterpret_cast<float4v*>(smem);
    const half_t* xin = p.x + size_t(crop) * PX * CIN;
    half_t* outp = p.out + size_t(crop) * PX * COUT;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c0 = 0; c0 < NTW; c0 += TPR) {
            const int nt = NTW - c0 < TPR ? NTW - c0 : TPR;  // tiles of this round (compile-time after unrolling)
            const int per = 2 * nt * 4 * 64 / NTHR;          // pieces per lane: nt
            float4v bv[TPR];
            half4 rv[TPR];
            int mm[TPR], nn[TPR], src[TPR];
#pragma unroll
            for (int u = 0; u < TPR; ++u)
                if (u < per) {
                    const int it = tid + u * NTHR;
                    const int l = it & 63, qq = (it >> 6) & 3, rest = it >> 8;
                    const int tt = rest % nt, h = rest / nt;
                    mm[u] = s * 32 + (l & 31);
                    nn[u] = (h * NTW + c0 + tt) * 32 + 8 * qq + 4 * (l >> 5);
                    src[u] = ((h * TPR + tt) * 4 + qq) * 64 + l;
                    bv[u] = *reinterpret_cast<const float4v*>(p.bp + nn[u]);
                    if constexpr (RES) rv[u] = *reinterpret_cast<const half4*>(xin + size_t(mm[u] < PX ? mm[u] : 0) * CIN + nn[u]);
                }
#pragma unroll
            for (int tt = 0; tt < TPR; ++tt)
                if (tt < nt) {
                    const float16v& pa = pacc[s][c0 + tt < NTW ? c0 + tt : NTW - 1];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        cb4[((wave * TPR + tt) * 4 + qq) * 64 + lane] = float4v{pa[4 * qq], pa[4 * qq + 1], pa[4 * qq + 2], pa[4 * qq + 3]};
                }
            lds_barrier();
#pragma unroll
            for (int u = 0; u < TPR; ++u)
                if (u < per) {
                    float4v part[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) part[q] = cb4[q * (2 * TPR * 4 * 64) + src[u]];
                    float4v y = (part[0] + part[1]) + (part[2] + part[3]);
                    y = y + bv[u];
                    if constexpr (RES) y = y + float4v{float(rv[u][0]), float(rv[u][1]), float(rv[u][2]), float(rv[u][3])};
                    if (mm[u] < PX)
                        *reinterpret_cast<half4*>(outp + size_t(mm[u]) * COUT + nn[u]) = half4{half_t(y[0]), half_t(y[1]), half_t(y[2]), half_t(y[3])};
                }
            if (!(s == 1 && c0 + TPR >= NTW)) lds_barrier();  // (the next round overwrites the buffers)
        }
    STAMP(6);
}

struct OncePerDeviceMb7 {
    std::atomic<bool> done[64];
    OncePerDeviceMb7() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
};

template <int K, int NTP, bool RES>
void launch_mb7_t(const Mb7Args& a, hipStream_t stream) {
    Mb7Params p{};
    p.x = static_cast<const half_t*>(a.x);
    p.wep = static_cast<const half8*>(a.wep);
    p.be = a.be;
    p.wds = static_cast<const uint4*>(a.wds);
    p.bd = a.bd;
    p.w1p = static_cast<const half8*>(a.w1p);
    p.b1 = a.b1;
    p.w2p = static_cast<const half8*>(a.w2p);
    p.b2 = a.b2;
    p.wpp = static_cast<const half8*>(a.wpp);
    p.bp = a.bp;
    p.out = static_cast<half_t*>(a.out);
    p.dbg_dw = static_cast<half_t*>(a.dbg_dw);
    p.dbg_gate = static_cast<half_t*>(a.dbg_gate);
    p.inv_hw = 1.0f / float(PX);
    static OncePerDeviceMb7 attr;
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    auto kern = &whenet_mb7_kernel<K, NTP, RES>;
    if (dev >= 0 && dev < 64 && !attr.done[dev].load(std::memory_order_acquire)) {
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(a.n), dim3(NTHR), LDS_BYTES, stream, p);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace

bool mb7_supported(int dtype, int k, int s, int H, int Cin, int Cexp, int R, int Cout, bool skip) {
    return dtype == WHENET_F16 && (k == 3 || k == 5) && s == 1 && H == 7 && Cin == CIN && Cexp == CEXP && R == RSE &&
           ((Cout == 192 && k == 5 && skip) || (Cout == 320 && k == 3 && !skip));
}

// The depthwise kernel as tap SEQUENCES: per (16-channel block, tap row ky of the 5 x 5 embedding, channel, parity) 8 halfs --
// parity 0: (w0 w1 w2 w3 w4 0 0 0), parity 1: (0 w0 w1 w2 w3 w4 0 0) -- from which lane i of a Toeplitz block takes parity i & 1 and
// places its dwords at dword 1 + (i >> 1) of the 6-dword row  A_i[n] = w[n - i - 2],  n = 4 rel + e  (front2.hip's pack_dw_toeplitz with
// xs = 2; a 3 x 3 kernel sits in the middle of the 5 x 5 one: same products, zero rows skipped by the kernel).
std::vector<half_t> pack_mb7_taps(const std::vector<float>& w, int k, int C) {
    WHENET_REQUIRE((k == 3 || k == 5) && C % 16 == 0 && int(w.size()) == k * k * C, WHENET_EINVAL, "pack_mb7_taps: bad shape");
    const int o = (5 - k) / 2;
    std::vector<half_t> out(size_t(C / 16) * 5 * 16 * 2 * 8, half_t(0));
    for (int c = 0; c < C; ++c)
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                const half_t v = half_t(w[size_t(ky * k + kx) * C + c]);
                const size_t base = ((size_t(c / 16) * 5 + (ky + o)) * 16 + (c % 16)) * 2;
                out[(base + 0) * 8 + (kx + o)] = v;
                out[(base + 1) * 8 + (kx + o) + 1] = v;
            }
    return out;
}

// Squeeze-excite kernels in binary16: reduce [C][6] half8 (element e of vector jg <-> output 8 jg + e), excite [6][C] half8.
void pack_mb7_se(const std::vector<float>& w1t /* [R][C] */, const std::vector<float>& w2 /* [R][C] */, int C, int R,
                 std::vector<half_t>* w1p, std::vector<half_t>* w2p) {
    WHENET_REQUIRE(R == RSE && int(w1t.size()) == R * C && int(w2.size()) == R * C, WHENET_EINVAL, "pack_mb7_se: bad shape");
    w1p->assign(size_t(C) * RSE, half_t(0));
    w2p->assign(size_t(C) * RSE, half_t(0));
    for (int c = 0; c < C; ++c)
        for (int jo = 0; jo < R; ++jo) {
            (*w1p)[(size_t(c) * 6 + jo / 8) * 8 + jo % 8] = half_t(w1t[size_t(jo) * C + c]);
            (*w2p)[(size_t(jo / 8) * C + c) * 8 + jo % 8] = half_t(w2[size_t(jo) * C + c]);
        }
}

void launch_mb7(const Mb7Args& a, hipStream_t stream) {
    WHENET_REQUIRE(a.n >= 1 && (a.Cout == 192 || a.Cout == 320) && (a.k == 3 || a.k == 5), WHENET_EINVAL, "mb7: blocks 13-16 only");
    if (a.k == 5 && a.Cout == 192 && a.skip) launch_mb7_t<5, 6, true>(a, stream);
    else if (a.k == 3 && a.Cout == 320 && !a.skip) launch_mb7_t<3, 10, false>(a, stream);
    else throw Error(WHENET_EINVAL, "mb7: no instantiation for this block");
}

std::string kernel_name_mb7(int k, int Cout, bool skip) {
    return "whenet_mb7_kernel<" + std::to_string(k) + ", " + std::to_string(Cout / 32) + (skip ? ", true>" : ", false>");
}

}  // namespace whenet
