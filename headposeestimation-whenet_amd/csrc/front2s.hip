// MBConv "front" for float32 STORAGE (WHENET_F32S, round 6): expand 1x1 conv + BN + Swish -> depthwise kxk conv + BN + Swish in
// ONE kernel with BOTH convolutions on the matrix cores -- front2.hip's scheme at the parity-grade precision.
//
// Reference: efficientnet 0.0.4 MBConvBlock, blocks 2..12 (/root/reference/whenet.py:8):
// Conv2D(in*6, 1x1, no bias) -> BN -> Swish -> DepthwiseConv2D(k, s, 'same') -> BN -> Swish.
//
// Why: front.hip's float instantiation runs the k*k taps as f32 VALU FMAs out of an LDS tile (28 instructions per output
// value for a 5x5 layer) and re-splits every activation into binary16 hi/lo halves in every expand task; the f32s fronts were
// 573 of the 1,123 us a 64-crop chain takes (profiles/r05/layers_f32s_b64.json) with the matrix pipe 92 % idle.
//
//   * expand: v_mfma_f32_32x32x16_f16 with PIXELS as MFMA rows and channels as columns, three products per 16 k
//     (w_lo * x_hi + w_hi * x_lo + w_hi * x_hi, f32 accumulation: device_math.h PwOps<float, true>, same host-split weight
//     images, snapshot.cpp::pack_pw_split).  The activation halves come either from a float32 tensor (split in registers, as
//     pw.hip does) or -- PRE -- from a tensor the producing kernel already stored as [hi Cin | lo Cin] binary16 per pixel
//     (same bytes as float32; the split is then paid once per value instead of once per (value, 32-channel tile)).
//   * the expanded tile lives in LDS channel-major, 16 bytes per group of 4 x-consecutive pixels:
//       TM = 2: four float32 values.  Taps = per-channel Toeplitz products on v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4
//               outer products per instruction, block = channel, exact f32 FMA chains): for 4 output pixels of a row,
//               out[i] += w[ky][x - S*i] * E[row][x], one instruction per input pixel x and ky, operand B = one dword of a
//               16-byte LDS read that serves 4 x and all ky.  No VALU instruction in the taps, no rounding beyond f32.
//       TM = 1: [hi x 4 | lo x 4] binary16.  Taps = three v_mfma_f32_4x4x4_16B_f16 per (ky, 4-pixel chunk) against host-split
//               Toeplitz images (3/4 of TM = 2's matrix instructions, +2.5 VALU instructions per expanded value for the split).
//     Which one is faster is a per-layer measurement (tools/probes/front2s_probe.hip; front2s_tuned.inc).
//   * a column = one x-group x 7 output rows, swept once: every 16-byte read feeds K ky into 7 rolling accumulators;
//   * outputs leave straight from the accumulator registers: a lane holds float32 values of ONE channel, a store instruction
//     writes 4 columns x 16 channels = four 64-byte runs (front2.hip needs an LDS stage to pair binary16 channels; here a
//     dword is a channel).
// Tile geometry, strips, 'SAME' zero handling, squeeze-excite shares: exactly front2.hip's (same consumer kernels; the order of
// every sum depends on the layer only).
//
// HBM bytes per crop: H^2*Cin*4 (x chunks, L2 hits) + Ho^2*Cexp*4 written once.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

namespace whenet {

namespace {

constexpr int RL2S = 7;                 // output rows per tap column

__device__ __forceinline__ int fdiv2s(int q, float rinv) { return int((float(q) + 0.5f) * rinv); }     // see front.hip

struct F2SParams {
    const unsigned char* x;           // [n,H,H,Cin] float32, or (PRE) [n,H,H][hi Cin | lo Cin] binary16
    const half_t* weps;               // expand weights, [hi image | lo image] (snapshot.cpp::pack_pw_split)
    const float* be;
    const float4v* wdt;               // tap operand image, 16 bytes per lane (pack_dw_toeplitz_s)
    const float* bd;
    float* out;
    float* rpart;
    const float* w1t;
    int H, Ho, Cin, Cexp, pad, NTe;
    int CC, TH, TXG, tiles_x, EH, EWp, RP, CP;
    int off_stage, off_red, off_sum;
    int R, RPse;
    unsigned lo_off;                  // 16-byte fragments between the hi and the lo image of weps
    float wsi_e, ws_e;                // 2^-shift / 2^shift of the scaled expand weights
    float wsi_d, ws_d;                // the same for the depthwise taps (TM = 1; 1 otherwise)
};

// byte offset of channel c's plane in the tile.  The 16 lanes of a ds_read_b128 group are channels {0,3,5,6} or {1,2,4,7} of a
// block x 4 columns (MI355X_MICROARCH.md, LDS: lane groups of ds_read_b128); they cover all 64 banks when
//   S = 1 (a channel's 4 columns are 64 contiguous bytes): the pitch CP is 64 mod 128 -- c * CP mod 256 is then a different
//          quarter for the four channels of a group;
//   S = 2 (16 bytes every 32): CP is a multiple of 256 and the plane is rotated by channel bits 1 and 2 -> offsets 0 / 16 / 128 / 144.
template <int S>
__device__ __forceinline__ int ch_off(int c, int CP) {
    if constexpr (S == 1) return c * CP;
    else return c * CP + ((c >> 1) & 1) * 16 + ((c >> 2) & 1) * 128;
}

struct Raw32 { float4v q0, q1; };      // 32 bytes of a lane's activation operand for one k-step, as loaded

// BN + Swish of one expand task (32 pixels x 32 channels; this lane: channel ch, four runs of 4 consecutive pixels)
// into the tile.  EDGE: the strip hangs over the in-image rows / groups -- those runs are skipped.
template <int TM, bool EDGE>
__device__ __forceinline__ void expand_store_s(const float16v& acc, float wsi, unsigned char* ep, const int (&eoff)[4],
                                               const int (&drd)[4], const int (&dcd)[4], int nr_left, int ng_left) {
    const float2v ws2 = {wsi, wsi};
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        if (!EDGE || (drd[qq] < nr_left && dcd[qq] < ng_left)) {
            const float2v y0 = swish2(float2v{acc[4 * qq], acc[4 * qq + 1]} * ws2);
            const float2v y1 = swish2(float2v{acc[4 * qq + 2], acc[4 * qq + 3]} * ws2);
            if constexpr (TM == 2) {
                *reinterpret_cast<float4v*>(ep + eoff[qq]) = float4v{y0[0], y0[1], y1[0], y1[1]};
            } else {
                half4 hi, lo;
                hi[0] = half_t(y0[0]);
                hi[1] = half_t(y0[1]);
                hi[2] = half_t(y1[0]);
                hi[3] = half_t(y1[1]);
                lo[0] = half_t(y0[0] - float(hi[0]));
                lo[1] = half_t(y0[1] - float(hi[1]));
                lo[2] = half_t(y1[0] - float(hi[2]));
                lo[3] = half_t(y1[1] - float(hi[3]));
                half8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = hi[e];
                    v[4 + e] = lo[e];
                }
                *reinterpret_cast<half8*>(ep + eoff[qq]) = v;
            }
        }
    }
}

// pixel px (0..3) of the 16-byte group at gp <- 0, in the tile's element form
template <int TM>
__device__ __forceinline__ void zero_pixel(unsigned char* gp, int px) {
    if constexpr (TM == 2) {
        reinterpret_cast<float*>(gp)[px] = 0.f;
    } else {
        reinterpret_cast<half_t*>(gp)[px] = half_t(0);
        reinterpret_cast<half_t*>(gp)[4 + px] = half_t(0);
    }
}

template <int K, int S, int KS, int NTHR, int TM, bool PRE>
__global__ __launch_bounds__(NTHR) void whenet_front2s_kernel(const F2SParams p) {
    static_assert(TM == 1 || TM == 2, "tap mode");
    constexpr int NCH = (3 * S + K + 3) / 4;          // 4-pixel input chunks a 4-pixel output group reads
    constexpr int NER = (RL2S - 1) * S + K;           // input rows of a column
    constexpr int NWAVE = NTHR / 64;
    constexpr int PF = KS <= 4 ? KS : 4;              // k-steps of activation operands that travel together (8 registers each)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: the task / item loops are uniform)
    const int g = lane >> 5, lm = lane & 31;
    const int tile = blockIdx.x;
    const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const int c0 = blockIdx.y * p.CC;
    const int ccur = (p.Cexp - c0 < p.CC) ? (p.Cexp - c0) : p.CC;
    const int b = blockIdx.z;
    const int H = p.H, Cin = p.Cin, RP = p.RP, CP = p.CP;
    const int oy0 = tyi * p.TH, ox0 = txi * p.TXG * 4;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;

    // ---- the in-image part of the tile, in tile coordinates (er, ex) = (iy - iy0, ix - ix0) -----------------
    const int er_lo = iy0 < 0 ? -iy0 : 0, er_hi = (iy0 + p.EH < H) ? p.EH : H - iy0;
    const int ex_lo = ix0 < 0 ? -ix0 : 0, ex_hi = (ix0 + p.EWp < H) ? p.EWp : H - ix0;
    const int g_lo = ex_lo >> 2, g_hi = (ex_hi + 3) >> 2;
    const int NG = g_hi - g_lo, NR = er_hi - er_lo;
    // A strip (the 32 rows of one MFMA tile) = SR rows x SC groups of 4 pixels, SR * SC = 8 (front2.hip)
    const int SCL = NG <= 2 ? 1 : (NG <= 4 ? 2 : 3), SC = 1 << SCL, SRL = 3 - SCL, SR = 1 << SRL;
    const int nsc = (NG + SC - 1) >> SCL, nsr = (NR + SR - 1) >> SRL;
    const int nstrip = nsc * nsr, ntile = (ccur + 31) >> 5, ntask = nstrip * ntile;

    const unsigned char* xb = p.x + size_t(b) * H * H * Cin * 4;
    const half8* wf0 = reinterpret_cast<const half8*>(p.weps) + size_t(c0 >> 5) * 64 + lane;

    // lane tables: operand side (MFMA row = lane & 31 -> pixel), accumulator side (run qq -> group 2 qq + g)
    const int ua = lm >> 2;
    const int pixoff_a = (ua >> SCL) * H + 4 * (ua & (SC - 1)) + (lm & 3);
    int eoff[4], drd[4], dcd[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int u = 2 * qq + g;
        drd[qq] = u >> SCL;
        dcd[qq] = u & (SC - 1);
        eoff[qq] = drd[qq] * RP + dcd[qq] * 16;
    }
    const int npix1 = H * H - 1;
    const unsigned cin4 = unsigned(Cin) * 4u;
    auto a_offset = [&](int rb, int cbk) -> unsigned {          // byte offset of this lane's operand row in the crop
        const int pb = (iy0 + er_lo + (rb << SRL)) * H + ix0 + 4 * (g_lo + (cbk << SCL));
        int pl = pb + pixoff_a;
        pl = pl < 0 ? 0 : (pl > npix1 ? npix1 : pl);           // rows / pixels outside the image: any valid address
        return __umul24(unsigned(pl), cin4) + unsigned(g) * (PRE ? 16u : 32u);      // (they are never stored, or fixed up below)
    };

    // ---- expand: tasks (channel tile, strip), a contiguous range per wave --------------------------------------
    const int t_begin = (wave * ntask) / NWAVE, t_end = ((wave + 1) * ntask) / NWAVE;
    half8 whi[KS], wlo[KS];
    Raw32 aq[PF];
    float16v bias16;                                           // splat of the lane's BN bias / wsi: the accumulators' initial value
    auto load_w = [&](int tl) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            whi[ks] = wf0[(size_t(ks) * p.NTe + tl) * 64];
            wlo[ks] = wf0[(size_t(ks) * p.NTe + tl) * 64 + p.lo_off];
        }
        const int ch = tl * 32 + lm;
        const float bias = (ch < ccur) ? p.be[c0 + ch] * p.ws_e : 0.f;      // (exact: ws_e is a power of two)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias16[r] = bias;
    };
    // k beyond Cin (Cin = 24 / 40: the upper half of the last k-step): the packed weights are zero there, but the bytes past a
    // pixel row are the next pixel's -- or, for the last pixel of the last crop, whatever follows the tensor: 0 x NaN would
    // poison the accumulators, so those lanes' operand is zeroed where it is consumed (front2.hip)
    const bool ktail = (Cin & 15) != 0 && g == 1;
    auto load_a = [&](unsigned off, int ks0) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (ks0 + u < KS) {
                if constexpr (PRE) {
                    aq[u].q0 = *reinterpret_cast<const float4v*>(xb + off + (ks0 + u) * 32);                  // hi: 8 halves
                    aq[u].q1 = *reinterpret_cast<const float4v*>(xb + off + unsigned(Cin) * 2u + (ks0 + u) * 32);   // lo
                } else {
                    aq[u].q0 = *reinterpret_cast<const float4v*>(xb + off + (ks0 + u) * 64);
                    aq[u].q1 = *reinterpret_cast<const float4v*>(xb + off + (ks0 + u) * 64 + 16);
                }
            }
    };
    auto mult = [&](const Raw32& a, int ks, const float16v& c) -> float16v {
        Raw32 r = a;
        if (ks == KS - 1 && ktail) r = Raw32{float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
        half8 hi, lo;
        if constexpr (PRE) {
            hi = __builtin_bit_cast(half8, r.q0);
            lo = __builtin_bit_cast(half8, r.q1);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half_t h0 = half_t(r.q0[e]), h1 = half_t(r.q1[e]);
                hi[e] = h0;
                hi[4 + e] = h1;
                lo[e] = half_t(r.q0[e] - float(h0));
                lo[4 + e] = half_t(r.q1[e] - float(h1));
            }
        }
        float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, wlo[ks], c, 0, 0, 0);           // (small terms first)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, whi[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, whi[ks], acc, 0, 0, 0);
        return acc;
    };
    STAMP(0);
    int tl = 0, rb = 0, cbk = 0;                               // the task being computed
    int ti = t_begin, rbi = 0, cbki = 0;                       // the next task whose operands are requested
    unsigned aoff = 0;
    auto issue = [&]() {                                       // strip (rbi, cbki) -> operand registers, then step to the next strip
        if (ti < t_end) {
            aoff = a_offset(rbi, cbki);
            load_a(aoff, 0);
        }
        ++ti;
        if (++cbki == nsc) {
            cbki = 0;
            if (++rbi == nsr) rbi = 0;                         // (next channel tile: the strips start over)
        }
    };
    const bool fix_l = ex_lo > 0, fix_r = p.EWp > ex_hi;       // (tile touches the left / right image border, or has
                                                               //  slack groups behind it)
    if (t_begin < t_end) {
        tl = __builtin_amdgcn_readfirstlane(t_begin / nstrip);          // (keeps the loop state in scalar registers)
        const int st = t_begin - tl * nstrip;
        rb = __builtin_amdgcn_readfirstlane(st / nsc);
        cbk = st - rb * nsc;
        rbi = rb;
        cbki = cbk;
        load_w(tl);
        issue();
    }
    {   // rows of the tile outside the image are 'SAME' zeros of the EXPANDED tensor (top / bottom tiles only)
        const int nz = er_lo + (p.EH - er_hi);
        if (nz > 0) {                                           // (uniform)
            const float r_nz = __builtin_amdgcn_rcpf(float(nz));
            for (int pr = tid; pr < ccur * nz; pr += NTHR) {
                const int c = fdiv2s(pr, r_nz), hr = pr - c * nz;
                const int er = hr < er_lo ? hr : er_hi + (hr - er_lo);
                unsigned char* rowp = E + ch_off<S>(c, CP) + er * RP;
                for (int q = 0; q < RP; q += 16) *reinterpret_cast<float4v*>(rowp + q) = float4v{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    STAMP(1);
    for (int t = t_begin; t < t_end; ++t) {
        float16v acc = mult(aq[0], 0, bias16);
#pragma unroll
        for (int u = 1; u < PF; ++u) acc = mult(aq[u], u, acc);
#pragma unroll
        for (int ks = PF; ks < KS; ks += PF) {
            load_a(aoff, ks);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (ks + u < KS) acc = mult(aq[u], ks + u, acc);
        }
        // this task's place in the tile; the operand registers are free: the operands of task t + 1 take off
        const int ch = tl * 32 + lm;
        unsigned char* ep = E + ch_off<S>(ch, CP) + (er_lo + (rb << SRL)) * RP + (g_lo + (cbk << SCL)) * 16;
        unsigned char* ep0 = ep;                               // (row of the strip's corner, at its first group)
        const int nr_left = NR - (rb << SRL), ng_left = NG - (cbk << SCL), cbk_this = cbk;
        int tln = tl;
        if (++cbk == nsc) {
            cbk = 0;
            if (++rb == nsr) {
                rb = 0;
                ++tln;
            }
        }
        issue();
        if (t + 1 < t_end && tln != tl) load_w(tln);           // (rare: at most once or twice per wave)
        tl = tln;
        if (ch < ccur) {
            if (nr_left >= SR && ng_left >= SC) expand_store_s<TM, false>(acc, p.wsi_e, ep, eoff, drd, dcd, nr_left, ng_left);
            else expand_store_s<TM, true>(acc, p.wsi_e, ep, eoff, drd, dcd, nr_left, ng_left);
        }
        // 'SAME' zeros inside the rows this task just wrote: the pixels of a border group that lie left / right of the
        // image were computed from clamped addresses.  The wave that wrote them overwrites them (DS operations of a
        // wave execute in order: no barrier), lane <-> (channel lm, row g, g + 2, ..) of the strip.
        if (ch < ccur) {
            if (fix_l && cbk_this == 0) {                      // (uniform) strip holds the left border group
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int dr = g + 2 * dd;
                    if (dr < SR && dr < nr_left) {
                        unsigned char* gp = ep0 + dr * RP;     // the border group (g_lo) of this row
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < (ex_lo & 3)) zero_pixel<TM>(gp, q);
                    }
                }
            }
            if (fix_r && cbk_this == nsc - 1) {                // (uniform) strip holds the right border group and beyond
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int dr = g + 2 * dd;
                    if (dr < SR && dr < nr_left) {
                        unsigned char* row0 = ep0 + dr * RP - (g_lo + cbk_this * SC) * 16;      // group 0 of the tile row
                        // (every pixel up to the tile's edge: the groups behind the last one written hold whatever the LDS held)
                        for (int px = ex_hi; px < p.EWp; ++px) zero_pixel<TM>(row0 + (px >> 2) * 16, px & 3);
                    }
                }
            }
        }
    }

    STAMP(2);
    // ---- depthwise taps: items (16-channel block, quad of columns), a contiguous range per wave ---------------
    const int ncb = (ccur + 15) >> 4;
    const int ncol = (p.TH / RL2S) * p.TXG, ncq = (ncol + 3) >> 2, nitem = ncb * ncq;
    const float r_txg = __builtin_amdgcn_rcpf(float(p.TXG));
    const int i_begin = (wave * nitem) / NWAVE, i_end = ((wave + 1) * nitem) / NWAVE;
    const int cl = lane >> 2, j = lane & 3;
    float4v A[K][NCH];                                         // 16 bytes per (ky, chunk): TM = 2 four floats, TM = 1 hi4 | lo4
    float bdv = 0.f;
    auto load_taps = [&](int cb) {
        const float4v* src = p.wdt + (size_t((c0 >> 4) + cb) * K * NCH) * 64 + lane;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) A[ky][ch] = src[(ky * NCH + ch) * 64];
        bdv = p.bd[c0 + cb * 16 + cl] * p.ws_d;
    };
    int cb = 0, cq = 0;
    if (i_begin < i_end) {
        cb = __builtin_amdgcn_readfirstlane(i_begin / ncq);
        cq = i_begin - cb * ncq;
        load_taps(cb);                                         // in flight across the barrier(s)
    }
    lds_barrier();
    // (the out-of-image pixels of the rows a wave expanded are zeroed by that wave: see the expand loop)
    STAMP(3);
    // this lane's first reduce-kernel values (used after the items: see the squeeze-excite half below)
    constexpr int W1V = 16;
    float w1v[W1V];
    {
        const int jo = tid >> 2, q = tid & 3;
#pragma unroll
        for (int i = 0; i < W1V; ++i) w1v[i] = 0.f;
        if (p.w1t != nullptr && wave * 16 < p.RPse) {          // (uniform; clamped addresses: no lane predicates)
            const float* wrow = p.w1t + size_t(jo < p.R ? jo : p.R - 1) * p.Cexp + c0;
#pragma unroll
            for (int i = 0; i < W1V; ++i) {
                const int c = q + 4 * i;
                w1v[i] = wrow[c < ccur ? c : ccur - 1];
            }
        }
    }
    float* s_red = reinterpret_cast<float*>(smem + p.off_red);          // [ncq][CC]
    float* s_sum = reinterpret_cast<float*>(smem + p.off_sum);          // [CC]
    unsigned char* outb = reinterpret_cast<unsigned char*>(p.out + size_t(b) * p.Ho * p.Ho * p.Cexp);
    const unsigned row_bytes = unsigned(p.Ho) * unsigned(p.Cexp) * 4u;

    for (int it = i_begin; it < i_end; ++it) {
        const int col = cq * 4 + j;                            // (this block's taps are in A: loaded in the prologue
        const int colc = col < ncol ? col : ncol - 1;          //  or behind the previous sweep)
        const int seg = fdiv2s(colc, r_txg), xgl = colc - seg * p.TXG;
        const int c = cb * 16 + cl;
        const unsigned char* bp = E + ch_off<S>(c, CP) + (seg * RL2S * S) * RP + xgl * (16 * S);
        float4v acc[RL2S];
        const float4v bd4 = float4v{bdv, bdv, bdv, bdv};       // (BN bias as the initial value: the C operand of a row's FIRST product --
        bool started[RL2S];                                    //  everything here unrolls, the flags fold at compile time)
#pragma unroll
        for (int r = 0; r < RL2S; ++r) started[r] = false;
#pragma unroll
        for (int er = 0; er < NER; ++er) {
            float4v bv[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) bv[ch] = *reinterpret_cast<const float4v*>(bp + er * RP + ch * 16);
            // (products in an order that puts consecutive matrix instructions on DIFFERENT accumulators -- one per ky -- instead of
            //  chains on one: a dependent 4x4 MFMA waits for its predecessor)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                constexpr int NT = TM == 2 ? 4 : 3;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        const int d = er - ky;
                        if (d >= 0 && d % S == 0 && d / S < RL2S) {
                            const float4v cacc = started[d / S] ? acc[d / S] : bd4;
                            if constexpr (TM == 2) {
                                acc[d / S] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[ky][ch][t], bv[ch][t], cacc, 0, 0, 0);
                            } else {
                                const half8 a8 = __builtin_bit_cast(half8, A[ky][ch]), b8 = __builtin_bit_cast(half8, bv[ch]);
                                const half4 ahi = {a8[0], a8[1], a8[2], a8[3]}, alo = {a8[4], a8[5], a8[6], a8[7]};
                                const half4 bhi = {b8[0], b8[1], b8[2], b8[3]}, blo = {b8[4], b8[5], b8[6], b8[7]};
                                acc[d / S] = __builtin_amdgcn_mfma_f32_4x4x4f16(t == 0 ? alo : ahi, t == 1 ? blo : bhi, cacc, 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {          // (the flags change behind the whole round of ky)
                        const int d = er - ky;
                        if (d >= 0 && d % S == 0 && d / S < RL2S) started[d / S] = true;
                    }
                }
            }
        }
        const int cb_this = cb, cq_this = cq;
        if (++cq == ncq) {                                     // the next block's taps travel during the epilogue
            cq = 0;
            ++cb;
            if (it + 1 < i_end) load_taps(cb);
        }
        // ---- BN + Swish, channel sums, and the way out: lane (channel cl, column j) holds 7 rows x 4 pixels of ONE channel; a
        // store instruction (row r, pixel i) writes 4 columns x 16 channels x 4 bytes = four 64-byte runs -----------------
        const bool okc = col < ncol;
        const int oxb = ox0 + 4 * xgl;
        bool ok[4];
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ok[i] = okc && oxb + i < p.Ho;
            m[i] = ok[i] ? 1.f : 0.f;
        }
        const unsigned px_bytes = unsigned(p.Cexp) * 4u;
        const unsigned obase = (__umul24(unsigned(oy0 + seg * RL2S), unsigned(p.Ho)) + unsigned(oxb)) * px_bytes +
                               unsigned(c0 + cb_this * 16 + cl) * 4u;
        float2v sum2 = {0.f, 0.f};
        const float2v m01 = {m[0], m[1]}, m23 = {m[2], m[3]};
        const float2v wd2 = {p.wsi_d, p.wsi_d};
#pragma unroll
        for (int r = 0; r < RL2S; ++r) {
            float2v x01 = float2v{acc[r][0], acc[r][1]}, x23 = float2v{acc[r][2], acc[r][3]};
            if constexpr (TM == 1) {
                x01 = x01 * wd2;
                x23 = x23 * wd2;
            }
            const float2v y01 = swish2(x01);
            const float2v y23 = swish2(x23);
            sum2 = y01 * m01 + sum2;
            sum2 = y23 * m23 + sum2;
            acc[r] = float4v{y01[0], y01[1], y23[0], y23[1]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ok[i]) {
#pragma unroll
                for (int r = 0; r < RL2S; ++r)
                    *reinterpret_cast<float*>(outb + obase + unsigned(r) * row_bytes + unsigned(i) * px_bytes) = acc[r][i];
            }
        }
        float sum = sum2[0] + sum2[1];
        sum += quad_xor1(sum);                                 // the 4 columns of the quad: (s0 + s1) + (s2 + s3)
        sum += quad_xor2(sum);
        if (j == 0) s_red[cq_this * p.CC + c] = sum;
    }
    STAMP(4);
    lds_barrier();
    STAMP(5);

    // ---- squeeze-excite, first half (as front.hip): the tile's channel sums, or this workgroup's share of the
    // reduce conv, in a fixed order --------------------------------------------------------------------------
    if (tid < ccur) {
        float t = 0.0f;
        for (int q = 0; q < ncq; ++q) t += s_red[q * p.CC + tid];
        if (p.w1t == nullptr) p.rpart[(size_t(b) * gridDim.x + tile) * p.Cexp + c0 + tid] = t;
        s_sum[tid] = t;
    }
    if (p.w1t == nullptr) {
        STAMP(6);
        return;
    }
    lds_barrier();
    {
        // 4 lanes per output j: lane q sums channels q, q+4, ..; combined (a0+a1)+(a2+a3)
        const int jo = tid >> 2, q = tid & 3;
        if (jo < p.RPse) {                                      // (whole quads of lanes)
            float accr = 0.0f;
            if (jo < p.R) {
                const float* wrow = p.w1t + size_t(jo) * p.Cexp + c0;
#pragma unroll
                for (int i = 0; i < W1V; ++i) {
                    const int c = q + 4 * i;
                    if (c < ccur) accr = fmaf(s_sum[c], w1v[i], accr);
                }
                for (int c = q + 4 * W1V; c < ccur; c += 4) accr = fmaf(s_sum[c], wrow[c], accr);
            }
            const float pair = accr + quad_xor1(accr);
            const float tot = pair + quad_xor2(pair);
            if (q == 0)
                p.rpart[((size_t(b) * gridDim.x + tile) * gridDim.y + blockIdx.y) * p.RPse + jo] = (jo < p.R) ? tot : 0.0f;
        }
    }
    STAMP(6);
}

struct OncePerDeviceS {
    std::atomic<bool> done[64];
    OncePerDeviceS() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
};

template <int K, int S, int KS, int NTHR, int TM, bool PRE>
void launch_f2s(const Front2sArgs& a, hipStream_t stream) {
    const Front2Plan& pl = a.plan;
    F2SParams p{};
    p.x = static_cast<const unsigned char*>(a.x);
    p.weps = static_cast<const half_t*>(a.weps);
    p.be = a.be;
    p.wdt = static_cast<const float4v*>(a.wdt);
    p.bd = a.bd;
    p.out = static_cast<float*>(a.out);
    p.rpart = a.rpart;
    p.w1t = a.w1t;
    p.H = a.H;  p.Ho = a.Ho;  p.Cin = a.Cin;  p.Cexp = a.Cexp;  p.pad = a.pad;  p.NTe = a.NTe;
    p.CC = pl.CC;  p.TH = pl.TH;  p.TXG = pl.TXG;  p.tiles_x = pl.tiles_x;
    p.EH = pl.EH;  p.EWp = pl.EWp;  p.RP = pl.RP;  p.CP = pl.CP;
    p.off_stage = pl.off_stage;  p.off_red = pl.off_red;  p.off_sum = pl.off_sum;
    p.R = a.R;  p.RPse = (a.R + 3) & ~3;
    p.lo_off = unsigned(a.KSe) * unsigned(a.NTe) * 64u;
    p.wsi_e = a.wsi;  p.ws_e = 1.0f / a.wsi;
    p.wsi_d = TM == 1 ? a.wsi_d : 1.0f;  p.ws_d = 1.0f / p.wsi_d;
    WHENET_REQUIRE(pl.lds_bytes <= 160 * 1024, WHENET_EINVAL, "front2s: the tile plan needs more than 160 KB of LDS");
    static OncePerDeviceS attr;
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr.done[dev].load(std::memory_order_acquire)) {
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_front2s_kernel<K, S, KS, NTHR, TM, PRE>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((whenet_front2s_kernel<K, S, KS, NTHR, TM, PRE>), dim3(pl.tiles_x * pl.tiles_y, pl.chunks, a.n), dim3(NTHR),
                       pl.lds_bytes, stream, p);
    WHENET_HIP_CHECK(hipGetLastError());
}

// the (tap mode, input form) pairs built into the library; the probe builds all four (-DWHENET_F2S_ALL)
#ifdef WHENET_F2S_ALL
#define F2S_FORMS(K, S, KS, NTHR)                                                            \
    if (a.tm == 1 && !a.pre) return launch_f2s<K, S, KS, NTHR, 1, false>(a, stream);         \
    if (a.tm == 1 && a.pre) return launch_f2s<K, S, KS, NTHR, 1, true>(a, stream);           \
    if (a.tm == 2 && !a.pre) return launch_f2s<K, S, KS, NTHR, 2, false>(a, stream);         \
    if (a.tm == 2 && a.pre) return launch_f2s<K, S, KS, NTHR, 2, true>(a, stream);
#else
#define F2S_FORMS(K, S, KS, NTHR)                                                            \
    if (a.tm == 1 && !a.pre) return launch_f2s<K, S, KS, NTHR, 1, false>(a, stream);         \
    if (a.tm == 2 && !a.pre) return launch_f2s<K, S, KS, NTHR, 2, false>(a, stream);
#endif

template <int NTHR>
void launch_f2s_shape(const Front2sArgs& a, hipStream_t stream) {
    const int key = a.k * 1000 + a.s * 100 + a.KSe;
    switch (key) {                                   // EfficientNet-B0's (kernel, stride, Cin / 16) shapes of blocks 2 - 12
        case 3201: F2S_FORMS(3, 2, 1, NTHR) break;       // b2
        case 3102: F2S_FORMS(3, 1, 2, NTHR) break;       // b3
        case 5202: F2S_FORMS(5, 2, 2, NTHR) break;       // b4
        case 5103: F2S_FORMS(5, 1, 3, NTHR) break;       // b5
        case 3203: F2S_FORMS(3, 2, 3, NTHR) break;       // b6
        case 3105: F2S_FORMS(3, 1, 5, NTHR) break;       // b7, b8
        case 5105: F2S_FORMS(5, 1, 5, NTHR) break;       // b9
        case 5107: F2S_FORMS(5, 1, 7, NTHR) break;       // b10, b11
        case 5207: F2S_FORMS(5, 2, 7, NTHR) break;       // b12
        default: break;
    }
    throw Error(WHENET_EINVAL, "front2s: unsupported (kernel, stride, Cin, tap mode, input form) combination");
}

}  // namespace

bool front2s_supported(int k, int s, int H, int Cin) {
    const int key = k * 1000 + s * 100 + ceil_div(Cin, 16);
    (void)H;
    for (int v : {3201, 3102, 5202, 5103, 3203, 3105, 5105, 5107, 5207})
        if (v == key) return true;
    return false;
}

// Toeplitz operand images of a depthwise kernel for the two tap modes (see the header comment), 16 bytes per lane:
//   [C / 16 blocks][ky][chunk][lane 0..63];  lane l <-> channel 16 * block + (l >> 2), output pixel i = l & 3,
//   element e <-> tap kx = 4 * chunk + e - s * i (zero outside 0..k-1).
//   tm = 2: four floats.   tm = 1: w * 2^shift = hi + lo in binary16, [hi x 4 | lo x 4]; *wsi = 2^-shift.
std::vector<float> pack_dw_toeplitz_s(const std::vector<float>& w, int k, int s, int C, int tm, float* wsi) {
    WHENET_REQUIRE(C % 16 == 0 && int(w.size()) == k * k * C && (tm == 1 || tm == 2), WHENET_EINVAL, "pack_dw_toeplitz_s: bad shape");
    const int nch = (3 * s + k + 3) / 4;
    std::vector<float> out(size_t(C / 16) * k * nch * 64 * 4, 0.f);
    double sc = 1.0;
    *wsi = 1.0f;
    if (tm == 1) {
        double mx = 0.0;
        for (float v : w) mx = std::max(mx, double(std::fabs(v)));
        int shift = 0;
        if (mx > 0.0) {
            shift = int(std::floor(std::log2(16384.0 / mx)));       // largest |w'| in [8192, 16384), as pack_pw_split
            shift = std::max(-24, std::min(24, shift));
        }
        sc = std::ldexp(1.0, shift);
        *wsi = float(std::ldexp(1.0, -shift));
    }
    for (int blk = 0; blk < C / 16; ++blk)
        for (int ky = 0; ky < k; ++ky)
            for (int ch = 0; ch < nch; ++ch)
                for (int l = 0; l < 64; ++l) {
                    float* o = &out[(((size_t(blk) * k + ky) * nch + ch) * 64 + l) * 4];
                    half_t* oh = reinterpret_cast<half_t*>(o);
                    for (int e = 0; e < 4; ++e) {
                        const int c = blk * 16 + (l >> 2), i = l & 3, kx = 4 * ch + e - s * i;
                        const float v = (kx >= 0 && kx < k) ? w[size_t(ky * k + kx) * C + c] : 0.f;
                        if (tm == 2) {
                            o[e] = v;
                        } else {
                            const double vs = double(v) * sc;
                            const half_t h = half_t(float(vs));
                            oh[e] = h;
                            oh[4 + e] = half_t(float(vs - double(float(h))));
                        }
                    }
                }
    return out;
}

Front2Plan make_front2s_plan(int k, int s, int Ho, int Cexp, int CC, int TH, int TXG, int threads) {
    const int OXG = ceil_div(Ho, 4), nch = (3 * s + k + 3) / 4;
    WHENET_REQUIRE(TH % RL2S == 0 && Ho % TH == 0 && TXG >= 1 && TXG <= OXG && (CC % 32 == 0 || CC == Cexp) && Cexp % 16 == 0 &&
                       (threads == 256 || threads == 512),
                   WHENET_EINVAL, "front2s: bad tile plan");
    Front2Plan p;
    p.threads = threads;
    p.CC = CC < Cexp ? CC : Cexp;
    p.TH = TH;
    p.TXG = TXG;
    p.xs = 0;
    p.tiles_x = ceil_div(OXG, TXG);
    p.tiles_y = Ho / TH;
    p.chunks = ceil_div(Cexp, p.CC);
    p.EH = (TH - 1) * s + k;
    p.EWp = 4 * (s * (TXG - 1) + nch);
    p.RP = p.EWp * 4;                                   // 16 bytes per 4-pixel group
    if (s == 1) {
        p.CP = (p.EH * p.RP + 63) / 128 * 128 + 64;     // smallest pitch >= the plane that is 64 mod 128 (ch_off())
    } else {
        p.CP = (p.EH * p.RP + 144 + 255) / 256 * 256;   // (+ the bank rotation of ch_off())
    }
    const int ncq = ceil_div((TH / RL2S) * TXG, 4);
    const size_t e_bytes = size_t(p.CC) * p.CP;
    p.off_stage = int((e_bytes + 15) & ~size_t(15));
    p.off_red = p.off_stage;                            // (no output stage: a lane stores its own float32 values)
    p.off_sum = p.off_red + ncq * p.CC * 4;
    p.lds_bytes = size_t(p.off_sum) + size_t(p.CC) * 4;
    return p;
}

namespace {
struct Tuned2s { int k, s, H, Cexp, CC, TH, TXG, tm, threads, use; };
const Tuned2s TUNED2S[] = {
#include "front2s_tuned.inc"
};
}  // namespace

// the layer's plan and tap mode (tm: 1 | 2); shapes outside the table: 32 channels, 7 rows, the widest tile that leaves two
// workgroups per CU, exact-f32 taps
Front2Plan plan_front2s(int k, int s, int H, int Ho, int Cexp, int* tm) {
    static const bool no_tuned = getenv("WHENET_FRONT_NO_TUNED") != nullptr;       // (probes only; read once)
    if (!no_tuned)
        for (const Tuned2s& t : TUNED2S)
            if (t.k == k && t.s == s && t.H == H && t.Cexp == Cexp) {
                if (tm) *tm = t.tm;
                return make_front2s_plan(k, s, Ho, Cexp, t.CC, t.TH, t.TXG, t.threads);
            }
    if (tm) *tm = 2;
    const int OXG = ceil_div(Ho, 4);
    for (int txg = OXG; txg >= 1; --txg) {
        const Front2Plan p = make_front2s_plan(k, s, Ho, Cexp, 32, RL2S, txg, 256);
        if (p.lds_bytes <= 80 * 1024 || txg == 1) return p;
    }
    throw Error(WHENET_EINVAL, "front2s: no tile plan fits");
}

// Layers on which this kernel beats whenet_front_kernel<float, .., true> (measured: front2s_tuned.inc)
bool front2s_preferred(int k, int s, int H, int Cexp) {
    for (const Tuned2s& t : TUNED2S)
        if (t.k == k && t.s == s && t.H == H && t.Cexp == Cexp) return t.use != 0;
    return false;
}

std::vector<Front2Plan> plan_front2s_candidates(int k, int s, int Ho, int Cexp) {
    std::vector<Front2Plan> out;
    const int OXG = ceil_div(Ho, 4);
    for (int CC : {32, 64, 96})
        for (int TH : {7, 14, 28})
            for (int TXG = 1; TXG <= OXG; ++TXG) {
                if (Ho % TH || (CC > Cexp && CC != 32)) continue;       // (ragged x tiles are candidates too)
                if (CC < Cexp && Cexp % CC && (Cexp % CC) % 16) continue;
                if (TXG < 2 && OXG > 1) continue;                        // (x tiles of one group: all halo)
                const Front2Plan p = make_front2s_plan(k, s, Ho, Cexp, CC, TH, TXG, 256);
                if (p.lds_bytes <= 150 * 1024) out.push_back(p);
            }
    return out;
}

void launch_front2s(const Front2sArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.KSe == ceil_div(a.Cin, 16), WHENET_EINVAL, "front2s: k-steps do not match Cin");
    WHENET_REQUIRE(a.weps != nullptr && a.wdt != nullptr && a.wsi > 0.f && (a.tm == 2 || a.wsi_d > 0.f), WHENET_EINVAL,
                   "front2s: missing split weight images");
    Front2sArgs b = a;
    if (b.plan.threads != 256) {              // the stage / sums offsets depend on the wave count
        b.plan = make_front2s_plan(a.k, a.s, a.Ho, a.Cexp, a.plan.CC, a.plan.TH, a.plan.TXG, a.plan.threads);
    }
#ifdef WHENET_F2S_ALL
    if (b.plan.threads == 512) return launch_f2s_shape<512>(b, stream);
#endif
    WHENET_REQUIRE(b.plan.threads == 256, WHENET_EINVAL, "front2s: the library carries the 256-lane form only");
    launch_f2s_shape<256>(b, stream);
}

std::string kernel_name_front2s(int k, int s, int kse, int threads, int tm, bool pre) {
    return "whenet_front2s_kernel<" + std::to_string(k) + ", " + std::to_string(s) + ", " + std::to_string(kse) + ", " +
           std::to_string(threads) + ", " + std::to_string(tm) + ", " + (pre ? "true" : "false") + ">";
}

}  // namespace whenet
