// MBConv "front", f16, round 3: expand 1x1 conv + BN + Swish -> depthwise kxk conv + BN + Swish in ONE kernel with
// BOTH convolutions on the matrix cores.  (front.hip keeps the f32 parity configuration: its depthwise taps are f32
// VALU FMAs.)
//
// Reference: efficientnet 0.0.4 MBConvBlock, blocks 2..16 (/root/reference/whenet.py:8):
// Conv2D(in*6, 1x1, no bias) -> BN -> Swish -> DepthwiseConv2D(k, s, 'same') -> BN -> Swish.
//
// Why: round 2's kernel ran the k*k taps as f32 FMAs out of an LDS tile (lane = 4 channels x 7 pixels): 700 FMAs + 220
// f16->f32 converts per lane for a 5x5 layer, i.e. ~60 % of its VALU instructions, with the matrix pipe 96 % idle
// (profiles/r02/pmc_f16_b64_by_kernel.txt).  The VALU is needed for the two Swish activations (2 quarter-rate
// transcendentals each); everything else now goes to the MFMA pipe:
//
//   * depthwise taps = per-channel Toeplitz products on v_mfma_f32_4x4x4_16B_f16 (16 independent 4x4x4 blocks per
//     instruction, 8 cycles; block = channel).  For a group of 4 output pixels of one row,
//         out[i] = sum_kx w[ky][kx] * E[row*S + ky][S*i + kx]      (i = 0..3)
//     is  D[i][j] = sum_k A[i][k] B[k][j]  with  A[i][k] = w[ky][4*chunk + k - S*i]  (a 4x4 slice of the Toeplitz
//     matrix, zero outside the kernel; packed on the host, snapshot.cpp::pack_dw_toeplitz) and B[k][j] = 4
//     consecutive input pixels of column j -- ONE 8-byte LDS read per lane, because
//   * the expanded tile E lives in LDS CHANNEL-major, [channel][row][x] in f16, x contiguous.  To get there without a
//     transpose the expand GEMM runs with pixels as MFMA rows and channels as columns (v_mfma_f32_32x32x16_f16 with
//     the operand roles of pw.hip swapped; the packed weight image is the same): a lane then holds one channel and
//     four runs of 4 consecutive pixels, each an 8-byte LDS write.  A "pixel group" is 4 x-consecutive pixels
//     aligned in TILE coordinates; a 32-row MFMA strip is 8 groups.  'SAME' zeros of the expanded tensor are the
//     zero-filled tile plus a fix-up pass over the <= 5 out-of-image pixels of the border groups;
//   * a column = one x-group x 7 output rows (7 divides 56/28/14/7).  A lane sweeps its column's input rows once:
//     every 8-byte read feeds K MFMAs (one per ky) into 7 rolling accumulators -- no VALU instruction in the taps;
//   * outputs (lane = channel x 4 pixels) go through a 2 KB per-wave LDS stage and leave as 16-byte NHWC pieces.
// The tile is full-width where the LDS allows (no halo recompute along x), TH = 7/14/28 rows.
// Per-workgroup squeeze-excite partial sums / reduce-conv shares are produced exactly as in front.hip (same
// consumer kernels), in a fixed order that depends on the layer only.
//
// HBM bytes per crop: H^2*Cin (x chunks, L2 hits) + Ho^2*Cexp written once.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

#include <atomic>
#include <cstdlib>
#include <string>
#include <vector>

namespace whenet {

namespace {

constexpr int RL = 7;                 // output rows per tap column

__device__ __forceinline__ int fdiv2(int q, float rinv) { return int((float(q) + 0.5f) * rinv); }     // see front.hip

struct F2Params {
    const half_t* x;
    const half_t* wep;
    const float* be;
    const half_t* wdt;
    const float* bd;
    half_t* out;
    float* rpart;
    const float* w1t;
    const float* in_gate;             // [n][Cin] f32 or NULL: per-crop scale of the INPUT channels (Front2Args::in_gate)
    int H, Ho, Cin, Cexp, pad, NTe;
    int CC, TH, TXG, tiles_x, EH, EWp, RP, CP;
    int off_stage, off_red, off_sum;
    int R, RPse;
    int ntiles, chunks, n, xcd;       // launch geometry (1-D grid of ntiles * chunks * n workgroups, xcd_unit())
};

// BN + Swish of one expand task (32 pixels x 32 channels; this lane: channel ch, four runs of 4 consecutive pixels)
// into the tile.  EDGE: the strip hangs over the in-image rows / groups -- those runs are skipped.
template <bool EDGE>
__device__ __forceinline__ void expand_store(const float16v& acc, float bias, unsigned char* ep, const int (&eoff)[4],
                                             const int (&drd)[4], const int (&dcd)[4], int nr_left, int ng_left) {
    (void)bias;                                    // (round 4: the BN bias is the accumulators' initial value -- one v_pk_add per two
                                                   //  values less in a VALU-bound epilogue)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        if (!EDGE || (drd[qq] < nr_left && dcd[qq] < ng_left)) {
            const float2v y0 = swish2(float2v{acc[4 * qq], acc[4 * qq + 1]});
            const float2v y1 = swish2(float2v{acc[4 * qq + 2], acc[4 * qq + 3]});
            half4 o;
            o[0] = half_t(y0[0]);
            o[1] = half_t(y0[1]);
            o[2] = half_t(y1[0]);
            o[3] = half_t(y1[1]);
            *reinterpret_cast<half4*>(ep + eoff[qq]) = o;
        }
    }
}

// XS: the tile's x origin is moved XS pixels to the left (7-wide 5x5 layers: XS = 2 puts image column 0 on a group
// boundary -- 8 instead of 12 pixel slots per row in the expand -- at the price of one more Toeplitz chunk).
// GATED: the expand contracts (in_gate[crop] * x) -- block 2 fed by block 1's depthwise output (option fold12).
template <int K, int S, int KS, int NTHR, int XS, bool GATED>
__global__ __launch_bounds__(NTHR) void whenet_front2_kernel(const F2Params p) {
    constexpr int NCH = (3 * S + K + XS + 3) / 4;     // 4-pixel input chunks a 4-pixel output group reads
    constexpr int NER = (RL - 1) * S + K;             // input rows of a column
    constexpr int NWAVE = NTHR / 64;
    constexpr int PF = KS <= 7 ? KS : 6;              // k-steps of activation operands that travel together (one round
                                                      // trip per task up to Cin = 112, two for Cin = 192)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: the task / item loops are uniform)
    const int g = lane >> 5, lm = lane & 31;
    int b, tile, chunk;
    xcd_front(int(blockIdx.x), p.ntiles, p.chunks, p.n, p.xcd != 0, b, tile, chunk);      // (device_math.h)
    const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const int c0 = chunk * p.CC;
    const int ccur = (p.Cexp - c0 < p.CC) ? (p.Cexp - c0) : p.CC;
    const int H = p.H, Cin = p.Cin, RP = p.RP, CP = p.CP;
    const int oy0 = tyi * p.TH, ox0 = txi * p.TXG * 4;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad - XS;

    // ---- the in-image part of the tile, in tile coordinates (er, ex) = (iy - iy0, ix - ix0) -----------------
    const int er_lo = iy0 < 0 ? -iy0 : 0, er_hi = (iy0 + p.EH < H) ? p.EH : H - iy0;
    const int ex_lo = ix0 < 0 ? -ix0 : 0, ex_hi = (ix0 + p.EWp < H) ? p.EWp : H - ix0;
    const int g_lo = ex_lo >> 2, g_hi = (ex_hi + 3) >> 2;
    const int NG = g_hi - g_lo, NR = er_hi - er_lo;
    // A strip (the 32 rows of one MFMA tile) = SR rows x SC groups of 4 pixels, SR * SC = 8: every lane's pixels sit
    // at FIXED offsets from the strip's corner, so a task costs the lanes one add (no division, no wrap logic).
    const int SCL = NG <= 2 ? 1 : (NG <= 4 ? 2 : 3), SC = 1 << SCL, SRL = 3 - SCL, SR = 1 << SRL;
    const int nsc = (NG + SC - 1) >> SCL, nsr = (NR + SR - 1) >> SRL;
    const int nstrip = nsc * nsr, ntile = (ccur + 31) >> 5, ntask = nstrip * ntile;

    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x + size_t(b) * H * H * Cin);
    const half8* wf0 = reinterpret_cast<const half8*>(p.wep) + size_t(c0 >> 5) * 64 + lane;

    // lane tables: operand side (MFMA row = lane & 31 -> pixel), accumulator side (run qq -> group 2 qq + g)
    const int ua = lm >> 2;
    const int pixoff_a = (ua >> SCL) * H + 4 * (ua & (SC - 1)) + (lm & 3);
    int eoff[4], drd[4], dcd[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int u = 2 * qq + g;
        drd[qq] = u >> SCL;
        dcd[qq] = u & (SC - 1);
        eoff[qq] = drd[qq] * RP + dcd[qq] * 8;
    }
    const int npix1 = H * H - 1;
    const unsigned cin2 = unsigned(Cin) * 2u;
    auto a_offset = [&](int rb, int cbk) -> unsigned {          // byte offset of this lane's operand row in the crop
        const int pb = (iy0 + er_lo + (rb << SRL)) * H + ix0 + 4 * (g_lo + (cbk << SCL));
        int pl = pb + pixoff_a;
        pl = pl < 0 ? 0 : (pl > npix1 ? npix1 : pl);           // rows / pixels outside the image: any valid address
        return __umul24(unsigned(pl), cin2) + unsigned(g) * 16u;       // (they are never stored, or fixed up below)
    };

    // ---- expand: tasks (channel tile, strip), a contiguous range per wave --------------------------------------
    const int t_begin = (wave * ntask) / NWAVE, t_end = ((wave + 1) * ntask) / NWAVE;
    // activation operands of D tasks are in flight (a ring of D register sets).  Measured: D = 4 / 2 for the shallow
    // layers (Cin = 16 / 24..40) changes nothing (b2: 59.1 vs 56.1 us at 64 crops) -- the ~0.6 us a task takes there is
    // not operand latency but the wave's own issue rate (one VALU instruction per ~5 cycles at 2 - 3 waves per SIMD)
    // plus the exposed FIRST load of the workgroup; so one task ahead it is.  (Re-measured at the end of round 4, after the stall
    // behind the prefetch was gone -- tools/build_variant.sh ring2 front2.hip -DWHENET_F2_RING=2: default line 154.9 / 152.5 / 148.6 k
    // crops/s for 1 / 2 / 3 tasks ahead, the deeper rings cost registers on the 14 x 14 layers.)
#ifndef WHENET_F2_RING
#define WHENET_F2_RING 1
#endif
    constexpr int D = WHENET_F2_RING;
    half8 w[KS], aq[D][PF];
    float bias_cur = 0.f;
    float16v bias16;                                           // splat of bias_cur, refreshed by load_w (see the task loop)
    // The folded block-1 project (engine.cpp, option fold12): the expand's input is the PREVIOUS block's gated
    // depthwise output; the gate scales the contraction index, so it is applied to this crop's copy of the weights
    // instead of to every pixel row -- in f32 (f32 composed weights x f32 gate, ONE rounding to f16 per weight; the
    // two-step form rounds the gate, the gated activation, both weight matrices and block 1's output).
    // (a template parameter: the gate rows cost KS x 8 registers, only the folded block 2 pays them)
    float4v gq[GATED ? KS : 1][2];
    if constexpr (GATED) {
        const float* gp = reinterpret_cast<const float*>(p.in_gate) + size_t(b) * Cin + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            gq[ks][0] = *reinterpret_cast<const float4v*>(gp + ks * 16);
            gq[ks][1] = *reinterpret_cast<const float4v*>(gp + ks * 16 + 4);
        }
    }
    auto load_w = [&](int tl) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (GATED) {
                // wep: the same fragment order as the f16 image, 8 floats per lane
                const float* wq = reinterpret_cast<const float*>(p.wep) + ((size_t(ks) * p.NTe + (c0 >> 5) + tl) * 64 + lane) * 8;
                const float4v lo = *reinterpret_cast<const float4v*>(wq) * gq[ks][0];
                const float4v hi = *reinterpret_cast<const float4v*>(wq + 4) * gq[ks][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w[ks][e] = half_t(lo[e]);
                    w[ks][4 + e] = half_t(hi[e]);
                }
            } else {
                w[ks] = wf0[(size_t(ks) * p.NTe + tl) * 64];
            }
        }
        const int ch = tl * 32 + lm;
        bias_cur = (ch < ccur) ? p.be[c0 + ch] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bias16[r] = bias_cur;
    };
    // k beyond Cin (Cin = 24 / 40: the upper half of the last k-step): the packed weights are zero there, but the 16
    // bytes past a pixel row are the next pixel's channels -- or, for the last pixel of the last crop, whatever follows
    // the tensor: 0 x NaN would poison the accumulators, so those lanes' operand is zeroed (one v_cndmask per dword on
    // the last k-step of the two layers concerned; same bits wherever the bytes were finite)
    // ... applied where the operand is CONSUMED (zero_ktail below, right in front of its MFMA): a select behind the load makes
    // the wave wait for the operands of the NEXT task the moment it has requested them (round 4, from the ISA: s_waitcnt
    // vmcnt(1) + 4 v_cndmask right behind the prefetch of every task of every front2 launch)
    const bool ktail = (Cin & 15) != 0 && g == 1;
    auto load_a = [&](half8 (&a)[PF], unsigned off, int ks0) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (ks0 + u < KS) a[u] = *reinterpret_cast<const half8*>(xb + off + (ks0 + u) * 32);
    };
    auto zero_ktail = [&](half8 (&a)[PF], int ks0) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (ks0 + u == KS - 1 && ktail) a[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    STAMP(0);
    int tl = 0, rb = 0, cbk = 0;                               // the task being computed
    int ti = t_begin, rbi = 0, cbki = 0;                       // the next task whose operands are requested
    unsigned aoffq[D];
    auto issue = [&](int slot) {                               // strip (rbi, cbki) -> ring slot, then step to the next strip
        if (ti < t_end) {
            aoffq[slot] = a_offset(rbi, cbki);
            load_a(aq[slot], aoffq[slot], 0);
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
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
    }
    {   // rows of the tile outside the image are 'SAME' zeros of the EXPANDED tensor (top / bottom tiles only)
        const int nz = er_lo + (p.EH - er_hi);
        if (nz > 0) {                                           // (uniform)
            const float r_nz = __builtin_amdgcn_rcpf(float(nz));
            for (int pr = tid; pr < ccur * nz; pr += NTHR) {
                const int c = fdiv2(pr, r_nz), hr = pr - c * nz;
                const int er = hr < er_lo ? hr : er_hi + (hr - er_lo);
                unsigned char* rowp = E + c * CP + ((c >> 3) & 1) * 8 + er * RP;
                for (int q = 0; q < RP; q += 8) *reinterpret_cast<half4*>(rowp + q) = half4{0, 0, 0, 0};
            }
        }
    }

    STAMP(1);
    for (int t0 = t_begin; t0 < t_end; t0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int t = t0 + d;
        if (t >= t_end) break;                                 // (uniform)
        // (this lane's channel: BN bias as the accumulators' initial value.  Round 5: the 16-register splat of the bias is kept in
        //  its OWN registers and handed to the first MFMA as the C operand with another destination -- written as `acc[r] =
        //  bias` in front of every task the compiler spent 23 v_mov per task on it, a fifth of the task's VALU instructions)
        float16v acc;
        zero_ktail(aq[d], 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[d][0], w[0], bias16, 0, 0, 0);
#pragma unroll
        for (int u = 1; u < PF; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[d][u], w[u], acc, 0, 0, 0);
#pragma unroll
        for (int ks = PF; ks < KS; ks += PF) {
            load_a(aq[d], aoffq[d], ks);
            zero_ktail(aq[d], ks);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (ks + u < KS) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[d][u], w[ks + u], acc, 0, 0, 0);
        }
        // this task's place in the tile; its ring slot is free: the operands of task t + D take off
        const int ch = tl * 32 + lm;
        unsigned char* ep = E + ch * CP + ((ch >> 3) & 1) * 8 + (er_lo + (rb << SRL)) * RP + (g_lo + (cbk << SCL)) * 8;
        unsigned char* ep0 = ep;                               // (row of the strip's corner, at its first group)
        const int nr_left = NR - (rb << SRL), ng_left = NG - (cbk << SCL), cbk_this = cbk;
        const float bias = bias_cur;
        int tln = tl;
        if (++cbk == nsc) {
            cbk = 0;
            if (++rb == nsr) {
                rb = 0;
                ++tln;
            }
        }
        issue(d);
        if (t + 1 < t_end && tln != tl) load_w(tln);           // (rare: at most once or twice per wave)
        tl = tln;
        if (ch < ccur) {
            if (nr_left >= SR && ng_left >= SC) expand_store<false>(acc, bias, ep, eoff, drd, dcd, nr_left, ng_left);
            else expand_store<true>(acc, bias, ep, eoff, drd, dcd, nr_left, ng_left);
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
                        half_t* rowp = reinterpret_cast<half_t*>(ep0 + dr * RP) - 4 * g_lo;      // pixel 0 of the tile row
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < ex_lo) rowp[q] = half_t(0);
                    }
                }
            }
            if (fix_r && cbk_this == nsc - 1) {                // (uniform) strip holds the right border group and beyond
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int dr = g + 2 * dd;
                    if (dr < SR && dr < nr_left) {
                        half_t* rowp = reinterpret_cast<half_t*>(ep0 + dr * RP) - 4 * (g_lo + cbk_this * SC) + ex_hi;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (ex_hi + q < p.EWp) rowp[q] = half_t(0);
                    }
                }
            }
        }
      }
    }

    STAMP(2);
    // ---- depthwise taps: items (16-channel block, quad of columns), a contiguous range per wave ---------------
    const int ncb = (ccur + 15) >> 4;
    const int ncol = (p.TH / RL) * p.TXG, ncq = (ncol + 3) >> 2, nitem = ncb * ncq;
    const float r_txg = __builtin_amdgcn_rcpf(float(p.TXG));
    const int i_begin = (wave * nitem) / NWAVE, i_end = ((wave + 1) * nitem) / NWAVE;
    const int cl = lane >> 2, j = lane & 3;
    half4 A[K][NCH];
    float bdv = 0.f;
    auto load_taps = [&](int cb) {
        const half4* src = reinterpret_cast<const half4*>(p.wdt) + (size_t((c0 >> 4) + cb) * K * NCH) * 64 + lane;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) A[ky][ch] = src[(ky * NCH + ch) * 64];
        bdv = p.bd[c0 + cb * 16 + cl];
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
        // (only the waves that hold an output jo < RPse need them -- block 2: R = 4, i.e. 16 lanes of wave 0; the other seven
        //  waves used to issue the same 16 strided loads and ~70 address instructions for nothing: round 5)
        if (p.w1t != nullptr && wave * 16 < p.RPse) {          // (uniform; clamped addresses: no lane predicates)
            const float* wrow = p.w1t + size_t(jo < p.R ? jo : p.R - 1) * p.Cexp + c0;
#pragma unroll
            for (int i = 0; i < W1V; ++i) {
                const int c = q + 4 * i;
                w1v[i] = wrow[c < ccur ? c : ccur - 1];
            }
        }
    }
    unsigned char* stg = smem + p.off_stage + wave * 2048;
    float* s_red = reinterpret_cast<float*>(smem + p.off_red);          // [ncq][CC]
    float* s_sum = reinterpret_cast<float*>(smem + p.off_sum);          // [CC]
    // piece coordinates of this lane in the output stage: slot = (row * 4 + i) * 4 + j, two 16-byte halves per slot
    const int jP = (lane >> 1) & 3, iP = (lane >> 3) & 3, hP = lane & 1, rP = lane >> 5;
    unsigned char* outb = reinterpret_cast<unsigned char*>(p.out + size_t(b) * p.Ho * p.Ho * p.Cexp);
    const unsigned row_bytes = unsigned(p.Ho) * unsigned(p.Cexp) * 2u;

    for (int it = i_begin; it < i_end; ++it) {
        const int col = cq * 4 + j;                            // (this block's taps are in A: loaded in the prologue
        const int colc = col < ncol ? col : ncol - 1;          //  or behind the previous sweep)
        const int seg = fdiv2(colc, r_txg), xgl = colc - seg * p.TXG;
        const int c = cb * 16 + cl;
        const unsigned char* bp = E + c * CP + ((c >> 3) & 1) * 8 + (seg * RL * S) * RP + xgl * (8 * S);
        float4v acc[RL];
        const float4v bd4 = float4v{bdv, bdv, bdv, bdv};       // (BN bias as the initial value: the C operand of a row's FIRST product --
        bool started[RL];                                      //  everything here unrolls, the flags fold at compile time)
#pragma unroll
        for (int r = 0; r < RL; ++r) started[r] = false;
#pragma unroll
        for (int er = 0; er < NER; ++er) {
            half4 bv[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) bv[ch] = *reinterpret_cast<const half4*>(bp + er * RP + ch * 8);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int d = er - ky;
                if (d >= 0 && d % S == 0 && d / S < RL) {
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        acc[d / S] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[ky][ch], bv[ch], started[d / S] ? acc[d / S] : bd4, 0, 0, 0);
                        started[d / S] = true;
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
        // ---- BN + Swish, channel sums, and the way out: lane (channel cl, column j) holds 7 rows x 4 pixels -----
        const bool okc = col < ncol;
        const int oxb = ox0 + 4 * xgl;
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (okc && oxb + i < p.Ho) ? 1.f : 0.f;
        const int colP = cq_this * 4 + jP;
        const int colPc = colP < ncol ? colP : ncol - 1;
        const int segP = fdiv2(colPc, r_txg), xglP = colPc - segP * p.TXG;
        const int oxP = ox0 + 4 * xglP + iP;
        const bool okP = colP < ncol && oxP < p.Ho;
        const unsigned obase = (__umul24(unsigned(oy0 + segP * RL + rP), unsigned(p.Ho)) + unsigned(oxP)) * unsigned(p.Cexp) * 2u +
                               unsigned(c0 + cb_this * 16 + hP * 8) * 2u;
        float2v sum2 = {0.f, 0.f};
        const float2v m01 = {m[0], m[1]}, m23 = {m[2], m[3]};
        unsigned char* sw = stg + j * 32 + cl * 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                 // rows 0..3, then rows 4..6, through the 2 KB stage
            const int r0 = half * 4, nr = half ? 3 : 4;
#pragma unroll
            for (int r = 0; r < nr; ++r) {
                const float2v y01 = swish2(float2v{acc[r0 + r][0], acc[r0 + r][1]});
                const float2v y23 = swish2(float2v{acc[r0 + r][2], acc[r0 + r][3]});
                sum2 = y01 * m01 + sum2;
                sum2 = y23 * m23 + sum2;
                *reinterpret_cast<half_t*>(sw + (r * 4 + 0) * 128) = half_t(y01[0]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 1) * 128) = half_t(y01[1]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 2) * 128) = half_t(y23[0]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 3) * 128) = half_t(y23[1]);
            }
            wave_lds_sync();
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int rl = 2 * pp + rP;                    // stage row of this lane's piece: 2 pp + (lane >> 5)
                const half8 v = *reinterpret_cast<const half8*>(stg + (pp * 64 + lane) * 16);
                if (okP && rl < nr) *reinterpret_cast<half8*>(outb + obase + unsigned(r0 + 2 * pp) * row_bytes) = v;
            }
            wave_lds_sync();
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
        if (p.w1t == nullptr) p.rpart[(size_t(b) * p.ntiles + tile) * p.Cexp + c0 + tid] = t;
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
                p.rpart[((size_t(b) * p.ntiles + tile) * p.chunks + chunk) * p.RPse + jo] = (jo < p.R) ? tot : 0.0f;
        }
    }
    STAMP(6);
}

}  // namespace

namespace {

// hipFuncSetAttribute once per (instantiation, device); handles are one per host thread, so the flag is atomic
struct OncePerDevice {
    std::atomic<bool> done[64];
    OncePerDevice() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
};

template <int K, int S, int KS, int NTHR, int XS = 0, bool GATED = false>
void launch_f2(const Front2Args& a, hipStream_t stream) {
    const Front2Plan& pl = a.plan;
    F2Params p{};
    p.x = static_cast<const half_t*>(a.x);
    p.wep = static_cast<const half_t*>(a.wep);
    p.be = a.be;
    p.wdt = static_cast<const half_t*>(a.wdt);
    p.bd = a.bd;
    p.out = static_cast<half_t*>(a.out);
    p.rpart = a.rpart;
    p.w1t = a.w1t;
    p.in_gate = a.in_gate;
    p.H = a.H;  p.Ho = a.Ho;  p.Cin = a.Cin;  p.Cexp = a.Cexp;  p.pad = a.pad;  p.NTe = a.NTe;
    p.CC = pl.CC;  p.TH = pl.TH;  p.TXG = pl.TXG;  p.tiles_x = pl.tiles_x;
    p.EH = pl.EH;  p.EWp = pl.EWp;  p.RP = pl.RP;  p.CP = pl.CP;
    p.off_stage = pl.off_stage;  p.off_red = pl.off_red;  p.off_sum = pl.off_sum;
    p.R = a.R;  p.RPse = (a.R + 3) & ~3;
    p.ntiles = pl.tiles_x * pl.tiles_y;  p.chunks = pl.chunks;  p.n = a.n;  p.xcd = a.xcd_grouped ? 1 : 0;
    WHENET_REQUIRE(pl.lds_bytes <= 160 * 1024, WHENET_EINVAL, "front2: the tile plan needs more than 160 KB of LDS");
    static OncePerDevice attr;
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr.done[dev].load(std::memory_order_acquire)) {
        // (two threads may both get here: the call is idempotent, the flag only saves repeating it)
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_front2_kernel<K, S, KS, NTHR, XS, GATED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((whenet_front2_kernel<K, S, KS, NTHR, XS, GATED>), dim3(unsigned(pl.tiles_x * pl.tiles_y) * unsigned(pl.chunks) * unsigned(a.n)), dim3(NTHR),
                       pl.lds_bytes, stream, p);
    WHENET_HIP_CHECK(hipGetLastError());
}

template <int NTHR>
void launch_f2_shape(const Front2Args& a, hipStream_t stream) {
    const int key = a.k * 1000 + a.s * 100 + a.KSe + a.plan.xs * 10000;
    switch (key) {                                   // EfficientNet-B0's eleven (kernel, stride, Cin / 16) shapes
        case 3201: launch_f2<3, 2, 1, NTHR>(a, stream); break;       // b2
        case 3202:                                                   // Cin 17..32: b2 fed by block 1's depthwise output (fold12)
            if (a.in_gate != nullptr) launch_f2<3, 2, 2, NTHR, 0, true>(a, stream);
            else launch_f2<3, 2, 2, NTHR>(a, stream);
            break;
        case 3102: launch_f2<3, 1, 2, NTHR>(a, stream); break;       // b3
        case 5202: launch_f2<5, 2, 2, NTHR>(a, stream); break;       // b4
        case 5103: launch_f2<5, 1, 3, NTHR>(a, stream); break;       // b5
        case 3203: launch_f2<3, 2, 3, NTHR>(a, stream); break;       // b6
        case 3105: launch_f2<3, 1, 5, NTHR>(a, stream); break;       // b7, b8
        case 5105: launch_f2<5, 1, 5, NTHR>(a, stream); break;       // b9
        case 5107: launch_f2<5, 1, 7, NTHR>(a, stream); break;       // b10, b11
        case 5207: launch_f2<5, 2, 7, NTHR>(a, stream); break;       // b12
        case 5112: launch_f2<5, 1, 12, NTHR>(a, stream); break;      // b13 - b15
        case 25112: launch_f2<5, 1, 12, NTHR, 2>(a, stream); break;  // b13 - b15, tile origin shifted by 2 pixels
        case 3112: launch_f2<3, 1, 12, NTHR>(a, stream); break;      // b16
        default: throw Error(WHENET_EINVAL, "front2: unsupported (kernel, stride, Cin) shape");
    }
}

}  // namespace

// Toeplitz operand image of a depthwise kernel for v_mfma_f32_4x4x4_16B_f16 (see the header comment):
//   [C / 16 blocks][ky][chunk][lane 0..63][4 halfs];  lane l <-> channel 16 * block + (l >> 2), output pixel i = l & 3,
//   element k <-> tap kx = 4 * chunk + k - s * i - xs (zero outside 0..k-1); xs = Front2Plan::xs.
std::vector<half_t> pack_dw_toeplitz(const std::vector<float>& w, int k, int s, int C, int xs) {
    WHENET_REQUIRE(C % 16 == 0 && int(w.size()) == k * k * C, WHENET_EINVAL, "pack_dw_toeplitz: bad shape");
    const int nch = (3 * s + k + xs + 3) / 4;
    std::vector<half_t> out(size_t(C / 16) * k * nch * 64 * 4, half_t(0));
    for (int blk = 0; blk < C / 16; ++blk)
        for (int ky = 0; ky < k; ++ky)
            for (int ch = 0; ch < nch; ++ch)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int c = blk * 16 + (l >> 2), i = l & 3, kx = 4 * ch + e - s * i - xs;
                        if (kx >= 0 && kx < k)
                            out[(((size_t(blk) * k + ky) * nch + ch) * 64 + l) * 4 + e] = half_t(w[size_t(ky * k + kx) * C + c]);
                    }
    return out;
}

Front2Plan make_front2_plan(int k, int s, int Ho, int Cexp, int CC, int TH, int TXG, int threads, int xs) {
    const int OXG = ceil_div(Ho, 4), nch = (3 * s + k + xs + 3) / 4;
    WHENET_REQUIRE(xs == 0 || (xs == 2 && k == 5 && s == 1), WHENET_EINVAL, "front2: the origin shift exists for 5x5 stride-1 layers");
    // (x tiles may be ragged: the last tile of a row then holds fewer output groups; outputs, channel sums and the expand
    //  strips beyond the image are masked / clipped by the kernel)
    WHENET_REQUIRE(TH % RL == 0 && Ho % TH == 0 && TXG >= 1 && TXG <= OXG && (CC % 32 == 0 || CC == Cexp) && Cexp % 16 == 0 &&
                       (threads == 256 || threads == 512),
                   WHENET_EINVAL, "front2: bad tile plan");
    Front2Plan p;
    p.threads = threads;
    p.CC = CC < Cexp ? CC : Cexp;
    p.TH = TH;
    p.TXG = TXG;
    p.xs = xs;
    p.tiles_x = ceil_div(OXG, TXG);
    p.tiles_y = Ho / TH;
    p.chunks = ceil_div(Cexp, p.CC);
    p.EH = (TH - 1) * s + k;
    p.EWp = 4 * (s * (TXG - 1) + nch);
    p.RP = p.EWp * 2;
    int cp = p.EH * p.RP + 8;                           // + 8: channels 8..15 of a block start one group later
    cp = (cp + 63) / 64 * 64;                           // pitch = 32 mod 64 bytes: the 8 channels of a tap read
    p.CP = (cp - 32 >= p.EH * p.RP + 8) ? cp - 32 : cp + 32;    // (half wave) cover all 64 banks
    const int ncq = ceil_div((TH / RL) * TXG, 4);
    const size_t e_bytes = size_t(p.CC) * p.CP;
    p.off_stage = int((e_bytes + 15) & ~size_t(15));
    p.off_red = p.off_stage + (threads / 64) * 2048;
    p.off_sum = p.off_red + ncq * p.CC * 4;
    p.lds_bytes = size_t(p.off_sum) + size_t(p.CC) * 4;
    return p;
}

namespace {
struct Tuned2 { int k, s, H, Cexp, CC, TH, TXG, xs, use, threads; };
const Tuned2 TUNED2[] = {
#include "front2_tuned.inc"
};
}  // namespace

Front2Plan plan_front2(int k, int s, int H, int Ho, int Cexp) {
    static const bool no_tuned = getenv("WHENET_FRONT_NO_TUNED") != nullptr;       // (probes only; read once)
    if (!no_tuned)
        for (const Tuned2& t : TUNED2)
            if (t.k == k && t.s == s && t.H == H && t.Cexp == Cexp) return make_front2_plan(k, s, Ho, Cexp, t.CC, t.TH, t.TXG, t.threads, t.xs);
    // shapes outside the table: 32 channels, 7 rows, the widest tile that leaves two workgroups per CU
    const int OXG = ceil_div(Ho, 4);
    for (int txg = OXG; txg >= 1; --txg)
        if (OXG % txg == 0) {
            const Front2Plan p = make_front2_plan(k, s, Ho, Cexp, 32, RL, txg, 256, 0);
            if (p.lds_bytes <= 80 * 1024 || txg == 1) return p;
        }
    throw Error(WHENET_EINVAL, "front2: no tile plan fits");
}

// Layers on which this kernel beats round 2's whenet_front_kernel (measured, tools/probes/front2_probe.hip)
bool front2_preferred(int k, int s, int H, int Cexp) {
    for (const Tuned2& t : TUNED2)
        if (t.k == k && t.s == s && t.H == H && t.Cexp == Cexp) return t.use != 0;
    return false;
}

std::vector<Front2Plan> plan_front2_candidates(int k, int s, int Ho, int Cexp) {
    std::vector<Front2Plan> out;
    const int OXG = ceil_div(Ho, 4);
    for (int CC : {32, 64, 96, 128})
        for (int TH : {7, 14, 28})
            for (int TXG = 1; TXG <= OXG; ++TXG) {
                if (Ho % TH || (CC > Cexp && CC != 32)) continue;       // (ragged x tiles are candidates too)
                if (CC < Cexp && Cexp % CC && (Cexp % CC) % 16) continue;
                if (TXG < 2 && OXG > 1) continue;                        // (x tiles of one group: all halo)
                for (int xs : {0, 2}) {
                    if (xs && !(k == 5 && s == 1 && Ho == 7)) continue;
                    const Front2Plan p = make_front2_plan(k, s, Ho, Cexp, CC, TH, TXG, 256, xs);
                    if (p.lds_bytes <= 150 * 1024) out.push_back(p);
                }
            }
    return out;
}

// Lanes per workgroup for a launch of n crops: the plan's (front2_tuned.inc).  Measured (tools/probes/front2_probe.hip,
// WHENET_FRONT_THREADS): 8-wave workgroups win only on block 2, whose 58 KB tile admits two workgroups per CU (216 vs
// 228 us at 256 crops, 57.5 vs 59.2 at 64); everywhere else they are equal or slower (b4: 178 vs 118 us at 256 crops).
// Same bits either way: a wave's share of tasks / items changes, not what a task or an item computes.
int front2_threads(const Front2Plan& p, int n) {
    static const int forced = [] { const char* e = getenv("WHENET_FRONT_THREADS"); return e ? atoi(e) : 0; }();   // probes only
    (void)n;
    return forced ? forced : p.threads;
}

void launch_front2(const Front2Args& a, hipStream_t stream) {
    WHENET_REQUIRE(a.KSe == ceil_div(a.Cin, 16), WHENET_EINVAL, "front2: k-steps do not match Cin");
    WHENET_REQUIRE(a.in_gate == nullptr || (a.k == 3 && a.s == 2 && a.KSe == 2 && a.Cin == 32), WHENET_EINVAL,
                   "front2: the gated-input form exists for the 3x3 / stride-2 / Cin = 32 shape only");
    Front2Args b = a;
    if (b.plan.threads != 256) {              // the stage / sums offsets depend on the wave count
        b.plan = make_front2_plan(a.k, a.s, a.Ho, a.Cexp, a.plan.CC, a.plan.TH, a.plan.TXG, a.plan.threads, a.plan.xs);
    }
    if (b.plan.threads == 512) launch_f2_shape<512>(b, stream);
    else launch_f2_shape<256>(b, stream);
}

std::string kernel_name_front2(int k, int s, int kse, int threads, int xs, bool gated) {
    return "whenet_front2_kernel<" + std::to_string(k) + ", " + std::to_string(s) + ", " + std::to_string(kse) + ", " +
           std::to_string(threads) + ", " + std::to_string(xs) + ", " + (gated ? "true" : "false") + ">";
}

}  // namespace whenet
