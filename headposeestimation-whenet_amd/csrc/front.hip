// MBConv "front": expand 1x1 conv (MFMA) + BN + Swish  ->  depthwise kxk conv + BN + Swish, ONE
// kernel; the 6x-expanded tensor lives only in LDS.
//
// Reference: efficientnet 0.0.4 MBConvBlock, blocks 2..16 (/root/reference/whenet.py:8; SURVEY.md
// Appendix B): Conv2D(in*6, 1x1, no bias) -> BN -> Swish -> DepthwiseConv2D(k, s, 'same') -> BN ->
// Swish.  As two launches the expanded tensor is written to and re-read from HBM: 2 x 7.8 MB of
// the 27.7 MB a crop moves (f16), plus a kernel boundary per block.
//
// Mapping (gfx950): workgroup = crop x chunk of CC expanded channels x output tile of TH rows x
// 7*NSX columns (same tiling as dw.hip).  It
//   1. zeroes its LDS tile E [(TH-1)s+k][(7*NSX-1)s+k] pixels x CC channels -- the zero halo is
//      TF 'SAME' padding of the EXPANDED tensor (padding is zero after expand+Swish, not
//      expand(0));
//   2. computes the expand conv for the in-image pixels of that input tile on the matrix cores
//      (transposed product, packed weight fragments, exactly as pw.hip; the block's input rows are
//      read straight from global/L2 as 16-byte MFMA fragments) and writes BN+Swish'ed values into E
//      (interior halo pixels are recomputed by the neighbouring tiles: (EH*EW)/(TH*s*7*NSX*s)
//      extra MFMA work, free on this HBM-bound path);
//   3. runs the depthwise taps out of E (lane = 4 channels x strip of 7 output pixels), BN+Swish,
//      stores NHWC and the tile's channel sums for the squeeze-excite mean (fixed order).
// HBM bytes per crop: H^2*Cin (x chunks, L2 hits) + Ho^2*Cexp written once: 13.9 MB -> the
// "2-kernel MBConv" traffic of BASELINE.md.  Arithmetic and summation order of both convs are those
// of pw.hip / dw.hip (the expanded activation is rounded to T in LDS exactly as it was in HBM); only
// the grouping of the squeeze-excite partial sums follows this kernel's own tiling.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

#include <atomic>
#include <cstdlib>
#include <string>
#include <vector>

namespace whenet {

namespace {

constexpr int P = 7;
constexpr int VC = 4;

// q / d for 0 <= q < 2^20 and 0 < d < 512, exact, with rinv = 1 / float(d) (1 ulp): (q + 0.5) / d is at least
// 0.5 / d away from an integer, far more than the rounding of the three float operations.  An integer
// division by a run-time divisor is ~25 instructions on this ISA, four of them quarter-rate 32-bit multiplies;
// this kernel used to spend a third of its VALU cycles in them.
__device__ __forceinline__ int fdiv(int q, float rinv) { return int((float(q) + 0.5f) * rinv); }

template <typename T, int K, int S, int NTHR, bool SP = false, bool GATED = false>
__global__ __launch_bounds__(NTHR, 4) void whenet_front_kernel(const T* __restrict__ x, const T* __restrict__ wep,
                                                            const float* __restrict__ be,
                                                            const float* __restrict__ wd,
                                                            const float* __restrict__ bd, T* __restrict__ out,
                                                            float* __restrict__ rpart, int H, int Ho, int Cin,
                                                            int Cexp, int pad, int KSe, int NTe, int CC, int TH,
                                                            int NSX, int tiles_x, int EH, int EW, int EP, int w_off,
                                                            const float* __restrict__ w1t, int R, int RP, float wsi,
                                                            const float* __restrict__ in_gate, int ntiles, int chunks, int n, int xcd) {
    // GATED (SP only, round 6): the expand contracts (in_gate[crop] * x) -- block 2 fed by block 1's depthwise output with block 1's
    // project folded into the expand weights (engine.cpp, option fold12); the gate multiplies the float32 operand before it is split.
    // SP (T = float, WHENET_F32S): the expand products as binary16 hi/lo pairs on the f16 matrix cores (device_math.h PwOps);
    // KSe then counts 16-deep k-steps and wep is the [hi | lo] image pair.  Everything behind the expand is unchanged.
    using OPS = PwOps<T, SP>;
    constexpr int V = OPS::V;                   // k elements of a lane's operand fragment
    constexpr int SZ = int(sizeof(T));
    using VT = typename Vec<T>::type;
    using WF = typename OPS::W;
    using AF = typename OPS::A;
    using OT = T __attribute__((ext_vector_type(4)));
    using VCT = OT;
    constexpr int NIX = (P - 1) * S + K;
    constexpr int NWAVE = NTHR / 64;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;                                            // [EH*EW] pixels, pitch EP
    float* s_red = reinterpret_cast<float*>(smem);                      // aliases E after the taps
    float* s_w = reinterpret_cast<float*>(smem + w_off);                // [K*K][CC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    int b, tile, chunk;
    xcd_front(int(blockIdx.x), ntiles, chunks, n, xcd != 0, b, tile, chunk);      // (device_math.h)
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int c0 = chunk * CC;
    const int ccur = (Cexp - c0 < CC) ? (Cexp - c0) : CC;
    const int oy0 = tyi * TH, ox0 = txi * NSX * P;
    const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;

    STAMP(0);
    // ---- expand (MFMA) over the in-image pixels of the input tile -> E ------------------------
    // A task = one 32-pixel strip x one 32-channel tile; a wave walks its tasks with the operands
    // of the NEXT task (first 4 k-steps + bias) already in flight while it applies BN+Swish to the
    // current one: the only exposed global-memory round trip is the first.
    const int iy_lo = iy0 < 0 ? 0 : iy0, ix_lo = ix0 < 0 ? 0 : ix0;
    const int iy_hi = (iy0 + EH < H) ? iy0 + EH : H, ix_hi = (ix0 + EW < H) ? ix0 + EW : H;
    const int RW = ix_hi - ix_lo, npx = (iy_hi - iy_lo) * RW;
    const int nstrip = (npx + 31) >> 5, ntile = (ccur + 31) >> 5;
    const int ntask = nstrip * ntile;
    const float r_nstrip = __builtin_amdgcn_rcpf(float(nstrip)), r_rw = __builtin_amdgcn_rcpf(float(RW));

    const T* xb = x + size_t(b) * H * H * Cin;
    const size_t wf0 = size_t(c0 >> 5) * 64, w_lo = size_t(KSe) * NTe * 64;      // 16-byte fragments
    struct Task {
        bool valid;
        int tl, eoff;
        const T* xrow;
        size_t wf;
    };
    auto make_task = [&](int t) -> Task {
        Task k;
        k.tl = fdiv(t, r_nstrip);
        const int strip = t - k.tl * nstrip;
        const int q = strip * 32 + lm;
        k.valid = t < ntask && q < npx;
        const int ry = k.valid ? fdiv(q, r_rw) : 0, rx = k.valid ? q - ry * RW : 0;
        const int iy = iy_lo + ry, ix = ix_lo + rx;
        k.xrow = xb + ((iy * H + ix) * Cin + g * V);            // (32-bit offsets inside the crop)
        k.eoff = ((iy - iy0) * EW + (ix - ix0)) * EP;
        k.wf = wf0 + (k.tl * 64 + lane);
        return k;
    };
    // PF k-steps of operands travel together.  Measured alternatives (B=64 and B=512, round 2): PF = 8 spills at
    // the 128-register budget of 4 workgroups per CU (-25 %); PF = 8 with 168 registers, i.e. 3 workgroups per CU,
    // is also 25 % slower although it saves a round trip per task; 5 waves per SIMD (96 registers) is neutral.
    // Keeping the LDS tile E in f32 (no v_cvt_f32_f16 in the taps, -16 % VALU instructions) is 3-40 % SLOWER:
    // the taps then read twice the LDS bytes.  The kernel sits on VALU issue with the LDS pipe half busy.
    constexpr int PF = SP ? 2 : 4;              // (32 k of operands either way)
    auto load_ops = [&](const Task& k, int ks, WF (&w)[PF], AF (&a)[PF]) {
#pragma unroll
        for (int u = 0; u < PF; ++u) w[u] = (ks + u < KSe) ? OPS::load_w(wep, k.wf + size_t(ks + u) * NTe * 64, w_lo) : OPS::zero_w();
#pragma unroll
        for (int u = 0; u < PF; ++u)
            a[u] = (k.valid && ks + u < KSe && (ks + u) * 2 * V + g * V < Cin) ? OPS::load_a(k.xrow + (ks + u) * 2 * V) : OPS::zero_a();
    };
    auto load_half = [&](const Task& k, int ks, int u0, WF (&w)[PF], AF (&a)[PF]) {      // k-steps ks, ks + 1 -> slots u0, u0 + 1
#pragma unroll
        for (int u = 0; u < PF / 2; ++u) w[u0 + u] = (ks + u < KSe) ? OPS::load_w(wep, k.wf + size_t(ks + u) * NTe * 64, w_lo) : OPS::zero_w();
#pragma unroll
        for (int u = 0; u < PF / 2; ++u)
            a[u0 + u] = (k.valid && ks + u < KSe && (ks + u) * 2 * V + g * V < Cin) ? OPS::load_a(k.xrow + (ks + u) * 2 * V) : OPS::zero_a();
    };
    auto load_bias = [&](const Task& k, float4v (&bv)[4]) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int nl = k.tl * 32 + 8 * qq + 4 * g;
            bv[qq] = (nl < ccur) ? *reinterpret_cast<const float4v*>(be + c0 + nl) : float4v{0.f, 0.f, 0.f, 0.f};
        }
    };

    // Loads are issued oldest-needed first (the vector-memory counter retires in order): the chunk's
    // depthwise taps, then the first task's operands; the taps are parked in LDS and E is zeroed (the
    // halo outside the image is TF 'SAME' padding of the EXPANDED tensor) while the operands are
    // still in flight -- the barrier below orders LDS traffic only.
    // (lane -> (tap, channel) without a division: 32 channels x NTHR/32 taps per pass)
    constexpr int TPP = NTHR / 32;                              // taps per pass
    constexpr int WT = (K * K + TPP - 1) / TPP, WC = (K == 5) ? 4 : 5;     // passes over taps x 32-channel groups
    float wreg[WT][WC];
#pragma unroll
    for (int jt = 0; jt < WT; ++jt)
#pragma unroll
        for (int jc = 0; jc < WC; ++jc) {
            const int tap = (tid >> 5) + jt * TPP, c = (tid & 31) + 32 * jc;
            wreg[jt][jc] = (tap < K * K && c < ccur) ? wd[size_t(tap) * Cexp + c0 + c] : 0.f;
        }
    Task cur = make_task(wave);
    WF w[PF];
    AF a[PF];
    float4v bv[4];
    if (wave < ntask) {
        load_bias(cur, bv);
        load_ops(cur, 0, w, a);
    }
    for (int i = tid; i < EH * EW * EP / 16; i += NTHR) reinterpret_cast<VT*>(E)[i] = vec_zero<T>();
#pragma unroll
    for (int jt = 0; jt < WT; ++jt)
#pragma unroll
        for (int jc = 0; jc < WC; ++jc) {
            const int tap = (tid >> 5) + jt * TPP, c = (tid & 31) + 32 * jc;
            if (tap < K * K && c < ccur) s_w[tap * CC + c] = wreg[jt][jc];
        }
    lds_barrier();
    STAMP(1);

    AF gq[PF];                                 // GATED: the crop's gate in the operand fragments' layout (Cin <= 2 V PF: one group of k-steps)
    if constexpr (GATED) {
#pragma unroll
        for (int u = 0; u < PF; ++u) gq[u] = (u * 2 * V + g * V < Cin) ? OPS::load_a(reinterpret_cast<const T*>(in_gate) + size_t(b) * Cin + u * 2 * V + g * V) : OPS::zero_a();
    }
    for (int t = wave; t < ntask; t += NWAVE) {
        if constexpr (GATED) {
#pragma unroll
            for (int u = 0; u < PF; ++u) OPS::gate_by(a[u], gq[u]);
        }
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // Round 4: (1) k-steps past KSe are not multiplied (their operands are zeros: same sum; in f32 a k-step is four 64-cycle
        // MFMAs and Cin = 16 / 24 / 40 fill 2 / 3 / 5 of 4 / 4 / 8 slots); (2) deep contractions are pipelined in the SAME
        // registers: while one half of the PF slots is multiplied the other half's next k-steps travel (round 3 loaded a whole
        // group and waited for it before every 4 k-steps).
        constexpr int HF = PF / 2;
        for (int ks = 0; ks < KSe; ks += PF) {
#pragma unroll
            for (int u = 0; u < HF; ++u)
                if (ks + u < KSe) OPS::step(w[u], OPS::prep(a[u]), acc);    // (wave-uniform)
            if (ks + PF < KSe) load_half(cur, ks + PF, 0, w, a);
#pragma unroll
            for (int u = HF; u < PF; ++u)
                if (ks + u < KSe) OPS::step(w[u], OPS::prep(a[u]), acc);
            if (ks + PF + HF < KSe) load_half(cur, ks + PF + HF, HF, w, a);
        }
        const Task nxt = make_task(t + NWAVE);
        if (t + NWAVE < ntask) load_ops(nxt, 0, w, a);
        if (cur.valid) {
            unsigned char* epix = E + cur.eoff;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int nl = cur.tl * 32 + 8 * qq + 4 * g;
                if (cur.tl * 32 + 8 * qq < ccur) {          // (wave-uniform: chunk widths are multiples of 8)
                    OT o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = T(conv_swish<T>(SP ? fmaf(acc[4 * qq + r], wsi, bv[qq][r]) : acc[4 * qq + r] + bv[qq][r]));
                    *reinterpret_cast<OT*>(epix + nl * SZ) = o;
                }
            }
        }
        if (t + NWAVE < ntask && nxt.tl != cur.tl) load_bias(nxt, bv);     // (wave-uniform, rare)
        cur = nxt;
    }
    STAMP(2);
    // first W1V reduce-kernel values of this lane's output (used after the taps: see the SE half below)
    constexpr int W1V = 16;
    const int fj = tid >> 2, fq = tid & 3;
    float w1v[W1V];
#pragma unroll
    for (int i = 0; i < W1V; ++i) w1v[i] = 0.f;
    if (w1t != nullptr && __builtin_amdgcn_readfirstlane(wave) * 16 < R) {      // (wave-uniform: waves without an output fj < R skip the loads)
        const float* wrow = w1t + size_t(fj < R ? fj : 0) * Cexp + c0;
#pragma unroll
        for (int i = 0; i < W1V; ++i) {
            const int c = fq + 4 * i;
            w1v[i] = (fj < R && c < ccur) ? wrow[c] : 0.f;
        }
    }
    lds_barrier();
    STAMP(3);

    // ---- depthwise taps out of E: lane = (4-channel group cg, strip sidx) ---------------------
    const int CG = ccur / VC;
    const int NPC = (NTHR / CG < TH * NSX) ? NTHR / CG : TH * NSX;    // tap slots (strips) in use
    const int sidx = fdiv(tid, __builtin_amdgcn_rcpf(float(CG)));
    const int cg = tid - sidx * CG;
    const int ty = fdiv(sidx, __builtin_amdgcn_rcpf(float(NSX))), sx = sidx - ty * NSX;
    const int oy = oy0 + ty;
    const bool lane_ok = sidx < NPC;
    const bool active = lane_ok && (sidx < TH * NSX) && (oy < Ho);
    float acc[P][VC];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int v = 0; v < VC; ++v) acc[p][v] = 0.0f;
    const unsigned char* row0 = E + ((ty * S) * EW + sx * P * S) * EP + cg * VC * SZ;
    const int row_pitch = EW * EP;
    if (active) {
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            float wr[K][VC];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float4v wv = *reinterpret_cast<const float4v*>(s_w + (ky * K + kx) * CC + cg * VC);
#pragma unroll
                for (int v = 0; v < VC; ++v) wr[kx][v] = wv[v];
            }
            const unsigned char* row = row0 + ky * row_pitch;
#pragma unroll
            for (int ix = 0; ix < NIX; ++ix) {
                const VCT xv = *reinterpret_cast<const VCT*>(row + ix * EP);
                float xf[VC];
#pragma unroll
                for (int v = 0; v < VC; ++v) xf[v] = float(xv[v]);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int d = ix - kx;
                    if (d >= 0 && (d % S) == 0 && (d / S) < P) {
#pragma unroll
                        for (int v = 0; v < VC; ++v) acc[d / S][v] = fmaf(xf[v], wr[kx][v], acc[d / S][v]);
                    }
                }
            }
        }
    }
    STAMP(4);
    float sum[VC] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float4v bs = *reinterpret_cast<const float4v*>(bd + c0 + cg * VC);
        T* dst = out + ((size_t(b) * Ho + oy) * Ho + (ox0 + sx * P)) * Cexp + c0 + cg * VC;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            VCT o;
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const float y = conv_swish<T>(acc[p][v] + bs[v]);
                sum[v] += y;
                o[v] = T(y);
            }
            *reinterpret_cast<VCT*>(dst + size_t(p) * Cexp) = o;
        }
    }
    STAMP(5);
    lds_barrier();                                     // every lane is done reading E (stores stay in flight)
    if (lane_ok) {
#pragma unroll
        for (int v = 0; v < VC; ++v) s_red[sidx * ccur + cg * VC + v] = sum[v];
    }
    lds_barrier();
    // ---- squeeze-excite, first half: this workgroup's share of the reduce conv ------------------
    // The tile's channel sums (fixed order) stay on the CU; since se_reduce is linear, the sum over
    // channels can be taken chunk by chunk and tile by tile: rpart[j] = sum_{c in chunk}
    // tilesum[c] * W1[c][j].  se.hip adds the (tiles x chunks) partial vectors of a crop in fixed
    // order, scales by 1/(H*W) and finishes the block.  This spreads the reduce kernel (221 KB for
    // C = 1152) over every workgroup of the layer instead of streaming it through one CU per crop.
    float* s_sum = s_w;                                  // the depthwise taps are no longer needed
    if (tid < ccur) {
        float t = 0.0f;
        for (int s = 0; s < NPC; ++s) t += s_red[s * ccur + tid];
        if (w1t == nullptr) {
            // plain form (wide early layers, whose SE kernels are tiny): the tile's channel sums
            // go to se.hip's full kernel: rpart is [n][ntiles][Cexp] here
            rpart[(size_t(b) * ntiles + tile) * Cexp + c0 + tid] = t;
        }
        s_sum[tid] = t;
    }
    if (w1t == nullptr) return;
    lds_barrier();
    {
        // 4 lanes per output j: lane q sums channels q, q+4, ..; combined (a0+a1)+(a2+a3)
        const int j = fj, q = fq;
        float acc = 0.0f;
        if (j < R) {
            const float* wrow = w1t + size_t(j) * Cexp + c0;
#pragma unroll
            for (int i = 0; i < W1V; ++i) {
                const int c = q + 4 * i;
                if (c < ccur) acc = fmaf(s_sum[c], w1v[i], acc);
            }
            for (int c = q + 4 * W1V; c < ccur; c += 4) acc = fmaf(s_sum[c], wrow[c], acc);
        }
        const float o1 = __shfl_xor(acc, 1, 64);
        const float pair = (q & 1) ? (o1 + acc) : (acc + o1);          // (a_even + a_odd) on both lanes
        const float o2 = __shfl_xor(pair, 2, 64);
        const float tot = (q & 2) ? (o2 + pair) : (pair + o2);         // (a0+a1) + (a2+a3)
        if (q == 0 && j < RP)
            rpart[((size_t(b) * ntiles + tile) * chunks + chunk) * RP + j] = (j < R) ? tot : 0.0f;
    }
    STAMP(6);
}

template <typename T, int K, int S, int NTHR, bool SP = false, bool GATED = false>
void launch_t(const FrontArgs& a, hipStream_t stream) {
    const FrontPlan& p = a.plan;
    dim3 grid(unsigned(p.tiles_x * p.tiles_y) * unsigned(p.chunks) * unsigned(a.n));     // 1-D: the kernel deals the workgroups to the XCDs (xcd_unit())
    WHENET_REQUIRE(p.lds_bytes <= 160 * 1024, WHENET_EINVAL, "front: the tile plan needs more than 160 KB of LDS");
    static std::atomic<bool> attr[64];           // (zero-initialised, one per instantiation; handles are one per host thread)
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (p.lds_bytes > 64 * 1024 && dev >= 0 && dev < 64 && !attr[dev].load(std::memory_order_acquire)) {
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_front_kernel<T, K, S, NTHR, SP, GATED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((whenet_front_kernel<T, K, S, NTHR, SP, GATED>), grid, dim3(NTHR), p.lds_bytes, stream,
                       static_cast<const T*>(a.x), static_cast<const T*>(SP ? a.weps : a.wep), a.be, a.wd, a.bd,
                       static_cast<T*>(a.out), a.rpart, a.H, a.Ho, a.Cin, a.Cexp, a.pad, SP ? a.KSes : a.KSe, a.NTe, p.CC, p.TH, p.NSX,
                       p.tiles_x, p.EH, p.EW, p.EP, p.w_off, a.w1t, a.R, (a.R + 3) & ~3, a.wsi, a.in_gate, p.tiles_x * p.tiles_y, p.chunks, a.n, a.xcd_grouped ? 1 : 0);
    WHENET_HIP_CHECK(hipGetLastError());
}

template <typename T, int NTHR, bool SP = false>
void launch_ks(const FrontArgs& a, hipStream_t stream) {
    if constexpr (SP) {
        if (a.in_gate != nullptr) {                 // block 2 with block 1's project folded in (the only gated shape)
            WHENET_REQUIRE(a.k == 3 && a.s == 2 && a.Cin == 32 && a.KSes == 2, WHENET_EINVAL, "front: the gated-input form exists for block 2's shape only");
            return launch_t<T, 3, 2, NTHR, true, true>(a, stream);
        }
    }
    if (a.k == 3 && a.s == 1) launch_t<T, 3, 1, NTHR, SP>(a, stream);
    else if (a.k == 3 && a.s == 2) launch_t<T, 3, 2, NTHR, SP>(a, stream);
    else if (a.k == 5 && a.s == 1) launch_t<T, 5, 1, NTHR, SP>(a, stream);
    else if (a.k == 5 && a.s == 2) launch_t<T, 5, 2, NTHR, SP>(a, stream);
    else throw Error(WHENET_EINVAL, "front: unsupported kernel/stride");
}

template <typename T, bool SP = false>
void launch_thr(const FrontArgs& a, hipStream_t stream) {
    switch (a.plan.threads) {
        case 256: launch_ks<T, 256, SP>(a, stream); break;
        case 512: launch_ks<T, 512, SP>(a, stream); break;
        case 1024: launch_ks<T, 1024, SP>(a, stream); break;
        default: throw Error(WHENET_EINVAL, "front: threads must be 256, 512 or 1024");
    }
}

}  // namespace

// Tile-shape search (same idea as plan_dw): chunk width CC (multiples of 32, or the whole
// layer), TH output rows, NSX 7-pixel strips; 256 lanes; LDS <= 64 KiB.  Score = useful lanes x
// halo efficiency (also the expand recompute factor) x occupancy.
namespace {

// LDS cycles per depthwise tap read of one wave-instruction, averaged over the lane groups of the
// 256 tap lanes, for pixel pitch EP (MI355X_MICROARCH.md, LDS: ds_read_b64 = 2 groups of 32 lanes,
// ds_read_b128 = 4 groups of 16 lanes in a fixed interleave; bank = (addr / 4) mod 64; a group costs
// as many cycles as its busiest bank has distinct dwords).  1.0 = conflict-free.
double tap_read_conflicts(int SZ, int cc, int TH, int NSX, int EW, int s, int EP) {
    static const int g128[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int CG = cc / VC, bytes = VC * SZ, ngroups = (bytes == 16) ? 4 : 2, glanes = 64 / ngroups;
    double total = 0.0;
    int counted = 0;
    for (int wave = 0; wave < 4; ++wave) {
        for (int g = 0; g < ngroups; ++g) {
            int seen[64][16];
            int nseen[64] = {};
            bool any = false;
            for (int l = 0; l < glanes; ++l) {
                const int lane = (bytes == 16) ? g128[g][l] : g * 32 + l;
                const int tid = wave * 64 + lane;
                const int cg = tid % CG, sidx = tid / CG;
                if (sidx >= TH * NSX) continue;
                const int ty = sidx / NSX, sx = sidx - ty * NSX;
                const int addr = ((ty * s) * EW + sx * P * s) * EP + cg * bytes;
                any = true;
                for (int d = 0; d < bytes; d += 4) {
                    const int dw = (addr + d) / 4, bank = dw % 64;
                    bool dup = false;
                    for (int i = 0; i < nseen[bank]; ++i) dup = dup || seen[bank][i] == dw;
                    if (!dup && nseen[bank] < 16) seen[bank][nseen[bank]++] = dw;
                }
            }
            if (!any) continue;
            int worst = 1;
            for (int b = 0; b < 64; ++b) worst = nseen[b] > worst ? nseen[b] : worst;
            total += worst;
            ++counted;
        }
    }
    return counted ? total / counted : 1.0;
}

// pixel pitch: the channels alone, or 16 bytes of padding when that has fewer tap-read bank
// conflicts (larger paddings remove the conflicts of the stride-2 layers too, but cost a workgroup
// per CU in LDS: measured slower)
int choose_pitch(int SZ, int cc, int TH, int NSX, int EW, int s) {
    const double c0 = tap_read_conflicts(SZ, cc, TH, NSX, EW, s, cc * SZ);
    const double c16 = tap_read_conflicts(SZ, cc, TH, NSX, EW, s, cc * SZ + 16);
    return (c0 <= c16 + 1e-9) ? cc * SZ : cc * SZ + 16;
}

}  // namespace

// Every tile plan that fits: chunk width CC (multiples of 32, or the whole layer), TH output rows,
// NSX 7-pixel strips; 256 tap lanes; LDS <= 64 KiB.  `score` is the a-priori figure of merit used
// for shapes outside the tuned table: useful tap lanes x halo efficiency (also the expand
// recompute factor) x occupancy.
std::vector<FrontPlan> plan_front_candidates(int dtype, int k, int s, int H, int Ho, int Cexp,
                                             std::vector<double>* scores) {
    constexpr int NTHR = 256;           // tap lanes the tile is planned for (the kernel may run more lanes)
    const int SZ = (dtype == WHENET_F16) ? 2 : 4;
    WHENET_REQUIRE(Cexp % 8 == 0 && Ho % P == 0, WHENET_EINVAL, "front: unsupported geometry");
    const int spr = Ho / P;
    std::vector<FrontPlan> out;
    for (int CC = 32; CC <= 160; CC += 32) {
        int cc = CC;
        if (cc > Cexp) cc = Cexp;
        if (cc != Cexp && cc % 32) continue;
        if (k == 5 && cc > 128) continue;          // the kernel keeps the chunk's taps in <= 13 registers per lane
        const int CG = cc / VC;
        if (CG > NTHR) continue;
        const int NS = NTHR / CG;
        for (int NSX = 1; NSX <= spr; ++NSX) {
            if (spr % NSX) continue;
            if (NSX > NS) break;
            int last_th = -1;
            for (int tiles_y = 1; tiles_y <= Ho; ++tiles_y) {
                const int TH = ceil_div(Ho, tiles_y);
                if (TH == last_th) continue;
                last_th = TH;
                if (TH * NSX > NS) continue;
                const int TW = NSX * P;
                const int EH = (TH - 1) * s + k, EW = (TW - 1) * s + k;
                const int EP_model = choose_pitch(SZ, cc, TH, NSX, EW, s);
                for (int pad = 0; pad <= 48; pad += 16) {
                const int EP = cc * SZ + pad;
                size_t tile_bytes = size_t(EH) * EW * EP;
                const size_t red_bytes = size_t(NTHR) * VC * 4;              // [NTHR/CG][cc] floats
                if (red_bytes > tile_bytes) tile_bytes = red_bytes;
                const size_t w_off = (tile_bytes + 15) & ~size_t(15);
                const size_t lds = w_off + size_t(k) * k * cc * 4;
                if (lds > 64 * 1024) continue;
                const int chunks = ceil_div(Cexp, cc);
                const double lane_use = double(Ho) * NSX * CG / (double(ceil_div(Ho, TH)) * NTHR) *
                                        (double(Cexp) / (double(chunks) * cc));
                const double halo = double(TH * s) * (TW * s) / (double(EH) * EW);
                int blocks_cu = int((160 * 1024) / lds);
                if (blocks_cu > 8) blocks_cu = 8;
                const double waves_cu = double(blocks_cu) * NTHR / 64.0;
                const double occ = waves_cu >= 16.0 ? 1.0 : waves_cu / 16.0;
                FrontPlan p;
                p.threads = 256;
                p.CC = cc;
                p.TH = TH;
                p.NSX = NSX;
                p.tiles_x = spr / NSX;
                p.tiles_y = ceil_div(Ho, TH);
                p.chunks = chunks;
                p.EH = EH;
                p.EW = EW;
                p.EP = EP;
                p.w_off = int(w_off);
                p.lds_bytes = lds;
                out.push_back(p);
                // (pitches other than the bank model's choice are candidates for the tuner only)
                if (scores) scores->push_back(EP == EP_model ? lane_use * (0.35 + 0.65 * halo) * (0.4 + 0.6 * occ) : 0.0);
                }
            }
        }
    }
    (void)H;
    return out;
}

namespace {
// Plans measured on MI355X for EfficientNet-B0's fifteen layer shapes, f16 (round 1) and f32 (round 4) (tools/probes/
// front_tune.hip: every candidate timed at 64 and 16 crops per launch; profiles/r01/
// front_tune_f16.txt).  The a-priori score above ranks candidates of one layer in roughly the
// right order but cannot see tail quantisation or the per-workgroup fixed costs.
struct TunedPlan { int k, s, H, Cexp, CC, TH, NSX, pad; };
const TunedPlan TUNED_F16[] = {
#include "front_tuned_f16.inc"
};
const TunedPlan TUNED_F32[] = {
#include "front_tuned_f32.inc"
};
}  // namespace

FrontPlan plan_front(int dtype, int k, int s, int H, int Ho, int Cexp) {
    std::vector<double> scores;
    const std::vector<FrontPlan> cand = plan_front_candidates(dtype, k, s, H, Ho, Cexp, &scores);
    WHENET_REQUIRE(!cand.empty(), WHENET_EINVAL, "front: no tile plan fits");
    static const bool no_tuned = getenv("WHENET_FRONT_NO_TUNED") != nullptr;       // (probes only; read once)
    if (!no_tuned) {
        const bool f16 = dtype == WHENET_F16;
        const TunedPlan* tb = f16 ? TUNED_F16 : TUNED_F32;
        const size_t nt = f16 ? sizeof(TUNED_F16) / sizeof(TunedPlan) : sizeof(TUNED_F32) / sizeof(TunedPlan);
        for (size_t i = 0; i < nt; ++i) {
            const TunedPlan& t = tb[i];
            if (t.k == k && t.s == s && t.H == H && t.Cexp == Cexp)
                for (const FrontPlan& p : cand)
                    if (p.CC == t.CC && p.TH == t.TH && p.NSX == t.NSX && p.EP == t.CC * (f16 ? 2 : 4) + t.pad) return p;
        }
    }
    size_t best = 0;
    for (size_t i = 1; i < cand.size(); ++i)
        if (scores[i] > scores[best] + 1e-9) best = i;
    return cand[best];
}

// Lanes per workgroup for a launch of n crops.  The tile (and every result bit) is the same either
// way: extra waves only share the expand tasks.  When the whole launch fits on the chip at once
// (<= 2 workgroups per CU) the kernel is one workgroup's critical path, so 8 waves shorten it;
// beyond that 4-wave workgroups pack the CUs better.
int front_threads(const FrontPlan& p, int n) {
    static const int forced = [] { const char* e = getenv("WHENET_FRONT_THREADS"); return e ? atoi(e) : 0; }();   // probes only
    if (forced) return forced;
    return (long(n) * p.ntiles() * p.chunks <= 512) ? 512 : 256;
}

void launch_front(const FrontArgs& a, int dtype, hipStream_t stream) {
    WHENET_REQUIRE(!a.split || (dtype == WHENET_F32 && a.weps != nullptr && a.KSes == ceil_div(a.Cin, 16)), WHENET_EINVAL,
                   "front: the split-product form needs float32 storage and the split weight images");
    WHENET_REQUIRE(a.in_gate == nullptr || a.split, WHENET_EINVAL, "front: the gated-input form needs the split-product form");
    if (dtype == WHENET_F16) launch_thr<half_t>(a, stream);
    else if (a.split) launch_thr<float, true>(a, stream);
    else launch_thr<float>(a, stream);
}

std::string kernel_name_front(int dtype, int k, int s, int threads) {
    return std::string("whenet_front_kernel<") + (dtype == WHENET_F16 ? "_Float16" : "float") + ", " +
           std::to_string(k) + ", " + std::to_string(s) + ", " + std::to_string(threads) + ">";
}

}  // namespace whenet
