// Head conv fused with GlobalAveragePooling2D (round 4; f16, and f32 further down): Conv2D(320 -> 1280, 1x1, no bias) + BN + Swish on the 7 x 7 map,
// then the mean over the 49 positions -- the 1280 pooled features per crop are all that leaves the kernel.
//
// Reference: efficientnet 0.0.4's head conv (the last layer of EfficientNetB0(include_top=False), /root/reference/whenet.py:8)
// followed by GlobalAveragePooling2D (/root/reference/whenet.py:10); SURVEY.md section 2.2 planned them as one kernel.
//
// Round 3 kept them apart: with 32-row MFMA tiles running over the rows of the whole launch, a crop's 49 rows fall into tiles
// at offsets that depend on its position in the batch, so per-tile partial sums would have made a crop's features depend on
// where it travels.  Here (as front7.hip) a workgroup owns a GROUP of G crops x a chunk of 64 out-channels and every crop
// gets its OWN two strips: strip 2j = pixels 0..31 of crop j, strip 2j + 1 = pixels 32..48 (+ 15 idle rows) -- the same
// grouping of the 49 values for every crop wherever it sits: bitwise batch invariance is kept, at the price of 64 instead
// of 49 MFMA rows per crop (the matrix pipe is 95 % idle on this path).  With G = 4 the 8 strips are the 8 waves.
//   * the chunk's weights (20 k-steps x 2 tiles x 1 KB = 40 KB) are staged once in LDS; every wave loads the 20 k-steps of
//     its strip's pixel rows in one round trip;
//   * BN bias + Swish in f32, pooled in f32 BEFORE any rounding (round 3 rounded the 49 x 1280 tensor to f16, wrote it --
//     125 KB per crop -- and the heads kernel read it back to pool it);
//   * per lane: the 16 values of its channel in fixed order; per channel: ((strip 0, g 0) + (strip 0, g 1)) + ((strip 1, g 0) +
//     (strip 1, g 1)), times 1 / 49.
// HBM bytes per crop: 49 * 320 * 2 in (x 20 chunks, L2 hits) + 1280 * 4 out.
#include "device_math.h"
#include "kernels.h"

#include <atomic>
#include <string>

namespace whenet {

namespace {

constexpr int H7_K = 320, H7_KS = H7_K / 16, H7_HW = 49, H7_NC = 64, H7_NT = H7_NC / 32;

template <int G, int NTHR>
__global__ __launch_bounds__(NTHR) void whenet_head7_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wep,
                                                            const float* __restrict__ bias, float* __restrict__ feat, int n,
                                                            int NTILES, int N, int xcd) {
    constexpr int NWAVE = NTHR / 64;
    constexpr int nstrip = 2 * G;
    constexpr int KS = H7_KS, NT = H7_NT, NC = H7_NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const half8* Wl = reinterpret_cast<const half8*>(smem);                     // [KS][NT][64 lanes]
    float* s_gap = reinterpret_cast<float*>(smem + KS * NT * 1024);             // [nstrip][2 (g)][NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    int grp, chunk;
    xcd_unit(int(blockIdx.x), (n + G - 1) / G, N / NC, grp, chunk, xcd != 0);           // (device_math.h)
    const int c0 = chunk * NC;
    const int crop0 = grp * G;
    const int nlast = n - 1;

    // ---- prologue: the chunk's weights -> LDS, this wave's strip of pixel rows -> registers -------------------------------
    constexpr int nwv = KS * NT * 64;
    constexpr int WV = (nwv + NTHR - 1) / NTHR;
    half8 wstage[WV];
    {
        const half8* src = reinterpret_cast<const half8*>(wep);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) {
                const int ks = v / (NT * 64), r = v % (NT * 64);
                wstage[i] = src[(size_t(ks) * NTILES + (c0 >> 5)) * 64 + r];
            }
        }
    }
    // few strips (groups of 2 crops): one (strip, tile) task per wave
    constexpr bool SPLIT = nstrip * NT <= NWAVE;
    const int strip = SPLIT ? (wave < nstrip * NT ? wave % nstrip : nstrip) : wave;
    const int t_lo = SPLIT ? wave / nstrip : 0, t_hi = SPLIT ? t_lo + 1 : NT;
    half8 a[KS];
    if (strip < nstrip) {
        int gc = crop0 + (strip >> 1);
        gc = gc < nlast ? gc : nlast;
        int px = (strip & 1) * 32 + lm;
        px = px < H7_HW ? px : H7_HW - 1;                       // (idle rows: any valid address, masked out of the sum)
        const unsigned char* xr = reinterpret_cast<const unsigned char*>(x) + (size_t(gc) * H7_HW + px) * (H7_K * 2) + g * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const half8*>(xr + ks * 32);
    }
    float bias_t[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_t[t] = bias[c0 + t * 32 + lm];
    {
        half8* dst = reinterpret_cast<half8*>(smem);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) dst[v] = wstage[i];
        }
    }
    lds_barrier();

    // ---- the conv, BN + Swish, and this lane's share of the pooling -------------------------------------------------------
    if (strip < nstrip) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < t_lo || t >= t_hi) continue;               // (uniform)
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias_t[t];   // (BN bias as the accumulators' initial value)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], Wl[(ks * NT + t) * 64 + lane], acc, 0, 0, 0);
            float sum = 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float2v y0 = swish2(float2v{acc[4 * qq], acc[4 * qq + 1]});
                const float2v y1 = swish2(float2v{acc[4 * qq + 2], acc[4 * qq + 3]});
                const float y[4] = {y0[0], y0[1], y1[0], y1[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = (strip & 1) * 32 + 8 * qq + 4 * g + r;      // this value's pixel of the crop
                    sum += (px < H7_HW) ? y[r] : 0.f;
                }
            }
            s_gap[(strip * 2 + g) * NC + t * 32 + lm] = sum;
        }
    }
    lds_barrier();
    // ---- mean over the 49 positions, fixed order ---------------------------------------------------------------------------
#pragma unroll
    for (int i0 = 0; i0 < G * NC; i0 += NTHR) {
        const int i = i0 + tid;
        if (i < G * NC) {
            const int cr = i / NC, c = i % NC;
            if (crop0 + cr < n && c0 + c < N) {
                const float* s0 = s_gap + ((2 * cr) * 2) * NC + c;
                feat[size_t(crop0 + cr) * N + c0 + c] = ((s0[0] + s0[NC]) + (s0[2 * NC] + s0[3 * NC])) * (1.0f / 49.0f);
            }
        }
    }
}

// ---- f32 (the parity configuration) ---------------------------------------------------------------------------------------
// The same decomposition on v_mfma_f32_32x32x2_f32 (exact f32 products, 64 cycles per instruction: this kernel is bound by
// the matrix pipe, 64 crops x 64 rows x 320 x 1280 MACs = 21 us at the 157 TF peak).  A workgroup owns G crops x ONE 32-channel
// tile (40 KB of weights in LDS, 40 k-steps of 8); wave = strip; the strip's activation rows are streamed in groups of 5 k-steps
// (one 16-byte load per lane and k-step: 4 consecutive k of the lane's pixel row, element t feeds the t-th instruction, as
// pw.hip), the next group in flight while the current one multiplies; every load is unconditional (clamped addresses) so that
// the waits are counted.  Epilogue, pooling order and the batch-invariance argument are the f16 kernel's.
// Round 3/4's f32 path until now: the split-K GEMM wrote the 49 x 1280 f32 tensor (16 MB per 64 crops) and the heads kernel
// read it back: 50 + 15 us per 64 crops.
constexpr int H7F_KS = H7_K / 8, H7F_NC = 32, H7F_KG = 5, H7F_NGRP = H7F_KS / H7F_KG;

// SP (WHENET_F32S, round 5): the same kernel with the products as binary16 hi/lo pairs on the f16 matrix cores
// (device_math.h PwOps<float, true>): 20 k-steps of 16, three v_mfma_f32_32x32x16_f16 each instead of eight v_mfma_f32_32x32x2_f32;
// the LDS image is [k-step][hi | lo][64 lanes] half8 -- the same 40 KB; wsi = 2^-shift of the scaled weights.
template <int G, bool SP>
__global__ __launch_bounds__(128 * G) void whenet_head7_f32_kernel(const float* __restrict__ x, const float* __restrict__ wep,
                                                                   const float* __restrict__ bias, float* __restrict__ feat, int n,
                                                                   int NTILES, int N, float wsi, int xcd) {
    constexpr int NTHR = 128 * G, nstrip = 2 * G, NC = H7F_NC, KG = H7F_KG;
    constexpr int KS = SP ? H7_K / 16 : H7F_KS;                                  // k-steps: 20 of 16 | 40 of 8
    constexpr int NGRP = KS / KG;
    constexpr int KF = SP ? 16 : 8;                                              // floats of a pixel row per k-step
    (void)nstrip;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float4v* Wl = reinterpret_cast<const float4v*>(smem);                 // f32: [KS][64 lanes]; SP: [KS][2][64 lanes] (16 B each)
    float* s_gap = reinterpret_cast<float*>(smem + H7F_KS * 1024);              // [nstrip][2 (g)][NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int strip = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    int grp, chunk;
    xcd_unit(int(blockIdx.x), (n + G - 1) / G, N / NC, grp, chunk, xcd != 0);
    const int c0 = chunk * NC;
    const int crop0 = grp * G;

    constexpr int nwv = H7F_KS * 64;                                            // 16-byte vectors of the staged image (both forms)
    static_assert(nwv % NTHR == 0, "weight staging: whole vectors per lane");
    constexpr int WV = nwv / NTHR;
    float4v wstage[WV];
    {
        const float4v* src = reinterpret_cast<const float4v*>(wep);
        const size_t w_lo = size_t(KS) * NTILES * 64;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if constexpr (SP) wstage[i] = src[(((v >> 6) & 1) ? w_lo : 0) + (size_t(v >> 7) * NTILES + (c0 >> 5)) * 64 + (v & 63)];
            else wstage[i] = src[(size_t(v >> 6) * NTILES + (c0 >> 5)) * 64 + (v & 63)];
        }
    }
    int gc = crop0 + (strip >> 1);
    gc = gc < n - 1 ? gc : n - 1;
    int pxl = (strip & 1) * 32 + lm;
    pxl = pxl < H7_HW ? pxl : H7_HW - 1;                        // (idle rows: any valid address, masked out of the sum)
    const float* xr = x + (size_t(gc) * H7_HW + pxl) * H7_K + g * (KF / 2);
    using OPS = PwOps<float, SP>;
    using AF = typename OPS::A;
    AF a0[KG], a1[KG];
    auto load_group = [&](AF (&a)[KG], int grp) {
#pragma unroll
        for (int u = 0; u < KG; ++u) a[u] = OPS::load_a(xr + (grp * KG + u) * KF);
    };
    load_group(a0, 0);
    const float bias_v = bias[c0 + lm];
    {
        float4v* dst = reinterpret_cast<float4v*>(smem);
#pragma unroll
        for (int i = 0; i < WV; ++i) dst[tid + i * NTHR] = wstage[i];
    }
    lds_barrier();

    float16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](const AF (&a)[KG], int grp) {
#pragma unroll
        for (int u = 0; u < KG; ++u) {
            if constexpr (SP) {
                const half8 whi = __builtin_bit_cast(half8, Wl[((grp * KG + u) * 2) * 64 + lane]);
                const half8 wlo = __builtin_bit_cast(half8, Wl[((grp * KG + u) * 2 + 1) * 64 + lane]);
                const typename OPS::P pa = OPS::prep(a[u]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa.hi, wlo, acc, 0, 0, 0);       // (operand roles swapped: rows = pixels)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa.lo, whi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa.hi, whi, acc, 0, 0, 0);
            } else {
                const float4v w = Wl[(grp * KG + u) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], w[t], acc, 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int grp = 0; grp < NGRP; grp += 2) {                  // (the scheduling barriers keep a whole group in flight: left to
        load_group(a1, grp + 1);                               //  itself the scheduler sinks every load to one k-step ahead of its use)
        __builtin_amdgcn_sched_barrier(0);
        compute(a0, grp);
        __builtin_amdgcn_sched_barrier(0);
        if (grp + 2 < NGRP) load_group(a0, grp + 2);           // (compile time)
        __builtin_amdgcn_sched_barrier(0);
        compute(a1, grp + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = (strip & 1) * 32 + 8 * (r >> 2) + 4 * g + (r & 3);      // this value's pixel of the crop
        const float y = conv_swish<float>(SP ? fmaf(acc[r], wsi, bias_v) : acc[r] + bias_v);
        sum += (px < H7_HW) ? y : 0.f;
    }
    s_gap[(strip * 2 + g) * NC + lm] = sum;
    lds_barrier();
    // ---- mean over the 49 positions, fixed order ---------------------------------------------------------------------------
    if (tid < G * NC) {
        const int cr = tid / NC, c = tid % NC;
        if (crop0 + cr < n && c0 + c < N) {
            const float* s0 = s_gap + ((2 * cr) * 2) * NC + c;
            feat[size_t(crop0 + cr) * N + c0 + c] = ((s0[0] + s0[NC]) + (s0[2 * NC] + s0[3 * NC])) * (1.0f / 49.0f);
        }
    }
}

template <int G>
void launch_h7_f32(const Head7Args& a, hipStream_t stream) {
    const size_t lds = size_t(H7F_KS) * 1024 + size_t(2 * G) * 2 * H7F_NC * 4;
    if (a.split)
        hipLaunchKernelGGL((whenet_head7_f32_kernel<G, true>), dim3(unsigned(a.N / H7F_NC) * unsigned(ceil_div(a.n, G))), dim3(128 * G), lds, stream,
                           static_cast<const float*>(a.x), static_cast<const float*>(a.weps), a.bias, a.feat, a.n, a.NTILES, a.N, a.wsi, a.xcd_grouped ? 1 : 0);
    else
        hipLaunchKernelGGL((whenet_head7_f32_kernel<G, false>), dim3(unsigned(a.N / H7F_NC) * unsigned(ceil_div(a.n, G))), dim3(128 * G), lds, stream,
                           static_cast<const float*>(a.x), static_cast<const float*>(a.wep), a.bias, a.feat, a.n, a.NTILES, a.N, 1.0f, a.xcd_grouped ? 1 : 0);
    WHENET_HIP_CHECK(hipGetLastError());
}

template <int G>
void launch_h7(const Head7Args& a, hipStream_t stream) {
    constexpr int NTHR = 512;
    const size_t lds = size_t(H7_KS) * H7_NT * 1024 + size_t(2 * G) * 2 * H7_NC * 4;
    hipLaunchKernelGGL((whenet_head7_kernel<G, NTHR>), dim3(unsigned(a.N / H7_NC) * unsigned(ceil_div(a.n, G))), dim3(NTHR), lds, stream,
                       static_cast<const half_t*>(a.x), static_cast<const half_t*>(a.wep), a.bias, a.feat, a.n, a.NTILES, a.N, a.xcd_grouped ? 1 : 0);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace

bool head7_supported(int dtype, int K, int N, int HW) {
    return (dtype == WHENET_F16 || dtype == WHENET_F32) && K == H7_K && HW == H7_HW && N % H7_NC == 0;
}

// groups of 4 crops (8 strips = the 8 waves) from 17 crops per launch up, of 2 below (as front7.hip; the group size changes no
// bit of a crop's features)
void launch_head7(const Head7Args& a, hipStream_t stream) {
    WHENET_REQUIRE(head7_supported(a.dtype, a.K, a.N, 49) && a.n >= 1 && a.x && a.wep && a.bias && a.feat, WHENET_EINVAL,
                   "head7: the 320 -> N (multiple of 64) head conv on 7 x 7 maps");
    if (a.dtype == WHENET_F16) {
        if (a.n <= 16) launch_h7<2>(a, stream);
        else launch_h7<4>(a, stream);
    } else {
        if (a.n <= 16) launch_h7_f32<2>(a, stream);
        else launch_h7_f32<4>(a, stream);
    }
}

std::string kernel_name_head7(int dtype, int n, bool split) {
    if (dtype == WHENET_F16) return std::string("whenet_head7_kernel<") + (n <= 16 ? "2" : "4") + ", 512>";
    return std::string("whenet_head7_f32_kernel<") + (n <= 16 ? "2" : "4") + (split ? ", true>" : ", false>");
}

}  // namespace whenet
