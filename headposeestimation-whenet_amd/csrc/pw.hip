// 1x1 convolutions (expand / project / head conv) as MFMA GEMMs over M = n*H*W pixel rows.
//
// Reference: efficientnet 0.0.4's Conv2D(.., 1x1, 'same', no bias) -> BN (-> Swish), the SE
// `Multiply([gate, x])` in front of the project conv and the identity-skip `Add` behind it,
// all instantiated by /root/reference/whenet.py:8 (33 such convs, 88 % of
// the MACs).  Fused here:  out = act( (a * gate[crop]) @ Wfolded + bias ) + skip.
//
// MFMA mapping (gfx950, wave64).  The product is computed TRANSPOSED:
//     D[i = out-channel n][j = pixel row m]  =  sum_k  Wt[n][k] * Act[m][k]
// i.e. the weights are the MFMA "A" operand and the activations the "B" operand, because
//   (1) both operands are then k-contiguous: a lane's B fragment is 16 contiguous bytes of
//       its own NHWC pixel row (global_load_dwordx4, no LDS transpose), and its A fragment is
//       16 contiguous bytes of the host-packed weight image (snapshot.h), so one wave reads a
//       dense 1 KiB block per tile-step;
//   (2) the 32x32 accumulator layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
//       leaves each lane with four runs of 4 consecutive out-channels of ONE pixel row:
//       the epilogue stores 8 B (f16) / 16 B (f32) pieces straight into NHWC, no shuffles.
//   f16: v_mfma_f32_32x32x16_f16, one instruction per 16-deep k-step (8 halfs per lane);
//   f32: v_mfma_f32_32x32x2_f32 x4 per 8-deep k-step (exact f32, bitwise an fmaf chain); a lane
//        loads 4 consecutive k of its row at once and feeds element t to the t-th instruction;
//        the packed weights follow the same (lane-group, t) -> k permutation.
// A wave owns 32 pixel rows x NT 32-wide out-channel tiles; a workgroup is 4 independent waves
// (128 rows) -- no LDS, no barriers: the weights of a layer are at most 0.8 MB and stay in L2,
// the activation rows are streamed exactly once per out-channel chunk.
// Workgroup -> tile mapping is XCD-aware: workgroups are dispatched round-robin over the 8 XCDs
// (id % 8), so the chunks of one 128-row block are given ids that are congruent mod 8 and
// adjacent in time: they hit the same XCD's L2 for the shared activation rows.
//
// HBM bytes per launch: M*(K + N)*sizeof(T) (+ M*N*sizeof(T) skip) ; FLOPs 2*M*K*N.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

template <typename T> struct Mfma;
template <> struct Mfma<half_t> {
    static __device__ __forceinline__ void step(const half8& w, const half8& a, float16v& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, acc, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    static __device__ __forceinline__ void step(const float4v& w, const float4v& a, float16v& acc) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], a[t], acc, 0, 0, 0);
    }
};

// NT  32-wide out-channel tiles per wave;  U  k-steps whose loads are issued together (software
// pipelining: (1+NT)*U 16-byte loads in flight per lane before the first MFMA of the group);
// SK  split-K factor: 1 = the 4 waves of a workgroup own 4 different 32-row strips,
//     4 = the 4 waves split the k-steps of ONE strip (interleaved) and combine through LDS --
//     used when there are too few rows to fill the chip (batch 1: M = 49..3136).
template <typename T, int NT, int U, int SK, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256) void whenet_pw_kernel(const T* __restrict__ A, const T* __restrict__ Wp,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ gate, const T* __restrict__ res,
                                                        T* __restrict__ out, int M, int K, int N, int KS, int NTILES,
                                                        int HW, int MT, int NCH) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int STRIPS = (SK == 1) ? 4 : 1;

    // XCD-aware decode of the 1-D grid (see header)
    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m0 = (mt * STRIPS + (SK == 1 ? wave : 0)) * 32;
    if (SK == 1 && m0 >= M) return;
    const int kpart = (SK == 1) ? 0 : wave;
    const int nt0 = nch * NT;
    const int g = lane >> 5;
    const int row = m0 + (lane & 31);
    const bool rvalid = row < M;
    const int rowc = rvalid ? row : (M - 1);

    const T* ap = A + size_t(rowc) * K + g * V;
    const float* gp = nullptr;
    if constexpr (GATE) gp = gate + size_t(rowc / HW) * K + g * V;
    const VT* wp = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64 + lane;

    float16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto load_a = [&](int ks) -> VT {
        VT a = vec_zero<T>();
        if (rvalid && ks * 2 * V + g * V < K) {
            a = *reinterpret_cast<const VT*>(ap + ks * 2 * V);
            if constexpr (GATE) {
                float f[V];
                vec_to_float<T>(a, f);
#pragma unroll
                for (int i = 0; i < V; i += 4) {
                    const float4v gv = *reinterpret_cast<const float4v*>(gp + ks * 2 * V + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
                }
                a = float_to_vec<T>(f);
            }
        }
        return a;
    };

    int ks = kpart;
    for (; ks + (U - 1) * SK < KS; ks += U * SK) {
        VT a[U];
        VT w[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const VT* wk = wp + size_t(ks + u * SK) * NTILES * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) w[u][t] = (nt0 + t < NTILES) ? wk[t * 64] : vec_zero<T>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = load_a(ks + u * SK);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (nt0 + t < NTILES) Mfma<T>::step(w[u][t], a[u], acc[t]);
    }
    for (; ks < KS; ks += SK) {
        const VT a = load_a(ks);
        const VT* wk = wp + size_t(ks) * NTILES * 64;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (nt0 + t < NTILES) Mfma<T>::step(wk[t * 64], a, acc[t]);
    }

    if constexpr (SK > 1) {
        __shared__ float s_red[(SK - 1) * NT * 16 * 64];
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[((wave - 1) * NT * 16 + t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SK - 1; ++w)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += s_red[(w * NT * 16 + t * 16 + r) * 64 + lane];
    }

    if (!rvalid) return;
    using OT = T __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= NTILES) continue;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int n0 = (nt0 + t) * 32 + 8 * qq + 4 * g;
            if (n0 >= N) continue;
            const float4v bv = *reinterpret_cast<const float4v*>(bias + n0);
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = acc[t][4 * qq + r] + bv[r];
                if constexpr (ACT == ACT_SWISH) y[r] = swish_f<IsF32<T>::value>(y[r]);
            }
            if constexpr (RES) {
                const OT rv = *reinterpret_cast<const OT*>(res + size_t(row) * N + n0);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
            }
            OT o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
            *reinterpret_cast<OT*>(out + size_t(row) * N + n0) = o;
        }
    }
}

// Scalar-FMA check kernel (option pw_impl=1): same operands (weights rounded to T, gate applied
// with the same single rounding), k-ordered fmaf chain per output.  Exists so that the MFMA
// fragment/accumulator mapping can be validated on the device against an independent kernel;
// it is never the default path.
template <typename T, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256) void whenet_pw_check_kernel(const T* __restrict__ A, const float* __restrict__ Wd,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ gate,
                                                              const T* __restrict__ res, T* __restrict__ out, int M,
                                                              int K, int N, int HW) {
    const int n4 = N / 4;
    const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= size_t(M) * n4) return;
    const int m = int(idx / n4);
    const int n0 = int(idx - size_t(m) * n4) * 4;
    const T* ap = A + size_t(m) * K;
    const float* gp = GATE ? gate + size_t(m / HW) * K : nullptr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        float a = float(ap[k]);
        if constexpr (GATE) a = float(T(a * gp[k]));
        const float4v w = *reinterpret_cast<const float4v*>(Wd + size_t(k) * N + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(a, w[r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = acc[r] + bias[n0 + r];
        if constexpr (ACT == ACT_SWISH) y = swish_f<IsF32<T>::value>(y);
        if constexpr (RES) y += float(res[size_t(m) * N + n0 + r]);
        out[size_t(m) * N + n0 + r] = T(y);
    }
}

template <typename T, int NT, int U, int SK, bool GATE, bool RES, int ACT>
void launch_mfma(const PwArgs& a, int MT, int NCH, hipStream_t stream) {
    const int blocks = 8 * ceil_div(MT, 8) * NCH;
    hipLaunchKernelGGL((whenet_pw_kernel<T, NT, U, SK, GATE, RES, ACT>), dim3(blocks), dim3(256), 0, stream,
                       static_cast<const T*>(a.a), static_cast<const T*>(a.wp), a.bias, a.gate,
                       static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.KS, a.NTILES, a.HW, MT,
                       NCH);
}

template <typename T, bool GATE, bool RES, int ACT>
void launch_variant(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    if (impl == 1) {
        const size_t work = size_t(a.M) * (a.N / 4);
        hipLaunchKernelGGL((whenet_pw_check_kernel<T, GATE, RES, ACT>), dim3(unsigned((work + 255) / 256)), dim3(256),
                           0, stream, static_cast<const T*>(a.a), a.wdense, a.bias, a.gate,
                           static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.HW);
        return;
    }
    const int cus = num_cus > 0 ? num_cus : 256;
    // Deep contractions (K >= 320: the project convs of blocks 7-16 and the head conv, all on
    // 14x14 / 7x7 maps) split K across the 4 waves of a workgroup.  The rule depends on the
    // layer only, never on the batch, so a crop's result is bitwise independent of the batch
    // it travels in (and of how a batch is sharded across GPUs).
    if (a.K >= 320) {
        launch_mfma<T, 1, 4, 4, GATE, RES, ACT>(a, ceil_div(a.M, 32), a.NTILES, stream);
        return;
    }
    // NT (32-wide out-channel tiles per wave): as many as keep >= 4 workgroups per CU in
    // flight; with fewer rows than that, favour parallelism (NT = 1).
    const int MT = ceil_div(a.M, 128);
    const int want = 4 * cus;
    if (MT * ceil_div(a.NTILES, 4) >= want) launch_mfma<T, 4, 2, 1, GATE, RES, ACT>(a, MT, ceil_div(a.NTILES, 4), stream);
    else if (MT * ceil_div(a.NTILES, 2) >= want) launch_mfma<T, 2, 4, 1, GATE, RES, ACT>(a, MT, ceil_div(a.NTILES, 2), stream);
    else launch_mfma<T, 1, 4, 1, GATE, RES, ACT>(a, MT, a.NTILES, stream);
}

template <typename T>
void launch_dtype(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    const bool gate = a.gate != nullptr, res = a.res != nullptr;
    // the network uses exactly three flavours: expand/head (swish), project (gate), project+skip
    if (!gate && !res && a.act == ACT_SWISH) launch_variant<T, false, false, ACT_SWISH>(a, impl, num_cus, stream);
    else if (gate && !res && a.act == ACT_NONE) launch_variant<T, true, false, ACT_NONE>(a, impl, num_cus, stream);
    else if (gate && res && a.act == ACT_NONE) launch_variant<T, true, true, ACT_NONE>(a, impl, num_cus, stream);
    else throw Error(WHENET_EINVAL, "pointwise: unsupported epilogue combination");
}

}  // namespace

void launch_pw(const PwArgs& a, int dtype, int impl, int num_cus, hipStream_t stream) {
    WHENET_REQUIRE(a.N % 4 == 0 && a.M > 0, WHENET_EINVAL, "pointwise: bad shape");
    if (dtype == WHENET_F16) launch_dtype<half_t>(a, impl, num_cus, stream);
    else launch_dtype<float>(a, impl, num_cus, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

const char* kernel_name_pw(int dtype, int impl, bool gate, bool res, int act) {
    (void)act;
    if (impl == 1) return dtype == WHENET_F16 ? "whenet_pw_check_kernel<_Float16>" : "whenet_pw_check_kernel<float>";
    if (dtype == WHENET_F16) return gate ? (res ? "whenet_pw_kernel<_Float16,gate,res>" : "whenet_pw_kernel<_Float16,gate>")
                                         : "whenet_pw_kernel<_Float16,swish>";
    return gate ? (res ? "whenet_pw_kernel<float,gate,res>" : "whenet_pw_kernel<float,gate>")
                : "whenet_pw_kernel<float,swish>";
}

}  // namespace whenet
