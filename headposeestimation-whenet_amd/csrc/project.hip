// MBConv back half as ONE launch: squeeze-excite gate + gated 1x1 project conv + BN (+ skip).
//
// Reference: efficientnet 0.0.4 MBConv block as instantiated by /root/reference/whenet.py:8
// (SURVEY.md Appendix B): SEBlock (mean over H,W -> Conv1x1+bias -> Swish -> Conv1x1+bias ->
// sigmoid -> multiply) followed by the project Conv1x1 + BN and, for stride-1 blocks with equal
// in/out filters, the identity skip.
//
// Why one kernel: at the batch sizes this path serves every launch of the layer schedule is a
// latency chain of ~10 us plus a ~5 us launch boundary, and the SE gate (2*C*R MACs per crop) was
// a launch of its own in front of each of the 16 project GEMMs.  Here every workgroup of the
// project GEMM first computes the gate of ITS crop into LDS (1024 lanes, the same arithmetic and
// summation order as se.hip, so the results are bitwise those of the two-launch schedule) and
// then multiplies.  The gate is recomputed by each workgroup of a crop (1..25 of them): that costs
// L2 reads of the tile partial sums and of the two tiny SE kernels, never HBM, and removes 16 of
// the 51 launches of a forward.  All weight loads of the SE phase are independent of its data, so
// they are issued before the partial sums arrive: the phase is ONE dependent memory round trip.
//
// GEMM mapping (same MFMA transposed product and fragment image as pw.hip): rows are tiled PER
// CROP in 32-row strips (a workgroup never straddles two crops, so one gate vector serves it);
//   K <  320  16 waves = 16 strips, each wave owns all NT out-channel tiles of its strip;
//   K >= 320  16 waves = 4 teams x 4 waves; a team owns one (strip, NT-tile chunk) and its 4 waves
//             split the k-steps (interleaved) and combine through LDS in wave order 0+1+2+3 --
//             the summation order of whenet_pw_kernel<T,1,8,4,..>, which this kernel replaces.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

namespace whenet {

namespace {

template <typename T, int NT, bool SPLIT, bool RES, int RP>
__global__ __launch_bounds__(1024) void whenet_project_kernel(
    const T* __restrict__ D, const T* __restrict__ Wp, const float* __restrict__ bias,
    const float* __restrict__ partial, int ntiles, float inv_hw, const float* __restrict__ w1t,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ gate_out, const T* __restrict__ res, T* __restrict__ out, int HW, int K, int N, int KS,
    int NTILES, int R, int SPC, int NCH, int WPC) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int NTHR = 1024, NW = 16;
    constexpr int U = (NT == 1) ? 8 : (NT == 2) ? 4 : (SPLIT ? 2 : 3);    // k-steps whose loads are issued together
    constexpr int SK = SPLIT ? 4 : 1;

    __shared__ float s_mean[1152];
    __shared__ float s_gate[1152];
    __shared__ float s_r[RP];
    __shared__ float s_red[SPLIT ? 4 * 3 * 16 * 64 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int crop = blockIdx.x / WPC, wg = blockIdx.x % WPC;
    const int C = K;
    STAMP(0);

    // ---------------- phase A: the crop's SE gate -> s_gate (arithmetic of se.hip) -------------
    {
        constexpr int JPW = (RP + NW - 1) / NW;         // fc1 outputs per wave (<= 3)
        constexpr int CPL = 1152 / 64;                   // channel slots per lane (18)
        // fc1 weights: independent of the data, issued first
        float wv1[JPW][CPL];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + NW * jj;
            const float* wrow = w1t + size_t(j < R ? j : 0) * C;
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int c = lane + 64 * u;
                wv1[jj][u] = (j < R && c < C) ? wrow[c] : 0.f;
            }
        }
        // squeeze
        const float* pp = partial + size_t(crop) * ntiles * C;
        for (int c = tid; c < C; c += NTHR) {
            float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < ntiles; i += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i + u < ntiles) t[u] += pp[size_t(i + u) * C + c];
            }
            s_mean[c] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * inv_hw;
        }
        // fc2 weights of this lane's first channel: issued before the fc1 arithmetic
        STAMP(1);
        constexpr bool PRE2 = RP <= 28;                  // (RP = 48: 54 + 48 live registers would spill)
        float wv2[RP];
        if constexpr (PRE2) {
            const int c = tid < C ? tid : 0;
#pragma unroll
            for (int j = 0; j < RP; ++j) wv2[j] = (j < R) ? w2[size_t(j) * C + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + NW * jj;
            float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int c = lane + 64 * u;
                p[u & 3] = fmaf((c < C) ? s_mean[c] : 0.f, wv1[jj][u], p[u & 3]);
            }
            float t = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
            if (lane == 0 && j < RP) s_r[j] = (j < R) ? swish_f<true>(t + b1[j]) : 0.f;
        }
        __syncthreads();
        STAMP(2);
        for (int c = tid; c < C; c += NTHR) {
            if (!PRE2 || c != tid) {
#pragma unroll
                for (int j = 0; j < RP; ++j) wv2[j] = (j < R) ? w2[size_t(j) * C + c] : 0.f;
            }
            float t0 = b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int j = 0; j < RP; j += 4) {
                t0 = fmaf(s_r[j], wv2[j], t0);
                t1 = fmaf(s_r[j + 1], wv2[j + 1], t1);
                t2 = fmaf(s_r[j + 2], wv2[j + 2], t2);
                t3 = fmaf(s_r[j + 3], wv2[j + 3], t3);
            }
            const float gv = sigmoid_f<true>((t0 + t1) + (t2 + t3));
            s_gate[c] = gv;
            if (wg == 0 && gate_out) gate_out[size_t(crop) * C + c] = gv;
        }
        __syncthreads();
    }
    STAMP(3);

    // ---------------- phase B: out = (D * gate) @ W + bias (+ res) ----------------------------
    const int team = SPLIT ? (wave >> 2) : wave;
    const int kpart = SPLIT ? (wave & 3) : 0;
    const int unit = wg * (SPLIT ? 4 : 16) + team;
    const int strip = unit / NCH, nch = unit % NCH;
    const bool tvalid = strip < SPC;
    if (!SPLIT && !tvalid) return;
    const int g = lane >> 5;
    const int p = strip * 32 + (lane & 31);
    const bool rvalid = tvalid && p < HW;
    const size_t row = size_t(crop) * HW + (rvalid ? p : 0);
    const int nt0 = nch * NT;

    const T* ap = D + row * K + g * V;
    const float* gp = s_gate + g * V;
    const VT* wp = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64 + lane;

    float16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto load_a = [&](int ks) -> VT {
        VT a = vec_zero<T>();
        if (rvalid && ks * 2 * V + g * V < K) a = *reinterpret_cast<const VT*>(ap + ks * 2 * V);
        return a;
    };
    auto gate_a = [&](VT a, int ks) -> VT {
        if (!(rvalid && ks * 2 * V + g * V < K)) return a;
        float f[V];
        vec_to_float<T>(a, f);
#pragma unroll
        for (int i = 0; i < V; i += 4) {
            const float4v gv = *reinterpret_cast<const float4v*>(gp + ks * 2 * V + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
        }
        return float_to_vec<T>(f);
    };

    if (tvalid) {
        for (int ks = kpart; ks < KS; ks += U * SK) {
            VT a[U];
            VT w[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k1 = ks + u * SK;
                const VT* wk = wp + size_t(k1) * NTILES * 64;
#pragma unroll
                for (int t = 0; t < NT; ++t) w[u][t] = (k1 < KS && nt0 + t < NTILES) ? wk[t * 64] : vec_zero<T>();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = (ks + u * SK < KS) ? load_a(ks + u * SK) : vec_zero<T>();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ks + u * SK < KS) a[u] = gate_a(a[u], ks + u * SK);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) Mfma<T>::step(w[u][t], a[u], acc[t]);
            }
        }
    }

    STAMP(4);
    if constexpr (SPLIT) {
        // combine the 4 k-parts of each team, one 32x32 tile at a time (12 KB of LDS per team)
        float* red = s_red + team * (3 * 16 * 64);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (kpart > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((kpart - 1) * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[(w * 16 + r) * 64 + lane];
            }
            if (t + 1 < NT) __syncthreads();
        }
        if (kpart > 0) return;
    }
    STAMP(5);

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
            for (int r = 0; r < 4; ++r) y[r] = acc[t][4 * qq + r] + bv[r];
            if constexpr (RES) {
                const OT rv = *reinterpret_cast<const OT*>(res + row * N + n0);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
            }
            OT o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
            *reinterpret_cast<OT*>(out + row * N + n0) = o;
        }
    }
    STAMP(6);
}

struct ProjChoice {
    int NT;
    bool split;
    int RP;
};

// Layer-static configuration (never a function of the batch): deep contractions split K, NT is
// the largest of {1,2,3} tiles per wave that divides the work evenly.
ProjChoice choose_project(const ProjectArgs& a) {
    ProjChoice c;
    c.split = a.K >= 320;
    c.NT = a.NTILES <= 3 ? a.NTILES : (a.NTILES % 3 == 0 || a.NTILES > 4) ? 3 : 2;
    c.RP = se_padded_r(a.R);
    return c;
}

template <typename T, int NT, bool SPLIT, bool RES, int RP>
void launch_inst(const ProjectArgs& a, hipStream_t stream) {
    const int SPC = ceil_div(a.HW, 32);
    const int NCH = ceil_div(a.NTILES, NT);
    const int WPC = ceil_div(SPC * NCH, SPLIT ? 4 : 16);
    hipLaunchKernelGGL((whenet_project_kernel<T, NT, SPLIT, RES, RP>), dim3(unsigned(a.n * WPC)), dim3(1024), 0,
                       stream, static_cast<const T*>(a.d), static_cast<const T*>(a.wp), a.bias, a.partial, a.ntiles,
                       a.inv_hw, a.w1t, a.b1, a.w2, a.b2, a.gate, static_cast<const T*>(a.res),
                       static_cast<T*>(a.out), a.HW, a.K, a.N, a.KS, a.NTILES, a.R, SPC, NCH, WPC);
}

// the (NT, SPLIT, RES, RP) combinations EfficientNet-B0's 16 blocks use
#define WHENET_PROJECT_TABLE(X) \
    X(1, false, false, 8)       \
    X(1, false, false, 4)       \
    X(1, false, true, 8)        \
    X(2, false, false, 8)       \
    X(2, false, true, 12)       \
    X(3, false, false, 12)      \
    X(3, true, true, 20)        \
    X(2, true, false, 20)       \
    X(2, true, true, 28)        \
    X(3, true, false, 28)       \
    X(3, true, true, 48)        \
    X(3, true, false, 48)

template <typename T>
void launch_dtype(const ProjectArgs& a, hipStream_t stream) {
    const ProjChoice c = choose_project(a);
    const bool res = a.res != nullptr;
#define X(NT_, SPLIT_, RES_, RP_)                                            \
    if (c.NT == NT_ && c.split == SPLIT_ && res == RES_ && c.RP == RP_) {    \
        launch_inst<T, NT_, SPLIT_, RES_, RP_>(a, stream);                   \
        return;                                                              \
    }
    WHENET_PROJECT_TABLE(X)
#undef X
    throw Error(WHENET_EINVAL, "project: layer shape outside the EfficientNet-B0 table");
}

}  // namespace

void launch_project(const ProjectArgs& a, int dtype, hipStream_t stream) {
    WHENET_REQUIRE(a.N % 4 == 0 && a.n > 0 && a.K <= 1152, WHENET_EINVAL, "project: bad shape");
    if (dtype == WHENET_F16) launch_dtype<half_t>(a, stream);
    else launch_dtype<float>(a, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

std::string kernel_name_project(const ProjectArgs& a, int dtype) {
    const ProjChoice c = choose_project(a);
    return std::string("whenet_project_kernel<") + (dtype == WHENET_F16 ? "_Float16" : "float") + ", " +
           std::to_string(c.NT) + ", " + (c.split ? "true" : "false") + ", " + (a.res ? "true" : "false") + ", " +
           std::to_string(c.RP) + ">";
}

}  // namespace whenet
