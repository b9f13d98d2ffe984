// SEBlock (squeeze-excite) gate, one workgroup per crop.
//
// Reference: efficientnet 0.0.4 SEBlock as instantiated by /root/reference/whenet.py:8
// (SURVEY.md Appendix B): mean over H,W (keepdims) -> Conv2D(reduced, 1x1, bias) -> Swish ->
// Conv2D(C, 1x1, bias) -> sigmoid; the gate multiplies the depthwise output (that multiply is
// fused into the project GEMM's operand load, pw.hip).  `reduced` = int(0.25 * block INPUT
// filters).  All f32, fixed summation order -> bitwise reproducible.
//
// 2*C*R MACs per crop (0.2 % of the network): the kernel is pure latency, so every phase is
// laid out for memory-level parallelism instead of arithmetic:
//   squeeze  lane <-> channel, tile partials summed with 4 independent loads in flight;
//   reduce   lane <-> channel slice, all R outputs accumulated at once in registers from
//            16-byte loads of the [C][RP] weight rows (RP = R padded to a multiple of 4),
//            then one shuffle tree per output and a 4-wave combine through LDS;
//   excite   lane <-> channel, W2 rows are coalesced across lanes, 4 loads in flight.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

template <int RP>
__global__ __launch_bounds__(1024) void whenet_se_kernel(const float* __restrict__ partial, int ntiles, float inv_hw,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ gate, int C, int R) {
    __shared__ float s_mean[1152];
    constexpr int NW = 16;                      // 1024 lanes: the kernel is a pure latency chain,
    constexpr int NTHR = NW * 64;               // so it is spread as thin as a workgroup allows
    __shared__ float s_r[RP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // squeeze: mean over the map = (sum of the depthwise kernel's tile partials) / (H*W);
    // 8 independent loads in flight per lane
    const float* pp = partial + size_t(b) * ntiles * C;
    for (int c = tid; c < C; c += NTHR) {
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < ntiles; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u < ntiles) t[u] += pp[size_t(i + u) * C + c];
        }
        s_mean[c] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * inv_hw;
    }
    __syncthreads();

    // reduce: r[j] = swish(b1[j] + sum_c mean[c] * W1[c][j]).  Wave w owns outputs j = w, w+16, ..;
    // its lanes stride over c (coalesced 256-byte rows of the transposed kernel), all loads of
    // an output are independent, one 6-step shuffle tree per output.
    constexpr int JPW = (RP + NW - 1) / NW;          // outputs per wave (<= 3)
    constexpr int CPL = 1152 / 64;                    // channel slots per lane (18)
    {
        float wv[JPW][CPL];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + NW * jj;
            const float* wrow = w1t + size_t(j < R ? j : 0) * C;
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int c = lane + 64 * u;
                wv[jj][u] = (j < R && c < C) ? wrow[c] : 0.f;          // all loads in flight at once
            }
        }
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + NW * jj;
            float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int c = lane + 64 * u;
                p[u & 3] = fmaf((c < C) ? s_mean[c] : 0.f, wv[jj][u], p[u & 3]);
            }
            float t = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
            if (lane == 0 && j < RP) s_r[j] = (j < R) ? swish_f<true>(t + b1[j]) : 0.f;
        }
    }
    __syncthreads();

    // excite: gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c]); all R loads of a channel are
    // issued before the first FMA (R <= RP is a compile-time bound -> fully unrolled)
    for (int c = tid; c < C; c += NTHR) {
        float wv[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) wv[j] = (j < R) ? w2[size_t(j) * C + c] : 0.f;
        float t0 = b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int j = 0; j < RP; j += 4) {
            t0 = fmaf(s_r[j], wv[j], t0);
            t1 = fmaf(s_r[j + 1], wv[j + 1], t1);
            t2 = fmaf(s_r[j + 2], wv[j + 2], t2);
            t3 = fmaf(s_r[j + 3], wv[j + 3], t3);
        }
        gate[size_t(b) * C + c] = sigmoid_f<true>((t0 + t1) + (t2 + t3));
    }
}

template <int RP>
void launch_rp(const SeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(whenet_se_kernel<RP>, dim3(a.n), dim3(1024), 0, stream, a.partial, a.ntiles, a.inv_hw, a.w1t,
                       a.b1, a.w2, a.b2, a.gate, a.C, a.R);
}

}  // namespace

int se_padded_r(int R) { return (R + 3) & ~3; }

void launch_se(const SeArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.C <= 1152, WHENET_EINVAL, "squeeze-excite: C > 1152");
    switch (se_padded_r(a.R)) {
        case 4: launch_rp<4>(a, stream); break;
        case 8: launch_rp<8>(a, stream); break;
        case 12: launch_rp<12>(a, stream); break;
        case 20: launch_rp<20>(a, stream); break;
        case 28: launch_rp<28>(a, stream); break;
        case 48: launch_rp<48>(a, stream); break;
        default: throw Error(WHENET_EINVAL, "squeeze-excite: unsupported reduced width");
    }
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
