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
__global__ __launch_bounds__(256) void whenet_se_kernel(const float* __restrict__ partial, int ntiles, float inv_hw,
                                                        const float* __restrict__ w1p, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ gate, int C, int R) {
    __shared__ float s_mean[1152];
    __shared__ float s_red[4][RP];
    __shared__ float s_r[RP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // squeeze: mean over the map = (sum of the depthwise kernel's tile partials) / (H*W)
    const float* pp = partial + size_t(b) * ntiles * C;
    for (int c = tid; c < C; c += 256) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int i = 0;
        for (; i + 4 <= ntiles; i += 4) {
            t0 += pp[size_t(i) * C + c];
            t1 += pp[size_t(i + 1) * C + c];
            t2 += pp[size_t(i + 2) * C + c];
            t3 += pp[size_t(i + 3) * C + c];
        }
        for (; i < ntiles; ++i) t0 += pp[size_t(i) * C + c];
        s_mean[c] = ((t0 + t1) + (t2 + t3)) * inv_hw;
    }
    __syncthreads();

    // reduce: r[j] = swish(b1[j] + sum_c mean[c] * W1[c][j])
    float acc[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) acc[j] = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float m = s_mean[c];
        const float4v* wr = reinterpret_cast<const float4v*>(w1p + size_t(c) * RP);
#pragma unroll
        for (int q = 0; q < RP / 4; ++q) {
            const float4v v = wr[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * q + i] = fmaf(m, v[i], acc[4 * q + i]);
        }
    }
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        float t = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) s_red[wave][j] = t;
    }
    __syncthreads();
    if (tid < R) s_r[tid] = swish_f<true>(((s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid])) + b1[tid]);
    __syncthreads();

    // excite: gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c])
    for (int c = tid; c < C; c += 256) {
        float t0 = b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int j = 0;
        for (; j + 4 <= R; j += 4) {
            t0 = fmaf(s_r[j], w2[size_t(j) * C + c], t0);
            t1 = fmaf(s_r[j + 1], w2[size_t(j + 1) * C + c], t1);
            t2 = fmaf(s_r[j + 2], w2[size_t(j + 2) * C + c], t2);
            t3 = fmaf(s_r[j + 3], w2[size_t(j + 3) * C + c], t3);
        }
        for (; j < R; ++j) t0 = fmaf(s_r[j], w2[size_t(j) * C + c], t0);
        gate[size_t(b) * C + c] = sigmoid_f<true>((t0 + t1) + (t2 + t3));
    }
}

template <int RP>
void launch_rp(const SeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(whenet_se_kernel<RP>, dim3(a.n), dim3(256), 0, stream, a.partial, a.ntiles, a.inv_hw, a.w1p,
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
