// SEBlock (squeeze-excite) gate, one workgroup per crop.
//
// Reference: efficientnet 0.0.4 SEBlock as instantiated by /root/reference/whenet.py:8
// (SURVEY.md Appendix B): mean over H,W (keepdims) -> Conv2D(reduced, 1x1, bias) -> Swish ->
// Conv2D(C, 1x1, bias) -> sigmoid; the gate multiplies the depthwise output (that multiply is
// fused into the project GEMM's operand load, pw.hip).  `reduced` = int(0.25 * block INPUT
// filters).  All f32, fixed summation order (tile partials in tile order) -> reproducible.
// Work: 2*C*R MACs per crop (0.2 % of the network); latency-bound.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

__global__ __launch_bounds__(256) void whenet_se_kernel(const float* __restrict__ partial, int ntiles, float inv_hw,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ gate, int C, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_mean = reinterpret_cast<float*>(smem);        // [C]
    float* s_r = s_mean + C;                                // [R]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // squeeze: mean over the map = (sum of the depthwise kernel's tile partials) / (H*W)
    const float* pp = partial + size_t(b) * ntiles * C;
    for (int c = tid; c < C; c += 256) {
        float t = 0.0f;
        for (int i = 0; i < ntiles; ++i) t += pp[size_t(i) * C + c];
        s_mean[c] = t * inv_hw;
    }
    __syncthreads();

    // reduce: r[j] = swish(b1[j] + sum_c mean[c] * W1[c][j]); one wave per j, shuffle tree
    for (int j = wave; j < R; j += 4) {
        const float* wrow = w1t + size_t(j) * C;
        float t = 0.0f;
        for (int c = lane; c < C; c += 64) t = fmaf(s_mean[c], wrow[c], t);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) s_r[j] = swish_f<true>(t + b1[j]);
    }
    __syncthreads();

    // excite: gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c])
    for (int c = tid; c < C; c += 256) {
        float t = b2[c];
        for (int j = 0; j < R; ++j) t = fmaf(s_r[j], w2[size_t(j) * C + c], t);
        gate[size_t(b) * C + c] = sigmoid_f<true>(t);
    }
}

}  // namespace

void launch_se(const SeArgs& a, hipStream_t stream) {
    const size_t lds = size_t(a.C + a.R) * sizeof(float);
    hipLaunchKernelGGL(whenet_se_kernel, dim3(a.n), dim3(256), lds, stream, a.partial, a.ntiles, a.inv_hw, a.w1t,
                       a.b1, a.w2, a.b2, a.gate, a.C, a.R);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
