// Squeeze-excite gate for ONE crop, executed by one whole workgroup of NTHR lanes (a multiple
// of 64): the body of se.hip as a device function, so that the kernel that produced the channel
// sums can finish the block itself (front.hip: the last workgroup of a crop to arrive runs this).
//   pp      [ntiles][C] tile partial sums of the crop        s_mean  LDS [>= C] floats
//   gate_b  [C] output                                        s_r     LDS [>= RP] floats
// Same arithmetic and summation order as whenet_se_kernel<RP>.
#pragma once

#include "device_math.h"

namespace whenet {

template <int RP, int NTHR>
__device__ __forceinline__ void se_gate_device(const float* __restrict__ pp, int ntiles, float inv_hw,
                                               const float* __restrict__ w1t, const float* __restrict__ b1,
                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                               float* __restrict__ gate_b, int C, int R, float* s_mean, float* s_r,
                                               int tid) {
    constexpr int NW = NTHR / 64;
    constexpr int CPL = 1152 / 64;          // channel slots per lane (18)
    constexpr int JB = 3;                   // outputs per wave per round (3 x 18 loads in flight)
    const int lane = tid & 63, wave = tid >> 6;

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

    for (int j0 = wave * JB; j0 < RP; j0 += NW * JB) {
        float wv[JB][CPL];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
            const float* wrow = w1t + size_t(j < R ? j : 0) * C;
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int c = lane + 64 * u;
                wv[jj][u] = (j < R && c < C) ? wrow[c] : 0.f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int j = j0 + jj;
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
        gate_b[c] = sigmoid_f<true>((t0 + t1) + (t2 + t3));
    }
}

}  // namespace whenet
