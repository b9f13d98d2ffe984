// SEBlock excite half as device functions, for the project GEMMs that compute the gate of their own rows' crops
// in their prologue (pw.hip, round 3) instead of reading it from a separate launch.
//
// Reference: efficientnet 0.0.4 SEBlock (/root/reference/whenet.py:8): the producer
// (front.hip / front2.hip) has already applied se_reduce to its channel sums; what is left is
//     r[j]    = swish(b1[j] + (sum over the crop's np partial vectors)[j] / (H*W))
//     gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c])
// The arithmetic and its ORDER are exactly se.hip's whenet_se_excite_kernel (4 running sums over the partial vectors
// p = u mod 4 combined (t0+t1)+(t2+t3); 4 FMA chains over j = q mod 4 combined the same way; precise expf), so the
// fused gate has the bits of the stand-alone kernel's (tests/test_gpu_parity.py::test_fused_squeeze_excite_*).
#pragma once

#include "device_math.h"
#include "kernels.h"      // struct SeFuse

namespace whenet {

// r[j] of one crop: pp = rpart + (crop * np) * RP + j
__device__ __forceinline__ float se_fused_r(const float* pp, int np, int RP, float inv_hw, float b1j, bool live) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < np; p += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (p + u < np) ? pp[size_t(p + u) * RP] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (p + u < np) t[u & 3] += x[u];
    }
    // (explicitly rounded product and sum: the stand-alone kernel and every fused prologue must agree bit for bit,
    //  whatever the compiler would contract in their different surroundings)
    const float r = __fadd_rn(__fmul_rn((t[0] + t[1]) + (t[2] + t[3]), inv_hw), b1j);
    return live ? swish_f<true>(r) : 0.f;
}

// gate of one channel: w2row = w2c + c * RP (16-byte aligned), s_r = the crop's r[0..RP) in LDS
__device__ __forceinline__ float se_fused_gate(const float* s_r, const float* w2row, float b2c, int RP) {
    float t0 = b2c, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int j = 0; j < RP; j += 4) {
        const float4v w = *reinterpret_cast<const float4v*>(w2row + j);
        t0 = fmaf(s_r[j], w[0], t0);
        t1 = fmaf(s_r[j + 1], w[1], t1);
        t2 = fmaf(s_r[j + 2], w[2], t2);
        t3 = fmaf(s_r[j + 3], w[3], t3);
    }
    return sigmoid_f<true>((t0 + t1) + (t2 + t3));
}

// The gate rows of crops crop_lo .. crop_lo + ncrop - 1 into LDS: s_gate[crop][K] in T, s_r = ncrop * RP floats of
// scratch.  Called by all NTHR lanes of the workgroup; ends with the data visible to every lane (LDS barrier).
template <typename T, int NTHR>
__device__ __forceinline__ void se_fused_to_lds(const SeFuse& se, int crop_lo, int ncrop, int K, T* s_gate, float* s_r) {
    const int tid = threadIdx.x;
    if (tid < ncrop * se.RP) {
        const int c = tid / se.RP, j = tid - c * se.RP;
        s_r[tid] = se_fused_r(se.rpart + (size_t(crop_lo + c) * se.np) * se.RP + j, se.np, se.RP, se.inv_hw,
                              j < se.R ? se.b1[j] : 0.f, j < se.R);
    }
    lds_barrier();
    for (int k = tid; k < K; k += NTHR) {
        const float* w2row = se.w2c + size_t(k) * se.RP;
        const float b2k = se.b2[k];
        for (int c = 0; c < ncrop; ++c) s_gate[c * K + k] = T(se_fused_gate(s_r + c * se.RP, w2row, b2k, se.RP));
    }
    lds_barrier();
}

}  // namespace whenet
