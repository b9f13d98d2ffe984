// Device-side helpers shared by the kernels.
#pragma once

#include "common.h"

namespace whenet {

// sigmoid / swish in f32.  PRECISE = true for the f32 parity configuration (libm expf and an
// IEEE division); false for f16 activations, whose 2^-11 rounding dwarfs the fast forms' error.
template <bool PRECISE>
__device__ __forceinline__ float sigmoid_f(float x) {
    if constexpr (PRECISE) {
        return 1.0f / (1.0f + expf(-x));
    } else {
        return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    }
}

template <bool PRECISE>
__device__ __forceinline__ float swish_f(float x) {
    return x * sigmoid_f<PRECISE>(x);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global
// loads and stores (s_waitcnt vmcnt(0)), which would stall prefetched operands and output stores
// at every phase boundary; the kernels that use this barrier exchange data through LDS only.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T> struct IsF32 { static constexpr bool value = false; };
template <> struct IsF32<float> { static constexpr bool value = true; };

// 16-byte vector <-> float lanes
template <typename T>
__device__ __forceinline__ void vec_to_float(const typename Vec<T>::type& v, float (&f)[Vec<T>::V]) {
#pragma unroll
    for (int i = 0; i < Vec<T>::V; ++i) f[i] = float(v[i]);
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::type float_to_vec(const float (&f)[Vec<T>::V]) {
    typename Vec<T>::type v;
#pragma unroll
    for (int i = 0; i < Vec<T>::V; ++i) v[i] = T(f[i]);
    return v;
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::type vec_zero() {
    typename Vec<T>::type v;
#pragma unroll
    for (int i = 0; i < Vec<T>::V; ++i) v[i] = T(0);
    return v;
}

// One k-step of the transposed 1x1-conv product on the matrix cores (see pw.hip):
//   f16: one v_mfma_f32_32x32x16_f16 (16 k);  f32: four v_mfma_f32_32x32x2_f32 (8 k, exact f32).
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

}  // namespace whenet
