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

}  // namespace whenet
