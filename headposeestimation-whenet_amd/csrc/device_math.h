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

// Swish of the convolution epilogues (7.2 M evaluations per crop).  Round 4: the f32 configuration uses the hardware forms
// here too -- x * v_rcp_f32(1 + v_exp_f32(-x * log2(e))), 5 instructions -- instead of libm's expf and an IEEE division
// (~25 instructions per value: the f32 fused kernels spent most of their VALU time there).  v_exp_f32 / v_rcp_f32 are 1-ulp
// operations and the rounded product x * log2(e) moves the result by <= |x| * 4e-8 relative: a few ulp, below the
// summation noise of the K-deep f32 dot products in front of it (measured: the f32 error against the float64 oracle on the
// 512-crop set is unchanged -- 6e-4 deg max before and after, profiles/r04/bench_f32_b64.json `check`, DESIGN/experiments
// "f32 Swish"; the per-kernel f32 tolerances in tests/test_gpu_parity.py stayed at 2e-5).  The squeeze-excite gates (sigmoid_f<true>, 1152 values per crop) keep the
// precise forms.  -DWHENET_PRECISE_CONV_SWISH=1 restores round 3's arithmetic.
#ifndef WHENET_PRECISE_CONV_SWISH
#define WHENET_PRECISE_CONV_SWISH 0
#endif
template <typename T>
__device__ __forceinline__ float conv_swish(float x) {
    return swish_f<(WHENET_PRECISE_CONV_SWISH != 0) && sizeof(T) == 4>(x);
}

// f32 -> f16 of a value the kernel has just computed, as TWO roundings (the f32 result, then binary16), whatever the optimiser
// would like to fuse.  Left alone it turns half(x * r) into v_fma_mix (one rounding) in some kernels and into v_mul + v_cvt
// in others, depending on the code around it: 1 value in 20 000 then differs by an ulp.  Kernels whose results must agree
// bitwise (stem.hip + dw.hip against stemdw.hip) convert through this; the same goes for `sum += x * r`, which the optimiser
// contracts into an FMA in one kernel and not in the other (opaque_f32() on the product).
__device__ __forceinline__ float opaque_f32(float y) {      // the value as a rounded f32 in a register: nothing fuses across this
    asm volatile("" : "+v"(y));
    return y;
}
__device__ __forceinline__ half_t f32_then_f16(float y) { return half_t(opaque_f32(y)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global
// loads and stores (s_waitcnt vmcnt(0)), which would stall prefetched operands and output stores
// at every phase boundary; the kernels that use this barrier exchange data through LDS only.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS traffic of ONE wave, ordered for its other lanes (DS operations of a wave execute in order; this only keeps the
// compiler from moving them and waits for the data)
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef float float2v __attribute__((ext_vector_type(2)));

// XCD-aware placement of a 1-D launch of units * per workgroups (round 6).  Workgroup L of a launch runs on XCD L % 8 (observed; each XCD
// has its own 4 MB L2).  The `per` members of a unit -- the channel chunks of one (crop, tile) or crop group: they all read the SAME input
// -- are dealt to ONE XCD instead of round-robin over all eight (a 7 x 7 or 14 x 14 layer then pulled its input through eight L2s: block
// 12's front kernel fetched 4.5 x its algorithmic bytes, the head conv 7.8 x), and consecutive units rotate over the XCDs.  A pure
// relabelling of which workgroup does what: results unchanged.  Measured (profiles/r06/ab_xcd_layers.txt, ab_xcd_map_*.txt, 64 crops): the
// front kernels of the 14 x 14 layers get SLOWER when they run alone (b7-b11: 15.5-21.8 -> 18.7-24.7 us: the chunks of a unit then hit the
// same L2 lines of one XCD at the same time) -- and three forwards in flight GAIN 0.8 % against the 3-D grid order of rounds 2-5, batch
// 512 and f32s the same 0.8 % (less traffic on the fabric when every CU is busy); front7 / head7 do not move either way.  `grouped` is chosen per kernel family
// (engine option "xcd_map", default: all three) and applied when the chip is shared: launches of >= 128 crops, or a handle with
// several forwards in flight (Engine::xcd_grouped) -- bit-neutral, so the choice may depend on the batch.
__device__ __forceinline__ void xcd_unit(int L, int units, int per, int& unit, int& member, bool grouped) {
    const int full = units & ~7;                           // units in whole rounds of eight
    if (grouped && L < full * per) {
        const int slot = L >> 3;
        member = slot % per;
        unit = (slot / per) * 8 + (L & 7);
    } else {                                               // members consecutive: a unit's chunks round-robin over the XCDs
        const int r = grouped ? L - full * per : L;
        unit = (grouped ? full : 0) + r / per;
        member = r % per;
    }
}

// Swish of two values with packed f32 arithmetic: v_pk_add_f32 / v_pk_mul_f32 for the three plain steps, the two
// quarter-rate transcendentals per value (v_exp_f32, v_rcp_f32) unpacked: 3.5 instructions per value instead of 5.5.
// Same operations in the same order as swish_f<false> (x * rcp(1 + exp2(-x * log2(e)))): same bits.
__device__ __forceinline__ float2v swish2(float2v x) {
    const float2v t = x * float2v{-1.4426950408889634f, -1.4426950408889634f};
    const float2v d = float2v{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + float2v{1.0f, 1.0f};
    return x * float2v{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

__device__ __forceinline__ float quad_xor1(float v) {      // DPP quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {      // DPP quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// The same for the fused front kernels, whose plain order is the 3-D grid's of rounds 2-5 -- tile fastest, then channel chunk, then crop
// (for the large layers, whose tile counts are multiples of eight, that order already keeps a tile's chunks on one XCD).
__device__ __forceinline__ void xcd_front(int L, int ntiles, int chunks, int n, bool grouped, int& crop, int& tile, int& chunk) {
    if (grouped) {
        int unit;
        xcd_unit(L, ntiles * n, chunks, unit, chunk, true);
        crop = unit / ntiles;
        tile = unit - crop * ntiles;
    } else {
        tile = L % ntiles;
        const int r = L / ntiles;
        chunk = r % chunks;
        crop = r / chunks;
    }
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

// Operand forms of the 1x1-convolution kernels (pw.hip, front.hip).
//   SP = false: operands in the storage type T, one Mfma<T>::step per k-step (f16: 16 k; f32: 8 k on the f32 matrix pipe).
//   SP = true (T = float; WHENET_F32S, round 5): float32 STORAGE, products on the f16 matrix pipe.  A lane's fragment is 8
//     consecutive floats of its pixel row (two 16-byte loads, k-step = 16 as in the f16 form); it is split in registers into
//     binary16 hi = f16(x), lo = f16(x - hi) and multiplied against the host-split weights (snapshot.cpp::pack_pw_split) as
//     lo_w * hi + hi_w * lo + hi_w * hi -- three v_mfma_f32_32x32x16_f16 per 16 k where the exact form issues eight
//     v_mfma_f32_32x32x2_f32 (5.3x less matrix time), f32 accumulation as before; the dropped lo*lo term is 2^-22 of the
//     product.  binary16 subnormal operands are honoured by the matrix cores (tools/probes/mfma_denorm_probe.hip), so the
//     small lo halves of small activations keep their absolute precision (2^-25).
//     RANGE (round-5 advice): hi = f16(x) is +-inf for |x| > 65504 and lo = x - hi then NaN -- the form is exact-grade only for
//     activations inside the binary16 range.  EfficientNet-B0's are O(1..100) behind every BatchNorm (the 512-crop set peaks at ~40);
//     a snapshot that breaks this shows as NaN angles, never as a silently wrong number, and option split_pw = 0 (the exact-f32
//     kernels on the same handle) is the way out.  The weights are scaled per layer on the host and cannot overflow.
template <typename T, bool SP> struct PwOps {
    static constexpr int V = Vec<T>::V;                     // k elements of a lane's fragment
    using VT = typename Vec<T>::type;
    struct A { VT v; };                                     // as loaded
    struct P { VT v; };                                     // as multiplied
    struct W { VT v; };
    static __device__ __forceinline__ A load_a(const T* p) { return A{*reinterpret_cast<const VT*>(p)}; }
    static __device__ __forceinline__ A zero_a() { return A{vec_zero<T>()}; }
    static __device__ __forceinline__ void gate(A& a, const T* g) { a.v = a.v * *reinterpret_cast<const VT*>(g); }
    static __device__ __forceinline__ void gate_by(A& a, const A& g) { a.v = a.v * g.v; }
    static __device__ __forceinline__ P prep(const A& a) { return P{a.v}; }
    // w: the layer's packed image; i: index of the lane's 16-byte fragment in it; lo_off: unused
    static __device__ __forceinline__ W load_w(const T* w, size_t i, size_t) { return W{reinterpret_cast<const VT*>(w)[i]}; }
    static __device__ __forceinline__ W zero_w() { return W{vec_zero<T>()}; }
    static __device__ __forceinline__ void step(const W& w, const P& p, float16v& acc) { Mfma<T>::step(w.v, p.v, acc); }
};
template <> struct PwOps<float, true> {
    static constexpr int V = 8;
    struct A { float4v x0, x1; };
    struct P { half8 hi, lo; };
    struct W { half8 hi, lo; };
    static __device__ __forceinline__ A load_a(const float* p) {
        return A{*reinterpret_cast<const float4v*>(p), *reinterpret_cast<const float4v*>(p + 4)};
    }
    static __device__ __forceinline__ A zero_a() { return A{float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}}; }
    static __device__ __forceinline__ void gate(A& a, const float* g) {
        a.x0 = a.x0 * *reinterpret_cast<const float4v*>(g);
        a.x1 = a.x1 * *reinterpret_cast<const float4v*>(g + 4);
    }
    static __device__ __forceinline__ void gate_by(A& a, const A& g) {
        a.x0 = a.x0 * g.x0;
        a.x1 = a.x1 * g.x1;
    }
    static __device__ __forceinline__ P prep(const A& a) {
        P p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const half_t h0 = half_t(a.x0[e]), h1 = half_t(a.x1[e]);
            p.hi[e] = h0;
            p.hi[4 + e] = h1;
            p.lo[e] = half_t(a.x0[e] - float(h0));
            p.lo[4 + e] = half_t(a.x1[e] - float(h1));
        }
        return p;
    }
    static __device__ __forceinline__ W load_w(const float* w, size_t i, size_t lo_off) {
        const half8* q = reinterpret_cast<const half8*>(w);
        return W{q[i], q[lo_off + i]};
    }
    static __device__ __forceinline__ W zero_w() { return W{half8{0, 0, 0, 0, 0, 0, 0, 0}, half8{0, 0, 0, 0, 0, 0, 0, 0}}; }
    static __device__ __forceinline__ void step(const W& w, const P& p, float16v& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.lo, p.hi, acc, 0, 0, 0);       // (small terms first)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, p.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, p.hi, acc, 0, 0, 0);
    }
};


}  // namespace whenet
