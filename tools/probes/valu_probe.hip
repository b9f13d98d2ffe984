// Probe: issue cost (cycles per wave64 instruction per SIMD) of the VALU / LDS instructions the depthwise
// taps and Swish epilogues are made of, at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_probe.hip -o tools/probes/valu_probe && ./tools/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 2048, CH = 16;   // 16 independent chains per lane

template <int OP>
__global__ void k(long long* out, float seed) {
    float a[CH];
    float2v p[CH];
    half2v h[CH];
    for (int i = 0; i < CH; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = float2v{a[i], a[i] + 1}; h[i] = half2v{(_Float16)a[i], (_Float16)(a[i] * 0.5f)}; }
    const float w = seed * 0.999f;
    const half2v hw = half2v{(_Float16)0.5f, (_Float16)0.25f};
    const float2v pw = float2v{w, w};
    __syncthreads();
    long long c0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pw));
            if (OP == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(h[i]), "v"(hw));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h[i]));
            if (OP == 6) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h[i]) : "v"(hw));
            if (OP == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(h[i]) : "v"(hw));
            if (OP == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
            if (OP == 9) asm volatile("v_rcp_f16 %0, %0" : "+v"(h[i]));
            if (OP == 10) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(h[i]), "v"(hw));
            if (OP == 11) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(w));
            if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w));
        }
    }
    long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < CH; ++i) s += a[i] + p[i][0] + p[i][1] + float(h[i][0]) + float(h[i][1]);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = (long long)s; }
}

template <int OP> int run(const char* name, long long* d) {
    for (int waves : {4, 8, 16}) {       // per workgroup of one CU: 1, 2, 4 waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 1.0001f);
        CK(hipDeviceSynchronize());
        long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        const double per = double(h[0]) / (double(ITERS) * CH);
        printf("%-18s %2d waves/CU: %6.2f cycles per instruction per wave, %5.2f per SIMD-slot\n", name, waves, per, per / (waves / 4));
    }
    return 0;
}

int main() {
    long long* d; CK(hipMalloc(&d, 64));
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_dot2_f32_f16", d); run<10>("v_dot2c_f32_f16", d);
    run<3>("v_exp_f32", d); run<4>("v_rcp_f32", d); run<8>("v_exp_f16", d); run<9>("v_rcp_f16", d);
    run<5>("v_cvt_f32_f16", d); run<11>("v_cvt_pk_f16_f32", d); run<6>("v_pk_mul_f16", d); run<7>("v_pk_fma_f16", d);
    run<12>("v_mul_f32", d); run<13>("v_cndmask_b32", d);
    return 0;
}
