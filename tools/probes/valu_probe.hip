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
    __shared__ float lds[8192];
    lds[threadIdx.x] = seed;
    const unsigned ldsaddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 4;     // conflict-free per wave
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
            if (OP == 14) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(ldsaddr + i * 256));
            if (OP == 15) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(ldsaddr * 2 + i * 512));
            if (OP == 17) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h[i]), "v"(w));
            if (OP == 18) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h[i]), "v"(w));
            if (OP == 19) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,0]" : "+v"(h[i]) : "v"(a[i]), "v"(w), "v"(w));
            if (OP == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 21) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 22) asm volatile("v_exp_f32 %0, -%0" : "+v"(a[i]));
            if (OP == 16) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h[i]));
        }
        if (OP == 14 || OP == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < CH; ++i) s += a[i] + p[i][0] + p[i][1] + float(h[i][0]) + float(h[i][1]);
    // every wave of block 0 reports its own span; the host takes the first wave's (oldest: the per-wave issue
    // rate) and the longest (all waves of the CU done: the throughput)
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 + (threadIdx.x >> 6)] = c1 - c0; out[1] = (long long)s; }
}

template <int OP> int run(const char* name, long long* d) {
    for (int waves : {4, 8, 16}) {       // per workgroup of one CU: 1, 2, 4 waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 1.0001f);
        CK(hipDeviceSynchronize());
        long long h[2 + 16]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[2 + w] > mx ? h[2 + w] : mx;
        const double first = double(h[2]) / (double(ITERS) * CH), all = double(mx) / (double(ITERS) * CH);
        printf("%-18s %2d waves/CU: oldest wave %6.2f cycles per instruction; all waves done after %6.2f -> %5.2f cycles per wave-instruction per SIMD\n",
               name, waves, first, all, all / (waves / 4));
    }
    return 0;
}

int main() {
    long long* d; CK(hipMalloc(&d, 256));
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_dot2_f32_f16", d); run<10>("v_dot2c_f32_f16", d);
    run<3>("v_exp_f32", d); run<4>("v_rcp_f32", d); run<8>("v_exp_f16", d); run<9>("v_rcp_f16", d);
    run<5>("v_cvt_f32_f16", d); run<11>("v_cvt_pk_f16_f32", d); run<6>("v_pk_mul_f16", d); run<7>("v_pk_fma_f16", d);
    run<17>("v_fma_mix_f32 (lo)", d); run<18>("v_fma_mix_f32 (hi)", d); run<19>("v_fma_mixlo_f16", d); run<20>("v_add_f32", d); run<21>("v_add_u32", d); run<22>("v_exp_f32 neg", d); run<12>("v_mul_f32", d); run<16>("v_cvt_f32_f16 sdwa", d); run<14>("ds_read_b32", d); run<15>("ds_read_b64", d);
    return 0;
}
