// Does v_mfma_f32_32x32x16_f16 honour binary16 SUBNORMAL inputs, or flush them to zero?
// (decides whether the lo half of a hi/lo split needs a scale: round 5, f32s path)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits, unsigned short bbits) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = __builtin_bit_cast(_Float16, bbits); }
    float16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    struct { unsigned short a, b; const char* what; double want; } cases[] = {
        {0x0001, 0x3c00, "a = 2^-24 (smallest subnormal), b = 1", 16 * 5.9604644775390625e-8},
        {0x0200, 0x3c00, "a = 2^-15 (subnormal), b = 1", 16 * 3.0517578125e-5},
        {0x0200, 0x0200, "a = b = 2^-15 (both subnormal)", 16 * 9.313225746154785e-10},
        {0x0400, 0x3c00, "a = 2^-14 (smallest normal), b = 1", 16 * 6.103515625e-5},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
        float h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%-45s -> %.9g (exact %.9g) %s\n", c.what, h, c.want, h == float(c.want) ? "HONOURED" : (h == 0 ? "FLUSHED" : "OTHER"));
    }
    return 0;
}
