// Probe: the 16-block v_mfma_f32_4x4x4_16B_f16 (block = one depthwise channel) -- operand / accumulator
// lane mapping checked against a host product, and its issue rate at 1, 2 and 4 waves per SIMD, alone and
// with one ds_read_b64 per instruction (the depthwise-taps-on-the-matrix-cores pattern of front.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4_probe.hip -o tools/probes/mfma4_probe && ./tools/probes/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const half4* a, const half4* b, float4v* d) {
    const int l = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a[l], b[l], c, 0, 0, 0);
    d[l] = c;
}

constexpr int ITERS = 1024, NACC = 8;

template <int MODE>
__global__ void rate_kernel(long long* out, float seed) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = _Float16(seed * 1e-3f * float(i & 63));
    __syncthreads();
    float4v acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = float4v{seed, 0.f, 0.f, 0.f};
    half4 a = {_Float16(seed), _Float16(0.5f), _Float16(0.25f), _Float16(0.125f)};
    half4 b[NACC];
    for (int i = 0; i < NACC; ++i) b[i] = half4{_Float16(seed + i), _Float16(1.f), _Float16(0.5f), _Float16(0.f)};
    const int lane = threadIdx.x & 63;
    const _Float16* base = lds + (lane * 4);            // 8 bytes per lane, conflict-free
    long long c0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) b[i] = *reinterpret_cast<const half4*>(base + ((it * NACC + i) & 31) * 256);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b[i], acc[i], 0, 0, 0);
        if (MODE == 2) {      // dependent chain: every instruction on ONE accumulator
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b[i], acc[0], 0, 0, 0);
        }
    }
    long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (lane == 0 && blockIdx.x == 0) { out[2 + (threadIdx.x >> 6)] = c1 - c0; out[1] = (long long)s; }
}

template <int MODE> int run(const char* name, long long* d) {
    for (int waves : {4, 8, 16}) {
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(waves * 64), 0, 0, d, 1.0001f);
        CK(hipDeviceSynchronize());
        long long h[2 + 16]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[2 + w] > mx ? h[2 + w] : mx;
        const double n = double(ITERS) * NACC * (MODE == 2 ? 2 : 1);
        printf("%-34s %2d waves/CU: oldest wave %6.2f cycles per MFMA; all done after %6.2f -> %5.2f cycles per MFMA per SIMD\n",
               name, waves, double(h[2]) / n, double(mx) / n, double(mx) / n / (waves / 4));
    }
    return 0;
}

int main() {
    // ---- layout -------------------------------------------------------------------------------------------
    std::vector<_Float16> A(16 * 4 * 4), B(16 * 4 * 4);            // [blk][i][k], [blk][k][j]
    srand(1);
    for (auto& v : A) v = _Float16(float(rand() % 17 - 8) / 8.f);
    for (auto& v : B) v = _Float16(float(rand() % 15 - 7) / 4.f);
    std::vector<float> D(16 * 16, 0.f);                            // [blk][i][j]
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0;
                for (int k = 0; k < 4; ++k) s += float(A[(blk * 4 + i) * 4 + k]) * float(B[(blk * 4 + k) * 4 + j]);
                D[(blk * 4 + i) * 4 + j] = s;
            }
    // hypothesis: lane l -> blk = l>>2; a[k] = A[blk][i = l&3][k]; b[k] = B[blk][k][j = l&3]; d[r] = D[blk][i = r][j = l&3]
    std::vector<half4> ha(64), hb(64);
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            ha[l][k] = A[((l >> 2) * 4 + (l & 3)) * 4 + k];
            hb[l][k] = B[((l >> 2) * 4 + k) * 4 + (l & 3)];
        }
    half4 *da, *db; float4v* dd;
    CK(hipMalloc(&da, 64 * sizeof(half4))); CK(hipMalloc(&db, 64 * sizeof(half4))); CK(hipMalloc(&dd, 64 * sizeof(float4v)));
    CK(hipMemcpy(da, ha.data(), 64 * sizeof(half4), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 64 * sizeof(half4), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipDeviceSynchronize());
    std::vector<float4v> hd(64);
    CK(hipMemcpy(hd.data(), dd, 64 * sizeof(float4v), hipMemcpyDeviceToHost));
    int bad_h1 = 0, bad_h2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2;
            if (std::fabs(hd[l][r] - D[(blk * 4 + r) * 4 + (l & 3)]) > 1e-3f) ++bad_h1;      // d[r] = D[blk][r][l&3]
            if (std::fabs(hd[l][r] - D[(blk * 4 + (l & 3)) * 4 + r]) > 1e-3f) ++bad_h2;      // d[r] = D[blk][l&3][r]
        }
    printf("layout: A row-per-lane (i = l&3), B column-per-lane (j = l&3), block = l>>2:\n");
    printf("  D[blk][i = reg][j = l&3]: %d mismatches of 256\n  D[blk][i = l&3][j = reg]: %d mismatches of 256\n", bad_h1, bad_h2);
    if (bad_h1 && bad_h2) {
        printf("  neither: dumping lane 0..7 results and the expected block 0/1 products\n");
        for (int l = 0; l < 8; ++l) printf("   lane %d: %g %g %g %g\n", l, hd[l][0], hd[l][1], hd[l][2], hd[l][3]);
        for (int blk = 0; blk < 2; ++blk)
            for (int i = 0; i < 4; ++i)
                printf("   D[%d][%d][:] = %g %g %g %g\n", blk, i, D[(blk * 4 + i) * 4], D[(blk * 4 + i) * 4 + 1], D[(blk * 4 + i) * 4 + 2], D[(blk * 4 + i) * 4 + 3]);
    }
    // ---- rate ---------------------------------------------------------------------------------------------
    long long* d; CK(hipMalloc(&d, 256));
    run<0>("mfma_4x4x4_16B_f16 (8 accumulators)", d);
    run<1>("  + one ds_read_b64 per MFMA", d);
    run<2>("  half of them on ONE accumulator", d);
    return 0;
}
