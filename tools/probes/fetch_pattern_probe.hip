// What does a CU pull from L2 as a function of HOW a wave's 16-byte loads are laid out?  (round 5: is the 7x7 / 14x14 split-K GEMM's
// ~45 GB/s per CU a bandwidth limit, or the price of fetching 32-byte pieces of 32 different rows per instruction?)
//   pattern 0: the split-K kernel's activation fragment: lane (j = lane & 31, g = lane >> 5) reads 16 B at row j, byte ks*32 + g*16
//              -> 32 cache lines per 1 KB wave-instruction; wave p takes k-steps p, p+4, ...
//   pattern 1: coalesced: lane (r = lane >> 3, q = lane & 7) reads 16 B at row r, byte kg*128 + q*16 -> 8 lines per 1 KB; wave p takes
//              128-byte k-groups p, p+4, ...; 8 instructions cover the workgroup's 64 rows
//   pattern 2: fully linear 1 KB per instruction (the packed weight image's pattern)
// Same bytes per workgroup (64 rows x K x 2 B), 4 waves, DEPTH loads in flight per wave, WGS workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT, int DEPTH>
__global__ __launch_bounds__(256) void k(const char* __restrict__ A, float* __restrict__ out, int K2 /* row bytes */, int rows_total) {
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 64) % rows_total;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    const int nline = K2 / 128;                       // 128-byte groups per row
    if (PAT == 0) {
        const int j = lane & 31, g = lane >> 5;
        for (int ks = p; ks < nline * 4; ks += 4 * DEPTH) {
            float4v v[DEPTH][2];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    int kk = ks + 4 * d; kk = kk < nline * 4 ? kk : nline * 4 - 1;
                    v[d][mb] = *reinterpret_cast<const float4v*>(A + size_t(row0 + mb * 32 + j) * K2 + kk * 32 + g * 16);
                }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d][0] + v[d][1];
        }
    } else if (PAT == 1) {
        const int r = lane >> 3, q = lane & 7;
        for (int kg = p; kg < nline; kg += 4 * DEPTH) {          // DEPTH groups of 8 loads in flight per wave
            float4v v[DEPTH][8];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                int kk = kg + 4 * d; kk = kk < nline ? kk : nline - 1;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[d][i] = *reinterpret_cast<const float4v*>(A + size_t(row0 + i * 8 + r) * K2 + kk * 128 + q * 16);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[d][i];
        }
    } else {
        const char* base = A + size_t(row0) * K2;      // 64 rows x K2 bytes as one linear block
        const int nchunk = 64 * K2 / 1024;
        for (int c = p; c < nchunk; c += 4 * 8) {
            float4v v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { int cc = c + 4 * i; cc = cc < nchunk ? cc : nchunk - 1; v[i] = *reinterpret_cast<const float4v*>(base + size_t(cc) * 1024 + lane * 16); }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

template <int PAT, int DEPTH>
double run(const char* A, float* out, int K2, int rows, int wgs) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PAT, DEPTH>), dim3(wgs), dim3(256), 0, 0, A, out, K2, rows);
    CK(hipEventRecord(e0));
    const int it = 50;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<PAT, DEPTH>), dim3(wgs), dim3(256), 0, 0, A, out, K2, rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / it;
}

int main() {
    const int K = 1152, K2 = K * 2, rows = 64 * 49;               // b13-15 project at 64 crops: 3136 rows, 7.2 MB
    char* A; float* out;
    CK(hipMalloc(&A, size_t(rows + 64) * K2)); CK(hipMemset(A, 0, size_t(rows + 64) * K2)); CK(hipMalloc(&out, 4096 * 4));
    for (int wgs : {49, 147, 256, 512, 1024}) {
        const double bytes = double(wgs) * 64 * K2;
        const double t0 = run<0, 2>(A, out, K2, rows, wgs), t0b = run<0, 4>(A, out, K2, rows, wgs), t1 = run<1, 1>(A, out, K2, rows, wgs), t2 = run<2, 1>(A, out, K2, rows, wgs);
        const double t1b = run<1, 2>(A, out, K2, rows, wgs), t1c = run<1, 4>(A, out, K2, rows, wgs);
        const int cus = wgs < 256 ? wgs : 256;
        printf("WGs %4d (147 KB each): fragment pattern %6.2f us (%5.1f GB/s per busy CU), 4 groups deep %6.2f us (%5.1f), coalesced rows %6.2f us (%5.1f), linear %6.2f us (%5.1f)"
               " | coalesced, 16 / 32 loads in flight per wave: %6.2f us (%5.1f) / %6.2f us (%5.1f)\n",
               wgs, t0, bytes / t0 / 1e3 / cus, t0b, bytes / t0b / 1e3 / cus, t1, bytes / t1 / 1e3 / cus, t2, bytes / t2 / 1e3 / cus,
               t1b, bytes / t1b / 1e3 / cus, t1c, bytes / t1c / 1e3 / cus);
    }
    return 0;
}
