// Probe: what does a tiny dependent kernel cost on this box, and at what shader clock does it run?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/clock_probe.hip -o gpurun_out/clock_probe && ./gpurun_out/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

// dependent chain of `n` L2/HBM loads (pointer chase) by one lane; reports shader cycles and wall ticks
__global__ void chase_kernel(const int* next, int n, long long* out) {
    long long c0 = clock64(), w0 = wall_clock64();
    int i = 0;
    for (int k = 0; k < n; ++k) i = next[i];
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = c1 - c0; out[1] = w1 - w0; out[2] = i;
}

// pure ALU spin for `n` iterations
__global__ void spin_kernel(int n, long long* out) {
    long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int k = 0; k < n; ++k) x = x * 1.0001f + 0.5f;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* d; CK(hipMalloc(&d, 64 << 20));
    long long* o; CK(hipMalloc(&o, 64));
    // pointer-chase ring with a 4 KiB stride over 32 MiB (beyond L2)
    { std::vector<int> h((64 << 20) / 4, 0); int stride = 1024, n = (int)h.size() / stride;
      for (int i = 0; i < n; ++i) h[(size_t)i * stride] = ((i + 1) % n) * stride;
      CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("wall clock rate %d kHz, max shader clock %d kHz\n", wall_khz, clk_khz);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };

    for (int rep = 0; rep < 2; ++rep) {
        // (a) chain of 66 empty kernels, eager and as a graph
        for (int w = 0; w < 3; ++w) { for (int i = 0; i < 66; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, s, nullptr); CK(hipStreamSynchronize(s)); }
        auto t0 = now();
        for (int it = 0; it < 20; ++it) { for (int i = 0; i < 66; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, s, nullptr); CK(hipStreamSynchronize(s)); }
        auto t1 = now();
        printf("eager: 66 empty kernels + sync: %.1f us  (%.2f us/kernel)\n", us(t0, t1) / 20, us(t0, t1) / 20 / 66);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 66; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, s, nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t0 = now();
        for (int it = 0; it < 50; ++it) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t1 = now();
        printf("graph: 66 empty kernels + sync: %.1f us  (%.2f us/kernel)\n", us(t0, t1) / 50, us(t0, t1) / 50 / 66);
        // (b) clocks inside a short and a long kernel
        long long h[3];
        for (int n : {64, 1024, 16384}) {
            hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(64), 0, s, d, n, o); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, o, 24, hipMemcpyDeviceToHost));
            double wall_us = h[1] / (wall_khz / 1000.0);
            printf("chase n=%5d: %lld shader cycles, %.1f us wall -> %.0f MHz, %.0f ns/load\n", n, h[0], wall_us, h[0] / wall_us, wall_us * 1000 / n);
        }
        for (int n : {1000, 100000, 3000000}) {
            for (int blocks : {1, 1024}) {
                hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, n, o); CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h, o, 24, hipMemcpyDeviceToHost));
                double wall_us = h[1] / (wall_khz / 1000.0);
                printf("spin n=%7d blocks=%4d: %lld cycles, %.1f us wall -> %.0f MHz\n", n, blocks, h[0], wall_us, h[0] / wall_us);
            }
        }
        // (c) event-pair timing of one empty kernel
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float acc = 0; 
        for (int i = 0; i < 50; ++i) { CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, s, nullptr); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); acc += ms; }
        printf("event pair around one empty kernel: %.2f us\n", acc / 50 * 1000);
    }
    return 0;
}
