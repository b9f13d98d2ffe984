// Probe: the two squeeze-excite kernels on the real layer shapes (random data; timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWHENET_STAMPS tools/probes/se_probe.hip \
//         headposeestimation-whenet_amd/csrc/convert.hip -o tools/probes/se_probe
#include "../../headposeestimation-whenet_amd/csrc/se.hip"

#include <cstdlib>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Shape { const char* name; int HW, C, R, ntiles, np; };
float* dalloc(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * (float(rand() % 2001) / 1000.f - 1.f);
    float* d; CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const Shape shapes[] = {{"b1", 12544, 32, 8, 56, 56},  {"b2", 3136, 96, 4, 64, 64},   {"b3", 3136, 144, 6, 14, 70},
                            {"b4", 784, 144, 6, 6, 30},    {"b5", 784, 240, 10, 4, 32},   {"b6", 196, 240, 10, 2, 8},
                            {"b7", 196, 480, 20, 1, 15},   {"b10", 196, 672, 28, 1, 21},  {"b12", 49, 672, 28, 1, 11},
                            {"b13", 49, 1152, 48, 1, 9}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long* d_st; CK(hipMalloc(&d_st, 8 * 4096 * sizeof(long long)));
    for (const Shape& sh : shapes) {
        const int RP = se_padded_r(sh.R);
        float* partial = dalloc(size_t(n) * sh.ntiles * sh.C, 1.f);
        float* rpart = dalloc(size_t(n) * sh.np * RP, 1.f);
        float* w1t = dalloc(size_t(sh.R) * sh.C, 0.05f);
        float* b1 = dalloc(RP, 0.1f);
        float* w2c = dalloc(size_t(sh.C) * RP, 0.05f);
        float* b2 = dalloc(sh.C, 0.1f);
        float* gate = dalloc(size_t(n) * sh.C, 0.f);
        SeArgs sa{};
        sa.partial = partial; sa.ntiles = sh.ntiles; sa.inv_hw = 1.f / sh.HW; sa.w1t = w1t; sa.b1 = b1; sa.w2c = w2c; sa.b2 = b2;
        sa.gate = gate; sa.C = sh.C; sa.R = sh.R; sa.n = n;
        SeExciteArgs xa{};
        xa.rpart = rpart; xa.np = sh.np; xa.inv_hw = sa.inv_hw; xa.b1 = b1; xa.w2c = w2c; xa.b2 = b2; xa.gate = gate; xa.C = sh.C;
        xa.R = sh.R; xa.n = n;
        auto time_loop = [&](auto&& fn, int iters) {
            for (int i = 0; i < 5; ++i) fn();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };
        const float t_full = time_loop([&] { launch_se(sa, s); }, 200);
        const float t_ex = time_loop([&] { launch_se_excite(xa, s); }, 200);
        const float t_empty = time_loop([&] { launch_empty(s); }, 200);
        CK(hipMemset(d_st, 0, 8 * 4096 * sizeof(long long)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
        launch_se_excite(xa, s);
        CK(hipStreamSynchronize(s));
        long long* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
        long long ss[8];
        CK(hipMemcpy(ss, d_st, sizeof(ss), hipMemcpyDeviceToHost));
        printf("%-4s n=%d C=%d R=%d: se (squeeze+reduce+excite) %.2f us | excite-only %.2f us (split %d) | empty %.2f | excite wg0: loads %.2f barrier %.2f gate %.2f\n",
               sh.name, n, sh.C, sh.R, t_full, t_ex, se_excite_split(sh.C), t_empty, (ss[1] - ss[0]) * 0.01,
               (ss[2] - ss[1]) * 0.01, (ss[3] - ss[2]) * 0.01);
        for (float* p : {partial, rpart, w1t, b1, w2c, b2, gate}) CK(hipFree(p));
    }
    return 0;
}
