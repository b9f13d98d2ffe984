// Probe: in-kernel timeline of the fused expand+depthwise "front" kernel on every B0 block shape,
// outside the engine (random data; timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWHENET_STAMPS tools/probes/front_probe.hip \
//         headposeestimation-whenet_amd/csrc/convert.hip -o tools/probes/front_probe
#include "../../headposeestimation-whenet_amd/csrc/front.hip"

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

using namespace whenet;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Shape { const char* name; int k, s, H, Cin, Cexp; };

template <typename T> T* dalloc(size_t n, float scale) {
    std::vector<T> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = T(scale * (float(rand() % 2001) / 1000.f - 1.f));
    T* d; CK(hipMalloc(&d, n * sizeof(T)));
    CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const Shape shapes[] = {{"b2", 3, 2, 112, 16, 96},   {"b3", 3, 1, 56, 24, 144},  {"b4", 5, 2, 56, 24, 144},
                            {"b5", 5, 1, 28, 40, 240},   {"b6", 3, 2, 28, 40, 240},  {"b7", 3, 1, 14, 80, 480},
                            {"b9", 5, 1, 14, 80, 480},   {"b10", 5, 1, 14, 112, 672}, {"b12", 5, 2, 14, 112, 672},
                            {"b13", 5, 1, 7, 192, 1152}, {"b16", 3, 1, 7, 192, 1152}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t NST = 8 * 65536;
    long long* d_st; CK(hipMalloc(&d_st, NST * sizeof(long long)));
    for (const Shape& sh : shapes)
    for (int thr : {256, 512, 1024}) {
        using T = half_t;
        const int Ho = ceil_div(sh.H, sh.s);
        const int padt = std::max((Ho - 1) * sh.s + sh.k - sh.H, 0);
        FrontArgs a{};
        a.k = sh.k; a.s = sh.s; a.H = sh.H; a.Ho = Ho; a.Cin = sh.Cin; a.Cexp = sh.Cexp; a.pad = padt / 2; a.n = n;
        a.KSe = ceil_div(sh.Cin, 16); a.NTe = ceil_div(sh.Cexp, 32);
        a.plan = plan_front(WHENET_F16, sh.k, sh.s, sh.H, Ho, sh.Cexp);
        a.plan.threads = thr;
        a.x = dalloc<T>(size_t(n) * sh.H * sh.H * sh.Cin, 1.f);
        a.wep = dalloc<T>(size_t(a.KSe) * a.NTe * 64 * 8, 0.05f);
        a.be = dalloc<float>(a.NTe * 32, 0.1f);
        a.wd = dalloc<float>(size_t(sh.k) * sh.k * sh.Cexp, 0.1f);
        a.bd = dalloc<float>(sh.Cexp, 0.1f);
        a.out = dalloc<T>(size_t(n) * Ho * Ho * sh.Cexp, 0.f);
        a.rpart = dalloc<float>(size_t(n) * a.plan.ntiles() * sh.Cexp, 0.f);
        a.R = std::max(1, sh.Cin / 4);
        const float* w1_all = dalloc<float>(size_t(a.R) * sh.Cexp, 0.05f);
        a.w1t = sh.Cexp >= 480 ? w1_all : nullptr;       // engine.cpp: SE reduce conv in the front kernel for blocks 7-16

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
        const float t_front = time_loop([&] { launch_front(a, WHENET_F16, s); }, 200);

        CK(hipMemset(d_st, 0, NST * sizeof(long long)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
        launch_front(a, WHENET_F16, s);
        CK(hipStreamSynchronize(s));
        long long* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
        std::vector<long long> st(NST);
        CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        const int nb = std::min<long long>(65536, (long long)a.plan.ntiles() * a.plan.chunks * n);
        long long t0 = st[0], tend = 0;
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, st[b * 8]); tend = std::max(tend, st[b * 8 + 6]); }
        const FrontPlan& p = a.plan;
        const double bytes = double(n) * (double(sh.H) * sh.H * sh.Cin + double(Ho) * Ho * sh.Cexp) * 2.0;
        printf("%-4s thr=%d k%d s%d H%d Cexp%d n=%d: %.2f us (%.0f GB/s alg) | plan CC=%d TH=%d NSX=%d tiles=%dx%d chunks=%d E=%dx%d lds=%zu | %d wgs, span %.2f us\n",
               sh.name, thr, sh.k, sh.s, sh.H, sh.Cexp, n, t_front, bytes / t_front * 1e-3, p.CC, p.TH, p.NSX, p.tiles_x, p.tiles_y,
               p.chunks, p.EH, p.EW, p.lds_bytes, nb, (tend - t0) * 0.01);
        const char* names[] = {"entry", "zero+w", "expand", "sync", "taps", "store", "sums+fc1"};
        printf("   median:");
        for (int i = 1; i <= 6; ++i) {
            std::vector<long long> v;
            for (int b = 0; b < nb; ++b) v.push_back(st[b * 8 + i] - st[b * 8 + i - 1]);
            std::sort(v.begin(), v.end());
            printf(" %s %.2f", names[i], v[v.size() / 2] * 0.01);
        }
        std::vector<long long> starts, total;
        for (int b = 0; b < nb; ++b) { starts.push_back(st[b * 8] - t0); total.push_back(st[b * 8 + 6] - st[b * 8]); }
        std::sort(starts.begin(), starts.end());
        std::sort(total.begin(), total.end());
        printf(" | wg total median %.2f max %.2f | starts: median +%.2f last +%.2f us\n", total[nb / 2] * 0.01,
               total.back() * 0.01, starts[nb / 2] * 0.01, starts.back() * 0.01);
        for (const void* q : {a.x, a.wep, (const void*)a.be, (const void*)a.wd, (const void*)a.bd, (const void*)a.out,
                              (const void*)a.rpart, (const void*)w1_all})
            CK(hipFree(const_cast<void*>(q)));
    }
    return 0;
}
