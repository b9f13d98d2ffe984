// Probe: in-kernel timeline of the fused SE+project kernel vs. the separate SE and project launches,
// on the real layer shapes, outside the engine (random data; timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWHENET_STAMPS tools/probes/project_probe.hip -o tools/probes/project_probe
#include "../../headposeestimation-whenet_amd/csrc/se.hip"
#include "../../headposeestimation-whenet_amd/csrc/pw.hip"
#include "../../headposeestimation-whenet_amd/csrc/project.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace whenet;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Shape { const char* name; int HW, K, N, R, res, ntiles; };

template <typename T> T* dalloc(size_t n, float scale) {
    std::vector<T> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = T(scale * (float(rand() % 2001) / 1000.f - 1.f));
    T* d; CK(hipMalloc(&d, n * sizeof(T)));
    CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const Shape shapes[] = {
        {"b1", 112 * 112, 32, 16, 8, 0, 56}, {"b2", 56 * 56, 96, 24, 4, 0, 64}, {"b3", 56 * 56, 144, 24, 6, 1, 14}, {"b4", 28 * 28, 144, 40, 6, 0, 8},
        {"b6", 14 * 14, 240, 80, 10, 0, 2},  {"b7", 14 * 14, 480, 80, 20, 1, 2}, {"b10", 14 * 14, 672, 112, 28, 1, 2},
        {"b13", 49, 1152, 192, 48, 1, 1},    {"b16", 49, 1152, 320, 48, 0, 1}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long* d_st; CK(hipMalloc(&d_st, 8 * 4096 * sizeof(long long)));
    for (const Shape& sh : shapes) {
        using T = half_t;
        const int KS = ceil_div(sh.K, 16), NTILES = ceil_div(sh.N, 32);
        T* D = dalloc<T>(size_t(n) * sh.HW * sh.K, 1.f);
        T* Wp = dalloc<T>(size_t(KS) * NTILES * 64 * 8, 0.05f);
        float* bias = dalloc<float>(NTILES * 32, 0.1f);
        float* partial = dalloc<float>(size_t(n) * sh.ntiles * sh.K, 1.f);
        float* w1t = dalloc<float>(size_t(sh.R) * sh.K, 0.05f);
        float* b1 = dalloc<float>(sh.R, 0.1f);
        float* w2 = dalloc<float>(size_t(sh.R) * sh.K, 0.05f);
        float* w2c = dalloc<float>(size_t(sh.R + 4) * sh.K, 0.05f);
        float* b2 = dalloc<float>(sh.K, 0.1f);
        float* gate = dalloc<float>(size_t(n) * sh.K, 0.f);
        T* res = dalloc<T>(size_t(n) * sh.HW * sh.N, 1.f);
        T* out = dalloc<T>(size_t(n) * sh.HW * sh.N, 0.f);

        ProjectArgs pa{};
        pa.d = D; pa.wp = Wp; pa.bias = bias; pa.partial = partial; pa.ntiles = sh.ntiles; pa.inv_hw = 1.f / sh.HW;
        pa.w1t = w1t; pa.b1 = b1; pa.w2 = w2; pa.b2 = b2; pa.gate = gate; pa.res = sh.res ? res : nullptr; pa.out = out;
        pa.n = n; pa.HW = sh.HW; pa.K = sh.K; pa.N = sh.N; pa.KS = KS; pa.NTILES = NTILES; pa.R = sh.R;
        SeArgs sa{};
        sa.partial = partial; sa.ntiles = sh.ntiles; sa.inv_hw = pa.inv_hw; sa.w1t = w1t; sa.b1 = b1; sa.w2c = w2c; sa.b2 = b2;
        sa.gate = gate; sa.C = sh.K; sa.R = sh.R; sa.n = n;
        PwArgs wa{};
        wa.a = D; wa.wp = Wp; wa.bias = bias; wa.gate = gate; wa.res = pa.res; wa.out = out; wa.M = n * sh.HW; wa.K = sh.K;
        wa.N = sh.N; wa.KS = KS; wa.NTILES = NTILES; wa.HW = sh.HW; wa.act = ACT_NONE;

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
        const float t_proj = time_loop([&] { launch_project(pa, WHENET_F16, s); }, 200);
        const float t_se = time_loop([&] { launch_se(sa, s); }, 200);
        const float t_pw = time_loop([&] { launch_pw(wa, WHENET_F16, 0, 256, s); }, 200);
        const float t_both = time_loop([&] { launch_se(sa, s); launch_pw(wa, WHENET_F16, 0, 256, s); }, 200);
        const float t_empty = time_loop([&] { launch_empty(s); }, 200);

        // one stamped launch
        CK(hipMemset(d_st, 0, 8 * 4096 * sizeof(long long)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
        launch_project(pa, WHENET_F16, s);
        CK(hipStreamSynchronize(s));
        long long* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
        std::vector<long long> st(8 * 4096);
        CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        int nb = 0;
        while (nb < 4096 && st[nb * 8] != 0) ++nb;
        long long t0 = st[0], tend = 0;
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, st[b * 8]); tend = std::max(tend, st[b * 8 + 6]); }
        printf("%-4s n=%d  fused %.2f us | se %.2f + pw %.2f (pair %.2f) | empty %.2f   [%d workgroups, span %.2f us]\n", sh.name, n,
               t_proj, t_se, t_pw, t_both, t_empty, nb, (tend - t0) * 0.01);
        const char* names[] = {"entry", "squeeze", "fc1", "gate", "kloop", "reduce", "store"};
        for (int pick : {0, nb / 2, nb - 1}) {
            printf("   wg %4d start +%.2f us:", pick, (st[pick * 8] - t0) * 0.01);
            for (int i = 1; i <= 6; ++i) printf(" %s %.2f", names[i], (st[pick * 8 + i] - st[pick * 8 + i - 1]) * 0.01);
            printf("\n");
        }
        // phase medians over workgroups
        printf("   median:");
        for (int i = 1; i <= 6; ++i) {
            std::vector<long long> v;
            for (int b = 0; b < nb; ++b) v.push_back(st[b * 8 + i] - st[b * 8 + i - 1]);
            std::sort(v.begin(), v.end());
            printf(" %s %.2f", names[i], v[v.size() / 2] * 0.01);
        }
        {
            CK(hipMemset(d_st, 0, 8 * 4096 * sizeof(long long)));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
            launch_se(sa, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
            std::vector<long long> ss(8 * 64);
            CK(hipMemcpy(ss.data(), d_st, ss.size() * sizeof(long long), hipMemcpyDeviceToHost));
            const char* sn[] = {"entry", "issue+squeeze", "barrier", "fc1", "barrier", "excite"};
            printf("\n   SE wg0:");
            for (int i = 1; i <= 5; ++i) printf(" %s %.2f", sn[i], (ss[i] - ss[i - 1]) * 0.01);
        }
        std::vector<long long> starts;
        for (int b = 0; b < nb; ++b) starts.push_back(st[b * 8] - t0);
        std::sort(starts.begin(), starts.end());
        printf("  | last workgroup starts +%.2f us\n", starts.back() * 0.01);
        for (void* p : {(void*)D, (void*)Wp, (void*)bias, (void*)partial, (void*)w1t, (void*)b1, (void*)w2, (void*)w2c, (void*)b2,
                        (void*)gate, (void*)res, (void*)out})
            CK(hipFree(p));
    }
    return 0;
}
