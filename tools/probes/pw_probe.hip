// Probe for the K >= 320 project GEMMs (whenet_pw_splitk_kernel): the b13 project shape (M = n * 49, K = 1152, N = 192, SE gate,
// skip) and the b10 shape (M = n * 196, K = 672, N = 112), timed at 256 / 64 / 16 / 1 crops, and with -DWHENET_STAMPS the phase
// timeline of a workgroup (wave 0): prologue + gate staging | k-loop | combine barrier | epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWHENET_STAMPS] tools/probes/pw_probe.hip -o tools/probes/pw_probe
#include "../../headposeestimation-whenet_amd/csrc/pw.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float frand(float s) { return s * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
struct Shape { const char* name; int HW, K, N; bool res; };

int main() {
    const Shape shapes[] = {{"b13 project", 49, 1152, 192, true}, {"b10 project", 196, 672, 112, true}, {"b16 project", 49, 1152, 320, false}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NMAX = 256;
    for (const Shape& sh : shapes) {
        srand(3);
        const int K = sh.K, N = sh.N, KS = ceil_div(K, 16), NT = ceil_div(N, 32);
        std::vector<half_t> a(size_t(NMAX) * sh.HW * K), wp(size_t(KS) * NT * 64 * 8), gate(size_t(NMAX) * K), res(size_t(NMAX) * sh.HW * N);
        for (auto& v : a) v = half_t(frand(1.f));
        for (auto& v : wp) v = half_t(frand(0.03f));
        for (auto& v : gate) v = half_t(0.5f + frand(0.4f));
        for (auto& v : res) v = half_t(frand(1.f));
        std::vector<float> bias(NT * 32, 0.1f);
        PwArgs p{};
        p.a = upload(a); p.wp = upload(wp); p.bias = upload(bias); p.gate = upload(gate); p.res = sh.res ? upload(res) : nullptr;
        half_t* d_out; CK(hipMalloc(&d_out, size_t(NMAX) * sh.HW * N * 2)); p.out = d_out;
        p.K = K; p.N = N; p.KS = KS; p.NTILES = NT; p.HW = sh.HW; p.act = ACT_NONE;
        printf("%s: K %d N %d HW %d\n", sh.name, K, N, sh.HW);
        for (int n : {256, 64, 16, 1}) {
            p.M = n * sh.HW;
            for (int w = 0; w < 3; ++w) launch_pw(p, WHENET_F16, 0, 256, st);
            CK(hipStreamSynchronize(st));
            const int iters = 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) launch_pw(p, WHENET_F16, 0, 256, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  n=%3d (%s): %7.2f us\n", n, kernel_name_pw(p, WHENET_F16, 0, 256).c_str(), ms * 1000.f / iters);
#ifdef WHENET_STAMPS
            {
                const bool two = use_split2(p.M, NT);
                const int B2 = two ? 2 : 1;
                const int MT = ceil_div(p.M, 32 * B2), NCH = ceil_div(NT, B2);
                const size_t nwg = size_t(8) * ceil_div(MT, 8) * NCH;
                long long* d_st; CK(hipMalloc(&d_st, nwg * 8 * sizeof(long long)));
                CK(hipMemset(d_st, 0, nwg * 8 * sizeof(long long)));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
                launch_pw(p, WHENET_F16, 0, 256, st);
                CK(hipStreamSynchronize(st));
                long long* nul = nullptr;
                CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
                std::vector<long long> hs(nwg * 8);
                CK(hipMemcpy(hs.data(), d_st, hs.size() * sizeof(long long), hipMemcpyDeviceToHost));
                double ph[4] = {0, 0, 0, 0}, life = 0; size_t cnt = 0;
                long long t0 = 0, t1 = 0;
                for (size_t w = 0; w < nwg; ++w) {
                    if (hs[w * 8] == 0) continue;                      // (workgroups past MT return before the first stamp)
                    for (int i = 0; i < 4; ++i) ph[i] += double(hs[w * 8 + i + 1] - hs[w * 8 + i]);
                    life += double(hs[w * 8 + 4] - hs[w * 8]);
                    t0 = (cnt == 0 || hs[w * 8] < t0) ? hs[w * 8] : t0;
                    t1 = std::max(t1, hs[w * 8 + 4]);
                    ++cnt;
                }
                printf("     timeline: %zu workgroups, kernel span %.1f us; workgroup life %.2f us = prologue + gate %.2f | k-loop %.2f | combine "
                       "barrier %.2f | sum + epilogue %.2f\n", cnt, double(t1 - t0) * 0.01, life / cnt * 0.01, ph[0] / cnt * 0.01, ph[1] / cnt * 0.01,
                       ph[2] / cnt * 0.01, ph[3] / cnt * 0.01);
                CK(hipFree(d_st));
            }
#endif
        }
    }
    return 0;
}
