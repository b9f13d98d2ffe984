// Probe for whenet_mb7_kernel (one launch per MBConv block of the 7 x 7 stage): launch time at 1 / 16 / 64 / 192 / 256 crops for both
// instantiations and the per-workgroup phase timeline (stamps of wave 0's lane 0, averaged over the workgroups).  Correctness is
// tests/test_mb7.py's business; the operands here are random.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DWHENET_STAMPS tools/probes/mb7_probe.hip -o tools/probes/mb7_probe
#include "../../headposeestimation-whenet_amd/csrc/mb7.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float frand(float scale) { return scale * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <typename T> std::vector<T> rnd(size_t n, float scale) { std::vector<T> v(n); for (auto& x : v) x = T(frand(scale)); return v; }

int main() {
    const int NMAX = 256;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        const int k = variant ? 3 : 5, Cout = variant ? 320 : 192;
        const bool skip = !variant;
        Mb7Args a{};
        a.x = upload(rnd<half_t>(size_t(NMAX) * 49 * 192, 1.0f));
        a.wep = upload(rnd<half_t>(size_t(12) * 36 * 64 * 8, 0.1f));
        a.be = upload(rnd<float>(1152, 0.1f));
        a.wds = upload(pack_mb7_taps(rnd<float>(size_t(k) * k * 1152, 0.2f), k, 1152));
        a.bd = upload(rnd<float>(1152, 0.1f));
        std::vector<half_t> w1p, w2p;
        pack_mb7_se(rnd<float>(48 * 1152, 0.05f), rnd<float>(48 * 1152, 0.05f), 1152, 48, &w1p, &w2p);
        a.w1p = upload(w1p);
        a.b1 = upload(rnd<float>(48, 0.1f));
        a.w2p = upload(w2p);
        a.b2 = upload(rnd<float>(1152, 0.1f));
        a.wpp = upload(rnd<half_t>(size_t(72) * (Cout / 32) * 64 * 8, 0.05f));
        a.bp = upload(rnd<float>(Cout, 0.1f));
        half_t* d_out; CK(hipMalloc(&d_out, size_t(NMAX) * 49 * Cout * 2));
        a.out = d_out;
        a.k = k; a.Cout = Cout; a.skip = skip;
        for (int n : {1, 16, 64, 192, 256}) {
            a.n = n;
            for (int i = 0; i < 5; ++i) launch_mb7(a, st);
            CK(hipEventRecord(e0, st));
            const int it = 50;
            for (int i = 0; i < it; ++i) launch_mb7(a, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mb7<%d, %d> n=%3d: %7.2f us per launch (back to back)\n", k, Cout / 32, n, ms * 1e3 / it);
#ifdef WHENET_STAMPS
            long long* d_st; CK(hipMalloc(&d_st, size_t(n) * 8 * sizeof(long long)));
            CK(hipMemset(d_st, 0, size_t(n) * 8 * sizeof(long long)));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
            launch_mb7(a, st);
            CK(hipStreamSynchronize(st));
            long long* nul = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
            std::vector<long long> h(size_t(n) * 8);
            CK(hipMemcpy(h.data(), d_st, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double ph[7] = {0, 0, 0, 0, 0, 0, 0};
            long long lo = h[0], hi = h[6];
            for (int b = 0; b < n; ++b) {
                for (int i = 0; i < 6; ++i) ph[i] += double(h[b * 8 + i + 1] - h[b * 8 + i]) / 100.0;
                ph[6] += double(h[b * 8 + 6] - h[b * 8]) / 100.0;
                lo = std::min(lo, h[b * 8]); hi = std::max(hi, h[b * 8 + 6]);
            }
#ifdef WHENET_MB7_TILE_STAMPS
            printf("   second tile of wave 0: %.2f us = expand MFMAs (incl. waiting for bias + weights) %.2f | Swish + E %.2f | taps 0 %.2f | epilogue 0 %.2f | taps 1 %.2f | epilogue 1 %.2f\n",
                   ph[6] / n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
#else
            printf("   timeline (kernel span %.1f us): workgroup life %.2f us = stage input %.2f | expand + taps %.2f | barrier %.2f | squeeze-excite %.2f | project %.2f | combine + store %.2f\n",
                   double(hi - lo) / 100.0, ph[6] / n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
#endif
            CK(hipFree(d_st));
#endif
        }
        fflush(stdout);
    }
    return 0;
}
