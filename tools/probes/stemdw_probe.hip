// Probe for stemdw.hip: the fused stem + block-1 depthwise kernel against stem.hip followed by dw.hip on random bytes --
// bitwise comparison of the depthwise output and the per-tile channel sums (first mismatches are printed), and timing
// at 256 / 64 / 16 / 1 crops.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWHENET_STAMPS] [-DPROBE_F32] -Iinclude tools/probes/stemdw_probe.hip -o tools/probes/stemdw_probe
#define WHENET_STEMDW_DEBUG 1
#include "../../headposeestimation-whenet_amd/csrc/stem.hip"
#include "../../headposeestimation-whenet_amd/csrc/dw.hip"
#include "../../headposeestimation-whenet_amd/csrc/stemdw.hip"

#include <cmath>
#include <cstring>
#include <vector>

using namespace whenet;
#ifdef PROBE_F32
using AT = float; constexpr int DT = WHENET_F32;
#else
using AT = half_t; constexpr int DT = WHENET_F16;
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float frand(float s) { return s * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main() {
    const int NMAX = 256;
    srand(11);
    std::vector<uint8_t> img(size_t(NMAX) * 224 * 224 * 3);
    // smooth images by default (neighbouring bytes close: the LUT reads of a wave mostly share banks, as on photographs);
    // NOISE=1: independent random bytes, the worst case for the LUT reads
    const bool noise = getenv("NOISE") != nullptr;
    for (size_t i = 0; i < img.size(); ++i) {
        const int x = int((i / 3) % 224), y = int((i / 3 / 224) % 224), c = int(i % 3);
        img[i] = noise ? uint8_t(rand() & 255) : uint8_t(128 + 100 * std::sin(0.05 * x + 0.03 * y + c) + (rand() % 7));
    }
    std::vector<float> w(27 * 32), b(32), lut(768), wd(9 * 32), bd(32);
    for (auto& v : w) v = frand(0.3f);
    for (auto& v : b) v = frand(0.2f);
    for (auto& v : wd) v = frand(0.4f);
    for (auto& v : bd) v = frand(0.2f);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 256; ++i) lut[c * 256 + i] = (float(i) / 255.f - mean[c]) / sd[c];
    const uint8_t* d_img = upload(img);
    const float *d_w = upload(w), *d_b = upload(b), *d_lut = upload(lut), *d_wd = upload(wd), *d_bd = upload(bd);
    StemDwTable htab; build_stemdw_table(w.data(), lut.data(), &htab);
    StemDwTable* d_tab; CK(hipMalloc(&d_tab, sizeof(htab))); CK(hipMemcpy(d_tab, &htab, sizeof(htab), hipMemcpyHostToDevice));
    const size_t act = size_t(NMAX) * 112 * 112 * 32;
    AT *d_stem, *d_dw0, *d_dw1; float *d_p0, *d_p1;
    CK(hipMalloc(&d_stem, act * sizeof(AT))); CK(hipMalloc(&d_dw0, act * sizeof(AT))); CK(hipMalloc(&d_dw1, act * sizeof(AT)));
    CK(hipMalloc(&d_p0, size_t(NMAX) * 56 * 32 * 4)); CK(hipMalloc(&d_p1, size_t(NMAX) * 56 * 32 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    const DwPlan plan = plan_dw(DT, 3, 1, 112, 112, 32);
    printf("plan: threads %d CV %d TH %d NSX %d tiles %d x %d chunks %d -> stemdw_supported %d\n", plan.threads, plan.CV, plan.TH, plan.NSX,
           plan.tiles_x, plan.tiles_y, plan.chunks, int(stemdw_supported(DT, plan, 3, 1, 112, 32)));
    auto two = [&](int n) {
        StemArgs a{d_img, d_stem, d_w, d_b, d_lut, n};
        launch_stem(a, DT, st);
        DwArgs d{};
        d.in = d_stem; d.out = d_dw0; d.w = d_wd; d.bias = d_bd; d.partial = d_p0; d.k = 3; d.s = 1; d.H = 112; d.Ho = 112; d.C = 32; d.pad = 1;
        d.n = n; d.plan = plan;
        launch_dw(d, DT, st);
    };
    auto one = [&](int n) {
        StemDwArgs a{DT, d_img, d_dw1, d_tab, d_w, d_lut, d_b, d_wd, d_bd, d_p1, n};
        launch_stemdw(a, st);
    };
    const int NCHK = 3;
    CK(hipMemset(d_dw1, 0xff, act * sizeof(AT)));
#ifndef PROBE_F32
    half_t* d_sdbg; CK(hipMalloc(&d_sdbg, size_t(NCHK) * 112 * 112 * 32 * 2));
    CK(hipMemset(d_sdbg, 0xff, size_t(NCHK) * 112 * 112 * 32 * 2));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stemdw_dbg), &d_sdbg, sizeof(d_sdbg)));
#endif
    two(NCHK); one(NCHK);
    CK(hipStreamSynchronize(st));
#ifndef PROBE_F32
    {
        half_t* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stemdw_dbg), &nul, sizeof(nul)));
        std::vector<uint16_t> s0(size_t(NCHK) * 112 * 112 * 32), s1(s0.size());
        CK(hipMemcpy(s0.data(), d_stem, s0.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), d_sdbg, s1.size() * 2, hipMemcpyDeviceToHost));
        size_t sb = 0;
        for (size_t i = 0; i < s0.size(); ++i)
            if (s0[i] != s1[i]) {
                if (sb < 12) printf("  stem mismatch crop %d y %3d x %3d c %2d: stem.hip %04x fused %04x\n", int(i / 32 / 112 / 112), int((i / 32 / 112) % 112),
                                    int((i / 32) % 112), int(i % 32), s0[i], s1[i]);
                ++sb;
            }
        printf("stem values: %zu of %zu differ\n", sb, s0.size());
    }
#endif
#ifdef PROBE_F32
    std::vector<uint32_t> h0(size_t(NCHK) * 112 * 112 * 32), h1(h0.size());
#else
    std::vector<uint16_t> h0(size_t(NCHK) * 112 * 112 * 32), h1(h0.size());
#endif
    CK(hipMemcpy(h0.data(), d_dw0, h0.size() * sizeof(AT), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), d_dw1, h1.size() * sizeof(AT), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i)
        if (h0[i] != h1[i]) {
            if (bad < 12) {
                const int c = int(i % 32), x = int((i / 32) % 112), y = int((i / 32 / 112) % 112), n = int(i / 32 / 112 / 112);
                printf("  mismatch crop %d y %3d x %3d c %2d: two %08x one %08x\n", n, y, x, c, unsigned(h0[i]), unsigned(h1[i]));
            }
            ++bad;
        }
    std::vector<float> p0(size_t(NCHK) * 56 * 32), p1(p0.size());
    CK(hipMemcpy(p0.data(), d_p0, p0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p1.data(), d_p1, p1.size() * 4, hipMemcpyDeviceToHost));
    printf("depthwise output: %zu of %zu values differ; channel sums %s\n", bad, h0.size(),
           std::memcmp(p0.data(), p1.data(), p0.size() * 4) == 0 ? "bitwise equal" : "DIFFER");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, int n) {
        for (int i = 0; i < 3; ++i) fn(n);
        CK(hipEventRecord(e0, st));
        const int it = n >= 64 ? 20 : 50;
        for (int i = 0; i < it; ++i) fn(n);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.f / it;
    };
#ifdef WHENET_STAMPS
    for (int n : {256, 64, 16}) {
        const size_t nwg = size_t(56) * n;
        long long* d_st; CK(hipMalloc(&d_st, nwg * 8 * sizeof(long long)));
        CK(hipMemset(d_st, 0, nwg * 8 * sizeof(long long)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
        one(n);
        CK(hipStreamSynchronize(st));
        long long* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &nul, sizeof(nul)));
        std::vector<long long> hs(nwg * 8);
        CK(hipMemcpy(hs.data(), d_st, hs.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double ph[6] = {0, 0, 0, 0, 0, 0}, life = 0;
        long long t0 = hs[0], t1 = 0;
        for (size_t w = 0; w < nwg; ++w) {
            for (int i = 0; i < 6; ++i) ph[i] += double(hs[w * 8 + i + 1] - hs[w * 8 + i]);
            life += double(hs[w * 8 + 6] - hs[w * 8]);
            t0 = std::min(t0, hs[w * 8]); t1 = std::max(t1, hs[w * 8 + 6]);
        }
        printf("n=%3d timeline: %zu workgroups, kernel span %.1f us; workgroup life %.2f us = loads + LUT %.2f | patch -> LDS %.2f | stem strips %.2f | "
               "depthwise taps %.2f | swish + stores %.2f | channel sums %.2f\n", n, nwg, double(t1 - t0) * 0.01, life / nwg * 0.01, ph[0] / nwg * 0.01,
               ph[1] / nwg * 0.01, ph[2] / nwg * 0.01, ph[3] / nwg * 0.01, ph[4] / nwg * 0.01, ph[5] / nwg * 0.01);
        CK(hipFree(d_st));
    }
#endif
    for (int n : {256, 64, 32, 16, 1}) printf("n=%3d: stem + dw %8.2f us   stemdw %8.2f us\n", n, timeit(two, n), timeit(one, n));
    return 0;
}
