// Probe for whenet_front7_kernel (the 7x7 blocks with a group of G crops per workgroup, round 4):
//   1. checked against a host restatement (expand 1x1 + BN + Swish -> f16 -> depthwise + BN + Swish -> f16, squeeze-excite
//      reduce-conv shares) for n = 1, 2, 5, 9 crops (tail groups) and every plan;
//   2. bitwise batch invariance: a crop's outputs inside a 9-crop launch == the same crop launched alone;
//   3. timed at 256 / 64 / 16 crops per launch next to whenet_front_kernel (the kernel the engine used on these layers up to
//      round 3) and whenet_front2_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/probes/front7_probe.hip -o tools/probes/front7_probe
#include "../../headposeestimation-whenet_amd/csrc/front.hip"
#include "../../headposeestimation-whenet_amd/csrc/front2.hip"
#include "../../headposeestimation-whenet_amd/csrc/front7.hip"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Shape { const char* name; int k, Cin, Cexp, R; };

static float frand(float scale) { return scale * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static float swish_ref(float x) { return x / (1.0f + std::exp(-x)); }

int main() {
    const Shape shapes[] = {{"b13", 5, 192, 1152, 48}, {"b16", 3, 192, 1152, 48}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* only = getenv("ONLY");
    const int NMAX = 256, NCHK = 9, H = 7, S = 1;
    int total_bad = 0;
    struct Cfg { int G, CC, thr; };
    const Cfg cfgs[] = {{4, 64, 512}, {4, 64, 256}, {4, 32, 256}, {4, 32, 512}, {8, 32, 512}, {8, 64, 512}, {2, 64, 256},
                        {4, 128, 512}, {1, 64, 256}, {2, 32, 256}, {2, 64, 512}};
    for (const Shape& sh : shapes) {
        if (only && std::string(only) != sh.name) continue;
        srand(7);
        const int Ho = 7, Cin = sh.Cin, Cexp = sh.Cexp, K = sh.k, pad = K / 2;
        const int KSe = ceil_div(Cin, 16), NTe = ceil_div(Cexp, 32), RPse = (sh.R + 3) & ~3;
        std::vector<half_t> hx(size_t(NMAX) * H * H * Cin + 64);
        for (auto& v : hx) v = half_t(frand(1.f));
        std::vector<half_t> W(size_t(Cin) * Cexp);
        for (auto& v : W) v = half_t(frand(0.5f / std::sqrt(float(Cin))));
        std::vector<half_t> wep(size_t(KSe) * NTe * 64 * 8, half_t(0));
        for (int ks = 0; ks < KSe; ++ks)
            for (int nt = 0; nt < NTe; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int n = nt * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + e;
                        if (n < Cexp && k < Cin) wep[((size_t(ks) * NTe + nt) * 64 + l) * 8 + e] = W[size_t(k) * Cexp + n];
                    }
        std::vector<float> be(NTe * 32, 0.f), wd(size_t(K) * K * Cexp), bd(Cexp), w1t(size_t(sh.R) * Cexp);
        for (int c = 0; c < Cexp; ++c) be[c] = frand(0.3f);
        for (auto& v : wd) v = frand(0.6f / K);
        for (auto& v : bd) v = frand(0.3f);
        for (auto& v : w1t) v = frand(0.05f);
        const std::vector<half_t> wdt7 = pack_dw_toeplitz(wd, K, S, Cexp, 4 - pad);
        const std::vector<half_t> wdt2 = pack_dw_toeplitz(wd, K, S, Cexp, (K == 5) ? 2 : 0);

        const half_t* d_x = upload(hx);
        const half_t* d_wep = upload(wep);
        const float* d_be = upload(be);
        const float* d_wd = upload(wd);
        const half_t* d_wdt7 = upload(wdt7);
        const half_t* d_wdt2 = upload(wdt2);
        const float* d_bd = upload(bd);
        const float* d_w1t = upload(w1t);
        half_t* d_out; CK(hipMalloc(&d_out, size_t(NMAX) * Ho * Ho * Cexp * sizeof(half_t)));
        const size_t rp_floats = size_t(NMAX) * (Cexp / 32) * RPse;
        float* d_rp; CK(hipMalloc(&d_rp, rp_floats * sizeof(float)));

        // ---- host reference for NCHK crops -----------------------------------------------------------------------------
        std::vector<float> refD(size_t(NCHK) * 49 * Cexp, 0.f), refY(refD.size(), 0.f);
        {
            std::vector<float> Eh(size_t(49) * Cexp), wdh(wd.size());
            for (size_t i = 0; i < wd.size(); ++i) wdh[i] = float(half_t(wd[i]));
            for (int b = 0; b < NCHK; ++b) {
                for (int px = 0; px < 49; ++px) {
                    const half_t* xr = &hx[(size_t(b) * 49 + px) * Cin];
                    float* er = &Eh[size_t(px) * Cexp];
                    for (int n = 0; n < Cexp; ++n) er[n] = 0.f;
                    for (int k = 0; k < Cin; ++k) {
                        const float xv = float(xr[k]);
                        const half_t* wr = &W[size_t(k) * Cexp];
                        for (int n = 0; n < Cexp; ++n) er[n] += xv * float(wr[n]);
                    }
                    for (int n = 0; n < Cexp; ++n) er[n] = float(half_t(swish_ref(er[n] + be[n])));
                }
                for (int oy = 0; oy < 7; ++oy)
                    for (int ox = 0; ox < 7; ++ox) {
                        float* o = &refD[((size_t(b) * 7 + oy) * 7 + ox) * Cexp];
                        float* y = &refY[((size_t(b) * 7 + oy) * 7 + ox) * Cexp];
                        for (int ky = 0; ky < K; ++ky)
                            for (int kx = 0; kx < K; ++kx) {
                                const int iy = oy - pad + ky, ix = ox - pad + kx;
                                if (iy < 0 || iy >= 7 || ix < 0 || ix >= 7) continue;
                                const float* er = &Eh[(size_t(iy) * 7 + ix) * Cexp];
                                const float* wr = &wdh[size_t(ky * K + kx) * Cexp];
                                for (int c = 0; c < Cexp; ++c) y[c] += er[c] * wr[c];
                            }
                        for (int c = 0; c < Cexp; ++c) {
                            y[c] = swish_ref(y[c] + bd[c]);
                            o[c] = float(half_t(y[c]));
                        }
                    }
            }
        }
        Front7Args a7{};
        a7.x = d_x; a7.wep = d_wep; a7.be = d_be; a7.wdt = d_wdt7; a7.bd = d_bd; a7.out = d_out; a7.rpart = d_rp; a7.w1t = d_w1t;
        a7.R = sh.R; a7.k = K; a7.Cin = Cin; a7.Cexp = Cexp; a7.NTe = NTe;

        auto check = [&](const Front7Plan& pl, int n, const char* what) -> int {
            std::vector<half_t> got(size_t(n) * 49 * Cexp);
            CK(hipMemcpy(got.data(), d_out, got.size() * sizeof(half_t), hipMemcpyDeviceToHost));
            int bad = 0;
            double maxerr = 0;
            for (size_t i = 0; i < got.size(); ++i) {
                const float g = float(got[i]), r = refD[i];
                const float err = std::fabs(g - r);
                if (!(err <= 4e-3f + 8e-3f * std::fabs(r))) {
                    if (bad < 5) {
                        const size_t c = i % Cexp, px = i / Cexp;
                        printf("    MISMATCH %s crop %zu oy %zu ox %zu c %zu: got %g want %g\n", what, px / 49, (px / 7) % 7, px % 7, c, g, r);
                    }
                    ++bad;
                }
                if (err > maxerr) maxerr = err;
            }
            std::vector<float> rp(size_t(n) * pl.chunks * RPse);
            CK(hipMemcpy(rp.data(), d_rp, rp.size() * sizeof(float), hipMemcpyDeviceToHost));
            int bad_se = 0;
            double max_se = 0;
            for (int b = 0; b < n; ++b)
                for (int jo = 0; jo < RPse; ++jo) {
                    double want = 0, gotv = 0, mag = 0;
                    if (jo < sh.R)
                        for (int c = 0; c < Cexp; ++c) {
                            double s = 0;
                            for (int px = 0; px < 49; ++px) s += refY[(size_t(b) * 49 + px) * Cexp + c];
                            want += s * w1t[size_t(jo) * Cexp + c];
                            mag += std::fabs(s * w1t[size_t(jo) * Cexp + c]);
                        }
                    for (int t = 0; t < pl.chunks; ++t) gotv += rp[(size_t(b) * pl.chunks + t) * RPse + jo];
                    const double err = std::fabs(want - gotv);
                    if (!(err <= 1e-3 + 3e-3 * mag)) ++bad_se;
                    max_se = std::max(max_se, err);
                }
            printf("  check %-30s n=%d: %d / %zu outputs off (max |err| %.2e), %d squeeze-excite partials off (max %.2e)\n", what, n, bad,
                   got.size(), maxerr, bad_se, max_se);
            return bad + bad_se;
        };
        auto run7 = [&](const Front7Plan& pl, int n, const half_t* x) {
            a7.plan = pl; a7.n = n; a7.x = x;
            launch_front7(a7, st);
        };
        auto time7 = [&](const Front7Plan& pl, int n) -> float {
            for (int w = 0; w < 3; ++w) run7(pl, n, d_x);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) run7(pl, n, d_x);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };
        printf("%s k%d 7x7 Cin%d Cexp%d\n", sh.name, K, Cin, Cexp);
        for (const Cfg& c : cfgs) {
            Front7Plan pl;
            try { pl = make_front7_plan(WHENET_F16, Cin, Cexp, c.G, c.CC, c.thr); } catch (const Error& e) { printf("  plan G=%d CC=%d: %s\n", c.G, c.CC, e.what()); continue; }
            if (pl.lds_bytes > 160 * 1024) { printf("  plan G=%d CC=%d thr=%d: lds %zu too large\n", c.G, c.CC, c.thr, pl.lds_bytes); continue; }
            char what[64];
            snprintf(what, sizeof what, "G=%d CC=%d thr=%d lds=%zu", c.G, c.CC, c.thr, pl.lds_bytes);
            int bad = 0;
            for (int n : {1, 2, 5, 9}) {
                CK(hipMemset(d_out, 0xff, size_t(NCHK) * 49 * Cexp * sizeof(half_t)));
                CK(hipMemset(d_rp, 0xff, rp_floats * sizeof(float)));
                run7(pl, n, d_x);
                CK(hipStreamSynchronize(st));
                bad += check(pl, n, what);
            }
            {   // batch invariance: crop 6 of the 9-crop launch (third of its group / seventh) == launched alone
                run7(pl, 9, d_x);
                CK(hipStreamSynchronize(st));
                std::vector<half_t> full(size_t(9) * 49 * Cexp), one(size_t(49) * Cexp);
                std::vector<float> rfull(size_t(9) * pl.chunks * RPse), rone(size_t(pl.chunks) * RPse);
                CK(hipMemcpy(full.data(), d_out, full.size() * sizeof(half_t), hipMemcpyDeviceToHost));
                CK(hipMemcpy(rfull.data(), d_rp, rfull.size() * sizeof(float), hipMemcpyDeviceToHost));
                for (int crop : {6, 8}) {
                    run7(pl, 1, d_x + size_t(crop) * 49 * Cin);
                    CK(hipStreamSynchronize(st));
                    CK(hipMemcpy(one.data(), d_out, one.size() * sizeof(half_t), hipMemcpyDeviceToHost));
                    CK(hipMemcpy(rone.data(), d_rp, rone.size() * sizeof(float), hipMemcpyDeviceToHost));
                    const bool same = std::memcmp(one.data(), full.data() + size_t(crop) * 49 * Cexp, one.size() * sizeof(half_t)) == 0 &&
                                      std::memcmp(rone.data(), rfull.data() + size_t(crop) * pl.chunks * RPse, rone.size() * sizeof(float)) == 0;
                    printf("  invariance %-26s crop %d alone vs inside 9: %s\n", what, crop, same ? "bitwise equal" : "DIFFERENT");
                    bad += same ? 0 : 1;
                }
            }
            total_bad += bad;
            {   // the group size changes nothing in a crop's bits: G = 4 / CC = 64 is the reference plan
                const Front7Plan ref = make_front7_plan(WHENET_F16, Cin, Cexp, 4, 64, 512);
                run7(ref, 9, d_x);
                CK(hipStreamSynchronize(st));
                std::vector<half_t> o0(size_t(9) * 49 * Cexp), o1(o0.size());
                CK(hipMemcpy(o0.data(), d_out, o0.size() * sizeof(half_t), hipMemcpyDeviceToHost));
                run7(pl, 9, d_x);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(o1.data(), d_out, o1.size() * sizeof(half_t), hipMemcpyDeviceToHost));
                const bool same = std::memcmp(o0.data(), o1.data(), o0.size() * sizeof(half_t)) == 0;
                printf("  outputs vs the G=4 CC=64 plan: %s\n", same ? "bitwise equal" : "DIFFERENT");
                bad += same ? 0 : 1;
            }
            const float t256 = time7(pl, 256), t64 = time7(pl, 64), t16 = time7(pl, 16);
            printf("  front7 %-30s: n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us | n=8 %5.2f n=4 %5.2f n=2 %5.2f n=1 %5.2f %s\n", what, t256, t64, t16,
                   time7(pl, 8), time7(pl, 4), time7(pl, 2), time7(pl, 1), bad ? "WRONG" : "");
            fflush(stdout);
        }
        // ---- the kernels the engine used on these layers before ---------------------------------------------------------
        FrontArgs a1{};
        a1.x = d_x; a1.wep = d_wep; a1.be = d_be; a1.wd = d_wd; a1.bd = d_bd; a1.out = d_out; a1.rpart = d_rp; a1.w1t = d_w1t; a1.R = sh.R;
        a1.k = K; a1.s = S; a1.H = H; a1.Ho = Ho; a1.Cin = Cin; a1.Cexp = Cexp; a1.pad = pad; a1.KSe = KSe; a1.NTe = NTe;
        a1.plan = plan_front(WHENET_F16, K, S, H, Ho, Cexp);
        Front2Args a2{};
        a2.x = d_x; a2.wep = d_wep; a2.be = d_be; a2.wdt = d_wdt2; a2.bd = d_bd; a2.out = d_out; a2.rpart = d_rp; a2.w1t = d_w1t; a2.R = sh.R;
        a2.k = K; a2.s = S; a2.H = H; a2.Ho = Ho; a2.Cin = Cin; a2.Cexp = Cexp; a2.pad = pad; a2.KSe = KSe; a2.NTe = NTe;
        a2.plan = plan_front2(K, S, H, Ho, Cexp);
        auto time_fn = [&](auto&& fn, int n) -> float {
            for (int w = 0; w < 3; ++w) fn(n);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) fn(n);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };
        auto f1 = [&](int n) { a1.n = n; a1.plan.threads = front_threads(a1.plan, n); launch_front(a1, WHENET_F16, st); };
        auto f2 = [&](int n) { a2.n = n; launch_front2(a2, st); };
        // rpart of front.hip / front2.hip: [n][tiles][chunks][RP] -- the buffer is large enough for CC = 32 chunks
        printf("  front.hip  (round 2)          : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us | n=8 %5.2f n=4 %5.2f n=2 %5.2f n=1 %5.2f\n", time_fn(f1, 256), time_fn(f1, 64), time_fn(f1, 16),
               time_fn(f1, 8), time_fn(f1, 4), time_fn(f1, 2), time_fn(f1, 1));
        printf("  front2.hip (round 3)          : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us\n", time_fn(f2, 256), time_fn(f2, 64), time_fn(f2, 16));
#ifdef WHENET_STAMPS
        for (const Cfg& c : {Cfg{4, 64, 512}, Cfg{4, 64, 256}}) {
            const Front7Plan pl = make_front7_plan(WHENET_F16, Cin, Cexp, c.G, c.CC, c.thr);
            for (int n : {256, 64}) {
                const size_t nwg = size_t(pl.chunks) * ceil_div(n, pl.G);
                long long* d_st; CK(hipMalloc(&d_st, nwg * 8 * sizeof(long long)));
                CK(hipMemset(d_st, 0, nwg * 8 * sizeof(long long)));
                run7(pl, n, d_x);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
                run7(pl, n, d_x);
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
                    t0 = std::min(t0, hs[w * 8]);
                    t1 = std::max(t1, hs[w * 8 + 6]);
                }
                printf("  timeline G=%d CC=%d thr=%d n=%d (%zu workgroups, kernel span %.1f us): workgroup life %.2f us = prologue+W stage %.2f | expand %.2f | "
                       "barrier %.2f | taps+epilogue %.2f | barrier %.2f | SE tail %.2f\n", c.G, c.CC, c.thr, n, nwg, double(t1 - t0) * 0.01,
                       life / nwg * 0.01, ph[0] / nwg * 0.01, ph[1] / nwg * 0.01, ph[2] / nwg * 0.01, ph[3] / nwg * 0.01,
                       ph[4] / nwg * 0.01, ph[5] / nwg * 0.01);
                CK(hipFree(d_st));
            }
        }
#endif
        for (const void* q : {(const void*)d_x, (const void*)d_wep, (const void*)d_be, (const void*)d_wd, (const void*)d_wdt7, (const void*)d_wdt2,
                              (const void*)d_bd, (const void*)d_w1t, (const void*)d_out, (const void*)d_rp})
            CK(hipFree(const_cast<void*>(q)));
    }
    printf("\n%d mismatches in total\n", total_bad);
    return total_bad ? 2 : 0;
}
