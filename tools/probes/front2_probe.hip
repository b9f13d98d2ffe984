// Probe for whenet_front2_kernel (depthwise taps on the matrix cores): every EfficientNet-B0 layer shape is
//   1. checked against a host restatement (expand 1x1 + BN + Swish -> f16 -> depthwise + BN + Swish -> f16, SE sums)
//      on 2 crops, for the default plan (and with TUNE=1 for every candidate plan);
//   2. timed at 256 / 64 / 16 crops per launch next to round 2's whenet_front_kernel with its tuned plan.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/front2_probe.hip -o tools/probes/front2_probe
// env: ONLY=b5 (one shape), TUNE=1 (time every candidate plan, print the table for front2_tuned.inc), NOCHECK=1
#include "../../headposeestimation-whenet_amd/csrc/front.hip"
#include "../../headposeestimation-whenet_amd/csrc/front2.hip"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Shape { const char* name; int k, s, H, Cin, Cexp, R; };

static float frand(float scale) { return scale * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static float swish_ref(float x) { return x / (1.0f + std::exp(-x)); }

int main() {
    const Shape shapes[] = {{"b2", 3, 2, 112, 16, 96, 4},     {"b3", 3, 1, 56, 24, 144, 6},    {"b4", 5, 2, 56, 24, 144, 6},
                            {"b5", 5, 1, 28, 40, 240, 10},    {"b6", 3, 2, 28, 40, 240, 10},   {"b7", 3, 1, 14, 80, 480, 20},
                            {"b9", 5, 1, 14, 80, 480, 20},    {"b10", 5, 1, 14, 112, 672, 28}, {"b12", 5, 2, 14, 112, 672, 28},
                            {"b13", 5, 1, 7, 192, 1152, 48},  {"b16", 3, 1, 7, 192, 1152, 48}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* only = getenv("ONLY");
    const bool tune = getenv("TUNE") != nullptr, nocheck = getenv("NOCHECK") != nullptr;
    const int NMAX = 256, NCHK = 2;
    std::string table;
    double sum_old = 0, sum_new = 0;
    int total_bad = 0;
    for (const Shape& sh : shapes) {
        if (only && std::string(only) != sh.name) continue;
        srand(7);
        const int Ho = ceil_div(sh.H, sh.s), H = sh.H, Cin = sh.Cin, Cexp = sh.Cexp, K = sh.k, S = sh.s;
        const int padt = std::max((Ho - 1) * S + K - H, 0), pad = padt / 2;
        const int KSe = ceil_div(Cin, 16), NTe = ceil_div(Cexp, 32);
        // ---- host tensors ----------------------------------------------------------------------------------
        std::vector<half_t> hx(size_t(NMAX) * H * H * Cin + 64);      // (+64: the kernel's k-tail reads 16 B past a pixel row)
        for (auto& v : hx) v = half_t(frand(1.f));
        std::vector<half_t> W(size_t(Cin) * Cexp);                       // [k][n], already f16
        for (auto& v : W) v = half_t(frand(0.5f / std::sqrt(float(Cin))));
        std::vector<half_t> wep(size_t(KSe) * NTe * 64 * 8, half_t(0));  // snapshot.h packing
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
        const std::vector<half_t> wdt = pack_dw_toeplitz(wd, K, S, Cexp, 0);
        const std::vector<half_t> wdt2 = (K == 5 && S == 1) ? pack_dw_toeplitz(wd, K, S, Cexp, 2) : wdt;

        const half_t* d_x = upload(hx);
        const half_t* d_wep = upload(wep);
        const float* d_be = upload(be);
        const float* d_wd = upload(wd);
        const half_t* d_wdt = upload(wdt);
        const half_t* d_wdt2 = upload(wdt2);
        const float* d_bd = upload(bd);
        const float* d_w1t = upload(w1t);
        half_t* d_out; CK(hipMalloc(&d_out, size_t(NMAX) * Ho * Ho * Cexp * sizeof(half_t)));
        const bool se_in_front = Cexp >= 480;
        const int RPse = (sh.R + 3) & ~3;
        float* d_rp; const size_t rp_floats = size_t(NMAX) * 64 * std::max(size_t(Cexp + 64), size_t(Cexp / 32 + 1) * (RPse + 4));
        CK(hipMalloc(&d_rp, rp_floats * sizeof(float)));

        // ---- host reference for NCHK crops -----------------------------------------------------------------
        std::vector<float> refD, refY;          // [NCHK][Ho][Ho][Cexp]: rounded output (as float), pre-rounding y
        if (!nocheck) {
            std::vector<float> Eh(size_t(H) * H * Cexp);
            refD.assign(size_t(NCHK) * Ho * Ho * Cexp, 0.f);
            refY = refD;
            std::vector<float> wdh(wd.size());
            for (size_t i = 0; i < wd.size(); ++i) wdh[i] = float(half_t(wd[i]));
            for (int b = 0; b < NCHK; ++b) {
                for (int px = 0; px < H * H; ++px) {
                    const half_t* xr = &hx[(size_t(b) * H * H + px) * Cin];
                    float* er = &Eh[size_t(px) * Cexp];
                    for (int n = 0; n < Cexp; ++n) er[n] = 0.f;
                    for (int k = 0; k < Cin; ++k) {
                        const float xv = float(xr[k]);
                        const half_t* wr = &W[size_t(k) * Cexp];
                        for (int n = 0; n < Cexp; ++n) er[n] += xv * float(wr[n]);
                    }
                    for (int n = 0; n < Cexp; ++n) er[n] = float(half_t(swish_ref(er[n] + be[n])));
                }
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Ho; ++ox) {
                        float* o = &refD[((size_t(b) * Ho + oy) * Ho + ox) * Cexp];
                        float* y = &refY[((size_t(b) * Ho + oy) * Ho + ox) * Cexp];
                        for (int ky = 0; ky < K; ++ky)
                            for (int kx = 0; kx < K; ++kx) {
                                const int iy = oy * S - pad + ky, ix = ox * S - pad + kx;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= H) continue;
                                const float* er = &Eh[(size_t(iy) * H + ix) * Cexp];
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
        auto check = [&](const Front2Plan& pl, const char* what) -> int {
            std::vector<half_t> got(size_t(NCHK) * Ho * Ho * Cexp);
            CK(hipMemcpy(got.data(), d_out, got.size() * sizeof(half_t), hipMemcpyDeviceToHost));
            int bad = 0;
            double maxerr = 0;
            for (size_t i = 0; i < got.size(); ++i) {
                const float g = float(got[i]), r = refD[i];
                const float err = std::fabs(g - r);
                if (!(err <= 4e-3f + 8e-3f * std::fabs(r))) {
                    if (bad < 5) {
                        const size_t c = i % Cexp, px = i / Cexp;
                        printf("    MISMATCH %s crop %zu oy %zu ox %zu c %zu: got %g want %g\n", what, px / (size_t(Ho) * Ho),
                               (px / Ho) % Ho, px % Ho, c, g, r);
                    }
                    ++bad;
                }
                if (err > maxerr) maxerr = err;
            }
            // squeeze-excite partials
            const int ntl = pl.ntiles();
            std::vector<float> rp(size_t(NCHK) * ntl * std::max(Cexp, pl.chunks * RPse));
            int bad_se = 0;
            double max_se = 0;
            if (!se_in_front) {
                CK(hipMemcpy(rp.data(), d_rp, size_t(NCHK) * ntl * Cexp * sizeof(float), hipMemcpyDeviceToHost));
                for (int b = 0; b < NCHK; ++b)
                    for (int c = 0; c < Cexp; ++c) {
                        double want = 0, gotv = 0;
                        for (int px = 0; px < Ho * Ho; ++px) want += refY[(size_t(b) * Ho * Ho + px) * Cexp + c];
                        for (int t = 0; t < ntl; ++t) gotv += rp[(size_t(b) * ntl + t) * Cexp + c];
                        const double err = std::fabs(want - gotv);
                        if (err > 2e-2 + 2e-3 * std::fabs(want) + 1e-3 * Ho * Ho * 0.3) ++bad_se;
                        max_se = std::max(max_se, err);
                    }
            } else {
                CK(hipMemcpy(rp.data(), d_rp, size_t(NCHK) * ntl * pl.chunks * RPse * sizeof(float), hipMemcpyDeviceToHost));
                for (int b = 0; b < NCHK; ++b)
                    for (int jo = 0; jo < RPse; ++jo) {
                        double want = 0, gotv = 0, mag = 0;
                        if (jo < sh.R)
                            for (int c = 0; c < Cexp; ++c) {
                                double s = 0;
                                for (int px = 0; px < Ho * Ho; ++px) s += refY[(size_t(b) * Ho * Ho + px) * Cexp + c];
                                want += s * w1t[size_t(jo) * Cexp + c];
                                mag += std::fabs(s * w1t[size_t(jo) * Cexp + c]);
                            }
                        for (int t = 0; t < ntl * pl.chunks; ++t) gotv += rp[(size_t(b) * ntl * pl.chunks + t) * RPse + jo];
                        const double err = std::fabs(want - gotv);
                        if (err > 1e-3 + 3e-3 * mag) ++bad_se;
                        max_se = std::max(max_se, err);
                    }
            }
            printf("  check %-26s: %d / %zu outputs off (max |err| %.2e), %d squeeze-excite partials off (max %.2e)\n", what, bad,
                   got.size(), maxerr, bad_se, max_se);
            return bad + bad_se;
        };

        Front2Args a2{};
        a2.x = d_x; a2.wep = d_wep; a2.be = d_be; a2.wdt = d_wdt; a2.bd = d_bd; a2.out = d_out; a2.rpart = d_rp;
        a2.w1t = se_in_front ? d_w1t : nullptr; a2.R = sh.R;
        a2.k = K; a2.s = S; a2.H = H; a2.Ho = Ho; a2.Cin = Cin; a2.Cexp = Cexp; a2.pad = pad; a2.KSe = KSe; a2.NTe = NTe;
        FrontArgs a1{};
        a1.x = d_x; a1.wep = d_wep; a1.be = d_be; a1.wd = d_wd; a1.bd = d_bd; a1.out = d_out; a1.rpart = d_rp;
        a1.w1t = se_in_front ? d_w1t : nullptr; a1.R = sh.R;
        a1.k = K; a1.s = S; a1.H = H; a1.Ho = Ho; a1.Cin = Cin; a1.Cexp = Cexp; a1.pad = pad; a1.KSe = KSe; a1.NTe = NTe;
        a1.plan = plan_front(WHENET_F16, K, S, H, Ho, Cexp);

        auto time2 = [&](const Front2Plan& pl, int n, int threads) -> float {
            a2.n = n;
            a2.plan = pl;
            a2.wdt = pl.xs ? d_wdt2 : d_wdt;
            a2.plan.threads = threads ? threads : front2_threads(pl, n);
            for (int w = 0; w < 3; ++w) launch_front2(a2, st);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) launch_front2(a2, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };
        auto time1 = [&](int n) -> float {
            a1.n = n;
            a1.plan.threads = front_threads(a1.plan, n);
            for (int w = 0; w < 3; ++w) launch_front(a1, WHENET_F16, st);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) launch_front(a1, WHENET_F16, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };

        const Front2Plan def = plan_front2(K, S, H, Ho, Cexp);
        printf("%s k%d s%d H%d Cin%d Cexp%d: plan CC=%d TH=%d TXG=%d tiles=%dx%d chunks=%d E=%dx%d RP=%d CP=%d lds=%zu\n", sh.name, K, S,
               H, Cin, Cexp, def.CC, def.TH, def.TXG, def.tiles_x, def.tiles_y, def.chunks, def.EH, def.EWp, def.RP, def.CP, def.lds_bytes);
        fflush(stdout);
        if (!nocheck) {
            for (int threads : {256, 512}) {
                CK(hipMemset(d_out, 0xff, size_t(NCHK) * Ho * Ho * Cexp * sizeof(half_t)));
                CK(hipMemset(d_rp, 0xff, rp_floats * sizeof(float)));
                a2.n = NCHK; a2.plan = def; a2.plan.threads = threads; a2.wdt = def.xs ? d_wdt2 : d_wdt;
                launch_front2(a2, st);
                CK(hipStreamSynchronize(st));
                total_bad += check(def, threads == 256 ? "default plan, 256 lanes" : "default plan, 512 lanes");
            }
        }
#ifdef WHENET_STAMPS
        {   // phase timeline of the default plan at 256 and 16 crops per launch: per-workgroup stamps (wave 0), averaged
            for (int n : {256, 16}) {
                const size_t nwg = size_t(n) * def.ntiles() * def.chunks;
                long long* d_st; CK(hipMalloc(&d_st, nwg * 8 * sizeof(long long)));
                CK(hipMemset(d_st, 0, nwg * 8 * sizeof(long long)));
                a2.n = n; a2.plan = def; a2.plan.threads = 256; a2.wdt = def.xs ? d_wdt2 : d_wdt;
                launch_front2(a2, st);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
                launch_front2(a2, st);
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
                printf("  timeline n=%d (%zu workgroups, kernel span %.1f us): workgroup life %.2f us = prologue %.2f | expand %.2f | "
                       "barrier+fixup %.2f | taps+epilogue %.2f | barrier %.2f | tail %.2f\n", n, nwg, double(t1 - t0) * 0.01,
                       life / nwg * 0.01, ph[0] / nwg * 0.01, ph[1] / nwg * 0.01, ph[2] / nwg * 0.01, ph[3] / nwg * 0.01,
                       ph[4] / nwg * 0.01, ph[5] / nwg * 0.01);
                CK(hipFree(d_st));
            }
        }
#endif
        const float o256 = time1(256), o64 = time1(64), o16 = time1(16);
        const float n256 = time2(def, 256, 0), n64 = time2(def, 64, 0), n16 = time2(def, 16, 0);
        printf("  round-2 kernel : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us\n", o256, o64, o16);
        printf("  front2 default : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us   (x%.2f at 64)\n", n256, n64, n16, o64 / n64);
        sum_old += o64 * (std::string(sh.name) == "b7" ? 2 : (std::string(sh.name) == "b10" ? 2 : (std::string(sh.name) == "b13" ? 3 : 1)));
        sum_new += n64 * (std::string(sh.name) == "b7" ? 2 : (std::string(sh.name) == "b10" ? 2 : (std::string(sh.name) == "b13" ? 3 : 1)));
        fflush(stdout);
        if (tune) {
            struct Row { Front2Plan p; float t256, t64, t16; int bad; };
            std::vector<Row> rows;
            for (const Front2Plan& pl : plan_front2_candidates(K, S, Ho, Cexp)) {
                Row r{pl, 0, 0, 0, 0};
                try {
                if (!nocheck) {
                    CK(hipMemset(d_out, 0xff, size_t(NCHK) * Ho * Ho * Cexp * sizeof(half_t)));
                    a2.n = NCHK; a2.plan = pl; a2.plan.threads = 256; a2.wdt = pl.xs ? d_wdt2 : d_wdt;
                    launch_front2(a2, st);
                    CK(hipStreamSynchronize(st));
                    char what[64];
                    snprintf(what, sizeof what, "CC=%d TH=%d TXG=%d xs=%d", pl.CC, pl.TH, pl.TXG, pl.xs);
                    r.bad = check(pl, what);
                    total_bad += r.bad;
                }
                r.t256 = time2(pl, 256, 0); r.t64 = time2(pl, 64, 0); r.t16 = time2(pl, 16, 0);
                rows.push_back(r);
                } catch (const Error& e) {
                    printf("   plan CC=%d TH=%d TXG=%d lds=%zu: launch failed (%s)\n", pl.CC, pl.TH, pl.TXG, pl.lds_bytes, e.what());
                    (void)hipGetLastError();
                }
            }
            auto merit = [](const Row& r) { return r.t256 / 4.0f + r.t64 + r.t16; };
            std::sort(rows.begin(), rows.end(), [&](const Row& x, const Row& y) { return merit(x) < merit(y); });
            for (size_t i = 0; i < rows.size() && i < 8; ++i)
                printf("   #%zu CC=%3d TH=%2d TXG=%2d xs=%d tiles=%dx%d chunks=%2d lds=%6zu : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us %s\n", i + 1,
                       rows[i].p.CC, rows[i].p.TH, rows[i].p.TXG, rows[i].p.xs, rows[i].p.tiles_x, rows[i].p.tiles_y, rows[i].p.chunks, rows[i].p.lds_bytes,
                       rows[i].t256, rows[i].t64, rows[i].t16, rows[i].bad ? "WRONG" : "");
            if (!rows.empty()) {
                char line[200];
                snprintf(line, sizeof line, "    {%d, %d, %d, %d, %d, %d, %d, %d, 1},   // %s: %.1f us @256, %.1f us @64, %.1f us @16\n", K, S, H, Cexp,
                         rows[0].p.CC, rows[0].p.TH, rows[0].p.TXG, rows[0].p.xs, sh.name, rows[0].t256, rows[0].t64, rows[0].t16);
                table += line;
            }
            fflush(stdout);
        }
        for (const void* q : {(const void*)d_x, (const void*)d_wep, (const void*)d_be, (const void*)d_wd, (const void*)d_wdt, (const void*)d_wdt2,
                              (const void*)d_bd, (const void*)d_w1t, (const void*)d_out, (const void*)d_rp})
            CK(hipFree(const_cast<void*>(q)));
    }
    printf("\nsum over the 15 fused layers at 64 crops per launch: round-2 kernel %.1f us, front2 %.1f us; %d mismatches in total\n", sum_old,
           sum_new, total_bad);
    if (tune) printf("\n// front2_tuned.inc\n%s", table.c_str());
    return total_bad ? 2 : 0;
}
