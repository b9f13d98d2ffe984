// Probe for whenet_front2s_kernel (WHENET_F32S fronts with both convolutions on the matrix cores): every EfficientNet-B0
// layer shape of blocks 2 - 12 is
//   1. checked against a float64 host restatement (expand 1x1 + BN + Swish -> depthwise + BN + Swish, SE sums) on 2 crops, in
//      both tap modes (1: binary16 hi/lo pairs, 2: exact float32) and both input forms (float32 / pre-split pairs);
//   2. timed at 256 / 64 / 16 crops per launch next to whenet_front_kernel<float, .., true> (round 5's f32s kernel) with its tuned plan.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DWHENET_F2S_ALL tools/probes/front2s_probe.hip -o tools/probes/front2s_probe
// env: ONLY=b5 (one shape), TUNE=1 (time every candidate plan in both tap modes, print the table for front2s_tuned.inc), NOCHECK=1
#include "../../headposeestimation-whenet_amd/csrc/front.hip"
#include "../../headposeestimation-whenet_amd/csrc/front2s.hip"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Shape { const char* name; int k, s, H, Cin, Cexp, R, mult; };

static float frand(float scale) { return scale * (float(rand() % 2001) / 1000.f - 1.f); }
template <typename T> T* upload(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static double swish_ref(double x) { return x / (1.0 + std::exp(-x)); }

int main() {
    const Shape shapes[] = {{"b2", 3, 2, 112, 16, 96, 4, 1},     {"b3", 3, 1, 56, 24, 144, 6, 1},    {"b4", 5, 2, 56, 24, 144, 6, 1},
                            {"b5", 5, 1, 28, 40, 240, 10, 1},    {"b6", 3, 2, 28, 40, 240, 10, 1},   {"b7", 3, 1, 14, 80, 480, 20, 2},
                            {"b9", 5, 1, 14, 80, 480, 20, 1},    {"b10", 5, 1, 14, 112, 672, 28, 2}, {"b12", 5, 2, 14, 112, 672, 28, 1}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* only = getenv("ONLY");
    const bool tune = getenv("TUNE") != nullptr, nocheck = getenv("NOCHECK") != nullptr;
    const int NMAX = 256, NCHK = 2;
    std::string table;
    double sum_old = 0, sum_new[2][2] = {{0, 0}, {0, 0}}, sum_best = 0;
    int total_bad = 0;
    for (const Shape& sh : shapes) {
        if (only && std::string(only) != sh.name) continue;
        srand(7);
        const int Ho = ceil_div(sh.H, sh.s), H = sh.H, Cin = sh.Cin, Cexp = sh.Cexp, K = sh.k, S = sh.s;
        const int padt = std::max((Ho - 1) * S + K - H, 0), pad = padt / 2;
        const int KSe = ceil_div(Cin, 16), NTe = ceil_div(Cexp, 32);
        // ---- host tensors ----------------------------------------------------------------------------------
        // x: values that ARE hi + lo in binary16 (so that the float32 and the pre-split form hold the same numbers)
        const size_t nx = size_t(NMAX) * H * H * Cin;
        std::vector<float> hx(nx + 64, 0.f);
        std::vector<half_t> hxp(2 * nx + 128, half_t(0));                 // [pixel][hi Cin | lo Cin]
        for (size_t i = 0; i < nx; ++i) {
            const float v = frand(1.f) + frand(1e-3f);
            const half_t h = half_t(v), l = half_t(v - float(h));
            hx[i] = float(h) + float(l);
            const size_t px = i / Cin, c = i % Cin;
            hxp[px * 2 * Cin + c] = h;
            hxp[px * 2 * Cin + Cin + c] = l;
        }
        std::vector<float> W(size_t(Cin) * Cexp);                        // [k][n]
        for (auto& v : W) v = frand(0.5f / std::sqrt(float(Cin)));
        // split image (snapshot.cpp::pack_pw_split)
        float wsi = 1.f;
        std::vector<half_t> weps(size_t(2) * KSe * NTe * 64 * 8, half_t(0));
        {
            double mx = 0;
            for (float v : W) mx = std::max(mx, double(std::fabs(v)));
            const int shift = int(std::floor(std::log2(16384.0 / mx)));
            const double sc = std::ldexp(1.0, shift);
            wsi = float(std::ldexp(1.0, -shift));
            const size_t per = size_t(KSe) * NTe * 64 * 8;
            for (int ks = 0; ks < KSe; ++ks)
                for (int nt = 0; nt < NTe; ++nt)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int n = nt * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + e;
                            const double v = (n < Cexp && k < Cin) ? double(W[size_t(k) * Cexp + n]) * sc : 0.0;
                            const half_t h = half_t(float(v));
                            const size_t i = ((size_t(ks) * NTe + nt) * 64 + l) * 8 + e;
                            weps[i] = h;
                            weps[per + i] = half_t(float(v - double(float(h))));
                        }
        }
        std::vector<float> be(NTe * 32, 0.f), wd(size_t(K) * K * Cexp), bd(Cexp), w1t(size_t(sh.R) * Cexp);
        for (int c = 0; c < Cexp; ++c) be[c] = frand(0.3f);
        for (auto& v : wd) v = frand(0.6f / K);
        for (auto& v : bd) v = frand(0.3f);
        for (auto& v : w1t) v = frand(0.05f);
        float wsi_d1 = 1.f, wsi_d2 = 1.f;
        const std::vector<float> wdt1 = pack_dw_toeplitz_s(wd, K, S, Cexp, 1, &wsi_d1);
        const std::vector<float> wdt2 = pack_dw_toeplitz_s(wd, K, S, Cexp, 2, &wsi_d2);

        const float* d_x = upload(hx);
        const half_t* d_xp = upload(hxp);
        const half_t* d_weps = upload(weps);
        const float* d_be = upload(be);
        const float* d_wd = upload(wd);
        const float* d_wdt1 = upload(wdt1);
        const float* d_wdt2 = upload(wdt2);
        const float* d_bd = upload(bd);
        const float* d_w1t = upload(w1t);
        float* d_out; CK(hipMalloc(&d_out, size_t(NMAX) * Ho * Ho * Cexp * sizeof(float)));
        const bool se_in_front = true;            // (the engine's fused fronts always apply the reduce conv)
        const int RPse = (sh.R + 3) & ~3;
        float* d_rp; const size_t rp_floats = size_t(NMAX) * 64 * std::max(size_t(Cexp + 64), size_t(Cexp / 32 + 1) * (RPse + 4));
        CK(hipMalloc(&d_rp, rp_floats * sizeof(float)));

        // ---- host reference for NCHK crops (float64) --------------------------------------------------------
        std::vector<double> refY;               // [NCHK][Ho][Ho][Cexp]
        if (!nocheck) {
            std::vector<double> Eh(size_t(H) * H * Cexp);
            refY.assign(size_t(NCHK) * Ho * Ho * Cexp, 0.0);
            for (int b = 0; b < NCHK; ++b) {
                for (int px = 0; px < H * H; ++px) {
                    const float* xr = &hx[(size_t(b) * H * H + px) * Cin];
                    double* er = &Eh[size_t(px) * Cexp];
                    for (int n = 0; n < Cexp; ++n) er[n] = 0.0;
                    for (int k = 0; k < Cin; ++k) {
                        const double xv = xr[k];
                        const float* wr = &W[size_t(k) * Cexp];
                        for (int n = 0; n < Cexp; ++n) er[n] += xv * double(wr[n]);
                    }
                    for (int n = 0; n < Cexp; ++n) er[n] = swish_ref(er[n] + be[n]);
                }
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Ho; ++ox) {
                        double* y = &refY[((size_t(b) * Ho + oy) * Ho + ox) * Cexp];
                        for (int ky = 0; ky < K; ++ky)
                            for (int kx = 0; kx < K; ++kx) {
                                const int iy = oy * S - pad + ky, ix = ox * S - pad + kx;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= H) continue;
                                const double* er = &Eh[(size_t(iy) * H + ix) * Cexp];
                                const float* wr = &wd[size_t(ky * K + kx) * Cexp];
                                for (int c = 0; c < Cexp; ++c) y[c] += er[c] * double(wr[c]);
                            }
                        for (int c = 0; c < Cexp; ++c) y[c] = swish_ref(y[c] + bd[c]);
                    }
            }
        }
        auto check = [&](int ntl, int chunks, const char* what) -> int {
            std::vector<float> got(size_t(NCHK) * Ho * Ho * Cexp);
            CK(hipMemcpy(got.data(), d_out, got.size() * sizeof(float), hipMemcpyDeviceToHost));
            int bad = 0;
            double maxerr = 0;
            for (size_t i = 0; i < got.size(); ++i) {
                const double g = got[i], r = refY[i];
                const double err = std::fabs(g - r);
                if (!(err <= 2e-5 + 2e-5 * std::fabs(r))) {
                    if (bad < 5) {
                        const size_t c = i % Cexp, px = i / Cexp;
                        printf("    MISMATCH %s crop %zu oy %zu ox %zu c %zu: got %g want %g\n", what, px / (size_t(Ho) * Ho),
                               (px / Ho) % Ho, px % Ho, c, g, r);
                    }
                    ++bad;
                }
                if (err > maxerr) maxerr = err;
            }
            std::vector<float> rp(size_t(NCHK) * ntl * chunks * RPse);
            int bad_se = 0;
            double max_se = 0;
            CK(hipMemcpy(rp.data(), d_rp, rp.size() * sizeof(float), hipMemcpyDeviceToHost));
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
                    for (int t = 0; t < ntl * chunks; ++t) gotv += rp[(size_t(b) * ntl * chunks + t) * RPse + jo];
                    const double err = std::fabs(want - gotv);
                    if (!(err <= 1e-5 + 2e-5 * mag)) ++bad_se;
                    max_se = std::max(max_se, err);
                }
            printf("  check %-34s: %d / %zu outputs off (max |err| %.2e), %d squeeze-excite partials off (max %.2e)\n", what, bad,
                   got.size(), maxerr, bad_se, max_se);
            return bad + bad_se;
        };

        Front2sArgs a2{};
        a2.weps = d_weps; a2.be = d_be; a2.bd = d_bd; a2.out = d_out; a2.rpart = d_rp;
        a2.w1t = se_in_front ? d_w1t : nullptr; a2.R = sh.R;
        a2.k = K; a2.s = S; a2.H = H; a2.Ho = Ho; a2.Cin = Cin; a2.Cexp = Cexp; a2.pad = pad; a2.KSe = KSe; a2.NTe = NTe;
        a2.wsi = wsi;
        auto set_form = [&](int tm, bool pre) {
            a2.tm = tm; a2.pre = pre;
            a2.x = pre ? static_cast<const void*>(d_xp) : static_cast<const void*>(d_x);
            a2.wdt = tm == 1 ? d_wdt1 : d_wdt2;
            a2.wsi_d = tm == 1 ? wsi_d1 : 1.0f;
        };
        FrontArgs a1{};
        a1.x = d_x; a1.wep = nullptr; a1.be = d_be; a1.wd = d_wd; a1.bd = d_bd; a1.out = d_out; a1.rpart = d_rp;
        a1.w1t = se_in_front ? d_w1t : nullptr; a1.R = sh.R;
        a1.k = K; a1.s = S; a1.H = H; a1.Ho = Ho; a1.Cin = Cin; a1.Cexp = Cexp; a1.pad = pad; a1.KSe = ceil_div(Cin, 8); a1.NTe = NTe;
        a1.split = true; a1.weps = d_weps; a1.KSes = KSe; a1.wsi = wsi;
        a1.plan = plan_front(WHENET_F32, K, S, H, Ho, Cexp);

        auto time2 = [&](const Front2Plan& pl, int n, int threads) -> float {
            a2.n = n;
            a2.plan = pl;
            a2.plan.threads = threads ? threads : pl.threads;
            for (int w = 0; w < 3; ++w) launch_front2s(a2, st);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) launch_front2s(a2, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };
        auto time1 = [&](int n) -> float {
            a1.n = n;
            a1.plan.threads = front_threads(a1.plan, n);
            for (int w = 0; w < 3; ++w) launch_front(a1, WHENET_F32, st);
            CK(hipStreamSynchronize(st));
            const int iters = (n >= 256) ? 12 : 40;
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < iters; ++w) launch_front(a1, WHENET_F32, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.f / iters;
        };

        int def_tm = 2;
        const Front2Plan def = plan_front2s(K, S, H, Ho, Cexp, &def_tm);
        printf("%s k%d s%d H%d Cin%d Cexp%d: plan CC=%d TH=%d TXG=%d tiles=%dx%d chunks=%d E=%dx%d RP=%d CP=%d lds=%zu tm=%d\n", sh.name, K, S,
               H, Cin, Cexp, def.CC, def.TH, def.TXG, def.tiles_x, def.tiles_y, def.chunks, def.EH, def.EWp, def.RP, def.CP, def.lds_bytes, def_tm);
        fflush(stdout);
        if (!nocheck) {
            {   // the reference kernel itself, against the same host restatement
                CK(hipMemset(d_out, 0xff, size_t(NCHK) * Ho * Ho * Cexp * sizeof(float)));
                CK(hipMemset(d_rp, 0xff, rp_floats * sizeof(float)));
                a1.n = NCHK; a1.plan.threads = 256;
                launch_front(a1, WHENET_F32, st);
                CK(hipStreamSynchronize(st));
                total_bad += check(a1.plan.ntiles(), a1.plan.chunks, "front.hip<float, split>");
            }
            for (int tm : {1, 2})
                for (int pre : {0, 1})
                    for (int threads : {256, 512}) {
                        CK(hipMemset(d_out, 0xff, size_t(NCHK) * Ho * Ho * Cexp * sizeof(float)));
                        CK(hipMemset(d_rp, 0xff, rp_floats * sizeof(float)));
                        set_form(tm, pre != 0);
                        a2.n = NCHK; a2.plan = def; a2.plan.threads = threads;
                        launch_front2s(a2, st);
                        CK(hipStreamSynchronize(st));
                        char what[96];
                        snprintf(what, sizeof what, "front2s tm=%d pre=%d, %d lanes", tm, pre, threads);
                        total_bad += check(def.ntiles(), def.chunks, what);
                    }
        }
#ifdef WHENET_STAMPS
        for (int tm : {1, 2}) {   // phase timeline of the default plan at 256 and 16 crops per launch: per-workgroup stamps (wave 0), averaged
            for (int n : {256, 16}) {
                const size_t nwg = size_t(n) * def.ntiles() * def.chunks;
                long long* d_st; CK(hipMalloc(&d_st, nwg * 8 * sizeof(long long)));
                CK(hipMemset(d_st, 0, nwg * 8 * sizeof(long long)));
                set_form(tm, true);
                a2.n = n; a2.plan = def; a2.plan.threads = 256;
                launch_front2s(a2, st);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(whenet_stamps), &d_st, sizeof(d_st)));
                launch_front2s(a2, st);
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
                printf("  timeline tm=%d n=%d (%zu workgroups, kernel span %.1f us): workgroup life %.2f us = prologue %.2f | expand %.2f | "
                       "barrier %.2f | taps+epilogue %.2f | barrier %.2f | tail %.2f\n", tm, n, nwg, double(t1 - t0) * 0.01,
                       life / nwg * 0.01, ph[0] / nwg * 0.01, ph[1] / nwg * 0.01, ph[2] / nwg * 0.01, ph[3] / nwg * 0.01,
                       ph[4] / nwg * 0.01, ph[5] / nwg * 0.01);
                CK(hipFree(d_st));
            }
        }
#endif
        const float o256 = time1(256), o64 = time1(64), o16 = time1(16);
        printf("  front.hip f32s        : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us\n", o256, o64, o16);
        sum_old += o64 * sh.mult;
        float best64 = 1e30f;
        for (int tm : {1, 2})
            for (int pre : {0, 1}) {
                set_form(tm, pre != 0);
                const float n256 = time2(def, 256, 0), n64 = time2(def, 64, 0), n16 = time2(def, 16, 0);
                const float n64w = time2(def, 64, 512);
                printf("  front2s tm=%d pre=%d     : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us   (x%.2f at 64; 512 lanes at 64: %.2f us)\n", tm, pre,
                       n256, n64, n16, o64 / n64, n64w);
                sum_new[tm - 1][pre] += n64 * sh.mult;
                if (pre) best64 = std::min(best64, n64);
            }
        sum_best += best64 * sh.mult;
        fflush(stdout);
        if (tune) {
            struct Row { Front2Plan p; int tm; float t256, t64, t16; int bad; };
            std::vector<Row> rows;
            const bool tune_pre = getenv("TUNE_PRE") != nullptr;
            for (int tm : {1, 2})
                for (const Front2Plan& pl : plan_front2s_candidates(K, S, Ho, Cexp)) {
                    Row r{pl, tm, 0, 0, 0, 0};
                    set_form(tm, tune_pre);
                    try {
                        if (!nocheck) {
                            CK(hipMemset(d_out, 0xff, size_t(NCHK) * Ho * Ho * Cexp * sizeof(float)));
                            CK(hipMemset(d_rp, 0xff, rp_floats * sizeof(float)));
                            a2.n = NCHK; a2.plan = pl; a2.plan.threads = 256;
                            launch_front2s(a2, st);
                            CK(hipStreamSynchronize(st));
                            char what[64];
                            snprintf(what, sizeof what, "tm=%d CC=%d TH=%d TXG=%d", tm, pl.CC, pl.TH, pl.TXG);
                            r.bad = check(pl.ntiles(), pl.chunks, what);
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
            for (size_t i = 0; i < rows.size() && i < 10; ++i)
                printf("   #%zu tm=%d CC=%3d TH=%2d TXG=%2d tiles=%dx%d chunks=%2d lds=%6zu : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us %s\n", i + 1,
                       rows[i].tm, rows[i].p.CC, rows[i].p.TH, rows[i].p.TXG, rows[i].p.tiles_x, rows[i].p.tiles_y, rows[i].p.chunks, rows[i].p.lds_bytes,
                       rows[i].t256, rows[i].t64, rows[i].t16, rows[i].bad ? "WRONG" : "");
            if (!rows.empty()) {
                char line[200];
                snprintf(line, sizeof line, "    {%d, %d, %d, %d, %d, %d, %d, %d, 256},   // %s: %.1f us @256, %.1f us @64, %.1f us @16\n", K, S, H, Cexp,
                         rows[0].p.CC, rows[0].p.TH, rows[0].p.TXG, rows[0].tm, sh.name, rows[0].t256, rows[0].t64, rows[0].t16);
                table += line;
            }
            fflush(stdout);
        }
        for (const void* q : {(const void*)d_x, (const void*)d_xp, (const void*)d_weps, (const void*)d_be, (const void*)d_wd, (const void*)d_wdt1,
                              (const void*)d_wdt2, (const void*)d_bd, (const void*)d_w1t, (const void*)d_out, (const void*)d_rp})
            CK(hipFree(const_cast<void*>(q)));
    }
    printf("\nsum over blocks 2-12 at 64 crops per launch (default plans): front.hip f32s %.1f us; front2s tm1 %.1f (pre %.1f), tm2 %.1f (pre %.1f), "
           "best-of pre %.1f us; %d mismatches in total\n", sum_old, sum_new[0][0], sum_new[0][1], sum_new[1][0], sum_new[1][1], sum_best, total_bad);
    if (tune) printf("\n// front2s_tuned.inc\n%s", table.c_str());
    return total_bad ? 2 : 0;
}
