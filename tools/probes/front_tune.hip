// Autotune probe: times EVERY feasible tile plan of the fused expand+depthwise kernel for each of
// EfficientNet-B0's layer shapes at 64 and 16 crops per launch (f16, random data) and prints the
// ranking plus the table that headposeestimation-whenet_amd/csrc/front_tuned_f16.inc carries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTUNE_F32] tools/probes/front_tune.hip -o tools/probes/front_tune
// -DTUNE_F32 (round 4): the same for the f32 instantiations (blocks 2-12 of the parity configuration run this kernel), table for
// front_tuned_f32.inc; TUNE_ONLY=b2 restricts the run to one shape.
#include "../../headposeestimation-whenet_amd/csrc/front.hip"

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

using namespace whenet;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Shape { const char* name; int k, s, H, Cin, Cexp, R; };
template <typename T> T* dalloc(size_t n, float scale) {
    std::vector<T> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = T(scale * (float(rand() % 2001) / 1000.f - 1.f));
    T* d; CK(hipMalloc(&d, n * sizeof(T)));
    CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
#ifdef TUNE_F32
using T = float;
constexpr int DT = WHENET_F32, KDEEP = 8, WV = 4, SZ = 4;
#else
using T = half_t;
constexpr int DT = WHENET_F16, KDEEP = 16, WV = 8, SZ = 2;
#endif
int main() {
    const Shape shapes[] = {{"b2", 3, 2, 112, 16, 96, 4},     {"b3", 3, 1, 56, 24, 144, 6},    {"b4", 5, 2, 56, 24, 144, 6},
                            {"b5", 5, 1, 28, 40, 240, 10},    {"b6", 3, 2, 28, 40, 240, 10},   {"b7", 3, 1, 14, 80, 480, 20},
                            {"b9", 5, 1, 14, 80, 480, 20},    {"b10", 5, 1, 14, 112, 672, 28}, {"b12", 5, 2, 14, 112, 672, 28},
                            {"b13", 5, 1, 7, 192, 1152, 48},  {"b16", 3, 1, 7, 192, 1152, 48}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NMAX = 256;
    std::string table;
    const char* only = getenv("TUNE_ONLY");
    for (const Shape& sh : shapes) {
        if (only && std::string(only) != sh.name) continue;
        const int Ho = ceil_div(sh.H, sh.s);
        const int padt = std::max((Ho - 1) * sh.s + sh.k - sh.H, 0);
        FrontArgs a{};
        a.k = sh.k; a.s = sh.s; a.H = sh.H; a.Ho = Ho; a.Cin = sh.Cin; a.Cexp = sh.Cexp; a.pad = padt / 2;
        a.KSe = ceil_div(sh.Cin, KDEEP); a.NTe = ceil_div(sh.Cexp, 32);
        a.x = dalloc<T>(size_t(NMAX) * sh.H * sh.H * sh.Cin, 1.f);
        a.wep = dalloc<T>(size_t(a.KSe) * a.NTe * 64 * WV, 0.05f);
        a.be = dalloc<float>(a.NTe * 32, 0.1f);
        a.wd = dalloc<float>(size_t(sh.k) * sh.k * sh.Cexp, 0.1f);
        a.bd = dalloc<float>(sh.Cexp, 0.1f);
        a.out = dalloc<T>(size_t(NMAX) * Ho * Ho * sh.Cexp, 0.f);
        a.R = sh.R;
        const float* w1_all = dalloc<float>(size_t(a.R) * sh.Cexp, 0.05f);
        a.w1t = (DT == WHENET_F32 || sh.Cexp >= 480) ? w1_all : nullptr;    // (f32: as the engine runs it in round 4)
        std::vector<double> scores;
        const std::vector<FrontPlan> cand = plan_front_candidates(DT, sh.k, sh.s, sh.H, Ho, sh.Cexp, &scores);
        size_t max_tiles = 1;
        for (const FrontPlan& p : cand) max_tiles = std::max(max_tiles, size_t(p.ntiles()));
        float* rp = nullptr;
        // [tiles][chunks][RP] partial vectors (up to Cexp/32 chunks x (R+3) values) or [tiles][Cexp] channel sums
        CK(hipMalloc(&rp, size_t(NMAX) * max_tiles * std::max(size_t(sh.Cexp + 64), size_t(sh.Cexp / 32 + 1) * (sh.R + 4)) * sizeof(float)));
        a.rpart = rp;
        struct Row { FrontPlan p; double score; float t64, t16, t256; };
        std::vector<Row> rows;
        for (size_t i = 0; i < cand.size(); ++i) {
            Row r{cand[i], scores[i], 0.f, 0.f, 0.f};
            if (getenv("TUNE_VERBOSE")) { printf("  cand %zu CC=%d TH=%d NSX=%d EP=%d lds=%zu\n", i, cand[i].CC, cand[i].TH, cand[i].NSX, cand[i].EP, cand[i].lds_bytes); fflush(stdout); }
            for (int n : {256, 64, 16}) {
                a.n = n;
                a.plan = cand[i];
                a.plan.threads = front_threads(a.plan, n);
                for (int w = 0; w < 3; ++w) launch_front(a, DT, s);
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                const int iters = (n == 256) ? 12 : 40;
                for (int w = 0; w < iters; ++w) launch_front(a, DT, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                (n == 256 ? r.t256 : (n == 64 ? r.t64 : r.t16)) = ms * 1000.f / iters;
            }
            rows.push_back(r);
        }
        // figure of merit: the in-flight schedule overlaps several 64-crop launches (throughput regime: the
        // 256-crop time per 64 crops), one forward at a time runs 64 crops as ~21-crop lanes, small batches run 1-16
        auto merit = [](const Row& r) { return r.t256 / 4.0f + r.t64 + r.t16; };
        std::sort(rows.begin(), rows.end(), [&](const Row& x, const Row& y) { return merit(x) < merit(y); });
        size_t by_score = 0;
        for (size_t i = 1; i < rows.size(); ++i)
            if (rows[i].score > rows[by_score].score) by_score = i;
        printf("%s k%d s%d H%d Cexp%d: %zu candidates; a-priori pick is rank %zu (%.1f / %.1f us)\n", sh.name, sh.k, sh.s, sh.H, sh.Cexp,
               rows.size(), by_score + 1, rows[by_score].t64, rows[by_score].t16);
        for (size_t i = 0; i < rows.size() && i < 8; ++i)
            printf("   #%zu CC=%3d TH=%2d NSX=%d tiles=%dx%d chunks=%2d lds=%5zu EP=%3d score %.3f : n=256 %7.2f us  n=64 %7.2f us  n=16 %6.2f us\n", i + 1,
                   rows[i].p.CC, rows[i].p.TH, rows[i].p.NSX, rows[i].p.tiles_x, rows[i].p.tiles_y, rows[i].p.chunks, rows[i].p.lds_bytes,
                   rows[i].p.EP, rows[i].score, rows[i].t256, rows[i].t64, rows[i].t16);
        {
            const FrontPlan cur = plan_front(DT, sh.k, sh.s, sh.H, Ho, sh.Cexp);
            for (size_t i = 0; i < rows.size(); ++i)
                if (rows[i].p.CC == cur.CC && rows[i].p.TH == cur.TH && rows[i].p.NSX == cur.NSX && rows[i].p.EP == cur.EP)
                    printf("   current plan is rank %zu: CC=%d TH=%d NSX=%d EP=%d : n=256 %7.2f  n=64 %7.2f  n=16 %6.2f\n", i + 1, cur.CC,
                           cur.TH, cur.NSX, cur.EP, rows[i].t256, rows[i].t64, rows[i].t16);
        }
        char line[200];
        snprintf(line, sizeof line, "    {%d, %d, %d, %d, %d, %d, %d, %d},   // %s: %.1f us @256, %.1f us @64, %.1f us @16\n", sh.k, sh.s, sh.H, sh.Cexp,
                 rows[0].p.CC, rows[0].p.TH, rows[0].p.NSX, rows[0].p.EP - rows[0].p.CC * SZ, sh.name, rows[0].t256, rows[0].t64, rows[0].t16);
        table += line;
        for (const void* q : {a.x, a.wep, (const void*)a.be, (const void*)a.wd, (const void*)a.bd, (const void*)a.out,
                              (const void*)a.rpart, (const void*)w1_all})
            CK(hipFree(const_cast<void*>(q)));
    }
    printf("\n// front_tuned_%s.inc\n%s", DT == WHENET_F32 ? "f32" : "f16", table.c_str());
    return 0;
}
