// MBConv "front" of the 7x7 blocks (13-16), f16, round 4: expand 1x1 conv + BN + Swish -> depthwise kxk conv (stride 1)
// + BN + Swish in ONE kernel whose workgroup owns a GROUP of G crops x a chunk of CC expanded channels.
//
// Reference: efficientnet 0.0.4 MBConvBlock, blocks 13..16 (/root/reference/whenet.py:8):
// Conv2D(192 -> 1152, 1x1, no bias) -> BN -> Swish -> DepthwiseConv2D(k = 5 | 3, stride 1, 'same') -> BN -> Swish on a
// 7 x 7 map.
//
// Why a kernel of its own (round-3 review, item 2).  One crop of these layers is 49 pixels: 1.5 MFMA strips.  front.hip /
// front2.hip give a workgroup one crop x one channel chunk, so every workgroup re-pulls the chunk's expand weights
// (24.6 KB), tap operands and squeeze-excite slice for 49 rows of work and is a chain of ~10 dependent memory round trips:
// 23 us per launch of 64 crops at 0.04 of HBM peak, 2.4x its algorithmic traffic (profiles/r03).  Here:
//   * a workgroup's rows are the 7 image rows x 8 pixel slots (7 pixels + 1) of G crops: with G = 4 that is 28 rows x 8
//     = 224 MFMA rows = exactly 7 strips of 32 -- the chunk's weights, taps and reduce-conv slice are fetched once per 4
//     crops, and the ~2 us fixed cost of a workgroup is amortised over 4x the work;
//   * the chunk's expand weights are staged ONCE in LDS (fragment order, lane-linear ds_read_b128) and every wave loads
//     the 12 k-steps of ITS strip's pixel rows in one round trip: the expand phase has one exposed global round trip;
//   * the tile holds the IMAGE only -- no halo.  'SAME' zeros are never materialised: out-of-image input rows are skipped
//     at compile time in the tap loop (29 instead of 55 (row, ky) products for 5x5), out-of-image 4-pixel chunks are never
//     multiplied (2 of the 3 Toeplitz chunks per product), and the one pad pixel per row (x = 7) is stored as zero by the
//     lane that computed it;
//   * LDS tile layout  E[crop quad][16-channel block][row][unit(channel, crop)][8 pixels] f16, 16 bytes per unit, where
//     unit = j * 16 + (cl ^ (4 j)) for crop j of the quad and channel cl of the block: the taps' ds_read_b128 (lane =
//     channel x crop) is bank-conflict free and the expand's ds_write_b64 (lane = channel) is 2-way at worst;
//   * depthwise taps: per-channel Toeplitz products on v_mfma_f32_4x4x4_16B_f16 exactly as front2.hip (block = channel,
//     column = one 4-pixel output group x 7 rows), but a quad of columns is the SAME x-group of 4 CROPS, so the x-group
//     (hence which Toeplitz chunks exist) is uniform per item.
// A crop's rows go through the same MFMA sequence whoever its neighbours in the group are (MFMA rows and 4x4x4 columns
// are independent), and its squeeze-excite sums are taken per crop in a fixed order: results are bitwise independent of
// the batch size and of the crop's position (tail groups are predicated).
//
// HBM bytes per crop: 49 * Cin * 2 (x chunks, L2 hits) + 49 * Cexp * 2 written once.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

#include <atomic>
#include <string>
#include <type_traits>

namespace whenet {

namespace {

constexpr int HW7 = 7;                 // the map is 7 x 7

struct F7Params {
    const void* x;                     // [n][49][Cin] T
    const void* wep;                   // packed expand weights (MFMA fragment order, snapshot.h)
    const float* be;                   // [Cexp]
    const void* wdt;                   // f16: pack_dw_toeplitz(w, k, 1, C, xs = 4 - k / 2) image, 3 chunks per (block, ky);
                                       // f32: the depthwise kernel itself, [k*k][Cexp] floats
    const float* bd;                   // [Cexp]
    void* out;                         // [n][49][Cexp] T
    float* rpart;                      // [n][chunks][RPse]: this workgroup's share of the SE reduce conv, per crop
    const float* w1t;                  // [R][Cexp]
    int n, Cin, Cexp, NTe, R, RPse;
    int off_w, off_stage, off_red, off_sum;
    float wsi;                         // WHENET_F32S: 2^-shift of the scaled split weights (wep is then the [hi | lo] image pair)
    int chunks, ngroups, xcd;          // launch geometry: a 1-D grid of chunks * ngroups workgroups (xcd_unit())
};

// K: depthwise kernel size (3 | 5); KS: k-steps of the expand contraction (Cin / 16); G: crops per workgroup; CC: expanded
// channels per workgroup; NTHR: lanes per workgroup.  (All compile-time: the index arithmetic is shifts and constants.)
template <int K, int KS, int G, int CC, int NTHR>
__global__ __launch_bounds__(NTHR) void whenet_front7_kernel(const F7Params p) {
    constexpr int NWAVE = NTHR / 64;
    constexpr int PAD = K / 2;
    constexpr int ROWS = HW7;
    constexpr int NT = CC / 32, NCB = CC / 16;
    constexpr int nrow = G * ROWS;                             // image rows of the group
    constexpr int nstrip = (nrow + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;
    const half8* Wl = reinterpret_cast<const half8*>(smem + p.off_w);          // [KS][NT][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    const int Cin = p.Cin;
    int grp, chunk;
    xcd_unit(int(blockIdx.x), p.ngroups, p.chunks, grp, chunk, p.xcd != 0);    // (device_math.h)
    const int c0 = chunk * CC;
    const int crop0 = grp * G;                                 // first crop of this group
    const int nlast = p.n - 1;

    STAMP(0);
    // ---- prologue: the chunk's expand weights -> LDS; this wave's first strip of pixel rows -> registers --------------
    constexpr int nwv = KS * NT * 64;
    constexpr int WV = (nwv + NTHR - 1) / NTHR;               // 16-byte vectors per lane
    half8 wstage[WV];
    {
        const half8* src = reinterpret_cast<const half8*>(p.wep);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) {                                     // v = (ks * NT + tile) * 64 + lane'
                const int ks = v / (NT * 64), r = v - ks * (NT * 64);
                wstage[i] = src[(size_t(ks) * p.NTe + (c0 >> 5)) * 64 + r];
            }
        }
    }
    // operand side: MFMA row lm = image row (lm >> 3) of the strip, pixel slot lm & 7 (slot 7 = pad: any valid address)
    auto a_offset = [&](int strip) -> unsigned {
        int R = strip * 4 + (lm >> 3);
        R = R < nrow ? R : nrow - 1;
        const int cr = (R * 37) >> 8;                          // R / 7 for R < 64
        const int row = R - cr * 7;
        int gc = crop0 + cr;
        gc = gc < nlast ? gc : nlast;
        int px = lm & 7;
        px = px < 7 ? px : 6;
        return unsigned((gc * 49 + row * 7 + px) * Cin + g * 8) * 2u;
    };
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
    half8 a[KS];
    // few strips (groups of 2 crops: 4 strips x NT tiles <= the waves): one (strip, tile) task per wave instead of one strip
    // per wave over all tiles -- at small launches the expand phase is the workgroup's critical path
    constexpr bool SPLIT = nstrip * NT <= NWAVE;
    int strip = SPLIT ? (wave < nstrip * NT ? wave % nstrip : nstrip) : wave;
    const int t_lo = SPLIT ? wave / nstrip : 0, t_hi = SPLIT ? t_lo + 1 : NT;
    // Round 5: a strip's 28 pixels are 28 x Cin x 2 = 10,752 CONTIGUOUS bytes.  Fetched as MFMA fragments (16 bytes of each of 28
    // pixel rows per lane group) every wave-instruction touches 28 cache lines per KB and a CU pulls ~27 GB/s; fetched as 8 pixels x
    // 128 contiguous bytes per instruction it pulls ~56 GB/s (tools/probes/fetch_pattern_probe.hip) -- and the activation rows are the
    // largest thing this workgroup fetches (75 KB of ~130).  So the rows arrive coalesced, 128-byte column groups at a time, and are
    // turned into fragments through the wave's own 4 KB of the (still unused) tile region: pitch 144 B, conflict-free both ways,
    // DS operations of a wave execute in order, no barrier.  Pure data movement: the fragments, hence the bits, are the same.
    // (Groups of 4 crops: 7 strips on 8 waves, 7 x 4,032 B <= the tile region.  Smaller groups keep the direct loads.)
    constexpr bool STG = !SPLIT && nstrip <= NWAVE && nstrip * 4032 <= (G + 3) / 4 * NCB * ROWS * 64 * 16 && KS % 4 == 0;
    if (strip < nstrip) {
        if constexpr (STG) {
            constexpr int PX = 28, SEG = KS / 4;                 // pixels of a strip; 128-byte column groups of a pixel row
            constexpr int rowb = KS * 32;                        // bytes of a pixel row
            unsigned char* sa = E + wave * 4032;
            const size_t last = size_t(p.n) * 49 * rowb - 16;    // (tail groups: any valid address; those crops are never stored)
            const size_t base = (size_t(crop0) * 49 + size_t(strip) * PX) * rowb;
            half8 raw[SEG][4];
#pragma unroll
            for (int t = 0; t < SEG; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int idx = i * 64 + lane;
                    idx = idx < PX * 8 ? idx : PX * 8 - 1;
                    size_t off = base + size_t(idx >> 3) * rowb + t * 128 + (idx & 7) * 16;
                    off = off < last ? off : last;
                    raw[t][i] = *reinterpret_cast<const half8*>(xb + off);
                }
            int pf = (lm >> 3) * 7 + ((lm & 7) < 7 ? (lm & 7) : 6);      // this lane's pixel of the strip (slot 7: any valid pixel)
            const int rlim = nrow - strip * 4;                           // image rows of the group left in this strip
            if ((lm >> 3) >= rlim) pf = (rlim - 1) * 7 + 6;              // (rows past the group: as a_offset clamps them)
#pragma unroll
            for (int t = 0; t < SEG; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = i * 64 + lane;
                    if (idx < PX * 8) *reinterpret_cast<half8*>(sa + (idx >> 3) * 144 + (idx & 7) * 16) = raw[t][i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) a[t * 4 + u] = *reinterpret_cast<const half8*>(sa + pf * 144 + u * 32 + g * 16);
            }
        } else {
            const unsigned off = a_offset(strip);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const half8*>(xb + off + ks * 32);
        }
    }
    float bias_t[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_t[t] = p.be[c0 + t * 32 + lm];
    {
        half8* dst = reinterpret_cast<half8*>(smem + p.off_w);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) dst[v] = wstage[i];
        }
    }
    lds_barrier();
    STAMP(1);

    // ---- expand: strips of 4 image rows x 8 pixel slots; every wave runs its strips over all channel tiles -------------
    for (; strip < nstrip; strip += NWAVE) {
        const bool more = strip + NWAVE < nstrip;              // (uniform)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < t_lo || t >= t_hi) continue;               // (uniform)
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias_t[t];   // (BN bias as the accumulators' initial value)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], Wl[(ks * NT + t) * 64 + lane], acc, 0, 0, 0);
            const int ch = t * 32 + lm, cb = ch >> 4, cl = ch & 15;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int R = strip * 4 + qq;                  // (uniform) image row of the group
                if (R < nrow) {
                    const int cr = (R * 37) >> 8, row = R - cr * 7;
                    const int jq = cr >> 2, j = cr & 3;
                    const float2v y0 = swish2(float2v{acc[4 * qq], acc[4 * qq + 1]});
                    const float2v y1 = swish2(float2v{acc[4 * qq + 2], acc[4 * qq + 3]});
                    half4 o;
                    o[0] = half_t(y0[0]);
                    o[1] = half_t(y0[1]);
                    o[2] = half_t(y1[0]);
                    o[3] = g ? half_t(0) : half_t(y1[1]);       // pixel slot 7 is 'SAME' padding of the EXPANDED tensor
                    unsigned char* ep = E + ((((jq * NCB + cb) * ROWS + row) * 64 + j * 16 + (cl ^ (j << 2))) << 4) + g * 8;
                    *reinterpret_cast<half4*>(ep) = o;
                }
            }
        }
        if (more) {                                            // (only plans with more strips than waves get here)
            const unsigned off = a_offset(strip + NWAVE);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const half8*>(xb + off + ks * 32);
        }
    }
    STAMP(2);

    // ---- depthwise taps: items (crop quad, 16-channel block, x-group); lane = channel cl x crop j of the quad ----------
    constexpr int NJQ = (G + 3) / 4;
    constexpr int nitem = NJQ * NCB * 2;
    const int cl = lane >> 2, j = lane & 3;
    half4 A[K][2];
    float bdv = 0.f;
    auto load_taps = [&](int it) {
        const int xgl = it & 1, cb = (it >> 1) % NCB;
        // chunk `rel` of the Toeplitz image multiplies input pixels 4 (xgl + rel - 1) ..+3: rel 1, 2 for x-group 0, rel 0, 1
        // for x-group 1 (the third chunk lies outside the image)
        const half4* src = reinterpret_cast<const half4*>(p.wdt) + (size_t((c0 >> 4) + cb) * K * 3 + (1 - xgl)) * 64 + lane;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            A[ky][0] = src[(ky * 3) * 64];
            A[ky][1] = src[(ky * 3 + 1) * 64];
        }
        bdv = p.bd[c0 + cb * 16 + cl];
    };
    int it = wave;
    if (it < nitem) load_taps(it);                            // in flight across the barrier
    // this lane's reduce-kernel values (squeeze-excite half below): 4 lanes per output jo, lane q takes the CC / 4
    // CONTIGUOUS channels q * CC / 4 .. (16-byte loads: the strided form is 32 scattered dword loads per lane, 16 cache
    // lines per wave-instruction -- it alone cost ~2 us of the workgroup's life)
    constexpr int W1Q = CC / 16;                               // float4 per lane
    float4v w1v[W1Q];
    {
        const int jo = (tid >> 2) & 63, q = tid & 3;
        const float* wrow = p.w1t + size_t(jo < p.R ? jo : p.R - 1) * p.Cexp + c0 + q * (CC / 4);
#pragma unroll
        for (int i = 0; i < W1Q; ++i) w1v[i] = *reinterpret_cast<const float4v*>(wrow + 4 * i);
    }
    lds_barrier();
    STAMP(3);

    unsigned char* stg = smem + p.off_stage + wave * 2048;
    float* s_red = reinterpret_cast<float*>(smem + p.off_red);          // [G][2][CC]
    float* s_sum = reinterpret_cast<float*>(smem + p.off_sum);          // [G][CC]
    const int unit = j * 16 + (cl ^ (j << 2));
    // piece coordinates of this lane in the output stage: slot = (row * 4 + i) * 4 + j, two 16-byte halves per slot
    const int jP = (lane >> 1) & 3, iP = (lane >> 3) & 3, hP = lane & 1, rP = lane >> 5;
    unsigned char* outb = reinterpret_cast<unsigned char*>(p.out);
    const unsigned row_bytes = unsigned(HW7) * unsigned(p.Cexp) * 2u;

    for (; it < nitem; it += NWAVE) {
        const int xgl = it & 1, cbq = it >> 1, cb = cbq % NCB, jq = cbq / NCB;       // (uniform)
        const unsigned char* bp = E + ((((jq * NCB + cb) * ROWS) * 64 + unit) << 4);
        float4v acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = float4v{bdv, bdv, bdv, bdv};        // (BN bias as the initial value)
#pragma unroll
        for (int er = 0; er < ROWS; ++er) {
            const half8 v = *reinterpret_cast<const half8*>(bp + er * 1024);
            const half4 b0 = {v[0], v[1], v[2], v[3]}, b1 = {v[4], v[5], v[6], v[7]};
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int d = er - ky + PAD;                   // output row fed by input row er through tap row ky
                if (d >= 0 && d < ROWS) {
                    acc[d] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[ky][0], b0, acc[d], 0, 0, 0);
                    acc[d] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[ky][1], b1, acc[d], 0, 0, 0);
                }
            }
        }
        if (it + NWAVE < nitem) load_taps(it + NWAVE);         // the next item's taps travel during the epilogue
        // ---- BN + Swish, per-crop channel sums, and the way out: lane (channel cl, crop j) holds 7 rows x 4 pixels -----
        const int cr = jq * 4 + j;                             // crop of the group
        const bool okc = cr < G && crop0 + cr < p.n;
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (okc && 4 * xgl + i < HW7) ? 1.f : 0.f;
        const int crP = jq * 4 + jP, oxP = 4 * xgl + iP;
        const bool okP = crP < G && crop0 + crP < p.n && oxP < HW7;
        const unsigned obase = (unsigned(crop0 + crP) * 49u + unsigned(rP) * 7u + unsigned(oxP)) * unsigned(p.Cexp) * 2u +
                               unsigned(c0 + cb * 16 + hP * 8) * 2u;
        float2v sum2 = {0.f, 0.f};
        const float2v m01 = {m[0], m[1]}, m23 = {m[2], m[3]};
        unsigned char* sw = stg + j * 32 + cl * 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                 // rows 0..3, then rows 4..6, through the 2 KB stage
            const int r0 = half * 4, nr = half ? 3 : 4;
#pragma unroll
            for (int r = 0; r < nr; ++r) {
                const float2v y01 = swish2(float2v{acc[r0 + r][0], acc[r0 + r][1]});
                const float2v y23 = swish2(float2v{acc[r0 + r][2], acc[r0 + r][3]});
                sum2 = y01 * m01 + sum2;
                sum2 = y23 * m23 + sum2;
                *reinterpret_cast<half_t*>(sw + (r * 4 + 0) * 128) = half_t(y01[0]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 1) * 128) = half_t(y01[1]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 2) * 128) = half_t(y23[0]);
                *reinterpret_cast<half_t*>(sw + (r * 4 + 3) * 128) = half_t(y23[1]);
            }
            wave_lds_sync();
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int rl = 2 * pp + rP;                    // stage row of this lane's piece
                const half8 v = *reinterpret_cast<const half8*>(stg + (pp * 64 + lane) * 16);
                if (okP && rl < nr) *reinterpret_cast<half8*>(outb + obase + unsigned(r0 + 2 * pp) * row_bytes) = v;
            }
            wave_lds_sync();
        }
        if (cr < G) s_red[(cr * 2 + xgl) * CC + cb * 16 + cl] = sum2[0] + sum2[1];
    }
    STAMP(4);
    lds_barrier();
    STAMP(5);

    // ---- squeeze-excite, first half: per crop, this chunk's share of the reduce conv (fixed order) ---------------------
#pragma unroll
    for (int i0 = 0; i0 < G * CC; i0 += NTHR) {
        const int i = i0 + tid;
        if (i < G * CC) {
            const int cr = i / CC, c = i % CC;                  // (compile-time powers of two)
            s_sum[i] = s_red[(cr * 2) * CC + c] + s_red[(cr * 2 + 1) * CC + c];
        }
    }
    lds_barrier();
    {
        // 4 lanes per output jo: lane q sums its CC / 4 channels in order; combined (a0+a1)+(a2+a3); 64 outputs x
        // (NTHR / 256) crops per pass
        const int jo = (tid >> 2) & 63, q = tid & 3;
        for (int cr = tid >> 8; cr < G; cr += NTHR / 256) {
            float accr = 0.0f;
            if (jo < p.R) {
                const float* sp = s_sum + cr * CC + q * (CC / 4);
#pragma unroll
                for (int i = 0; i < W1Q; ++i) {
                    const float4v sv = *reinterpret_cast<const float4v*>(sp + 4 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accr = fmaf(sv[e], w1v[i][e], accr);
                }
            }
            const float pair = accr + quad_xor1(accr);
            const float tot = pair + quad_xor2(pair);
            if (q == 0 && jo < p.RPse && crop0 + cr < p.n)
                p.rpart[(size_t(crop0 + cr) * p.chunks + chunk) * p.RPse + jo] = (jo < p.R) ? tot : 0.0f;
        }
    }
    STAMP(6);
}

// ---- the f32 (parity) configuration of the same kernel --------------------------------------------------------------------
// Same decomposition; what differs is the arithmetic the f32 path owes the 1e-3 degree bar:
//   * expand on v_mfma_f32_32x32x2_f32 (exact f32, an fmaf chain in k order -- the same instruction sequence as pw.hip /
//     front.hip with the operand roles swapped, so the expanded values have the same bits), 24 k-steps of 8: at 64 FLOP per
//     cycle and SIMD this phase IS the kernel (2 tasks of 96 MFMAs per wave, two waves per SIMD: ~10 us per workgroup) --
//     the f32 kernel is matrix-pipe bound, not latency bound;
//   * the tile holds f32 (32 bytes per unit); the depthwise taps are f32 FMAs on the VALU in dw.hip's order (ky ascending,
//     kx ascending; out-of-image taps skipped -- adding their zeros is the identity), so the depthwise output has the bits
//     of front.hip's and dw.hip's;
//   * outputs are stored directly (64 contiguous bytes per pixel and 16-channel block).
// KS8 = Cin / 8 k-steps.
// SP (WHENET_F32S, round 5): the expand products as binary16 hi/lo pairs on the f16 matrix cores (device_math.h PwOps<float, true>):
// KS8 / 2 k-steps of 16, three v_mfma_f32_32x32x16_f16 each; the staged weights are [k-step][tile][hi | lo][64 lanes] half8 -- the
// same bytes; the strip's rows are split ONCE when they arrive and serve every channel tile.  Taps, sums and stores unchanged.
template <int K, int KS8, int G, int CC, int NTHR, bool SP = false>
__global__ __launch_bounds__(NTHR) void whenet_front7_f32_kernel(const F7Params p) {
    constexpr int NWAVE = NTHR / 64;
    constexpr int PAD = K / 2;
    constexpr int ROWS = HW7;
    constexpr int NT = CC / 32, NCB = CC / 16;
    constexpr int nrow = G * ROWS;
    constexpr int nstrip = (nrow + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;                                                   // units of 32 bytes: 8 pixel slots f32
    const float4v* Wl = reinterpret_cast<const float4v*>(smem + p.off_w);      // [KS8][NT][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    const int Cin = p.Cin;
    int grp, chunk;
    xcd_unit(int(blockIdx.x), p.ngroups, p.chunks, grp, chunk, p.xcd != 0);
    const int c0 = chunk * CC;
    const int crop0 = grp * G;
    const int nlast = p.n - 1;

    // ---- prologue: expand weights -> LDS; this wave's strip of pixel rows -> registers (all 24 k-steps: one round trip) --
    constexpr int nwv = KS8 * NT * 64;
    constexpr int WV = (nwv + NTHR - 1) / NTHR;
    constexpr int KS16 = KS8 / 2;
    float4v wstage[WV];
    {
        const float4v* src = reinterpret_cast<const float4v*>(p.wep);
        const size_t w_lo = size_t(KS16) * p.NTe * 64;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) {
                if constexpr (SP) {                            // v = ((ks * NT + t) * 2 + h) * 64 + lane
                    const int l = v & 63, h = (v >> 6) & 1, kt = v >> 7, ks = kt / NT, t = kt % NT;
                    wstage[i] = src[(h ? w_lo : 0) + (size_t(ks) * p.NTe + (c0 >> 5) + t) * 64 + l];
                } else {
                    const int ks = v / (NT * 64), r = v % (NT * 64);
                    wstage[i] = src[(size_t(ks) * p.NTe + (c0 >> 5)) * 64 + r];
                }
            }
        }
    }
    auto a_offset = [&](int strip) -> unsigned {
        int R = strip * 4 + (lm >> 3);
        R = R < nrow ? R : nrow - 1;
        const int cr = (R * 37) >> 8;                          // R / 7 for R < 64
        const int row = R - cr * 7;
        int gc = crop0 + cr;
        gc = gc < nlast ? gc : nlast;
        int px = lm & 7;
        px = px < 7 ? px : 6;
        return unsigned((gc * 49 + row * 7 + px) * Cin + g * (SP ? 8 : 4)) * 4u;
    };
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
    using OPS = PwOps<float, true>;
    float4v a[SP ? 1 : KS8];
    OPS::P ap[SP ? KS16 : 1];                                  // SP: the strip's rows as binary16 hi / lo fragments
    auto load_rows = [&](unsigned off) {
        if constexpr (SP) {
            OPS::A raw[KS16];
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks) raw[ks] = OPS::load_a(reinterpret_cast<const float*>(xb + off + ks * 64));
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks) ap[ks] = OPS::prep(raw[ks]);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS8; ++ks) a[ks] = *reinterpret_cast<const float4v*>(xb + off + ks * 32);
        }
    };
    constexpr bool SPLIT = nstrip * NT <= NWAVE;               // (as the f16 kernel: one (strip, tile) task per wave)
    int strip = SPLIT ? (wave < nstrip * NT ? wave % nstrip : nstrip) : wave;
    const int t_lo = SPLIT ? wave / nstrip : 0, t_hi = SPLIT ? t_lo + 1 : NT;
    // Round 5, as the f16 kernel: a strip's 28 pixels are 28 x Cin x 4 contiguous bytes; they arrive as 8 pixels x 128 bytes per
    // wave-instruction and become fragments through the wave's own 4 KB of the still unused tile region (pitch 144 B).  Same fragments.
    constexpr bool STG = !SPLIT && nstrip <= NWAVE && KS8 % 4 == 0;
    if (strip < nstrip) {
        if constexpr (STG) {
            constexpr int PX = 28, SEG = KS8 / 4;                // 128-byte column groups of a pixel row (4 k-steps of 8 each)
            constexpr int rowb = KS8 * 32;
            unsigned char* sa = E + wave * 4032;
            const size_t last = size_t(p.n) * 49 * rowb - 16;
            const size_t base = (size_t(crop0) * 49 + size_t(strip) * PX) * rowb;
            int pf = (lm >> 3) * 7 + ((lm & 7) < 7 ? (lm & 7) : 6);
            const int rlim = nrow - strip * 4;
            if ((lm >> 3) >= rlim) pf = (rlim - 1) * 7 + 6;
            OPS::A rawsp[SP ? KS16 : 1];
#pragma unroll
            for (int t0 = 0; t0 < SEG; t0 += 3) {                // three column groups in flight (12 registers)
                float4v raw[3][4];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int idx = i * 64 + lane;
                        idx = idx < PX * 8 ? idx : PX * 8 - 1;
                        size_t off = base + size_t(idx >> 3) * rowb + (t0 + tt) * 128 + (idx & 7) * 16;
                        off = off < last ? off : last;
                        raw[tt][i] = *reinterpret_cast<const float4v*>(xb + off);
                    }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    const int t = t0 + tt;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = i * 64 + lane;
                        if (idx < PX * 8) *reinterpret_cast<float4v*>(sa + (idx >> 3) * 144 + (idx & 7) * 16) = raw[tt][i];
                    }
                    if constexpr (SP) {                          // a 128-byte group = 2 k-steps of 16: 32 B per lane and k-step
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            rawsp[t * 2 + u].x0 = *reinterpret_cast<const float4v*>(sa + pf * 144 + u * 64 + g * 32);
                            rawsp[t * 2 + u].x1 = *reinterpret_cast<const float4v*>(sa + pf * 144 + u * 64 + g * 32 + 16);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) a[t * 4 + u] = *reinterpret_cast<const float4v*>(sa + pf * 144 + u * 32 + g * 16);
                    }
                }
            }
            if constexpr (SP) {
#pragma unroll
                for (int ks = 0; ks < KS16; ++ks) ap[ks] = OPS::prep(rawsp[ks]);
            }
        } else {
            load_rows(a_offset(strip));
        }
    }
    float bias_t[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_t[t] = p.be[c0 + t * 32 + lm];
    {
        float4v* dst = reinterpret_cast<float4v*>(smem + p.off_w);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < nwv) dst[v] = wstage[i];
        }
    }
    lds_barrier();

    // ---- expand ----------------------------------------------------------------------------------------------------------
    for (; strip < nstrip; strip += NWAVE) {
        const bool more = strip + NWAVE < nstrip;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < t_lo || t >= t_hi) continue;               // (uniform)
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if constexpr (SP) {
#pragma unroll
                for (int ks = 0; ks < KS16; ++ks) {
                    const half8 whi = __builtin_bit_cast(half8, Wl[((ks * NT + t) * 2) * 64 + lane]);
                    const half8 wlo = __builtin_bit_cast(half8, Wl[((ks * NT + t) * 2 + 1) * 64 + lane]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[ks].hi, wlo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[ks].lo, whi, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[ks].hi, whi, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS8; ++ks) {
                    const float4v w = Wl[(ks * NT + t) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][u], w[u], acc, 0, 0, 0);
                }
            }
            const float bias = bias_t[t];
            const int ch = t * 32 + lm, cb = ch >> 4, cl = ch & 15;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int R = strip * 4 + qq;
                if (R < nrow) {
                    const int cr = (R * 37) >> 8, row = R - cr * 7;
                    const int jq = cr >> 2, j = cr & 3;
                    float4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = conv_swish<float>(SP ? fmaf(acc[4 * qq + r], p.wsi, bias) : acc[4 * qq + r] + bias);
                    if (g) o[3] = 0.f;                          // pixel slot 7 is 'SAME' padding of the EXPANDED tensor
                    unsigned char* ep = E + ((((jq * NCB + cb) * ROWS + row) * 64 + j * 16 + (cl ^ (j << 2))) << 5) + g * 16;
                    *reinterpret_cast<float4v*>(ep) = o;
                }
            }
        }
        if (more) load_rows(a_offset(strip + NWAVE));
    }

    // ---- depthwise taps on the VALU: items (crop quad, 16-channel block, x-group); lane = channel cl x crop j ------------
    constexpr int NJQ = (G + 3) / 4;
    constexpr int nitem = NJQ * NCB * 2;
    const int cl = lane >> 2, j = lane & 3;
    float wt[K * K];
    float bdv = 0.f;
    auto load_taps = [&](int it) {
        const int cb = (it >> 1) % NCB;
        const float* src = reinterpret_cast<const float*>(p.wdt) + c0 + cb * 16 + cl;
#pragma unroll
        for (int q = 0; q < K * K; ++q) wt[q] = src[size_t(q) * p.Cexp];
        bdv = p.bd[c0 + cb * 16 + cl];
    };
    int it = wave;
    if (it < nitem) load_taps(it);
    constexpr int W1Q = CC / 16;
    float4v w1v[W1Q];
    {
        const int jo = (tid >> 2) & 63, q = tid & 3;
        const float* wrow = p.w1t + size_t(jo < p.R ? jo : p.R - 1) * p.Cexp + c0 + q * (CC / 4);
#pragma unroll
        for (int i = 0; i < W1Q; ++i) w1v[i] = *reinterpret_cast<const float4v*>(wrow + 4 * i);
    }
    lds_barrier();

    float* s_red = reinterpret_cast<float*>(smem + p.off_red);          // [G][2][CC]
    float* s_sum = reinterpret_cast<float*>(smem + p.off_sum);          // [G][CC]
    const int unit = j * 16 + (cl ^ (j << 2));
    float* outp = reinterpret_cast<float*>(p.out);

    for (; it < nitem; it += NWAVE) {
        const int xgl = it & 1, cbq = it >> 1, cb = cbq % NCB, jq = cbq / NCB;       // (uniform)
        const unsigned char* bp = E + ((((jq * NCB + cb) * ROWS) * 64 + unit) << 5);
        float acc[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r][i] = 0.f;
        // the two x-groups read different pixels of the row: two unrolled bodies, chosen by the (uniform) group
        auto rows = [&](auto xg) {
            constexpr int XG = decltype(xg)::value;
#pragma unroll
            for (int er = 0; er < ROWS; ++er) {
                const float4v lo = *reinterpret_cast<const float4v*>(bp + er * 2048);
                const float4v hi = *reinterpret_cast<const float4v*>(bp + er * 2048 + 16);
                const float in[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int d = er - ky + PAD;
                    if (d >= 0 && d < ROWS) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) {
                                const int px = 4 * XG + i + kx - PAD;          // input pixel of this tap (0..6 in the image)
                                if (px >= 0 && px < HW7) acc[d][i] = fmaf(in[px], wt[ky * K + kx], acc[d][i]);
                            }
                    }
                }
            }
        };
        if (xgl == 0) rows(std::integral_constant<int, 0>{});
        else rows(std::integral_constant<int, 1>{});
        const float bd_this = bdv;
        if (it + NWAVE < nitem) load_taps(it + NWAVE);
        const int cr = jq * 4 + j;
        const bool okc = cr < G && crop0 + cr < p.n;
        float sum = 0.f;
        float* ob = outp + (size_t(crop0 + cr) * 49 + 4 * xgl) * p.Cexp + c0 + cb * 16 + cl;
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = conv_swish<float>(acc[r][i] + bd_this);
                if (okc && 4 * xgl + i < HW7) {
                    sum += y;
                    ob[size_t(r * 7 + i) * p.Cexp] = y;
                }
            }
        if (cr < G) s_red[(cr * 2 + xgl) * CC + cb * 16 + cl] = okc ? sum : 0.f;
    }
    lds_barrier();

    // ---- squeeze-excite, first half (as the f16 kernel) -------------------------------------------------------------------
#pragma unroll
    for (int i0 = 0; i0 < G * CC; i0 += NTHR) {
        const int i = i0 + tid;
        if (i < G * CC) {
            const int cr = i / CC, c = i % CC;
            s_sum[i] = s_red[(cr * 2) * CC + c] + s_red[(cr * 2 + 1) * CC + c];
        }
    }
    lds_barrier();
    {
        const int jo = (tid >> 2) & 63, q = tid & 3;
        for (int cr = tid >> 8; cr < G; cr += NTHR / 256) {
            float accr = 0.0f;
            if (jo < p.R) {
                const float* sp = s_sum + cr * CC + q * (CC / 4);
#pragma unroll
                for (int i = 0; i < W1Q; ++i) {
                    const float4v sv = *reinterpret_cast<const float4v*>(sp + 4 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accr = fmaf(sv[e], w1v[i][e], accr);
                }
            }
            const float pair = accr + quad_xor1(accr);
            const float tot = pair + quad_xor2(pair);
            if (q == 0 && jo < p.RPse && crop0 + cr < p.n)
                p.rpart[(size_t(crop0 + cr) * p.chunks + chunk) * p.RPse + jo] = (jo < p.R) ? tot : 0.0f;
        }
    }
}

struct OncePerDevice7 {
    std::atomic<bool> done[64];
    OncePerDevice7() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
};

template <int K, int KS, int G, int CC, int NTHR, bool F32, bool SP = false>
void launch_f7(const Front7Args& a, hipStream_t stream) {
    const Front7Plan& pl = a.plan;
    F7Params p{};
    p.x = a.x;
    p.wep = SP ? a.weps : a.wep;
    p.wsi = a.wsi;
    p.be = a.be;
    p.wdt = a.wdt;
    p.bd = a.bd;
    p.out = a.out;
    p.rpart = a.rpart;
    p.w1t = a.w1t;
    p.n = a.n;  p.Cin = a.Cin;  p.Cexp = a.Cexp;  p.NTe = a.NTe;
    p.R = a.R;  p.RPse = (a.R + 3) & ~3;
    p.off_w = pl.off_w;  p.off_stage = pl.off_stage;  p.off_red = pl.off_red;  p.off_sum = pl.off_sum;
    WHENET_REQUIRE(pl.lds_bytes <= 160 * 1024, WHENET_EINVAL, "front7: the tile plan needs more than 160 KB of LDS");
    static OncePerDevice7 attr;
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    auto kern = [] {
        if constexpr (F32) return &whenet_front7_f32_kernel<K, KS, G, CC, NTHR, SP>;
        else return &whenet_front7_kernel<K, KS, G, CC, NTHR>;
    }();
    if (dev >= 0 && dev < 64 && !attr.done[dev].load(std::memory_order_acquire)) {
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done[dev].store(true, std::memory_order_release);
    }
    p.chunks = pl.chunks;  p.ngroups = ceil_div(a.n, pl.G);  p.xcd = a.xcd_grouped ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(unsigned(p.chunks) * unsigned(p.ngroups)), dim3(NTHR), pl.lds_bytes, stream, p);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace

// LDS: E [quads][CC / 16][7][64 units][16 B] | expand weights [KS][CC / 32][64][16 B] | output stage 2 KB per wave |
// per-(crop, x-group) channel sums | per-crop channel sums
Front7Plan make_front7_plan(int dtype, int Cin, int Cexp, int G, int CC, int threads) {
    WHENET_REQUIRE((threads == 256 || threads == 512) && G >= 1 && G <= 8 && (CC == 32 || CC == 64 || CC == 128) && Cexp % CC == 0 &&
                       Cin % 16 == 0,
                   WHENET_EINVAL, "front7: bad tile plan");
    Front7Plan p;
    p.threads = threads;
    p.G = G;
    p.CC = CC;
    p.chunks = Cexp / CC;
    const bool f32 = dtype == WHENET_F32;
    const int njq = (G + 3) / 4, ncb = CC / 16, ks = Cin / (f32 ? 8 : 16), nt = CC / 32;
    const size_t e_bytes = size_t(njq) * ncb * HW7 * 64 * (f32 ? 32 : 16);
    p.off_w = int(e_bytes);
    // f16: the output stage (2 KB per wave) ALIASES the weight region -- the staged weights are dead once the expand phase is
    // over, and a workgroup barrier separates the phases; the f32 kernel stores its outputs directly
    const int w_bytes = ks * nt * 1024, stage_bytes = f32 ? 0 : (threads / 64) * 2048;
    p.off_stage = p.off_w;
    p.off_red = p.off_w + (w_bytes > stage_bytes ? w_bytes : stage_bytes);
    p.off_sum = p.off_red + G * 2 * CC * 4;
    p.lds_bytes = size_t(p.off_sum) + size_t(G) * CC * 4;
    return p;
}

// The plan for a launch of n crops (measured on MI355X, tools/probes/front7_probe.hip, profiles/r04/front7_probe.txt):
// 64 channels x 8 waves always; groups of 4 crops (7 full MFMA strips, the chunk's weights fetched once per 4 crops) from 17
// crops per launch up, groups of 2 below (twice the workgroups when the launch cannot fill the chip anyway: 6.6 vs 7.8 us at
// one crop, 7.5 vs 8.3 us at 16).  The channel chunk must NOT depend on n: the squeeze-excite partial vectors are summed
// per chunk, so only plans with the same CC give a crop the same bits; the group size changes nothing in them.
// f32: the expand phase is matrix-pipe bound (v_mfma_f32_32x32x2_f32: 64 cycles each, 96 per task), so a single crop travels
// alone (G = 1: 4 tasks on the 4 SIMDs instead of 8 half-empty ones).
Front7Plan front7_plan_for(int dtype, int Cin, int Cexp, int n) {
    const int G = (dtype == WHENET_F32 && n == 1) ? 1 : (n <= 16 ? 2 : 4);
    return make_front7_plan(dtype, Cin, Cexp, G, 64, 512);
}

bool front7_supported(int k, int s, int H, int Cin) { return (k == 3 || k == 5) && s == 1 && H == HW7 && Cin == 192; }

void launch_front7(const Front7Args& a, hipStream_t stream) {
    WHENET_REQUIRE(front7_supported(a.k, 1, HW7, a.Cin) && a.w1t != nullptr && a.R >= 1 && a.R <= 64 && a.n >= 1, WHENET_EINVAL,
                   "front7: 7 x 7 maps, 3x3 / 5x5 stride-1 kernels, Cin = 192, squeeze-excite reduce conv in the kernel");
    const int key = ((a.k * 10 + a.plan.G) * 1000 + a.plan.CC) * 1000 + a.plan.threads;
    if (a.dtype == WHENET_F32) {
        switch (key) {
#define F7_CASE32(K, G, CC, T) case ((K * 10 + G) * 1000 + CC) * 1000 + T: \
        if (a.split) launch_f7<K, 24, G, CC, T, true, true>(a, stream); else launch_f7<K, 24, G, CC, T, true>(a, stream); break;
            F7_CASE32(5, 4, 64, 512) F7_CASE32(3, 4, 64, 512)
            F7_CASE32(5, 2, 64, 512) F7_CASE32(3, 2, 64, 512)
            F7_CASE32(5, 1, 64, 512) F7_CASE32(3, 1, 64, 512)
#undef F7_CASE32
            default: throw Error(WHENET_EINVAL, "front7: no f32 instantiation for this (kernel, G, CC, lanes) plan");
        }
        return;
    }
    switch (key) {
#define F7_CASE(K, G, CC, T) case ((K * 10 + G) * 1000 + CC) * 1000 + T: launch_f7<K, 12, G, CC, T, false>(a, stream); break;
        F7_CASE(5, 4, 64, 512) F7_CASE(3, 4, 64, 512)
        F7_CASE(5, 2, 64, 512) F7_CASE(3, 2, 64, 512)
#ifdef WHENET_FRONT7_ALL_PLANS                   // the probe's sweep (tools/probes/front7_probe.hip)
        F7_CASE(5, 4, 64, 256) F7_CASE(3, 4, 64, 256)
        F7_CASE(5, 4, 32, 256) F7_CASE(3, 4, 32, 256)
        F7_CASE(5, 4, 32, 512) F7_CASE(3, 4, 32, 512)
        F7_CASE(5, 2, 64, 256) F7_CASE(3, 2, 64, 256)
        F7_CASE(5, 2, 32, 256) F7_CASE(3, 2, 32, 256)
        F7_CASE(5, 8, 32, 512) F7_CASE(3, 8, 32, 512)
        F7_CASE(5, 8, 64, 512) F7_CASE(3, 8, 64, 512)
        F7_CASE(5, 4, 128, 512) F7_CASE(3, 4, 128, 512)
        F7_CASE(5, 1, 64, 256) F7_CASE(3, 1, 64, 256)
#endif
#undef F7_CASE
        default: throw Error(WHENET_EINVAL, "front7: no instantiation for this (kernel, G, CC, lanes) plan");
    }
}

std::string kernel_name_front7(int dtype, int k, const Front7Plan& p, bool split) {
    return std::string(dtype == WHENET_F32 ? "whenet_front7_f32_kernel<" : "whenet_front7_kernel<") + std::to_string(k) +
           (dtype == WHENET_F32 ? ", 24, " : ", 12, ") + std::to_string(p.G) + ", " + std::to_string(p.CC) + ", " +
           std::to_string(p.threads) + (dtype == WHENET_F32 ? (split ? ", true>" : ", false>") : ">");
}

}  // namespace whenet
