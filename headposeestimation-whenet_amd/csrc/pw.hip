// 1x1 convolutions (expand / project / head conv) as MFMA GEMMs over M = n*H*W pixel rows.
//
// Reference: efficientnet 0.0.4's Conv2D(.., 1x1, 'same', no bias) -> BN (-> Swish), the SE
// `Multiply([gate, x])` in front of the project conv and the identity-skip `Add` behind it,
// all instantiated by /root/reference/whenet.py:8 (33 such convs, 88 % of
// the MACs).  Fused here:  out = act( (a * gate[crop]) @ Wfolded + bias ) + skip.
//
// MFMA mapping (gfx950, wave64).  The product is computed TRANSPOSED:
//     D[i = out-channel n][j = pixel row m]  =  sum_k  Wt[n][k] * Act[m][k]
// i.e. the weights are the MFMA "A" operand and the activations the "B" operand, because
//   (1) both operands are then k-contiguous: a lane's B fragment is 16 contiguous bytes of
//       its own NHWC pixel row (global_load_dwordx4, no LDS transpose), and its A fragment is
//       16 contiguous bytes of the host-packed weight image (snapshot.h), so one wave reads a
//       dense 1 KiB block per tile-step;
//   (2) the 32x32 accumulator layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
//       leaves each lane with four runs of 4 consecutive out-channels of ONE pixel row:
//       the epilogue stores 8 B (f16) / 16 B (f32) pieces straight into NHWC, no shuffles.
//   f16: v_mfma_f32_32x32x16_f16, one instruction per 16-deep k-step (8 halfs per lane);
//   f32: v_mfma_f32_32x32x2_f32 x4 per 8-deep k-step (exact f32, bitwise an fmaf chain); a lane
//        loads 4 consecutive k of its row at once and feeds element t to the t-th instruction;
//        the packed weights follow the same (lane-group, t) -> k permutation.
// A wave owns 32 pixel rows x NT 32-wide out-channel tiles; a workgroup is 4 independent waves
// (128 rows) -- no LDS, no barriers: the weights of a layer are at most 0.8 MB and stay in L2,
// the activation rows are streamed exactly once per out-channel chunk.
// Workgroup -> tile mapping is XCD-aware: workgroups are dispatched round-robin over the 8 XCDs
// (id % 8), so the chunks of one 128-row block are given ids that are congruent mod 8 and
// adjacent in time: they hit the same XCD's L2 for the shared activation rows.
//
// HBM bytes per launch: M*(K + N)*sizeof(T) (+ M*N*sizeof(T) skip) ; FLOPs 2*M*K*N.
#include "device_math.h"
#include "kernels.h"
#include "se_device.h"
#include "stamps.h"

#ifndef WHENET_PW_DEPTH
#define WHENET_PW_DEPTH 2       // operand groups a split-K wave keeps in flight (probes build other depths)
#endif

namespace whenet {

namespace {

// Combine of the four waves' partial accumulators + bias / activation / skip / store (both split-K kernels): wave p owns a quarter of
// the accumulators, receives the other three waves' share through LDS and adds them in wave order ((p0 + p1) + p2) + p3.
template <typename T, int B2, bool RES, int ACT, bool SP>
__device__ __forceinline__ void splitk_combine_store(float16v (&acc)[B2][B2], float* s_red, int kpart, int lane, int mt, int nt0, int M,
                                                     int N, int NTILES, const float* __restrict__ bias, const T* __restrict__ res,
                                                     T* __restrict__ out, float wsi) {
    using OT = T __attribute__((ext_vector_type(4)));
    constexpr int MB = B2, NT = B2;
    constexpr int NACC = MB * NT * 16, SL = NACC / 4;
    const int g = lane >> 5;
    // ---- combine: wave p owns accumulators [p*SL, (p+1)*SL) of the flattened (mb, t, r) index -----
    const int f0 = kpart * SL;                       // first flattened accumulator of this wave's slice
    const int own_tile = f0 >> 4, own_mb = own_tile / NT, own_t = own_tile % NT;
    const int own_row = (mt * MB + own_mb) * 32 + (lane & 31);
    const bool own_valid = own_row < M && nt0 + own_t < NTILES;
    OT rv[SL / 4];
    float4v bv[SL / 4];
#pragma unroll
    for (int i4 = 0; i4 < SL / 4; ++i4) {            // skip rows and bias of the slice: in flight across the exchange
        const int qq = ((f0 & 15) >> 2) + i4;
        const int n0 = (nt0 + own_t) * 32 + 8 * qq + 4 * g;
        const bool ok = own_valid && n0 < N;
        bv[i4] = ok ? *reinterpret_cast<const float4v*>(bias + n0) : float4v{0.f, 0.f, 0.f, 0.f};
        if constexpr (RES) rv[i4] = ok ? *reinterpret_cast<const OT*>(res + size_t(own_row) * N + n0) : OT{};
    }
    float own[SL];
#pragma unroll
    for (int dst = 0; dst < 4; ++dst) {
        if (dst == kpart) {
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int f = dst * SL + i;
                own[i] = acc[(f >> 4) / NT][(f >> 4) % NT][f & 15];
            }
        } else {
            const int src = (kpart < dst) ? kpart : kpart - 1;
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int f = dst * SL + i;
                s_red[((dst * 3 + src) * SL + i) * 64 + lane] = acc[(f >> 4) / NT][(f >> 4) % NT][f & 15];
            }
        }
    }
    lds_barrier();
    STAMP(3);
    float sum[SL];
#pragma unroll
    for (int src = 0; src < 4; ++src) {
        if (src == kpart) {
#pragma unroll
            for (int i = 0; i < SL; ++i) sum[i] = (src == 0) ? own[i] : sum[i] + own[i];
        } else {
            const int si = (src < kpart) ? src : src - 1;
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const float v = s_red[((kpart * 3 + si) * SL + i) * 64 + lane];
                sum[i] = (src == 0) ? v : sum[i] + v;
            }
        }
    }
    if (!own_valid) return;
#pragma unroll
    for (int i4 = 0; i4 < SL / 4; ++i4) {
        const int qq = ((f0 & 15) >> 2) + i4;
        const int n0 = (nt0 + own_t) * 32 + 8 * qq + 4 * g;
        if (n0 >= N) continue;
        OT o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = SP ? fmaf(sum[4 * i4 + r], wsi, bv[i4][r]) : sum[4 * i4 + r] + bv[i4][r];
            if constexpr (ACT == ACT_SWISH) y = conv_swish<T>(y);
            if constexpr (RES) y += float(rv[i4][r]);
            o[r] = T(y);
        }
        *reinterpret_cast<OT*>(out + size_t(own_row) * N + n0) = o;
    }
    STAMP(4);
}

// Deep contractions (K >= 320: the project convs of blocks 7-16 and the head conv, all on 14x14 / 7x7
// maps, i.e. few rows).  A workgroup owns B2*32 rows x B2*32 out-channels (B2 x B2 MFMA tiles per wave);
// its 4 waves split the k-steps interleaved (wave p takes k-steps p, p+4, ..) and the four partial
// accumulators are combined through LDS in wave order ((p0 + p1) + p2) + p3 -- for every batch size and
// for both tile shapes, so a crop's bits do not depend on the batch it travels in.
//   * B2 = 2 when that still launches enough workgroups to spread over the chip: every operand fragment
//     then feeds two MFMAs (half the L2 -> CU traffic per flop of the 1 x 1 form);
//   * the k-loop is software-pipelined in registers: the loads of the NEXT group of U k-steps (weights,
//     activation rows, gate) are issued before the MFMAs of the current one, so a wave always has one
//     group in flight while it multiplies (the kernel is latency-bound otherwise: few rows, deep K);
//   * the SE gate multiplies the activation fragment as a T x T product (one packed multiply per two
//     halfs; se.hip stores the gate in T).  Round 3: the gate rows of the <= 3 crops a workgroup's rows belong to
//     are staged ONCE in LDS (<= 6.9 KB) instead of being fetched per pixel row and k-step from global memory -- that
//     fetch was a third of the workgroup's operand bytes, and a CU pulls only ~50 GB/s from L2 (same bits);
//   * the combine is spread over the 4 waves: wave p owns a quarter of the accumulators (B2 = 2: one of
//     the four tiles; B2 = 1: four consecutive out-channels), receives the other three waves' share of it
//     through LDS and runs bias / skip / store for it.
// GM: 0 = no gate, 1 = gate [n][K] from global memory (block 1, and every block with option se_fuse=0), 2 = the
// workgroup computes the gate rows of its own crops from the producer's squeeze-excite partial vectors (se_device.h)
template <typename T, int B2, int GM, bool RES, int ACT, bool SP = false>
__global__ __launch_bounds__(256, B2 == 2 ? 2 : 3) void whenet_pw_splitk_kernel(
    const T* __restrict__ A, const T* __restrict__ Wp, const float* __restrict__ bias, const T* __restrict__ gate,
    const T* __restrict__ res, T* __restrict__ out, int M, int K, int N, int KS, int NTILES, int HW, int MT, int NCH,
    const SeFuse se, float wsi) {
    constexpr bool GATE = GM != 0;
    using OPS = PwOps<T, SP>;
    constexpr int V = OPS::V;                       // k elements per lane and k-step (SP: 8 floats)
    constexpr int SV = Vec<T>::V;                   // elements of a 16-byte vector of the storage type
    using VT = typename Vec<T>::type;
    using OT = T __attribute__((ext_vector_type(4)));
    constexpr int MB = B2, NT = B2, SK = 4;
    constexpr int U = (B2 == 2) ? 2 : 4;            // k-steps per load group; two groups are in flight
    constexpr int NACC = MB * NT * 16, SL = NACC / 4;
    __shared__ float s_red[4 * 3 * SL * 64];        // [owner][source (3 others)][SL][lane]
    constexpr int GCROPS = MB + 1, GK = 1152;       // crops a workgroup's MB*32 rows can touch (HW >= 49), max K
    __shared__ __attribute__((aligned(16))) T s_gate[GATE ? GCROPS * GK : 8];
    __shared__ float s_r[GM == 2 ? GCROPS * 48 : 4];

    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;
    STAMP(0);

    const int lane = threadIdx.x & 63;
    const int kpart = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt0 = nch * NT;
    const int g = lane >> 5;
    bool rvalid[MB];
    const T* ap[MB];
    const T* gp[MB];                                // (LDS) this lane's row of the staged gate, per row block
    const int row_first = mt * MB * 32;
    const int crop_lo = row_first / HW;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int row = (mt * MB + mb) * 32 + (lane & 31);
        rvalid[mb] = row < M;
        const int rowc = rvalid[mb] ? row : (M - 1);
        ap[mb] = A + size_t(rowc) * K + g * V;
        gp[mb] = GATE ? s_gate + (rowc / HW - crop_lo) * K + g * V : nullptr;
    }
    const int row_last = (row_first + MB * 32 < M ? row_first + MB * 32 : M) - 1;
    const int ncrop = row_last / HW - crop_lo + 1;                     // <= GCROPS
    // gate rows of crops crop_lo .. crop_hi -> LDS, 16 bytes per lane (K is a multiple of 16): all loads issued at once, in front
    // of the first operand group, written to LDS once that group is on its way as well
    constexpr int GV = GM == 1 ? (GCROPS * GK / SV + 255) / 256 : 1;
    VT gv[GV];
    if constexpr (GM == 1) {
        const VT* src = reinterpret_cast<const VT*>(gate + size_t(crop_lo) * K);
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            const int i = int(threadIdx.x) + j * 256;
            gv[j] = src[i < ncrop * K / SV ? i : 0];
        }
    }
    const size_t w_lane = size_t(nt0) * 64 + lane;                     // this lane's fragment of tile nt0, k-step 0
    const size_t w_lo = size_t(KS) * NTILES * 64;                      // (SP) fragments between the hi and the lo image

    float16v acc[MB][NT];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.0f;

    struct Ops {
        typename OPS::A a[U][MB];
        typename OPS::W w[U][NT];
    };
    // Every load of the k-loop is UNCONDITIONAL (addresses clamped to the last k-step / tile / row; the MFMAs of a k-step past
    // KS or of a tile past NTILES are skipped by wave-uniform branches, rows past M are never stored): with loads under
    // per-lane or per-group conditions the compiler cannot count what is outstanding and waits with vmcnt(0) -- for the group
    // it has just issued as well, which serialises the "next group in flight" the loop is written for (round 4, from the ISA).
    auto issue = [&](int ks, Ops& o) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k1 = ks + u * SK;
            const int k1c = k1 < KS ? k1 : KS - 1;                     // (wave-uniform)
            const size_t wk = w_lane + size_t(k1c) * NTILES * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) o.w[u][t] = OPS::load_w(Wp, wk + (nt0 + t < NTILES ? t : 0) * 64, w_lo);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) o.a[u][mb] = OPS::load_a(ap[mb] + k1c * 2 * V);
        }
    };
    auto compute = [&](const Ops& o, int ks) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k1 = ks + u * SK;
            if (k1 >= KS) continue;                                    // (wave-uniform)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                typename OPS::A a = o.a[u][mb];
                if constexpr (GATE) OPS::gate(a, gp[mb] + k1 * 2 * V);
                const typename OPS::P pa = OPS::prep(a);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) OPS::step(o.w[u][t], pa, acc[mb][t]);
            }
        }
    };

    constexpr int D = WHENET_PW_DEPTH;              // operand groups in flight per wave
    Ops o[D];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(kpart + d * U * SK, o[d]);
    if constexpr (GM == 1) {
        VT* dst = reinterpret_cast<VT*>(s_gate);
#pragma unroll
        for (int j = 0; j < GV; ++j) {
            const int i = int(threadIdx.x) + j * 256;
            if (i < ncrop * K / SV) dst[i] = gv[j];
        }
        lds_barrier();
    }
    if constexpr (GM == 2) se_fused_to_lds<T, 256>(se, crop_lo, ncrop, K, s_gate, s_r);    // (operands already in flight)
    STAMP(1);
    for (int ks = kpart; ks < KS; ks += D * U * SK) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(ks + (d + D - 1) * U * SK, o[(d + D - 1) % D]);
            compute(o[d], ks + d * U * SK);
        }
    }
    STAMP(2);

    splitk_combine_store<T, B2, RES, ACT, SP>(acc, s_red, kpart, lane, mt, nt0, M, N, NTILES, bias, res, out, wsi);
}

// Round 5: the same product with the activation rows fetched COALESCED and handed to the matrix cores through LDS.
// Why: a wave's activation fragment is 16 bytes of each of 32 pixel rows (row pitch K x sizeof(T) >= 640 B here): one wave-instruction
// touches 32 cache lines for 1 KB.  tools/probes/fetch_pattern_probe.hip, 147 workgroups x 147 KB (b13-15's project conv at 64 crops):
// 27 GB/s per CU with that pattern, 55 GB/s with 8 rows x 128 contiguous bytes per instruction (8 lines per KB) -- the vector cache's
// line rate, not the L2's bandwidth, was the wall of this kernel (0.13-0.14 of HBM peak, rounds 3-4).  Here
//   * the K axis is cut into 128-byte GROUPS of a pixel row (f16: 4 k-steps of 16; f32: 4 of 8; f32s: 2 of 16) and wave p owns groups
//     p, p + 4, ...; a group of the workgroup's MB x 32 rows is MB x 4 wave-instructions of 8 rows x 128 B;
//   * the loads of the NEXT group travel in registers while the current one is multiplied out of the wave's own 4-8 KB LDS region
//     (DS operations of a wave execute in order: the stores of group i+1 follow the fragment reads of group i, no barrier);
//   * slots of 16 B are XOR-swizzled with the row ((r ^ r >> 3) & 7): the coalesced stores and the fragment reads (lane = row j,
//     piece q) are both bank-conflict free;
//   * weights (already linear 1 KB per instruction), gate staging, the combine and the epilogue are the kernel's above.
// Summation order: wave p's partial sum now runs over its k-GROUPS; the combine order is unchanged -- a function of the layer only.
template <typename T, int B2, int GM, bool RES, int ACT, bool SP = false>
__global__ __launch_bounds__(256, B2 == 2 ? 2 : 3) void whenet_pw_splitk_staged_kernel(
    const T* __restrict__ A, const T* __restrict__ Wp, const float* __restrict__ bias, const T* __restrict__ gate,
    const T* __restrict__ res, T* __restrict__ out, int M, int K, int N, int KS, int NTILES, int HW, int MT, int NCH,
    const SeFuse se, float wsi) {
    constexpr bool GATE = GM != 0;
    using OPS = PwOps<T, SP>;
    constexpr int V = OPS::V;
    constexpr int SV = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int MB = B2, NT = B2;
    constexpr int R = MB * 32;                            // rows of the workgroup's tile
    constexpr int KSB = 2 * V * int(sizeof(T));           // bytes of a pixel row per k-step: 32 (f16, f32) | 64 (f32s)
    constexpr int KGS = 128 / KSB;                        // k-steps per 128-byte group
    constexpr int NLD = R / 8;                            // 16-byte loads per lane and group (8 rows x 128 B per wave-instruction)
    constexpr int NACC = MB * NT * 16, SL = NACC / 4;
    constexpr int STAGE_BYTES = 4 * R * 128, RED_BYTES = 4 * 3 * SL * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES];
    float* s_red = reinterpret_cast<float*>(smem);        // (aliases the staging regions: used after the k-loop, behind a barrier)
    constexpr int GCROPS = MB + 1, GK = 1152;
    __shared__ __attribute__((aligned(16))) T s_gate[GATE ? GCROPS * GK : 8];
    __shared__ float s_r[GM == 2 ? GCROPS * 48 : 4];
    // GM == 3 (round 6): the gate on the matrix cores, per wave, for the k-groups the wave owns -- no workgroup-wide stage, no barrier.
    //   z[crop][c] = sum_j r[crop][j] * W2[j][c]  is a v_mfma_f32_32x32x16_f16 product with the crops' r vectors as rows 0 / 4 / 8 / 12
    //   (the accumulator then holds crop 2 s + (lane >> 5) of channel (lane & 31) in register 4 s) and a 32-channel tile of the excite
    //   kernel's operand image as columns: ceil(R / 16) instructions per tile (f32s: three, hi/lo pairs), two tiles per 128-byte group
    //   (f32s: one).  r itself is computed by every wave (lanes = j; the np partial vectors of its <= 3 crops) with se.hip's arithmetic.
    constexpr int GT = 128 / int(sizeof(T)) / 32;         // excite tiles per k-group: 2 (binary16 rows) | 1 (float rows)
    constexpr int KSRM = 3;                               // ceil(R / 16) <= 3 (R <= 48)
    constexpr int NIMG = SP ? 2 : 1;
    __shared__ __attribute__((aligned(16))) half_t s_rw[GM == 3 ? 4 * NIMG * 4 * 48 : 8];      // [wave][hi | lo][crop 0..3][48]

    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;

    const int lane = threadIdx.x & 63;
    const int kpart = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt0 = nch * NT;
    const int g = lane >> 5, j = lane & 31;
    const int row_first = mt * R;
    const int crop_lo = row_first / HW;
    const T* gp[MB];                                      // (LDS) this lane's row of the staged gate, per row block
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int row = row_first + mb * 32 + j;
        const int rowc = row < M ? row : (M - 1);
        gp[mb] = GATE ? s_gate + (rowc / HW - crop_lo) * K + g * V : nullptr;
    }
    const int row_last = (row_first + R < M ? row_first + R : M) - 1;
    const int ncrop = row_last / HW - crop_lo + 1;
    constexpr int GV = GM == 1 ? (GCROPS * GK / SV + 255) / 256 : 1;
    VT gv[GV];
    if constexpr (GM == 1) {
        const VT* src = reinterpret_cast<const VT*>(gate + size_t(crop_lo) * K);
#pragma unroll
        for (int jj = 0; jj < GV; ++jj) {
            const int i = int(threadIdx.x) + jj * 256;
            gv[jj] = src[i < ncrop * K / SV ? i : 0];
        }
    }
    const size_t w_lane = size_t(nt0) * 64 + lane;
    const size_t w_lo = size_t(KS) * NTILES * 64;

    // ---- coalesced source addresses and the swizzled LDS slots ------------------------------------------------------------
    const int lr = lane >> 3, lq = lane & 7;
    const int rowbytes = K * int(sizeof(T));
    const unsigned char* arow[NLD];
    int woff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = i * 8 + lr;
        const int row = row_first + r;
        arow[i] = reinterpret_cast<const unsigned char*>(A) + size_t(row < M ? row : M - 1) * rowbytes;
        woff[i] = r * 128 + ((lq ^ ((r ^ (r >> 3)) & 7)) << 4);
    }
    unsigned char* stg = smem + kpart * (R * 128);
    int rbase[MB], rswz[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = mb * 32 + j;
        rbase[mb] = r * 128;
        rswz[mb] = (r ^ (r >> 3)) & 7;
    }
    const int NG = (KS + KGS - 1) / KGS;

    float16v acc[MB][NT];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.0f;

    using WF = typename OPS::W;
    auto issue_a = [&](int gi, VT (&an)[NLD]) {             // (unconditional: clamped group / byte offset, see the kernel above)
        const int gic = gi < NG ? gi : NG - 1;
        int off = gic * 128 + lq * 16;
        off = off < rowbytes - 16 ? off : rowbytes - 16;
#pragma unroll
        for (int i = 0; i < NLD; ++i) an[i] = *reinterpret_cast<const VT*>(arow[i] + off);
    };
    auto issue_w = [&](int gi, WF (&w)[KGS][NT]) {
#pragma unroll
        for (int s = 0; s < KGS; ++s) {
            const int k1 = gi * KGS + s;
            const int k1c = k1 < KS ? k1 : KS - 1;
            const size_t wk = w_lane + size_t(k1c) * NTILES * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) w[s][t] = OPS::load_w(Wp, wk + (nt0 + t < NTILES ? t : 0) * 64, w_lo);
        }
    };
    auto stage = [&](const VT (&an)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<VT*>(stg + woff[i]) = an[i];
    };
    auto compute = [&](int gi, const WF (&w)[KGS][NT]) {
#pragma unroll
        for (int s = 0; s < KGS; ++s) {
            const int k1 = gi * KGS + s;
            if (k1 >= KS) continue;                         // (wave-uniform)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                typename OPS::A a;
                if constexpr (SP) {
                    const int q0 = s * 4 + g * 2;
                    a.x0 = *reinterpret_cast<const float4v*>(stg + rbase[mb] + ((q0 ^ rswz[mb]) << 4));
                    a.x1 = *reinterpret_cast<const float4v*>(stg + rbase[mb] + (((q0 + 1) ^ rswz[mb]) << 4));
                } else {
                    a.v = *reinterpret_cast<const VT*>(stg + rbase[mb] + (((s * 2 + g) ^ rswz[mb]) << 4));
                }
                if constexpr (GATE) OPS::gate(a, gp[mb] + k1 * 2 * V);
                const typename OPS::P pa = OPS::prep(a);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) OPS::step(w[s][t], pa, acc[mb][t]);
            }
        }
    };

    // ---- GM == 3: excite operands of a k-group, and the group's gate ---------------------------------------------------------
    struct GW {
        half8 w[NIMG][KSRM][GT];
        float b2[GT];
    };
    const int NTse = K >> 5;                                // 32-channel tiles of the excite kernel (K is a multiple of 32 here)
    const half8* w2img = reinterpret_cast<const half8*>(se.w2p);
    const size_t w2_lo = size_t(se.KSr) * NTse * 64;        // (f32s) fragments between the hi and the lo image
    auto issue_g = [&](int gi, GW& gw) {
        if constexpr (GM == 3) {
            const int gic = gi < NG ? gi : NG - 1;
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int tile = gic * GT + t, tilec = tile < NTse ? tile : NTse - 1;
#pragma unroll
                for (int ks = 0; ks < KSRM; ++ks) {
                    const size_t idx = (size_t(ks < se.KSr ? ks : se.KSr - 1) * NTse + tilec) * 64 + lane;
                    gw.w[0][ks][t] = w2img[idx];
                    if constexpr (SP) gw.w[1][ks][t] = w2img[w2_lo + idx];
                }
                gw.b2[t] = se.b2[tilec * 32 + j];
            }
        }
    };
    // the crops' r vectors as the products' row operand: rows 0 / 4 / 8 / 12 carry crops 0..3 (read from the wave's LDS copy where
    // they are multiplied: kept in registers they cost 12 - 24 VGPRs of a kernel that has none to spare at B2 = 2)
    const half_t* rw_lane = s_rw + kpart * (NIMG * 4 * 48) + (((j >> 2) & 1) + 2 * (j >> 3)) * 48 + g * 8;
    const bool rw_has = (j & 3) == 0 && j < 16;
    auto rfrag = [&](int im, int ks) -> half8 {
        return rw_has ? *reinterpret_cast<const half8*>(rw_lane + im * (4 * 48) + ks * 16) : half8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    auto gate_group = [&](int gi, const GW& gw) {
        if constexpr (GM == 3) {
            if (gi >= NG) return;                           // (wave-uniform)
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int tile = gi * GT + t;
                if (tile >= NTse) continue;                 // (wave-uniform: K = 480 ends in half a group)
                float16v z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KSRM; ++ks) {
                    if (ks >= se.KSr) continue;             // (wave-uniform)
                    const half8 rhi = rfrag(0, ks);
                    if constexpr (SP) {
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(rhi, gw.w[1][ks][t], z, 0, 0, 0);            // (small terms first)
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(rfrag(1, ks), gw.w[0][ks][t], z, 0, 0, 0);
                    }
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(rhi, gw.w[0][ks][t], z, 0, 0, 0);
                }
                const int ch = tile * 32 + j;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int c = 2 * sl + g;               // the crop this lane holds in register 4 sl
                    if (c < ncrop) {
                        const float zz = SP ? fmaf(z[4 * sl], se.w2_wsi, gw.b2[t]) : z[4 * sl] + gw.b2[t];
                        s_gate[c * K + ch] = T(sigmoid_f<SP>(zz));
                    }
                }
            }
            wave_lds_sync();
        }
    };
    if constexpr (GM == 3) {
        half_t* rw = s_rw + kpart * (NIMG * 4 * 48);
        if (lane < 48) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float r = 0.f;
                if (c < ncrop && lane < se.RP)
                    r = se_fused_r(se.rpart + (size_t(crop_lo + c) * se.np) * se.RP + lane, se.np, se.RP, se.inv_hw,
                                   lane < se.R ? se.b1[lane] : 0.f, lane < se.R);
                const half_t hi = half_t(r);
                rw[c * 48 + lane] = hi;
                if constexpr (SP) rw[4 * 48 + c * 48 + lane] = half_t(r - float(hi));
            }
        }
        wave_lds_sync();
    }

    VT an[NLD];
    WF w0[KGS][NT], w1[KGS][NT];
    GW g0, g1;
    issue_a(kpart, an);
    issue_w(kpart, w0);
    issue_g(kpart, g0);
    if constexpr (GM == 1) {
        VT* dst = reinterpret_cast<VT*>(s_gate);
#pragma unroll
        for (int jj = 0; jj < GV; ++jj) {
            const int i = int(threadIdx.x) + jj * 256;
            if (i < ncrop * K / SV) dst[i] = gv[jj];
        }
        lds_barrier();
    }
    if constexpr (GM == 2) se_fused_to_lds<T, 256>(se, crop_lo, ncrop, K, s_gate, s_r);
    gate_group(kpart, g0);
    stage(an);
    for (int gi = kpart; gi < NG; gi += 8) {                // two groups per trip: the weight registers swap roles
        issue_a(gi + 4, an);
        issue_w(gi + 4, w1);
        issue_g(gi + 4, g1);
        compute(gi, w0);
        gate_group(gi + 4, g1);
        stage(an);
        if (gi + 4 < NG) {                                  // (wave-uniform)
            issue_a(gi + 8, an);
            issue_w(gi + 8, w0);
            issue_g(gi + 8, g0);
            compute(gi + 4, w1);
            gate_group(gi + 8, g0);
            stage(an);
        }
    }
    lds_barrier();                                          // every wave is done with its staging region: s_red may overwrite it
    splitk_combine_store<T, B2, RES, ACT, SP>(acc, s_red, kpart, lane, mt, nt0, M, N, NTILES, bias, res, out, wsi);
}

// ------------------------------------------------------------------------------------------
// Large-M variant.  Same MFMA mapping; what changes is the data movement around it:
//   * a workgroup owns 128 rows x NT*32 out-channels; the NT weight fragments of a k-step are
//     shared by its 4 waves, so they are staged ONCE per workgroup into LDS (UK k-steps per
//     stage, double-buffered, prefetched through registers while the previous stage is being
//     multiplied) instead of being fetched by every wave from L1/L2.  The packed image is
//     already in fragment order, so the LDS image is lane-linear: ds_read_b128, conflict-free;
//   * NT is chosen so that, whenever the layer allows it, ONE workgroup covers all N
//     out-channels: the activation rows are then read from HBM exactly once;
//   * the epilogue transposes the accumulators through LDS (f32, per-wave region) so that each
//     lane stores 16 bytes and a wave-instruction writes whole 128-byte row segments of the
//     NHWC output (the direct form scatters 8-byte pieces: the write path, not the MFMA, was
//     what bounded the 6x-expanding layers).
template <typename T, int NT, int GM, bool RES, int ACT, bool SP = false>
__global__ __launch_bounds__(256) void whenet_pw_tile_kernel(const T* __restrict__ A, const T* __restrict__ Wp,
                                                             const float* __restrict__ bias,
                                                             const T* __restrict__ gate,
                                                             const T* __restrict__ res, T* __restrict__ out, int M,
                                                             int K, int N, int KS, int NTILES, int HW, int MT,
                                                             int NCH, const SeFuse se, float wsi) {
    constexpr bool GATE = GM != 0;
    constexpr int TGK = 256;                                // K < 320 here: block 6's 240 is the widest gated layer
    __shared__ __attribute__((aligned(16))) T s_gate[GATE ? 2 * TGK : 8];        // 128 rows touch <= 2 crops (HW >= 196)
    __shared__ float s_r[GM == 2 ? 2 * 12 : 4];
    using OPS = PwOps<T, SP>;
    constexpr int V = OPS::V;                               // k elements per lane and k-step (SP: 8 floats)
    constexpr int SV = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int UK = SP ? 2 : 4;                          // k-steps per LDS stage (8: measured -3 % on the K = 32..96 layers)
    constexpr int WT = SP ? 2 * NT : NT;                    // staged fragments per k-step (SP: NT hi + NT lo)
    constexpr int STAGE_VECS = UK * WT * 64;                // 16-byte vectors per stage
    constexpr int CPT = (STAGE_VECS + 255) / 256;           // staging copies per lane
    constexpr int SW = IsF32<T>::value ? 32 : 64;           // epilogue stage width: 128-byte output rows
    constexpr int TPS = SW / 32;                            // 32-wide tiles per epilogue stage
    constexpr int CW = SW / 8;                              // out-channels per 16-byte output chunk
    constexpr int SROW = SW + 4;                            // staged row pitch in floats (16-byte pad)
    constexpr int EPI_FLOATS = 4 * 32 * SROW;
    constexpr int W_FLOATS = 2 * STAGE_VECS * 4;
    __shared__ __attribute__((aligned(16))) float smem[(EPI_FLOATS > W_FLOATS) ? EPI_FLOATS : W_FLOATS];
    VT* s_w = reinterpret_cast<VT*>(smem);                  // [2][UK][NT][64 lanes]

    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;                                   // whole workgroup

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = (mt * 4 + wave) * 32;
    const int nt0 = nch * NT;
    const int g = lane >> 5;
    const int row = m0 + (lane & 31);
    const bool rvalid = row < M;
    const int rowc = rvalid ? row : (M - 1);

    const T* ap = A + size_t(rowc) * K + g * V;
    const int crop_lo = (mt * 128) / HW;
    // the gate rows of this workgroup's <= 2 crops live in LDS for both gated forms (round 4: GM = 1 read them from global
    // memory next to every activation fragment -- and multiplied at once, so each prefetched k-step waited for two loads)
    const T* gp = GATE ? s_gate + (rowc / HW - crop_lo) * K + g * V : nullptr;
    const VT* wsrc = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64;

    float16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // Round 4: every global load of the k-loop is UNCONDITIONAL (k-steps past KS and tiles past NTILES read a clamped address and are
    // never multiplied; rows past M repeat row M - 1 and are never stored), the gate is applied when a fragment is CONSUMED, not when
    // it is requested.  Before, the loads sat under per-lane conditions (the compiler waits with vmcnt(0) then) and the gate
    // multiply right behind each load made the "prefetch" of a group four dependent round trips.
    const bool ktail = K % (2 * V) != 0 && g == 1;          // (only un-gated expand convs with Cin = 24 / 40 have one)
    using AR = typename OPS::A;
    using PR = typename OPS::P;
    auto load_raw = [&](int ks) -> AR {
        const int kc = ks < KS ? ks : KS - 1;
        AR a = OPS::load_a(ap + (ktail && kc == KS - 1 ? -g * V : 0) + kc * 2 * V);
        if (ktail && kc == KS - 1) a = OPS::zero_a();       // (the bytes past the row; their weights are zero, the bytes may be anything)
        return a;
    };
    auto gated = [&](AR a, int ks) -> PR {
        if constexpr (GATE)
            if (ks < KS) OPS::gate(a, gp + ks * 2 * V);      // T x T, one rounding (LDS)
        return OPS::prep(a);
    };

    VT wreg[CPT];
    static_assert(STAGE_VECS % 256 == 0, "every lane stages whole vectors");
    const size_t w_lo = size_t(KS) * NTILES * 64;           // (SP) fragments between the hi and the lo image
    auto fetch_w = [&](int grp) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int i = c * 256 + tid;
            const int u = i / (WT * 64);
            const int j = i - u * (WT * 64);
            const int ks = grp * UK + u;
            const int tj = j >> 6;                          // staged fragment: tile tj (SP: tj >= NT is tile tj - NT of the lo image)
            const int tl = SP && tj >= NT ? tj - NT : tj;
            const int jc = nt0 + tl < NTILES ? tl * 64 + (j & 63) : (j & 63);
            wreg[c] = wsrc[(SP && tj >= NT ? w_lo : 0) + size_t(ks < KS ? ks : KS - 1) * NTILES * 64 + jc];
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int i = c * 256 + tid;
            if (i < STAGE_VECS) s_w[buf * STAGE_VECS + i] = wreg[c];
        }
    };

    const int G = (KS + UK - 1) / UK;
    AR araw[UK];
    PR areg[UK];
    fetch_w(0);
#pragma unroll
    for (int u = 0; u < UK; ++u) araw[u] = load_raw(u);
    const int row_last = (mt * 128 + 128 < M ? mt * 128 + 128 : M) - 1;
    if constexpr (GM == 1) {                                // the gate rows of the <= 2 crops -> LDS (16 bytes per lane)
        const int cnt = (row_last / HW - crop_lo + 1) * K / SV;
        const VT* src = reinterpret_cast<const VT*>(gate + size_t(crop_lo) * K);
        for (int i = tid; i < cnt; i += 256) reinterpret_cast<VT*>(s_gate)[i] = src[i];
    }
    if constexpr (GM == 2)                                  // ... or computed from the producer's squeeze-excite partial vectors
        se_fused_to_lds<T, 256>(se, crop_lo, row_last / HW - crop_lo + 1, K, s_gate, s_r);
    store_w(0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UK; ++u) areg[u] = gated(araw[u], u);
    for (int grp = 0; grp < G; ++grp) {
        // (the last iteration requests a group that does not exist: clamped addresses, an LDS buffer nobody reads -- straight-line
        //  code, so that the compiler counts what is outstanding)
        AR anext[UK];
        fetch_w(grp + 1);
#pragma unroll
        for (int u = 0; u < UK; ++u) anext[u] = load_raw((grp + 1) * UK + u);
        const VT* wl = s_w + (grp & 1) * STAGE_VECS + lane;
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            if (grp * UK + u < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) {
                        typename OPS::W wf;
                        if constexpr (SP) {
                            wf.hi = __builtin_bit_cast(half8, wl[(u * WT + t) * 64]);
                            wf.lo = __builtin_bit_cast(half8, wl[(u * WT + NT + t) * 64]);
                        } else {
                            wf.v = wl[(u * WT + t) * 64];
                        }
                        OPS::step(wf, areg[u], acc[t]);
                    }
            }
        }
        store_w((grp + 1) & 1);
#pragma unroll
        for (int u = 0; u < UK; ++u) areg[u] = gated(anext[u], (grp + 1) * UK + u);
        lds_barrier();                                      // (LDS only: nothing else is exchanged here)
    }

    // ---- epilogue: bias / activation in f32, LDS transpose, 16-byte row-contiguous stores ----
    // Round 4: the bias vectors of a stage and the skip rows its lanes will add are requested TOGETHER at the top of the stage
    // (unconditionally, from clamped addresses): the epilogue used to load each of the 4 bias vectors, and then each of the up to
    // 4 skip pieces, under a condition right in front of its use -- eight dependent round trips per stage and wave.
    float* s_e = smem + wave * 32 * SROW;                   // aliases s_w: every wave is past the last barrier
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TPS) {
        // 16-byte chunks of this stage that exist (narrow layers: N = 16 has 2 per row, not 8): the lanes walk
        // only those, so a 16-channel project conv stores in one pass instead of four quarter-empty ones
        int vc = N - (nt0 + t0) * 32;
        vc = (vc > SW ? SW : vc);
        vc = (vc > (NT - t0) * 32 ? (NT - t0) * 32 : vc) / CW;
        const int vcs = vc > 0 ? vc : 1;                    // (a stage past the layer's last tile: nothing is stored, the requests
        const float rvc = __builtin_amdgcn_rcpf(float(vcs));    //  below still need addresses inside the tensors)
        float4v bvs[TPS][4];
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int n = (nt0 + t0 + tt) * 32 + 8 * qq + 4 * g;
                bvs[tt][qq] = *reinterpret_cast<const float4v*>(bias + (n < N ? n : 0));
            }
        int sr[4], sch[4];
        VT rvs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            const int r = int((float(idx) + 0.5f) * rvc);                         // idx / vc, exact (see front.hip)
            sr[i] = r;
            sch[i] = idx - r * vcs;
            if constexpr (RES) {
                int rowc2 = m0 + (r < 32 ? r : 31);
                rowc2 = rowc2 < M ? rowc2 : M - 1;
                const int nc = vc > 0 ? (nt0 + t0) * 32 + sch[i] * CW : 0;
                rvs[i] = *reinterpret_cast<const VT*>(res + size_t(rowc2) * N + nc);
            }
        }
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = t0 + tt;
            if (t < NT && nt0 + t < NTILES) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int n = (nt0 + t) * 32 + 8 * qq + 4 * g;
                    float4v y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = SP ? fmaf(acc[t < NT ? t : 0][4 * qq + r], wsi, bvs[tt][qq][r]) : acc[t < NT ? t : 0][4 * qq + r] + bvs[tt][qq][r];
                        if constexpr (ACT == ACT_SWISH) v = conv_swish<T>(v);
                        y[r] = (n < N) ? v : 0.f;
                    }
                    *reinterpret_cast<float4v*>(s_e + (lane & 31) * SROW + tt * 32 + 8 * qq + 4 * g) = y;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            if (i * 64 >= 32 * vc) break;                           // (wave-uniform)
            const int r = sr[i], ch = sch[i];
            const int n = (nt0 + t0) * 32 + ch * CW;
            const int rowg = m0 + r;
            if (idx < 32 * vc && rowg < M) {
                float y[CW];
#pragma unroll
                for (int c4 = 0; c4 < CW; c4 += 4) {
                    const float4v v = *reinterpret_cast<const float4v*>(s_e + r * SROW + ch * CW + c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[c4 + j] = v[j];
                }
                if constexpr (RES) {
#pragma unroll
                    for (int j = 0; j < CW; ++j) y[j] += float(rvs[i][j]);
                }
                *reinterpret_cast<VT*>(out + size_t(rowg) * N + n) = float_to_vec<T>(y);
            }
        }
        lds_barrier();
    }
}

// Scalar-FMA check kernel (option pw_impl=1): same operands (weights rounded to T, gate applied
// with the same single rounding), k-ordered fmaf chain per output.  Exists so that the MFMA
// fragment/accumulator mapping can be validated on the device against an independent kernel;
// it is never the default path.
template <typename T, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256) void whenet_pw_check_kernel(const T* __restrict__ A, const float* __restrict__ Wd,
                                                              const float* __restrict__ bias,
                                                              const T* __restrict__ gate,
                                                              const T* __restrict__ res, T* __restrict__ out, int M,
                                                              int K, int N, int HW) {
    const int n4 = N / 4;
    const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= size_t(M) * n4) return;
    const int m = int(idx / n4);
    const int n0 = int(idx - size_t(m) * n4) * 4;
    const T* ap = A + size_t(m) * K;
    const T* gp = GATE ? gate + size_t(m / HW) * K : nullptr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        float a = float(ap[k]);
        if constexpr (GATE) a = float(T(a * float(gp[k])));      // T x T product, rounded once
        const float4v w = *reinterpret_cast<const float4v*>(Wd + size_t(k) * N + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(a, w[r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = acc[r] + bias[n0 + r];
        if constexpr (ACT == ACT_SWISH) y = conv_swish<T>(y);
        if constexpr (RES) y += float(res[size_t(m) * N + n0 + r]);
        out[size_t(m) * N + n0 + r] = T(y);
    }
}

// A K >= 320 conv takes the 2 x 2 register-blocked split-K kernel (same bits as the 1 x 1 form) when
// that still launches enough 64 x 64 workgroups to spread over the chip
// (>= 128 of them; measured: B=64 +5 %, B=512 +8 %, B <= 16 unchanged)
bool use_split2(int M, int NTILES) { return ceil_div(M, 64) * ceil_div(NTILES, 2) >= 128; }

// The LDS-staged split-K kernel where it measured faster (round 5, tools/staged_ab.py, one chain of 64 crops, us per launch, direct ->
// staged): f16 K = 1152: b13-15 13.4 -> 12.7, b16 20.0 -> 14.5; K = 672 on 14 x 14 maps (64 x 64 tiles): 14.2 -> 13.6; K = 480: 12.0 ->
// 12.3 and K = 672 on 7 x 7 maps: 10.1 -> 11.0 stay direct.  The exact-f32 products are bound by the f32 matrix pipe,
// not by the fetch: staged is 5-10 % slower there.  A function of the layer (and the handle's dtype) only.
// (NOT of the batch: the two kernels sum in different orders, and a crop's bits must not depend on the batch it travels in -- so the
//  tile shape B2, which follows the row count, is not part of the rule.)
bool use_staged(const PwArgs& a, bool exact_f32) {
    return a.staged && !exact_f32 && (a.K >= 1152 || (a.K >= 672 && a.HW >= 196));
}

struct PwChoice {
    int kind;      // 1 = split-K kernel (whenet_pw_kernel<T,1,8,4,..>), 2 = LDS-staged tile kernel
    int NT, NCH;
};

// The summation order of a layer must not depend on the batch (a crop's result is bitwise
// independent of the batch it travels in and of how batches are split over streams/GPUs):
// deep contractions (K >= 320: project convs of blocks 7-16 and the head conv, all on 14x14 /
// 7x7 maps, i.e. few rows) ALWAYS split K over the 4 waves of a workgroup, one 32x32 tile per
// workgroup; everything else takes the 128-row LDS-staged kernel, whose NT only decides which
// wave computes which tile.
PwChoice choose_pw(const PwArgs& a, int num_cus) {
    const int cus = num_cus > 0 ? num_cus : 256;
    if (a.K >= 320) return PwChoice{1, 1, a.NTILES};
    // NT = as many of the layer's tiles per workgroup as possible (<= 6) while keeping >= 2
    // workgroups per CU.  Expanding layers (K << N) re-read their tiny activation rows cheaply:
    // NT is capped at 3 there so that 4 waves per SIMD stay resident (NT >= 5 needs > 170 registers).
    const int MT = ceil_div(a.M, 128);
    const int nt_cap = (a.K * 4 <= a.N) ? 3 : 6;
    int NT = 1;
    for (int cand = (a.NTILES < nt_cap ? a.NTILES : nt_cap); cand >= 1; --cand) {
        const int nch = ceil_div(a.NTILES, cand);
        if (MT * nch >= 2 * cus || cand == 1) {
            NT = ceil_div(a.NTILES, nch);        // balance the chunks
            break;
        }
    }
    return PwChoice{2, NT, ceil_div(a.NTILES, NT)};
}

template <typename T, int B2, int GM, bool RES, int ACT, bool SP>
void launch_splitk(const PwArgs& a, hipStream_t stream) {
    const int MT = ceil_div(a.M, 32 * B2), NCH = ceil_div(a.NTILES, B2);
    if (GM == 3 || use_staged(a, IsF32<T>::value && !SP)) {
        hipLaunchKernelGGL((whenet_pw_splitk_staged_kernel<T, B2, GM, RES, ACT, SP>), dim3(8 * ceil_div(MT, 8) * NCH), dim3(256), 0,
                           stream, static_cast<const T*>(a.a), static_cast<const T*>(SP ? a.wps : a.wp), a.bias,
                           static_cast<const T*>(a.gate), static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K,
                           a.N, SP ? a.KSs : a.KS, a.NTILES, a.HW, MT, NCH, a.se, a.wsi);
        return;
    }
    if constexpr (GM != 3)                  // (the matrix-core gate exists in the staged kernel only)
        hipLaunchKernelGGL((whenet_pw_splitk_kernel<T, B2, GM, RES, ACT, SP>), dim3(8 * ceil_div(MT, 8) * NCH), dim3(256), 0,
                           stream, static_cast<const T*>(a.a), static_cast<const T*>(SP ? a.wps : a.wp), a.bias,
                           static_cast<const T*>(a.gate), static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K,
                           a.N, SP ? a.KSs : a.KS, a.NTILES, a.HW, MT, NCH, a.se, a.wsi);
}

template <typename T, int NT, int GM, bool RES, int ACT, bool SP>
void launch_tile(const PwArgs& a, int MT, int NCH, hipStream_t stream) {
    const int blocks = 8 * ceil_div(MT, 8) * NCH;
    hipLaunchKernelGGL((whenet_pw_tile_kernel<T, NT, GM, RES, ACT, SP>), dim3(blocks), dim3(256), 0, stream,
                       static_cast<const T*>(a.a), static_cast<const T*>(SP ? a.wps : a.wp), a.bias, static_cast<const T*>(a.gate),
                       static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, SP ? a.KSs : a.KS, a.NTILES, a.HW, MT,
                       NCH, a.se, a.wsi);
}

template <typename T, int GM, bool RES, int ACT, bool SP = false>
void launch_variant(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    if (impl == 1) {
        WHENET_REQUIRE(GM != 2, WHENET_EINVAL, "pointwise: the check kernel takes the gate from global memory (se_fuse=0)");
        const size_t work = size_t(a.M) * (a.N / 4);
        hipLaunchKernelGGL((whenet_pw_check_kernel<T, GM != 0, RES, ACT>), dim3(unsigned((work + 255) / 256)), dim3(256),
                           0, stream, static_cast<const T*>(a.a), a.wdense, a.bias, static_cast<const T*>(a.gate),
                           static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.HW);
        return;
    }
    const PwChoice ch = choose_pw(a, num_cus);
    if (ch.kind == 1) {
        WHENET_REQUIRE(a.K % (2 * PwOps<T, SP>::V) == 0, WHENET_EINVAL, "pointwise: a deep contraction is a whole number of k-steps");
        if (use_split2(a.M, a.NTILES)) launch_splitk<T, 2, GM, RES, ACT, SP>(a, stream);
        else launch_splitk<T, 1, GM, RES, ACT, SP>(a, stream);
        return;
    }
    const int MT = ceil_div(a.M, 128);
    switch (ch.NT) {
        case 1: launch_tile<T, 1, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
        case 2: launch_tile<T, 2, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
        case 3: launch_tile<T, 3, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
        case 4: launch_tile<T, 4, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
        case 5: launch_tile<T, 5, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
        default: launch_tile<T, 6, GM, RES, ACT, SP>(a, MT, ch.NCH, stream); break;
    }
}

template <typename T, bool SP = false>
void launch_dtype(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    const bool res = a.res != nullptr;
    const int gm = a.se.rpart != nullptr ? (a.se.w2p != nullptr ? 3 : 2) : (a.gate != nullptr ? 1 : 0);
    // the network uses exactly these flavours: expand/head (swish), project (gate from memory | fused SE), + skip
    if (gm == 0 && !res && a.act == ACT_SWISH) launch_variant<T, 0, false, ACT_SWISH, SP>(a, impl, num_cus, stream);
    else if (gm == 1 && !res && a.act == ACT_NONE) launch_variant<T, 1, false, ACT_NONE, SP>(a, impl, num_cus, stream);
    else if (gm == 1 && res && a.act == ACT_NONE) launch_variant<T, 1, true, ACT_NONE, SP>(a, impl, num_cus, stream);
    else if (gm == 2 && !res && a.act == ACT_NONE) launch_variant<T, 2, false, ACT_NONE, SP>(a, impl, num_cus, stream);
    else if (gm == 2 && res && a.act == ACT_NONE) launch_variant<T, 2, true, ACT_NONE, SP>(a, impl, num_cus, stream);
    else if (gm == 3 && a.act == ACT_NONE) {
        // (the matrix-core gate lives in the LDS-staged split-K kernel only: deep contractions of binary16 / split-product handles)
        WHENET_REQUIRE(impl == 0 && a.K >= 320 && a.K % 32 == 0 && (sizeof(T) == 2 || SP) && a.se.KSr >= 1 && a.se.KSr <= 3 &&
                           a.se.RP <= 48, WHENET_EINVAL, "pointwise: the matrix-core squeeze-excite form is outside its limits");
        if constexpr (sizeof(T) == 2 || SP) {
            if (res) {
                if (use_split2(a.M, a.NTILES)) launch_splitk<T, 2, 3, true, ACT_NONE, SP>(a, stream);
                else launch_splitk<T, 1, 3, true, ACT_NONE, SP>(a, stream);
            } else {
                if (use_split2(a.M, a.NTILES)) launch_splitk<T, 2, 3, false, ACT_NONE, SP>(a, stream);
                else launch_splitk<T, 1, 3, false, ACT_NONE, SP>(a, stream);
            }
        }
    }
    else throw Error(WHENET_EINVAL, "pointwise: unsupported epilogue combination");
}

}  // namespace

void launch_pw(const PwArgs& a, int dtype, int impl, int num_cus, hipStream_t stream) {
    WHENET_REQUIRE(a.N % 4 == 0 && a.M > 0, WHENET_EINVAL, "pointwise: bad shape");
    const bool gated = a.gate != nullptr || a.se.rpart != nullptr;
    WHENET_REQUIRE(!(a.gate != nullptr && a.se.rpart != nullptr), WHENET_EINVAL, "pointwise: gate AND fused squeeze-excite");
    WHENET_REQUIRE(!gated || a.K < 320 || (a.K <= 1152 && a.K % 16 == 0 && a.HW >= 49), WHENET_EINVAL,
                   "pointwise: gated deep contraction outside the staged-gate limits (K <= 1152, K % 16 == 0, HW >= 49)");
    WHENET_REQUIRE(!gated || a.K >= 320 || (a.K <= 256 && a.HW >= 196), WHENET_EINVAL,
                   "pointwise: gated shallow contraction outside the staged-gate limits (K <= 256, HW >= 196)");
    WHENET_REQUIRE(a.se.rpart == nullptr || (a.se.RP % 4 == 0 && a.se.RP <= 48 && a.se.np >= 1 &&
                                             (a.K >= 320 || (a.K <= 256 && a.HW >= 196 && a.se.RP <= 12))),
                   WHENET_EINVAL, "pointwise: fused squeeze-excite outside its limits");
    const bool split = a.split && impl == 0;          // (the scalar check kernel multiplies the f32 weights themselves)
    WHENET_REQUIRE(!a.split || (dtype == WHENET_F32 && a.wps != nullptr && a.KSs == ceil_div(a.K, 16)), WHENET_EINVAL,
                   "pointwise: the split-product form needs float32 storage and the split weight images");
    if (dtype == WHENET_F16) launch_dtype<half_t>(a, impl, num_cus, stream);
    else if (split) launch_dtype<float, true>(a, impl, num_cus, stream);
    else launch_dtype<float>(a, impl, num_cus, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

std::string kernel_name_pw(const PwArgs& a, int dtype, int impl, int num_cus) {
    // the instantiation launch_pw() will pick, spelled as rocprofv3 prints it
    const char* t = dtype == WHENET_F16 ? "_Float16" : "float";
    const char* gate = a.se.rpart ? (a.se.w2p ? "3" : "2") : (a.gate ? "1" : "0");
    const char* res = a.res ? "true" : "false";
    char buf[96];
    if (impl == 1) {
        std::snprintf(buf, sizeof(buf), "whenet_pw_check_kernel<%s, %s, %s, %d>", t, a.gate ? "true" : "false", res, a.act);
    } else {
        const PwChoice ch = choose_pw(a, num_cus);
        if (ch.kind == 1) {
            const int b2 = use_split2(a.M, a.NTILES) ? 2 : 1;
            const bool st = (a.se.rpart && a.se.w2p) || use_staged(a, dtype == WHENET_F32 && !a.split);
            std::snprintf(buf, sizeof(buf), "whenet_pw_splitk%s_kernel<%s, %d, %s, %s, %d%s>", st ? "_staged" : "", t, b2, gate, res, a.act,
                          a.split ? ", true" : (st ? ", false" : ""));
        }
        else std::snprintf(buf, sizeof(buf), "whenet_pw_tile_kernel<%s, %d, %s, %s, %d%s>", t, ch.NT, gate, res, a.act, a.split ? ", true" : "");
    }
    return buf;
}

}  // namespace whenet
