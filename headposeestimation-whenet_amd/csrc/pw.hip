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

namespace whenet {

namespace {

// NT  32-wide out-channel tiles per wave;  U  k-steps whose loads are issued together (software
// pipelining: (1+NT)*U 16-byte loads in flight per lane before the first MFMA of the group);
// SK  split-K factor: 1 = the 4 waves of a workgroup own 4 different 32-row strips,
//     4 = the 4 waves split the k-steps of ONE strip (interleaved) and combine through LDS --
//     used when there are too few rows to fill the chip (batch 1: M = 49..3136).
template <typename T, int NT, int U, int SK, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256) void whenet_pw_kernel(const T* __restrict__ A, const T* __restrict__ Wp,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ gate, const T* __restrict__ res,
                                                        T* __restrict__ out, int M, int K, int N, int KS, int NTILES,
                                                        int HW, int MT, int NCH) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int STRIPS = (SK == 1) ? 4 : 1;

    // XCD-aware decode of the 1-D grid (see header)
    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m0 = (mt * STRIPS + (SK == 1 ? wave : 0)) * 32;
    if (SK == 1 && m0 >= M) return;
    const int kpart = (SK == 1) ? 0 : wave;
    const int nt0 = nch * NT;
    const int g = lane >> 5;
    const int row = m0 + (lane & 31);
    const bool rvalid = row < M;
    const int rowc = rvalid ? row : (M - 1);

    const T* ap = A + size_t(rowc) * K + g * V;
    const float* gp = nullptr;
    if constexpr (GATE) gp = gate + size_t(rowc / HW) * K + g * V;
    const VT* wp = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64 + lane;

    float16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto load_a = [&](int ks) -> VT {
        VT a = vec_zero<T>();
        if (rvalid && ks * 2 * V + g * V < K) {
            a = *reinterpret_cast<const VT*>(ap + ks * 2 * V);
            if constexpr (GATE) {
                float f[V];
                vec_to_float<T>(a, f);
#pragma unroll
                for (int i = 0; i < V; i += 4) {
                    const float4v gv = *reinterpret_cast<const float4v*>(gp + ks * 2 * V + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
                }
                a = float_to_vec<T>(f);
            }
        }
        return a;
    };

    // k-loop: the (1+NT)*U loads of a group are issued before its first MFMA; steps past KS
    // (wave-uniform) load nothing and multiply zeros.
    for (int ks = kpart; ks < KS; ks += U * SK) {
        VT a[U];
        VT w[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k1 = ks + u * SK;
            const VT* wk = wp + size_t(k1) * NTILES * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) w[u][t] = (k1 < KS && nt0 + t < NTILES) ? wk[t * 64] : vec_zero<T>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = (ks + u * SK < KS) ? load_a(ks + u * SK) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (nt0 + t < NTILES) Mfma<T>::step(w[u][t], a[u], acc[t]);
    }

    if constexpr (SK > 1) {
        __shared__ float s_red[(SK - 1) * NT * 16 * 64];
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[((wave - 1) * NT * 16 + t * 16 + r) * 64 + lane] = acc[t][r];
        }
        lds_barrier();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SK - 1; ++w)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += s_red[(w * NT * 16 + t * 16 + r) * 64 + lane];
    }

    if (!rvalid) return;
    using OT = T __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= NTILES) continue;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int n0 = (nt0 + t) * 32 + 8 * qq + 4 * g;
            if (n0 >= N) continue;
            const float4v bv = *reinterpret_cast<const float4v*>(bias + n0);
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = acc[t][4 * qq + r] + bv[r];
                if constexpr (ACT == ACT_SWISH) y[r] = swish_f<IsF32<T>::value>(y[r]);
            }
            if constexpr (RES) {
                const OT rv = *reinterpret_cast<const OT*>(res + size_t(row) * N + n0);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
            }
            OT o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
            *reinterpret_cast<OT*>(out + size_t(row) * N + n0) = o;
        }
    }
}

// Register-blocked form of the split-K kernel for deep contractions with MANY rows: a workgroup owns
// 64 rows x 64 out-channels (2 x 2 MFMA tiles per wave), its 4 waves still split the k-steps
// interleaved and combine through LDS in wave order -- the partial sums and the order of
// whenet_pw_kernel<T,1,8,4,..>, so the bits do not change -- but every operand fragment feeds two
// MFMAs: half the L2 -> CU traffic per flop (the 1 x 1 form re-reads an activation strip once per
// out-channel tile and a weight tile once per 32 rows: 8x the algorithmic bytes at 64 crops).
template <typename T, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256, 2) void whenet_pw_split2_kernel(const T* __restrict__ A, const T* __restrict__ Wp,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ gate,
                                                               const T* __restrict__ res, T* __restrict__ out, int M,
                                                               int K, int N, int KS, int NTILES, int HW, int MT,
                                                               int NCH) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int MB = 2, NT = 2, U = 4, SK = 4;
    __shared__ float s_red[(SK - 1) * MB * NT * 16 * 64];

    const int id = blockIdx.x;
    const int q = id >> 3;
    const int nch = q % NCH;
    const int mt = (id & 7) + 8 * (q / NCH);
    if (mt >= MT) return;

    const int lane = threadIdx.x & 63;
    const int kpart = threadIdx.x >> 6;
    const int nt0 = nch * NT;
    const int g = lane >> 5;
    int row[MB];
    bool rvalid[MB];
    const T* ap[MB];
    const float* gp[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        row[mb] = (mt * MB + mb) * 32 + (lane & 31);
        rvalid[mb] = row[mb] < M;
        const int rowc = rvalid[mb] ? row[mb] : (M - 1);
        ap[mb] = A + size_t(rowc) * K + g * V;
        gp[mb] = GATE ? gate + size_t(rowc / HW) * K + g * V : nullptr;
    }
    const VT* wp = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64 + lane;

    float16v acc[MB][NT];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.0f;

    auto load_a = [&](int mb, int ks) -> VT {
        VT a = vec_zero<T>();
        if (rvalid[mb] && ks * 2 * V + g * V < K) {
            a = *reinterpret_cast<const VT*>(ap[mb] + ks * 2 * V);
            if constexpr (GATE) {
                float f[V];
                vec_to_float<T>(a, f);
#pragma unroll
                for (int i = 0; i < V; i += 4) {
                    const float4v gv = *reinterpret_cast<const float4v*>(gp[mb] + ks * 2 * V + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
                }
                a = float_to_vec<T>(f);
            }
        }
        return a;
    };

    for (int ks = kpart; ks < KS; ks += U * SK) {
        VT a[U][MB];
        VT w[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k1 = ks + u * SK;
            const VT* wk = wp + size_t(k1) * NTILES * 64;
#pragma unroll
            for (int t = 0; t < NT; ++t) w[u][t] = (k1 < KS && nt0 + t < NTILES) ? wk[t * 64] : vec_zero<T>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[u][mb] = (ks + u * SK < KS) ? load_a(mb, ks + u * SK) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) Mfma<T>::step(w[u][t], a[u][mb], acc[mb][t]);
    }

    if (kpart > 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s_red[(((kpart - 1) * MB + mb) * NT + t) * 1024 + r * 64 + lane] = acc[mb][t][r];
    }
    lds_barrier();
    if (kpart > 0) return;
#pragma unroll 1
    for (int w = 0; w < SK - 1; ++w)                 // (not unrolled: 64 LDS values in registers at a time)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][t][r] += s_red[((w * MB + mb) * NT + t) * 1024 + r * 64 + lane];

    using OT = T __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        if (!rvalid[mb]) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (nt0 + t >= NTILES) continue;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int n0 = (nt0 + t) * 32 + 8 * qq + 4 * g;
                if (n0 >= N) continue;
                const float4v bv = *reinterpret_cast<const float4v*>(bias + n0);
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = acc[mb][t][4 * qq + r] + bv[r];
                    if constexpr (ACT == ACT_SWISH) y[r] = swish_f<IsF32<T>::value>(y[r]);
                }
                if constexpr (RES) {
                    const OT rv = *reinterpret_cast<const OT*>(res + size_t(row[mb]) * N + n0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
                }
                OT o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
                *reinterpret_cast<OT*>(out + size_t(row[mb]) * N + n0) = o;
            }
        }
    }
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
template <typename T, int NT, bool GATE, bool RES, int ACT>
__global__ __launch_bounds__(256) void whenet_pw_tile_kernel(const T* __restrict__ A, const T* __restrict__ Wp,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ gate,
                                                             const T* __restrict__ res, T* __restrict__ out, int M,
                                                             int K, int N, int KS, int NTILES, int HW, int MT,
                                                             int NCH) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    constexpr int UK = 4;                                   // k-steps per LDS stage
    constexpr int STAGE_VECS = UK * NT * 64;                // 16-byte vectors per stage
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
    const float* gp = nullptr;
    if constexpr (GATE) gp = gate + size_t(rowc / HW) * K + g * V;
    const VT* wsrc = reinterpret_cast<const VT*>(Wp) + size_t(nt0) * 64;

    float16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto load_a = [&](int ks) -> VT {
        VT a = vec_zero<T>();
        if (rvalid && ks < KS && ks * 2 * V + g * V < K) {
            a = *reinterpret_cast<const VT*>(ap + ks * 2 * V);
            if constexpr (GATE) {
                float f[V];
                vec_to_float<T>(a, f);
#pragma unroll
                for (int i = 0; i < V; i += 4) {
                    const float4v gv = *reinterpret_cast<const float4v*>(gp + ks * 2 * V + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
                }
                a = float_to_vec<T>(f);
            }
        }
        return a;
    };

    VT wreg[CPT];
    auto fetch_w = [&](int grp) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int i = c * 256 + tid;
            const int u = i / (NT * 64);
            const int j = i - u * (NT * 64);
            const int ks = grp * UK + u;
            wreg[c] = (i < STAGE_VECS && ks < KS && nt0 + (j >> 6) < NTILES) ? wsrc[size_t(ks) * NTILES * 64 + j]
                                                                              : vec_zero<T>();
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
    VT areg[UK];
    fetch_w(0);
#pragma unroll
    for (int u = 0; u < UK; ++u) areg[u] = load_a(u);
    store_w(0);
    __syncthreads();
    for (int grp = 0; grp < G; ++grp) {
        const bool more = grp + 1 < G;
        VT anext[UK];
        if (more) {
            fetch_w(grp + 1);
#pragma unroll
            for (int u = 0; u < UK; ++u) anext[u] = load_a((grp + 1) * UK + u);
        }
        const VT* wl = s_w + (grp & 1) * STAGE_VECS + lane;
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            if (grp * UK + u < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt0 + t < NTILES) Mfma<T>::step(wl[(u * NT + t) * 64], areg[u], acc[t]);
            }
        }
        if (more) {
            store_w((grp + 1) & 1);
#pragma unroll
            for (int u = 0; u < UK; ++u) areg[u] = anext[u];
        }
        __syncthreads();
    }

    // ---- epilogue: bias / activation in f32, LDS transpose, 16-byte row-contiguous stores ----
    float* s_e = smem + wave * 32 * SROW;                   // aliases s_w: every wave is past the last barrier
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TPS) {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = t0 + tt;
            if (t < NT && nt0 + t < NTILES) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int n = (nt0 + t) * 32 + 8 * qq + 4 * g;
                    float4v y;
                    if (n < N) {
                        const float4v bv = *reinterpret_cast<const float4v*>(bias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[t < NT ? t : 0][4 * qq + r] + bv[r];
                            if constexpr (ACT == ACT_SWISH) v = swish_f<IsF32<T>::value>(v);
                            y[r] = v;
                        }
                    } else {
                        y = float4v{0.f, 0.f, 0.f, 0.f};
                    }
                    *reinterpret_cast<float4v*>(s_e + (lane & 31) * SROW + tt * 32 + 8 * qq + 4 * g) = y;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx >> 3, ch = idx & 7;
            const int n = (nt0 + t0) * 32 + ch * CW;
            const int rowg = m0 + r;
            if (rowg < M && n < N && t0 + ((ch * CW) >> 5) < NT) {
                float y[CW];
#pragma unroll
                for (int c4 = 0; c4 < CW; c4 += 4) {
                    const float4v v = *reinterpret_cast<const float4v*>(s_e + r * SROW + ch * CW + c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[c4 + j] = v[j];
                }
                if constexpr (RES) {
                    const VT rv = *reinterpret_cast<const VT*>(res + size_t(rowg) * N + n);
#pragma unroll
                    for (int j = 0; j < CW; ++j) y[j] += float(rv[j]);
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
                                                              const float* __restrict__ gate,
                                                              const T* __restrict__ res, T* __restrict__ out, int M,
                                                              int K, int N, int HW) {
    const int n4 = N / 4;
    const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= size_t(M) * n4) return;
    const int m = int(idx / n4);
    const int n0 = int(idx - size_t(m) * n4) * 4;
    const T* ap = A + size_t(m) * K;
    const float* gp = GATE ? gate + size_t(m / HW) * K : nullptr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        float a = float(ap[k]);
        if constexpr (GATE) a = float(T(a * gp[k]));
        const float4v w = *reinterpret_cast<const float4v*>(Wd + size_t(k) * N + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(a, w[r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = acc[r] + bias[n0 + r];
        if constexpr (ACT == ACT_SWISH) y = swish_f<IsF32<T>::value>(y);
        if constexpr (RES) y += float(res[size_t(m) * N + n0 + r]);
        out[size_t(m) * N + n0 + r] = T(y);
    }
}

// A K >= 320 conv takes the 2 x 2 register-blocked split-K kernel (same bits as the 1 x 1 form) when
// that still launches enough 64 x 64 workgroups to spread over the chip
// (>= 128 of them; measured: B=64 +5 %, B=512 +8 %, B <= 16 unchanged)
bool use_split2(int M, int NTILES) { return ceil_div(M, 64) * ceil_div(NTILES, 2) >= 128; }

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

template <typename T, int NT, int U, int SK, bool GATE, bool RES, int ACT>
void launch_mfma(const PwArgs& a, int MT, int NCH, hipStream_t stream) {
    const int blocks = 8 * ceil_div(MT, 8) * NCH;
    hipLaunchKernelGGL((whenet_pw_kernel<T, NT, U, SK, GATE, RES, ACT>), dim3(blocks), dim3(256), 0, stream,
                       static_cast<const T*>(a.a), static_cast<const T*>(a.wp), a.bias, a.gate,
                       static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.KS, a.NTILES, a.HW, MT,
                       NCH);
}

template <typename T, int NT, bool GATE, bool RES, int ACT>
void launch_tile(const PwArgs& a, int MT, int NCH, hipStream_t stream) {
    const int blocks = 8 * ceil_div(MT, 8) * NCH;
    hipLaunchKernelGGL((whenet_pw_tile_kernel<T, NT, GATE, RES, ACT>), dim3(blocks), dim3(256), 0, stream,
                       static_cast<const T*>(a.a), static_cast<const T*>(a.wp), a.bias, a.gate,
                       static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.KS, a.NTILES, a.HW, MT,
                       NCH);
}

template <typename T, bool GATE, bool RES, int ACT>
void launch_variant(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    if (impl == 1) {
        const size_t work = size_t(a.M) * (a.N / 4);
        hipLaunchKernelGGL((whenet_pw_check_kernel<T, GATE, RES, ACT>), dim3(unsigned((work + 255) / 256)), dim3(256),
                           0, stream, static_cast<const T*>(a.a), a.wdense, a.bias, a.gate,
                           static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.HW);
        return;
    }
    const PwChoice ch = choose_pw(a, num_cus);
    if (ch.kind == 1) {
        if (use_split2(a.M, a.NTILES)) {
            const int MT2 = ceil_div(a.M, 64), NCH2 = ceil_div(a.NTILES, 2);
            hipLaunchKernelGGL((whenet_pw_split2_kernel<T, GATE, RES, ACT>), dim3(8 * ceil_div(MT2, 8) * NCH2), dim3(256),
                               0, stream, static_cast<const T*>(a.a), static_cast<const T*>(a.wp), a.bias, a.gate,
                               static_cast<const T*>(a.res), static_cast<T*>(a.out), a.M, a.K, a.N, a.KS, a.NTILES,
                               a.HW, MT2, NCH2);
            return;
        }
        launch_mfma<T, 1, 8, 4, GATE, RES, ACT>(a, ceil_div(a.M, 32), a.NTILES, stream);
        return;
    }
    const int MT = ceil_div(a.M, 128);
    switch (ch.NT) {
        case 1: launch_tile<T, 1, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
        case 2: launch_tile<T, 2, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
        case 3: launch_tile<T, 3, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
        case 4: launch_tile<T, 4, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
        case 5: launch_tile<T, 5, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
        default: launch_tile<T, 6, GATE, RES, ACT>(a, MT, ch.NCH, stream); break;
    }
}

template <typename T>
void launch_dtype(const PwArgs& a, int impl, int num_cus, hipStream_t stream) {
    const bool gate = a.gate != nullptr, res = a.res != nullptr;
    // the network uses exactly three flavours: expand/head (swish), project (gate), project+skip
    if (!gate && !res && a.act == ACT_SWISH) launch_variant<T, false, false, ACT_SWISH>(a, impl, num_cus, stream);
    else if (gate && !res && a.act == ACT_NONE) launch_variant<T, true, false, ACT_NONE>(a, impl, num_cus, stream);
    else if (gate && res && a.act == ACT_NONE) launch_variant<T, true, true, ACT_NONE>(a, impl, num_cus, stream);
    else throw Error(WHENET_EINVAL, "pointwise: unsupported epilogue combination");
}

}  // namespace

void launch_pw(const PwArgs& a, int dtype, int impl, int num_cus, hipStream_t stream) {
    WHENET_REQUIRE(a.N % 4 == 0 && a.M > 0, WHENET_EINVAL, "pointwise: bad shape");
    if (dtype == WHENET_F16) launch_dtype<half_t>(a, impl, num_cus, stream);
    else launch_dtype<float>(a, impl, num_cus, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

std::string kernel_name_pw(const PwArgs& a, int dtype, int impl, int num_cus) {
    // the instantiation launch_pw() will pick, spelled as rocprofv3 prints it
    const char* t = dtype == WHENET_F16 ? "_Float16" : "float";
    const char* gate = a.gate ? "true" : "false";
    const char* res = a.res ? "true" : "false";
    char buf[96];
    if (impl == 1) {
        std::snprintf(buf, sizeof(buf), "whenet_pw_check_kernel<%s, %s, %s, %d>", t, gate, res, a.act);
    } else {
        const PwChoice ch = choose_pw(a, num_cus);
        if (ch.kind == 1 && use_split2(a.M, a.NTILES))
            std::snprintf(buf, sizeof(buf), "whenet_pw_split2_kernel<%s, %s, %s, %d>", t, gate, res, a.act);
        else if (ch.kind == 1) std::snprintf(buf, sizeof(buf), "whenet_pw_kernel<%s, 1, 8, 4, %s, %s, %d>", t, gate, res, a.act);
        else std::snprintf(buf, sizeof(buf), "whenet_pw_tile_kernel<%s, %d, %s, %s, %d>", t, ch.NT, gate, res, a.act);
    }
    return buf;
}

}  // namespace whenet
