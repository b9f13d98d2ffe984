// "Tail" megakernel: MBConv blocks 7..16 (14x14 and 7x7 maps) + head conv + GAP + Dense heads +
// decode, ONE workgroup per crop, ONE launch instead of 42.
//
// Reference: the same stages as pw.hip / dw.hip / se.hip / head.hip -- efficientnet 0.0.4
// MBConvBlock x10, head Conv1x1(1280)+BN+Swish (instantiated by /root/reference/whenet.py:8),
// GlobalAveragePooling2D + Dense 120/66/66 (whenet.py:10-13), softmax-expectation decode
// (whenet.py:28-33, utils.py:7-11).
//
// Why: on these maps a layer moves 20-260 KB per crop; as separate launches every layer costs a
// kernel boundary plus a cold latency chain (10-30 us measured per launch at batch 64, 750 of the
// 1160 us of a forward) while the chip idles.  Here a crop's activations never leave its CU:
//   X   block input/output [HW][C]           LDS (row pitch C*sizeof(T)+16: conflict-free
//                                            16-byte MFMA fragment reads)
//   E   expanded tensor, one channel chunk,  LDS, zero halo materialises TF 'SAME' padding;
//       [(Ho-1)s+k]^2 pixels x CC channels   written by the expand GEMM's epilogue, read by the
//                                            depthwise taps (lane = 4 channels x strip of 7 px)
//   D   depthwise output, all channels       global scratch (<= 263 KB per crop, L2-resident):
//                                            the project GEMM needs every channel after the gate
//   SE  channel sums -> gate                 LDS, f32, fixed summation order
// Weights stream from L2 in the host-packed MFMA fragment order (snapshot.h), 4 k-steps of loads
// in flight per wave.  GEMM work is distributed as (32-row strip, 32-channel tile) tasks over
// the 12 waves; MFMA mapping and epilogue arithmetic are those of pw.hip (transposed product).
// Per crop: 99.6 M MACs, 6.6 MB (f16) of weights read through L2, 62.7 KB in, 1 KB out.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

constexpr int NTHR = 768;     // 12 waves: 149 VGPRs, no scratch; 1024 lanes would cap at 128 VGPRs and spill
constexpr int NWAVE = 12;
constexpr int DENSE_WAVES = 8;   // 1280 / 8 = 160 features per wave in the Dense phase
constexpr int P = 7;
constexpr int VC = 4;
constexpr int SUM_FLOATS = 1152;

template <typename T> struct TailCfg;
// CC14 / CC7: expanded channels per chunk on the 14x14 / 7x7 maps; RED: strip partial sums
// (28 x CC14 | 7 x CC7 floats); DWW: depthwise taps of a chunk (25 x CC floats)
template <> struct TailCfg<half_t> { static constexpr int CC14 = 96, CC7 = 256, RED = 2688, DWW = 6400; };
template <> struct TailCfg<float> { static constexpr int CC14 = 32, CC7 = 128, RED = 896, DWW = 3200; };

__host__ __device__ constexpr int align16(int x) { return (x + 15) & ~15; }

// ---- one (strip, tile) GEMM task: acc += sum_k W[tile][k] * act[row][k] ---------------------
template <typename T, int U, typename LoadA>
__device__ __forceinline__ void gemm_task(float16v& acc, int KS, const typename Vec<T>::type* __restrict__ wfrag,
                                          int wstride, LoadA&& load_a) {
    using VT = typename Vec<T>::type;
    for (int ks = 0; ks < KS; ks += U) {
        VT w[U], a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = (ks + u < KS) ? wfrag[size_t(ks + u) * wstride] : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = (ks + u < KS) ? load_a(ks + u) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u) Mfma<T>::step(w[u], a[u], acc);
    }
}

// ---- depthwise taps of one chunk: E (LDS) -> D (global), per-strip channel sums -> s_red -----
template <typename T, int K, int S>
__device__ __forceinline__ void dw_chunk(const unsigned char* __restrict__ E, int EW, int EP, T* __restrict__ D,
                                         const float* __restrict__ s_dww, const float* __restrict__ bd,
                                         float* __restrict__ s_red, int Ho, int C, int c0, int ccur, int tid) {
    using VCT = T __attribute__((ext_vector_type(VC)));
    constexpr int NIX = (P - 1) * S + K;
    const int CG = ccur / VC;
    const int nstrip = Ho * (Ho / P);
    const int cg = tid % CG;
    const int sidx = tid / CG;
    if (sidx >= nstrip) return;
    const int spr = Ho / P;                         // strips per output row
    const int oy = sidx / spr;
    const int sx = sidx - oy * spr;
    float acc[P][VC];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int v = 0; v < VC; ++v) acc[p][v] = 0.0f;
#pragma unroll 1   // one kernel row at a time: keeps this phase's register footprint small
    for (int ky = 0; ky < K; ++ky) {
        float wr[K][VC];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float4v wv = *reinterpret_cast<const float4v*>(s_dww + (ky * K + kx) * ccur + cg * VC);
#pragma unroll
            for (int v = 0; v < VC; ++v) wr[kx][v] = wv[v];
        }
        const unsigned char* row = E + size_t((oy * S + ky) * EW + sx * P * S) * EP + cg * VC * sizeof(T);
#pragma unroll
        for (int ix = 0; ix < NIX; ++ix) {
            const VCT xv = *reinterpret_cast<const VCT*>(row + size_t(ix) * EP);
            float x[VC];
#pragma unroll
            for (int v = 0; v < VC; ++v) x[v] = float(xv[v]);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int d = ix - kx;
                if (d >= 0 && (d % S) == 0 && (d / S) < P) {
#pragma unroll
                    for (int v = 0; v < VC; ++v) acc[d / S][v] = fmaf(x[v], wr[kx][v], acc[d / S][v]);
                }
            }
        }
    }
    const float4v bs = *reinterpret_cast<const float4v*>(bd + c0 + cg * VC);
    float sum[VC] = {0.f, 0.f, 0.f, 0.f};
    T* dst = D + (size_t(oy) * Ho + sx * P) * C + c0 + cg * VC;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        VCT o;
#pragma unroll
        for (int v = 0; v < VC; ++v) {
            const float y = swish_f<IsF32<T>::value>(acc[p][v] + bs[v]);
            sum[v] += y;
            o[v] = T(y);
        }
        *reinterpret_cast<VCT*>(dst + size_t(p) * C) = o;
    }
#pragma unroll
    for (int v = 0; v < VC; ++v) s_red[sidx * ccur + cg * VC + v] = sum[v];
}

template <typename T>
__global__ __launch_bounds__(NTHR) void whenet_tail_kernel(TailArgs a) {
    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    using OT = T __attribute__((ext_vector_type(4)));
    constexpr int SZ = int(sizeof(T));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_sum = reinterpret_cast<float*>(smem + a.fixed_off);     // [1152] channel sums -> means
    float* s_gate = s_sum + SUM_FLOATS;                                // [1152]
    float* s_r = s_gate + SUM_FLOATS;                                  // [64]
    float* s_red = s_r + 64;                                           // [RED_FLOATS] strip partial sums
    float* s_dww = s_red + TailCfg<T>::RED;                                 // [DWW_FLOATS] depthwise taps of the chunk
    unsigned char* X = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int b = blockIdx.x;
    T* D = static_cast<T*>(a.d_scratch) + size_t(b) * a.d_stride;
    // debug timeline (option: op_tail timing): crop 0, lane 0 stamps the 100 MHz wall clock
    auto stamp = [&](int slot) {
        if (a.timing != nullptr && b == 0 && tid == 0) a.timing[slot] = wall_clock64();
    };
    stamp(0);

    // ---- X <- block-7 input [196][80] from global ------------------------------------------
    {
        const int C = a.first.cin, HW = a.first.h_in * a.first.h_in;
        const int pitch = C * SZ + 16, vpr = C * SZ / 16;
        const VT* src = reinterpret_cast<const VT*>(static_cast<const T*>(a.x_in) + size_t(b) * HW * C);
        for (int i = tid; i < HW * vpr; i += NTHR) {
            const int r = i / vpr, v = i - r * vpr;
            *reinterpret_cast<VT*>(X + size_t(r) * pitch + v * 16) = src[i];
        }
    }
    __syncthreads();

    for (int bi = 0; bi < a.nblk; ++bi) {
        const TailBlock B = a.blk[bi];          // uniform: scalar loads from the device table
        const int HWi = B.h_in * B.h_in, HWo = B.h_out * B.h_out;
        const int pin = B.cin * SZ + 16, pout = B.cout * SZ + 16;
        const int EW = (B.h_out - 1) * B.s + B.k;
        const int CC = (B.h_in == 14) ? TailCfg<T>::CC14 : TailCfg<T>::CC7;
        const int EP = CC * SZ + 16;
        unsigned char* E = smem + align16(HWi * pin);
        const int nstrip_i = (HWi + 31) >> 5, nstrip_o = (HWo + 31) >> 5;

        for (int i = tid; i < EW * EW * EP / 16; i += NTHR) reinterpret_cast<VT*>(E)[i] = vec_zero<T>();
        __syncthreads();
        stamp(1 + bi * 8 + 0);

        // ================= phase 1: expand (MFMA) -> E, depthwise -> D, channel sums =========
        for (int c0 = 0; c0 < B.cexp; c0 += CC) {
            const int ccur = (B.cexp - c0 < CC) ? (B.cexp - c0) : CC;
            const int ntile = ccur >> 5;
            // depthwise taps of this chunk -> LDS (consumed after the barrier below)
            for (int i = tid; i < B.k * B.k * ccur; i += NTHR) {
                const int tap = i / ccur, c = i - tap * ccur;
                s_dww[i] = B.wd[size_t(tap) * B.cexp + c0 + c];
            }
            for (int t = wave; t < nstrip_i * ntile; t += NWAVE) {
                const int tile = t / nstrip_i, strip = t - tile * nstrip_i;
                const int p = strip * 32 + lm;
                const bool valid = p < HWi;
                const unsigned char* xrow = X + size_t(valid ? p : 0) * pin + g * V * SZ;
                float16v acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const VT* wf = reinterpret_cast<const VT*>(B.we) + size_t((c0 >> 5) + tile) * 64 + lane;
                gemm_task<T, 8>(acc, B.kse, wf, B.nte * 64, [&](int ks) -> VT {
                    return valid ? *reinterpret_cast<const VT*>(xrow + size_t(ks) * 2 * V * SZ) : vec_zero<T>();
                });
                if (valid) {
                    const int py = p / B.h_in, px = p - py * B.h_in;
                    unsigned char* epix = E + size_t((py + B.pad) * EW + px + B.pad) * EP;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int nl = tile * 32 + 8 * qq + 4 * g;
                        const float4v bv = *reinterpret_cast<const float4v*>(B.be + c0 + nl);
                        OT o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = T(swish_f<IsF32<T>::value>(acc[4 * qq + r] + bv[r]));
                        *reinterpret_cast<OT*>(epix + nl * SZ) = o;
                    }
                }
            }
            __syncthreads();
            if (c0 == 0) stamp(1 + bi * 8 + 1);
            if (B.k == 3) dw_chunk<T, 3, 1>(E, EW, EP, D, s_dww, B.bd, s_red, B.h_out, B.cexp, c0, ccur, tid);
            else if (B.s == 1) dw_chunk<T, 5, 1>(E, EW, EP, D, s_dww, B.bd, s_red, B.h_out, B.cexp, c0, ccur, tid);
            else dw_chunk<T, 5, 2>(E, EW, EP, D, s_dww, B.bd, s_red, B.h_out, B.cexp, c0, ccur, tid);
            __syncthreads();
            if (c0 == 0) stamp(1 + bi * 8 + 2);
            if (tid < ccur) {
                const int nstrip = B.h_out * (B.h_out / P);
                float t = 0.0f;
                for (int s = 0; s < nstrip; ++s) t += s_red[s * ccur + tid];
                s_sum[c0 + tid] = t;
            }
        }
        __syncthreads();

        stamp(1 + bi * 8 + 3);
        // ================= phase 2: squeeze-excite gate ======================================
        {
            const int C = B.cexp, R = B.r;
            const float inv_hw = 1.0f / float(HWo);
            for (int c = tid; c < C; c += NTHR) s_sum[c] *= inv_hw;
            __syncthreads();
            for (int j = wave; j < R; j += NWAVE) {
                const float* wrow = B.w1t + size_t(j) * C;
                float p4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int cb = lane; cb < C; cb += 256) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = cb + 64 * u;
                        if (c < C) p4[u] = fmaf(s_sum[c], wrow[c], p4[u]);
                    }
                }
                float t = (p4[0] + p4[1]) + (p4[2] + p4[3]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                if (lane == 0) s_r[j] = swish_f<true>(t + B.b1[j]);
            }
            __syncthreads();
            for (int c = tid; c < C; c += NTHR) {
                float t0 = B.b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
                int j = 0;
                for (; j + 4 <= R; j += 4) {
                    t0 = fmaf(s_r[j], B.w2[size_t(j) * C + c], t0);
                    t1 = fmaf(s_r[j + 1], B.w2[size_t(j + 1) * C + c], t1);
                    t2 = fmaf(s_r[j + 2], B.w2[size_t(j + 2) * C + c], t2);
                    t3 = fmaf(s_r[j + 3], B.w2[size_t(j + 3) * C + c], t3);
                }
                for (; j < R; ++j) t0 = fmaf(s_r[j], B.w2[size_t(j) * C + c], t0);
                s_gate[c] = sigmoid_f<true>((t0 + t1) + (t2 + t3));
            }
            __syncthreads();
        }

        stamp(1 + bi * 8 + 4);
        // ================= phase 3: project (MFMA) D*gate -> X (+skip) ========================
        for (int t = wave; t < nstrip_o * B.ntp; t += NWAVE) {
            const int tile = t / nstrip_o, strip = t - tile * nstrip_o;
            const int p = strip * 32 + lm;
            const bool valid = p < HWo;
            const T* drow = D + size_t(valid ? p : 0) * B.cexp + g * V;
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const VT* wf = reinterpret_cast<const VT*>(B.wp) + size_t(tile) * 64 + lane;
            gemm_task<T, 4>(acc, B.ksp, wf, B.ntp * 64, [&](int ks) -> VT {
                if (!valid) return vec_zero<T>();
                const VT av = *reinterpret_cast<const VT*>(drow + size_t(ks) * 2 * V);
                float f[V];
                vec_to_float<T>(av, f);
                const float* gp = s_gate + ks * 2 * V + g * V;
#pragma unroll
                for (int i = 0; i < V; i += 4) {
                    const float4v gv = *reinterpret_cast<const float4v*>(gp + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i + j] *= gv[j];
                }
                return float_to_vec<T>(f);
            });
            if (valid) {
                unsigned char* xo = X + size_t(p) * pout;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int n = tile * 32 + 8 * qq + 4 * g;
                    if (n < B.cout) {
                        const float4v bv = *reinterpret_cast<const float4v*>(B.bp + n);
                        float y[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = acc[4 * qq + r] + bv[r];
                        if (B.has_skip) {
                            const OT rv = *reinterpret_cast<const OT*>(xo + n * SZ);
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
                        }
                        OT o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
                        // non-skip blocks change the row pitch: every wave must be done READING the
                        // old X (nobody reads it in this phase) -- only positions are rewritten
                        *reinterpret_cast<OT*>(xo + n * SZ) = o;
                    }
                }
            }
        }
        __syncthreads();
        stamp(1 + bi * 8 + 5);
    }

    const int HWl = a.last.h_out * a.last.h_out, Cl = a.last.cout, pl = Cl * SZ + 16;
    if (a.dump_x != nullptr) {          // test hook: the block chain's output, [HW][C] as f32
        float* dst = a.dump_x + size_t(b) * HWl * Cl;
        for (int i = tid; i < HWl * Cl; i += NTHR) {
            const int r = i / Cl, c = i - r * Cl;
            dst[i] = float(*reinterpret_cast<const T*>(X + size_t(r) * pl + c * SZ));
        }
        return;
    }

    // ================= head conv (MFMA) + BN + Swish, fused GAP ===============================
    float* s_fp = reinterpret_cast<float*>(smem + align16(HWl * pl));      // [2][1280] strip partials
    float* s_feat = s_fp + 2 * FEAT;                                       // [1280]
    float* s_part = s_feat + FEAT;                                         // [DENSE_WAVES][256]
    float* s_logit = s_part + DENSE_WAVES * 256;                                 // [256]
    {
        const int nstrip = (HWl + 31) >> 5;                                // 2
        for (int t = wave; t < nstrip * a.nth; t += NWAVE) {
            const int tile = t / nstrip, strip = t - tile * nstrip;
            const int p = strip * 32 + lm;
            const bool valid = p < HWl;
            const unsigned char* xrow = X + size_t(valid ? p : 0) * pl + g * V * SZ;
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const VT* wf = reinterpret_cast<const VT*>(a.wh) + size_t(tile) * 64 + lane;
            gemm_task<T, 8>(acc, a.ksh, wf, a.nth * 64, [&](int ks) -> VT {
                return valid ? *reinterpret_cast<const VT*>(xrow + size_t(ks) * 2 * V * SZ) : vec_zero<T>();
            });
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int n = tile * 32 + 8 * qq + 4 * g;
                const float4v bv = *reinterpret_cast<const float4v*>(a.bh + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = valid ? swish_f<IsF32<T>::value>(acc[4 * qq + r] + bv[r]) : 0.0f;
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (lm == 0) s_fp[strip * FEAT + n + r] = v;
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < FEAT; c += NTHR) {
            const float f = (s_fp[c] + s_fp[FEAT + c]) * (1.0f / 49.0f);
            s_feat[c] = f;
            if (a.feat != nullptr) a.feat[size_t(b) * FEAT + c] = f;
        }
        __syncthreads();
    }

    stamp(90);
    // ================= Dense 120|66|66 (whenet.py:11-13) =========================================
    {
        const int c_lo = wave * (FEAT / DENSE_WAVES);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < DENSE_WAVES && lane < N_LOGITS / 4) {
            const float* wr = a.wdense + size_t(c_lo) * N_LOGITS + lane * 4;
#pragma unroll 8
            for (int c = 0; c < FEAT / DENSE_WAVES; ++c) {
                const float f = s_feat[c_lo + c];
                const float4v wv = *reinterpret_cast<const float4v*>(wr + size_t(c) * N_LOGITS);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(f, wv[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_part[wave * 256 + lane * 4 + i] = acc[i];
        }
        __syncthreads();
        if (tid < N_LOGITS) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < DENSE_WAVES; ++w) t += s_part[w * 256 + tid];
            t += a.bdense[tid];
            s_logit[tid] = t;
            if (a.logits != nullptr) a.logits[size_t(b) * N_LOGITS + tid] = t;
        }
        __syncthreads();
    }

    stamp(91);
    // ================= decode (utils.py:7-11, whenet.py:28-33): wave h <-> head h ================
    if (wave >= 3) return;
    const int lo = (wave == 0) ? 0 : (wave == 1 ? N_YAW : N_YAW + N_PITCH);
    const int nb = (wave == 0) ? N_YAW : N_PITCH;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int j = lane; j < nb; j += 64) {
        const float v = s_logit[lo + j];
        if (v > mx) { mx = v; mi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mx, off, 64);
        const int oi = __shfl_xor(mi, off, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    float se = 0.0f;
    float e[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = lane + 64 * i;
        e[i] = (j < nb) ? expf(s_logit[lo + j] - mx) : 0.0f;
        se += e[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
    float ex = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = lane + 64 * i;
        if (j < nb) ex += (e[i] / se) * float(j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ex += __shfl_xor(ex, off, 64);
    if (lane == 0) {
        a.ypr[size_t(b) * 3 + wave] = ex * 3.0f - ((wave == 0) ? 180.0f : 99.0f);
        if (a.argmax != nullptr) a.argmax[size_t(b) * 3 + wave] = mi;
    }
}

template <typename T>
size_t tail_lds_bytes(const TailArgs& a, const TailBlock* host_blk, int* fixed_off) {
    const int SZ = int(sizeof(T));
    int need = 0;
    for (int bi = 0; bi < a.nblk; ++bi) {
        const TailBlock B = a.blk[bi];          // uniform: scalar loads from the device table
        const int HWi = B.h_in * B.h_in, HWo = B.h_out * B.h_out;
        const int EW = (B.h_out - 1) * B.s + B.k;
        const int CC = (B.h_in == 14) ? TailCfg<T>::CC14 : TailCfg<T>::CC7;
        const int x_in = align16(HWi * (B.cin * SZ + 16));
        const int x_out = align16(HWo * (B.cout * SZ + 16));
        const int e = EW * EW * (CC * SZ + 16);
        need = std::max(need, std::max(x_in + e, x_out));
    }
    const TailBlock& L = host_blk[a.nblk - 1];
    const int head = align16(L.h_out * L.h_out * (L.cout * SZ + 16)) + (3 * FEAT + DENSE_WAVES * 256 + 256) * 4;
    need = std::max(need, head);
    need = align16(need);
    *fixed_off = need;
    return size_t(need) + size_t(2 * SUM_FLOATS + 64 + TailCfg<T>::RED + TailCfg<T>::DWW) * sizeof(float);
}

template <typename T>
void launch_t(TailArgs a, const TailBlock* host_blk, hipStream_t stream) {
    int fixed = 0;
    const size_t lds = tail_lds_bytes<T>(a, host_blk, &fixed);
    WHENET_REQUIRE(lds <= 160 * 1024, WHENET_EINVAL, "tail kernel: LDS budget exceeded");
    a.fixed_off = fixed;
    static bool attr_set[64] = {};
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {      // > 64 KiB of dynamic LDS needs the opt-in
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_tail_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(whenet_tail_kernel<T>, dim3(a.n), dim3(NTHR), lds, stream, a);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace

void launch_tail(const TailArgs& a, const TailBlock* host_blk, int dtype, hipStream_t stream) {
    WHENET_REQUIRE(a.nblk >= 1 && a.nblk <= 10 && a.n >= 1 && a.blk != nullptr, WHENET_EINVAL, "tail kernel: bad arguments");
    if (dtype == WHENET_F16) launch_t<half_t>(a, host_blk, stream);
    else launch_t<float>(a, host_blk, stream);
}

}  // namespace whenet
