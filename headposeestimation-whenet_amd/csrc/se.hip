// SEBlock (squeeze-excite) gate, one workgroup per crop.
//
// Reference: efficientnet 0.0.4 SEBlock as instantiated by /root/reference/whenet.py:8
// (SURVEY.md Appendix B): mean over H,W (keepdims) -> Conv2D(reduced, 1x1, bias) -> Swish ->
// Conv2D(C, 1x1, bias) -> sigmoid; the gate multiplies the depthwise output (that multiply is
// fused into the project GEMM's operand load, pw.hip).  `reduced` = int(0.25 * block INPUT
// filters).  All f32, fixed summation order -> bitwise reproducible.
//
// 2*C*R MACs per crop (0.2 % of the network): the kernel is pure latency, so it is laid out as
// ONE dependent memory round trip where the shapes allow it:
//   * every weight load is independent of the data, so the fc1 rows (and, when they fit the
//     register budget, the lane's fc2 row) are issued BEFORE the tile partial sums; the barriers
//     order LDS traffic only (lds_barrier), so those loads stay in flight across them;
//   * squeeze: the 8 running sums of a channel (tiles i = u mod 8) are spread over 8 lanes when
//     8*C <= 1024 (blocks 1-2, 56-64 tiles) so that all tile loads are in flight at once, and
//     combined through LDS in the same fixed order; otherwise a lane issues 16 tiles per trip;
//   * excite: the se_expand kernel is stored channel-major [C][RP]: a lane's whole row is RP/4
//     16-byte loads from one 64..192-byte segment (the [R][C] form touched RP pages per wave).
// f32 throughout, fixed summation order (bitwise reproducible, independent of the batch).
#include "device_math.h"
#include "kernels.h"
#include "se_device.h"
#include "stamps.h"

namespace whenet {

namespace {

// The gate is stored in the type of the activations it multiplies (pw.hip applies it as a T x T product).
__device__ __forceinline__ void store_gate(void* gate, int f16, size_t i, float v) {
    if (f16) static_cast<half_t*>(gate)[i] = half_t(v);
    else static_cast<float*>(gate)[i] = v;
}

template <int RP>
__global__ __launch_bounds__(1024) void whenet_se_kernel(const float* __restrict__ partial, int ntiles, float inv_hw,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        const float* __restrict__ w2c, const float* __restrict__ b2,
                                                        void* __restrict__ gate, int C, int R, int gate_f16) {
    constexpr int NW = 16, NTHR = NW * 64;
    constexpr int JPW = (RP + NW - 1) / NW;          // fc1 outputs per wave (<= 3)
    constexpr int CPL = 1152 / 64;                    // channel slots per lane (18)
    constexpr int NCI = (RP == 48) ? 2 : 1;           // channels per lane in excite (C <= 1024 unless RP = 48)
    constexpr bool PRE2 = RP <= 28;                   // fc2 row prefetched before the fc1 arithmetic
    __shared__ float s_mean[1152];
    __shared__ float s_r[RP];
    __shared__ float s_t[8][128];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    STAMP(0);
    // fc1 rows: wave w owns outputs j = w, w+16, ..; its lanes stride over c (coalesced)
    float wv1[JPW][CPL];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int j = wave + NW * jj;
        const float* wrow = w1t + size_t(j < R ? j : 0) * C;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int c = lane + 64 * u;
            wv1[jj][u] = (j < R && c < C) ? wrow[c] : 0.f;
        }
    }

    // squeeze: mean over the map = (sum of the producing kernel's tile partials) / (H*W), as 8
    // running sums t[u] over tiles i = u mod 8, combined ((t0+t1)+(t2+t3))+((t4+t5)+(t6+t7))
    const float* pp = partial + size_t(b) * ntiles * C;
    const bool wide = 8 * C <= NTHR && ntiles <= 64;          // (uniform)
    if (wide) {
        const int u = tid / C, c = tid - u * C;
        if (u < 8) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (u + 8 * k < ntiles) ? pp[size_t(u + 8 * k) * C + c] : 0.f;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (u + 8 * k < ntiles) t += x[k];
            s_t[u][c] = t;
        }
    } else {
        for (int c = tid; c < C; c += NTHR) {
            float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < ntiles; i += 16) {
                float x0[8], x1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x0[u] = (i + u < ntiles) ? pp[size_t(i + u) * C + c] : 0.f;
                    x1[u] = (i + 8 + u < ntiles) ? pp[size_t(i + 8 + u) * C + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i + u < ntiles) t[u] += x0[u];
                    if (i + 8 + u < ntiles) t[u] += x1[u];
                }
            }
            s_mean[c] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) * inv_hw;
        }
    }

    STAMP(1);
    // fc2 row(s) of this lane's channel(s)
    float wv2[NCI][RP];
    auto load_w2 = [&]() {
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) {
            const int c = tid + ci * NTHR;
            const float4v* wrow = reinterpret_cast<const float4v*>(w2c + size_t(c < C ? c : 0) * RP);
#pragma unroll
            for (int j = 0; j < RP; j += 4) {
                const float4v v = wrow[j >> 2];
#pragma unroll
                for (int q = 0; q < 4; ++q) wv2[ci][j + q] = v[q];
            }
        }
    };
    if constexpr (PRE2) load_w2();

    if (wide) {
        lds_barrier();
        if (tid < C)
            s_mean[tid] = (((s_t[0][tid] + s_t[1][tid]) + (s_t[2][tid] + s_t[3][tid])) +
                           ((s_t[4][tid] + s_t[5][tid]) + (s_t[6][tid] + s_t[7][tid]))) * inv_hw;
    }
    lds_barrier();
    STAMP(2);

    // reduce: r[j] = swish(b1[j] + sum_c mean[c] * W1[c][j]); one 6-step shuffle tree per output
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int j = wave + NW * jj;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int c = lane + 64 * u;
            p[u & 3] = fmaf((c < C) ? s_mean[c] : 0.f, wv1[jj][u], p[u & 3]);
        }
        float t = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0 && j < RP) s_r[j] = (j < R) ? swish_f<true>(t + b1[j]) : 0.f;
    }
    STAMP(3);
    if constexpr (!PRE2) load_w2();
    lds_barrier();
    STAMP(4);

    // excite: gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c])
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci) {
        const int c = tid + ci * NTHR;
        if (c < C) {
            float t0 = b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int j = 0; j < RP; j += 4) {
                t0 = fmaf(s_r[j], wv2[ci][j], t0);
                t1 = fmaf(s_r[j + 1], wv2[ci][j + 1], t1);
                t2 = fmaf(s_r[j + 2], wv2[ci][j + 2], t2);
                t3 = fmaf(s_r[j + 3], wv2[ci][j + 3], t3);
            }
            store_gate(gate, gate_f16, size_t(b) * C + c, sigmoid_f<true>((t0 + t1) + (t2 + t3)));
        }
    }
    STAMP(5);
}

// Second half of the SEBlock for blocks whose producer (front.hip) already applied the reduce conv
// to its tile/chunk channel sums: r[j] = swish(b1[j] + (sum over the crop's np partial vectors) /
// (H*W)); gate[c] = sigmoid(b2[c] + sum_j r[j] * W2[j][c]).  SPLIT workgroups per crop each take a
// channel slice (one CU streams ~50 GB/s: 221 KB of excite kernel for C = 1152 is spread over 4).
// The partial vectors and the lane's excite row are independent loads: one memory round trip.
template <int RP>
__global__ __launch_bounds__(256) void whenet_se_excite_kernel(const float* __restrict__ rpart, int np, float inv_hw,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ w2c,
                                                               const float* __restrict__ b2, void* __restrict__ gate,
                                                               int C, int R, int SPLIT, int gate_f16) {
    constexpr int NTHR = 256;
    // channels per lane.  RP = 48 (C = 1152): ONE channel per lane (8 slices per crop) -- with two, the excite rows alone are 96
    // registers and the kernel needs 121: under the 3-forward load it then cannot share a SIMD with the four 104-register waves
    // of block 2's fused kernel (512 - 4 x 104 = 96 registers are free) and waits for a workgroup of theirs to retire: 11 us
    // per launch under load against 4.8 us alone, while the 88-register RP = 28 instantiation stays at 4.9 us (round 4).
    constexpr int NCI = (RP >= 48) ? 1 : 2;
    __shared__ float s_r[RP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / SPLIT, slice = blockIdx.x - b * SPLIT;
    const int per = (C + SPLIT - 1) / SPLIT;
    const int c_lo = slice * per, c_hi = (c_lo + per < C) ? c_lo + per : C;
    STAMP(0);

    // excite rows of this lane's channels (independent of the data)
    float wv[NCI][RP];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci) {
        const int c = c_lo + tid + ci * NTHR;
        const float4v* wrow = reinterpret_cast<const float4v*>(w2c + size_t(c < c_hi ? c : c_lo) * RP);
#pragma unroll
        for (int j = 0; j < RP; j += 4) {
            const float4v v = wrow[j >> 2];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[ci][j + q] = v[q];
        }
    }
    // r[j]: 4 running sums over the partial vectors p = u mod 4, combined (t0+t1)+(t2+t3)
    if (tid < RP)          // (se_device.h: the arithmetic shared with the project GEMMs' fused prologue)
        s_r[tid] = se_fused_r(rpart + size_t(b) * np * RP + tid, np, RP, inv_hw, tid < R ? b1[tid] : 0.f, tid < R);
    STAMP(1);
    lds_barrier();
    STAMP(2);
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci) {
        const int c = c_lo + tid + ci * NTHR;
        if (c < c_hi) {
            float t0 = b2[c], t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int j = 0; j < RP; j += 4) {
                t0 = fmaf(s_r[j], wv[ci][j], t0);
                t1 = fmaf(s_r[j + 1], wv[ci][j + 1], t1);
                t2 = fmaf(s_r[j + 2], wv[ci][j + 2], t2);
                t3 = fmaf(s_r[j + 3], wv[ci][j + 3], t3);
            }
            store_gate(gate, gate_f16, size_t(b) * C + c, sigmoid_f<true>((t0 + t1) + (t2 + t3)));
        }
    }
    STAMP(3);
}

template <int RP>
void launch_ex(const SeExciteArgs& a, hipStream_t stream) {
    const int split = se_excite_split(a.C);
    // a workgroup gates 256 channels per lane-slot: 1 slot for RP >= 48, 2 otherwise (NCI in the kernel); channels past
    // that would silently get no gate.  In B0 RP = 48 only occurs with C = 1152 (split 8 -> 144 channels per workgroup).
    WHENET_REQUIRE((a.C + split - 1) / split <= (RP >= 48 ? 1 : 2) * 256, WHENET_EINVAL,
                   "se excite: channels per workgroup exceed what the kernel covers");
    hipLaunchKernelGGL(whenet_se_excite_kernel<RP>, dim3(a.n * split), dim3(256), 0, stream, a.rpart, a.np, a.inv_hw,
                       a.b1, a.w2c, a.b2, a.gate, a.C, a.R, split, a.gate_f16);
}

template <int RP>
void launch_rp(const SeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(whenet_se_kernel<RP>, dim3(a.n), dim3(1024), 0, stream, a.partial, a.ntiles, a.inv_hw, a.w1t,
                       a.b1, a.w2c, a.b2, a.gate, a.C, a.R, a.gate_f16);
}

}  // namespace

int se_padded_r(int R) { return (R + 3) & ~3; }

// excite workgroups per crop: a 256-lane workgroup covers <= 512 channels (two per lane; ONE per lane for C > 768: see the
// kernel), and wide layers are split further so that no CU streams more than ~64 KB of excite kernel
int se_excite_split(int C) { return C > 768 ? 8 : (C > 288 ? 2 : 1); }

void launch_se_excite(const SeExciteArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.C <= 1152 && a.np >= 1, WHENET_EINVAL, "squeeze-excite: bad shape");
    switch (se_padded_r(a.R)) {
        case 4: launch_ex<4>(a, stream); break;
        case 8: launch_ex<8>(a, stream); break;
        case 12: launch_ex<12>(a, stream); break;
        case 20: launch_ex<20>(a, stream); break;
        case 28: launch_ex<28>(a, stream); break;
        case 48: launch_ex<48>(a, stream); break;
        default: throw Error(WHENET_EINVAL, "squeeze-excite: unsupported reduced width");
    }
    WHENET_HIP_CHECK(hipGetLastError());
}

void launch_se(const SeArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.C <= 1152, WHENET_EINVAL, "squeeze-excite: C > 1152");
    switch (se_padded_r(a.R)) {
        case 4: launch_rp<4>(a, stream); break;
        case 8: launch_rp<8>(a, stream); break;
        case 12: launch_rp<12>(a, stream); break;
        case 20: launch_rp<20>(a, stream); break;
        case 28: launch_rp<28>(a, stream); break;
        case 48: launch_rp<48>(a, stream); break;
        default: throw Error(WHENET_EINVAL, "squeeze-excite: unsupported reduced width");
    }
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
