// GlobalAveragePooling2D + Dense(yaw 120 | pitch 66 | roll 66) + softmax-expectation decode.
//
// Reference: /root/reference/whenet.py:10 (GAP over the 7x7x1280 head-conv output),
// whenet.py:11-13 (three linear Dense heads), utils.py:7-11 (softmax: subtract row max, exp,
// divide by the sum) and whenet.py:28-33 (expectation over bin indices, *3 - 180 / - 99).
// "Bin argmax" (north-star) = first index of the maximum logit, as np.argmax.
//
// One workgroup per crop, all f32: 0.1 % of the network's MACs; fused so the 1280 features and
// the 252 logits never leave the CU.  Fixed reduction orders -> bitwise reproducible.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

constexpr int HW = 49;

template <typename T>
__global__ __launch_bounds__(1024) void whenet_heads_kernel(const T* __restrict__ x, const float* __restrict__ feat_in,
                                                           const float* __restrict__ logits_in,
                                                           const float* __restrict__ w, const float* __restrict__ bvec,
                                                           float* __restrict__ feat_out, float* __restrict__ logits_out,
                                                           float* __restrict__ ypr, int32_t* __restrict__ amax) {
    __shared__ float s_feat[FEAT];
    constexpr int NTHR = 1024, NW = 16;            // a pure latency chain: spread thin
    __shared__ float s_part[NW][N_LOGITS + 4];
    __shared__ float s_gap[3][FEAT];
    __shared__ float s_logit[N_LOGITS + 4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;

    if (logits_in == nullptr) {
        // ---- GAP: mean over the 49 positions (whenet.py:10); lane <-> 4 channels ------------
        if (x != nullptr) {
            // lane <-> (4 channels, one third of the 49 positions): 16-17 independent loads in flight
            using V4 = T __attribute__((ext_vector_type(4)));
            const T* xb = x + size_t(b) * HW * FEAT;
            const int c4 = tid % (FEAT / 4), part = tid / (FEAT / 4);
            if (part < 3) {
                float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    const int p = part + 3 * i;
                    if (p < HW) {
                        const V4 v = *reinterpret_cast<const V4*>(xb + size_t(p) * FEAT + c4 * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] += float(v[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) s_gap[part][c4 * 4 + j] = t[j];
            }
            __syncthreads();
            for (int c = tid; c < FEAT; c += NTHR) s_feat[c] = ((s_gap[0][c] + s_gap[1][c]) + s_gap[2][c]) * (1.0f / 49.0f);
        } else {
            for (int c = tid; c < FEAT; c += NTHR) s_feat[c] = feat_in[size_t(b) * FEAT + c];
        }
        __syncthreads();
        if (feat_out != nullptr)
            for (int c = tid; c < FEAT; c += NTHR) feat_out[size_t(b) * FEAT + c] = s_feat[c];

        // ---- Dense: logits[j] = sum_c feat[c]*W[c][j] + b[j]  (whenet.py:11-13) ------------
        // 16 waves split the 1280-long contraction; lane l owns logits 4l..4l+3 (16-byte loads of
        // the [1280][252] kernel rows, 8 rows in flight).
        const int wave = tid >> 6, lane = tid & 63;
        const int c_lo = wave * (FEAT / NW);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane < N_LOGITS / 4) {
            const float* wr = w + size_t(c_lo) * N_LOGITS + lane * 4;
#pragma unroll 8
            for (int c = 0; c < FEAT / NW; ++c) {
                const float f = s_feat[c_lo + c];
                const float4v wv = *reinterpret_cast<const float4v*>(wr + size_t(c) * N_LOGITS);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(f, wv[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_part[wave][lane * 4 + i] = acc[i];
        }
        __syncthreads();
        if (tid < N_LOGITS) {
            float t = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) t += s_part[w2][tid];
            s_logit[tid] = t + bvec[tid];
        }
    } else {
        if (tid < N_LOGITS) s_logit[tid] = logits_in[size_t(b) * N_LOGITS + tid];
    }
    __syncthreads();
    if (logits_out != nullptr && tid < N_LOGITS) logits_out[size_t(b) * N_LOGITS + tid] = s_logit[tid];

    // ---- decode: wave h handles head h (utils.py:7-11, whenet.py:28-33) --------------------
    const int wave = tid >> 6, lane = tid & 63;
    if (wave >= 3) return;
    const int lo = (wave == 0) ? 0 : (wave == 1 ? N_YAW : N_YAW + N_PITCH);
    const int nb = (wave == 0) ? N_YAW : N_PITCH;
    // row max + first argmax
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
    // a = exp(x - max); b = sum(a); expectation = sum(a/b * idx)
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
        ypr[size_t(b) * 3 + wave] = ex * 3.0f - ((wave == 0) ? 180.0f : 99.0f);
        if (amax != nullptr) amax[size_t(b) * 3 + wave] = mi;
    }
}

// The same stage spread over HSPLIT workgroups per crop.  One CU pulls ~50 GB/s from L2 whatever it does, and the
// Dense kernel is 1.29 MB: a single workgroup per crop needs >= 22 us for it at ANY batch size.  Here workgroup q of a
// crop pools and contracts channels [320q, 320q + 320) only (322 KB of the kernel), writes its 252 partial logits
// write-through, and takes a ticket on the crop's counter; the workgroup that draws the last ticket adds the four
// partial vectors in fixed order (q = 0..3), the bias, and decodes.  The counter is reset by that workgroup (the next
// launch finds it zero); the order of the additions does not depend on which workgroup finishes last.
constexpr int HSPLIT = 4, HCH = FEAT / HSPLIT;          // 320 channels per workgroup

__device__ __forceinline__ float ld_l2_f32(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_l2_f32(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// FEAT_IN: x is the pooled feature vector [n][1280] f32 (head7.hip pooled it), not the head conv's output tensor
template <typename T, bool FEAT_IN>
__global__ __launch_bounds__(FEAT_IN ? 256 : 512) void whenet_heads_split_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bvec,
                                                                float* __restrict__ logits_out, float* __restrict__ ypr,
                                                                int32_t* __restrict__ amax, float* __restrict__ part,
                                                                unsigned* __restrict__ count) {
    // FEAT_IN: 4 waves (no pooling to spread): a 256-lane workgroup of 64 registers shares a SIMD with the four 104-register
    // waves of block 2's fused kernel under the 3-forward load; the 8-wave form needs two such waves per SIMD and waits
    constexpr int NTHR = FEAT_IN ? 256 : 512, NW = NTHR / 64, GP = 6;      // 6 position groups in the pooling
    // (FEAT_IN: no pooling scratch -- 9.5 instead of 17 KB of LDS, so that under the 3-forward load a workgroup fits beside the
    //  two 75 KB workgroups of block 2's fused kernel on a CU instead of waiting for one of them to retire)
    __shared__ float s_gap[FEAT_IN ? 1 : GP][FEAT_IN ? 4 : HCH];
    __shared__ float s_feat[HCH];
    __shared__ float s_part[NW][N_LOGITS + 4];
    __shared__ float s_logit[N_LOGITS + 4];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x;
    const int q = blockIdx.x, b = blockIdx.y;
    const int wave = tid >> 6, lane = tid & 63;

    // ---- GAP over the 49 positions for this workgroup's channels: lane <-> (4 channels, positions p = grp mod 6)
    if constexpr (FEAT_IN) {
        for (int c = tid; c < HCH; c += NTHR) s_feat[c] = reinterpret_cast<const float*>(x)[size_t(b) * FEAT + q * HCH + c];
    } else {
        using V4 = T __attribute__((ext_vector_type(4)));
        const T* xb = x + size_t(b) * HW * FEAT + q * HCH;
        const int c4 = tid % (HCH / 4), grp = tid / (HCH / 4);
        if (grp < GP) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int p = grp + GP * i;
                if (p < HW) {
                    const V4 v = *reinterpret_cast<const V4*>(xb + size_t(p) * FEAT + c4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] += float(v[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s_gap[grp][c4 * 4 + j] = t[j];
        }
    }
    __syncthreads();
    if constexpr (!FEAT_IN) {
        if (tid < HCH)
            s_feat[tid] = (((s_gap[0][tid] + s_gap[1][tid]) + (s_gap[2][tid] + s_gap[3][tid])) + (s_gap[4][tid] + s_gap[5][tid])) *
                          (1.0f / 49.0f);
        __syncthreads();
    }

    // ---- Dense partial: 8 waves x 40 channels, lane l owns logits 4l..4l+3 --------------------------------
    {
        const int c_lo = wave * (HCH / NW);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane < N_LOGITS / 4) {
            const float* wr = w + size_t(q * HCH + c_lo) * N_LOGITS + lane * 4;
#pragma unroll 10
            for (int c = 0; c < HCH / NW; ++c) {
                const float f = s_feat[c_lo + c];
                const float4v wv = *reinterpret_cast<const float4v*>(wr + size_t(c) * N_LOGITS);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(f, wv[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_part[wave][lane * 4 + i] = acc[i];
        }
    }
    __syncthreads();
    float* mine = part + (size_t(b) * HSPLIT + q) * N_LOGITS;
    if (tid < N_LOGITS) {
        float t = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) t += s_part[w2][tid];
        st_l2_f32(mine + tid, t);
    }
    // ---- ticket: the partial vector is at the device's coherence point before the counter moves.  The four
    // workgroups of a crop may sit on different XCDs (non-coherent L2s): the partial vectors are written and read with
    // agent-scope atomic accesses (sc1: write-through stores, loads that do not hit a stale local line), the stores
    // have been acknowledged (vmcnt) before the counter is touched, and the counter itself is an agent-scope RMW.
    // (Round 3 measured the textbook form -- release / acquire fences, i.e. buffer_wbl2 sc1 + buffer_inv sc1 in every
    // workgroup: 11 % of the whole forward's throughput at batch 64 for the same results; the in-flight corruption it
    // was tried against came from the counters' initialisation, engine.cpp ensure_capacity.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == HSPLIT - 1) __hip_atomic_store(count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ticket = t;
    }
    __syncthreads();
    if (s_ticket != HSPLIT - 1) return;

    // ---- last workgroup of the crop: logits = ((p0 + p1) + p2) + p3 + bias, then the decode -----------------
    if (tid < N_LOGITS) {
        const float* pb = part + size_t(b) * HSPLIT * N_LOGITS + tid;
        const float p0 = ld_l2_f32(pb), p1 = ld_l2_f32(pb + N_LOGITS), p2 = ld_l2_f32(pb + 2 * N_LOGITS),
                    p3 = ld_l2_f32(pb + 3 * N_LOGITS);
        const float t = (((p0 + p1) + p2) + p3) + bvec[tid];
        s_logit[tid] = t;
        if (logits_out != nullptr) logits_out[size_t(b) * N_LOGITS + tid] = t;
    }
    __syncthreads();
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
        ypr[size_t(b) * 3 + wave] = ex * 3.0f - ((wave == 0) ? 180.0f : 99.0f);
        if (amax != nullptr) amax[size_t(b) * 3 + wave] = mi;
    }
}

}  // namespace

int heads_split() { return HSPLIT; }

void launch_heads_split(const HeadsArgs& a, float* part, unsigned* count, int dtype, hipStream_t stream) {
    WHENET_REQUIRE((a.x != nullptr || a.feat_in != nullptr) && part != nullptr && count != nullptr, WHENET_EINVAL,
                   "heads (split): missing buffers");
    if (a.feat_in != nullptr)
        hipLaunchKernelGGL((whenet_heads_split_kernel<float, true>), dim3(HSPLIT, a.n), dim3(256), 0, stream, a.feat_in, a.w, a.b,
                           a.logits, a.ypr, a.argmax, part, count);
    else if (dtype == WHENET_F16)
        hipLaunchKernelGGL((whenet_heads_split_kernel<half_t, false>), dim3(HSPLIT, a.n), dim3(512), 0, stream,
                           static_cast<const half_t*>(a.x), a.w, a.b, a.logits, a.ypr, a.argmax, part, count);
    else
        hipLaunchKernelGGL((whenet_heads_split_kernel<float, false>), dim3(HSPLIT, a.n), dim3(512), 0, stream,
                           static_cast<const float*>(a.x), a.w, a.b, a.logits, a.ypr, a.argmax, part, count);
    WHENET_HIP_CHECK(hipGetLastError());
}

void launch_heads(const HeadsArgs& a, int dtype, hipStream_t stream) {
    if (dtype == WHENET_F16 && a.x != nullptr)
        hipLaunchKernelGGL(whenet_heads_kernel<half_t>, dim3(a.n), dim3(1024), 0, stream,
                           static_cast<const half_t*>(a.x), a.feat_in, a.logits_in, a.w, a.b, a.feat, a.logits, a.ypr,
                           a.argmax);
    else
        hipLaunchKernelGGL(whenet_heads_kernel<float>, dim3(a.n), dim3(1024), 0, stream,
                           static_cast<const float*>(a.x), a.feat_in, a.logits_in, a.w, a.b, a.feat, a.logits, a.ypr,
                           a.argmax);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
