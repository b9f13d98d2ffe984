// "Trunk" kernel: MBConv blocks 7..16 (14x14 and 7x7 maps) + head conv + GAP + Dense heads + decode as
// ONE persistent launch.  A crop is processed by a CLUSTER of C workgroups (one per CU) that split
// every layer's expanded channels between them and synchronise only with each other.
//
// Reference: the same stages as pw.hip / front.hip / se.hip / head.hip -- efficientnet 0.0.4
// MBConvBlock x10 and the head Conv1x1(1280)+BN+Swish (instantiated by /root/reference/whenet.py:8;
// SURVEY.md Appendix B), GlobalAveragePooling2D + Dense 120/66/66 (whenet.py:10-13), softmax-expectation
// decode (whenet.py:28-33, utils.py:7-11).
//
// Why: as separate launches these 32 layers cost 17-25 us EACH at 64 crops (573 of the
// 1035 us chain) for tensors of 20-260 KB per crop -- every launch is a cold latency chain on a mostly
// idle chip.  A single workgroup per crop (the removed tail kernel) is bound by ONE CU streaming all
// 6.6 MB of weights (~1.1 ms per crop).  Here:
//   * cluster member m owns a contiguous range of the expanded channels' 32-wide tiles in every block:
//     it streams 1/C of the expand, depthwise, squeeze-excite and project weights (plain loads, L2-hot:
//     every cluster reads the same weights) -- the weights are spread over all CUs;
//   * the block input X ([HW][Cin], <= 47 KB f16) is gathered into every member's LDS; the member runs
//     the expand conv of ITS channels on the matrix cores into an LDS tile E (zero halo = TF 'SAME'
//     padding of the expanded tensor), the depthwise taps out of E, BN+Swish, and keeps the result D
//     ([HWo][own channels]) for itself (own global scratch, L2-resident: LDS holds X and E meanwhile);
//   * squeeze-excite: the reduce conv is linear in the channel sums, so each member applies it to ITS
//     sums; the C partial vectors (R <= 48 floats each) are the only thing exchanged before the gate;
//   * project conv: split over K = the member's own channels: D * gate is staged once into LDS (B
//     operand), the member's partial [HWo][Cout] f32 goes to scratch, and after the second exchange every
//     member reduces 1/C of the output rows in FIXED member order (+ bias, + skip), writes them in the
//     activation type, and the next block gathers.
// Three cluster-local exchanges per block instead of three kernel boundaries; nothing is grid-wide.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"; placement-independent): every exchanged byte is written with write-through (sc1) stores
// and read with sc1 loads (L1 bypassed); a producer drains its stores (s_waitcnt vmcnt(0)) in every wave,
// the workgroup meets at a barrier, ONE lane adds 1 to the cluster's monotonic counter (relaxed, agent
// scope) and polls it (relaxed) until all C members of the current phase have arrived.  The counter is
// zeroed by a memset node ahead of every launch.  Polls are bounded: on a timeout the error word of the
// cluster is set and the kernel runs on (the host reports WHENET_EHIP).
// Residency: grid = min(n, CUs / C) clusters x C workgroups <= one workgroup per CU; clusters loop over
// crops.  Members of a cluster are adjacent block ids, so a cluster is complete as soon as it is dispatched.
//
// Summation orders are fixed by (layer, C) only: a crop's result is bitwise independent of the batch
// size, its position in the batch and the cluster that processes it.
#include "device_math.h"
#include "kernels.h"

#include <algorithm>

namespace whenet {

namespace {

constexpr int MAXW = 16;      // waves per workgroup of the widest instantiation (1024 lanes)
constexpr int P = 7;
constexpr int VC = 4;
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- write-through / L1-bypassing accesses of exchanged data -------------------------------------
struct Buf {                  // wave-uniform buffer descriptor + helpers (offsets in bytes)
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ Buf(const void* base, unsigned nbytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(nbytes), 0x00020000)) {}
    __device__ __forceinline__ u32x4 ld16(unsigned off) const { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
    __device__ __forceinline__ void st16(unsigned off, u32x4 v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
};
__device__ __forceinline__ float ld_sc1_f32(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_f32(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1_u64(const void* p) {
    return __hip_atomic_load(static_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u64(void* p, unsigned long long v) {
    __hip_atomic_store(static_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> __device__ __forceinline__ typename Vec<T>::type as_vec(u32x4 v) {
    typename Vec<T>::type o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}
template <typename T> __device__ __forceinline__ u32x4 as_u32(typename Vec<T>::type v) {
    u32x4 o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}

// Cluster-local barrier in two halves, so that work that does not depend on the other members (weight
// prefetch, LDS set-up) fills the wait.  `counter` is the cluster's monotonic arrival counter;
// `target` = C * (number of barriers so far, this one included).
//   arrive: publication of this workgroup's sc1 stores -- every wave drains its stores, the workgroup
//           meets, ONE lane adds 1 (relaxed, agent scope);
//   wait:   ONE lane polls (relaxed) until all C members have arrived, then the workgroup meets again.
__device__ __forceinline__ void cluster_arrive(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores have reached memory
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cluster_wait(unsigned* counter, unsigned target, unsigned* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// L2 warm-up of weights that a LATER phase reads: one dword per 128-byte line, destination parked in a
// register until touch_retire().  Every weight of this kernel is read once per crop, so without this each
// phase's first use pays a full memory round trip (~1 us measured); a touch a block ahead turns those into
// L2 hits.
constexpr int NTOUCH = 6;
struct Touch {
    unsigned r[NTOUCH];
};
struct TouchRegion {          // rows x row_bytes, row pitch pitch_bytes
    const char* base;
    int lpr, lines, pitch;
    __device__ __forceinline__ TouchRegion(const void* b, int rows, int row_bytes, int pitch_bytes)
        : base(static_cast<const char*>(b)), lpr((row_bytes + 127) >> 7), lines(rows * ((row_bytes + 127) >> 7)),
          pitch(pitch_bytes) {}
    __device__ __forceinline__ const char* line(int i) const {
        const int r = i / lpr, c = i - r * lpr;
        return base + size_t(r) * pitch + c * 128;
    }
};
// (a compiler-visible load: hipcc tracks its destination register and waits for it only at touch_retire's
// use; the empty asm with a memory clobber keeps the load from being sunk down to that use)
__device__ __forceinline__ void touch_line(unsigned& dst, const void* p) {
    dst = *static_cast<const unsigned*>(p);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void touch_retire(Touch& t) {
#pragma unroll
    for (int u = 0; u < NTOUCH; ++u) asm volatile("" ::"v"(t.r[u]));
}

// contiguous, balanced split of `total` items over `parts`; part `i` gets [lo, lo + cnt)
__host__ __device__ inline void split_range(int total, int parts, int i, int* lo, int* cnt) {
    const int base = total / parts, extra = total % parts;
    *lo = i * base + (i < extra ? i : extra);
    *cnt = base + (i < extra ? 1 : 0);
}

__host__ __device__ constexpr int align16(int x) { return (x + 15) & ~15; }

// ---- one (strip, tile) GEMM task: acc += sum_k W[tile][k] * act[row][k] ---------------------------
// U weight fragments (global, 1 KiB per wave each) are in flight before the first MFMA of a group; the
// activation fragments come from LDS.
template <typename T, int U, typename LoadW, typename LoadA>
__device__ __forceinline__ void gemm_task(float16v& acc, int ks0, int ks1, LoadW&& load_w, LoadA&& load_a) {
    using VT = typename Vec<T>::type;
    for (int ks = ks0; ks < ks1; ks += U) {
        VT w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = (ks + u < ks1) ? load_w(ks + u) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ks + u < ks1) {                       // (wave-uniform)
                const VT av = load_a(ks + u);
                Mfma<T>::step(w[u], av, acc);
            }
        }
    }
}

// ---- depthwise taps of one sub-chunk: E (LDS) -> D (own global scratch), strip channel sums -> s_red
template <typename T, int K, int S, int TNT>
__device__ __forceinline__ void dw_chunk(const unsigned char* __restrict__ E, int EW, int EP, T* __restrict__ D,
                                         int dpitch, int dcol0, const float* __restrict__ s_dww,
                                         const float* __restrict__ s_bd, float* __restrict__ s_red, int Ho, int ccur,
                                         int tid) {
    using VCT = T __attribute__((ext_vector_type(VC)));
    constexpr int NIX = (P - 1) * S + K;
    const int CG = ccur / VC;
    const int spr = Ho / P;                          // strips per output row
    const int nstrip = Ho * spr;
    for (int lt = tid; lt < CG * nstrip; lt += TNT) {
        const int cg = lt % CG;
        const int sidx = lt / CG;
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
        const float4v bs = *reinterpret_cast<const float4v*>(s_bd + cg * VC);
        float sum[VC] = {0.f, 0.f, 0.f, 0.f};
        T* dst = D + (size_t(oy) * Ho + sx * P) * dpitch + dcol0 + cg * VC;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            VCT o;
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const float y = swish_f<IsF32<T>::value>(acc[p][v] + bs[v]);
                sum[v] += y;
                o[v] = T(y);
            }
            *reinterpret_cast<VCT*>(dst + size_t(p) * dpitch) = o;
        }
#pragma unroll
        for (int v = 0; v < VC; ++v) s_red[sidx * ccur + cg * VC + v] = sum[v];
    }
}

// TNT lanes per workgroup: 1024 (4 waves per SIMD, 128 registers per lane) or 512 (2 waves per SIMD, 256
// registers: nothing spills, deeper prefetch per wave)
template <typename T, int TNT>
__global__ __launch_bounds__(TNT) void whenet_trunk_kernel(TrunkArgs a) {
    constexpr int TNW = TNT / 64;
    constexpr int V = Vec<T>::V;
    constexpr int SZ = int(sizeof(T));
    constexpr int KPT = 32 / (2 * V);                 // k-steps per 32-channel tile (f16: 2, f32: 4)
    using VT = typename Vec<T>::type;
    using OT = T __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, lm = lane & 31;
    const int C = a.C;
    const int cluster = blockIdx.x / C;
    const int m = blockIdx.x - cluster * C;           // member index inside the cluster

    // ---- this cluster's scratch (see TrunkArgs) ---------------------------------------------------
    unsigned char* sc = a.scratch + size_t(cluster) * a.scratch_stride;
    T* XB[2] = {reinterpret_cast<T*>(sc), reinterpret_cast<T*>(sc) + a.xmax};
    T* DB = reinterpret_cast<T*>(sc + a.off_d) + size_t(m) * a.dmax;                   // own depthwise output
    float* PBase = reinterpret_cast<float*>(sc + a.off_p);                               // [C][pmax] project partials
    float* RBase = reinterpret_cast<float*>(sc + a.off_r);                               // [C][64]  SE reduce partials
    float* LBase = reinterpret_cast<float*>(sc + a.off_l);                               // [C][256] logits partials
    unsigned* counter = a.counters + size_t(cluster) * 16;
    unsigned* err = counter + 1;
    unsigned phase = 0;                                // C * (cluster barriers ARRIVED at so far)

    // small per-workgroup arrays at the top of the LDS allocation (all block layouts stay below)
    float* s_sum = reinterpret_cast<float*>(smem + a.fixed_off);      // [1152] own channel sums
    float* s_gate = s_sum + 1152;                                      // [1152] own gate
    float* s_r = s_gate + 1152;                                        // [64]
    float* s_be = s_r + 64;                                            // [1152] expand bias of the own channels
    float* s_bd = s_be + 1152;                                         // [1152] depthwise bias of the own channels

    auto stamp = [&](int slot) {
        if (a.timing != nullptr && blockIdx.x == 0 && tid == 0) a.timing[slot] = wall_clock64();
    };

    // set-up of an expand/depthwise sub-chunk that does not depend on the block input: zero E (the halo is
    // TF 'SAME' padding of the EXPANDED tensor), park the chunk's depthwise taps in LDS
    auto setup_chunk = [&](const TrunkBlock& B, int c0, int ccur) {
        const int EW = (B.h_out - 1) * B.s + B.k;
        const int EP = ccur * SZ + 16;
        unsigned char* E = smem + B.off_e;
        float* s_dww = reinterpret_cast<float*>(smem + B.off_dww);
        const int ntap = B.k * B.k * ccur;
        for (int i0 = tid; i0 < ntap; i0 += 4 * TNT) {
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * TNT;
                const int tap = i / ccur, c = i - tap * ccur;
                wv[u] = (i < ntap) ? B.wd[size_t(tap) * B.cexp + c0 + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * TNT < ntap) s_dww[i0 + u * TNT] = wv[u];
        }
        for (int i = tid; i < EW * EW * EP / 16; i += TNT) reinterpret_cast<VT*>(E)[i] = vec_zero<T>();
    };

    // L2 warm-up of everything block `bi` (or, bi == nblk, the head) will read of the weights: lines
    // tid, tid + TNT, .. of the concatenation of the member's weight regions
    Touch touch;
#pragma unroll
    for (int u = 0; u < NTOUCH; ++u) touch.r[u] = 0;
    auto touch_block = [&](int bi) {
        if (bi > a.nblk) return;
        if (bi == a.nblk) {
            if (a.dump_x != nullptr) return;
            int ht0, htcnt;
            split_range(a.nth, C, m, &ht0, &htcnt);
            const TouchRegion rg[3] = {
                TouchRegion(static_cast<const char*>(a.wh) + size_t(ht0) * 1024, a.ksh, htcnt * 1024, a.nth * 1024),
                TouchRegion(a.wdense + size_t(ht0) * 32 * N_LOGITS, 1, htcnt * 32 * N_LOGITS * 4, 0),
                TouchRegion(a.bh + ht0 * 32, 1, htcnt * 32 * 4, 0)};
#pragma unroll
            for (int u = 0; u < NTOUCH; ++u) {
                int v = tid + u * TNT;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (v >= 0 && v < rg[q].lines) touch_line(touch.r[u], rg[q].line(v));
                    v -= rg[q].lines;
                }
            }
            return;
        }
        const TrunkBlock N = a.blk[bi];
        int t0, tcnt;
        split_range(N.cexp >> 5, C, m, &t0, &tcnt);
        const int own = tcnt * 32, c0 = t0 * 32;
        const TouchRegion rg[8] = {
            TouchRegion(static_cast<const char*>(N.we) + size_t(t0) * 1024, N.kse, tcnt * 1024, N.nte * 1024),
            TouchRegion(N.wd + c0, N.k * N.k, own * 4, N.cexp * 4),
            TouchRegion(N.w1t + c0, N.r, own * 4, N.cexp * 4),
            TouchRegion(N.w2c + size_t(c0) * N.rp, 1, own * N.rp * 4, 0),
            TouchRegion(static_cast<const char*>(N.wp) + size_t(t0) * KPT * N.ntp * 1024, 1, tcnt * KPT * N.ntp * 1024, 0),
            TouchRegion(N.b2 + c0, 1, own * 4, 0),
            TouchRegion(N.bp, 1, N.cout * 4, 0),
            TouchRegion(N.b1, 1, N.r * 4, 0)};
#pragma unroll
        for (int u = 0; u < NTOUCH; ++u) {
            int v = tid + u * TNT;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (v >= 0 && v < rg[q].lines) touch_line(touch.r[u], rg[q].line(v));
                v -= rg[q].lines;
            }
        }
    };

    bool pending_wait = false;                         // the last cluster barrier was arrived at but not yet waited for
    for (int crop = cluster; crop < a.n; crop += a.nclusters) {
        stamp(0);
        int cur = 0;                                   // XB[cur] = input of the current block (bi > 0)
        // crop -> its row in the block-6 output: lanes of the layer-wise front half are contiguous per lane
        size_t xoff = 0;
        {
            int ls = 0;
#pragma unroll
            for (int l = 1; l < 8; ++l)
                if (l < a.nlanes && crop >= a.lane_start[l]) ls = a.lane_start[l];
            xoff = size_t(ls) * a.lane_stride + size_t(crop - ls) * a.x_in_stride;
        }
        touch_block(0);
        touch_retire(touch);
        for (int bi = 0; bi < a.nblk; ++bi) {
            const TrunkBlock B = a.blk[bi];            // uniform: scalar loads from the device table
            const bool detail = a.timing != nullptr && bi == a.timing_block;
            auto dstamp = [&](int i) {
                if (detail && blockIdx.x == 0 && tid == 0) a.timing[128 + i] = wall_clock64();
            };
            const int HWi = B.h_in * B.h_in, HWo = B.h_out * B.h_out;
            const int pin = B.cin * SZ + 16;
            const int EW = (B.h_out - 1) * B.s + B.k;
            const T* xsrc = (bi == 0) ? static_cast<const T*>(a.x_in) + xoff : XB[cur];
            int t0, tcnt;                              // own 32-channel tiles of the expanded tensor
            split_range(B.cexp >> 5, C, m, &t0, &tcnt);
            const int own_ch = tcnt * 32, c_own0 = t0 * 32;
            unsigned char* X = smem;
            unsigned char* E = smem + B.off_e;
            float* s_dww = reinterpret_cast<float*>(smem + B.off_dww);
            float* s_red = reinterpret_cast<float*>(smem + B.off_red);
            const int nstrip_i = (HWi + 31) >> 5, nstrip_o = (HWo + 31) >> 5;
            dstamp(0);
            touch_block(bi + 1);                       // the NEXT block's weights start their way into L2 now

            // ---- independent of the block input: biases of the own channels, sub-chunk 0 set-up (these
            // loads and LDS writes fill the wait for the previous block's output) ----------------------
            for (int c = tid; c < own_ch; c += TNT) {
                s_be[c] = B.be[c_own0 + c];
                s_bd[c] = B.bd[c_own0 + c];
            }
            setup_chunk(B, c_own0, ((tcnt < B.sub_tiles) ? tcnt : B.sub_tiles) * 32);
            dstamp(1);
            if (pending_wait) {
                cluster_wait(counter, phase, err);     // ---- exchange 3 of the previous block: its output
                pending_wait = false;
            }
            dstamp(2);

            // ================= gather X (block input, all channels) into LDS ====================
            {
                const int vpr = B.cin * SZ / 16;
                const int total = HWi * vpr;
                const Buf xb(xsrc, unsigned(HWi * B.cin * SZ));
                for (int i0 = tid; i0 < total; i0 += 4 * TNT) {
                    u32x4 xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * TNT;
                        xv[u] = xb.ld16(unsigned(i < total ? i : 0) * 16u);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * TNT;
                        if (i < total) {
                            const int r = i / vpr, v = i - r * vpr;
                            *reinterpret_cast<u32x4*>(X + size_t(r) * pin + v * 16) = xv[u];
                        }
                    }
                }
            }
            __syncthreads();
            stamp(1 + bi * 8 + 0);
            dstamp(3);

            // ================= phase 1: expand (MFMA) -> E, depthwise -> D, channel sums ==========
            for (int ts = 0; ts < tcnt; ts += B.sub_tiles) {
                const int ntile = (tcnt - ts < B.sub_tiles) ? (tcnt - ts) : B.sub_tiles;
                const int ccur = ntile * 32;
                const int c0 = c_own0 + ts * 32;       // first expanded channel of this sub-chunk
                const int EP = ccur * SZ + 16;
                if (ts > 0) {
                    setup_chunk(B, c0, ccur);
                    __syncthreads();
                }
                for (int t = wave; t < nstrip_i * ntile; t += TNW) {
                    const int tile = t / nstrip_i, strip = t - tile * nstrip_i;
                    const int p = strip * 32 + lm;
                    const bool valid = p < HWi;
                    const unsigned char* xrow = X + size_t(valid ? p : 0) * pin + g * V * SZ;
                    float16v acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    const VT* wf = reinterpret_cast<const VT*>(B.we) + size_t(t0 + ts + tile) * 64 + lane;
                    const int wstride = B.nte * 64;
                    gemm_task<T, 8>(
                        acc, 0, B.kse, [&](int ks) -> VT { return wf[size_t(ks) * wstride]; },
                        [&](int ks) -> VT {
                            return valid ? *reinterpret_cast<const VT*>(xrow + size_t(ks) * 2 * V * SZ) : vec_zero<T>();
                        });
                    if (valid) {
                        const int py = p / B.h_in, px = p - py * B.h_in;
                        unsigned char* epix = E + size_t((py + B.pad) * EW + px + B.pad) * EP;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int nl = tile * 32 + 8 * qq + 4 * g;
                            const float4v bv = *reinterpret_cast<const float4v*>(s_be + ts * 32 + nl);
                            OT o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = T(swish_f<IsF32<T>::value>(acc[4 * qq + r] + bv[r]));
                            *reinterpret_cast<OT*>(epix + nl * SZ) = o;
                        }
                    }
                }
                __syncthreads();
                if (ts == 0) stamp(1 + bi * 8 + 1);
                dstamp(ts == 0 ? 4 : 8);
                const int dcol0 = ts * 32;
                if (B.k == 3) dw_chunk<T, 3, 1, TNT>(E, EW, EP, DB, own_ch, dcol0, s_dww, s_bd + dcol0, s_red, B.h_out, ccur, tid);
                else if (B.s == 1) dw_chunk<T, 5, 1, TNT>(E, EW, EP, DB, own_ch, dcol0, s_dww, s_bd + dcol0, s_red, B.h_out, ccur, tid);
                else dw_chunk<T, 5, 2, TNT>(E, EW, EP, DB, own_ch, dcol0, s_dww, s_bd + dcol0, s_red, B.h_out, ccur, tid);
                __syncthreads();
                if (ts == 0) stamp(1 + bi * 8 + 2);
                dstamp(ts == 0 ? 5 : 9);
                if (tid < ccur) {
                    const int nstrip = B.h_out * (B.h_out / P);
                    float t = 0.0f;
                    for (int s = 0; s < nstrip; ++s) t += s_red[s * ccur + tid];
                    s_sum[dcol0 + tid] = t;
                }
                __syncthreads();                       // s_red / E / s_dww are rewritten by the next sub-chunk
                dstamp(ts == 0 ? 6 : 10);
            }
            stamp(1 + bi * 8 + 3);

            // ================= squeeze-excite, first half: own share of the reduce conv =========
            // wave w owns outputs j = w, w + TNW, ..; all its loads of the reduce kernel are issued first
            {
                float* rb = RBase + size_t(m) * 64;
                constexpr int JR = 64 / TNW;           // R <= 64
                float p4[JR][4];
#pragma unroll
                for (int jj = 0; jj < JR; ++jj)
#pragma unroll
                    for (int u = 0; u < 4; ++u) p4[jj][u] = 0.f;
                for (int cb = lane; cb < own_ch; cb += 256) {
                    float wv[JR][4];
#pragma unroll
                    for (int jj = 0; jj < JR; ++jj) {
                        const int j = wave + TNW * jj;
                        const float* wrow = B.w1t + size_t(j < B.r ? j : 0) * B.cexp + c_own0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = cb + 64 * u;
                            wv[jj][u] = (j < B.r && c < own_ch) ? wrow[c] : 0.f;
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < JR; ++jj)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = cb + 64 * u;
                            if (c < own_ch) p4[jj][u] = fmaf(s_sum[c], wv[jj][u], p4[jj][u]);
                        }
                }
#pragma unroll
                for (int jj = 0; jj < JR; ++jj) {
                    const int j = wave + TNW * jj;
                    float t = (p4[jj][0] + p4[jj][1]) + (p4[jj][2] + p4[jj][3]);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                    if (lane == 0 && j < B.r) st_sc1_f32(rb + j, t);
                }
            }
            dstamp(11);
            touch_retire(touch);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 1: SE partial vectors (+ own D stores drained)
            dstamp(12);

            // ---- while the other members arrive: everything of the second half that does not depend on them
            unsigned char* Dg = smem;                  // X and E are dead from here on
            const int pd = own_ch * SZ + 16;
            const int ks0 = t0 * KPT, ks1 = (t0 + tcnt) * KPT;
            VT* Wl = reinterpret_cast<VT*>(smem + align16(HWo * pd));       // project weights of the own k-steps
            if (B.wp_lds) {
                const VT* wsrc = reinterpret_cast<const VT*>(B.wp) + size_t(ks0) * B.ntp * 64;
                const int total = (ks1 - ks0) * B.ntp * 64;
                for (int i0 = tid; i0 < total; i0 += 4 * TNT) {
                    VT wv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wv[u] = (i0 + u * TNT < total) ? wsrc[i0 + u * TNT] : vec_zero<T>();
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * TNT < total) Wl[i0 + u * TNT] = wv[u];
                }
            }
            dstamp(13);
            // own D back from scratch (first DR vectors per lane stay in flight across the wait) and this
            // lane's excite row
            constexpr int DR = 6;
            const int vprd = own_ch * SZ / 16;
            const int dtotal = HWo * vprd;
            const Buf db(DB, unsigned(HWo * own_ch * SZ));
            u32x4 dreg[DR];
#pragma unroll
            for (int u = 0; u < DR; ++u) {
                const int i = tid + u * TNT;
                dreg[u] = db.ld16(unsigned(i < dtotal ? i : 0) * 16u);
            }
            constexpr int RPV = 12;                    // RP <= 48
            float4v w2r[RPV];
            float b2r = 0.f;
            {
                const int c = (tid < own_ch) ? tid : 0;
                const float4v* wrow = reinterpret_cast<const float4v*>(B.w2c + size_t(c_own0 + c) * B.rp);
#pragma unroll
                for (int j = 0; j < RPV; ++j) w2r[j] = (4 * j < B.rp) ? wrow[j] : float4v{0.f, 0.f, 0.f, 0.f};
                b2r = B.b2[c_own0 + c];
            }
            const float b1r = (tid < B.r) ? B.b1[tid] : 0.f;
            dstamp(14);
            cluster_wait(counter, phase, err);
            stamp(1 + bi * 8 + 4);
            dstamp(15);

            // ================= squeeze-excite, second half: gate of the own channels =============
            if (tid < 64) {
                float t = 0.0f;
                if (tid < B.r) {
                    float pv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) pv[j] = (j < C) ? ld_sc1_f32(RBase + size_t(j) * 64 + tid) : 0.f;
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < C) t += pv[j];         // fixed member order
                    t = swish_f<true>(t * (1.0f / float(HWo)) + b1r);
                }
                s_r[tid] = t;                          // zero beyond R (the excite rows are zero-padded to RP)
            }
            __syncthreads();
            dstamp(16);
            for (int c = tid; c < own_ch; c += TNT) {
                float t0a, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                if (c == tid) {
                    t0a = b2r;
#pragma unroll
                    for (int j = 0; j < RPV; ++j) {
                        if (4 * j < B.rp) {
                            t0a = fmaf(s_r[4 * j], w2r[j][0], t0a);
                            t1 = fmaf(s_r[4 * j + 1], w2r[j][1], t1);
                            t2 = fmaf(s_r[4 * j + 2], w2r[j][2], t2);
                            t3 = fmaf(s_r[4 * j + 3], w2r[j][3], t3);
                        }
                    }
                } else {                               // (own_ch > 1024: small clusters only)
                    const float4v* wrow = reinterpret_cast<const float4v*>(B.w2c + size_t(c_own0 + c) * B.rp);
                    t0a = B.b2[c_own0 + c];
                    for (int j = 0; j < B.rp; j += 4) {
                        const float4v wv = wrow[j >> 2];
                        t0a = fmaf(s_r[j], wv[0], t0a);
                        t1 = fmaf(s_r[j + 1], wv[1], t1);
                        t2 = fmaf(s_r[j + 2], wv[2], t2);
                        t3 = fmaf(s_r[j + 3], wv[3], t3);
                    }
                }
                s_gate[c] = sigmoid_f<true>((t0a + t1) + (t2 + t3));
            }
            __syncthreads();
            dstamp(17);

            // ================= D * gate into LDS (B operand of the project GEMM) ==================
            {
                auto put = [&](int i, u32x4 raw) {
                    const int r = i / vprd, v = i - r * vprd;
                    float f[V];
                    vec_to_float<T>(as_vec<T>(raw), f);
#pragma unroll
                    for (int e = 0; e < V; ++e) f[e] *= s_gate[v * V + e];
                    *reinterpret_cast<VT*>(Dg + size_t(r) * pd + v * 16) = float_to_vec<T>(f);
                };
#pragma unroll
                for (int u = 0; u < DR; ++u)
                    if (tid + u * TNT < dtotal) put(tid + u * TNT, dreg[u]);
                for (int i0 = tid + DR * TNT; i0 < dtotal; i0 += 4 * TNT) {        // (small clusters only)
                    u32x4 dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) dv[u] = db.ld16(unsigned(i0 + u * TNT < dtotal ? i0 + u * TNT : 0) * 16u);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * TNT < dtotal) put(i0 + u * TNT, dv[u]);
                }
            }
            __syncthreads();
            stamp(1 + bi * 8 + 5);
            dstamp(18);

            // ================= project (MFMA), K = own channels -> partial [HWo][Cout] f32 ========
            {
                float* pb = PBase + size_t(m) * a.pmax;
                const Buf pbuf(pb, unsigned(HWo * B.cout * 4));
                for (int t = wave; t < nstrip_o * B.ntp; t += TNW) {
                    const int tile = t / nstrip_o, strip = t - tile * nstrip_o;
                    const int p = strip * 32 + lm;
                    const bool valid = p < HWo;
                    const unsigned char* drow = Dg + size_t(valid ? p : 0) * pd + g * V * SZ;
                    float16v acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    auto load_d = [&](int ks) -> VT {
                        return valid ? *reinterpret_cast<const VT*>(drow + size_t(ks - ks0) * 2 * V * SZ) : vec_zero<T>();
                    };
                    if (B.wp_lds) {
                        const VT* wl = Wl + size_t(tile) * 64 + lane;
                        const int wstride = B.ntp * 64;
                        gemm_task<T, 4>(acc, ks0, ks1, [&](int ks) -> VT { return wl[size_t(ks - ks0) * wstride]; }, load_d);
                    } else {
                        const VT* wf = reinterpret_cast<const VT*>(B.wp) + size_t(tile) * 64 + lane;
                        const int wstride = B.ntp * 64;
                        gemm_task<T, 8>(acc, ks0, ks1, [&](int ks) -> VT { return wf[size_t(ks) * wstride]; }, load_d);
                    }
                    if (valid) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int n = tile * 32 + 8 * qq + 4 * g;
                            if (n < B.cout) {
                                float4v y;
#pragma unroll
                                for (int r = 0; r < 4; ++r) y[r] = acc[4 * qq + r];
                                u32x4 raw;
                                __builtin_memcpy(&raw, &y, 16);
                                pbuf.st16(unsigned(p * B.cout + n) * 4u, raw);
                            }
                        }
                    }
                }
            }
            dstamp(19);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 2: project partials
            dstamp(20);

            // ================= reduce own rows: bias + partials in member order (+ skip) -> x_out ==
            {
                const int nxt = (bi == 0) ? 0 : (cur ^ 1);
                T* xo = XB[nxt];
                const int upr = B.cout / 4;            // float4 units per row
                int u0, ucnt;
                split_range(HWo * upr, C, m, &u0, &ucnt);
                const Buf pall(PBase, unsigned(size_t(C) * a.pmax * 4));
                const Buf xs(xsrc, unsigned(HWi * B.cin * SZ));
                const Buf xb(xo, unsigned(HWo * B.cout * SZ));
                // bias and skip of this lane's first unit do not depend on the partials: loaded before the wait
                auto load_skip = [&](int u) -> OT {
                    OT rv;
                    if constexpr (SZ == 2) {
                        const unsigned long long raw = ld_sc1_u64(xsrc + size_t(u) * 4);
                        __builtin_memcpy(&rv, &raw, 8);
                    } else {
                        const u32x4 raw = xs.ld16(unsigned(u) * 16u);
                        __builtin_memcpy(&rv, &raw, 16);
                    }
                    return rv;
                };
                OT skip0;
#pragma unroll
                for (int r = 0; r < 4; ++r) skip0[r] = T(0);
                float4v bias0 = {0.f, 0.f, 0.f, 0.f};
                if (tid < ucnt) {
                    bias0 = *reinterpret_cast<const float4v*>(B.bp + ((u0 + tid) % upr) * 4);
                    if (B.has_skip) skip0 = load_skip(u0 + tid);
                }
                cluster_wait(counter, phase, err);
                stamp(1 + bi * 8 + 6);
                dstamp(21);
                for (int i = tid; i < ucnt; i += TNT) {
                    const int u = u0 + i;
                    float4v bv = bias0;
                    OT rv = skip0;
                    if (i != tid) {
                        bv = *reinterpret_cast<const float4v*>(B.bp + (u % upr) * 4);
                        if (B.has_skip) rv = load_skip(u);
                    }
                    float y[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int j0 = 0; j0 < C; j0 += 4) {            // fixed member order, 4 loads in flight
                        u32x4 raw[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            raw[j] = pall.ld16(unsigned((size_t(j0 + j < C ? j0 + j : j0) * a.pmax + size_t(u) * 4) * 4));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (j0 + j < C) {
                                float4v pv;
                                __builtin_memcpy(&pv, &raw[j], 16);
#pragma unroll
                                for (int r = 0; r < 4; ++r) y[r] += pv[r];
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] += bv[r];
                    if (B.has_skip) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] += float(rv[r]);
                    }
                    OT o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = T(y[r]);
                    if constexpr (SZ == 2) {
                        unsigned long long w;
                        __builtin_memcpy(&w, &o, 8);
                        st_sc1_u64(xo + size_t(u) * 4, w);
                    } else {
                        u32x4 w;
                        __builtin_memcpy(&w, &o, 16);
                        xb.st16(unsigned(u) * 16u, w);
                    }
                }
                cur = nxt;
            }
            dstamp(22);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 3: block output (waited for by its consumer)
            pending_wait = true;
            stamp(1 + bi * 8 + 7);
            dstamp(23);
        }

        const TrunkBlock L = a.blk[a.nblk - 1];
        const int HWl = L.h_out * L.h_out, Cl = L.cout;
        cluster_wait(counter, phase, err);
        pending_wait = false;
        if (a.dump_x != nullptr) {                     // test hook: the block chain's output, [HW][C] as f32
            if (m == 0) {
                float* dst = a.dump_x + size_t(crop) * HWl * Cl;
                const T* src = XB[cur];
                const Buf xs(src, unsigned(HWl * Cl * SZ));
                for (int i = tid; i < HWl * Cl / 4; i += TNT) {
                    OT rv;
                    if constexpr (SZ == 2) {
                        const unsigned long long raw = ld_sc1_u64(src + size_t(i) * 4);
                        __builtin_memcpy(&rv, &raw, 8);
                    } else {
                        const u32x4 raw = xs.ld16(unsigned(i) * 16u);
                        __builtin_memcpy(&rv, &raw, 16);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[size_t(i) * 4 + r] = float(rv[r]);
                }
            }
            // the next crop's blocks reuse XB: nobody may overwrite it before member 0 has read it
            phase += unsigned(C);
            cluster_arrive(counter);
            cluster_wait(counter, phase, err);
            continue;
        }

        // ================= head conv (MFMA), own out-channel tiles + BN + Swish, fused GAP ========
        const int pl = Cl * SZ + 16;
        unsigned char* X = smem;
        float* s_fp = reinterpret_cast<float*>(smem + align16(HWl * pl));      // [2][own_n] strip partial sums
        int ht0, htcnt;
        split_range(a.nth, C, m, &ht0, &htcnt);
        const int own_n = htcnt * 32;
        float* s_feat = s_fp + 2 * own_n;                                      // [own_n]
        float* s_part = s_feat + own_n;                                        // [TNW][256]
        {
            const int vpr = Cl * SZ / 16;
            const int total = HWl * vpr;
            const Buf xb(XB[cur], unsigned(HWl * Cl * SZ));
            for (int i0 = tid; i0 < total; i0 += 4 * TNT) {
                u32x4 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = xb.ld16(unsigned(i0 + u * TNT < total ? i0 + u * TNT : 0) * 16u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * TNT;
                    if (i < total) {
                        const int r = i / vpr, v = i - r * vpr;
                        *reinterpret_cast<u32x4*>(X + size_t(r) * pl + v * 16) = xv[u];
                    }
                }
            }
        }
        __syncthreads();
        stamp(88);
        {
            const int nstrip = (HWl + 31) >> 5;                                // 2
            for (int t = wave; t < nstrip * htcnt; t += TNW) {
                const int tile = t / nstrip, strip = t - tile * nstrip;
                const int p = strip * 32 + lm;
                const bool valid = p < HWl;
                const unsigned char* xrow = X + size_t(valid ? p : 0) * pl + g * V * SZ;
                float16v acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const VT* wf = reinterpret_cast<const VT*>(a.wh) + size_t(ht0 + tile) * 64 + lane;
                const int wstride = a.nth * 64;
                float4v bv[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    bv[qq] = *reinterpret_cast<const float4v*>(a.bh + (ht0 * 32 + tile * 32 + 8 * qq + 4 * g));
                gemm_task<T, 8>(
                    acc, 0, a.ksh, [&](int ks) -> VT { return wf[size_t(ks) * wstride]; },
                    [&](int ks) -> VT {
                        return valid ? *reinterpret_cast<const VT*>(xrow + size_t(ks) * 2 * V * SZ) : vec_zero<T>();
                    });
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int nl = tile * 32 + 8 * qq + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = valid ? swish_f<IsF32<T>::value>(acc[4 * qq + r] + bv[qq][r]) : 0.0f;
#pragma unroll
                        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                        if (lm == 0) s_fp[strip * own_n + nl + r] = v;
                    }
                }
            }
            __syncthreads();
            for (int c = tid; c < own_n; c += TNT) {
                const float f = (s_fp[c] + s_fp[own_n + c]) * (1.0f / 49.0f);
                s_feat[c] = f;
                if (a.feat != nullptr) a.feat[size_t(crop) * FEAT + ht0 * 32 + c] = f;
            }
            __syncthreads();
        }
        stamp(89);
        // ================= Dense 120|66|66 (whenet.py:11-13): partial over the own features ========
        {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (lane < N_LOGITS / 4) {
                const float* wr = a.wdense + size_t(ht0 * 32) * N_LOGITS + lane * 4;
                for (int c0 = wave; c0 < own_n; c0 += 8 * TNW) {
                    float4v wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * TNW;
                        wv[u] = (c < own_n) ? *reinterpret_cast<const float4v*>(wr + size_t(c) * N_LOGITS)
                                            : float4v{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * TNW;
                        if (c < own_n) {
                            const float f = s_feat[c];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] = fmaf(f, wv[u][i], acc[i]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s_part[wave * 256 + lane * 4 + i] = acc[i];
            }
            __syncthreads();
            if (tid < N_LOGITS) {
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < TNW; ++w) t += s_part[w * 256 + tid];
                st_sc1_f32(LBase + size_t(m) * 256 + tid, t);
            }
        }
        phase += unsigned(C);
        cluster_arrive(counter);                       // ---- exchange: logits partials
        const float bdr = (tid < N_LOGITS) ? a.bdense[tid] : 0.f;
        cluster_wait(counter, phase, err);
        stamp(90);
        if (m == 0) {
            float* s_logit = s_part;                   // [256]  (s_part is dead after the barrier above)
            if (tid < N_LOGITS) {
                float pv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) pv[j] = (j < C) ? ld_sc1_f32(LBase + size_t(j) * 256 + tid) : 0.f;
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < C) t += pv[j];             // fixed member order
                t += bdr;
                s_logit[tid] = t;
                if (a.logits != nullptr) a.logits[size_t(crop) * N_LOGITS + tid] = t;
            }
            __syncthreads();
            // ============= decode (utils.py:7-11, whenet.py:28-33): wave h <-> head h ==============
            if (wave < 3) {
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
                    a.ypr[size_t(crop) * 3 + wave] = ex * 3.0f - ((wave == 0) ? 180.0f : 99.0f);
                    if (a.argmax != nullptr) a.argmax[size_t(crop) * 3 + wave] = mi;
                }
            }
        }
        stamp(91);
        // LBase / XB of this crop may be overwritten by the next crop only after member 0 has read them:
        // member 0 reads LBase right after the exchange above, and every member's next write to LBase /
        // XB lies behind >= 2 further cluster exchanges that member 0 takes part in -- no extra barrier.
        __syncthreads();
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host side: LDS / scratch planning and the launcher
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int LDS_LIMIT = 160 * 1024;
constexpr int FIXED_BYTES = (4 * 1152 + 64) * 4;

template <typename T>
int front_bytes(const TrunkBlock& b, int sub_tiles, int* off_e, int* off_dww, int* off_red) {
    const int SZ = int(sizeof(T));
    const int HWi = b.h_in * b.h_in;
    const int EW = (b.h_out - 1) * b.s + b.k;
    const int ccur = sub_tiles * 32;
    const int nstrip = b.h_out * (b.h_out / P);
    *off_e = align16(HWi * (b.cin * SZ + 16));
    *off_dww = *off_e + align16(EW * EW * (ccur * SZ + 16));
    *off_red = *off_dww + align16(b.k * b.k * ccur * 4);
    return *off_red + align16(nstrip * ccur * 4);
}

}  // namespace

// Fills the geometry-dependent fields of `blk` (sub_tiles, LDS offsets) for cluster size C and returns the
// plan (LDS bytes, scratch layout).  Pure host logic.
TrunkPlan plan_trunk(TrunkBlock* blk, int nblk, int dtype, int C, int head_nth, int head_cin) {
    WHENET_REQUIRE(C >= 1 && C <= 16 && nblk >= 1, WHENET_EINVAL, "trunk: cluster size must be 1..16");
    const int SZ = dtype == WHENET_F16 ? 2 : 4;
    TrunkPlan p{};
    p.C = C;
    int need = 0;
    size_t xmax = 0, dmax = 0, pmax = 0;
    for (int bi = 0; bi < nblk; ++bi) {
        TrunkBlock& b = blk[bi];
        WHENET_REQUIRE(b.cexp % 32 == 0 && b.cin % 8 == 0 && b.cout % 4 == 0 && b.h_out % P == 0 && b.r <= 64,
                       WHENET_EINVAL, "trunk: unsupported block geometry");
        const int tiles = b.cexp / 32;
        const int own_max = (tiles + C - 1) / C;
        int best = 0;
        for (int sub = own_max; sub >= 1; --sub) {
            int oe, od, orr;
            const int fb = dtype == WHENET_F16 ? front_bytes<half_t>(b, sub, &oe, &od, &orr)
                                               : front_bytes<float>(b, sub, &oe, &od, &orr);
            if (fb + FIXED_BYTES <= LDS_LIMIT) {
                best = sub;
                break;
            }
        }
        WHENET_REQUIRE(best >= 1, WHENET_EINVAL, "trunk: block does not fit the LDS budget");
        // balance the sub-chunks of the largest member: ceil(own / ceil(own / best))
        const int nsub = (own_max + best - 1) / best;
        b.sub_tiles = (own_max + nsub - 1) / nsub;
        int oe, od, orr;
        const int fb = dtype == WHENET_F16 ? front_bytes<half_t>(b, b.sub_tiles, &oe, &od, &orr)
                                           : front_bytes<float>(b, b.sub_tiles, &oe, &od, &orr);
        b.off_e = oe;
        b.off_dww = od;
        b.off_red = orr;
        const int HWo = b.h_out * b.h_out;
        int dg = align16(HWo * (own_max * 32 * SZ + 16));
        // the own k-steps of the project weights are staged in LDS next to D * gate when they fit
        const int kpt = 32 / (2 * (16 / SZ));
        const int wl = own_max * kpt * b.ntp * 1024;
        b.wp_lds = (dg + wl + FIXED_BYTES <= LDS_LIMIT) ? 1 : 0;
        if (b.wp_lds) dg += wl;
        need = std::max(need, std::max(fb, dg));
        xmax = std::max(xmax, std::max(size_t(b.h_in) * b.h_in * b.cin, size_t(HWo) * b.cout));
        dmax = std::max(dmax, size_t(HWo) * own_max * 32);
        pmax = std::max(pmax, size_t(HWo) * b.cout);
    }
    {   // head: X [49][cin] + strip partials + features + dense partials
        const int own_n = ((head_nth + C - 1) / C) * 32;
        const int head = align16(49 * (head_cin * SZ + 16)) + (3 * own_n + MAXW * 256) * 4;
        need = std::max(need, head);
    }
    p.fixed_off = align16(need);
    p.lds_bytes = size_t(p.fixed_off) + FIXED_BYTES;
    WHENET_REQUIRE(p.lds_bytes <= size_t(LDS_LIMIT), WHENET_EINVAL, "trunk: LDS budget exceeded");
    p.xmax = (xmax + 63) & ~size_t(63);
    p.dmax = (dmax + 63) & ~size_t(63);
    p.pmax = (pmax + 63) & ~size_t(63);
    size_t off = 2 * p.xmax * SZ;
    off = (off + 255) & ~size_t(255);
    p.off_d = off;
    off += size_t(C) * p.dmax * SZ;
    off = (off + 255) & ~size_t(255);
    p.off_p = off;
    off += size_t(C) * p.pmax * 4;
    p.off_r = off;
    off += size_t(C) * 64 * 4;
    p.off_l = off;
    off += size_t(C) * 256 * 4;
    p.scratch_stride = (off + 255) & ~size_t(255);
    return p;
}

template <typename T, int TNT>
void launch_trunk_t(const TrunkArgs& a, size_t lds_bytes, hipStream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {      // > 64 KiB of dynamic LDS needs the opt-in
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_trunk_kernel<T, TNT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((whenet_trunk_kernel<T, TNT>), dim3(unsigned(a.nclusters * a.C)), dim3(TNT), lds_bytes, stream, a);
}

void launch_trunk(const TrunkArgs& a, size_t lds_bytes, int dtype, int threads, hipStream_t stream) {
    WHENET_REQUIRE(a.nblk >= 1 && a.n >= 1 && a.blk != nullptr && a.C >= 1 && a.nclusters >= 1, WHENET_EINVAL,
                   "trunk kernel: bad arguments");
    WHENET_REQUIRE(threads == 512 || threads == 1024, WHENET_EINVAL, "trunk kernel: threads must be 512 or 1024");
    // the arrival counters (and error words) restart from zero on every launch / graph replay
    WHENET_HIP_CHECK(hipMemsetAsync(a.counters, 0, size_t(a.nclusters) * 16 * sizeof(unsigned), stream));
    if (dtype == WHENET_F16) {
        if (threads == 512) launch_trunk_t<half_t, 512>(a, lds_bytes, stream);
        else launch_trunk_t<half_t, 1024>(a, lds_bytes, stream);
    } else {
        if (threads == 512) launch_trunk_t<float, 512>(a, lds_bytes, stream);
        else launch_trunk_t<float, 1024>(a, lds_bytes, stream);
    }
    WHENET_HIP_CHECK(hipGetLastError());
}

std::string kernel_name_trunk(int dtype, int threads) {
    return std::string("whenet_trunk_kernel<") + (dtype == WHENET_F16 ? "_Float16" : "float") + ", " +
           std::to_string(threads) + ">";
}

}  // namespace whenet
