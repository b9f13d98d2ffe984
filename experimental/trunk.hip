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

constexpr int TNT = 512;       // lanes per workgroup: 8 waves, 2 per SIMD, 256 registers per lane (nothing spills; the
constexpr int TNW = TNT / 64;  // 1024-lane build spilled ~100 registers around every phase boundary and measured slower).
                               // A wave issues one VALU instruction per ~5.3 cycles (tools/probes/valu_probe.hip) and
                               // every phase is a chain of dependent memory / LDS / matrix latencies: a phase is as
                               // fast as its slowest wave, so the work is cut into about one task per wave
constexpr int RW = TNW / 2, RT = TNT / 2;     // waves / lanes of one role in the pipelined front phase
constexpr int P = 7;
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Pointers that come out of the block table in memory are generic to the compiler: every access through them
// would be a FLAT load (address-space check, counted on both the vector-memory and the LDS counter, so an
// `s_waitcnt` for an LDS read also waits for them).  All of them point to device memory: say so.
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename U> __device__ __forceinline__ const U GLOBAL_AS* gptr(const void* p) {
    return (const U GLOBAL_AS*)(p);
}

// ---- write-through / L1-bypassing accesses of exchanged data -------------------------------------
struct Buf {                  // wave-uniform buffer descriptor + helpers (offsets in bytes)
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ Buf(const void* base, unsigned nbytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(nbytes), 0x00020000)) {}
    __device__ __forceinline__ u32x4 ld16(unsigned off) const { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
    __device__ __forceinline__ void st16(unsigned off, u32x4 v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
};
__device__ __forceinline__ float ld_sc1_f32(const float* p) {
    return __hip_atomic_load((const float GLOBAL_AS*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_f32(float* p, float v) {
    __hip_atomic_store((float GLOBAL_AS*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1_u64(const void* p) {
    return __hip_atomic_load((const unsigned long long GLOBAL_AS*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_u64(void* p, unsigned long long v) {
    __hip_atomic_store((unsigned long long GLOBAL_AS*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// contiguous, balanced split of `total` items over `parts`; part `i` gets [lo, lo + cnt)
__host__ __device__ inline void split_range(int total, int parts, int i, int* lo, int* cnt) {
    const int base = total / parts, extra = total % parts;
    *lo = i * base + (i < extra ? i : extra);
    *cnt = base + (i < extra ? 1 : 0);
}

__host__ __device__ constexpr int align16(int x) { return (x + 15) & ~15; }

// ---- one (strip, tile) GEMM task: acc += sum_k W[tile][k] * act[row][k] ---------------------------
// Software-pipelined by hand: the U weight fragments AND the U activation fragments of a group are
// requested before its first MFMA (a dependent LDS round trip per MFMA otherwise: ~280 cycles per k-step
// measured).
template <typename T, int U, typename LoadW, typename LoadA>
__device__ __forceinline__ void gemm_task(float16v& acc, int ks0, int ks1, LoadW&& load_w, LoadA&& load_a) {
    using VT = typename Vec<T>::type;
    for (int ks = ks0; ks < ks1; ks += U) {
        VT w[U], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = (ks + u < ks1) ? load_w(ks + u) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u) av[u] = (ks + u < ks1) ? load_a(ks + u) : vec_zero<T>();
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ks + u < ks1) Mfma<T>::step(w[u], av[u], acc);      // (wave-uniform)
    }
}

// ---- depthwise taps of one sub-chunk: E (LDS, f32) -> D (own global scratch), strip channel sums -> s_red
// lane-task = VC channels x strip of 7 output pixels.  The expanded activation is kept in f32 in LDS: the
// taps are pure f32 FMAs on ds_read_b128 operands (a v_cvt_f32_f16 costs as much as two FMAs on this chip),
// and the expanded tensor is never rounded to the activation type at all.
template <typename T, int K, int S, int VC>
__device__ __forceinline__ void dw_tasks(const unsigned char* __restrict__ E, int EW, int EP, T* __restrict__ D,
                                         int dpitch, int dcol0, const float* __restrict__ s_dww, int wpitch,
                                         const float* __restrict__ s_bd, float* __restrict__ s_red, int Ho, int ccur,
                                         int t_lo, int t_n) {
    using VCT = T __attribute__((ext_vector_type(VC)));
    typedef float floatc __attribute__((ext_vector_type(VC)));
    constexpr int NIX = (P - 1) * S + K;
    const int CG = ccur / VC;
    const int spr = Ho / P;                          // strips per output row
    const int nstrip = Ho * spr;
    for (int lt = t_lo; lt < CG * nstrip; lt += t_n) {
        const int cg = lt % CG;
        const int sidx = lt / CG;
        const int oy = sidx / spr;
        const int sx = sidx - oy * spr;
        float acc[P][VC];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int v = 0; v < VC; ++v) acc[p][v] = 0.0f;
#pragma unroll 1   // one kernel row at a time: bounds the register footprint (K taps + NIX inputs in flight)
        for (int ky = 0; ky < K; ++ky) {
            floatc wr[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wr[kx] = *reinterpret_cast<const floatc*>(s_dww + (ky * K + kx) * wpitch + cg * VC);
            const unsigned char* row = E + size_t((oy * S + ky) * EW + sx * P * S) * EP + cg * VC * 4;
            floatc xin[NIX];
#pragma unroll
            for (int ix = 0; ix < NIX; ++ix) xin[ix] = *reinterpret_cast<const floatc*>(row + size_t(ix) * EP);
#pragma unroll
            for (int ix = 0; ix < NIX; ++ix) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int d = ix - kx;
                    if (d >= 0 && (d % S) == 0 && (d / S) < P) {
#pragma unroll
                        for (int v = 0; v < VC; ++v) acc[d / S][v] = fmaf(xin[ix][v], wr[kx][v], acc[d / S][v]);
                    }
                }
            }
        }
        const floatc bs = *reinterpret_cast<const floatc*>(s_bd + cg * VC);
        floatc sum;
#pragma unroll
        for (int v = 0; v < VC; ++v) sum[v] = 0.f;
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
        *reinterpret_cast<floatc*>(s_red + sidx * wpitch + cg * VC) = sum;
    }
}

// Barriers inside the kernel order LDS traffic only (lds_barrier: no vmcnt drain -- a __syncthreads() after
// the depthwise phase waited ~3 us for the acknowledgements of its global stores); global data is published
// by cluster_arrive, which drains explicitly.
template <typename T>
__global__ __launch_bounds__(TNT) void whenet_trunk_kernel(TrunkArgs a) {
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
    float* s_sum = reinterpret_cast<float*>(smem + a.fixed_off);      // [own_cap] own channel sums
    float* s_be = s_sum + a.own_cap;                                   // [own_cap] expand bias of the own channels
    float* s_bd = s_be + a.own_cap;                                    // [own_cap] depthwise bias of the own channels
    T* s_gate = reinterpret_cast<T*>(s_bd + a.own_cap);               // [own_cap] own gate, in the activation type
    float* s_r = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(s_gate) + a.own_cap * 4);   // [64]

    auto stamp = [&](int slot) {
        if (a.timing != nullptr && blockIdx.x == 0 && tid == 0) a.timing[slot] = wall_clock64();
    };

    bool pending_wait = false;                         // the last cluster barrier was arrived at but not yet waited for
    for (int crop = cluster; crop < a.n; crop += a.nclusters) {
        stamp(0);
        if (a.timing != nullptr && blockIdx.x == 0 && tid == 0) a.timing[92] = clock64();      // shader cycles
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
        for (int bi = 0; bi < a.nblk; ++bi) {
            const TrunkBlock B = a.blk[bi];            // uniform: scalar loads from the device table
            const VT GLOBAL_AS* g_we = gptr<VT>(B.we);
            const VT GLOBAL_AS* g_wp = gptr<VT>(B.wp);
            const float GLOBAL_AS* g_be = gptr<float>(B.be);
            const float GLOBAL_AS* g_wd = gptr<float>(B.wd);
            const float GLOBAL_AS* g_bd = gptr<float>(B.bd);
            const float GLOBAL_AS* g_w1t = gptr<float>(B.w1t);
            const float GLOBAL_AS* g_b1 = gptr<float>(B.b1);
            const float GLOBAL_AS* g_w2c = gptr<float>(B.w2c);
            const float GLOBAL_AS* g_b2 = gptr<float>(B.b2);
            const float GLOBAL_AS* g_bp = gptr<float>(B.bp);
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
            const int sub_ch = B.sub_tiles * 32;       // channels of a full sub-chunk (the last one may be narrower)
            const int EP = sub_ch * 4 + 16;            // pixel pitch of E (f32 + 16-byte pad)
            const int nsub = (tcnt + B.sub_tiles - 1) / B.sub_tiles;
            const bool dbuf = B.dbuf != 0;             // two E / taps / strip-sum buffers: expand(s+1) runs beside taps(s)
            unsigned char* X = smem;
            const int nstrip_i = (HWi + 31) >> 5, nstrip_o = (HWo + 31) >> 5;
            const int nstrip_dw = B.h_out * (B.h_out / P);
            const int ntap = B.k * B.k;
            auto Ebuf = [&](int s) -> unsigned char* { return smem + B.off_e + (dbuf ? (s & 1) * B.e_bytes : 0); };
            auto Wbuf = [&](int s) -> float* { return reinterpret_cast<float*>(smem + B.off_dww) + (dbuf ? (s & 1) * ntap * sub_ch : 0); };
            auto Rbuf = [&](int s) -> float* { return reinterpret_cast<float*>(smem + B.off_red) + (dbuf ? (s & 1) * nstrip_dw * sub_ch : 0); };
            auto sub_cc = [&](int s) -> int {           // channels of sub-chunk s
                const int left = tcnt - s * B.sub_tiles;
                return ((left < B.sub_tiles) ? left : B.sub_tiles) * 32;
            };
            dstamp(0);

            // depthwise taps of sub-chunk s -> its LDS buffer, by lanes [t_lo, t_lo + t_n) (global -> registers
            // first, LDS after: the loads of one call are all in flight together)
            auto load_taps = [&](int s, int t_lo, int t_n) {
                const int cc = sub_cc(s);
                float* dst = Wbuf(s);
                const float GLOBAL_AS* src = g_wd + c_own0 + s * sub_ch;
                const int total = ntap * (cc / 4);                   // float4 units
                for (int i0 = t_lo; i0 < total; i0 += 4 * t_n) {
                    float4v wv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * t_n;
                        const int tap = i / (cc / 4), c4 = i - tap * (cc / 4);
                        wv[u] = (i < total) ? *reinterpret_cast<const float4v GLOBAL_AS*>(src + size_t(tap) * B.cexp + c4 * 4)
                                            : float4v{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * t_n;
                        const int tap = i / (cc / 4), c4 = i - tap * (cc / 4);
                        if (i < total) *reinterpret_cast<float4v*>(dst + tap * sub_ch + c4 * 4) = wv[u];
                    }
                }
            };
            // expand (MFMA) of sub-chunk s -> E buffer, (strip, tile) tasks over waves [w_lo, w_lo + w_n)
            auto expand_tasks = [&](int s, int w_lo, int w_n) {
                const int ntile = sub_cc(s) / 32;
                unsigned char* E = Ebuf(s);
                for (int t = wave - w_lo; t < nstrip_i * ntile; t += w_n) {
                    const int tile = t / nstrip_i, strip = t - tile * nstrip_i;
                    const int p = strip * 32 + lm;
                    const bool valid = p < HWi;
                    const unsigned char* xrow = X + size_t(valid ? p : 0) * pin + g * V * SZ;
                    float16v acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    const VT GLOBAL_AS* wf = g_we + size_t(t0 + s * B.sub_tiles + tile) * 64 + lane;
                    const int wstride = B.nte * 64;
                    gemm_task<T, 7>(
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
                            const float4v bv = *reinterpret_cast<const float4v*>(s_be + s * sub_ch + nl);
                            float4v o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = swish_f<IsF32<T>::value>(acc[4 * qq + r] + bv[r]);
                            *reinterpret_cast<float4v*>(epix + nl * 4) = o;
                        }
                    }
                }
            };
            auto taps_tasks = [&](int s, int t_lo, int t_n) {
                const int cc = sub_cc(s);
                const int dcol0 = s * sub_ch;
                if (B.vc == 2) {
                    if (B.k == 3) dw_tasks<T, 3, 1, 2>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                    else if (B.s == 1) dw_tasks<T, 5, 1, 2>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                    else dw_tasks<T, 5, 2, 2>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                } else {
                    if (B.k == 3) dw_tasks<T, 3, 1, 4>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                    else if (B.s == 1) dw_tasks<T, 5, 1, 4>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                    else dw_tasks<T, 5, 2, 4>(Ebuf(s), EW, EP, DB, own_ch, dcol0, Wbuf(s), sub_ch, s_bd + dcol0, Rbuf(s), B.h_out, cc, t_lo, t_n);
                }
            };
            // channel sums of sub-chunk s over its strips, fixed order, by lanes t_lo .. (one lane per channel)
            auto colsum = [&](int s, int t_lo) {
                const int cc = sub_cc(s);
                const float* red = Rbuf(s);
                const int c = tid - t_lo;
                if (c >= 0 && c < cc) {
                    float t = 0.0f;
                    for (int q0 = 0; q0 < nstrip_dw; q0 += 14) {
                        float v[14];
#pragma unroll
                        for (int q = 0; q < 14; ++q) v[q] = (q0 + q < nstrip_dw) ? red[(q0 + q) * sub_ch + c] : 0.f;
#pragma unroll
                        for (int q = 0; q < 14; ++q)
                            if (q0 + q < nstrip_dw) t += v[q];
                    }
                    s_sum[s * sub_ch + c] = t;
                }
            };
            // depthwise taps of sub-chunk s in two halves: request (registers), commit (LDS) -- one float4 per lane
            auto taps_request = [&](int s, int t, int t_n, float4v& r0, float4v& r1) {
                const int cc = sub_cc(s);
                const float GLOBAL_AS* src = g_wd + c_own0 + s * sub_ch;
                const int total = ntap * (cc / 4);
                const int i0 = t, i1 = t + t_n;
                if (i0 < total) r0 = *reinterpret_cast<const float4v GLOBAL_AS*>(src + size_t(i0 / (cc / 4)) * B.cexp + (i0 % (cc / 4)) * 4);
                if (i1 < total) r1 = *reinterpret_cast<const float4v GLOBAL_AS*>(src + size_t(i1 / (cc / 4)) * B.cexp + (i1 % (cc / 4)) * 4);
            };
            auto taps_commit = [&](int s, int t, int t_n, const float4v& r0, const float4v& r1) {
                const int cc = sub_cc(s);
                float* dst = Wbuf(s);
                const int total = ntap * (cc / 4);
                const int i0 = t, i1 = t + t_n;
                if (i0 < total) *reinterpret_cast<float4v*>(dst + (i0 / (cc / 4)) * sub_ch + (i0 % (cc / 4)) * 4) = r0;
                if (i1 < total) *reinterpret_cast<float4v*>(dst + (i1 / (cc / 4)) * sub_ch + (i1 % (cc / 4)) * 4) = r1;
                if (total > 2 * t_n) {                 // (wide sub-chunks only)
                    const float GLOBAL_AS* src = g_wd + c_own0 + s * sub_ch;
                    for (int i = t + 2 * t_n; i < total; i += t_n)
                        *reinterpret_cast<float4v*>(dst + (i / (cc / 4)) * sub_ch + (i % (cc / 4)) * 4) =
                            *reinterpret_cast<const float4v GLOBAL_AS*>(src + size_t(i / (cc / 4)) * B.cexp + (i % (cc / 4)) * 4);
                }
            };

            // ---- independent of the block input (fills the wait for the previous block's output): biases of
            // the own channels, zeroed E buffers (the halo is TF 'SAME' padding of the EXPANDED tensor; every
            // sub-chunk rewrites the interior), depthwise taps of sub-chunk 0
            {
                for (int c = tid; c < own_ch; c += TNT) {
                    s_be[c] = g_be[c_own0 + c];
                    s_bd[c] = g_bd[c_own0 + c];
                }
                const int ez = (dbuf ? 2 : 1) * B.e_bytes / 16;
                float4v* e4 = reinterpret_cast<float4v*>(smem + B.off_e);
                for (int i = tid; i < ez; i += TNT) e4[i] = float4v{0.f, 0.f, 0.f, 0.f};
                load_taps(0, tid, TNT);
            }
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
                for (int i0 = tid; i0 < total; i0 += 3 * TNT) {
                    u32x4 xv[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int i = i0 + u * TNT;
                        xv[u] = xb.ld16(unsigned(i < total ? i : 0) * 16u);
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int i = i0 + u * TNT;
                        if (i < total) {
                            const int r = i / vpr, v = i - r * vpr;
                            *reinterpret_cast<u32x4*>(X + size_t(r) * pin + v * 16) = xv[u];
                        }
                    }
                }
            }
            lds_barrier();
            stamp(1 + bi * 8 + 0);
            dstamp(3);

            // ================= phase 1: expand (MFMA) -> E, depthwise -> D, channel sums ==========
            if (dbuf) {
                // software pipeline over the sub-chunks, one barrier per step: waves 0..7 run the depthwise
                // lane-tasks of sub-chunk s while waves 8..15 expand sub-chunk s + 1 into the other E buffer
                // (one (strip, tile) task each), fetch its taps and fold the strip sums of s - 1: matrix cores,
                // VALU and LDS of the CU all stay busy and no wave runs more than one task per step
                expand_tasks(0, 0, TNW);
                lds_barrier();
                dstamp(4);
                for (int s = 0; s < nsub; ++s) {
                    if (wave < RW) {
                        taps_tasks(s, tid, RT);
                    } else {
                        float4v tr0 = {0.f, 0.f, 0.f, 0.f}, tr1 = tr0;
                        const bool more = s + 1 < nsub;
                        if (more) {
                            taps_request(s + 1, tid - RT, RT, tr0, tr1);
                            expand_tasks(s + 1, RW, RW);
                            taps_commit(s + 1, tid - RT, RT, tr0, tr1);
                        }
                        if (s > 0) colsum(s - 1, TNT - ((sub_ch + 63) & ~63));      // (lanes of the last waves: fewest expand tasks)
                    }
                    lds_barrier();
                    if (s < 8) dstamp(5 + s);
                }
                colsum(nsub - 1, 0);
            } else {
                for (int s = 0; s < nsub; ++s) {
                    if (s > 0) {                       // E: zero again (single buffer), park the taps
                        float4v* e4 = reinterpret_cast<float4v*>(smem + B.off_e);
                        for (int i = tid; i < B.e_bytes / 16; i += TNT) e4[i] = float4v{0.f, 0.f, 0.f, 0.f};
                        load_taps(s, tid, TNT);
                        lds_barrier();
                    }
                    expand_tasks(s, 0, TNW);
                    lds_barrier();
                    taps_tasks(s, tid, TNT);
                    lds_barrier();
                    colsum(s, 0);
                    if (s < 4) dstamp(4 + s);
                    lds_barrier();
                }
            }
            // reduce-conv rows of this lane for the squeeze-excite partial: requested before the last barrier
            // wave w owns outputs j = w, w + TNW, ..
            constexpr int JR = 64 / TNW;               // R <= 64
            constexpr int CU4 = 5;                     // own channels per lane: <= 5 (own_ch <= 320), else looped
            float w1r[JR][CU4];
#pragma unroll
            for (int jj = 0; jj < JR; ++jj) {
                const int j = wave + TNW * jj;
                const float GLOBAL_AS* wrow = g_w1t + size_t(j < B.r ? j : 0) * B.cexp + c_own0;
#pragma unroll
                for (int u = 0; u < CU4; ++u) {
                    const int c = lane + 64 * u;
                    w1r[jj][u] = (j < B.r && c < own_ch) ? wrow[c] : 0.f;
                }
            }
            lds_barrier();
            stamp(1 + bi * 8 + 3);
            dstamp(13);

            // ================= squeeze-excite, first half: own share of the reduce conv =========
            {
                float* rb = RBase + size_t(m) * 64;
#pragma unroll
                for (int jj = 0; jj < JR; ++jj) {
                    const int j = wave + TNW * jj;
                    float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < CU4; ++u) {
                        const int c = lane + 64 * u;
                        if (c < own_ch) p4[u & 3] = fmaf(s_sum[c], w1r[jj][u], p4[u & 3]);
                    }
                    if (j < B.r) {
                        const float GLOBAL_AS* wrow = g_w1t + size_t(j) * B.cexp + c_own0;
                        for (int c = lane + 64 * CU4; c < own_ch; c += 64) p4[0] = fmaf(s_sum[c], wrow[c], p4[0]);   // (small clusters)
                    }
                    float t = (p4[0] + p4[1]) + (p4[2] + p4[3]);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                    if (lane == 0 && j < B.r) st_sc1_f32(rb + j, t);
                }
            }
            dstamp(14);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 1: SE partial vectors (+ own D stores drained)
            dstamp(15);

            // ---- while the other members arrive: everything of the second half that does not depend on them:
            // the own k-steps of the project weights, the own depthwise output and the lane's share of the
            // excite kernel are requested now and consumed after the wait
            unsigned char* Dg = smem;                  // X and E are dead from here on
            const int pd = own_ch * SZ + 16;
            const int ks0 = t0 * KPT, ks1 = (t0 + tcnt) * KPT;
            VT* Wl = reinterpret_cast<VT*>(smem + align16(HWo * pd));       // gated project weights of the own k-steps
            constexpr int WR = 7;
            const int wtotal = B.wp_lds ? (ks1 - ks0) * B.ntp * 64 : 0;
            VT wreg[WR];
            {
                const VT GLOBAL_AS* wsrc = g_wp + size_t(ks0) * B.ntp * 64;
#pragma unroll
                for (int u = 0; u < WR; ++u) wreg[u] = (tid + u * TNT < wtotal) ? wsrc[tid + u * TNT] : vec_zero<T>();
            }
            constexpr int DR = 4;
            const int vprd = own_ch * SZ / 16;
            const int dtotal = HWo * vprd;
            const Buf db(DB, unsigned(HWo * own_ch * SZ));
            u32x4 dreg[DR];
#pragma unroll
            for (int u = 0; u < DR; ++u) {
                const int i = tid + u * TNT;
                dreg[u] = db.ld16(unsigned(i < dtotal ? i : 0) * 16u);
            }
            // excite: two lanes per own channel, lane h takes the float4 pieces q = h, h + 2, .. of the row
            constexpr int GV = 6;                      // RP <= 48: 12 float4 pieces per row
            const int gch = tid >> 1, gh = tid & 1;
            float4v w2r[GV];
            float b2r = 0.f;
            {
                const int c = (gch < own_ch) ? gch : 0;
                const float4v GLOBAL_AS* wrow = reinterpret_cast<const float4v GLOBAL_AS*>(g_w2c + size_t(c_own0 + c) * B.rp);
#pragma unroll
                for (int j = 0; j < GV; ++j) w2r[j] = (4 * (2 * j + gh) < B.rp) ? wrow[2 * j + gh] : float4v{0.f, 0.f, 0.f, 0.f};
                b2r = g_b2[c_own0 + c];
            }
            const float b1r = (tid < B.r) ? g_b1[tid] : 0.f;
            dstamp(16);
            cluster_wait(counter, phase, err);
            stamp(1 + bi * 8 + 4);
            dstamp(17);

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
            lds_barrier();
            dstamp(18);
            for (int c = gch; c < own_ch; c += TNT / 2) {
                float t0a = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                float bias2 = b2r;
                if (c == gch) {
#pragma unroll
                    for (int j = 0; j < GV; ++j) {
                        const int q = 2 * j + gh;
                        if (4 * q < B.rp) {
                            t0a = fmaf(s_r[4 * q], w2r[j][0], t0a);
                            t1 = fmaf(s_r[4 * q + 1], w2r[j][1], t1);
                            t2 = fmaf(s_r[4 * q + 2], w2r[j][2], t2);
                            t3 = fmaf(s_r[4 * q + 3], w2r[j][3], t3);
                        }
                    }
                } else {                               // (own_ch > 512: small clusters only)
                    const float4v GLOBAL_AS* wrow = reinterpret_cast<const float4v GLOBAL_AS*>(g_w2c + size_t(c_own0 + c) * B.rp);
                    bias2 = g_b2[c_own0 + c];
                    for (int q = gh; 4 * q < B.rp; q += 2) {
                        const float4v wv = wrow[q];
                        t0a = fmaf(s_r[4 * q], wv[0], t0a);
                        t1 = fmaf(s_r[4 * q + 1], wv[1], t1);
                        t2 = fmaf(s_r[4 * q + 2], wv[2], t2);
                        t3 = fmaf(s_r[4 * q + 3], wv[3], t3);
                    }
                }
                const float mine = (t0a + t1) + (t2 + t3);
                const float other = __shfl_xor(mine, 1, 64);
                const float tot = gh ? (other + mine) : (mine + other);          // (even pieces) + (odd pieces)
                if (gh == 0) s_gate[c] = T(sigmoid_f<true>(bias2 + tot));
            }
            lds_barrier();
            dstamp(19);

            // ================= project operands into LDS: D (own channels) and gate * W (own k-steps) ==
            // the gate scales the K dimension of the product, so it is folded into the weight fragments
            // (one rounding to the activation type, as the activation * gate product had before)
            {
                const bool gate_w = B.wp_lds != 0;
                auto put = [&](int i, u32x4 raw) {
                    const int r = i / vprd, v = i - r * vprd;
                    VT dv = as_vec<T>(raw);
                    if (!gate_w) dv = dv * *reinterpret_cast<const VT*>(s_gate + v * V);
                    *reinterpret_cast<VT*>(Dg + size_t(r) * pd + v * 16) = dv;
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
                if (gate_w) {
                    auto putw = [&](int i, VT wv) {
                        // fragment i = (k-step, tile, lane): its V weights sit at k = ks * 2V + (lane >> 5) * V + e
                        const int ksl = i / (B.ntp * 64);
                        const int kl = ksl * 2 * V + ((i >> 5) & 1) * V;
                        Wl[i] = wv * *reinterpret_cast<const VT*>(s_gate + kl);
                    };
#pragma unroll
                    for (int u = 0; u < WR; ++u)
                        if (tid + u * TNT < wtotal) putw(tid + u * TNT, wreg[u]);
                    const VT GLOBAL_AS* wsrc = g_wp + size_t(ks0) * B.ntp * 64;
                    for (int i = tid + WR * TNT; i < wtotal; i += TNT) putw(i, wsrc[i]);
                }
            }
            lds_barrier();
            stamp(1 + bi * 8 + 5);
            dstamp(20);

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
                        gemm_task<T, 6>(acc, ks0, ks1, [&](int ks) -> VT { return wl[size_t(ks - ks0) * wstride]; }, load_d);
                    } else {
                        const VT GLOBAL_AS* wf = g_wp + size_t(tile) * 64 + lane;
                        const int wstride = B.ntp * 64;
                        gemm_task<T, 6>(acc, ks0, ks1, [&](int ks) -> VT { return wf[size_t(ks) * wstride]; }, load_d);
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
            dstamp(21);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 2: project partials
            dstamp(22);

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
                // bias and skip of this lane's first units do not depend on the partials: loaded before the wait
                constexpr int RU = 2;
                OT skip0[RU];
                float4v bias0[RU];
#pragma unroll
                for (int q = 0; q < RU; ++q) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) skip0[q][r] = T(0);
                    bias0[q] = float4v{0.f, 0.f, 0.f, 0.f};
                    const int i = tid + q * TNT;
                    if (i < ucnt) {
                        bias0[q] = *reinterpret_cast<const float4v GLOBAL_AS*>(g_bp + ((u0 + i) % upr) * 4);
                        if (B.has_skip) skip0[q] = load_skip(u0 + i);
                    }
                }
                cluster_wait(counter, phase, err);
                stamp(1 + bi * 8 + 6);
                dstamp(23);
                for (int i0 = tid; i0 < ucnt; i0 += RU * TNT) {
                    float y[RU][4];
#pragma unroll
                    for (int q = 0; q < RU; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[q][r] = 0.f;
                    for (int j0 = 0; j0 < C; j0 += 4) {            // fixed member order, 4 x RU loads in flight
                        u32x4 raw[RU][4];
#pragma unroll
                        for (int q = 0; q < RU; ++q) {
                            const int i = i0 + q * TNT;
                            const int u = u0 + (i < ucnt ? i : 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                raw[q][j] = pall.ld16(unsigned((size_t(j0 + j < C ? j0 + j : j0) * a.pmax + size_t(u) * 4) * 4));
                        }
#pragma unroll
                        for (int q = 0; q < RU; ++q)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (j0 + j < C) {
                                    float4v pv;
                                    __builtin_memcpy(&pv, &raw[q][j], 16);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) y[q][r] += pv[r];
                                }
                            }
                    }
#pragma unroll
                    for (int q = 0; q < RU; ++q) {
                        const int i = i0 + q * TNT;
                        if (i >= ucnt) continue;
                        const int u = u0 + i;
                        float4v bv = bias0[q];
                        OT rv = skip0[q];
                        if (i0 != tid) {               // (more than RU units per lane: small clusters only)
                            bv = *reinterpret_cast<const float4v GLOBAL_AS*>(g_bp + (u % upr) * 4);
                            if (B.has_skip) rv = load_skip(u);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[q][r] += bv[r];
                        if (B.has_skip) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[q][r] += float(rv[r]);
                        }
                        OT o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = T(y[q][r]);
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
                }
                cur = nxt;
            }
            dstamp(24);
            phase += unsigned(C);
            cluster_arrive(counter);                   // ---- exchange 3: block output (waited for by its consumer)
            pending_wait = true;
            stamp(1 + bi * 8 + 7);
            dstamp(25);
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
        lds_barrier();
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
                gemm_task<T, 6>(
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
            lds_barrier();
            for (int c = tid; c < own_n; c += TNT) {
                const float f = (s_fp[c] + s_fp[own_n + c]) * (1.0f / 49.0f);
                s_feat[c] = f;
                if (a.feat != nullptr) a.feat[size_t(crop) * FEAT + ht0 * 32 + c] = f;
            }
            lds_barrier();
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
            lds_barrier();
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
            lds_barrier();
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
        if (a.timing != nullptr && blockIdx.x == 0 && tid == 0) a.timing[93] = clock64();
        // LBase / XB of this crop may be overwritten by the next crop only after member 0 has read them:
        // member 0 reads LBase right after the exchange above, and every member's next write to LBase /
        // XB lies behind >= 2 further cluster exchanges that member 0 takes part in -- no extra barrier.
        lds_barrier();
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host side: LDS / scratch planning and the launcher
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int LDS_LIMIT = 160 * 1024;

inline int fixed_bytes(int own_cap) { return own_cap * 4 * 4 + 64 * 4; }     // s_sum, s_be, s_bd, s_gate + s_r

// LDS bytes of the front phase of block b with sub-chunks of `sub` tiles, single / double buffered
int front_bytes(const TrunkBlock& b, int SZ, int sub, bool dbuf, int* off_e, int* e_bytes, int* off_dww, int* off_red) {
    const int HWi = b.h_in * b.h_in;
    const int EW = (b.h_out - 1) * b.s + b.k;
    const int cc = sub * 32;
    const int nstrip = b.h_out * (b.h_out / P);
    const int nb = dbuf ? 2 : 1;
    *off_e = align16(HWi * (b.cin * SZ + 16));
    *e_bytes = align16(EW * EW * (cc * 4 + 16));
    *off_dww = *off_e + nb * *e_bytes;
    *off_red = *off_dww + nb * align16(b.k * b.k * cc * 4);
    return *off_red + nb * align16(nstrip * cc * 4);
}

}  // namespace

// Fills the geometry-dependent fields of `blk` (sub_tiles, buffering, LDS offsets) for cluster size C and
// returns the plan (LDS bytes, scratch layout).  Pure host logic.
TrunkPlan plan_trunk(TrunkBlock* blk, int nblk, int dtype, int C, int head_nth, int head_cin) {
    WHENET_REQUIRE(C >= 1 && C <= 16 && nblk >= 1, WHENET_EINVAL, "trunk: cluster size must be 1..16");
    const int SZ = dtype == WHENET_F16 ? 2 : 4;
    TrunkPlan p{};
    p.C = C;
    int need = 0, own_cap = 0;
    size_t xmax = 0, dmax = 0, pmax = 0;
    for (int bi = 0; bi < nblk; ++bi) own_cap = std::max(own_cap, ((blk[bi].cexp / 32 + C - 1) / C) * 32);
    const int FIXED = fixed_bytes(own_cap);
    for (int bi = 0; bi < nblk; ++bi) {
        TrunkBlock& b = blk[bi];
        WHENET_REQUIRE(b.cexp % 32 == 0 && b.cin % 8 == 0 && b.cout % 4 == 0 && b.h_out % P == 0 && b.r <= 64,
                       WHENET_EINVAL, "trunk: unsupported block geometry");
        const int tiles = b.cexp / 32;
        const int own_max = (tiles + C - 1) / C;
        const int nstrip_i = (b.h_in * b.h_in + 31) / 32;
        const int nstrip_dw = b.h_out * (b.h_out / P);
        // Pipelined form (two buffers): 8 waves run depthwise lane-tasks (2 channels x 7 pixels per lane) while 8
        // waves expand the next sub-chunk, one (strip, tile) task per wave.  Otherwise one buffer and all 16
        // waves per phase.
        int best = 0;
        bool best_dbuf = false;
        for (int pass = 0; pass < 2 && best == 0; ++pass) {
            const bool dbuf = pass == 0;
            const int lanes = dbuf ? RT : TNT, waves = dbuf ? RW : TNW;
            int want = std::min(lanes / (16 * nstrip_dw), waves / nstrip_i);      // 16 lane-tasks per tile per strip
            want = std::max(1, std::min(want, own_max));
            for (int sub = want; sub >= 1; --sub) {
                int oe, eb, od, orr;
                if (front_bytes(b, SZ, sub, dbuf, &oe, &eb, &od, &orr) + FIXED <= LDS_LIMIT) {
                    best = sub;
                    best_dbuf = dbuf;
                    break;
                }
            }
        }
        WHENET_REQUIRE(best >= 1, WHENET_EINVAL, "trunk: block does not fit the LDS budget (cluster too small?)");
        // balance the sub-chunks of the largest member: ceil(own / ceil(own / best))
        const int nsub = (own_max + best - 1) / best;
        b.sub_tiles = (own_max + nsub - 1) / nsub;
        b.dbuf = best_dbuf ? 1 : 0;
        b.vc = (b.sub_tiles * 16 * nstrip_dw <= (best_dbuf ? RT : TNT)) ? 2 : 4;      // 2 channels per lane while every lane-task finds a lane
        int oe, eb, od, orr;
        const int fb = front_bytes(b, SZ, b.sub_tiles, best_dbuf, &oe, &eb, &od, &orr);
        b.off_e = oe;
        b.e_bytes = eb;
        b.off_dww = od;
        b.off_red = orr;
        const int HWo = b.h_out * b.h_out;
        int dg = align16(HWo * (own_max * 32 * SZ + 16));
        WHENET_REQUIRE(dg + FIXED <= LDS_LIMIT, WHENET_EINVAL,
                       "trunk: a member's depthwise output does not fit LDS (cluster too small)");
        // the own k-steps of the project weights are staged in LDS (gate folded in) next to D when they fit
        const int kpt = 32 / (2 * (16 / SZ));
        const int wl = own_max * kpt * b.ntp * 1024;
        b.wp_lds = (dg + wl + FIXED <= LDS_LIMIT) ? 1 : 0;
        if (b.wp_lds) dg += wl;
        need = std::max(need, std::max(fb, dg));
        xmax = std::max(xmax, std::max(size_t(b.h_in) * b.h_in * b.cin, size_t(HWo) * b.cout));
        dmax = std::max(dmax, size_t(HWo) * own_max * 32);
        pmax = std::max(pmax, size_t(HWo) * b.cout);
    }
    {   // head: X [49][cin] + strip partials + features + dense partials
        const int own_n = ((head_nth + C - 1) / C) * 32;
        const int head = align16(49 * (head_cin * SZ + 16)) + (3 * own_n + TNW * 256) * 4;
        need = std::max(need, head);
    }
    p.own_cap = own_cap;
    p.fixed_off = align16(need);
    p.lds_bytes = size_t(p.fixed_off) + FIXED;
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

template <typename T>
void launch_trunk_t(const TrunkArgs& a, size_t lds_bytes, hipStream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    WHENET_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {      // > 64 KiB of dynamic LDS needs the opt-in
        WHENET_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(whenet_trunk_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((whenet_trunk_kernel<T>), dim3(unsigned(a.nclusters * a.C)), dim3(TNT), lds_bytes, stream, a);
}

void launch_trunk(const TrunkArgs& a, size_t lds_bytes, int dtype, hipStream_t stream) {
    WHENET_REQUIRE(a.nblk >= 1 && a.n >= 1 && a.blk != nullptr && a.C >= 1 && a.nclusters >= 1, WHENET_EINVAL,
                   "trunk kernel: bad arguments");
    // the arrival counters (and error words) restart from zero on every launch / graph replay
    WHENET_HIP_CHECK(hipMemsetAsync(a.counters, 0, size_t(a.nclusters) * 16 * sizeof(unsigned), stream));
    if (dtype == WHENET_F16) launch_trunk_t<half_t>(a, lds_bytes, stream);
    else launch_trunk_t<float>(a, lds_bytes, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

std::string kernel_name_trunk(int dtype) {
    return std::string("whenet_trunk_kernel<") + (dtype == WHENET_F16 ? "_Float16" : "float") + ">";
}

}  // namespace whenet
