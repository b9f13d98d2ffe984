// Depthwise kxk convolution (TF 'SAME') + BatchNorm + Swish, NHWC, im2col-free.
//
// Reference: efficientnet 0.0.4 MBConvBlock's DepthwiseConv2D(k, strides, 'same', no bias) ->
// BN -> Swish, instantiated by /root/reference/whenet.py:8; 16 layers,
// k in {3,5}, stride in {1,2}, 34.5 M MACs per crop -- 5.7 FLOP/B at f16: HBM-bound.
//
// Mapping (gfx950):
//   * one workgroup = one crop x one channel chunk (CV 16-byte vectors) x one output tile of
//     TH rows x (7*NSX) columns.  7 divides every feature-map width of the network
//     (112, 56, 28, 14, 7), so a lane owns a strip of P = 7 consecutive output pixels of one
//     row for one 16-byte channel vector and keeps its P*V accumulators in registers;
//   * the input tile with its halo is staged ONCE into LDS with coalesced 16-byte loads (a
//     pixel's chunk of CV*16 B is contiguous in NHWC); 'SAME' padding is materialised there as
//     zeros, so the inner loops carry no bounds checks;
//   * each staged 16-byte vector a lane reads from LDS feeds up to k taps (register reuse along
//     the strip), i.e. k*((P-1)*s+k) LDS reads per P*k*k tap-pixels;
//   * BN is folded into the weights/bias on the host; Swish is applied in f32 before the single
//     rounding to the activation type;
//   * the squeeze-excite spatial mean needs a reduction over the whole map: every workgroup
//     writes the f32 sum of its tile's outputs to partial[crop][tile][channel] in a fixed
//     order (no atomics -> bitwise reproducible), se.hip finishes the mean.
// HBM bytes per launch: (H*H + Ho*Ho) * C * sizeof(T) per crop (halo re-reads are L2 hits).
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

constexpr int P = 7;    // output pixels per lane
constexpr int VC = 4;   // channels per compute lane (f32: one 16-byte vector, f16: 8 bytes)

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void whenet_dw_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ partial, int H, int Ho, int C, int pad,
                                                        int CV, int TH, int NSX, int tiles_x, int IH, int IW,
                                                        int w_off_bytes) {
    constexpr int VL = Vec<T>::V;                  // elements per 16-byte staging load
    using VLT = typename Vec<T>::type;
    using VCT = T __attribute__((ext_vector_type(VC)));
    constexpr int NIX = (P - 1) * S + K;           // input columns one strip touches

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* s_tile = reinterpret_cast<T*>(smem);                                    // [IH*IW][CC]
    float* s_red = reinterpret_cast<float*>(smem);                             // aliases the tile
    float* s_w = reinterpret_cast<float*>(smem + w_off_bytes);                 // [K*K][CC]

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int tile = blockIdx.x;
    const int tyi = tile / tiles_x;
    const int txi = tile - tyi * tiles_x;
    const int CC = CV * VL;                        // channels of this chunk
    const int c0 = blockIdx.y * CC;                // first channel of this chunk
    const int b = blockIdx.z;
    const int oy0 = tyi * TH;
    const int ox0 = txi * NSX * P;
    const int iy0 = oy0 * S - pad;
    const int ix0 = ox0 * S - pad;

    // ---- stage weights [K*K][CC] and the input tile (16-byte loads) ------------------------
    for (int i = tid; i < K * K * CC; i += nthreads) {
        const int tap = i / CC;
        const int c = i - tap * CC;
        s_w[i] = w[tap * C + c0 + c];
    }
    {
        const int NPL = nthreads / CV;             // pixel slots per pass
        const int cvl = tid % CV;
        const int pslot = tid / CV;
        if (pslot < NPL) {
            const T* src = in + (size_t(b) * H * H) * C + c0 + cvl * VL;
            const int npix = IH * IW;
            // (r, c) of this lane's pixel advance incrementally: one division per lane, none per load
            int r = pslot / IW, c = pslot - r * IW;
            const int dr = NPL / IW, dc = NPL - dr * IW;
            // batches of 8 independent 16-byte loads in flight, then 8 LDS writes
            for (int pix0 = pslot; pix0 < npix; pix0 += 8 * NPL) {
                VLT v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int iy = iy0 + r, ix = ix0 + c;
                    v[u] = vec_zero<T>();
                    if (pix0 + u * NPL < npix && iy >= 0 && iy < H && ix >= 0 && ix < H)
                        v[u] = *reinterpret_cast<const VLT*>(src + (size_t(iy) * H + ix) * C);
                    r += dr;
                    c += dc;
                    if (c >= IW) {
                        c -= IW;
                        ++r;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (pix0 + u * NPL < npix)
                        *reinterpret_cast<VLT*>(s_tile + size_t(pix0 + u * NPL) * CC + cvl * VL) = v[u];
            }
        }
    }
    __syncthreads();

    // ---- compute: lane = (4-channel group cg, strip sidx) ----------------------------------
    const int CG = CC / VC;
    const int NPC = nthreads / CG;                 // strips a workgroup can hold
    const int cg = tid % CG;
    const int sidx = tid / CG;
    const int ty = sidx / NSX;
    const int sx = sidx - ty * NSX;
    const int oy = oy0 + ty;
    const bool lane_ok = sidx < NPC;
    const bool active = lane_ok && (sidx < TH * NSX) && (oy < Ho);

    float acc[P][VC];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int v = 0; v < VC; ++v) acc[p][v] = 0.0f;

    if (active) {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            float wr[K][VC];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float4v wv = *reinterpret_cast<const float4v*>(s_w + (ky * K + kx) * CC + cg * VC);
#pragma unroll
                for (int v = 0; v < VC; ++v) wr[kx][v] = wv[v];
            }
            const T* row = s_tile + size_t((ty * S + ky) * IW + sx * P * S) * CC + cg * VC;
#pragma unroll
            for (int ix = 0; ix < NIX; ++ix) {
                const VCT xv = *reinterpret_cast<const VCT*>(row + ix * CC);
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
    }

    // ---- epilogue: bias (+folded BN), Swish, store, per-tile channel sums -------------------
    float sum[VC];
#pragma unroll
    for (int v = 0; v < VC; ++v) sum[v] = 0.0f;
    if (active) {
        const float4v bs = *reinterpret_cast<const float4v*>(bias + c0 + cg * VC);
        T* dst = out + ((size_t(b) * Ho + oy) * Ho + (ox0 + sx * P)) * C + c0 + cg * VC;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            VCT o;
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const float y = opaque_f32(conv_swish<T>(acc[p][v] + bs[v]));     // (pinned: stemdw.hip must reproduce these bits)
                sum[v] += y;
                o[v] = T(y);
            }
            *reinterpret_cast<VCT*>(dst + size_t(p) * C) = o;
        }
    }
    lds_barrier();                               // everyone is done reading s_tile
    if (lane_ok) {
#pragma unroll
        for (int v = 0; v < VC; ++v) s_red[sidx * CC + cg * VC + v] = sum[v];
    }
    lds_barrier();
    if (tid < CC) {
        float t = 0.0f;
        for (int s = 0; s < NPC; ++s) t += s_red[s * CC + tid];
        partial[(size_t(b) * gridDim.x + tile) * C + c0 + tid] = t;
    }
}

template <typename T, int K, int S>
void launch_t(const DwArgs& a, hipStream_t stream) {
    constexpr int VL = Vec<T>::V;
    const DwPlan& p = a.plan;
    dim3 grid(p.ntiles(), p.chunks, a.n);
    const int CC = p.CV * VL;
    size_t tile_bytes = size_t(p.IH) * p.IW * CC * sizeof(T);
    const size_t red_bytes = size_t(p.threads / (CC / VC)) * CC * sizeof(float);
    if (red_bytes > tile_bytes) tile_bytes = red_bytes;
    tile_bytes = (tile_bytes + 15) & ~size_t(15);
    const size_t w_bytes = size_t(K) * K * CC * sizeof(float);
    WHENET_REQUIRE(tile_bytes + w_bytes == p.lds_bytes, WHENET_EINVAL, "depthwise: plan/launch LDS mismatch");
    hipLaunchKernelGGL((whenet_dw_kernel<T, K, S>), grid, dim3(p.threads), tile_bytes + w_bytes, stream,
                       static_cast<const T*>(a.in), static_cast<T*>(a.out), a.w, a.bias, a.partial, a.H, a.Ho, a.C,
                       a.pad, p.CV, p.TH, p.NSX, p.tiles_x, p.IH, p.IW, int(tile_bytes));
    WHENET_HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_ks(const DwArgs& a, hipStream_t stream) {
    if (a.k == 3 && a.s == 1) launch_t<T, 3, 1>(a, stream);
    else if (a.k == 3 && a.s == 2) launch_t<T, 3, 2>(a, stream);
    else if (a.k == 5 && a.s == 1) launch_t<T, 5, 1>(a, stream);
    else if (a.k == 5 && a.s == 2) launch_t<T, 5, 2>(a, stream);
    else throw Error(WHENET_EINVAL, "depthwise: unsupported kernel/stride");
}

}  // namespace

// Tile-shape search.  Candidates: block size 256 (128 as well for stride 2, whose input
// footprint per output is 4x), CV = divisors of the layer's channel-vector count, NSX =
// divisors of Ho/7, TH = as many rows as the lanes allow.  Score = fraction of lanes doing
// useful work, derated by the halo overhead of the staged tile; LDS capped at 64 KiB so at
// least two workgroups fit a CU.
DwPlan plan_dw(int dtype, int k, int s, int H, int Ho, int C) {
    const int V = (dtype == WHENET_F16) ? 8 : 4;
    WHENET_REQUIRE(C % V == 0 && Ho % P == 0, WHENET_EINVAL, "depthwise: unsupported geometry");
    const int cvecs = C / V;
    const int strips_per_row = Ho / P;
    DwPlan best;
    double best_score = -1.0;
    const int tcount = (s == 2) ? 2 : 1;
    const int tcand[2] = {256, 128};
    for (int ti = 0; ti < tcount; ++ti) {
        const int threads = tcand[ti];
        for (int CV = 1; CV <= cvecs && CV <= 32; ++CV) {
            if (cvecs % CV) continue;
            const int CG = CV * V / VC;            // 4-channel compute lanes per strip
            if (CG > threads) continue;
            const int NS = threads / CG;           // strips a workgroup holds
            for (int NSX = 1; NSX <= strips_per_row; ++NSX) {
                if (strips_per_row % NSX) continue;
                if (NSX > NS) break;
                int TH = NS / NSX;
                if (TH > Ho) TH = Ho;
                const int tiles_y = ceil_div(Ho, TH);
                // shrink TH to the smallest value giving the same tile count (less halo, less waste)
                TH = ceil_div(Ho, tiles_y);
                const int TW = NSX * P;
                const int IH = (TH - 1) * s + k, IW = (TW - 1) * s + k;
                size_t tile_bytes = size_t(IH) * IW * CV * 16;
                const size_t red_bytes = size_t(NS) * CV * V * 4;   // [NS][CC] f32
                if (red_bytes > tile_bytes) tile_bytes = red_bytes;
                const size_t lds = ((tile_bytes + 15) & ~size_t(15)) + size_t(k) * k * CV * V * 4;
                if (lds > 64 * 1024) continue;
                const double lane_use = double(Ho) * NSX * CG / (double(tiles_y) * threads);
                const double halo = double(TH * s) * (TW * s) / (double(IH) * IW);   // <= 1
                double coalesce = (CV * 16 >= 64) ? 1.0 : 0.6 + 0.4 * (CV * 16) / 64.0;
                // a pixel's channels split over several chunks in pieces shorter than a 128-byte line: every chunk's workgroup
                // fetches the whole line (round 4, PMC: block 1's f32 depthwise conv -- 32 channels = 128 B per pixel, planned
                // as 2 chunks of 64 B -- fetched 222 MB per 64 crops for a 103 MB input, L2 hit rate 5 %)
                if (cvecs / CV > 1 && CV * 16 < 128) coalesce *= 0.5 + 0.5 * (CV * 16) / 128.0;
                // occupancy: how many waves a CU can hold with this LDS footprint (160 KiB per CU)
                int blocks_cu = int((160 * 1024) / lds);
                if (blocks_cu > 8) blocks_cu = 8;
                const double waves_cu = double(blocks_cu) * threads / 64.0;
                const double occ = waves_cu >= 16.0 ? 1.0 : waves_cu / 16.0;
                const double score = lane_use * (0.5 + 0.5 * halo) * coalesce * (0.4 + 0.6 * occ);
                if (score > best_score + 1e-9) {
                    best_score = score;
                    best.threads = threads;
                    best.CV = CV;
                    best.TH = TH;
                    best.NSX = NSX;
                    best.tiles_x = strips_per_row / NSX;
                    best.tiles_y = tiles_y;
                    best.chunks = cvecs / CV;
                    best.IH = IH;
                    best.IW = IW;
                    best.lds_bytes = lds;
                }
            }
        }
    }
    WHENET_REQUIRE(best_score > 0, WHENET_EINVAL, "depthwise: no tile plan fits");
    (void)H;
    return best;
}

void launch_dw(const DwArgs& a, int dtype, hipStream_t stream) {
    if (dtype == WHENET_F16) launch_ks<half_t>(a, stream);
    else launch_ks<float>(a, stream);
}

const char* kernel_name_dw(int dtype, int k, int s) {
    static const char* names[2][2][2] = {
        {{"whenet_dw_kernel<float, 3, 1>", "whenet_dw_kernel<float, 3, 2>"},
         {"whenet_dw_kernel<float, 5, 1>", "whenet_dw_kernel<float, 5, 2>"}},
        {{"whenet_dw_kernel<_Float16, 3, 1>", "whenet_dw_kernel<_Float16, 3, 2>"},
         {"whenet_dw_kernel<_Float16, 5, 1>", "whenet_dw_kernel<_Float16, 5, 2>"}}};
    return names[dtype == WHENET_F16][k == 5][s == 2];
}

}  // namespace whenet
