// Stem conv fused with block 1's depthwise conv (round 4; f16, and f32 further down): uint8 crop -> normalise -> Conv2D(32, 3x3, s2) + BN + Swish
// -> DepthwiseConv2D(3x3, s1) + BN + Swish, plus the per-tile channel sums of block 1's squeeze-excite.
//
// Reference: /root/reference/whenet.py:23-26 (normalise) feeding efficientnet 0.0.4's stem and the first MBConv block's
// depthwise stage (whenet.py:8).  Block 1 has no expand conv: its depthwise conv reads the stem's
// 112 x 112 x 32 output directly.  As two kernels (stem.hip, dw.hip) that tensor is written and read back -- 51 + 64 MB per
// 64 crops, the two most HBM-bound launches of the chain (2.4 and 3.5 TB/s; 26 + 33 us).  Here it only exists as an LDS tile.
//
// A workgroup owns dw.hip's tile of block 1 -- 16 x 14 output pixels x 32 channels of one crop, i.e. plan_dw()'s (CV 4, TH 16,
// NSX 2) plan, so the per-tile channel sums land where se.hip expects them -- and computes the 18 x 16 stem pixels under it
// (halo 1: 1.29 x the stem arithmetic, which is 81 MFMAs per workgroup) from a 37 x 33 pixel patch of the crop:
//   1. the patch's bytes (26 aligned dwords per row), the LUT and the stem weight fragments -- both already split into binary16
//      hi | lo as stem.hip splits them, once per model on the host (StemDwTable) -- are requested together; the patch goes
//      through the LUT into LDS, 16 bytes (4 values) per store;
//   2. 9 strips of 32 stem pixels over the 4 waves, stem.hip's arithmetic instruction for instruction (k-step = kernel row,
//      hi/lo split products accumulated in f32, BN bias + Swish, one rounding to f16) -> the depthwise input tile in LDS
//      [288 pixels][32 channels]; stem pixels outside the 112 x 112 map are the depthwise conv's 'SAME' zeros;
//   3. dw.hip's compute and epilogue on that tile, instruction for instruction (lane = 4 channels x a strip of 7 output pixels).
// The results are BITWISE those of the two kernels (tests/test_gpu_parity.py: option stem_fuse = 0 against 1).
// HBM bytes per crop: 150,528 in (1.36 x through L2) + 802,816 out.
#include "device_math.h"
#include "kernels.h"
#include "stamps.h"

namespace whenet {

namespace {

constexpr int SD_TH = 16, SD_NSX = 2, SD_TILES_X = 8, SD_TILES_Y = 7;       // plan_dw(f16, 3, 1, 112, 112, 32)
constexpr int SD_IH = SD_TH + 2, SD_IW = SD_NSX * 7 + 2;                    // 18 x 16 stem pixels per tile
constexpr int SD_NPIX = SD_IH * SD_IW, SD_NSTRIP = SD_NPIX / 32;            // 288, 9
constexpr int SD_PR = 2 * SD_IH + 1, SD_PC = 2 * SD_IW + 1;                 // 37 x 33 input pixels per tile
constexpr int SD_RDW = 26;                                                  // aligned dwords per patch row (2 + 99 + 3 bytes)
constexpr int SD_ROWW = 4 * SD_RDW;                                         // staged row: the 104 bytes' values, value v at index v + 2
constexpr int SD_C = 32, SD_CG = SD_C / 4, SD_P = 7;
static_assert(SD_NPIX % 32 == 0 && SD_TH * SD_NSX * SD_CG == 256, "tile / lane geometry");

#ifdef WHENET_STEMDW_DEBUG
__device__ half_t* g_stemdw_dbg = nullptr;
#endif

__global__ __launch_bounds__(256) void whenet_stemdw_kernel(const uint8_t* __restrict__ in, half_t* __restrict__ out,
                                                            const StemDwTable* __restrict__ tab, const float* __restrict__ bias,
                                                            const float* __restrict__ wd, const float* __restrict__ bd,
                                                            float* __restrict__ partial) {
    __shared__ uint32_t s_lut[3 * 256];                                      // packed binary16 (hi | lo << 16), as stem.hip
    __shared__ __attribute__((aligned(16))) uint32_t s_img[SD_PR * SD_ROWW + 8];    // (+8: the last row's fragment reads run past it)
    __shared__ __attribute__((aligned(16))) half_t s_tile[SD_NPIX * SD_C];   // the depthwise input tile; aliased by s_red
    __shared__ __attribute__((aligned(16))) float s_w[9 * SD_C];
    float* s_red = reinterpret_cast<float*>(s_tile);                         // [32 strips][32 channels]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int tile = blockIdx.x;
    const int tyi = tile / SD_TILES_X, txi = tile - tyi * SD_TILES_X;
    const int b = blockIdx.y;
    const int oy0 = tyi * SD_TH, ox0 = txi * SD_NSX * SD_P;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;                                  // stem pixel of the tile's corner (may be -1)
    STAMP(0);

    // ---- the patch, the LUT, the stem weight fragments, the depthwise weights: one global round trip ------------------------
    // patch row r = input row 2 * sy0 + r; its 99 values start at byte 6 * sx0 of the row (== 2 mod 4): dword j of the row
    // covers values 4j - 2 .. 4j + 1.  Image rows are 168 dwords: a dword is inside the row or outside it, never across.
    constexpr int NLD = (SD_PR * SD_RDW + 255) / 256;                        // 4
    const uint8_t* img = in + size_t(b) * IMG * IMG * 3;
    const int byte0 = 6 * sx0 - 2;                                           // (multiple of 4; -8 for the leftmost tiles)
    uint32_t raw[NLD];
    bool rok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        const int r = d / SD_RDW, j = d - r * SD_RDW;
        const int iy = 2 * sy0 + r;
        const int bo = byte0 + 4 * j;
        rok[i] = d < SD_PR * SD_RDW && iy >= 0 && iy < IMG && bo >= 0 && bo < IMG * 3;     // (else: padding, zero)
        raw[i] = rok[i] ? *reinterpret_cast<const uint32_t*>(img + size_t(iy) * (IMG * 3) + bo) : 0u;
    }
    uint32_t lutv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) lutv[i] = tab->lut[tid + 256 * i];
    half8 whi[3], wlo[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        whi[ky] = tab->whi[ky][lane];
        wlo[ky] = tab->wlo[ky][lane];
    }
    float4v bv[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) bv[qq] = *reinterpret_cast<const float4v*>(bias + 8 * qq + 4 * g);
    const float wdv0 = wd[tid], wdv1 = (tid < 9 * SD_C - 256) ? wd[256 + tid] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) s_lut[tid + 256 * i] = lutv[i];
    s_w[tid] = wdv0;
    if (tid < 9 * SD_C - 256) s_w[256 + tid] = wdv1;
    if (tid < 8) s_img[SD_PR * SD_ROWW + tid] = 0u;
    __syncthreads();
    STAMP(1);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        if (d >= SD_PR * SD_RDW) continue;
        const int j = d % SD_RDW;
        int ch = (j + 1) % 3;                                                 // channel of value 4j - 2 (values start at a pixel)
        uint4 val;
        uint32_t* vp = reinterpret_cast<uint32_t*>(&val);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (dwords outside the image were not loaded: raw = 0, the lookup is harmless and the result discarded)
            const uint32_t t = s_lut[ch * 256 + ((raw[i] >> (8 * q)) & 0xff)];
            vp[q] = rok[i] ? t : 0u;
            ch = (ch == 2) ? 0 : ch + 1;
        }
        *reinterpret_cast<uint4*>(&s_img[4 * d]) = val;                       // row r, values 4j - 2 .. 4j + 1 at index 4j .. 4j + 3
    }
    __syncthreads();
    STAMP(2);

    // ---- the stem conv of the tile's 288 pixels -> s_tile (stem.hip's strip loop) -------------------------------------------
    for (int strip = wave; strip < SD_NSTRIP; strip += 4) {
        const int p = strip * 32 + lm;
        const int pr = p / SD_IW, pc = p - pr * SD_IW;
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // 8 packed values of this lane's k-group.  g = 1 holds the ninth value of the kernel row and seven that belong to
            // the next pixels: their weights are zero (the fragment table), the values finite -- stem.hip zeroes them instead
            const uint32_t* row = &s_img[(2 * pr + ky) * SD_ROWW + 2 + pc * 6 + 8 * g];
            uint32_t d[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 v = *reinterpret_cast<const uint2*>(row + 2 * i);
                d[2 * i] = v.x;
                d[2 * i + 1] = v.y;
            }
            uint32_t ph[4], pl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ph[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
                pl[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
            }
            const half8 xhi = __builtin_bit_cast(half8, uint4{ph[0], ph[1], ph[2], ph[3]});
            const half8 xlo = __builtin_bit_cast(half8, uint4{pl[0], pl[1], pl[2], pl[3]});
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[ky], xhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[ky], xlo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[ky], xhi, acc, 0, 0, 0);
        }
        // lane holds channels 8*qq + 4*g + r of tile pixel p; pixels outside the stem's map are the depthwise conv's zeros
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool inside = sy >= 0 && sy < STEM_HW && sx >= 0 && sx < STEM_HW;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f32_then_f16(swish_f<false>(acc[4 * qq + r] + bv[qq][r]));
            if (!inside) o = half4{half_t(0.f), half_t(0.f), half_t(0.f), half_t(0.f)};     // (two selects, no branches)
            *reinterpret_cast<half4*>(s_tile + p * SD_C + 8 * qq + 4 * g) = o;
#ifdef WHENET_STEMDW_DEBUG          // (tools/probes/stemdw_probe.hip: the stem values of the tile, to be compared with stem.hip's)
            if (inside && g_stemdw_dbg)
                *reinterpret_cast<half4*>(g_stemdw_dbg + ((size_t(b) * STEM_HW + sy) * STEM_HW + sx) * SD_C + 8 * qq + 4 * g) = o;
#endif
        }
    }
    __syncthreads();
    STAMP(3);

    // ---- depthwise 3x3 on the tile (dw.hip's compute: lane = 4-channel group cg x strip sidx) -------------------------------
    using VCT = half_t __attribute__((ext_vector_type(4)));
    const int cg = tid % SD_CG;
    const int sidx = tid / SD_CG;                                             // 0..31 = TH * NSX
    const int ty = sidx / SD_NSX, sx = sidx - ty * SD_NSX;
    const int oy = oy0 + ty;
    float acc[SD_P][4];
#pragma unroll
    for (int p = 0; p < SD_P; ++p)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[p][v] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float wr[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float4v t = *reinterpret_cast<const float4v*>(s_w + (ky * 3 + kx) * SD_C + cg * 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) wr[kx][v] = t[v];
        }
        const half_t* row = s_tile + size_t((ty + ky) * SD_IW + sx * SD_P) * SD_C + cg * 4;
#pragma unroll
        for (int ix = 0; ix < SD_P + 2; ++ix) {
            const VCT xv = *reinterpret_cast<const VCT*>(row + ix * SD_C);
            float x[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) x[v] = float(xv[v]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int d = ix - kx;
                if (d >= 0 && d < SD_P) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[d][v] = fmaf(x[v], wr[kx][v], acc[d][v]);
                }
            }
        }
    }
    STAMP(4);
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float4v bs = *reinterpret_cast<const float4v*>(bd + cg * 4);
        half_t* dst = out + ((size_t(b) * STEM_HW + oy) * STEM_HW + (ox0 + sx * SD_P)) * SD_C + cg * 4;
#pragma unroll
        for (int p = 0; p < SD_P; ++p) {
            VCT o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float y = opaque_f32(conv_swish<half_t>(acc[p][v] + bs[v]));
                sum[v] += y;
                o[v] = half_t(y);
            }
            *reinterpret_cast<VCT*>(dst + size_t(p) * SD_C) = o;
        }
    }
    STAMP(5);
    lds_barrier();                               // everyone is done reading s_tile
#pragma unroll
    for (int v = 0; v < 4; ++v) s_red[sidx * SD_C + cg * 4 + v] = sum[v];
    lds_barrier();
    if (tid < SD_C) {
        float t = 0.0f;
        for (int s = 0; s < SD_TH * SD_NSX; ++s) t += s_red[s * SD_C + tid];
        partial[(size_t(b) * (SD_TILES_X * SD_TILES_Y) + tile) * SD_C + tid] = t;
    }
    STAMP(6);
}

// ---- f32 (the parity configuration) ---------------------------------------------------------------------------------------
// The same kernel with stem.hip's f32 matrix-core stem (v_mfma_f32_32x32x2_f32: 15 per strip, an fmaf chain in k order) and
// dw.hip's f32 instantiation; the tile is f32 (36 KB: the LUT, dead once the patch is staged, lives in its first 3 KB, which keeps
// the workgroup at 52 KB -- three per CU).  As two kernels the stem output costs 103 MB written + 133 MB read per 64 crops.
__global__ __launch_bounds__(256) void whenet_stemdw_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                const float* __restrict__ lut, const float* __restrict__ wd,
                                                                const float* __restrict__ bd, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float s_img[SD_PR * SD_ROWW + 8];
    __shared__ __attribute__((aligned(16))) float s_tile[SD_NPIX * SD_C];     // aliased by s_lut (before) and s_red (after)
    __shared__ __attribute__((aligned(16))) float s_w[9 * SD_C];
    float* s_lut = s_tile;
    float* s_red = s_tile;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int tile = blockIdx.x;
    const int tyi = tile / SD_TILES_X, txi = tile - tyi * SD_TILES_X;
    const int b = blockIdx.y;
    const int oy0 = tyi * SD_TH, ox0 = txi * SD_NSX * SD_P;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;

    constexpr int NLD = (SD_PR * SD_RDW + 255) / 256;
    const uint8_t* img = in + size_t(b) * IMG * IMG * 3;
    const int byte0 = 6 * sx0 - 2;
    uint32_t raw[NLD];
    bool rok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        const int r = d / SD_RDW, j = d - r * SD_RDW;
        const int iy = 2 * sy0 + r;
        const int bo = byte0 + 4 * j;
        rok[i] = d < SD_PR * SD_RDW && iy >= 0 && iy < IMG && bo >= 0 && bo < IMG * 3;
        raw[i] = rok[i] ? *reinterpret_cast<const uint32_t*>(img + size_t(iy) * (IMG * 3) + bo) : 0u;
    }
    float lutv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) lutv[i] = lut[tid + 256 * i];
    // weight operands: MFMA u of row ky contracts k = 2 u + g of the row's 9 values (k = 9: zero) -- as stem.hip
    float wv[3][5];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int j = 2 * u + g;
            wv[ky][u] = (j < 9) ? w[(ky * 9 + j) * STEM_C + lm] : 0.0f;
        }
    float4v bv[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) bv[qq] = *reinterpret_cast<const float4v*>(bias + 8 * qq + 4 * g);
    const float wdv0 = wd[tid], wdv1 = (tid < 9 * SD_C - 256) ? wd[256 + tid] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) s_lut[tid + 256 * i] = lutv[i];
    s_w[tid] = wdv0;
    if (tid < 9 * SD_C - 256) s_w[256 + tid] = wdv1;
    if (tid < 8) s_img[SD_PR * SD_ROWW + tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        if (d >= SD_PR * SD_RDW) continue;
        const int j = d % SD_RDW;
        int ch = (j + 1) % 3;
        float4v val;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = s_lut[ch * 256 + ((raw[i] >> (8 * q)) & 0xff)];
            val[q] = rok[i] ? t : 0.f;
            ch = (ch == 2) ? 0 : ch + 1;
        }
        *reinterpret_cast<float4v*>(&s_img[4 * d]) = val;
    }
    __syncthreads();                              // (also: the LUT is dead, the tile may be written)

    for (int strip = wave; strip < SD_NSTRIP; strip += 4) {
        const int p = strip * 32 + lm;
        const int pr = p / SD_IW, pc = p - pr * SD_IW;
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = &s_img[(2 * pr + ky) * SD_ROWW + 2 + pc * 6 + g];
#pragma unroll
            for (int u = 0; u < 5; ++u)                        // (u = 4, g = 1 reads the next pixel's value against a zero weight)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ky][u], row[2 * u], acc, 0, 0, 0);
        }
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool inside = sy >= 0 && sy < STEM_HW && sx >= 0 && sx < STEM_HW;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float4v o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = opaque_f32(conv_swish<float>(acc[4 * qq + r] + bv[qq][r]));
            if (!inside) o = float4v{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<float4v*>(s_tile + p * SD_C + 8 * qq + 4 * g) = o;
        }
    }
    __syncthreads();

    const int cg = tid % SD_CG;
    const int sidx = tid / SD_CG;
    const int ty = sidx / SD_NSX, sx = sidx - ty * SD_NSX;
    const int oy = oy0 + ty;
    float acc[SD_P][4];
#pragma unroll
    for (int p = 0; p < SD_P; ++p)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[p][v] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float wr[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float4v t = *reinterpret_cast<const float4v*>(s_w + (ky * 3 + kx) * SD_C + cg * 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) wr[kx][v] = t[v];
        }
        const float* row = s_tile + size_t((ty + ky) * SD_IW + sx * SD_P) * SD_C + cg * 4;
#pragma unroll
        for (int ix = 0; ix < SD_P + 2; ++ix) {
            const float4v xv = *reinterpret_cast<const float4v*>(row + ix * SD_C);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int d = ix - kx;
                if (d >= 0 && d < SD_P) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[d][v] = fmaf(xv[v], wr[kx][v], acc[d][v]);
                }
            }
        }
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float4v bs = *reinterpret_cast<const float4v*>(bd + cg * 4);
        float* dst = out + ((size_t(b) * STEM_HW + oy) * STEM_HW + (ox0 + sx * SD_P)) * SD_C + cg * 4;
#pragma unroll
        for (int p = 0; p < SD_P; ++p) {
            float4v o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float y = opaque_f32(conv_swish<float>(acc[p][v] + bs[v]));
                sum[v] += y;
                o[v] = y;
            }
            *reinterpret_cast<float4v*>(dst + size_t(p) * SD_C) = o;
        }
    }
    lds_barrier();
#pragma unroll
    for (int v = 0; v < 4; ++v) s_red[sidx * SD_C + cg * 4 + v] = sum[v];
    lds_barrier();
    if (tid < SD_C) {
        float t = 0.0f;
        for (int s = 0; s < SD_TH * SD_NSX; ++s) t += s_red[s * SD_C + tid];
        partial[(size_t(b) * (SD_TILES_X * SD_TILES_Y) + tile) * SD_C + tid] = t;
    }
}

}  // namespace

// the fused kernel is dw.hip's block-1 tile: only that plan puts the channel sums where se.hip reads them
bool stemdw_supported(int dtype, const DwPlan& p, int k, int s, int H, int C) {
    const int cv = dtype == WHENET_F16 ? 4 : 8;          // 16-byte vectors per pixel: all 32 channels in one chunk
    return (dtype == WHENET_F16 || dtype == WHENET_F32) && k == 3 && s == 1 && H == STEM_HW && C == SD_C && p.threads == 256 &&
           p.CV == cv && p.TH == SD_TH && p.NSX == SD_NSX && p.tiles_x == SD_TILES_X && p.tiles_y == SD_TILES_Y && p.chunks == 1;
}

void build_stemdw_table(const float* w, const float* lut, StemDwTable* out) {
    // (host: _Float16 conversions round to nearest even, as v_cvt_f16_f32 does; the subtraction is exact in f32)
    for (int i = 0; i < 3 * 256; ++i) {
        const half_t hi = half_t(lut[i]), lo = half_t(lut[i] - float(hi));
        out->lut[i] = uint32_t(__builtin_bit_cast(unsigned short, hi)) | (uint32_t(__builtin_bit_cast(unsigned short, lo)) << 16);
    }
    for (int ky = 0; ky < 3; ++ky)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int j = 8 * (lane >> 5) + e;                            // (kx, ci) of the kernel row; 9.. are zero weights
                const float v = (j < 9) ? w[(ky * 9 + j) * STEM_C + (lane & 31)] : 0.0f;
                const half_t hi = half_t(v);
                out->whi[ky][lane][e] = hi;
                out->wlo[ky][lane][e] = half_t(v - float(hi));
            }
}

void launch_stemdw(const StemDwArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.in && a.out && a.bias && a.wd && a.bd && a.partial && a.n >= 1, WHENET_EINVAL, "stemdw: missing argument");
    if (a.dtype == WHENET_F16) {
        WHENET_REQUIRE(a.tab != nullptr, WHENET_EINVAL, "stemdw: the f16 kernel takes the host-built table");
        hipLaunchKernelGGL(whenet_stemdw_kernel, dim3(SD_TILES_X * SD_TILES_Y, a.n), dim3(256), 0, stream, a.in,
                           static_cast<half_t*>(a.out), a.tab, a.bias, a.wd, a.bd, a.partial);
    } else {
        WHENET_REQUIRE(a.dtype == WHENET_F32 && a.w && a.lut, WHENET_EINVAL, "stemdw: the f32 kernel takes the f32 weights and LUT");
        hipLaunchKernelGGL(whenet_stemdw_f32_kernel, dim3(SD_TILES_X * SD_TILES_Y, a.n), dim3(256), 0, stream, a.in,
                           static_cast<float*>(a.out), a.w, a.bias, a.lut, a.wd, a.bd, a.partial);
    }
    WHENET_HIP_CHECK(hipGetLastError());
}

const char* kernel_name_stemdw(int dtype) { return dtype == WHENET_F16 ? "whenet_stemdw_kernel" : "whenet_stemdw_f32_kernel"; }

}  // namespace whenet
