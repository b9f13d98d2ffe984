// Stem + block 1's depthwise conv as ONE row-streaming kernel (f16 configuration, uint8 input).
//
// Reference: /root/reference/whenet.py:23-26 (normalise) -> efficientnet 0.0.4 stem Conv3x3/s2 'same' + BN + Swish
// -> MBConv block 1 (expand ratio 1: no expand conv) DepthwiseConv2D 3x3/s1 'same' + BN + Swish, and the channel
// sums of that output for the block's squeeze-excite mean (whenet.py:8).
//
// As two launches (stem.hip, dw.hip) the 112x112x32 stem output crosses HBM twice: 1.6 MB of the 2.6 MB those two
// kernels move per crop.  Here a workgroup owns a crop x band of 16 output rows and walks down the band two rows at
// a time; the stem rows live only in a 4-row ring in LDS:
//   A  the 5 input rows of the next two stem rows (requested one step ahead) go through the 3x256 normalisation
//      LUT into LDS as f32,
//   B  the stem conv of those two rows runs on the matrix cores exactly as in stem.hip's f16 kernel (K = 27 as three
//      16-deep k-steps, weights and pixels split hi + lo in binary16, f32 accumulation), BN + Swish, one rounding to
//      f16 into ring slot (row mod 4); rows outside the image are written as zeros -- TF 'SAME' padding of the
//      depthwise INPUT -- and the ring's first/last columns are zero for the same reason,
//   C  the depthwise taps of two output rows run out of the ring (lane = 4 channels x strip of 7 pixels, 256 lanes =
//      2 rows x 16 strips x 8 channel groups), BN + Swish, NHWC store, running channel sums.
// One band recomputes 2 of its 18 stem rows.  Arithmetic of both convs is that of stem.hip / dw.hip (same operand
// rounding, same summation order per output); only the grouping of the squeeze-excite partial sums follows this
// kernel's bands.  HBM bytes per crop: 150,528 in + 802,816 out.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

constexpr int BAND = 16;                         // output rows per workgroup
constexpr int NBANDS = STEM_HW / BAND;           // 7
constexpr int ROW_FLOATS = 225 * 3 + 1;          // staged input row (+ the zero pad column)
constexpr int ROW_DWORDS = IMG * 3 / 4;          // 168
constexpr int SIMG_ROWS = 5;
constexpr int SCOLS = STEM_HW + 2;               // ring columns: 1 zero pad each side
constexpr int SPITCH = 80;                       // bytes per ring pixel (32 halfs + 16: spreads the tap reads over banks)
constexpr int NLD = (SIMG_ROWS * ROW_DWORDS + 255) / 256;       // 4 dwords per lane per step

__global__ __launch_bounds__(256, 3) void whenet_stem_dw_kernel(const uint8_t* __restrict__ in, half_t* __restrict__ out,
                                                             float* __restrict__ partial,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ lut,
                                                             const float* __restrict__ wd,
                                                             const float* __restrict__ bd) {
    __shared__ uint32_t s_lut[3 * 256];            // normalisation LUT as packed binary16 (hi | lo << 16): v = hi + lo
    __shared__ __attribute__((aligned(16))) uint32_t s_img[SIMG_ROWS * ROW_FLOATS];
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[4 * SCOLS * SPITCH];
    __shared__ __attribute__((aligned(16))) float s_wd[10 * STEM_C + STEM_C];       // depthwise taps [9][32], bias [32], stem bias [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int band = blockIdx.x, b = blockIdx.y;
    const int R0 = band * BAND;
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in + size_t(b) * IMG * IMG * 3);

    // stem pair s = stem rows (R0 - 1 + 2s, R0 + 2s); its input rows are 2*(R0 - 1 + 2s) + 0..4
    auto issue_rows = [&](int s, uint32_t (&raw)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int d = tid + 256 * i;
            const int rr = d / ROW_DWORDS, j = d - rr * ROW_DWORDS;
            const int iy = 2 * (R0 - 1 + 2 * s) + rr;
            raw[i] = (d < SIMG_ROWS * ROW_DWORDS && iy >= 0 && iy < IMG) ? in32[iy * ROW_DWORDS + j] : 0u;
        }
    };
    auto stage_rows = [&](int s, const uint32_t (&raw)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int d = tid + 256 * i;
            if (d >= SIMG_ROWS * ROW_DWORDS) continue;
            const int rr = d / ROW_DWORDS, j = d - rr * ROW_DWORDS;
            const int iy = 2 * (R0 - 1 + 2 * s) + rr;
            uint32_t* dst = &s_img[rr * ROW_FLOATS + 4 * j];
            if (iy >= 0 && iy < IMG) {
                const uint32_t v = raw[i];
                int ch = (4 * j) % 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dst[q] = s_lut[ch * 256 + ((v >> (8 * q)) & 0xff)];
                    ch = (ch == 2) ? 0 : ch + 1;
                }
            } else {              // above the image (unused rows) or the bottom pad row: zero in the normalised domain
                dst[0] = dst[1] = dst[2] = dst[3] = 0u;
            }
        }
    };

    uint32_t raw[NLD];
    issue_rows(0, raw);

    // ---- operands that do not depend on the image -------------------------------------------------
    float wv[3][8];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * g + e;
            wv[ky][e] = (j < 9) ? w[(ky * 9 + j) * STEM_C + lm] : 0.0f;
        }
    const float bias_l = (tid < STEM_C) ? bias[tid] : 0.0f;
    // depthwise lane mapping: 8 channel groups x 16 strips x 2 rows
    const int cg = tid & 7, strip = (tid >> 3) & 15, rsel = tid >> 7;
    // (the 9 x 32 depthwise taps and the stem bias sit in LDS and are re-read per step: holding them in registers
    // costs a workgroup per CU)
    const float wd_l = (tid < 9 * STEM_C + STEM_C) ? (tid < 9 * STEM_C ? wd[tid] : bd[tid - 9 * STEM_C]) : 0.0f;   // 320 > 256: two passes
    const float wd_h = (tid + 256 < 9 * STEM_C + STEM_C) ? (tid + 256 < 9 * STEM_C ? wd[tid + 256] : bd[tid + 256 - 9 * STEM_C]) : 0.0f;

    s_wd[tid] = wd_l;
    if (tid + 256 < 10 * STEM_C) s_wd[tid + 256] = wd_h;
    if (tid < STEM_C) s_wd[10 * STEM_C + tid] = bias_l;
    for (int i = tid; i < 3 * 256; i += 256) {
        const float v = lut[i];
        const half_t hi = half_t(v), lo = half_t(v - float(hi));
        s_lut[i] = uint32_t(__builtin_bit_cast(unsigned short, hi)) | (uint32_t(__builtin_bit_cast(unsigned short, lo)) << 16);
    }
    if (tid < SIMG_ROWS * 4) s_img[(tid >> 2) * ROW_FLOATS + 672 + (tid & 3)] = 0u;
    // the ring's pad columns (0 and SCOLS-1) of every slot: zero once, never written again
    for (int i = tid; i < 4 * 2 * (SPITCH / 16); i += 256) {
        const int slot = i / (2 * (SPITCH / 16)), rest = i - slot * 2 * (SPITCH / 16);
        const int col = (rest / (SPITCH / 16)) ? (SCOLS - 1) : 0, part = rest % (SPITCH / 16);
        *reinterpret_cast<float4v*>(s_ring + (slot * SCOLS + col) * SPITCH + part * 16) = float4v{0.f, 0.f, 0.f, 0.f};
    }
    half8 whi[3], wlo[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            whi[ky][e] = half_t(wv[ky][e]);
            wlo[ky][e] = half_t(wv[ky][e] - float(whi[ky][e]));
        }
    __syncthreads();

    // ---- B: two stem rows -> ring ------------------------------------------------------------------
    auto stem_pair = [&](int s) {
        const int ra = R0 - 1 + 2 * s;
        for (int st = wave; st < 2 * STEM_HW / 32; st += 4) {          // 7 strips of 32 pixels
            const int p = st * 32 + lm;
            const int rs = (p >= STEM_HW) ? 1 : 0, ox = p - rs * STEM_HW;
            const int r = ra + rs;
            unsigned char* dst = s_ring + (((r & 3) * SCOLS) + ox + 1) * SPITCH;
            // rows outside the image are the 'SAME' padding of the depthwise INPUT: written as zeros below (no
            // divergence around the MFMAs: a strip can straddle a valid and a padding row)
            const bool in_image = r >= 0 && r < STEM_HW;
            float16v acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // 8 packed (hi | lo << 16) values of this lane's k-group -> the hi and the lo fragment (2 permutes per pair)
                const uint32_t* row = &s_img[(2 * rs + ky) * ROW_FLOATS + ox * 6 + 8 * g];
                uint32_t d[8];
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint2 v = *reinterpret_cast<const uint2*>(row + 2 * i);     // (8-byte aligned: 24*ox)
                        d[2 * i] = v.x;
                        d[2 * i + 1] = v.y;
                    }
                } else {
                    d[0] = row[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) d[i] = 0u;
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
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4v bvq = *reinterpret_cast<const float4v*>(s_wd + 10 * STEM_C + 8 * qq + 4 * g);
                half4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = in_image ? half_t(swish_f<false>(acc[4 * qq + k] + bvq[k])) : half_t(0);
                *reinterpret_cast<half4*>(dst + (8 * qq + 4 * g) * 2) = o;
            }
        }
    };

    // ---- C: two depthwise output rows out of the ring -------------------------------------------------
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    auto dw_pair = [&](int j) {
        const int r = R0 + 2 * j + rsel;
        float acc[7][4];
#pragma unroll
        for (int p = 0; p < 7; ++p)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[p][v] = 0.0f;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            float4v wdw[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wdw[kx] = *reinterpret_cast<const float4v*>(s_wd + (ky * 3 + kx) * STEM_C + cg * 4);
            const unsigned char* row = s_ring + ((((r - 1 + ky) & 3) * SCOLS) + 7 * strip) * SPITCH + cg * 8;
#pragma unroll
            for (int ix = 0; ix < 9; ++ix) {
                const half4 xv = *reinterpret_cast<const half4*>(row + ix * SPITCH);
                float xf[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) xf[v] = float(xv[v]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int d = ix - kx;
                    if (d >= 0 && d < 7) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[d][v] = fmaf(xf[v], wdw[kx][v], acc[d][v]);
                    }
                }
            }
        }
        const float4v bdw = *reinterpret_cast<const float4v*>(s_wd + 9 * STEM_C + cg * 4);
        half_t* dst = out + ((size_t(b) * STEM_HW + r) * STEM_HW + 7 * strip) * STEM_C + cg * 4;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            half4 o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float y = swish_f<false>(acc[p][v] + bdw[v]);
                sum[v] += y;
                o[v] = half_t(y);
            }
            *reinterpret_cast<half4*>(dst + size_t(p) * STEM_C) = o;
        }
    };

    // ---- the band ---------------------------------------------------------------------------------------
    stage_rows(0, raw);
    issue_rows(1, raw);
    lds_barrier();
    stem_pair(0);
    lds_barrier();
#pragma unroll 1
    for (int j = 0; j < BAND / 2; ++j) {
        stage_rows(j + 1, raw);
        if (j + 2 <= BAND / 2) issue_rows(j + 2, raw);
        lds_barrier();                     // rows staged; every wave is past C(j-1)
        stem_pair(j + 1);
        lds_barrier();                     // ring rows R0+2j-1 .. R0+2j+2 complete
        dw_pair(j);
    }

    // ---- channel sums of the band (fixed order) for block 1's squeeze-excite mean ----------------------
    float* s_red = reinterpret_cast<float*>(s_img);      // (the staged rows are dead: every wave is past the last stem pair)
#pragma unroll
    for (int v = 0; v < 4; ++v) s_red[(tid >> 3) * 32 + cg * 4 + v] = sum[v];
    lds_barrier();
    if (tid < 32) {
        float t = 0.0f;
        for (int s = 0; s < 32; ++s) t += s_red[s * 32 + tid];
        partial[(size_t(b) * NBANDS + band) * STEM_C + tid] = t;
    }
}

}  // namespace

int stem_dw_bands() { return NBANDS; }

void launch_stem_dw(const StemDwArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(whenet_stem_dw_kernel, dim3(NBANDS, a.n), dim3(256), 0, stream, a.in, static_cast<half_t*>(a.out),
                       a.partial, a.w, a.bias, a.lut, a.wd, a.bd);
    WHENET_HIP_CHECK(hipGetLastError());
}

const char* kernel_name_stem_dw() { return "whenet_stem_dw_kernel"; }

}  // namespace whenet
