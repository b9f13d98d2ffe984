// Stem: uint8 RGB crop -> normalise -> Conv2D(32, 3x3, stride 2, 'same', no bias) -> BN -> Swish.
//
// Reference: /root/reference/whenet.py:23-26 (float64 normalise, cast to float32 by Keras)
// feeding efficientnet 0.0.4's stem (whenet.py:8).  TF 'SAME' with
// stride 2 on the even 224 input pads ONLY bottom/right (pad 0 before, 1 after), and the
// padding is zero in the *normalised* domain.
//
// Mapping (gfx950): one workgroup = one crop x two output rows.  The five input rows it needs
// (672 B each) are read once as coalesced dwords, pushed through the 3x256 normalisation LUT
// (bit-exact image of the reference's float64 arithmetic) and parked in LDS as float; each of
// 224 lanes then owns one output pixel and all 32 output channels.  Weights are indexed
// uniformly across the wave, so they arrive through the scalar cache (s_load) and feed the
// FMAs as SGPR operands: no LDS or VGPR traffic for the 864 weights.
// HBM bytes per crop: 150,528 in (u8) + 401,408 * sizeof(T) out; 10.8 M MACs.
#include "device_math.h"
#include "kernels.h"

#include <cstdlib>

namespace whenet {

namespace {

constexpr int ROWS_PER_BLOCK = 2;
constexpr int IN_ROWS = 2 * ROWS_PER_BLOCK + 1;      // 5
constexpr int ROW_FLOATS = 225 * 3 + 1;              // 675 (+1 pad): column 224 is the zero pad
constexpr int ROW_DWORDS = IMG * 3 / 4;              // 168

// INF32 = true: the input is the already NORMALISED float32 image [n,224,224,3] that the reference
// hands to Model.predict (whenet.py:27) -- the path for real-valued crops, which have no byte LUT.
template <typename T, bool INF32>
__global__ __launch_bounds__(256) void whenet_stem_kernel(const uint8_t* __restrict__ in, T* __restrict__ out,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ lut) {
    __shared__ float s_lut[3 * 256];
    __shared__ float s_img[IN_ROWS * ROW_FLOATS];

    const int tid = threadIdx.x;
    const int oy0 = blockIdx.x * ROWS_PER_BLOCK;
    const int b = blockIdx.y;

    if constexpr (!INF32)
        for (int i = tid; i < 3 * 256; i += 256) s_lut[i] = lut[i];
    // zero pad column (x = 224) of every staged row
    if (tid < IN_ROWS * 4) s_img[(tid >> 2) * ROW_FLOATS + 672 + (tid & 3)] = 0.0f;
    __syncthreads();

    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in + size_t(b) * IMG * IMG * 3);
    const float4v* inf = reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(in) + size_t(b) * IMG * IMG * 3);
    for (int d = tid; d < IN_ROWS * ROW_DWORDS; d += 256) {
        const int r = d / ROW_DWORDS;
        const int j = d - r * ROW_DWORDS;
        const int iy = 2 * oy0 + r;
        float* dst = &s_img[r * ROW_FLOATS + 4 * j];
        if (INF32 && iy < IMG) {
            const float4v v = inf[iy * ROW_DWORDS + j];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        } else if (iy < IMG) {
            const uint32_t v = in32[iy * ROW_DWORDS + j];
            int ch = (4 * j) % 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dst[i] = s_lut[ch * 256 + ((v >> (8 * i)) & 0xff)];
                ch = (ch == 2) ? 0 : ch + 1;
            }
        } else {          // bottom pad row (iy == 224): zero in the normalised domain
            dst[0] = dst[1] = dst[2] = dst[3] = 0.0f;
        }
    }
    __syncthreads();

    if (tid >= ROWS_PER_BLOCK * STEM_HW) return;
    const int oyl = tid / STEM_HW;
    const int ox = tid - oyl * STEM_HW;

    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const float* row = &s_img[(2 * oyl + ky) * ROW_FLOATS + ox * 6];
#pragma unroll
        for (int i = 0; i < 9; ++i) x[ky * 9 + i] = row[i];      // (kx, ci) contiguous
    }

    float acc[STEM_C];
#pragma unroll
    for (int co = 0; co < STEM_C; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
        for (int co = 0; co < STEM_C; ++co) acc[co] = fmaf(x[tap], w[tap * STEM_C + co], acc[co]);
    }

    constexpr int V = Vec<T>::V;
    using VT = typename Vec<T>::type;
    VT* dst = reinterpret_cast<VT*>(out + ((size_t(b) * STEM_HW + (oy0 + oyl)) * STEM_HW + ox) * STEM_C);
#pragma unroll
    for (int g = 0; g < STEM_C / V; ++g) {
        float y[V];
#pragma unroll
        for (int i = 0; i < V; ++i) y[i] = conv_swish<T>(acc[g * V + i] + bias[g * V + i]);
        dst[g] = float_to_vec<T>(y);
    }
}

// ---- f16 configuration: the stem conv on the matrix cores ---------------------------------------
// The scalar form above spends 864 FMAs per output pixel in the VALU (10.8 M per crop: the kernel was VALU-bound
// at 3x its Swish floor).  Here the conv is a K = 27 contraction per pixel: k-step ky holds the 9 contiguous
// (kx, ci) values of input row 2*oy + ky (padded to 16), weights are the MFMA A operand (32 out-channels),
// 32 output pixels the B operand.  Inputs and weights are split x = hi + lo in binary16 and three products
// (hi*hi, hi*lo, lo*hi) are accumulated in f32, so the result keeps ~22 bits: no accuracy is traded for the
// matrix cores; the LUT and the staged rows hold the pair packed in one dword, so building a fragment is two
// permutes per two pixels' values instead of conversions (the f16 rounding still happens once, at the output, as before).  9 MFMAs per 32-pixel strip.
// Workgroup = crop x 4 output rows (14 strips over 4 waves); outputs are transposed through LDS so that every
// lane stores 16 contiguous bytes of NHWC.
constexpr int MR = 4;
constexpr int MIN_ROWS = 2 * MR + 1;                 // 9
constexpr int OPITCH = 40;                           // halfs per staged output pixel (80 B: 16-byte aligned)

template <bool INF32>
__global__ __launch_bounds__(256) void whenet_stem_mfma_kernel(const uint8_t* __restrict__ in, half_t* __restrict__ out,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ lut) {
    __shared__ uint32_t s_lut[3 * 256];          // the normalisation LUT as packed binary16 (hi | lo << 16): v = hi + lo
    __shared__ __attribute__((aligned(16))) uint32_t s_img[MIN_ROWS * ROW_FLOATS];     // staged rows, packed the same way
    __shared__ __attribute__((aligned(16))) half_t s_out[4][32 * OPITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int oy0 = blockIdx.x * MR;
    const int b = blockIdx.y;

    // weight fragments (independent of the image: in flight while the rows are staged)
    float wv[3][8];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * g + e;
            wv[ky][e] = (j < 9) ? w[(ky * 9 + j) * STEM_C + lm] : 0.0f;
        }
    float4v bv[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) bv[qq] = *reinterpret_cast<const float4v*>(bias + 8 * qq + 4 * g);

    // the image rows are requested before anything is waited for: ONE global round trip per workgroup (the
    // LUT, the weights and the rows travel together)
    constexpr int NLD = (MIN_ROWS * ROW_DWORDS + 255) / 256;          // 6 dwords (or float4s) per lane
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in + size_t(b) * IMG * IMG * 3);
    const float4v* inf = reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(in) + size_t(b) * IMG * IMG * 3);
    uint32_t raw[INF32 ? 1 : NLD];
    float4v rawf[INF32 ? NLD : 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        const int r = d / ROW_DWORDS, j = d - r * ROW_DWORDS;
        const int iy = 2 * oy0 + r;
        const bool ok = d < MIN_ROWS * ROW_DWORDS && iy < IMG;
        if constexpr (INF32) rawf[i] = ok ? inf[iy * ROW_DWORDS + j] : float4v{0.f, 0.f, 0.f, 0.f};
        else raw[i] = ok ? in32[iy * ROW_DWORDS + j] : 0u;
    }
    auto pack = [](float v) -> uint32_t {
        // hi + lo in binary16 (22 bits kept).  Real-valued input beyond the binary16 range saturates at +-65504
        // (a normalised pixel of that size is ~3.7 million grey levels): without the clamp hi = inf, lo = nan
        if constexpr (INF32) v = fminf(fmaxf(v, -65504.0f), 65504.0f);
        const half_t hi = half_t(v), lo = half_t(v - float(hi));
        return uint32_t(__builtin_bit_cast(unsigned short, hi)) | (uint32_t(__builtin_bit_cast(unsigned short, lo)) << 16);
    };
    if constexpr (!INF32)
        for (int i = tid; i < 3 * 256; i += 256) s_lut[i] = pack(lut[i]);
    if (tid < MIN_ROWS * 4) s_img[(tid >> 2) * ROW_FLOATS + 672 + (tid & 3)] = 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        if (d >= MIN_ROWS * ROW_DWORDS) continue;
        const int r = d / ROW_DWORDS, j = d - r * ROW_DWORDS;
        const int iy = 2 * oy0 + r;
        uint32_t* dst = &s_img[r * ROW_FLOATS + 4 * j];
        if (INF32) {
            dst[0] = pack(rawf[i][0]); dst[1] = pack(rawf[i][1]); dst[2] = pack(rawf[i][2]); dst[3] = pack(rawf[i][3]);
        } else if (iy < IMG) {
            const uint32_t v = raw[i];
            int ch = (4 * j) % 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dst[q] = s_lut[ch * 256 + ((v >> (8 * q)) & 0xff)];
                ch = (ch == 2) ? 0 : ch + 1;
            }
        } else {          // bottom pad row (iy == 224): zero in the normalised domain
            dst[0] = dst[1] = dst[2] = dst[3] = 0u;
        }
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

    constexpr int NSTRIP = MR * STEM_HW / 32;        // 14
    half_t* so = s_out[wave];
    half_t* obase = out + (size_t(b) * STEM_HW + oy0) * STEM_HW * STEM_C;
    for (int strip = wave; strip < NSTRIP; strip += 4) {
        const int p = strip * 32 + lm;
        const int oyl = p / STEM_HW, ox = p - oyl * STEM_HW;
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // 8 packed (hi | lo << 16) values of this lane's k-group -> the hi and the lo fragment (2 permutes per pair)
            const uint32_t* row = &s_img[(2 * oyl + ky) * ROW_FLOATS + ox * 6 + 8 * g];
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
        // BN bias + Swish, one rounding to f16; lane holds channels 8*qq + 4*g + r of pixel lm
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f32_then_f16(swish_f<false>(acc[4 * qq + r] + bv[qq][r]));
            *reinterpret_cast<half4*>(so + lm * OPITCH + 8 * qq + 4 * g) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's own LDS region: no barrier needed)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = lane + 64 * i;
            const int px = idx >> 2, part = idx & 3;
            const half8 v = *reinterpret_cast<const half8*>(so + px * OPITCH + part * 8);
            *reinterpret_cast<half8*>(obase + size_t(strip * 32 + px) * STEM_C + part * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next strip overwrites
    }
}

// ---- f32 configuration on the matrix cores (round 4) -----------------------------------------------------------------------
// The same decomposition with v_mfma_f32_32x32x2_f32 -- exact f32, an fmaf chain in k order, so the result has the bits of
// the scalar kernel above (k = (ky, kx, ci) ascending): the 9 contiguous (kx, ci) values of a row are 5 k-pairs (the tenth
// value is a zero weight), 15 MFMAs per 32-pixel strip instead of 864 FMAs per pixel in the VALU.  Rows and LUT stay f32.
constexpr int OPITCH32 = 36;                         // floats per staged output pixel (144 B: 16-byte aligned)

template <bool INF32>
__global__ __launch_bounds__(256) void whenet_stem_mfma_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ lut) {
    __shared__ float s_lut[3 * 256];
    __shared__ __attribute__((aligned(16))) float s_img[MIN_ROWS * ROW_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_out[4][32 * OPITCH32];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, lm = lane & 31;
    const int oy0 = blockIdx.x * MR;
    const int b = blockIdx.y;

    // weight operands: MFMA u of row ky contracts k = 2 u + g of the row's 9 values (k = 9: zero)
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

    constexpr int NLD = (MIN_ROWS * ROW_DWORDS + 255) / 256;
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in + size_t(b) * IMG * IMG * 3);
    const float4v* inf = reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(in) + size_t(b) * IMG * IMG * 3);
    uint32_t raw[INF32 ? 1 : NLD];
    float4v rawf[INF32 ? NLD : 1];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        const int r = d / ROW_DWORDS, j = d - r * ROW_DWORDS;
        const int iy = 2 * oy0 + r;
        const bool ok = d < MIN_ROWS * ROW_DWORDS && iy < IMG;
        if constexpr (INF32) rawf[i] = ok ? inf[iy * ROW_DWORDS + j] : float4v{0.f, 0.f, 0.f, 0.f};
        else raw[i] = ok ? in32[iy * ROW_DWORDS + j] : 0u;
    }
    if constexpr (!INF32)
        for (int i = tid; i < 3 * 256; i += 256) s_lut[i] = lut[i];
    if (tid < MIN_ROWS * 4) s_img[(tid >> 2) * ROW_FLOATS + 672 + (tid & 3)] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = tid + 256 * i;
        if (d >= MIN_ROWS * ROW_DWORDS) continue;
        const int r = d / ROW_DWORDS, j = d - r * ROW_DWORDS;
        const int iy = 2 * oy0 + r;
        float* dst = &s_img[r * ROW_FLOATS + 4 * j];
        if (INF32) {
            dst[0] = rawf[i][0]; dst[1] = rawf[i][1]; dst[2] = rawf[i][2]; dst[3] = rawf[i][3];     // (zero beyond the image)
        } else if (iy < IMG) {
            const uint32_t v = raw[i];
            int ch = (4 * j) % 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dst[q] = s_lut[ch * 256 + ((v >> (8 * q)) & 0xff)];
                ch = (ch == 2) ? 0 : ch + 1;
            }
        } else {          // bottom pad row (iy == 224): zero in the normalised domain
            dst[0] = dst[1] = dst[2] = dst[3] = 0.0f;
        }
    }
    __syncthreads();

    constexpr int NSTRIP = MR * STEM_HW / 32;        // 14
    float* so = s_out[wave];
    float* obase = out + (size_t(b) * STEM_HW + oy0) * STEM_HW * STEM_C;
    for (int strip = wave; strip < NSTRIP; strip += 4) {
        const int p = strip * 32 + lm;
        const int oyl = p / STEM_HW, ox = p - oyl * STEM_HW;
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = &s_img[(2 * oyl + ky) * ROW_FLOATS + ox * 6 + g];
#pragma unroll
            for (int u = 0; u < 5; ++u)                        // (u = 4, g = 1 reads the next pixel's value against a zero weight)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ky][u], row[2 * u], acc, 0, 0, 0);
        }
        // BN bias + Swish; lane holds channels 8*qq + 4*g + r of pixel lm
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float4v o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = conv_swish<float>(acc[4 * qq + r] + bv[qq][r]);
            *reinterpret_cast<float4v*>(so + lm * OPITCH32 + 8 * qq + 4 * g) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's own LDS region: no barrier needed)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            const int px = idx >> 3, part = idx & 7;
            const float4v v = *reinterpret_cast<const float4v*>(so + px * OPITCH32 + part * 4);
            *reinterpret_cast<float4v*>(obase + size_t(strip * 32 + px) * STEM_C + part * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next strip overwrites
    }
}

}  // namespace

// WHENET_STEM_SCALAR_F32=1 (read once): the f32 stem as round 3's scalar kernel -- the two forms are compared bitwise by
// tests/test_gpu_parity.py::test_f32_stem_on_the_matrix_cores_has_the_scalar_kernels_bits
static bool stem_scalar_f32() {
    static const bool v = [] { const char* e = getenv("WHENET_STEM_SCALAR_F32"); return e && e[0] == '1'; }();
    return v;
}

void launch_stem(const StemArgs& a, int dtype, hipStream_t stream) {
    dim3 grid(STEM_HW / ROWS_PER_BLOCK, a.n);
    dim3 grid_m(STEM_HW / MR, a.n);
    if (a.in_f32 != nullptr) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in_f32);
        if (dtype == WHENET_F16)
            hipLaunchKernelGGL((whenet_stem_mfma_kernel<true>), grid_m, dim3(256), 0, stream, src,
                               static_cast<half_t*>(a.out), a.w, a.bias, a.lut);
        else if (stem_scalar_f32())
            hipLaunchKernelGGL((whenet_stem_kernel<float, true>), grid, dim3(256), 0, stream, src,
                               static_cast<float*>(a.out), a.w, a.bias, a.lut);
        else
            hipLaunchKernelGGL((whenet_stem_mfma_f32_kernel<true>), grid_m, dim3(256), 0, stream, src,
                               static_cast<float*>(a.out), a.w, a.bias, a.lut);
    } else if (dtype == WHENET_F16)
        hipLaunchKernelGGL((whenet_stem_mfma_kernel<false>), grid_m, dim3(256), 0, stream, a.in,
                           static_cast<half_t*>(a.out), a.w, a.bias, a.lut);
    else if (stem_scalar_f32())
        hipLaunchKernelGGL((whenet_stem_kernel<float, false>), grid, dim3(256), 0, stream, a.in,
                           static_cast<float*>(a.out), a.w, a.bias, a.lut);
    else
        hipLaunchKernelGGL((whenet_stem_mfma_f32_kernel<false>), grid_m, dim3(256), 0, stream, a.in,
                           static_cast<float*>(a.out), a.w, a.bias, a.lut);
    WHENET_HIP_CHECK(hipGetLastError());
}

const char* kernel_name_stem(int dtype) {
    return dtype == WHENET_F16 ? "whenet_stem_mfma_kernel<false>"
                               : (stem_scalar_f32() ? "whenet_stem_kernel<float, false>" : "whenet_stem_mfma_f32_kernel<false>");
}

}  // namespace whenet
