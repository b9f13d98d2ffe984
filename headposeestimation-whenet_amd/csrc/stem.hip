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
        for (int i = 0; i < V; ++i) y[i] = swish_f<IsF32<T>::value>(acc[g * V + i] + bias[g * V + i]);
        dst[g] = float_to_vec<T>(y);
    }
}

}  // namespace

void launch_stem(const StemArgs& a, int dtype, hipStream_t stream) {
    dim3 grid(STEM_HW / ROWS_PER_BLOCK, a.n);
    if (a.in_f32 != nullptr) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in_f32);
        if (dtype == WHENET_F16)
            hipLaunchKernelGGL((whenet_stem_kernel<half_t, true>), grid, dim3(256), 0, stream, src,
                               static_cast<half_t*>(a.out), a.w, a.bias, a.lut);
        else
            hipLaunchKernelGGL((whenet_stem_kernel<float, true>), grid, dim3(256), 0, stream, src,
                               static_cast<float*>(a.out), a.w, a.bias, a.lut);
    } else if (dtype == WHENET_F16)
        hipLaunchKernelGGL((whenet_stem_kernel<half_t, false>), grid, dim3(256), 0, stream, a.in,
                           static_cast<half_t*>(a.out), a.w, a.bias, a.lut);
    else
        hipLaunchKernelGGL((whenet_stem_kernel<float, false>), grid, dim3(256), 0, stream, a.in,
                           static_cast<float*>(a.out), a.w, a.bias, a.lut);
    WHENET_HIP_CHECK(hipGetLastError());
}

const char* kernel_name_stem(int dtype) {
    return dtype == WHENET_F16 ? "whenet_stem_kernel<_Float16, false>" : "whenet_stem_kernel<float, false>";
}

}  // namespace whenet
