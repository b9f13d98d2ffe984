// f32 <-> activation-type conversions used only by the single-stage entry points
// (whenet_op_*), which take and return float32 so the tests can feed oracle tensors.
#include "device_math.h"
#include "kernels.h"

namespace whenet {

namespace {

template <typename T>
__global__ __launch_bounds__(256) void whenet_f32_to_act_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                size_t count) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256) dst[i] = T(src[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void whenet_act_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                                size_t count) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
        dst[i] = float(src[i]);
}

__global__ void whenet_empty_kernel() {}

inline unsigned grid_for(size_t count) {
    size_t g = (count + 255) / 256;
    return unsigned(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace

void launch_empty(hipStream_t stream) {
    hipLaunchKernelGGL(whenet_empty_kernel, dim3(1), dim3(64), 0, stream);
    WHENET_HIP_CHECK(hipGetLastError());
}

void launch_f32_to_act(const float* src, void* dst, size_t count, int dtype, hipStream_t stream) {
    if (dtype == WHENET_F16)
        hipLaunchKernelGGL(whenet_f32_to_act_kernel<half_t>, dim3(grid_for(count)), dim3(256), 0, stream, src,
                           static_cast<half_t*>(dst), count);
    else
        hipLaunchKernelGGL(whenet_f32_to_act_kernel<float>, dim3(grid_for(count)), dim3(256), 0, stream, src,
                           static_cast<float*>(dst), count);
    WHENET_HIP_CHECK(hipGetLastError());
}

void launch_act_to_f32(const void* src, float* dst, size_t count, int dtype, hipStream_t stream) {
    if (dtype == WHENET_F16)
        hipLaunchKernelGGL(whenet_act_to_f32_kernel<half_t>, dim3(grid_for(count)), dim3(256), 0, stream,
                           static_cast<const half_t*>(src), dst, count);
    else
        hipLaunchKernelGGL(whenet_act_to_f32_kernel<float>, dim3(grid_for(count)), dim3(256), 0, stream,
                           static_cast<const float*>(src), dst, count);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
