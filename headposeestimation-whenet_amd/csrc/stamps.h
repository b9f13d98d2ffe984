// In-kernel timeline stamps for the probes under tools/probes (compiled with -DWHENET_STAMPS);
// the library build defines STAMP() as nothing.  One row of 8 wall-clock (100 MHz) stamps per
// workgroup, written by its lane 0.
#pragma once

#ifdef WHENET_STAMPS
namespace whenet {
__device__ long long* whenet_stamps = nullptr;
}
#define STAMP(i)                                                                                       \
    do {                                                                                               \
        if (threadIdx.x == 0 && ::whenet::whenet_stamps)                                               \
            ::whenet::whenet_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = \
                wall_clock64();                                                                        \
    } while (0)
#else
#define STAMP(i)
#endif
