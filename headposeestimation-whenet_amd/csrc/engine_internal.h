// Definitions shared by the engine's translation units (engine.cpp: construction, options, arena, launch schedule,
// graphs, forward paths, profiling; engine_post.cpp: frame pre-processing and detector post-processing entry points;
// engine_ops.cpp: single-stage entry points for the tests).  Not part of any interface.
#pragma once

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace whenet {
namespace detail {

constexpr size_t X_ELEMS = size_t(112) * 112 * 32;    // largest block input/output per crop (stem out)
constexpr size_t E_ELEMS = size_t(112) * 112 * 96;    // largest expanded tensor per crop (b2 expand)
constexpr size_t D_ELEMS = size_t(56) * 56 * 144;     // largest depthwise output per crop (b3 dw)
constexpr size_t HC_ELEMS = size_t(49) * FEAT;        // head conv output per crop
constexpr size_t IN_BYTES = size_t(IMG) * IMG * 3;
constexpr int MAX_GRAPHS = 16;

struct DeviceGuard {
    explicit DeviceGuard(int dev) { WHENET_HIP_CHECK(hipSetDevice(dev)); }
};

struct TempBufs {     // hipMalloc'd scratch of the single-stage entry points
    std::vector<void*> ptrs;
    void* get(size_t nbytes) {
        void* p = nullptr;
        WHENET_HIP_CHECK(hipMalloc(&p, nbytes ? nbytes : 16));
        ptrs.push_back(p);
        return p;
    }
    ~TempBufs() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

inline void copy_name(char* dst, size_t cap, const std::string& s) {
    std::memset(dst, 0, cap);
    std::memcpy(dst, s.data(), std::min(cap - 1, s.size()));
}

}  // namespace detail
}  // namespace whenet
