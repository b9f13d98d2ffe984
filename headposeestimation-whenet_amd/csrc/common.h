// Shared definitions for libwhenet_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/whenet_hip.h"

namespace whenet {

// Error carrying a C-ABI return code; never crosses the extern "C" boundary (capi.cpp
// catches everything and turns it into the code + whenet_last_error()).
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define WHENET_HIP_CHECK(expr)                                                               \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            throw ::whenet::Error(WHENET_EHIP, std::string(#expr) + ": " +                   \
                                                   hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                                   std::to_string(__LINE__) + ")");          \
        }                                                                                    \
    } while (0)

#define WHENET_REQUIRE(cond, code, msg)                                  \
    do {                                                                 \
        if (!(cond)) throw ::whenet::Error((code), std::string(msg));    \
    } while (0)

using half_t = _Float16;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

// Number of activation elements in one 16-byte vector.
template <typename T> struct Vec;
template <> struct Vec<float>  { static constexpr int V = 4; using type = float4v; };
template <> struct Vec<half_t> { static constexpr int V = 8; using type = half8; };

constexpr int IMG = 224;
constexpr int STEM_C = 32;
constexpr int STEM_HW = 112;
constexpr int FEAT = 1280;
constexpr int N_YAW = 120, N_PITCH = 66, N_ROLL = 66, N_LOGITS = 252;
constexpr double BN_EPS = 1e-3;   // efficientnet 0.0.4 GlobalParams.batch_norm_epsilon

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace whenet
