// Host-side model preparation: WHNPACK1 snapshot -> BN-folded, kernel-ready tensors.
// Replaces keras Model.load_weights (/root/reference/whenet.py:15-16) plus everything TF
// does lazily at first Session.run (constant folding, layout choice).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "spec.h"

namespace whenet {

struct RawTensor {
    std::vector<uint32_t> dims;
    const float* data = nullptr;   // points into the snapshot blob
    size_t count = 0;
};

// name -> tensor view; validates magic/version/bounds and the full WHENet tensor census.
std::map<std::string, RawTensor> parse_snapshot(const void* blob, size_t nbytes);

// A 1x1 convolution with its BatchNorm folded in:  out[m][n] = sum_k a[m][k]*w[k][n] + bias[n].
// `packed` is the MFMA operand image (see pw.hip for the fragment order):
//   [ks][ntile][lane 0..63][V]   V = 4 (f32) / 8 (f16) elements = 16 B per lane,
//   element e of lane l  <->  n = ntile*32 + (l & 31),  k = ks*2V + (l >> 5)*V + e,
//   zero where n >= N or k >= K.
struct HostPw {
    int K = 0, N = 0, KS = 0, NTILES = 0;
    std::vector<uint8_t> packed;     // dtype-sized elements
    std::vector<float> bias;         // [N]
    std::vector<float> dense;        // [K][N] folded weights rounded to the activation dtype
                                     // (operand of the scalar check kernel, pw_impl=1)
    // WHENET_F32S (float storage, products on the f16 matrix pipe): w * 2^wshift = hi + lo in binary16, two images in the
    // f16 fragment order (k-step = 16: lane l <-> n = 32 nt + (l & 31), k = 16 ks + 8 (l >> 5) + e), [hi image | lo image]
    std::vector<uint8_t> packed_split;
    int KS_split = 0;
    float wsi = 1.0f;                // 2^-wshift: the accumulators are multiplied by it in the epilogue
};

struct HostDw {
    int k = 0, C = 0;
    std::vector<float> w;            // [k*k][C]  BN scale folded in
    std::vector<float> bias;         // [C]
};

struct HostSe {
    int C = 0, R = 0;
    std::vector<float> w1t;          // [R][C]  (se_reduce kernel transposed)
    std::vector<float> b1;           // [R]
    std::vector<float> w2;           // [R][C]  (se_expand kernel)
    std::vector<float> w2c;          // [C][RP] the same kernel channel-major, R zero-padded to a multiple of 4
    std::vector<float> b2;           // [C]
    HostPw excite;                   // se_expand as an MFMA operand image, K = R, N = C (f16 handles: packed; f32s: packed_split); empty otherwise
};

struct HostBlock {
    BlockSpec spec;
    HostPw expand;                   // unused when !spec.has_expand()
    HostDw dw;
    HostSe se;
    HostPw project;
};

// whenet.py:23-26 for every byte value and channel: float64 arithmetic, one rounding to float32 (Keras' cast, :27).
void normalise_table(float lut[3][256]);

struct HostModel {
    int dtype = WHENET_F32;
    float lut[3][256];               // whenet.py:23-26 for every byte value, per channel
    std::vector<float> stem_w;       // [27][32]  tap = (ky*3+kx)*3+ci, BN folded
    std::vector<float> stem_b;       // [32]
    std::vector<HostBlock> blocks;   // 16
    HostPw head;                     // 320 -> 1280
    HostPw fold12;                   // 32 -> 96: block 1's project composed with block 2's expand (snapshot.cpp)
    std::vector<float> fold12_w32;   // its weights in f32, f16 fragment order [2][3][64 lanes][8] (front2.hip)
    std::vector<float> dense_w;      // [1280][252]  yaw|pitch|roll  (whenet.py:11-13)
    std::vector<float> dense_b;      // [252]
    int64_t params_backbone = 0, params_heads = 0;
    int n_tensors = 0;
};

HostModel build_host_model(const std::map<std::string, RawTensor>& t, int dtype, bool split = false);

}  // namespace whenet
