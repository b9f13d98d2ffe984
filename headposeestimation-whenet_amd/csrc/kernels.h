// Launchers of the hand-written gfx950 kernels.  Every launcher enqueues exactly one kernel
// on `stream` and returns; `dtype` selects the float / _Float16 instantiation.
#pragma once

#include <string>
#include <vector>

#include "common.h"

namespace whenet {

// ---- stem.hip ---------------------------------------------------------------------------
// uint8 crops -> LUT normalise (whenet.py:23-26) -> Conv3x3/s2 'same' + BN + Swish.
struct StemArgs {
    const uint8_t* in;     // [n,224,224,3]
    void* out;             // [n,112,112,32] T
    const float* w;        // [27][32]
    const float* bias;     // [32]
    const float* lut;      // [3][256]
    int n;
    const float* in_f32 = nullptr;   // != NULL: the normalised float32 image [n,224,224,3] instead of `in`
};
void launch_stem(const StemArgs& a, int dtype, hipStream_t stream);

// ---- dw.hip -----------------------------------------------------------------------------
// Depthwise kxk conv (TF 'same') + BN + Swish, NHWC, LDS-staged halo tiles, plus the
// per-tile channel sums the squeeze-excite mean is built from.
struct DwPlan {
    int threads = 256;     // block size (128 for the stride-2 layers: 4x input footprint)
    int CV = 1;            // 16-byte channel vectors per block
    int TH = 1;            // output rows per block
    int NSX = 1;           // 7-pixel output strips per tile row  (tile width = 7*NSX)
    int tiles_x = 1, tiles_y = 1, chunks = 1;
    int IH = 0, IW = 0;    // input tile (with halo)
    size_t lds_bytes = 0;
    int ntiles() const { return tiles_x * tiles_y; }
};
DwPlan plan_dw(int dtype, int k, int s, int H, int Ho, int C);

struct DwArgs {
    const void* in;        // [n,H,H,C] T
    void* out;             // [n,Ho,Ho,C] T
    const float* w;        // [k*k][C]
    const float* bias;     // [C]
    float* partial;        // [n][ntiles][C]  sums of the tile's outputs (pre-rounding f32)
    int k, s, H, Ho, C, pad, n;
    DwPlan plan;
};
void launch_dw(const DwArgs& a, int dtype, hipStream_t stream);

// ---- se.hip -----------------------------------------------------------------------------
// SEBlock: mean over H,W -> Conv1x1+bias -> Swish -> Conv1x1+bias -> sigmoid.
struct SeArgs {
    const float* partial;  // [n][ntiles][C]
    int ntiles;
    float inv_hw;
    const float* w1t;      // [R][C]  se_reduce kernel, transposed
    const float* b1;       // [R]
    const float* w2c;      // [C][RP]  se_expand kernel, channel-major, R zero-padded to RP
    const float* b2;       // [C]
    void* gate;            // [n][C], float or (gate_f16) half: the type of the activations it multiplies
    int C, R, n;
    int gate_f16 = 0;
};
void launch_se(const SeArgs& a, hipStream_t stream);
// second half of the SEBlock when the producer (front.hip) already applied the reduce conv to its
// channel sums: r = swish(b1 + sum of the crop's np partial vectors / (H*W)) -> excite -> sigmoid
struct SeExciteArgs {
    const float* rpart;    // [n][np][RP]
    int np;
    float inv_hw;
    const float* b1;       // [R]
    const float* w2c;      // [C][RP]
    const float* b2;       // [C]
    void* gate;            // [n][C], float or (gate_f16) half
    int C, R, n;
    int gate_f16 = 0;
};
void launch_se_excite(const SeExciteArgs& a, hipStream_t stream);
int se_excite_split(int C);     // workgroups per crop
int se_padded_r(int R);

// ---- pw.hip -----------------------------------------------------------------------------
// 1x1 convolution as an MFMA GEMM over M = n*H*W rows:
//   out[m][:] = act( (a[m][:] * gate[m / HW][:]) @ W + bias ) (+ res[m][:])
enum { ACT_NONE = 0, ACT_SWISH = 1 };
struct SeFuse {            // second half of a SEBlock computed by the project GEMM itself (se_device.h)
    const float* rpart = nullptr;    // [n][np][RP] producer's squeeze-excite partial vectors, or nullptr (not fused)
    const float* b1 = nullptr;       // [R]
    const float* w2c = nullptr;      // [C][RP] excite kernel, channel-major, R zero-padded to RP
    const float* b2 = nullptr;       // [C]
    int np = 0, R = 0, RP = 0;
    float inv_hw = 0.f;
    // round 6 (GM = 3, the LDS-staged split-K kernel): every wave computes the gate of the k-groups IT owns on the matrix cores from
    // this operand image of the excite kernel ([ceil(R/16)][K/32][64 lanes][8] binary16; f32s: [hi | lo] images scaled by 2^shift)
    const void* w2p = nullptr;
    int KSr = 0;                     // ceil(R / 16)
    float w2_wsi = 1.0f;             // f32s: 2^-shift
};
struct PwArgs {
    const void* a;         // [M][K] T
    const void* wp;        // packed MFMA operand image (snapshot.h)
    const float* wdense;   // [K][N] (check kernel only)
    const float* bias;     // [N]
    const void* gate;      // [n][K] T or nullptr
    const void* res;       // [M][N] T or nullptr
    void* out;             // [M][N] T
    int M, K, N, KS, NTILES, HW, act;
    SeFuse se;             // set: the kernel computes the gate of its rows' crops itself (gate must be nullptr)
    // WHENET_F32S: float storage, products as binary16 hi/lo pairs on the f16 matrix pipe (pw.hip PwOps<float, true>)
    bool split = false;
    const void* wps = nullptr;     // [hi image | lo image], f16 fragment order (snapshot.cpp::pack_pw_split)
    int KSs = 0;                   // ceil(K / 16)
    float wsi = 1.0f;              // 2^-shift of the scaled weights
    bool staged = true;            // K >= 320: activation rows fetched coalesced and staged through LDS (whenet_pw_splitk_staged_kernel)
};
void launch_pw(const PwArgs& a, int dtype, int impl, int num_cus, hipStream_t stream);

// ---- head.hip ---------------------------------------------------------------------------
// GlobalAveragePooling2D + Dense(120|66|66) + softmax-expectation decode + argmax
// (whenet.py:10-13, 28-33; utils.py:7-11).
struct HeadsArgs {
    const void* x;         // [n][49][1280] T  (head conv output), or nullptr with feat_in
    const float* feat_in;  // [n][1280] (decode-only / tests) or nullptr
    const float* logits_in;// [n][252] decode-only or nullptr
    const float* w;        // [1280][252]
    const float* b;        // [252]
    float* feat;           // [n][1280] or nullptr
    float* logits;         // [n][252] or nullptr
    float* ypr;            // [n][3]
    int32_t* argmax;       // [n][3] or nullptr
    int n;
};
void launch_heads(const HeadsArgs& a, int dtype, hipStream_t stream);
// the same stage over heads_split() workgroups per crop; part: [n][heads_split()][252] floats of scratch,
// count: [n] counters that are zero before the first launch (the kernel leaves them zero)
void launch_heads_split(const HeadsArgs& a, float* part, unsigned* count, int dtype, hipStream_t stream);
int heads_split();

// ---- head7.hip --------------------------------------------------------------------------
// head conv 1x1 (320 -> 1280) + BN + Swish fused with GlobalAveragePooling2D (whenet.py:8-10), f16 and f32: only the pooled
// features leave the kernel.  A group of crops per workgroup, every crop on its own two MFMA strips (batch-invariant).
struct Head7Args {
    int dtype;             // WHENET_F16 / WHENET_F32
    const void* x;         // [n,7,7,K] T
    const void* wep;       // packed head-conv weights (MFMA fragment order, snapshot.h)
    const float* bias;     // [N]
    float* feat;           // [n][N] pooled features, f32
    int K, N, NTILES, n;
    // WHENET_F32S: products as binary16 hi/lo pairs (device_math.h PwOps<float, true>)
    bool split = false;
    const void* weps = nullptr;    // [hi image | lo image]
    float wsi = 1.0f;
    bool xcd_grouped = false;      // the channel chunks of a crop group on one XCD (device_math.h xcd_unit)
};
bool head7_supported(int dtype, int K, int N, int HW);
void launch_head7(const Head7Args& a, hipStream_t stream);
std::string kernel_name_head7(int dtype, int n, bool split = false);

// ---- stemdw.hip -------------------------------------------------------------------------
// stem conv + BN + Swish fused with block 1's depthwise 3x3 + BN + Swish (whenet.py:8, 23-26), f16 and f32: the 112 x 112 x 32 stem
// output only exists as LDS tiles.  Bitwise the two kernels' results; the tile is plan_dw()'s block-1 plan.
// what every workgroup would otherwise rebuild from the f32 tensors: built once per model on the host (build_stemdw_table),
// with the roundings stem.hip applies on the device
struct StemDwTable {
    uint32_t lut[3 * 256];          // the normalisation LUT as binary16 hi | lo << 16 (v = hi + lo)
    half8 whi[3][64], wlo[3][64];   // stem weights split the same way, as MFMA fragments per kernel row and lane
};
void build_stemdw_table(const float* w /* [27][32] */, const float* lut /* [3][256] */, StemDwTable* out);

struct StemDwArgs {
    int dtype;             // WHENET_F16 / WHENET_F32
    const uint8_t* in;     // [n,224,224,3]
    void* out;             // [n,112,112,32] T: block 1's depthwise output
    const StemDwTable* tab;// (device) f16
    const float* w;        // f32: stem [27][32]
    const float* lut;      // f32: [3][256]
    const float* bias;     // stem [32]
    const float* wd;       // depthwise [9][32]
    const float* bd;       // depthwise bias [32]
    float* partial;        // [n][56 tiles][32] channel sums for se.hip
    int n;
};
bool stemdw_supported(int dtype, const DwPlan& p, int k, int s, int H, int C);
void launch_stemdw(const StemDwArgs& a, hipStream_t stream);
const char* kernel_name_stemdw(int dtype);

// ---- front.hip --------------------------------------------------------------------------
// expand 1x1 (MFMA) + BN + Swish -> depthwise kxk + BN + Swish in one kernel (blocks 2..16).
struct FrontPlan {
    int threads = 256;     // lanes per workgroup (the tile is planned for 256 tap lanes; the extra
                           // waves only share the expand tasks)
    int CC = 32;           // expanded channels per workgroup
    int TH = 1, NSX = 1;   // output tile: TH rows x 7*NSX columns
    int tiles_x = 1, tiles_y = 1, chunks = 1;
    int EH = 0, EW = 0;    // LDS tile of the expanded input (with halo)
    int EP = 0;            // pixel pitch of that tile in bytes (channels + bank-conflict padding)
    int w_off = 0;         // byte offset of the depthwise taps in LDS
    size_t lds_bytes = 0;
    int ntiles() const { return tiles_x * tiles_y; }
};
FrontPlan plan_front(int dtype, int k, int s, int H, int Ho, int Cexp);
std::vector<FrontPlan> plan_front_candidates(int dtype, int k, int s, int H, int Ho, int Cexp,
                                             std::vector<double>* scores);
int front_threads(const FrontPlan& p, int n);    // lanes per workgroup for a launch of n crops (256 | 512)
struct FrontArgs {
    const void* x;         // [n,H,H,Cin] T  block input
    const void* wep;       // packed expand weights (MFMA fragment order)
    const float* be;       // [Cexp]
    const float* wd;       // [k*k][Cexp]
    const float* bd;       // [Cexp]
    void* out;             // [n,Ho,Ho,Cexp] T
    float* rpart;          // w1t != NULL: [n][ntiles][chunks][RP] this workgroup's share of the SE reduce conv
                           // (unscaled);  w1t == NULL: [n][ntiles][Cexp] the tile's channel sums (for launch_se)
    const float* w1t;      // [R][Cexp] se_reduce kernel, transposed, or NULL
    int R;
    int k, s, H, Ho, Cin, Cexp, pad, KSe, NTe, n;
    FrontPlan plan;
    // WHENET_F32S: the expand's products as binary16 hi/lo pairs (device_math.h PwOps<float, true>)
    bool split = false;
    const void* weps = nullptr;    // [hi image | lo image] of the expand weights
    int KSes = 0;                  // ceil(Cin / 16)
    float wsi = 1.0f;
    const float* in_gate = nullptr; // split only: [n][Cin] f32 -- the expand contracts (in_gate[crop] * x): block 2 fed by block 1's
                                    // depthwise output with block 1's project folded into weps (engine.cpp, option fold12)
    bool xcd_grouped = false;       // the channel chunks of a (crop, tile) on one XCD (device_math.h xcd_unit; engine option "xcd_map")
};
void launch_front(const FrontArgs& a, int dtype, hipStream_t stream);
std::string kernel_name_front(int dtype, int k, int s, int threads);

// ---- front2.hip -------------------------------------------------------------------------
// f16: the same stage with the depthwise taps on the matrix cores (per-channel Toeplitz products,
// v_mfma_f32_4x4x4_16B_f16) and the expanded tile channel-major in LDS.
struct Front2Plan {
    int threads = 256;     // lanes per workgroup (256 | 512)
    int CC = 32;           // expanded channels per workgroup (multiple of 32, or the whole layer)
    int TH = 7;            // output rows per tile (multiple of 7)
    int TXG = 1;           // 4-pixel output groups per tile row
    int xs = 0;            // tile origin moved this many pixels to the left (0 | 2)
    int tiles_x = 1, tiles_y = 1, chunks = 1;
    int EH = 0, EWp = 0;   // LDS tile of the expanded input: rows, pixels per row (multiple of 4)
    int RP = 0, CP = 0;    // row / channel pitch in bytes
    int off_stage = 0, off_red = 0, off_sum = 0;
    size_t lds_bytes = 0;
    int ntiles() const { return tiles_x * tiles_y; }
};
Front2Plan make_front2_plan(int k, int s, int Ho, int Cexp, int CC, int TH, int TXG, int threads, int xs);
Front2Plan plan_front2(int k, int s, int H, int Ho, int Cexp);
std::vector<Front2Plan> plan_front2_candidates(int k, int s, int Ho, int Cexp);
int front2_threads(const Front2Plan& p, int n);
std::vector<half_t> pack_dw_toeplitz(const std::vector<float>& w, int k, int s, int C, int xs);
bool front2_preferred(int k, int s, int H, int Cexp);
struct Front2Args {
    const void* x;         // [n,H,H,Cin] half  block input
    const void* wep;       // packed expand weights (MFMA fragment order, snapshot.h)
    const float* be;       // [Cexp]
    const void* wdt;       // pack_dw_toeplitz() image of the depthwise kernel (for plan.xs)
    const float* bd;       // [Cexp]
    void* out;             // [n,Ho,Ho,Cexp] half
    float* rpart;          // as FrontArgs::rpart
    const float* w1t;      // [R][Cexp] se_reduce kernel, transposed, or NULL
    const float* in_gate;  // [n][Cin] f32 or NULL: the expand contracts (in_gate[crop] * x) -- block 2 fed by block 1's
                           // depthwise output with block 1's project folded into wep (engine.cpp, option fold12);
                           // wep is then the F32 fragment image (8 floats per lane, HostModel::fold12_w32)
    int R;
    int k, s, H, Ho, Cin, Cexp, pad, KSe, NTe, n;
    Front2Plan plan;
    bool xcd_grouped = false;       // as FrontArgs::xcd_grouped
};
void launch_front2(const Front2Args& a, hipStream_t stream);
std::string kernel_name_front2(int k, int s, int kse, int threads, int xs, bool gated);

// ---- front2s.hip ------------------------------------------------------------------------
// WHENET_F32S (float32 storage): the same stage with both convolutions on the matrix cores -- expand as binary16 hi/lo
// products (PwOps<float, true>), taps as per-channel Toeplitz products, tile in LDS at float32 precision.
struct Front2sArgs {
    const void* x;         // [n,H,H,Cin] float, or (pre) [n,H,H][hi Cin | lo Cin] binary16 pairs written by the producer
    const void* weps;      // [hi image | lo image] of the expand weights (snapshot.cpp::pack_pw_split)
    const float* be;       // [Cexp]
    const void* wdt;       // pack_dw_toeplitz_s() image of the depthwise kernel for tap mode tm
    const float* bd;       // [Cexp]
    void* out;             // [n,Ho,Ho,Cexp] float
    float* rpart;          // as FrontArgs::rpart
    const float* w1t;      // [R][Cexp] se_reduce kernel, transposed, or NULL
    int R;
    int k, s, H, Ho, Cin, Cexp, pad, KSe, NTe, n;
    float wsi = 1.0f;      // 2^-shift of the scaled expand weights
    float wsi_d = 1.0f;    // 2^-shift of the scaled depthwise taps (tm = 1)
    int tm = 2;            // taps: 1 = binary16 hi/lo pairs on v_mfma_f32_4x4x4_16B_f16, 2 = exact float32 on v_mfma_f32_4x4x1_16B_f32
    bool pre = false;      // x is in the pre-split pair form
    Front2Plan plan;       // from make_front2s_plan() (16 bytes per 4-pixel group)
};
bool front2s_supported(int k, int s, int H, int Cin);
bool front2s_preferred(int k, int s, int H, int Cexp);
Front2Plan make_front2s_plan(int k, int s, int Ho, int Cexp, int CC, int TH, int TXG, int threads);
Front2Plan plan_front2s(int k, int s, int H, int Ho, int Cexp, int* tm);
std::vector<Front2Plan> plan_front2s_candidates(int k, int s, int Ho, int Cexp);
std::vector<float> pack_dw_toeplitz_s(const std::vector<float>& w, int k, int s, int C, int tm, float* wsi);
void launch_front2s(const Front2sArgs& a, hipStream_t stream);
std::string kernel_name_front2s(int k, int s, int kse, int threads, int tm, bool pre);

// ---- front7.hip -------------------------------------------------------------------------
// the 7 x 7 blocks (13-16), both dtypes: the same stage with a GROUP of G crops per workgroup, the image-only LDS tile (no halo)
// and the chunk's expand weights staged once in LDS (round 4).
struct Front7Plan {
    int threads = 512;     // lanes per workgroup (256 | 512)
    int G = 4;             // crops per workgroup
    int CC = 64;           // expanded channels per workgroup (32 | 64 | 128)
    int chunks = 1;
    int off_w = 0, off_stage = 0, off_red = 0, off_sum = 0;
    size_t lds_bytes = 0;
};
Front7Plan make_front7_plan(int dtype, int Cin, int Cexp, int G, int CC, int threads);
Front7Plan front7_plan_for(int dtype, int Cin, int Cexp, int n);
bool front7_supported(int k, int s, int H, int Cin);
struct Front7Args {
    int dtype = WHENET_F16;
    const void* x;         // [n,7,7,Cin] T  block input
    const void* wep;       // packed expand weights (MFMA fragment order, snapshot.h)
    const float* be;       // [Cexp]
    const void* wdt;       // f16: pack_dw_toeplitz(w, k, 1, Cexp, 4 - k / 2) image of the depthwise kernel; f32: the kernel, [k*k][Cexp]
    const float* bd;       // [Cexp]
    void* out;             // [n,7,7,Cexp] T
    float* rpart;          // [n][chunks][RP] this workgroup's share of the SE reduce conv, per crop (unscaled)
    const float* w1t;      // [R][Cexp] se_reduce kernel, transposed
    int R;
    int k, Cin, Cexp, NTe, n;
    Front7Plan plan;
    // WHENET_F32S: the expand's products as binary16 hi/lo pairs (device_math.h PwOps<float, true>)
    bool split = false;
    const void* weps = nullptr;    // [hi image | lo image] of the expand weights
    float wsi = 1.0f;
    bool xcd_grouped = false;      // the channel chunks of a crop group on one XCD (device_math.h xcd_unit)
};
void launch_front7(const Front7Args& a, hipStream_t stream);
std::string kernel_name_front7(int dtype, int k, const Front7Plan& p, bool split = false);

// ---- mb7.hip ----------------------------------------------------------------------------
// f16, blocks 13-16 (7 x 7 maps, 192 -> 1152 -> 192 | 320): the whole MBConv block -- expand, depthwise, squeeze-excite, gate, project,
// skip -- as ONE launch, one workgroup per crop, every intermediate tensor in LDS (round 6).
struct Mb7Args {
    const void* x;         // [n,7,7,192] half  block input (also the skip operand)
    const void* wep;       // packed expand weights (MFMA fragment order, snapshot.h)
    const float* be;       // [1152]
    const void* wds;       // pack_mb7_taps() image of the depthwise kernel
    const float* bd;       // [1152]
    const void* w1p;       // pack_mb7_se(): se_reduce kernel, binary16 [1152][6][8]
    const float* b1;       // [48]
    const void* w2p;       // pack_mb7_se(): se_expand kernel, binary16 [6][1152][8]
    const float* b2;       // [1152]
    const void* wpp;       // packed project weights (MFMA fragment order)
    const float* bp;       // [Cout]
    void* out;             // [n,7,7,Cout] half
    void* dbg_dw = nullptr;    // single-stage calls (tests): the depthwise output [n,7,7,1152] half ...
    void* dbg_gate = nullptr;  // ... and the gate [n,1152] half; nullptr in the forward pass
    int k, Cout, n;
    bool skip;
};
bool mb7_supported(int dtype, int k, int s, int H, int Cin, int Cexp, int R, int Cout, bool skip);
std::vector<half_t> pack_mb7_taps(const std::vector<float>& w /* [k*k][C] */, int k, int C);
void pack_mb7_se(const std::vector<float>& w1t /* [R][C] */, const std::vector<float>& w2 /* [R][C] */, int C, int R,
                 std::vector<half_t>* w1p, std::vector<half_t>* w2p);
void launch_mb7(const Mb7Args& a, hipStream_t stream);
std::string kernel_name_mb7(int k, int Cout, bool skip);

// ---- yolo.hip ---------------------------------------------------------------------------
// YOLOv3 post-processing (yolo_v3/model.py:125-232): decode + score threshold + per-class NMS.
struct YoloLayer {
    const float* feats;    // device [gh][gw][A*(5+C)]
    int gh, gw;
    int first;             // index of this layer's first box in the concatenated list
    float anchor[3][2];    // (w, h) of the layer's anchors (anchor_mask applied)
};
struct YoloArgs {
    YoloLayer layer[3];
    int num_layers, num_classes, na;       // na = anchors per layer (3)
    int N, NP;                             // boxes in total; key capacity per class (power of two >= N)
    float input_h, input_w, image_h, image_w;
    float off_y, off_x, scale_y, scale_x;  // letterbox correction (model.py:160-162), float32 as the graph computes it
    float score_thr, iou_thr;
    int max_boxes;
    float* boxes;                          // [N][4] y_min, x_min, y_max, x_max
    float* all_scores;                     // [N][C] or nullptr (tests)
    int* counts;                           // [C]
    unsigned long long* keys;              // [C][NP]
    float* out_boxes;                      // [C][max_boxes][4]
    float* out_scores;                     // [C][max_boxes]
    int* out_index;                        // [C][max_boxes]
    int* out_count;                        // [C]
};
void launch_yolo_eval(const YoloArgs& a, hipStream_t stream);
int yolo_max_select();

// ---- frame.hip --------------------------------------------------------------------------
// Per-head pre-processing of a frame (demo_video.py:13-24): bbox margins -> crop window ->
// (BGR->RGB) -> cv2.resize-compatible fixed-point bilinear to 224x224 uint8.
constexpr int CROP_PLAN_INTS = 8 + 6 * IMG;     // header {y0,x0,h,w,area2x,xmax,-,-} + xofs|a0|a1|yofs|b0|b1
void frame_box_rect(int frame_h, int frame_w, const float bbox[4], int32_t rect[4]);
void build_crop_plan(const int32_t rect[4], int32_t* plan);
void launch_crop_resize(const uint8_t* d_frame, int fw, int swap_rb, const int32_t* d_plan, int k, uint8_t* d_out,
                        hipStream_t stream);

// ---- convert.hip ------------------------------------------------------------------------
void launch_empty(hipStream_t stream);   // boundary calibration for whenet_profile()
void launch_f32_to_act(const float* src, void* dst, size_t count, int dtype, hipStream_t stream);
void launch_act_to_f32(const void* src, float* dst, size_t count, int dtype, hipStream_t stream);

const char* kernel_name_stem(int dtype);
const char* kernel_name_dw(int dtype, int k, int s);
std::string kernel_name_pw(const PwArgs& a, int dtype, int impl, int num_cus);

}  // namespace whenet
