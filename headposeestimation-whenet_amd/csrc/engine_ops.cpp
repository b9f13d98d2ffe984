// Single-stage entry points (whenet_op_*): run exactly the kernels the forward uses on caller-supplied inputs, so that every
// kernel can be compared with the oracle on every layer shape (tests/test_gpu_parity.py), plus the raw device-memory helpers.
#include "engine_internal.h"

namespace whenet {

using namespace detail;

void Engine::op_stem(const uint8_t* crops, int n, float* out) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(crops && out, WHENET_EINVAL, "op_stem: NULL argument");
    ensure_capacity(n);
    TempBufs tmp;
    const size_t N = size_t(n);
    float* d_out = static_cast<float*>(tmp.get(N * X_ELEMS * sizeof(float)));
    WHENET_HIP_CHECK(hipMemcpyAsync(in_u8_, crops, N * IN_BYTES, hipMemcpyHostToDevice, stream_));
    StemArgs a{in_u8_, x0_, d_stem_w_, d_stem_b_, d_lut_, n};
    launch_stem(a, dtype_, stream_);
    launch_act_to_f32(x0_, d_out, N * X_ELEMS, dtype_, stream_);
    WHENET_HIP_CHECK(hipMemcpyAsync(out, d_out, N * X_ELEMS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::op_block(int index, const float* in, int n, float* expand_out, float* dw_out, float* gate, float* out) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(index >= 1 && index <= int(blocks_.size()), WHENET_EINVAL, "op_block: index must be 1..16");
    WHENET_REQUIRE(in != nullptr, WHENET_EINVAL, "op_block: NULL input");
    ensure_capacity(n);
    const DevBlock& b = blocks_[size_t(index - 1)];
    const BlockSpec& sp = b.spec;
    const size_t N = size_t(n);
    const size_t in_elems = N * sp.h_in * sp.h_in * sp.cin;
    const size_t exp_elems = N * sp.h_in * sp.h_in * sp.cexp();
    const size_t dw_elems = N * sp.h_out * sp.h_out * sp.cexp();
    const size_t out_elems = N * sp.h_out * sp.h_out * sp.cout;
    TempBufs tmp;
    float* d_f32 = static_cast<float*>(tmp.get(std::max({in_elems, exp_elems, dw_elems, out_elems}) * sizeof(float)));
    WHENET_HIP_CHECK(hipMemcpyAsync(d_f32, in, in_elems * sizeof(float), hipMemcpyHostToDevice, stream_));
    launch_f32_to_act(d_f32, x0_, in_elems, dtype_, stream_);
    WHENET_HIP_CHECK(hipMemsetAsync(gate_, 0xff, N * 1152 * sizeof(float), stream_));     // (NaN unless a launch writes it)
    single_stage_call_ = true;
    try {
        enqueue_block(b, view(0), x0_, x1_, n, stream_, nullptr);
    } catch (...) {
        single_stage_call_ = false;
        throw;
    }
    single_stage_call_ = false;
    auto fetch = [&](const void* src, size_t elems, float* dst) {
        if (!dst) return;
        launch_act_to_f32(src, d_f32, elems, dtype_, stream_);
        WHENET_HIP_CHECK(hipMemcpyAsync(dst, d_f32, elems * sizeof(float), hipMemcpyDeviceToHost, stream_));
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    };
    if (sp.has_expand()) fetch(e_, exp_elems, expand_out);
    fetch(d_, dw_elems, dw_out);
    if (gate) {
        fetch(gate_, N * sp.cexp(), gate);         // (stored in the activation type: see se.hip)
    }
    fetch(x1_, out_elems, out);
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::op_block_range(int first, int last, const float* in, int n, float* out) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(first >= 1 && first <= last && last <= int(blocks_.size()), WHENET_EINVAL,
                   "op_block_range: need 1 <= first <= last <= 16");
    WHENET_REQUIRE(in != nullptr && out != nullptr, WHENET_EINVAL, "op_block_range: NULL buffer");
    ensure_capacity(n);
    const BlockSpec& si = blocks_[size_t(first - 1)].spec;
    const BlockSpec& so = blocks_[size_t(last - 1)].spec;
    const size_t in_elems = size_t(n) * si.h_in * si.h_in * si.cin;
    const size_t out_elems = size_t(n) * so.h_out * so.h_out * so.cout;
    TempBufs tmp;
    float* d_f32 = static_cast<float*>(tmp.get(std::max(in_elems, out_elems) * sizeof(float)));
    WHENET_HIP_CHECK(hipMemcpyAsync(d_f32, in, in_elems * sizeof(float), hipMemcpyHostToDevice, stream_));
    launch_f32_to_act(d_f32, x0_, in_elems, dtype_, stream_);
    const View v = view(0);
    single_stage_call_ = true;
    const void* res = nullptr;
    try {
        res = enqueue_blocks(first, last, v, v.x0, n, stream_, nullptr);
    } catch (...) {
        single_stage_call_ = false;
        throw;
    }
    single_stage_call_ = false;
    launch_act_to_f32(res, d_f32, out_elems, dtype_, stream_);
    WHENET_HIP_CHECK(hipMemcpyAsync(out, d_f32, out_elems * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::op_head(const float* in, int n, float* feat, float* logits, float* ypr, int32_t* argmax) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(in != nullptr, WHENET_EINVAL, "op_head: NULL input");
    ensure_capacity(n);
    const size_t N = size_t(n);
    const size_t in_elems = N * 49 * 320;
    TempBufs tmp;
    float* d_f32 = static_cast<float*>(tmp.get(in_elems * sizeof(float)));
    float* d_feat = static_cast<float*>(tmp.get(N * FEAT * sizeof(float)));
    WHENET_HIP_CHECK(hipMemcpyAsync(d_f32, in, in_elems * sizeof(float), hipMemcpyHostToDevice, stream_));
    launch_f32_to_act(d_f32, x0_, in_elems, dtype_, stream_);
    const bool fuse_head = head_fuse_ && pw_impl_ == 0 && head7_supported(dtype_, head_.K, head_.N, 49);
    HeadsArgs h{};
    if (fuse_head) {          // the forward's form: head conv + pooling as one kernel (head7.hip), Dense heads on its features
        Head7Args a{};
        a.dtype = dtype_;
        a.x = x0_;
        a.wep = head_.wp;
        a.bias = head_.bias;
        a.feat = d_feat;
        a.K = head_.K;
        a.N = head_.N;
        a.NTILES = head_.NTILES;
        a.split = split_ && split_pw_ && head_.wps != nullptr;
        a.weps = head_.wps;
        a.wsi = head_.wsi;
        a.n = n;
        launch_head7(a, stream_);
        h.feat_in = d_feat;
    } else {
        PwArgs a{};
        a.a = x0_;
        a.wp = head_.wp;
        a.wdense = head_.wdense;
        a.bias = head_.bias;
        a.out = hc_;
        a.M = n * 49;
        a.K = head_.K;
        a.N = head_.N;
        a.KS = head_.KS;
        a.NTILES = head_.NTILES;
        set_split(a, head_);
        a.HW = 49;
        a.act = ACT_SWISH;
        launch_pw(a, dtype_, pw_impl_, num_cus_, stream_);
        h.x = hc_;
        h.feat = d_feat;
    }
    h.w = d_dense_w_;
    h.b = d_dense_b_;
    h.logits = o_logits_;
    h.ypr = o_ypr_;
    h.argmax = o_amax_;
    h.n = n;
    launch_heads(h, dtype_, stream_);
    if (feat) WHENET_HIP_CHECK(hipMemcpyAsync(feat, d_feat, N * FEAT * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (logits) WHENET_HIP_CHECK(hipMemcpyAsync(logits, o_logits_, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (ypr) WHENET_HIP_CHECK(hipMemcpyAsync(ypr, o_ypr_, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(argmax, o_amax_, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::op_decode(const float* logits, int n, float* ypr, int32_t* argmax) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(logits && ypr, WHENET_EINVAL, "op_decode: NULL argument");
    ensure_capacity(n);
    const size_t N = size_t(n);
    TempBufs tmp;
    float* d_lg = static_cast<float*>(tmp.get(N * N_LOGITS * sizeof(float)));
    WHENET_HIP_CHECK(hipMemcpyAsync(d_lg, logits, N * N_LOGITS * sizeof(float), hipMemcpyHostToDevice, stream_));
    HeadsArgs h{};
    h.logits_in = d_lg;
    h.w = d_dense_w_;
    h.b = d_dense_b_;
    h.ypr = o_ypr_;
    h.argmax = o_amax_;
    h.n = n;
    launch_heads(h, WHENET_F32, stream_);
    WHENET_HIP_CHECK(hipMemcpyAsync(ypr, o_ypr_, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(argmax, o_amax_, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

// ------------------------------------------------------------------------------------------
void* Engine::dev_alloc(size_t nbytes) {
    DeviceGuard guard(device_);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, nbytes ? nbytes : 16);
    if (e != hipSuccess) throw Error(WHENET_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return p;
}
void Engine::dev_free(void* p) {
    DeviceGuard guard(device_);
    if (p) WHENET_HIP_CHECK(hipFree(p));
}
void Engine::h2d(void* d, const void* s, size_t nbytes) {
    DeviceGuard guard(device_);
    WHENET_HIP_CHECK(hipMemcpy(d, s, nbytes, hipMemcpyHostToDevice));
}
void Engine::d2h(void* d, const void* s, size_t nbytes) {
    DeviceGuard guard(device_);
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    WHENET_HIP_CHECK(hipMemcpy(d, s, nbytes, hipMemcpyDeviceToHost));
}

}  // namespace whenet
