#include "engine.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>

namespace whenet {

namespace {

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

void copy_name(char* dst, size_t cap, const std::string& s) {
    std::memset(dst, 0, cap);
    std::memcpy(dst, s.data(), std::min(cap - 1, s.size()));
}

}  // namespace

// ------------------------------------------------------------------------------------------
// construction / weights
// ------------------------------------------------------------------------------------------
void* Engine::upload_bytes(const void* p, size_t nbytes) {
    void* d = nullptr;
    WHENET_HIP_CHECK(hipMalloc(&d, nbytes ? nbytes : 16));
    weight_allocs_.push_back(d);
    if (nbytes) WHENET_HIP_CHECK(hipMemcpy(d, p, nbytes, hipMemcpyHostToDevice));
    return d;
}

template <typename T> T* Engine::upload(const std::vector<T>& v) {
    return static_cast<T*>(upload_bytes(v.data(), v.size() * sizeof(T)));
}

DevPw Engine::upload_pw(const HostPw& h) {
    DevPw d;
    d.K = h.K;
    d.N = h.N;
    d.KS = h.KS;
    d.NTILES = h.NTILES;
    d.wp = upload_bytes(h.packed.data(), h.packed.size());
    d.wdense = upload(h.dense);
    d.bias = upload(h.bias);
    return d;
}

Engine::Engine(const void* snapshot, size_t nbytes, int device_id, int dtype) : device_(device_id), dtype_(dtype) {
    // host-side preparation first: a malformed snapshot is reported as such even on a box
    // without a GPU
    HostModel m = build_host_model(parse_snapshot(snapshot, nbytes), dtype);
    params_backbone_ = m.params_backbone;
    params_heads_ = m.params_heads;
    n_tensors_ = m.n_tensors;

    open_device(device_id);
    DeviceGuard guard(device_);
    has_model_ = true;

    d_lut_ = static_cast<float*>(upload_bytes(&m.lut[0][0], sizeof(m.lut)));
    d_stem_w_ = upload(m.stem_w);
    d_stem_b_ = upload(m.stem_b);
    for (const HostBlock& hb : m.blocks) {
        DevBlock b;
        b.spec = hb.spec;
        if (hb.spec.has_expand()) b.expand = upload_pw(hb.expand);
        b.dw.k = hb.dw.k;
        b.dw.C = hb.dw.C;
        b.dw.w = upload(hb.dw.w);
        b.dw.bias = upload(hb.dw.bias);
        b.dw.plan = plan_dw(dtype_, hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.h_out, hb.dw.C);
        b.se.C = hb.se.C;
        b.se.R = hb.se.R;
        b.se.w1t = upload(hb.se.w1t);
        b.se.b1 = upload(hb.se.b1);
        b.se.w2 = upload(hb.se.w2);
        b.se.w2c = upload(hb.se.w2c);
        b.se.b2 = upload(hb.se.b2);
        b.project = upload_pw(hb.project);
        partial_per_crop_ = std::max(partial_per_crop_, size_t(b.dw.plan.ntiles()) * b.dw.C);
        if (hb.spec.has_expand()) {
            b.fplan = plan_front(dtype_, hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.h_out, hb.dw.C);
            partial_per_crop_ = std::max(partial_per_crop_, size_t(b.fplan.ntiles()) * b.dw.C);
            // blocks whose front kernel applies the SE reduce conv write [tiles][chunks][RP] partial vectors: with
            // narrow chunks that exceeds [tiles][C] (C = 1152, 32-channel chunks: 36 x 48 = 1728 floats per tile)
            partial_per_crop_ = std::max(partial_per_crop_, size_t(b.fplan.ntiles()) * size_t(b.fplan.chunks) *
                                                                size_t(se_padded_r(b.se.R)));
            if (dtype_ == WHENET_F16) {
                // f16: the same stage with the depthwise taps on the matrix cores (front2.hip) where that kernel is the
                // faster one; its Toeplitz image of the depthwise kernel is built once here
                b.f2plan = plan_front2(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.h_out, hb.dw.C);
                b.f2_preferred = front2_preferred(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.dw.C);
                b.dw.wt = upload(pack_dw_toeplitz(hb.dw.w, hb.spec.k, hb.spec.s, hb.dw.C, b.f2plan.xs));
                partial_per_crop_ = std::max(partial_per_crop_, size_t(b.f2plan.ntiles()) * b.dw.C);
                partial_per_crop_ = std::max(partial_per_crop_, size_t(b.f2plan.ntiles()) * size_t(b.f2plan.chunks) *
                                                                    size_t(se_padded_r(b.se.R)));
            }
        }
        blocks_.push_back(b);
    }
    head_ = upload_pw(m.head);
    fold12_pw_ = upload_pw(m.fold12);
    d_fold12_w32_ = upload(m.fold12_w32);
    d_dense_w_ = upload(m.dense_w);
    d_dense_b_ = upload(m.dense_b);
    WHENET_HIP_CHECK(hipDeviceSynchronize());
}

void Engine::open_device(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw Error(WHENET_ENODEV, std::string("no HIP device visible (") + hipGetErrorString(e) +
                                       "); libwhenet_hip has no CPU fallback");
    WHENET_REQUIRE(device_id >= 0 && device_id < count, WHENET_ENODEV,
                   "device_id " + std::to_string(device_id) + " out of range (" + std::to_string(count) + " devices)");
    DeviceGuard guard(device_id);
    WHENET_HIP_CHECK(hipGetDeviceProperties(&prop_, device_id));
    WHENET_REQUIRE(std::strstr(prop_.gcnArchName, "gfx950") != nullptr, WHENET_ENODEV,
                   std::string("device is ") + prop_.gcnArchName + "; this library carries gfx950 code only");
    num_cus_ = prop_.multiProcessorCount;

    WHENET_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    // Only the main stream exists up front: the runtime places a new stream on its least-loaded
    // hardware queue (4 of them), so streams are created in the order concurrency needs them -- the
    // main streams of the engines of one handle, then sub-batch lanes / the copy stream on first use --
    // instead of nine per engine, most of them idle ballast that skews that placement.
    WHENET_HIP_CHECK(hipEventCreateWithFlags(&fork_ev_, hipEventDisableTiming));
    for (int i = 0; i < MAX_LANES - 1; ++i) {
        hipEvent_t ev = nullptr;
        WHENET_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        join_ev_.push_back(ev);
    }
}

// A handle without a network: what whenet_yolo_eval / whenet_op_crop_resize need (device, stream, scratch).
Engine::Engine(int device_id) : device_(device_id), dtype_(WHENET_F32) { open_device(device_id); }

void Engine::require_model() const {
    WHENET_REQUIRE(has_model_, WHENET_EINVAL, "this handle was created without a network (whenet_create_postproc)");
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
    graphs_.clear();
    for (Slot& s : slots_) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_ypr) (void)hipHostFree(s.h_ypr);
        if (s.h_amax) (void)hipHostFree(s.h_amax);
        if (s.h_logits) (void)hipHostFree(s.h_logits);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_ypr) (void)hipFree(s.d_ypr);
        if (s.d_amax) (void)hipFree(s.d_amax);
        if (s.d_logits) (void)hipFree(s.d_logits);
        if (s.h_frame) (void)hipHostFree(s.h_frame);
        if (s.d_frame) (void)hipFree(s.d_frame);
        if (s.h_plan) (void)hipHostFree(s.h_plan);
        if (s.d_plan) (void)hipFree(s.d_plan);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    void* arena[] = {x0_, x1_, e_, d_, hc_, partial_, gate_, hcount_, in_u8_, o_ypr_, o_amax_, o_logits_, in_f32_, yolo_scratch_};
    for (void* p : arena)
        if (p) (void)hipFree(p);
    for (void* p : weight_allocs_) (void)hipFree(p);
    for (hipStream_t st : lane_streams_) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t ev : join_ev_) (void)hipEventDestroy(ev);
    if (fork_ev_) (void)hipEventDestroy(fork_ev_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
}

void Engine::set_option(const std::string& key, long value) {
    DeviceGuard guard(device_);
    if (key == "graph") {
        use_graph_ = value != 0;
    } else if (key == "pw_impl") {
        WHENET_REQUIRE(value == 0 || value == 1, WHENET_EINVAL, "pw_impl must be 0 (MFMA) or 1 (check kernel)");
        pw_impl_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "fuse_front") {
        fuse_front_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "se_fuse") {
        WHENET_REQUIRE(value >= 0 && value <= 2, WHENET_EINVAL, "se_fuse must be 0 (never), 1 (where it pays) or 2 (always)");
        se_fuse_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "front_impl") {
        WHENET_REQUIRE(value >= 0 && value <= 2, WHENET_EINVAL,
                       "front_impl must be 0 (front.hip everywhere), 1 (per layer, default) or 2 (front2.hip everywhere, f16)");
        front_impl_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "fold12") {
        fold12_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "poison") {
        poison_ = value != 0;
    } else if (key == "lanes") {
        WHENET_REQUIRE(value >= 1 && value <= MAX_LANES, WHENET_EINVAL, "lanes must be 1..8");
        lanes_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "split_heads") {
        split_heads_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "lane_graphs") {
        lane_graphs_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "min_lane_crops") {
        WHENET_REQUIRE(value >= 1, WHENET_EINVAL, "min_lane_crops must be >= 1");
        min_lane_crops_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "repeat") {
        WHENET_REQUIRE(value >= 1 && value <= 16, WHENET_EINVAL, "repeat must be 1..16");
        repeat_ = int(value);
        sync();
        drop_graphs();
    } else {
        throw Error(WHENET_EINVAL, "unknown option '" + key + "'");
    }
}

void Engine::get_info(whenet_info_t* out) const {
    std::memset(out, 0, sizeof(*out));
    out->abi_version = WHENET_ABI_VERSION;
    out->dtype = dtype_;
    out->device_id = device_;
    out->compute_units = num_cus_;
    out->params_backbone = params_backbone_;
    out->params_heads = params_heads_;
    out->n_tensors = n_tensors_;
    {
        // stem + per block {expand, dw | front} {se} project + head conv + heads
        int k = 1 + 2;
        for (const DevBlock& b : blocks_) {
            const bool has_expand = b.spec.expand != 1;
            const bool front = fuse_front_ && has_expand;
            k += front ? 1 : (has_expand ? 2 : 1);
            {   // (project alone when it computes the gate itself: the rule of enqueue_block)
                const bool f2 = dtype_ == WHENET_F16 && (front_impl_ == 2 || (front_impl_ == 1 && b.f2_preferred));
                const int np = f2 ? b.f2plan.ntiles() * b.f2plan.chunks : b.fplan.ntiles() * b.fplan.chunks;
                const bool pays = b.project.K < 320 && np <= 24;
                k += (front && pw_impl_ == 0 && (se_fuse_ == 2 || (se_fuse_ == 1 && pays))) ? 1 : 2;
            }
        }
        if (fold12_active()) k -= 1;              // block 1's project launch
        out->n_kernels_per_forward = k;
    }
    out->macs_per_crop = 384857312;
    out->arena_bytes = int64_t(arena_bytes_);
    out->capacity = cap_;
    out->graph_enabled = use_graph_ ? 1 : 0;
    copy_name(out->device_name, sizeof(out->device_name), prop_.name);
    copy_name(out->arch, sizeof(out->arch), prop_.gcnArchName);
}

// ------------------------------------------------------------------------------------------
// arena
// ------------------------------------------------------------------------------------------
void Engine::drop_graphs() {
    for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
    graphs_.clear();
}

void Engine::release_arena() {
    void** arena[] = {&x0_, &x1_, &e_, &d_, &hc_, reinterpret_cast<void**>(&partial_), reinterpret_cast<void**>(&gate_), reinterpret_cast<void**>(&hcount_),
                      reinterpret_cast<void**>(&in_u8_), reinterpret_cast<void**>(&o_ypr_),
                      reinterpret_cast<void**>(&o_amax_), reinterpret_cast<void**>(&o_logits_)};
    for (void** p : arena) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    cap_ = 0;
    arena_bytes_ = 0;
}

void Engine::ensure_capacity(int n) {
    WHENET_REQUIRE(n >= 1, WHENET_EINVAL, "n must be >= 1");
    if (n <= cap_) return;
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    drop_graphs();
    release_arena();
    const size_t N = size_t(n), es = esz();
    size_t total = 0;
    auto alloc = [&](size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            release_arena();
            throw Error(WHENET_ENOMEM, "activation arena for n=" + std::to_string(n) + ": " + hipGetErrorString(e));
        }
        total += bytes;
        return p;
    };
    x0_ = alloc(N * X_ELEMS * es);
    x1_ = alloc(N * X_ELEMS * es);
    e_ = alloc(N * E_ELEMS * es);
    d_ = alloc(N * D_ELEMS * es);
    hc_ = alloc(N * HC_ELEMS * es);
    partial_ = static_cast<float*>(alloc(N * partial_per_crop_ * sizeof(float)));
    gate_ = static_cast<float*>(alloc(N * 1152 * sizeof(float)));
    hcount_ = static_cast<unsigned*>(alloc(N * sizeof(unsigned)));
    // (on the engine's OWN stream, and complete before any forward is enqueued: hipMemset runs on the null stream, which
    //  the engine's non-blocking streams do not wait for -- with other engines keeping the GPU busy it used to land in
    //  the middle of this engine's first heads kernel and leave the per-crop ticket counters off by one for good)
    WHENET_HIP_CHECK(hipMemsetAsync(hcount_, 0, N * sizeof(unsigned), stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    in_u8_ = static_cast<uint8_t*>(alloc(N * IN_BYTES));
    o_ypr_ = static_cast<float*>(alloc(N * 3 * sizeof(float)));
    o_amax_ = static_cast<int32_t*>(alloc(N * 3 * sizeof(int32_t)));
    o_logits_ = static_cast<float*>(alloc(N * N_LOGITS * sizeof(float)));
    cap_ = n;
    arena_bytes_ = total;
}

// ------------------------------------------------------------------------------------------
// the launch schedule
// ------------------------------------------------------------------------------------------
namespace {

struct Rec {
    LaunchRecorder* rec;
    hipStream_t s;
    int repeat = 1;      // debug option "repeat": issue every (idempotent) launch this many times
    template <typename F>
    void operator()(const std::string& layer, const char* kind, const std::string& kernel, double bytes, double flops,
                    F&& fn) {
        if (!rec) {
            for (int i = 0; i < repeat; ++i) fn();
            return;
        }
        if (rec->first_pass) {
            LaunchRecorder::Entry e;
            e.layer = layer;
            e.kind = kind;
            e.kernel = kernel;
            e.bytes = bytes;
            e.flops = flops;
            WHENET_HIP_CHECK(hipEventCreate(&e.stop));
            rec->entries.push_back(e);
        }
        LaunchRecorder::Entry& e = rec->entries.at(rec->cursor++);
        fn();
        WHENET_HIP_CHECK(hipEventRecord(e.stop, s));     // ONE event between consecutive launches
    }
};

}  // namespace

// Block 1's project (linear) and block 2's expand are one affine map of block 1's gated depthwise output
// (snapshot.cpp builds its weights): with front2.hip on block 2, block 1 stops after its squeeze-excite and block 2's
// front kernel reads the 112 x 112 x 32 depthwise output directly, scaling its copy of the weights by the crop's gate.
// One launch and 77 MB of HBM traffic per 64 crops less; block 1's 16-channel output no longer exists (nothing else
// reads it: block 2 has no skip).
bool Engine::fold12_active() const {
    return fold12_ && dtype_ == WHENET_F16 && fuse_front_ && pw_impl_ == 0 && blocks_.size() >= 2 &&
           (front_impl_ == 2 || (front_impl_ == 1 && blocks_[1].f2_preferred));
}

void* Engine::enqueue_blocks(int first, int last, const View& v, void* cur, int n, hipStream_t s, LaunchRecorder* rec) {
    const bool fold = fold12_active() && first <= 1 && last >= 2;
    for (int i = first; i <= last; ++i) {
        void* nxt = (cur == v.x0) ? v.x1 : v.x0;
        enqueue_block(blocks_[size_t(i - 1)], v, cur, nxt, n, s, rec, fold && i <= 2 ? i : 0);
        cur = nxt;
    }
    return cur;
}

void Engine::enqueue_block(const DevBlock& b, const View& v, const void* in, void* out, int n, hipStream_t s,
                           LaunchRecorder* rec, int fold) {
    Rec R{rec, s, repeat_};
    const BlockSpec& sp = b.spec;
    const std::string p = "b" + std::to_string(sp.index);
    const double es = double(esz());
    const int hw_in = sp.h_in * sp.h_in, hw_out = sp.h_out * sp.h_out;
    const int cexp = sp.cexp();
    const void* dw_in = in;
    int se_ntiles = b.dw.plan.ntiles();
    const bool fused = fuse_front_ && sp.has_expand() && pw_impl_ == 0;
    bool se_in_front = false;
    int se_chunks = b.fplan.chunks;
    const bool use_f2 = fused && dtype_ == WHENET_F16 && (front_impl_ == 2 || (front_impl_ == 1 && b.f2_preferred));
    if (use_f2) {
        Front2Args a{};
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wdt = b.dw.wt;
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        se_in_front = true;                  // the SE reduce conv is applied by the front kernel to its channel sums
        a.w1t = b.se.w1t;
        a.R = b.se.R;
        a.k = sp.k;
        a.s = sp.s;
        a.H = sp.h_in;
        a.Ho = sp.h_out;
        a.Cin = sp.cin;
        a.Cexp = cexp;
        a.pad = sp.pad_before();
        a.KSe = b.expand.KS;
        a.NTe = b.expand.NTILES;
        if (fold == 2) {                     // input = block 1's depthwise output, gated; weights = project1 x expand2
            a.wep = d_fold12_w32_;            // (f32: scaled by the gate, then rounded once -- front2.hip)
            a.be = fold12_pw_.bias;
            a.Cin = fold12_pw_.K;
            a.KSe = fold12_pw_.KS;
            a.NTe = fold12_pw_.NTILES;
            a.in_gate = static_cast<const float*>(v.gate);
        }
        a.n = n;
        a.plan = b.f2plan;
        a.plan.threads = front2_threads(b.f2plan, n);
        se_ntiles = b.f2plan.ntiles();
        se_chunks = b.f2plan.chunks;
        R(p + "/front", "front", kernel_name_front2(sp.k, sp.s, a.KSe, a.plan.threads, a.plan.xs).c_str(),
          double(n) * (hw_in * a.Cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * a.Cin * cexp + double(hw_out) * sp.k * sp.k * cexp), [&] { launch_front2(a, s); });
    } else if (fused) {
        FrontArgs a{};
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wd = b.dw.w;
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        se_in_front = true;                  // the SE reduce conv is applied by the front kernel to its channel sums
        a.w1t = b.se.w1t;
        a.R = b.se.R;
        a.k = sp.k;
        a.s = sp.s;
        a.H = sp.h_in;
        a.Ho = sp.h_out;
        a.Cin = sp.cin;
        a.Cexp = cexp;
        a.pad = sp.pad_before();
        a.KSe = b.expand.KS;
        a.NTe = b.expand.NTILES;
        a.n = n;
        a.plan = b.fplan;
        a.plan.threads = front_threads(b.fplan, n);
        se_ntiles = b.fplan.ntiles();
        R(p + "/front", "front", kernel_name_front(dtype_, sp.k, sp.s, a.plan.threads).c_str(), double(n) * (hw_in * sp.cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * sp.cin * cexp + double(hw_out) * sp.k * sp.k * cexp),
          [&] { launch_front(a, dtype_, s); });
    } else if (sp.has_expand()) {
        PwArgs a{};
        a.a = in;
        a.wp = b.expand.wp;
        a.wdense = b.expand.wdense;
        a.bias = b.expand.bias;
        a.out = v.e;
        a.M = n * hw_in;
        a.K = b.expand.K;
        a.N = b.expand.N;
        a.KS = b.expand.KS;
        a.NTILES = b.expand.NTILES;
        a.HW = hw_in;
        a.act = ACT_SWISH;
        R(p + "/expand", "pw", kernel_name_pw(a, dtype_, pw_impl_, num_cus_).c_str(), double(a.M) * (a.K + a.N) * es,
          2.0 * a.M * a.K * a.N, [&] { launch_pw(a, dtype_, pw_impl_, num_cus_, s); });
        dw_in = v.e;
    }
    if (!fused) {
        DwArgs a{};
        a.in = dw_in;
        a.out = fold == 1 ? out : v.d;
        a.w = b.dw.w;
        a.bias = b.dw.bias;
        a.partial = v.partial;
        a.k = sp.k;
        a.s = sp.s;
        a.H = sp.h_in;
        a.Ho = sp.h_out;
        a.C = cexp;
        a.pad = sp.pad_before();
        a.n = n;
        a.plan = b.dw.plan;
        R(p + "/dw", "dw", kernel_name_dw(dtype_, sp.k, sp.s), double(n) * (hw_in + hw_out) * cexp * es,
          2.0 * n * hw_out * sp.k * sp.k * cexp, [&] { launch_dw(a, dtype_, s); });
    }
    // Second half of the SEBlock.  Where it pays, the project GEMM's workgroups compute the gate of their own rows'
    // crops from the front kernel's partial vectors in their prologue (se_device.h; no launch, the gate never reaches
    // HBM); otherwise a stand-alone launch writes the gate.  Measured per block at 64 crops (tools/ab_layers.sh,
    // se_fuse=2 against 0): every workgroup has to pull the WHOLE excite kernel (K x R floats) and all partial vectors
    // of its crops, so the prologue costs +1 us for blocks 4-6 (<= 12 KB, <= 20 vectors: a 7-8 us launch saved), +7 /
    // +12 us for blocks 3 / 2 (40 / 48 partial vectors: three dependent round trips), +4..9 us for the 14x14 blocks
    // (38-75 KB: break-even) and +24 us for the 7x7 blocks (221 KB per workgroup at ~50 GB/s per CU).  The bits are
    // the same either way.  Option se_fuse: 0 = never, 1 = where it pays (default), 2 = every fused-front block.
    const int se_np = se_ntiles * se_chunks;
    const bool se_pays = b.project.K < 320 && se_np <= 24;
    WHENET_REQUIRE(fold == 0 || (fold == 1 && !fused && sp.index == 1) || (fold == 2 && use_f2 && sp.index == 2), WHENET_EINVAL,
                   "fold12: block outside the folded pair");
    const bool se_fused = se_in_front && pw_impl_ == 0 && (se_fuse_ == 2 || (se_fuse_ == 1 && se_pays));
    SeFuse sef{};
    if (se_fused) {
        sef.rpart = v.partial;
        sef.b1 = b.se.b1;
        sef.w2c = b.se.w2c;
        sef.b2 = b.se.b2;
        sef.np = se_ntiles * se_chunks;
        sef.R = b.se.R;
        sef.RP = se_padded_r(b.se.R);
        sef.inv_hw = 1.0f / float(hw_out);
    } else if (se_in_front) {
        // the front kernel already applied se_reduce to its channel sums (v.partial holds the
        // (tiles x chunks) partial vectors of every crop): finish the SEBlock
        SeExciteArgs a{};
        a.rpart = v.partial;
        a.np = se_ntiles * se_chunks;
        a.inv_hw = 1.0f / float(hw_out);
        a.b1 = b.se.b1;
        a.w2c = b.se.w2c;
        a.b2 = b.se.b2;
        a.gate = v.gate;
        a.gate_f16 = dtype_ == WHENET_F16;
        a.C = b.se.C;
        a.R = b.se.R;
        a.n = n;
        R(p + "/se", "se", ("whenet_se_excite_kernel<" + std::to_string(se_padded_r(a.R)) + ">").c_str(),
          double(n) * (a.np * se_padded_r(a.R) + a.C) * 4.0 + double(a.C) * se_padded_r(a.R) * 4.0, 2.0 * n * a.C * a.R,
          [&] { launch_se_excite(a, s); });
    } else {
        SeArgs a{};
        a.partial = v.partial;
        a.ntiles = se_ntiles;
        a.inv_hw = 1.0f / float(hw_out);
        a.w1t = b.se.w1t;
        a.b1 = b.se.b1;
        a.w2c = b.se.w2c;
        a.b2 = b.se.b2;
        a.gate = v.gate;
        a.gate_f16 = dtype_ == WHENET_F16 && fold != 1;      // (fold12: block 2 scales f32 weights by an f32 gate)
        a.C = b.se.C;
        a.R = b.se.R;
        a.n = n;
        R(p + "/se", "se", ("whenet_se_kernel<" + std::to_string(se_padded_r(a.R)) + ">").c_str(), double(n) * (a.ntiles + 1) * a.C * 4.0 + 2.0 * a.C * a.R * 4.0,
          4.0 * n * a.C * a.R, [&] { launch_se(a, s); });
    }
    if (fold != 1) {
        PwArgs a{};
        a.a = v.d;
        a.wp = b.project.wp;
        a.wdense = b.project.wdense;
        a.bias = b.project.bias;
        a.gate = se_fused ? nullptr : v.gate;
        a.se = sef;
        a.res = sp.has_skip() ? in : nullptr;
        a.out = out;
        a.M = n * hw_out;
        a.K = b.project.K;
        a.N = b.project.N;
        a.KS = b.project.KS;
        a.NTILES = b.project.NTILES;
        a.HW = hw_out;
        a.act = ACT_NONE;
        R(p + "/project", "pw", kernel_name_pw(a, dtype_, pw_impl_, num_cus_).c_str(),
          double(a.M) * (a.K + a.N + (sp.has_skip() ? a.N : 0)) * es, 2.0 * a.M * a.K * a.N,
          [&] { launch_pw(a, dtype_, pw_impl_, num_cus_, s); });
    }
}

void Engine::enqueue_forward(const View& v, const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits,
                             hipStream_t s, LaunchRecorder* rec, const float* d_in_f32) {
    Rec R{rec, s, repeat_};
    const double es = double(esz());
    {
        StemArgs a{d_in, v.x0, d_stem_w_, d_stem_b_, d_lut_, n};
        a.in_f32 = d_in_f32;
        R("stem", "stem", kernel_name_stem(dtype_), double(n) * (IN_BYTES + X_ELEMS * es), 2.0 * n * 10838016.0,
          [&] { launch_stem(a, dtype_, s); });
    }
    void* cur = enqueue_blocks(1, int(blocks_.size()), v, v.x0, n, s, rec);
    {
        PwArgs a{};
        a.a = cur;
        a.wp = head_.wp;
        a.wdense = head_.wdense;
        a.bias = head_.bias;
        a.out = v.hc;
        a.M = n * 49;
        a.K = head_.K;
        a.N = head_.N;
        a.KS = head_.KS;
        a.NTILES = head_.NTILES;
        a.HW = 49;
        a.act = ACT_SWISH;
        R("head", "pw", kernel_name_pw(a, dtype_, pw_impl_, num_cus_).c_str(), double(a.M) * (a.K + a.N) * es,
          2.0 * a.M * a.K * a.N, [&] { launch_pw(a, dtype_, pw_impl_, num_cus_, s); });
    }
    {
        HeadsArgs a{};
        a.x = v.hc;
        a.w = d_dense_w_;
        a.b = d_dense_b_;
        a.logits = d_logits;
        a.ypr = d_ypr;
        a.argmax = d_amax;
        a.n = n;
        // (v.partial is free here: the last squeeze-excite is long done)
        const bool split = split_heads_ && partial_per_crop_ >= size_t(heads_split()) * N_LOGITS;
        const std::string hname = std::string(split ? "whenet_heads_split_kernel<" : "whenet_heads_kernel<") +
                                  (dtype_ == WHENET_F16 ? "_Float16>" : "float>");
        R("heads", "heads", hname.c_str(),
          double(n) * (HC_ELEMS * es + (N_LOGITS + 6) * 4.0) + double(FEAT) * N_LOGITS * 4.0, 2.0 * n * FEAT * N_LOGITS,
          [&] {
              if (split) launch_heads_split(a, v.partial, v.hcount, dtype_, s);
              else launch_heads(a, dtype_, s);
          });
    }
}

hipStream_t Engine::lane_stream(int i) {
    while (int(lane_streams_.size()) <= i) {
        hipStream_t st = nullptr;
        WHENET_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        lane_streams_.push_back(st);
    }
    return lane_streams_[size_t(i)];
}

// Drop the on-demand streams (and the graphs that reference them): called before a handle creates
// more engines, so that the new engines' main streams are placed on hardware queues as on a fresh
// process instead of behind this engine's idle lane / copy streams.
void Engine::release_aux_streams() {
    DeviceGuard guard(device_);
    sync();
    drop_graphs();
    for (hipStream_t st : lane_streams_) (void)hipStreamDestroy(st);
    lane_streams_.clear();
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    copy_stream_ = nullptr;
}

hipStream_t Engine::copy_stream() {
    if (!copy_stream_) WHENET_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    return copy_stream_;
}

Engine::View Engine::view(int crop_off) const {
    const size_t o = size_t(crop_off), es = esz();
    View v;
    v.x0 = static_cast<char*>(x0_) + o * X_ELEMS * es;
    v.x1 = static_cast<char*>(x1_) + o * X_ELEMS * es;
    v.e = static_cast<char*>(e_) + o * E_ELEMS * es;
    v.d = static_cast<char*>(d_) + o * D_ELEMS * es;
    v.hc = static_cast<char*>(hc_) + o * HC_ELEMS * es;
    v.partial = partial_ + o * partial_per_crop_;
    v.hcount = hcount_ + o;
    v.gate = gate_ + o * 1152;
    return v;
}

// One forward = up to `lanes_` independent sub-batches, each a 51-kernel chain on its own
// stream (forked from / joined back into `s` with events, so that under capture they become
// parallel branches of ONE graph).  Crops are independent, so the split changes nothing in the
// results; what it buys is overlap: most kernels of this network are short (10-30 us) and
// latency-bound, and two chains in flight fill each other's launch / drain bubbles.
void Engine::enqueue_lanes(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s) {
    int lanes = lanes_;
    while (lanes > 1 && n / lanes < min_lane_crops_) --lanes;
    if (lanes <= 1) {
        enqueue_forward(view(0), d_in, n, d_ypr, d_amax, d_logits, s, nullptr);
        return;
    }
    WHENET_HIP_CHECK(hipEventRecord(fork_ev_, s));
    int off = 0;
    for (int i = 0; i < lanes; ++i) {
        const int cnt = n / lanes + (i < n % lanes ? 1 : 0);
        hipStream_t st = (i == 0) ? s : lane_stream(i - 1);
        if (i > 0) WHENET_HIP_CHECK(hipStreamWaitEvent(st, fork_ev_, 0));
        enqueue_forward(view(off), d_in + size_t(off) * IN_BYTES, cnt, d_ypr + size_t(off) * 3,
                        d_amax ? d_amax + size_t(off) * 3 : nullptr, d_logits ? d_logits + size_t(off) * N_LOGITS : nullptr,
                        st, nullptr);
        if (i > 0) {
            WHENET_HIP_CHECK(hipEventRecord(join_ev_[size_t(i - 1)], st));
            WHENET_HIP_CHECK(hipStreamWaitEvent(s, join_ev_[size_t(i - 1)], 0));
        }
        off += cnt;
    }
}

// Capture fn's launches on stream s into an executable graph (cached under key) and return it.
template <typename F>
hipGraphExec_t Engine::cached_graph(const GraphKey& key, hipStream_t s, F&& fn) {
    auto it = graphs_.find(key);
    if (it != graphs_.end()) return it->second;
    if (graphs_.size() >= size_t(MAX_GRAPHS)) {
        sync_streams(s);                 // (s may be a lane stream: the main stream's graphs are in flight too)
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        drop_graphs();
    }
    hipGraph_t graph = nullptr;
    WHENET_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    try {
        fn();
    } catch (...) {
        (void)hipStreamEndCapture(s, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    WHENET_HIP_CHECK(hipStreamEndCapture(s, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) throw Error(WHENET_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    graphs_.emplace(key, exec);
    return exec;
}

void Engine::run_forward(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s) {
    if (poison_) {
        // debug option "poison": every activation buffer holds NaN bit patterns when the forward starts, so a kernel
        // that reads what no kernel of THIS forward wrote shows up in the results (tests/test_gpu_parity.py)
        const size_t N = size_t(n), es = esz();
        WHENET_HIP_CHECK(hipMemsetAsync(x0_, 0xff, N * X_ELEMS * es, s));
        WHENET_HIP_CHECK(hipMemsetAsync(x1_, 0xff, N * X_ELEMS * es, s));
        WHENET_HIP_CHECK(hipMemsetAsync(e_, 0xff, N * E_ELEMS * es, s));
        WHENET_HIP_CHECK(hipMemsetAsync(d_, 0xff, N * D_ELEMS * es, s));
        WHENET_HIP_CHECK(hipMemsetAsync(hc_, 0xff, N * HC_ELEMS * es, s));
        WHENET_HIP_CHECK(hipMemsetAsync(partial_, 0xff, N * partial_per_crop_ * sizeof(float), s));
        WHENET_HIP_CHECK(hipMemsetAsync(gate_, 0xff, N * 1152 * sizeof(float), s));
    }
    if (!use_graph_) {
        enqueue_lanes(d_in, n, d_ypr, d_amax, d_logits, s);
        return;
    }
    int lanes = lanes_;
    while (lanes > 1 && n / lanes < min_lane_crops_) --lanes;
    if (lanes > 1) (void)lane_stream(lanes - 2);        // the lane streams exist before any capture starts
    if (lane_graphs_ && lanes > 1) {
        // One graph PER LANE, each launched on its own stream: the chains then run as independent queues.
        // (Branches of a single graph cost ~5 us per edge on this runtime, which eats the overlap.)
        WHENET_HIP_CHECK(hipEventRecord(fork_ev_, s));
        int off = 0;
        for (int i = 0; i < lanes; ++i) {
            const int cnt = n / lanes + (i < n % lanes ? 1 : 0);
            hipStream_t st = (i == 0) ? s : lane_stream(i - 1);
            const uint8_t* in_i = d_in + size_t(off) * IN_BYTES;
            float* ypr_i = d_ypr + size_t(off) * 3;
            int32_t* am_i = d_amax ? d_amax + size_t(off) * 3 : nullptr;
            float* lg_i = d_logits ? d_logits + size_t(off) * N_LOGITS : nullptr;
            hipGraphExec_t g = cached_graph(GraphKey{cnt, off, in_i, ypr_i, am_i, lg_i}, st, [&] {
                enqueue_forward(view(off), in_i, cnt, ypr_i, am_i, lg_i, st, nullptr);
            });
            if (i > 0) WHENET_HIP_CHECK(hipStreamWaitEvent(st, fork_ev_, 0));
            WHENET_HIP_CHECK(hipGraphLaunch(g, st));
            if (i > 0) {
                WHENET_HIP_CHECK(hipEventRecord(join_ev_[size_t(i - 1)], st));
                WHENET_HIP_CHECK(hipStreamWaitEvent(s, join_ev_[size_t(i - 1)], 0));
            }
            off += cnt;
        }
        return;
    }
    hipGraphExec_t g = cached_graph(GraphKey{n, -1, d_in, d_ypr, d_amax, d_logits}, s,
                                    [&] { enqueue_lanes(d_in, n, d_ypr, d_amax, d_logits, s); });
    WHENET_HIP_CHECK(hipGraphLaunch(g, s));
}

// ------------------------------------------------------------------------------------------
// public paths
// ------------------------------------------------------------------------------------------
void Engine::forward_device(const uint8_t* d_crops, int n, float* d_ypr, int32_t* d_argmax, float* d_logits,
                            hipStream_t stream) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(d_crops != nullptr && d_ypr != nullptr, WHENET_EINVAL, "crops and ypr must not be NULL");
    WHENET_REQUIRE((reinterpret_cast<uintptr_t>(d_crops) & 3) == 0, WHENET_EINVAL, "crops must be 4-byte aligned");
    ensure_capacity(n);
    run_forward(d_crops, n, d_ypr, d_argmax, d_logits, stream ? stream : stream_);
}

void Engine::forward_host(const uint8_t* crops, int n, float* ypr, int32_t* argmax, float* logits) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(crops != nullptr && ypr != nullptr, WHENET_EINVAL, "crops and ypr must not be NULL");
    ensure_capacity(n);
    const size_t N = size_t(n);
    WHENET_HIP_CHECK(hipMemcpyAsync(in_u8_, crops, N * IN_BYTES, hipMemcpyHostToDevice, stream_));
    run_forward(in_u8_, n, o_ypr_, o_amax_, o_logits_, stream_);
    WHENET_HIP_CHECK(hipMemcpyAsync(ypr, o_ypr_, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(argmax, o_amax_, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    if (logits)
        WHENET_HIP_CHECK(hipMemcpyAsync(logits, o_logits_, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

// Model.predict on the NORMALISED float32 image (whenet.py:27) + decode: the path for real-valued
// crops (whenet.py:25 divides whatever dtype it is given by 255), which the byte LUT cannot serve.
// Eager launches (no graph): a compatibility path, not the one bench.py times.
void Engine::forward_host_f32(const float* x, int n, float* ypr, int32_t* argmax, float* logits) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(x != nullptr && ypr != nullptr, WHENET_EINVAL, "image and ypr must not be NULL");
    ensure_capacity(n);
    const size_t N = size_t(n);
    if (n > in_f32_cap_) {
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        if (in_f32_) (void)hipFree(in_f32_);
        in_f32_ = nullptr;
        in_f32_cap_ = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&in_f32_), N * IN_BYTES * sizeof(float));
        if (e != hipSuccess) throw Error(WHENET_ENOMEM, std::string("float input buffer: ") + hipGetErrorString(e));
        in_f32_cap_ = n;
    }
    WHENET_HIP_CHECK(hipMemcpyAsync(in_f32_, x, N * IN_BYTES * sizeof(float), hipMemcpyHostToDevice, stream_));
    enqueue_forward(view(0), nullptr, n, o_ypr_, o_amax_, o_logits_, stream_, nullptr, in_f32_);
    WHENET_HIP_CHECK(hipMemcpyAsync(ypr, o_ypr_, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(argmax, o_amax_, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    if (logits)
        WHENET_HIP_CHECK(hipMemcpyAsync(logits, o_logits_, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::sync_streams(hipStream_t s) {
    WHENET_HIP_CHECK(hipStreamSynchronize(s));
    for (hipStream_t st : lane_streams_) WHENET_HIP_CHECK(hipStreamSynchronize(st));
}

void Engine::sync() {
    DeviceGuard guard(device_);
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    if (copy_stream_) WHENET_HIP_CHECK(hipStreamSynchronize(copy_stream_));
    for (hipStream_t st : lane_streams_) WHENET_HIP_CHECK(hipStreamSynchronize(st));
}

void Engine::ensure_slot(Slot& s, int n) {
    if (!s.copied) {
        WHENET_HIP_CHECK(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        WHENET_HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (n <= s.capacity) return;
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_ypr) (void)hipHostFree(s.h_ypr);
    if (s.h_amax) (void)hipHostFree(s.h_amax);
    if (s.h_logits) (void)hipHostFree(s.h_logits);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_ypr) (void)hipFree(s.d_ypr);
    if (s.d_amax) (void)hipFree(s.d_amax);
    if (s.d_logits) (void)hipFree(s.d_logits);
    s.h_in = nullptr; s.h_ypr = nullptr; s.h_amax = nullptr; s.h_logits = nullptr;
    s.d_in = nullptr; s.d_ypr = nullptr; s.d_amax = nullptr; s.d_logits = nullptr;
    s.capacity = 0;
    const size_t N = size_t(n);
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_in), N * IN_BYTES, hipHostMallocDefault));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_ypr), N * 3 * sizeof(float), hipHostMallocDefault));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_amax), N * 3 * sizeof(int32_t), hipHostMallocDefault));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_logits), N * N_LOGITS * sizeof(float), hipHostMallocDefault));
    WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_in), N * IN_BYTES));
    WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_ypr), N * 3 * sizeof(float)));
    WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_amax), N * 3 * sizeof(int32_t)));
    WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_logits), N * N_LOGITS * sizeof(float)));
    s.capacity = n;
}

int Engine::submit(const uint8_t* crops, int n) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(crops != nullptr, WHENET_EINVAL, "crops must not be NULL");
    Slot* slot = nullptr;
    for (Slot& s : slots_)
        if (!s.busy) {
            slot = &s;
            break;
        }
    WHENET_REQUIRE(slot != nullptr, WHENET_EINVAL, "too many submissions in flight (collect one first)");
    ensure_capacity(n);
    ensure_slot(*slot, n);
    const size_t N = size_t(n);
    std::memcpy(slot->h_in, crops, N * IN_BYTES);
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->d_in, slot->h_in, N * IN_BYTES, hipMemcpyHostToDevice, copy_stream()));
    WHENET_HIP_CHECK(hipEventRecord(slot->copied, copy_stream()));
    WHENET_HIP_CHECK(hipStreamWaitEvent(stream_, slot->copied, 0));
    run_forward(slot->d_in, n, slot->d_ypr, slot->d_amax, slot->d_logits, stream_);
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_ypr, slot->d_ypr, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_amax, slot->d_amax, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_logits, slot->d_logits, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipEventRecord(slot->done, stream_));
    slot->busy = true;
    slot->n = n;
    slot->ticket = next_ticket_++;
    return slot->ticket;
}

void Engine::collect(int ticket, float* ypr, int32_t* argmax, float* logits) {
    DeviceGuard guard(device_);
    Slot* slot = nullptr;
    for (Slot& s : slots_)
        if (s.busy && s.ticket == ticket) slot = &s;
    WHENET_REQUIRE(slot != nullptr, WHENET_EINVAL, "unknown or already collected ticket " + std::to_string(ticket));
    WHENET_HIP_CHECK(hipEventSynchronize(slot->done));
    const size_t N = size_t(slot->n);
    if (ypr) std::memcpy(ypr, slot->h_ypr, N * 3 * sizeof(float));
    if (argmax) std::memcpy(argmax, slot->h_amax, N * 3 * sizeof(int32_t));
    if (logits) std::memcpy(logits, slot->h_logits, N * N_LOGITS * sizeof(float));
    slot->busy = false;
}

Engine::Slot* Engine::free_slot() {
    for (Slot& s : slots_)
        if (!s.busy) return &s;
    throw Error(WHENET_EINVAL, "too many submissions in flight (collect one first)");
}

void Engine::ensure_slot_frame(Slot& s, size_t frame_bytes, int k) {
    if (frame_bytes > s.frame_cap) {
        if (s.h_frame) (void)hipHostFree(s.h_frame);
        if (s.d_frame) (void)hipFree(s.d_frame);
        s.h_frame = nullptr; s.d_frame = nullptr; s.frame_cap = 0;
        WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_frame), frame_bytes, hipHostMallocDefault));
        WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_frame), frame_bytes));
        s.frame_cap = frame_bytes;
    }
    if (k > s.plan_cap) {
        if (s.h_plan) (void)hipHostFree(s.h_plan);
        if (s.d_plan) (void)hipFree(s.d_plan);
        s.h_plan = nullptr; s.d_plan = nullptr; s.plan_cap = 0;
        const size_t bytes = size_t(k) * CROP_PLAN_INTS * sizeof(int32_t);
        WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_plan), bytes, hipHostMallocDefault));
        WHENET_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_plan), bytes));
        s.plan_cap = k;
    }
}

namespace {
void check_rects(int fh, int fw, const int32_t* rects, int k) {
    for (int i = 0; i < k; ++i) {
        const int32_t* r = rects + 4 * i;
        WHENET_REQUIRE(r[0] >= 0 && r[1] >= 0 && r[2] <= fh && r[3] <= fw && r[0] < r[2] && r[1] < r[3], WHENET_EINVAL,
                       "crop window " + std::to_string(i) + " is empty or outside the frame");
    }
}
}  // namespace

// One frame of demo_video.py:49-58 as ONE submission: the frame crosses PCIe once; every head is
// cropped / colour-swapped / resized on the device (frame.hip) straight into the forward's input.
int Engine::submit_frame(const uint8_t* frame, int fh, int fw, int swap_rb, const int32_t* rects, int k) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(frame != nullptr && fh > 0 && fw > 0 && k >= 0 && (k == 0 || rects != nullptr), WHENET_EINVAL,
                   "submit_frame: bad arguments");
    check_rects(fh, fw, rects, k);
    Slot* slot = free_slot();
    if (k > 0) {
        ensure_capacity(k);
        ensure_slot(*slot, k);
        const size_t fbytes = size_t(fh) * fw * 3;
        ensure_slot_frame(*slot, fbytes, k);
        std::memcpy(slot->h_frame, frame, fbytes);
        for (int i = 0; i < k; ++i) build_crop_plan(rects + 4 * i, slot->h_plan + size_t(i) * CROP_PLAN_INTS);
        const size_t N = size_t(k);
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->d_frame, slot->h_frame, fbytes, hipMemcpyHostToDevice, copy_stream()));
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->d_plan, slot->h_plan, N * CROP_PLAN_INTS * sizeof(int32_t),
                                        hipMemcpyHostToDevice, copy_stream()));
        WHENET_HIP_CHECK(hipEventRecord(slot->copied, copy_stream()));
        WHENET_HIP_CHECK(hipStreamWaitEvent(stream_, slot->copied, 0));
        launch_crop_resize(slot->d_frame, fw, swap_rb, slot->d_plan, k, slot->d_in, stream_);
        run_forward(slot->d_in, k, slot->d_ypr, slot->d_amax, slot->d_logits, stream_);
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_ypr, slot->d_ypr, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_amax, slot->d_amax, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_logits, slot->d_logits, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    } else {
        ensure_slot(*slot, 1);
    }
    WHENET_HIP_CHECK(hipEventRecord(slot->done, stream_));
    slot->busy = true;
    slot->n = k;
    slot->ticket = next_ticket_++;
    return slot->ticket;
}

void Engine::op_crop_resize(const uint8_t* frame, int fh, int fw, int swap_rb, const int32_t* rects, int k,
                            uint8_t* crops_out) {
    DeviceGuard guard(device_);
    WHENET_REQUIRE(frame != nullptr && rects != nullptr && crops_out != nullptr && fh > 0 && fw > 0 && k > 0,
                   WHENET_EINVAL, "op_crop_resize: bad arguments");
    check_rects(fh, fw, rects, k);
    std::vector<int32_t> plan(size_t(k) * CROP_PLAN_INTS);
    for (int i = 0; i < k; ++i) build_crop_plan(rects + 4 * i, plan.data() + size_t(i) * CROP_PLAN_INTS);
    const size_t fbytes = size_t(fh) * fw * 3, obytes = size_t(k) * IN_BYTES;
    uint8_t* d_frame = static_cast<uint8_t*>(dev_alloc(fbytes));
    int32_t* d_plan = static_cast<int32_t*>(dev_alloc(plan.size() * sizeof(int32_t)));
    uint8_t* d_out = static_cast<uint8_t*>(dev_alloc(obytes));
    try {
        WHENET_HIP_CHECK(hipMemcpy(d_frame, frame, fbytes, hipMemcpyHostToDevice));
        WHENET_HIP_CHECK(hipMemcpy(d_plan, plan.data(), plan.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        launch_crop_resize(d_frame, fw, swap_rb, d_plan, k, d_out, stream_);
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        WHENET_HIP_CHECK(hipMemcpy(crops_out, d_out, obytes, hipMemcpyDeviceToHost));
    } catch (...) {
        dev_free(d_frame); dev_free(d_plan); dev_free(d_out);
        throw;
    }
    dev_free(d_frame); dev_free(d_plan); dev_free(d_out);
}

// yolo_eval (yolo_v3/model.py:193-232) on host feature maps: H2D, decode + NMS on the device, the selected boxes
// back, concatenated class by class like the reference.  Returns the number of detections.
int Engine::yolo_eval(const float* const* feats, const int* grid_h, const int* grid_w, int num_layers,
                      const float* anchors, int num_anchors, int num_classes, float image_h, float image_w,
                      float score_threshold, float iou_threshold, int max_boxes, float* boxes, float* scores,
                      int32_t* classes, int32_t* index, float* all_boxes, float* all_scores) {
    DeviceGuard guard(device_);
    WHENET_REQUIRE(feats && grid_h && grid_w && anchors && boxes && scores && classes, WHENET_EINVAL,
                   "yolo_eval: NULL argument");
    WHENET_REQUIRE((num_layers == 3 && num_anchors == 9) || (num_layers == 2 && num_anchors == 6), WHENET_EINVAL,
                   "yolo_eval: 3 maps with 9 anchors or 2 maps with 6 (model.py:203)");
    WHENET_REQUIRE(num_classes >= 1 && num_classes <= 1024 && image_h > 0 && image_w > 0, WHENET_EINVAL,
                   "yolo_eval: bad num_classes / image shape");
    WHENET_REQUIRE(max_boxes >= 1, WHENET_EINVAL, "yolo_eval: max_boxes must be >= 1");      // (any value, as model.py:193)
    // model.py:203: anchor_mask = [[6,7,8],[3,4,5],[0,1,2]] for 3 maps, [[3,4,5],[1,2,3]] for 2
    static const int ANCHOR_MASK3[3][3] = {{6, 7, 8}, {3, 4, 5}, {0, 1, 2}};
    static const int ANCHOR_MASK2[2][3] = {{3, 4, 5}, {1, 2, 3}};
    YoloArgs a{};
    a.num_layers = num_layers;
    a.num_classes = num_classes;
    a.na = 3;
    a.input_h = float(grid_h[0] * 32);                    // model.py:204
    a.input_w = float(grid_w[0] * 32);
    a.image_h = image_h;
    a.image_w = image_w;
    {   // model.py:158-162, float32 like the graph: new_shape = round(image_shape * min(input_shape / image_shape))
        const float ry = a.input_h / image_h, rx = a.input_w / image_w;
        const float r = ry < rx ? ry : rx;
        const float new_h = std::nearbyintf(image_h * r), new_w = std::nearbyintf(image_w * r);      // half to even
        a.off_y = (a.input_h - new_h) / 2.0f / a.input_h;
        a.off_x = (a.input_w - new_w) / 2.0f / a.input_w;
        a.scale_y = a.input_h / new_h;
        a.scale_x = a.input_w / new_w;
    }
    a.score_thr = score_threshold;
    a.iou_thr = iou_threshold;
    // one engine-owned scratch block, grown on demand (round 2 paid ~10 hipMalloc/hipFree per frame here)
    struct Carver {
        Engine* e;
        size_t used = 0;
        std::vector<std::pair<size_t, size_t>> pieces;           // (offset, bytes)
        size_t add(size_t nbytes) {
            const size_t off = (used + 255) & ~size_t(255);
            used = off + (nbytes ? nbytes : 16);
            return off;
        }
    };
    int N = 0;
    const size_t per = size_t(5 + num_classes) * 3;
    size_t feat_off[3] = {0, 0, 0}, feat_bytes[3] = {0, 0, 0};
    Carver cv{this};
    for (int l = 0; l < num_layers; ++l) {
        WHENET_REQUIRE(feats[l] && grid_h[l] > 0 && grid_w[l] > 0 && grid_h[l] <= 4096 && grid_w[l] <= 4096, WHENET_EINVAL,
                       "yolo_eval: bad feature map");
        YoloLayer& L = a.layer[l];
        L.gh = grid_h[l];
        L.gw = grid_w[l];
        L.first = N;
        for (int k = 0; k < 3; ++k) {
            const int m = (num_layers == 3) ? ANCHOR_MASK3[l][k] : ANCHOR_MASK2[l][k];
            L.anchor[k][0] = anchors[2 * m];
            L.anchor[k][1] = anchors[2 * m + 1];
        }
        feat_bytes[l] = size_t(L.gh) * L.gw * per * sizeof(float);
        feat_off[l] = cv.add(feat_bytes[l]);
        N += L.gh * L.gw * 3;
    }
    a.N = N;
    a.NP = 1;
    while (a.NP < N) a.NP <<= 1;
    if (max_boxes > N) max_boxes = N;                       // (no more selections than boxes)
    a.max_boxes = max_boxes;
    const size_t C = size_t(num_classes), MB = size_t(max_boxes);
    const size_t o_boxes = cv.add(size_t(N) * 4 * sizeof(float));
    const size_t o_all = all_scores ? cv.add(size_t(N) * C * sizeof(float)) : 0;
    const size_t o_counts = cv.add(C * sizeof(int));
    const size_t o_keys = cv.add(C * size_t(a.NP) * sizeof(unsigned long long));
    const size_t o_ob = cv.add(C * MB * 4 * sizeof(float));
    const size_t o_os = cv.add(C * MB * sizeof(float));
    const size_t o_oi = cv.add(C * MB * sizeof(int));
    const size_t o_oc = cv.add(C * sizeof(int));
    if (cv.used > yolo_scratch_bytes_) {
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        if (yolo_scratch_) (void)hipFree(yolo_scratch_);
        yolo_scratch_ = nullptr;
        yolo_scratch_bytes_ = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&yolo_scratch_), cv.used);
        if (e != hipSuccess) throw Error(WHENET_ENOMEM, std::string("yolo_eval scratch: ") + hipGetErrorString(e));
        yolo_scratch_bytes_ = cv.used;
    }
    unsigned char* base = yolo_scratch_;
    for (int l = 0; l < num_layers; ++l) {
        float* d = reinterpret_cast<float*>(base + feat_off[l]);
        WHENET_HIP_CHECK(hipMemcpyAsync(d, feats[l], feat_bytes[l], hipMemcpyHostToDevice, stream_));
        a.layer[l].feats = d;
    }
    a.boxes = reinterpret_cast<float*>(base + o_boxes);
    a.all_scores = all_scores ? reinterpret_cast<float*>(base + o_all) : nullptr;
    a.counts = reinterpret_cast<int*>(base + o_counts);
    a.keys = reinterpret_cast<unsigned long long*>(base + o_keys);
    a.out_boxes = reinterpret_cast<float*>(base + o_ob);
    a.out_scores = reinterpret_cast<float*>(base + o_os);
    a.out_index = reinterpret_cast<int*>(base + o_oi);
    a.out_count = reinterpret_cast<int*>(base + o_oc);
    launch_yolo_eval(a, stream_);
    std::vector<float> hb(C * MB * 4), hs(C * MB);
    std::vector<int> hi(C * MB), hc(C);
    WHENET_HIP_CHECK(hipMemcpyAsync(hb.data(), a.out_boxes, hb.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(hs.data(), a.out_scores, hs.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(hi.data(), a.out_index, hi.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(hc.data(), a.out_count, hc.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (all_boxes)
        WHENET_HIP_CHECK(hipMemcpyAsync(all_boxes, a.boxes, size_t(N) * 4 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (all_scores)
        WHENET_HIP_CHECK(hipMemcpyAsync(all_scores, a.all_scores, size_t(N) * C * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    int out = 0;                                          // model.py:227-229: concatenated class by class
    for (size_t c = 0; c < C; ++c)
        for (int k = 0; k < hc[c]; ++k, ++out) {
            std::memcpy(boxes + size_t(out) * 4, hb.data() + (c * MB + size_t(k)) * 4, 4 * sizeof(float));
            scores[out] = hs[c * MB + size_t(k)];
            classes[out] = int32_t(c);
            if (index) index[out] = hi[c * MB + size_t(k)];
        }
    return out;
}

// Per-launch timing of the forward AS THE TIMED PATH RUNS IT: the same sub-batch chains on the
// same streams, concurrently, launched eagerly with one HIP event recorded on the chain's stream
// between consecutive kernels (a launch's time = previous event -> its own event, i.e. kernel plus
// the boundary in front of it).  Entries: one per launch of a chain; durations averaged over the
// chains and the iterations; bytes / flops are those of ONE chain's launch (its sub-batch).
int Engine::profile(const uint8_t* d_crops, int n, int iters, whenet_launch_stat_t* stats, int cap) {
    DeviceGuard guard(device_);
    require_model();
    WHENET_REQUIRE(d_crops != nullptr && iters >= 1, WHENET_EINVAL, "profile: bad arguments");
    ensure_capacity(n);
    int lanes = lanes_;
    while (lanes > 1 && n / lanes < min_lane_crops_) --lanes;
    std::vector<LaunchRecorder> recs;
    recs.resize(size_t(lanes));
    struct Cleanup {
        std::vector<LaunchRecorder>& r;
        ~Cleanup() {
            for (auto& lr : r) {
                if (lr.start) (void)hipEventDestroy(lr.start);
                for (auto& e : lr.entries)
                    if (e.stop) (void)hipEventDestroy(e.stop);
            }
        }
    } cleanup{recs};
    for (auto& lr : recs) WHENET_HIP_CHECK(hipEventCreate(&lr.start));
    // one untimed eager pass so that lazy code-object loading does not land in the numbers
    enqueue_forward(view(0), d_crops, n, o_ypr_, o_amax_, o_logits_, stream_, nullptr);
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int it = 0; it < iters; ++it) {
        WHENET_HIP_CHECK(hipEventRecord(fork_ev_, stream_));
        int off = 0;
        for (int i = 0; i < lanes; ++i) {
            const int cnt = n / lanes + (i < n % lanes ? 1 : 0);
            hipStream_t st = (i == 0) ? stream_ : lane_stream(i - 1);
            if (i > 0) WHENET_HIP_CHECK(hipStreamWaitEvent(st, fork_ev_, 0));
            LaunchRecorder& lr = recs[size_t(i)];
            lr.cursor = 0;
            WHENET_HIP_CHECK(hipEventRecord(lr.start, st));
            enqueue_forward(view(off), d_crops + size_t(off) * IN_BYTES, cnt, o_ypr_ + size_t(off) * 3,
                            o_amax_ + size_t(off) * 3, o_logits_ + size_t(off) * N_LOGITS, st, &lr);
            {   // calibration entry: an empty kernel timed the same way = the boundary + event cost
                Rec R{&lr, st, 1};
                R("(boundary)", "calib", "whenet_empty_kernel", 0.0, 0.0, [&] { launch_empty(st); });
            }
            lr.first_pass = false;
            if (i > 0) {
                WHENET_HIP_CHECK(hipEventRecord(join_ev_[size_t(i - 1)], st));
                WHENET_HIP_CHECK(hipStreamWaitEvent(stream_, join_ev_[size_t(i - 1)], 0));
            }
            off += cnt;
        }
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        for (auto& lr : recs) {
            hipEvent_t prev = lr.start;
            for (auto& e : lr.entries) {
                float ms = 0.f;
                WHENET_HIP_CHECK(hipEventElapsedTime(&ms, prev, e.stop));
                e.total_ms += ms;
                prev = e.stop;
            }
        }
    }
    const int count = int(recs[0].entries.size());
    for (int i = 0; i < count && i < cap && stats; ++i) {
        const auto& e = recs[0].entries[size_t(i)];
        double tot = 0;
        for (auto& lr : recs) tot += lr.entries[size_t(i)].total_ms;
        whenet_launch_stat_t& o = stats[i];
        copy_name(o.layer, sizeof(o.layer), e.layer);
        copy_name(o.kind, sizeof(o.kind), e.kind);
        copy_name(o.kernel, sizeof(o.kernel), e.kernel);
        o.avg_us = tot * 1000.0 / (double(iters) * lanes);
        o.alg_bytes = e.bytes;
        o.alg_flops = e.flops;
    }
    return count;
}

// ------------------------------------------------------------------------------------------
// single-stage entry points (tests)
// ------------------------------------------------------------------------------------------
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
    enqueue_block(b, view(0), x0_, x1_, n, stream_, nullptr);
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
    const void* res = enqueue_blocks(first, last, v, v.x0, n, stream_, nullptr);
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
    a.HW = 49;
    a.act = ACT_SWISH;
    launch_pw(a, dtype_, pw_impl_, num_cus_, stream_);
    HeadsArgs h{};
    h.x = hc_;
    h.w = d_dense_w_;
    h.b = d_dense_b_;
    h.feat = d_feat;
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
