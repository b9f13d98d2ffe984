#include "engine_internal.h"

#include <mutex>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>

namespace whenet {

using namespace detail;

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
    if (!h.packed_split.empty()) {
        d.wps = upload_bytes(h.packed_split.data(), h.packed_split.size());
        d.KSs = h.KS_split;
        d.wsi = h.wsi;
    }
    d.wdense = upload(h.dense);
    d.bias = upload(h.bias);
    return d;
}

Engine::Engine(const void* snapshot, size_t nbytes, int device_id, int dtype)
    : device_(device_id), dtype_(dtype == WHENET_F32S ? WHENET_F32 : dtype), split_(dtype == WHENET_F32S) {
    // host-side preparation first: a malformed snapshot is reported as such even on a box
    // without a GPU
    HostModel m = build_host_model(parse_snapshot(snapshot, nbytes), dtype_, split_);
    params_backbone_ = m.params_backbone;
    params_heads_ = m.params_heads;
    n_tensors_ = m.n_tensors;

    open_device(device_id);
    DeviceGuard guard(device_);
    has_model_ = true;

    d_lut_ = static_cast<float*>(upload_bytes(&m.lut[0][0], sizeof(m.lut)));
    d_stem_w_ = upload(m.stem_w);
    d_stem_b_ = upload(m.stem_b);
    {
        StemDwTable t;
        build_stemdw_table(m.stem_w.data(), &m.lut[0][0], &t);
        d_stemdw_tab_ = static_cast<StemDwTable*>(upload_bytes(&t, sizeof(t)));
    }
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
        if (split_ && !hb.se.excite.packed_split.empty()) {
            b.se.w2p = upload_bytes(hb.se.excite.packed_split.data(), hb.se.excite.packed_split.size());
            b.se.KSr = hb.se.excite.KS_split;
            b.se.w2_wsi = hb.se.excite.wsi;
        } else if (dtype_ == WHENET_F16 && !hb.se.excite.packed.empty()) {
            b.se.w2p = upload_bytes(hb.se.excite.packed.data(), hb.se.excite.packed.size());
            b.se.KSr = hb.se.excite.KS;
        }
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
            if (split_ && front2s_supported(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.cin)) {
                // f32s: the same stage with both convolutions on the matrix cores (front2s.hip, round 6)
                b.f2s_supported = true;
                b.f2splan = plan_front2s(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.h_out, hb.dw.C, &b.f2s_tm);
                b.f2s_preferred = front2s_preferred(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.dw.C);
                b.dw.wts = upload(pack_dw_toeplitz_s(hb.dw.w, hb.spec.k, hb.spec.s, hb.dw.C, b.f2s_tm, &b.dw.wts_wsi));
                partial_per_crop_ = std::max(partial_per_crop_, size_t(b.f2splan.ntiles()) * size_t(b.f2splan.chunks) *
                                                                    size_t(se_padded_r(b.se.R)));
            }
            // the 7 x 7 blocks, both dtypes: a group of crops per workgroup (front7.hip); f16: its Toeplitz image is group-aligned
            b.f7_supported = front7_supported(hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.cin);
            if (b.f7_supported) {
                if (dtype_ == WHENET_F16) b.dw.wt7 = upload(pack_dw_toeplitz(hb.dw.w, hb.spec.k, 1, hb.dw.C, 4 - hb.spec.k / 2));
                b.f7_chunks = front7_plan_for(dtype_, hb.spec.cin, hb.dw.C, 64).chunks;
                partial_per_crop_ = std::max(partial_per_crop_, size_t(b.f7_chunks) * size_t(se_padded_r(b.se.R)));
            }
        }
        if (mb7_supported(dtype_, hb.spec.k, hb.spec.s, hb.spec.h_in, hb.spec.cin, hb.dw.C, hb.se.R, hb.spec.cout, hb.spec.has_skip())) {
            // f16, blocks 13-16: the whole block as one launch (mb7.hip, round 6) -- tap sequences and binary16 squeeze-excite kernels
            std::vector<half_t> w1p, w2p;
            pack_mb7_se(hb.se.w1t, hb.se.w2, hb.se.C, hb.se.R, &w1p, &w2p);
            b.mb7.wds = upload(pack_mb7_taps(hb.dw.w, hb.spec.k, hb.dw.C));
            b.mb7.w1p = upload(w1p);
            b.mb7.w2p = upload(w2p);
            b.mb7.ok = true;
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
    auto free_slot_buffers = [](Slot& s) {
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
    };
    for (Slot& s : slots_) free_slot_buffers(s);
    free_slot_buffers(host_slot_);
    if (hout_ypr_) (void)hipHostFree(hout_ypr_);
    if (hout_amax_) (void)hipHostFree(hout_amax_);
    if (hout_logits_) (void)hipHostFree(hout_logits_);
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
        WHENET_REQUIRE(value >= 0 && value <= 3, WHENET_EINVAL,
                       "se_fuse must be 0 (never), 1 (where the prologue form pays), 2 (prologue form on every block) or 3 (1 + the matrix-core form on blocks 7-16, default)");
        se_fuse_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "front_impl") {
        WHENET_REQUIRE(value >= 0 && value <= 2, WHENET_EINVAL,
                       "front_impl must be 0 (front.hip everywhere), 1 (per layer, default) or 2 (front2.hip / front2s.hip wherever they exist)");
        front_impl_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "f2s_mask") {
        // probes: bit i set = block i runs front2s.hip (f32s handles, front_impl = 1); -1 = the measured per-layer table
        WHENET_REQUIRE(value >= -1 && value < (1 << 17), WHENET_EINVAL, "f2s_mask must be -1 or a bit mask of blocks 2..12");
        f2s_mask_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "head_fuse") {
        head_fuse_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "concurrent") {
        xcd_always_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "xcd_map") {
        WHENET_REQUIRE(value >= 0 && value <= 7, WHENET_EINVAL, "xcd_map must be a bit mask 0..7 (1 = front / front2, 2 = front7, 4 = head7)");
        xcd_map_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "mb7") {
        mb7_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "front7") {
        front7_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "fold12") {
        fold12_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "stem_fuse") {
        stem_fuse_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "poison") {
        poison_ = value != 0;
    } else if (key == "lanes" || key == "device_lanes") {
        // "lanes": every forward; "device_lanes" (set by the handle's "inflight" option): the asynchronous device-side
        // entry points only -- a blocking host forward keeps host_lanes_
        WHENET_REQUIRE(value >= 1 && value <= MAX_LANES, WHENET_EINVAL, "lanes must be 1..8");
        lanes_ = int(value);
        if (key == "lanes") host_lanes_ = int(value);
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
    } else if (key == "pw_staged") {
        pw_staged_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "split_pw") {
        WHENET_REQUIRE(split_ || value == 0, WHENET_EINVAL, "split_pw: the handle was not created as WHENET_F32S");
        split_pw_ = value != 0;
        sync();
        drop_graphs();
    } else if (key == "host_pinned_max") {
        WHENET_REQUIRE(value >= 0 && value <= 4096, WHENET_EINVAL, "host_pinned_max must be 0..4096");
        sync();
        host_pinned_max_ = int(value);
    } else if (key == "se_fuse_tiny") {
        WHENET_REQUIRE(value >= 0 && value <= 64, WHENET_EINVAL, "se_fuse_tiny must be 0..64");
        se_fuse_tiny_ = int(value);
        sync();
        drop_graphs();
    } else if (key == "host_lanes") {
        WHENET_REQUIRE(value >= 1 && value <= MAX_LANES, WHENET_EINVAL, "host_lanes must be 1..8");
        host_lanes_ = int(value);
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
        // stem + per block {expand, dw | front} {se} project + head conv + heads: the launches enqueue_block() issues
        int k = 1 + 2;
        for (const DevBlock& b : blocks_) {
            const BlockSchedule bs = block_schedule(b);
            if (bs.use_mb7) { k += 1; continue; }       // the whole block is one launch (mb7.hip)
            k += bs.fused ? 1 : (b.spec.has_expand() ? 2 : 1);
            k += bs.se_fused ? 1 : 2;                   // project alone when it computes the gate itself
        }
        if (fold12_active()) k -= 1;                    // block 1's project launch
        if (stem_fuse_active()) k -= 1;                 // the stem conv runs inside block 1's depthwise kernel
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
// Which kernels a block runs (options fuse_front / front_impl / se_fuse / pw_impl and the per-layer tables): ONE statement of
// it, used by enqueue_block() and by get_info()'s launch count.
Engine::BlockSchedule Engine::block_schedule(const DevBlock& b, int n) const {
    BlockSchedule r;
    r.fused = fuse_front_ && b.spec.has_expand() && pw_impl_ == 0;
    r.use_f7 = r.fused && front_impl_ == 1 && front7_ && b.f7_supported;
    // round 6: blocks 13-16 of an f16 handle as ONE launch (mb7.hip): one workgroup per crop, every intermediate tensor in LDS
    r.use_mb7 = r.use_f7 && mb7_ && dtype_ == WHENET_F16 && b.mb7.ok;
    r.use_f2 = r.fused && !r.use_f7 && dtype_ == WHENET_F16 && (front_impl_ == 2 || (front_impl_ == 1 && b.f2_preferred));
    const bool f2s_pick = f2s_mask_ >= 0 ? ((f2s_mask_ >> b.spec.index) & 1) != 0 : b.f2s_preferred;     // (option "f2s_mask": probes)
    r.use_f2s = r.fused && !r.use_f7 && split_ && split_pw_ && b.f2s_supported && b.expand.wps != nullptr &&
                (front_impl_ == 2 || (front_impl_ == 1 && f2s_pick));
    r.se_in_front = r.fused;                 // the front kernels apply the SE reduce conv to their channel sums
    r.se_ntiles = r.use_f7 ? 1 : (r.use_f2 ? b.f2plan.ntiles() : (r.use_f2s ? b.f2splan.ntiles() : (r.fused ? b.fplan.ntiles() : b.dw.plan.ntiles())));
    r.se_chunks = r.use_f7 ? b.f7_chunks : (r.use_f2 ? b.f2plan.chunks : (r.use_f2s ? b.f2splan.chunks : b.fplan.chunks));
    const bool se_pays = b.project.K < 320 && r.se_ntiles * r.se_chunks <= 24;
    // option "se_fuse_tiny": chains of at most that many crops are launch-bound (B=1: 46-49 launches x ~8 us), so every launch
    // saved pays -- EXCEPT on the 7 x 7 blocks (K = 1152), whose project workgroups would each pull the whole 221 KB excite
    // kernel: se_fuse=2 on all blocks measured 451 us against 420 us at B=1 (round 5).  Blocks 2-12 fuse; same bits either way.
    const bool tiny = !single_stage_call_ && n > 0 && n <= se_fuse_tiny_ && b.project.K < 1152;
    // option se_fuse = 3 (round 6): as 1, plus the deep contractions (blocks 7 - 16): their LDS-staged split-K project GEMM computes the
    // gate of each wave's own k-groups on the matrix cores (pw.hip GM = 3) -- f16 and f32s handles.  Ten launches fewer and SLOWER:
    // every project workgroup pulls the 110 KB excite image through a CU that already fetches at its ~55 GB/s limit (f16, 64 crops:
    // project 12.0 -> 17 us on 14x14, 13.1 -> 21.8 us on 7x7, against a 6.6 - 7.4 us squeeze-excite launch; B = 1: 288 -> 298 us;
    // line 162 -> 157 k crops/s; profiles/r06/se_mfma_ab.txt).  Not the default.
    r.se_mfma = r.se_in_front && pw_impl_ == 0 && se_fuse_ == 3 && pw_staged_ && b.project.K >= 320 && b.se.w2p != nullptr &&
                (dtype_ == WHENET_F16 || (split_ && split_pw_));
    r.se_fused = r.se_in_front && pw_impl_ == 0 && (se_fuse_ == 2 || r.se_mfma || ((se_fuse_ == 1 || se_fuse_ == 3) && (se_pays || tiny)));
    return r;
}

bool Engine::fold12_active() const {
    if (!(fold12_ && fuse_front_ && pw_impl_ == 0 && blocks_.size() >= 2)) return false;
    if (dtype_ == WHENET_F16) return front_impl_ == 2 || (front_impl_ == 1 && blocks_[1].f2_preferred);
    // f32s (round 6): front.hip's split form takes the gate on its float32 operand; the exact-f32 form does not fold (its expand is
    // matrix-pipe bound: doubling the contraction costs more than block 1's project launch saves)
    const BlockSchedule b2 = block_schedule(blocks_[1]);
    return split_ && split_pw_ && fold12_pw_.wps != nullptr && b2.fused && !b2.use_f2s && !b2.use_f7;
}

// The stem's output is read by block 1's depthwise conv only (block 1 has no expand conv and no skip): for handles fed
// uint8 crops the two are one kernel (stemdw.hip) when block 1's depthwise tile plan is the one that kernel is built for.
bool Engine::stem_fuse_active() const {
    if (!(stem_fuse_ && !blocks_.empty())) return false;
    const DevBlock& b = blocks_[0];
    return !b.spec.has_expand() && !b.spec.has_skip() &&
           stemdw_supported(dtype_, b.dw.plan, b.spec.k, b.spec.s, b.spec.h_in, b.spec.cexp());
}

void* Engine::enqueue_blocks(int first, int last, const View& v, void* cur, int n, hipStream_t s, LaunchRecorder* rec,
                             bool b1_dw_done) {
    const bool fold = fold12_active() && first <= 1 && last >= 2;
    for (int i = first; i <= last; ++i) {
        void* nxt = (cur == v.x0) ? v.x1 : v.x0;
        enqueue_block(blocks_[size_t(i - 1)], v, cur, nxt, n, s, rec, fold && i <= 2 ? i : 0, b1_dw_done && i == 1);
        cur = nxt;
    }
    return cur;
}

void Engine::enqueue_block(const DevBlock& b, const View& v, const void* in, void* out, int n, hipStream_t s,
                           LaunchRecorder* rec, int fold, bool dw_done) {
    Rec R{rec, s, repeat_};
    const BlockSpec& sp = b.spec;
    const std::string p = "b" + std::to_string(sp.index);
    const double es = double(esz());
    const int hw_in = sp.h_in * sp.h_in, hw_out = sp.h_out * sp.h_out;
    const int cexp = sp.cexp();
    const void* dw_in = in;
    const BlockSchedule bs = block_schedule(b, n);
    const bool fused = bs.fused, use_f2 = bs.use_f2, se_in_front = bs.se_in_front, se_fused = bs.se_fused;
    const int se_ntiles = bs.se_ntiles, se_chunks = bs.se_chunks;
    // (checked before anything is enqueued: a violated invariant must not leave half a block in a stream capture)
    WHENET_REQUIRE(fold == 0 || (fold == 1 && !fused && sp.index == 1) ||
                       (fold == 2 && sp.index == 2 && (use_f2 || (split_ && fused && !bs.use_f2s && !bs.use_f7))), WHENET_EINVAL,
                   "fold12: block outside the folded pair");
    if (bs.use_mb7) {
        WHENET_REQUIRE(fold == 0, WHENET_EINVAL, "mb7: block outside the 7 x 7 stage");
        Mb7Args a{};
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wds = b.mb7.wds;
        a.bd = b.dw.bias;
        a.w1p = b.mb7.w1p;
        a.b1 = b.se.b1;
        a.w2p = b.mb7.w2p;
        a.b2 = b.se.b2;
        a.wpp = b.project.wp;
        a.bp = b.project.bias;
        a.out = out;
        if (single_stage_call_) {            // whenet_op_block reads the block's intermediate tensors back
            a.dbg_dw = v.d;
            a.dbg_gate = v.gate;
        }
        a.k = sp.k;
        a.Cout = sp.cout;
        a.skip = sp.has_skip();
        a.n = n;
        R(p + "/mbconv", "mbconv", kernel_name_mb7(sp.k, sp.cout, a.skip).c_str(),
          double(n) * (hw_in * sp.cin * (a.skip ? 2 : 1) + hw_out * sp.cout) * es,
          2.0 * n * (double(hw_in) * sp.cin * cexp + double(hw_out) * sp.k * sp.k * cexp + double(hw_out) * cexp * sp.cout),
          [&] { launch_mb7(a, s); });
        return;
    }
    if (bs.use_f7) {
        Front7Args a{};
        a.dtype = dtype_;
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wdt = dtype_ == WHENET_F16 ? static_cast<const void*>(b.dw.wt7) : static_cast<const void*>(b.dw.w);
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        a.w1t = b.se.w1t;                    // the SE reduce conv is applied by the front kernel to its channel sums
        a.R = b.se.R;
        a.k = sp.k;
        a.Cin = sp.cin;
        a.Cexp = cexp;
        a.NTe = b.expand.NTILES;
        a.split = split_ && split_pw_ && b.expand.wps != nullptr;
        a.weps = b.expand.wps;
        a.wsi = b.expand.wsi;
        a.n = n;
        a.xcd_grouped = xcd_grouped(2, n);
        a.plan = front7_plan_for(dtype_, sp.cin, cexp, n);
        R(p + "/front", "front", kernel_name_front7(dtype_, sp.k, a.plan, a.split).c_str(), double(n) * (hw_in * sp.cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * sp.cin * cexp + double(hw_out) * sp.k * sp.k * cexp), [&] { launch_front7(a, s); });
    } else if (use_f2) {
        Front2Args a{};
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wdt = b.dw.wt;
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        a.w1t = b.se.w1t;                    // the SE reduce conv is applied by the front kernel to its channel sums
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
        a.xcd_grouped = xcd_grouped(1, n);
        a.plan = b.f2plan;
        a.plan.threads = front2_threads(b.f2plan, n);
        R(p + "/front", "front", kernel_name_front2(sp.k, sp.s, a.KSe, a.plan.threads, a.plan.xs, a.in_gate != nullptr).c_str(),
          double(n) * (hw_in * a.Cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * a.Cin * cexp + double(hw_out) * sp.k * sp.k * cexp), [&] { launch_front2(a, s); });
    } else if (bs.use_f2s) {
        Front2sArgs a{};
        a.x = in;
        a.weps = b.expand.wps;
        a.be = b.expand.bias;
        a.wdt = b.dw.wts;
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        a.w1t = b.se.w1t;                    // the SE reduce conv is applied by the front kernel to its channel sums
        a.R = b.se.R;
        a.k = sp.k;
        a.s = sp.s;
        a.H = sp.h_in;
        a.Ho = sp.h_out;
        a.Cin = sp.cin;
        a.Cexp = cexp;
        a.pad = sp.pad_before();
        a.KSe = b.expand.KSs;
        a.NTe = b.expand.NTILES;
        a.wsi = b.expand.wsi;
        a.wsi_d = b.dw.wts_wsi;
        a.tm = b.f2s_tm;
        a.n = n;
        a.plan = b.f2splan;
        R(p + "/front", "front", kernel_name_front2s(sp.k, sp.s, a.KSe, a.plan.threads, a.tm, false).c_str(),
          double(n) * (hw_in * sp.cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * sp.cin * cexp + double(hw_out) * sp.k * sp.k * cexp), [&] { launch_front2s(a, s); });
    } else if (fused) {
        FrontArgs a{};
        a.x = in;
        a.wep = b.expand.wp;
        a.be = b.expand.bias;
        a.wd = b.dw.w;
        a.bd = b.dw.bias;
        a.out = v.d;
        a.rpart = v.partial;
        a.w1t = b.se.w1t;                    // the SE reduce conv is applied by the front kernel to its channel sums
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
        a.split = split_ && split_pw_ && b.expand.wps != nullptr;
        a.weps = b.expand.wps;
        a.KSes = b.expand.KSs;
        a.wsi = b.expand.wsi;
        if (fold == 2) {                     // input = block 1's depthwise output, gated; weights = project1 x expand2 (f32s)
            a.weps = fold12_pw_.wps;
            a.be = fold12_pw_.bias;
            a.Cin = fold12_pw_.K;
            a.KSe = fold12_pw_.KS;
            a.KSes = fold12_pw_.KSs;
            a.NTe = fold12_pw_.NTILES;
            a.wsi = fold12_pw_.wsi;
            a.in_gate = static_cast<const float*>(v.gate);
        }
        a.n = n;
        a.xcd_grouped = xcd_grouped(1, n);
        a.plan = b.fplan;
        a.plan.threads = front_threads(b.fplan, n);
        R(p + "/front", "front", kernel_name_front(dtype_, sp.k, sp.s, a.plan.threads).c_str(), double(n) * (hw_in * a.Cin + hw_out * cexp) * es,
          2.0 * n * (double(hw_in) * a.Cin * cexp + double(hw_out) * sp.k * sp.k * cexp),
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
        set_split(a, b.expand);
        a.HW = hw_in;
        a.act = ACT_SWISH;
        R(p + "/expand", "pw", kernel_name_pw(a, dtype_, pw_impl_, num_cus_).c_str(), double(a.M) * (a.K + a.N) * es,
          2.0 * a.M * a.K * a.N, [&] { launch_pw(a, dtype_, pw_impl_, num_cus_, s); });
        dw_in = v.e;
    }
    if (!fused && !dw_done) {
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
        if (bs.se_mfma) {
            sef.w2p = b.se.w2p;
            sef.KSr = b.se.KSr;
            sef.w2_wsi = b.se.w2_wsi;
        }
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
        set_split(a, b.project);
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
    const bool stemdw = stem_fuse_active() && d_in_f32 == nullptr;
    if (stemdw) {
        // stem + block 1's depthwise conv (stemdw.hip): writes what enqueue_block() would have block 1's dw.hip launch write
        const DevBlock& b1 = blocks_[0];
        const bool fold = fold12_active() && blocks_.size() >= 2;
        StemDwArgs a{};
        a.dtype = dtype_;
        a.w = d_stem_w_;
        a.lut = d_lut_;
        a.in = d_in;
        a.out = fold ? v.x1 : v.d;
        a.tab = d_stemdw_tab_;
        a.bias = d_stem_b_;
        a.wd = b1.dw.w;
        a.bd = b1.dw.bias;
        a.partial = v.partial;
        a.n = n;
        R("stem+b1/dw", "stem", kernel_name_stemdw(dtype_), double(n) * (IN_BYTES + X_ELEMS * es), 2.0 * n * (10838016.0 + 112.0 * 112 * 9 * 32),
          [&] { launch_stemdw(a, s); });
    } else {
        StemArgs a{d_in, v.x0, d_stem_w_, d_stem_b_, d_lut_, n};
        a.in_f32 = d_in_f32;
        R("stem", "stem", kernel_name_stem(dtype_), double(n) * (IN_BYTES + X_ELEMS * es), 2.0 * n * 10838016.0,
          [&] { launch_stem(a, dtype_, s); });
    }
    void* cur = enqueue_blocks(1, int(blocks_.size()), v, v.x0, n, s, rec, stemdw);
    const bool fuse_head = head_fuse_ && pw_impl_ == 0 && split_heads_ && head7_supported(dtype_, head_.K, head_.N, 49) &&
                           partial_per_crop_ >= size_t(heads_split()) * N_LOGITS;
    if (fuse_head) {
        // head conv + GlobalAveragePooling2D as one kernel (head7.hip): v.hc receives the pooled features [n][1280] f32
        Head7Args a{};
        a.dtype = dtype_;
        a.x = cur;
        a.wep = head_.wp;
        a.bias = head_.bias;
        a.feat = reinterpret_cast<float*>(v.hc);
        a.K = head_.K;
        a.N = head_.N;
        a.NTILES = head_.NTILES;
        a.split = split_ && split_pw_ && head_.wps != nullptr;
        a.weps = head_.wps;
        a.wsi = head_.wsi;
        a.n = n;
        a.xcd_grouped = xcd_grouped(4, n);
        R("head", "pw", kernel_name_head7(dtype_, n, a.split).c_str(), double(n) * (49.0 * a.K * es + a.N * 4.0), 2.0 * n * 49.0 * a.K * a.N,
          [&] { launch_head7(a, s); });
        HeadsArgs hargs{};
        hargs.feat_in = reinterpret_cast<const float*>(v.hc);
        hargs.w = d_dense_w_;
        hargs.b = d_dense_b_;
        hargs.logits = d_logits;
        hargs.ypr = d_ypr;
        hargs.argmax = d_amax;
        hargs.n = n;
        R("heads", "heads", "whenet_heads_split_kernel<float, true>", double(n) * ((FEAT + N_LOGITS + 6) * 4.0) + double(FEAT) * N_LOGITS * 4.0,
          2.0 * n * FEAT * N_LOGITS, [&] { launch_heads_split(hargs, v.partial, v.hcount, dtype_, s); });
        return;
    }
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
        set_split(a, head_);
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
                                  (dtype_ == WHENET_F16 ? "_Float16" : "float") + (split ? ", false>" : ">");
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
int Engine::lanes_for(int n, int want) const {
    int lanes = want > 0 ? std::min(want, MAX_LANES) : lanes_;
    while (lanes > 1 && n / lanes < min_lane_crops_) --lanes;
    return lanes;
}

void Engine::enqueue_lanes(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s, int want) {
    const int lanes = lanes_for(n, want);
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

void Engine::run_forward(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s, int want) {
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
        enqueue_lanes(d_in, n, d_ypr, d_amax, d_logits, s, want);
        return;
    }
    const int lanes = lanes_for(n, want);
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
    hipGraphExec_t g = cached_graph(GraphKey{n, -lanes, d_in, d_ypr, d_amax, d_logits}, s,
                                    [&] { enqueue_lanes(d_in, n, d_ypr, d_amax, d_logits, s, lanes); });
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
    if (n <= host_pinned_max_) {
        // The reference's own call shape -- get_angle(uint8[1,224,224,3]) per head (demo.py:14, demo_video.py:27): latency.
        // Copies to and from PAGEABLE memory are synchronous inside the runtime (the three result copies each wait for the
        // forward and go through its staging buffer one after the other); through a pinned slot the call is memcpy ->
        // H2D -> graph -> 3 D2H, all asynchronous on ONE stream, one wait, memcpy out (round 5: 490 -> see DESIGN us at B=1 f32).
        Slot& sl = host_slot_;
        ensure_slot(sl, std::max(n, std::min(host_pinned_max_, 16)));
        std::memcpy(sl.h_in, crops, N * IN_BYTES);
        WHENET_HIP_CHECK(hipMemcpyAsync(sl.d_in, sl.h_in, N * IN_BYTES, hipMemcpyHostToDevice, stream_));
        run_forward(sl.d_in, n, sl.d_ypr, sl.d_amax, sl.d_logits, stream_, host_lanes_);
        WHENET_HIP_CHECK(hipMemcpyAsync(sl.h_ypr, sl.d_ypr, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
        if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(sl.h_amax, sl.d_amax, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        if (logits)
            WHENET_HIP_CHECK(hipMemcpyAsync(sl.h_logits, sl.d_logits, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
        WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
        std::memcpy(ypr, sl.h_ypr, N * 3 * sizeof(float));
        if (argmax) std::memcpy(argmax, sl.h_amax, N * 3 * sizeof(int32_t));
        if (logits) std::memcpy(logits, sl.h_logits, N * N_LOGITS * sizeof(float));
        return;
    }
    // (Measured, round 4: issuing the copy lane by lane in front of per-lane graphs, so that the first chain runs while the
    //  second lane's crops travel, changes nothing -- 68.5 k vs 69.6 k crops/s at 64 crops: the 9.6 MB copy from pageable
    //  memory is 0.18 ms of a 0.92 ms call, and two graphs on two streams lose what the overlap gains.)
    ensure_host_out(n);
    WHENET_HIP_CHECK(hipMemcpyAsync(in_u8_, crops, N * IN_BYTES, hipMemcpyHostToDevice, stream_));
    // a blocking call has the GPU to itself whatever "inflight" says: the forward runs as host_lanes_ chains
    run_forward(in_u8_, n, o_ypr_, o_amax_, o_logits_, stream_, host_lanes_);
    // results: asynchronous copies into pinned memory, ONE wait (copies into the caller's pageable arrays each wait for the forward
    // and for each other inside the runtime)
    WHENET_HIP_CHECK(hipMemcpyAsync(hout_ypr_, o_ypr_, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (argmax) WHENET_HIP_CHECK(hipMemcpyAsync(hout_amax_, o_amax_, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    if (logits)
        WHENET_HIP_CHECK(hipMemcpyAsync(hout_logits_, o_logits_, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    std::memcpy(ypr, hout_ypr_, N * 3 * sizeof(float));
    if (argmax) std::memcpy(argmax, hout_amax_, N * 3 * sizeof(int32_t));
    if (logits) std::memcpy(logits, hout_logits_, N * N_LOGITS * sizeof(float));
}

void Engine::ensure_host_out(int n) {
    if (n <= hout_cap_) return;
    // One allocation that covers every blocking call below the fan-out threshold (264 KB of pinned memory), made under a process-wide
    // lock: growing these buffers (hipHostFree + hipHostMalloc) while another handle's thread was inside its own forward crashed
    // inside the runtime (2-3 of 8 runs of tests/test_multi_device.py with two handles on one device).
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    if (hout_ypr_) (void)hipHostFree(hout_ypr_);
    if (hout_amax_) (void)hipHostFree(hout_amax_);
    if (hout_logits_) (void)hipHostFree(hout_logits_);
    hout_ypr_ = nullptr; hout_amax_ = nullptr; hout_logits_ = nullptr;
    hout_cap_ = 0;
    const size_t N = size_t(std::max(n, 256));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hout_ypr_), N * 3 * sizeof(float), hipHostMallocDefault));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hout_amax_), N * 3 * sizeof(int32_t), hipHostMallocDefault));
    WHENET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hout_logits_), N * N_LOGITS * sizeof(float), hipHostMallocDefault));
    hout_cap_ = int(N);
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

hipStream_t Engine::copy_stream_handle() {
    DeviceGuard guard(device_);
    return copy_stream();
}

hipEvent_t Engine::copied_event(int ticket) const {
    for (const Slot& s : slots_)
        if (s.busy && s.ticket == ticket) return s.copied;
    throw Error(WHENET_EINVAL, "unknown ticket " + std::to_string(ticket));
}

int Engine::submit(const uint8_t* crops, int n, int stage, int want_lanes, hipStream_t copy_on, hipEvent_t copy_after) {
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
    // Through a pinned slot: a copy straight from the caller's pageable memory (hipMemcpyAsync stages it inside the runtime)
    // blocks the host until the DMA is done and serialises the submissions -- measured round 4: 67.8 k vs 90.4 k crops/s with
    // three 64-crop batches in flight.
    // stage 1 (the fan-out of one large blocking call, capi.cpp): straight from the caller's memory on the copy stream -- the
    // host blocks for THIS chunk's DMA only, while the forwards of the chunks before it run.
    const hipStream_t cs = copy_on != nullptr ? copy_on : copy_stream();
    if (copy_after != nullptr) WHENET_HIP_CHECK(hipStreamWaitEvent(cs, copy_after, 0));
    if (stage == 1) {
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->d_in, crops, N * IN_BYTES, hipMemcpyHostToDevice, cs));
    } else {
        std::memcpy(slot->h_in, crops, N * IN_BYTES);
        WHENET_HIP_CHECK(hipMemcpyAsync(slot->d_in, slot->h_in, N * IN_BYTES, hipMemcpyHostToDevice, cs));
    }
    WHENET_HIP_CHECK(hipEventRecord(slot->copied, cs));
    WHENET_HIP_CHECK(hipStreamWaitEvent(stream_, slot->copied, 0));
    run_forward(slot->d_in, n, slot->d_ypr, slot->d_amax, slot->d_logits, stream_, want_lanes);
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_ypr, slot->d_ypr, N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_amax, slot->d_amax, N * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipMemcpyAsync(slot->h_logits, slot->d_logits, N * N_LOGITS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipEventRecord(slot->done, stream_));
    slot->busy = true;
    slot->n = n;
    slot->ticket = next_ticket_++;
    return slot->ticket;
}

// After a failed fan-out: wait for whatever was enqueued and hand every slot back.
void Engine::abandon_submissions() {
    DeviceGuard guard(device_);
    (void)hipStreamSynchronize(stream_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    for (Slot& s : slots_) s.busy = false;
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
        o.crops = n / lanes + (0 < n % lanes ? 1 : 0);       // (chain 0's sub-batch: what bytes / flops are counted for)
        o.chains = lanes;
    }
    return count;
}

}  // namespace whenet
