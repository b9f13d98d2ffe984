// The per-GPU inference engine behind the C ABI: device weights, activation arena, launch
// schedule of the kernels of one forward, hipGraph cache, per-launch profiling, and the
// pinned-buffer submit/collect pipeline.  One engine = one device + one stream; not thread-safe.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "snapshot.h"

namespace whenet {

struct DevPw {
    int K = 0, N = 0, KS = 0, NTILES = 0;
    void* wp = nullptr;
    void* wps = nullptr;       // WHENET_F32S: split weight images (HostPw::packed_split)
    int KSs = 0;
    float wsi = 1.0f;
    float* wdense = nullptr;
    float* bias = nullptr;
};
struct DevDw {
    int k = 0, C = 0;
    float* w = nullptr;
    float* bias = nullptr;
    half_t* wt = nullptr;      // f16: Toeplitz operand image of the kernel for front2.hip (pack_dw_toeplitz)
    half_t* wt7 = nullptr;     // f16, 7 x 7 blocks: the image front7.hip reads (group-aligned: xs = 4 - k / 2)
    float* wts = nullptr;      // WHENET_F32S: Toeplitz operand image for front2s.hip in the block's tap mode (pack_dw_toeplitz_s)
    float wts_wsi = 1.0f;      //   2^-shift of its scaled taps (tap mode 1)
    DwPlan plan;
};
struct DevSe {
    int C = 0, R = 0;
    float *w1t = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    float* w2c = nullptr;      // [C][RP] (se.hip reads a channel's whole excite row with 16-byte loads)
    void* w2p = nullptr;       // the excite kernel as an MFMA operand image (HostSe::excite), or nullptr
    int KSr = 0;
    float w2_wsi = 1.0f;
};
struct DevMb7 {                 // f16, blocks 13-16: operand images of the one-launch block kernel (mb7.hip)
    bool ok = false;
    half_t *wds = nullptr, *w1p = nullptr, *w2p = nullptr;
};
struct DevBlock {
    BlockSpec spec;
    DevPw expand;
    DevDw dw;
    FrontPlan fplan;       // fused expand+depthwise tiling (blocks with an expand conv)
    Front2Plan f2plan;     // f16: the same stage with the taps on the matrix cores (front2.hip)
    bool f2_preferred = false;
    Front2Plan f2splan;    // WHENET_F32S: front2s.hip's plan, tap mode, and whether it is the faster kernel on this layer
    int f2s_tm = 2;
    bool f2s_supported = false, f2s_preferred = false;
    bool f7_supported = false; // f16: the block's shape has a front7.hip kernel (7 x 7 maps, blocks 13-16)
    int f7_chunks = 1;         // channel chunks of its plans (the same for every group size: see front7_plan_for)
    DevSe se;
    DevPw project;
    DevMb7 mb7;
};

// Collects one entry per kernel launch when profiling (event pair around each launch).
struct LaunchRecorder {
    struct Entry {
        std::string layer, kind, kernel;
        double bytes = 0, flops = 0;
        hipEvent_t stop = nullptr;      // recorded right after the launch; the previous entry's stop
        double total_ms = 0;             // (or the chain's start event) is this launch's start
    };
    std::vector<Entry> entries;
    hipEvent_t start = nullptr;          // recorded on the chain's stream before its first launch
    size_t cursor = 0;
    bool first_pass = true;
};

constexpr int MAX_LANES = 8;

class Engine {
  public:
    Engine(const void* snapshot, size_t nbytes, int device_id, int dtype);      // dtype: WHENET_F32 / F16 / F32S
    explicit Engine(int device_id);      // no network: device + stream + scratch for the frame / detector stages only
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    void set_option(const std::string& key, long value);
    void get_info(whenet_info_t* out) const;

    void forward_host(const uint8_t* crops, int n, float* ypr, int32_t* argmax, float* logits);
    void forward_host_f32(const float* x, int n, float* ypr, int32_t* argmax, float* logits);
    void forward_device(const uint8_t* d_crops, int n, float* d_ypr, int32_t* d_argmax, float* d_logits,
                        hipStream_t stream);
    void sync();
    void release_aux_streams();
    // stage: 0 = through the slot's pinned staging buffer, 1 = DMA straight from the caller's memory (pageable: the host blocks for the
    // copy; registered: asynchronous).
    // copy_on: the stream the H2D copy is issued on (default: this engine's own copy stream) -- a fan-out puts every chunk's copy on ONE
    // stream: in order, at the link's full rate
    // copy_after: the copy starts only after this event (the previous chunk's copy on ANOTHER engine's copy stream)
    int submit(const uint8_t* crops, int n, int stage = 0, int want_lanes = 0, hipStream_t copy_on = nullptr, hipEvent_t copy_after = nullptr);
    hipStream_t copy_stream_handle();
    hipEvent_t copied_event(int ticket) const;     // recorded when the H2D copy of that submission is done
    void abandon_submissions();
    bool has_pending() const {
        for (const Slot& s : slots_)
            if (s.busy) return true;
        return false;
    }
    int submit_frame(const uint8_t* frame, int fh, int fw, int swap_rb, const int32_t* rects, int k);
    void op_crop_resize(const uint8_t* frame, int fh, int fw, int swap_rb, const int32_t* rects, int k,
                        uint8_t* crops_out);
    void collect(int ticket, float* ypr, int32_t* argmax, float* logits);
    int profile(const uint8_t* d_crops, int n, int iters, whenet_launch_stat_t* stats, int cap);

    void op_stem(const uint8_t* crops, int n, float* out);
    int yolo_eval(const float* const* feats, const int* grid_h, const int* grid_w, int num_layers, const float* anchors,
                  int num_anchors, int num_classes, float image_h, float image_w, float score_threshold,
                  float iou_threshold, int max_boxes, float* boxes, float* scores, int32_t* classes, int32_t* index,
                  float* all_boxes, float* all_scores);
    void op_block(int index, const float* in, int n, float* expand_out, float* dw_out, float* gate, float* out);
    void op_block_range(int first, int last, const float* in, int n, float* out);
    void op_head(const float* in, int n, float* feat, float* logits, float* ypr, int32_t* argmax);
    void op_decode(const float* logits, int n, float* ypr, int32_t* argmax);

    void* dev_alloc(size_t nbytes);
    void dev_free(void* p);
    void h2d(void* d, const void* s, size_t nbytes);
    void d2h(void* d, const void* s, size_t nbytes);

    std::string last_error;

  private:
    struct Slot {          // one in-flight submission of the pinned pipeline
        int capacity = 0, n = 0, ticket = -1;
        bool busy = false;
        uint8_t* h_in = nullptr;
        uint8_t* d_in = nullptr;
        float *h_ypr = nullptr, *d_ypr = nullptr;
        int32_t *h_amax = nullptr, *d_amax = nullptr;
        float *h_logits = nullptr, *d_logits = nullptr;
        hipEvent_t copied = nullptr, done = nullptr;
        // frame submissions: the frame and the crop plans travel instead of the crops
        size_t frame_cap = 0;
        uint8_t *h_frame = nullptr, *d_frame = nullptr;
        int plan_cap = 0;
        int32_t *h_plan = nullptr, *d_plan = nullptr;
    };

    template <typename T> T* upload(const std::vector<T>& v);
    void* upload_bytes(const void* p, size_t nbytes);
    DevPw upload_pw(const HostPw& h);
    // WHENET_F32S: 1x1 products as binary16 hi/lo pairs (option "split_pw" switches the pointwise kernels between the two forms)
    void set_split(PwArgs& a, const DevPw& w) const {
        a.staged = pw_staged_;
        a.split = split_ && split_pw_ && w.wps != nullptr;
        a.wps = w.wps;
        a.KSs = w.KSs;
        a.wsi = w.wsi;
    }
    bool split_ = false;        // the handle was created as WHENET_F32S
    bool split_pw_ = true;      // option "split_pw"
    bool pw_staged_ = true;     // option "pw_staged": split-K GEMMs fetch their activation rows coalesced, through LDS
    void ensure_capacity(int n);
    void release_arena();
    void drop_graphs();
    size_t esz() const { return dtype_ == WHENET_F16 ? 2 : 4; }

    struct View {          // the activation arena as seen by a sub-batch starting at some crop
        void *x0, *x1, *e, *d, *hc;
        float *partial, *gate;
        unsigned* hcount;
    };
    View view(int crop_off) const;
    // enqueue the kernels of one forward on `s` (eager); rec != nullptr -> event pairs
    void enqueue_forward(const View& v, const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits,
                         hipStream_t s, LaunchRecorder* rec, const float* d_in_f32 = nullptr);
    // fold: 0 = the block as it stands; 1 = block 1 without its project (its depthwise output goes to `out`);
    //       2 = block 2 fed by that output, block 1's project folded into its expand weights (fold12_active())
    void enqueue_block(const DevBlock& b, const View& v, const void* in, void* out, int n, hipStream_t s,
                       LaunchRecorder* rec, int fold = 0, bool dw_done = false);
    // blocks first..last (1-based) as the forward pass runs them; returns the buffer (x0 / x1 of `v`) holding the result
    void* enqueue_blocks(int first, int last, const View& v, void* cur, int n, hipStream_t s, LaunchRecorder* rec,
                         bool b1_dw_done = false);
    bool stem_fuse_active() const;
    bool fold12_active() const;
    struct BlockSchedule {     // which kernels a block runs under the current options
        bool fused = false, use_f2 = false, use_f2s = false, use_f7 = false, se_in_front = false, se_fused = false, se_mfma = false;
        bool use_mb7 = false;  // the whole block is one launch (mb7.hip): no front / squeeze-excite / project launches
        int se_ntiles = 1, se_chunks = 1;
    };
    // n = crops of the chain the block runs in (0: not batch-specific, e.g. the launch count of get_info)
    BlockSchedule block_schedule(const DevBlock& b, int n = 0) const;
    int se_fuse_tiny_ = 0;             // option "se_fuse_tiny"
    int f2s_mask_ = -1;                // option "f2s_mask" (probes)
    bool single_stage_call_ = false;   // op_block / op_block_range: the schedule must not depend on the test's batch size
    int lanes_for(int n, int want) const;     // chains a forward of n crops runs as (want = 0: option "lanes")
    void enqueue_lanes(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s, int want = 0);
    void run_forward(const uint8_t* d_in, int n, float* d_ypr, int32_t* d_amax, float* d_logits, hipStream_t s, int want = 0);
    void ensure_slot(Slot& s, int n);
    hipStream_t lane_stream(int i);     // created on first use
    using GraphKey = std::tuple<int, int, const void*, void*, void*, void*>;   // n, lane offset (-L: whole batch as L chains), buffers
    template <typename F>
    hipGraphExec_t cached_graph(const GraphKey& key, hipStream_t s, F&& fn);
    void sync_streams(hipStream_t s);
    hipStream_t copy_stream();          // created on first use
    void ensure_slot_frame(Slot& s, size_t frame_bytes, int k);
    Slot* free_slot();

    void open_device(int device_id);
    void require_model() const;
    bool has_model_ = false;
    int device_ = 0, dtype_ = WHENET_F32, num_cus_ = 256;
    bool use_graph_ = true;
    int pw_impl_ = 0;
    int repeat_ = 1;
    bool split_heads_ = true;   // option "split_heads": GAP + Dense over 4 workgroups per crop, the last one decodes
    bool fuse_front_ = true;    // option "fuse_front": expand + depthwise as one kernel (front.hip / front2.hip)
    int se_fuse_ = 1;           // option "se_fuse": the project GEMM computes the SE gate of its own crops in its prologue (no
                                // squeeze-excite launch): 0 = never, 1 = on the blocks where that is faster, 2 = every fused-front block
    int front_impl_ = 1;        // option "front_impl": 0 = front.hip everywhere, 1 = per layer (f16: front2.hip where it is
                                // the faster kernel), 2 = front2.hip everywhere (f16)
    bool poison_ = false;       // debug option "poison": NaN-fill the activation arena before every forward
    bool head_fuse_ = true;     // option "head_fuse": the head conv pools its own output (head7.hip, f16 and f32); 0 = round 3's two stages
    bool front7_ = true;        // option "front7": blocks 13-16 of an f16 handle run front7.hip (a group of crops per workgroup)
                                // when front_impl = 1; 0 = the per-layer choice of round 3 (front.hip there)
    bool mb7_ = false;          // option "mb7": blocks 13-16 of an f16 handle run as ONE launch each (mb7.hip, round 6: one workgroup per crop,
                                // every intermediate tensor in LDS).  Measured (profiles/r06/mb7_*): 29 us per launch whatever the batch up to 256
                                // crops against 3 x 8 us at one crop and 67 us at 256 -- +3 % at batch 512, +-1 % at 64 x 3 in flight, -4 % one
                                // forward at a time, +60 us at batch 1.  Not the default: the schedule must not depend on the batch.
    int xcd_map_ = 7;           // option "xcd_map" (round 6), bit mask: the channel chunks that share an input are dealt to ONE XCD in
                                // 1 = front.hip / front2.hip, 2 = front7.hip, 4 = head7.hip (device_math.h xcd_unit).  Same bits either way.
                                // Measured on one box against the 3-D grid order of rounds 2-5 (profiles/r06/ab_xcd_map_*.txt): +0.8 % at
                                // 64 crops x 3 in flight, at batch 512 and for f32s; the 14 x 14 front kernels alone get slower with it
    bool xcd_always_ = false;   // option "concurrent" (set by the handle when it owns several engines): other forwards share the chip
    bool xcd_grouped(int bit, int n) const {      // the grouped placement pays when the chip is full: several forwards in flight, or a
        return (xcd_map_ & bit) != 0 && (xcd_always_ || n >= 128);     // launch of >= 128 crops; one small forward alone loses 1 % with it
    }
    bool stem_fuse_ = true;     // option "stem_fuse": uint8 input -- the stem conv is computed inside block 1's depthwise kernel (stemdw.hip)
    bool fold12_ = true;        // option "fold12": block 1's project folded into block 2's expand (f16 + front2.hip on block 2)
    int lanes_ = 2;             // concurrent sub-batch chains per forward (option "lanes"; round 3: 2 -- with the faster
                                // front kernels a third chain only adds contention: 100.1 k vs 97.2 k crops/s at batch 64,
                                // equal from 256 crops up and for f32)
    bool lane_graphs_ = false;  // one graph per lane on its own stream instead of one forked graph (option "lane_graphs")
    int min_lane_crops_ = 16;   // do not split below this many crops per chain
    int host_lanes_ = 2;        // chains of a BLOCKING host forward (it has the GPU to itself whatever "inflight" says)
    std::vector<hipStream_t> lane_streams_;
    std::vector<hipEvent_t> join_ev_;
    hipEvent_t fork_ev_ = nullptr;
    hipStream_t stream_ = nullptr, copy_stream_ = nullptr;
    hipDeviceProp_t prop_{};
    int64_t params_backbone_ = 0, params_heads_ = 0;
    int n_tensors_ = 0;

    // weights
    std::vector<void*> weight_allocs_;
    float *d_lut_ = nullptr, *d_stem_w_ = nullptr, *d_stem_b_ = nullptr;
    StemDwTable* d_stemdw_tab_ = nullptr;   // stemdw.hip's packed LUT + stem weight fragments
    std::vector<DevBlock> blocks_;
    DevPw head_;
    DevPw fold12_pw_;           // block 1 project x block 2 expand, 32 -> 96 (snapshot.cpp)
    float* d_fold12_w32_ = nullptr;   // its f32 fragment image (front2.hip rounds AFTER the per-crop gate)
    float *d_dense_w_ = nullptr, *d_dense_b_ = nullptr;

    // activation arena (grown to the largest n seen)
    int cap_ = 0;
    size_t arena_bytes_ = 0;
    void *x0_ = nullptr, *x1_ = nullptr, *e_ = nullptr, *d_ = nullptr, *hc_ = nullptr;
    float *partial_ = nullptr, *gate_ = nullptr;
    unsigned* hcount_ = nullptr;       // per-crop tickets of the split heads kernel (zero between launches)
    uint8_t* in_u8_ = nullptr;
    float* in_f32_ = nullptr;       // normalised float32 input of forward_host_f32 (grown on demand)
    int in_f32_cap_ = 0;
    float* o_ypr_ = nullptr;
    int32_t* o_amax_ = nullptr;
    float* o_logits_ = nullptr;
    size_t partial_per_crop_ = 0;
    unsigned char* yolo_scratch_ = nullptr;      // device scratch of yolo_eval, grown on demand
    size_t yolo_scratch_bytes_ = 0;
    std::vector<int> yolo_counts_;               // host staging of the per-class detection counts

    std::map<GraphKey, hipGraphExec_t> graphs_;

    Slot slots_[WHENET_MAX_INFLIGHT];
    Slot host_slot_;                   // pinned staging of small BLOCKING host forwards (forward_host, n <= host_pinned_max_)
    // pinned landing zone of the RESULTS of larger blocking host forwards: three asynchronous D2H copies and one wait instead of three
    // synchronous copies into the caller's pageable arrays (round 6)
    float* hout_ypr_ = nullptr;
    int32_t* hout_amax_ = nullptr;
    float* hout_logits_ = nullptr;
    int hout_cap_ = 0;
    void ensure_host_out(int n);
    int host_pinned_max_ = 8;      // measured round 5: pinned wins up to 8 crops (B=1 f32 420 vs 445 us), loses at 16-32
    int next_ticket_ = 0;
};

}  // namespace whenet
