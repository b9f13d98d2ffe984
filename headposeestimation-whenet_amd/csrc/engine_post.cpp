// Frame-level entry points around the hot path: one submission per video frame (crop + resize +
// colour order on the device, then the forward), the crop/resize stage alone, and the YOLOv3 detector's post-processing.
#include "engine_internal.h"

namespace whenet {

using namespace detail;

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
        size_t used = 0;
        size_t add(size_t nbytes) {
            const size_t off = (used + 255) & ~size_t(255);
            used = off + (nbytes ? nbytes : 16);
            return off;
        }
    };
    int N = 0;
    const size_t per = size_t(5 + num_classes) * 3;
    size_t feat_off[3] = {0, 0, 0}, feat_bytes[3] = {0, 0, 0};
    Carver cv;
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
    // The per-class counts first, then ONLY the selected rows, straight into the caller's arrays (model.py:227-229:
    // concatenated class by class).  Round 3 copied all C x max_boxes slots into temporaries: ~20 MB per frame for 80
    // classes x 10,647 boxes where the reference returns a handful of detections.
    yolo_counts_.resize(C);
    WHENET_HIP_CHECK(hipMemcpyAsync(yolo_counts_.data(), a.out_count, C * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (all_boxes)
        WHENET_HIP_CHECK(hipMemcpyAsync(all_boxes, a.boxes, size_t(N) * 4 * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (all_scores)
        WHENET_HIP_CHECK(hipMemcpyAsync(all_scores, a.all_scores, size_t(N) * C * sizeof(float), hipMemcpyDeviceToHost, stream_));
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    int out = 0;
    for (size_t c = 0; c < C; ++c) {
        const size_t k = size_t(yolo_counts_[c]);
        if (k == 0) continue;
        WHENET_HIP_CHECK(hipMemcpyAsync(boxes + size_t(out) * 4, a.out_boxes + c * MB * 4, k * 4 * sizeof(float),
                                        hipMemcpyDeviceToHost, stream_));
        WHENET_HIP_CHECK(hipMemcpyAsync(scores + out, a.out_scores + c * MB, k * sizeof(float), hipMemcpyDeviceToHost, stream_));
        if (index)
            WHENET_HIP_CHECK(hipMemcpyAsync(index + out, a.out_index + c * MB, k * sizeof(int), hipMemcpyDeviceToHost, stream_));
        for (size_t j = 0; j < k; ++j) classes[size_t(out) + j] = int32_t(c);
        out += int(k);
    }
    WHENET_HIP_CHECK(hipStreamSynchronize(stream_));
    return out;
}

}  // namespace whenet
