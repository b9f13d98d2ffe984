// YOLOv3 post-processing on the device: box decode + score threshold + per-class greedy NMS.
//
// Reference: /root/reference/yolo_v3/model.py:125-150 (yolo_head), :153-178 (yolo_correct_boxes),
// :181-190 (yolo_boxes_and_scores), :193-232 (yolo_eval: anchor masks, `box_scores >= score_threshold`,
// tf.image.non_max_suppression per class, results concatenated class by class); run by the reference
// inside sess.run (yolo_postprocess.py:198-204) on the detector's 2 or 3 output maps.  SURVEY.md §8f row 4.
//
// Mapping (gfx950).  The work is small (10,647 boxes for a 416x416 input) and branchy, so it is two launches:
//   1. whenet_yolo_decode_kernel: one lane per box (layer, y, x, anchor), coalesced over the map.  It writes the
//      corrected box (y_min, x_min, y_max, x_max in image pixels) to a dense [N][4] array and, for every class
//      whose score = confidence * class probability passes the threshold, appends a 64-bit key
//      (score bits << 32 | ~box index) to that class's candidate list (one atomic counter per class).
//   2. whenet_yolo_nms_kernel: one workgroup per class.  The keys are sorted descending (bitonic, in LDS up to
//      4096 candidates, in the global key array beyond that): score first, LOWER box index first among equal
//      scores -- the append order of step 1 is arbitrary, the sorted order is not.  Wave 0 then walks the
//      candidates in that order: the lanes test the candidate against the boxes selected so far (TensorFlow's
//      IOU(): min/max-normalised corners, zero for a box of non-positive area, suppressed when IoU > threshold,
//      float32 with individually rounded operations), one ballot decides, selected boxes live in LDS.
// All arithmetic is float32 with the operation order of the reference's graph; exp / sigmoid go through expf,
// so box coordinates agree with a float32 CPU run to an ulp or two, not bitwise (tests/test_yolo.py).
#include "kernels.h"

namespace whenet {

namespace {

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void whenet_yolo_decode_kernel(YoloArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    int l = 0;
    while (l + 1 < a.num_layers && i >= a.layer[l + 1].first) ++l;
    const YoloLayer& L = a.layer[l];
    const int j = i - L.first;                                  // ((y * gw + x) * A + anchor)
    const int an = j % a.na, cell = j / a.na;
    const int y = cell / L.gw, x = cell - y * L.gw;
    const float* t = L.feats + size_t(j) * (5 + a.num_classes);

    // yolo_head (model.py:141-145)
    const float bx = __fdiv_rn(__fadd_rn(sigmoid_ref(t[0]), float(x)), float(L.gw));
    const float by = __fdiv_rn(__fadd_rn(sigmoid_ref(t[1]), float(y)), float(L.gh));
    const float bw = __fdiv_rn(__fmul_rn(expf(t[2]), L.anchor[an][0]), a.input_w);
    const float bh = __fdiv_rn(__fmul_rn(expf(t[3]), L.anchor[an][1]), a.input_h);
    const float conf = sigmoid_ref(t[4]);
    // yolo_correct_boxes (model.py:155-177): y first
    const float cy = __fmul_rn(__fsub_rn(by, a.off_y), a.scale_y), cx = __fmul_rn(__fsub_rn(bx, a.off_x), a.scale_x);
    const float hh = __fmul_rn(bh, a.scale_y), ww = __fmul_rn(bw, a.scale_x);
    const float hy = __fdiv_rn(hh, 2.0f), hx = __fdiv_rn(ww, 2.0f);
    float4 box;
    box.x = __fmul_rn(__fsub_rn(cy, hy), a.image_h);
    box.y = __fmul_rn(__fsub_rn(cx, hx), a.image_w);
    box.z = __fmul_rn(__fadd_rn(cy, hy), a.image_h);
    box.w = __fmul_rn(__fadd_rn(cx, hx), a.image_w);
    reinterpret_cast<float4*>(a.boxes)[i] = box;
    // yolo_boxes_and_scores (model.py:188) + the mask of yolo_eval (model.py:212)
    for (int c = 0; c < a.num_classes; ++c) {
        const float score = __fmul_rn(conf, sigmoid_ref(t[5 + c]));
        if (a.all_scores) a.all_scores[size_t(i) * a.num_classes + c] = score;
        if (score >= a.score_thr) {
            const int slot = atomicAdd(&a.counts[c], 1);
            a.keys[size_t(c) * a.NP + slot] =
                (static_cast<unsigned long long>(__float_as_uint(score)) << 32) | (0xffffffffu - unsigned(i));
        }
    }
}

// TensorFlow's IOU() (core/kernels/non_max_suppression_op.cc), float32, every operation rounded on its own
__device__ __forceinline__ float iou_tf(const float4 a, const float4 b) {
    const float ymin_a = fminf(a.x, a.z), xmin_a = fminf(a.y, a.w), ymax_a = fmaxf(a.x, a.z), xmax_a = fmaxf(a.y, a.w);
    const float ymin_b = fminf(b.x, b.z), xmin_b = fminf(b.y, b.w), ymax_b = fmaxf(b.x, b.z), xmax_b = fmaxf(b.y, b.w);
    const float area_a = __fmul_rn(__fsub_rn(ymax_a, ymin_a), __fsub_rn(xmax_a, xmin_a));
    const float area_b = __fmul_rn(__fsub_rn(ymax_b, ymin_b), __fsub_rn(xmax_b, xmin_b));
    if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
    const float iymin = fmaxf(ymin_a, ymin_b), ixmin = fmaxf(xmin_a, xmin_b);
    const float iymax = fminf(ymax_a, ymax_b), ixmax = fminf(xmax_a, xmax_b);
    const float inter = __fmul_rn(fmaxf(__fsub_rn(iymax, iymin), 0.0f), fmaxf(__fsub_rn(ixmax, ixmin), 0.0f));
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}

constexpr int NMS_THREADS = 1024;
constexpr int NMS_LDS_KEYS = 4096;             // 32 KB of keys
constexpr int NMS_MAX_SELECT = 256;

__global__ __launch_bounds__(NMS_THREADS) void whenet_yolo_nms_kernel(YoloArgs a) {
    __shared__ unsigned long long s_keys[NMS_LDS_KEYS];
    __shared__ float4 s_sel[NMS_MAX_SELECT];
    const int c = blockIdx.x;
    const int tid = threadIdx.x;
    int n = a.counts[c];
    if (n > a.N) n = a.N;
    int P = 1;
    while (P < n) P <<= 1;
    unsigned long long* gk = a.keys + size_t(c) * a.NP;
    const bool in_lds = P <= NMS_LDS_KEYS;
    unsigned long long* k = in_lds ? s_keys : gk;           // (generic pointer: LDS or global)
    if (in_lds) {
        for (int i = tid; i < P; i += NMS_THREADS) s_keys[i] = (i < n) ? gk[i] : 0ull;
    } else {
        for (int i = n + tid; i < P; i += NMS_THREADS) gk[i] = 0ull;          // NP >= next power of two of N
    }
    __syncthreads();
    // bitonic sort, descending
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += NMS_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long x = k[lo], y = k[hi];
                if (desc ? (x < y) : (x > y)) {
                    k[lo] = y;
                    k[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    if (tid >= 64) return;
    // greedy selection by wave 0
    const float4* boxes = reinterpret_cast<const float4*>(a.boxes);
    float* ob = a.out_boxes + size_t(c) * a.max_boxes * 4;
    float* os = a.out_scores + size_t(c) * a.max_boxes;
    int* oi = a.out_index + size_t(c) * a.max_boxes;
    int nsel = 0;
    for (int i = 0; i < n && nsel < a.max_boxes; ++i) {
        const unsigned long long key = k[i];
        const int idx = int(0xffffffffu - unsigned(key & 0xffffffffull));
        const float4 box = boxes[idx];
        bool sup = false;
        for (int j = tid; j < nsel; j += 64) {
            float4 sj;
            if (j < NMS_MAX_SELECT) {
                sj = s_sel[j];
            } else {                                      // beyond the LDS list: the output array itself, read from L2
                sj.x = __hip_atomic_load(ob + j * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sj.y = __hip_atomic_load(ob + j * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sj.z = __hip_atomic_load(ob + j * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sj.w = __hip_atomic_load(ob + j * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sup = sup || (iou_tf(box, sj) > a.iou_thr);
        }
        if (__ballot(sup) == 0ull) {
            if (tid == 0) {
                if (nsel < NMS_MAX_SELECT) s_sel[nsel] = box;
                __hip_atomic_store(ob + nsel * 4 + 0, box.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ob + nsel * 4 + 1, box.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ob + nsel * 4 + 2, box.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ob + nsel * 4 + 3, box.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                os[nsel] = __uint_as_float(unsigned(key >> 32));
                oi[nsel] = idx;
            }
            ++nsel;
            // the new box is visible to the wave's next reads (LDS list; beyond it the write-through stores have
            // reached L2 before the L2 loads above are issued)
            if (nsel > NMS_MAX_SELECT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (tid == 0) a.out_count[c] = nsel;
}

}  // namespace

int yolo_max_select() { return NMS_MAX_SELECT; }

void launch_yolo_eval(const YoloArgs& a, hipStream_t stream) {
    WHENET_REQUIRE(a.N > 0 && a.num_classes > 0 && a.max_boxes > 0 && a.max_boxes <= a.N, WHENET_EINVAL,
                   "yolo_eval: bad sizes (max_boxes must be 1..number of boxes)");
    WHENET_HIP_CHECK(hipMemsetAsync(a.counts, 0, size_t(a.num_classes) * sizeof(int), stream));
    hipLaunchKernelGGL(whenet_yolo_decode_kernel, dim3((a.N + 255) / 256), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(whenet_yolo_nms_kernel, dim3(a.num_classes), dim3(NMS_THREADS), 0, stream, a);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
