/*
 * libwhenet_hip.so -- C ABI of the MI355X-native WHENet inference path.
 *
 * The reference has no FFI / plugin / operator interface for this path: its boundary is
 * the Python class `WHENet` in module `whenet` (/root/reference/whenet.py:6-34), whose
 * arithmetic is delegated to Keras/TensorFlow.  Every entry point below therefore cites
 * the Python statement(s) of the reference it replaces; the ctypes binding a maintainer
 * adds on the reference side is the drop-in `whenet.py` of this repo.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns 0 (WHENET_OK) or a negative code and never throws;
 *     whenet_last_error() gives the message for the last failure on that handle
 *     (or, with a NULL handle, of the last failed whenet_create* on this thread);
 *   - all outputs are caller-allocated; inputs are never written;
 *   - a handle owns one device and 1..4 ENGINES (option "inflight", default 1), each engine with its own main
 *     stream, lane / copy streams created on demand, activation arena, hipGraphs and a copy of the device weights;
 *     a handle is NOT thread-safe: use one handle per host thread / per GPU;
 *   - crops are uint8 RGB, NHWC, [n,224,224,3] contiguous -- exactly the array the
 *     reference's callers build (demo.py:8-12, demo_video.py:21-24);
 *   - there is no CPU fallback anywhere in this library: without a gfx950 device
 *     whenet_create* fails with WHENET_ENODEV.
 */
#ifndef WHENET_HIP_H
#define WHENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* (round 4 also: options front7, head_fuse, stem_fuse.)
 * 2 (round 3): whenet_op_trunk / whenet_op_stem_dw removed with their kernels; whenet_create_postproc and
 * whenet_op_block_range added; options front_impl, se_fuse, fold12, poison.
 * 3 (round 4): whenet_launch_stat_t carries the crops and chains of the launch it describes (what whenet_profile
 * actually ran: with option "inflight" > 1 a forward is ONE chain of the whole batch).
 * 4 (round 5): dtype WHENET_F32S; whenet_normalise_table; options pw_staged, split_pw, fanout_min / _chunk / _stage / _depth,
 * host_pinned_max, host_lanes, se_fuse_tiny.  (Additions only: a version-3 caller runs unchanged.)
 * (round 6, still 4 -- options only: mb7, f2s_mask, se_fuse = 3, fanout_engines, fanout_stage = 2 | 3; the fan-out's default form is 2.) */
#define WHENET_ABI_VERSION 4
#define WHENET_API __attribute__((visibility("default")))

/* return codes (negative errno-style) */
#define WHENET_OK        0
#define WHENET_ENOENT   (-2)    /* snapshot file missing/unreadable  (Keras: OSError)        */
#define WHENET_EIO      (-5)
#define WHENET_ENOMEM   (-12)
#define WHENET_ENODEV   (-19)   /* no usable gfx950 device                                  */
#define WHENET_EINVAL   (-22)   /* bad argument / shape            (Keras: ValueError)      */
#define WHENET_EFORMAT  (-74)   /* not a WHNPACK1 snapshot or tensor shapes do not match    */
#define WHENET_EHIP     (-1000) /* HIP runtime call or kernel launch failed                  */

/* arithmetic type of activations and 1x1-conv weights (accumulation is always f32) */
#define WHENET_F32 0            /* parity configuration: <=1e-3 deg vs the float64 oracle   */
#define WHENET_F16 1            /* throughput configuration (north-star fp16)               */
#define WHENET_F32S 2           /* float32 storage and accumulation as WHENET_F32, the 1x1 products as binary16 hi/lo pairs on the
                                 * f16 matrix cores (w = hi + lo, x = hi + lo; lo_w*hi_x + hi_w*lo_x + hi_w*hi_x: ~22 bits per
                                 * product, 3 f16 MFMAs where WHENET_F32 issues 8 f32 MFMAs).  whenet_info_t.dtype reports
                                 * WHENET_F32 (the storage type); option "split_pw" 0 runs the exact-f32 kernels on such a handle.
                                 * Precondition: activations inside the binary16 range (|x| <= 65504; EfficientNet-B0's are O(100)
                                 * behind every BatchNorm) -- beyond it the hi half is inf and the angles come out NaN, never silently
                                 * wrong */

#define WHENET_IMG      224
#define WHENET_NLOGITS  252     /* 120 yaw | 66 pitch | 66 roll  (whenet.py:11-13)          */
#define WHENET_NFEAT    1280

typedef struct whenet_ctx whenet_t;

typedef struct whenet_info {
    int32_t abi_version;
    int32_t dtype;              /* WHENET_F32 / WHENET_F16 */
    int32_t device_id;
    int32_t compute_units;
    int64_t params_backbone;    /* 4,049,564 */
    int64_t params_heads;       /*   322,812 */
    int32_t n_tensors;          /* 315 */
    int32_t n_kernels_per_forward;
    int64_t macs_per_crop;      /* 384,857,312 */
    int64_t arena_bytes;        /* current activation arena */
    int32_t capacity;           /* crops the arena currently holds */
    int32_t graph_enabled;
    char    device_name[64];
    char    arch[32];
} whenet_info_t;

/* One kernel launch of a forward pass, as timed by whenet_profile(). */
typedef struct whenet_launch_stat {
    char    layer[32];          /* e.g. "b3/dw", "b3/project", "stem", "heads"              */
    char    kind[16];           /* stem | pw | dw | se | heads                              */
    char    kernel[64];         /* kernel symbol family, matches rocprofv3's kernel name    */
    double  avg_us;             /* mean duration over chains x iterations (HIP events)        */
    double  alg_bytes;          /* algorithmic bytes of this launch: in + out (+skip) once  */
    double  alg_flops;          /* 2 * MACs of this launch                                   */
    int32_t crops;              /* crops this launch processed (the chain's sub-batch)       */
    int32_t chains;             /* concurrent sub-batch chains the figures are averaged over */
} whenet_launch_stat_t;

/* ---- construction: replaces WHENet.__init__ (whenet.py:7-20): graph build +
 * model.load_weights(snapshot).  `snapshot_path` is a WHNPACK1 file (the 315 Keras arrays;
 * tools/convert_h5.py turns a Keras HDF5 snapshot into one). */
WHENET_API int whenet_create(const char* snapshot_path, int device_id, int dtype, whenet_t** out);
WHENET_API int whenet_create_from_memory(const void* snapshot, size_t nbytes, int device_id, int dtype,
                              whenet_t** out);
/* A handle WITHOUT a network: device, stream and scratch for the frame / detector stages only
 * (whenet_yolo_eval, whenet_op_crop_resize, whenet_frame_rects); every forward entry point returns WHENET_EINVAL. */
WHENET_API int whenet_create_postproc(int device_id, whenet_t** out);
WHENET_API void whenet_destroy(whenet_t* h);
WHENET_API const char* whenet_last_error(const whenet_t* h);
WHENET_API int whenet_get_info(const whenet_t* h, whenet_info_t* out);

/* options: "graph" (0/1, default 1: replay the forward as a hipGraph),
 *          "fuse_front" (0/1, default 1: expand 1x1 + depthwise as ONE kernel per block, the
 *                  expanded tensor stays in LDS; 0 = two launches through HBM),
 *          "se_fuse" (0..3, default 1: second half of the squeeze-excite block inside the project conv's launch -- its
 *                  workgroups compute the gate of their own rows' crops -- 0 = never (a squeeze-excite launch per block,
 *                  51 launches per forward, 50 with fold12), 2 = on every block with a fused front kernel (36 / 35 launches), 1 = on the
 *                  blocks where that is the faster schedule; the results are bitwise the same;
 *                  3 (round 6, f16 / f32s handles) = 1 plus blocks 7-16, whose project conv then computes the gate of each wave's own
 *                  channel groups on the matrix cores: ten launches fewer, another rounding path, measured slower),
 *          "front_impl" (0..2, default 1: which fused kernel a handle uses -- 0 = front.hip (depthwise taps
 *                  as f32 VALU FMAs) on every block, 2 = the kernel with the taps as Toeplitz products on the matrix
 *                  cores wherever it exists (f16: front2.hip, f16 tap weights; f32s: front2s.hip, exact-f32 or hi/lo taps;
 *                  blocks 2-12), 1 = per layer, whichever was measured faster; exact-f32 handles always run front.hip),
 *          "front7" (0/1, default 1: with front_impl = 1, blocks 13-16 (7 x 7 maps) of an f16 handle run front7.hip -- a GROUP of
 *                  2 or 4 crops per workgroup, the image-only LDS tile, the chunk's expand weights staged once in LDS; the group
 *                  size follows the launch size and changes no bit of a crop's result; 0 = round 3's per-layer choice),
 *          "head_fuse" (0/1, default 1: the head conv (whenet.py:8, last layer) pools its own output, the
 *                  GlobalAveragePooling2D of whenet.py:10, in one kernel (head7.hip; f16, and f32 since round 4b); 0 = conv, then pooling
 *                  inside the heads kernel),
 *          "stem_fuse" (0/1, default 1: handles of EITHER dtype fed uint8 crops compute the stem conv (whenet.py:8, first layer) inside block 1's
 *                  depthwise kernel (stemdw.hip): the 112 x 112 x 32 stem output never reaches HBM, one launch less; results are
 *                  BITWISE those of the two kernels (f16 and f32); 0 = stem.hip, then dw.hip.  The float32-input entry points keep the two.
 *                  f32 precision note: conv-epilogue Swish uses v_exp_f32 / v_rcp_f32 (1-ulp hardware forms, device_math.h) in both
 *                  dtypes since round 4; build with -DWHENET_PRECISE_CONV_SWISH=1 for expf + IEEE division),
 *          "fold12" (0/1, default 1: f16 handles whose block 2 runs front2.hip feed that kernel from block 1's depthwise
 *                  output, with block 1's project conv (linear) composed into block 2's expand weights when the
 *                  snapshot is loaded -- one launch and a 112x112x16 round trip through HBM less; 0 = the two convs
 *                  as two steps.  Same function, different rounding points: results agree to f16 rounding),
 *          "poison" (0/1, default 0, debug: the activation arena is filled with NaN bit patterns before every forward --
 *                  a kernel that reads what the forward did not write shows up in the results),
 *          "lanes" (1..8, default 2: concurrent sub-batch chains per forward, never fewer than 16 crops each),
 *          "lane_graphs" (0/1, default 0: 1 = one graph per lane launched on its own stream instead of
 *                  one forked graph; measured equal),
 *          "inflight" (1..4, default 1: n > 1 gives the handle n engines -- own streams, activation
 *                  arena, graphs, replicated weights -- and spreads whenet_forward_u8_device calls with
 *                  stream == NULL and whenet_submit_* calls over them round-robin, each forward as
 *                  one chain; independent forwards then overlap on the GPU (a forward is a chain of 46
 *                  (f16) / 49 (f32) dependent launches: whenet_info_t.n_kernels_per_forward).  The caller gives every forward in flight its own output
 *                  buffers; whenet_sync waits for all of them.  Results are bitwise those of n = 1),
 *          "fanout_min" (>= 0, default 256: a blocking whenet_forward_u8 of at least this many crops is cut into
 *                  "fanout_chunk"-crop forwards (default 128; the first two are half-size so that the GPU starts early); chunk c goes
 *                  to engine c % E through that engine's pinned submission slots, at most "fanout_depth" (1..4, default 2)
 *                  outstanding per engine, E = max("inflight", "fanout_engines").  "fanout_engines" (1..4, default 2): engines beyond
 *                  the handle's own are created on the first such call and belong to the fan-out alone -- "inflight", the round-robin
 *                  of the other entry points and their chains per forward are not touched.  Results are bitwise those of one forward.
 *                  0 = never.  "fanout_stage": 2 (default, round 6) = the caller's array is registered with the runtime for the
 *                  duration of the call (hipHostRegister / hipHostUnregister: 2 us - 0.2 ms), so the chunk copies are asynchronous:
 *                  one host thread enqueues everything, all copies travel in order on ONE stream at the link's full rate; an
 *                  array the runtime will not register -- or whose pages overlap an array another handle's call holds -- takes
 *                  form 1.  0 / 1 = round 5's forms, one host thread per engine: chunks copied into pinned staging first / the
 *                  runtime's pageable path.  -1 = calibrate 0 against 1: both forms run once untimed, then twice each timed, the
 *                  faster serves every later call.  3 (probe) = as 2 with a copy stream per engine),
 *          "host_pinned_max" (0..4096, default 8: a blocking whenet_forward_u8 of at most this many crops travels through a
 *                  pinned staging slot -- one asynchronous H2D, the forward, three asynchronous D2H, ONE wait -- instead of
 *                  four synchronous copies from / to the caller's pageable memory: the latency path of the reference's
 *                  per-head call shape (demo.py:14, demo_video.py:27)),
 *          "se_fuse_tiny" (0..64, default 0: chains of at most this many crops behave as se_fuse = 2),
 *          "host_lanes" (1..8, default 2: chains a BLOCKING host forward runs as; "lanes" sets both),
 *          "min_lane_crops" (>= 1, default 16: the smallest sub-batch a lane may get; "lanes" is cut down until every
 *                  lane has at least this many crops.  Tests set 1 to force several lanes on small batches),
 *          "repeat" (1..16, default 1, measurement only: the captured graph holds this many back-to-back copies of
 *                  the forward, so that one graph launch times `repeat` forwards without the launch boundary
 *                  between them; results are those of one forward),
 *          "pw_staged" (0/1, default 1: the K >= 1152 and the 14 x 14 K = 672 project GEMMs of f16 / f32s handles fetch their
 *                  activation rows coalesced -- 8 rows x 128 contiguous bytes per wave-instruction -- and hand them to the matrix
 *                  cores through per-wave LDS (whenet_pw_splitk_staged_kernel) instead of loading MFMA fragments (16 bytes of each
 *                  of 32 rows: 32 cache lines per KB) from global memory; 0 = the direct kernel everywhere.  Another summation
 *                  order: results agree to rounding.  Chosen by layer, never by batch),
 *          "split_pw" (0/1, default 1, WHENET_F32S handles only: 0 runs the exact-f32 kernels -- bitwise a WHENET_F32 handle),
 *          "xcd_map" (bit mask 0..7, default 7: the workgroups of a launch that read the SAME input -- the channel chunks of one crop and
 *                  tile -- are dealt to one XCD (one L2) instead of round-robin over the eight: 1 = the fused expand+depthwise kernels,
 *                  2 = the 7 x 7 form, 4 = the head conv.  A relabelling of workgroups: the same bits.  +0.8 % with the chip full, slower for
 *                  one small forward alone -- so it is applied to launches of >= 128 crops and to every launch of a handle with
 *                  "inflight" > 1),
 *          "mb7" (0/1, default 0, WHENET_F16 handles: blocks 13-16 -- the 7 x 7 stage -- run as ONE launch each, one workgroup per crop
 *                  with the expanded tensor, the depthwise output, the squeeze-excite gate in LDS (mb7.hip) instead of front + squeeze-
 *                  excite + project launches.  29 us per launch whatever the batch up to 256 crops: +3 % at batch 512, +-1 % at 64
 *                  crops x 3 in flight, -4 % one forward at a time, +60 us at batch 1 -- the schedule must not depend on the batch, so it
 *                  is not the default.  Another rounding path of the same function: binary16 squeeze-excite kernels, other orders of
 *                  summation; bitwise independent of the batch like every other schedule),
 *          "pw_impl" (0 = MFMA kernels, 1 = scalar-FMA check kernels, same results class) */
WHENET_API int whenet_set_option(whenet_t* h, const char* key, long value);

/* ---- the hot path: replaces WHENet.get_angle (whenet.py:22-34) =
 * normalise (23-26) -> Model.predict (27) -> softmax-expectation decode (28-33).
 *   crops   uint8 [n,224,224,3] RGB
 *   ypr     float [n,3]   yaw, pitch, roll in degrees           (required)
 *   argmax  int32 [n,3]   argmax bin of each head's logits      (may be NULL)
 *   logits  float [n,252] what Model.predict returns, concatenated (may be NULL)
 * Host-pointer form: copies in, runs, copies out, returns when the results are in place. */
WHENET_API int whenet_forward_u8(whenet_t* h, const uint8_t* crops, int n,
                      float* ypr, int32_t* argmax, float* logits);

/* The same path for REAL-VALUED input: `image` is the normalised float32 image [n,224,224,3] that
 * the reference hands to Model.predict (whenet.py:27) -- i.e. (img/255 - mean)/std computed by
 * the caller as whenet.py:23-26 does (float64, then cast).  whenet.py:25 divides any numeric array
 * by 255, so crops that are not 8-bit integers (no byte LUT applies) take this entry point; the
 * drop-in get_angle routes them here.  Host pointers, blocking, eager launches.  An f16 handle computes in
 * binary16 activations: normalised inputs beyond +-65504 saturate there (an f32 handle takes any finite float32). */
WHENET_API int whenet_forward_f32(whenet_t* h, const float* image, int n,
                       float* ypr, int32_t* argmax, float* logits);

/* Device-pointer form: all pointers are device memory on the handle's GPU; the work is
 * enqueued on `stream` (a hipStream_t; NULL = the handle's own stream, or with option "inflight"
 * > 1 the next of the handle's engines) and the call returns without waiting.  This is the form
 * bench.py times (inputs resident in HBM).
 * ORDERING: an engine has ONE activation arena, so two forwards of the same engine must not overlap.
 * Calls with stream == NULL are ordered by the engine's own stream.  A caller that passes its own
 * streams must order successive calls itself (same stream, or an event between them): the library does
 * not insert cross-stream dependencies, and two forwards enqueued on different caller streams of one
 * engine would race on the arena.  For concurrent forwards use option "inflight" (one arena per engine). */
WHENET_API int whenet_forward_u8_device(whenet_t* h, const uint8_t* d_crops, int n,
                             float* d_ypr, int32_t* d_argmax, float* d_logits, void* stream);
WHENET_API int whenet_sync(whenet_t* h);

/* Pipelined host form for per-frame callers (demo_video.py:56-58 runs one get_angle per
 * detected head, sequentially): submit copies the crops into pinned memory and enqueues
 * H2D + forward + D2H on the handle's stream; collect waits for that submission. Up to
 * WHENET_MAX_INFLIGHT submissions may be outstanding, collected in FIFO order. */
#define WHENET_MAX_INFLIGHT 4
WHENET_API int whenet_submit_u8(whenet_t* h, const uint8_t* crops, int n, int* ticket);
WHENET_API int whenet_collect(whenet_t* h, int ticket, float* ypr, int32_t* argmax, float* logits);

/* ---- per-frame pre-processing on the device: replaces the host work of
 * process_detection (demo_video.py:13-24) and crop_and_pred (demo.py:8-11) for ALL heads of a
 * frame, and the per-head get_angle calls of demo_video.py:56-58, by one submission.
 *
 * whenet_frame_rects: the bbox margin arithmetic of demo_video.py:13-19 (float32, as YOLO's boxes
 * are; y_max / x_max use the already-moved y_min / x_min) followed by the int() truncation and
 * slice clipping of demo_video.py:21.  bboxes [k,4] = (y_min, x_min, y_max, x_max) as
 * YOLO.detect returns them; rects [k,4] = (y0, x0, y1, x1), the window img[y0:y1, x0:x1].
 * Pure host arithmetic, no GPU needed.  (demo.py:9-10 uses its integer bbox as the window directly.) */
#define WHENET_RGB 0            /* frame is already RGB            (demo.py:8 converts first)       */
#define WHENET_BGR 1            /* frame is BGR as cv2 delivers it (demo_video.py:22 swaps per crop) */
WHENET_API int whenet_frame_rects(int frame_h, int frame_w, const float* bboxes, int k, int32_t* rects);
/* whenet_normalise_table: the image of /root/reference/whenet.py:23-26 (`img/255`, `(img-mean)/std` in float64) followed by
 * Keras' cast to float32 (whenet.py:27) for every byte value: lut[c*256 + v], c = R,G,B.  This is the table the stem
 * kernels apply to uint8 crops.  Pure host arithmetic, no GPU needed. */
WHENET_API int whenet_normalise_table(float lut[768]);
/* frame uint8 [frame_h, frame_w, 3] (host).  Copies the frame into pinned memory and enqueues
 * H2D(frame) -> crop + colour order + cv2.resize-compatible bilinear to [k,224,224,3] on the device
 * -> forward -> D2H of the results; whenet_collect(ticket, ...) returns the k heads' outputs in
 * rect order.  k = 0 (no head in the frame) is valid.  An empty or out-of-frame window is
 * WHENET_EINVAL (cv2.resize raises on an empty source). */
WHENET_API int whenet_submit_frame(whenet_t* h, const uint8_t* frame, int frame_h, int frame_w, int channel_order,
                        const int32_t* rects, int k, int* ticket);
/* the crop/resize kernel alone (host pointers): crops uint8 [k,224,224,3] RGB, exactly the array
 * get_angle would have been handed */
WHENET_API int whenet_op_crop_resize(whenet_t* h, const uint8_t* frame, int frame_h, int frame_w, int channel_order,
                          const int32_t* rects, int k, uint8_t* crops);

/* ---- the detector's post-processing on the device: replaces yolo_eval (yolo_v3/model.py:193-232 =
 * yolo_head :125-150, yolo_correct_boxes :153-178, yolo_boxes_and_scores :181-190, per-class
 * tf.image.non_max_suppression), which the reference runs inside sess.run (yolo_postprocess.py:198-204).
 *   feats        num_layers host pointers to the detector's output maps, float32 [grid_h][grid_w][3*(5+num_classes)]
 *                (batch of one, as YOLO.detect feeds it), coarsest map first like the Keras model's outputs
 *   anchors      num_anchors x (w, h) as in yolo_anchors.txt; 3 maps need 9 anchors, 2 maps ("tiny") 6
 *   image_h/w    size of the original image (`input_image_shape`); the network input is 32 x the first map's grid
 *   max_boxes    per class, any value >= 1 as the reference (default 20; more than 256 selections per class spill from
 *                LDS to the output array); score: `>= score_threshold`; NMS drops IoU `> iou_threshold`
 *   boxes        float [num_classes*max_boxes][4]  y_min, x_min, y_max, x_max in image pixels (not clipped, as the reference)
 *   scores, classes, index (may be NULL: the box's position in the concatenated (map, y, x, anchor) list)
 *   count        number of detections written, class by class, descending score inside a class
 *   all_boxes [N][4], all_scores [N][num_classes] (may be NULL): every decoded box / score, for tests
 * Equal scores are taken lower index first (TensorFlow leaves ties to its heap). */
WHENET_API int whenet_yolo_eval(whenet_t* h, const float* const* feats, const int* grid_h, const int* grid_w,
                     int num_layers, const float* anchors, int num_anchors, int num_classes, float image_h,
                     float image_w, float score_threshold, float iou_threshold, int max_boxes, float* boxes,
                     float* scores, int32_t* classes, int32_t* index, int* count, float* all_boxes,
                     float* all_scores);

/* ---- measurement: run `iters` eager forwards of `n` device-resident crops exactly as the
 * timed path runs them (same concurrent sub-batch chains, same streams) with ONE HIP event
 * recorded on the chain's stream between consecutive kernel launches; a launch's time is
 * previous event -> its own event.  Fills up to `cap` entries, one per launch of a chain in
 * launch order, averaged over chains and iterations; alg_bytes / alg_flops are those of one
 * chain's launch (its sub-batch).  *count = launches per chain. */
WHENET_API int whenet_profile(whenet_t* h, const uint8_t* d_crops, int n, int iters,
                   whenet_launch_stat_t* stats, int cap, int* count);

/* ---- single-stage entry points (host pointers, float32 activations in/out, converted to
 * the handle's dtype on the device).  They run exactly the kernels the forward uses, on
 * caller-supplied inputs, so each kernel can be compared with the oracle on every layer
 * shape.  Any output pointer may be NULL. */
/* stem: normalise + Conv3x3/s2 + BN + Swish.  out [n,112,112,32] */
WHENET_API int whenet_op_stem(whenet_t* h, const uint8_t* crops, int n, float* out);
/* MBConv block `index` (1..16) on input [n,H,W,Cin]:
 *   expand_out [n,H,W,Cexp] (NULL for block 1), dw_out [n,Ho,Wo,Cexp], gate [n,Cexp],
 *   out [n,Ho,Wo,Cout] (after project + BN + skip) */
WHENET_API int whenet_op_block(whenet_t* h, int index, const float* in, int n,
                    float* expand_out, float* dw_out, float* gate, float* out);
/* MBConv blocks first..last (1 <= first <= last <= 16) chained exactly as the forward pass chains them -- including
 * option fold12 (block 1's project folded into block 2's expand) when the range holds blocks 1 and 2 -- on input
 * [n,H,W,Cin] of block `first`; out [n,Ho,Wo,Cout] of block `last`. */
WHENET_API int whenet_op_block_range(whenet_t* h, int first, int last, const float* in, int n, float* out);
/* head: Conv1x1(1280)+BN+Swish + GAP + Dense heads + decode on input [n,7,7,320]:
 *   feat [n,1280], logits [n,252], ypr [n,3], argmax [n,3] */
WHENET_API int whenet_op_head(whenet_t* h, const float* in, int n,
                   float* feat, float* logits, float* ypr, int32_t* argmax);
/* decode only (whenet.py:28-33) on caller logits [n,252] -> ypr [n,3], argmax [n,3] */
WHENET_API int whenet_op_decode(whenet_t* h, const float* logits, int n, float* ypr, int32_t* argmax);

/* layer geometry as the engine sees it (for cross-checking against whenet_hip/spec.py):
 * fills out[0..7] = {k, stride, expand, cin, cout, h_in, h_out, se_reduced} for block
 * `index` (1..16). */
WHENET_API int whenet_block_spec(int index, int32_t out[8]);

/* the depthwise tile plan of block `index` for `dtype` (pure host logic; no GPU needed):
 * out = {threads, CV, TH, NSX, tiles_x, tiles_y, chunks, IH, IW, lds_bytes, pad_before, C} */
WHENET_API int whenet_dw_plan(int dtype, int index, int32_t out[12]);

/* the fused expand+depthwise tile plan of block `index` (2..16) for `dtype` (pure host logic):
 * out = {threads, CC, TH, NSX, tiles_x, tiles_y, chunks, EH, EW, lds_bytes, w_off, Cexp} */
WHENET_API int whenet_front_plan(int dtype, int index, int32_t out[12]);

/* raw device-memory helpers so a host without torch can use the device-pointer form */
WHENET_API int whenet_device_alloc(whenet_t* h, size_t nbytes, void** d_ptr);
WHENET_API int whenet_device_free(whenet_t* h, void* d_ptr);
WHENET_API int whenet_memcpy_h2d(whenet_t* h, void* d_dst, const void* src, size_t nbytes);
WHENET_API int whenet_memcpy_d2h(whenet_t* h, void* dst, const void* d_src, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* WHENET_HIP_H */
