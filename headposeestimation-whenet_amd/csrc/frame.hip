// Per-head pre-processing of a video frame on the GPU: crop window -> (BGR->RGB) -> bilinear
// resize to 224x224 uint8, written straight into the forward's input buffer.
//
// Reference: /root/reference/demo_video.py:13-24 (`process_detection`: bbox margins, slice,
// cv2.cvtColor(BGR2RGB), cv2.resize(.., (224, 224))) and /root/reference/demo.py:8-11.  The
// reference does this per head on the host and ships 150,528 B per head to the device; here the
// frame crosses PCIe once and every head is cut out of it by one launch.
//
// cv2.resize's default (INTER_LINEAR on 8-bit data) is OpenCV's fixed-point bilinear
// (modules/imgproc/src/resize.cpp): coefficient tables in 11-bit fixed point built from
// fx = (float)((dx + 0.5) * scale - 0.5), a horizontal pass into int32, a vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2x shrink is switched
// to INTER_AREA (2x2 box mean).  The tables are integer data computed on the HOST with the same
// float/double operations as OpenCV (build_crop_plan), so the kernel is pure integer arithmetic
// and the result is bit-exact with the restatement in oracle/preprocess_oracle.py.
#include <cmath>

#include "kernels.h"

namespace whenet {

namespace {

constexpr int OUT = IMG;                    // 224
constexpr int COEF_SCALE = 1 << 11;         // INTER_RESIZE_COEF_SCALE

__global__ __launch_bounds__(256) void whenet_crop_resize_kernel(const uint8_t* __restrict__ frame, int fw,
                                                                 int swap_rb, const int32_t* __restrict__ plan,
                                                                 uint8_t* __restrict__ out) {
    const int dy = blockIdx.x, crop = blockIdx.y, dx = threadIdx.x;
    if (dx >= OUT) return;
    const int32_t* P = plan + size_t(crop) * CROP_PLAN_INTS;
    const int y0 = P[0], x0 = P[1], ch = P[2], cw = P[3], area2x = P[4], xmax = P[5];
    const int32_t* T = P + 8;               // xofs | a0 | a1 | yofs | b0 | b1, 224 ints each
    uint8_t* o = out + ((size_t(crop) * OUT + dy) * OUT + dx) * 3;
    if (area2x) {
        const uint8_t* r0 = frame + (size_t(y0 + 2 * dy) * fw + x0 + 2 * dx) * 3;
        const uint8_t* r1 = r0 + size_t(fw) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = swap_rb ? 2 - c : c;
            o[c] = uint8_t((int(r0[cs]) + int(r0[3 + cs]) + int(r1[cs]) + int(r1[3 + cs]) + 2) >> 2);
        }
        return;
    }
    const int sx = T[dx], a0 = T[OUT + dx], a1 = T[2 * OUT + dx];
    const int sy = T[3 * OUT + dy], b0 = T[4 * OUT + dy], b1 = T[5 * OUT + dy];
    const int ya = sy < 0 ? 0 : (sy > ch - 1 ? ch - 1 : sy);
    const int yb = sy + 1 < 0 ? 0 : (sy + 1 > ch - 1 ? ch - 1 : sy + 1);
    const bool two = dx < xmax;             // (sx + 1 < cw by construction)
    const uint8_t* ra = frame + (size_t(y0 + ya) * fw + x0 + sx) * 3;
    const uint8_t* rb = frame + (size_t(y0 + yb) * fw + x0 + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = swap_rb ? 2 - c : c;
        const int d0 = two ? int(ra[cs]) * a0 + int(ra[3 + cs]) * a1 : int(ra[cs]) * COEF_SCALE;
        const int d1 = two ? int(rb[cs]) * a0 + int(rb[3 + cs]) * a1 : int(rb[cs]) * COEF_SCALE;
        int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[c] = uint8_t(v);
    }
    (void)cw;
}

// saturate_cast<short>(float): cvRound (round half to even, the default FP rounding mode) + saturation
inline int32_t to_short(float v) {
    long r = lrintf(v);
    if (r < -32768) r = -32768;
    if (r > 32767) r = 32767;
    return int32_t(r);
}

void axis_tables(int src, bool horizontal, int32_t* ofs, int32_t* c0, int32_t* c1, int* nmax) {
    const double inv_scale = double(OUT) / double(src);
    const double scale = 1.0 / inv_scale;
    *nmax = OUT;
    for (int d = 0; d < OUT; ++d) {
        float f = float((d + 0.5) * scale - 0.5);
        int s = int(std::floor(f));
        f -= float(s);
        if (horizontal) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s + 1 >= src) {
                if (d < *nmax) *nmax = d;
                if (s >= src - 1) { f = 0.f; s = src - 1; }
            }
        }
        ofs[d] = s;
        c0[d] = to_short((1.f - f) * float(COEF_SCALE));
        c1[d] = to_short(f * float(COEF_SCALE));
    }
}

}  // namespace

// demo_video.py:13-19 in float32 (YOLO hands back float32 boxes, yolo_postprocess.py:198-205),
// including the order dependence (y_max / x_max use the already-moved y_min / x_min), then the
// int() truncation and slice clipping of demo_video.py:21.
void frame_box_rect(int frame_h, int frame_w, const float bbox[4], int32_t rect[4]) {
    float y_min = bbox[0], x_min = bbox[1], y_max = bbox[2], x_max = bbox[3];
    const float fh = float(frame_h), fw = float(frame_w);
    {
        const float v = y_min - std::fabs(y_min - y_max) / 10.0f;
        y_min = (v > 0.f) ? v : 0.f;                      // max(0, v): returns 0 unless v > 0
    }
    {
        const float v = y_max + std::fabs(y_min - y_max) / 10.0f;
        y_max = (v < fh) ? v : fh;                        // min(H, v): returns H unless v < H
    }
    {
        const float v = x_min - std::fabs(x_min - x_max) / 5.0f;
        x_min = (v > 0.f) ? v : 0.f;
    }
    {
        const float v = x_max + std::fabs(x_min - x_max) / 5.0f;
        x_max = (v < fw) ? v : fw;
    }
    if (!(x_max < fw)) x_max = fw;
    int y0 = int(y_min), x0 = int(x_min), y1 = int(y_max), x1 = int(x_max);
    if (y1 > frame_h) y1 = frame_h;
    if (x1 > frame_w) x1 = frame_w;
    rect[0] = y0; rect[1] = x0; rect[2] = y1; rect[3] = x1;
}

void build_crop_plan(const int32_t rect[4], int32_t* plan) {
    const int y0 = rect[0], x0 = rect[1], ch = rect[2] - rect[0], cw = rect[3] - rect[1];
    WHENET_REQUIRE(ch > 0 && cw > 0, WHENET_EINVAL, "empty crop window (cv2.resize rejects an empty source)");
    plan[0] = y0; plan[1] = x0; plan[2] = ch; plan[3] = cw;
    plan[4] = (ch == 2 * OUT && cw == 2 * OUT) ? 1 : 0;
    plan[6] = 0; plan[7] = 0;
    int32_t* T = plan + 8;
    int xmax = OUT, ymax = OUT;
    axis_tables(cw, true, T, T + OUT, T + 2 * OUT, &xmax);
    axis_tables(ch, false, T + 3 * OUT, T + 4 * OUT, T + 5 * OUT, &ymax);
    plan[5] = xmax;
}

void launch_crop_resize(const uint8_t* d_frame, int fw, int swap_rb, const int32_t* d_plan, int k, uint8_t* d_out,
                        hipStream_t stream) {
    if (k <= 0) return;
    hipLaunchKernelGGL(whenet_crop_resize_kernel, dim3(OUT, k), dim3(256), 0, stream, d_frame, fw, swap_rb, d_plan,
                       d_out);
    WHENET_HIP_CHECK(hipGetLastError());
}

}  // namespace whenet
