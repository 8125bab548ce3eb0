// Detector-stage kernels around the convolution engine (BASELINE config 5): what torchvision's Faster R-CNN
// (the detector /root/reference models/detection.py:23-29 instantiates: fasterrcnn_resnet50_fpn) does between its
// convolutions -- FPN top-down merge, RPN anchor decode, non-maximum suppression, multi-level RoIAlign, the box head's
// softmax / decode / threshold.  torchvision is not part of this build (absent from the image): the algorithms follow its
// published source (torchvision/models/detection/{rpn,roi_heads,_utils,anchor_utils}.py, torchvision/ops/{poolers,roi_align,
// boxes}.py); the oracle restates the same in torch (oracle/ref_cpu.py, UNPINNED).
// All tensors NHWC float32 unless stated; none of these kernels is on the training hot path.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// FPN top-down step: lateral[n,y,x,:] += top[n, src(y), src(x), :], src(i) = min(floor(i * in / out), in - 1)
// (F.interpolate(mode="nearest", size=lateral.shape[-2:]) + add, feature_pyramid_network.py forward)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_nearest_add_kernel(const float* __restrict__ top, int top_cs, int th, int tw,
                                                                   float* __restrict__ lat, int lat_cs, int N, int H, int W, int C4) {
    const float sy = (float)th / (float)H, sx = (float)tw / (float)W;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % C4);
        int64_t p = i / C4;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int n = (int)(p / H);
        const int yy = adh_min_i((int)floorf(y * sy), th - 1), xx = adh_min_i((int)floorf(x * sx), tw - 1);
        const f32x4 t = *reinterpret_cast<const f32x4*>(top + (((int64_t)n * th + yy) * tw + xx) * top_cs + q * 4);
        f32x4* d = reinterpret_cast<f32x4*>(lat + (((int64_t)n * H + y) * W + x) * lat_cs + q * 4);
        *d = *d + t;
    }
}

extern "C" int adh_upsample_nearest_add(void* stream, const float* top, int top_cs, int th, int tw, float* lateral, int lat_cs, int N,
                                        int H, int W, int C) {
    if (!top || !lateral || N < 1 || H < 1 || W < 1 || th < 1 || tw < 1 || C < 4 || (C & 3) || (top_cs & 3) || (lat_cs & 3)) return ADH_E_ARG;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample_nearest_add_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       top, top_cs, th, tw, lateral, lat_cs, N, H, W, C / 4);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// RPN: anchors of one pyramid level (anchor_utils.py: base anchors round([-w,-h,w,h]/2) shifted by (x*stride_w, y*stride_h), position
// major, anchor minor -- the order permute_and_flatten gives the head outputs in NHWC), BoxCoder(1,1,1,1).decode with dw / dh clamped at
// log(1000/16), clip_boxes_to_image.  cls [N,H,W,>=A] logits, reg [N,H,W,>=4A] deltas (a*4 + c) -> boxes [N,H*W*A,4], logits [N,H*W*A]
// ------------------------------------------------------------------------------------------------
#define ADH_BBOX_XFORM_CLIP 4.135166556742356f   // log(1000 / 16)

__global__ __launch_bounds__(256) void rpn_decode_kernel(const float* __restrict__ cls, int cls_cs, const float* __restrict__ reg, int reg_cs,
                                                         int N, int H, int W, int A, int stride_h, int stride_w,
                                                         const float* __restrict__ base, float img_h, float img_w,
                                                         float* __restrict__ boxes, float* __restrict__ logits) {
    const int64_t total = (int64_t)N * H * W * A;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int a = (int)(i % A);
        int64_t p = i / A;
        const int x = (int)(p % W);
        const int64_t pix = p;
        p /= W;
        const int y = (int)(p % H);
        const float ax1 = base[a * 4 + 0] + (float)(x * stride_w), ay1 = base[a * 4 + 1] + (float)(y * stride_h);
        const float ax2 = base[a * 4 + 2] + (float)(x * stride_w), ay2 = base[a * 4 + 3] + (float)(y * stride_h);
        const float* d = reg + pix * reg_cs + a * 4;
        const float w = ax2 - ax1, h = ay2 - ay1, cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
        const float dw = fminf(d[2], ADH_BBOX_XFORM_CLIP), dh = fminf(d[3], ADH_BBOX_XFORM_CLIP);
        const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        float* o = boxes + i * 4;
        o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.f), img_w);
        o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.f), img_h);
        o[2] = fminf(fmaxf(pcx + 0.5f * pw, 0.f), img_w);
        o[3] = fminf(fmaxf(pcy + 0.5f * ph, 0.f), img_h);
        logits[i] = cls[pix * cls_cs + a];
    }
}

extern "C" int adh_rpn_decode(void* stream, const float* cls, int cls_cs, const float* reg, int reg_cs, int N, int H, int W, int A,
                              int stride_h, int stride_w, const float* base_anchors, float img_h, float img_w, float* boxes, float* logits) {
    if (!cls || !reg || !base_anchors || !boxes || !logits || N < 1 || H < 1 || W < 1 || A < 1 || cls_cs < A || reg_cs < 4 * A) return ADH_E_ARG;
    const int64_t total = (int64_t)N * H * W * A;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream, cls, cls_cs,
                       reg, reg_cs, N, H, W, A, stride_h, stride_w, base_anchors, img_h, img_w, boxes, logits);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Grouped non-maximum suppression (torchvision.ops.batched_nms semantics: boxes of different groups never suppress each other):
// boxes [M,4] SORTED by descending score, group [M] int32.  Pass 1: mask[i][w] bit b = box j = 64 w + b (j > i) has the group of i and
// IoU(i, j) > thr (IoU = inter / (area_i + area_j - inter), ops/boxes.py).  Pass 2: one workgroup walks the boxes in order and ORs
// the mask rows of the kept ones into the removed set (LDS) -- keep[i] = 1 / 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ group, int M, float thr,
                                                      unsigned long long* __restrict__ mask, int words) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;                           // only j > i matters
    __shared__ float cbx[64][4];
    __shared__ int cgr[64];
    const int t = threadIdx.x;
    const int j0 = cb * 64;
    if (j0 + t < M) {
        cbx[t][0] = boxes[(j0 + t) * 4 + 0]; cbx[t][1] = boxes[(j0 + t) * 4 + 1];
        cbx[t][2] = boxes[(j0 + t) * 4 + 2]; cbx[t][3] = boxes[(j0 + t) * 4 + 3];
        cgr[t] = group[j0 + t];
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= M) return;
    const float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    const float ai = (x2 - x1) * (y2 - y1);
    const int gi = group[i];
    unsigned long long bits = 0;
    const int nj = adh_min_i(64, M - j0);
    for (int b = 0; b < nj; ++b) {
        const int j = j0 + b;
        if (j <= i || cgr[b] != gi) continue;
        const float w = fmaxf(fminf(x2, cbx[b][2]) - fmaxf(x1, cbx[b][0]), 0.f);
        const float h = fmaxf(fminf(y2, cbx[b][3]) - fmaxf(y1, cbx[b][1]), 0.f);
        const float inter = w * h;
        const float aj = (cbx[b][2] - cbx[b][0]) * (cbx[b][3] - cbx[b][1]);
        if (inter / (ai + aj - inter) > thr) bits |= 1ull << b;
    }
    mask[(int64_t)i * words + cb] = bits;
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int M, int words, int* __restrict__ keep) {
    extern __shared__ unsigned long long removed[];
    for (int w = threadIdx.x; w < words; w += 256) removed[w] = 0ull;
    __syncthreads();
    for (int i = 0; i < M; ++i) {
        const bool gone = (removed[i >> 6] >> (i & 63)) & 1ull;     // uniform: every thread reads the same word
        if (threadIdx.x == 0) keep[i] = gone ? 0 : 1;
        if (!gone) {
            __syncthreads();
            for (int w = (i >> 6) + threadIdx.x; w < words; w += 256) removed[w] |= mask[(int64_t)i * words + w];
            __syncthreads();
        }
    }
}

extern "C" int adh_nms_words(int M) { return (M + 63) / 64; }

extern "C" int adh_nms_sorted(void* stream, const float* boxes, const int32_t* group, int M, float iou_threshold, void* mask_workspace,
                              int32_t* keep) {
    if (!boxes || !group || !mask_workspace || !keep || M < 1 || M > 16384) return ADH_E_ARG;
    const int words = (M + 63) / 64;
    unsigned long long* mask = (unsigned long long*)mask_workspace;
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(mask, 0, (size_t)M * words * 8, s);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, s, boxes, group, M, iou_threshold, mask, words);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), words * 8, s, mask, M, words, keep);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// MultiScaleRoIAlign (ops/poolers.py + ops/roi_align.py, aligned = False, output 7x7, sampling_ratio 2) over up to four pyramid levels:
// level = clamp(floor(4 + log2(sqrt(area) / 224) + 1e-6), 2, 5) - 2; spatial scale of level l = scales[l]; per bin the mean of 2 x 2
// bilinear samples.  rois [R,5] = (image, x1, y1, x2, y2); out [R][C * 49] CHANNEL-major (torch's x.flatten(1) of [R, C, 7, 7]), so
// that fc6's weight keeps torchvision's column order.  One workgroup per RoI, thread = (bin, channel quad).
// ------------------------------------------------------------------------------------------------
// (struct adh_fpn_levels: include/adam_dehaze_hip.h)

__device__ __forceinline__ f32x4 roi_bilinear(const float* __restrict__ f, int H, int W, int cs, int n, float y, float x, int q) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return f32x4{0.f, 0.f, 0.f, 0.f};
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = f + (int64_t)n * H * W * cs + q * 4;
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(b + ((int64_t)yl * W + xl) * cs), v2 = *reinterpret_cast<const f32x4*>(b + ((int64_t)yl * W + xh) * cs);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(b + ((int64_t)yh * W + xl) * cs), v4 = *reinterpret_cast<const f32x4*>(b + ((int64_t)yh * W + xh) * cs);
    return (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4;
}

__global__ __launch_bounds__(256) void roi_align_fpn_kernel(const adh_fpn_levels L, const float* __restrict__ rois, int C, float* __restrict__ out) {
    const int r = blockIdx.x;
    const float* roi = rois + (int64_t)r * 5;
    const int n = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float s = sqrtf((x2 - x1) * (y2 - y1));
    int lvl = (int)floorf(4.f + log2f(s / 224.f) + 1e-6f);
    lvl = adh_min_i(adh_max_i(lvl, 2), 5) - 2;
    lvl = adh_min_i(lvl, L.nlevels - 1);
    const float sc = L.scale[lvl];
    const int H = L.H[lvl], W = L.W[lvl], cs = L.cs[lvl];
    const float* f = L.f[lvl];
    const float rsw = x1 * sc, rsh = y1 * sc;
    const float rw = fmaxf(x2 * sc - rsw, 1.f), rh = fmaxf(y2 * sc - rsh, 1.f);
    const float bw = rw / 7.f, bh = rh / 7.f;
    const int C4 = C / 4;
    for (int i = threadIdx.x; i < 49 * C4; i += 256) {
        const int q = i % C4, bin = i / C4, ph = bin / 7, pw = bin % 7;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const float y = rsh + ph * bh + (iy + 0.5f) * bh / 2.f, x = rsw + pw * bw + (ix + 0.5f) * bw / 2.f;
                acc += roi_bilinear(f, H, W, cs, n, y, x, q);
            }
        acc = acc * 0.25f;
        float* o = out + (int64_t)r * C * 49 + bin;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[(int64_t)(q * 4 + e) * 49] = acc[e];
    }
}

extern "C" int adh_roi_align_fpn(void* stream, const adh_fpn_levels* levels, const float* rois, int R, int C, float* out) {
    if (!levels || !rois || !out || R < 1 || C < 4 || (C & 3) || levels->nlevels < 1 || levels->nlevels > 4) return ADH_E_ARG;
    for (int l = 0; l < levels->nlevels; ++l)
        if (!levels->f[l] || levels->H[l] < 1 || levels->W[l] < 1 || levels->cs[l] < C || (levels->cs[l] & 3)) return ADH_E_ARG;
    hipLaunchKernelGGL(roi_align_fpn_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, *levels, rois, C, out);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Box head post-processing (roi_heads.py postprocess_detections, up to the NMS): per RoI softmax over the class logits, per foreground
// class BoxCoder(10,10,5,5).decode of the class's deltas, clip to the image, valid = score > thresh and w, h >= min_size.
// logits [R, >=NC], deltas [R, >=4 NC] (class-major), props [R,4], img_hw [N,2] (h, w) per image, img [R] image index ->
// boxes [R,NC-1,4], scores [R,NC-1], valid [R,NC-1] int32.  One wave per RoI.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void box_postprocess_kernel(const float* __restrict__ logits, int l_cs, const float* __restrict__ deltas, int d_cs,
                                                             const float* __restrict__ props, const float* __restrict__ img_hw,
                                                             const int* __restrict__ img, int NC, float thresh, float min_size,
                                                             float* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ valid) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const float* lg = logits + (int64_t)r * l_cs;
    float m = -INFINITY;
    for (int c = lane; c < NC; c += 64) m = fmaxf(m, lg[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int c = lane; c < NC; c += 64) sum += expf(lg[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float* p = props + (int64_t)r * 4;
    const float w = p[2] - p[0], h = p[3] - p[1], cx = p[0] + 0.5f * w, cy = p[1] + 0.5f * h;
    const float ih = img_hw[img[r] * 2 + 0], iw = img_hw[img[r] * 2 + 1];
    for (int c = 1 + lane; c < NC; c += 64) {
        const float* d = deltas + (int64_t)r * d_cs + c * 4;
        const float dx = d[0] / 10.f, dy = d[1] / 10.f, dw = fminf(d[2] / 5.f, ADH_BBOX_XFORM_CLIP), dh = fminf(d[3] / 5.f, ADH_BBOX_XFORM_CLIP);
        const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        const float bx1 = fminf(fmaxf(pcx - 0.5f * pw, 0.f), iw), by1 = fminf(fmaxf(pcy - 0.5f * ph, 0.f), ih);
        const float bx2 = fminf(fmaxf(pcx + 0.5f * pw, 0.f), iw), by2 = fminf(fmaxf(pcy + 0.5f * ph, 0.f), ih);
        const float sc = expf(lg[c] - m) / sum;
        const int64_t o = (int64_t)r * (NC - 1) + (c - 1);
        boxes[o * 4 + 0] = bx1; boxes[o * 4 + 1] = by1; boxes[o * 4 + 2] = bx2; boxes[o * 4 + 3] = by2;
        scores[o] = sc;
        valid[o] = (sc > thresh && (bx2 - bx1) >= min_size && (by2 - by1) >= min_size) ? 1 : 0;
    }
}

extern "C" int adh_box_postprocess(void* stream, const float* logits, int l_cs, const float* deltas, int d_cs, const float* props,
                                   const float* img_hw, const int32_t* img, int R, int NC, float score_thresh, float min_size, float* boxes,
                                   float* scores, int32_t* valid) {
    if (!logits || !deltas || !props || !img_hw || !img || !boxes || !scores || !valid || R < 1 || NC < 2 || l_cs < NC || d_cs < 4 * NC)
        return ADH_E_ARG;
    hipLaunchKernelGGL(box_postprocess_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, logits, l_cs, deltas, d_cs, props, img_hw, img, NC,
                       score_thresh, min_size, boxes, scores, valid);
    return adh_check_launch();
}
