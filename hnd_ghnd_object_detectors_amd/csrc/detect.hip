// Validation-path operators: the box branch of torchvision 0.4.2's eval-mode detector that
// src/utils/main_util.py:75-113 (evaluate) drives through src/models/org/rcnn.py:124-127 to pick the checkpoint
// (src/mimic_runner.py:94-100).  Convolutions and the box-head Linear layers run on hnd_conv2d_igemm; this file holds
// what is not a GEMM:
//   hnd_rpn_decode      AnchorGenerator.grid_anchors + BoxCoder.decode_single(weights 1,1,1,1) + the (h, w, a) flattening
//                       of concat_box_prediction_layers, straight from the RPN head's NHWC output
//   hnd_clip_boxes      ops.boxes.clip_boxes_to_image
//   hnd_nms             ops.nms (csrc/cpu/nms_cpu.cpp): IoU bit-matrix + one sequential scan; kept set bit-exact
//   hnd_roi_align       ops.roi_align (csrc/cpu/ROIAlign_cpu.cpp, non-"aligned"), NHWC in / NHWC out
//   hnd_box_decode_clip BoxCoder.decode(weights 10,10,5,5) + clip of RoIHeads.postprocess_detections, all classes
// Index / byte results (kept sets) must equal the CPU operator's exactly on identical inputs, so every IoU / bilinear
// expression below is written as single correctly-rounded IEEE operations in the reference's order
// (no FMA contraction, no reassociation).  NOTE: hipcc's __fmul_rn / __fadd_rn are plain `*` / `+` and WOULD be
// fused into FMAs under the library-wide -ffp-contract=fast; contraction is therefore switched off for this whole
// translation unit (pragma below + -ffp-contract=off for this file in the Makefile); the _rn spellings only document
// the intended operation order.
#include "common.h"

#pragma clang fp contract(off)

using hnd::f32x4;

namespace {

inline int grid_for(long long work, int threads = 256) {
  long long b = (work + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

struct Box {
  float x1, y1, x2, y2;
};

// BoxCoder.decode_single for one box, one 4-tuple of deltas (torchvision/models/detection/_utils.py)
__device__ __forceinline__ Box decode_one(Box b, float d0, float d1, float d2, float d3, float wx, float wy, float ww,
                                          float wh, float clip) {
  const float widths = __fsub_rn(b.x2, b.x1), heights = __fsub_rn(b.y2, b.y1);
  const float ctr_x = __fadd_rn(b.x1, __fmul_rn(0.5f, widths)), ctr_y = __fadd_rn(b.y1, __fmul_rn(0.5f, heights));
  const float dx = __fdiv_rn(d0, wx), dy = __fdiv_rn(d1, wy);
  float dw = __fdiv_rn(d2, ww), dh = __fdiv_rn(d3, wh);
  dw = fminf(dw, clip);
  dh = fminf(dh, clip);
  const float pcx = __fadd_rn(__fmul_rn(dx, widths), ctr_x), pcy = __fadd_rn(__fmul_rn(dy, heights), ctr_y);
  const float pw = __fmul_rn(expf(dw), widths), ph = __fmul_rn(expf(dh), heights);
  Box o;
  o.x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw));
  o.y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
  o.x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw));
  o.y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
  return o;
}

struct RpnDecodeArgs {
  const float* head;     // [n][h][w][ldc]: channel a = objectness logit of anchor a, A + a*4 + c = delta c of anchor a
  float* objectness;     // [n][total]
  float* proposals;      // [n][total][4]
  int n, h, w, ldc, A;
  long long total, offset;          // anchors per image over all levels; first anchor of this level
  float stride_h, stride_w, clip;
  float base[16][4];                // cell anchors of this level (rounded, rpn.py generate_anchors), A <= 16
};

__global__ void rpn_decode_kernel(const RpnDecodeArgs a) {
  const long long per = (long long)a.h * a.w * a.A, all = per * a.n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < all; e += (long long)gridDim.x * blockDim.x) {
    const int an = (int)(e % a.A);
    long long p = e / a.A;
    const int x = (int)(p % a.w);
    p /= a.w;
    const int y = (int)(p % a.h), img = (int)(p / a.h);
    const float* src = a.head + (((size_t)img * a.h + y) * a.w + x) * a.ldc;
    const float sx = __fmul_rn((float)x, a.stride_w), sy = __fmul_rn((float)y, a.stride_h);     // arange * stride
    Box anc;
    anc.x1 = __fadd_rn(sx, a.base[an][0]);
    anc.y1 = __fadd_rn(sy, a.base[an][1]);
    anc.x2 = __fadd_rn(sx, a.base[an][2]);
    anc.y2 = __fadd_rn(sy, a.base[an][3]);
    const float* d = src + a.A + an * 4;
    const Box o = decode_one(anc, d[0], d[1], d[2], d[3], 1.f, 1.f, 1.f, 1.f, a.clip);
    const long long dst = (long long)img * a.total + a.offset + ((long long)y * a.w + x) * a.A + an;
    a.objectness[dst] = src[an];
    *(f32x4*)(a.proposals + dst * 4) = f32x4{o.x1, o.y1, o.x2, o.y2};
  }
}

__global__ void clip_boxes_kernel(float* boxes, long long n, float height, float width) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    f32x4 b = *(f32x4*)(boxes + e * 4);
    b.x = fminf(fmaxf(b.x, 0.f), width);
    b.z = fminf(fmaxf(b.z, 0.f), width);
    b.y = fminf(fmaxf(b.y, 0.f), height);
    b.w = fminf(fmaxf(b.w, 0.f), height);
    *(f32x4*)(boxes + e * 4) = b;
  }
}

// ---------------------------------------------------------------------------------------------- NMS
// bit j of mask[i][j / 64] (i, j = positions in descending-score order, j > i) = IoU(box order[i], box order[j]) > thr
__global__ void nms_mask_kernel(const float* __restrict__ boxes, const long long* __restrict__ order, int n, float thr,
                                unsigned long long* __restrict__ mask, int words) {
  __shared__ float cb[64][5];
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;                              // only j > i matters
  const int t = threadIdx.x;                         // 64 threads
  const int cj = cbk * 64 + t;
  if (cj < n) {
    const f32x4 b = *(const f32x4*)(boxes + (size_t)order[cj] * 4);
    cb[t][0] = b.x; cb[t][1] = b.y; cb[t][2] = b.z; cb[t][3] = b.w;
    cb[t][4] = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const f32x4 bi = *(const f32x4*)(boxes + (size_t)order[i] * 4);
  const float iarea = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
  unsigned long long bits = 0;
  const int jn = min(64, n - cbk * 64);
  for (int j = 0; j < jn; ++j) {
    if (cbk * 64 + j <= i) continue;
    const float xx1 = fmaxf(bi.x, cb[j][0]), yy1 = fmaxf(bi.y, cb[j][1]);
    const float xx2 = fminf(bi.z, cb[j][2]), yy2 = fminf(bi.w, cb[j][3]);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, cb[j][4]), inter));
    if (ovr > thr) bits |= 1ull << j;
  }
  mask[(size_t)i * words + cbk] = bits;
}

// greedy scan in score order: keep[order[i]] = 1 unless an earlier kept box suppressed position i.  One workgroup
// walks the positions 64 at a time.  For block b: (1) GATHER -- every thread ORs word b of the rows of earlier
// survivors (their bitmap sits in LDS; loads are independent across the 256 threads), one workgroup-wide OR gives the
// positions of block b already dead; (2) CHAIN -- wave 0 holds the block's 64 diagonal words one per lane and
// resolves the in-block dependency with 64 register steps (v_readlane, no memory access on the serial path).
// Same greedy result as one position at a time (1.0 ms -> 0.28 ms at ~4800 boxes: profiles/r02_eval_mask_kernel_stats.md).
constexpr int NMS_SCAN_THREADS = 1024;
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {   // lane is wave-uniform
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}
__global__ void __launch_bounds__(NMS_SCAN_THREADS) nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                                    const long long* __restrict__ order, int n,
                                                                    int words, unsigned char* __restrict__ keep) {
  extern __shared__ unsigned long long keptw[];                 // [words] survivors of the blocks done so far
  __shared__ unsigned long long part[NMS_SCAN_THREADS / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int k = t; k < n; k += NMS_SCAN_THREADS) keep[k] = 0;
  for (int b = 0; b < words; ++b) {
    const int base = b * 64, cnt = min(64, n - base);
    // the block's diagonal words: issued first so the load is in flight during the gather (wave 0 only)
    const unsigned long long d = (wave == 0 && lane < cnt) ? mask[(size_t)(base + lane) * words + b] : 0ull;  // bits j > lane
    unsigned long long acc = 0;
    for (int r = t; r < base; r += NMS_SCAN_THREADS)            // rows of earlier blocks; r >> 6 < b
      if ((keptw[r >> 6] >> (r & 63)) & 1ull) acc |= mask[(size_t)r * words + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (wave == 0) {
      unsigned long long cur = lane < NMS_SCAN_THREADS / 64 ? part[lane] : 0ull;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) cur |= __shfl_xor(cur, o);
      cur = readlane64(cur, 0);
      for (int j = 0; j < cnt; ++j) {
        const unsigned long long dj = readlane64(d, j);
        if (!((cur >> j) & 1ull)) cur |= dj;
      }
      const unsigned long long kept = ~cur & (cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull));
      if (lane == 0) keptw[b] = kept;
      if (lane < cnt && ((kept >> lane) & 1ull)) keep[order[base + lane]] = 1;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- RoIAlign
struct RoiAlignArgs {
  const float* feat;            // [n][h][w][c]
  const float* rois;            // [K][5]: batch index, x1, y1, x2, y2 (image coordinates)
  const long long* idx;         // [k]: rows of `rois` / `out` handled by this launch (one pyramid level)
  float* out;                   // [K][ph][pw][c]
  int k, h, w, c, ph, pw, sampling;
  float scale;
};

__device__ __forceinline__ void bilinear_setup(float y, int size, bool& bad, int& lo, int& hi, float& l, float& hw) {
  bad = y < -1.0f || y > (float)size;
  if (y <= 0.f) y = 0.f;
  lo = (int)y;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    y = (float)lo;
  } else {
    hi = lo + 1;
  }
  l = __fsub_rn(y, (float)lo);
  hw = __fsub_rn(1.f, l);
}

__global__ void roi_align_kernel(const RoiAlignArgs a) {
  const int c4n = a.c >> 2;
  const long long total = (long long)a.k * a.ph * a.pw * c4n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    long long p = e / c4n;
    const int pw_ = (int)(p % a.pw);
    p /= a.pw;
    const int ph_ = (int)(p % a.ph);
    const long long r = a.idx[p / a.ph];
    const float* roi = a.rois + r * 5;
    const int b = (int)roi[0];
    const float sw = __fmul_rn(roi[1], a.scale), sh = __fmul_rn(roi[2], a.scale);
    const float ew = __fmul_rn(roi[3], a.scale), eh = __fmul_rn(roi[4], a.scale);
    const float rw = fmaxf(__fsub_rn(ew, sw), 1.f), rh = fmaxf(__fsub_rn(eh, sh), 1.f);
    const float bin_h = __fdiv_rn(rh, (float)a.ph), bin_w = __fdiv_rn(rw, (float)a.pw);
    const int gh = a.sampling > 0 ? a.sampling : (int)ceilf(__fdiv_rn(rh, (float)a.ph));
    const int gw = a.sampling > 0 ? a.sampling : (int)ceilf(__fdiv_rn(rw, (float)a.pw));
    const float count = (float)(gh * gw);
    const float* fb = a.feat + (size_t)b * a.h * a.w * a.c + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
      // roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / roi_bin_grid_h
      const float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph_, bin_h)),
                                __fdiv_rn(__fmul_rn((float)iy + .5f, bin_h), (float)gh));
      bool ybad;
      int ylo, yhi;
      float ly, hy;
      bilinear_setup(y, a.h, ybad, ylo, yhi, ly, hy);
      for (int ix = 0; ix < gw; ++ix) {
        const float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw_, bin_w)),
                                  __fdiv_rn(__fmul_rn((float)ix + .5f, bin_w), (float)gw));
        bool xbad;
        int xlo, xhi;
        float lx, hx;
        bilinear_setup(x, a.w, xbad, xlo, xhi, lx, hx);
        if (ybad || xbad) continue;                    // the operator adds an all-zero-weight sample: + 0
        const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
        const f32x4 v1 = *(const f32x4*)(fb + ((size_t)ylo * a.w + xlo) * a.c);
        const f32x4 v2 = *(const f32x4*)(fb + ((size_t)ylo * a.w + xhi) * a.c);
        const f32x4 v3 = *(const f32x4*)(fb + ((size_t)yhi * a.w + xlo) * a.c);
        const f32x4 v4 = *(const f32x4*)(fb + ((size_t)yhi * a.w + xhi) * a.c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float val = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1[q]), __fmul_rn(w2, v2[q])),
                                                __fmul_rn(w3, v3[q])), __fmul_rn(w4, v4[q]));
          acc[q] = __fadd_rn(acc[q], val);
        }
      }
    }
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = __fdiv_rn(acc[q], count);
    *(f32x4*)(a.out + (((size_t)r * a.ph + ph_) * a.pw + pw_) * a.c + c4 * 4) = o;
  }
}

// ---------------------------------------------------------------------------------------------- class-wise decode
__global__ void box_decode_clip_kernel(const float* __restrict__ deltas, int ld, const float* __restrict__ rois,
                                       const float* __restrict__ img_hw, int nroi, int ncls, float wx, float wy,
                                       float ww, float wh, float clip, float* __restrict__ out) {
  const long long total = (long long)nroi * ncls;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int cls = (int)(e % ncls);
    const long long r = e / ncls;
    const float* roi = rois + r * 5;
    const int img = (int)roi[0];
    const float* d = deltas + r * ld + cls * 4;
    Box o = decode_one(Box{roi[1], roi[2], roi[3], roi[4]}, d[0], d[1], d[2], d[3], wx, wy, ww, wh, clip);
    const float height = img_hw[img * 2], width = img_hw[img * 2 + 1];
    o.x1 = fminf(fmaxf(o.x1, 0.f), width);
    o.x2 = fminf(fmaxf(o.x2, 0.f), width);
    o.y1 = fminf(fmaxf(o.y1, 0.f), height);
    o.y2 = fminf(fmaxf(o.y2, 0.f), height);
    *(f32x4*)(out + e * 4) = f32x4{o.x1, o.y1, o.x2, o.y2};
  }
}

}  // namespace

extern "C" {

int hnd_rpn_decode(const float* head, int n, int h, int w, int ldc, int num_anchors, const float* base_anchors,
                   float stride_h, float stride_w, int64_t level_offset, int64_t anchors_per_image, float xform_clip,
                   float* objectness, float* proposals, void* stream) {
  HND_REQUIRE(head && base_anchors && objectness && proposals, "hnd_rpn_decode: null pointer");
  HND_REQUIRE(n > 0 && h > 0 && w > 0 && num_anchors > 0 && num_anchors <= 16 && ldc >= 5 * num_anchors,
              "hnd_rpn_decode: bad geometry (A=%d, ldc=%d)", num_anchors, ldc);
  HND_REQUIRE(level_offset >= 0 && level_offset + (int64_t)h * w * num_anchors <= anchors_per_image,
              "hnd_rpn_decode: level does not fit in anchors_per_image");
  RpnDecodeArgs a;
  a.head = head; a.objectness = objectness; a.proposals = proposals;
  a.n = n; a.h = h; a.w = w; a.ldc = ldc; a.A = num_anchors;
  a.total = anchors_per_image; a.offset = level_offset;
  a.stride_h = stride_h; a.stride_w = stride_w; a.clip = xform_clip;
  for (int i = 0; i < num_anchors; ++i)
    for (int j = 0; j < 4; ++j) a.base[i][j] = base_anchors[i * 4 + j];        // HOST pointer: 4*A floats
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(grid_for((long long)n * h * w * num_anchors)), dim3(256), 0,
                     hnd::as_stream(stream), a);
  return hnd::check_launch("hnd_rpn_decode");
}

int hnd_clip_boxes(float* boxes, int64_t n, float height, float width, void* stream) {
  HND_REQUIRE(boxes != nullptr || n == 0, "hnd_clip_boxes: null pointer");
  if (n <= 0) return HND_OK;
  hipLaunchKernelGGL(clip_boxes_kernel, dim3(grid_for(n)), dim3(256), 0, hnd::as_stream(stream), boxes, (long long)n,
                     height, width);
  return hnd::check_launch("hnd_clip_boxes");
}

size_t hnd_nms_workspace(int64_t n) {
  const int64_t words = (n + 63) / 64;
  return (size_t)(n * words) * sizeof(unsigned long long);
}

int hnd_nms(const float* boxes, const int64_t* order, int64_t n, float iou_threshold, void* workspace,
            uint8_t* keep, void* stream) {
  if (n <= 0) return HND_OK;
  HND_REQUIRE(boxes && order && workspace && keep, "hnd_nms: null pointer");
  // (RoIHeads.postprocess_detections can reach 1000 proposals x 90 classes = 90 000 candidates with a score threshold
  // near 0; the bit matrix is then 90 000 x 1407 words = 1 GB of caller-provided workspace)
  HND_REQUIRE(n <= 131072, "hnd_nms: at most 131072 boxes per call (got %lld)", (long long)n);
  const int words = (int)((n + 63) / 64);
  if (hipMemsetAsync(workspace, 0, hnd_nms_workspace(n), hnd::as_stream(stream)) != hipSuccess) return hnd::check_launch("hnd_nms(memset)");
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, hnd::as_stream(stream), boxes,
                     (const long long*)order, (int)n, iou_threshold, (unsigned long long*)workspace, words);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(NMS_SCAN_THREADS), words * sizeof(unsigned long long), hnd::as_stream(stream),
                     (const unsigned long long*)workspace, (const long long*)order, (int)n, words, keep);
  return hnd::check_launch("hnd_nms");
}

int hnd_roi_align(const float* feat, int n, int h, int w, int c, const float* rois, const int64_t* idx, int64_t k,
                  float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, float* out, void* stream) {
  if (k <= 0) return HND_OK;
  HND_REQUIRE(feat && rois && idx && out, "hnd_roi_align: null pointer");
  HND_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && pooled_h > 0 && pooled_w > 0,
              "hnd_roi_align: bad geometry (c=%d must be a multiple of 4)", c);
  RoiAlignArgs a{feat, rois, (const long long*)idx, out, (int)k, h, w, c, pooled_h, pooled_w, sampling_ratio,
                 spatial_scale};
  hipLaunchKernelGGL(roi_align_kernel, dim3(grid_for((long long)k * pooled_h * pooled_w * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), a);
  return hnd::check_launch("hnd_roi_align");
}

int hnd_box_decode_clip(const float* deltas, int ld, const float* rois, const float* image_hw, int64_t nroi, int ncls,
                        float wx, float wy, float ww, float wh, float xform_clip, float* out, void* stream) {
  if (nroi <= 0) return HND_OK;
  HND_REQUIRE(deltas && rois && image_hw && out && ncls > 0 && ld >= 4 * ncls, "hnd_box_decode_clip: bad arguments");
  hipLaunchKernelGGL(box_decode_clip_kernel, dim3(grid_for(nroi * ncls)), dim3(256), 0, hnd::as_stream(stream), deltas,
                     ld, rois, image_hw, (int)nroi, ncls, wx, wy, ww, wh, xform_clip, out);
  return hnd::check_launch("hnd_box_decode_clip");
}

}  // extern "C"
