// Validation-path operators of the mask / keypoint branches: what torchvision 0.4.2's
// roi_heads.py / transform.py do around the branch convolutions when src/models/org/rcnn.py:124-127 runs a
// Mask / Keypoint R-CNN in eval mode (the reference evaluates them with iou_types bbox+segm / bbox+keypoints,
// src/utils/coco_eval_util.py:225-233).  The convolutions (mask_fcn1-4, conv5_mask, mask_fcn_logits, the eight
// keypoint convs, kps_score_lowres) run on hnd_conv2d_igemm (transposed convs as its data-gradient form); here:
//   hnd_mask_probs             maskrcnn_inference: sigmoid of the predicted class's logit plane
//   hnd_paste_masks            paste_masks_in_image: pad 1, bilinear resize (align_corners=False) of each MxM mask to
//                              its (truncated, expanded) box, pasted into a zero image
//   hnd_upsample_bilinear_nhwc KeypointRCNNPredictor's interpolate(scale_factor=2, bilinear, align_corners=False)
//   hnd_heatmaps_to_keypoints  heatmaps_to_keypoints: per RoI bicubic (A=-0.75, align_corners=False) resize of each
//                              heatmap to (ceil(h), ceil(w)), first-index argmax, keypoint coordinates and score
// Index results (argmax) and thresholded bits depend on the last bit of these expressions, so the file is compiled
// without FMA contraction and every expression follows the order of the CPU operator it restates
// (ATen/native/UpSample.h: area_pixel_compute_source_index, cubic_convolution1/2, upsample_get_cubic_coefficients).
#include "common.h"

#pragma clang fp contract(off)

namespace {

inline int grid_for(long long work, int threads = 256) {
  long long b = (work + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 65535LL * 32 ? 65535LL * 32 : b));
}

// GT-mask resize of CustomRCNNTransform.resize (reference src/models/org/rcnn.py:54-57):
//   misc_nn_ops.interpolate(mask[None].float(), scale_factor=s)[0].byte()   -- mode 'nearest'
// ATen: src = min((int64)floorf(dst * scale), in - 1) with scale = (float)(1.0 / scale_factor) per axis.  The float
// round trip of a uint8 is the identity, so the kernel moves bytes.
__global__ void resize_mask_nearest_kernel(const unsigned char* __restrict__ in, long long k, int h, int w, int oh,
                                           int ow, float rh, float rw, unsigned char* __restrict__ out) {
  const long long total = k * oh * ow;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % ow);
    const long long t = e / ow;
    const int oy = (int)(t % oh);
    const long long b = t / oh;
    int sy = (int)floorf((float)oy * rh), sx = (int)floorf((float)ox * rw);
    sy = sy < h - 1 ? sy : h - 1;
    sx = sx < w - 1 ? sx : w - 1;
    out[e] = in[(b * h + sy) * w + sx];
  }
}

// Run boundaries of the thresholded masks in COCO's column-major order (reference src/utils/coco_eval_util.py:101:
// `masks > 0.5`, then pycocotools' mask.encode): position p = x*h + y of mask k is a boundary when its bit differs from
// the bit at p-1 -- the pixel above, or for y = 0 the last pixel of the previous column.  Row-major, coalesced reads;
// the few thousand boundaries of an image are appended through one atomic counter in any order (the host sorts the
// keys k*h*w + p): only they cross PCIe instead of a 107 MB bit stack.
__global__ void mask_boundaries_kernel(const float* __restrict__ probs, long long n, int h, int w, float thr,
                                       long long* __restrict__ out, long long capacity,
                                       unsigned long long* __restrict__ count, unsigned char* __restrict__ first) {
  const long long hw = (long long)h * w, total = n * hw;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long k = e / hw, r = e - k * hw;
    const int y = (int)(r / w), x = (int)(r - (long long)y * w);
    const float* m = probs + k * hw;
    const bool bit = m[r] > thr;
    if (y == 0 && x == 0) {
      first[k] = bit ? 1 : 0;
      continue;
    }
    const bool prev = y > 0 ? (m[r - w] > thr) : (m[(long long)(h - 1) * w + (x - 1)] > thr);
    if (bit != prev) {
      const unsigned long long slot = atomicAdd(count, 1ull);
      if ((long long)slot < capacity) out[slot] = k * hw + (long long)x * h + y;
    }
  }
}

__global__ void mask_probs_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, long long k,
                                  int m, int ldc, float* __restrict__ probs) {
  const long long total = k * m * m;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / (m * m);
    const float x = logits[e * ldc + labels[r]];
    probs[e] = 1.0f / (1.0f + expf(-x));
  }
}

// area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false): scale*(dst+0.5)-0.5 clamped at 0
__device__ __forceinline__ void linear_src(float scale, int dst, int in_size, int& i0, int& i1, float& w0, float& w1) {
  float real = scale * ((float)dst + 0.5f) - 0.5f;
  real = real < 0.f ? 0.f : real;
  i0 = (int)real;
  i0 = i0 > in_size - 1 ? in_size - 1 : i0;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  w1 = real - (float)i0;
  w1 = w1 < 0.f ? 0.f : (w1 > 1.f ? 1.f : w1);
  w0 = 1.0f - w1;
}

__global__ void paste_masks_kernel(const float* __restrict__ probs, const long long* __restrict__ boxes, long long k,
                                   int m, int im_h, int im_w, float* __restrict__ out) {
  const long long plane = (long long)im_h * im_w, total = k * plane;
  const int mp = m + 2;                                          // the mask zero-padded by one pixel
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / plane;
    const int rem = (int)(e - r * plane), Y = rem / im_w, X = rem - Y * im_w;
    const long long x0 = boxes[r * 4 + 0], y0 = boxes[r * 4 + 1], x1 = boxes[r * 4 + 2], y1 = boxes[r * 4 + 3];
    long long w = x1 - x0 + 1, h = y1 - y0 + 1;
    w = w < 1 ? 1 : w;
    h = h < 1 ? 1 : h;
    const long long xa = x0 > 0 ? x0 : 0, xb = (x1 + 1) < im_w ? (x1 + 1) : im_w;
    const long long ya = y0 > 0 ? y0 : 0, yb = (y1 + 1) < im_h ? (y1 + 1) : im_h;
    float v = 0.f;
    // the reference assigns mask[(y_0-box[1]):(y_1-box[1]), ...]: python slicing also truncates at the resized mask
    const long long my = Y - y0, mx = X - x0;
    if (X >= xa && X < xb && Y >= ya && Y < yb && my < h && mx < w) {
      int iy0, iy1, ix0, ix1;
      float wy0, wy1, wx0, wx1;
      linear_src((float)mp / (float)h, (int)my, mp, iy0, iy1, wy0, wy1);
      linear_src((float)mp / (float)w, (int)mx, mp, ix0, ix1, wx0, wx1);
      const float* p = probs + r * m * m;
      auto at = [&](int yy, int xx) -> float {
        return (yy >= 1 && yy <= m && xx >= 1 && xx <= m) ? p[(yy - 1) * m + (xx - 1)] : 0.f;
      };
      const float top = wx0 * at(iy0, ix0) + wx1 * at(iy0, ix1);
      const float bot = wx0 * at(iy1, ix0) + wx1 * at(iy1, ix1);
      v = wy0 * top + wy1 * bot;
    }
    out[e] = v;
  }
}

__global__ void upsample_bilinear_kernel(const float* __restrict__ in, long long k, int h, int w, int c, int oh, int ow,
                                         float rh, float rw, float* __restrict__ out) {
  const long long total = k * oh * ow * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % c);
    long long t = e / c;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const long long r = t / oh;
    int iy0, iy1, ix0, ix1;
    float wy0, wy1, wx0, wx1;
    linear_src(rh, oy, h, iy0, iy1, wy0, wy1);
    linear_src(rw, ox, w, ix0, ix1, wx0, wx1);
    const float* p = in + r * h * w * c + ch;
    const float top = wx0 * p[((long long)iy0 * w + ix0) * c] + wx1 * p[((long long)iy0 * w + ix1) * c];
    const float bot = wx0 * p[((long long)iy1 * w + ix0) * c] + wx1 * p[((long long)iy1 * w + ix1) * c];
    out[e] = wy0 * top + wy1 * bot;
  }
}

// ATen/native/UpSample.h, A = -0.75
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  const float x1 = t;
  c[0] = cubic2(x1 + 1.0f, A);
  c[1] = cubic1(x1, A);
  const float x2 = 1.0f - t;
  c[2] = cubic1(x2, A);
  c[3] = cubic2(x2 + 1.0f, A);
}

constexpr int KP_THREADS = 256;

// one block per (roi, keypoint)
__global__ void __launch_bounds__(KP_THREADS) heatmaps_to_keypoints_kernel(
    const float* __restrict__ maps, int hm, int wm, int ldc, int nkp, const float* __restrict__ rois,
    float* __restrict__ xy, float* __restrict__ scores) {
  extern __shared__ float smap[];                                // [hm][wm] of this (roi, keypoint)
  __shared__ float red_v[KP_THREADS];
  __shared__ long long red_i[KP_THREADS];
  const int r = blockIdx.x / nkp, kp = blockIdx.x - r * nkp;
  const float* src = maps + (long long)r * hm * wm * ldc + kp;
  for (int i = threadIdx.x; i < hm * wm; i += KP_THREADS) smap[i] = src[(long long)i * ldc];
  __syncthreads();
  const float bx1 = rois[r * 4 + 0], by1 = rois[r * 4 + 1], bx2 = rois[r * 4 + 2], by2 = rois[r * 4 + 3];
  float widths = bx2 - bx1, heights = by2 - by1;
  widths = widths < 1.f ? 1.f : widths;
  heights = heights < 1.f ? 1.f : heights;
  const int ow = (int)ceilf(widths), oh = (int)ceilf(heights);
  const float scale_h = (float)hm / (float)oh, scale_w = (float)wm / (float)ow;
  float best = -INFINITY;
  long long best_i = -1;
  const long long total = (long long)oh * ow;
  for (long long p = threadIdx.x; p < total; p += KP_THREADS) {
    const int oy = (int)(p / ow), ox = (int)(p - (long long)oy * ow);
    const float real_y = scale_h * ((float)oy + 0.5f) - 0.5f, real_x = scale_w * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(real_y), fx = floorf(real_x);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coeffs(real_y - fy, cy);
    cubic_coeffs(real_x - fx, cx);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int yy = iy - 1 + j;
      yy = yy < 0 ? 0 : (yy > hm - 1 ? hm - 1 : yy);
      float row = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int xx = ix - 1 + i;
        xx = xx < 0 ? 0 : (xx > wm - 1 ? wm - 1 : xx);
        row = row + smap[yy * wm + xx] * cx[i];
      }
      v = v + row * cy[j];
    }
    if (v > best || best_i < 0) {                                // strict >: the first index wins inside a thread
      best = v;
      best_i = p;
    }
  }
  red_v[threadIdx.x] = best;
  red_i[threadIdx.x] = best_i;
  __syncthreads();
  for (int s = KP_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v2 = red_v[threadIdx.x + s];
      const long long i2 = red_i[threadIdx.x + s];
      const float v1 = red_v[threadIdx.x];
      const long long i1 = red_i[threadIdx.x];
      if (i2 >= 0 && (i1 < 0 || v2 > v1 || (v2 == v1 && i2 < i1))) {   // argmax: first index among equal maxima
        red_v[threadIdx.x] = v2;
        red_i[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long pos = red_i[0];
    const long long x_int = pos % ow, y_int = (pos - x_int) / ow;
    const float width_correction = widths / (float)ow, height_correction = heights / (float)oh;
    const float x = ((float)x_int + 0.5f) * width_correction, y = ((float)y_int + 0.5f) * height_correction;
    float* o = xy + ((long long)r * nkp + kp) * 3;
    o[0] = x + bx1;
    o[1] = y + by1;
    o[2] = 1.0f;
    scores[(long long)r * nkp + kp] = red_v[0];
  }
}

}  // namespace

extern "C" {

int hnd_mask_probs(const float* logits, const int64_t* labels, int64_t k, int m, int ldc, float* probs, void* stream) {
  if (k <= 0) return HND_OK;
  HND_REQUIRE(logits && labels && probs && m > 0 && ldc > 0, "hnd_mask_probs: bad arguments");
  hipLaunchKernelGGL(mask_probs_kernel, dim3(grid_for(k * m * m)), dim3(256), 0, hnd::as_stream(stream), logits,
                     (const long long*)labels, (long long)k, m, ldc, probs);
  return hnd::check_launch("hnd_mask_probs");
}

int hnd_paste_masks(const float* probs, const int64_t* boxes, int64_t k, int m, int im_h, int im_w, float* out,
                    void* stream) {
  if (k <= 0) return HND_OK;
  HND_REQUIRE(probs && boxes && out && m > 0 && im_h > 0 && im_w > 0, "hnd_paste_masks: bad arguments");
  hipLaunchKernelGGL(paste_masks_kernel, dim3(grid_for(k * (long long)im_h * im_w)), dim3(256), 0, hnd::as_stream(stream),
                     probs, (const long long*)boxes, (long long)k, m, im_h, im_w, out);
  return hnd::check_launch("hnd_paste_masks");
}

int hnd_mask_run_boundaries(const float* probs, int64_t n, int h, int w, float threshold, int64_t* out, int64_t capacity,
                            int64_t* count, uint8_t* first, void* stream) {
  HND_REQUIRE(count != nullptr, "hnd_mask_run_boundaries: null counter");
  if (hipMemsetAsync(count, 0, sizeof(int64_t), hnd::as_stream(stream)) != hipSuccess)
    return hnd::check_launch("hnd_mask_run_boundaries(memset)");
  if (n <= 0) return HND_OK;
  HND_REQUIRE(probs && out && first && h > 0 && w > 0 && capacity >= 0, "hnd_mask_run_boundaries: bad arguments");
  hipLaunchKernelGGL(mask_boundaries_kernel, dim3(grid_for(n * (long long)h * w)), dim3(256), 0, hnd::as_stream(stream),
                     probs, (long long)n, h, w, threshold, (long long*)out, (long long)capacity,
                     (unsigned long long*)count, first);
  return hnd::check_launch("hnd_mask_run_boundaries");
}

int hnd_resize_mask_nearest_u8(const unsigned char* in, int64_t k, int h, int w, int oh, int ow, double scale_factor,
                               unsigned char* out, void* stream) {
  if (k <= 0 || oh <= 0 || ow <= 0) return HND_OK;
  HND_REQUIRE(in && out && h > 0 && w > 0 && scale_factor > 0.0, "hnd_resize_mask_nearest_u8: bad arguments");
  const float r = (float)(1.0 / scale_factor);                   // compute_scales_value with a given scale_factor
  hipLaunchKernelGGL(resize_mask_nearest_kernel, dim3(grid_for(k * (long long)oh * ow)), dim3(256), 0,
                     hnd::as_stream(stream), in, (long long)k, h, w, oh, ow, r, r, out);
  return hnd::check_launch("hnd_resize_mask_nearest_u8");
}

int hnd_upsample_bilinear_nhwc(const float* in, int64_t k, int h, int w, int c, int factor, float* out, void* stream) {
  if (k <= 0) return HND_OK;
  HND_REQUIRE(in && out && h > 0 && w > 0 && c > 0 && factor >= 1, "hnd_upsample_bilinear_nhwc: bad arguments");
  const float r = 1.0f / (float)factor;                          // area_pixel_compute_scale with a given scale_factor
  hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(grid_for(k * (long long)h * factor * w * factor * c)), dim3(256), 0,
                     hnd::as_stream(stream), in, (long long)k, h, w, c, h * factor, w * factor, r, r, out);
  return hnd::check_launch("hnd_upsample_bilinear_nhwc");
}

int hnd_heatmaps_to_keypoints(const float* maps, int64_t k, int h, int w, int ldc, int num_keypoints, const float* rois,
                              float* xy, float* scores, void* stream) {
  if (k <= 0) return HND_OK;
  HND_REQUIRE(maps && rois && xy && scores && h > 0 && w > 0 && num_keypoints > 0 && ldc >= num_keypoints,
              "hnd_heatmaps_to_keypoints: bad arguments");
  HND_REQUIRE((size_t)h * w * sizeof(float) <= 64 * 1024, "hnd_heatmaps_to_keypoints: heatmap %dx%d exceeds the LDS tile",
              h, w);
  HND_REQUIRE(k * num_keypoints <= 0x7fffffffLL, "hnd_heatmaps_to_keypoints: too many (roi, keypoint) pairs");
  hipLaunchKernelGGL(heatmaps_to_keypoints_kernel, dim3((unsigned)(k * num_keypoints)), dim3(KP_THREADS),
                     (size_t)h * w * sizeof(float), hnd::as_stream(stream), maps, h, w, ldc, num_keypoints, rois, xy,
                     scores);
  return hnd::check_launch("hnd_heatmaps_to_keypoints");
}

}  // extern "C"
