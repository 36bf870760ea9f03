// HBM-bound kernels of the HND/GHND step (gfx950): transform, pooling, BatchNorm statistics /
// backward, the fused L2 mimic loss + gradient, fused flat Adam, weight packing.
// All are streaming passes: 16-byte accesses per lane, grid-stride over <= 2048*8 blocks,
// per-wave shuffle reductions then per-block partials (no float atomics -> deterministic).
#include "common.h"

#include <hip/hip_fp16.h>
#include <math.h>

namespace {

using hnd::f32x4;

constexpr int kMaxBlocks = 256 * 16;

inline int grid_for(long long work_items, int threads = 256) {
  long long b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// Streaming loops (round 4): a workgroup takes a CONTIGUOUS chunk of 256 * kEwU vectors per iteration and every thread has
// its kEwU 16-byte loads in flight before the first use; the channel of a lane is fixed when the channel groups per pixel
// divide the block (no 64-bit modulo per element).  tools/bench_elementwise.py: 4.7 -> ~6 TB/s, torch's own elementwise
// kernels reach 6.0-6.2 on the same tensors.
constexpr int kEwU = 4;
#define HND_NT_LOAD4(p) __builtin_nontemporal_load((const f32x4*)(p))   // single-use operands of streaming passes
inline int grid_for_chunks(long long n4) {
  long long b = (n4 + 256 * kEwU - 1) / (256 * kEwU);
  if (b < 1) b = 1;
  if (b > 256 * 64) b = 256 * 64;
  return (int)b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------ pack / fold
struct PackArgs {
  const float* src;
  float* dst;
  int cout, cin, kh, kw, transposed, chan_pad, i0, istep, ni, j0, jstep, nj, rows_pad, kdim;
};

__device__ __forceinline__ void pack_weights_body(const PackArgs& a, int block, int nblocks) {
  const long long total = (long long)a.rows_pad * a.kdim;
  for (long long e = block * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)nblocks * blockDim.x) {
    const int k = (int)(e % a.kdim);
    const int r = hnd::chan_of_row((int)(e / a.kdim));      // packed row e / kdim holds this output channel
    const int tap = k / a.chan_pad, c = k - tap * a.chan_pad;
    float v = 0.f;
    const int rows = a.transposed ? a.cin : a.cout, chans = a.transposed ? a.cout : a.cin;
    if (r < rows && c < chans && tap < a.ni * a.nj) {
      const int ta = tap / a.nj, tb = tap - ta * a.nj;
      const int i = a.i0 + ta * a.istep, j = a.j0 + tb * a.jstep;
      const int o = a.transposed ? c : r, ic = a.transposed ? r : c;
      v = a.src[(((size_t)o * a.cin + ic) * a.kh + i) * a.kw + j];
    }
    a.dst[e] = v;
  }
}

__global__ void pack_weights_kernel(const PackArgs a) { pack_weights_body(a, blockIdx.x, gridDim.x); }

// several operands in one launch (blockIdx.y = operand): the trainable head re-packs ~28 small operands every step
constexpr int kMaxBatchPacks = 32;
struct PackBatch {
  PackArgs op[kMaxBatchPacks];
};
__global__ void pack_weights_batch_kernel(const PackBatch b) { pack_weights_body(b.op[blockIdx.y], blockIdx.x, gridDim.x); }

__global__ void fbn_fold_kernel(const float* w, const float* b, const float* mean, const float* var, float* scale,
                                float* shift, int c, int cs, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cs) return;
  if (i >= c) { scale[i] = 0.f; shift[i] = 0.f; return; }
  const float s = w[i] * (1.0f / sqrtf(var[i] + eps));   // weight * running_var.rsqrt(); eps = 0 for torchvision 0.4.2
  scale[i] = s;
  shift[i] = b[i] - mean[i] * s;
}

// ------------------------------------------------------------------------------------ transform
struct TransformArgs {
  const float* src;
  const uint8_t* src8;     // U8 kernels: decoded image, value / 255 first (ToTensor)
  int hwc, flip;           // uint8 layout [h][w][3] vs [3][h][w]; horizontal flip of the source
  float* dst;
  int h, w, out_h, out_w, hp, wp;
  float rh, rw;
  float mean[3], inv_unused[3], std[3];
};

template <bool U8>
__device__ __forceinline__ void transform_body(const TransformArgs& a, int block, int nblocks) {
  const int total = a.hp * a.wp;
  const size_t plane = (size_t)a.h * a.w;
  for (int p = block * blockDim.x + threadIdx.x; p < total; p += nblocks * blockDim.x) {
    const int y = p / a.wp, x = p - y * a.wp;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (y < a.out_h && x < a.out_w) {
      // F.interpolate(mode='bilinear', align_corners=False) source index (area_pixel_compute_source_index)
      float sy = a.rh * ((float)y + 0.5f) - 0.5f, sx = a.rw * ((float)x + 0.5f) - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      int y0 = (int)sy, x0 = (int)sx;
      if (y0 > a.h - 1) y0 = a.h - 1;
      if (x0 > a.w - 1) x0 = a.w - 1;
      const int y1 = y0 + (y0 < a.h - 1 ? 1 : 0), x1 = x0 + (x0 < a.w - 1 ? 1 : 0);
      const float ly = sy - (float)y0, lx = sx - (float)x0;
      const float hy = 1.f - ly, hx = 1.f - lx;
      float v[3];
      // U8 only: image.flip(-1) happens before the resize, i.e. on source columns
      const int fx0 = (U8 && a.flip) ? a.w - 1 - x0 : x0, fx1 = (U8 && a.flip) ? a.w - 1 - x1 : x1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float q00, q01, q10, q11;
        if (!U8) {
          const float* s = a.src + c * plane;
          q00 = s[(size_t)y0 * a.w + fx0]; q01 = s[(size_t)y0 * a.w + fx1];
          q10 = s[(size_t)y1 * a.w + fx0]; q11 = s[(size_t)y1 * a.w + fx1];
        } else {
          const size_t cs = a.hwc ? 1 : plane, ps = a.hwc ? 3 : 1;      // channel / pixel strides
          const uint8_t* s = a.src8 + c * cs;
          q00 = (float)s[((size_t)y0 * a.w + fx0) * ps] / 255.f; q01 = (float)s[((size_t)y0 * a.w + fx1) * ps] / 255.f;
          q10 = (float)s[((size_t)y1 * a.w + fx0) * ps] / 255.f; q11 = (float)s[((size_t)y1 * a.w + fx1) * ps] / 255.f;
        }
        // normalise first ((x-mean)/std, rcnn.py:74), then interpolate (rcnn.py:75)
        const float p00 = (q00 - a.mean[c]) / a.std[c], p01 = (q01 - a.mean[c]) / a.std[c];
        const float p10 = (q10 - a.mean[c]) / a.std[c], p11 = (q11 - a.mean[c]) / a.std[c];
        v[c] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
      }
      o.x = v[0]; o.y = v[1]; o.z = v[2];
    }
    *(f32x4*)(a.dst + (size_t)p * 4) = o;
  }
}

template <bool U8>
__global__ void transform_kernel(const TransformArgs a) {
  transform_body<U8>(a, blockIdx.x, gridDim.x);
}

// The whole batch in one launch (blockIdx.y = image): 16 launches of 18 us each were launch-bound (1.7 TB/s).
// One instantiation per source type and the per-image arguments taken BY VALUE: with both types behind a run-time branch
// (or the arguments by reference into the array) hipcc contracts the interpolation into different fma chains than in
// the one-image kernel -- 1 ulp apart, which the few-step training goldens of the neural filter amplify past their 1e-3.
constexpr int kMaxBatchImages = 32;
struct TransformBatch {
  TransformArgs img[kMaxBatchImages];
};
template <bool U8>
__global__ void transform_batch_kernel(const TransformBatch b) {
  const TransformArgs a = b.img[blockIdx.y];
  transform_body<U8>(a, blockIdx.x, gridDim.x);
}

// Ground-truth boxes [k][4] (x1, y1, x2, y2) of every image of the batch rescaled in one launch (blockIdx.y = image):
// the reference's resize_boxes is four multiplies and a stack per image, i.e. 80 four-microsecond launches per step.
struct BoxScaleBatch {
  const float* src[kMaxBatchImages];
  float* dst[kMaxBatchImages];
  int k[kMaxBatchImages];
  float rw[kMaxBatchImages], rh[kMaxBatchImages];
};
__global__ void scale_boxes_batch_kernel(const BoxScaleBatch b) {
  const int i = blockIdx.y, n = b.k[i] * 4;
  const float rw = b.rw[i], rh = b.rh[i];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x)
    b.dst[i][e] = b.src[i][e] * ((e & 1) ? rh : rw);
}

// packed[r][tap * chan_pad + c] *= scale[c]  (c < nscale): folds a per-channel scale of the GEMM's K operand into the
// packed weights (hnd_scale_packed_k)
__global__ void scale_packed_k_kernel(float* __restrict__ w, const float* __restrict__ scale, long long total, int kdim,
                                      int ntaps, int chan_pad, int nscale) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e % kdim);
    const int tap = k / chan_pad, c = k - tap * chan_pad;
    if (tap < ntaps && c < nscale) w[e] *= scale[c];
  }
}

// ------------------------------------------------------------------------------------ max pool
// (32-bit indices with precomputed reciprocals: the 64-bit divisions by c/4, w, h of the first version cost more than the
// memory accesses -- 4.6-4.9 TB/s where a plain streaming kernel reaches 6)
struct PoolDivs {
  hnd::FastDiv c4n, w, h;     // h, w of the tensor the thread index walks (output for fwd, input for bwd)
};

// Forward: one thread walks a strip of kPoolStrip output rows of one column and four channels; the input row shared by two
// consecutive windows (2 oy + 1) stays in registers, so a row is loaded 1.125x instead of 1.5x and a window costs six
// loads instead of nine.  Scan order inside a window as in torch (rows top to bottom, left to right; the first element is
// always taken, then a strictly greater value or a NaN wins).
constexpr int kPoolStrip = 8;
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                   int n, int h, int w, int c, int oh, int ow, int nstrips, const PoolDivs dv) {
  const unsigned c4n = c >> 2;
  const unsigned total = (unsigned)n * nstrips * ow * c4n;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    unsigned p = hnd::fdiv(e, dv.c4n);
    const int c4 = (int)(e - p * c4n);
    unsigned q = hnd::fdiv(p, dv.w);
    const int ox = (int)(p - q * ow);
    const unsigned b = hnd::fdiv(q, dv.h);
    const int st = (int)(q - b * nstrips);
    const int oy0 = st * kPoolStrip;
    const int oy1 = oy0 + kPoolStrip < oh ? oy0 + kPoolStrip : oh;
    const bool cv0 = ox > 0, cv2 = 2 * ox + 1 < w;                     // column 2 ox is always inside
    const float* xc = x + ((size_t)b * h * w + (size_t)(2 * ox)) * c + c4 * 4;     // (row 0, column 2 ox)
    const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 top[3] = {ninf, ninf, ninf};
    bool topv = oy0 > 0;
    if (topv) {
      const float* r = xc + (size_t)(2 * oy0 - 1) * w * c;
      if (cv0) top[0] = *(const f32x4*)(r - c);
      top[1] = *(const f32x4*)r;
      if (cv2) top[2] = *(const f32x4*)(r + c);
    }
    for (int oy = oy0; oy < oy1; ++oy) {
      const float* rm = xc + (size_t)(2 * oy) * w * c;
      const bool botv = 2 * oy + 1 < h;
      f32x4 mid[3] = {ninf, ninf, ninf}, bot[3] = {ninf, ninf, ninf};
      if (cv0) mid[0] = *(const f32x4*)(rm - c);
      mid[1] = *(const f32x4*)rm;
      if (cv2) mid[2] = *(const f32x4*)(rm + c);
      if (botv) {
        const float* rb = rm + (size_t)w * c;
        if (cv0) bot[0] = *(const f32x4*)(rb - c);
        bot[1] = *(const f32x4*)rb;
        if (cv2) bot[2] = *(const f32x4*)(rb + c);
      }
      f32x4 best = ninf;
      int bi[4] = {0, 0, 0, 0};
      bool any = false;
#define HND_POOL_TAP(v, ok, tap)                                                          \
  if (ok) {                                                                               \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                         \
      if (!any || (v)[k] > best[k] || (v)[k] != (v)[k]) { best[k] = (v)[k]; bi[k] = (tap); } \
    any = true;                                                                           \
  }
      HND_POOL_TAP(top[0], topv && cv0, 0)
      HND_POOL_TAP(top[1], topv, 1)
      HND_POOL_TAP(top[2], topv && cv2, 2)
      HND_POOL_TAP(mid[0], cv0, 3)
      HND_POOL_TAP(mid[1], true, 4)
      HND_POOL_TAP(mid[2], cv2, 5)
      HND_POOL_TAP(bot[0], botv && cv0, 6)
      HND_POOL_TAP(bot[1], botv, 7)
      HND_POOL_TAP(bot[2], botv && cv2, 8)
#undef HND_POOL_TAP
      const size_t o = (((size_t)b * oh + oy) * ow + ox) * c + c4 * 4;
      *(f32x4*)(y + o) = best;
      *(uchar4*)(idx + o) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2],
                                        (unsigned char)bi[3]);
      top[0] = bot[0]; top[1] = bot[1]; top[2] = bot[2];
      topv = botv;
    }
  }
}

// Backward of the pool fused with the stem's ReLU and FrozenBN scale.  One thread owns the 2 x 2 input pixels (2a..2a+1,
// 2b..2b+1) of four channels: they lie in the windows (a..a+1, b..b+1) and nowhere else (an even coordinate belongs to one
// window, an odd one to two), so four (index, gradient) loads serve four outputs -- the one-pixel-per-thread version
// loaded 2.25 windows per pixel behind data-dependent loop bounds.  Each pixel adds its windows in the order (a,b), (a,b+1),
// (a+1,b), (a+1,b+1), as before.
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                   const float* __restrict__ act, const float* __restrict__ scale,
                                   float* __restrict__ dx, int n, int h, int w, int c, int oh, int ow,
                                   const PoolDivs dv) {
  const unsigned c4n = c >> 2;
  const unsigned total = (unsigned)n * oh * ow * c4n;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    unsigned p = hnd::fdiv(e, dv.c4n);
    const int c4 = (int)(e - p * c4n);
    unsigned q = hnd::fdiv(p, dv.w);
    const int qb = (int)(p - q * ow);
    const unsigned b = hnd::fdiv(q, dv.h);
    const int qa = (int)(q - b * oh);
    const int iy = 2 * qa, ix = 2 * qb;
    const bool row1 = iy + 1 < h, col1 = ix + 1 < w;            // (iy < h and ix < w always: oh = ceil(h / 2))
    const bool wrow1 = qa + 1 < oh, wcol1 = qb + 1 < ow;
    const size_t o00 = (((size_t)b * oh + qa) * ow + qb) * c + c4 * 4;
    const size_t o01 = o00 + c, o10 = o00 + (size_t)ow * c, o11 = o10 + c;
    const size_t x00 = (((size_t)b * h + iy) * w + ix) * c + c4 * 4;
    const size_t x01 = x00 + c, x10 = x00 + (size_t)w * c, x11 = x10 + c;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const uchar4 zu = make_uchar4(255, 255, 255, 255);
    // all loads first
    const uchar4 i00 = *(const uchar4*)(idx + o00);
    const f32x4 d00 = *(const f32x4*)(dy + o00);
    const uchar4 i01 = wcol1 ? *(const uchar4*)(idx + o01) : zu;
    const f32x4 d01 = wcol1 ? *(const f32x4*)(dy + o01) : z4;
    const uchar4 i10 = wrow1 ? *(const uchar4*)(idx + o10) : zu;
    const f32x4 d10 = wrow1 ? *(const f32x4*)(dy + o10) : z4;
    const uchar4 i11 = (wrow1 && wcol1) ? *(const uchar4*)(idx + o11) : zu;
    const f32x4 d11 = (wrow1 && wcol1) ? *(const f32x4*)(dy + o11) : z4;
    const f32x4 a00 = HND_NT_LOAD4(act + x00);
    const f32x4 a01 = col1 ? HND_NT_LOAD4(act + x01) : z4;
    const f32x4 a10 = row1 ? HND_NT_LOAD4(act + x10) : z4;
    const f32x4 a11 = (row1 && col1) ? HND_NT_LOAD4(act + x11) : z4;
    const f32x4 s = *(const f32x4*)(scale + c4 * 4);
#define HND_TAKE(g, id, d, tap)            \
  if ((id).x == (tap)) (g).x += (d).x;     \
  if ((id).y == (tap)) (g).y += (d).y;     \
  if ((id).z == (tap)) (g).z += (d).z;     \
  if ((id).w == (tap)) (g).w += (d).w;
#define HND_PUT(xo, a, g)                                                        \
  {                                                                              \
    f32x4 r;                                                                     \
    r.x = (a).x > 0.f ? (g).x * s.x : 0.f; r.y = (a).y > 0.f ? (g).y * s.y : 0.f; \
    r.z = (a).z > 0.f ? (g).z * s.z : 0.f; r.w = (a).w > 0.f ? (g).w * s.w : 0.f; \
    *(f32x4*)(dx + (xo)) = r;                                                    \
  }
    f32x4 g = z4;                        // (2a, 2b): centre of window (a, b)
    HND_TAKE(g, i00, d00, 4)
    HND_PUT(x00, a00, g)
    if (col1) {                          // (2a, 2b+1): right of (a, b), left of (a, b+1)
      g = z4;
      HND_TAKE(g, i00, d00, 5)
      HND_TAKE(g, i01, d01, 3)
      HND_PUT(x01, a01, g)
    }
    if (row1) {                          // (2a+1, 2b): below (a, b), above (a+1, b)
      g = z4;
      HND_TAKE(g, i00, d00, 7)
      HND_TAKE(g, i10, d10, 1)
      HND_PUT(x10, a10, g)
    }
    if (row1 && col1) {                  // (2a+1, 2b+1): a corner of all four windows
      g = z4;
      HND_TAKE(g, i00, d00, 8)
      HND_TAKE(g, i01, d01, 6)
      HND_TAKE(g, i10, d10, 2)
      HND_TAKE(g, i11, d11, 0)
      HND_PUT(x11, a11, g)
    }
#undef HND_TAKE
#undef HND_PUT
  }
}

// ------------------------------------------------------------------------------------ BN forward
// One 1024-thread block per 16 channels.  A wave's load covers 4 tiles x 16 channels (four 64-byte runs instead of 64
// scattered dwords), the 16 waves stride over the tiles 64 at a time; fp64 accumulation in a fixed order: per lane in
// increasing tile order, then the 4 tile sub-lanes (xor 16, xor 32), then the 16 waves in wave order.
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const float* __restrict__ partials, int ntiles, int c, int cs,
                                   double count, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, long long* nbt, float momentum, float eps, float* scale,
                                   float* shift, float* save_mean, float* save_rstd) {
  __shared__ double red[2][16][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & 15, ts = lane >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (ch < c) {
    // eight loads in flight per lane (the grid is only c / 16 blocks: the loop ran at the memory latency), added in
    // the original order
    int t = wave * 4 + ts;
    for (; t + 3 * 64 < ntiles; t += 4 * 64) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partials[((size_t)(t + u * 64) * 2 + 0) * cs + ch];
        b[u] = partials[((size_t)(t + u * 64) * 2 + 1) * cs + ch];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
    for (; t < ntiles; t += 64) {
      s1 += (double)partials[((size_t)t * 2 + 0) * cs + ch];
      s2 += (double)partials[((size_t)t * 2 + 1) * cs + ch];
    }
  }
  s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
  if (ts == 0) { red[0][wave][cl] = s1; red[1][wave][cl] = s2; }
  __syncthreads();
  if (threadIdx.x >= 16 || ch >= cs) return;
  if (ch >= c) {
    scale[ch] = 0.f; shift[ch] = 0.f; save_mean[ch] = 0.f; save_rstd[ch] = 0.f;
    return;
  }
  s1 = 0.0; s2 = 0.0;
  for (int w = 0; w < 16; ++w) { s1 += red[0][w][cl]; s2 += red[1][w][cl]; }
  {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;   // biased, used for normalisation
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * rstd;
    scale[ch] = sc;
    shift[ch] = beta[ch] - (float)mean * sc;
    save_mean[ch] = (float)mean;
    save_rstd[ch] = rstd;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
    if (nbt && ch == 0) *nbt += 1;
  }
}

__global__ void affine_relu_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ y, long long n4, int cs4,
                                   int relu, uint8_t* __restrict__ mask_out) {
  const bool fixed = (256 % cs4) == 0;               // this lane's channel group never changes
  const int c4f = threadIdx.x % cs4;
  f32x4 s = *(const f32x4*)(scale + c4f * 4), b = *(const f32x4*)(shift + c4f * 4);
  const long long chunk = 256ll * kEwU;
  for (long long base = blockIdx.x * chunk + threadIdx.x; base < n4; base += (long long)gridDim.x * chunk) {
    f32x4 v[kEwU];
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e < n4) v[u] = HND_NT_LOAD4(x + e * 4);
    }
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e >= n4) continue;
      if (!fixed) {
        const int c4 = (int)(e % cs4);
        s = *(const f32x4*)(scale + c4 * 4);
        b = *(const f32x4*)(shift + c4 * 4);
      }
      f32x4 r = v[u] * s + b;
      if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      *(f32x4*)(y + e * 4) = r;
      if (mask_out)     // ReLU-mask nibble of these four channels (hnd_conv_desc.mask_out's layout)
        mask_out[e] = (uint8_t)((r.x > 0.f ? 1 : 0) | (r.y > 0.f ? 2 : 0) | (r.z > 0.f ? 4 : 0) | (r.w > 0.f ? 8 : 0));
    }
  }
}

__global__ void relu_mask_nibbles_kernel(const float* __restrict__ x, uint8_t* __restrict__ bits, long long n4) {
  const long long chunk = 256ll * kEwU;
  for (long long base = blockIdx.x * chunk + threadIdx.x; base < n4; base += (long long)gridDim.x * chunk) {
    f32x4 v[kEwU];
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e < n4) v[u] = *(const f32x4*)(x + e * 4);        // (default policy: the data gradient re-reads nothing of x)
    }
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e < n4)
        bits[e] = (uint8_t)((v[u].x > 0.f ? 1 : 0) | (v[u].y > 0.f ? 2 : 0) | (v[u].z > 0.f ? 4 : 0) | (v[u].w > 0.f ? 8 : 0));
    }
  }
}

// ------------------------------------------------------------------------------------ BN backward
constexpr int kBnTilePix = 1024;   // pixels per block in the reduce pass

// partials[tile][0][ch] = sum d, partials[tile][1][ch] = sum d*xhat, d = relu-masked g
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                     long long npix, int cs, float* __restrict__ partials) {
  __shared__ float red[2][256][4];
  const int cs4 = cs >> 2;                 // float4 groups per pixel: 1, 16, 32 or 64 (divides 256)
  const int rows = 256 / cs4;              // pixels per pass
  const int c4 = threadIdx.x % cs4, r = threadIdx.x / cs4;
  const f32x4 sc = *(const f32x4*)(scale + c4 * 4), sh = *(const f32x4*)(shift + c4 * 4);
  const f32x4 mu = *(const f32x4*)(mean + c4 * 4), rs = *(const f32x4*)(rstd + c4 * 4);
  const long long p0 = (long long)blockIdx.x * kBnTilePix;
  long long p1 = p0 + kBnTilePix;
  if (p1 > npix) p1 = npix;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  // (kEwU pixels of both operands in flight per thread, summed in the original order)
  for (long long pb = p0 + r; pb < p1; pb += (long long)rows * kEwU) {
    f32x4 dv[kEwU], xq[kEwU];
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long p = pb + (long long)rows * u;
      if (p < p1) {
        const size_t o = (size_t)p * cs + c4 * 4;
        dv[u] = HND_NT_LOAD4(g + o);
        xq[u] = HND_NT_LOAD4(x + o);
      }
    }
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      if (pb + (long long)rows * u >= p1) continue;
      f32x4 d = dv[u];
      const f32x4 xv = xq[u];
      if (relu) {
        const f32x4 out = xv * sc + sh;
        d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
        d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
      }
      s1 += d;
      s2 += d * ((xv - mu) * rs);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][threadIdx.x][k] = s1[k]; red[1][threadIdx.x][k] = s2[k]; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < cs * 2; idx += 256) {
    const int which = idx / cs, ch = idx - which * cs;
    const int g4 = ch >> 2, k = ch & 3;
    float s = 0.f;
    for (int rr = 0; rr < rows; ++rr) s += red[which][rr * cs4 + g4][k];
    partials[((size_t)blockIdx.x * 2 + which) * cs + ch] = s;
  }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int ntiles, int c, int cs, double count,
                                       const float* gamma, const float* mean, const float* rstd, float* dgamma,
                                       float* dbeta, float* k123) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ch >= cs) return;
  if (ch >= c) {
    if (lane == 0) { k123[ch] = 0.f; k123[cs + ch] = 0.f; k123[2 * cs + ch] = 0.f; }
    return;
  }
  double s1 = 0.0, s2 = 0.0;
  for (int t = lane; t < ntiles; t += 64) {
    s1 += (double)partials[((size_t)t * 2 + 0) * cs + ch];
    s2 += (double)partials[((size_t)t * 2 + 1) * cs + ch];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
    const double k1 = (double)gamma[ch] * (double)rstd[ch];
    const double k2 = -k1 * (double)rstd[ch] * s2 / count;
    const double k3 = k1 * ((double)mean[ch] * (double)rstd[ch] * s2 - s1) / count;
    k123[ch] = (float)k1;
    k123[cs + ch] = (float)k2;
    k123[2 * cs + ch] = (float)k3;
  }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ k123, int relu, float* __restrict__ dx, long long n4,
                                    int cs) {
  const int cs4 = cs >> 2;
  const bool fixed = (256 % cs4) == 0;
  int c4 = threadIdx.x % cs4;
  f32x4 sc = *(const f32x4*)(scale + c4 * 4), sh = *(const f32x4*)(shift + c4 * 4);
  f32x4 k1 = *(const f32x4*)(k123 + c4 * 4), k2 = *(const f32x4*)(k123 + cs + c4 * 4),
        k3 = *(const f32x4*)(k123 + 2 * cs + c4 * 4);
  const long long chunk = 256ll * kEwU;
  for (long long base = blockIdx.x * chunk + threadIdx.x; base < n4; base += (long long)gridDim.x * chunk) {
    f32x4 dv[kEwU], xv[kEwU];
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e < n4) {
        dv[u] = HND_NT_LOAD4(g + e * 4);
        xv[u] = HND_NT_LOAD4(x + e * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e >= n4) continue;
      if (!fixed) {
        c4 = (int)(e % cs4);
        sc = *(const f32x4*)(scale + c4 * 4); sh = *(const f32x4*)(shift + c4 * 4);
        k1 = *(const f32x4*)(k123 + c4 * 4); k2 = *(const f32x4*)(k123 + cs + c4 * 4);
        k3 = *(const f32x4*)(k123 + 2 * cs + c4 * 4);
      }
      f32x4 d = dv[u];
      if (relu) {
        const f32x4 out = xv[u] * sc + sh;
        d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
        d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
      }
      *(f32x4*)(dx + e * 4) = k1 * d + k2 * xv[u] + k3;
    }
  }
}

// ------------------------------------------------------------------------------------ loss
constexpr int kMseBlocks = 1024;   // (256 ... 16384 measured: 5.5 ... 4.8 TB/s, flat below 2048)
constexpr int kMaxPairs = 8;
struct MseArgs {
  hnd_mse_pair pair[kMaxPairs];
  int first_block[kMaxPairs + 1];
  int npairs;
};

__global__ void mse_kernel(const MseArgs a, double* __restrict__ scratch) {
  __shared__ double wsum[4];
  int k = 0;
  while (k + 1 < a.npairs && (int)blockIdx.x >= a.first_block[k + 1]) ++k;
  const hnd_mse_pair P = a.pair[k];
  const int nb = a.first_block[k + 1] - a.first_block[k], lb = blockIdx.x - a.first_block[k];
  const long long n4 = P.numel >> 2;                 // numel is a multiple of 4 (NHWC, C % 4 == 0)
  const float gf = 2.f * P.factor;
  float acc = 0.f;
  // The pair's nb workgroups take its 16 KB chunks (256 * kEwU vectors per tensor) round-robin, so the chunks in flight at
  // any moment are neighbours in memory (a contiguous range per workgroup kept nb * 3 distant streams open: 4.9 TB/s
  // against 6.0 of a plain elementwise kernel); kEwU vectors of each operand in flight per thread.  Fixed order: per thread
  // in increasing chunk order, then the wave, then the four waves -- bit-reproducible run to run.
  const long long chunk = 256ll * kEwU;
  for (long long base = (long long)lb * chunk + threadIdx.x; base < n4; base += (long long)nb * chunk) {
    f32x4 tv[kEwU], sv[kEwU];
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e < n4) {
        tv[u] = *(const f32x4*)(P.teacher + e * 4);
        sv[u] = *(const f32x4*)(P.student + e * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < kEwU; ++u) {
      const long long e = base + 256ll * u;
      if (e >= n4) continue;
      const f32x4 s = sv[u];
      const f32x4 df = s - tv[u];
      acc += (df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w);
      if (P.grad) {
        f32x4 gg = df * gf;
        if (P.relu_mask) {
          gg.x = s.x > 0.f ? gg.x : 0.f; gg.y = s.y > 0.f ? gg.y : 0.f;
          gg.z = s.z > 0.f ? gg.z : 0.f; gg.w = s.w > 0.f ? gg.w : 0.f;
        }
        *(f32x4*)(P.grad + e * 4) = gg;
      }
    }
  }
  double dsum = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = dsum;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void mse_finalize_kernel(const MseArgs a, const double* __restrict__ scratch, double* __restrict__ out) {
  __shared__ double term[kMaxPairs];
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (k < a.npairs) {
    double s = 0.0;
    for (int b = a.first_block[k] + lane; b < a.first_block[k + 1]; b += 64) s += scratch[b];
    s = wave_sum_d(s);
    if (lane == 0) {
      term[k] = s * (double)a.pair[k].factor;
      out[1 + k] = term[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < a.npairs; ++i) tot += term[i];
    out[0] = tot;
  }
}

__global__ void scale_by_scalar_kernel(float* x, long long n, const float* s) {
  const float f = *s;
  if (f == 1.0f) return;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    x[e] *= f;
}

// ------------------------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float beta1, float beta2, float eps,
                            float bc2_sqrt, float grad_scale) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const float gr = g[e] * grad_scale;
    const float mm = m[e] + (gr - m[e]) * (1.f - beta1);          // exp_avg.lerp_(grad, 1-beta1)
    const float vv = v[e] * beta2 + (1.f - beta2) * gr * gr;      // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    m[e] = mm;
    v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] = p[e] - step_size * (mm / denom);
  }
}

__global__ void subsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c,
                                  int oh, int ow) {
  const int c4n = c >> 2;
  const long long total = (long long)n * oh * ow * c4n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    long long p = e / c4n;
    const int ox = (int)(p % ow);
    p /= ow;
    const int oy = (int)(p % oh), b = (int)(p / oh);
    *(f32x4*)(y + e * 4) = *(const f32x4*)(x + (((size_t)b * h + oy * 2) * w + ox * 2) * c + c4 * 4);
  }
}

// backward of F.interpolate(coarse, size=(H, W), mode='nearest') (the FPN's top-down path, torchvision 0.4.2
// ops/feature_pyramid_network.py): fine pixel (Y, X) reads coarse pixel (floor(Y h / H), floor(X w / W)), so coarse pixel
// (y, x) receives the sum over its preimage rows [ceil(y H / h), ceil((y + 1) H / h)) x columns likewise -- a fixed order
// (row-major), no atomics.  accumulate: added to what g_coarse holds.
__global__ void upsample_nearest_bwd_kernel(const float* __restrict__ g_fine, float* __restrict__ g_coarse, int n, int H,
                                            int W, int h, int w, int c, int accumulate) {
  const int c4n = c >> 2;
  const long long total = (long long)n * h * w * c4n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    long long p = e / c4n;
    const int x = (int)(p % w);
    p /= w;
    const int y = (int)(p % h), b = (int)(p / h);
    const int y0 = (int)(((long long)y * H + h - 1) / h), y1 = (int)(((long long)(y + 1) * H + h - 1) / h);
    const int x0 = (int)(((long long)x * W + w - 1) / w), x1 = (int)(((long long)(x + 1) * W + w - 1) / w);
    f32x4 acc = accumulate ? *(const f32x4*)(g_coarse + e * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int yy = y0; yy < y1 && yy < H; ++yy)
      for (int xx = x0; xx < x1 && xx < W; ++xx)
        acc += *(const f32x4*)(g_fine + (((size_t)b * H + yy) * W + xx) * c + c4 * 4);
    *(f32x4*)(g_coarse + e * 4) = acc;
  }
}

__global__ void add_inplace_kernel(float* __restrict__ x, const float* __restrict__ y, long long n4) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *(const f32x4*)(x + e * 4);
    v += *(const f32x4*)(y + e * 4);
    *(f32x4*)(x + e * 4) = v;
  }
}

__global__ void fill_kernel(float* x, long long n, float v) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    x[e] = v;
}

// ------------------------------------------------------------------------------------ bottleneck codec
constexpr int kMinMaxBlocks = 512;

__global__ void minmax_kernel(const float* __restrict__ x, long long npix, int c, int cs, float* __restrict__ part) {
  __shared__ float smin[4], smax[4];
  float lo = INFINITY, hi = -INFINITY;
  const long long total = npix * cs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    if ((int)(e % cs) < c) {
      const float v = x[e];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2 + 0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

// The codec is byte work: it must reproduce myutils.tensor_util.quantize_tensor bit for bit on identical fp32 input
// (tests/test_ops_gpu.py, torch.equal on the bytes).  Every operation below is therefore a single correctly rounded
// IEEE fp32 operation in the reference's order.  hipcc's __fdiv_rn / __fadd_rn / __fmul_rn are the plain operators,
// so FMA contraction is switched off inside these three kernels (none of the expressions below has a multiply feeding
// an add today; the pragma keeps it that way) and the division is the correctly rounded one (v_div_scale / fmas /
// fixup sequence, checked in the ISA).
__global__ void qparams_kernel(const float* __restrict__ part, int nblocks, float qmax, float* __restrict__ qp) {
#pragma clang fp contract(off)
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    lo = fminf(lo, part[i * 2]);
    hi = fmaxf(hi, part[i * 2 + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  if (threadIdx.x == 0) {
    const float scale = __fdiv_rn(__fsub_rn(hi, lo), qmax);        // (max - min) / (qmax - qmin), qmin = 0
    float zp = __fsub_rn(0.f, __fdiv_rn(lo, scale));               // qmin - min / scale
    zp = zp < 0.f ? 0.f : (zp > qmax ? qmax : zp);
    qp[0] = lo; qp[1] = hi; qp[2] = scale; qp[3] = truncf(zp);     // int(zero_point)
  }
}

__global__ void quantize_kernel(const float* __restrict__ x, long long npix, int c, int cs,
                                const float* __restrict__ qp, float qmax, uint8_t* __restrict__ q) {
#pragma clang fp contract(off)
  const float scale = qp[2], zp = qp[3];
  const long long total = npix * cs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if ((int)(e % cs) < c) {
      v = __fadd_rn(zp, __fdiv_rn(x[e], scale));                   // zero_point + x / scale
      v = v < 0.f ? 0.f : (v > qmax ? qmax : v);                   // clamp_(qmin, qmax)
      v = rintf(v);                                                // round_(): half to even
    }
    q[e] = (uint8_t)v;
  }
}

__global__ void dequantize_kernel(const uint8_t* __restrict__ q, const float* __restrict__ qp, float* __restrict__ x,
                                  long long npix, int c, int cs) {
#pragma clang fp contract(off)
  const float scale = qp[2], zp = qp[3];
  const long long total = npix * cs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x)
    x[e] = ((int)(e % cs) < c) ? __fmul_rn(scale, __fsub_rn((float)q[e], zp)) : 0.f;      // scale * (q - zero_point)
}

__global__ void roundtrip_f16_kernel(float* x, long long n) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    x[e] = __half2float(__float2half(x[e]));
}

}  // namespace

// ======================================================================================== C ABI
extern "C" {

static int pack_args(const hnd_pack_desc& p, PackArgs& a, const char* who, int idx) {
  HND_REQUIRE(p.src && p.dst, "%s: operand %d: null pointer", who, idx);
  HND_REQUIRE(p.cout > 0 && p.cin > 0 && p.kh > 0 && p.kw > 0 && p.ni > 0 && p.nj > 0 && p.istep > 0 && p.jstep > 0,
              "%s: operand %d: bad geometry", who, idx);
  HND_REQUIRE(p.i0 >= 0 && p.i0 + (p.ni - 1) * p.istep < p.kh && p.j0 >= 0 && p.j0 + (p.nj - 1) * p.jstep < p.kw,
              "%s: operand %d: tap sub-grid outside the kernel", who, idx);
  const int chans = p.transposed ? p.cout : p.cin, rows = p.transposed ? p.cin : p.cout;
  HND_REQUIRE(p.chan_pad >= chans, "%s: operand %d: chan_pad < channels", who, idx);
  a.src = p.src; a.dst = p.dst;
  a.cout = p.cout; a.cin = p.cin; a.kh = p.kh; a.kw = p.kw; a.transposed = p.transposed; a.chan_pad = p.chan_pad;
  a.i0 = p.i0; a.istep = p.istep; a.ni = p.ni; a.j0 = p.j0; a.jstep = p.jstep; a.nj = p.nj;
  a.rows_pad = (rows + 63) / 64 * 64;
  a.kdim = (p.ni * p.nj * p.chan_pad + 31) / 32 * 32;
  return HND_OK;
}

int hnd_pack_weights(const float* src, float* dst, int cout, int cin, int kh, int kw, int transposed, int chan_pad,
                     int i0, int istep, int ni, int j0, int jstep, int nj, void* stream) {
  const hnd_pack_desc p{src, dst, cout, cin, kh, kw, transposed, chan_pad, i0, istep, ni, j0, jstep, nj};
  PackArgs a;
  int rc = pack_args(p, a, "hnd_pack_weights", 0);
  if (rc) return rc;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for((long long)a.rows_pad * a.kdim)), dim3(256), 0,
                     hnd::as_stream(stream), a);
  return hnd::check_launch("hnd_pack_weights");
}

int hnd_pack_weights_batched(const hnd_pack_desc* ops_, int count, void* stream) {
  HND_REQUIRE(ops_ && count > 0, "hnd_pack_weights_batched: bad arguments");
  for (int i0 = 0; i0 < count; i0 += kMaxBatchPacks) {
    const int nb = count - i0 < kMaxBatchPacks ? count - i0 : kMaxBatchPacks;
    PackBatch b;
    long long most = 0;
    for (int k = 0; k < nb; ++k) {
      int rc = pack_args(ops_[i0 + k], b.op[k], "hnd_pack_weights_batched", i0 + k);
      if (rc) return rc;
      const long long t = (long long)b.op[k].rows_pad * b.op[k].kdim;
      if (t > most) most = t;
    }
    int bx = grid_for(most);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(bx, nb), dim3(256), 0, hnd::as_stream(stream), b);
    int rc = hnd::check_launch("hnd_pack_weights_batched");
    if (rc) return rc;
  }
  return HND_OK;
}

int hnd_scale_packed_k(float* packed, int rows_pad, int kdim, int ntaps, int chan_pad, const float* scale, int nscale,
                       void* stream) {
  HND_REQUIRE(packed && scale && rows_pad > 0 && kdim > 0 && ntaps > 0 && chan_pad > 0 && nscale > 0 &&
                  nscale <= chan_pad && (long long)ntaps * chan_pad <= kdim, "hnd_scale_packed_k: bad arguments");
  const long long total = (long long)rows_pad * kdim;
  hipLaunchKernelGGL(scale_packed_k_kernel, dim3(grid_for(total)), dim3(256), 0, hnd::as_stream(stream), packed, scale,
                     total, kdim, ntaps, chan_pad, nscale);
  return hnd::check_launch("hnd_scale_packed_k");
}

int hnd_fbn_fold(const float* weight, const float* bias, const float* mean, const float* var, float* scale,
                 float* shift, int c, int cs, float eps, void* stream) {
  HND_REQUIRE(weight && bias && mean && var && scale && shift && c > 0 && cs >= c, "hnd_fbn_fold: bad arguments");
  hipLaunchKernelGGL(fbn_fold_kernel, dim3((cs + 255) / 256), dim3(256), 0, hnd::as_stream(stream), weight, bias, mean,
                     var, scale, shift, c, cs, eps);
  return hnd::check_launch("hnd_fbn_fold");
}

int hnd_transform_image(const float* src, int h, int w, float* dst, int index, int out_h, int out_w, int hp, int wp,
                        float scale_h, float scale_w, const float mean[3], const float std[3], void* stream) {
  HND_REQUIRE(src && dst && mean && std, "hnd_transform_image: null pointer");
  HND_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0 && out_h <= hp && out_w <= wp && index >= 0,
              "hnd_transform_image: bad geometry (out %dx%d, padded %dx%d)", out_h, out_w, hp, wp);
  TransformArgs a;
  a.src = src;
  a.src8 = nullptr;
  a.hwc = 0;
  a.flip = 0;
  a.dst = dst + (size_t)index * hp * wp * 4;
  a.h = h; a.w = w; a.out_h = out_h; a.out_w = out_w; a.hp = hp; a.wp = wp;
  a.rh = scale_h; a.rw = scale_w;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean[i]; a.std[i] = std[i]; a.inv_unused[i] = 0.f; }
  hipLaunchKernelGGL(transform_kernel<false>, dim3(grid_for((long long)hp * wp)), dim3(256), 0,
                     hnd::as_stream(stream), a);
  return hnd::check_launch("hnd_transform_image");
}

int hnd_transform_image_u8(const uint8_t* src, int h, int w, int hwc, int flip, float* dst, int index, int out_h,
                           int out_w, int hp, int wp, float scale_h, float scale_w, const float mean[3],
                           const float std[3], void* stream) {
  HND_REQUIRE(src && dst && mean && std, "hnd_transform_image_u8: null pointer");
  HND_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0 && out_h <= hp && out_w <= wp && index >= 0,
              "hnd_transform_image_u8: bad geometry (out %dx%d, padded %dx%d)", out_h, out_w, hp, wp);
  TransformArgs a;
  a.src = nullptr;
  a.src8 = src;
  a.hwc = hwc != 0;
  a.flip = flip != 0;
  a.dst = dst + (size_t)index * hp * wp * 4;
  a.h = h; a.w = w; a.out_h = out_h; a.out_w = out_w; a.hp = hp; a.wp = wp;
  a.rh = scale_h; a.rw = scale_w;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean[i]; a.std[i] = std[i]; a.inv_unused[i] = 0.f; }
  hipLaunchKernelGGL(transform_kernel<true>, dim3(grid_for((long long)hp * wp)), dim3(256), 0,
                     hnd::as_stream(stream), a);
  return hnd::check_launch("hnd_transform_image_u8");
}

int hnd_transform_images(const hnd_image_desc* imgs, int count, float* dst, int hp, int wp, const float mean[3],
                         const float std[3], void* stream) {
  HND_REQUIRE(imgs && dst && mean && std && count > 0 && hp > 0 && wp > 0, "hnd_transform_images: bad arguments");
  int bx = grid_for((long long)hp * wp);
  if (bx > 2048) bx = 2048;
  for (int u8 = 0; u8 < 2; ++u8) {                      // one launch per source type present (normally one in total)
    TransformBatch b;
    int nb = 0;
    auto flush = [&]() -> int {
      if (nb == 0) return HND_OK;
      if (u8) hipLaunchKernelGGL(transform_batch_kernel<true>, dim3(bx, nb), dim3(256), 0, hnd::as_stream(stream), b);
      else hipLaunchKernelGGL(transform_batch_kernel<false>, dim3(bx, nb), dim3(256), 0, hnd::as_stream(stream), b);
      nb = 0;
      return hnd::check_launch("hnd_transform_images");
    };
    for (int i = 0; i < count; ++i) {
      const hnd_image_desc& im = imgs[i];
      if ((im.is_u8 != 0) != (u8 != 0)) continue;
      HND_REQUIRE(im.src && im.h > 0 && im.w > 0 && im.out_h > 0 && im.out_w > 0 && im.out_h <= hp && im.out_w <= wp,
                  "hnd_transform_images: image %d: bad geometry (out %dx%d, padded %dx%d)", i, im.out_h, im.out_w, hp, wp);
      TransformArgs& a = b.img[nb++];
      a.src = u8 ? nullptr : (const float*)im.src;
      a.src8 = u8 ? (const uint8_t*)im.src : nullptr;
      a.hwc = im.hwc != 0;
      a.flip = im.flip != 0;
      a.dst = dst + (size_t)i * hp * wp * 4;
      a.h = im.h; a.w = im.w; a.out_h = im.out_h; a.out_w = im.out_w; a.hp = hp; a.wp = wp;
      a.rh = im.scale_h; a.rw = im.scale_w;
      for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = std[c]; a.inv_unused[c] = 0.f; }
      if (nb == kMaxBatchImages) {
        int rc = flush();
        if (rc) return rc;
      }
    }
    int rc = flush();
    if (rc) return rc;
  }
  return HND_OK;
}

int hnd_scale_boxes(const hnd_boxes_desc* items, int count, void* stream) {
  HND_REQUIRE(items && count > 0, "hnd_scale_boxes: bad arguments");
  for (int i0 = 0; i0 < count; i0 += kMaxBatchImages) {
    const int nb = count - i0 < kMaxBatchImages ? count - i0 : kMaxBatchImages;
    BoxScaleBatch b;
    int kmax = 0;
    for (int j = 0; j < nb; ++j) {
      const hnd_boxes_desc& it = items[i0 + j];
      HND_REQUIRE(it.k >= 0 && (it.k == 0 || (it.src && it.dst)), "hnd_scale_boxes: image %d: null pointer", i0 + j);
      b.src[j] = it.src; b.dst[j] = it.dst; b.k[j] = it.k; b.rw[j] = it.scale_w; b.rh[j] = it.scale_h;
      if (it.k > kmax) kmax = it.k;
    }
    if (kmax == 0) continue;
    hipLaunchKernelGGL(scale_boxes_batch_kernel, dim3((kmax * 4 + 255) / 256, nb), dim3(256), 0, hnd::as_stream(stream),
                       b);
    int rc = hnd::check_launch("hnd_scale_boxes");
    if (rc) return rc;
  }
  return HND_OK;
}

int hnd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int oh, int ow,
                         void* stream) {
  HND_REQUIRE(x && y && idx, "hnd_maxpool3x3s2_fwd: null pointer");
  HND_REQUIRE(c % 4 == 0 && oh == (h + 2 - 3) / 2 + 1 && ow == (w + 2 - 3) / 2 + 1,
              "hnd_maxpool3x3s2_fwd: bad geometry");
  HND_REQUIRE((long long)n * h * w * (c / 4) < (1ll << 31), "hnd_maxpool3x3s2_fwd: more than 2^31 vectors");
  PoolDivs dv;
  dv.c4n = hnd::make_fastdiv((unsigned)(c / 4));
  const int nstrips = (oh + kPoolStrip - 1) / kPoolStrip;
  dv.w = hnd::make_fastdiv((unsigned)ow);
  dv.h = hnd::make_fastdiv((unsigned)nstrips);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)n * nstrips * ow * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), x, y, idx, n, h, w, c, oh, ow, nstrips, dv);
  return hnd::check_launch("hnd_maxpool3x3s2_fwd");
}

int hnd_maxpool3x3s2_bwd_relu_scale(const float* dy, const uint8_t* idx, const float* act, const float* fbn_scale,
                                    float* dx, int n, int h, int w, int c, int oh, int ow, void* stream) {
  HND_REQUIRE(dy && idx && act && fbn_scale && dx, "hnd_maxpool3x3s2_bwd_relu_scale: null pointer");
  HND_REQUIRE(c % 4 == 0, "hnd_maxpool3x3s2_bwd_relu_scale: c %% 4 != 0");
  HND_REQUIRE((long long)n * h * w * (c / 4) < (1ll << 31), "hnd_maxpool3x3s2_bwd_relu_scale: more than 2^31 vectors");
  PoolDivs dv;
  dv.c4n = hnd::make_fastdiv((unsigned)(c / 4));
  HND_REQUIRE(oh == (h + 1) / 2 && ow == (w + 1) / 2, "hnd_maxpool3x3s2_bwd_relu_scale: bad geometry");
  dv.w = hnd::make_fastdiv((unsigned)ow);
  dv.h = hnd::make_fastdiv((unsigned)oh);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)n * oh * ow * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), dy, idx, act, fbn_scale, dx, n, h, w, c, oh, ow, dv);
  return hnd::check_launch("hnd_maxpool3x3s2_bwd_relu_scale");
}

int hnd_bn_finalize(const float* partials, int ntiles, int c, int cs, int64_t count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_rstd,
                    void* stream) {
  HND_REQUIRE(partials && gamma && beta && scale && shift && save_mean && save_rstd, "hnd_bn_finalize: null pointer");
  HND_REQUIRE(ntiles > 0 && c > 0 && cs >= c && count > 0, "hnd_bn_finalize: bad sizes");
  HND_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "hnd_bn_finalize: running stats mismatch");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((cs + 15) / 16), dim3(1024), 0, hnd::as_stream(stream), partials, ntiles, c,
                     cs, (double)count, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked,
                     momentum, eps, scale, shift, save_mean, save_rstd);
  return hnd::check_launch("hnd_bn_finalize");
}

int hnd_affine_relu(const float* x, const float* scale, const float* shift, float* y, int64_t npix, int cs, int relu,
                    uint8_t* mask_out, void* stream) {
  HND_REQUIRE(x && scale && shift && y && npix > 0 && cs > 0 && cs % 4 == 0, "hnd_affine_relu: bad arguments");
  const long long n4 = (long long)npix * (cs / 4);
  hipLaunchKernelGGL(affine_relu_kernel, dim3(grid_for_chunks(n4)), dim3(256), 0, hnd::as_stream(stream), x, scale,
                     shift, y, n4, cs / 4, relu, mask_out);
  return hnd::check_launch("hnd_affine_relu");
}

int hnd_relu_mask_nibbles(const float* x, uint8_t* bits, int64_t n4, void* stream) {
  HND_REQUIRE(x && bits && n4 > 0 && ((uintptr_t)x % 16) == 0, "hnd_relu_mask_nibbles: bad arguments");
  hipLaunchKernelGGL(relu_mask_nibbles_kernel, dim3(grid_for_chunks(n4)), dim3(256), 0, hnd::as_stream(stream), x, bits,
                     (long long)n4);
  return hnd::check_launch("hnd_relu_mask_nibbles");
}

int hnd_bn_bwd_ntiles(int64_t npix) { return (int)((npix + kBnTilePix - 1) / kBnTilePix); }

int hnd_bn_bwd_reduce(const float* g, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* rstd, int relu, int64_t npix, int cs, float* partials, void* stream) {
  HND_REQUIRE(g && x && scale && shift && mean && rstd && partials, "hnd_bn_bwd_reduce: null pointer");
  HND_REQUIRE(npix > 0 && cs % 4 == 0 && cs <= 1024 && 256 % (cs / 4) == 0,
              "hnd_bn_bwd_reduce: channel stride %d must be 4*2^k <= 1024", cs);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(hnd_bn_bwd_ntiles(npix)), dim3(256), 0, hnd::as_stream(stream), g, x,
                     scale, shift, mean, rstd, relu, (long long)npix, cs, partials);
  return hnd::check_launch("hnd_bn_bwd_reduce");
}

int hnd_bn_bwd_finalize(const float* partials, int ntiles, int c, int cs, int64_t count, const float* gamma,
                        const float* mean, const float* rstd, float* dgamma, float* dbeta, float* k123, void* stream) {
  HND_REQUIRE(partials && gamma && mean && rstd && dgamma && dbeta && k123, "hnd_bn_bwd_finalize: null pointer");
  HND_REQUIRE(ntiles > 0 && c > 0 && cs >= c && count > 0, "hnd_bn_bwd_finalize: bad sizes");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((cs + 3) / 4), dim3(256), 0, hnd::as_stream(stream), partials, ntiles,
                     c, cs, (double)count, gamma, mean, rstd, dgamma, dbeta, k123);
  return hnd::check_launch("hnd_bn_bwd_finalize");
}

int hnd_bn_bwd_apply(const float* g, const float* x, const float* scale, const float* shift, const float* k123,
                     int relu, float* dx, int64_t npix, int cs, void* stream) {
  HND_REQUIRE(g && x && scale && shift && k123 && dx && npix > 0 && cs % 4 == 0, "hnd_bn_bwd_apply: bad arguments");
  const long long n4 = (long long)npix * (cs / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for_chunks(n4)), dim3(256), 0, hnd::as_stream(stream), g, x, scale,
                     shift, k123, relu, dx, n4, cs);
  return hnd::check_launch("hnd_bn_bwd_apply");
}

size_t hnd_mse_scratch_elems(void) { return kMseBlocks; }

int hnd_mse_sum_fwd_bwd(const hnd_mse_pair* pairs, int npairs, double* loss_out, double* scratch, void* stream) {
  HND_REQUIRE(pairs && loss_out && scratch, "hnd_mse_sum_fwd_bwd: null pointer");
  HND_REQUIRE(npairs >= 1 && npairs <= kMaxPairs, "hnd_mse_sum_fwd_bwd: 1..%d pairs supported", kMaxPairs);
  MseArgs a;
  a.npairs = npairs;
  long long total = 0;
  for (int k = 0; k < npairs; ++k) {
    HND_REQUIRE(pairs[k].teacher && pairs[k].student && pairs[k].numel > 0 && pairs[k].numel % 4 == 0,
                "hnd_mse_sum_fwd_bwd: pair %d invalid (numel must be a positive multiple of 4)", k);
    a.pair[k] = pairs[k];
    total += pairs[k].numel;
  }
  int used = 0;
  for (int k = 0; k < npairs; ++k) {
    a.first_block[k] = used;
    long long nb = (long long)(kMseBlocks - npairs) * pairs[k].numel / total + 1;
    const long long maxb = (pairs[k].numel / 4 + 256 * kEwU - 1) / (256 * kEwU);
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    used += (int)nb;
  }
  a.first_block[npairs] = used;
  hipStream_t s = hnd::as_stream(stream);
  hipLaunchKernelGGL(mse_kernel, dim3(used), dim3(256), 0, s, a, scratch);
  int rc = hnd::check_launch("hnd_mse_sum_fwd_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64 * kMaxPairs), 0, s, a, scratch, loss_out);
  return hnd::check_launch("hnd_mse_sum_fwd_bwd(finalize)");
}

int hnd_scale_by_device_scalar(float* x, int64_t numel, const float* scale_dev, void* stream) {
  HND_REQUIRE(x && scale_dev && numel > 0, "hnd_scale_by_device_scalar: bad arguments");
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(grid_for(numel)), dim3(256), 0, hnd::as_stream(stream), x,
                     (long long)numel, scale_dev);
  return hnd::check_launch("hnd_scale_by_device_scalar");
}

int hnd_adam_step_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, float lr,
                       float beta1, float beta2, float eps, int64_t step, float grad_scale, void* stream) {
  HND_REQUIRE(param && grad && exp_avg && exp_avg_sq && numel > 0 && step >= 1, "hnd_adam_step_flat: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(numel)), dim3(256), 0, hnd::as_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, (long long)numel, step_size, beta1, beta2, eps, bc2_sqrt, grad_scale);
  return hnd::check_launch("hnd_adam_step_flat");
}

int hnd_subsample2(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream) {
  HND_REQUIRE(x && y && c % 4 == 0 && oh == (h + 1) / 2 && ow == (w + 1) / 2, "hnd_subsample2: bad arguments");
  hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for((long long)n * oh * ow * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), x, y, n, h, w, c, oh, ow);
  return hnd::check_launch("hnd_subsample2");
}

int hnd_upsample_nearest_bwd(const float* g_fine, float* g_coarse, int n, int H, int W, int h, int w, int c, int accumulate,
                             void* stream) {
  HND_REQUIRE(g_fine && g_coarse && n > 0 && H >= h && W >= w && h > 0 && w > 0 && c > 0 && c % 4 == 0,
              "hnd_upsample_nearest_bwd: bad arguments (the fine map must not be smaller than the coarse one; c %% 4 == 0)");
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), g_fine, g_coarse, n, H, W, h, w, c, accumulate);
  return hnd::check_launch("hnd_upsample_nearest_bwd");
}

int hnd_add_inplace(float* x, const float* y, int64_t numel, void* stream) {
  HND_REQUIRE(x && y && numel > 0 && numel % 4 == 0, "hnd_add_inplace: bad arguments (numel %% 4 == 0)");
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, hnd::as_stream(stream), x, y,
                     (long long)(numel / 4));
  return hnd::check_launch("hnd_add_inplace");
}

size_t hnd_minmax_scratch_elems(void) { return 2 * kMinMaxBlocks; }

int hnd_quantize_u8(const float* x, int64_t npix, int c, int cs, uint8_t* q, float* qparams, float* scratch,
                    void* stream) {
  HND_REQUIRE(x && q && qparams && scratch && npix > 0 && c > 0 && cs >= c, "hnd_quantize_u8: bad arguments");
  hipStream_t s = hnd::as_stream(stream);
  int nb = grid_for((long long)npix * cs);
  if (nb > kMinMaxBlocks) nb = kMinMaxBlocks;
  hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, s, x, (long long)npix, c, cs, scratch);
  hipLaunchKernelGGL(qparams_kernel, dim3(1), dim3(64), 0, s, scratch, nb, 255.f, qparams);
  hipLaunchKernelGGL(quantize_kernel, dim3(grid_for((long long)npix * cs)), dim3(256), 0, s, x, (long long)npix, c, cs,
                     qparams, 255.f, q);
  return hnd::check_launch("hnd_quantize_u8");
}

int hnd_dequantize_u8(const uint8_t* q, const float* qparams, float* x, int64_t npix, int c, int cs, void* stream) {
  HND_REQUIRE(q && qparams && x && npix > 0 && c > 0 && cs >= c, "hnd_dequantize_u8: bad arguments");
  hipLaunchKernelGGL(dequantize_kernel, dim3(grid_for((long long)npix * cs)), dim3(256), 0, hnd::as_stream(stream), q,
                     qparams, x, (long long)npix, c, cs);
  return hnd::check_launch("hnd_dequantize_u8");
}

int hnd_roundtrip_f16(float* x, int64_t numel, void* stream) {
  HND_REQUIRE(x && numel > 0, "hnd_roundtrip_f16: bad arguments");
  hipLaunchKernelGGL(roundtrip_f16_kernel, dim3(grid_for(numel)), dim3(256), 0, hnd::as_stream(stream), x,
                     (long long)numel);
  return hnd::check_launch("hnd_roundtrip_f16");
}

int hnd_fill(float* x, int64_t numel, float value, void* stream) {
  HND_REQUIRE(x && numel > 0, "hnd_fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(numel)), dim3(256), 0, hnd::as_stream(stream), x, (long long)numel,
                     value);
  return hnd::check_launch("hnd_fill");
}

}  // extern "C"
