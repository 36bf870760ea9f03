// Neural-filter (Ext4ResNet) kernels, gfx950: the pieces of src/models/ext/classifier.py:16-37 that are not
// convolutions or BatchNorm (those reuse conv_igemm / conv_wgrad / the bn_* kernels), plus the SGD step of
// src/ext_runner.py:118-120.  All HBM-bound or tiny; NHWC fp32, channel stride cs (>= logical c).
#include "common.h"

using hnd::f32x4;

namespace {

constexpr int kMaxBlocks = 256 * 16;

inline int grid_for(long long work_items, int threads = 256) {
  long long b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// nn.AdaptiveAvgPool2d window of output o: [floor(o*in/out), ceil((o+1)*in/out))
__device__ __forceinline__ int win_lo(int o, int in, int out) { return (int)(((long long)o * in) / out); }
__device__ __forceinline__ int win_hi(int o, int in, int out) { return (int)(((long long)(o + 1) * in + out - 1) / out); }

// ---------------------------------------------------------------------------- adaptive average pool
// one thread per (n, oy, ox, 4 channels): the window rows are contiguous runs of (we-ws)*c floats
__global__ void adaptive_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w,
                                            int c, int oh, int ow) {
  const int c4n = c >> 2;
  const long long total = (long long)n * oh * ow * c4n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    long long p = e / c4n;
    const int ox = (int)(p % ow);
    p /= ow;
    const int oy = (int)(p % oh), b = (int)(p / oh);
    const int hs = win_lo(oy, h, oh), he = win_hi(oy, h, oh), ws = win_lo(ox, w, ow), we = win_hi(ox, w, ow);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = hs; iy < he; ++iy) {
      const float* row = x + (((size_t)b * h + iy) * w + ws) * c + c4 * 4;
      for (int ix = 0; ix < we - ws; ++ix) acc += *(const f32x4*)(row + (size_t)ix * c);
    }
    const float area = (float)((he - hs) * (we - ws));
    acc.x /= area; acc.y /= area; acc.z /= area; acc.w /= area;
    *(f32x4*)(y + (size_t)e * 4) = acc;
  }
}

// dx[iy][ix] = sum over the output windows containing (iy, ix) of dy / area  (gather form, deterministic)
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int h, int w,
                                            int c, int oh, int ow) {
  const int c4n = c >> 2;
  const long long total = (long long)n * h * w * c4n;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    long long p = e / c4n;
    const int ix = (int)(p % w);
    p /= w;
    const int iy = (int)(p % h), b = (int)(p / h);
    int oy0 = (int)(((long long)iy * oh) / h) - 1, oy1 = (int)(((long long)(iy + 1) * oh + h - 1) / h) + 1;
    int ox0 = (int)(((long long)ix * ow) / w) - 1, ox1 = (int)(((long long)(ix + 1) * ow + w - 1) / w) + 1;
    oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > oh ? oh : oy1; ox1 = ox1 > ow ? ow : ox1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy0; oy < oy1; ++oy) {
      const int hs = win_lo(oy, h, oh), he = win_hi(oy, h, oh);
      if (iy < hs || iy >= he) continue;
      for (int ox = ox0; ox < ox1; ++ox) {
        const int ws = win_lo(ox, w, ow), we = win_hi(ox, w, ow);
        if (ix < ws || ix >= we) continue;
        const float inv = 1.f / (float)((he - hs) * (we - ws));
        const f32x4 g = *(const f32x4*)(dy + (((size_t)b * oh + oy) * ow + ox) * c + c4 * 4);
        acc.x += g.x * inv; acc.y += g.y * inv; acc.z += g.z * inv; acc.w += g.w * inv;
      }
    }
    *(f32x4*)(dx + (size_t)e * 4) = acc;
  }
}

// ---------------------------------------------------------------------------- linear on flatten(1) of NCHW
// x is NHWC with channel stride cs; the weight is torch's [nout][c*hw] over the NCHW flatten (k = ch*hw + pix)
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                  const float* __restrict__ bias, float* __restrict__ out, int hw, int c, int cs,
                                  int nout) {
  __shared__ float red[256];
  const int b = blockIdx.x / nout, j = blockIdx.x % nout;
  const int k = c * hw;
  float acc = 0.f;
  for (int e = threadIdx.x; e < k; e += blockDim.x) {        // e walks NHWC order: coalesced activations
    const int pix = e / c, ch = e - pix * c;
    acc += x[((size_t)b * hw + pix) * cs + ch] * wgt[(size_t)j * k + (size_t)ch * hw + pix];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)b * nout + j] = red[0] + (bias ? bias[j] : 0.f);
}

// dW[j][k] = sum_b dout[b][j] * x_flat[b][k];  db[j] = sum_b dout[b][j]   (batch order fixed: deterministic)
__global__ void linear_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                         float* __restrict__ dw, float* __restrict__ db, int n, int hw, int c,
                                         int cs, int nout) {
  const int k = c * hw;
  const long long total = (long long)nout * k;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e / k), kk = (int)(e - (long long)j * k);
    const int ch = kk / hw, pix = kk - ch * hw;
    float acc = 0.f;
    for (int b = 0; b < n; ++b) acc += dout[(size_t)b * nout + j] * x[((size_t)b * hw + pix) * cs + ch];
    dw[e] = acc;
    if (kk == 0 && db) {
      float s = 0.f;
      for (int b = 0; b < n; ++b) s += dout[(size_t)b * nout + j];
      db[j] = s;
    }
  }
}

// dx[b][pix][ch] = sum_j dout[b][j] * W[j][ch*hw + pix]; pad channels (ch >= c) are written as 0
__global__ void linear_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ wgt,
                                        float* __restrict__ dx, int n, int hw, int c, int cs, int nout) {
  const int k = c * hw;
  const long long total = (long long)n * hw * cs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % cs);
    const long long p = e / cs;
    const int pix = (int)(p % hw), b = (int)(p / hw);
    float acc = 0.f;
    if (ch < c)
      for (int j = 0; j < nout; ++j) acc += dout[(size_t)b * nout + j] * wgt[(size_t)j * k + (size_t)ch * hw + pix];
    dx[e] = acc;
  }
}

// ---------------------------------------------------------------------------- softmax(dim=1), one thread per row
__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float m = -INFINITY;
  for (int j = 0; j < cols; ++j) m = fmaxf(m, x[(size_t)r * cols + j]);
  float s = 0.f;
  for (int j = 0; j < cols; ++j) s += expf(x[(size_t)r * cols + j] - m);
  for (int j = 0; j < cols; ++j) y[(size_t)r * cols + j] = expf(x[(size_t)r * cols + j] - m) / s;
}

// ---------------------------------------------------------------------------- mean cross entropy of [rows, cols] logits
// F.cross_entropy(logits, labels) (reference src/ext_runner.py:58) and its gradient in one launch of one workgroup:
// loss = mean over the counted rows of (logsumexp(x) - x[label]); dlogits = (softmax(x) - onehot(label)) / counted.
// Rows whose label is `ignore_index` (torch's default -100) contribute nothing.  Fixed summation order (row-strided
// partials, LDS tree): bitwise reproducible.
__global__ void softmax_ce_kernel(const float* __restrict__ x, const long long* __restrict__ labels, int rows, int cols,
                                  long long ignore_index, float* __restrict__ loss, float* __restrict__ dx) {
  __shared__ float red[256];
  __shared__ int cnt[256];
  float acc = 0.f;
  int n = 0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const long long lab = labels[r];
    if (lab == ignore_index || lab < 0 || lab >= cols) continue;
    const float* xr = x + (size_t)r * cols;
    float m = -INFINITY;
    for (int j = 0; j < cols; ++j) m = fmaxf(m, xr[j]);
    float s = 0.f;
    for (int j = 0; j < cols; ++j) s += expf(xr[j] - m);
    acc += (logf(s) + m) - xr[lab];
    ++n;
  }
  red[threadIdx.x] = acc;
  cnt[threadIdx.x] = n;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] += red[threadIdx.x + w];
      cnt[threadIdx.x] += cnt[threadIdx.x + w];
    }
    __syncthreads();
  }
  const int counted = cnt[0];
  const float inv = counted > 0 ? 1.f / (float)counted : 0.f;
  if (threadIdx.x == 0) loss[0] = counted > 0 ? red[0] * inv : NAN;      // torch: mean over zero rows is nan
  for (int r = threadIdx.x; r < rows; r += 256) {
    const long long lab = labels[r];
    const float* xr = x + (size_t)r * cols;
    float* dr = dx + (size_t)r * cols;
    if (lab == ignore_index || lab < 0 || lab >= cols) {
      for (int j = 0; j < cols; ++j) dr[j] = 0.f;
      continue;
    }
    float m = -INFINITY;
    for (int j = 0; j < cols; ++j) m = fmaxf(m, xr[j]);
    float s = 0.f;
    for (int j = 0; j < cols; ++j) s += expf(xr[j] - m);
    for (int j = 0; j < cols; ++j) dr[j] = (expf(xr[j] - m) / s - (j == (int)lab ? 1.f : 0.f)) * inv;
  }
}

// ---------------------------------------------------------------------------- per-channel sum (conv bias gradient)
// two passes, fixed summation order: grid (pixel chunk, 64-channel group) -> scratch[chunk][c]; then over chunks
constexpr int kSumChunks = 128;

__global__ void channel_sum_partial_kernel(const float* __restrict__ x, float* __restrict__ scratch, long long npix,
                                           int c, int cs, int nchunks) {
  __shared__ float red[256];
  const int lane = threadIdx.x & 63, stripe = threadIdx.x >> 6;
  const int chunk = blockIdx.x, ch = blockIdx.y * 64 + lane;
  const long long per = (npix + nchunks - 1) / nchunks;
  const long long p0 = chunk * per, p1 = (p0 + per < npix) ? p0 + per : npix;
  float acc = 0.f;
  if (ch < c)
    for (long long p = p0 + stripe; p < p1; p += 4) acc += x[(size_t)p * cs + ch];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (stripe == 0 && ch < c)
    scratch[(size_t)chunk * c + ch] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
}

__global__ void channel_sum_final_kernel(const float* __restrict__ scratch, float* __restrict__ out, int c, int nchunks) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float acc = 0.f;
  for (int k = 0; k < nchunks; ++k) acc += scratch[(size_t)k * c + ch];
  out[ch] = acc;
}

// ---------------------------------------------------------------------------- torch.optim.SGD (momentum, weight decay)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                           float lr, float momentum, float dampening, float weight_decay, float grad_scale,
                           int first, int nesterov) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float w = p[i];
    float d = g[i] * grad_scale;
    if (weight_decay != 0.f) d = d + weight_decay * w;          // d_p = d_p.add(p, alpha=weight_decay)
    if (momentum != 0.f) {
      float b = first ? d : momentum * buf[i] + (1.f - dampening) * d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] = w - lr * d;
  }
}

}  // namespace

extern "C" {

int hnd_adaptive_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream) {
  HND_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && c > 0 && c % 4 == 0,
              "hnd_adaptive_avgpool_fwd: bad arguments (c=%d must be a multiple of 4)", c);
  hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel, dim3(grid_for((long long)n * oh * ow * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), x, y, n, h, w, c, oh, ow);
  return hnd::check_launch("hnd_adaptive_avgpool_fwd");
}

int hnd_adaptive_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int oh, int ow, void* stream) {
  HND_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && c > 0 && c % 4 == 0,
              "hnd_adaptive_avgpool_bwd: bad arguments (c=%d must be a multiple of 4)", c);
  hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 4))), dim3(256), 0,
                     hnd::as_stream(stream), dy, dx, n, h, w, c, oh, ow);
  return hnd::check_launch("hnd_adaptive_avgpool_bwd");
}

int hnd_linear_fwd(const float* x, const float* weight, const float* bias, float* out, int n, int hw, int c, int cs,
                   int nout, void* stream) {
  HND_REQUIRE(x && weight && out && n > 0 && hw > 0 && c > 0 && cs >= c && nout > 0, "hnd_linear_fwd: bad arguments");
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(n * nout), dim3(256), 0, hnd::as_stream(stream), x, weight, bias, out, hw,
                     c, cs, nout);
  return hnd::check_launch("hnd_linear_fwd");
}

int hnd_linear_bwd(const float* x, const float* weight, const float* dout, float* dweight, float* dbias, float* dx,
                   int n, int hw, int c, int cs, int nout, void* stream) {
  HND_REQUIRE(x && weight && dout && n > 0 && hw > 0 && c > 0 && cs >= c && nout > 0, "hnd_linear_bwd: bad arguments");
  if (dweight) {
    hipLaunchKernelGGL(linear_bwd_params_kernel, dim3(grid_for((long long)nout * c * hw)), dim3(256), 0,
                       hnd::as_stream(stream), x, dout, dweight, dbias, n, hw, c, cs, nout);
    const int rc = hnd::check_launch("hnd_linear_bwd(params)");
    if (rc) return rc;
  }
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_input_kernel, dim3(grid_for((long long)n * hw * cs)), dim3(256), 0,
                       hnd::as_stream(stream), dout, weight, dx, n, hw, c, cs, nout);
    return hnd::check_launch("hnd_linear_bwd(input)");
  }
  return HND_OK;
}

int hnd_softmax_rows(const float* x, float* y, int rows, int cols, void* stream) {
  HND_REQUIRE(x && y && rows > 0 && cols > 0, "hnd_softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, hnd::as_stream(stream), x, y, rows,
                     cols);
  return hnd::check_launch("hnd_softmax_rows");
}

int hnd_softmax_ce_rows_fwd_bwd(const float* logits, const int64_t* labels, int rows, int cols, int64_t ignore_index,
                                float* loss, float* dlogits, void* stream) {
  HND_REQUIRE(logits && labels && loss && dlogits && rows > 0 && cols > 0, "hnd_softmax_ce_rows_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, hnd::as_stream(stream), logits,
                     (const long long*)labels, rows, cols, (long long)ignore_index, loss, dlogits);
  return hnd::check_launch("hnd_softmax_ce_rows_fwd_bwd");
}

size_t hnd_channel_sum_scratch_elems(int c) { return (size_t)kSumChunks * (size_t)(c > 0 ? c : 0); }

int hnd_channel_sum(const float* x, float* out, int64_t npix, int c, int cs, float* scratch, void* stream) {
  HND_REQUIRE(x && out && scratch && npix > 0 && c > 0 && cs >= c, "hnd_channel_sum: bad arguments");
  int nchunks = (int)((npix + 255) / 256);
  if (nchunks > kSumChunks) nchunks = kSumChunks;
  hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(nchunks, (c + 63) / 64), dim3(256), 0, hnd::as_stream(stream), x,
                     scratch, (long long)npix, c, cs, nchunks);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 63) / 64), dim3(64), 0, hnd::as_stream(stream), scratch, out,
                     c, nchunks);
  return hnd::check_launch("hnd_channel_sum");
}

int hnd_sgd_step_flat(float* param, const float* grad, float* momentum_buf, int64_t numel, float lr, float momentum,
                      float dampening, float weight_decay, int nesterov, int first_step, float grad_scale,
                      void* stream) {
  HND_REQUIRE(param && grad && numel > 0 && (momentum == 0.f || momentum_buf), "hnd_sgd_step_flat: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(numel)), dim3(256), 0, hnd::as_stream(stream), param, grad,
                     momentum_buf, (long long)numel, lr, momentum, dampening, weight_decay, grad_scale, first_step,
                     nesterov);
  return hnd::check_launch("hnd_sgd_step_flat");
}

}  // extern "C"
