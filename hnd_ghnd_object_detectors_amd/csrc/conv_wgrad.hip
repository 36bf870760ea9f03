// Weight gradient of the trainable convs as a split-K implicit GEMM on fp32 MFMA (gfx950).
//
//   dW[co][col] = sum_m dy[m][co] * a(m; col),   col = (tap, ci),  m = n*oh*ow output pixels
//
// The reduction runs over pixels (1.1M..4.3M at batch 16), so both operands are staged exactly as
// they lie in HBM -- [pixel][channel], channel contiguous -- and the MFMA fragments are read from
// LDS with ds_read_b32 (lane = channel).  v_mfma_f32_16x16x4_f32: lane l supplies channel l%16 of pixel l/16, so a
// fragment read touches 4 pixel rows x 16 channels; the LDS row stride is padded to 16 mod 32 floats, which puts the
// two pixel rows of each 32-lane read group on different bank halves (conflict-free).  16x16x4 rather than 32x32x2
// because this kernel runs 3-4 waves per SIMD, where the 32x32x2 shape loses a third of the matrix rate
// (tools/probes/mfma_f32_probe.hip).
// Block tile: BMW (64|128 output channels) x 128 (cols) x 16 (pixels); 4 waves 2x2.
// Each block reduces one contiguous pixel range (split-K); partial tiles go to slabs
// [split][co_pad][ncols_pad] and a second kernel sums the slabs in fixed order and writes the
// torch OIHW layout -> bitwise reproducible, no float atomics.
#include "common.h"

#include <math.h>
#include <stdlib.h>

namespace {

using hnd::FastDiv;
using hnd::f32x4;
using hnd::fdiv;

constexpr int BNW = 128;

struct WgradArgs {
  hnd_wgrad_desc d;
  FastDiv div_ow, div_oh, div_cin;
  int ncols, ncols_pad, co_pad, rtiles, ctiles, steps_per_split, M;
};

constexpr int lds_stride(int n) { return n + 16; }        // == 16 (mod 32) floats for n = 64, 128

template <int BMW, int BKW>
constexpr size_t wgrad_lds_bytes() { return (size_t)2 * BKW * (lds_stride(BMW) + lds_stride(BNW)) * sizeof(float); }

template <int BMW, int BKW>
__global__ void __launch_bounds__(256, (BKW == 16 ? 3 : 2)) wgrad_kernel(const WgradArgs a) {
  constexpr int LDA = lds_stride(BMW), LDB = lds_stride(BNW);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BKW][LDA]   dy tile
  float* Bs = smem + 2 * BKW * LDA;       // [2][BKW][LDB]   gathered activation tile
  const hnd_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // batched launch: blockIdx.y selects one of `groups` independent problems (operands / slabs a fixed stride apart)
  const int grp = blockIdx.y;
  const float* __restrict__ xg = d.x + (size_t)grp * (size_t)d.x_group_stride;
  const float* __restrict__ dyg = d.dy + (size_t)grp * (size_t)d.dy_group_stride;
  int bid = blockIdx.x;
  const int tiles = a.rtiles * a.ctiles;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int rt = bid / a.ctiles, ct = bid - rt * a.ctiles;
  const int r0 = rt * BMW, c0 = ct * BNW;
  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M + BKW - 1) / BKW - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;

  // A (dy) loader: thread -> 4 channels (float4) of rows (tid / (BMW/4)) + i*(256/(BMW/4))
  constexpr int A_TPR = BMW / 4;            // threads per row
  constexpr int A_RPI = 256 / A_TPR;        // rows per pass
  constexpr int A_N = BKW / A_RPI;          // passes
  const int a_c4 = (tid % A_TPR) * 4, a_r = tid / A_TPR;
  const bool a_cok = (r0 + a_c4) < d.cout;  // cout is a multiple of 4 on this path (or masked below)
  // B (activation) loader: thread -> fixed column group (4 consecutive ci of one tap)
  const int b_c4 = (tid & 31) * 4, b_r = tid >> 5;   // 8 rows per pass, 4 passes
  const int col = c0 + b_c4;
  const bool b_cok = col < a.ncols;
  const int tap = b_cok ? (int)fdiv((unsigned)col, a.div_cin) : 0;
  const int ci = col - tap * d.cin;
  const int ti = tap / d.kw, tj = tap - ti * d.kw;
  const bool has_pro = d.pro_scale != nullptr;
  f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
  if (has_pro && b_cok) {
    ps = *(const f32x4*)(d.pro_scale + ci);
    if (d.pro_shift) pb = *(const f32x4*)(d.pro_shift + ci);
  }

  constexpr int B_N = BKW / 8;             // passes of 8 pixel rows for the activation tile
  f32x4 ra[A_N], rb[B_N];
  unsigned bok = 0;
  const bool a_vec = a_cok && (r0 + a_c4 + 3 < d.cout);
  // Branch-free loads (clamped addresses, validity applied by selects in lstore) so the global loads of tile
  // s+1 stay in flight behind the MFMAs of tile s instead of being waited for inside divergent branches.
  auto gload = [&](int s) {
    const int mbase = (step0 + s) * BKW;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int m = mbase + a_r + i * A_RPI;
      const bool ok = m < a.M && a_cok;
      const float* p = dyg + (size_t)(ok ? m : 0) * d.ldy + (ok ? r0 + a_c4 : 0);
      f32x4 v;
      if (a_vec) {                       // thread-constant: the whole float4 lies inside [0, cout)
        v = *(const f32x4*)p;
      } else {                           // channel tail (cout = 3 stored as 4): scalar loads of the valid part
        v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
          v.x = p[0];
          if (r0 + a_c4 + 1 < d.cout) v.y = p[1];
          if (r0 + a_c4 + 2 < d.cout) v.z = p[2];
        }
      }
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      ra[i] = v;
    }
    bok = 0;
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int m = mbase + b_r + i * 8;
      const int mm = m < a.M ? m : 0;
      const unsigned t = fdiv((unsigned)mm, a.div_ow);
      const int ow_ = mm - (int)t * d.ow;
      const unsigned n_ = fdiv(t, a.div_oh);
      const int oh_ = (int)t - (int)n_ * d.oh;
      const int ih = oh_ * d.stride - d.pad + ti, iw = ow_ * d.stride - d.pad + tj;
      const bool ok = m < a.M && b_cok && (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
      const size_t off = ok ? ((size_t)((int)n_ * d.h + ih) * d.w_ + iw) * d.cin + ci : 0;
      rb[i] = *(const f32x4*)(xg + off);
      bok |= (ok ? 1u : 0u) << i;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      *(f32x4*)(As + (buf * BKW + a_r + i * A_RPI) * LDA + a_c4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      f32x4 v = rb[i];
      if (has_pro) {
        v = v * ps + pb;
        if (d.pro_relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      const bool ok = (bok >> i) & 1;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *(f32x4*)(Bs + (buf * BKW + b_r + i * 8) * LDB + b_c4) = v;
    }
  };

  constexpr int MI = BMW / 32;   // 16-row MFMA tiles per wave along co (wave tile BMW/2 x 64)
  constexpr int NI = 4;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l16 = lane & 15, g4 = lane >> 4;
  auto compute = [&](int buf) {
    const float* Ap = As + buf * BKW * LDA + g4 * LDA + wm * (BMW / 2) + l16;
    const float* Bp = Bs + buf * BKW * LDB + g4 * LDB + wn * 64 + l16;
#pragma unroll
    for (int kk = 0; kk < BKW / 4; ++kk) {
      float av[MI], bv[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[mi] = Ap[kk * 4 * LDA + mi * 16];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = Bp[kk * 4 * LDB + ni * 16];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  if (nsteps > 0) {
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool more = (s + 1) < nsteps;
      if (more) gload(s + 1);
      compute(cur);
      if (more) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  // partial tile -> slab[split][co][col]; D[i=co][j=col]: lane%16 -> col, lane/16 and regs -> co
  float* slab = d.slabs + ((size_t)grp * d.splitk + split) * a.co_pad * a.ncols_pad;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = r0 + wm * (BMW / 2) + mi * 16 + 4 * g4 + r;
        const int cc = c0 + wn * 64 + ni * 16 + l16;
        slab[(size_t)co * a.ncols_pad + cc] = acc[mi][ni][r];
      }
}

// sum the slabs and write dW in torch OIHW layout.  Fixed order (bit-reproducible): wave q of the block sums the
// slabs k = q, q+4, q+8, ... in increasing k, then the four partial sums are added in the order ((0+1)+2)+3.  Four
// k-lanes per output keep 4x more loads in flight than one serial chain per output (up to 512 slabs deep).
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int splitk, int co_pad,
                                    int ncols_pad, int cout, int cin, int cin_real, int kh, int kw,
                                    long long dw_group_stride) {
  __shared__ float red[4][64];
  slabs += (size_t)blockIdx.y * splitk * co_pad * ncols_pad;
  dw += (size_t)blockIdx.y * (size_t)dw_group_stride;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;   // over cout * kh*kw * cin_real in (co, tap, ci) order
  const int per_co = kh * kw * cin_real;
  const bool ok = idx < cout * per_co;
  const int co = ok ? idx / per_co : 0, rem = ok ? idx - co * per_co : 0;
  const int tap = rem / cin_real, ci = rem - tap * cin_real;
  const size_t off = (size_t)co * ncols_pad + tap * cin + ci;
  const size_t stride = (size_t)co_pad * ncols_pad;
  float s = 0.f;
  if (ok)
    {
      int k = q;
      for (; k + 12 < splitk; k += 16) {        // four loads in flight, added in the original order
        const float a0 = slabs[off + k * stride], a1 = slabs[off + (k + 4) * stride];
        const float a2 = slabs[off + (k + 8) * stride], a3 = slabs[off + (k + 12) * stride];
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; k < splitk; k += 4) s += slabs[off + k * stride];
    }
  red[q][lane] = s;
  __syncthreads();
  if (q == 0 && ok) {
    const int i = tap / kw, j = tap - i * kw;
    dw[(((size_t)co * cin_real + ci) * kh + i) * kw + j] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The two 3-channel weight gradients of the b3ch bottleneck (layer1.conv3: 2x2x64 -> 3, layer1.conv4: 2x2x3 -> 64):
//   dW[co][ci][i][j] = sum_m dy[m][co] * a(m; i, j, ci)
// is the outer product of a 64-channel tensor and a 4-channel one over four taps, reduced over 1.1 M pixels.  On the
// MFMA tile above 60 of 64 rows (conv3) or 112 of 128 columns (conv4) are padding: 0.44 / 0.30 ms for 1.7 GFLOP.  Here
// the THICK tensor (conv3: the input x, conv4: dy) streams through once -- 16 lanes x float4 per pixel, four pixels per
// wave instruction -- the thin one (conv3: dy, conv4: x, 16 bytes per pixel) is fetched per tap from the cache, and
// every lane keeps its 4 channels x 4 taps x 4 thin channels = 64 sums in registers: HBM-bound (275 MB).
// Reduction order is fixed: a lane's pixels in ascending order, then pixel slots (xor 16, xor 32), waves 0..3, and the
// per-block partials in wgrad_reduce_kernel's order -> bitwise reproducible.
// THICK_IS_X: conv3 form (thick = x with the BN/ReLU prologue, thin = dy); else conv4 form (thick = dy, thin = x).
struct ThinWgradArgs {
  hnd_wgrad_desc d;
  FastDiv div_w, div_h;       // thick pixel -> (n, y, x)
  int th, tw;                 // thick map extent
  int nh, nw;                 // thin map extent
  long long P;                // thick pixels
  int per_block;              // thick pixels per block (multiple of 64)
};

template <bool THICK_IS_X>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(const ThinWgradArgs a) {
  __shared__ float red[4][1024];
  const hnd_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = lane & 15, sub = lane >> 4;          // channels 4*c4 .. 4*c4+3 of pixel slot `sub`
  const float* __restrict__ thick = THICK_IS_X ? d.x : d.dy;
  const float* __restrict__ thin = THICK_IS_X ? d.dy : d.x;
  f32x4 tps = {1.f, 1.f, 1.f, 1.f}, tpb = {0.f, 0.f, 0.f, 0.f};     // prologue of the tensor x (thick or thin)
  const bool has_pro = d.pro_scale != nullptr;
  if (has_pro) {
    tps = *(const f32x4*)(d.pro_scale + (THICK_IS_X ? 4 * c4 : 0));
    if (d.pro_shift) tpb = *(const f32x4*)(d.pro_shift + (THICK_IS_X ? 4 * c4 : 0));
  }
  const float floor_ = d.pro_relu ? 0.f : -INFINITY;
  auto pro = [&](f32x4 v) {
    if (has_pro) {
      v = v * tps + tpb;
      v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
    }
    return v;
  };
  f32x4 acc[4][4];                                    // [tap][thin channel] x (4 thick channels)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long p0 = (long long)blockIdx.x * a.per_block;
  long long p1 = p0 + a.per_block;
  if (p1 > a.P) p1 = a.P;
  const int sgn = THICK_IS_X ? -1 : 1;                // thin pixel = thick pixel + sgn * (tap - pad)
  // 64 pixels per block and iteration: every wave requests four 1 KB rows of the thick tensor and their 16 thin
  // pixels before the first fma (one row per iteration ran at the memory latency: 1 TB/s)
  constexpr int UN = 4;
  for (long long pb = p0; pb < p1; pb += 16 * UN) {
    f32x4 v[UN], q[UN][4];
    unsigned okm = 0;                                 // bit 4u + t: thin pixel of tap t exists
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long p = pb + (u * 4 + wave) * 4 + sub;
      const bool pok = p < p1;
      const unsigned pp = (unsigned)(pok ? p : p0);
      const unsigned t_ = hnd::fdiv(pp, a.div_w), px = pp - t_ * (unsigned)a.tw;
      const unsigned n_ = hnd::fdiv(t_, a.div_h), py = t_ - n_ * (unsigned)a.th;
      v[u] = *(const f32x4*)(thick + (size_t)pp * 64 + 4 * c4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ty = (int)py + sgn * ((t >> 1) - d.pad), tx = (int)px + sgn * ((t & 1) - d.pad);
        const bool ok = pok && (unsigned)ty < (unsigned)a.nh && (unsigned)tx < (unsigned)a.nw;
        const size_t off = ok ? ((size_t)((int)n_ * a.nh + ty) * a.nw + tx) * 4 : 0;
        q[u][t] = *(const f32x4*)(thin + off);
        okm |= (ok ? 1u : 0u) << (4 * u + t);
      }
    }
    // (nothing above consumes a loaded value: all 20 loads are in flight before the first fma)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const f32x4 vv = THICK_IS_X ? pro(v[u]) : v[u];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 w = THICK_IS_X ? q[u][t] : pro(q[u][t]);
        if (!((okm >> (4 * u + t)) & 1)) w = f32x4{0.f, 0.f, 0.f, 0.f};      // padding: a zero of the NORMALISED tensor
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] += vv * w[c];
      }
    }
  }
  // pixel slots of the wave, then the four waves, in fixed order
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = acc[t][c][k];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        acc[t][c][k] = x;
      }
  if (sub == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][(t * 4 + c) * 64 + 4 * c4 + k] = acc[t][c][k];
  }
  __syncthreads();
  // slab[block][co][col], col = tap * cin + ci (wgrad_reduce_kernel's layout): conv3 form co = thin channel (co_pad 4,
  // 256 cols), conv4 form co = thick channel (co_pad 64, 16 cols)
  float* slab = d.slabs + (size_t)blockIdx.x * 1024;
  for (int e = tid; e < 1024; e += 256) {
    const int tc = e >> 6, ch = e & 63, t = tc >> 2, c = tc & 3;
    const float x = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    if (THICK_IS_X) slab[c * 256 + t * 64 + ch] = x;
    else slab[ch * 16 + t * 4 + c] = x;
  }
}

bool thin_wgrad_applies(const hnd_wgrad_desc& d, bool& thick_is_x) {
  const char* e = getenv("HND_THIN_WGRAD");           // 0 = the MFMA kernel (read per call: in-process A/B)
  if ((e && atoi(e) == 0) || d.kh != 2 || d.kw != 2 || d.stride != 1 || d.groups > 1 || (d.pad != 0 && d.pad != 1)) return false;
  if (d.oh != d.h + 2 * d.pad - 1 || d.ow != d.w_ + 2 * d.pad - 1) return false;
  if (d.cin == 64 && d.cin_real == 64 && d.cout <= 4 && d.ldy == 4) { thick_is_x = true; return true; }
  if (d.cin == 4 && d.cout == 64 && d.ldy == 64) { thick_is_x = false; return true; }
  return false;
}

int thin_wgrad_blocks(const hnd_wgrad_desc& d, bool thick_is_x, int& per_block) {
  const long long P = thick_is_x ? (long long)d.n * d.h * d.w_ : (long long)d.n * d.oh * d.ow;
  long long blocks = (P + 511) / 512;                 // >= 512 pixels per block ...
  const long long cap = 512;
  if (blocks > cap) blocks = cap;                     // ... and at most 512 partial slabs of 4 KB (measured: 1024 -> 0.156 ms, 512 -> 0.118, 256 -> 0.130)
  if (blocks < 1) blocks = 1;
  per_block = (int)(((P + blocks - 1) / blocks + 63) / 64 * 64);
  return (int)((P + per_block - 1) / per_block);
}

// k-step depth: 16 (32 KB LDS, 4 resident blocks per CU; 32 and caps of 2 / 3 blocks per CU measured neutral in rounds 1-2)
constexpr int wgrad_bk() { return 16; }

int plan(const hnd_wgrad_desc& d, WgradArgs& a, int BKW) {
  a.d = d;
  a.M = d.n * d.oh * d.ow;
  a.ncols = d.kh * d.kw * d.cin;
  a.ctiles = (a.ncols + BNW - 1) / BNW;
  a.ncols_pad = a.ctiles * BNW;
  const int bmw = d.cout >= 128 ? 128 : 64;
  a.rtiles = (d.cout + bmw - 1) / bmw;
  a.co_pad = a.rtiles * bmw;
  a.div_ow = hnd::make_fastdiv((unsigned)d.ow);
  a.div_oh = hnd::make_fastdiv((unsigned)d.oh);
  a.div_cin = hnd::make_fastdiv((unsigned)d.cin);
  const int total_steps = (a.M + BKW - 1) / BKW;
  int splitk = d.splitk;
  if (splitk <= 0) {
    const int tiles = a.rtiles * a.ctiles * (d.groups > 1 ? d.groups : 1);
    splitk = (1024 + tiles - 1) / tiles;          // ~4 blocks per CU
    const int max_by_work = (total_steps + 15) / 16;   // at least 16 k-steps per split
    if (splitk > max_by_work) splitk = max_by_work;
    if (splitk < 1) splitk = 1;
    if (splitk > 512) splitk = 512;
  }
  a.steps_per_split = (total_steps + splitk - 1) / splitk;
  a.d.splitk = (total_steps + a.steps_per_split - 1) / a.steps_per_split;   // drop empty splits
  return bmw;
}

}  // namespace

namespace hnd {
bool stem7_wgrad_applies(const hnd_wgrad_desc& d);             // conv_stem.hip: the stem's dW from an LDS patch
int stem7_wgrad_blocks(const hnd_wgrad_desc& d);
int launch_stem7_wgrad(const hnd_wgrad_desc& d, int ncols_pad, hipStream_t stream);
bool wgrad_ring_applies(const hnd_wgrad_desc& d);              // conv_wgrad_ring.hip: operands straight from global
size_t wgrad_ring_workspace(const hnd_wgrad_desc& d);
int launch_wgrad_ring(const hnd_wgrad_desc& d, int& splits, int& co_pad, int& ncols_pad, hipStream_t stream);
}  // namespace hnd

extern "C" size_t hnd_conv2d_wgrad_workspace(const hnd_wgrad_desc* desc) {
  if (!desc) return 0;
  WgradArgs a;
  hnd_wgrad_desc d = *desc;
  if (d.cin <= 0 || d.cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.oh <= 0 || d.ow <= 0 || d.n <= 0) return 0;
  plan(d, a, wgrad_bk());
  size_t need = (size_t)(d.groups > 1 ? d.groups : 1) * a.d.splitk * a.co_pad * a.ncols_pad * sizeof(float);
  if (hnd::stem7_wgrad_applies(d)) {                    // one [64][ncols_pad] slab per persistent workgroup
    const size_t stem = (size_t)hnd::stem7_wgrad_blocks(d) * 64 * a.ncols_pad * sizeof(float);
    if (stem > need) need = stem;
  }
  if (hnd::wgrad_ring_applies(d)) {                     // one [cout][cols] slab per workgroup of the ring kernel
    const size_t ring = hnd::wgrad_ring_workspace(d);
    if (ring > need) need = ring;
  }
  bool tx;
  if (thin_wgrad_applies(d, tx)) {                      // one 4 KB partial per block
    int per_block;
    const size_t thin = (size_t)thin_wgrad_blocks(d, tx, per_block) * 1024 * sizeof(float);
    if (thin > need) need = thin;
  }
  return need;
}

extern "C" int hnd_conv2d_wgrad_variant(const hnd_wgrad_desc* desc) {
  if (!desc) return -1;
  const hnd_wgrad_desc& d = *desc;
  bool tx;
  if (hnd::stem7_wgrad_applies(d)) return 1;
  if (thin_wgrad_applies(d, tx)) return 2;
  if (hnd::wgrad_ring_applies(d)) return 3;
  return 0;
}

extern "C" int hnd_conv2d_wgrad(const hnd_wgrad_desc* desc, void* stream) {
  HND_REQUIRE(desc != nullptr, "hnd_conv2d_wgrad: null descriptor");
  const hnd_wgrad_desc& d = *desc;
  HND_REQUIRE(d.x && d.dy && d.dw && d.slabs, "hnd_conv2d_wgrad: null x/dy/dw/slabs");
  HND_REQUIRE(d.n > 0 && d.h > 0 && d.w_ > 0 && d.oh > 0 && d.ow > 0 && d.cout > 0 && d.kh > 0 && d.kw > 0 &&
                  d.stride > 0,
              "hnd_conv2d_wgrad: non-positive geometry");
  HND_REQUIRE(d.cin == 4 || d.cin % 32 == 0, "hnd_conv2d_wgrad: cin=%d must be 4 or a multiple of 32", d.cin);
  HND_REQUIRE(d.cin_real > 0 && d.cin_real <= d.cin, "hnd_conv2d_wgrad: bad cin_real");
  HND_REQUIRE(d.ldy >= d.cout && d.ldy % 4 == 0, "hnd_conv2d_wgrad: ldy=%d must be a multiple of 4 >= cout", d.ldy);
  HND_REQUIRE((long long)d.n * d.oh * d.ow < (1ll << 31) && (long long)d.n * d.h * d.w_ < (1ll << 31),
              "hnd_conv2d_wgrad: pixel count exceeds int32");
  WgradArgs a;
  const int bk = wgrad_bk();
  const int bmw = plan(d, a, bk);
  hipStream_t s = hnd::as_stream(stream);
  const int groups = d.groups > 1 ? d.groups : 1;
  HND_REQUIRE(groups == 1 || (d.x_group_stride > 0 && d.dy_group_stride > 0 && d.dw_group_stride > 0),
              "hnd_conv2d_wgrad: group strides must be positive when groups > 1");
  if (hnd::stem7_wgrad_applies(d)) {
    int rc = hnd::launch_stem7_wgrad(d, a.ncols_pad, s);
    if (rc) return rc;
    const int total = d.cout * d.kh * d.kw * d.cin_real;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, 1), dim3(256), 0, s, d.slabs, d.dw,
                       hnd::stem7_wgrad_blocks(d), 64, a.ncols_pad, d.cout, d.cin, d.cin_real, d.kh, d.kw, 0ll);
    return hnd::check_launch("hnd_conv2d_wgrad(stem7 reduce)");
  }
  bool thick_is_x;
  if (thin_wgrad_applies(d, thick_is_x)) {
    ThinWgradArgs t;
    t.d = d;
    t.th = thick_is_x ? d.h : d.oh; t.tw = thick_is_x ? d.w_ : d.ow;
    t.nh = thick_is_x ? d.oh : d.h; t.nw = thick_is_x ? d.ow : d.w_;
    t.div_w = hnd::make_fastdiv((unsigned)t.tw);
    t.div_h = hnd::make_fastdiv((unsigned)t.th);
    t.P = (long long)d.n * t.th * t.tw;
    const int blocks = thin_wgrad_blocks(d, thick_is_x, t.per_block);
    if (thick_is_x) hipLaunchKernelGGL(thin_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(thin_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, s, t);
    int rc = hnd::check_launch("hnd_conv2d_wgrad(thin)");
    if (rc) return rc;
    const int total = d.cout * d.kh * d.kw * d.cin_real;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, 1), dim3(256), 0, s, d.slabs, d.dw, blocks,
                       thick_is_x ? 4 : 64, thick_is_x ? 256 : 16, d.cout, d.cin, d.cin_real, d.kh, d.kw, 0ll);
    return hnd::check_launch("hnd_conv2d_wgrad(thin reduce)");
  }
  if (hnd::wgrad_ring_applies(d)) {
    int splits = 0, co_pad = 0, ncols_pad = 0;
    int rc = hnd::launch_wgrad_ring(d, splits, co_pad, ncols_pad, s);
    if (rc) return rc;
    const int total = d.cout * d.kh * d.kw * d.cin_real;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, groups), dim3(256), 0, s, d.slabs, d.dw, splits,
                       co_pad, ncols_pad, d.cout, d.cin, d.cin_real, d.kh, d.kw, (long long)d.dw_group_stride);
    return hnd::check_launch("hnd_conv2d_wgrad(ring reduce)");
  }
  const dim3 grid(a.rtiles * a.ctiles * a.d.splitk, groups);
  constexpr size_t lds128 = wgrad_lds_bytes<128, 16>(), lds64 = wgrad_lds_bytes<64, 16>();
  if (bmw == 128) hipLaunchKernelGGL((wgrad_kernel<128, 16>), grid, dim3(256), lds128, s, a);
  else hipLaunchKernelGGL((wgrad_kernel<64, 16>), grid, dim3(256), lds64, s, a);
  int rc = hnd::check_launch("hnd_conv2d_wgrad");
  if (rc) return rc;
  const int total = d.cout * d.kh * d.kw * d.cin_real;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, groups), dim3(256), 0, s, d.slabs, d.dw,
                     a.d.splitk, a.co_pad, a.ncols_pad, d.cout, d.cin, d.cin_real, d.kh, d.kw,
                     (long long)d.dw_group_stride);
  return hnd::check_launch("hnd_conv2d_wgrad(reduce)");
}
