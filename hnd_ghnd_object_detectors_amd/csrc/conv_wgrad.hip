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
    for (int k = q; k < splitk; k += 4) s += slabs[off + k * stride];
  red[q][lane] = s;
  __syncthreads();
  if (q == 0 && ok) {
    const int i = tap / kw, j = tap - i * kw;
    dw[(((size_t)co * cin_real + ci) * kh + i) * kw + j] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  }
}

// k-step depth of the build in use: 16 (32 KB LDS, 4 resident blocks per CU) unless HND_WGRAD_BK=32
int wgrad_bk() {
  static const int bk = (getenv("HND_WGRAD_BK") && atoi(getenv("HND_WGRAD_BK")) == 32) ? 32 : 16;
  return bk;
}

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
  return need;
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
  const dim3 grid(a.rtiles * a.ctiles * a.d.splitk, groups);
  // tuning knob HND_WGRAD_BPC: cap the resident blocks per CU by padding the dynamic LDS request.  Measured in the
  // step (batch 16): 4 blocks/CU (the register / LDS limit, default) 5.95 ms, 3 -> 6.34 ms, 2 -> 5.97 ms for the
  // cout >= 128 launches -- this kernel is not bound by matrix-pipe contention, so the uncapped occupancy stays
  static const int bpc = getenv("HND_WGRAD_BPC") ? atoi(getenv("HND_WGRAD_BPC")) : 0;
  const size_t floor_lds = bpc > 0 ? (size_t)(160 * 1024) / (size_t)bpc - 2048 : 0;
  auto lds_of = [&](size_t need) { return need > floor_lds || floor_lds > 65536 ? need : floor_lds; };
  if (bk == 16) {
    if (bmw == 128)
      hipLaunchKernelGGL((wgrad_kernel<128, 16>), grid, dim3(256), lds_of(wgrad_lds_bytes<128, 16>()), s, a);
    else
      hipLaunchKernelGGL((wgrad_kernel<64, 16>), grid, dim3(256), lds_of(wgrad_lds_bytes<64, 16>()), s, a);
  } else {
    if (bmw == 128)
      hipLaunchKernelGGL((wgrad_kernel<128, 32>), grid, dim3(256), lds_of(wgrad_lds_bytes<128, 32>()), s, a);
    else
      hipLaunchKernelGGL((wgrad_kernel<64, 32>), grid, dim3(256), lds_of(wgrad_lds_bytes<64, 32>()), s, a);
  }
  int rc = hnd::check_launch("hnd_conv2d_wgrad");
  if (rc) return rc;
  const int total = d.cout * d.kh * d.kw * d.cin_real;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64, groups), dim3(256), 0, s, d.slabs, d.dw,
                     a.d.splitk, a.co_pad, a.ncols_pad, d.cout, d.cin, d.cin_real, d.kh, d.kw,
                     (long long)d.dw_group_stride);
  return hnd::check_launch("hnd_conv2d_wgrad(reduce)");
}
