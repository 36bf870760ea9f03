// Weight gradient on fp32 MFMA for gfx950 (MI355X), ring design (round 4): the cout >= 128 launches of hnd_conv2d_wgrad
// -- the direct 2x2 head convs and the grouped Winograd-domain reductions  S_f = Z_f^T V_f  (5.1 ms of the round-3 step
// on conv_wgrad.hip's LDS-staged split-K kernel at 0.56-0.69 of the matrix peak: its 16-deep k-steps pay two barriers and
// 4-byte LDS fragment reads per 16 pixels).
//
//   dW[co][col] = sum_m dy[m][co] * a(m; col),   col = (tap, ci),   m = pixels (1.1 M at batch 16) = the GEMM's K
//
// What made bres2 / bstream fast carries over, with BOTH operands streaming and the accumulators resident:
//   * no LDS at all.  v_mfma_f32_16x16x4_f32 takes, per lane, ONE element of A (row lane % 16, k = lane / 16) and one of
//     B; here k = pixel and rows / columns = channels, and both tensors lie [pixel][channel] with the channel contiguous.
//     A lane's 16-byte load of pixel m0 + lane / 16, channels c0 + 4 (lane % 16) .. + 3 is therefore FOUR fragments at
//     once -- register r is the fragment of the 16 rows {c0 + 4 i + r} -- straight from global memory, no transpose;
//   * one wave per SIMD with the whole register file: a wave owns 64 AH x 64 BH of dW (AH, BH in {1, 2}; 128 x 128 = 64
//     accumulator tiles = 256 registers), so a k-step of 4 pixels is AH + BH loads per lane against 16 AH BH MFMAs
//     (64 MFMAs = 2048 matrix-pipe cycles per 4 loads at 128 x 128);
//   * the loads run RING = 8 k-steps ahead through inline-asm `global_load_dwordx4` with one hand-counted
//     `s_waitcnt vmcnt(7 (AH + BH))` per k-step (memory operations retire in issue order: at most that many younger
//     loads exist, so the slot has landed; tools/audit_bres_asm.py checks the register side in the disassembly);
//   * a workgroup (2 x 2 waves: 128 AH x 128 BH of dW) is alive for its whole pixel range; the partial tile goes to a
//     slab and wgrad_reduce_kernel (conv_wgrad.hip) sums the slabs in fixed order -> bitwise reproducible.
// Taps (the direct 2x2 convs): column half h of a wave is one tap's 64 channels, so its pixel is the output pixel shifted
// by the tap; out-of-range taps and pixels beyond M read a page of zeros, and the BN(+ReLU) prologue of the forward pass
// is applied to the fragments in registers (padding is a zero of the NORMALISED tensor: selected after the prologue).
//
// Roofline: fp32 MFMA (157.3 TFLOP/s); 2 M cout kh kw cin flop per launch.  Operand traffic per workgroup and k-step:
// 4 pixels x (128 AH + 128 BH) channels x 4 B = 4 KB per 2048 cycles at 256 x 256 -- 2 B / clk / CU.
#include "common.h"

#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <utility>

namespace {

using hnd::f32x4;
using hnd::FastDiv;

__device__ float g_wr_zero[256];        // 1 KB of zeros: the source of pixels beyond M and of out-of-range taps

struct WringArgs {
  hnd_wgrad_desc d;
  FastDiv div_ow, div_oh;
  int M;                      // pixels per group = K extent of the reduction
  int rtiles, ctiles;         // workgroup tiles over (cout, cols)
  int blocks_per_split;       // ring blocks (8 k-steps = 32 pixels) per workgroup
  int total_blocks;           // ceil(M / 32)
  int co_pad, ncols_pad;      // slab extent (wgrad_reduce_kernel's layout)
  int splits;
};

template <int N, class F, int... I>
__device__ __forceinline__ void wfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wfor(F&& f) {
  wfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ void rload(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void rwait(f32x4& a0, f32x4& a1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void rwait(f32x4& a0, f32x4& a1, f32x4& a2) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a0), "+v"(a1), "+v"(a2) : "n"(N));
}
template <int N>
__device__ __forceinline__ void rwait(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}

// AH / BH: 64-channel halves of a wave's tile along cout / along the columns.  TAPS: columns are (tap, ci) of a conv with
// kh * kw > 1 or padding, each 64-column half inside ONE tap (cin % 64 == 0); else a 1x1 problem (x pixel = dy pixel).
// WPS: workgroups per CU = waves per SIMD.  The 1x1 problems run one wave per SIMD on 128-row wave tiles (all the vector
// work of a k-step is two address computations); the tap problems run TWO waves per SIMD on 64-row wave tiles, so that
// one wave's address arithmetic, prologue and padding selects (~100 vector instructions per k-step) issue while the
// other wave's MFMAs own the matrix pipe -- interleaving them by hand inside one wave cost more registers than it hid.
// WC: wave columns of the workgroup -- 2: 2 x 2 waves, a 128 AH x 128 BH tile of dW; 1: 4 x 1 waves, 256 AH x 64 BH (the
// Winograd-domain weight gradient of a conv with 64 input channels: 64 columns per component).
template <int AH, int BH, bool TAPS, int WPS, int WC = 2>
__global__ void __launch_bounds__(256, WPS) wgrad_ring_kernel(const WringArgs a) {
  constexpr int RING = 8, NL = AH + BH, VM = NL * (RING - 1);
  const hnd_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = WC == 2 ? wave >> 1 : wave, wc = WC == 2 ? wave & 1 : 0;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int grp = blockIdx.y;
  const float* __restrict__ xg = d.x + (size_t)grp * (size_t)d.x_group_stride;
  const float* __restrict__ dyg = d.dy + (size_t)grp * (size_t)d.dy_group_stride;
  int bid = blockIdx.x;
  const int tiles = a.rtiles * a.ctiles;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int rt = bid / a.ctiles, ct = bid - rt * a.ctiles;
  const int co0 = rt * (64 * AH * (4 / WC)) + wr * (64 * AH);    // this wave's first output channel ...
  const int col0 = ct * (64 * BH * WC) + wc * (64 * BH);         // ... and first column
  const int blk0 = split * a.blocks_per_split;
  int nblk = a.total_blocks - blk0;
  if (nblk > a.blocks_per_split) nblk = a.blocks_per_split;
  const int M = a.M;
  const int m_end = min(M, (blk0 + nblk) * 32);              // pixels of this workgroup: [32 blk0, m_end)

  const float* zero = (const float*)g_wr_zero + 4 * l16;
  // column half h: its tap and first input channel (wave-uniform), the prologue constants of the lane's 4 channels
  int tap_i[BH], tap_j[BH], ci0[BH];
  f32x4 ps[BH], pb[BH];
#pragma unroll
  for (int h = 0; h < BH; ++h) {
    const int c = col0 + 64 * h;
    const int tap = TAPS ? c / d.cin : 0;
    ci0[h] = c - tap * d.cin;
    tap_i[h] = tap / d.kw;
    tap_j[h] = tap - tap_i[h] * d.kw;
    ps[h] = f32x4{1.f, 1.f, 1.f, 1.f};
    pb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (TAPS && d.pro_scale) {
      ps[h] = *(const f32x4*)(d.pro_scale + ci0[h] + 4 * l16);
      if (d.pro_shift) pb[h] = *(const f32x4*)(d.pro_shift + ci0[h] + 4 * l16);
    }
  }
  const float relu_floor = (TAPS && d.pro_scale && d.pro_relu) ? 0.f : -INFINITY;

  // ---- the lane's load addresses, k-step after k-step (pixel m = 4 ks + g4: A = dy row m, B = one x row per column half).
  // The k-steps are requested strictly in order, so the pixel's coordinates are carried along instead of divided out:
  // (oh_, ow_) and px0 = (n H + oh_ s) W + ow_ s, the input pixel of tap (0, 0) without padding.
  int m_run = 4 * (blk0 * 8) + g4;
  int ow_run = 0, oh_run = 0, px_run = 0;
  if (TAPS) {
    const unsigned mm = (unsigned)(m_run < M ? m_run : 0);
    const unsigned t = hnd::fdiv(mm, a.div_ow), n_ = hnd::fdiv(t, a.div_oh);
    ow_run = (int)(mm - t * (unsigned)d.ow);
    oh_run = (int)(t - n_ * (unsigned)d.oh);
    px_run = ((int)n_ * d.h + oh_run * d.stride) * d.w_ + ow_run * d.stride;
  }
  const int row_wrap = d.stride * d.w_ - d.ow * d.stride;            // px0 correction when ow_ wraps into the next row
  const int img_wrap = d.h * d.w_ - d.oh * d.stride * d.w_;          // ... and when oh_ wraps into the next image
  int tap_off[BH], tap_dh[BH], tap_dw[BH];
#pragma unroll
  for (int h = 0; h < BH; ++h) {
    tap_dh[h] = tap_i[h] - d.pad;
    tap_dw[h] = tap_j[h] - d.pad;
    tap_off[h] = tap_dh[h] * d.w_ + tap_dw[h];
  }
  const unsigned dy_row = (unsigned)d.ldy * 4u, x_row = (unsigned)d.cin * 4u;      // bytes per pixel
  const char* dy_base = (const char*)(dyg + co0 + 4 * l16);
  auto addr = [&](const float*& pa, const float* (&pbv)[BH], unsigned& okb) __attribute__((always_inline)) {
    const bool mok = m_run < m_end;
    pa = mok ? (const float*)(dy_base + (size_t)((unsigned)m_run * (unsigned long long)dy_row)) : zero;
    okb = 0;
    if (!TAPS) {
      pbv[0] = mok ? (const float*)((const char*)(xg + col0 + 4 * l16) + (size_t)((unsigned)m_run * (unsigned long long)x_row))
                   : zero;
      okb = mok ? 3u : 0u;
    } else {
#pragma unroll
      for (int h = 0; h < BH; ++h) {
        const int ih = oh_run * d.stride + tap_dh[h], iw = ow_run * d.stride + tap_dw[h];
        const bool ok = mok && (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
        const unsigned px = (unsigned)(px_run + tap_off[h]);
        const float* src = (const float*)((const char*)(xg + ci0[h] + 4 * l16) + (size_t)(px * (unsigned long long)x_row));
        pbv[h] = ok ? src : zero;
        okb |= (ok ? 1u : 0u) << h;
      }
      ow_run += 4;                                           // one k-step = 4 pixels on (ow >= 4: at most one row wrap)
      px_run += 4 * d.stride;
      const bool wrap = ow_run >= d.ow;
      ow_run -= wrap ? d.ow : 0;
      oh_run += wrap ? 1 : 0;
      px_run += wrap ? row_wrap : 0;
      const bool wrap2 = oh_run >= d.oh;
      oh_run -= wrap2 ? d.oh : 0;
      px_run += wrap2 ? img_wrap : 0;
    }
    m_run += 4;
  };

  f32x4 ra[RING][AH], rb[RING][BH];
  unsigned okr[RING];
  auto issue = [&](auto U) __attribute__((always_inline)) {       // the next k-step, into ring slot u
    constexpr int u = decltype(U)::value;
    const float* pa;
    const float* pbv[BH];
    unsigned okb;
    addr(pa, pbv, okb);
    okr[u] = okb;
    rload<0>(ra[u][0], pa);
    if constexpr (AH == 2) rload<256>(ra[u][1], pa);
    rload<0>(rb[u][0], pbv[0]);
    if constexpr (BH == 2) {
      if constexpr (TAPS) rload<0>(rb[u][1], pbv[1]);
      else rload<256>(rb[u][1], pbv[0]);
    }
  };

  f32x4 acc[4 * AH][4 * BH];
#pragma unroll
  for (int i = 0; i < 4 * AH; ++i)
#pragma unroll
    for (int j = 0; j < 4 * BH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nblk > 0) {
    wfor<RING>([&](auto U) __attribute__((always_inline)) { issue(U); });
    for (int b = 0; b < nblk * (8 / RING); ++b) {            // a block = 8 k-steps = 8 / RING turns of the ring
      wfor<RING>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        if constexpr (NL == 2) rwait<VM>(ra[u][0], rb[u][0]);
        else if constexpr (NL == 3 && AH == 2) rwait<VM>(ra[u][0], ra[u][1], rb[u][0]);
        else if constexpr (NL == 3) rwait<VM>(ra[u][0], rb[u][0], rb[u][1]);
        else rwait<VM>(ra[u][0], ra[u][1], rb[u][0], rb[u][1]);
        f32x4 bv[BH];
#pragma unroll
        for (int h = 0; h < BH; ++h) {
          f32x4 v = rb[u][h];
          if (TAPS) {
            v = v * ps[h] + pb[h];
            v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor);
            v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
            const bool ok = (okr[u] >> h) & 1;               // padding / beyond M: a zero AFTER the prologue
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
          }
          bv[h] = v;
        }
#pragma unroll
        for (int ha = 0; ha < AH; ++ha)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float av = ra[u][ha][r];
#pragma unroll
            for (int hb = 0; hb < BH; ++hb)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[4 * ha + r][4 * hb + q] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[hb][q], acc[4 * ha + r][4 * hb + q], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        issue(U);                                            // beyond this workgroup's range: the page of zeros
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's last (unused) requests land before the end
  }

  // ---- partial tile -> slab[group][split][co][col].  D[i][j]: the lane holds column j = l16 and rows i = 4 g4 + q in
  // its four registers; tile (ha, r) x (hb, c) covers co = co0 + 64 ha + 4 i + r, col = col0 + 64 hb + 4 j + c, so the
  // four tiles c = 0..3 of a lane are 16 contiguous bytes of one slab row
  float* slab = d.slabs + ((size_t)grp * a.splits + split) * (size_t)a.co_pad * (size_t)a.ncols_pad;
#pragma unroll
  for (int ha = 0; ha < AH; ++ha)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + 64 * ha + 4 * (4 * g4 + q) + r;
#pragma unroll
        for (int hb = 0; hb < BH; ++hb) {
          f32x4 v = {acc[4 * ha + r][4 * hb + 0][q], acc[4 * ha + r][4 * hb + 1][q], acc[4 * ha + r][4 * hb + 2][q],
                     acc[4 * ha + r][4 * hb + 3][q]};
          *(f32x4*)(slab + (size_t)co * a.ncols_pad + col0 + 64 * hb + 4 * l16) = v;
        }
      }
}

int cu_count_wr() {
  static std::atomic<int> cached{0};
  int v = cached.load(std::memory_order_relaxed);
  if (v == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached.store(v, std::memory_order_relaxed);
  }
  return v;
}

// The tap form (direct 2x2 convs; ~100 vector instructions of address arithmetic, prologue and padding selects per
// k-step).  Measured at batch 16 (tools/bench_wgrad.py, staged kernel -> one wave per SIMD on the large tile / two waves
// per SIMD on 64 x 64 wave tiles): 64 -> 256 (conv1) 1.455 -> 1.361 / 1.384 ms, 64 -> 128 (conv5) 0.800 -> 0.807 /
// 0.752 ms.  So: 256 output channels take the large tile, 128 the small one with two waves per SIMD.
bool taps_small(const hnd_wgrad_desc& d) { return d.cout % 256 != 0; }

template <int AH, int BH>
int launch_ab(const WringArgs& a, bool taps, dim3 grid, hipStream_t s) {
  if (taps) {
    if constexpr (AH == 1 && BH == 1) {
      if (taps_small(a.d)) {
        hipLaunchKernelGGL((wgrad_ring_kernel<1, 1, true, 2>), grid, dim3(256), 0, s, a);
        return hnd::check_launch("hnd_conv2d_wgrad(ring, taps x2)");
      }
    }
    hipLaunchKernelGGL((wgrad_ring_kernel<AH, BH, true, 1>), grid, dim3(256), 0, s, a);
    return hnd::check_launch("hnd_conv2d_wgrad(ring, taps)");
  }
  hipLaunchKernelGGL((wgrad_ring_kernel<AH, BH, false, 1>), grid, dim3(256), 0, s, a);
  return hnd::check_launch("hnd_conv2d_wgrad(ring)");
}

// 64 columns (cin = 64, 1x1 / grouped): four waves stacked along cout, two workgroups per CU (148 registers per lane)
int launch_narrow(const WringArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_ring_kernel<1, 1, false, 2, 1>), grid, dim3(256), 0, s, a);
  return hnd::check_launch("hnd_conv2d_wgrad(ring, 64 columns)");
}

}  // namespace

namespace hnd {

// Taken for: cout a multiple of 128, cin a multiple of 64 with no channel padding, kh * kw * cin a multiple of 128;
// 1x1 problems (the grouped Winograd-domain reductions) must be dense (stride 1, no padding, x pixel = dy pixel).
bool wgrad_ring_applies(const hnd_wgrad_desc& d) {
  // HND_WGRAD_RING=0: never (the LDS-staged kernel of conv_wgrad.hip: A/B, tests); HND_DEBUG_PICKER=wgrad_ring_taps: also
  // the tap form.  Default: the
  // 1x1 (grouped Winograd-domain) form only -- in the step the tap form of the head's direct 2x2 convs is no faster than
  // the staged kernel (conv1 1.37 vs 1.31 ms, conv5 0.74 vs 0.69 ms, profiles/r04_per_launch_events*.txt; in isolation
  // it wins by 5-7 %, tools/bench_wgrad.py): its ~100 vector instructions per k-step are not hidden yet.
  const char* e = getenv("HND_WGRAD_RING");
  if (e && atoi(e) == 0) return false;
  const int mode = hnd::debug_picker("wgrad_ring_taps") > 0 ? 2 : 1;
  if (mode != 2 && (d.kh * d.kw > 1 || d.pad != 0 || d.stride != 1)) return false;
  if (d.cout % 128 != 0 || d.cin % 64 != 0 || d.cin_real != d.cin || d.ldy % 4 != 0 || d.ldy < d.cout) return false;
  const bool taps = d.kh * d.kw > 1 || d.pad != 0 || d.stride != 1;
  const bool narrow = !taps && d.cin == 64 && d.cout % 256 == 0;      // 64 columns: the 4 x 1 wave arrangement
  if ((d.kh * d.kw * d.cin) % 128 != 0 && !narrow) return false;
  if (!taps && (d.oh != d.h || d.ow != d.w_)) return false;
  if (!taps && d.pro_scale) return false;               // (the prologue lives on the tap path)
  if (taps && d.ow < 4) return false;                   // (a k-step of 4 pixels wraps at most one row)
  // 32-bit pixel arithmetic: element offsets are formed in size_t, pixel indices in int
  if ((long long)d.n * d.oh * d.ow >= (1ll << 31) - 64 || (long long)d.n * d.h * d.w_ >= (1ll << 31)) return false;
  return true;
}

static bool wring_narrow(const hnd_wgrad_desc& d) {
  return d.kh * d.kw == 1 && d.pad == 0 && d.stride == 1 && d.cin == 64 && d.cout % 256 == 0;
}

static void wring_plan(const hnd_wgrad_desc& d, WringArgs& a, int& ah, int& bh) {
  a.d = d;
  a.M = d.n * d.oh * d.ow;
  const int ncols = d.kh * d.kw * d.cin;
  const bool taps = d.kh * d.kw > 1 || d.pad != 0 || d.stride != 1;
  const bool narrow = wring_narrow(d);
  const bool small = (taps && taps_small(d)) || narrow;     // 64 x 64 wave tiles, two waves per SIMD
  ah = (d.cout % 256 == 0 && !small) ? 2 : 1;
  bh = (ncols % 256 == 0 && !small) ? 2 : 1;
  a.rtiles = narrow ? d.cout / 256 : d.cout / (128 * ah);
  a.ctiles = narrow ? 1 : ncols / (128 * bh);
  a.co_pad = d.cout;
  a.ncols_pad = ncols;
  a.div_ow = make_fastdiv((unsigned)d.ow);
  a.div_oh = make_fastdiv((unsigned)d.oh);
  a.total_blocks = (a.M + 31) / 32;
  const int groups = d.groups > 1 ? d.groups : 1;
  const int tiles = a.rtiles * a.ctiles * groups;
  // one workgroup per CU (two for the tap form), as many as divide evenly
  int splits = d.splitk > 0 ? d.splitk : (cu_count_wr() * (small ? 2 : 1)) / tiles;
  if (splits < 1) splits = 1;
  if (splits > a.total_blocks) splits = a.total_blocks;
  if (splits > 512) splits = 512;
  a.blocks_per_split = (a.total_blocks + splits - 1) / splits;
  a.splits = (a.total_blocks + a.blocks_per_split - 1) / a.blocks_per_split;     // drop empty splits
}

size_t wgrad_ring_workspace(const hnd_wgrad_desc& d) {
  WringArgs a;
  int ah, bh;
  wring_plan(d, a, ah, bh);
  return (size_t)(d.groups > 1 ? d.groups : 1) * a.splits * a.co_pad * a.ncols_pad * sizeof(float);
}

// launches the partial-sum kernel; the caller runs wgrad_reduce_kernel over `splits` slabs of [co_pad][ncols_pad]
int launch_wgrad_ring(const hnd_wgrad_desc& d, int& splits, int& co_pad, int& ncols_pad, hipStream_t s) {
  WringArgs a;
  int ah, bh;
  wring_plan(d, a, ah, bh);
  splits = a.splits;
  co_pad = a.co_pad;
  ncols_pad = a.ncols_pad;
  const bool taps = d.kh * d.kw > 1 || d.pad != 0 || d.stride != 1;
  const dim3 grid(a.rtiles * a.ctiles * a.splits, d.groups > 1 ? d.groups : 1);
  if (wring_narrow(d)) return launch_narrow(a, grid, s);
  if (ah == 2) return bh == 2 ? launch_ab<2, 2>(a, taps, grid, s) : launch_ab<2, 1>(a, taps, grid, s);
  return bh == 2 ? launch_ab<1, 2>(a, taps, grid, s) : launch_ab<1, 1>(a, taps, grid, s);
}

}  // namespace hnd
