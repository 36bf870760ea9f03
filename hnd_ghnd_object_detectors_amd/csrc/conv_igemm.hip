// Implicit-GEMM convolution on fp32 MFMA for gfx950 (MI355X): forward convs and data-gradients.
//
//   C[m][co] = sum_k A[m][k] * W[co][k],   m = output pixel, k = (tap, ci)
//
// Block tile BM (pixels, 128 or 64) x BN (channels, 128 or 64) x BK (k, 16 or 32); 256 threads = 4 waves in a 2x2
// grid, each wave owns BM/2 x BN/2 as (BM/32) x (BN/32) accumulator tiles of v_mfma_f32_16x16x4_f32.
// WHY 16x16x4 AND NOT 32x32x2 (both exact fp32, both 64 flop/clk/SIMD): with three waves per SIMD issuing matrix
// instructions -- what this kernel needs to hide its gather latency -- the 32x32x2 shape sustains only 63 % of the
// matrix rate on MI355X while 16x16x4 holds 95 % (tools/probes/mfma_f32_probe.hip, profiles/r02_mfma_probe.txt: one
// or two waves per SIMD reach 99 % with either shape, four waves 67 % / 79 %).  Round 1's 32x32x2 build sat at
// exactly that ~65 %.
// The host picks the tile per launch: 128x128 where there are many tiles, smaller tiles where the grid would otherwise
// fill the 256 CUs unevenly (layer3/layer4: M = 67k / 17k pixels at batch 16).
// Both operands are staged K-contiguous in LDS (unpadded [row][BK] rows, 16-byte chunks XOR-swizzled by f(row)) so a
// lane reads 4 consecutive k of its row with one ds_read_b128 and feeds 4 MFMAs: lane group g = lane>>4 holds
// k = 4g..4g+3 of a 16-wide k group, identically for A and B, i.e. MFMA s sums k = s, 4+s, 8+s, 12+s -- a fixed
// permutation of the reduction order that fp32 addition tolerates (and that is the same for every tile variant, so
// results do not depend on the tile picked).
// The packed weight rows are channel-interleaved (hnd::chan_of_row): a lane's NI accumulator tiles are NI consecutive
// output channels, which makes every epilogue access a 16-byte vector access.
// Global -> register -> LDS staging is double buffered: tile t+1 is in flight while tile t is on
// the matrix cores; one barrier per k-step.  The A gather applies the fused BatchNorm/ReLU
// prologue in registers, and zero-fills out-of-range taps AFTER the prologue (padding is a zero
// of the normalised tensor, resnet_layer.py:43-50).
//
// Roofline: compute bound on the fp32 matrix pipe (157.3 TFLOP/s): per block k-step 128*BN*32*2
// flop vs (128+BN)*32*4 B staged => 64 flop/B at BN=128.
#include <atomic>

#include "common.h"
#include "conv_epilogue.h"

#include <stdlib.h>

#include <type_traits>

#ifndef HND_BPC_128
#define HND_BPC_128 3
#endif

namespace {

using hnd::f32x16;
using hnd::f32x4;
using hnd::sq_acc;


template <int BM, int BN, int BK, bool PRO>
constexpr size_t lds_bytes() {
  return (size_t)(2 * (BM + BN) * BK) * sizeof(float);
}

// resident blocks per CU the launch bounds ask for (LDS and VGPR budgets both allow it)
template <int BM, int BN, int BK>
constexpr int blocks_per_cu() {
  return BK == 16 ? ((BM == 64 && BN == 64) ? 5 : ((BM == 128 && BN == 128) ? HND_BPC_128 : 4))
                  : ((BM == 64 && BN == 64) ? 4 : 2);
}

template <int BM, int BN, int BK, bool CIN4, bool PRO>
__global__ void __launch_bounds__(256, (blocks_per_cu<BM, BN, BK>()))
igemm_kernel(const hnd_conv_desc d, const int ntiles) {
  // LDS rows are BK floats, unpadded, with the 16-byte chunks of a row XOR-swizzled by f(row) so that the
  // ds_read_b128 fragment reads (16 rows x 4 chunks per wave instruction) and the staging writes are both
  // bank-conflict free: chunk c of row r lives at position c ^ f(r), f(r) = (r >> 2) & 3 for 64-byte rows,
  // (r >> 1) & 7 for 128-byte rows.
  constexpr int LDK = BK;
  constexpr int CH = BK / 4;
  constexpr int TPR = BK / 4;            // threads per staged row (one float4 each)
  constexpr int RPP = 256 / TPR;         // rows staged per pass
  constexpr int RA = BM / RPP;           // A rows gathered per thread
  constexpr int RB = BN / RPP;           // B rows loaded per thread
  constexpr int WTM = BM / 2, WTN = BN / 2;      // wave tile: 2 x 2 waves per block
  constexpr int MI = WTM / 16, NI = WTN / 16;    // 16x16 accumulator tiles per wave
  constexpr int KG = BK / 16;                    // 16-deep k groups per k-step (4 MFMAs of k = 4 each)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NST = 2;                     // LDS stages (double buffer)
  float* As = smem;                          // [NST][BM][BK]
  float* Bs = smem + NST * BM * BK;          // [NST][BN][BK]
  int* rowoff = (int*)Bs;                    // [BM] output pixel index (or -1); aliases the B staging buffer,
  int* resoff = rowoff + BM;                 // [BM] res1 pixel index (mode 1)  -- filled after the k loop

  // XCD-aware bijective remap: blocks b and b+8 share an XCD/L2, give each XCD a contiguous run
  // of logical tiles so the N-tiles of one pixel tile hit the same L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mt = bid / ntiles, nt = bid - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int M = d.n * d.oh * d.ow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int arow = tid / TPR, kq = tid % TPR;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int fa = (CH == 4 ? (arow >> 2) : (arow >> 1)) & (CH - 1);     // swizzle of this thread's staged rows
  const int fl = (CH == 4 ? (l16 >> 2) : (l16 >> 1)) & (CH - 1);       // swizzle of this lane's fragment rows

  // per-thread gather rows (fixed for the whole k loop)
  int a_nb[RA], a_ih[RA], a_iw[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + arow + RPP * i;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    const int ow_ = mm % d.ow, t = mm / d.ow, oh_ = t % d.oh, n_ = t / d.oh;
    a_nb[i] = n_ * d.h * d.w_;
    a_ih[i] = oh_ * d.sh + d.bh;
    a_iw[i] = ow_ * d.sw + d.bw;
  }
  const float* wrow[RB];
  const float* wbase = d.w + (d.w_group_rows > 0 ? (size_t)(m0 / d.w_group_rows) * (size_t)d.w_group_stride : 0);
#pragma unroll
  for (int i = 0; i < RB; ++i) wrow[i] = wbase + (size_t)(n0 + arow + RPP * i) * d.kdim + kq * 4;

  const int ntaps = d.kh * d.kw;
  const unsigned kw_inv = (65536u + d.kw - 1) / d.kw;
  const int T = d.kdim / BK;
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;

  // ---- staging pieces.  Tile `tnext` is gathered into registers (ra/rb), later written to LDS.  The pieces are
  // interleaved one by one between groups of MFMAs (see step()): issued in the shadow of the matrix pipe they
  // cost no time, whereas issued as one block before / after the MFMAs they cost 13 % (measured by ablation).
  f32x4 ra[RA], rb[RB];
  f32x4 rps = {1.f, 1.f, 1.f, 1.f}, rpb = {0.f, 0.f, 0.f, 0.f};
  unsigned okmask = 0;
  int c0 = 0, khi = 0, kwi = 0, tnext = 0;      // tap state / index of the next tile to gather
  int g_ti = 0, g_tj = 0, g_cc = 0;
  bool g_tapok = true;

  auto load_begin = [&]() {
    g_ti = khi; g_tj = kwi; g_cc = c0 + kq * 4; g_tapok = true;
    if (CIN4) {
      const int tap = tnext * TPR + kq;
      g_ti = (int)((tap * kw_inv) >> 16);
      g_tj = tap - g_ti * d.kw;
      g_tapok = tap < ntaps;
      g_cc = 0;
    }
    okmask = 0;
    if (PRO) {
      rps = *(const f32x4*)(d.pro_scale + g_cc);
      rpb = *(const f32x4*)(d.pro_shift + g_cc);
    }
  };
  // Branch-free gather: out-of-range taps read element 0 of x (always mapped) and are zeroed by a select when the
  // tile is written to LDS (a divergent `if (ok) load` would make hipcc wait for the load inside the branch).
  auto load_a = [&](int i) {
    const int ih = a_ih[i] + g_ti * d.dh, iw = a_iw[i] + g_tj * d.dw;
    const bool ok = a_ok[i] && g_tapok && (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
    const unsigned pix = ok ? (unsigned)(a_nb[i] + ih * d.w_ + iw) : 0u;
    const unsigned off = pix * (unsigned)d.cin + (ok ? (unsigned)g_cc : 0u);
    ra[i] = *(const f32x4*)(d.x + off);
    okmask |= (ok ? 1u : 0u) << i;
  };
  auto load_b = [&](int i) { rb[i] = *(const f32x4*)(wrow[i] + (size_t)tnext * BK); };
  auto load_end = [&]() {
    if (!CIN4) {   // advance the uniform tap state
      c0 += BK;
      if (c0 >= d.cin) {
        c0 = 0;
        if (++kwi == d.kw) { kwi = 0; ++khi; }
      }
    }
    ++tnext;
  };
  auto store_a = [&](int buf, int i) {
    f32x4 v = ra[i];
    if (PRO) {
      v = v * rps + rpb;
      v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor);
      v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
    }
    const bool ok = (okmask >> i) & 1;       // padding is a zero of the NORMALISED tensor: select after the prologue
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    *(f32x4*)(As + buf * BM * LDK + (arow + RPP * i) * LDK + ((kq ^ fa) * 4)) = v;
  };
  auto store_b = [&](int buf, int i) {
    *(f32x4*)(Bs + buf * BN * LDK + (arow + RPP * i) * LDK + ((kq ^ fa) * 4)) = rb[i];
  };
  auto load_tile = [&]() {
    load_begin();
#pragma unroll
    for (int i = 0; i < RA; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < RB; ++i) load_b(i);
    load_end();
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) store_a(buf, i);
#pragma unroll
    for (int i = 0; i < RB; ++i) store_b(buf, i);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One k-step: MFMAs on LDS buffer `buf` in sub-groups of NI MFMAs (one A fragment element against the NI B
  // fragments); if ST the register tile is written to the other buffer piece by piece during the first half of the
  // sub-groups, if LD the following tile is gathered during the second half.  A lane's ds_read_b128 of a 16-row
  // fragment (row l16, chunk lane>>4) holds k = 4*(lane>>4) + s for s = 0..3: element s feeds MFMA s of the k group.
  constexpr int G = KG * MI * 4, NP = RA + RB;
  static_assert(NP <= G / 2, "staging pieces must fit in half of the MFMA sub-groups");
  auto step = [&](int buf, auto st_tag, auto ld_tag) {
    constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
    const float* Ap = As + buf * BM * LDK + (wm * WTM + l16) * LDK;
    const float* Bp = Bs + buf * BN * LDK + (wn * WTN + l16) * LDK;
    f32x4 a, an, b[NI], bn[NI];
    {
      const int o = ((0 * 4 + g4) ^ fl) * 4;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4*)(Bp + ni * 16 * LDK + o);
      a = *(const f32x4*)(Ap + o);
    }
    an = a;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        // the next A fragment (and, at the end of a k group, the next group's B fragments): four sub-groups ahead
        if (mi + 1 < MI) {
          an = *(const f32x4*)(Ap + (mi + 1) * 16 * LDK + ((kg * 4 + g4) ^ fl) * 4);
        } else if (kg + 1 < KG) {
          const int o = (((kg + 1) * 4 + g4) ^ fl) * 4;
          an = *(const f32x4*)(Ap + o);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bn[ni] = *(const f32x4*)(Bp + ni * 16 * LDK + o);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int g = (kg * MI + mi) * 4 + s;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[ni][s], acc[mi][ni], 0, 0, 0);
          if (ST && g < NP) {
            if (g < RA) store_a(buf ^ 1, g);
            else store_b(buf ^ 1, g - RA);
          }
          if (LD && g >= G / 2) {
            const int q = g - G / 2;
            if (q == 0) load_begin();
            if (q < RA) load_a(q);
            else if (q < NP) load_b(q - RA);
            if (q == NP - 1) load_end();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        a = an;
        if (mi + 1 == MI && kg + 1 < KG) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b[ni] = bn[ni];
        }
      }
    }
  };
  using std::true_type;
  using std::false_type;

  int cur = 0, t = 0;
  {
    load_tile();           // tile 0 -> registers -> LDS buffer 0
    store_tile(0);
    if (T > 1) load_tile();   // tile 1 -> registers
    __syncthreads();
    for (; t + 2 < T; ++t) {  // steady state: compute t | write t+1 | gather t+2
      step(cur, true_type{}, true_type{});
      __syncthreads();
      cur ^= 1;
    }
    if (t + 1 < T) {
      step(cur, true_type{}, false_type{});
      __syncthreads();
      cur ^= 1;
    }
    step(cur, false_type{}, false_type{});
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // (the k loop ended with a barrier: the staging buffers are free; As becomes the stats scratch, Bs the row tables)
  if (tid < BM) {
    const int m = m0 + tid;
    int po = -1, pr = 0;
    if (m < M) {
      const int ow_ = m % d.ow, t = m / d.ow, oh_ = t % d.oh, n_ = t / d.oh;
      const int yr = oh_ * d.y_sh + d.y_oh, yc = ow_ * d.y_sw + d.y_ow;
      po = (n_ * d.yh + yr) * d.yw + yc;
      if (d.res1_mode == 1)
        pr = (n_ * d.res1_h + (yr * d.res1_h) / d.yh) * d.res1_w + (yc * d.res1_w) / d.yw;
    }
    rowoff[tid] = po;
    resoff[tid] = pr;
  }

  __syncthreads();
  float* red = As;   // [2 (wm)][2 (sum,sumsq)][BN], reused after the final barrier
  // the wave's NI tiles are rows rho0 + 16*ni + l16 of the packed weight matrix = channels col0 + ni (chan_of_row)
  const int rho0 = n0 + wn * WTN;
  const int col0 = (rho0 & ~63) + l16 * 4 + ((rho0 >> 4) & 3);
  const uintptr_t align_bits = (uintptr_t)d.y | (uintptr_t)d.res1 | (uintptr_t)d.res2 | (uintptr_t)d.mask;
  const bool tile_full = (m0 + BM <= M) && (n0 + BN <= d.cout) && (d.ldc % NI == 0) && (align_bits % (4 * NI) == 0);
  float es[NI], eb[NI], s1[NI], s2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const bool col_ok = col0 + ni < d.cout;
    es[ni] = (d.epi_scale && col_ok) ? d.epi_scale[col0 + ni] : 1.f;
    eb[ni] = (d.epi_shift && col_ok) ? d.epi_shift[col0 + ni] : 0.f;
    s1[ni] = 0.f;
    s2[ni] = 0.f;
  }
  hnd::epilogue_tile<MI, NI, !CIN4>(d, acc, rowoff, resoff, wm * WTM + 4 * g4, col0, es, eb, s1, s2, tile_full);
  if (d.stats) {
    // per 128-row tile and channel: fold the four row groups of the wave (lane>>4), then the two row-waves
    const int cl = col0 - n0;                  // channel index inside the block tile
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float a1 = s1[ni], a2 = s2[ni];
      a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
      a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
      if (g4 == 0) {
        red[(wm * 2 + 0) * BN + cl + ni] = a1;
        red[(wm * 2 + 1) * BN + cl + ni] = a2;
      }
    }
    __syncthreads();
    if (tid < BN && (n0 + tid) < d.cout) {
      float* st = d.stats + (size_t)mt * 2 * d.cout + n0 + tid;
      st[0] = red[(0 * 2 + 0) * BN + tid] + red[(1 * 2 + 0) * BN + tid];
      st[d.cout] = red[(0 * 2 + 1) * BN + tid] + red[(1 * 2 + 1) * BN + tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cout <= 4: the 3-channel bottleneck of the b3ch head (layer1.conv3 forward: 2x2x64 -> 3, and the data gradient of
// conv4: 2x2x64 -> 3).  On the 128 x 64 MFMA tile 60 of 64 output columns are padding (4.6 TFLOP/s "useful"); here one
// thread owns one output pixel and its <= 4 channels on the vector ALU, weights broadcast from LDS.
// BIT-IDENTICAL to the MFMA path by construction (tests/test_ops_gpu.py compares the two bit for bit):
//   * v_mfma_f32_16x16x4_f32 is an fmaf chain over its 4 k values; the MFMA kernel feeds MFMA s of a 16-deep k group
//     with k = 4g + s (g = 0..3), so an accumulator sees k in the order 0,4,8,12, 1,5,9,13, ... -- the loop below;
//   * prologue / epilogue are the same expressions compiled in this translation unit (same FMA contraction);
//   * the BN-statistics partials are folded in the MFMA epilogue's tree: per (row half, row group 4g..4g+3) a serial
//     sum over the 16 rows a lane owns, then (A0 + A1) + (A2 + A3), then the two halves.
constexpr int THIN_LDA = 68;      // LDS row stride (floats) of the staged 128 x 64 activation chunk: 64 + 4 pad
template <bool PRO>
__global__ void __launch_bounds__(128, 4) thin_n_kernel(const hnd_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  float* as = tsm;                                // [128][THIN_LDA] one 64-channel chunk of the tile's 128 pixels
  int* pnb = (int*)(as + 128 * THIN_LDA);         // [128] image base (pixels), ih0, iw0 of every row of the tile
  int* pih = pnb + 128;
  int* piw = pih + 128;
  float* xs = as;                                 // epilogue: [4][128] values of this tile (aliases the chunk buffer)
  float* part = xs + 4 * 128;                     // [4][2][4][2] partial (sum, sumsq) per channel / half / group
  const int tid = threadIdx.x;
  const int mt = blockIdx.x, m0 = mt * 128;
  const int M = d.n * d.oh * d.ow;
  // packed row of channel c is 16 * c (hnd::chan_of_row(16 c) == c for c < 4).  The weight addresses are uniform over
  // the workgroup and the packed weights are read-only while this kernel runs: read through the constant address
  // space they are fetched by scalar loads (s_load_dwordx16 per channel and k group) and feed the fmas as SGPR
  // operands -- as plain global loads hipcc issued 16 vector loads + waits per k group, the kernel's whole time.
  typedef const __attribute__((address_space(4))) float* cptr;   // constant address space: uniform loads go to SMEM
  const cptr w0 = (cptr)(d.w);
  const cptr w1 = (cptr)(d.w + (size_t)16 * d.kdim);
  const cptr w2 = (cptr)(d.w + (size_t)32 * d.kdim);
  const cptr w3 = (cptr)(d.w + (size_t)48 * d.kdim);
  const int m = m0 + tid;
  const bool rok = m < M;
  const int mm = rok ? m : 0;
  const int ow_ = mm % d.ow, t_ = mm / d.ow, oh_ = t_ % d.oh, n_ = t_ / d.oh;
  pnb[tid] = rok ? n_ * d.h * d.w_ : -1;
  pih[tid] = oh_ * d.sh + d.bh;
  piw[tid] = ow_ * d.sw + d.bw;
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // Per (tap, 64-channel chunk): the 128 x 64 activation block is gathered COALESCED -- 16 consecutive lanes fetch the
  // 256 contiguous bytes of one pixel -- with the prologue applied, written to LDS, and every thread then reads back
  // its own pixel's row (one lane per pixel straight from global memory touched 64 cache lines per load instruction
  // and ran at the L1's line rate).
  const int piece = tid & 15, prow = tid >> 4;    // this thread stages float4 `piece` of rows prow, prow + 8, ...
  const int cpt = d.cin >> 6;                     // cin % 64 == 0 on this path (thin_n_applies)
  int kbase = 0;
  for (int khi = 0; khi < d.kh; ++khi)
    for (int kwi = 0; kwi < d.kw; ++kwi)
      for (int cc = 0; cc < cpt; ++cc, kbase += 64) {
        const int c0 = cc << 6;
        f32x4 rps = {1.f, 1.f, 1.f, 1.f}, rpb = {0.f, 0.f, 0.f, 0.f};
        if (PRO) {
          rps = *(const f32x4*)(d.pro_scale + c0 + 4 * piece);
          rpb = *(const f32x4*)(d.pro_shift + c0 + 4 * piece);
        }
        __syncthreads();                          // the row tables are written / the previous chunk has been read
        f32x4 st[16];
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r = prow + 8 * j;
          const int ih = pih[r] + khi * d.dh, iw = piw[r] + kwi * d.dw, nb = pnb[r];
          const bool ok = nb >= 0 && (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
          const float* px = d.x + (ok ? (size_t)(unsigned)(nb + ih * d.w_ + iw) * (unsigned)d.cin + c0 + 4 * piece : 0);
          st[j] = *(const f32x4*)px;
          okm |= (ok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          f32x4 a = st[j];
          if (PRO) {
            a = a * rps + rpb;
            a.x = fmaxf(a.x, relu_floor); a.y = fmaxf(a.y, relu_floor);
            a.z = fmaxf(a.z, relu_floor); a.w = fmaxf(a.w, relu_floor);
          }
          const bool ok = (okm >> j) & 1;         // padding is a zero of the NORMALISED tensor
          a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f; a.z = ok ? a.z : 0.f; a.w = ok ? a.w : 0.f;
          *(f32x4*)(as + (prow + 8 * j) * THIN_LDA + 4 * piece) = a;
        }
        __syncthreads();
        const float* row = as + tid * THIN_LDA;
        // (a register prefetch of the next chunk behind these fmas spilled SGPRs and ran 15 % slower: four resident
        // workgroups per CU already cover one another's gather)
#pragma unroll 1
        for (int G = 0; G < 4; ++G) {              // 16-deep k groups of this chunk, in order
          f32x4 v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) v[g] = *(const f32x4*)(row + 16 * G + 4 * g);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float a = v[g][s_];
              const int k = kbase + 16 * G + 4 * g + s_;
              acc[0] = fmaf(a, w0[k], acc[0]);
              acc[1] = fmaf(a, w1[k], acc[1]);
              acc[2] = fmaf(a, w2[k], acc[2]);
              acc[3] = fmaf(a, w3[k], acc[3]);
            }
        }
      }
  __syncthreads();                                // every thread is done with the chunk buffer (xs aliases it)
  // ---- epilogue (the scalar path of epilogue_rows)
  int po = -1;
  if (rok) po = (n_ * d.yh + oh_ * d.y_sh + d.y_oh) * d.yw + ow_ * d.y_sw + d.y_ow;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = 0.f;
    if (po >= 0 && c < d.cout) {
      const float es = d.epi_scale ? d.epi_scale[c] : 1.f, eb = d.epi_shift ? d.epi_shift[c] : 0.f;
      const size_t o = (size_t)po * d.ldc + c;
      x = acc[c] * es + eb;
      if (d.res1) x += d.res1[o];
      if (d.res2) x += d.res2[o];
      if (d.mask) x = d.mask[o] > 0.f ? x : 0.f;
      x = d.relu ? fmaxf(x, 0.f) : x;
      d.y[o] = x;
    }
    xs[c * 128 + tid] = x;                        // rows outside the problem contribute an exact +0
  }
  if (!d.stats) return;
  __syncthreads();
  if (tid < 32) {                                 // (channel, row half, row group): a lane's 16 rows, serially
    const int c = tid >> 3, wm = (tid >> 2) & 1, g4 = tid & 3;
    float s1 = 0.f, s2 = 0.f;
    for (int mi = 0; mi < 4; ++mi)
      for (int i = 0; i < 4; ++i) {
        const float x = xs[c * 128 + wm * 64 + mi * 16 + 4 * g4 + i];
        s1 += x;
        s2 = sq_acc(s2, x);
      }
    part[((c * 2 + wm) * 4 + g4) * 2 + 0] = s1;
    part[((c * 2 + wm) * 4 + g4) * 2 + 1] = s2;
  }
  __syncthreads();
  if (tid < 2 * d.cout) {
    const int c = tid >> 1, q = tid & 1;
    float h[2];
    for (int wm = 0; wm < 2; ++wm) {
      const float* pp = part + ((c * 2 + wm) * 4) * 2 + q;
      h[wm] = (pp[0] + pp[2]) + (pp[4] + pp[6]);
    }
    d.stats[(size_t)mt * 2 * d.cout + (size_t)q * d.cout + c] = h[0] + h[1];
  }
}

bool thin_n_applies(const hnd_conv_desc& d) {
  static const int on = getenv("HND_THIN_N") ? atoi(getenv("HND_THIN_N")) : 1;
  return on && !d.mask_bits && !d.mask_out && !d.bwd_x && d.cin != 4 && d.cin % 64 == 0 && d.cout <= 4 && d.res1_mode == 0 &&
         d.w_group_rows == 0 &&
         d.kdim == d.kh * d.kw * d.cin && d.kdim <= 4096;
}

int launch_thin_n(const hnd_conv_desc& d, hipStream_t stream) {
  const long long M = (long long)d.n * d.oh * d.ow;
  const int mtiles = (int)((M + 127) / 128);
  const size_t lds = ((size_t)128 * THIN_LDA + 3 * 128) * sizeof(float);
  if (d.pro_scale) hipLaunchKernelGGL(thin_n_kernel<true>, dim3(mtiles), dim3(128), lds, stream, d);
  else hipLaunchKernelGGL(thin_n_kernel<false>, dim3(mtiles), dim3(128), lds, stream, d);
  return hnd::check_launch("hnd_conv2d_igemm(thin)");
}

template <int BM, int BN, int BK, bool CIN4, bool PRO>
int launch_pro(const hnd_conv_desc& d, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};    // per device: the attribute lives on the device's function
  auto kern = igemm_kernel<BM, BN, BK, CIN4, PRO>;
  const size_t lds = lds_bytes<BM, BN, BK, PRO>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(igemm<%d,%d>) failed: %s", BM, BN, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const long long M = (long long)d.n * d.oh * d.ow;
  const int mtiles = (int)((M + BM - 1) / BM);
  const int ntiles = (d.cout + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(256), lds, stream, d, ntiles);
  return hnd::check_launch("hnd_conv2d_igemm");
}

// Tile choice.  Resident blocks: 2 per CU for every tile but 64x64 (4 per CU).  A launch runs in
// ceil(blocks / slots) rounds, so a grid of e.g. 1050 128x128 tiles (layer3, 512 slots) wastes a third of its
// last round; smaller tiles quantise better at a lower per-tile efficiency (factors measured with
// tools/bench_conv.py on MI355X).
template <int BM, int BN, int BK, bool CIN4>
int launch(const hnd_conv_desc& d, hipStream_t stream) {
  return d.pro_scale ? launch_pro<BM, BN, BK, CIN4, true>(d, stream) : launch_pro<BM, BN, BK, CIN4, false>(d, stream);
}

// (k-step depth per tile: 16 for 128x128 / 128x64 / 64x128 -- smaller LDS footprint, 3-4 resident blocks per CU -- and 32
// for 64x64; measured in rounds 1-2)
int pick_tile(const hnd_conv_desc& d) {
  {                                                     // testing / tuning override (HND_DEBUG_PICKER=igemm_tile=0..3)
    const int t = hnd::debug_picker("igemm_tile");
    if (t >= 0 && t <= 3) {
      if ((t == 0 || t == 2) && d.cout % 128 != 0) return t + 1;
      if (d.stats && (t == 2 || t == 3)) return t - 2;
      if (d.mask_out && (t == 1 || t == 3)) return t - 1;     // mask nibbles are written by the 128-column tiles
      return t;
    }
  }
  const long long M = (long long)d.n * d.oh * d.ow;
  const bool n128 = d.cout % 128 == 0;
  // eff: large-grid throughput relative to the 128x128 tile (128.6 / 115.6 / 117.1 / 113.8 TFLOP/s measured on a
  // 16800-tile 3x3 conv with tools/bench_conv.py); bpc: resident blocks per CU of the build in use.
  // A launch takes `full` rounds of 256*bpc blocks plus a last partial round in which the busiest CU holds
  // b = ceil(rem / 256) blocks; a CU with b of bpc blocks needs lone + (1 - lone)(b - 1)/(bpc - 1) of a round
  // (a block alone on a CU does not run bpc times faster: 0.57 of a round at bpc 2, 0.34 at bpc 4 -- fitted to
  // the 512->512 @25x42 and 256->256 @50x84 3x3 convs, which the model then predicts within 3 %).
  struct Cand { int id, bm, bn, bpc; double eff, lone; };
  // (a re-fit of `eff` on the isolated-launch sweep of tools/bench_conv.py --tiles for the 16x16x4 kernel -- 1.00 /
  // 0.945 / 0.977 / 0.913 -- picked differently and made the step 1.5 % slower: inside the step the launches run with
  // the previous layer's output in L2 / MALL, so the round-1 constants, fitted in the step, stay)
  const Cand cands[4] = {{0, 128, 128, 3, 1.00, 0.45}, {1, 128, 64, 4, 0.90, 0.34}, {2, 64, 128, 4, 0.91, 0.34},
                         {3, 64, 64, 4, 0.885, 0.34}};
  int best = n128 ? 0 : 1;
  double best_t = 1e300;
  for (const Cand& c : cands) {
    if (c.bn == 128 && !n128) continue;
    if (d.stats && c.bm != 128) continue;          // BN statistics partials are per 128-pixel tile
    if (d.mask_out && c.bn != 128) continue;       // a lane must own whole mask nibbles (4 consecutive channels)
    const long long blocks = ((M + c.bm - 1) / c.bm) * ((d.cout + c.bn - 1) / c.bn);
    const long long slots = 256ll * c.bpc;
    const long long full = blocks / slots, rem = blocks % slots;
    const long long b = (rem + 255) / 256;
    const double last = rem == 0 ? 0.0 : c.lone + (1.0 - c.lone) * (double)(b - 1) / (double)(c.bpc - 1);
    const double t = ((double)full + last) * c.bpc * c.bm * c.bn / c.eff;
    if (t < best_t) { best_t = t; best = c.id; }
  }
  return best;
}

}  // namespace

namespace hnd {
int bres_variant(const hnd_conv_desc& d);                      // conv_bres.hip: B-resident persistent GEMM
int launch_bres(const hnd_conv_desc& d, hipStream_t stream);
bool stem7_applies(const hnd_conv_desc& d);                    // conv_stem.hip: 7x7 s2 stem from an LDS patch
int launch_stem7(const hnd_conv_desc& d, hipStream_t stream);
int bstream_variant(const hnd_conv_desc& d);                   // conv_bstream.hip: B-streamed persistent GEMM (long K)
int launch_bstream(const hnd_conv_desc& d, hipStream_t stream);
size_t bstream_workspace(const hnd_conv_desc& d);
bool bx3_applies(const hnd_conv_desc& d);                      // conv_bx3.hip: fp32 emulated on the bf16 pipe, B resident
int launch_bx3(const hnd_conv_desc& d, hipStream_t stream);
int bxs_variant(const hnd_conv_desc& d);                       // conv_bxs.hip: ... B streamed (taps, long K, any epilogue)
int launch_bxs(const hnd_conv_desc& d, hipStream_t stream);
size_t bxs_workspace(const hnd_conv_desc& d);
}  // namespace hnd

extern "C" int hnd_conv2d_igemm(const hnd_conv_desc* desc, void* stream) {
  HND_REQUIRE(desc != nullptr, "hnd_conv2d_igemm: null descriptor");
  const hnd_conv_desc& d = *desc;
  HND_REQUIRE(d.x && d.w && d.y, "hnd_conv2d_igemm: null x/w/y");
  HND_REQUIRE(d.n > 0 && d.h > 0 && d.w_ > 0 && d.oh > 0 && d.ow > 0 && d.cout > 0 && d.kh > 0 && d.kw > 0,
              "hnd_conv2d_igemm: non-positive geometry");
  HND_REQUIRE(d.cin == 4 || d.cin % 32 == 0, "hnd_conv2d_igemm: cin=%d must be 4 or a multiple of 32", d.cin);
  HND_REQUIRE(d.kdim % 32 == 0 && d.kdim >= d.kh * d.kw * d.cin,
              "hnd_conv2d_igemm: kdim=%d must be a multiple of 32 covering kh*kw*cin=%d", d.kdim,
              d.kh * d.kw * d.cin);
  HND_REQUIRE(d.cin != 4 || d.kh * d.kw <= 64, "hnd_conv2d_igemm: at most 64 taps when cin==4");
  HND_REQUIRE(d.ldc >= d.cout, "hnd_conv2d_igemm: ldc < cout");
  HND_REQUIRE(d.w_group_rows >= 0 && d.w_group_rows % 128 == 0 && (d.w_group_rows == 0 || d.w_group_stride > 0),
              "hnd_conv2d_igemm: w_group_rows=%d must be 0 or a multiple of 128 with a positive stride", d.w_group_rows);
  HND_REQUIRE((long long)d.n * d.oh * d.ow < (1ll << 31) && (long long)d.n * d.yh * d.yw < (1ll << 31) &&
                  (long long)d.n * d.h * d.w_ < (1ll << 31),
              "hnd_conv2d_igemm: pixel count exceeds int32");
  HND_REQUIRE(d.res1_mode == 0 || (d.res1_h > 0 && d.res1_w > 0), "hnd_conv2d_igemm: res1 upsample needs dims");
  HND_REQUIRE(d.pro_scale == nullptr || d.pro_shift != nullptr, "hnd_conv2d_igemm: pro_shift is required with pro_scale");
  HND_REQUIRE((long long)d.n * d.yh * d.yw * d.ldc < (1ll << 32) - 1, "hnd_conv2d_igemm: output exceeds 2^32 elements");
  HND_REQUIRE(!(d.mask && d.mask_bits), "hnd_conv2d_igemm: give mask or mask_bits, not both");
  HND_REQUIRE(!(d.mask_bits || d.mask_out) || (d.ldc % 4 == 0 && d.cin != 4),
              "hnd_conv2d_igemm: mask nibbles need ldc %% 4 == 0 (ldc=%d) and a cin %% 32 == 0 launch", d.ldc);
  HND_REQUIRE(!d.mask_out || d.cout % 128 == 0,
              "hnd_conv2d_igemm: mask_out needs cout %% 128 == 0 (a lane must own whole nibbles), cout=%d", d.cout);
  HND_REQUIRE((long long)d.n * d.h * d.w_ * d.cin < (1ll << 32) - 1, "hnd_conv2d_igemm: input exceeds 2^32 elements");
  HND_REQUIRE(!d.bwd_x || (d.stats && d.bwd_scale && d.bwd_shift && d.bwd_mean && d.bwd_rstd && !d.res1 && !d.res2 &&
                           !d.mask && !d.mask_bits && !d.relu && d.cin != 4),
              "hnd_conv2d_igemm: bwd_x needs stats + the four per-channel vectors and a plain epilogue");
  hipStream_t s = hnd::as_stream(stream);
  if (d.cin == 4) {
    if (hnd::stem7_applies(d)) return hnd::launch_stem7(d, s);
    return launch<128, 64, 32, true>(d, s);   // the 3->64 decoder conv (and any other 4-channel-input conv)
  }
  if (thin_n_applies(d)) return launch_thin_n(d, s);
  if (hnd::bx3_applies(d)) return hnd::launch_bx3(d, s);        // only with hnd_conv_desc.w_bf16x3 attached
  if (hnd::bxs_variant(d)) return hnd::launch_bxs(d, s);        // only with hnd_conv_desc.w_bf16x3s attached
  if (hnd::bres_variant(d)) return hnd::launch_bres(d, s);
  if (hnd::bstream_variant(d)) return hnd::launch_bstream(d, s);
  switch (pick_tile(d)) {
    case 0: return launch<128, 128, 16, false>(d, s);
    case 1: return launch<128, 64, 16, false>(d, s);
    case 2: return launch<64, 128, 16, false>(d, s);
    default: return launch<64, 64, 32, false>(d, s);
  }
}

extern "C" size_t hnd_conv2d_igemm_workspace(const hnd_conv_desc* desc) {
  if (!desc || desc->cin == 4 || thin_n_applies(*desc) || hnd::bx3_applies(*desc)) return 0;
  if (hnd::bxs_variant(*desc)) return hnd::bxs_workspace(*desc);
  if (hnd::bres_variant(*desc)) return 0;
  return hnd::bstream_workspace(*desc);
}

extern "C" int hnd_conv2d_igemm_tile(const hnd_conv_desc* desc) {
  if (!desc) return -1;
  if (desc->cin == 4) return hnd::stem7_applies(*desc) ? 9 : 1;
  if (thin_n_applies(*desc)) return 4;
  if (hnd::bx3_applies(*desc)) return 13;
  if (const int v = hnd::bxs_variant(*desc)) return v == 2 ? 14 : 15;
  if (const int v = hnd::bres_variant(*desc)) return v == 2 ? 5 : (v == 1 ? 6 : (v == 4 ? 7 : 8));
  if (const int v = hnd::bstream_variant(*desc)) return v == 2 ? 11 : 12;
  return pick_tile(*desc);
}
