// Epilogue shared by the implicit-GEMM kernels (conv_igemm.hip, conv_bres.hip): one 16-row group of a wave's
// accumulator tiles -> scale/shift, residuals, ReLU-backward mask, ReLU, store (+ BN-statistics partials).
#pragma once
#include "common.h"

namespace hnd {

// Sum of squares of the BN statistics on the CHECKED epilogue path (tile edges, cout = 3): product and sum are two
// roundings, spelled out so that the vector-ALU kernel for cout <= 4 (thin_n_kernel) reproduces the partials bit for
// bit.  Under -ffp-contract=fast the backend fuses any fmul + fadd it sees (a source-level contract(off) pragma does
// not stop it), so the product is hidden behind an empty asm.
__device__ __forceinline__ float sq_acc(float s2, float x) {
  float sq = x * x;
  asm volatile("" : "+v"(sq));
  return s2 + sq;
}

// Epilogue of one 16-row group of a wave: the lane holds rows 4*(lane>>4)+i (i = 0..3) and, thanks to the channel
// interleave of the packed weights (hnd::chan_of_row), NI CONSECUTIVE output channels starting at col0 -- so residual /
// mask loads and the stores are 16-byte (NI = 4) or 8-byte (NI = 2) vector accesses, 16 lanes covering 64 consecutive
// channels of a pixel.
//
// epilogue_rows_full: the whole tile is in range and every pointer / ldc is vector-aligned.  WHICH optional operands
// exist (R1 residual / FPN top-down, R2 second residual, MK ReLU-backward mask) is a TEMPLATE parameter: as a run-time
// `if (d.res1) load else zero` hipcc branches around each group of loads and puts `s_waitcnt vmcnt(0)` on the
// no-operand side (the zeros overwrite registers that are load destinations on the other side) -- three full drains of
// the memory pipe per 16-row group, i.e. twelve per tile, which stalled every store burst and every prefetch in flight
// (round 3: the Winograd component GEMMs, which have no optional operand at all, went 115 -> 140 TFLOP/s with the
// epilogue compiled out; profiles/r03_bres_ablation.txt).  The callers switch ONCE per tile (epilogue_tile).
// MK: 0 = no mask operand, 1 = fp32 mask tensor (d.mask), 2 = mask nibbles (d.mask_bits: one byte per pixel and group of
// four channels -- a lane's NI = 4 consecutive channels are exactly one byte).  d.mask_out (the nibbles of the STORED
// values; NI = 4 builds only -- the launcher keeps such launches on 128-column tiles) is a run-time branch around a
// store: no load sits behind it, so it does not cost the drains described above.
// BS (d.bwd_x): the statistics are the BatchNorm-BACKWARD partials of the stored value (sum d, sum d * xhat; see
// hnd_conv_desc.bwd_x) -- one more operand of the output's geometry, plain epilogues only.
template <int NI>
struct BwdConsts {
  float sc[NI], sh[NI], mu[NI], rs[NI];
};

template <int NI, bool R1, bool R2, int MK, bool BS = false>
__device__ __forceinline__ void epilogue_rows_full(const hnd_conv_desc& d, const f32x4 (&acc)[NI], const int* rowoff,
                                                   const int* resoff, int rbase, int col0, const float (&es)[NI],
                                                   const float (&eb)[NI], float (&s1)[NI], float (&s2)[NI],
                                                   const BwdConsts<NI>& bc,
                                                   const float __attribute__((ext_vector_type(NI))) (&bxv)[4]) {
  typedef float vec __attribute__((ext_vector_type(NI)));
  unsigned off[4];
  vec r1v[4], r2v[4], mkv[4];
  unsigned mkb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    off[i] = (unsigned)rowoff[rbase + i] * (unsigned)d.ldc + (unsigned)col0;
    r1v[i] = 0.f; r2v[i] = 0.f; mkv[i] = 1.f; mkb[i] = 0xfu;
  }
  if (R1) {
    if (d.res1_mode == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r1v[i] = *(const vec*)(d.res1 + (size_t)resoff[rbase + i] * d.ldc + col0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r1v[i] = *(const vec*)(d.res1 + off[i]);
    }
  }
  if (R2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r2v[i] = *(const vec*)(d.res2 + off[i]);
  }
  if (MK == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) mkv[i] = *(const vec*)(d.mask + off[i]);
  }
  if (MK == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) mkb[i] = d.mask_bits[off[i] >> 2];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vec v;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float x = acc[ni][i] * es[ni] + eb[ni];
      if (R1) x += r1v[i][ni];
      if (R2) x += r2v[i][ni];
      if (MK == 1) x = mkv[i][ni] > 0.f ? x : 0.f;
      if (MK == 2) x = ((mkb[i] >> ((off[i] & 3u) + ni)) & 1u) ? x : 0.f;      // (NI = 2: half a nibble per lane)
      x = d.relu ? fmaxf(x, 0.f) : x;
      v[ni] = x;
      if (BS) {
        const float xr = bxv[i][ni];
        const float dd = (d.bwd_relu && !(xr * bc.sc[ni] + bc.sh[ni] > 0.f)) ? 0.f : x;
        s1[ni] += dd;
        s2[ni] += dd * ((xr - bc.mu[ni]) * bc.rs[ni]);
      } else {
        s1[ni] += x;
        s2[ni] += x * x;
      }
    }
    *(vec*)(d.y + off[i]) = v;
    if (NI == 4 && d.mask_out) {
      unsigned nib = 0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) nib |= (v[ni] > 0.f ? 1u : 0u) << ni;
      d.mask_out[off[i] >> 2] = (uint8_t)nib;
    }
  }
  asm volatile("" ::: "memory");
}

// tile edges, odd ldc (91-class logits), unaligned views: scalar, fully checked
template <int NI, bool ALLOW_BS = false>
__device__ __forceinline__ void epilogue_rows_checked(const hnd_conv_desc& d, const f32x4 (&acc)[NI], const int* rowoff,
                                                      const int* resoff, int rbase, int col0, const float (&es)[NI],
                                                      const float (&eb)[NI], float (&s1)[NI], float (&s2)[NI]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int po = rowoff[rbase + i];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = col0 + ni;
      if (po < 0 || col >= d.cout) continue;
      const size_t o = (size_t)po * d.ldc + col;
      float x = acc[ni][i] * es[ni] + eb[ni];
      if (d.res1) x += d.res1_mode == 1 ? d.res1[(size_t)resoff[rbase + i] * d.ldc + col] : d.res1[o];
      if (d.res2) x += d.res2[o];
      if (d.mask) x = d.mask[o] > 0.f ? x : 0.f;
      if (d.mask_bits) x = ((d.mask_bits[o >> 2] >> (o & 3)) & 1u) ? x : 0.f;
      x = d.relu ? fmaxf(x, 0.f) : x;
      d.y[o] = x;
      if (ALLOW_BS && d.bwd_x) {
        const float xr = d.bwd_x[o];
        const float dd = (d.bwd_relu && !(xr * d.bwd_scale[col] + d.bwd_shift[col] > 0.f)) ? 0.f : x;
        s1[ni] += dd;
        s2[ni] += dd * ((xr - d.bwd_mean[col]) * d.bwd_rstd[col]);
        continue;
      }
      s1[ni] += x;
      s2[ni] = sq_acc(s2[ni], x);
    }
    // mask nibbles on the checked path (tile edges): a byte is four channels of ONE lane (col0 is a multiple of 4 here:
    // NI == 4 and the launcher requires ldc % 4 == 0 with mask_out), written whole when its first channel is in range
    if (NI == 4 && d.mask_out && po >= 0 && col0 < d.cout) {
      unsigned nib = 0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (col0 + ni < d.cout) nib |= (d.y[(size_t)po * d.ldc + col0 + ni] > 0.f ? 1u : 0u) << ni;
      d.mask_out[((size_t)po * d.ldc + col0) >> 2] = (uint8_t)nib;
    }
  }
}

// The wave's MI row groups (rows rbase0 + 16*mi + 0..3 of the tables), dispatched once on the operand set.
template <int MI, int NI, bool R1, bool R2, int MK, bool BS = false>
__device__ __forceinline__ void epilogue_tile_full(const hnd_conv_desc& d, const f32x4 (&acc)[MI][NI], const int* rowoff,
                                                   const int* resoff, int rbase0, int col0, const float (&es)[NI],
                                                   const float (&eb)[NI], float (&s1)[NI], float (&s2)[NI]) {
  BwdConsts<NI> bc;
  if (BS) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {            // (a full tile: col0 + ni < cout)
      bc.sc[ni] = d.bwd_scale[col0 + ni];
      bc.sh[ni] = d.bwd_shift[col0 + ni];
      bc.mu[ni] = d.bwd_mean[col0 + ni];
      bc.rs[ni] = d.bwd_rstd[col0 + ni];
    }
  }
  // BS: the rows of bwd_x of group mi + 1 are requested before group mi is processed (requested inside the group, the
  // four round trips of a tile stood in line: the launch with the sums took 1.55 ms instead of 1.22)
  typedef float vec __attribute__((ext_vector_type(NI)));
  vec bx[2][4];
  auto fetch = [&](int mi) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      bx[mi & 1][i] = *(const vec*)(d.bwd_x + (size_t)((unsigned)rowoff[rbase0 + 16 * mi + i] * (unsigned)d.ldc +
                                                       (unsigned)col0));
  };
  if (BS) fetch(0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    if (BS && mi + 1 < MI) fetch(mi + 1);
    epilogue_rows_full<NI, R1, R2, MK, BS>(d, acc[mi], rowoff, resoff, rbase0 + 16 * mi, col0, es, eb, s1, s2, bc,
                                           bx[mi & 1]);
  }
}

// ALLOW_BS: only the tiled kernel is built with the BatchNorm-backward statistics (the persistent kernels never see `stats`)
template <int MI, int NI, bool ALLOW_BS = false>
__device__ __forceinline__ void epilogue_tile(const hnd_conv_desc& d, const f32x4 (&acc)[MI][NI], const int* rowoff,
                                              const int* resoff, int rbase0, int col0, const float (&es)[NI],
                                              const float (&eb)[NI], float (&s1)[NI], float (&s2)[NI], bool full) {
  if (!full) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      epilogue_rows_checked<NI, ALLOW_BS>(d, acc[mi], rowoff, resoff, rbase0 + 16 * mi, col0, es, eb, s1, s2);
    return;
  }
  if (ALLOW_BS && d.bwd_x) {     // (the launcher admits plain epilogues only)
    epilogue_tile_full<MI, NI, false, false, 0, true>(d, acc, rowoff, resoff, rbase0, col0, es, eb, s1, s2);
    return;
  }
  const int sel = (d.res1 ? 1 : 0) | (d.res2 ? 2 : 0) | (d.mask ? 4 : (d.mask_bits ? 8 : 0));
#define HND_EPI(R1, R2, MK) epilogue_tile_full<MI, NI, R1, R2, MK>(d, acc, rowoff, resoff, rbase0, col0, es, eb, s1, s2)
  switch (sel) {
    case 0: HND_EPI(false, false, 0); break;
    case 1: HND_EPI(true, false, 0); break;
    case 2: HND_EPI(false, true, 0); break;
    case 3: HND_EPI(true, true, 0); break;
    case 4: HND_EPI(false, false, 1); break;
    case 5: HND_EPI(true, false, 1); break;
    case 6: HND_EPI(false, true, 1); break;
    case 7: HND_EPI(true, true, 1); break;
    case 8: HND_EPI(false, false, 2); break;
    case 9: HND_EPI(true, false, 2); break;
    case 10: HND_EPI(false, true, 2); break;
    default: HND_EPI(true, true, 2); break;
  }
#undef HND_EPI
}

}  // namespace hnd
