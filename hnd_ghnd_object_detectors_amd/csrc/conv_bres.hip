// B-resident persistent GEMM on fp32 MFMA for gfx950 (MI355X): the K <= 256 (BN = 128) / K <= 512 (BN = 64) class of
// hnd_conv2d_igemm launches with no taps -- every 1x1 convolution and data gradient of the frozen ResNet / FPN and
// every Winograd component GEMM  M_f[tiles x cout] = V_f[tiles x cin] * U_f^T  (34 ms of the 107 ms round-2 step,
// where the tiled kernel sat at 105-115 TFLOP/s: a 16-k-step tile pays its exposed first-load latency and its
// store burst once per 8 MFLOP, profiles/r02_igemm_sq_counters.txt: matrix pipe 68 % busy, waves parked 25 %).
//
// Structure (one workgroup of 8 waves per CU, 256 workgroups, each alive for the whole launch):
//   * the [BN x K] slice of the packed weight matrix a workgroup needs is loaded ONCE into LDS (128 KB for
//     128 x 256, XOR-swizzled 16-byte chunks -> conflict-free ds_read_b128 fragments) and stays there;
//   * A is never staged: a wave reads its MFMA A-fragments straight from global memory into registers (a lane
//     owns row l16 of a 16-row group and 4 consecutive k = one 16-byte load per (row group, 16-deep k group));
//     the fragment of k group g+4 is requested while group g is on the matrix pipe (a 4-deep register ring), and
//     the first groups of the wave's NEXT 64-row chunk are requested before this chunk's epilogue -- so there is
//     no tile prologue, no LDS write, and NO barrier anywhere in the main loop: the 8 waves free-run, two per SIMD,
//     each filling the other's waits;
//   * work split: `nsl = cout / BN` workgroups that share one L2 (same XCD) form a team, one per weight slice; the
//     64-row chunks are divided statically and evenly among the 256 / nsl teams (persistent: no grid quantisation),
//     inside a team's range the wave rows take chunks round-robin.  A chunk read by one member of a team is an L2 hit
//     for the others, so A crosses HBM once.
// The accumulation order per output element is the implicit-GEMM kernel's (k groups ascending; inside a group MFMA s
// sums k = s, 4+s, 8+s, 12+s), the prologue / epilogue expressions are the same code (conv_epilogue.h): results are
// BIT-IDENTICAL to igemm_kernel's (tests/test_ops_gpu.py::test_bres_kernel_is_bit_identical_to_the_tiled_kernel).
//
// Roofline: fp32 MFMA (157.3 TFLOP/s).  Per wave and k group: 64 MFMAs (2048 cycles of its SIMD's pipe at two
// waves per SIMD: 4096) against 4 global_load_dwordx4 + 4 ds_read_b128 -- 8 B/clk/CU from L2, 4 % of the LDS.
#include <atomic>

#include "common.h"
#include "conv_epilogue.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

using hnd::f32x4;
using hnd::FastDiv;

struct BresArgs {
  FastDiv div_ow, div_oh;     // m -> (n, oh, ow)
  int nsl;                    // weight slices = workgroups per team
  int nchunks;                // ceil(M / 64)
  int cpg;                    // chunks per weight group (Winograd component), 0 = one group
};

// The workgroup's [BN x K] weight slice -> LDS (chunk c of row r at position c ^ (r & 15)).  Up to 16 loads per thread are
// in flight before the first LDS write: written as one load + one write per iteration hipcc waits `vmcnt(0)` in every
// iteration -- 32 (16) dependent round trips to L2, ~25 (13) us per slice at the head of EVERY launch and at every
// Winograd component boundary, 4-8 % of a 0.3-0.6 ms launch (round 4).
template <int BN, int K, int NT>
__device__ __forceinline__ void slice_to_lds(float* Bs, const float* wsrc, int tid) {
  constexpr int cpr = K >> 2;                         // 16-byte chunks per row (a power of two)
  constexpr int PER = BN * cpr / NT;                  // chunks per thread
  constexpr int UMAX = NT == 256 ? 16 : 8;            // (the 8-wave kernel has 256 registers per lane: smaller batches)
  constexpr int UB = PER < UMAX ? PER : UMAX;
  static_assert(BN * cpr % NT == 0 && PER % UB == 0, "slice does not divide among the threads");
#pragma unroll 1
  for (int b0 = 0; b0 < PER; b0 += UB) {
    f32x4 t[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = tid + (b0 + u) * NT, r = e / cpr, ck = e - r * cpr;
      t[u] = *(const f32x4*)(wsrc + (size_t)r * K + (ck << 2));
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = tid + (b0 + u) * NT, r = e / cpr, ck = e - r * cpr;
      *(f32x4*)(Bs + r * K + ((ck ^ (r & 15)) << 2)) = t[u];
    }
  }
}

// WN = wave columns (BN = 64 * WN); the 8 / WN wave rows take 64-row chunks round-robin.  K = 64 * KQ is a template
// parameter and the k loop is fully unrolled: with a loop back edge inside the chunk hipcc drains every load in
// flight (`s_waitcnt vmcnt(0)`) at the loop header, which empties the register ring every four k groups.
template <int WN, int KQ, bool PRO>
__global__ void __launch_bounds__(512, 1) bres_kernel(const hnd_conv_desc d, const BresArgs a) {
  constexpr int WM = 8 / WN, BN = 64 * WN, MI = 4, NI = 4, RING = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = 64 * KQ, KG = 4 * KQ;
  float* Bs = smem;                                   // [BN][K]: chunk c of row r at position c ^ (r & 15)
  float* pro = Bs + BN * K;                           // [2][K] prologue scale, shift
  int* tabs = (int*)(pro + (PRO ? 2 * K : 0));        // [8 waves][2][64]: output / res1 pixel of the wave's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int l16 = lane & 15, g4 = lane >> 4;
  int* rowoff = tabs + wave * 128;
  int* resoff = rowoff + 64;

  // team (shares an XCD / L2) and weight slice of this workgroup
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % a.nsl, tpx = per_xcd / a.nsl;
  const int team = xcd * tpx + idx / a.nsl, nteams = 8 * tpx;
  const int n0 = slice * BN;
  const int M = d.n * d.oh * d.ow;
  const int c_lo = (int)((long long)a.nchunks * team / nteams);
  const int c_hi = (int)((long long)a.nchunks * (team + 1) / nteams);

  // A row m -> element offset of its input pixel (1x1 taps, no padding: always in range)
  auto a_off = [&](int m) -> unsigned {
    m = m < M ? m : M - 1;
    const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
    const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
    return ((n_ * (unsigned)d.h + oh_ * (unsigned)d.sh) * (unsigned)d.w_ + ow_ * (unsigned)d.sw) * (unsigned)d.cin +
           (unsigned)(g4 * 4);
  };

  // epilogue constants of this wave's 4 consecutive channels
  const int col0 = n0 + wn * 64 + l16 * 4;
  float es[NI], eb[NI], s1[NI], s2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    es[ni] = d.epi_scale ? d.epi_scale[col0 + ni] : 1.f;
    eb[ni] = d.epi_shift ? d.epi_shift[col0 + ni] : 0.f;
    s1[ni] = 0.f;
    s2[ni] = 0.f;
  }
  const uintptr_t align_bits = (uintptr_t)d.y | (uintptr_t)d.res1 | (uintptr_t)d.res2 | (uintptr_t)d.mask;
  const bool vec_ok = (d.ldc % NI == 0) && (align_bits % (4 * NI) == 0);
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;
  if (PRO) {
    for (int k = tid; k < K; k += 512) {
      pro[k] = d.pro_scale[k];
      pro[K + k] = d.pro_shift[k];
    }
  }
  // LDS offsets (floats) of this lane's B fragment chunk for k group u of an unrolled quartet: the logical chunk
  // 4*kg + g4 sits at (4*kg + g4) ^ l16 = 4*(kg & ~3) + ((4*(kg & 3) + g4) ^ l16)
  int bsw[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) bsw[u] = ((4 * u + g4) ^ l16) * 4;
  const float* Bw = Bs + (wn * 64 + l16) * K;

  int c = c_lo;
  while (c < c_hi) {            // one pass per weight group met by this team's range (1 or 2, rarely 3)
    const int grp = a.cpg > 0 ? c / a.cpg : 0;
    const int seg_hi = a.cpg > 0 ? min(c_hi, (grp + 1) * a.cpg) : c_hi;
    __syncthreads();                                  // every wave is done with the previous slice
    slice_to_lds<BN, K, 512>(Bs, d.w + (size_t)grp * (size_t)d.w_group_stride + (size_t)n0 * K, tid);
    __syncthreads();

    int cc = c + wm;
    if (cc < seg_hi) {
      unsigned aoff[MI];
      f32x4 ring[RING][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) aoff[mi] = a_off(cc * 64 + mi * 16 + l16);
#pragma unroll
      for (int u = 0; u < RING; ++u)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ring[u][mi] = *(const f32x4*)(d.x + aoff[mi] + u * 16);
      for (; cc < seg_hi; cc += WM) {
        const int cn = cc + WM < seg_hi ? cc + WM : cc;     // the wave's next chunk (itself at the end: harmless)
        unsigned noff[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) noff[mi] = a_off(cn * 64 + mi * 16 + l16);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bcur[NI], bnxt[NI], ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f}, psn = ps, pbn = pb;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bcur[ni] = *(const f32x4*)(Bw + ni * 16 * K + bsw[0]);
        if (PRO) {
          ps = *(const f32x4*)(pro + g4 * 4);
          pb = *(const f32x4*)(pro + K + g4 * 4);
        }
#pragma unroll
        for (int kg0 = 0; kg0 < KG; kg0 += 4) {
          const bool more = kg0 + 4 < KG;
          unsigned poff[MI];                          // where the ring is refilled from: 4 groups ahead
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) poff[mi] = more ? aoff[mi] + (unsigned)((kg0 + 4) * 16) : noff[mi];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // B fragments (and prologue constants) of the following k group; past the end: group 0 again, unused
            // ((kg0 + u + 1) & 3 == (u + 1) & 3 whether or not the quartet wraps: kg0 and KG are multiples of 4)
            const int kn = u < 3 ? kg0 + u + 1 : (more ? kg0 + 4 : 0);
            const int bo = (kn & ~3) * 16 + bsw[(u + 1) & 3];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              f32x4 av = ring[u][mi];
              if (PRO) {
                av = av * ps + pb;
                av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor);
                av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
              }
#pragma unroll
              for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bcur[ni][s], acc[mi][ni], 0, 0, 0);
                if (mi == 0) {
                  bnxt[s] = *(const f32x4*)(Bw + s * 16 * K + bo);
                  if (PRO && s == 3) {
                    psn = *(const f32x4*)(pro + kn * 16 + g4 * 4);
                    pbn = *(const f32x4*)(pro + K + kn * 16 + g4 * 4);
                  }
                }
                if (s == 3) ring[u][mi] = *(const f32x4*)(d.x + poff[mi] + u * 16);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bcur[ni] = bnxt[ni];
            if (PRO) { ps = psn; pb = pbn; }
          }
        }
        // ---- epilogue of this 64 x 64 wave tile (the next chunk's first k groups are already in flight)
        {
          const int m = cc * 64 + lane;
          int po = -1, pr = 0;
          if (m < M) {
            const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
            const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
            const int yr = (int)oh_ * d.y_sh + d.y_oh, yc = (int)ow_ * d.y_sw + d.y_ow;
            po = ((int)n_ * d.yh + yr) * d.yw + yc;
            if (d.res1_mode == 1)
              pr = ((int)n_ * d.res1_h + (yr * d.res1_h) / d.yh) * d.res1_w + (yc * d.res1_w) / d.yw;
          }
          __builtin_amdgcn_wave_barrier();            // the wave's previous epilogue has read its tables
          rowoff[lane] = po;
          resoff[lane] = pr;
          __builtin_amdgcn_wave_barrier();            // LDS executes one wave's accesses in order
          const bool full = vec_ok && (cc * 64 + 64 <= M);
          hnd::epilogue_tile<MI, NI>(d, acc, rowoff, resoff, 4 * g4, col0, es, eb, s1, s2, full);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aoff[mi] = noff[mi];
      }
    }
    c = seg_hi;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bres2: the same data flow with ONE wave per SIMD (256 threads, 512 registers per lane) for launches whose epilogue
// has no residual / mask operand (every Winograd component GEMM, conv1 / downsample 1x1 convs).  What the 8-wave kernel
// above cannot do in 256 registers, and what it loses 10 % to (profiles/r03_bres_ablation.txt):
//   * the finished tile is copied to a second register set and its scale/shift/ReLU + 16 stores are issued one row
//     per group of 4 MFMAs INSIDE the first k group of the next tile -- no store burst, the matrix pipe never waits;
//   * the A ring is 8 k groups deep (16 K floats in flight per wave) and its loads are inline asm with hand-counted
//     `s_waitcnt vmcnt(28)`: hipcc drains `vmcnt(0)` at every loop header for loads it can see, i.e. once per tile
//     for the youngest prefetch AND the tile's stores.  The count is safe whatever else the wave has in flight: a
//     ring slot is consumed 8 k groups after it was requested, by then at least 7 x 4 younger ring loads exist, and
//     memory operations retire in issue order, so "at most 28 outstanding" implies the slot has landed; stores or
//     compiler-issued loads in between only make the wait stronger.  (Audit: tools/audit_bres_asm.py checks in the
//     disassembly that no instruction touches a ring register between its load and the wait that names it.)
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ACC = the slot lives in the accumulator half of the register file ("a": VMEM can target it and an MFMA reads its A
// operand from it).  Half of the ring's slots do: with all 128 ring registers in architectural VGPRs hipcc ran out of
// those and shuffled just-requested ring registers into AGPRs (copies of data that had not landed yet); with all of
// them in AGPRs, beside the two accumulator sets, it ran out of these instead.
template <int OFF, bool ACC>
__device__ __forceinline__ void ring_load(f32x4& dst, const float* p) {
  if (ACC) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
  else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}
template <int N, bool ACC>
__device__ __forceinline__ void ring_wait(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  if (ACC) asm volatile("s_waitcnt vmcnt(%4)" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "n"(N));
  else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}

// RES: the epilogue adds `res1` (identity / downsample sum of a Bottleneck, or the nearest-upsampled FPN top-down map):
// its 16 rows per tile are requested over the tile's k groups KG-8 .. KG-5 -- asm loads into AGPRs in the same in-order
// stream as the ring, released by hand-counted waits -- and consumed, like the accumulators, during the next tile's
// first k group.
// NI = 16-column accumulator tiles per wave (4: the 64-column wave tile).  (A 32-column build for K = 1024 slices was
// measured in round 3 -- 110 vs 106 TF alone, nothing in the step -- and removed in round 4.)
template <int WN, int KQ, bool PRO, bool RES, int NI = 4>
__global__ void __launch_bounds__(256, 1) bres2_kernel(const hnd_conv_desc d, const BresArgs a) {
  constexpr int WM = 4 / WN, WTN = 16 * NI, BN = WTN * WN, MI = 4, RING = 8;
  typedef float vecn __attribute__((ext_vector_type(NI)));
  static_assert(!RES || NI == 4, "the residual build keeps the 64-column wave tile");
  constexpr int K = 64 * KQ, KG = 4 * KQ;
  static_assert(KG >= RING, "the ring reaches at most one tile ahead");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                   // [BN][K]: chunk c of row r at position c ^ (r & 15)
  float* pro = Bs + BN * K;                           // [2][K] prologue scale, shift
  int* tabs = (int*)(pro + (PRO ? 2 * K : 0));        // [4 waves][2 tiles][2][64]: output / res1 pixel of a tile's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int l16 = lane & 15, g4 = lane >> 4;
  int* tab0 = tabs + wave * 256;

  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % a.nsl, tpx = per_xcd / a.nsl;
  const int team = xcd * tpx + idx / a.nsl, nteams = 8 * tpx;
  const int n0 = slice * BN;
  const int M = d.n * d.oh * d.ow;
  const int c_lo = (int)((long long)a.nchunks * team / nteams);
  const int c_hi = (int)((long long)a.nchunks * (team + 1) / nteams);

  auto a_ptr = [&](int m) -> const float* {
    m = m < M ? m : M - 1;
    const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
    const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
    const size_t pix = ((size_t)n_ * d.h + oh_ * (unsigned)d.sh) * (size_t)d.w_ + ow_ * (unsigned)d.sw;
    return d.x + pix * (size_t)d.cin + (size_t)(g4 * 4);
  };
  // output pixel of row m (-1 beyond M) and the pixel its residual is read from (mode 1: nearest-upsampled res1)
  auto out_pix = [&](int m, int& pr) -> int {
    pr = 0;
    if (m >= M) return -1;
    const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
    const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
    const int yr = (int)oh_ * d.y_sh + d.y_oh, yc = (int)ow_ * d.y_sw + d.y_ow;
    const int po = ((int)n_ * d.yh + yr) * d.yw + yc;
    pr = d.res1_mode == 1 ? ((int)n_ * d.res1_h + (yr * d.res1_h) / d.yh) * d.res1_w + (yc * d.res1_w) / d.yw : po;
    return po;
  };

  const int rho0 = n0 + wn * WTN;                     // first packed weight row of this wave's tiles
  const int col0 = (rho0 & ~63) + l16 * 4 + ((rho0 >> 4) & 3);      // hnd::chan_of_row: NI consecutive channels
  float es[NI], eb[NI], s1[NI], s2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    es[ni] = d.epi_scale ? d.epi_scale[col0 + ni] : 1.f;
    eb[ni] = d.epi_shift ? d.epi_shift[col0 + ni] : 0.f;
    s1[ni] = 0.f;
    s2[ni] = 0.f;
  }
  const bool vec_ok = (d.ldc % NI == 0) && (((uintptr_t)d.y | (uintptr_t)d.res1) % (4 * NI) == 0);
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;
  if (PRO) {
    for (int k = tid; k < K; k += 256) {
      pro[k] = d.pro_scale[k];
      pro[K + k] = d.pro_shift[k];
    }
  }
  int bsw[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) bsw[u] = ((4 * u + g4) ^ l16) * 4;
  const float* Bw = Bs + (wn * WTN + l16) * K;

  int c = c_lo;
  while (c < c_hi) {
    const int grp = a.cpg > 0 ? c / a.cpg : 0;
    const int seg_hi = a.cpg > 0 ? min(c_hi, (grp + 1) * a.cpg) : c_hi;
    __syncthreads();
    slice_to_lds<BN, K, 256>(Bs, d.w + (size_t)grp * (size_t)d.w_group_stride + (size_t)n0 * K, tid);
    __syncthreads();

    int cc = c + wm;
    if (cc < seg_hi) {
      const float* aptr[MI];
      f32x4 ring[RING][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) aptr[mi] = a_ptr(cc * 64 + mi * 16 + l16);
      static_for<RING>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        static_for<MI>([&](auto I) __attribute__((always_inline)) {
          ring_load<u * 64, (u & 1) != 0>(ring[u][decltype(I)::value], aptr[decltype(I)::value]);
        });
      });
      f32x4 out[MI][NI];                 // the finished tile, stored during the next tile's first k group
      f32x4 resv[MI][4];                 // ... and its residual rows (RES), requested during its last k group
      bool have_out = false;
      int par = 0;                       // row tables alternate: the previous tile's are read while this one's are written
      for (; cc < seg_hi; cc += WM, par ^= 1) {
        int* rowoff = tab0 + par * 128;             // this tile
        const int* prow = tab0 + (par ^ 1) * 128;   // the previous one
        const bool full = vec_ok && (cc * 64 + 64 <= M);
        {
          int pr;
          const int po = out_pix(cc * 64 + lane, pr);
          rowoff[lane] = po;
          rowoff[64 + lane] = pr;
          __builtin_amdgcn_wave_barrier();
        }
        const int cn = cc + WM < seg_hi ? cc + WM : cc;
        const float* nptr[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) nptr[mi] = a_ptr(cn * 64 + mi * 16 + l16);
        f32x4 acc[MI][NI];
        f32x4 bcur[NI], bnxt[NI], ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f}, psn = ps, pbn = pb;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bcur[ni] = *(const f32x4*)(Bw + ni * 16 * K + bsw[0]);
        if (PRO) {
          ps = *(const f32x4*)(pro + g4 * 4);
          pb = *(const f32x4*)(pro + K + g4 * 4);
        }
        static_for<KG>([&](auto G) __attribute__((always_inline)) {
          constexpr int g = decltype(G)::value, slot = g % RING;
          constexpr int kn = g + 1 < KG ? g + 1 : 0;
          const int bo = (kn & ~3) * 16 + bsw[kn & 3];
          ring_wait<4 * (RING - 1), (slot & 1) != 0>(ring[slot][0], ring[slot][1], ring[slot][2], ring[slot][3]);
          static_for<MI>([&](auto MIc) __attribute__((always_inline)) {
            constexpr int mi = decltype(MIc)::value;
            f32x4 av = ring[slot][mi];
            if (PRO) {
              av = av * ps + pb;
              av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor);
              av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    av[s], bcur[ni][s], (g == 0 && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
              if (mi == 0) {
                if (s < NI) bnxt[s] = *(const f32x4*)(Bw + s * 16 * K + bo);
                if (PRO && s == 3) {
                  psn = *(const f32x4*)(pro + kn * 16 + g4 * 4);
                  pbn = *(const f32x4*)(pro + K + kn * 16 + g4 * 4);
                }
              }
              // the previous tile's residual rows of row group mi: requested at its k group KG-8+mi; since then the
              // wave has issued 3 + 8 (3 - mi) + 16 + mi loads, so "at most 43 - 7 mi outstanding" means they have landed
              if constexpr (RES && g == 0) {
                if (s == 0) ring_wait<43 - 7 * mi, true>(resv[mi][0], resv[mi][1], resv[mi][2], resv[mi][3]);
              }
              if (g == 0 && have_out) {              // row 4*g4 + s of row group mi of the previous tile
                vecn v;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                  float x = out[mi][ni][s] * es[ni] + eb[ni];
                  if (RES) x += resv[mi][s][ni];
                  v[ni] = d.relu ? fmaxf(x, 0.f) : x;
                }
                const size_t yo = (size_t)(unsigned)prow[mi * 16 + 4 * g4 + s] * (unsigned)d.ldc + col0;
                *(vecn*)(d.y + yo) = v;
                if (NI == 4 && d.mask_out)          // ReLU-mask nibble of the stored values (hnd_conv_desc.mask_out)
                  d.mask_out[yo >> 2] = (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) |
                                                  (v[3] > 0.f ? 8 : 0));
              }
              // this tile's residual rows, consumed one tile later: four per k group over k groups KG-8 .. KG-5, as asm
              // loads in the ring's own in-order stream.  (As compiler-visible loads hipcc waited for them with counts
              // that ignore the asm loads in flight -- `vmcnt(3)` drained the whole ring behind them: 104 TF against
              // the tiled kernel's 112.)  Always issued, whatever the tile: the hand-counted waits below need every
              // wave to have the same number of loads in flight (rows beyond M read pixel 0).  They land in the
              // accumulator half of the register file: into architectural VGPRs hipcc copied the (not yet landed)
              // values to AGPRs on the spot.
              if constexpr (RES && g >= KG - 8 && g < KG - 4 && mi == 1) {
                constexpr int q = g - (KG - 8);
                ring_load<0, true>(resv[q][s], d.res1 + (size_t)(unsigned)rowoff[64 + q * 16 + 4 * g4 + s] *
                                                               (unsigned)d.ldc + col0);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            // refill this slot's row group for the k group RING ahead (in this tile or the wave's next one)
            if (g + RING < KG) ring_load<(g + RING) * 64, (slot & 1) != 0>(ring[slot][mi], aptr[mi]);
            else ring_load<(g + RING - KG) * 64, (slot & 1) != 0>(ring[slot][mi], nptr[mi]);
            __builtin_amdgcn_sched_barrier(0);
          });
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bcur[ni] = bnxt[ni];
          if (PRO) { ps = psn; pb = pbn; }
        });
        // ---- hand the tile over: accumulators -> `out`
        {
          if (full) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) out[mi][ni] = acc[mi][ni];
            have_out = true;
          } else {                                     // ragged / unaligned tile: checked path, not deferred
            hnd::epilogue_tile<MI, NI>(d, acc, rowoff, rowoff + 64, 4 * g4, col0, es, eb, s1, s2, false);
            have_out = false;
          }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aptr[mi] = nptr[mi];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (unused) prefetches land before reuse
      if (have_out) {                                    // the segment's last tile (its tables: the parity just left)
        const int* prow = tab0 + (par ^ 1) * 128;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            vecn v;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              float x = out[mi][ni][s] * es[ni] + eb[ni];
              if (RES) x += resv[mi][s][ni];
              v[ni] = d.relu ? fmaxf(x, 0.f) : x;
            }
            const size_t yo = (size_t)(unsigned)prow[mi * 16 + 4 * g4 + s] * (unsigned)d.ldc + col0;
            *(vecn*)(d.y + yo) = v;
            if (NI == 4 && d.mask_out)
              d.mask_out[yo >> 2] = (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) |
                                              (v[3] > 0.f ? 8 : 0));
          }
      }
    }
    c = seg_hi;
  }
}

int bres_kmax() {
  const char* e = getenv("HND_BRES");          // 0 = off, else the largest K taken (read per call: in-process A/B)
  return e ? atoi(e) : 512;
}

int cu_count() {
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

template <int WN, int KQ, bool PRO>
int launch_bres_t(const hnd_conv_desc& d, const BresArgs& a, size_t lds, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = bres_kernel<WN, KQ, PRO>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(bres<%d,%d>) failed: %s", WN, KQ, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, d, a);
  return hnd::check_launch("hnd_conv2d_igemm(bres)");
}

template <int WN, int KQ>
int launch_bres_p(const hnd_conv_desc& d, const BresArgs& a, size_t lds, int grid, hipStream_t stream) {
  return d.pro_scale ? launch_bres_t<WN, KQ, true>(d, a, lds, grid, stream)
                     : launch_bres_t<WN, KQ, false>(d, a, lds, grid, stream);
}

template <int WN, int KQ, bool PRO, bool RES>
int launch_bres2_t(const hnd_conv_desc& d, const BresArgs& a, size_t lds, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = bres2_kernel<WN, KQ, PRO, RES>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(bres2<%d,%d>) failed: %s", WN, KQ, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, d, a);
  return hnd::check_launch("hnd_conv2d_igemm(bres2)");
}

template <int WN, int KQ>
int launch_bres2_p(const hnd_conv_desc& d, const BresArgs& a, size_t lds, int grid, hipStream_t stream) {
  if (d.res1) return launch_bres2_t<WN, KQ, false, true>(d, a, lds, grid, stream);      // (never with a prologue)
  return d.pro_scale ? launch_bres2_t<WN, KQ, true, false>(d, a, lds, grid, stream)
                     : launch_bres2_t<WN, KQ, false, false>(d, a, lds, grid, stream);
}

}  // namespace

namespace hnd {

// 0 = not taken; 1 / 2 = the 8-wave kernel with a 64- / 128-column weight slice (K <= 512 / 256); 3 / 4 = the
// one-wave-per-SIMD kernel (bres2) with a 64- / 128-column slice, for epilogues without a mask operand
int bres_variant(const hnd_conv_desc& d) {
  const int kmax = bres_kmax();
  if (kmax <= 0 || d.kh != 1 || d.kw != 1 || d.bh != 0 || d.bw != 0 || d.stats != nullptr) return 0;
  if (d.cin != d.kdim) return 0;
  if (d.w_group_rows % 64 != 0) return 0;
  if (d.kdim > kmax) return 0;
  const int wn = (d.kdim <= 256 && d.cout % 128 == 0) ? 2 : 1;
  // instantiated depths: 64 / 128 / 256 with the 128-column slice, 256 / 512 with the 64-column slice
  if (wn == 2 ? (d.kdim != 64 && d.kdim != 128 && d.kdim != 256) : (d.kdim != 256 && d.kdim != 512)) return 0;
  const int bn = 64 * wn;
  if (d.cout % bn != 0) return 0;
  const int per_xcd = cu_count() / 8, nsl = d.cout / bn;
  if (per_xcd < 1 || nsl > per_xcd || per_xcd % nsl != 0) return 0;
  if ((long long)(d.oh - 1) * d.sh >= d.h || (long long)(d.ow - 1) * d.sw >= d.w_) return 0;
  const long long M = (long long)d.n * d.oh * d.ow;
  const long long nchunks = (M + 63) / 64, nteams = 8ll * (per_xcd / nsl);
  const long long per_team = nchunks / nteams;
  const bool plain = !d.res2 && !d.mask && !d.mask_bits && !(d.res1 && d.pro_scale);  // one-wave kernel: optional res1 only
  const char* v2 = getenv("HND_BRES2");                   // 0 = never the one-wave kernel (A/B)
  const bool all = hnd::debug_picker("bres_all") > 0;     // every eligible launch, not only where it was measured to win
  if (plain && d.kdim >= 128 && !(v2 && atoi(v2) == 0) && per_team >= 2ll * (4 / wn)) {
    // measured (profiles/r03_bres_vs_tiled.txt): the one-wave kernel wins on long runs of chunks; its prologue form
    // (8 VALU per A fragment beside a single wave's MFMAs) does not, and K = 512 needs >= 48 chunks per team
    const bool spills = (d.pro_scale || d.res1) && d.kdim == 512;      // those two builds do not fit 512 registers
    if (all && !spills) return 2 + wn;
    // the residual build (RES): its rows travel as asm loads in the ring's own in-order stream with hand-counted
    // waits (as compiler-visible loads they drained the ring once per tile and lost to the tiled kernel, 104 vs 112 TF).
    // Measured (tools/bench_bres.py): 256->256 @200x336 + upsampled residual 109 -> 123 TF, 128->512 @100x168 + res
    // 95 -> 102, 256->1024 @50x84 + res 112 -> 116.  K = 512 with a residual spills and stays on the 8-wave kernel.
    if (!d.pro_scale && !spills && per_team >= (d.w_group_rows > 0 || d.kdim == 512 ? 48 : 8)) return 2 + wn;
  }
  if (per_team < 2ll * (8 / wn)) return 0;              // every wave row gets at least two chunks
  if (!all) {
    // where the free-running waves beat the tiled kernel (profiles/r03_bres_vs_tiled.txt, both with the specialised
    // epilogue): long runs of chunks per weight slice.  A Winograd launch whose components are short reloads its
    // slice every few chunks; an epilogue with residual / mask loads drains the wave's prefetch ring (one wave cannot
    // hold both in 256 registers), which only the K = 512 launches amortise.
    if (d.w_group_rows > 0 && per_team < 64) return 0;
    if ((d.res1 || d.res2 || d.mask || d.mask_bits) && d.kdim < 512) return 0;
    if (d.kdim == 128 && d.cout >= 512) return 0;
  }
  return wn;
}

int launch_bres(const hnd_conv_desc& d, hipStream_t stream) {
  const int var = bres_variant(d);
  if (var == 0) {
    set_error("launch_bres: descriptor not eligible");
    return HND_ERR_INVALID;
  }
  const bool v2 = var > 2;
  const int wn = v2 ? var - 2 : var;
  const long long M = (long long)d.n * d.oh * d.ow;
  BresArgs a;
  a.div_ow = make_fastdiv((unsigned)d.ow);
  a.div_oh = make_fastdiv((unsigned)d.oh);
  a.nsl = d.cout / (64 * wn);
  a.nchunks = (int)((M + 63) / 64);
  a.cpg = d.w_group_rows / 64;
  const bool pro = d.pro_scale != nullptr;
  const size_t lds = ((size_t)64 * wn * d.kdim + (pro ? 2 * (size_t)d.kdim : 0) + 8 * 256) * sizeof(float);
  const int grid = (cu_count() / 8) * 8;
  if (v2) {
    if (wn == 2) return d.kdim == 128 ? launch_bres2_p<2, 2>(d, a, lds, grid, stream)
                                      : launch_bres2_p<2, 4>(d, a, lds, grid, stream);
    return d.kdim == 256 ? launch_bres2_p<1, 4>(d, a, lds, grid, stream) : launch_bres2_p<1, 8>(d, a, lds, grid, stream);
  }
  if (wn == 2) {
    if (d.kdim == 64) return launch_bres_p<2, 1>(d, a, lds, grid, stream);
    if (d.kdim == 128) return launch_bres_p<2, 2>(d, a, lds, grid, stream);
    return launch_bres_p<2, 4>(d, a, lds, grid, stream);
  }
  if (d.kdim == 256) return launch_bres_p<1, 4>(d, a, lds, grid, stream);
  return launch_bres_p<1, 8>(d, a, lds, grid, stream);
}

}  // namespace hnd
