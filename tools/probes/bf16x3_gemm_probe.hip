// PROBE (not product code): can the bf16 matrix pipe of gfx950 stand in for fp32 MFMA?  VERDICT r4 item 3.
//
// C[M][256] = A[M][256] * W^T, W [256][256], fp32 in HBM, the fpn.inner0 shape (M = 16 x 200 x 336 = 1 075 200).
// Every fp32 value is split IN REGISTERS into three bf16 planes by truncation -- hi = top 16 bits, mid = top 16 bits of
// (x - hi), lo = x - hi - mid: 8 + 8 + 8 significant bits, x == hi + mid + lo exactly -- and the six products with
// i + j <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) go to v_mfma_f32_16x16x32_bf16 with fp32 accumulation,
// smallest terms first inside every 32-deep k step.  The three dropped products are <= 2^-24 of the result each.
//
// What is measured: (1) TFLOP/s-equivalent (2 M N K / time) of this kernel, of the same kernel with the split's vector
// work removed (planes = the raw top halves: the MFMA-side ceiling of this structure), and of the library's native
// fp32-MFMA kernel (libhnd_hip.so, hnd_conv2d_igemm -> bres2) on the same operands; (2) error vs an fp64 GEMM (rel-L2,
// max |d| / rms, max elementwise relative error where |ref| > 1e-3 rms) of both, on N(0,1) data, wide-dynamic-range rows,
// denormal inputs, and Inf / NaN inputs (which output elements are non-finite).
//
// Kernel structure (deliberately plain: compiler-scheduled, no asm ring): one workgroup of 8 waves per CU; its 64-column
// weight slice lives in LDS for the whole launch as three pre-split bf16 planes (96 KB, XOR-swizzled 16-byte chunks);
// a wave owns 64-row chunks, reads its A fragments straight from global memory (lane = row l16 of a 16-row group, 8
// consecutive k: two 16-byte loads per row group and k step, next step prefetched), splits them, and issues 96 MFMAs per
// k step for its 64 x 64 tile.
//
// build: hipcc -O3 --offload-arch=gfx950 -I include tools/probes/bf16x3_gemm_probe.hip \
//            -L hnd_ghnd_object_detectors_amd -lhnd_hip -Wl,-rpath,$PWD/hnd_ghnd_object_detectors_amd -o /tmp/bf16x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "hnd_hip.h"

extern "C" int hnd_pack_weights(const float* w, float* dst, int cout, int cin, int kh, int kw, int transposed, int chan_pad,
                                int i0, int istep, int ni, int j0, int jstep, int nj, void* stream);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

#ifndef KS_UNROLL
#define KS_UNROLL 2          // k steps unrolled per loop trip (8 = all: 256 registers spill at two waves per SIMD)
#endif
#ifndef NWAVES
#define NWAVES 8             // waves per workgroup (one workgroup per CU): 8 = two per SIMD (256 registers), 4 = one (512)
#endif
constexpr int K = 256, N = 256, BN = 64, NSL = N / BN, NT = 64 * NWAVES;
constexpr int PLANE = BN * K;                     // bf16 elements per plane of a slice

// split mode: 0 = truncation split (exact), 1 = + Inf/NaN guard, 2 = NO split (all planes = top halves; timing only)
template <int MODE>
__device__ __forceinline__ void split8(const f4& x0, const f4& x1, bf8& h, bf8& m, bf8& l) {
  uint32_t xs[8], hs[8], ms[8], ls[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // (through a scalar: __builtin_bit_cast applied to a vector ELEMENT expression made hipcc 7.2 use element 0 for every i)
    const float f0 = x0[i], f1 = x1[i];
    xs[i] = __float_as_uint(f0);
    xs[4 + i] = __float_as_uint(f1);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (MODE == 2) {
      hs[i] = ms[i] = ls[i] = xs[i];
      continue;
    }
    const uint32_t hb = xs[i] & 0xffff0000u;
    const float r1 = __uint_as_float(xs[i]) - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    hs[i] = hb;
    ms[i] = mb;
    ls[i] = __float_as_uint(r2);
    if (MODE == 1) {
      // Inf - Inf = NaN would turn an infinite input into NaN outputs, and a NaN whose payload sits in the low 16 bits
      // would truncate to Inf: non-finite inputs keep hi = the value (NaN made quiet), mid = lo = 0
      const bool nonfinite = (xs[i] & 0x7f800000u) == 0x7f800000u;
      const bool isnan_ = nonfinite && (xs[i] & 0x007fffffu) != 0;
      hs[i] = isnan_ ? 0x7fc00000u : hs[i];
      ms[i] = nonfinite ? 0u : ms[i];
      ls[i] = nonfinite ? 0u : ls[i];
    }
  }
  u4 hp, mp, lp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // two top halves per dword: element 2i in the low half
    hp[i] = __builtin_amdgcn_perm(hs[2 * i + 1], hs[2 * i], 0x07060302u);
    mp[i] = __builtin_amdgcn_perm(ms[2 * i + 1], ms[2 * i], 0x07060302u);
    lp[i] = __builtin_amdgcn_perm(ls[2 * i + 1], ls[2 * i], 0x07060302u);
  }
  h = __builtin_bit_cast(bf8, hp);
  m = __builtin_bit_cast(bf8, mp);
  l = __builtin_bit_cast(bf8, lp);
}

// wimg: per slice the LDS image [3 planes][64 rows][32 chunks of 8 bf16], chunk c of row r at position c ^ (r & 15);
// LDS row ni * 16 + l16 holds output channel n0 + l16 * 4 + ni (a lane's four tiles = four consecutive channels)
template <int MODE>
__global__ void __launch_bounds__(NT, 1) bf16x3_kernel(const float* __restrict__ A, const uint16_t* __restrict__ wimg,
                                                        float* __restrict__ C, int M) {
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];       // [3][64][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % NSL, tpx = per_xcd / NSL;
  const int team = xcd * tpx + idx / NSL, nteams = 8 * tpx;
  const int nchunks = (M + 63) / 64;
  const int c_lo = (int)((long long)nchunks * team / nteams), c_hi = (int)((long long)nchunks * (team + 1) / nteams);
  {
    const u4* src = (const u4*)(wimg + (size_t)slice * 3 * PLANE);
    u4* dst = (u4*)Bs;
    for (int i = tid; i < 3 * PLANE / 8; i += NT) dst[i] = src[i];
  }
  __syncthreads();
  const int n0 = slice * BN;
  for (int cc = c_lo + wave; cc < c_hi; cc += NWAVES) {
    const float* ap[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      int m = cc * 64 + mi * 16 + l16;
      m = m < M ? m : M - 1;
      ap[mi] = A + (size_t)m * K + g4 * 8;
    }
    f4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    f4 cur[4][2], nxt[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      cur[mi][0] = *(const f4*)(ap[mi]);
      cur[mi][1] = *(const f4*)(ap[mi] + 4);
    }
#pragma unroll KS_UNROLL
    for (int ks = 0; ks < K / 32; ++ks) {
      if (ks + 1 < K / 32) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          nxt[mi][0] = *(const f4*)(ap[mi] + (ks + 1) * 32);
          nxt[mi][1] = *(const f4*)(ap[mi] + (ks + 1) * 32 + 4);
        }
      }
      bf8 ah[4], am[4], al[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) split8<MODE>(cur[mi][0], cur[mi][1], ah[mi], am[mi], al[mi]);
      const int pos = ((ks * 4 + g4) ^ l16) * 8;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const uint16_t* br = Bs + (ni * 16 + l16) * K + pos;
        const bf8 bh = *(const bf8*)(br), bm = *(const bf8*)(br + PLANE), bl = *(const bf8*)(br + 2 * PLANE);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          f4 c = acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, c, 0, 0, 0);      // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mi], bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mi], bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, c, 0, 0, 0);
          acc[mi][ni] = c;
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        cur[mi][0] = nxt[mi][0];
        cur[mi][1] = nxt[mi][1];
      }
    }
    // C/D layout: column = lane & 15 (-> channel n0 + l16 * 4 + ni), row = g4 * 4 + reg of the 16-row group
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = cc * 64 + mi * 16 + g4 * 4 + r;
        if (m < M) *(f4*)(C + (size_t)m * N + n0 + l16 * 4) = f4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// v2: the structure of the library's one-wave-per-SIMD fp32 kernel (conv_bres.hip, bres2) carried over: 4 waves per
// workgroup (512 registers per lane), A fragments through a counted inline-asm register ring 4 k steps deep that runs
// across tile boundaries (hipcc drains every load it can see at a loop header: the plain kernel above overlaps nothing),
// and the split of k step s+1 software-pipelined into the MFMAs of k step s: per group of 6 MFMAs (one accumulator tile:
// 96 matrix-pipe cycles, 48 of them free for vector issue) one pair of elements is split (11 vector instructions).
template <int N_, class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N_, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl<N_>(f, std::make_integer_sequence<int, N_>{}); }

template <int OFF, bool ACC>
__device__ __forceinline__ void rload(f4& dst, const float* p) {
  if (ACC) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
  else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}
template <int CNT, bool ACC>
__device__ __forceinline__ void rwait(f4& a0, f4& a1, f4& a2, f4& a3, f4& a4, f4& a5, f4& a6, f4& a7) {
  if (ACC) asm volatile("s_waitcnt vmcnt(%8)" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7) : "n"(CNT));
  else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "n"(CNT));
}

// one pair of fp32 values -> one dword of each plane (used for the very first k step only)
template <int MODE>
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hp, uint32_t& mp, uint32_t& lp) {
  const uint32_t a0 = __float_as_uint(x0), a1 = __float_as_uint(x1);
  if (MODE == 2) { hp = mp = lp = __builtin_amdgcn_perm(a1, a0, 0x07060302u); return; }
  const uint32_t h0 = a0 & 0xffff0000u, h1 = a1 & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
  const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
  hp = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  mp = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  lp = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// STORES = the previous tile's 16 row stores were issued before this tile's first refill: they sit in the in-order
// memory stream between the slots the first three k steps wait for and the youngest refills (exact counts: full tiles only)
// which ring slots live in the accumulator half of the register file (VMEM can target it; vector instructions reach it
// through v_accvgpr_read).  With half of the ring in architectural registers hipcc sat at 256 of them and moved
// just-requested ring registers away before their wait (garbage results): all four slots in AGPRs.
#ifndef RING_ACC_MASK
#define RING_ACC_MASK 0xf
#endif
#define RACC(slot_) (((RING_ACC_MASK >> (slot_)) & 1) != 0)
template <int MODE>
__global__ void __launch_bounds__(256, 1) bf16x3_ring_kernel(const float* __restrict__ A, const uint16_t* __restrict__ wimg,
                                                             float* __restrict__ C, int M) {
  constexpr int RING = 4, KS = K / 32, MI = 4, NI = 4;
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % NSL, tpx = per_xcd / NSL;
  const int team = xcd * tpx + idx / NSL, nteams = 8 * tpx;
  const int nchunks = M / 64;                           // (the host checks M % 64 == 0)
  const int c_lo = (int)((long long)nchunks * team / nteams), c_hi = (int)((long long)nchunks * (team + 1) / nteams);
  {
    const u4* src = (const u4*)(wimg + (size_t)slice * 3 * PLANE);
    u4* dst = (u4*)Bs;
    for (int i = tid; i < 3 * PLANE / 8; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const int n0 = slice * BN;
  auto a_ptr = [&](int cc, int mi) -> const float* { return A + (size_t)(cc * 64 + mi * 16 + l16) * K + g4 * 8; };
  int cc = c_lo + wave;
  if (cc >= c_hi) return;
  const float* aptr[MI];
  f4 ring[RING][MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) aptr[mi] = a_ptr(cc, mi);
  sfor<RING>([&](auto U) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value;
    sfor<MI>([&](auto I) __attribute__((always_inline)) {
      constexpr int mi = decltype(I)::value;
      rload<u * 128, RACC(u)>(ring[u][mi][0], aptr[mi]);
      rload<u * 128 + 16, RACC(u)>(ring[u][mi][1], aptr[mi]);
    });
  });
  uint32_t pl[2][3][MI][4];                 // [parity][hi / mid / lo][row group]: 4 dwords = 8 bf16 each
  rwait<8 * (RING - 1), RACC(0)>(ring[0][0][0], ring[0][0][1], ring[0][1][0], ring[0][1][1], ring[0][2][0], ring[0][2][1],
                               ring[0][3][0], ring[0][3][1]);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 v = ring[0][mi][j >> 1];
      const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
      split_pair<MODE>(x0, x1, pl[0][0][mi][j], pl[0][1][mi][j], pl[0][2][mi][j]);
    }
  bool stored = false;
  f4 acc[MI][NI];
  for (; cc < c_hi; cc += 4) {
    const int cn = cc + 4 < c_hi ? cc + 4 : cc;
    const float* nptr[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) nptr[mi] = a_ptr(cn, mi);
    sfor<KS>([&](auto G) __attribute__((always_inline)) {
      constexpr int ks = decltype(G)::value, slot = ks % RING, par = ks & 1;
      constexpr int slot1 = (ks + 1) % RING;            // the step whose planes are made during this one
      // slot `slot` was split during the previous step: refill it for the step RING ahead
      sfor<MI>([&](auto I) __attribute__((always_inline)) {
        constexpr int mi = decltype(I)::value;
        if constexpr (ks + RING < KS) {
          rload<(ks + RING) * 128, RACC(slot)>(ring[slot][mi][0], aptr[mi]);
          rload<(ks + RING) * 128 + 16, RACC(slot)>(ring[slot][mi][1], aptr[mi]);
        } else {
          rload<(ks + RING - KS) * 128, RACC(slot)>(ring[slot][mi][0], nptr[mi]);
          rload<(ks + RING - KS) * 128 + 16, RACC(slot)>(ring[slot][mi][1], nptr[mi]);
        }
      });
      // the next step's slot was requested RING - 1 steps ago: 8 (RING - 1) younger ring loads may still be in flight,
      // plus the previous tile's 16 stores while they are younger than it (k steps 0 .. 2 of every tile but the first)
      // The register-tied wait must be ONE statement on every path: tied waits in the two arms of a branch made hipcc merge
      // the ring registers with copies placed BEFORE the wait in one arm (reads of data still in flight).  So the tile
      // without stores ahead of it first waits untied for the smaller count.
      if constexpr (ks < RING - 1) {
        if (!stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (RING - 1)));
        rwait<8 * (RING - 1) + 16, RACC(slot1)>(ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1],
                                               ring[slot1][2][0], ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
      } else {
        rwait<8 * (RING - 1), RACC(slot1)>(ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1],
                                          ring[slot1][2][0], ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
      }
      const int pos = ((ks * 4 + g4) ^ l16) * 8;
      bf8 bcur[3], bnxt[3];
      {
        const uint16_t* br = Bs + l16 * K + pos;
        bcur[0] = *(const bf8*)(br); bcur[1] = *(const bf8*)(br + PLANE); bcur[2] = *(const bf8*)(br + 2 * PLANE);
      }
      sfor<NI>([&](auto NIc) __attribute__((always_inline)) {
        constexpr int ni = decltype(NIc)::value;
        if constexpr (ni + 1 < NI) {
          const uint16_t* br = Bs + ((ni + 1) * 16 + l16) * K + pos;
          bnxt[0] = *(const bf8*)(br); bnxt[1] = *(const bf8*)(br + PLANE); bnxt[2] = *(const bf8*)(br + 2 * PLANE);
        }
        sfor<MI>([&](auto MIc) __attribute__((always_inline)) {
          constexpr int mi = decltype(MIc)::value;
          auto frag = [&](int q) __attribute__((always_inline)) {
            const u4 t = {pl[par][q][mi][0], pl[par][q][mi][1], pl[par][q][mi][2], pl[par][q][mi][3]};
            return __builtin_bit_cast(bf8, t);
          };
          const bf8 ah = frag(0), am = frag(1), al = frag(2);
          // one pair of the NEXT step's elements rides between this tile's six MFMAs: piece p = ni * 4 + mi -> row group
          // p / 4, pair p % 4
          constexpr int p = ni * 4 + mi, rg = p >> 2, j = p & 3;
          const f4 v = ring[slot1][rg][j >> 1];
          const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
          f4 c = (ks == 0) ? f4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bcur[0], c, 0, 0, 0);      // smallest terms first
          const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[2], c, 0, 0, 0);
          const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[1], c, 0, 0, 0);
          const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[0], c, 0, 0, 0);
          const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[1], c, 0, 0, 0);
          if (MODE == 2) {
            pl[par ^ 1][0][rg][j] = pl[par ^ 1][1][rg][j] = pl[par ^ 1][2][rg][j] =
                __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
          } else {
            pl[par ^ 1][0][rg][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
            pl[par ^ 1][1][rg][j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
            pl[par ^ 1][2][rg][j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
          }
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[0], c, 0, 0, 0);
          acc[mi][ni] = c;
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (ni + 1 < NI) { bcur[0] = bnxt[0]; bcur[1] = bnxt[1]; bcur[2] = bnxt[2]; }
      });
    });
    // the tile's 16 row stores (full tiles only); the next tile's first refills come after them in the memory stream
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = cc * 64 + mi * 16 + g4 * 4 + r;
        *(f4*)(C + (size_t)m * N + n0 + l16 * 4) = f4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
      }
    stored = true;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) aptr[mi] = nptr[mi];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void ref64_kernel(const float* A, const float* W, double* C, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i % N;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
  C[i] = s;
}

static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) * (1.0f / 16777216.0f); }
static float nrand() { float u1 = urand() + 1e-7f, u2 = urand(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

static void make_wimg(const std::vector<float>& W, std::vector<uint16_t>& img) {
  img.assign((size_t)NSL * 3 * PLANE, 0);
  for (int s = 0; s < NSL; ++s)
    for (int r = 0; r < BN; ++r) {
      const int ni = r >> 4, l16 = r & 15, ch = s * BN + l16 * 4 + ni;
      for (int k = 0; k < K; ++k) {
        uint32_t x;
        memcpy(&x, &W[(size_t)ch * K + k], 4);
        const uint32_t hb = x & 0xffff0000u;
        float xf, hf, mf;
        memcpy(&xf, &x, 4);
        memcpy(&hf, &hb, 4);
        const float r1 = xf - hf;
        uint32_t r1b;
        memcpy(&r1b, &r1, 4);
        const uint32_t mb = r1b & 0xffff0000u;
        memcpy(&mf, &mb, 4);
        const float r2 = r1 - mf;
        uint32_t lb;
        memcpy(&lb, &r2, 4);
        const int c = k >> 3, pos = c ^ (r & 15);
        const size_t o = (size_t)s * 3 * PLANE + (size_t)r * K + pos * 8 + (k & 7);
        img[o] = (uint16_t)(hb >> 16);
        img[o + PLANE] = (uint16_t)(mb >> 16);
        img[o + 2 * PLANE] = (uint16_t)(lb >> 16);
      }
    }
}

struct Err { double rel_l2, max_rms, max_rel; long nonfinite, nonfinite_ref, mismatch; };
static Err compare(const std::vector<float>& got, const std::vector<double>& ref) {
  double num = 0, den = 0, mx = 0, mxrel = 0;
  long nf = 0, nfr = 0, mis = 0;
  double ss = 0;
  long cnt = 0;
  for (size_t i = 0; i < ref.size(); ++i)
    if (std::isfinite(ref[i])) { ss += ref[i] * ref[i]; ++cnt; }
  const double rms = sqrt(ss / (cnt > 0 ? cnt : 1)) + 1e-300;
  for (size_t i = 0; i < ref.size(); ++i) {
    const bool fr = std::isfinite(ref[i]), fg = std::isfinite((double)got[i]);
    nf += !fg;
    nfr += !fr;
    if (fr != fg || (!fr && (std::isnan(ref[i]) != std::isnan((double)got[i])))) ++mis;
    if (!fr || !fg) continue;
    const double d = (double)got[i] - ref[i];
    num += d * d;
    den += ref[i] * ref[i];
    if (fabs(d) > mx) mx = fabs(d);
    if (fabs(ref[i]) > 1e-3 * rms && fabs(d) / fabs(ref[i]) > mxrel) mxrel = fabs(d) / fabs(ref[i]);
  }
  return {sqrt(num / (den + 1e-300)), mx / rms, mxrel, nf, nfr, mis};
}

int main(int argc, char** argv) {
  const int Mbig = 16 * 200 * 336, Mref = 8192;
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; shape M=%d N=%d K=%d (fpn.inner0), %d timed launches per kernel; build: %d waves per workgroup, "
         "k loop unrolled x%d\n", prop.gcnArchName, prop.multiProcessorCount, Mbig, N, K, iters, NWAVES, KS_UNROLL);
  const int grid = (prop.multiProcessorCount / 8) * 8;
  const size_t lds = 3 * PLANE * sizeof(uint16_t);
  CK(hipFuncSetAttribute((const void*)bf16x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bf16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bf16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bf16x3_ring_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bf16x3_ring_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static_assert((16 * 200 * 336) % 64 == 0 && 8192 % 64 == 0, "the ring kernel takes full 64-row tiles only");

  std::vector<float> W((size_t)N * K);
  for (auto& v : W) v = nrand() * 0.0625f;
  if (getenv("PROBE_IDENTITY"))                 // debugging aid: W = I, so C must equal A
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) W[(size_t)n * K + k] = n == k ? 1.f : 0.f;
  std::vector<uint16_t> img;
  make_wimg(W, img);
  float *dA, *dW, *dWp, *dC;
  uint16_t* dImg;
  double* dRef;
  CK(hipMalloc(&dA, (size_t)Mbig * K * 4));
  CK(hipMalloc(&dC, (size_t)Mbig * N * 4));
  CK(hipMalloc(&dW, W.size() * 4));
  CK(hipMalloc(&dWp, (size_t)N * K * 4));
  CK(hipMalloc(&dImg, img.size() * 2));
  CK(hipMalloc(&dRef, (size_t)Mref * N * 8));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dImg, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  if (hnd_pack_weights(dW, dWp, N, K, 1, 1, 0, K, 0, 1, 1, 0, 1, 1, nullptr)) { fprintf(stderr, "pack failed\n"); return 1; }

  auto native = [&](int n, int h, int w) {
    hnd_conv_desc d;
    memset(&d, 0, sizeof d);
    d.x = dA; d.w = dWp; d.y = dC;
    d.n = n; d.h = h; d.w_ = w; d.cin = K; d.oh = h; d.ow = w; d.yh = h; d.yw = w; d.cout = N; d.ldc = N;
    d.y_sh = d.y_sw = 1; d.kh = d.kw = 1; d.sh = d.sw = 1; d.dh = d.dw = 1; d.kdim = K;
    const int rc = hnd_conv2d_igemm(&d, nullptr);
    if (rc) { fprintf(stderr, "hnd_conv2d_igemm failed: %s\n", hnd_last_error_string()); exit(1); }
  };

  // ---------------------------------------------------------------- accuracy
  std::vector<float> A((size_t)Mref * K), got((size_t)Mref * N);
  std::vector<double> ref((size_t)Mref * N);
  const char* names[4] = {"N(0,1)", "wide range (rows x 2^U(-20,20), elements x 2^N(0,4))", "denormal inputs (|x| ~ 1e-40 ... 1e-38)",
                          "Inf / NaN inputs (1 Inf, 1 -Inf, 1 NaN row entries)"};
  printf("\n%-58s %-22s %10s %10s %10s %9s\n", "data set", "kernel", "rel-L2", "max|d|/rms", "max rel", "non-finite (got / fp64 ref / mismatched)");
  for (int ds = 0; ds < 4; ++ds) {
    rng_state = 777u + ds;
    for (int m = 0; m < Mref; ++m) {
      const float rowscale = ds == 1 ? exp2f(urand() * 40.f - 20.f) : 1.f;
      for (int k = 0; k < K; ++k) {
        float v = nrand();
        if (ds == 1) v *= rowscale * exp2f(nrand() * 4.f);
        if (ds == 2) v *= (m & 1) ? 1e-38f : 1e-40f;
        A[(size_t)m * K + k] = v;
      }
    }
    if (ds == 3) {
      A[(size_t)5 * K + 7] = INFINITY;
      A[(size_t)9 * K + 100] = -INFINITY;
      uint32_t nanbits = 0x7f800001u;              // a NaN whose payload lives in the low bits
      memcpy(&A[(size_t)17 * K + 33], &nanbits, 4);
    }
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    ref64_kernel<<<(Mref * N + 255) / 256, 256>>>(dA, dW, dRef, Mref);
    CK(hipMemcpy(ref.data(), dRef, ref.size() * 8, hipMemcpyDeviceToHost));
    for (int kern = 0; kern < 4; ++kern) {
      CK(hipMemset(dC, 0xff, (size_t)Mref * N * 4));
      if (kern == 0) native(1, 64, 128);
      else if (kern == 1) bf16x3_kernel<0><<<grid, NT, lds>>>(dA, dImg, dC, Mref);
      else if (kern == 2) bf16x3_kernel<1><<<grid, NT, lds>>>(dA, dImg, dC, Mref);
      else bf16x3_ring_kernel<0><<<grid, 256, lds>>>(dA, dImg, dC, Mref);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
      const Err e = compare(got, ref);
      if (getenv("PROBE_DEBUG") && ds == 0 && kern == atoi(getenv("PROBE_DEBUG"))) {
        // where the error lives: by 64-column slice, by channel % 4 (= accumulator tile ni), by 16-row group (mi)
        double es[4] = {0}, en[4] = {0}, em[4] = {0}, rs[4] = {0}, rn[4] = {0}, rm[4] = {0};
        for (int m = 0; m < 256; ++m)
          for (int n = 0; n < N; ++n) {
            const double d = got[(size_t)m * N + n] - ref[(size_t)m * N + n], r2 = ref[(size_t)m * N + n] * ref[(size_t)m * N + n];
            es[n / 64] += d * d; rs[n / 64] += r2;
            en[n % 4] += d * d; rn[n % 4] += r2;
            em[(m % 64) / 16] += d * d; rm[(m % 64) / 16] += r2;
          }
        for (int i = 0; i < 4; ++i)
          printf("  debug: slice %d rel %.2e | ni %d rel %.2e | mi %d rel %.2e\n", i, sqrt(es[i] / rs[i]), i, sqrt(en[i] / rn[i]), i,
                 sqrt(em[i] / rm[i]));
        if (getenv("PROBE_IDENTITY"))
          for (int m = 0; m < 3; ++m) {
            printf("  row %d: C[m][n] == A[m'][k'] at:", m);
            for (int n = 0; n < 12; ++n) {
              int hit = -1;
              for (int q = 0; q < 64 * K && hit < 0; ++q)
                if (A[q] == got[(size_t)m * N + n]) hit = q;
              printf(" (%d,%d)", hit < 0 ? -1 : hit / K, hit < 0 ? -1 : hit % K);
            }
            printf("\n");
          }
        for (int m = 0; m < 2; ++m) {
          printf("  row %d got:", m);
          for (int n = 0; n < 8; ++n) printf(" %9.5f", got[(size_t)m * N + n]);
          printf("\n  row %d ref:", m);
          for (int n = 0; n < 8; ++n) printf(" %9.5f", ref[(size_t)m * N + n]);
          printf("\n");
        }
      }
      printf("%-58s %-22s %10.2e %10.2e %10.2e   %ld / %ld / %ld\n", names[ds],
             kern == 0 ? "native fp32 MFMA" : (kern == 1 ? "bf16x3" : (kern == 2 ? "bf16x3 + Inf/NaN guard" : "bf16x3 v2 (asm ring)")),
             e.rel_l2, e.max_rms, e.max_rel,
             e.nonfinite, e.nonfinite_ref, e.mismatch);
    }
  }

  // ---------------------------------------------------------------- timing at the full shape
  {
    std::vector<float> Ab((size_t)Mbig * K);
    rng_state = 4242u;
    for (auto& v : Ab) v = nrand();
    CK(hipMemcpy(dA, Ab.data(), Ab.size() * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * Mbig * N * K;
  printf("\n%-44s %10s %14s\n", "kernel (M = 1 075 200)", "ms", "TFLOP/s-equiv");
  double t_native = 0;
  for (int kern = 0; kern < 6; ++kern) {
    auto run = [&]() {
      if (kern == 0) native(16, 200, 336);
      else if (kern == 1) bf16x3_kernel<0><<<grid, NT, lds>>>(dA, dImg, dC, Mbig);
      else if (kern == 2) bf16x3_kernel<1><<<grid, NT, lds>>>(dA, dImg, dC, Mbig);
      else if (kern == 3) bf16x3_kernel<2><<<grid, NT, lds>>>(dA, dImg, dC, Mbig);
      else if (kern == 4) bf16x3_ring_kernel<0><<<grid, 256, lds>>>(dA, dImg, dC, Mbig);
      else bf16x3_ring_kernel<2><<<grid, 256, lds>>>(dA, dImg, dC, Mbig);
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    if (kern == 0) t_native = ms;
    printf("%-44s %10.4f %14.1f   (x%.2f native)\n",
           kern == 0 ? "native fp32 MFMA (libhnd_hip.so, bres2)" : kern == 1 ? "bf16x3, split in registers"
           : kern == 2 ? "bf16x3 + Inf/NaN guard" : kern == 3 ? "bf16x3 MFMAs without the split's VALU work"
           : kern == 4 ? "bf16x3 v2: asm ring, split between the MFMAs" : "bf16x3 v2 without the split's VALU work",
           ms, flop / ms / 1e9, t_native / ms);
  }
  printf("\nsplit cost per A fragment (static): per fp32 element 2 v_and + 2 v_sub + 1.5 v_perm = 5.5 vector instructions; per wave\n"
         "and 32-deep k step 32 elements per lane -> 176 VALU beside 96 v_mfma_f32_16x16x32_bf16 (16 cycles each, 8 of them free for\n"
         "vector issue: 192 slots).  fp32 MFMA does the same k step in 128 v_mfma_f32_16x16x4_f32 of 32 cycles.\n");
  return 0;
}
