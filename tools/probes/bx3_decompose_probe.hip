// PROBE (not product code): where does the time of the bf16x3 emulation kernel (csrc/conv_bx3.hip) go?
//
// The library kernel's inner structure (4 waves per workgroup, 64-column three-plane weight slice resident in LDS, A rows
// through a counted inline-asm register ring 4 k steps deep, the split of k step s+1 between the MFMAs of step s) on the
// fpn.inner0 shape (M = 1 075 200, N = K = 256), with its parts switched off one at a time:
//   full            loads + split + MFMAs + stores
//   no-mem          the ring is never refilled and nothing is stored: MFMAs + split + LDS reads only
//   no-mem no-split MFMAs + LDS reads only
//   mem only        the ring's loads, waits and the stores; no MFMA, no split
//   full, A wraps   the full kernel with every chunk's A rows taken from the first 64 chunks (4 MB: L2-resident) -- the
//                   MFMA / issue side at full memory-instruction count, the memory side removed
// and the in-kernel clock of each build (s_memtime / s_memrealtime stamps around the tile loop, median over workgroups;
// /opt/skills/guides/MI355X_MICROARCH.md, DVFS give-back item 6).  Stamps go to a buffer of their own.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/bx3_decompose_probe.hip -o tools/probes/bin/bx3_decompose_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

constexpr int K = 256, N = 256, BN = 64, NSL = N / BN;
constexpr int PLANE = BN * K;

template <int N_, class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N_, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl<N_>(f, std::make_integer_sequence<int, N_>{}); }

template <int OFF>
__device__ __forceinline__ void rload(f4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void rwait(f4& a0, f4& a1, f4& a2, f4& a3, f4& a4, f4& a5, f4& a6, f4& a7) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7) : "n"(CNT));
}

__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hp, uint32_t& mp, uint32_t& lp) {
  const uint32_t a0 = __float_as_uint(x0), a1 = __float_as_uint(x1);
  const uint32_t h0 = a0 & 0xffff0000u, h1 = a1 & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
  const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
  hp = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  mp = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  lp = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// MEM: ring refills, waits and stores; SPLIT: the vector work of the split; MFMA: the matrix instructions;
// wrap: 0 = off, else A rows of chunk c come from chunk c % wrap
template <bool MEM, bool SPLIT, bool MFMA, int PF = 0, bool ST = true>
__global__ void __launch_bounds__(256, 1) bx3_probe_kernel(const float* __restrict__ A, const uint16_t* __restrict__ wimg,
                                                           float* __restrict__ C, int M, int wrap,
                                                           unsigned long long* __restrict__ stamps) {
  constexpr int RING = 4, KS = K / 32, MI = 4, NI = 4;
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % NSL, tpx = per_xcd / NSL;
  const int team = xcd * tpx + idx / NSL, nteams = 8 * tpx;
  const int nchunks = M / 64;
  const int c_lo = (int)((long long)nchunks * team / nteams), c_hi = (int)((long long)nchunks * (team + 1) / nteams);
  {
    const u4* src = (const u4*)(wimg + (size_t)slice * 3 * PLANE);
    u4* dst = (u4*)Bs;
    for (int i = tid; i < 3 * PLANE / 8; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const int n0 = slice * BN;
  const bool spread = (wrap >> 17) & 1, coal = (wrap >> 18) & 1;
  wrap &= 0xffff;
  auto a_ptr = [&](int cc, int mi) -> const float* {
    int cs = wrap > 0 ? cc % wrap : cc;
    if (spread) cs = (int)(((long long)cs + (long long)slice * (nchunks / 4)) % nchunks);     // every workgroup its own rows
    // coal (timing only, the values land in the wrong lanes): consecutive lanes read consecutive 16-byte pieces of a row --
    // "row group" mi's two loads (offsets 0 / 16 bytes... here 0 and 8 rows further) cover rows 16 mi .. 16 mi + 15 as 2 x 8
    // rows x 128 bytes: 8 lines per instruction instead of 16 half-lines x 4 quarter waves
    if (coal) return A + (size_t)(cs * 64 + mi * 16 + (lane >> 3)) * K + (lane & 7) * 4;
    return A + (size_t)(cs * 64 + mi * 16 + l16) * K + g4 * 8;
  };
  const int cadj = coal ? 8 * K - 4 : 0;      // (coal: the second load of a row group = the next 8 rows)
  int cc = c_lo + wave;
  if (cc >= c_hi) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* aptr[MI];
  f4 ring[RING][MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) aptr[mi] = a_ptr(cc, mi);
  sfor<RING>([&](auto U) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value;
    sfor<MI>([&](auto I) __attribute__((always_inline)) {
      constexpr int mi = decltype(I)::value;
      rload<u * 128>(ring[u][mi][0], aptr[mi]);
      rload<u * 128 + 16>(ring[u][mi][1], aptr[mi] + cadj);
    });
  });
  uint32_t pl[2][3][MI][4];
  float dummy = 0.f;
  if (MEM) {
#pragma unroll
    for (int i = 0; i < (ST ? 16 : 0) + (PF > 0 ? RING : 0); ++i) asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(aptr[0]));
    rwait<8 * (RING - 1) + (ST ? 16 : 0) + (PF > 0 ? RING : 0)>(ring[0][0][0], ring[0][0][1], ring[0][1][0], ring[0][1][1], ring[0][2][0], ring[0][2][1],
                               ring[0][3][0], ring[0][3][1]);
  } else {      // every slot lands once, here
    sfor<RING>([&](auto U) __attribute__((always_inline)) {
      constexpr int u = decltype(U)::value;
      rwait<0>(ring[u][0][0], ring[u][0][1], ring[u][1][0], ring[u][1][1], ring[u][2][0], ring[u][2][1], ring[u][3][0],
               ring[u][3][1]);
    });
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 v = ring[0][mi][j >> 1];
      const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
      split_pair(x0, x1, pl[0][0][mi][j], pl[0][1][mi][j], pl[0][2][mi][j]);
      pl[1][0][mi][j] = pl[0][0][mi][j]; pl[1][1][mi][j] = pl[0][1][mi][j]; pl[1][2][mi][j] = pl[0][2][mi][j];
    }
  f4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
  for (; cc < c_hi; cc += 4) {
    const int cn = cc + 4 < c_hi ? cc + 4 : cc;
    const float* nptr[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) nptr[mi] = a_ptr(cn, mi);
    // PF: one 4-byte load per lane and k step touches the 64 lines (64 rows x 128 B) of that k step of the chunk PF tiles
    // ahead: 8 KB pulled towards L2 / L1 by one instruction; it has RING k steps to land (in-order counter)
    const float* pfp = nullptr;
    if (PF > 0) {
      int cp = cc + 4 * PF;
      cp = cp < nchunks ? cp : nchunks - 1;
      const int cs = wrap > 0 ? cp % wrap : cp;
      pfp = A + (size_t)(cs * 64 + lane) * K;
    }
    sfor<KS>([&](auto G) __attribute__((always_inline)) {
      constexpr int ks = decltype(G)::value, slot = ks % RING, par = ks & 1;
      constexpr int slot1 = (ks + 1) % RING;
      constexpr int PFC = PF > 0 ? RING : 0;          // prefetches younger than the awaited slot
      if constexpr (MEM) {
        sfor<MI>([&](auto I) __attribute__((always_inline)) {
          constexpr int mi = decltype(I)::value;
          if constexpr (ks + RING < KS) {
            rload<(ks + RING) * 128>(ring[slot][mi][0], aptr[mi]);
            rload<(ks + RING) * 128 + 16>(ring[slot][mi][1], aptr[mi] + cadj);
          } else {
            rload<(ks + RING - KS) * 128>(ring[slot][mi][0], nptr[mi]);
            rload<(ks + RING - KS) * 128 + 16>(ring[slot][mi][1], nptr[mi] + cadj);
          }
        });
        if constexpr (PF > 0)
          asm volatile("global_load_dword %0, %1, off offset:%2" : "+v"(dummy) : "v"(pfp), "n"(ks * 128));
        if constexpr (ks < RING - 1) {
          rwait<8 * (RING - 1) + (ST ? 16 : 0) + PFC>(ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1],
                                           ring[slot1][2][0], ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
        } else {
          rwait<8 * (RING - 1) + PFC>(ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1],
                                      ring[slot1][2][0], ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
        }
      }
      if constexpr (!MEM)      // (an empty statement that "rewrites" the slot: the split must not be hoisted out of the loop)
        asm volatile("" : "+a"(ring[slot1][0][0]), "+a"(ring[slot1][0][1]), "+a"(ring[slot1][1][0]), "+a"(ring[slot1][1][1]),
                          "+a"(ring[slot1][2][0]), "+a"(ring[slot1][2][1]), "+a"(ring[slot1][3][0]), "+a"(ring[slot1][3][1]));
      const int pos = ((ks * 4 + g4) ^ l16) * 8;
      bf8 bcur[3], bnxt[3];
      if (MFMA) {
        const uint16_t* br = Bs + l16 * K + pos;
        bcur[0] = *(const bf8*)(br); bcur[1] = *(const bf8*)(br + PLANE); bcur[2] = *(const bf8*)(br + 2 * PLANE);
      }
      sfor<NI>([&](auto NIc) __attribute__((always_inline)) {
        constexpr int ni = decltype(NIc)::value;
        if constexpr (MFMA && ni + 1 < NI) {
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
          constexpr int p = ni * 4 + mi, rg = p >> 2, j = p & 3;
          const f4 v = ring[slot1][rg][j >> 1];
          const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
          f4 c = acc[mi][ni];
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bcur[0], c, 0, 0, 0);
          uint32_t h0 = 0, h1 = 0, m0 = 0, m1 = 0;
          float r0_ = 0, r1_ = 0, q0 = 0, q1 = 0;
          if (SPLIT) { h0 = __float_as_uint(x0) & 0xffff0000u; h1 = __float_as_uint(x1) & 0xffff0000u; }
          __builtin_amdgcn_sched_barrier(0);
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[2], c, 0, 0, 0);
          if (SPLIT) { r0_ = x0 - __uint_as_float(h0); r1_ = x1 - __uint_as_float(h1); }
          __builtin_amdgcn_sched_barrier(0);
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[1], c, 0, 0, 0);
          if (SPLIT) { m0 = __float_as_uint(r0_) & 0xffff0000u; m1 = __float_as_uint(r1_) & 0xffff0000u; }
          __builtin_amdgcn_sched_barrier(0);
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[0], c, 0, 0, 0);
          if (SPLIT) { q0 = r0_ - __uint_as_float(m0); q1 = r1_ - __uint_as_float(m1); }
          __builtin_amdgcn_sched_barrier(0);
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[1], c, 0, 0, 0);
          if (SPLIT) {
            pl[par ^ 1][0][rg][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
            pl[par ^ 1][1][rg][j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
            pl[par ^ 1][2][rg][j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (MFMA) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[0], c, 0, 0, 0);
          acc[mi][ni] = c;
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (MFMA && ni + 1 < NI) { bcur[0] = bnxt[0]; bcur[1] = bnxt[1]; bcur[2] = bnxt[2]; }
      });
    });
    if (MEM && ST) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = cc * 64 + mi * 16 + g4 * 4 + r;
          *(f4*)(C + (size_t)m * N + n0 + l16 * 4) = f4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) aptr[mi] = nptr[mi];
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory");
  if (!MEM || !ST) {       // keep the accumulators alive
    float s = dummy;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) s += acc[mi][ni].x + acc[mi][ni].y + acc[mi][ni].z + acc[mi][ni].w;
    if (s == 123.456f) C[0] = s + __uint_as_float(pl[0][0][0][0] ^ pl[1][1][1][1] ^ pl[0][2][2][2]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[2 * b] = t1 - t0; stamps[2 * b + 1] = r1 - r0; }
}

// LDS-staged A path (round 5 experiment): the ring's loads read 8 rows x 128 contiguous bytes per instruction (8 cache lines
// instead of 16 half-lines per quarter wave), the landed slot goes AGPR -> LDS (ds_write_b128 straight from the accumulator
// half of the register file) in row-major order with a XOR swizzle, and the MFMA fragments (row l16, 8 k) are read back with
// ds_read_b128.  64 KB of staging (2 buffers x 4 waves x 8 KB) beside the 96 KB weight slice = the CU's 160 KB exactly.
// The host compares its output with the register-ring build's; as committed the two do NOT agree (the fragment address map
// still has a bug), so this build is a TIMING experiment: it was 5 % slower than the register ring and was not pursued.
__global__ void __launch_bounds__(256, 1) bx3_lds_kernel(const float* __restrict__ A, const uint16_t* __restrict__ wimg,
                                                         float* __restrict__ C, int M, unsigned long long* __restrict__ stamps) {
  constexpr int RING = 4, KS = K / 32, MI = 4, NI = 4;
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % NSL, tpx = per_xcd / NSL;
  const int team = xcd * tpx + idx / NSL, nteams = 8 * tpx;
  const int nchunks = M / 64;
  const int c_lo = (int)((long long)nchunks * team / nteams), c_hi = (int)((long long)nchunks * (team + 1) / nteams);
  {
    const u4* src = (const u4*)(wimg + (size_t)slice * 3 * PLANE);
    u4* dst = (u4*)Bs;
    for (int i = tid; i < 3 * PLANE / 8; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  char* stage = (char*)Bs + 3 * PLANE * 2 + wave * 16384;      // [2 buffers][64 rows][128 bytes]
  const unsigned stage_lds = (unsigned)(uintptr_t)stage;       // (LDS byte address: the low 32 bits of the generic pointer)
  const int n0 = slice * BN;
  // ring instruction j of a k step: rows 8 j .. 8 j + 7 of the tile, lane i -> row 8 j + (i >> 3), 16-byte piece i & 7
  auto a_ptr = [&](int cc, int j) -> const float* { return A + (size_t)(cc * 64 + j * 8 + (lane >> 3)) * K + (lane & 7) * 4; };
  // LDS write address of that lane: row r = 8 j + (i >> 3) (r & 7 = i >> 3 & 7), piece p at (p ^ (r & 7))
  const unsigned wr_off = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16));
  // fragment read addresses: row 16 mi + l16, pieces 2 g4 and 2 g4 + 1
  const unsigned rd0 = (unsigned)(l16 * 128 + (((2 * g4) ^ (l16 & 7)) * 16)), rd1 = (unsigned)(l16 * 128 + (((2 * g4 + 1) ^ (l16 & 7)) * 16));
  int cc = c_lo + wave;
  if (cc >= c_hi) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* aptr[8];
  f4 ring[RING][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) aptr[j] = a_ptr(cc, j);
  sfor<RING>([&](auto U) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value;
    sfor<8>([&](auto J) __attribute__((always_inline)) { rload<u * 128>(ring[u][decltype(J)::value], aptr[decltype(J)::value]); });
  });
  uint32_t pl[2][3][MI][4];
  float dummy = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(aptr[0]));
  rwait<8 * (RING - 1) + 16>(ring[0][0], ring[0][1], ring[0][2], ring[0][3], ring[0][4], ring[0][5], ring[0][6], ring[0][7]);
  auto to_lds = [&](auto SLOT, int buf) __attribute__((always_inline)) {
    constexpr int sl = decltype(SLOT)::value;
    sfor<8>([&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      const unsigned addr = stage_lds + wr_off + (unsigned)buf * 8192u;
      const f4 src = ring[sl][j];
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "a"(src), "n"(j * 1024) : "memory");
    });
  };
  f4 F[MI][2];
  auto from_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      F[mi][0] = *(const f4*)(stage + buf * 8192 + mi * 2048 + rd0);
      F[mi][1] = *(const f4*)(stage + buf * 8192 + mi * 2048 + rd1);
    }
  };
  to_lds(std::integral_constant<int, 0>{}, 0);
  from_lds(0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 v = F[mi][j >> 1];
      const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
      split_pair(x0, x1, pl[0][0][mi][j], pl[0][1][mi][j], pl[0][2][mi][j]);
    }
  f4 acc[MI][NI];
  for (; cc < c_hi; cc += 4) {
    const int cn = cc + 4 < c_hi ? cc + 4 : cc;
    const float* nptr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nptr[j] = a_ptr(cn, j);
    sfor<KS>([&](auto G) __attribute__((always_inline)) {
      constexpr int ks = decltype(G)::value, slot = ks % RING, par = ks & 1;
      constexpr int slot1 = (ks + 1) % RING;
      sfor<8>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (ks + RING < KS) rload<(ks + RING) * 128>(ring[slot][j], aptr[j]);
        else rload<(ks + RING - KS) * 128>(ring[slot][j], nptr[j]);
      });
      if constexpr (ks < RING - 1) {
        rwait<8 * (RING - 1) + 16>(ring[slot1][0], ring[slot1][1], ring[slot1][2], ring[slot1][3], ring[slot1][4], ring[slot1][5],
                                   ring[slot1][6], ring[slot1][7]);
      } else {
        rwait<8 * (RING - 1)>(ring[slot1][0], ring[slot1][1], ring[slot1][2], ring[slot1][3], ring[slot1][4], ring[slot1][5],
                              ring[slot1][6], ring[slot1][7]);
      }
      // the next k step's rows: AGPR ring slot -> LDS -> MFMA-layout fragments in registers
      to_lds(std::integral_constant<int, slot1>{}, (ks + 1) & 1);
      from_lds((ks + 1) & 1);
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
          constexpr int p = ni * 4 + mi, rg = p >> 2, j = p & 3;
          const f4 v = F[rg][j >> 1];
          const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
          f4 c = (ks == 0) ? f4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bcur[0], c, 0, 0, 0);
          const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[2], c, 0, 0, 0);
          const float r0_ = x0 - __uint_as_float(h0), r1_ = x1 - __uint_as_float(h1);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[1], c, 0, 0, 0);
          const uint32_t m0 = __float_as_uint(r0_) & 0xffff0000u, m1 = __float_as_uint(r1_) & 0xffff0000u;
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[0], c, 0, 0, 0);
          const float q0 = r0_ - __uint_as_float(m0), q1 = r1_ - __uint_as_float(m1);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[1], c, 0, 0, 0);
          pl[par ^ 1][0][rg][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
          pl[par ^ 1][1][rg][j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
          pl[par ^ 1][2][rg][j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
          __builtin_amdgcn_sched_barrier(0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[0], c, 0, 0, 0);
          acc[mi][ni] = c;
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (ni + 1 < NI) { bcur[0] = bnxt[0]; bcur[1] = bnxt[1]; bcur[2] = bnxt[2]; }
      });
    });
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = cc * 64 + mi * 16 + g4 * 4 + r;
        *(f4*)(C + (size_t)m * N + n0 + l16 * 4) = f4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) aptr[j] = nptr[j];
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[2 * b] = t1 - t0; stamps[2 * b + 1] = r1 - r0; }
}

static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) * (1.0f / 16777216.0f); }
static float nrand() { float u1 = urand() + 1e-7f, u2 = urand(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int Mbig = 16 * 200 * 336;
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = (prop.multiProcessorCount / 8) * 8;
  const size_t lds = 3 * PLANE * sizeof(uint16_t);
  printf("device %s, %d CUs; M=%d N=%d K=%d; %d timed launches per build\n", prop.gcnArchName, prop.multiProcessorCount, Mbig, N, K, iters);
  std::vector<uint16_t> img((size_t)NSL * 3 * PLANE);
  for (auto& v : img) { float f = nrand() * 0.0625f; uint32_t x; memcpy(&x, &f, 4); v = (uint16_t)(x >> 16); }
  std::vector<float> Ab((size_t)Mbig * K);
  rng_state = 4242u;
  for (auto& v : Ab) v = nrand();
  float *dA, *dC;
  uint16_t* dImg;
  unsigned long long* dSt;
  CK(hipMalloc(&dA, Ab.size() * 4));
  CK(hipMalloc(&dC, (size_t)Mbig * N * 4));
  CK(hipMalloc(&dImg, img.size() * 2));
  CK(hipMalloc(&dSt, (size_t)grid * 16));
  CK(hipMemcpy(dA, Ab.data(), Ab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dImg, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * Mbig * N * K;
  struct V { const char* name; int kern; int wrap; };
  const V vs[] = {{"full (loads + split + MFMAs + stores)", 0, 0}, {"no-mem: MFMAs + split + LDS reads", 1, 0},
                  {"no-mem, no split: MFMAs + LDS reads", 2, 0}, {"mem only: ring loads, waits, stores", 3, 0},
                  {"mem + split, no MFMA", 4, 0}, {"full, A rows from the first 64 chunks (L2-resident)", 0, 64},
                  {"mem + MFMAs, no split", 5, 0}, {"full + L2 prefetch 1 tile ahead", 6, 0},
                  {"full + L2 prefetch 2 tiles ahead", 7, 0}, {"mem only + L2 prefetch 1 tile ahead", 8, 0},
                  {"mem only + L2 prefetch 2 tiles ahead", 9, 0},
                  {"mem only, no stores", 10, 0},
                  {"mem only, every slice workgroup reads its own rows (no 4x re-read)", 3, 1 << 17},
                  {"mem only, own rows, no stores", 10, 1 << 17},
                  {"full, no stores", 11, 0},
                  {"mem only, no stores, COALESCED rows (8 lines per load)", 10, 1 << 18},
                  {"mem only, COALESCED rows", 3, 1 << 18},
                  {"full, COALESCED rows (wrong values: timing only)", 0, 1 << 18},
                  {"mem only, own rows, no stores, COALESCED", 10, (1 << 17) | (1 << 18)},
                  {"full, A staged through LDS (coalesced ring loads)", 12, 0}};
  CK(hipFuncSetAttribute((const void*)bx3_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 65536));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, false, false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, true, true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)bx3_probe_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  printf("\n%-56s %9s %12s %10s %16s\n", "build", "ms", "TF-equiv", "clock GHz", "cycles per tile");
  for (const V& v : vs) {
    auto run = [&]() {
      switch (v.kern) {
        case 0: bx3_probe_kernel<true, true, true><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 1: bx3_probe_kernel<false, true, true><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 2: bx3_probe_kernel<false, false, true><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 3: bx3_probe_kernel<true, false, false><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 4: bx3_probe_kernel<true, true, false><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 5: bx3_probe_kernel<true, false, true><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 6: bx3_probe_kernel<true, true, true, 1><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 7: bx3_probe_kernel<true, true, true, 2><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 8: bx3_probe_kernel<true, false, false, 1><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 9: bx3_probe_kernel<true, false, false, 2><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 10: bx3_probe_kernel<true, false, false, 0, false><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
        case 12: bx3_lds_kernel<<<grid, 256, lds + 65536>>>(dA, dImg, dC, Mbig, dSt); break;
        default: bx3_probe_kernel<true, true, true, 0, false><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, v.wrap, dSt); break;
      }
    };
    // >= 2 s of back-to-back launches first: the clock the chip holds under this body
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    int n = 0;
    float warm = 0.f;
    do {
      for (int i = 0; i < 50; ++i) run();
      n += 50;
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&warm, e0, e1));
    } while (warm < 2000.f && n < 20000);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<unsigned long long> st((size_t)grid * 2);
    CK(hipMemcpy(st.data(), dSt, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < grid; ++b)
      if (st[2 * b + 1] > 0) { clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1); cyc.push_back((double)st[2 * b]); }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double tiles_per_wave = (double)(Mbig / 64) / (grid / NSL) / 4.0;
    printf("%-56s %9.4f %12.1f %10.2f %16.0f\n", v.name, ms, flop / ms / 1e9, clk.empty() ? 0.0 : clk[clk.size() / 2],
           cyc.empty() ? 0.0 : cyc[cyc.size() / 2] / tiles_per_wave);
  }
  {   // the LDS-staged build against the register-ring build: same products, same order -> same bits
    std::vector<float> c0((size_t)4096 * N), c1((size_t)4096 * N);
    bx3_probe_kernel<true, true, true><<<grid, 256, lds>>>(dA, dImg, dC, Mbig, 0, dSt);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c0.data(), dC + (size_t)(Mbig - 4096) * N, c0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dC, 0xff, (size_t)Mbig * N * 4));
    bx3_lds_kernel<<<grid, 256, lds + 65536>>>(dA, dImg, dC, Mbig, dSt);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c1.data(), dC + (size_t)(Mbig - 4096) * N, c1.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < c0.size(); ++i) bad += memcmp(&c0[i], &c1[i], 4) != 0;
    printf("\nLDS-staged build vs register-ring build, last 4096 rows: %zu of %zu elements differ\n", bad, c0.size());
  }
  printf("\n(per tile and wave: 768 MFMAs of 16 cycles = 12 288 cycles at full rate; s_memtime counts shader-clock cycles,\n"
         "s_memrealtime 100 MHz ticks; cycles per tile = the median workgroup's loop cycles / its tiles per wave)\n");
  return 0;
}
