// B-resident persistent GEMM with fp32 EMULATED on the bf16 matrix pipe of gfx950 -- the design the round-5 probe
// (tools/probes/bf16x3_gemm_probe.hip, profiles/r05_bf16x3_probe.txt) measured; opt-in in round 5, the DEFAULT for the
// launches it covers since round 6.  Taken for launches whose descriptor carries a pre-split weight
// image (hnd_conv_desc.w_bf16x3, made by hnd_pack_bf16x3); the host attaches one by LAYER (hnd_bf16x3_recommended below:
// rows one image contributes, depth, output channels -- never the launch's batch), HND_BF16X3=0 attaches none.
//
// Arithmetic.  Every fp32 operand is the exact sum of three bf16 planes obtained by truncation (hi = top 16 bits, mid = top
// 16 bits of x - hi, lo = x - hi - mid: 8 + 8 + 8 significant bits).  A product a b is taken as the six plane products with
// i + j <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, smallest
// terms first inside every 32-deep k step; the three dropped products are <= 2^-24 of a b each.  Measured against fp64 on
// N(0,1) and wide-dynamic-range data: rel-L2 2.4e-7 / 2.1e-7, native fp32 MFMA 2.9e-7 / 2.7e-7.  NOT bit-identical to the
// fp32 kernels (a different summation): a second rounding family, bit-identical inside (conv_bxs.hip is its other member).
// Stated deviations (include/hnd_hip.h at w_bf16x3, tests/test_bx3_gpu.py): an Inf / NaN input makes every dependent output
// NaN (Inf - Inf in the split) and moves no other bit; plane values below 2^-126 are flushed by the bf16 pipe.
//
// Structure = bres2 (conv_bres.hip): one wave per SIMD, the weight slice resident in LDS (three pre-split planes of
// [64 columns][K], 96 KB at K = 256, XOR-swizzled 16-byte chunks), A fragments straight from global memory through a
// counted inline-asm register ring 4 k steps deep that runs across tile boundaries, and the split of k step s + 1 riding
// between the MFMAs of k step s: per accumulator tile six MFMAs (96 matrix-pipe cycles, 48 free for vector issue) and one
// pair of elements split (11 vector instructions + 2 accumulator-register reads).
//
// Roofline: bf16 MFMA at 6 products = 2.5 PFLOP/s / 6 = 0.42 PFLOP/s-equivalent of fp32 work, 2.7x the fp32 MFMA peak;
// at K = N = 256 the operands' 4 M (K + N) bytes bound a launch at 0.35 ms of HBM time against 0.34 ms of matrix time.
#include <atomic>

#include "common.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

using hnd::f32x4;
using hnd::FastDiv;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct Bx3Args {
  FastDiv div_ow, div_oh;     // m -> (n, oh, ow) of the A rows
  int nsl;                    // 64-column weight slices = workgroups per team
  int nchunks;                // ceil(M / 64)
  int mrows;                  // M.  Rows past it (the last chunk's tail) are CLAMPED to row M - 1 in every address -- A rows,
                              // residual rows, mask bytes, stores -- so the tail lanes recompute and rewrite row M - 1 with
                              // identical bits and every tile runs the same number of memory operations (exact waits)
  int cpg;                    // chunks per weight group (Winograd component), 0 = one group
  int res_up;                 // RES: res1 is the exactly 2x coarser map, nearest-upsampled (the FPN's top-down path)
};

template <int N, class F, int... I>
__device__ __forceinline__ void xfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void xfor(F&& f) {
  xfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// All ring slots live in the accumulator half of the register file (VMEM can target it; the vector ALU reaches it through
// v_accvgpr_read): with ring registers among the architectural ones hipcc sat at its limit and moved just-requested
// registers away before their wait.
template <int OFF>
__device__ __forceinline__ void rload(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
}
// The register-tied wait of a slot must be ONE statement on every path: tied waits in the two arms of a branch made hipcc
// merge the ring registers with copies placed BEFORE the wait in one arm.
template <int CNT>
__device__ __forceinline__ void rwait(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6,
                                      f32x4& a7) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
               : "n"(CNT));
}

__device__ __forceinline__ void bload(uint32_t& dst, const uint8_t* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=a"(dst) : "v"(p));
}
template <int CNT>
__device__ __forceinline__ void bwait8(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t& a4, uint32_t& a5,
                                       uint32_t& a6, uint32_t& a7) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
               : "n"(CNT));
}

// STORES: global stores per tile (16 rows, + 16 mask bytes with mask_out)
template <int CNT>
__device__ __forceinline__ void rwait8(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6,
                                       f32x4& a7) {
  rwait<CNT>(a0, a1, a2, a3, a4, a5, a6, a7);
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

// KS = K / 32 (4: K = 128, 8: K = 256).  Block = 4 waves, one 64-row chunk each at a time, all on the workgroup's 64 columns.
// RES: the epilogue adds res1 (same geometry as y: the identity / downsample sum of a Bottleneck).  Its 16 rows per tile
// travel as asm loads in the ring's own in-order stream -- requested at the top of k step KS - 4, BEFORE that step's refill,
// released at the tile's end by a wait that names exactly the 4 x 8 refills issued after them (compiler-visible loads
// would be waited for with counts that ignore the ring and drain it).  MO: mask_out (the ReLU-mask nibbles of the stored
// values): 16 more byte stores per tile in the same stream.
// In-order bookkeeping of a wait at k step ks for slot ks + 1 (requested at step ks + 1 - RING): younger operations are the
// (RING - 1) x 8 refills since, + the previous tile's STORES when the tile boundary lies in between (ks <= RING - 2), + the 16 residual loads when their issue point does (KS - 4 <= ks <= KS - 4 + RING - 2).
// MK (with RES): mask_bits -- the ReLU-backward mask as nibbles, one byte per row and lane; 16 byte loads per tile issued
// right BEFORE the residual loads (so the residual's wait covers them), applied before the store.
template <int KS, bool RES, bool MO, bool MK = false>
__global__ void __launch_bounds__(256, 1) bx3_kernel(const hnd_conv_desc d, const Bx3Args a) {
  static_assert(!MK || RES, "the mask build rides on the residual build's bookkeeping");
  constexpr int RESLOADS = RES ? (MK ? 32 : 16) : 0;
  // (RES: ring 128 + accumulators 64 + residual rows 64 = all 256 accumulator registers, and hipcc then parks just-requested
  // ring registers elsewhere before their wait; the residual builds run a ring of 2 -- their A operand was written by the
  // previous launch and comes from the memory-side cache)
  constexpr int RING = RES ? 2 : 4, K = 32 * KS, MI = 4, NI = 4, PLANE = 64 * K, STORES = MO ? 32 : 16;
  static_assert(KS % RING == 0, "ring slots are compile-time: the ring must divide the k steps of a tile");
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];       // [3 planes][64 rows][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3, per_xcd = gridDim.x >> 3;
  const int slice = idx % a.nsl, tpx = per_xcd / a.nsl;
  const int team = xcd * tpx + idx / a.nsl, nteams = 8 * tpx;
  const int n0 = slice * 64;
  const int c_lo = (int)((long long)a.nchunks * team / nteams);
  const int c_hi = (int)((long long)a.nchunks * (team + 1) / nteams);

  // A row m -> its first element (1x1 taps, no padding: always in range); 8 consecutive k per lane and k step
  auto a_ptr = [&](int cc, int mi) -> const float* {
    const unsigned m = min((unsigned)(cc * 64 + mi * 16 + l16), (unsigned)(a.mrows - 1));
    const unsigned t = hnd::fdiv(m, a.div_ow), ow_ = m - t * (unsigned)d.ow;
    const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
    const size_t pix = ((size_t)n_ * d.h + oh_ * (unsigned)d.sh) * (size_t)d.w_ + ow_ * (unsigned)d.sw;
    return d.x + pix * (size_t)d.cin + (size_t)(g4 * 8);
  };
  // epilogue constants of the lane's 4 consecutive channels (hnd::chan_of_row of its packed rows)
  const int col0 = n0 + l16 * 4;
  float es[NI], eb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    es[ni] = d.epi_scale ? d.epi_scale[col0 + ni] : 1.f;
    eb[ni] = d.epi_shift ? d.epi_shift[col0 + ni] : 0.f;
  }
  const size_t slice_elems = (size_t)3 * PLANE;

  int c = c_lo;
  while (c < c_hi) {            // one pass per weight group met by this team's range
    const int grp = a.cpg > 0 ? c / a.cpg : 0;
    const int seg_hi = a.cpg > 0 ? min(c_hi, (grp + 1) * a.cpg) : c_hi;
    __syncthreads();                                  // every wave is done with the previous slice
    {
      const u32x4* src = (const u32x4*)(d.w_bf16x3 + ((size_t)grp * a.nsl + slice) * slice_elems);
      u32x4* dst = (u32x4*)Bs;
      constexpr int NV = 3 * PLANE / 8, UB = 12;      // 16-byte vectors of the slice; loads in flight per thread
      static_assert(NV % (256 * UB) == 0, "slice does not divide among the threads");
#pragma unroll 1
      for (int i0 = 0; i0 < NV; i0 += 256 * UB) {
        u32x4 t[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) t[u] = src[i0 + u * 256 + tid];
#pragma unroll
        for (int u = 0; u < UB; ++u) dst[i0 + u * 256 + tid] = t[u];
      }
    }
    __syncthreads();

    int cc = c + wave;
    if (cc < seg_hi) {
      const float* aptr[MI];
      f32x4 ring[RING][MI][2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) aptr[mi] = a_ptr(cc, mi);
      xfor<RING>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        xfor<MI>([&](auto I) __attribute__((always_inline)) {
          constexpr int mi = decltype(I)::value;
          // (KS == RING: every k step of the first tile is requested here)
          rload<(u % KS) * 128>(ring[u][mi][0], aptr[mi]);
          rload<(u % KS) * 128 + 16>(ring[u][mi][1], aptr[mi]);
        });
      });
      uint32_t pl[2][3][MI][4];                 // [parity][hi / mid / lo][row group]: 4 dwords = 8 bf16
      // The waits of a tile's first RING - 1 k steps name the previous tile's STORES stores, which sit in the in-order
      // stream between the slots they wait for and the youngest refills.  A segment's first tile has no previous tile:
      // STORES throw-away 4-byte loads (one register, cache hits) take the stores' place, so that every tile runs the SAME
      // counts -- a run-time "first tile" flag made hipcc duplicate the tied waits over a branch and park ring registers in
      // one arm before their wait, and no path-insensitive audit of the assembly could follow it.
      // (their one destination register stays reserved until the segment's final drain: the data lands later, whatever
      // the compiler believes)
      float dummy = 0.f;
#pragma unroll
      for (int i = 0; i < STORES; ++i) asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(aptr[0]));
      rwait<8 * (RING - 1) + STORES>(ring[0][0][0], ring[0][0][1], ring[0][1][0], ring[0][1][1], ring[0][2][0], ring[0][2][1],
                                     ring[0][3][0], ring[0][3][1]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = ring[0][mi][j >> 1];
          const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
          split_pair(x0, x1, pl[0][0][mi][j], pl[0][1][mi][j], pl[0][2][mi][j]);
        }
      f32x4 acc[MI][NI];
      // the B fragments of the NEXT k step's first column tile are read during the current step's last one (round 6): at
      // the top of a step the matrix pipe no longer waits for an LDS round trip
      bf8 bfirst[3];
      {
        const uint16_t* br = Bs + l16 * K + ((g4 ^ l16) * 8);
        bfirst[0] = *(const bf8*)(br); bfirst[1] = *(const bf8*)(br + PLANE); bfirst[2] = *(const bf8*)(br + 2 * PLANE);
      }
      f32x4 resv[MI][4];                        // RES: the tile's residual rows (row 4 g4 + r of row group mi)
      uint32_t resm[MI][4];                     // MK: ... and their mask bytes
      for (; cc < seg_hi; cc += 4) {
        const int cn = cc + 4 < seg_hi ? cc + 4 : cc;       // the wave's next chunk (itself at the end: harmless)
        const float* nptr[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) nptr[mi] = a_ptr(cn, mi);
        xfor<KS>([&](auto G) __attribute__((always_inline)) {
          constexpr int ks = decltype(G)::value, slot = ks % RING, par = ks & 1;
          constexpr int slot1 = (ks + 1) % RING;            // the step whose planes are made during this one
          if constexpr (MK && ks == KS - 4) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                bload(resm[mi][r], d.mask_bits + (((size_t)min(cc * 64 + mi * 16 + g4 * 4 + r, a.mrows - 1) * (size_t)d.ldc + col0) >> 2));
          }
          if constexpr (RES && ks == KS - 4) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const unsigned mr = (unsigned)(cc * 64 + mi * 16 + g4 * 4);      // four consecutive output pixels
              const unsigned m0 = min(mr, (unsigned)(a.mrows - 4));          // (upsampled: M % 4 == 0, whole groups clamp)
              // dense: pixel m0 + r.  Nearest 2x upsampling (yh = 2 res1_h, yw = 2 res1_w, ow % 4 == 0: the four pixels share
              // a row and start at an even column): pixel (n, y, x) reads (n, y / 2, x / 2).  Both addresses are computed
              // and one is SELECTED, so that the asm loads sit in one straight block whatever the mode (an if / else here
              // comes out as two correlated branches, which tools/audit_bres_asm.py cannot follow)
              const unsigned t = hnd::fdiv(m0, a.div_ow), ow_ = m0 - t * (unsigned)d.ow;
              const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
              const size_t pu = ((size_t)n_ * d.res1_h + (oh_ >> 1)) * (size_t)d.res1_w + (ow_ >> 1);
              const bool up = a.res_up != 0;
              const unsigned ml = (unsigned)(a.mrows - 1);
              const size_t p0 = up ? pu : (size_t)min(mr, ml), p1 = up ? pu : (size_t)min(mr + 1, ml),
                           p2 = up ? pu + 1 : (size_t)min(mr + 2, ml), p3 = up ? pu + 1 : (size_t)min(mr + 3, ml);
              rload<0>(resv[mi][0], d.res1 + p0 * (size_t)d.ldc + col0);
              rload<0>(resv[mi][1], d.res1 + p1 * (size_t)d.ldc + col0);
              rload<0>(resv[mi][2], d.res1 + p2 * (size_t)d.ldc + col0);
              rload<0>(resv[mi][3], d.res1 + p3 * (size_t)d.ldc + col0);
            }
          }
          // slot `slot` was split during the previous step: refill it for the step RING ahead (this tile or the next)
          xfor<MI>([&](auto I) __attribute__((always_inline)) {
            constexpr int mi = decltype(I)::value;
            if constexpr (ks + RING < KS) {
              rload<(ks + RING) * 128>(ring[slot][mi][0], aptr[mi]);
              rload<(ks + RING) * 128 + 16>(ring[slot][mi][1], aptr[mi]);
            } else {
              rload<(ks + RING - KS) * 128>(ring[slot][mi][0], nptr[mi]);
              rload<(ks + RING - KS) * 128 + 16>(ring[slot][mi][1], nptr[mi]);
            }
          });
          // The next step's slot was requested RING - 1 steps ago: 8 (RING - 1) younger ring loads may be in flight, plus
          // the previous tile's STORES stores (the segment's first tile: as many throw-away loads) while they are younger
          // than it (k steps 0 .. RING - 2).  Exact counts: full tiles only (the launcher requires M % 64 == 0).
          constexpr int kResYounger = (RES && ks >= KS - 4 && ks <= KS - 4 + RING - 2) ? RESLOADS : 0;
          if constexpr (ks < RING - 1) {
            // (the counter holds 6 bits: a larger allowance is clipped to 63 -- a stronger, still correct wait)
            rwait<(8 * (RING - 1) + STORES + kResYounger < 63 ? 8 * (RING - 1) + STORES + kResYounger : 63)>(
                ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1], ring[slot1][2][0],
                ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
          } else {
            rwait<8 * (RING - 1) + kResYounger>(ring[slot1][0][0], ring[slot1][0][1], ring[slot1][1][0], ring[slot1][1][1],
                                                ring[slot1][2][0], ring[slot1][2][1], ring[slot1][3][0], ring[slot1][3][1]);
          }
          const int pos = ((ks * 4 + g4) ^ l16) * 8;
          bf8 bcur[3], bnxt[3];
          bcur[0] = bfirst[0]; bcur[1] = bfirst[1]; bcur[2] = bfirst[2];
          xfor<NI>([&](auto NIc) __attribute__((always_inline)) {
            constexpr int ni = decltype(NIc)::value;
            if constexpr (ni + 1 < NI) {
              const uint16_t* br = Bs + ((ni + 1) * 16 + l16) * K + pos;
              bnxt[0] = *(const bf8*)(br); bnxt[1] = *(const bf8*)(br + PLANE); bnxt[2] = *(const bf8*)(br + 2 * PLANE);
            } else {                            // (the next step, or step 0 of the next tile: the same resident slice)
              const uint16_t* br = Bs + l16 * K + ((((ks + 1) % KS) * 4 + g4) ^ l16) * 8;
              bfirst[0] = *(const bf8*)(br); bfirst[1] = *(const bf8*)(br + PLANE); bfirst[2] = *(const bf8*)(br + 2 * PLANE);
            }
            xfor<MI>([&](auto MIc) __attribute__((always_inline)) {
              constexpr int mi = decltype(MIc)::value;
              auto frag = [&](int q) __attribute__((always_inline)) {
                const u32x4 t = {pl[par][q][mi][0], pl[par][q][mi][1], pl[par][q][mi][2], pl[par][q][mi][3]};
                return __builtin_bit_cast(bf8, t);
              };
              const bf8 ah = frag(0), am = frag(1), al = frag(2);
              // one pair of the NEXT step's elements rides between this tile's six MFMAs: piece p = ni * 4 + mi -> row
              // group p / 4, pair p % 4
              constexpr int p = ni * 4 + mi, rg = p >> 2, j = p & 3;
              const f32x4 v = ring[slot1][rg][j >> 1];
              const float x0 = (j & 1) ? v.z : v.x, x1 = (j & 1) ? v.w : v.y;
              f32x4 cacc = (ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni];
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bcur[0], cacc, 0, 0, 0);      // smallest terms first
              const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
              __builtin_amdgcn_sched_barrier(0);
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[2], cacc, 0, 0, 0);
              const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
              __builtin_amdgcn_sched_barrier(0);
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[1], cacc, 0, 0, 0);
              const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
              __builtin_amdgcn_sched_barrier(0);
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bcur[0], cacc, 0, 0, 0);
              const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
              __builtin_amdgcn_sched_barrier(0);
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[1], cacc, 0, 0, 0);
              pl[par ^ 1][0][rg][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
              pl[par ^ 1][1][rg][j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
              pl[par ^ 1][2][rg][j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
              __builtin_amdgcn_sched_barrier(0);
              cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcur[0], cacc, 0, 0, 0);
              acc[mi][ni] = cacc;
              __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (ni + 1 < NI) { bcur[0] = bnxt[0]; bcur[1] = bnxt[1]; bcur[2] = bnxt[2]; }
          });
        });
        // the tile's residual rows: requested at k step KS - 4, the 4 x 8 refills of steps KS - 4 .. KS - 1 came after them
        if constexpr (RES) {
          rwait8<32>(resv[0][0], resv[0][1], resv[0][2], resv[0][3], resv[1][0], resv[1][1], resv[1][2], resv[1][3]);
          rwait8<32>(resv[2][0], resv[2][1], resv[2][2], resv[2][3], resv[3][0], resv[3][1], resv[3][2], resv[3][3]);
          if constexpr (MK) {                        // (older than the residual rows: landed with them)
            bwait8<32>(resm[0][0], resm[0][1], resm[0][2], resm[0][3], resm[1][0], resm[1][1], resm[1][2], resm[1][3]);
            bwait8<32>(resm[2][0], resm[2][1], resm[2][2], resm[2][3], resm[3][0], resm[3][1], resm[3][2], resm[3][3]);
          }
        }
        // the tile's 16 row stores: C/D layout row = 4 g4 + reg of a 16-row group, column = l16 -> channels col0 .. col0 + 3;
        // dense output (the launcher requires it): output pixel = GEMM row
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t m = (size_t)min(cc * 64 + mi * 16 + g4 * 4 + r, a.mrows - 1);
            f32x4 v;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              float x = acc[mi][ni][r] * es[ni] + eb[ni];
              if (RES) x += resv[mi][r][ni];
              if (MK) x = ((resm[mi][r] >> ni) & 1u) ? x : 0.f;
              v[ni] = d.relu ? fmaxf(x, 0.f) : x;
            }
            const size_t yo = m * (size_t)d.ldc + col0;
            *(f32x4*)(d.y + yo) = v;
            if (MO)
              d.mask_out[yo >> 2] = (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) |
                                              (v[3] > 0.f ? 8 : 0));
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aptr[mi] = nptr[mi];
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory");   // the ring's last (unused) prefetches land before reuse
    }
    c = seg_hi;
  }
}

// packed fp32 operand [groups][rows_pad][K] -> per (k part, group, 64-row slice) the LDS image [3 planes][64 rows][KI] of
// bf16, chunk c (8 values) of row r at position c ^ (r & 15).  KI = K for K = 128 / 256; K = 512 is two k parts of 256 (the
// launch runs two passes, the second adding the first one's partial result): image part p starts at p * groups * nsl slices.
__global__ void pack_bx3_kernel(const float* __restrict__ w, uint16_t* __restrict__ img, int rows_pad, int K, int KI, int groups,
                                long long group_stride, long long total) {
  const int nsl = rows_pad / 64;
  const size_t plane = (size_t)64 * KI;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    long long t = e / K;
    const int row = (int)(t % rows_pad), g = (int)(t / rows_pad);
    const float x = w[(size_t)g * (size_t)group_stride + (size_t)row * K + k];
    const uint32_t xb = __float_as_uint(x), hb = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const int part = k / KI, kk = k - part * KI;
    const int s = row / 64, r = row % 64, c = kk >> 3, pos = c ^ (r & 15);
    uint16_t* o = img + (((size_t)part * groups + g) * nsl + s) * 3 * plane + (size_t)r * KI + pos * 8 + (kk & 7);
    o[0] = (uint16_t)(hb >> 16);
    o[plane] = (uint16_t)(mb >> 16);
    o[2 * plane] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

int cu_count_bx3() {
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

template <int KS, bool RES, bool MO, bool MK = false>
int launch_bx3_t(const hnd_conv_desc& d, const Bx3Args& a, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = bx3_kernel<KS, RES, MO, MK>;
  const size_t lds = (size_t)3 * 64 * 32 * KS * sizeof(uint16_t);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(bx3<%d>) failed: %s", KS, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, d, a);
  return hnd::check_launch("hnd_conv2d_igemm(bx3)");
}

}  // namespace

namespace hnd {

// Which launches SHOULD run on the emulation: decided from the LAYER alone -- depth, output channels and the GEMM rows ONE image
// contributes -- never from the batch of the launch, so that image i's maps are the same bits alone and inside any batch
// (DESIGN section 4 rule 4).  Priced at the path's design point, 16 images per GPU (BASELINE.json configs[1]): the launch
// pays when a team of cout / 64 workgroups gets at least 8 chunks of 64 rows there (16 with several passes over k: the
// slice load and the ramp are paid once per pass) -- profiles/r05_bf16x3_shapes.txt.
static bool bx3_recommended(long long rows_per_image, int kdim, int cout) {
  if (rows_per_image <= 0 || cout <= 0 || cout % 64 != 0) return false;
  if (kdim != 128 && (kdim % 256 != 0 || kdim > 2048)) return false;
  // (round 6 A/B, same box, img/s at batch 16 / batch 4: K >= 1024 launches on the B-streamed build instead 214.2 / 170.7,
  // K >= 512 210.5 / 161.2, as shipped 216.2 / 176.6 -- the passes over k stay)
  const int per_xcd = cu_count_bx3() / 8, nsl = cout / 64;
  if (per_xcd < 1 || nsl > per_xcd || per_xcd % nsl != 0) return false;
  const long long nteams = 8ll * (per_xcd / nsl);
  const long long chunks_at_16 = (rows_per_image * 16 + 63) / 64;
  return chunks_at_16 / nteams >= (kdim > 512 ? 16 : 8);
}

// CAN the emulation take this launch (a weight image attached = the caller asks for it; nothing here reads the row count
// beyond "at least one chunk"): tap-free K = 128 or 256 P <= 2048 (P passes over k), cout a multiple of 64 that divides an
// XCD's CUs, a dense output (output pixel = GEMM row; any row count -- the last chunk's tail is clamped to row M - 1), and
// an epilogue of scale / shift, a residual (same geometry, or the FPN's 2x nearest-upsampled coarser map), the
// ReLU-backward mask as nibbles (with a residual), ReLU and the ReLU-mask nibbles.
bool bx3_applies(const hnd_conv_desc& d) {
  if (!d.w_bf16x3) return false;
  if (d.kh != 1 || d.kw != 1 || d.bh != 0 || d.bw != 0 || d.cin != d.kdim) return false;
  if (d.kdim != 128 && (d.kdim % 256 != 0 || d.kdim > 2048)) return false;
  if (d.stats || d.pro_scale || d.res2 || d.mask || d.bwd_x) return false;
  if (d.mask_bits && d.mask_out) return false;
  // (the mask build rides on the residual build: the launch's own same-geometry residual, or the partial result of K > 256)
  if (d.mask_bits && d.kdim <= 256 && (!d.res1 || d.res1_mode != 0)) return false;
  if (d.res1 && ((uintptr_t)d.res1 % 16) != 0) return false;
  // a residual of y's geometry, or the exactly 2x coarser map of the FPN's top-down path (whole 4-pixel groups per row)
  if (d.res1 && d.res1_mode == 1 && (d.yh != 2 * d.res1_h || d.yw != 2 * d.res1_w || d.ow % 4 != 0)) return false;
  if (d.cout % 64 != 0 || d.ldc % 4 != 0 || ((uintptr_t)d.y % 16) != 0) return false;
  const int per_xcd = cu_count_bx3() / 8, nsl = d.cout / 64;
  if (per_xcd < 1 || nsl > per_xcd || per_xcd % nsl != 0) return false;
  if (d.y_sh != 1 || d.y_sw != 1 || d.y_oh != 0 || d.y_ow != 0 || d.yh != d.oh || d.yw != d.ow) return false;
  if ((long long)(d.oh - 1) * d.sh >= d.h || (long long)(d.ow - 1) * d.sw >= d.w_) return false;
  const long long M = (long long)d.n * d.oh * d.ow;
  if (M < 64 || d.w_group_rows % 64 != 0) return false;
  if (M % 64 != 0 && d.w_group_rows != 0) return false;      // (a tail: one weight group)
  return true;
}

static int launch_bx3_one(const hnd_conv_desc& d, int kpart, hipStream_t stream) {
  Bx3Args a;
  a.div_ow = make_fastdiv((unsigned)d.ow);
  a.div_oh = make_fastdiv((unsigned)d.oh);
  a.nsl = d.cout / 64;
  a.mrows = (int)((long long)d.n * d.oh * d.ow);
  a.nchunks = (a.mrows + 63) / 64;
  a.cpg = d.w_group_rows / 64;
  a.res_up = d.res1 && d.res1_mode == 1;
  const int grid = (cu_count_bx3() / 8) * 8;
  if (d.mask_bits)      // (with a residual of y's geometry and no mask_out: bx3_applies)
    return kpart == 128 ? launch_bx3_t<4, true, false, true>(d, a, grid, stream)
                        : launch_bx3_t<8, true, false, true>(d, a, grid, stream);
  const int sel = (kpart == 128 ? 0 : 4) | (d.res1 ? 2 : 0) | (d.mask_out ? 1 : 0);
  switch (sel) {
    case 0: return launch_bx3_t<4, false, false>(d, a, grid, stream);
    case 1: return launch_bx3_t<4, false, true>(d, a, grid, stream);
    case 2: return launch_bx3_t<4, true, false>(d, a, grid, stream);
    case 3: return launch_bx3_t<4, true, true>(d, a, grid, stream);
    case 4: return launch_bx3_t<8, false, false>(d, a, grid, stream);
    case 5: return launch_bx3_t<8, false, true>(d, a, grid, stream);
    case 6: return launch_bx3_t<8, true, false>(d, a, grid, stream);
    default: return launch_bx3_t<8, true, true>(d, a, grid, stream);
  }
}

int launch_bx3(const hnd_conv_desc& d, hipStream_t stream) {
  if (!bx3_applies(d)) {
    set_error("launch_bx3: descriptor not eligible");
    return HND_ERR_INVALID;
  }
  if (d.kdim <= 256) return launch_bx3_one(d, d.kdim, stream);
  // K = 256 P: the resident slice of three planes holds 256 k.  Pass 1: k 0 .. 255, y = acc * scale + shift (+ the launch's
  // own residual; no ReLU, no mask); pass p: k 256 (p - 1) .. with res1 = y: y = acc * scale + y; the last pass applies the
  // ReLU-backward mask, the ReLU and writes the mask nibbles.  The kernel takes the row stride of x from cin and the depth
  // from its template, so a later pass is the same launch 256 floats further on.  (fp32 partial sums in y between passes:
  // what the accumulator holds anyway)
  const long long groups = d.w_group_rows > 0 ? ((long long)d.n * d.oh * d.ow) / d.w_group_rows : 1;
  const int parts = d.kdim / 256;
  const size_t part_elems = (size_t)groups * (size_t)(d.cout / 64) * (size_t)3 * 64 * 256;
  for (int p = 0; p < parts; ++p) {
    hnd_conv_desc q = d;
    if (p + 1 < parts) {
      q.relu = 0;
      q.mask_out = nullptr;
      q.mask_bits = nullptr;
    }
    if (p > 0) {
      q.x = d.x + (size_t)256 * p;
      q.w_bf16x3 = d.w_bf16x3 + part_elems * (size_t)p;
      q.epi_shift = nullptr;
      q.res1 = d.y;
      q.res1_mode = 0;
    }
    const int rc = launch_bx3_one(q, 256, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace hnd

extern "C" int hnd_bf16x3_recommended(int64_t rows_per_image, int kdim, int cout) {
  return hnd::bx3_recommended((long long)rows_per_image, kdim, cout) ? 1 : 0;
}

extern "C" size_t hnd_pack_bf16x3_elems(int rows_pad, int kdim, int groups) {
  if (rows_pad <= 0 || rows_pad % 64 != 0 || kdim <= 0 || kdim % 8 != 0 || groups < 1) return 0;
  return (size_t)groups * (size_t)rows_pad * (size_t)kdim * 3;
}

extern "C" int hnd_pack_bf16x3(const float* w_packed, uint16_t* img, int rows_pad, int kdim, int groups,
                               int64_t group_stride, void* stream) {
  HND_REQUIRE(w_packed && img && rows_pad > 0 && rows_pad % 64 == 0 && (kdim == 128 || (kdim % 256 == 0 && kdim <= 2048)) &&
                  groups >= 1 && (groups == 1 || group_stride >= (int64_t)rows_pad * kdim),
              "hnd_pack_bf16x3: bad arguments (kdim must be 128 or a multiple of 256 up to 2048)");
  const long long total = (long long)groups * rows_pad * kdim;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_bx3_kernel, dim3((unsigned)blocks), dim3(256), 0, hnd::as_stream(stream), w_packed, img, rows_pad,
                     kdim, kdim > 256 ? 256 : kdim, groups, (long long)group_stride, total);
  return hnd::check_launch("hnd_pack_bf16x3");
}
