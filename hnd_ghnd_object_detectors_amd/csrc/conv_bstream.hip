// B-streamed persistent GEMM on fp32 MFMA for gfx950 (MI355X): the LONG-K class of hnd_conv2d_igemm launches -- the
// 1x1 convolutions and data gradients of layer3 / layer4 and the FPN laterals (K = 1024 / 2048) and the stride-2 3x3
// convolutions with their data gradients (K = 512 ... 4608 over taps), 19 ms of the 97 ms round-3 step on the tiled
// kernel at 100-112 TFLOP/s.  conv_bres.hip showed what lifts a wave from 0.67 to 0.79 of the matrix peak: A fragments
// straight from global memory into a deep register ring (no LDS round trip, no exposed load latency), one wave per
// SIMD with the whole 512-register file, and no barrier per 16 k.  There the weight slice is RESIDENT in LDS, which
// stops at K = 512.  Here the slice streams through LDS instead:
//   * B: three LDS stages of [BN x 64 k] (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128 fragments).  The
//     four waves fetch a stage cooperatively -- BN / 64 16-byte loads per lane and k group, requested FOUR stages
//     ahead, parked in registers for eight k groups (see the counter note), written to LDS two stages ahead; ONE
//     workgroup barrier per stage (64 k = 256 MFMAs per wave, 8192 cycles) instead of one per 16 k;
//   * A: as in bres2 -- a lane owns row l16 of a 16-row group and 4 consecutive k, one global_load_dwordx4 per (row
//     group, k group) into a ring 8 k groups deep; out-of-range taps of the 3x3 convs read a page of zeros;
//   * the stream never stops: loads run 8 (A) / 16 (B) k groups ahead of the MFMAs ACROSS tile boundaries, a
//     workgroup walks its tiles back to back and only the epilogue (scale/shift, residuals, mask, ReLU, stores:
//     conv_epilogue.h, any operand set) sits between two tiles.
// Counter note: every vector-memory operation of a wave retires in issue order through one counter (vmcnt), so a wait
// for a load issued at time t also waits for everything older.  A B load that were consumed EARLIER than the ring
// loads issued just before it would therefore drain the ring.  All loads of k group g -- four ring loads and the NB B
// loads -- are consumed exactly 8 k groups later, behind one `s_waitcnt vmcnt(7 * (4 + NB))`: at most that many
// younger loads exist, so the slot has landed (tools/audit_bres_asm.py checks the register side in the disassembly).
// Accumulation order per output element = the tiled kernel's (k groups ascending over (tap, ci); MFMA s of a group sums
// k = s, 4+s, 8+s, 12+s) and prologue / epilogue are the same code: results are BIT-IDENTICAL to igemm_kernel's.
//
// Roofline: fp32 MFMA (157.3 TFLOP/s).  Per wave and k group: 64 MFMAs (2048 cycles) against 4 + NB global loads,
// 4 ds_read_b128 and NB ds_write_b128.
#include <atomic>

#include "common.h"
#include "conv_epilogue.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

using hnd::f32x4;
using hnd::FastDiv;

__device__ float g_zero_page[256];     // source of out-of-range taps: 8 k groups x 64 B + 64 B (zero-initialised)

struct BstreamArgs {
  FastDiv div_ow, div_oh;     // m -> (n, oh, ow)
  int mtiles, ntiles;         // tile grid
  int kg8;                    // iterations of 8 k groups (128 k) per tile
  int dbg;                    // HND_BSTREAM_DBG: 1 = no epilogue (timing), 2 = heads never raise their flag (tests the timeout)
  int spin_limit;             // polls of a relay flag before the wait gives up
  float* relay;               // stream-K relay workspace (hnd_conv2d_igemm_workspace), or null: tiles round-robin
  int* err;                   // host-visible sticky error word (pinned, mapped): a relay wait that timed out raises it
};

template <int N, class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ACC: the register lives in the accumulator half of the file (see conv_bres.hip: half of the ring does)
template <int OFF, bool ACC>
__device__ __forceinline__ void vload(f32x4& dst, const float* p) {
  if (ACC) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
  else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}
template <int N, bool ACC>
__device__ __forceinline__ void slot_wait(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& b0, f32x4& b1) {
  if (ACC)
    asm volatile("s_waitcnt vmcnt(%6)" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(b0), "+v"(b1) : "n"(N));
  else
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "n"(N));
}
template <int N, bool ACC>
__device__ __forceinline__ void slot_wait(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& b0) {
  if (ACC) asm volatile("s_waitcnt vmcnt(%5)" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(b0) : "n"(N));
  else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0) : "n"(N));
}

// WN = wave columns: block tile (64 * 4 / WN) x (64 * WN), every wave a 64 x 64 tile of 4 x 4 MFMA tiles.
//
// Work split ("relay", a stream-K that keeps the accumulation order): the launch is T tiles x kg8 iterations of 128 k;
// workgroup w (numbered so that one XCD holds consecutive w) takes the units [U w / G, U (w+1) / G) of that linear
// space, so every CU gets the same number of MFMAs whatever T is -- no partial last round (M = 16 800 / 67 200 pixels
// at layer4 / layer3: 528 or 1050 tiles on 256 CUs lost 31 % / 18 % to it).  A range that ends inside a tile computes
// the tile's HEAD k range and parks the 128 x 128 accumulators in the workspace (64 KB per workgroup); the next
// workgroup, whose range starts inside that tile, loads them and CONTINUES the same k chain before the epilogue --
// the sum is the sequential one, bit for bit.  Order inside a workgroup: head first, whole tiles, tail last; with at
// least one tile of work per workgroup (the host checks) the head of w-1 is finished before the tail of w starts, so
// the flag wait never spins in practice and cannot deadlock (the writer waits for nobody).  Without a workspace the
// tiles go round-robin (no relay).
template <int WN, bool PRO, bool TAPS>
__global__ void __launch_bounds__(256, 1) bstream_kernel(const hnd_conv_desc d, const BstreamArgs a) {
  constexpr int WM = 4 / WN, BM = 64 * WM, BN = 64 * WN, MI = 4, NI = 4, NB = BN / 64, NST = 3;
  constexpr int VM = 7 * (4 + NB);
  constexpr int STG = BN * 64;                          // floats per LDS stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                     // [NST][BN][64]: chunk c of row r at position c ^ (r & 15)
  float* epi = Bs + NST * STG;                          // [2][cout] epilogue scale, shift
  float* pro = epi + 2 * d.cout;                        // [2][cin] prologue scale, shift
  int* tabs = (int*)(pro + (PRO ? 2 * d.cin : 0));      // [4 waves][2][64]: output / res1 pixel of the wave's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int l16 = lane & 15, g4 = lane >> 4;
  int* rowoff = tabs + wave * 128;
  int* resoff = rowoff + 64;

  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);       // blocks of one XCD are consecutive
  const int T = a.mtiles * a.ntiles;
  const int M = d.n * d.oh * d.ow;
  const int kg8 = a.kg8;

  // ---- this workgroup's segments: (tile, first iteration, end iteration, kind)
  enum { FULL = 0, HEAD = 1, TAIL = 2 };
  int nseg, first_full = 0, nfull = 0, tA = 0, offA = 0, tB = 0, offB = 0;
  bool has_head = false;
  if (a.relay) {
    const long long U = (long long)T * kg8;
    const long long u0 = U * lb / G, u1 = U * (lb + 1) / G;
    tA = (int)(u0 / kg8); offA = (int)(u0 - (long long)tA * kg8);
    tB = (int)(u1 / kg8); offB = (int)(u1 - (long long)tB * kg8);
    has_head = offB > 0;
    first_full = tA + (offA > 0 ? 1 : 0);
    nfull = tB - first_full;
    nseg = (has_head ? 1 : 0) + nfull + (offA > 0 ? 1 : 0);
  } else {
    if (lb >= T) return;
    nfull = nseg = (T - lb + G - 1) / G;                // tiles lb, lb + G, ...
  }
  // ---- relay bookkeeping behind the accumulator sets: [G] flags, launch counter, finished-workgroup ticket.
  // A flag carries the EPOCH of the launch that raised it (counter + 1), never 0 / 1: a reader waits for exactly this
  // launch's value and nobody resets anything, so a writer that arrives after its reader gave up (see the timeout
  // below) cannot leave a flag that a later launch on this workspace would mistake for its own.  The counter is advanced
  // by the last workgroup to finish; every workgroup reads it before it takes its ticket, so all G see one value.
  float* relay_p = a.relay;                             // [G][16][256] float4 accumulator sets
  int* relay_f = (int*)(a.relay + (size_t)G * 16384);   // [G] flags, [G] = launch counter, [G + 1] = ticket
  int epoch = 0;                                        // (thread 0 only)
  if (a.relay && tid == 0) epoch = __hip_atomic_load(relay_f + G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  auto launch_done = [&]() {
    if (a.relay && tid == 0) {
      if (__hip_atomic_fetch_add(relay_f + G + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
        __hip_atomic_store(relay_f + G + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(relay_f + G, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (nseg == 0) {
    launch_done();
    return;
  }
  auto seg_of = [&](int i, int& tile, int& it0, int& it1, int& kind) {
    i = i < nseg ? i : nseg - 1;
    if (!a.relay) { tile = lb + i * G; it0 = 0; it1 = kg8; kind = FULL; return; }
    if (has_head) {
      if (i == 0) { tile = tB; it0 = 0; it1 = offB; kind = HEAD; return; }
      --i;
    }
    if (i < nfull) { tile = first_full + i; it0 = 0; it1 = kg8; kind = FULL; return; }
    tile = tA; it0 = offA; it1 = kg8; kind = TAIL;
  };

  for (int c = tid; c < d.cout; c += 256) {
    epi[c] = d.epi_scale ? d.epi_scale[c] : 1.f;
    epi[d.cout + c] = d.epi_shift ? d.epi_shift[c] : 0.f;
  }
  if (PRO)
    for (int c = tid; c < d.cin; c += 256) {
      pro[c] = d.pro_scale[c];
      pro[d.cin + c] = d.pro_shift[c];
    }

  // ---- A side: per row group the lane's source row.  1x1: a pointer; taps: (first pixel of the image, ih0, iw0)
  struct Rows {
    const float* p[MI];
    int pix[MI], ih0[MI], iw0[MI];
  };
  auto rows_of = [&](int mt, Rows& r) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int m = mt * BM + wm * 64 + mi * 16 + l16;
      m = m < M ? m : M - 1;
      const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
      const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
      const int ih0 = (int)oh_ * d.sh + d.bh, iw0 = (int)ow_ * d.sw + d.bw;
      if (TAPS) {
        r.pix[mi] = (int)n_ * d.h * d.w_;
        r.ih0[mi] = ih0;
        r.iw0[mi] = iw0;
      } else {
        r.p[mi] = d.x + ((size_t)((int)n_ * d.h + ih0) * (size_t)d.w_ + iw0) * (size_t)d.cin + (size_t)(g4 * 4);
      }
    }
  };
  // the lane's four load addresses for the 128 k that start at `kofs` of a tile (one tap: cin % 128 == 0)
  auto a_addr = [&](const Rows& r, int kofs, const float* (&lp)[MI], unsigned& okbits) {
    okbits = 0xfu;
    if (!TAPS) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) lp[mi] = r.p[mi] + kofs;
    } else {
      const int tap = kofs / d.cin, ci0 = kofs - tap * d.cin;
      const int ti = tap / d.kw, tj = tap - ti * d.kw;
      okbits = 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int ih = r.ih0[mi] + ti * d.dh, iw = r.iw0[mi] + tj * d.dw;
        const bool ok = (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
        const float* src = d.x + (size_t)(unsigned)(r.pix[mi] + ih * d.w_ + iw) * (size_t)d.cin + (size_t)(ci0 + g4 * 4);
        lp[mi] = ok ? src : (const float*)g_zero_page + g4 * 4;
        okbits |= (ok ? 1u : 0u) << mi;
      }
    }
  };
  // ---- B side: this thread's part of a tile's weight slice (row tid>>4 of each 16-row load, chunk tid&15 of a stage)
  const int r4 = tid >> 4, c16 = tid & 15;
  const size_t kd16 = (size_t)16 * (size_t)d.kdim;
  auto b_base = [&](int tile) -> const float* {
    const int mt = tile / a.ntiles, nt = tile - mt * a.ntiles;
    const size_t grp = d.w_group_rows > 0 ? (size_t)((mt * BM) / d.w_group_rows) * (size_t)d.w_group_stride : 0;
    return d.w + grp + (size_t)(nt * BN + r4) * (size_t)d.kdim + (size_t)(c16 * 4);
  };
  const int wpos0 = r4 * 64 + ((c16 ^ r4) << 2);        // LDS position of that part inside 16 rows of a stage
  int bsw[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) bsw[u] = ((4 * u + g4) ^ l16) * 4;
  const int frow = (wn * 64 + l16) * 64;                // the lane's fragment row inside a stage
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;
  const bool vec_ok = (d.ldc % NI == 0) &&
                      (((uintptr_t)d.y | (uintptr_t)d.res1 | (uintptr_t)d.res2 | (uintptr_t)d.mask) % (4 * NI) == 0);

  // ---- the load streams: A runs one iteration (8 k groups) ahead of the MFMAs, B two (4 stages); both walk the
  // segment list on their own (past its end they stay on the last iteration: harmless extra loads)
  struct Pos { int seg, it, it1, tile; };
  auto pos_init = [&](Pos& q) {
    int kind;
    q.seg = 0;
    seg_of(0, q.tile, q.it, q.it1, kind);
  };
  auto pos_next = [&](Pos& q) -> bool {                 // true: entered a new segment
    if (q.it + 1 < q.it1) { ++q.it; return false; }
    if (q.seg + 1 >= nseg) return false;
    int kind;
    ++q.seg;
    seg_of(q.seg, q.tile, q.it, q.it1, kind);
    return true;
  };
  Pos pa, pb;
  pos_init(pa);
  pos_init(pb);
  Rows ra;
  rows_of(pa.tile / a.ntiles, ra);
  const float* bt = b_base(pb.tile);

  // ---- fill: the first iteration's two B stages go straight to LDS; then the loads of "k groups -8 .. -1"
  {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* src = bt + (size_t)(2 * pb.it + s) * 64;
#pragma unroll
      for (int q = 0; q < 4 * NB; ++q)
        *(f32x4*)(Bs + s * STG + q * 1024 + wpos0) = *(const f32x4*)(src + (size_t)q * kd16);
    }
  }
  __syncthreads();
  if (pos_next(pb)) bt = b_base(pb.tile);

  f32x4 ring[8][MI];
  f32x4 bst[8][2];
  unsigned okr[8];
  {
    const float* lp[MI];
    unsigned okb;
    a_addr(ra, pa.it * 128, lp, okb);
    const float* bl0 = bt + (size_t)(2 * pb.it) * 64;
    const float* bl1 = bl0 + 64;
    sfor<8>([&](auto U) __attribute__((always_inline)) {
      constexpr int u = decltype(U)::value, u4 = u & 3;
      constexpr bool acc = (u & 1) != 0;
      const float* bl = u < 4 ? bl0 : bl1;
      okr[u] = okb;
      vload<u * 64, acc>(ring[u][0], lp[0]);
      vload<u * 64, acc>(ring[u][1], lp[1]);
      if (NB == 2) vload<0, false>(bst[u][0], bl + (size_t)(2 * u4) * kd16);
      vload<u * 64, acc>(ring[u][2], lp[2]);
      vload<u * 64, acc>(ring[u][3], lp[3]);
      if (NB == 2) vload<0, false>(bst[u][1], bl + (size_t)(2 * u4 + 1) * kd16);
      else vload<0, false>(bst[u][0], bl + (size_t)u4 * kd16);
    });
  }
  if (pos_next(pa)) rows_of(pa.tile / a.ntiles, ra);
  if (pos_next(pb)) bt = b_base(pb.tile);

  int rbuf = 0;                                         // LDS stage the MFMAs read; the stage written is rbuf + 2
  f32x4 bcur[NI], bnxt[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bcur[ni] = *(const f32x4*)(Bs + frow + ni * 16 * 64 + bsw[0]);

  for (int sg = 0; sg < nseg; ++sg) {
    int tile, it0, it1, kind;
    seg_of(sg, tile, it0, it1, kind);
    const int mt = tile / a.ntiles, nt = tile - mt * a.ntiles;
    const int m0w = mt * BM + wm * 64;                  // first row of this wave
    f32x4 acc[MI][NI];
    if (kind != HEAD) {
      const int m = m0w + lane;
      int po = -1, pr = 0;
      if (m < M) {
        const unsigned t = hnd::fdiv((unsigned)m, a.div_ow), ow_ = (unsigned)m - t * (unsigned)d.ow;
        const unsigned n_ = hnd::fdiv(t, a.div_oh), oh_ = t - n_ * (unsigned)d.oh;
        const int yr = (int)oh_ * d.y_sh + d.y_oh, yc = (int)ow_ * d.y_sw + d.y_ow;
        po = ((int)n_ * d.yh + yr) * d.yw + yc;
        if (d.res1_mode == 1)
          pr = ((int)n_ * d.res1_h + (yr * d.res1_h) / d.yh) * d.res1_w + (yc * d.res1_w) / d.yw;
      }
      __builtin_amdgcn_wave_barrier();                  // the wave's previous epilogue has read its tables
      rowoff[lane] = po;
      resoff[lane] = pr;
      __builtin_amdgcn_wave_barrier();
    }
    if (kind == TAIL) {
      // the head of this tile: accumulators parked by workgroup lb - 1 (which computed them FIRST, see above)
      if (tid == 0) {
        // Bounded (~2 s): the head was computed FIRST by its workgroup, so this does not spin in practice.  Should the
        // flag never come (a workspace that was not zero-filled once, a neighbour that faulted or was starved) the
        // launch must neither hang the GPU nor pass for correct: the wait gives up, raises the host-visible sticky
        // error word -- every later hnd_conv2d_igemm / hnd_sync_check then fails with HND_ERR_LAUNCH / HND_ERR_ASYNC --
        // and this tile's output is garbage by declaration.  Nothing is reset here (epoch flags, see above).
        int spin = 0;
        while (__hip_atomic_load(relay_f + (lb - 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          if (++spin >= a.spin_limit) {
            if (a.err) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const f32x4* src = (const f32x4*)(relay_p + (size_t)(lb - 1) * 16384) + tid;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_nontemporal_load(src + (mi * NI + ni) * 256);
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int it = it0; it < it1; ++it) {
      // this iteration's loads: A for the stream's next iteration, B for the one after that
      const float* lp[MI];
      unsigned okb;
      a_addr(ra, pa.it * 128, lp, okb);
      const float* bl0 = bt + (size_t)(2 * pb.it) * 64;
      const float* bl1 = bl0 + 64;
      const int cbase = TAPS ? (it * 128) % d.cin : it * 128;       // input channel of this iteration's first k
      f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pb_ = {0.f, 0.f, 0.f, 0.f}, psn = ps, pbn = pb_;
      if (PRO) {
        ps = *(const f32x4*)(pro + cbase + g4 * 4);
        pb_ = *(const f32x4*)(pro + d.cin + cbase + g4 * 4);
      }
      sfor<8>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value, u4 = u & 3;
        constexpr bool sacc = (u & 1) != 0;
        const float* bl = u < 4 ? bl0 : bl1;
        if constexpr (NB == 2) slot_wait<VM, sacc>(ring[u][0], ring[u][1], ring[u][2], ring[u][3], bst[u][0], bst[u][1]);
        else slot_wait<VM, sacc>(ring[u][0], ring[u][1], ring[u][2], ring[u][3], bst[u][0]);
        const unsigned okc = okr[u];
        okr[u] = okb;
        const int wofs = ((rbuf + 2) % NST) * STG + wpos0;
        // the next k group's B fragments: the same stage, or chunk 0 of the next one
        const int rnext = u4 == 3 ? (rbuf + 1) % NST : rbuf;
        const float* Bn = Bs + rnext * STG + frow + bsw[(u4 + 1) & 3];
        sfor<MI>([&](auto MIc) __attribute__((always_inline)) {
          constexpr int mi = decltype(MIc)::value;
          f32x4 av = ring[u][mi];
          if (PRO) {
            av = av * ps + pb_;
            av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor);
            av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
            if (TAPS) {                       // padding is a zero of the NORMALISED tensor
              const bool ok = (okc >> mi) & 1;
              av.x = ok ? av.x : 0.f; av.y = ok ? av.y : 0.f; av.z = ok ? av.z : 0.f; av.w = ok ? av.w : 0.f;
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bcur[ni][s], acc[mi][ni], 0, 0, 0);
            if (mi == 0) bnxt[s] = *(const f32x4*)(Bn + s * 16 * 64);
            if (mi == 1 && s < NB) *(f32x4*)(Bs + wofs + (NB * u4 + s) * 1024) = bst[u][s];
            if (PRO && mi == 2 && s == 0 && u < 7) {     // (every iteration starts from its own cbase)
              psn = *(const f32x4*)(pro + cbase + (u + 1) * 16 + g4 * 4);
              pbn = *(const f32x4*)(pro + d.cin + cbase + (u + 1) * 16 + g4 * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          vload<u * 64, sacc>(ring[u][mi], lp[mi]);
          if (NB == 2 && mi == 1) vload<0, false>(bst[u][0], bl + (size_t)(2 * u4) * kd16);
          if (NB == 2 && mi == 3) vload<0, false>(bst[u][1], bl + (size_t)(2 * u4 + 1) * kd16);
          if (NB == 1 && mi == 3) vload<0, false>(bst[u][0], bl + (size_t)u4 * kd16);
          __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bcur[ni] = bnxt[ni];
        if (PRO) { ps = psn; pb_ = pbn; }
        if (u4 == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (u4 == 3) rbuf = (rbuf + 1) % NST;
      });
      if (pos_next(pa)) rows_of(pa.tile / a.ntiles, ra);
      if (pos_next(pb)) bt = b_base(pb.tile);
    }

    if (kind == HEAD) {
      // park the accumulators for workgroup lb + 1 and raise the flag; no epilogue
      f32x4* dst = (f32x4*)(relay_p + (size_t)lb * 16384) + tid;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) __builtin_nontemporal_store(acc[mi][ni], dst + (mi * NI + ni) * 256);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid == 0 && a.dbg != 2) __hip_atomic_store(relay_f + lb, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // ---- epilogue of this tile; the loads of the next segment are already in flight
      const int col0 = nt * BN + wn * 64 + l16 * 4;     // hnd::chan_of_row of the wave's packed rows
      float es[NI], eb[NI], s1[NI], s2[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        es[ni] = epi[col0 + ni];
        eb[ni] = epi[d.cout + col0 + ni];
        s1[ni] = 0.f;
        s2[ni] = 0.f;
      }
      const bool full = vec_ok && (m0w + 64 <= M);
      if (a.dbg != 1) hnd::epilogue_tile<MI, NI>(d, acc, rowoff, resoff, 4 * g4, col0, es, eb, s1, s2, full);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stream's last (unused) prefetches land before the end
  launch_done();
}

int cu_count_() {
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

// The sticky error word: one int of pinned, device-mapped host memory per process.  A kernel raises it with a
// system-scope store, the host reads it without touching the GPU.
std::atomic<int*> g_err_host{nullptr};
int* g_err_dev = nullptr;

int* relay_error_word() {
  int* h = g_err_host.load(std::memory_order_acquire);
  if (h) return h;
  static std::atomic<bool> busy{false};
  bool expect = false;
  if (!busy.compare_exchange_strong(expect, true)) {
    while (!(h = g_err_host.load(std::memory_order_acquire))) {}
    return h;
  }
  int* hp = nullptr;
  if (hipHostMalloc((void**)&hp, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !hp) {
    (void)hipGetLastError();
    static int fallback = 0;                            // no pinned memory: timeouts stay unreported (err = null)
    g_err_dev = nullptr;
    g_err_host.store(&fallback, std::memory_order_release);
    return &fallback;
  }
  *hp = 0;
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
    (void)hipGetLastError();
    dp = nullptr;
  }
  g_err_dev = (int*)dp;
  g_err_host.store(hp, std::memory_order_release);
  return hp;
}

template <int WN, bool PRO, bool TAPS>
int launch_t(const hnd_conv_desc& d, const BstreamArgs& a, size_t lds, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = bstream_kernel<WN, PRO, TAPS>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(bstream<%d>) failed: %s", WN, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, d, a);
  return hnd::check_launch("hnd_conv2d_igemm(bstream)");
}

template <int WN>
int launch_w(const hnd_conv_desc& d, const BstreamArgs& a, size_t lds, int grid, hipStream_t stream) {
  const bool taps = d.kh * d.kw > 1 || d.bh != 0 || d.bw != 0;
  if (d.pro_scale) return taps ? launch_t<WN, true, true>(d, a, lds, grid, stream)
                               : launch_t<WN, true, false>(d, a, lds, grid, stream);
  return taps ? launch_t<WN, false, true>(d, a, lds, grid, stream) : launch_t<WN, false, false>(d, a, lds, grid, stream);
}

}  // namespace

namespace hnd {

// tiles of the launch and the persistent grid
static void bstream_grid(const hnd_conv_desc& d, int wn, int& mtiles, int& ntiles, int& grid) {
  const long long M = (long long)d.n * d.oh * d.ow;
  const int bm = 64 * (4 / wn), bn = 64 * wn;
  mtiles = (int)((M + bm - 1) / bm);
  ntiles = d.cout / bn;
  grid = (cu_count_() / 8) * 8;
}

// 0 = not taken, 1 = 256 x 64 block tile (one wave column), 2 = 128 x 128 (two)
int bstream_variant(const hnd_conv_desc& d) {
  const char* e = getenv("HND_BSTREAM");                // 0 = off (A/B)
  if (e && e[0] == '0') return 0;
  const bool all = hnd::debug_picker("bstream_all") > 0;   // every eligible launch (A/B tools, tests)
  if (d.stats != nullptr || d.cin % 32 != 0) return 0;
  if (d.kdim % 128 != 0 || d.kdim < 256) return 0;
  const bool taps = d.kh * d.kw > 1 || d.bh != 0 || d.bw != 0;
  if (taps ? (d.cin % 128 != 0 || d.kdim != d.kh * d.kw * d.cin) : (d.kdim != d.cin)) return 0;
  if (!taps && ((long long)(d.oh - 1) * d.sh >= d.h || (long long)(d.ow - 1) * d.sw >= d.w_)) return 0;
  if (d.cout % 64 != 0 || d.cout > 4096 || d.cin > 4096) return 0;
  if ((long long)d.n * d.h * d.w_ * d.cin >= (1ll << 32)) return 0;      // 32-bit pixel arithmetic on the tap path
  const int wn = d.cout % 128 == 0 ? 2 : 1;
  const int bm = 64 * (4 / wn);
  if (d.w_group_rows % bm != 0) return 0;
  const size_t lds = ((size_t)3 * 64 * wn * 64 + 2 * (size_t)d.cout + (d.pro_scale ? 2 * (size_t)d.cin : 0) + 4 * 128) *
                     sizeof(float);
  if (lds > 160 * 1024) return 0;
  if (all) return wn;
  // Where it is taken by default (round 3, batch 16, per-launch HIP events of the step, tiled -> this kernel): the
  // stride-2 3x3 convs over taps (1.39 -> 1.24 ms, 1.45 -> 1.24 ms), the K = 2048 1x1 convs and data gradients of layer4
  // (0.33 -> 0.29 ms), K = 1024 with >= 512 output channels (1.19 -> 1.12 ms).  The K = 1024 -> 256 launches of layer3
  // gain 3 % alone and nothing in the step (the teacher / FPN streams already fill the tiled kernel's partial last
  // round, and a persistent one-wave-per-SIMD kernel shares a CU with nobody): they stay on the tiled kernel, as does
  // K <= 512 without taps (the B-resident kernels).
  // (HND_DEBUG_PICKER=bstream_k1024: the K = 1024 -> 256 launches too; re-measured in round 5, profiles/r05_picker_ab.txt)
  if (!taps && (d.kdim < 1024 || (d.kdim < 2048 && d.cout < 512 && hnd::debug_picker("bstream_k1024") <= 0))) return 0;
  // The parity launches of a stride-2 data gradient (strided output, 1 / 2 / 4 taps: K = 128 ... 2048) are short: per
  // layer, all four on the tiled kernel beat three here + one tiled (round 4, per-launch events of the step: layer2.0
  // 0.925 -> 0.785 ms, layer3.0 0.801 -> 0.745, layer4.0 0.814 -> 0.789 with only its K = 2048 launch kept here).
  if (taps && d.y_sh > 1 && d.kdim < 2048 && hnd::debug_picker("bstream_parity") <= 0) return 0;
  // (K = 1024 -> 256 with >= 4 tiles per workgroup was tried again in round 4, with and without the shared trunk: 95.2 vs
  // 94.5-95.3 ms, nothing)
  // ... and only with at least one tile per workgroup (the relay's condition): a persistent kernel that leaves CUs
  // idle loses to the tiled kernel's small blocks (validation at batch 1: 148 -> 128 img/s when it took those too)
  int mtiles, ntiles, grid;
  bstream_grid(d, wn, mtiles, ntiles, grid);
  if ((long long)mtiles * ntiles < grid) return 0;
  return wn;
}

// the relay needs at least one whole tile of work per workgroup (see the kernel)
size_t bstream_workspace(const hnd_conv_desc& d) {
  const int wn = bstream_variant(d);
  if (wn == 0) return 0;
  int mtiles, ntiles, grid;
  bstream_grid(d, wn, mtiles, ntiles, grid);
  if ((long long)mtiles * ntiles < grid) return 0;
  return (size_t)grid * (16384 * sizeof(float) + sizeof(int)) + 16 * sizeof(int);     // sets, flags, counter + ticket
}

// the same sticky error word for the emulated B-streamed kernel (conv_bxs.hip)
int* relay_err_host() { return relay_error_word(); }
int* relay_err_dev() {
  (void)relay_error_word();
  return g_err_dev;
}

int relay_timeouts(int reset) {
  int* h = g_err_host.load(std::memory_order_acquire);
  if (!h) return 0;
  const int v = __atomic_load_n(h, __ATOMIC_RELAXED);
  if (reset) __atomic_store_n(h, 0, __ATOMIC_RELAXED);
  return v;
}

int launch_bstream(const hnd_conv_desc& d, hipStream_t stream) {
  const int wn = bstream_variant(d);
  if (wn == 0) {
    set_error("launch_bstream: descriptor not eligible");
    return HND_ERR_INVALID;
  }
  const int bn = 64 * wn;
  BstreamArgs a;
  int grid;
  const int* errw = relay_error_word();
  if (__atomic_load_n(errw, __ATOMIC_RELAXED) != 0) {
    set_error("hnd_conv2d_igemm(bstream): an earlier launch gave up waiting for a neighbour's partial tile (relay "
              "time-out): results since then are invalid; hnd_relay_timeouts(1) acknowledges");
    return HND_ERR_LAUNCH;
  }
  a.err = g_err_dev;
  a.spin_limit = 1 << 21;
  if (const char* e = getenv("HND_BSTREAM_SPIN")) a.spin_limit = atoi(e) > 0 ? atoi(e) : a.spin_limit;
  a.div_ow = make_fastdiv((unsigned)d.ow);
  a.div_oh = make_fastdiv((unsigned)d.oh);
  bstream_grid(d, wn, a.mtiles, a.ntiles, grid);
  a.kg8 = d.kdim / 128;
  a.dbg = getenv("HND_BSTREAM_DBG") ? atoi(getenv("HND_BSTREAM_DBG")) : 0;
  a.relay = (d.relay_ws && bstream_workspace(d) > 0) ? d.relay_ws : nullptr;
  const size_t lds = ((size_t)3 * bn * 64 + 2 * (size_t)d.cout + (d.pro_scale ? 2 * (size_t)d.cin : 0) + 4 * 128) *
                     sizeof(float);
  return wn == 2 ? launch_w<2>(d, a, lds, grid, stream) : launch_w<1>(d, a, lds, grid, stream);
}

}  // namespace hnd
