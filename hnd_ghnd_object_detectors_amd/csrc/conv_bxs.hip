// B-STREAMED persistent GEMM with fp32 EMULATED on the bf16 matrix pipe of gfx950 (round 6): the launches the B-resident
// emulation kernel (conv_bx3.hip) cannot take -- convolutions over TAPS (the stride-2 3x3 convs of layers 2-4 and the parity
// launches of their data gradients, the head's direct 2x2 convs with BatchNorm(+ReLU) on load and BatchNorm statistics /
// backward sums in the epilogue), long K (1x1 convs of layer4, K = 2048), strided outputs (the downsample data gradients)
// and fp32 masks.  Arithmetic = conv_bx3.hip's (every fp32 operand the exact sum of three bf16 planes by truncation, the six
// plane products with i + j <= 2 on v_mfma_f32_16x16x32_bf16, smallest terms first inside every 32-deep k step, fp32
// accumulate; k steps ascending over (tap, channel)); structure = conv_bstream.hip's:
//   * B: the pre-split weight image (hnd_pack_bf16x3s: per 64-column slice and 64-k stage three planes of [64 rows][64 k]
//     bf16, XOR-swizzled 16-byte chunks) streams through THREE LDS stages by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave
//     instruction, no registers), requested two stages ahead; one workgroup barrier per stage (64 k = 192 MFMAs per wave);
//   * A: fp32 straight from global memory, a lane owns row l16 of a 16-row group and 8 consecutive k of a 32-deep step (two
//     global_load_dwordx4 into the accumulator half of the register file), ring of 4 k steps = one 128-k iteration ahead;
//     out-of-range taps read a page of zeros; the BatchNorm(+ReLU) prologue is applied during the split (padding stays 0);
//   * the split of k step s + 1 into bf16 planes rides between the MFMAs of k step s (conv_bx3.hip);
//     (measured and not kept, round 6: the stage's DMA pieces issued one per MFMA tile instead of at the top of the step --
//     4 % slower; the next step's first B fragments read a step early -- nothing)
//   * tiles 128 x 128 (two wave columns) or 256 x 64, every wave 64 x 64; the stream-K relay of conv_bstream.hip (a
//     workgroup whose share ends inside a tile parks the accumulators, its neighbour continues the same k chain) keeps
//     every CU busy whatever the tile count; epilogue = conv_epilogue.h (any operand set, statistics, backward sums).
// Counter note (vmcnt, in-order retire): per k step s = 4 i + u every lane issues, in this order, [u even: the NBL LDS-DMA
// pieces of stage 2 i + 2 + u / 2] then the 8 ring loads of step s + 4.  The wait for ring slot s + 1 (issued at step s - 3)
// therefore allows 24 + (u even ? 2 : 1) NBL younger operations, the wait for stage 2 i + u / 2 (issued at step s - 4)
// 32 + NBL.  Epilogue loads / stores between two tiles are younger still: they only make these waits stricter.
#include <atomic>

#include "common.h"
#include "conv_epilogue.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace hnd {
int* relay_err_host();            // conv_bstream.hip: the process-wide sticky error word (host view)
int* relay_err_dev();             // ... and its device view (nullptr when no pinned memory was to be had)
}  // namespace hnd

namespace {

using hnd::f32x4;
using hnd::FastDiv;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ float g_bxs_zero_page[128];     // source of out-of-range taps: a lane reads 2 x 16 B at g4 * 32 + {0, 128} B

struct BxsArgs {
  FastDiv div_ow, div_oh;     // m -> (n, oh, ow)
  FastDiv div_cin, div_kw;    // k -> (tap, channel), tap -> (i, j)
  int mtiles, ntiles;         // tile grid
  int nit;                    // iterations of 128 k per tile
  int spin_limit;             // polls of a relay flag before the wait gives up
  float* relay;               // stream-K relay workspace (hnd_conv2d_igemm_workspace), or null: tiles round-robin
  int* err;                   // host-visible sticky error word
};

template <int N, class F, int... I>
__device__ __forceinline__ void zfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void zfor(F&& f) {
  zfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ring slots live in the accumulator half of the register file (conv_bx3.hip)
template <int OFF>
__device__ __forceinline__ void aload(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void await8(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6, f32x4& a7) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
               : "n"(CNT));
}
// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 * lane (M0 is compiler-reserved: saved,
// written and restored inside the one statement)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// WN = wave columns: block tile (64 * 4 / WN) x (64 * WN), every wave a 64 x 64 tile of 4 x 4 MFMA tiles.
// PRO: BatchNorm(+ReLU) of the input on load (pro_scale / pro_shift per input channel); TAPS: kh * kw > 1 or padding.
template <int WN, bool PRO, bool TAPS>
__global__ void __launch_bounds__(256, 1) bxs_kernel(const hnd_conv_desc d, const BxsArgs a) {
  // BIG (the 256 x 64 tile: one 64-column slice per workgroup): an LDS stage holds a whole 128-k iteration of the slice and
  // two stages alternate -- ONE workgroup barrier per 128 k instead of two (round 6: SQ counters showed the 128 x 128 build's
  // waves parked at waits / barriers 23-30 % of their life against 8-19 % for the barrier-free B-resident kernel)
  constexpr bool BIG = WN == 1;
  constexpr int WM = 4 / WN, BM = 64 * WM, BN = 64 * WN, MI = 4, NI = 4, NST = BIG ? 2 : 3;
  constexpr int PLANE = 64 * 64;            // bf16 elements of one plane of a slice stage
  constexpr int SLICE = 3 * PLANE;          // one 64-column slice, one 64-k stage: 24 KB
  constexpr int STG = 2 * SLICE;            // bf16 elements of an LDS stage: two slices x 64 k, or one slice x 128 k (48 KB)
  constexpr int NBL = 12;                   // LDS-DMA pieces per lane and stage
  // waits (see the counter note): 128 x 128 build as described there; BIG: the pieces of iteration i + 1 are issued at the
  // top of iteration i (after its barrier), so the stage wait has the 4 x 8 ring loads of one iteration behind it, a slot
  // wait at steps 0 .. 2 three steps' ring loads + the pieces, at step 3 (slot 0 of the next iteration) three steps' loads
  constexpr int W_ODD = 24 + NBL, W_EVEN = 24 + 2 * NBL, W_STAGE = BIG ? 32 : 32 + NBL;
  constexpr int W_BIG_012 = 24 + NBL, W_BIG_3 = 24;
  static_assert(W_EVEN <= 63 && W_STAGE <= 63, "vmcnt holds 6 bits");
  extern __shared__ __attribute__((aligned(16))) uint16_t Bs[];     // [NST][WN][3 planes][64 rows][64 k]
  float* pro = (float*)(Bs + NST * STG);                            // [2][cin] prologue scale, shift
  int* tabs = (int*)(pro + (PRO ? 2 * d.cin : 0));                  // [4 waves][2][64]: output / res1 pixel of the wave's rows
  float* red = (float*)(tabs + 4 * 128);                            // [WM][2][BN] statistics of the tile (d.stats only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int l16 = lane & 15, g4 = lane >> 4;
  int* rowoff = tabs + wave * 128;
  int* resoff = rowoff + 64;
  const unsigned lds0 = (unsigned)(size_t)Bs;                       // LDS byte address of stage 0 (flat -> local: low 32 bits)
  const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);

  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);       // blocks of one XCD are consecutive
  const int T = a.mtiles * a.ntiles;
  const int M = d.n * d.oh * d.ow;
  const int nit = a.nit;

  // ---- this workgroup's segments: (tile, first iteration, end iteration, kind) -- conv_bstream.hip's relay
  enum { FULL = 0, HEAD = 1, TAIL = 2 };
  int nseg, first_full = 0, nfull = 0, tA = 0, offA = 0, tB = 0, offB = 0;
  bool has_head = false;
  if (a.relay) {
    const long long U = (long long)T * nit;
    const long long u0 = U * lb / G, u1 = U * (lb + 1) / G;
    tA = (int)(u0 / nit); offA = (int)(u0 - (long long)tA * nit);
    tB = (int)(u1 / nit); offB = (int)(u1 - (long long)tB * nit);
    has_head = offB > 0;
    first_full = tA + (offA > 0 ? 1 : 0);
    nfull = tB - first_full;
    nseg = (has_head ? 1 : 0) + nfull + (offA > 0 ? 1 : 0);
  } else {
    if (lb >= T) return;
    nfull = nseg = (T - lb + G - 1) / G;                // tiles lb, lb + G, ...
  }
  float* relay_p = a.relay;                             // [G][16][256] float4 accumulator sets
  int* relay_f = (int*)(a.relay + (size_t)G * 16384);   // [G] flags (launch epochs), [G] = launch counter, [G + 1] = ticket
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
    if (!a.relay) { tile = lb + i * G; it0 = 0; it1 = nit; kind = FULL; return; }
    if (has_head) {
      if (i == 0) { tile = tB; it0 = 0; it1 = offB; kind = HEAD; return; }
      --i;
    }
    if (i < nfull) { tile = first_full + i; it0 = 0; it1 = nit; kind = FULL; return; }
    tile = tA; it0 = offA; it1 = nit; kind = TAIL;
  };

  if (PRO)
    for (int c = tid; c < d.cin; c += 256) {
      pro[c] = d.pro_scale[c];
      pro[d.cin + c] = d.pro_shift[c];
    }

  // ---- A side.  Per row group the lane's source row: 1x1 a pointer; taps (first pixel of the image, ih0, iw0)
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
        r.p[mi] = d.x + ((size_t)((int)n_ * d.h + ih0) * (size_t)d.w_ + iw0) * (size_t)d.cin + (size_t)(g4 * 8);
      }
    }
  };
  // the lane's four load addresses for the 64 k that start at `kofs` (one tap: cin % 64 == 0), the validity of the tap per
  // row group and the input channel of the first k
  auto a_addr = [&](const Rows& r, int kofs, const float* (&lp)[MI], unsigned& okbits, int& chan0) {
    okbits = 0xfu;
    chan0 = kofs;
    if (!TAPS) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) lp[mi] = r.p[mi] + kofs;
    } else {
      const int tap = (int)hnd::fdiv((unsigned)kofs, a.div_cin), ci0 = kofs - tap * d.cin;
      const int ti = (int)hnd::fdiv((unsigned)tap, a.div_kw), tj = tap - ti * d.kw;
      chan0 = ci0;
      okbits = 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int ih = r.ih0[mi] + ti * d.dh, iw = r.iw0[mi] + tj * d.dw;
        const bool ok = (unsigned)ih < (unsigned)d.h && (unsigned)iw < (unsigned)d.w_;
        const float* src = d.x + (size_t)(unsigned)(r.pix[mi] + ih * d.w_ + iw) * (size_t)d.cin + (size_t)(ci0 + g4 * 8);
        lp[mi] = ok ? src : (const float*)g_bxs_zero_page + g4 * 8;
        okbits |= (ok ? 1u : 0u) << mi;
      }
    }
  };
  // ---- B side: stage t (64 k) of 64-column slice s of weight group g is the contiguous 24 KB block
  // ((g * nsl + s) * 2 nit + t) of the image; piece j of lane `tid` = bytes j * 4096 + tid * 16 of it
  const int nsl = d.cout / 64;
  auto b_base = [&](int tile) -> const uint16_t* {
    const int mt = tile / a.ntiles, nt = tile - mt * a.ntiles;
    const size_t grp = d.w_group_rows > 0 ? (size_t)((mt * BM) / d.w_group_rows) : 0;
    return d.w_bf16x3s + ((grp * (size_t)nsl + (size_t)(nt * WN)) * (size_t)(2 * nit)) * (size_t)SLICE + (size_t)tid * 8;
  };
  const size_t slice_stride = (size_t)(2 * nit) * (size_t)SLICE;     // elements between two slices of one group
  // stage_in_tile counts 64-k stages; BIG: the two consecutive stages of an iteration are 48 contiguous KB of the image
  auto b_issue = [&](const uint16_t* bt, int stage_in_tile, int lds_stage) {
    const unsigned dst0 = lds0 + (unsigned)lds_stage * (unsigned)(STG * 2) + wave_u * 1024u;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint16_t* src = bt + (BIG ? (size_t)0 : (size_t)w * slice_stride) + (size_t)(stage_in_tile + (BIG ? w : 0)) * (size_t)SLICE;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        glds16(src + j * 2048, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + (unsigned)(w * SLICE * 2 + j * 4096))));
    }
  };
  const float relu_floor = d.pro_relu ? 0.f : -INFINITY;
  const bool vec_ok = (d.ldc % NI == 0) &&
                      (((uintptr_t)d.y | (uintptr_t)d.res1 | (uintptr_t)d.res2 | (uintptr_t)d.mask) % (4 * NI) == 0);

  // ---- the load stream runs ONE iteration (128 k: 4 k steps, 2 stages) ahead of the MFMAs and walks the segment list on
  // its own (past its end it stays on the last iteration: harmless extra loads)
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
  Pos pn;
  pos_init(pn);
  Rows rn;
  rows_of(pn.tile / a.ntiles, rn);
  const uint16_t* bt = b_base(pn.tile);

  // per iteration: two halves of 64 k, each with its own address set (cin = 64: two taps per iteration)
  const float* lpn[2][MI];            // the iteration being loaded (of the one being computed only validity / channels are kept)
  unsigned okc[2], okn[2];
  int chc[2], chn[2];
  a_addr(rn, pn.it * 128, lpn[0], okn[0], chn[0]);
  a_addr(rn, pn.it * 128 + 64, lpn[1], okn[1], chn[1]);

  f32x4 ring[4][MI][2];
  // ---- fill, in the steady state's issue order: B(0) A(0) A(1) B(1) A(2) A(3); BIG: B(0, 1) A(0) A(1) A(2) A(3)
  b_issue(bt, 2 * pn.it, 0);
  zfor<4>([&](auto U) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value, hf = u >> 1, o = (u & 1) * 128;
    if constexpr (u == 2 && !BIG) b_issue(bt, 2 * pn.it + 1, 1);
    zfor<MI>([&](auto I) __attribute__((always_inline)) {
      constexpr int mi = decltype(I)::value;
      aload<o>(ring[u][mi][0], lpn[hf][mi]);
      aload<o + 16>(ring[u][mi][1], lpn[hf][mi]);
    });
  });
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    okc[hf] = okn[hf];
    chc[hf] = chn[hf];
  }
  if (pos_next(pn)) {
    rows_of(pn.tile / a.ntiles, rn);
    bt = b_base(pn.tile);
  }
  a_addr(rn, pn.it * 128, lpn[0], okn[0], chn[0]);
  a_addr(rn, pn.it * 128 + 64, lpn[1], okn[1], chn[1]);

  // planes of the k step about to be computed / being made: [parity][hi / mid / lo][row group]: 4 dwords = 8 bf16
  uint32_t pl[2][3][MI][4];
  // prologue constants of the lane's 8 channels of the step being split
  f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pb0 = {0.f, 0.f, 0.f, 0.f}, pb1 = pb0;
  auto pro_fetch = [&](int chan) {
    if (PRO) {
      ps0 = *(const f32x4*)(pro + chan + g4 * 8);
      ps1 = *(const f32x4*)(pro + chan + g4 * 8 + 4);
      pb0 = *(const f32x4*)(pro + d.cin + chan + g4 * 8);
      pb1 = *(const f32x4*)(pro + d.cin + chan + g4 * 8 + 4);
    }
  };
  // value j (0..7) of the lane's 8 consecutive k of a ring slot, through the prologue
  auto pro_apply = [&](float x, int j, bool ok) -> float {
    if (PRO) {
      const float s = j < 4 ? ps0[j & 3] : ps1[j & 3], b = j < 4 ? pb0[j & 3] : pb1[j & 3];
      x = fmaxf(x * s + b, relu_floor);
      if (TAPS) x = ok ? x : 0.f;               // padding is a zero of the NORMALISED tensor
    }
    return x;
  };
  if (PRO) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the prologue table is complete
  }
  // step 0 of the first iteration: split before the loop (a wait in the place of step "-1")
  await8<BIG ? W_BIG_3 : W_ODD>(ring[0][0][0], ring[0][0][1], ring[0][1][0], ring[0][1][1], ring[0][2][0], ring[0][2][1],
                                ring[0][3][0], ring[0][3][1]);
  pro_fetch(chc[0]);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = ring[0][mi][j >> 1];
      const bool ok = (okc[0] >> mi) & 1;
      const float x0 = pro_apply((j & 1) ? v.z : v.x, 2 * j, ok), x1 = pro_apply((j & 1) ? v.w : v.y, 2 * j + 1, ok);
      const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
      const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
      const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
      const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
      pl[0][0][mi][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
      pl[0][1][mi][j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
      pl[0][2][mi][j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }

  int rbuf = 0;                                         // LDS stage the MFMAs read; stage rbuf + 2 is being filled
  const int swz = l16 >> 1;                             // chunk c of row r sits at position c ^ ((r >> 1) & 7)
  const int frow = (BIG ? 0 : wn * SLICE) + l16 * 64;   // the lane's fragment row (ni = 0) inside a stage (BIG: of its first half)

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
      // the head of this tile: accumulators parked by workgroup lb - 1 (which computed them FIRST); bounded wait, a
      // time-out raises the sticky host-visible error word (conv_bstream.hip, hnd_relay_timeouts)
      if (tid == 0) {
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
      zfor<4>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value, hf = u >> 1, par = u & 1, u1 = (u + 1) & 3;
        if constexpr (BIG ? (u == 0) : ((u & 1) == 0)) {
          // this lane's pieces of stage `rbuf` have landed (issued two stages / one iteration ago); then everybody's
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(W_STAGE) : "memory");
          int wst = rbuf + (BIG ? 1 : 2);
          wst = wst >= NST ? wst - NST : wst;
          b_issue(bt, 2 * pn.it + (BIG ? 0 : hf), wst);         // the same half (BIG: the whole) of the NEXT iteration
        }
        // slot u was split during the previous step: refill it with step u of the next iteration
        zfor<MI>([&](auto I) __attribute__((always_inline)) {
          constexpr int mi = decltype(I)::value;
          aload<par * 128>(ring[u][mi][0], lpn[hf][mi]);
          aload<par * 128 + 16>(ring[u][mi][1], lpn[hf][mi]);
        });
        await8<BIG ? (u == 3 ? W_BIG_3 : W_BIG_012) : ((u & 1) ? W_ODD : W_EVEN)>(ring[u1][0][0], ring[u1][0][1], ring[u1][1][0], ring[u1][1][1], ring[u1][2][0],
                                         ring[u1][2][1], ring[u1][3][0], ring[u1][3][1]);
        // the step being split (u + 1 of this iteration, or step 0 of the next): its prologue constants and tap validity
        constexpr int hs = ((u + 1) >> 1) & 1;            // half of the iteration the split step lies in
        const unsigned oks = u == 3 ? okn[0] : okc[hs];
        pro_fetch((u == 3 ? chn[0] : chc[hs]) + ((u + 1) & 1) * 32);
        const uint16_t* stage = Bs + rbuf * STG + frow + (BIG ? hf * SLICE : 0);
        const int pos = ((par * 4 + g4) ^ swz) * 8;
        bf8 bcur[3], bnxt[3];
        bcur[0] = *(const bf8*)(stage + pos);
        bcur[1] = *(const bf8*)(stage + PLANE + pos);
        bcur[2] = *(const bf8*)(stage + 2 * PLANE + pos);
        zfor<NI>([&](auto NIc) __attribute__((always_inline)) {
          constexpr int ni = decltype(NIc)::value;
          if constexpr (ni + 1 < NI) {
            const uint16_t* br = stage + (ni + 1) * 16 * 64 + pos;
            bnxt[0] = *(const bf8*)(br); bnxt[1] = *(const bf8*)(br + PLANE); bnxt[2] = *(const bf8*)(br + 2 * PLANE);
          }
          zfor<MI>([&](auto MIc) __attribute__((always_inline)) {
            constexpr int mi = decltype(MIc)::value;
            auto frag = [&](int q) __attribute__((always_inline)) {
              const u32x4 t = {pl[par][q][mi][0], pl[par][q][mi][1], pl[par][q][mi][2], pl[par][q][mi][3]};
              return __builtin_bit_cast(bf8, t);
            };
            const bf8 ah = frag(0), am = frag(1), al = frag(2);
            // one pair of the NEXT step's elements rides between this tile's six MFMAs: piece p = ni * 4 + mi -> row group
            // p / 4, pair p % 4
            constexpr int p = ni * 4 + mi, rg = p >> 2, j = p & 3;
            const f32x4 v = ring[u1][rg][j >> 1];
            const bool ok = (oks >> rg) & 1;
            const float x0 = pro_apply((j & 1) ? v.z : v.x, 2 * j, ok), x1 = pro_apply((j & 1) ? v.w : v.y, 2 * j + 1, ok);
            f32x4 cacc = acc[mi][ni];
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
        if constexpr (BIG ? (u == 3) : ((u & 1) == 1)) rbuf = rbuf + 1 >= NST ? 0 : rbuf + 1;
      });
      // the loaded iteration becomes the computed one; the stream moves on
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        okc[hf] = okn[hf];
        chc[hf] = chn[hf];
      }
      if (pos_next(pn)) {
        rows_of(pn.tile / a.ntiles, rn);
        bt = b_base(pn.tile);
      }
      a_addr(rn, pn.it * 128, lpn[0], okn[0], chn[0]);
      a_addr(rn, pn.it * 128 + 64, lpn[1], okn[1], chn[1]);
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
      if (tid == 0) __hip_atomic_store(relay_f + lb, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // ---- epilogue of this tile; the loads of the next segment are already in flight
      const int col0 = nt * BN + wn * 64 + l16 * 4;     // hnd::chan_of_row of the wave's packed rows
      float es[NI], eb[NI], s1[NI], s2[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        es[ni] = d.epi_scale ? d.epi_scale[col0 + ni] : 1.f;
        eb[ni] = d.epi_shift ? d.epi_shift[col0 + ni] : 0.f;
        s1[ni] = 0.f;
        s2[ni] = 0.f;
      }
      const bool full = vec_ok && (m0w + 64 <= M);
      hnd::epilogue_tile<MI, NI, true>(d, acc, rowoff, resoff, 4 * g4, col0, es, eb, s1, s2, full);
      if (d.stats) {
        // per 128-row statistics tile and channel: the four row groups of the wave, then the two waves of the tile
        const int cl = wn * 64 + l16 * 4;
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
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // WM = 2: one statistics tile (waves wm = 0, 1); WM = 4: two (waves 0, 1 and 2, 3)
        for (int e = tid; e < (WM / 2) * BN; e += 256) {
          const int half = e / BN, c = e - half * BN;
          const long long st_tile = (long long)mt * (WM / 2) + half;
          if (st_tile * 128 < M) {
            float* st = d.stats + (size_t)st_tile * 2 * d.cout + nt * BN + c;
            st[0] = red[((2 * half) * 2 + 0) * BN + c] + red[((2 * half + 1) * 2 + 0) * BN + c];
            st[d.cout] = red[((2 * half) * 2 + 1) * BN + c] + red[((2 * half + 1) * 2 + 1) * BN + c];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // `red` is free for the next tile
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stream's last (unused) prefetches land before the end
  launch_done();
}

// packed fp32 operand [groups][rows_pad][K] -> the stream image: per (group, 64-row slice, 64-k stage) three planes
// [64 rows][64 k] of bf16, chunk c (8 values) of row r at position c ^ ((r >> 1) & 7)
__global__ void pack_bxs_kernel(const float* __restrict__ w, uint16_t* __restrict__ img, int rows_pad, int K, int groups,
                                long long group_stride, long long total) {
  const int nsl = rows_pad / 64, nst = K / 64;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    long long t = e / K;
    const int row = (int)(t % rows_pad), g = (int)(t / rows_pad);
    const float x = w[(size_t)g * (size_t)group_stride + (size_t)row * K + k];
    const uint32_t xb = __float_as_uint(x), hb = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const int s = row / 64, r = row % 64, st = k / 64, kk = k % 64, c = kk >> 3, pos = c ^ ((r >> 1) & 7);
    uint16_t* o = img + ((((size_t)g * nsl + s) * nst + st) * 3) * 4096 + (size_t)r * 64 + pos * 8 + (kk & 7);
    o[0] = (uint16_t)(hb >> 16);
    o[4096] = (uint16_t)(mb >> 16);
    o[8192] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

int cu_count_bxs() {
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

template <int WN, bool PRO, bool TAPS>
int launch_t(const hnd_conv_desc& d, const BxsArgs& a, size_t lds, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = bxs_kernel<WN, PRO, TAPS>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      hnd::set_error("hipFuncSetAttribute(bxs<%d>) failed: %s", WN, hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, d, a);
  return hnd::check_launch("hnd_conv2d_igemm(bxs)");
}

template <int WN>
int launch_w(const hnd_conv_desc& d, const BxsArgs& a, size_t lds, int grid, hipStream_t stream) {
  const bool taps = d.kh * d.kw > 1 || d.bh != 0 || d.bw != 0;
  if (d.pro_scale) return taps ? launch_t<WN, true, true>(d, a, lds, grid, stream)
                               : launch_t<WN, true, false>(d, a, lds, grid, stream);
  return taps ? launch_t<WN, false, true>(d, a, lds, grid, stream) : launch_t<WN, false, false>(d, a, lds, grid, stream);
}

size_t bxs_lds_bytes(const hnd_conv_desc& d, int wn) {
  return (size_t)(wn == 1 ? 2 : 3) * 2 * 3 * 4096 * sizeof(uint16_t) + (d.pro_scale ? 2 * (size_t)d.cin : 0) * sizeof(float) +
         4 * 128 * sizeof(int) + (size_t)(4 / wn) * 2 * 64 * wn * sizeof(float);
}

}  // namespace

namespace hnd {

static void bxs_grid(const hnd_conv_desc& d, int wn, int& mtiles, int& ntiles, int& grid) {
  const long long M = (long long)d.n * d.oh * d.ow;
  const int bm = 64 * (4 / wn), bn = 64 * wn;
  mtiles = (int)((M + bm - 1) / bm);
  ntiles = d.cout / bn;
  grid = (cu_count_bxs() / 8) * 8;
}

// 0 = not taken (no stream image attached, or a shape the kernel does not cover), 1 = 256 x 64 block tile, 2 = 128 x 128.
// Attaching hnd_conv_desc.w_bf16x3s ASKS for the kernel (the host attaches it by LAYER: hnd_bf16x3s_recommended).
int bxs_variant(const hnd_conv_desc& d) {
  if (!d.w_bf16x3s) return 0;
  if (d.cin % 64 != 0 || d.kdim % 128 != 0 || d.kdim < 128) return 0;
  const bool taps = d.kh * d.kw > 1 || d.bh != 0 || d.bw != 0;
  if (taps ? (d.kdim != d.kh * d.kw * d.cin) : (d.kdim != d.cin)) return 0;
  if (!taps && ((long long)(d.oh - 1) * d.sh >= d.h || (long long)(d.ow - 1) * d.sw >= d.w_)) return 0;
  if (d.cout % 64 != 0 || d.cout > 4096 || d.cin > 4096) return 0;
  if ((long long)d.n * d.h * d.w_ * d.cin >= (1ll << 32)) return 0;      // 32-bit pixel arithmetic on the tap path
  // 128 x 128 tiles where the output has whole 128-column blocks, else 256 x 64 (HND_DEBUG_PICKER=bxs_wn1: always the
  // latter -- A/B of the one-barrier-per-128-k build)
  const int wn = (d.cout % 128 == 0 && hnd::debug_picker("bxs_wn1") <= 0) ? 2 : 1;
  const int bm = 64 * (4 / wn);
  if (d.w_group_rows % bm != 0) return 0;
  if (d.stats && d.cout != d.ldc && d.bwd_x) return 0;
  if (bxs_lds_bytes(d, wn) > 160 * 1024) return 0;
  return wn;
}

// the relay needs at least one whole tile of work per workgroup (conv_bstream.hip)
size_t bxs_workspace(const hnd_conv_desc& d) {
  const int wn = bxs_variant(d);
  if (wn == 0) return 0;
  int mtiles, ntiles, grid;
  bxs_grid(d, wn, mtiles, ntiles, grid);
  if ((long long)mtiles * ntiles < grid) return 0;
  return (size_t)grid * (16384 * sizeof(float) + sizeof(int)) + 16 * sizeof(int);     // sets, flags, counter + ticket
}

int launch_bxs(const hnd_conv_desc& d, hipStream_t stream) {
  const int wn = bxs_variant(d);
  if (wn == 0) {
    set_error("launch_bxs: descriptor not eligible");
    return HND_ERR_INVALID;
  }
  BxsArgs a;
  int grid;
  const int* errw = relay_err_host();
  if (__atomic_load_n(errw, __ATOMIC_RELAXED) != 0) {
    set_error("hnd_conv2d_igemm(bxs): an earlier launch gave up waiting for a neighbour's partial tile (relay time-out): "
              "results since then are invalid; hnd_relay_timeouts(1) acknowledges");
    return HND_ERR_LAUNCH;
  }
  a.err = relay_err_dev();
  a.spin_limit = 1 << 21;
  if (const char* e = getenv("HND_BSTREAM_SPIN")) a.spin_limit = atoi(e) > 0 ? atoi(e) : a.spin_limit;
  a.div_ow = make_fastdiv((unsigned)d.ow);
  a.div_oh = make_fastdiv((unsigned)d.oh);
  a.div_cin = make_fastdiv((unsigned)d.cin);
  a.div_kw = make_fastdiv((unsigned)d.kw);
  bxs_grid(d, wn, a.mtiles, a.ntiles, grid);
  a.nit = d.kdim / 128;
  a.relay = (d.relay_ws && bxs_workspace(d) > 0) ? d.relay_ws : nullptr;
  const size_t lds = bxs_lds_bytes(d, wn);
  return wn == 2 ? launch_w<2>(d, a, lds, grid, stream) : launch_w<1>(d, a, lds, grid, stream);
}

}  // namespace hnd

// By LAYER, never by batch (like hnd_bf16x3_recommended): rows one image contributes priced at 16 images per GPU.  Settled
// by per-launch HIP events of the step (profiles/r06_bxs_shapes.txt, r06_per_launch_events.txt): the kernel wins 1.2-1.5x
// from K = 256 on; at K = 128 (the masked conv1 data gradients of layer2, one-tap parity launches) the launch is HBM-bound
// either way and the tiled kernel keeps it.
extern "C" int hnd_bf16x3s_recommended(int64_t rows_per_image, int kdim, int cout, int taps) {
  (void)taps;
  if (rows_per_image <= 0 || cout <= 0 || cout % 64 != 0 || kdim % 128 != 0) return 0;
  if (const char* e = getenv("HND_BXS")) if (e[0] == '0') return 0;       // A/B: the native B-streamed / tiled kernels
  return (rows_per_image * 16 >= 16384 && kdim >= 256) ? 1 : 0;
}

extern "C" size_t hnd_pack_bf16x3s_elems(int rows_pad, int kdim, int groups) {
  if (rows_pad <= 0 || rows_pad % 64 != 0 || kdim <= 0 || kdim % 128 != 0 || groups < 1) return 0;
  return (size_t)groups * (size_t)rows_pad * (size_t)kdim * 3;
}

extern "C" int hnd_pack_bf16x3s(const float* w_packed, uint16_t* img, int rows_pad, int kdim, int groups, int64_t group_stride,
                                void* stream) {
  HND_REQUIRE(w_packed && img && rows_pad > 0 && rows_pad % 64 == 0 && kdim > 0 && kdim % 128 == 0 && groups >= 1 &&
                  (groups == 1 || group_stride >= (int64_t)rows_pad * kdim),
              "hnd_pack_bf16x3s: bad arguments (rows_pad %% 64, kdim %% 128)");
  const long long total = (long long)groups * rows_pad * kdim;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_bxs_kernel, dim3((unsigned)blocks), dim3(256), 0, hnd::as_stream(stream), w_packed, img, rows_pad,
                     kdim, groups, (long long)group_stride, total);
  return hnd::check_launch("hnd_pack_bf16x3s");
}
