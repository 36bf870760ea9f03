// Stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels, image stored NHWC4) from an LDS-staged input patch:
// custom/resnet.py:26-30,96 / torchvision ResNet.conv1, FrozenBatchNorm + ReLU in the epilogue.
//
// In the generic implicit-GEMM kernel every thread gathers ONE 16-byte tap per (pixel, tap) from global memory -- 49
// scattered reads per output pixel -- and the stem ran at 64 TFLOP/s (0.41 of the fp32 MFMA peak) although its GEMM
// shape (M = 4.3 M pixels, N = 64, K = 196) reaches 90-95 on the same kernel without the gather.  Here a workgroup
// owns an 8 x 32 tile of output pixels at a time (workgroups are persistent: two per CU walk the tiles, the patch
// of the next tile is in flight in registers while this one is on the matrix pipe): the 21 x 69 input patch it needs
// (23 KB) is copied to LDS once, coalesced, zero-filled outside the image; the packed weights (64 rows x 208, 54 KB, rows padded to 212 floats so the 16 rows of
// a fragment read fall on different banks) sit beside it; an MFMA A-fragment -- lane (l16, g4) = pixel l16 of a
// 16-pixel run, tap 4*kg + g4, 4 channels -- is then exactly ONE ds_read_b128 of the patch.  No k-loop staging, no
// barrier after the fill.
// The k order is the generic kernel's (k = 4*tap + channel; MFMA s of k group kg sums channel s of taps 4kg..4kg+3)
// and the epilogue expression is the same, so the result is bit-identical (tests/test_ops_gpu.py).  Round 4: the MFMAs of
// the pad channel (s = 3: zero weights for a 3-channel image) are not issued -- 39 instead of 52 per row group.
#include <type_traits>

#include "common.h"

#include <stdlib.h>

namespace {

using hnd::f32x4;

constexpr int TH = 8, TW = 32;                   // output tile of a workgroup (rows x cols)
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;  // input patch: 21 x 69 pixels of 4 floats
constexpr int KG = 13;                           // 16-deep k groups: 52 taps (49 real; the packed rows are zero beyond)
constexpr int WLD = 212;                         // LDS row stride of the weights (floats): 53 sixteen-byte slots

__global__ void __launch_bounds__(256, 2) stem7_kernel(const hnd_conv_desc d, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                           // [PH][PW][4]
  float* ws = smem + PH * PW * 4;                // [64][WLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int ntiles = d.n * tiles_x * tiles_y;

  // ---- once per workgroup: the weights (k < 208 of the 64 packed rows).  Workgroups are persistent (two per CU
  // walk the tiles with the grid's stride), so the 54 KB are not re-read for each of the 16 800 tiles of a batch.
  int w3 = 0;                                    // does any weight of the 4th input channel differ from 0?
  for (int e = tid; e < 64 * (KG * 4); e += 256) {
    const int r = e / (KG * 4), c = e - r * (KG * 4);
    const f32x4 wv = *(const f32x4*)(d.w + (size_t)r * d.kdim + c * 4);
    *(f32x4*)(ws + r * WLD + c * 4) = wv;
    w3 |= (wv.w != 0.f) ? 1 : 0;
  }
  // MFMA s of a k group multiplies channel s of four taps.  For the 3-channel image stored NHWC4 (the only user:
  // custom/resnet.py:26) the weights of channel 3 are exactly 0, so s = 3 -- a quarter of the matrix work -- adds 0 to
  // every accumulator and is not issued; a genuine 4-channel conv keeps it.
  const bool four = __syncthreads_or(w3) != 0;
  // this lane's tap of every k group: offset (floats) inside the patch; taps >= 49 read tap 48 (their weights are 0)
  int toff[KG];
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) {
    int t = 4 * kg + g4;
    t = t < 49 ? t : 48;
    toff[kg] = ((t / 7) * PW + (t % 7)) * 4;
  }
  // wave w owns output rows 2w, 2w+1 of the tile; row group mi = (row 2w + mi / 2, cols 16 * (mi & 1) + l16)
  const float* ap[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    ap[mi] = patch + ((2 * (2 * wave + (mi >> 1))) * PW + 2 * (16 * (mi & 1) + l16)) * 4;
  const float* bp = ws + l16 * WLD + g4 * 4;
  const int col0 = l16 * 4;
  float es[4], eb[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    es[ni] = d.epi_scale ? d.epi_scale[col0 + ni] : 1.f;
    eb[ni] = d.epi_shift ? d.epi_shift[col0 + ni] : 0.f;
  }

  // the patch of a tile: PH * PW = 1449 pixels over 256 threads = 6 float4 per thread, zero outside the image
  constexpr int PPT = (PH * PW + 255) / 256;
  f32x4 pre[PPT];
  auto fetch = [&](int tile) {                   // global -> registers (issued early, written to LDS late)
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
    const float* img = d.x + (size_t)n * d.h * d.w_ * 4;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int e = tid + 256 * q;
      const int py = e / PW, px = e - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = e < PH * PW && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w_;
      const f32x4 v = *(const f32x4*)(img + (ok ? ((size_t)iy * d.w_ + ix) * 4 : 0));
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      pre[q] = ok ? v : z;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int e = tid + 256 * q;
      if (e < PH * PW) *(f32x4*)(patch + e * 4) = pre[q];
    }
  };

  // The tile loop exists twice, for `four` as a compile-time constant: as a run-time test it put a branch behind every 12
  // MFMAs (4 per k group: 52 x 4 restarts of the matrix pipe per tile, round 5).
  auto run = [&](auto FOURc) __attribute__((always_inline)) {
  constexpr bool kFour = decltype(FOURc)::value;
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                             // the previous tile's fragment reads are done
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);       // in flight during this tile's MFMAs
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      f32x4 a[4], bq[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bq[ni] = *(const f32x4*)(bp + ni * 16 * WLD + kg * 16);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = *(const f32x4*)(ap[mi] + toff[kg]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s], bq[ni][s], acc[mi][ni], 0, 0, 0);
        if constexpr (kFour) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][3], bq[ni][3], acc[mi][ni], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: the lane holds pixels 4*g4 + i of each 16-pixel run and channels 4*l16 .. 4*l16 + 3
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int oy = oy0 + 2 * wave + (mi >> 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = ox0 + 16 * (mi & 1) + 4 * g4 + i;
        if (oy >= d.oh || ox >= d.ow) continue;
        f32x4 v;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const float x = acc[mi][ni][i] * es[ni] + eb[ni];
          v[ni] = d.relu ? fmaxf(x, 0.f) : x;
        }
        *(f32x4*)(d.y + (((size_t)n * d.yh + oy) * d.yw + ox) * (size_t)d.ldc + col0) = v;
      }
    }
  }
  };
  if (four) run(std::true_type{});
  else run(std::false_type{});
}


// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution (student conv1 is trainable: custom/resnet.py:26, autograd at
// src/mimic_runner.py:53):  dW[co][tap][c] = sum_pixels dy[p][co] * x[2p + tap][c].  The generic split-K kernel gathered
// one 16-byte tap per (pixel, tap) from global memory and ran at 51 TFLOP/s.  Here a workgroup walks 8 x 16 output
// tiles: dy of the tile is transposed into LDS ([co][pixel], so an MFMA A-fragment -- output channel l16, 4
// consecutive pixels -- is one ds_read_b128), the 21 x 37 input patch sits beside it and a B-fragment -- column
// (tap, c) = l16, the same 4 pixels -- is four ds_read_b32 two pixels apart.  The 64 x 208 accumulator stays in
// registers (each wave owns 3-4 of the 13 column tiles) across all tiles of the workgroup and is written ONCE as a slab
// in the generic kernel's layout; wgrad_reduce_kernel sums the slabs in fixed order (bit-reproducible).
constexpr int WTH = 8, WTW = 16;                     // output tile (rows x cols) = 128 pixels = 8 k groups of 16
constexpr int WPH = 2 * WTH + 5, WPW = 2 * WTW + 5;  // input patch 21 x 37
constexpr int DYLD = WTH * WTW + 4;                  // row stride of the transposed dy tile: 33 sixteen-byte slots

// NTN = column tiles: 10 (three real channels, channel-major columns) or 13 -- a template parameter: as a run-time value
// every "third owned tile" / "second shared tile" MFMA sat behind a branch of its own (16 per k group), and the matrix
// pipe restarted after each (round 5: 0.59 of peak)
template <int NTN>
__global__ void __launch_bounds__(256, 3) stem7_wgrad_kernel(const hnd_wgrad_desc d, const int tiles_x, const int tiles_y,
                                                             const int ncols_pad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                               // [WPH][WPW][4]
  float* dyt = smem + ((WPH * WPW * 4 + 3) & ~3);    // [64][DYLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int ntiles = d.n * tiles_x * tiles_y;

  // Columns of the GEMM are (tap, channel) pairs.  With three real channels (cin_real == 3: the NHWC4 image; the
  // reduce kernel reads only the real channels of a slab) they are enumerated channel-major, n = 49 c + tap: 147 columns
  // = 10 column tiles instead of the 13 of the tap-major order 4 tap + c, whose every 4th column is the pad channel.
  // Work split (round 4): a wave owns all 64 output channels of the column tiles wave, wave + 4 (, wave + 8), and ONE row
  // group (16 output channels: mi == wave) of the remaining tiles 8, 9 (tile 12 with four channels) -- 10 (13) units of
  // 16 x 16 per wave and k step, where 3 + 3 + 2 + 2 whole tiles left two waves waiting at the barrier.
  constexpr int ntn = NTN;
  constexpr int nown = ntn == 10 ? 2 : 3, nsh = ntn - 4 * nown;  // tiles owned whole / shared by row group
  auto col_of = [&](int nt, int& boff_, int& ncol_) {
    const int n = nt * 16 + l16;
    int t, c;
    if (ntn == 13) { t = n >> 2; c = n & 3; }
    else { c = n / 49; t = n - 49 * c; }
    const bool real = ntn == 13 ? t < 49 : n < 147;
    if (!real) { t = 48; c = ntn == 13 ? (n & 3) : 2; }           // padding columns: read something valid, never stored
    boff_ = ((t / 7) * WPW + (t % 7)) * 4 + c + 32 * g4;          // + 2 * (4 g4) pixels of 4 floats
    ncol_ = real ? 4 * t + c : -1;                                // column in the slab (the generic kernel's layout)
  };
  int boff[3], ncol[3], boffs[2], ncols[2];
#pragma unroll
  for (int q = 0; q < 3; ++q) col_of(wave + 4 * q, boff[q], ncol[q]);
#pragma unroll
  for (int j = 0; j < 2; ++j) col_of(4 * nown + j, boffs[j], ncols[j]);
  f32x4 acc[4][3], accs[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[mi][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  accs[0] = accs[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the next tile's input patch (777 pixels: 4 float4 per thread) and dy tile (128 pixels x 64 channels: 8 float4 per
  // thread) are requested before this tile's MFMAs and written to LDS after them (round 4: the wave spent 49 % of its
  // life parked at the fill of the tile it was about to compute, SQ_WAIT_ANY in profiles/r04_wgrad_sq_counters.txt)
  constexpr int PPT = (WPH * WPW + 255) / 256, DPT = (WTH * WTW) / 16;
  f32x4 pre_p[PPT], pre_d[DPT];
  const int c4 = tid & 15;
  auto fetch = [&](int tile) {
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int oy0 = ty * WTH, ox0 = tx * WTW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float* img = d.x + (size_t)n * d.h * d.w_ * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int e = tid + 256 * q;
      const int py = e / WPW, px = e - py * WPW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = e < WPH * WPW && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w_;
      const f32x4 v = *(const f32x4*)(img + (ok ? ((size_t)iy * d.w_ + ix) * 4 : 0));
      pre_p[q] = ok ? v : z;
    }
#pragma unroll
    for (int j = 0; j < DPT; ++j) {   // thread = (pixel group, 4 channels); 16 threads read one pixel's 256 bytes
      const int p = (tid >> 4) + 16 * j;             // pixel of the tile: row p / 16, col p % 16
      const int oy = oy0 + p / WTW, ox = ox0 + p % WTW;
      const bool ok = oy < d.oh && ox < d.ow;
      const f32x4 v = *(const f32x4*)(d.dy + (ok ? (((size_t)n * d.oh + oy) * d.ow + ox) * (size_t)d.ldy + c4 * 4 : 0));
      pre_d[j] = ok ? v : z;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int e = tid + 256 * q;
      if (e < WPH * WPW) *(f32x4*)(patch + e * 4) = pre_p[q];
    }
#pragma unroll
    for (int j = 0; j < DPT; ++j) {                  // dy tile, transposed: [channel][pixel]
      const int p = (tid >> 4) + 16 * j;
      dyt[(c4 * 4 + 0) * DYLD + p] = pre_d[j].x;
      dyt[(c4 * 4 + 1) * DYLD + p] = pre_d[j].y;
      dyt[(c4 * 4 + 2) * DYLD + p] = pre_d[j].z;
      dyt[(c4 * 4 + 3) * DYLD + p] = pre_d[j].w;
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                 // the previous tile's fragment reads are done
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);       // in flight during this tile's MFMAs
#pragma unroll
    for (int r = 0; r < WTH; ++r) {                  // one k group = the 16 pixels of tile row r
      f32x4 a[4], bq[3], bs[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = *(const f32x4*)(dyt + (mi * 16 + l16) * DYLD + r * WTW + 4 * g4);
      const f32x4 as_ = *(const f32x4*)(dyt + (wave * 16 + l16) * DYLD + r * WTW + 4 * g4);    // row group mi == wave
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float* bp = patch + boff[q] + 2 * r * WPW * 4;
        bq[q] = f32x4{bp[0], bp[8], bp[16], bp[24]};     // pixels 4 g4 + s: two input pixels (8 floats) apart
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* bp = patch + boffs[j] + 2 * r * WPW * 4;
        bs[j] = f32x4{bp[0], bp[8], bp[16], bp[24]};
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc[mi][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s_], bq[q][s_], acc[mi][q], 0, 0, 0);
          if (nown == 3) acc[mi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][s_], bq[2][s_], acc[mi][2], 0, 0, 0);
        }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        accs[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_[s_], bs[0][s_], accs[0], 0, 0, 0);
        if (nsh == 2) accs[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_[s_], bs[1][s_], accs[1], 0, 0, 0);
      }
    }
  }
  // ---- one slab per workgroup: [64][ncols_pad], column = 4 tap + c (the generic kernel's layout; with cin_real == 3 the
  // pad-channel columns are neither computed nor written -- wgrad_reduce_kernel never reads them)
  float* slab = d.slabs + (size_t)blockIdx.x * 64 * ncols_pad;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (q >= nown || ncol[q] < 0) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[(size_t)(mi * 16 + 4 * g4 + i) * ncols_pad + ncol[q]] = acc[mi][q][i];
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j >= nsh || ncols[j] < 0) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) slab[(size_t)(wave * 16 + 4 * g4 + i) * ncols_pad + ncols[j]] = accs[j][i];
  }
}

}  // namespace

namespace hnd {

static bool stem7_on() {                // HND_STEM7=0: the generic kernels (A/B, bit-identity tests)
  const char* e = getenv("HND_STEM7");
  return !(e && atoi(e) == 0);
}

bool stem7_applies(const hnd_conv_desc& d) {
  if (!stem7_on()) return false;
  return d.cin == 4 && d.kh == 7 && d.kw == 7 && d.sh == 2 && d.sw == 2 && d.dh == 1 && d.dw == 1 && d.bh == -3 &&
         d.bw == -3 && d.cout == 64 && d.ldc % 4 == 0 && d.kdim >= 4 * KG * 4 && !d.pro_scale && !d.res1 && !d.res2 &&
         !d.mask && !d.stats && d.w_group_rows == 0 && d.y_sh == 1 && d.y_sw == 1 && d.y_oh == 0 && d.y_ow == 0 &&
         d.yh == d.oh && d.yw == d.ow && (uintptr_t)d.y % 16 == 0 && (uintptr_t)d.x % 16 == 0;
}

int launch_stem7(const hnd_conv_desc& d, hipStream_t stream) {
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t lds = ((size_t)PH * PW * 4 + 64 * WLD) * sizeof(float);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)stem7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(stem7) failed: %s", hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_done[dev & 63] = true;
  }
  const int tiles_x = (d.ow + TW - 1) / TW, tiles_y = (d.oh + TH - 1) / TH;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int ntiles = d.n * tiles_x * tiles_y;
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;        // persistent: two workgroups per CU walk the tiles
  hipLaunchKernelGGL(stem7_kernel, dim3((unsigned)grid), dim3(256), lds, stream, d, tiles_x, tiles_y);
  return check_launch("hnd_conv2d_igemm(stem7)");
}


bool stem7_wgrad_applies(const hnd_wgrad_desc& d) {
  if (!stem7_on()) return false;
  return d.cin == 4 && d.kh == 7 && d.kw == 7 && d.stride == 2 && d.pad == 3 && d.cout == 64 && d.ldy % 4 == 0 &&
         !d.pro_scale && d.groups <= 1 && d.oh == (d.h + 6 - 7) / 2 + 1 && d.ow == (d.w_ + 6 - 7) / 2 + 1 &&
         (uintptr_t)d.dy % 16 == 0 && (uintptr_t)d.x % 16 == 0;
}

int stem7_wgrad_blocks(const hnd_wgrad_desc& d) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int ntiles = d.n * ((d.ow + WTW - 1) / WTW) * ((d.oh + WTH - 1) / WTH);
  return ntiles < 3 * cus ? ntiles : 3 * cus;        // persistent: three workgroups per CU walk the tiles
}

// writes stem7_wgrad_blocks(d) slabs of [64][ncols_pad] floats into d.slabs
int launch_stem7_wgrad(const hnd_wgrad_desc& d, int ncols_pad, hipStream_t stream) {
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t lds = ((size_t)((WPH * WPW * 4 + 3) & ~3) + 64 * DYLD) * sizeof(float);
  if (!attr_done[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)stem7_wgrad_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)stem7_wgrad_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(stem7_wgrad) failed: %s", hipGetErrorString(e));
      return HND_ERR_LAUNCH;
    }
    attr_done[dev & 63] = true;
  }
  const int tiles_x = (d.ow + WTW - 1) / WTW, tiles_y = (d.oh + WTH - 1) / WTH;
  if (d.cin_real == 3)
    hipLaunchKernelGGL(stem7_wgrad_kernel<10>, dim3((unsigned)stem7_wgrad_blocks(d)), dim3(256), lds, stream, d, tiles_x,
                       tiles_y, ncols_pad);
  else
    hipLaunchKernelGGL(stem7_wgrad_kernel<13>, dim3((unsigned)stem7_wgrad_blocks(d)), dim3(256), lds, stream, d, tiles_x,
                       tiles_y, ncols_pad);
  return check_launch("hnd_conv2d_wgrad(stem7)");
}

}  // namespace hnd
