// Shared helpers for libhnd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hnd_hip.h"

namespace hnd {

void set_error(const char* fmt, ...);
int check_launch(const char* what);   // hipGetLastError -> status

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// floor(x / d) for 0 <= x < 2^31 with a precomputed multiplier (d >= 1)
struct FastDiv {
  unsigned mul, shr, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.shr = 0; return f; }
  unsigned l = 0;
  while ((1u << l) < d) ++l;                       // l = ceil(log2 d)
  unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  f.mul = (unsigned)m;
  f.shr = l;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, const FastDiv f) {
  if (f.d == 1) return x;
  unsigned t = __umulhi(x, f.mul);
  return (t + ((x - t) >> 1)) >> (f.shr - 1);
}

// Packed GEMM-operand row r (hnd_pack_weights, hnd_wino*_weights) holds output channel chan_of_row(r): inside every
// group of 64 rows the four 16-row MFMA tiles are interleaved, so the four accumulator tiles a lane of the implicit-
// GEMM kernel owns (rows 16*ni + lane%16 of its wave's 64) are four CONSECUTIVE output channels.
__host__ __device__ inline int chan_of_row(int r) { return (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3); }

// Workgroup b runs on XCD b % 8 (each XCD has its own L2).  Kernels whose neighbouring workgroups read overlapping
// data (the Winograd input transforms: tile (ty, tx) shares two of its eight columns with (ty, tx + 1)) give
// consecutive LOGICAL blocks to one XCD, so the shared part is an L2 hit instead of a second trip to HBM.  Bijective for
// any grid.  (Measured and not used for the 3x3 stride-2 max pooling: 0.64 -> 0.70 ms.)
__device__ __forceinline__ int xcd_contiguous_block() {
  const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, slot = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// HND_DEBUG_PICKER: the ONE switch behind which the kernel pickers' experiment / test overrides live (A/B tools and the
// bit-identity tests force a variant with it; nothing in normal use sets it).  Comma-separated keys, optionally key=value:
//   bres_all          every launch the B-resident kernels can take, not only where they were measured to win
//   bstream_all       the same for the B-streamed kernel
//   igemm_tile=0..3   block tile of the tiled kernel (128x128, 128x64, 64x128, 64x64)
//   wgrad_ring_taps   the ring weight-gradient kernel also for tap (direct 2x2) problems
//   bstream_k1024 / bstream_parity   the B-streamed kernel also for K = 1024 -> 256 launches / for stride-2 parity launches
//   bxs_wn1                          (B-streamed emulation kernel) the 256 x 64 tile also where the output has 128-column blocks
// Returns -1 when `key` is absent, its value (1 without "=value") otherwise.  Read per call: in-process A/B.
inline int debug_picker(const char* key) {
  const char* e = getenv("HND_DEBUG_PICKER");
  if (!e) return -1;
  const size_t n = strlen(key);
  for (const char* p = e; *p;) {
    const char* q = p;
    while (*q && *q != ',') ++q;
    if ((size_t)(q - p) >= n && strncmp(p, key, n) == 0 && (p[n] == ',' || p[n] == 0 || p[n] == '=')) return p[n] == '=' ? atoi(p + n + 1) : 1;
    p = *q ? q + 1 : q;
  }
  return -1;
}

#define HND_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      hnd::set_error(__VA_ARGS__);            \
      return HND_ERR_INVALID;                 \
    }                                         \
  } while (0)

}  // namespace hnd
