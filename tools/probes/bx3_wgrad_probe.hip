// PROBE (not product code): would the bf16x3 emulation pay for the WEIGHT GRADIENTS?  VERDICT r5 item 2 names them; the
// library runs them on fp32 MFMA (csrc/conv_wgrad_ring.hip, 0.76 of the fp32 peak).
//
//   S[g][ca][cb] = sum_k A[g][k][ca] * B[g][k][cb]      (g = Winograd component, k = tile, ca = dy channel, cb = x channel)
//
// both operands [k][channel] with the channel contiguous, both split into three bf16 planes on the fly.
// v_mfma_f32_16x16x32_bf16 wants 8 CONSECUTIVE k of one channel per lane, so a lane loads single dwords: channel
// c0 + lane % 16, pixels k0 + 8 (lane / 16) + j -- a wave instruction moves four 64-byte segments.  No LDS.  Deliberately
// plain (compiler-scheduled loads, the next 32-pixel step prefetched into registers): a floor for what a counted ring
// would reach, to decide whether that kernel is worth writing.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/bx3_wgrad_probe.hip -o /tmp/bx3_wgrad_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

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

struct Planes { bf8 h, m, l; };

template <bool SPLIT>
__device__ __forceinline__ Planes split8(const float (&x)[8]) {
  uint32_t hs[4], ms[4], ls[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (SPLIT) split_pair(x[2 * j], x[2 * j + 1], hs[j], ms[j], ls[j]);
    else { hs[j] = __builtin_amdgcn_perm(__float_as_uint(x[2 * j + 1]), __float_as_uint(x[2 * j]), 0x07060302u); ms[j] = hs[j]; ls[j] = hs[j]; }
  }
  Planes p;
  const u4 th = {hs[0], hs[1], hs[2], hs[3]}, tm = {ms[0], ms[1], ms[2], ms[3]}, tl = {ls[0], ls[1], ls[2], ls[3]};
  p.h = __builtin_bit_cast(bf8, th); p.m = __builtin_bit_cast(bf8, tm); p.l = __builtin_bit_cast(bf8, tl);
  return p;
}

// wave tile 64 AH x 64 BH, workgroup 2 x 2 waves.  SPLIT = false: planes = raw top halves (timing only: no vector work)
template <int AH, int BH, bool SPLIT>
__global__ void __launch_bounds__(256, 1) wg_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ slab, int K, int CA, int CB, int tiles_b,
                                                    int blocks_per_split, int total_blocks) {
  constexpr int NA = 4 * AH, NB = 4 * BH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  int bid = blockIdx.x;
  const int tiles = (CA / (128 * AH)) * tiles_b;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int ta = bid / tiles_b, tb = bid - ta * tiles_b;
  const int ca0 = ta * 128 * AH + (wave >> 1) * 64 * AH, cb0 = tb * 128 * BH + (wave & 1) * 64 * BH;
  const int blk0 = split * blocks_per_split;
  int nblk = total_blocks - blk0;
  if (nblk > blocks_per_split) nblk = blocks_per_split;
  const float* pa = A + ((size_t)g * K + (size_t)blk0 * 32 + 8 * q) * CA + ca0 + r;
  const float* pb = B + ((size_t)g * K + (size_t)blk0 * 32 + 8 * q) * CB + cb0 + r;
  f4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[i][n] = f4{0.f, 0.f, 0.f, 0.f};
  float la[NA][8], lb[NB][8];
  auto load = [&](float (&xa)[NA][8], float (&xb)[NB][8]) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) xa[i][j] = pa[(size_t)j * CA + 16 * i];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[n][j] = pb[(size_t)j * CB + 16 * n];
  };
  if (nblk > 0) load(la, lb);
  for (int s = 0; s < nblk; ++s) {
    float na[NA][8], nb[NB][8];
    pa += (size_t)32 * CA;
    pb += (size_t)32 * CB;
    if (s + 1 < nblk) load(na, nb);
    Planes bp[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bp[n] = split8<SPLIT>(lb[n]);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const Planes ap = split8<SPLIT>(la[i]);
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        f4 c = acc[i][n];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.l, bp[n].h, c, 0, 0, 0);      // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.h, bp[n].l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.m, bp[n].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.m, bp[n].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.h, bp[n].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap.h, bp[n].h, c, 0, 0, 0);
        acc[i][n] = c;
      }
    }
    if (s + 1 < nblk) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) la[i][j] = na[i][j];
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 8; ++j) lb[n][j] = nb[n][j];
    }
  }
  // C/D layout: column = lane % 16 (cb), row = 4 (lane / 16) + reg (ca)
  float* out = slab + (((size_t)split * gridDim.y + g) * CA) * CB;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(size_t)(ca0 + 16 * i + 4 * q + e) * CB + cb0 + 16 * n + r] = acc[i][n][e];
}

__global__ void reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int splits) {
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = slab[e];
  for (int k = 1; k < splits; ++k) s += slab[(size_t)k * n + e];
  out[e] = s;
}

template <int AH, int BH, bool SPLIT>
static float run(const float* A, const float* B, float* slab, float* out, int G, int K, int CA, int CB, int splits, int reps) {
  const int tiles_b = CB / (128 * BH), tiles = (CA / (128 * AH)) * tiles_b, total_blocks = K / 32;
  const int bps = (total_blocks + splits - 1) / splits;
  const long long n = (long long)G * CA * CB;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < reps + 2; ++it) {
    if (it == 2) CK(hipEventRecord(e0));
    hipLaunchKernelGGL((wg_kernel<AH, BH, SPLIT>), dim3(tiles * splits, G), dim3(256), 0, 0, A, B, slab, K, CA, CB, tiles_b, bps, total_blocks);
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, slab, out, n, splits);
  }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static void fill(std::vector<float>& v, uint32_t seed) {
  uint64_t s = seed * 2654435761ull + 12345;
  for (auto& x : v) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const float u1 = ((s >> 40) + 1) / 16777217.0f;
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const float u2 = (s >> 40) / 16777216.0f;
    x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
}

int main() {
  // (1) correctness on a small problem against fp64
  {
    const int G = 2, K = 2048, CA = 128, CB = 256, splits = 3;
    std::vector<float> hA((size_t)G * K * CA), hB((size_t)G * K * CB), hO((size_t)G * CA * CB);
    fill(hA, 1); fill(hB, 2);
    float *A, *B, *slab, *out;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4));
    CK(hipMalloc(&slab, hO.size() * 4 * splits)); CK(hipMalloc(&out, hO.size() * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    run<1, 2, true>(A, B, slab, out, G, K, CA, CB, splits, 1);
    CK(hipMemcpy(hO.data(), out, hO.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int g = 0; g < G; ++g)
      for (int a = 0; a < CA; a += 7)
        for (int b = 0; b < CB; b += 5) {
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)hA[((size_t)g * K + k) * CA + a] * (double)hB[((size_t)g * K + k) * CB + b];
          const double d = hO[((size_t)g * CA + a) * CB + b] - ref;
          num += d * d; den += ref * ref;
        }
    printf("correctness (G=%d K=%d %dx%d, 3 splits): rel-L2 vs fp64 %.3e\n", G, K, CA, CB, sqrt(num / den));
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(slab)); CK(hipFree(out));
  }
  // (2) timing on the step's Winograd-domain weight gradients at batch 16
  struct Shape { const char* name; int G, K, CA, CB; double native_ms; };
  const Shape shapes[] = {
      {"conv7 F(6x6,2x2) 256x256, 49 comps x 31008 tiles", 49, 31008, 256, 256, 1.526},
      {"conv6 F(6x6,2x2) 256x128, 49 comps x 31008 tiles", 49, 31008, 256, 128, 0.873},
  };
  for (const Shape& s : shapes) {
    float *A, *B, *slab, *out;
    const size_t na = (size_t)s.G * s.K * s.CA, nb = (size_t)s.G * s.K * s.CB, no = (size_t)s.G * s.CA * s.CB;
    std::vector<float> h(1 << 22);
    fill(h, 7);
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&slab, no * 4 * 8)); CK(hipMalloc(&out, no * 4));
    for (size_t o = 0; o < na; o += h.size()) CK(hipMemcpy(A + o, h.data(), (na - o < h.size() ? na - o : h.size()) * 4, hipMemcpyHostToDevice));
    for (size_t o = 0; o < nb; o += h.size()) CK(hipMemcpy(B + o, h.data(), (nb - o < h.size() ? nb - o : h.size()) * 4, hipMemcpyHostToDevice));
    const double gf = 2.0 * s.G * (double)s.K * s.CA * s.CB / 1e9;
    for (int splits : {1, 2, 3, 4, 6}) {
      float t1, t0;
      if (s.CB >= 256) { t1 = run<1, 2, true>(A, B, slab, out, s.G, s.K, s.CA, s.CB, splits, 10); t0 = run<1, 2, false>(A, B, slab, out, s.G, s.K, s.CA, s.CB, splits, 10); }
      else { t1 = run<1, 1, true>(A, B, slab, out, s.G, s.K, s.CA, s.CB, splits, 10); t0 = run<1, 1, false>(A, B, slab, out, s.G, s.K, s.CA, s.CB, splits, 10); }
      printf("%-52s splits %d: %.3f ms = %.1f TF-eq (no split work: %.3f ms)   native fp32 MFMA in the step: %.3f ms = %.1f TF\n",
             s.name, splits, t1, gf / t1, t0, s.native_ms, gf / s.native_ms);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(slab)); CK(hipFree(out));
  }
  return 0;
}
