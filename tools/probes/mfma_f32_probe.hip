// Probe: sustained fp32 MFMA rate on this MI355X for the two exact-f32 shapes, on RANDOM operands (the chip lowers its
// clock under matrix load, MI355X_MICROARCH.md "DVFS give-back" (7): the clock it holds can depend on the MFMA shape).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f32_probe.hip -o tools/probes/mfma_f32_probe && ./mfma_f32_probe
// Per variant: TFLOP/s chip-wide and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ in, float* __restrict__ out, int iters,
                                             unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + i];
    b[i] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i];
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + j) & 7], b[k], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
      for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC * 4];
    for (int j = 0; j < NACC * 4; ++j)
      for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)              // same flops per iteration: 8 * NACC * 4096 = 4 * 4NACC * 2048
#pragma unroll
        for (int j = 0; j < NACC * 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + j) & 7], b[k], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC * 4; ++j)
      for (int r = 0; r < 4; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
  (void)lane;
}

template <int SHAPE, int NACC>
void run(const char* name, int blocks_per_cu, bool zeros) {
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  std::vector<float> h((size_t)blocks * 256 * 16);
  for (auto& v : h) v = zeros ? 0.f : (float)rand() / RAND_MAX - 0.5f;
  float *in, *out;
  unsigned long long* st;
  hipMalloc(&in, h.size() * 4);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&st, (size_t)blocks * 16);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
  hipEventRecord(e0);
  const int reps = 40;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs((size_t)blocks * 2);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double clk = 0;
  for (int b = 0; b < blocks; ++b) clk += (double)hs[b * 2] / (double)hs[b * 2 + 1] * 100.0;      // MHz
  clk /= blocks;
  const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * NACC * 4096.0 * reps;
  printf("%-34s %d blocks/CU %s: %7.1f TFLOP/s  in-kernel clock %6.0f MHz  -> %.1f %% of 64 flop/clk/SIMD\n", name,
         blocks_per_cu, zeros ? "zeros " : "random", flops / (ms * 1e-3) / 1e12, clk,
         100.0 * flops / (ms * 1e-3) / (1024.0 * 64.0 * clk * 1e6));
  hipFree(in); hipFree(out); hipFree(st);
}

int main() {
  for (int z = 0; z < 2; ++z)
    for (int bpc = 1; bpc <= 4; ++bpc) {          // 256-thread blocks: bpc waves per SIMD
      run<32, 4>("v_mfma_f32_32x32x2_f32  4 acc", bpc, z);
      run<16, 4>("v_mfma_f32_16x16x4_f32 16 acc", bpc, z);
    }
  return 0;
}
