// PROBE: operand / result lane maps of v_mfma_f32_16x16x32_bf16 as the bf16x3 probe assumes them, and its in-register
// fp32 -> (hi, mid, lo) split.  One wave; integer-valued data (exact in bf16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

__global__ void k(const float* A /*[16][32]*/, const float* B /*[32][16] as B[k][col]*/, float* D /*[16][16]*/, float* D2) {
  const int lane = threadIdx.x, l16 = lane & 15, g4 = lane >> 4;
  uint32_t ab[8], bb[8];
  for (int j = 0; j < 8; ++j) {
    ab[j] = __builtin_bit_cast(uint32_t, A[l16 * 32 + g4 * 8 + j]);
    bb[j] = __builtin_bit_cast(uint32_t, B[(g4 * 8 + j) * 16 + l16]);
  }
  u4 ap, bp;
  for (int i = 0; i < 4; ++i) {
    ap[i] = __builtin_amdgcn_perm(ab[2 * i + 1], ab[2 * i], 0x07060302u);
    bp[i] = (bb[2 * i] >> 16) | (bb[2 * i + 1] & 0xffff0000u);
  }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ap), __builtin_bit_cast(bf8, bp), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g4 * 4 + r) * 16 + l16] = c[r];
  // the three-plane split of non-integer A (B integer): D2 = sum of hi/mid/lo products with B
  f4 c2 = {0, 0, 0, 0};
  uint32_t hs[8], ms[8], ls[8];
  for (int j = 0; j < 8; ++j) {
    const float x = A[l16 * 32 + g4 * 8 + j] * 1.2345678f;
    const uint32_t xb = __builtin_bit_cast(uint32_t, x), hb = xb & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hb);
    const uint32_t mb = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    hs[j] = hb; ms[j] = mb; ls[j] = __builtin_bit_cast(uint32_t, r2);
  }
  u4 hp, mp, lp;
  for (int i = 0; i < 4; ++i) {
    hp[i] = __builtin_amdgcn_perm(hs[2 * i + 1], hs[2 * i], 0x07060302u);
    mp[i] = __builtin_amdgcn_perm(ms[2 * i + 1], ms[2 * i], 0x07060302u);
    lp[i] = __builtin_amdgcn_perm(ls[2 * i + 1], ls[2 * i], 0x07060302u);
  }
  c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, lp), __builtin_bit_cast(bf8, bp), c2, 0, 0, 0);
  c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, mp), __builtin_bit_cast(bf8, bp), c2, 0, 0, 0);
  c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, hp), __builtin_bit_cast(bf8, bp), c2, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D2[(g4 * 4 + r) * 16 + l16] = c2[r];
}

int main() {
  float A[16 * 32], B[32 * 16], D[256], D2[256];
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 13 - 6); B[i] = (float)((i * 5 + 1) % 11 - 5); }
  float *dA, *dB, *dD, *dD2;
  hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D); hipMalloc(&dD2, sizeof D2);
  hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD, dD2);
  hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost); hipMemcpy(D2, dD2, sizeof D2, hipMemcpyDeviceToHost);
  double e1 = 0, e1t = 0, e2 = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double s = 0, s2 = 0;
      for (int kk = 0; kk < 32; ++kk) { s += (double)A[r * 32 + kk] * B[kk * 16 + c]; s2 += (double)(A[r * 32 + kk] * 1.2345678f) * B[kk * 16 + c]; }
      e1 = fmax(e1, fabs(D[r * 16 + c] - s));
      e1t = fmax(e1t, fabs(D[c * 16 + r] - s));
      e2 = fmax(e2, fabs(D2[r * 16 + c] - s2) / (fabs(s2) + 1.0));
    }
  printf("integer data: max |D - A B| with D[row = 4 (l>>4) + reg][col = l & 15]: %g   (transposed reading: %g)\n", e1, e1t);
  printf("three-plane split of A (x 1.2345678), B integer: max rel err %g (fp32 eps 6e-8)\n", e2);
  return 0;
}
