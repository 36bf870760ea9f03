// Index bookkeeping of the validation path without library kernels: what torchvision 0.4.2's
// RegionProposalNetwork.filter_proposals / RoIHeads.postprocess_detections / MultiScaleRoIAlign do with
// torch.topk, torch.sort and torch.nonzero (rpn.py, roi_heads.py, poolers.py), here as
//   * ordered compaction of a predicate (the indices where it holds, ascending == torch.nonzero(...).squeeze(1)):
//     score > threshold, level == l, box at least min_size wide and high, flag != 0;
//   * a STABLE descending argsort of fp32 keys (== torch.sort(descending=True, stable=True)[1]; top-k = its head):
//     up to 4096 keys one workgroup sorts (key, index) pairs bitonically in LDS; beyond that a 4-pass LSD radix sort on
//     the order-preserving 32-bit image of the float, stable by construction, so ties keep ascending index order.
// All byte / index work: results are exact.
#include "common.h"

namespace {

constexpr int CMP_THREADS = 1024;

struct PredFlag { const unsigned char* f; __device__ bool operator()(long long i) const { return f[i] != 0; } };
struct PredGt { const float* x; float thr; __device__ bool operator()(long long i) const { return x[i] > thr; } };
struct PredEq { const long long* x; long long v; __device__ bool operator()(long long i) const { return x[i] == v; } };
struct PredMinSize {
  const float* b; float m;
  __device__ bool operator()(long long i) const {
    const hnd::f32x4 q = *(const hnd::f32x4*)(b + 4 * i);
    return (q.z - q.x >= m) && (q.w - q.y >= m);
  }
};

// one workgroup walks the range 1024 elements at a time: ballot ranks inside a wave, wave totals through LDS
template <class Pred>
__global__ void __launch_bounds__(CMP_THREADS) compact_kernel(const Pred pred, const long long n, long long* __restrict__ out,
                                                              long long* __restrict__ count) {
  __shared__ int wave_tot[CMP_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long base = 0;
  for (long long start = 0; start < n; start += CMP_THREADS) {
    const long long i = start + tid;
    const bool p = i < n && pred(i);
    const unsigned long long bal = __ballot(p);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < CMP_THREADS / 64; ++w) {
      const int t = wave_tot[w];
      before += w < wave ? t : 0;
      total += t;
    }
    if (p) out[base + before + rank] = i;
    base += total;
    __syncthreads();
  }
  if (tid == 0) *count = base;
}

template <class Pred>
int launch_compact(const Pred& pred, long long n, long long* out, long long* count, hipStream_t s, const char* what) {
  hipLaunchKernelGGL(compact_kernel<Pred>, dim3(1), dim3(CMP_THREADS), 0, s, pred, n, out, count);
  return hnd::check_launch(what);
}

// order-preserving image of a float, inverted: ascending unsigned order == descending float order
// (-0.0 compares equal to +0.0 and every NaN sorts as the largest value, as torch.sort does)
__device__ __forceinline__ unsigned desc_key(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;
  if ((u & 0x7fffffffu) > 0x7f800000u) u = 0x7fc00000u;
  const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}

// ---- n <= 4096: bitonic sort of (key << 32 | index) in LDS, one workgroup
constexpr int BIT_N = 4096;
__global__ void __launch_bounds__(1024) bitonic_argsort_kernel(const float* __restrict__ keys, const int n,
                                                               long long* __restrict__ order) {
  __shared__ unsigned long long v[BIT_N];
  const int tid = threadIdx.x;
  for (int i = tid; i < BIT_N; i += 1024)
    v[i] = i < n ? (((unsigned long long)desc_key(keys[i]) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= BIT_N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < BIT_N / 2; t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;     // the pair this thread owns
        const bool up = (lo & k) == 0;
        const unsigned long long a = v[lo], b = v[hi];
        if ((a > b) == up) { v[lo] = b; v[hi] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 1024) order[i] = (long long)(unsigned)(v[i] & 0xffffffffull);
}

// ---- n > 4096: LSD radix sort, 8 bits per pass, one chunk of RCH elements per workgroup
constexpr int RCH = 2048;
__global__ void __launch_bounds__(256) radix_hist_kernel(const float* __restrict__ fkeys, const unsigned* __restrict__ ukeys,
                                                         const int n, const int shift, const int nblk,
                                                         int* __restrict__ hist) {        // hist[digit][block]
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lo = blockIdx.x * RCH, hi = min(n, lo + RCH);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned k = fkeys ? desc_key(fkeys[i]) : ukeys[i];
    atomicAdd(&h[(k >> shift) & 255], 1);
  }
  __syncthreads();
  hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256) radix_scan_kernel(int* __restrict__ hist, const int nblk) {
  __shared__ int tot[256];
  const int d = threadIdx.x;
  int run = 0;
  for (int b = 0; b < nblk; ++b) {          // exclusive prefix inside the digit, over the blocks in order
    const int c = hist[d * nblk + b];
    hist[d * nblk + b] = run;
    run += c;
  }
  tot[d] = run;
  __syncthreads();
  int base = 0;
  for (int e = 0; e < d; ++e) base += tot[e];
  for (int b = 0; b < nblk; ++b) hist[d * nblk + b] += base;
}

// one wave per chunk, 64 elements at a time in order: lanes with the same digit find each other with 8 ballots; the
// rank inside the peer group keeps the pass stable
__global__ void __launch_bounds__(64) radix_scatter_kernel(const float* __restrict__ fkeys, const unsigned* __restrict__ ukeys,
                                                           const unsigned* __restrict__ idx_in, const int n, const int shift,
                                                           const int nblk, const int* __restrict__ offs,
                                                           unsigned* __restrict__ keys_out, unsigned* __restrict__ idx_out) {
  __shared__ int cnt[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) cnt[d] = offs[d * nblk + blockIdx.x];
  __builtin_amdgcn_wave_barrier();
  const int lo = blockIdx.x * RCH, hi = min(n, lo + RCH);
  for (int start = lo; start < hi; start += 64) {
    const int i = start + lane;
    const bool ok = i < hi;
    const unsigned k = ok ? (fkeys ? desc_key(fkeys[i]) : ukeys[i]) : 0u;
    const unsigned id = ok ? (idx_in ? idx_in[i] : (unsigned)i) : 0u;
    const int dg = (k >> shift) & 255;
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(ok && ((dg >> b) & 1));
      peers &= ((dg >> b) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    int pos = 0;
    if (ok) pos = cnt[dg] + rank;
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) cnt[dg] += __popcll(peers);       // one lane per peer group
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      keys_out[pos] = k;
      idx_out[pos] = id;
    }
  }
}

__global__ void widen_kernel(const unsigned* __restrict__ idx, const int n, long long* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = (long long)idx[i];
}

}  // namespace

extern "C" {

int hnd_nonzero_u8(const uint8_t* flags, int64_t n, int64_t* out, int64_t* count, void* stream) {
  HND_REQUIRE(count && (n == 0 || (flags && out)), "hnd_nonzero_u8: null pointer");
  return launch_compact(PredFlag{flags}, n, (long long*)out, (long long*)count, hnd::as_stream(stream), "hnd_nonzero_u8");
}

int hnd_nonzero_gt_f32(const float* x, int64_t n, float threshold, int64_t* out, int64_t* count, void* stream) {
  HND_REQUIRE(count && (n == 0 || (x && out)), "hnd_nonzero_gt_f32: null pointer");
  return launch_compact(PredGt{x, threshold}, n, (long long*)out, (long long*)count, hnd::as_stream(stream),
                        "hnd_nonzero_gt_f32");
}

int hnd_nonzero_eq_i64(const int64_t* x, int64_t n, int64_t value, int64_t* out, int64_t* count, void* stream) {
  HND_REQUIRE(count && (n == 0 || (x && out)), "hnd_nonzero_eq_i64: null pointer");
  return launch_compact(PredEq{(const long long*)x, (long long)value}, n, (long long*)out, (long long*)count,
                        hnd::as_stream(stream), "hnd_nonzero_eq_i64");
}

int hnd_nonzero_min_size(const float* boxes, int64_t n, float min_size, int64_t* out, int64_t* count, void* stream) {
  HND_REQUIRE(count && (n == 0 || (boxes && out)), "hnd_nonzero_min_size: null pointer");
  HND_REQUIRE((uintptr_t)boxes % 16 == 0, "hnd_nonzero_min_size: boxes must be 16-byte aligned");
  return launch_compact(PredMinSize{boxes, min_size}, n, (long long*)out, (long long*)count, hnd::as_stream(stream),
                        "hnd_nonzero_min_size");
}

size_t hnd_argsort_desc_workspace(int64_t n) {
  if (n <= BIT_N) return 16;
  const size_t nblk = (size_t)((n + RCH - 1) / RCH);
  return 4 * (size_t)n * sizeof(unsigned) + 256 * nblk * sizeof(int) + 64;
}

int hnd_argsort_desc_f32(const float* keys, int64_t n, int64_t* order, void* workspace, void* stream) {
  if (n <= 0) return HND_OK;
  HND_REQUIRE(keys && order, "hnd_argsort_desc_f32: null pointer");
  HND_REQUIRE(n < (1ll << 31), "hnd_argsort_desc_f32: at most 2^31 - 1 keys");
  hipStream_t s = hnd::as_stream(stream);
  if (n <= BIT_N) {
    hipLaunchKernelGGL(bitonic_argsort_kernel, dim3(1), dim3(1024), 0, s, keys, (int)n, (long long*)order);
    return hnd::check_launch("hnd_argsort_desc_f32(bitonic)");
  }
  HND_REQUIRE(workspace != nullptr, "hnd_argsort_desc_f32: workspace required beyond %d keys", BIT_N);
  const int nn = (int)n, nblk = (nn + RCH - 1) / RCH;
  unsigned* ka = (unsigned*)workspace;
  unsigned* kb = ka + n;
  unsigned* ia = kb + n;
  unsigned* ib = ia + n;
  int* hist = (int*)(ib + n);
  for (int pass = 0; pass < 4; ++pass) {
    const float* fk = pass == 0 ? keys : nullptr;
    const unsigned* uk = pass == 0 ? nullptr : ((pass & 1) ? kb : ka);
    const unsigned* ii = pass == 0 ? nullptr : ((pass & 1) ? ib : ia);
    unsigned* ko = (pass & 1) ? ka : kb;
    unsigned* io = (pass & 1) ? ia : ib;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(256), 0, s, fk, uk, nn, 8 * pass, nblk, hist);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(256), 0, s, hist, nblk);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(64), 0, s, fk, uk, ii, nn, 8 * pass, nblk, hist, ko, io);
  }
  hipLaunchKernelGGL(widen_kernel, dim3((nn + 255) / 256), dim3(256), 0, s, ia, nn, (long long*)order);   // pass 3 -> a
  return hnd::check_launch("hnd_argsort_desc_f32(radix)");
}

}  // extern "C"
