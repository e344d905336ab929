// a2: predicate head — out[P,K] = sigmoid(x[P,F] @ W[K,F]^T + b) on fp32 MFMA
// (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces RelationPredictor.forward (reference lib/modeling/model.py:85-88).
// Both operands are K-contiguous ("NT" GEMM) with awkward sizes (F = 11070 is
// not a multiple of 4, K = 132 not a multiple of 16): tails are zero-filled in
// the LDS stage, never padded in HBM.  K = 132 fits 9 MFMA column blocks of 16
// (144, 8 % waste) so one workgroup tile covers every predicate.
//
// The problem is small (0.16 - 3 GFLOP) and bound by fetching x and W once, so
// the F dimension is split over workgroups (split-K) to put every CU on the
// fetch; partial tiles go to fp32 slabs in the caller's workspace and a second
// tiny kernel reduces them in a fixed order (bitwise reproducible, no atomics)
// and applies bias + sigmoid.
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;         // rows (pairs) per workgroup tile
constexpr int NBLK = 9;        // 16-wide column blocks per tile
constexpr int TN = 16 * NBLK;  // 144 columns (predicates) per tile
constexpr int KC = 32;         // F elements per LDS stage
constexpr int LDS_STRIDE = KC + 2;  // 34: (2*row + k) % 32 -> conflict-free fragment reads
constexpr int THREADS = 256;
constexpr int MAX_SPLIT = 64;

__global__ __launch_bounds__(THREADS) void linear_splitk_kernel(
    const float* __restrict__ x, int64_t P, int64_t F, int64_t ldx, const float* __restrict__ W,
    int64_t K, int64_t k_per_split, float* __restrict__ partial) {
  __shared__ float xs[TM * LDS_STRIDE];
  __shared__ float wsm[TN * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  const int64_t col0 = (int64_t)blockIdx.y * TN;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < F) ? kbeg + k_per_split : F;

  f32x4 acc[NBLK];
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[nb][e] = 0.f;

  const int lk = tid & 31;   // k within the stage handled by this thread
  const int lr = tid >> 5;   // first row handled (step 8)

  for (int64_t k0 = kbeg; k0 < kend; k0 += KC) {
    const int64_t k = k0 + lk;
    const bool kv = k < kend;
#pragma unroll
    for (int r = 0; r < TM / 8; ++r) {
      const int row = lr + 8 * r;
      const int64_t gr = row0 + row;
      xs[row * LDS_STRIDE + lk] = (kv && gr < P) ? x[gr * ldx + k] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < TN / 8; ++r) {
      const int row = lr + 8 * r;
      const int64_t gc = col0 + row;
      wsm[row * LDS_STRIDE + lk] = (kv && gc < K) ? W[gc * F + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const float av = xs[(wave * 16 + li) * LDS_STRIDE + kk * 4 + kq];
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) {
        const float bv = wsm[(nb * 16 + li) * LDS_STRIDE + kk * 4 + kq];
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // C/D layout: column = lane&15, row = (lane>>4)*4 + reg
  float* dst = partial + (int64_t)blockIdx.z * P * K;
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) {
    const int64_t gc = col0 + nb * 16 + li;
    if (gc >= K) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t gr = row0 + wave * 16 + kq * 4 + e;
      if (gr < P) dst[gr * K + gc] = acc[nb][e];
    }
  }
}

__global__ void linear_reduce_kernel(const float* __restrict__ partial, int64_t PK, int64_t K,
                                     int splits, const float* __restrict__ b, int apply_sigmoid,
                                     float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < PK;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * PK + i];
    if (b != nullptr) s += b[i % K];
    if (apply_sigmoid) s = 1.f / (1.f + expf(-s));
    out[i] = s;
  }
}

int choose_splits(int64_t P, int64_t F, int64_t K) {
  const int64_t tiles = tspn::ceil_div(P, TM) * tspn::ceil_div(K, TN);
  int64_t s = tspn::ceil_div(512, tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(1, F / (2 * KC)));
  s = std::min<int64_t>(s, MAX_SPLIT);
  return (int)std::max<int64_t>(s, 1);
}

}  // namespace

extern "C" size_t tspn_predicate_head_workspace_bytes(int64_t P, int64_t F, int64_t K) {
  if (P <= 0 || F <= 0 || K <= 0) return 0;
  return (size_t)choose_splits(P, F, K) * (size_t)P * (size_t)K * sizeof(float);
}

extern "C" int tspn_predicate_head_f32(const float* x, int64_t P, int64_t F, int64_t ldx,
                                       const float* W, const float* b, int64_t K, float* out,
                                       int apply_sigmoid, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  TSPN_REQUIRE(P >= 0 && F > 0 && K > 0 && ldx >= F, TSPN_EINVAL,
               "tspn_predicate_head_f32: bad sizes P=%lld F=%lld K=%lld ldx=%lld", (long long)P,
               (long long)F, (long long)K, (long long)ldx);
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(x && W && out, TSPN_EINVAL, "tspn_predicate_head_f32: null pointer");
  const int splits = choose_splits(P, F, K);
  const size_t need = (size_t)splits * (size_t)P * (size_t)K * sizeof(float);
  TSPN_REQUIRE(workspace != nullptr && workspace_bytes >= need, TSPN_EWORKSPACE,
               "tspn_predicate_head_f32: workspace %zu < %zu bytes", workspace_bytes, need);
  int64_t kps = tspn::ceil_div(F, splits);
  kps = tspn::ceil_div(kps, KC) * KC;
  const int64_t gx = tspn::ceil_div(P, TM), gy = tspn::ceil_div(K, TN);
  TSPN_REQUIRE(gx < (1LL << 31) && gy < 65536, TSPN_EUNSUPPORTED,
               "tspn_predicate_head_f32: grid too large");
  hipStream_t s = TSPN_STREAM(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(linear_splitk_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)splits),
                     dim3(THREADS), 0, s, x, P, F, ldx, W, K, kps, partial);
  int rc = tspn::check_launch("tspn_predicate_head_f32(splitk)");
  if (rc) return rc;
  const int64_t PK = P * K;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(PK, 256), 4096);
  hipLaunchKernelGGL(linear_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, PK, K, splits,
                     b, apply_sigmoid, out);
  return tspn::check_launch("tspn_predicate_head_f32(reduce)");
}
