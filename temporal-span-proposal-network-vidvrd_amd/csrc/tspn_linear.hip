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

// f2 (fused a15): K-ranges of the split-K slices when the block-L1 preprocessing of
// VRDataset._feature_preprocess (reference lib/dataset/vrdataset.py:219-243) is folded into the GEMM:
// slices never straddle a normalisation block, each slice also returns sum|x| per row, and the
// reduce kernel divides the slices of a block by that block's L1 norm — x is read exactly once.
constexpr int MAX_NORM_SPLITS = 128;
struct SplitTable {
  int n;
  int kbeg[MAX_NORM_SPLITS];
  int kend[MAX_NORM_SPLITS];
  short blk[MAX_NORM_SPLITS];  // normalisation block id, -1 = not normalised
};

template <bool NORM>
__global__ __launch_bounds__(THREADS) void linear_splitk_kernel(
    const float* __restrict__ x, int64_t P, int64_t F, int64_t ldx, const float* __restrict__ W,
    int64_t ldw, int64_t K, int64_t k_per_split, float* __restrict__ partial, SplitTable tab,
    float* __restrict__ norm_partial) {
  __shared__ float xs[TM * LDS_STRIDE];
  __shared__ float wsm[TN * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  const int64_t col0 = (int64_t)blockIdx.y * TN;
  int64_t kbeg, kend;
  if (NORM) {
    kbeg = tab.kbeg[blockIdx.z];
    kend = tab.kend[blockIdx.z];
  } else {
    kbeg = (int64_t)blockIdx.z * k_per_split;
    kend = (kbeg + k_per_split < F) ? kbeg + k_per_split : F;
  }
  float asum[TM / 8];
#pragma unroll
  for (int r = 0; r < TM / 8; ++r) asum[r] = 0.f;

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
      const float xv = (kv && gr < P) ? x[gr * ldx + k] : 0.f;
      xs[row * LDS_STRIDE + lk] = xv;
      if (NORM) asum[r] += fabsf(xv);
    }
#pragma unroll
    for (int r = 0; r < TN / 8; ++r) {
      const int row = lr + 8 * r;
      const int64_t gc = col0 + row;
      wsm[row * LDS_STRIDE + lk] = (kv && gc < K) ? W[gc * ldw + k] : 0.f;
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

  if (NORM && blockIdx.y == 0) {  // sum|x| of this slice per row: 32 lanes share a row
#pragma unroll
    for (int r = 0; r < TM / 8; ++r) {
      float a = asum[r];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off);
      const int64_t gr = row0 + lr + 8 * r;
      if (lk == 0 && gr < P) norm_partial[(int64_t)blockIdx.z * P + gr] = a;
    }
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

__global__ void linear_reduce_norm_kernel(const float* __restrict__ partial,
                                          const float* __restrict__ norm_partial, int64_t P,
                                          int64_t K, SplitTable tab, const float* __restrict__ b,
                                          int apply_sigmoid, float* __restrict__ out) {
  const int64_t PK = P * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < PK;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / K;
    float s = 0.f;
    int z = 0;
    while (z < tab.n) {  // slices of one block are contiguous in the table
      const int blk = tab.blk[z];
      float acc = 0.f, l1 = 0.f;
      int z2 = z;
      for (; z2 < tab.n && tab.blk[z2] == blk; ++z2) {
        acc += partial[(int64_t)z2 * PK + i];
        if (blk >= 0) l1 += norm_partial[(int64_t)z2 * P + row];
      }
      if (blk >= 0) acc = acc / (l1 == 0.f ? 1.f : l1);
      s += acc;
      z = z2;
    }
    if (b != nullptr) s += b[i % K];
    if (apply_sigmoid) s = 1.f / (1.f + expf(-s));
    out[i] = s;
  }
}

// host: slice table for the fused-normalisation form
int build_split_table(int64_t P, int64_t F, int64_t K, int64_t first, int64_t block, int64_t nblocks,
                      SplitTable* tab) {
  const int64_t tiles = tspn::ceil_div(P, TM) * tspn::ceil_div(K, TN);
  int64_t target = std::min<int64_t>(std::max<int64_t>(tspn::ceil_div(512, tiles), 1), 64);
  const int64_t len = std::max<int64_t>(tspn::ceil_div(tspn::ceil_div(F, target), KC) * KC, 2 * KC);
  int n = 0;
  auto add_region = [&](int64_t lo, int64_t hi, int blk) {
    if (hi <= lo) return true;
    const int64_t pieces = tspn::ceil_div(hi - lo, len);
    const int64_t step = tspn::ceil_div(tspn::ceil_div(hi - lo, pieces), KC) * KC;
    for (int64_t k = lo; k < hi; k += step) {
      if (n >= MAX_NORM_SPLITS) return false;
      tab->kbeg[n] = (int)k;
      tab->kend[n] = (int)std::min<int64_t>(k + step, hi);
      tab->blk[n] = (short)blk;
      ++n;
    }
    return true;
  };
  bool ok = add_region(0, first, -1);
  for (int64_t bl = 0; ok && bl < nblocks; ++bl)
    ok = add_region(first + bl * block, first + (bl + 1) * block, (int)bl);
  // the unnormalised tail must not be merged with the head region (-1) by the reduce loop: it is not
  // adjacent to it in the table, and runs of equal ids are only merged when contiguous, which is fine
  ok = ok && add_region(first + nblocks * block, F, -1);
  tab->n = n;
  return ok ? n : -1;
}

// logits[p,k] = sigmoid(rs[s(p),k] + ro[o(p),k] + b[k]): the predicate head on cat(f_s, f_o) is linear
// in the two halves, so it is evaluated per tracklet (rs = f W_s^T, ro = f W_o^T) and combined per pair.
__global__ void pair_combine_kernel(const float* __restrict__ rs, const float* __restrict__ ro,
                                    const int64_t* __restrict__ pairs, int64_t P, int64_t K,
                                    const float* __restrict__ b, float* __restrict__ out) {
  const int64_t total = P * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / K, k = i - p * K;
    float v = rs[pairs[2 * p] * K + k] + ro[pairs[2 * p + 1] * K + k];
    if (b != nullptr) v += b[k];
    out[i] = 1.f / (1.f + expf(-v));
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
  return tspn::linear(x, P, F, ldx, W, F, b, K, out, apply_sigmoid, workspace, workspace_bytes, stream);
}

// out[P,K] = act(x[P,F] @ W[K, :F]^T + b) with row strides ldx / ldw (a column slice of a wider W)
int tspn::linear(const float* x, int64_t P, int64_t F, int64_t ldx, const float* W, int64_t ldw,
                 const float* b, int64_t K, float* out, int apply_sigmoid, void* workspace,
                 size_t workspace_bytes, void* stream) {
  TSPN_REQUIRE(ldw >= F, TSPN_EINVAL, "tspn_predicate_head_f32: bad ldw");
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
  hipLaunchKernelGGL(linear_splitk_kernel<false>, dim3((unsigned)gx, (unsigned)gy, (unsigned)splits),
                     dim3(THREADS), 0, s, x, P, F, ldx, W, ldw, K, kps, partial, SplitTable{},
                     (float*)nullptr);
  int rc = tspn::check_launch("tspn_predicate_head_f32(splitk)");
  if (rc) return rc;
  const int64_t PK = P * K;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(PK, 256), 4096);
  hipLaunchKernelGGL(linear_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, PK, K, splits,
                     b, apply_sigmoid, out);
  return tspn::check_launch("tspn_predicate_head_f32(reduce)");
}


extern "C" size_t tspn_predicate_head_norm_workspace_bytes(int64_t P, int64_t F, int64_t K,
                                                           int64_t first, int64_t block,
                                                           int64_t nblocks) {
  if (P <= 0 || F <= 0 || K <= 0 || first < 0 || block <= 0 || nblocks < 0 ||
      first + block * nblocks > F)
    return 0;
  SplitTable tab;
  const int n = build_split_table(P, F, K, first, block, nblocks, &tab);
  if (n < 0) return 0;
  return (size_t)n * (size_t)P * (size_t)(K + 1) * sizeof(float);
}

extern "C" int tspn_predicate_head_norm_f32(const float* x, int64_t P, int64_t F, int64_t ldx,
                                            const float* W, const float* b, int64_t K,
                                            int64_t first, int64_t block, int64_t nblocks,
                                            float* out, int apply_sigmoid, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  TSPN_REQUIRE(P >= 0 && F > 0 && K > 0 && ldx >= F, TSPN_EINVAL,
               "tspn_predicate_head_norm_f32: bad sizes P=%lld F=%lld K=%lld ldx=%lld", (long long)P,
               (long long)F, (long long)K, (long long)ldx);
  TSPN_REQUIRE(first >= 0 && block > 0 && nblocks >= 0 && first + block * nblocks <= F, TSPN_EINVAL,
               "tspn_predicate_head_norm_f32: blocks [%lld, %lld) exceed F=%lld", (long long)first,
               (long long)(first + block * nblocks), (long long)F);
  TSPN_REQUIRE(F < (1LL << 31) && nblocks < 32768, TSPN_EUNSUPPORTED,
               "tspn_predicate_head_norm_f32: F or nblocks too large");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(x && W && out, TSPN_EINVAL, "tspn_predicate_head_norm_f32: null pointer");
  SplitTable tab;
  const int n = build_split_table(P, F, K, first, block, nblocks, &tab);
  TSPN_REQUIRE(n > 0, TSPN_EUNSUPPORTED, "tspn_predicate_head_norm_f32: more than %d K-slices needed",
               MAX_NORM_SPLITS);
  const size_t need = (size_t)n * (size_t)P * (size_t)(K + 1) * sizeof(float);
  TSPN_REQUIRE(workspace != nullptr && workspace_bytes >= need, TSPN_EWORKSPACE,
               "tspn_predicate_head_norm_f32: workspace %zu < %zu bytes", workspace_bytes, need);
  const int64_t gx = tspn::ceil_div(P, TM), gy = tspn::ceil_div(K, TN);
  TSPN_REQUIRE(gx < (1LL << 31) && gy < 65536, TSPN_EUNSUPPORTED,
               "tspn_predicate_head_norm_f32: grid too large");
  hipStream_t s = TSPN_STREAM(stream);
  float* partial = static_cast<float*>(workspace);
  float* norm_partial = partial + (size_t)n * P * K;
  hipLaunchKernelGGL(linear_splitk_kernel<true>, dim3((unsigned)gx, (unsigned)gy, (unsigned)n),
                     dim3(THREADS), 0, s, x, P, F, ldx, W, F, K, (int64_t)0, partial, tab, norm_partial);
  int rc = tspn::check_launch("tspn_predicate_head_norm_f32(splitk)");
  if (rc) return rc;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(P * K, 256), 4096);
  hipLaunchKernelGGL(linear_reduce_norm_kernel, dim3(blocks), dim3(256), 0, s, partial, norm_partial,
                     P, K, tab, b, apply_sigmoid, out);
  return tspn::check_launch("tspn_predicate_head_norm_f32(reduce)");
}


size_t tspn::pair_predicate_workspace_bytes(int64_t NT, int64_t D, int64_t K) {
  return tspn::align_up(tspn_predicate_head_workspace_bytes(NT, D, K), 256) +
         2 * tspn::align_up((size_t)NT * K * sizeof(float), 256);
}

// rel_logits[P,K] = sigmoid(cat(fbar[s], fbar[o]) @ cls_w[K,2D]^T + b), factorised per tracklet
int tspn::pair_predicate(const float* fbar, int64_t NT, int64_t D, const int64_t* pairs, int64_t P,
                         const float* cls_w, const float* cls_b, int64_t K, float* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (P == 0 || NT == 0) return TSPN_OK;
  const size_t lin = tspn::align_up(tspn_predicate_head_workspace_bytes(NT, D, K), 256);
  const size_t rbytes = tspn::align_up((size_t)NT * K * sizeof(float), 256);
  TSPN_REQUIRE(workspace && workspace_bytes >= lin + 2 * rbytes, TSPN_EWORKSPACE,
               "pair_predicate: workspace too small");
  char* ws = static_cast<char*>(workspace);
  float* rs = reinterpret_cast<float*>(ws + lin);
  float* ro = reinterpret_cast<float*>(ws + lin + rbytes);
  int rc = tspn::linear(fbar, NT, D, D, cls_w, 2 * D, nullptr, K, rs, 0, ws, lin, stream);
  if (rc) return rc;
  rc = tspn::linear(fbar, NT, D, D, cls_w + D, 2 * D, nullptr, K, ro, 0, ws, lin, stream);
  if (rc) return rc;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(P * K, 256), 4096);
  hipLaunchKernelGGL(pair_combine_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), rs, ro, pairs,
                     P, K, cls_b, out);
  return tspn::check_launch("pair_predicate(combine)");
}
