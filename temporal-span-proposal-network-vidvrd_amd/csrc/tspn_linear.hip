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

// Span-restricted RelOIPool + predicate head.  The head is linear, so instead of pooling the pair
// features over each pair's span it is applied to every (tracklet, frame) row first -- G = f W'^T with
// W' = cls_w [K,2D] read as [2K, D] (column 2k: subject half, 2k+1: object half) -- prefix-summed over
// time in float64, and a pair's logit is a difference of two prefix rows per half divided by the span length.
__global__ void span_prefix_kernel(const float* __restrict__ G, int64_t NT, int T, int K2,
                                   double* __restrict__ PS) {
  const int64_t total = NT * K2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t trk = i / K2, c = i - trk * K2;
    const float* g = G + (trk * T) * (int64_t)K2 + c;
    double* ps = PS + (trk * (T + 1)) * (int64_t)K2 + c;
    double acc = 0.0;
    ps[0] = 0.0;
    for (int t = 0; t < T; ++t) {
      acc += (double)g[(int64_t)t * K2];
      ps[(int64_t)(t + 1) * K2] = acc;
    }
  }
}

__global__ void span_combine_kernel(const double* __restrict__ PS, const int64_t* __restrict__ pairs,
                                    const int64_t* __restrict__ spans, int64_t P, int T, int K,
                                    const float* __restrict__ b, float* __restrict__ out) {
  const int64_t total = P * K;
  const int64_t K2 = 2 * (int64_t)K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / K, k = i - p * K;
    int64_t a = spans[2 * p], e = spans[2 * p + 1];
    a = a < 0 ? 0 : (a > T - 1 ? T - 1 : a);       // out-of-range / unused (-1) spans fall back to
    e = e < a + 1 ? (spans[2 * p] < 0 ? T : a + 1) : (e > T ? T : e);  // the whole segment / one frame
    const double* ps = PS + (pairs[2 * p] * (T + 1)) * K2 + 2 * k;
    const double* po = PS + (pairs[2 * p + 1] * (T + 1)) * K2 + 2 * k + 1;
    double v = ((ps[e * K2] - ps[a * K2]) + (po[e * K2] - po[a * K2])) / (double)(e - a);
    if (b != nullptr) v += (double)b[k];
    out[i] = (float)(1.0 / (1.0 + exp(-v)));
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


namespace {
struct SpanLayout {
  size_t lin, g, ps, total;
};
SpanLayout span_layout(int64_t NT, int64_t T, int64_t D, int64_t K) {
  SpanLayout L{};
  L.lin = tspn::align_up(tspn_predicate_head_workspace_bytes(NT * T, D, 2 * K), 256);
  L.g = tspn::align_up((size_t)NT * T * 2 * K * sizeof(float), 256);
  L.ps = tspn::align_up((size_t)NT * (T + 1) * 2 * K * sizeof(double), 256);
  L.total = L.lin + L.g + L.ps;
  return L;
}
}  // namespace

extern "C" size_t tspn_span_predicate_workspace_bytes(int64_t NT, int64_t T, int64_t D, int64_t K) {
  if (NT <= 0 || T <= 0 || D <= 0 || K <= 0) return 0;
  return span_layout(NT, T, D, K).total;
}

extern "C" int tspn_span_predicate_f32(const float* feats, int64_t NT, int64_t T, int64_t D,
                                       const int64_t* pairs, const int64_t* spans, int64_t P,
                                       const float* cls_w, const float* cls_b, int64_t K, float* out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  TSPN_REQUIRE(NT >= 0 && T > 0 && D > 0 && K > 0 && P >= 0 && T < (1 << 30) && K < (1 << 20), TSPN_EINVAL,
               "tspn_span_predicate_f32: bad sizes NT=%lld T=%lld D=%lld K=%lld P=%lld", (long long)NT,
               (long long)T, (long long)D, (long long)K, (long long)P);
  if (P == 0 || NT == 0) return TSPN_OK;
  TSPN_REQUIRE(feats && pairs && spans && cls_w && out, TSPN_EINVAL, "tspn_span_predicate_f32: null pointer");
  const SpanLayout L = span_layout(NT, T, D, K);
  TSPN_REQUIRE(workspace && workspace_bytes >= L.total, TSPN_EWORKSPACE,
               "tspn_span_predicate_f32: workspace %zu < %zu bytes", workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  float* G = reinterpret_cast<float*>(ws + L.lin);
  double* PS = reinterpret_cast<double*>(ws + L.lin + L.g);
  // G[(trk, t), 2k + half] = f[trk, t, :] . cls_w[k, half*D : (half+1)*D]
  int rc = tspn::linear(feats, NT * T, D, D, cls_w, D, nullptr, 2 * K, G, 0, ws, L.lin, stream);
  if (rc) return rc;
  hipStream_t s = TSPN_STREAM(stream);
  const int b1 = (int)std::min<int64_t>(tspn::ceil_div(NT * 2 * K, 256), 8192);
  hipLaunchKernelGGL(span_prefix_kernel, dim3(b1), dim3(256), 0, s, G, NT, (int)T, (int)(2 * K), PS);
  if ((rc = tspn::check_launch("tspn_span_predicate_f32(prefix)"))) return rc;
  const int b2 = (int)std::min<int64_t>(tspn::ceil_div(P * K, 256), 8192);
  hipLaunchKernelGGL(span_combine_kernel, dim3(b2), dim3(256), 0, s, PS, pairs, spans, P, (int)T, (int)K,
                     cls_b, out);
  return tspn::check_launch("tspn_span_predicate_f32(combine)");
}
