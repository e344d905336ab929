// f1: top-k triplet decode on the GPU (gfx950).
//
// Replaces the per-segment Python of the reference's predict loop
// (lib/modeling/predict.py:66-106): per pair the `topk_pair` best predicates
// (torch.sort descending, first columns), over the segment the `topk_seg` best of those
// (torch.sort of the flattened [P, topk_pair] scores), then for each winner the pair's tracklet
// ids, the predicate id, and subject / object class = argmax over 35 class logits of the row
// `row_mul * tid` (predict.py:88-89 reads row (N-1)*tid of the pair-feature matrix: pass
// row_mul = N-1 and the feature matrix to reproduce that, or row_mul = 1 and the per-tracklet
// class logits).  In the reference this is ~6 ms of Python list comprehensions per segment and
// everything upstream of it must cross PCIe; here the [P,K] logits never leave HBM and only
// topk_seg x (score, triplet, pair) does — which also shrinks the multi-GPU result gather ~500x.
//
// Order: larger score first, lower index first on ties (= stable descending sort) at both levels.
// Integer / compare work only: bit-exact against the oracle.
//
//   kernel 1  one wave per pair: R rounds of wave arg-max over the K scores held in registers
//   kernel 2  one workgroup per segment: exact radix-select of the M-th largest candidate
//             (4 x 8-bit histograms over order-preserving keys), index-select among ties,
//             compaction into LDS, bitonic sort of the <= 1024 winners, label gathers.
#include <algorithm>
#include <cmath>

#include "tspn_common.h"

namespace {

constexpr int VPT = 4;        // values per lane in kernel 1 -> K <= 256
constexpr int SEG_THREADS = 1024;
constexpr int MAX_M = 1024;

__device__ __forceinline__ unsigned order_key(float v) {
  // monotone map float -> uint32 (larger float => larger key); -0 < +0 is harmless here
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) {
  return va > vb || (va == vb && ia < ib);
}

__global__ __launch_bounds__(256) void pair_topk_kernel(const float* __restrict__ logits,
                                                        int64_t rows, int K, int R,
                                                        float* __restrict__ sc,
                                                        int* __restrict__ ix) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = logits + row * K;
  float v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int k = lane + 64 * i;
    v[i] = k < K ? src[k] : -INFINITY;
  }
  unsigned used = 0;
  for (int r = 0; r < R; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int k = lane + 64 * i;
      if (k < K && !((used >> i) & 1u) && better(v[i], k, bv, bi)) {
        bv = v[i];
        bi = k;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      if (better(ov, oi, bv, bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if ((bi & 63) == lane && bi < K) used |= 1u << (bi >> 6);
    if (lane == 0) {
      sc[row * R + r] = bv;
      ix[row * R + r] = bi;
    }
  }
}

__device__ int argmax_first(const float* p, int n) {
  float bv = p[0];
  int bi = 0;
  for (int i = 1; i < n; ++i)
    if (p[i] > bv) {
      bv = p[i];
      bi = i;
    }
  return bi;
}

__global__ __launch_bounds__(SEG_THREADS) void segment_topk_kernel(
    const float* __restrict__ sc, const int* __restrict__ ix, const int64_t* __restrict__ pairs,
    const float* __restrict__ cls_sub, const float* __restrict__ cls_obj, int64_t ld, int64_t seg_rows,
    int64_t row_mul, int P, int R, int NO, int M, float* __restrict__ out_score,
    int64_t* __restrict__ out_trip, int64_t* __restrict__ out_tid) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need, s_count;
  __shared__ float kv[MAX_M];
  __shared__ int ki[MAX_M];

  const int tid = threadIdx.x;
  const int64_t seg = blockIdx.x;
  const int Q = P * R;
  const float* cand = sc + seg * Q;

  // ---- exact M-th largest key by 4 radix passes (most significant byte first)
  if (tid == 0) {
    s_prefix = 0;
    s_need = (unsigned)M;
  }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < Q; i += SEG_THREADS) {
      const unsigned k = order_key(cand[i]);
      if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, acc = 0;
      int b = 255;
      for (; b >= 0; --b) {
        if (acc + hist[b] >= need) break;
        acc += hist[b];
      }
      s_prefix = prefix | ((unsigned)b << shift);
      s_need = need - acc;  // how many we still need inside bucket b
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;     // key of the M-th largest candidate
  const unsigned ties_needed = s_need;  // of the candidates equal to it, the lowest indices win
  __syncthreads();  // everyone has read s_prefix / s_need before they are reused below
  // ---- among ties: the `ties_needed`-th smallest flat index (4 byte-wide passes, lowest first)
  if (tid == 0) {
    s_prefix = 0;
    s_need = ties_needed;
  }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < Q; i += SEG_THREADS) {
      if (order_key(cand[i]) == kth && (((unsigned)i) & himask) == prefix)
        atomicAdd(&hist[(((unsigned)i) >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, acc = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (acc + hist[b] >= need) break;
        acc += hist[b];
      }
      s_prefix = prefix | ((unsigned)b << shift);
      s_need = need - acc;
    }
    __syncthreads();
  }
  const unsigned last_tie_idx = s_prefix;  // ties with index <= this are selected

  // ---- compaction of the M winners into LDS (unordered), then bitonic sort
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (int i = tid; i < Q; i += SEG_THREADS) {
    const float v = cand[i];
    const unsigned k = order_key(v);
    if (k > kth || (k == kth && (unsigned)i <= last_tie_idx)) {
      const unsigned slot = atomicAdd(&s_count, 1u);
      if (slot < (unsigned)MAX_M) {
        kv[slot] = v;
        ki[slot] = i;
      }
    }
  }
  __syncthreads();
  int m2 = 1;
  while (m2 < M) m2 <<= 1;
  for (int i = tid; i < m2; i += SEG_THREADS)
    if (i >= M) {
      kv[i] = -INFINITY;
      ki[i] = 0x7fffffff;
    }
  __syncthreads();
  for (int k = 2; k <= m2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < m2; i += SEG_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const float vi = kv[i], vl = kv[l];
          const int ii = ki[i], il = ki[l];
          const bool fwd = (i & k) == 0;
          const bool swap = fwd ? better(vl, il, vi, ii) : better(vi, ii, vl, il);
          if (swap) {
            kv[i] = vl;
            kv[l] = vi;
            ki[i] = il;
            ki[l] = ii;
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- gathers: pair ids, predicate id, class labels
  for (int r = tid; r < M; r += SEG_THREADS) {
    const int flat = ki[r];
    const int pi = flat / R;
    const int64_t ts = pairs[(seg * P + pi) * 2], to = pairs[(seg * P + pi) * 2 + 1];
    const int64_t o = seg * M + r;
    out_score[o] = kv[r];
    out_tid[2 * o] = ts;
    out_tid[2 * o + 1] = to;
    out_trip[3 * o] = argmax_first(cls_sub + (seg * seg_rows + row_mul * ts) * ld, NO);
    out_trip[3 * o + 1] = ix[seg * Q + flat];
    out_trip[3 * o + 2] = argmax_first(cls_obj + (seg * seg_rows + row_mul * to) * ld, NO);
  }
}

}  // namespace

extern "C" size_t tspn_decode_topk_workspace_bytes(int64_t S, int64_t P, int64_t topk_pair) {
  if (S <= 0 || P <= 0 || topk_pair <= 0) return 0;
  return (size_t)S * (size_t)P * (size_t)topk_pair * (sizeof(float) + sizeof(int));
}

extern "C" int tspn_decode_topk_f32(const float* rel_logit, const int64_t* pairs,
                                    const float* cls_sub, const float* cls_obj, int64_t ld,
                                    int64_t seg_rows, int64_t row_mul, int64_t S, int64_t P,
                                    int64_t K, int64_t NO, int64_t topk_pair, int64_t topk_seg,
                                    float* out_score, int64_t* out_triplet, int64_t* out_pair_tid,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  TSPN_REQUIRE(S >= 0 && P >= 0 && K > 0 && NO > 0 && topk_pair > 0 && topk_seg > 0 && ld >= NO &&
                   seg_rows > 0 && row_mul >= 0,
               TSPN_EINVAL, "tspn_decode_topk_f32: bad sizes");
  TSPN_REQUIRE(K <= 64 * VPT, TSPN_EUNSUPPORTED, "tspn_decode_topk_f32: K=%lld > %d", (long long)K,
               64 * VPT);
  const int64_t R = std::min<int64_t>(topk_pair, K);
  const int64_t M = std::min<int64_t>(topk_seg, P * R);
  TSPN_REQUIRE(M <= MAX_M, TSPN_EUNSUPPORTED, "tspn_decode_topk_f32: topk_seg=%lld > %d",
               (long long)M, MAX_M);
  TSPN_REQUIRE(P * R < (1LL << 30), TSPN_EUNSUPPORTED, "tspn_decode_topk_f32: P*topk_pair too large");
  if (S == 0 || P == 0) return TSPN_OK;
  TSPN_REQUIRE(rel_logit && pairs && cls_sub && cls_obj && out_score && out_triplet && out_pair_tid,
               TSPN_EINVAL, "tspn_decode_topk_f32: null pointer");
  const size_t need = tspn_decode_topk_workspace_bytes(S, P, R);
  TSPN_REQUIRE(workspace && workspace_bytes >= need, TSPN_EWORKSPACE,
               "tspn_decode_topk_f32: workspace %zu < %zu bytes", workspace_bytes, need);
  float* sc = static_cast<float*>(workspace);
  int* ix = reinterpret_cast<int*>(sc + S * P * R);
  hipStream_t s = TSPN_STREAM(stream);
  const int64_t rows = S * P;
  const int64_t nb = tspn::ceil_div(rows, 4);
  TSPN_REQUIRE(nb < (1LL << 31) && S < (1LL << 31), TSPN_EUNSUPPORTED,
               "tspn_decode_topk_f32: grid too large");
  hipLaunchKernelGGL(pair_topk_kernel, dim3((unsigned)nb), dim3(256), 0, s, rel_logit, rows, (int)K,
                     (int)R, sc, ix);
  int rc = tspn::check_launch("tspn_decode_topk_f32(pair)");
  if (rc) return rc;
  hipLaunchKernelGGL(segment_topk_kernel, dim3((unsigned)S), dim3(SEG_THREADS), 0, s, sc, ix, pairs,
                     cls_sub, cls_obj, ld, seg_rows, row_mul, (int)P, (int)R, (int)NO, (int)M,
                     out_score, out_triplet, out_pair_tid);
  return tspn::check_launch("tspn_decode_topk_f32(segment)");
}
