// a5/a6: PPN — tracklet-level relationness matrix and top-k pair indices,
// one launch per batch of segments (gfx950).
//
// Replaces PPNHead.forward (reference lib/modeling/relpn/ppn.py:107-112:
// two Linear-ReLU-Linear embeddings, sub @ obj^T, sigmoid) and the
// `torch.sort(pair_matrix.view(-1), descending=True)[:num_pair_proposals]` of
// PPN._forward_test (ppn.py:84-85).  In the reference this is ~10 launch-bound
// torch ops per segment; here the whole thing runs in one workgroup per segment
// with every intermediate in LDS.  The sort is a bitonic network over
// (value, flat index) with the total order "larger value first, lower index
// first on ties" (= a stable descending sort; the reference's unstable sort
// leaves tie order unspecified — SURVEY.md §7 hard part 3).
// Indices are into the N x N matrix including the diagonal (s*N + o).
#include <algorithm>
#include <cmath>

#include "tspn_common.h"

namespace {

constexpr int PPN_THREADS = 1024;

__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) {
  return ka > kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(PPN_THREADS) void ppn_kernel(
    const float* __restrict__ cls, int N, int Cin, int H, int Cout,
    const float* __restrict__ ws1, const float* __restrict__ bs1, const float* __restrict__ ws2,
    const float* __restrict__ bs2, const float* __restrict__ wo1, const float* __restrict__ bo1,
    const float* __restrict__ wo2, const float* __restrict__ bo2, int topk, int n2p,
    float* __restrict__ out_mat, int64_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_cls = reinterpret_cast<float*>(smem_raw);  // [N][Cin]
  float* s_hid = s_cls + N * Cin;                      // [N][H]
  float* s_es = s_hid + N * H;                         // [N][Cout]
  float* s_eo = s_es + N * Cout;                       // [N][Cout]
  float* s_key = s_eo + N * Cout;                      // [n2p]
  int* s_idx = reinterpret_cast<int*>(s_key + n2p);    // [n2p]

  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const float* c = cls + b * N * Cin;
  for (int i = tid; i < N * Cin; i += PPN_THREADS) s_cls[i] = c[i];
  __syncthreads();

  for (int role = 0; role < 2; ++role) {
    const float* w1 = role ? wo1 : ws1;
    const float* b1 = role ? bo1 : bs1;
    const float* w2 = role ? wo2 : ws2;
    const float* b2 = role ? bo2 : bs2;
    float* emb = role ? s_eo : s_es;
    for (int i = tid; i < N * H; i += PPN_THREADS) {
      const int n = i / H, h = i - n * H;
      float acc = 0.f;
      for (int k = 0; k < Cin; ++k) acc += s_cls[n * Cin + k] * w1[h * Cin + k];
      acc += b1[h];
      s_hid[i] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int i = tid; i < N * Cout; i += PPN_THREADS) {
      const int n = i / Cout, o = i - n * Cout;
      float acc = 0.f;
      for (int k = 0; k < H; ++k) acc += s_hid[n * H + k] * w2[o * H + k];
      emb[i] = acc + b2[o];
    }
    __syncthreads();
  }

  const int n2 = N * N;
  for (int i = tid; i < n2p; i += PPN_THREADS) {
    if (i < n2) {
      const int s = i / N, o = i - s * N;
      float acc = 0.f;
      for (int k = 0; k < Cout; ++k) acc += s_es[s * Cout + k] * s_eo[o * Cout + k];
      const float v = 1.f / (1.f + expf(-acc));
      out_mat[b * n2 + i] = v;
      s_key[i] = v;
      s_idx[i] = i;
    } else {
      s_key[i] = -INFINITY;
      s_idx[i] = 0x7fffffff;
    }
  }
  __syncthreads();

  for (int k = 2; k <= n2p; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2p; i += PPN_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const float ki = s_key[i], kl = s_key[l];
          const int ii = s_idx[i], il = s_idx[l];
          const bool fwd = (i & k) == 0;
          const bool swap = fwd ? before(kl, il, ki, ii) : before(ki, ii, kl, il);
          if (swap) {
            s_key[i] = kl;
            s_key[l] = ki;
            s_idx[i] = il;
            s_idx[l] = ii;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < topk; i += PPN_THREADS) out_idx[b * topk + i] = (int64_t)s_idx[i];
}

}  // namespace

extern "C" int tspn_ppn_pair_matrix_topk_f32(const float* cls, int64_t B, int64_t N, int64_t Cin,
                                             int64_t H, int64_t Cout, const float* ws1,
                                             const float* bs1, const float* ws2, const float* bs2,
                                             const float* wo1, const float* bo1, const float* wo2,
                                             const float* bo2, int64_t topk, float* out_mat,
                                             int64_t* out_idx, void* stream) {
  TSPN_REQUIRE(B >= 0 && N > 0 && Cin > 0 && H > 0 && Cout > 0, TSPN_EINVAL,
               "tspn_ppn_pair_matrix_topk_f32: bad sizes");
  TSPN_REQUIRE(topk >= 0 && topk <= N * N, TSPN_EINVAL,
               "tspn_ppn_pair_matrix_topk_f32: topk=%lld not in [0, N*N=%lld]", (long long)topk,
               (long long)(N * N));
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(cls && ws1 && bs1 && ws2 && bs2 && wo1 && bo1 && wo2 && bo2 && out_mat,
               TSPN_EINVAL, "tspn_ppn_pair_matrix_topk_f32: null pointer");
  TSPN_REQUIRE(topk == 0 || out_idx, TSPN_EINVAL, "tspn_ppn_pair_matrix_topk_f32: null out_idx");
  TSPN_REQUIRE(N <= 128 && Cin <= 256 && H <= 256 && Cout <= 256, TSPN_EUNSUPPORTED,
               "tspn_ppn_pair_matrix_topk_f32: N=%lld Cin=%lld H=%lld Cout=%lld over limits "
               "(N<=128, channels<=256)",
               (long long)N, (long long)Cin, (long long)H, (long long)Cout);
  int n2p = 1;
  while (n2p < N * N) n2p <<= 1;
  const size_t smem = sizeof(float) * (size_t)(N * Cin + N * H + 2 * N * Cout) +
                      (sizeof(float) + sizeof(int)) * (size_t)n2p;
  TSPN_REQUIRE(smem <= 160 * 1024, TSPN_EUNSUPPORTED,
               "tspn_ppn_pair_matrix_topk_f32: needs %zu B of LDS (> 160 KiB)", smem);
  static tspn::LdsLimit lds;
  if (smem > 48 * 1024)
    if (int rc = lds.ensure(reinterpret_cast<const void*>(ppn_kernel), smem, "tspn_ppn_pair_matrix_topk_f32"))
      return rc;
  TSPN_REQUIRE(B < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_ppn_pair_matrix_topk_f32: B too large");
  hipLaunchKernelGGL(ppn_kernel, dim3((unsigned)B), dim3(PPN_THREADS), smem, TSPN_STREAM(stream),
                     cls, (int)N, (int)Cin, (int)H, (int)Cout, ws1, bs1, ws2, bs2, wo1, bo1, wo2,
                     bo2, (int)topk, n2p, out_mat, out_idx);
  return tspn::check_launch("tspn_ppn_pair_matrix_topk_f32");
}
