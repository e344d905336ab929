// a15, first half: the proposal pair filter of the in-repo pair-feature builder (gfx950).
//
// The reference keeps, of the (N+G)(N+G-1) ordered pairs stored in a segment's -relation.h5 file,
// those whose two tracks are both PROPOSALS (trackid < 0; ground-truth tracks carry their dataset id):
//     proposal_idx = [ind for ind, (t1, t2) in enumerate(pairs) if trackid[t1] < 0 and trackid[t2] < 0]
//     num_tracks   = sum(trackid < 0)                    (lib/dataset/vrdataset.py:140-148)
// and then indexes feats / pairs / labels with that list (vrdataset.py:66-67).  Here: one workgroup per
// segment does an order-preserving stream compaction (wave ballots + a running base), and a row
// gather moves the kept feature rows.  Integer / byte work, HBM-bound: plain coalesced kernels.
#include <algorithm>

#include "tspn_common.h"

namespace {

constexpr int FILTER_THREADS = 256;

__global__ __launch_bounds__(FILTER_THREADS) void proposal_pair_filter_kernel(
    const int64_t* __restrict__ pairs, const int64_t* __restrict__ pair_off,
    const int64_t* __restrict__ trackid, const int64_t* __restrict__ track_off,
    int64_t* __restrict__ out_idx, int64_t* __restrict__ out_count,
    int64_t* __restrict__ out_num_tracks) {
  __shared__ int wave_tot[FILTER_THREADS / 64];
  __shared__ int bad_flag;
  const int s = blockIdx.x;
  const int64_t p0 = pair_off[s], p1 = pair_off[s + 1];
  const int64_t m0 = track_off[s], M = track_off[s + 1] - m0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) bad_flag = 0;
  __syncthreads();

  // num_tracks = sum(trackid < 0)
  int local = 0;
  for (int64_t i = threadIdx.x; i < M; i += FILTER_THREADS) local += trackid[m0 + i] < 0 ? 1 : 0;
  for (int off = 32; off; off >>= 1) local += __shfl_down(local, off);
  if (lane == 0) wave_tot[wave] = local;
  __syncthreads();
  int ntracks = 0;
  for (int w = 0; w < FILTER_THREADS / 64; ++w) ntracks += wave_tot[w];
  __syncthreads();

  int64_t base = 0;  // kept so far (same value in every thread)
  for (int64_t c0 = p0; c0 < p1; c0 += FILTER_THREADS) {
    const int64_t p = c0 + threadIdx.x;
    bool keep = false;
    if (p < p1) {
      const int64_t a = pairs[2 * p], b = pairs[2 * p + 1];
      if (a < 0 || a >= M || b < 0 || b >= M)
        bad_flag = 1;  // benign race: every writer stores 1
      else
        keep = trackid[m0 + a] < 0 && trackid[m0 + b] < 0;
    }
    const unsigned long long mask = __ballot(keep);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(mask);
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < FILTER_THREADS / 64; ++w) {
      const int tw = wave_tot[w];
      wbase += w < wave ? tw : 0;
      total += tw;
    }
    if (keep) out_idx[p0 + base + wbase + before] = p - p0;
    base += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_count[s] = bad_flag ? -1 : base;
    out_num_tracks[s] = ntracks;
  }
}

// out[r, :] = src[idx[r], :]; a workgroup row per blockIdx.y slab, lanes along the row.
template <typename V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const V* __restrict__ src, int64_t ld,
                                                          int64_t F, const int64_t* __restrict__ idx,
                                                          int64_t idx_base, int64_t R,
                                                          V* __restrict__ out) {
  const int64_t r = blockIdx.y;
  if (r >= R) return;
  const V* s = src + (idx_base + idx[r]) * ld;
  V* o = out + r * F;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < F;
       c += (int64_t)gridDim.x * blockDim.x)
    o[c] = s[c];
}

}  // namespace

extern "C" int tspn_proposal_pair_filter_i64(const int64_t* pairs, const int64_t* pair_off,
                                             const int64_t* trackid, const int64_t* track_off,
                                             int64_t S, int64_t* out_idx, int64_t* out_count,
                                             int64_t* out_num_tracks, void* stream) {
  TSPN_REQUIRE(S >= 0 && S < (1LL << 31), TSPN_EINVAL, "tspn_proposal_pair_filter_i64: S=%lld",
               (long long)S);
  if (S == 0) return TSPN_OK;
  TSPN_REQUIRE(pairs && pair_off && trackid && track_off && out_idx && out_count && out_num_tracks,
               TSPN_EINVAL, "tspn_proposal_pair_filter_i64: null pointer");
  hipLaunchKernelGGL(proposal_pair_filter_kernel, dim3((unsigned)S), dim3(FILTER_THREADS), 0,
                     TSPN_STREAM(stream), pairs, pair_off, trackid, track_off, out_idx, out_count,
                     out_num_tracks);
  return tspn::check_launch("tspn_proposal_pair_filter_i64");
}

extern "C" int tspn_gather_rows_f32(const float* src, int64_t ld, int64_t F, const int64_t* idx,
                                    int64_t idx_base, int64_t R, float* out, void* stream) {
  TSPN_REQUIRE(ld >= F && F > 0 && R >= 0, TSPN_EINVAL, "tspn_gather_rows_f32: bad sizes");
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(src && idx && out, TSPN_EINVAL, "tspn_gather_rows_f32: null pointer");
  hipStream_t s = TSPN_STREAM(stream);
  const bool v2 = F % 2 == 0 && ld % 2 == 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 7) == 0;
  const int64_t cols = v2 ? F / 2 : F;
  const unsigned gx = (unsigned)std::min<int64_t>(tspn::ceil_div(cols, 256), 64);
  for (int64_t r0 = 0; r0 < R; r0 += 65535) {  // blockIdx.y limit
    const int64_t nr = std::min<int64_t>(65535, R - r0);
    if (v2)
      hipLaunchKernelGGL(gather_rows_kernel<float2>, dim3(gx, (unsigned)nr), dim3(256), 0, s,
                         reinterpret_cast<const float2*>(src), ld / 2, cols, idx + r0, idx_base, nr,
                         reinterpret_cast<float2*>(out + r0 * F));
    else
      hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(gx, (unsigned)nr), dim3(256), 0, s, src, ld,
                         cols, idx + r0, idx_base, nr, out + r0 * F);
    int rc = tspn::check_launch("tspn_gather_rows_f32");
    if (rc) return rc;
  }
  return TSPN_OK;
}
