// a16: trajectory ("cubic") IoU over N1 x N2 x T boxes (gfx950).
//
// Replaces cubic_iou / _intersect / _union (reference
// lib/modeling/trajectory.py:85-141).  The operation order of the reference is
// kept so that integer-valued boxes give bit-identical results: per frame
// w = (min(r1,r2) + 1) - max(l1,l2) clipped at 0, likewise h, inters += w*h in
// fp32 in frame order (trajectory.py:95-106); areas summed over frames in order
// (trajectory.py:110-123); iou = inters / (area1 + area2 - inters)
// (trajectory.py:137-140).  The library is built with -ffp-contract=off so no
// multiply-add is fused behind the reference's back.
//
// HBM-bound and tiny (N^2*T*32 B): one lane per (i, j), 16-B box loads.
#include <algorithm>

#include "tspn_common.h"

namespace {

__global__ __launch_bounds__(256) void traj_iou_kernel(const float4* __restrict__ b1, int64_t N1,
                                                       const float4* __restrict__ b2, int64_t N2,
                                                       int64_t T, float* __restrict__ out) {
  const int64_t bidx = blockIdx.y;
  const int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= N1 * N2) return;
  const int64_t i = ij / N2, j = ij - i * N2;
  const float4* r1 = b1 + (bidx * N1 + i) * T;
  const float4* r2 = b2 + (bidx * N2 + j) * T;
  float inters = 0.f, a1 = 0.f, a2 = 0.f;
  for (int64_t t = 0; t < T; ++t) {
    const float4 p = r1[t], q = r2[t];
    float w = (fminf(p.z, q.z) + 1.f) - fmaxf(p.x, q.x);
    w = fmaxf(w, 0.f);
    float h = (fminf(p.w, q.w) + 1.f) - fmaxf(p.y, q.y);
    h = fmaxf(h, 0.f);
    inters += w * h;
    a1 += (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
    a2 += (q.z - q.x + 1.f) * (q.w - q.y + 1.f);
  }
  const float uni = (a1 + a2) - inters;
  out[bidx * N1 * N2 + ij] = inters / uni;
}

// ---- association-time IoU (f3): every live trajectory of the previous segment's relations against every
// tracklet of the current segment, on their common frames, in ONE launch.
//
// Replaces the per-candidate calls `_traj_iou(r.straj, straj)` / `_traj_iou(r.otraj, otraj)` of
// reference lib/modeling/association.py:35-48,101-106 (-> trajectory.py:144-158 traj_iou -> cubic_iou with
// float64 boxes).  Rounding recipe of that call chain, kept to the bit:
//   * max / min of the float64 coordinates are STORED as float32 (`out=` arrays of trajectory.py:91-94);
//     (+1), the subtraction, the clip, w*h and the running sum over frames are float32, in frame order;
//   * box areas are float64, summed over the common frames in numpy's pairwise order (np.sum of a contiguous
//     run: eight interleaved partial sums per block of <= 128, halves split at a multiple of 8 above that);
//   * union = (area1 + area2) - float64(inters); iou = float32(float64(inters) / union).
// Row u of `a` holds trajectory u from the current segment's first frame on; its first len_a[u] frames are the
// common ones (0: the trajectories do not overlap -> 0, association.py:36-37).
template <class F>
__device__ inline double np_block_sum(F&& f, int64_t lo, int64_t n) {
  if (n < 8) {
    double res = 0.;
    for (int64_t i = 0; i < n; ++i) res += f(lo + i);
    return res;
  }
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = f(lo + k);
  int64_t i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int k = 0; k < 8; ++k) r[k] += f(lo + i + k);
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += f(lo + i);
  return res;
}

template <class F>
__device__ inline double np_pairwise_sum(F&& f, int64_t n) {
  if (n <= 128) return np_block_sum(f, 0, n);
  struct Frame { int64_t lo, n; int stage; double left; };
  Frame st[48];               // depth <= log2(n / 64): far below 48 for any int64 n
  int sp = 0;
  double ret = 0.;
  st[sp++] = Frame{0, n, 0, 0.};
  while (sp > 0) {
    Frame& fr = st[sp - 1];
    if (fr.stage == 0) {
      if (fr.n <= 128) { ret = np_block_sum(f, fr.lo, fr.n); --sp; continue; }
      int64_t n2 = fr.n / 2; n2 -= n2 % 8;
      fr.stage = 1;
      st[sp++] = Frame{fr.lo, n2, 0, 0.};
    } else if (fr.stage == 1) {
      int64_t n2 = fr.n / 2; n2 -= n2 % 8;
      fr.left = ret;
      fr.stage = 2;
      st[sp++] = Frame{fr.lo + n2, fr.n - n2, 0, 0.};
    } else {
      ret = fr.left + ret;
      --sp;
    }
  }
  return ret;
}

__global__ __launch_bounds__(256) void traj_iou_tail_f64_kernel(const double* __restrict__ a,
                                                                const int32_t* __restrict__ len_a,
                                                                const double* __restrict__ b, int64_t U, int64_t N,
                                                                int64_t L, float* __restrict__ out) {
  const int64_t un = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (un >= U * N) return;
  const int64_t u = un / N, n = un - u * N;
  const int64_t len = len_a[u];
  if (len <= 0) { out[un] = 0.f; return; }
  const double* pa = a + u * L * 4;
  const double* pb = b + n * L * 4;
  float inters = 0.f;
  for (int64_t t = 0; t < len; ++t) {
    const double* p = pa + 4 * t;
    const double* q = pb + 4 * t;
    const float lo_x = (float)fmax(p[0], q[0]), hi_x = (float)fmin(p[2], q[2]);
    float w = (hi_x + 1.f) - lo_x;
    w = fmaxf(w, 0.f);
    const float lo_y = (float)fmax(p[1], q[1]), hi_y = (float)fmin(p[3], q[3]);
    float h = (hi_y + 1.f) - lo_y;
    h = fmaxf(h, 0.f);
    inters += w * h;
  }
  auto area_of = [](const double* r) {
    return [r](int64_t t) { return (r[4 * t + 2] - r[4 * t] + 1.) * (r[4 * t + 3] - r[4 * t + 1] + 1.); };
  };
  const double a1 = np_pairwise_sum(area_of(pa), len);
  const double a2 = np_pairwise_sum(area_of(pb), len);
  const double uni = (a1 + a2) - (double)inters;
  out[un] = (float)((double)inters / uni);
}

}  // namespace

extern "C" int tspn_traj_iou_tail_f64(const double* a, const int32_t* len_a, const double* b, int64_t U,
                                      int64_t N, int64_t L, float* out, void* stream) {
  TSPN_REQUIRE(U >= 0 && N >= 0 && L >= 0, TSPN_EINVAL, "tspn_traj_iou_tail_f64: bad sizes");
  if (U == 0 || N == 0) return TSPN_OK;
  TSPN_REQUIRE(len_a && out && (L == 0 || (a && b)), TSPN_EINVAL, "tspn_traj_iou_tail_f64: null pointer");
  TSPN_REQUIRE(U * N < ((int64_t)1 << 31) * 256, TSPN_EUNSUPPORTED, "tspn_traj_iou_tail_f64: too many pairs");
  hipLaunchKernelGGL(traj_iou_tail_f64_kernel, dim3((unsigned)tspn::ceil_div(U * N, 256)), dim3(256), 0,
                     TSPN_STREAM(stream), a, len_a, b, U, N, L, out);
  return tspn::check_launch("tspn_traj_iou_tail_f64");
}

extern "C" int tspn_traj_iou_f32(const float* boxes1, int64_t N1, const float* boxes2, int64_t N2,
                                 int64_t B, int64_t T, float* out, void* stream) {
  TSPN_REQUIRE(N1 >= 0 && N2 >= 0 && B >= 0 && T > 0, TSPN_EINVAL, "tspn_traj_iou_f32: bad sizes");
  if (boxes2 == nullptr) {
    boxes2 = boxes1;
    N2 = N1;
  }
  if (B == 0 || N1 == 0 || N2 == 0) return TSPN_OK;
  TSPN_REQUIRE(boxes1 && out, TSPN_EINVAL, "tspn_traj_iou_f32: null pointer");
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(boxes1) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(boxes2) & 15) == 0,
               TSPN_EINVAL, "tspn_traj_iou_f32: boxes must be 16-byte aligned");
  TSPN_REQUIRE(B < 65536, TSPN_EUNSUPPORTED, "tspn_traj_iou_f32: B too large");
  dim3 grid((unsigned)tspn::ceil_div(N1 * N2, 256), (unsigned)B);
  hipLaunchKernelGGL(traj_iou_kernel, grid, dim3(256), 0, TSPN_STREAM(stream),
                     reinterpret_cast<const float4*>(boxes1), N1,
                     reinterpret_cast<const float4*>(boxes2), N2, T, out);
  return tspn::check_launch("tspn_traj_iou_f32");
}
