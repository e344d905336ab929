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

}  // namespace

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
