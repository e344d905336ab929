// a15: block-L1 normalisation of the baseline pair feature (gfx950).
//
// Replaces VRDataset._feature_preprocess (reference
// lib/dataset/vrdataset.py:219-243) and utils.normalize
// (lib/utils/miscellaneous.py:32-35): `nblocks` consecutive `block`-wide column
// ranges starting at `first` are divided by their L1 norm; a zero norm is
// replaced by 1 (the row stays zero).  In place, HBM-bound: one wave per
// (row, block) — coalesced 256-B reads, a 6-step shuffle tree, coalesced
// rewrite; the second pass hits L2.
#include "tspn_common.h"

namespace {

__global__ __launch_bounds__(256) void block_l1_kernel(float* __restrict__ feats, int64_t P,
                                                       int64_t ld, int64_t first, int64_t block,
                                                       int64_t nblocks) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= P * nblocks) return;
  const int64_t row = item / nblocks, k = item - row * nblocks;
  float* s = feats + row * ld + first + k * block;
  float acc = 0.f;
  for (int64_t c = lane; c < block; c += 64) acc += fabsf(s[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  const float l1 = acc == 0.f ? 1.f : acc;
  for (int64_t c = lane; c < block; c += 64) s[c] = s[c] / l1;
}

}  // namespace

extern "C" int tspn_feature_preprocess_f32(float* feats, int64_t P, int64_t F, int64_t ld,
                                           int64_t first, int64_t block, int64_t nblocks,
                                           void* stream) {
  TSPN_REQUIRE(P >= 0 && F > 0 && ld >= F && first >= 0 && block > 0 && nblocks >= 0,
               TSPN_EINVAL, "tspn_feature_preprocess_f32: bad sizes");
  TSPN_REQUIRE(first + block * nblocks <= F, TSPN_EINVAL,
               "tspn_feature_preprocess_f32: blocks [%lld, %lld) exceed F=%lld", (long long)first,
               (long long)(first + block * nblocks), (long long)F);
  if (P == 0 || nblocks == 0) return TSPN_OK;
  TSPN_REQUIRE(feats, TSPN_EINVAL, "tspn_feature_preprocess_f32: null pointer");
  const int64_t nb = tspn::ceil_div(P * nblocks, 4);
  TSPN_REQUIRE(nb < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_feature_preprocess_f32: grid too large");
  hipLaunchKernelGGL(block_l1_kernel, dim3((unsigned)nb), dim3(256), 0, TSPN_STREAM(stream),
                     feats, P, ld, first, block, nblocks);
  return tspn::check_launch("tspn_feature_preprocess_f32");
}
