// The N^2 pairwise tracklet-feature builder and its small companions (gfx950).
//
// The reference never builds pair tensors on line: they arrive precomputed from
// HDF5 (lib/dataset/vrdataset.py:190-217).  What it fixes is the pair ORDER
// (lib/modeling/predict.py:133-140), the box convention (+1 pixel-inclusive,
// lib/modeling/trajectory.py:96-106) and the consumer layout "NxCxT"
// (lib/modeling/relpn/dpn_anchor.py:38).  These kernels are HBM-bound byte
// movers: coalesced 128-B rows on both sides, LDS-staged transposes, no GEMM.
#include <algorithm>

#include "tspn_common.h"

namespace {

// ---------------------------------------------------------------- pair order
__global__ void pair_index_kernel(int64_t N, int64_t base, int64_t* __restrict__ pairs) {
  const int64_t P = N * (N - 1);
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = p / (N - 1);
    const int64_t r = p - i * (N - 1);
    const int64_t j = r + (r >= i ? 1 : 0);
    pairs[2 * p] = base + i;
    pairs[2 * p + 1] = base + j;
  }
}

// ------------------------------------------------ [R,T,D] -> [R,D,T] transpose
// One workgroup moves a 32(t) x 32(d) tile through LDS: reads are 128-B rows
// along d, writes are 128-B rows along t.  `src_row` (optional) redirects the
// source row (pair gather: subject / object tracklet of pair p); the
// destination row is blockIdx.z with `dst_rows_per_src` channel groups.
constexpr int TT = 32;

__global__ __launch_bounds__(256) void transpose_gather_kernel(
    const float* __restrict__ src, const int64_t* __restrict__ pairs, int64_t T, int64_t D,
    float* __restrict__ dst, int gather) {
  __shared__ float tile[TT][TT + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t d0 = (int64_t)blockIdx.x * TT;
  const int64_t t0 = (int64_t)blockIdx.y * TT;
  const int64_t z = blockIdx.z;
  int64_t srow, drow_off;
  if (gather) {
    // z = 2*p + role; destination [P, 2D, T]: channel offset role*D
    srow = pairs[z];
    drow_off = (z >> 1) * 2 * D + (z & 1) * D;
  } else {
    srow = z;
    drow_off = z * D;
  }
  const float* s = src + srow * T * D;
#pragma unroll
  for (int r = 0; r < TT; r += 8) {
    const int64_t t = t0 + ty + r, d = d0 + tx;
    tile[ty + r][tx] = (t < T && d < D) ? s[t * D + d] : 0.f;
  }
  __syncthreads();
  float* o = dst + drow_off * T;
#pragma unroll
  for (int r = 0; r < TT; r += 8) {
    const int64_t d = d0 + ty + r, t = t0 + tx;
    if (d < D && t < T) o[d * T + t] = tile[tx][ty + r];
  }
}

// (round 6) The same move for whole tracklets: a workgroup = 64 channels x ALL T frames of one (pair, role).  The destination
// of such a block, [64 rows][T], is ONE contiguous run of 64 T floats, so the tile is laid down in LDS in the destination's
// own order and leaves as plain 16-byte copies (whole lines; the 32 x 32 form above writes 128-byte pieces that straddle
// two lines whenever a row is not a multiple of 128 bytes: T = 150 -> 600).  Loads are 16 bytes per lane too: lane =
// (frame t & 15, channel quad), a quad = 64 contiguous bytes of a frame row, and the four floats of a lane go to four rows
// of the tile with the lanes of a wave along t: bank-consecutive LDS writes.
// The workgroup index runs CHANNEL-BLOCK-major (all pairs of channels [64 b, 64 b + 64), then the next block): every
// workgroup in flight then reads from the same 64-channel slice of the N tracklets -- N T 256 bytes, 1.2 MB at N = 32,
// T = 150: resident in every XCD's L2 -- where the pair-major order of the 32 x 32 form re-fetched the object tracklet of
// almost every pair from beyond L2 (FETCH_SIZE 1.24 GB for 39 MB of unique input, profiles/r6/pair_builder.csv).
// Needs D % 64 == 0, T <= TG_TMAX and 16-byte aligned operands; the 32 x 32 form serves everything else.
constexpr int TG_C = 64;
constexpr int TG_TMAX = 160;          // 40 KB of LDS per workgroup

__global__ __launch_bounds__(256) void transpose_gather_rows_kernel(
    const float* __restrict__ src, const int64_t* __restrict__ pairs, int T, int64_t D, float* __restrict__ dst,
    int gather, int64_t nz) {
  extern __shared__ __attribute__((aligned(16))) float tile_rows[];      // [64][T], the destination block's own order
  const int64_t wg = blockIdx.x;
  const int64_t dblk = wg / nz, z = wg - dblk * nz;
  int64_t srow, drow_off;
  if (gather) {
    srow = pairs[z];                          // z = 2 p + role; destination [P, 2D, T]: channel offset role * D
    drow_off = (z >> 1) * 2 * D + (z & 1) * D;
  } else {
    srow = z;
    drow_off = z * D;
  }
  const int tid = threadIdx.x;
  const int tl = tid & 15, cq = tid >> 4;     // a wave: 16 frames x 4 channel quads
  const float* s = src + srow * T * D + dblk * TG_C + 4 * cq;
  for (int t = tl; t < T; t += 16) {
    const float4 v = *reinterpret_cast<const float4*>(s + (int64_t)t * D);
    float* w = tile_rows + (4 * cq) * T + t;
    w[0] = v.x;
    w[T] = v.y;
    w[2 * T] = v.z;
    w[3 * T] = v.w;
  }
  __syncthreads();
  float4* o = reinterpret_cast<float4*>(dst + (drow_off + dblk * TG_C) * T);
  const float4* ti = reinterpret_cast<const float4*>(tile_rows);
  const int n4 = TG_C * T / 4;
  for (int i = tid; i < n4; i += 256) o[i] = ti[i];
}

// --------------------------------------------------- relative box geometry
// One lane per (pair, frame).  The motion channels need the previous frame's
// offsets: they come from the neighbouring lane by a wavefront shuffle; only the
// first lane of a wave (or of a row) recomputes them from the boxes.
struct Geo {
  float g0, g1, g2, g3, g4, g7;
};

__device__ __forceinline__ Geo geo_at(const float4 s, const float4 o) {
  Geo g;
  const float ws = s.z - s.x + 1.f, hs = s.w - s.y + 1.f;
  const float wo = o.z - o.x + 1.f, ho = o.w - o.y + 1.f;
  const float cxs = 0.5f * (s.x + s.z), cys = 0.5f * (s.y + s.w);
  const float cxo = 0.5f * (o.x + o.z), cyo = 0.5f * (o.y + o.w);
  g.g0 = (cxs - cxo) / wo;
  g.g1 = (cys - cyo) / ho;
  g.g2 = logf(ws / wo);
  g.g3 = logf(hs / ho);
  const float iw = fmaxf(fminf(s.z, o.z) + 1.f - fmaxf(s.x, o.x), 0.f);
  const float ih = fmaxf(fminf(s.w, o.w) + 1.f - fmaxf(s.y, o.y), 0.f);
  const float inter = iw * ih;
  const float as = ws * hs, ao = wo * ho;
  g.g4 = inter / (as + ao - inter);
  g.g7 = inter / as;
  return g;
}

__global__ __launch_bounds__(256) void pair_geometry_kernel(
    const float4* __restrict__ boxes, const int64_t* __restrict__ pairs, int64_t P, int64_t T,
    float* __restrict__ out) {
  const int64_t p = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t tc = t < T ? t : T - 1;
  const float4* sb = boxes + pairs[2 * p] * T;
  const float4* ob = boxes + pairs[2 * p + 1] * T;
  const Geo g = geo_at(sb[tc], ob[tc]);
  float p0 = __shfl_up(g.g0, 1);
  float p1 = __shfl_up(g.g1, 1);
  if (lane == 0 && tc > 0) {
    const Geo gp = geo_at(sb[tc - 1], ob[tc - 1]);
    p0 = gp.g0;
    p1 = gp.g1;
  }
  const float g5 = tc > 0 ? g.g0 - p0 : 0.f;
  const float g6 = tc > 0 ? g.g1 - p1 : 0.f;
  if (t < T) {
    float* o = out + p * TSPN_GEOM_CHANNELS * T + t;
    o[0 * T] = g.g0;
    o[1 * T] = g.g1;
    o[2 * T] = g.g2;
    o[3 * T] = g.g3;
    o[4 * T] = g.g4;
    o[5 * T] = g5;
    o[6 * T] = g6;
    o[7 * T] = g.g7;
  }
}

// ----------------------------------------------------------- RelOIPool (mean)
// x[R,T,D] -> out[R,D]: thread per (r,d), coalesced along d, frames in order.
__global__ void temporal_mean_td_kernel(const float* __restrict__ x, int64_t R, int64_t T,
                                        int64_t D, float* __restrict__ out, int mean = 1) {
  const int64_t total = R * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, d = i - r * D;
    const float* s = x + r * T * D + d;
    float acc = 0.f;
    for (int64_t t = 0; t < T; ++t) acc += s[t * D];
    out[i] = mean ? acc / (float)T : acc;
  }
}

// The same for D % 4 == 0 and 16-byte aligned rows: a thread owns four channels (16-byte loads, whole 1-KiB row
// segments per wave) and keeps TEN frames in flight; every channel is still summed frame by frame in order, so the
// result is bit for bit the scalar kernel's (round 4: 0.236 -> 0.128 ms per 629 MB at cfg2 = 4.9 TB/s; the scalar form left the loop
// latency-bound at 2.7 TB/s).
__global__ __launch_bounds__(256) void temporal_mean_td4_kernel(const float4* __restrict__ x, int64_t R, int64_t T,
                                                                int64_t D4, float4* __restrict__ out) {
  const int64_t total = R * D4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D4, d = i - r * D4;
    const float4* s = x + r * T * D4 + d;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t t = 0;
    for (; t + 10 <= T; t += 10) {
      float4 v[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) v[k] = s[(t + k) * D4];
#pragma unroll
      for (int k = 0; k < 10; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    for (; t < T; ++t) {
      const float4 v = s[t * D4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float ft = (float)T;
    out[i] = make_float4(acc.x / ft, acc.y / ft, acc.z / ft, acc.w / ft);
  }
}

// x[R,C,T] -> out[R,C]: one wave per (r,c) row, lanes stride over t, shuffle tree.
__global__ __launch_bounds__(256) void temporal_mean_ct_kernel(const float* __restrict__ x,
                                                               int64_t rows, int64_t T,
                                                               float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = x + row * T;
  float acc = 0.f;
  for (int64_t t = lane; t < T; t += 64) acc += s[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) out[row] = acc / (float)T;
}

// out[P, 2D] = cat(src[pairs[p,0]], src[pairs[p,1]])
__global__ void pair_rows_kernel(const float* __restrict__ src, int64_t D,
                                 const int64_t* __restrict__ pairs, int64_t P,
                                 float* __restrict__ out) {
  const int64_t total = P * 2 * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pr = i / D, d = i - pr * D;  // pr = 2*p + role
    out[i] = src[pairs[pr] * D + d];
  }
}

int grid_for(int64_t total, int block = 256, int cap = 8192) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(tspn::ceil_div(total, block), cap));
}

}  // namespace

extern "C" int tspn_pair_index_i64(int64_t N, int64_t base, int64_t* pairs, void* stream) {
  TSPN_REQUIRE(N >= 0, TSPN_EINVAL, "tspn_pair_index_i64: N=%lld", (long long)N);
  if (N < 2) return TSPN_OK;
  TSPN_REQUIRE(pairs, TSPN_EINVAL, "tspn_pair_index_i64: null pointer");
  hipLaunchKernelGGL(pair_index_kernel, dim3(grid_for(N * (N - 1))), dim3(256), 0,
                     TSPN_STREAM(stream), N, base, pairs);
  return tspn::check_launch("tspn_pair_index_i64");
}

extern "C" int tspn_transpose_td_f32(const float* x, int64_t R, int64_t T, int64_t D, float* out,
                                     void* stream) {
  TSPN_REQUIRE(R >= 0 && T > 0 && D > 0, TSPN_EINVAL, "tspn_transpose_td_f32: bad sizes");
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_transpose_td_f32: null pointer");
  const int64_t gy = tspn::ceil_div(T, TT);
  TSPN_REQUIRE(gy < 65536 && R < 65536, TSPN_EUNSUPPORTED, "tspn_transpose_td_f32: grid too large");
  dim3 grid((unsigned)tspn::ceil_div(D, TT), (unsigned)gy, (unsigned)R);
  hipLaunchKernelGGL(transpose_gather_kernel, grid, dim3(256), 0, TSPN_STREAM(stream), x,
                     (const int64_t*)nullptr, T, D, out, 0);
  return tspn::check_launch("tspn_transpose_td_f32");
}

extern "C" int tspn_pair_gather_f32(const float* feats, const float* boxes, int64_t NT, int64_t T,
                                    int64_t D, const int64_t* pairs, int64_t P, float* out_feat,
                                    float* out_geom, void* stream) {
  TSPN_REQUIRE(NT >= 0 && T > 0 && D > 0 && P >= 0, TSPN_EINVAL, "tspn_pair_gather_f32: bad sizes");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(pairs, TSPN_EINVAL, "tspn_pair_gather_f32: null pairs");
  TSPN_REQUIRE(!out_feat || feats, TSPN_EINVAL, "tspn_pair_gather_f32: out_feat needs feats");
  TSPN_REQUIRE(!out_geom || boxes, TSPN_EINVAL, "tspn_pair_gather_f32: out_geom needs boxes");
  TSPN_REQUIRE(!boxes || (reinterpret_cast<uintptr_t>(boxes) & 15) == 0, TSPN_EINVAL,
               "tspn_pair_gather_f32: boxes must be 16-byte aligned");
  hipStream_t s = TSPN_STREAM(stream);
  const bool rows_form = out_feat && D % TG_C == 0 && T <= TG_TMAX && ((reinterpret_cast<uintptr_t>(feats) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(out_feat) & 15) == 0) && (2 * P) * (D / TG_C) < (1LL << 31);
  if (rows_form) {
    const size_t smem = sizeof(float) * TG_C * T;
    hipLaunchKernelGGL(transpose_gather_rows_kernel, dim3((unsigned)((2 * P) * (D / TG_C))), dim3(256), smem, s, feats, pairs,
                       (int)T, D, out_feat, 1, 2 * P);
    int rc = tspn::check_launch("tspn_pair_gather_f32(feat)");
    if (rc) return rc;
  } else if (out_feat) {
    const int64_t gy = tspn::ceil_div(T, TT);
    // blockIdx.z is limited to 65535: walk the pair list in slabs
    const int64_t zmax = 65534;
    for (int64_t z0 = 0; z0 < 2 * P; z0 += zmax) {
      const int64_t nz = std::min<int64_t>(zmax, 2 * P - z0);
      TSPN_REQUIRE(gy < 65536, TSPN_EUNSUPPORTED, "tspn_pair_gather_f32: T too large");
      dim3 grid((unsigned)tspn::ceil_div(D, TT), (unsigned)gy, (unsigned)nz);
      hipLaunchKernelGGL(transpose_gather_kernel, grid, dim3(256), 0, s, feats, pairs + z0, T, D,
                         out_feat + (z0 / 2) * 2 * D * T, 1);
      int rc = tspn::check_launch("tspn_pair_gather_f32(feat)");
      if (rc) return rc;
    }
  }
  if (out_geom) {
    const int64_t ymax = 65535;
    for (int64_t p0 = 0; p0 < P; p0 += ymax) {
      const int64_t np = std::min<int64_t>(ymax, P - p0);
      dim3 grid((unsigned)tspn::ceil_div(T, 256), (unsigned)np);
      hipLaunchKernelGGL(pair_geometry_kernel, grid, dim3(256), 0, s,
                         reinterpret_cast<const float4*>(boxes), pairs + 2 * p0, np, T,
                         out_geom + p0 * TSPN_GEOM_CHANNELS * T);
      int rc = tspn::check_launch("tspn_pair_gather_f32(geom)");
      if (rc) return rc;
    }
  }
  return TSPN_OK;
}

extern "C" int tspn_temporal_mean_f32(const float* x, int64_t R, int64_t T, int64_t Cdim,
                                      int layout_tc, float* out, void* stream) {
  TSPN_REQUIRE(R >= 0 && T > 0 && Cdim > 0, TSPN_EINVAL, "tspn_temporal_mean_f32: bad sizes");
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_temporal_mean_f32: null pointer");
  hipStream_t s = TSPN_STREAM(stream);
  if (layout_tc && Cdim % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    hipLaunchKernelGGL(temporal_mean_td4_kernel, dim3(grid_for(R * (Cdim / 4))), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), R, T, Cdim / 4, reinterpret_cast<float4*>(out));
  } else if (layout_tc) {
    hipLaunchKernelGGL(temporal_mean_td_kernel, dim3(grid_for(R * Cdim)), dim3(256), 0, s, x, R, T,
                       Cdim, out);
  } else {
    const int64_t rows = R * Cdim;
    const int64_t nb = tspn::ceil_div(rows, 4);
    TSPN_REQUIRE(nb < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_temporal_mean_f32: grid too large");
    hipLaunchKernelGGL(temporal_mean_ct_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, rows, T,
                       out);
  }
  return tspn::check_launch("tspn_temporal_mean_f32");
}

extern "C" int tspn_temporal_sum_f32(const float* x, int64_t R, int64_t T, int64_t Cdim, float* out, void* stream) {
  TSPN_REQUIRE(R >= 0 && T > 0 && Cdim > 0, TSPN_EINVAL, "tspn_temporal_sum_f32: bad sizes");
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_temporal_sum_f32: null pointer");
  hipLaunchKernelGGL(temporal_mean_td_kernel, dim3(grid_for(R * Cdim)), dim3(256), 0, TSPN_STREAM(stream), x, R, T,
                     Cdim, out, 0);
  return tspn::check_launch("tspn_temporal_sum_f32");
}

extern "C" int tspn_pair_rows_f32(const float* src, int64_t NT, int64_t D, const int64_t* pairs,
                                  int64_t P, float* out, void* stream) {
  TSPN_REQUIRE(NT >= 0 && D > 0 && P >= 0, TSPN_EINVAL, "tspn_pair_rows_f32: bad sizes");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(src && pairs && out, TSPN_EINVAL, "tspn_pair_rows_f32: null pointer");
  hipLaunchKernelGGL(pair_rows_kernel, dim3(grid_for(P * 2 * D)), dim3(256), 0,
                     TSPN_STREAM(stream), src, D, pairs, P, out);
  return tspn::check_launch("tspn_pair_rows_f32");
}
