// f3: temporal span decode + 1-D NMS per tracklet pair (gfx950).
//
// The reference stops at the DPN heads: `RelNMS.forward` is a bare expression
// (lib/modeling/relpn/rel_nms.py:14-15) and no span decode exists; it fixes only the anchor grid
// (`anchor = shift +- size/2`, location-major / size-minor: relpn/anchor_generator.py:48-59,76-104)
// and the constants nms_threshold = 0.5, top_k = NUM_DURATION_PROPOSALS (rel_nms.py:8-11).  The
// decode is build-defined (oracle.decode_spans states it):
//   candidate c = t*A + a: centre t, width sizes[a];  (d_c, d_w) = duration channels (2a, 2a+1)
//   ctr = t + d_c*size ; w = size*exp(min(d_w, log(1000/16)))          -- in float64
//   start/end = clip(ctr -+ w/2, 0, T) rounded to fp32 ; frames [floor(start), ceil(end))
//   rank by relationness logit (larger first, lower c first), first `pre_nms` enter a greedy NMS
//   (suppress j if inter > thr*union, float64 on the fp32 spans), first `top_k` survivors kept.
// The float64 arithmetic makes the result reproducible to the bit on any IEEE machine, so the
// span indices are checked bit-exactly against the CPU oracle.
//
// One workgroup per pair: bitonic sort of the A*T (logit, c) keys in LDS, span decode of the
// leaders, an m x m/64 suppression bit matrix built by all threads, then one wave walks the
// sorted list (lane w owns word w of the `removed` set; one shuffle + one LDS read per step).
#include <algorithm>
#include <cmath>

#include "tspn_common.h"

namespace {

constexpr int SP_THREADS = 256;
constexpr int SP_MAX_N = 4096;      // A*T candidates per pair
constexpr int SP_MAX_PRE = 1024;    // candidates entering NMS
constexpr int SP_MAX_A = 8;

struct SpanSizes {
  float v[SP_MAX_A];
};

__device__ __forceinline__ bool sp_before(float ka, int ia, float kb, int ib) {
  return ka > kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(SP_THREADS) void decode_spans_kernel(
    const float* __restrict__ heads, int A, int T, SpanSizes sizes, int top_k, double thr, int m,
    int n2, int64_t* __restrict__ out_anchor, int64_t* __restrict__ out_span,
    float* __restrict__ out_span_f, float* __restrict__ out_score, int64_t* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // [compact: start m, end m, logit m, cand m][union: sort keys n2 + idx n2 | mask m*mw u64]
  float* c_start = reinterpret_cast<float*>(smem_raw);
  float* c_end = c_start + SP_MAX_PRE;
  float* c_logit = c_end + SP_MAX_PRE;
  int* c_cand = reinterpret_cast<int*>(c_logit + SP_MAX_PRE);
  char* u = reinterpret_cast<char*>(c_cand + SP_MAX_PRE);
  float* s_key = reinterpret_cast<float*>(u);
  int* s_idx = reinterpret_cast<int*>(s_key + n2);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(u);
  __shared__ int s_kept[SP_MAX_PRE];
  __shared__ int s_nkept;

  const int tid = threadIdx.x;
  const int64_t p = blockIdx.x;
  const int n = A * T;
  const float* rel = heads + p * 3 * A * (int64_t)T;  // rows [0,A) relationness, [A,3A) duration
  const float* dur = rel + (int64_t)A * T;

  for (int c = tid; c < n2; c += SP_THREADS) {
    if (c < n) {
      const int t = c / A, a = c - t * A;
      s_key[c] = rel[a * T + t];
      s_idx[c] = c;
    } else {
      s_key[c] = -INFINITY;
      s_idx[c] = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += SP_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const float ki = s_key[i], kl = s_key[l];
          const int ii = s_idx[i], il = s_idx[l];
          const bool fwd = (i & k) == 0;
          const bool swap = fwd ? sp_before(kl, il, ki, ii) : sp_before(ki, ii, kl, il);
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
  // ---- decode the m leaders (float64), compact
  const double dw_clamp = 4.135166556742356;  // log(1000/16)
  for (int i = tid; i < m; i += SP_THREADS) {
    const int c = s_idx[i];
    const int t = c / A, a = c - t * A;
    const double sz = (double)sizes.v[a];
    const double dc = (double)dur[(2 * a) * T + t];
    const double dw = (double)dur[(2 * a + 1) * T + t];
    const double ctr = (double)t + dc * sz;
    const double w = sz * exp(fmin(dw, dw_clamp));
    const double lo = fmin(fmax(ctr - 0.5 * w, 0.0), (double)T);
    const double hi = fmin(fmax(ctr + 0.5 * w, 0.0), (double)T);
    c_start[i] = (float)lo;
    c_end[i] = (float)hi;
    c_logit[i] = s_key[i];
    c_cand[i] = c;
  }
  __syncthreads();
  // ---- suppression bit matrix: word (i, w) holds j in [64w, 64w+64), j > i
  const int mw = (m + 63) >> 6;
  for (int item = tid; item < m * mw; item += SP_THREADS) {
    const int i = item / mw, w = item - i * mw;
    unsigned long long bits = 0ull;
    const int j0 = w << 6;
    if (j0 + 63 > i) {
      const double si = (double)c_start[i], ei = (double)c_end[i];
      for (int b = 0; b < 64; ++b) {
        const int j = j0 + b;
        if (j > i && j < m) {
          const double sj = (double)c_start[j], ej = (double)c_end[j];
          const double inter = fmax(0.0, fmin(ei, ej) - fmax(si, sj));
          const double uni = (ei - si) + (ej - sj) - inter;
          if (inter > thr * uni) bits |= 1ull << b;
        }
      }
    }
    s_mask[item] = bits;
  }
  __syncthreads();
  // ---- greedy walk by one wave: lane w owns word w of `removed`
  if (tid < 64) {
    unsigned long long removed = 0ull;
    int nk = 0;
    for (int i = 0; i < m && nk < top_k; ++i) {
      const unsigned long long word = __shfl(removed, i >> 6);
      if (!((word >> (i & 63)) & 1ull)) {
        if (tid == 0) s_kept[nk] = i;
        ++nk;
        if (tid < mw) removed |= s_mask[i * mw + tid];
      }
    }
    if (tid == 0) s_nkept = nk;
  }
  __syncthreads();
  const int nk = s_nkept;
  if (tid == 0) out_count[p] = nk;
  for (int r = tid; r < top_k; r += SP_THREADS) {
    const int64_t o = p * top_k + r;
    if (r < nk) {
      const int i = s_kept[r];
      const float lo = c_start[i], hi = c_end[i];
      out_anchor[o] = c_cand[i];
      out_span_f[2 * o] = lo;
      out_span_f[2 * o + 1] = hi;
      long long is = (long long)floor((double)lo);
      is = is < 0 ? 0 : (is > T - 1 ? T - 1 : is);
      long long ie = (long long)ceil((double)hi);
      ie = ie < is + 1 ? is + 1 : (ie > T ? T : ie);
      out_span[2 * o] = is;
      out_span[2 * o + 1] = ie;
      out_score[o] = (float)(1.0 / (1.0 + exp(-(double)c_logit[i])));
    } else {
      out_anchor[o] = -1;
      out_span[2 * o] = -1;
      out_span[2 * o + 1] = -1;
      out_span_f[2 * o] = 0.f;
      out_span_f[2 * o + 1] = 0.f;
      out_score[o] = 0.f;
    }
  }
}

}  // namespace

extern "C" int tspn_decode_spans_f32(const float* heads, int64_t P, int64_t A, int64_t T,
                                     const float* sizes_host, int64_t top_k, double nms_threshold,
                                     int64_t pre_nms, int64_t* out_anchor, int64_t* out_span,
                                     float* out_span_f, float* out_score, int64_t* out_count,
                                     void* stream) {
  TSPN_REQUIRE(P >= 0 && A > 0 && T > 0 && top_k > 0 && pre_nms > 0, TSPN_EINVAL,
               "tspn_decode_spans_f32: bad sizes P=%lld A=%lld T=%lld top_k=%lld pre_nms=%lld",
               (long long)P, (long long)A, (long long)T, (long long)top_k, (long long)pre_nms);
  TSPN_REQUIRE(A <= SP_MAX_A && A * T <= SP_MAX_N, TSPN_EUNSUPPORTED,
               "tspn_decode_spans_f32: A=%lld (max %d), A*T=%lld (max %d)", (long long)A, SP_MAX_A,
               (long long)(A * T), SP_MAX_N);
  TSPN_REQUIRE(sizes_host, TSPN_EINVAL, "tspn_decode_spans_f32: null sizes");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(heads && out_anchor && out_span && out_span_f && out_score && out_count, TSPN_EINVAL,
               "tspn_decode_spans_f32: null pointer");
  const int n = (int)(A * T);
  const int m = (int)std::min<int64_t>(std::min<int64_t>(pre_nms, SP_MAX_PRE), n);
  TSPN_REQUIRE(top_k <= SP_MAX_PRE, TSPN_EUNSUPPORTED, "tspn_decode_spans_f32: top_k=%lld > %d",
               (long long)top_k, SP_MAX_PRE);
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  const int mw = (m + 63) >> 6;
  const size_t compact = (size_t)SP_MAX_PRE * 16;
  const size_t uni = std::max<size_t>((size_t)n2 * 8, (size_t)m * mw * 8);
  const size_t smem = compact + uni;
  TSPN_REQUIRE(smem + 8192 <= 160 * 1024, TSPN_EUNSUPPORTED,
               "tspn_decode_spans_f32: needs %zu B of LDS", smem);
  SpanSizes sz{};
  for (int a = 0; a < A; ++a) sz.v[a] = sizes_host[a];
  static tspn::LdsLimit lds;
  if (smem > 48 * 1024)
    if (int rc = lds.ensure(reinterpret_cast<const void*>(decode_spans_kernel), smem, "tspn_decode_spans_f32"))
      return rc;
  TSPN_REQUIRE(P < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_decode_spans_f32: P too large");
  // the kernel writes top_k columns; with top_k > m the tail is the "unused" filler
  hipLaunchKernelGGL(decode_spans_kernel, dim3((unsigned)P), dim3(SP_THREADS), smem,
                     TSPN_STREAM(stream), heads, (int)A, (int)T, sz, (int)top_k, nms_threshold,
                     m, n2, out_anchor, out_span, out_span_f, out_score, out_count);
  return tspn::check_launch("tspn_decode_spans_f32");
}
