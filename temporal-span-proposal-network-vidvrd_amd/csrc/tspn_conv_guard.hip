// A-posteriori accuracy guard of the temporal conv (round 6).
//
// The contract of the encoder is a plain fp32 Conv1d (reference lib/modeling/relpn/dpn.py:69-73) and north_star allows
// 1e-4 on its outputs.  Winograd F(6,3) is exact in real arithmetic but its fp32 error is only bounded RELATIVELY
// (|err| <= 64 eps sum|x_k||w_k|): on temporally independent, heavy-tailed features it reaches 4e-4 where
// the direct kernel holds 8e-5 (profiles/r3/conv_error_realistic.txt).  No a-priori bound is usable as a switch: the
// rigorous ones (64 eps max|x| max_m sum|w|, or 64 eps |x|_2 |w|_2 by Cauchy-Schwarz) already exceed 1e-4 on BASELINE's
// own synthetic inputs, whose measured error is 8e-6 (profiles/r6/conv_guard.md).  So the guard MEASURES: after the
// conv, `rows` workgroups recompute a few outputs each in float64 from the raw weights and compare.  The columns are
// not only random: the Winograd error is largest where an input outlier sits (its rounding error, amplified by the
// transforms, lands on the other frames of its sextet), so the input transform reports the sextet that holds the
// launch's largest |x| (`hot`) and that sextet's six frames are always among the checked columns.
// Result: float bits of the largest |y - y_ref| -> status word TSPN_STATUS_CONV_ERR (system-scope atomic max), number of
// outputs checked -> TSPN_STATUS_CONV_CHECKS.  The host reads them without a synchronisation before its next call and
// decides (model.py: warn once, use the direct kernel from then on).  ~20 us per call beside a 24 ms conv.
#include "tspn_common.h"
#include "tspn_status.h"

namespace {

constexpr int THREADS = 256;
constexpr int NSEXT = 4;        // sextets checked per launch: the hot one + three hashed ones
constexpr int NCOL = 6 * NSEXT;

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {   // small integer hash (xorshift-multiply)
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__global__ __launch_bounds__(THREADS) void conv3_spot_check_kernel(
    const float* __restrict__ x, int64_t B, int T, int Cp, const float* __restrict__ Wt, int M, int Cw, int split,
    const float* __restrict__ bias, const float* __restrict__ y, int64_t ldy, int relu, int nq,
    const unsigned long long* __restrict__ hot, unsigned seed, int32_t* __restrict__ status) {
  __shared__ double part[THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Mp = split > 0 ? 2 * M : M;
  // this workgroup's row: one per stratum of the Mp rows
  const int strata = gridDim.x;
  const int span = Mp >= strata ? Mp / strata : 1;
  const int m = (int)(((int64_t)blockIdx.x * span + mix(seed, blockIdx.x) % (unsigned)span) % Mp);
  const float* wrow = Wt + ((int64_t)(m < M ? m : m - M) * Cw + (m < M ? 0 : split)) * 3;   // [Cp][3] of this row
  const int64_t nsext = B * nq;
  float worst = 0.f;
  int checked = 0;
  for (int s = 0; s < NSEXT; ++s) {
    int64_t S;
    if (s == 0 && hot) S = (int64_t)(*hot & 0xffffffffull);
    else S = (int64_t)(((unsigned long long)mix(seed ^ 0xA5A5u, s) << 20 ^ mix(seed, 77 + s)) % (unsigned long long)nsext);
    if (S >= nsext) S = nsext - 1;
    const int64_t b = S / nq;
    const int q = (int)(S - b * nq);
    for (int i = 0; i < 6; ++i) {
      const int t = 6 * q + i;
      if (t >= T) break;                                     // uniform
      double acc = 0.0;
      for (int c = tid; c < Cp; c += THREADS) {
        const float w0 = wrow[3 * c], w1 = wrow[3 * c + 1], w2 = wrow[3 * c + 2];
        const float* xc = x + (b * T + t) * (int64_t)Cp + c;
        const float x0 = t > 0 ? xc[-(int64_t)Cp] : 0.f, x1 = xc[0], x2 = t + 1 < T ? xc[Cp] : 0.f;
        acc += (double)w0 * (double)x0 + (double)w1 * (double)x1 + (double)w2 * (double)x2;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      __syncthreads();                                       // `part` of the previous column has been read
      if (lane == 0) part[wave] = acc;
      __syncthreads();
      if (tid == 0) {
        double ref = part[0] + part[1] + part[2] + part[3] + (bias ? (double)bias[m] : 0.0);
        if (relu && ref < 0.0) ref = 0.0;
        const float got = y[(b * Mp + m) * ldy + t];
        worst = fmaxf(worst, (float)fabs((double)got - ref));
        ++checked;
      }
    }
  }
  if (tid == 0) {
    __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(status + TSPN_STATUS_CONV_ERR), __float_as_uint(worst),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(status + TSPN_STATUS_CONV_CHECKS, checked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

std::atomic<unsigned> g_seed{0x1234567u};

}  // namespace

extern "C" int tspn_conv3_spot_check_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* W, int64_t M,
                                         int64_t Cw, int64_t split, const float* bias, int relu, const float* y,
                                         int64_t ldy, const uint64_t* hot, int64_t rows, void* stream) {
  const char* what = "tspn_conv3_spot_check_f32";
  TSPN_REQUIRE(B >= 0 && T > 0 && Cin > 0 && M > 0 && Cw > 0 && split >= 0 && rows >= 0 && ldy >= T, TSPN_EINVAL,
               "%s: bad sizes", what);
  TSPN_REQUIRE(split == 0 ? Cw == Cin : (Cw == 2 * split && Cin == split), TSPN_EINVAL,
               "%s: W is [M, Cw, 3] with Cw = Cin, or Cw = 2 split and Cin = split (got Cin=%lld Cw=%lld split=%lld)", what,
               (long long)Cin, (long long)Cw, (long long)split);
  if (B == 0 || rows == 0) return TSPN_OK;
  TSPN_REQUIRE(x && W && y, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE(T < (1 << 30) && Cin < (1 << 30) && M < (1 << 29) && rows < (1 << 16), TSPN_EUNSUPPORTED, "%s: too large", what);
  int32_t* status = tspn::status_device_ptr();
  TSPN_REQUIRE(status, TSPN_EINVAL, "%s: no device status block attached (tspn_status_attach): nowhere to report", what);
  const unsigned seed = g_seed.fetch_add(0x9E3779B9u, std::memory_order_relaxed);
  hipLaunchKernelGGL(conv3_spot_check_kernel, dim3((unsigned)rows), dim3(THREADS), 0, TSPN_STREAM(stream), x, B, (int)T,
                     (int)Cin, W, (int)M, (int)Cw, (int)split, bias, y, ldy, relu, (int)tspn::ceil_div(T, 6),
                     reinterpret_cast<const unsigned long long*>(hot), seed, status);
  return tspn::check_launch(what);
}
