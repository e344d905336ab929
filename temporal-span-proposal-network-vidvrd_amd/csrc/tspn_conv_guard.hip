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
// decides (model.py: warn once, use the direct kernel from then on).
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

// One workgroup = one output row at all NCOL columns AT ONCE: a thread owns the channels tid, tid + 256, ... -- for each it
// loads its three taps and the eight input frames a sextet's six outputs need (every load of a channel step is independent:
// the first form, column after column with a block reduction each, spent 0.4 ms per call waiting for one load after
// another), accumulates 24 float64 partial sums in registers, and the block reduces them once.
__global__ __launch_bounds__(THREADS) void conv3_spot_check_kernel(
    const float* __restrict__ x, int64_t B, int T, int Cp, const float* __restrict__ Wt, int M, int Cw, int split,
    const float* __restrict__ bias, const float* __restrict__ y, int64_t ldy, int relu, int nq,
    unsigned long long* __restrict__ scratch, unsigned seed, int32_t* __restrict__ status) {
  __shared__ double part[THREADS / 64][NCOL];
  __shared__ unsigned worst_bits, nchecked;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { worst_bits = 0u; nchecked = 0u; }
  const int Mp = split > 0 ? 2 * M : M;
  // this workgroup's row: one per stratum of the Mp rows
  const int strata = gridDim.x;
  const int span = Mp >= strata ? Mp / strata : 1;
  const int m = (int)(((int64_t)blockIdx.x * span + mix(seed, blockIdx.x) % (unsigned)span) % Mp);
  const float* wrow = Wt + ((int64_t)(m < M ? m : m - M) * Cw + (m < M ? 0 : split)) * 3;   // [Cp][3] of this row
  const int64_t nsext = B * nq;
  // the hot sextet: the largest key over the slots the input transform reported into (0 = nothing reported)
  unsigned long long hotkey = 0;
  {
    const unsigned long long* slots = scratch + TSPN_CONV_CHECK_HOT_OFFSET / 8;
    unsigned long long k = lane < TSPN_CONV_CHECK_HOT_SLOTS ? slots[32 * lane] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(k, o, 64);
      k = other > k ? other : k;
    }
    hotkey = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)k);
  }
  int64_t bs[NSEXT];
  int qs[NSEXT];
#pragma unroll
  for (int s = 0; s < NSEXT; ++s) {
    int64_t S;
    if (s == 0 && (hotkey >> 32) != 0) S = (int64_t)(hotkey & 0xffffffffull);
    else S = (int64_t)(((unsigned long long)mix(seed ^ 0xA5A5u, s) << 20 ^ mix(seed, 77 + s)) % (unsigned long long)nsext);
    if (S >= nsext) S = nsext - 1;
    bs[s] = S / nq;
    qs[s] = (int)(S - bs[s] * nq);
  }
  const float* xs[NSEXT];                                    // first frame of each sextet's tracklet
#pragma unroll
  for (int s = 0; s < NSEXT; ++s) xs[s] = x + bs[s] * T * (int64_t)Cp;
  double acc[NCOL];
#pragma unroll
  for (int j = 0; j < NCOL; ++j) acc[j] = 0.0;
  for (int c = tid; c < Cp; c += THREADS) {
    const double w0 = wrow[3 * c], w1 = wrow[3 * c + 1], w2 = wrow[3 * c + 2];
    // frames 6 q - 1 .. 6 q + 6 of this channel for all four sextets: 32 UNCONDITIONAL loads from clamped frames in one
    // batch, then the selects.  With a load under its condition hipcc emits a branch and a wait per load (and sinks the
    // load back under the select unless the value is pinned): 256 dependent round trips per thread, 375 us per call
    // (profiles/r6/conv_guard.md)
    float xr[NSEXT][8];
#pragma unroll
    for (int s = 0; s < NSEXT; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = 6 * qs[s] + i - 1;
        const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
        xr[s][i] = xs[s][(int64_t)tc * Cp + c];
      }
    static_assert(NSEXT == 4, "the two pins below");
    static_assert(TSPN_CONV_CHECK_HOT_SLOTS == 64 && TSPN_CONV_CHECK_HOT_OFFSET % 8 == 0, "one slot per lane");
#pragma unroll
    for (int h = 0; h < 2; ++h)
      asm volatile("" : "+v"(xr[2 * h][0]), "+v"(xr[2 * h][1]), "+v"(xr[2 * h][2]), "+v"(xr[2 * h][3]), "+v"(xr[2 * h][4]),
                        "+v"(xr[2 * h][5]), "+v"(xr[2 * h][6]), "+v"(xr[2 * h][7]), "+v"(xr[2 * h + 1][0]), "+v"(xr[2 * h + 1][1]),
                        "+v"(xr[2 * h + 1][2]), "+v"(xr[2 * h + 1][3]), "+v"(xr[2 * h + 1][4]), "+v"(xr[2 * h + 1][5]),
                        "+v"(xr[2 * h + 1][6]), "+v"(xr[2 * h + 1][7]));
#pragma unroll
    for (int s = 0; s < NSEXT; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = 6 * qs[s] + i - 1;
        xr[s][i] = (t >= 0 && t < T) ? xr[s][i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
        acc[6 * s + i] += w0 * (double)xr[s][i] + w1 * (double)xr[s][i + 1] + w2 * (double)xr[s][i + 2];
    }
  }
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    double v = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) part[wave][j] = v;
  }
  __syncthreads();
  if (tid < NCOL) {
    const int s = tid / 6, i = tid - 6 * s, t = 6 * qs[s] + i;
    if (t < T) {
      double ref = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid] + (bias ? (double)bias[m] : 0.0);
      if (relu && ref < 0.0) ref = 0.0;
      const float got = y[(bs[s] * Mp + m) * ldy + t];
      atomicMax(&worst_bits, __float_as_uint((float)fabs((double)got - ref)));
      atomicAdd(&nchecked, 1u);
    }
  }
  __syncthreads();
  // The workgroups meet in device memory (scratch words 1 - 3); only the LAST one to arrive touches the status block in host
  // memory.  With every workgroup doing its own two system-scope atomics the kernel took 285 us: 256 PCIe round trips on
  // two addresses, one behind the other (profiles/r6/conv_guard.md).
  if (tid == 0) {
    unsigned* sc = reinterpret_cast<unsigned*>(scratch);
    __hip_atomic_fetch_max(sc + 2, worst_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sc + 4, nchecked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_fetch_add(sc + 6, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived + 1 == gridDim.x) {
      const unsigned wb = __hip_atomic_exchange(sc + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nc = __hip_atomic_exchange(sc + 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sc + 6, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
      __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(status + TSPN_STATUS_CONV_ERR), wb, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(status + TSPN_STATUS_CONV_CHECKS, (int)nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

std::atomic<unsigned> g_seed{0x1234567u};

}  // namespace

extern "C" int tspn_conv3_spot_check_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* W, int64_t M,
                                         int64_t Cw, int64_t split, const float* bias, int relu, const float* y,
                                         int64_t ldy, uint64_t* scratch, int64_t rows, void* stream) {
  const char* what = "tspn_conv3_spot_check_f32";
  TSPN_REQUIRE(B >= 0 && T > 0 && Cin > 0 && M > 0 && Cw > 0 && split >= 0 && rows >= 0 && ldy >= T, TSPN_EINVAL,
               "%s: bad sizes", what);
  TSPN_REQUIRE(split == 0 ? Cw == Cin : (Cw == 2 * split && Cin == split), TSPN_EINVAL,
               "%s: W is [M, Cw, 3] with Cw = Cin, or Cw = 2 split and Cin = split (got Cin=%lld Cw=%lld split=%lld)", what,
               (long long)Cin, (long long)Cw, (long long)split);
  if (B == 0 || rows == 0) return TSPN_OK;
  TSPN_REQUIRE(x && W && y && scratch, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, TSPN_EINVAL, "%s: scratch must be 8-byte aligned", what);
  TSPN_REQUIRE(T < (1 << 30) && Cin < (1 << 30) && M < (1 << 29) && rows < (1 << 16), TSPN_EUNSUPPORTED, "%s: too large", what);
  int32_t* status = tspn::status_device_ptr();
  TSPN_REQUIRE(status, TSPN_EINVAL, "%s: no device status block attached (tspn_status_attach): nowhere to report", what);
  const unsigned seed = g_seed.fetch_add(0x9E3779B9u, std::memory_order_relaxed);
  hipLaunchKernelGGL(conv3_spot_check_kernel, dim3((unsigned)rows), dim3(THREADS), 0, TSPN_STREAM(stream), x, B, (int)T,
                     (int)Cin, W, (int)M, (int)Cw, (int)split, bias, y, ldy, relu, (int)tspn::ceil_div(T, 6),
                     reinterpret_cast<unsigned long long*>(scratch), seed, status);
  return tspn::check_launch(what);
}
