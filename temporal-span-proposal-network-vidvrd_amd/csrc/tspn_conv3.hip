// a8: temporal context encoder — Conv1d(k=3, s=1, p=1) as an implicit GEMM on
// fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces `F.relu(self.conv(feats))` of DPNHead.forward
// (reference lib/modeling/relpn/dpn.py:70).  The same kernel computes the
// per-tracklet projections of the factorised pair form.
//
// GEMM view:  Y[m, n] = sum_{tap, ci} Wp[tap][ci][m] * X[ci][n + tap - 1]
//   m  = output channel (A operand = packed weights, m contiguous)
//   n  = flat column (b, t) over the whole batch: no per-sequence padding, the
//        +-1 taps that would cross a sequence boundary are masked to zero in
//        registers (lane-constant masks)
//   K  = 3 * Cin; the x tile is staged ONCE per channel chunk (with a 1-column
//        halo on both sides) and read three times at shifted columns, so x
//        traffic is 1/3 of an im2col formulation.
//
// Tiling: workgroup 128(m) x 128(n), 4 waves as 2x2, each wave 64x64 = 2x2
// MFMA blocks of 32x32 (64 accumulator VGPRs); K chunk = 16 channels x 3 taps;
// LDS double-buffered (66 KB -> 2 workgroups / CU), global->register->LDS
// staging with the next chunk's loads in flight under the MFMAs.
// fp32 MFMA runs at 64 FLOP/clk/SIMD (= 157 TF/s chip peak): one LDS dword per
// operand per 4096 FLOP, so LDS/L2 bandwidth is far from limiting; the design
// goal is simply to keep the matrix pipe issuing back-to-back.
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int KC = 16;             // input channels per chunk
constexpr int BNP = BN + 4;        // slots: column -1 .. BN (BN+2 used), padded
constexpr int THREADS = 256;
constexpr int A_STAGE = 3 * KC * BM;  // floats per A buffer
constexpr int B_STAGE = KC * BNP;     // floats per B buffer
constexpr size_t SMEM_BYTES = sizeof(float) * 2 * (A_STAGE + B_STAGE);

__global__ void pack_conv3_kernel(const float* __restrict__ W, int64_t M, int64_t Cin,
                                  int64_t split, float* __restrict__ packed) {
  const int64_t Mp = split > 0 ? 2 * M : M;
  const int64_t Cp = split > 0 ? split : Cin;
  const int64_t total = 3 * Cp * Mp;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = o % Mp;
    const int64_t ci = (o / Mp) % Cp;
    const int64_t tap = o / (Mp * Cp);
    const int64_t m = r < M ? r : r - M;
    const int64_t c = r < M ? ci : ci + split;
    packed[o] = W[(m * Cin + c) * 3 + tap];
  }
}

template <bool VEC_A>
__global__ __launch_bounds__(THREADS, 2) void conv3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n,
    int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);  // [2][3][KC][BM]
  float* Bs = As + 2 * A_STAGE;                     // [2][KC][BNP]

  // ---- workgroup -> tile: bijective XCD remap, then 8x8 tile groups so that the
  // workgroups resident on one XCD share weight panels and x panels in its L2.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  constexpr int GM = 8;
  const int group_sz = GM * tiles_n;
  const int group = wg / group_sz;
  const int first_m = group * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int in_group = wg - group * group_sz;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = tile_m * BM;
  const int64_t n0 = (int64_t)tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- per-thread staging descriptors (constant over the K loop)
  // B main: slot = 1 + (tid & 127) <-> column n0 + (tid & 127); rows (tid>>7)*8 .. +7
  const int bslot = 1 + (tid & 127);
  const int brow0 = (tid >> 7) * 8;
  const float* bptr = nullptr;
  {
    const int64_t n = n0 + (tid & 127);
    if (n < ncols) {
      const int64_t b = n / T;
      bptr = x + (b * Cin) * (int64_t)T + (n - b * T);
    }
  }
  // B halo: threads 0..31: row = tid & 15, side = tid >> 4 (0: column n0-1, 1: column n0+BN)
  const float* hptr = nullptr;
  const int hrow = tid & 15;
  const int hslot = (tid >> 4) & 1 ? BN + 1 : 0;
  if (tid < 32) {
    const int64_t n = (tid >> 4) ? n0 + BN : n0 - 1;
    if (n >= 0 && n < ncols) {
      const int64_t b = n / T;
      hptr = x + (b * Cin) * (int64_t)T + (n - b * T);
    }
  }

  float4 a_reg[6];
  float b_reg[8];
  float h_reg = 0.f;

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int idx4 = tid + r * THREADS;  // 0..1535
      const int row = idx4 >> 5;           // 0..47 = tap*KC + ci
      const int m = (idx4 & 31) * 4;
      const int tap = row / KC, ci = row - tap * KC;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + ci < Cin) {
        const float* src = Wp + ((int64_t)tap * Cin + c0 + ci) * M + m0 + m;
        if (VEC_A) {
          if (m0 + m < M) v = *reinterpret_cast<const float4*>(src);
        } else {
          if (m0 + m + 0 < M) v.x = src[0];
          if (m0 + m + 1 < M) v.y = src[1];
          if (m0 + m + 2 < M) v.z = src[2];
          if (m0 + m + 3 < M) v.w = src[3];
        }
      }
      a_reg[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ci = c0 + brow0 + r;
      b_reg[r] = (bptr != nullptr && ci < Cin) ? bptr[(int64_t)ci * T] : 0.f;
    }
    if (tid < 32) {
      const int ci = c0 + hrow;
      h_reg = (hptr != nullptr && ci < Cin) ? hptr[(int64_t)ci * T] : 0.f;
    }
  };

  auto store_chunk = [&](int buf) {
    float* Ab = As + buf * A_STAGE;
    float* Bb = Bs + buf * B_STAGE;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int idx4 = tid + r * THREADS;
      *reinterpret_cast<float4*>(Ab + idx4 * 4) = a_reg[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) Bb[(brow0 + r) * BNP + bslot] = b_reg[r];
    if (tid < 32) Bb[hrow * BNP + hslot] = h_reg;
  };

  // ---- lane-constant sequence-boundary masks for the +-1 taps
  bool mask_l[2], mask_r[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    const int t = (int)(n % T);
    mask_l[ni] = t != 0;
    mask_r[ni] = t != T - 1;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nchunks = (Cin + KC - 1) / KC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk((c + 1) * KC);

    const float* Ab = As + buf * A_STAGE + wm * 64 + li;
    const float* Bb = Bs + buf * B_STAGE + wn * 64 + li;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const int k = 2 * kk + kh;
        const float a0 = Ab[(tap * KC + k) * BM];
        const float a1 = Ab[(tap * KC + k) * BM + 32];
        float b0 = Bb[k * BNP + tap];
        float b1 = Bb[k * BNP + tap + 32];
        if (tap == 0) {
          b0 = mask_l[0] ? b0 : 0.f;
          b1 = mask_l[1] ? b1 : 0.f;
        } else if (tap == 2) {
          b0 = mask_r[0] ? b0 : 0.f;
          b1 = mask_r[1] ? b1 : 0.f;
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }

    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
  // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  For a fixed register the 32 lanes
  // of a half-wave write 32 consecutive t -> 128-B contiguous stores.
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    if (n >= ncols) continue;
    const int64_t b = n / T;
    const int64_t t = n - b * T;
    float* ycol = y + (b * M) * (int64_t)T + t;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          float v = acc[mi][ni][e];
          if (bias != nullptr) v += bias[m];
          if (relu) v = fmaxf(v, 0.f);
          ycol[(int64_t)m * T] = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int tspn_pack_conv3_f32(const float* W, int64_t M, int64_t Cin, int64_t split,
                                   float* packed, void* stream) {
  TSPN_REQUIRE(W && packed, TSPN_EINVAL, "tspn_pack_conv3_f32: null pointer");
  TSPN_REQUIRE(M > 0 && Cin > 0 && split >= 0, TSPN_EINVAL,
               "tspn_pack_conv3_f32: bad sizes M=%lld Cin=%lld split=%lld", (long long)M,
               (long long)Cin, (long long)split);
  TSPN_REQUIRE(split == 0 || Cin == 2 * split, TSPN_EINVAL,
               "tspn_pack_conv3_f32: split=%lld requires Cin == 2*split (Cin=%lld)",
               (long long)split, (long long)Cin);
  const int64_t total = 3 * M * Cin;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv3_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), W, M,
                     Cin, split, packed);
  return tspn::check_launch("tspn_pack_conv3_f32");
}

extern "C" int tspn_conv3_f32(const float* x, int64_t B, int64_t Cin, int64_t T,
                              const float* packed, int64_t M, const float* bias, int relu,
                              float* y, void* stream) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0, TSPN_EINVAL,
               "tspn_conv3_f32: bad sizes B=%lld Cin=%lld T=%lld M=%lld", (long long)B,
               (long long)Cin, (long long)T, (long long)M);
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && packed && y, TSPN_EINVAL, "tspn_conv3_f32: null pointer");
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv3_f32: dimension too large");
  const int64_t ncols = B * T;
  const int64_t tiles_m = tspn::ceil_div(M, BM);
  const int64_t tiles_n = tspn::ceil_div(ncols, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_f32: grid too large");
  const bool vec = (M % 4 == 0) && ((reinterpret_cast<uintptr_t>(packed) & 15) == 0);
  auto kern = vec ? conv3_mfma_kernel<true> : conv3_mfma_kernel<false>;
  static thread_local bool attr_set[2] = {false, false};
  if (!attr_set[vec]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)SMEM_BYTES);
    if (e != hipSuccess)
      return tspn::fail(TSPN_ELAUNCH, "tspn_conv3_f32: hipFuncSetAttribute: %s",
                        hipGetErrorString(e));
    attr_set[vec] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), SMEM_BYTES,
                     TSPN_STREAM(stream), x, packed, bias, y, (int)Cin, (int)T, (int)M, ncols,
                     (int)tiles_m, (int)tiles_n, relu);
  return tspn::check_launch("tspn_conv3_f32");
}
