// a8/a10: relationness + span-regression heads as ONE [H, C] 1x1 GEMM on fp32
// MFMA (v_mfma_f32_16x16x4_f32), with the pair activation formed on the fly.
//
// Replaces `self.duration_pred(t)` (reference lib/modeling/relpn/dpn.py:71) and
// `self.relness_pred(t)` (lib/modeling/relpn/dpn_anchor.py:105).
//
//   mode 0 (dense):       h_p[c,t] = a[ia[p]][c][t]
//   mode 1 (factorised):  h_p[c,t] = relu(a[ia[p]][c][t] + b[ib[p]][c][t] (+ bias[c]))
//   out[p][o][t] = bh[o] + sum_c Wh[o][c] * h_p[c][t]
//
// The activation h is the MFMA B operand and is produced directly in the B
// fragment layout (lane = (k = lane>>4, column = lane&15)), so the N^2 x T x C
// pair tensor never exists in memory — each wave streams the two tracklet
// projection rows it needs (L2 / Infinity-Cache resident) with 8-byte loads,
// adds, applies ReLU and feeds the matrix pipe.  One wave = NP consecutive pairs
// x 32 frames; the two 16-column MFMA blocks take the even / odd frames of the
// float2 a lane loads, so a half-wave row reads one full 128-B line.
#include "tspn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NP = 4;       // pairs per wave
constexpr int WAVES = 4;    // waves per workgroup
constexpr int TB = 32;      // frames per wave

template <int MODE, bool VEC2>
__global__ __launch_bounds__(WAVES * 64) void heads_kernel(
    const float* __restrict__ a, const float* __restrict__ b, int64_t lda,
    const int64_t* __restrict__ ia, const int64_t* __restrict__ ib, int64_t istride,
    const float* __restrict__ bias, const float* __restrict__ Wh, const float* __restrict__ bh,
    int H, int64_t P, int C, int T, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_tblk = (T + TB - 1) / TB;
  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t group = wave_global / n_tblk;
  const int tblk = (int)(wave_global - group * n_tblk);
  const int64_t p0 = group * NP;
  if (p0 >= P) return;  // no barriers in this kernel: whole waves may leave

  const int j = lane & 15, kq = lane >> 4;
  const int t0 = tblk * TB;
  // frame handled in MFMA block 0 / block 1
  int tA, tB;
  if (VEC2) {
    tA = t0 + 2 * j;
    tB = tA + 1;
  } else {
    tA = t0 + j;
    tB = t0 + 16 + j;
  }
  // clamped load positions (stores are masked separately)
  int lA, lB;
  if (VEC2) {
    lA = min(tA, T - 2);
    lB = lA + 1;
  } else {
    lA = min(tA, T - 1);
    lB = min(tB, T - 1);
  }

  const float* pa[NP];
  const float* pb[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int64_t p = min(p0 + q, P - 1);
    const int64_t ra = ia ? ia[p * istride] : p;
    pa[q] = a + ra * lda * (int64_t)T;
    if (MODE == 1) {
      const int64_t rb = ib ? ib[p * istride] : p;
      pb[q] = b + rb * lda * (int64_t)T;
    } else {
      pb[q] = nullptr;
    }
  }

  f32x4 acc[NP][2];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q][s][e] = 0.f;

  const int o_a = lane & 15;  // A-operand row (head output)
#pragma unroll 4
  for (int c0 = 0; c0 < C; c0 += 4) {
    const int c = c0 + kq;
    const bool cv = c < C;
    const int cl = cv ? c : C - 1;
    const float wa = (cv && o_a < H) ? Wh[(int64_t)o_a * C + cl] : 0.f;
    float bc = 0.f;
    if (MODE == 1 && bias != nullptr) bc = bias[cl];
    const int64_t roff = (int64_t)cl * T;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      float h0, h1;
      if (VEC2) {
        const float2 va = *reinterpret_cast<const float2*>(pa[q] + roff + lA);
        h0 = va.x;
        h1 = va.y;
        if (MODE == 1) {
          const float2 vb = *reinterpret_cast<const float2*>(pb[q] + roff + lA);
          h0 = fmaxf(h0 + vb.x + bc, 0.f);
          h1 = fmaxf(h1 + vb.y + bc, 0.f);
        }
      } else {
        h0 = pa[q][roff + lA];
        h1 = pa[q][roff + lB];
        if (MODE == 1) {
          h0 = fmaxf(h0 + pb[q][roff + lA] + bc, 0.f);
          h1 = fmaxf(h1 + pb[q][roff + lB] + bc, 0.f);
        }
      }
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, h0, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, h1, acc[q][1], 0, 0, 0);
    }
  }

  // C/D layout of the 16x16 MFMA: column = lane&15 (frame), row = (lane>>4)*4 + reg (head)
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int64_t p = p0 + q;
    if (p >= P) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = kq * 4 + e;
      if (o >= H) continue;
      const float bo = bh ? bh[o] : 0.f;
      float* dst = out + (p * H + o) * (int64_t)T;
      const float v0 = acc[q][0][e] + bo;
      const float v1 = acc[q][1][e] + bo;
      if (VEC2) {
        if (tA < T) *reinterpret_cast<float2*>(dst + tA) = make_float2(v0, v1);
      } else {
        if (tA < T) dst[tA] = v0;
        if (tB < T) dst[tB] = v1;
      }
    }
  }
}

}  // namespace

extern "C" int tspn_heads_f32(int mode, const float* a, const float* b, int64_t lda,
                              const int64_t* ia, const int64_t* ib, int64_t idx_stride,
                              const float* bias, const float* Wh, const float* bh, int64_t H, int64_t P, int64_t C,
                              int64_t T, float* out, void* stream) {
  TSPN_REQUIRE(mode == 0 || mode == 1, TSPN_EINVAL, "tspn_heads_f32: mode must be 0 or 1");
  TSPN_REQUIRE(H > 0 && H <= 16, TSPN_EUNSUPPORTED, "tspn_heads_f32: H=%lld not in [1,16]",
               (long long)H);
  TSPN_REQUIRE(P >= 0 && C > 0 && T > 0 && lda >= C && idx_stride >= 1, TSPN_EINVAL,
               "tspn_heads_f32: bad sizes P=%lld C=%lld T=%lld lda=%lld", (long long)P,
               (long long)C, (long long)T, (long long)lda);
  TSPN_REQUIRE(C < (1 << 24) && T < (1 << 24), TSPN_EUNSUPPORTED, "tspn_heads_f32: dim too large");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(a && Wh && out, TSPN_EINVAL, "tspn_heads_f32: null pointer");
  TSPN_REQUIRE(mode == 0 || b, TSPN_EINVAL, "tspn_heads_f32: mode 1 needs operand b");
  const bool vec2 = (T % 2 == 0) && T >= 2 &&
                    ((reinterpret_cast<uintptr_t>(a) & 7) == 0) &&
                    (mode == 0 || (reinterpret_cast<uintptr_t>(b) & 7) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  const int64_t n_tblk = tspn::ceil_div(T, TB);
  const int64_t nwaves = tspn::ceil_div(P, NP) * n_tblk;
  const int64_t nblocks = tspn::ceil_div(nwaves, WAVES);
  TSPN_REQUIRE(nblocks < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_heads_f32: grid too large");
  dim3 grid((unsigned)nblocks), block(WAVES * 64);
  hipStream_t s = TSPN_STREAM(stream);
#define TSPN_HEADS_LAUNCH(MODE, VEC)                                                        \
  hipLaunchKernelGGL((heads_kernel<MODE, VEC>), grid, block, 0, s, a, b, lda, ia, ib, idx_stride, \
                     bias, Wh, bh, (int)H, P, (int)C, (int)T, out)
  if (mode == 0) {
    if (vec2) TSPN_HEADS_LAUNCH(0, true); else TSPN_HEADS_LAUNCH(0, false);
  } else {
    if (vec2) TSPN_HEADS_LAUNCH(1, true); else TSPN_HEADS_LAUNCH(1, false);
  }
#undef TSPN_HEADS_LAUNCH
  return tspn::check_launch("tspn_heads_f32");
}
