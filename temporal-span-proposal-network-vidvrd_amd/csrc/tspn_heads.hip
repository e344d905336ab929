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
#include <cstdlib>
#include <type_traits>

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


// ---------------------------------------------------------------------------------------------
// Blocked pair stage for the canonical pair table (all ordered pairs (s,o), s != o, s-major, per
// video — reference lib/modeling/predict.py:133-140).  A workgroup owns an 8-subject x 8-object
// block of one video for 32 frames and walks the channels: the 16 tracklet-projection rows it
// needs are staged through LDS once per 16-channel chunk and reused by all 64 pairs, so HBM /
// Infinity-Cache traffic is ~6x lower than streaming two rows per pair (the v1 kernel above
// measured 8.6 GB of fetch per 8 videos, 7x the unique bytes; profiles/r1).  Output slots are
// computed (p = b*N*(N-1) + s*(N-1) + o - (o>s)), no index table is read.
// Wave w handles subjects {2w, 2w+1} x 8 objects; lane = (k = lane>>4, column = lane&15); the
// two 16-column MFMA blocks take the even / odd frames so LDS reads are conflict-free b64.
constexpr int PG_S = 8, PG_O = 8, PG_CK = 16, PG_T = 32;
constexpr int PG_ROWS = PG_S + PG_O;
constexpr int PG_STAGE = PG_ROWS * PG_CK * PG_T;  // floats per LDS buffer (32 KB)

template <bool VEC2>
__global__ __launch_bounds__(256, 2) void heads_pairgrid_kernel(
    const float* __restrict__ y, int C, int T, int N, const float* __restrict__ Wh,
    const float* __restrict__ bh, int H, float* __restrict__ out, int ntb, int nob, int nsb,
    int64_t ngroups) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* S = reinterpret_cast<float*>(smem_raw);  // [2][16 rows][16 ch][32 t]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  // Workgroup -> (video, frame block, subject block, object block).  The nsb*nob workgroups of
  // one (video, frame block) group read the same tracklet rows and walk the channels in step, so
  // they are placed on ONE XCD (ids congruent mod 8 share an XCD) and dispatched back to back:
  // each row chunk then comes from HBM once and from that XCD's L2 for the other blocks.
  const int per_group = nsb * nob;
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int64_t G = (int64_t)(k / per_group) * 8 + xcd;
  if (G >= ngroups) return;  // uniform per workgroup, before any barrier
  const int member = k % per_group;
  const int sb = member / nob, ob = member - sb * nob;
  const int64_t b = G / ntb;
  const int tb = (int)(G - b * ntb);
  const int t0 = tb * PG_T;
  const int64_t rowlen = 2 * (int64_t)C * T;  // floats per tracklet row of y

  // ---- staging: per chunk every thread moves 16 x 8 bytes: iteration `it` = tile row `it`
  // (0..7 subject U rows, 8..15 object V rows), channel = tid>>4, frames t0 + 2*(tid&15) (+1).
  const int st_ch = tid >> 4, st_t2 = tid & 15;
  int lt = t0 + 2 * st_t2;                 // clamped load frame (tile columns past T are never stored)
  if (VEC2) lt = min(lt, T - 2); else lt = min(lt, T - 1);
  const int lt1 = VEC2 ? lt + 1 : min(t0 + 2 * st_t2 + 1, T - 1);
  const float* src[PG_ROWS];
#pragma unroll
  for (int it = 0; it < PG_ROWS; ++it) {
    const int local = it < PG_S ? sb * PG_S + it : ob * PG_O + (it - PG_S);
    const int64_t trk = b * N + min(local, N - 1);
    src[it] = y + trk * rowlen + (it < PG_S ? 0 : (int64_t)C * T);
  }
  float2 st[PG_ROWS];
  float wreg[PG_CK / 4];
  const int o_a = lane & 15;
  auto load_chunk = [&](int c0) {
    const int c = min(c0 + st_ch, C - 1);
#pragma unroll
    for (int it = 0; it < PG_ROWS; ++it) {
      const float* p = src[it] + (int64_t)c * T;
      if (VEC2) {
        st[it] = *reinterpret_cast<const float2*>(p + lt);
      } else {
        st[it].x = p[lt];
        st[it].y = p[lt1];
      }
    }
#pragma unroll
    for (int ks = 0; ks < PG_CK / 4; ++ks) {
      const int cw = c0 + ks * 4 + kq;
      wreg[ks] = Wh[(int64_t)min(o_a, H - 1) * C + min(cw, C - 1)];
    }
  };
  auto store_chunk = [&](int buf) {
    float* dst = S + buf * PG_STAGE + st_ch * PG_T + 2 * st_t2;
#pragma unroll
    for (int it = 0; it < PG_ROWS; ++it)
      *reinterpret_cast<float2*>(dst + it * PG_CK * PG_T) = st[it];
  };

  f32x4 acc[2][PG_O][2];
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[si][oj][e][r] = 0.f;

  const int nchunks = (C + PG_CK - 1) / PG_CK;
  load_chunk(0);
  store_chunk(0);
  float wa[PG_CK / 4];
#pragma unroll
  for (int ks = 0; ks < PG_CK / 4; ++ks) wa[ks] = wreg[ks];
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const int c0 = c * PG_CK;
    if (c + 1 < nchunks) load_chunk(c0 + PG_CK);
    const float* Sb = S + buf * PG_STAGE + 2 * j;
#pragma unroll
    for (int ks = 0; ks < PG_CK / 4; ++ks) {
      const int ch = ks * 4 + kq;
      // A operand: head weights; rows >= H and channels >= C contribute zero
      const float w = (o_a < H && c0 + ch < C) ? wa[ks] : 0.f;
      float2 u[2], v[PG_O];
#pragma unroll
      for (int si = 0; si < 2; ++si)
        u[si] = *reinterpret_cast<const float2*>(Sb + ((2 * wave + si) * PG_CK + ch) * PG_T);
#pragma unroll
      for (int oj = 0; oj < PG_O; ++oj)
        v[oj] = *reinterpret_cast<const float2*>(Sb + ((PG_S + oj) * PG_CK + ch) * PG_T);
#pragma unroll
      for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int oj = 0; oj < PG_O; ++oj) {
          const float h0 = fmaxf(u[si].x + v[oj].x, 0.f);
          const float h1 = fmaxf(u[si].y + v[oj].y, 0.f);
          acc[si][oj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, h0, acc[si][oj][0], 0, 0, 0);
          acc[si][oj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, h1, acc[si][oj][1], 0, 0, 0);
        }
    }
    if (c + 1 < nchunks) {
      store_chunk(buf ^ 1);
#pragma unroll
      for (int ks = 0; ks < PG_CK / 4; ++ks) wa[ks] = wreg[ks];
    }
    __syncthreads();
  }

  // C/D layout of the 16x16 MFMA: column = lane&15 -> frames t0+2j (block 0) / t0+2j+1 (block 1),
  // row = (lane>>4)*4 + reg -> head output
  const int tA = t0 + 2 * j;
#pragma unroll
  for (int si = 0; si < 2; ++si) {
    const int s = sb * PG_S + 2 * wave + si;
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj) {
      const int o = ob * PG_O + oj;
      if (s >= N || o >= N || s == o) continue;
      const int64_t p = b * N * (int64_t)(N - 1) + (int64_t)s * (N - 1) + o - (o > s ? 1 : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ho = kq * 4 + r;
        if (ho >= H) continue;
        const float bo = bh ? bh[ho] : 0.f;
        float* dst = out + (p * H + ho) * (int64_t)T;
        const float v0 = acc[si][oj][0][r] + bo, v1 = acc[si][oj][1][r] + bo;
        if (VEC2) {
          if (tA < T) *reinterpret_cast<float2*>(dst + tA) = make_float2(v0, v1);
        } else {
          if (tA < T) dst[tA] = v0;
          if (tA + 1 < T) dst[tA + 1] = v1;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Third structure of the blocked pair stage (same tiling: 8 subjects x 8 objects x 32 frames per
// workgroup, 16-channel chunks).  Ablation of the kernel above (profiles/r1) put 0.55 ms of 2.55
// on register staging (16 loads + 16 ds_writes per lane and chunk) and 0.54 ms on the add+ReLU
// that hipcc chains through one temporary in front of every MFMA.  Here
//   * the projection rows are staged by LDS-DMA: rows of y are padded to a multiple of 4 frames
//     (ldt) so a 16-byte piece never leaves its row; a piece = 8 channels x 32 frames of one row;
//   * the activations of the NEXT (k-step, subject) phase are computed under the 16 MFMAs of the
//     current one (explicit software pipeline, issue pattern 2 VALU : 1 MFMA).
__device__ __forceinline__ void hglds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void heads_pairgrid3_kernel(
    const float* __restrict__ y, int64_t ldt, int C, int T, int N, const float* __restrict__ Wh,
    const float* __restrict__ bh, int H, float* __restrict__ out, int ntb, int nob, int nsb,
    int64_t ngroups) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* S = reinterpret_cast<float*>(smem_raw);  // [2][16 rows][16 ch][32 t]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int per_group = nsb * nob;
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int64_t G = (int64_t)(k / per_group) * 8 + xcd;
  if (G >= ngroups) return;
  const int member = k % per_group;
  const int sb = member / nob, ob = member - sb * nob;
  const int64_t b = G / ntb;
  const int tb = (int)(G - b * ntb);
  const int t0 = tb * PG_T;
  const int64_t rowlen = 2 * (int64_t)C * ldt;

  // ---- DMA sources: wave w stages tile rows 4w..4w+3; lane -> (channel line lane>>3, 4 frames).
  // The row base is wave-uniform (SGPRs, advanced by scalar adds); only the (line, frame group) offset of the
  // lane lives in a VGPR: fp32 MFMA and VALU do not overlap on gfx950 (profiles/r2/mfma_valu_overlap.txt), so the
  // 17 per-lane 64-bit pointer updates per chunk of the first version cost 1.7 % of the kernel.
  const float* srow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * wave + r;
    const int local = row < PG_S ? sb * PG_S + row : ob * PG_O + (row - PG_S);
    const int64_t trk = b * N + min(local, N - 1);
    srow[r] = y + trk * rowlen + (row < PG_S ? 0 : (int64_t)C * ldt);
  }
  // groups past the row end are never stored
  const unsigned loff = (unsigned)((lane >> 3) * ldt + min((int64_t)t0 + (lane & 7) * 4, ldt - 4));
  const int64_t half_step = 8 * ldt, chunk_step = 16 * ldt;
  auto stage_piece = [&](int buf, auto p_tag) {  // piece p = 2*r + hh of this wave
    constexpr int p = decltype(p_tag)::value;
    constexpr int r = p >> 1, hh = p & 1;
    hglds16(srow[r] + hh * half_step + loff, S + buf * PG_STAGE + ((4 * wave + r) * PG_CK + 8 * hh) * PG_T);
    if (hh == 1) srow[r] += chunk_step;
  };
  const int o_a = lane & 15;
  const float* wsrc = Wh + (int64_t)min(o_a, H - 1) * C + kq;
  float wreg[PG_CK / 4], wa[PG_CK / 4];
  auto load_w = [&](int c0) {
#pragma unroll
    for (int ks = 0; ks < PG_CK / 4; ++ks) wreg[ks] = wsrc[c0 + ks * 4];
  };

  f32x4 acc[2][PG_O][2];
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[si][oj][e][r] = 0.f;

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  using P6 = std::integral_constant<int, 6>;
  using P7 = std::integral_constant<int, 7>;

  const int nchunks = C / PG_CK;
  stage_piece(0, P0{}); stage_piece(0, P1{}); stage_piece(0, P2{}); stage_piece(0, P3{});
  stage_piece(0, P4{}); stage_piece(0, P5{}); stage_piece(0, P6{}); stage_piece(0, P7{});
  load_w(0);
#pragma unroll
  for (int ks = 0; ks < PG_CK / 4; ++ks) wa[ks] = (o_a < H) ? wreg[ks] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();

  struct UV {
    float2 u[2];
    float2 v[PG_O];
  };
  auto read_uv = [&](const float* Sb, int ks) {
    UV f;
    const int ch = ks * 4 + kq;
#pragma unroll
    for (int si = 0; si < 2; ++si)
      f.u[si] = *reinterpret_cast<const float2*>(Sb + ((2 * wave + si) * PG_CK + ch) * PG_T);
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj)
      f.v[oj] = *reinterpret_cast<const float2*>(Sb + ((PG_S + oj) * PG_CK + ch) * PG_T);
    return f;
  };
  struct HB {
    float h[PG_O][2];
  };
  auto compute_h = [&](const UV& f, int si) {
    HB r;
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj) {
      r.h[oj][0] = fmaxf(f.u[si].x + f.v[oj].x, 0.f);
      r.h[oj][1] = fmaxf(f.u[si].y + f.v[oj].y, 0.f);
    }
    return r;
  };

  auto chunk_body = [&](int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int buf = c & 1;
    const float* Sb = S + buf * PG_STAGE + 2 * j;
    if (MORE) load_w((c + 1) * PG_CK);
    UV uv = read_uv(Sb, 0);
    UV uvn = uv;
    HB hc = compute_h(uv, 0);
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) {
      const int ks = ph >> 1, si = ph & 1;
      if (si == 0 && ks < 3) uvn = read_uv(Sb, ks + 1);
      HB hn = hc;
      if (ph < 7) hn = (si == 0) ? compute_h(uv, 1) : compute_h(uvn, 0);
#pragma unroll
      for (int oj = 0; oj < PG_O; ++oj) {
        acc[si][oj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks], hc.h[oj][0], acc[si][oj][0], 0, 0, 0);
        acc[si][oj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks], hc.h[oj][1], acc[si][oj][1], 0, 0, 0);
      }
      if (MORE) {
        if (ph == 0) stage_piece(buf ^ 1, P0{});
        if (ph == 1) stage_piece(buf ^ 1, P1{});
        if (ph == 2) stage_piece(buf ^ 1, P2{});
        if (ph == 3) stage_piece(buf ^ 1, P3{});
        if (ph == 4) stage_piece(buf ^ 1, P4{});
        if (ph == 5) stage_piece(buf ^ 1, P5{});
        if (ph == 6) stage_piece(buf ^ 1, P6{});
        if (ph == 7) stage_piece(buf ^ 1, P7{});
      }
#define TSPN_G                                           \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
  __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G
      TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G TSPN_G
#undef TSPN_G
      __builtin_amdgcn_sched_barrier(0);
      hc = hn;
      if (si == 1) uv = uvn;
    }
    if (MORE) {
#pragma unroll
      for (int ks = 0; ks < PG_CK / 4; ++ks) wa[ks] = (o_a < H) ? wreg[ks] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
    __syncthreads();
  };
  for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c, std::true_type{});
  chunk_body(nchunks - 1, std::false_type{});

  const int tA = t0 + 2 * j;
#pragma unroll
  for (int si = 0; si < 2; ++si) {
    const int s = sb * PG_S + 2 * wave + si;
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj) {
      const int o = ob * PG_O + oj;
      if (s >= N || o >= N || s == o) continue;
      const int64_t p = b * N * (int64_t)(N - 1) + (int64_t)s * (N - 1) + o - (o > s ? 1 : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ho = kq * 4 + r;
        if (ho >= H) continue;
        const float bo = bh ? bh[ho] : 0.f;
        float* dst = out + (p * H + ho) * (int64_t)T;
        const float v0 = acc[si][oj][0][r] + bo, v1 = acc[si][oj][1][r] + bo;
        if (tA + 1 < T) {
          *reinterpret_cast<float2*>(dst + tA) = make_float2(v0, v1);
        } else if (tA < T) {
          dst[tA] = v0;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fourth structure of the blocked pair stage: NO matrix instruction.
// Measured on gfx950 (tools/probes/mfma_valu_overlap_probe.hip, profiles/r2/mfma_valu_overlap.txt): an fp32 MFMA
// runs at exactly the vector FMA rate (64 FLOP / clk / SIMD) and does not overlap with VALU instructions on its
// SIMD.  The head GEMM has only H = 12 output rows, so on v_mfma_f32_16x16x4_f32 a quarter of the rows is
// padding and the two VALU of relu(U[s] + V[o]) per activation come on top (kernel above: MFMA busy 67 %, VALU
// 20 %, 3.28 ms per 16 videos of config 2).  Here a lane owns one (subject, frame) column and keeps the H head
// sums of its eight objects in registers; the head weights are wave-uniform, so they travel in SGPRs and every
// multiply-add pair is one  v_pk_fma_f32 acc2, s_w, v_act2 : 2 + H / 2 VALU per activation, no padding, no fragment layout,
// and the sum over channels is one fmaf chain in channel order (deterministic, = a plain fp32 dot product).
// Same tiling and LDS staging as the kernel above (8 subjects x 8 objects x 32 frames per workgroup, 16-channel
// chunks by LDS-DMA, double-buffered); wave w = subjects {2w, 2w+1}, lane = (subject lane >> 5, frame lane & 31).
// Weights: packed once per call as Wp[chunk][16 ch][12 h] (pack_heads12_kernel), read by s_load_dwordx4 (inline
// asm: hipcc turns loads it cannot prove un-clobbered — the LDS-DMA builtin counts as a store — into per-lane
// VMEM loads, and SLP-vectorises the multiply-adds into v_pk_fma_f32 on VGPR copies of the weights).  SMEM data
// returns out of order, so every use is behind an `s_waitcnt lgkmcnt(0)` tied to the registers.
__global__ void pack_heads12_kernel(const float* __restrict__ Wh, int C, float* __restrict__ Wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;    // i = c * 12 + h
  if (i < C * 12) Wp[i] = Wh[(int64_t)(i % 12) * C + i / 12];
}

template <int OFF>
__device__ __forceinline__ void sload4(f32x4& d, const float* base) {
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(d) : "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void swait6(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f));
}
// (acc.x, acc.y) += w * (act.x, act.y): v_pk_fma_f32 with the weight broadcast from the low (even head) or the
// high (odd head) dword of an SGPR pair.  A plain v_fma_f32 issues once per 4 cycles and SIMD like every VALU
// instruction (measured: 4.3), i.e. at HALF the fp32 peak; only the packed form reaches 64 FLOP / clk / SIMD.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pkfma_lo(f32x2& acc, f32x2 w_sgpr, f32x2 act) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(w_sgpr), "v"(act));
}
__device__ __forceinline__ void pkfma_hi(f32x2& acc, f32x2 w_sgpr, f32x2 act) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w_sgpr), "v"(act));
}

// CK = channels per LDS stage (round 6): 16 (two workgroups per CU, as in rounds 2 - 5) or 8 -- 16-KB stages, THREE workgroups
// per CU: a third wave per SIMD to issue from while the other two sit at their chunk barriers (profiles/r6/pair_stage_counters.md:
// with two, the vector pipe issues in 78 % of the cycles).  Same sums in the same order either way.
template <int CK>
__global__ __launch_bounds__(256, CK == 8 ? 3 : 2) void heads_pairgrid4_kernel(
    const float* __restrict__ y, int64_t ldt, int C, int T, int N, const float* __restrict__ Wp,
    const float* __restrict__ bh, float* __restrict__ out, int ntb, int nob, int nsb, int64_t ngroups) {
  constexpr int H = 12;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* S = reinterpret_cast<float*>(smem_raw);  // [2][16 rows][CK ch][32 t]
  constexpr int STAGE = PG_ROWS * CK * PG_T;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int si = lane >> 5, tl = lane & 31;
  const int per_group = nsb * nob;
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int64_t G = (int64_t)(k / per_group) * 8 + xcd;
  if (G >= ngroups) return;
  const int member = k % per_group;
  const int sb = member / nob, ob = member - sb * nob;
  const int64_t b = G / ntb;
  const int tb = (int)(G - b * ntb);
  const int t0 = tb * PG_T;
  const int64_t rowlen = 2 * (int64_t)C * ldt;

  // ---- DMA sources (as in heads_pairgrid3_kernel): wave w stages tile rows 4w..4w+3.  Buffer loads (round 4): one SGPR
  // descriptor of this video's projections (the launcher checks they stay below 2 GB), a 32-bit lane offset per row, the
  // channel chunk / half as the scalar offset -- cheaper to issue than global_load_lds (tools/probes/lds_dma_issue_probe.hip)
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(y) + b * N * rowlen, 0, (int)(unsigned)((int64_t)N * rowlen * 4), 0x00020000);
  unsigned voff[4];
  const unsigned loff = (unsigned)((lane >> 3) * ldt + min((int64_t)t0 + (lane & 7) * 4, ldt - 4));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * wave + r;
    const int local = row < PG_S ? sb * PG_S + row : ob * PG_O + (row - PG_S);
    const int64_t trk = min(local, N - 1);
    voff[r] = (unsigned)((trk * rowlen + (row < PG_S ? 0 : (int64_t)C * ldt) + loff) * 4);
  }
  const int half_bytes = (int)(8 * ldt * 4);
  int y_soff = 0;                                    // byte offset of the next chunk's channels
  auto stage_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int hh = 0; hh < CK / 8; ++hh)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc_y, (__attribute__((address_space(3))) void*)(S + buf * STAGE + ((4 * wave + r) * CK + 8 * hh) * PG_T), 16,
            (int)voff[r], y_soff + hh * half_bytes, 0, 0);
    }
    y_soff += (CK / 8) * half_bytes;
  };

  f32x2 acc[PG_O / 2][H];     // (object 2 op, object 2 op + 1) x head
#pragma unroll
  for (int op = 0; op < PG_O / 2; ++op)
#pragma unroll
    for (int h = 0; h < H; ++h) acc[op][h] = f32x2{0.f, 0.f};

  // A step = two channels: 2 x 9 LDS values per lane and 2 x 12 weights (six SGPR quads), fetched while the
  // 2 x 8 x (2 + 12) VALU of the previous step run.
  struct Step {
    float u[2];
    float v[PG_O][2];
  };
  auto fetch_uv = [&](Step& st, const float* ub, const float* vb, int kk) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ch = 2 * kk + e;
      st.u[e] = ub[ch * PG_T];
#pragma unroll
      for (int oj = 0; oj < PG_O; ++oj) st.v[oj][e] = vb[(oj * CK + ch) * PG_T];
    }
  };
  auto compute = [&](const Step& st, const f32x4 (&w)[6]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int op = 0; op < PG_O / 2; ++op) {
        const f32x2 a{fmaxf(st.u[e] + st.v[2 * op][e], 0.f), fmaxf(st.u[e] + st.v[2 * op + 1][e], 0.f)};
#pragma unroll
        for (int h = 0; h < H; h += 2) {
          const f32x4& wq = w[3 * e + (h >> 2)];
          const f32x2 wp = (h & 2) ? f32x2{wq[2], wq[3]} : f32x2{wq[0], wq[1]};
          pkfma_lo(acc[op][h], wp, a);
          pkfma_hi(acc[op][h + 1], wp, a);
        }
      }
    }
  };
#define TSPN_WLOAD(W, KK, BASE)                                                                        \
  sload4<(2 * (KK)) * 48>(W[0], BASE); sload4<(2 * (KK)) * 48 + 16>(W[1], BASE);                       \
  sload4<(2 * (KK)) * 48 + 32>(W[2], BASE); sload4<(2 * (KK) + 1) * 48>(W[3], BASE);                   \
  sload4<(2 * (KK) + 1) * 48 + 16>(W[4], BASE); sload4<(2 * (KK) + 1) * 48 + 32>(W[5], BASE);
#define TSPN_WWAIT(W) swait6(W[0], W[1], W[2], W[3], W[4], W[5]);

  const int nchunks = C / CK;
  stage_chunk(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();
  const float* urow = S + ((2 * wave + si) * CK) * PG_T + tl;   // + buffer + channel * 32
  const float* vrow = S + (PG_S * CK) * PG_T + tl;              // + buffer + (object * 16 + channel) * 32
  const float* wbase = Wp;                                         // wave-uniform: weights of the current chunk
  Step s0, s1;
  f32x4 w0[6], w1[6];
  fetch_uv(s0, urow, vrow, 0);
  TSPN_WLOAD(w0, 0, wbase)
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) stage_chunk(buf ^ 1);
    const float* ub = urow + buf * STAGE;
    const float* vb = vrow + buf * STAGE;
    __builtin_amdgcn_sched_barrier(0);
#define TSPN_STEP2(KK)                                           \
    fetch_uv(s1, ub, vb, (KK) + 1);                              \
    TSPN_WLOAD(w1, (KK) + 1, wbase)                              \
    __builtin_amdgcn_sched_barrier(0);                           \
    compute(s0, w0);                                             \
    __builtin_amdgcn_sched_barrier(0);                           \
    TSPN_WWAIT(w1)                                               \
    if ((KK) + 2 < CK / 2) {                                     \
      fetch_uv(s0, ub, vb, (KK) + 2);                            \
      TSPN_WLOAD(w0, ((KK) + 2) & 7, wbase)                      \
    }                                                            \
    __builtin_amdgcn_sched_barrier(0);                           \
    compute(s1, w1);                                             \
    __builtin_amdgcn_sched_barrier(0);                           \
    if ((KK) + 2 < CK / 2) { TSPN_WWAIT(w0) }
    TSPN_WWAIT(w0)
    TSPN_STEP2(0) TSPN_STEP2(2)
    if constexpr (CK == 16) { TSPN_STEP2(4) TSPN_STEP2(6) }
#undef TSPN_STEP2
    wbase += CK * H;
    if (c + 1 < nchunks) { TSPN_WLOAD(w0, 0, wbase) }   // weights do not depend on the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
    __syncthreads();
    if (c + 1 < nchunks)      // first step of the next chunk: its tile landed before the barrier
      fetch_uv(s0, urow + (buf ^ 1) * STAGE, vrow + (buf ^ 1) * STAGE, 0);
  }
#undef TSPN_WLOAD
#undef TSPN_WWAIT

  const int t = t0 + tl;
  const int s = sb * PG_S + 2 * wave + si;
  if (t < T && s < N) {
#pragma unroll
    for (int oj = 0; oj < PG_O; ++oj) {
      const int o = ob * PG_O + oj;
      if (o >= N || s == o) continue;
      const int64_t p = b * N * (int64_t)(N - 1) + (int64_t)s * (N - 1) + o - (o > s ? 1 : 0);
#pragma unroll
      for (int h = 0; h < H; ++h)
        out[(p * H + h) * (int64_t)T + t] = acc[oj >> 1][h][oj & 1] + (bh ? bh[h] : 0.f);
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


extern "C" int tspn_heads_pairgrid_f32(const float* y, int64_t B, int64_t N, int64_t C, int64_t T,
                                       const float* Wh, const float* bh, int64_t H, float* out,
                                       void* stream) {
  return tspn::heads_pairgrid(y, T, B, N, C, T, Wh, bh, H, out, stream);
}

// y[B*N][2C][ldt] (ldt >= T frames per row)
// `Wp12` (optional, C * 12 floats of scratch): with it and H == 12 the scalar-weight VALU kernel runs
int tspn::heads_pairgrid(const float* y, int64_t ldt, int64_t B, int64_t N, int64_t C, int64_t T,
                         const float* Wh, const float* bh, int64_t H, float* out, void* stream, float* Wp12) {
  TSPN_REQUIRE(B >= 0 && N >= 0 && C > 0 && T > 0 && ldt >= T, TSPN_EINVAL,
               "tspn_heads_pairgrid_f32: bad sizes B=%lld N=%lld C=%lld T=%lld ldt=%lld", (long long)B,
               (long long)N, (long long)C, (long long)T, (long long)ldt);
  TSPN_REQUIRE(H > 0 && H <= 16, TSPN_EUNSUPPORTED, "tspn_heads_pairgrid_f32: H=%lld not in [1,16]",
               (long long)H);
  TSPN_REQUIRE(C < (1 << 24) && ldt < (1 << 24) && N < (1 << 15), TSPN_EUNSUPPORTED,
               "tspn_heads_pairgrid_f32: dim too large");
  if (B == 0 || N < 2) return TSPN_OK;
  TSPN_REQUIRE(y && Wh && out, TSPN_EINVAL, "tspn_heads_pairgrid_f32: null pointer");
  const int ntb = (int)tspn::ceil_div(T, PG_T);
  const int nsb = (int)tspn::ceil_div(N, PG_S), nob = (int)tspn::ceil_div(N, PG_O);
  const size_t smem = sizeof(float) * 2 * PG_STAGE;
  const int64_t ngroups = B * ntb;
  const int64_t nwg = tspn::ceil_div(ngroups, 8) * 8 * nsb * nob;
  TSPN_REQUIRE(nwg < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_heads_pairgrid_f32: grid too large");
  // v3 (LDS-DMA staging + pipelined activations) needs 16-byte pieces inside a row and even frame
  // pairs in the output; anything else runs the register-staged kernel
  const bool v3 = (ldt % 4 == 0) && ldt >= 4 && (C % PG_CK == 0) && (T % 2 == 0) &&
                  ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  const bool vec2 = (T % 2 == 0) && (ldt % 2 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  TSPN_REQUIRE(v3 || ldt == T, TSPN_EUNSUPPORTED,
               "tspn_heads_pairgrid_f32: padded rows (ldt != T) need the DMA kernel's preconditions");
  // (its operand pieces are buffer loads with 32-bit offsets from the video's first row: a video's projections below 2 GB)
  const bool fits32 = N * 2 * C * ldt * 4 < (1LL << 31);
  if (v3 && H == 12 && Wp12 != nullptr && fits32) {   // scalar-weight VALU form (no 12 -> 16 row padding, no MFMA / VALU serialisation)
    hipLaunchKernelGGL(pack_heads12_kernel, dim3((unsigned)tspn::ceil_div(C * 12, 256)), dim3(256), 0,
                       TSPN_STREAM(stream), Wh, (int)C, Wp12);
#ifndef TSPN_PAIRGRID4_CK
#define TSPN_PAIRGRID4_CK 8      /* channels per LDS stage of heads_pairgrid4_kernel: 8 (three workgroups per CU) or 16 (two) */
#endif
    constexpr int CK4 = TSPN_PAIRGRID4_CK;
    static_assert(CK4 == 8 || CK4 == 16, "heads_pairgrid4_kernel is built for 8- or 16-channel stages");
    const size_t smem4 = sizeof(float) * 2 * PG_ROWS * CK4 * PG_T;
    static tspn::LdsLimit lds4;
    if (int rc = lds4.ensure(reinterpret_cast<const void*>(heads_pairgrid4_kernel<CK4>), smem4, "tspn_heads_pairgrid_f32"))
      return rc;
    hipLaunchKernelGGL(heads_pairgrid4_kernel<CK4>, dim3((unsigned)nwg), dim3(256), smem4, TSPN_STREAM(stream), y, ldt,
                       (int)C, (int)T, (int)N, Wp12, bh, out, ntb, nob, nsb, ngroups);
    return tspn::check_launch("tspn_heads_pairgrid_f32");
  }
  const void* fn = v3 ? reinterpret_cast<const void*>(heads_pairgrid3_kernel)
                      : (vec2 ? reinterpret_cast<const void*>(heads_pairgrid_kernel<true>)
                              : reinterpret_cast<const void*>(heads_pairgrid_kernel<false>));
  const int which = v3 ? 2 : (vec2 ? 1 : 0);
  static tspn::LdsLimit lds[3];
  if (int rc = lds[which].ensure(fn, smem, "tspn_heads_pairgrid_f32")) return rc;
  hipStream_t st = TSPN_STREAM(stream);
  if (v3) {
    hipLaunchKernelGGL(heads_pairgrid3_kernel, dim3((unsigned)nwg), dim3(256), smem, st, y, ldt, (int)C,
                       (int)T, (int)N, Wh, bh, (int)H, out, ntb, nob, nsb, ngroups);
  } else if (vec2) {
    hipLaunchKernelGGL(heads_pairgrid_kernel<true>, dim3((unsigned)nwg), dim3(256), smem, st, y, (int)C,
                       (int)T, (int)N, Wh, bh, (int)H, out, ntb, nob, nsb, ngroups);
  } else {
    hipLaunchKernelGGL(heads_pairgrid_kernel<false>, dim3((unsigned)nwg), dim3(256), smem, st, y,
                       (int)C, (int)T, (int)N, Wh, bh, (int)H, out, ntb, nob, nsb, ngroups);
  }
  return tspn::check_launch("tspn_heads_pairgrid_f32");
}
