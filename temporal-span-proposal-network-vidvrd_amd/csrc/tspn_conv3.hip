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
// staging through two register sets, loads issued two chunks ahead.
// fp32 MFMA runs at 64 FLOP/clk/SIMD (= 157 TF/s chip peak): one LDS dword per
// operand per 4096 FLOP, so LDS/L2 bandwidth is far from limiting; the design
// goal is simply to keep the matrix pipe issuing back-to-back.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

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
    int relu, int ldy) {
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

  // ---- per-thread staging descriptors (constant over the K loop).  Source addresses are
  // clamped into range and the loaded value is masked, so staging is branch-free.
  // B main: slot = 1 + (tid & 127) <-> column n0 + (tid & 127); rows (tid>>7)*8 .. +7
  const int bslot = 1 + (tid & 127);
  const int brow0 = (tid >> 7) * 8;
  bool bvalid;
  const float* bptr;
  {
    const int64_t n = n0 + (tid & 127);
    bvalid = n < ncols;
    const int64_t nc = bvalid ? n : ncols - 1;
    const int64_t b = nc / T;
    bptr = x + (b * Cin) * (int64_t)T + (nc - b * T);
  }
  // B halo: threads 0..31: row = tid & 15, side = tid >> 4 (0: column n0-1, 1: column n0+BN)
  const int hrow = tid & 15;
  const int hslot = (tid >> 4) & 1 ? BN + 1 : 0;
  bool hvalid;
  const float* hptr;
  {
    const int64_t n = ((tid >> 4) & 1) ? n0 + BN : n0 - 1;
    hvalid = tid < 32 && n >= 0 && n < ncols;
    const int64_t nc = hvalid ? n : 0;
    const int64_t b = nc / T;
    hptr = x + (b * Cin) * (int64_t)T + (nc - b * T);
  }
  // A: 6 float4 per thread; row = tap*KC + ci, 4 consecutive m
  const int am = (tid & 31) * 4;
  const bool avalid = m0 + am < M;  // M % 4 == 0 on the VEC_A path
  const int amc = avalid ? m0 + am : 0;

  // Two staging register sets: chunk j travels global -> set (j&1) -> LDS buffer (j&1); its loads
  // are issued two chunks (~5 us of MFMA work) before the ds_write that consumes them, which is
  // what hides L2-miss latency (one chunk of distance left the matrix pipe ~20 % idle).
  float4 a_reg[2][6];
  float b_reg[2][8];
  float h_reg[2] = {0.f, 0.f};

  // Loads are unconditional (clamped addresses) and their results are not touched until the
  // ds_write two chunks later: any use here would make the wave wait for the load at once.
  auto load_chunk = [&](auto set_tag, int c0) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int row = (tid >> 5) + r * 8;  // 0..47 = tap*KC + ci
      const int tap = row / KC, ci = c0 + row - tap * KC;
      const int cic = ci < Cin ? ci : Cin - 1;
      const float* src = Wp + ((int64_t)tap * Cin + cic) * M + amc;
      if (VEC_A) {
        a_reg[S][r] = *reinterpret_cast<const float4*>(src);
      } else {
        a_reg[S][r].x = src[0];
        a_reg[S][r].y = src[m0 + am + 1 < M ? 1 : 0];
        a_reg[S][r].z = src[m0 + am + 2 < M ? 2 : 0];
        a_reg[S][r].w = src[m0 + am + 3 < M ? 3 : 0];
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ci = c0 + brow0 + r;
      b_reg[S][r] = bptr[(int64_t)(ci < Cin ? ci : Cin - 1) * T];
    }
    {
      const int ci = c0 + hrow;
      h_reg[S] = hptr[(int64_t)(ci < Cin ? ci : Cin - 1) * T];
    }
  };

  // Masking happens on the way into LDS.  Channels past Cin are zeroed on the weight side only
  // (a zero A row kills the product); columns past the tensor are zeroed on the x side.
  auto store_chunk = [&](auto set_tag, int c0) {  // set S -> LDS buffer S
    constexpr int S = decltype(set_tag)::value;
    float* Ab = As + S * A_STAGE;
    float* Bb = Bs + S * B_STAGE;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int row = (tid >> 5) + r * 8;
      const int tap = row / KC, ci = c0 + row - tap * KC;
      const bool ok = avalid && ci < Cin;
      float4 v = a_reg[S][r];
      if (!VEC_A) {
        v.y = m0 + am + 1 < M ? v.y : 0.f;
        v.z = m0 + am + 2 < M ? v.z : 0.f;
        v.w = m0 + am + 3 < M ? v.w : 0.f;
      }
      v.x = ok ? v.x : 0.f;
      v.y = ok ? v.y : 0.f;
      v.z = ok ? v.z : 0.f;
      v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(Ab + (tid + r * THREADS) * 4) = v;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) Bb[(brow0 + r) * BNP + bslot] = bvalid ? b_reg[S][r] : 0.f;
    if (tid < 32) Bb[hrow * BNP + hslot] = hvalid ? h_reg[S] : 0.f;
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

  // One k-step = channels (2kk, 2kk+1) x 3 taps: 6 A values + 6 B values feed 12 MFMAs.
  struct Frag {
    float a[3][2];
    float b[2][3];
  };
  auto read_frag = [&](const float* Ab, const float* Bb, int kk) {
    Frag f;
    const int k = 2 * kk + kh;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      f.a[tap][0] = Ab[(tap * KC + k) * BM];
      f.a[tap][1] = Ab[(tap * KC + k) * BM + 32];
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 3; ++j) f.b[ni][j] = Bb[k * BNP + ni * 32 + j];
    return f;
  };

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  const int nchunks = (Cin + KC - 1) / KC;
  load_chunk(Set0{}, 0);
  if (nchunks > 1) load_chunk(Set1{}, KC);
  store_chunk(Set0{}, 0);
  if (nchunks > 2) load_chunk(Set0{}, 2 * KC);
  __syncthreads();

  constexpr int KSTEPS = KC / 2;
  constexpr int STORE_AT = 1;  // k-step after which chunk c+1 is staged into LDS and c+3 requested
  // Chunk c is computed from LDS buffer (c&1); set/buffer NEXT = (c+1)&1 holds chunk c+1.
  auto chunk_body = [&](auto next_tag, int c) {
    constexpr int NEXT = decltype(next_tag)::value;
    constexpr int BUF = NEXT ^ 1;
    const float* Ab = As + BUF * A_STAGE + wm * 64 + li;
    const float* Bb = Bs + BUF * B_STAGE + wn * 64 + li;
    Frag cur = read_frag(Ab, Bb, 0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      Frag nxt = cur;
      if (kk + 1 < KSTEPS) nxt = read_frag(Ab, Bb, kk + 1);
      // Pin the order: the next k-step's LDS reads are issued BEFORE this step's 12 MFMAs, so
      // their latency is covered by 768 cycles of matrix work.
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        float b0 = cur.b[0][tap], b1 = cur.b[1][tap];
        if (tap == 0) {
          b0 = mask_l[0] ? b0 : 0.f;
          b1 = mask_l[1] ? b1 : 0.f;
        } else if (tap == 2) {
          b0 = mask_r[0] ? b0 : 0.f;
          b1 = mask_r[1] ? b1 : 0.f;
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0], b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0], b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1], b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1], b1, acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk == STORE_AT) {
        // LDS buffer NEXT has been free since the barrier that ended chunk c-1.
        if (c + 1 < nchunks) store_chunk(next_tag, (c + 1) * KC);
        if (c + 3 < nchunks) load_chunk(next_tag, (c + 3) * KC);
      }
      cur = nxt;
    }
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    chunk_body(Set1{}, c);
    if (c + 1 < nchunks) chunk_body(Set0{}, c + 1);
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
    float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          float v = acc[mi][ni][e];
          if (bias != nullptr) v += bias[m];
          if (relu) v = fmaxf(v, 0.f);
          ycol[(int64_t)m * ldy] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Fast path: the same tiling with the operand tiles moved HBM/L2 -> LDS by LDS-DMA
// (global_load_lds): no staging registers, no ds_write pass, a dozen VMEM instructions per wave
// and chunk.  Ablation on MI355X (profiles/r1/conv3_ablation.md) showed the register-staged
// kernel losing ~15 % of the matrix pipe to the *instructions* of its staging pass (address
// arithmetic + 15 loads + 15 ds_writes + masks per lane and chunk), not to memory latency.
// Requires Cin % KC == 0, M % 4 == 0 and 16-B aligned weights; edge tiles need no masking in
// flight: out-of-range rows / columns read clamped addresses and are never stored, and the only
// valid column they could leak into (the last one, through its +1 tap) is already masked as a
// sequence end.
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int KCD>
__global__ __launch_bounds__(THREADS, (KCD == 8 ? 4 : 2)) void conv3_mfma_dma_kernel(
    const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n,
    int relu, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int A_ST = 3 * KCD * BM;  // floats per A buffer
  constexpr int B_ST = KCD * BNP;     // floats per B buffer
  constexpr int RA = 3 * KCD / 4;     // A rows (tap, ci) per wave and chunk
  constexpr int RB = KCD / 4;         // x rows per wave and chunk
  float* As = reinterpret_cast<float*>(smem_raw);  // [2][3][KCD][BM]
  float* Bs = As + 2 * A_ST;                        // [2][KCD][BNP]

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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- DMA source pointers (advance by one chunk per iteration)
  // A: wave w fills LDS rows [RA*w, RA*w+RA) of the chunk's 3*KCD (tap, ci) rows, two rows
  // (1 KiB) per instruction: lane -> row RA*w + 2i + (lane>>5), 4 consecutive m at (lane&31)*4.
  const float* asrc[RA / 2];
  {
    const int am = (lane & 31) * 4;
    const int amc = m0 + am < M ? m0 + am : 0;
#pragma unroll
    for (int i = 0; i < RA / 2; ++i) {
      const int row = wave * RA + 2 * i + (lane >> 5);
      asrc[i] = Wp + ((int64_t)(row / KCD) * Cin + (row % KCD)) * M + amc;
    }
  }
  // B: wave w fills rows [RB*w, RB*w+RB); one instruction = 64 consecutive columns of one row.
  const float* bsrc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t n = n0 + 64 * h + lane;
    const int64_t nc = n < ncols ? n : ncols - 1;
    const int64_t b = nc / T;
    bsrc[h] = x + (b * Cin + wave * RB) * (int64_t)T + (nc - b * T);
  }
  // halo columns n0-1 and n0+BN: 2*KCD lanes, ordinary load + ds_write
  const int hrow = tid % KCD;
  const int hside = (tid / KCD) & 1;
  const int hslot = hside ? BN + 1 : 0;
  bool hvalid;
  const float* hptr;
  {
    const int64_t n = hside ? n0 + BN : n0 - 1;
    hvalid = tid < 2 * KCD && n >= 0 && n < ncols;
    const int64_t nc = hvalid ? n : 0;
    const int64_t b = nc / T;
    hptr = x + (b * Cin + hrow) * (int64_t)T + (nc - b * T);
  }
  const int64_t a_step = (int64_t)KCD * M;
  const int64_t b_step = (int64_t)KCD * T;

  // One DMA instruction of the chunk's NDMA (= RA/2 weight pieces + 2*RB x pieces).  They are
  // issued one at a time between MFMA groups: a VMEM instruction placed right after an MFMA
  // issues while that MFMA occupies the matrix pipe, whereas a burst of them at the head of the
  // chunk kept the wave off the pipe for ~7 % of the time (profiles/r1/conv3_ablation.md).
  constexpr int NDMA = RA / 2 + 2 * RB;
  auto stage_one = [&](int buf, auto d_tag) {
    constexpr int d = decltype(d_tag)::value;
    if constexpr (d < RA / 2) {
      glds16(asrc[d], As + buf * A_ST + (wave * RA + 2 * d) * BM);
      asrc[d] += a_step;
    } else if constexpr (d < NDMA) {
      constexpr int r = (d - RA / 2) >> 1, h = (d - RA / 2) & 1;
      glds4(bsrc[h] + (int64_t)r * T, Bs + buf * B_ST + (wave * RB + r) * BNP + 1 + 64 * h);
      if constexpr (d == NDMA - 1) {
        bsrc[0] += b_step;
        bsrc[1] += b_step;
      }
    }
  };
  auto stage_all = [&](int buf) {
    stage_one(buf, std::integral_constant<int, 0>{});
    stage_one(buf, std::integral_constant<int, 1>{});
    stage_one(buf, std::integral_constant<int, 2>{});
    stage_one(buf, std::integral_constant<int, 3>{});
    stage_one(buf, std::integral_constant<int, 4>{});
    stage_one(buf, std::integral_constant<int, 5>{});
    stage_one(buf, std::integral_constant<int, 6>{});
    stage_one(buf, std::integral_constant<int, 7>{});
    stage_one(buf, std::integral_constant<int, 8>{});
    stage_one(buf, std::integral_constant<int, 9>{});
    stage_one(buf, std::integral_constant<int, 10>{});
    stage_one(buf, std::integral_constant<int, 11>{});
    stage_one(buf, std::integral_constant<int, 12>{});
    stage_one(buf, std::integral_constant<int, 13>{});
  };

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

  struct Frag {
    float a[3][2];
    float b[2][3];
  };
  auto read_frag = [&](const float* Ab, const float* Bb, int kk) {
    Frag f;
    const int k = 2 * kk + kh;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      f.a[tap][0] = Ab[(tap * KCD + k) * BM];
      f.a[tap][1] = Ab[(tap * KCD + k) * BM + 32];
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 3; ++j) f.b[ni][j] = Bb[k * BNP + ni * 32 + j];
    return f;
  };

  const int nchunks = Cin / KCD;
  stage_all(0);
  float h_reg = *hptr;
  hptr += b_step;
  if (tid < 2 * KCD) Bs[hrow * BNP + hslot] = hvalid ? h_reg : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();

  constexpr int KSTEPS = KCD / 2;
  // The chunk body is instantiated twice: with staging of the next chunk (all chunks but the
  // last) and without (the last) — a run-time `if` around each DMA would put a branch between
  // every MFMA group.
  auto chunk_body = [&](int buf, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    if (MORE) {  // (buffer buf^1 has been free since the barrier that ended the previous chunk)
      h_reg = *hptr;
      hptr += b_step;
    }
    const float* Ab = As + buf * A_ST + wm * 64 + li;
    const float* Bb = Bs + buf * B_ST + wn * 64 + li;
    Frag cur = read_frag(Ab, Bb, 0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      Frag nxt = cur;
      if (kk + 1 < KSTEPS) nxt = read_frag(Ab, Bb, kk + 1);
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        float b0 = cur.b[0][tap], b1 = cur.b[1][tap];
        if (tap == 0) {
          b0 = mask_l[0] ? b0 : 0.f;
          b1 = mask_l[1] ? b1 : 0.f;
        } else if (tap == 2) {
          b0 = mask_r[0] ? b0 : 0.f;
          b1 = mask_r[1] ? b1 : 0.f;
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0], b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0], b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1], b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1], b1, acc[1][1], 0, 0, 0);
        if (MORE) {
          if (kk == 0 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 0>{});
          if (kk == 0 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 1>{});
          if (kk == 0 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 2>{});
          if (kk == 1 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 3>{});
          if (kk == 1 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 4>{});
          if (kk == 1 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 5>{});
          if (kk == 2 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 6>{});
          if (kk == 2 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 7>{});
          if (kk == 2 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 8>{});
          if (kk == 3 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 9>{});
          if (kk == 3 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 10>{});
          if (kk == 3 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 11>{});
          if (KSTEPS > 4 && kk == 4 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 12>{});
          if (KSTEPS > 4 && kk == 4 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 13>{});
        }
      }
      // Issue pattern for the k-step (measured +2.5 % over issuing the reads as one burst ahead of
      // the MFMAs): after each MFMA one LDS read of the NEXT k-step, one VALU (mask / address)
      // and, every fourth group, one LDS-DMA piece issue in the shadow of that MFMA's 64 cycles.
#define TSPN_G(NV, NVM)                                 \
  __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x020, NVM, 0);
      TSPN_G(4, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1)
      TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1)
      TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1)
#undef TSPN_G
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    if (MORE && tid < 2 * KCD) Bs[(buf ^ 1) * B_ST + hrow * BNP + hslot] = hvalid ? h_reg : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
    __syncthreads();
  };
  for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c & 1, std::true_type{});
  chunk_body((nchunks - 1) & 1, std::false_type{});

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    if (n >= ncols) continue;
    const int64_t b = n / T;
    const int64_t t = n - b * T;
    float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          float v = acc[mi][ni][e];
          if (bias != nullptr) v += bias[m];
          if (relu) v = fmaxf(v, 0.f);
          ycol[(int64_t)m * ldy] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Channels-last variant: x is [B, T, Cin] — the tracklet layout itself — so the fused path needs
// no transpose pass, a column's channels are contiguous (16-B aligned DMA pieces: 9 per chunk and
// workgroup for the x tile instead of 32 four-byte ones), the halo columns are ordinary units of
// the same DMA, and flat column n is simply row n of x.  LDS image of the x tile:
// [4 channel groups][132 column slots][4 channels]; a lane reads the two channels it needs for two
// consecutive MFMA k-steps with one ds_read_b64 (lanes k=0: channels 4g,4g+1; k=1: 4g+2,4g+3 — the
// weight fragment uses the same channel pairing, any pairing is a valid K order).
constexpr int CL_KC = 16;
constexpr int CL_NG = CL_KC / 4;
constexpr int CL_SLP = 132;                        // padded column slots per channel group (130 used)
constexpr int CL_A_ST = 3 * CL_KC * BM;            // floats
constexpr int CL_B_ST = CL_NG * CL_SLP * 4;        // floats
constexpr int CL_UNITS = CL_NG * CL_SLP;           // 16-B units per x tile (incl. padding)
constexpr size_t CL_SMEM_BYTES = sizeof(float) * 2 * (CL_A_ST + CL_B_ST);

__global__ __launch_bounds__(THREADS, 2) void conv3_mfma_cl_kernel(
    const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n,
    int relu, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);  // [2][3][16][BM]
  float* Bs = As + 2 * CL_A_ST;                     // [2][4][132][4]

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- DMA sources.  A: as in the channels-first kernel (6 pieces of 2 rows x 128 m per wave).
  const float* asrc[6];
  {
    const int am = (lane & 31) * 4;
    const int amc = m0 + am < M ? m0 + am : 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = wave * 12 + 2 * i + (lane >> 5);
      asrc[i] = Wp + ((int64_t)(row >> 4) * Cin + (row & 15)) * M + amc;
    }
  }
  // x: piece p = wave + 4q covers units [64p, 64p+64); unit u = (group g = u/132, slot = u%132),
  // slot <-> column n0 + slot - 1 (clamped; clamped columns are never stored).
  const float* bsrc[3];
  bool bval[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int u = 64 * (wave + 4 * q) + lane;
    const int g = u / CL_SLP, slot = u - g * CL_SLP;
    bval[q] = u < CL_UNITS && slot < BN + 2;
    int64_t n = n0 + slot - 1;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bsrc[q] = x + n * Cin + 4 * (g < CL_NG ? g : 0);
  }
  const int64_t a_step = (int64_t)CL_KC * M;

  constexpr int NDMA = 6 + 3;
  auto stage_one = [&](int buf, auto d_tag) {
    constexpr int d = decltype(d_tag)::value;
    if constexpr (d < 6) {
      glds16(asrc[d], As + buf * CL_A_ST + (wave * 12 + 2 * d) * BM);
      asrc[d] += a_step;
    } else if constexpr (d < NDMA) {
      constexpr int q = d - 6;
      if (bval[q]) glds16(bsrc[q], Bs + buf * CL_B_ST + 64 * (wave + 4 * q) * 4);
      bsrc[q] += CL_KC;
    }
  };

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

  // fragments of one channel group (4 channels = 2 MFMA k-steps x 3 taps = 24 MFMAs)
  struct Frag {
    float a[3][2][2];  // [tap][mi][row of the lane's channel pair]
    float2 b[2][3];    // [ni][tap]
  };
  auto read_frag = [&](const float* Ab, const float* Bb, int g) {
    Frag f;
    const int r0 = 4 * g + 2 * kh;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f.a[tap][mi][0] = Ab[(tap * CL_KC + r0) * BM + mi * 32];
        f.a[tap][mi][1] = Ab[(tap * CL_KC + r0 + 1) * BM + mi * 32];
      }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int tap = 0; tap < 3; ++tap)
        f.b[ni][tap] = *reinterpret_cast<const float2*>(Bb + (g * CL_SLP + ni * 32 + tap) * 4);
    return f;
  };

  const int nchunks = Cin / CL_KC;
  stage_one(0, std::integral_constant<int, 0>{});
  stage_one(0, std::integral_constant<int, 1>{});
  stage_one(0, std::integral_constant<int, 2>{});
  stage_one(0, std::integral_constant<int, 3>{});
  stage_one(0, std::integral_constant<int, 4>{});
  stage_one(0, std::integral_constant<int, 5>{});
  stage_one(0, std::integral_constant<int, 6>{});
  stage_one(0, std::integral_constant<int, 7>{});
  stage_one(0, std::integral_constant<int, 8>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();

  auto chunk_body = [&](int buf, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const float* Ab = As + buf * CL_A_ST + wm * 64 + li;
    const float* Bb = Bs + buf * CL_B_ST + (wn * 64 + li) * 4 + 2 * kh;
    Frag cur = read_frag(Ab, Bb, 0);
#pragma unroll
    for (int g = 0; g < CL_NG; ++g) {
      Frag nxt = cur;
      if (g + 1 < CL_NG) nxt = read_frag(Ab, Bb, g + 1);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          float b0 = e ? cur.b[0][tap].y : cur.b[0][tap].x;
          float b1 = e ? cur.b[1][tap].y : cur.b[1][tap].x;
          if (tap == 0) {
            b0 = mask_l[0] ? b0 : 0.f;
            b1 = mask_l[1] ? b1 : 0.f;
          } else if (tap == 2) {
            b0 = mask_r[0] ? b0 : 0.f;
            b1 = mask_r[1] ? b1 : 0.f;
          }
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0][e], b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][0][e], b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1][e], b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[tap][1][e], b1, acc[1][1], 0, 0, 0);
          if (MORE) {
            if (g == 0 && e == 0 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 0>{});
            if (g == 0 && e == 0 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 1>{});
            if (g == 0 && e == 0 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 2>{});
            if (g == 0 && e == 1 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 3>{});
            if (g == 0 && e == 1 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 4>{});
            if (g == 0 && e == 1 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 5>{});
            if (g == 1 && e == 0 && tap == 0) stage_one(buf ^ 1, std::integral_constant<int, 6>{});
            if (g == 1 && e == 0 && tap == 1) stage_one(buf ^ 1, std::integral_constant<int, 7>{});
            if (g == 1 && e == 0 && tap == 2) stage_one(buf ^ 1, std::integral_constant<int, 8>{});
          }
        }
      // 24 MFMAs; behind each: one LDS read of the next group, one VALU, every 4th a DMA piece
#define TSPN_G(NV, NVM)                                 \
  __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);   \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x020, NVM, 0);
      TSPN_G(4, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1) TSPN_G(1, 0) TSPN_G(1, 0)
      TSPN_G(1, 0) TSPN_G(1, 1) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1)
      TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1) TSPN_G(1, 0) TSPN_G(1, 0)
      TSPN_G(1, 0) TSPN_G(1, 1) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 0) TSPN_G(1, 1)
#undef TSPN_G
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
    __syncthreads();
  };
  for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c & 1, std::true_type{});
  chunk_body((nchunks - 1) & 1, std::false_type{});

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    if (n >= ncols) continue;
    const int64_t b = n / T;
    const int64_t t = n - b * T;
    float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          float v = acc[mi][ni][e];
          if (bias != nullptr) v += bias[m];
          if (relu) v = fmaxf(v, 0.f);
          ycol[(int64_t)m * ldy] = v;
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
  const bool dma = vec && (Cin % 8 == 0);
  const int kcd = (Cin % 16 != 0) ? 8 : 16;
  const bool dma8 = dma && kcd == 8;
  auto kern = dma ? (dma8 ? conv3_mfma_dma_kernel<8> : conv3_mfma_dma_kernel<16>)
                  : (vec ? conv3_mfma_kernel<true> : conv3_mfma_kernel<false>);
  const int which = dma ? (dma8 ? 3 : 2) : (vec ? 1 : 0);
  const size_t smem = dma8 ? sizeof(float) * 2 * (3 * 8 * BM + 8 * BNP) : SMEM_BYTES;
  static tspn::LdsLimit lds[4];
  if (int rc = lds[which].ensure(reinterpret_cast<const void*>(kern), smem, "tspn_conv3_f32")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), smem,
                     TSPN_STREAM(stream), x, packed, bias, y, (int)Cin, (int)T, (int)M, ncols,
                     (int)tiles_m, (int)tiles_n, relu, (int)T);
  return tspn::check_launch("tspn_conv3_f32");
}


extern "C" int tspn_conv3_tc_f32(const float* x, int64_t B, int64_t T, int64_t Cin,
                                 const float* packed, int64_t M, const float* bias, int relu,
                                 float* y, void* stream) {
  return tspn::conv3_tc_direct(x, B, T, Cin, packed, M, bias, relu, y, T, stream);
}

// internal form with an output row stride ldy >= T (y[b][m][ldy]); used by the fused driver
int tspn::conv3_tc_direct(const float* x, int64_t B, int64_t T, int64_t Cin, const float* packed,
                          int64_t M, const float* bias, int relu, float* y, int64_t ldy,
                          void* stream) {
  TSPN_REQUIRE(ldy >= T && ldy < (1 << 24), TSPN_EINVAL, "tspn_conv3_tc_f32: bad ldy");
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0, TSPN_EINVAL,
               "tspn_conv3_tc_f32: bad sizes B=%lld T=%lld Cin=%lld M=%lld", (long long)B,
               (long long)T, (long long)Cin, (long long)M);
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && packed && y, TSPN_EINVAL, "tspn_conv3_tc_f32: null pointer");
  TSPN_REQUIRE(Cin % CL_KC == 0 && M % 4 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_f32: needs Cin %% 16 == 0 and M %% 4 == 0 (Cin=%lld M=%lld)",
               (long long)Cin, (long long)M);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & 15) == 0,
               TSPN_EUNSUPPORTED, "tspn_conv3_tc_f32: x and packed must be 16-byte aligned");
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_f32: dimension too large");
  const int64_t ncols = B * T;
  const int64_t tiles_m = tspn::ceil_div(M, BM);
  const int64_t tiles_n = tspn::ceil_div(ncols, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_tc_f32: grid too large");
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_mfma_cl_kernel), CL_SMEM_BYTES,
                          "tspn_conv3_tc_f32"))
    return rc;
  hipLaunchKernelGGL(conv3_mfma_cl_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS),
                     CL_SMEM_BYTES, TSPN_STREAM(stream), x, packed, bias, y, (int)Cin, (int)T, (int)M,
                     ncols, (int)tiles_m, (int)tiles_n, relu, (int)ldy);
  return tspn::check_launch("tspn_conv3_tc_f32");
}
