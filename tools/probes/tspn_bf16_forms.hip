// PROBE SOURCE, not part of the library build: csrc/tspn_bf16.hip plus two experimental forms of the bf16 temporal
// conv (conv3_bf16_wide_kernel: 4 waves x 512 registers; conv3_bf16_direct_kernel: 8 waves, weights straight from
// global memory into MFMA operand registers).  Both are correct (tests/test_gpu_bf16.py passes on either) and both
// lose to the shipped conv3_bf16_big_kernel, because the chip is power-limited under bf16 MFMA load: see
// profiles/r2/bf16_conv_power_wall.md.  Build:
//   TSPN_VARIANT_SRC=tools/probes/tspn_bf16_forms.hip tools/build_variant.sh bf16wide tspn_bf16.hip
//   TSPN_VARIANT_SRC=tools/probes/tspn_bf16_forms.hip tools/build_variant.sh bf16direct tspn_bf16.hip -DTSPN_BF16_DIRECT
// bf16-operand relation-scoring path (BASELINE config 3: N=64, T=900, D=1024, bf16) for gfx950.
//
// Semantics (build-defined; the reference has no reduced-precision path — the closest statement is
// its own modules cast with `.bfloat16()`, lib/modeling/relpn/dpn.py:55-73 and model.py:76-88, which
// tests/golden/g8 pins): operands (tracklet features, conv / head / classifier weights) are bf16,
// every product is exact and accumulated in fp32, biases are fp32, the encoder activation
// relu(conv + b) is rounded to bf16 ONCE (what a bf16 Conv1d + ReLU hands to the 1x1 heads), the
// span-pooled feature (mean over frames) is rounded to bf16, head outputs and logits stay fp32.
// With the factorised encoder that means the tracklet projections U, V stay fp32
// and only relu(U[s] + V[o]) is rounded.
//
// Kernels:
//   conv3_bf16_big_kernel       k=3 temporal conv of the tracklet projections as implicit GEMM on
//                               v_mfma_f32_32x32x16_bf16; channels-last x AND channels-last fp32 y
//                               ([tracklet*frame][2C]) so that the pair stage finds the 8 channels a
//                               lane needs contiguous;
//   heads_pairgrid_bf16_kernel  pair stage: relu(U[s]+V[o]) -> bf16 in registers (v_pk_add_f32,
//                               v_cvt_pk_bf16_f32, v_pk_max_i16) as B operand of the [3A,C] head GEMM
//                               on v_mfma_f32_16x16x32_bf16;
//   helpers                     fp32->bf16 cast, weight packing, temporal mean.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ src, int64_t n, __bf16* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (__bf16)src[i];  // round to nearest even (v_cvt_pk_bf16_f32)
}

// conv.weight [M, Cin, 3] fp32 -> [3 taps][Cp/8][Mp][8] bf16 (split > 0: rows [0,M) take input
// channels [0,split) = subject half, rows [M,2M) take [split, 2 split) = object half; Cp = split)
__global__ void pack_conv3_bf16_kernel(const float* __restrict__ W, int64_t M, int64_t Cin,
                                       int64_t split, __bf16* __restrict__ packed) {
  const int64_t Mp = split > 0 ? 2 * M : M;
  const int64_t Cp = split > 0 ? split : Cin;
  const int64_t total = 3 * Cp * Mp;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = o & 7;
    const int64_t r = (o >> 3) % Mp;
    const int64_t cg = (o >> 3) / Mp % (Cp >> 3);
    const int64_t tap = o / (Mp * Cp);
    const int64_t ci = cg * 8 + j;
    const int64_t m = r < M ? r : r - M;
    const int64_t c = r < M ? ci : ci + split;
    packed[o] = (__bf16)W[(m * Cin + c) * 3 + tap];
  }
}

// head weights [H, C] fp32 -> [C/8][16][8] bf16, rows H..15 zero
__global__ void pack_heads_bf16_kernel(const float* __restrict__ W, int64_t H, int64_t C,
                                       __bf16* __restrict__ packed) {
  const int64_t total = C * 16;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = o & 7, h = (o >> 3) & 15, cg = o >> 7;
    packed[o] = h < H ? (__bf16)W[h * C + cg * 8 + j] : (__bf16)0.f;
  }
}

// mean over frames of bf16 features [R, T, D] -> fp32 [R, D] holding bf16-rounded values
__global__ __launch_bounds__(256) void temporal_mean_bf16_kernel(const __bf16* __restrict__ x,
                                                                 int64_t R, int T, int D,
                                                                 float* __restrict__ out) {
  __shared__ float part[4][64][8];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t groups = D >> 3;
  const int64_t item = blockIdx.x * 64LL + tx;  // (row, channel group)
  const bool ok = item < R * groups;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const int64_t r = item / groups, g = item - r * groups;
    const __bf16* p = x + (r * T) * (int64_t)D + g * 8;
    for (int t = ty; t < T; t += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + (int64_t)t * D);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[ty][tx][j] = acc[j];
  __syncthreads();
  if (ty == 0 && ok) {
    const int64_t r = item / groups, g = item - r * groups;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = (part[0][tx][j] + part[1][tx][j]) + (part[2][tx][j] + part[3][tx][j]);
      out[r * D + g * 8 + j] = (float)(__bf16)(s / (float)T);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv3, bf16 operands: implicit GEMM on v_mfma_f32_32x32x16_bf16 (lanes k = 0 / 1 take the two
// 8-channel groups of a 16-channel k-step).  LDS images, both filled by 16-byte LDS-DMA pieces:
//   weights [3 taps][2 channel groups][BM m][8 ch]     a lane's A fragment = one ds_read_b128
//   x       [2 channel groups][BN + 4 column slots][8 ch]   read at 3 shifts (halo columns are ordinary
//                                                      units of the same DMA; no im2col)
constexpr int BM = 128, BN = 128;

// ------------------------------------------------------------------------------------------------
// Ring constants shared with the shipped 256 x 256 kernel below: a chunk is ONE k-step (16 input
// channels x 3 taps) and the LDS holds a ring of 4 stages; the DMA of chunk c+3 is issued while
// chunk c is computed, `s_waitcnt vmcnt(N)` counts only the pieces of chunk c+1 out, and the barrier
// is a bare s_barrier (no fence, which would drain the whole DMA queue).  (A 128 x 128 tile with
// this ring reached 0.9 PFLOP/s and was removed: see DESIGN.md §4b.)
constexpr int R_KC = 16, R_KG = 2, R_NST = 4;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------
// conv3 bf16 (shipped structure): 256 x 256 tile, 8 waves (2 x 4, wave tile 128 m x 64 n =
// 4 x 2 blocks), ring of 4 stages of one k-step each (32.1 KB per stage, 128.5 KB, 1 workgroup/CU with
// 2 waves per SIMD).  Ablation of the earlier 128 x 128 kernels at the config-3 shape: MFMA + epilogue
// 3.1 ms, + LDS fragment reads 4.1, + LDS-DMA 6.5 -- the operand stream from L2 (61 GB per launch,
// 9.4 TB/s) and one ds_read_b128 per MFMA are the limiters, both set by the tile: 256 x 256 halves the
// bytes per MFMA from L2 and needs 0.75 fragment reads per MFMA.
constexpr int G_THREADS = 512;
constexpr int G_BM = 256, G_BN = 256;
constexpr int G_SLP = 260;
constexpr int G_A_ST = 3 * R_KG * G_BM * 16;  // 24576
constexpr int G_X_ST = R_KG * G_SLP * 16;     // 8320
constexpr int G_ST = G_A_ST + G_X_ST;         // 32896
constexpr int G_X_UNITS = R_KG * G_SLP;       // 520 -> 9 pieces
constexpr size_t G_SMEM = (size_t)R_NST * G_ST;

__global__ __launch_bounds__(G_THREADS, 1) void conv3_bf16_big_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n, int ldm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // workgroup -> tile: bijective XCD remap, then groups of 2 weight panels x all column tiles
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GM = 2;
  const int group_sz = GM * tiles_n;
  const int group = wg / group_sz;
  const int first_m = group * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int in_group = wg - group * group_sz;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = tile_m * G_BM;
  const int64_t n0 = (int64_t)tile_n * G_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, kh = lane >> 5;

  // weight pieces: pa = ((tap*2 + group)*4 + quarter), 64 rows each; wave w stages pa = 3w .. 3w+2
  const __bf16* asrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pa = wave * 3 + i;
    const int tap = pa >> 3, kg = (pa >> 2) & 1, quarter = pa & 3;
    int m = m0 + 64 * quarter + lane;
    m = m < M ? m : 0;
    asrc[i] = Wp + (((int64_t)tap * (Cin >> 3) + kg) * M + m) * 8;
  }
  const int64_t a_step = (int64_t)R_KG * M * 8;
  // x pieces: wave w stages units [64w, 64w+64); wave 0 also the 8 units of piece 8
  const __bf16* bsrc[2];
  bool bval[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = 64 * (wave + 8 * q) + lane;
    const int g = u / G_SLP, slot = u - g * G_SLP;
    bval[q] = u < G_X_UNITS && slot < G_BN + 2 && (q == 0 || wave == 0);
    int64_t n = n0 + slot - 1;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bsrc[q] = x + n * Cin + 8 * (g < R_KG ? g : 0);
  }
  auto stage_chunk = [&](int st) {
#if defined(TSPN_BF16_ABL_NODMA)
    return;
#endif
    char* sa = smem + st * G_ST;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      glds16(asrc[i], sa + (wave * 3 + i) * 1024);
      asrc[i] += a_step;
    }
    if (bval[0]) glds16(bsrc[0], sa + G_A_ST + 64 * wave * 16);
    bsrc[0] += R_KC;
    if (wave == 0) {
      if (bval[1]) glds16(bsrc[1], sa + G_A_ST + 64 * 8 * 16);
      bsrc[1] += R_KC;
    }
  };
  auto wait_keep = [&](auto chunks_tag) {   // pieces in flight per chunk: 4 (waves 1-7) or 5 (wave 0)
    constexpr int CH = decltype(chunks_tag)::value;
    if (wave == 0) wait_vmcnt<5 * CH>(); else wait_vmcnt<4 * CH>();
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

  bool mask_l[2], mask_r[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    const int t = (int)(n % T);
    mask_l[ni] = t != 0;
    mask_r[ni] = t != T - 1;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nchunks = Cin / R_KC;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // Software pipeline across chunks (fragment registers a/b[tap]):
  //   top of chunk c:  tap 0 of chunk c is already in registers
  //   MFMA tap 0, with the reads of taps 1, 2 between them
  //   wait for the DMA of chunk c+1, barrier, issue the DMA of chunk c+3
  //   MFMA tap 1, with the tap-0 reads of chunk c+1 between them | MFMA tap 2
  // The barrier sits in the middle of a chunk, so the LDS reads of the next chunk start under the
  // MFMAs of this one; a stage is re-filled only after the barrier that follows its last read.
  bf16x8 a[3][4], b[3][2];
  auto load_tap = [&](int st, int tap) {
    const char* Ab = smem + st * G_ST + (kh * G_BM + wm * 128 + li) * 16;
    const char* Xb = smem + st * G_ST + G_A_ST + (kh * G_SLP + wn * 64 + li) * 16;
#if defined(TSPN_BF16_ABL_NOLDS)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      a[tap][mi] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)st, (unsigned)tap, (unsigned)mi, (unsigned)lane});
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      b[tap][ni] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, (unsigned)ni, (unsigned)tap, (unsigned)st});
    (void)Ab; (void)Xb;
    return;
#endif
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      a[tap][mi] = *reinterpret_cast<const bf16x8*>(Ab + (tap * R_KG * G_BM + mi * 32) * 16);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      b[tap][ni] = *reinterpret_cast<const bf16x8*>(Xb + (ni * 32 + tap) * 16);
  };
  auto mfma_tap = [&](int tap) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bf16x8 bb = b[tap][ni];
      if (tap == 0) bb = mask_l[ni] ? bb : zero8;
      if (tap == 2) bb = mask_r[ni] ? bb : zero8;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap][mi], bb, acc[mi][ni], 0, 0, 0);
    }
  };

  stage_chunk(0);
  if (nchunks > 1) stage_chunk(1);
  if (nchunks > 2) stage_chunk(2);
  if (nchunks > 2) wait_keep(K2{}); else if (nchunks > 1) wait_keep(K1{}); else wait_keep(K0{});
  __builtin_amdgcn_s_barrier();

  // (Tried: the two waves of every SIMD as two groups half a chunk apart -- one in its MFMA phase while
  // the other issues DMA and reads fragments, swapping at every barrier.  5.6 ms against 4.8: the memory
  // phase, i.e. the operand stream from L2 into LDS, is the longer one.  33 KB per chunk and CU at the
  // ~70 GB/s per CU an L2-resident gather into LDS reaches is 1100+ cycles against 1536 cycles of MFMA
  // per chunk; the stream and the MFMAs have to overlap almost perfectly to go beyond ~55 % of peak.)
  load_tap(0, 0);
  int c = 0;
  for (; c + 3 < nchunks; ++c) {        // steady state: chunks c+1 .. c+3 exist
    const int st = c & 3;
    mfma_tap(0);
    load_tap(st, 1);
    load_tap(st, 2);
    // 8 MFMAs; two fragment reads behind each of the first six
#define TSPN_MR(NM, NR)                                \
  __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);  \
  __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    TSPN_MR(1, 2) TSPN_MR(1, 2) TSPN_MR(1, 2) TSPN_MR(1, 2) TSPN_MR(1, 2) TSPN_MR(1, 2) TSPN_MR(2, 0)
    __builtin_amdgcn_sched_barrier(0);
    wait_keep(K1{});                    // chunk c+2 may still fly; c+1 has landed
    __builtin_amdgcn_s_barrier();
    // the two waves of a SIMD (w and w+4) issue their DMA pieces at different times
    if (wm == 0) stage_chunk((c + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
    mfma_tap(1);
    load_tap((c + 1) & 3, 0);
    TSPN_MR(2, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 0)
#undef TSPN_MR
    __builtin_amdgcn_sched_barrier(0);
    if (wm != 0) stage_chunk((c + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
    mfma_tap(2);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; c < nchunks; ++c) {            // tail: nothing left to issue
    const int st = c & 3;
    load_tap(st, 1);
    load_tap(st, 2);
    mfma_tap(0);
    if (c + 2 < nchunks) wait_keep(K1{}); else wait_keep(K0{});
    __builtin_amdgcn_s_barrier();
    mfma_tap(1);
    if (c + 1 < nchunks) load_tap((c + 1) & 3, 0);
    mfma_tap(2);
  }
  __builtin_amdgcn_s_barrier();         // every wave is done with the stages before the epilogue reuses them

  // ---- epilogue.  A lane holds 4 consecutive channels of ONE frame per register quad, so direct
  // stores would touch 32 different 16-KB-strided rows per instruction (measured: 1.1 ms of 6.0 at the
  // config-3 shape).  Each wave instead transposes its 128 m x 32 n half-tiles through 16 KB of the
  // (now idle) stage memory -- unit (16 B) u of row n is kept at u ^ n, conflict-free both ways -- and
  // stores two full 512-byte row segments per instruction.
  {
    char* tw = smem + wave * 16384;
    const int unit = lane & 31, rsel = lane >> 5;
    const int mcol = m0 + wm * 128 + unit * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && mcol < M) bv = *reinterpret_cast<const f32x4*>(bias + mcol);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int eq = 0; eq < 4; ++eq) {
          const int u = mi * 8 + 2 * eq + kh;  // 16-byte unit of channels mi*32 + 8 eq + 4 kh .. +3
          const f32x4 v = {acc[mi][ni][4 * eq], acc[mi][ni][4 * eq + 1], acc[mi][ni][4 * eq + 2],
                           acc[mi][ni][4 * eq + 3]};
          *reinterpret_cast<f32x4*>(tw + li * 512 + ((u ^ li) << 4)) = v;
        }
      // (same wave wrote and reads: the compiler's lgkmcnt wait orders the two)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 2 * r + rsel;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tw + row * 512 + ((unit ^ row) << 4)) + bv;
        const int64_t n = n0 + wn * 64 + ni * 32 + row;
#if defined(TSPN_BF16_ABL_NOSTORE)
        if (n < ncols && mcol < M && v[0] == 12345.678f)
#else
        if (n < ncols && mcol < M)
#endif
          *reinterpret_cast<f32x4*>(y + n * (int64_t)ldm + mcol) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv3 bf16, wide form (Cin % 32 == 0): 256 x 256 tile on FOUR waves, one per SIMD with the whole 512-register
// file.  Wave w owns rows 64 w .. 64 w + 63 for all 256 columns: 2 x 8 accumulator blocks = 256 AGPRs.
//  * weights never touch LDS: the packed layout [tap][Cin/8][M][8] already holds an A fragment of 32x32x16 as two
//    512-byte runs, so a fragment is ONE global_load_dwordx4 per lane straight into the MFMA operand registers, and
//    no other wave needs this wave's rows (no sharing through L1 or LDS to rely on).  Twelve tap-steps (4 chunks
//    of 16 channels, 4 x 1536 MFMA cycles) of lookahead in 96 VGPRs: VMEM returns in order, so that distance is
//    also the latency budget of every x piece issued in between.
//  * x goes through LDS by DMA in super-stages of 32 channels ([4 channel groups][260 column slots][16 B] =
//    16.6 KB, ring of 4), issued three super-stages ahead; ONE barrier per super-stage (per 96 MFMAs of a wave),
//    placed before the last tap-step, whose counted wait on the weights of 12 tap-steps ago has just proven
//    that this wave's pieces of the next super-stage landed (they were issued right before those weights).
//  * LDS reads: 8 B fragments per 16 MFMAs (the 8-wave kernel: 18 per 24, and all its weights on top).
// Loads beyond the end of the K loop are issued anyway, from valid addresses, so that every vmcnt is a constant.
constexpr int W_THREADS = 256;
constexpr int W_BM = 256, W_BN = 256;
constexpr int W_SLP = 260;                       // column slots per channel group (256 + 2 halo, padded)
constexpr int W_XUNITS = 4 * W_SLP;              // 16-byte units per super-stage: 1040 = 16 pieces + 16 units
constexpr int W_XST = W_XUNITS * 16;             // 16640 B
constexpr int W_NXS = 4;                         // ring of super-stages
constexpr int W_LA = 12;                         // weight lookahead in tap-steps
constexpr int W_NDMA = 5;                        // DMA instructions per wave and super-stage
constexpr size_t W_SMEM = (size_t)W_NXS * W_XST; // 66560 B (the epilogue reuses it: 4 x 8 KB)

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
template <class F>
__device__ __forceinline__ void tspn_static_for8(F&& f) {
  f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
  f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
  f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
}
template <int VM>
__device__ __forceinline__ void wait_frag2(bf16x8& r0, bf16x8& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}
__device__ __forceinline__ void load_frag_bf16(bf16x8& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(base) : "memory");
}

__global__ __launch_bounds__(W_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3_bf16_wide_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n, int ldm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GM = 2;
  const int group_sz = GM * tiles_n;
  const int group = wg / group_sz;
  const int first_m = group * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int in_group = wg - group * group_sz;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = tile_m * W_BM;
  const int64_t n0 = (int64_t)tile_n * W_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int nsuper = Cin >> 5;
  const int nsteps = 6 * nsuper;

  // ---- weights: fragment (tap, chunk, mi) = rows m0 + 64 wave + 32 mi + li, channel group 2 chunk + kh
  unsigned voff[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    int m = m0 + wave * 64 + mi * 32 + li;
    m = m < M ? m : M - 1;                       // rows past M: any valid row, never stored
    voff[mi] = (unsigned)(kh * M + m) * 16u;
  }
  const int64_t tap_stride = (int64_t)(Cin >> 3) * M * 16;   // bytes between taps
  const int64_t chunk_stride = (int64_t)2 * M * 16;          // bytes between chunks
  const char* const wbase = reinterpret_cast<const char*>(Wp);
  const char* anext[3] = {wbase, wbase + tap_stride, wbase + 2 * tap_stride};   // per tap: the next chunk to load
  int aleft = nsteps;                            // tap-steps not yet requested
  bf16x8 a[W_LA][2];
  auto load_a = [&](int slot, int tap) {         // requests tap-step (steps issued so far); past the end: chunk 0
    const char* src = aleft > 0 ? anext[tap] : wbase;
    load_frag_bf16(a[slot][0], voff[0], src);
    load_frag_bf16(a[slot][1], voff[1], src);
#if !defined(TSPN_BF16W_PROBE_HOTA)
    anext[tap] += chunk_stride;
#endif
    --aleft;
  };

  // ---- x pieces: unit u = 64 (wave + 4 q) + lane, q = 0..3; the 16 units of piece 16 are fetched by every wave
  const __bf16* bsrc[W_NDMA];
#pragma unroll
  for (int q = 0; q < W_NDMA; ++q) {
    int u = q < 4 ? 64 * (wave + 4 * q) + lane : 1024 + (lane & 15);
    const int g = u / W_SLP, slot = u - g * W_SLP;
    int64_t n = n0 + slot - 1;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bsrc[q] = x + n * Cin + 8 * g;
  }
  int xleft = nsuper;                            // super-stages not yet requested
  auto stage_piece = [&](int st, int q) {        // piece q of the next super-stage -> ring stage st
    char* dst = smem + st * W_XST + (q < 4 ? 64 * (wave + 4 * q) : 1024) * 16;
#if !defined(TSPN_BF16W_ABL_NODMA)              // timing probes (wrong results)
    if (q < 4 || lane < 16) glds16(bsrc[q], dst);
#endif
#if !defined(TSPN_BF16W_PROBE_HOTX)
    if (xleft > 1) bsrc[q] += 32;                // past the end: the last super-stage again (never read)
#endif
  };

  // sequence ends: tap 0 of frame 0 and tap 2 of frame T - 1 read the neighbouring tracklet and must contribute
  // zero.  Per lane one bit per 32-column block; per wave the same bits OR-ed over the lanes (SGPRs), so that the
  // blocks without a sequence end (almost all of them at T = 900) skip the selects behind a scalar branch.
  unsigned lbits = 0, rbits = 0;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    const int64_t n = n0 + ni * 32 + li;
    const int t = (int)(n % T);
    lbits |= (t == 0 ? 1u : 0u) << ni;
    rbits |= (t == T - 1 ? 1u : 0u) << ni;
  }
  unsigned wl = 0, wr = 0;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    wl |= (__builtin_amdgcn_ballot_w64((lbits >> ni) & 1) != 0 ? 1u : 0u) << ni;
    wr |= (__builtin_amdgcn_ballot_w64((rbits >> ni) & 1) != 0 ? 1u : 0u) << ni;
  }
  wl = __builtin_amdgcn_readfirstlane(wl);
  wr = __builtin_amdgcn_readfirstlane(wr);

  f32x16 acc[2][8];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // B fragments of the current tap-step; b[ni] is refilled for the next one as soon as its two MFMAs have been
  // issued.  The reads are inline asm with their own counted waits: left to the compiler, every tap-step began with
  // lgkmcnt(0), i.e. with the full LDS latency of the read issued last (LDS returns in order: the fragment a pair
  // of MFMAs needs always has exactly 7 younger reads behind it).
  bf16x8 b[8];
  const unsigned xlane = lds_addr(smem) + (kh * W_SLP + li) * 16;
  auto load_b1 = [&](auto ni_tag, unsigned stage_addr, auto off_tag) {
    constexpr int NI = decltype(ni_tag)::value, OFF = decltype(off_tag)::value;
    bf16x8& dst = b[NI];                         // (named first: an asm operand alone does not capture `b` here)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(stage_addr), "n"(OFF + NI * 32 * 16) : "memory");
  };
  auto wait_b = [&](int ni) {
    bf16x8& r = b[ni];
    asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(r));
  };
  // ---- prologue, in the order the steady state would have issued it (DMA of a super-stage right before the
  // weights of the tap-step 12 ahead): x(0) | A(0..4) | x(1) | A(5..10) | x(2) | A(11)
#pragma unroll
  for (int q = 0; q < W_NDMA; ++q) stage_piece(0, q);
  --xleft;
#pragma unroll
  for (int k = 0; k < 5; ++k) load_a(k, k % 3);
#pragma unroll
  for (int q = 0; q < W_NDMA; ++q) stage_piece(1, q);
  --xleft;
#pragma unroll
  for (int k = 5; k < 11; ++k) load_a(k, k % 3);
#pragma unroll
  for (int q = 0; q < W_NDMA; ++q) stage_piece(2, q);
  --xleft;
  load_a(11, 2);
  wait_vmcnt<2 * W_LA + 2 * W_NDMA>();           // everything younger than x(0)
  __builtin_amdgcn_s_barrier();
  tspn_static_for8([&](auto ni_tag) { load_b1(ni_tag, xlane, std::integral_constant<int, 0>{}); });
  __builtin_amdgcn_sched_barrier(0);

  // tap-step s = 6 S + P: chunk 2 S + P / 3, tap P % 3; weights in a[SB + P].
  auto tap_step = [&](auto p_tag, auto sb_tag, int S) {
    constexpr int P = decltype(p_tag)::value;
    constexpr int SB = decltype(sb_tag)::value;
    constexpr int TAP = P % 3;
    constexpr int SLOT = SB + P;
    // younger than the loads of this slot: 11 tap-steps of weights and the x pieces issued since
    wait_frag2<2 * (W_LA - 1) + W_NDMA * (P == 5 ? 1 : 2)>(a[SLOT][0], a[SLOT][1]);
    const int st = S & 3;
#if !defined(TSPN_BF16W_ABL_NOBAR)
    if (P == 5) __builtin_amdgcn_s_barrier();    // every wave: pieces of S + 1 landed, reads of S - 1 long done
#endif
    __builtin_amdgcn_sched_barrier(0);
    // where the B fragments of tap-step s + 1 live: chunk (P + 1) / 3 of this stage, or the start of the next one
    const unsigned stn = xlane + (P < 5 ? st : ((S + 1) & 3)) * W_XST;
    constexpr int OFFN = P < 5 ? (2 * ((P + 1) / 3) * W_SLP + (P + 1) % 3) * 16 : 0;
#if !defined(TSPN_BF16W_ABL_NOEDGE)
    if (TAP != 1 && (TAP == 0 ? wl : wr) != 0) {
      // a sequence end inside this wave's columns (rare at T = 900): zero the lanes in place.  asm, behind ONE scalar
      // branch per tap-step: the compiler's form was 32 unconditional selects fed from 16 lane masks in SGPR pairs
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        if (((TAP == 0 ? wl : wr) >> ni) & 1) {
          u32x4 r = __builtin_bit_cast(u32x4, b[ni]);
          unsigned tmp;
          asm volatile(
              "v_bfe_u32 %4, %5, %6, 1\n\t"
              "v_cmp_eq_u32 vcc, 1, %4\n\t"
              "v_cndmask_b32 %0, %0, 0, vcc\n\t"
              "v_cndmask_b32 %1, %1, 0, vcc\n\t"
              "v_cndmask_b32 %2, %2, 0, vcc\n\t"
              "v_cndmask_b32 %3, %3, 0, vcc"
              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "=&v"(tmp)
              : "v"(TAP == 0 ? lbits : rbits), "n"(ni)
              : "vcc");
          b[ni] = __builtin_bit_cast(bf16x8, r);
        }
      }
    }
#endif
    tspn_static_for8([&](auto ni_tag) {
      constexpr int ni = decltype(ni_tag)::value;
      wait_b(ni);
      acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SLOT][0], b[ni], acc[0][ni], 0, 0, 0);
      acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SLOT][1], b[ni], acc[1][ni], 0, 0, 0);
      if (P == 5 && ni < W_NDMA) {               // x(S + 3), one piece per two MFMAs, into the stage S - 1 left
        __builtin_amdgcn_sched_barrier(0);
        stage_piece((S + 3) & 3, ni);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_b1(ni_tag, stn, std::integral_constant<int, OFFN>{});
    });
    if (P == 5) --xleft;
    load_a(SLOT, TAP);                           // tap-step s + 12
    __builtin_amdgcn_sched_barrier(0);
  };
  auto super_stage = [&](auto sb_tag, int S) {
    tap_step(std::integral_constant<int, 0>{}, sb_tag, S);
    tap_step(std::integral_constant<int, 1>{}, sb_tag, S);
    tap_step(std::integral_constant<int, 2>{}, sb_tag, S);
    tap_step(std::integral_constant<int, 3>{}, sb_tag, S);
    tap_step(std::integral_constant<int, 4>{}, sb_tag, S);
    tap_step(std::integral_constant<int, 5>{}, sb_tag, S);
  };
  {
    int S = 0;
    for (; S + 1 < nsuper; S += 2) {
      super_stage(std::integral_constant<int, 0>{}, S);
      super_stage(std::integral_constant<int, 6>{}, S + 1);
    }
    if (S < nsuper) super_stage(std::integral_constant<int, 0>{}, S);
  }
  // The surplus weight loads of the tail are asynchronous writes into a[][]: those registers must stay allocated
  // until the loads have landed (a dead asm output is a register the compiler hands to someone else at once).
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),
                 "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[4][0]), "+v"(a[4][1]), "+v"(a[5][0]), "+v"(a[5][1]));
  asm volatile(""
               : "+v"(a[6][0]), "+v"(a[6][1]), "+v"(a[7][0]), "+v"(a[7][1]), "+v"(a[8][0]), "+v"(a[8][1]),
                 "+v"(a[9][0]), "+v"(a[9][1]), "+v"(a[10][0]), "+v"(a[10][1]), "+v"(a[11][0]), "+v"(a[11][1]));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  __builtin_amdgcn_s_barrier();                  // every wave is done with the ring before the epilogue reuses it

  // ---- epilogue: per 32-column block the wave transposes its 64 m x 32 n block through 8 KB of LDS (16-byte
  // unit u of row n kept at u ^ (n & 15): conflict-free both ways) and stores 256-byte row segments.
  {
    char* tw = smem + wave * 8192;
    const int unit = lane & 15, rsel = lane >> 4;
    const int mcol = m0 + wave * 64 + unit * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && mcol < M) bv = *reinterpret_cast<const f32x4*>(bias + mcol);
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int eq = 0; eq < 4; ++eq) {
          const int u = mi * 8 + 2 * eq + kh;    // 16-byte unit of rows 32 mi + 8 eq + 4 kh .. + 3
          const f32x4 v = {acc[mi][ni][4 * eq], acc[mi][ni][4 * eq + 1], acc[mi][ni][4 * eq + 2],
                           acc[mi][ni][4 * eq + 3]};
          *reinterpret_cast<f32x4*>(tw + li * 256 + ((u ^ (li & 15)) << 4)) = v;
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = 4 * r + rsel;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tw + row * 256 + ((unit ^ (row & 15)) << 4)) + bv;
        const int64_t n = n0 + ni * 32 + row;
        if (n < ncols && mcol < M) *reinterpret_cast<f32x4*>(y + n * (int64_t)ldm + mcol) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv3 bf16, direct-weights form with TWO waves per SIMD (Cin % 32 == 0): the same 256 x 256 tile, x ring and
// weight path as the wide form above, on 8 waves of 64 rows x 128 columns (2 x 4 accumulator blocks, 128 AGPRs).
// Why: a bf16 MFMA lasts 32 cycles, and a lone wave pays every VMEM instruction it issues in MFMA time (measured on
// the wide form: ~200 cycles per LDS-DMA piece, ~50 per weight load; 5900 cycles per super-stage against 3072 of
// MFMA).  With two waves per SIMD one wave's VMEM issue runs under the other's MFMAs.  Waves w and w + 4 (one
// SIMD) own the same 64 rows and the two column halves, so the second request for a weight line hits L1.
// Weight lookahead: 6 tap-steps (48 VGPRs).  x pieces per super-stage: two per wave, wave 0 a third (16 units).
constexpr int D_THREADS = 512;
constexpr int D_LA = 6;

__global__ __launch_bounds__(D_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_bf16_direct_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n, int ldm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GM = 2;
  const int group_sz = GM * tiles_n;
  const int group = wg / group_sz;
  const int first_m = group * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int in_group = wg - group * group_sz;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = tile_m * W_BM;
  const int64_t n0 = (int64_t)tile_n * W_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int li = lane & 31, kh = lane >> 5;
  const int nsuper = Cin >> 5;
  const int nsteps = 6 * nsuper;

  unsigned voff[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    int m = m0 + wm * 64 + mi * 32 + li;
    m = m < M ? m : M - 1;
    voff[mi] = (unsigned)(kh * M + m) * 16u;
  }
  const int64_t tap_stride = (int64_t)(Cin >> 3) * M * 16;
  const int64_t chunk_stride = (int64_t)2 * M * 16;
  const char* const wbase = reinterpret_cast<const char*>(Wp);
  const char* anext[3] = {wbase, wbase + tap_stride, wbase + 2 * tap_stride};
  int aleft = nsteps;
  bf16x8 a[D_LA][2];
  auto load_a = [&](int slot, int tap) {
    const char* src = aleft > 0 ? anext[tap] : wbase;
    load_frag_bf16(a[slot][0], voff[0], src);
    load_frag_bf16(a[slot][1], voff[1], src);
    anext[tap] += chunk_stride;
    --aleft;
  };

  // x pieces: wave w takes units [64 w, 64 w + 64) and [64 (w + 8), ...); wave 0 also the 16 units of piece 16
  const __bf16* bsrc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    int u = q < 2 ? 64 * (wave + 8 * q) + lane : 1024 + (lane & 15);
    const int g = u / W_SLP, slot = u - g * W_SLP;
    int64_t n = n0 + slot - 1;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bsrc[q] = x + n * Cin + 8 * g;
  }
  int xleft = nsuper;
  auto stage_piece = [&](int st, int q) {
    char* dst = smem + st * W_XST + (q < 2 ? 64 * (wave + 8 * q) : 1024) * 16;
    if (q < 2 || lane < 16) glds16(bsrc[q], dst);
    if (xleft > 1) bsrc[q] += 32;
  };
  auto stage_all = [&](int st) {                 // 2 pieces, wave 0: 3
    stage_piece(st, 0);
    stage_piece(st, 1);
    if (wave == 0) stage_piece(st, 2);
    --xleft;
  };

  unsigned lbits = 0, rbits = 0;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int64_t n = n0 + wn * 128 + ni * 32 + li;
    const int t = (int)(n % T);
    lbits |= (t == 0 ? 1u : 0u) << ni;
    rbits |= (t == T - 1 ? 1u : 0u) << ni;
  }
  unsigned wl = 0, wr = 0;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    wl |= (__builtin_amdgcn_ballot_w64((lbits >> ni) & 1) != 0 ? 1u : 0u) << ni;
    wr |= (__builtin_amdgcn_ballot_w64((rbits >> ni) & 1) != 0 ? 1u : 0u) << ni;
  }
  wl = __builtin_amdgcn_readfirstlane(wl);
  wr = __builtin_amdgcn_readfirstlane(wr);

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  bf16x8 b[4];
  const unsigned xlane = lds_addr(smem) + (kh * W_SLP + wn * 128 + li) * 16;
  auto load_b1 = [&](auto ni_tag, unsigned stage_addr, auto off_tag) {
    constexpr int NI = decltype(ni_tag)::value, OFF = decltype(off_tag)::value;
    bf16x8& dst = b[NI];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(stage_addr), "n"(OFF + NI * 32 * 16) : "memory");
  };
  auto wait_b = [&](int ni) {                    // 3 younger reads behind the fragment a pair of MFMAs needs
    bf16x8& r = b[ni];
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r));
  };
  auto for4 = [&](auto&& f) {
    f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{});
    f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
  };
  // counted wait on the weights of a slot; NP = pieces this wave issues per super-stage
  auto wait_a = [&](auto k_tag, bf16x8& r0, bf16x8& r1) {
    constexpr int K = decltype(k_tag)::value;    // x bursts younger than the slot's loads: 0 or 1
    if (K == 0) wait_frag2<2 * (D_LA - 1)>(r0, r1);
    else if (wave == 0) wait_frag2<2 * (D_LA - 1) + 3>(r0, r1);
    else wait_frag2<2 * (D_LA - 1) + 2>(r0, r1);
  };

  // ---- prologue in steady-state order: x(0) | x(1) | A(0..4) | x(2) | A(5)
  stage_all(0);
  stage_all(1);
#pragma unroll
  for (int k = 0; k < 5; ++k) load_a(k, k % 3);
  stage_all(2);
  load_a(5, 2);
  if (wave == 0) wait_vmcnt<2 * D_LA + 6>(); else wait_vmcnt<2 * D_LA + 4>();   // everything younger than x(0)
  __builtin_amdgcn_s_barrier();
  for4([&](auto ni_tag) { load_b1(ni_tag, xlane, std::integral_constant<int, 0>{}); });
  __builtin_amdgcn_sched_barrier(0);

  auto tap_step = [&](auto p_tag, int S) {
    constexpr int P = decltype(p_tag)::value;
    constexpr int TAP = P % 3;
    wait_a(std::integral_constant<int, (P == 5 ? 0 : 1)>{}, a[P][0], a[P][1]);
    const int st = S & 3;
    if (P == 5) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned stn = xlane + (P < 5 ? st : ((S + 1) & 3)) * W_XST;
    constexpr int OFFN = P < 5 ? (2 * ((P + 1) / 3) * W_SLP + (P + 1) % 3) * 16 : 0;
    if (TAP != 1 && (TAP == 0 ? wl : wr) != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if (((TAP == 0 ? wl : wr) >> ni) & 1) {
          u32x4 r = __builtin_bit_cast(u32x4, b[ni]);
          unsigned tmp;
          asm volatile(
              "v_bfe_u32 %4, %5, %6, 1\n\t"
              "v_cmp_eq_u32 vcc, 1, %4\n\t"
              "v_cndmask_b32 %0, %0, 0, vcc\n\t"
              "v_cndmask_b32 %1, %1, 0, vcc\n\t"
              "v_cndmask_b32 %2, %2, 0, vcc\n\t"
              "v_cndmask_b32 %3, %3, 0, vcc"
              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "=&v"(tmp)
              : "v"(TAP == 0 ? lbits : rbits), "n"(ni)
              : "vcc");
          b[ni] = __builtin_bit_cast(bf16x8, r);
        }
      }
    }
    for4([&](auto ni_tag) {
      constexpr int ni = decltype(ni_tag)::value;
      wait_b(ni);
      acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][0], b[ni], acc[0][ni], 0, 0, 0);
      acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][1], b[ni], acc[1][ni], 0, 0, 0);
      if (P == 5 && ni < 2) {                    // x(S + 3) into the stage S - 1 left
        __builtin_amdgcn_sched_barrier(0);
        stage_piece((S + 3) & 3, ni);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (P == 5 && ni == 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (wave == 0) stage_piece((S + 3) & 3, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_b1(ni_tag, stn, std::integral_constant<int, OFFN>{});
    });
    if (P == 5) --xleft;
    load_a(P, TAP);                              // tap-step s + 6
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int S = 0; S < nsuper; ++S) {
    tap_step(std::integral_constant<int, 0>{}, S);
    tap_step(std::integral_constant<int, 1>{}, S);
    tap_step(std::integral_constant<int, 2>{}, S);
    tap_step(std::integral_constant<int, 3>{}, S);
    tap_step(std::integral_constant<int, 4>{}, S);
    tap_step(std::integral_constant<int, 5>{}, S);
  }
  // surplus loads of the tail: their destination registers stay allocated until they have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),
                 "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[4][0]), "+v"(a[4][1]), "+v"(a[5][0]), "+v"(a[5][1]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  __builtin_amdgcn_s_barrier();

  {
    char* tw = smem + wave * 8192;
    const int unit = lane & 15, rsel = lane >> 4;
    const int mcol = m0 + wm * 64 + unit * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && mcol < M) bv = *reinterpret_cast<const f32x4*>(bias + mcol);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int eq = 0; eq < 4; ++eq) {
          const int u = mi * 8 + 2 * eq + kh;
          const f32x4 v = {acc[mi][ni][4 * eq], acc[mi][ni][4 * eq + 1], acc[mi][ni][4 * eq + 2],
                           acc[mi][ni][4 * eq + 3]};
          *reinterpret_cast<f32x4*>(tw + li * 256 + ((u ^ (li & 15)) << 4)) = v;
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = 4 * r + rsel;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tw + row * 256 + ((unit ^ (row & 15)) << 4)) + bv;
        const int64_t n = n0 + wn * 128 + ni * 32 + row;
        if (n < ncols && mcol < M) *reinterpret_cast<f32x4*>(y + n * (int64_t)ldm + mcol) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pair stage, bf16: out[p][h][t] = bh[h] + sum_c Wh[h][c] * bf16(relu(U[s][t][c] + V[o][t][c])) for
// the canonical pair table.  Workgroup = (video, 8 subjects x 8 objects, 16 frames); wave w owns
// subjects 2w, 2w+1 x 8 objects (16 accumulator tiles of 16 heads x 16 frames).  Per k-step of 32
// channels the 16 projection rows (8 U + 8 V) x 16 frames x 32 ch fp32 = 32 KB are staged by LDS-DMA
// (double-buffered).  The DMA source of each lane is chosen (comment at `src` below) so that a piece
// fetches complete 128-byte lines AND the B fragment of lane (f = l&15, kg = l>>4) -- channels
// 8kg .. 8kg+7 of frame f -- is two conflict-free ds_read_b128.  The VALU work (1.5 packed
// instructions per activation) hides under the operand stream, which is what bounds the kernel.
constexpr int HP_FB = 16;
constexpr int HP_KC = 32;
constexpr int HP_ROW = HP_FB * HP_KC * 4;  // 2048 B

__device__ __forceinline__ unsigned relu_pack(float a, float b) {
  f32x2 s = {a, b};
  const bf16x2 h = __builtin_convertvector(s, bf16x2);
  const s16x2 z = {0, 0};
  // ReLU on the packed pair: negative floats are negative int16 (v_pk_max_i16); -0 -> +0
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, h), z));
}

// NW waves; workgroup = 2 NW subjects x OB objects x 16 frames, wave w owns subjects 2w, 2w+1.
// <4, 8>: 8 x 8 pairs, 32 KB per stage, 2 workgroups/CU (small N).  <8, 16>: 16 x 16 pairs, 64 KB per
// stage, 1 workgroup/CU -- half the bytes streamed from L2 per activation, which is what bounds the
// kernel (ablation at the config-3 shape, 8 x 8: 4.5 ms, without the DMA stream 2.2, without the
// VALU work still 4.5).
template <int NW, int OB>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void heads_pairgrid_bf16_kernel(
    const float* __restrict__ y, int64_t ldm, int B, int N, int C, int T,
    const __bf16* __restrict__ Whp, const float* __restrict__ bh, int H, float* __restrict__ out,
    int nsb, int nob, int nfb) {
  constexpr int SBLK = 2 * NW;
  constexpr int ROWS = SBLK + OB;
  constexpr int ST = ROWS * HP_ROW + 1024;  // + the k-step's slice of the head weights (one piece)
  static_assert(ROWS == 4 * NW, "each wave stages 4 rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ob = wg % nob;
  wg /= nob;
  const int sb = wg % nsb;
  wg /= nsb;
  const int fb = wg % nfb;
  const int b = wg / nfb;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = lane & 15, kg = lane >> 4;
  const int t0 = fb * HP_FB;

  // DMA sources: wave w stages rows 4w .. 4w+3 (2 pieces each); row r < SBLK: subject SBLK sb + r
  // (U half, channels [0,C)), else object OB ob + r - SBLK (V half, channels [C,2C))
  // LDS image of a row: two pieces of 8 frames; inside a piece position = 16 X + slot with
  //   slot = (f & 7) + 8 ((q >> 1) & 1),  X = 2 (q >> 2) + (q & 1)      (q = 16-byte channel quad 0..7)
  // so that (a) one DMA piece fetches 8 complete 128-byte lines of y (8 frames x 32 channels) and
  // (b) the fragment read of lane (f, kg) for quad 2 kg + r sits at slot (f & 7) + 8 (kg & 1): the
  // four 16-lane groups of a ds_read_b128 each cover all 16 slots -- conflict-free.
  const float* src[8];
  {
    const int fq = lane & 7;
    const int q = (lane >> 5) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wave * 4 + (i >> 1), j = i & 1;
      int trk = r < SBLK ? sb * SBLK + r : ob * OB + r - SBLK;
      trk = min(trk, N - 1);
      const int t = min(t0 + 8 * j + fq, T - 1);
      src[i] = y + (((int64_t)b * N + trk) * T + t) * ldm + (r < SBLK ? 0 : C) + 4 * q;
    }
  }
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(Whp) + lane;          // + 64 per k-step
  auto stage = [&](int buf) {
#if defined(TSPN_HPB_ABL_NODMA)
    return;
#endif
    char* dst = smem + buf * ST + wave * 4 * HP_ROW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      glds16(src[i], dst + i * 1024);
      src[i] += HP_KC;
    }
    // head weights of the k-step, [4 kg][16 h][8 ch] bf16 = the packed layout itself; staged through
    // LDS as well so that no register-returning global load (whose wait the compiler would place at the
    // top of the loop, serialising the whole DMA queue with the compute) is left in the loop
    if (wave == 0) {
      glds16(wsrc, smem + buf * ST + ROWS * HP_ROW);
      wsrc += 64;
    }
  };

  f32x4 acc[2][OB];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[s][o] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = C / HP_KC;
  // fragment of lane (f, kg): quads 2 kg (here) and 2 kg + 1 (256 bytes further)
  const int frag_off = (64 * (f >> 3) + 32 * (kg >> 1) + 8 * (kg & 1) + (f & 7)) * 16;
  stage(0);
  __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0)
  __builtin_amdgcn_s_barrier();

  for (int k = 0; k < nk; ++k) {
    const int buf = k & 1;
    if (k + 1 < nk) stage(buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 wfrag = *reinterpret_cast<const bf16x8*>(smem + buf * ST + ROWS * HP_ROW + lane * 16);
    const char* base = smem + buf * ST + frag_off;
    f32x4 u[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u[s][0] = *reinterpret_cast<const f32x4*>(base + (2 * wave + s) * HP_ROW);
      u[s][1] = *reinterpret_cast<const f32x4*>(base + (2 * wave + s) * HP_ROW + 256);
    }
    // V fragments are read two objects ahead of their use (LDS latency off the critical path).  The
    // reads and their counted waits are written out: left to itself the compiler issues every
    // fragment read right before its first use and waits for it at once (32 exposed LDS round trips
    // per k-step).  LDS returns in order, so "lgkmcnt(n)" = all but the newest n reads have landed.
    const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(base);
    f32x4 vq[3][2];
#define TSPN_VREAD(slot, o)                                                                             \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][0]) : "v"(vaddr), "n"((SBLK + (o)) * HP_ROW)); \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][1]) : "v"(vaddr), "n"((SBLK + (o)) * HP_ROW + 256));
    TSPN_VREAD(0, 0)
    TSPN_VREAD(1, 1)
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      if (o + 2 < OB) {
        TSPN_VREAD((o + 2) % 3, o + 2)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else if (o + 1 < OB) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      }
      const f32x4 v0 = vq[o % 3][0], v1 = vq[o % 3][1];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#if defined(TSPN_HPB_ABL_NOVALU)
        u32x4 pk = {__builtin_bit_cast(unsigned, u[s][0][0]) ^ __builtin_bit_cast(unsigned, v0[0]),
                    __builtin_bit_cast(unsigned, u[s][0][1]) ^ __builtin_bit_cast(unsigned, v0[2]),
                    __builtin_bit_cast(unsigned, u[s][1][0]) ^ __builtin_bit_cast(unsigned, v1[1]),
                    __builtin_bit_cast(unsigned, u[s][1][1]) ^ __builtin_bit_cast(unsigned, v1[3])};
#else
        const f32x4 a0 = u[s][0] + v0, a1 = u[s][1] + v1;
        u32x4 pk = {relu_pack(a0[0], a0[1]), relu_pack(a0[2], a0[3]), relu_pack(a1[0], a1[1]),
                    relu_pack(a1[2], a1[3])};
#endif
        acc[s][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag, __builtin_bit_cast(bf16x8, pk),
                                                             acc[s][o], 0, 0, 0);
      }
    }
#undef TSPN_VREAD
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): the next k-step is in LDS
    __builtin_amdgcn_s_barrier();
  }

  // epilogue: lane = (frame f, head group hg): heads 4 hg .. 4 hg + 3
  const int t = t0 + f;
  const int hg = lane >> 4;
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = (4 * hg + r < H) ? bh[4 * hg + r] : 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sg = sb * SBLK + 2 * wave + s;
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      const int og = ob * OB + o;
      if (sg >= N || og >= N || sg == og || t >= T) continue;
      const int64_t p = (int64_t)b * N * (N - 1) + (int64_t)sg * (N - 1) + (og < sg ? og : og - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 4 * hg + r;
        if (h < H) out[(p * H + h) * T + t] = acc[s][o][r] + bias[r];
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int tspn_cast_bf16(const float* src, int64_t n, uint16_t* dst, void* stream) {
  TSPN_REQUIRE(n >= 0, TSPN_EINVAL, "tspn_cast_bf16: n=%lld", (long long)n);
  if (n == 0) return TSPN_OK;
  TSPN_REQUIRE(src && dst, TSPN_EINVAL, "tspn_cast_bf16: null pointer");
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(n, 256), 16384);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), src, n,
                     reinterpret_cast<__bf16*>(dst));
  return tspn::check_launch("tspn_cast_bf16");
}

extern "C" int tspn_pack_conv3_bf16(const float* W, int64_t M, int64_t Cin, int64_t split,
                                    uint16_t* packed, void* stream) {
  TSPN_REQUIRE(W && packed && M > 0 && Cin > 0 && split >= 0, TSPN_EINVAL,
               "tspn_pack_conv3_bf16: bad arguments");
  TSPN_REQUIRE(split == 0 || Cin == 2 * split, TSPN_EINVAL,
               "tspn_pack_conv3_bf16: split=%lld requires Cin == 2*split (Cin=%lld)", (long long)split,
               (long long)Cin);
  TSPN_REQUIRE((split > 0 ? split : Cin) % 8 == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv3_bf16: packed input channels must be a multiple of 8");
  const int64_t total = 3 * M * Cin;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv3_bf16_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), W, M, Cin,
                     split, reinterpret_cast<__bf16*>(packed));
  return tspn::check_launch("tspn_pack_conv3_bf16");
}

extern "C" int tspn_pack_heads_bf16(const float* W, int64_t H, int64_t C, uint16_t* packed, void* stream) {
  TSPN_REQUIRE(W && packed && H > 0 && H <= 16 && C > 0 && C % 8 == 0, TSPN_EINVAL,
               "tspn_pack_heads_bf16: bad arguments (H=%lld <= 16, C=%lld %% 8 == 0)", (long long)H,
               (long long)C);
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(C * 16, 256), 8192);
  hipLaunchKernelGGL(pack_heads_bf16_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), W, H, C,
                     reinterpret_cast<__bf16*>(packed));
  return tspn::check_launch("tspn_pack_heads_bf16");
}

extern "C" int tspn_temporal_mean_bf16(const uint16_t* x, int64_t R, int64_t T, int64_t D, float* out,
                                       void* stream) {
  TSPN_REQUIRE(R >= 0 && T > 0 && D > 0 && D % 8 == 0 && T < (1 << 30) && D < (1 << 30), TSPN_EINVAL,
               "tspn_temporal_mean_bf16: bad sizes R=%lld T=%lld D=%lld (D %% 8 == 0)", (long long)R,
               (long long)T, (long long)D);
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out && aligned16(x), TSPN_EINVAL, "tspn_temporal_mean_bf16: null / unaligned pointer");
  const int64_t items = R * (D / 8);
  hipLaunchKernelGGL(temporal_mean_bf16_kernel, dim3((unsigned)tspn::ceil_div(items, 64)), dim3(256), 0,
                     TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x), R, (int)T, (int)D, out);
  return tspn::check_launch("tspn_temporal_mean_bf16");
}

extern "C" int tspn_conv3_tc_bf16(const uint16_t* x, int64_t B, int64_t T, int64_t Cin,
                                  const uint16_t* packed, int64_t M, const float* bias, float* y,
                                  int64_t ldm, void* stream) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0 && ldm >= M, TSPN_EINVAL,
               "tspn_conv3_tc_bf16: bad sizes B=%lld T=%lld Cin=%lld M=%lld ldm=%lld", (long long)B,
               (long long)T, (long long)Cin, (long long)M, (long long)ldm);
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && packed && y, TSPN_EINVAL, "tspn_conv3_tc_bf16: null pointer");
  TSPN_REQUIRE(Cin % R_KC == 0 && M % 4 == 0 && ldm % 4 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_bf16: needs Cin %% 16 == 0, M %% 4 == 0, ldm %% 4 == 0 (Cin=%lld M=%lld ldm=%lld)",
               (long long)Cin, (long long)M, (long long)ldm);
  TSPN_REQUIRE(aligned16(x) && aligned16(packed) && aligned16(y) && (bias == nullptr || aligned16(bias)),
               TSPN_EUNSUPPORTED, "tspn_conv3_tc_bf16: pointers must be 16-byte aligned");
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24) && ldm < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_bf16: dimension too large");
  const int64_t ncols = B * T;
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = tspn::ceil_div(ncols, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_tc_bf16: grid too large");
#if defined(TSPN_BF16_DIRECT)
  if (Cin % 32 == 0) {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_bf16_direct_kernel), W_SMEM, "tspn_conv3_tc_bf16"))
      return rc;
    const int64_t tm = tspn::ceil_div(M, W_BM), tn = tspn::ceil_div(ncols, W_BN);
    hipLaunchKernelGGL(conv3_bf16_direct_kernel, dim3((unsigned)(tm * tn)), dim3(D_THREADS), W_SMEM,
                       TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(packed), bias, y, (int)Cin, (int)T, (int)M, ncols,
                       (int)tm, (int)tn, (int)ldm);
    return tspn::check_launch("tspn_conv3_tc_bf16");
  }
#endif
#if !defined(TSPN_BF16_NO_WIDE)
  if (Cin % 32 == 0) {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_bf16_wide_kernel), W_SMEM, "tspn_conv3_tc_bf16"))
      return rc;
    const int64_t tm = tspn::ceil_div(M, W_BM), tn = tspn::ceil_div(ncols, W_BN);
    hipLaunchKernelGGL(conv3_bf16_wide_kernel, dim3((unsigned)(tm * tn)), dim3(W_THREADS), W_SMEM,
                       TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(packed), bias, y, (int)Cin, (int)T, (int)M, ncols,
                       (int)tm, (int)tn, (int)ldm);
    return tspn::check_launch("tspn_conv3_tc_bf16");
  }
#endif
  {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_bf16_big_kernel), G_SMEM, "tspn_conv3_tc_bf16"))
      return rc;
    const int64_t tm = tspn::ceil_div(M, G_BM), tn = tspn::ceil_div(ncols, G_BN);
    hipLaunchKernelGGL(conv3_bf16_big_kernel, dim3((unsigned)(tm * tn)), dim3(G_THREADS), G_SMEM,
                       TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(packed), bias, y, (int)Cin, (int)T, (int)M, ncols,
                       (int)tm, (int)tn, (int)ldm);
    return tspn::check_launch("tspn_conv3_tc_bf16");
  }
}

extern "C" int tspn_heads_pairgrid_bf16(const float* y, int64_t ldm, int64_t B, int64_t N, int64_t C,
                                        int64_t T, const uint16_t* head_packed, const float* head_b,
                                        int64_t H, float* out, void* stream) {
  TSPN_REQUIRE(B >= 0 && N >= 0 && C > 0 && T > 0 && H > 0 && H <= 16 && ldm >= 2 * C, TSPN_EINVAL,
               "tspn_heads_pairgrid_bf16: bad sizes B=%lld N=%lld C=%lld T=%lld H=%lld ldm=%lld",
               (long long)B, (long long)N, (long long)C, (long long)T, (long long)H, (long long)ldm);
  if (B == 0 || N < 2) return TSPN_OK;
  TSPN_REQUIRE(y && head_packed && head_b && out, TSPN_EINVAL, "tspn_heads_pairgrid_bf16: null pointer");
  TSPN_REQUIRE(C % HP_KC == 0 && ldm % 4 == 0 && aligned16(y) && aligned16(head_packed), TSPN_EUNSUPPORTED,
               "tspn_heads_pairgrid_bf16: needs C %% 32 == 0, ldm %% 4 == 0, 16-byte aligned y / weights");
  const bool big = N > 12;
  const int64_t sblk = big ? 16 : 8;
  const int64_t nsb = tspn::ceil_div(N, sblk), nfb = tspn::ceil_div(T, HP_FB);
  const int64_t grid = B * nsb * nsb * nfb;
  TSPN_REQUIRE(grid < (1LL << 31) && N < (1 << 20) && T < (1 << 24) && C < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_heads_pairgrid_bf16: problem too large");
  const size_t smem = 2 * ((size_t)(2 * sblk) * HP_ROW + 1024);
  static tspn::LdsLimit lds[2];
  if (int rc = big ? lds[1].ensure(reinterpret_cast<const void*>(heads_pairgrid_bf16_kernel<8, 16>), smem,
                                   "tspn_heads_pairgrid_bf16")
                   : lds[0].ensure(reinterpret_cast<const void*>(heads_pairgrid_bf16_kernel<4, 8>), smem,
                                   "tspn_heads_pairgrid_bf16"))
    return rc;
  if (big)
    hipLaunchKernelGGL((heads_pairgrid_bf16_kernel<8, 16>), dim3((unsigned)grid), dim3(512), smem,
                       TSPN_STREAM(stream), y, ldm, (int)B, (int)N, (int)C, (int)T,
                       reinterpret_cast<const __bf16*>(head_packed), head_b, (int)H, out, (int)nsb, (int)nsb,
                       (int)nfb);
  else
    hipLaunchKernelGGL((heads_pairgrid_bf16_kernel<4, 8>), dim3((unsigned)grid), dim3(256), smem,
                       TSPN_STREAM(stream), y, ldm, (int)B, (int)N, (int)C, (int)T,
                       reinterpret_cast<const __bf16*>(head_packed), head_b, (int)H, out, (int)nsb, (int)nsb,
                       (int)nfb);
  return tspn::check_launch("tspn_heads_pairgrid_bf16");
}

// ---- whole pass ---------------------------------------------------------------------------------
namespace {
struct Bf16Layout {
  size_t bias2, y, fbar, lin, lin_bytes, total;
};
Bf16Layout bf16_layout(const tspn_fused_bf16_desc* d) {
  Bf16Layout L{};
  const size_t NT = (size_t)d->B * d->N, C = 2 * (size_t)d->D;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += tspn::align_up(bytes, 256);
    return at;
  };
  L.bias2 = take(2 * C * sizeof(float));
  L.y = take(NT * d->T * 2 * C * sizeof(float));
  L.fbar = take(NT * d->D * sizeof(float));
  L.lin_bytes = tspn::pair_predicate_workspace_bytes((int64_t)NT, d->D, d->K);
  L.lin = take(L.lin_bytes);
  L.total = off;
  return L;
}
}  // namespace

extern "C" size_t tspn_forward_fused_bf16_workspace_bytes(const tspn_fused_bf16_desc* d) {
  if (!d || d->B <= 0 || d->N <= 0 || d->T <= 0 || d->D <= 0 || d->K <= 0) return 0;
  return bf16_layout(d).total;
}

extern "C" int tspn_forward_fused_bf16(const tspn_fused_bf16_desc* d, void* stream) {
  TSPN_REQUIRE(d, TSPN_EINVAL, "tspn_forward_fused_bf16: null descriptor");
  TSPN_REQUIRE(d->B >= 0 && d->N >= 0 && d->T > 0 && d->D > 0 && d->A > 0 && d->K > 0, TSPN_EINVAL,
               "tspn_forward_fused_bf16: bad sizes");
  TSPN_REQUIRE(3 * d->A <= 16, TSPN_EUNSUPPORTED, "tspn_forward_fused_bf16: 3A=%lld > 16", (long long)(3 * d->A));
  const int64_t NT = d->B * d->N, P = d->B * d->N * (d->N - 1), C = 2 * d->D;
  TSPN_REQUIRE(d->P == P, TSPN_EINVAL,
               "tspn_forward_fused_bf16: P=%lld, the canonical pair table has B*N*(N-1)=%lld rows",
               (long long)d->P, (long long)P);
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(d->feats && d->pairs && d->conv_packed && d->conv_bias && d->head_packed && d->head_b &&
                   d->cls_w && d->cls_b && d->out_heads && d->out_logits && d->workspace,
               TSPN_EINVAL, "tspn_forward_fused_bf16: null pointer");
  TSPN_REQUIRE(d->D % 16 == 0, TSPN_EUNSUPPORTED, "tspn_forward_fused_bf16: needs D %% 16 == 0 (D=%lld)",
               (long long)d->D);
  const Bf16Layout L = bf16_layout(d);
  TSPN_REQUIRE(d->workspace_bytes >= L.total, TSPN_EWORKSPACE,
               "tspn_forward_fused_bf16: workspace %zu < %zu bytes", d->workspace_bytes, L.total);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 255) == 0, TSPN_EINVAL,
               "tspn_forward_fused_bf16: workspace must be 256-byte aligned");
  char* ws = static_cast<char*>(d->workspace);
  float* bias2 = reinterpret_cast<float*>(ws + L.bias2);
  float* y = reinterpret_cast<float*>(ws + L.y);
  float* fbar = reinterpret_cast<float*>(ws + L.fbar);
  hipStream_t s = TSPN_STREAM(stream);
  // conv bias rides on the subject half: U' = U + b, V' = V
  if (hipMemsetAsync(bias2 + C, 0, C * sizeof(float), s) != hipSuccess ||
      hipMemcpyAsync(bias2, d->conv_bias, C * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return tspn::fail(TSPN_ELAUNCH, "tspn_forward_fused_bf16: bias staging failed");
  int rc;
  if (d->ev_conv_begin) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_begin), s);
  rc = tspn_conv3_tc_bf16(d->feats, NT, d->T, d->D, d->conv_packed, 2 * C, bias2, y, 2 * C, stream);
  if (d->ev_conv_end) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_end), s);
  if (rc) return rc;
  if ((rc = tspn_heads_pairgrid_bf16(y, 2 * C, d->B, d->N, C, d->T, d->head_packed, d->head_b, 3 * d->A,
                                     d->out_heads, stream)))
    return rc;
  if ((rc = tspn_temporal_mean_bf16(d->feats, NT, d->T, d->D, fbar, stream))) return rc;
  return tspn::pair_predicate(fbar, NT, d->D, d->pairs, P, d->cls_w, d->cls_b, d->K, d->out_logits,
                              ws + L.lin, L.lin_bytes, stream);
}
