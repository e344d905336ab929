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
    float* __restrict__ y, int Cin, int T, int M, int64_t ncols, int tiles_m, int tiles_n, int ldm, int w_bytes) {
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

  // weight pieces: pa = ((tap*2 + group)*4 + quarter), 64 rows each; wave w stages pa = 3w .. 3w+2.
  // The pieces are BUFFER loads (buffer_load_dwordx4 ... offen lds: an SGPR descriptor, a fixed 32-bit lane offset and a
  // scalar offset that advances per chunk) rather than global_load_lds_dwordx4 with a 64-bit pointer per lane: beside MFMAs
  // the buffer form costs its wave 110 - 140 cycles of issue per piece against 175 - 195 (tools/probes/lds_dma_issue_probe.hip,
  // profiles/r4/lds_dma_issue_probe.txt), and the per-chunk pointer arithmetic moves from the vector to the scalar unit.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wp), 0, w_bytes, 0x00020000);
  unsigned aoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pa = wave * 3 + i;
    const int tap = pa >> 3, kg = (pa >> 2) & 1, quarter = pa & 3;
    int m = m0 + 64 * quarter + lane;
    m = m < M ? m : 0;
    aoff[i] = (unsigned)((((int64_t)tap * (Cin >> 3) + kg) * M + m) * 16);
  }
  const int a_step = R_KG * M * 16;                        // bytes per chunk
  // x pieces: wave w stages units [64w, 64w+64); wave 0 also the 8 units of piece 8.  Descriptor based at the first
  // column this tile reads (any clip count: the lane offsets stay below 260 columns)
  const int64_t nbase = n0 > 0 ? n0 - 1 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x) + nbase * Cin, 0, (int)min((int64_t)(G_BN + 4) * Cin * 2, (ncols - nbase) * (int64_t)Cin * 2), 0x00020000);
  unsigned boff[2];
  bool bval[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = 64 * (wave + 8 * q) + lane;
    const int g = u / G_SLP, slot = u - g * G_SLP;
    bval[q] = u < G_X_UNITS && slot < G_BN + 2 && (q == 0 || wave == 0);
    int64_t n = n0 + slot - 1;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    boff[q] = (unsigned)((n - nbase) * Cin * 2 + 16 * (g < R_KG ? g : 0));
  }
  int a_soff = 0, x_soff = 0;                              // scalar offsets of the next chunk to stage
  auto bglds16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff, char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, (int)voff, soff, 0, 0);
  };
  auto stage_chunk = [&](int st) {
    char* sa = smem + st * G_ST;
#pragma unroll
    for (int i = 0; i < 3; ++i) bglds16(rsrc_a, aoff[i], a_soff, sa + (wave * 3 + i) * 1024);
    a_soff += a_step;
    if (bval[0]) bglds16(rsrc_x, boff[0], x_soff, sa + G_A_ST + 64 * wave * 16);
    if (wave == 0) {
      if (bval[1]) bglds16(rsrc_x, boff[1], x_soff, sa + G_A_ST + 64 * 8 * 16);
    }
    x_soff += R_KC * 2;
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
        if (n < ncols && mcol < M)
          *reinterpret_cast<f32x4*>(y + n * (int64_t)ldm + mcol) = v;
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
#ifndef TSPN_HPB_SW
#define TSPN_HPB_SW 2        // subjects per wave of the <8, 16> form (probe knob: 4 = 4 subjects x 8 objects per wave)
#endif
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
template <int NW, int OB, int SW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void heads_pairgrid_bf16_kernel(
    const float* __restrict__ y, int64_t ldm, int B, int N, int C, int T,
    const __bf16* __restrict__ Whp, const float* __restrict__ bh, int H, float* __restrict__ out,
    int nsb, int nob, int nfb) {
  constexpr int SBLK = 2 * NW;
  constexpr int WS = SBLK / SW, WO = NW / WS, OW = OB / WO;   // waves along subjects / objects, objects per wave
  static_assert(WS * WO == NW && OW * WO == OB, "wave tiling");
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
  const int ws = wave % WS, wo = wave / WS;      // this wave's subjects SW ws .., objects OW wo ..

  // DMA sources: wave w stages rows 4w .. 4w+3 (2 pieces each); row r < SBLK: subject SBLK sb + r
  // (U half, channels [0,C)), else object OB ob + r - SBLK (V half, channels [C,2C))
  // LDS image of a row: two pieces of 8 frames; inside a piece position = 16 X + slot with
  //   slot = (f & 7) + 8 ((q >> 1) & 1),  X = 2 (q >> 2) + (q & 1)      (q = 16-byte channel quad 0..7)
  // so that (a) one DMA piece fetches 8 complete 128-byte lines of y (8 frames x 32 channels) and
  // (b) the fragment read of lane (f, kg) for quad 2 kg + r sits at slot (f & 7) + 8 (kg & 1): the
  // four 16-lane groups of a ds_read_b128 each cover all 16 slots -- conflict-free.
  // (buffer loads: descriptor = this video's rows of y, a fixed 32-bit lane offset per piece, one scalar offset that
  // advances 128 bytes per k-step -- cheaper to issue beside MFMAs than global_load_lds with eight 64-bit pointers per
  // lane, and eight registers and sixteen vector adds per k-step less; tools/probes/lds_dma_issue_probe.hip)
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(y) + (int64_t)b * N * T * ldm, 0, (int)(unsigned)((int64_t)N * T * ldm * 4), 0x00020000);
  unsigned voff[8];
  {
    const int fq = lane & 7;
    const int q = (lane >> 5) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wave * 4 + (i >> 1), j = i & 1;
      int trk = r < SBLK ? sb * SBLK + r : ob * OB + r - SBLK;
      trk = min(trk, N - 1);
      const int t = min(t0 + 8 * j + fq, T - 1);
      voff[i] = (unsigned)((((int64_t)trk * T + t) * ldm + (r < SBLK ? 0 : C) + 4 * q) * 4);
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Whp), 0, C * 32, 0x00020000);
  int y_soff = 0, w_soff = 0;
  auto stage = [&](int buf) {
    char* dst = smem + buf * ST + wave * 4 * HP_ROW;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)voff[i], y_soff, 0, 0);
    y_soff += HP_KC * 4;
    // head weights of the k-step, [4 kg][16 h][8 ch] bf16 = the packed layout itself; staged through
    // LDS as well so that no register-returning global load (whose wait the compiler would place at the
    // top of the loop, serialising the whole DMA queue with the compute) is left in the loop
    if (wave == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(smem + buf * ST + ROWS * HP_ROW),
                                               16, lane * 16, w_soff, 0, 0);
      w_soff += 1024;
    }
  };

  f32x4 acc[SW][OW];
#pragma unroll
  for (int s = 0; s < SW; ++s)
#pragma unroll
    for (int o = 0; o < OW; ++o) acc[s][o] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    f32x4 u[SW][2];
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      u[s][0] = *reinterpret_cast<const f32x4*>(base + (SW * ws + s) * HP_ROW);
      u[s][1] = *reinterpret_cast<const f32x4*>(base + (SW * ws + s) * HP_ROW + 256);
    }
    // V fragments are read two objects ahead of their use (LDS latency off the critical path).  The
    // reads and their counted waits are written out: left to itself the compiler issues every
    // fragment read right before its first use and waits for it at once (32 exposed LDS round trips
    // per k-step).  LDS returns in order, so "lgkmcnt(n)" = all but the newest n reads have landed.
    const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(base + (SBLK + OW * wo) * HP_ROW);
    f32x4 vq[3][2];
#define TSPN_VREAD(slot, o)                                                                             \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][0]) : "v"(vaddr), "n"((o) * HP_ROW)); \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][1]) : "v"(vaddr), "n"((o) * HP_ROW + 256));
    TSPN_VREAD(0, 0)
    TSPN_VREAD(1, 1)
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      if (o + 2 < OW) {
        TSPN_VREAD((o + 2) % 3, o + 2)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else if (o + 1 < OW) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      }
      const f32x4 v0 = vq[o % 3][0], v1 = vq[o % 3][1];
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const f32x4 a0 = u[s][0] + v0, a1 = u[s][1] + v1;
        u32x4 pk = {relu_pack(a0[0], a0[1]), relu_pack(a0[2], a0[3]), relu_pack(a1[0], a1[1]),
                    relu_pack(a1[2], a1[3])};
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
  for (int s = 0; s < SW; ++s) {
    const int sg = sb * SBLK + SW * ws + s;
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      const int og = ob * OB + OW * wo + o;
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

// ------------------------------------------------------------------------------------------------
// Dense (reference-faithful) bf16 form: DPNHead on a materialised pair tensor.
// x [P, C, T] fp32 (the layout DPNHead.forward is handed, dpn.py:69) -> bf16 channels-last [P, T, C]
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float* __restrict__ x, int C, int T,
                                                                  __bf16* __restrict__ out) {
  __shared__ float tile[32][33];
  const int p = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* xp = x + (int64_t)p * C * T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < C && t < T) ? xp[(int64_t)c * T + t] : 0.f;
  }
  __syncthreads();
  __bf16* op = out + (int64_t)p * T * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (t < T && c < C) op[(int64_t)t * C + c] = (__bf16)tile[tx][ty + 8 * i];
  }
}

// out[p][h][t] = bh[h] + sum_c Wh[h][c] * bf16(relu(y[p][t][c])) on v_mfma_f32_16x16x32_bf16; y fp32 channels-last
// [P, T, ldm] (the conv kernel's output, bias included).  Wave = (pair, 16 frames): lane (f, kq) reads channels
// 8 kq .. 8 kq + 7 of frame f per k-step of 32 channels -- the 16 lanes of a kq group cover 16 full 128-byte lines
// between the four groups.  HBM-bound by construction (y is read once), four k-steps in flight per wave.
__global__ __launch_bounds__(256) void heads_dense_bf16_kernel(
    const float* __restrict__ y, int64_t ldm, int64_t P, int C, int T, const __bf16* __restrict__ Whp,
    const float* __restrict__ bh, int H, float* __restrict__ out, int nfb) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= P * nfb) return;
  const int64_t p = item / nfb;
  const int fb = (int)(item - p * nfb);
  const int f = lane & 15, kq = lane >> 4;
  const int t0 = fb * 16;
  const int t = min(t0 + f, T - 1);
  const float* src = y + (p * T + t) * ldm + 8 * kq;
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(Whp) + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = C / HP_KC;
  int k = 0;
  for (; k + 4 <= nk; k += 4) {
    f32x4 b0[4], b1[4];
    bf16x8 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b0[j] = *reinterpret_cast<const f32x4*>(src + (k + j) * HP_KC);
      b1[j] = *reinterpret_cast<const f32x4*>(src + (k + j) * HP_KC + 4);
      w[j] = wsrc[(k + j) * 64];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 pk = {relu_pack(b0[j][0], b0[j][1]), relu_pack(b0[j][2], b0[j][3]), relu_pack(b1[j][0], b1[j][1]),
                        relu_pack(b1[j][2], b1[j][3])};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], __builtin_bit_cast(bf16x8, pk), acc, 0, 0, 0);
    }
  }
  for (; k < nk; ++k) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + k * HP_KC);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(src + k * HP_KC + 4);
    const u32x4 pk = {relu_pack(b0[0], b0[1]), relu_pack(b0[2], b0[3]), relu_pack(b1[0], b1[1]), relu_pack(b1[2], b1[3])};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsrc[k * 64], __builtin_bit_cast(bf16x8, pk), acc, 0, 0, 0);
  }
  if (t0 + f < T) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 4 * kq + r;
      if (h < H) out[(p * H + h) * T + t0 + f] = acc[r] + bh[h];
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
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24) && ldm < (1 << 24) && 3 * Cin * M * 2 < (1LL << 31),
               TSPN_EUNSUPPORTED, "tspn_conv3_tc_bf16: dimension too large");
  const int64_t ncols = B * T;
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = tspn::ceil_div(ncols, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_tc_bf16: grid too large");
  {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_bf16_big_kernel), G_SMEM, "tspn_conv3_tc_bf16"))
      return rc;
    const int64_t tm = tspn::ceil_div(M, G_BM), tn = tspn::ceil_div(ncols, G_BN);
    hipLaunchKernelGGL(conv3_bf16_big_kernel, dim3((unsigned)(tm * tn)), dim3(G_THREADS), G_SMEM,
                       TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(packed), bias, y, (int)Cin, (int)T, (int)M, ncols,
                       (int)tm, (int)tn, (int)ldm, (int)(3 * Cin * M * 2));
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
  TSPN_REQUIRE(grid < (1LL << 31) && N < (1 << 20) && T < (1 << 24) && C < (1 << 24) && N * T * ldm * 4 < (1LL << 31),
               TSPN_EUNSUPPORTED, "tspn_heads_pairgrid_bf16: problem too large (a video's projections must stay below 2 GB)");
  const size_t smem = 2 * ((size_t)(2 * sblk) * HP_ROW + 1024);
  static tspn::LdsLimit lds[2];
  if (int rc = big ? lds[1].ensure(reinterpret_cast<const void*>(heads_pairgrid_bf16_kernel<8, 16, TSPN_HPB_SW>), smem,
                                   "tspn_heads_pairgrid_bf16")
                   : lds[0].ensure(reinterpret_cast<const void*>(heads_pairgrid_bf16_kernel<4, 8, 2>), smem,
                                   "tspn_heads_pairgrid_bf16"))
    return rc;
  if (big)
    hipLaunchKernelGGL((heads_pairgrid_bf16_kernel<8, 16, TSPN_HPB_SW>), dim3((unsigned)grid), dim3(512), smem,
                       TSPN_STREAM(stream), y, ldm, (int)B, (int)N, (int)C, (int)T,
                       reinterpret_cast<const __bf16*>(head_packed), head_b, (int)H, out, (int)nsb, (int)nsb,
                       (int)nfb);
  else
    hipLaunchKernelGGL((heads_pairgrid_bf16_kernel<4, 8, 2>), dim3((unsigned)grid), dim3(256), smem,
                       TSPN_STREAM(stream), y, ldm, (int)B, (int)N, (int)C, (int)T,
                       reinterpret_cast<const __bf16*>(head_packed), head_b, (int)H, out, (int)nsb, (int)nsb,
                       (int)nfb);
  return tspn::check_launch("tspn_heads_pairgrid_bf16");
}

extern "C" int tspn_transpose_cast_bf16(const float* x, int64_t P, int64_t C, int64_t T, uint16_t* out, void* stream) {
  TSPN_REQUIRE(P >= 0 && C > 0 && T > 0 && C < (1 << 24) && T < (1 << 24) && P < 65536 * 32768LL, TSPN_EINVAL,
               "tspn_transpose_cast_bf16: bad sizes P=%lld C=%lld T=%lld", (long long)P, (long long)C, (long long)T);
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_transpose_cast_bf16: null pointer");
  // gridDim.z <= 65535: pairs in slabs
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = std::min<int64_t>(65535, P - p0);
    hipLaunchKernelGGL(transpose_cast_bf16_kernel,
                       dim3((unsigned)tspn::ceil_div(T, 32), (unsigned)tspn::ceil_div(C, 32), (unsigned)np), dim3(256), 0,
                       TSPN_STREAM(stream), x + p0 * C * T, (int)C, (int)T, reinterpret_cast<__bf16*>(out) + p0 * T * C);
  }
  return tspn::check_launch("tspn_transpose_cast_bf16");
}

extern "C" int tspn_heads_dense_bf16(const float* y, int64_t ldm, int64_t P, int64_t C, int64_t T,
                                     const uint16_t* head_packed, const float* head_b, int64_t H, float* out,
                                     void* stream) {
  TSPN_REQUIRE(P >= 0 && C > 0 && T > 0 && H > 0 && H <= 16 && ldm >= C, TSPN_EINVAL,
               "tspn_heads_dense_bf16: bad sizes P=%lld C=%lld T=%lld H=%lld ldm=%lld", (long long)P, (long long)C,
               (long long)T, (long long)H, (long long)ldm);
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(y && head_packed && head_b && out, TSPN_EINVAL, "tspn_heads_dense_bf16: null pointer");
  TSPN_REQUIRE(C % HP_KC == 0 && ldm % 4 == 0 && aligned16(y) && aligned16(head_packed), TSPN_EUNSUPPORTED,
               "tspn_heads_dense_bf16: needs C %% 32 == 0, ldm %% 4 == 0, 16-byte aligned y / weights");
  const int64_t nfb = tspn::ceil_div(T, 16);
  const int64_t blocks = tspn::ceil_div(P * nfb, 4);
  TSPN_REQUIRE(blocks < (1LL << 31) && C < (1 << 24) && T < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_heads_dense_bf16: problem too large");
  hipLaunchKernelGGL(heads_dense_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, TSPN_STREAM(stream), y, ldm, P,
                     (int)C, (int)T, reinterpret_cast<const __bf16*>(head_packed), head_b, (int)H, out, (int)nfb);
  return tspn::check_launch("tspn_heads_dense_bf16");
}

extern "C" int tspn_temporal_encoder_heads_bf16(const uint16_t* x_tc, int64_t P, int64_t C, int64_t T,
                                                const uint16_t* conv_packed, const float* conv_bias,
                                                const uint16_t* head_packed, const float* head_b, int64_t H,
                                                float* y_ws, float* out_heads, void* stream) {
  TSPN_REQUIRE(P >= 0 && C > 0 && T > 0 && H > 0, TSPN_EINVAL, "tspn_temporal_encoder_heads_bf16: bad sizes");
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(x_tc && conv_packed && head_packed && head_b && y_ws && out_heads, TSPN_EINVAL,
               "tspn_temporal_encoder_heads_bf16: null pointer");
  if (int rc = tspn_conv3_tc_bf16(x_tc, P, T, C, conv_packed, C, conv_bias, y_ws, C, stream)) return rc;
  return tspn_heads_dense_bf16(y_ws, C, P, C, T, head_packed, head_b, H, out_heads, stream);
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
  // the predicate logits depend on the tracklet means only: computed FIRST (as in tspn_forward_fused_f32) so that a
  // caller can decode / gather them on another stream behind ev_logits_ready while the encoder runs
  if ((rc = tspn_temporal_mean_bf16(d->feats, NT, d->T, d->D, fbar, stream))) return rc;
  if ((rc = tspn::pair_predicate(fbar, NT, d->D, d->pairs, P, d->cls_w, d->cls_b, d->K, d->out_logits, ws + L.lin,
                                 L.lin_bytes, stream)))
    return rc;
  if (d->ev_logits_ready) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_logits_ready), s);
  if (d->ev_conv_begin) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_begin), s);
  rc = tspn_conv3_tc_bf16(d->feats, NT, d->T, d->D, d->conv_packed, 2 * C, bias2, y, 2 * C, stream);
  if (d->ev_conv_end) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_end), s);
  if (rc) return rc;
  return tspn_heads_pairgrid_bf16(y, 2 * C, d->B, d->N, C, d->T, d->head_packed, d->head_b, 3 * d->A, d->out_heads,
                                  stream);
}
