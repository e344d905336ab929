// Winograd F(4,3) temporal conv, second structure: weight fragments straight from global memory into
// registers (gfx950, fp32).  Same arithmetic as tspn_wino43.hip (same transforms, same MFMA order of the
// contraction), different data movement:
//
//   * In conv3_wino43_cl_kernel the four waves of a workgroup own disjoint 32-row slices of the weight
//     tile, so the LDS was only a transit buffer for the weights (6 DMA pieces + 24 ds_read_b32 per wave
//     and chunk, and a barrier that had to wait for them).  Here the weights are packed FRAGMENT-MAJOR,
//         Wf[m / 32][chunk = ch / 8][j = 0..5][lane = 32 kh + li][e = 0..3]  =  U_j[8 chunk + 4 kh + e][32 (m/32) + li],
//     so that the A operands of the four k-steps of (chunk, j) are ONE global_load_dwordx4 per lane
//     (a contiguous 1-KiB line per wave, 6 KiB per wave and chunk, one linear stream per wave).
//   * Both operands of chunk c+1 are fetched into the registers of chunk c as soon as those are free
//     (after the MFMAs of their position pair): weights from global memory, V from the LDS tile that was
//     transformed one chunk earlier.  At the top of a chunk every operand is already in registers.
//   * x arrives in SUPER-STAGES of 32 channels (four chunks) with row-major LDS rows [slot][8 groups + pad],
//     so that a DMA piece covers whole 128-byte lines of x (59 consecutive 16-byte units = 6.5 rows) instead
//     of 64 quarter-used lines: five pieces per wave, issued as one burst every fourth chunk, more than a
//     super-stage ahead of their first use (2 buffers).  One bare s_barrier per chunk; every VMEM wait is a
//     counted vmcnt on weight registers (the weight loads of the next chunk stay in flight across the
//     barrier, and the in-order return of VMEM data makes the x pieces land before the weights issued
//     after them are consumed).  LDS: x 2 x 19 KB + V 4 x 7 KB = 66 KB.
//
// Needs M % 32 == 0 and Cin % 8 == 0 (the canonical kernel takes everything else).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int THREADS = 256;
constexpr int BM = 128;
constexpr int QT = 32;                 // quads per workgroup
constexpr int KC = 8;
constexpr int NSLOT = 4 * QT + 2;      // frame rows of the x tile (one halo row on each side)
constexpr int XROW = 9;                // 16-byte units per row of an x super-stage: 8 channel groups + 1 pad
constexpr int XP_UNITS = 59;           // units per DMA piece: 20 pieces (5 per wave) cover 1180 >= NSLOT * XROW = 1170
constexpr int XS_ST = 20 * XP_UNITS * 4 + 32;   // floats per super-stage buffer
constexpr int V_ST = 2 * 7 * QT * 4;   // [2 g][6 j + 1 scratch plane][32 quads][4 ch]
constexpr int NXS = 2, NVS = 4;       // V stage of chunk k: k % 4 (static in the unrolled loop)
constexpr size_t SMEM_BYTES = sizeof(float) * (NXS * XS_ST + NVS * V_ST);

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// canonical [6][Cin][M] -> fragment-major [M/32][Cin/8][6][64][4]
__global__ void repack_wino43_frag_kernel(const float* __restrict__ in, int64_t Cin, int64_t M,
                                          float* __restrict__ out) {
  const int64_t total = 6 * Cin * M;
  const int64_t nch = Cin / KC;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(o & 3);
    const int lane = (int)((o >> 2) & 63);
    const int64_t r = o >> 8;
    const int j = (int)(r % 6);
    const int64_t c = (r / 6) % nch;
    const int64_t mb = r / (6 * nch);
    const int64_t ch = 8 * c + 4 * (lane >> 5) + e;
    const int64_t m = 32 * mb + (lane & 31);
    out[o] = in[((int64_t)j * Cin + ch) * M + m];
  }
}

// The weight loads are inline asm (the compiler does not see them as asynchronous), so every use of
// their destination registers is preceded by one of these counted waits, tied to the registers by "+v".
template <int VM>
__device__ __forceinline__ void wait_a(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}
template <int VM>
__device__ __forceinline__ void wait_vm_lgkm0() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
}
// one 1-KiB fragment line: lane offset in a VGPR, wave-uniform base in SGPRs, immediate line offset
template <int OFF>
__device__ __forceinline__ void load_frag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}

__global__ __launch_bounds__(THREADS, 2) void conv3_wino43r_kernel(
    const float* __restrict__ x, const float* __restrict__ Wf, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int nq, int64_t nquads, int64_t ncols, int tiles_m,
    int tiles_n, int relu, int ldy, int GM, int vec4) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Xs = reinterpret_cast<float*>(smem_raw);
  float* Vs = Xs + NXS * XS_ST;

  // workgroup -> tile: bijective XCD remap, then groups of GM weight panels x all quad tiles
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int group_sz = GM * tiles_n;
  const int group = wg / group_sz;
  const int first_m = group * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int in_group = wg - group * group_sz;
  const int tile_m = first_m + in_group % gm;
#if defined(TSPN_W43R_ROT)   // probe: every XCD sweeps the quad tiles from a different starting tile
  const int tile_n = (in_group / gm + xcd * (tiles_n / 8)) % tiles_n;
#else
  const int tile_n = in_group / gm;
#endif
  const int m0 = tile_m * BM;
  const int64_t Q0 = (int64_t)tile_n * QT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int nchunks = Cin / KC;

  // quad Q -> (tracklet b, quad q in it); row of its first output frame in the flat [B*T] frame space
  auto quad_row = [&](int64_t Q, int& q) -> int64_t {
    const int64_t b = Q / nq;
    q = (int)(Q - b * nq);
    return b * T + 4 * q;
  };
  int q_first;
  const int64_t row0 = quad_row(Q0, q_first) - 1;   // slot s of the x tile <-> frame row row0 + s

  // ---- weight fragment stream of this wave (32 output rows); rows beyond M: re-read block 0, never stored
  const char* abase;            // wave-uniform; advanced by one chunk (6 KiB) per refill round
  const unsigned aoff = lane * 16;
  {
    int mb = (m0 >> 5) + wave;
    mb = mb < (M >> 5) ? mb : 0;
    abase = reinterpret_cast<const char*>(Wf) + (int64_t)mb * nchunks * (6 * 64 * 16);
  }

  // ---- x super-stage DMA.  Unit U of a buffer = (slot = U / 9, group g = U % 9; g == 8 is padding);
  // piece pg = 5 wave + k holds units [59 pg, 59 pg + 59) in lanes 0..58.  Every piece has valid lanes for
  // every super-stage (also a last one with fewer than 8 channel groups), so each wave issues exactly
  // five pieces per burst and the counted waits below are the same for all waves.
  const int ngroups = Cin >> 2;                  // 4-channel groups in a row of x
  const int nsuper = (ngroups + 7) >> 3;         // super-stages (the last may be partial)
  const float* bptr[5];
  int bgrp[5];                                   // channel group of the lane's unit (99 = never valid)
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int U = XP_UNITS * (5 * wave + k) + lane;
    const int slot = U / XROW, g = U - slot * XROW;
    const bool ok = lane < XP_UNITS && slot < NSLOT && g < 8;
    bgrp[k] = ok ? g : 99;
    int64_t n = row0 + slot;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bptr[k] = x + n * Cin + 4 * (ok ? g : 0);
  }
  auto stage_burst = [&](int S) {                // super-stage S -> buffer S & 1
    const int glim = min(8, ngroups - 8 * S);
    float* dst = Xs + (S & 1) * XS_ST + XP_UNITS * 5 * wave * 4;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#if !defined(TSPN_W43R_ABL_NODMA)
      if (bgrp[k] < glim) glds16(bptr[k], dst + XP_UNITS * k * 4);
#endif
#if !defined(TSPN_W43R_PROBE_HOTX)   // probe: re-read the same channels of x (cache-hot bursts, wrong results)
      bptr[k] += 32;
#endif
    }
  };

  // ---- transform item of this thread: quad tk, channel group tg; wave = part (V0 | V5 | V1,V2 | V3,V4).
  // ONE branch-free form for all four parts, so that its VALU can be interleaved with the MFMAs:
  //     s = c0 X0 + c1 X1,  r = c2 X2 + c3 X3,  plane p1 <- s - r,  plane p2 <- s + r
  //   part   X0 X1 X2 X3      c0    c1     c2    c3      p1  p2
  //   V0     d0 d2 d4 d4     4 m0  -5 m2  -m4    0        0   6 (scratch)     V0 = 4 d0 - 5 d2 + d4
  //   V5     d1 d3 d5 d5     4 m1  -5 m3  -m5    0        5   6 (scratch)     V5 = 4 d1 - 5 d3 + d5
  //   V1,V2  d4 d2 d1 d3      m4   -4 m2   4 m1  -m3      1   2               s = d4 - 4 d2, r = 4 d1 - d3
  //   V3,V4  d4 d2 d1 d3      m4   -m2     2 m1  -2 m3    3   4               s = d4 - d2,   r = 2 d1 - 2 d3
  // (m_i = 1 if frame 4q + i - 1 lies inside the tracklet, else 0: the sequence-end masks.)
  const int tk = tid & 31, tg = (tid >> 5) & 1;
  int tslot;
  float tc[4];
  int to[4];           // wave-uniform slot offsets of X0..X3
  int tp1, tp2;        // wave-uniform output planes
  {
    const int64_t Q = Q0 + tk;
    int q = 0;
    const bool okq = Q < nquads;
    const int64_t r = okq ? quad_row(Q, q) : row0 + 1;
    tslot = (int)(r - 1 - row0);
    float tm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = 4 * q + i - 1;
      tm[i] = (okq && t >= 0 && t < T) ? 1.f : 0.f;
    }
    if (wave == 0) {
      to[0] = 0; to[1] = 2; to[2] = 4; to[3] = 4; tp1 = 0; tp2 = 6;
      tc[0] = 4.f * tm[0]; tc[1] = -5.f * tm[2]; tc[2] = -tm[4]; tc[3] = 0.f;
    } else if (wave == 1) {
      to[0] = 1; to[1] = 3; to[2] = 5; to[3] = 5; tp1 = 5; tp2 = 6;
      tc[0] = 4.f * tm[1]; tc[1] = -5.f * tm[3]; tc[2] = -tm[5]; tc[3] = 0.f;
    } else if (wave == 2) {
      to[0] = 4; to[1] = 2; to[2] = 1; to[3] = 3; tp1 = 1; tp2 = 2;
      tc[0] = tm[4]; tc[1] = -4.f * tm[2]; tc[2] = 4.f * tm[1]; tc[3] = -tm[3];
    } else {
      to[0] = 4; to[1] = 2; to[2] = 1; to[3] = 3; tp1 = 3; tp2 = 4;
      tc[0] = tm[4]; tc[1] = -tm[2]; tc[2] = 2.f * tm[1]; tc[3] = -2.f * tm[3];
    }
  }
  // two phases, so that the LDS reads of the x tile fly under the MFMAs that precede the arithmetic.
  // All LDS addresses are per-lane constants (4 read pointers into x buffer 0, 2 write pointers into V
  // stage 0); buffer, channel-group and stage offsets are immediates in the unrolled loop: MFMA and VALU
  // share an issue port, so every address add between two MFMAs can delay the second one.  (The
  // arithmetic is written on float pairs; hipcc unpacks v_pk_* f32 operations that sit in the shadow of
  // an MFMA, so the loop carries 24 plain VALU per chunk for it.)
  const float* txp[4];
  float* tvp[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) txp[i] = Xs + ((tslot + to[i]) * XROW + tg) * 4;
  tvp[0] = Vs + (tg * 7 * QT + tk + tp1 * QT) * 4;
  tvp[1] = Vs + (tg * 7 * QT + tk + tp2 * QT) * 4;
  f32x2 tc2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) tc2[i] = f32x2{tc[i], tc[i]};
  f32x4 td[4];
  auto transform_read = [&](int c2) {           // x of chunk c2: super-stage c2 / 4, channel groups 2 (c2 % 4) + tg
#if defined(TSPN_W43R_ABL_NOXFORM)
    return;
#endif
    const int off = ((c2 >> 2) & 1) * XS_ST + 8 * (c2 & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) td[i] = *reinterpret_cast<const f32x4*>(txp[i] + off);
  };
  auto transform_write = [&](int vst) {
#if defined(TSPN_W43R_ABL_NOXFORM)
    return;
#endif
    auto lo = [](const f32x4& q) { return f32x2{q[0], q[1]}; };
    auto hi = [](const f32x4& q) { return f32x2{q[2], q[3]}; };
    const f32x2 s0 = __builtin_elementwise_fma(tc2[0], lo(td[0]), tc2[1] * lo(td[1]));
    const f32x2 s1 = __builtin_elementwise_fma(tc2[0], hi(td[0]), tc2[1] * hi(td[1]));
    const f32x2 r0 = __builtin_elementwise_fma(tc2[2], lo(td[2]), tc2[3] * lo(td[3]));
    const f32x2 r1 = __builtin_elementwise_fma(tc2[2], hi(td[2]), tc2[3] * hi(td[3]));
    const f32x2 d0 = s0 - r0, d1 = s1 - r1, p0 = s0 + r0, p1 = s1 + r1;
    *reinterpret_cast<f32x4*>(tvp[0] + vst * V_ST) = f32x4{d0[0], d0[1], d1[0], d1[1]};
    *reinterpret_cast<f32x4*>(tvp[1] + vst * V_ST) = f32x4{p0[0], p0[1], p1[0], p1[1]};
  };
  auto transform = [&](int c2, int vst) { transform_read(c2); transform_write(vst); };

  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  f32x4 a[6], v[6];
  const int voff = (kh * 7 * QT + li) * 4;
  auto load_v = [&](int vst, int j) {
    v[j] = *reinterpret_cast<const f32x4*>(Vs + vst * V_ST + voff + j * QT * 4);
  };
  auto load_a_pair = [&](auto jp_tag) {     // positions 2 jp, 2 jp + 1 of the chunk at abase
    constexpr int JP = decltype(jp_tag)::value;
#if !defined(TSPN_W43R_ABL_NOALOAD)
    if (JP == 0) { load_frag<0>(a[0], aoff, abase); load_frag<1024>(a[1], aoff, abase); }
    if (JP == 1) { load_frag<2048>(a[2], aoff, abase); load_frag<3072>(a[3], aoff, abase); }
    if (JP == 2) { load_frag<0>(a[4], aoff, abase + 4096); load_frag<1024>(a[5], aoff, abase + 4096); }
#endif
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  auto mfma_pair = [&](int ja, int jb) {
#if defined(TSPN_W43R_SETPRIO)
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[ja] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ja][e], v[ja][e], acc[ja], 0, 0, 0);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jb][e], v[jb][e], acc[jb], 0, 0, 0);
    }
#if defined(TSPN_W43R_SETPRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // ---- prologue: x super-stages 0 and 1 landed, V_0 in registers, V_1 in LDS, A_0 in flight
  stage_burst(0);
  if (nsuper > 1) stage_burst(1);
  __syncthreads();
  transform(0, 0);
  if (nchunks > 1) transform(1, 1);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 6; ++j) load_v(0, j);
  wait_vm_lgkm0<0>();
  __syncthreads();
  // the weights of chunk 0 are the youngest VMEM operations, in the order the waits of a chunk expect
  load_a_pair(P0{}); load_a_pair(P1{}); load_a_pair(P2{});
  abase += 6 * 1024;
  __builtin_amdgcn_sched_barrier(0);

  // chunk c: MFMAs on the registers (A_c, V_c); meanwhile transform x_{c+2} -> V_{c+2}, refill the
  // registers with (A_{c+1}, V_{c+1}) position pair by position pair, and -- in the third chunk of a
  // super-stage S -- issue the DMA burst of super-stage S+2 into the buffer S has just left (its last
  // read was the transform of chunk 4S+3 during chunk 4S+1).  V stage of chunk k: k % 4.
  // `cc` = c modulo 8 (a compile-time constant in the unrolled loop: every LDS offset is an immediate);
  // `S` = c / 4 is only read by the burst.
  // VMEM issue order of a chunk: [x pieces: ND] a0 a1 | a2 a3 | a4 a5; vmcnt counts are the number of
  // YOUNGER operations at each wait.  The burst is older than the weight loads of its chunk, which the
  // next chunk consumes: it has landed long before its first reader (4 chunks later) without a wait of
  // its own.
  auto chunk_body = [&](auto cc_tag, int S, auto has1_tag, auto has2_tag, auto burst_tag) {
    const int cc = cc_tag;                             // c % 8 (integral_constant or a runtime int)
    constexpr bool HAS1 = decltype(has1_tag)::value;   // chunk c+1 exists: refill
    constexpr bool HAS2 = decltype(has2_tag)::value;   // chunk c+2 exists: transform
    constexpr bool BURST = decltype(burst_tag)::value; // c = 4S+2 and super-stage S+2 exists
    constexpr int ND = BURST ? 5 : 0, NA = HAS1 ? 2 : 0;
    const int s1 = (cc + 1) & 3, s2 = (cc + 2) & 3;
    wait_a<4>(a[0], a[1]);
    if (BURST) stage_burst(S + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (HAS2) transform_read(cc + 2);
    mfma_pair(0, 1);
    if (HAS1) { load_a_pair(P0{}); load_v(s1, 0); load_v(s1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<2 + ND + NA>(a[2], a[3]);
    if (HAS2) transform_write(s2);
    mfma_pair(2, 3);
    if (HAS1) { load_a_pair(P1{}); load_v(s1, 2); load_v(s1, 3); }
#if !defined(TSPN_W43R_NOINTERLEAVE)
    if (HAS2) {   // the transform's arithmetic under the MFMAs of this pair: 1 MFMA : 2 VALU, the stores last
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    wait_a<ND + 2 * NA>(a[4], a[5]);
    mfma_pair(4, 5);
    if (HAS1) { load_a_pair(P2{}); load_v(s1, 4); load_v(s1, 5); }
#if !defined(TSPN_W43R_PROBE_HOTA)   // probe: re-read the same 6 KiB of weights (cache-hot stream, wrong results)
    if (HAS1) abase += 6 * 1024;
#endif
    __builtin_amdgcn_sched_barrier(0);
#if defined(TSPN_W43R_ABL_NOALOAD)
    wait_vm_lgkm0<0>();
#else
    // LDS operations return in order: behind the transform's ds_writes only the four refill reads of
    // positions 2..5 were issued, and those may stay in flight across the barrier (their stage is not
    // rewritten before the barrier after next)
    if (HAS1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if !defined(TSPN_W43R_ABL_NOBARRIER)    // timing probe only
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    using TT = std::true_type;
    using FF = std::false_type;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>;
    using C5 = std::integral_constant<int, 5>;
    using C6 = std::integral_constant<int, 6>;
    using C7 = std::integral_constant<int, 7>;
    int S = 0;
    // steady super-stages (all four chunks have successors, super-stage S+2 exists), two per iteration
    for (; S + 3 < nsuper; S += 2) {
      chunk_body(C0{}, S, TT{}, TT{}, FF{});
      chunk_body(C1{}, S, TT{}, TT{}, FF{});
      chunk_body(C2{}, S, TT{}, TT{}, TT{});
      chunk_body(C3{}, S, TT{}, TT{}, FF{});
      chunk_body(C4{}, S + 1, TT{}, TT{}, FF{});
      chunk_body(C5{}, S + 1, TT{}, TT{}, FF{});
      chunk_body(C6{}, S + 1, TT{}, TT{}, TT{});
      chunk_body(C7{}, S + 1, TT{}, TT{}, FF{});
    }
    if (S + 2 < nsuper) {      // one more steady super-stage (S is even here)
      chunk_body(C0{}, S, TT{}, TT{}, FF{});
      chunk_body(C1{}, S, TT{}, TT{}, FF{});
      chunk_body(C2{}, S, TT{}, TT{}, TT{});
      chunk_body(C3{}, S, TT{}, TT{}, FF{});
      ++S;
    }
    int c = 4 * S;             // the last (up to) eight chunks: no bursts left, offsets computed at run time
    for (; c + 2 < nchunks; ++c) chunk_body(c & 7, c >> 2, TT{}, TT{}, FF{});
    if (c + 1 < nchunks) {
      chunk_body(c & 7, c >> 2, TT{}, FF{}, FF{});
      ++c;
    }
    chunk_body(c & 7, c >> 2, FF{}, FF{}, FF{});
  }

  // ---- output transform + store: lane column = quad -> frames 4q .. 4q+3
  {
    const int64_t Q = Q0 + li;
    if (Q < nquads) {
      int q;
      const int64_t r = quad_row(Q, q);
      const int64_t b = (r - 4 * q) / T;
      const int t = 4 * q;
      float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          const float p12 = acc[1][e] + acc[2][e], m12 = acc[1][e] - acc[2][e];
          const float p34 = acc[3][e] + acc[4][e], m34 = acc[3][e] - acc[4][e];
          float o0 = acc[0][e] + p12 + p34;
          float o1 = m12 + 2.f * m34;
          float o2 = p12 + 4.f * p34;
          float o3 = m12 + 8.f * m34 + acc[5][e];
          if (bias != nullptr) {
            const float bb = bias[m];
            o0 += bb; o1 += bb; o2 += bb; o3 += bb;
          }
          if (relu) {
            o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f);
          }
          float* dst = ycol + (int64_t)m * ldy;
          if (vec4) {   // rows padded to >= 4 nq frames and 16-byte aligned: frames >= T land in the padding
            *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
          } else {
            dst[0] = o0;
            if (t + 1 < T) dst[1] = o1;
            if (t + 2 < T) dst[2] = o2;
            if (t + 3 < T) dst[3] = o3;
          }
        }
      }
    }
  }
}

}  // namespace

bool tspn::wino43_frag_supported(int64_t Cin, int64_t M) { return Cin > 0 && M > 0 && Cin % KC == 0 && M % 32 == 0; }

extern "C" int tspn_repack_wino43_frag_f32(const float* packed6, int64_t Cin, int64_t M, float* frag,
                                           void* stream) {
  TSPN_REQUIRE(packed6 && frag, TSPN_EINVAL, "tspn_repack_wino43_frag_f32: null pointer");
  TSPN_REQUIRE(packed6 != frag, TSPN_EINVAL, "tspn_repack_wino43_frag_f32: in-place repack is not possible");
  TSPN_REQUIRE(tspn::wino43_frag_supported(Cin, M), TSPN_EUNSUPPORTED,
               "tspn_repack_wino43_frag_f32: needs Cin %% 8 == 0 and M %% 32 == 0 (Cin=%lld M=%lld)",
               (long long)Cin, (long long)M);
  const int64_t total = 6 * Cin * M;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(repack_wino43_frag_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), packed6, Cin, M,
                     frag);
  return tspn::check_launch("tspn_repack_wino43_frag_f32");
}

int tspn::conv3_tc_wino43r(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                           const float* bias, int relu, float* y, int64_t ldy, void* stream) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0 && ldy >= T && ldy < (1 << 24), TSPN_EINVAL,
               "tspn_conv3_tc_wino43r_f32: bad sizes B=%lld T=%lld Cin=%lld M=%lld ldy=%lld", (long long)B,
               (long long)T, (long long)Cin, (long long)M, (long long)ldy);
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && y, TSPN_EINVAL, "tspn_conv3_tc_wino43r_f32: null pointer");
  TSPN_REQUIRE(tspn::wino43_frag_supported(Cin, M), TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_wino43r_f32: needs Cin %% 8 == 0, M %% 32 == 0 (Cin=%lld M=%lld)", (long long)Cin,
               (long long)M);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(frag) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y) & 3) == 0,
               TSPN_EUNSUPPORTED, "tspn_conv3_tc_wino43r_f32: x/frag must be 16-byte aligned");
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_wino43r_f32: dimension too large");
  const int64_t nq = tspn::ceil_div(T, 4);
  const int64_t nquads = B * nq;
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = tspn::ceil_div(nquads, QT);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_tc_wino43r_f32: grid too large");
  const int vec4 = (ldy % 4 == 0) && (ldy >= 4 * nq) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  static tspn::LdsLimit lds;   // 66.7 KB of dynamic LDS: above the 64 KB default limit
  if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_wino43r_kernel), SMEM_BYTES,
                          "tspn_conv3_tc_wino43r_f32"))
    return rc;
  const int gm_tiles = tspn::kWinoPanelGroup;
  hipLaunchKernelGGL(conv3_wino43r_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), SMEM_BYTES,
                     TSPN_STREAM(stream), x, frag, bias, y, (int)Cin, (int)T, (int)M, (int)nq, nquads, B * T,
                     (int)tiles_m, (int)tiles_n, relu, (int)ldy, gm_tiles, vec4);
  return tspn::check_launch("tspn_conv3_tc_wino43r_f32");
}

extern "C" int tspn_conv3_tc_wino43r_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag,
                                         int64_t M, const float* bias, int relu, float* y, void* stream) {
  return tspn::conv3_tc_wino43r(x, B, T, Cin, frag, M, bias, relu, y, T, stream);
}
