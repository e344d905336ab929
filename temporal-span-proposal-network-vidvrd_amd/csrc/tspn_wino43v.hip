// Winograd F(4,3) temporal conv, third structure: the input transform runs in its own HBM-bound pass and the
// MFMA kernel carries (almost) no VALU (gfx950, fp32).
//
// Why (measured, tools/probes/mfma_valu_overlap_probe.hip, profiles/r2/mfma_valu_overlap.txt): on gfx950 an
// fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4 / 4x4x1_16B) and ANY VALU instruction of the co-resident
// wave do NOT overlap on a SIMD — fp32 MFMA runs at exactly the vector FMA rate and a mixed (MFMA wave +
// VALU wave) SIMD takes the SUM of the two times.  conv3_wino43r_kernel issues 24 VALU (the input
// transform V = B^T d, once per workgroup) per 24 MFMAs of a chunk: 24 x ~4.3 of 1536 + 103 cycles = 6.3 %
// of the kernel, more than its barrier and all its waits together.  Here
//   * wino43_input_transform_kernel writes V once per launch in the layout the MFMA kernel's B-operand
//     reads want:  Vg[cg = ch / 4][j = 0..5][Q = tracklet * nq + quad (padded to 32)][ch % 4]
//     (0.63 GB of x in, 0.96 GB of V out at 16 videos of config 2: 0.3 ms at HBM speed).  Same formulas, same
//     operation order as the in-kernel transform of tspn_wino43r.hip / tspn_wino43.hip: V is bit-identical.
//   * conv3_wino43v_kernel stages V by LDS-DMA in super-stages of 32 channels (four 8-channel chunks):
//     a (channel group, position) row of the tile is 32 quads x 16 B = 512 contiguous bytes of Vg, a DMA
//     piece (1 KiB per wave-instruction) is two such rows, six pieces per wave and super-stage, issued as
//     one burst TWO super-stages ahead (ring of 3 buffers of 2 x 24 KB = 144 KB: one workgroup per CU).  In-order VMEM
//     return makes every piece land before younger weight loads are consumed, so the burst needs no wait
//     of its own, and ONE bare s_barrier per super-stage (not per chunk) orders the ring.
//   * weights: fragment-major, one global_load_dwordx4 per lane and (chunk, position) straight into the MFMA
//     operand registers, refilled position pair by position pair, counted vmcnt (as in tspn_wino43r.hip).
//   * same MFMA order of the contraction as the other two F(4,3) kernels: results are BIT-IDENTICAL.
// Needs Cin % 32 == 0 and M % 32 == 0 (everything else: tspn_wino43r.hip / tspn_wino43.hip).
#include <algorithm>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup shape (build-time, -DTSPN_W43V_WAVES=4|8, -DTSPN_W43V_ROWSPLIT; measured in profiles/r2/conv_traffic_sweep.md):
//   8 waves, tile 128 rows x 64 quads (SHIPPED): waves w and w + 4 — the two waves of one SIMD — take the SAME 32
//     weight rows on the two 32-quad halves of the tile, so the second wave's weight loads hit the CU's L1: the
//     weight stream from L2 halves, and so does the fabric traffic (FETCH_SIZE 17.1 M KiB per launch against 30.6
//     for the 4-wave form and 18.0 for conv3_wino43r_kernel), 3 % faster than either.  One workgroup per CU
//     (144 KB of LDS), both waves of a SIMD in the same workgroup: hence one barrier per super-stage, not per chunk.
//   4 waves, tile 128 rows x 32 quads, two workgroups per CU (72 KB each): the round-1 tiling.
//   8 waves + ROWSPLIT, tile 256 rows x 32 quads: shares the V tile instead of the weights; no gain (V is the small stream).
#ifndef TSPN_W43V_WAVES
#define TSPN_W43V_WAVES 8
#endif
constexpr int NW = TSPN_W43V_WAVES;
#if TSPN_W43V_WAVES == 8 && !defined(TSPN_W43V_ROWSPLIT)
constexpr int QH = 2;                     // 32-quad halves per workgroup
#else
constexpr int QH = 1;
#endif
constexpr int THREADS = 64 * NW;
constexpr int BM = 32 * NW / QH;
constexpr int QT = 32;                    // quads per wave (and per V half-tile)
constexpr int QWG = QT * QH;              // quads per workgroup
constexpr int KC = 8;                     // channels per chunk
constexpr int VROW = QT * 4;              // floats per (channel group, position) row: 512 B
constexpr int VCH = 12 * VROW;            // floats per chunk: [2 g][6 j][32 quads][4 ch] = 6 KB
constexpr int VSS = 4 * VCH * QH;         // floats per super-stage (4 chunks, QH half-tiles): 24 KB x QH
constexpr int NVB = 3;                    // ring of super-stage buffers
constexpr int NPIECE = 24 * QH / NW;      // DMA pieces (1 KiB) per wave and super-stage
constexpr size_t SMEM_BYTES = sizeof(float) * NVB * VSS;
static_assert(NW == 4 || NW == 8, "TSPN_W43V_WAVES must be 4 or 8");
// Weight panels (128 rows) per tile group: a group sweeps all quad tiles, so V is re-read from beyond L2 once
// per group and a panel's weights once per 32 concurrent workgroups of an XCD.  Measured for the shipped shape
// at 16 videos of config 2: 1 / 2 / 3 / 4 / 5 / 6 / 8 panels -> 30.3 / 30.2 / 30.0 / 29.2-29.6 / 30.2 / 30.1 / 30.3 ms and
// FETCH_SIZE 32.7 / 16.6 / 16.6 / 17.1 / 16.1 / 17.0 / 19.9 M KiB.  Build-time knob (-DTSPN_WINO_GM=n).
#if TSPN_WINO_GM != 2
constexpr int kPanelGroupV = TSPN_WINO_GM;
#else
constexpr int kPanelGroupV = QH == 2 ? 4 : 3;
#endif

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Input transform V = B^T d for F(4,3), six positions per quad of four output frames:
//   d_i = x[tracklet][4q + i - 1][ch], i = 0..5, with m_i = 1 inside the tracklet and 0 outside (conv padding)
//   V0 = 4 d0 - 5 d2 + d4                 V5 = 4 d1 - 5 d3 + d5
//   V1 = (d4 - 4 d2) - (4 d1 - d3)        V2 = (d4 - 4 d2) + (4 d1 - d3)
//   V3 = (d4 - d2) - (2 d1 - 2 d3)        V4 = (d4 - d2) + (2 d1 - 2 d3)
// written as  s = fma(c0, X0, c1 * X1), r = fma(c2, X2, c3 * X3), s -/+ r  with the masks folded into the
// coefficients — operation for operation what tspn_wino43r.hip / tspn_wino43.hip do in their main loop.
// A wave = 8 quads x 8 channel groups (loads: whole 128-byte lines of x; stores: 128-byte runs of Vg).
__global__ __launch_bounds__(256) void wino43_input_transform_kernel(
    const float* __restrict__ x, float* __restrict__ Vg, int T, int Cin, int nq, int64_t nquads,
    int64_t nqp, int64_t ncols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cgl = lane & 7, ql = lane >> 3;
  const int64_t Q = ((int64_t)blockIdx.x * 4 + wave) * 8 + ql;     // < nqp by construction of the grid
  const int cg = blockIdx.y * 8 + cgl;
  if (4 * cg >= Cin) return;
  const bool okq = Q < nquads;
  const int64_t b = okq ? Q / nq : 0;
  const int q = okq ? (int)(Q - b * nq) : 0;
  float m[6];
  f32x4 d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int t = 4 * q + i - 1;
    m[i] = (okq && t >= 0 && t < T) ? 1.f : 0.f;
    int64_t n = b * T + t;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    d[i] = *reinterpret_cast<const f32x4*>(x + n * Cin + 4 * cg);
  }
  auto sr = [&](float c0, const f32x4& X0, float c1, const f32x4& X1) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(c0, X0[e], c1 * X1[e]);
    return o;
  };
  f32x4 V[6];
  {
    const f32x4 s = sr(4.f * m[0], d[0], -5.f * m[2], d[2]), r = sr(-m[4], d[4], 0.f, d[4]);
    V[0] = s - r;
  }
  {
    const f32x4 s = sr(4.f * m[1], d[1], -5.f * m[3], d[3]), r = sr(-m[5], d[5], 0.f, d[5]);
    V[5] = s - r;
  }
  {
    const f32x4 s = sr(m[4], d[4], -4.f * m[2], d[2]), r = sr(4.f * m[1], d[1], -m[3], d[3]);
    V[1] = s - r;
    V[2] = s + r;
  }
  {
    const f32x4 s = sr(m[4], d[4], -m[2], d[2]), r = sr(2.f * m[1], d[1], -2.f * m[3], d[3]);
    V[3] = s - r;
    V[4] = s + r;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j)
    *reinterpret_cast<f32x4*>(Vg + (((int64_t)cg * 6 + j) * nqp + Q) * 4) = V[j];
}

// The weight loads are inline asm (the compiler does not see them as asynchronous), so every use of
// their destination registers is preceded by one of these counted waits, tied to the registers by "+v".
template <int VM>
__device__ __forceinline__ void wait_a(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}
template <int OFF>
__device__ __forceinline__ void load_frag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}

__global__ __launch_bounds__(THREADS, 2) void conv3_wino43v_kernel(
    const float* __restrict__ Vg, const float* __restrict__ Wf, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int nq, int64_t nquads, int64_t nqp, int tiles_m,
    int tiles_n, int relu, int ldy, int GM, int vec4) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Vs = reinterpret_cast<float*>(smem_raw);

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
#if defined(TSPN_W43V_ROT)   // probe: every XCD sweeps the quad tiles from a different starting tile
  const int tile_n = (in_group / gm + xcd * (tiles_n / 8)) % tiles_n;
#else
  const int tile_n = in_group / gm;
#endif
  const int m0 = tile_m * BM;
  const int64_t Q0 = (int64_t)tile_n * QWG;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = QH == 2 ? (wave & 3) : wave;  // 32-row block of the tile
  const int wq = QH == 2 ? (wave >> 2) : 0;      // 32-quad half of the tile
  const int li = lane & 31, kh = lane >> 5;
  const int nsuper = Cin >> 5;                  // super-stages of 32 channels (Cin % 32 == 0)

  // ---- weight fragment stream of this wave (32 output rows); rows beyond M: re-read block 0, never stored
  const char* abase;            // wave-uniform; advanced by one chunk (6 KiB) per refill round
  const unsigned aoff = lane * 16;
  {
    int mb = (m0 >> 5) + wrow;
    mb = mb < (M >> 5) ? mb : 0;
    abase = reinterpret_cast<const char*>(Wf) + (int64_t)mb * (Cin / KC) * (6 * 64 * 16);
  }

  // ---- V super-stage DMA: piece p = 6 wave + k holds tile rows 2p (lanes 0..31) and 2p + 1 (lanes 32..63);
  // row rr = chunk-in-super-stage * 12 + g * 6 + j  <->  Vg[(8 S + 2 cl + g)][j][Q0 .. Q0 + 31][0..3]
  const float* vsrc[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int pp = NPIECE * wave + k;              // piece of the workgroup: half-tile pp / 24, rows 2 (pp % 24) + kh
    const int rr = 2 * (pp % 24) + kh;
    const int cl = rr / 12, rem = rr - cl * 12;
    const int g = rem / 6, j = rem - g * 6;
    vsrc[k] = Vg + (((int64_t)(2 * cl + g) * 6 + j) * nqp + Q0 + QT * (pp / 24) + li) * 4;
  }
  const int64_t super_step = (int64_t)8 * 6 * nqp * 4;     // floats between super-stages
  auto stage_burst = [&](int S) {                // super-stage S -> ring buffer S % 3
    float* dst = Vs + (S % NVB) * VSS + NPIECE * wave * 256;
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
#if !defined(TSPN_W43V_ABL_NODMA)
      glds16(vsrc[k], dst + k * 256);
#endif
#if !defined(TSPN_W43V_PROBE_HOTV)   // probe: re-read the same super-stage of V (cache-hot bursts, wrong results)
      vsrc[k] += super_step;
#endif
    }
  };

  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // Weight registers: TWO sets.  Chunk c multiplies out of set c & 1; as soon as a position pair of chunk c
  // has issued its MFMAs, the same registers are refilled with that pair of chunk c + 2, so a weight line has
  // two chunk times (about 1.8 us at two waves per SIMD) to arrive: an L2 miss served by the Infinity Cache or
  // HBM no longer parks the wave (one chunk ahead, as in tspn_wino43r.hip, the waves were parked 11 % of
  // their time on these waits).  V comes from LDS and stays one chunk ahead.
  f32x4 a[2][6], v[6];
  const float* vlane = Vs + wq * (4 * VCH) + (kh * 6 * QT + li) * 4;   // + buffer + chunk + position offsets
  auto load_v = [&](const float* vbuf, int cl, int j) {
#if !defined(TSPN_W43V_ABL_NOVLOAD)
    v[j] = *reinterpret_cast<const f32x4*>(vbuf + cl * VCH + j * VROW);
#endif
  };
  auto load_a_pair = [&](int set, auto jp_tag, const char* base) {     // positions 2 jp, 2 jp + 1 of the chunk at base
    constexpr int JP = decltype(jp_tag)::value;
#if defined(TSPN_W43V_ABL_NOALOAD)
    return;
#endif
    if (JP == 0) { load_frag<0>(a[set][0], aoff, base); load_frag<1024>(a[set][1], aoff, base); }
    if (JP == 1) { load_frag<2048>(a[set][2], aoff, base); load_frag<3072>(a[set][3], aoff, base); }
    if (JP == 2) { load_frag<0>(a[set][4], aoff, base + 4096); load_frag<1024>(a[set][5], aoff, base + 4096); }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  auto mfma_pair = [&](int set, int ja, int jb) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[ja] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][ja][e], v[ja][e], acc[ja], 0, 0, 0);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][jb][e], v[jb][e], acc[jb], 0, 0, 0);
    }
  };

  // ---- prologue: super-stages 0 and 1 landed, V of chunk 0 in registers, the weights of chunks 0 and 1 in flight
  const int nchunks = Cin / KC;                  // >= 4
  stage_burst(0);
  if (nsuper > 1) stage_burst(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 6; ++j) load_v(vlane, 0, j);
  // the weights of chunks 0 and 1 are the youngest VMEM operations, in the order the waits of a chunk expect
  load_a_pair(0, P0{}, abase); load_a_pair(0, P1{}, abase); load_a_pair(0, P2{}, abase);
  load_a_pair(1, P0{}, abase + 6 * 1024); load_a_pair(1, P1{}, abase + 6 * 1024); load_a_pair(1, P2{}, abase + 6 * 1024);
  abase += 2 * 6 * 1024;                         // abase -> weights of chunk c + 2 at the top of chunk c
  __builtin_amdgcn_sched_barrier(0);

  // chunk c = 4 S + cl: MFMAs on the registers (A_c in set c & 1, V_c); meanwhile refill set c & 1 with A_{c+2}
  // and v[] with V_{c+1}, position pair by position pair, and — in the first chunk of super-stage S — issue
  // the DMA burst of super-stage S + 2 into the ring buffer that S - 1 left at the barrier just passed.
  // VMEM issue order:  chunk c-2: [B] A01(c) A23(c) A45(c) | chunk c-1: [B] A01(c+1) A23(c+1) A45(c+1) |
  // chunk c: wait A01(c); [B]; mfma; A01(c+2); wait A23(c); mfma; A23(c+2); wait A45(c); mfma; A45(c+2).
  // The vmcnt of a wait = the number of YOUNGER operations: 4 (or 2, 0) of chunk c itself, the 6 of chunk c+1,
  // what chunk c has issued so far, and the pieces of a burst issued in chunk c-1 or earlier in chunk c.
  // A burst is older than weight loads that are waited for within the next two chunks: it has landed for this
  // wave long before the barrier that ends its super-stage — a whole super-stage before its first reader.
  auto chunk_body = [&](auto cl_tag, const float* vcur, const float* vnext, int S, auto has1_tag, auto has2_tag,
                        auto burst_tag, auto pburst_tag) {
    constexpr int CL = decltype(cl_tag)::value;          // chunk in the super-stage; set = CL & 1 (4 is even)
    constexpr bool HAS1 = decltype(has1_tag)::value;     // chunk c+1 exists: its weights are in flight, refill v[]
    constexpr bool HAS2 = decltype(has2_tag)::value;     // chunk c+2 exists: refill the weight set
    constexpr bool BURST = decltype(burst_tag)::value;   // CL == 0 and super-stage S+2 exists
    constexpr bool PBURST = decltype(pburst_tag)::value; // the previous chunk issued a burst (CL == 1)
    constexpr int SET = CL & 1;
    constexpr int N1 = HAS1 ? 6 : 0, NA = HAS2 ? 2 : 0, NDC = BURST ? NPIECE : 0, NDP = PBURST ? NPIECE : 0;
    const float* vn = CL == 3 ? vnext : vcur;            // where V of chunk c+1 lives
    constexpr int NCL = (CL + 1) & 3;
    wait_a<4 + N1 + NDP>(a[SET][0], a[SET][1]);
    if (BURST) stage_burst(S + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_pair(SET, 0, 1);
    if (HAS2) load_a_pair(SET, P0{}, abase);
    if (HAS1) { load_v(vn, NCL, 0); load_v(vn, NCL, 1); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<2 + N1 + NDP + NDC + NA>(a[SET][2], a[SET][3]);
    mfma_pair(SET, 2, 3);
    if (HAS2) load_a_pair(SET, P1{}, abase);
    if (HAS1) { load_v(vn, NCL, 2); load_v(vn, NCL, 3); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<N1 + NDP + NDC + 2 * NA>(a[SET][4], a[SET][5]);
    mfma_pair(SET, 4, 5);
#if defined(TSPN_W43V_PROBE_HOTA)     // probe: re-read the same 12 KiB of weights (cache-hot stream, wrong results)
    if (HAS2) load_a_pair(SET, P2{}, abase);
#else
    if (HAS2) { load_a_pair(SET, P2{}, abase); abase += 6 * 1024; }
#endif
    if (HAS1) { load_v(vn, NCL, 4); load_v(vn, NCL, 5); }
    __builtin_amdgcn_sched_barrier(0);
    if (CL == 3 && HAS1) {
      // end of a super-stage: every wave has (implicitly) waited for its pieces of super-stage S + 1 and
      // has issued its last reads of buffer S % 3; the six refill reads above target buffer (S + 1) % 3,
      // whose pieces landed before the PREVIOUS barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(TSPN_W43V_ABL_NOBARRIER)
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    using TT = std::true_type;
    using FF = std::false_type;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    int S = 0, buf = 0;
    const float* vcur = vlane;
    auto next_of = [&](int bcur) { return vlane + ((bcur + 1) % NVB) * VSS; };
    for (; S + 2 < nsuper; ++S) {            // steady super-stages: burst S + 2, all chunks have two successors
      const float* vnext = next_of(buf);
      chunk_body(C0{}, vcur, vnext, S, TT{}, TT{}, TT{}, FF{});
      chunk_body(C1{}, vcur, vnext, S, TT{}, TT{}, FF{}, TT{});
      chunk_body(C2{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      chunk_body(C3{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      vcur = vnext;
      buf = (buf + 1) % NVB;
    }
    if (S + 1 < nsuper) {                    // second to last super-stage: nothing left to stage
      const float* vnext = next_of(buf);
      chunk_body(C0{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      chunk_body(C1{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      chunk_body(C2{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      chunk_body(C3{}, vcur, vnext, S, TT{}, TT{}, FF{}, FF{});
      vcur = vnext;
      buf = (buf + 1) % NVB;
      ++S;
    }
    (void)nchunks;
    chunk_body(C0{}, vcur, vcur, S, TT{}, TT{}, FF{}, FF{});   // last super-stage: the weight stream ends
    chunk_body(C1{}, vcur, vcur, S, TT{}, TT{}, FF{}, FF{});
    chunk_body(C2{}, vcur, vcur, S, TT{}, FF{}, FF{}, FF{});
    chunk_body(C3{}, vcur, vcur, S, FF{}, FF{}, FF{}, FF{});
  }

  // ---- output transform + store: lane column = quad -> frames 4q .. 4q+3
  {
    const int64_t Q = Q0 + QT * wq + li;
    if (Q < nquads) {
      const int64_t b = Q / nq;
      const int q = (int)(Q - b * nq);
      const int t = 4 * q;
      float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wrow * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
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

int64_t padded_quads(int64_t B, int64_t T) { return tspn::ceil_div(B * tspn::ceil_div(T, 4), QWG) * QWG; }

}  // namespace

bool tspn::wino43v_supported(int64_t Cin, int64_t M) { return Cin > 0 && M > 0 && Cin % 32 == 0 && M % 32 == 0; }

size_t tspn::wino43v_workspace_bytes(int64_t B, int64_t T, int64_t Cin) {
  if (B <= 0 || T <= 0 || Cin <= 0) return 0;
  return (size_t)(Cin / 4) * 6 * (size_t)padded_quads(B, T) * 4 * sizeof(float);
}

extern "C" size_t tspn_conv3_tc_wino43v_workspace_bytes(int64_t B, int64_t T, int64_t Cin) {
  return tspn::wino43v_workspace_bytes(B, T, Cin);
}

namespace {
int check_common(const char* what, int64_t B, int64_t T, int64_t Cin, int64_t M, int64_t ldy) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0 && ldy >= T && ldy < (1 << 24), TSPN_EINVAL,
               "%s: bad sizes B=%lld T=%lld Cin=%lld M=%lld ldy=%lld", what, (long long)B, (long long)T,
               (long long)Cin, (long long)M, (long long)ldy);
  TSPN_REQUIRE(tspn::wino43v_supported(Cin, M), TSPN_EUNSUPPORTED,
               "%s: needs Cin %% 32 == 0, M %% 32 == 0 (Cin=%lld M=%lld)", what, (long long)Cin, (long long)M);
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED, "%s: dimension too large", what);
  return TSPN_OK;
}
}  // namespace

// step 1: V = B^T d of x [B, T, Cin] into `workspace` (HBM-bound)
int tspn::wino43v_input_transform(const float* x, int64_t B, int64_t T, int64_t Cin, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  const char* what = "tspn_conv3_tc_wino43v_f32(input transform)";
  if (int rc = check_common(what, B, T, Cin, 32, T)) return rc;
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, TSPN_EINVAL, "%s: x must be a 16-byte aligned pointer", what);
  const size_t need = tspn::wino43v_workspace_bytes(B, T, Cin);
  TSPN_REQUIRE(workspace && workspace_bytes >= need, TSPN_EWORKSPACE, "%s: workspace %zu < %zu bytes", what,
               workspace_bytes, need);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, TSPN_EINVAL, "%s: workspace must be 16-byte aligned",
               what);
  const int64_t nq = tspn::ceil_div(T, 4);
  const int64_t nquads = B * nq, nqp = padded_quads(B, T);
  TSPN_REQUIRE(nqp / 32 < (1LL << 31) && Cin / 32 < 65536, TSPN_EUNSUPPORTED, "%s: grid too large", what);
  hipLaunchKernelGGL(wino43_input_transform_kernel, dim3((unsigned)(nqp / 32), (unsigned)(Cin / 32)), dim3(256), 0,
                     TSPN_STREAM(stream), x, static_cast<float*>(workspace), (int)T, (int)Cin, (int)nq, nquads, nqp,
                     B * T);
  return tspn::check_launch(what);
}

// step 2: the MFMA kernel on the transformed input
int tspn::wino43v_contract(const void* workspace, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                           const float* bias, int relu, float* y, int64_t ldy, void* stream) {
  const char* what = "tspn_conv3_tc_wino43v_f32";
  if (int rc = check_common(what, B, T, Cin, M, ldy)) return rc;
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(workspace && frag && y, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(frag) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 3) == 0,
               TSPN_EUNSUPPORTED, "%s: frag must be 16-byte aligned", what);
  const int64_t nq = tspn::ceil_div(T, 4);
  const int64_t nquads = B * nq, nqp = padded_quads(B, T);
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = nqp / QWG;
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "%s: grid too large", what);
  const int vec4 = (ldy % 4 == 0) && (ldy >= 4 * nq) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  static tspn::LdsLimit lds;   // 72 KB of dynamic LDS: above the 64 KB default limit
  if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_wino43v_kernel), SMEM_BYTES, what)) return rc;
  hipLaunchKernelGGL(conv3_wino43v_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), SMEM_BYTES,
                     TSPN_STREAM(stream), static_cast<const float*>(workspace), frag, bias, y, (int)Cin, (int)T, (int)M,
                     (int)nq, nquads, nqp, (int)tiles_m, (int)tiles_n, relu, (int)ldy, kPanelGroupV, vec4);
  return tspn::check_launch(what);
}

int tspn::conv3_tc_wino43v(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                           const float* bias, int relu, float* y, int64_t ldy, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (int rc = check_common("tspn_conv3_tc_wino43v_f32", B, T, Cin, M, ldy)) return rc;
  if (int rc = tspn::wino43v_input_transform(x, B, T, Cin, workspace, workspace_bytes, stream)) return rc;
  return tspn::wino43v_contract(workspace, B, T, Cin, frag, M, bias, relu, y, ldy, stream);
}

extern "C" int tspn_conv3_tc_wino43v_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag,
                                         int64_t M, const float* bias, int relu, float* y, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  return tspn::conv3_tc_wino43v(x, B, T, Cin, frag, M, bias, relu, y, T, workspace, workspace_bytes, stream);
}
