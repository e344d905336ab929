// Winograd F(6,3) temporal conv (gfx950, fp32): six output frames from eight inputs, 8 channel-GEMMs on a sixth
// of the columns = 4/9 of the direct MFMA work; T = 150 tiles exactly (25 sextets), any other T masks the last
// sextet of a tracklet.
//
// Accuracy: exact in real arithmetic.  In fp32, at the K = 3 x 2048 contraction of the headline config, the
// error against float64 is <= 64 eps sum_k |x_k||w_k| per output (direct kernel: <= 16 eps ...); on temporally
// smooth features it equals the direct kernel's, on temporally independent heavy-tailed ones it is up to 5x
// larger (tests/test_gpu_wino63.py, profiles/r3/conv_error_realistic.txt).
//
// Points 0, +-1, +-2, +-1/2, inf (Lavin & Gray).  d_i = x[6s + i - 1], i = 0..7 (zero outside the tracklet):
//   V0 = d0 - d6 + 5.25 (d4 - d2)                       V7 = d7 - d1 + 5.25 (d3 - d5)
//   V1,V2 = (d2 + d6 - 4.25 d4) +- (d1 + d5 - 4.25 d3)
//   V3,V4 = (d6 + 0.25 d2 - 1.25 d4) +- (0.5 d1 - 2.5 d3 + 2 d5)
//   V5,V6 = (d6 + 4 (d2 - 1.25 d4)) +- (2 d1 - 2.5 d3 + 0.5 d5)
//   U0 = g0, U1,U2 = -2/9 (g0 +- g1 + g2), U3,U4 = g0/90 +- g1/45 + 2 g2/45, U5,U6 = (32 g0 +- 16 g1 + 8 g2)/45, U7 = g2
//   y0 = M0 + p12 + p34 + p56        y1 = m12 + 2 m34 + m56/2       y2 = p12 + 4 p34 + p56/4
//   y3 = m12 + 8 m34 + m56/8         y4 = p12 + 16 p34 + p56/16     y5 = m12 + 32 m34 + m56/32 + M7
//   (p_ab = M_a + M_b, m_ab = M_a - M_b, M_j = U_j . V_j contracted over the channels)
//
// Structure: the input transform is its own HBM-bound pass (fp32 MFMA and VALU do not overlap on gfx950);
// V [Cin/4][8][sextets padded to 64][4] is staged by LDS-DMA in super-stages of 32 channels (64 KB, ring of 2 =
// 128 KB of LDS); fragment-major weights go straight into MFMA operand registers.  Workgroup = 128 rows x 64
// sextets on FOUR waves, one per SIMD, each with the whole 512-register file: wave w owns rows 32 w .. 32 w + 31
// for BOTH sextet halves (256 accumulator registers, all AGPRs), so a weight fragment is loaded once per
// workgroup and feeds two MFMAs, and there is room for the weights of TWO chunks ahead.  That distance is the
// point: VMEM returns in order, so the V pieces a wave requests from HBM (2-4 us under load) hold back every
// younger weight load of that wave.  An earlier 8-wave form (2 waves per SIMD, 128 accumulators, one chunk of
// lookahead, weight rows shared through L1) stalled the MFMA pipe 12 % of the time on exactly that, and its
// workgroups drifted apart until L2 no longer served a weight panel to its sharers (42.6 M KiB fetched per
// launch against 16.8 M here); numbers in profiles/r2/wino63_ablation.md.
// One barrier per super-stage, at the end of its THIRD chunk: by then every wave has seen its pieces of S + 1
// land (in-order return behind a counted wait) and has issued its last reads of S; the fourth chunk refills its
// V registers from S + 1, and the pieces of S + 2 go into the buffer S left during the first chunk of S + 1.
// Needs Cin % 32 == 0 and M % 32 == 0.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NW = 4;                     // waves per workgroup: one per SIMD, 512 registers each
constexpr int THREADS = 64 * NW;
constexpr int BM = 128;                   // output rows per workgroup (4 blocks of 32)
constexpr int ST = 32;                    // sextets per V half-tile (a wave multiplies both halves)
constexpr int SWG = 64;                   // sextets per workgroup (two halves)
constexpr int KC = 8;                     // channels per chunk
constexpr int NJ = 8;                     // positions
constexpr int VROW = ST * 4;              // floats per (channel group, position) row: 512 B
constexpr int VCH = 2 * NJ * VROW;        // floats per chunk and half-tile: [2 g][8 j][32 sextets][4 ch] = 8 KB
constexpr int VHALF = 4 * VCH;            // floats per super-stage and half-tile: 32 KB
constexpr int VSS = 2 * VHALF;            // floats per super-stage: 64 KB
constexpr int NVB = 2;                    // ring of super-stage buffers
constexpr int NPIECE = 16;                // DMA pieces (1 KiB) per wave and super-stage: 64 / 4
constexpr size_t SMEM_BYTES = sizeof(float) * NVB * VSS;
#ifndef TSPN_WINO63_GM
// Weight panels (128 rows) per tile group: the 32 workgroups an XCD runs at a time are GM panels x 32 / GM sextet
// tiles and stream GM x 8.4 MB of weights + 32 / GM x 4.2 MB of V through its L2 -- least for GM = 4 (67 MB per
// round, 26.9 GB per launch at cfg2:16).  Measured on one box (kernel + pre-pass ms / FETCH_SIZE M KiB):
// 1: 24.20 / 28.2, 2: 23.76 / 16.8, 3: 23.94 / 14.8, 4: 23.76 / 13.4, 8: 24.33 / 16.4.
#define TSPN_WINO63_GM 4
#endif

#ifndef TSPN_WINO63_VAUX
#define TSPN_WINO63_VAUX 0                // cache policy bits of the V bursts (2 = nt: 29.2 M KiB fetched but 27.6 ms)
#endif
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, TSPN_WINO63_VAUX);
}

// nn.Conv1d weight W [M, Cin, 3] -> fragment-major transformed weights
//   frag[m' / 32][chunk = ch / 8][j = 0..7][lane = 32 kh + li][e = 0..3] = U_j[8 chunk + 4 kh + e][32 (m'/32) + li]
// `split` (the factorised pair form): Cin == 2 * split; rows [0, M) take W[:, :split], rows [M, 2M) take W[:, split:].
__global__ void pack_wino63_frag_kernel(const float* __restrict__ W, int64_t M, int64_t Cin, int64_t split,
                                        float* __restrict__ out) {
  const int64_t Mp = split ? 2 * M : M, Cp = split ? split : Cin;
  const int64_t nch = Cp / KC;
  const int64_t total = NJ * Cp * Mp;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(o & 3);
    const int lane = (int)((o >> 2) & 63);
    const int64_t r = o >> 8;
    const int j = (int)(r % NJ);
    const int64_t c = (r / NJ) % nch;
    const int64_t mb = r / (NJ * nch);
    const int64_t ch = 8 * c + 4 * (lane >> 5) + e;
    const int64_t m = 32 * mb + (lane & 31);
    const float* g = (split && m >= M) ? W + ((m - M) * Cin + split + ch) * 3 : W + (m * Cin + ch) * 3;
    const double g0 = g[0], g1 = g[1], g2 = g[2];
    double u;
    switch (j) {
      case 0: u = g0; break;
      case 1: u = -2.0 / 9.0 * (g0 + g1 + g2); break;
      case 2: u = -2.0 / 9.0 * (g0 - g1 + g2); break;
      case 3: u = g0 / 90.0 + g1 / 45.0 + 2.0 * g2 / 45.0; break;
      case 4: u = g0 / 90.0 - g1 / 45.0 + 2.0 * g2 / 45.0; break;
      case 5: u = (32.0 * g0 + 16.0 * g1 + 8.0 * g2) / 45.0; break;
      case 6: u = (32.0 * g0 - 16.0 * g1 + 8.0 * g2) / 45.0; break;
      default: u = g2; break;
    }
    out[o] = (float)u;
  }
}

// Input transform V = B^T d.  A wave = 8 sextets x 8 channel groups (loads: whole 128-byte lines of x; stores:
// 128-byte runs of Vg).  Frames outside the tracklet are zero (the conv's padding); sextets past the end of the
// launch are zero too.
// `hot` (optional, for the accuracy guard of tspn_conv_guard.hip): the kernel reads every input value anyway, so it also
// reports WHERE the launch's largest |x| sits -- key = (float bits << 32 | sextet), one 64-bit atomic max per wave, no return
// value, into one of TSPN_CONV_CHECK_HOT_SLOTS slots 256 bytes apart (the reader takes the max over the slots).  Measured
// forms (profiles/r6/conv_guard.md): ONE word with an atomic load as a filter in front: +50 - 70 us (100 000 waves on one
// L2 line); one word with a cached plain load as the filter: +300 - 600 us (the stale filter lets thousands of atomics
// through to one address); 64 slots, unfiltered: within the noise of the kernel without it.
__global__ __launch_bounds__(256) void wino63_input_transform_kernel(
    const float* __restrict__ x, float* __restrict__ Vg, int T, int Cin, int nq, int64_t nsext, int64_t nsp,
    int64_t ncols, unsigned long long* __restrict__ hot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cgl = lane & 7, ql = lane >> 3;
  const int64_t S = ((int64_t)blockIdx.x * 4 + wave) * 8 + ql;     // < nsp by construction of the grid
  const int cg = blockIdx.y * 8 + cgl;
  if (4 * cg >= Cin) return;
  const bool ok = S < nsext;
  const int64_t b = ok ? S / nq : 0;
  const int q = ok ? (int)(S - b * nq) : 0;
  f32x4 d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = 6 * q + i - 1;
    int64_t n = b * T + t;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + n * Cin + 4 * cg);
    d[i] = (ok && t >= 0 && t < T) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (hot) {                                                         // uniform
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(d[i][0]), fabsf(d[i][1]))), fmaxf(fabsf(d[i][2]), fabsf(d[i][3])));
    unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned)(S < nsext ? S : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o, 64);
      key = other > key ? other : key;
    }
    if (lane == 0)
      __hip_atomic_fetch_max(hot + 32 * ((blockIdx.x * 4 + wave + 7 * blockIdx.y) & (TSPN_CONV_CHECK_HOT_SLOTS - 1)), key,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  f32x4 V[8];
  V[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
  V[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
  {
    const f32x4 t1 = d[2] + d[6] - 4.25f * d[4], t2 = d[1] + d[5] - 4.25f * d[3];
    V[1] = t1 + t2;
    V[2] = t1 - t2;
  }
  {
    const f32x4 t1 = d[6] + 0.25f * d[2] - 1.25f * d[4], t2 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    V[3] = t1 + t2;
    V[4] = t1 - t2;
  }
  {
    const f32x4 t1 = d[6] + 4.f * (d[2] - 1.25f * d[4]), t2 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    V[5] = t1 + t2;
    V[6] = t1 - t2;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    *reinterpret_cast<f32x4*>(Vg + (((int64_t)cg * NJ + j) * nsp + S) * 4) = V[j];
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

// BUFV: the V pieces are buffer loads (buffer_load_dwordx4 ... offen lds: one SGPR descriptor of the workspace, a fixed
// 32-bit lane offset per piece, a scalar offset per super-stage) instead of global_load_lds_dwordx4 with sixteen 64-bit
// pointers per lane that are bumped by vector adds every super-stage: cheaper to issue in the MFMA shadows
// (tools/probes/lds_dma_issue_probe.hip) and 16 registers less.  Needs a workspace below 4 GB (the launcher chooses).
template <bool BUFV>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3_wino63_kernel(
    const float* __restrict__ Vg, const float* __restrict__ Wf, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int nq, int64_t nsext, int64_t nsp, int tiles_m,
    int tiles_n, int relu, int ldy, int GM, int vec2) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Vs = reinterpret_cast<float*>(smem_raw);

  // workgroup -> tile: bijective XCD remap, then groups of GM weight panels x all sextet tiles
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
  const int tile_n = in_group / gm;
  const int m0 = tile_m * BM;
  const int64_t S0 = (int64_t)tile_n * SWG;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = 32-row block of the tile
  const int li = lane & 31, kh = lane >> 5;
  const int nsuper = Cin >> 5;

  const char* abase;            // wave-uniform; points at the chunk TWO ahead of the one being multiplied
  const unsigned aoff = lane * 16;
  {
    int mb = (m0 >> 5) + wave;
    mb = mb < (M >> 5) ? mb : 0;
    abase = reinterpret_cast<const char*>(Wf) + (int64_t)mb * (Cin / KC) * (NJ * 64 * 16);
  }

  // V super-stage DMA: piece pp = 16 wave + k, same LDS image as the 8-wave form
  const float* vsrc[BUFV ? 1 : NPIECE];
  unsigned voff[BUFV ? NPIECE : 1];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int pp = NPIECE * wave + k;
    const int rr = 2 * (pp & 31) + kh;
    const int cl = rr >> 4, rem = rr & 15;
    const int g = rem >> 3, j = rem & 7;
    const int64_t e = (((int64_t)(2 * cl + g) * NJ + j) * nsp + S0 + ST * (pp >> 5) + li) * 4;
    if constexpr (BUFV) voff[k] = (unsigned)(e * 4); else vsrc[k] = Vg + e;
  }
  const int64_t super_step = (int64_t)8 * NJ * nsp * 4;
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vg), 0, (int)0xffffffffu, 0x00020000);
  const unsigned super_bytes = (unsigned)(super_step * 4);
  auto stage_piece = [&](int S, int k) {          // piece k of this wave, super-stage S -> ring buffer S & 1
    if constexpr (BUFV) {
      const int soff = (int)((unsigned)S * super_bytes);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (__attribute__((address_space(3))) void*)(Vs + (S & 1) * VSS + (NPIECE * wave + k) * 256),
                                               16, (int)voff[k], soff, 0, 0);
    } else {
      glds16(vsrc[k], Vs + (S & 1) * VSS + (NPIECE * wave + k) * 256);
    }
    if constexpr (!BUFV) vsrc[k] += super_step;
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[cb][j][e] = 0.f;

  f32x4 a[2][NJ], v[2][NJ];
  const float* vlane = Vs + (kh * NJ * ST + li) * 4;     // + buffer + half + chunk + position offsets
  auto load_v = [&](const float* vbuf, int cl, int j) {
    v[0][j] = *reinterpret_cast<const f32x4*>(vbuf + cl * VCH + j * VROW);
    v[1][j] = *reinterpret_cast<const f32x4*>(vbuf + VHALF + cl * VCH + j * VROW);
  };
  auto load_a_pair = [&](f32x4* ap, auto jp_tag, const char* base) {     // positions 2 jp, 2 jp + 1 of the chunk at base
    constexpr int JP = decltype(jp_tag)::value;
    if (JP == 0) { load_frag<0>(ap[0], aoff, base); load_frag<1024>(ap[1], aoff, base); }
    if (JP == 1) { load_frag<2048>(ap[2], aoff, base); load_frag<3072>(ap[3], aoff, base); }
    if (JP == 2) { load_frag<0>(ap[4], aoff, base + 4096); load_frag<1024>(ap[5], aoff, base + 4096); }
    if (JP == 3) { load_frag<2048>(ap[6], aoff, base + 4096); load_frag<3072>(ap[7], aoff, base + 4096); }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  // 16 MFMAs of a position pair.  With one wave per SIMD nothing else feeds the MFMA pipe while this wave issues
  // VMEM instructions, so the 16 DMA pieces of a super-stage go out ONE per four MFMAs (S1 >= 0: pieces 4 slot ..
  // 4 slot + 3 of super-stage S1), in the shadow of the MFMA that was issued just before: issued as one burst they
  // cost 8 % of the kernel (GRBM cycles 59.2 M vs 54.5 M without DMA, profiles/r2/wino63_ablation.md).
  auto mfma_pair = [&](const f32x4* ap, int ja, int jb, int S1, int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0][ja] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ja][e], v[0][ja][e], acc[0][ja], 0, 0, 0);
      acc[0][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[jb][e], v[0][jb][e], acc[0][jb], 0, 0, 0);
      acc[1][ja] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ja][e], v[1][ja][e], acc[1][ja], 0, 0, 0);
      if (S1 >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        stage_piece(S1, 4 * slot + e);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[1][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[jb][e], v[1][jb][e], acc[1][jb], 0, 0, 0);
    }
  };

  // ---- prologue: super-stage 0 landed, V of chunk 0 in registers, the weights of chunks 0 and 1 in flight
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) stage_piece(0, k);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NJ; ++j) load_v(vlane, 0, j);
  load_a_pair(a[0], P0{}, abase); load_a_pair(a[0], P1{}, abase); load_a_pair(a[0], P2{}, abase); load_a_pair(a[0], P3{}, abase);
  abase += NJ * 1024;
  load_a_pair(a[1], P0{}, abase); load_a_pair(a[1], P1{}, abase); load_a_pair(a[1], P2{}, abase); load_a_pair(a[1], P3{}, abase);
  abase += NJ * 1024;
  __builtin_amdgcn_sched_barrier(0);

  // chunk c = 4 S + cl multiplies (A_c in a[c & 1], V_c); position pair by position pair (slot p) it refills
  // a[c & 1] with A_{c+2} and v with V_{c+1}.  VMEM issue order of slot p: [4 V pieces of super-stage S + 1, cl == 0]
  // then the two weight loads.  The wait before slot p needs the weight loads of slot p of chunk c - 2: younger
  // than those are the later slots of chunk c - 2, everything chunk c - 1 issued, and slots < p of this chunk.
  auto chunk_body = [&](auto cl_tag, const float* vcur, const float* vnext, int S, auto has1_tag, auto has2_tag,
                        auto more_tag) {
    constexpr int CL = decltype(cl_tag)::value;
    constexpr bool HAS1 = decltype(has1_tag)::value;   // chunk c + 1 exists: refill v
    constexpr bool HAS2 = decltype(has2_tag)::value;   // chunk c + 2 exists: refill a[c & 1]
    constexpr bool MORE = decltype(more_tag)::value;   // super-stage S + 1 exists: pieces at CL 0, barrier at CL 2
    constexpr int PAR = CL & 1;
    constexpr int N1 = (HAS1 ? 8 : 0) + ((MORE && CL == 1) ? NPIECE : 0);   // issued by chunk c - 1
    constexpr int N2 = (MORE && CL == 2) ? 6 : 2;      // per later slot of chunk c - 2 (CL 2: it carried the pieces)
    constexpr int N0 = (HAS2 ? 2 : 0) + ((MORE && CL == 0) ? 4 : 0);         // per earlier slot of this chunk
    const int S1 = (MORE && CL == 0) ? S + 1 : -1;
    const float* vn = CL == 3 ? vnext : vcur;
    constexpr int NCL = (CL + 1) & 3;
    f32x4* ap = a[PAR];
    wait_a<3 * N2 + N1>(ap[0], ap[1]);
    mfma_pair(ap, 0, 1, S1, 0);
    if (HAS2) load_a_pair(ap, P0{}, abase);
    if (HAS1) { load_v(vn, NCL, 0); load_v(vn, NCL, 1); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<2 * N2 + N1 + N0>(ap[2], ap[3]);
    mfma_pair(ap, 2, 3, S1, 1);
    if (HAS2) load_a_pair(ap, P1{}, abase);
    if (HAS1) { load_v(vn, NCL, 2); load_v(vn, NCL, 3); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<N2 + N1 + 2 * N0>(ap[4], ap[5]);
    mfma_pair(ap, 4, 5, S1, 2);
    if (HAS2) load_a_pair(ap, P2{}, abase);
    if (HAS1) { load_v(vn, NCL, 4); load_v(vn, NCL, 5); }
    __builtin_amdgcn_sched_barrier(0);
    wait_a<N1 + 3 * N0>(ap[6], ap[7]);
    mfma_pair(ap, 6, 7, S1, 3);
    if (HAS2) {
      load_a_pair(ap, P3{}, abase);
      abase += NJ * 1024;
    }
    if (HAS1) { load_v(vn, NCL, 6); load_v(vn, NCL, 7); }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE && CL == 2) {
      // the last wait of this chunk needed the weight loads chunk 4 S issued AFTER its last piece of super-stage
      // S + 1 (in-order VMEM return): this wave's pieces have landed; the reads above (V of chunk 4 S + 3) were the last
      // ones of buffer S & 1; the next chunk refills v from buffer S + 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
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
    int S = 0;
    for (; S + 1 < nsuper; ++S) {
      const float* vcur = vlane + (S & 1) * VSS;
      const float* vnext = vlane + ((S + 1) & 1) * VSS;
      chunk_body(C0{}, vcur, vnext, S, TT{}, TT{}, TT{});
      chunk_body(C1{}, vcur, vnext, S, TT{}, TT{}, TT{});
      chunk_body(C2{}, vcur, vnext, S, TT{}, TT{}, TT{});
      chunk_body(C3{}, vcur, vnext, S, TT{}, TT{}, TT{});
    }
    const float* vcur = vlane + (S & 1) * VSS;
    chunk_body(C0{}, vcur, vcur, S, TT{}, TT{}, FF{});
    chunk_body(C1{}, vcur, vcur, S, TT{}, TT{}, FF{});
    chunk_body(C2{}, vcur, vcur, S, TT{}, FF{}, FF{});
    chunk_body(C3{}, vcur, vcur, S, FF{}, FF{}, FF{});
  }

  // ---- output transform + store: lane column = sextet -> frames 6s .. 6s+5
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int64_t Sx = S0 + ST * cb + li;
    if (Sx < nsext) {
      const int64_t b = Sx / nq;
      const int q = (int)(Sx - b * nq);
      const int t = 6 * q;
      float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          const float p12 = acc[cb][1][e] + acc[cb][2][e], m12 = acc[cb][1][e] - acc[cb][2][e];
          const float p34 = acc[cb][3][e] + acc[cb][4][e], m34 = acc[cb][3][e] - acc[cb][4][e];
          const float p56 = acc[cb][5][e] + acc[cb][6][e], m56 = acc[cb][5][e] - acc[cb][6][e];
          float o[6];
          o[0] = acc[cb][0][e] + p12 + p34 + p56;
          o[1] = m12 + 2.f * m34 + 0.5f * m56;
          o[2] = p12 + 4.f * p34 + 0.25f * p56;
          o[3] = m12 + 8.f * m34 + 0.125f * m56;
          o[4] = p12 + 16.f * p34 + 0.0625f * p56;
          o[5] = m12 + 32.f * m34 + 0.03125f * m56 + acc[cb][7][e];
          if (bias != nullptr) {
            const float bb = bias[m];
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] += bb;
          }
          if (relu) {
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = fmaxf(o[i], 0.f);
          }
          float* dst = ycol + (int64_t)m * ldy;
          if (vec2) {
            *reinterpret_cast<f32x2*>(dst) = f32x2{o[0], o[1]};
            *reinterpret_cast<f32x2*>(dst + 2) = f32x2{o[2], o[3]};
            *reinterpret_cast<f32x2*>(dst + 4) = f32x2{o[4], o[5]};
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i)
              if (t + i < T) dst[i] = o[i];
          }
        }
      }
    }
  }
}

int64_t padded_sextets(int64_t B, int64_t T) { return tspn::ceil_div(B * tspn::ceil_div(T, 6), SWG) * SWG; }

int check_common(const char* what, int64_t B, int64_t T, int64_t Cin, int64_t M, int64_t ldy) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0 && ldy >= T && ldy < (1 << 24), TSPN_EINVAL,
               "%s: bad sizes B=%lld T=%lld Cin=%lld M=%lld ldy=%lld", what, (long long)B, (long long)T,
               (long long)Cin, (long long)M, (long long)ldy);
  TSPN_REQUIRE(tspn::wino63_supported(Cin, M), TSPN_EUNSUPPORTED,
               "%s: needs Cin %% 32 == 0, M %% 32 == 0 (Cin=%lld M=%lld)", what, (long long)Cin, (long long)M);
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED, "%s: dimension too large", what);
  return TSPN_OK;
}

}  // namespace

bool tspn::wino63_supported(int64_t Cin, int64_t M) { return Cin > 0 && M > 0 && Cin % 32 == 0 && M % 32 == 0; }

size_t tspn::wino63_workspace_bytes(int64_t B, int64_t T, int64_t Cin) {
  if (B <= 0 || T <= 0 || Cin <= 0) return 0;
  return (size_t)(Cin / 4) * NJ * (size_t)padded_sextets(B, T) * 4 * sizeof(float);
}

extern "C" size_t tspn_conv3_tc_wino63_workspace_bytes(int64_t B, int64_t T, int64_t Cin) {
  return tspn::wino63_workspace_bytes(B, T, Cin);
}

extern "C" int tspn_pack_conv3_wino63_frag_f32(const float* W, int64_t M, int64_t Cin, int64_t split, float* frag,
                                               void* stream) {
  TSPN_REQUIRE(W && frag, TSPN_EINVAL, "tspn_pack_conv3_wino63_frag_f32: null pointer");
  TSPN_REQUIRE(M > 0 && Cin > 0 && split >= 0, TSPN_EINVAL, "tspn_pack_conv3_wino63_frag_f32: bad sizes");
  TSPN_REQUIRE(split == 0 || Cin == 2 * split, TSPN_EINVAL,
               "tspn_pack_conv3_wino63_frag_f32: split=%lld requires Cin == 2*split (Cin=%lld)", (long long)split,
               (long long)Cin);
  const int64_t Mp = split ? 2 * M : M, Cp = split ? split : Cin;
  TSPN_REQUIRE(Cp % KC == 0 && Mp % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv3_wino63_frag_f32: needs (packed) Cin %% 8 == 0 and M %% 32 == 0 (Cin=%lld M=%lld)",
               (long long)Cp, (long long)Mp);
  const int64_t total = NJ * Cp * Mp;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_wino63_frag_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), W, M, Cin, split, frag);
  return tspn::check_launch("tspn_pack_conv3_wino63_frag_f32");
}

// step 1: V = B^T d of x [B, T, Cin] into `workspace` (HBM-bound)
int tspn::wino63_input_transform(const float* x, int64_t B, int64_t T, int64_t Cin, void* workspace,
                                 size_t workspace_bytes, void* stream, uint64_t* hot) {
  const char* what = "tspn_conv3_tc_wino63_f32(input transform)";
  if (int rc = check_common(what, B, T, Cin, 32, T)) return rc;
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, TSPN_EINVAL, "%s: x must be a 16-byte aligned pointer", what);
  const size_t need = tspn::wino63_workspace_bytes(B, T, Cin);
  TSPN_REQUIRE(workspace && workspace_bytes >= need, TSPN_EWORKSPACE, "%s: workspace %zu < %zu bytes", what,
               workspace_bytes, need);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, TSPN_EINVAL, "%s: workspace must be 16-byte aligned",
               what);
  const int64_t nq = tspn::ceil_div(T, 6);
  const int64_t nsext = B * nq, nsp = padded_sextets(B, T);
  TSPN_REQUIRE(nsp / 32 < (1LL << 31) && Cin / 32 < 65536, TSPN_EUNSUPPORTED, "%s: grid too large", what);
  hipLaunchKernelGGL(wino63_input_transform_kernel, dim3((unsigned)(nsp / 32), (unsigned)(Cin / 32)), dim3(256), 0,
                     TSPN_STREAM(stream), x, static_cast<float*>(workspace), (int)T, (int)Cin, (int)nq, nsext, nsp,
                     B * T, reinterpret_cast<unsigned long long*>(hot));
  return tspn::check_launch(what);
}

// 0 = buffer-load V pieces where the workspace allows (default), 1 = 64-bit pointer pieces everywhere
static std::atomic<int> g_piece_form{0};

// step 2: the MFMA kernel on the transformed input
int tspn::wino63_contract(const void* workspace, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                          const float* bias, int relu, float* y, int64_t ldy, void* stream) {
  const char* what = "tspn_conv3_tc_wino63_f32";
  if (int rc = check_common(what, B, T, Cin, M, ldy)) return rc;
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(workspace && frag && y, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(frag) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 3) == 0,
               TSPN_EUNSUPPORTED, "%s: frag must be 16-byte aligned", what);
  const int64_t nq = tspn::ceil_div(T, 6);
  const int64_t nsext = B * nq, nsp = padded_sextets(B, T);
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = nsp / SWG;
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "%s: grid too large", what);
  const int vec2 = (ldy % 2 == 0) && (ldy >= 6 * nq) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0);
  // buffer-load form of the V pieces where every byte offset into the workspace fits 32 bits
#ifndef TSPN_WINO63_BUFV
#define TSPN_WINO63_BUFV 1
#endif
  // (tspn_conv3_tc_wino63_set_piece_form(1) forces the pointer form: tests compare the two)
  const bool bufv = TSPN_WINO63_BUFV && tspn::wino63_workspace_bytes(B, T, Cin) < (1ull << 32) &&   /* offsets are unsigned 32-bit */
                    g_piece_form.load(std::memory_order_relaxed) == 0;
  static tspn::LdsLimit lds[2];   // 128 KB of dynamic LDS: above the 64 KB default limit
  auto launch = [&](auto kern, tspn::LdsLimit& lim) {
    if (int rc = lim.ensure(reinterpret_cast<const void*>(kern), SMEM_BYTES, what)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), SMEM_BYTES, TSPN_STREAM(stream),
                       static_cast<const float*>(workspace), frag, bias, y, (int)Cin, (int)T, (int)M, (int)nq, nsext, nsp,
                       (int)tiles_m, (int)tiles_n, relu, (int)ldy, TSPN_WINO63_GM, vec2);
    return tspn::check_launch(what);
  };
  return bufv ? launch(conv3_wino63_kernel<true>, lds[1]) : launch(conv3_wino63_kernel<false>, lds[0]);
}

int tspn::conv3_tc_wino63(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                          const float* bias, int relu, float* y, int64_t ldy, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (int rc = check_common("tspn_conv3_tc_wino63_f32", B, T, Cin, M, ldy)) return rc;
  if (int rc = tspn::wino63_input_transform(x, B, T, Cin, workspace, workspace_bytes, stream)) return rc;
  return tspn::wino63_contract(workspace, B, T, Cin, frag, M, bias, relu, y, ldy, stream);
}

extern "C" int tspn_conv3_tc_wino63_set_piece_form(int form) {
  TSPN_REQUIRE(form == 0 || form == 1, TSPN_EINVAL, "tspn_conv3_tc_wino63_set_piece_form: form must be 0 or 1");
  return g_piece_form.exchange(form, std::memory_order_relaxed);
}

extern "C" int tspn_conv3_tc_wino63_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag,
                                        int64_t M, const float* bias, int relu, float* y, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return tspn::conv3_tc_wino63(x, B, T, Cin, frag, M, bias, relu, y, T, workspace, workspace_bytes, stream);
}
