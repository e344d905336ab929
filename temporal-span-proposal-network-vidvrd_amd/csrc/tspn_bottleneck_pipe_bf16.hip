// Fused bottleneck tail, PERSISTENT and PIPELINED ACROSS TILES (round 4; CM = 256: the res4 blocks of a C4 backbone).
//
//     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )        3x3 / pad 1 / stride 1, then 1x1 expand
//
// Same arithmetic, tile (128 pixels x all channels) and operand paths as bottleneck_bf16_kernel<256>
// (tspn_bottleneck_bf16.hip: phase 2 = the 3x3 as a ring of linear x ranges + fragment-major weights straight into
// MFMA operand registers; h2 rounded to bf16 once into an LDS image; phase 3 = the expand in eight sub-passes with the
// epilogue of sub-pass i in the MFMA gaps of sub-pass i + 1, W3 rows fetched permuted, 32 contiguous bytes per lane) --
// bit-identical results -- but the two phases of a tile no longer run one after the other on the same four waves:
//
//   * one workgroup of EIGHT waves per CU walks tiles  b, b + G, b + 2 G, ...  (G = grid size <= number of CUs);
//   * waves 0-3 (team A) run phase 2 of tile t, waves 4-7 (team B) run phase 3 of tile t - 1 AT THE SAME TIME: a SIMD
//     holds one A wave and one B wave, so the MFMA-bound 3x3 of one tile fills the pipe while the expand of the
//     previous tile waits for its residual rows and stores (measured before: the phases of a launch ran in lockstep on
//     all CUs, the chip was either all-MFMA or all-memory, and their times ADDED: profiles/r3/bottleneck_tail_ablation.md);
//   * a wave's vector-memory operations retire in order on one counter, stores included: with the phases on
//     different waves the weight fragments of phase 2 no longer wait behind the output stores of phase 3;
//   * the x ring (4 stages) and ONE h2 image live side by side in LDS (136 KB).  The hardware has one barrier per
//     workgroup; joining team A's 13 ring barriers per tile would force team B into lockstep with it (first form of
//     this kernel: 578 us per 72 frames against 281 / 382 for the teams alone -- every window paid max(A, B) of its
//     jitter).  So there is NO s_barrier after the start: team A synchronises its four waves on a counter in LDS
//     (ds_add + a short poll), and the h2 image changes hands through two more counters -- A adds to `written` behind
//     its h2 stores, B starts a tile when written >= 4 (t + 1); B adds to `read` behind its last h2 read, A overwrites
//     the image when read >= 4 t.  A polling wave sleeps (s_sleep) and leaves its SIMD to its partner.
#include <algorithm>
#include <type_traits>

#include "tspn_common.h"
#include "tspn_status.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int CM = 256;
constexpr int THREADS = 512;
constexpr int BN = 128;                 // pixels per tile
constexpr int KC = 64;                  // channels per chunk
constexpr int SLP = 132;                // padded pixel slots per channel group
constexpr int B_ST = 8 * SLP * 16;      // bytes per x stage = per 64 channels of the h2 image
constexpr int MI = 2, NI = 4;           // 32-row blocks / 32-pixel blocks per A wave (wave wm: rows [64 wm, 64 wm + 64))
constexpr int CCH = CM / KC;            // 4
constexpr int NCHUNKS = 9 * CCH;
constexpr int C4 = 4 * CM;
constexpr int NST = 4, DIST = NST - 1;
constexpr int NRNG = 3 * CCH;           // 12 ranges per tile
constexpr int EXTRA_OFF = NST * B_ST;   // slots 128, 129 of the stages: [stage][8 groups][2] x 16 B
constexpr int ZERO_OFF = EXTRA_OFF + 1024;
constexpr int SYNC_OFF = ZERO_OFF + 64;  // three counters: +0 team A's ring barrier, +4 h2 written, +8 h2 read
constexpr int H2_OFF = ZERO_OFF + 256;  // h2 image [32 groups][132 slots][8 bf16]
constexpr int BIAS_OFF = H2_OFF + CCH * B_ST;   // b3 (4 CM floats): read in phase 3's epilogue without touching vmcnt
constexpr int SMEM = BIAS_OFF + 4 * CM * 4;
static_assert(SMEM <= 160 * 1024, "LDS budget");

__device__ __bf16 g_zero_page_bp[128];  // source of padding taps (never written)

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void load_wfrag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait_w(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}
// counters in LDS: one add per wave (lane 0), polled by every wave that waits
__device__ __forceinline__ void lds_add1(char* Bs, int off, int lane) {
  if (lane == 0)
    __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(Bs + off), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (round 6) bounded, through the asm wait of tspn_status.h: a wave that gives up raises TSPN_FAULT_HANDOVER and ends --
// until then this was an unbounded `while` (a lost arrival = a GPU reset)
__device__ __forceinline__ void lds_wait_ge(char* Bs, int off, unsigned target, int32_t* status) {
  tspn_dev::flag_wait((unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + off, (int)target, status, (int)blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
}
// barrier of team A's four waves (the caller has already waited for its own DMA / LDS operations)
__device__ __forceinline__ void a_barrier(char* Bs, int lane, unsigned& epoch, int32_t* status) {
  epoch += 4;
  lds_add1(Bs, SYNC_OFF, lane);
  lds_wait_ge(Bs, SYNC_OFF, epoch, status);
}

// tile of workgroup-slot `v` (0 .. tiles - 1): consecutive tiles stay on one XCD (shared halo rows), as in the
// one-tile-per-workgroup kernel
__device__ __forceinline__ int64_t tile_of(int v, int ntiles) {
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = v & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
}

// ------------------------------------------------------------------------------------------------ team A: phase 2
__device__ __forceinline__ void team_a(const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2,
                                       const float* __restrict__ bias2, int H, int W, int64_t npix, int ntiles,
                                       char* Bs, int w4_in, int lane_in, int32_t* status) {
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  using T = std::true_type;
  using F = std::false_type;
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;

  const int G = gridDim.x;
  const int iters = (ntiles - (int)blockIdx.x + G - 1) / G;      // tiles of this workgroup
  unsigned epoch = 0;                                      // 4 x (ring barriers passed)
  for (int it = 0; it < iters; ++it) {
    // everything derived from the lane / wave number is recomputed per tile from an opaque copy: hoisted out of the
    // tile loop these values stay live across the whole tile and the 254-register body spills (140 registers)
    int lane = lane_in, w4 = w4_in;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(w4));
    const int li = lane & 31, kh = lane >> 5;
    const unsigned woff = lane * 16;
    const int wm = w4;                                     // rows [64 wm, 64 wm + 64) of h2, all 128 pixels
    const char* const zslot = Bs + ZERO_OFF;
    char* const extra = Bs + EXTRA_OFF;
    const int slot = 64 * (w4 & 1) + lane;                 // x pieces: one pixel per lane, channel groups bg, bg + 2, ..
    const int bg = w4 >> 1;
    const int64_t n0 = tile_of(it * G + (int)blockIdx.x, ntiles) * BN;
    auto tap_mask = [&](int64_t n) {
      unsigned m = 0;
      const bool okn = n < npix;
      const int64_t nc = okn ? n : 0;
      const int64_t nb = nc / ((int64_t)H * W);
      const int r = (int)(nc - nb * H * W);
      const int oh = r / W, ow = r - oh * W;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          if (okn && oh - 1 + a >= 0 && oh - 1 + a < H && ow - 1 + b >= 0 && ow - 1 + b < W) m |= 1u << (a * 3 + b);
      return m;
    };
    unsigned rmask[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) rmask[ni] = tap_mask(n0 + ni * 32 + li);

    const char* wbase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      wbase[mi] = reinterpret_cast<const char*>(Wf2) + (int64_t)(MI * wm + mi) * NCHUNKS * 4096;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    f32x4 a[4][MI];
    auto mfma_step = [&](const f32x4 (&aw)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, aw[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[ni], acc[mi][ni], 0, 0, 0);
      }
    };
    auto load_step = [&](auto ks_tag) {
      constexpr int KS = decltype(ks_tag)::value;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) load_wfrag<1024 * KS>(a[KS][mi], woff, wbase[mi]);
    };
    auto bump = [&]() {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) wbase[mi] += 4096;
    };
    auto stage_r = [&](int buf, int i) {                   // range i = 3 c + ra: four pieces per wave (+ one: wave 0)
      const int c = i / 3, ra = i - 3 * c;
      const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + slot;
      const __bf16* xs = (q >= 0 && q < npix) ? h1 + q * CM + c * KC + 8 * bg : g_zero_page_bp + 8 * bg;
      char* dst = Bs + buf * B_ST + (bg * SLP + 64 * (w4 & 1)) * 16;
#pragma unroll
      for (int p = 0; p < 4; ++p) glds16(xs + 16 * p, dst + 2 * p * SLP * 16);
      if (w4 == 0 && lane < 16) {                          // slots 128, 129: [group][2]
        const int g = lane >> 1, e = lane & 1;
        const int64_t q2 = n0 + (int64_t)(ra - 1) * W - 1 + 128 + e;
        const __bf16* xs2 = (q2 >= 0 && q2 < npix) ? h1 + q2 * CM + c * KC + 8 * g : g_zero_page_bp + 8 * g;
        glds16(xs2, extra + buf * 256);                    // the DMA adds lane * 16
      }
    };
#pragma unroll
    for (int i = 0; i < DIST; ++i) stage_r(i, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a_barrier(Bs, lane, epoch, status);                            // the first ranges of the tile have landed
    load_step(K0{}); load_step(K1{}); load_step(K2{}); load_step(K3{});
    bump();
    __builtin_amdgcn_sched_barrier(0);

    auto read_r = [&](int buf, int tap, int rb, int g2, bf16x8 (&b)[NI]) {
      const char* Bb = Bs + buf * B_ST + ((g2 + kh) * SLP + li + rb) * 16;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const char* bp = Bb + ni * 32 * 16;
        if (ni == 3) bp = (li + rb >= 32) ? extra + buf * 256 + ((g2 + kh) * 2 + (li + rb - 32)) * 16 : bp;
        if (!((rmask[ni] >> tap) & 1u)) bp = zslot;        // the tap falls off the image: read zeros
        b[ni] = *reinterpret_cast<const bf16x8*>(bp);
      }
    };
    // one round = the four k-steps of tap (ra, rb) of range i; VMEM issue order and counted waits exactly as in
    // bottleneck_bf16_kernel<256> (its comments apply)
    auto round_body = [&](int i, int buf, int ra, auto rb_tag, auto stage_tag, auto more_tag, auto last_tag) {
      constexpr int rb = decltype(rb_tag)::value;
      constexpr bool STAGE = decltype(stage_tag)::value, MORE = decltype(more_tag)::value, LAST = decltype(last_tag)::value;
      constexpr int NX = STAGE ? 4 : 0, R = MORE ? MI : 0, L = MI;
      const int tap = 3 * ra + rb;
      bf16x8 b0[NI] = {}, b1[NI] = {};
      wait_w<3 * L>(a[0][0], a[0][1]);
      if (STAGE) stage_r(buf >= 1 ? buf - 1 : NST - 1, i + DIST);
      __builtin_amdgcn_sched_barrier(0);
      read_r(buf, tap, rb, 0, b0);
      read_r(buf, tap, rb, 2, b1);
      mfma_step(a[0], b0);
      if (MORE) load_step(K0{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<2 * L + NX + R>(a[1][0], a[1][1]);
      read_r(buf, tap, rb, 4, b0);
      mfma_step(a[1], b1);
      if (MORE) load_step(K1{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<L + NX + 2 * R>(a[2][0], a[2][1]);
      read_r(buf, tap, rb, 6, b1);
      mfma_step(a[2], b0);
      if (MORE) load_step(K2{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<NX + 3 * R>(a[3][0], a[3][1]);
      mfma_step(a[3], b1);
      if (MORE) { load_step(K3{}); bump(); }
      __builtin_amdgcn_sched_barrier(0);
      if (LAST) {      // every LDS read of this range has returned; the next range has landed
        if (MORE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        a_barrier(Bs, lane, epoch, status);                        // one per range: 12 per tile
      }
    };
    {
      int buf = 0, ra = 0;
      int i = 0;
      for (; i + DIST < NRNG; ++i) {
        round_body(i, buf, ra, R0{}, T{}, T{}, F{});
        round_body(i, buf, ra, R1{}, F{}, T{}, F{});
        round_body(i, buf, ra, R2{}, F{}, T{}, T{});
        buf = buf + 1 == NST ? 0 : buf + 1;
        ra = ra == 2 ? 0 : ra + 1;
      }
      for (; i + 1 < NRNG; ++i) {
        round_body(i, buf, ra, R0{}, F{}, T{}, F{});
        round_body(i, buf, ra, R1{}, F{}, T{}, F{});
        round_body(i, buf, ra, R2{}, F{}, T{}, T{});
        buf = buf + 1 == NST ? 0 : buf + 1;
        ra = ra == 2 ? 0 : ra + 1;
      }
      round_body(i, buf, ra, R0{}, F{}, T{}, F{});
      round_body(i, buf, ra, R1{}, F{}, T{}, F{});
      round_body(i, buf, ra, R2{}, F{}, F{}, T{});          // ends with a ring barrier: nobody reads the stages any more
    }
    lds_wait_ge(Bs, SYNC_OFF + 8, 4u * it, status);                // team B has finished reading the previous tile's h2
    // h2 = relu(acc + b2) -> bf16 -> the h2 image (B-operand layout)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * (MI * wm + mi) + 8 * q + 4 * kh;
        const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bf16x4 v;
          v[0] = (__bf16)fmaxf(acc[mi][ni][4 * q] + bv.x, 0.f);
          v[1] = (__bf16)fmaxf(acc[mi][ni][4 * q + 1] + bv.y, 0.f);
          v[2] = (__bf16)fmaxf(acc[mi][ni][4 * q + 2] + bv.z, 0.f);
          v[3] = (__bf16)fmaxf(acc[mi][ni][4 * q + 3] + bv.w, 0.f);
          *reinterpret_cast<bf16x4*>(Bs + H2_OFF + ((ch >> 3) * SLP + ni * 32 + li) * 16 + 8 * kh) = v;
        }
      }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    lds_add1(Bs, SYNC_OFF + 4, lane);                      // this wave's rows of h2(it) are written
  }
}

// ------------------------------------------------------------------------------------------------ team B: phase 3
__device__ __forceinline__ void team_b(const __bf16* __restrict__ Wf3, const float* __restrict__ bias3,
                                       const __bf16* __restrict__ residual, __bf16* __restrict__ out, int64_t npix,
                                       int ntiles, char* Bs, int w4_in, int lane_in, int32_t* status) {
  constexpr int KSTEPS = CM / 16;                  // 16
  constexpr int MS = 2, NS = 2, NSUB = 8;
  constexpr int NGRP = MS * NS;                    // epilogue groups per sub-pass
#ifndef TSPN_BP_RING
#define TSPN_BP_RING 4
#endif
#ifndef TSPN_BP_NRES
#define TSPN_BP_NRES 2
#endif
  constexpr int RING = TSPN_BP_RING, NRES = TSPN_BP_NRES;
  const int G = gridDim.x;
  const int iters = (ntiles - (int)blockIdx.x + G - 1) / G;
  for (int it = 0; it < iters; ++it) {
    int lane = lane_in, w4 = w4_in;                        // opaque per tile: see team_a
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(w4));
    const int li = lane & 31, kh = lane >> 5;
    const int wm = w4;
    const char* Hb = Bs + H2_OFF + (kh * SLP + li) * 16;
    auto sub_rb = [&](int sp) { return (wm * 4 + (sp >> 1)) * MI; };
    auto sub_nb = [&](int sp) { return (sp & 1) * NS; };
    const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;
    const char* const w3base = reinterpret_cast<const char*>(Wf3) + woff3;
    auto a_ptr = [&](int sp, int k, int ms) {
      return w3base + (int64_t)(sub_rb(sp) + ms) * (CCH * 4096) + k * 1024;
    };
    const unsigned voff_in = (unsigned)(li * C4 + 16 * kh) * 2, voff_out = (unsigned)(16 * kh) * 2;
    const int64_t n0 = tile_of(it * G + (int)blockIdx.x, ntiles) * BN;
    unsigned okmask = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) okmask |= (n0 + b * 32 + li < npix ? 1u : 0u) << b;
    auto row_base = [&](int sp, int grp) {
      const int ms = grp % MS, nj = grp / MS;
      const int64_t pb = n0 + (sub_nb(sp) + nj) * 32;
      return (pb < npix ? pb : 0) * C4 + 32 * (sub_rb(sp) + ms);
    };
    f32x4 ar[RING][MS];
#pragma unroll
    for (int j = 0; j < RING; ++j)
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) ar[j][ms] = *reinterpret_cast<const f32x4*>(a_ptr(0, j, ms));
    f32x16 accA[MS][NS], accB[MS][NS];
    bf16x8 rres[NRES][2];
    bf16x8 ohold;

    // ring slots do not depend on the sub-pass (KSTEPS % RING == 0, NGRP % NRES == 0): the sub-pass index is a run-time
    // value, as in bottleneck_bf16_kernel<256> (fully unrolled, the address arithmetic of all eight sub-passes is
    // hoisted and the body spills)
    auto kstep = [&](f32x16 (&c)[MS][NS], int sp, auto k_tag) {
      constexpr int k = decltype(k_tag)::value;
      constexpr int slot = k % RING;
      const int nb = sub_nb(sp);
      bf16x8 b[NS];
#pragma unroll
      for (int nj = 0; nj < NS; ++nj)
        b[nj] = *reinterpret_cast<const bf16x8*>(Hb + ((2 * k) * SLP + (nb + nj) * 32) * 16);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, ar[slot][ms]);
#pragma unroll
        for (int nj = 0; nj < NS; ++nj) c[ms][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[nj], c[ms][nj], 0, 0, 0);
      }
      constexpr int dsp = (k + RING) / KSTEPS, kn = (k + RING) % KSTEPS;
      if (sp + dsp < NSUB) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) ar[slot][ms] = *reinterpret_cast<const f32x4*>(a_ptr(sp + dsp, kn, ms));
      }
    };
    auto res_issue = [&](int e, auto grp_tag) {
      constexpr int grp = decltype(grp_tag)::value;
      constexpr int slot = grp % NRES;
      const bool okp = (okmask >> (sub_nb(e) + grp / MS)) & 1u;
      const char* rp = reinterpret_cast<const char*>(residual + row_base(e, grp)) + (okp ? voff_in : voff_out);
      rres[slot][0] = *reinterpret_cast<const bf16x8*>(rp);
      rres[slot][1] = *reinterpret_cast<const bf16x8*>(rp + 16);
    };
    auto group_finish = [&](f32x16 (&c)[MS][NS], int e, auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      constexpr int grp = g >> 1, h = g & 1, ms = grp % MS, nj = grp / MS;
      constexpr int slot = grp % NRES;
      const int chm = 32 * (sub_rb(e) + ms) + 16 * kh + 8 * h;
      const float4 bv0 = *reinterpret_cast<const float4*>(Bs + BIAS_OFF + 4 * chm);
      const float4 bv1 = *reinterpret_cast<const float4*>(Bs + BIAS_OFF + 4 * chm + 16);
      const float v[8] = {c[ms][nj][8 * h] + bv0.x,     c[ms][nj][8 * h + 1] + bv0.y, c[ms][nj][8 * h + 2] + bv0.z,
                          c[ms][nj][8 * h + 3] + bv0.w, c[ms][nj][8 * h + 4] + bv1.x, c[ms][nj][8 * h + 5] + bv1.y,
                          c[ms][nj][8 * h + 6] + bv1.z, c[ms][nj][8 * h + 7] + bv1.w};
      const bf16x8 rv = rres[slot][h];
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf(v[j] + (float)rv[j], 0.f);
      if constexpr (h == 0) {
        ohold = o;
      } else {
        const bool okp = (okmask >> (sub_nb(e) + nj)) & 1u;
        char* op = reinterpret_cast<char*>(out + row_base(e, grp)) + voff_in;
        if (okp) {
          *reinterpret_cast<bf16x8*>(op) = ohold;
          *reinterpret_cast<bf16x8*>(op + 16) = o;
        }
        // the group NRES after (e, grp) in the stream takes the ring slot
        constexpr int de = (grp + NRES) / NGRP, gn = (grp + NRES) % NGRP;
        if (e + de < NSUB) res_issue(e + de, std::integral_constant<int, gn>{});
      }
    };
    auto zero = [&](f32x16 (&c)[MS][NS]) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int nj = 0; nj < NS; ++nj)
#pragma unroll
          for (int e = 0; e < 16; ++e) c[ms][nj][e] = 0.f;
    };
    // sub-pass sp into `cur` while the epilogue of sub-pass sp - 1 (in `prev`) drains
    auto subpass = [&](f32x16 (&cur)[MS][NS], f32x16 (&prev)[MS][NS], int sp, auto drain_tag) {
      constexpr bool DRAIN = decltype(drain_tag)::value;
      zero(cur);
      auto two = [&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        kstep(cur, sp, std::integral_constant<int, 2 * g>{});
        __builtin_amdgcn_sched_barrier(0);
        kstep(cur, sp, std::integral_constant<int, 2 * g + 1>{});
        if constexpr (DRAIN) group_finish(prev, sp - 1, g_tag);
        __builtin_amdgcn_sched_barrier(0);
      };
      two(std::integral_constant<int, 0>{}); two(std::integral_constant<int, 1>{});
      two(std::integral_constant<int, 2>{}); two(std::integral_constant<int, 3>{});
      two(std::integral_constant<int, 4>{}); two(std::integral_constant<int, 5>{});
      two(std::integral_constant<int, 6>{}); two(std::integral_constant<int, 7>{});
    };
    res_issue(0, std::integral_constant<int, 0>{});        // residual rows and W3 fragments are in flight while ...
    res_issue(0, std::integral_constant<int, 1>{});
    if constexpr (NRES > 2) {
      res_issue(0, std::integral_constant<int, 2>{});
      res_issue(0, std::integral_constant<int, 3>{});
    }
    static_assert(NRES == 2 || NRES == 4, "residual ring: two or four groups");
    lds_wait_ge(Bs, SYNC_OFF + 4, 4u * (it + 1), status);          // ... team A finishes h2 of this tile
    subpass(accA, accB, 0, std::false_type{});
#pragma unroll 1
    for (int sp = 1; sp < NSUB; sp += 2) {
      subpass(accB, accA, sp, std::true_type{});
      if (sp + 1 < NSUB) subpass(accA, accB, sp + 1, std::true_type{});
    }
    // every read of the h2 image has been issued; it must have RETURNED before team A may overwrite the image
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_add1(Bs, SYNC_OFF + 8, lane);                      // this wave is done with the image
    {                                                      // the last sub-pass drains on its own, under A's h2 write
      constexpr int L = NSUB - 1;
      group_finish(accB, L, std::integral_constant<int, 0>{}); group_finish(accB, L, std::integral_constant<int, 1>{});
      group_finish(accB, L, std::integral_constant<int, 2>{}); group_finish(accB, L, std::integral_constant<int, 3>{});
      group_finish(accB, L, std::integral_constant<int, 4>{}); group_finish(accB, L, std::integral_constant<int, 5>{});
      group_finish(accB, L, std::integral_constant<int, 6>{}); group_finish(accB, L, std::integral_constant<int, 7>{});
    }
  }
}

__global__ __launch_bounds__(THREADS, 2) void bottleneck_pipe_bf16_kernel(
    const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2, const float* __restrict__ bias2,
    const __bf16* __restrict__ Wf3, const float* __restrict__ bias3, const __bf16* __restrict__ residual,
    __bf16* __restrict__ out, int H, int W, int64_t npix, int ntiles, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char Bs[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 4) reinterpret_cast<float*>(Bs + ZERO_OFF)[tid] = 0.f;     // the zero slot ...
  if (tid >= 16 && tid < 19) reinterpret_cast<unsigned*>(Bs + SYNC_OFF)[tid - 16] = 0u;   // ... and the three counters
  if (tid < CM) *reinterpret_cast<float4*>(Bs + BIAS_OFF + 16 * tid) = *reinterpret_cast<const float4*>(bias3 + 4 * tid);
  __syncthreads();                                                       // the only workgroup barrier of the kernel
  if (wave < 4) team_a(h1, Wf2, bias2, H, W, npix, ntiles, Bs, wave, lane, status);
  else team_b(Wf3, bias3, residual, out, npix, ntiles, Bs, wave - 4, lane, status);
}

}  // namespace

extern "C" int tspn_bottleneck_tail_pipe_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CMi,
                                              const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                              const float* bias3, const uint16_t* residual, uint16_t* out,
                                              int64_t max_workgroups, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "tspn_bottleneck_tail_pipe_bf16: bad sizes");
  TSPN_REQUIRE(CMi == CM, TSPN_EUNSUPPORTED,
               "tspn_bottleneck_tail_pipe_bf16: built for 256 bottleneck channels (got %lld)", (long long)CMi);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(h1 && frag2 && bias2 && frag3 && bias3 && residual && out, TSPN_EINVAL,
               "tspn_bottleneck_tail_pipe_bf16: null pointer");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(h1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) && al16(residual) && al16(out),
               TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_pipe_bf16: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_pipe_bf16: dimension too large");
  const int64_t npix = NB * H * W;
  const int64_t tiles = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles < (1LL << 30), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_pipe_bf16: too many tiles");
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int64_t grid = max_workgroups > 0 ? max_workgroups : cus;      // one workgroup per CU (136 KB of LDS each)
  grid = std::min<int64_t>(grid, tiles);
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(bottleneck_pipe_bf16_kernel), SMEM, "tspn_bottleneck_tail_pipe_bf16"))
    return rc;
  hipLaunchKernelGGL(bottleneck_pipe_bf16_kernel, dim3((unsigned)grid), dim3(THREADS), SMEM, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(h1), reinterpret_cast<const __bf16*>(frag2), bias2,
                     reinterpret_cast<const __bf16*>(frag3), bias3, reinterpret_cast<const __bf16*>(residual),
                     reinterpret_cast<__bf16*>(out), (int)H, (int)W, npix, (int)tiles, tspn::status_device_ptr());
  return tspn::check_launch("tspn_bottleneck_tail_pipe_bf16");
}
