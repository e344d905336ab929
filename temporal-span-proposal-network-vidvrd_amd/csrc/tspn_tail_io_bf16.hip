// Fused bottleneck tail at CM = 256 (the res4 chain of the C4 backbone: 43 % of its time) with the work split by ROLE
// instead of by phase (round 5):
//     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )        3x3 / pad 1 / stride 1, then 1x1 expand
// -- the arithmetic, tiling (128 linear pixels per workgroup, ring of linear h1 ranges, h2 image in LDS, W3 rows permuted
// at load) and contraction order of bottleneck_bf16_kernel<256> (tspn_bottleneck_bf16.hip), bit-identical results.
//
// What round 4 established about that kernel (profiles/r4/bottleneck_pipeline_study.md): its 3x3 phase runs at ~1 PFLOP/s,
// its expand MFMAs at 0.9, its 118 MB of residual + output per 8 frames at 5.4 TB/s -- and the launch takes their SUM,
// because every vector-memory operation of a wave retires in order on one counter: a residual row (HBM, 2 - 3 us) issued
// in front of a W3 fragment (L2) holds that fragment back, a store holds everything behind it, so the wave that feeds
// the MFMA pipe spends its time waiting for memory it does not need.  Here a workgroup has EIGHT waves, two per SIMD:
//   waves 0-3 (compute): all MFMAs.  Their only vector-memory traffic is the weight stream from L2 (W2, W3 fragments
//                        straight into operand registers through small rings); B operands come from LDS.
//   waves 4-7 (io):      everything that touches HBM.  Phase 2: the LDS-DMA of the h1 ranges (a ring of four stages, three
//                        ranges ahead, counted vmcnt on a queue that holds nothing else).  Phase 3: the epilogue -- they
//                        take a sub-pass's fp32 sums from an LDS exchange buffer, add b3 and the residual rows (requested
//                        four sub-passes ahead into their own registers), ReLU, round, store.
// The two roles meet at one s_barrier per h1 range and one per expand sub-pass (fp32 sums double-buffered in LDS), with
// explicit lgkmcnt / vmcnt waits in front of each (no __syncthreads: its fence would drain the weight rings).  Both roles
// run the SAME barrier skeleton (the loops below are shared, the bodies are role-specific), so the counts match by
// construction.  138 KB of LDS, one workgroup per CU.
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512;
constexpr int BN = 128;                 // pixels per workgroup
constexpr int CM = 256, C4 = 4 * CM, KC = 64, CCH = CM / KC;
constexpr int SLP = 132;                // padded pixel slots per channel group
constexpr int B_ST = 8 * SLP * 16;      // bytes per h1 range stage = per 64 channels of the h2 image
constexpr int NST = 4, DIST = 3;        // ring of range stages, filled DIST ranges ahead
constexpr int NRNG = 3 * CCH;           // ranges per tile: (64-channel part, tap row)
constexpr int EXTRA_OFF = NST * B_ST;   // slots 128, 129 of a stage: [stage][8 groups][2 slots] x 16 B
constexpr int ZERO_OFF = EXTRA_OFF + NST * 256;
constexpr int B3_OFF = ZERO_OFF + 256;
constexpr int XCH_OFF = B3_OFF + C4 * 4;    // fp32 sums of a sub-pass: [2 sets][4 waves][2 blocks][4 quads][64 lanes] x 16 B
constexpr int XCH_WAVE = 2 * 4 * 64 * 16, XCH_SET = 4 * XCH_WAVE;
constexpr int SMEM = XCH_OFF + 2 * XCH_SET;
constexpr int NSUB = 16;                // expand sub-passes per compute wave: (row block, pair of pixel blocks)
constexpr int RD = 4;                   // residual rows requested this many sub-passes ahead
static_assert(CCH * B_ST == NST * B_ST && SMEM <= 160 * 1024, "h2 image = the four stages; LDS budget");

// meet the other role: LDS traffic of this wave complete (reads consumed / writes landed), then the workgroup barrier
__device__ __forceinline__ void role_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(THREADS, 1) void tail_io_bf16_kernel(
    const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2, const float* __restrict__ bias2,
    const __bf16* __restrict__ Wf3, const float* __restrict__ bias3, const __bf16* __restrict__ residual,
    __bf16* __restrict__ out, int H, int W, int64_t npix) {
  extern __shared__ __attribute__((aligned(16))) char Bs[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;      // consecutive pixel tiles stay on one XCD (shared halo rows)
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t n0 = (int64_t)wg * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool io = wave >= 4;
  const int w4 = wave & 3;
  const int li = lane & 31, kh = lane >> 5;
  if (tid < 4) reinterpret_cast<float*>(Bs + ZERO_OFF)[tid] = 0.f;               // published by the first barrier
  for (int i = tid; i < CM; i += THREADS)                                          // b3 -> LDS, likewise
    *reinterpret_cast<float4*>(Bs + B3_OFF + 16 * i) = *reinterpret_cast<const float4*>(bias3 + 4 * i);
  const char* const zslot = Bs + ZERO_OFF;
  constexpr unsigned OOB = 0x80000000u;

  // ================================================================ phase 2: 3x3 conv, K = 4 parts x 9 taps x 64
  // ---- io side: the ranges.  Range i = 3 c + ra holds pixels n0 + (ra - 1) W - 1 .. + 129 of channel part c; wave w4 stages
  // the pixel (slot) 64 (w4 & 1) + lane, channel groups bg, bg + 2, bg + 4, bg + 6 (bg = w4 >> 1); slots 128, 129 go to a
  // side region (every io wave issues that piece -- same bytes, same place -- so that each has FIVE pieces per range)
  const int64_t rbase = n0 - W - 1 > 0 ? n0 - W - 1 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_h1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(h1) + rbase * CM, 0, 0x7fffffff, 0x00020000);
  const int slot = 64 * (w4 & 1) + lane, bg = w4 >> 1;
  auto stage_r = [&](int buf, int i) {
    const int c = i / 3, ra = i - 3 * c;
    const int soff = c * KC * 2;
    const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + slot;
    const unsigned voff = (q >= 0 && q < npix) ? (unsigned)((q - rbase) * CM * 2 + 16 * bg) : OOB;
    char* dst = Bs + buf * B_ST + (bg * SLP + 64 * (w4 & 1)) * 16;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_h1, (__attribute__((address_space(3))) void*)(dst + 2 * p * SLP * 16), 16,
                                               (int)(voff == OOB ? OOB : voff + 32 * p), soff, 0, 0);
    const int g = (lane >> 1) & 7, e = lane & 1;
    const int64_t q2 = n0 + (int64_t)(ra - 1) * W - 1 + 128 + e;
    const unsigned voff2 = (lane < 16 && q2 >= 0 && q2 < npix) ? (unsigned)((q2 - rbase) * CM * 2 + 16 * g) : OOB;
    if (lane < 16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_h1, (__attribute__((address_space(3))) void*)(Bs + EXTRA_OFF + buf * 256), 16,
                                               (int)voff2, soff, 0, 0);
  };
  // wait until at most `ranges` of this wave's ranges (5 pieces each) are still in flight
  auto wait_ranges = [&](int ranges) {
    if (ranges >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (ranges == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // sub-pass e of wave w4 in phase 3: row block mb = 8 w4 + (e >> 1) (channels 32 mb ..), pixel blocks 2 (e & 1), 2 (e & 1) + 1
  char* const xw = Bs + XCH_OFF + w4 * XCH_WAVE + lane * 16;   // exchange buffer of this wave pair: + set, + (block, quad) * 1024

  // The two roles run SEPARATE straight-line programs with the same barrier sequence -- 1 + NRNG + 1 + NSUB -- instead of one
  // loop nest with role branches inside: with the branches inside, the 128 accumulator registers are loop-carried through
  // the io path as well and hipcc copies and spills them around every branch (1 200 spilled registers in that form).
  if (io) {
    // ================================================================ io waves
#pragma unroll
    for (int i = 0; i < DIST; ++i) stage_r(i, i);
    wait_ranges(DIST - 1);                                   // range 0 has landed
    role_barrier();                                          // [0]
    for (int i = 0; i < NRNG; ++i) {
      // stage (i + DIST) % NST = (i - 1) % NST was read in the previous interval; its barrier lies behind us
      if (i + DIST < NRNG) stage_r((i + DIST) & (NST - 1), i + DIST);
      const int last_issued = i + DIST < NRNG ? i + DIST : NRNG - 1;
      wait_ranges(last_issued - (i + 1));                   // range i + 1 has landed before the barrier publishes it
      role_barrier();                                        // [1 + i]
    }
    // (the compute waves write the h2 image now)  residual rows of the first RD sub-passes meanwhile
    const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(residual) + n0 * C4, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(out + n0 * C4, 0, 0x7fffffff, 0x00020000);
    unsigned po[4];                                          // byte offset of this lane's pixel of each pixel block (channel 16 kh)
#pragma unroll
    for (int b = 0; b < 4; ++b) po[b] = (n0 + b * 32 + li < npix) ? (unsigned)(((b * 32 + li) * C4 + 16 * kh) * 2) : OOB;
    bf16x8 res[RD][2][2];                                    // residual rows in flight [sub-pass % RD][pixel block][half]
    auto res_issue_slot = [&](int slot_, int e) {           // slot_ = e % RD, compile-time at every call site
      const int mb = 8 * w4 + (e >> 1);
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned p = (e & 1) ? po[2 + pj] : po[pj];
          res[slot_][pj][h] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, (int)(p == OOB ? OOB : p + 16 * h), mb * 64, 0));
        }
    };
#pragma unroll
    for (int e = 0; e < RD; ++e) res_issue_slot(e, e);
    role_barrier();                                          // [1 + NRNG]: h2 complete (nothing of ours depends on it)
    role_barrier();                                          // [2 + NRNG]: sub-pass 0's sums are in set 0
    u32x4_t keep[2] = {};
    static_assert(NSUB % RD == 0, "the sub-pass loop is unrolled by the residual ring's depth");
    for (int e0 = 0; e0 < NSUB; e0 += RD) {
#pragma unroll
      for (int u = 0; u < RD; ++u) {
        const int e = e0 + u;                                // this sub-pass's sums were published by the last barrier
        const int mb = 8 * w4 + (e >> 1);
        const char* const src = xw + (e & 1) * XCH_SET;
        const char* const b3s = Bs + B3_OFF + (32 * mb + 16 * kh) * 4;
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(b3s + 16 * q);
          bv[4 * q] = t.x; bv[4 * q + 1] = t.y; bv[4 * q + 2] = t.z; bv[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          float v[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + (pj * 4 + q) * 1024);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
          }
          u32x4_t o2[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf((v[8 * h + j] + bv[8 * h + j]) + (float)res[u][pj][h][j], 0.f);
            o2[h] = __builtin_bit_cast(u32x4_t, o);
          }
          const unsigned p = (e & 1) ? po[2 + pj] : po[pj];
          __builtin_amdgcn_raw_buffer_store_b128(o2[0], rsrc_out, (int)(p == OOB ? OOB : p), mb * 64, 0);
          __builtin_amdgcn_raw_buffer_store_b128(o2[1], rsrc_out, (int)(p == OOB ? OOB : p + 16), mb * 64, 0);
          // store-data hazard (tools/lint_store_hazard.py, profiles/r5/bottleneck_block_study.md §3): the data registers
          // of a group stay live until the next group's stores have been issued
          asm volatile("" ::"v"(keep[0]), "v"(keep[1]));
          keep[0] = o2[0];
          keep[1] = o2[1];
        }
        if (e + RD < NSUB) res_issue_slot(u, e + RD);        // its ring slot is free now
        if (e + 1 < NSUB) role_barrier();                    // [3 + NRNG + e]: set e & 1 is free, sub-pass e + 1's sums are published
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n s_nop 15" ::"v"(keep[0]), "v"(keep[1]));
  } else {
    // ================================================================ compute waves: wave w4 = rows [64 w4, 64 w4 + 64) x all 128 pixels
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    unsigned rmask[4];                                       // taps of this lane's B columns that fall inside the image
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int64_t n = n0 + ni * 32 + li;
      unsigned m = 0;
      if (n < npix) {
        const int64_t nb = n / ((int64_t)H * W);
        const int r = (int)(n - nb * H * W);
        const int oh = r / W, ow = r - oh * W;
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            if (oh - 1 + a >= 0 && oh - 1 + a < H && ow - 1 + b >= 0 && ow - 1 + b < W) m |= 1u << (a * 3 + b);
      }
      rmask[ni] = m;
    }
    // W2 fragments of this wave: one contiguous stream per row block, fragment f = 12 i + j (range i, k-step j = 4 rb + ks)
    // at f KiB -- fragment-major packing of tspn_pack_conv2d_frag_bf16, [row block][part][tap][k-step]
    const unsigned woff = lane * 16;
    const char* const w2b0 = reinterpret_cast<const char*>(Wf2) + (int64_t)(2 * w4) * (9 * CCH * 4096) + woff;
    const char* const w2b1 = w2b0 + 9 * CCH * 4096;
    // B fragments of k-step j of a range: tap (ra, rb = j / 4), channels 16 (j % 4) ..  With ONE MFMA wave per SIMD every
    // vector instruction between two MFMAs is a cycle the matrix pipe may idle, so the fragment addresses are prepared once
    // per range: byte offset of channel group kh and the stride per channel group -- (stage, SLP 16) in the image,
    // (side region, 32) for slots 128 / 129, (zero slot, 0) for taps that fall off the image -- one v_mad per read.
    constexpr int D2 = 6;                                    // W2 ring, k-steps (fragments come from L2)
    f32x4 a2[D2][2];
    auto load_w2 = [&](int slot_, int f) {
      a2[slot_][0] = *reinterpret_cast<const f32x4*>(w2b0 + (int64_t)f * 1024);
      a2[slot_][1] = *reinterpret_cast<const f32x4*>(w2b1 + (int64_t)f * 1024);
    };
#pragma unroll
    for (int d = 0; d < D2; ++d) load_w2(d, d);
    role_barrier();                                          // [0]
    static_assert(NRNG % 2 == 0 && 24 % D2 == 0, "two ranges = 24 k-steps per unrolled body: compile-time ring slots");
    for (int i2 = 0; i2 < NRNG; i2 += 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = i2 + r;
        const int buf = i & (NST - 1), ra = i % 3;
        unsigned bo[3][4], bs[3][4];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            unsigned o = (unsigned)(buf * B_ST + (ni * 32 + li + rb) * 16), st = SLP * 16;
            if (ni == 3 && li + rb >= 32) { o = (unsigned)(EXTRA_OFF + buf * 256 + (li + rb - 32) * 16); st = 32; }
            if (!((rmask[ni] >> (3 * ra + rb)) & 1u)) { o = ZERO_OFF; st = 0; }
            bo[rb][ni] = o + kh * st;
            bs[rb][ni] = 2 * st;
          }
        auto read_b = [&](int j, bf16x8 (&b)[4]) {
          const int rb = j >> 2, ks = j & 3;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bs + bo[rb][ni] + ks * bs[rb][ni]);
        };
        bf16x8 bc[4], bn[4];
        read_b(0, bc);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          if (j + 1 < 12) read_b(j + 1, bn);                 // the next k-step's fragments fly under this k-step's MFMAs
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a2[(12 * r + j) % D2][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bc[ni], acc[mi][ni], 0, 0, 0);
          }
          if (12 * i + j + D2 < 12 * NRNG) load_w2((12 * r + j) % D2, 12 * i + j + D2);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bc[ni] = bn[ni];
          __builtin_amdgcn_sched_barrier(0);
        }
        role_barrier();                                      // [1 + i]
      }
    }
    // ---- h2 = relu(acc + b2) -> bf16 -> LDS (B-operand image [32 groups][SLP][8]) over the stages
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * (2 * w4 + mi) + 8 * q + 4 * kh;
        const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          bf16x4 v;
          v[0] = (__bf16)fmaxf(acc[mi][ni][4 * q] + bv.x, 0.f);
          v[1] = (__bf16)fmaxf(acc[mi][ni][4 * q + 1] + bv.y, 0.f);
          v[2] = (__bf16)fmaxf(acc[mi][ni][4 * q + 2] + bv.z, 0.f);
          v[3] = (__bf16)fmaxf(acc[mi][ni][4 * q + 3] + bv.w, 0.f);
          *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SLP + ni * 32 + li) * 16 + 8 * kh) = v;
        }
      }
    role_barrier();                                          // [1 + NRNG]: h2 complete
    // ---- phase 3: 1x1 expand, K = 256; the fp32 sums of a sub-pass go to the io wave of the same number through LDS
    const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;   // permuted W3 rows
    const char* const hb = Bs + (kh * SLP + li) * 16;
    for (int sp = 0; sp < NSUB; ++sp) {
      const int mb = 8 * w4 + (sp >> 1), nb = 2 * (sp & 1);
      const char* const w3b = reinterpret_cast<const char*>(Wf3) + (int64_t)mb * (CCH * 4096) + woff3;
      f32x16 c3[2];
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
#pragma unroll
        for (int e = 0; e < 16; ++e) c3[pj][e] = 0.f;
      constexpr int D3 = 8;
      f32x4 a3[D3];
#pragma unroll
      for (int d = 0; d < D3; ++d) a3[d] = *reinterpret_cast<const f32x4*>(w3b + d * 1024);
      bf16x8 hc[2], hn[2];
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) hc[pj] = *reinterpret_cast<const bf16x8*>(hb + (nb + pj) * 32 * 16);
#pragma unroll
      for (int k = 0; k < CM / 16; ++k) {
        if (k + 1 < CM / 16) {
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) hn[pj] = *reinterpret_cast<const bf16x8*>(hb + (2 * (k + 1) * SLP + (nb + pj) * 32) * 16);
        }
        const bf16x8 av = __builtin_bit_cast(bf16x8, a3[k % D3]);
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) c3[pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, hc[pj], c3[pj], 0, 0, 0);
        if (k + D3 < CM / 16) a3[k % D3] = *reinterpret_cast<const f32x4*>(w3b + (k + D3) * 1024);
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) hc[pj] = hn[pj];
        __builtin_amdgcn_sched_barrier(0);
      }
      // set sp & 1 was read by the io wave before barrier [1 + NRNG + sp] (its sub-pass sp - 2)
      char* const dst = xw + (sp & 1) * XCH_SET;
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(dst + (pj * 4 + q) * 1024) = f32x4{c3[pj][4 * q], c3[pj][4 * q + 1], c3[pj][4 * q + 2], c3[pj][4 * q + 3]};
      role_barrier();                                        // [2 + NRNG + sp]: sub-pass sp's sums are published
    }
  }
}

}  // namespace

extern "C" int tspn_bottleneck_tail_io_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM_,
                                            const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                            const float* bias3, const uint16_t* residual, uint16_t* out, void* stream) {
  const char* what = "tspn_bottleneck_tail_io_bf16";
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "%s: bad sizes", what);
  TSPN_REQUIRE(CM_ == CM, TSPN_EUNSUPPORTED, "%s: built for 256 bottleneck channels (got %lld)", what, (long long)CM_);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(h1 && frag2 && bias2 && frag3 && bias3 && residual && out, TSPN_EINVAL, "%s: null pointer", what);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(h1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) && al16(residual) && al16(out),
               TSPN_EUNSUPPORTED, "%s: operands must be 16-byte aligned", what);
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED, "%s: dimension too large", what);
  const int64_t npix = NB * H * W;
  const int64_t tiles = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles < (1LL << 31), TSPN_EUNSUPPORTED, "%s: grid too large", what);
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(tail_io_bf16_kernel), SMEM, what)) return rc;
  hipLaunchKernelGGL(tail_io_bf16_kernel, dim3((unsigned)tiles), dim3(THREADS), SMEM, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(h1), reinterpret_cast<const __bf16*>(frag2), bias2,
                     reinterpret_cast<const __bf16*>(frag3), bias3, reinterpret_cast<const __bf16*>(residual),
                     reinterpret_cast<__bf16*>(out), (int)H, (int)W, npix);
  return tspn::check_launch(what);
}
