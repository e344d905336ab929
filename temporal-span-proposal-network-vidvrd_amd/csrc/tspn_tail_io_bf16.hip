// Fused bottleneck tail at CM = 256 (the res4 chain of the C4 backbone: 43 % of its time) with the work split by ROLE
// instead of by phase (round 5):
//     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )        3x3 / pad 1 / stride 1, then 1x1 expand
// -- the arithmetic, tiling (128 linear pixels per workgroup, ring of linear h1 ranges, h2 image in LDS, W3 rows permuted
// at load) and contraction order of bottleneck_bf16_kernel<256> (tspn_bottleneck_bf16.hip), bit-identical results.
//
// Why (profiles/r5/tail_role_split.md): probe builds of that kernel without its residual loads and output stores need 76
// instead of 165 us per 18 frames, and a workgroup ALONE on the chip needs 56 us for its tile (36 without them) -- every
// vector-memory operation of a wave retires in order on one counter, so a residual row (HBM, 2 - 3 us) issued in front of a
// W3 fragment (L2) holds that fragment back and a store holds everything behind it: the wave that feeds the MFMA pipe spends
// half of its time waiting for memory it does not need.  Here a workgroup has EIGHT waves, two per SIMD:
//   waves 0-3 (compute): all MFMAs.  Their only vector-memory traffic is the weight stream from L2 (W2, W3 fragments
//                        straight into operand registers through rings); B operands come from LDS.
//   waves 4-7 (io):      everything that touches HBM.  Phase 2: the LDS-DMA of the h1 ranges (a ring of four stages, three
//                        ranges ahead, counted vmcnt) and, behind the last range, the residual rows of the first four expand
//                        sub-passes.  Phase 3: the epilogue -- io wave w takes the fp32 sums of compute wave w's sub-pass
//                        (32 channels x all 128 pixels) from an LDS exchange buffer, adds b3 and the residual, ReLU, rounds,
//                        stores -- with lane l on piece (l & 3) of pixel 16 t + (l >> 2), so every residual load and every
//                        store covers 64 contiguous bytes per lane quad (the MFMA accumulator layout gives 16 bytes of 64
//                        different lines per instruction, which the L1 serves at a third of the rate: tools/probes/
//                        tcp_line_coalesce_probe.hip); the results of an even sub-pass wait in registers for the odd one, so
//                        that the two halves of a 128-byte line are stored back to back.
// Phase 2: the two roles meet at one s_barrier per h1 range.  Phase 3: NO workgroup barrier -- compute wave w and io wave w
// hand the exchange buffer back and forth through two counters in LDS (published / consumed sub-passes; LDS operations of a
// wave execute in order, so a counter written behind the data is seen behind the data), the compute wave writes the sums of
// sub-pass e in the middle of the MFMAs of sub-pass e + 1 (two accumulator sets) and never waits unless its io wave is a
// whole sub-pass behind.  A sub-pass is one 32-row block x ALL 128 pixels: every W3 fragment enters the CU once.
// 147 KB of LDS, one workgroup per CU.
// Measured (profiles/r5/tail_role_split.md): 122 - 125 against 146 - 151 us per 18 frames (ten launches back to back), bit-identical.
#include <algorithm>
#include <type_traits>

#include "tspn_common.h"
#include "tspn_status.h"

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
constexpr int XCH_OFF = B3_OFF + C4 * 4;    // fp32 sums of a sub-pass, per wave pair: [128 pixels][32 channels + 4 floats of padding]
constexpr int XP = 32 * 4 + 16;             // bytes per pixel row: the 16 lanes of a 16-byte write touch 64 different banks
constexpr int XCH_WAVE = BN * XP;
constexpr int FLAG_OFF = XCH_OFF + 4 * XCH_WAVE;   // [4 wave pairs][published, consumed] sub-pass counters
constexpr int SMEM = FLAG_OFF + 64;
constexpr int NSUB = 8;                 // expand sub-passes per compute wave: 32 channels (one row block) x all 128 pixels --
                                        // every W3 fragment enters the CU ONCE (with 64-pixel sub-passes it came twice, and the
                                        // 1 MB of W3 per tile through the L1 took longer than the expand's MFMAs)
constexpr int RD = 4;                   // residual rows requested this many sub-passes ahead
constexpr int R3 = 16;                  // W3 ring, k-steps (= one sub-pass ahead)
static_assert(CCH * B_ST == NST * B_ST && SMEM <= 160 * 1024, "h2 image = the four stages; LDS budget");

// meet the other role: LDS traffic of this wave complete (reads consumed / writes landed), then the workgroup barrier
__device__ __forceinline__ void role_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// sub-pass counters in LDS: tspn_status.h (flag_set / flag_wait).  The wait is BOUNDED; a wave that gives up raises
// TSPN_FAULT_HANDOVER through the device status block and ENDS -- it never reads or overwrites an exchange buffer it was
// not handed (round 6; until then it fell through with wrong data and only a parity test would have noticed).
using tspn_dev::flag_set;
using tspn_dev::flag_wait;

__global__ __launch_bounds__(THREADS, 1) void tail_io_bf16_kernel(
    const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2, const float* __restrict__ bias2,
    const __bf16* __restrict__ Wf3, const float* __restrict__ bias3, const __bf16* __restrict__ residual,
    __bf16* __restrict__ out, int H, int W, int64_t npix, int32_t* __restrict__ status) {
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
  if (tid < 64) reinterpret_cast<float*>(Bs + ZERO_OFF)[tid] = 0.f;              // 16 zero slots; published by the first barrier
  if (tid >= 64 && tid < 80) reinterpret_cast<int*>(Bs + FLAG_OFF)[tid - 64] = 0;  // likewise
  for (int i = tid; i < CM; i += THREADS)                                          // b3 -> LDS, likewise
    *reinterpret_cast<float4*>(Bs + B3_OFF + 16 * i) = *reinterpret_cast<const float4*>(bias3 + 4 * i);
  constexpr unsigned OOB = 0x80000000u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs;
  const unsigned f_pub = lds0 + FLAG_OFF + 8 * w4, f_con = f_pub + 4;            // counters of this wave pair
  char* const xw = Bs + XCH_OFF + w4 * XCH_WAVE;                                   // exchange buffer of this wave pair

  // The two roles run SEPARATE straight-line programs with the same s_barrier sequence (1 + NRNG + 1) instead of one loop
  // nest with role branches inside: with the branches inside, the 128 accumulator registers are loop-carried through the io
  // path as well and hipcc copies and spills them around every branch (1 200 spilled registers in that form).
  if (io) {
    // ================================================================ io waves
    // ---- phase 2: the ranges.  Range i = 3 c + ra holds pixels n0 + (ra - 1) W - 1 .. + 129 of channel part c; wave w4
    // stages the pixel (slot) 64 (w4 & 1) + lane, channel groups bg, bg + 2, bg + 4, bg + 6 (bg = w4 >> 1); slots 128, 129 go
    // to a side region (every io wave issues that piece -- same bytes, same place -- so that each has FIVE pieces per range)
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
    // ---- phase 3 addresses: sub-pass e of wave pair w4 = channels 256 w4 + 32 e .. + 31 (half a 128-byte line) of all 128
    // pixels; item t of a lane = piece (lane & 3) (8 channels) of pixel 16 t + (lane >> 2): a quad of lanes = 64 contiguous
    // bytes.  The results of an even sub-pass wait in registers and are stored together with the odd one's: the two halves of a
    // line leave back to back (L2 hands part-written lines to the fabric as they are, profiles/r3)
    const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(residual) + n0 * C4, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(out + n0 * C4, 0, 0x7fffffff, 0x00020000);
    // The quad -> pixel map is a permutation chosen for the LDS side (round 6): a ds_read_b128 is served in the lane
    // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) and a quad covers 64 of a pixel row's 144 bytes in two
    // reads of alternate 16-byte slots, so with pp = lane >> 2 the quads of a group met on the same banks (two-way: 2 050
    // of the 4 230 conflict cycles per tile, SQ_LDS_BANK_CONFLICT in profiles/r5/probes/tail_io_sq_counters.txt).  With
    // the quads of a group on pixels {p, p + 1, p + 8, p + 9} their row bases (9 p mod 16 slots) are 0, 9, 8, 1 -- even
    // and odd slots interleave, all sixteen distinct.  Global side unchanged: a quad still owns 64 contiguous bytes.
    const int pc = lane & 3, pp = (int)((0xFDCE5764B98A1320ull >> (4 * (lane >> 2))) & 15);
    const unsigned lvo = (unsigned)(pp * C4 * 2 + 16 * pc);                       // the lane's part of every global offset
    const int64_t left = npix - n0 - pp;                                           // pixels 16 t below this exist
    const int plimit = left > BN ? BN : (int)left;
    auto soff_of = [&](int e, int t) { return (16 * t * C4 + 256 * w4 + 32 * e) * 2; };
    auto voff_of = [&](int e, int t) { (void)e; return (int)(16 * t < plimit ? lvo : OOB); };
    bf16x8 res[RD][8];                                       // residual pieces in flight [sub-pass % RD][item]
    auto res_issue = [&](int slot_, int e) {                // slot_ = e % RD, compile-time at every call site
#pragma unroll
      for (int t = 0; t < 8; ++t)
        res[slot_][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, voff_of(e, t), soff_of(e, t), 0));
    };

#pragma unroll
    for (int i = 0; i < DIST; ++i) stage_r(i, i);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");        // range 0 has landed (five pieces per range)
    role_barrier();                                          // [0]
    for (int i = 0; i + DIST + 1 < NRNG; ++i) {
      // stage (i + DIST) % NST = (i - 1) % NST was read in the previous interval; its barrier lies behind us
      stage_r((i + DIST) & (NST - 1), i + DIST);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // range i + 1 has landed before the barrier publishes it
      role_barrier();                                        // [1 + i]
    }
    stage_r((NRNG - 1) & (NST - 1), NRNG - 1);               // the last range ...
#pragma unroll
    for (int e = 0; e < RD; ++e) res_issue(e, e);            // ... and behind it 32 residual pieces: the waits count them in
    static_assert(10 + 8 * RD <= 63, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + 8 * RD) : "memory");
    role_barrier();                                          // [NRNG - 3]
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 + 8 * RD) : "memory");
    role_barrier();                                          // [NRNG - 2]
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * RD) : "memory");
    role_barrier();                                          // [NRNG - 1]
    role_barrier();                                          // [NRNG]: the last range has been read
    role_barrier();                                          // [1 + NRNG]: h2 complete (nothing of ours depends on it)

    u32x4_t keep = {}, keep2 = {};
    u32x4_t held[8];                                         // results of the even sub-pass of a pair
    static_assert(NSUB % RD == 0 && RD % 2 == 0, "the sub-pass loop is unrolled by the residual ring's depth; pairs inside");
    for (int e0 = 0; e0 < NSUB; e0 += RD) {
#pragma unroll
      for (int u = 0; u < RD; ++u) {
        const int e = e0 + u;
        float bv[8];
        {
          const char* const b3s = Bs + B3_OFF + (256 * w4 + 32 * e + 8 * pc) * 4;
          const float4 t0 = *reinterpret_cast<const float4*>(b3s), t1 = *reinterpret_cast<const float4*>(b3s + 16);
          bv[0] = t0.x; bv[1] = t0.y; bv[2] = t0.z; bv[3] = t0.w; bv[4] = t1.x; bv[5] = t1.y; bv[6] = t1.z; bv[7] = t1.w;
        }
        flag_wait(f_pub, e + 1, status, (int)blockIdx.x);                             // the sums of sub-pass e are in the exchange buffer
        f32x4 sv[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const char* const src = xw + (16 * t + pp) * XP + 32 * pc;
          sv[t][0] = *reinterpret_cast<const f32x4*>(src);
          sv[t][1] = *reinterpret_cast<const f32x4*>(src + 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_set(f_con, e + 1);                              // the buffer may take sub-pass e + 1
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf((sv[t][j >> 2][j & 3] + bv[j]) + (float)res[u][t][j], 0.f);
          const u32x4_t o4 = __builtin_bit_cast(u32x4_t, o);
          if ((u & 1) == 0) {
            held[t] = o4;
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(held[t], rsrc_out, voff_of(e - 1, t), soff_of(e - 1, t), 0);
            __builtin_amdgcn_raw_buffer_store_b128(o4, rsrc_out, voff_of(e, t), soff_of(e, t), 0);
            // store-data hazard (tools/lint_store_hazard.py, profiles/r5/bottleneck_block_study.md §3): the data registers
            // of a store stay live until the next stores have been issued
            asm volatile("" ::"v"(keep), "v"(keep2));
            keep = o4;
            keep2 = held[t];
          }
        }
        if (e + RD < NSUB) res_issue(u, e + RD);             // its ring slot is free now
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n s_nop 15" ::"v"(keep), "v"(keep2));
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
    {
      // (round 6) n0 is wave-uniform: ONE division for the tile's first pixel, the lane's part in 32 bits with a
      // float reciprocal + correction, the nine tap bits in closed form.  The per-lane 64-bit divisions and the nine-tap
      // loop this replaces were ~1 500 vector instructions in front of the first MFMA of every tile.
      const int HW = H * W;                                  // < 2^30 (launcher)
      const int r0 = (npix >> 31) == 0 ? (int)((unsigned)n0 % (unsigned)HW) : (int)(n0 % HW);
      const float rcpW = 1.0f / (float)W;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        int rr = r0 + ni * 32 + li;                          // pixel index inside ITS image, after the wrap
        if (HW >= BN) rr -= rr >= HW ? HW : 0;
        else rr = (int)((unsigned)rr % (unsigned)HW);
        int oh = (int)((float)rr * rcpW);                    // within one of rr / W for rr < 2^30, H < 2^20
        int ow = rr - oh * W;
        if (ow < 0) { --oh; ow += W; }
        if (ow >= W) { ++oh; ow -= W; }
        const unsigned cm = (ow >= 1 ? 1u : 0u) | 2u | (ow <= W - 2 ? 4u : 0u);
        const unsigned m = (oh >= 1 ? cm : 0u) | (cm << 3) | (oh <= H - 2 ? cm << 6 : 0u);
        rmask[ni] = (n0 + ni * 32 + li < npix) ? m : 0u;
      }
    }
    // W2 fragments of this wave: one contiguous stream per row block, fragment f = 12 i + j (range i, k-step j = 4 rb + ks)
    // at f KiB -- fragment-major packing of tspn_pack_conv2d_frag_bf16, [row block][part][tap][k-step]
    const unsigned woff = lane * 16;
    const char* const w2b0 = reinterpret_cast<const char*>(Wf2) + (int64_t)(2 * w4) * (9 * CCH * 4096) + woff;
    const char* const w2b1 = w2b0 + 9 * CCH * 4096;
    // B fragments of k-step j of a range: tap (ra, rb = j / 4), channels 16 (j % 4) ..  With ONE MFMA wave per SIMD every
    // vector instruction between two MFMAs is a cycle the matrix pipe may idle, so the fragment addresses are prepared once
    // per range: byte offset of channel group kh and the stride per channel group -- (stage, SLP 16) in the image,
    // (side region, 32) for slots 128 / 129, (zero slot, 0) for taps that fall off the image -- one v_mad per read; the
    // reads of k-step j + 1 are issued IN FRONT of the MFMAs of k-step j (fenced: left alone, the scheduler moves them
    // behind half of the MFMAs and the next k-step waits for LDS).
    constexpr int D2 = 4;                                    // W2 ring, k-steps (fragments come from L2); a ring of six spills
    f32x4 a2[D2][2];
    auto load_w2 = [&](int slot_, int f) {
      a2[slot_][0] = *reinterpret_cast<const f32x4*>(w2b0 + (int64_t)f * 1024);
      a2[slot_][1] = *reinterpret_cast<const f32x4*>(w2b1 + (int64_t)f * 1024);
    };
#pragma unroll
    for (int d = 0; d < D2; ++d) load_w2(d, d);
    role_barrier();                                          // [0]
    // (fragment addresses of a tap = four k-steps are prepared under the MFMAs of the tap before it, two sets alternating)
    unsigned bo[2][4], bs[2][4];                             // [tap parity][pixel block]
    auto tap_addr = [&](int i, int rb, unsigned (&o_)[4], unsigned (&s_)[4]) {
      const int buf = i & (NST - 1), ra = i % 3;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        unsigned o = (unsigned)(buf * B_ST + (ni * 32 + li + rb) * 16), st = SLP * 16;
        if (ni == 3 && li + rb >= 32) { o = (unsigned)(EXTRA_OFF + buf * 256 + (li + rb - 32) * 16); st = 32; }
        // (a tap off the image reads the zero slot ON THE BANKS its pixel slot would have used: one shared zero slot made
        // every fragment read with a masked lane two-way conflicted with the lane whose slot is = 0 mod 16)
        if (!((rmask[ni] >> (3 * ra + rb)) & 1u)) { o = (unsigned)(ZERO_OFF + ((li + rb) & 15) * 16); st = 0; }
        o_[ni] = o + kh * st;
        s_[ni] = 2 * st;
      }
    };
    tap_addr(0, 0, bo[0], bs[0]);
    static_assert(NRNG % 2 == 0 && 24 % D2 == 0, "two ranges = 24 k-steps = six taps per unrolled body: compile-time ring slots and tap parities");
    for (int i2 = 0; i2 < NRNG; i2 += 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = i2 + r;
        auto read_b = [&](int j, bf16x8 (&b)[4]) {
          const int rb = j >> 2, ks = j & 3, par = (3 * r + rb) & 1;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bs + bo[par][ni] + ks * bs[par][ni]);
        };
        bf16x8 bb[2][4];
        read_b(0, bb[0]);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          // the next k-step's fragments fly under this k-step's MFMAs: one read (and its address arithmetic) behind each of
          // the first four MFMAs, the two W2 fragments of k-step j + D2 behind the next two
          if ((j & 3) == 0) {                                // first k-step of a tap: the addresses of the tap behind it
            const int rb = j >> 2;
            if (rb < 2) tap_addr(i, rb + 1, bo[(3 * r + rb + 1) & 1], bs[(3 * r + rb + 1) & 1]);
            else if (i + 1 < NRNG) tap_addr(i + 1, 0, bo[(3 * r + 3) & 1], bs[(3 * r + 3) & 1]);
          }
          if (j + 1 < 12) read_b(j + 1, bb[(j + 1) & 1]);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a2[(12 * r + j) % D2][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[j & 1][ni], acc[mi][ni], 0, 0, 0);
          }
          if (12 * i + j + D2 < 12 * NRNG) load_w2((12 * r + j) % D2, 12 * i + j + D2);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // VALU
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
          }
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        role_barrier();                                      // [1 + i]
      }
    }
    // ---- the first R3 k-steps of the W3 stream fly while h2 is written
    // sub-pass sp: row block 8 w4 + sp, all four pixel blocks; k-step k of row block mb at (mb CCH 4 + k) KiB
    const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;   // permuted W3 rows
    const char* const w3w = reinterpret_cast<const char*>(Wf3) + (int64_t)(8 * w4) * (CCH * 4096) + woff3;
    f32x4 a3[R3];
    auto load_w3 = [&](int slot_, int sp, int k) {
      a3[slot_] = *reinterpret_cast<const f32x4*>(w3w + (int64_t)sp * (CCH * 4096) + k * 1024);
    };
#pragma unroll
    for (int d = 0; d < R3; ++d) load_w3(d, 0, d);
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
    // ---- phase 3: 1x1 expand, K = 256.  Sub-pass sp accumulates in set sp & 1; the sums of sub-pass sp - 1 go to the
    // exchange buffer in the middle of sub-pass sp.
    const char* const hb = Bs + (kh * SLP + li) * 16;
    char* const xl = xw + li * XP + 64 * kh;                 // + (32 nj) XP + 16 q
    auto write_sums = [&](f32x16 (&c)[4], int e) {          // sums of sub-pass e -> exchange buffer, then publish
      flag_wait(f_con, e, status, (int)blockIdx.x);                                   // the io wave has taken sub-pass e - 1 out of it
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(xl + nj * 32 * XP + 16 * q) = f32x4{c[nj][4 * q], c[nj][4 * q + 1], c[nj][4 * q + 2], c[nj][4 * q + 3]};
      flag_set(f_pub, e + 1);
    };
    f32x16 cA[4], cB[4];
    bf16x8 hh[2][4];
    auto read_h = [&](int k, bf16x8 (&b)[4]) {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) b[nj] = *reinterpret_cast<const bf16x8*>(hb + ((2 * k) * SLP + nj * 32) * 16);
    };
    auto ksteps = [&](f32x16 (&c)[4], int sp, auto k0_tag, auto k1_tag) {     // k-steps k0 .. k1 - 1 of sub-pass sp
      constexpr int k0 = decltype(k0_tag)::value, k1 = decltype(k1_tag)::value;
#pragma unroll
      for (int k = k0; k < k1; ++k) {
        read_h((k + 1) & 15, hh[(k + 1) & 1]);               // the next k-step's h2 fragments in front of this k-step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 av = __builtin_bit_cast(bf16x8, a3[k % R3]);
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)       // the first k-step starts from a constant zero: no 64 v_mov per sub-pass
          c[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, hh[k & 1][nj], k == 0 ? z : c[nj], 0, 0, 0);
        if (sp + 1 < NSUB) load_w3(k % R3, sp + 1, k);       // R3 = 16: the same k-step of the next sub-pass
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(R3 == 16, "the ring holds exactly one sub-pass");
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    read_h(0, hh[0]);
    ksteps(cA, 0, I0{}, I16{});
    for (int sp = 1; sp < NSUB; sp += 2) {
      ksteps(cB, sp, I0{}, I8{});
      write_sums(cA, sp - 1);
      ksteps(cB, sp, I8{}, I16{});
      if (sp + 1 < NSUB) {
        ksteps(cA, sp + 1, I0{}, I8{});
        write_sums(cB, sp);
        ksteps(cA, sp + 1, I8{}, I16{});
      }
    }
    write_sums(cB, NSUB - 1);
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
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && H * W < (1LL << 30), TSPN_EUNSUPPORTED, "%s: dimension too large", what);
  const int64_t npix = NB * H * W;
  const int64_t tiles = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles < (1LL << 31), TSPN_EUNSUPPORTED, "%s: grid too large", what);
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(tail_io_bf16_kernel), SMEM, what)) return rc;
  hipLaunchKernelGGL(tail_io_bf16_kernel, dim3((unsigned)tiles), dim3(THREADS), SMEM, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(h1), reinterpret_cast<const __bf16*>(frag2), bias2,
                     reinterpret_cast<const __bf16*>(frag3), bias3, reinterpret_cast<const __bf16*>(residual),
                     reinterpret_cast<__bf16*>(out), (int)H, (int)W, npix, tspn::status_device_ptr());
  return tspn::check_launch(what);
}
