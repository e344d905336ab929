// Fused tail of a ResNet bottleneck block on bf16 operands (BASELINE cfg5 backbone):
//     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )        3x3 / pad 1 / stride 1, then 1x1 expand
// in ONE kernel (detectron2 BottleneckBlock.forward after conv1, modeling/backbone/resnet.py; FrozenBN folded into
// W / b by the caller).  Round 2 ran the two convolutions as separate launches: the 3x3 is MFMA-bound, the 1x1
// expand (K = 64..256, an output four times its input plus a residual read) is bound by its 8-byte-per-lane
// epilogue traffic (3.3 TB/s at res4, 374 TFLOP/s), and h2 made a round trip through HBM in between.  Here:
//   phase 2  the 3x3 conv as conv2d_nhwc_bf16_kernel does it in its range mode (implicit GEMM, M = CM output
//            channels, N = 128 pixels per workgroup, K = CM / 64 channel parts x 9 taps x 64 channels; fragment-major
//            weights straight into MFMA operand registers, counted vmcnt, bare s_barrier) with the WHOLE channel
//            range of the tile in one workgroup: wave (wm, wn) owns rows [64 wm, 64 wm + 64) x pixel blocks
//            [NI wn, NI wn + NI), (WM, WN) = (4, 1) / (2, 2) / (1, 4) for CM = 256 / 128 / 64.  The x operand goes
//            through LDS as LINEAR RANGES of 130 pixels of a 64-channel part: tap (a, b) of pixel n is pixel
//            n + (a - 1) W + (b - 1), so one range serves the three taps of a row at slot offsets 0..2, taps off the
//            image zeroed at the read by the lane's tap mask -- a ring of four range stages at CM = 256 (12 chunks
//            of 12 k-steps per tile), all three ranges of a part staged at once at CM <= 128 (comments at the code);
//   h2       relu(acc + b2) is rounded to bf16 ONCE (the same rounding point as the unfused chain) and written
//            into the now idle stage memory in the B-operand layout [CM / 8 groups][132 slots][8 bf16];
//   phase 3  the 1x1 expand (M = 4 CM rows, K = CM) as eight sub-passes per wave on two alternating accumulator
//            sets: B fragments from the h2 image in LDS (conflict-free 16-byte reads), W3 fragments from L2 as one
//            continuous stream through a 4-deep register ring; the epilogue of sub-pass i is fed into the MFMA
//            gaps of sub-pass i + 1, half a group per two k-steps, residual rows requested two groups ahead;
//   epilogue the lanes fetch the W3 fragments with their ROWS PERMUTED inside each 32-row block, so that the MFMA
//            layout (registers 4 q + j = rows 8 q + 4 half + j) becomes 16 consecutive channels per lane: residual
//            read, ReLU, the single rounding and the store are 32 contiguous bytes per lane with no LDS transpose
//            and no cross-lane movement, and the two 32-row blocks of a 128-byte line are finished back to back --
//            a launch writes exactly its output (WRITE_SIZE 59.9 MB for 59.0 MB at res4; 101 MB before, when L2
//            handed part-written lines to the fabric).
// Same contraction order and the same rounding points as conv2d_nhwc_bf16 applied twice: bit-identical results
// (tests/test_gpu_roi_head.py).  Two workgroups per CU (67.6 KB of LDS at CM = 256): while one is in its
// memory-heavy epilogue the other runs MFMAs.
#include <algorithm>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int BN = 128;                 // pixels per workgroup
constexpr int KC = 64;                  // channels per chunk
constexpr int SLP = 132;                // padded pixel slots per channel group
constexpr int B_ST = 8 * SLP * 16;      // bytes per x stage = per 64 channels of the h2 image

template <int OFF>
__device__ __forceinline__ void load_wfrag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait_w(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}
// NEXT (CM = 256, round 4): the kernel also computes conv1 of the FOLLOWING block on its own output tile,
//     h1n = relu(W1n . out + b1n)        1x1, K = 4 CM = 1024 -> CM rows,
// so that the 4 CM-channel map is read once per block (as the next block's residual) instead of twice.  Phase 3 then
// walks the output channels PASS-MAJOR (the four waves together finish channels [256 p, 256 p + 256) of all 128
// pixels in pass p), writes the rounded bf16 results of a pass into a second LDS image [32 groups][132 slots][8] next to
// the h2 image as well as to HBM, and after every pass the workgroup contracts that 256-channel chunk with W1n: the
// inner loop of phase 2 again (wave = 64 rows x 128 pixels, eight accumulators, the registers phase 2 used; fragments
// of W1n straight from L2 through a ring of four k-steps), k-steps in the natural channel order 0..1023 on one
// accumulator chain = the contraction order of the stand-alone conv1 launch: bit-identical h1n.  The MFMAs of a
// chunk run while the stores of the pass drain and the residual rows of the next pass land.  136 KB of LDS and
// about 400 registers per lane: one workgroup per CU, which is what a launch of 8 frames (225 tiles) gives anyway.
template <int CM, bool NEXT>
__global__ __launch_bounds__(THREADS, (NEXT ? 1 : (CM == 64 ? 3 : 2))) void bottleneck_bf16_kernel(
    const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2, const float* __restrict__ bias2,
    const __bf16* __restrict__ Wf3, const float* __restrict__ bias3, const __bf16* __restrict__ residual,
    __bf16* __restrict__ out, int H, int W, int64_t npix, const __bf16* __restrict__ Wf1n,
    const float* __restrict__ bias1n, __bf16* __restrict__ h1n) {
  static_assert(!NEXT || CM == 256, "the fused conv1 of the next block is built for CM = 256");
  constexpr int MI = 2;                                   // 32-row blocks per wave
  constexpr int WM = CM / 64, WN = 4 / WM, NI = 4 / WN;   // waves along rows / pixels, 32-pixel blocks per wave
  constexpr int CCH = CM / KC;                            // 64-channel chunks per tap = chunks of phase 3
  constexpr int NCHUNKS = 9 * CCH;
  constexpr int C4 = 4 * CM;
  // x stages: a ring of NST, filled NST - 1 chunks ahead (the DMA latency under load is 2-4 chunk times: with one
  // chunk of lookahead the MFMA pipe idled more than half of phase 2); NST = 4 costs nothing at CM = 256, where the
  // h2 image needs the same 67.6 KB
  constexpr int NST = CM == 256 ? 4 : 3;
  constexpr int DIST = NST - 1;
  extern __shared__ __attribute__((aligned(16))) char Bs[];   // max(NST stages, h2 image) = max(NST, CCH) * B_ST
  // NEXT: two chunk images [32 groups][SLPC slots][8 bf16] behind the stages / the h2 image, one per sub-pass parity
  // a 16-byte slot of zeros behind the side regions: fragment reads of taps that fall off the image are redirected to
  // it BY ADDRESS (round 4; round 3 zeroed the loaded registers with four v_cndmask per fragment, which made the
  // compiler wait for every LDS read right behind its issue: three or four exposed LDS latencies per k-step with one
  // wave per SIMD -- the "issue structure" that kept phase 2 at half the MFMA rate with all memory traffic removed)
  constexpr int ZERO_OFF = (NST > CCH ? NST : CCH) * B_ST + 1024;
  // b3 (4 CM floats) lives in LDS (round 4): read from global memory inside the epilogue of phase 3, the two float4 of
  // every half group sat on the wave's one in-order vector-memory counter -- the compiler's wait for them was
  // `s_waitcnt vmcnt(0)`, i.e. a full drain of the W3 ring, the residual ring and the output stores 64 times per tile
  constexpr int BIAS_OFF = ZERO_OFF + 256;
  constexpr int CHUNK_OFF = BIAS_OFF + 4 * CM * 4;
  constexpr int SLPC = 68;                                  // 64 pixels + padding: conflict-free 16-byte reads like SLP
  constexpr int CHUNK_BYTES = (CM / 8) * SLPC * 16;         // CM channels of 64 pixels: CM / 8 groups x SLPC slots x 16 B

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;      // consecutive pixel tiles stay on one XCD (shared halo rows)
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t n0 = (int64_t)wg * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, kh = lane >> 5;
  const unsigned woff = lane * 16;
  if (tid < 4) reinterpret_cast<float*>(Bs + ZERO_OFF)[tid] = 0.f;   // published by the first barrier of phase 2
  for (int i = tid; i < CM; i += THREADS)                              // b3 -> LDS, published by the same barrier
    *reinterpret_cast<float4*>(Bs + BIAS_OFF + 16 * i) = *reinterpret_cast<const float4*>(bias3 + 4 * i);
  const char* const zslot = Bs + ZERO_OFF;

  // ---------------------------------------------------------------- phase 2: 3x3 conv, K = 9 taps x CM
  const char* wbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    wbase[mi] = reinterpret_cast<const char*>(Wf2) + (int64_t)(MI * wm + mi) * NCHUNKS * 4096;
  // x pieces: the four pieces of a lane belong to ONE pixel (slot), channel groups bg, bg + 2, bg + 4, bg + 6
  const int slot = 64 * (wave & 1) + lane;
  const int bg = wave >> 1;
  // The h1 ranges are staged by BUFFER loads (buffer_load_dwordx4 ... offen lds; round 4): one SGPR descriptor based at the
  // first pixel a range of this tile can start at, a 32-bit lane offset, the 64-channel part as the scalar offset; a range
  // pixel outside the tensor is an offset beyond the descriptor's range and arrives as ZEROS (tools/probes/
  // buffer_lds_oob_probe.hip): no zero page, no choice between two 64-bit pointers per lane, and the buffer form is the
  // cheaper one to issue beside MFMAs (tools/probes/lds_dma_issue_probe.hip)
  constexpr unsigned OOB = 0x80000000u;
  const int64_t rbase = n0 - W - 1 > 0 ? n0 - W - 1 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_h1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(h1) + rbase * CM, 0, 0x7fffffff, 0x00020000);
  auto bglds16 = [&](unsigned voff, int soff, char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_h1, (__attribute__((address_space(3))) void*)l, 16, (int)voff, soff, 0, 0);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x4 a[4][MI];      // weight fragments of the four k-steps of a chunk
  auto read_b = [&](const char* Bb, int g2, bf16x8 (&b)[NI]) {   // fragments of channel groups g2 + kh of NI pixel blocks
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bb + (g2 * SLP + ni * 32) * 16);
  };
  auto mfma_step = [&](f32x16 (&c)[MI][NI], const f32x4 (&aw)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const bf16x8 av = __builtin_bit_cast(bf16x8, aw[mi]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[ni], c[mi][ni], 0, 0, 0);
    }
  };
  auto load_step = [&](auto ks_tag) {
    constexpr int KS = decltype(ks_tag)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) load_wfrag<1024 * KS>(a[KS][mi], woff, wbase[mi]);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  auto bump = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) wbase[mi] += 4096;
  };

#ifndef TSPN_BT_ONCE_MAX
#define TSPN_BT_ONCE_MAX 128       // largest CM that takes the once-staged form of phase 2 (probe knob)
#endif
  if constexpr (CM > TSPN_BT_ONCE_MAX) {
    // ---- CM = 256: a ring of linear RANGES.  Tap (a, b) of pixel n is pixel n + (a - 1) W + (b - 1) of the
    // channels-last map, so ONE staged range -- pixels n0 + (a - 1) W - 1 .. + 129 of a 64-channel part -- serves the
    // three taps (a, 0..2) at slot offsets 0..2: a chunk of the ring is a range = 12 k-steps = 96 MFMAs per wave, a
    // tile streams 12 ranges (203 KB) instead of 36 tap chunks (576 KB) through the LDS-DMA and meets at 12 barriers
    // instead of 36 (probe builds: the x DMA cost 13 of the phase's 80 us per 16 frames, profiles/r3/
    // bottleneck_tail_ablation.md).  Slots 128, 129 of a range live in a 256-byte side region per stage; taps that fall
    // off the image are zeroed AT THE READ by the lane's own tap mask (a range runs across row and image boundaries).
    // Same contraction order as before: channel part by part, its nine taps in a row.
    constexpr int NRNG = 3 * CCH;
    static_assert(NRNG > DIST && NI == 4 && WN == 1, "range ring");
    char* const extra = Bs + NST * B_ST;                     // [stage][8 groups][2 slots] x 16 B
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
    unsigned rmask[NI];                                      // of the pixels this lane reads as its B columns
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) rmask[ni] = tap_mask(n0 + ni * 32 + li);
    auto stage_r = [&](int buf, int i) {                     // range i = 3 c + ra: four pieces per wave (+ one: wave 0)
      const int c = i / 3, ra = i - 3 * c;
      const int soff = c * KC * 2;
      const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + slot;
      const unsigned voff = (q >= 0 && q < npix) ? (unsigned)((q - rbase) * CM * 2 + 16 * bg) : OOB;
      char* dst = Bs + buf * B_ST + (bg * SLP + 64 * (wave & 1)) * 16;
#pragma unroll
      for (int p = 0; p < 4; ++p) bglds16(voff + 32 * p, soff, dst + 2 * p * SLP * 16);
      if (wave == 0 && lane < 16) {                          // slots 128, 129: [group][2]
        const int g = lane >> 1, e = lane & 1;
        const int64_t q2 = n0 + (int64_t)(ra - 1) * W - 1 + 128 + e;
        const unsigned voff2 = (q2 >= 0 && q2 < npix) ? (unsigned)((q2 - rbase) * CM * 2 + 16 * g) : OOB;
        bglds16(voff2, soff, extra + buf * 256);             // the DMA adds lane * 16
      }
    };
#pragma unroll
    for (int i = 0; i < DIST; ++i) stage_r(i, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_step(K0{}); load_step(K1{}); load_step(K2{}); load_step(K3{});
    bump();
    __builtin_amdgcn_sched_barrier(0);

    // fragment of channel groups g2 + kh of the NI pixel blocks at slot offset rb of a stage, masked by tap 3 ra + rb
    auto read_r = [&](int buf, int tap, int rb, int g2, bf16x8 (&b)[NI]) {
      const char* Bb = Bs + buf * B_ST + ((g2 + kh) * SLP + li + rb) * 16;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const char* bp = Bb + ni * 32 * 16;
        if (ni == 3) bp = (li + rb >= 32) ? extra + buf * 256 + ((g2 + kh) * 2 + (li + rb - 32)) * 16 : bp;
        if (!((rmask[ni] >> tap) & 1u)) bp = zslot;          // the tap falls off the image: read zeros
        b[ni] = *reinterpret_cast<const bf16x8*>(bp);
      }
    };
    // one round = the four k-steps of tap (ra, rb) of range i.  VMEM issue order as in the tap ring: [range i + DIST:
    // 4 pieces, first round only] a0' | a1' | a2' | a3' (MI loads each, the weights four k-steps ahead); counts =
    // YOUNGER operations at each wait (wave 0 issues one piece more: its waits are one operation stricter than needed).
    // The range requested DIST - 1 >= 1 ranges ago precedes every weight load of this range: in-order VMEM return makes
    // any of these waits the wait for it as well.
    auto round_body = [&](int i, int buf, int ra, auto rb_tag, auto stage_tag, auto more_tag, auto last_tag) {
      constexpr int rb = decltype(rb_tag)::value;
      constexpr bool STAGE = decltype(stage_tag)::value, MORE = decltype(more_tag)::value, LAST = decltype(last_tag)::value;
      constexpr int NX = STAGE ? 4 : 0, R = MORE ? MI : 0, L = MI;
      const int tap = 3 * ra + rb;
      bf16x8 b0[NI] = {}, b1[NI] = {};
      wait_w<3 * L>(a[0][0], a[0][1]);
      if (STAGE) stage_r(buf >= 1 ? buf - 1 : NST - 1, i + DIST);      // the stage range i - 1 has just left
      __builtin_amdgcn_sched_barrier(0);
      read_r(buf, tap, rb, 0, b0);
      read_r(buf, tap, rb, 2, b1);
      mfma_step(acc, a[0], b0);
      if (MORE) load_step(K0{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<2 * L + NX + R>(a[1][0], a[1][1]);
      read_r(buf, tap, rb, 4, b0);
      mfma_step(acc, a[1], b1);
      if (MORE) load_step(K1{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<L + NX + 2 * R>(a[2][0], a[2][1]);
      read_r(buf, tap, rb, 6, b1);
      mfma_step(acc, a[2], b0);
      if (MORE) load_step(K2{});
      __builtin_amdgcn_sched_barrier(0);
      wait_w<NX + 3 * R>(a[3][0], a[3][1]);
      mfma_step(acc, a[3], b1);
      if (MORE) { load_step(K3{}); bump(); }
      __builtin_amdgcn_sched_barrier(0);
      if (LAST) {      // every LDS read of this range has returned; the next range has landed (see above)
        if (MORE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
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
      round_body(i, buf, ra, R2{}, F{}, F{}, T{});            // ends with a barrier: nobody reads the stages any more
    }
  } else {
    // ---- CM <= 128: a chunk (64 channels of one tap) is only 8 / 16 MFMAs per wave -- a barrier, a DMA hand-over and
    // an L2 round trip for the weights per chunk left the MFMA pipe idle most of phase 2 (CM = 64: 109 us per 8 frames
    // of 720p against 28 at the rate of the wide layers).  Here the operand of ALL nine taps of a 64-channel half is
    // staged ONCE: tap (a, b) of pixel n is pixel n + (a - 1) W + (b - 1) of the channels-last map, so three LINEAR
    // ranges of 130 pixels -- stage a = pixels n0 + (a - 1) W - 1 .. + 129 -- hold everything (the same 16.9 KB per
    // stage as a tap's chunk; slots 128, 129 in a 768-byte side region: the DMA writes 64 consecutive slots per
    // instruction); a tap reads its stage at a slot offset, and taps that fall off the image are zeroed AT THE READ by
    // the lane's own tap mask (the ranges run across row and image boundaries).  One barrier, then 36 k-steps back to
    // back per half; the weights stream through a ring of RW k-steps, counted vmcnt.
    constexpr int RW = CM == 64 ? 12 : (CM == 128 ? 8 : 4);
    static_assert(NST >= 3 && MI == 2, "once-staged form");
    char* const extra = Bs + 3 * B_ST;
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
    unsigned rmask[NI];                                      // of the pixels this lane reads as its B columns
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) rmask[ni] = tap_mask(n0 + (wn * NI + ni) * 32 + li);
    f32x4 aw[RW][MI];
    const char* wl[MI];
    auto load_k = [&](auto j_tag) {                          // k-step j of the 36 of a half into its ring slot
      constexpr int j = decltype(j_tag)::value;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) load_wfrag<1024 * (j % 4)>(aw[j % RW][mi], woff, wl[mi]);
      if constexpr (j % 4 == 3) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) wl[mi] += 4096;          // the next tap of this half
      }
    };
    auto step = [&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      constexpr int tap = j / 4, ks = j % 4, ra = tap / 3, rb = tap % 3;
      // younger loads at this wait: the ring runs RW - 1 k-steps ahead until k-step 36 - RW issued the last refill
      constexpr int YOUNGER = MI * (RW - 1 - (j > 36 - RW ? j - (36 - RW) : 0));
      wait_w<YOUNGER>(aw[j % RW][0], aw[j % RW][1]);
      const int g = 2 * ks + kh;
      bf16x8 b[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int sl = (wn * NI + ni) * 32 + li + rb;
        const char* bp = Bs + ra * B_ST + (g * SLP + sl) * 16;
        if ((wn * NI + ni) == 3 && sl >= 128) bp = extra + ((ra * 8 + g) * 2 + (sl - 128)) * 16;
        if constexpr (CM >= 128) {
          if (!((rmask[ni] >> tap) & 1u)) bp = zslot;        // the tap falls off the image: read zeros
          b[ni] = *reinterpret_cast<const bf16x8*>(bp);
        } else {     // CM = 64 (three waves per SIMD, 168 registers): the address select spills; zero the registers
          f32x4 bv = *reinterpret_cast<const f32x4*>(bp);
          if (!((rmask[ni] >> tap) & 1u)) bv = f32x4{0.f, 0.f, 0.f, 0.f};
          b[ni] = __builtin_bit_cast(bf16x8, bv);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, aw[j % RW][mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[ni], acc[mi][ni], 0, 0, 0);
      }
      if constexpr (j + RW < 36) load_k(std::integral_constant<int, j + RW>{});
    };
    auto four = [&](auto t_tag) {
      constexpr int t = decltype(t_tag)::value;
      step(std::integral_constant<int, 4 * t>{});     step(std::integral_constant<int, 4 * t + 1>{});
      step(std::integral_constant<int, 4 * t + 2>{}); step(std::integral_constant<int, 4 * t + 3>{});
    };
    auto prologue = [&](auto t_tag) {
      constexpr int t = decltype(t_tag)::value;
      if constexpr (4 * t < RW) {
        load_k(std::integral_constant<int, 4 * t>{});     load_k(std::integral_constant<int, 4 * t + 1>{});
        load_k(std::integral_constant<int, 4 * t + 2>{}); load_k(std::integral_constant<int, 4 * t + 3>{});
      }
    };
    auto do_half = [&](auto half_tag) {
      constexpr int half = decltype(half_tag)::value;
      if constexpr (half > 0) __syncthreads();               // everybody has read the previous half's ranges
#pragma unroll
      for (int ra = 0; ra < 3; ++ra) {
        const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + slot;
        const unsigned voff = (q >= 0 && q < npix) ? (unsigned)((q - rbase) * CM * 2 + 16 * bg) : OOB;
        char* dst = Bs + ra * B_ST + (bg * SLP + 64 * (wave & 1)) * 16;
#pragma unroll
        for (int p = 0; p < 4; ++p) bglds16(voff + 32 * p, half * KC * 2, dst + 2 * p * SLP * 16);
      }
      if (wave == 0 && lane < 48) {                          // slots 128, 129 of the three ranges: [range][group][2]
        const int ra = lane >> 4, g = (lane >> 1) & 7, e = lane & 1;
        const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + 128 + e;
        const unsigned voff = (q >= 0 && q < npix) ? (unsigned)((q - rbase) * CM * 2 + 16 * g) : OOB;
        bglds16(voff, half * KC * 2, extra);                 // the DMA adds lane * 16
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) wl[mi] = wbase[mi] + half * (9 * 4096);
      prologue(std::integral_constant<int, 0>{}); prologue(std::integral_constant<int, 1>{}); prologue(std::integral_constant<int, 2>{});
      // the DMA pieces are OLDER than the MI RW weight loads: in-order return makes this the wait for the staged ranges
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * RW) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      four(std::integral_constant<int, 0>{}); four(std::integral_constant<int, 1>{}); four(std::integral_constant<int, 2>{});
      four(std::integral_constant<int, 3>{}); four(std::integral_constant<int, 4>{}); four(std::integral_constant<int, 5>{});
      four(std::integral_constant<int, 6>{}); four(std::integral_constant<int, 7>{}); four(std::integral_constant<int, 8>{});
    };
    do_half(std::integral_constant<int, 0>{});
    if constexpr (CCH > 1) do_half(std::integral_constant<int, 1>{});
    if constexpr (CCH > 2) { do_half(std::integral_constant<int, 2>{}); do_half(std::integral_constant<int, 3>{}); }
    static_assert(CCH <= 4, "once-staged form: up to four 64-channel parts");
    __syncthreads();                                         // nobody reads the stages any more
  }

  // ---------------------------------------------------------------- h2 = relu(acc + b2) -> bf16 -> LDS (B-operand image)
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
        *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SLP + (wn * NI + ni) * 32 + li) * 16 + 8 * kh) = v;
      }
    }
  __syncthreads();

  // ---------------------------------------------------------------- phase 3: 1x1 expand, K = CM, software-pipelined
  // Eight sub-passes per wave, each a [32 MS rows x 32 NS pixels] accumulator set (half of what the wave owns in a pass
  // of CM rows); two sets alternate: while the MFMAs of sub-pass i + 1 run, the epilogue of sub-pass i -- bias,
  // residual, ReLU, rounding, 16-byte stores -- is fed into the gaps half a group per two k-steps.  The W3 fragments
  // are one continuous stream through a register ring that runs across sub-pass boundaries, the residual rows a
  // second ring of NRES groups requested NRES groups before their use.  At CM = 64 a sub-pass is only four MFMAs, so
  // both rings are FOUR sub-passes deep there (16 k-steps of W3, 4 residual groups: 96 KB of residual rows in flight
  // per CU): with one sub-pass of lookahead every sub-pass waited a full memory latency and the res2 tails ran at
  // 2.2 TB/s.  (Loads return in order: a residual row requested before a W3 fragment has to land before that fragment
  // is used, so the W3 ring must be as deep as the residual lookahead.)
  constexpr int KSTEPS = CM / 16;
  constexpr int MS = CM >= 128 ? 2 : 1;            // row blocks per sub-pass
  constexpr int NS = CM == 256 ? 2 : 1;            // pixel blocks per sub-pass
  constexpr int NSUB = 8;
  constexpr int G = MS * 2 * NS;                   // epilogue halves per sub-pass (8 channels x 32 pixels per lane-row)
  constexpr int NGRP = MS * NS;                    // epilogue groups per sub-pass (16 channels per lane)
  constexpr int RING = CM == 64 ? 16 : 4;          // W3 ring, k-steps
  constexpr int NRES = CM == 64 ? 4 : 2;           // residual ring, groups
  constexpr bool UNROLLED = CM == 64;              // sub-pass index known at compile time (ring slots depend on it)
  static_assert(KSTEPS == 2 * G, "phase-3 schedule: one epilogue half per two k-steps");
  static_assert(UNROLLED || (KSTEPS % RING == 0 && NGRP % NRES == 0), "ring slots must not depend on a run-time sub-pass");
  const char* Hb = Bs + (kh * SLP + li) * 16;
  // sub-pass sp = 2 pass + part: rows (wm 4 + pass) MI + mi0 .., pixel blocks wn NI + ni0 ..
  // (NEXT: pass-major -- pass p of the four waves together covers row blocks 8 p .. 8 p + 7 = channels [256 p, 256 p + 256))
  auto sub_rb = [&](int sp) {
    return (NEXT ? (sp >> 1) * 4 + wm : wm * 4 + (sp >> 1)) * MI + (NS < NI ? 0 : (sp & 1) * MS);
  };
  auto sub_nb = [&](int sp) { return wn * NI + (NS < NI ? (sp & 1) * NS : 0); };
  // a sub-pass index is either an int (run-time loop, CM >= 128) or an integral_constant (CM = 64, fully unrolled)
  auto cval = [](auto sp) constexpr {              // its compile-time value; 0 when the ring slots do not depend on it
    if constexpr (std::is_integral_v<decltype(sp)>) return 0; else return decltype(sp)::value;
  };
  auto plus = [](auto sp, auto d) {                // sp + d, keeping the kind
    if constexpr (std::is_integral_v<decltype(sp)>) return sp + (int)decltype(d)::value;
    else return std::integral_constant<int, decltype(sp)::value + decltype(d)::value>{};
  };
  // W3 rows are permuted AT LOAD TIME: the lane that feeds MFMA row r = 8 q + 4 h + j of a 32-row block fetches the
  // fragment slot of channel 16 h + 4 q + j (same 1-KiB fragment line, lanes permuted).  An accumulator lane (pixel
  // li, half kh) holds rows 8 q + 4 kh + j in registers 4 q + j -- with the permutation these are the 16 CONSECUTIVE
  // channels 16 kh + 0..15 of the block: 32 contiguous bytes per lane, 64 per pixel and lane pair, with no cross-lane
  // movement at all (round 3 first used v_permlane32_swap pairs for 8 consecutive channels per lane)
  const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;
  // Every vector-memory access of this phase is a BUFFER instruction (round 4): an SGPR descriptor, the lane's fixed 32-bit
  // offset in one VGPR and the wave-uniform part (row block, k-step, pixel block) as the scalar offset.  The pointer form
  // made hipcc build a 64-bit address per lane for each of the 48 accesses of a sub-pass (119 v_lshl_add_u64 and
  // `global_load_dwordx4 v[..], v[a:a+1], off` in the ISA) in a phase whose instruction stream, not its bytes, is what
  // bounds it (profiles/r4/bottleneck_pipeline_study.md §8).  Pixels beyond the end get an out-of-range offset: their loads
  // return zeros, their stores are dropped -- no branch around the stores.
  const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wf3), 0, C4 * CM * 2, 0x00020000);
  // (NEXT sits at 510 of 512 registers and spills with the descriptors: it keeps the pointer form)
  const char* const w3base = reinterpret_cast<const char*>(Wf3) + woff3;
  auto w3_load = [&](int sp, int k, int ms) {       // fragment of k-step k, row block ms of sub-pass sp
    const int so = (sub_rb(sp) + ms) * (CCH * 4096) + k * 1024;
    if constexpr (NEXT) return *reinterpret_cast<const f32x4*>(w3base + so);
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w3, (int)woff3, so, 0));
  };
  f32x4 ar[RING][MS];
#pragma unroll
  for (int j = 0; j < RING; ++j)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) ar[j][ms] = w3_load(j / KSTEPS, j % KSTEPS, ms);
  f32x16 accA[MS][NS], accB[MS][NS];

  // one k-step of sub-pass sp into `c`; refills the ring slot with the fragment RING k-steps ahead in the stream
  auto kstep = [&](f32x16 (&c)[MS][NS], auto sp, auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    constexpr int slot = (cval(decltype(sp){}) * KSTEPS + k) % RING;
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
    constexpr int dsp = (k + RING) / KSTEPS, kn = (k + RING) % KSTEPS;   // the fragment that takes this slot
    if ((int)sp + dsp < NSUB) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) ar[slot][ms] = w3_load((int)sp + dsp, kn, ms);
    }
  };
  // epilogue: group grp = (nj, ms), ms fastest = one 32-row x 32-pixel accumulator block = 16 consecutive channels
  // per lane; it is finished in two halves (registers 8 h .. 8 h + 7 = channels 16 kh + 8 h ..), one per two
  // k-steps, and both 16-byte stores are issued TOGETHER after the second half: the lane pair of a pixel writes a
  // whole 64-byte sector in back-to-back instructions, and the other half of the 128-byte line (ms + 1) follows one
  // group later.  L2 hands part-written lines to the fabric quickly, each time as whole sectors: WRITE_SIZE of a
  // res4 launch of 8 frames (59.0 MB of output) was 101 MB with 32-byte pieces two epilogue steps apart, 76 MB one
  // step apart, 75 MB with whole sectors but the halves of a line two groups apart, 59.9 MB like this
  // (tools/probe_store_order.sh).  The residual is read the same way, 32 contiguous bytes per lane.
  bf16x8 rres[NRES][2];                              // residual rows in flight: [group in the stream % NRES][half]
  bf16x8 ohold;                                      // first half of the group being finished
  // addresses of the residual / output rows = a wave-uniform base (tile, pixel block, channel block) + a 32-bit lane
  // offset (pixel li of the block, channels 16 kh ..).  Pixels beyond the end read pixel 0 of their block (block 0 of
  // the tensor when the whole block is beyond the end) and are not stored.
  unsigned okmask = 0;                               // bit b: pixel li of pixel block b of this tile exists
#pragma unroll
  for (int b = 0; b < 4; ++b) okmask |= (n0 + b * 32 + li < npix ? 1u : 0u) << b;
  const unsigned voff_in = (unsigned)(li * C4 + 16 * kh) * 2;
  constexpr unsigned OOB3 = 0x80000000u;
  // descriptors based at the tile's first pixel; uniform byte offset of a group's rows inside the tile
  const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(residual) + n0 * C4, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(out + n0 * C4, 0, 0x7fffffff, 0x00020000);
  auto row_soff = [&](int sp, int grp) {
    const int ms = grp % MS, nj = grp / MS;
    return ((sub_nb(sp) + nj) * 32 * C4 + 32 * (sub_rb(sp) + ms)) * 2;
  };
  auto row_base = [&](int sp, int grp) {             // (pointer form, NEXT) uniform element offset of the group's rows
    const int ms = grp % MS, nj = grp / MS;
    const int64_t pb = n0 + (sub_nb(sp) + nj) * 32;
    return (pb < npix ? pb : 0) * C4 + 32 * (sub_rb(sp) + ms);
  };
  // group gg = e NGRP + grp of the stream of epilogue groups (e = the sub-pass whose results it finishes)
  auto res_issue = [&](auto e, auto grp_tag) {
    constexpr int grp = decltype(grp_tag)::value;
    constexpr int slot = (cval(decltype(e){}) * NGRP + grp) % NRES;
    const bool okp = (okmask >> (sub_nb(e) + grp / MS)) & 1u;
    if constexpr (NEXT) {
      const char* rp = reinterpret_cast<const char*>(residual + row_base(e, grp)) + (okp ? voff_in : (unsigned)(16 * kh) * 2);
      rres[slot][0] = *reinterpret_cast<const bf16x8*>(rp);
      rres[slot][1] = *reinterpret_cast<const bf16x8*>(rp + 16);
    } else {
      const int vo = (int)(okp ? voff_in : OOB3), so = row_soff(e, grp);
      rres[slot][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, vo, so, 0));
      rres[slot][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, vo + 16, so, 0));
    }
  };
  // the group NRES after (e, grp) in the stream, if there is one
  auto res_issue_ahead = [&](auto e, auto grp_tag) {
    constexpr int grp = decltype(grp_tag)::value;
    constexpr int de = (grp + NRES) / NGRP, gn = (grp + NRES) % NGRP;
    if ((int)e + de < NSUB) res_issue(plus(e, std::integral_constant<int, de>{}), std::integral_constant<int, gn>{});
  };
  auto group_finish = [&](f32x16 (&c)[MS][NS], auto e, auto g_tag) {
    constexpr int g = decltype(g_tag)::value;
    constexpr int grp = g >> 1, h = g & 1, ms = grp % MS, nj = grp / MS;
    constexpr int slot = (cval(decltype(e){}) * NGRP + grp) % NRES;
    const int chm = 32 * (sub_rb(e) + ms) + 16 * kh + 8 * h;
    // (NEXT is at 510 of 512 registers: the two LDS addresses spill it; it keeps the loads from global memory)
    constexpr bool BIAS_GLOBAL = NEXT;
    const float4 bv0 = BIAS_GLOBAL ? *reinterpret_cast<const float4*>(bias3 + chm)
                            : *reinterpret_cast<const float4*>(Bs + BIAS_OFF + 4 * chm);
    const float4 bv1 = BIAS_GLOBAL ? *reinterpret_cast<const float4*>(bias3 + chm + 4)
                            : *reinterpret_cast<const float4*>(Bs + BIAS_OFF + 4 * chm + 16);
    const float v[8] = {c[ms][nj][8 * h] + bv0.x,     c[ms][nj][8 * h + 1] + bv0.y, c[ms][nj][8 * h + 2] + bv0.z,
                        c[ms][nj][8 * h + 3] + bv0.w, c[ms][nj][8 * h + 4] + bv1.x, c[ms][nj][8 * h + 5] + bv1.y,
                        c[ms][nj][8 * h + 6] + bv1.z, c[ms][nj][8 * h + 7] + bv1.w};
    const bf16x8 rv = rres[slot][h];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf(v[j] + (float)rv[j], 0.f);
    if constexpr (NEXT) {
      // the sub-pass's chunk image (buffer e & 1): channel group (channel - 256 pass) / 8, slot = pixel of its 64
      const int g8 = ((32 * (sub_rb(e) + ms)) & 255) / 8 + 2 * kh + h;
      *reinterpret_cast<bf16x8*>(Bs + CHUNK_OFF + ((int)e & 1) * CHUNK_BYTES + (g8 * SLPC + nj * 32 + li) * 16) = o;
    }
    if constexpr (h == 0) {
      ohold = o;
    } else {
      bool okp = (okmask >> (sub_nb(e) + nj)) & 1u;
      if constexpr (NEXT) {
        char* op = reinterpret_cast<char*>(out + row_base(e, grp)) + voff_in;
        if (okp) {
          *reinterpret_cast<bf16x8*>(op) = ohold;
          *reinterpret_cast<bf16x8*>(op + 16) = o;
        }
      } else {
        const int vo = (int)(okp ? voff_in : OOB3), so = row_soff(e, grp);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ohold), rsrc_out, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc_out, vo + 16, so, 0);
      }
      res_issue_ahead(e, std::integral_constant<int, grp>{});   // its ring slot is free now
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
  using M1 = std::integral_constant<int, -1>;
  // sub-pass sp into `cur` while the epilogue of sub-pass sp - 1 (in `prev`) drains: half g is finished after k-step
  // 2 g + 1
  auto no_hook = [](auto) {};
  // `hook(k)`: extra work issued after k-step k (NEXT: a k-step of the following block's conv1 on the previous chunk)
  auto subpass_h = [&](f32x16 (&cur)[MS][NS], f32x16 (&prev)[MS][NS], auto sp, auto drain_tag, auto&& hook) {
    constexpr bool DRAIN = decltype(drain_tag)::value;
    zero(cur);
    auto two = [&](auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      kstep(cur, sp, std::integral_constant<int, 2 * g>{});
      hook(std::integral_constant<int, 2 * g>{});
      __builtin_amdgcn_sched_barrier(0);
      kstep(cur, sp, std::integral_constant<int, 2 * g + 1>{});
      hook(std::integral_constant<int, 2 * g + 1>{});
      if constexpr (DRAIN) group_finish(prev, plus(sp, M1{}), g_tag);
      __builtin_amdgcn_sched_barrier(0);
    };
    two(std::integral_constant<int, 0>{});
    two(std::integral_constant<int, 1>{});
    if constexpr (G > 2) { two(std::integral_constant<int, 2>{}); two(std::integral_constant<int, 3>{}); }
    if constexpr (G > 4) {
      two(std::integral_constant<int, 4>{}); two(std::integral_constant<int, 5>{});
      two(std::integral_constant<int, 6>{}); two(std::integral_constant<int, 7>{});
    }
  };
  auto subpass = [&](f32x16 (&cur)[MS][NS], f32x16 (&prev)[MS][NS], auto sp, auto drain_tag) {
    constexpr bool DRAIN = decltype(drain_tag)::value;
    zero(cur);
    auto two = [&](auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      kstep(cur, sp, std::integral_constant<int, 2 * g>{});
      __builtin_amdgcn_sched_barrier(0);
      kstep(cur, sp, std::integral_constant<int, 2 * g + 1>{});
      if constexpr (DRAIN) group_finish(prev, plus(sp, M1{}), g_tag);
      __builtin_amdgcn_sched_barrier(0);
    };
    two(std::integral_constant<int, 0>{});
    two(std::integral_constant<int, 1>{});
    if constexpr (G > 2) { two(std::integral_constant<int, 2>{}); two(std::integral_constant<int, 3>{}); }
    if constexpr (G > 4) {
      two(std::integral_constant<int, 4>{}); two(std::integral_constant<int, 5>{});
      two(std::integral_constant<int, 6>{}); two(std::integral_constant<int, 7>{});
    }
  };
  // the first NRES groups of the stream
  {
    auto first = [&](auto gg_tag) {
      constexpr int gg = decltype(gg_tag)::value;
      if constexpr (gg < NRES) {
        if constexpr (UNROLLED) res_issue(std::integral_constant<int, gg / NGRP>{}, std::integral_constant<int, gg % NGRP>{});
        else res_issue((int)(gg / NGRP), std::integral_constant<int, gg % NGRP>{});
      }
    };
    first(std::integral_constant<int, 0>{}); first(std::integral_constant<int, 1>{});
    first(std::integral_constant<int, 2>{}); first(std::integral_constant<int, 3>{});
  }
  if constexpr (NEXT) {
    // ---- tail + conv1 of the next block.  A sub-pass finishes 256 channels (one pass) of 64 pixels; its rounded
    // results are the chunk c = sub-pass index of conv1n's operand: K = 256 channels, N = 64 pixels.  Three things
    // run in the MFMA stream of sub-pass s: its own k-steps, the epilogue of sub-pass s - 1 (-> chunk buffer
    // (s - 1) & 1) and conv1n on chunk s - 2 (buffer s & 1), one k-step of each per k-step; a barrier between
    // sub-passes publishes a chunk and frees the other buffer.  Output pixel half h sees chunks h, 2 + h, 4 + h, 6 + h
    // = channels 0..1023 in natural order on one accumulator chain: the stand-alone conv1's contraction order.
    (void)no_hook;
    f32x16 acc1[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[mi][ni][e] = 0.f;
    // W1n fragments, rows permuted like W3's (16 consecutive channels per accumulator lane): row block 2 wm + mi,
    // 16 chunks of 4 KiB per row block, k-step s at s KiB.  The stream of (chunk, k-step) pairs runs through a ring
    // of four; chunk c uses k-steps 16 (c / 2) + 0..15 (each k-step is fetched for both pixel halves: L2 hits)
    const char* w1p[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      w1p[mi] = reinterpret_cast<const char*>(Wf1n) + woff3 + (int64_t)(MI * wm + mi) * (4 * CCH * 4096);
    f32x4 a1[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a1[j][mi] = *reinterpret_cast<const f32x4*>(w1p[mi] + j * 1024);
    const char* Cb = Bs + CHUNK_OFF + (kh * SLPC + li) * 16;
    // k-step ks of chunk c (parity PAR known at compile time; kbase = 16 (c / 2); `more`: a chunk follows)
    auto c1step = [&](auto par_tag, int kbase, bool more, auto ks_tag) {
      constexpr int PAR = decltype(par_tag)::value;
      constexpr int ks = decltype(ks_tag)::value;
      constexpr int slot = ks % 4;
      bf16x8 b[2];
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
        b[nj] = *reinterpret_cast<const bf16x8*>(Cb + PAR * CHUNK_BYTES + ((2 * ks) * SLPC + nj * 32) * 16);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a1[slot][mi]);
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
          acc1[mi][2 * PAR + nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[nj], acc1[mi][2 * PAR + nj], 0, 0, 0);
      }
      if constexpr (ks + 4 < 16) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a1[slot][mi] = *reinterpret_cast<const f32x4*>(w1p[mi] + (int64_t)(kbase + ks + 4) * 1024);
      } else if (more) {                                   // the first k-steps of the next chunk
        const int nb = PAR ? kbase + 16 : kbase;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a1[slot][mi] = *reinterpret_cast<const f32x4*>(w1p[mi] + (int64_t)(nb + ks + 4 - 16) * 1024);
      }
    };
    auto publish = [&]() {                                 // chunk written / chunk read by every wave
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    static_assert(G == 8 && KSTEPS == 16 && MS == 2 && NS == 2, "NEXT: CM = 256 schedule");
    subpass(accA, accB, 0, std::false_type{});
    subpass(accB, accA, 1, std::true_type{});              // epilogue of sub-pass 0 -> buffer 0
    publish();
#pragma unroll 1
    for (int pass = 1; pass < 4; ++pass) {
      const int kb = 16 * (pass - 1);
      // sub-pass 2 pass: epilogue of 2 pass - 1 -> buffer 1, conv1n on chunk 2 pass - 2 (buffer 0)
      subpass_h(accA, accB, 2 * pass, std::true_type{}, [&](auto k) { c1step(P0{}, kb, true, k); });
      publish();
      // sub-pass 2 pass + 1: epilogue of 2 pass -> buffer 0, conv1n on chunk 2 pass - 1 (buffer 1)
      subpass_h(accB, accA, 2 * pass + 1, std::true_type{}, [&](auto k) { c1step(P1{}, kb, true, k); });
      publish();
    }
    // drain: epilogue of sub-pass 7 -> buffer 1, interleaved with conv1n on chunk 6 (buffer 0)
    {
      auto fin2 = [&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        c1step(P0{}, 48, true, std::integral_constant<int, 2 * g>{});
        __builtin_amdgcn_sched_barrier(0);
        c1step(P0{}, 48, true, std::integral_constant<int, 2 * g + 1>{});
        group_finish(accB, (int)(NSUB - 1), g_tag);
        __builtin_amdgcn_sched_barrier(0);
      };
      fin2(std::integral_constant<int, 0>{}); fin2(std::integral_constant<int, 1>{});
      fin2(std::integral_constant<int, 2>{}); fin2(std::integral_constant<int, 3>{});
      fin2(std::integral_constant<int, 4>{}); fin2(std::integral_constant<int, 5>{});
      fin2(std::integral_constant<int, 6>{}); fin2(std::integral_constant<int, 7>{});
    }
    publish();
    {
      auto last = [&](auto ks_tag) { c1step(P1{}, 48, false, ks_tag); __builtin_amdgcn_sched_barrier(0); };
      last(std::integral_constant<int, 0>{});  last(std::integral_constant<int, 1>{});
      last(std::integral_constant<int, 2>{});  last(std::integral_constant<int, 3>{});
      last(std::integral_constant<int, 4>{});  last(std::integral_constant<int, 5>{});
      last(std::integral_constant<int, 6>{});  last(std::integral_constant<int, 7>{});
      last(std::integral_constant<int, 8>{});  last(std::integral_constant<int, 9>{});
      last(std::integral_constant<int, 10>{}); last(std::integral_constant<int, 11>{});
      last(std::integral_constant<int, 12>{}); last(std::integral_constant<int, 13>{});
      last(std::integral_constant<int, 14>{}); last(std::integral_constant<int, 15>{});
    }
    // h1n = relu(acc1 + b1n), rounded once; a lane holds channels 32 (2 wm + mi) + 16 kh + 0..15 of pixel ni 32 + li
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int chb = 32 * (MI * wm + mi) + 16 * kh;
      float bv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t4 = *reinterpret_cast<const float4*>(bias1n + chb + 4 * q);
        bv[4 * q] = t4.x; bv[4 * q + 1] = t4.y; bv[4 * q + 2] = t4.z; bv[4 * q + 3] = t4.w;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bf16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = (__bf16)fmaxf(acc1[mi][ni][j] + bv[j], 0.f);
          o1[j] = (__bf16)fmaxf(acc1[mi][ni][8 + j] + bv[8 + j], 0.f);
        }
        if ((okmask >> ni) & 1u) {
          char* hp = reinterpret_cast<char*>(h1n + (n0 + ni * 32 + li) * CM + chb);
          *reinterpret_cast<bf16x8*>(hp) = o0;
          *reinterpret_cast<bf16x8*>(hp + 16) = o1;
        }
      }
    }
    return;
  }
  if constexpr (UNROLLED) {
    subpass(accA, accB, std::integral_constant<int, 0>{}, std::false_type{});
    subpass(accB, accA, std::integral_constant<int, 1>{}, std::true_type{});
    subpass(accA, accB, std::integral_constant<int, 2>{}, std::true_type{});
    subpass(accB, accA, std::integral_constant<int, 3>{}, std::true_type{});
    subpass(accA, accB, std::integral_constant<int, 4>{}, std::true_type{});
    subpass(accB, accA, std::integral_constant<int, 5>{}, std::true_type{});
    subpass(accA, accB, std::integral_constant<int, 6>{}, std::true_type{});
    subpass(accB, accA, std::integral_constant<int, 7>{}, std::true_type{});
  } else {
    subpass(accA, accB, 0, std::false_type{});
#pragma unroll 1
    for (int sp = 1; sp < NSUB; sp += 2) {
      subpass(accB, accA, sp, std::true_type{});
      if (sp + 1 < NSUB) subpass(accA, accB, sp + 1, std::true_type{});
    }
  }
  // the last sub-pass (odd, in accB) drains on its own
  {
    auto fin = [&](auto g_tag) {
      if constexpr (UNROLLED) group_finish(accB, std::integral_constant<int, NSUB - 1>{}, g_tag);
      else group_finish(accB, (int)(NSUB - 1), g_tag);
    };
    fin(std::integral_constant<int, 0>{});
    fin(std::integral_constant<int, 1>{});
    if constexpr (G > 2) { fin(std::integral_constant<int, 2>{}); fin(std::integral_constant<int, 3>{}); }
    if constexpr (G > 4) {
      fin(std::integral_constant<int, 4>{}); fin(std::integral_constant<int, 5>{});
      fin(std::integral_constant<int, 6>{}); fin(std::integral_constant<int, 7>{});
    }
  }
}

template <int CM, bool NEXT>
int launch(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, const uint16_t* frag2, const float* bias2,
           const uint16_t* frag3, const float* bias3, const uint16_t* residual, uint16_t* out, void* stream,
           const uint16_t* frag1n = nullptr, const float* bias1n = nullptr, uint16_t* h1n = nullptr) {
  const int64_t npix = NB * H * W;
  const int64_t tiles = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_bf16: grid too large");
  constexpr int nst = CM == 256 ? 4 : 3;
  constexpr int cch = CM / KC;
  // + slots 128, 129 (once-staged form / range ring); NEXT: + the chunk image of a pass
  // stages / h2 image + slots 128, 129 (side regions) + the zero slot (+ NEXT: two chunk images)
  constexpr size_t smem = (size_t)(cch > nst ? cch : nst) * B_ST + 1024 + 256 + (size_t)4 * CM * 4 +
                          (NEXT ? (size_t)2 * (CM / 8) * 68 * 16 : 0);      // ... + b3
  static_assert(smem <= 160 * 1024, "LDS budget");
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(bottleneck_bf16_kernel<CM, NEXT>), smem, "tspn_bottleneck_tail_bf16"))
    return rc;
  hipLaunchKernelGGL((bottleneck_bf16_kernel<CM, NEXT>), dim3((unsigned)tiles), dim3(THREADS), smem, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(h1), reinterpret_cast<const __bf16*>(frag2), bias2,
                     reinterpret_cast<const __bf16*>(frag3), bias3, reinterpret_cast<const __bf16*>(residual),
                     reinterpret_cast<__bf16*>(out), (int)H, (int)W, npix, reinterpret_cast<const __bf16*>(frag1n), bias1n,
                     reinterpret_cast<__bf16*>(h1n));
  return tspn::check_launch("tspn_bottleneck_tail_bf16");
}

}  // namespace

extern "C" int tspn_bottleneck_tail_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                         const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                         const float* bias3, const uint16_t* residual, uint16_t* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "tspn_bottleneck_tail_bf16: bad sizes");
  TSPN_REQUIRE(CM == 64 || CM == 128 || CM == 256, TSPN_EUNSUPPORTED,
               "tspn_bottleneck_tail_bf16: bottleneck channels must be 64, 128 or 256 (got %lld)", (long long)CM);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(h1 && frag2 && bias2 && frag3 && bias3 && residual && out, TSPN_EINVAL,
               "tspn_bottleneck_tail_bf16: null pointer");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(h1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) && al16(residual) && al16(out),
               TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_bf16: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_bf16: dimension too large");
  if (CM == 256) return launch<256, false>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
  if (CM == 128) return launch<128, false>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
  return launch<64, false>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
}

extern "C" int tspn_bottleneck_tail_next_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                              const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                              const float* bias3, const uint16_t* residual, uint16_t* out,
                                              const uint16_t* frag1n, const float* bias1n, uint16_t* h1n, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "tspn_bottleneck_tail_next_bf16: bad sizes");
  TSPN_REQUIRE(CM == 256, TSPN_EUNSUPPORTED,
               "tspn_bottleneck_tail_next_bf16: built for 256 bottleneck channels (got %lld)", (long long)CM);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(h1 && frag2 && bias2 && frag3 && bias3 && residual && out && frag1n && bias1n && h1n, TSPN_EINVAL,
               "tspn_bottleneck_tail_next_bf16: null pointer");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(h1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) && al16(residual) && al16(out) &&
                   al16(frag1n) && al16(bias1n) && al16(h1n),
               TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_next_bf16: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_next_bf16: dimension too large");
  return launch<256, true>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream, frag1n, bias1n, h1n);
}
