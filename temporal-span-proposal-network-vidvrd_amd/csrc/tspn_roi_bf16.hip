// bf16-operand conv2d for the RoI feature head (the VidOR end-to-end config is bf16).
//
// Same operator and structure as conv2d_nhwc_frag_kernel (tspn_roi.hip): implicit GEMM, M = output
// channels, N = output pixels, K = taps x Cin; tile 128 x 128, wave w = rows [32 w, 32 w + 32) x all 128
// pixels; FRAGMENT-MAJOR weights loaded straight into MFMA operand registers; x through LDS by DMA with a
// zero page for padding taps; bare s_barrier + counted vmcnt.  On v_mfma_f32_32x32x16_bf16:
//   * operands are bf16 values (x, weights with the batch norm folded in fp32 and rounded once, residual),
//     every product is exact, accumulation is fp32, bias is fp32, the result act(acc + bias + residual) is
//     rounded to bf16 once (round to nearest even) -- the semantics restated in
//     oracle/roi_head_oracle.py:conv2d_bf16;
//   * K chunk = 64 channels of one tap = four MFMA k-steps of 16 channels (16 MFMAs per wave and chunk);
//     a pixel's 64 channels are one 128-byte line of x, staged as eight 16-byte pieces.  Tap chunks (round 5): LINE-MAJOR
//     pieces -- lane l of a DMA instruction fetches piece (l & 7) ^ f of pixel 8 q + (l >> 3), so an instruction reads
//     eight whole lines (a quad of lanes = one 64-byte half line) instead of 16 bytes of 64 different lines: L2-resident
//     x enters the CU at 44 instead of 14.5 bytes per clock (tools/probes/tcp_line_coalesce_probe.hip; res4's conv1
//     46 -> 39 us per 18 frames).  The LDS image is pixel-major [128 pixels][8 positions][8 bf16], position = piece ^
//     f(pixel), f(P) = (P & 7) ^ ((P >> 3) & 1): 16 consecutive pixels reading the same piece hit 64 different banks.
//     Linear ranges (RNG) keep [8 channel groups][132 slots][8 bf16];
//   * weights  Wf[Cout/32][Cin/64][tap][ks = 0..3][lane = 32 kh + li][8] = w[32 mb + li][64 c + 16 ks + 8 kh + j][tap]:
//     one global_load_dwordx4 per lane and k-step, refilled for the next chunk right after the k-step's MFMAs.
//     The contraction runs in THIS order -- 64-channel chunk by chunk, all taps of a chunk in a row (chunk i = tap
//     i % taps of channels 64 (i / taps) ..) -- so that the fused bottleneck tail, which stages all nine taps of a
//     64-channel half at once (tspn_bottleneck_bf16.hip), adds the same products in the same order.
//   * MI = 32-row blocks per wave (template): MI = 2 (tile 256 x 128, Cout % 64 == 0) shares every x fragment
//     between two weight fragments -- 0.5 instead of 1 LDS fragment read per MFMA, which is what bounds
//     the MI = 1 kernel (8 waves x 16 KB of ds_read_b128 per chunk against 1024 MFMA cycles per SIMD).
// Needs Cin % 64 == 0, Cout % 32 == 0.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int BN = 128;
constexpr int KC = 64;                  // channels per chunk
constexpr int SLP = 132;                // padded pixel slots per channel group
constexpr int B_ST = 8 * SLP * 16;      // bytes per x stage: [8 groups][132 slots][8 bf16]

template <int OFF>
__device__ __forceinline__ void load_wfrag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait_w(f32x4& r) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(VM));
}
template <int VM>
__device__ __forceinline__ void wait_w(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(VM));
}

// w fp32 [Cout][Cin][KH][KW] -> bf16 fragment-major [Cout/32][Cin/64][taps][4][64][8]
__global__ void pack_conv2d_frag_bf16_kernel(const float* __restrict__ w, int64_t Cout, int64_t Cin, int64_t ntaps,
                                             __bf16* __restrict__ packed) {
  const int64_t total = ntaps * Cin * Cout;
  const int64_t cch = Cin / KC;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(o & 7), lane = (int)((o >> 3) & 63), ks = (int)((o >> 9) & 3);
    const int64_t q = o >> 11;
    const int64_t tap = q % ntaps, c = (q / ntaps) % cch, mb = q / (cch * ntaps);
    const int64_t co = 32 * mb + (lane & 31), ci = KC * c + 16 * ks + 8 * (lane >> 5) + j;
    packed[o] = (__bf16)w[(co * Cin + ci) * ntaps + tap];
  }
}

// waves per SIMD the 32-row form is compiled for (112 VGPRs): the memory-bound short-K 1x1 layers run on it
#ifndef TSPN_ROI_BF16_MI1_WAVES
#define TSPN_ROI_BF16_MI1_WAVES 3
#endif
// RNG (3x3 / stride 1 / pad 1 only): the x stages are LINEAR RANGES instead of tap chunks -- tap (a, b) of output pixel
// n is input pixel n + (a - 1) W + (b - 1), so one staged range (pixels n0 + (a - 1) W - 1 .. + 129 of a 64-channel
// part) serves the three taps (a, 0..2) at slot offsets 0..2: a third of the DMA volume and of the barriers; slots
// 128, 129 in a 256-byte side region per stage; taps off the image are zeroed at the read by the lane's tap mask.
// Same contraction order (tspn_bottleneck_bf16.hip does the same: the two stay bit-identical).
template <int MI, bool RNG>
__global__ __launch_bounds__(THREADS, (MI == 1 ? TSPN_ROI_BF16_MI1_WAVES : 2)) void conv2d_nhwc_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wf, const float* __restrict__ bias,
    const __bf16* __restrict__ residual, __bf16* __restrict__ out, int H, int W, int Cin, int Cout, int KH,
    int KW, int stride, int pad, int OH, int OW, int64_t npix, int tiles_m, int tiles_n, int relu) {
  constexpr int BM = 128 * MI;
  constexpr int EPI_BYTES = 4 * 32 * (32 * MI + 4) * 4;   // epilogue transpose: 4 waves x 32 pixels x padded row
  constexpr int ZERO_OFF = 2 * B_ST + 512;                // RNG: 16 bytes of zeros, the target of taps that fall off the image
  __shared__ __attribute__((aligned(16))) char Bs[ZERO_OFF + 16 > EPI_BYTES ? ZERO_OFF + 16 : EPI_BYTES];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GM = 4;
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
  const int li = lane & 31, kh = lane >> 5;

  const int cchunks = Cin / KC;
  const int nchunks = KH * KW * cchunks;
  const char* wbase[MI];                           // per 32-row block: 4 KiB per chunk, [4 ks][64 lanes][16 B]
  const unsigned woff = lane * 16;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + MI * wave + mi;
    mb = mb < (Cout >> 5) ? mb : 0;
    wbase[mi] = reinterpret_cast<const char*>(Wf) + (int64_t)mb * nchunks * 4096;
  }
  // x pieces: the four pieces of a lane belong to ONE output pixel (slot), channel groups bg, bg + 2, bg + 4, bg + 6
  // The pieces are BUFFER loads (buffer_load_dwordx4 ... offen lds; round 4): an SGPR descriptor based at the image the
  // tile starts in, a 32-bit lane offset, and a lane whose tap falls off the image simply gets an offset beyond
  // the descriptor's range -- the hardware then writes ZEROS into LDS (tools/probes/buffer_lds_oob_probe.hip), so there is
  // no zero page and no per-lane choice between two 64-bit pointers; beside MFMAs the buffer form is also the cheaper one
  // to issue (tools/probes/lds_dma_issue_probe.hip: 110 - 140 against 175 - 195 cycles per piece).
  const int slot = 64 * (wave & 1) + lane;         // (RNG: lane = pixel slot, channel groups bg, bg + 2, bg + 4, bg + 6)
  const int bg = wave >> 1;
  constexpr unsigned OOB = 0x80000000u;            // beyond num_records = 2^31 - 1: the piece arrives as zeros
  auto in_pixel = [&](int64_t n, int& ih0, int& iw0) {      // first tap's input pixel index of output pixel n (may be < 0)
    const int64_t nb = n / ((int64_t)OH * OW);
    const int r = (int)(n - nb * OH * OW);
    const int oh = r / OW, ow = r - oh * OW;
    ih0 = oh * stride - pad; iw0 = ow * stride - pad;
    return (nb * H + ih0) * (int64_t)W + iw0;
  };
  // base = the first pixel of the image the tile starts in: every valid tap of the tile lies at or behind it
  const int64_t base_pix = ((n0 < npix ? n0 : 0) / ((int64_t)OH * OW)) * H * W;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x) + base_pix * Cin, 0, 0x7fffffff, 0x00020000);
  // tap chunks: piece instruction q = 4 wave + p covers pixels 8 q .. 8 q + 7 of the tile; lane l: pixel 8 q + (l >> 3),
  // LDS position l & 7 = piece ((l & 7) ^ f(pixel)), f = (l >> 3) ^ (q & 1)
  unsigned pboff[4];                               // byte offset of the lane's piece (first tap) from the base, per instruction
  unsigned long long tapmask[4];
  if constexpr (!RNG) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t n = n0 + 8 * (4 * wave + p) + (lane >> 3);
      const bool okn = n < npix;
      const int64_t nc = okn ? n : 0;
      int ih0, iw0;
      const int64_t ip = in_pixel(nc, ih0, iw0);
      pboff[p] = (unsigned)((ip - base_pix) * Cin * 2 + 16 * ((lane & 7) ^ (lane >> 3) ^ (p & 1)));
      unsigned long long m = 0;
      for (int a = 0; a < KH; ++a)
        for (int b = 0; b < KW; ++b)
          if (okn && ih0 + a >= 0 && ih0 + a < H && iw0 + b >= 0 && iw0 + b < W) m |= 1ull << (a * KW + b);
      tapmask[p] = m;
    }
  }
  auto bglds16 = [&](unsigned voff, int soff, char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)l, 16, (int)voff, soff, 0, 0);
  };
  auto stage_x = [&](int buf, int i) {             // exactly four pieces per wave
    const int ntaps = KH * KW;
    const int c = i / ntaps, tap = i - c * ntaps;
    const int ta = tap / KW, tb = tap - ta * KW;
    const unsigned tapoff = (unsigned)((ta * W + tb) * Cin * 2);
    const int soff = c * KC * 2;
    char* dst = Bs + buf * B_ST + 4 * wave * 1024;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool valid = (tapmask[p] >> tap) & 1ull;
      bglds16(valid ? pboff[p] + tapoff : OOB, soff, dst + p * 1024);
    }
  };

  f32x16 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x4 a[4][MI];      // weight fragments of the four k-steps (8 bf16 each, carried as 4 dwords)
  // B fragment of k-step ks: pixel 32 ni + li, piece 2 ks + kh at position (2 ks + kh) ^ f(li); with the lane's base holding
  // f's bits the k-step is one XOR on the offset
  const unsigned fli = (unsigned)((li & 7) ^ ((li >> 3) & 1));
  const unsigned lane_b = (unsigned)(li * 128) + (((unsigned)kh ^ (fli & 1)) << 4) + ((fli >> 1) << 5);
  auto read_b = [&](const char* Bb, int ks, bf16x8 (&b)[4]) {
    const char* bp = Bb + (lane_b ^ (unsigned)(ks << 5));
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      b[ni] = *reinterpret_cast<const bf16x8*>(bp + ni * 4096);
  };
  auto mfma_step = [&](const f32x4 (&aw)[MI], const bf16x8 (&b)[4]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const bf16x8 av = __builtin_bit_cast(bf16x8, aw[mi]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  auto load_step = [&](auto ks_tag) {               // the MI fragments of k-step ks of the chunk at wbase
    constexpr int KS = decltype(ks_tag)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) load_wfrag<1024 * KS>(a[KS][mi], woff, wbase[mi]);
  };
  auto wait_step = [&](auto vm_tag, int ks) {
    constexpr int VM = decltype(vm_tag)::value;
    if constexpr (MI == 1) wait_w<VM>(a[ks][0]); else wait_w<VM>(a[ks][0], a[ks][1]);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  auto bump = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) wbase[mi] += 4096;
  };

  if constexpr (RNG) {
    char* const extra = Bs + 2 * B_ST;                       // [stage][8 groups][2 slots] x 16 B
    if (tid < 4) reinterpret_cast<float*>(Bs + ZERO_OFF)[tid] = 0.f;   // published by the barrier below
    const char* const zslot = Bs + ZERO_OFF;
    unsigned rmask[4];                                       // tap masks of the pixels this lane reads as its B columns
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int64_t n = n0 + ni * 32 + li;
      const bool okn = n < npix;
      const int64_t nc = okn ? n : 0;
      const int64_t nb = nc / ((int64_t)OH * OW);
      const int r = (int)(nc - nb * OH * OW);
      const int oh = r / OW, ow = r - oh * OW;
      unsigned m = 0;
      for (int ta = 0; ta < 3; ++ta)
        for (int tb = 0; tb < 3; ++tb)
          if (okn && oh - 1 + ta >= 0 && oh - 1 + ta < H && ow - 1 + tb >= 0 && ow - 1 + tb < W) m |= 1u << (ta * 3 + tb);
      rmask[ni] = m;
    }
    const int64_t npin = npix;                               // stride 1, pad 1: input pixels = output pixels
    // ranges: descriptor based at pixel max(n0 - W - 1, 0); a range pixel outside the tensor is an out-of-range offset
    const int64_t rbase = n0 - W - 1 > 0 ? n0 - W - 1 : 0;
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x) + rbase * Cin, 0, 0x7fffffff, 0x00020000);
    auto stage_r = [&](int buf, int i) {                     // range i = 3 c + ra
      const int c = i / 3, ra = i - 3 * c;
      const int soff = c * KC * 2;
      const int64_t q = n0 + (int64_t)(ra - 1) * W - 1 + slot;
      const unsigned voff = (q >= 0 && q < npin) ? (unsigned)((q - rbase) * Cin * 2 + 16 * bg) : OOB;
      char* dst = Bs + buf * B_ST + (bg * SLP + 64 * (wave & 1)) * 16;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, (__attribute__((address_space(3))) void*)(dst + 2 * p * SLP * 16), 16,
                                                 (int)(voff + 32 * p), soff, 0, 0);
      if (wave == 0 && lane < 16) {                          // slots 128, 129: [group][2]
        const int g = lane >> 1, e = lane & 1;
        const int64_t q2 = n0 + (int64_t)(ra - 1) * W - 1 + 128 + e;
        const unsigned voff2 = (q2 >= 0 && q2 < npin) ? (unsigned)((q2 - rbase) * Cin * 2 + 16 * g) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, (__attribute__((address_space(3))) void*)(extra + buf * 256), 16,
                                                 (int)voff2, soff, 0, 0);      // the DMA adds lane * 16
      }
    };
    auto read_r = [&](int buf, int tap, int rb, int ks, bf16x8 (&b)[4]) {
      const char* Bb = Bs + buf * B_ST + ((2 * ks + kh) * SLP + li + rb) * 16;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const char* bp = Bb + ni * 32 * 16;
        if (ni == 3) bp = (li + rb >= 32) ? extra + buf * 256 + ((2 * ks + kh) * 2 + (li + rb - 32)) * 16 : bp;
        // a tap off the image reads the zero slot: masking by ADDRESS leaves the loaded registers untouched, so the
        // compiler no longer waits for every read right behind its issue (round 4; four v_cndmask per fragment before)
        if (!((rmask[ni] >> tap) & 1u)) bp = zslot;
        b[ni] = *reinterpret_cast<const bf16x8*>(bp);
      }
    };
    const int nrng = 3 * cchunks;
    stage_r(0, 0);
    __syncthreads();
    load_step(K0{}); load_step(K1{}); load_step(K2{}); load_step(K3{});
    bump();
    __builtin_amdgcn_sched_barrier(0);
    // one round = the four k-steps of tap (ra, rb) of range i; VMEM issue order and counts as in chunk_body below, the
    // x pieces (4; wave 0 issues 5: its waits are one operation stricter than needed) only in the first round
    auto round_body = [&](int i, int ra, auto rb_tag, auto stage_tag, auto more_tag, auto last_tag) {
      constexpr int rb = decltype(rb_tag)::value;
      constexpr bool STAGE = decltype(stage_tag)::value, MORE = decltype(more_tag)::value, LAST = decltype(last_tag)::value;
      constexpr int NX = STAGE ? 4 : 0, R = MORE ? MI : 0, L = MI;
      const int buf = i & 1, tap = 3 * ra + rb;
      bf16x8 b0[4], b1[4];
      wait_step(std::integral_constant<int, 3 * L>{}, 0);
      if (STAGE) stage_r(buf ^ 1, i + 1);
      __builtin_amdgcn_sched_barrier(0);
      read_r(buf, tap, rb, 0, b0);
      read_r(buf, tap, rb, 1, b1);
      mfma_step(a[0], b0);
      if (MORE) load_step(K0{});
      __builtin_amdgcn_sched_barrier(0);
      wait_step(std::integral_constant<int, 2 * L + NX + R>{}, 1);
      read_r(buf, tap, rb, 2, b0);
      mfma_step(a[1], b1);
      if (MORE) load_step(K1{});
      __builtin_amdgcn_sched_barrier(0);
      wait_step(std::integral_constant<int, L + NX + 2 * R>{}, 2);
      read_r(buf, tap, rb, 3, b1);
      mfma_step(a[2], b0);
      if (MORE) load_step(K2{});
      __builtin_amdgcn_sched_barrier(0);
      wait_step(std::integral_constant<int, NX + 3 * R>{}, 3);
      mfma_step(a[3], b1);
      if (MORE) { load_step(K3{}); bump(); }
      __builtin_amdgcn_sched_barrier(0);
      if (LAST) {      // the next range has landed (older than the weight loads in flight), every LDS read has returned
        if (MORE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * MI) : "memory");
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
    int ra = 0;
    for (int i = 0; i + 1 < nrng; ++i) {
      round_body(i, ra, R0{}, T{}, T{}, F{});
      round_body(i, ra, R1{}, F{}, T{}, F{});
      round_body(i, ra, R2{}, F{}, T{}, T{});
      ra = ra == 2 ? 0 : ra + 1;
    }
    round_body(nrng - 1, ra, R0{}, F{}, T{}, F{});
    round_body(nrng - 1, ra, R1{}, F{}, T{}, F{});
    round_body(nrng - 1, ra, R2{}, F{}, F{}, T{});
  } else {
  // ---- prologue: x_0 landed; the weights of chunk 0 are the youngest VMEM operations
  stage_x(0, 0);
  __syncthreads();
  load_step(K0{}); load_step(K1{}); load_step(K2{}); load_step(K3{});
  bump();
  __builtin_amdgcn_sched_barrier(0);

  // chunk i.  VMEM issue order: [x_{i+1}: 4 pieces] a0' | a1' | a2' | a3' (MI loads each); counts = YOUNGER
  // operations at each wait.
  auto chunk_body = [&](int i, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int NX = MORE ? 4 : 0, R = MORE ? MI : 0, L = MI;
    const int buf = i & 1;
    const char* Bb = Bs + buf * B_ST;
    bf16x8 b0[4], b1[4];
    wait_step(std::integral_constant<int, 3 * L>{}, 0);            // younger: a1 a2 a3 of this chunk
    if (MORE) stage_x(buf ^ 1, i + 1);
    __builtin_amdgcn_sched_barrier(0);
    read_b(Bb, 0, b0);
    read_b(Bb, 1, b1);
    mfma_step(a[0], b0);
    if (MORE) load_step(K0{});
    __builtin_amdgcn_sched_barrier(0);
    wait_step(std::integral_constant<int, 2 * L + NX + R>{}, 1);   // younger: a2 a3, the x pieces, a0'
    read_b(Bb, 2, b0);
    mfma_step(a[1], b1);
    if (MORE) load_step(K1{});
    __builtin_amdgcn_sched_barrier(0);
    wait_step(std::integral_constant<int, L + NX + 2 * R>{}, 2);
    read_b(Bb, 3, b1);
    mfma_step(a[2], b0);
    if (MORE) load_step(K2{});
    __builtin_amdgcn_sched_barrier(0);
    wait_step(std::integral_constant<int, NX + 3 * R>{}, 3);
    mfma_step(a[3], b1);
    if (MORE) { load_step(K3{}); bump(); }
    __builtin_amdgcn_sched_barrier(0);
    // x_{i+1} has landed (older than the 4 MI weight loads), every LDS read of this chunk has returned
    if (MORE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * MI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i + 1 < nchunks; ++i) chunk_body(i, std::true_type{});
  chunk_body(nchunks - 1, std::false_type{});
  }

  // ---- epilogue.  A lane holds 4 consecutive channels of ONE pixel per register quad, so direct stores would
  // write 8-byte fragments to 32 different rows per instruction (measured: the 1x1 convs of the backbone ran at
  // 2.2-2.8 TB/s of effective traffic).  Each wave instead transposes its 32-pixel x (32 MI)-channel blocks
  // through the now idle stage memory -- fp32 (acc + bias), rows padded by 16 bytes: conflict-free 16-byte
  // writes -- and every lane then handles 8 consecutive channels of a pixel: residual read, ReLU, the single
  // rounding to bf16 and the store are 16 bytes per lane, whole 64 MI-byte row segments per pixel.
  {
    constexpr int CW = 32 * MI;                 // channels of this wave
    constexpr int PITCH = CW + 4;               // floats per pixel row in LDS (16 bytes of padding)
    constexpr int CPL = CW / 8;                 // 8-channel groups per pixel row = lanes per pixel
    constexpr int PPP = 64 / CPL;               // pixels per pass
    float* tw = reinterpret_cast<float*>(Bs) + wave * (32 * PITCH);
    const int mw = m0 + MI * wave * 32;         // first channel of this wave
    float4 bv[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = mw + mi * 32 + 8 * q + 4 * kh;
        bv[mi][q] = (bias && m < Cout) ? *reinterpret_cast<const float4*>(bias + m) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    const int pl = lane / CPL, cg = lane - pl * CPL;
    const int mc = mw + 8 * cg;                 // this lane's 8 channels in the read phase
    // residual rows of the whole tile are requested UP FRONT (round 4): with the load inside the pass loop every pass
    // was one exposed HBM round trip (load, s_waitcnt vmcnt(0), use, store: 16 of them in a row per wave at MI = 2) and
    // the 1x1 expand convs with a residual ran at 2.2 - 2.5 TB/s.  Lanes outside the tensor read row 0 (not used).
    constexpr int NPASS = 32 / PPP;
    bf16x8 rres[4][NPASS];
    if (residual) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          const int64_t n = n0 + ni * 32 + pass * PPP + pl;
          const bool ok = n < npix && mc < Cout;
          rres[ni][pass] = *reinterpret_cast<const bf16x8*>(residual + (ok ? n * Cout + mc : 0));
        }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[mi][ni][4 * q] + bv[mi][q].x, acc[mi][ni][4 * q + 1] + bv[mi][q].y,
                                       acc[mi][ni][4 * q + 2] + bv[mi][q].z, acc[mi][ni][4 * q + 3] + bv[mi][q].w);
          *reinterpret_cast<float4*>(tw + li * PITCH + mi * 32 + 8 * q + 4 * kh) = v;
        }
      // (same wave wrote and reads: LDS operations of a wave execute in order)
#pragma unroll
      for (int pass = 0; pass < 32 / PPP; ++pass) {
        const int px = pass * PPP + pl;
        const float4 v0 = *reinterpret_cast<const float4*>(tw + px * PITCH + 8 * cg);
        const float4 v1 = *reinterpret_cast<const float4*>(tw + px * PITCH + 8 * cg + 4);
        const int64_t n = n0 + ni * 32 + px;
        if (n < npix && mc < Cout) {
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (residual) {
            const bf16x8 rv = rres[ni][pass];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
          }
          bf16x8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (__bf16)(relu ? fmaxf(v[k], 0.f) : v[k]);
          *reinterpret_cast<bf16x8*>(out + n * Cout + mc) = o;
        }
      }
    }
  }
}

}  // namespace

extern "C" int tspn_pack_conv2d_frag_bf16(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                                          uint16_t* frag, void* stream) {
  TSPN_REQUIRE(w && frag, TSPN_EINVAL, "tspn_pack_conv2d_frag_bf16: null pointer");
  TSPN_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && KH * KW <= 64, TSPN_EINVAL,
               "tspn_pack_conv2d_frag_bf16: bad sizes");
  TSPN_REQUIRE(Cout % 32 == 0 && Cin % KC == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv2d_frag_bf16: needs Cout %% 32 == 0 and Cin %% 64 == 0 (Cout=%lld Cin=%lld)",
               (long long)Cout, (long long)Cin);
  const int64_t total = KH * KW * Cin * Cout;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv2d_frag_bf16_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), w, Cout, Cin,
                     KH * KW, reinterpret_cast<__bf16*>(frag));
  return tspn::check_launch("tspn_pack_conv2d_frag_bf16");
}

extern "C" int tspn_conv2d_nhwc_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                                     const uint16_t* frag, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                                     int64_t pad, const float* bias, const uint16_t* residual, int relu,
                                     uint16_t* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               TSPN_EINVAL, "tspn_conv2d_nhwc_bf16: bad sizes");
  TSPN_REQUIRE(KH * KW <= 64, TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_bf16: at most 64 taps");
  const int64_t OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_conv2d_nhwc_bf16: empty output");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && out, TSPN_EINVAL, "tspn_conv2d_nhwc_bf16: null pointer");
  TSPN_REQUIRE(Cin % KC == 0 && Cout % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_bf16: needs Cin %% 64 == 0 and Cout %% 32 == 0 (Cin=%lld Cout=%lld)", (long long)Cin,
               (long long)Cout);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag) && al16(out) && (!bias || al16(bias)) && (!residual || al16(residual)),
               TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_bf16: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && Cin < (1 << 24) && Cout < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_bf16: dimension too large");
  const int64_t npix = NB * OH * OW;
  // x goes out as buffer loads: 32-bit byte offsets from the first pixel of the image the tile starts in, num_records
  // 2^31 - 1 (an offset beyond it would read as padding zeros: silently wrong).  A 128-pixel tile spans at most
  // 127 / (OH OW) + 2 images, a tap reaches KH rows further.
  TSPN_REQUIRE(((int64_t)(BN - 1) / (OH * OW) + 2) * H * W * Cin * 2 + KH * W * Cin * 2 < (1LL << 31), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_bf16: the images one 128-pixel tile spans must stay below 2 GB (H=%lld W=%lld Cin=%lld)",
               (long long)H, (long long)W, (long long)Cin);
  // 64 rows per wave where Cout allows — except the 1x1 layers with K <= 256 (res2 - res4 expand convs): four K
  // chunks of MFMAs between a 64 KB operand prologue and a residual + store epilogue are latency-bound, and
  // the 32-row form at three workgroups per CU hides more of it (backbone conv time 8.59 -> 8.12 ms per 16
  // frames; 1024 <- 256 + residual 3.1 -> 3.3 TB/s, 256 <- 64 + residual 3.9 -> 4.9 TB/s; K = 1024 layers lose)
#ifndef TSPN_ROI_BF16_SHORTK_MI1
#define TSPN_ROI_BF16_SHORTK_MI1 256
#endif
  const int mi = (Cout % 64 == 0 && !(KH * KW == 1 && Cin <= TSPN_ROI_BF16_SHORTK_MI1)) ? 2 : 1;
  const int64_t tiles_m = tspn::ceil_div(Cout, 128 * mi), tiles_n = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_bf16: grid too large");
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), 0, TSPN_STREAM(stream),
                       reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(frag), bias,
                       reinterpret_cast<const __bf16*>(residual), reinterpret_cast<__bf16*>(out), (int)H, (int)W,
                       (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)OH, (int)OW, npix,
                       (int)tiles_m, (int)tiles_n, relu);
  };
  const bool rng = KH == 3 && KW == 3 && stride == 1 && pad == 1;
  if (mi == 2) { if (rng) launch(conv2d_nhwc_bf16_kernel<2, true>); else launch(conv2d_nhwc_bf16_kernel<2, false>); }
  else { if (rng) launch(conv2d_nhwc_bf16_kernel<1, true>); else launch(conv2d_nhwc_bf16_kernel<1, false>); }
  return tspn::check_launch("tspn_conv2d_nhwc_bf16");
}
