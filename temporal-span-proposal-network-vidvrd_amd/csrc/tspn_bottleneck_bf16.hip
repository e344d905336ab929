// Fused tail of a ResNet bottleneck block on bf16 operands (BASELINE cfg5 backbone):
//     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )        3x3 / pad 1 / stride 1, then 1x1 expand
// in ONE kernel (detectron2 BottleneckBlock.forward after conv1, modeling/backbone/resnet.py; FrozenBN folded into
// W / b by the caller).  Round 2 ran the two convolutions as separate launches: the 3x3 is MFMA-bound, the 1x1
// expand (K = 64..256, an output four times its input plus a residual read) is bound by its 8-byte-per-lane
// epilogue traffic (3.3 TB/s at res4, 374 TFLOP/s), and h2 made a round trip through HBM in between.  Here:
//   phase 2  the 3x3 conv exactly as conv2d_nhwc_bf16_kernel does it (implicit GEMM, M = CM output channels, N = 128
//            pixels per workgroup, K = 9 taps x CM in chunks of 64 channels of one tap; x through LDS by DMA with a
//            zero page for padding taps, fragment-major weights straight into MFMA operand registers, counted
//            vmcnt, one bare s_barrier per chunk) -- the WHOLE channel range of the tile stays in one workgroup:
//            wave (wm, wn) owns rows [64 wm, 64 wm + 64) x pixel blocks [NI wn, NI wn + NI), with
//            (WM, WN) = (4, 1) / (2, 2) / (1, 4) for CM = 256 / 128 / 64;
//   h2       relu(acc + b2) is rounded to bf16 ONCE (the same rounding point as the unfused chain) and written
//            into the now idle stage memory in the B-operand layout [CM / 8 groups][132 slots][8 bf16];
//   phase 3  the 1x1 expand as four passes of a [CM rows x 128 pixels] GEMM with K = CM: B fragments from the h2
//            image in LDS (conflict-free 16-byte reads), W3 fragments from L2 through a 4-deep register ring;
//   epilogue v_permlane32_swap pairs turn the MFMA layout (4 consecutive channels per lane and half-wave) into 8
//            consecutive channels per lane, so the residual read, ReLU, the single rounding and the store are
//            16 bytes per lane with no LDS transpose (CDNA4 guide, T21).
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

constexpr int THREADS = 256;
constexpr int BN = 128;                 // pixels per workgroup
constexpr int KC = 64;                  // channels per chunk
constexpr int SLP = 132;                // padded pixel slots per channel group
constexpr int B_ST = 8 * SLP * 16;      // bytes per x stage = per 64 channels of the h2 image

__device__ __bf16 g_zero_page_bt[128];  // source of padding taps (never written)

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
__device__ __forceinline__ void swap32(float& a, float& b) {
  // v_permlane32_swap_b32 vdst, src: lanes 32-63 of `a` <-> lanes 0-31 of `b`.  Inline asm, not
  // __builtin_amdgcn_permlane32_swap: hipcc 7.2 dropped the builtin's SECOND result here (it reused the first for both
  // halves: channels 4-7 / 12-15 of every 16 came out wrong; tools/probes/permlane32_swap_probe.hip shows the
  // instruction itself is fine).  The two v_nop are the VALU-write -> permlane-read wait states (CDNA4 guide, T21).
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int CM>
__global__ __launch_bounds__(THREADS, 2) void bottleneck_bf16_kernel(
    const __bf16* __restrict__ h1, const __bf16* __restrict__ Wf2, const float* __restrict__ bias2,
    const __bf16* __restrict__ Wf3, const float* __restrict__ bias3, const __bf16* __restrict__ residual,
    __bf16* __restrict__ out, int H, int W, int64_t npix) {
  constexpr int MI = 2;                                   // 32-row blocks per wave
  constexpr int WM = CM / 64, WN = 4 / WM, NI = 4 / WN;   // waves along rows / pixels, 32-pixel blocks per wave
  constexpr int CCH = CM / KC;                            // 64-channel chunks per tap = chunks of phase 3
  constexpr int NCHUNKS = 9 * CCH;
  constexpr int C4 = 4 * CM;
  extern __shared__ __attribute__((aligned(16))) char Bs[];   // max(2 stages, h2 image) = max(2, CCH) * B_ST

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

  // ---------------------------------------------------------------- phase 2: 3x3 conv, K = 9 taps x CM
  const char* wbase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    wbase[mi] = reinterpret_cast<const char*>(Wf2) + (int64_t)(MI * wm + mi) * NCHUNKS * 4096;
  // x pieces: the four pieces of a lane belong to ONE pixel (slot), channel groups bg, bg + 2, bg + 4, bg + 6
  const int slot = 64 * (wave & 1) + lane;
  const int bg = wave >> 1;
  int64_t pbase;
  unsigned tapmask = 0;
  {
    const int64_t n = n0 + slot;
    const bool okn = n < npix;
    const int64_t nc = okn ? n : 0;
    const int64_t nb = nc / ((int64_t)H * W);
    const int r = (int)(nc - nb * H * W);
    const int oh = r / W, ow = r - oh * W;
    pbase = ((nb * H + oh - 1) * (int64_t)W + ow - 1) * CM;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        if (okn && oh - 1 + a >= 0 && oh - 1 + a < H && ow - 1 + b >= 0 && ow - 1 + b < W) tapmask |= 1u << (a * 3 + b);
  }
  auto stage_x = [&](int buf, int i) {             // exactly four pieces per wave
    const int tap = i / CCH, c = i - tap * CCH;
    const int ta = tap / 3, tb = tap - ta * 3;
    const bool valid = (tapmask >> tap) & 1u;
    const __bf16* xs = valid ? h1 + pbase + ((int64_t)ta * W + tb) * CM + c * KC + 8 * bg : g_zero_page_bt + 8 * bg;
    char* dst = Bs + buf * B_ST + (bg * SLP + 64 * (wave & 1)) * 16;
#pragma unroll
    for (int p = 0; p < 4; ++p) glds16(xs + 16 * p, dst + 2 * p * SLP * 16);
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

  stage_x(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  load_step(K0{}); load_step(K1{}); load_step(K2{}); load_step(K3{});
  bump();
  __builtin_amdgcn_sched_barrier(0);

  // lane's fragment base inside a stage: slot = pixel block offset + li, channel-group half kh
  const int bofs = ((kh * SLP) + wn * NI * 32 + li) * 16;
  // chunk i.  VMEM issue order: [x_{i+1}: 4 pieces] a0' | a1' | a2' | a3' (MI loads each); counts = YOUNGER operations
  auto chunk_body = [&](int i, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int NX = MORE ? 4 : 0, R = MORE ? MI : 0, L = MI;
    const int buf = i & 1;
    const char* Bb = Bs + buf * B_ST + bofs;
    bf16x8 b0[NI], b1[NI];
    wait_w<3 * L>(a[0][0], a[0][1]);
    if (MORE) stage_x(buf ^ 1, i + 1);
    __builtin_amdgcn_sched_barrier(0);
    read_b(Bb, 0, b0);
    read_b(Bb, 2, b1);
    mfma_step(acc, a[0], b0);
    if (MORE) load_step(K0{});
    __builtin_amdgcn_sched_barrier(0);
    wait_w<2 * L + NX + R>(a[1][0], a[1][1]);
    read_b(Bb, 4, b0);
    mfma_step(acc, a[1], b1);
    if (MORE) load_step(K1{});
    __builtin_amdgcn_sched_barrier(0);
    wait_w<L + NX + 2 * R>(a[2][0], a[2][1]);
    read_b(Bb, 6, b1);
    mfma_step(acc, a[2], b0);
    if (MORE) load_step(K2{});
    __builtin_amdgcn_sched_barrier(0);
    wait_w<NX + 3 * R>(a[3][0], a[3][1]);
    mfma_step(acc, a[3], b1);
    if (MORE) { load_step(K3{}); bump(); }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * MI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i + 1 < NCHUNKS; ++i) chunk_body(i, std::true_type{});
  chunk_body(NCHUNKS - 1, std::false_type{});      // ends with a barrier: nobody reads the stages any more

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

  // ---------------------------------------------------------------- phase 3: 1x1 expand, four passes of CM rows, K = CM
  constexpr int KSTEPS = CM / 16;
  constexpr int RING = KSTEPS < 4 ? KSTEPS : 4;
  const char* Hb = Bs + bofs;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int rb0 = (wm * 4 + pass) * MI;                       // first 32-row block of this wave and pass
    const char* w3[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      w3[mi] = reinterpret_cast<const char*>(Wf3) + (int64_t)(rb0 + mi) * (CCH * 4096) + woff;
    f32x4 ar[RING][MI];
#pragma unroll
    for (int k = 0; k < RING; ++k)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) ar[k][mi] = *reinterpret_cast<const f32x4*>(w3[mi] + k * 1024);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
      bf16x8 b[NI];
      read_b(Hb, 2 * k, b);
      mfma_step(acc, ar[k % RING], b);
      if (k + RING < KSTEPS) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ar[k % RING][mi] = *reinterpret_cast<const f32x4*>(w3[mi] + (k + RING) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the ring a ring: no hoisting of later refills (register pressure)
    }
    // epilogue: 8 consecutive channels per lane after the half-wave swaps; + b3 + residual, ReLU, one rounding
#if defined(TSPN_BT_DIRECT_EPILOGUE)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch0 = 32 * (rb0 + mi) + 8 * q + 4 * kh;
        const float4 bv = *reinterpret_cast<const float4*>(bias3 + ch0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int64_t n = n0 + (wn * NI + ni) * 32 + li;
          if (n < npix) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(residual + n * C4 + ch0);
            bf16x4 o;
            o[0] = (__bf16)fmaxf(acc[mi][ni][4 * q] + bv.x + (float)rv[0], 0.f);
            o[1] = (__bf16)fmaxf(acc[mi][ni][4 * q + 1] + bv.y + (float)rv[1], 0.f);
            o[2] = (__bf16)fmaxf(acc[mi][ni][4 * q + 2] + bv.z + (float)rv[2], 0.f);
            o[3] = (__bf16)fmaxf(acc[mi][ni][4 * q + 3] + bv.w + (float)rv[3], 0.f);
            *reinterpret_cast<bf16x4*>(out + n * C4 + ch0) = o;
          }
        }
      }
#else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // bias in the MFMA layout (own channels 16 s + 4 kh + e and + 8), BEFORE the swaps: the permlane then reads
        // a VALU result (hazard padded by hipcc), never an MFMA result directly
        const int chm = 32 * (rb0 + mi) + 16 * s + 4 * kh;
        const float4 bv0 = *reinterpret_cast<const float4*>(bias3 + chm);
        const float4 bv1 = *reinterpret_cast<const float4*>(bias3 + chm + 8);
        const int ch0 = 32 * (rb0 + mi) + 16 * s + 8 * kh;      // first of this lane's 8 channels AFTER the swaps
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float lo[4] = {acc[mi][ni][8 * s] + bv0.x, acc[mi][ni][8 * s + 1] + bv0.y, acc[mi][ni][8 * s + 2] + bv0.z,
                         acc[mi][ni][8 * s + 3] + bv0.w};
          float hi[4] = {acc[mi][ni][8 * s + 4] + bv1.x, acc[mi][ni][8 * s + 5] + bv1.y, acc[mi][ni][8 * s + 6] + bv1.z,
                         acc[mi][ni][8 * s + 7] + bv1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) swap32(lo[e], hi[e]);
          const int64_t n = n0 + (wn * NI + ni) * 32 + li;
          if (n < npix) {
            const bf16x8 rv = *reinterpret_cast<const bf16x8*>(residual + n * C4 + ch0);
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf(v[j] + (float)rv[j], 0.f);
            *reinterpret_cast<bf16x8*>(out + n * C4 + ch0) = o;
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // at most NI residual loads in flight per group
      }
#endif
  }
}

template <int CM>
int launch(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, const uint16_t* frag2, const float* bias2,
           const uint16_t* frag3, const float* bias3, const uint16_t* residual, uint16_t* out, void* stream) {
  const int64_t npix = NB * H * W;
  const int64_t tiles = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_bottleneck_tail_bf16: grid too large");
  constexpr size_t smem = (size_t)(CM / KC > 2 ? CM / KC : 2) * B_ST;
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(bottleneck_bf16_kernel<CM>), smem, "tspn_bottleneck_tail_bf16"))
    return rc;
  hipLaunchKernelGGL(bottleneck_bf16_kernel<CM>, dim3((unsigned)tiles), dim3(THREADS), smem, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(h1), reinterpret_cast<const __bf16*>(frag2), bias2,
                     reinterpret_cast<const __bf16*>(frag3), bias3, reinterpret_cast<const __bf16*>(residual),
                     reinterpret_cast<__bf16*>(out), (int)H, (int)W, npix);
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
  if (CM == 256) return launch<256>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
  if (CM == 128) return launch<128>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
  return launch<64>(h1, NB, H, W, frag2, bias2, frag3, bias3, residual, out, stream);
}
