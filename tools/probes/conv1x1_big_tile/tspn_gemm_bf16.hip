// 1x1 convolution on bf16 operands as a BIG-TILE GEMM (the 1x1 layers of the RoI head's res5 --
// detectron2 Res5ROIHeads, which the reference configures in detectron/trainer.py:23-33 and owns no code for).
//
//     out[n][m] = bf16( act( sum_k x[n][k] w[m][k] + bias[m] + residual[n][m] ) )        channels-last, fp32 accumulation
//
// Why a second kernel next to conv2d_nhwc_bf16_kernel (tspn_roi_bf16.hip): that kernel's tile is 256 rows x 128 pixels
// with the weights streamed from L2 straight into registers -- per 64-channel chunk a workgroup pulls 32 KB of weights
// and 16 KB of x for 1024 MFMA cycles per SIMD, and res5's long-K 1x1 layers (2048 -> 512, 1024 -> 2048, 512 -> 2048 on
// 2400 RoIs = 117 600 pixels) ran at 0.63 - 0.87 PFLOP/s on it (profiles/r4/bottleneck_pipeline_study.md §4: a deeper
// operand ring changed nothing).  Here the tile is 256 rows x 256 pixels on eight waves (wave = 128 x 64 = 4 x 2 blocks
// of v_mfma_f32_32x32x16_bf16), BOTH operands go through LDS by 16-byte LDS-DMA pieces (32 KB per 32-channel chunk and
// 2048 MFMA cycles per SIMD: a third of the bytes per MFMA), in a ring of four stages filled three chunks ahead with
// counted vmcnt and one bare s_barrier per chunk -- the structure of conv3_bf16_big_kernel (tspn_bf16.hip), which
// holds 1.2 PFLOP/s.  One workgroup per CU (129 KB of LDS).
//
// Bit-identical to conv2d_nhwc_bf16_kernel: the same MFMA, a k-step = channels 16 s .. 16 s + 15 with lane half kh
// taking channels 8 kh .. 8 kh + 7, k-steps in channel order on one accumulator chain per output block, and the same
// epilogue arithmetic ((acc + bias) + residual, ReLU, one rounding to nearest even).
//
// Epilogue without an LDS transpose: the weight ROWS are permuted inside every 32-row block at pack time
// (tspn_pack_conv1x1_rows_bf16: row slot i holds channel 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3)), so that the
// accumulator registers 4 q + j of lane (pixel li, half kh) -- MFMA rows 8 q + 4 kh + j -- are the 16 CONSECUTIVE
// channels 16 kh .. 16 kh + 15 of the block: bias (from LDS), residual read, ReLU, rounding and store are 32 contiguous
// bytes per lane; the lane pair of a pixel finishes a 64-byte sector in two back-to-back instructions and the two
// blocks of a 128-byte line follow each other (what made the fused bottleneck tail write exactly its output,
// tspn_bottleneck_bf16.hip).  The residual rows of the first pixel half are requested before the first operand piece
// (they are the oldest vector-memory operations: no counted wait changes), those of the second half right after the
// main loop into the registers the operand fragments leave.
#include <algorithm>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int Q_THREADS = 512;
constexpr int Q_BM = 256, Q_BN = 256;
constexpr int Q_KC = 32;                       // channels per chunk = two MFMA k-steps
constexpr int Q_KG = Q_KC / 8;                 // 8-channel groups per chunk
constexpr int Q_SLP = 260;                     // x slots per group (256 + padding: the two lane halves hit different banks)
constexpr int Q_A_ST = Q_KG * Q_BM * 16;       // 16384
constexpr int Q_X_ST = Q_KG * Q_SLP * 16;      // 16640
constexpr int Q_ST = Q_A_ST + Q_X_ST;          // 33024
constexpr int Q_NST = 4;
constexpr int Q_BIAS_OFF = Q_NST * Q_ST;       // 256 floats of bias behind the ring
constexpr size_t Q_SMEM = (size_t)Q_BIAS_OFF + Q_BM * 4;

// the channel that row slot i of a 32-row block holds
__host__ __device__ constexpr int row_perm(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// w fp32 [Cout][Cin] -> bf16 [Cin / 8 groups][Mp row slots][8], Mp = Cout rounded up to 32, rows permuted per block
__global__ void pack_conv1x1_rows_bf16_kernel(const float* __restrict__ w, int64_t Cout, int64_t Cin, int64_t Mp,
                                              __bf16* __restrict__ packed) {
  const int64_t total = Cin * Mp;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(o & 7);
    const int64_t r = (o >> 3) % Mp, g = (o >> 3) / Mp;
    const int64_t ch = (r & ~(int64_t)31) + row_perm((int)(r & 31));
    packed[o] = ch < Cout ? (__bf16)w[ch * Cin + 8 * g + j] : (__bf16)0.f;
  }
}

template <bool RES>
__global__ __launch_bounds__(Q_THREADS, 1) void conv1x1_big_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wr, const float* __restrict__ bias,
    const __bf16* __restrict__ residual, __bf16* __restrict__ out, int Cin, int M, int Mp, int64_t npix, int H, int W,
    int OH, int OW, int stride, int tiles_m, int tiles_n, int GM, int relu, int w_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // workgroup -> tile: every XCD takes a contiguous range of tile indices; inside it groups of GM weight panels x all
  // pixel tiles, the GM panels of a pixel tile next to each other (its x tile is read GM times in a row from one L2)
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
  const int m0 = tile_m * Q_BM;
  const int64_t n0 = (int64_t)tile_n * Q_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, kh = lane >> 5;

  // ---- epilogue addresses: lane (li, kh), block (mi, ni) = pixel n0 + 64 wn + 32 ni + li, channels m0 + 128 wm +
  // 32 mi + 16 kh + 0..15
  const int cw = wm * 128 + 16 * kh;                       // + 32 mi: first channel of the lane inside the tile
  auto pix_of = [&](int ni) { return n0 + wn * 64 + ni * 32 + li; };
  bf16x8 rres0[4][2];                                      // residual rows of pixel half ni = 0
  if constexpr (RES) {
    const int64_t n = pix_of(0);
    const int64_t row = (n < npix ? n : 0) * (int64_t)M;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int c = m0 + cw + 32 * mi;
      const __bf16* rp = residual + row + (c < M ? c : 0);
      rres0[mi][0] = *reinterpret_cast<const bf16x8*>(rp);
      rres0[mi][1] = *reinterpret_cast<const bf16x8*>(rp + 8);
    }
  }
  if (tid < Q_BM / 4) {                                    // bias of the tile's 256 channels -> LDS (published by the first barrier)
    const int c = m0 + 4 * tid;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && c < M) bv = *reinterpret_cast<const f32x4*>(bias + c);
    *reinterpret_cast<f32x4*>(smem + Q_BIAS_OFF + 16 * tid) = bv;
  }

  // ---- operand pieces: 16 weight pieces + 16 x pieces of 64 x 16 bytes per chunk, four per wave.  Wave w stages
  // weight pieces 2 w, 2 w + 1 (channel group w >> 1, row quarters 2 (w & 1) + {0, 1}) and the x pieces of the same
  // group and pixel quarters.
  const int pg = wave >> 1;
  // buffer loads (buffer_load_dwordx4 ... offen lds): one descriptor per operand, a fixed 32-bit lane offset per piece and
  // the chunk as the scalar offset -- 110 - 140 issue cycles per piece beside MFMAs instead of the 175 - 235 of
  // global_load_lds with a 64-bit address per lane (tools/probes/lds_dma_issue_probe.hip)
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wr), 0, w_bytes, 0x00020000);
  const int64_t img = (int64_t)OH * OW;
  const int64_t base_pix = ((n0 < npix ? n0 : 0) / img) * H * W;       // first pixel of the image the tile starts in
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x) + base_pix * Cin, 0, 0x7fffffff, 0x00020000);
  unsigned aoff[2], xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int quarter = 2 * (wave & 1) + i;
    int r = m0 + 64 * quarter + lane;
    r = r < Mp ? r : 0;
    aoff[i] = (unsigned)(((int64_t)pg * Mp + r) * 16);
    int64_t n = n0 + 64 * quarter + lane;
    n = n < npix ? n : npix - 1;
    int64_t ip = n;
    if (stride != 1 || H != OH || W != OW) {
      const int64_t nb = n / img;
      const int rr = (int)(n - nb * img);
      const int oh = rr / OW, ow = rr - oh * OW;
      ip = (nb * H + (int64_t)oh * stride) * W + (int64_t)ow * stride;
    }
    xoff[i] = (unsigned)((ip - base_pix) * Cin * 2 + 16 * pg);
  }
  const int a_step = Q_KG * Mp * 16;                         // bytes per chunk
  int a_soff = 0, x_soff = 0;
#if defined(TSPN_Q_ABL_ROT)          // probe build: every pixel tile walks the K chunks from its own starting point (wrong sums)
  int rot_c = (tile_n * TSPN_Q_ABL_ROT) % (Cin / Q_KC);
#endif
  auto bglds16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff, char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, (int)voff, soff, 0, 0);
  };
  auto stage_chunk = [&](int st) {
    char* sa = smem + st * Q_ST + (2 * wave) * 1024;
    char* sx = smem + st * Q_ST + Q_A_ST + (pg * Q_SLP + 128 * (wave & 1)) * 16;
#if defined(TSPN_Q_ABL_ROT)
    a_soff = rot_c * a_step; x_soff = rot_c * Q_KC * 2;
    rot_c = rot_c + 1 == Cin / Q_KC ? 0 : rot_c + 1;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#if !defined(TSPN_Q_ABL_NOW)         // probe build: no weight pieces
      bglds16(rsrc_w, aoff[i], a_soff, sa + i * 1024);
#endif
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#if !defined(TSPN_Q_ABL_NOX)         // probe build: no x pieces
      bglds16(rsrc_x, xoff[i], x_soff, sx + i * 1024);
#endif
    }
    a_soff += a_step;
    x_soff += Q_KC * 2;
  };
  auto wait_keep = [&](auto chunks_tag) {                  // four pieces in flight per chunk and wave
    constexpr int CH = decltype(chunks_tag)::value;
#if defined(TSPN_Q_ABL_NOW) || defined(TSPN_Q_ABL_NOX)
    wait_vmcnt<2 * CH>();
#else
    wait_vmcnt<4 * CH>();
#endif
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nchunks = Cin / Q_KC;
  // Software pipeline across chunks (fragment registers a / b [k-step of the chunk]):
  //   top of chunk c:  k-step 0 of chunk c is in registers
  //   MFMAs of k-step 0, the fragment reads of k-step 1 between them
  //   wait for the DMA of chunk c + 1, barrier, issue the DMA of chunk c + 3 (into the stage chunk c - 1 has left)
  //   MFMAs of k-step 1, the reads of k-step 0 of chunk c + 1 between them
  bf16x8 a[2][4], b[2][2];
  auto load_ks = [&](int st, int ks) {
    const char* Ab = smem + st * Q_ST + ((2 * ks + kh) * Q_BM + wm * 128 + li) * 16;
    const char* Xb = smem + st * Q_ST + Q_A_ST + ((2 * ks + kh) * Q_SLP + wn * 64 + li) * 16;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[ks][mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * 16);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b[ks][ni] = *reinterpret_cast<const bf16x8*>(Xb + ni * 32 * 16);
  };
  auto mfma_ks = [&](int ks) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][mi], b[ks][ni], acc[mi][ni], 0, 0, 0);
  };

  stage_chunk(0);
  if (nchunks > 1) stage_chunk(1);
  if (nchunks > 2) stage_chunk(2);
  if (nchunks > 2) wait_keep(K2{}); else if (nchunks > 1) wait_keep(K1{}); else wait_keep(K0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias rows are in LDS before the barrier publishes them
  __builtin_amdgcn_s_barrier();

  load_ks(0, 0);
  int c = 0;
  for (; c + 3 < nchunks; ++c) {        // steady state: chunks c + 1 .. c + 3 exist
    const int st = c & 3;
    mfma_ks(0);
    load_ks(st, 1);
    // 8 MFMAs, one fragment read behind each of the first six
#define TSPN_MR(NM, NR)                                \
  __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);  \
  __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
    TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(2, 0)
    __builtin_amdgcn_sched_barrier(0);
    wait_keep(K1{});                    // chunk c + 2 may still fly; c + 1 has landed
    __builtin_amdgcn_s_barrier();
    if (wm == 0) stage_chunk((c + 3) & 3);          // the two waves of a SIMD (w, w + 4) issue their pieces at different times
    __builtin_amdgcn_sched_barrier(0);
    mfma_ks(1);
    load_ks((c + 1) & 3, 0);
    TSPN_MR(2, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 1) TSPN_MR(1, 0)
#undef TSPN_MR
    __builtin_amdgcn_sched_barrier(0);
    if (wm != 0) stage_chunk((c + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; c < nchunks; ++c) {            // tail: nothing left to issue
    const int st = c & 3;
    load_ks(st, 1);
    mfma_ks(0);
    if (c + 2 < nchunks) wait_keep(K1{}); else wait_keep(K0{});
    __builtin_amdgcn_s_barrier();
    if (c + 1 < nchunks) load_ks((c + 1) & 3, 0);
    mfma_ks(1);
  }

  // ---- epilogue
  bf16x8 rres1[4][2];                                      // residual rows of pixel half ni = 1
  if constexpr (RES) {
    const int64_t n = pix_of(1);
    const int64_t row = (n < npix ? n : 0) * (int64_t)M;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int cc = m0 + cw + 32 * mi;
      const __bf16* rp = residual + row + (cc < M ? cc : 0);
      rres1[mi][0] = *reinterpret_cast<const bf16x8*>(rp);
      rres1[mi][1] = *reinterpret_cast<const bf16x8*>(rp + 8);
    }
  }
  const float* const bl = reinterpret_cast<const float*>(smem + Q_BIAS_OFF);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = pix_of(ni);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int ct = cw + 32 * mi;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + ct + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] = acc[mi][ni][4 * q + j] + bv[j];
      }
      if constexpr (RES) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 rv = ni == 0 ? rres0[mi][h] : rres1[mi][h];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[8 * h + j] += (float)rv[j];
        }
      }
      bf16x8 o0, o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o0[j] = (__bf16)(relu ? fmaxf(v[j], 0.f) : v[j]);
        o1[j] = (__bf16)(relu ? fmaxf(v[8 + j], 0.f) : v[8 + j]);
      }
      if (n < npix && m0 + ct < M) {
        __bf16* op = out + n * (int64_t)M + m0 + ct;
        *reinterpret_cast<bf16x8*>(op) = o0;
        *reinterpret_cast<bf16x8*>(op + 8) = o1;
      }
    }
  }
}

}  // namespace

extern "C" int tspn_pack_conv1x1_rows_bf16(const float* w, int64_t Cout, int64_t Cin, uint16_t* rows, void* stream) {
  TSPN_REQUIRE(w && rows, TSPN_EINVAL, "tspn_pack_conv1x1_rows_bf16: null pointer");
  TSPN_REQUIRE(Cout > 0 && Cin > 0, TSPN_EINVAL, "tspn_pack_conv1x1_rows_bf16: bad sizes");
  TSPN_REQUIRE(Cout % 32 == 0 && Cin % Q_KC == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv1x1_rows_bf16: needs Cout %% 32 == 0 and Cin %% 32 == 0 (Cout=%lld Cin=%lld)",
               (long long)Cout, (long long)Cin);
  const int64_t total = Cin * Cout;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv1x1_rows_bf16_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), w, Cout, Cin, Cout,
                     reinterpret_cast<__bf16*>(rows));
  return tspn::check_launch("tspn_pack_conv1x1_rows_bf16");
}

extern "C" int tspn_conv1x1_big_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                                     const uint16_t* rows, int64_t Cout, int64_t stride, const float* bias,
                                     const uint16_t* residual, int relu, uint16_t* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && stride > 0, TSPN_EINVAL,
               "tspn_conv1x1_big_bf16: bad sizes");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && rows && out, TSPN_EINVAL, "tspn_conv1x1_big_bf16: null pointer");
  TSPN_REQUIRE(Cin % Q_KC == 0 && Cout % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv1x1_big_bf16: needs Cin %% 32 == 0 and Cout %% 32 == 0 (Cin=%lld Cout=%lld)", (long long)Cin,
               (long long)Cout);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(rows) && al16(out) && (!bias || al16(bias)) && (!residual || al16(residual)),
               TSPN_EUNSUPPORTED, "tspn_conv1x1_big_bf16: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && Cin < (1 << 24) && Cout < (1 << 24) && Cin * Cout * 2 < (1LL << 31) &&
                   H * W * Cin * 2 < (1LL << 30),
               TSPN_EUNSUPPORTED, "tspn_conv1x1_big_bf16: dimension too large");
  const int64_t OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t npix = NB * OH * OW;
  const int64_t tiles_m = tspn::ceil_div(Cout, Q_BM), tiles_n = tspn::ceil_div(npix, Q_BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv1x1_big_bf16: grid too large");
  const int gm = (int)std::min<int64_t>(tiles_m, 4);
  const void* fn = residual ? reinterpret_cast<const void*>(conv1x1_big_bf16_kernel<true>)
                            : reinterpret_cast<const void*>(conv1x1_big_bf16_kernel<false>);
  static tspn::LdsLimit lds[2];
  if (int rc = lds[residual ? 1 : 0].ensure(fn, Q_SMEM, "tspn_conv1x1_big_bf16")) return rc;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(Q_THREADS), Q_SMEM, TSPN_STREAM(stream),
                       reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(rows), bias,
                       reinterpret_cast<const __bf16*>(residual), reinterpret_cast<__bf16*>(out), (int)Cin, (int)Cout,
                       (int)Cout, npix, (int)H, (int)W, (int)OH, (int)OW, (int)stride, (int)tiles_m, (int)tiles_n, gm, relu,
                       (int)(Cin * Cout * 2));
  };
  if (residual) launch(conv1x1_big_bf16_kernel<true>); else launch(conv1x1_big_bf16_kernel<false>);
  return tspn::check_launch("tspn_conv1x1_big_bf16");
}
