// RoI feature head, SURVEY.md §8 row f4 (first slice): from a C4 feature map to per-tracklet RoI features.
//
// The reference has no code of its own here: lib/detectron/trainer.py:23-33 configures detectron2's
// R101-C4 model, whose ROI head is  ROIAlign(14x14, aligned, adaptive sampling) -> res5 (three
// bottleneck blocks, the first with stride 2, FrozenBN) -> mean over 7x7 -> 2048-d feature per box.
// These kernels are the two operators that head needs on the GPU (gfx950, fp32):
//   * conv2d_nhwc_kernel: KHxKW conv (1x1 / 3x3, stride 1 / 2, zero padding) on channels-last tensors as
//     an implicit GEMM on v_mfma_f32_32x32x2_f32 -- M = output channels, N = output pixels, K = taps x Cin;
//     128 x 128 tiles, 4 waves x (2 x 2 blocks of 32 x 32), K chunk = 16 channels of ONE tap, operand tiles
//     staged by LDS-DMA (weights: 512-byte rows; x: one 16-byte piece per (pixel, channel group), the source
//     of a padding tap is a zero page), double-buffered; epilogue fused: + bias (BatchNorm folded by the
//     host), + residual, ReLU, float4 stores into the channels-last output.
//   * roi_align_nhwc_kernel: detectron2's ROIAlign (aligned or legacy, fixed or adaptive sampling grid) on a
//     channels-last feature map; one workgroup per output bin row, a thread = 4 channels.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int BM = 128, BN = 128;
constexpr int SLP = 132;               // padded pixel slots per channel group (128 used)
// KC = channels per K chunk (template parameter: 32 where Cin allows, else 16): 2 KC MFMAs per wave
// between two barriers; LDS 2 x (KC x 128 + KC/4 x 132 x 4) floats = 33 / 66 KB
template <int KC>
constexpr size_t smem_bytes() { return sizeof(float) * 2 * (KC * BM + (KC / 4) * SLP * 4); }

__device__ float g_zero_page[64];      // source of padding taps (never written)

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// w [Cout][Cin][KH][KW] (torch Conv2d layout) -> packed [KH*KW][Cin][Cout]
__global__ void pack_conv2d_kernel(const float* __restrict__ w, int64_t Cout, int64_t Cin, int64_t KH,
                                   int64_t KW, float* __restrict__ packed) {
  const int64_t total = KH * KW * Cin * Cout;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t co = o % Cout;
    const int64_t ci = (o / Cout) % Cin;
    const int64_t tap = o / (Cout * Cin);
    packed[o] = w[(co * Cin + ci) * (KH * KW) + tap];
  }
}

template <int KC>
__global__ __launch_bounds__(THREADS, 2) void conv2d_nhwc_kernel(
    const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bias,
    const float* __restrict__ residual, float* __restrict__ out, int H, int W, int Cin, int Cout, int KH,
    int KW, int stride, int pad, int OH, int OW, int64_t npix, int tiles_m, int tiles_n, int relu) {
  constexpr int NG = KC / 4;             // 4-channel groups per chunk
  constexpr int A_ST = KC * BM;          // floats
  constexpr int B_ST = NG * SLP * 4;     // floats
  constexpr int NP = KC / 8;             // DMA pieces per wave and operand
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);   // [2][KC ch][128 m]
  float* Bs = As + 2 * A_ST;                         // [2][KC/4 groups][132 slots][4 ch]

  // workgroup -> tile: bijective XCD remap, then groups of GM weight panels x all pixel tiles
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
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- weight pieces: wave w stages rows 2 NP w .. 2 NP (w+1) - 1 of the KC x 128 tile, two rows per piece
  const int am = (lane & 31) * 4;
  const int amc = m0 + am < Cout ? m0 + am : 0;       // columns beyond Cout: re-read column 0, never stored
  const int arow = 2 * NP * wave + (lane >> 5);       // + 2 i
  // ---- x pieces: all pieces of a lane belong to ONE output pixel (slot), channel groups bg, bg + 2, ...
  const int slot = 64 * (wave & 1) + lane;
  const int bg = wave >> 1;
  int64_t pbase;          // offset (floats) of input pixel (ih0, iw0) of this lane's output pixel, channel 0
  unsigned long long tapmask = 0;   // bit (kh * KW + kw): the tap lies inside the image
  {
    const int64_t n = n0 + slot;
    const bool okn = n < npix;
    const int64_t nc = okn ? n : 0;
    const int64_t nb = nc / ((int64_t)OH * OW);
    const int r = (int)(nc - nb * OH * OW);
    const int oh = r / OW, ow = r - oh * OW;
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    pbase = ((nb * H + ih0) * (int64_t)W + iw0) * Cin;
    for (int a = 0; a < KH; ++a)
      for (int b = 0; b < KW; ++b)
        if (okn && ih0 + a >= 0 && ih0 + a < H && iw0 + b >= 0 && iw0 + b < W) tapmask |= 1ull << (a * KW + b);
  }

  const int cchunks = Cin / KC;
  const int nchunks = KH * KW * cchunks;
  // chunk i = (tap = i / cchunks, c = i % cchunks)
  auto stage = [&](int buf, int i) {
    const int tap = i / cchunks, c = i - tap * cchunks;
    const int ta = tap / KW, tb = tap - ta * KW;
    const float* wsrc = Wp + ((int64_t)tap * Cin + c * KC + arow) * Cout + amc;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      glds16(wsrc + 2 * p * (int64_t)Cout, As + buf * A_ST + (2 * NP * wave + 2 * p) * BM);
    const bool valid = (tapmask >> tap) & 1ull;
    const float* xs = valid ? x + pbase + ((int64_t)ta * W + tb) * Cin + c * KC + 4 * bg : g_zero_page + 4 * bg;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      glds16(xs + 8 * p, Bs + buf * B_ST + ((bg + 2 * p) * SLP + 64 * (wave & 1)) * 4);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  struct Frag {
    float a[2][2];   // [mi][row of the lane's channel pair]
    float2 b[2];     // [ni]
  };
  auto read_frag = [&](const float* Ab, const float* Bb, int g) {
    Frag f;
    const int r0 = 4 * g + 2 * kh;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f.a[mi][0] = Ab[r0 * BM + mi * 32];
      f.a[mi][1] = Ab[(r0 + 1) * BM + mi * 32];
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) f.b[ni] = *reinterpret_cast<const float2*>(Bb + (g * SLP + ni * 32) * 4);
    return f;
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();
  // chunk i: 2 KC MFMAs per wave on buffer i & 1, the DMA pieces of chunk i+1 and the fragment reads of
  // the next channel group issued between them (1 MFMA : 1 LDS read, a DMA piece behind every other MFMA of
  // the first groups)
  auto chunk_body = [&](int i, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int buf = i & 1;
    if (MORE) stage(buf ^ 1, i + 1);
    const float* Ab = As + buf * A_ST + wm * 64 + li;
    const float* Bb = Bs + buf * B_ST + (wn * 64 + li) * 4 + 2 * kh;
    Frag cur = read_frag(Ab, Bb, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      Frag nxt = cur;
      if (g + 1 < NG) nxt = read_frag(Ab, Bb, g + 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float b0 = e ? cur.b[0].y : cur.b[0].x;
        const float b1 = e ? cur.b[1].y : cur.b[1].x;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0][e], b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0][e], b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[1][e], b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[1][e], b1, acc[1][1], 0, 0, 0);
      }
#define TSPN_G(NVM)                                     \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
  __builtin_amdgcn_sched_group_barrier(0x020, NVM, 0);
      // (the 2 NP pieces of this wave: four per channel group)
      if (g < NP / 2 && MORE) { TSPN_G(1) TSPN_G(0) TSPN_G(1) TSPN_G(0) TSPN_G(1) TSPN_G(0) TSPN_G(1) TSPN_G(0) }
      else { TSPN_G(0) TSPN_G(0) TSPN_G(0) TSPN_G(0) TSPN_G(0) TSPN_G(0) TSPN_G(0) TSPN_G(0) }
#undef TSPN_G
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
    __syncthreads();
  };
  for (int i = 0; i + 1 < nchunks; ++i) chunk_body(i, std::true_type{});
  chunk_body(nchunks - 1, std::false_type{});

  // ---- epilogue: a lane holds 4 consecutive output channels of one pixel per register quad
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + li;
    if (n >= npix) continue;
    float* orow = out + n * Cout;
    const float* rrow = residual ? residual + n * Cout : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 64 + mi * 32 + 8 * q + 4 * kh;
        if (m >= Cout) continue;                       // Cout % 4 == 0: a quad is inside or outside
        float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2],
                               acc[mi][ni][4 * q + 3]);
        if (bias) {
          const float4 bv = *reinterpret_cast<const float4*>(bias + m);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (rrow) {
          const float4 rv = *reinterpret_cast<const float4*>(rrow + m);
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(orow + m) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Fast variant for Cout % 32 == 0: FRAGMENT-MAJOR weights loaded straight into MFMA operand registers
// (the recipe of the temporal conv, tspn_wino63.hip).  Tile 128 output channels x 128 pixels; wave w = rows [32 w, 32 w + 32)
// x all 128 pixels (4 accumulator blocks), so a wave's weight slice is private and never needs LDS:
//     Wf[Cout / 32][tap][Cin / 16][lane = 32 kh + li][8]  =  w[32 mb + li][16 c + 4 g + 2 kh + r][tap],  8 = (g, r)
// is two global_load_dwordx4 per lane and chunk (one 2-KiB line per wave), refilled for chunk i+1 as soon
// as the MFMAs of chunk i have read them (rolling, counted vmcnt).  Only the x tile goes through LDS
// (2 DMA pieces per wave and chunk, double-buffered, bare s_barrier per chunk).
template <int OFF>
__device__ __forceinline__ void load_wfrag(f32x4& dst, unsigned lane_off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait_w(f32x4& r) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(VM));
}

// w [Cout][Cin][KH][KW] -> fragment-major [Cout/32][KH*KW][Cin/16][64][8]
__global__ void pack_conv2d_frag_kernel(const float* __restrict__ w, int64_t Cout, int64_t Cin, int64_t ntaps,
                                        float* __restrict__ packed) {
  const int64_t total = ntaps * Cin * Cout;
  const int64_t cch = Cin / 16;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(o & 1), g = (int)((o >> 1) & 3), lane = (int)((o >> 3) & 63);
    const int64_t q = o >> 9;
    const int64_t c = q % cch, tap = (q / cch) % ntaps, mb = q / (cch * ntaps);
    const int64_t co = 32 * mb + (lane & 31), ci = 16 * c + 4 * g + 2 * (lane >> 5) + r;
    packed[o] = w[(co * Cin + ci) * ntaps + tap];
  }
}

// Stem form (Cin <= 4, e.g. RGB): the image has 4 channels per pixel (zero padded) and one K chunk spans FOUR
// TAPS x 4 channels -- a 16-byte DMA piece is exactly one (pixel, tap) -- so a 7x7 stem needs 13 chunks, not 49:
//   frag[Cout/32][ceil(taps/4)][64 lanes][8 = (g, r)] = w[32 mb + li][2 kh + r][tap = 4 c + g]   (0 beyond Cin / taps)
__global__ void pack_conv2d_frag_cin4_kernel(const float* __restrict__ w, int64_t Cout, int64_t Cin, int64_t ntaps,
                                             float* __restrict__ packed) {
  const int64_t nch = (ntaps + 3) / 4;
  const int64_t total = (Cout / 32) * nch * 512;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(o & 1), g = (int)((o >> 1) & 3), lane = (int)((o >> 3) & 63);
    const int64_t q = o >> 9;
    const int64_t c = q % nch, mb = q / nch;
    const int64_t co = 32 * mb + (lane & 31), ci = 2 * (lane >> 5) + r, tap = 4 * c + g;
    packed[o] = (ci < Cin && tap < ntaps) ? w[(co * Cin + ci) * ntaps + tap] : 0.f;
  }
}

constexpr int F_B_ST = 4 * SLP * 4;    // floats per x stage: [4 groups][132 slots][4 ch]

template <bool CIN4>
__global__ __launch_bounds__(THREADS, 2) void conv2d_nhwc_frag_kernel(
    const float* __restrict__ x, const float* __restrict__ Wf, const float* __restrict__ bias,
    const float* __restrict__ residual, float* __restrict__ out, int H, int W, int Cin, int Cout, int KH,
    int KW, int stride, int pad, int OH, int OW, int64_t npix, int tiles_m, int tiles_n, int relu) {
  __shared__ __attribute__((aligned(16))) float Bs[2 * F_B_ST];

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

  const int cchunks = Cin >> 4;
  const int nchunks = CIN4 ? (KH * KW + 3) / 4 : KH * KW * cchunks;
  // weight fragment stream of this wave: chunk i = (tap, c) is line i of its block (tap-major like i)
  const char* wbase;
  const unsigned woff = lane * 32;
  {
    int mb = (m0 >> 5) + wave;
    mb = mb < (Cout >> 5) ? mb : 0;               // rows beyond Cout: re-read block 0, never stored
    wbase = reinterpret_cast<const char*>(Wf) + (int64_t)mb * nchunks * 2048;
  }
  // x pieces: both pieces of a lane belong to ONE output pixel (slot), channel groups bg and bg + 2
  const int slot = 64 * (wave & 1) + lane;
  const int bg = wave >> 1;
  int64_t pbase;
  unsigned long long tapmask = 0;
  {
    const int64_t n = n0 + slot;
    const bool okn = n < npix;
    const int64_t nc = okn ? n : 0;
    const int64_t nb = nc / ((int64_t)OH * OW);
    const int r = (int)(nc - nb * OH * OW);
    const int oh = r / OW, ow = r - oh * OW;
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    pbase = ((nb * H + ih0) * (int64_t)W + iw0) * (CIN4 ? 4 : Cin);
    for (int a = 0; a < KH; ++a)
      for (int b = 0; b < KW; ++b)
        if (okn && ih0 + a >= 0 && ih0 + a < H && iw0 + b >= 0 && iw0 + b < W) tapmask |= 1ull << (a * KW + b);
  }
  auto stage_x = [&](int buf, int i) {             // exactly two pieces per wave (padding taps read the zero page)
    if constexpr (CIN4) {                          // group g of chunk i = tap 4 i + g, the pixel's 4 channels
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int tap = 4 * i + bg + 2 * p;
        const int ta = tap / KW, tb = tap - ta * KW;
        const bool valid = tap < KH * KW && ((tapmask >> tap) & 1ull);
        const float* xs = valid ? x + pbase + ((int64_t)ta * W + tb) * 4 : g_zero_page;
        glds16(xs, Bs + buf * F_B_ST + ((bg + 2 * p) * SLP + 64 * (wave & 1)) * 4);
      }
    } else {
      const int tap = i / cchunks, c = i - tap * cchunks;
      const int ta = tap / KW, tb = tap - ta * KW;
      const bool valid = (tapmask >> tap) & 1ull;
      const float* xs = valid ? x + pbase + ((int64_t)ta * W + tb) * Cin + c * 16 + 4 * bg : g_zero_page + 4 * bg;
      glds16(xs, Bs + buf * F_B_ST + (bg * SLP + 64 * (wave & 1)) * 4);
      glds16(xs + 8, Bs + buf * F_B_ST + ((bg + 2) * SLP + 64 * (wave & 1)) * 4);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;

  f32x4 a01, a23;       // weights of channel groups (0, 1) and (2, 3): [0..1] / [2..3] = rows r = 0, 1 of a group
  auto read_b = [&](const float* Bb, int g, float2 (&b)[4]) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const float2*>(Bb + (g * SLP + ni * 32) * 4);
  };
  auto mfma_group = [&](float a0, float a1, const float2 (&b)[4]) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[ni].x, acc[ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[ni].y, acc[ni], 0, 0, 0);
  };

  // ---- prologue: x_0 landed; the weights of chunk 0 are the two youngest VMEM operations
  stage_x(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed before the barrier publishes them
  __syncthreads();
  load_wfrag<0>(a01, woff, wbase);
  load_wfrag<16>(a23, woff, wbase);
  wbase += 2048;
  __builtin_amdgcn_sched_barrier(0);

  // chunk i.  VMEM issue order: [x_{i+1}: 2 pieces] w01' | w23'; counts = YOUNGER operations at each wait.
  auto chunk_body = [&](int i, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int NX = MORE ? 2 : 0, NW = MORE ? 1 : 0;
    const int buf = i & 1;
    const float* Bb = Bs + buf * F_B_ST + li * 4 + 2 * kh;
    float2 b0[4], b1[4];
    wait_w<1>(a01);                                   // younger: w23 of this chunk
    if (MORE) stage_x(buf ^ 1, i + 1);
    __builtin_amdgcn_sched_barrier(0);
    read_b(Bb, 0, b0);
    read_b(Bb, 1, b1);
    mfma_group(a01[0], a01[1], b0);
    read_b(Bb, 2, b0);
    mfma_group(a01[2], a01[3], b1);
    if (MORE) load_wfrag<0>(a01, woff, wbase);
    __builtin_amdgcn_sched_barrier(0);
    wait_w<NX + NW>(a23);                             // younger: the x pieces and w01' issued above
    read_b(Bb, 3, b1);
    mfma_group(a23[0], a23[1], b0);
    mfma_group(a23[2], a23[3], b1);
    if (MORE) { load_wfrag<16>(a23, woff, wbase); wbase += 2048; }
    __builtin_amdgcn_sched_barrier(0);
    // x_{i+1} has landed (older than the two weight loads), every LDS read of this chunk has returned
    if (MORE) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i + 1 < nchunks; ++i) chunk_body(i, std::true_type{});
  chunk_body(nchunks - 1, std::false_type{});

  // ---- epilogue: a lane holds 4 consecutive output channels of one pixel per register quad
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int64_t n = n0 + ni * 32 + li;
    if (n >= npix) continue;
    float* orow = out + n * Cout;
    const float* rrow = residual ? residual + n * Cout : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + wave * 32 + 8 * q + 4 * kh;
      if (m >= Cout) continue;
      float4 v = make_float4(acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]);
      if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + m);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (rrow) {
        const float4 rv = *reinterpret_cast<const float4*>(rrow + m);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      *reinterpret_cast<float4*>(orow + m) = v;
    }
  }
}

// detectron2 ROIAlign (layers/csrc/ROIAlignRotated is NOT this one; this is ROIAlign/ROIAlign_cpu.cpp's
// arithmetic, restated in oracle/roi_head_oracle.py): one workgroup per (roi, ph), threads over
// (pw, 4-channel group).
__device__ __forceinline__ void bilinear_setup(float y, float x, int H, int W, int& yl, int& yh, int& xl, int& xh,
                                               float& w1, float& w2, float& w3, float& w4, bool& empty) {
  empty = y < -1.0f || y > (float)H || x < -1.0f || x > (float)W;
  y = fmaxf(y, 0.f);
  x = fmaxf(x, 0.f);
  yl = (int)y;
  xl = (int)x;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// four channels of a map pixel as fp32 (bf16 values are exact in fp32)
template <bool BF16IN>
__device__ __forceinline__ float4 load4(const void* base, int64_t idx) {
  if constexpr (BF16IN) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(base) + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  } else {
    return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
  }
}

template <bool BF16IN, bool BF16OUT>
__global__ __launch_bounds__(256) void roi_align_nhwc_kernel(
    const void* __restrict__ feat, int NF, int H, int W, int C, const float* __restrict__ rois, int64_t R,
    int P, float scale, int sampling_ratio, int aligned, int bs, int OP, void* __restrict__ out_v) {
  // bins (ph, pw) = (bs * oph, bs * opw) only: OP = ceil(P / bs) rows and columns of the P x P grid are produced
  const int64_t item = blockIdx.x;                 // (roi, oph)
  const int64_t r = item / OP;
  const int oph = (int)(item - r * OP);
  const int ph = oph * bs;
  const float* roi = rois + r * 5;
  int bi = (int)roi[0];
  bi = bi < 0 ? 0 : (bi >= NF ? NF - 1 : bi);
  const float off = aligned ? 0.5f : 0.f;
  const float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
  float rw = roi[3] * scale - off - sw, rh = roi[4] * scale - off - sh;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  const float bin_h = rh / (float)P, bin_w = rw / (float)P;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
  const float count = (float)max(gh * gw, 1);
  const int64_t fb = (int64_t)bi * H * W * C;
  const int c4 = C >> 2;
  for (int idx = threadIdx.x; idx < OP * c4; idx += blockDim.x) {
    const int opw = idx / c4, cg = idx - opw * c4;
    const int pw = opw * bs;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = sw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
        int yl, yh, xl, xh;
        float w1, w2, w3, w4;
        bool empty;
        bilinear_setup(y, xx, H, W, yl, yh, xl, xh, w1, w2, w3, w4, empty);
        if (empty) continue;
        const float4 v1 = load4<BF16IN>(feat, fb + ((int64_t)yl * W + xl) * C + 4 * cg);
        const float4 v2 = load4<BF16IN>(feat, fb + ((int64_t)yl * W + xh) * C + 4 * cg);
        const float4 v3 = load4<BF16IN>(feat, fb + ((int64_t)yh * W + xl) * C + 4 * cg);
        const float4 v4 = load4<BF16IN>(feat, fb + ((int64_t)yh * W + xh) * C + 4 * cg);
        acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
        acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
        acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
        acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
      }
    }
    acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
    const int64_t o = ((r * OP + oph) * OP + opw) * (int64_t)C + 4 * cg;
    if constexpr (BF16OUT) {     // the fp32 result rounded once to bf16 (operand of the bf16 conv kernels)
      const bf16x4 ob = {(__bf16)acc.x, (__bf16)acc.y, (__bf16)acc.z, (__bf16)acc.w};
      *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(out_v) + o) = ob;
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(out_v) + o) = acc;
    }
  }
}

// max_pool2d(kernel k, stride s, padding p) on a channels-last fp32 map (detectron2 BasicStem: 3 / 2 / 1);
// padding positions do not take part (-inf).  A thread = 4 channels of one output pixel.
template <bool BF16OUT>
__global__ __launch_bounds__(256) void max_pool_nhwc_kernel(const float* __restrict__ x, int64_t NB, int H, int W,
                                                            int C, int k, int stride, int pad, int OH, int OW,
                                                            void* __restrict__ out_v) {
  const int c4 = C >> 2;
  const int64_t total = NB * OH * OW * c4;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(o % c4);
    const int64_t pix = o / c4;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int64_t nb = pix / ((int64_t)OW * OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int a = 0; a < k; ++a) {
      const int ih = oh * stride - pad + a;
      if (ih < 0 || ih >= H) continue;
      for (int b = 0; b < k; ++b) {
        const int iw = ow * stride - pad + b;
        if (iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((nb * H + ih) * (int64_t)W + iw) * C + 4 * cg);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    const int64_t oo = pix * C + 4 * cg;
    if constexpr (BF16OUT) {
      const bf16x4 ob = {(__bf16)m.x, (__bf16)m.y, (__bf16)m.z, (__bf16)m.w};
      *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(out_v) + oo) = ob;
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(out_v) + oo) = m;
    }
  }
}

}  // namespace

extern "C" int tspn_pack_conv2d_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                                    float* packed, void* stream) {
  TSPN_REQUIRE(w && packed, TSPN_EINVAL, "tspn_pack_conv2d_f32: null pointer");
  TSPN_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && KH * KW <= 64, TSPN_EINVAL,
               "tspn_pack_conv2d_f32: bad sizes");
  const int64_t total = KH * KW * Cin * Cout;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv2d_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), w, Cout, Cin, KH, KW,
                     packed);
  return tspn::check_launch("tspn_pack_conv2d_f32");
}

extern "C" int tspn_conv2d_nhwc_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                                    const float* packed, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                                    int64_t pad, const float* bias, const float* residual, int relu, float* out,
                                    void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               TSPN_EINVAL, "tspn_conv2d_nhwc_f32: bad sizes");
  TSPN_REQUIRE(KH * KW <= 64, TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_f32: at most 64 taps");
  const int64_t OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_conv2d_nhwc_f32: empty output (H=%lld W=%lld)", (long long)H,
               (long long)W);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && packed && out, TSPN_EINVAL, "tspn_conv2d_nhwc_f32: null pointer");
  TSPN_REQUIRE(Cin % 16 == 0 && Cout % 4 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_f32: needs Cin %% 16 == 0 and Cout %% 4 == 0 (Cin=%lld Cout=%lld)", (long long)Cin,
               (long long)Cout);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(packed) && al16(out) && (!bias || al16(bias)) && (!residual || al16(residual)),
               TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_f32: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && Cin < (1 << 24) && Cout < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_f32: dimension too large");
  const int64_t npix = NB * OH * OW;
  const int64_t tiles_m = tspn::ceil_div(Cout, BM), tiles_n = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_f32: grid too large");
  // (32-channel chunks were measured slower: 92 vs 102 TFLOP/s on the res5 shapes; 16 ships)
  hipLaunchKernelGGL(conv2d_nhwc_kernel<16>, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS),
                     smem_bytes<16>(), TSPN_STREAM(stream), x, packed, bias, residual, out, (int)H, (int)W,
                     (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)OH, (int)OW, npix,
                     (int)tiles_m, (int)tiles_n, relu);
  return tspn::check_launch("tspn_conv2d_nhwc_f32");
}

extern "C" int tspn_pack_conv2d_frag_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                                         float* packed, void* stream) {
  TSPN_REQUIRE(w && packed, TSPN_EINVAL, "tspn_pack_conv2d_frag_f32: null pointer");
  TSPN_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && KH * KW <= 64, TSPN_EINVAL,
               "tspn_pack_conv2d_frag_f32: bad sizes");
  TSPN_REQUIRE(Cout % 32 == 0 && Cin % 16 == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv2d_frag_f32: needs Cout %% 32 == 0 and Cin %% 16 == 0 (Cout=%lld Cin=%lld)",
               (long long)Cout, (long long)Cin);
  const int64_t total = KH * KW * Cin * Cout;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv2d_frag_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), w, Cout, Cin,
                     KH * KW, packed);
  return tspn::check_launch("tspn_pack_conv2d_frag_f32");
}

extern "C" int tspn_conv2d_nhwc_frag_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                                         const float* frag, int64_t Cout, int64_t KH, int64_t KW,
                                         int64_t stride, int64_t pad, const float* bias, const float* residual,
                                         int relu, float* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               TSPN_EINVAL, "tspn_conv2d_nhwc_frag_f32: bad sizes");
  TSPN_REQUIRE(KH * KW <= 64, TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_frag_f32: at most 64 taps");
  const int64_t OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_conv2d_nhwc_frag_f32: empty output");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && out, TSPN_EINVAL, "tspn_conv2d_nhwc_frag_f32: null pointer");
  TSPN_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_frag_f32: needs Cin %% 16 == 0 and Cout %% 32 == 0 (Cin=%lld Cout=%lld)",
               (long long)Cin, (long long)Cout);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag) && al16(out) && (!bias || al16(bias)) && (!residual || al16(residual)),
               TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_frag_f32: operands must be 16-byte aligned");
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && Cin < (1 << 24) && Cout < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_frag_f32: dimension too large");
  const int64_t npix = NB * OH * OW;
  const int64_t tiles_m = tspn::ceil_div(Cout, BM), tiles_n = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv2d_nhwc_frag_f32: grid too large");
  hipLaunchKernelGGL(conv2d_nhwc_frag_kernel<false>, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), 0,
                     TSPN_STREAM(stream), x, frag, bias, residual, out, (int)H, (int)W, (int)Cin, (int)Cout,
                     (int)KH, (int)KW, (int)stride, (int)pad, (int)OH, (int)OW, npix, (int)tiles_m, (int)tiles_n,
                     relu);
  return tspn::check_launch("tspn_conv2d_nhwc_frag_f32");
}

extern "C" int tspn_pack_conv2d_frag_cin4_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                                              float* packed, void* stream) {
  TSPN_REQUIRE(w && packed, TSPN_EINVAL, "tspn_pack_conv2d_frag_cin4_f32: null pointer");
  TSPN_REQUIRE(Cout > 0 && Cin > 0 && Cin <= 4 && KH > 0 && KW > 0 && KH * KW <= 64 && Cout % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_pack_conv2d_frag_cin4_f32: needs Cin <= 4, Cout %% 32 == 0, at most 64 taps");
  const int64_t total = (Cout / 32) * ((KH * KW + 3) / 4) * 512;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv2d_frag_cin4_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), w, Cout, Cin,
                     KH * KW, packed);
  return tspn::check_launch("tspn_pack_conv2d_frag_cin4_f32");
}

extern "C" int tspn_conv2d_nhwc_cin4_f32(const float* x, int64_t NB, int64_t H, int64_t W, const float* frag,
                                         int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                                         const float* bias, int relu, float* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, TSPN_EINVAL,
               "tspn_conv2d_nhwc_cin4_f32: bad sizes");
  TSPN_REQUIRE(KH * KW <= 64 && Cout % 32 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_cin4_f32: needs at most 64 taps and Cout %% 32 == 0");
  const int64_t OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_conv2d_nhwc_cin4_f32: empty output");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && out, TSPN_EINVAL, "tspn_conv2d_nhwc_cin4_f32: null pointer");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag) && al16(out) && (!bias || al16(bias)), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_cin4_f32: operands must be 16-byte aligned");
  const int64_t npix = NB * OH * OW;
  const int64_t tiles_m = tspn::ceil_div(Cout, BM), tiles_n = tspn::ceil_div(npix, BN);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31) && H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED,
               "tspn_conv2d_nhwc_cin4_f32: problem too large");
  hipLaunchKernelGGL(conv2d_nhwc_frag_kernel<true>, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), 0,
                     TSPN_STREAM(stream), x, frag, bias, static_cast<const float*>(nullptr), out, (int)H, (int)W, 4,
                     (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)OH, (int)OW, npix, (int)tiles_m,
                     (int)tiles_n, relu);
  return tspn::check_launch("tspn_conv2d_nhwc_cin4_f32");
}

static int roi_align_launch(const void* feat, bool bf16_in, int64_t NF, int64_t H, int64_t W, int64_t C,
                            const float* rois, int64_t R, int64_t P, float spatial_scale, int sampling_ratio,
                            int aligned, int bin_stride, void* out, bool bf16_out, void* stream) {
  TSPN_REQUIRE(NF > 0 && H > 0 && W > 0 && C > 0 && R >= 0 && P > 0 && sampling_ratio >= 0 && bin_stride >= 1,
               TSPN_EINVAL, "tspn_roi_align_nhwc_f32: bad sizes");
  if (R == 0) return TSPN_OK;
  TSPN_REQUIRE(feat && rois && out, TSPN_EINVAL, "tspn_roi_align_nhwc_f32: null pointer");
  TSPN_REQUIRE(C % 4 == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               TSPN_EUNSUPPORTED, "tspn_roi_align_nhwc_f32: needs C %% 4 == 0 and 16-byte aligned tensors");
  TSPN_REQUIRE(R * P < (1LL << 31) && H < (1 << 20) && W < (1 << 20), TSPN_EUNSUPPORTED,
               "tspn_roi_align_nhwc_f32: problem too large");
  const int64_t OP = tspn::ceil_div(P, bin_stride);
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(R * OP)), dim3(256), 0, TSPN_STREAM(stream), feat, (int)NF, (int)H,
                       (int)W, (int)C, rois, R, (int)P, spatial_scale, sampling_ratio, aligned, (int)bin_stride, (int)OP, out);
  };

  if (bf16_in) launch(roi_align_nhwc_kernel<true, true>);
  else if (bf16_out) launch(roi_align_nhwc_kernel<false, true>);
  else launch(roi_align_nhwc_kernel<false, false>);
  return tspn::check_launch("tspn_roi_align_nhwc");
}

extern "C" int tspn_roi_align_nhwc_f32(const float* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                                       const float* rois, int64_t R, int64_t P, float spatial_scale,
                                       int sampling_ratio, int aligned, int bin_stride, float* out, void* stream) {
  return roi_align_launch(feat, false, NF, H, W, C, rois, R, P, spatial_scale, sampling_ratio, aligned, bin_stride, out, false,
                          stream);
}

extern "C" int tspn_roi_align_nhwc_f32_bf16out(const float* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                                               const float* rois, int64_t R, int64_t P, float spatial_scale,
                                               int sampling_ratio, int aligned, int bin_stride, uint16_t* out, void* stream) {
  return roi_align_launch(feat, false, NF, H, W, C, rois, R, P, spatial_scale, sampling_ratio, aligned, bin_stride, out, true,
                          stream);
}

extern "C" int tspn_roi_align_nhwc_bf16(const uint16_t* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                                        const float* rois, int64_t R, int64_t P, float spatial_scale,
                                        int sampling_ratio, int aligned, int bin_stride, uint16_t* out, void* stream) {
  return roi_align_launch(feat, true, NF, H, W, C, rois, R, P, spatial_scale, sampling_ratio, aligned, bin_stride, out, true,
                          stream);
}

extern "C" int tspn_max_pool_nhwc_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t C, int64_t k,
                                      int64_t stride, int64_t pad, void* out, int out_bf16, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k, TSPN_EINVAL,
               "tspn_max_pool_nhwc_f32: bad sizes");
  const int64_t OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_max_pool_nhwc_f32: empty output");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_max_pool_nhwc_f32: null pointer");
  TSPN_REQUIRE(C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               TSPN_EUNSUPPORTED, "tspn_max_pool_nhwc_f32: needs C %% 4 == 0 and 16-byte aligned tensors");
  const int64_t total = NB * OH * OW * (C / 4);
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 1 << 20);
  if (out_bf16)
    hipLaunchKernelGGL(max_pool_nhwc_kernel<true>, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), x, NB, (int)H,
                       (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)OH, (int)OW, out);
  else
    hipLaunchKernelGGL(max_pool_nhwc_kernel<false>, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), x, NB, (int)H,
                       (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)OH, (int)OW, out);
  return tspn::check_launch("tspn_max_pool_nhwc_f32");
}
