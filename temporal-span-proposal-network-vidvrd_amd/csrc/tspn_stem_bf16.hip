// bf16-operand stem of the C4 backbone (detectron2 BasicStem = conv 7x7 / stride 2 / pad 3 on RGB +
// FrozenBN + ReLU, then max_pool2d(3, 2, 1)) for the bf16 backbone of BASELINE cfg5.
//
// Round 2 ran the stem in fp32 on the generic 128-row tile: with 64 output channels half of its MFMAs were
// discarded, the map left as 944 MB of fp32 per 16 frames of 720p and the pool read it back -- 2.1 ms of a
// 10 ms backbone.  Here (semantics: like every other conv of the bf16 backbone -- image and folded weights are bf16
// VALUES, products exact, fp32 accumulation, act(acc + bias) rounded to bf16 once; oracle/roi_head_oracle.py:
// resnet_c4_bf16):
//   1. stem_s2d_bf16_kernel: 2 x 2 space-to-depth of the zero-padded image, fp32 -> bf16:
//        S[nb][r][c][(ra, rb, ch)] = x[nb][2 r + ra - 3][2 c + rb - 3][ch]   (12 values + 4 zeros = one 32-byte pixel)
//      so the 7 x 7 / 2 conv becomes a VALID 4 x 4 / 1 conv with 16 channels: K = 16 taps x 16 = 256, no masks.
//   2. stem_conv_bf16_kernel: persistent workgroups, a tile = 128 consecutive output pixels of one row.  Its
//      whole operand -- 4 rows x 131 s2d pixels = 16.8 KB, CONTIGUOUS per row -- is staged by LDS-DMA one tile ahead
//      (double buffer); every (tap, channel half) fragment is a conflict-free 16-byte LDS read at a pixel offset.
//      The 64 x 256 weights live in 128 registers per lane for the whole kernel (fragment-major, loaded once per
//      workgroup); wave w = all 64 rows x pixels [32 w, 32 w + 32), 32 MFMAs (32x32x16 bf16) per tile.  Epilogue:
//      + bias, ReLU, transposed through LDS so that a lane stores 8 consecutive channels (16 bytes) of a pixel.
//   3. max_pool_nhwc_bf16_kernel: the 3 x 3 / 2 max pool on the bf16 map (max commutes with the rounding, so
//      "round, then pool" equals the oracle's "pool, then round").
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KS = 7, KPAD = 3;           // the stem's kernel size and padding (stride 2)
constexpr int TPX = 128;                  // output pixels per tile
constexpr int SROW = 132;                 // staged s2d pixels per row (TPX + 3 used)
constexpr int STAGE_BYTES = 4 * SROW * 32;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// x fp32 [NB][H][W][3] -> S bf16 [NB][SH][SW][16]; one thread = one s2d pixel (two 16-byte stores)
__global__ __launch_bounds__(256) void stem_s2d_bf16_kernel(const float* __restrict__ x, int64_t NB, int H, int W,
                                                            int SH, int SW, __bf16* __restrict__ S) {
  const int64_t total = NB * SH * SW;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(o % SW), r = (int)((o / SW) % SH);
    const int64_t nb = o / ((int64_t)SW * SH);
    bf16x8 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (__bf16)0.f; hi[j] = (__bf16)0.f; }
#pragma unroll
    for (int ra = 0; ra < 2; ++ra)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int ih = 2 * r + ra - KPAD, iw = 2 * c + rb - KPAD;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        const float* p = x + ((nb * H + ih) * (int64_t)W + iw) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const int k = (ra * 2 + rb) * 3 + ch;
          const __bf16 v = (__bf16)p[ch];
          if (k < 8) lo[k] = v; else hi[k - 8] = v;
        }
      }
    bf16x8* dst = reinterpret_cast<bf16x8*>(S + o * 16);
    dst[0] = lo;
    dst[1] = hi;
  }
}

// w fp32 [Cout][3][7][7] (batch norm already folded) -> bf16 fragment-major [Cout/32][16 k-steps = (a', b')][64 lanes][8]:
// lane = 32 kh + li holds row 32 mb + li, s2d channels 8 kh .. 8 kh + 7 of tap (a', b')
__global__ void pack_stem_bf16_kernel(const float* __restrict__ w, int Cout, __bf16* __restrict__ frag) {
  const int total = Cout * 256;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
    const int j = o & 7, lane = (o >> 3) & 63, ks = (o >> 9) & 15, mb = o >> 13;
    const int row = 32 * mb + (lane & 31), c16 = 8 * (lane >> 5) + j;
    const int a = 2 * (ks >> 2) + c16 / 6, b = 2 * (ks & 3) + (c16 % 6) / 3, ch = c16 % 3;
    const bool ok = c16 < 12 && a < KS && b < KS;
    frag[o] = (__bf16)(ok ? w[((row * 3 + ch) * KS + a) * KS + b] : 0.f);
  }
}

template <int MB>
__global__ __launch_bounds__(256, 2) void stem_conv_bf16_kernel(const __bf16* __restrict__ S, const __bf16* __restrict__ Wf,
                                                                const float* __restrict__ bias, __bf16* __restrict__ out,
                                                                int SH, int SW, int OH, int OW, int tiles_w,
                                                                int64_t ntiles) {
  constexpr int COUT = 32 * MB;
  constexpr int PITCH = COUT + 4;            // floats per pixel row of the epilogue transpose
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages + the epilogue transpose: 68.6 KB for MB = 2
  char* stage = smem;
  float* tr = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;

  // weights: 16 k-steps x MB fragments, resident in registers for the whole kernel
  f32x4 a[16][MB];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
      a[ks][mi] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wf) + ((mi * 16 + ks) * 64 + lane) * 16);
  float4 bv[MB][4];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[mi][q] = *reinterpret_cast<const float4*>(bias + mi * 32 + 8 * q + 4 * kh);

  // wave r stages row r of a tile: 2 * (TPX + 3) 16-byte pieces, contiguous in global memory
  auto stage_tile = [&](int64_t tile, int buf) {
    const int tw = (int)(tile % tiles_w);
    const int64_t rowi = tile / tiles_w;                 // nb * OH + oh
    const int oh = (int)(rowi % OH);
    const int64_t nb = rowi / OH;
    const int ow0 = tw * TPX;
    const int npiece = 2 * min(TPX + 3, SW - ow0);       // never past the end of the s2d row
    const char* src = reinterpret_cast<const char*>(S) + (((nb * SH + oh + wave) * (int64_t)SW + ow0) * 32);
    char* dst = stage + buf * STAGE_BYTES + wave * (SROW * 32);
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int p = it * 64 + lane;
      if (p < npiece) glds16(src + p * 16, dst + it * 1024);   // the builtin adds lane * 16 to the LDS address
    }
  };

  int64_t tile = blockIdx.x;
  if (tile >= ntiles) return;
  stage_tile(tile, 0);
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's row of the tile has landed (and its last stores left)
    __syncthreads();
    const int64_t next = tile + gridDim.x;
    if (next < ntiles) stage_tile(next, buf ^ 1);

    f32x16 acc[MB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
    const char* Bb = stage + buf * STAGE_BYTES + (32 * wave + li) * 32 + 16 * kh;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bb + ((ks >> 2) * SROW + (ks & 3)) * 32);
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][mi]), b, acc[mi], 0, 0, 0);
    }

    // epilogue: (acc + bias) through the wave's own LDS slab, then 8 consecutive channels per lane
    const int tw = (int)(tile % tiles_w);
    const int64_t rowi = tile / tiles_w;
    const int ow0 = tw * TPX + 32 * wave;
    float* t = tr + wave * (32 * PITCH);
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = make_float4(acc[mi][4 * q] + bv[mi][q].x, acc[mi][4 * q + 1] + bv[mi][q].y,
                                     acc[mi][4 * q + 2] + bv[mi][q].z, acc[mi][4 * q + 3] + bv[mi][q].w);
        *reinterpret_cast<float4*>(t + li * PITCH + mi * 32 + 8 * q + 4 * kh) = v;
      }
    constexpr int CPL = COUT / 8;               // lanes per pixel
    constexpr int PPP = 64 / CPL;               // pixels per pass
    const int pl = lane / CPL, cg = lane - pl * CPL;
#pragma unroll
    for (int pass = 0; pass < 32 / PPP; ++pass) {
      const int px = pass * PPP + pl;
      const float4 v0 = *reinterpret_cast<const float4*>(t + px * PITCH + 8 * cg);
      const float4 v1 = *reinterpret_cast<const float4*>(t + px * PITCH + 8 * cg + 4);
      if (ow0 + px < OW) {
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)fmaxf(v[k], 0.f);
        *reinterpret_cast<bf16x8*>(out + (rowi * OW + ow0 + px) * COUT + 8 * cg) = o;
      }
    }
  }
}

// The stem conv AND its 3x3 / stride 2 / pad 1 max pool in one kernel: the conv map (236 MB per 8 frames of 720p) is
// never written, and the pool's 1.5x re-read of it disappears.  Work item = one POOLED row (nb, ph); a persistent
// workgroup walks it left to right in tiles of 64 pooled pixels = conv rows 2 ph - 1 .. 2 ph + 1 x 128 conv columns:
//   * operand = 6 rows x 131 s2d pixels (25.3 KB) by LDS-DMA one tile ahead, double-buffered;
//   * the three conv rows run one after the other through the same 16-k-step MFMA loop as above and are reduced to
//     their elementwise maximum in the accumulator layout (rows outside the map are skipped: wave-uniform);
//   * relu(max + bias) is rounded to bf16 and written as [conv column][channel] into an LDS slab of 1 + 128 columns;
//     slot 0 holds conv column -1 of the tile = column 127 of the previous tile of the row (copied after use), zero
//     at the start of a row;
//   * a lane takes the maximum of columns 2 i - 1, 2 i, 2 i + 1 for 8 channels and stores 16 bytes: a pooled pixel
//     leaves as one 128-byte line.
// x + bias, ReLU and the rounding are monotonic, so "max first, then bias / ReLU / round" gives bit for bit
// pool(round(relu(conv + bias))).  Every odd conv row is computed twice (1.5x the MFMAs of the stem, which are 2 % of
// the backbone's); 69.3 KB of LDS, two workgroups per CU.
constexpr int PROWS = 6;
constexpr int PSTAGE_BYTES = PROWS * SROW * 32;

template <int MB>
__global__ __launch_bounds__(256, 2) void stem_pool_bf16_kernel(const __bf16* __restrict__ S, const __bf16* __restrict__ Wf,
                                                                const float* __restrict__ bias, __bf16* __restrict__ out,
                                                                int SH, int SW, int OH, int OW, int PH, int PW,
                                                                int tiles_w, int64_t nrows) {
  constexpr int COUT = 32 * MB;
  constexpr int PITCHB = COUT * 2 + 16;       // bytes per conv column of the slab
  constexpr int CPL = COUT / 8;               // lanes per pooled pixel (16 bytes each)
  constexpr int PPP = 64 / CPL;               // pooled pixels per pass
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages + the slab
  char* stage = smem;
  char* slab = smem + 2 * PSTAGE_BYTES;       // (1 + TPX) columns

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;

  f32x4 a[16][MB];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
      a[ks][mi] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wf) + ((mi * 16 + ks) * 64 + lane) * 16);

  // item it of this workgroup: pooled row blockIdx.x + (it / tiles_w) * gridDim.x, tile it % tiles_w
  const int64_t rows_mine = blockIdx.x < nrows ? (nrows - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t nitems = rows_mine * tiles_w;
  // wave w stages s2d rows w and w + 4 of the six
  auto stage_item = [&](int64_t it, int buf) {
    const int tw = (int)(it % tiles_w);
    const int64_t row = blockIdx.x + (it / tiles_w) * gridDim.x;
    const int ph = (int)(row % PH);
    const int64_t nb = row / PH;
    const int c0 = tw * TPX;
    const int npiece = 2 * min(TPX + 3, SW - c0);
    for (int j = wave; j < PROWS; j += 4) {
      const int sr = 2 * ph - 1 + j;
      if (sr < 0 || sr >= SH) continue;                  // rows of conv rows outside the map: never read
      const char* src = reinterpret_cast<const char*>(S) + (((nb * SH + sr) * (int64_t)SW + c0) * 32);
      char* dst = stage + buf * PSTAGE_BYTES + j * (SROW * 32);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int p = k * 64 + lane;
        if (p < npiece) glds16(src + p * 16, dst + k * 1024);
      }
    }
  };

  if (nitems == 0) return;
  stage_item(0, 0);
  int buf = 0;
  for (int64_t it = 0; it < nitems; ++it, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows of the item have landed
    __syncthreads();                                     // ... everybody's; the slab of the previous item is free
    if (it + 1 < nitems) stage_item(it + 1, buf ^ 1);
    const int tw = (int)(it % tiles_w);
    const int64_t row = blockIdx.x + (it / tiles_w) * gridDim.x;
    const int ph = (int)(row % PH);

    f32x16 vmax[MB];
    bool have = false;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int r = 2 * ph - 1 + rr;
      if (r < 0 || r >= OH) continue;
      f32x16 acc[MB];
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
      const char* Bb = stage + buf * PSTAGE_BYTES + (rr * SROW + 32 * wave + li) * 32 + 16 * kh;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bb + ((ks >> 2) * SROW + (ks & 3)) * 32);
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][mi]), b, acc[mi], 0, 0, 0);
      }
      if (!have) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) vmax[mi] = acc[mi];
        have = true;
      } else {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
          for (int e = 0; e < 16; ++e) vmax[mi][e] = fmaxf(vmax[mi][e], acc[mi][e]);
      }
    }

    // relu(max + bias) -> bf16 -> slab[1 + conv column][channel]; columns beyond the map are zero (below every ReLU output)
    const bool colok = tw * TPX + 32 * wave + li < OW;
    char* sp = slab + (1 + 32 * wave + li) * PITCHB;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = mi * 32 + 8 * q + 4 * kh;
        const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
        bf16x4 v;
        v[0] = (__bf16)(colok ? fmaxf(vmax[mi][4 * q] + bv.x, 0.f) : 0.f);
        v[1] = (__bf16)(colok ? fmaxf(vmax[mi][4 * q + 1] + bv.y, 0.f) : 0.f);
        v[2] = (__bf16)(colok ? fmaxf(vmax[mi][4 * q + 2] + bv.z, 0.f) : 0.f);
        v[3] = (__bf16)(colok ? fmaxf(vmax[mi][4 * q + 3] + bv.w, 0.f) : 0.f);
        *reinterpret_cast<bf16x4*>(sp + ch * 2) = v;
      }
    __syncthreads();

    // pooled pixel 16 wave + i of the tile = conv columns 32 wave + 2 i - 1 .. + 1 = slab slots 32 wave + 2 i .. + 2
    const int pl = lane / CPL, cg = lane - pl * CPL;
#pragma unroll
    for (int pass = 0; pass < 16 / PPP; ++pass) {
      const int i = pass * PPP + pl;
      const char* cp = slab + (32 * wave + 2 * i) * PITCHB + cg * 16;
      bf16x8 l = *reinterpret_cast<const bf16x8*>(cp);
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(cp + PITCHB);
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(cp + 2 * PITCHB);
      if (tw == 0 && wave == 0 && i == 0) {              // conv column -1 of the row: padding
#pragma unroll
        for (int k = 0; k < 8; ++k) l[k] = (__bf16)0.f;
      }
      const int pw = tw * (TPX / 2) + 16 * wave + i;
      if (pw < PW) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)fmaxf(fmaxf((float)l[k], (float)m[k]), (float)r[k]);
        *reinterpret_cast<bf16x8*>(out + (row * PW + pw) * COUT + 8 * cg) = o;
      }
    }
    __syncthreads();
    if (tid < CPL)                                       // column 127 becomes column -1 of the next tile of the row
      *reinterpret_cast<bf16x8*>(slab + tid * 16) = *reinterpret_cast<const bf16x8*>(slab + TPX * PITCHB + tid * 16);
  }
}

// max_pool2d(k, stride, pad) on a channels-last bf16 map; a thread = 8 channels (16 bytes) of one output pixel
__global__ __launch_bounds__(256) void max_pool_nhwc_bf16_kernel(const __bf16* __restrict__ x, int64_t NB, int H, int W,
                                                                 int C, int k, int stride, int pad, int OH, int OW,
                                                                 __bf16* __restrict__ out) {
  const int c8 = C >> 3;
  const int64_t total = NB * OH * OW * c8;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(o % c8);
    const int64_t pix = o / c8;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int64_t nb = pix / ((int64_t)OW * OH);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int a = 0; a < k; ++a) {
      const int ih = oh * stride - pad + a;
      if (ih < 0 || ih >= H) continue;
      for (int b = 0; b < k; ++b) {
        const int iw = ow * stride - pad + b;
        if (iw < 0 || iw >= W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((nb * H + ih) * (int64_t)W + iw) * C + 8 * cg);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    bf16x8 ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) ob[j] = (__bf16)m[j];
    *reinterpret_cast<bf16x8*>(out + pix * C + 8 * cg) = ob;
  }
}

inline int64_t out_dim(int64_t n) { return (n + 2 * KPAD - KS) / 2 + 1; }

}  // namespace

extern "C" size_t tspn_stem_bf16_workspace_bytes(int64_t NB, int64_t H, int64_t W) {
  if (NB <= 0 || H <= 0 || W <= 0) return 0;
  return tspn::align_up((size_t)NB * (size_t)(out_dim(H) + 3) * (size_t)(out_dim(W) + 3) * 32, 256);
}

extern "C" int tspn_pack_stem_bf16(const float* w, int64_t Cout, uint16_t* frag, void* stream) {
  TSPN_REQUIRE(w && frag, TSPN_EINVAL, "tspn_pack_stem_bf16: null pointer");
  TSPN_REQUIRE(Cout == 32 || Cout == 64, TSPN_EUNSUPPORTED, "tspn_pack_stem_bf16: Cout must be 32 or 64 (got %lld)",
               (long long)Cout);
  hipLaunchKernelGGL(pack_stem_bf16_kernel, dim3((unsigned)Cout), dim3(256), 0, TSPN_STREAM(stream), w, (int)Cout,
                     reinterpret_cast<__bf16*>(frag));
  return tspn::check_launch("tspn_pack_stem_bf16");
}

extern "C" int tspn_stem_conv_bf16(const float* x, int64_t NB, int64_t H, int64_t W, const uint16_t* frag, int64_t Cout,
                                   const float* bias, void* workspace, size_t workspace_bytes, uint16_t* out,
                                   void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "tspn_stem_conv_bf16: bad sizes");
  TSPN_REQUIRE(Cout == 32 || Cout == 64, TSPN_EUNSUPPORTED, "tspn_stem_conv_bf16: Cout must be 32 or 64 (got %lld)",
               (long long)Cout);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && bias && workspace && out, TSPN_EINVAL, "tspn_stem_conv_bf16: null pointer");
  const int64_t OH = out_dim(H), OW = out_dim(W), SH = OH + 3, SW = OW + 3;
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && NB * SH * SW < (1LL << 40), TSPN_EUNSUPPORTED,
               "tspn_stem_conv_bf16: image too large");
  const size_t need = tspn_stem_bf16_workspace_bytes(NB, H, W);
  TSPN_REQUIRE(workspace_bytes >= need, TSPN_EWORKSPACE, "tspn_stem_conv_bf16: workspace %zu < %zu bytes", workspace_bytes,
               need);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(frag) && al16(bias) && al16(workspace) && al16(out), TSPN_EUNSUPPORTED,
               "tspn_stem_conv_bf16: operands must be 16-byte aligned");
  __bf16* S = static_cast<__bf16*>(workspace);
  hipStream_t s = TSPN_STREAM(stream);
  const int64_t npx = NB * SH * SW;
  hipLaunchKernelGGL(stem_s2d_bf16_kernel, dim3((unsigned)std::min<int64_t>(tspn::ceil_div(npx, 256), 1 << 20)), dim3(256),
                     0, s, x, NB, (int)H, (int)W, (int)SH, (int)SW, S);
  if (int rc = tspn::check_launch("tspn_stem_conv_bf16 (space-to-depth)")) return rc;
  const int64_t tiles_w = tspn::ceil_div(OW, TPX), ntiles = NB * OH * tiles_w;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const unsigned grid = (unsigned)std::min<int64_t>(ntiles, 2LL * cus);
  const __bf16* Wf = reinterpret_cast<const __bf16*>(frag);
  __bf16* o = reinterpret_cast<__bf16*>(out);
  const size_t smem = 2 * STAGE_BYTES + 4 * 32 * ((size_t)Cout + 4) * sizeof(float);
  if (Cout == 64) {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(stem_conv_bf16_kernel<2>), smem, "tspn_stem_conv_bf16")) return rc;
    hipLaunchKernelGGL(stem_conv_bf16_kernel<2>, dim3(grid), dim3(256), smem, s, S, Wf, bias, o, (int)SH, (int)SW, (int)OH,
                       (int)OW, (int)tiles_w, ntiles);
  } else {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(stem_conv_bf16_kernel<1>), smem, "tspn_stem_conv_bf16")) return rc;
    hipLaunchKernelGGL(stem_conv_bf16_kernel<1>, dim3(grid), dim3(256), smem, s, S, Wf, bias, o, (int)SH, (int)SW, (int)OH,
                       (int)OW, (int)tiles_w, ntiles);
  }
  return tspn::check_launch("tspn_stem_conv_bf16");
}

extern "C" int tspn_stem_pool_bf16(const float* x, int64_t NB, int64_t H, int64_t W, const uint16_t* frag, int64_t Cout,
                                   const float* bias, void* workspace, size_t workspace_bytes, uint16_t* out,
                                   void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "tspn_stem_pool_bf16: bad sizes");
  TSPN_REQUIRE(Cout == 32 || Cout == 64, TSPN_EUNSUPPORTED, "tspn_stem_pool_bf16: Cout must be 32 or 64 (got %lld)",
               (long long)Cout);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag && bias && workspace && out, TSPN_EINVAL, "tspn_stem_pool_bf16: null pointer");
  const int64_t OH = out_dim(H), OW = out_dim(W), SH = OH + 3, SW = OW + 3;
  const int64_t PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;     // max_pool2d(3, 2, 1)
  TSPN_REQUIRE(H < (1 << 20) && W < (1 << 20) && NB * SH * SW < (1LL << 40), TSPN_EUNSUPPORTED,
               "tspn_stem_pool_bf16: image too large");
  const size_t need = tspn_stem_bf16_workspace_bytes(NB, H, W);
  TSPN_REQUIRE(workspace_bytes >= need, TSPN_EWORKSPACE, "tspn_stem_pool_bf16: workspace %zu < %zu bytes", workspace_bytes,
               need);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(frag) && al16(bias) && al16(workspace) && al16(out), TSPN_EUNSUPPORTED,
               "tspn_stem_pool_bf16: operands must be 16-byte aligned");
  __bf16* S = static_cast<__bf16*>(workspace);
  hipStream_t s = TSPN_STREAM(stream);
  const int64_t npx = NB * SH * SW;
  hipLaunchKernelGGL(stem_s2d_bf16_kernel, dim3((unsigned)std::min<int64_t>(tspn::ceil_div(npx, 256), 1 << 20)), dim3(256),
                     0, s, x, NB, (int)H, (int)W, (int)SH, (int)SW, S);
  if (int rc = tspn::check_launch("tspn_stem_pool_bf16 (space-to-depth)")) return rc;
  const int64_t tiles_w = tspn::ceil_div(OW, TPX), nrows = NB * PH;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // whole pooled rows per workgroup, as evenly as the row count allows (1440 rows of 8 frames of 720p: 480 x 3)
  const int64_t slots = 2LL * cus, per = tspn::ceil_div(nrows, slots);
  const unsigned grid = (unsigned)tspn::ceil_div(nrows, per);
  const __bf16* Wf = reinterpret_cast<const __bf16*>(frag);
  __bf16* o = reinterpret_cast<__bf16*>(out);
  const size_t smem = 2 * PSTAGE_BYTES + (size_t)(1 + TPX) * ((size_t)Cout * 2 + 16);
  if (Cout == 64) {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(stem_pool_bf16_kernel<2>), smem, "tspn_stem_pool_bf16")) return rc;
    hipLaunchKernelGGL(stem_pool_bf16_kernel<2>, dim3(grid), dim3(256), smem, s, S, Wf, bias, o, (int)SH, (int)SW, (int)OH,
                       (int)OW, (int)PH, (int)PW, (int)tiles_w, nrows);
  } else {
    static tspn::LdsLimit lds;
    if (int rc = lds.ensure(reinterpret_cast<const void*>(stem_pool_bf16_kernel<1>), smem, "tspn_stem_pool_bf16")) return rc;
    hipLaunchKernelGGL(stem_pool_bf16_kernel<1>, dim3(grid), dim3(256), smem, s, S, Wf, bias, o, (int)SH, (int)SW, (int)OH,
                       (int)OW, (int)PH, (int)PW, (int)tiles_w, nrows);
  }
  return tspn::check_launch("tspn_stem_pool_bf16");
}

extern "C" int tspn_max_pool_nhwc_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t C, int64_t k,
                                       int64_t stride, int64_t pad, uint16_t* out, void* stream) {
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k, TSPN_EINVAL,
               "tspn_max_pool_nhwc_bf16: bad sizes");
  const int64_t OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  TSPN_REQUIRE(OH > 0 && OW > 0, TSPN_EINVAL, "tspn_max_pool_nhwc_bf16: empty output");
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && out, TSPN_EINVAL, "tspn_max_pool_nhwc_bf16: null pointer");
  TSPN_REQUIRE(C % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               TSPN_EUNSUPPORTED, "tspn_max_pool_nhwc_bf16: needs C %% 8 == 0 and 16-byte aligned tensors");
  const int64_t total = NB * OH * OW * (C / 8);
  hipLaunchKernelGGL(max_pool_nhwc_bf16_kernel, dim3((unsigned)std::min<int64_t>(tspn::ceil_div(total, 256), 1 << 20)),
                     dim3(256), 0, TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x), NB, (int)H, (int)W, (int)C, (int)k,
                     (int)stride, (int)pad, (int)OH, (int)OW, reinterpret_cast<__bf16*>(out));
  return tspn::check_launch("tspn_max_pool_nhwc_bf16");
}
