// A whole identity-shortcut ResNet bottleneck block on bf16 operands in ONE launch (BASELINE cfg5
// backbone; detectron2 BottleneckBlock.forward, modeling/backbone/resnet.py, FrozenBN folded into W / b by the caller):
//     h1  = relu(W1 . x + b1)                         1x1, 4 CM -> CM
//     h2  = relu(W2 (*) h1 + b2)                      3x3 / pad 1 / stride 1
//     out = relu(W3 . h2 + b3 + x)                    1x1, CM -> 4 CM, the block's input as the residual
// for the memory-bound stages res2 / res3 (CM = 64 / 128).  The three-launch chain moves 28 CM bytes of HBM traffic per
// pixel (conv1: 8 in + 2 out; tail: 2 h1 + 8 residual + 8 out), this kernel 16 CM plus the halo of its tile: the
// 4 CM-channel map is read once (as conv1's operand; the residual re-read of the tile's own pixels hits L2) and
// h1 / h2 never leave the CU.
//
// Tile = TR x TW output pixels of one image, TR = 10 (CM = 64) / 6 (CM = 128), TW = 30.  conv1 runs on the tile with
// its one-pixel halo, (TR + 2) x 32 pixels = TR + 2 32-pixel MFMA column blocks, straight from global memory (the channels-last map IS the B-operand
// layout: a lane's eight channels of a pixel are 16 contiguous bytes); pixels outside the image give h1 = 0 (the 3x3's
// zero padding).  h1 sits in LDS as the B-operand image [CM / 8 groups][32 (TR + 2) + 4 slots][8 bf16] with slot = 32 r + c, so
// tap (a, b) of output column n = 32 r + c is slot n + 32 a + b -- the linear-range form of the fused tail
// (tspn_bottleneck_bf16.hip) with the LDS row pitch in place of the image width and no tap masks: the halo is in the
// tile.  Output columns with c >= 30 (two per row) are computed on whatever their slots hold and never stored.
// Contraction order and rounding points are those of conv2d_nhwc_bf16_kernel / bottleneck_bf16_kernel (64-channel part
// by part, its taps in a row, k-steps in order, one fp32 accumulator chain; relu(acc + b) rounded to bf16 once per
// layer; (acc + b3) + residual): the results are bit-identical to the three launches (tests/test_gpu_roi_head.py).
//
// What bounds these stages is the operand traffic between L2 and the CUs, not HBM and not the MFMA pipe (probe builds
// of the first form of this kernel -- 4 x 30 tiles, every phase on 2 x 2 waves: 326 / 223 us per 9 frames of 720p at
// CM = 64 / 128; with every weight fragment read from one L1-hot line 262 / 123, with conv1's x operand from one hot
// line 232 / 214, with both 162 / 79; profiles/r5/bottleneck_block_study.md).  So every phase tiles its waves to read
// each operand byte as few times as it can: the phases whose B operand lives in LDS (3x3, expand) split the waves along
// the ROWS of the weights only -- every weight fragment enters the CU once per tile --, conv1 (B operand from global
// memory) reads x once at CM = 64 (four waves along the pixels) and twice at CM = 128 (2 x 2: its weights are four
// times larger).  Weight fragments go from L2 straight into MFMA operand registers (fragment-major packing of
// tspn_pack_conv2d_frag_bf16) through rings several k-steps deep, W3 with its rows permuted at load time so that an
// accumulator lane holds 16 consecutive channels (32 contiguous bytes of residual / output per lane, no LDS transpose).
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int TW = 30;                  // output pixels of a tile row
constexpr int RP = TW + 2;              // LDS row pitch in slots = one 32-pixel column block per halo row
// rows of a tile: the weights are re-streamed from L2 per tile, so CM = 64 (small accumulators) takes ten rows
constexpr int tile_rows(int cm) { return cm == 64 ? 10 : 6; }

// PROJ: the first block of a stage -- its input has CIN channels at ST times the output resolution, conv1 is a 1x1 of
// stride ST and the residual is the PROJECTION shortcut  sc = bf16(Ws . x + bs)  (1x1, stride ST, CIN -> 4 CM, rounded
// to bf16 like the separate launch's output), computed in the expand phase from the tile's own input pixels instead of
// being written to and read back from HBM (res2.0: 265 + 265 MB per 9 frames of 720p).  Identity blocks: CIN = 4 CM,
// ST = 1, the input itself is the residual.
// RES = 0: identity block, the input is the residual.  1: projection shortcut computed here (PROJ, above).  2: the
// residual is a tensor of its own (`Wfs` then points at it: [NB, H, W, 4 CM], e.g. the output of a separately launched
// projection shortcut) -- res3.0 (256 -> 128 -> 512, stride 2): conv1 + 3x3 + expand in one launch, the shortcut conv as
// before (its GEMM inside this kernel needs the tile's input in all four row-waves and 384 registers: 230 against 222 us).
template <int CM, int CIN, int ST, int RES>
__global__ __launch_bounds__(THREADS, 2) void bottleneck_block_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wf1, const float* __restrict__ bias1,
    const __bf16* __restrict__ Wf2, const float* __restrict__ bias2, const __bf16* __restrict__ Wf3,
    const float* __restrict__ bias3, const __bf16* __restrict__ Wfs, const float* __restrict__ biass,
    __bf16* __restrict__ out, int H, int W, int Hin, int Win, int tiles_x, int tiles_y, int ntiles) {
  constexpr bool PROJ = RES == 1;
  static_assert(RES != 0 || (CIN == 4 * CM && ST == 1), "identity blocks take their input as the residual");
  constexpr int C4 = 4 * CM;
  constexpr int C1 = CIN / 64, C2 = CM / 64;        // 64-channel parts of conv1's / the 3x3's and expand's K
  constexpr int TR = tile_rows(CM);
  constexpr int NPB1 = TR + 2;                      // column blocks of conv1 (halo rows)
  constexpr int SL1 = NPB1 * 32 + 4;                // slots of the h1 image: + the two a garbage column may touch; = 4 mod 16
  constexpr int SL2 = TR * 32 + 4;                  // slots of the h2 image
  static_assert(SL1 % 16 == 4 && SL2 % 16 == 4, "conflict-free 16-byte fragment reads");
  constexpr int B3_OFF = (CM / 8) * SL1 * 16;       // b3 (4 CM floats) behind the h1 image
  extern __shared__ __attribute__((aligned(16))) char Bs[];   // h1 image, then (same memory) the h2 image

  // consecutive tiles stay on one XCD (shared halo rows in its L2)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (wg >= ntiles) return;
  const int per_img = tiles_x * tiles_y;
  const int img = wg / per_img, tin = wg - img * per_img;
  const int ty = tin / tiles_x, tx = tin - ty * tiles_x;
  const int y0 = ty * TR, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const unsigned woff = lane * 16;

  // one descriptor per tensor, based at this image: every offset below is a 32-bit byte offset inside the image
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x) + (int64_t)img * Hin * Win * CIN, 0, (int)(unsigned)((int64_t)Hin * Win * CIN * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
      out + (int64_t)img * H * W * C4, 0, (int)(unsigned)((int64_t)H * W * C4 * 2), 0x00020000);
  // where the residual rows come from (32-bit offsets inside the image, like the output's)
  const __amdgpu_buffer_rsrc_t rs_r = RES == 2 ? __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(Wfs) + (int64_t)img * H * W * C4, 0, (int)(unsigned)((int64_t)H * W * C4 * 2), 0x00020000) : rs_x;
  // byte offset of the input pixel under output pixel (yy, xx): a 1x1 conv of stride ST reads pixel (ST yy, ST xx)
  auto xpix = [&](int yy, int xx) { return (unsigned)(((yy * ST) * Win + xx * ST) * CIN * 2); };
  constexpr unsigned OOB = 0x80000000u;             // beyond every descriptor: loads give zeros, stores are dropped
  auto ldw = [&](const __bf16* base, int64_t byte_off) {           // a weight fragment: 16 bytes per lane
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off + woff);
  };

  for (int i = tid; i < CM; i += THREADS) {          // b3 (and bs) -> LDS (read in the epilogue of the expand; published by the first barrier)
    *reinterpret_cast<float4*>(Bs + B3_OFF + 16 * i) = *reinterpret_cast<const float4*>(bias3 + 4 * i);
    if constexpr (PROJ) *reinterpret_cast<float4*>(Bs + B3_OFF + 16 * CM + 16 * i) = *reinterpret_cast<const float4*>(biass + 4 * i);
  }

  // ---------------------------------------------------------------- conv1 on the (TR + 2) x 32 halo tile
  if constexpr (CM == 128) {
    // Four waves along the ROWS (one 32-row block each, all NPB1 column blocks): every W1 fragment enters the CU once.
    // x then has to be shared: its 64-channel chunks go through LDS by DMA -- a ring of two stages [8 groups][SL1 slots]
    // [8 bf16] that lives in the memory of the h1 image (which is only written after the last chunk has been read).
    // With x from global memory on 2 x 2 waves this phase took 83 of the kernel's 159 us per 9 frames: both operands
    // read twice, 622 MB through L2 (profiles/r5/bottleneck_block_study.md).
    constexpr int XST = 8 * SL1 * 16;                // bytes per stage; 2 XST = the h1 image
    static_assert(2 * XST == (CM / 8) * SL1 * 16 && NPB1 == 8, "x ring = h1 image");
    f32x16 acc[NPB1];
#pragma unroll
    for (int pj = 0; pj < NPB1; ++pj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pj][e] = 0.f;
    // DMA pieces: wave w stages channel groups 2 w, 2 w + 1 of all 256 slots: piece (gg, sq) = group 2 w + gg, slots
    // 64 sq .. + 63, lane = slot (the DMA writes lane * 16 behind the piece's base)
    unsigned xv[4];
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {
      const int yy = y0 - 1 + 2 * sq + (lane >> 5), xx = x0 - 1 + li;
      xv[sq] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xpix(yy, xx) : OOB;   // outside: zeros
    }
    auto stage = [&](int buf, int c) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs_x, (__attribute__((address_space(3))) void*)(Bs + buf * XST + ((2 * wave + gg) * SL1 + 64 * sq) * 16), 16,
              (int)xv[sq], c * 128 + (2 * wave + gg) * 16, 0, 0);
    };
    const int64_t w1row = (int64_t)C1 * 4096;
    f32x4 a[2][4];                                   // W1 fragments of a chunk's four k-steps, this chunk's and the next's
    auto load_a = [&](int buf, int c) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[buf][ks] = ldw(Wf1, wave * w1row + (int64_t)(4 * c + ks) * 1024);
    };
    stage(0, 0);
    load_a(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* xb = Bs + (kh * SL1 + li) * 16;
#pragma unroll
    for (int c = 0; c < C1; ++c) {
      // stage (c + 1) & 1 was last read in chunk c - 1; the barrier at its end lies behind every wave
      if (c + 1 < C1) { stage((c + 1) & 1, c + 1); load_a((c + 1) & 1, c + 1); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[c & 1][ks]);
#pragma unroll
        for (int pj = 0; pj < NPB1; ++pj)
          acc[pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              av, *reinterpret_cast<const bf16x8*>(xb + (c & 1) * XST + (2 * ks * SL1 + pj * 32) * 16), acc[pj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c + 1 have landed ...
      __syncthreads();                                     // ... everybody's have, and nobody reads stage c & 1 any more
    }
    // h1 = relu(acc + b1) -> bf16 -> LDS (over the x ring); exactly 0 outside the image
    unsigned inmask = 0;
#pragma unroll
    for (int pj = 0; pj < NPB1; ++pj) {
      const int yy = y0 - 1 + pj, xx = x0 - 1 + li;
      inmask |= (yy >= 0 && yy < H && xx >= 0 && xx < W ? 1u : 0u) << pj;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * wave + 8 * q + 4 * kh;
      const float4 bv = *reinterpret_cast<const float4*>(bias1 + ch);
#pragma unroll
      for (int pj = 0; pj < NPB1; ++pj) {
        bf16x4 v;
        v[0] = (__bf16)fmaxf(acc[pj][4 * q] + bv.x, 0.f);
        v[1] = (__bf16)fmaxf(acc[pj][4 * q + 1] + bv.y, 0.f);
        v[2] = (__bf16)fmaxf(acc[pj][4 * q + 2] + bv.z, 0.f);
        v[3] = (__bf16)fmaxf(acc[pj][4 * q + 3] + bv.w, 0.f);
        if (!((inmask >> pj) & 1u)) v = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SL1 + pj * 32 + li) * 16 + 8 * kh) = v;
      }
    }
    if (tid < 4 * (CM / 8)) {                        // the four slots behind the tile: only garbage columns read them; keep them finite
      const int g = tid >> 2, sl = NPB1 * 32 + (tid & 3);
      *reinterpret_cast<f32x4*>(Bs + (g * SL1 + sl) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (CM == 64) {
    // x straight from global memory into B-operand registers: four waves along the pixels read it exactly once
    constexpr int WM = CM == 64 ? 1 : 2, WN = 4 / WM;      // waves along rows / column blocks
    constexpr int MI = (CM / 32) / WM, PB = NPB1 / WN;     // 2 row blocks x 2 (CM = 64) / 4 (CM = 128) column blocks per wave
    const int wm = wave % WM, wn = wave / WM;
    f32x16 acc[MI][PB];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int pj = 0; pj < PB; ++pj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][pj][e] = 0.f;
    unsigned xo[PB];                                 // byte offset of this lane's pixel of each column block (channel 8 kh)
    bool inimg[PB];
#pragma unroll
    for (int pj = 0; pj < PB; ++pj) {
      const int yy = y0 - 1 + PB * wn + pj, xx = x0 - 1 + li;
      inimg[pj] = yy >= 0 && yy < H && xx >= 0 && xx < W;
      xo[pj] = inimg[pj] ? xpix(yy, xx) + 16 * kh : OOB;
    }
    const int64_t w1row = (int64_t)C1 * 4096;        // bytes per 32-row block of Wf1
    // operands D1 k-steps ahead (first form: one k-step ahead, every k-step waited a full round trip)
    constexpr int D1 = CM == 64 ? 4 : 3;
    static_assert(D1 <= CIN / 16, "conv1 ring");
    f32x4 a[D1][MI];
    bf16x8 b[D1][PB];
    auto load_ks = [&](int slot, int k) {            // k = 4 c + ks: channels 16 k ..
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[slot][mi] = ldw(Wf1, (MI * wm + mi) * w1row + (int64_t)k * 1024);
#pragma unroll
      for (int pj = 0; pj < PB; ++pj)
        b[slot][pj] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo[pj], k * 32, 0));
    };
    constexpr int KS1 = CIN / 16;
#pragma unroll
    for (int d = 0; d < D1; ++d) load_ks(d, d);
#pragma unroll
    for (int k = 0; k < KS1; ++k) {
      const int slot = k % D1;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[slot][mi]);
#pragma unroll
        for (int pj = 0; pj < PB; ++pj) acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[slot][pj], acc[mi][pj], 0, 0, 0);
      }
      if (k + D1 < KS1) load_ks(slot, k + D1);
      __builtin_amdgcn_sched_barrier(0);             // keep the ring D1 deep: the scheduler would hoist every load of the unrolled loop
    }
    // h1 = relu(acc + b1) -> bf16 -> LDS; exactly 0 outside the image.  Accumulator registers 4 q + j of lane (li, kh)
    // are rows 8 q + 4 kh + j of the block.
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * (MI * wm + mi) + 8 * q + 4 * kh;
        const float4 bv = *reinterpret_cast<const float4*>(bias1 + ch);
#pragma unroll
        for (int pj = 0; pj < PB; ++pj) {
          bf16x4 v;
          v[0] = (__bf16)fmaxf(acc[mi][pj][4 * q] + bv.x, 0.f);
          v[1] = (__bf16)fmaxf(acc[mi][pj][4 * q + 1] + bv.y, 0.f);
          v[2] = (__bf16)fmaxf(acc[mi][pj][4 * q + 2] + bv.z, 0.f);
          v[3] = (__bf16)fmaxf(acc[mi][pj][4 * q + 3] + bv.w, 0.f);
          if (!inimg[pj]) v = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SL1 + (PB * wn + pj) * 32 + li) * 16 + 8 * kh) = v;
        }
      }
    if (tid < 4 * (CM / 8)) {                        // the four slots behind the tile: only garbage columns read them; keep them finite
      const int g = tid >> 2, sl = NPB1 * 32 + (tid & 3);
      *reinterpret_cast<f32x4*>(Bs + (g * SL1 + sl) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- 3x3 on the h1 image: K = C2 parts x 9 taps x 64
  {
    constexpr int WM = CM / 32, WN = 4 / WM;         // one 32-row block per wave: every W2 fragment enters the CU once
    constexpr int PB = TR / WN;                      // (CM = 64: two waves along the rows x two along the pixels)
    const int wm = wave % WM, wn = wave / WM;
    f32x16 acc[PB];
#pragma unroll
    for (int pj = 0; pj < PB; ++pj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pj][e] = 0.f;
    const int64_t w2row = (int64_t)C2 * 9 * 4096;
    const char* hb = Bs + (kh * SL1 + PB * wn * 32 + li) * 16;   // + (group pair, tap offset, column block)
    constexpr int D2 = 8;                            // weight fragments (L2) eight k-steps ahead
    f32x4 a[D2];
    auto load_w = [&](int slot, int j) {             // j = (c 9 + tap) 4 + ks: fragments are stored in this order
      a[slot] = ldw(Wf2, wm * w2row + (int64_t)j * 1024);
    };
    constexpr int KS2 = C2 * 36;
#pragma unroll
    for (int d = 0; d < D2; ++d) load_w(d, d);
#pragma unroll
    for (int j = 0; j < KS2; ++j) {
      const int slot = j % D2;
      const int ks = j & 3, ct = j >> 2, c = ct / 9, tap = ct - 9 * c, ta = tap / 3, tb = tap - 3 * ta;
      const char* bp = hb + ((8 * c + 2 * ks) * SL1 + ta * RP + tb) * 16;
      const bf16x8 av = __builtin_bit_cast(bf16x8, a[slot]);
#pragma unroll
      for (int pj = 0; pj < PB; ++pj)
        acc[pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *reinterpret_cast<const bf16x8*>(bp + pj * 32 * 16), acc[pj], 0, 0, 0);
      if (j + D2 < KS2) load_w(slot, j + D2);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                 // every wave has read what it needs of h1: h2 takes its memory
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * wm + 8 * q + 4 * kh;
      const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
#pragma unroll
      for (int pj = 0; pj < PB; ++pj) {
        bf16x4 v;
        v[0] = (__bf16)fmaxf(acc[pj][4 * q] + bv.x, 0.f);
        v[1] = (__bf16)fmaxf(acc[pj][4 * q + 1] + bv.y, 0.f);
        v[2] = (__bf16)fmaxf(acc[pj][4 * q + 2] + bv.z, 0.f);
        v[3] = (__bf16)fmaxf(acc[pj][4 * q + 3] + bv.w, 0.f);
        *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SL2 + (PB * wn + pj) * 32 + li) * 16 + 8 * kh) = v;
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- expand + residual + ReLU: M = 4 CM, K = CM
  {
    // four waves along the rows: wave w owns row blocks [NPASS w, NPASS w + NPASS), one per pass, all TR column blocks --
    // every W3 fragment enters the CU once.  W3 rows permuted at load time (as in bottleneck_bf16_kernel): the lane that
    // feeds MFMA row 8 q + 4 h + j fetches the fragment slot of channel 16 h + 4 q + j, so accumulator lane (li, kh)
    // holds channels 16 kh + 0..15 of the block
    const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;
    constexpr int KS3 = CM / 16;
    constexpr int NPASS = CM / 32;
    const int64_t w3row = (int64_t)C2 * 4096;
    const char* hb = Bs + (kh * SL2 + li) * 16;
    // the TR output pixels of this lane (column block = tile row): valid if c < TW and inside the image
    unsigned po[TR];
#pragma unroll
    for (int pj = 0; pj < TR; ++pj) {
      const int yy = y0 + pj, xx = x0 + li;
      const bool ok = li < TW && yy < H && xx < W;
      po[pj] = ok ? (unsigned)(((yy * W + xx) * C4 + 16 * kh) * 2) : OOB;
    }
    u32x4_t keep[2] = {};                            // store data of the previous group (see the epilogue)
    // The wave's row blocks go in PAIRS (two 32-channel blocks = one 128-byte line of a pixel) and the tile's column blocks
    // in groups of two: the four 16-byte stores that complete a pixel's line are issued back to back.  (First form: one
    // row block per pass, the two halves of a line a whole pass apart -- L2 hands part-written lines to the fabric as
    // they are, profiles/r3: the expand phase took 150 of the kernel's 247 us at CM = 64, 1.8 TB/s of stores.)
    constexpr int NPP = NPASS / 2, NG = TR / 2;
    static_assert(NPASS % 2 == 0 && TR % 2 == 0, "pairs of row blocks, pairs of column blocks");
    const char* const b3s = Bs + B3_OFF + 64 * kh;   // bias3 in LDS (staged below): 16 floats per (row block, lane half)
    f32x4 a[2][KS3];                                 // W3 fragments of the pair: [row block][k-step]
    bf16x8 res[2][2][2][2];                          // residual rows two groups deep: [buffer][row block][column block][half]
    // PROJ: the residual of a group = bf16(Ws . x + bs) on the group's own input pixels: K = CIN in KSX k-steps, B operand
    // straight from global memory (lane = pixel, 16 bytes per k-step), Ws rows permuted like W3's.  The result goes into
    // `res` in the layout the loaded rows have.
    constexpr int KSX = CIN / 16;
    constexpr bool WS_RESIDENT = PROJ && KSX <= 4;   // the pair's Ws fragments stay in registers across the groups
    const int64_t wsrow = (int64_t)C1 * 4096;
    const char* const bss = Bs + B3_OFF + 16 * CM + 64 * kh;
    unsigned xc[TR];                                 // PROJ: byte offset of the input pixel under each output pixel (channel 8 kh)
    if constexpr (PROJ) {
#pragma unroll
      for (int pj = 0; pj < TR; ++pj) {
        const int yy = y0 + pj, xx = x0 + li;
        xc[pj] = (li < TW && yy < H && xx < W) ? xpix(yy, xx) + 16 * kh : OOB;
      }
    }
    f32x4 asr[WS_RESIDENT ? 2 : 1][WS_RESIDENT ? KSX : 1];
    auto res_issue = [&](int buf, int mbA, int g) {
      if constexpr (!PROJ) {
#pragma unroll
        for (int pj = 0; pj < 2; ++pj)
#pragma unroll
          for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              res[buf][ms][pj][h] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                  rs_r, (int)(po[2 * g + pj] == OOB ? OOB : po[2 * g + pj] + 16 * h), (mbA + ms) * 64, 0));
      } else {
        f32x16 sc[2][2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[ms][pj][e] = 0.f;
        constexpr int DX = 4;                        // operands four k-steps ahead
        bf16x8 xf[DX][2];
        f32x4 aw[DX][2];
        auto load_k = [&](int slot, int k) {
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
            xf[slot][pj] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xc[2 * g + pj], k * 32, 0));
          if constexpr (!WS_RESIDENT) {
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
              aw[slot][ms] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wfs) + (mbA + ms) * wsrow + k * 1024 + woff3);
          }
        };
#pragma unroll
        for (int d = 0; d < DX && d < KSX; ++d) load_k(d, d);
#pragma unroll
        for (int k = 0; k < KSX; ++k) {
#pragma unroll
          for (int ms = 0; ms < 2; ++ms) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, WS_RESIDENT ? asr[WS_RESIDENT ? ms : 0][WS_RESIDENT ? k : 0] : aw[k % DX][ms]);
#pragma unroll
            for (int pj = 0; pj < 2; ++pj) sc[ms][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, xf[k % DX][pj], sc[ms][pj], 0, 0, 0);
          }
          if (k + DX < KSX) load_k(k % DX, k + DX);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
          float bs16[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(bss + (mbA + ms) * 128 + 16 * i);
            bs16[4 * i] = t.x; bs16[4 * i + 1] = t.y; bs16[4 * i + 2] = t.z; bs16[4 * i + 3] = t.w;
          }
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              bf16x8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = (__bf16)(sc[ms][pj][8 * h + j] + bs16[8 * h + j]);   // the shortcut launch's rounding
              res[buf][ms][pj][h] = o;
            }
        }
      }
    };
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp) {
      const int mbA = NPASS * wave + 2 * pp;         // channels 32 mbA .. + 63
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int k = 0; k < KS3; ++k)
          a[ms][k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wf3) + (mbA + ms) * w3row + k * 1024 + woff3);
      if constexpr (WS_RESIDENT) {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int k = 0; k < KSX; ++k)
            asr[ms][k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wfs) + (mbA + ms) * wsrow + k * 1024 + woff3);
      }
      res_issue(0, mbA, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x16 acc[2][2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ms][pj][e] = 0.f;
#pragma unroll
        for (int k = 0; k < KS3; ++k) {
          bf16x8 b[2];
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) b[pj] = *reinterpret_cast<const bf16x8*>(hb + (2 * k * SL2 + (2 * g + pj) * 32) * 16);
#pragma unroll
          for (int ms = 0; ms < 2; ++ms) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[ms][k]);
#pragma unroll
            for (int pj = 0; pj < 2; ++pj) acc[ms][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[pj], acc[ms][pj], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!PROJ && g + 1 < NG) res_issue((g + 1) & 1, mbA, g + 1);   // the next group's rows fly under this group's epilogue
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          const unsigned pofs = po[2 * g + pj];
#pragma unroll
          for (int ms = 0; ms < 2; ++ms) {
            float bv[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float4 t = *reinterpret_cast<const float4*>(b3s + (mbA + ms) * 128 + 16 * i);
              bv[4 * i] = t.x; bv[4 * i + 1] = t.y; bv[4 * i + 2] = t.z; bv[4 * i + 3] = t.w;
            }
            u32x4_t o2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              bf16x8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j)
                o[j] = (__bf16)fmaxf((acc[ms][pj][8 * h + j] + bv[8 * h + j]) + (float)res[PROJ ? 0 : (g & 1)][ms][pj][h][j], 0.f);
              o2[h] = __builtin_bit_cast(u32x4_t, o);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o2[0], rs_o, (int)(pofs == OOB ? OOB : pofs), (mbA + ms) * 64, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o2[1], rs_o, (int)(pofs == OOB ? OOB : pofs + 16), (mbA + ms) * 64, 0);
            // STORE-DATA HAZARD (found the hard way, round 5): a 128-bit buffer store reads its data registers AFTER it
            // has issued, and hipcc does not separate it from a following VALU write of those registers when the store
            // has an SGPR soffset (GCNHazardRecognizer::createsVALUHazard assumes that form is safe).  With several waves
            // per SIMD the next group's `v_add_f32 v32, ...` landed in the data of the store before it, in the lanes the
            // store reads last (pixels 12..15 / 28..29 of a block; wrong bits in ~0.02 % of the outputs, different ones
            // every launch; none with one workgroup per CU).  The data registers of a group therefore stay LIVE until the
            // stores of the NEXT group have been issued (an empty asm that names them), i.e. for ~50 vector instructions
            // (tools/lint_store_hazard.py checks the ISA of every kernel for the pattern).
            asm volatile("" ::"v"(keep[0]), "v"(keep[1]));
            keep[0] = o2[0];
            keep[1] = o2[1];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (PROJ && g + 1 < NG) res_issue(0, mbA, g + 1);
      }
    }
    // ... and the last group's until well after its stores (the wave ends here)
    asm volatile("s_nop 15\n s_nop 15" ::"v"(keep[0]), "v"(keep[1]));
  }
}

template <int CM, int CIN, int ST, int RES>
int launch(const uint16_t* x, int64_t NB, int64_t Hin, int64_t Win, const uint16_t* f1, const float* b1, const uint16_t* f2,
           const float* b2, const uint16_t* f3, const float* b3, const uint16_t* fs, const float* bs, uint16_t* out,
           void* stream, const char* what) {
  constexpr int TR = tile_rows(CM);
  const int64_t H = (Hin - 1) / ST + 1, W = (Win - 1) / ST + 1;
  const int64_t tiles_x = tspn::ceil_div(W, TW), tiles_y = tspn::ceil_div(H, TR);
  const int64_t ntiles = NB * tiles_x * tiles_y;
  TSPN_REQUIRE(ntiles < (1LL << 30), TSPN_EUNSUPPORTED, "%s: grid too large", what);
  const int64_t grid = tspn::ceil_div(ntiles, 8) * 8;            // whole rounds over the eight XCDs
  constexpr size_t smem = (size_t)(CM / 8) * ((TR + 2) * 32 + 4) * 16 + (RES == 1 ? 2 : 1) * 4 * CM * 4;   // h1 image (h2 takes its place) + b3 (+ bs)
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(bottleneck_block_bf16_kernel<CM, CIN, ST, RES>), smem, what)) return rc;
  hipLaunchKernelGGL((bottleneck_block_bf16_kernel<CM, CIN, ST, RES>), dim3((unsigned)grid), dim3(THREADS), smem,
                     TSPN_STREAM(stream), reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(f1), b1,
                     reinterpret_cast<const __bf16*>(f2), b2, reinterpret_cast<const __bf16*>(f3), b3,
                     reinterpret_cast<const __bf16*>(fs), bs, reinterpret_cast<__bf16*>(out), (int)H, (int)W, (int)Hin,
                     (int)Win, (int)tiles_x, (int)tiles_y, (int)ntiles);
  return tspn::check_launch(what);
}

}  // namespace

extern "C" int tspn_bottleneck_block_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                          const uint16_t* frag1, const float* bias1, const uint16_t* frag2,
                                          const float* bias2, const uint16_t* frag3, const float* bias3, uint16_t* out,
                                          void* stream) {
  const char* what = "tspn_bottleneck_block_bf16";
  TSPN_REQUIRE(NB >= 0 && H > 0 && W > 0, TSPN_EINVAL, "%s: bad sizes", what);
  TSPN_REQUIRE(CM == 64 || CM == 128, TSPN_EUNSUPPORTED, "%s: bottleneck channels must be 64 or 128 (got %lld)", what,
               (long long)CM);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag1 && bias1 && frag2 && bias2 && frag3 && bias3 && out, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE(x != out, TSPN_EINVAL, "%s: the block cannot run in place (tiles read their neighbours' pixels)", what);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag1) && al16(bias1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) && al16(out),
               TSPN_EUNSUPPORTED, "%s: operands must be 16-byte aligned", what);
  // 32-bit byte offsets inside one image
  TSPN_REQUIRE(H * W * 4 * CM * 2 < (1LL << 31), TSPN_EUNSUPPORTED, "%s: one image's map must stay below 2 GB", what);
  if (CM == 128)
    return launch<128, 512, 1, 0>(x, NB, H, W, frag1, bias1, frag2, bias2, frag3, bias3, nullptr, nullptr, out, stream, what);
  return launch<64, 256, 1, 0>(x, NB, H, W, frag1, bias1, frag2, bias2, frag3, bias3, nullptr, nullptr, out, stream, what);
}

extern "C" int tspn_bottleneck_block_proj_bf16(const uint16_t* x, int64_t NB, int64_t Hin, int64_t Win, int64_t CIN,
                                               int64_t stride, int64_t CM, const uint16_t* frag1, const float* bias1,
                                               const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                               const float* bias3, const uint16_t* frags, const float* biass,
                                               uint16_t* out, void* stream) {
  const char* what = "tspn_bottleneck_block_proj_bf16";
  TSPN_REQUIRE(NB >= 0 && Hin > 0 && Win > 0, TSPN_EINVAL, "%s: bad sizes", what);
  // (res3.0 -- 256 -> 128 -> 512, stride 2 -- was built and measured as well: its shortcut GEMM needs the tile's input
  // pixels in all four row-waves and 384 registers per lane; 230 us against 222 for its four launches, so it is not here)
  TSPN_REQUIRE(CM == 64 && CIN == 64 && stride == 1, TSPN_EUNSUPPORTED,
               "%s: built for the first block of res2 (64 -> 64 -> 256, stride 1); got CIN=%lld CM=%lld stride=%lld", what,
               (long long)CIN, (long long)CM, (long long)stride);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag1 && bias1 && frag2 && bias2 && frag3 && bias3 && frags && biass && out, TSPN_EINVAL,
               "%s: null pointer", what);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag1) && al16(bias1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) &&
                   al16(frags) && al16(biass) && al16(out),
               TSPN_EUNSUPPORTED, "%s: operands must be 16-byte aligned", what);
  const int64_t H = (Hin - 1) / stride + 1, W = (Win - 1) / stride + 1;
  TSPN_REQUIRE(H * W * 4 * CM * 2 < (1LL << 31) && Hin * Win * CIN * 2 < (1LL << 31), TSPN_EUNSUPPORTED,
               "%s: one image's maps must stay below 2 GB", what);
  return launch<64, 64, 1, 1>(x, NB, Hin, Win, frag1, bias1, frag2, bias2, frag3, bias3, frags, biass, out, stream, what);
}

extern "C" int tspn_bottleneck_block_res_bf16(const uint16_t* x, int64_t NB, int64_t Hin, int64_t Win, int64_t CIN,
                                              int64_t stride, int64_t CM, const uint16_t* frag1, const float* bias1,
                                              const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                              const float* bias3, const uint16_t* residual, uint16_t* out, void* stream) {
  const char* what = "tspn_bottleneck_block_res_bf16";
  TSPN_REQUIRE(NB >= 0 && Hin > 0 && Win > 0, TSPN_EINVAL, "%s: bad sizes", what);
  TSPN_REQUIRE(CM == 128 && CIN == 256 && stride == 2, TSPN_EUNSUPPORTED,
               "%s: built for the first block of res3 (256 -> 128 -> 512, stride 2); got CIN=%lld CM=%lld stride=%lld", what,
               (long long)CIN, (long long)CM, (long long)stride);
  if (NB == 0) return TSPN_OK;
  TSPN_REQUIRE(x && frag1 && bias1 && frag2 && bias2 && frag3 && bias3 && residual && out, TSPN_EINVAL, "%s: null pointer", what);
  TSPN_REQUIRE(residual != out, TSPN_EINVAL, "%s: out must not alias the residual", what);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  TSPN_REQUIRE(al16(x) && al16(frag1) && al16(bias1) && al16(frag2) && al16(bias2) && al16(frag3) && al16(bias3) &&
                   al16(residual) && al16(out),
               TSPN_EUNSUPPORTED, "%s: operands must be 16-byte aligned", what);
  const int64_t H = (Hin - 1) / stride + 1, W = (Win - 1) / stride + 1;
  TSPN_REQUIRE(H * W * 4 * CM * 2 < (1LL << 31) && Hin * Win * CIN * 2 < (1LL << 31), TSPN_EUNSUPPORTED,
               "%s: one image's maps must stay below 2 GB", what);
  return launch<128, 256, 2, 2>(x, NB, Hin, Win, frag1, bias1, frag2, bias2, frag3, bias3, residual, nullptr, out, stream, what);
}
