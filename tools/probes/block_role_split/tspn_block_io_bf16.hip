// Identity bottleneck block of res2 (CM = 64: 256 -> 64 -> 64 -> 256 channels, stride 1) in one launch, with the work split
// by ROLE and the tiles walked by PERSISTENT workgroups (round 5):
//     out = relu( W3 . relu(W2 (*) relu(W1 . x + b1) + b2) + b3 + x )
// -- the tile (10 x 30 output pixels + halo), the three phases, the LDS images, the contraction order and the rounding points
// of bottleneck_block_bf16_kernel<64, 256, 1, 0> (tspn_block_bf16.hip): bit-identical results.
//
// Why: that kernel's phases add up (conv1 76 + 3x3 21 + expand 26 + stores 71 + residual 33 us per 9 frames of 720p,
// profiles/r5/bottleneck_block_study.md) because every vector-memory operation of a wave retires in order on one counter -- the
// residual rows (HBM) and the stores sit in front of the weight fragments the MFMAs wait for -- and a second workgroup per CU
// hides little of it (213 us with two per CU, 242 with one).  The role-split res4 tail (tspn_tail_io_bf16.hip,
// profiles/r5/tail_role_split.md) showed the cure; here, where the expand phase has only 4 k-steps, it needs one thing more:
//   waves 0-3 (compute): conv1, 3x3 and the expand's MFMAs of tile t, then straight on to tile t + 1.  Their vector-memory
//                        traffic: x as conv1's B operand and the weight fragments.  They synchronise among themselves through
//                        LDS counters (one per wave, the waiter takes the minimum), never through s_barrier.
//   waves 4-7 (io):      request THEIR 64 channels of the tile's 300 residual pixels at the start of the tile (160 registers
//                        per lane; they land under conv1), take the expand's fp32 sums group by group (64 channels x 64 pixels
//                        per wave pair) from an LDS exchange buffer, add b3 and the residual, ReLU, round, store -- line-major:
//                        lane l = piece l & 7 of pixel 8 t + (l >> 3), whole 128-byte lines per instruction -- while the
//                        compute waves are already in conv1 of the NEXT tile.
// Two more LDS counters per wave pair carry the hand-over (groups published / consumed, running across tiles).  121 KB of
// LDS, one workgroup of 512 threads per CU, grid = min(tiles, CUs).
#include <algorithm>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512;
constexpr int CM = 64, CIN = 256, C4 = 256;
constexpr int TW = 30, TR = 10;            // output pixels of a tile row, rows of a tile
constexpr int RP = TW + 2;                 // LDS row pitch in slots = one 32-pixel column block per halo row
constexpr int NPB1 = TR + 2;               // column blocks of conv1 (halo rows)
constexpr int SL1 = NPB1 * 32 + 4;         // slots of the h1 image (= 4 mod 16: conflict-free 16-byte fragment reads)
constexpr int SL2 = TR * 32 + 4;           // slots of the h2 image (over the h1 image)
constexpr int B3_OFF = (CM / 8) * SL1 * 16;
constexpr int XCH_OFF = B3_OFF + C4 * 4;   // fp32 sums of a group per wave pair: [64 pixels][64 channels + 4 floats of padding]
constexpr int XP = 64 * 4 + 16;
constexpr int XCH_WAVE = 64 * XP;
constexpr int FLAG_OFF = XCH_OFF + 4 * XCH_WAVE;
constexpr int F_PUB = 0, F_CON = 4;        // + 8 w: groups published / consumed by wave pair w
constexpr int F_BAR = 32;                  // + 4 w: the compute waves' barrier counters
constexpr int SMEM = FLAG_OFF + 64;
constexpr int NG = TR / 2;                 // expand groups per tile: pairs of tile rows (column blocks)
static_assert(SL1 % 16 == 4 && SL2 % 16 == 4 && SMEM <= 160 * 1024, "conflict-free fragment reads; LDS budget");

// Counters in LDS.  A wave's LDS operations execute in order: a counter written behind the data is seen behind the data.
// The polls are one asm block each (a C++ loop makes hipcc spill the straight-line code around it) and BOUNDED: a hand-over
// that were ever lost would give wrong results that every parity test sees, not a wave that never ends.
template <int OFF>
__device__ __forceinline__ void flag_set(unsigned base, int v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void flag_wait(unsigned base, int target) {
  int v, sv, n;
  asm volatile(
      "s_mov_b32 %2, 0x100000\n\t"
      "1:\n\t"
      "ds_read_b32 %0, %3 offset:%5\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %1, %0\n\t"
      "s_cmp_ge_i32 %1, %4\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sub_u32 %2, %2, 1\n\t"
      "s_cmp_eq_u32 %2, 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sleep 1\n\t"
      "s_branch 1b\n\t"
      "2:"
      : "=&v"(v), "=&s"(sv), "=&s"(n)
      : "v"(base), "s"(target), "n"(OFF)
      : "memory", "scc");
}
template <int OFF>
__device__ __forceinline__ void flag_wait4(unsigned base, int target) {    // until ALL FOUR counters at OFF .. OFF + 15 have reached it
  int v0, v1, v2, v3, sv, n;
  asm volatile(
      "s_mov_b32 %5, 0x100000\n\t"
      "1:\n\t"
      "ds_read_b32 %0, %6 offset:%8\n\t"
      "ds_read_b32 %1, %6 offset:%9\n\t"
      "ds_read_b32 %2, %6 offset:%10\n\t"
      "ds_read_b32 %3, %6 offset:%11\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_min_i32 %0, %0, %1\n\t"
      "v_min_i32 %2, %2, %3\n\t"
      "v_min_i32 %0, %0, %2\n\t"
      "s_nop 0\n\t"
      "v_readfirstlane_b32 %4, %0\n\t"
      "s_cmp_ge_i32 %4, %7\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sub_u32 %5, %5, 1\n\t"
      "s_cmp_eq_u32 %5, 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sleep 1\n\t"
      "s_branch 1b\n\t"
      "2:"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&s"(sv), "=&s"(n)
      : "v"(base), "s"(target), "n"(OFF), "n"(OFF + 4), "n"(OFF + 8), "n"(OFF + 12)
      : "memory", "scc");
}

__global__ __launch_bounds__(THREADS, 1) void block_io_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ Wf1, const float* __restrict__ bias1,
    const __bf16* __restrict__ Wf2, const float* __restrict__ bias2, const __bf16* __restrict__ Wf3,
    const float* __restrict__ bias3, __bf16* __restrict__ out, int H, int W, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char Bs[];   // h1 image, then (same memory) the h2 image; b3; exchange buffers; counters

  // consecutive first tiles stay on one XCD (shared halo rows in its L2); workgroup b then takes first(b) + G, + 2 G, ...
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int per_img = tiles_x * tiles_y;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  constexpr unsigned OOB = 0x80000000u;             // beyond every descriptor: loads give zeros, stores are dropped
  const int64_t img_elems = (int64_t)H * W * C4;
  const int img_bytes = (int)(unsigned)(img_elems * 2);

  for (int i = tid; i < CM; i += THREADS)           // b3 -> LDS
    *reinterpret_cast<float4*>(Bs + B3_OFF + 16 * i) = *reinterpret_cast<const float4*>(bias3 + 4 * i);
  if (tid >= 256 && tid < 272) reinterpret_cast<int*>(Bs + FLAG_OFF)[tid - 256] = 0;
  __syncthreads();                                  // the only workgroup barrier
  const unsigned fb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + FLAG_OFF;

  if (wave >= 4) {
    // ================================================================ io waves
    const int w4 = wave - 4;
    const unsigned fp = fb + 8 * w4;
    const int pc = lane & 7, pp = lane >> 3;
    // item t_ of group g (tile rows 2 g, 2 g + 1): row 2 g + (t_ >> 2), column 8 (t_ & 3) + pp, channels 64 w4 + 8 pc ..
    unsigned lvo = (unsigned)((pp * C4 + 64 * w4 + 8 * pc) * 2);
    unsigned xro = (unsigned)(XCH_OFF + w4 * XCH_WAVE + pp * XP + 32 * pc);
    float bv[8];
    {
      const char* const b3s = Bs + B3_OFF + (64 * w4 + 8 * pc) * 4;
      const float4 t0 = *reinterpret_cast<const float4*>(b3s), t1 = *reinterpret_cast<const float4*>(b3s + 16);
      bv[0] = t0.x; bv[1] = t0.y; bv[2] = t0.z; bv[3] = t0.w; bv[4] = t1.x; bv[5] = t1.y; bv[6] = t1.z; bv[7] = t1.w;
    }
    u32x4_t keep = {};
    int t = 0;
    for (int wg = wg0; wg < ntiles; wg += nwg, ++t) {
      asm volatile("" : "+v"(lvo), "+v"(xro));      // (keeps the per-item address arithmetic inside the tile loop: hoisted, it spills)
      const int img = wg / per_img, tin = wg - img * per_img;
      const int ty = tin / tiles_x, tx = tin - ty * tiles_x;
      const int y0 = ty * TR, x0 = tx * TW;
      const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x) + (int64_t)img * img_elems, 0, img_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)img * img_elems, 0, img_bytes, 0x00020000);
      auto voff_of = [&](int g, int t_) {
        const int yy = y0 + 2 * g + (t_ >> 2), col = 8 * (t_ & 3) + pp;
        return (int)((col < TW && yy < H && x0 + col < W) ? lvo : OOB);
      };
      auto soff_of = [&](int g, int t_) { return ((y0 + 2 * g + (t_ >> 2)) * W + x0 + 8 * (t_ & 3)) * C4 * 2; };
      bf16x8 res[NG][8];                             // the wave pair's 64 channels of the tile's residual pixels
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t_ = 0; t_ < 8; ++t_)
          res[g][t_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_r, voff_of(g, t_), soff_of(g, t_), 0));
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        flag_wait<F_PUB>(fp, NG * t + g + 1);        // the sums of group g are in the exchange buffer
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 sv[4][2];
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const char* const src = Bs + xro + 8 * (4 * half + t4) * XP;
            sv[t4][0] = *reinterpret_cast<const f32x4*>(src);
            sv[t4][1] = *reinterpret_cast<const f32x4*>(src + 16);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (half == 1) flag_set<F_CON>(fp, NG * t + g + 1);     // the buffer may take the next group
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const int t_ = 4 * half + t4;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)fmaxf((sv[t4][j >> 2][j & 3] + bv[j]) + (float)res[g][t_][j], 0.f);
            const u32x4_t o4 = __builtin_bit_cast(u32x4_t, o);
            __builtin_amdgcn_raw_buffer_store_b128(o4, rs_o, voff_of(g, t_), soff_of(g, t_), 0);
            // store-data hazard (tools/lint_store_hazard.py, profiles/r5/bottleneck_block_study.md §3): the data registers of a
            // store stay live until the next store has been issued
            asm volatile("" ::"v"(keep));
            keep = o4;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n s_nop 15" ::"v"(keep));
    return;
  }

  // ================================================================ compute waves
  const unsigned fp = fb + 8 * wave, fw = fb + 4 * wave;
  unsigned woff = lane * 16;
  const unsigned woff3 = (unsigned)((kh << 5) | (((li >> 2) & 1) << 4) | ((li >> 3) << 2) | (li & 3)) * 16;   // W3 rows permuted at load
  int nbar = 0;                                      // barriers passed so far
  auto sync_compute = [&]() {                        // all four compute waves have arrived (their LDS writes are visible)
    ++nbar;
    flag_set<F_BAR>(fw, nbar);
    flag_wait4<F_BAR>(fb, nbar);
  };
  int t = 0;
  for (int wg = wg0; wg < ntiles; wg += nwg, ++t) {
    asm volatile("" : "+v"(woff));                   // (see the io waves)
    const int img = wg / per_img, tin = wg - img * per_img;
    const int ty = tin / tiles_x, tx = tin - ty * tiles_x;
    const int y0 = ty * TR, x0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x) + (int64_t)img * img_elems, 0, img_bytes, 0x00020000);
    auto ldw = [&](const __bf16* base, int64_t byte_off) {           // a weight fragment: 16 bytes per lane
      return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off + woff);
    };
    // ---------------------------------------------------------------- conv1 on the (TR + 2) x 32 halo tile: x straight from
    // global memory into B-operand registers, four waves along the pixels (three column blocks each) read it exactly once
    {
      constexpr int MI = 2, PB = NPB1 / 4;
      const int wn = wave;
      f32x16 acc[MI][PB];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pj = 0; pj < PB; ++pj)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][pj][e] = 0.f;
      unsigned xo[PB];
      bool inimg[PB];
#pragma unroll
      for (int pj = 0; pj < PB; ++pj) {
        const int yy = y0 - 1 + PB * wn + pj, xx = x0 - 1 + li;
        inimg[pj] = yy >= 0 && yy < H && xx >= 0 && xx < W;
        xo[pj] = inimg[pj] ? (unsigned)((yy * W + xx) * CIN * 2) + 16 * kh : OOB;
      }
      constexpr int64_t w1row = (int64_t)(CIN / 64) * 4096;
      constexpr int D1 = 4, KS1 = CIN / 16;
      f32x4 a[D1][MI];
      bf16x8 b[D1][PB];
      auto load_ks = [&](int slot, int k) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[slot][mi] = ldw(Wf1, mi * w1row + (int64_t)k * 1024);
#pragma unroll
        for (int pj = 0; pj < PB; ++pj)
          b[slot][pj] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo[pj], k * 32, 0));
      };
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
        __builtin_amdgcn_sched_barrier(0);
      }
      // the h1 image takes the memory of the previous tile's h2 image: every compute wave has finished its expand reads
      sync_compute();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = 32 * mi + 8 * q + 4 * kh;
          const float4 bvv = *reinterpret_cast<const float4*>(bias1 + ch);
#pragma unroll
          for (int pj = 0; pj < PB; ++pj) {
            bf16x4 v;
            v[0] = (__bf16)fmaxf(acc[mi][pj][4 * q] + bvv.x, 0.f);
            v[1] = (__bf16)fmaxf(acc[mi][pj][4 * q + 1] + bvv.y, 0.f);
            v[2] = (__bf16)fmaxf(acc[mi][pj][4 * q + 2] + bvv.z, 0.f);
            v[3] = (__bf16)fmaxf(acc[mi][pj][4 * q + 3] + bvv.w, 0.f);
            if (!inimg[pj]) v = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SL1 + (PB * wn + pj) * 32 + li) * 16 + 8 * kh) = v;
          }
        }
      if (tid < 4 * (CM / 8)) {                      // the four slots behind the tile: only garbage columns read them; keep them finite
        const int g = tid >> 2, sl = NPB1 * 32 + (tid & 3);
        *reinterpret_cast<f32x4*>(Bs + (g * SL1 + sl) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    sync_compute();

    // ---------------------------------------------------------------- 3x3 on the h1 image: K = 9 taps x 64; two waves along
    // the rows x two along the pixels: every W2 fragment enters the CU twice
    {
      constexpr int WM = CM / 32, WN = 4 / WM, PB = TR / WN;
      const int wm = wave % WM, wn = wave / WM;
      f32x16 acc[PB];
#pragma unroll
      for (int pj = 0; pj < PB; ++pj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pj][e] = 0.f;
      constexpr int64_t w2row = (int64_t)9 * 4096;
      const char* hb = Bs + (kh * SL1 + PB * wn * 32 + li) * 16;
      constexpr int D2 = 8, KS2 = 36;
      f32x4 a[D2];
      auto load_w = [&](int slot, int j) { a[slot] = ldw(Wf2, wm * w2row + (int64_t)j * 1024); };
#pragma unroll
      for (int d = 0; d < D2; ++d) load_w(d, d);
#pragma unroll
      for (int j = 0; j < KS2; ++j) {
        const int slot = j % D2;
        const int ks = j & 3, tap = j >> 2, ta = tap / 3, tb = tap - 3 * ta;
        const char* bp = hb + ((2 * ks) * SL1 + ta * RP + tb) * 16;
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[slot]);
#pragma unroll
        for (int pj = 0; pj < PB; ++pj)
          acc[pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *reinterpret_cast<const bf16x8*>(bp + pj * 32 * 16), acc[pj], 0, 0, 0);
        if (j + D2 < KS2) load_w(slot, j + D2);
        __builtin_amdgcn_sched_barrier(0);
      }
      sync_compute();                                // every wave has read what it needs of h1: h2 takes its memory
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * wm + 8 * q + 4 * kh;
        const float4 bvv = *reinterpret_cast<const float4*>(bias2 + ch);
#pragma unroll
        for (int pj = 0; pj < PB; ++pj) {
          bf16x4 v;
          v[0] = (__bf16)fmaxf(acc[pj][4 * q] + bvv.x, 0.f);
          v[1] = (__bf16)fmaxf(acc[pj][4 * q + 1] + bvv.y, 0.f);
          v[2] = (__bf16)fmaxf(acc[pj][4 * q + 2] + bvv.z, 0.f);
          v[3] = (__bf16)fmaxf(acc[pj][4 * q + 3] + bvv.w, 0.f);
          *reinterpret_cast<bf16x4*>(Bs + ((ch >> 3) * SL2 + (PB * wn + pj) * 32 + li) * 16 + 8 * kh) = v;
        }
      }
    }
    sync_compute();

    // ---------------------------------------------------------------- expand: M = 256, K = 64.  Wave w owns row blocks 2 w,
    // 2 w + 1 (channels 64 w .. + 63), all ten column blocks in groups of two; the sums of a group go to io wave w
    {
      constexpr int KS3 = CM / 16;
      constexpr int64_t w3row = 4096;
      const char* hb = Bs + (kh * SL2 + li) * 16;
      f32x4 a[2][KS3];
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int k = 0; k < KS3; ++k)
          a[ms][k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wf3) + (2 * wave + ms) * w3row + k * 1024 + woff3);
      char* const xl = Bs + XCH_OFF + wave * XCH_WAVE + li * XP + 64 * kh;     // + (32 pj) XP + 128 ms + 16 q
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
        flag_wait<F_CON>(fp, NG * t + g);            // the io wave has taken the previous group out of the buffer
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(xl + pj * 32 * XP + 128 * ms + 16 * q) =
                  f32x4{acc[ms][pj][4 * q], acc[ms][pj][4 * q + 1], acc[ms][pj][4 * q + 2], acc[ms][pj][4 * q + 3]};
        flag_set<F_PUB>(fp, NG * t + g + 1);
      }
    }
  }
}

}  // namespace

namespace tspn {
// called by tspn_bottleneck_block_bf16 (tspn_block_bf16.hip) for CM = 64; the arguments have been validated there
int launch_block_io_cm64(const uint16_t* x, int64_t NB, int64_t H, int64_t W, const uint16_t* f1, const float* b1,
                         const uint16_t* f2, const float* b2, const uint16_t* f3, const float* b3, uint16_t* out, void* stream,
                         const char* what) {
  const int64_t tiles_x = ceil_div(W, TW), tiles_y = ceil_div(H, TR);
  const int64_t ntiles = NB * tiles_x * tiles_y;
  TSPN_REQUIRE(ntiles < (1LL << 26), TSPN_EUNSUPPORTED, "%s: too many tiles", what);         // (32-bit LDS counters: 5 groups, 4 barriers per tile)
  static LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(block_io_bf16_kernel), SMEM, what)) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t grid = std::min<int64_t>(ntiles, cus > 0 ? cus : 256);
  hipLaunchKernelGGL(block_io_bf16_kernel, dim3((unsigned)grid), dim3(THREADS), SMEM, TSPN_STREAM(stream),
                     reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(f1), b1, reinterpret_cast<const __bf16*>(f2), b2,
                     reinterpret_cast<const __bf16*>(f3), b3, reinterpret_cast<__bf16*>(out), (int)H, (int)W, (int)tiles_x, (int)tiles_y,
                     (int)ntiles);
  return check_launch(what);
}
}  // namespace tspn
