// Winograd F(4,3) over time for the k=3 temporal conv of the tracklet projections (gfx950, fp32).
//
// Four adjacent output frames (t .. t+3) from the six inputs d = x[t-1 .. t+4]:
//     y = A^T [ (G g) . (B^T d) ]      contraction over input channels only, 6 positions j
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (packed from fp64)
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// i.e. 6 channel-GEMMs on a quarter of the columns: HALF the MFMA work of the direct form, 3/4 of
// F(2,3)'s.  Exact in real arithmetic; in fp32 the larger transform coefficients cost accuracy: measured
// against float64 on this path's data (|x| <= 1, weights ~ N(0, 0.01), K = 2048) the worst element is
// within 2.5x of the direct fp32 conv's own error (1-2e-5 on outputs of rms 0.4-0.8), far inside the
// 1e-4 bound of the path; F(2,3) is more accurate than the direct form and stays available.
//
// Structure = conv3_wino2_cl_kernel's: 4 waves, workgroup tile 128 output channels x 32 quads (128
// frames), wave = 32 channels x 32 quads x 6 positions (6 accumulator blocks); K chunk 8 channels;
// weights [6 j][8 ch][128 m] and the x tile by LDS-DMA, double-buffered; the input transform is
// computed once per workgroup into an LDS V tile [2 g][6 j (+1)][32 quads][4 ch] -- wave p computes
// part p of every (group, quad) item: V0 | V5 | V1,V2 | V3,V4 -- for chunk c+1 under the MFMAs of c.
// Quads never straddle tracklets: a tracklet has ceil(T/4) quads, frames >= T of its last quad are
// masked on input and not stored (any T).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "tspn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int BM = 128;
constexpr int QT = 32;                 // quads per workgroup
constexpr int KC = 8;
constexpr int SLP = 132;               // x slots per channel group (4 QT + 2 = 130 used)
constexpr int A_ST = 6 * KC * BM;      // floats
constexpr int X_ST = 2 * SLP * 4;
constexpr int V_ST = 2 * 7 * QT * 4;     // [2 g][6 j + 1 scratch plane][32 quads][4 ch]
constexpr size_t SMEM_BYTES = sizeof(float) * 2 * (A_ST + X_ST + V_ST);

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ void pack_conv3_wino43_kernel(const float* __restrict__ W, int64_t M, int64_t Cin,
                                         int64_t split, float* __restrict__ packed) {
  const int64_t Mp = split > 0 ? 2 * M : M;
  const int64_t Cp = split > 0 ? split : Cin;
  const int64_t total = Cp * Mp;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = o % Mp;
    const int64_t ci = o / Mp;
    const int64_t m = r < M ? r : r - M;
    const int64_t c = r < M ? ci : ci + split;
    const float* g = W + (m * Cin + c) * 3;
    const double g0 = g[0], g1 = g[1], g2 = g[2];
    packed[0 * total + o] = (float)(g0 / 4.0);
    packed[1 * total + o] = (float)(-(g0 + g1 + g2) / 6.0);
    packed[2 * total + o] = (float)(-(g0 - g1 + g2) / 6.0);
    packed[3 * total + o] = (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0);
    packed[4 * total + o] = (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0);
    packed[5 * total + o] = (float)g2;
  }
}

__global__ __launch_bounds__(THREADS, 2) void conv3_wino43_cl_kernel(
    const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int T, int M, int nq, int64_t nquads, int64_t ncols, int tiles_m,
    int tiles_n, int relu, int ldy, int GM, int vec4) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);
  float* Xs = As + 2 * A_ST;
  float* Vs = Xs + 2 * X_ST;

  // workgroup -> tile: bijective XCD remap, then groups of GM weight panels x all quad tiles
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
  const int m0 = tile_m * BM;
  const int64_t Q0 = (int64_t)tile_n * QT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;

  // quad Q -> (tracklet b, quad q in it); row of its first output frame in the flat [B*T] frame space
  auto quad_row = [&](int64_t Q, int& q) -> int64_t {
    const int64_t b = Q / nq;
    q = (int)(Q - b * nq);
    return b * T + 4 * q;
  };
  int q_first;
  const int64_t row0 = quad_row(Q0, q_first) - 1;   // slot s of the x tile <-> frame row row0 + s

  // ---- DMA sources.  Weights: piece p = 6 wave + i = (j = p / 4, channel pair d = p % 4), 128 m each.
  const float* asrc[6];
  int aoff[6];
  {
    const int am = (lane & 31) * 4;
    const int amc = m0 + am < M ? m0 + am : 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int p = wave * 6 + i;
      const int j = p >> 2, d = p & 3;
      asrc[i] = Wp + ((int64_t)j * Cin + 2 * d + (lane >> 5)) * M + amc;
      aoff[i] = (j * KC + 2 * d) * BM;
    }
  }
  const float* bsrc[2];
  bool bval[2];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int u = 64 * (wave + 4 * qq) + lane;
    const int g = u / SLP, slot = u - g * SLP;
    bval[qq] = u < 2 * SLP && slot < 4 * QT + 2;
    int64_t n = row0 + slot;
    n = n < 0 ? 0 : (n < ncols ? n : ncols - 1);
    bsrc[qq] = x + n * Cin + 4 * (g < 2 ? g : 0);
  }
  const int64_t a_step = (int64_t)KC * M;
  auto stage_a = [&](int buf, auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
#if !defined(TSPN_W43_ABL_NODMA)
    glds16(asrc[i], As + buf * A_ST + aoff[i]);
#endif
    asrc[i] += a_step;
  };
  auto stage_x = [&](int buf, auto q_tag) {
    constexpr int qq = decltype(q_tag)::value;
#if !defined(TSPN_W43_ABL_NODMA)
    if (bval[qq]) glds16(bsrc[qq], Xs + buf * X_ST + 64 * (wave + 4 * qq) * 4);
#endif
    bsrc[qq] += KC;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;

  // ---- transform item of this thread: quad tk, channel group tg; wave = part (V0 | V5 | V1,V2 | V3,V4).
  // The sequence-end masks (frame 4q + i - 1 outside the tracklet -> 0) are folded into the transform
  // coefficients of the thread, so a part costs 6 (waves 0, 1) or 12 (waves 2, 3) packed VALU per chunk.
  const int tk = tid & 31, tg = (tid >> 5) & 1;
  int tslot;          // slot of d0 (frame 4q - 1) of this quad
  float tc[4];        // coefficients (see transform)
  {
    const int64_t Q = Q0 + tk;
    int q = 0;
    const bool okq = Q < nquads;
    const int64_t r = okq ? quad_row(Q, q) : row0 + 1;
    tslot = (int)(r - 1 - row0);
    float tm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = 4 * q + i - 1;
      tm[i] = (okq && t >= 0 && t < T) ? 1.f : 0.f;
    }
    // (V0 / V5 in the same s - r form and rounding order as the branch-free transform of tspn_wino43r.hip,
    // so that the two kernels stay bit-identical)
    if (wave == 0) {          // V0 = (4 d0 - 5 d2) - (-d4)
      tc[0] = 4.f * tm[0]; tc[1] = -5.f * tm[2]; tc[2] = -tm[4]; tc[3] = 0.f;
    } else if (wave == 1) {   // V5 = (4 d1 - 5 d3) - (-d5)
      tc[0] = 4.f * tm[1]; tc[1] = -5.f * tm[3]; tc[2] = -tm[5]; tc[3] = 0.f;
    } else if (wave == 2) {   // s = d4 - 4 d2, r = 4 d1 - d3: V1 = s - r, V2 = s + r
      tc[0] = tm[4]; tc[1] = -4.f * tm[2]; tc[2] = 4.f * tm[1]; tc[3] = -tm[3];
    } else {                  // s = d4 - d2, r = 2 d1 - 2 d3: V3 = s - r, V4 = s + r
      tc[0] = tm[4]; tc[1] = -tm[2]; tc[2] = 2.f * tm[1]; tc[3] = -2.f * tm[3];
    }
  }
  auto transform = [&](int xbuf, int vbuf) {
#if defined(TSPN_W43_ABL_NOXFORM)
    return;
#endif
    const float* xp = Xs + xbuf * X_ST + (tg * SLP + tslot) * 4;
    float* vp = Vs + vbuf * V_ST + (tg * 7 * QT + tk) * 4;
    auto D = [&](int i) { return *reinterpret_cast<const f32x4*>(xp + 4 * i); };
    auto bc = [](float v) { return f32x4{v, v, v, v}; };
    if (wave < 2) {
      const f32x4 sv = __builtin_elementwise_fma(bc(tc[0]), D(wave), bc(tc[1]) * D(wave + 2));
      *reinterpret_cast<f32x4*>(vp + (wave == 0 ? 0 : 5) * QT * 4) = sv - bc(tc[2]) * D(wave + 4);
    } else {
      const f32x4 sv = __builtin_elementwise_fma(bc(tc[0]), D(4), bc(tc[1]) * D(2));
      const f32x4 rv = __builtin_elementwise_fma(bc(tc[2]), D(1), bc(tc[3]) * D(3));
      const int j = wave == 2 ? 1 : 3;
      *reinterpret_cast<f32x4*>(vp + j * QT * 4) = sv - rv;
      *reinterpret_cast<f32x4*>(vp + (j + 1) * QT * 4) = sv + rv;
    }
  };

  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int nchunks = Cin / KC;
  stage_a(0, I0{}); stage_a(0, I1{}); stage_a(0, I2{}); stage_a(0, I3{}); stage_a(0, I4{}); stage_a(0, I5{});
  stage_x(0, I0{}); stage_x(0, I1{});
  if (nchunks > 1) { stage_x(1, I0{}); stage_x(1, I1{}); }
  __syncthreads();
  transform(0, 0);
  __syncthreads();

  // chunk c: MFMAs on (A_c, V_c); meanwhile DMA A_{c+1}, x_{c+2} and transform x_{c+1} -> V_{c+1}
  auto chunk_body = [&](int c, auto more1_tag, auto more2_tag) {
    constexpr bool MORE1 = decltype(more1_tag)::value;
    constexpr bool MORE2 = decltype(more2_tag)::value;
    const int buf = c & 1;
    const float* Ab = As + buf * A_ST + (4 * kh) * BM + wave * 32 + li;
    const float* Vb = Vs + buf * V_ST + (kh * 7 * QT + li) * 4;
    float4 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = *reinterpret_cast<const float4*>(Vb + j * QT * 4);
    float a[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[j][e] = Ab[(j * KC + e) * BM];
    if (MORE1) transform(buf ^ 1, buf ^ 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* vp = reinterpret_cast<const float*>(&v[j]);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], vp[e], acc[j], 0, 0, 0);
        if (MORE1) {
          if (e == 0 && j == 1) stage_a(buf ^ 1, I0{});
          if (e == 0 && j == 4) stage_a(buf ^ 1, I1{});
          if (e == 1 && j == 1) stage_a(buf ^ 1, I2{});
          if (e == 1 && j == 4) stage_a(buf ^ 1, I3{});
          if (e == 2 && j == 1) stage_a(buf ^ 1, I4{});
          if (e == 2 && j == 4) stage_a(buf ^ 1, I5{});
        }
        if (MORE2) {
          if (e == 3 && j == 0) stage_x(buf, I0{});
          if (e == 3 && j == 3) stage_x(buf, I1{});
        }
      }
    }
    __syncthreads();
  };
  int c = 0;
  for (; c + 2 < nchunks; ++c) chunk_body(c, std::true_type{}, std::true_type{});
  if (c + 1 < nchunks) {
    chunk_body(c, std::true_type{}, std::false_type{});
    ++c;
  }
  chunk_body(c, std::false_type{}, std::false_type{});

  // ---- output transform + store: lane column = quad -> frames 4q .. 4q+3
  {
    const int64_t Q = Q0 + li;
    if (Q < nquads) {
      int q;
      const int64_t r = quad_row(Q, q);
      const int64_t b = (r - 4 * q) / T;
      const int t = 4 * q;
      float* ycol = y + (b * M) * (int64_t)ldy + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < M) {
          const float p12 = acc[1][e] + acc[2][e], m12 = acc[1][e] - acc[2][e];
          const float p34 = acc[3][e] + acc[4][e], m34 = acc[3][e] - acc[4][e];
          float o0 = acc[0][e] + p12 + p34;
          float o1 = m12 + 2.f * m34;
          float o2 = p12 + 4.f * p34;
          float o3 = m12 + 8.f * m34 + acc[5][e];
          if (bias != nullptr) {
            const float bb = bias[m];
            o0 += bb; o1 += bb; o2 += bb; o3 += bb;
          }
          if (relu) {
            o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f);
          }
          float* dst = ycol + (int64_t)m * ldy;
          if (vec4) {   // rows padded to >= 4 nq frames and 16-byte aligned: frames >= T land in the padding
            *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
          } else {
            dst[0] = o0;
            if (t + 1 < T) dst[1] = o1;
            if (t + 2 < T) dst[2] = o2;
            if (t + 3 < T) dst[3] = o3;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int tspn_pack_conv3_wino43_f32(const float* W, int64_t M, int64_t Cin, int64_t split,
                                          float* packed, void* stream) {
  TSPN_REQUIRE(W && packed, TSPN_EINVAL, "tspn_pack_conv3_wino43_f32: null pointer");
  TSPN_REQUIRE(M > 0 && Cin > 0 && split >= 0, TSPN_EINVAL, "tspn_pack_conv3_wino43_f32: bad sizes");
  TSPN_REQUIRE(split == 0 || Cin == 2 * split, TSPN_EINVAL,
               "tspn_pack_conv3_wino43_f32: split=%lld requires Cin == 2*split (Cin=%lld)",
               (long long)split, (long long)Cin);
  const int64_t total = M * Cin;
  const int blocks = (int)std::min<int64_t>(tspn::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(pack_conv3_wino43_kernel, dim3(blocks), dim3(256), 0, TSPN_STREAM(stream), W, M, Cin,
                     split, packed);
  return tspn::check_launch("tspn_pack_conv3_wino43_f32");
}

int tspn::conv3_tc_wino43(const float* x, int64_t B, int64_t T, int64_t Cin, const float* packed6,
                          int64_t M, const float* bias, int relu, float* y, int64_t ldy, void* stream) {
  TSPN_REQUIRE(B >= 0 && Cin > 0 && T > 0 && M > 0 && ldy >= T && ldy < (1 << 24), TSPN_EINVAL,
               "tspn_conv3_tc_wino43_f32: bad sizes B=%lld T=%lld Cin=%lld M=%lld ldy=%lld", (long long)B,
               (long long)T, (long long)Cin, (long long)M, (long long)ldy);
  if (B == 0) return TSPN_OK;
  TSPN_REQUIRE(x && packed6 && y, TSPN_EINVAL, "tspn_conv3_tc_wino43_f32: null pointer");
  TSPN_REQUIRE(Cin % KC == 0 && M % 4 == 0, TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_wino43_f32: needs Cin %% 8 == 0, M %% 4 == 0 (Cin=%lld M=%lld)", (long long)Cin,
               (long long)M);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(packed6) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y) & 3) == 0,
               TSPN_EUNSUPPORTED, "tspn_conv3_tc_wino43_f32: x/packed must be 16-byte aligned");
  TSPN_REQUIRE(Cin < (1 << 24) && T < (1 << 24) && M < (1 << 24), TSPN_EUNSUPPORTED,
               "tspn_conv3_tc_wino43_f32: dimension too large");
  const int64_t nq = tspn::ceil_div(T, 4);
  const int64_t nquads = B * nq;
  const int64_t tiles_m = tspn::ceil_div(M, BM), tiles_n = tspn::ceil_div(nquads, QT);
  TSPN_REQUIRE(tiles_m * tiles_n < (1LL << 31), TSPN_EUNSUPPORTED, "tspn_conv3_tc_wino43_f32: grid too large");
  const int vec4 = (ldy % 4 == 0) && (ldy >= 4 * nq) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  static tspn::LdsLimit lds;
  if (int rc = lds.ensure(reinterpret_cast<const void*>(conv3_wino43_cl_kernel), SMEM_BYTES,
                          "tspn_conv3_tc_wino43_f32"))
    return rc;
  const int gm_tiles = tspn::kWinoPanelGroup;
  hipLaunchKernelGGL(conv3_wino43_cl_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(THREADS), SMEM_BYTES,
                     TSPN_STREAM(stream), x, packed6, bias, y, (int)Cin, (int)T, (int)M, (int)nq, nquads,
                     B * T, (int)tiles_m, (int)tiles_n, relu, (int)ldy, gm_tiles, vec4);
  return tspn::check_launch("tspn_conv3_tc_wino43_f32");
}

extern "C" int tspn_conv3_tc_wino43_f32(const float* x, int64_t B, int64_t T, int64_t Cin,
                                        const float* packed6, int64_t M, const float* bias, int relu,
                                        float* y, void* stream) {
  return tspn::conv3_tc_wino43(x, B, T, Cin, packed6, M, bias, relu, y, T, stream);
}
