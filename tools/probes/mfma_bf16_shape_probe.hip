// Probe: sustained FLOP/s of bare bf16 MFMA streams under the power cap, 32x32x16 against 16x16x32, with pseudo-random
// operands held in registers (1 and 2 waves per SIMD, every CU busy, ~80 ms per run so that the clock settles).
// Also fp32: 32x32x2 against 16x16x4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ bf16x8 rnd_frag(unsigned& s) {   // bf16 values in roughly [-2, 2): random sign / mantissa, small exponent range
  u32x4 r;
  for (int i = 0; i < 4; ++i) {
    const unsigned a = rnd(s), b = rnd(s);
    r[i] = ((a & 0x807f) | 0x3f00 | ((a >> 9) & 0x0080)) | (((b & 0x807f) | 0x3f00 | ((b >> 9) & 0x0080)) << 16);
  }
  return __builtin_bit_cast(bf16x8, r);
}

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, unsigned seed) {
  unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
  float r = 0.f;
  if (SHAPE == 0) {          // bf16 32x32x16: 8 accumulators, 8 A x 8 B fragments rotated
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = rnd_frag(s); b[j] = rnd_frag(s); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + k) & 7], b[k], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][15];
  } else if (SHAPE == 1) {   // bf16 16x16x32: 16 accumulators (same FLOP per iteration: 128 MFMAs of half the size)
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0, 0, 0, 0};
    bf16x8 a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = rnd_frag(s); b[j] = rnd_frag(s); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + k) & 7], b[(k + (j >> 3)) & 7], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) r += acc[j][0] + acc[j][3];
  } else if (SHAPE == 2) {   // fp32 32x32x2
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = (float)(rnd(s) >> 8) * (1.f / 8388608.f) - 1.f; b[j] = (float)(rnd(s) >> 8) * (1.f / 16777216.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + k) & 7], b[k], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][15];
  } else {                   // fp32 16x16x4
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = (float)(rnd(s) >> 8) * (1.f / 8388608.f) - 1.f; b[j] = (float)(rnd(s) >> 8) * (1.f / 16777216.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + k) & 7], b[(k + (j >> 3)) & 7], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) r += acc[j][0] + acc[j][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE>
void run(const char* name, int blocks, double flop_per_iter_wave, int iters) {
  float* out;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<SHAPE><<<blocks, 256>>>(out, iters / 10, 1u);
  (void)hipDeviceSynchronize();
  float best = 1e30f, last = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    probe<SHAPE><<<blocks, 256>>>(out, iters, 2u + rep);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&last, e0, e1);
    best = last < best ? last : best;
  }
  const double flop = (double)blocks * 4 * iters * flop_per_iter_wave;
  printf("%-44s %8.2f ms (last %8.2f)  %7.1f TFLOP/s\n", name, best, last, flop / best / 1e9);
  (void)hipFree(out);
}

int main() {
  const double f_bf = 64.0 * 32768.0, f_32 = 64.0 * 4096.0;
  run<0>("bf16 32x32x16, 1 wave/SIMD", 256, f_bf, 40000);
  run<1>("bf16 16x16x32, 1 wave/SIMD", 256, f_bf, 40000);
  run<0>("bf16 32x32x16, 2 waves/SIMD", 512, f_bf, 20000);
  run<1>("bf16 16x16x32, 2 waves/SIMD", 512, f_bf, 20000);
  run<2>("fp32 32x32x2, 1 wave/SIMD", 256, f_32, 20000);
  run<3>("fp32 16x16x4, 1 wave/SIMD", 256, f_32, 20000);
  run<2>("fp32 32x32x2, 2 waves/SIMD", 512, f_32, 10000);
  run<3>("fp32 16x16x4, 2 waves/SIMD", 512, f_32, 10000);
  return 0;
}
