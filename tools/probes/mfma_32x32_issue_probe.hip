// Probe: issue rate of v_mfma_f32_32x32x2_f32 in the conv kernel's pattern (6 accumulators, position pairs,
// 4 k-steps each = 24 MFMAs per "chunk") with 1 and 2 waves per SIMD, optionally with an s_barrier per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BAR>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, float seed) {
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  f32x4 a[6], v[6];
  for (int j = 0; j < 6; ++j) { a[j] = f32x4{seed, seed + 1, seed + 2, seed + threadIdx.x}; v[j] = f32x4{1, 2, 3, seed}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * p][e], v[2 * p][e], acc[2 * p], 0, 0, 0);
        acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * p + 1][e], v[2 * p + 1][e], acc[2 * p + 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR == 1) __builtin_amdgcn_s_barrier();
    if (BAR == 2 && (it & 3) == 3) __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int j = 0; j < 6; ++j) s += acc[j][0] + acc[j][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BAR>
void run(const char* name, int blocks) {
  const int iters = 3000;
  float* out;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<BAR><<<blocks, 256>>>(out, 100, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<BAR><<<blocks, 256>>>(out, iters, 1.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)(blocks / 256) * iters * 24;
  printf("%-40s %8.3f ms  %6.2f ns per MFMA per SIMD = %5.1f cycles at 2.4 GHz (%5.1f at 2.25)\n", name, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ms * 1e6 / per_simd * 2.25);
  (void)hipFree(out);
}

int main() {
  run<0>("1 wave/SIMD, no barrier", 256);
  run<0>("2 waves/SIMD, no barrier", 512);
  run<1>("2 waves/SIMD, barrier per chunk", 512);
  run<2>("2 waves/SIMD, barrier per 4 chunks", 512);
  run<1>("1 wave/SIMD, barrier per chunk", 256);
  return 0;
}
