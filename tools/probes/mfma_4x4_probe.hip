// Probe: issue rate of v_mfma_f32_4x4x1_16B_f32 (2 passes) against v_mfma_f32_16x16x4_f32 (8 passes) on gfx950,
// alone and with the VALU mix of the pair stage (2 VALU per activation, 3 MFMAs per activation column set).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_4x4_probe tools/probes/mfma_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, float seed) {
  f32x4 acc[24];
  for (int i = 0; i < 24; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f;
  float u = seed * 3.f + threadIdx.x, v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed - i;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 24 x 4x4x1_16B, no VALU
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[3 * i + 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, v[i], acc[3 * i + 0], 4, 3, 0);
        acc[3 * i + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, v[i], acc[3 * i + 1], 4, 3, 0);
        acc[3 * i + 2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, v[i], acc[3 * i + 2], 4, 3, 0);
      }
    } else if (MODE == 1) {   // same + 2 VALU per activation
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float h = fmaxf(u + v[i], 0.f);
        acc[3 * i + 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, h, acc[3 * i + 0], 4, 3, 0);
        acc[3 * i + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, h, acc[3 * i + 1], 4, 3, 0);
        acc[3 * i + 2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, h, acc[3 * i + 2], 4, 3, 0);
      }
      u += 1.0f;
    } else if (MODE == 2) {   // 6 x 16x16x4 (same MACs as 24 x 4x4x1... 6*1024 = 24*256)
#pragma unroll
      for (int i = 0; i < 6; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, v[i], acc[i], 0, 0, 0);
    } else {                  // 8 x 16x16x4 + 2 VALU per MFMA (the current kernel's mix: 16 rows padded)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float h = fmaxf(u + v[i], 0.f);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, h, acc[i], 0, 0, 0);
      }
      u += 1.0f;
    }
  }
  float s = 0;
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int mfma_per_iter, int macs_per_mfma) {
  const int blocks = 512, iters = 20000;
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 512 blocks x 4 waves = 2 waves per SIMD on 256 CUs
  const double mfma_per_simd = 2.0 * iters * mfma_per_iter;
  const double ns_per_mfma = ms * 1e6 / mfma_per_simd;
  const double tflops = 2.0 * blocks * 4 * (double)iters * mfma_per_iter * macs_per_mfma / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms  %6.2f ns per MFMA per SIMD (= %5.1f cycles at 2.4 GHz)  %6.1f TFLOP/s\n", name, ms,
         ns_per_mfma, ns_per_mfma * 2.4, tflops);
  hipFree(out);
}

int main() {
  run<0>("4x4x1_16B x24, bare", 24, 256);
  run<1>("4x4x1_16B x24 + 16 VALU", 24, 256);
  run<2>("16x16x4 x6, bare", 6, 1024);
  run<3>("16x16x4 x8 + 16 VALU", 8, 1024);
  return 0;
}
