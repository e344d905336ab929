// Probe: do fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2 / 4x4x1_16B) and plain fp32 VALU overlap on one SIMD (gfx950)?
// 512-thread workgroups, one per CU: waves w and w+4 share a SIMD.  Waves 0-3 run `nm` MFMAs, waves 4-7 run `nv` VALU
// FMAs (independent chains).  If the pipes overlap, T(mixed) ~ max(T(mfma only), T(valu only)); if fp32 MFMA executes
// on the vector ALUs, T(mixed) ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int VK>
__global__ __launch_bounds__(512, 1) void probe(float* out, int nm, int nv, float seed) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (KIND == 0) {
      f32x4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
      float a = seed + threadIdx.x, b = seed * 2.f;
      for (int it = 0; it < nm; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else if (KIND == 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      float a = seed + threadIdx.x, b = seed * 2.f;
      for (int it = 0; it < nm; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
    } else {
      f32x4 acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
      float a = seed + threadIdx.x, b = seed * 2.f;
      for (int it = 0; it < nm; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 1, 0);
      for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    }
  } else {
    float c[16];
    for (int i = 0; i < 16; ++i) c[i] = seed + i + threadIdx.x;
    const float m = seed * 0.999f, d = seed * 0.001f;
    for (int it = 0; it < nv; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (VK == 0) c[i] = __builtin_fmaf(c[i], m, d);
        if (VK == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c[i]) : "v"(d));
        if (VK == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(c[i]) : "v"(d));
        if (VK == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(d));
        if (VK == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c[i]) : "v"(m));
        if (VK == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(c[i]) : "v"(m));
        if (VK == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(c[i]) : "v"(m));
      }
    for (int i = 0; i < 16; ++i) r += c[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int VK>
float run(int nm, int nv) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<KIND, VK><<<256, 512>>>(out, nm / 10 + 1, nv / 10 + 1, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<KIND, VK><<<256, 512>>>(out, nm, nv, 1.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms;
}

template <int KIND, int VK>
void one(const char* mname, const char* vname, int nm) {
  const float tm = run<KIND, VK>(nm, 0);
  const int nv = 80000;
  const float tv = run<KIND, VK>(0, nv);
  const int nv2 = (int)(nv * tm / tv);
  const float tv2 = run<KIND, VK>(0, nv2);
  const float tx = run<KIND, VK>(nm, nv2);
  printf("%-24s + %-18s mfma %6.3f | valu %6.3f | mixed %6.3f ms -> overlap %3.0f %%\n", mname, vname, tm, tv2, tx,
         100.0 * (tm + tv2 - tx) / (tm < tv2 ? tm : tv2));
}

int main() {
  one<0, 0>("16x16x4 f32", "v_fma_f32", 20000);
  one<1, 0>("32x32x2 f32", "v_fma_f32", 20000);
  one<2, 0>("4x4x1_16B f32", "v_fma_f32", 40000);
  one<1, 1>("32x32x2 f32", "v_add_f32", 20000);
  one<1, 2>("32x32x2 f32", "v_max_f32", 20000);
  one<1, 3>("32x32x2 f32", "v_add_u32", 20000);
  one<1, 4>("32x32x2 f32", "v_and_b32", 20000);
  one<1, 5>("32x32x2 f32", "v_mov_b32", 20000);
  one<1, 6>("32x32x2 f32", "v_cvt_pk_bf16_f32", 20000);
  one<0, 2>("16x16x4 f32", "v_max_f32", 20000);
  one<0, 3>("16x16x4 f32", "v_add_u32", 20000);
  return 0;
}
