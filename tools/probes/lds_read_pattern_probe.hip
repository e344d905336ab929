// ds_read_b128 throughput of a CU by address pattern (gfx950): how many clocks does a wave-wide 16-byte-per-lane read cost
// when 4 / 8 waves issue them back to back?  Patterns (lane = 32 kh + li):
//   0  linear: lane l -> 16 l
//   1  the conv kernels' B-operand image [group][132 slots][16 B]: (kh * 132 + li + rb) * 16, rb = 0 / 1 / 2 (tap offsets)
//   2  the same with 128 slots per group (no padding)
//   3  pixel-major, XOR-swizzled (round 5, tap chunks): li * 128 + 16 * ((kh) ^ f(li)), f = (li & 7) ^ ((li >> 3) & 1)
//   4  pixel-major without the swizzle: li * 128 + 16 * kh
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_pattern_probe lds_read_pattern_probe.hip && ./lds_read_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int PAT, int RB>
__global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(128))) char smem[65536];
  const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  unsigned off;
  if (PAT == 0) off = lane * 16;
  else if (PAT == 1) off = (kh * 132 + li + RB) * 16;
  else if (PAT == 2) off = (kh * 128 + li + RB) * 16;
  else if (PAT == 3) off = li * 128 + 16 * (kh ^ ((li & 7) ^ ((li >> 3) & 1)));
  else off = li * 128 + 16 * kh;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    f32x4 r[8];
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + off;
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a), "n"(k * 4352));   // (other rows of the image)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += r[k];
  }
  const unsigned long long t1 = now();
  if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int PAT, int RB>
void run(const char* name, int waves, unsigned long long* out, float* sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<PAT, RB>), dim3(1), dim3(64 * waves), 0, 0, out, sink, iters);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[8];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  double mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
  printf("%-58s %d waves: %6.1f ticks per wave-read, %6.1f B per tick and CU\n", name, waves, mx / (iters * 8.0),
         waves * iters * 8.0 * 1024 / mx);
}

int main() {
  unsigned long long* out; float* sink;
  CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&sink, 64));
  for (int waves : {4, 8}) {
    run<0, 0>("linear", waves, out, sink);
    run<1, 0>("[group][132 slots], tap offset 0", waves, out, sink);
    run<1, 1>("[group][132 slots], tap offset 1", waves, out, sink);
    run<1, 2>("[group][132 slots], tap offset 2", waves, out, sink);
    run<2, 0>("[group][128 slots], tap offset 0", waves, out, sink);
    run<2, 1>("[group][128 slots], tap offset 1", waves, out, sink);
    run<3, 0>("pixel-major, XOR-swizzled", waves, out, sink);
    run<4, 0>("pixel-major, no swizzle", waves, out, sink);
  }
  return 0;
}
