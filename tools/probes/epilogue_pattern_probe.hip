// The epilogue of the fused bottleneck kernels: every lane adds a residual to 16 consecutive channels (32 bytes) of ONE pixel
// and stores 32 bytes -- a wave-instruction touches 16 bytes of 64 different 128-byte lines.  Does the L1 / TA care?
//   form 0: as shipped -- lane (li, kh): pixel li, bytes 64 b + 32 kh + {0, 16} of its 2048-byte row, blocks b = 0..31 in pairs
//   form 1: line-major -- lane l: pixel (l >> 3), piece (l & 7) of a 128-byte line; eight lines per instruction
// Each workgroup (256 threads, two per CU) walks 128-pixel tiles of a [npix][1024] bf16 map: read residual, add, store to out.
//   hipcc --offload-arch=gfx950 -O3 -o epilogue_pattern_probe epilogue_pattern_probe.hip && ./epilogue_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int MODE>      // MODE 0: load + store, 1: loads only, 2: stores only
__global__ __launch_bounds__(256, 2) void probe(const char* __restrict__ res, char* __restrict__ out, long long npix, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const long long ntiles = npix / 128;
  f32x4 acc = {1.f, 2.f, 3.f, 4.f};
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const char* rb = res + tile * 128 * 2048;
    char* ob = out + tile * 128 * 2048;
    // a wave owns 256 channels (512 bytes of every row) x 128 pixels, like a wave of the tail's expand phase
    if (FORM == 0) {
#pragma unroll 1
      for (int pb = 0; pb < 4; ++pb) {                  // pixel block of 32
#pragma unroll
        for (int g = 0; g < 4; ++g) {                   // a 128-byte line = two 32-channel blocks, stored back to back
          f32x4 r[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const size_t o = (size_t)(pb * 32 + li) * 2048 + wave * 512 + g * 128 + (h >> 1) * 64 + 32 * kh + 16 * (h & 1);
            if (MODE != 2) r[h] = *reinterpret_cast<const f32x4*>(rb + o); else r[h] = acc;
          }
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const size_t o = (size_t)(pb * 32 + li) * 2048 + wave * 512 + g * 128 + (h >> 1) * 64 + 32 * kh + 16 * (h & 1);
            if (MODE != 1) *reinterpret_cast<f32x4*>(ob + o) = r[h] + acc; else acc += r[h];
          }
        }
      }
    } else {
#pragma unroll 1
      for (int pb = 0; pb < 4; ++pb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 r[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {                 // 8 pixels x 128 bytes per instruction
            const size_t o = (size_t)(pb * 32 + h * 8 + (lane >> 3)) * 2048 + wave * 512 + g * 128 + 16 * (lane & 7);
            if (MODE != 2) r[h] = *reinterpret_cast<const f32x4*>(rb + o); else r[h] = acc;
          }
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const size_t o = (size_t)(pb * 32 + h * 8 + (lane >> 3)) * 2048 + wave * 512 + g * 128 + 16 * (lane & 7);
            if (MODE != 1) *reinterpret_cast<f32x4*>(ob + o) = r[h] + acc; else acc += r[h];
          }
        }
      }
    }
  }
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int FORM, int MODE>
void run(const char* name, const char* res, char* out, long long npix, float* sink) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<FORM, MODE>), dim3(512), dim3(256), 0, 0, res, out, npix, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)(npix / 128) * 128 * 2048 * (MODE == 0 ? 2 : 1);
  printf("%-52s %lld pixels: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, npix, best * 1e3, bytes / best / 1e9,
         bytes / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  char *res, *out; float* sink;
  const long long maxpix = 64800LL * 2;
  CHECK(hipMalloc(&res, (size_t)maxpix * 2048));
  CHECK(hipMalloc(&out, (size_t)maxpix * 2048));
  CHECK(hipMemset(res, 0, (size_t)maxpix * 2048));
  CHECK(hipMalloc(&sink, 64));
  for (long long npix : {64800LL, 129600LL}) {
    run<0, 0>("lane = pixel, 32 B per lane (shipped): load + store", res, out, npix, sink);
    run<1, 0>("line-major: load + store", res, out, npix, sink);
    run<0, 1>("lane = pixel: loads only", res, out, npix, sink);
    run<1, 1>("line-major: loads only", res, out, npix, sink);
    run<0, 2>("lane = pixel: stores only", res, out, npix, sink);
    run<1, 2>("line-major: stores only", res, out, npix, sink);
  }
  return 0;
}
