// How fast can a CU pull the x operand of a 1x1 conv (pixel pitch `pitch` bytes, a 64-channel chunk = one 128-byte line per
// pixel) through its L1, as a function of WHICH 16 bytes each lane of a wave-instruction asks for?
//   pattern 0: the conv kernels' form since round 2: lane = pixel, the instruction reads 16 B of 64 different lines; eight
//              instructions (of two waves) sweep a line
//   pattern 1: line-major: lane l reads piece (l & 7) of pixel (l >> 3): 8 whole lines per instruction
//   pattern 2: piece-major: lane l reads piece (l >> 3) of pixel (l & 7): the same 8 whole lines, lanes of a line 8 apart
//   pattern 3: half lines: lane l reads piece (l & 3) + 4 h of pixel (l >> 2): 16 half lines per instruction
//   pattern 4: line-major with the pieces of a line XOR-swizzled (piece (l & 7) ^ (l >> 3) ^ (p & 1)): what a bank-conflict-free
//              pixel-major LDS image needs; a quad of lanes still covers one 64-byte half line, in permuted order
// form 0: buffer_load_dwordx4 ... lds (DMA into a ring of DEPTH 16-KB stages, counted vmcnt), form 1: buffer loads into registers.
// Workgroups of 256 threads, two per CU, each walks 128-pixel tiles chunk by chunk (16 chunks of a 1024-channel pixel) like
// conv2d_nhwc_bf16_kernel does; `hot` = every workgroup re-reads one small window (L2-resident).
//   hipcc --offload-arch=gfx950 -O3 -o tcp_line_coalesce_probe tcp_line_coalesce_probe.hip && ./tcp_line_coalesce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int FORM, int DEPTH>
__global__ __launch_bounds__(256, 2) void probe(const char* __restrict__ x, long long npix, int pitch, int chunks, int tiles_per_wg,
                                                int hot, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char smem[DEPTH * 16384];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 r[4] = {};
  float s = 0.f;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long long tile = hot ? (blockIdx.x & 7) : (long long)t * gridDim.x + blockIdx.x;
    const long long p0 = tile * 128;
    if (p0 + 128 > npix) break;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + p0 * pitch), 0, (int)0x7fffffff, 0x00020000);
    unsigned off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int pix, byte;
      if (PAT == 0) { pix = 64 * (wave & 1) + lane; byte = 16 * (wave >> 1) + 32 * p; }
      else if (PAT == 1) { pix = 8 * (4 * wave + p) + (lane >> 3); byte = 16 * (lane & 7); }
      else if (PAT == 2) { pix = 8 * (4 * wave + p) + (lane & 7); byte = 16 * (lane >> 3); }
      else if (PAT == 3) { pix = 16 * (2 * wave + (p >> 1)) + (lane >> 2); byte = 16 * (lane & 3) + 64 * (p & 1); }
      else { pix = 8 * (4 * wave + p) + (lane >> 3); byte = 16 * ((lane & 7) ^ (lane >> 3) ^ (p & 1)); }
      off[p] = (unsigned)(pix * pitch + byte);
    }
    auto issue = [&](int c) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (FORM == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (c % DEPTH) * 16384 + (4 * wave + p) * 1024),
                                                   16, (int)off[p], c * 128, 0, 0);
        else
          r[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[p], c * 128, 0));
      }
    };
    if (FORM == 0) {
      for (int c = 0; c < DEPTH - 1 && c < chunks; ++c) issue(c);
      for (int c = 0; c < chunks; ++c) {
        if (c + DEPTH - 1 < chunks) issue(c + DEPTH - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      for (int c = 0; c < chunks; ++c) {
        issue(c);
#pragma unroll
        for (int p = 0; p < 4; ++p) s += r[p][0];          // (the compiler keeps several chunks in flight by unrolling)
      }
    }
  }
  if (FORM == 0) s = reinterpret_cast<float*>(smem)[threadIdx.x];
  if (s == 123.456f) sink[0] = s;
}

template <int PAT, int FORM, int DEPTH>
void run(const char* name, const char* x, long long npix, int pitch, int hot, float* sink) {
  const int chunks = pitch / 128, grid = 512;
  const int tiles_per_wg = hot ? 4 : (int)((npix / 128 + grid - 1) / grid);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<PAT, FORM, DEPTH>), dim3(grid), dim3(256), 0, 0, x, npix, pitch, chunks, tiles_per_wg, hot, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double tiles = hot ? 4.0 * grid : (double)(npix / 128);
  const double bytes = tiles * 128 * pitch;
  printf("%-44s pitch %5d %s: %8.1f us  %7.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, pitch, hot ? "hot " : "hbm ", best * 1e3,
         bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  const long long npix = 64800LL * 2;       // 36 res4 frames
  char* x; float* sink;
  CHECK(hipMalloc(&x, (size_t)npix * 2048 + 4096));
  CHECK(hipMemset(x, 1, (size_t)npix * 2048 + 4096));
  CHECK(hipMalloc(&sink, 64));
  for (int hot = 0; hot < 2; ++hot)
    for (int pitch : {2048, 512}) {
      run<0, 0, 2>("lane = pixel (shipped), DMA, 2 stages", x, npix, pitch, hot, sink);
      run<0, 0, 4>("lane = pixel (shipped), DMA, 4 stages", x, npix, pitch, hot, sink);
      run<1, 0, 2>("line-major, DMA, 2 stages", x, npix, pitch, hot, sink);
      run<1, 0, 4>("line-major, DMA, 4 stages", x, npix, pitch, hot, sink);
      run<2, 0, 2>("piece-major 8 lines, DMA, 2 stages", x, npix, pitch, hot, sink);
      run<2, 0, 4>("piece-major 8 lines, DMA, 4 stages", x, npix, pitch, hot, sink);
      run<3, 0, 4>("half lines, DMA, 4 stages", x, npix, pitch, hot, sink);
      run<4, 0, 2>("line-major swizzled, DMA, 2 stages", x, npix, pitch, hot, sink);
      run<4, 1, 1>("line-major swizzled, registers", x, npix, pitch, hot, sink);
      run<0, 1, 1>("lane = pixel, registers", x, npix, pitch, hot, sink);
      run<1, 1, 1>("line-major, registers", x, npix, pitch, hot, sink);
      run<2, 1, 1>("piece-major 8 lines, registers", x, npix, pitch, hot, sink);
    }
  return 0;
}
