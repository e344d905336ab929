// What does ISSUING an LDS-DMA piece (1 KiB per wave-instruction) cost its wave on gfx950, and does the addressing form matter?
//   mode 0: global_load_lds_dwordx4 (per-lane 64-bit address)      mode 1: buffer_load_dwordx4 ... offen lds (SGPR descriptor +
//   per-lane 32-bit offset).  Eight waves per workgroup, one workgroup per CU, every wave issues `pieces` pieces per iteration
// into a ring of `depth` iterations in LDS (counted vmcnt), optionally `mfmas` v_mfma_f32_32x32x16_bf16 per iteration behind
// them; the source is a buffer of `src_mb` MB walked linearly (small: L2 / Infinity Cache resident).  s_memtime brackets the
// issue block.   hipcc --offload-arch=gfx950 -O3 -o lds_dma_issue_probe lds_dma_issue_probe.hip && ./lds_dma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int MODE, int PIECES, int DEPTH, int MFMAS>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ src, size_t src_bytes, int iters,
                                                unsigned long long* __restrict__ out, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every workgroup walks its own window of the buffer; a piece = 64 lanes x 16 B contiguous
  const size_t stride_it = (size_t)8 * PIECES * 1024;             // bytes per iteration and workgroup
  size_t off = ((size_t)blockIdx.x * 977 * stride_it) % src_bytes;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
  f32x16 acc = {};
  bf16x8 a = {}, b = {};
  a[0] = (__bf16)(float)lane; b[0] = (__bf16)1.f;
  unsigned long long t_issue = 0, t_all0 = 0;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 regs[PIECES] = {};
  auto issue = [&](int it) {
    char* dst = smem + ((it % DEPTH) * 8 + wave) * PIECES * 1024;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      size_t o = off + ((size_t)(wave * PIECES + p)) * 1024 + lane * 16;
      if (o >= src_bytes) o -= src_bytes;
      if (MODE == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
      else if (MODE == 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16,
                                                 (int)o, 0, 0, 0);
      else if (MODE == 2)          // register-returning loads (the weight path of the conv kernels): SGPR base + 32-bit lane offset
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(regs[p]) : "v"((unsigned)o), "s"(src) : "memory");
      else
        regs[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0));
    }
    off += stride_it;
    if (off >= src_bytes) off -= src_bytes;
  };
  if (MODE < 2)        // (register-returning loads have no ring: every load is waited for in its own iteration)
    for (int it = 0; it < DEPTH - 1; ++it) issue(it);
  __syncthreads();
  t_all0 = now();
  for (int it = DEPTH - 1; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = now();
    __builtin_amdgcn_sched_barrier(0);
    issue(it);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = now();
    __builtin_amdgcn_sched_barrier(0);
    t_issue += t1 - t0;
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 2) {     // register-returning loads: the destination registers stay reserved until this tied wait
#pragma unroll
      for (int p = 0; p < PIECES; ++p) asm volatile("s_waitcnt vmcnt(0)" : "+v"(regs[p]));
    } else {
      wait_vm<PIECES * (DEPTH - 1)>();
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t_all1 = now();
  wait_vm<0>();
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = t_issue;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = t_all1 - t_all0;
  }
  asm volatile("" : "+v"(regs[0]), "+v"(regs[PIECES - 1]));
  if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0] + smem[threadIdx.x] + regs[0][0] + regs[PIECES - 1][1];
}

template <int MODE, int PIECES, int DEPTH, int MFMAS>
void run(const char* name, const char* src, size_t bytes, int iters, unsigned long long* dout, float* sink) {
  const int nwg = 256;
  const size_t smem = (size_t)DEPTH * 8 * PIECES * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, PIECES, DEPTH, MFMAS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  std::vector<unsigned long long> h(nwg * 16);
  float best_ms = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, PIECES, DEPTH, MFMAS>), dim3(nwg), dim3(512), smem, 0, src, bytes, iters, dout, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best_ms) best_ms = ms;
  }
  CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double si = 0, sa = 0;
  for (int i = 0; i < nwg * 8; ++i) { si += h[2 * i]; sa += h[2 * i + 1]; }
  const double n_it = iters - (DEPTH - 1);
  const double per_piece = si / (nwg * 8) / n_it / PIECES - 40.0 / PIECES;   // minus the stamp's own ~40 cycles
  const double per_iter = sa / (nwg * 8) / n_it;
  const double tb = (double)nwg * 8 * PIECES * 1024 * iters / (best_ms * 1e-3) / 1e12;
  printf("%-44s src %4zu MB  pieces/iter %d depth %d mfma/iter %2d: issue %6.0f cyc/piece, iteration %6.0f cyc, %5.2f TB/s chip-wide\n",
         name, bytes >> 20, PIECES, DEPTH, MFMAS, per_piece, per_iter, tb);
}

int main() {
  float* sink; unsigned long long* dout;
  CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&dout, 256 * 16 * 8));
  for (size_t mb : {2, 64, 1024}) {
    char* src; const size_t bytes = mb << 20;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMemset(src, 1, bytes));
    const int iters = 400;
    run<0, 4, 4, 0>("global_load_lds_dwordx4", src, bytes, iters, dout, sink);
    run<1, 4, 4, 0>("buffer_load_dwordx4 offen lds", src, bytes, iters, dout, sink);
    run<0, 4, 4, 16>("global_load_lds_dwordx4 + 16 MFMA", src, bytes, iters, dout, sink);
    run<1, 4, 4, 16>("buffer_load_dwordx4 offen lds + 16 MFMA", src, bytes, iters, dout, sink);
    run<0, 2, 4, 16>("global_load_lds_dwordx4 + 16 MFMA", src, bytes, iters, dout, sink);
    run<1, 2, 4, 16>("buffer_load_dwordx4 offen lds + 16 MFMA", src, bytes, iters, dout, sink);
    run<2, 4, 2, 16>("global_load_dwordx4 saddr -> VGPR + 16 MFMA", src, bytes, iters, dout, sink);
    run<3, 4, 2, 16>("buffer_load_dwordx4 offen -> VGPR + 16 MFMA", src, bytes, iters, dout, sink);
    run<0, 4, 2, 16>("global_load_lds_dwordx4 + 16 MFMA", src, bytes, iters, dout, sink);
    run<0, 2, 8, 16>("global_load_lds_dwordx4 + 16 MFMA", src, bytes, iters, dout, sink);
    CHECK(hipFree(src));
  }
  return 0;
}
