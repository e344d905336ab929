// What does `buffer_load_dwordx4 ... offen lds` write into LDS for a lane whose offset is beyond the descriptor's range?
// (raw buffer, stride 0, num_records = bytes).  Answer printed: the 16 bytes of an in-range lane, of an out-of-range lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, int bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xEEEEEEEEu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u + threadIdx.x * 16 : threadIdx.x * 16;   // odd lanes: far out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *src, *out;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x10000000u + i;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 1024, out);
  std::vector<unsigned> o(256);
  hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 3, 62, 63})
    printf("lane %2d (%s): %08x %08x %08x %08x\n", l, (l & 1) ? "out of range" : "in range", o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  return 0;
}
