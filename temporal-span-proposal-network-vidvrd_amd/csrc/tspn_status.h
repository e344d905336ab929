// Device status block (round 6): the one channel by which a kernel can tell the host that something went wrong
// WITHOUT a synchronisation.  TSPN_STATUS_WORDS int32 words in pinned, device-mapped HOST memory that the caller
// allocates and attaches once per device (tspn_status_attach); kernels get its device address as an argument, write it
// with system-scope atomics, the host reads it with plain loads.  Word TSPN_STATUS_FAULT != 0 makes every later launch
// entry of the library on that device return TSPN_EDEVICE (tspn::check_launch) until tspn_status_clear().
//
// Who raises: the bounded LDS hand-over waits of tail_io_bf16_kernel / bottleneck_pipe_bf16_kernel (a wave that gives up
// raises and ENDS -- it never continues into the data path with a buffer it did not receive), tspn_status_selftest.
// Who records (not a fault): conv3_spot_check_kernel, the a-posteriori accuracy guard of the F(6,3) temporal conv.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tspn_mi355x.h"

namespace tspn {
// device address of the status block attached for the CURRENT device, or nullptr
int32_t* status_device_ptr();
// TSPN_EDEVICE (message in tspn_last_error) if the current device's fault word is set, else TSPN_OK
int status_check(const char* what);
}  // namespace tspn

namespace tspn_dev {

// ---- sub-pass counters in LDS.  A wave's LDS operations execute in order: a counter written behind the data is seen
// behind the data.
__device__ __forceinline__ void flag_set(unsigned addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Wait until the counter at LDS address `addr` has reached `target`.  The spin is BOUNDED (POLLS = 2^20: tens of
// milliseconds where a tile takes ~55 us): a hand-over that were ever lost must not become a wave that never ends and a
// GPU that has to be reset.  A wave that gives up ORs TSPN_FAULT_HANDOVER into the status block (`info` beside it) and
// ENDS; without a block it traps.
// ONE asm block, the give-up path included: a C++ loop -- or just a C++ `if` on the outcome -- at the call sites of
// tail_io_bf16_kernel makes hipcc spill registers of the 3 600-instruction straight-line code around it (the loop ~430,
// the `if` 8: its accumulators and operand rings are live across every call site, and the kernel sits at 254 of 256).
// The give-up path overwrites the address register: nothing runs behind it.
template <int POLLS = (1 << 20)>
__device__ __forceinline__ void flag_wait(unsigned addr, int target, int32_t* status, int info) {
  static_assert(TSPN_STATUS_FAULT == 0 && TSPN_STATUS_FAULT_INFO == 1, "the offsets below");
  int v, sv, n;
  asm volatile(
      "s_mov_b32 %2, %5\n\t"
      "1:\n\t"
      "ds_read_b32 %0, %3\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %1, %0\n\t"
      "s_cmp_ge_i32 %1, %4\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sub_u32 %2, %2, 1\n\t"
      "s_cmp_eq_u32 %2, 0\n\t"
      "s_cbranch_scc1 3f\n\t"
      "s_sleep 1\n\t"
      "s_branch 1b\n\t"
      "3:\n\t"                                        // gave up
      "s_cmp_eq_u64 %6, 0\n\t"
      "s_cbranch_scc1 4f\n\t"
      "v_mov_b32 %0, 0\n\t"
      "v_mov_b32 %3, %8\n\t"
      "global_atomic_or %0, %3, %6 sc1\n\t"          // system scope: the block is host memory
      "v_mov_b32 %3, %7\n\t"
      "global_store_dword %0, %3, %6 offset:4 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_endpgm\n\t"
      "4:\n\t"
      "s_trap 2\n\t"
      "s_endpgm\n\t"
      "2:"
      : "=&v"(v), "=&s"(sv), "=&s"(n)
      : "v"(addr), "s"(target), "n"(POLLS), "s"(status), "s"(info), "n"(TSPN_FAULT_HANDOVER)
      : "memory", "scc");
}

}  // namespace tspn_dev
