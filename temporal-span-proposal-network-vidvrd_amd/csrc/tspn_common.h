// Shared host-side helpers for the TSPN C-ABI (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "tspn_mi355x.h"

namespace tspn {

// thread-local last-error buffer (tspn_last_error)
char* err_buf();
constexpr int kErrBufLen = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), kErrBufLen, fmt, ap);
  va_end(ap);
  return code;
}

// (tspn_status.hip) TSPN_EDEVICE if a kernel of an earlier launch raised a fault through the device status block
int status_check(const char* what);

// Every launch entry ends here: the HIP launch error of THIS call, else a device fault raised by an EARLIER one
// (read from pinned host memory: no synchronisation).
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TSPN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return status_check(what);
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel.  One static
// LdsLimit per launch site remembers, per device ordinal, the largest limit already set there, so the
// attribute call is paid once per (kernel, device) — not once per thread, which left every device but
// the first one a thread used without the raised limit.
struct LdsLimit {
  static constexpr int kMaxDevices = 64;
  std::atomic<size_t> set[kMaxDevices] = {};
  int ensure(const void* fn, size_t bytes, const char* what) {
    int dev = -1;
    (void)hipGetDevice(&dev);
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && set[dev].load(std::memory_order_relaxed) >= bytes) return TSPN_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
      return fail(TSPN_ELAUNCH, "%s: hipFuncSetAttribute(%zu bytes of LDS): %s", what, bytes,
                  hipGetErrorString(e));
    if (tracked) set[dev].store(bytes, std::memory_order_relaxed);
    return TSPN_OK;
  }
};

// Internal (not exported) forms with explicit row strides, shared between translation units.
// y[b][m][ldy]: the fused driver pads rows of the tracklet projections to a multiple of 4 frames so
// that the pair stage can stage them with 16-byte LDS-DMA pieces.
int conv3_tc_direct(const float* x, int64_t B, int64_t T, int64_t Cin, const float* packed, int64_t M,
                    const float* bias, int relu, float* y, int64_t ldy, void* stream);
// Winograd F(6,3) (tspn_wino63.hip): input transform V = B^T d as its own pass into `workspace`, then the
// MFMA contraction on fragment-major weights; Cin % 32 == 0, M % 32 == 0
bool wino63_supported(int64_t Cin, int64_t M);
size_t wino63_workspace_bytes(int64_t B, int64_t T, int64_t Cin);
int wino63_input_transform(const float* x, int64_t B, int64_t T, int64_t Cin, void* workspace,
                           size_t workspace_bytes, void* stream, uint64_t* hot = nullptr);
int wino63_contract(const void* workspace, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                    const float* bias, int relu, float* y, int64_t ldy, void* stream);
int conv3_tc_wino63(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag, int64_t M,
                    const float* bias, int relu, float* y, int64_t ldy, void* workspace,
                    size_t workspace_bytes, void* stream);
int heads_pairgrid(const float* y, int64_t ldt, int64_t B, int64_t N, int64_t C, int64_t T,
                   const float* Wh, const float* bh, int64_t H, float* out, void* stream, float* Wp12 = nullptr);
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
int linear(const float* x, int64_t P, int64_t F, int64_t ldx, const float* W, int64_t ldw,
           const float* b, int64_t K, float* out, int apply_sigmoid, void* workspace,
           size_t workspace_bytes, void* stream);
size_t pair_predicate_workspace_bytes(int64_t NT, int64_t D, int64_t K);
int pair_predicate(const float* fbar, int64_t NT, int64_t D, const int64_t* pairs, int64_t P,
                   const float* cls_w, const float* cls_b, int64_t K, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);

}  // namespace tspn

#define TSPN_REQUIRE(cond, code, ...)                 \
  do {                                                \
    if (!(cond)) return tspn::fail(code, __VA_ARGS__); \
  } while (0)

#define TSPN_STREAM(s) reinterpret_cast<hipStream_t>(s)
