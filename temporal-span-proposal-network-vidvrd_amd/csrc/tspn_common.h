// Shared host-side helpers for the TSPN C-ABI (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

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

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TSPN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TSPN_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace tspn

#define TSPN_REQUIRE(cond, code, ...)                 \
  do {                                                \
    if (!(cond)) return tspn::fail(code, __VA_ARGS__); \
  } while (0)

#define TSPN_STREAM(s) reinterpret_cast<hipStream_t>(s)
