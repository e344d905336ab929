// Version + error reporting for the TSPN C-ABI.
#include "tspn_common.h"

namespace tspn {
char* err_buf() {
  static thread_local char buf[kErrBufLen] = {0};
  return buf;
}
}  // namespace tspn

extern "C" int tspn_version(void) { return TSPN_ABI_VERSION; }
extern "C" size_t tspn_fused_desc_size(void) { return sizeof(tspn_fused_desc); }
extern "C" size_t tspn_fused_bf16_desc_size(void) { return sizeof(tspn_fused_bf16_desc); }

extern "C" const char* tspn_last_error(void) { return tspn::err_buf(); }

extern "C" const char* tspn_error_string(int code) {
  switch (code) {
    case TSPN_OK: return "ok";
    case TSPN_EINVAL: return "invalid argument";
    case TSPN_EUNSUPPORTED: return "unsupported shape";
    case TSPN_EWORKSPACE: return "workspace too small";
    case TSPN_ELAUNCH: return "HIP launch error";
    case TSPN_EDEVICE: return "device fault raised by an earlier launch";
    default: return "unknown error";
  }
}
