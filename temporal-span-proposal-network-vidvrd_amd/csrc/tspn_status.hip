// Device status block: host side (attach / read / clear), the fault check every launch entry ends with, and a self-test
// kernel that raises through the same device code the role-split res4 tail uses (tspn_status.h).
#include "tspn_status.h"

#include "tspn_common.h"

namespace {

constexpr int kMaxDevices = 64;
std::atomic<int32_t*> g_host[kMaxDevices] = {};
std::atomic<int32_t*> g_dev[kMaxDevices] = {};

int current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  return (dev >= 0 && dev < kMaxDevices) ? dev : -1;
}

// One wave waits (with a short bound) for an LDS counter nobody sets: the wait gives up, raises, ends.
__global__ __launch_bounds__(64) void status_selftest_kernel(int32_t* status, int32_t* reached_end) {
  __shared__ int flag[4];
  if (threadIdx.x < 4) flag[threadIdx.x] = 0;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) int*)flag;
  tspn_dev::flag_wait<256>(addr, 1, status, 0x5e1f);
  if (threadIdx.x == 0 && reached_end) *reached_end = 1;   // never: the wave has ended inside the wait
}

}  // namespace

int32_t* tspn::status_device_ptr() {
  const int dev = current_device();
  return dev < 0 ? nullptr : g_dev[dev].load(std::memory_order_acquire);
}

int tspn::status_check(const char* what) {
  const int dev = current_device();
  if (dev < 0) return TSPN_OK;
  const int32_t* host = g_host[dev].load(std::memory_order_acquire);
  if (!host) return TSPN_OK;
  const int32_t f = __atomic_load_n(host + TSPN_STATUS_FAULT, __ATOMIC_RELAXED);
  if (f == 0) return TSPN_OK;
  return tspn::fail(TSPN_EDEVICE,
                    "%s: device %d reported fault 0x%x (%s%s; info 0x%x) in an EARLIER launch of this library: results "
                    "produced since are not to be trusted; tspn_status_clear() re-arms",
                    what, dev, (unsigned)f, (f & TSPN_FAULT_HANDOVER) ? "an LDS hand-over between waves timed out" : "",
                    (f & ~TSPN_FAULT_HANDOVER) ? " + unknown bits" : "",
                    (unsigned)__atomic_load_n(host + TSPN_STATUS_FAULT_INFO, __ATOMIC_RELAXED));
}

extern "C" int tspn_status_attach(int32_t* host_words) {
  const int dev = current_device();
  TSPN_REQUIRE(dev >= 0, TSPN_ELAUNCH, "tspn_status_attach: no current HIP device");
  if (!host_words) {
    g_dev[dev].store(nullptr, std::memory_order_release);
    g_host[dev].store(nullptr, std::memory_order_release);
    return TSPN_OK;
  }
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(host_words) & 63) == 0, TSPN_EINVAL,
               "tspn_status_attach: the block must be 64-byte aligned");
  void* dptr = nullptr;
  hipError_t e = hipHostGetDevicePointer(&dptr, host_words, 0);
  if (e != hipSuccess || !dptr) {
    (void)hipGetLastError();
    return tspn::fail(TSPN_EINVAL,
                      "tspn_status_attach: %p is not pinned, device-mapped host memory (hipHostMalloc / "
                      "torch .pin_memory()): %s",
                      (void*)host_words, hipGetErrorString(e));
  }
  g_host[dev].store(host_words, std::memory_order_release);
  g_dev[dev].store(static_cast<int32_t*>(dptr), std::memory_order_release);
  return TSPN_OK;
}

extern "C" int tspn_status_fault(void) {
  const int dev = current_device();
  const int32_t* host = dev < 0 ? nullptr : g_host[dev].load(std::memory_order_acquire);
  return host ? __atomic_load_n(host + TSPN_STATUS_FAULT, __ATOMIC_RELAXED) : 0;
}

extern "C" int tspn_status_clear(void) {
  const int dev = current_device();
  int32_t* host = dev < 0 ? nullptr : g_host[dev].load(std::memory_order_acquire);
  if (!host) return TSPN_OK;
  __atomic_store_n(host + TSPN_STATUS_FAULT_INFO, 0, __ATOMIC_RELAXED);
  __atomic_store_n(host + TSPN_STATUS_FAULT, 0, __ATOMIC_RELEASE);
  return TSPN_OK;
}

extern "C" int tspn_status_selftest(int32_t* reached_end, void* stream) {
  const char* what = "tspn_status_selftest";
  int32_t* status = tspn::status_device_ptr();
  TSPN_REQUIRE(status, TSPN_EINVAL, "%s: no status block attached on this device (the kernel would trap)", what);
  hipLaunchKernelGGL(status_selftest_kernel, dim3(1), dim3(64), 0, TSPN_STREAM(stream), status, reached_end);
  hipError_t e = hipGetLastError();   // (not check_launch: this launch must not trip over the fault it is about to raise)
  if (e != hipSuccess) return tspn::fail(TSPN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TSPN_OK;
}
