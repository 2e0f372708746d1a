// ABI bookkeeping of libpcp_hip.so.
#include "pcp_common.h"

extern "C" int pcp_abi_version(void) { return PCP_ABI_VERSION; }

extern "C" const char *pcp_status_string(int status) {
  switch (status) {
    case PCP_OK: return "ok";
    case PCP_ERR_ARG: return "bad argument";
    case PCP_ERR_WORKSPACE: return "workspace too small";
    case PCP_ERR_LAUNCH: return "kernel launch failed";
    case PCP_ERR_UNSUPPORTED: return "unsupported shape for this build";
    default: return "unknown status";
  }
}

// ---- the option table (include/pcp_hip.h: the library's only process-wide mutable state) and the per-device CU count cache ---------------
#include <atomic>

namespace {
std::atomic<long long> g_option[PCP_OPT_COUNT] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
static_assert(PCP_OPT_COUNT == 7, "one initialiser per option");
constexpr int MAX_DEVICES = 64;
std::atomic<int> g_cus[MAX_DEVICES];          // 0 = not asked yet
}  // namespace

extern "C" int pcp_set_option(int32_t option, int64_t value) {
  if (option < 0 || option >= PCP_OPT_COUNT) return PCP_ERR_ARG;
  g_option[option].store(value < 0 ? -1 : (long long)value, std::memory_order_relaxed);
  return PCP_OK;
}

extern "C" int64_t pcp_get_option(int32_t option) {
  if (option < 0 || option >= PCP_OPT_COUNT) return -1;
  return g_option[option].load(std::memory_order_relaxed);
}

long long pcp_option(int option, long long builtin) {
  const long long v = g_option[option].load(std::memory_order_relaxed);
  return v < 0 ? builtin : v;
}

// CUs of the device the calling thread launches on, cached by device ordinal (a process may drive several devices)
int pcp_current_device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  int v = g_cus[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  g_cus[dev].store(v, std::memory_order_relaxed);
  return v;
}
