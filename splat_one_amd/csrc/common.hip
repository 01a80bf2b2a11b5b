// common.hip -- error channel and device queries of libsplat_one_amd.so
#include "so_common.hpp"

namespace so {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace so

extern "C" int so_abi_version(void) { return SO_ABI_VERSION; }
extern "C" const char *so_last_error(void) { return so::g_err; }
extern "C" int so_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}
