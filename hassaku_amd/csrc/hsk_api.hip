// hsk_api.hip -- error plumbing, version and device query of the C ABI.
#include "hsk_common.h"

#include <stdarg.h>

static thread_local char g_hsk_error[512] = "";

void hsk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_hsk_error, sizeof(g_hsk_error), fmt, ap);
  va_end(ap);
}

extern "C" int hsk_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* hsk_last_error(void) { return g_hsk_error; }

extern "C" int hsk_device_info(int32_t* cu_count, int32_t* wave_size, char* arch, int32_t arch_len) {
  int dev = 0;
  HSK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HSK_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return HSK_OK;
}
