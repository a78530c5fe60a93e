// Error state + library-level entry points of libtrajdiff_hip.so.
#include "tdx_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void tdx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tdx_last_error(void) { return g_err; }
extern "C" int tdx_version(void) { return 300; }

extern "C" int tdx_device_ok(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    TDX_FAIL(TDX_ERR_HIP, "tdx_device_ok: no HIP device");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    TDX_FAIL(TDX_ERR_UNSUPPORTED, "tdx_device_ok: device is %s, this library is built for gfx950 only", prop.gcnArchName);
  return TDX_OK;
}
