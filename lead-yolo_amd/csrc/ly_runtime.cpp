// Error reporting and version for the C-ABI (include/lead_yolo_hip.h).  Host-only translation unit.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void ly_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ly_last_error(void) { return g_err; }

extern "C" int ly_abi_version(void) { return 2; }
