#include "common.h"

namespace gdr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gdr

extern "C" const char* gdr_last_error(void) { return gdr::g_err; }
extern "C" int gdr_abi_version(void) { return 1; }
