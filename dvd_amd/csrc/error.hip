#include "common.h"

namespace dvd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dvd

extern "C" const char* dvd_last_error(void) { return dvd::g_err; }
extern "C" int dvd_version(void) { return 1000; }
