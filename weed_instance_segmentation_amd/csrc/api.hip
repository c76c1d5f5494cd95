// libwm2f: version / error reporting of the C ABI (include/wm2f.h).
#include "common.h"

namespace wm2f {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace wm2f

extern "C" int wm2f_version(void) { return WM2F_VERSION; }

extern "C" const char* wm2f_last_error(void) { return wm2f::g_err; }
