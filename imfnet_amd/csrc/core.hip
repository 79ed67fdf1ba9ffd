// Library-level entry points: version, thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace imf {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace imf

extern "C" {
int imf_version(void) { return 100; }   /* 0.1.0 */
const char *imf_last_error(void) { return imf::g_err; }
}
