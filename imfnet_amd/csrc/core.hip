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

void *imf_event_create(void) {
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? (void *)e : nullptr;
}
void imf_event_destroy(void *ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
int imf_event_record(void *ev, void *stream) {
  IMF_REQUIRE(ev, "imf_event_record: null event");
  IMF_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return IMF_OK;
}
/* Streams owned by the library's callers but created here: a framework's stream pool may hand the same stream out
 * twice (torch.cuda.Stream() wraps around after 32), and imf_fragment_forward needs three DISTINCT ones. */
void *imf_stream_create(void) {
  hipStream_t s = nullptr;
  return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? (void *)s : nullptr;
}
void imf_stream_destroy(void *s) {
  if (s) (void)hipStreamDestroy((hipStream_t)s);
}
float imf_event_elapsed_ms(void *b, void *e) {
  float ms = -1.f;
  if (!b || !e || hipEventElapsedTime(&ms, (hipEvent_t)b, (hipEvent_t)e) != hipSuccess) return -1.f;
  return ms;
}
}
