// Error string + version for the C ABI.
#include <stdarg.h>

#include "common.h"

namespace avd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace avd

extern "C" const char* avd_last_error(void) { return avd::g_err; }
extern "C" int avd_version(void) { return 1; }
// 1 in the diagnostic build (-DAVD_DIAG: the AVD_* environment switches exist), 0 in the shipped library
extern "C" int avd_diagnostics_enabled(void) {
#ifdef AVD_DIAG
    return 1;
#else
    return 0;
#endif
}
