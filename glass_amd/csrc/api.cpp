// Library-wide pieces of the C ABI: version + per-thread error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/glass_hip.h"

namespace glass {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace glass

extern "C" int glass_version(void) { return GLASS_ABI_VERSION; }

extern "C" const char* glass_last_error_string(void) { return glass::g_err; }
