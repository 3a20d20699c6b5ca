// Error reporting + version for the C ABI (no device code here).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/clip_event_hip.h"

static thread_local char g_err[512] = "";

void ce_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ce_last_error(void) { return g_err; }
extern "C" int ce_version(void) { return 1; }
