// Error reporting and version for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/octa_hip.h"

static thread_local char g_err[512] = "";

void octa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int octa_version(void) { return OCTA_HIP_ABI_VERSION; }
extern "C" const char* octa_last_error(void) { return g_err; }
