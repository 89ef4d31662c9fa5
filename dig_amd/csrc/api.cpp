// Library-level entry points: version and last-error string (thread-local).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dig_hip.h"

static thread_local char g_err[512] = "";

void dig_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dig_version(void) { return 100; }
extern "C" const char* dig_last_error(void) { return g_err; }
