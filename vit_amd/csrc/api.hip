// Library-level entry points: version and the thread-local error string.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void va_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* va_last_error(void) { return g_err; }
extern "C" int va_version(void) { return 100; }
