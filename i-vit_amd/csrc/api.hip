// api.hip -- version and per-thread error string of libivit_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void ivit_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

IVIT_EXPORT int ivit_version(void) { return 100; }  // 0.1.0

IVIT_EXPORT const char* ivit_last_error_string(void) { return g_err; }
