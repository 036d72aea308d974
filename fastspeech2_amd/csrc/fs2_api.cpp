// fs2_api.cpp — library-level entry points: version + thread-local last-error string.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void fs2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fs2_last_error(void) { return g_err; }
extern "C" int fs2_version(void) { return 100; }  // 0.1.0
