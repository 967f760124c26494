// Library-wide plumbing: version and the thread-local error string.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = {0};

void mpn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpn_version(void) { return MPN_VERSION; }

extern "C" int mpn_last_error(char* buf, size_t n) {
    const size_t len = strlen(g_err);
    if (buf && n) {
        const size_t m = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, m);
        buf[m] = 0;
    }
    return (int)len;
}
