// Error plumbing + ABI version of libmerlot_hip.so (host only).
#include <cstdarg>
#include <cstdio>

#include "../../include/merlot_hip.h"

static thread_local char g_err[512] = "";

void merlot_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* merlot_last_error(void) { return g_err; }
extern "C" int merlot_abi_version(void) { return MERLOT_ABI_VERSION; }
