// Error reporting + ABI version of libvqwave.
#include "vqw_common.h"

static thread_local char g_err[512] = "";

int vqw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char* vqw_last_error(void) { return g_err; }
extern "C" int vqw_abi_version(void) { return VQW_ABI_VERSION; }
