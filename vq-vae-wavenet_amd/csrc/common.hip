// Error reporting + ABI version of libvqwave.
#include <stdlib.h>

#include "vqw_common.h"

static thread_local char g_err[512] = "";

int vqw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// Per-device facts are looked up for the CURRENT device of the calling thread on every call: the library keeps no
// process-wide mutable state, so one process may drive several devices.  hipGetDevice and
// hipDeviceGetAttribute are host-side table reads (no driver round trip).
int vqw_device_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

// Debug switches ("0" disables): read per call, so a test can flip them between calls.
int vqw_env_enabled(const char* name) {
    const char* e = getenv(name);
    return (e && e[0] == '0') ? 0 : 1;
}

extern "C" const char* vqw_last_error(void) { return g_err; }
extern "C" int vqw_abi_version(void) { return VQW_ABI_VERSION; }
