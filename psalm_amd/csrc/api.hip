// Error reporting + library identity for the C ABI (include/psalm_hip.h).
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

extern "C" void psalm_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* psalm_last_error() { return g_err; }
extern "C" int psalm_abi_version() { return 3; }   // 3: operand-form arguments of the split-f16 entry points (x8)
// "hip-gfx950" for the product library; the host-emulation build used by the CPU tests reports "emu".
extern "C" const char* psalm_backend() {
#ifdef PSALM_EMU_BUILD
    return "emu";
#else
    return "hip-gfx950";
#endif
}
