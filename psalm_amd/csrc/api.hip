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

// Buffer clears / device-to-device copies of the per-image path as plain stream operations (captured into the hipGraph as memset / memcpy
// nodes), so that the path launches no framework kernel for them (north star: PyTorch for tensor containers and launch glue only).
extern "C" int psalm_memset_zero(void* p, long bytes, void* stream) {
    if (bytes <= 0) return 0;
    PSALM_CHECK_ARG(p != nullptr, "psalm_memset_zero: null pointer");
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) { psalm_set_error("psalm_memset_zero: hipMemsetAsync failed"); return (int)e; }
    return 0;
}
extern "C" int psalm_copy_d2d(void* dst, const void* src, long bytes, void* stream) {
    if (bytes <= 0) return 0;
    PSALM_CHECK_ARG(dst != nullptr && src != nullptr, "psalm_copy_d2d: null pointer");
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { psalm_set_error("psalm_copy_d2d: hipMemcpyAsync failed"); return (int)e; }
    return 0;
}
