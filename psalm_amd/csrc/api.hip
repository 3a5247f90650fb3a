// Error reporting + library identity for the C ABI (include/psalm_hip.h).
#include "common.h"
#include "psalm_hip.h"      // PSALM_ABI_VERSION (and every declaration checked against its definition in this translation unit)

#include <atomic>
#include <cstring>

static thread_local char g_err[512] = "";

extern "C" void psalm_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* psalm_last_error() { return g_err; }
// Process-wide tuning switches (psalm_hip.h: PSALM_TUNE_*).  Atomic words: a host thread that flips one while another thread launches makes that
// thread's NEXT launch see the old or the new value, never a torn one; every switch selects between forms with identical results.
static std::atomic<int> g_tuning[PSALM_TUNE_COUNT] = {{1}, {1}, {1}, {1}, {1}, {0}, {0}, {0}};
extern "C" int psalm_set_tuning(int key, int value) {
    if (key < 0 || key >= PSALM_TUNE_COUNT) { psalm_set_error("psalm_set_tuning: unknown key"); return -1; }
    g_tuning[key].store(value, std::memory_order_relaxed);
    return 0;
}
extern "C" int psalm_get_tuning(int key) {
    if (key < 0 || key >= PSALM_TUNE_COUNT) return 0;
    return g_tuning[key].load(std::memory_order_relaxed);
}
extern "C" int psalm_abi_version() { return PSALM_ABI_VERSION; }
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
// int32 -> int64 (the label / query-index vectors of the results: the reference hands out LongTensors) as a kernel of this library, so that a
// captured launch sequence of the path holds no framework kernel
__global__ void cast_i32_i64_kernel(const int* __restrict__ src, long long* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (long long)src[i];
}
extern "C" int psalm_cast_i32_i64(const int* src, long long* dst, long n, void* stream) {
    if (n <= 0) return 0;
    PSALM_CHECK_ARG(dst != nullptr && src != nullptr, "psalm_cast_i32_i64: null pointer");
    hipLaunchKernelGGL(cast_i32_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    PSALM_LAUNCH_END("psalm_cast_i32_i64");
}
extern "C" int psalm_copy_d2d(void* dst, const void* src, long bytes, void* stream) {
    if (bytes <= 0) return 0;
    PSALM_CHECK_ARG(dst != nullptr && src != nullptr, "psalm_copy_d2d: null pointer");
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { psalm_set_error("psalm_copy_d2d: hipMemcpyAsync failed"); return (int)e; }
    return 0;
}
