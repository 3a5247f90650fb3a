// Shared device/host helpers for the PSALM gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define PSALM_F32 0
#define PSALM_BF16 1

typedef unsigned short bf16_t;  // raw bfloat16 bits; all arithmetic is done in fp32

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    unsigned u = ((unsigned)v) << 16;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round-to-nearest-even
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// 64-lane wavefront reductions (CDNA wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ----------------------------------------------------------------------------- host side
extern "C" void psalm_set_error(const char* msg);

#define PSALM_CHECK_ARG(cond, msg)                                  \
    do {                                                            \
        if (!(cond)) { psalm_set_error(msg); return -1; }           \
    } while (0)

#define PSALM_LAUNCH_END(name)                                      \
    do {                                                            \
        hipError_t e__ = hipGetLastError();                         \
        if (e__ != hipSuccess) {                                    \
            char buf__[256];                                        \
            snprintf(buf__, sizeof(buf__), "%s: launch failed: %s", name, hipGetErrorString(e__)); \
            psalm_set_error(buf__);                                 \
            return (int)e__ ? (int)e__ : -2;                        \
        }                                                           \
        return 0;                                                   \
    } while (0)

// dtype dispatch: binds T to float or bf16_t
#define PSALM_DISPATCH(code, T, ...)                                \
    do {                                                            \
        if ((code) == PSALM_F32) { typedef float T; __VA_ARGS__; }  \
        else if ((code) == PSALM_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else { psalm_set_error("bad dtype code"); return -1; }      \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
