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
// two floats -> packed bf16 pair (lo = a, hi = b), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950
typedef float psalm_f32x2_v __attribute__((ext_vector_type(2)));
typedef __bf16 psalm_bf16x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const psalm_f32x2_v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, psalm_bf16x2_v));
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// 8 consecutive elements per lane: one 16-byte (bf16) or two 16-byte (fp32) accesses; p must be 16-byte aligned.
struct alignas(16) psalm_u32x4 { unsigned x, y, z, w; };
struct alignas(8) psalm_u32x2 { unsigned x, y; };
struct alignas(16) psalm_f32x4 { float x, y, z, w; };
__device__ __forceinline__ void ld8(const float* p, float* d) {
    const psalm_f32x4 a = reinterpret_cast<const psalm_f32x4*>(p)[0], b = reinterpret_cast<const psalm_f32x4*>(p)[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float* d) {
    const psalm_u32x4 a = *reinterpret_cast<const psalm_u32x4*>(p);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        d[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void st8(float* p, const float* v) {
    reinterpret_cast<psalm_f32x4*>(p)[0] = psalm_f32x4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<psalm_f32x4*>(p)[1] = psalm_f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8(bf16_t* p, const float* v) {
    *reinterpret_cast<psalm_u32x4*>(p) =
        psalm_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}

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

// ... over aligned groups of LPR = 16 / 32 / 64 lanes (a row per group in the row kernels that put several short rows on one wavefront).  The
// butterfly runs from LPR / 2 down: a row that occupies the first LPR lanes of a 64-lane wave_sum / wave_max (zeros elsewhere) gets the same
// additions in the same order, i.e. the same bits.
template <int LPR> __device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int LPR> __device__ __forceinline__ float seg_max(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4): the LDS destination is the WAVE-UNIFORM
// `lds_wave_base` + lane*16 (hardware adds the lane offset; the base goes through M0), the global source is per lane.
// Completion is tracked by vmcnt; a following __syncthreads() drains it.
#ifdef PSALM_EMU_BUILD
#define PSALM_WAVES_PER_EU(n)
#else
#define PSALM_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))   // register budget of a kernel: at least n resident waves per SIMD
#endif
#ifdef PSALM_EMU_BUILD   // host build of the same kernel sources for the CPU tests (tests/emu): functional stand-in
__device__ __forceinline__ void psalm_glds16(const void* g, void* lds_wave_base) { emu::global_load_lds(g, lds_wave_base, 16); }
#else
__device__ __forceinline__ void psalm_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#endif

// Counted wait on the vector-memory counter (LDS-DMA copies included) + raw workgroup barrier, for software pipelines
// that keep copies in flight across the barrier (a plain __syncthreads() drains vmcnt to 0).  N must be a literal.
#ifdef PSALM_EMU_BUILD
#define PSALM_WAIT_VMCNT(N) do { } while (0)          /* the stand-in's copies are synchronous */
#define PSALM_RAW_BARRIER() __syncthreads()
#define PSALM_SCHED_FENCE() do { } while (0)
#define PSALM_OPAQUE_VGPR(x) do { } while (0)
#define PSALM_OPAQUE_SGPR(x) do { } while (0)
__device__ __forceinline__ float psalm_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float psalm_exp2(float x) { return exp2f(x); }
#else
// makes an int look freshly defined to the optimiser (keeps loop-invariant LDS fragment reads from being hoisted into registers)
#define PSALM_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
// the same for a wave-uniform int: what is derived from it is recomputed where it is used instead of being hoisted out of a loop into (spilled) SGPRs
#define PSALM_OPAQUE_SGPR(x) asm volatile("" : "+s"(x))
#define PSALM_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PSALM_RAW_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define PSALM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)       // no instruction is scheduled across this point
__device__ __forceinline__ float psalm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     /* v_rcp_f32: 1 ulp */
__device__ __forceinline__ float psalm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  /* v_exp_f32 (no denormal range fix-up: callers add 1) */
#endif

// ---- split-f16 operand rows.  An operand row is [hi (Kp) | lo (Kp)] f16 words with x s = hi + lo (s: the row's power-of-two scale,
// |hi| < 2^14, |lo| <= ulp(hi) / 2 <= 4): the GEMM forms hi.hi + lo.hi + hi.lo as three f16 products (22-bit operands).
// y = x s (the scaled element) -> hi word and lo word
__device__ __forceinline__ void psalm_split_words(float y, unsigned& hw, unsigned& sw) {
    const _Float16 h = (_Float16)y;
    hw = (unsigned)__builtin_bit_cast(unsigned short, h);
    sw = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(y - (float)h));
}

// Bounds-checked 4-byte accesses through a buffer descriptor (buffer_store_dword / buffer_load_dword ... offen): an access at a byte offset
// >= the descriptor's size is DROPPED (store) or returns 0 (load) by the hardware -- epilogues write their ragged last row / column tiles
// without a branch per element.  `bytes` <= 0x7fffffff; PSALM_BUF_OOB is the offset that marks a lane as out of range (instruction-level
// immediate offsets added to it must not wrap).
#define PSALM_BUF_OOB 0x80000000u
#ifdef PSALM_EMU_BUILD
__device__ __forceinline__ unsigned psalm_swap_adjacent(unsigned v) { return __shfl_xor(v, 1); }   // value of lane ^ 1
__device__ __forceinline__ unsigned psalm_quad_swap2(unsigned v) { return __shfl_xor(v, 2); }      // value of lane ^ 2
template <int P> __device__ __forceinline__ unsigned psalm_quad_bcast(unsigned v) { return __shfl(v, ((threadIdx.x & 63) & ~3) | P); }   // lane P of this lane's group of 4
struct psalm_rsrc { char* base; unsigned bytes; };
__device__ __forceinline__ psalm_rsrc psalm_make_rsrc(const void* p, unsigned bytes) { return psalm_rsrc{(char*)p, bytes}; }
__device__ __forceinline__ void psalm_buf_store_f32(float v, psalm_rsrc r, unsigned off) { if (off < r.bytes && r.bytes - off >= 4u) *reinterpret_cast<float*>(r.base + off) = v; }
__device__ __forceinline__ void psalm_buf_store_u32(unsigned v, psalm_rsrc r, unsigned off) { if (off < r.bytes && r.bytes - off >= 4u) *reinterpret_cast<unsigned*>(r.base + off) = v; }
__device__ __forceinline__ float psalm_buf_load_f32(psalm_rsrc r, unsigned off) { return (off < r.bytes && r.bytes - off >= 4u) ? *reinterpret_cast<const float*>(r.base + off) : 0.f; }
__device__ __forceinline__ psalm_u32x4 psalm_buf_load_b128(psalm_rsrc r, unsigned off) {
    return (off < r.bytes && r.bytes - off >= 16u) ? *reinterpret_cast<const psalm_u32x4*>(r.base + off) : psalm_u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ void psalm_buf_store_f32_s(float v, psalm_rsrc r, unsigned voff, unsigned soff) { if (voff < PSALM_BUF_OOB) psalm_buf_store_f32(v, r, voff + soff); }
__device__ __forceinline__ float psalm_buf_load_f32_s(psalm_rsrc r, unsigned voff, unsigned soff) { return voff < PSALM_BUF_OOB ? psalm_buf_load_f32(r, voff + soff) : 0.f; }
__device__ __forceinline__ psalm_u32x4 psalm_buf_load_b128_s(psalm_rsrc r, unsigned voff, unsigned soff) {
    return voff < PSALM_BUF_OOB ? psalm_buf_load_b128(r, voff + soff) : psalm_u32x4{0u, 0u, 0u, 0u};
}
#else
__device__ __forceinline__ unsigned psalm_swap_adjacent(unsigned v) {                              // DPP quad_perm [1,0,3,2]: no LDS crossbar
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned psalm_quad_swap2(unsigned v) {                                 // DPP quad_perm [2,3,0,1]
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
}
template <int P> __device__ __forceinline__ unsigned psalm_quad_bcast(unsigned v) {                // DPP quad_perm [P,P,P,P]: lane P of the group of 4
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, P * 0x55, 0xF, 0xF, true);
}
typedef __amdgpu_buffer_rsrc_t psalm_rsrc;
__device__ __forceinline__ psalm_rsrc psalm_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);   // raw buffer (stride 0), 32-bit data format
}
__device__ __forceinline__ void psalm_buf_store_f32(float v, psalm_rsrc r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);   // (the builtin's data operand is an integer: a float argument would be VALUE-converted)
}
__device__ __forceinline__ void psalm_buf_store_u32(unsigned v, psalm_rsrc r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0); }
__device__ __forceinline__ float psalm_buf_load_f32(psalm_rsrc r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
typedef unsigned psalm_u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ psalm_u32x4 psalm_buf_load_b128(psalm_rsrc r, unsigned off) {          // buffer_load_dwordx4 ... offen: one VGPR of address
    const psalm_u32x4_v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return psalm_u32x4{v.x, v.y, v.z, v.w};
}
// ... with a wave-uniform part of the offset in an SGPR (soff; buffers < 2^31 bytes, a per-lane voff >= PSALM_BUF_OOB still drops the access):
// one VGPR of address per access instead of a 64-bit pointer
__device__ __forceinline__ void psalm_buf_store_f32_s(float v, psalm_rsrc r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float psalm_buf_load_f32_s(psalm_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ psalm_u32x4 psalm_buf_load_b128_s(psalm_rsrc r, unsigned voff, unsigned soff) {
    const psalm_u32x4_v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return psalm_u32x4{v.x, v.y, v.z, v.w};
}
#endif

// ----------------------------------------------------------------------------- host side
extern "C" void psalm_set_error(const char* msg);
extern "C" int psalm_get_tuning(int key);          // api.hip; keys: PSALM_TUNE_* (mirrored from include/psalm_hip.h for the translation units that do not include it)
#ifndef PSALM_TUNE_COUNT
#define PSALM_TUNE_GEMM_XCD_KSPLIT 0
#define PSALM_TUNE_ATTN_XCD_HEADS 1
#define PSALM_TUNE_GEMM_MID 2
#define PSALM_TUNE_DECODER_FUSE 3
#define PSALM_TUNE_ROW_GROUPS 4
#define PSALM_TUNE_COUNT 8
#endif

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
