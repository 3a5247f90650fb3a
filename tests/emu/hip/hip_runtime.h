// TEST INFRASTRUCTURE -- a functional host-side stand-in for <hip/hip_runtime.h>.
//
// The authoring container has no GPU, and GPU minutes are scarce, so the unmodified HIP kernel
// sources under psalm_amd/csrc are ALSO compiled for the host against this header
// (tests/emu/build_emu.py: `clang++ -x c++ -I tests/emu ...`) into tests/emu/_build/libpsalm_emu.so.
// That library exports the same C ABI as libpsalm_hip.so and lets the CPU test-suite run every
// kernel's indexing / LDS / wave-shuffle / MFMA-fragment logic at small sizes before a gpurun.
//
// It is NOT a product fallback: psalm_amd never loads it; only tests do, explicitly.
//
// Model: one block at a time; every HIP thread of the block is a ucontext fiber scheduled
// round-robin on the calling OS thread.  __syncthreads() and the wave-level operations
// (__shfl*, MFMA) are rendezvous points.  A wave is 64 consecutive linear thread ids (CDNA).
// MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md §3:
//   32x32xK: A[i=l&31][k = KL*(l>>5)+j]  B[k = KL*(l>>5)+j][n=l&31]  D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
//   16x16xK: A[i=l&15][k = KL*(l>>4)+j]  B[k = KL*(l>>4)+j][n=l&15]  D[row=4*(l>>4)+r][col=l&15]
// with KL = elements per lane (8 for bf16 x16/x32 forms, 1 for the f32 x2/x4 forms).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_smem();

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
typedef void* hipStream_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static const int hipMemcpyDeviceToDevice = 3;
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
static const int hipFuncAttributeMaxDynamicSharedMemorySize = 8;

// Fiber switch.  glibc's swapcontext saves / restores the signal mask with two system calls per switch, and a block of 256
// fibers switches thousands of times per barrier-heavy kernel: on x86-64 the switch is therefore a 15-instruction routine that
// exchanges the callee-saved registers and the stack pointer (weak symbol: the header is compiled into several objects).
#if defined(__x86_64__)
#define PSALM_EMU_FAST_SWITCH 1
extern "C" void psalm_emu_ctx_switch(void** save_sp, void* load_sp);
__asm__(R"(
    .text
    .weak psalm_emu_ctx_switch
    .type psalm_emu_ctx_switch,@function
psalm_emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size psalm_emu_ctx_switch, .-psalm_emu_ctx_switch
)");
#endif

namespace emu {

struct Fiber {
#ifdef PSALM_EMU_FAST_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int lin = 0;
};
struct Wave {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    const void* slot[64];
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    dim3 bid, bdim, gdim;
    std::function<void()> fn;
    unsigned long progress = 0;
};

inline Block*& blk() { static thread_local Block* b = nullptr; return b; }
inline Fiber*& cur() { static thread_local Fiber* f = nullptr; return f; }
#ifdef PSALM_EMU_FAST_SWITCH
inline void*& sched() { static thread_local void* sp = nullptr; return sp; }
#else
inline ucontext_t& sched() { static thread_local ucontext_t c; return c; }
#endif
inline std::vector<char>& smem_buf() { static thread_local std::vector<char> b; return b; }
inline void* dyn_smem() { return smem_buf().data(); }
static const size_t STACK = 256 * 1024;

#ifdef PSALM_EMU_FAST_SWITCH
inline void yield() { psalm_emu_ctx_switch(&cur()->sp, sched()); }
#else
inline void yield() { swapcontext(&cur()->ctx, &sched()); }
#endif

inline void release_checks_on_exit() {
    Block* b = blk();
    Fiber* f = cur();
    f->done = true;
    b->progress++;
    b->alive--;
    if (b->alive > 0 && b->arrived == b->alive) { b->arrived = 0; b->gen++; }
    Wave& w = b->waves[f->lin / 64];
    w.alive--;
    if (w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; w.gen++; }
}

inline void trampoline() {
    blk()->fn();
    release_checks_on_exit();
    yield();                      // a finished fiber is never resumed
    abort();
}

inline void syncthreads() {
    Block* b = blk();
    unsigned g = b->gen;
    if (++b->arrived == b->alive) { b->arrived = 0; b->gen++; b->progress++; }
    else while (b->gen == g) yield();
}

inline void wave_rendezvous() {
    Block* b = blk();
    Wave& w = b->waves[cur()->lin / 64];
    unsigned g = w.gen;
    if (++w.arrived == w.alive) { w.arrived = 0; w.gen++; b->progress++; }
    else while (w.gen == g) yield();
}

// every live lane of the wave deposits a pointer; returns the wave's slot table (valid until wave_done()).
inline const void* const* wave_gather(const void* mine) {
    Wave& w = blk()->waves[cur()->lin / 64];
    w.slot[cur()->lin % 64] = mine;
    wave_rendezvous();
    return w.slot;
}
inline void wave_done() { wave_rendezvous(); }

template <class F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F fn) {
    static thread_local std::vector<char*> stacks;
    unsigned nthr = block.x * block.y * block.z;
    while (stacks.size() < nthr) stacks.push_back((char*)malloc(STACK));
    smem_buf().assign(shmem + 64, 0);
    Block b;
    b.fn = fn;
    b.bdim = block;
    b.gdim = grid;
    blk() = &b;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                b.bid = dim3(bx, by, bz);
                b.fibers.assign(nthr, Fiber());
                b.waves.assign((nthr + 63) / 64, Wave());
                b.alive = nthr;
                b.arrived = 0;
                for (unsigned t = 0; t < nthr; ++t) {
                    Fiber& f = b.fibers[t];
                    f.lin = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.stack = stacks[t];
                    b.waves[t / 64].alive++;
#ifdef PSALM_EMU_FAST_SWITCH
                    {   // initial frame: six callee-saved slots, then the address `ret` jumps to; rsp % 16 == 8 on entry, as after a call
                        void** sp = (void**)(((uintptr_t)f.stack + STACK) & ~(uintptr_t)15);
                        *--sp = nullptr;
                        *--sp = (void*)(void (*)())trampoline;
                        for (int r = 0; r < 6; ++r) *--sp = nullptr;
                        f.sp = sp;
                    }
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = &sched();
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
                }
                int stalled = 0;
                for (;;) {
                    bool any = false;
                    unsigned long p0 = b.progress;
                    for (unsigned t = 0; t < nthr; ++t) {
                        Fiber& f = b.fibers[t];
                        if (f.done) continue;
                        any = true;
                        cur() = &f;
#ifdef PSALM_EMU_FAST_SWITCH
                        psalm_emu_ctx_switch(&sched(), f.sp);
#else
                        swapcontext(&sched(), &f.ctx);
#endif
                    }
                    if (!any) break;
                    if (b.progress == p0) {
                        if (++stalled > 4) {
                            fprintf(stderr, "[hip-emu] DEADLOCK in block (%u,%u,%u): divergent barrier / wave op\n", bx, by, bz);
                            abort();
                        }
                    } else stalled = 0;
                }
            }
    blk() = nullptr;
}
}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::blk()->bid)
#define blockDim (emu::blk()->bdim)
#define gridDim (emu::blk()->gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })

using std::max;
using std::min;
inline void __syncthreads() { emu::syncthreads(); }

template <class T>
inline T __shfl(T v, int src, int width = 64) {
    T mine = v;
    int lane = emu::cur()->lin % 64;
    const void* const* all = emu::wave_gather(&mine);
    int base = lane - (lane % width);
    int s = base + (src % width);
    emu::Block* b = emu::blk();
    int lin = (emu::cur()->lin / 64) * 64 + s;
    T r = (lin < (int)b->fibers.size() && !b->fibers[lin].done) ? *(const T*)all[s] : mine;
    emu::wave_done();
    return r;
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (emu::cur()->lin % 64 % width) ^ mask, width); }
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::cur()->lin % 64 % width;
    return __shfl(v, (l + (int)d < width) ? l + (int)d : l, width);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = emu::cur()->lin % 64 % width;
    return __shfl(v, (l - (int)d >= 0) ? l - (int)d : l, width);
}
inline unsigned long long __ballot(int pred) {
    int mine = pred;
    int lane = emu::cur()->lin % 64;
    const void* const* all = emu::wave_gather(&mine);
    emu::Block* b = emu::blk();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) {
        int lin = (emu::cur()->lin / 64) * 64 + i;
        if (lin < (int)b->fibers.size() && !b->fibers[lin].done && *(const int*)all[i]) m |= 1ull << i;
    }
    (void)lane;
    emu::wave_done();
    return m;
}
inline int __builtin_amdgcn_readfirstlane_emu(int v) { return __shfl(v, 0); }
#define __builtin_amdgcn_readfirstlane __builtin_amdgcn_readfirstlane_emu

template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
inline unsigned atomicCAS(unsigned* p, unsigned c, unsigned v) { unsigned o = *p; if (o == c) *p = v; return o; }
inline void __threadfence() {}
inline void __builtin_amdgcn_s_setprio_emu(int) {}
#define __builtin_amdgcn_s_setprio __builtin_amdgcn_s_setprio_emu
#define __builtin_amdgcn_sched_barrier(x)
#define __builtin_amdgcn_s_barrier() __syncthreads()

inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __saturatef(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

// ---------------------------------------------------------------- direct global -> LDS copy (global_load_lds_dwordx4)
// Hardware semantics (guide §5): LDS destination = WAVE-UNIFORM base (M0) + lane * size; the global source is per lane.
// The stand-in checks that every live lane passes the same base, then copies `size` bytes for this lane.
namespace emu {
inline void global_load_lds(const void* g, void* lds_wave_base, int size) {
    const void* mine = lds_wave_base;
    const int lane = cur()->lin % 64;
    const void* const* all = wave_gather(&mine);
    Block* b = blk();
    for (int i = 0; i < 64; ++i) {
        int lin = (cur()->lin / 64) * 64 + i;
        if (lin < (int)b->fibers.size() && !b->fibers[lin].done && *(void* const*)all[i] != lds_wave_base) {
            fprintf(stderr, "[hip-emu] global_load_lds: LDS base is not wave-uniform\n");
            abort();
        }
    }
    wave_done();
    memcpy((char*)lds_wave_base + (size_t)lane * size, g, size);
}
}  // namespace emu

// ---------------------------------------------------------------- MFMA
namespace emu {
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

inline float bf2f(__bf16 v) {
    unsigned short s;
    memcpy(&s, &v, 2);
    unsigned u = (unsigned)s << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct AB8 { bf16x8_t a, b; };
struct AB1 { float a, b; };

template <int MN, int KL, class AB, class ACC, int NREG>
inline ACC mfma(const AB& mine, ACC c, float (*geta)(const AB&, int), float (*getb)(const AB&, int)) {
    const int lane = cur()->lin % 64;
    const void* const* all = wave_gather(&mine);
    const int kgroups = 64 / MN;
    ACC d = c;
    for (int r = 0; r < NREG; ++r) {
        int row, col;
        if (MN == 32) { row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); col = lane & 31; }
        else { row = 4 * (lane >> 4) + r; col = lane & 15; }
        float acc = d[r];
        for (int kg = 0; kg < kgroups; ++kg)
            for (int j = 0; j < KL; ++j) {
                const AB& la = *(const AB*)all[row + MN * kg];
                const AB& lb = *(const AB*)all[col + MN * kg];
                acc = fmaf(geta(la, j), getb(lb, j), acc);
            }
        d[r] = acc;
    }
    wave_done();
    return d;
}
inline float ga8(const AB8& x, int j) { return bf2f(x.a[j]); }
inline float gb8(const AB8& x, int j) { return bf2f(x.b[j]); }
inline float ga1(const AB1& x, int) { return x.a; }
inline float gb1(const AB1& x, int) { return x.b; }
}  // namespace emu

inline emu::f32x16_t emu_mfma_32x32x16_bf16(emu::bf16x8_t a, emu::bf16x8_t b, emu::f32x16_t c, int, int, int) {
    emu::AB8 m{a, b};
    return emu::mfma<32, 8, emu::AB8, emu::f32x16_t, 16>(m, c, emu::ga8, emu::gb8);
}
inline emu::f32x4_t emu_mfma_16x16x32_bf16(emu::bf16x8_t a, emu::bf16x8_t b, emu::f32x4_t c, int, int, int) {
    emu::AB8 m{a, b};
    return emu::mfma<16, 8, emu::AB8, emu::f32x4_t, 4>(m, c, emu::ga8, emu::gb8);
}
inline emu::f32x16_t emu_mfma_32x32x2f32(float a, float b, emu::f32x16_t c, int, int, int) {
    emu::AB1 m{a, b};
    return emu::mfma<32, 1, emu::AB1, emu::f32x16_t, 16>(m, c, emu::ga1, emu::gb1);
}
inline emu::f32x4_t emu_mfma_16x16x4f32(float a, float b, emu::f32x4_t c, int, int, int) {
    emu::AB1 m{a, b};
    return emu::mfma<16, 1, emu::AB1, emu::f32x4_t, 4>(m, c, emu::ga1, emu::gb1);
}
namespace emu {
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
struct AB8H { f16x8_t a, b; };
inline float ga8h(const AB8H& x, int j) { return (float)x.a[j]; }
inline float gb8h(const AB8H& x, int j) { return (float)x.b[j]; }
}  // namespace emu
// v_mfma_f32_32x32x16_f16: same fragment layout as the bf16 form, IEEE half operands, fp32 accumulate
inline emu::f32x16_t emu_mfma_32x32x16_f16(emu::f16x8_t a, emu::f16x8_t b, emu::f32x16_t c, int, int, int) {
    emu::AB8H m{a, b};
    return emu::mfma<32, 8, emu::AB8H, emu::f32x16_t, 16>(m, c, emu::ga8h, emu::gb8h);
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 emu_mfma_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_16x16x4f32
