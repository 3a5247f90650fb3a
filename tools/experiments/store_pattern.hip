// Diagnostic (not product): what does the STORE PATTERN of a GEMM epilogue cost?  r03a time line (profiles/r03a_gemm_timeline.json): the
// epilogue of the direct-to-LDS GEMM kernel stores a 256 x 256 fp32 tile in 30 us and two 128 x 128 tiles per CU in 16-19 us -- 4 B/clk/CU,
// 2 TB/s over the chip -- a fifth of the M4096 N2048 K512 launch and a sixth of the Phi [k|v|q|fc1] launch.  Every block of this kernel
// writes one BM x BN fp32 tile of a (M, N) matrix from registers (values made of the indices, nothing read), with the lane -> address map of
//   0 "strided32": thread = 8 consecutive columns of a row, two 16-byte stores 16 bytes apart (lanes 32 bytes apart) -- the epilogue TODAY
//   1 "contig16" : thread = 4 consecutive columns, one 16-byte store, lanes contiguous (1 KiB per wave instruction)
//   2 "accT"     : the accumulator layout of a 32x32 MFMA tile computed TRANSPOSED (lane = row, 4 x 4 consecutive columns): 16-byte
//                  stores straight from the accumulators, 32-byte row segments, no LDS pass
//   3 "acc"      : the native accumulator layout (lane = column, 16 rows): 4-byte stores, 128-byte row segments
// and the cache policy  0 plain, 1 nontemporal (nt), 2 write-through (sc1).
//   hipcc --offload-arch=gfx950 -O3 -w tools/experiments/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
struct alignas(16) f4 { float x, y, z, w; };

template <int POL>
__device__ __forceinline__ void st16(float* p, f4 v) {
    if constexpr (POL == 0) *reinterpret_cast<f4*>(p) = v;
    else if constexpr (POL == 1) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4*>(p));
    } else {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 t{v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
    }
}
template <int POL>
__device__ __forceinline__ void st4(float* p, float v) {
    if constexpr (POL == 0) *p = v;
    else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int BM, int BN, int NT, int PAT, int POL>
__global__ void __launch_bounds__(NT) store_kernel(float* __restrict__ C, long ldc, int tiles_n, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = (blockIdx.x / tiles_n) * BM, bn = (blockIdx.x % tiles_n) * BN;
    float* base = C + (long)bm * ldc + bn;
    const float v0 = seed + tid * 1e-3f + blockIdx.x;
    if constexpr (PAT == 0) {
        constexpr int TPR = BN / 8, RPI = NT / TPR;
        const int c8 = (tid % TPR) * 8;
#pragma unroll 8
        for (int it = 0; it < BM / RPI; ++it) {
            float* d = base + (long)(it * RPI + tid / TPR) * ldc + c8;
            st16<POL>(d, f4{v0, v0 + it, v0 + 1.f, v0 + 2.f});
            st16<POL>(d + 4, f4{v0 + 3.f, v0 + it, v0 + 4.f, v0 + 5.f});
        }
    } else if constexpr (PAT == 1) {
        constexpr int TPR = BN / 4, RPI = NT / TPR;
        const int c4 = (tid % TPR) * 4;
#pragma unroll 8
        for (int it = 0; it < BM / RPI; ++it)
            st16<POL>(base + (long)(it * RPI + tid / TPR) * ldc + c4, f4{v0, v0 + it, v0 + 1.f, v0 + 2.f});
    } else {
        // waves as WM x WN over the tile like the GEMM kernel: 512 threads = 2 x 4, 256 threads = 2 x 2; each wave (BM/WM) x (BN/WN) of 32 x 32 tiles
        constexpr int NW = NT / 64, WN = NW == 8 ? 4 : 2, WM = NW / WN, TM = BM / WM / 32, TN = BN / WN / 32;
        const int wm = wave / WN, wn = wave % WN, n32 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* t = base + (long)(wm * (BM / WM) + i * 32) * ldc + wn * (BN / WN) + j * 32;
                if constexpr (PAT == 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) st16<POL>(t + (long)n32 * ldc + g * 8 + 4 * hi, f4{v0, v0 + g, v0 + i, v0 + j});
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st4<POL>(t + (long)((r & 3) + 8 * (r >> 2) + 4 * hi) * ldc + n32, v0 + r + i + j);
                }
            }
    }
}

template <int BM, int BN, int NT, int PAT, int POL>
static void run(const char* what, float* C, int M, int N) {
    const int tiles_m = M / BM, tiles_n = N / BN, blocks = tiles_m * tiles_n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((store_kernel<BM, BN, NT, PAT, POL>), dim3(blocks), dim3(NT), 0, 0, C, (long)N, tiles_n, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((store_kernel<BM, BN, NT, PAT, POL>), dim3(blocks), dim3(NT), 0, 0, C, (long)N, tiles_n, (float)r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)blocks * BM * BN * 4;
    static const char* pn[] = {"strided32", "contig16", "accT", "acc"};
    static const char* fn[] = {"plain", "nt", "sc1"};
    printf("{\"case\": \"%s\", \"tile\": \"%dx%d\", \"threads\": %d, \"blocks\": %d, \"pattern\": \"%s\", \"policy\": \"%s\", \"us\": %.2f, \"TB_s\": %.3f, \"MB\": %.1f}\n",
           what, BM, BN, NT, blocks, pn[PAT], fn[POL], us, bytes / us / 1e6, bytes / 1e6);
    fflush(stdout);
}

template <int BM, int BN, int NT>
static void all(const char* what, float* C, int M, int N) {
    run<BM, BN, NT, 0, 0>(what, C, M, N); run<BM, BN, NT, 0, 1>(what, C, M, N); run<BM, BN, NT, 0, 2>(what, C, M, N);
    run<BM, BN, NT, 1, 0>(what, C, M, N); run<BM, BN, NT, 1, 1>(what, C, M, N); run<BM, BN, NT, 1, 2>(what, C, M, N);
    run<BM, BN, NT, 2, 0>(what, C, M, N); run<BM, BN, NT, 2, 1>(what, C, M, N); run<BM, BN, NT, 2, 2>(what, C, M, N);
    run<BM, BN, NT, 3, 0>(what, C, M, N);
}

int main() {
    float* C;
    hipMalloc(&C, (size_t)1024 * 14336 * 4 + (size_t)21504 * 1024 * 4);
    all<256, 256, 512>("phi_w1 M1024(899) N14336: 224 tiles", C, 1024, 14336);
    all<128, 128, 256>("swin_fc1 M4096 N2048: 512 tiles", C, 4096, 2048);
    all<256, 128, 512>("swin_fc1 M4096 N2048: 256 tiles", C, 4096, 2048);
    all<64, 128, 256>("swin_proj M5184 N512: 324 tiles", C, 5184, 512);
    all<128, 128, 256>("pd_ffn1 M21504 N1024: 1344 tiles", C, 21504, 1024);
    return 0;
}
