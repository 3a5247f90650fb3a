// Diagnostic (not product): does the ACCESS PATTERN of the attention kernels' fragment loads bound them?  Each wave runs `iters` tiles; a
// tile = NL 16-byte (or 4-byte) loads per lane from an L2-resident buffer + (optionally) 64 dependent v_mfma_f32_32x32x2_f32 that consume
// them, as a key tile of causal_attention_f32_splitk_kernel does.  Patterns:
//   0  "row"   : lane (n32, hi) reads 16-byte chunk c of ITS row (rows 256 B apart) -- the K fragment loads today: 64 lines per instruction
//   1  "coal"  : lane i reads byte 16 i of a 1 KiB block per instruction -- fragment-ordered storage: 8 lines per instruction
//   2  "dword2": 4-byte loads, lanes n32 consecutive, hi -> another row (the V loads today: 2 lines per instruction), 4x as many loads
//   hipcc --offload-arch=gfx950 -O3 -w tools/experiments/load_pattern.hip -o /tmp/load_pattern && /tmp/load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PAT, int MFMA>
__global__ void __launch_bounds__(256) tile_kernel(const float* __restrict__ src, float* __restrict__ out, int iters, long nfloat) {
    const int lane = threadIdx.x & 63, n32 = lane & 31, hi = lane >> 5;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float b = 1.0f + 1e-3f * lane, sum = 0.f;
    for (int it = 0; it < iters; ++it) {
        const long tile = (wid * 7 + it * 13) % (nfloat / 2048 - 1);          // a 8 KiB tile of the buffer (32 rows x 64 floats)
        const float* base = src + tile * 2048;
        float v[32];
        if constexpr (PAT == 2) {
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = base[(long)((c & 15) * 2 + hi) * 64 + (c >> 4) * 32 + n32];
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 t = PAT == 0 ? *reinterpret_cast<const float4*>(base + n32 * 64 + 32 * hi + 4 * c)
                                          : *reinterpret_cast<const float4*>(base + c * 256 + lane * 4);
                v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
            }
        }
        if constexpr (MFMA) {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c], b + 1.f, acc, 0, 0, 0);
            }
            b = b * 0.999f + 1e-9f * acc[0];
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) sum += v[c];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[wid * 64 + lane] = sum;
}

template <int PAT, int MFMA>
static void run(int wps, const float* src, float* out, long nfloat) {
    const int blocks = 256 * wps, iters = 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((tile_kernel<PAT, MFMA>), dim3(blocks), dim3(256), 0, 0, src, out, iters, nfloat);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((tile_kernel<PAT, MFMA>), dim3(blocks), dim3(256), 0, 0, src, out, iters, nfloat);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20, tiles_per_simd = (double)wps * iters;
    printf("{\"pattern\": %d, \"mfma\": %d, \"waves_per_simd\": %d, \"us\": %.2f, \"cycles_per_tile_per_simd_at_2.4GHz\": %.0f}\n", PAT, MFMA, wps, us,
           us * 2400.0 / tiles_per_simd);
}

int main() {
    const long nfloat = 1L << 21;                                          // 8 MB, L2 / MALL resident
    float *src, *out;
    hipMalloc(&src, nfloat * sizeof(float));
    hipMalloc(&out, 256L * 4 * 4 * 64 * sizeof(float));
    hipMemset(src, 0, nfloat * sizeof(float));
    for (int wps = 2; wps <= 3; ++wps) {
        run<0, 0>(wps, src, out, nfloat); run<1, 0>(wps, src, out, nfloat); run<2, 0>(wps, src, out, nfloat);
        run<0, 1>(wps, src, out, nfloat); run<1, 1>(wps, src, out, nfloat); run<2, 1>(wps, src, out, nfloat);
    }
    hipFree(src); hipFree(out);
    return 0;
}
