// Diagnostic (not product, not linked into libpsalm_hip.so): what fraction of the fp32 matrix-pipe peak do DEPENDENT chains of
// v_mfma_f32_32x32x2_f32 reach on gfx950 as a function of (a) resident waves per SIMD, (b) independent accumulators per wave,
// (c) VALU work that depends on the chain's result between chains (an online-softmax stand-in), (d) L2-resident loads in front of
// each chain?  The fp32 attention kernels (csrc/attention.hip) sit at ~37 % of the pipe with every cheap lever measured and
// rejected (profiles/r02n_attn_*); this isolates the instruction-issue side of that.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_f32_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
// prints one JSON line per configuration: {"waves_per_simd", "acc", "valu", "loads", "us", "mfma_per_s", "frac_of_peak"}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One wave = `iters` tiles; a tile = CH products per accumulator on ACC accumulators (ACC * CH products), then VALU stand-in work on the
// results (VALU x 16 dependent exp/fma per register), then the results feed the next tile's B operand (a real dependency, as P feeds O).
template <int ACC, int VALU, int LOADS>
__global__ void __launch_bounds__(256) chain_kernel(const float* __restrict__ src, float* __restrict__ out, int iters, long stride) {
    constexpr int CH = 32;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float* p = src + (wid * 64 + lane) * 4 % stride;
    float a = 1.0f + 1e-3f * lane, b = 1.0f - 1e-3f * lane;
    f32x16 acc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        float ld[LOADS > 0 ? LOADS * 4 : 1];
        if constexpr (LOADS > 0) {
#pragma unroll
            for (int c = 0; c < LOADS; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(p + ((long)(it * LOADS + c) * 256) % stride);
                ld[4 * c] = t.x; ld[4 * c + 1] = t.y; ld[4 * c + 2] = t.z; ld[4 * c + 3] = t.w;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const float av = LOADS > 0 ? a + ld[(c * ACC + k) % (LOADS * 4)] : a;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[k], 0, 0, 0);
            }
        if constexpr (VALU > 0) {
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < ACC; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[k][r]);
#pragma unroll
            for (int k = 0; k < ACC; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[k][r] - m;
#pragma unroll
                    for (int v = 0; v < VALU; ++v) x = __expf(x * 0.5f) - 1.0f;
                    acc[k][r] = x;
                }
            b = 1.0f + 1e-6f * acc[0][0];
        } else {
            b = b * 0.999f + 1e-9f * acc[0][0];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[wid * 64 + lane] = s;
}

template <int ACC, int VALU, int LOADS>
static void run(int waves_per_simd, const float* src, float* out, long stride) {
    const int cus = 256, iters = 64 / ACC;                       // the same number of products per wave whatever ACC
    const int blocks = cus * waves_per_simd;                     // 256-thread blocks: one wave on each of a CU's 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((chain_kernel<ACC, VALU, LOADS>), dim3(blocks), dim3(256), 0, 0, src, out, iters, stride);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((chain_kernel<ACC, VALU, LOADS>), dim3(blocks), dim3(256), 0, 0, src, out, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double mfma = (double)blocks * 4 * iters * ACC * 32;   // products per launch
    const double flops = mfma * 4096.0;                          // 32 x 32 x 2 x 2
    printf("{\"waves_per_simd\": %d, \"acc\": %d, \"valu\": %d, \"loads\": %d, \"us\": %.2f, \"tflops\": %.1f, \"frac_of_157TF\": %.3f}\n",
           waves_per_simd, ACC, VALU, LOADS, us, flops / us / 1e6, flops / us / 1e6 / 157.3);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const long stride = 1 << 20;                                 // 4 MB of floats: L2-resident source
    float *src, *out;
    hipMalloc(&src, stride * sizeof(float) + 4096);
    hipMalloc(&out, 256L * 8 * 256 * sizeof(float));
    hipMemset(src, 0, stride * sizeof(float) + 4096);
    for (int w = 1; w <= 4; ++w) {
        run<1, 0, 0>(w, src, out, stride);                       // one dependent chain per wave
        run<2, 0, 0>(w, src, out, stride);                       // two independent chains per wave
        run<1, 2, 0>(w, src, out, stride);                       // + softmax-like VALU on the result between chains
        run<2, 2, 0>(w, src, out, stride);
        run<1, 2, 8>(w, src, out, stride);                       // + 8 x 16-byte L2 loads feeding the chain (the K fragments)
        run<2, 2, 8>(w, src, out, stride);
    }
    hipFree(src); hipFree(out);
    return 0;
}
