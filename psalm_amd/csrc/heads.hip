// Prediction-head front end of the mask decoder, fused:  decoder_norm (LayerNorm) -> mask_embed, a 3-layer MLP D -> D -> D -> D with
// ReLU between the layers (mask2former_transformer_decoder.py:695-709, MLP :187-199).  It runs 10 times per image on Q = 100 query
// rows: as separate launches (LayerNorm + 3 skinny GEMMs) every step is a ~8 us latency chain for 13 MFLOP; here one block owns 32
// rows for the whole chain -- the activations never leave LDS, the weights (D x D bf16, 128 KB per layer) stream from L2 straight
// into MFMA B fragments (16-byte loads, half a layer's K range in flight per lane), one barrier per layer.
//   wave w computes the 32-column tiles t = w, w+4, ... of the layer output (D = 256: two tiles per wave) over the full K = D.
#include "common.h"

typedef float hd_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hd_bf16x8 __attribute__((ext_vector_type(8)));

// LayerNorm of the block's 32 rows straight from global memory (8 rows per wave, a lane holds D / 64 values of a row; D % 64 == 0):
// bf16 result to the LDS activation image and to ln_out.  All loads are unconditional (row index clamped) and issued up front --
// a `row < rows ? x[..] : 0` per element compiles to a branch + s_waitcnt vmcnt(0) per load, i.e. 32 serialized L2 round trips.
// Same arithmetic as layernorm_vec_kernel (two-pass variance).
template <int D>
__device__ __forceinline__ void head_ln_rows(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, bf16_t* __restrict__ act0, int pitch,
                                             bf16_t* __restrict__ ln_out, int r0, int rows, int wave, int lane) {
    static_assert(D % 64 == 0, "hidden width");
    constexpr int VPL = D / 64;
    float g[VPL], b[VPL], v[8][VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { g[i] = gamma[i * 64 + lane]; b[i] = beta[i * 64 + lane]; }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const float* xr = x + (long)min(r0 + wave * 8 + rr, rows - 1) * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[rr][i] = xr[i * 64 + lane];
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int rl = wave * 8 + rr, row = r0 + rl;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += v[rr][i];
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float d = v[rr][i] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            const bf16_t o = row < rows ? f32_to_bf16((v[rr][i] - mean) * rstd * g[i] + b[i]) : (bf16_t)0;
            act0[rl * pitch + c] = o;
            if (row < rows) ln_out[(long)row * D + c] = o;
        }
    }
}

// LayerNorm of the 32 pre-norm fp32 rows held in LDS (S, pitch SP), outputs as psalm_layernorm3.  The `add` rows are fetched for all 8
// rows of the wave before the statistics (unconditional, clamped row), so the global round trip overlaps the reductions.
template <int D>
__device__ __forceinline__ void head_ln_from_lds(const float* S, int SP, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float eps, float* __restrict__ y, bf16_t* __restrict__ y2, const float* __restrict__ add,
                                                 int add_rows, bf16_t* __restrict__ y3, int r0, int rows, int wave, int lane) {
    static_assert(D % 64 == 0, "output width");
    constexpr int VPL = D / 64;
    float g[VPL], b[VPL], av[8][VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { g[i] = gamma[i * 64 + lane]; b[i] = beta[i * 64 + lane]; }
    if (y3) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const float* ar = add + (long)(min(r0 + wave * 8 + rr, rows - 1) % add_rows) * D;
#pragma unroll
            for (int i = 0; i < VPL; ++i) av[rr][i] = ar[i * 64 + lane];
        }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int rl = wave * 8 + rr, row = r0 + rl;
        float v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { v[i] = S[rl * SP + i * 64 + lane]; s += v[i]; }
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float d = v[i] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
        if (row < rows) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = i * 64 + lane;
                const float o = (v[i] - mean) * rstd * g[i] + b[i];
                y[(long)row * D + c] = o;
                if (y2) y2[(long)row * D + c] = f32_to_bf16(o);
                if (y3) y3[(long)row * D + c] = f32_to_bf16(o + av[rr][i]);
            }
        }
    }
}

// accumulators + bias + residual -> the fp32 pre-norm rows in LDS.  The residual values of all the wave's tiles are fetched first
// (unconditional, clamped row), then added: one round trip instead of one per element.
template <int D, int TPW>
__device__ __forceinline__ void head_store_prenorm(const hd_f32x16 (&acc)[TPW], const float* __restrict__ bias, const float* __restrict__ res,
                                                   long ldr, float* S, int SP, int r0, int rows, int wave, int n32, int hi) {
    constexpr int NT = D / 32;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile < NT) {
            const int col = tile * 32 + n32;
            const float bv = bias ? bias[col] : 0.f;
            float rv[16];
            if (res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = res[(long)min(r0 + (r & 3) + 8 * (r >> 2) + 4 * hi, rows - 1) * ldr + col];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * hi) * SP + col] = acc[t][r] + bv + rv[r];
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256) ln_mlp3_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, const bf16_t* __restrict__ w0,
                                                      const float* __restrict__ b0, const bf16_t* __restrict__ w1,
                                                      const float* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                      const float* __restrict__ b2, bf16_t* __restrict__ ln_out,
                                                      bf16_t* __restrict__ out, int rows) {
    static_assert(D % 64 == 0 && D <= 256, "hidden width");
    constexpr int PITCH = D + 8;                             // 16-byte aligned rows, 16 consecutive rows on 16 distinct 16-byte slots
    constexpr int NT = D / 32;                               // 32-column output tiles per layer
    constexpr int TPW = (NT + 3) / 4;                        // tiles per wave
    constexpr int KS = D / 16;                               // MFMA k-steps per layer
    constexpr int KC = KS < 8 ? KS : 8;                      // k-steps whose weight fragments are fetched together
    __shared__ __attribute__((aligned(16))) bf16_t act[2][32 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * 32;
    head_ln_rows<D>(x, ldx, gamma, beta, eps, act[0], PITCH, ln_out, r0, rows, wave, lane);   // decoder_norm -> act[0], ln_out
    __syncthreads();
    // ---- three layers
    const bf16_t* W[3] = {w0, w1, w2};
    const float* Bv[3] = {b0, b1, b2};
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const bf16_t* in = act[layer & 1];
        bf16_t* nxt = act[(layer + 1) & 1];
        hd_f32x16 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KC) {
            psalm_u32x4 bw[TPW][KC];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tile = wave + 4 * t;
                if (tile < NT) {
                    const bf16_t* wr = W[layer] + (long)(tile * 32 + n32) * D + 8 * hi;     // weight row = output column
#pragma unroll
                    for (int k = 0; k < KC; ++k) bw[t][k] = *reinterpret_cast<const psalm_u32x4*>(wr + (k0 + k) * 16);
                }
            }
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const psalm_u32x4 a = *reinterpret_cast<const psalm_u32x4*>(&in[n32 * PITCH + (k0 + k) * 16 + 8 * hi]);
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    if (wave + 4 * t < NT)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hd_bf16x8, a), __builtin_bit_cast(hd_bf16x8, bw[t][k]),
                                                                         acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < NT) {
                const int col = tile * 32 + n32;
                const float bias = Bv[layer][col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float y = acc[t][r] + bias;
                    if (layer < 2) nxt[rl * PITCH + col] = f32_to_bf16(y > 0.f ? y : 0.f);
                    else if (r0 + rl < rows) out[(long)(r0 + rl) * D + col] = f32_to_bf16(y);
                }
            }
        }
        if (layer < 2) __syncthreads();
    }
}

// Variant with the weights staged through LDS.  The direct fragment loads above touch 32 cache lines per instruction (lane = weight
// row) and only ceil(rows / 32) CUs issue them; here every wave copies whole 1 KiB pieces (8 weight rows x 64 k, coalesced
// global_load_lds, the GEMM kernel's XOR-swizzled [rows][64] image) into a 2-deep ring of 64-deep K chunks, and the three layers
// form ONE chunk stream: the first chunk of layer j+1 is in flight while the last chunk of layer j is being multiplied.
template <int D>
__global__ void __launch_bounds__(256) ln_mlp3_staged_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, const bf16_t* __restrict__ w0,
                                                             const float* __restrict__ b0, const bf16_t* __restrict__ w1,
                                                             const float* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                             const float* __restrict__ b2, bf16_t* __restrict__ ln_out,
                                                             bf16_t* __restrict__ out, int rows) {
    static_assert(D % 64 == 0 && D <= 256, "hidden width");
    constexpr int PITCH = D + 8, NT = D / 32, TPW = (NT + 3) / 4;
    constexpr int NC = D / 64;                               // 64-deep K chunks per layer
    constexpr int CPW = D / 8 / 4;                           // 1 KiB copies per wave per chunk (D/8 copies of 8 rows)
    __shared__ __attribute__((aligned(16))) bf16_t act[2][32 * PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Ws[2][D * 64];
    const int tid = threadIdx.x, lane = tid & 63, n32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * 32;
    auto issue = [&](int g) {                                // chunk g of the 3 * NC chunk stream -> ring slot g & 1
        const int layer = g / NC, c = g % NC;
        const bf16_t* Wl = layer == 0 ? w0 : (layer == 1 ? w1 : w2);
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int copy = wave + 4 * i, r = copy * 8 + lane / 8;
            const int kc = (lane % 8) ^ ((r >> 1) & 7);
            psalm_glds16(Wl + (long)r * D + c * 64 + kc * 8, &Ws[g & 1][copy * 8 * 64]);
        }
    };
    issue(0);                                                // in flight during the LayerNorm
    head_ln_rows<D>(x, ldx, gamma, beta, eps, act[0], PITCH, ln_out, r0, rows, wave, lane);
    hd_f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int fsw = (n32 >> 1) & 7;
#pragma unroll 1
    for (int g = 0; g < 3 * NC; ++g) {
        const int layer = g / NC, c = g % NC;
        PSALM_WAIT_VMCNT(0);                                 // this wave's pieces of chunk g have landed ...
        PSALM_RAW_BARRIER();                                 // ... everyone's have; slot (g+1)&1 is no longer read; previous layer's activations visible
        if (g + 1 < 3 * NC) issue(g + 1);
        const bf16_t* in = act[layer & 1];
        const bf16_t* Wc = Ws[g & 1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const psalm_u32x4 a = *reinterpret_cast<const psalm_u32x4*>(&in[n32 * PITCH + c * 64 + kk * 16 + 8 * hi]);
            const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (wave + 4 * t < NT) {
                    const psalm_u32x4 b = *reinterpret_cast<const psalm_u32x4*>(&Wc[((wave + 4 * t) * 32 + n32) * 64 + co]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hd_bf16x8, a), __builtin_bit_cast(hd_bf16x8, b), acc[t], 0, 0, 0);
                }
        }
        if (c == NC - 1) {                                   // layer finished: bias (+ ReLU) -> the other activation buffer / the result
            const float* bl = layer == 0 ? b0 : (layer == 1 ? b1 : b2);
            bf16_t* nxt = act[(layer + 1) & 1];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tile = wave + 4 * t;
                if (tile < NT) {
                    const int col = tile * 32 + n32;
                    const float bias = bl[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float yv = acc[t][r] + bias;
                        if (layer < 2) nxt[rl * PITCH + col] = f32_to_bf16(yv > 0.f ? yv : 0.f);
                        else if (r0 + rl < rows) out[(long)(r0 + rl) * D + col] = f32_to_bf16(yv);
                        acc[t][r] = 0.f;
                    }
                }
            }
        }
    }
}

// x (rows, D) f32, row stride ldx;  gamma / beta (D) f32;  w_j (D, D) bf16 row-major (nn.Linear weight), b_j (D) f32;
// ln_out (rows, D) bf16 = LayerNorm(x) (the operand of the class / SEG / region heads);  out (rows, D) bf16 = W2.relu(W1.relu(W0.ln+b0)+b1)+b2.
// D in {64, 128, 256}.
// 1 (default): weights staged through LDS;  0: direct per-lane fragment loads (A/B switch, tools/bench_heads.py)
static int g_heads_staged = 1;
extern "C" int psalm_heads_set_variant(int staged) { g_heads_staged = staged ? 1 : 0; return 0; }

extern "C" int psalm_ln_mlp3(const float* x, long ldx, const float* gamma, const float* beta, float eps, const void* w0, const float* b0,
                             const void* w1, const float* b1, const void* w2, const float* b2, void* ln_out_bf16, void* out_bf16,
                             int rows, int D, void* stream) {
    PSALM_CHECK_ARG(D == 64 || D == 128 || D == 256, "psalm_ln_mlp3: D must be 64, 128 or 256");
    PSALM_CHECK_ARG(((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2) % 16 == 0, "psalm_ln_mlp3: weights must be 16-byte aligned");
    if (rows == 0) return 0;
    const dim3 grid((rows + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
#define HD_LAUNCH(KERN_, D_) hipLaunchKernelGGL((KERN_<D_>), grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, (const bf16_t*)w0, b0, \
                                               (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (bf16_t*)ln_out_bf16, (bf16_t*)out_bf16, rows)
    if (g_heads_staged) {
        if (D == 256) HD_LAUNCH(ln_mlp3_staged_kernel, 256);
        else if (D == 128) HD_LAUNCH(ln_mlp3_staged_kernel, 128);
        else HD_LAUNCH(ln_mlp3_staged_kernel, 64);
    } else {
        if (D == 256) HD_LAUNCH(ln_mlp3_kernel, 256);
        else if (D == 128) HD_LAUNCH(ln_mlp3_kernel, 128);
        else HD_LAUNCH(ln_mlp3_kernel, 64);
    }
#undef HD_LAUNCH
    PSALM_LAUNCH_END("psalm_ln_mlp3");
}

// ------------------------------------------------------------------------------------------------------------------------
// Post-norm residual sub-layer tail of the mask decoder, fused:  y = LayerNorm(residual + a . W^T + bias)  (CrossAttentionLayer /
// SelfAttentionLayer forward_post, mask2former_transformer_decoder.py:40-50, 99-111: out-projection of nn.MultiheadAttention, residual
// add, LayerNorm).  Q = 100 rows, D = 256: as a skinny GEMM + a LayerNorm launch it is two ~8 us latency chains, 18 times per image.
// One block owns 32 rows and the full D-wide output row, so the LayerNorm happens in the block: operand rows staged in LDS, weights
// streamed from L2 into MFMA fragments, bias + residual added from registers, the fp32 rows meet in LDS for the row statistics.
// Outputs as psalm_layernorm3: y f32, optional y2 = bf16(y), optional y3 = bf16(y + add[row % add_rows]).
template <int D>
__global__ void __launch_bounds__(256) linear_res_ln_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ res, long ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ y, bf16_t* __restrict__ y2, const float* __restrict__ add,
                                                            int add_rows, bf16_t* __restrict__ y3, int rows, int K) {
    static_assert(D % 64 == 0 && D <= 256, "output width");
    constexpr int NT = D / 32, TPW = (NT + 3) / 4;
    constexpr int SP = D + 4;                                // fp32 row pitch of the pre-norm rows
    HIP_DYNAMIC_SHARED(unsigned char, smem_raw)
    const int AP = K + 8;                                    // bf16 row pitch of the staged operand rows
    bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);        // [32][AP]
    float* S = reinterpret_cast<float*>(smem_raw + (((size_t)32 * AP * 2 + 15) & ~(size_t)15));   // [32][SP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * 32;
    for (int e = tid; e < 32 * (K / 8); e += 256) {          // operand rows -> LDS (rows beyond `rows`: zeros)
        const int rl = e / (K / 8), c8 = (e % (K / 8)) * 8;
        psalm_u32x4 v{0, 0, 0, 0};
        if (r0 + rl < rows) v = *reinterpret_cast<const psalm_u32x4*>(a + (long)(r0 + rl) * lda + c8);
        *reinterpret_cast<psalm_u32x4*>(&As[rl * AP + c8]) = v;
    }
    __syncthreads();
    hd_f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int ks = K / 16;
    for (int k0 = 0; k0 < ks; k0 += 8) {                     // 8 k-steps of weight fragments in flight per tile
        psalm_u32x4 bw[TPW][8];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < NT) {
                const bf16_t* wr = w + (long)(tile * 32 + n32) * K + 8 * hi;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k0 + k < ks) bw[t][k] = *reinterpret_cast<const psalm_u32x4*>(wr + (k0 + k) * 16);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k0 + k < ks) {
                const psalm_u32x4 af = *reinterpret_cast<const psalm_u32x4*>(&As[n32 * AP + (k0 + k) * 16 + 8 * hi]);
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    if (wave + 4 * t < NT)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hd_bf16x8, af), __builtin_bit_cast(hd_bf16x8, bw[t][k]),
                                                                         acc[t], 0, 0, 0);
            }
        }
    }
    head_store_prenorm<D, TPW>(acc, bias, res, ldr, S, SP, r0, rows, wave, n32, hi);
    __syncthreads();
    head_ln_from_lds<D>(S, SP, gamma, beta, eps, y, y2, add, add_rows, y3, r0, rows, wave, lane);
}

// LDS-staged weights (see ln_mlp3_staged_kernel): 64-deep K chunks of the (D, K) weight matrix in a 2-deep ring; K % 64 == 0.
// The fp32 pre-norm rows reuse the ring's memory after the last chunk.
template <int D>
__global__ void __launch_bounds__(256) linear_res_ln_staged_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ w,
                                                                   const float* __restrict__ bias, const float* __restrict__ res, long ldr,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float eps, float* __restrict__ y, bf16_t* __restrict__ y2,
                                                                   const float* __restrict__ add, int add_rows, bf16_t* __restrict__ y3,
                                                                   int rows, int K) {
    static_assert(D % 64 == 0 && D <= 256, "output width");
    constexpr int NT = D / 32, TPW = (NT + 3) / 4, CPW = D / 8 / 4, SP = D + 4, KMAX = 512;
    static_assert(32 * SP * 4 <= 2 * D * 64 * 2, "pre-norm rows must fit in the weight ring");
    __shared__ __attribute__((aligned(16))) bf16_t As[32 * (KMAX + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t Ws[2][D * 64];
    const int AP = K + 8;
    const int tid = threadIdx.x, lane = tid & 63, n32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * 32, nc = K / 64;
    auto issue = [&](int c) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int copy = wave + 4 * i, r = copy * 8 + lane / 8;
            const int kc = (lane % 8) ^ ((r >> 1) & 7);
            psalm_glds16(w + (long)r * K + c * 64 + kc * 8, &Ws[c & 1][copy * 8 * 64]);
        }
    };
    issue(0);
    for (int e = tid; e < 32 * (K / 8); e += 256) {          // operand rows -> LDS (rows beyond `rows`: zeros)
        const int rl = e / (K / 8), c8 = (e % (K / 8)) * 8;
        psalm_u32x4 v{0, 0, 0, 0};
        if (r0 + rl < rows) v = *reinterpret_cast<const psalm_u32x4*>(a + (long)(r0 + rl) * lda + c8);
        *reinterpret_cast<psalm_u32x4*>(&As[rl * AP + c8]) = v;
    }
    hd_f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int fsw = (n32 >> 1) & 7;
#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
        PSALM_WAIT_VMCNT(0);
        PSALM_RAW_BARRIER();                                 // chunk c landed for everyone (and, c == 0, the operand rows are in LDS)
        if (c + 1 < nc) issue(c + 1);
        const bf16_t* Wc = Ws[c & 1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const psalm_u32x4 af = *reinterpret_cast<const psalm_u32x4*>(&As[n32 * AP + c * 64 + kk * 16 + 8 * hi]);
            const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (wave + 4 * t < NT) {
                    const psalm_u32x4 b = *reinterpret_cast<const psalm_u32x4*>(&Wc[((wave + 4 * t) * 32 + n32) * 64 + co]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hd_bf16x8, af), __builtin_bit_cast(hd_bf16x8, b), acc[t], 0, 0, 0);
                }
        }
    }
    __syncthreads();                                         // every wave is done reading the ring: it now holds the pre-norm rows
    float* S = reinterpret_cast<float*>(&Ws[0][0]);
    head_store_prenorm<D, TPW>(acc, bias, res, ldr, S, SP, r0, rows, wave, n32, hi);
    __syncthreads();
    head_ln_from_lds<D>(S, SP, gamma, beta, eps, y, y2, add, add_rows, y3, r0, rows, wave, lane);
}

// a (rows,K) bf16 row stride lda;  w (D,K) bf16 (nn.Linear weight);  bias (D) f32 or NULL;  residual (rows,D) f32 row stride ldr or NULL;
// y (rows,D) f32 = LayerNorm(residual + a.w^T + bias);  y2 (rows,D) bf16 or NULL;  y3 (rows,D) bf16 = y + add[row % add_rows] or NULL
// (add (add_rows,D) f32).  D in {64,128,256}, K % 16 == 0, K <= 512.
extern "C" int psalm_linear_res_ln(const void* a_bf16, long lda, const void* w_bf16, const float* bias, const float* residual, long ldr,
                                   const float* gamma, const float* beta, float eps, float* y, void* y2_bf16, const float* add,
                                   int add_rows, void* y3_bf16, int rows, int D, int K, void* stream) {
    PSALM_CHECK_ARG(D == 64 || D == 128 || D == 256, "psalm_linear_res_ln: D must be 64, 128 or 256");
    PSALM_CHECK_ARG(K > 0 && K % 16 == 0 && K <= 512, "psalm_linear_res_ln: K % 16 == 0, K <= 512 (operand rows + fp32 rows in 64 KB of LDS)");
    PSALM_CHECK_ARG(((uintptr_t)a_bf16 | (uintptr_t)w_bf16) % 16 == 0 && (lda * 2) % 16 == 0, "psalm_linear_res_ln: 16-byte aligned operand rows");
    PSALM_CHECK_ARG(y3_bf16 == nullptr || (add != nullptr && add_rows > 0), "psalm_linear_res_ln: y3 needs the add table");
    if (rows == 0) return 0;
    const dim3 grid((rows + 31) / 32);
    const size_t shmem = (((size_t)32 * (K + 8) * 2 + 15) & ~(size_t)15) + (size_t)32 * (D + 4) * 4;
    hipStream_t s = (hipStream_t)stream;
#define HD_LAUNCH(D_) hipLaunchKernelGGL((linear_res_ln_kernel<D_>), grid, dim3(256), shmem, s, (const bf16_t*)a_bf16, lda, (const bf16_t*)w_bf16, \
                                         bias, residual, ldr, gamma, beta, eps, y, (bf16_t*)y2_bf16, add, add_rows > 0 ? add_rows : 1,       \
                                         (bf16_t*)y3_bf16, rows, K)
    if (g_heads_staged && K % 64 == 0) {
#define HD_LAUNCH_S(D_) hipLaunchKernelGGL((linear_res_ln_staged_kernel<D_>), grid, dim3(256), 0, s, (const bf16_t*)a_bf16, lda,             \
                                           (const bf16_t*)w_bf16, bias, residual, ldr, gamma, beta, eps, y, (bf16_t*)y2_bf16, add,           \
                                           add_rows > 0 ? add_rows : 1, (bf16_t*)y3_bf16, rows, K)
        if (D == 256) HD_LAUNCH_S(256);
        else if (D == 128) HD_LAUNCH_S(128);
        else HD_LAUNCH_S(64);
#undef HD_LAUNCH_S
    }
    else if (D == 256) HD_LAUNCH(256);
    else if (D == 128) HD_LAUNCH(128);
    else HD_LAUNCH(64);
#undef HD_LAUNCH
    PSALM_LAUNCH_END("psalm_linear_res_ln");
}
