// GEMM family for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )      (nn.Linear layout: W is [out,in])
//   epilogue:  v = acc + bias[n];  v = act(v) for n >= act_col_start;  v += residual[m,n];  store as f32|bf16
//              (act | ACT_POST_RESIDUAL applies the activation after the residual add instead)
//
// Covers every dense contraction of the PSALM inference path: Phi q/k/v/dense/fc1/fc2
// (modeling_phi.py:189-260), Swin qkv/proj/mlp/reduction (swin_trans.py:109-149,28-34,266), projector and
// FPN convolutions through im2col (multimodal_projector/builder.py:85-111, msdeformattn.py:196-254), MSDeformAttn
// value/offset/weight/output projections (ops/modules/ms_deform_attn.py:98-123), the predictor's
// in_proj/out_proj/FFN/MLPs and the mask einsum `bqc,bchw->bqhw` (mask2former_transformer_decoder.py:749).
//
// Two arithmetic modes, chosen by the dtype of W:
//   * W bf16  -> v_mfma_f32_32x32x16_bf16, fp32 accumulate.  A may be f32 or bf16 in memory (converted to bf16
//                while staging into LDS, so fp32 residual streams feed the matrix cores without a cast kernel).
//   * W f32   -> v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation (bitwise an fmaf chain), the
//                "reference-precision" mode used to prove structural parity with the fp32 CPU reference.
//
// Tiling (64-wide wavefronts): block = 256 threads = 4 waves as 2x2; block tile BM x 128 (BM = 128 or 64),
// K-step 32 (bf16) / 16 (f32); each wave owns (BM/2) x 64 as (BM/64) x 2 MFMA 32x32 tiles
// (16 fp32 accumulators per tile per lane).  Operands are register-staged global -> LDS with the next
// K-tile's global loads issued before the current tile's MFMAs (issue-early / write-late), LDS double-buffered,
// one barrier per K-step.  LDS rows are padded to 80 B (bf16) so the ds_read_b128 fragment reads of 16
// consecutive rows land on 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// blockIdx -> tile mapping is XCD-aware: consecutive tiles along N (sharing the A row-panel) are placed on the
// same XCD (block b runs on XCD b % 8) so the panel is fetched into that XCD's L2 once.
#include "common.h"

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_GELU 2
#define ACT_GELU_NEW 3
#define ACT_POST_RESIDUAL 16   // flag: apply the activation AFTER adding the residual (ResNet block: relu(out + residual))

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct alignas(16) u32x4_s { unsigned x, y, z, w; };
struct alignas(16) f32x4_g { float x, y, z, w; };

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (act == ACT_GELU_NEW) {
        const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
        return 0.5f * v * (1.f + tanhf(u));
    }
    return v;
}

__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); }

// 8 consecutive K elements of one row -> 8 bf16 (16 B).  Out-of-range rows / K-chunks give zeros.
__device__ __forceinline__ u32x4_s load8_bf16(const bf16_t* p, bool ok) {
    if (!ok) return u32x4_s{0, 0, 0, 0};
    return *reinterpret_cast<const u32x4_s*>(p);
}
__device__ __forceinline__ u32x4_s load8_bf16(const float* p, bool ok) {
    if (!ok) return u32x4_s{0, 0, 0, 0};
    const f32x4_g a = *reinterpret_cast<const f32x4_g*>(p);
    const f32x4_g b = *reinterpret_cast<const f32x4_g*>(p + 4);
    return u32x4_s{pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w)};
}

struct GemmArgs {
    const void* A; const void* W; const float* bias; const void* res; void* C;
    long lda, ldw, ldr, ldc;
    int M, N, K, act, act_col_start;
    int tiles_m, tiles_n;
};

// XCD-aware tile id: hardware places block b on XCD b%8; give each XCD a contiguous run of tile ids
// (bijective for any grid size, cf. guide §5 "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename TC>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, const f32x16& acc, int row0, int col, int lane) {
    if (col >= g.N) return;
    const float b = g.bias ? g.bias[col] : 0.f;
    const int act = g.act & 15;
    const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
    const bool do_act = act != ACT_NONE && col >= g.act_col_start;
    TC* C = (TC*)g.C;
    const TC* R = (const TC*)g.res;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) {
            float v = acc[r] + b;
            if (do_act && !post) v = apply_act(v, act);
            if (R) v += ldf(R + (long)row * g.ldr + col);
            if (do_act && post) v = apply_act(v, act);
            stf(C + (long)row * g.ldc + col, v);
        }
    }
}

// ------------------------------------------------------------------------------------------- bf16 MFMA
template <typename TA, typename TC, int BM>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmArgs g) {
    constexpr int BN = 128, BK = 32, LDS_STRIDE = 40;           // 40 bf16 = 80 B per row
    constexpr int TM = BM / 64;                                  // 32x32 tiles per wave along M
    constexpr int A_CHUNKS = BM * 4 / 256;                       // 16-B chunks per thread per A tile
    constexpr int B_CHUNKS = BN * 4 / 256;
    __shared__ __attribute__((aligned(16))) bf16_t As[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][BN * LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int bm = (tile / g.tiles_n) * BM, bn = (tile % g.tiles_n) * BN;
    const TA* A = (const TA*)g.A;
    const bf16_t* W = (const bf16_t*)g.W;

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4_s ra[A_CHUNKS], rb[B_CHUNKS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            const bool ok = (bm + row < g.M) && (k0 + kc < g.K);
            ra[i] = load8_bf16(A + (long)(bm + row) * g.lda + k0 + kc, ok);
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            const bool ok = (bn + row < g.N) && (k0 + kc < g.K);
            rb[i] = load8_bf16(W + (long)(bn + row) * g.ldw + k0 + kc, ok);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            *reinterpret_cast<u32x4_s*>(&As[buf][row * LDS_STRIDE + kc]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 8;
            *reinterpret_cast<u32x4_s*>(&Bs[buf][row * LDS_STRIDE + kc]) = rb[i];
        }
    };

    const int nk = (g.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);                  // issue next tile's global loads early
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[2];
            const int koff = ks * 16 + 8 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 32 + (lane & 31);
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As[buf][row * LDS_STRIDE + koff]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + (lane & 31);
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs[buf][row * LDS_STRIDE + koff]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);                        // write late, into the other buffer
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            epilogue_store<TC>(g, acc[i][j], bm + wm * (BM / 2) + i * 32, bn + wn * 64 + j * 32 + (lane & 31), lane);
}

// ------------------------------------------------------------------------------------------- f32 MFMA (exact)
template <typename TC, int BM>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs g) {
    constexpr int BN = 128, BK = 16, PADM = BM + 4, PADN = BN + 4;
    constexpr int TM = BM / 64;
    constexpr int A_CHUNKS = BM * 4 / 256, B_CHUNKS = BN * 4 / 256;   // float4 chunks per thread
    __shared__ float As[2][BK * PADM];                                // k-major: As[k][row]
    __shared__ float Bs[2][BK * PADN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int bm = (tile / g.tiles_n) * BM, bn = (tile % g.tiles_n) * BN;
    const float* A = (const float*)g.A;
    const float* W = (const float*)g.W;

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4_g ra[A_CHUNKS], rb[B_CHUNKS];
    auto ld4g = [&](const float* p, bool ok) -> f32x4_g {
        if (!ok) return f32x4_g{0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4_g*>(p);
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            ra[i] = ld4g(A + (long)(bm + row) * g.lda + k0 + kc, (bm + row < g.M) && (k0 + kc < g.K));
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            rb[i] = ld4g(W + (long)(bn + row) * g.ldw + k0 + kc, (bn + row < g.N) && (k0 + kc < g.K));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            As[buf][(kc + 0) * PADM + row] = ra[i].x; As[buf][(kc + 1) * PADM + row] = ra[i].y;
            As[buf][(kc + 2) * PADM + row] = ra[i].z; As[buf][(kc + 3) * PADM + row] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = (c & 3) * 4;
            Bs[buf][(kc + 0) * PADN + row] = rb[i].x; Bs[buf][(kc + 1) * PADN + row] = rb[i].y;
            Bs[buf][(kc + 2) * PADN + row] = rb[i].z; Bs[buf][(kc + 3) * PADN + row] = rb[i].w;
        }
    };
    const int nk = (g.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            const int k = 2 * k2 + (lane >> 5);
            float af[TM], bfr[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[buf][k * PADM + wm * (BM / 2) + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = Bs[buf][k * PADN + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            epilogue_store<TC>(g, acc[i][j], bm + wm * (BM / 2) + i * 32, bn + wn * 64 + j * 32 + (lane & 31), lane);
}

// C = act(A . W^T + bias) + residual.   A (M,K) lda, dtype a_dtype;  W (N,K) ldw, dtype w_dtype (selects the
// arithmetic mode);  bias (N) f32 or NULL;  residual (M,N) ldr, dtype c_dtype, or NULL;  C (M,N) ldc, c_dtype.
// Constraints: K % 8 == 0; 16-byte aligned row starts (lda*sizeof % 16 == 0 etc.);  w f32 requires a f32.
extern "C" int psalm_gemm(const void* A, int a_dtype, long lda, const void* W, int w_dtype, long ldw, const float* bias,
                          const void* residual, long ldr, void* C, int c_dtype, long ldc, int M, int N, int K, int act,
                          int act_col_start, void* stream) {
    if (M == 0 || N == 0) return 0;
    PSALM_CHECK_ARG(K > 0 && K % 8 == 0, "psalm_gemm: K must be a positive multiple of 8");
    const long asz = a_dtype == PSALM_F32 ? 4 : 2, wsz = w_dtype == PSALM_F32 ? 4 : 2;
    PSALM_CHECK_ARG(((uintptr_t)A % 16 == 0) && (lda * asz) % 16 == 0, "psalm_gemm: A rows must be 16-byte aligned");
    PSALM_CHECK_ARG(((uintptr_t)W % 16 == 0) && (ldw * wsz) % 16 == 0, "psalm_gemm: W rows must be 16-byte aligned");
    PSALM_CHECK_ARG(!(w_dtype == PSALM_F32 && a_dtype != PSALM_F32), "psalm_gemm: fp32 weights need fp32 activations");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.res = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act; g.act_col_start = act_col_start;
    g.tiles_n = cdiv(N, 128);
    // 128-row tiles unless that leaves the 256 CUs under-filled
    const bool small = (long)cdiv(M, 128) * g.tiles_n < 256 || M <= 64;
    const int BM = small ? 64 : 128;
    g.tiles_m = cdiv(M, BM);
    const dim3 grid(g.tiles_m * g.tiles_n), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_BF16(TA, TC)                                                                             \
    do {                                                                                                \
        if (BM == 128) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TC, 128>), grid, block, 0, s, g);     \
        else hipLaunchKernelGGL((gemm_bf16_kernel<TA, TC, 64>), grid, block, 0, s, g);                \
    } while (0)
    if (w_dtype == PSALM_BF16) {
        if (a_dtype == PSALM_F32 && c_dtype == PSALM_F32) LAUNCH_BF16(float, float);
        else if (a_dtype == PSALM_F32 && c_dtype == PSALM_BF16) LAUNCH_BF16(float, bf16_t);
        else if (a_dtype == PSALM_BF16 && c_dtype == PSALM_F32) LAUNCH_BF16(bf16_t, float);
        else if (a_dtype == PSALM_BF16 && c_dtype == PSALM_BF16) LAUNCH_BF16(bf16_t, bf16_t);
        else { psalm_set_error("psalm_gemm: bad dtype"); return -1; }
    } else if (w_dtype == PSALM_F32) {
        if (c_dtype == PSALM_F32) {
            if (BM == 128) hipLaunchKernelGGL((gemm_f32_kernel<float, 128>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<float, 64>), grid, block, 0, s, g);
        } else if (c_dtype == PSALM_BF16) {
            if (BM == 128) hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, 128>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, 64>), grid, block, 0, s, g);
        } else { psalm_set_error("psalm_gemm: bad dtype"); return -1; }
    } else { psalm_set_error("psalm_gemm: bad weight dtype"); return -1; }
#undef LAUNCH_BF16
    PSALM_LAUNCH_END("psalm_gemm");
}
