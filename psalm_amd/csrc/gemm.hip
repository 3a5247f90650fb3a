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
#include <algorithm>
#include <atomic>
#include <cstring>
#include <type_traits>

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_GELU 2
#define ACT_GELU_NEW 3
#define ACT_POST_RESIDUAL 16   // flag: apply the activation AFTER adding the residual (ResNet block: relu(out + residual))
#define ACT_BIAS_ROW 32        // flag: bias is indexed by the output ROW (C = W_x . X^T products, i.e. transposed projections)

// Phase stamps for tools/experiments/gemm_timeline.py, which builds a SEPARATE copy of this library with -DPSALM_GEMM_TIMELINE (block-level
// time line of the direct-to-LDS kernel: entry / prologue issued / first tile visible / K loop done / tile in LDS / stored); the product
// build compiles PSALM_TL() to nothing.
#ifdef PSALM_GEMM_TIMELINE
__device__ unsigned long long* g_psalm_tl = nullptr;
extern "C" int psalm_gemm_timeline_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_psalm_tl), &p, sizeof(p)); }
#define PSALM_TL(i)                                                                                                                      \
    do {                                                                                                                                 \
        if (g_psalm_tl && threadIdx.x == 0)                                                                                              \
            g_psalm_tl[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                    \
    } while (0)
#define PSALM_TL_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// ... and, in the same experiment build only, ABLATION of the slice-form phased K loop (tools/experiments/gemm_timeline.py --ablate): bit 1 no
// global -> LDS copies, 2 no LDS fragment reads, 4 no matrix instructions, 8 no phase barriers -- results are garbage, the K-loop time of
// each combination says which resource the loop waits for.  The product build compiles PSALM_ABL() to the constant 0.
__device__ int g_psalm_ablate = 0;
extern "C" int psalm_gemm_ablate(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_psalm_ablate), &v, sizeof(v)); }
#define PSALM_ABL() g_psalm_ablate
#else
#define PSALM_TL(i) do { } while (0)
#define PSALM_TL_DRAIN() do { } while (0)
#define PSALM_ABL() 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct alignas(16) u32x4_s { unsigned x, y, z, w; };
struct alignas(8) u32x2_s { unsigned x, y; };
struct alignas(16) f32x4_g { float x, y, z, w; };

// erf to < 1 ulp without the library routine's divergent ranges (the activation runs once per output element in the epilogue's critical
// path): both range polynomials are evaluated and selected per lane -- |a| > 0.9277: 1 - exp(p(|a|)), else a + a q(a^2)  (the minimax
// polynomials of N. Juffa's vectorisable erff; checked against scipy.special.erf over [-6, 6]: 3.3e-8 relative in exact arithmetic).
__device__ __forceinline__ float psalm_erff(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    // (explicit fmaf: the expression is instantiated in several epilogues whose outputs are compared bit for bit -- r04b: the compiler had
    //  contracted 0.5 v (1 + erf) differently in two of them, 1 ulp apart on the hardware)
    if (act == ACT_GELU) { const float hv = 0.5f * v; return fmaf(hv, psalm_erff(v * 0.70710678118654752440f), hv); }
    if (act == ACT_GELU_NEW) {
        // 0.5 v (1 + tanh u) = v / (1 + exp(-2u)): one v_exp_f32 + one v_rcp_f32 instead of the library tanhf (~4x the instructions; the
        // activation runs once per output element in the epilogue's critical path -- 65536 per 256 x 256 tile).  exp(-2u) = inf for
        // u << 0 gives v * 0 = -0, the limit; a few ulp from the tanh form, both a few ulp from the exact value.
        // exp(-2u) = 2^(v (c1 + c2 v^2)),  c1 = -2 sqrt(2/pi) log2(e),  c2 = 0.044715 c1: the constants folded by hand (no fast-math
        // reassociation), v_exp_f32 is base 2, v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division __fdividef expands to here
        const float p = fmaf(v * v, -0.10294324f, -2.3022082f);
        return v * psalm_rcp(1.f + psalm_exp2(v * p));
    }
    return v;
}

// compile-time activation (A >= 0) for the straight-line epilogues; A < 0: the launch's run-time code
template <int A> __device__ __forceinline__ float apply_act_t(float v, int act_rt) { return apply_act(v, A < 0 ? act_rt : A); }
template <int V> struct psalm_ic { static constexpr int value = V; };

__device__ __forceinline__ unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }

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
    int row_fast;       // 1: consecutive tile ids walk the M tiles of one N column first (share the W tile in one XCD's L2)
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
    const bool brow = (g.act & ACT_BIAS_ROW) != 0;
    const float b = (g.bias && !brow) ? g.bias[col] : 0.f;
    const int act = g.act & 15;
    const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
    const bool do_act = act != ACT_NONE && col >= g.act_col_start;
    TC* C = (TC*)g.C;
    const TC* R = (const TC*)g.res;
    // residual / per-row bias of all 16 rows fetched up front, unconditionally (row clamped): inside the `row < g.M` guard each
    // load became its own branch + s_waitcnt vmcnt(0), 16 serialized round trips per tile (r01 ISA audit)
    float rv[16], rb[16];
    if (R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = ldf(R + (long)min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), g.M - 1) * g.ldr + col);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    if (brow && g.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[r] = g.bias[min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), g.M - 1)];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[r] = b;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r] + rb[r];
        if (do_act && !post) v = apply_act(v, act);
        v += rv[r];
        if (do_act && post) v = apply_act(v, act);
        if (row < g.M) stf(C + (long)row * g.ldc + col, v);
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


// ------------------------------------------------------------------------------------------- bf16 MFMA, direct-to-LDS
// The fast path (A and W both bf16 in memory, K % 64 == 0): operands go HBM/L2 -> LDS with global_load_lds_dwordx4
// (no staging registers, no ds_write pass), LDS double-buffered, ONE barrier per 64-deep K step; the next tile's
// copies are issued before the current tile's MFMAs and drained by the barrier's vmcnt(0).
// LDS image per operand tile: [rows][64 k] bf16 = 128-byte rows, written 8 rows (1 KiB) per wave instruction.  A
// linear image would put the 16 rows of a ds_read_b128 lane group on only two 16-byte slots of the 256-byte bank row
// (8-way conflict); the image is therefore XOR-swizzled: 16-byte slot p of row r holds k-chunk  p ^ ((r >> 1) & 7).
// global_load_lds writes lane-linearly, so the permutation is applied to the per-lane GLOBAL source address
// (still one full 128-byte line per 8 lanes) and again on the fragment reads (same involution, guide §5.4 rule 21).
// Rows beyond M / N are clamped to the last valid row (their products land in C rows / columns that are never stored).
// Split-K (gridDim.y > 1): each z-slice writes its raw fp32 partial tile to a slab; splitk_reduce_kernel applies the
// epilogue.  Used when the tile grid alone cannot fill the 256 CUs (small M or N with a long K).
__device__ __forceinline__ void load8_f32(const float* p, float* d) {
    const f32x4_g a = reinterpret_cast<const f32x4_g*>(p)[0], b = reinterpret_cast<const f32x4_g*>(p)[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void load8_f32(const bf16_t* p, float* d) {
    const u32x4_s a = *reinterpret_cast<const u32x4_s*>(p);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        d[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void store8(float* p, const float* v) {
    reinterpret_cast<f32x4_g*>(p)[0] = f32x4_g{v[0], v[1], v[2], v[3]};
    reinterpret_cast<f32x4_g*>(p)[1] = f32x4_g{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
    *reinterpret_cast<u32x4_s*>(p) = u32x4_s{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}

struct GemmFastArgs {
    GemmArgs g;
    float* slab;        // split-K partials [splits][M][N] fp32 (nullptr when splits == 1)
    int k_per_split;    // multiple of 64
    int vec_store;      // 1: C / residual rows allow 8-element vector accesses (N % 8 == 0, 16-byte aligned rows)
    // implicit-GEMM convolution (CONV kernels): A is the NHWC input x (B,H,W,Cin); row m = (b, oy, ox) of the (B,Ho,Wo) output,
    // k = (ky, kx, c).  Cin % 64 == 0, so one 64-deep K tile lies inside one filter tap and the tap is block-uniform.
    int cH, cW, cC, cK, cS, cP, cHo, cWo;
    const bf16_t* zeros;   // >= 16 bytes of zeros: source of the padded (out-of-image) taps
    // per-row scales of A (M) and W (N) of the split-f16 forms; acc * a_scale[m] * w_scale[n]
    const float* a_scale;
    const float* w_scale;
    // split-f16 ("X3") operands: A / W rows are [hi (Kp) | lo (Kp)] f16 (psalm_split_f16); the K loop runs over the 3 Kp-long products
    // hi.hi + lo.hi + hi.lo, i.e. logical k in [0, 3Kp) reads A column (k < 2Kp ? k : k - 2Kp) and W column (k < Kp ? k : k - Kp)
    int x3_kp;
    // split-f16 OUTPUT (X3 kernels without split-K; psalm_gemm_x3_split): the columns >= so_col_start of act(A.W^T + bias) are written, instead
    // of to C, in the operand form the NEXT split-f16 GEMM reads -- row r of `so` (row stride ldso f16) holds hi at column
    // so_col_off + (col - so_col_start) and lo so_kp columns further, scaled by a per-row power of two 2^(12 - e), e = floor(log2 bound_r):
    //   bound_r = max(a_scale[r] * so_par[0] + so_par[1],  G * so_par[2] + so_par[3]),   G = max over ALL rows of a_scale (so_global != 0) or 0
    // an upper bound of |value| in row r that needs no pass over the output: |a_rk| < 2^14 a_scale[r] (psalm_split_f16's row scaling), so with
    // so_par[0] = 2^14 max_n sum_k |w_nk| and so_par[1] = max |bias| the first term bounds every column (|act(x)| <= |x| for the activations
    // here); the second term is the caller's bound for values written into the same rows by ANOTHER kernel (Phi: the attention output, a
    // convex combination of v rows).  The bound may be loose by 2^10 without loss: hi + lo keep 22 bits down to 2^-25 of the scaled range.
    // 1 / scale goes to so_inv[r].
    unsigned short* so = nullptr;
    long ldso = 0;
    int so_kp = 0, so_col_off = 0, so_col_start = 0, so_global = 0;
    // so_paired: the W rows (and bias / w_scale entries) >= so_col_start were PERMUTED by the caller inside every group of 64 -- physical row
    // 64 g + 32 b + n holds logical row 64 g + 2 n + b (psalm_gemm_x3_split, `paired`) -- so that the two 32-column MFMA tiles of a
    // wave hold ADJACENT logical columns in the same lane: the 2-byte outputs leave as 4-byte stores of 128-byte row segments straight from
    // the accumulators (the fp32 tiles' store pattern, r03b: 5.6 TB/s) instead of through the LDS transpose.
    int so_paired = 0;
    float* so_inv = nullptr;
    const float* so_par = nullptr;
    // split-K placement (r06): 1 = the K slice, not the tile, decides a block's XCD.  The hardware deals linear block ids round-robin over the 8
    // XCDs; with (tile, slice) = (blockIdx.x, blockIdx.y) every XCD held a run of TILES and walked their whole K range -- Phi [dense|fc2]: all 899
    // A rows per XCD, 8 x 38 MB of operand reads from the fabric for 129 MB of compulsory traffic (profiles/r05_pmc_hbm_traffic.json: 3.6x).
    // With slice = linear id % splits the blocks of one XCD share a K slice: each A / W slice is fetched into one XCD's L2 (two when splits = 4 ...).
    int xcd_ksplit = 0;
};
// scale / inverse scale of a split-f16 row from an upper bound of its magnitudes: bound * sc in [2^12, 2^13)
__device__ __forceinline__ void split_scale_from_bound(float bound, float& sc, float& inv) {
    bound = fminf(fmaxf(bound, 7.888609e-31f), 1.2676506e30f);                 // [2^-100, 2^100]: sc and inv stay normal numbers
    const unsigned eb = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;
    sc = __builtin_bit_cast(float, (266u - eb) << 23);
    inv = __builtin_bit_cast(float, (eb - 12u) << 23);
}

// Tile configurations (BM x BN, WM x WN waves, each wave (BM/WM) x (BN/WN) as 32x32x16 MFMA tiles):
//   256 x 256, 2 x 4 waves (512 threads, 128 KB LDS, 1 block/CU): large GEMMs -- half the L2->LDS bytes per flop of
//              a 128^2 tile (the 128^2 kernel needs ~64 B/clk/CU from L2 at full MFMA rate, beyond what L2 sustains);
//   128 x 128, 2 x 2 waves (256 threads, 64 KB LDS, 2 blocks/CU): mid-size GEMMs where 256^2 tiles cannot fill 256 CUs;
//    64 x 128, 2 x 2 waves: M <= 192.
template <int N> __device__ __forceinline__ void wait_vmcnt_le() {      // s_waitcnt vmcnt(N) with a compile-time N
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12 || N == 16 || N == 18 || N == 24, "add the literal");
    if constexpr (N == 0) PSALM_WAIT_VMCNT(0);
    else if constexpr (N == 4) PSALM_WAIT_VMCNT(4);
    else if constexpr (N == 6) PSALM_WAIT_VMCNT(6);
    else if constexpr (N == 8) PSALM_WAIT_VMCNT(8);
    else if constexpr (N == 12) PSALM_WAIT_VMCNT(12);
    else if constexpr (N == 16) PSALM_WAIT_VMCNT(16);
    else if constexpr (N == 18) PSALM_WAIT_VMCNT(18);
    else PSALM_WAIT_VMCNT(24);
}

// BK = 128: 256-byte LDS rows, 4 rows per 1 KiB copy, slot p of row r holds k-chunk p ^ (r & 15)  (half the K steps -- and
//           barriers / exposed copy latencies -- of BK = 64; 128 KB LDS, 1 block/CU: for grids that cannot fill 2 blocks/CU anyway);
// BK = 64: 128-byte LDS rows, 8 rows per 1 KiB copy, slot p of row r holds k-chunk p ^ ((r >> 1) & 7);
// BK = 32:  64-byte LDS rows, 16 rows per copy,       slot p of row r holds k-chunk p ^ ((r >> 2) & 3)   (same rule: the
//           16 rows of a ds_read_b128 lane group must land on 16 distinct 16-byte slots of the 256-byte bank row).
// PH8 = true (256 x 256, 2 x 4 waves, BK 64, 2 buffers): the K loop below is replaced by the 4-phases-per-K-tile schedule
// described at "PH8 schedule" further down -- the two wave rows run one barrier interval apart, so that on every SIMD one wave
// is in a pure-MFMA segment while its partner reads fragments / issues copies, and copies stay in flight across barriers.
// X3 = true: split-f16 operands (see GemmFastArgs::x3_kp): same 16-bit element traffic, copies and swizzle as the bf16 kernel; the K-tile
// source columns are remapped, the matrix instruction is v_mfma_f32_32x32x16_f16 and the epilogue applies the per-row power-of-two
// scales of A and W.  An fp32-class GEMM (22-bit operands, fp32 accumulate) at 1/3 of the f16 MFMA rate.
// X3 = 2 ("slice" form of the split-f16 GEMM, for the tile configurations without the phased loop): instead of walking the 3 Kp-long
// panel, a K step covers ONE 64-deep slice of the true K range and brings FOUR tiles (A hi, A lo, W hi, W lo) into the stage, from which
// the three products hi.hi + lo.hi + hi.lo are formed: 4 instead of 6 tile copies per slice, 2/3 of the LDS fragment reads per MFMA, and
// -- the point -- 3x the matrix work per barrier / per copy round trip.  The mid-size GEMMs of the image (Swin stage 2, pixel decoder:
// 12..48 K steps) ran their K loops at L2 latency with X3 = 1 (r02k: 100-200 TFLOP/s algorithmic on the 128^2 / 64x128 tiles).
// SO = true (X3 == 1 only): the epilogue can emit split-f16 output (GemmFastArgs::so; psalm_gemm_x3_split).  A separate instantiation so that
// the plain kernels' epilogue -- at the register limit on the 256 x 256 tile -- is untouched.
// PH8_ = 6 / 7 (r06, "LW"): the block carries 2 / 4 LOADER wavefronts besides its WM x WN matrix wavefronts.  The loaders issue every
// global -> LDS copy of the K loop (three stages, running one slice ahead of the slice being multiplied) and nothing else; the matrix waves issue
// no vector-memory instruction inside the loop.  Why (profiles/r06c_mid_ablation.jsonl, the generic loop with parts switched off): on the
// mid-size GEMMs the copies alone take as long as the matrix instructions alone, and the two ADD UP -- a wave that issues a global_load_lds
// stalls in issue until the CU's one address path has taken it (~34 clocks per 1 KiB piece per CU whoever issues), every wave of the block
// reaches its copies at the same point behind the barrier, and nothing feeds the matrix pipes meanwhile.  Interleaving the copies with the
// matrix instructions of the same wave (r06b, tools/experiments/r06_ilv_copy_issue.patch) made every shape 5 - 10 % SLOWER, and eight-wave
// 256 x 128 blocks whose waves all copy gained 0 - 9 %.  With loaders the matrix waves' loop holds no vector-memory instruction (the compiler
// then pipelines the fragment reads with counted lgkmcnt waits by itself) and the copies of slice t + 2 run beside the products of slice t.
// Measured (profiles/r06d_gemm_mid_sweep_loader_waves.json): 5 - 23 % on the long-K shapes whose tile count fits one round of blocks; nothing
// on short K (Kp <= 256: prologue and epilogue outweigh the loop) -- the copy path of a CU (~30 B / clock from L2 into LDS) stays the bound.
template <typename TC, int BM, int BN, int WM, int WN, int NS, bool CONV = false, int BK = 64, int PH8_ = 0, int X3 = 0,
          bool SO = false, bool PAIR = false>
__global__ void __launch_bounds__(64 * (WM * WN + (PH8_ == 6 ? 2 : (PH8_ == 7 ? 4 : 0)))) gemm_bf16_glds_kernel(GemmFastArgs fa) {
    constexpr int PH8 = (PH8_ >= 1 && PH8_ <= 4) ? PH8_ : 0;     // phased 256 x 256 schedules
    constexpr int NLW = PH8_ == 6 ? 2 : (PH8_ == 7 ? 4 : 0);     // loader wavefronts
    constexpr bool LW = NLW > 0;
    static_assert(PH8_ >= 0 && PH8_ <= 7 && PH8_ != 5, "PH8_: 0 generic loop, 1 - 4 phased 256 x 256 loops, 6 / 7 loader waves");
    static_assert(!LW || (X3 == 2 && (NS == 3 || NS == 2) && BK == 32 && !CONV), "LW: slice form on 32-deep slices, three (or two) stages");
    static_assert(!PAIR || SO, "paired stores: a form of the split-f16 output");
    static_assert(!SO || X3 == 1 || (X3 == 2 && BK == 32 && (NS == 2 || LW)), "split-f16 output: K-panel form, or 32-deep slices in two stages (three with loader waves)");
    static_assert(X3 >= 0 && X3 <= 2, "X3: 0 bf16 operands, 1 split-f16 K-panel form, 2 split-f16 slice form");
    static_assert(!PH8 || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && NS == 2 && (BK == 64 || (BK == 32 && X3 == 2 && (PH8 == 3 || PH8 == 4))) && !CONV), "PH8 configuration");
    static_assert(!X3 || !CONV, "split-f16 variants: plain GEMM");
    static_assert(X3 != 1 || BK == 64, "split-f16 K-panel form: 64-deep K tiles");
    static_assert(X3 != 2 || ((!PH8 && (BK == 64 || BK == 32)) || ((PH8 == 3 || PH8 == 4) && BK == 32)), "split-f16 slice form: generic K loop, or the phased 256 x 256 loop on 32-deep slices");
    constexpr int XS = X3 == 2 ? 2 : 1;                          // operand images per stage (slice form: hi and lo)
    const GemmArgs& g = fa.g;
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RPC = 512 / BK;                                // rows per 1 KiB copy (8 | 16)
    constexpr int SLOTS = BK / 8;                                // 16-byte slots per row (8 | 4)
    constexpr int SWS = BK == 128 ? 0 : (BK == 64 ? 1 : 2);      // swizzle: slot ^= (row >> SWS) & (SLOTS - 1)
    static_assert(BK == 128 || BK == 64 || BK == 32, "BK");
    static_assert(!CONV || BK == 64, "implicit-GEMM convolution uses 64-deep K tiles");
    constexpr int NCW = LW ? NLW : NW;                           // wavefronts that issue copies
    constexpr int A_CH = BM / RPC / NCW, B_CH = BN / RPC / NCW;  // 1 KiB copies per copying wave per tile
    static_assert(A_CH >= 1 && B_CH >= 1 && TM >= 1 && TN >= 1, "tile / wave configuration");
    constexpr int SMEM_BYTES = NS * (BM + BN) * BK * 2 * XS;  // NS-deep ring of operand tiles
    constexpr int EP = (BM * BN * 4 > SMEM_BYTES) ? WM : 1;     // epilogue passes (one wave-row of the tile per pass)
    static_assert(BM / EP * BN * 4 <= SMEM_BYTES, "epilogue slab must fit in the operand buffers");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    __shared__ __attribute__((aligned(16))) bf16_t smem[NS][(BM + BN) * BK * XS];

    const int tid = threadIdx.x, lane = tid & 63;
    PSALM_TL(0);
    const int wave = (PH8 || LW) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;   // PH8 / LW: scalar (wave-dependent barriers / roles)
    const int cw = LW ? max(wave - NW, 0) : wave;                // index among the copying wavefronts
    const int wm = wave / WN, wn = wave % WN;
    int tile, ksl;                                               // tile id, K-slice index of this block
    if (fa.xcd_ksplit) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y;
        ksl = lin % (int)gridDim.y;
        tile = lin / (int)gridDim.y;
    } else {
        tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
        ksl = blockIdx.y;
    }
    const int bm = (g.row_fast ? tile % g.tiles_m : tile / g.tiles_n) * BM;
    const int bn = (g.row_fast ? tile / g.tiles_m : tile % g.tiles_n) * BN;
    const int kbeg = ksl * fa.k_per_split;
    const int kend = min(g.K, kbeg + fa.k_per_split);
    const bf16_t* A = (const bf16_t*)g.A;
    const bf16_t* W = (const bf16_t*)g.W;

    // per-lane global sources of this wave's chunks (chunk c covers tile rows 8c..8c+7; lane -> row 8c + lane/8, slot lane%8)
    constexpr bool PHS = PH8 != 0 && X3 == 2;                    // phased loop on 32-deep slices: 16-row chunks, one (hi, lo) copy pair per wave per half
    // A chunk of copy i.  PHS: copy q of every wave together = "A half q" = m-tiles {2q, 2q+1} of BOTH wave rows (16-row groups 4q..4q+3
    // and 8+4q..8+4q+3, one per wave).
    auto a_chunk = [&](int i) -> int {
        if constexpr (PHS) return 4 * i + (wave & 3) + 8 * (wave >> 2);
        else return cw + NCW * i;
    };
    const bf16_t* asrc[A_CH];
    const bf16_t* bsrc[B_CH];
    const int lrow = lane / SLOTS, slot = lane % SLOTS;
    int ay[A_CH], ax[A_CH];                                      // CONV: top-left input pixel of this lane's output pixel
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int r = a_chunk(i) * RPC + lrow;
        const int kc = slot ^ ((r >> SWS) & (SLOTS - 1));
        const int m = min(bm + r, g.M - 1);
        if constexpr (CONV) {
            const int ox = m % fa.cWo, oy = (m / fa.cWo) % fa.cHo, b = m / (fa.cWo * fa.cHo);
            ay[i] = oy * fa.cS - fa.cP;
            ax[i] = ox * fa.cS - fa.cP;
            asrc[i] = A + (long)b * fa.cH * fa.cW * fa.cC + kc * 8;           // + ((y*W + x)*C + c0) per K tile
        } else {
            ay[i] = ax[i] = 0;
            asrc[i] = A + (long)m * g.lda + (X3 == 1 ? 0 : kbeg) + kc * 8;
        }
    }
    // 1 KiB copy i of this wave covers W-tile rows 8 * b_chunk(i) ...  PH8: copies {2h, 2h+1} of every wave together cover the
    // columns of n-tile h of all four wave columns (= "B half h": the rows one phase reads), two copies per wave per half.
    auto b_chunk = [&](int i) -> int {
        if constexpr (PHS) return 4 * (wave >> 1) + 2 * i + (wave & 1);      // copy j of every wave = "B half j": n-tile j of all four wave columns
        else if constexpr (PH8) { const int e = 2 * wave + (i & 1); return 8 * (e >> 2) + 4 * (i >> 1) + (e & 3); }
        else return cw + NCW * i;
    };
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int r = b_chunk(i) * RPC + lrow;
        const int kc = slot ^ ((r >> SWS) & (SLOTS - 1));
        bsrc[i] = W + (long)min(bn + r, g.N - 1) * g.ldw + (X3 == 1 ? 0 : kbeg) + kc * 8;
    }
    // operand column of the K tile at offset koff of this block's K range (identity except for the split-f16 variant)
    auto x3_acol = [&](int koff) -> int {
        if constexpr (X3 == 1) { const int k = kbeg + koff; return k < 2 * fa.x3_kp ? k : k - 2 * fa.x3_kp; }
        else return koff;
    };
    auto x3_wcol = [&](int koff) -> int {
        if constexpr (X3 == 1) { const int k = kbeg + koff; return k < fa.x3_kp ? k : k - fa.x3_kp; }
        else return koff;
    };
    auto mma16 = [&](const bf16x8& a_, const bf16x8& b_, const f32x16& c_) -> f32x16 {
        if constexpr (X3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0);
    };
    auto issue = [&](int buf, int koff) {
        bf16_t* As = smem[buf];
        bf16_t* Bs = smem[buf] + BM * BK;
        if constexpr (CONV) {
            const int k0 = kbeg + koff, tap = k0 / fa.cC, c0 = k0 - tap * fa.cC;   // block-uniform filter tap of this K tile
            const int ky = tap / fa.cK, kx = tap - ky * fa.cK;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int y = ay[i] + ky, x = ax[i] + kx;
                const bool in = y >= 0 && y < fa.cH && x >= 0 && x < fa.cW;
                const bf16_t* src = in ? asrc[i] + ((long)y * fa.cW + x) * fa.cC + c0 : fa.zeros;
                psalm_glds16(src, As + a_chunk(i) * RPC * BK);
            }
        } else {
            const int ka = x3_acol(koff);
#pragma unroll
            for (int i = 0; i < A_CH; ++i) psalm_glds16(asrc[i] + ka, As + a_chunk(i) * RPC * BK);
        }
        const int kw = x3_wcol(koff);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) psalm_glds16(bsrc[i] + kw, Bs + b_chunk(i) * RPC * BK);
        if constexpr (X3 == 2) {                                 // slice form: the lo images of the same K slice, behind the hi images
            bf16_t* Al = smem[buf] + (BM + BN) * BK;
            bf16_t* Bl = Al + BM * BK;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) psalm_glds16(asrc[i] + fa.x3_kp + koff, Al + a_chunk(i) * RPC * BK);
#pragma unroll
            for (int i = 0; i < B_CH; ++i) psalm_glds16(bsrc[i] + fa.x3_kp + koff, Bl + b_chunk(i) * RPC * BK);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row (lane&31) of a 32-row sub-tile, k-chunk (2*kk + hi) ^ f,  f = ((lane&31) >> SWS) & (SLOTS-1)
    const int n32 = lane & 31, hi = lane >> 5, fsw = (n32 >> SWS) & (SLOTS - 1);
    // r05: the operands of the direct fp32 epilogue that do not depend on the K loop -- the column scales / bias of the wave's TN columns, the
    // row scales of its first m-tile -- are fetched BEFORE the loop (20 registers on the 64 x 128 tile): behind it they were one more exposed
    // global round trip in launches of 17..50 us.  (Not on the phased 256 x 256 kernels: no registers to spare; not for split-f16 output tiles.)
    constexpr bool EPF = std::is_same<TC, float>::value && !PH8 && !SO && !CONV;
    float pf_wsc[TN], pf_bias[TN], pf_asc[16];
    if constexpr (EPF) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = min(bn + wn * (BN / WN) + j * 32 + n32, g.N - 1);
            pf_wsc[j] = 1.f;
            if constexpr (X3) pf_wsc[j] = fa.w_scale[col];
            pf_bias[j] = (fa.slab == nullptr && g.bias && !(g.act & ACT_BIAS_ROW)) ? g.bias[col] : 0.f;   // (a per-ROW bias has M entries: LDS path)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pf_asc[r] = 1.f;
            if constexpr (X3) pf_asc[r] = fa.a_scale[min(bm + wm * (BM / WM) + 4 * hi + (r & 3) + 8 * (r >> 2), g.M - 1)];
        }
    }
    const int a_row0 = wm * (BM / WM) + n32, b_row0 = wn * (BN / WN) + n32;

    // NS-deep ring, prefetch distance NS-1 tiles, ONE barrier per K step.  At the top of step kt the copies of tiles
    // kt .. kt+NS-2 are in flight; a counted vmcnt retires exactly tile kt (the copies of later tiles stay in flight
    // across the raw barrier), the barrier makes every wave's tile-kt data visible and proves that all waves are done
    // with the stage read in step kt-1, which is then refilled with tile kt+NS-1.
    constexpr int LPT = (A_CH + B_CH) * XS;                      // copy instructions per wave per tile
    const int nk = (kend - kbeg) / BK;
    if constexpr (LW) {
        // ---- loader / matrix wavefronts (see the kernel comment).  Stage t % 3 holds slice t.  Barrier B(t), reached by every wave once per
        // slice:  loaders arrive when slice t has LANDED (counted vmcnt: slice t + 1 may still be in flight), matrix waves when they are done
        // READING slice t - 1 (lgkmcnt(0)).  After B(t) the loaders refill stage (t + 2) % 3 = (t - 1) % 3 -- whose last reads retired before B(t)
        // -- with slice t + 2 while the matrix waves multiply slice t.  (Two stages, the 256 x 256 block: the refill after B(t) is slice t + 1 into the
        // stage slice t - 1 was read from, and the loaders wait for ALL of it before B(t + 1).)
        if (wave >= NW) {
#pragma unroll
            for (int p = 0; p < NS - 1; ++p)
                if (p < nk) issue(p, p * BK);
            PSALM_TL(1);
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                if (NS == 3 && kt + 1 < nk) wait_vmcnt_le<LPT>(); else wait_vmcnt_le<0>();     // (two stages: nothing else is in flight)
                __builtin_amdgcn_s_barrier();
                if (kt + NS - 1 < nk && !(PSALM_ABL() & 1)) issue((kt + NS - 1) % NS, (kt + NS - 1) * BK);
            }
            return;                                              // (every copy has landed: the last wait was vmcnt(0))
        }
        // a matrix wave whose rows all lie in the padding below row M (Phi's M = 899 on 256-row tiles) keeps the barriers and leaves out its reads and
        // products: those rows are never stored (the phased kernel's PH8 = 4 idea, per wave here)
        const bool idle = __builtin_amdgcn_readfirstlane((int)(bm + wm * (BM / WM) >= g.M)) != 0;
        const bf16_t* As = smem[0];
        int stg = 0;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            PSALM_RAW_BARRIER();
            if (kt == 0) PSALM_TL(2);
            const bf16_t* Bs = As + BM * BK;
            const bf16_t* Al = As + (BM + BN) * BK;
            const bf16_t* Bl = Al + BM * BK;
            if (!idle) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
                bf16x8 ah[TM], al[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As[(a_row0 + 32 * i) * BK + co]));
                    al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Al[(a_row0 + 32 * i) * BK + co]));
                }
                if constexpr (TN <= 2) {
                    bf16x8 bh[TN], bl[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs[(b_row0 + 32 * j) * BK + co]));
                        bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bl[(b_row0 + 32 * j) * BK + co]));
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = mma16(ah[i], bh[j], acc[i][j]);
                            acc[i][j] = mma16(al[i], bh[j], acc[i][j]);
                            acc[i][j] = mma16(ah[i], bl[j], acc[i][j]);
                        }
                } else {                                         // wide wave tiles: the W fragments of one n-tile at a time (register budget of three waves per SIMD)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs[(b_row0 + 32 * j) * BK + co]));
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bl[(b_row0 + 32 * j) * BK + co]));
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            acc[i][j] = mma16(ah[i], bh, acc[i][j]);
                            acc[i][j] = mma16(al[i], bh, acc[i][j]);
                            acc[i][j] = mma16(ah[i], bl, acc[i][j]);
                        }
                    }
                }
            }
            }
            stg = stg + 1 == NS ? 0 : stg + 1;
            As = smem[stg];                                      // next stage
        }
    } else if constexpr (PHS) {
        // ---- PH8 schedule on 32-deep SLICES of the split-f16 operands (r04).  A stage holds the FOUR images of one slice of the true K range
        // (A hi | W hi | A lo | W lo, 16 KB each: the 128 KB of the K-panel form's two 64-deep stages) and a phase forms the three products
        // hi.hi + lo.hi + hi.lo of its 64 x 32 quadrant: 12 matrix instructions per phase instead of 8, per 2 copies and 6 + 6 fragment reads
        // -- 2/3 of the K-panel form's L2 -> LDS bytes and LDS fragment reads per product, 2/3 of its barriers, and W hi is fetched ONCE (the
        // K-panel form walks it twice, 32 K steps apart: its second pass missed the XCD's L2 -- counter traffic 1.8x compulsory, r03p).
        // Halves, phases, refill points, hazards and the vmcnt counts are those of the 64-deep schedule below: a half is the SAME rows, now
        // as a (hi, lo) pair of 16-row chunks per wave instead of two 8-row chunks.
        static_assert(PH8 == 3 || PH8 == 4, "slice form: copies inside the MFMA segment, bare barriers");
        constexpr int LO = (BM + BN) * BK;                       // the lo images sit behind the hi images of a stage
        // PH8 == 4 (psalm_gemm_set_tile_policy(2582), the default since r04p): the wave leaves out the matrix instructions and fragment
        // reads of its 32-row m-tiles that lie entirely in the padding below row M.  Those rows are never stored, so no output changes; the
        // launch does not get shorter either (the full tiles set its duration) -- the point is power: Phi's M = 899 puts 12 % of the matrix
        // instructions of [k|v|q|fc1] on padding rows (wave row 1 of the last row of tiles: 3 valid rows of 128), and the launch is clock-limited
        // (1.76 GHz by the r04j SQ-counter pass).  Phases, barriers, copies and waits are untouched.
        constexpr bool SKIP_PAD = PH8 == 4;
        int mt_valid = 4;                                        // m-tiles of this wave that hold at least one row < M
        if constexpr (SKIP_PAD) {
            const int rows_left = g.M - bm - wm * (BM / WM);
            mt_valid = __builtin_amdgcn_readfirstlane(rows_left <= 0 ? 0 : (rows_left >= 128 ? 4 : (rows_left + 31) >> 5));
        }
        const int abl = PSALM_ABL();                             // (experiment build only; the constant 0 in the product)
        bf16x8 ah[2][2] = {}, al[2][2] = {}, bh[2] = {}, bl[2] = {};   // [m-tile of the half][kk] / [kk]
        // (`part`: std::true_type in the copy of the K loop that a wave with all-padding m-tiles runs -- PH8 == 4 only; the other waves, and
        //  every wave of the product kernel, run the copy in which the tests below are compile-time constants)
        auto read_a = [&](auto part, const bf16_t* As_, int q) __attribute__((always_inline)) {
            if (abl & 2) return;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (decltype(part)::value && 2 * q + i >= mt_valid) continue;
                    const int off = (a_row0 + 32 * (2 * q + i)) * BK + co;
                    ah[i][kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As_[off]));
                    al[i][kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As_[LO + off]));
                }
            }
        };
        auto read_b = [&](const bf16_t* Bs_, int j) {
            if (abl & 2) return;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int off = (b_row0 + 32 * j) * BK + ((2 * kk + hi) ^ fsw) * 8;
                bh[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs_[off]));
                bl[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs_[LO + off]));
            }
        };
        // half copies: which = 0 the hi image's chunk, 1 the lo image's, 2 both
        auto stage_a = [&](int buf, int koff, int q, int which = 2) {
            if (abl & 1) return;
            bf16_t* d = smem[buf] + a_chunk(q) * RPC * BK;
            if (which != 1) psalm_glds16(asrc[q] + koff, d);
            if (which != 0) psalm_glds16(asrc[q] + fa.x3_kp + koff, d + LO);
        };
        auto stage_b = [&](int buf, int koff, int j, int which = 2) {
            if (abl & 1) return;
            bf16_t* d = smem[buf] + BM * BK + b_chunk(j) * RPC * BK;
            if (which != 1) psalm_glds16(bsrc[j] + koff, d);
            if (which != 0) psalm_glds16(bsrc[j] + fa.x3_kp + koff, d + LO);
        };
        // 12 matrix instructions, the two accumulators of the quadrant alternating; the phase's two copies after the 2nd and the 8th
        auto mma = [&](auto part, int q, int j, auto&& copy) __attribute__((always_inline)) {
            if (abl & 4) { copy(0); copy(1); return; }
            const bool on0 = !decltype(part)::value || 2 * q < mt_valid, on1 = !decltype(part)::value || 2 * q + 1 < mt_valid;     // (wave-uniform)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (on0) acc[2 * q][j] = mma16(ah[0][kk], bh[kk], acc[2 * q][j]);
                if (on1) acc[2 * q + 1][j] = mma16(ah[1][kk], bh[kk], acc[2 * q + 1][j]);
                __builtin_amdgcn_sched_barrier(0);
                copy(kk);
                __builtin_amdgcn_sched_barrier(0);
                if (on0) acc[2 * q][j] = mma16(al[0][kk], bh[kk], acc[2 * q][j]);
                if (on1) acc[2 * q + 1][j] = mma16(al[1][kk], bh[kk], acc[2 * q + 1][j]);
                if (on0) acc[2 * q][j] = mma16(ah[0][kk], bl[kk], acc[2 * q][j]);
                if (on1) acc[2 * q + 1][j] = mma16(ah[1][kk], bl[kk], acc[2 * q + 1][j]);
            }
        };
#define PHS_ENTER_MFMA() do { __builtin_amdgcn_sched_barrier(0); if (!(abl & 8)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
                              __builtin_amdgcn_s_setprio(1); } while (0)
#define PHS_LEAVE_MFMA() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); if (!(abl & 8)) __builtin_amdgcn_s_barrier(); \
                              __builtin_amdgcn_sched_barrier(0); } while (0)
        // mode 0: steady state (tile t+2 exists);  1: t = nk-2 (only B0 of tile t+1 left to copy; drain);  2: t = nk-1
        auto tile_phases = [&](auto part, int t, int mode) __attribute__((always_inline)) {
            const int cur = t & 1;
            const bf16_t* As_ = smem[cur];
            const bf16_t* Bs_ = smem[cur] + BM * BK;
            read_b(Bs_, 0);                                      // P1
            read_a(part, As_, 0);
            PHS_ENTER_MFMA();
            mma(part, 0, 0, [&](int w_) { if (mode <= 1) stage_b(cur ^ 1, (t + 1) * BK, 0, w_); });
            PHS_LEAVE_MFMA();
            read_b(Bs_, 1);                                      // P2
            PHS_ENTER_MFMA();
            mma(part, 0, 1, [&](int w_) { if (mode == 0) stage_a(cur, (t + 2) * BK, 0, w_); });
            PHS_LEAVE_MFMA();
            read_a(part, As_, 1);                                      // P3
            PHS_ENTER_MFMA();
            mma(part, 1, 1, [&](int w_) { if (mode == 0) stage_b(cur, (t + 2) * BK, 1, w_); });
            PHS_LEAVE_MFMA();
            read_b(Bs_, 0);                                      // P4
            // everything but the 2 most recent halves has landed = all of tile t+1 (this phase's copies are issued after the wait)
            if (mode == 0) wait_vmcnt_le<4>();
            else if (mode == 1) wait_vmcnt_le<0>();
            PHS_ENTER_MFMA();
            mma(part, 1, 0, [&](int w_) { if (mode == 0) stage_a(cur, (t + 2) * BK, 1, w_); });
            PHS_LEAVE_MFMA();
        };
        stage_a(0, 0, 0); stage_a(0, 0, 1); stage_b(0, 0, 0); stage_b(0, 0, 1);        // tile 0 (8 copies per wave)
        stage_a(1, BK, 0); stage_b(1, BK, 1); stage_a(1, BK, 1);                       // tile 1 except B0 (6 copies)
        PSALM_TL(1);
        wait_vmcnt_le<6>();
        PSALM_RAW_BARRIER();
        PSALM_TL(2);
        if (wm == 1) PSALM_RAW_BARRIER();                        // wave row 1 starts one barrier interval late
        auto k_loop = [&](auto part) __attribute__((always_inline)) {
            int t = 0;
#pragma unroll 1
            for (; t + 2 < nk; ++t) tile_phases(part, t, 0);
            tile_phases(part, t, 1);
            tile_phases(part, t + 1, 2);
        };
        if constexpr (SKIP_PAD) {                                // two copies of the loop: the same barriers in both, a wave takes one of them
            if (mt_valid < 4) k_loop(std::true_type{});
            else k_loop(std::false_type{});
        } else {
            k_loop(std::false_type{});
        }
        if (wm == 0) PSALM_RAW_BARRIER();                        // pairs with wave row 1's last barrier
#undef PHS_ENTER_MFMA
#undef PHS_LEAVE_MFMA
    } else if constexpr (PH8) {
        // ---- PH8 schedule (256 x 256 tile, host guarantees nk >= 2).  A K tile is consumed in 4 phases, one 64 x 32 quadrant of
        // the wave's 128 x 64 output each (8 MFMAs):   P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0)
        // where A-half q = m-tiles {2q, 2q+1} of both wave rows (copies q, q+2 of every wave) and B-half j = n-tile j of all four
        // wave columns (copies 2j, 2j+1).  A phase is   [ds_read the fragments that change; issue ONE half-tile copy (2 per wave);
        // lgkmcnt(0)]  barrier  [8 MFMAs at raised priority]  barrier.   Wave row 1 runs one barrier interval behind wave row 0
        // (one extra barrier before its first phase, one after row 0's last), so each SIMD (waves w, w+4) always has one wave in
        // the MFMA segment and the other in the read/copy segment.
        // LDS hazards.  WAR: a half is refilled exactly one phase after its last ds_read; those reads retire (lgkmcnt(0)) before
        // the reading phase's first barrier, for the lagging row too (its first barrier of phase p is the leading row's second).
        // RAW: the only wait is vmcnt(6) in P4 -- everything but the 3 most recent halves has landed = all of tile t+1 -- placed
        // before P4's first barrier; the first read of tile t+1 is one phase later, after a barrier every wave's wait precedes.
        //   tile t:  P1 refills B0 of the OTHER buffer with tile t+1 (last read: P4 of tile t-1);  P2 / P3 / P4 refill A0 / B1 / A1
        //   of THIS buffer with tile t+2 (last read: P1 / P2 / P3 of tile t).  Copies never drain inside the loop.
        bf16x8 af[2][BK / 16], bq[BK / 16];
        auto read_a = [&](const bf16_t* As_, int q) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i][kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As_[(a_row0 + 32 * (2 * q + i)) * BK + co]));
            }
        };
        auto read_b = [&](const bf16_t* Bs_, int j) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
                bq[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs_[(b_row0 + 32 * j) * BK + co]));
            }
        };
        // half-tile copies: which = 0 / 1 selects one of the wave's two 1 KiB copies, 2 = both
        auto stage_a = [&](int buf, int koff, int q, int which = 2) {   // A-half q of tile (koff / BK) -> buffer buf
            bf16_t* As_ = smem[buf];
            const int ka = x3_acol(koff);
            if (which != 1) psalm_glds16(asrc[q] + ka, As_ + (wave + NW * q) * RPC * BK);
            if (which != 0) psalm_glds16(asrc[q + 2] + ka, As_ + (wave + NW * (q + 2)) * RPC * BK);
        };
        auto stage_b = [&](int buf, int koff, int j, int which = 2) {   // B-half j
            bf16_t* Bs_ = smem[buf] + BM * BK;
            const int kw = x3_wcol(koff);
            if (which != 1) psalm_glds16(bsrc[2 * j] + kw, Bs_ + b_chunk(2 * j) * RPC * BK);
            if (which != 0) psalm_glds16(bsrc[2 * j + 1] + kw, Bs_ + b_chunk(2 * j + 1) * RPC * BK);
        };
        // PH8 == 2: the phase's two copies are issued INSIDE the MFMA segment (after the 2nd and the 6th MFMA: the matrix pipe is
        // busy with the MFMA just issued while the copy is accepted), not in the read segment -- r01 PMC: the copies' issue stalls
        // (~80 cycles each behind the other waves' copies) made the read segment ~1.8x the MFMA segment it runs beside.
        auto mma = [&](int q, int j, auto&& copy) {
            {
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[2 * q + i][j] = mma16(af[i][kk], bq[kk], acc[2 * q + i][j]);
                    if (PH8 >= 2 && (kk == 0 || kk == 2)) {
                        __builtin_amdgcn_sched_barrier(0);
                        copy(kk >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        // PH8 == 3: as 2, and the fragment reads are NOT drained before the phase's first barrier (bare s_barrier; the compiler's own
        // lgkmcnt wait sits in front of the first MFMA): the LDS latency overlaps the wait for the partner row's MFMA segment.  Safe
        // only with the late copies of variants 2 / 3: a half is then refilled in the MFMA segment of the phase after its last read,
        // which the lagging row's readers reach the barrier in front of only after their own MFMA segment (= after their reads retired).
#define PH8_BAR() do { if constexpr (PH8 == 3) __builtin_amdgcn_s_barrier(); else PSALM_RAW_BARRIER(); } while (0)
#define PH8_ENTER_MFMA() do { __builtin_amdgcn_sched_barrier(0); PH8_BAR(); __builtin_amdgcn_sched_barrier(0); \
                              __builtin_amdgcn_s_setprio(1); } while (0)
#define PH8_LEAVE_MFMA() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); PH8_BAR(); \
                              __builtin_amdgcn_sched_barrier(0); } while (0)
        // mode 0: steady state (tile t+2 exists);  1: t = nk-2 (only B0 of tile t+1 left to copy; drain);  2: t = nk-1
        auto tile_phases = [&](int t, int mode) {
            const int cur = t & 1;
            const bf16_t* As_ = smem[cur];
            const bf16_t* Bs_ = smem[cur] + BM * BK;
            constexpr bool early = PH8 == 1;                     // copies in the read segment (1) or inside the MFMA segment (2)
            read_b(Bs_, 0);                                  // P1
            read_a(As_, 0);
            if (early && mode <= 1) stage_b(cur ^ 1, (t + 1) * BK, 0);
            PH8_ENTER_MFMA();
            mma(0, 0, [&](int w_) { if (mode <= 1) stage_b(cur ^ 1, (t + 1) * BK, 0, w_); });
            PH8_LEAVE_MFMA();
            read_b(Bs_, 1);                                  // P2
            if (early && mode == 0) stage_a(cur, (t + 2) * BK, 0);
            PH8_ENTER_MFMA();
            mma(0, 1, [&](int w_) { if (mode == 0) stage_a(cur, (t + 2) * BK, 0, w_); });
            PH8_LEAVE_MFMA();
            read_a(As_, 1);                                  // P3
            if (early && mode == 0) stage_b(cur, (t + 2) * BK, 1);
            PH8_ENTER_MFMA();
            mma(1, 1, [&](int w_) { if (mode == 0) stage_b(cur, (t + 2) * BK, 1, w_); });
            PH8_LEAVE_MFMA();
            read_b(Bs_, 0);                                  // P4
            if (early && mode == 0) stage_a(cur, (t + 2) * BK, 1);
            // everything but the most recent halves has landed = all of tile t+1  (PH8 == 1: 3 halves issued since tile t+1's B0;
            // PH8 == 2: 2 -- this phase's copies are issued after the wait)
            if (mode == 0) { if constexpr (early) wait_vmcnt_le<6>(); else wait_vmcnt_le<4>(); }
            else if (mode == 1) wait_vmcnt_le<0>();
            PH8_ENTER_MFMA();
            mma(1, 0, [&](int w_) { if (mode == 0) stage_a(cur, (t + 2) * BK, 1, w_); });
            PH8_LEAVE_MFMA();
        };
        stage_a(0, 0, 0); stage_a(0, 0, 1); stage_b(0, 0, 0); stage_b(0, 0, 1);        // tile 0 (8 copies per wave)
        stage_a(1, BK, 0); stage_b(1, BK, 1); stage_a(1, BK, 1);                       // tile 1 except B0 (6 copies)
        PSALM_TL(1);
        wait_vmcnt_le<6>();
        PSALM_RAW_BARRIER();
        PSALM_TL(2);
        if (wm == 1) PSALM_RAW_BARRIER();                        // wave row 1 starts one barrier interval late
        int t = 0;
#pragma unroll 1
        for (; t + 2 < nk; ++t) tile_phases(t, 0);
        tile_phases(t, 1);
        tile_phases(t + 1, 2);
        if (wm == 0) PSALM_RAW_BARRIER();                        // pairs with wave row 1's last barrier
#undef PH8_ENTER_MFMA
#undef PH8_LEAVE_MFMA
#undef PH8_BAR
    } else {
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(p, p * BK);
    PSALM_TL(1);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NS;
        if constexpr (NS == 2) {
            wait_vmcnt_le<0>();
        } else {
            const int ahead = min(nk - 1 - kt, NS - 2);          // tiles after kt whose copies have been issued
            if (NS >= 4 && ahead >= 2) wait_vmcnt_le<(NS >= 4 ? 2 * LPT : 0)>();
            else if (ahead >= 1) wait_vmcnt_le<LPT>();
            else wait_vmcnt_le<0>();
        }
        if (!(PSALM_ABL() & 8)) PSALM_RAW_BARRIER();
        if (kt == 0) PSALM_TL(2);
        if (kt + NS - 1 < nk && !(PSALM_ABL() & 1)) issue((kt + NS - 1) % NS, (kt + NS - 1) * BK);
        const bf16_t* As = smem[buf];
        const bf16_t* Bs = smem[buf] + BM * BK;
        // register double-buffered fragments: the ds_read_b128s of k-step kk+1 are issued BEFORE the MFMAs of k-step kk, so
        // the LDS latency hides behind TM*TN matrix instructions instead of stalling every group (the straightforward loop
        // compiled to `ds_read x4; s_waitcnt lgkmcnt(0); mfma x4` -- 45 % MFMA duty at best, r01 final PMC)
        bf16x8 af[2][TM], bfr[2][TN];
        auto load_frags = [&](int kk, int slot_) {
            const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[slot_][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As[(a_row0 + 32 * i) * BK + co]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[slot_][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs[(b_row0 + 32 * j) * BK + co]));
        };
        // (the scheduler otherwise sinks the reads back next to their first use to save registers; the pin costs TM+TN
        //  fragment registers, which the 256x256 configuration -- 253 VGPRs -- does not have)
        if constexpr (X3 == 2) {                                 // slice form: three products from the four images of this K slice
            const bf16_t* Al = As + (BM + BN) * BK;
            const bf16_t* Bl = Al + BM * BK;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
                bf16x8 ah[TM] = {}, al[TM] = {}, bh[TN] = {}, bl[TN] = {};
                if (!(PSALM_ABL() & 2)) {                        // (experiment build: fragment reads off)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&As[(a_row0 + 32 * i) * BK + co]));
                        al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Al[(a_row0 + 32 * i) * BK + co]));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bs[(b_row0 + 32 * j) * BK + co]));
                        bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_s*>(&Bl[(b_row0 + 32 * j) * BK + co]));
                    }
                }
                if (PSALM_ABL() & 4) continue;                   // (experiment build: matrix instructions off)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = mma16(ah[i], bh[j], acc[i][j]);
                        acc[i][j] = mma16(al[i], bh[j], acc[i][j]);
                        acc[i][j] = mma16(ah[i], bl[j], acc[i][j]);
                    }
            }
            continue;
        }
        constexpr bool PIN = (TM * TN <= 4);
        load_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) load_frags(kk + 1, (kk + 1) & 1);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mma16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j]);
        }
    }
    }
    PSALM_TL(3);
    // ---- fp32 output (C, or a split-K slab): straight from the accumulators.  In the accumulator layout a lane owns ONE column and 16 rows of
    // a 32 x 32 tile, so a 4-byte store instruction of the wave writes two full 128-byte row segments -- as fast as 16-byte row-contiguous
    // stores on this chip (tools/experiments/store_pattern.hip, r03b: 5.6 TB/s for the 224 tiles of Phi [k|v|q|fc1], 5.7 contiguous, 4.5 for
    // the 16-byte stores 32 bytes apart of the LDS path below) -- with no transpose pass, no barrier and no idle wave row.  r03a time line:
    // the LDS epilogue was 34 of 182 us of that launch and 21 of 52 us of M4096 N2048 K512.  Ragged edges: rows >= M / columns >= N are
    // dropped by the buffer descriptor's bounds check (no branch per element).  Split-f16 OUTPUT tiles (2-byte elements) keep the LDS path.
    if constexpr (std::is_same<TC, float>::value) {
        const bool split = fa.slab != nullptr;
        const int act = g.act & 15;
        // (erf / tanh activations stay on the LDS path: its store loop is rolled, here every element would get its own inlined copy; so
        //  do the few per-ROW-bias GEMMs of the mask decoder: 16 more registers per lane here for a path that small GEMMs take)
        bool lds_path = !split && ((act != ACT_NONE && act != ACT_RELU && bn + BN > g.act_col_start) || (g.act & ACT_BIAS_ROW) != 0);
        if constexpr (SO) lds_path = lds_path || (fa.so != nullptr && bn + BN > fa.so_col_start);
        if (!lds_path) {
            const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
            float* Cb = split ? fa.slab + (long)ksl * g.M * g.N : (float*)g.C;
            const long ldo = split ? (long)g.N : g.ldc;
            const int rows_here = min(g.M - bm, BM);
            const psalm_rsrc rc = psalm_make_rsrc(Cb + (long)bm * ldo, (unsigned)min((long)rows_here * ldo * 4, 0x7ffff000L));
            const float* R = split ? nullptr : (const float*)g.res;
            const psalm_rsrc rr = psalm_make_rsrc(R ? R + (long)bm * g.ldr : Cb, R ? (unsigned)min((long)rows_here * g.ldr * 4, 0x7ffff000L) : 0u);
            float wsc[TN], bias_c[TN];
            unsigned coff[TN];
            bool actc[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = bn + wn * (BN / WN) + j * 32 + n32;
                const bool ok = col < g.N;
                coff[j] = ok ? (unsigned)col * 4u : PSALM_BUF_OOB;
                if constexpr (EPF) {
                    wsc[j] = pf_wsc[j];
                    bias_c[j] = pf_bias[j];
                } else {
                    wsc[j] = 1.f;
                    if constexpr (X3) wsc[j] = fa.w_scale[min(col, g.N - 1)];
                    bias_c[j] = (!split && g.bias) ? g.bias[min(col, g.N - 1)] : 0.f;
                }
                actc[j] = !split && act != ACT_NONE && col >= g.act_col_start;
            }
            // Software-pipelined over the TM x TN accumulator tiles: per tile  compute 16 results -> issue the LOADS of the next tile
            // (residual; row scales at a new m-tile) -> issue this tile's 16 stores.  vmcnt retires in order, so a load issued behind a
            // tile's stores is only "landed" once those stores have completed -- the first form of this loop waited out a full store
            // drain per tile (vmcnt(15) in front of every store of a residual GEMM).
            auto row_off = [&](int i, int r) { return wm * (BM / WM) + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2); };   // tile-local row
            float asc[16], rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (EPF) asc[r] = pf_asc[r];
                else {
                    asc[r] = 1.f;
                    if constexpr (X3) asc[r] = fa.a_scale[min(bm + row_off(0, r), g.M - 1)];
                }
                // (unconditional: without a residual the descriptor has size 0 and every load returns 0 without a memory request -- as `R ? load : 0`
                //  the eight-wave instantiations compiled to a branch + s_waitcnt vmcnt(0) per load, 16 serialised round trips per tile)
                rv[r] = psalm_buf_load_f32(rr, (unsigned)((long)row_off(0, r) * g.ldr * 4) + coff[0]);
            }
#pragma unroll
            for (int t = 0; t < TM * TN; ++t) {
                const int i = t / TN, j = t % TN, i2 = (t + 1) / TN, j2 = (t + 1) % TN;
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    x[r] = acc[i][j][r];
                    if constexpr (X3) x[r] = fmaf(x[r], asc[r] * wsc[j], bias_c[j]);
                    else x[r] += bias_c[j];
                    if (actc[j] && !post) x[r] = fmaxf(x[r], 0.f);           // ReLU (the only activation on this path)
                    x[r] += rv[r];
                    if (actc[j] && post) x[r] = fmaxf(x[r], 0.f);
                }
                if (t + 1 < TM * TN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (X3) { if (j2 == 0) asc[r] = fa.a_scale[min(bm + row_off(i2, r), g.M - 1)]; }
                        rv[r] = psalm_buf_load_f32(rr, (unsigned)((long)row_off(i2, r) * g.ldr * 4) + coff[j2]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) psalm_buf_store_f32(x[r], rc, (unsigned)((long)row_off(i, r) * ldo * 4) + coff[j]);
            }
            PSALM_TL(4);
            PSALM_TL_DRAIN();
            PSALM_TL(5);
            return;
        }
    }
    __syncthreads();                                             // all waves done with the operand ring before it is reused
    // ---- split-f16 OUTPUT tiles: all arithmetic (scales, bias, activation, the hi / lo split) in the ACCUMULATOR layout -- 16 x TM x TN
    // independent elements per lane, every wave busy, no loads in the way -- each element leaves as ONE packed word [hi | lo] for the LDS
    // transpose, and the row-major pass only unzips 8 words into the two 16-byte operand vectors and stores them.  Time line (r03c-e,
    // store phase of a 256 x 256 gelu_new / 128 x 128 erf-gelu / 64 x 128 relu tile): arithmetic inside the row-major store loop, one
    // wave row idle per pass: 25 / 20 / 8 us; this form: 26 / 13 / 4.5 us; stores straight from the accumulators with adjacent lanes
    // trading elements by DPP (4-byte stores, 64-byte row segments): 32 / 15 / 5.4 us -- measured and dropped: the 2-byte elements make
    // every store instruction touch twice the cache lines of the fp32 tile's.  Tiles larger than the LDS go in two passes of m-tiles
    // {2e, 2e+1} of BOTH wave rows (LDS row band b = wave row).
    if constexpr (SO) {
        if (fa.so != nullptr && bn + BN > fa.so_col_start) {
            constexpr int EPS = (BM * BN * 4 > SMEM_BYTES) ? 2 : 1;   // passes
            static_assert(TM % EPS == 0 && BM / EPS * BN * 4 <= SMEM_BYTES, "split-output epilogue: pass geometry");
            constexpr int TMP = TM / EPS, ROWS_P = BM / EPS, BAND = ROWS_P / WM;   // m-tiles per wave / rows / rows per wave row, per pass
            constexpr int TPR = BN / 8, RPI = NT / TPR, NIT = ROWS_P / RPI;
            unsigned* Cw = reinterpret_cast<unsigned*>(&smem[0][0]);
            const int act = g.act & 15;
            float gmax = 0.f;
            if (fa.so_global) {
                for (int r0 = 0; r0 < g.M; r0 += 512) {              // max over ALL rows of a_scale: 8 independent loads per lane in flight
                    float t8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t8[k] = fa.a_scale[min(r0 + 64 * k + lane, g.M - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) gmax = fmaxf(gmax, t8[k]);
                }
                gmax = wave_max(gmax);
            }
            const float p0 = fa.so_par[0], p1 = fa.so_par[1];
            const float floor_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaf(gmax, fa.so_par[2], fa.so_par[3]))));
            float wsc[TN], bias_c[TN];
            bool actc[TN], soc[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = bn + wn * (BN / WN) + j * 32 + n32, cc = min(col, g.N - 1);
                wsc[j] = fa.w_scale[cc];
                bias_c[j] = g.bias ? g.bias[cc] : 0.f;
                actc[j] = act != ACT_NONE && col >= g.act_col_start;
                soc[j] = col >= fa.so_col_start;
            }
            if constexpr (PAIR) {                                     // (host: so_col_start % BN == 0 -- no tile straddles it)
                static_assert(TN == 2, "paired split-f16 output: two 32-column tiles per wave");
                const psalm_rsrc srs = psalm_make_rsrc(fa.so, (unsigned)((long)g.M * fa.ldso * 2));
                const int lc = bn + wn * (BN / WN) + 2 * n32;                 // logical column of this lane's j = 0 element; j = 1: lc + 1
                const unsigned cbyte = lc + 1 < g.N ? (unsigned)(fa.so_col_off + (lc - fa.so_col_start)) * 2u : PSALM_BUF_OOB;
                const unsigned second = (unsigned)fa.so_kp * 2u;
                const bool writes_inv = bn == fa.so_col_start && wn == 0 && n32 == 0;
                // The element code is instantiated for the launch's activation as a compile-time constant and selected ONCE,
                // outside the loops: with run-time codes every element carried the activation switch, the per-lane act_col_start branch
                // and the form branch (r03k: 33 us of epilogue on the Phi fc1 tiles whichever way the words were stored -- 4 scalar
                // branches and an exec-mask region per element, no overlap between the 128 dependent chains of a lane).
                auto body = [&](auto AC) {
                    constexpr int A = decltype(AC)::value;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row0 = bm + wm * (BM / WM) + i * 32 + 4 * hi;
                        float asc[16], sc[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + (r & 3) + 8 * (r >> 2);
                            asc[r] = fa.a_scale[min(row, g.M - 1)];
                            float inv_;
                            split_scale_from_bound(fmaxf(fmaf(asc[r], p0, p1), floor_), sc[r], inv_);
                            if (writes_inv && row < g.M) fa.so_inv[row] = inv_;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + (r & 3) + 8 * (r >> 2);
                            float x0 = fmaf(acc[i][0][r], asc[r] * wsc[0], bias_c[0]);
                            float x1 = fmaf(acc[i][1][r], asc[r] * wsc[1], bias_c[1]);
                            const float y0 = apply_act_t<A>(x0, act), y1 = apply_act_t<A>(x1, act);
                            x0 = actc[0] ? y0 : x0;
                            x1 = actc[1] ? y1 : x1;
                            unsigned h0, s0, h1, s1;
                            psalm_split_words(x0 * sc[r], h0, s0);
                            psalm_split_words(x1 * sc[r], h1, s1);
                            const unsigned off = row < g.M && cbyte != PSALM_BUF_OOB ? (unsigned)row * (unsigned)(fa.ldso * 2) + cbyte : PSALM_BUF_OOB;
                            psalm_buf_store_u32(h0 | (h1 << 16), srs, off);
                            psalm_buf_store_u32(s0 | (s1 << 16), srs, off == PSALM_BUF_OOB ? off : off + second);
                        }
                    }
                };
                if (act == ACT_GELU) body(psalm_ic<ACT_GELU>{});
                else if (act == ACT_RELU) body(psalm_ic<ACT_RELU>{});
                else if (act == ACT_GELU_NEW) body(psalm_ic<ACT_GELU_NEW>{});
                else body(psalm_ic<-1>{});
                PSALM_TL(4);
                PSALM_TL_DRAIN();
                PSALM_TL(5);
                return;
            }
            if constexpr (!PAIR) {
            const int c8 = (tid % TPR) * 8, col0 = bn + c8;
            float* C = (float*)g.C;
            auto pass = [&](auto EC) {                                // (the pass index is a compile-time constant: acc[e * TMP + ii] is a register choice)
                constexpr int e = decltype(EC)::value;
                if (e > 0) __syncthreads();                           // previous pass read out
                // (element code instantiated for a compile-time activation and selected once per pass: see the paired form above)
                auto body = [&](auto AC) {
                    constexpr int A = decltype(AC)::value;
#pragma unroll
                    for (int ii = 0; ii < TMP; ++ii) {
                        const int i = e * TMP + ii;
                        const int lrow0 = wm * (BM / WM) + i * 32 + 4 * hi;          // tile-local row of accumulator element r = 0
                        const int prow0 = wm * BAND + ii * 32 + 4 * hi;              // its row in this pass's LDS image
                        float asc[16], sc[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            asc[r] = fa.a_scale[min(bm + lrow0 + (r & 3) + 8 * (r >> 2), g.M - 1)];
                            float inv_;
                            split_scale_from_bound(fmaxf(fmaf(asc[r], p0, p1), floor_), sc[r], inv_);
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float x = fmaf(acc[i][j][r], asc[r] * wsc[j], bias_c[j]);
                                const float y = apply_act_t<A>(x, act);
                                x = actc[j] ? y : x;
                                unsigned hw_, sw_;
                                psalm_split_words(x * sc[r], hw_, sw_);
                                const unsigned word = soc[j] ? (hw_ | (sw_ << 16)) : __builtin_bit_cast(unsigned, x);
                                Cw[(prow0 + (r & 3) + 8 * (r >> 2)) * BN + wn * (BN / WN) + j * 32 + n32] = word;
                            }
                    }
                };
                if (act == ACT_GELU) body(psalm_ic<ACT_GELU>{});
                else if (act == ACT_RELU) body(psalm_ic<ACT_RELU>{});
                else if (act == ACT_GELU_NEW) body(psalm_ic<ACT_GELU_NEW>{});
                else body(psalm_ic<-1>{});
                __syncthreads();
                if (col0 >= g.N) return;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int rl = it * RPI + tid / TPR;                          // row of the LDS image -> row of the tile
                    const int row = bm + (rl / BAND) * (BM / WM) + e * BAND + rl % BAND;
                    const u32x4_s w0 = *reinterpret_cast<const u32x4_s*>(&Cw[rl * BN + c8]);
                    const u32x4_s w1 = *reinterpret_cast<const u32x4_s*>(&Cw[rl * BN + c8 + 4]);
                    if (row >= g.M) continue;
                    if (col0 >= fa.so_col_start) {
                        const u32x4_s hv{(w0.x & 0xffffu) | (w0.y << 16), (w0.z & 0xffffu) | (w0.w << 16), (w1.x & 0xffffu) | (w1.y << 16),
                                         (w1.z & 0xffffu) | (w1.w << 16)};
                        const u32x4_s lv{(w0.x >> 16) | (w0.y & 0xffff0000u), (w0.z >> 16) | (w0.w & 0xffff0000u),
                                         (w1.x >> 16) | (w1.y & 0xffff0000u), (w1.z >> 16) | (w1.w & 0xffff0000u)};
                        unsigned short* d = fa.so + (long)row * fa.ldso + fa.so_col_off + (col0 - fa.so_col_start);
                        *reinterpret_cast<u32x4_s*>(d) = hv;
                        *reinterpret_cast<u32x4_s*>(d + fa.so_kp) = lv;
                        if (col0 == fa.so_col_start) {
                            float sc_, inv_;
                            split_scale_from_bound(fmaxf(fmaf(fa.a_scale[row], p0, p1), floor_), sc_, inv_);
                            fa.so_inv[row] = inv_;
                        }
                    } else {                                                       // fp32 columns of a tile that straddles so_col_start
                        float* dst = C + (long)row * g.ldc + col0;
                        const float v[8] = {__builtin_bit_cast(float, w0.x), __builtin_bit_cast(float, w0.y), __builtin_bit_cast(float, w0.z),
                                            __builtin_bit_cast(float, w0.w), __builtin_bit_cast(float, w1.x), __builtin_bit_cast(float, w1.y),
                                            __builtin_bit_cast(float, w1.z), __builtin_bit_cast(float, w1.w)};
                        if (fa.vec_store) store8(dst, v);
                        else {
#pragma unroll
                            for (int c = 0; c < 8; ++c) if (col0 + c < g.N) dst[c] = v[c];
                        }
                    }
                }
            };
            pass(psalm_ic<0>{});
            if constexpr (EPS > 1) pass(psalm_ic<1>{});
            static_assert(EPS <= 2, "split-output epilogue: at most two passes");
            PSALM_TL(4);
            PSALM_TL_DRAIN();
            PSALM_TL(5);
            return;
            }
        }
    }
    // ---- epilogue through LDS (bf16 outputs, erf / tanh activations on fp32 outputs): the accumulator layout (lane = one column, 16 scattered rows)
    // would store 2 bytes per lane; the tile is instead transposed through the (now idle) operand buffers and written as whole rows,
    // 8 consecutive columns (16 B bf16 / 32 B fp32) per lane, BN/8 lanes per row.  Tiles larger than the buffers go
    // in EP passes of BM/EP rows (one wave-row each).
    float* Cs = reinterpret_cast<float*>(&smem[0][0]);
    constexpr int ROWS_E = BM / EP;                               // rows per pass
    constexpr int TPR = BN / 8;                                   // threads per row
    constexpr int RPI = NT / TPR;                                 // rows per iteration
    const int c8 = (tid % TPR) * 8;
    const int col0 = bn + c8;
    const bool split = fa.slab != nullptr;
    const int act = g.act & 15;
    const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
    float bias8[8];
    const bool brow = (g.act & ACT_BIAS_ROW) != 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) bias8[c] = (!split && !brow && g.bias && col0 + c < g.N) ? g.bias[col0 + c] : 0.f;
    TC* C = (TC*)g.C;
    const TC* R = (const TC*)g.res;
    float* P = split ? fa.slab + (long)ksl * g.M * g.N : nullptr;
    // split-f16 output: the row-independent term of the magnitude bound and the two row-term parameters, in registers before the store loop
    // (inside it every read of so_par was a fresh global load behind the loop's own stores -- the compiler cannot prove they do not alias --
    // and the maximum over all rows of a_scale a chain of M / 64 dependent loads per wave: r03c time line, 23 us of store phase on the
    // Phi [k|v|q|fc1] tiles against 10 us for the plain fp32 tiles of the same launch)
    float so_floor = 0.f, so_p0 = 0.f, so_p1 = 0.f;
    if constexpr (SO) {
        if (fa.so != nullptr && bn + BN > fa.so_col_start) {
            float gmax = 0.f;
            if (fa.so_global) {
                for (int r0 = 0; r0 < g.M; r0 += 512) {                  // 8 independent loads per lane in flight
                    float t8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t8[k] = fa.a_scale[min(r0 + 64 * k + lane, g.M - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) gmax = fmaxf(gmax, t8[k]);
                }
                gmax = wave_max(gmax);
            }
            so_p0 = fa.so_par[0];
            so_p1 = fa.so_par[1];
            so_floor = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaf(gmax, fa.so_par[2], fa.so_par[3]))));
        }
    }
#pragma unroll 1
    for (int ep = 0; ep < EP; ++ep) {
        if (ep > 0) __syncthreads();                              // previous pass fully read out
        if (EP == 1 || wm == ep) {
            const int rbase = EP == 1 ? wm * (BM / WM) : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        Cs[row * BN + wn * (BN / WN) + j * 32 + n32] = acc[i][j][r];   // (split-f16: scaled in the store loop below)
                    }
        }
        constexpr int NIT = ROWS_E / RPI;
        // split-f16: per-row scales -- a_scale of this thread's NIT rows and w_scale of its 8 columns, fetched once here
        // (in the accumulator -> LDS pass they were 2 loads per accumulator element: 256 per lane, and spilled)
        float asc[X3 ? NIT : 1], wsc[8];
        if constexpr (X3) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) asc[it] = fa.a_scale[min(bm + ep * ROWS_E + it * RPI + tid / TPR, g.M - 1)];
#pragma unroll
            for (int c = 0; c < 8; ++c) wsc[c] = fa.w_scale[min(col0 + c, g.N - 1)];
        }
        bool so_here = false;                                     // this thread's 8 columns go out in split-f16 form
        if constexpr (SO) so_here = fa.so != nullptr && col0 >= fa.so_col_start;
        __syncthreads();
        if (ep == 0) PSALM_TL(4);
        if (col0 >= g.N) continue;
        // one store iteration (rows it * RPI + tid / TPR of the pass); false = past the last row of the matrix.  `itc` indexes the
        // row-scale registers and must be a compile-time constant there (fully unrolled caller); bf16 / fp32 kernels keep the compact
        // 2x-unrolled loop (a full unroll, tried with a residual prefetch, doubles the code of every instantiation -- 2.4 -> 4.7 MB of
        // ISA; unmeasured, and the instruction-cache footprint was judged the larger risk).
        auto store_it = [&](int it, int itc) -> bool {
            const int rl = it * RPI + tid / TPR;
            const int row = bm + ep * ROWS_E + rl;
            if (row >= g.M) return false;
            const f32x4_g v0 = *reinterpret_cast<const f32x4_g*>(&Cs[rl * BN + c8]);
            const f32x4_g v1 = *reinterpret_cast<const f32x4_g*>(&Cs[rl * BN + c8 + 4]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (X3) {                            // dequantise: per-row scale of A x per-row scale of W
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] *= asc[itc] * wsc[c];
            }
            if (split) {
                float* dst = P + (long)row * g.N + col0;
                if (fa.vec_store) {
                    reinterpret_cast<f32x4_g*>(dst)[0] = f32x4_g{v[0], v[1], v[2], v[3]};
                    reinterpret_cast<f32x4_g*>(dst)[1] = f32x4_g{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) if (col0 + c < g.N) dst[c] = v[c];
                }
                return true;
            }
            float rres[8];
            if (R) {
                if (fa.vec_store) load8_f32(R + (long)row * g.ldr + col0, rres);
                else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) rres[c] = col0 + c < g.N ? ldf(R + (long)row * g.ldr + col0 + c) : 0.f;
                }
            }
            const float rb = (brow && g.bias) ? g.bias[row] : 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float x = v[c] + bias8[c] + rb;
                const bool do_act = act != ACT_NONE && col0 + c >= g.act_col_start;
                if (do_act && !post) x = apply_act(x, act);
                if (R) x += rres[c];
                if (do_act && post) x = apply_act(x, act);
                v[c] = x;
            }
            if constexpr (SO) {
                if (so_here) {
                    float sc, inv;
                    split_scale_from_bound(fmaxf(fmaf(asc[itc], so_p0, so_p1), so_floor), sc, inv);
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a0 = v[2 * k] * sc, a1 = v[2 * k + 1] * sc;
                        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                        const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                        hw[k] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                        lw[k] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                    }
                    unsigned short* d = fa.so + (long)row * fa.ldso + fa.so_col_off + (col0 - fa.so_col_start);
                    *reinterpret_cast<u32x4_s*>(d) = u32x4_s{hw[0], hw[1], hw[2], hw[3]};
                    *reinterpret_cast<u32x4_s*>(d + fa.so_kp) = u32x4_s{lw[0], lw[1], lw[2], lw[3]};
                    if (col0 == fa.so_col_start) fa.so_inv[row] = inv;
                    return true;
                }
            }
            TC* dst = C + (long)row * g.ldc + col0;
            if (fa.vec_store) store8(dst, v);
            else {
#pragma unroll
                for (int c = 0; c < 8; ++c) if (col0 + c < g.N) stf(dst + c, v[c]);
            }
            return true;
        };
        if constexpr (X3) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (!store_it(it, it)) break;
        } else {
#pragma unroll 2
            for (int it = 0; it < NIT; ++it)
                if (!store_it(it, 0)) break;
        }
    }
    PSALM_TL_DRAIN();                                            // (timeline build: the stores have left the wave)
    PSALM_TL(5);
}

// ------------------------------------------------------------------------------------------- skinny GEMM (M <= 128)
// The Mask2Former predictor issues ~100 dependent GEMMs with M = 100 queries (N, K in {256, 512, 2048}): a few MFLOP each,
// pure latency.  The tiled kernel above costs ~10 us per launch there (LDS staging + one barrier per K step + the LDS epilogue
// for 4 blocks of work); this one has NO staging and NO K-loop barrier: a block owns one 32x32 output tile, its 4 wavefronts
// split K four ways, every lane pulls its MFMA fragments straight from global / L2 with 16-byte loads (all loads of a chunk in
// flight together), the 4 partial tiles meet once in LDS, and the epilogue is applied from registers.
// X3 = true: split-f16 operands ([hi | lo] f16 rows of 2 kp columns, per-row scales; see GemmFastArgs::x3_kp): g.K = 3 kp.
struct SkinnyX3 { const float* a_scale; const float* w_scale; int kp; };
template <typename TC, bool X3 = false>
__global__ void __launch_bounds__(256) gemm_bf16_skinny_kernel(GemmArgs g, SkinnyX3 x3) {
    __shared__ float part[3][32 * 32];                           // partial tiles of waves 1..3
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int bm = blockIdx.y * 32, bn = blockIdx.x * 32;
    const bf16_t* A = (const bf16_t*)g.A + (long)min(bm + n32, g.M - 1) * g.lda + 8 * hi;
    const bf16_t* W = (const bf16_t*)g.W + (long)min(bn + n32, g.N - 1) * g.ldw + 8 * hi;
    const int ksteps = g.K / 16;                                 // K % 64 == 0 -> divisible by 4
    const int per = ksteps / 4, k0 = wave * per;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // Chunks of CH k-steps with NO per-load guard: a guarded load (`if (kb + i < per) load`) compiles to a branch and an
    // s_waitcnt vmcnt(0) per load -- r01 ISA audit: 48 of this kernel's 49 loads were serialized that way, i.e. the "all loads of a
    // chunk in flight together" design was not what ran.  Full chunks of 8, then of 4 (K = 256: exactly one), then single steps.
    int kb = 0;
    auto chunk = [&](auto CHT) {
        constexpr int CH = decltype(CHT)::value;
        u32x4_s fa_[CH], fb_[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            long ka = (long)(k0 + kb + i) * 16, kw = ka;
            if constexpr (X3) {                                  // 16-deep k-step -> operand columns (kp % 64 == 0: never straddles)
                if (ka >= 2 * x3.kp) ka -= 2 * x3.kp;
                if (kw >= x3.kp) kw -= x3.kp;
            }
            fa_[i] = *reinterpret_cast<const u32x4_s*>(A + ka);
            fb_[i] = *reinterpret_cast<const u32x4_s*>(W + kw);
        }
        __builtin_amdgcn_sched_barrier(0);                       // all 2 CH loads issued before the first MFMA waits on one
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if constexpr (X3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa_[i]), __builtin_bit_cast(f16x8, fb_[i]), acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa_[i]), __builtin_bit_cast(bf16x8, fb_[i]), acc, 0, 0, 0);
        }
        kb += CH;
    };
    while (kb + 8 <= per) chunk(std::integral_constant<int, 8>{});
    if (kb + 4 <= per) chunk(std::integral_constant<int, 4>{});
    while (kb < per) chunk(std::integral_constant<int, 1>{});
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += part[0][r * 64 + lane] + part[1][r * 64 + lane] + part[2][r * 64 + lane];
        if constexpr (X3) {                                      // per-row scales of A (this lane's 16 rows) and W (its column)
            const float ws = x3.w_scale[min(bn + n32, g.N - 1)];
            float as_[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) as_[r] = x3.a_scale[min(bm + (r & 3) + 8 * (r >> 2) + 4 * hi, g.M - 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= as_[r] * ws;
        }
        epilogue_store<TC>(g, acc, bm, bn + n32, lane);
    }
}

// C = epilogue(sum_z slab[z]); one thread per 4 consecutive columns.
template <typename TC>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g, const float* __restrict__ slab, int splits) {
    const long n4 = (g.N + 3) / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)g.M * n4) return;
    const int row = (int)(idx / n4), c0 = (int)(idx % n4) * 4;
    const int act = g.act & 15;
    const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
    TC* C = (TC*)g.C;
    const TC* R = (const TC*)g.res;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = (c0 + 3 < g.N) && (g.N % 4 == 0);
    if (full) {                                                  // 4 slabs per round trip (unconditional loads, clamped z, masked add)
        for (int z0 = 0; z0 < splits; z0 += 4) {
            f32x4_g t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[k] = *reinterpret_cast<const f32x4_g*>(slab + ((long)min(z0 + k, splits - 1) * g.M + row) * g.N + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (z0 + k < splits) { v[0] += t[k].x; v[1] += t[k].y; v[2] += t[k].z; v[3] += t[k].w; }
        }
    } else {
        for (int z = 0; z < splits; ++z) {
            const float* p = slab + ((long)z * g.M + row) * g.N + c0;
            for (int i = 0; i < 4; ++i) if (c0 + i < g.N) v[i] += p[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = c0 + i;
        if (col >= g.N) break;
        float x = v[i] + (g.bias ? g.bias[(g.act & ACT_BIAS_ROW) ? row : col] : 0.f);
        const bool do_act = act != ACT_NONE && col >= g.act_col_start;
        if (do_act && !post) x = apply_act(x, act);
        if (R) x += ldf(R + (long)row * g.ldr + col);
        if (do_act && post) x = apply_act(x, act);
        stf(C + (long)row * g.ldc + col, x);
    }
}

// Split-K reduce fused with the LayerNorm that follows the GEMM in the network (Phi: x = x + [attn|mlp].W2^T, then the next
// layer's input_layernorm): one block per output row sums the slabs, applies bias / activation / residual, writes the fp32
// row AND its LayerNorm (the next GEMM's A operand) -- one pass over the row instead of reduce + a separate LN launch.
// N % 4 == 0, N <= 8192.
// NV = 4-column vectors per thread (N <= 1024 * NV).  Every global load is unconditional (column / slab index clamped, the value
// masked afterwards) and the slab reads go in groups of 4 slabs x NV vectors: guarded loads inside the runtime z loop had compiled
// to one s_waitcnt vmcnt(0) round trip per load (r01 ISA audit: 72 of 72), i.e. ~(splits + 3) * NV dependent L2 / HBM latencies
// per block.  The per-element summation order (z ascending) is unchanged.
// SPLIT = true: the normalised row also (ln_out may be NULL: only) leaves in split-f16 operand form -- [hi | lo] f16 rows of 2 * sp_kp
// columns with the exact row-maximum scale of psalm_split_f16 (N % 64 == 0: no K padding to zero), 1/scale in sp_inv[row].
template <typename TL, int NV, bool SPLIT = false>
__global__ void __launch_bounds__(256) splitk_reduce_ln_kernel(GemmArgs g, const float* __restrict__ slab, int splits,
                                                               const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta,
                                                               float ln_eps, TL* __restrict__ ln_out, long ld_ln,
                                                               unsigned short* __restrict__ sp_out = nullptr, float* __restrict__ sp_inv = nullptr,
                                                               int sp_kp = 0) {
    __shared__ float red[12];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int act = g.act & 15;
    const bool post = (g.act & ACT_POST_RESIDUAL) != 0;
    float* C = (float*)g.C;
    const float* R = (const float*)g.res;
    int c0[NV], cc[NV];                                          // this thread's columns / clamped copy for the loads
    f32x4_g a[NV], bi[NV], ga[NV], be[NV];
    float rr[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        c0[i] = (tid + 256 * i) * 4;
        cc[i] = min(c0[i], g.N - 4);
        a[i] = f32x4_g{0.f, 0.f, 0.f, 0.f};
        ga[i] = *reinterpret_cast<const f32x4_g*>(ln_gamma + cc[i]);
        be[i] = *reinterpret_cast<const f32x4_g*>(ln_beta + cc[i]);
        bi[i] = g.bias ? f32x4_g{g.bias[cc[i]], g.bias[cc[i] + 1], g.bias[cc[i] + 2], g.bias[cc[i] + 3]} : f32x4_g{0.f, 0.f, 0.f, 0.f};
    }
    if (R) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[i][k] = R[(long)row * g.ldr + cc[i] + k];
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[i][k] = 0.f;
    }
    for (int z0 = 0; z0 < splits; z0 += 4) {
        f32x4_g t[NV][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* sp = slab + ((long)min(z0 + k, splits - 1) * g.M + row) * g.N;
#pragma unroll
            for (int i = 0; i < NV; ++i) t[i][k] = *reinterpret_cast<const f32x4_g*>(sp + cc[i]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (z0 + k < splits) {
#pragma unroll
                for (int i = 0; i < NV; ++i) { a[i].x += t[i][k].x; a[i].y += t[i][k].y; a[i].z += t[i][k].z; a[i].w += t[i][k].w; }
            }
    }
    f32x4_g v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
        const float bb[4] = {bi[i].x, bi[i].y, bi[i].z, bi[i].w};
        const bool valid = c0[i] < g.N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = c0[i] + k;
            float y = x[k] + bb[k];
            const bool do_act = act != ACT_NONE && col >= g.act_col_start;
            if (do_act && !post) y = apply_act(y, act);
            y += rr[i][k];
            if (do_act && post) y = apply_act(y, act);
            x[k] = y;
            sum += valid ? y : 0.f;
        }
        v[i] = f32x4_g{x[0], x[1], x[2], x[3]};
        if (valid) *reinterpret_cast<f32x4_g*>(C + (long)row * g.ldc + c0[i]) = v[i];
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / g.N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (c0[i] < g.N) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / g.N + ln_eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = f32x4_g{(v[i].x - mean) * rstd * ga[i].x + be[i].x, (v[i].y - mean) * rstd * ga[i].y + be[i].y,
                       (v[i].z - mean) * rstd * ga[i].z + be[i].z, (v[i].w - mean) * rstd * ga[i].w + be[i].w};
        if (c0[i] < g.N) {
            if (ln_out) {
                TL* o = ln_out + (long)row * ld_ln + c0[i];
                stf(o + 0, v[i].x); stf(o + 1, v[i].y); stf(o + 2, v[i].z); stf(o + 3, v[i].w);
            }
            if constexpr (SPLIT) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        }
    }
    if constexpr (SPLIT) {
        amax = wave_max(amax);
        if (lane == 0) red[8 + wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
        // psalm_split_f16's row scale: the row maximum into [2^13, 2^14)
        int se = 13 - ((int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127);
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        const bool zero = !(amax > 0.f) || !(amax < 3.0e38f);
        const float sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
        if (tid == 0) sp_inv[row] = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
        unsigned short* orow = sp_out + (long)row * 2 * sp_kp;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (c0[i] < g.N) {
                const float x4[4] = {v[i].x * sc, v[i].y * sc, v[i].z * sc, v[i].w * sc};
                unsigned hw[2], lw[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    unsigned h0, h1, l0, l1;
                    psalm_split_words(x4[2 * k], h0, l0);
                    psalm_split_words(x4[2 * k + 1], h1, l1);
                    hw[k] = h0 | (h1 << 16);
                    lw[k] = l0 | (l1 << 16);
                }
                *reinterpret_cast<u32x2_s*>(orow + c0[i]) = u32x2_s{hw[0], hw[1]};
                *reinterpret_cast<u32x2_s*>(orow + sp_kp + c0[i]) = u32x2_s{lw[0], lw[1]};
            }
    }
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

// ---- exact-fp32 skinny GEMM (M <= 192): the latency-bound M = 100 GEMMs of the mask decoder in the fp32 / f16x3 modes.  Same shape as
// the bf16 skinny kernel (block = one 32 x 32 output tile, 4 wavefronts split K, fragments straight from global / L2, no K-loop barrier),
// on v_mfma_f32_32x32x2_f32.  Lane (n, hi) contracts k = 8 c + 4 hi + i in step i of chunk c, so A and W fragments are 16-byte loads.
// In the f16x3 mode this replaces split + split-f16 skinny GEMM (two launches, ~15 us) for these ~100 tiny GEMMs per image.
template <typename TC, int NWV>
__device__ __forceinline__ void gemm_f32_skinny_body(const GemmArgs& g, float (*part)[32 * 32], int bx, int by) {
    // NWV wavefronts split K (16 for these latency-bound problems: the fp32 MFMA takes 64 cycles, a wave's serial chain of K / 2 / NWV of them
    // IS the kernel's duration -- r02k: 9 us per launch with 4 waves); the partial tiles meet in LDS and every thread finishes one element.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int bm = by * 32, bn = bx * 32;
    const float* A = (const float*)g.A + (long)min(bm + n32, g.M - 1) * g.lda + 4 * hi;
    const float* W = (const float*)g.W + (long)min(bn + n32, g.N - 1) * g.ldw + 4 * hi;
    const int chunks = g.K / 8, cpw = (chunks + NWV - 1) / NWV;
    const int c_lo = min(chunks, wave * cpw), c_hi = min(chunks, c_lo + cpw);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // r05: the bias / residual values of the element(s) this thread finishes are fetched before the K loop, not behind the LDS reduction
    constexpr int EPT = 1024 / (64 * NWV);
    float pbias[EPT], pres[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * 64 * NWV, r = e >> 6, ln = e & 63;
        const int row = min(bm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), g.M - 1), col = min(bn + (ln & 31), g.N - 1);
        pbias[k] = g.bias ? g.bias[(g.act & ACT_BIAS_ROW) ? row : col] : 0.f;
        pres[k] = g.res ? ldf((const TC*)g.res + (long)row * g.ldr + col) : 0.f;
    }
    int c = c_lo;
    auto group = [&](auto NT_) {
        constexpr int NT = decltype(NT_)::value;
        f32x4_g fa_[NT], fb_[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            fa_[i] = *reinterpret_cast<const f32x4_g*>(A + (long)(c + i) * 8);
            fb_[i] = *reinterpret_cast<const f32x4_g*>(W + (long)(c + i) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);                       // all loads of the group in flight before the first MFMA waits
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].x, fb_[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].y, fb_[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].z, fb_[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].w, fb_[i].w, acc, 0, 0, 0);
        }
        c += NT;
    };
    while (c + 8 <= c_hi) group(std::integral_constant<int, 8>{});
    if (c + 4 <= c_hi) group(std::integral_constant<int, 4>{});
    if (c + 2 <= c_hi) group(std::integral_constant<int, 2>{});        // K = 256 on 16 waves: 2 chunks per wave -- one round trip, not two
    if (c < c_hi) group(std::integral_constant<int, 1>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    // element e = r * 64 + lane of the accumulator layout: row (r&3) + 8(r>>2) + 4(lane>>5), column lane & 31
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * 64 * NWV, r = e >> 6, ln = e & 63;
        const int row = bm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = bn + (ln & 31);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) v += part[w][e];
        if (row < g.M && col < g.N) {
            const int act = g.act & 15;
            const bool post = (g.act & ACT_POST_RESIDUAL) != 0, do_act = act != ACT_NONE && col >= g.act_col_start;
            if (g.bias) v += pbias[k];
            if (do_act && !post) v = apply_act(v, act);
            if (g.res) v += pres[k];
            if (do_act && post) v = apply_act(v, act);
            stf((TC*)g.C + (long)row * g.ldc + col, v);
        }
    }
}
template <typename TC, int NWV>
__global__ void __launch_bounds__(64 * NWV) gemm_f32_skinny_kernel(GemmArgs g) {
    __shared__ float part[NWV][32 * 32];
    gemm_f32_skinny_body<TC, NWV>(g, part, blockIdx.x, blockIdx.y);
}
// Two independent skinny problems in ONE launch (blockIdx.z picks the problem; the grid covers the larger tile grid, the other's surplus blocks
// leave at once): the mask decoder's self-attention projections  [q | k] = (x + pos) . Wqk^T  and  v = x . Wv^T  -- different A operands, so not
// one GEMM -- were two dependent-looking ~6 us launches of its serial tail (mask2former_transformer_decoder.py:24-38).  Same body, same words.
template <typename TC, int NWV>
__global__ void __launch_bounds__(64 * NWV) gemm_f32_skinny_pair_kernel(GemmArgs g0, GemmArgs g1) {
    __shared__ float part[NWV][32 * 32];
    const GemmArgs& g = blockIdx.z ? g1 : g0;
    if ((int)blockIdx.x * 32 >= g.N || (int)blockIdx.y * 32 >= g.M) return;      // (block-uniform)
    gemm_f32_skinny_body<TC, NWV>(g, part, blockIdx.x, blockIdx.y);
}

// (r06: the process-wide tuning words below are atomics -- a test or `--gemm-policy` flipping one while another host thread (PSALM.replica) launches
//  gives that thread's next launch the old or the new value, never a torn one)
// Tuning / test knob: 0 = automatic tile selection (default), 256 / 128 / 64 = force that BM for the direct-to-LDS path
// (A/B measurements in tools/bench_gemm.py, and the CPU tests reach the 256^2 configuration at small sizes with it).
static std::atomic<int> g_tile_policy{0};
static thread_local char g_last_kernel[192] = "";    // template instantiation of the calling thread's last direct-to-LDS launch
extern "C" const char* psalm_gemm_last_kernel() { return g_last_kernel; }
static std::atomic<long> g_skinny_nmax{4096};    // skinny kernel for M <= 128 and N <= this
static std::atomic<int> g_ring_depth{2};      // operand-ring depth of the 128x128 configuration (2 or 3), see psalm_gemm_set_tile_policy
// operand-ring depth of the 64x128 configuration: 0 = automatic (3 when the K range of a block is >= 1024: with <= 1 block per CU the
// 2-deep loop is one exposed copy round trip per K step -- r01 A/B: Swin fc2 M4096 N512 K2048 25.4 -> 22.7 us; short-K problems lose
// to the longer prologue), 2 / 3 / 4 = forced (policy codes 642 / 643 / 644; 640 = automatic)
static std::atomic<int> g_ring64{0};
// 256x256 plain-GEMM tiles run the 4-phases-per-K-tile (PH8) K loop, variant 3 (r01 A/B on MI355X, tools/bench_gemm.py --ph8: Phi w1
// 74.9 -> 69.4 us, 4096^3 1007 -> 1091 TF/s, 8192^3 1053 -> 1191; bitwise equal to the 2-buffer loop on 360 / 360 repetitions);
// 0 = the plain 2-buffer loop, 1 / 2 = the other copy placements (psalm_gemm_set_tile_policy 2560 / 2568..2570)
static std::atomic<int> g_ph8{3};
// split-f16 GEMMs on the 128x128 / 64x128 tiles, K loop form (psalm_gemm_set_tile_policy 3300 + v):
//   v = 0  automatic (below)                    1  "slice" K loop: 4 operand images per 64-deep slice, 3 products per barrier
//   2  32-deep slices in a 4-deep ring          3 / 4  32-deep slices in a 2- / 3-deep ring       5  the K-panel form (3 Kp-long loop)
// r02 sweep (profiles/r02l_gemm_x3_slice_ab.json): forms 1 / 2 need 128 KB / 96 KB stages, leave ONE block per CU and lose to the K-panel
// form at 2-3 blocks per CU on every short-K problem (M4096 N2048 K512: 74 vs 50 us); 1 wins only where a long K meets a grid that
// 64 x 128 tiles can fill without split-K (M4096 N512 K2048, Swin stage-2 fc2: 39 vs 53 us) -- select_fast_config picks it there.
// r03 sweep (profiles/r03h_gemm_x3_sweep_slice32.json): form 3 keeps the K-panel form's LDS footprint (64 KB on 128^2: two blocks per CU)
// with 1.5x the matrix work per copy round trip and 2/3 of the copies: 3-17 % faster on 17 of 21 shapes of the image (M5184 N512 K512
// 22.0 -> 18.3 us, M21504 N256 K1024 60.3 -> 52.6, M65536 N128 K512 43.4 -> 38.1), equal within noise on the rest: the automatic choice.
static std::atomic<int> g_x3_slice{0};
// split-f16 GEMMs on the 256 x 256 tile: 1 = the phased K loop walks 32-deep SLICES (four operand images per stage, three products per
// phase: 2/3 of the L2 -> LDS bytes, fragment reads and barriers of the K-panel form, W hi fetched once), 0 = the K-panel form (3 Kp-long
// loop over 64-deep tiles).  psalm_gemm_set_tile_policy(2580 / 2581).  r04a on MI355X (profiles/r04a_gemm_x3_sweep.json, back to back): Phi
// [k|v|q|fc1] 158.9 -> 150.5 us, [dense|fc2] on 256^2 tiles 131.1 -> 124.4 us (128^2 tiles: 142), M65536 N256 K2304 231 -> 215; in the model
// [k|v|q|fc1] 172 -> 164 us by events (profiles/r04a_bench_phased_slice_ab.txt).  2 (policy 2582) = the same with the matrix instructions
// and fragment reads of all-padding m-tiles left out (<.., 32, 4, 2, ..>): the launch is clock-limited (1.76 GHz, r04j SQ counters), the
// left-out work is power.  r04p on MI355X (profiles/r04p_skip_pad.json, back to back, identical output words): M899 N14336 K2048 150.3 ->
// 141.1 us, M899 N2048 K10240 123.3 -> 117.4, M1024 N14336 K2048 (nothing to leave out) 150.3 -> 149.9: the default.
static std::atomic<int> g_ph8_slice{2};
// Products per algorithmic product of the split-f16 GEMMs launched by THIS host thread: 3 (default) = hi.hi + lo.hi + hi.lo -- the fp32-class
// arithmetic of precision "f16x3"; 1 = hi.hi only -- plain f16 operands (11-bit mantissas under the same per-row power-of-two scales), one
// third of the matrix work.  The reduced-precision LLM side mode of BASELINE.json configs[4] (PSALM(llm_products=1)): NOT at the parity
// bar, measured and labelled as such.  Implementation: the K-panel kernels walk logical k in [0, 3 Kp) = [hi.hi | lo.hi | hi.lo]; stopping
// at Kp is the hi.hi product -- no kernel of its own.
static thread_local int g_x3_products = 3;
extern "C" int psalm_gemm_x3_set_products(int n) {
    if (n != 1 && n != 3) { psalm_set_error("psalm_gemm_x3_set_products: 1 (hi.hi only) or 3"); return -1; }
    g_x3_products = n;
    return 0;
}
// r06 "mid" forms of the split-f16 slice GEMM -- blocks with loader wavefronts (LW, see the kernel comment) -- policy codes 4400 + v:
//   0  automatic (select_mid_form)      9  the r05 kernels whatever the selection says
//   6 / 7  256 x 128 / 128 x 256 blocks (eight 64 x 64 wave tiles) + two loaders      8  256 x 128 + four loaders
//   10 / 11  128 x 128 / 64 x 128 blocks (four waves) + two loaders      12 / 13  (experiment) forms 8 / 10 with the K range split to fill the chip
// (the forms 1 - 5 of the round's first sweep -- interleaved copy issue, eight-wave blocks whose waves all copy -- are gone from the product:
//  profiles/r06b_gemm_mid_sweep_ilv_and_8wave_tiles.json, tools/experiments/r06_ilv_copy_issue.patch)
static std::atomic<int> g_mid_form{0};
static thread_local bool g_x3_auto_slice = false;    // set by select_fast_config (per host thread: read back by the same thread's launch): this problem takes the slice form on 64 x 128 tiles
extern "C" int psalm_gemm_set_tile_policy(int bm) {
    if (bm == 640 || (bm >= 642 && bm <= 644)) { g_ring64 = bm - 640; return 0; }   // 64x128, BK 64, ring depth auto / 2 / 3 / 4
    if (bm == 1282 || bm == 1283) { g_ring_depth = bm - 1280; return 0; }      // 128x128, BK 64, ring depth 2 / 3 (tuning)
    if (bm == 1323 || bm == 1324) { g_ring_depth = bm - 1000; return 0; }      // 128x128, BK 32, ring depth 3 / 4 (tuning)
    if (bm == 7777 || bm == 7778) { g_skinny_nmax = bm == 7777 ? (1L << 20) : 4096; return 0; }   // skinny-kernel N limit (tuning)
    if (bm >= 2568 && bm <= 2570) { g_ph8 = bm - 2567; return 0; }             // 256x256 PH8 K loop variant 1 / 2 / 3
    if (bm == 2560) { g_ph8 = 0; return 0; }                                    // ... off
    if (bm >= 2580 && bm <= 2582) { g_ph8_slice = bm - 2580; return 0; }        // split-f16 on 256 x 256 tiles: K-panel form / 32-deep slices / slices with the all-padding m-tiles left out (2582: the DEFAULT since r04p)
    if (bm >= 4400 && bm <= 4419) { g_mid_form.store(bm - 4400); return 0; }     // r06 mid-size forms (see g_mid_form)
    if (bm >= 3300 && bm <= 3308) { g_x3_slice = bm - 3300; return 0; }         // split-f16 K loop form on the 128 / 64-row tiles (7 / 8: r05 ring depths)
    if (bm == 128128) { g_ring_depth = 128; return 0; }                        // 128x128, BK 128 (K % 128 == 0 problems only)
    if (bm != 0 && bm != 256 && bm != 128 && bm != 64 && bm != 12864) { psalm_set_error("psalm_gemm_set_tile_policy: 0, 256, 128 or 64"); return -1; }
    g_tile_policy = bm;
    return 0;
}

// Tile / split-K selection of the direct-to-LDS path (pure function of the problem size).
static void select_fast_config(int M, int N, int K, bool have_ws, long workspace_bytes, int& BM, int& BN, int& splits, bool x3 = false) {
    // 256^2 tiles pay off only when they alone fill the chip and the K loop is long enough to amortise the bigger
    // prologue / two-pass epilogue (measured r1e: 4096^3 1092 vs 937 TF/s, Phi [k|v|q|fc1] 713 vs 630; K <= 256 or
    // < 200 tiles: the 128^2 configuration (2 blocks/CU, optional split-K) wins)
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    const bool can_split = have_ws && K >= 1024;
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    bool no_split = false;
    if (M >= 512 && N >= 256 && K >= 512 && t256 >= 200) { BM = 256; BN = 256; }
    else if (M > 192 && t128 >= 100 && t128 < 200 && K <= 2048) { BM = 64; BN = 128; no_split = true; }   // r1l: Swin fc2 24 vs 35 us (split-K)
    else if (M > 192) { BM = 128; BN = 128; }
    else { BM = 64; BN = 128; }
    // split-f16 GEMMs (K = 3 Kp, fp32 output): short K loops and grids that cannot fill two 128^2 blocks per CU run better on 64 x 128
    // tiles (r02 sweep tools/bench_gemm_x3.py, profiles/r02f_gemm_x3_policies.json: M21504 N1024 K256 105 -> 67 us, M65536 N512 K128
    // 112 -> 77, M16384 N1024 K256 67 -> 53, M1024 N4096 K1024 57 -> 42)
    if (x3 && M > 192 && (K <= 1024 || (t128 < 448 && K <= 4096))) { BM = 64; BN = 128; no_split = true; }
    // split-f16, few 256^2 tiles but a long K (Phi [dense|fc2]: 4 x 8 tiles, Kp 10240): the slice-form phased kernel on 256^2 tiles with the K
    // range split 256 / tiles ways beats the 128^2 slice kernel (r04a: 142 -> 124 us); shorter K ranges per block lose to it (M1024 N1024 K4096,
    // M256 N2048 K18432 in the same sweep: 16 / 32 slices of 256 / 576)
    if (x3 && g_ph8 && g_ph8_slice && !g_tile_policy && M >= 512 && N >= 512 && t256 >= 24 && t256 < 160 && can_split &&
        (long)K / 3 >= 1024 * ((256 + t256 - 1) / t256)) { BM = 256; BN = 256; no_split = false; }
    g_x3_auto_slice = false;
    if (x3 && !g_tile_policy && M > 192 && t128 < 200 && K >= 4096 && K <= 8192 && (long)cdiv(M, 64) * cdiv(N, 128) >= 200) {
        BM = 64; BN = 128; no_split = true; g_x3_auto_slice = true;      // long K, small grid: 64 x 128 slice form, no split-K
    }
    if (g_tile_policy == 12864) { if (M > 192) { BM = 128; BN = 64; } }       // experiment: 48 KB LDS -> 3 blocks/CU
    else if (g_tile_policy) { BM = g_tile_policy; BN = BM == 256 ? 256 : 128; no_split = false; }
    const long tiles = (long)cdiv(M, BM) * cdiv(N, BN);
    const long fill = BM == 256 ? 256 : 448;                                // blocks that fill the chip (1 vs ~2 per CU)
    splits = 1;
    if (tiles < (BM == 256 ? 160 : 200) && can_split && !no_split) {
        splits = (int)((fill + tiles - 1) / tiles);
        if (splits > K / 512) splits = K / 512;                          // >= 8 K-steps per slice
        if (splits > 32) splits = 32;
        const long per = (long)M * N * (long)sizeof(float);
        if ((long)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
        if (splits < 2) splits = 1;
    }
    if (splits > 1) {
        const int kps = cdiv(cdiv(K, 64), splits) * 64;
        splits = cdiv(K, kps);
    }
}

// Automatic choice among the r06 mid forms (g_mid_form) for an un-split slice-form problem: 9 = keep the r05 kernel.  Rule read off the sweep
// profiles/r06d_gemm_mid_sweep_loader_waves.json (33 problem / output-form pairs x 10 forms, back to back on one MI355X): the loader-wave blocks
// win where the K loop is long enough to matter (Kp >= 512: 16 slices) AND their grid is ONE round of blocks that fills most of the chip -- they
// hold one block per CU (two for the 64 x 128 form), so 264 tiles cost two rounds (M1296 N3072 K1024 on 128 x 128: 37 -> 50 us) and 128 tiles
// leave half the chip idle.  Largest tile first (fewest L2 -> LDS bytes per product): 256 x 128 with four loaders, 128 x 128 and 64 x 128 with two.
// In the sweep the rule takes: M5184 N1536 K512 33.5 -> 31.5 us, M4096 N2048 K512 34.3 -> 31.9, M21504 N256 K1024 53.1 -> 43.2, M16384 N256 K1024
// 39.9 -> 33.6, M5184 N512 K512 17.8 -> 16.9, M1296 N1024 K1024 23.5 -> 18.0, M4096 N512 K1024 24.6 -> 19.5, M1024 N4096 K1024 (split-f16 output)
// 40.8 -> 38.3; every Kp <= 256 problem keeps its r05 kernel (LW: -3 % ... +25 % there).
// K slices for a loader-wave grid of `tiles` blocks: the largest of 8 / 4 / 2 that keeps the launch within one round of 256 blocks and leaves every
// slice >= 8 slices of 32 (powers of two: the slice then decides the XCD, GemmFastArgs::xcd_ksplit)
static int mid_split_count(long tiles, int Kp) {
    int s = 1;
    while (s < 8 && tiles * (s * 2) <= 256 && Kp / (s * 2) >= 256) s *= 2;
    return s;
}
static int select_mid_form(int M, int N, int Kp, int* msplit = nullptr) {
    if (msplit) *msplit = 1;
    if (Kp < 512) return 9;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 128), t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 128);
    if (t256 >= 160 && t256 <= 256) return 8;
    if (t128 >= 160 && t128 <= 256) return 10;
    if (t64 >= 160 && t64 <= 512) return 11;
    return 9;
}

// Which kernel psalm_gemm launches for a problem: out[0] = path (0 register-staged, 1 direct-to-LDS, 2 skinny), out[1] = BM,
// out[2] = BN, out[3] = split-K slices.  (bench.py uses it to attribute measured launch times to kernel instantiations.)
extern "C" int psalm_gemm_describe(int M, int N, int K, int a_dtype, int w_dtype, long workspace_bytes, int* out4) {
    const bool x3 = a_dtype == 2 && w_dtype == 2;                              // dtype code 2: split-f16 operands (psalm_gemm_x3, K = 3 Kp)
    if (x3 && M <= 128 && N <= g_skinny_nmax && !g_tile_policy) {
        out4[0] = 2; out4[1] = 32; out4[2] = 32; out4[3] = 1;
    } else if (x3) {
        int BM, BN, splits;
        select_fast_config(M, N, K, workspace_bytes > 0, workspace_bytes, BM, BN, splits, true);
        // the slice forms (every default K loop of the split-f16 GEMM) walk the TRUE K range Kp = K / 3: the split count is re-derived on it,
        // as launch_fast does (this function reported the K-panel form's count)
        const int kp = K / 3;
        bool slice = BM == 256 ? (g_ph8 && g_ph8_slice) : g_x3_slice != 5;
        if (BM == 256 && slice) {
            const int kps_ = splits > 1 ? cdiv(cdiv(kp, 64), splits) * 64 : kp, sp_ = cdiv(kp, kps_);
            slice = kps_ >= 64 && kp - (sp_ - 1) * kps_ >= 64;
        }
        if ((slice || g_x3_products == 1) && splits > 1) {        // slice forms AND the one-product mode walk Kp, not 3 Kp (launch_fast's `p1`)
            const int kps = cdiv(cdiv(kp, 64), splits) * 64;
            splits = cdiv(kp, kps);
        }
        // r06 mid forms (launch_fast): the default 32-deep slice form of the 64 / 128-row tiles, un-split (fp32 output: never the 64-deep auto slice form)
        if (BM != 256 && g_x3_products == 3 && splits == 1 && !g_tile_policy && M > 192 && (g_x3_slice == 0 || g_x3_slice == 3) && !(g_x3_auto_slice && BM == 64)) {
            int mid = g_mid_form.load();
            if (mid == 0) mid = psalm_get_tuning(PSALM_TUNE_GEMM_MID) ? select_mid_form(M, N, kp) : 9;
            if (mid == 6 || mid == 8) { BM = 256; BN = 128; }
            else if (mid == 7) { BM = 128; BN = 256; }
            else if (mid == 10) { BM = 128; BN = 128; }
            else if (mid == 11) { BM = 64; BN = 128; }
        }
        out4[0] = 1; out4[1] = BM; out4[2] = BN; out4[3] = splits;
    } else if (a_dtype == PSALM_BF16 && w_dtype == PSALM_BF16 && K % 64 == 0 && M <= 128 && N <= g_skinny_nmax && !g_tile_policy) {
        out4[0] = 2; out4[1] = 32; out4[2] = 32; out4[3] = 1;                     // skinny kernel
    } else if (a_dtype == PSALM_BF16 && w_dtype == PSALM_BF16 && K % 64 == 0) {
        int BM, BN, splits;
        select_fast_config(M, N, K, workspace_bytes > 0, workspace_bytes, BM, BN, splits);
        out4[0] = 1; out4[1] = BM; out4[2] = BN; out4[3] = splits;
    } else {
        const bool small = (long)cdiv(M, 128) * cdiv(N, 128) < 256 || M <= 64;
        out4[0] = 0; out4[1] = small ? 64 : 128; out4[2] = 128; out4[3] = 1;
    }
    return 0;
}

// Launch of the direct-to-LDS kernel (plain GEMM or implicit-GEMM convolution) + split-K reduce.
struct LnEpilogue { const float* gamma; const float* beta; float eps; void* out; int dtype; long ld; void* split_out = nullptr; float* split_inv = nullptr; };
extern "C" int psalm_layernorm_split(const float* x, long ldx, float* y, long ldy, const float* gamma, const float* beta, int rows, int C,
                                     float eps, void* split1, float* inv1, const float* add, long add_rows, void* split2, float* inv2,
                                     void* stream);
extern "C" int psalm_layernorm(const void* x, int x_dtype, long ldx, void* y, int y_dtype, long ldy, void* y2_bf16, long ldy2,
                               const float* gamma, const float* beta, int rows, int C, float eps, void* stream);

static int launch_fast(GemmArgs g, GemmFastArgs fa, bool conv, int c_dtype, void* workspace, long workspace_bytes, hipStream_t s,
                       const LnEpilogue* ln = nullptr, bool x3 = false) {
    const int M = g.M, N = g.N, K = g.K;
    int BM, BN, splits;
    select_fast_config(M, N, K, workspace != nullptr, workspace_bytes, BM, BN, splits, x3);
    if (fa.so) splits = 1;                                        // split-f16 output is written by the GEMM epilogue itself: no split-K
    int kps = splits > 1 ? cdiv(cdiv(K, 64), splits) * 64 : K;
    const int x3s = g_x3_slice.load();
    int slice = !x3 || BM == 256 ? 0 : (x3s == 5 ? 0 : x3s ? x3s : (g_x3_auto_slice && BM == 64 && !fa.so ? 1 : 3));
    // r05 ring depths (one block per CU by LDS: for grids that put <= 1 block on a CU anyway, the stages that a second block would have used
    // buy prefetch distance instead): 7 = 64 x 128, 64-deep slices, 3 stages (144 KB);  8 = 128 x 128, 32-deep slices, 3 stages (96 KB)
    if (slice == 7 && BM != 64) slice = 3;
    if (slice == 8 && BM != 128) slice = 3;
    if ((slice == 7 || slice == 8) && fa.so) slice = 3;
    if (x3 && BM == 256 && g_ph8 && g_ph8_slice) {        // 256 x 256: the phased loop on 32-deep slices (form 6) when every K slice of the grid has >= 2 of them
        const int kp_ = fa.x3_kp, kps_ = splits > 1 ? cdiv(cdiv(kp_, 64), splits) * 64 : kp_, sp_ = cdiv(kp_, kps_);
        if (kps_ >= 64 && kp_ - (sp_ - 1) * kps_ >= 64) slice = 6;
    }
    const bool p1 = x3 && g_x3_products == 1;                    // hi.hi only: the K-panel kernels over the first Kp of their 3 Kp-long range
    if (p1) {                                                    // (tiles / split count as chosen for the three-product problem above)
        slice = 0;
        g.K = fa.x3_kp;
        kps = splits > 1 ? cdiv(cdiv(g.K, 64), splits) * 64 : g.K;
        splits = cdiv(g.K, kps);
    }
    if (fa.so && slice != 3 && slice != 6) slice = 0;             // split-f16 output: K-panel form, form 3 or form 6
    // r06 mid forms: on the slice forms of the 64 / 128-row tiles (32-deep slices in two stages; the 64-deep "auto slice" form of the long-K,
    // few-tile problems), un-split by select_fast_config.  A form may bring its own split-K (fp32 output only): few large tiles x K slices.
    int mid = 0;
    if (x3 && (slice == 3 || slice == 1) && splits == 1 && !g_tile_policy && M > 192) {
        int msplit = 1;
        mid = g_mid_form.load();
        if (mid == 0) mid = psalm_get_tuning(PSALM_TUNE_GEMM_MID) ? select_mid_form(M, N, fa.x3_kp, &msplit) : 9;
        if (mid == 12 || mid == 13) {                            // experiment forms: 256 x 128 (four loaders) / 128 x 128 (two) with the K range split to fill the chip
            const long t = mid == 12 ? (long)cdiv(M, 256) * cdiv(N, 128) : (long)cdiv(M, 128) * cdiv(N, 128);
            msplit = mid_split_count(t, fa.x3_kp);
            mid = mid == 12 ? 8 : 10;
        }
        if (fa.so && fa.so_paired && fa.so_col_start % 256 != 0 && mid == 7) mid = 9;      // (paired stores: no tile straddles so_col_start)
        if (slice == 1 && !(msplit > 1 && workspace && !fa.so)) mid = 9;                   // the 64-deep form is only left for a split-K loader form
        if (mid == 6 || mid == 8) { BM = 256; BN = 128; }
        else if (mid == 7) { BM = 128; BN = 256; }
        else if (mid == 10) { BM = 128; BN = 128; }
        else if (mid == 11) { BM = 64; BN = 128; }
        else mid = 0;
        if (mid) {
            slice = 3;
            if (msplit > 1 && workspace && !fa.so && (long)msplit * M * N * 4 <= workspace_bytes) splits = msplit;
        }
    }
    if (slice) {                                                  // slice form: the kernel's K loop runs over the true (padded) K = Kp
        g.K = fa.x3_kp;
        kps = splits > 1 ? cdiv(cdiv(g.K, 64), splits) * 64 : g.K;
        splits = cdiv(g.K, kps);
    }
    g.tiles_m = cdiv(M, BM);
    g.tiles_n = cdiv(N, BN);
    g.row_fast = N > M ? 1 : 0;                                  // the larger operand's tiles stay in one XCD's L2
    const long tiles = (long)g.tiles_m * g.tiles_n;
    fa.g = g;
    fa.k_per_split = kps;
    fa.slab = splits > 1 ? (float*)workspace : nullptr;
    const long csz = c_dtype == PSALM_F32 ? 4 : 2;
    if (splits > 1) fa.vec_store = (N % 8 == 0) ? 1 : 0;                       // fp32 slab rows of N floats
    else fa.vec_store = (N % 8 == 0 && (uintptr_t)g.C % 16 == 0 && (g.ldc * csz) % 16 == 0 &&
                         (!g.res || ((uintptr_t)g.res % 16 == 0 && (g.ldr * csz) % 16 == 0))) ? 1 : 0;
    const dim3 grid((unsigned)tiles, splits);
    fa.xcd_ksplit = (splits > 1 && (splits % 8 == 0 || 8 % splits == 0) && psalm_get_tuning(PSALM_TUNE_GEMM_XCD_KSPLIT)) ? 1 : 0;
    if (fa.so && fa.so_paired) {                                  // paired stores: instantiated for the kernels the automatic selection uses
        const bool ph_ = x3 && !slice && BM == 256 && g_ph8 && fa.k_per_split >= 128 && (g.K - (splits - 1) * fa.k_per_split) >= 128;
        const bool kpanel_small = x3 && !slice && BM != 256;     // K-panel form on 128 x 128 / 64 x 128 tiles (reached by the one-product mode)
        if (!(slice == 3 || slice == 6 || ph_ || kpanel_small) || fa.so_col_start % BN != 0) {
            psalm_set_error("psalm_gemm_x3_split: paired output is not available under this tile policy / for this column start");
            return -1;
        }
    }
    const bool f32out = c_dtype == PSALM_F32 || splits > 1;   // partials are fp32 regardless of the output dtype
    // The direct fp32 epilogue addresses a tile's rows through a buffer descriptor of at most 0x7ffff000 bytes: a row stride that puts BM rows
    // beyond it would silently drop the tail rows -- refuse instead (2 M columns at BM = 256; the path's widest output is 65536).
    if (f32out && (long)BM * (splits > 1 ? (long)N : (g.ldc > g.ldr ? g.ldc : g.ldr)) * 4 >= 0x7ffff000L) {
        psalm_set_error("psalm_gemm: output / residual row stride too large for the direct fp32 epilogue (BM * ld * 4 must stay below 2 GiB)");
        return -1;
    }
    // every launch goes through GO(): it records the exact template instantiation (psalm_gemm_last_kernel: the name a kernel trace shows,
    // so that per-kernel attributions made from launch arguments -- bench.py -- agree with rocprofv3) and launches it
#define GO(NT_, TCN_, TC_, ...)                                                                                                              \
    do {                                                                                                                                     \
        snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_bf16_glds_kernel<" TCN_ ", " #__VA_ARGS__ ">");                                 \
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<TC_, __VA_ARGS__>), grid, dim3(NT_), 0, s, fa);                                           \
    } while (0)
    // (BM, BN, WM, WN, NS, CONV, BK, PH8, X3, SO) in the output type the call needs (fp32 results / slabs, or bf16)
#define GO_T(NT_, ...) do { if (f32out) GO(NT_, "float", float, __VA_ARGS__); else GO(NT_, "unsigned short", bf16_t, __VA_ARGS__); } while (0)
    if (slice) {                                                  // split-f16 slice form (see the kernel comment): K loop over the true K range
        // 3 / 4: 32-deep slices in a 2- / 3-deep ring -- the stage of the K-panel form (64 KB on 128^2: two blocks per CU stay resident)
        // with 1.5x the matrix work per copy round trip
        // (r06 experiment, measured and taken out again: the 256 x 256 tile as eight 32 x 256 matrix waves + four loader waves on two stages
        //  -- `GO(768, "float", float, 256, 256, 8, 1, 2, false, 32, 7, 2, false)`, 168 registers + 328 bytes of scratch -- returns the phased loop's
        //  words 25 - 50 % slower: Phi [k|v|q|fc1] 144 -> 202 us, [dense|fc2] 122 -> 185, 4096^3 321 -> 399; profiles/r06j_phi_lw256.txt)
        if (slice == 6 && g_ph8_slice == 2 && fa.so && fa.so_paired) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 4, 2, true, true);
        else if (slice == 6 && g_ph8_slice == 2 && fa.so) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 4, 2, true);
        else if (slice == 6 && g_ph8_slice == 2) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 4, 2, false);
        else if (slice == 6 && fa.so && fa.so_paired) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 3, 2, true, true);
        else if (slice == 6 && fa.so) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 3, 2, true);
        else if (slice == 6) GO(512, "float", float, 256, 256, 2, 4, 2, false, 32, 3, 2, false);
        // r06 mid forms: blocks with loader wavefronts (three stages of 32-deep slices)
#define GO_MID(NT_, ...)                                                                                                                       \
        do {                                                                                                                                   \
            if (fa.so && fa.so_paired) GO(NT_, "float", float, __VA_ARGS__, 2, true, true);                                                    \
            else if (fa.so) GO(NT_, "float", float, __VA_ARGS__, 2, true);                                                                     \
            else GO(NT_, "float", float, __VA_ARGS__, 2, false);                                                                               \
        } while (0)
        else if (mid == 6) GO_MID(640, 256, 128, 4, 2, 3, false, 32, 6);
        else if (mid == 7) GO_MID(640, 128, 256, 2, 4, 3, false, 32, 6);
        else if (mid == 8) GO_MID(768, 256, 128, 4, 2, 3, false, 32, 7);
        else if (mid == 10) GO_MID(384, 128, 128, 2, 2, 3, false, 32, 6);
        else if (mid == 11) GO_MID(384, 64, 128, 2, 2, 3, false, 32, 6);
#undef GO_MID
        else if (BM == 128 && slice >= 3 && fa.so && fa.so_paired) GO(256, "float", float, 128, 128, 2, 2, 2, false, 32, 0, 2, true, true);
        else if (slice == 3 && fa.so && fa.so_paired) GO(256, "float", float, 64, 128, 2, 2, 2, false, 32, 0, 2, true, true);
        else if (BM == 128 && slice >= 3 && fa.so) GO(256, "float", float, 128, 128, 2, 2, 2, false, 32, 0, 2, true);
        else if (BM == 128 && slice == 8) GO(256, "float", float, 128, 128, 2, 2, 3, false, 32, 0, 2, false);
        else if (BM == 128 && slice >= 3) GO(256, "float", float, 128, 128, 2, 2, 2, false, 32, 0, 2, false);
        else if (slice == 3 && fa.so) GO(256, "float", float, 64, 128, 2, 2, 2, false, 32, 0, 2, true);
        else if (slice == 3) GO(256, "float", float, 64, 128, 2, 2, 2, false, 32, 0, 2, false);
        else if (slice == 4) GO(256, "float", float, 64, 128, 2, 2, 3, false, 32, 0, 2, false);
        else if (slice == 7) GO(256, "float", float, 64, 128, 2, 2, 3, false, 64, 0, 2, false);
        else if (BM == 128 && slice == 2) GO(256, "float", float, 128, 128, 2, 2, 4, false, 32, 0, 2, false);
        else if (BM == 128) GO(256, "float", float, 128, 128, 2, 2, 2, false, 64, 0, 2, false);
        else if (slice == 2) GO(256, "float", float, 64, 128, 2, 2, 4, false, 32, 0, 2, false);
        else GO(256, "float", float, 64, 128, 2, 2, 2, false, 64, 0, 2, false);
    } else if (x3) {                                              // split-f16 variant: fp32 output (or fp32 split-K slabs) only
        const bool ph = BM == 256 && g_ph8 && fa.k_per_split >= 128 && (g.K - (splits - 1) * fa.k_per_split) >= 128;
        const int r64 = g_ring64.load(), ring64 = r64 ? r64 : (fa.k_per_split >= 1024 ? 3 : 2);
#define GO_X3(NT_, ...) do { if (fa.so) GO(NT_, "float", float, __VA_ARGS__, true); else GO(NT_, "float", float, __VA_ARGS__, false); } while (0)
        if (ph && fa.so && fa.so_paired) GO(512, "float", float, 256, 256, 2, 4, 2, false, 64, 3, 1, true, true);
        else if (fa.so && fa.so_paired && BM == 128) GO(256, "float", float, 128, 128, 2, 2, 2, false, 64, 0, 1, true, true);
        else if (fa.so && fa.so_paired && BM == 64) GO(256, "float", float, 64, 128, 2, 2, 2, false, 64, 0, 1, true, true);
        else if (ph) GO_X3(512, 256, 256, 2, 4, 2, false, 64, 3, 1);
        else if (BM == 256) GO_X3(512, 256, 256, 2, 4, 2, false, 64, 0, 1);
        else if (BM == 128 && g_ring_depth == 3) GO_X3(256, 128, 128, 2, 2, 3, false, 64, 0, 1);
        else if (BM == 128) GO_X3(256, 128, 128, 2, 2, 2, false, 64, 0, 1);
        else if (ring64 >= 3) GO_X3(256, 64, 128, 2, 2, 3, false, 64, 0, 1);
        else GO_X3(256, 64, 128, 2, 2, 2, false, 64, 0, 1);
#undef GO_X3
    } else if (conv) {
        if (BM == 256) GO_T(512, 256, 256, 2, 4, 2, true, 64, 0, 0, false);
        else if (BM == 128) GO_T(256, 128, 128, 2, 2, 2, true, 64, 0, 0, false);
        else GO_T(256, 64, 128, 2, 2, 2, true, 64, 0, 0, false);
    } else {
        if (BM == 256 && g_ph8 && fa.k_per_split % 64 == 0 && K % 64 == 0 && fa.k_per_split >= 128 &&
            (K - (splits - 1) * fa.k_per_split) >= 128) {
            if (g_ph8 == 1) GO_T(512, 256, 256, 2, 4, 2, false, 64, 1, 0, false);
            else if (g_ph8 == 2) GO_T(512, 256, 256, 2, 4, 2, false, 64, 2, 0, false);
            else GO_T(512, 256, 256, 2, 4, 2, false, 64, 3, 0, false);
        }
        else if (BM == 256) GO_T(512, 256, 256, 2, 4, 2, false, 64, 0, 0, false);
        else if (BM == 128 && BN == 64) GO_T(256, 128, 64, 2, 2, 2, false, 64, 0, 0, false);
        else if (BM == 128) {
            if (g_ring_depth == 3) GO_T(256, 128, 128, 2, 2, 3, false, 64, 0, 0, false);
            else if (g_ring_depth == 128 && K % 128 == 0 && fa.k_per_split % 128 == 0) GO_T(256, 128, 128, 2, 2, 2, false, 128, 0, 0, false);
            else if (g_ring_depth == 324) GO_T(256, 128, 128, 2, 2, 4, false, 32, 0, 0, false);
            else if (g_ring_depth == 323) GO_T(256, 128, 128, 2, 2, 3, false, 32, 0, 0, false);
            else GO_T(256, 128, 128, 2, 2, 2, false, 64, 0, 0, false);
        }
        else {
            const int r64 = g_ring64.load(), ring = r64 ? r64 : (fa.k_per_split >= 1024 ? 3 : 2);
            if (ring == 4) GO_T(256, 64, 128, 2, 2, 4, false, 64, 0, 0, false);
            else if (ring == 3) GO_T(256, 64, 128, 2, 2, 3, false, 64, 0, 0, false);
            else GO_T(256, 64, 128, 2, 2, 2, false, 64, 0, 0, false);
        }
    }
#undef GO_T
#undef GO
    auto also = [&](const char* k2) { strncat(g_last_kernel, " + ", sizeof(g_last_kernel) - strlen(g_last_kernel) - 1);
                                      strncat(g_last_kernel, k2, sizeof(g_last_kernel) - strlen(g_last_kernel) - 1); };
    if (splits > 1) also(ln ? "splitk_reduce_ln_kernel" : "splitk_reduce_kernel");
    else if (ln) also(ln->split_out ? "layernorm_split_kernel" : "layernorm_vec_kernel");
    if (splits > 1) {
        if (ln && ln->split_out) {                                // ... + the normalised rows in split-f16 form (psalm_gemm_x3_ln_split)
#define RLNS_LAUNCH(NV_) hipLaunchKernelGGL((splitk_reduce_ln_kernel<float, NV_, true>), dim3(M), dim3(256), 0, s, g, (const float*)workspace, splits, \
                                            ln->gamma, ln->beta, ln->eps, (float*)ln->out, ln->ld, (unsigned short*)ln->split_out, ln->split_inv, N)
            const int nv = N <= 1024 ? 1 : (N <= 2048 ? 2 : (N <= 4096 ? 4 : 8));
            if (nv == 1) RLNS_LAUNCH(1); else if (nv == 2) RLNS_LAUNCH(2); else if (nv == 4) RLNS_LAUNCH(4); else RLNS_LAUNCH(8);
#undef RLNS_LAUNCH
            PSALM_LAUNCH_END("psalm_gemm_x3_ln_split");
        }
        if (ln) {                                                 // reduce + epilogue + LayerNorm in one pass (fp32 C, checked by the caller)
#define RLN_LAUNCH(TL_, NV_) hipLaunchKernelGGL((splitk_reduce_ln_kernel<TL_, NV_>), dim3(M), dim3(256), 0, s, g, (const float*)workspace, splits, \
                                                ln->gamma, ln->beta, ln->eps, (TL_*)ln->out, ln->ld)
            const int nv = N <= 1024 ? 1 : (N <= 2048 ? 2 : (N <= 4096 ? 4 : 8));       // 4-column vectors per thread
            if (ln->dtype == PSALM_F32) {
                if (nv == 1) RLN_LAUNCH(float, 1); else if (nv == 2) RLN_LAUNCH(float, 2); else if (nv == 4) RLN_LAUNCH(float, 4); else RLN_LAUNCH(float, 8);
            } else {
                if (nv == 1) RLN_LAUNCH(bf16_t, 1); else if (nv == 2) RLN_LAUNCH(bf16_t, 2); else if (nv == 4) RLN_LAUNCH(bf16_t, 4); else RLN_LAUNCH(bf16_t, 8);
            }
#undef RLN_LAUNCH
            PSALM_LAUNCH_END("psalm_gemm_ln");
        }
        const long n4 = (N + 3) / 4;
        const dim3 rgrid((unsigned)(((long)M * n4 + 255) / 256));
        if (c_dtype == PSALM_F32) hipLaunchKernelGGL((splitk_reduce_kernel<float>), rgrid, dim3(256), 0, s, g, (const float*)workspace, splits);
        else hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), rgrid, dim3(256), 0, s, g, (const float*)workspace, splits);
    }
    if (ln) {                                                     // un-split GEMM: the LayerNorm runs as its own (vectorised) kernel
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { psalm_set_error("psalm_gemm_ln: GEMM launch failed"); return (int)e; }
        if (ln->split_out)
            return psalm_layernorm_split((const float*)g.C, g.ldc, (float*)ln->out, ln->ld, ln->gamma, ln->beta, M, N, ln->eps, ln->split_out,
                                         ln->split_inv, nullptr, 0, nullptr, nullptr, (void*)s);
        return psalm_layernorm(g.C, PSALM_F32, g.ldc, ln->out, ln->dtype, ln->ld, nullptr, 0, ln->gamma, ln->beta, M, N, ln->eps, (void*)s);
    }
    PSALM_LAUNCH_END("psalm_gemm");
}

// Convolution as an implicit GEMM on the direct-to-LDS kernel: no im2col matrix in HBM.
//   x (B,H,W,Cin) bf16 NHWC;  Wt (Cout, k*k*Cin) bf16 with K order (ky, kx, c) (= weight.permute(0,2,3,1));  bias (Cout) f32 or NULL;
//   residual / out (B*Ho*Wo, Cout) c_dtype, row strides ldr / ldc;  Cin % 64 == 0;  zeros: >= 16 bytes of zeros on the device.
// Replaces F.conv2d at multimodal_projector/builder.py:85-111 (3x3 s2 / 3x3 s1 / 1x1 s2) and msdeformattn.py:248-254 (FPN 3x3).
extern "C" int psalm_conv2d_nhwc(const void* x, int B, int H, int W, int Cin, const void* Wt, int Cout, int ksize, int stride, int pad,
                                 const float* bias, const void* residual, long ldr, void* out, int c_dtype, long ldc, int act,
                                 const void* zeros, void* workspace, long workspace_bytes, void* stream) {
    PSALM_CHECK_ARG(Cin % 64 == 0 && Cin > 0, "psalm_conv2d_nhwc: Cin must be a multiple of 64");
    PSALM_CHECK_ARG(ksize >= 1 && stride >= 1 && pad >= 0 && zeros != nullptr, "psalm_conv2d_nhwc: bad geometry / zeros buffer missing");
    PSALM_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)Wt % 16 == 0 && (uintptr_t)zeros % 16 == 0, "psalm_conv2d_nhwc: 16-byte aligned operands");
    PSALM_CHECK_ARG(c_dtype == PSALM_F32 || c_dtype == PSALM_BF16, "psalm_conv2d_nhwc: bad output dtype");
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    if (B == 0 || Ho <= 0 || Wo <= 0 || Cout == 0) return 0;
    GemmArgs g;
    g.A = x; g.W = Wt; g.bias = bias; g.res = residual; g.C = out;
    g.lda = 0; g.ldw = (long)ksize * ksize * Cin; g.ldr = ldr; g.ldc = ldc;
    g.M = B * Ho * Wo; g.N = Cout; g.K = ksize * ksize * Cin; g.act = act; g.act_col_start = 0;
    g.row_fast = 0; g.tiles_m = g.tiles_n = 0;
    GemmFastArgs fa;
    fa.cH = H; fa.cW = W; fa.cC = Cin; fa.cK = ksize; fa.cS = stride; fa.cP = pad; fa.cHo = Ho; fa.cWo = Wo;
    fa.zeros = (const bf16_t*)zeros;
    fa.a_scale = fa.w_scale = nullptr;
    fa.x3_kp = 0;
    return launch_fast(g, fa, true, c_dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

// C = act(A . W^T + bias) + residual.   A (M,K) lda, dtype a_dtype;  W (N,K) ldw, dtype w_dtype (selects the
// arithmetic mode);  bias (N) f32 or NULL;  residual (M,N) ldr, dtype c_dtype, or NULL;  C (M,N) ldc, c_dtype.
// Constraints: K % 8 == 0; 16-byte aligned row starts (lda*sizeof % 16 == 0 etc.);  w f32 requires a f32.
// workspace (optional, workspace_bytes): scratch for split-K partials; without it small-grid GEMMs run un-split.
extern "C" int psalm_gemm(const void* A, int a_dtype, long lda, const void* W, int w_dtype, long ldw, const float* bias,
                          const void* residual, long ldr, void* C, int c_dtype, long ldc, int M, int N, int K, int act,
                          int act_col_start, void* workspace, long workspace_bytes, void* stream) {
    if (M == 0 || N == 0) return 0;
    PSALM_CHECK_ARG(K > 0 && K % 8 == 0, "psalm_gemm: K must be a positive multiple of 8");
    const long asz = a_dtype == PSALM_F32 ? 4 : 2, wsz = w_dtype == PSALM_F32 ? 4 : 2;
    PSALM_CHECK_ARG(((uintptr_t)A % 16 == 0) && (lda * asz) % 16 == 0, "psalm_gemm: A rows must be 16-byte aligned");
    PSALM_CHECK_ARG(((uintptr_t)W % 16 == 0) && (ldw * wsz) % 16 == 0, "psalm_gemm: W rows must be 16-byte aligned");
    PSALM_CHECK_ARG(!(w_dtype == PSALM_F32 && a_dtype != PSALM_F32), "psalm_gemm: fp32 weights need fp32 activations");
    PSALM_CHECK_ARG(c_dtype == PSALM_F32 || c_dtype == PSALM_BF16, "psalm_gemm: bad output dtype");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.res = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act; g.act_col_start = act_col_start;
    g.row_fast = 0;
    g.tiles_n = cdiv(N, 128);
    hipStream_t s = (hipStream_t)stream;

    if (a_dtype == PSALM_BF16 && w_dtype == PSALM_BF16 && K % 64 == 0 && M <= 128 && N <= g_skinny_nmax && !g_tile_policy) {
        // ---- skinny path: latency-bound GEMMs of the predictor (measured r1x: ~3 us vs ~10 us per launch)
        const dim3 grid(cdiv(N, 32), cdiv(M, 32));
        snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_bf16_skinny_kernel<%s, false>", c_dtype == PSALM_F32 ? "float" : "unsigned short");
        if (c_dtype == PSALM_F32) hipLaunchKernelGGL((gemm_bf16_skinny_kernel<float>), grid, dim3(256), 0, s, g, SkinnyX3{nullptr, nullptr, 0});
        else hipLaunchKernelGGL((gemm_bf16_skinny_kernel<bf16_t>), grid, dim3(256), 0, s, g, SkinnyX3{nullptr, nullptr, 0});
        PSALM_LAUNCH_END("psalm_gemm");
    }
    if (a_dtype == PSALM_BF16 && w_dtype == PSALM_BF16 && K % 64 == 0) {
        // ---- direct-to-LDS fast path
        GemmFastArgs fa;
        fa.cH = fa.cW = fa.cC = fa.cK = fa.cS = fa.cP = fa.cHo = fa.cWo = 0;
        fa.zeros = nullptr;
        fa.a_scale = fa.w_scale = nullptr;
    fa.x3_kp = 0;
        return launch_fast(g, fa, false, c_dtype, workspace, workspace_bytes, s);
    }

    if (a_dtype == PSALM_F32 && w_dtype == PSALM_F32 && M <= 192 && N <= 8192 && K % 8 == 0 && !g_tile_policy) {
        // ---- exact-fp32 skinny path (mask-decoder GEMMs with M = 100 query rows in the fp32 / f16x3 modes)
        const dim3 grid(cdiv(N, 32), cdiv(M, 32));
        snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_skinny_kernel<%s, %d>", c_dtype == PSALM_F32 ? "float" : "unsigned short", K >= 256 ? 16 : 4);
        if (K >= 256) {                                           // 16 wavefronts split K
            if (c_dtype == PSALM_F32) hipLaunchKernelGGL((gemm_f32_skinny_kernel<float, 16>), grid, dim3(1024), 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_skinny_kernel<bf16_t, 16>), grid, dim3(1024), 0, s, g);
        } else {
            if (c_dtype == PSALM_F32) hipLaunchKernelGGL((gemm_f32_skinny_kernel<float, 4>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_skinny_kernel<bf16_t, 4>), grid, dim3(256), 0, s, g);
        }
        PSALM_LAUNCH_END("psalm_gemm");
    }
    // ---- register-staged path (fp32 activations converted on the fly, odd K, or exact fp32 arithmetic)
    g.tiles_n = cdiv(N, 128);
    // 128-row tiles unless that leaves the 256 CUs under-filled
    const bool small = (long)cdiv(M, 128) * g.tiles_n < 256 || M <= 64;
    const int BM = small ? 64 : 128;
    g.tiles_m = cdiv(M, BM);
    const dim3 grid(g.tiles_m * g.tiles_n), block(256);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%s%s, %d>", w_dtype == PSALM_BF16 ? "gemm_bf16_kernel" : "gemm_f32_kernel",
             w_dtype == PSALM_BF16 ? (a_dtype == PSALM_F32 ? "float, " : "unsigned short, ") : "", c_dtype == PSALM_F32 ? "float" : "unsigned short", BM);
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
        } else {
            if (BM == 128) hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, 128>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, 64>), grid, block, 0, s, g);
        }
    } else { psalm_set_error("psalm_gemm: bad weight dtype"); return -1; }
#undef LAUNCH_BF16
    PSALM_LAUNCH_END("psalm_gemm");
}

// Two exact-fp32 skinny GEMMs (psalm_gemm with float32 operands, M <= 192, N <= 8192, K % 8 == 0) as one launch:
// C_i = act_i(A_i . W_i^T + bias_i), i = 0, 1; contiguous operand rows (row strides K / N).  See gemm_f32_skinny_pair_kernel.
extern "C" int psalm_gemm_f32_pair(const float* A0, const float* W0, const float* bias0, float* C0, int M0, int N0, int K0, int act0,
                                   const float* A1, const float* W1, const float* bias1, float* C1, int M1, int N1, int K1, int act1, void* stream) {
    PSALM_CHECK_ARG(A0 && W0 && C0 && A1 && W1 && C1, "psalm_gemm_f32_pair: null operand");
    PSALM_CHECK_ARG(M0 > 0 && M1 > 0 && M0 <= 192 && M1 <= 192 && N0 > 0 && N1 > 0 && N0 <= 8192 && N1 <= 8192 && K0 > 0 && K1 > 0 && K0 % 8 == 0 && K1 % 8 == 0 &&
                        (K0 >= 256) == (K1 >= 256),
                    "psalm_gemm_f32_pair: M <= 192, N <= 8192, K % 8 == 0, both K below or both from 256 (psalm_gemm's 4- / 16-wave skinny kernels)");
    PSALM_CHECK_ARG((uintptr_t)A0 % 16 == 0 && (uintptr_t)W0 % 16 == 0 && (uintptr_t)A1 % 16 == 0 && (uintptr_t)W1 % 16 == 0, "psalm_gemm_f32_pair: 16-byte aligned operands");
    PSALM_CHECK_ARG(!((act0 | act1) & (ACT_BIAS_ROW | ACT_POST_RESIDUAL)), "psalm_gemm_f32_pair: plain activations only");
    GemmArgs g[2];
    const float* As[2] = {A0, A1}; const float* Ws[2] = {W0, W1}; const float* bs[2] = {bias0, bias1}; float* Cs[2] = {C0, C1};
    const int Ms[2] = {M0, M1}, Ns[2] = {N0, N1}, Ks[2] = {K0, K1}, acts[2] = {act0, act1};
    for (int i = 0; i < 2; ++i) {
        g[i].A = As[i]; g[i].W = Ws[i]; g[i].bias = bs[i]; g[i].res = nullptr; g[i].C = Cs[i];
        g[i].lda = Ks[i]; g[i].ldw = Ks[i]; g[i].ldr = 0; g[i].ldc = Ns[i];
        g[i].M = Ms[i]; g[i].N = Ns[i]; g[i].K = Ks[i]; g[i].act = acts[i]; g[i].act_col_start = 0;
        g[i].tiles_m = g[i].tiles_n = 0; g[i].row_fast = 0;
    }
    const dim3 grid(cdiv(std::max(N0, N1), 32), cdiv(std::max(M0, M1), 32), 2);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_f32_skinny_pair_kernel<float, %d>", K0 >= 256 ? 16 : 4);
    if (K0 >= 256) hipLaunchKernelGGL((gemm_f32_skinny_pair_kernel<float, 16>), grid, dim3(1024), 0, (hipStream_t)stream, g[0], g[1]);
    else hipLaunchKernelGGL((gemm_f32_skinny_pair_kernel<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, g[0], g[1]);
    PSALM_LAUNCH_END("psalm_gemm_f32_pair");
}

// psalm_gemm followed by LayerNorm over the N columns of the (fp32) result:  C as psalm_gemm,  ln_out = LN(C) * gamma + beta
// in ln_dtype (row stride ld_ln).  With split-K the LayerNorm is fused into the slab reduction (one pass over each row).
// Requires bf16 A / W with K % 64 == 0 (the direct-to-LDS path), c_dtype F32, N % 4 == 0, N <= 8192, 16-byte aligned rows.
// Replaces e.g. `x = x + dense(attn) + fc2(mlp)` + the next layer's `input_layernorm` (modeling_phi.py:263-300).
extern "C" int psalm_gemm_ln(const void* A, int a_dtype, long lda, const void* W, int w_dtype, long ldw, const float* bias,
                             const void* residual, long ldr, void* C, int c_dtype, long ldc, int M, int N, int K, int act,
                             int act_col_start, const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, int ln_dtype,
                             long ld_ln, void* workspace, long workspace_bytes, void* stream) {
    if (M == 0 || N == 0) return 0;
    PSALM_CHECK_ARG(a_dtype == PSALM_BF16 && w_dtype == PSALM_BF16 && K > 0 && K % 64 == 0, "psalm_gemm_ln: bf16 operands, K % 64 == 0");
    PSALM_CHECK_ARG(c_dtype == PSALM_F32 && N % 4 == 0 && N <= 8192, "psalm_gemm_ln: fp32 output, N % 4 == 0, N <= 8192");
    PSALM_CHECK_ARG(!(act & ACT_BIAS_ROW), "psalm_gemm_ln: row bias not supported");
    PSALM_CHECK_ARG(((uintptr_t)A % 16 == 0) && (lda * 2) % 16 == 0 && ((uintptr_t)W % 16 == 0) && (ldw * 2) % 16 == 0 &&
                        (uintptr_t)C % 16 == 0 && (ldc * 4) % 16 == 0 && (uintptr_t)ln_gamma % 16 == 0 && (uintptr_t)ln_beta % 16 == 0 &&
                        (!residual || ((uintptr_t)residual % 16 == 0 && (ldr * 4) % 16 == 0)),
                    "psalm_gemm_ln: 16-byte aligned rows");
    PSALM_CHECK_ARG(ln_dtype == PSALM_F32 || ln_dtype == PSALM_BF16, "psalm_gemm_ln: bad LayerNorm output dtype");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.res = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act; g.act_col_start = act_col_start;
    g.row_fast = 0; g.tiles_m = g.tiles_n = 0;
    GemmFastArgs fa;
    fa.cH = fa.cW = fa.cC = fa.cK = fa.cS = fa.cP = fa.cHo = fa.cWo = 0;
    fa.zeros = nullptr;
    fa.a_scale = fa.w_scale = nullptr;
    fa.x3_kp = 0;
    LnEpilogue ln{ln_gamma, ln_beta, ln_eps, ln_out, ln_dtype, ld_ln};
    return launch_fast(g, fa, false, c_dtype, workspace, workspace_bytes, (hipStream_t)stream, &ln);
}

// ------------------------------------------------------------------------------------------- split-f16 ("X3") path
// fp32-class GEMMs on the f16 matrix cores.  Plain bf16 operands (8 mantissa bits) cannot meet the reference's fp32 results to the
// north star's tolerance (mask IoU within 1e-3, identical labels): the masked-attention feedback of the mask decoder
// (mask2former_transformer_decoder.py:754-760, `sigmoid(mask) < 0.5` decides which keys a query may see) turns operand rounding into
// label flips; an operand-mantissa sweep on MI355X (tools/exp_bits.py, profiles/r02b_*) puts the threshold between 15 and 17 bits.
// A 22-bit operand is carried as TWO f16 values:  x * s = hi + lo,  hi = f16(x s),  lo = f16(x s - hi)  (both conversions round to
// nearest even; the subtraction is exact in fp32), with one power-of-two scale s per row that places the row's absolute maximum in
// [2^13, 2^14) -- inside f16's range with headroom, and small elements lose nothing that matters: lo goes subnormal only for
// |x| < 2^-16 amax, an absolute error below 2^-38 amax.  The product  A . W^T = sa sw (Ahi.Whi + Alo.Whi + Ahi.Wlo) + O(2^-22)
// is then ONE f16 GEMM over a 3x longer K panel, accumulated in fp32 by the MFMA, with the scales applied in the epilogue.
//
// psalm_split_f16: x (rows, K) f32, row stride ldx  ->  out (rows, 2 Kp) f16 = [hi (Kp) | lo (Kp)], Kp = K rounded up to 64 (pad
// columns zero), row stride ldo (elements), and inv_scale (rows) = 1 / s (a power of two; 1 for an all-zero row).  One wavefront / row.
// LPR lanes per row (16 / 32 / 64): short rows (K <= 128 / 256) share a wavefront so that every lane has a 32-byte vector to convert --
// with one wavefront per row a K = 128 row (Swin stage 0, 65536 rows) kept 16 of 64 lanes busy.
template <int LPR>
__global__ void __launch_bounds__(256) split_f16_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ out, long ldo,
                                                        float* __restrict__ inv_scale, int rows, int K, int Kp) {
    constexpr int RPW = 64 / LPR;                                // rows per wavefront
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const float* xr = x + (live ? row : 0) * ldx;
    float amax = 0.f;
    for (int c = sub * 8; c < K; c += LPR * 8) {
        float v[8];
        load8_f32(xr + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(v[k]));
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (!live) return;
    // s = 2^(13 - floor(log2 amax)), exponent clamped so that s and 1/s stay normal fp32 numbers
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    int se = 13 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    const bool zero = !(amax > 0.f) || !(amax < 3.0e38f);        // all-zero (or non-finite) row: leave it unscaled
    const float sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
    const float inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
    if (sub == 0) inv_scale[row] = inv;
    unsigned short* orow = out + row * ldo;
    for (int c = sub * 8; c < Kp; c += LPR * 8) {
        float v[8];
        if (c < K) load8_f32(xr + c, v);                         // K % 8 == 0: a vector is entirely inside or outside the row
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        unsigned hw[4], lw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h0, h1, l0, l1;
            psalm_split_words(v[2 * k] * sc, h0, l0);
            psalm_split_words(v[2 * k + 1] * sc, h1, l1);
            hw[k] = h0 | (h1 << 16);
            lw[k] = l0 | (l1 << 16);
        }
        *reinterpret_cast<u32x4_s*>(orow + c) = u32x4_s{hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<u32x4_s*>(orow + Kp + c) = u32x4_s{lw[0], lw[1], lw[2], lw[3]};
    }
}

// Few, long rows (Phi: 899 tokens x 2048 / 10240 columns): a whole 256-thread block per row -- one wavefront per row was a 20-iteration
// dependent load loop on 899 wavefronts (r02e: ~14 us per launch, 2.6 ms per image).
__global__ void __launch_bounds__(256) split_f16_row_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ out, long ldo,
                                                            float* __restrict__ inv_scale, int K, int Kp) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    float amax = 0.f;
    for (int c = tid * 8; c < K; c += 2048) {
        float v[8];
        load8_f32(xr + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(v[k]));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    int se = 13 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    const bool zero = !(amax > 0.f) || !(amax < 3.0e38f);
    const float sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
    const float inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
    if (tid == 0) inv_scale[row] = inv;
    unsigned short* orow = out + row * ldo;
    for (int c = tid * 8; c < Kp; c += 2048) {
        float v[8];
        if (c < K) load8_f32(xr + c, v);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        unsigned hw[4], lw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h0, h1, l0, l1;
            psalm_split_words(v[2 * k] * sc, h0, l0);
            psalm_split_words(v[2 * k + 1] * sc, h1, l1);
            hw[k] = h0 | (h1 << 16);
            lw[k] = l0 | (l1 << 16);
        }
        *reinterpret_cast<u32x4_s*>(orow + c) = u32x4_s{hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<u32x4_s*>(orow + Kp + c) = u32x4_s{lw[0], lw[1], lw[2], lw[3]};
    }
}

extern "C" int psalm_split_f16(const float* x, long ldx, void* out, long ldo, float* inv_scale, int rows, int K, void* stream) {
    if (rows == 0) return 0;
    const int Kp = (K + 63) / 64 * 64;
    PSALM_CHECK_ARG(K > 0 && K % 8 == 0 && (uintptr_t)x % 16 == 0 && (ldx * 4) % 16 == 0, "psalm_split_f16: K % 8 == 0, 16-byte aligned input rows");
    PSALM_CHECK_ARG((uintptr_t)out % 16 == 0 && (ldo * 2) % 16 == 0 && ldo >= 2L * Kp, "psalm_split_f16: output rows of >= 2*ceil64(K) f16, 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (Kp >= 1024 && rows <= 4096) {
        hipLaunchKernelGGL(split_f16_row_kernel, dim3(rows), dim3(256), 0, s, x, ldx, (unsigned short*)out, ldo, inv_scale, K, Kp);
        PSALM_LAUNCH_END("psalm_split_f16");
    }
    if (Kp <= 128) hipLaunchKernelGGL((split_f16_kernel<16>), dim3(cdiv(rows, 16)), dim3(256), 0, s, x, ldx, (unsigned short*)out, ldo, inv_scale, rows, K, Kp);
    else if (Kp <= 256) hipLaunchKernelGGL((split_f16_kernel<32>), dim3(cdiv(rows, 8)), dim3(256), 0, s, x, ldx, (unsigned short*)out, ldo, inv_scale, rows, K, Kp);
    else hipLaunchKernelGGL((split_f16_kernel<64>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, (unsigned short*)out, ldo, inv_scale, rows, K, Kp);
    PSALM_LAUNCH_END("psalm_split_f16");
}

// C = act((A . W^T) + bias) + residual from split-f16 operands:  A2 (M, 2 Kp) / W2 (N, 2 Kp) f16 [hi | lo] with row strides lda / ldw
// (elements) and per-row scales a_scale (M) / w_scale (N) as written by psalm_split_f16;  Kp % 64 == 0.  C / residual fp32.
// Same tile selection, split-K and epilogue as psalm_gemm (on a K range of 3 Kp); M <= 128 problems take the skinny kernel.
struct SplitOut { void* so; long ldso; int so_kp, so_col_off, so_col_start, so_global; float* so_inv; const float* so_par; int so_paired; };
static int gemm_x3_impl(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                        const float* bias, const void* residual, long ldr, void* C, long ldc, int M, int N, int act,
                        int act_col_start, void* workspace, long workspace_bytes, void* stream, const SplitOut* so, const char* name,
                        const LnEpilogue* ln = nullptr) {
    if (M == 0 || N == 0) return 0;
    PSALM_CHECK_ARG(Kp > 0 && Kp % 64 == 0, "psalm_gemm_x3: Kp must be a positive multiple of 64");
    PSALM_CHECK_ARG((uintptr_t)A2 % 16 == 0 && (lda * 2) % 16 == 0 && (uintptr_t)W2 % 16 == 0 && (ldw * 2) % 16 == 0 && lda >= 2L * Kp && ldw >= 2L * Kp,
                    "psalm_gemm_x3: 16-byte aligned operand rows of >= 2*Kp f16");
    PSALM_CHECK_ARG(a_scale && w_scale, "psalm_gemm_x3: scales required");
    GemmArgs g;
    g.A = A2; g.W = W2; g.bias = bias; g.res = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.M = M; g.N = N; g.K = 3 * Kp; g.act = act; g.act_col_start = act_col_start;
    g.row_fast = 0; g.tiles_m = g.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    if (!so && !ln && M <= 128 && N <= g_skinny_nmax && !g_tile_policy) {
        const dim3 grid(cdiv(N, 32), cdiv(M, 32));
        g.K = g_x3_products * Kp;                                // (1 product: the first Kp of the K panel = hi.hi)
        snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_bf16_skinny_kernel<float, true>");
        hipLaunchKernelGGL((gemm_bf16_skinny_kernel<float, true>), grid, dim3(256), 0, s, g, SkinnyX3{a_scale, w_scale, Kp});
        PSALM_LAUNCH_END(name);
    }
    GemmFastArgs fa;
    fa.cH = fa.cW = fa.cC = fa.cK = fa.cS = fa.cP = fa.cHo = fa.cWo = 0;
    fa.zeros = nullptr;
    fa.a_scale = a_scale; fa.w_scale = w_scale;
    fa.x3_kp = Kp;
    if (so) {
        fa.so = (unsigned short*)so->so; fa.ldso = so->ldso; fa.so_kp = so->so_kp; fa.so_col_off = so->so_col_off;
        fa.so_col_start = so->so_col_start; fa.so_global = so->so_global; fa.so_inv = so->so_inv; fa.so_par = so->so_par;
        fa.so_paired = so->so_paired;
    }
    return launch_fast(g, fa, false, PSALM_F32, workspace, workspace_bytes, s, ln, true);
}
extern "C" int psalm_gemm_x3(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                             const float* bias, const void* residual, long ldr, void* C, long ldc, int M, int N, int act,
                             int act_col_start, void* workspace, long workspace_bytes, void* stream) {
    return gemm_x3_impl(A2, lda, a_scale, W2, ldw, w_scale, Kp, bias, residual, ldr, C, ldc, M, N, act, act_col_start, workspace,
                        workspace_bytes, stream, nullptr, "psalm_gemm_x3");
}

// psalm_gemm_x3 whose output columns >= split_col_start leave the kernel as the split-f16 A operand of the NEXT GEMM (no fp32 round trip, no
// psalm_split_f16 pass): value (r, n) goes to row r of `split_out` (row stride ld_split f16, 16-byte aligned rows) at column
// split_col_off + (n - split_col_start) (hi) and split_kp columns further (lo), scaled by a per-row power of two derived from a magnitude
// BOUND -- see GemmFastArgs::so.  bound_par: 4 floats on the device {2^14 max_n sum_k |w_nk|, max |bias|, g1, g0}: the row bound is
// max(a_scale[r] * par[0] + par[1],  (global_rows ? max_r a_scale[r] : 0) * g1 + g0).  1/scale -> split_inv[r].  Columns below
// split_col_start are written to C as usual (C may be NULL when split_col_start == 0).  No residual, no split-K; N, split_col_start,
// split_col_off, split_kp multiples of 8.  The un-written columns of split_out (K padding of the consumer) are the caller's to zero.
extern "C" int psalm_gemm_x3_split(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                                   const float* bias, void* C, long ldc, int M, int N, int act, int act_col_start, void* split_out,
                                   long ld_split, int split_kp, int split_col_off, int split_col_start, int paired, float* split_inv,
                                   const float* bound_par, int global_rows, void* workspace, long workspace_bytes, void* stream) {
    // paired != 0: the W rows >= split_col_start were permuted by the caller for paired stores (GemmFastArgs::so_paired)
    PSALM_CHECK_ARG(paired == 0 || paired == 1, "psalm_gemm_x3_split: paired is 0 or 1");
    PSALM_CHECK_ARG(!paired || (split_col_start % 256 == 0 && (N - split_col_start) % 64 == 0 && (long)M * ld_split * 2 < (1L << 31)),
                    "psalm_gemm_x3_split: paired output needs split_col_start % 256 == 0, (N - split_col_start) % 64 == 0, split_out < 2 GiB");
    PSALM_CHECK_ARG(split_out && split_inv && bound_par, "psalm_gemm_x3_split: split output, scale array and bound parameters required");
    PSALM_CHECK_ARG(N % 8 == 0 && split_col_start % 8 == 0 && split_col_start >= 0 && split_col_start < N && split_col_off % 8 == 0 &&
                        split_col_off >= 0 && split_kp % 8 == 0 && (uintptr_t)split_out % 16 == 0 && (ld_split * 2) % 16 == 0 &&
                        split_col_off + (N - split_col_start) <= split_kp && ld_split >= 2L * split_kp,
                    "psalm_gemm_x3_split: N / column offsets multiples of 8, 16-byte aligned split rows of >= 2*split_kp f16");
    PSALM_CHECK_ARG(C || split_col_start == 0, "psalm_gemm_x3_split: C required for the columns below split_col_start");
    SplitOut so{split_out, ld_split, split_kp, split_col_off, split_col_start, global_rows, split_inv, bound_par, paired};
    return gemm_x3_impl(A2, lda, a_scale, W2, ldw, w_scale, Kp, bias, nullptr, 0, C, ldc, M, N, act, act_col_start, workspace, workspace_bytes,
                        stream, &so, "psalm_gemm_x3_split");
}


// x = (A . W^T) + bias + residual (fp32, written to C) followed by LayerNorm(x) leaving in split-f16 operand form (and, optionally, as fp32
// rows ln_out): the residual-add GEMM + the NEXT block's input LayerNorm of a pre-norm transformer layer (Phi: [dense | fc2] + residual, then
// input_layernorm of the following layer, modeling_phi.py:263-300).  With split-K (the usual case for this GEMM: few tiles, long K) the
// partial-sum reduce, epilogue, LayerNorm and split run as ONE row pass; otherwise the LayerNorm is psalm_layernorm_split on C.
// N % 64 == 0, N <= 2048; split_out rows of 2*N f16 (contiguous), split_inv (M).
extern "C" int psalm_gemm_x3_ln_split(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                                      const float* bias, const void* residual, long ldr, void* C, long ldc, int M, int N,
                                      const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, long ld_ln, void* split_out,
                                      float* split_inv, void* workspace, long workspace_bytes, void* stream) {
    PSALM_CHECK_ARG(N % 64 == 0 && N <= 2048 && split_out && split_inv && C, "psalm_gemm_x3_ln_split: N % 64 == 0, N <= 2048, outputs required");
    PSALM_CHECK_ARG((uintptr_t)C % 16 == 0 && (ldc * 4) % 16 == 0 && (uintptr_t)ln_gamma % 16 == 0 && (uintptr_t)ln_beta % 16 == 0 &&
                        (uintptr_t)split_out % 16 == 0 && (!ln_out || ((uintptr_t)ln_out % 16 == 0 && (ld_ln * 4) % 16 == 0)) &&
                        (!residual || ((uintptr_t)residual % 16 == 0 && (ldr * 4) % 16 == 0)),
                    "psalm_gemm_x3_ln_split: 16-byte aligned rows");
    LnEpilogue ln{ln_gamma, ln_beta, ln_eps, ln_out, PSALM_F32, ld_ln, split_out, split_inv};
    return gemm_x3_impl(A2, lda, a_scale, W2, ldw, w_scale, Kp, bias, residual, ldr, C, ldc, M, N, 0, 0, workspace, workspace_bytes, stream,
                        nullptr, "psalm_gemm_x3_ln_split", &ln);
}
