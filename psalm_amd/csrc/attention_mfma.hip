// Matrix-core (MFMA) attention kernels for the bf16 mode of the PSALM path (gfx950, 64-lane wavefronts).
//
//   psalm_causal_attention_mfma   Phi prefill attention (modeling_phi.py:189-245, eager softmax :137-160, RoPE :92-122)
//   psalm_mha_attention_mfma      predictor cross / self attention, split over keys (mask2former_transformer_decoder.py:645-666)
//   psalm_window_attention_mfma   Swin (shifted-)window attention, 12x12 windows, head_dim 32 (swin_trans.py:117-149,369-387)
//
// Formulation ("swapped QK^T", one wavefront per 32-query tile):
//   S^T[key][q]  = K . Q^T      v_mfma_f32_32x32x16_bf16 with A = K rows, B = Q rows
//                  -> a lane owns ONE query column (lane & 31) and 16 keys per 32-key sub-tile, so the softmax
//                     row statistics are lane-local (one exchange with lane ^ 32 per 64-key tile) and the
//                     accumulator rescale factor is a per-lane scalar;
//   O^T[d][q]   += V^T . P^T    A = V^T rows (d), B = P^T straight from the S^T accumulator registers: the MFMA k index
//                     is an arbitrary labelling of the 16 keys of a k-step as long as A and B agree, so k = 8*hi + j is
//                     mapped to the key the lane already holds in accumulator register 8*ks + j
//                     (key = 16*ks + 8*(j>>2) + 4*hi + (j&3)); V^T is then read as two 8-byte runs per operand and no
//                     cross-lane shuffle or LDS transpose of P is needed.
// A small pre-pass (phi_qkv_prep_kernel) applies the partial RoPE, folds 1/sqrt(d) into Q, and lays the three operands
// out head-major with the contraction index contiguous: Qr/Kr (b,h,token,64) and Vt (b,h,64,token), so K and V^T tiles
// are both plain row-major [64][64] bf16 images that go global -> LDS with global_load_lds (shared by the 4 waves of a
// block, double-buffered, the next tile in flight during the current tile's MFMAs).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct alignas(16) u32x4_a { unsigned x, y, z, w; };
struct alignas(8) u32x2_a { unsigned x, y; };

__device__ __forceinline__ unsigned pk2(float a, float b) { return pack_bf16x2(a, b); }

// ---------------------------------------------------------------------------------------------- pre-pass
// grid (Lp/64, heads, B), block 256.  buf (B*L, ld) bf16 with q/k/v column blocks.  Lp = ceil(L/64)*64.
//   Qr[((b*heads+h)*Lp + t)*64 + d] = rope(q)[t][d] * scale      (rows t >= L are zero)
//   Kr[...]                          = rope(k)[t][d]
//   Vt[((b*heads+h)*64 + d)*Lp + t]  = v[t][d]                     (columns t >= L are zero)
__global__ void __launch_bounds__(256) phi_qkv_prep_kernel(const bf16_t* __restrict__ buf, long ld, int q_off, int k_off,
                                                           int v_off, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, bf16_t* __restrict__ Qr,
                                                           bf16_t* __restrict__ Kr, bf16_t* __restrict__ Vt, int L, int Lp,
                                                           int heads, float scale) {
    constexpr int HD = 64, ROT = 32, HALF = 16;
    __shared__ bf16_t Vs[64][HD + 2];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const long bh = (long)b * heads + h;
    const int tl = tid >> 2, part = tid & 3;          // token within tile, 16-wide d quarter
    const int t = t0 + tl;
    const bool live = t < L;
    const bf16_t* row = buf + ((long)b * L + (live ? t : 0)) * ld;
    float qv[16], kv[16];
    auto ld16 = [](const bf16_t* p, float* dst) {                  // 16 consecutive bf16 (32 B, 16-byte aligned) -> fp32
        const u32x4_a a = reinterpret_cast<const u32x4_a*>(p)[0], c = reinterpret_cast<const u32x4_a*>(p)[1];
        const unsigned w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dst[2 * i] = __builtin_bit_cast(float, w[i] << 16);
            dst[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
        }
    };
    if (live) {
        const bf16_t* qp = row + q_off + h * HD;
        const bf16_t* kp = row + k_off + h * HD;
        ld16(qp + part * HALF, qv);
        ld16(kp + part * HALF, kv);
        if (part < 2) {                                // rotary quarter: needs its partner quarter (c +- 16)
            float qo[16], ko[16];
            ld16(qp + (1 - part) * HALF, qo);
            ld16(kp + (1 - part) * HALF, ko);
            const float sgn = part == 0 ? -1.f : 1.f;  // rotate_half(x) = cat(-x[16:32], x[0:16])
#pragma unroll
            for (int c = 0; c < HALF; ++c) {
                const float cs = cosT[(long)t * ROT + part * HALF + c], sn = sinT[(long)t * ROT + part * HALF + c];
                qv[c] = qv[c] * cs + sgn * qo[c] * sn;
                kv[c] = kv[c] * cs + sgn * ko[c] * sn;
            }
        }
        const bf16_t* vp = row + v_off + h * HD + part * HALF;
        const u32x4_a v0 = reinterpret_cast<const u32x4_a*>(vp)[0], v1 = reinterpret_cast<const u32x4_a*>(vp)[1];
        const unsigned vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Vs[tl][part * HALF + 2 * i] = (bf16_t)(vw[i] & 0xffffu);
            Vs[tl][part * HALF + 2 * i + 1] = (bf16_t)(vw[i] >> 16);
        }
    } else {
#pragma unroll
        for (int c = 0; c < HALF; ++c) { qv[c] = 0.f; kv[c] = 0.f; Vs[tl][part * HALF + c] = 0; }
    }
    {
        bf16_t* qd = Qr + (bh * Lp + t) * HD + part * HALF;
        bf16_t* kd = Kr + (bh * Lp + t) * HD + part * HALF;
        u32x4_a a0{pk2(qv[0] * scale, qv[1] * scale), pk2(qv[2] * scale, qv[3] * scale), pk2(qv[4] * scale, qv[5] * scale), pk2(qv[6] * scale, qv[7] * scale)};
        u32x4_a a1{pk2(qv[8] * scale, qv[9] * scale), pk2(qv[10] * scale, qv[11] * scale), pk2(qv[12] * scale, qv[13] * scale), pk2(qv[14] * scale, qv[15] * scale)};
        u32x4_a b0{pk2(kv[0], kv[1]), pk2(kv[2], kv[3]), pk2(kv[4], kv[5]), pk2(kv[6], kv[7])};
        u32x4_a b1{pk2(kv[8], kv[9]), pk2(kv[10], kv[11]), pk2(kv[12], kv[13]), pk2(kv[14], kv[15])};
        reinterpret_cast<u32x4_a*>(qd)[0] = a0; reinterpret_cast<u32x4_a*>(qd)[1] = a1;
        reinterpret_cast<u32x4_a*>(kd)[0] = b0; reinterpret_cast<u32x4_a*>(kd)[1] = b1;
    }
    __syncthreads();
    // transposed write: thread -> (d = tid>>2, 16 consecutive tokens)
    {
        const int d = tid >> 2, tq = (tid & 3) * 16;
        bf16_t* dst = Vt + (bh * HD + d) * Lp + t0 + tq;
        unsigned w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = (unsigned)Vs[tq + 2 * i][d] | ((unsigned)Vs[tq + 2 * i + 1][d] << 16);
        reinterpret_cast<u32x4_a*>(dst)[0] = u32x4_a{w[0], w[1], w[2], w[3]};
        reinterpret_cast<u32x4_a*>(dst)[1] = u32x4_a{w[4], w[5], w[6], w[7]};
    }
}

// ---------------------------------------------------------------------------------------------- main kernel
// grid (ceil(L/128), heads, B), block 256 = 4 wavefronts x 32 queries.  Heaviest (last) query blocks are scheduled first.
// K / V^T tiles of 64 keys are shared by the 4 waves through LDS: global_load_lds_dwordx4 into a double buffer, one
// barrier per tile; both images are [64 rows][64 bf16] with 128-byte rows and the same XOR swizzle as the GEMM operand
// tiles (16-byte slot p of row r holds chunk p ^ ((r >> 1) & 7), applied to the per-lane global source address).
__global__ void __launch_bounds__(256) causal_attention_mfma_kernel(const bf16_t* __restrict__ Qr, const bf16_t* __restrict__ Kr,
                                                                    const bf16_t* __restrict__ Vt,
                                                                    const unsigned char* __restrict__ key_mask,
                                                                    bf16_t* __restrict__ out, long ldo, int o_off, int L, int Lp,
                                                                    int heads) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2][2 * 64 * 64 + 4 * 32];     // [stage][K tile | V^T tile] (+ key-mask bit sets in stage 0's tail)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q0 = qb * 128 + wave * 32;
    const long bh = (long)b * heads + h;
    const bf16_t* Qh = Qr + bh * Lp * 64;
    const bf16_t* Kh = Kr + bh * Lp * 64;
    const bf16_t* Vh = Vt + bh * 64 * Lp;
    const int qi = q0 + n;                                     // this lane's query (rows >= L are zero padding)
    const int qrow = min(qi, Lp - 1);

    bf16x8 bq[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        bq[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a*>(Qh + (long)qrow * 64 + 16 * kk + 8 * hi));

    // per-lane sources of this wave's 1 KiB chunks: K chunks {wave, wave+4} (8 keys each), V^T chunks {wave, wave+4} (8 d rows)
    const int lrow = lane >> 3, slot = lane & 7;
    const bf16_t* ksrc[2];
    const bf16_t* vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + 4 * i) * 8 + lrow;
        const int c = (slot ^ ((r >> 1) & 7)) * 8;
        ksrc[i] = Kh + (long)r * 64 + c;                        // + k0 * 64 per tile
        vsrc[i] = Vh + (long)r * Lp + c;                        // + k0 per tile
    }
    auto issue = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            psalm_glds16(ksrc[i] + (long)k0 * 64, &smem[buf][(wave + 4 * i) * 8 * 64]);
            psalm_glds16(vsrc[i] + k0, &smem[buf][64 * 64 + (wave + 4 * i) * 8 * 64]);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    constexpr float NEG = -1.0e30f;
    float m = NEG, l = 0.f;
    const int fsw = (n >> 1) & 7;

    const int last_q = min(L - 1, qb * 128 + 127);
    const int ntiles = last_q / 64 + 1;                         // key tiles needed by the block's last query (<= 32: L <= 2048)
    issue(0, 0);
    // key-padding mask of every tile as 64-bit sets (bit j of set t <-> key 64t + j), built once: no dependent global load
    // inside the tile loop.  Wave w builds tiles w, w+4, ...
    unsigned long long* kbits_s = reinterpret_cast<unsigned long long*>(&smem[0][2 * 64 * 64]);
    for (int t = wave; t < ntiles; t += 4) {
        const int kj = t * 64 + lane;
        const unsigned long long bits = __ballot(kj < L && key_mask[(long)b * L + kj] != 0);
        if (lane == 0) kbits_s[t] = bits;
    }
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * 64, buf = kt & 1;
        __syncthreads();                                        // tile kt landed (vmcnt(0)); everyone is done with the other buffer
        if (kt + 1 < ntiles) issue(buf ^ 1, k0 + 64);
        if (k0 > q0 + 31) continue;                             // tile entirely in this wave's future (uniform per wave)
        const bf16_t* Ks = smem[buf];
        const bf16_t* Vs = smem[buf] + 64 * 64;
        const unsigned long long kbits = kbits_s[kt];
        const unsigned long long kb_hi = kbits >> (4 * hi);

        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 ka = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4_a*>(&Ks[(32 * t + n) * 64 + (((2 * kk + hi) ^ fsw) * 8)]));
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, bq[kk], s[t], 0, 0, 0);
            }
        }
        // mask + tile max.  Tiles entirely below the diagonal with no padded key (almost all of them) skip the mask arithmetic.
        float mloc = NEG;
        if (k0 + 63 <= q0 && kbits == ~0ull) {                  // wave-uniform
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;  // key index within the 64-key tile
                    const bool ok = ((kb_hi >> (kl - 4 * hi)) & 1ull) && (k0 + kl <= qi);
                    s[t][r] = ok ? s[t][r] : NEG;
                    mloc = fmaxf(mloc, s[t][r]);
                }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float mn = fmaxf(m, mloc);
        const float alpha = __expf(m - mn);
        m = mn;
        float psum = 0.f;
        unsigned pb[2][2][4];                                               // P^T as bf16 B operands [sub-tile][k-step][4 dwords]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float x0 = s[t][8 * ks + 2 * jj], x1 = s[t][8 * ks + 2 * jj + 1];
                    const float p0 = x0 > 0.5f * NEG ? __expf(x0 - mn) : 0.f;
                    const float p1 = x1 > 0.5f * NEG ? __expf(x1 - mn) : 0.f;
                    psum += p0 + p1;
                    pb[t][ks][jj] = pk2(p0, p1);
                }
        l = l * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pfrag = __builtin_bit_cast(bf16x8, u32x4_a{pb[t][ks][0], pb[t][ks][1], pb[t][ks][2], pb[t][ks][3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // V^T row d = 32*dt + n; keys 32t + 16ks + 4hi + {0..3} and + 8: chunks c0 = 4t + 2ks, c0 + 1
                    const bf16_t* vrow = &Vs[(32 * dt + n) * 64 + 4 * hi];
                    const u32x2_a v0 = *reinterpret_cast<const u32x2_a*>(vrow + (((4 * t + 2 * ks) ^ fsw) * 8));
                    const u32x2_a v1 = *reinterpret_cast<const u32x2_a*>(vrow + (((4 * t + 2 * ks + 1) ^ fsw) * 8));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4_a{v0.x, v0.y, v1.x, v1.y}), pfrag,
                                                                    o[dt], 0, 0, 0);
                }
            }
    }
    const float ltot = l + __shfl_xor(l, 32);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (qi < L) {
        bf16_t* op = out + ((long)b * L + qi) * ldo + o_off + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                    // 4 consecutive d per 8-byte store
                const int d = 32 * dt + 8 * g + 4 * hi;
                *reinterpret_cast<u32x2_a*>(op + d) =
                    u32x2_a{pk2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv), pk2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv)};
            }
    }
}

// Phi prefill attention on the matrix cores (bf16 in/out, fp32 softmax statistics and accumulation).
// qkv (B*L, ld) bf16 holds the q | k | v column blocks at q_off / k_off / v_off (+ h*64 per head); out (B*L, ldo) bf16
// receives the attention output at column o_off (+ h*64) -- it may alias the q block of `qkv` (in-place).
// cos/sin (>=L, 32) fp32 tables; key_mask (B,L) u8.  workspace: psalm_causal_attention_mfma_workspace() bytes,
// 16-byte aligned, owned by the caller.  head_dim 64, rotary 32 (Phi-1.5).
extern "C" long psalm_causal_attention_mfma_workspace(int B, int L, int heads) {
    const long Lp = (L + 63) / 64 * 64;
    return 3L * B * heads * Lp * 64 * (long)sizeof(bf16_t);
}

extern "C" int psalm_causal_attention_mfma(const void* qkv, long ld, int q_off, int k_off, int v_off, void* out, long ldo,
                                           int o_off, const float* cos_table, const float* sin_table,
                                           const unsigned char* key_mask, void* workspace, int B, int L, int heads, int head_dim,
                                           int rot, void* stream) {
    PSALM_CHECK_ARG(head_dim == 64 && rot == 32, "psalm_causal_attention_mfma: head_dim 64 / rotary 32 only (Phi-1.5)");
    PSALM_CHECK_ARG(L <= 2048, "psalm_causal_attention_mfma: L <= 2048 (Phi-1.5 max_position_embeddings)");
    PSALM_CHECK_ARG(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ((uintptr_t)qkv % 16) == 0,
                    "psalm_causal_attention_mfma: q/k/v column blocks must be 16-byte aligned (ld and offsets multiples of 8)");
    PSALM_CHECK_ARG(ldo % 4 == 0 && o_off % 4 == 0 && ((uintptr_t)out % 8) == 0 && ((uintptr_t)workspace % 16) == 0,
                    "psalm_causal_attention_mfma: alignment (ldo, o_off multiples of 4 elements; workspace 16 B)");
    if (B == 0 || L == 0) return 0;
    const int Lp = (L + 63) / 64 * 64;
    const long per = (long)B * heads * Lp * 64;
    bf16_t* Qr = (bf16_t*)workspace;
    bf16_t* Kr = Qr + per;
    bf16_t* Vt = Kr + per;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(phi_qkv_prep_kernel, dim3(Lp / 64, heads, B), dim3(256), 0, s, (const bf16_t*)qkv, ld, q_off, k_off, v_off,
                       cos_table, sin_table, Qr, Kr, Vt, L, Lp, heads, 1.0f / sqrtf((float)head_dim));
    hipLaunchKernelGGL(causal_attention_mfma_kernel, dim3(cdiv(L, 128), heads, B), dim3(256), 0, s, Qr, Kr, Vt, key_mask,
                       (bf16_t*)out, ldo, o_off, L, Lp, heads);
    PSALM_LAUNCH_END("psalm_causal_attention_mfma");
}


// ============================================================================================ Swin window attention
// One block = one (window, head); 5 wavefronts, wave w owns queries [32w, 32w+32) of the 144 (padded to 160).
// K (row-major, padded rows) and V^T (d-major) of the head are staged once in LDS; the head's relative-position-bias
// column (529 floats) and a per-key table {4*(yj*23+xj), shift-mask label} are staged next to them, so the bias gather
// (swin_trans.py:131-134) is one LDS read at byte address  Aq4 - Bk4  per score.  The whole 160-key score row of a
// query lives in accumulator registers (5 x 16 fp32), so the softmax is exact two-pass (no online rescale).
// Same swapped-operand trick as above: S^T = K.Q^T, O^T = V^T.P^T with P^T taken straight from the accumulators.
template <bool V> struct MaskTag { static constexpr bool value = V; };

template <int WS>
__global__ void __launch_bounds__(320) window_attention_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                                    const float* __restrict__ bias_table,
                                                                    bf16_t* __restrict__ out, int nWh, int nWw, int C, int heads,
                                                                    int shift) {
    constexpr int N = WS * WS, NT = (N + 31) / 32, NP = NT * 32, HD = 32;
    constexpr int KS = HD + 8;                    // K row pitch (bf16): 80 B -> conflict-free ds_read_b128
    constexpr int VS = NP + 4;                    // V^T row pitch (bf16): 328 B at NP=160 -> conflict-free ds_read_b64
    constexpr int NB = (2 * WS - 1) * (2 * WS - 1);
    static_assert(NT == 5, "specialised for 12x12 windows (144 tokens -> 5 tiles of 32)");
    __shared__ __attribute__((aligned(16))) bf16_t Ks[NP * KS];
    __shared__ __attribute__((aligned(16))) bf16_t Vts[HD * VS];
    __shared__ float Bs[NB];
    __shared__ __attribute__((aligned(16))) int Kt[NP];   // (4*(yj*(2WS-1)+xj)) | label << 16 ; padded keys: label 15
    const int win = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const long row0 = (long)win * N;
    const int wwin = win % (nWh * nWw);
    const int wh = wwin / nWw, ww = wwin % nWw;
    const bool last_r = shift > 0 && wh == nWh - 1, last_c = shift > 0 && ww == nWw - 1;
    auto label = [&](int yy, int xx) -> int {     // swin_trans.py:371-387 (slices (0,-ws), (-ws,-shift), (-shift,None))
        const int ly = last_r ? (yy < WS - shift ? 1 : 2) : 0;
        const int lx = last_c ? (xx < WS - shift ? 1 : 2) : 0;
        return ly * 3 + lx;
    };
    // ---- stage K rows, V^T, bias column, key table
    for (int c = tid; c < NP * 4; c += 320) {                      // 16-byte chunks: (key, d-quarter)
        const int key = c >> 2, dq = (c & 3) * 8;
        u32x4_a kx{0, 0, 0, 0}, vx{0, 0, 0, 0};
        if (key < N) {
            const bf16_t* p = qkv + (row0 + key) * 3 * C + h * HD + dq;
            kx = *reinterpret_cast<const u32x4_a*>(p + C);
            vx = *reinterpret_cast<const u32x4_a*>(p + 2 * C);
        }
        *reinterpret_cast<u32x4_a*>(&Ks[key * KS + dq]) = kx;
        const unsigned vw[4] = {vx.x, vx.y, vx.z, vx.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Vts[(dq + 2 * i) * VS + key] = (bf16_t)(vw[i] & 0xffffu);
            Vts[(dq + 2 * i + 1) * VS + key] = (bf16_t)(vw[i] >> 16);
        }
    }
    for (int e = tid; e < NB; e += 320) Bs[e] = bias_table[(long)e * heads + h];
    for (int k = tid; k < NP; k += 320) {
        const int yj = k / WS, xj = k % WS;
        Kt[k] = k < N ? ((4 * (yj * (2 * WS - 1) + xj)) | (label(yj, xj) << 16)) : (15 << 16);
    }
    __syncthreads();

    // ---- this wave's query tile
    const int qi = 32 * wave + n;                                   // query of this lane (>= N: padding)
    const int qc = qi < N ? qi : N - 1;
    bf16x8 bq[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        bq[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a*>(qkv + (row0 + qc) * 3 * C + h * HD + 16 * kk + 8 * hi));
    const int yi = qc / WS, xi = qc % WS;
    const int Aq4 = 4 * ((yi + WS - 1) * (2 * WS - 1) + xi + WS - 1);
    const int qlabel = label(yi, xi);
    const float scale = rsqrtf((float)HD);

    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 ka = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a*>(&Ks[(32 * t + n) * KS + 16 * kk + 8 * hi]));
            s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, bq[kk], s[t], 0, 0, 0);
        }
    }
    constexpr float NEG = -1.0e30f;
    float mx = NEG;
    const char* Bbytes = reinterpret_cast<const char*>(Bs);
    // scale + relative-position bias (+ shift mask): the -100 mask can only differ from 0 in windows of the last window row /
    // column of a shifted block (block-uniform), and padded keys (144..159) are the compile-time registers g >= 2 of tile 4.
    auto finish_scores = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * t + 8 * g >= N) {                                   // padded keys: no contribution
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[t][4 * g + i] = NEG;
                    continue;
                }
                // accumulator registers 4g..4g+3 hold keys 32t + 8g + 4hi + {0,1,2,3}: one 16-byte read of the key table
                const u32x4_a kt4 = *reinterpret_cast<const u32x4_a*>(&Kt[32 * t + 8 * g + 4 * hi]);
                const unsigned kv[4] = {kt4.x, kt4.y, kt4.z, kt4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = s[t][4 * g + i] * scale + *reinterpret_cast<const float*>(Bbytes + (Aq4 - (int)(kv[i] & 0xffffu)));
                    if (MASKED && (int)(kv[i] >> 16) != qlabel) v += -100.0f;
                    s[t][4 * g + i] = v;
                    mx = fmaxf(mx, v);
                }
            }
    };
    if (last_r || last_c) finish_scores(MaskTag<true>{});
    else finish_scores(MaskTag<false>{});
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned pb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float x0 = s[t][8 * ks + 2 * jj], x1 = s[t][8 * ks + 2 * jj + 1];
                const float p0 = x0 > 0.5f * NEG ? __expf(x0 - mx) : 0.f;
                const float p1 = x1 > 0.5f * NEG ? __expf(x1 - mx) : 0.f;
                sum += p0 + p1;
                pb[jj] = pk2(p0, p1);
            }
            const bf16_t* vp = &Vts[n * VS + 32 * t + 16 * ks + 4 * hi];
            const u32x2_a v0 = *reinterpret_cast<const u32x2_a*>(vp), v1 = *reinterpret_cast<const u32x2_a*>(vp + 8);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4_a{v0.x, v0.y, v1.x, v1.y}),
                                                        __builtin_bit_cast(bf16x8, u32x4_a{pb[0], pb[1], pb[2], pb[3]}), o, 0, 0, 0);
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    if (qi < N) {
        bf16_t* op = out + (row0 + qi) * C + h * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<u32x2_a*>(op + 8 * g + 4 * hi) =
                u32x2_a{pk2(o[4 * g] * inv, o[4 * g + 1] * inv), pk2(o[4 * g + 2] * inv, o[4 * g + 3] * inv)};
    }
}

// qkv (B*nW*ws*ws, 3C) bf16 rows in window_partition order; bias_table ((2ws-1)^2, heads) f32; out (B*nW*ws*ws, C) bf16.
// ws must be 12 (Swin-B / Swin-L of the reference, swin_trans.py:660-719), head_dim 32.
extern "C" int psalm_window_attention_mfma(const void* qkv, const float* bias_table, void* out, int B, int nWh, int nWw, int C,
                                           int heads, int ws, int shift, void* stream) {
    PSALM_CHECK_ARG(C == heads * 32, "psalm_window_attention_mfma: head_dim must be 32");
    PSALM_CHECK_ARG(ws == 12, "psalm_window_attention_mfma: window size must be 12");
    PSALM_CHECK_ARG(C % 8 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "psalm_window_attention_mfma: alignment");
    const int nwin = B * nWh * nWw;
    if (nwin == 0) return 0;
    hipLaunchKernelGGL((window_attention_mfma_kernel<12>), dim3(nwin, heads), dim3(320), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                       bias_table, (bf16_t*)out, nWh, nWw, C, heads, shift);
    PSALM_LAUNCH_END("psalm_window_attention_mfma");
}


// ============================================================================================ predictor MHA (split-KV)
// nn.MultiheadAttention core for <= 128 queries (Mask2Former: 100) against Lk keys (HW of a feature level, or the 100
// queries themselves), head_dim 32.  The key axis is split over blockIdx.x so that 8 heads x S splits fill the chip;
// each block (4 waves x 32 queries) streams its keys in 64-key tiles through LDS (global_load_lds, double-buffered):
//   K tile  [64 keys][32 d]  64-byte rows,  16-byte slot p of row r holds d-chunk  p ^ ((r >> 2) & 3)
//   V^T tile [32 d][64 keys] 128-byte rows, slot p of row r holds key-chunk        p ^ ((r >> 1) & 7)
// V arrives TRANSPOSED (Vt (heads*32, ldvt): produced by the value projection GEMM with swapped operands), so both
// images are row-major copies.  Swapped-operand MFMAs as in the kernels above.  With S > 1 each block writes its
// un-normalised partial (O, m, l); mha_combine_kernel merges them.  mask (Lq, Lk) u8 1 = blocked, ignored for rows
// flagged in row_all_masked (mask2former_transformer_decoder.py:647).
__global__ void __launch_bounds__(256) mha_attention_mfma_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                                 long ldk, const bf16_t* __restrict__ Vt, long ldvt,
                                                                 const unsigned char* __restrict__ mask,
                                                                 const unsigned char* __restrict__ row_all_masked,
                                                                 bf16_t* __restrict__ out, long ldo, float* __restrict__ part_o,
                                                                 float* __restrict__ part_ml, int Lq, int Lk, int heads,
                                                                 int keys_per_split, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2][2 * 64 * 32];     // [stage][K tile (2048) | V^T tile (2048)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z, S = gridDim.x;
    const int kbeg = sp * keys_per_split, kend = min(Lk, kbeg + keys_per_split);
    const int qi = 32 * wave + n;
    const int qc = min(qi, Lq - 1);
    const bf16_t* Qb = Q + (long)b * Lq * ldq;
    const bf16_t* Kb = K + (long)b * Lk * ldk;
    const bf16_t* Vb = Vt + (long)b * heads * 32 * ldvt;
    bf16x8 bq[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        bq[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a*>(Qb + (long)qc * ldq + h * 32 + 16 * kk + 8 * hi));
    const unsigned char* mrow = nullptr;
    if (mask && !(row_all_masked && row_all_masked[(long)b * Lq + qc])) mrow = mask + ((long)b * Lq + qc) * Lk;

    // this wave's chunks: K chunk `wave` (16 keys x 64 B), V^T chunk `wave` (8 d rows x 128 B)
    const int krow = 16 * wave + (lane >> 2), kslot = lane & 3;
    const int kdch = (kslot ^ ((krow >> 2) & 3)) * 8;
    const int vrow = 8 * wave + (lane >> 3), vslot = lane & 7;
    const int vch = (vslot ^ ((vrow >> 1) & 7)) * 8;
    const bf16_t* vsrc = Vb + (long)(h * 32 + vrow) * ldvt;
    auto issue = [&](int buf, int k0) {
        psalm_glds16(Kb + (long)min(k0 + krow, Lk - 1) * ldk + h * 32 + kdch, &smem[buf][wave * 16 * 32]);
        psalm_glds16(vsrc + min(k0 + vch, (int)ldvt - 8), &smem[buf][64 * 32 + wave * 8 * 64]);
    };

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    constexpr float NEG = -1.0e30f;
    float m = NEG, l = 0.f;
    const int fk = (n >> 2) & 3, fv = (n >> 1) & 7;
    const int ntiles = (kend - kbeg + 63) / 64;
    if (ntiles > 0) issue(0, kbeg);
    // mask bytes of the lane's 32 keys per tile as 8 dwords, fetched one tile ahead (no dependent load in front of the softmax)
    unsigned mw_cur[8], mw_nxt[8];
    auto load_mask = [&](unsigned* mw, int k0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kj = k0 + 32 * t + 8 * g + 4 * hi;
                mw[4 * t + g] = (mrow && kj < Lk) ? *reinterpret_cast<const unsigned*>(mrow + kj) : 0u;     // Lk % 4 == 0 (host check)
            }
    };
    if (ntiles > 0) load_mask(mw_cur, kbeg);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kbeg + kt * 64, buf = kt & 1;
        __syncthreads();
        if (kt + 1 < ntiles) { issue(buf ^ 1, k0 + 64); load_mask(mw_nxt, k0 + 64); }
        const bf16_t* Ks = smem[buf];
        const bf16_t* Vs = smem[buf] + 64 * 32;
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 ka = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_a*>(&Ks[(32 * t + n) * 32 + (((2 * kk + hi) ^ fk) * 8)]));
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, bq[kk], s[t], 0, 0, 0);
            }
        }
        float mloc = NEG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kj = k0 + 32 * t + 8 * g + 4 * hi;            // keys kj..kj+3 live in accumulator registers 4g..4g+3
                const unsigned mw = mw_cur[4 * t + g];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (kj + i < kend) && !((mw >> (8 * i)) & 0xffu);
                    const float v = ok ? s[t][4 * g + i] * scale : NEG;
                    s[t][4 * g + i] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float mn = fmaxf(m, mloc);
        const float alpha = __expf(m - mn);
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned pb[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float x0 = s[t][8 * ks + 2 * jj], x1 = s[t][8 * ks + 2 * jj + 1];
                    const float p0 = x0 > 0.5f * NEG ? __expf(x0 - mn) : 0.f;
                    const float p1 = x1 > 0.5f * NEG ? __expf(x1 - mn) : 0.f;
                    psum += p0 + p1;
                    pb[jj] = pk2(p0, p1);
                }
                const bf16_t* vr = &Vs[n * 64 + 4 * hi];
                const u32x2_a v0 = *reinterpret_cast<const u32x2_a*>(vr + (((4 * t + 2 * ks) ^ fv) * 8));
                const u32x2_a v1 = *reinterpret_cast<const u32x2_a*>(vr + (((4 * t + 2 * ks + 1) ^ fv) * 8));
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4_a{v0.x, v0.y, v1.x, v1.y}),
                                                            __builtin_bit_cast(bf16x8, u32x4_a{pb[0], pb[1], pb[2], pb[3]}), o, 0, 0, 0);
            }
        l = l * alpha + psum;
#pragma unroll
        for (int i = 0; i < 8; ++i) mw_cur[i] = mw_nxt[i];
    }
    const float ltot = l + __shfl_xor(l, 32);
    if (S == 1) {
        if (qi < Lq) {
            const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
            bf16_t* op = out + ((long)b * Lq + qi) * ldo + h * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<u32x2_a*>(op + 8 * g + 4 * hi) =
                    u32x2_a{pk2(o[4 * g] * inv, o[4 * g + 1] * inv), pk2(o[4 * g + 2] * inv, o[4 * g + 3] * inv)};
        }
        return;
    }
    // partials: part_o [b][h][split][128 q][32 d] fp32, part_ml [b][h][split][128][2]
    const long pbase = (((long)b * heads + h) * S + sp) * 128 + qi;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<psalm_f32x4*>(part_o + pbase * 32 + 8 * g + 4 * hi) = psalm_f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
    if (hi == 0) { part_ml[pbase * 2] = m; part_ml[pbase * 2 + 1] = ltot; }
}

// one wavefront per (b, q, h): lanes 0..31 = d; merge the S partial softmax states (coalesced 128-byte reads per split)
__global__ void __launch_bounds__(256) mha_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                          bf16_t* __restrict__ out, long ldo, int B, int Lq, int heads, int S) {
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (long)B * Lq * heads) return;
    const int h = (int)(wv % heads);
    const int q = (int)((wv / heads) % Lq), b = (int)(wv / ((long)heads * Lq));
    const long base = ((long)b * heads + h) * S * 128 + q;
    // lanes stride the splits for the max, then every lane walks all splits for its d (lanes >= 32 idle in the sum)
    float mx = -1.0e30f;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, part_ml[(base + (long)s * 128) * 2]);
    mx = wave_max(mx);
    const int d = lane & 31;
    float num = 0.f, den = 0.f;
    for (int s = (lane >> 5); s < S; s += 2) {                  // the two half-waves take alternate splits
        const long p = base + (long)s * 128;
        const float ms = part_ml[p * 2], ls = part_ml[p * 2 + 1];
        const float f = ls > 0.f ? __expf(ms - mx) : 0.f;
        den += ls * f;
        num += part_o[p * 32 + d] * f;
    }
    num += __shfl_xor(num, 32);
    den += __shfl_xor(den, 32);
    if (lane < 32) out[((long)b * Lq + q) * ldo + h * 32 + d] = f32_to_bf16(den > 0.f ? num / den : 0.f);
}

// q (B*Lq, *) row stride ldq;  k (B*Lk, *) row stride ldk;  vt (B*heads*32, ldvt): V TRANSPOSED, row (h*32+d), columns = keys,
// zero-padded to ldvt >= ceil(Lk/8)*8;  out (B*Lq, heads*32) row stride ldo, bf16.  Lq <= 128, head_dim 32.
// mask (B,Lq,Lk) u8 (needs Lk % 4 == 0) / row_all_masked (B,Lq) u8, both optional.
// workspace: psalm_mha_attention_mfma_workspace(B, heads, Lk) bytes (split-KV partials).
static int mha_splits(int heads, int B, int Lk) {
    int S = (256 + heads * B - 1) / (heads * B);                 // ~1 block per CU
    const int maxS = (Lk + 63) / 64;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return S;
}
extern "C" long psalm_mha_attention_mfma_workspace(int B, int heads, int Lk) {
    const long S = mha_splits(heads, B, Lk);
    return S > 1 ? (long)B * heads * S * 128 * (32 + 2) * (long)sizeof(float) : 0;
}
extern "C" int psalm_mha_attention_mfma(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out,
                                        long ldo, const unsigned char* mask, const unsigned char* row_all_masked, void* workspace,
                                        int B, int Lq, int Lk, int heads, int head_dim, void* stream) {
    PSALM_CHECK_ARG(head_dim == 32, "psalm_mha_attention_mfma: head_dim must be 32");
    PSALM_CHECK_ARG(Lq >= 1 && Lq <= 128, "psalm_mha_attention_mfma: 1 <= Lq <= 128");
    PSALM_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= (Lk + 7) / 8 * 8 &&
                        (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)vt % 16 == 0 && (uintptr_t)out % 8 == 0,
                    "psalm_mha_attention_mfma: operand alignment (row strides multiples of 8 elements, 16-byte bases)");
    PSALM_CHECK_ARG(!mask || Lk % 4 == 0, "psalm_mha_attention_mfma: masked attention needs Lk % 4 == 0");
    if (B == 0 || Lk == 0) return 0;
    const int S = mha_splits(heads, B, Lk);
    PSALM_CHECK_ARG(S == 1 || workspace, "psalm_mha_attention_mfma: workspace required");
    int kps = ((Lk + S - 1) / S + 63) / 64 * 64;
    const int Seff = (Lk + kps - 1) / kps;
    float* po = (float*)workspace;
    float* pml = po ? po + (long)B * heads * Seff * 128 * 32 : nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mha_attention_mfma_kernel, dim3(Seff, heads, B), dim3(256), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)vt, ldvt, mask, row_all_masked, (bf16_t*)out, ldo, po, pml, Lq, Lk, heads, kps,
                       1.0f / sqrtf((float)head_dim));
    if (Seff > 1)
        hipLaunchKernelGGL(mha_combine_kernel, dim3((unsigned)(((long)B * Lq * heads + 3) / 4)), dim3(256), 0, st, po, pml,
                           (bf16_t*)out, ldo, B, Lq, heads, Seff);
    PSALM_LAUNCH_END("psalm_mha_attention_mfma");
}
