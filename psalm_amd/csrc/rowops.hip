// Wavefront-per-row kernels: LayerNorm and the Swin data-movement ops with LayerNorm fused in,
// GroupNorm (NHWC), broadcast add, row gather / segment mean.  All statistics in fp32.
// HBM-bound streaming kernels: one 64-lane wave owns one row, lanes stride the channel dimension
// (coalesced 256 B / 128 B per wave-load), reductions by __shfl_xor butterflies (no LDS).
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------- LayerNorm core (one wave, one row)
// Two-pass (mean, then centred variance) like torch.nn.functional.layer_norm; the row is re-read from L1/L2.
// split-f16 emission helpers (f16x3 mode; see psalm_split_f16 in gemm.hip)
__device__ __forceinline__ void emit_split8(const float* o8, float sc, unsigned short* hi_dst, unsigned short* lo_dst) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned h0, h1, l0, l1;
        psalm_split_words(o8[2 * k] * sc, h0, l0);
        psalm_split_words(o8[2 * k + 1] * sc, h1, l1);
        hw[k] = h0 | (h1 << 16);
        lw[k] = l0 | (l1 << 16);
    }
    *reinterpret_cast<psalm_u32x4*>(hi_dst) = psalm_u32x4{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<psalm_u32x4*>(lo_dst) = psalm_u32x4{lw[0], lw[1], lw[2], lw[3]};
}
__device__ __forceinline__ void split_scale(float amax, float& sc, float& inv) {     // as psalm_split_f16: row maximum into [2^13, 2^14)
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    int se = 13 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    const bool zero = !(amax > 0.f) || !(amax < 3.0e38f);
    sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
    inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
}
template <typename TI>
__device__ __forceinline__ void row_stats(const TI* __restrict__ x, int C, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ldf(x + c);
    mean = wave_sum(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ldf(x + c) - mean; v += d * d; }
    rstd = rsqrtf(wave_sum(v) / C + eps);
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) layernorm_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                        bf16_t* __restrict__ y2, long ldy2, const float* __restrict__ add, long add_rows,
                                                        bf16_t* __restrict__ y3, long ldy3, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + row * ldx;
    float mean, rstd;
    row_stats(xr, C, lane, eps, mean, rstd);
    TO* yr = y + row * ldy;
    for (int c = lane; c < C; c += 64) {
        const float v = (ldf(xr + c) - mean) * rstd * gamma[c] + beta[c];
        stf(yr + c, v);
        if (y2) y2[row * ldy2 + c] = f32_to_bf16(v);
        if (y3) y3[row * ldy3 + c] = f32_to_bf16(v + add[(row % add_rows) * C + c]);
    }
}

// Vector form for C % 8 == 0, C <= 2048 and 16-byte aligned rows: the row is read ONCE, 8 consecutive channels per lane
// per step (16-byte bf16 / 2 x 16-byte fp32 accesses), kept in registers for the mean and the centred variance.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) layernorm_vec_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                            bf16_t* __restrict__ y2, long ldy2, const float* __restrict__ add,
                                                            long add_rows, bf16_t* __restrict__ y3, long ldy3,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int rows,
                                                            int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + row * ldx;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            ld8(xr + c, v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[i][k];
        }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            float g8[8], b8[8], o8[8];
            ld8(gamma + c, g8);
            ld8(beta + c, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = (v[i][k] - mean) * rstd * g8[k] + b8[k];
            st8(y + row * ldy + c, o8);
            if (y2) st8(y2 + row * ldy2 + c, o8);
            if (y3) {
                float a8[8];
                ld8(add + (row % add_rows) * C + c, a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) o8[k] += a8[k];
                st8(y3 + row * ldy3 + c, o8);
            }
        }
    }
}

// y2 (optional, bf16, row stride ldy2): a second copy of the result in the GEMM operand dtype, so a fp32 residual
// stream and the bf16 A operand of the next projection come out of one pass.
// y3 (optional, bf16, row stride ldy3) = result + add[row % add_rows] (add (add_rows, C) fp32): the "tensor + positional
// embedding" operand of the next attention projection (msdeformattn.py:51-58, mask2former_transformer_decoder.py:35-37,93-96).
extern "C" int psalm_layernorm3(const void* x, int x_dtype, long ldx, void* y, int y_dtype, long ldy, void* y2_bf16, long ldy2,
                                const float* add, long add_rows, void* y3_bf16, long ldy3, const float* gamma, const float* beta,
                                int rows, int C, float eps, void* stream) {
    if (rows == 0) return 0;
    const long xs = x_dtype == PSALM_F32 ? 4 : 2, ys = y_dtype == PSALM_F32 ? 4 : 2;
    PSALM_CHECK_ARG(!y3_bf16 || (add && add_rows > 0), "psalm_layernorm3: y3 needs the `add` table");
    const bool vec = C % 8 == 0 && C <= 2048 && (uintptr_t)x % 16 == 0 && (ldx * xs) % 16 == 0 && (uintptr_t)y % 16 == 0 &&
                     (ldy * ys) % 16 == 0 && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0 &&
                     (!y2_bf16 || ((uintptr_t)y2_bf16 % 16 == 0 && (ldy2 * 2) % 16 == 0)) &&
                     (!y3_bf16 || ((uintptr_t)y3_bf16 % 16 == 0 && (ldy3 * 2) % 16 == 0 && (uintptr_t)add % 16 == 0));
    PSALM_DISPATCH(x_dtype, TI, PSALM_DISPATCH(y_dtype, TO, {
        if (vec)
            hipLaunchKernelGGL((layernorm_vec_kernel<TI, TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                               (const TI*)x, ldx, (TO*)y, ldy, (bf16_t*)y2_bf16, ldy2, add, add_rows, (bf16_t*)y3_bf16, ldy3, gamma, beta,
                               rows, C, eps);
        else
            hipLaunchKernelGGL((layernorm_kernel<TI, TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                               (const TI*)x, ldx, (TO*)y, ldy, (bf16_t*)y2_bf16, ldy2, add, add_rows, (bf16_t*)y3_bf16, ldy3, gamma, beta,
                               rows, C, eps);
    }));
    PSALM_LAUNCH_END("psalm_layernorm");
}

// ---------------------------------------------------------------- LayerNorm chain of the mask decoder's query rows (fp32)
//   y1 = LN1(x)        y2 = y1 + add[row % add_rows]   (optional)        y3 = LN2(y1)   (optional)
// One launch for what mask2former_transformer_decoder.py runs as norm -> with_pos_embed(query, query_pos) -> decoder_norm (the cross-attention /
// FFN post-norms at :72-74,:170-172, `output + query_pos` at :35-37, forward_prediction_heads' decoder_norm at :750) and r05 issued as
// psalm_layernorm3 + psalm_add_bcast + psalm_layernorm3 on 100 x 256 values: three dependent ~6 us launches of the decoder's serial tail.
// Arithmetic, reduction order and stored intermediates are those of layernorm_vec_kernel / add_bcast_kernel (y1 is consumed from the registers
// it was stored from), so the chain returns the same words as the three launches (tests/test_1_ops.py).  C % 8 == 0, C <= 2048, 16-byte aligned rows.
__global__ void __launch_bounds__(256) layernorm_chain_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y1, long ldy1,
                                                              const float* __restrict__ g1, const float* __restrict__ b1,
                                                              const float* __restrict__ add, long add_rows, float* __restrict__ y2, long ldy2,
                                                              const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ y3,
                                                              long ldy3, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float v[4][8];
    auto stats = [&](float& mean, float& rstd) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[i][k];
            }
        }
        mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / C + eps);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) ld8(xr + c, v[i]);
    }
    float mean, rstd;
    stats(mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            float g8[8], b8[8];
            ld8(g1 + c, g8);
            ld8(b1 + c, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = (v[i][k] - mean) * rstd * g8[k] + b8[k];
            st8(y1 + row * ldy1 + c, v[i]);
            if (y2) {
                float a8[8], o8[8];
                ld8(add + (row % add_rows) * C + c, a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) o8[k] = v[i][k] + a8[k];
                st8(y2 + row * ldy2 + c, o8);
            }
        }
    }
    if (!y3) return;
    stats(mean, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            float g8[8], b8[8], o8[8];
            ld8(g2 + c, g8);
            ld8(b2 + c, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = (v[i][k] - mean) * rstd * g8[k] + b8[k];
            st8(y3 + row * ldy3 + c, o8);
        }
    }
}
extern "C" int psalm_layernorm_chain(const float* x, long ldx, float* y1, long ldy1, const float* g1, const float* b1, const float* add,
                                     long add_rows, float* y2, long ldy2, const float* g2, const float* b2, float* y3, long ldy3, int rows,
                                     int C, float eps, void* stream) {
    if (rows == 0) return 0;
    PSALM_CHECK_ARG(x && y1 && g1 && b1 && (!y2 || (add && add_rows > 0)) && (!y3 || (g2 && b2)), "psalm_layernorm_chain: null argument");
    auto al = [](const void* p, long ld) { return (uintptr_t)p % 16 == 0 && (ld * 4) % 16 == 0; };
    PSALM_CHECK_ARG(C % 8 == 0 && C > 0 && C <= 2048 && al(x, ldx) && al(y1, ldy1) && al(g1, 0) && al(b1, 0) && (!y2 || (al(y2, ldy2) && al(add, C))) &&
                        (!y3 || (al(y3, ldy3) && al(g2, 0) && al(b2, 0))),
                    "psalm_layernorm_chain: C % 8 == 0, C <= 2048, 16-byte aligned rows");
    hipLaunchKernelGGL(layernorm_chain_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y1, ldy1, g1, b1, add, add_rows, y2, ldy2,
                       g2, b2, y3, ldy3, rows, C, eps);
    PSALM_LAUNCH_END("psalm_layernorm_chain");
}

extern "C" int psalm_layernorm(const void* x, int x_dtype, long ldx, void* y, int y_dtype, long ldy, void* y2_bf16, long ldy2,
                               const float* gamma, const float* beta, int rows, int C, float eps, void* stream) {
    return psalm_layernorm3(x, x_dtype, ldx, y, y_dtype, ldy, y2_bf16, ldy2, nullptr, 0, nullptr, 0, gamma, beta, rows, C, eps, stream);
}

// ---------------------------------------------------------------- LayerNorm emitting the NEXT GEMM's A operand in split-f16 form (f16x3 mode)
// y = LN(x) (fp32, optional) and, from the same registers, split(y) and / or split(y + add[row % add_rows]) as psalm_split_f16 would write
// them ([hi (Kp) | lo (Kp)] f16 + per-row power-of-two scale): the separate split pass (read 8 B + write 4 B per element and one launch
// per GEMM) disappears for LayerNorm-fed projections -- Phi's [k|v|q|fc1] input, the pixel decoder's value / offset / FFN inputs.
// One wavefront per row, row in registers (C % 8 == 0, C <= 2048).
// LPR: lanes per row (r06; see swin_window_merge_ln_kernel): 32 puts two rows of C <= 256 on a wavefront (the pixel decoder's 21504 x 256 rows), 16 four of
// C <= 128; same bits as one row per wavefront.
template <int LPR>
__global__ void __launch_bounds__(256) layernorm_split_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int rows, int C,
                                                              float eps, unsigned short* __restrict__ s1, float* __restrict__ inv1,
                                                              const float* __restrict__ add, long add_rows, unsigned short* __restrict__ s2,
                                                              float* __restrict__ inv2, int Kp) {
    constexpr int RPW = 64 / LPR, NI = LPR == 64 ? 4 : 1;
    const int lane = (threadIdx.x & 63) % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (threadIdx.x & 63) / LPR;
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float v[NI][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < C) {
            ld8(xr + c, v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[i][k];
        }
    }
    const float mean = seg_sum<LPR>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(seg_sum<LPR>(q) / C + eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < C) {
            float g8[8], b8[8];
            ld8(gamma + c, g8);
            ld8(beta + c, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[i][k] = (v[i][k] - mean) * rstd * g8[k] + b8[k]; a1 = fmaxf(a1, fabsf(v[i][k])); }
            if (y) st8(y + row * ldy + c, v[i]);
            if (s2) {
                float d8[8];
                ld8(add + (row % add_rows) * C + c, d8);
#pragma unroll
                for (int k = 0; k < 8; ++k) a2 = fmaxf(a2, fabsf(v[i][k] + d8[k]));
            }
        }
    }
    float sc1, iv1, sc2 = 1.f, iv2 = 1.f;
    split_scale(seg_max<LPR>(a1), sc1, iv1);
    if (s2) split_scale(seg_max<LPR>(a2), sc2, iv2);
    if (lane == 0) {
        if (s1) inv1[row] = iv1;
        if (s2) inv2[row] = iv2;
    }
    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < Kp) {
            const bool in = c < C;
            if (s1) emit_split8(in ? v[i] : zero8, sc1, s1 + row * 2L * Kp + c, s1 + row * 2L * Kp + Kp + c);
            if (s2) {
                float t8[8];
                if (in) {
                    ld8(add + (row % add_rows) * C + c, t8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) t8[k] += v[i][k];
                }
                emit_split8(in ? t8 : zero8, sc2, s2 + row * 2L * Kp + c, s2 + row * 2L * Kp + Kp + c);
            }
        }
    }
}

// x (rows,C) f32 row stride ldx; y (rows,C) f32 row stride ldy or NULL; split1 / inv1: split(y) or NULL; split2 / inv2: split(y + add[row %
// add_rows]) or NULL (add (add_rows, C) f32).  Split rows are 2 * ceil64(C) f16, contiguous.  C % 8 == 0, C <= 2048, 16-byte aligned rows.
extern "C" int psalm_layernorm_split(const float* x, long ldx, float* y, long ldy, const float* gamma, const float* beta, int rows, int C,
                                     float eps, void* split1, float* inv1, const float* add, long add_rows, void* split2, float* inv2,
                                     void* stream) {
    if (rows == 0) return 0;
    PSALM_CHECK_ARG(C % 8 == 0 && C > 0 && C <= 2048, "psalm_layernorm_split: C % 8 == 0, C <= 2048");
    PSALM_CHECK_ARG((uintptr_t)x % 16 == 0 && (ldx * 4) % 16 == 0 && (!y || ((uintptr_t)y % 16 == 0 && (ldy * 4) % 16 == 0)) &&
                        (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0 && (!split1 || (uintptr_t)split1 % 16 == 0) &&
                        (!split2 || ((uintptr_t)split2 % 16 == 0 && add && add_rows > 0 && (uintptr_t)add % 16 == 0)),
                    "psalm_layernorm_split: 16-byte aligned rows; split2 needs the `add` table");
    PSALM_CHECK_ARG((!split1 || inv1) && (!split2 || inv2) && (split1 || split2 || y), "psalm_layernorm_split: outputs / scale arrays missing");
    const int Kp = (C + 63) / 64 * 64;
#define LNS_LAUNCH(LPR_) hipLaunchKernelGGL((layernorm_split_kernel<LPR_>), dim3(cdiv(rows, 4 * (64 / LPR_))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, \
                                            gamma, beta, rows, C, eps, (unsigned short*)split1, inv1, add, add_rows, (unsigned short*)split2, inv2, Kp)
    const int lpr = !psalm_get_tuning(PSALM_TUNE_ROW_GROUPS) ? 64 : (C <= 128 ? 16 : (C <= 256 ? 32 : 64));
    if (lpr == 16) LNS_LAUNCH(16); else if (lpr == 32) LNS_LAUNCH(32); else LNS_LAUNCH(64);
#undef LNS_LAUNCH
    PSALM_LAUNCH_END("psalm_layernorm_split");
}

// ---------------------------------------------------------------- Swin: LN1 + pad + cyclic shift + window partition
// swin_trans.py:206-227.  x (B,H,W,C) -> out (B*nWh*nWw*ws*ws, C); padded tokens are exact zeros (the pad is
// applied AFTER norm1, swin_trans.py:207-214); torch.roll(x,-s)[i] = x[(i+s) mod n].
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) swin_window_gather_kernel(const TI* __restrict__ x, TO* __restrict__ out,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int B, int H, int W, int C,
                                                                 int ws, int shift, float eps) {
    const int lane = threadIdx.x & 63;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws, N = ws * ws;
    const int Hp = nWh * ws, Wp = nWw * ws;
    const long rows = (long)B * nWh * nWw * N;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int tok = (int)(r % N);
    long t = r / N;
    const int ww = (int)(t % nWw);
    t /= nWw;
    const int wh = (int)(t % nWh);
    const int b = (int)(t / nWh);
    const int y = (wh * ws + tok / ws + shift) % Hp, xx = (ww * ws + tok % ws + shift) % Wp;
    TO* o = out + r * C;
    if (y < H && xx < W) {
        const TI* xr = x + (((long)b * H + y) * W + xx) * C;
        float mean, rstd;
        row_stats(xr, C, lane, eps, mean, rstd);
        for (int c = lane; c < C; c += 64) stf(o + c, (ldf(xr + c) - mean) * rstd * gamma[c] + beta[c]);
    } else {
        for (int c = lane; c < C; c += 64) stf(o + c, 0.f);
    }
}

extern "C" int psalm_swin_window_gather(const void* x, int x_dtype, void* out, int out_dtype, const float* gamma,
                                        const float* beta, int B, int H, int W, int C, int ws, int shift, float eps,
                                        void* stream) {
    const long rows = (long)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws) * ws * ws;
    if (rows == 0) return 0;
    PSALM_DISPATCH(x_dtype, TI, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((swin_window_gather_kernel<TI, TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const TI*)x, (TO*)out, gamma, beta, B, H, W, C, ws, shift, eps);
    }));
    PSALM_LAUNCH_END("psalm_swin_window_gather");
}

// ---------------------------------------------------------------- Swin: window reverse + un-shift + crop + residual
// swin_trans.py:235-250:  out[b,y,x,:] = shortcut[b,y,x,:] + win[row(b,y,x),:],  roll(+s): x[y] = shifted[(y-s) mod Hp]
template <typename TW, typename TX>
__global__ void __launch_bounds__(256) swin_window_merge_kernel(const TW* __restrict__ win, const TX* __restrict__ shortcut,
                                                                TX* __restrict__ out, int B, int H, int W, int C, int ws,
                                                                int shift) {
    const int lane = threadIdx.x & 63;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws, N = ws * ws;
    const int Hp = nWh * ws, Wp = nWw * ws;
    const long rows = (long)B * H * W;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int xx = (int)(r % W);
    const int y = (int)((r / W) % H);
    const int b = (int)(r / ((long)W * H));
    const int yp = (y - shift + Hp) % Hp, xp = (xx - shift + Wp) % Wp;
    const long wr = (((long)b * nWh + yp / ws) * nWw + xp / ws) * N + (yp % ws) * ws + (xp % ws);
    const TW* w = win + wr * C;
    const TX* s = shortcut + r * C;
    TX* o = out + r * C;
    for (int c = lane; c < C; c += 64) stf(o + c, ldf(s + c) + ldf(w + c));
}

extern "C" int psalm_swin_window_merge(const void* win, int win_dtype, const void* shortcut, void* out, int x_dtype, int B,
                                       int H, int W, int C, int ws, int shift, void* stream) {
    const long rows = (long)B * H * W;
    if (rows == 0) return 0;
    PSALM_DISPATCH(win_dtype, TW, PSALM_DISPATCH(x_dtype, TX, {
        hipLaunchKernelGGL((swin_window_merge_kernel<TW, TX>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const TW*)win, (const TX*)shortcut, (TX*)out, B, H, W, C, ws, shift);
    }));
    PSALM_LAUNCH_END("psalm_swin_window_merge");
}

// ... the same fused with norm2 (swin_trans.py:250-251): x_new = shortcut + merged is written once (fp32 stream) and its
// LayerNorm (the MLP's input) comes out of the same registers.  C % 8 == 0, C <= 2048.
// LPR (r06): lanes per row.  64: one row per wavefront, up to four 8-column chunks per lane (C <= 2048).  16 / 32: a row of C <= 128 / 256 columns on
// a quarter / half of the wavefront, 4 / 2 rows per wavefront -- the stage-1 / stage-2 rows of Swin (65536 x 128, 16384 x 256) left 48 / 32 lanes of every
// wavefront idle; same lane <-> column map inside a row and the same reduction order (seg_sum), so the same bits.
template <typename TW, typename TH, int LPR = 64>
__global__ void __launch_bounds__(256) swin_window_merge_ln_kernel(const TW* __restrict__ win, const float* __restrict__ shortcut,
                                                                   float* __restrict__ out_x, TH* __restrict__ out_h,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   int B, int H, int W, int C, int ws, int shift, float eps,
                                                                   unsigned short* __restrict__ h_split, float* __restrict__ h_inv, int Kp) {
    constexpr int RPW = 64 / LPR, NI = LPR == 64 ? 4 : 1;
    const int lane = (threadIdx.x & 63) % LPR;                    // lane inside the row's group
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws, N = ws * ws;
    const int Hp = nWh * ws, Wp = nWw * ws;
    const long rows = (long)B * H * W;
    const long r = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (threadIdx.x & 63) / LPR;
    if (r >= rows) return;                                        // (a whole group: shuffles stay inside groups)
    const int xx = (int)(r % W);
    const int y = (int)((r / W) % H);
    const int b = (int)(r / ((long)W * H));
    const int yp = (y - shift + Hp) % Hp, xp = (xx - shift + Wp) % Wp;
    const long wr = (((long)b * nWh + yp / ws) * nWw + xp / ws) * N + (yp % ws) * ws + (xp % ws);
    float v[NI][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < C) {
            float a[8];
            ld8(shortcut + r * C + c, v[i]);
            ld8(win + wr * C + c, a);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[i][k] += a[k]; sum += v[i][k]; }
            st8(out_x + r * C + c, v[i]);
        }
    }
    const float mean = seg_sum<LPR>(sum) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if ((i * LPR + lane) * 8 < C) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(seg_sum<LPR>(q) / C + eps);
    if (h_split) {                                               // f16x3: norm2's result leaves as the fc1 GEMM's split-f16 A operand
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * LPR + lane) * 8;
            if (c < C) {
                float g8[8], b8[8];
                ld8(gamma + c, g8);
                ld8(beta + c, b8);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[i][k] = (v[i][k] - mean) * rstd * g8[k] + b8[k]; amax = fmaxf(amax, fabsf(v[i][k])); }
            }
        }
        float sc, inv;
        split_scale(seg_max<LPR>(amax), sc, inv);
        if (lane == 0) h_inv[r] = inv;
        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * LPR + lane) * 8;
            if (c < Kp) emit_split8(c < C ? v[i] : zero8, sc, h_split + r * 2L * Kp + c, h_split + r * 2L * Kp + Kp + c);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < C) {
            float g8[8], b8[8], o8[8];
            ld8(gamma + c, g8);
            ld8(beta + c, b8);
#pragma unroll
            for (int k = 0; k < 8; ++k) o8[k] = (v[i][k] - mean) * rstd * g8[k] + b8[k];
            st8(out_h + r * C + c, o8);
        }
    }
}

extern "C" int psalm_swin_window_merge_ln(const void* win, int win_dtype, const float* shortcut, float* out_x, void* out_h, int h_dtype,
                                          const float* gamma, const float* beta, int B, int H, int W, int C, int ws, int shift,
                                          float eps, void* stream) {
    const long rows = (long)B * H * W;
    if (rows == 0) return 0;
    PSALM_CHECK_ARG(C % 8 == 0 && C <= 2048, "psalm_swin_window_merge_ln: C % 8 == 0 and C <= 2048");
    PSALM_DISPATCH(win_dtype, TW, PSALM_DISPATCH(h_dtype, TH, {
        hipLaunchKernelGGL((swin_window_merge_ln_kernel<TW, TH>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const TW*)win, shortcut, out_x, (TH*)out_h, gamma, beta, B, H, W, C, ws, shift, eps, nullptr, nullptr, 0);
    }));
    PSALM_LAUNCH_END("psalm_swin_window_merge_ln");
}

// f16x3 forms of the two Swin LayerNorm-fused data-movement kernels: the normalised rows leave as the split-f16 A operand of the GEMM
// they feed (qkv after norm1 + window partition; fc1 after window reverse + residual + norm2) -- see psalm_layernorm_split.
//   psalm_swin_window_merge_ln_split: win / shortcut / out_x fp32 as psalm_swin_window_merge_ln; h_split (B*H*W, 2*ceil64(C)) f16, h_inv.
//   psalm_swin_window_gather_split  : x fp32 (B,H,W,C) -> split (B*nW*ws*ws, 2*ceil64(C)) f16 + inv; padded tokens are zero rows.
extern "C" int psalm_swin_window_merge_ln_split(const float* win, const float* shortcut, float* out_x, void* h_split, float* h_inv,
                                                const float* gamma, const float* beta, int B, int H, int W, int C, int ws, int shift,
                                                float eps, void* stream) {
    const long rows = (long)B * H * W;
    if (rows == 0) return 0;
    PSALM_CHECK_ARG(C % 8 == 0 && C <= 2048 && h_split && h_inv, "psalm_swin_window_merge_ln_split: C % 8 == 0, C <= 2048, outputs required");
#define MLN_LAUNCH(LPR_) hipLaunchKernelGGL((swin_window_merge_ln_kernel<float, float, LPR_>), dim3(cdiv(rows, 4 * (64 / LPR_))), dim3(256), 0, (hipStream_t)stream, \
                                            win, shortcut, out_x, (float*)nullptr, gamma, beta, B, H, W, C, ws, shift, eps, (unsigned short*)h_split, h_inv, (C + 63) / 64 * 64)
    const int lpr = !psalm_get_tuning(PSALM_TUNE_ROW_GROUPS) ? 64 : (C <= 128 ? 16 : (C <= 256 ? 32 : 64));
    if (lpr == 16) MLN_LAUNCH(16); else if (lpr == 32) MLN_LAUNCH(32); else MLN_LAUNCH(64);
#undef MLN_LAUNCH
    PSALM_LAUNCH_END("psalm_swin_window_merge_ln_split");
}

template <int LPR>                                            // lanes per row: see swin_window_merge_ln_kernel
__global__ void __launch_bounds__(256) swin_window_gather_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, float* __restrict__ inv_out,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int B, int H,
                                                                       int W, int C, int ws, int shift, float eps, int Kp) {
    constexpr int RPW = 64 / LPR, NI = LPR == 64 ? 4 : 1;
    const int lane = (threadIdx.x & 63) % LPR;
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws, N = ws * ws;
    const int Hp = nWh * ws, Wp = nWw * ws;
    const long rows = (long)B * nWh * nWw * N;
    const long r = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (threadIdx.x & 63) / LPR;
    if (r >= rows) return;
    const int tok = (int)(r % N);
    long t = r / N;
    const int ww = (int)(t % nWw);
    t /= nWw;
    const int wh = (int)(t % nWh);
    const int b = (int)(t / nWh);
    const int y = (wh * ws + tok / ws + shift) % Hp, xx = (ww * ws + tok % ws + shift) % Wp;
    const bool live = y < H && xx < W;                            // (uniform over the row's lanes)
    float v[NI][8];
    float sc = 1.f, inv = 1.f;
    if (live) {
        const float* xr = x + (((long)b * H + y) * W + xx) * C;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * LPR + lane) * 8;
            if (c < C) {
                ld8(xr + c, v[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[i][k];
            }
        }
        const float mean = seg_sum<LPR>(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if ((i * LPR + lane) * 8 < C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(seg_sum<LPR>(q) / C + eps);
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = (i * LPR + lane) * 8;
            if (c < C) {
                float g8[8], b8[8];
                ld8(gamma + c, g8);
                ld8(beta + c, b8);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[i][k] = (v[i][k] - mean) * rstd * g8[k] + b8[k]; amax = fmaxf(amax, fabsf(v[i][k])); }
            }
        }
        split_scale(seg_max<LPR>(amax), sc, inv);
    }
    if (lane == 0) inv_out[r] = inv;
    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = (i * LPR + lane) * 8;
        if (c < Kp) emit_split8((live && c < C) ? v[i] : zero8, sc, out + r * 2L * Kp + c, out + r * 2L * Kp + Kp + c);
    }
}
extern "C" int psalm_swin_window_gather_split(const float* x, void* out, float* inv_out, const float* gamma, const float* beta, int B, int H,
                                              int W, int C, int ws, int shift, float eps, void* stream) {
    const long rows = (long)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws) * ws * ws;
    if (rows == 0) return 0;
    PSALM_CHECK_ARG(C % 8 == 0 && C <= 2048 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "psalm_swin_window_gather_split: C % 8 == 0, C <= 2048, aligned");
#define WGS_LAUNCH(LPR_) hipLaunchKernelGGL((swin_window_gather_split_kernel<LPR_>), dim3(cdiv(rows, 4 * (64 / LPR_))), dim3(256), 0, (hipStream_t)stream, x, \
                                            (unsigned short*)out, inv_out, gamma, beta, B, H, W, C, ws, shift, eps, (C + 63) / 64 * 64)
    const int lpr = !psalm_get_tuning(PSALM_TUNE_ROW_GROUPS) ? 64 : (C <= 128 ? 16 : (C <= 256 ? 32 : 64));
    if (lpr == 16) WGS_LAUNCH(16); else if (lpr == 32) WGS_LAUNCH(32); else WGS_LAUNCH(64);
#undef WGS_LAUNCH
    PSALM_LAUNCH_END("psalm_swin_window_gather_split");
}

// ---------------------------------------------------------------- Swin PatchMerging: 2x2 gather-concat + LN(4C)
// swin_trans.py:269-296: channel blocks [x(0::2,0::2), x(1::2,0::2), x(0::2,1::2), x(1::2,1::2)], odd H/W zero-padded.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) patch_merge_ln_kernel(const TI* __restrict__ x, TO* __restrict__ out,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int B, int H, int W, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long rows = (long)B * H2 * W2;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int x2 = (int)(r % W2), y2 = (int)((r / W2) % H2), b = (int)(r / ((long)W2 * H2));
    const int C4 = 4 * C;
    auto get = [&](int c4) -> float {
        const int blk = c4 / C, c = c4 % C;
        const int y = 2 * y2 + (blk & 1), xx = 2 * x2 + (blk >> 1);
        return (y < H && xx < W) ? ldf(x + (((long)b * H + y) * W + xx) * C + c) : 0.f;
    };
    float s = 0.f;
    for (int c = lane; c < C4; c += 64) s += get(c);
    const float mean = wave_sum(s) / C4;
    float v = 0.f;
    // (fused multiply-adds written out -- the form hipcc compiled this loop to since r01 -- so that the register-resident form below and the host build
    //  round alike)
    for (int c = lane; c < C4; c += 64) { const float d = get(c) - mean; v = fmaf(d, d, v); }
    const float rstd = rsqrtf(wave_sum(v) / C4 + eps);
    TO* o = out + r * C4;
    for (int c = lane; c < C4; c += 64) stf(o + c, fmaf((get(c) - mean) * rstd, gamma[c], beta[c]));
}

// fp32 -> fp32 form for 4C = 64 NJ (r06): the row (<= 2048 values) is read ONCE into registers -- the kernel above re-reads it for the variance and for
// the output and pays an integer division per element for the (quadrant, channel) of a column: 35 us for the 16384 x 512 rows of Swin stage 1 -> 2 (1.7 TB/s).
// Lane <-> column map (column lane + 64 j), summation order and expressions are the kernel above's: same words.
template <int NJ>
__global__ void __launch_bounds__(256) patch_merge_ln_regs_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int B, int H, int W, float eps) {
    constexpr int C4 = 64 * NJ, C = C4 / 4;
    const int lane = threadIdx.x & 63;
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long rows = (long)B * H2 * W2;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int x2 = (int)(r % W2), y2 = (int)((r / W2) % H2), b = (int)(r / ((long)W2 * H2));
    float val[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int blk = (64 * j) / C, c = (64 * j) % C + lane;           // (compile-time quadrant: C % 64 == 0)
        const int y = 2 * y2 + (blk & 1), xx = 2 * x2 + (blk >> 1);
        val[j] = (y < H && xx < W) ? x[(((long)b * H + y) * W + xx) * C + c] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += val[j];
    const float mean = wave_sum(s) / C4;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const float d = val[j] - mean; v = fmaf(d, d, v); }
    const float rstd = rsqrtf(wave_sum(v) / C4 + eps);
    float* o = out + r * C4;
    // (the fused multiply-adds are WRITTEN OUT, as the generic kernel compiles them: left to the compiler, the unrolled NJ = 16 form squared 4 of its 16
    //  differences with a packed multiply + add -- two roundings -- and the output moved by an ulp on the MI355X; the host build cannot see this)
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[lane + 64 * j] = fmaf((val[j] - mean) * rstd, gamma[lane + 64 * j], beta[lane + 64 * j]);
}

extern "C" int psalm_patch_merge_ln(const void* x, int x_dtype, void* out, int out_dtype, const float* gamma,
                                    const float* beta, int B, int H, int W, int C, float eps, void* stream) {
    const long rows = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
    if (rows == 0) return 0;
    if (x_dtype == PSALM_F32 && out_dtype == PSALM_F32 && (C == 128 || C == 256 || C == 512) && psalm_get_tuning(PSALM_TUNE_ROW_GROUPS)) {
#define PML_LAUNCH(NJ_) hipLaunchKernelGGL((patch_merge_ln_regs_kernel<NJ_>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, \
                                           gamma, beta, B, H, W, eps)
        if (C == 128) PML_LAUNCH(8); else if (C == 256) PML_LAUNCH(16); else PML_LAUNCH(32);
#undef PML_LAUNCH
        PSALM_LAUNCH_END("psalm_patch_merge_ln");
    }
    PSALM_DISPATCH(x_dtype, TI, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((patch_merge_ln_kernel<TI, TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const TI*)x, (TO*)out, gamma, beta, B, H, W, C, eps);
    }));
    PSALM_LAUNCH_END("psalm_patch_merge_ln");
}

// ---------------------------------------------------------------- GroupNorm on NHWC  (msdeformattn.py:199-202,248-254)
// x (B, HW, C), G groups of C/G channels; statistics over (HW x C/G) per (b,g).  Deterministic two-stage:
//   stage 1: each block reduces a chunk of rows for all channels -> partial (sum, sumsq) per (b, chunk, g)
//   stage 2: the apply kernel first folds the partials of its (b,g) (few hundred values), then normalises.
template <typename TI>
__global__ void __launch_bounds__(256) groupnorm_stats_kernel(const TI* __restrict__ x, float* __restrict__ partial, int HW,
                                                              int C, int G, int rows_per_chunk, int nchunks) {
    __shared__ float ssum[256], ssq[256];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    // thread owns channel (tid % C) when C <= 256 (C is 256 in PSALM); general C handled by looping
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + tid;
        float s = 0.f, q = 0.f;
        if (c < C)
            for (int r = r0; r < r1; ++r) { const float v = ldf(x + ((long)b * HW + r) * C + c); s += v; q += v * v; }
        ssum[tid] = s;
        ssq[tid] = q;
        __syncthreads();
        // fold the cpg channels of each group (tid is the first channel of a group)
        if (c < C && (c % cpg) == 0) {
            float gs = 0.f, gq = 0.f;
            for (int i = 0; i < cpg; ++i) { gs += ssum[tid + i]; gq += ssq[tid + i]; }
            float* p = partial + (((long)b * nchunks + chunk) * G + c / cpg) * 2;
            p[0] = gs;
            p[1] = gq;
        }
        __syncthreads();
    }
}

// stage 2: one wavefront per (b, g) folds the chunk partials (double accumulation, fixed order -> deterministic)
__global__ void __launch_bounds__(64) groupnorm_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                                int HW, int C, int G, int nchunks, float eps) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
        const float* p = partial + (((long)b * nchunks + k) * G + g) * 2;
        s += p[0];
        q += p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane == 0) {
        const double n = (double)HW * (C / G);
        const double m = s / n;
        double var = q / n - m * m;
        if (var < 0) var = 0;
        stats[((long)b * G + g) * 2] = (float)m;
        stats[((long)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// stage 3: normalise, 8 consecutive channels per thread (16-byte bf16 / 2 x 16-byte fp32 accesses)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) groupnorm_apply_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int HW, int C, int G, int relu) {
    const int b = blockIdx.y;
    const int cpg = C / G;
    const long total8 = (long)HW * C / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        const int c0 = (int)(e % C);
        float v[8];
        ld8(x + (long)b * HW * C + e, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k, g = c / cpg;
            const float m = stats[((long)b * G + g) * 2], r = stats[((long)b * G + g) * 2 + 1];
            float o = (v[k] - m) * r * gamma[c] + beta[c];
            if (relu) o = fmaxf(o, 0.f);
            v[k] = o;
        }
        st8(y + (long)b * HW * C + e, v);
    }
}

// workspace: at least B * (nchunks + 1) * G * 2 floats with nchunks = ceil(HW / 64)
extern "C" int psalm_groupnorm_nhwc(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                                    float* workspace, int B, int HW, int C, int G, float eps, int relu, void* stream) {
    if (B == 0 || HW == 0) return 0;
    PSALM_CHECK_ARG(G <= 256 && C % G == 0 && C % 8 == 0, "psalm_groupnorm_nhwc: need G <= 256, C % G == 0, C % 8 == 0");
    PSALM_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "psalm_groupnorm_nhwc: 16-byte aligned tensors");
    const int rows_per_chunk = 64;
    const int nchunks = cdiv(HW, rows_per_chunk);
    float* stats = workspace + (long)B * nchunks * G * 2;
    hipStream_t s = (hipStream_t)stream;
    PSALM_DISPATCH(x_dtype, TI, {
        hipLaunchKernelGGL((groupnorm_stats_kernel<TI>), dim3(nchunks, B), dim3(256), 0, s, (const TI*)x, workspace, HW, C, G,
                           rows_per_chunk, nchunks);
    });
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(G, B), dim3(64), 0, s, workspace, stats, HW, C, G, nchunks, eps);
    long gx = ((long)HW * C / 8 + 255) / 256;
    if (gx > 8192) gx = 8192;
    PSALM_DISPATCH(x_dtype, TI, PSALM_DISPATCH(y_dtype, TO, {
        hipLaunchKernelGGL((groupnorm_apply_kernel<TI, TO>), dim3((unsigned)(gx < 1 ? 1 : gx), B), dim3(256), 0, s, (const TI*)x,
                           (TO*)y, stats, gamma, beta, HW, C, G, relu);
    }));
    PSALM_LAUNCH_END("psalm_groupnorm_nhwc");
}

// ---------------------------------------------------------------- out[r,:] = a[r,:] + b[r % b_rows,:]
template <typename TA, typename TB, typename TO>
__global__ void __launch_bounds__(256) add_bcast_kernel(const TA* __restrict__ a, const TB* __restrict__ b, TO* __restrict__ out,
                                                        long rows, int C, long b_rows) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i % C);
        stf(out + i, ldf(a + i) + ldf(b + (r % b_rows) * C + c));
    }
}

extern "C" int psalm_add_bcast(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, long rows,
                               int C, long b_rows, void* stream) {
    if (rows == 0) return 0;
    const long total = rows * C;
    const int grid = (int)((total + 2047) / 2048);
    PSALM_DISPATCH(a_dtype, TA, PSALM_DISPATCH(b_dtype, TB, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((add_bcast_kernel<TA, TB, TO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const TA*)a,
                           (const TB*)b, (TO*)out, rows, C, b_rows);
    })));
    PSALM_LAUNCH_END("psalm_add_bcast");
}

// ---------------------------------------------------------------- row gather from up to 4 source tables
// dst[r,:] = src[src_id[r]][src_row[r],:]   (token splicing, llava_phi.py:581-766: embedding rows, image tokens,
// seg queries, region features);  src_id < 0 -> zeros (batch right-padding, llava_phi.py:874-884)
struct GatherSrc { const void* p[4]; int dtype[4]; };
template <typename TO>
__global__ void __launch_bounds__(256) gather_rows_kernel(GatherSrc src, const int* __restrict__ src_id,
                                                          const int* __restrict__ src_row, TO* __restrict__ dst, long rows,
                                                          int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int id = src_id[r];
    TO* o = dst + r * C;
    if (id < 0) {
        for (int c = lane; c < C; c += 64) stf(o + c, 0.f);
    } else if (src.dtype[id] == PSALM_F32) {
        const float* s = (const float*)src.p[id] + (long)src_row[r] * C;
        for (int c = lane; c < C; c += 64) stf(o + c, s[c]);
    } else {
        const bf16_t* s = (const bf16_t*)src.p[id] + (long)src_row[r] * C;
        for (int c = lane; c < C; c += 64) stf(o + c, bf16_to_f32(s[c]));
    }
}

extern "C" int psalm_gather_rows(const void* src0, int dt0, const void* src1, int dt1, const void* src2, int dt2,
                                 const void* src3, int dt3, const int* src_id, const int* src_row, void* dst, int dst_dtype,
                                 long rows, int C, void* stream) {
    if (rows == 0) return 0;
    GatherSrc s;
    s.p[0] = src0; s.p[1] = src1; s.p[2] = src2; s.p[3] = src3;
    s.dtype[0] = dt0; s.dtype[1] = dt1; s.dtype[2] = dt2; s.dtype[3] = dt3;
    PSALM_DISPATCH(dst_dtype, TO, {
        hipLaunchKernelGGL((gather_rows_kernel<TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, s, src_id,
                           src_row, (TO*)dst, rows, C);
    });
    PSALM_LAUNCH_END("psalm_gather_rows");
}

// ---------------------------------------------------------------- segment mean over listed rows
// out[s,:] = mean_{i in [off[s], off[s+1])} x[rows[i],:]   -- class-name / [SEG] / region pooling of LLM states
// (llava_phi.py:552-565 AdaptiveAvgPool1d(1), :972-978, :1299-1316, :302-307)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) segment_mean_kernel(const TI* __restrict__ x, long ldx, const int* __restrict__ off,
                                                           const int* __restrict__ rows, TO* __restrict__ out, int nseg,
                                                           int C) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int i0 = off[s], i1 = off[s + 1];
    const float inv = i1 > i0 ? 1.f / (i1 - i0) : 0.f;
    // r06: fp32 rows of C % 256 == 0 columns (the LLM states: 2048) take every column of a row with four 16-byte loads per lane, all in flight, and
    // fetch the row index ONCE per row -- the loop below re-read rows[i] in front of every 4-byte load of every column step (a chain of
    // 2 * C / 64 * (i1 - i0) dependent global loads per wave: 38 us per launch for a 100-segment call, profiles/r05_bench_breakdown.json).  Same sums
    // in the same order: acc_c = ((0 + x_i0,c) + x_i0+1,c) + ..., then * inv.
    if constexpr (std::is_same<TI, float>::value) {
        if (C % 256 == 0 && C <= 4096 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0) {
            float acc[16][4];
            const int nv = C / 256;                              // 16-byte vectors per lane (<= 16)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v][0] = acc[v][1] = acc[v][2] = acc[v][3] = 0.f;
            for (int i = i0; i < i1; ++i) {
                const float* xr = x + (long)rows[i] * ldx + lane * 4;
                psalm_f32x4 t[16];
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (v < nv) t[v] = *reinterpret_cast<const psalm_f32x4*>(xr + v * 256);
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (v < nv) { acc[v][0] += t[v].x; acc[v][1] += t[v].y; acc[v][2] += t[v].z; acc[v][3] += t[v].w; }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v)
                if (v < nv) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) stf(out + (long)s * C + v * 256 + lane * 4 + k, acc[v][k] * inv);
                }
            return;
        }
    }
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        for (int i = i0; i < i1; ++i) acc += ldf(x + (long)rows[i] * ldx + c);
        stf(out + (long)s * C + c, acc * inv);
    }
}

extern "C" int psalm_segment_mean(const void* x, int x_dtype, long ldx, const int* seg_offsets, const int* seg_rows, void* out,
                                  int out_dtype, int nseg, int C, void* stream) {
    if (nseg == 0) return 0;
    PSALM_DISPATCH(x_dtype, TI, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((segment_mean_kernel<TI, TO>), dim3(cdiv(nseg, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const TI*)x, ldx, seg_offsets, seg_rows, (TO*)out, nseg, C);
    }));
    PSALM_LAUNCH_END("psalm_segment_mean");
}

