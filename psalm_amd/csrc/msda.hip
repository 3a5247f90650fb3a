// Multi-scale deformable attention forward (bilinear sample-and-accumulate) for gfx950.
//
// Replaces the reference's CUDA op `ms_deformable_im2col_gpu_kernel`
// (psalm/model/mask_decoder/Mask2Former_Simplify/modeling/pixel_decoder/ops/src/cuda/
//  ms_deform_im2col_cuda.cuh:242-304, bilinear helper :38-89) -- same arithmetic:
//     out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,p])
//     h_im = loc_y*H_l - 0.5, w_im = loc_x*W_l - 0.5; a sample contributes only when
//     -1 < h_im < H_l and -1 < w_im < W_l; out-of-range corners read as 0.
//
// This is an HBM/L2 gather, not a GEMM: no MFMA.  Mapping for CDNA4:
//   * one thread owns VEC=4 consecutive channels of one (b,q,head): a (q,head) pair is D/4 = 8 lanes
//     that read one 128-byte (fp32) / 64-byte (bf16) contiguous segment per bilinear corner, so a
//     64-lane wavefront issues 8 fully-used 128 B segments per corner load (16 B per lane);
//   * consecutive threads walk (head, q) in the order the value/out tensors are laid out, so the
//     output store is a dense 16 B/lane stream and neighbouring queries (neighbouring pixels) hit
//     the same L2 lines of `value`; the whole value tensor (22 MB fp32 at 1024^2) is L2/MALL resident;
//   * corner fetches are unconditional (clamped addresses, validity in the weights), so the loads of a sampling point -- and, in the
//     8-channel kernel, of two points -- are in flight together; the rest of the latency is hidden by occupancy (4-8 waves/SIMD).
//   * FUSED variant (used by the pixel decoder): takes the raw outputs of the fused
//     [sampling_offsets | attention_weights] projection and computes the softmax over L*P and the
//     sampling locations in-kernel (ops/modules/ms_deform_attn.py:101-110), so the (B,Lq,M,L,P,2)
//     location and (B,Lq,M,L,P) weight tensors are never materialised in HBM.
#include "common.h"
#include <atomic>

struct alignas(16) f32x4_s { float x, y, z, w; };
struct alignas(8) bf16x4_s { bf16_t x, y, z, w; };

__device__ __forceinline__ f32x4_s ld4(const float* p) { return *reinterpret_cast<const f32x4_s*>(p); }
__device__ __forceinline__ f32x4_s ld4(const bf16_t* p) {
    bf16x4_s v = *reinterpret_cast<const bf16x4_s*>(p);
    return f32x4_s{bf16_to_f32(v.x), bf16_to_f32(v.y), bf16_to_f32(v.z), bf16_to_f32(v.w)};
}
__device__ __forceinline__ void st4(float* p, f32x4_s v) { *reinterpret_cast<f32x4_s*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4_s v) {
    *reinterpret_cast<bf16x4_s*>(p) = bf16x4_s{f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
}

template <int V> struct msda_ic { static constexpr int value = V; };
#define MSDA_MAX_LEVELS 8
struct MsdaLevels {
    int H[MSDA_MAX_LEVELS], W[MSDA_MAX_LEVELS], start[MSDA_MAX_LEVELS];
};

// Branch-free: corner addresses clamped into the level (always loadable), validity of the sample / of each corner folded into the
// corner weight (an invalid corner contributes an exact 0, as in the reference) -- a guarded `valid ? ld4(p) : 0` compiles to a branch
// and an s_waitcnt vmcnt(0) per corner, which chains all corner fetches of a lane one after the other.
template <typename TV>
__device__ __forceinline__ void msda_sample(const TV* __restrict__ vbase, int Hl, int Wl, int row_stride, float loc_x,
                                            float loc_y, float wgt, f32x4_s& acc) {
    const float h_im = loc_y * Hl - 0.5f;
    const float w_im = loc_x * Wl - 0.5f;
    const bool inb = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
    const int h_low = (int)floorf(inb ? h_im : 0.f), w_low = (int)floorf(inb ? w_im : 0.f);
    const float lh = h_im - h_low, lw = w_im - w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool h0 = h_low >= 0, h1 = h_low + 1 <= Hl - 1, w0 = w_low >= 0, w1 = w_low + 1 <= Wl - 1;
    const int ha = max(h_low, 0), hb = min(h_low + 1, Hl - 1), wa = max(w_low, 0), wb = min(w_low + 1, Wl - 1);
    const f32x4_s v1 = ld4(vbase + ((long)ha * Wl + wa) * row_stride);
    const f32x4_s v2 = ld4(vbase + ((long)ha * Wl + wb) * row_stride);
    const f32x4_s v3 = ld4(vbase + ((long)hb * Wl + wa) * row_stride);
    const f32x4_s v4 = ld4(vbase + ((long)hb * Wl + wb) * row_stride);
    const float w1c = (inb && h0 && w0) ? hh * hw : 0.f, w2c = (inb && h0 && w1) ? hh * lw : 0.f;
    const float w3c = (inb && h1 && w0) ? lh * hw : 0.f, w4c = (inb && h1 && w1) ? lh * lw : 0.f;
    const float wg = inb ? wgt : 0.f;
    acc.x += wg * (w1c * v1.x + w2c * v2.x + w3c * v3.x + w4c * v4.x);
    acc.y += wg * (w1c * v1.y + w2c * v2.y + w3c * v3.y + w4c * v4.y);
    acc.z += wg * (w1c * v1.z + w2c * v2.z + w3c * v3.z + w4c * v4.z);
    acc.w += wg * (w1c * v1.w + w2c * v2.w + w3c * v3.w + w4c * v4.w);
}

// 8 channels per lane (16-byte bf16 / 2 x 16-byte fp32 corner reads): half the load instructions and half the redundant
// per-(query, head) softmax / location arithmetic of the 4-channel mapping.  Used by the fused kernel when D % 8 == 0.
// Branch-free form: the four corner addresses are clamped into the level (always loadable) and validity (sample inside the padded
// image, corner inside the image) is folded into the corner weights -- the arithmetic on valid corners is the expression of the
// reference kernel (ms_deform_im2col_bilinear, ms_deform_im2col_cuda.cuh:38-89), invalid corners contribute an exact 0.  With the
// `if (corner valid) load` form every one of the 48 corner fetches of a (query, head) compiled to its own branch + s_waitcnt vmcnt(0):
// a chain of 48 dependent L2 round trips per lane (r01 ISA audit), which is what bounded the kernel, not bandwidth.
template <typename TV>
struct MsdaTaps8 {
    const TV* p[4];
    float w[4];
};
template <typename TV>
__device__ __forceinline__ MsdaTaps8<TV> msda_taps8(const TV* __restrict__ vbase, int Hl, int Wl, int row_stride, float loc_x, float loc_y) {
    const float h_im = loc_y * Hl - 0.5f;
    const float w_im = loc_x * Wl - 0.5f;
    const bool inb = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
    const int h_low = (int)floorf(inb ? h_im : 0.f), w_low = (int)floorf(inb ? w_im : 0.f);
    const float lh = h_im - h_low, lw = w_im - w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool h0 = h_low >= 0, h1 = h_low + 1 <= Hl - 1, w0 = w_low >= 0, w1 = w_low + 1 <= Wl - 1;
    const int ha = max(h_low, 0), hb = min(h_low + 1, Hl - 1), wa = max(w_low, 0), wb = min(w_low + 1, Wl - 1);
    MsdaTaps8<TV> t;
    t.p[0] = vbase + ((long)ha * Wl + wa) * row_stride;
    t.p[1] = vbase + ((long)ha * Wl + wb) * row_stride;
    t.p[2] = vbase + ((long)hb * Wl + wa) * row_stride;
    t.p[3] = vbase + ((long)hb * Wl + wb) * row_stride;
    t.w[0] = (inb && h0 && w0) ? hh * hw : 0.f;
    t.w[1] = (inb && h0 && w1) ? hh * lw : 0.f;
    t.w[2] = (inb && h1 && w0) ? lh * hw : 0.f;
    t.w[3] = (inb && h1 && w1) ? lh * lw : 0.f;
    return t;
}

// XCD_BANDS = true (B == 1, every level height a multiple of 8, block count a multiple of 8): the hardware places block b on XCD b % 8;
// XCD k is given the queries of the k-th horizontal BAND (rows [k H_l / 8, (k+1) H_l / 8) of every level), in level order.  A query
// samples every level around its own normalised position, so the corner rows an XCD touches are its band of each level (+ a halo):
// 1/8 of the 22 MB fp32 value tensor = 2.75 MB, resident in that XCD's 4 MB L2 -- with the linear order every XCD streamed the whole
// tensor through its L2 (r01 PMC: 1.9x the compulsory bytes from the fabric).
// QUAD = true (D == 32: the 4 lanes of a (query, head) are one DPP quad): lane g owns sampling point p = g of every level -- its softmax
// term, sampling location, clamped corner offsets and the four corner weights (times the attention weight) are computed ONCE and handed to
// the other three lanes with quad broadcasts, instead of every lane computing all L * P samples for itself.  r04e: the LDS-staged form of this
// kernel (tools/experiments/msda_lds_staged.hip) removed the L2 fetches and got SLOWER (108 vs 57 us): what the kernel spends its time on is
// this per-sample arithmetic, ~60 instructions per sample, 4 x redundant -- for bf16 value (64-byte corners); with fp32 value the L2 request rate
// is the bound and this form is slower (see the launch code): the host selects it for bf16 only.
template <typename TV, typename TO, int L, int P, bool XCD_BANDS = false, bool QUAD = false>
__global__ void __launch_bounds__(256) msda_fused8_kernel(const TV* __restrict__ value, MsdaLevels lv,
                                                          const float* __restrict__ ow, TO* __restrict__ out, int B, int S,
                                                          int M, int D) {
    const int G = D >> 3;
    const int Lq = S;
    const long total = (long)B * Lq * M * G;
    constexpr int LP = L * P;
    for (long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x; idx0 < total; idx0 += (long)gridDim.x * blockDim.x) {
        long idx = idx0;
        if constexpr (XCD_BANDS)                                     // (grid covers `total` exactly once: no grid-stride wrap)
            idx = ((long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x);   // position inside the band's (query, head, group) list
        const int g = (int)(idx % G);
        long t = idx / G;
        const int m = (int)(t % M);
        t /= M;
        int q = (int)(t % Lq);
        const int b = XCD_BANDS ? 0 : (int)(t / Lq);
        if constexpr (XCD_BANDS) {                                   // t = index inside the band: level by cumulative band sizes
            const int band = blockIdx.x & 7;
            int r = (int)t, lq_ = 0, base = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int nb = lv.H[l] * lv.W[l] / 8;
                if (l == lq_ && r >= nb && l + 1 < L) { r -= nb; ++lq_; }
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (l == lq_) base = lv.start[l] + band * (lv.H[l] * lv.W[l] / 8);
            q = base + r;
        }
        int lq = 0;
#pragma unroll
        for (int l = 1; l < L; ++l)
            if (q >= lv.start[l]) lq = l;
        const int qi = q - lv.start[lq];
        const float ref_x = ((qi % lv.W[lq]) + 0.5f) / lv.W[lq];
        const float ref_y = ((qi / lv.W[lq]) + 0.5f) / lv.H[lq];
        const float* row = ow + ((long)b * Lq + q) * (M * LP * 3);
        const float* offp = row + m * LP * 2;
        const float* lgp = row + M * LP * 2 + m * LP;
        const int row_stride = M * D;
        const TV* vb = value + (long)b * S * row_stride + m * D + g * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        if constexpr (QUAD) {
            static_assert(P == 4, "one sampling point per lane of the quad");
            // ---- this lane's share: point p = g of every level
            float lgq[L];
            float mx = -3.4e38f;
#pragma unroll
            for (int l = 0; l < L; ++l) { lgq[l] = lgp[l * P + g]; mx = fmaxf(mx, lgq[l]); }
            mx = fmaxf(mx, __builtin_bit_cast(float, psalm_swap_adjacent(__builtin_bit_cast(unsigned, mx))));
            mx = fmaxf(mx, __builtin_bit_cast(float, psalm_quad_swap2(__builtin_bit_cast(unsigned, mx))));
            float den = 0.f;
#pragma unroll
            for (int l = 0; l < L; ++l) { lgq[l] = __expf(lgq[l] - mx); den += lgq[l]; }
            den += __builtin_bit_cast(float, psalm_swap_adjacent(__builtin_bit_cast(unsigned, den)));
            den += __builtin_bit_cast(float, psalm_quad_swap2(__builtin_bit_cast(unsigned, den)));
            const float inv = 1.f / den;
            // corner BYTE offsets from `value` (this lane's channel offset added at the fetch): fetched through a buffer descriptor -- one address
            // register per corner (with 64-bit pointers the compiler materialised all 48 addresses up front: 214 VGPRs and serialized fetches)
            const psalm_rsrc vr = psalm_make_rsrc(value, (unsigned)((long)B * S * row_stride * (long)sizeof(TV)));
            const unsigned lane_b = (unsigned)(((long)b * S * row_stride + m * D + g * 8) * (long)sizeof(TV));
            auto fetch = [&](unsigned off, float* v8) {            // 8 channels at byte offset `off` of this lane's slice
                if constexpr (sizeof(TV) == 4) {
                    const psalm_u32x4 a = psalm_buf_load_b128(vr, lane_b + off), b2 = psalm_buf_load_b128(vr, lane_b + off + 16u);
                    v8[0] = __builtin_bit_cast(float, a.x); v8[1] = __builtin_bit_cast(float, a.y); v8[2] = __builtin_bit_cast(float, a.z);
                    v8[3] = __builtin_bit_cast(float, a.w); v8[4] = __builtin_bit_cast(float, b2.x); v8[5] = __builtin_bit_cast(float, b2.y);
                    v8[6] = __builtin_bit_cast(float, b2.z); v8[7] = __builtin_bit_cast(float, b2.w);
                } else {
                    const psalm_u32x4 a = psalm_buf_load_b128(vr, lane_b + off);
                    const unsigned w4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v8[2 * e] = __builtin_bit_cast(float, w4[e] << 16);
                        v8[2 * e + 1] = __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                    }
                }
            };
#pragma unroll 1                                                   // (unrolled, the compiler hoists the fetches of all 12 samples: 256 VGPRs + AGPRs, 1 wave per SIMD)
            for (int l = 0; l < L; ++l) {
                // level table: constant kernarg offsets + select on l (see the note in the generic loop below)
                int Hl = lv.H[0], Wl = lv.W[0], sl = lv.start[0];
                float lgl = lgq[0];
#pragma unroll
                for (int j = 1; j < L; ++j)
                    if (l == j) { Hl = lv.H[j]; Wl = lv.W[j]; sl = lv.start[j]; lgl = lgq[j]; }
                // ---- this lane's sample of the level: point p = g
                const int i = l * P + g;
                const float lx = ref_x + offp[2 * i] / Wl, ly = ref_y + offp[2 * i + 1] / Hl;
                const MsdaTaps8<TV> tp = msda_taps8<TV>(vb + (long)sl * row_stride, Hl, Wl, row_stride, lx, ly);
                const float wgt = lgl * inv;
                unsigned toff[4];                                 // corner byte offsets from `value` + this (b, head)'s base, without the lane's channels
                float tw[4];                                      // corner weight x attention weight
#pragma unroll
                for (int c = 0; c < 4; ++c) { toff[c] = (unsigned)((tp.p[c] - vb) * (long)sizeof(TV)); tw[c] = tp.w[c] * wgt; }
                // ---- all four samples of the level for this lane's 8 channels, taps from the owning lane; two samples' fetches in flight
                auto pair = [&](auto PC0) {
                    constexpr int p0 = decltype(PC0)::value;
                    float w[2][4], v[2][4][8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        fetch(psalm_quad_bcast<p0>(toff[c]), v[0][c]);
                        fetch(psalm_quad_bcast<p0 + 1>(toff[c]), v[1][c]);
                        w[0][c] = __builtin_bit_cast(float, psalm_quad_bcast<p0>(__builtin_bit_cast(unsigned, tw[c])));
                        w[1][c] = __builtin_bit_cast(float, psalm_quad_bcast<p0 + 1>(__builtin_bit_cast(unsigned, tw[c])));
                    }
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            acc[k] += w[pp][0] * v[pp][0][k] + w[pp][1] * v[pp][1][k] + w[pp][2] * v[pp][2][k] + w[pp][3] * v[pp][3][k];
                };
                pair(msda_ic<0>{});
                pair(msda_ic<2>{});
            }
            static_assert(L <= 3, "levels of the quad form");
            st8(out + (((long)b * Lq + q) * M + m) * D + g * 8, acc);
            continue;
        }
        float lg[LP];
        float mx = -3.4e38f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { lg[i] = lgp[i]; mx = fmaxf(mx, lg[i]); }
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { lg[i] = __expf(lg[i] - mx); den += lg[i]; }
        const float inv = 1.f / den;
        constexpr int PB = (P % 2 == 0) ? 2 : 1;                  // sampling points whose 4 * PB corner fetches are in flight together
#pragma unroll 1                                                   // (unrolled, the tap set-up of all L*P samples is hoisted: 256 VGPRs, 1 wave/SIMD)
        for (int l = 0; l < L; ++l) {
            // The level table is read with CONSTANT kernarg offsets and selected on `l`: indexing the by-value struct with the
            // rolled loop counter compiled (ROCm 7.2, gfx950) to `s_load_dword sX, s[base+3+4l], 0x1d` -- an unaligned SGPR base
            // plus an unaligned immediate; the scalar memory unit drops the low two bits of each separately, so W[l] / start[l]
            // came back from the wrong slot (garbage geometry -> wild corner addresses -> GPU memory fault, r01 round end).
            int Hl = lv.H[0], Wl = lv.W[0], sl = lv.start[0];
#pragma unroll
            for (int j = 1; j < L; ++j)
                if (l == j) { Hl = lv.H[j]; Wl = lv.W[j]; sl = lv.start[j]; }
            const TV* vl = vb + (long)sl * row_stride;
#pragma unroll
            for (int p0 = 0; p0 < P; p0 += PB) {
                MsdaTaps8<TV> tp[PB];
                float v[PB][4][8];
#pragma unroll
                for (int pp = 0; pp < PB; ++pp) {
                    const int i = l * P + p0 + pp;
                    const float lx = ref_x + offp[2 * i] / Wl;
                    const float ly = ref_y + offp[2 * i + 1] / Hl;
                    tp[pp] = msda_taps8<TV>(vl, Hl, Wl, row_stride, lx, ly);
#pragma unroll
                    for (int c = 0; c < 4; ++c) ld8(tp[pp].p[c], v[pp][c]);
                }
                __builtin_amdgcn_sched_barrier(0);               // all 4 PB corner fetches issued before the first use waits on one
#pragma unroll
                for (int pp = 0; pp < PB; ++pp) {
                    const float wgt = lg[l * P + p0 + pp] * inv;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        acc[k] += wgt * (tp[pp].w[0] * v[pp][0][k] + tp[pp].w[1] * v[pp][1][k] + tp[pp].w[2] * v[pp][2][k] +
                                         tp[pp].w[3] * v[pp][3][k]);
                }
            }
        }
        st8(out + (((long)b * Lq + q) * M + m) * D + g * 8, acc);
    }
}

// Plugin form: explicit sampling locations / attention weights (the reference op's contract).
// DEV = true: the level table is read from DEVICE memory (`shapes_dev` (L,2) / `starts_dev` (L) int64 -- exactly the tensors the
// reference op receives, ms_deform_attn_cuda.cu:64-75), so the caller needs no host copy of them (no D2H sync on the seam).
template <typename TV, typename TO, int LP_UNROLL, bool DEV = false>
__global__ void __launch_bounds__(256) msda_forward_kernel(const TV* __restrict__ value, MsdaLevels lv,
                                                           const float* __restrict__ loc, const float* __restrict__ attw,
                                                           TO* __restrict__ out, int B, int S, int M, int D, int L, int Lq,
                                                           int P, const long* __restrict__ shapes_dev = nullptr,
                                                           const long* __restrict__ starts_dev = nullptr) {
    const int G = D >> 2;  // lanes per (q,head)
    const long total = (long)B * Lq * M * G;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        long t = idx / G;
        const int m = (int)(t % M);
        t /= M;
        const int q = (int)(t % Lq);
        const int b = (int)(t / Lq);
        const long qm = ((long)b * Lq + q) * M + m;
        const float* lp = loc + qm * L * P * 2;
        const float* wp = attw + qm * L * P;
        const int row_stride = M * D;
        const TV* vb = value + (long)b * S * row_stride + m * D + g * 4;
        f32x4_s acc{0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) {
            const int Hl = DEV ? (int)shapes_dev[2 * l] : lv.H[l], Wl = DEV ? (int)shapes_dev[2 * l + 1] : lv.W[l];
            const long sl = DEV ? starts_dev[l] : (long)lv.start[l];
            // device-side level table: never gather outside `value` on a table that does not describe it (a level that does not fit
            // inside the S rows contributes nothing; the host wrapper reports such a table the first time it sees it)
            if (DEV && !(Hl > 0 && Wl > 0 && sl >= 0 && sl + (long)Hl * Wl <= (long)S)) continue;
            const TV* vl = vb + sl * row_stride;
#pragma unroll LP_UNROLL
            for (int p = 0; p < P; ++p) {
                const int i = l * P + p;
                msda_sample<TV>(vl, Hl, Wl, row_stride, lp[2 * i], lp[2 * i + 1], wp[i], acc);
            }
        }
        st4(out + qm * D + g * 4, acc);
    }
}

// Fused form: raw projection output `ow` (B*Lq, M*L*P*3) = [offsets (M,L,P,2) | logits (M,L*P)],
// query q is pixel (i,j) of level lq, reference point ((j+.5)/W, (i+.5)/H) (msdeformattn.py:76-87, valid_ratio 1).
template <typename TV, typename TO, int L, int P>
__global__ void __launch_bounds__(256) msda_fused_kernel(const TV* __restrict__ value, MsdaLevels lv,
                                                         const float* __restrict__ ow, TO* __restrict__ out, int B, int S,
                                                         int M, int D) {
    const int G = D >> 2;
    const int Lq = S;
    const long total = (long)B * Lq * M * G;
    constexpr int LP = L * P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        long t = idx / G;
        const int m = (int)(t % M);
        t /= M;
        const int q = (int)(t % Lq);
        const int b = (int)(t / Lq);
        int lq = 0;
#pragma unroll
        for (int l = 1; l < L; ++l)
            if (q >= lv.start[l]) lq = l;
        const int qi = q - lv.start[lq];
        const float ref_x = ((qi % lv.W[lq]) + 0.5f) / lv.W[lq];
        const float ref_y = ((qi / lv.W[lq]) + 0.5f) / lv.H[lq];
        const float* row = ow + ((long)b * Lq + q) * (M * LP * 3);
        const float* offp = row + m * LP * 2;
        const float* lgp = row + M * LP * 2 + m * LP;
        float lg[LP];
        float mx = -3.4e38f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { lg[i] = lgp[i]; mx = fmaxf(mx, lg[i]); }
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { lg[i] = __expf(lg[i] - mx); den += lg[i]; }
        const float inv = 1.f / den;
        const int row_stride = M * D;
        const TV* vb = value + (long)b * S * row_stride + m * D + g * 4;
        f32x4_s acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const TV* vl = vb + (long)lv.start[l] * row_stride;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int i = l * P + p;
                const float lx = ref_x + offp[2 * i] / lv.W[l];
                const float ly = ref_y + offp[2 * i + 1] / lv.H[l];
                msda_sample<TV>(vl, lv.H[l], lv.W[l], row_stride, lx, ly, lg[i] * inv, acc);
            }
        }
        st4(out + (((long)b * Lq + q) * M + m) * D + g * 4, acc);
    }
}

// A/B knob of the r04 measurement pass (tools/bench_msda.py): 1 = quad-shared taps (default), 0 = every lane computes all samples (r01 - r03)
static std::atomic<int> g_msda_quad{1};
extern "C" int psalm_msda_set_policy(int v) {
    if (v != 0 && v != 1) { psalm_set_error("psalm_msda_set_policy: 0 / 1 (quad-shared taps off / on)"); return -1; }
    g_msda_quad = v;
    return 0;
}

static int fill_levels(MsdaLevels& lv, const int64_t* shapes, const int64_t* starts, int L, int S) {
    if (L > MSDA_MAX_LEVELS) return -1;
    long tot = 0;
    for (int l = 0; l < L; ++l) {
        lv.H[l] = (int)shapes[2 * l];
        lv.W[l] = (int)shapes[2 * l + 1];
        lv.start[l] = (int)starts[l];
        tot += (long)lv.H[l] * lv.W[l];
    }
    return tot == S ? 0 : -2;
}

extern "C" int psalm_msda_forward(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                                  const int64_t* level_start_host, const float* sampling_loc, const float* attn_weight,
                                  void* out, int out_dtype, int B, int S, int M, int D, int L, int Lq, int P, void* stream) {
    PSALM_CHECK_ARG(D % 4 == 0 && D > 0, "psalm_msda_forward: head dim must be a multiple of 4");
    MsdaLevels lv = {};
    int rc = fill_levels(lv, spatial_shapes_host, level_start_host, L, S);
    PSALM_CHECK_ARG(rc != -1, "psalm_msda_forward: too many levels (max 8)");
    PSALM_CHECK_ARG(rc == 0, "psalm_msda_forward: sum(H_l*W_l) != S");
    const long total = (long)B * Lq * M * (D / 4);
    if (total == 0) return 0;
    const int block = 256;
    const int grid = (int)((total + block - 1) / block < 65536 * 8 ? (total + block - 1) / block : 65536 * 8);
    PSALM_DISPATCH(value_dtype, TV, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((msda_forward_kernel<TV, TO, 4>), dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           (const TV*)value, lv, sampling_loc, attn_weight, (TO*)out, B, S, M, D, L, Lq, P);
    }));
    PSALM_LAUNCH_END("psalm_msda_forward");
}

// Same op with the level table on the DEVICE (the reference's own calling convention: spatial_shapes / level_start_index are CUDA
// int64 tensors, ms_deform_attn.h:25-44): asynchronous, no host copy.  sum(H_l*W_l) == S is the caller's contract; the kernel skips a level
// whose rows [start, start + H*W) do not lie inside the S rows of `value` (no out-of-bounds gather), and hip_ops.msda_forward_dev validates
// each distinct table once.
extern "C" int psalm_msda_forward_dev(const void* value, int value_dtype, const int64_t* spatial_shapes_dev,
                                      const int64_t* level_start_dev, const float* sampling_loc, const float* attn_weight, void* out,
                                      int out_dtype, int B, int S, int M, int D, int L, int Lq, int P, void* stream) {
    PSALM_CHECK_ARG(D % 4 == 0 && D > 0, "psalm_msda_forward_dev: head dim must be a multiple of 4");
    PSALM_CHECK_ARG(L >= 1 && spatial_shapes_dev && level_start_dev, "psalm_msda_forward_dev: level table missing");
    MsdaLevels lv = {};
    const long total = (long)B * Lq * M * (D / 4);
    if (total == 0) return 0;
    const int block = 256;
    const int grid = (int)((total + block - 1) / block < 65536 * 8 ? (total + block - 1) / block : 65536 * 8);
    PSALM_DISPATCH(value_dtype, TV, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((msda_forward_kernel<TV, TO, 4, true>), dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           (const TV*)value, lv, sampling_loc, attn_weight, (TO*)out, B, S, M, D, L, Lq, P,
                           (const long*)spatial_shapes_dev, (const long*)level_start_dev);
    }));
    PSALM_LAUNCH_END("psalm_msda_forward_dev");
}

extern "C" int psalm_msda_fused(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                                const int64_t* level_start_host, const float* offsets_logits, void* out, int out_dtype,
                                int B, int S, int M, int D, int L, int P, void* stream) {
    PSALM_CHECK_ARG(D % 4 == 0 && D > 0, "psalm_msda_fused: head dim must be a multiple of 4");
    PSALM_CHECK_ARG(L == 3 && P == 4, "psalm_msda_fused: specialised for L=3 levels, P=4 points (PSALM pixel decoder)");
    MsdaLevels lv = {};
    int rc = fill_levels(lv, spatial_shapes_host, level_start_host, L, S);
    PSALM_CHECK_ARG(rc == 0, "psalm_msda_fused: bad level table");
    const int block = 256;
    if (D % 8 == 0 && (uintptr_t)value % 16 == 0 && (uintptr_t)out % 16 == 0) {       // 8 channels per lane
        const long total8 = (long)B * S * M * (D / 8);
        if (total8 == 0) return 0;
        bool bands = B == 1 && total8 % (block * 8) == 0;                         // XCD-band order (see the kernel comment)
        for (int l = 0; l < L; ++l) bands = bands && lv.H[l] % 8 == 0;
        // D == 32: the 4 channel-group lanes of a (query, head) are a DPP quad -> taps computed once per group (32-bit byte offsets).  For bf16
        // value only: r04f on MI355X (profiles/r04f_bench_msda_quad_taps.jsonl) bf16 52.6 -> 34.9 us, but fp32 57.3 -> 66.4 us -- with 128-byte
        // corners the fp32 kernel is bound by the L2 request rate (1.05 GB per launch, ~18 TB/s), not by the tap arithmetic, and the rolled level
        // loop keeps fewer fetches in flight.
        const bool quad = g_msda_quad && D == 32 && value_dtype == PSALM_BF16 && (long)B * S * M * D * 2 < (1L << 31);
#define MSDA_LAUNCH8(BANDS_, QUAD_, GRID_)                                                                                              \
        PSALM_DISPATCH(value_dtype, TV, PSALM_DISPATCH(out_dtype, TO, {                                                                 \
            hipLaunchKernelGGL((msda_fused8_kernel<TV, TO, 3, 4, BANDS_, QUAD_>), dim3((unsigned)(GRID_)), dim3(block), 0,              \
                               (hipStream_t)stream, (const TV*)value, lv, offsets_logits, (TO*)out, B, S, M, D);                        \
        }))
        if (bands) {
            if (quad) MSDA_LAUNCH8(true, true, total8 / block); else MSDA_LAUNCH8(true, false, total8 / block);
            PSALM_LAUNCH_END("psalm_msda_fused");
        }
        if (quad) MSDA_LAUNCH8(false, true, (total8 + block - 1) / block); else MSDA_LAUNCH8(false, false, (total8 + block - 1) / block);
#undef MSDA_LAUNCH8
        PSALM_LAUNCH_END("psalm_msda_fused");
    }
    const long total = (long)B * S * M * (D / 4);
    if (total == 0) return 0;
    const int grid = (int)((total + block - 1) / block);
    PSALM_DISPATCH(value_dtype, TV, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((msda_fused_kernel<TV, TO, 3, 4>), dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           (const TV*)value, lv, offsets_logits, (TO*)out, B, S, M, D);
    }));
    PSALM_LAUNCH_END("psalm_msda_fused");
}
