// Post-processing of the predictor outputs into the evaluator-facing results
// (llava_phi.py:1401-1466: semantic LP:402-406, instance LP:407-447, panoptic LP:325-386, referring LP:308-324,
//  region LP:387-400).  HBM-bound streaming kernels over the (Q, H, W) full-resolution mask logits plus a few
// single-block kernels for the small sequential pieces (top-k, segment merging).
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---------------------------------------------------------------- class softmax: probs, transposed/padded probs, max score + label
// cls (Q, C1) fp32 logits (C1 = classes + void).  One wave per query.
// probs (Q, C1); probsT (C1-1, Kpad) zero-padded beyond Q (A operand of the semantic GEMM, void column dropped,
// LP:403); score[q] = max_c softmax, label[q] = argmax_c (first index on ties, like torch.max).
template <typename TP>
__global__ void __launch_bounds__(256) class_softmax_kernel(const float* __restrict__ cls, float* __restrict__ probs,
                                                            TP* __restrict__ probsT, float* __restrict__ score,
                                                            int* __restrict__ label, int Q, int C1, int Kpad) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const float* row = cls + (long)q * C1;
    float mx = -3.0e38f;
    int mi = 0x7fffffff;
    for (int c = lane; c < C1; c += 64) {
        const float v = row[c];
        if (v > mx) { mx = v; mi = c; }
    }
    const float gmx = wave_max(mx);
    int cand = (mx == gmx) ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
    float s = 0.f;
    for (int c = lane; c < C1; c += 64) s += __expf(row[c] - gmx);
    const float inv = 1.f / wave_sum(s);
    for (int c = lane; c < C1; c += 64) {
        const float p = __expf(row[c] - gmx) * inv;
        probs[(long)q * C1 + c] = p;
        if (c < C1 - 1) stf(probsT + (long)c * Kpad + q, p);
    }
    if (lane == 0) { score[q] = inv; label[q] = cand; }     // exp(0) * inv
}

extern "C" int psalm_class_softmax(const float* cls, float* probs, void* probsT, int probsT_dtype, float* score, int* label, int Q,
                                   int C1, int Kpad, void* stream) {
    if (Q == 0) return 0;
    PSALM_DISPATCH(probsT_dtype, TP, {
        hipLaunchKernelGGL((class_softmax_kernel<TP>), dim3(cdiv(Q, 4)), dim3(256), 0, (hipStream_t)stream, cls, probs, (TP*)probsT,
                           score, label, Q, C1, Kpad);
    });
    PSALM_LAUNCH_END("psalm_class_softmax");
}

// ---------------------------------------------------------------- sigT[p, q] = sigmoid(mask[q, p]), zero-padded to Kpad columns
// (W operand of the semantic GEMM: sem[c, p] = sum_q probs[q, c] * sigmoid(mask[q, p]), LP:402-406)
template <typename TO>
__global__ void __launch_bounds__(256) sigmoid_transpose_kernel(const float* __restrict__ mask, TO* __restrict__ out, int Q, long HW,
                                                                int Kpad) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int q0 = 0; q0 < Kpad; q0 += 64) {
        for (int r = ty; r < 64; r += 4) {
            const int q = q0 + r;
            const long p = p0 + tx;
            tile[r][tx] = (q < Q && p < HW) ? sigmoidf_(mask[(long)q * HW + p]) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            const long p = p0 + r;
            const int q = q0 + tx;
            if (p < HW && q < Kpad) stf(out + p * Kpad + q, tile[tx][r]);
        }
        __syncthreads();
    }
}

extern "C" int psalm_sigmoid_transpose(const float* mask, void* out, int out_dtype, int Q, long HW, int Kpad, void* stream) {
    if (HW == 0) return 0;
    PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((sigmoid_transpose_kernel<TO>), dim3((unsigned)((HW + 63) / 64)), dim3(256), 0, (hipStream_t)stream, mask,
                           (TO*)out, Q, HW, Kpad);
    });
    PSALM_LAUNCH_END("psalm_sigmoid_transpose");
}

// ---------------------------------------------------------------- per-query mask score  (LP:443-444)
// score[q] = sum(sigmoid(m) * [m>0]) / (sum([m>0]) + 1e-6).  Deterministic two-stage reduction.
__global__ void __launch_bounds__(256) mask_score_partial_kernel(const float* __restrict__ mask, float* __restrict__ partial, long HW,
                                                                 int nchunks) {
    __shared__ float sn[4], sd[4];
    const int q = blockIdx.y, chunk = blockIdx.x;
    const long per = (HW + nchunks - 1) / nchunks;
    const long p0 = chunk * per, p1 = min(HW, p0 + per);
    float num = 0.f, den = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const float m = mask[(long)q * HW + p];
        if (m > 0.f) { num += sigmoidf_(m); den += 1.f; }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { sn[threadIdx.x >> 6] = num; sd[threadIdx.x >> 6] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((long)q * nchunks + chunk) * 2 + 0] = sn[0] + sn[1] + sn[2] + sn[3];
        partial[((long)q * nchunks + chunk) * 2 + 1] = sd[0] + sd[1] + sd[2] + sd[3];
    }
}
// one wavefront per query: lane l sums the partials l, l + 64, ... in double, the 64 lane sums meet in a shuffle tree (r04: ONE thread per
// query walked its 512 partials of the fused semantic pass in a chain of dependent loads -- 54 us for 100 KB; the sums are of <= 512 floats
// of like magnitude in double, whose grouping does not reach the float result)
__global__ void __launch_bounds__(64) mask_score_final_kernel(const float* __restrict__ partial, float* __restrict__ score, int Q, int nchunks) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= Q) return;
    double num = 0.0, den = 0.0;
    for (int k = lane; k < nchunks; k += 64) { num += partial[((long)q * nchunks + k) * 2]; den += partial[((long)q * nchunks + k) * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long nb = __builtin_bit_cast(long long, num), db = __builtin_bit_cast(long long, den);
        const int nlo = __shfl_xor((int)(nb & 0xffffffffll), o), nhi = __shfl_xor((int)(nb >> 32), o);
        const int dlo = __shfl_xor((int)(db & 0xffffffffll), o), dhi = __shfl_xor((int)(db >> 32), o);
        num += __builtin_bit_cast(double, ((long long)nhi << 32) | (long long)(unsigned)nlo);
        den += __builtin_bit_cast(double, ((long long)dhi << 32) | (long long)(unsigned)dlo);
    }
    if (lane == 0) score[q] = (float)(num / (den + 1e-6));
}

// workspace: Q * 64 * 2 floats
extern "C" int psalm_mask_scores(const float* mask, float* score, float* workspace, int Q, long HW, void* stream) {
    if (Q == 0) return 0;
    const int nchunks = 64;
    hipLaunchKernelGGL(mask_score_partial_kernel, dim3(nchunks, Q), dim3(256), 0, (hipStream_t)stream, mask, workspace, HW, nchunks);
    hipLaunchKernelGGL(mask_score_final_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, workspace, score, Q, nchunks);
    PSALM_LAUNCH_END("psalm_mask_scores");
}

// ---------------------------------------------------------------- fused semantic inference (bf16 mode)
// sem[c, p] = sum_q probs[q, c] * sigmoid(mask[q, p])   (class_name_semantic_inference, LP:402-406) in ONE pass over the
// (Q, HW) fp32 mask logits: a persistent block walks 128-pixel tiles; per tile the logits are read coalesced, squashed, rounded
// to bf16 and laid down pixel-major in LDS (= the MFMA B operand, k = q contiguous); probsT (C, Kpad) sits in LDS for the whole
// kernel (A operand); each wave owns 32 pixels x all classes (5 x 32-row MFMA tiles, K = Kpad = 128) and writes whole 128-byte
// row segments of the (C, HW) fp32 result.  HBM traffic = the compulsory read of the masks + write of the result; replaces the
// sigmoid-transpose pass (419 MB read + 268 MB write) + the K=128 GEMM over 24576 tiny tiles.
typedef float pp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));

template <int NI>                                           // queries per wave: q = wave + 4 i, i < NI (NI = ceil(Q / 4) rounded to 25 / 32)
__global__ void __launch_bounds__(256, 2) semantic_from_masks_kernel(const float* __restrict__ mask, const bf16_t* __restrict__ probsT,
                                                                     float* __restrict__ out, float* __restrict__ partial, int Q, int C,
                                                                     long HW, int ntiles) {
    constexpr int KP = 128, PITCH = KP + 4;                  // 264-byte rows: conflict-light 2-byte scatter writes, 8-byte aligned reads
    constexpr int CT = 5;                                    // class tiles of 32 (C <= 160)
    __shared__ __attribute__((aligned(16))) bf16_t Ps[CT * 32 * PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Ss[128 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row bases of the logit loads stay in SGPRs
    for (int e = tid; e < CT * 32 * (KP / 8); e += 256) {    // probsT -> LDS, zero rows beyond C
        const int c = e / (KP / 8), k8 = (e % (KP / 8)) * 8;
        psalm_u32x4 v{0, 0, 0, 0};
        if (c < C) v = *reinterpret_cast<const psalm_u32x4*>(probsT + (long)c * KP + k8);
        *reinterpret_cast<unsigned long long*>(&Ps[c * PITCH + k8]) = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
        *reinterpret_cast<unsigned long long*>(&Ps[c * PITCH + k8 + 4]) = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    }
    for (int e = tid; e < 128 * (KP - 4 * NI); e += 256) {   // q columns the staging loop never writes: zero once
        const int p = e / (KP - 4 * NI), q = 4 * NI + e % (KP - 4 * NI);
        Ss[p * PITCH + q] = f32_to_bf16(0.f);
    }
    float num[NI], den[NI];                                  // per-lane mask-score sums of this wave's queries
#pragma unroll
    for (int i = 0; i < NI; ++i) { num[i] = 0.f; den[i] = 0.f; }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long p0 = (long)t * 128;
        __syncthreads();                                     // previous tile's MFMA reads are done (and the fills above are visible)
        // branch-free staging: rows beyond Q re-read row Q-1 (their probsT columns are zero, so the finite values they leave in LDS
        // vanish in the product; their score sums are never stored); pixels beyond HW re-read pixel HW-1 and are masked to 0
        const long pa = min(p0 + lane, HW - 1), pb = min(p0 + 64 + lane, HW - 1);
        const bool va = p0 + lane < HW, vb = p0 + 64 + lane < HW;
        constexpr int G = NI % 5 == 0 ? NI : 8;              // queries whose logits are fetched together (2 G loads in flight per lane)
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += G) {
            float ma[G], mb[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {                    // unconditional (clamped) loads, masked afterwards: no branches
                const float* row = mask + (long)min(wave + 4 * (i0 + g), Q - 1) * HW;
                ma[g] = row[pa];
                mb[g] = row[pb];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i = i0 + g, q = wave + 4 * i;
                const float xa = va ? ma[g] : 0.f, xb = vb ? mb[g] : 0.f;
                const float sa = sigmoidf_(xa), sb = sigmoidf_(xb);
                Ss[lane * PITCH + q] = f32_to_bf16(sa);
                Ss[(64 + lane) * PITCH + q] = f32_to_bf16(sb);
                num[i] += (xa > 0.f ? sa : 0.f) + (xb > 0.f ? sb : 0.f);
                den[i] += (xa > 0.f ? 1.f : 0.f) + (xb > 0.f ? 1.f : 0.f);
            }
        }
        __syncthreads();
        pp_f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        int arow = n * PITCH;                                // probsT fragments are re-read from LDS per tile (160 VGPRs if kept)
        PSALM_OPAQUE_VGPR(arow);
#pragma unroll
        for (int kk = 0; kk < KP / 16; ++kk) {
            const int ko = 16 * kk + 8 * hi;
            const unsigned long long* bp = reinterpret_cast<const unsigned long long*>(&Ss[(32 * wave + n) * PITCH + ko]);
            const unsigned long long b0 = bp[0], b1 = bp[1];
            const pp_bf16x8 bfrag = __builtin_bit_cast(pp_bf16x8, psalm_u32x4{(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)});
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const unsigned long long* ap = reinterpret_cast<const unsigned long long*>(&Ps[32 * ct * PITCH + arow + ko]);
                const unsigned long long a0 = ap[0], a1 = ap[1];
                const pp_bf16x8 afrag = __builtin_bit_cast(pp_bf16x8, psalm_u32x4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)});
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[ct], 0, 0, 0);
            }
        }
        // stores as (scalar row base) + (32-bit lane offset): lane offset = pixel + 4*hi rows, row base = out + cb * HW + p0
        const bool pv = p0 + 32 * wave + n < HW;
        const unsigned voff = (unsigned)(32 * wave + n) + (unsigned)(4 * hi) * (unsigned)HW;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cb = 32 * ct + (r & 3) + 8 * (r >> 2);
                float* rowp = out + (long)cb * HW + p0;
                if (pv && cb + 4 * hi < C) rowp[voff] = acc[ct][r];
            }
    }
    if (partial) {                                           // (q, block, 2) partial sums -> mask_score_final_kernel (fixed order)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + 4 * i;
            const float a = wave_sum(num[i]), b = wave_sum(den[i]);
            if (lane == 0 && q < Q) {
                partial[((long)q * gridDim.x + blockIdx.x) * 2 + 0] = a;
                partial[((long)q * gridDim.x + blockIdx.x) * 2 + 1] = b;
            }
        }
    }
}

// ---- split-f16 ("X3", fp32-class) form of the fused semantic pass, for precision = "f16x3": the same single pass over the logits, but
// both operands are carried as f16 hi + lo (probabilities: fixed scale 2^13; sigmoids: a power-of-two scale per PIXEL from its largest
// logit) and each (class tile, k-step) is three f16 MFMAs (hi.hi + lo.hi + hi.lo), fp32 accumulate, scales undone at the store.  Replaces, in that mode,
// sigmoid_transpose (419 MB read + 512 MB write) + split (1 GB) + the K = 384 GEMM + the separate mask-score pass (r02e: 1.0 ms) by the
// compulsory read + write.  LDS: 2 x (160 + 128) rows x 264 B = 152 KB, one persistent block per CU.
typedef _Float16 pp_f16x8 __attribute__((ext_vector_type(8)));
template <int NI>
__global__ void __launch_bounds__(256, 1) semantic_from_masks_x3_kernel(const float* __restrict__ mask, const float* __restrict__ probsT,
                                                                        float* __restrict__ out, float* __restrict__ partial, int Q, int C,
                                                                        long HW, int ntiles) {
    constexpr int KP = 128, PITCH = KP + 4, CT = 5;
    constexpr float SC = 8192.0f;
    static_assert(NI % 5 == 0 || NI == 32, "all logits of a wave's queries are held in registers for the per-pixel scale");
    __shared__ __attribute__((aligned(16))) unsigned short Ph[CT * 32 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Pl[CT * 32 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Sh[128 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Sl[128 * PITCH];
    __shared__ float pmax[4][128];                           // per-wave maximum logit of each pixel of the tile
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto split = [&](float a, unsigned short& h, unsigned short& l) {     // a already scaled into f16 range
        const _Float16 hh = (_Float16)a;
        const _Float16 ll = (_Float16)(a - (float)hh);
        h = __builtin_bit_cast(unsigned short, hh);
        l = __builtin_bit_cast(unsigned short, ll);
    };
    for (int e = tid; e < CT * 32 * KP; e += 256) {          // probsT (C, 128) f32 -> hi / lo in LDS (probabilities <= 1: scale 2^13), zero rows beyond C
        const int c = e / KP, k = e % KP;
        unsigned short h = 0, l = 0;
        if (c < C) split(probsT[(long)c * KP + k] * SC, h, l);
        Ph[c * PITCH + k] = h;
        Pl[c * PITCH + k] = l;
    }
    for (int e = tid; e < 128 * (KP - 4 * NI); e += 256) {   // q columns the staging loop never writes: zero once
        const int p = e / (KP - 4 * NI), q = 4 * NI + e % (KP - 4 * NI);
        Sh[p * PITCH + q] = 0;
        Sl[p * PITCH + q] = 0;
    }
    float num[NI], den[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { num[i] = 0.f; den[i] = 0.f; }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long p0 = (long)t * 128;
        __syncthreads();                                     // previous tile's MFMA reads / pmax reads are done
        const long pa = min(p0 + lane, HW - 1), pb = min(p0 + 64 + lane, HW - 1);
        const bool va = p0 + lane < HW, vb = p0 + 64 + lane < HW;
        // all logits of this wave's NI queries for its two pixels (unconditional clamped loads, 2 NI in flight per lane)
        float ma[NI], mb[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float* row = mask + (long)min(wave + 4 * i, Q - 1) * HW;
            ma[i] = row[pa];
            mb[i] = row[pb];
        }
        // per-pixel scale: sigmoid is monotone, so the largest sigmoid of a pixel belongs to its largest logit.  A FIXED scale would
        // leave pixels whose masks are all strongly negative (most of a random-weight image) with subnormal f16 operands -- their class
        // scores differ only in the bits that loses (r02: 0.009 % label flips vs 0.0007 % with per-pixel scales).
        float xa_m = -3.0e38f, xb_m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (wave + 4 * i < Q) { xa_m = fmaxf(xa_m, ma[i]); xb_m = fmaxf(xb_m, mb[i]); }
        pmax[wave][lane] = xa_m;
        pmax[wave][64 + lane] = xb_m;
        __syncthreads();
        auto pix_scale = [&](int px, float& sc, float& inv) {  // power of two placing the pixel's largest sigmoid in [2^12, 2^14)
            const float mxl = fmaxf(fmaxf(pmax[0][px], pmax[1][px]), fmaxf(pmax[2][px], pmax[3][px]));
            const float smax = sigmoidf_(mxl);
            int e = (int)((__builtin_bit_cast(unsigned, smax) >> 23) & 0xffu) - 127;     // floor(log2 smax), <= 0
            int se = 13 - e;
            se = se > 100 ? 100 : se;
            const bool zero = !(smax > 0.f);
            sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
            inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
        };
        float sca, scb, ia, ib;
        pix_scale(lane, sca, ia);
        pix_scale(64 + lane, scb, ib);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + 4 * i;
            const float xa = va ? ma[i] : 0.f, xb = vb ? mb[i] : 0.f;
            const float sa = sigmoidf_(xa), sb = sigmoidf_(xb);
            unsigned short h, l;
            split(sa * sca, h, l);
            Sh[lane * PITCH + q] = h;
            Sl[lane * PITCH + q] = l;
            split(sb * scb, h, l);
            Sh[(64 + lane) * PITCH + q] = h;
            Sl[(64 + lane) * PITCH + q] = l;
            num[i] += (xa > 0.f ? sa : 0.f) + (xb > 0.f ? sb : 0.f);
            den[i] += (xa > 0.f ? 1.f : 0.f) + (xb > 0.f ? 1.f : 0.f);
        }
        __syncthreads();
        pp_f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        int arow = n * PITCH;
        PSALM_OPAQUE_VGPR(arow);
        auto frag = [&](const unsigned short* base) -> pp_f16x8 {
            const unsigned long long* p = reinterpret_cast<const unsigned long long*>(base);
            const unsigned long long a0 = p[0], a1 = p[1];
            return __builtin_bit_cast(pp_f16x8, psalm_u32x4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)});
        };
#pragma unroll
        for (int kk = 0; kk < KP / 16; ++kk) {
            const int ko = 16 * kk + 8 * hi;
            const pp_f16x8 bh = frag(&Sh[(32 * wave + n) * PITCH + ko]), bl = frag(&Sl[(32 * wave + n) * PITCH + ko]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const pp_f16x8 ah = frag(&Ph[32 * ct * PITCH + arow + ko]), al = frag(&Pl[32 * ct * PITCH + arow + ko]);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[ct], 0, 0, 0);
            }
        }
        float osc, oinv;                                     // this lane's OUTPUT pixel (32 wave + n): 1 / (2^13 * its scale)
        pix_scale(32 * wave + n, osc, oinv);
        oinv *= 1.0f / SC;
        const bool pv = p0 + 32 * wave + n < HW;
        const unsigned voff = (unsigned)(32 * wave + n) + (unsigned)(4 * hi) * (unsigned)HW;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cb = 32 * ct + (r & 3) + 8 * (r >> 2);
                float* rowp = out + (long)cb * HW + p0;
                if (pv && cb + 4 * hi < C) rowp[voff] = acc[ct][r] * oinv;
            }
    }
    if (partial) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + 4 * i;
            const float a = wave_sum(num[i]), b = wave_sum(den[i]);
            if (lane == 0 && q < Q) {
                partial[((long)q * gridDim.x + blockIdx.x) * 2 + 0] = a;
                partial[((long)q * gridDim.x + blockIdx.x) * 2 + 1] = b;
            }
        }
    }
}

// ---- the same pass as TWO wave groups per CU that run half a tile apart (r03: the one-group kernel above serialises load latency, the
// sigmoid / split VALU work, the matrix phase and the stores of a tile -- 15 us per 128-pixel tile, 2.0 TB/s).  One persistent block of 8
// waves per CU; group g = wave >> 2 owns every other 64-pixel tile of the block and walks   X | C | M1 | M2   with a block barrier after each
// slot, group 1 two slots behind group 0 -- so one group's VALU slots (X: wait for the logits, sigmoids, pixel maxima; C: per-pixel scale,
// split into f16 hi / lo, 8-byte LDS writes) run beside the other group's matrix slots (M1 / M2: 3 products per class tile and k-step; M2
// ends with the stores), on the same SIMDs.  The logits of a group's NEXT tile are requested at the end of X and consumed three slots
// later; the wait there is a counted vmcnt (the stores issued in between are unconditional buffer stores -- out-of-range ones carry an
// offset the descriptor drops -- so the count is static).  K is padded to KP = 112 when Q allows (7 instead of 8 k-steps; LDS 136 KB).
template <int KP>
__global__ void __launch_bounds__(512, 1) semantic_from_masks_x3_pair_kernel(const float* __restrict__ mask, const float* __restrict__ probsT,
                                                                             float* __restrict__ out, float* __restrict__ partial, int Q,
                                                                             int C, long HW, int ntiles, int contig) {
    constexpr int PITCH = KP + 4, CT = 5, NI = KP / 4, KS = KP / 16, KS1 = (KS + 1) / 2;
    constexpr float SC = 8192.0f;
#ifdef PSALM_SEM_ABL                                          // (tools/experiments/r06_semantic_ablate.py: what the pass costs without ...)
    constexpr int ABL = PSALM_SEM_ABL;                        // 1: the stores  2: the matrix instructions  4: the sigmoids  8: the loads  16: the split
#else
    constexpr int ABL = 0;
#endif
    __shared__ __attribute__((aligned(16))) unsigned short Ph[CT * 32 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Pl[CT * 32 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Sh[2][64 * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Sl[2][64 * PITCH];
    __shared__ float pmax[2][4][64];                          // per-wave maximum logit of each pixel of the group's tile
    __shared__ float oinv_s[2][64];                           // 1 / (2^13 * pixel scale) for the stores
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, w = wave & 3;
    for (int e = tid; e < CT * 32 * KP; e += 512) {           // probsT (C, 128) f32 -> hi / lo in LDS (probabilities <= 1: scale 2^13), zero rows beyond C
        const int c = e / KP, k = e % KP;
        unsigned short h = 0, l = 0;
        if (c < C) {
            const float a = probsT[(long)c * 128 + k] * SC;
            const _Float16 hh = (_Float16)a;
            h = __builtin_bit_cast(unsigned short, hh);
            l = __builtin_bit_cast(unsigned short, (_Float16)(a - (float)hh));
        }
        Ph[c * PITCH + k] = h;
        Pl[c * PITCH + k] = l;
    }
    const psalm_rsrc orsrc = psalm_make_rsrc(out, (unsigned)((unsigned long)C * (unsigned long)HW * 4ul));
    const psalm_rsrc mrsrc = psalm_make_rsrc(mask, (unsigned)((unsigned long)Q * (unsigned long)HW * 4ul));
    const unsigned row_bytes = (unsigned)HW * 4u;
    const int qb = w * NI;                                    // this wave's queries in the VALU slots: [qb, qb + NI)
    const int ps = w & 1, u0 = (w >> 1) ? 3 : 0, nu = (w >> 1) ? 2 : 3;   // matrix slots: pixel half ps, class tiles u0 .. u0 + nu - 1
    const int iters = (ntiles + 2 * (int)gridDim.x - 1) / (2 * (int)gridDim.x);
    float num[NI], den[NI], m[NI], sg[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { num[i] = 0.f; den[i] = 0.f; sg[i] = 0.f; }
    // r06: the wave's query slots beyond Q (Q = 100: 12 of wave 3's 28) are skipped by wave-uniform branches -- as per-query select masks they were 28
    // SGPR pairs; with the 28 load and 48 store row offsets (loop invariants the compiler hoists) the loop body carried ~560 v_readlane / v_writelane
    // instructions of SGPR spills per tile in a kernel that is bound by VALU issue.  The row offsets are now derived inside the loop from an
    // opaque copy of the wave's first query / class tile (a few dozen SALU instructions per tile).
    const int nvq = __builtin_amdgcn_readfirstlane(max(0, min(NI, Q - qb)));
    // tile order: strided (all CUs walk neighbouring 256-byte pieces of every row at the same time) or, contig != 0, one contiguous
    // pixel range per block (experiment: PSALM_SEM_ORDER=1)
    auto tile_of = [&](int it) { return contig ? (long)blockIdx.x * (2 * iters) + 2 * it + grp : (long)blockIdx.x + (long)gridDim.x * (2 * it + grp); };
    auto request = [&](int it) {                              // unconditional clamped loads: NI in flight per lane
        const long t = min(tile_of(it), (long)ntiles - 1);
        const unsigned voff = (unsigned)min(t * 64 + lane, HW - 1) * 4u;
        int qb_ = qb;
        PSALM_OPAQUE_SGPR(qb_);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (ABL & 8) m[i] = 0.001f * (float)(lane + i);
            else m[i] = psalm_buf_load_f32_s(mrsrc, voff, (unsigned)min(qb_ + i, Q - 1) * row_bytes);   // row offset: SGPR
        }
    };
    request(0);
    // 48 dropped stores behind the first request: the loop's wait for a tile's logits then sees the SAME instruction stream behind them on
    // the entry path as on the back edge (48 stores of the previous tile) and the compiler's counted vmcnt stays exact
#pragma unroll
    for (int r = 0; r < 48; ++r) psalm_buf_store_f32(0.f, orsrc, PSALM_BUF_OOB + 4u * r);      // (distinct addresses: not merged)
    __syncthreads();                                          // P staged
    if (grp == 1) { PSALM_RAW_BARRIER(); PSALM_RAW_BARRIER(); }
    for (int it = 0; it < iters; ++it) {
        const long t = tile_of(it);
        const bool live = t < ntiles;
        const long p0 = min(t, (long)ntiles - 1) * 64;
        const bool v = live && p0 + lane < HW;
        // ---- X: sigmoids, mask-score sums, this wave's maximum logit per pixel; then the next tile's logits are requested
        float xm = -3.0e38f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < nvq) {                                        // (wave-uniform; sg[i] of the other slots stays 0)
                const float x = v ? m[i] : 0.f;
                const float sv = (ABL & 4) ? 0.5f + 0.125f * x : sigmoidf_(x);
                sg[i] = sv;
                xm = fmaxf(xm, m[i]);
                const bool pos = x > 0.f;
                num[i] += pos ? sv : 0.f;
                den[i] += pos ? 1.f : 0.f;
            }
        }
        pmax[grp][w][lane] = xm;
        if (it + 1 < iters) request(it + 1);
        PSALM_RAW_BARRIER();
        // ---- C: power of two placing the pixel's largest sigmoid in [2^12, 2^14) (see the one-group kernel); split; LDS
        {
            const float mxl = fmaxf(fmaxf(pmax[grp][0][lane], pmax[grp][1][lane]), fmaxf(pmax[grp][2][lane], pmax[grp][3][lane]));
            const float smax = sigmoidf_(mxl);
            const int e = (int)((__builtin_bit_cast(unsigned, smax) >> 23) & 0xffu) - 127;
            int se = 13 - e;
            se = se > 100 ? 100 : se;
            const bool zero = !(smax > 0.f);
            const float sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
            const float inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
            if (w == 0) oinv_s[grp][lane] = inv * (1.0f / SC);
            unsigned short* sh = &Sh[grp][lane * PITCH + qb];
            unsigned short* sl = &Sl[grp][lane * PITCH + qb];
#pragma unroll
            for (int i = 0; i < NI; i += 4) {
                if constexpr (ABL & 16) break;
                unsigned hw[4], lw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = sg[i + k] * sc;
                    const _Float16 hh = (_Float16)a;
                    hw[k] = __builtin_bit_cast(unsigned short, hh);
                    lw[k] = __builtin_bit_cast(unsigned short, (_Float16)(a - (float)hh));
                }
                *reinterpret_cast<psalm_u32x2*>(sh + i) = psalm_u32x2{hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16)};
                *reinterpret_cast<psalm_u32x2*>(sl + i) = psalm_u32x2{lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16)};
            }
        }
        PSALM_RAW_BARRIER();
        // ---- M1 / M2: (class tile, k-step) = hi.hi + lo.hi + hi.lo, fp32 accumulate
        pp_f32x16 acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        auto frag = [&](const unsigned short* base) -> pp_f16x8 {
            const unsigned long long* p = reinterpret_cast<const unsigned long long*>(base);
            const unsigned long long a0 = p[0], a1 = p[1];
            return __builtin_bit_cast(pp_f16x8, psalm_u32x4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)});
        };
        auto ksteps = [&](int k0, int k1) {
#pragma unroll
            for (int kk = k0; kk < k1; ++kk) {
                const int ko = 16 * kk + 8 * hi;
                const pp_f16x8 bh = frag(&Sh[grp][(32 * ps + n) * PITCH + ko]), bl = frag(&Sl[grp][(32 * ps + n) * PITCH + ko]);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (u < nu && !(ABL & 2)) {
                        const pp_f16x8 ah = frag(&Ph[(32 * (u0 + u) + n) * PITCH + ko]), al = frag(&Pl[(32 * (u0 + u) + n) * PITCH + ko]);
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[u], 0, 0, 0);
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[u], 0, 0, 0);
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[u], 0, 0, 0);
                    }
                }
            }
        };
        ksteps(0, KS1);
        PSALM_RAW_BARRIER();
        ksteps(KS1, KS);
        {
            const float oinv = oinv_s[grp][32 * ps + n];
            const bool pv = live && p0 + 32 * ps + n < HW;
            // lane part of the address: the pixel inside a class row + this half-wave's 4 rows; the class row of (u, r) is wave-uniform
            const unsigned pix = pv ? (unsigned)(p0 + 32 * ps + n) * 4u + (unsigned)(4 * hi) * row_bytes : PSALM_BUF_OOB;
            int u0_ = u0;
            PSALM_OPAQUE_SGPR(u0_);
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cb = 32 * (u0_ + u) + (r & 3) + 8 * (r >> 2);
                    const bool ok = u < nu && cb + 4 * hi < C;
                    if constexpr (!(ABL & 1)) psalm_buf_store_f32_s(acc[u][r] * oinv, orsrc, ok ? pix : PSALM_BUF_OOB, (unsigned)min(cb, C - 1) * row_bytes);
                    else if (acc[u][r] * oinv == 123.456f && ok) psalm_buf_store_f32_s(1.f, orsrc, pix, 0u);     // (keeps the accumulators alive)
                }
        }
        PSALM_RAW_BARRIER();
    }
    if (grp == 0) { PSALM_RAW_BARRIER(); PSALM_RAW_BARRIER(); }
    if (partial) {                                            // (q, 2 * block + group, 2) partial sums -> mask_score_final_kernel (fixed order)
        const long nb = 2 * (long)gridDim.x, b = 2 * (long)blockIdx.x + grp;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = qb + i;
            const float a = wave_sum(num[i]), bsum = wave_sum(den[i]);
            if (lane == 0 && q < Q) {
                partial[(q * nb + b) * 2 + 0] = a;
                partial[(q * nb + b) * 2 + 1] = bsum;
            }
        }
    }
}

// fp32-class form: probsT (C, 128) FLOAT32 (psalm_class_softmax with an fp32 transposed copy); otherwise as psalm_semantic_from_masks.
extern "C" int psalm_semantic_from_masks_x3(const float* mask, const float* probsT_f32, float* out, float* mask_score, float* workspace,
                                            int Q, int C, long HW, int Kpad, void* stream) {
    PSALM_CHECK_ARG(Kpad == 128 && Q >= 1 && Q <= 128 && C >= 1 && C <= 160, "psalm_semantic_from_masks_x3: Kpad 128, Q <= 128, C <= 160");
    PSALM_CHECK_ARG(mask_score == nullptr || workspace != nullptr, "psalm_semantic_from_masks_x3: mask_score needs the workspace");
    PSALM_CHECK_ARG(HW <= (1L << 27), "psalm_semantic_from_masks_x3: HW <= 2^27 (32-bit lane offsets)");
    if (HW == 0) return 0;
    if ((unsigned long)C * (unsigned long)HW * 4ul < (1ul << 31) && (unsigned long)Q * (unsigned long)HW * 4ul < (1ul << 31)) {          // the buffer descriptor of the stores spans `out` (offsets >= 2^31: dropped)
        const int nt64 = (int)((HW + 63) / 64);
        const int grid2 = nt64 < 512 ? (nt64 + 1) / 2 : 256;                  // 1 persistent block (two wave groups) per CU
        const int order_env = 0;                                              // strided tile order (the contiguous-range order measured equal: profiles/r03n_semantic_tile_order.jsonl)
        if (Q <= 112)
            hipLaunchKernelGGL(semantic_from_masks_x3_pair_kernel<112>, dim3(grid2), dim3(512), 0, (hipStream_t)stream, mask, probsT_f32, out,
                               mask_score ? workspace : nullptr, Q, C, HW, nt64, order_env);
        else
            hipLaunchKernelGGL(semantic_from_masks_x3_pair_kernel<128>, dim3(grid2), dim3(512), 0, (hipStream_t)stream, mask, probsT_f32, out,
                               mask_score ? workspace : nullptr, Q, C, HW, nt64, order_env);
        if (mask_score)
            hipLaunchKernelGGL(mask_score_final_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, workspace, mask_score, Q, 2 * grid2);
        PSALM_LAUNCH_END("psalm_semantic_from_masks_x3");
    }
    const int ntiles = (int)((HW + 127) / 128);
    const int grid = ntiles < 256 ? ntiles : 256;            // 1 persistent block per CU (LDS 152 KB)
    if (Q <= 100)
        hipLaunchKernelGGL(semantic_from_masks_x3_kernel<25>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, probsT_f32, out,
                           mask_score ? workspace : nullptr, Q, C, HW, ntiles);
    else
        hipLaunchKernelGGL(semantic_from_masks_x3_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, probsT_f32, out,
                           mask_score ? workspace : nullptr, Q, C, HW, ntiles);
    if (mask_score)
        hipLaunchKernelGGL(mask_score_final_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, workspace, mask_score, Q, grid);
    PSALM_LAUNCH_END("psalm_semantic_from_masks_x3");
}

// mask (Q, HW) f32 logits; probsT (C, 128) bf16 = softmax probabilities transposed and zero-padded (psalm_class_softmax);
// out (C, HW) f32.  Q <= 128, C <= 160.  mask_score (Q) f32 or NULL: the per-query mask score of psalm_mask_scores, accumulated from
// the same read of the logits (workspace: Q * 512 * 2 floats).
extern "C" int psalm_semantic_from_masks(const float* mask, const void* probsT_bf16, float* out, float* mask_score, float* workspace,
                                         int Q, int C, long HW, int Kpad, void* stream) {
    PSALM_CHECK_ARG(Kpad == 128 && Q >= 1 && Q <= 128 && C >= 1 && C <= 160, "psalm_semantic_from_masks: Kpad 128, Q <= 128, C <= 160");
    PSALM_CHECK_ARG((uintptr_t)probsT_bf16 % 16 == 0, "psalm_semantic_from_masks: probsT must be 16-byte aligned");
    PSALM_CHECK_ARG(mask_score == nullptr || workspace != nullptr, "psalm_semantic_from_masks: mask_score needs the workspace");
    PSALM_CHECK_ARG(HW <= (1L << 27), "psalm_semantic_from_masks: HW <= 2^27 (32-bit lane offsets)");
    if (HW == 0) return 0;
    const int ntiles = (int)((HW + 127) / 128);
    const int grid = ntiles < 512 ? ntiles : 512;            // 2 persistent blocks per CU (LDS 76 KB each)
    if (Q <= 100)
        hipLaunchKernelGGL(semantic_from_masks_kernel<25>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, (const bf16_t*)probsT_bf16,
                           out, mask_score ? workspace : nullptr, Q, C, HW, ntiles);
    else
        hipLaunchKernelGGL(semantic_from_masks_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, (const bf16_t*)probsT_bf16,
                           out, mask_score ? workspace : nullptr, Q, C, HW, ntiles);
    if (mask_score)
        hipLaunchKernelGGL(mask_score_final_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, workspace, mask_score, Q, grid);
    PSALM_LAUNCH_END("psalm_semantic_from_masks");
}

// ---------------------------------------------------------------- top-k + instance selection (single block)
// vals (n) = row-major (Q, stride) matrix restricted to the first C columns; picks the k largest (ties: lowest flat
// index), in descending order.  Then the reference's filtering (LP:417-446):
//   label = idx % C, query = idx / C; keep only is_thing[label] (if is_thing != NULL);
//   out_score = value * mask_score[query]; compacted in pick order.  count[0] = number kept.
// Radix select on the 49-bit key (sortable value bits << 17 | (131071 - index)): keys are distinct, so the k-th largest key is a
// sharp threshold (no tie handling) and sorting the k survivors by key reproduces "value descending, index ascending".
// Up to 131072 candidates (100 queries x 1310 classes: the open-vocabulary evaluations with 459 / 847 class prompts fit); the keys are
// recomputed from the values in every pass instead of being held in registers.
#define TOPK_IDX_BITS 17
#define TOPK_IDX_MAX ((1 << TOPK_IDX_BITS) - 1)
__device__ __forceinline__ unsigned long long topk_key(float v, int idx) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                  // monotone map float -> unsigned
    return ((unsigned long long)u << TOPK_IDX_BITS) | (unsigned long long)(TOPK_IDX_MAX - idx);
}

__global__ void __launch_bounds__(1024) topk_select_kernel(const float* __restrict__ vals, int Q, int C, int stride, int k,
                                                           const int* __restrict__ is_thing, const float* __restrict__ mask_score,
                                                           float* __restrict__ out_score, int* __restrict__ out_class,
                                                           int* __restrict__ out_query, int* __restrict__ count, int apply_sigmoid) {
    __shared__ int hist[256];
    __shared__ unsigned long long sel_key[128];
    __shared__ float sel_val[128];
    __shared__ unsigned long long prefix_s;
    __shared__ int remaining_s, nsel;
    const int tid = threadIdx.x;
    const int n = Q * C;
    const int kk = min(min(k, n), 128);
    auto value = [&](int i) -> float {
        float x = vals[(long)(i / C) * stride + (i % C)];
        if (apply_sigmoid) x = sigmoidf_(x);
        return x;
    };
    if (tid == 0) { prefix_s = 0ull; remaining_s = kk; nsel = 0; }
    __syncthreads();
    if (n <= 16 * 1024) {
        // r06, up to 16 candidates per thread (100 queries x 133 classes: 13): the keys stay in REGISTERS and the kk-th largest one is found by
        // bisection on its 49 bits -- T <- T | bit whenever at least kk keys are >= T | bit -- one block-wide count per bit.  The digit histograms below cost one LDS atomic per candidate and pass, and scores crowd into a few bins of the
        // leading digits: ~13 k serialised atomics x 7 passes were ~45 of this kernel's 57 us.  Same threshold (the keys are distinct), same picks.
        unsigned long long key[16];
        float val[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = tid + 1024 * e;
            val[e] = i < n ? value(i) : 0.f;
            key[e] = i < n ? topk_key(val[e], i) : 0ull;                       // (0 is below every real key: real keys have index bits > 0 or value bits > 0)
        }
        // one bit per step; the count of a step = sum over the 16 registers of popcount(ballot(key >= candidate)) -- the compare's own lane mask, scalar
        // arithmetic, no shuffles -- and one LDS atomic per wavefront into the step's own counter (49 counters, zeroed once: no reset race)
        __shared__ int cnt_s[49];
        if (tid < 49) cnt_s[tid] = 0;
        __syncthreads();
        unsigned long long T = 0ull;
        for (int bit = 48; bit >= 0 && kk > 0; --bit) {
            const unsigned long long cand = T | (1ull << bit);
            int c = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) c += __builtin_popcountll(__ballot(key[e] >= cand));
            if ((tid & 63) == 0) atomicAdd(&cnt_s[bit], c);
            __syncthreads();
            if (cnt_s[bit] >= kk) T = cand;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (kk > 0 && tid + 1024 * e < n && key[e] >= T) {
                const int slot = atomicAdd(&nsel, 1);
                sel_key[slot] = key[e];
                sel_val[slot] = val[e];
            }
        __syncthreads();
    } else {
    // 7 passes over 8-bit digits, most significant first (bits 55..0 cover the 49-bit key)
    for (int shift = 48; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        const int rem0 = remaining_s;
        const unsigned long long himask = shift == 48 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = topk_key(value(i), i);
            if ((key & himask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
        }
        __syncthreads();
        // Where does the k-th key fall: the digit d with  (keys above d) < rem <= (keys above d) + hist[d],  d = 0 if no digit >= 1 gets
        // there.  Thread d sums the bins above its own -- independent LDS reads -- and the ONE thread whose digit matches publishes it
        // (r04: thread 0 walked the 256 bins in a chain of dependent LDS reads, 7 times: about half of this kernel's 97 us).
        if (tid < 256) {
            int above = 0;
#pragma unroll 8
            for (int d = tid + 1; d < 256; ++d) above += hist[d];
            const bool mine = tid == 0 ? above < rem0 : (above < rem0 && rem0 <= above + hist[tid]);
            if (mine) {
                remaining_s = rem0 - above;
                prefix_s = prefix | ((unsigned long long)tid << shift);
            }
        }
        __syncthreads();
    }
    const unsigned long long thr = prefix_s;                          // the kk-th largest key
    for (int i = tid; i < n && kk > 0; i += 1024) {
        const float x = value(i);
        const unsigned long long key = topk_key(x, i);
        if (key >= thr) {
            const int slot = atomicAdd(&nsel, 1);
            sel_key[slot] = key;
            sel_val[slot] = x;
        }
    }
    __syncthreads();
    }
    for (int e = nsel + tid; e < 128; e += 1024) { sel_key[e] = 0ull; sel_val[e] = 0.f; }
    __syncthreads();
    // ---- the <= 128 survivors: sorted (descending by key = value descending, index ascending) and filtered by ONE wavefront in registers -- two
    // (key, value) pairs per lane, a bitonic network of shuffles, the output slots from two ballots.  r06: the LDS form spent 28 block-wide barriers
    // of 16 waves on the sort and walked the picks behind one thread (most of what was left of this kernel's 57 us once the selection was cheap).
    if (tid >= 64) return;
    const int lane = tid;
    unsigned long long k2[2] = {sel_key[lane], sel_key[64 + lane]};
    float v2[2] = {sel_val[lane], sel_val[64 + lane]};
    auto shfl64 = [&](unsigned long long x, int m) -> unsigned long long {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(x & 0xffffffffull), m), hi_ = (unsigned)__shfl_xor((int)(unsigned)(x >> 32), m);
        return ((unsigned long long)hi_ << 32) | lo;
    };
#pragma unroll
    for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
        for (int st = size >> 1; st > 0; st >>= 1) {
            if (st == 64) {                                                     // partner = the lane's other pair (size == 128: descending throughout)
                if (k2[0] < k2[1]) {
                    const unsigned long long tk = k2[0]; k2[0] = k2[1]; k2[1] = tk;
                    const float tv = v2[0]; v2[0] = v2[1]; v2[1] = tv;
                }
            } else {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int p_ = sl * 64 + lane;
                    const unsigned long long ok = shfl64(k2[sl], st);
                    const float ov = __shfl_xor(v2[sl], st);
                    const bool desc = (p_ & size) == 0, first = (p_ & st) == 0;       // first: this element is the lower index of the pair
                    const bool take_max = desc == first;
                    if (take_max ? ok > k2[sl] : ok < k2[sl]) { k2[sl] = ok; v2[sl] = ov; }
                }
            }
        }
    }
    int lab[2], qq[2];
    bool keep[2];
    float sc[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int r = sl * 64 + lane;
        const int idx = TOPK_IDX_MAX - (int)(k2[sl] & (unsigned long long)TOPK_IDX_MAX);
        lab[sl] = idx % C;
        qq[sl] = idx / C;
        keep[sl] = r < kk && (!is_thing || is_thing[r < kk ? lab[sl] : 0]);
        sc[sl] = v2[sl] * ((mask_score && r < kk) ? mask_score[qq[sl]] : 1.f);
    }
    const unsigned long long b0 = __ballot(keep[0]), b1 = __ballot(keep[1]);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int n0 = __builtin_popcountll(b0);
    const int pos[2] = {__builtin_popcountll(b0 & below), n0 + __builtin_popcountll(b1 & below)};
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        if (keep[sl]) {
            out_score[pos[sl]] = sc[sl];
            out_class[pos[sl]] = lab[sl];
            out_query[pos[sl]] = qq[sl];
        }
    if (lane == 0) count[0] = n0 + __builtin_popcountll(b1);
}

extern "C" int psalm_topk_select(const float* vals, int Q, int C, int stride, int k, const int* is_thing, const float* mask_score,
                                 float* out_score, int* out_class, int* out_query, int* count, int apply_sigmoid, void* stream) {
    PSALM_CHECK_ARG((long)Q * C <= (1L << TOPK_IDX_BITS) && k <= 128, "psalm_topk_select: at most 131072 candidates, k <= 128");
    hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, vals, Q, C, stride, k, is_thing, mask_score,
                       out_score, out_class, out_query, count, apply_sigmoid);
    PSALM_LAUNCH_END("psalm_topk_select");
}

// ---------------------------------------------------------------- out[i] = (mask[query[i]] > 0) as float, i < count (LP:437)
__global__ void __launch_bounds__(256) binarize_gather_kernel(const float* __restrict__ mask, const int* __restrict__ query,
                                                              const int* __restrict__ count, float* __restrict__ out, long HW) {
    const int i = blockIdx.y;
    if (count && i >= count[0]) return;
    const int q = query ? query[i] : i;
    const float* src = mask + (long)q * HW;
    float* dst = out + (long)i * HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) dst[p] = src[p] > 0.f ? 1.f : 0.f;
}

extern "C" int psalm_binarize_gather(const float* mask, const int* query, const int* count, float* out, int n, long HW, void* stream) {
    if (n == 0 || HW == 0) return 0;
    long gx = (HW + 1023) / 1024;
    hipLaunchKernelGGL(binarize_gather_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx), n), dim3(256), 0, (hipStream_t)stream, mask, query,
                       count, out, HW);
    PSALM_LAUNCH_END("psalm_binarize_gather");
}

// ---------------------------------------------------------------- panoptic (LP:325-386)
// stage a: per pixel argmax over kept queries (label != void && score > thr) of score * sigmoid(mask); integer area counts.
//   counts (Q,3) int32 zero-initialised by the caller: [area(argmax==q), area(sigmoid>=0.5), their intersection]
__global__ void __launch_bounds__(256) panoptic_argmax_kernel(const float* __restrict__ mask, const float* __restrict__ score,
                                                              const int* __restrict__ label, int* __restrict__ argq,
                                                              int* __restrict__ counts, int Q, long HW, int void_label, float thr) {
    HIP_DYNAMIC_SHARED(int, sm)
    int* lc = sm;                       // (Q,3) block-local counts
    float* ksc = (float*)(sm + 3 * Q);  // kept score or -1
    for (int i = threadIdx.x; i < 3 * Q; i += 256) lc[i] = 0;
    for (int q = threadIdx.x; q < Q; q += 256) ksc[q] = (label[q] != void_label && score[q] > thr) ? score[q] : -1.f;
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        float best = -1.f;
        int bq = -1;
        for (int q = 0; q < Q; ++q) {
            if (ksc[q] < 0.f) continue;
            const float s = sigmoidf_(mask[(long)q * HW + p]);
            const float v = ksc[q] * s;
            if (s >= 0.5f) atomicAdd(&lc[3 * q + 1], 1);
            if (v > best) { best = v; bq = q; }
        }
        argq[p] = bq;
        if (bq >= 0) {
            atomicAdd(&lc[3 * bq + 0], 1);
            if (sigmoidf_(mask[(long)bq * HW + p]) >= 0.5f) atomicAdd(&lc[3 * bq + 2], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += 256)
        if (lc[i]) atomicAdd(&counts[i], lc[i]);
}

// ... the form the entry point launches when HW % 4 == 0 and the planes are 16-byte aligned (r05; the kernel above stays for the rest).
// Same integers out -- the counts are sums of indicator values and the arg-max walks the kept queries in the same (ascending) order with
// the same strict `>` -- from a loop the memory system can fill: the kept queries are compacted into an LDS list ONCE per block (the old
// loop tested all Q per pixel and `continue`d), a thread owns 4 consecutive pixels (16-byte loads), the loads of 4 kept queries are issued
// before the first is used, and the `sigmoid >= 0.5` area counts leave through one ballot + population count per wavefront and pixel
// column instead of an LDS atomic per (pixel, query).  SQ counters of the old kernel (r04j): 86 % of the wavefront cycles parked.
__global__ void __launch_bounds__(256) panoptic_argmax_vec4_kernel(const float* __restrict__ mask, const float* __restrict__ score,
                                                                   const int* __restrict__ label, int* __restrict__ argq,
                                                                   int* __restrict__ counts, int Q, long HW, int void_label, float thr) {
    HIP_DYNAMIC_SHARED(int, sm)
    int* lc = sm;                        // (Q,3) block-local counts
    float* ksc = (float*)(sm + 3 * Q);   // kept scores, compacted
    int* kq = sm + 4 * Q;                // their query indices (ascending)
    __shared__ int nk_s;
    for (int i = threadIdx.x; i < 3 * Q; i += 256) lc[i] = 0;
    if (Q <= 256) {                      // r06: thread q tests query q; its slot = the kept queries in front of it (flags in LDS: kq doubles as the flag array)
        const int q = threadIdx.x;       // (every block re-derives the list: thread 0 walking Q label / score pairs was each block's first ~10 us)
        const float sc = q < Q ? score[q] : 0.f;
        const bool keep = q < Q && label[q] != void_label && sc > thr;
        __shared__ int flag_s[256];
        flag_s[q] = keep ? 1 : 0;
        __syncthreads();
        int pos = 0;
        for (int j = 0; j < q; ++j) pos += flag_s[j];
        if (keep) { ksc[pos] = sc; kq[pos] = q; }
        if (q == 255) nk_s = pos + flag_s[255];
    } else if (threadIdx.x == 0) {       // (serial scan: keeps the order without a prefix sum)
        int n = 0;
        for (int q = 0; q < Q; ++q)
            if (label[q] != void_label && score[q] > thr) { ksc[n] = score[q]; kq[n] = q; ++n; }
        nk_s = n;
    }
    __syncthreads();
    const int nk = nk_s, lane = threadIdx.x & 63;
    const long nquad = HW >> 2;
    for (long pq = (long)blockIdx.x * 256 + threadIdx.x; pq - threadIdx.x < nquad; pq += (long)gridDim.x * 256) {
        const bool live = pq < nquad;                                // (whole wavefronts stay in the loop: the ballots below need them)
        const long p = (live ? pq : 0) << 2;
        float best[4] = {-1.f, -1.f, -1.f, -1.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
        int bq[4] = {-1, -1, -1, -1};
        for (int k0 = 0; k0 < nk; k0 += 4) {
            psalm_f32x4 m4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = kq[min(k0 + u, nk - 1)];
                m4[u] = *reinterpret_cast<const psalm_f32x4*>(mask + (long)q * HW + p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u >= nk) break;                             // (block-uniform)
                const int q = kq[k0 + u];
                const float sc = ksc[k0 + u];
                const float mv[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
                int c1 = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s_ = sigmoidf_(mv[e]);
                    const float v = sc * s_;
                    c1 += __builtin_popcountll(__ballot(live && s_ >= 0.5f));
                    if (v > best[e]) { best[e] = v; bq[e] = q; bs[e] = s_; }
                }
                if (lane == 0 && c1) atomicAdd(&lc[3 * q + 1], c1);
            }
        }
        if (live) {
            *reinterpret_cast<psalm_u32x4*>(argq + p) = psalm_u32x4{(unsigned)bq[0], (unsigned)bq[1], (unsigned)bq[2], (unsigned)bq[3]};
            // the 4 pixels of a thread mostly share their winner: one LDS atomic per run of equal winners
            int run_q = bq[0], run_a = 0, run_i = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (bq[e] != run_q) {
                    if (run_q >= 0) { atomicAdd(&lc[3 * run_q + 0], run_a); if (run_i) atomicAdd(&lc[3 * run_q + 2], run_i); }
                    run_q = bq[e]; run_a = 0; run_i = 0;
                }
                run_a += 1;
                run_i += (bq[e] >= 0 && bs[e] >= 0.5f) ? 1 : 0;
            }
            if (run_q >= 0) { atomicAdd(&lc[3 * run_q + 0], run_a); if (run_i) atomicAdd(&lc[3 * run_q + 2], run_i); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += 256)
        if (lc[i]) atomicAdd(&counts[i], lc[i]);
}

// stage b (one thread): the sequential merge; final_id[q] = segment id or 0; info (n,3) = (id, isthing, category)
__global__ void panoptic_merge_kernel(const float* __restrict__ score, const int* __restrict__ label, const int* __restrict__ counts,
                                      const int* __restrict__ is_thing, int* __restrict__ final_id, int* __restrict__ info,
                                      int* __restrict__ ninfo, int Q, int void_label, float thr, float overlap_thr, int num_classes) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cur = 0, n = 0;
    // stuff_memory: class -> id, kept in `info`-independent scratch at the tail of final_id (size num_classes)
    int* stuff = final_id + Q;
    for (int c = 0; c < num_classes; ++c) stuff[c] = 0;
    for (int q = 0; q < Q; ++q) {
        final_id[q] = 0;
        if (!(label[q] != void_label && score[q] > thr)) continue;
        const int area = counts[3 * q], orig = counts[3 * q + 1], inter = counts[3 * q + 2];
        if (area > 0 && orig > 0 && inter > 0) {
            if ((float)area / (float)orig < overlap_thr) continue;
            const int pc = label[q];
            const int thing = is_thing[pc];
            if (!thing) {
                if (stuff[pc] != 0) { final_id[q] = stuff[pc]; continue; }
                stuff[pc] = cur + 1;
            }
            cur += 1;
            final_id[q] = cur;
            info[3 * n] = cur; info[3 * n + 1] = thing ? 1 : 0; info[3 * n + 2] = pc;
            n++;
        }
    }
    ninfo[0] = n;
}

// stage b for Q <= 128 (r06): the same merge with one thread per query.  What is sequential in the loop above is only (i) "the first kept query of a
// stuff class opens the segment, later ones join it" and (ii) the running segment number -- a first-occurrence search and a prefix count over <= 128
// flags in LDS; every global operand (label, score, the three counts, the thing flag) is fetched by all queries at once instead of one after the
// other behind thread 0 (27 us for 100 queries, profiles/r05_kernel_stats.txt).  final_id, info and ninfo are word for word the loop's.
__global__ void __launch_bounds__(128) panoptic_merge_par_kernel(const float* __restrict__ score, const int* __restrict__ label,
                                                                 const int* __restrict__ counts, const int* __restrict__ is_thing,
                                                                 int* __restrict__ final_id, int* __restrict__ info, int* __restrict__ ninfo, int Q,
                                                                 int void_label, float thr, float overlap_thr) {
    __shared__ int s_lab[128], s_new[128], s_id[128];
    const int q = threadIdx.x;
    bool valid = false, thing = false;
    int pc = -1;
    if (q < Q) {
        pc = label[q];
        const int area = counts[3 * q], orig = counts[3 * q + 1], inter = counts[3 * q + 2];
        valid = pc != void_label && score[q] > thr && area > 0 && orig > 0 && inter > 0;
        if (valid && (float)area / (float)orig < overlap_thr) valid = false;
        if (valid) thing = is_thing[pc] != 0;
    }
    s_lab[q] = (valid && !thing) ? pc : -1;                    // kept stuff queries by class (-1: not one)
    __syncthreads();
    int jfirst = q;
    if (valid && !thing) {
        for (int j = q - 1; j >= 0; --j)
            if (s_lab[j] == pc) jfirst = j;                     // the smallest j wins
    }
    const bool isnew = valid && (thing || jfirst == q);
    s_new[q] = isnew ? 1 : 0;
    __syncthreads();
    int before = 0;
    for (int j = 0; j < q; ++j) before += s_new[j];
    s_id[q] = isnew ? before + 1 : 0;
    __syncthreads();
    if (q < Q) {
        final_id[q] = isnew ? before + 1 : (valid ? s_id[jfirst] : 0);
        if (isnew) { info[3 * before] = before + 1; info[3 * before + 1] = thing ? 1 : 0; info[3 * before + 2] = pc; }
    }
    if (q == 127) ninfo[0] = before + s_new[127];
}

// stage c: pan[p] = final_id[argq[p]] where sigmoid(mask[argq[p], p]) >= 0.5, else 0
__global__ void __launch_bounds__(256) panoptic_write_kernel(const float* __restrict__ mask, const int* __restrict__ argq,
                                                             const int* __restrict__ final_id, int* __restrict__ pan, long HW) {
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        const int q = argq[p];
        int id = 0;
        if (q >= 0 && sigmoidf_(mask[(long)q * HW + p]) >= 0.5f) id = final_id[q];
        pan[p] = id;
    }
}

// argq (HW) i32 scratch; counts (Q*3) i32 scratch; final_id (Q + num_classes) i32 scratch; pan (HW) i32; info (Q*3) i32; ninfo (1) i32
extern "C" int psalm_panoptic(const float* mask, const float* score, const int* label, const int* is_thing, int* argq, int* counts,
                              int* final_id, int* pan, int* info, int* ninfo, int Q, long HW, int num_classes, float obj_thr,
                              float overlap_thr, void* stream) {
    if (HW == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int) * 3 * Q, s) != hipSuccess) { psalm_set_error("psalm_panoptic: memset failed"); return -2; }
    long gx = (HW + 255) / 256;
    const int grid = (int)(gx > 4096 ? 4096 : gx);
    const size_t shmem = (size_t)(3 * Q) * sizeof(int) + (size_t)Q * sizeof(float);
    if (HW % 4 == 0 && (uintptr_t)mask % 16 == 0 && (uintptr_t)argq % 16 == 0) {
        const long gq = (HW / 4 + 255) / 256;
        hipLaunchKernelGGL(panoptic_argmax_vec4_kernel, dim3((unsigned)(gq > 4096 ? 4096 : gq)), dim3(256), (size_t)(5 * Q) * sizeof(int), s, mask, score,
                           label, argq, counts, Q, HW, num_classes, obj_thr);
    } else {
        hipLaunchKernelGGL(panoptic_argmax_kernel, dim3(grid), dim3(256), shmem, s, mask, score, label, argq, counts, Q, HW, num_classes,
                           obj_thr);
    }
    if (Q <= 128)
        hipLaunchKernelGGL(panoptic_merge_par_kernel, dim3(1), dim3(128), 0, s, score, label, counts, is_thing, final_id, info, ninfo, Q, num_classes, obj_thr,
                           overlap_thr);
    else
        hipLaunchKernelGGL(panoptic_merge_kernel, dim3(1), dim3(64), 0, s, score, label, counts, is_thing, final_id, info, ninfo, Q,
                           num_classes, obj_thr, overlap_thr, num_classes);
    hipLaunchKernelGGL(panoptic_write_kernel, dim3(grid), dim3(256), 0, s, mask, argq, final_id, pan, HW);
    PSALM_LAUNCH_END("psalm_panoptic");
}

// ---------------------------------------------------------------- region scores (LP:390-399): out[q, k] = sigmoid(logits[k, q]) * mask_score[q]
__global__ void region_scores_kernel(const float* __restrict__ logits, const float* __restrict__ mask_score, float* __restrict__ out,
                                     int K, int Q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * Q) return;
    const int q = i / K, k = i % K;
    out[i] = sigmoidf_(logits[(long)k * Q + q]) * mask_score[q];
}
extern "C" int psalm_region_scores(const float* logits, const float* mask_score, float* out, int K, int Q, void* stream) {
    if (K * Q == 0) return 0;
    hipLaunchKernelGGL(region_scores_kernel, dim3(cdiv(K * Q, 256)), dim3(256), 0, (hipStream_t)stream, logits, mask_score, out, K, Q);
    PSALM_LAUNCH_END("psalm_region_scores");
}
