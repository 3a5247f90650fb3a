/* CPU ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the reference's MSDeformAttn forward arithmetic, loop for loop:
 *   psalm/model/mask_decoder/Mask2Former_Simplify/modeling/pixel_decoder/ops/src/cuda/
 *   ms_deform_im2col_cuda.cuh:242-304  (ms_deformable_im2col_gpu_kernel: index decomposition,
 *                                       level/point loops, the -1 < h_im < H test)
 *   ms_deform_im2col_cuda.cuh:38-89    (ms_deform_attn_im2col_bilinear: floor, 4 guarded corners)
 * PINNING: checked in tests/test_msda.py against the reference's own pure-PyTorch formula
 * (ops/functions/ms_deform_attn_func.py:52-78, restated as oracle.psalm_oracle.msda_core_grid_sample)
 * on the exact seed/shapes of the reference's only known-answer test (ops/test.py:24-63).
 * The accumulator type is double when compiled with -DMSDA_ACC_DOUBLE (used to bound fp32 error).
 */
#include <math.h>
#include <stdint.h>

#ifdef MSDA_ACC_DOUBLE
typedef double acc_t;
#else
typedef float acc_t;
#endif

static acc_t bilinear(const float* bottom, int height, int width, int nheads, int channels, acc_t h, acc_t w, int m, int c) {
    const int h_low = (int)floor(h), w_low = (int)floor(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const acc_t lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
    const int w_stride = nheads * channels, h_stride = width * w_stride;
    const int base = m * channels + c;
    acc_t v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = bottom[h_low * h_stride + w_low * w_stride + base];
    if (h_low >= 0 && w_high <= width - 1) v2 = bottom[h_low * h_stride + w_high * w_stride + base];
    if (h_high <= height - 1 && w_low >= 0) v3 = bottom[h_high * h_stride + w_low * w_stride + base];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = bottom[h_high * h_stride + w_high * w_stride + base];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* value (B,S,M,D)  shapes (L,2)=(H,W)  starts (L)  loc (B,Lq,M,L,P,2)=(x,y)  w (B,Lq,M,L,P)  out (B,Lq,M*D) */
void msda_forward_ref(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc, const float* attw,
                      float* out, int B, int S, int M, int D, int L, int Lq, int P) {
    const long n = (long)B * Lq * M * D;
    for (long index = 0; index < n; ++index) {
        long t = index;
        const int c = (int)(t % D); t /= D;
        const long sampling_index = t;
        const int m = (int)(t % M); t /= M;
        t /= Lq;
        const int b = (int)t;
        long wptr = sampling_index * L * P, lptr = wptr << 1;
        const int qid_stride = M * D;
        acc_t col = 0;
        for (int l = 0; l < L; ++l) {
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const float* vptr = value + ((long)b * S + starts[l]) * qid_stride;
            for (int p = 0; p < P; ++p) {
                const acc_t loc_w = loc[lptr], loc_h = loc[lptr + 1], weight = attw[wptr];
                const acc_t h_im = loc_h * Hl - 0.5, w_im = loc_w * Wl - 0.5;
                if (h_im > -1 && w_im > -1 && h_im < Hl && w_im < Wl)
                    col += bilinear(vptr, Hl, Wl, M, D, h_im, w_im, m, c) * weight;
                wptr += 1; lptr += 2;
            }
        }
        out[index] = (float)col;
    }
}
