// Data-layout kernels around the GEMMs: im2col for the patch-embed / projector / FPN convolutions, bilinear
// resampling (PyTorch align_corners=False index rule), sine position embedding.  Activations are NHWC
// ("tokens x channels") everywhere so 1x1 convolutions are plain GEMMs and 3x3 / strided ones are im2col + GEMM.
#include "common.h"

// ---------------------------------------------------------------- PatchEmbed im2col from the NCHW fp32 image
// swin_trans.py:427-436: Conv2d(3, E, k=ps, stride=ps) after right/bottom zero padding to a multiple of ps.
// out (B*Hp*Wp, Kpad) with K order (c, ky, kx) == the flattened conv weight (E, 3, ps, ps); columns >= 3*ps*ps are 0.
template <typename TO>
__global__ void __launch_bounds__(256) patch_im2col_kernel(const float* __restrict__ img, TO* __restrict__ out, int B, int Cin,
                                                           int H, int W, int ps, int Hp, int Wp, int Kpad) {
    const long total = (long)B * Hp * Wp * Kpad;
    const int K = Cin * ps * ps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kpad);
        long t = i / Kpad;
        const int px = (int)(t % Wp);
        t /= Wp;
        const int py = (int)(t % Hp);
        const int b = (int)(t / Hp);
        float v = 0.f;
        if (k < K) {
            const int c = k / (ps * ps), ky = (k / ps) % ps, kx = k % ps;
            const int y = py * ps + ky, x = px * ps + kx;
            if (y < H && x < W) v = img[(((long)b * Cin + c) * H + y) * W + x];
        }
        stf(out + i, v);
    }
}

extern "C" int psalm_patch_im2col(const float* img, void* out, int out_dtype, int B, int Cin, int H, int W, int ps, int Kpad,
                                  void* stream) {
    const int Hp = (H + ps - 1) / ps, Wp = (W + ps - 1) / ps;
    const long total = (long)B * Hp * Wp * Kpad;
    if (total == 0) return 0;
    PSALM_CHECK_ARG(Kpad >= Cin * ps * ps, "psalm_patch_im2col: Kpad too small");
    const int grid = (int)((total + 2047) / 2048);
    PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((patch_im2col_kernel<TO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (TO*)out, B, Cin, H, W,
                           ps, Hp, Wp, Kpad);
    });
    PSALM_LAUNCH_END("psalm_patch_im2col");
}

// ---------------------------------------------------------------- generic NHWC im2col, K order (ky, kx, c)
// x (B,H,W,C) -> out (B*Ho*Wo, k*k*C); weights are pre-permuted on the host to (Cout, ky, kx, Cin).
template <typename T>
__global__ void __launch_bounds__(256) im2col_nhwc_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C,
                                                          int k, int stride, int pad, int Ho, int Wo) {
    const long total = (long)B * Ho * Wo * k * k * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int kx = (int)(t % k);
        t /= k;
        const int ky = (int)(t % k);
        t /= k;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int y = oy * stride - pad + ky, xx = ox * stride - pad + kx;
        T v = 0;
        if (y >= 0 && y < H && xx >= 0 && xx < W) v = x[(((long)b * H + y) * W + xx) * C + c];
        out[i] = v;
    }
}

extern "C" int psalm_im2col_nhwc(const void* x, void* out, int dtype, int B, int H, int W, int C, int k, int stride, int pad,
                                 void* stream) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long total = (long)B * Ho * Wo * k * k * C;
    if (total == 0) return 0;
    long g = (total + 2047) / 2048;
    const int grid = (int)(g > 1048576 ? 1048576 : g);
    PSALM_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((im2col_nhwc_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, B, H, W, C,
                           k, stride, pad, Ho, Wo);
    });
    PSALM_LAUNCH_END("psalm_im2col_nhwc");
}

// ---------------------------------------------------------------- bilinear resize, PyTorch align_corners=False rule
// src index = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out (float); neighbour +1 clamped at the border.
struct BilinIdx { int i0, i1; float l0, l1; };
__device__ __forceinline__ BilinIdx bilin_idx(int d, float scale, int in_size) {
    float s = scale * (d + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    BilinIdx r;
    r.i0 = (int)s;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = s - r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// ly0 (lx0 a00 + lx1 a01) + ly1 (lx0 a10 + lx1 a11) with the fused multiply-adds WRITTEN OUT: the three resize kernels below are reached through
// one entry point (by dtype / alignment / height) and must round alike -- left to the compiler, each instantiation contracts the expression
// its own way (r04j on the MI355X: the one-pixel-per-thread kernel and the row-walking kernel 1 ulp apart on 30 % of the pixels).  This
// form is also what torch's CPU kernel evaluates (bit-identical on the host emulator).
__device__ __forceinline__ float bilin_blend(float ly0, float ly1, float lx0, float lx1, float a00, float a01, float a10, float a11) {
    return fmaf(ly0, fmaf(lx0, a00, lx1 * a01), ly1 * fmaf(lx0, a10, lx1 * a11));
}

// planes: in (N, h, w) -> out (N, H, W), with an optional crop of the input to (hc, wc) first
// (sem_seg_postprocess = crop + resize, llava_phi.py:1427-1429)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) resize_planes_kernel(const TI* __restrict__ in, TO* __restrict__ out, long N, int h, int w,
                                                            int hc, int wc, int H, int W) {
    const long total = N * H * W;
    const float sh = (float)hc / H, sw = (float)wc / W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const long n = i / ((long)W * H);
        const BilinIdx iy = bilin_idx(y, sh, hc), ix = bilin_idx(x, sw, wc);
        const TI* p = in + n * h * w;
        const float v = bilin_blend(iy.l0, iy.l1, ix.l0, ix.l1, ldf(p + (long)iy.i0 * w + ix.i0), ldf(p + (long)iy.i0 * w + ix.i1),
                                    ldf(p + (long)iy.i1 * w + ix.i0), ldf(p + (long)iy.i1 * w + ix.i1));
        stf(out + i, v);
    }
}

// fp32 -> fp32 form with W % 4 == 0: one thread produces 4 consecutive output pixels (one 16-byte store; the 4-corner reads
// hit L1/L2 -- the source planes are 16x smaller than the output at the 256^2 -> 1024^2 mask upsampling)
__global__ void __launch_bounds__(256) resize_planes_vec4_kernel(const float* __restrict__ in, float* __restrict__ out, long N, int h,
                                                                 int w, int hc, int wc, int H, int W) {
    const int W4 = W >> 2;
    const long total4 = N * H * W4;
    const float sh = (float)hc / H, sw = (float)wc / W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int x0 = (int)(i % W4) * 4;
        const int y = (int)((i / W4) % H);
        const long n = i / ((long)W4 * H);
        const BilinIdx iy = bilin_idx(y, sh, hc);
        const float* r0 = in + n * h * w + (long)iy.i0 * w;
        const float* r1 = in + n * h * w + (long)iy.i1 * w;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const BilinIdx ix = bilin_idx(x0 + k, sw, wc);
            o[k] = bilin_blend(iy.l0, iy.l1, ix.l0, ix.l1, r0[ix.i0], r0[ix.i1], r1[ix.i0], r1[ix.i1]);
        }
        *reinterpret_cast<psalm_f32x4*>(out + ((n * H + y) * (long)W + x0)) = psalm_f32x4{o[0], o[1], o[2], o[3]};
    }
}

// fp32 -> fp32, W % 4 == 0, the form the mask up-sampling runs (100 planes 256^2 -> 1024^2: 419 MB written, 26 MB read): the vec4 kernel
// above spends ~35 VALU instructions and 4 scattered loads per output pixel on index arithmetic that neighbouring outputs share (r04:
// 160 us = 2.7 TB/s, VALU / address bound, not HBM bound).  Here a block owns RB consecutive output rows of one plane and a thread 4 consecutive
// pixels of each of them: the 4 column stencils are computed ONCE per thread, a row's stencil once per row (block-uniform), and the
// 2 x 8 source values are re-fetched only when the source row pair changes -- the previous lower row becomes the upper one (block-uniform
// branches; 4x up-sampling: 8 + 8 + 8 + 8 loads per 32 outputs).  Same expression per output (bilin_blend), 16-byte stores.
template <int RB>
__global__ void __launch_bounds__(256) resize_planes_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int hc,
                                                                 int wc, int H, int W) {
    const int W4 = W >> 2;
    const int bands = (H + RB - 1) / RB;
    const long n = blockIdx.x / bands;
    const int y0 = (int)(blockIdx.x % bands) * RB;
    const float sh = (float)hc / H, sw = (float)wc / W;
    const float* plane = in + n * h * w;
    for (int x4 = threadIdx.x; x4 < W4; x4 += 256) {
        BilinIdx ix[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ix[k] = bilin_idx(4 * x4 + k, sw, wc);
        float a[2][8];                                            // [upper | lower source row][pixel k: value at i0, value at i1]
        int p0 = -1, p1 = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[0][k] = a[1][k] = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int y = y0 + r;
            if (y >= H) break;
            const BilinIdx iy = bilin_idx(y, sh, hc);
            if (iy.i0 != p0) {
                if (iy.i0 == p1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[0][k] = a[1][k];
                } else {
                    const float* rp = plane + (long)iy.i0 * w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[0][2 * k] = rp[ix[k].i0]; a[0][2 * k + 1] = rp[ix[k].i1]; }
                }
                p0 = iy.i0;
            }
            if (iy.i1 != p1) {
                if (iy.i1 == p0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[1][k] = a[0][k];
                } else {
                    const float* rp = plane + (long)iy.i1 * w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[1][2 * k] = rp[ix[k].i0]; a[1][2 * k + 1] = rp[ix[k].i1]; }
                }
                p1 = iy.i1;
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = bilin_blend(iy.l0, iy.l1, ix[k].l0, ix[k].l1, a[0][2 * k], a[0][2 * k + 1], a[1][2 * k], a[1][2 * k + 1]);
            *reinterpret_cast<psalm_f32x4*>(out + ((n * H + y) * (long)W + 4 * x4)) = psalm_f32x4{o[0], o[1], o[2], o[3]};
        }
    }
}

extern "C" int psalm_resize_planes(const void* in, int in_dtype, void* out, int out_dtype, long N, int h, int w, int hc, int wc,
                                   int H, int W, void* stream) {
    const long total = N * H * W;
    if (total == 0) return 0;
    PSALM_CHECK_ARG(hc <= h && wc <= w && hc > 0 && wc > 0, "psalm_resize_planes: bad crop");
    if (in_dtype == PSALM_F32 && out_dtype == PSALM_F32 && W % 4 == 0 && (uintptr_t)out % 16 == 0 && H >= 8 &&
        N * ((H + 7) / 8) <= 0x7fffffffL) {
        hipLaunchKernelGGL((resize_planes_rows_kernel<8>), dim3((unsigned)(N * ((H + 7) / 8))), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, h, w, hc, wc, H, W);
        PSALM_LAUNCH_END("psalm_resize_planes");
    }
    if (in_dtype == PSALM_F32 && out_dtype == PSALM_F32 && W % 4 == 0 && (uintptr_t)out % 16 == 0) {
        long g4 = (total / 4 + 255) / 256;
        hipLaunchKernelGGL(resize_planes_vec4_kernel, dim3((unsigned)(g4 > 1048576 ? 1048576 : g4)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, N, h, w, hc, wc, H, W);
        PSALM_LAUNCH_END("psalm_resize_planes");
    }
    long g = (total + 1023) / 1024;
    const int grid = (int)(g > 1048576 ? 1048576 : g);
    PSALM_DISPATCH(in_dtype, TI, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((resize_planes_kernel<TI, TO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const TI*)in, (TO*)out, N,
                           h, w, hc, wc, H, W);
    }));
    PSALM_LAUNCH_END("psalm_resize_planes");
}

// NHWC: out (B,H,W,C) = lateral (B,H,W,C) + bilinear_up(small (B,h,w,C))   -- FPN top-down step, msdeformattn.py:306
// (the reference upsamples in fp32: `.float()` then cast back)
template <typename TL, typename TS, typename TO>
__global__ void __launch_bounds__(256) upsample_add_nhwc_kernel(const TL* __restrict__ lat, const TS* __restrict__ small,
                                                                TO* __restrict__ out, int B, int h, int w, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    const float sh = (float)h / H, sw = (float)w / W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const BilinIdx iy = bilin_idx(y, sh, h), ix = bilin_idx(x, sw, w);
        const TS* p = small + (long)b * h * w * C + c;
        const float v = iy.l0 * (ix.l0 * ldf(p + ((long)iy.i0 * w + ix.i0) * C) + ix.l1 * ldf(p + ((long)iy.i0 * w + ix.i1) * C)) +
                        iy.l1 * (ix.l0 * ldf(p + ((long)iy.i1 * w + ix.i0) * C) + ix.l1 * ldf(p + ((long)iy.i1 * w + ix.i1) * C));
        stf(out + i, ldf(lat + i) + v);
    }
}

extern "C" int psalm_upsample_add_nhwc(const void* lateral, int lat_dtype, const void* small, int small_dtype, void* out,
                                       int out_dtype, int B, int h, int w, int H, int W, int C, void* stream) {
    const long total = (long)B * H * W * C;
    if (total == 0) return 0;
    long g = (total + 2047) / 2048;
    const int grid = (int)(g > 1048576 ? 1048576 : g);
    PSALM_DISPATCH(lat_dtype, TL, PSALM_DISPATCH(small_dtype, TS, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((upsample_add_nhwc_kernel<TL, TS, TO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const TL*)lateral,
                           (const TS*)small, (TO*)out, B, h, w, H, W, C);
    })));
    PSALM_LAUNCH_END("psalm_upsample_add_nhwc");
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC and back (API boundary only)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) permute_nchw_nhwc_kernel(const TI* __restrict__ in, TO* __restrict__ out, int B, int C,
                                                                long HW, int to_nhwc) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // i indexes the OUTPUT
        if (to_nhwc) {
            const int c = (int)(i % C);
            const long p = (i / C) % HW;
            const long b = i / ((long)C * HW);
            stf(out + i, ldf(in + (b * C + c) * HW + p));
        } else {
            const long p = i % HW;
            const int c = (int)((i / HW) % C);
            const long b = i / ((long)C * HW);
            stf(out + i, ldf(in + (b * HW + p) * C + c));
        }
    }
}

extern "C" int psalm_permute_layout(const void* in, int in_dtype, void* out, int out_dtype, int B, int C, long HW, int to_nhwc,
                                    void* stream) {
    const long total = (long)B * C * HW;
    if (total == 0) return 0;
    long g = (total + 2047) / 2048;
    const int grid = (int)(g > 1048576 ? 1048576 : g);
    PSALM_DISPATCH(in_dtype, TI, PSALM_DISPATCH(out_dtype, TO, {
        hipLaunchKernelGGL((permute_nchw_nhwc_kernel<TI, TO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const TI*)in, (TO*)out,
                           B, C, HW, to_nhwc);
    }));
    PSALM_LAUNCH_END("psalm_permute_layout");
}

// ---------------------------------------------------------------- region pooling (visual prompts)
// context_cluster.py:333-400: per region, bilinear-sample (grid_sample, align_corners=True, zero padding) the image's
// projector tokens viewed as an (h, w, C) map at n points (y,x in [0,1)), then average over the points.
// tokens (B*h*w, C) f32; img_of_region (R) int32; pts (R, n, 2) f32 (y, x); out (R, C) f32.  One block per region.
__global__ void __launch_bounds__(256) region_pool_kernel(const float* __restrict__ tokens, const int* __restrict__ img_of_region,
                                                          const float* __restrict__ pts, float* __restrict__ out, int h, int w,
                                                          int C, int n) {
    const int r = blockIdx.x;
    const float* fmap = tokens + (long)img_of_region[r] * h * w * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int p = 0; p < n; ++p) {
            const float gy = 2.0f * pts[((long)r * n + p) * 2 + 0] - 1.0f, gx = 2.0f * pts[((long)r * n + p) * 2 + 1] - 1.0f;
            const float iy = (gy + 1.f) * 0.5f * (h - 1), ix = (gx + 1.f) * 0.5f * (w - 1);
            const float fy = floorf(iy), fx = floorf(ix);
            const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
            const float wy1 = iy - fy, wx1 = ix - fx, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
            float v = 0.f;
            if (y0 >= 0 && y0 < h && x0 >= 0 && x0 < w) v += wy0 * wx0 * fmap[((long)y0 * w + x0) * C + c];
            if (y0 >= 0 && y0 < h && x1 >= 0 && x1 < w) v += wy0 * wx1 * fmap[((long)y0 * w + x1) * C + c];
            if (y1 >= 0 && y1 < h && x0 >= 0 && x0 < w) v += wy1 * wx0 * fmap[((long)y1 * w + x0) * C + c];
            if (y1 >= 0 && y1 < h && x1 >= 0 && x1 < w) v += wy1 * wx1 * fmap[((long)y1 * w + x1) * C + c];
            acc += v;
        }
        out[(long)r * C + c] = acc / n;
    }
}

extern "C" int psalm_region_pool(const float* tokens, const int* img_of_region, const float* pts, float* out, int R, int h, int w,
                                 int C, int n, void* stream) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(region_pool_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, tokens, img_of_region, pts, out, h, w, C, n);
    PSALM_LAUNCH_END("psalm_region_pool");
}

// ---------------------------------------------------------------- im2col straight into split-f16 form (precision = "f16x3")
// The A operand of a convolution-as-GEMM in the f16x3 mode: row m = output pixel, K = (ky, kx, c), emitted as [hi (Kp) | lo (Kp)] f16 with
// a per-row power-of-two scale (see psalm_split_f16) WITHOUT materialising the fp32 im2col matrix: for the FPN 3x3 convolution at 256^2
// (msdeformattn.py:248-254) that matrix is 604 MB written, read twice by the split and written again (2.4 GB of traffic); here the
// 67 MB NHWC input is read (9x, from L2) and the 604 MB split operand written once.  One wavefront per output pixel, two passes over
// its k*k taps (row maximum, then conversion); C % 8 == 0.
// WPR = 4 (r06): all four wavefronts of the block share ONE output pixel (row maximum through LDS) -- for the few, long rows of the projector's
// convolutions (256 output pixels x K = 9216 / 18432: one wavefront per row left 192 compute units idle and walked 36 + 36 dependent iterations;
// 33 -> 11 us per call on the serial path between the vision tower and the LLM).  The maximum is exact in any order: same words.
template <int WPR>
__global__ void __launch_bounds__(256) im2col_split_f16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, float* __restrict__ inv_scale,
                                                               int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int K, int Kp) {
    const int lane = WPR == 1 ? (threadIdx.x & 63) : threadIdx.x;          // position among the row's lanes
    constexpr int STEP = 512 * WPR;                                         // K elements per pass of the row's lanes
    const long row = WPR == 1 ? (long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long)blockIdx.x;
    if (row >= (long)B * Ho * Wo) return;
    const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((long)Wo * Ho));
    const float* xb = x + (long)b * H * W * C;
    auto load = [&](int e, float* v) {                       // 8 consecutive K elements starting at e (inside one tap: C % 8 == 0)
        const int tap = e / C, c = e - tap * C;
        const int ky = tap / k, kx = tap - ky * k;
        const int y = oy * stride - pad + ky, xx = ox * stride - pad + kx;
        if (e < K && y >= 0 && y < H && xx >= 0 && xx < W) ld8(xb + ((long)y * W + xx) * C + c, v);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
    };
    float amax = 0.f;
    for (int e = lane * 8; e < K; e += STEP) {
        float v[8];
        load(e, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    }
    amax = wave_max(amax);
    if constexpr (WPR > 1) {
        __shared__ float wmax[WPR];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WPR; ++w) amax = fmaxf(amax, wmax[w]);
    }
    int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    int se = 13 - ex;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    const bool zero = !(amax > 0.f) || !(amax < 3.0e38f);
    const float sc = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
    const float inv = zero ? 1.f : __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
    if (lane == 0) inv_scale[row] = inv;
    unsigned short* orow = out + row * 2L * Kp;
    for (int e = lane * 8; e < Kp; e += STEP) {
        float v[8];
        load(e, v);
        unsigned hw[4], lw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = v[2 * i] * sc, a1 = v[2 * i + 1] * sc;
            const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
            const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
            hw[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lw[i] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        *reinterpret_cast<psalm_u32x4*>(orow + e) = psalm_u32x4{hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<psalm_u32x4*>(orow + Kp + e) = psalm_u32x4{lw[0], lw[1], lw[2], lw[3]};
    }
}

// x (B,H,W,C) f32 NHWC -> out (B*Ho*Wo, 2*Kp) f16 [hi | lo], Kp = ceil64(k*k*C), inv_scale (B*Ho*Wo): psalm_im2col_nhwc + psalm_split_f16
// in one pass.  C % 8 == 0; 16-byte aligned x / out.
extern "C" int psalm_im2col_split_f16(const float* x, void* out, float* inv_scale, int B, int H, int W, int C, int k, int stride, int pad,
                                      void* stream) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long rows = (long)B * Ho * Wo;
    if (rows <= 0) return 0;
    PSALM_CHECK_ARG(C > 0 && C % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "psalm_im2col_split_f16: C % 8 == 0, 16-byte aligned buffers");
    const int K = k * k * C, Kp = (K + 63) / 64 * 64;
    if (rows <= 2048 && K >= 2048 && psalm_get_tuning(PSALM_TUNE_ROW_GROUPS))      // few long rows: a block per row
        hipLaunchKernelGGL((im2col_split_f16_kernel<4>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)out, inv_scale, B, H, W, C, k,
                           stride, pad, Ho, Wo, K, Kp);
    else
        hipLaunchKernelGGL((im2col_split_f16_kernel<1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)out, inv_scale, B, H,
                           W, C, k, stride, pad, Ho, Wo, K, Kp);
    PSALM_LAUNCH_END("psalm_im2col_split_f16");
}

// ---------------------------------------------------------------- image pre-processing
// The reference's eval-time input pipeline (coco_panoptic_mapper.py:60-91,134-163): detectron2 ResizeShortestEdge -> Pillow
// `Image.resize(..., BILINEAR)` on the uint8 image, FixedSizeCrop -> pad bottom/right with 128, then (x - mean) / std in fp32.
// Pillow's resampler (src/libImaging/Resample.c, 8 bits per channel) is a separable antialiasing filter in FIXED POINT:
//   coefficients  k = (int)(0.5 + w * 2^22)  (w: normalised triangle weights over a support that grows with the down-scale factor),
//   horizontal pass  t = clip8((2^21 + sum_x in[x] * k[x]) >> 22)  rounded to uint8, then the vertical pass the same way on t.
// Both passes are restated here with the same integer arithmetic, so the result is BIT-identical to Pillow (tests/test_8_preprocess.py);
// the coefficient tables (outSize x ksize int32 + bounds) are computed on the host exactly as precompute_coeffs / normalize_coeffs_8bpc
// do (double arithmetic) and ride in the call's blob.  A pass whose size does not change is skipped, as ImagingResample does.
//   resample_h: in (H, W, 3) u8 -> tmp (H, nw, 3) u8        resample_v_pad_norm: tmp (H, nw, 3) u8 -> out (3, S, S) f32 + padding mask
__global__ void __launch_bounds__(256) pil_resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ tmp,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                             int H, int W, int nw) {
    const long total = (long)H * nw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % nw), y = (int)(i / nw);
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (long)xx * ksize;
        const unsigned char* p = in + ((long)y * W + xmin) * 3;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int x = 0; x < xmax; ++x) {
            const int kv = k[x];
            s0 += p[3 * x] * kv; s1 += p[3 * x + 1] * kv; s2 += p[3 * x + 2] * kv;
        }
        unsigned char* o = tmp + i * 3;
        o[0] = (unsigned char)min(max(s0 >> 22, 0), 255);
        o[1] = (unsigned char)min(max(s1 >> 22, 0), 255);
        o[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
    }
}

__global__ void __launch_bounds__(256) pil_resample_v_pad_norm_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out,
                                                                      unsigned char* __restrict__ pad_mask, const int* __restrict__ bounds,
                                                                      const int* __restrict__ kk, int ksize, int Hin, int nh, int nw, int S,
                                                                      float m0, float m1, float m2, float d0, float d1, float d2,
                                                                      int vertical) {
    const long total = (long)S * S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % S), yy = (int)(i / S);
        int v0 = 128, v1 = 128, v2 = 128;                        // FixedSizeCrop pad value
        const bool inside = yy < nh && xx < nw;
        if (inside) {
            if (vertical) {
                const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
                const int* k = kk + (long)yy * ksize;
                int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
                for (int y = 0; y < ymax; ++y) {
                    const unsigned char* p = tmp + ((long)(ymin + y) * nw + xx) * 3;
                    const int kv = k[y];
                    s0 += p[0] * kv; s1 += p[1] * kv; s2 += p[2] * kv;
                }
                v0 = min(max(s0 >> 22, 0), 255); v1 = min(max(s1 >> 22, 0), 255); v2 = min(max(s2 >> 22, 0), 255);
            } else {
                const unsigned char* p = tmp + ((long)yy * nw + xx) * 3;
                v0 = p[0]; v1 = p[1]; v2 = p[2];
            }
        }
        out[i] = ((float)v0 - m0) / d0;                          // (image - pixel_mean) / pixel_std, coco_panoptic_mapper.py:158
        out[total + i] = ((float)v1 - m1) / d1;
        out[2 * total + i] = ((float)v2 - m2) / d2;
        pad_mask[i] = inside ? 0 : 1;
    }
}

// img (H,W,3) u8 RGB -> out (3,S,S) f32 normalised, pad_mask (S,S) u8 (1 = padding).  (nh, nw): the resized extent (<= S).
// bounds_h / kk_h: nw x {2, ksize_h} int32 (ignored when nw == W); bounds_v / kk_v: nh x {2, ksize_v} (ignored when nh == H);
// tmp: >= H * nw * 3 bytes of scratch (unused when nw == W).  mean / std: host arrays of 3.
extern "C" int psalm_image_preprocess(const unsigned char* img, int H, int W, float* out, unsigned char* pad_mask, int S, int nh, int nw,
                                      const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v,
                                      unsigned char* tmp, const float* mean3_host, const float* std3_host, void* stream) {
    PSALM_CHECK_ARG(H > 0 && W > 0 && nh > 0 && nw > 0 && nh <= S && nw <= S, "psalm_image_preprocess: bad geometry");
    const bool horiz = nw != W, vert = nh != H;
    PSALM_CHECK_ARG(!horiz || (bounds_h && kk_h && tmp && ksize_h > 0), "psalm_image_preprocess: horizontal tables / scratch missing");
    PSALM_CHECK_ARG(!vert || (bounds_v && kk_v && ksize_v > 0), "psalm_image_preprocess: vertical tables missing");
    hipStream_t s = (hipStream_t)stream;
    const unsigned char* src = img;
    if (horiz) {
        const long total = (long)H * nw;
        hipLaunchKernelGGL(pil_resample_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, tmp, bounds_h, kk_h, ksize_h, H, W, nw);
        src = tmp;
    }
    const long tot = (long)S * S;
    hipLaunchKernelGGL(pil_resample_v_pad_norm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, out, pad_mask, bounds_v, kk_v,
                       ksize_v, H, nh, nw, S, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], vert ? 1 : 0);
    PSALM_LAUNCH_END("psalm_image_preprocess");
}
