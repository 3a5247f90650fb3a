// Evaluator-facing outputs on the device: what the reference's evaluators compute on the HOST after pulling ~1 GB of
// fp32 masks per image across PCIe (`.cpu().numpy()` at psalm/eval/segmentation_evaluation/panoptic_evaluation.py:124-126,179-186,
// psalm/eval/referring_segmentation.py:120-122, region_segmentation.py:134-137), produced next to the data instead, in compact form:
//   psalm_semantic_labels      sem_seg (C,HW) f32 -> argmax labels (HW) i32                     panoptic_evaluation.py:125
//   psalm_confusion_accumulate conf[(C+1) pred + gt] += 1, gt == ignore -> C                    panoptic_evaluation.py:129-134
//   psalm_panoptic_rgb         panoptic ids (HW) i32 -> (HW,3) u8, panopticapi id2rgb           panoptic_evaluation.py:204
//   psalm_mask_rle_count/_emit binary masks (n,H,W) -> COCO RLE run boundaries (column-major)   region_segmentation.py:282 (mask.encode)
//   psalm_iou_counts           per (prediction, target) pair: intersection / output / target pixel counts of classes {0,1} with the
//                              target's 255 = ignore                                            referring_segmentation.py:101-113
// All integer work: results are bit-exact against the host formulas (tests/test_8_evalout.py).
#include "common.h"

// ---------------------------------------------------------------- semantic labels: argmax over the class planes (ties -> lowest index)
__global__ void __launch_bounds__(256) semantic_labels_kernel(const float* __restrict__ sem, int* __restrict__ labels, int C, long HW) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        float best = sem[i];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = sem[(long)c * HW + i];
            if (v > best || (best != best && v == v)) { best = v; bi = c; }      // torch.argmax: first maximal element (NaN-free inputs)
        }
        labels[i] = bi;
    }
}
extern "C" int psalm_semantic_labels(const float* sem, int* labels, int C, long HW, void* stream) {
    if (HW == 0) return 0;
    PSALM_CHECK_ARG(C >= 1, "psalm_semantic_labels: C >= 1");
    hipLaunchKernelGGL(semantic_labels_kernel, dim3((unsigned)((HW + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, sem, labels, C, HW);
    PSALM_LAUNCH_END("psalm_semantic_labels");
}

// ---------------------------------------------------------------- confusion matrix
// conf: (C+1) x (C+1) int64, row = prediction, column = ground truth (np.bincount((C+1) * pred + gt)).  Block-private LDS histogram
// (<= 36 K bins of u32 = 144 KB; COCO: 134^2 = 18 K), flushed with one 64-bit atomic per touched bin.
__global__ void __launch_bounds__(256) confusion_kernel(const int* __restrict__ pred, const int* __restrict__ gt, long n, int C, int ignore,
                                                        unsigned long long* __restrict__ conf) {
    HIP_DYNAMIC_SHARED(unsigned, hist)
    const int bins = (C + 1) * (C + 1);
    for (int b = threadIdx.x; b < bins; b += 256) hist[b] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int g = gt[i];
        if (g == ignore) g = C;
        const int p = pred[i];
        if (p >= 0 && p <= C && g >= 0 && g <= C) atomicAdd(&hist[(C + 1) * p + g], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += 256)
        if (hist[b]) atomicAdd(&conf[b], (unsigned long long)hist[b]);
}
extern "C" int psalm_confusion_accumulate(const int* pred, const int* gt, long n, int num_classes, int ignore_label, long long* conf,
                                          void* stream) {
    if (n == 0) return 0;
    const int bins = (num_classes + 1) * (num_classes + 1);
    PSALM_CHECK_ARG(num_classes >= 1 && bins * 4 <= 144 * 1024, "psalm_confusion_accumulate: (C+1)^2 bins must fit 144 KB of LDS (C <= 190)");
    const int grid = (int)((n + 256 * 64 - 1) / (256 * 64) < 1024 ? (n + 256 * 64 - 1) / (256 * 64) : 1024);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(256), (size_t)bins * 4, (hipStream_t)stream, pred, gt, n, num_classes, ignore_label,
                       (unsigned long long*)conf);
    PSALM_LAUNCH_END("psalm_confusion_accumulate");
}

// ---------------------------------------------------------------- panoptic id map -> RGB (panopticapi.utils.id2rgb)
__global__ void __launch_bounds__(256) panoptic_rgb_kernel(const int* __restrict__ ids, unsigned char* __restrict__ rgb, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned v = (unsigned)ids[i];
        rgb[3 * i] = (unsigned char)(v & 255u);
        rgb[3 * i + 1] = (unsigned char)((v >> 8) & 255u);
        rgb[3 * i + 2] = (unsigned char)((v >> 16) & 255u);
    }
}
extern "C" int psalm_panoptic_rgb(const int* ids, unsigned char* rgb, long n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(panoptic_rgb_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, ids, rgb, n);
    PSALM_LAUNCH_END("psalm_panoptic_rgb");
}

// ---------------------------------------------------------------- COCO RLE (pycocotools maskApi.c rleEncode): column-major runs
// A "boundary" is a position j = x*H + y (column-major) whose pixel differs from its predecessor (the pixel before position 0 is 0).
// Thread = one column of one mask, walking down the rows: a wavefront reads 64 adjacent columns of a row = one coalesced segment.
//   count: boundaries per column -> col_cnt (n, W)      scan: exclusive prefix per mask -> col_off (n, W), total (n)
//   emit : boundary positions, ascending, at out[base[i] + col_off[i][x] + k]
// The host turns boundaries into run lengths (differences) and into the COCO string -- proportional to the number of runs, not pixels.
template <typename T, bool EMIT>
__global__ void __launch_bounds__(256) mask_rle_kernel(const T* __restrict__ masks, int H, int W, int* __restrict__ col_cnt,
                                                       const int* __restrict__ col_off, const long* __restrict__ base, int* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (x >= W) return;
    const T* m = masks + (long)i * H * W;
    bool prev = x > 0 ? (m[(long)(H - 1) * W + x - 1] != (T)0) : false;
    int k = 0;
    int* o = EMIT ? out + base[i] + col_off[(long)i * W + x] : nullptr;
    for (int y = 0; y < H; ++y) {
        const bool v = m[(long)y * W + x] != (T)0;
        if (v != prev) {
            if (EMIT) o[k] = x * H + y;
            ++k;
            prev = v;
        }
    }
    if (!EMIT) col_cnt[(long)i * W + x] = k;
}
__global__ void __launch_bounds__(256) mask_rle_scan_kernel(const int* __restrict__ col_cnt, int* __restrict__ col_off, int* __restrict__ total, int W) {
    __shared__ int part[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int per = (W + 255) / 256, x0 = tid * per;
    int s = 0;
    for (int x = x0; x < min(W, x0 + per); ++x) s += col_cnt[(long)i * W + x];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
        total[i] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int x = x0; x < min(W, x0 + per); ++x) { col_off[(long)i * W + x] = run; run += col_cnt[(long)i * W + x]; }
}
extern "C" int psalm_mask_rle_count(const void* masks, int dtype_is_u8, int n, int H, int W, int* col_cnt, int* col_off, int* total, void* stream) {
    if (n == 0 || H == 0 || W == 0) return 0;
    const dim3 grid(cdiv(W, 256), n);
    if (dtype_is_u8) hipLaunchKernelGGL((mask_rle_kernel<unsigned char, false>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)masks, H, W, col_cnt, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((mask_rle_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)masks, H, W, col_cnt, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(mask_rle_scan_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, col_cnt, col_off, total, W);
    PSALM_LAUNCH_END("psalm_mask_rle_count");
}
extern "C" int psalm_mask_rle_emit(const void* masks, int dtype_is_u8, int n, int H, int W, const int* col_off, const long* base, int* out, void* stream) {
    if (n == 0 || H == 0 || W == 0) return 0;
    const dim3 grid(cdiv(W, 256), n);
    if (dtype_is_u8) hipLaunchKernelGGL((mask_rle_kernel<unsigned char, true>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)masks, H, W, nullptr, col_off, base, out);
    else hipLaunchKernelGGL((mask_rle_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)masks, H, W, nullptr, col_off, base, out);
    PSALM_LAUNCH_END("psalm_mask_rle_emit");
}

// ---------------------------------------------------------------- intersection / union counts (intersectionAndUnionGPU, K = 2)
// pair p: prediction mask pred[pred_idx[p]] (HW, nonzero = 1) vs target tgt[tgt_idx[p]] (HW u8, 255 = ignore).  counts (npairs, 6) i64 =
// [I0, I1, O0, O1, T0, T1] over the non-ignored pixels (I: pred == tgt == c, O: pred == c, T: tgt == c); union = O + T - I.
template <typename T>
__global__ void __launch_bounds__(256) iou_counts_kernel(const T* __restrict__ pred, const unsigned char* __restrict__ tgt, const int* __restrict__ pred_idx,
                                                         const int* __restrict__ tgt_idx, long HW, unsigned long long* __restrict__ counts) {
    __shared__ unsigned long long red[4][6];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* pm = pred + (long)pred_idx[p] * HW;
    const unsigned char* tm = tgt + (long)tgt_idx[p] * HW;
    unsigned c[6] = {0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + tid; i < HW; i += (long)gridDim.x * 256) {
        const unsigned t = tm[i];
        if (t == 255u) continue;
        const unsigned o = pm[i] != (T)0 ? 1u : 0u;
        if (o == t) c[o] += 1;
        c[2 + o] += 1;
        if (t < 2u) c[4 + t] += 1;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float f = wave_sum((float)c[k]);                          // per-thread counts < 2^24 / 64: exact in fp32
        if (lane == 0) red[wave][k] = (unsigned long long)f;
    }
    __syncthreads();
    if (tid < 6) atomicAdd(&counts[(long)p * 6 + tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}
extern "C" int psalm_iou_counts(const void* pred, int pred_is_u8, const unsigned char* tgt, const int* pred_idx, const int* tgt_idx, int npairs,
                                long HW, long long* counts_zeroed, void* stream) {
    if (npairs == 0 || HW == 0) return 0;
    const int gx = (int)((HW + 256 * 256 - 1) / (256 * 256));   // <= 256 pixels per thread: per-thread counts stay far below 2^18
    const dim3 grid(gx, npairs);
    if (pred_is_u8) hipLaunchKernelGGL((iou_counts_kernel<unsigned char>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)pred, tgt, pred_idx, tgt_idx, HW, (unsigned long long*)counts_zeroed);
    else hipLaunchKernelGGL((iou_counts_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)pred, tgt, pred_idx, tgt_idx, HW, (unsigned long long*)counts_zeroed);
    PSALM_LAUNCH_END("psalm_iou_counts");
}

// ---------------------------------------------------------------- gRefCOCO: the fused prediction of compute_metric (eval_grefcoco.py:113-131)
// out[p] = OR over the candidates i with scores[i] > thr of [masks[i][p] set]; when NO candidate passes the threshold, the single candidate
// with the largest score -- the reference's "no candidate -> top-1" fall-back.  The selection is re-derived by every block from the <= 1024
// scores (one thread, a few hundred compares) instead of a host round trip.
//   "set": a uint8 mask element != 0; a float mask element after the reference's `preds.astype(np.uint8)` (compute_metric, eval_grefcoco.py:116):
//   truncated toward zero first, so fractional values in (-1, 1) are NOT set (`!= 0.f` counted them); non-finite values are not set.
//   Ties / NaN in the fall-back: the FIRST maximal score wins (strict `>` walk from index 0; torch.topk does not promise an order among equal
//   scores); a NaN score never compares greater, so all-NaN scores select candidate 0.
__global__ void __launch_bounds__(256) fuse_masks_kernel(const void* __restrict__ masks, int is_u8, const float* __restrict__ scores, int n, long HW,
                                                         float thr, unsigned char* __restrict__ out) {
    __shared__ int sel[1024];
    __shared__ int nsel_s;
    if (threadIdx.x == 0) {
        int k = 0, best = 0;
        for (int i = 0; i < n; ++i) {
            if (scores[i] > thr) sel[k++] = i;
            if (scores[i] > scores[best]) best = i;
        }
        if (k == 0 && n > 0) sel[k++] = best;
        nsel_s = k;
    }
    __syncthreads();
    const int nsel = nsel_s;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        unsigned v = 0;
        for (int k = 0; k < nsel; ++k) {
            const long idx = (long)sel[k] * HW + p;
            v |= is_u8 ? (unsigned)(reinterpret_cast<const unsigned char*>(masks)[idx] != 0) : (unsigned)(((int)reinterpret_cast<const float*>(masks)[idx] & 0xff) != 0 && fabsf(reinterpret_cast<const float*>(masks)[idx]) < 3.0e38f);
        }
        out[p] = (unsigned char)v;
    }
}
extern "C" int psalm_fuse_masks(const void* masks, int dtype_is_u8, const float* scores, int n, long HW, float thr, unsigned char* out, void* stream) {
    if (HW == 0) return 0;
    PSALM_CHECK_ARG(n >= 1 && n <= 1024, "psalm_fuse_masks: 1 <= n <= 1024 candidates");
    const long gx = (HW + 1023) / 1024;
    hipLaunchKernelGGL(fuse_masks_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream, masks, dtype_is_u8, scores, n, HW, thr, out);
    PSALM_LAUNCH_END("psalm_fuse_masks");
}
