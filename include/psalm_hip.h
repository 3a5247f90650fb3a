/* C ABI of libpsalm_hip.so -- hand-written gfx950 (MI355X / CDNA4) kernels for the PSALM
 * segmentation-inference path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - dtype codes: PSALM_F32 = 0 (float), PSALM_BF16 = 1 (bfloat16 bits);
 *   - every entry point launches asynchronously on `stream` (a hipStream_t), never allocates,
 *     never synchronises, and returns 0 on success; on failure it returns non-zero and
 *     psalm_last_error() describes why (argument validation or launch failure);
 *   - tensors are dense row-major in the shapes written next to each argument.
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference repo;
 * OPS = psalm/model/mask_decoder/Mask2Former_Simplify/modeling/pixel_decoder/ops).
 */
#ifndef PSALM_HIP_H
#define PSALM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PSALM_F32 0
#define PSALM_BF16 1

const char* psalm_last_error(void);
/* Version of this binary interface: bumped whenever an entry point's arguments change.  A binding MUST compare it with the constant it was
 * written against before making any other call (psalm_amd/hip_ops.py does): a stale library loaded by a newer binding would otherwise take
 * integers for pointers.   4: the e4m3 cross-term ("x8") operand form and its `form` / `x8` / `split_form` arguments are gone (r04);
 * 5: process-wide policy getenv()s replaced by nothing (PSALM_ATTN_PAIR, PSALM_SEM_ORDER, PSALM_MSDA_LINEAR are gone).
 * 6: psalm_gemm_x3_set_products, psalm_fuse_masks, the stage-level psalm_phi_forward (r05); psalm_causal_attention_f32_workspace grew by one
 *    byte per 32-key tile.
 * 7: psalm_set_tuning / psalm_get_tuning, psalm_layernorm_chain, psalm_gemm_f32_pair, psalm_postprocess*, psalm_predictor_kv + the kv_ready argument of
 *    psalm_predictor_forward (r06). */
#define PSALM_ABI_VERSION 7
int psalm_abi_version(void);
const char* psalm_backend(void); /* "hip-gfx950" */

/* Process-wide tuning switches, each an atomic word (safe to flip while other host threads launch: their next launch sees the old or the new
 * value).  Every switch chooses between kernel forms whose results are identical word for word; they exist for A/B measurements and tests.
 *   PSALM_TUNE_GEMM_XCD_KSPLIT  1 (default): in split-K launches of the direct-to-LDS GEMM the K slice decides a block's XCD (each operand slice
 *                               is fetched into one XCD's L2); 0: the r01-r05 placement (a run of tiles per XCD, every XCD walks the whole K range)
 *   PSALM_TUNE_ATTN_XCD_HEADS   1 (default): psalm_causal_attention_f32* places all query-tile blocks of a head on one XCD (4 heads' K / V per L2);
 *                               0: the r02-r05 placement (a head's blocks spread over all eight)
 *   PSALM_TUNE_GEMM_MID         1 (default): mid-size split-f16 GEMMs take the 64 x 64 wave-tile kernel where the selection prefers it; 0: r05 kernels
 *   PSALM_TUNE_DECODER_FUSE     1 (default): psalm_predictor_forward issues the query rows' LayerNorm chains / paired projections as single launches
 *                               (psalm_layernorm_chain, psalm_gemm_f32_pair); 0: the r05 launch sequence
 *   PSALM_TUNE_ROW_GROUPS       1 (default): the LayerNorm-fused row kernels (psalm_layernorm_split, psalm_swin_window_gather_split,
 *                               psalm_swin_window_merge_ln_split) put 4 / 2 rows of <= 128 / 256 columns on one wavefront, psalm_patch_merge_ln keeps
 *                               an fp32 row of 512 / 1024 / 2048 values in registers, psalm_im2col_split_f16 gives few long rows a block each;
 *                               0: the r01-r05 forms (one row per wavefront)
 *   5..7                        unused */
#define PSALM_TUNE_GEMM_XCD_KSPLIT 0
#define PSALM_TUNE_ATTN_XCD_HEADS 1
#define PSALM_TUNE_GEMM_MID 2
#define PSALM_TUNE_DECODER_FUSE 3
#define PSALM_TUNE_ROW_GROUPS 4
#define PSALM_TUNE_COUNT 8
int psalm_set_tuning(int key, int value);
int psalm_get_tuning(int key);

/* Replaces MSDA.ms_deform_attn_forward, the reference's own native-op seam:
 *   OPS/src/vision.cpp:18-21 (pybind), OPS/src/ms_deform_attn.h:25-44 (dispatch),
 *   OPS/src/cuda/ms_deform_attn_cuda.cu:25-85 (host), OPS/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (kernel).
 * value (B,S,M,D) value_dtype; spatial_shapes_host (L,2) int64 (H,W); level_start_host (L) int64;
 * sampling_loc (B,Lq,M,L,P,2) f32 (x,y) in [0,1]; attn_weight (B,Lq,M,L,P) f32; out (B,Lq,M*D) out_dtype.
 * D % 4 == 0.  The reference's im2col_step batching is unnecessary (one launch covers the batch). */
int psalm_msda_forward(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                       const int64_t* level_start_host, const float* sampling_loc, const float* attn_weight, void* out,
                       int out_dtype, int B, int S, int M, int D, int L, int Lq, int P, void* stream);
/* Same op with the level table in DEVICE memory -- the reference's own convention (spatial_shapes / level_start_index are CUDA int64
 * tensors, ms_deform_attn.h:25-44; read by the kernel, ms_deform_im2col_cuda.cuh:261-265): fully asynchronous, no host copy.  This is
 * the entry the `MultiScaleDeformableAttention` plugin module binds. */
int psalm_msda_forward_dev(const void* value, int value_dtype, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev,
                           const float* sampling_loc, const float* attn_weight, void* out, int out_dtype, int B, int S, int M, int D,
                           int L, int Lq, int P, void* stream);


/* The same gather with MSDeformAttn.forward's location/softmax arithmetic fused in
 * (OPS/modules/ms_deform_attn.py:101-110): offsets_logits (B,S,M*L*P*3) f32 is the output of the
 * concatenated [sampling_offsets ; attention_weights] projection of (src+pos); queries are the
 * S pixels themselves (encoder self-attention, msdeformattn.py:76-87 reference points). L=3, P=4. */
int psalm_msda_fused(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                     const int64_t* level_start_host, const float* offsets_logits, void* out, int out_dtype, int B, int S,
                     int M, int D, int L, int P, void* stream);
/* Tuning / test knob of psalm_msda_fused (head dim 32): 1 (default) = the bilinear taps of a sample are computed once per (query, head) and
 * shared by its 4 channel-group lanes, 0 = every lane computes all samples (the r01 - r03 kernel).  The shared form is only selected for a bf16
 * `value` (fp32 value keeps the per-lane form whatever the knob says) and sums the softmax denominator in another order: equal to ~1e-5
 * relative, not bit for bit (tests/test_3_msda.py).  Like psalm_gemm_set_tile_policy a process-wide tuning / test knob: do not call it
 * while another host thread is launching. */
int psalm_msda_set_policy(int v);


/* ------------------------------------------------------------------------------------------------------------------
 * Dense contractions (every nn.Linear / 1x1 conv / im2col'd conv / einsum of the path).
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]   (nn.Linear weight layout, W is [out,in]).
 *   act: 0 none, 1 relu, 2 gelu(erf), 3 gelu_new(tanh); |16 = apply after the residual add (ResNet block);
 *   |32 = bias is indexed by the output row (bias[M]) instead of the column -- transposed projections C = W . X^T;
 *   activation applies to columns >= act_col_start only (fused [k|v|q|fc1] projection of a Phi layer).
 *   w_dtype selects the arithmetic: BF16 -> v_mfma_f32_32x32x16_bf16 / fp32 accumulate (A f32 or bf16, converted
 *   while staging); F32 -> v_mfma_f32_32x32x2_f32 (exact fp32; A must be f32).  lda/ldw/ldr/ldc are row strides in
 *   elements; K % 8 == 0; rows 16-byte aligned.  A and W both bf16 with K % 64 == 0 take the direct-to-LDS
 *   (global_load_lds_dwordx4, XOR-swizzled LDS image) kernel; `workspace` (workspace_bytes, may be NULL/0) is caller-owned
 *   scratch for split-K partials, used when the tile grid alone cannot fill the 256 CUs.
 * Replaces torch.nn.functional.linear / conv2d at: modeling_phi.py:189-260 (q/k/v/dense/fc1/fc2),
 * swin_trans.py:28-34,109-149,266-296, multimodal_projector/builder.py:85-111,365-375, msdeformattn.py:196-254,
 * OPS/modules/ms_deform_attn.py:98-123, mask2former_transformer_decoder.py:187-199,709,723,744,749,
 * llava_phi.py:163,183-185 (projectors), llava_phi.py:402-406 (semantic einsum). */
int psalm_gemm(const void* A, int a_dtype, long lda, const void* W, int w_dtype, long ldw, const float* bias,
               const void* residual, long ldr, void* C, int c_dtype, long ldc, int M, int N, int K, int act,
               int act_col_start, void* workspace, long workspace_bytes, void* stream);

/* psalm_gemm + LayerNorm over the N columns of its fp32 result (ln_out = LN(C)*gamma+beta, ln_dtype, row stride ld_ln): with
 * split-K the LayerNorm is fused into the slab reduction.  bf16 A / W, K % 64 == 0, c_dtype F32, N % 4 == 0, N <= 8192.
 * Replaces the residual projection + the NEXT layer's input_layernorm of a Phi layer (modeling_phi.py:263-300). */
int psalm_gemm_ln(const void* A, int a_dtype, long lda, const void* W, int w_dtype, long ldw, const float* bias,
                  const void* residual, long ldr, void* C, int c_dtype, long ldc, int M, int N, int K, int act, int act_col_start,
                  const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, int ln_dtype, long ld_ln, void* workspace,
                  long workspace_bytes, void* stream);
/* Convolution as an implicit GEMM on the direct-to-LDS kernel (no im2col matrix in HBM): x (B,H,W,Cin) bf16 NHWC,
 * Wt (Cout, k*k*Cin) bf16 with K order (ky,kx,c); out / residual (B*Ho*Wo, Cout); Cin % 64 == 0; `zeros` = >= 16 zero bytes
 * on the device (source of the padded taps).  act as psalm_gemm.  Replaces F.conv2d at
 * multimodal_projector/builder.py:85-111 and msdeformattn.py:248-254. */
int psalm_conv2d_nhwc(const void* x, int B, int H, int W, int Cin, const void* Wt, int Cout, int ksize, int stride, int pad,
                      const float* bias, const void* residual, long ldr, void* out, int c_dtype, long ldc, int act,
                      const void* zeros, void* workspace, long workspace_bytes, void* stream);
/* Split-f16 ("X3") fp32-class GEMM on the f16 matrix cores -- the arithmetic of precision="f16x3", the mode that meets the reference's
 * fp32 results (torch fp32 nn.Linear / F.conv2d / einsum on the CPU path: llava_phi.py:1350-1398 and everything below it) to the north
 * star's tolerance at ~1/3 of the f16 MFMA rate instead of the fp32 MFMA's 1/16.
 * psalm_split_f16: x (rows,K) f32 -> out (rows, 2*Kp) f16 = [hi | lo], Kp = ceil64(K), x*s = hi + lo with s a per-row power of two
 *   (row max in [2^13,2^14)); inv_scale[row] = 1/s.   K % 8 == 0, 16-byte aligned rows.
 * psalm_gemm_x3: C = act(A.W^T + bias) + residual, C / residual fp32, from split operands A2 (M,2Kp) / W2 (N,2Kp) + their scales:
 *   one f16 GEMM over the 3*Kp-long panel hi.hi + lo.hi + hi.lo (or, slice by slice, the same three products), fp32 accumulate, scales in
 *   the epilogue; tiles / split-K as psalm_gemm. */
int psalm_split_f16(const float* x, long ldx, void* out, long ldo, float* inv_scale, int rows, int K, void* stream);
/* Products formed per algorithmic product by the psalm_gemm_x3* calls of the CALLING host thread: 3 (default; hi.hi + lo.hi + hi.lo, the
 * fp32-class arithmetic) or 1 (hi.hi only = f16 operands with 11-bit mantissas under the same row scales, a third of the matrix work) -- the
 * reduced-precision LLM side mode of BASELINE.json configs[4]; the reference has no counterpart (fp32, psalm/eval/panoptic_segmentation.py:126-127).
 * Results under 1 do NOT meet the fp32 parity bar.  Thread-local; read at launch (a captured graph keeps what it was captured with). */
int psalm_gemm_x3_set_products(int n);
/* im2col (K order ky,kx,c; as psalm_im2col_nhwc) emitted directly in split form -- the convolution-as-GEMM A operand of the f16x3 mode
 * (F.conv2d at multimodal_projector/builder.py:85-111 and msdeformattn.py:248-254) without the fp32 im2col matrix in HBM. */
int psalm_im2col_split_f16(const float* x, void* out, float* inv_scale, int B, int H, int W, int C, int k, int stride, int pad, void* stream);
/* LayerNorm whose result leaves in split-f16 form (the next GEMM's A operand in the f16x3 mode; nn.LayerNorm at modeling_phi.py:263-300,
 * msdeformattn.py:57-66): y = LN(x) fp32 (optional) + split(y) (optional) + split(y + add[row % add_rows]) (optional), each split as
 * psalm_split_f16 writes it (rows of 2*ceil64(C) f16 + inv_scale).  fp32 in, C % 8 == 0, C <= 2048. */
int psalm_layernorm_split(const float* x, long ldx, float* y, long ldy, const float* gamma, const float* beta, int rows, int C, float eps,
                          void* split1, float* inv1, const float* add, long add_rows, void* split2, float* inv2, void* stream);
/* f16x3 forms of the two Swin LayerNorm-fused data-movement steps (swin_trans.py:206-227 norm1 + pad + shift + window partition;
 * :235-251 window reverse + un-shift + residual, then norm2): the normalised rows leave as the split-f16 A operand of the GEMM they feed. */
int psalm_swin_window_gather_split(const float* x, void* out, float* inv_out, const float* gamma, const float* beta, int B, int H, int W, int C,
                                   int ws, int shift, float eps, void* stream);
int psalm_swin_window_merge_ln_split(const float* win, const float* shortcut, float* out_x, void* h_split, float* h_inv, const float* gamma,
                                     const float* beta, int B, int H, int W, int C, int ws, int shift, float eps, void* stream);
int psalm_gemm_x3(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                  const float* bias, const void* residual, long ldr, void* C, long ldc, int M, int N, int act, int act_col_start,
                  void* workspace, long workspace_bytes, void* stream);
/* psalm_gemm_x3 whose output columns >= split_col_start leave the kernel already in split form -- the A operand of the NEXT split-f16 GEMM
 * (nn.Linear -> activation -> nn.Linear chains: Swin Mlp swin_trans.py:35-53, Phi fc1 -> fc2 modeling_phi.py:248-260, the encoder FFN
 * msdeformattn.py:68-72) -- without the fp32 round trip and the psalm_split_f16 pass.  Value (r, n) goes to row r of split_out (row stride
 * ld_split f16) at column split_col_off + (n - split_col_start) (hi) and split_kp columns further (lo).  The per-row power-of-two scale is
 * derived from a magnitude BOUND that needs no pass over the output: bound_par = 4 device floats {2^14 * max_n sum_k |w_nk|, max |bias|,
 * g1, g0};  bound_r = max(a_scale[r] * par[0] + par[1], (global_rows ? max_r a_scale[r] : 0) * g1 + g0)  (the second term: the caller's
 * bound for values ANOTHER kernel writes into the same rows, see psalm_causal_attention_f32_split);  bound_r * scale in [2^12, 2^13);
 * 1/scale -> split_inv[r].  Columns < split_col_start go to C (fp32) as in psalm_gemm_x3.  No residual, no split-K; N and the column
 * arguments are multiples of 8.  Columns of split_out this call does not write (K padding) are the caller's to zero.
 * paired != 0 = PAIRED stores: the caller has permuted the W rows (with their w_scale and
 * bias entries) >= split_col_start inside every group of 64 -- physical row 64 g + 32 b + n holds logical row 64 g + 2 n + b -- so that a
 * lane of the accumulator layout owns two ADJACENT output columns and the operand leaves in 4-byte stores of whole 128-byte row segments
 * straight from the registers (no LDS transpose).  Output identical bit for bit.  Needs split_col_start % 256 == 0,
 * (N - split_col_start) % 64 == 0, split_out < 2 GiB. */
/* x = A.W^T + bias + residual (fp32 -> C) followed by LayerNorm(x) leaving as the split-f16 operand of the next GEMM (and optionally as
 * fp32 rows ln_out): the residual GEMM + the following block's input LayerNorm of a pre-norm layer (Phi [dense | fc2] + residual, then the
 * next input_layernorm, modeling_phi.py:263-300).  With split-K the partial-sum reduce, epilogue, LayerNorm and split are ONE row pass.
 * N % 64 == 0, N <= 2048; split_out rows of 2*N f16 as psalm_split_f16 writes them (exact row-maximum scale), split_inv (M). */
int psalm_gemm_x3_ln_split(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                           const float* bias, const void* residual, long ldr, void* C, long ldc, int M, int N, const float* ln_gamma,
                           const float* ln_beta, float ln_eps, void* ln_out, long ld_ln, void* split_out, float* split_inv,
                           void* workspace, long workspace_bytes, void* stream);
int psalm_gemm_x3_split(const void* A2, long lda, const float* a_scale, const void* W2, long ldw, const float* w_scale, int Kp,
                        const float* bias, void* C, long ldc, int M, int N, int act, int act_col_start, void* split_out, long ld_split,
                        int split_kp, int split_col_off, int split_col_start, int paired, float* split_inv, const float* bound_par,
                        int global_rows, void* workspace, long workspace_bytes, void* stream);

/* Eval-time image pre-processing on the device: replaces detectron2 `T.ResizeShortestEdge` (= Pillow
 * `Image.resize(BILINEAR)` on uint8) + `T.FixedSizeCrop` + `(image - pixel_mean) / pixel_std` of
 * psalm/model/datasets_mapper/coco_panoptic_mapper.py:60-91,134-163.  img (H,W,3) u8 RGB -> out (3,S,S) f32, pad_mask (S,S) u8 (1 = pad).
 * (nh,nw): resized extent; bounds_* / kk_*: Pillow's fixed-point coefficient tables per output column / row (psalm_amd/preprocess.py),
 * NULL for an axis whose size does not change; tmp: H*nw*3 bytes of scratch; mean / std: HOST arrays of 3.  Bit-identical to Pillow. */
int psalm_image_preprocess(const unsigned char* img, int H, int W, float* out, unsigned char* pad_mask, int S, int nh, int nw,
                           const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v,
                           unsigned char* tmp, const float* mean3_host, const float* std3_host, void* stream);

/* Evaluator-facing outputs on the device (csrc/evalout.hip): the host arithmetic of the reference's evaluators moved next
 * to the data, so that compact results cross PCIe instead of ~1 GB of fp32 masks per image.
 *   psalm_semantic_labels      `output["sem_seg"].argmax(dim=0)`                      panoptic_evaluation.py:125   sem (C,HW) f32 -> labels (HW) i32
 *   psalm_confusion_accumulate `np.bincount((C+1)*pred + gt)`, gt == ignore -> C      panoptic_evaluation.py:129-134  conf (C+1,C+1) i64 += ...
 *   psalm_panoptic_rgb         panopticapi `id2rgb(panoptic_img)`                     panoptic_evaluation.py:204   ids (HW) i32 -> (HW,3) u8
 *   psalm_mask_rle_count/_emit pycocotools `mask.encode(np.asfortranarray(m))` run boundaries (column-major positions where the pixel
 *                              differs from its predecessor)                           region_segmentation.py:282   masks (n,H,W) f32|u8
 *       count: col_cnt / col_off (n,W) i32 scratch, total (n) i32;  emit: base (n) i64 = exclusive prefix of total (host), out i32
 *   psalm_iou_counts           intersectionAndUnionGPU(pred, gt, K=2, ignore 255)      referring_segmentation.py:101-113
 *       per pair p: counts[p] = [I0,I1,O0,O1,T0,T1] i64 (accumulated: zero the buffer first); union = O + T - I                        */
int psalm_semantic_labels(const float* sem, int* labels, int C, long HW, void* stream);
int psalm_confusion_accumulate(const int* pred, const int* gt, long n, int num_classes, int ignore_label, long long* conf, void* stream);
int psalm_panoptic_rgb(const int* ids, unsigned char* rgb, long n, void* stream);
int psalm_mask_rle_count(const void* masks, int dtype_is_u8, int n, int H, int W, int* col_cnt, int* col_off, int* total, void* stream);
int psalm_mask_rle_emit(const void* masks, int dtype_is_u8, int n, int H, int W, const int* col_off, const long* base, int* out, void* stream);
int psalm_iou_counts(const void* pred, int pred_is_u8, const unsigned char* tgt, const int* pred_idx, const int* tgt_idx, int npairs, long HW,
                     long long* counts_zeroed, void* stream);
/*   psalm_fuse_masks           gRefCOCO's fused prediction (psalm/eval/eval_grefcoco.py:113-131 compute_metric + :277-285 fuse_masks): the union of
 *                              the candidate masks whose score exceeds thr; none above thr -> the top-1 candidate (torch.topk(scores, 1)).
 *       masks (n,H,W) f32|u8 (nonzero = 1), scores (n) f32, n <= 1024 -> out (HW) u8 */
int psalm_fuse_masks(const void* masks, int dtype_is_u8, const float* scores, int n, long HW, float thr, unsigned char* out, void* stream);

/* fp32 matrix-core MHA core of the predictor (nn.MultiheadAttention, mask2former_transformer_decoder.py:645-666; bool mask with the
 * all-masked-row rule :647) for the fp32 / f16x3 modes: exact-fp32 products (v_mfma_f32_16x16x4_f32), split over 64..256-key chunks.
 * Same operands as psalm_mha_attention (fp32, row strides in elements, head_dim 32, Lq <= 128) + a workspace of
 * psalm_mha_attention_f32_workspace(B, heads, Lq, Lk) bytes (0 when the keys fit one chunk). */
long psalm_mha_attention_f32_workspace(int B, int heads, int Lq, int Lk);
int psalm_mha_attention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* out, long ldo,
                            const unsigned char* mask, const unsigned char* row_all_masked, void* workspace, int B, int Lq, int Lk, int heads,
                            int head_dim, void* stream);

/* Phi prefill attention (modeling_phi.py:189-245; eager softmax :137-160; partial RoPE :92-122) for fp32 buffers on the fp32 matrix cores,
 * split over keys inside a block (fp32 / f16x3 modes).  Operands as psalm_causal_attention + a 16-byte aligned workspace of
 * psalm_causal_attention_f32_workspace(B, L, heads) bytes (RoPE'd Q / K and the padded key mask). */
long psalm_causal_attention_f32_workspace(int B, int L, int heads);
int psalm_causal_attention_f32(const float* qkv, long ld, int q_off, int k_off, int v_off, float* out, long ldo, int o_off,
                               const float* cos_table, const float* sin_table, const unsigned char* key_mask, void* workspace, int B, int L,
                               int heads, int head_dim, int rot, void* stream);
/* ... whose output leaves as split-f16 operand columns of the next GEMM (Phi [dense | fc2]): row r of split_out (row stride ld_split f16)
 * gets hi at columns split_col_off + head*64 + d and lo split_kp columns further, scaled by 1 / split_inv[r] -- the row scales a preceding
 * psalm_gemm_x3_split wrote for the same rows (its bound has to cover |v|: the output is a convex combination of v rows). */
int psalm_causal_attention_f32_split(const float* qkv, long ld, int q_off, int k_off, int v_off, void* split_out, long ld_split, int split_kp,
                                     int split_col_off, const float* split_inv, const float* cos_table, const float* sin_table,
                                     const unsigned char* key_mask, void* workspace, int B, int L, int heads, int head_dim, int rot,
                                     void* stream);

/* zero `bytes` bytes / copy `bytes` bytes device-to-device, as stream operations (hipMemsetAsync / hipMemcpyAsync; graph-capturable) */
int psalm_memset_zero(void* p, long bytes, void* stream);
int psalm_copy_d2d(void* dst, const void* src, long bytes, void* stream);
/* dst[i] = (int64) src[i]: the LongTensor label / query-index vectors of `Instances` (llava_phi.py:317-323, 428-447) */
int psalm_cast_i32_i64(const int* src, long long* dst, long n, void* stream);
/* Template instantiation (as a kernel trace names it, e.g. "gemm_bf16_glds_kernel<float, 256, 256, 2, 4, 2, false, 64, 3, 3, true>") of the
 * calling thread's most recent GEMM launch through psalm_gemm / psalm_gemm_x3* / psalm_conv2d_nhwc / psalm_gemm_ln. */
const char* psalm_gemm_last_kernel(void);
/* Which kernel psalm_gemm launches for a problem size: out4 = {path (0 register-staged, 1 direct-to-LDS), BM, BN, split-K slices}. */
int psalm_gemm_describe(int M, int N, int K, int a_dtype, int w_dtype, long workspace_bytes, int* out4);
/* Tuning / test knob for the direct-to-LDS path: 0 = automatic tile selection (default), 256 | 128 | 64 = force BM;
 * 2560 | 2568 | 2569 | 2570 = K loop of the 256x256 configuration: plain 2-buffer loop | 4-phase schedule, copy placement 1 | 2 | 3
 * (3 = default).  Other codes: experiment switches documented at psalm_gemm_set_tile_policy in csrc/gemm.hip.  Process-wide and NOT
 * synchronised: a tuning / test knob, to be set before launches start (tests restore the defaults); never call it while another host thread
 * (a PSALM.replica() worker) is inside a GEMM call. */
int psalm_gemm_set_tile_policy(int bm);

/* ------------------------------------------------------------------------------------------------------------------
 * Row / normalisation kernels (one 64-lane wavefront per row, fp32 statistics). */
/* nn.LayerNorm over the last dim (swin_trans.py:181,187,548; modeling_phi.py:263-300; msdeformattn.py:37,45;
 * mask2former_transformer_decoder.py:19,77,143,451).  x (rows,C) row stride ldx; y (rows,C) row stride ldy; y2_bf16
 * (optional) receives a second bf16 copy of the result (the next GEMM's A operand beside a fp32 residual stream). */
int psalm_layernorm(const void* x, int x_dtype, long ldx, void* y, int y_dtype, long ldy, void* y2_bf16, long ldy2,
                    const float* gamma, const float* beta, int rows, int C, float eps, void* stream);
/* ... with a third output y3_bf16 = result + add[row % add_rows] (add (add_rows,C) f32): the "tensor + positional embedding"
 * operand of the next attention projection (msdeformattn.py:51-58; mask2former_transformer_decoder.py:35-37,93-96). */
int psalm_layernorm3(const void* x, int x_dtype, long ldx, void* y, int y_dtype, long ldy, void* y2_bf16, long ldy2, const float* add,
                     long add_rows, void* y3_bf16, long ldy3, const float* gamma, const float* beta, int rows, int C, float eps,
                     void* stream);
/* LayerNorm chain of the mask decoder's query rows, fp32 (mask2former_transformer_decoder.py:72-74 / :170-172 post-norms, :35-37 `output + query_pos`,
 * :750 decoder_norm of forward_prediction_heads):  y1 = LN(x; g1, b1);  y2 = y1 + add[row % add_rows] (y2 may be NULL);  y3 = LN(y1; g2, b2) (y3 may be
 * NULL).  The same words as psalm_layernorm3 + psalm_add_bcast + psalm_layernorm3.  C % 8 == 0, C <= 2048, 16-byte aligned rows. */
int psalm_layernorm_chain(const float* x, long ldx, float* y1, long ldy1, const float* g1, const float* b1, const float* add, long add_rows,
                          float* y2, long ldy2, const float* g2, const float* b2, float* y3, long ldy3, int rows, int C, float eps, void* stream);
/* Two exact-fp32 skinny GEMMs C_i = act_i(A_i . W_i^T + bias_i) (contiguous rows; M <= 192, N <= 8192, K % 8 == 0, both K below 256 or both from 256) in one launch: the mask
 * decoder's self-attention projections [q | k] = (x + pos) . Wqk^T and v = x . Wv^T (mask2former_transformer_decoder.py:24-38).  The same words as two
 * psalm_gemm calls with float32 operands. */
int psalm_gemm_f32_pair(const float* A0, const float* W0, const float* bias0, float* C0, int M0, int N0, int K0, int act0, const float* A1,
                        const float* W1, const float* bias1, float* C1, int M1, int N1, int K1, int act1, void* stream);
/* SwinTransformerBlock.forward front half (swin_trans.py:206-225): norm1 -> zero-pad to a multiple of ws ->
 * roll(-shift) -> window_partition.  x (B*H*W,C) -> out (B*nW*ws*ws, C). */
int psalm_swin_window_gather(const void* x, int x_dtype, void* out, int out_dtype, const float* gamma,
                             const float* beta, int B, int H, int W, int C, int ws, int shift, float eps, void* stream);
/* ... and its back half (swin_trans.py:233-250): window_reverse -> roll(+shift) -> crop -> + shortcut. */
int psalm_swin_window_merge(const void* win, int win_dtype, const void* shortcut, void* out, int x_dtype, int B, int H,
                            int W, int C, int ws, int shift, void* stream);
/* ... fused with the block's norm2: out_x = shortcut + merged (fp32 residual stream), out_h = LayerNorm(out_x) in h_dtype
 * (A operand of the MLP's fc1).  C % 8 == 0, C <= 2048. */
int psalm_swin_window_merge_ln(const void* win, int win_dtype, const float* shortcut, float* out_x, void* out_h, int h_dtype,
                               const float* gamma, const float* beta, int B, int H, int W, int C, int ws, int shift, float eps,
                               void* stream);
/* PatchMerging.forward up to the reduction GEMM (swin_trans.py:269-296): 2x2 gather-concat + LayerNorm(4C). */
int psalm_patch_merge_ln(const void* x, int x_dtype, void* out, int out_dtype, const float* gamma, const float* beta,
                         int B, int H, int W, int C, float eps, void* stream);
/* nn.GroupNorm(G, C) (+ optional ReLU) on NHWC tokens (msdeformattn.py:199-202,248-254).
 * workspace: B * (ceil(HW/64) + 1) * G * 2 floats.  C % 8 == 0. */
int psalm_groupnorm_nhwc(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                         float* workspace, int B, int HW, int C, int G, float eps, int relu, void* stream);
/* out[r,:] = a[r,:] + b[r % b_rows,:]  (positional / level / query embedding adds: msdeformattn.py:51-58,
 * mask2former_transformer_decoder.py:35-37,93-96). */
int psalm_add_bcast(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, long rows, int C,
                    long b_rows, void* stream);
/* inputs_embeds assembly (llava_phi.py:581-766,874-948): dst[r] = src{src_id[r]}[src_row[r]], src_id < 0 -> zeros. */
int psalm_gather_rows(const void* src0, int dt0, const void* src1, int dt1, const void* src2, int dt2, const void* src3,
                      int dt3, const int* src_id, const int* src_row, void* dst, int dst_dtype, long rows, int C,
                      void* stream);
/* get_seg_query / get_class_name_embedding / get_SEG_embedding / get_region_embedding (llava_phi.py:1299-1316,
 * 552-565,972-978,302-307): CSR row-set mean pooling of hidden states. */
int psalm_segment_mean(const void* x, int x_dtype, long ldx, const int* seg_offsets, const int* seg_rows, void* out,
                       int out_dtype, int nseg, int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Attention. */
/* WindowAttention.forward core (swin_trans.py:117-149) with the relative-position bias gather (:131-134) and the
 * shifted-window -100 mask (:369-387) computed in-kernel.  qkv (B*nW*ws*ws, 3C); out (B*nW*ws*ws, C); head_dim 32. */
int psalm_window_attention(const void* qkv, const float* bias_table, void* out, int dtype, int B, int nWh, int nWw, int C,
                           int heads, int ws, int shift, void* stream);
/* The fp32 / 12x12-window form whose output leaves as the split-f16 A operand of the projection GEMM (swin_trans.py:144-153 -> self.proj;
 * f16x3 mode): split_out rows of 2*split_kp f16 (hi at column head*32 + d, lo split_kp further), split_inv (rows).  One power-of-two scale per
 * window from a bound: a_inv = the row scales of the qkv GEMM's split-f16 A operand (rows in window order), bound_par = 2 device floats
 * {2^14 max_n sum_k |w_nk| over the v rows of the qkv weight, max |b_v|}; the output is a convex combination of the window's v rows. */
int psalm_window_attention_split(const float* qkv, const float* bias_table, const float* a_inv, const float* bound_par, void* split_out,
                                 int split_kp, float* split_inv, int B, int nWh, int nWw, int C, int heads, int ws, int shift, void* stream);
/* The same on the matrix cores for bf16 buffers and 12x12 windows (one block per (window, head), K / V^T / bias column
 * staged in LDS, scores of a whole 144-key row kept in MFMA accumulators). */
int psalm_window_attention_mfma(const void* qkv, const float* bias_table, void* out, int B, int nWh, int nWw, int C, int heads,
                                int ws, int shift, void* stream);
/* PhiAttention prefill core (modeling_phi.py:189-245,137-160) with partial RoPE (:92-122) fused into the loads:
 * q/k/v are column blocks of one row-strided buffer; causal + key padding mask (B,L) u8; fp32 softmax. */
int psalm_causal_attention(const void* qkv, int dtype, long ld, int q_off, int k_off, int v_off, void* out, long ldo,
                           int o_off, const float* cos_table, const float* sin_table, const unsigned char* key_mask, int B,
                           int L, int heads, int head_dim, int rot, void* stream);
/* The same operation on the matrix cores for bf16 buffers (v_mfma_f32_32x32x16_bf16, fp32 softmax statistics and
 * accumulation): RoPE/scale/transpose pre-pass into `workspace`, then one wavefront per 32-query tile.  `out` may alias
 * the q block of `qkv`.  workspace: psalm_causal_attention_mfma_workspace(B,L,heads) bytes, 16-byte aligned. */
long psalm_causal_attention_mfma_workspace(int B, int L, int heads);
int psalm_causal_attention_mfma(const void* qkv, long ld, int q_off, int k_off, int v_off, void* out, long ldo, int o_off,
                                const float* cos_table, const float* sin_table, const unsigned char* key_mask,
                                void* workspace, int B, int L, int heads, int head_dim, int rot, void* stream);
/* nn.MultiheadAttention core of the predictor (mask2former_transformer_decoder.py:35-45,93-105,645-666), head_dim 32;
 * mask (B,Lq,Lk) u8 1 = blocked; row_all_masked (B,Lq) u8 implements TD:647. */
int psalm_mha_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out, long ldo,
                        int dtype, const unsigned char* mask, const unsigned char* row_all_masked, int B, int Lq, int Lk,
                        int heads, int head_dim, void* stream);
/* The same on the matrix cores (bf16), split over the key axis so that 8 heads x S splits fill the chip; V is passed
 * TRANSPOSED: vt (B*heads*32, ldvt), row h*32+d, columns = keys, zero-padded to ldvt >= ceil(Lk/8)*8 (the value projection
 * GEMM with swapped operands writes it directly).  Lq <= 128; masked attention needs Lk % 4 == 0.
 * workspace: psalm_mha_attention_mfma_workspace(B, heads, Lk) bytes. */
long psalm_mha_attention_mfma_workspace(int B, int heads, int Lk);
int psalm_mha_attention_mfma(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                             const unsigned char* mask, const unsigned char* row_all_masked, void* workspace, int B, int Lq,
                             int Lk, int heads, int head_dim, void* stream);
/* forward_prediction_heads' attention-mask branch (mask2former_transformer_decoder.py:754-760): bilinear resize of the
 * mask logits to (Ht,Wt), sigmoid < 0.5, plus the all-masked row flags. */
int psalm_attn_mask(const float* masks, unsigned char* out, unsigned char* row_all_masked, int BQ, int h, int w, int Ht,
                    int Wt, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Image / layout kernels. */
/* PatchEmbed im2col (swin_trans.py:427-443): img (B,Cin,H,W) f32 -> (B*Hp*Wp, Kpad), K order (c,ky,kx), zero pad. */
int psalm_patch_im2col(const float* img, void* out, int out_dtype, int B, int Cin, int H, int W, int ps, int Kpad,
                       void* stream);
/* k x k / stride / pad im2col of NHWC tokens (projector convs builder.py:85-111, FPN conv msdeformattn.py:248-254). */
int psalm_im2col_nhwc(const void* x, void* out, int dtype, int B, int H, int W, int C, int k, int stride, int pad,
                      void* stream);
/* F.interpolate(bilinear, align_corners=False) on N planes with optional top-left crop first
 * (llava_phi.py:1401-1406; detectron2 sem_seg_postprocess at llava_phi.py:1427-1429). */
int psalm_resize_planes(const void* in, int in_dtype, void* out, int out_dtype, long N, int h, int w, int hc, int wc, int H,
                        int W, void* stream);
/* FPN top-down step (msdeformattn.py:300-308): out = lateral + bilinear_up(small), NHWC. */
int psalm_upsample_add_nhwc(const void* lateral, int lat_dtype, const void* small, int small_dtype, void* out, int out_dtype,
                            int B, int h, int w, int H, int W, int C, void* stream);
/* NCHW <-> NHWC. */
int psalm_permute_layout(const void* in, int in_dtype, void* out, int out_dtype, int B, int C, long HW, int to_nhwc,
                         void* stream);
/* region_pooling.forward (visual_prompt_module/context_cluster.py:357-400): grid_sample(align_corners=True) of the
 * projector tokens at n points per region + mean.  pts (R,n,2) f32 (y,x) in [0,1). */
int psalm_region_pool(const float* tokens, const int* img_of_region, const float* pts, float* out, int R, int h, int w, int C,
                      int n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Post-processing (llava_phi.py:308-447). */
/* softmax over C1 class logits; probsT = transpose of the first C1-1 columns padded to Kpad (semantic einsum operand);
 * per-query max score / label (panoptic, llava_phi.py:328). */
int psalm_class_softmax(const float* cls, float* probs, void* probsT, int probsT_dtype, float* score, int* label, int Q, int C1,
                        int Kpad, void* stream);
/* sigmoid(mask)^T padded to Kpad: second operand of class_name_semantic_inference (llava_phi.py:402-406). */
int psalm_sigmoid_transpose(const float* mask, void* out, int out_dtype, int Q, long HW, int Kpad, void* stream);
/* class_name_semantic_inference (llava_phi.py:402-406) fused for the bf16 mode: sem[c,p] = sum_q probsT[c,q] * sigmoid(mask[q,p]) in
 * one pass over the mask logits (no sigmoid^T tensor in HBM).  probsT (C,128) bf16 from psalm_class_softmax; Q <= 128, C <= 160.
 * mask_score (Q) f32 or NULL: psalm_mask_scores' result from the same read (workspace Q*512*2 floats, else NULL). */
int psalm_semantic_from_masks(const float* mask, const void* probsT_bf16, float* out, float* mask_score, float* workspace, int Q, int C,
                              long HW, int Kpad, void* stream);
/* fp32-class (split-f16) form for precision="f16x3": probsT (C,128) FLOAT32; both operands carried as f16 hi + lo with the fixed scale
 * 2^13 (probabilities / sigmoids <= 1), three f16 MFMAs per product, fp32 accumulate.  Same single pass over the logits. */
int psalm_semantic_from_masks_x3(const float* mask, const float* probsT_f32, float* out, float* mask_score, float* workspace, int Q, int C,
                                 long HW, int Kpad, void* stream);
/* mask score = sum(sigmoid(m)*[m>0]) / (sum([m>0]) + 1e-6) (llava_phi.py:318-320,439-441). workspace Q*64*2 floats. */
int psalm_mask_scores(const float* mask, float* score, float* workspace, int Q, long HW, void* stream);
/* topk over Q*C candidates + thing filter + score product (llava_phi.py:407-447, 308-324). */
int psalm_topk_select(const float* vals, int Q, int C, int stride, int k, const int* is_thing, const float* mask_score,
                      float* out_score, int* out_class, int* out_query, int* count, int apply_sigmoid, void* stream);
/* pred_masks = (mask[query] > 0).float() (llava_phi.py:316,437). */
int psalm_binarize_gather(const float* mask, const int* query, const int* count, float* out, int n, long HW, void* stream);
/* class_name_panoptic_inference (llava_phi.py:325-386) entirely on device: keep / argmax / area tests / stuff merge. */
int psalm_panoptic(const float* mask, const float* score, const int* label, const int* is_thing, int* argq, int* counts,
                   int* final_id, int* pan, int* info, int* ninfo, int Q, long HW, int num_classes, float obj_thr,
                   float overlap_thr, void* stream);
/* region_inference scores (llava_phi.py:387-400). */
int psalm_region_scores(const float* logits, const float* mask_score, float* out, int K, int Q, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Stage-level entries ("C-ABI groups behind B2"; csrc/stages.hip): one call issues the launch sequence of a whole
 * stage of PSALM.eval_seg from native code, through the op-level entries above, in the order and with the arguments psalm_amd/model.py uses
 * -- same bits.  The caller owns every buffer; no allocation, no synchronisation.
 *
 * psalm_phi_forward: PhiModel.forward on inputs_embeds (transformers modeling_phi.py:343-396 as called from
 * psalm/model/language_model/llava_phi.py:1350-1365), precision "f16x3".  Per layer (host array `layers`): w1 = [k | v | q | fc1] and
 * w2 = [dense | fc2] in split-f16 form (rows of 2*ceil64(K) f16 + inverse row scales, psalm_split_f16), biases (b2 = dense + fc2), the layer's
 * input LayerNorm, bnd = the 4 bound parameters of psalm_gemm_x3_split for the fc1 rows, paired = the fc1 rows of w1 / b1 are permuted for
 * paired stores.  embeds (B*L, hidden) f32; key_mask (B, L) u8; cos / sin (L, rot) f32; hidden_out (B*L, hidden) f32 = final LayerNorm output.
 * workspace: psalm_phi_forward_workspace(d, B, L) bytes, 256-byte aligned; gemm_workspace: split-K scratch as for psalm_gemm_x3. */
typedef struct psalm_phi_layer {
    const void* w1; const float* w1_scale; const float* b1;
    const void* w2; const float* w2_scale; const float* b2;
    const float* ln_g; const float* ln_b;
    const float* bnd;
    int paired;
} psalm_phi_layer;
typedef struct psalm_phi_desc {
    int num_layers, hidden, intermediate, heads, head_dim, rot;
    float ln_eps;
    const psalm_phi_layer* layers;           /* HOST array of num_layers entries (device pointers inside) */
    const float* final_g; const float* final_b;
} psalm_phi_desc;
long psalm_phi_forward_workspace(const psalm_phi_desc* d, int B, int L);
int psalm_phi_forward(const psalm_phi_desc* d, const float* embeds, const unsigned char* key_mask, const float* cos_table, const float* sin_table,
                      int B, int L, float* hidden_out, void* workspace, long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes,
                      void* stream);

/* psalm_swin_forward: SwinTransformer.forward (swin_trans.py:608-633; blocks :194-253, window attention :117-149, patch merging :266-296, patch
 * embedding :427-443), precision "f16x3", 12 x 12 windows, head dim 32.  GEMM weights in split-f16 form (`*_w` rows of 2*ceil64(K) f16, `*_ws`
 * inverse row scales); qkv_bnd / fc1_bnd: the bound parameters of psalm_window_attention_split (over the v rows of qkv) / psalm_gemm_x3_split;
 * fc1_paired: fc1 rows permuted for paired stores; rpb: relative_position_bias_table ((2*12-1)^2, heads) f32.  Stage s: `dim` channels, `heads`,
 * `depth` blocks, the output norm (swin_trans.py:622-626) and -- but for the last stage -- the PatchMerging norm + reduction weight.
 * images (B,3,H,W) f32; outs_host: HOST array of num_stages DEVICE pointers, stage s receives (B*h_s*w_s, dim_s) f32 tokens (post norm_s). */
typedef struct psalm_swin_block {
    const float* n1_g; const float* n1_b; const float* n2_g; const float* n2_b;
    const void* qkv_w; const float* qkv_ws; const float* qkv_b; const float* qkv_bnd;
    const void* proj_w; const float* proj_ws; const float* proj_b;
    const void* fc1_w; const float* fc1_ws; const float* fc1_b; const float* fc1_bnd; int fc1_paired;
    const void* fc2_w; const float* fc2_ws; const float* fc2_b;
    const float* rpb;
} psalm_swin_block;
typedef struct psalm_swin_stage {
    int depth, heads, dim;
    const psalm_swin_block* blocks;          /* HOST array of depth entries */
    const float* out_g; const float* out_b;
    const float* ds_g; const float* ds_b; const void* ds_w; const float* ds_ws;   /* NULL in the last stage */
} psalm_swin_stage;
typedef struct psalm_swin_desc {
    int num_stages, patch, window, pe_kpad, mlp_ratio;
    const void* pe_w; const float* pe_ws; const float* pe_b; const float* pe_ln_g; const float* pe_ln_b;
    const psalm_swin_stage* stages;          /* HOST array */
} psalm_swin_desc;
long psalm_swin_forward_workspace(const psalm_swin_desc* d, int B, int H, int W);
int psalm_swin_forward(const psalm_swin_desc* d, const float* images, int B, int H, int W, float* const* outs_host, void* workspace,
                       long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes, void* stream);

/* psalm_projector_forward: the conv projector between Swin's last stage and the LLM (psalm/model/multimodal_projector/builder.py:365-375; the
 * BasicBlock :85-111 with eval BatchNorm folded into the convolution weights, conv2 applied twice as in the reference).  Weights (Cout, k*k*Cin)
 * with K order (ky, kx, c), split-f16 form.  res5 (B*h*w, in_dim) f32 NHWC tokens -> out (B*ho*wo, out_dim) f32, ho = (h - 1) / 2 + 1. */
typedef struct psalm_projector_desc {
    int in_dim, mid_dim, out_dim;
    const void* c1_w; const float* c1_ws; const float* c1_b;        /* conv1 (3x3, stride 2) + bn1 folded */
    const void* c2_w; const float* c2_ws;                            /* conv2 (3x3), first application (no norm) */
    const void* c2f_w; const float* c2f_ws; const float* c2f_b;     /* conv2 second application + bn2 folded */
    const void* ds_w; const float* ds_ws; const float* ds_b;        /* downsample 1x1 stride 2 + its norm folded */
    const void* fc_w; const float* fc_ws; const float* fc_b;
} psalm_projector_desc;
long psalm_projector_forward_workspace(const psalm_projector_desc* d, int B, int h, int w);
int psalm_projector_forward(const psalm_projector_desc* d, const float* res5, int B, int h, int w, float* out, void* workspace, long workspace_bytes,
                            void* gemm_workspace, long gemm_workspace_bytes, void* stream);

/* psalm_pixel_decoder_forward: MSDeformAttnPixelDecoder.forward_features for ONE image (msdeformattn.py:268-315; encoder layer :57-72;
 * MSDeformAttn ops/modules/ms_deform_attn.py:82-124), precision "f16x3", 3 levels x 4 points, head dim 32.  feats_host: HOST array of the 4 DEVICE token
 * buffers res2 .. res5 of the image ((h_i*w_i, in_dims[i]) f32), hw_host: their (h, w) pairs; lvl_pos (S, D) f32 = sine position embedding + level embedding of
 * the level-concatenated tokens (res5 | res4 | res3; S = their total).  Outputs: mask_features (h_2*w_2, mask_dim) f32, ms_out (S, D) f32 (the encoder
 * output, level-concatenated).  Per encoder layer: ow = [sampling_offsets ; attention_weights] stacked; l1_bnd / l1_paired as psalm_gemm_x3_split. */
typedef struct psalm_pd_enc_layer {
    const void* value_w; const float* value_ws; const float* value_b;
    const void* ow_w; const float* ow_ws; const float* ow_b;
    const void* out_w; const float* out_ws; const float* out_b;
    const float* n1_g; const float* n1_b;
    const void* l1_w; const float* l1_ws; const float* l1_b; const float* l1_bnd; int l1_paired;
    const void* l2_w; const float* l2_ws; const float* l2_b;
    const float* n2_g; const float* n2_b;
} psalm_pd_enc_layer;
typedef struct psalm_pd_desc {
    int D, G, M, num_layers, ffn, mask_dim;
    int in_dims[4];                                                  /* channels of res2, res3, res4, res5 */
    const void* ip_w[3]; const float* ip_ws[3]; const float* ip_b[3]; const float* ip_gn_g[3]; const float* ip_gn_b[3];   /* input_proj of res5, res4, res3 */
    const void* adapter_w; const float* adapter_ws; const float* adapter_b; const float* adapter_gn_g; const float* adapter_gn_b;
    const void* layer_w; const float* layer_ws; const float* layer_b; const float* layer_gn_g; const float* layer_gn_b;
    const void* mf_w; const float* mf_ws; const float* mf_b;
    const psalm_pd_enc_layer* layers;                                /* HOST array */
} psalm_pd_desc;
long psalm_pixel_decoder_forward_workspace(const psalm_pd_desc* d, const int* hw_host);
int psalm_pixel_decoder_forward(const psalm_pd_desc* d, const float* const* feats_host, const int* hw_host, const float* lvl_pos, float* mask_features,
                                float* ms_out, void* workspace, long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes, void* stream);

/* psalm_predictor_forward: MultiScaleMaskedTransformerDecoder.forward for ONE image (mask2former_transformer_decoder.py:596-693; heads :695-762;
 * layers :19-199), precision "f16x3", head dim 32, Q <= 128 queries, <= 3 levels.  Layer weights are plain float32 (the M = Q GEMMs run on the exact-fp32
 * kernel); lvl_k / lvl_v: the cross-attention K / V projections of all decoder layers that read a level, stacked along N, split-f16 form.
 * ms_host / prpos_host: HOST arrays of the per-level DEVICE buffers (h_l*w_l, D) f32 -- encoder output and (sine position + level embedding);
 * mask_features (H2*W2, mask_dim) f32; seg_query (Q, D) f32; class_emb (n_cls, D) / seg_emb (n_seg, D) / region_emb (n_reg, D) f32 or NULL.
 * Outputs: pred_masks (Q, H2*W2) f32; cls_logits (Q, n_cls), seg_logits (Q, n_seg), region_logits (n_reg, Q) f32 for the embeddings given.
 * workspace: psalm_predictor_forward_workspace(d, hw_levels_host, H2, W2, n_reg) bytes. */
typedef struct psalm_pr_layer {
    const float* cq_w; const float* cq_b; const float* co_w; const float* co_b; const float* cn_g; const float* cn_b;
    const float* sqk_w; const float* sqk_b; const float* sv_w; const float* sv_b; const float* so_w; const float* so_b; const float* sn_g; const float* sn_b;
    const float* f1_w; const float* f1_b; const float* f2_w; const float* f2_b; const float* fn_g; const float* fn_b;
} psalm_pr_layer;
typedef struct psalm_pr_desc {
    int D, heads, Q, num_layers, num_levels, ffn, mask_dim;
    const void* lvl_k_w[3]; const float* lvl_k_ws[3]; const float* lvl_k_b[3];
    const void* lvl_v_w[3]; const float* lvl_v_ws[3]; const float* lvl_v_b[3];
    const float* level_embed; const float* query_embed; const float* dn_g; const float* dn_b;
    const float* mask_embed_w[3]; const float* mask_embed_b[3];
    const float* SEG_w[2]; const float* SEG_b[2]; const float* CLASS_w[2]; const float* CLASS_b[2]; const float* REGION_w[2]; const float* REGION_b[2];
    const psalm_pr_layer* layers;                                   /* HOST array */
} psalm_pr_desc;
long psalm_predictor_forward_workspace(const psalm_pr_desc* d, const int* hw_levels_host, int H2, int W2, int n_extra_rows);
/* kv_ready != 0: the level K / V projections and the mask-feature split are in `workspace` already -- psalm_predictor_kv wrote them, with the same descriptor /
 * geometry / n_extra_rows (= n_reg) and the SAME workspace; it needs nothing from the LLM, so a caller can issue it on another stream behind the pixel decoder
 * while the LLM runs (psalm_amd/model.py does) and join before this call. */
int psalm_predictor_kv(const psalm_pr_desc* d, const float* const* ms_host, const int* hw_levels_host, const float* const* prpos_host,
                       const float* mask_features, int H2, int W2, int n_extra_rows, void* workspace, long workspace_bytes, void* gemm_workspace,
                       long gemm_workspace_bytes, void* stream);
int psalm_predictor_forward(const psalm_pr_desc* d, const float* const* ms_host, const int* hw_levels_host, const float* const* prpos_host,
                            const float* mask_features, int H2, int W2, const float* seg_query, const float* class_emb, int n_cls, const float* seg_emb,
                            int n_seg, const float* region_emb, int n_reg, float* pred_masks, float* cls_logits, float* seg_logits, float* region_logits,
                            void* workspace, long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes, int kv_ready, void* stream);


/* psalm_postprocess: llava_phi.py:1401-1466 for ONE image from native code -- the mask logits up-sampled to the padded image size (LP:1401-1406),
 * cropped to the un-padded box and resized to the original size (sem_seg_postprocess, LP:1418-1429), then the task's inference function:
 *   PSALM_POST_SEMANTIC   class_name_semantic_inference on the padded-size masks, THEN crop / resize of the class map (LP:301,402-406,1437-1440)
 *   PSALM_POST_INSTANCE   class_name_instance_inference, no thing filter (LP:407-447)
 *   PSALM_POST_PANOPTIC   semantic + instance (thing filter) + class_name_panoptic_inference (LP:325-386)
 *   PSALM_POST_REFERRING  SEG_instance_inference (LP:308-324)
 *   PSALM_POST_REGION     region_inference (LP:387-400)
 * through the op-level entries above in psalm_amd/model.py's order -- the same words (tests/test_6_model_emu.py, tests/test_9_e2e_gpu.py).  The caller
 * owns every buffer (outputs at their maximum sizes, one workspace of psalm_postprocess_workspace() bytes); nothing is allocated, nothing
 * synchronises; the data-dependent counts stay on the device (`counts`) for the caller's one read-back.  precision "f16x3" / "fp32"; Q <= 128 and at most
 * 160 classes for the tasks that form the class map (the fused split-f16 pass; wider vocabularies: the op-level calls).
 *
 * io (device pointers unless noted; unused ones NULL):
 *   in   pred_masks (Q,h,w) f32 | mask_up (Q,Hpad,Wpad) f32 or NULL (the up-sampled logits if the caller already formed them -- the captured graph of
 *        model.py does; NULL: formed here, in `mask_pred` when no crop / resize follows, else in the workspace) | cls_logits (Q,C1) | seg_logits (Q,1)
 *        | region_logits (k,Q) | is_thing (C1-1) i32
 *   out  mask_pred (Q,out_h,out_w) f32: the masks the task's function saw (semantic: (Q,Hpad,Wpad), written only when mask_up is NULL) | sem_seg (C1-1,out_h,out_w) f32 |
 *        scores (Q [, k for region: (Q,k)]) f32 | classes (Q) i64 | query (Q) i64 | inst_masks (Q,out_h,out_w) f32 | boxes (Q,4) f32 zeros |
 *        pan (out_h,out_w) i32 | counts i32: [0] instances kept, [1] segments, [2..2+3Q) segments_info rows (id, isthing, category)
 *   mask_pred_is (HOST int*, may be NULL): 0 = `mask_pred` holds the returned masks, 1 = they are `mask_up` itself (no crop / resize was needed, or the
 *        semantic task, whose `mask_pred` is the padded-size tensor). */
#define PSALM_POST_SEMANTIC 0
#define PSALM_POST_INSTANCE 1
#define PSALM_POST_PANOPTIC 2
#define PSALM_POST_REFERRING 3
#define PSALM_POST_REGION 4
typedef struct psalm_post_desc {
    int task, Q, h, w, Hpad, Wpad, crop_h, crop_w, out_h, out_w, C1, n_region;
    float obj_thr, overlap_thr;
} psalm_post_desc;
typedef struct psalm_post_io {
    const float* pred_masks; const float* mask_up; const float* cls_logits; const float* seg_logits; const float* region_logits; const int* is_thing;
    float* mask_pred; float* sem_seg; float* scores; long long* classes; long long* query; float* inst_masks; float* boxes; int* pan; int* counts;
} psalm_post_io;
long psalm_postprocess_workspace(const psalm_post_desc* d, int have_mask_up);
int psalm_postprocess(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);
/* The five names of SURVEY.md section 8(b): psalm_postprocess with d->task checked against the name. */
int psalm_postprocess_semantic(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);
int psalm_postprocess_instance(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);
int psalm_postprocess_panoptic(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);
int psalm_postprocess_referring(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);
int psalm_postprocess_region(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
