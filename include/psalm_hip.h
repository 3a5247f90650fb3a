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
int psalm_abi_version(void);
const char* psalm_backend(void); /* "hip-gfx950" */

/* Replaces MSDA.ms_deform_attn_forward, the reference's own native-op seam:
 *   OPS/src/vision.cpp:18-21 (pybind), OPS/src/ms_deform_attn.h:25-44 (dispatch),
 *   OPS/src/cuda/ms_deform_attn_cuda.cu:25-85 (host), OPS/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (kernel).
 * value (B,S,M,D) value_dtype; spatial_shapes_host (L,2) int64 (H,W); level_start_host (L) int64;
 * sampling_loc (B,Lq,M,L,P,2) f32 (x,y) in [0,1]; attn_weight (B,Lq,M,L,P) f32; out (B,Lq,M*D) out_dtype.
 * D % 4 == 0.  The reference's im2col_step batching is unnecessary (one launch covers the batch). */
int psalm_msda_forward(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                       const int64_t* level_start_host, const float* sampling_loc, const float* attn_weight, void* out,
                       int out_dtype, int B, int S, int M, int D, int L, int Lq, int P, void* stream);

/* The same gather with MSDeformAttn.forward's location/softmax arithmetic fused in
 * (OPS/modules/ms_deform_attn.py:101-110): offsets_logits (B,S,M*L*P*3) f32 is the output of the
 * concatenated [sampling_offsets ; attention_weights] projection of (src+pos); queries are the
 * S pixels themselves (encoder self-attention, msdeformattn.py:76-87 reference points). L=3, P=4. */
int psalm_msda_fused(const void* value, int value_dtype, const int64_t* spatial_shapes_host,
                     const int64_t* level_start_host, const float* offsets_logits, void* out, int out_dtype, int B, int S,
                     int M, int D, int L, int P, void* stream);

#ifdef __cplusplus
}
#endif
#endif
