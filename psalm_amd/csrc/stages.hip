// Stage-level entries of the C ABI ("C-ABI groups behind B2"): one call issues the whole launch sequence of a stage
// of PSALM.eval_seg from native code -- no interpreter between the launches -- using nothing but the op-level entries of this library, in the
// order and with the arguments psalm_amd/model.py uses (results are bit for bit those of the op-by-op sequence: tests/test_6_model_emu.py,
// tests/test_9_e2e_gpu.py).  Never allocates (the caller passes the workspace), never synchronises, asynchronous on `stream`.
//
//   psalm_phi_forward    PhiModel.forward over inputs_embeds for the prefill (transformers modeling_phi.py:343-396 as called at
//                        psalm/model/language_model/llava_phi.py:1350-1365): 24 x { [k|v|q|fc1] GEMM with the gelu(fc1) columns leaving as the
//                        split-f16 operand of the next GEMM; RoPE + causal attention writing its columns of the same operand;
//                        [dense|fc2] GEMM + residual + the NEXT layer's LayerNorm + its split }, final LayerNorm.  precision "f16x3".
#include "common.h"
#include "psalm_hip.h"

static inline long al256(long v) { return (v + 255) / 256 * 256; }

struct PhiLayout {
    long x0, x1, h, hinv, big, a2, inv2, attn, total;
    int Kp;
};
static PhiLayout phi_layout(const psalm_phi_desc* d, int B, int L) {
    PhiLayout o;
    const long M = (long)B * L, H = d->hidden, I = d->intermediate;
    o.Kp = (int)((H + 63) / 64 * 64);
    long p = 0;
    o.x0 = p; p += al256(M * H * 4);
    o.x1 = p; p += al256(M * H * 4);
    o.h = p; p += al256(M * 2 * o.Kp * 2);
    o.hinv = p; p += al256(M * 4);
    o.big = p; p += al256(M * 3 * H * 4);
    o.a2 = p; p += al256(M * 2 * (H + I) * 2);
    o.inv2 = p; p += al256(M * 4);
    o.attn = p; p += al256(psalm_causal_attention_f32_workspace(B, L, d->heads));
    o.total = p;
    return o;
}

static int phi_check(const psalm_phi_desc* d) {
    PSALM_CHECK_ARG(d && d->layers && d->num_layers >= 1, "psalm_phi_forward: descriptor with >= 1 layer");
    PSALM_CHECK_ARG(d->head_dim == 64 && d->rot == 32 && d->hidden == d->heads * d->head_dim, "psalm_phi_forward: head_dim 64, rotary dim 32 (Phi-1.5)");
    PSALM_CHECK_ARG((d->hidden + d->intermediate) % 64 == 0 && d->hidden % 8 == 0 && d->intermediate % 8 == 0,
                    "psalm_phi_forward: (hidden + intermediate) % 64 == 0, hidden % 8 == 0");
    return 0;
}

extern "C" long psalm_phi_forward_workspace(const psalm_phi_desc* d, int B, int L) {
    if (phi_check(d) != 0 || B <= 0 || L <= 0) return -1;
    return phi_layout(d, B, L).total;
}

extern "C" int psalm_phi_forward(const psalm_phi_desc* d, const float* embeds, const unsigned char* key_mask, const float* cos_table,
                                 const float* sin_table, int B, int L, float* hidden_out, void* workspace, long workspace_bytes,
                                 void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    if (phi_check(d) != 0) return -1;
    PSALM_CHECK_ARG(embeds && key_mask && cos_table && sin_table && hidden_out && workspace && B > 0 && L > 0, "psalm_phi_forward: null argument");
    const PhiLayout lo = phi_layout(d, B, L);
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_phi_forward: workspace of psalm_phi_forward_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    const int M = B * L, H = d->hidden, I = d->intermediate, Kp = lo.Kp, K2 = H + I;
    float* x[2] = {(float*)(ws + lo.x0), (float*)(ws + lo.x1)};
    void* h = ws + lo.h;
    float* hinv = (float*)(ws + lo.hinv);
    float* big = (float*)(ws + lo.big);
    void* a2 = ws + lo.a2;
    float* inv2 = (float*)(ws + lo.inv2);
    void* attn = ws + lo.attn;
    int rc;
    // the K padding columns of h (Kp > H) must read as zeros for the consumer GEMM: psalm_layernorm_split / psalm_gemm_x3_ln_split write whole rows
    // of 2 * ceil64(H) only when H % 64 == 0 -- the Python path has the same requirement (PSALM.llm: Hd % 64 for the fused form)
    const psalm_phi_layer* l0 = &d->layers[0];
    // layer 0's input LayerNorm, straight into split form (model.py: o.layernorm_split(x, llm0.ln))
    rc = psalm_layernorm_split(embeds, H, nullptr, H, l0->ln_g, l0->ln_b, M, H, d->ln_eps, h, hinv, nullptr, 0, nullptr, nullptr, stream);
    if (rc) return rc;
    const float* xin = embeds;
    int cur = 0;
    for (int i = 0; i < d->num_layers; ++i) {
        const psalm_phi_layer* ly = &d->layers[i];
        const bool last = i == d->num_layers - 1;
        PSALM_CHECK_ARG(ly->w1 && ly->w1_scale && ly->b1 && ly->w2 && ly->w2_scale && ly->b2 && ly->bnd, "psalm_phi_forward: layer weights missing");
        // [k | v | q | gelu_new(fc1)]: k, v, q -> big (fp32), gelu(fc1) -> columns [H, H + I) of the operand a2 with the row scales inv2
        rc = psalm_gemm_x3_split(h, 2L * Kp, hinv, ly->w1, 2L * Kp, ly->w1_scale, Kp, ly->b1, big, 3L * H, M, 3 * H + I, /*gelu_new*/ 3, 3 * H, a2,
                                 2L * K2, K2, H, 3 * H, ly->paired, inv2, ly->bnd, 1, gemm_workspace, gemm_workspace_bytes, stream);
        if (rc) return rc;
        // RoPE + causal attention (q at 2H, k at 0, v at H of big) -> columns [0, H) of a2 under the same row scales
        rc = psalm_causal_attention_f32_split(big, 3L * H, 2 * H, 0, H, a2, 2L * K2, K2, 0, inv2, cos_table, sin_table, key_mask, attn, B, L, d->heads,
                                              d->head_dim, d->rot, stream);
        if (rc) return rc;
        float* xout = x[cur];
        if (last || H % 64 != 0 || H > 2048) {
            rc = psalm_gemm_x3(a2, 2L * K2, inv2, ly->w2, 2L * K2, ly->w2_scale, K2, ly->b2, xin, H, xout, H, M, H, 0, 0, gemm_workspace,
                               gemm_workspace_bytes, stream);
            if (rc) return rc;
            if (last) {
                rc = psalm_layernorm(xout, PSALM_F32, H, hidden_out, PSALM_F32, H, nullptr, 0, d->final_g, d->final_b, M, H, d->ln_eps, stream);
            } else {
                const psalm_phi_layer* nx = &d->layers[i + 1];
                rc = psalm_layernorm_split(xout, H, nullptr, H, nx->ln_g, nx->ln_b, M, H, d->ln_eps, h, hinv, nullptr, 0, nullptr, nullptr, stream);
            }
            if (rc) return rc;
        } else {
            const psalm_phi_layer* nx = &d->layers[i + 1];
            // x = a2 . w2^T + b2 + x; h = split(LayerNorm_{i+1}(x)): one row pass behind the K slices
            rc = psalm_gemm_x3_ln_split(a2, 2L * K2, inv2, ly->w2, 2L * K2, ly->w2_scale, K2, ly->b2, xin, H, xout, H, M, H, nx->ln_g, nx->ln_b,
                                        d->ln_eps, nullptr, H, h, hinv, gemm_workspace, gemm_workspace_bytes, stream);
            if (rc) return rc;
        }
        xin = xout;
        cur ^= 1;
    }
    return 0;
}
