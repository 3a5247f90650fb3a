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

// ================================================================================================= Swin tower
//   psalm_swin_forward   SwinTransformer.forward (psalm/model/visual_prompt... swin_trans.py:608-633; blocks :194-253, window attention :117-149,
//                        patch merging :266-296, patch embedding :427-443) for precision "f16x3": patch im2col + GEMM + LayerNorm, then per block
//                        { norm1 + shift + window partition -> split operand; qkv GEMM; window attention -> split operand; proj GEMM; window reverse +
//                        residual + norm2 -> split operand; fc1 GEMM with gelu -> split operand; fc2 GEMM + residual }, per stage the output norm and
//                        (but for the last) patch merging + reduction GEMM.  The op-by-op sequence of PSALM.swin, 7 launches per block.
static inline int c64(int v) { return (v + 63) / 64 * 64; }

struct SwinGeom { int Hc[8], Wc[8], C[8]; long rows[8], rows_w[8]; };
static SwinGeom swin_geom(const psalm_swin_desc* d, int B, int H, int W) {
    SwinGeom g;
    int hc = (H + d->patch - 1) / d->patch, wc = (W + d->patch - 1) / d->patch;
    for (int s = 0; s < d->num_stages; ++s) {
        g.Hc[s] = hc; g.Wc[s] = wc; g.C[s] = d->stages[s].dim;
        g.rows[s] = (long)B * hc * wc;
        const int nWh = (hc + d->window - 1) / d->window, nWw = (wc + d->window - 1) / d->window;
        g.rows_w[s] = (long)B * nWh * nWw * d->window * d->window;
        hc = (hc + 1) / 2; wc = (wc + 1) / 2;
    }
    return g;
}
struct SwinLayout { long x[3], xw, xwinv, qkv, aw, awinv, pw, h, hinv, hs, hsinv, cols, colsp, colsinv, total; };
static SwinLayout swin_layout(const psalm_swin_desc* d, const SwinGeom& g) {
    long mx = 0, mxw = 0, mqkv = 0, mpw = 0, mh = 0, mhs = 0, mrows = 0, mrw = 0, mcols = 0, mcolsp = 0;
    for (int s = 0; s < d->num_stages; ++s) {
        const long C = g.C[s], Kp = c64((int)C), K4 = c64((int)(d->mlp_ratio * C));
        mx = std::max(mx, g.rows[s] * C * 4);
        mxw = std::max(mxw, g.rows_w[s] * 2 * Kp * 2);
        mqkv = std::max(mqkv, g.rows_w[s] * 3 * C * 4);
        mpw = std::max(mpw, g.rows_w[s] * C * 4);
        mh = std::max(mh, g.rows[s] * 2 * Kp * 2);
        mhs = std::max(mhs, g.rows[s] * 2 * K4 * 2);
        mrows = std::max(mrows, g.rows[s]);
        mrw = std::max(mrw, g.rows_w[s]);
        if (s + 1 < d->num_stages) {                             // patch merging: (rows / 4, 4 C) f32 and its split form
            const long r2 = g.rows[s + 1];
            mcols = std::max(mcols, r2 * 4 * C * 4);
            mcolsp = std::max(mcolsp, r2 * 2 * c64((int)(4 * C)) * 2);
        }
    }
    mcols = std::max(mcols, g.rows[0] * d->pe_kpad * 4);       // patch-embedding im2col and its split form
    mcolsp = std::max(mcolsp, g.rows[0] * 2 * c64(d->pe_kpad) * 2);
    SwinLayout o;
    long p = 0;
    for (int i = 0; i < 3; ++i) { o.x[i] = p; p += al256(mx); }
    o.xw = p; p += al256(mxw);
    o.xwinv = p; p += al256(mrw * 4);
    o.qkv = p; p += al256(mqkv);
    o.aw = p; p += al256(mxw);
    o.awinv = p; p += al256(mrw * 4);
    o.pw = p; p += al256(mpw);
    o.h = p; p += al256(mh);
    o.hinv = p; p += al256(mrows * 4);
    o.hs = p; p += al256(mhs);
    o.hsinv = p; p += al256(mrows * 4);
    o.cols = p; p += al256(mcols);
    o.colsp = p; p += al256(mcolsp);
    o.colsinv = p; p += al256(mrows * 4);
    o.total = p;
    return o;
}
static int swin_check(const psalm_swin_desc* d) {
    PSALM_CHECK_ARG(d && d->stages && d->num_stages >= 1 && d->num_stages <= 8 && d->window == 12 && d->patch >= 1 && d->pe_kpad % 8 == 0,
                    "psalm_swin_forward: descriptor (1..8 stages, 12 x 12 windows)");
    for (int s = 0; s < d->num_stages; ++s) {
        const psalm_swin_stage* st = &d->stages[s];
        PSALM_CHECK_ARG(st->blocks && st->depth >= 1 && st->dim % 8 == 0 && st->dim <= 2048 && st->dim == st->heads * 32,
                        "psalm_swin_forward: stage dims multiples of 8, <= 2048, head dim 32");
    }
    return 0;
}
extern "C" long psalm_swin_forward_workspace(const psalm_swin_desc* d, int B, int H, int W) {
    if (swin_check(d) != 0 || B <= 0 || H <= 0 || W <= 0) return -1;
    return swin_layout(d, swin_geom(d, B, H, W)).total;
}

extern "C" int psalm_swin_forward(const psalm_swin_desc* d, const float* images, int B, int H, int W, float* const* outs_host, void* workspace,
                                  long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    if (swin_check(d) != 0) return -1;
    PSALM_CHECK_ARG(images && outs_host && workspace && B > 0 && H > 0 && W > 0, "psalm_swin_forward: null argument");
    const SwinGeom g = swin_geom(d, B, H, W);
    const SwinLayout lo = swin_layout(d, g);
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_swin_forward: workspace of psalm_swin_forward_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    float* X[3] = {(float*)(ws + lo.x[0]), (float*)(ws + lo.x[1]), (float*)(ws + lo.x[2])};
    void* xw = ws + lo.xw; float* xwinv = (float*)(ws + lo.xwinv);
    float* qkv = (float*)(ws + lo.qkv);
    void* aw = ws + lo.aw; float* awinv = (float*)(ws + lo.awinv);
    float* pw = (float*)(ws + lo.pw);
    void* hsp = ws + lo.h; float* hinv = (float*)(ws + lo.hinv);
    void* hs = ws + lo.hs; float* hsinv = (float*)(ws + lo.hsinv);
    float* cols = (float*)(ws + lo.cols); void* colsp = ws + lo.colsp; float* colsinv = (float*)(ws + lo.colsinv);
    const int wsz = d->window;
    const float eps = 1e-5f;
    int rc;
#define SW(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- patch embedding: im2col (K order c, ky, kx) -> split -> GEMM -> LayerNorm
    {
        const int M = (int)g.rows[0], Kpe = d->pe_kpad, Kp = c64(Kpe), E = g.C[0];
        SW(psalm_patch_im2col(images, cols, PSALM_F32, B, 3, H, W, d->patch, Kpe, stream));
        SW(psalm_split_f16(cols, Kpe, colsp, 2L * Kp, colsinv, M, Kpe, stream));
        SW(psalm_gemm_x3(colsp, 2L * Kp, colsinv, d->pe_w, 2L * Kp, d->pe_ws, Kp, d->pe_b, nullptr, 0, X[1], E, M, E, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        SW(psalm_layernorm3(X[1], PSALM_F32, E, X[0], PSALM_F32, E, nullptr, 0, nullptr, 0, nullptr, 0, d->pe_ln_g, d->pe_ln_b, M, E, eps, stream));
    }
    int cur = 0;                                                // X[cur] holds the stream x
    for (int s = 0; s < d->num_stages; ++s) {
        const psalm_swin_stage* st = &d->stages[s];
        const int C = g.C[s], Kp = c64(C), N4 = d->mlp_ratio * C, K4 = c64(N4), Hc = g.Hc[s], Wc = g.Wc[s];
        const int M = (int)g.rows[s], Mw = (int)g.rows_w[s];
        const int nWh = (Hc + wsz - 1) / wsz, nWw = (Wc + wsz - 1) / wsz;
        for (int b = 0; b < st->depth; ++b) {
            const psalm_swin_block* bl = &st->blocks[b];
            const int shift = (b % 2 == 0) ? 0 : wsz / 2;
            float* x = X[cur];
            float* xn = X[(cur + 1) % 3];
            float* xo = X[(cur + 2) % 3];
            SW(psalm_swin_window_gather_split(x, xw, xwinv, bl->n1_g, bl->n1_b, B, Hc, Wc, C, wsz, shift, eps, stream));
            SW(psalm_gemm_x3(xw, 2L * Kp, xwinv, bl->qkv_w, 2L * Kp, bl->qkv_ws, Kp, bl->qkv_b, nullptr, 0, qkv, 3L * C, Mw, 3 * C, 0, 0, gemm_workspace,
                             gemm_workspace_bytes, stream));
            if (Kp != C) SW(psalm_memset_zero(aw, (long)Mw * 2 * Kp * 2, stream));
            SW(psalm_window_attention_split(qkv, bl->rpb, xwinv, bl->qkv_bnd, aw, Kp, awinv, B, nWh, nWw, C, st->heads, wsz, shift, stream));
            SW(psalm_gemm_x3(aw, 2L * Kp, awinv, bl->proj_w, 2L * Kp, bl->proj_ws, Kp, bl->proj_b, nullptr, 0, pw, C, Mw, C, 0, 0, gemm_workspace,
                             gemm_workspace_bytes, stream));
            SW(psalm_swin_window_merge_ln_split(pw, x, xn, hsp, hinv, bl->n2_g, bl->n2_b, B, Hc, Wc, C, wsz, shift, eps, stream));
            if (K4 != N4) SW(psalm_memset_zero(hs, (long)M * 2 * K4 * 2, stream));
            SW(psalm_gemm_x3_split(hsp, 2L * Kp, hinv, bl->fc1_w, 2L * Kp, bl->fc1_ws, Kp, bl->fc1_b, nullptr, 0, M, N4, /*gelu (erf)*/ 2, 0, hs, 2L * K4, K4, 0, 0,
                                   bl->fc1_paired, hsinv, bl->fc1_bnd, 0, gemm_workspace, gemm_workspace_bytes, stream));
            SW(psalm_gemm_x3(hs, 2L * K4, hsinv, bl->fc2_w, 2L * K4, bl->fc2_ws, K4, bl->fc2_b, xn, C, xo, C, M, C, 0, 0, gemm_workspace, gemm_workspace_bytes,
                             stream));
            cur = (cur + 2) % 3;
        }
        float* x = X[cur];
        SW(psalm_layernorm3(x, PSALM_F32, C, outs_host[s], PSALM_F32, C, nullptr, 0, nullptr, 0, nullptr, 0, st->out_g, st->out_b, M, C, eps, stream));
        if (s + 1 < d->num_stages) {
            PSALM_CHECK_ARG(st->ds_w && st->ds_ws && st->ds_g && st->ds_b, "psalm_swin_forward: downsample weights missing");
            const int M2 = (int)g.rows[s + 1], K = 4 * C, Kq = c64(K), C2 = g.C[s + 1];
            SW(psalm_patch_merge_ln(x, PSALM_F32, cols, PSALM_F32, st->ds_g, st->ds_b, B, Hc, Wc, C, eps, stream));
            SW(psalm_split_f16(cols, K, colsp, 2L * Kq, colsinv, M2, K, stream));
            float* xn = X[(cur + 1) % 3];
            SW(psalm_gemm_x3(colsp, 2L * Kq, colsinv, st->ds_w, 2L * Kq, st->ds_ws, Kq, nullptr, nullptr, 0, xn, C2, M2, C2, 0, 0, gemm_workspace,
                             gemm_workspace_bytes, stream));
            cur = (cur + 1) % 3;
        }
    }
#undef SW
    return 0;
}

// ================================================================================================= projector
//   psalm_projector_forward   the conv projector (psalm/model/multimodal_projector/builder.py:365-375; BasicBlock :85-111 with eval BatchNorm
//                             folded into the convolution weights and `conv2` applied twice, as the reference does): 3x3 s2 conv + ReLU, 3x3 conv,
//                             1x1 s2 shortcut, 3x3 conv + shortcut + ReLU, linear.  Convolutions = im2col straight into split-f16 form + GEMM.
struct ProjLayout { long colsA, invA, y1, colsB, invB, y2, ds, y3, y3p, y3inv, total; int ho, wo; };
static ProjLayout proj_layout(const psalm_projector_desc* d, int B, int h, int w) {
    ProjLayout o;
    o.ho = (h + 2 - 3) / 2 + 1; o.wo = (w + 2 - 3) / 2 + 1;
    const long Mi = (long)B * o.ho * o.wo, Cin = d->in_dim, Cm = d->mid_dim;
    long p = 0;
    o.colsA = p; p += al256(Mi * 2 * c64((int)(9 * Cin)) * 2);   // patches of res5 (3x3) -- reused for the 1x1 shortcut patches
    o.invA = p; p += al256(Mi * 4);
    o.y1 = p; p += al256(Mi * Cm * 4);
    o.colsB = p; p += al256(Mi * 2 * c64((int)(9 * Cm)) * 2);
    o.invB = p; p += al256(Mi * 4);
    o.y2 = p; p += al256(Mi * Cm * 4);
    o.ds = p; p += al256(Mi * Cm * 4);
    o.y3 = p; p += al256(Mi * Cm * 4);
    o.y3p = p; p += al256(Mi * 2 * c64((int)Cm) * 2);
    o.y3inv = p; p += al256(Mi * 4);
    o.total = p;
    return o;
}
extern "C" long psalm_projector_forward_workspace(const psalm_projector_desc* d, int B, int h, int w) {
    if (!d || B <= 0 || h <= 0 || w <= 0 || d->in_dim % 8 || d->mid_dim % 8) return -1;
    return proj_layout(d, B, h, w).total;
}
extern "C" int psalm_projector_forward(const psalm_projector_desc* d, const float* res5, int B, int h, int w, float* out, void* workspace,
                                       long workspace_bytes, void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    PSALM_CHECK_ARG(d && res5 && out && workspace && B > 0 && d->in_dim % 8 == 0 && d->mid_dim % 8 == 0, "psalm_projector_forward: arguments (channel counts multiples of 8)");
    const ProjLayout lo = proj_layout(d, B, h, w);
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_projector_forward: workspace of psalm_projector_forward_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    const int Mi = B * lo.ho * lo.wo, Cin = d->in_dim, Cm = d->mid_dim, KA = 9 * Cin, KB = 9 * Cm;
    void* colsA = ws + lo.colsA; float* invA = (float*)(ws + lo.invA);
    void* colsB = ws + lo.colsB; float* invB = (float*)(ws + lo.invB);
    float* y1 = (float*)(ws + lo.y1); float* y2 = (float*)(ws + lo.y2); float* ds = (float*)(ws + lo.ds); float* y3 = (float*)(ws + lo.y3);
    void* y3p = ws + lo.y3p; float* y3inv = (float*)(ws + lo.y3inv);
    int rc;
#define PJ(call) do { rc = (call); if (rc) return rc; } while (0)
    PJ(psalm_im2col_split_f16(res5, colsA, invA, B, h, w, Cin, 3, 2, 1, stream));
    PJ(psalm_gemm_x3(colsA, 2L * c64(KA), invA, d->c1_w, 2L * c64(KA), d->c1_ws, c64(KA), d->c1_b, nullptr, 0, y1, Cm, Mi, Cm, /*relu*/ 1, 0, gemm_workspace, gemm_workspace_bytes, stream));
    PJ(psalm_im2col_split_f16(y1, colsB, invB, B, lo.ho, lo.wo, Cm, 3, 1, 1, stream));
    PJ(psalm_gemm_x3(colsB, 2L * c64(KB), invB, d->c2_w, 2L * c64(KB), d->c2_ws, c64(KB), nullptr, nullptr, 0, y2, Cm, Mi, Cm, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
    PJ(psalm_im2col_split_f16(res5, colsA, invA, B, h, w, Cin, 1, 2, 0, stream));
    PJ(psalm_gemm_x3(colsA, 2L * c64(Cin), invA, d->ds_w, 2L * c64(Cin), d->ds_ws, c64(Cin), d->ds_b, nullptr, 0, ds, Cm, Mi, Cm, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
    PJ(psalm_im2col_split_f16(y2, colsB, invB, B, lo.ho, lo.wo, Cm, 3, 1, 1, stream));
    PJ(psalm_gemm_x3(colsB, 2L * c64(KB), invB, d->c2f_w, 2L * c64(KB), d->c2f_ws, c64(KB), d->c2f_b, ds, Cm, y3, Cm, Mi, Cm, /*relu after the residual*/ 1 | 16, 0,
                     gemm_workspace, gemm_workspace_bytes, stream));
    PJ(psalm_split_f16(y3, Cm, y3p, 2L * c64(Cm), y3inv, Mi, Cm, stream));
    PJ(psalm_gemm_x3(y3p, 2L * c64(Cm), y3inv, d->fc_w, 2L * c64(Cm), d->fc_ws, c64(Cm), d->fc_b, nullptr, 0, out, d->out_dim, Mi, d->out_dim, 0, 0, gemm_workspace,
                     gemm_workspace_bytes, stream));
#undef PJ
    return 0;
}

// ================================================================================================= pixel decoder (one image)
//   psalm_pixel_decoder_forward   MSDeformAttnPixelDecoder.forward_features for ONE image (Mask2Former_Simplify/modeling/pixel_decoder/
//                                 msdeformattn.py:268-315; encoder layer :57-72, MSDeformAttn ops/modules/ms_deform_attn.py:82-124): input
//                                 projections + GroupNorm of res5 / res4 / res3 into one level-concatenated token buffer, N encoder layers
//                                 { value / [offsets | weights] GEMMs, fused deformable gather, output GEMM + residual + LayerNorm, FFN with the ReLU(linear1)
//                                 rows leaving as linear2's split operand, + residual + LayerNorm }, then the FPN step on res2 (adapter + GroupNorm,
//                                 up-sample + add, 3x3 conv + GroupNorm) and the mask-feature 1x1 convolution.  The op-by-op sequence of PSALM.pixel_decoder.
struct PdGeom { int h[4], w[4]; long start[3], S, HW2; };
static PdGeom pd_geom(const int* hw_host) {                    // hw_host: (h, w) of res2, res3, res4, res5
    PdGeom g;
    for (int i = 0; i < 4; ++i) { g.h[i] = hw_host[2 * i]; g.w[i] = hw_host[2 * i + 1]; }
    g.start[0] = 0;                                             // level order of the encoder: res5, res4, res3
    g.start[1] = (long)g.h[3] * g.w[3];
    g.start[2] = g.start[1] + (long)g.h[2] * g.w[2];
    g.S = g.start[2] + (long)g.h[1] * g.w[1];
    g.HW2 = (long)g.h[0] * g.w[0];
    return g;
}
struct PdLayout { long tokp, tokinv, t, gnws, src[2], x1, sp1, inv1, sp2, inv2, sp3, inv3, qin, value, ow, att, hdd, hddinv, lat, lat2, y, cols, colsinv, y2, total; };
static PdLayout pd_layout(const psalm_pd_desc* d, const PdGeom& g) {
    const long D = d->D, S = g.S, HW2 = g.HW2, KpD = c64((int)D), Kf = c64(d->ffn);
    long mtok = 0, mrows = 0;
    for (int i = 0; i < 4; ++i) {
        const long r = (long)g.h[i] * g.w[i];
        mtok = std::max(mtok, r * 2 * c64(d->in_dims[i]) * 2);
        mrows = std::max(mrows, r);
    }
    mrows = std::max(mrows, S);
    PdLayout o;
    long p = 0;
    o.tokp = p; p += al256(mtok);
    o.tokinv = p; p += al256(mrows * 4);
    o.t = p; p += al256(std::max(S, HW2) * D * 4);
    o.gnws = p; p += al256((std::max(S, HW2) / 64 + 2) * (long)d->G * 2 * 4);
    o.src[0] = p; p += al256(S * D * 4);
    o.src[1] = p; p += al256(S * D * 4);
    o.x1 = p; p += al256(S * D * 4);
    o.sp1 = p; p += al256(S * 2 * KpD * 2); o.inv1 = p; p += al256(S * 4);       // src_s (linear1 operand) / on-the-fly splits
    o.sp2 = p; p += al256(S * 2 * KpD * 2); o.inv2 = p; p += al256(S * 4);       // src_a (next layer's value operand)
    o.sp3 = p; p += al256(S * 2 * KpD * 2); o.inv3 = p; p += al256(S * 4);       // qin   (next layer's offset / weight operand)
    o.qin = p; p += al256(S * D * 4);
    o.value = p; p += al256(S * D * 4);
    o.ow = p; p += al256(S * (long)d->M * 3 * 4 * 3 * 4);
    o.att = p; p += al256(S * D * 4);
    o.hdd = p; p += al256(S * 2 * Kf * 2); o.hddinv = p; p += al256(S * 4);
    o.lat = p; p += al256(HW2 * D * 4);
    o.lat2 = p; p += al256(HW2 * D * 4);
    o.y = p; p += al256(HW2 * D * 4);
    o.cols = p; p += al256(HW2 * 2 * c64((int)(9 * D)) * 2); o.colsinv = p; p += al256(HW2 * 4);
    o.y2 = p; p += al256(HW2 * D * 4);
    o.total = p;
    return o;
}
static int pd_check(const psalm_pd_desc* d) {
    PSALM_CHECK_ARG(d && d->layers && d->num_layers >= 1 && d->D % 8 == 0 && d->D <= 2048 && d->D == d->M * 32 && d->ffn % 8 == 0 && d->mask_dim % 8 == 0,
                    "psalm_pixel_decoder_forward: descriptor (D = 32 * heads, multiples of 8)");
    for (int i = 0; i < 4; ++i) PSALM_CHECK_ARG(d->in_dims[i] % 8 == 0, "psalm_pixel_decoder_forward: input channel counts multiples of 8");
    return 0;
}
extern "C" long psalm_pixel_decoder_forward_workspace(const psalm_pd_desc* d, const int* hw_host) {
    if (pd_check(d) != 0 || !hw_host) return -1;
    return pd_layout(d, pd_geom(hw_host)).total;
}
extern "C" int psalm_pixel_decoder_forward(const psalm_pd_desc* d, const float* const* feats_host, const int* hw_host, const float* lvl_pos,
                                           float* mask_features, float* ms_out, void* workspace, long workspace_bytes, void* gemm_workspace,
                                           long gemm_workspace_bytes, void* stream) {
    if (pd_check(d) != 0) return -1;
    PSALM_CHECK_ARG(feats_host && hw_host && lvl_pos && mask_features && ms_out && workspace, "psalm_pixel_decoder_forward: null argument");
    const PdGeom g = pd_geom(hw_host);
    const PdLayout lo = pd_layout(d, g);
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_pixel_decoder_forward: workspace of psalm_pixel_decoder_forward_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    const int D = d->D, S = (int)g.S, HW2 = (int)g.HW2, KpD = c64(D), F = d->ffn, Kf = c64(F), NOW = d->M * 3 * 4 * 3;
    const float eps = 1e-5f;
    void* tokp = ws + lo.tokp; float* tokinv = (float*)(ws + lo.tokinv);
    float* t = (float*)(ws + lo.t); float* gnws = (float*)(ws + lo.gnws);
    float* src[2] = {(float*)(ws + lo.src[0]), (float*)(ws + lo.src[1])};
    float* x1 = (float*)(ws + lo.x1);
    void* sp1 = ws + lo.sp1; float* inv1 = (float*)(ws + lo.inv1);
    void* sp2 = ws + lo.sp2; float* inv2 = (float*)(ws + lo.inv2);
    void* sp3 = ws + lo.sp3; float* inv3 = (float*)(ws + lo.inv3);
    float* qin = (float*)(ws + lo.qin); float* value = (float*)(ws + lo.value); float* ow = (float*)(ws + lo.ow); float* att = (float*)(ws + lo.att);
    void* hdd = ws + lo.hdd; float* hddinv = (float*)(ws + lo.hddinv);
    float* lat = (float*)(ws + lo.lat); float* lat2 = (float*)(ws + lo.lat2); float* y = (float*)(ws + lo.y); float* y2 = (float*)(ws + lo.y2);
    void* cols = ws + lo.cols; float* colsinv = (float*)(ws + lo.colsinv);
    int rc;
#define PD(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- input projections of res5, res4, res3 (encoder level order) + GroupNorm, written into the level-concatenated buffer
    float* s0 = src[0];
    for (int l = 0; l < 3; ++l) {
        const int fi = 3 - l, hwl = g.h[fi] * g.w[fi], Cin = d->in_dims[fi], Kp = c64(Cin);
        PD(psalm_split_f16(feats_host[fi], Cin, tokp, 2L * Kp, tokinv, hwl, Cin, stream));
        PD(psalm_gemm_x3(tokp, 2L * Kp, tokinv, d->ip_w[l], 2L * Kp, d->ip_ws[l], Kp, d->ip_b[l], nullptr, 0, t, D, hwl, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        PD(psalm_groupnorm_nhwc(t, PSALM_F32, s0 + g.start[l] * D, PSALM_F32, d->ip_gn_g[l], d->ip_gn_b[l], gnws, 1, hwl, D, d->G, eps, 0, stream));
    }
    PD(psalm_add_bcast(s0, PSALM_F32, lvl_pos, PSALM_F32, qin, PSALM_F32, S, D, S, stream));
    const int64_t shapes[6] = {g.h[3], g.w[3], g.h[2], g.w[2], g.h[1], g.w[1]};
    const int64_t starts[3] = {g.start[0], g.start[1], g.start[2]};
    int cur = 0;
    for (int i = 0; i < d->num_layers; ++i) {
        const psalm_pd_enc_layer* ly = &d->layers[i];
        const bool more = i + 1 < d->num_layers;
        float* sc = src[cur];
        float* sn = src[cur ^ 1];
        if (i == 0) {                                            // first layer: fp32 operands, split on the fly (as Ops.gemm does)
            PD(psalm_split_f16(sc, D, sp2, 2L * KpD, inv2, S, D, stream));
            PD(psalm_gemm_x3(sp2, 2L * KpD, inv2, ly->value_w, 2L * KpD, ly->value_ws, KpD, ly->value_b, nullptr, 0, value, D, S, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
            PD(psalm_split_f16(qin, D, sp3, 2L * KpD, inv3, S, D, stream));
            PD(psalm_gemm_x3(sp3, 2L * KpD, inv3, ly->ow_w, 2L * KpD, ly->ow_ws, KpD, ly->ow_b, nullptr, 0, ow, NOW, S, NOW, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        } else {
            PD(psalm_gemm_x3(sp2, 2L * KpD, inv2, ly->value_w, 2L * KpD, ly->value_ws, KpD, ly->value_b, nullptr, 0, value, D, S, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
            PD(psalm_gemm_x3(sp3, 2L * KpD, inv3, ly->ow_w, 2L * KpD, ly->ow_ws, KpD, ly->ow_b, nullptr, 0, ow, NOW, S, NOW, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        }
        PD(psalm_msda_fused(value, PSALM_F32, shapes, starts, ow, att, PSALM_F32, 1, S, d->M, 32, 3, 4, stream));
        PD(psalm_split_f16(att, D, sp1, 2L * KpD, inv1, S, D, stream));
        PD(psalm_gemm_x3(sp1, 2L * KpD, inv1, ly->out_w, 2L * KpD, ly->out_ws, KpD, ly->out_b, sc, D, x1, D, S, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        // norm1: fp32 stream (-> sn) + linear1's split operand (sp1)
        PD(psalm_layernorm_split(x1, D, sn, D, ly->n1_g, ly->n1_b, S, D, eps, sp1, inv1, nullptr, 0, nullptr, nullptr, stream));
        if (Kf != F) PD(psalm_memset_zero(hdd, (long)S * 2 * Kf * 2, stream));
        PD(psalm_gemm_x3_split(sp1, 2L * KpD, inv1, ly->l1_w, 2L * KpD, ly->l1_ws, KpD, ly->l1_b, nullptr, 0, S, F, /*relu*/ 1, 0, hdd, 2L * Kf, Kf, 0, 0, ly->l1_paired, hddinv,
                               ly->l1_bnd, 0, gemm_workspace, gemm_workspace_bytes, stream));
        PD(psalm_gemm_x3(hdd, 2L * Kf, hddinv, ly->l2_w, 2L * Kf, ly->l2_ws, Kf, ly->l2_b, sn, D, x1, D, S, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        // norm2: fp32 stream (-> sc: the layer's input buffer is free again) + the NEXT layer's operands: split(src) and split(src + pos)
        PD(psalm_layernorm_split(x1, D, sc, D, ly->n2_g, ly->n2_b, S, D, eps, more ? sp2 : nullptr, more ? inv2 : nullptr, more ? lvl_pos : nullptr, more ? S : 0,
                                 more ? sp3 : nullptr, more ? inv3 : nullptr, stream));
        // (the stream stays in src[cur])
    }
    PD(psalm_copy_d2d(ms_out, src[cur], (long)S * D * 4, stream));
    // ---- FPN step on res2 + mask features
    {
        const int C2 = d->in_dims[0], Kp2 = c64(C2), hs = g.h[1], ws_ = g.w[1], H2 = g.h[0], W2 = g.w[0], K9 = 9 * D, Kp9 = c64(K9);
        PD(psalm_split_f16(feats_host[0], C2, tokp, 2L * Kp2, tokinv, HW2, C2, stream));
        PD(psalm_gemm_x3(tokp, 2L * Kp2, tokinv, d->adapter_w, 2L * Kp2, d->adapter_ws, Kp2, d->adapter_b, nullptr, 0, lat, D, HW2, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        PD(psalm_groupnorm_nhwc(lat, PSALM_F32, lat2, PSALM_F32, d->adapter_gn_g, d->adapter_gn_b, gnws, 1, HW2, D, d->G, eps, 1, stream));
        PD(psalm_upsample_add_nhwc(lat2, PSALM_F32, src[cur] + g.start[2] * D, PSALM_F32, y, PSALM_F32, 1, hs, ws_, H2, W2, D, stream));
        PD(psalm_im2col_split_f16(y, cols, colsinv, 1, H2, W2, D, 3, 1, 1, stream));
        PD(psalm_gemm_x3(cols, 2L * Kp9, colsinv, d->layer_w, 2L * Kp9, d->layer_ws, Kp9, d->layer_b, nullptr, 0, y2, D, HW2, D, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        PD(psalm_groupnorm_nhwc(y2, PSALM_F32, y, PSALM_F32, d->layer_gn_g, d->layer_gn_b, gnws, 1, HW2, D, d->G, eps, 1, stream));
        PD(psalm_split_f16(y, D, cols, 2L * KpD, colsinv, HW2, D, stream));
        PD(psalm_gemm_x3(cols, 2L * KpD, colsinv, d->mf_w, 2L * KpD, d->mf_ws, KpD, d->mf_b, nullptr, 0, mask_features, d->mask_dim, HW2, d->mask_dim, 0, 0, gemm_workspace,
                         gemm_workspace_bytes, stream));
    }
#undef PD
    return 0;
}

// ================================================================================================= masked-attention decoder (one image)
//   psalm_predictor_forward   MultiScaleMaskedTransformerDecoder.forward for ONE image (Mask2Former_Simplify/modeling/transformer_decoder/
//                             mask2former_transformer_decoder.py:596-693; prediction heads :695-762; attention / FFN layers :19-199): the K / V projections
//                             of the three levels (all decoder layers of a level stacked into one GEMM), then per layer { thresholded attention mask from the
//                             current mask logits; masked cross-attention; self-attention; FFN; mask head: decoder norm + mask_embed MLP + the
//                             (Q x H2*W2) mask GEMM }, and after the last layer the class / SEG / region logits.  precision "f16x3": the M = Q GEMMs run on
//                             the exact-fp32 skinny kernel, the level projections and the mask GEMM in split-f16 arithmetic.  The op-by-op sequence of
//                             PSALM.predictor.
struct PrLayout { long kin, vin, sp, spinv, K[3], V[3], mfp, mfinv, dec, me0, me1, me2, mes, mesinv, masks, amask, flags, outq, qp, a, x1, out[2], qk, v, hdd, mha, t0, t1, total; };
static PrLayout pr_layout(const psalm_pr_desc* d, const int* hwl, int H2, int W2, int n_extra) {
    const long D = d->D, Q = d->Q, HW2 = (long)H2 * W2;
    long mhw = 0, mmha = 0;
    for (int l = 0; l < d->num_levels; ++l) {
        const long hw = (long)hwl[2 * l] * hwl[2 * l + 1];
        mhw = std::max(mhw, hw);
        mmha = std::max(mmha, psalm_mha_attention_f32_workspace(1, d->heads, (int)Q, (int)hw));
    }
    mmha = std::max(mmha, psalm_mha_attention_f32_workspace(1, d->heads, (int)Q, (int)Q));
    PrLayout o;
    long p = 0;
    o.kin = p; p += al256(mhw * D * 4);
    o.vin = p; p += al256(mhw * D * 4);
    o.sp = p; p += al256(mhw * 2 * c64((int)D) * 2); o.spinv = p; p += al256(mhw * 4);
    for (int l = 0; l < 3; ++l) {
        const long hw = l < d->num_levels ? (long)hwl[2 * l] * hwl[2 * l + 1] : 0;
        const long nl_l = l < d->num_levels ? (d->num_layers - l + d->num_levels - 1) / d->num_levels : 0;
        o.K[l] = p; p += al256(hw * nl_l * D * 4);
        o.V[l] = p; p += al256(hw * nl_l * D * 4);
    }
    o.mfp = p; p += al256(HW2 * 2 * c64(d->mask_dim) * 2); o.mfinv = p; p += al256(HW2 * 4);
    o.dec = p; p += al256(Q * D * 4);
    o.me0 = p; p += al256(Q * D * 4); o.me1 = p; p += al256(Q * D * 4); o.me2 = p; p += al256(Q * std::max<long>(D, d->mask_dim) * 4);
    o.mes = p; p += al256(Q * 2 * c64(d->mask_dim) * 2); o.mesinv = p; p += al256(Q * 4);
    o.masks = p; p += al256(Q * HW2 * 4);
    o.amask = p; p += al256(Q * mhw); o.flags = p; p += al256(Q);
    o.outq = p; p += al256(Q * D * 4); o.qp = p; p += al256(Q * D * 4); o.a = p; p += al256(Q * D * 4); o.x1 = p; p += al256(Q * D * 4);
    o.out[0] = p; p += al256(Q * D * 4); o.out[1] = p; p += al256(Q * D * 4);
    o.qk = p; p += al256(Q * 2 * D * 4); o.v = p; p += al256(Q * D * 4);
    o.hdd = p; p += al256(Q * (long)d->ffn * 4);
    o.mha = p; p += al256(mmha);
    o.t0 = p; p += al256(std::max<long>(Q, n_extra) * D * 4); o.t1 = p; p += al256(std::max<long>(Q, n_extra) * D * 4);
    o.total = p;
    return o;
}
static int pr_check(const psalm_pr_desc* d) {
    PSALM_CHECK_ARG(d && d->layers && d->num_layers >= 1 && d->num_levels >= 1 && d->num_levels <= 3 && d->Q >= 1 && d->Q <= 128 && d->D == 32 * d->heads &&
                        d->D % 8 == 0 && d->mask_dim % 8 == 0 && d->ffn % 8 == 0, "psalm_predictor_forward: descriptor (Q <= 128, D = 32 * heads, <= 3 levels)");
    return 0;
}
extern "C" long psalm_predictor_forward_workspace(const psalm_pr_desc* d, const int* hw_levels_host, int H2, int W2, int n_extra_rows) {
    if (pr_check(d) != 0 || !hw_levels_host || H2 <= 0 || W2 <= 0) return -1;
    return pr_layout(d, hw_levels_host, H2, W2, n_extra_rows).total;
}
// The part of psalm_predictor_forward that needs nothing from the LLM: the K / V projections of the three levels (every decoder layer of a level in one
// GEMM) and the split of the mask features -- into the SAME workspace the forward call is then given with kv_ready = 1.  r06: model.py issues it on
// the side stream right behind the pixel decoder, beside the LLM (~0.2 ms of launches leave the critical path: 6 adds, 7 splits, 6 GEMMs); the
// workspace prefix it writes (PrLayout kin .. mfinv) does not depend on the region count.
static int predictor_kv_impl(const psalm_pr_desc* d, const PrLayout& lo, const float* const* ms_host, const int* hw_levels_host, const float* const* prpos_host,
                             const float* mask_features, int H2, int W2, char* ws, void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    const int D = d->D, nl = d->num_layers, nlev = d->num_levels, HW2 = H2 * W2, MD = d->mask_dim;
    float* kin = (float*)(ws + lo.kin); float* vin = (float*)(ws + lo.vin);
    void* sp = ws + lo.sp; float* spinv = (float*)(ws + lo.spinv);
    const int KpD = c64(D), KpM = c64(MD);
    int rc;
#define PR(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- K / V of the three levels: (level tokens + position + level embedding) . Wk^T, (level tokens + level embedding) . Wv^T, every decoder layer of the level in one GEMM
    for (int l = 0; l < nlev; ++l) {
        const int hw = hw_levels_host[2 * l] * hw_levels_host[2 * l + 1];
        const int N = ((nl - l + nlev - 1) / nlev) * D;
        float* Kl = (float*)(ws + lo.K[l]); float* Vl = (float*)(ws + lo.V[l]);
        PR(psalm_add_bcast(ms_host[l], PSALM_F32, prpos_host[l], PSALM_F32, kin, PSALM_F32, hw, D, hw, stream));
        PR(psalm_add_bcast(ms_host[l], PSALM_F32, d->level_embed + (long)l * D, PSALM_F32, vin, PSALM_F32, hw, D, 1, stream));
        PR(psalm_split_f16(kin, D, sp, 2L * KpD, spinv, hw, D, stream));
        PR(psalm_gemm_x3(sp, 2L * KpD, spinv, d->lvl_k_w[l], 2L * KpD, d->lvl_k_ws[l], KpD, d->lvl_k_b[l], nullptr, 0, Kl, N, hw, N, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
        PR(psalm_split_f16(vin, D, sp, 2L * KpD, spinv, hw, D, stream));
        PR(psalm_gemm_x3(sp, 2L * KpD, spinv, d->lvl_v_w[l], 2L * KpD, d->lvl_v_ws[l], KpD, d->lvl_v_b[l], nullptr, 0, Vl, N, hw, N, 0, 0, gemm_workspace, gemm_workspace_bytes, stream));
    }
    // the mask features as the W operand of the 1 + num_layers mask GEMMs: split once when they are large (PSALM._wop: > 4096 rows), else exact fp32
    if (HW2 > 4096) PR(psalm_split_f16(mask_features, MD, ws + lo.mfp, 2L * KpM, (float*)(ws + lo.mfinv), HW2, MD, stream));
#undef PR
    return 0;
}
extern "C" int psalm_predictor_kv(const psalm_pr_desc* d, const float* const* ms_host, const int* hw_levels_host, const float* const* prpos_host,
                                  const float* mask_features, int H2, int W2, int n_extra_rows, void* workspace, long workspace_bytes,
                                  void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    if (pr_check(d) != 0) return -1;
    PSALM_CHECK_ARG(ms_host && hw_levels_host && prpos_host && mask_features && workspace, "psalm_predictor_kv: null argument");
    const PrLayout lo = pr_layout(d, hw_levels_host, H2, W2, std::max(n_extra_rows, 0));
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_predictor_kv: workspace of psalm_predictor_forward_workspace() bytes, 256-byte aligned");
    return predictor_kv_impl(d, lo, ms_host, hw_levels_host, prpos_host, mask_features, H2, W2, (char*)workspace, gemm_workspace, gemm_workspace_bytes, stream);
}
extern "C" int psalm_predictor_forward(const psalm_pr_desc* d, const float* const* ms_host, const int* hw_levels_host, const float* const* prpos_host,
                                       const float* mask_features, int H2, int W2, const float* seg_query, const float* class_emb, int n_cls,
                                       const float* seg_emb, int n_seg, const float* region_emb, int n_reg, float* pred_masks, float* cls_logits,
                                       float* seg_logits, float* region_logits, void* workspace, long workspace_bytes, void* gemm_workspace,
                                       long gemm_workspace_bytes, int kv_ready, void* stream) {
    if (pr_check(d) != 0) return -1;
    PSALM_CHECK_ARG(ms_host && hw_levels_host && prpos_host && mask_features && seg_query && pred_masks && workspace, "psalm_predictor_forward: null argument");
    const int D = d->D, Q = d->Q, nh = d->heads, nl = d->num_layers, nlev = d->num_levels, HW2 = H2 * W2, MD = d->mask_dim, F = d->ffn;
    const PrLayout lo = pr_layout(d, hw_levels_host, H2, W2, std::max(n_reg, 0));
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_predictor_forward: workspace of psalm_predictor_forward_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    const float eps = 1e-5f;
    float* Kl[3]; float* Vl[3];
    for (int l = 0; l < 3; ++l) { Kl[l] = (float*)(ws + lo.K[l]); Vl[l] = (float*)(ws + lo.V[l]); }
    void* mfp = ws + lo.mfp; float* mfinv = (float*)(ws + lo.mfinv);
    float* dec = (float*)(ws + lo.dec); float* me0 = (float*)(ws + lo.me0); float* me1 = (float*)(ws + lo.me1); float* me2 = (float*)(ws + lo.me2);
    void* mes = ws + lo.mes; float* mesinv = (float*)(ws + lo.mesinv);
    float* masks = (float*)(ws + lo.masks);
    unsigned char* amask = (unsigned char*)(ws + lo.amask); unsigned char* flags = (unsigned char*)(ws + lo.flags);
    float* outq = (float*)(ws + lo.outq); float* qp = (float*)(ws + lo.qp); float* a = (float*)(ws + lo.a); float* x1 = (float*)(ws + lo.x1);
    float* outb[2] = {(float*)(ws + lo.out[0]), (float*)(ws + lo.out[1])};
    float* qk = (float*)(ws + lo.qk); float* v = (float*)(ws + lo.v); float* hdd = (float*)(ws + lo.hdd);
    void* mha = ws + lo.mha;
    float* t0 = (float*)(ws + lo.t0); float* t1 = (float*)(ws + lo.t1);
    int rc;
#define PR(call) do { rc = (call); if (rc) return rc; } while (0)
    // exact-fp32 GEMM of the M = Q (or a handful of prompt) rows: psalm_gemm with float32 operands (skinny kernel)
    auto g32 = [&](const float* A, int M, int K, const float* W, const float* b, const float* res, float* C, int N, int act) -> int {
        return psalm_gemm(A, PSALM_F32, K, W, PSALM_F32, K, b, res, res ? N : 0, C, PSALM_F32, N, M, N, K, act, 0, gemm_workspace, gemm_workspace_bytes, stream);
    };
    auto ln = [&](const float* x, const float* g_, const float* b_, float* y, int rows) -> int {
        return psalm_layernorm3(x, PSALM_F32, D, y, PSALM_F32, D, nullptr, 0, nullptr, 0, nullptr, 0, g_, b_, rows, D, eps, stream);
    };
    const int KpM = c64(MD);
    // ---- K / V of the three levels + the split of the mask features: here, or already in the workspace (psalm_predictor_kv)
    int nl_l[3] = {0, 0, 0};
    for (int l = 0; l < nlev; ++l) nl_l[l] = (nl - l + nlev - 1) / nlev;
    if (!kv_ready) PR(predictor_kv_impl(d, lo, ms_host, hw_levels_host, prpos_host, mask_features, H2, W2, ws, gemm_workspace, gemm_workspace_bytes, stream));
    const bool mf_split = HW2 > 4096;
    // (fused LayerNorm chain / paired projections: D % 8 == 0, the skinny GEMM's M <= 192; PSALM_TUNE_DECODER_FUSE switches them off)
    const bool fuse = D % 8 == 0 && D <= 2048 && Q <= 192 && psalm_get_tuning(PSALM_TUNE_DECODER_FUSE) != 0;
    auto mask_head = [&](const float* out_, bool have_dec = false) -> int {   // decoder_norm -> mask_embed MLP -> (Q, H2*W2) mask logits; `dec` stays for the class heads
        int r;
        if (!have_dec && (r = ln(out_, d->dn_g, d->dn_b, dec, Q))) return r;
        if ((r = g32(dec, Q, D, d->mask_embed_w[0], d->mask_embed_b[0], nullptr, me0, D, 1))) return r;
        if ((r = g32(me0, Q, D, d->mask_embed_w[1], d->mask_embed_b[1], nullptr, me1, D, 1))) return r;
        if ((r = g32(me1, Q, D, d->mask_embed_w[2], d->mask_embed_b[2], nullptr, me2, MD, 0))) return r;
        if (mf_split) {
            if ((r = psalm_split_f16(me2, MD, mes, 2L * KpM, mesinv, Q, MD, stream))) return r;
            return psalm_gemm_x3(mes, 2L * KpM, mesinv, mfp, 2L * KpM, mfinv, KpM, nullptr, nullptr, 0, masks, HW2, Q, HW2, 0, 0, gemm_workspace, gemm_workspace_bytes, stream);
        }
        return g32(me2, Q, MD, mask_features, nullptr, nullptr, masks, HW2, 0);
    };
    const float* out = seg_query;
    PR(mask_head(out));
    PR(psalm_add_bcast(out, PSALM_F32, d->query_embed, PSALM_F32, outq, PSALM_F32, Q, D, Q, stream));
    int cur = 0;
    for (int i = 0; i < nl; ++i) {
        const psalm_pr_layer* ly = &d->layers[i];
        const int l = i % nlev, j = i / nlev, h = hw_levels_host[2 * l], w = hw_levels_host[2 * l + 1], hw = h * w, N = nl_l[l] * D;
        PR(psalm_attn_mask(masks, amask, flags, Q, H2, W2, h, w, stream));
        PR(g32(outq, Q, D, ly->cq_w, ly->cq_b, nullptr, qp, D, 0));
        PR(psalm_mha_attention_f32(qp, D, Kl[l] + (long)j * D, N, Vl[l] + (long)j * D, N, a, D, amask, flags, mha, 1, Q, hw, nh, 32, stream));
        PR(g32(a, Q, D, ly->co_w, ly->co_b, out, x1, D, 0));
        float* o1 = outb[cur];
        // (r06: the post-norm and `+ query_pos` as one launch; the two self-attention projections as one launch -- the same words as the r05 sequence
        //  psalm_layernorm3, psalm_add_bcast, psalm_gemm x 2, which model.py's op-by-op path still issues and the stage test compares against)
        if (fuse) {
            PR(psalm_layernorm_chain(x1, D, o1, D, ly->cn_g, ly->cn_b, d->query_embed, Q, outq, D, nullptr, nullptr, nullptr, 0, Q, D, eps, stream));
            PR(psalm_gemm_f32_pair(outq, ly->sqk_w, ly->sqk_b, qk, Q, 2 * D, D, 0, o1, ly->sv_w, ly->sv_b, v, Q, D, D, 0, stream));
        } else {
            PR(ln(x1, ly->cn_g, ly->cn_b, o1, Q));
            PR(psalm_add_bcast(o1, PSALM_F32, d->query_embed, PSALM_F32, outq, PSALM_F32, Q, D, Q, stream));
            PR(g32(outq, Q, D, ly->sqk_w, ly->sqk_b, nullptr, qk, 2 * D, 0));
            PR(g32(o1, Q, D, ly->sv_w, ly->sv_b, nullptr, v, D, 0));
        }
        PR(psalm_mha_attention_f32(qk, 2L * D, qk + D, 2L * D, v, D, a, D, nullptr, nullptr, mha, 1, Q, Q, nh, 32, stream));
        PR(g32(a, Q, D, ly->so_w, ly->so_b, o1, x1, D, 0));
        float* o2 = outb[cur ^ 1];
        PR(ln(x1, ly->sn_g, ly->sn_b, o2, Q));
        PR(g32(o2, Q, D, ly->f1_w, ly->f1_b, nullptr, hdd, F, 1));
        PR(g32(hdd, Q, F, ly->f2_w, ly->f2_b, o2, x1, D, 0));
        if (fuse) {                                              // FFN post-norm, `+ query_pos` for the next layer and the mask head's decoder_norm: one launch
            PR(psalm_layernorm_chain(x1, D, o1, D, ly->fn_g, ly->fn_b, d->query_embed, Q, outq, D, d->dn_g, d->dn_b, dec, D, Q, D, eps, stream));
        } else {
            PR(ln(x1, ly->fn_g, ly->fn_b, o1, Q));
            PR(psalm_add_bcast(o1, PSALM_F32, d->query_embed, PSALM_F32, outq, PSALM_F32, Q, D, Q, stream));
        }
        out = o1;
        PR(mask_head(out, fuse));
        cur ^= 1;
    }
    PR(psalm_copy_d2d(pred_masks, masks, (long)Q * HW2 * 4, stream));
    // ---- prediction heads of the LAST layer (the earlier ones are auxiliary training outputs, TD:672-690)
    if (class_emb && n_cls > 0) {
        PSALM_CHECK_ARG(cls_logits != nullptr, "psalm_predictor_forward: cls_logits output missing");
        PR(g32(dec, Q, D, d->CLASS_w[0], d->CLASS_b[0], nullptr, t0, D, 1));
        PR(g32(t0, Q, D, d->CLASS_w[1], d->CLASS_b[1], nullptr, t1, D, 0));
        PR(g32(t1, Q, D, class_emb, nullptr, nullptr, cls_logits, n_cls, 0));
    }
    if (seg_emb && n_seg > 0) {
        PSALM_CHECK_ARG(seg_logits != nullptr, "psalm_predictor_forward: seg_logits output missing");
        PR(g32(dec, Q, D, d->SEG_w[0], d->SEG_b[0], nullptr, t0, D, 1));
        PR(g32(t0, Q, D, d->SEG_w[1], d->SEG_b[1], nullptr, t1, D, 0));
        PR(g32(t1, Q, D, seg_emb, nullptr, nullptr, seg_logits, n_seg, 0));
    }
    if (region_emb && n_reg > 0) {                               // einsum 'kd,ld->kl' (TD:744): (k, Q)
        PSALM_CHECK_ARG(region_logits != nullptr, "psalm_predictor_forward: region_logits output missing");
        PR(g32(dec, Q, D, d->REGION_w[0], d->REGION_b[0], nullptr, t0, D, 1));
        PR(g32(t0, Q, D, d->REGION_w[1], d->REGION_b[1], nullptr, t1, D, 0));
        PR(g32(region_emb, n_reg, D, t1, nullptr, nullptr, region_logits, Q, 0));
    }
#undef PR
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------ post-processing
// psalm_postprocess: llava_phi.py:1401-1466 for one image (see psalm_hip.h), the launch sequence of PSALM._post_head + PSALM._post_tail.
struct PostLayout { long up, semp, probs, probsT, score, label, mscore, msws, cl, qq, argq, pcnt, fid, total; };
static PostLayout post_layout(const psalm_post_desc* d, bool have_up) {
    PostLayout o;
    const long Q = d->Q, HWp = (long)d->Hpad * d->Wpad, HWo = (long)d->out_h * d->out_w, C1 = std::max(d->C1, 1);
    const bool resize_after = !(d->crop_h == d->Hpad && d->crop_w == d->Wpad && d->out_h == d->Hpad && d->out_w == d->Wpad);
    long p = 0;
    o.up = p; if (!have_up && resize_after && d->task != PSALM_POST_SEMANTIC) p += al256(Q * HWp * 4);          // up-sampled logits in front of the crop / resize
    o.semp = p; if (resize_after && d->task == PSALM_POST_SEMANTIC) p += al256((C1 - 1) * HWp * 4);              // semantic: the class map at the padded size
    o.probs = p; p += al256(Q * C1 * 4);
    o.probsT = p; p += al256(C1 * 128 * 4);
    o.score = p; p += al256(Q * 4);
    o.label = p; p += al256(Q * 4);
    o.mscore = p; p += al256(Q * 4);
    o.msws = p; p += al256(Q * 512 * 2 * 4);
    o.cl = p; p += al256(Q * 4);
    o.qq = p; p += al256(Q * 4);
    o.argq = p; p += al256(std::max(HWo, HWp) * 4);
    o.pcnt = p; p += al256(Q * 3 * 4);
    o.fid = p; p += al256((Q + C1 + 1) * 4);
    o.total = p;
    return o;
}
static int post_check(const psalm_post_desc* d) {
    PSALM_CHECK_ARG(d && d->task >= PSALM_POST_SEMANTIC && d->task <= PSALM_POST_REGION, "psalm_postprocess: descriptor / task code");
    PSALM_CHECK_ARG(d->Q > 0 && d->Q <= 128 && d->h > 0 && d->w > 0 && d->Hpad > 0 && d->Wpad > 0 && d->crop_h > 0 && d->crop_w > 0 && d->crop_h <= d->Hpad &&
                        d->crop_w <= d->Wpad && d->out_h > 0 && d->out_w > 0,
                    "psalm_postprocess: Q <= 128, positive geometry, crop box inside the padded image");
    if (d->task <= PSALM_POST_PANOPTIC)
        PSALM_CHECK_ARG(d->C1 >= 2 && (d->task == PSALM_POST_INSTANCE || d->C1 - 1 <= 160),
                        "psalm_postprocess: class logits of C1 >= 2 columns; at most 160 classes where the class map is formed (op-level calls beyond)");
    if (d->task == PSALM_POST_REGION) PSALM_CHECK_ARG(d->n_region > 0, "psalm_postprocess: region task needs n_region > 0");
    return 0;
}
extern "C" long psalm_postprocess_workspace(const psalm_post_desc* d, int have_mask_up) {
    if (post_check(d) != 0) return -1;
    return post_layout(d, have_mask_up != 0).total;
}
extern "C" int psalm_postprocess(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace, long workspace_bytes, void* stream) {
    if (post_check(d) != 0) return -1;
    PSALM_CHECK_ARG(io && (io->pred_masks || io->mask_up) && workspace, "psalm_postprocess: null argument");
    const bool have_up = io->mask_up != nullptr;
    const PostLayout lo = post_layout(d, have_up);
    PSALM_CHECK_ARG(workspace_bytes >= lo.total && (uintptr_t)workspace % 256 == 0, "psalm_postprocess: workspace of psalm_postprocess_workspace() bytes, 256-byte aligned");
    char* ws = (char*)workspace;
    const int Q = d->Q, C1 = d->C1, task = d->task, Kpad = (Q + 63) / 64 * 64;
    const long HWp = (long)d->Hpad * d->Wpad, HWo = (long)d->out_h * d->out_w;
    const bool resize_after = !(d->crop_h == d->Hpad && d->crop_w == d->Wpad && d->out_h == d->Hpad && d->out_w == d->Wpad);
    int rc;
#define PO(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- PSALM._post_head: the mask logits at the padded image size, the class softmax
    const float* up = io->mask_up;
    if (!up) {
        const bool to_out = !resize_after || task == PSALM_POST_SEMANTIC;        // (semantic: `mask_pred` is the padded-size tensor, LP:1437-1440)
        PSALM_CHECK_ARG(!to_out || io->mask_pred, "psalm_postprocess: mask_pred buffer missing");
        float* dst = to_out ? io->mask_pred : (float*)(ws + lo.up);
        PO(psalm_resize_planes(io->pred_masks, PSALM_F32, dst, PSALM_F32, Q, d->h, d->w, d->h, d->w, d->Hpad, d->Wpad, stream));      // LP:1401-1406
        up = dst;
    }
    float* probs = (float*)(ws + lo.probs); float* probsT = (float*)(ws + lo.probsT); float* score = (float*)(ws + lo.score);
    int* label = (int*)(ws + lo.label);
    if (task <= PSALM_POST_PANOPTIC) {
        PSALM_CHECK_ARG(io->cls_logits, "psalm_postprocess: cls_logits missing");
        PO(psalm_memset_zero(probsT, (long)(C1 - 1) * Kpad * 4, stream));
        PO(psalm_class_softmax(io->cls_logits, probs, probsT, PSALM_F32, score, label, Q, C1, Kpad, stream));
    }
    // ---- PSALM._post_tail
    const float* mp = up;
    long HW = HWp;
    int mh = d->Hpad, mw = d->Wpad;
    if (resize_after && task != PSALM_POST_SEMANTIC) {                                  // sem_seg_postprocess (LP:1427-1429)
        PSALM_CHECK_ARG(io->mask_pred, "psalm_postprocess: mask_pred buffer missing");
        PO(psalm_resize_planes(up, PSALM_F32, io->mask_pred, PSALM_F32, Q, d->Hpad, d->Wpad, d->crop_h, d->crop_w, d->out_h, d->out_w, stream));
        mp = io->mask_pred; HW = HWo; mh = d->out_h; mw = d->out_w;
    }
    if (mask_pred_is) *mask_pred_is = (mp == io->mask_up && io->mask_up) ? 1 : 0;
    float* mscore = (float*)(ws + lo.mscore); float* msws = (float*)(ws + lo.msws);
    int* cl = (int*)(ws + lo.cl); int* qq = (int*)(ws + lo.qq);
    auto zero_topk = [&]() -> int {
        int r;
        if ((r = psalm_memset_zero(io->scores, (long)Q * 4, stream))) return r;
        return psalm_memset_zero(cl, (lo.qq - lo.cl) + (long)Q * 4, stream);          // class and query vectors: adjacent in the workspace, one fill
    };
    if (task == PSALM_POST_SEMANTIC) {                                                  // on the padded-size masks, post-processed AFTER inference (LP:301,1437-1440)
        PSALM_CHECK_ARG(io->sem_seg && Kpad == 128, "psalm_postprocess: sem_seg buffer; Q in (64, 128]");
        float* sem = resize_after ? (float*)(ws + lo.semp) : io->sem_seg;
        PO(psalm_semantic_from_masks_x3(mp, probsT, sem, nullptr, nullptr, Q, C1 - 1, HW, Kpad, stream));                  // LP:402-406
        if (resize_after)
            PO(psalm_resize_planes(sem, PSALM_F32, io->sem_seg, PSALM_F32, C1 - 1, d->Hpad, d->Wpad, d->crop_h, d->crop_w, d->out_h, d->out_w, stream));
        if (mask_pred_is) *mask_pred_is = io->mask_up ? 1 : 0;
        return 0;
    }
    PSALM_CHECK_ARG(io->scores && io->inst_masks && io->boxes && io->counts, "psalm_postprocess: scores / inst_masks / boxes / counts buffers");
    if (task == PSALM_POST_INSTANCE) {
        PSALM_CHECK_ARG(io->classes && io->query, "psalm_postprocess: classes / query buffers");
        PO(psalm_mask_scores(mp, mscore, msws, Q, HW, stream));
        PO(zero_topk());
        PO(psalm_memset_zero(io->counts, 4, stream));
        PO(psalm_topk_select(probs, Q, C1 - 1, C1, Q, nullptr, mscore, io->scores, cl, qq, io->counts, 0, stream));
        PO(psalm_cast_i32_i64(cl, io->classes, Q, stream));
        PO(psalm_cast_i32_i64(qq, io->query, Q, stream));
        PO(psalm_binarize_gather(mp, qq, io->counts, io->inst_masks, Q, HW, stream));
        PO(psalm_memset_zero(io->boxes, (long)Q * 4 * 4, stream));
    } else if (task == PSALM_POST_PANOPTIC) {
        PSALM_CHECK_ARG(io->classes && io->query && io->sem_seg && io->pan && io->is_thing && Kpad == 128, "psalm_postprocess: panoptic buffers; Q in (64, 128]");
        PO(psalm_semantic_from_masks_x3(mp, probsT, io->sem_seg, mscore, msws, Q, C1 - 1, HW, Kpad, stream));          // LP:402-406, 443-444
        PO(psalm_memset_zero(io->counts, (long)(2 + 3 * Q) * 4, stream));
        PO(zero_topk());
        PO(psalm_topk_select(probs, Q, C1 - 1, C1, Q, io->is_thing, mscore, io->scores, cl, qq, io->counts, 0, stream));   // LP:407-447
        PO(psalm_binarize_gather(mp, qq, io->counts, io->inst_masks, Q, HW, stream));
        PO(psalm_panoptic(mp, score, label, io->is_thing, (int*)(ws + lo.argq), (int*)(ws + lo.pcnt), (int*)(ws + lo.fid), io->pan, io->counts + 2,
                          io->counts + 1, Q, HW, C1 - 1, d->obj_thr, d->overlap_thr, stream));
        PO(psalm_cast_i32_i64(cl, io->classes, Q, stream));
        PO(psalm_cast_i32_i64(qq, io->query, Q, stream));
        PO(psalm_memset_zero(io->boxes, (long)Q * 4 * 4, stream));
    } else if (task == PSALM_POST_REFERRING) {
        PSALM_CHECK_ARG(io->seg_logits && io->query, "psalm_postprocess: seg_logits / query");
        PO(psalm_mask_scores(mp, mscore, msws, Q, HW, stream));
        PO(zero_topk());
        PO(psalm_memset_zero(io->counts, 4, stream));
        PO(psalm_topk_select(io->seg_logits, Q, 1, 1, Q, nullptr, mscore, io->scores, cl, qq, io->counts, 1, stream));     // LP:308-324
        PO(psalm_binarize_gather(mp, qq, io->counts, io->inst_masks, Q, HW, stream));
        PO(psalm_cast_i32_i64(qq, io->query, Q, stream));
        PO(psalm_memset_zero(io->boxes, (long)Q * 4 * 4, stream));
    } else {                                                                            // region
        PSALM_CHECK_ARG(io->region_logits, "psalm_postprocess: region_logits");
        PO(psalm_mask_scores(mp, mscore, msws, Q, HW, stream));
        PO(psalm_region_scores(io->region_logits, mscore, io->scores, d->n_region, Q, stream));                            // LP:387-400
        PO(psalm_binarize_gather(mp, nullptr, nullptr, io->inst_masks, Q, HW, stream));
        PO(psalm_memset_zero(io->boxes, (long)Q * 4 * 4, stream));
    }
    (void)mh; (void)mw;
#undef PO
    return 0;
}
#define PSALM_POST_NAMED(name_, code_)                                                                                                             \
    extern "C" int psalm_postprocess_##name_(const psalm_post_desc* d, const psalm_post_io* io, int* mask_pred_is, void* workspace,               \
                                             long workspace_bytes, void* stream) {                                                                 \
        PSALM_CHECK_ARG(d && d->task == code_, "psalm_postprocess_" #name_ ": descriptor of another task");                                         \
        return psalm_postprocess(d, io, mask_pred_is, workspace, workspace_bytes, stream);                                                         \
    }
PSALM_POST_NAMED(semantic, PSALM_POST_SEMANTIC)
PSALM_POST_NAMED(instance, PSALM_POST_INSTANCE)
PSALM_POST_NAMED(panoptic, PSALM_POST_PANOPTIC)
PSALM_POST_NAMED(referring, PSALM_POST_REFERRING)
PSALM_POST_NAMED(region, PSALM_POST_REGION)
