// Attention kernels of the PSALM inference path (first generation: fp32 VALU math, LDS-staged K/V tiles,
// wave-parallel online softmax).  All softmax statistics and accumulators are fp32; inputs may be f32 or bf16.
//
//   psalm_window_attention  Swin (shifted-)window attention, 12x12 windows, head_dim 32
//                           (swin_trans.py:117-149 + the shift mask of :369-387 computed in-kernel)
//   psalm_causal_attention  Phi prefill attention with partial RoPE fused into the Q/K loads
//                           (modeling_phi.py:189-245, :137-160, :92-122)
//   psalm_mha_attention     nn.MultiheadAttention core for the predictor: 100 queries x (HW | 100) keys, optional
//                           boolean mask with the "all-masked row => unmasked" rule
//                           (mask2former_transformer_decoder.py:645-666, :647)
//   psalm_attn_mask         bilinear resize of mask logits + (sigmoid < 0.5) -> u8 mask + all-masked row flags
//                           (mask2former_transformer_decoder.py:754-759)
#include "common.h"
#include <cstdlib>

// ============================================================================================ Swin window attention
// qkv (B*nW*N, 3C) rows ordered like window_partition (swin_trans.py:37-49); head h uses columns
// [h*32, h*32+32) of each of the q | k | v thirds.  One block = one (window, head): K, V and the head's
// relative-position-bias column are staged in LDS; thread t < N owns query row t (q in registers) and streams
// the N keys with an online softmax, 4 keys per rescale.
template <typename T, int HD>
__global__ void __launch_bounds__(192) window_attention_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                               T* __restrict__ out, int nWh, int nWw, int C, int heads, int ws,
                                                               int shift) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int N = ws * ws;
    float* Ks = smem;                       // [N][HD+1]
    float* Vs = Ks + N * (HD + 1);          // [N][HD+1]
    float* Bs = Vs + N * (HD + 1);          // [(2ws-1)^2]
    const int win = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int nbias = (2 * ws - 1) * (2 * ws - 1);
    const long row0 = (long)win * N;
    for (int e = tid; e < N * HD; e += blockDim.x) {
        const int r = e / HD, c = e % HD;
        const T* p = qkv + (row0 + r) * 3 * C + h * HD + c;
        Ks[r * (HD + 1) + c] = ldf(p + C);
        Vs[r * (HD + 1) + c] = ldf(p + 2 * C);
    }
    for (int e = tid; e < nbias; e += blockDim.x) Bs[e] = bias_table[(long)e * heads + h];
    __syncthreads();
    if (tid >= N) return;
    const float scale = rsqrtf((float)HD);
    float q[HD], o[HD];
    {
        const T* p = qkv + (row0 + tid) * 3 * C + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) { q[c] = ldf(p + c) * scale; o[c] = 0.f; }
    }
    const int yi = tid / ws, xi = tid % ws;
    // shift-mask region label of a token (swin_trans.py:371-387): slices (0,-ws), (-ws,-shift), (-shift,None)
    const int wwin = win % (nWh * nWw);
    const int wh = wwin / nWw, ww = wwin % nWw;
    const int Hp = nWh * ws, Wp = nWw * ws;
    auto label = [&](int yy, int xx) -> int {
        const int gy = wh * ws + yy, gx = ww * ws + xx;
        const int ly = gy < Hp - ws ? 0 : (gy < Hp - shift ? 1 : 2);
        const int lx = gx < Wp - ws ? 0 : (gx < Wp - shift ? 1 : 2);
        return ly * 3 + lx;
    };
    const int my_label = shift > 0 ? label(yi, xi) : 0;
    float m = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < N; j0 += 4) {
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            if (j < N) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc += q[c] * Ks[j * (HD + 1) + c];
                const int yj = j / ws, xj = j % ws;
                acc += Bs[(yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1)];
                if (shift > 0 && label(yj, xj) != my_label) acc += -100.0f;
                s[u] = acc;
            } else s[u] = -3.0e38f;
        }
        const float mc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        const float mn = fmaxf(m, mc);
        const float alpha = __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            if (j < N) {
                const float p = __expf(s[u] - mn);
                l += p;
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] += p * Vs[j * (HD + 1) + c];
            }
        }
        m = mn;
    }
    const float inv = 1.f / l;
    T* op = out + (row0 + tid) * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) stf(op + c, o[c] * inv);
}

// ---- fp32 matrix-core form (v_mfma_f32_16x16x4_f32: exact fp32 products), used for fp32 buffers with 12x12 windows.
// One wavefront per (window, head): N = 144 tokens = 9 tiles of 16.  Swapped products, as in the causal kernel below:
//   S^T (16 keys x 16 q) = K_tile . Q^T    8 MFMAs (head dim 32 = 8 steps of 4; lane (n, kk) contracts d = 8 kk + s in step s, so its
//                                          K and Q fragments are 8 consecutive floats)
//   the WHOLE score column block of a 16-query tile (9 key tiles x 4 registers) stays in registers: exact two-pass softmax, the
//   statistics of a query live in the 4 lanes {q, q+16, q+32, q+48} (two xor-shuffles);
//   O^T (16 d x 16 q)   += V_tile^T . P^T   4 MFMAs per key tile and d-tile, P straight from the score registers: step r contracts the
//                                          keys 4 kk + r that lane (q, kk) holds in register r; V is read un-transposed from LDS.
// V (+ the bias column) is staged in LDS (rows padded to 36 floats: the 4-byte V reads of a step hit 64 distinct banks).  The K fragments --
// 32 bytes per lane and key tile, the same for every query tile of a wavefront -- are loaded ONCE into 72 registers (one wavefront per pair:
// 330 registers, one wavefront per SIMD) or, in the KLDS flavour, read from a second LDS copy (three wavefronts per pair, 119 registers, 44 KB of
// LDS per block: three blocks per CU).  r04 fetched them from L1 / L2 once per query tile.
// SO = true: the output leaves as the split-f16 A operand of the projection GEMM: `out` is then the f16 operand buffer (rows of 2 * so_kp:
// hi at column h*32 + d, lo so_kp further).  One power-of-two scale per WINDOW from a magnitude bound that needs no pass over the output:
// an output row is a convex combination of the window's v rows, |v_jd| <= a_inv[j] * par[0] + par[1] (a_inv: the row scales of the qkv
// GEMM's split-f16 A operand, par = {2^14 max_n sum_k |w_nk| over the v rows of the qkv weight, max |b_v|}), so max_j of that bounds all of
// them; every head's wavefront derives the same scale, head 0's writes 1/scale to so_inv.
// NWV wavefronts per (window, head) share the staged V / bias (/ K) and take every NWV-th query tile: a Swin-B stage-3 launch has only 36
// windows x 16 heads = 576 (window, head) pairs for 1024 SIMDs, each a serial chain of matrix instructions and softmax arithmetic when one
// wavefront owns all 9 tiles.
#ifndef PSALM_WINATTN_KLDS_MAX
#define PSALM_WINATTN_KLDS_MAX 640          // (window, head) pairs up to which the three-wavefront, K-through-LDS flavour runs (A/B builds: -D...)
#endif
template <int HD, int WS, bool SO = false, int NWV = 3, bool KLDS = false>
__global__ void __launch_bounds__(64 * NWV) PSALM_WAVES_PER_EU(KLDS ? 3 : 1) window_attention_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_table,
                                                                       float* __restrict__ out, int nWh, int nWw, int C, int heads, int shift,
                                                                       const float* __restrict__ a_inv = nullptr,
                                                                       const float* __restrict__ so_par = nullptr,
                                                                       float* __restrict__ so_inv = nullptr, int so_kp = 0) {
    static_assert(HD == 32 && WS == 12, "Swin-B window geometry");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int N = WS * WS, NT = N / 16, LS = HD + 4, NB = (2 * WS - 1) * (2 * WS - 1);
    HIP_DYNAMIC_SHARED(float, smem)
    float* Vs = smem;                       // [N][LS]
    float* Bs = Vs + N * LS;                // [NB]
    float* Ks = smem + (N * LS + NB + (N * 2 + 3) / 4 + 3) / 4 * 4;      // [N][LS]  (KLDS only; behind the 288-byte key-index table, 16-byte aligned)
    const int win = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, kk = lane >> 4;
    const long row0 = (long)win * N;
    // r05: EVERY global load of the prologue -- V (first batch), K, the bias column, the operand scales, the first query fragment -- is issued
    // before the first LDS store waits for one: the rolled loops of r04 were 18 + 9 dependent global round trips per wavefront when one
    // wavefront stages alone (about half of a stage-3 launch), and the batched form still took them one after another.  144 rows x 8 float4 of
    // V (and of K in the KLDS flavour), 529 bias entries.
    constexpr int VIT = N * (HD / 4) / (64 * NWV), BIT = (NB + 64 * NWV - 1) / (64 * NWV), VB = VIT % 6 == 0 ? 6 : VIT;   // loads per batch
    static_assert(N * (HD / 4) % (64 * NWV) == 0 && (VIT == VB || VIT == 3 * VB), "V staging: one or three whole batches");
    static_assert(!KLDS || VIT == VB, "KLDS: one batch per operand");
    // K fragments of ALL nine key tiles: 72 registers that every query tile of this wavefront reuses (r04: re-fetched from L1 / L2 per query
    // tile, the first products of a tile behind that round trip).  KLDS: three wavefronts per pair AND three such blocks per CU -- a stage-3
    // launch's 576 pairs all resident -- need <= 168 registers: the fragments then come from an LDS copy of K, 16 bytes per read.
    f32x4 kf[KLDS ? 1 : NT][2];
    f32x4 kt[KLDS ? VB : 1];
    if constexpr (!KLDS) {
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) {
            const float* kp = qkv + (row0 + 16 * tk + n16) * 3 * C + C + h * HD + 8 * kk;
            kf[tk][0] = reinterpret_cast<const f32x4*>(kp)[0];
            kf[tk][1] = reinterpret_cast<const f32x4*>(kp)[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < VB; ++i) {
            const int e = tid + i * 64 * NWV, r = e >> 3, c4 = (e & 7) * 4;
            kt[i] = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * 3 * C + h * HD + c4 + C);
        }
    }
    auto load_q = [&](int tq_, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        const float* p = qkv + (row0 + 16 * tq_ + n16) * 3 * C + h * HD + 8 * kk;
        a = reinterpret_cast<const f32x4*>(p)[0];
        b = reinterpret_cast<const f32x4*>(p)[1];
    };
    auto load_v = [&](int b0, f32x4 (&vt)[VB]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VB; ++i) {
            const int e = tid + (b0 + i) * 64 * NWV, r = e >> 3, c4 = (e & 7) * 4;
            vt[i] = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * 3 * C + h * HD + c4 + 2 * C);
        }
    };
    auto store_v = [&](int b0, const f32x4 (&vt)[VB], float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VB; ++i) {
            const int e = tid + (b0 + i) * 64 * NWV, r = e >> 3, c4 = (e & 7) * 4;
            *reinterpret_cast<f32x4*>(&dst[r * LS + c4]) = vt[i];
        }
    };
    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f};           // the NEXT tile's query fragment, fetched one tile ahead
    float gmax = 0.f;
    {
        f32x4 vt[VB];
        float bt[BIT];
        load_v(0, vt);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int e = tid + i * 64 * NWV;
            bt[i] = e < NB ? bias_table[(long)e * heads + h] : 0.f;
        }
        if constexpr (SO) {
#pragma unroll
            for (int i = 0; i < (N + 63) / 64; ++i) gmax = fmaxf(gmax, lane + 64 * i < N ? a_inv[row0 + lane + 64 * i] : 0.f);
        }
        if (wave < NT) load_q(wave, qa, qb);
        PSALM_SCHED_FENCE();
        store_v(0, vt, Vs);
        if constexpr (KLDS) store_v(0, kt, Ks);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int e = tid + i * 64 * NWV;
            if (e < NB) Bs[e] = bt[i];
        }
        PSALM_SCHED_FENCE();
        if constexpr (VIT > VB) {
            load_v(VB, vt);
            PSALM_SCHED_FENCE();
            store_v(VB, vt, Vs);
            PSALM_SCHED_FENCE();
            load_v(2 * VB, vt);
            PSALM_SCHED_FENCE();
            store_v(2 * VB, vt, Vs);
            PSALM_SCHED_FENCE();
        }
    }
    // r05: the relative-position index of a (query, key) pair is (yi - yj + WS-1) (2 WS-1) + (xi - xj + WS-1) = c(query) - c'(key) with
    // c'(j) = (j / WS) (2 WS-1) + j % WS; the key's part, as a BYTE offset into Bs, sits in a 288-byte LDS table instead of being re-derived per
    // score element (two divisions by 12 and the index arithmetic: ~10 of the ~16 VALU instructions an element cost -- ISA of r04: 590 per query tile)
    unsigned short* Ki = reinterpret_cast<unsigned short*>(Bs + NB);
    for (int e = tid; e < N; e += 64 * NWV) Ki[e] = (unsigned short)(((e / WS) * (2 * WS - 1) + e % WS) * 4);
    float so_sc = 1.f;
    if constexpr (SO) {
        gmax = wave_max(gmax);
        float bound = fminf(fmaxf(gmax * so_par[0] + so_par[1], 7.888609e-31f), 1.2676506e30f);        // [2^-100, 2^100]
        const unsigned eb = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;                        // bound * scale in [2^12, 2^13)
        so_sc = __builtin_bit_cast(float, (266u - eb) << 23);
        if (h == 0 && wave == 0) {
            const float inv = __builtin_bit_cast(float, (eb - 12u) << 23);
            for (int r = lane; r < N; r += 64) so_inv[row0 + r] = inv;
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    // shift-mask region label of a token (swin_trans.py:371-387): slices (0,-ws), (-ws,-shift), (-shift,None)
    const int wwin = win % (nWh * nWw);
    const int wh = wwin / nWw, ww = wwin % nWw;
    const int Hp = nWh * WS, Wp = nWw * WS;
    auto label = [&](int t) -> int {
        const int gy = wh * WS + t / WS, gx = ww * WS + t % WS;
        const int ly = gy < Hp - WS ? 0 : (gy < Hp - shift ? 1 : 2);
        const int lx = gx < Wp - WS ? 0 : (gx < Wp - shift ? 1 : 2);
        return ly * 3 + lx;
    };
    // labels (0..8) of the 36 keys this lane holds (16 tk + 4 kk + r), one nibble each: 5 registers instead of 36 -- the kernel's register
    // count decides how many wavefronts a SIMD holds (r05: 321 registers = ONE; the matrix instructions of a query tile then wait out its
    // own softmax arithmetic and K fetches with nothing else to issue)
    unsigned klab[(NT * 4 + 7) / 8] = {};
    // (block-uniform) only the last row / column of windows of a shifted layer holds tokens of more than one region: every other window takes
    // the copy of the tile loop without the 36 compare-and-select pairs per query tile
    const bool masked = shift > 0 && (wh == nWh - 1 || ww == nWw - 1);
    if (masked) {
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) klab[(4 * tk + r) >> 3] |= (unsigned)label(16 * tk + 4 * kk + r) << (4 * ((4 * tk + r) & 7));
    }
    auto tiles = [&](auto MASKED) __attribute__((always_inline)) {
#pragma unroll 1
    for (int tq = wave; tq < NT; tq += NWV) {
        const int qi = 16 * tq + n16;                                    // this lane's query (column of S^T / O^T)
        float qf[8];
        qf[0] = qa.x * scale; qf[1] = qa.y * scale; qf[2] = qa.z * scale; qf[3] = qa.w * scale;
        qf[4] = qb.x * scale; qf[5] = qb.y * scale; qf[6] = qb.z * scale; qf[7] = qb.w * scale;
        if (tq + NWV < NT) load_q(tq + NWV, qa, qb);
        const int yi = qi / WS, xi = qi % WS;
        const int qc4 = ((yi + WS - 1) * (2 * WS - 1) + (xi + WS - 1)) * 4;            // byte offset of this query's part of the bias index
        const unsigned short* Kik = Ki + 4 * kk;                                        // this lane's keys: 16 tk + 4 kk + r
        const int qlab = decltype(MASKED)::value ? label(qi) : 0;
        f32x4 sc[NT];
        float mx = -3.0e38f;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            f32x4 k0, k1;
            if constexpr (KLDS) {
                const float* kp = &Ks[(16 * tk + n16) * LS + 8 * kk];
                k0 = reinterpret_cast<const f32x4*>(kp)[0];
                k1 = reinterpret_cast<const f32x4*>(kp)[1];
            } else {
                k0 = kf[tk][0];
                k1 = kf[tk][1];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[3], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[5], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[6], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[7], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[r] + *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bs) + (qc4 - (int)Kik[16 * tk + r]));
                if (decltype(MASKED)::value && (int)((klab[(4 * tk + r) >> 3] >> (4 * ((4 * tk + r) & 7))) & 15u) != qlab) v += -100.0f;
                acc[r] = v;
                mx = fmaxf(mx, v);
            }
            sc[tk] = acc;
            if constexpr (KLDS) PSALM_SCHED_FENCE();              // register budget: one key tile's reads in flight, not nine
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float l = 0.f;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[tk][r] - mx);
                sc[tk][r] = p;
                l += p;
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};       // O^T d-tiles: rows d = 16 t + 4 kk + r, column q
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = &Vs[(16 * tk + 4 * kk + r) * LS + n16];
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], sc[tk][r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16], sc[tk][r], o1, 0, 0, 0);
            }
            if constexpr (KLDS) PSALM_SCHED_FENCE();
        }
        const float inv = 1.f / l;
        if constexpr (SO) {
            unsigned short* op = reinterpret_cast<unsigned short*>(out) + (row0 + qi) * 2L * so_kp + h * HD + 4 * kk;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                unsigned hw[2], lw[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float a0 = (t ? o1[2 * k] : o0[2 * k]) * inv * so_sc, a1 = (t ? o1[2 * k + 1] : o0[2 * k + 1]) * inv * so_sc;
                    const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                    const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                    hw[k] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    lw[k] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
                *reinterpret_cast<psalm_u32x2*>(op + 16 * t) = psalm_u32x2{hw[0], hw[1]};
                *reinterpret_cast<psalm_u32x2*>(op + 16 * t + so_kp) = psalm_u32x2{lw[0], lw[1]};
            }
        } else {
            float* op = out + (row0 + qi) * C + h * HD + 4 * kk;
            *reinterpret_cast<psalm_f32x4*>(op) = psalm_f32x4{o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv};
            *reinterpret_cast<psalm_f32x4*>(op + 16) = psalm_f32x4{o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv};
        }
    }
    };
    if (masked) tiles(std::true_type{});
    else tiles(std::false_type{});
}

// psalm_window_attention (fp32 buffers, 12 x 12 windows) whose output leaves as the split-f16 A operand of the projection GEMM
// (swin_trans.py:144-153 -> self.proj): split_out rows of 2*split_kp f16 (hi at column head*32 + d, lo split_kp further), split_inv (rows);
// a_inv: the row scales of the qkv GEMM's split-f16 A operand (rows in window order), bound_par: 2 device floats
// {2^14 max_n sum_k |w_nk| over the v rows of the qkv weight, max |b_v|}.  Columns [C, split_kp) are the caller's to zero.
extern "C" int psalm_window_attention_split(const float* qkv, const float* bias_table, const float* a_inv, const float* bound_par,
                                            void* split_out, int split_kp, float* split_inv, int B, int nWh, int nWw, int C, int heads,
                                            int ws, int shift, void* stream) {
    PSALM_CHECK_ARG(C == heads * 32 && ws == 12, "psalm_window_attention_split: head_dim 32, 12 x 12 windows");
    PSALM_CHECK_ARG(a_inv && bound_par && split_out && split_inv && split_kp >= C && split_kp % 8 == 0 && (uintptr_t)qkv % 16 == 0 &&
                        (uintptr_t)split_out % 16 == 0, "psalm_window_attention_split: operand scales, bound parameters, aligned buffers");
    const int nwin = B * nWh * nWw;
    if (nwin == 0) return 0;
    const size_t lds = (size_t)(144 * 36 + 23 * 23) * sizeof(float) + 288;                 // V rows + bias column + the key-index table
    if ((long)nwin * heads <= PSALM_WINATTN_KLDS_MAX)                     // flavour: as psalm_window_attention
        hipLaunchKernelGGL((window_attention_f32_mfma_kernel<32, 12, true, 3, true>), dim3(nwin, heads), dim3(192), lds + 16 + 144 * 36 * sizeof(float),
                           (hipStream_t)stream, qkv, bias_table, (float*)split_out, nWh, nWw, C, heads, shift, a_inv, bound_par, split_inv, split_kp);
    else
        hipLaunchKernelGGL((window_attention_f32_mfma_kernel<32, 12, true, 1>), dim3(nwin, heads), dim3(64), lds, (hipStream_t)stream, qkv,
                           bias_table, (float*)split_out, nWh, nWw, C, heads, shift, a_inv, bound_par, split_inv, split_kp);
    PSALM_LAUNCH_END("psalm_window_attention_split");
}

extern "C" int psalm_window_attention(const void* qkv, const float* bias_table, void* out, int dtype, int B, int nWh, int nWw,
                                      int C, int heads, int ws, int shift, void* stream) {
    PSALM_CHECK_ARG(C == heads * 32, "psalm_window_attention: head_dim must be 32");
    PSALM_CHECK_ARG(ws * ws <= 192, "psalm_window_attention: window too large (ws*ws <= 192)");
    const int nwin = B * nWh * nWw;
    if (nwin == 0) return 0;
    const int N = ws * ws;
    if (dtype == PSALM_F32 && ws == 12 && C % 4 == 0 && (uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 16 == 0) {     // fp32 matrix-core kernel
        const size_t lds = (size_t)(N * 36 + (2 * ws - 1) * (2 * ws - 1)) * sizeof(float) + (size_t)((N * 2 + 3) / 4 * 4);
        // Two flavours (r05, kernel-trace durations of a 1024^2 image's four stages, profiles/r05q_winattn_trace_blocks.txt).  Up to 640 (window,
        // head) pairs -- stage 3: 576, stage 4: 288 -- three wavefronts share a pair, K comes from an LDS copy and the kernel keeps to 119
        // registers, so that three blocks fit a CU and the whole launch is resident: 47.2 -> 34.0 us and 31.0 -> 23.5 us.  Larger grids: one
        // wavefront per pair with K resident in registers (occupancy 1): stage 1, 1936 pairs: 98 -> 78 us; stage 2, 968: 50 -> 40 (the
        // three-wavefront flavour there: 47 us).  Both produce the r04 kernel's words.
        if ((long)nwin * heads <= PSALM_WINATTN_KLDS_MAX)
            hipLaunchKernelGGL((window_attention_f32_mfma_kernel<32, 12, false, 3, true>), dim3(nwin, heads), dim3(192), lds + 16 + 144 * 36 * sizeof(float),
                               (hipStream_t)stream, (const float*)qkv, bias_table, (float*)out, nWh, nWw, C, heads, shift);
        else
            hipLaunchKernelGGL((window_attention_f32_mfma_kernel<32, 12, false, 1>), dim3(nwin, heads), dim3(64), lds, (hipStream_t)stream,
                               (const float*)qkv, bias_table, (float*)out, nWh, nWw, C, heads, shift);
        PSALM_LAUNCH_END("psalm_window_attention");
    }
    const size_t shmem = (size_t)(2 * N * 33 + (2 * ws - 1) * (2 * ws - 1)) * sizeof(float);
    PSALM_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((window_attention_kernel<T, 32>), dim3(nwin, heads), dim3(192), shmem, (hipStream_t)stream,
                           (const T*)qkv, bias_table, (T*)out, nWh, nWw, C, heads, ws, shift);
    });
    PSALM_LAUNCH_END("psalm_window_attention");
}

// ============================================================================================ Phi causal attention
// q/k/v are column blocks of one row-strided buffer (ld elements per token): q at col q_off + h*64, etc.
// Partial RoPE on the first `rot` dims of each head (modeling_phi.py:218-231): x*cos + rotate_half(x)*sin with
// rotate_half(x) = cat(-x[rot/2:], x[:rot/2]); cos/sin tables (Lmax, rot) fp32 are computed on the host exactly as the
// reference does (emb = cat(freqs, freqs)).  allowed(i,j) = j <= i && key_mask[b,j]  (causal + padding).
// Block = 128 threads = 128 consecutive queries of one (batch, head); K/V tiles of 64 keys are staged in LDS
// (RoPE applied while staging); each thread owns one query row (q, o in registers), 4 keys per softmax rescale.
template <typename T, int HD, int ROT, int QT>
__global__ void __launch_bounds__(QT) causal_attention_kernel(const T* __restrict__ base, long ld, int q_off, int k_off,
                                                               int v_off, T* out, long ldo, int o_off,
                                                               const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                               const unsigned char* __restrict__ key_mask, int L, int heads,
                                                               float scale) {
    constexpr int KT = 64, rot = ROT;
    __shared__ float Ks[KT][HD + 1];
    __shared__ float Vs[KT][HD + 1];
    __shared__ unsigned char Ms[KT];
    const int tid = threadIdx.x;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int qi = qt * QT + tid;
    const bool active = qi < L;
    const long tok0 = (long)b * L;
    constexpr int half = ROT / 2;
    float q[HD], o[HD];
    if (active) {
        const T* p = base + (tok0 + qi) * ld + q_off + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) q[c] = ldf(p + c);
        // RoPE on q
#pragma unroll
        for (int c = 0; c < half; ++c) {
            const float x1 = q[c], x2 = q[c + half];
            const float c1 = cosT[(long)qi * rot + c], s1 = sinT[(long)qi * rot + c];
            const float c2 = cosT[(long)qi * rot + c + half], s2 = sinT[(long)qi * rot + c + half];
            q[c] = x1 * c1 - x2 * s1;
            q[c + half] = x2 * c2 + x1 * s2;
        }
#pragma unroll
        for (int c = 0; c < HD; ++c) { q[c] *= scale; o[c] = 0.f; }
    } else {
#pragma unroll
        for (int c = 0; c < HD; ++c) { q[c] = 0.f; o[c] = 0.f; }
    }
    float m = -3.0e38f, l = 0.f;
    const int last_q = min(L - 1, qt * QT + QT - 1);
    const int ntiles = last_q / KT + 1;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        for (int e = tid; e < KT * HD; e += QT) {
            const int r = e / HD, c = e % HD;
            const int kj = kt * KT + r;
            float kv = 0.f, vv = 0.f;
            if (kj < L) {
                const T* p = base + (tok0 + kj) * ld;
                kv = ldf(p + k_off + h * HD + c);
                vv = ldf(p + v_off + h * HD + c);
                if (c < rot) {
                    const float other = ldf(p + k_off + h * HD + (c < half ? c + half : c - half));
                    const float cs = cosT[(long)kj * rot + c], sn = sinT[(long)kj * rot + c];
                    kv = kv * cs + (c < half ? -other : other) * sn;
                }
            }
            Ks[r][c] = kv;
            Vs[r][c] = vv;
        }
        for (int e = tid; e < KT; e += QT) { const int kj = kt * KT + e; Ms[e] = (kj < L) ? key_mask[(long)b * L + kj] : 0; }
        __syncthreads();
        if (active && kt * KT <= qi) {
            for (int j0 = 0; j0 < KT; j0 += 4) {
                float s[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u, kj = kt * KT + j;
                    ok[u] = (kj <= qi) && Ms[j];
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < HD; ++c) acc += q[c] * Ks[j][c];
                    s[u] = ok[u] ? acc : -3.0e38f;
                }
                const float mc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                if (mc > -1.0e38f) {
                    const float mn = fmaxf(m, mc);
                    const float alpha = __expf(m - mn);
                    l *= alpha;
#pragma unroll
                    for (int c = 0; c < HD; ++c) o[c] *= alpha;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ok[u]) {
                            const float p = __expf(s[u] - mn);
                            l += p;
#pragma unroll
                            for (int c = 0; c < HD; ++c) o[c] += p * Vs[j0 + u][c];
                        }
                    }
                    m = mn;
                }
            }
        }
    }
    if (active) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        T* op = out + (tok0 + qi) * ldo + o_off + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) stf(op + c, o[c] * inv);
    }
}

// ---- fp32 matrix-core form (exact-fp32 products: v_mfma_f32_32x32x2_f32), used for fp32 buffers (precision "fp32" / "f16x3").
// The per-thread VALU kernel above spends an LDS read per FMA (27 ms of an 88 ms fp32-mode image, r02 breakdown); here a wavefront
// owns 32 queries of one (batch, head) and walks 32-key tiles:
//   S^T (32 keys x 32 q)  = K_tile . Q^T      32 MFMAs  (A = K rows from LDS, B = the wave's Q, resident in 32 VGPRs)
//   O^T (64 d  x 32 q)   += V_tile^T . P^T    32 MFMAs  (A = V^T from LDS, B = P straight from the S accumulators)
// "Swapped" products: a lane owns ONE query column of S^T / O^T, so the online-softmax statistics (max, sum, rescale) are lane-local
// plus one exchange with the partner lane 32 away (the two k-halves), and P never leaves registers: the accumulator layout of S^T
// (lane (q, hi) holds keys (r&3) + 8(r>>2) + 4hi, r = 0..15) is used as the key order of the second product -- step r contracts keys
// {.. + 0, .. + 4}; V^T is read from LDS in the same order.  Likewise the first product contracts head dims in the order
// d = 32 hi + s (step s = 0..31), so both K fragments and V^T fragments are contiguous 16-byte LDS reads.
// Block = NW wavefronts = 32 NW consecutive queries sharing the K / V tiles (RoPE applied to K while staging; V transposed while staging).
// K / V tiles are double-buffered: the next tile's global loads are issued before the current tile's 64 MFMAs and land in LDS after them
// (one barrier per tile) -- with single-buffered staging the exposed load latency was 1.5x the MFMA time per tile (r02e: 134 us per layer).
// Causal work grows with the query index: blocks are issued last query tile first.  (Tried r02: 2-wave blocks, several per CU -- slower,
// 189 us: twice the staging traffic and no better balance, since all blocks are resident from the start.)
template <int HD, int ROT, int NW>
__global__ void __launch_bounds__(64 * NW) causal_attention_f32_mfma_kernel(const float* __restrict__ base, long ld, int q_off, int k_off,
                                                                        int v_off, float* out, long ldo, int o_off,
                                                                        const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                        const unsigned char* __restrict__ key_mask, int L, int heads,
                                                                        float scale) {
    static_assert(HD == 64 && ROT == 32 && NW == 4, "Phi-1.5 geometry; 256 threads stage one 32-key tile in one pass");
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int KT = 32, KS = HD + 4, VS = KT + 4, half = ROT / 2;   // padded LDS row strides (floats): conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float Ks[2][KT * KS];        // [key][d]   double-buffered: tile kt+1 is fetched (global ->
    __shared__ __attribute__((aligned(16))) float Vt[2][HD * VS];        // [d][key]   registers) before tile kt's MFMAs and written after
    __shared__ unsigned char Ms[2][KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int qt = gridDim.x - 1 - blockIdx.x, h = blockIdx.y, b = blockIdx.z;      // heaviest (last) query tiles first
    const long tok0 = (long)b * L;
    const int q0 = qt * (32 * NW) + wave * 32;                           // this wave's first query
    const int qi = q0 + n32;                                             // this lane's query column
    // ---- Q: lane (q, hi) holds (RoPE'd, scaled) Q[q][32 hi + s], s = 0..31
    float qv[32];
    {
        const float* p = base + (tok0 + min(qi, L - 1)) * ld + q_off + h * HD + 32 * hi;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const psalm_f32x4 t = *reinterpret_cast<const psalm_f32x4*>(p + c);
            qv[c] = t.x; qv[c + 1] = t.y; qv[c + 2] = t.z; qv[c + 3] = t.w;
        }
        if (hi == 0) {                                                   // rotary dims 0..31 live entirely in the hi = 0 lanes
            const float* cs = cosT + (long)min(qi, L - 1) * ROT;
            const float* sn = sinT + (long)min(qi, L - 1) * ROT;
#pragma unroll
            for (int c = 0; c < half; ++c) {
                const float x1 = qv[c], x2 = qv[c + half];
                qv[c] = x1 * cs[c] - x2 * sn[c];
                qv[c + half] = x2 * cs[c + half] + x1 * sn[c + half];
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) qv[c] *= scale;
    }
    f32x16 o0, o1;                                                       // O^T d-tiles 0 / 1: rows d = 32 t + (r&3) + 8(r>>2) + 4hi, column q
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    const int last_q = min(L - 1, qt * (32 * NW) + 32 * NW - 1);
    const int ntiles = last_q / KT + 1;                                  // key tiles the block's last query can see
    // staging: thread -> key tid/8 of the tile, head dims 8 (tid%8) .. +7
    const int sr = tid >> 3, sc0 = (tid & 7) * 8;
    float kv[8], vv[8], ko[8], cs8[8], sn8[8];
    unsigned char mk = 0;
    auto fetch = [&](int kt) {                                           // global -> registers (all loads unconditional, row clamped)
        const int kj = min(kt * KT + sr, L - 1);
        const float* p = base + (tok0 + kj) * ld;
        ld8(p + k_off + h * HD + sc0, kv);
        ld8(p + v_off + h * HD + sc0, vv);
        const int cr = sc0 < ROT ? sc0 : 0;                               // (non-rotary threads fetch a valid dummy: results unused)
        ld8(p + k_off + h * HD + (cr < half ? cr + half : cr - half), ko);
        ld8(cosT + (long)kj * ROT + cr, cs8);
        ld8(sinT + (long)kj * ROT + cr, sn8);
        if (tid < KT) { const int kk = kt * KT + tid; mk = key_mask[(long)b * L + min(kk, L - 1)] && kk < L; }
    };
    auto stage = [&](int buf) {                                          // registers -> LDS (RoPE on K, V transposed)
        if (sc0 < ROT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) kv[i] = kv[i] * cs8[i] + (sc0 < half ? -ko[i] : ko[i]) * sn8[i];
        }
        *reinterpret_cast<psalm_f32x4*>(&Ks[buf][sr * KS + sc0]) = psalm_f32x4{kv[0], kv[1], kv[2], kv[3]};
        *reinterpret_cast<psalm_f32x4*>(&Ks[buf][sr * KS + sc0 + 4]) = psalm_f32x4{kv[4], kv[5], kv[6], kv[7]};
#pragma unroll
        for (int i = 0; i < 8; ++i) Vt[buf][(sc0 + i) * VS + sr] = vv[i];
        if (tid < KT) Ms[buf][tid] = mk;
    };
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) fetch(kt + 1);                              // in flight during this tile's 64 MFMAs
        if (kt * KT <= min(q0 + 31, L - 1)) {                            // (else: tile entirely above this wave's diagonal; wave-uniform)
            // ---- S^T = K . Q^T
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            {
                const float* kp = &Ks[buf][n32 * KS + 32 * hi];
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    const psalm_f32x4 kf = *reinterpret_cast<const psalm_f32x4*>(kp + c);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qv[c], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qv[c + 1], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qv[c + 2], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qv[c + 3], sacc, 0, 0, 0);
                }
            }
            // ---- mask + online softmax of this lane's query column (16 of its 32 keys here, the other 16 in lane ^ 32)
            float mc = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool ok = (kt * KT + j <= qi) && Ms[buf][j];
                sacc[r] = ok ? sacc[r] : -3.0e38f;
                mc = fmaxf(mc, sacc[r]);
            }
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            const float mn = fmaxf(m, mc);
            const float alpha = __expf(m - mn);                          // m = mn = -3e38 (nothing visible yet): exp(0) = 1, harmless
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = sacc[r] > -1.0e38f ? __expf(sacc[r] - mn) : 0.f;
                sacc[r] = p;
                psum += p;
            }
            psum += __shfl_xor(psum, 32);
            l = l * alpha + psum;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // ---- O^T += V^T . P^T   (step r contracts keys (r&3) + 8(r>>2) + {0, 4}: exactly what lane (q, hi) holds in sacc[r])
            {
                const float* v0 = &Vt[buf][n32 * VS + 4 * hi];
                const float* v1 = &Vt[buf][(32 + n32) * VS + 4 * hi];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const psalm_f32x4 a0 = *reinterpret_cast<const psalm_f32x4*>(v0 + 8 * g);
                    const psalm_f32x4 a1 = *reinterpret_cast<const psalm_f32x4*>(v1 + 8 * g);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, sacc[4 * g], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, sacc[4 * g], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, sacc[4 * g + 1], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, sacc[4 * g + 1], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, sacc[4 * g + 2], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, sacc[4 * g + 2], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, sacc[4 * g + 3], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, sacc[4 * g + 3], o1, 0, 0, 0);
                }
            }
        }
        if (kt + 1 < ntiles) stage(buf ^ 1);       // the other buffer: last read in tile kt-1, which every wave left at the previous barrier
        __syncthreads();
    }
    if (qi < L) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        float* op = out + (tok0 + qi) * ldo + o_off + h * HD + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                    // rows d = 8g + 4hi + {0..3} of each d-tile: one 16-byte store
            *reinterpret_cast<psalm_f32x4*>(op + 8 * g) = psalm_f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            *reinterpret_cast<psalm_f32x4*>(op + 32 + 8 * g) = psalm_f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
        }
    }
}

// ---- key-split form (the one psalm_causal_attention_f32 launches).  The kernel above gives one wavefront ALL key tiles of its 32 queries:
// the last query tile walks 29 tiles alone while 255 other CUs have long finished (r02e/f: 129 us per layer, 4.4 us per tile against
// 1.7 us of MFMA -- one wave per SIMD, nothing to overlap LDS / exp latencies with).  Here a BLOCK owns one 32-query tile and its 4
// wavefronts take every 4th key tile (critical path 8 tiles instead of 29; 928 blocks of graded size, heaviest first, two resident per
// CU), each keeping a private online-softmax state that is merged through LDS at the end.  No tile is shared between wavefronts, so K
// and V^T fragments are read straight from global / L2 (K rows and V rows are contiguous 128 / 256 B per lane group); RoPE, the 1/sqrt(d)
// scale and the padded key mask come from a small pre-pass (phi_rope_prep_f32_kernel) into the caller's workspace.
__global__ void __launch_bounds__(256) phi_rope_prep_f32_kernel(const float* __restrict__ base, long ld, int q_off, int k_off,
                                                                const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                const unsigned char* __restrict__ key_mask, float* __restrict__ Qr,
                                                                float* __restrict__ Kr, unsigned char* __restrict__ Mk,
                                                                unsigned char* __restrict__ Tk, int L, int Lp, int heads, float scale) {
    constexpr int HD = 64, ROT = 32, half = 16;
    const int h = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * 32 + (threadIdx.x >> 3), c0 = (threadIdx.x & 7) * 8;     // token, 8-wide head-dim chunk
    if (t >= Lp) return;
    float* qd = Qr + (((long)b * heads + h) * Lp + t) * HD + c0;
    float* kd = Kr + (((long)b * heads + h) * Lp + t) * HD + c0;
    // Tk[b][tile]: every key of this 32-key tile is a real, un-masked token (r05: such a tile below the diagonal needs no per-element mask work in
    // the attention kernel).  The block's 32 token threads (c0 == 0) each hold one key's flag: any invalid one clears the shared flag (a
    // first form let thread 0 walk the 32 bytes alone: +3 us on this 8 us kernel, profiles/r05_kernel_stats.txt).  Blocks of head 0 only.
    __shared__ int tile_all;
    if (h == 0) {                                        // (block-uniform)
        const bool valid = t < L && key_mask[(long)b * L + min(t, L - 1)] != 0;
        if (threadIdx.x == 0) tile_all = 1;
        __syncthreads();
        if (c0 == 0) {
            if (t < Lp) Mk[(long)b * Lp + t] = valid ? 1 : 0;
            if (!valid) tile_all = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) Tk[(long)b * (Lp / 32) + blockIdx.x] = tile_all ? 1 : 0;
    }
    float q[8], k[8];
    if (t < L) {
        const float* p = base + ((long)b * L + t) * ld;
        ld8(p + q_off + h * HD + c0, q);
        ld8(p + k_off + h * HD + c0, k);
        if (c0 < ROT) {
            float qo[8], ko[8], cs[8], sn[8];
            const int oc = c0 < half ? c0 + half : c0 - half;
            ld8(p + q_off + h * HD + oc, qo);
            ld8(p + k_off + h * HD + oc, ko);
            ld8(cosT + (long)t * ROT + c0, cs);
            ld8(sinT + (long)t * ROT + c0, sn);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                q[i] = q[i] * cs[i] + (c0 < half ? -qo[i] : qo[i]) * sn[i];
                k[i] = k[i] * cs[i] + (c0 < half ? -ko[i] : ko[i]) * sn[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] *= scale;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { q[i] = 0.f; k[i] = 0.f; }
    }
    st8(qd, q);
    st8(kd, k);
}

// SO = true: the output leaves as split-f16 operand columns (hi at `out` + o_off, lo so_kp f16 further; `out` is then the f16 buffer, ldo its
// row stride in f16) under the per-row scales 1 / so_inv[row] ANOTHER kernel chose (psalm_gemm_x3_split's bound covers |v| of every row).
// pair = 1 (default): a block takes the query tiles (nqt-1-p, p) one after the other -- nqt + 1 key tiles of work whatever p.  With one
// tile per block (pair = 0, heaviest first) most of a Phi layer's 928 blocks are resident at once, so the order balances little and a CU
// holding several late tiles sets the time (r02n, L = 899: 72.4 -> 61.5 us; SQ counters of the paired form: matrix pipe 37 % busy,
// waves 52 % issue-stalled behind the partner wave's products, 31 % in s_waitcnt -- profiles/r02n_attn_*).
// PSALM_WAVES_PER_EU(2): 138 instead of 135 + 48 accumulation registers -> 3 resident waves per SIMD (r02n: 80.4 -> 75.0 us unpaired).
template <bool SO>
__global__ void __launch_bounds__(256) PSALM_WAVES_PER_EU(3)
causal_attention_f32_splitk_kernel(const float* __restrict__ Qr, const float* __restrict__ Kr, const unsigned char* __restrict__ Mk,
                                   const unsigned char* __restrict__ Tk, const float* __restrict__ base, long ld, int v_off, float* out, long ldo,
                                   int o_off, int L, int Lp, int heads, const float* __restrict__ so_inv, int so_kp, int pair, int xcd_heads) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int HD = 64, OS = HD + 4;
    __shared__ __attribute__((aligned(16))) float Os[4][32 * OS];         // per-wave O (q-major) for the merge
    __shared__ float Ml[4][2][32];                                        // per-wave (m, l) per query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int nqt = Lp / 32;
    // Block -> (query-tile index bx, head h, batch b).  xcd_heads (r06; host: heads * B is a multiple of 8): the hardware deals linear block ids
    // round-robin over the 8 XCDs, so with (bx, h, b) = blockIdx every head's blocks were spread over all eight and each XCD's 4 MB L2 saw the
    // K / V of all 32 heads (15 MB: 172.9 MB fetched per launch against 30 MB compulsory, profiles/r05_pmc_hbm_traffic.json).  Here XCD x runs
    // the heads x, x + 8, ... one after the other, all query tiles of a head on the same L2.
    int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (xcd_heads) {
        const int lin = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        const int j = lin >> 3, hb = (lin & 7) + 8 * (j / (int)gridDim.x);
        bx = j % (int)gridDim.x;
        h = hb % heads;
        b = hb / heads;
    }
    const long bh = (long)b * heads + h;
    const int npass = pair ? ((int)(nqt - 1 - bx) > bx ? 2 : 1) : 1;
    for (int pass = 0; pass < npass; ++pass) {
    const int qt = pair ? (pass == 0 ? nqt - 1 - bx : bx) : (int)(gridDim.x - 1 - bx);
    const int w0 = pass ? 3 - wave : wave;                                // second tile: key tiles dealt in the opposite wave order
    if (pass) __syncthreads();                                            // the first tile's merge has been read out of Os / Ml
    const int qi = qt * 32 + n32;                                         // this lane's query column (row qi < Lp of Qr)
    float qv[32];
    {
        const float* p = Qr + (bh * Lp + qi) * HD + 32 * hi;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const psalm_f32x4 t = *reinterpret_cast<const psalm_f32x4*>(p + c);
            qv[c] = t.x; qv[c + 1] = t.y; qv[c + 2] = t.z; qv[c + 3] = t.w;
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    // r05: every fragment fetch goes through a buffer descriptor -- a per-lane offset that never changes (one VGPR, set up once) + a
    // wave-uniform offset in an SGPR that walks the key tiles.  The pointer form computed a 64-bit address per V row on the vector ALU:
    // 134 of the loop's 326 VALU instructions, 48 of them quarter-rate integer multiplies (ISA of r04: 32 v_mul_lo_u32 + 16 v_mad_u64_u32 per
    // key tile).  Rows >= L of V (the padding of the last tile; the pointer form clamped them to row L - 1) read as zeros: their
    // probabilities are zeros either way.
    const psalm_rsrc vrs = psalm_make_rsrc(base + (long)b * L * ld + v_off + h * HD, (unsigned)((((long)L - 1) * ld + HD) * 4));
    const unsigned vvo = (unsigned)((4L * hi * ld + n32) * 4);            // lane: key 4 hi of a group of 8, head dim n32 (+32: second d-tile)
    const unsigned vrow = (unsigned)(ld * 4);
    const psalm_rsrc krs = psalm_make_rsrc(Kr + bh * Lp * HD, (unsigned)((long)Lp * HD * 4));
    const unsigned kvo = (unsigned)((n32 * HD + 32 * hi) * 4);
    const int nkt = Lp / 32;
    for (int kt = __builtin_amdgcn_readfirstlane(w0); kt <= qt; kt += 4) {                                // key tiles 0..qt (the diagonal tile is qt)
        // ---- fragments of this tile, all loads issued up front
        psalm_f32x4 kf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const psalm_u32x4 t = psalm_buf_load_b128_s(krs, kvo + 16u * c, (unsigned)(kt * 32 * HD * 4));
            kf[c] = psalm_f32x4{__builtin_bit_cast(float, t.x), __builtin_bit_cast(float, t.y), __builtin_bit_cast(float, t.z), __builtin_bit_cast(float, t.w)};
        }
        float va[16], vb[16];                                             // V^T fragments: step r = 4g + i contracts key 8g + 4hi + i
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned so_ = (unsigned)(kt * 32 + 8 * g + i) * vrow;
                va[4 * g + i] = psalm_buf_load_f32_s(vrs, vvo, so_);
                vb[4 * g + i] = psalm_buf_load_f32_s(vrs, vvo + 128u, so_);
            }
        // a tile below the diagonal whose 32 keys are all real tokens (Tk, written by the pre-pass) needs no mask: the causal and key-mask
        // tests pass for every element, so the selects below would all take their first operand -- same values, ~90 VALU instructions less
        const bool plain = __builtin_amdgcn_readfirstlane((int)(kt < qt && Tk[(long)b * nkt + kt] != 0)) != 0;       // (wave-uniform)
        // ---- S^T = K . Q^T
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].x, qv[4 * c], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].y, qv[4 * c + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].z, qv[4 * c + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].w, qv[4 * c + 3], sacc, 0, 0, 0);
        }
        float mc = -3.0e38f, mn, alpha, psum = 0.f;
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mc = fmaxf(mc, sacc[r]);
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            mn = fmaxf(m, mc);
            alpha = __expf(m - mn);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(sacc[r] - mn);
                sacc[r] = p;
                psum += p;
            }
        } else {
            unsigned mw[4];                                               // key-valid bytes of keys 8g + 4hi .. +3
#pragma unroll
            for (int g = 0; g < 4; ++g) mw[g] = *reinterpret_cast<const unsigned*>(Mk + (long)b * Lp + kt * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool ok = (kt * 32 + j <= qi) && ((mw[r >> 2] >> (8 * (r & 3))) & 0xffu);
                sacc[r] = ok ? sacc[r] : -3.0e38f;
                mc = fmaxf(mc, sacc[r]);
            }
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            mn = fmaxf(m, mc);
            alpha = __expf(m - mn);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = sacc[r] > -1.0e38f ? __expf(sacc[r] - mn) : 0.f;
                sacc[r] = p;
                psum += p;
            }
        }
        psum += __shfl_xor(psum, 32);
        l = l * alpha + psum;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], sacc[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[r], sacc[r], o1, 0, 0, 0);
        }
    }
    // ---- merge the 4 key-interleaved states: O = sum_w O_w e^(m_w - M) / sum_w l_w e^(m_w - M)
    {
        float* ow = &Os[wave][n32 * OS + 4 * hi];
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                    // rows d = 8g + 4hi + {0..3} (+32 for the second d-tile)
            *reinterpret_cast<psalm_f32x4*>(ow + 8 * g) = psalm_f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
            *reinterpret_cast<psalm_f32x4*>(ow + 32 + 8 * g) = psalm_f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
        }
        if (hi == 0) { Ml[wave][0][n32] = m; Ml[wave][1][n32] = l; }
    }
    __syncthreads();
    {
        const int q = tid >> 3, d0 = (tid & 7) * 8;                       // thread -> query q of the tile, 8 head dims
        const int tq = qt * 32 + q;
        if (tq < L) {
            float M = -3.0e38f;
#pragma unroll
            for (int w = 0; w < 4; ++w) M = fmaxf(M, Ml[w][0][q]);
            float Lsum = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw_ = Ml[w][0][q];
                const float f = mw_ > -1.0e38f ? __expf(mw_ - M) : 0.f;
                Lsum += Ml[w][1][q] * f;
                const psalm_f32x4 a = *reinterpret_cast<const psalm_f32x4*>(&Os[w][q * OS + d0]);
                const psalm_f32x4 c = *reinterpret_cast<const psalm_f32x4*>(&Os[w][q * OS + d0 + 4]);
                acc[0] += a.x * f; acc[1] += a.y * f; acc[2] += a.z * f; acc[3] += a.w * f;
                acc[4] += c.x * f; acc[5] += c.y * f; acc[6] += c.z * f; acc[7] += c.w * f;
            }
            const float inv = Lsum > 0.f ? 1.f / Lsum : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] *= inv;
            if constexpr (SO) {
                const long row = (long)b * L + tq;
                const float sc = 1.f / so_inv[row];                      // power of two: exact
                unsigned hw[4], lw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned h0, h1, l0, l1;
                    psalm_split_words(acc[2 * k] * sc, h0, l0);
                    psalm_split_words(acc[2 * k + 1] * sc, h1, l1);
                    hw[k] = h0 | (h1 << 16);
                    lw[k] = l0 | (l1 << 16);
                }
                unsigned short* d = reinterpret_cast<unsigned short*>(out) + row * ldo + o_off + h * HD + d0;
                *reinterpret_cast<psalm_u32x4*>(d) = psalm_u32x4{hw[0], hw[1], hw[2], hw[3]};
                *reinterpret_cast<psalm_u32x4*>(d + so_kp) = psalm_u32x4{lw[0], lw[1], lw[2], lw[3]};
            } else {
                st8(out + ((long)b * L + tq) * ldo + o_off + h * HD + d0, acc);
            }
        }
    }
    }
}

extern "C" long psalm_causal_attention_f32_workspace(int B, int L, int heads) {
    const long Lp = (L + 31) / 32 * 32;
    return 2L * B * heads * Lp * 64 * (long)sizeof(float) + (long)B * Lp + (long)B * (Lp / 32) + 64;
}

// Phi prefill attention for fp32 buffers, same operands as psalm_causal_attention + a workspace of psalm_causal_attention_f32_workspace
// bytes (16-byte aligned).  head_dim 64, rotary dim 32; column offsets / row strides multiples of 4 elements.
static int causal_attention_f32_impl(const float* qkv, long ld, int q_off, int k_off, int v_off, void* out, long ldo, int o_off,
                                     const float* cos_table, const float* sin_table, const unsigned char* key_mask, void* workspace,
                                     int B, int L, int heads, int head_dim, int rot, void* stream, const float* so_inv, int so_kp,
                                     const char* name) {
    PSALM_CHECK_ARG(head_dim == 64 && rot == 32, "psalm_causal_attention_f32: head_dim 64, rotary dim 32 (Phi-1.5)");
    PSALM_CHECK_ARG(ld % 4 == 0 && q_off % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && (uintptr_t)qkv % 16 == 0 &&
                        (uintptr_t)out % 16 == 0 && workspace && (uintptr_t)workspace % 16 == 0 &&
                        (so_inv ? (ldo % 8 == 0 && o_off % 8 == 0 && so_kp % 8 == 0) : (ldo % 4 == 0 && o_off % 4 == 0)),
                    "psalm_causal_attention_f32: 16-byte aligned rows / offsets and a workspace");
    if (B == 0 || L == 0) return 0;
    const int Lp = (L + 31) / 32 * 32;
    float* Qr = (float*)workspace;
    float* Kr = Qr + (long)B * heads * Lp * 64;
    unsigned char* Mk = (unsigned char*)(Kr + (long)B * heads * Lp * 64);
    unsigned char* Tk = Mk + (long)B * Lp;                                 // per 32-key tile: all keys real and un-masked
    PSALM_CHECK_ARG(((long)L - 1) * ld * 4 + 256 < 0x7fffffffL, "psalm_causal_attention_f32: L * ld * 4 must stay below 2 GiB (buffer-descriptor fetches)");
    const float scale = 1.0f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(phi_rope_prep_f32_kernel, dim3(Lp / 32, heads, B), dim3(256), 0, s, qkv, ld, q_off, k_off, cos_table, sin_table, key_mask,
                       Qr, Kr, Mk, Tk, L, Lp, heads, scale);
    const int pair = 1;                                                   // balanced pairs of query tiles per block (r02n; the one-tile form stays in the kernel)
    const int nqt = Lp / 32;
    const dim3 grid(pair ? (nqt + 1) / 2 : nqt, heads, B);
    const int xcd_heads = ((heads * B) % 8 == 0 && psalm_get_tuning(PSALM_TUNE_ATTN_XCD_HEADS)) ? 1 : 0;
    if (so_inv)
        hipLaunchKernelGGL(causal_attention_f32_splitk_kernel<true>, grid, dim3(256), 0, s, (const float*)Qr, (const float*)Kr,
                           (const unsigned char*)Mk, (const unsigned char*)Tk, qkv, ld, v_off, (float*)out, ldo, o_off, L, Lp, heads, so_inv, so_kp, pair, xcd_heads);
    else
        hipLaunchKernelGGL(causal_attention_f32_splitk_kernel<false>, grid, dim3(256), 0, s, (const float*)Qr, (const float*)Kr,
                           (const unsigned char*)Mk, (const unsigned char*)Tk, qkv, ld, v_off, (float*)out, ldo, o_off, L, Lp, heads, (const float*)nullptr, 0, pair, xcd_heads);
    PSALM_LAUNCH_END(name);
}
extern "C" int psalm_causal_attention_f32(const float* qkv, long ld, int q_off, int k_off, int v_off, float* out, long ldo, int o_off,
                                          const float* cos_table, const float* sin_table, const unsigned char* key_mask, void* workspace,
                                          int B, int L, int heads, int head_dim, int rot, void* stream) {
    return causal_attention_f32_impl(qkv, ld, q_off, k_off, v_off, out, ldo, o_off, cos_table, sin_table, key_mask, workspace, B, L, heads,
                                     head_dim, rot, stream, nullptr, 0, "psalm_causal_attention_f32");
}
// ... whose output leaves as split-f16 operand columns of the NEXT GEMM ([dense | fc2], modeling_phi.py:189-260): row r of split_out (row
// stride ld_split f16) receives hi at columns split_col_off + h*64 + d and lo split_kp columns further, scaled by 1 / split_inv[r] -- the
// row scales psalm_gemm_x3_split wrote for the same rows (its bound covers the attention output: a convex combination of v rows).
extern "C" int psalm_causal_attention_f32_split(const float* qkv, long ld, int q_off, int k_off, int v_off, void* split_out, long ld_split,
                                                int split_kp, int split_col_off, const float* split_inv, const float* cos_table,
                                                const float* sin_table, const unsigned char* key_mask, void* workspace, int B, int L,
                                                int heads, int head_dim, int rot, void* stream) {
    PSALM_CHECK_ARG(split_out && split_inv && ld_split >= 2L * split_kp && split_col_off + heads * 64 <= split_kp,
                    "psalm_causal_attention_f32_split: split buffer rows of >= 2*split_kp f16 and the row scales");
    return causal_attention_f32_impl(qkv, ld, q_off, k_off, v_off, split_out, ld_split, split_col_off, cos_table, sin_table, key_mask, workspace,
                                     B, L, heads, head_dim, rot, stream, split_inv, split_kp, "psalm_causal_attention_f32_split");
}

extern "C" int psalm_causal_attention(const void* qkv, int dtype, long ld, int q_off, int k_off, int v_off, void* out, long ldo,
                                      int o_off, const float* cos_table, const float* sin_table,
                                      const unsigned char* key_mask, int B, int L, int heads, int head_dim, int rot,
                                      void* stream) {
    PSALM_CHECK_ARG(head_dim == 64, "psalm_causal_attention: head_dim must be 64 (Phi-1.5)");
    PSALM_CHECK_ARG(rot == 32, "psalm_causal_attention: rotary dim must be 32 (Phi-1.5: 0.5 * 64)");
    if (B == 0 || L == 0) return 0;
    const float scale = 1.0f / sqrtf((float)head_dim);
    if (dtype == PSALM_F32 && ld % 4 == 0 && ldo % 4 == 0 && q_off % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && o_off % 4 == 0 &&
        (uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 16 == 0) {          // fp32 matrix-core kernel (16-byte accesses)
        hipLaunchKernelGGL((causal_attention_f32_mfma_kernel<64, 32, 4>), dim3(cdiv(L, 128), heads, B), dim3(256), 0, (hipStream_t)stream,
                           (const float*)qkv, ld, q_off, k_off, v_off, (float*)out, ldo, o_off, cos_table, sin_table, key_mask, L,
                           heads, scale);
        PSALM_LAUNCH_END("psalm_causal_attention");
    }
    PSALM_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((causal_attention_kernel<T, 64, 32, 128>), dim3(cdiv(L, 128), heads, B), dim3(128), 0,
                           (hipStream_t)stream, (const T*)qkv, ld, q_off, k_off, v_off, (T*)out, ldo, o_off, cos_table,
                           sin_table, key_mask, L, heads, scale);
    });
    PSALM_LAUNCH_END("psalm_causal_attention");
}

// ============================================================================================ generic MHA (predictor)
// q (B,Lq,*) row stride ldq, k/v (B,Lk,*) row strides ldk/ldv; head h uses columns [h*32, h*32+32).
// mask (B,Lq,Lk) u8, 1 = NOT allowed; row_all_masked (B,Lq) u8: 1 -> ignore the mask for that row (TD:647).
// One wave per (b, h, q): lanes split the keys (64 per step), each lane keeps a private online-softmax state
// (m, l, o[32]); the 64 states are merged with a butterfly at the end.
template <typename T, int HD>
__global__ void __launch_bounds__(256) mha_attention_kernel(const T* __restrict__ Q, long ldq, const T* __restrict__ K, long ldk,
                                                            const T* __restrict__ V, long ldv, T* __restrict__ O, long ldo,
                                                            const unsigned char* __restrict__ mask,
                                                            const unsigned char* __restrict__ row_all_masked, int B, int Lq,
                                                            int Lk, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)B * heads * Lq;
    if (w >= total) return;
    const int qi = (int)(w % Lq);
    const int h = (int)((w / Lq) % heads);
    const int b = (int)(w / ((long)Lq * heads));
    float q[HD], o[HD];
    {
        const T* p = Q + ((long)b * Lq + qi) * ldq + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) { q[c] = ldf(p + c) * scale; o[c] = 0.f; }
    }
    const unsigned char* mrow = nullptr;
    if (mask && !(row_all_masked && row_all_masked[(long)b * Lq + qi])) mrow = mask + ((long)b * Lq + qi) * Lk;
    float m = -3.0e38f, l = 0.f;
    for (int j = lane; j < Lk; j += 64) {
        if (mrow && mrow[j]) continue;
        const T* kp = K + ((long)b * Lk + j) * ldk + h * HD;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * ldf(kp + c);
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn), p = __expf(s - mn);
        l = l * alpha + p;
        const T* vp = V + ((long)b * Lk + j) * ldv + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = o[c] * alpha + p * ldf(vp + c);
        m = mn;
    }
    const float mall = wave_max(m);
    const float f = (m > -1.0e38f) ? __expf(m - mall) : 0.f;
    const float lall = wave_sum(l * f);
    const float inv = lall > 0.f ? 1.f / lall : 0.f;
    T* op = O + ((long)b * Lq + qi) * ldo + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        const float v = wave_sum(o[c] * f);
        if (lane == (c & 63)) stf(op + c, v * inv);
    }
}

extern "C" int psalm_mha_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out, long ldo,
                                   int dtype, const unsigned char* mask, const unsigned char* row_all_masked, int B, int Lq,
                                   int Lk, int heads, int head_dim, void* stream) {
    PSALM_CHECK_ARG(head_dim == 32, "psalm_mha_attention: head_dim must be 32");
    const long total = (long)B * heads * Lq;
    if (total == 0 || Lk == 0) return 0;
    const float scale = 1.0f / sqrtf((float)head_dim);
    PSALM_DISPATCH(dtype, T, {
        hipLaunchKernelGGL((mha_attention_kernel<T, 32>), dim3(cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)q,
                           ldq, (const T*)k, ldk, (const T*)v, ldv, (T*)out, ldo, mask, row_all_masked, B, Lq, Lk, heads, scale);
    });
    PSALM_LAUNCH_END("psalm_mha_attention");
}

// ---- fp32 matrix-core form, split over keys (v_mfma_f32_16x16x4_f32), used for fp32 buffers.
// The per-query-wave kernel above re-reads every K / V row once per query (800 waves x Lk rows x 256 B from L2 at Lk = 16384) and does the
// products on the VALU (2.6 ms of a 34 ms f16x3 image).  Here one wavefront owns ALL queries (<= 128, NQT tiles of 16) of one head for a
// chunk of 64..256 keys: Q fragments resident in registers, K / V tiles of 64 keys staged in LDS once, swapped products
// (S^T = K . Q^T, O^T += V^T . P^T with P straight from the score registers, as in the window kernel), an online-softmax state per
// query tile.  heads x ceil(Lk / 256) wavefronts; with more than one chunk each writes (O unnormalised, m, l) to the workspace and
// mha_f32_combine_kernel merges the chunks.
// keys per wavefront: as few as fill the chip (heads x chunks >= ~1024 wavefronts = one per SIMD), at least one 64-key LDS tile
static int mha_f32_chunk(int B, int heads, int Lk) {
    long c = ((long)Lk * heads * B + 1023) / 1024;
    c = (c + 63) / 64 * 64;
    return (int)(c < 64 ? 64 : (c > 256 ? 256 : c));
}
template <int NQT>
__global__ void __launch_bounds__(64) mha_attention_f32_mfma_kernel(const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk,
                                                                    const float* __restrict__ V, long ldv, float* __restrict__ O, long ldo,
                                                                    const unsigned char* __restrict__ mask,
                                                                    const unsigned char* __restrict__ row_all_masked, float* __restrict__ part,
                                                                    int Lq, int Lk, int heads, int splits, int chunk, float scale) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int HD = 32, LS = HD + 4, KT = 64;
    __shared__ __attribute__((aligned(16))) float Ks[KT * LS];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LS];
    const int lane = threadIdx.x, n16 = lane & 15, kk = lane >> 4;
    const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int k_lo = sp * chunk, k_hi = min(Lk, k_lo + chunk);
    float qf[NQT][8];
    const unsigned char* mrow[NQT];
    bool qok[NQT], use_m[NQT];
    {
        // r05: the query fragments and the rows' all-masked flags of ALL tiles are fetched before the first one is used (the flag load sat
        // behind a branch per tile: NQT dependent round trips before the first key tile; the flag address is always loadable)
        f32x4 qa[NQT], qc[NQT];
        unsigned char am[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const int qi = 16 * t + n16;
            const float* p = Q + ((long)b * Lq + min(qi, Lq - 1)) * ldq + h * HD + 8 * kk;
            qa[t] = reinterpret_cast<const f32x4*>(p)[0];
            qc[t] = reinterpret_cast<const f32x4*>(p)[1];
            am[t] = row_all_masked ? row_all_masked[(long)b * Lq + min(qi, Lq - 1)] : (unsigned char)0;
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const int qi = 16 * t + n16;
            qok[t] = qi < Lq;
            qf[t][0] = qa[t].x * scale; qf[t][1] = qa[t].y * scale; qf[t][2] = qa[t].z * scale; qf[t][3] = qa[t].w * scale;
            qf[t][4] = qc[t].x * scale; qf[t][5] = qc[t].y * scale; qf[t][6] = qc[t].z * scale; qf[t][7] = qc[t].w * scale;
            // mask row of this lane's query (always a loadable address when a mask is given; `use_m` says whether it applies)
            mrow[t] = mask ? mask + ((long)b * Lq + min(qi, Lq - 1)) * Lk : nullptr;
            use_m[t] = mask && qok[t] && !am[t];
        }
    }
    const bool fast_mask = mask && (Lk & 3) == 0 && (((uintptr_t)mask) & 3) == 0;
    f32x4 o0[NQT], o1[NQT];
    float m[NQT], l[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) { o0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; o1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; m[t] = -3.0e38f; l[t] = 0.f; }
    for (int kb = k_lo; kb < k_hi; kb += KT) {
        __syncthreads();
        {                                                                 // 64 keys x 8 float4 per operand (rows past Lk: clamped, masked below)
            // r05: the 16 loads of a tile are all in flight before the first LDS store waits for one (rolled: 8 dependent round trips per tile)
            f32x4 kt_[KT * (HD / 4) / 64], vt_[KT * (HD / 4) / 64];
#pragma unroll
            for (int i = 0; i < KT * (HD / 4) / 64; ++i) {
                const int e = lane + 64 * i, r = e >> 3, c4 = (e & 7) * 4;
                const long row = (long)b * Lk + min(kb + r, Lk - 1);
                kt_[i] = *reinterpret_cast<const f32x4*>(K + row * ldk + h * HD + c4);
                vt_[i] = *reinterpret_cast<const f32x4*>(V + row * ldv + h * HD + c4);
            }
            PSALM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < KT * (HD / 4) / 64; ++i) {
                const int e = lane + 64 * i, r = e >> 3, c4 = (e & 7) * 4;
                *reinterpret_cast<f32x4*>(&Ks[r * LS + c4]) = kt_[i];
                *reinterpret_cast<f32x4*>(&Vs[r * LS + c4]) = vt_[i];
            }
        }
        __syncthreads();
        unsigned mbn[NQT];                                                // blocked flags of the NEXT key tile (byte r = key kj + r)
#pragma unroll
        for (int t = 0; t < NQT; ++t) mbn[t] = fast_mask ? *reinterpret_cast<const unsigned*>(mrow[t] + min(kb + 4 * kk, Lk - 4)) : 0u;
#pragma unroll 1
        for (int tk = 0; tk < KT / 16; ++tk) {
            const int key0 = kb + 16 * tk;
            if (key0 >= k_hi) break;
            const float* kp = &Ks[(16 * tk + n16) * LS + 8 * kk];
            const psalm_f32x4 k0 = reinterpret_cast<const psalm_f32x4*>(kp)[0], k1 = reinterpret_cast<const psalm_f32x4*>(kp)[1];
            float va[4], vb[4];                                           // V^T fragments of the 4 contraction steps (d-tiles 0 / 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] = Vs[(16 * tk + 4 * kk + r) * LS + n16]; vb[r] = Vs[(16 * tk + 4 * kk + r) * LS + 16 + n16]; }
            const int kj = key0 + 4 * kk;                                 // this lane's 4 keys: kj .. kj + 3
            // blocked flags (byte r = key kj + r) of all query tiles, fetched together and unconditionally: a load guarded per tile
            // compiles to a branch + s_waitcnt vmcnt(0) each, i.e. NQT dependent L2 round trips per key tile
            unsigned mbv[NQT];
            if (fast_mask) {
                // software-pipelined by one key tile: this tile's flags were fetched during the previous tile (`mbn`), the next tile's are
                // issued now and first needed one tile of MFMAs later (keys past Lk are excluded by the range test below)
#pragma unroll
                for (int t = 0; t < NQT; ++t) mbv[t] = mbn[t];
                const int kc = min(kj + 16, Lk - 4);
#pragma unroll
                for (int t = 0; t < NQT; ++t) mbn[t] = *reinterpret_cast<const unsigned*>(mrow[t] + kc);
            } else {
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    mbv[t] = 0;
                    if (mask) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) mbv[t] |= (unsigned)mrow[t][min(kj + r, Lk - 1)] << (8 * r);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[t][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[t][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[t][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[t][3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[t][4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[t][5], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[t][6], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[t][7], acc, 0, 0, 0);
                const unsigned mb = use_m[t] ? mbv[t] : 0u;
                float mc = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (kj + r < k_hi) && !((mb >> (8 * r)) & 0xffu);
                    acc[r] = ok ? acc[r] : -3.0e38f;
                    mc = fmaxf(mc, acc[r]);
                }
                mc = fmaxf(mc, __shfl_xor(mc, 16));
                mc = fmaxf(mc, __shfl_xor(mc, 32));
                const float mn = fmaxf(m[t], mc);
                const float alpha = __expf(m[t] - mn);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = acc[r] > -1.0e38f ? __expf(acc[r] - mn) : 0.f;
                    acc[r] = p;
                    ps += p;
                }
                ps += __shfl_xor(ps, 16);
                ps += __shfl_xor(ps, 32);
                l[t] = l[t] * alpha + ps;
                m[t] = mn;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o0[t][r] *= alpha; o1[t][r] *= alpha; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[r], acc[r], o0[t], 0, 0, 0);
                    o1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[r], acc[r], o1[t], 0, 0, 0);
                }
            }
            if (fast_mask) {
#pragma unroll
                for (int t = 0; t < NQT; ++t) PSALM_OPAQUE_VGPR(mbn[t]);    // (keeps the optimiser from sinking each load into its `use_m` branch)
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const int qi = 16 * t + n16;
        if (!qok[t]) continue;
        if (splits == 1) {
            const float inv = l[t] > 0.f ? 1.f / l[t] : 0.f;
            float* op = O + ((long)b * Lq + qi) * ldo + h * HD + 4 * kk;
            *reinterpret_cast<psalm_f32x4*>(op) = psalm_f32x4{o0[t][0] * inv, o0[t][1] * inv, o0[t][2] * inv, o0[t][3] * inv};
            *reinterpret_cast<psalm_f32x4*>(op + 16) = psalm_f32x4{o1[t][0] * inv, o1[t][1] * inv, o1[t][2] * inv, o1[t][3] * inv};
        } else {                                                          // partial state: [O (32) | m | l | pad 2] per (b, h, split, q)
            float* pp = part + ((((long)b * heads + h) * splits + sp) * Lq + qi) * 36;
            *reinterpret_cast<psalm_f32x4*>(pp + 4 * kk) = psalm_f32x4{o0[t][0], o0[t][1], o0[t][2], o0[t][3]};
            *reinterpret_cast<psalm_f32x4*>(pp + 16 + 4 * kk) = psalm_f32x4{o1[t][0], o1[t][1], o1[t][2], o1[t][3]};
            if (kk == 0) { pp[32] = m[t]; pp[33] = l[t]; }
        }
    }
}

// merge of the key chunks: thread = (b, h, q, d); weights e^(m_s - M)
__global__ void __launch_bounds__(256) mha_f32_combine_kernel(const float* __restrict__ part, float* __restrict__ O, long ldo, int B, int Lq, int heads,
                                                              int splits) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * heads * Lq * 32;
    if (idx >= total) return;
    const int d = (int)(idx & 31);
    const int qi = (int)((idx >> 5) % Lq);
    const int h = (int)((idx >> 5) / Lq % heads);
    const int b = (int)((idx >> 5) / ((long)Lq * heads));
    const float* p0 = part + (((long)b * heads + h) * splits * Lq + qi) * 36;
    const long sstride = (long)Lq * 36;
    float M = -3.0e38f;
    for (int s = 0; s < splits; ++s) M = fmaxf(M, p0[s * sstride + 32]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float ms = p0[s * sstride + 32];
        const float f = ms > -1.0e38f ? __expf(ms - M) : 0.f;
        L += p0[s * sstride + 33] * f;
        acc += p0[s * sstride + d] * f;
    }
    O[((long)b * Lq + qi) * ldo + h * 32 + d] = L > 0.f ? acc / L : 0.f;
}

// ... the form the entry point launches for <= 256 key chunks (r05).  Same sums in the same order -- thread (q, d) still adds its chunks one after
// the other, s = 0, 1, ... -- but the chunk weights e^(m_s - M) are computed ONCE per (q, s) by the 32 lanes of a query together (they were
// recomputed by each of the 32 head-dim threads, behind two dependent loads per chunk) and parked in LDS with the l_s; the per-thread loop
// then touches memory once per chunk (the coalesced 128-byte O row).  r05a trace: 23 us average, 82 us for the 16384-key level (128 chunks),
// for 25 600 threads' worth of work; 18 launches per image.
__global__ void __launch_bounds__(256) mha_f32_combine_lds_kernel(const float* __restrict__ part, float* __restrict__ O, long ldo, int B, int Lq,
                                                                  int heads, int splits) {
    __shared__ float fs[8][256], ls[8][256];
    const int ql = threadIdx.x >> 5, d = threadIdx.x & 31;
    const int qi = blockIdx.x * 8 + ql, h = blockIdx.y, b = blockIdx.z;
    const bool live = qi < Lq;
    const float* p0 = part + (((long)b * heads + h) * splits * Lq + (live ? qi : 0)) * 36;
    const long sstride = (long)Lq * 36;
    float M = -3.0e38f;
    for (int s_ = d; s_ < splits; s_ += 32) M = fmaxf(M, p0[s_ * sstride + 32]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 32));          // the 32 lanes of this query (a maximum: any order)
    for (int s_ = d; s_ < splits; s_ += 32) {
        const float ms = p0[s_ * sstride + 32];
        fs[ql][s_] = ms > -1.0e38f ? __expf(ms - M) : 0.f;
        ls[ql][s_] = p0[s_ * sstride + 33];
    }
    __syncthreads();
    if (!live) return;
    float L = 0.f, acc = 0.f;
    for (int s_ = 0; s_ < splits; ++s_) {
        const float f = fs[ql][s_];
        L += ls[ql][s_] * f;
        acc += p0[s_ * sstride + d] * f;
    }
    O[((long)b * Lq + qi) * ldo + h * 32 + d] = L > 0.f ? acc / L : 0.f;
}

extern "C" long psalm_mha_attention_f32_workspace(int B, int heads, int Lq, int Lk) {
    const int splits = cdiv(Lk, mha_f32_chunk(B, heads, Lk));
    return splits > 1 ? (long)B * heads * splits * Lq * 36 * (long)sizeof(float) : 0;
}

// fp32 q / k / v / out (row strides in elements, 16-byte aligned rows, head_dim 32, Lq <= 128); workspace: psalm_mha_attention_f32_workspace bytes.
extern "C" int psalm_mha_attention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* out, long ldo,
                                       const unsigned char* mask, const unsigned char* row_all_masked, void* workspace, int B, int Lq, int Lk,
                                       int heads, int head_dim, void* stream) {
    PSALM_CHECK_ARG(head_dim == 32, "psalm_mha_attention_f32: head_dim must be 32");
    PSALM_CHECK_ARG(Lq >= 1 && Lq <= 128, "psalm_mha_attention_f32: 1 <= Lq <= 128");
    PSALM_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 &&
                        (uintptr_t)v % 16 == 0 && (uintptr_t)out % 16 == 0, "psalm_mha_attention_f32: 16-byte aligned rows");
    if (B == 0 || Lk == 0) return 0;
    const int chunk = mha_f32_chunk(B, heads, Lk);
    const int splits = cdiv(Lk, chunk);
    PSALM_CHECK_ARG(splits == 1 || workspace != nullptr, "psalm_mha_attention_f32: workspace required when the keys are split");
    const float scale = 1.0f / sqrtf((float)head_dim);
    const dim3 grid(splits, heads, B);
    const int nqt = cdiv(Lq, 16);
    hipStream_t s = (hipStream_t)stream;
#define MHA_F32_LAUNCH(N_) hipLaunchKernelGGL((mha_attention_f32_mfma_kernel<N_>), grid, dim3(64), 0, s, q, ldq, k, ldk, v, ldv, out, ldo, mask, \
                                              row_all_masked, (float*)workspace, Lq, Lk, heads, splits, chunk, scale)
    if (nqt <= 1) MHA_F32_LAUNCH(1);
    else if (nqt <= 2) MHA_F32_LAUNCH(2);
    else if (nqt <= 4) MHA_F32_LAUNCH(4);
    else if (nqt <= 7) MHA_F32_LAUNCH(7);
    else MHA_F32_LAUNCH(8);
#undef MHA_F32_LAUNCH
    if (splits > 1 && splits <= 256) {
        hipLaunchKernelGGL(mha_f32_combine_lds_kernel, dim3(cdiv(Lq, 8), heads, B), dim3(256), 0, s, (const float*)workspace, out, ldo, B, Lq, heads, splits);
    } else if (splits > 1) {
        const long total = (long)B * heads * Lq * 32;
        hipLaunchKernelGGL(mha_f32_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)workspace, out, ldo, B, Lq,
                           heads, splits);
    }
    PSALM_LAUNCH_END("psalm_mha_attention_f32");
}

// ============================================================================================ attention-mask generation
// masks (B*Q, h, w) f32 logits -> bilinear (align_corners=False, PyTorch index rule) to (Ht,Wt) -> mask = logit < 0
// (== sigmoid < 0.5) -> u8 (B*Q, Ht*Wt); row_all_masked[bq] = 1 when every key is masked.
__device__ __forceinline__ float bilinear_at(const float* __restrict__ src, int h, int w, float sy, float sx) {
    sy = fmaxf(sy, 0.f);
    sx = fmaxf(sx, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * src[(long)y0 * w + x0] + lx * src[(long)y0 * w + x1]) +
           ly * (hx * src[(long)y1 * w + x0] + lx * src[(long)y1 * w + x1]);
}

__global__ void __launch_bounds__(256) attn_mask_kernel(const float* __restrict__ masks, unsigned char* __restrict__ out,
                                                        unsigned char* __restrict__ row_all_masked, int h, int w, int Ht,
                                                        int Wt) {
    __shared__ int cnt[4];
    const int bq = blockIdx.x, tid = threadIdx.x;
    const float* src = masks + (long)bq * h * w;
    const float sh = (float)h / Ht, sw = (float)w / Wt;
    int unmasked = 0;
    for (int i = tid; i < Ht * Wt; i += 256) {
        const int y = i / Wt, x = i % Wt;
        const float v = bilinear_at(src, h, w, sh * (y + 0.5f) - 0.5f, sw * (x + 0.5f) - 0.5f);
        const unsigned char mk = v < 0.f ? 1 : 0;
        out[(long)bq * Ht * Wt + i] = mk;
        unmasked += !mk;
    }
    const float tot = wave_sum((float)unmasked);
    if ((tid & 63) == 0) cnt[tid >> 6] = (int)tot;
    __syncthreads();
    if (tid == 0) row_all_masked[bq] = (cnt[0] + cnt[1] + cnt[2] + cnt[3]) == 0 ? 1 : 0;
}

extern "C" int psalm_attn_mask(const float* masks, unsigned char* out, unsigned char* row_all_masked, int BQ, int h, int w,
                               int Ht, int Wt, void* stream) {
    if (BQ == 0) return 0;
    hipLaunchKernelGGL(attn_mask_kernel, dim3(BQ), dim3(256), 0, (hipStream_t)stream, masks, out, row_all_masked, h, w, Ht, Wt);
    PSALM_LAUNCH_END("psalm_attn_mask");
}
