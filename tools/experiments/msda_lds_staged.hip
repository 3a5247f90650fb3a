// EXPERIMENT, NOT PART OF THE PRODUCT LIBRARY (r04; not compiled by psalm_amd/build.py): the LDS-staged form of the fused MSDeformAttn kernel
// that VERDICT r02 / r03 asked for ("LDS-staged MSDA neighbourhoods per (8 x 8 query tile, head)").  Built, tested on the emulator and on the
// MI355X (bit-identical to the L2-gather kernel, tests of r04e), measured, and taken out:
//     tools/bench_msda.py, S = 21504, 8 heads x 32, fp32:   L2-gather kernel (XCD bands) 57.1 us   |   LDS-staged kernel 108.3 us
//     in the model (bench.py, 6 launches per image):        54.1 us, 47.07 images/s               |   112.0 us, 46.49 images/s
// (profiles/r04e_bench_msda.jsonl, r04e_bench_msda_model_ab.txt).  Why it loses: the corner FETCHES were not what bounds the kernel -- the per-sample
// arithmetic is (softmax pieces, sampling location, bilinear taps, clamps: ~60 VALU instructions per sample, computed redundantly by the 4
// channel-group lanes of a (query, head)); the staged form keeps all of it, adds the copy + barrier in front of every block, and runs at 3
// waves per SIMD (64 KB of LDS per block) where the gather kernel hides its L2 latency with 7.  The lever that follows from the measurement
// is in the product kernel since r04: the taps are computed ONCE per (query, head, point) and broadcast inside the 4-lane group (msda.hip).
// To build this again: paste the three pieces below back into psalm_amd/csrc/msda.hip (kernel next to msda_fused8_kernel, geometry + policy in
// front of psalm_msda_forward, launch at the top of psalm_msda_fused).
#if 0
// ---- LDS-staged form of the fused kernel (r04; VERDICT r02 / r03: "LDS-staged MSDA neighbourhoods").  The fused8 kernel above fetches every
// bilinear corner from L2: 2.06 M samples x 4 corners x 128 B = 1.05 GB of L2 requests per launch for a 22 MB value tensor (r02h counters:
// 8.2 M requests, 89 % L2 hits, ~20 TB/s -- L2-request bound, 0.16 of the HBM roofline).  The samples of neighbouring queries cluster: a query
// at normalised position p samples every level around p (offsets of a few level pixels), so all queries of one TILE of the normalised image
// -- n_l x n_l pixels of level l, n_l = 2 / 4 / 8 for the 32 / 64 / 128 pyramids: 84 queries -- read the same three small windows.
// Block = (tile, head): the windows (n_l + 2 R + 2)^2 positions x 32 channels of the head, R = MSDA_R level pixels of halo, 500 positions =
// 64 KB at R = 3) are copied ONCE into LDS with global_load_lds (8 positions per wave instruction; the 16-byte chunk order inside a position
// XOR-swizzled by the position so that the 16 lanes of a ds_read_b128 group spread over the banks), then every (query, 8-channel group)
// thread takes its 48 corner values from LDS.  A sample whose 2 x 2 corner patch leaves the window (offsets beyond R) reads that patch from
// global memory instead: any offset is handled, only slower.  L2 -> CU traffic: 2048 blocks x 64 KB = 131 MB instead of 1.05 GB.
#define MSDA_R 3
struct MsdaTiles { int tgy, tgx, n[3], ww[3], wbase[3], qoff[3], npos; };
template <typename TO>
__global__ void __launch_bounds__(384) PSALM_WAVES_PER_EU(3) msda_fused_lds_kernel(const float* __restrict__ value, MsdaLevels lv, MsdaTiles tl,
                                                             const float* __restrict__ ow, TO* __restrict__ out, int S, int M) {
    constexpr int L = 3, P = 4, LP = 12, D = 32;
    __shared__ __attribute__((aligned(16))) float win[504 * D];          // positions x 32 channels (chunk-swizzled), 63 copy instructions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (tile, head): XCD k (= block % 8) owns the k-th band of tile rows (its L2 holds that band of every level), heads innermost
    const int ntiles = tl.tgy * tl.tgx;
    int tile, m;
    if (tl.tgy % 8 == 0) {
        const int band = blockIdx.x & 7, j = blockIdx.x >> 3, per_band = ntiles / 8;
        tile = band * per_band + j / M;
        m = j % M;
    } else { tile = blockIdx.x / M; m = blockIdx.x % M; }
    const int ty = tile / tl.tgx, tx = tile % tl.tgx;
    const int row_stride = M * D;
    // ---- stage the three windows
    for (int i = wave; i < 63; i += 6) {
        int pp = min(8 * i + (lane >> 3), tl.npos - 1);
        const int slot = lane & 7, chunk = slot ^ (pp & 7);
        int l = 0;
#pragma unroll
        for (int j = 1; j < L; ++j)
            if (pp >= tl.wbase[j]) l = j;
        int Hl = lv.H[0], Wl = lv.W[0], sl = lv.start[0], nl = tl.n[0], wwl = tl.ww[0], wb = tl.wbase[0];
#pragma unroll
        for (int j = 1; j < L; ++j)
            if (l == j) { Hl = lv.H[j]; Wl = lv.W[j]; sl = lv.start[j]; nl = tl.n[j]; wwl = tl.ww[j]; wb = tl.wbase[j]; }
        const int wy = (pp - wb) / wwl, wx = (pp - wb) - wy * wwl;
        const int y = min(max(ty * nl - MSDA_R - 1 + wy, 0), Hl - 1), x = min(max(tx * nl - MSDA_R - 1 + wx, 0), Wl - 1);
        psalm_glds16(value + ((long)(sl + y * Wl + x) * M + m) * D + chunk * 4, win + (long)i * 8 * D);
    }
    __syncthreads();                                                     // (drains the copies: vmcnt(0) + barrier)
    // ---- thread -> (query of the tile, 8-channel group)
    const int item = tid;
    if (item >= 84 * 4) return;
    const int qidx = item >> 2, g = item & 3;
    int lq = 0;
#pragma unroll
    for (int j = 1; j < L; ++j)
        if (qidx >= tl.qoff[j]) lq = j;
    int Hq = lv.H[0], Wq = lv.W[0], sq = lv.start[0], nq = tl.n[0], qo = tl.qoff[0];
#pragma unroll
    for (int j = 1; j < L; ++j)
        if (lq == j) { Hq = lv.H[j]; Wq = lv.W[j]; sq = lv.start[j]; nq = tl.n[j]; qo = tl.qoff[j]; }
    const int qy = ty * nq + (qidx - qo) / nq, qx = tx * nq + (qidx - qo) % nq;
    const int q = sq + qy * Wq + qx;
    const float ref_x = (qx + 0.5f) / Wq, ref_y = (qy + 0.5f) / Hq;
    const float* row = ow + (long)q * (M * LP * 3);
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
    const float* vb = value + m * D + g * 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        int Hl = lv.H[0], Wl = lv.W[0], sl = lv.start[0], nl = tl.n[0], wwl = tl.ww[0], wb = tl.wbase[0];
#pragma unroll
        for (int j = 1; j < L; ++j)
            if (l == j) { Hl = lv.H[j]; Wl = lv.W[j]; sl = lv.start[j]; nl = tl.n[j]; wwl = tl.ww[j]; wb = tl.wbase[j]; }
        const int oy = ty * nl - MSDA_R - 1, ox = tx * nl - MSDA_R - 1;   // level pixel of window position (0, 0)
        const float* vl = vb + (long)sl * row_stride;
#pragma unroll 2
        for (int p = 0; p < P; ++p) {
            const int i = l * P + p;
            // bilinear taps: the expression of msda_taps8 (ms_deform_im2col_cuda.cuh:38-89; corners clamped into the level, validity in the weights)
            const float h_im = (ref_y + offp[2 * i + 1] / Hl) * Hl - 0.5f;
            const float w_im = (ref_x + offp[2 * i] / Wl) * Wl - 0.5f;
            const bool inb = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
            const int h_low = (int)floorf(inb ? h_im : 0.f), w_low = (int)floorf(inb ? w_im : 0.f);
            const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
            const bool h0 = h_low >= 0, h1 = h_low + 1 <= Hl - 1, w0 = w_low >= 0, w1 = w_low + 1 <= Wl - 1;
            const int ha = max(h_low, 0), hb = min(h_low + 1, Hl - 1), wa = max(w_low, 0), wbx = min(w_low + 1, Wl - 1);
            const float cw[4] = {(inb && h0 && w0) ? hh * hw : 0.f, (inb && h0 && w1) ? hh * lw : 0.f,
                                 (inb && h1 && w0) ? lh * hw : 0.f, (inb && h1 && w1) ? lh * lw : 0.f};
            const int ys[4] = {ha, ha, hb, hb}, xs[4] = {wa, wbx, wa, wbx};
            float v[4][8];
            const bool in_win = !inb || (ha >= oy && hb < oy + wwl && wa >= ox && wbx < ox + wwl);
            if (in_win) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int pos = inb ? wb + (ys[c] - oy) * wwl + (xs[c] - ox) : wb;      // (a sample outside the padded image: weights 0, any slot)
                    const psalm_f32x4 a = *reinterpret_cast<const psalm_f32x4*>(&win[pos * D + (((2 * g) ^ (pos & 7)) << 2)]);
                    const psalm_f32x4 b = *reinterpret_cast<const psalm_f32x4*>(&win[pos * D + (((2 * g + 1) ^ (pos & 7)) << 2)]);
                    v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
                }
            } else {                                                     // offsets beyond the halo: the patch comes from global memory
#pragma unroll
                for (int c = 0; c < 4; ++c) ld8(vl + ((long)ys[c] * Wl + xs[c]) * row_stride, v[c]);
            }
            const float wgt = inb ? lg[i] * inv : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += wgt * (cw[0] * v[0][k] + cw[1] * v[1][k] + cw[2] * v[2][k] + cw[3] * v[3][k]);
        }
    }
    st8(out + ((long)q * M + m) * D + g * 8, acc);
}


// ---- host side
static int g_msda_lds = 1;
// Geometry of the LDS-staged fused kernel: fp32 value, head dim 32, one image, a 4 : 2 : 1 pyramid (any level order) whose coarsest level has
// even sides -> tile grid (hmin / 2) x (wmin / 2), n_l = 2 / 4 / 8 pixels of level l per tile side.
static bool msda_tiles(const MsdaLevels& lv, int L, int B, int S, int D, int value_dtype, MsdaTiles& tl) {
    if (!(L == 3 && B == 1 && D == 32 && value_dtype == PSALM_F32 && S > 0)) return false;
    int hmin = lv.H[0], wmin = lv.W[0];
    for (int l = 1; l < L; ++l) { hmin = lv.H[l] < hmin ? lv.H[l] : hmin; wmin = lv.W[l] < wmin ? lv.W[l] : wmin; }
    if (hmin % 2 || wmin % 2 || hmin < 2 || wmin < 2) return false;
    tl.tgy = hmin / 2; tl.tgx = wmin / 2;
    int seen = 0, pos = 0, qo = 0;
    for (int l = 0; l < L; ++l) {
        const int ny = lv.H[l] / tl.tgy, nx = lv.W[l] / tl.tgx;
        if (lv.H[l] % tl.tgy || lv.W[l] % tl.tgx || ny != nx || !(ny == 2 || ny == 4 || ny == 8) || (seen & ny)) return false;
        seen |= ny;
        tl.n[l] = ny; tl.ww[l] = ny + 2 * MSDA_R + 2; tl.wbase[l] = pos; tl.qoff[l] = qo;
        pos += tl.ww[l] * tl.ww[l];
        qo += ny * ny;
    }
    tl.npos = pos;
    return pos <= 504 && qo == 84;
}
// 1 if psalm_msda_fused takes the LDS-staged kernel for this level table (tests / bench attribution)
extern "C" int psalm_msda_lds_applicable(const int64_t* spatial_shapes_host, const int64_t* level_start_host, int L, int S, int B, int D,
                                         int value_dtype) {
    MsdaLevels lv = {};
    MsdaTiles tl = {};
    return g_msda_lds && fill_levels(lv, spatial_shapes_host, level_start_host, L, S) == 0 && msda_tiles(lv, L, B, S, D, value_dtype, tl) ? 1 : 0;
}


// ---- in psalm_msda_fused, before the gather launch
    MsdaTiles tl = {};
    if (g_msda_lds && (uintptr_t)value % 16 == 0 && (uintptr_t)out % 16 == 0 && msda_tiles(lv, L, B, S, D, value_dtype, tl)) {
        const unsigned nblk = (unsigned)(tl.tgy * tl.tgx * M);
        PSALM_DISPATCH(out_dtype, TO, {
            hipLaunchKernelGGL((msda_fused_lds_kernel<TO>), dim3(nblk), dim3(384), 0, (hipStream_t)stream, (const float*)value, lv, tl,
                               offsets_logits, (TO*)out, S, M);
        });
        PSALM_LAUNCH_END("psalm_msda_fused");
    }
#endif
