// dynamic_head.hip -- SURVEY 8(f-2): the producer of mask_logits, CondInstMaskHead.forward
// (condinst_head.py:1139-1164 with parse_dynamic_params :1120-1137 and aligned_bilinear :146-167),
// forward and backward, on gfx950.
//
//   in[n]   = cat( (coors[n] - location)/soi[level[n]] (2 ch), feat[img[n]] (C ch) )           per pixel
//   h1 = relu(W0 in + b0)  (8) ; h2 = relu(W1 h1 + b1)  (8) ; y = W2 h2 + b2  (1)                per pixel
//   logits[n] = aligned_bilinear(y, factor)                                                      [N,1,fH,fW]
// The reference runs three grouped F.conv2d over a [1, N*C, H, W] view (one 1x1 "conv" of 8 channels
// per instance: far too small for MIOpen/MFMA) plus pad/interpolate/pad/crop.  Here:
//
// Every shape that indexes a register array is a template parameter (C, rel, factor): a runtime bound there costs
// select chains or scratch.  The 233 parameters of an instance are wave-uniform: they are read with scalar loads and
// reach the FMAs as SGPR-pair operands of v_pk_fma_f32 (no LDS traffic, no VGPRs); the SGPR file holds ~70 at a
// time, so the MLP is cut into fenced segments (BXI_SEGMENT) that the scheduler may not merge.
//
// dyn_fwd_kernel   grid = N x tiles(8x32 of y).  Thread t evaluates pixels 2t and 2t+1 of the tile + 1 halo
//                  (each weight feeds both), y goes to LDS, the up-sampled tile is written with float2/float4
//                  stores.  Reads the feature tile (L2-resident, 4*C B per y pixel per instance), writes
//                  4*f^2 B per y pixel per instance.
// dyn_bwd2_kernel  (factors 1 and 2: every shipped configuration) the same work as dyn_bwd_kernel below at FOUR workgroups per CU: <= 128
//                  registers, 38 KB of LDS -- the operand rows in two passes of 2 x 5 output blocks, the tile's features staged once;
//                  described at the kernel.
// dyn_bwd_kernel   (factor 4 and the run-time factor) grid = B x tiles x kSlots.  A workgroup owns one 8x32 tile of one image and every
//                  kSlots-th instance of that image (list compacted by wave 0 with ballots):
//                    phase 1 (thread = pixel): dy by the transposed interpolation (a fixed (2f-1)^2 tap window,
//                       offsets/weights kept in registers, the next instance's taps prefetched), forward
//                       recomputed, MLP backward; accumulates d feat over its instances in registers
//                       and stages 51 operand rows of 256 pixels in LDS;
//                    phase 2 (thread = 4x5 block of outputs x 1/16 of the pixels): dW = dH^T X, a
//                       [8..19] x 256 contraction too thin for MFMA; biases are the column against a row of
//                       ones; the 16 pixel slices of a block are the 16 lanes of a DPP row
//                       -> one partial per (instance, tile).
//                  No atomics: partials are reduced in fixed order by dyn_reduce_kernel.
// dyn_reduce_kernel  g_params[n,q] = sum over tiles ; g_feat[b,c,p] = sum over slots.
// Measured (MI355X, B=2 C=16 100x128 -> 200x256, rocprofv3, round 5): N=32: fwd 12.0 us, bwd 27.2-27.8 us (dyn_bwd_kernel: 28.2-28.8), reduce 5.6 us;
// N=128: fwd 31 us, bwd 70.0-70.8 us (dyn_bwd_kernel: 78.6-79.4), reduce 6.3 us
// (PyTorch-ROCm running the reference's op sequence: 228 us forward, 785 us forward+backward).
#include "dynamic_head_device.hpp"

namespace bxi {

// ---- forward ---------------------------------------------------------------------------------------
template <int C, bool REL, int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void dyn_fwd_kernel(DynArgs a, const float* __restrict__ params, float* __restrict__ logits) {
    constexpr int kHalo = (kFwdR + 1 + DynHalo<F>::after) * (kFwdC + 1 + DynHalo<F>::after);
    constexpr int TR = kHalo <= 512 ? kFwdR : kYR;           // factors that need the halo on all four sides keep the 8-row tile
    __shared__ float ytile[512];
    const int tiles_x = (a.W + kFwdC - 1) / kFwdC, tiles_y = (a.H + TR - 1) / TR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    BXI_T(3, blockIdx.x, 0);
    dyn_tile_forward<C, REL, F, false, TR, kFwdC>(a, params, logits, n, ty, tx, ytile, nullptr, nullptr, DynEpi{});
    BXI_T(3, blockIdx.x, 4);
}

// ---- backward --------------------------------------------------------------------------------------
// weight of output index R on source index r (transposed aligned_bilinear)
__device__ __forceinline__ float upsample_weight(int R, int r, int f, int n_in) {
    int i0, i1; float fr;
    upsample_src(R, f, n_in, i0, i1, fr);
    return (i0 == r ? 1.f - fr : 0.f) + (i1 == r ? fr : 0.f);
}

// d y[r][c] = sum over the output pixels that sampled y[r][c].  The non-zero taps of source index r are
// R in [f r - f + 1 + f/2, f r + f - 1 + f/2] (2f-1 of them), or [0, f - 1 + f/2] for r = 0 (the clamp
// at :159-160 folds the first f/2 outputs onto source 0): a (2F-1)^2 window, loaded without branches.
template <int F>
__device__ __forceinline__ float gather_dy(const float* __restrict__ gz, int r, int c, int f, int H, int W) {
    const int OH = H * f, OW = W * f, half = f / 2;
    const int Rs = r == 0 ? 0 : f * r - f + 1 + half, Cs = c == 0 ? 0 : f * c - f + 1 + half;
    float dout = 0.f;
    if constexpr (F > 0) {
        constexpr int NT = 2 * F - 1;
        float wy[NT], wx[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            wy[k] = Rs + k < OH ? upsample_weight(Rs + k, r, F, H) : 0.f;
            wx[k] = Cs + k < OW ? upsample_weight(Cs + k, c, F, W) : 0.f;
        }
        float g[NT][NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) g[i][j] = gz[(int64_t)min(Rs + i, OH - 1) * OW + min(Cs + j, OW - 1)];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) dout += wy[i] * wx[j] * g[i][j];
    } else {
        const int Rb = min(f * (r + 1) + half, OH), Cb = min(f * (c + 1) + half, OW);
        for (int R = Rs; R < Rb; ++R) {
            const float wy = upsample_weight(R, r, f, H);
            for (int Cc = Cs; Cc < Cb; ++Cc) dout += wy * upsample_weight(Cc, c, f, W) * gz[(int64_t)R * OW + Cc];
        }
    }
    return dout;
}

// lanes whose bit is set in `mask` (a wave-uniform 64-bit lane mask) take a, the others b
__device__ __forceinline__ float lane_select(unsigned long long mask, float a, float b) {
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// sum over the 16 lanes of a DPP row; every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}

template <int C, bool REL, int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8)))
void dyn_bwd_kernel(DynArgs a, const float* __restrict__ params, const float* __restrict__ params_again,
                    const float* __restrict__ g_logits, float* __restrict__ feat_part /*[slots,B,C,H,W]*/,
                    float* __restrict__ param_part /*[N,T,P]*/, int slots /* 1 .. kSlots workgroups share an (image, tile) */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using D = Dyn<C, REL>;
    constexpr int CIN = D::CIN;
    constexpr int NROW = 1 + kDC + kDC + kDC + kDC + CIN;   // dout, dh2, dh1, h2, h1, in ; then a row of ones, a row of zeros
    constexpr int kOnes = NROW, kZeros = NROW + 1;
    constexpr int kChunkN = 1024;                           // instance ids scanned per pass
    float* rows = lds;                                      // [NROW + 2][kRowPad]
    __shared__ int mine[kChunkN + 2];                       // this workgroup's instances of the current chunk
    __shared__ int n_mine;
    const int tiles_x = (a.W + kYC - 1) / kYC, tiles_y = (a.H + kYR - 1) / kYR, T = tiles_x * tiles_y;
    int t = blockIdx.x;
    const int slot = t % slots; t /= slots;
    const int tile = t % T;
    const int b = t / T;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;
    const int lr = tid / kYC, lc = tid % kYC;
    const int r = ty * kYR + lr, c = tx * kYC + lc;
    const bool valid = r < a.H && c < a.W;
    const int rr = min(r, a.H - 1), cc = min(c, a.W - 1);   // clamped: loads need no branch
    const int f = factor_of<F>(a), OH = a.H * f, OW = a.W * f;
    constexpr int P = D::P, cin = CIN, off = REL ? 2 : 0, w1 = D::W1, w2 = D::W2;
    const int64_t HW = (int64_t)a.H * a.W;
    BXI_T(4, blockIdx.x, 0);
    rows[kOnes * kRowPad + tid] = 1.f;                      // operands of the bias gradients / of the padded block rows
    rows[kZeros * kRowPad + tid] = 0.f;

    // the features of this thread's pixel are the same for every instance: loaded once (uniform base + 32-bit
    // lane offset, so that the addresses cost one VGPR, not two per channel)
    float xf[C];
    {
        const float* fb = a.feat + (int64_t)b * C * HW;
        const unsigned po = (unsigned)(rr * a.W + cc);
#pragma unroll
        for (int k = 0; k < C; ++k) xf[k] = (fb + (int64_t)k * HW)[po];
    }
    // taps of the transposed up-sampling (gather_dy), fixed per thread when the window is small enough to keep
    constexpr bool kKeepTaps = F == 1 || F == 2;
    constexpr int NT = kKeepTaps ? 2 * F - 1 : 1;
    unsigned goff[NT * NT];
    float gw[NT * NT], g[NT * NT];
    if constexpr (kKeepTaps) {
        const int half = F / 2;
        const int Rs = rr == 0 ? 0 : F * rr - F + 1 + half, Cs = cc == 0 ? 0 : F * cc - F + 1 + half;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int jx = 0; jx < NT; ++jx) {
                const float wy = Rs + i < OH ? upsample_weight(Rs + i, rr, F, a.H) : 0.f;
                const float wx = Cs + jx < OW ? upsample_weight(Cs + jx, cc, F, a.W) : 0.f;
                gw[i * NT + jx] = wy * wx;
                goff[i * NT + jx] = (unsigned)(min(Rs + i, OH - 1) * OW + min(Cs + jx, OW - 1));
            }
    }
    auto load_taps = [&](int n) {
        const float* gz = g_logits + (int64_t)n * OH * OW;
#pragma unroll
        for (int q = 0; q < NT * NT; ++q) g[q] = gz[goff[q]];
    };

    v2f dfeat[C / 2];
#pragma unroll
    for (int k = 0; k < C / 2; ++k) dfeat[k] = v2f{0.f, 0.f};

    // ---- phase-2 role of this thread (fixed for the whole kernel) ------------------------------------------
    // rows: 0 dout | 1.. dh2 | 9.. dh1 | 17.. h2 | 25.. h1 | 33.. in.   Output families, each against (X, ones):
    //   0: dh1[4 og ..] x (in, 1)  -> dW0, db0      NG0 column groups of 5
    //   1: dh2[4 og ..] x (h1, 1)  -> dW1, db1      2 column groups
    //   2: dout         x (h2, 1)  -> dW2, db2      2 column groups
    constexpr int NG0 = (CIN + 1 + 4) / 5, NB = 2 * NG0 + 4 + 2;
    static_assert(NB * 16 <= 256, "phase-2 blocks must fit the workgroup");
    const int blk = tid >> 4, sl = tid & 15;
    const unsigned long long m_b0 = __ballot(sl & 1), m_b1 = __ballot(sl & 2), m_b2 = __ballot(sl & 4), m_b3 = __ballot(sl & 8);
    int aoff[4], boff[5];        // LDS offsets (floats) of this thread's operand rows at pixel 4*sl
    int q_st[2];                 // this lane's two outputs: index into the instance's P gradients, -1 = none
    {
        int fam, og, ig, xrow, arow0, ncol, qbase, qsx, qbias;
        if (blk < 2 * NG0) { fam = 0; og = blk / NG0; ig = blk % NG0; ncol = CIN; xrow = 1 + 4 * kDC; arow0 = 1 + kDC + 4 * og; }
        else if (blk < 2 * NG0 + 4) { fam = 1; og = (blk - 2 * NG0) / 2; ig = (blk - 2 * NG0) % 2; ncol = kDC; xrow = 1 + 3 * kDC; arow0 = 1 + 4 * og; }
        else { fam = 2; og = 0; ig = (blk - 2 * NG0 - 4) % 2; ncol = kDC; xrow = 1 + 2 * kDC; arow0 = 0; }
        const int ycol0 = 5 * ig;
#pragma unroll
        for (int x = 0; x < 4; ++x) aoff[x] = (fam == 2 ? (x == 0 ? 0 : kZeros) : arow0 + x) * kRowPad + 4 * sl;
#pragma unroll
        for (int y = 0; y < 5; ++y) {
            const int i = ycol0 + y;
            boff[y] = (i < ncol ? xrow + i : (i == ncol ? kOnes : kZeros)) * kRowPad + 4 * sl;
        }
        qbase = fam == 0 ? (4 * og) * cin : fam == 1 ? w1 + (4 * og) * kDC : w2;
        qsx = fam == 0 ? cin : fam == 1 ? kDC : 0;
        qbias = fam == 0 ? D::B0 + 4 * og : fam == 1 ? D::B1 + 4 * og : D::B2;
        // after the row reduction every lane holds the 20 totals; lane sl stores elements sl and 16 + sl
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = sl + 16 * e2, x = e / 5, y = e % 5, i = ycol0 + y;
            const bool rowok = fam == 2 ? x == 0 : true;
            int q = -1;
            if (e < 20 && rowok && blk < NB) {
                if (i < ncol) q = qbase + x * qsx + i;
                else if (i == ncol) q = qbias + (fam == 2 ? 0 : x);
            }
            q_st[e2] = q;
        }
    }

    int seen = 0;   // instances of image b in the chunks already scanned
    bool first = true;
    for (int nb = 0; nb < a.N; nb += kChunkN) {
        // ---- the instances of image b in this chunk, every kSlots-th of them is this workgroup's (wave 0) ------
        const int cnt = min(kChunkN, a.N - nb);
        __syncthreads();
        if (tid < kWave) {
            int s0 = seen, mine0 = (seen + slots - 1 - slot) / slots, nm = 0;
            for (int base = 0; base < cnt; base += kWave) {
                const int k = base + tid;
                const bool hit = k < cnt && (int)a.img[nb + k] == b;
                const unsigned long long m = __ballot(hit);
                const int ord = s0 + __popcll(m & ((1ull << tid) - 1ull));     // ordinal among image b's instances
                if (hit && ord % slots == slot) mine[ord / slots - mine0] = k;
                const int tot = __popcll(m);
                nm = (s0 + tot + slots - 1 - slot) / slots - mine0;
                s0 += tot;
            }
            if (tid == 0) n_mine = nm;
            seen = s0;
        }
        __syncthreads();
        seen = __shfl(seen, 0, kWave);   // only wave 0 counted; the other waves do not use `seen` (kept uniform anyway)
        const int M = n_mine;
        if (M > 0 && kKeepTaps) load_taps(nb + __builtin_amdgcn_readfirstlane(mine[0]));
        for (int m = 0; m < M; ++m) {
            const int n = nb + __builtin_amdgcn_readfirstlane(mine[m]);
            // ---- phase 1: thread = pixel ------------------------------------------------------------------
            const float* __restrict__ wts = params + (int64_t)n * P;     // uniform: scalar loads
            v2f in2[CIN / 2], h1[kDC / 2], h2[kDC / 2], dh2[kDC / 2], dh1[kDC / 2];
            float dout = 0.f;
            if constexpr (REL) {
                const float soi = a.soi[a.level[n]];
                in2[0] = v2f{(a.coors[2 * n] - (float)(cc * a.in_stride + a.in_stride / 2)) / soi,
                             (a.coors[2 * n + 1] - (float)(rr * a.in_stride + a.in_stride / 2)) / soi};
            }
#pragma unroll
            for (int k = 0; k < C / 2; ++k) in2[off / 2 + k] = v2f{xf[2 * k], xf[2 * k + 1]};
            if constexpr (kKeepTaps) {
#pragma unroll
                for (int q = 0; q < NT * NT; ++q) dout += gw[q] * g[q];
            } else {
                dout = gather_dy<F>(g_logits + (int64_t)n * OH * OW, rr, cc, f, a.H, a.W);
            }
            if (first) BXI_T(4, blockIdx.x, 1);
            (void)mlp_forward<C, REL>(wts, in2, h1, h2);
            if (!valid) {                                    // pixels outside the map stage zeros
                dout = 0.f;
#pragma unroll
                for (int i = 0; i < CIN / 2; ++i) in2[i] = v2f{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < kDC / 2; ++i) h1[i] = h2[i] = v2f{0.f, 0.f};
            }
            // MLP backward.  `params_again` is the same array under a second name, so that the weights are
            // loaded again (scalar cache hits) rather than kept, and spilled, from the forward pass.
            const float* __restrict__ wts_b = params_again + (int64_t)n * P;
            const v2f dout2 = {dout, dout};
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) {
                const v2f d = dout2 * w2_at(wts_b, w2 + 2 * i);
                dh2[i] = v2f{h2[i].x > 0.f ? d.x : 0.f, h2[i].y > 0.f ? d.y : 0.f};
            }
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) dh1[i] = v2f{0.f, 0.f};
#pragma unroll
            for (int o = 0; o < kDC; ++o) {                  // o outermost: the weights are consumed in address order
                const float so = (o & 1) ? dh2[o / 2].y : dh2[o / 2].x;
                const v2f bo = {so, so};
#pragma unroll
                for (int i = 0; i < kDC / 2; ++i) dh1[i] = pk_fma(bo, w2_at(wts_b, w1 + o * kDC + 2 * i), dh1[i]);
            }
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) dh1[i] = v2f{h1[i].x > 0.f ? dh1[i].x : 0.f, h1[i].y > 0.f ? dh1[i].y : 0.f};
#pragma unroll
            for (int o = 0; o < kDC; ++o) {
                if (o % 4 == 0) {
                    // pin the accumulators here: otherwise the FMAs are sunk below the fences, away from their loads
#pragma unroll
                    for (int kk = 0; kk < C / 2; ++kk) asm volatile("" : "+v"(dfeat[kk]));
                    BXI_SEGMENT();
                }
                const float so = (o & 1) ? dh1[o / 2].y : dh1[o / 2].x;
                const v2f bo = {so, so};
#pragma unroll
                for (int kk = 0; kk < C / 2; ++kk) dfeat[kk] = pk_fma(bo, w2_at(wts_b, o * cin + off + 2 * kk), dfeat[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < C / 2; ++kk) asm volatile("" : "+v"(dfeat[kk]));
            BXI_SEGMENT();
            // stage the operand rows
            rows[0 * kRowPad + tid] = dout;
#pragma unroll
            for (int i = 0; i < kDC; ++i) {
                rows[(1 + i) * kRowPad + tid] = (i & 1) ? dh2[i / 2].y : dh2[i / 2].x;
                rows[(1 + kDC + i) * kRowPad + tid] = (i & 1) ? dh1[i / 2].y : dh1[i / 2].x;
                rows[(1 + 2 * kDC + i) * kRowPad + tid] = (i & 1) ? h2[i / 2].y : h2[i / 2].x;
                rows[(1 + 3 * kDC + i) * kRowPad + tid] = (i & 1) ? h1[i / 2].y : h1[i / 2].x;
            }
#pragma unroll
            for (int i = 0; i < CIN; ++i) rows[(1 + 4 * kDC + i) * kRowPad + tid] = (i & 1) ? in2[i / 2].y : in2[i / 2].x;
            // the next instance's taps travel while phase 2 runs (lds_barrier does not wait for them)
            if (kKeepTaps && m + 1 < M) load_taps(nb + __builtin_amdgcn_readfirstlane(mine[m + 1]));
            lds_barrier();
            if (first) BXI_T(4, blockIdx.x, 2);
            // ---- phase 2: the parameter gradients, dW = dH^T X over the 256 staged pixels -------------------
            // Thread (blk, s): a 4x5 block of outputs over the pixels {4 (16 j + s) .. +3, j < 4}: 9 operand
            // float4 per 80 FMAs (a thread-per-parameter dot product needs 2 per 4, and the LDS pipe then
            // costs 4x the FMAs).  The 16 slices of a block are the 16 lanes of a DPP row.
            if (blk < NB) {
                v2f acc2[4][5];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 5; ++y) acc2[x][y] = v2f{0.f, 0.f};
#pragma unroll 2
                for (int jj = 0; jj < kYR * kYC / 64; ++jj) {
                    float4 av[4], bv[5];
#pragma unroll
                    for (int x = 0; x < 4; ++x) av[x] = *reinterpret_cast<const float4*>(rows + aoff[x] + jj * 64);
#pragma unroll
                    for (int y = 0; y < 5; ++y) bv[y] = *reinterpret_cast<const float4*>(rows + boff[y] + jj * 64);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 5; ++y) {
                            acc2[x][y] = pk_fma(v2f{av[x].x, av[x].y}, v2f{bv[y].x, bv[y].y}, acc2[x][y]);
                            acc2[x][y] = pk_fma(v2f{av[x].z, av[x].w}, v2f{bv[y].z, bv[y].w}, acc2[x][y]);
                        }
                }
                float acc[4][5];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 5; ++y) acc[x][y] = row16_sum(acc2[x][y].x + acc2[x][y].y);
                // lane sl keeps totals sl and 16 + sl: two binary select trees instead of 20 branchy stores from lane 0
                float* dst = param_part + ((int64_t)n * T + tile) * P;
                // (explicit v_cndmask: written as ?: the compiler turns a select of array elements into a
                //  dynamically indexed array in scratch)
                auto flat = [&](int e) { return acc[e / 5][e % 5]; };
                float lo8[8], lo4[4], lo2[2];
#pragma unroll
                for (int i = 0; i < 8; ++i) lo8[i] = lane_select(m_b0, flat(2 * i + 1), flat(2 * i));
#pragma unroll
                for (int i = 0; i < 4; ++i) lo4[i] = lane_select(m_b1, lo8[2 * i + 1], lo8[2 * i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) lo2[i] = lane_select(m_b2, lo4[2 * i + 1], lo4[2 * i]);
                const float v0 = lane_select(m_b3, lo2[1], lo2[0]);
                const float h2a = lane_select(m_b0, flat(17), flat(16)), h2b = lane_select(m_b0, flat(19), flat(18));
                const float v1 = lane_select(m_b1, h2b, h2a);
                if (q_st[0] >= 0) dst[q_st[0]] = v0;
                if (q_st[1] >= 0) dst[q_st[1]] = v1;
            }
            lds_barrier();
            if (first) BXI_T(4, blockIdx.x, 3);
            first = false;
        }
    }
    if (valid) {
        float* o = feat_part + (((int64_t)slot * a.B + b) * C) * HW + (int64_t)r * a.W + c;
#pragma unroll
        for (int k = 0; k < C; ++k) o[k * HW] = (k & 1) ? dfeat[k / 2].y : dfeat[k / 2].x;
    }
    BXI_T(4, blockIdx.x, 4);
}

// ---- backward, second form: four workgroups per CU ---------------------------------------------------------------------
// The same two phases per (tile, instance) with two thirds of the LDS and 128 registers, so that FOUR workgroups share a CU instead of two
// (the kernel is a chain of scalar-load waits, LDS round trips and barriers: a SIMD that holds two waves idles through most of them):
//   * the tile's feature rows are the same for every instance: staged ONCE (C rows), read back per instance instead of held in registers;
//   * the operand rows of an instance go through one 16-row region in TWO passes -- (dh2, h1) -> dW1, db1 ; (dh1 | rel, features) -> dW0, db0 --
//     each a 2 x 5 block of outputs per thread (20 accumulators instead of 40).  LDS rows (256 floats each):
//         [rel 0..1][features 0..C-1][ones] [region 0..15][ones][pad]      the columns of a pass are CONSECUTIVE rows: one offset register
//     (pass A: rel, features, ones, <first region row: an output nobody stores>; pass B: region 8..15 = h1, ones, pad);
//   * dW2 / db2 (nine sums of dout x h2) never touch the operand rows: products in registers, reduced over the 16 lanes of a DPP row, sixteen
//     row partials per workgroup summed in fixed order by nine threads;
//   * the taps of dy are separable (three row offsets / weights, three column offsets / weights) and are loaded at the top of an instance,
//     consumed after the forward pass: nothing of them is live during the contractions.
// Run-to-run identical like the first form (no atomics; fixed summation order), not bit-identical to it (the order differs).
// (Round 6 put the two contractions on the matrix pipe -- v_mfma_f32_16x16x4_f32, a wave contracting its own 64 pixels from a pixel-major,
// odd-stride operand layout, no cross-lane reduction left: 102 registers, every test green, and SLOWER: 29.0 vs 28.0 us at 32 instances, 70.8 vs
// 67.7 at 128.  48 MFMAs per wave and instance at 29 % tile use (M = 8 of 16; N = 9, 16, 3 of 16) occupy the pipe longer than the 120 packed FMAs
// and their DPP sums occupy the VALU, and workgroups that run in step hide little of it.  Commit 535283b; profiles/NOTES.md R6-6.)
template <int C, bool REL, int F>
__global__ __launch_bounds__(256, 4)
void dyn_bwd2_kernel(DynArgs a, const float* __restrict__ params, const float* __restrict__ params_again,
                     const float* __restrict__ g_logits, float* __restrict__ feat_part /*[slots,B,C,H,W]*/,
                     float* __restrict__ param_part /*[N,T,P]*/, int slots) {
    static_assert(F == 1 || F == 2, "the (2F-1)^2 taps of dy are kept per thread");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using D = Dyn<C, REL>;
    constexpr int CIN = D::CIN, off = REL ? 2 : 0;
    constexpr int kOnesA = CIN, RX = CIN + 1, kOnesB = RX + 16;   // rows; RX + 17 = pad
    constexpr int kChunkN = 256;
    float* rows = lds;                                      // [CIN + 1 + 16 + 2][kRowPad]
    __shared__ int mine[kChunkN + 2];
    __shared__ int n_mine;
    __shared__ float red2[16 * 12];                         // dW2 / db2: one partial per DPP row of the workgroup
    const int tiles_x = (a.W + kYC - 1) / kYC, tiles_y = (a.H + kYR - 1) / kYR, T = tiles_x * tiles_y;
    int t = blockIdx.x;
    const int slot = t % slots; t /= slots;
    const int tile = t % T;
    const int b = t / T;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;
    const int lr = tid / kYC, lc = tid % kYC;
    const int r = ty * kYR + lr, c = tx * kYC + lc;
    const bool valid = r < a.H && c < a.W;
    const int rr = min(r, a.H - 1), cc = min(c, a.W - 1);
    const int OH = a.H * F, OW = a.W * F;
    constexpr int P = D::P, w1 = D::W1, w2 = D::W2;
    const int64_t HW = (int64_t)a.H * a.W;
    rows[kOnesA * kRowPad + tid] = 1.f;
    rows[kOnesB * kRowPad + tid] = 1.f;
    rows[(kOnesB + 1) * kRowPad + tid] = 0.f;
    {
        const float* fb = a.feat + (int64_t)b * C * HW;
        const unsigned po = (unsigned)(rr * a.W + cc);
        float xf[C];
#pragma unroll
        for (int k = 0; k < C; ++k) xf[k] = (fb + (int64_t)k * HW)[po];
#pragma unroll
        for (int k = 0; k < C; ++k) rows[(off + k) * kRowPad + tid] = valid ? xf[k] : 0.f;      // pixels outside the map stage zeros
    }
    // d y[r][c] = sum_i wy[i] sum_j wx[j] g[Rs + i][Cs + j] (gather_dy's window, separable)
    constexpr int NT = 2 * F - 1;
    unsigned roff[NT], coff[NT];
    float wy[NT], wx[NT];
    {
        const int half = F / 2;
        const int Rs = rr == 0 ? 0 : F * rr - F + 1 + half, Cs = cc == 0 ? 0 : F * cc - F + 1 + half;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            wy[i] = valid && Rs + i < OH ? upsample_weight(Rs + i, rr, F, a.H) : 0.f;              // (dout = 0 outside the map)
            wx[i] = Cs + i < OW ? upsample_weight(Cs + i, cc, F, a.W) : 0.f;
            roff[i] = (unsigned)(min(Rs + i, OH - 1) * OW);
            coff[i] = (unsigned)min(Cs + i, OW - 1);
        }
    }

    v2f dfeat[C / 2];
#pragma unroll
    for (int k = 0; k < C / 2; ++k) dfeat[k] = v2f{0.f, 0.f};

    // ---- the thread's two phase-2 roles: a 2 x 5 block of dW0|db0 (pass A) and of dW1|db1 (pass B) over the pixels {4 (16 j + sl) .. +3} ----
    constexpr int NG0 = (CIN + 1 + 4) / 5, NBA = 4 * NG0, NBB = 8;
    static_assert(NBA * 16 <= 256, "pass-A blocks must fit the workgroup");
    static_assert(5 * NG0 - 1 <= RX + 17, "the padded columns of pass A stay inside the rows");
    const int blk = tid >> 4, sl = tid & 15;
    const unsigned long long m_b0 = __ballot(sl & 1), m_b1 = __ballot(sl & 2), m_b2 = __ballot(sl & 4), m_b3 = __ballot(sl & 8);
    constexpr unsigned kNone = 0xffffffffu;                 // (unsigned element offsets: the stores address as base + 32-bit offset)
    int aA, bA, aB, bB;
    unsigned qA = kNone, qB = kNone;
    {
        const int ex = sl / 5, ey = sl % 5;                  // the output this lane stores: element sl < 10 of the block
        int og = blk / NG0, ig = blk % NG0;
        aA = (RX + 2 * og) * kRowPad + 4 * sl;
        bA = (5 * ig) * kRowPad + 4 * sl;                    // columns = rows 5 ig .. 5 ig + 4: inputs, ones (row CIN), then nobody's
        if (sl < 10 && blk < NBA) {
            const int i = 5 * ig + ey;
            if (i < CIN) qA = (2 * og + ex) * CIN + i;
            else if (i == CIN) qA = D::B0 + 2 * og + ex;
        }
        og = (blk / 2) & 3; ig = blk % 2;
        aB = (RX + 2 * og) * kRowPad + 4 * sl;
        bB = (RX + kDC + 5 * ig) * kRowPad + 4 * sl;         // columns = h1 0..7, ones, pad
        if (sl < 10 && blk < NBB) {
            const int i = 5 * ig + ey;
            if (i < kDC) qB = w1 + (2 * og + ex) * kDC + i;
            else if (i == kDC) qB = D::B1 + 2 * og + ex;
        }
    }
    // one pass: acc[x][y] = sum over the 256 staged pixels of row (a0 + x) x row (b0 + y); lane sl < 10 of a block stores element sl
    auto contract = [&](int a0, int b0, unsigned q, float* dst) {
        v2f acc2[2][5];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) acc2[x][y] = v2f{0.f, 0.f};
#pragma unroll 1
        for (int jj = 0; jj < kYR * kYC / 64; ++jj) {
            float4 av[2], bv[5];
#pragma unroll
            for (int x = 0; x < 2; ++x) av[x] = *reinterpret_cast<const float4*>(rows + a0 + x * kRowPad + jj * 64);
#pragma unroll
            for (int y = 0; y < 5; ++y) bv[y] = *reinterpret_cast<const float4*>(rows + b0 + y * kRowPad + jj * 64);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 5; ++y) {
                    acc2[x][y] = pk_fma(v2f{av[x].x, av[x].y}, v2f{bv[y].x, bv[y].y}, acc2[x][y]);
                    acc2[x][y] = pk_fma(v2f{av[x].z, av[x].w}, v2f{bv[y].z, bv[y].w}, acc2[x][y]);
                }
        }
        float v[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) v[e] = row16_sum(acc2[e / 5][e % 5].x + acc2[e / 5][e % 5].y);
        float l1[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) l1[i] = lane_select(m_b0, v[2 * i + 1], v[2 * i]);
        const float l2a = lane_select(m_b1, l1[1], l1[0]), l2b = lane_select(m_b1, l1[3], l1[2]);
        const float l3 = lane_select(m_b2, l2b, l2a);
        const float out = lane_select(m_b3, l1[4], l3);
        // (base in scalar registers + a 32-bit byte offset: left to the compiler the address becomes a 64-bit register pair per store, kept -- and
        //  spilled -- across the whole instance loop)
        if (q != kNone) asm volatile("global_store_dword %0, %1, %2" ::"v"(q * 4u), "v"(out), "s"(dst) : "memory");
    };

    int seen = 0;
    for (int nb = 0; nb < a.N; nb += kChunkN) {
        const int cnt = min(kChunkN, a.N - nb);
        __syncthreads();
        if (tid < kWave) {
            int s0 = seen, mine0 = (seen + slots - 1 - slot) / slots, nm = 0;
            for (int base = 0; base < cnt; base += kWave) {
                const int k = base + tid;
                const bool hit = k < cnt && (int)a.img[nb + k] == b;
                const unsigned long long m = __ballot(hit);
                const int ord = s0 + __popcll(m & ((1ull << tid) - 1ull));
                if (hit && ord % slots == slot) mine[ord / slots - mine0] = k;
                const int tot = __popcll(m);
                nm = (s0 + tot + slots - 1 - slot) / slots - mine0;
                s0 += tot;
            }
            if (tid == 0) n_mine = nm;
            seen = s0;
        }
        __syncthreads();
        seen = __shfl(seen, 0, kWave);
        const int M = n_mine;
        for (int m = 0; m < M; ++m) {
            const int n = nb + __builtin_amdgcn_readfirstlane(mine[m]);
            float* dst = param_part + ((int64_t)n * T + tile) * P;
            // ---- phase 1: thread = pixel ------------------------------------------------------------------
            float g[NT * NT];
            {
                const float* gz = g_logits + (int64_t)n * OH * OW;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int jx = 0; jx < NT; ++jx) g[i * NT + jx] = gz[roff[i] + coff[jx]];
            }
            const float* __restrict__ wts = params + (int64_t)n * P;
            v2f in2[CIN / 2], h1[kDC / 2], h2[kDC / 2], dh2[kDC / 2], dh1[kDC / 2];
            if constexpr (REL) {
                const float soi = a.soi[a.level[n]];
                in2[0] = v2f{(a.coors[2 * n] - (float)(cc * a.in_stride + a.in_stride / 2)) / soi,
                             (a.coors[2 * n + 1] - (float)(rr * a.in_stride + a.in_stride / 2)) / soi};
                if (!valid) in2[0] = v2f{0.f, 0.f};
                rows[0 * kRowPad + tid] = in2[0].x;          // (read in pass A only: the previous instance's pass A is behind a barrier)
                rows[1 * kRowPad + tid] = in2[0].y;
            }
#pragma unroll
            for (int k = 0; k < C / 2; ++k) in2[off / 2 + k] = v2f{rows[(off + 2 * k) * kRowPad + tid], rows[(off + 2 * k + 1) * kRowPad + tid]};
            (void)mlp_forward<C, REL>(wts, in2, h1, h2);
            if (!valid) {
#pragma unroll
                for (int i = 0; i < kDC / 2; ++i) h1[i] = h2[i] = v2f{0.f, 0.f};
            }
            float dout = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float rowsum = 0.f;
#pragma unroll
                for (int jx = 0; jx < NT; ++jx) rowsum += wx[jx] * g[i * NT + jx];
                dout += wy[i] * rowsum;
            }
            const float* __restrict__ wts_b = params_again + (int64_t)n * P;
            const v2f dout2 = {dout, dout};
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) {
                const v2f d = dout2 * w2_at(wts_b, w2 + 2 * i);
                dh2[i] = v2f{h2[i].x > 0.f ? d.x : 0.f, h2[i].y > 0.f ? d.y : 0.f};
            }
            // dW2 | db2: nine products, summed over the DPP row here, over the workgroup's sixteen rows after the barrier
            {
                float p9[9];
#pragma unroll
                for (int i = 0; i < kDC / 2; ++i) { const v2f pr = dout2 * h2[i]; p9[2 * i] = pr.x; p9[2 * i + 1] = pr.y; }
                p9[8] = dout;
#pragma unroll
                for (int i = 0; i < 9; ++i) p9[i] = row16_sum(p9[i]);
                if (sl == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) red2[blk * 12 + i] = p9[i];
                }
            }
            // pass B operands: dh2, h1
#pragma unroll
            for (int i = 0; i < kDC; ++i) {
                rows[(RX + i) * kRowPad + tid] = (i & 1) ? dh2[i / 2].y : dh2[i / 2].x;
                rows[(RX + kDC + i) * kRowPad + tid] = (i & 1) ? h1[i / 2].y : h1[i / 2].x;
            }
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) dh1[i] = v2f{0.f, 0.f};
#pragma unroll
            for (int o = 0; o < kDC; ++o) {
                const float so = (o & 1) ? dh2[o / 2].y : dh2[o / 2].x;
                const v2f bo = {so, so};
#pragma unroll
                for (int i = 0; i < kDC / 2; ++i) dh1[i] = pk_fma(bo, w2_at(wts_b, w1 + o * kDC + 2 * i), dh1[i]);
            }
#pragma unroll
            for (int i = 0; i < kDC / 2; ++i) dh1[i] = v2f{h1[i].x > 0.f ? dh1[i].x : 0.f, h1[i].y > 0.f ? dh1[i].y : 0.f};
#pragma unroll
            for (int o = 0; o < kDC; ++o) {
                if (o % 4 == 0) {
#pragma unroll
                    for (int kk = 0; kk < C / 2; ++kk) asm volatile("" : "+v"(dfeat[kk]));
                    BXI_SEGMENT();
                }
                const float so = (o & 1) ? dh1[o / 2].y : dh1[o / 2].x;
                const v2f bo = {so, so};
#pragma unroll
                for (int kk = 0; kk < C / 2; ++kk) dfeat[kk] = pk_fma(bo, w2_at(wts_b, o * CIN + off + 2 * kk), dfeat[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < C / 2; ++kk) asm volatile("" : "+v"(dfeat[kk]));
            BXI_SEGMENT();
            lds_barrier();
            // ---- pass B: dW1 | db1 (blocks 0..7), and the nine threads that finish dW2 | db2 ------------------------------
            if (blk < NBB) contract(aB, bB, qB, dst);
            if (tid >= 192 && tid < 192 + 9) {               // (a wave that has no pass-B block)
                unsigned i = (unsigned)(tid - 192);
                asm volatile("" : "+v"(i));                  // (one address register + sixteen immediate offsets, not sixteen hoisted addresses)
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) sum += red2[u * 12 + i];
                const unsigned qq = i < (unsigned)kDC ? w2 + i : (unsigned)D::B2;
                asm volatile("global_store_dword %0, %1, %2" ::"v"(qq * 4u), "v"(sum), "s"(dst) : "memory");
            }
            lds_barrier();
            // pass A operands: dh1 (rel and the features are in their rows already)
#pragma unroll
            for (int i = 0; i < kDC; ++i) rows[(RX + i) * kRowPad + tid] = (i & 1) ? dh1[i / 2].y : dh1[i / 2].x;
            lds_barrier();
            if (blk < NBA) contract(aA, bA, qA, dst);
            lds_barrier();
        }
    }
    {
        // (the pixel's offset is derived again here: computed at the top it is a register pair held -- spilled -- through the instance loop)
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int r2 = ty * kYR + tid2 / kYC, c2 = tx * kYC + tid2 % kYC;
        if (r2 < a.H && c2 < a.W) {
            float* o = feat_part + (((int64_t)slot * a.B + b) * C) * HW + (unsigned)(r2 * a.W + c2);
#pragma unroll
            for (int k = 0; k < C; ++k) o[k * HW] = (k & 1) ? dfeat[k / 2].y : dfeat[k / 2].x;
        }
    }
}

__global__ __launch_bounds__(256) void dyn_reduce_kernel(const float* __restrict__ feat_part, int64_t feat_elems,
                                                         float* __restrict__ g_feat, const float* __restrict__ param_part,
                                                         int N, int T, int P, float* __restrict__ g_params, int slots) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < feat_elems) {
        float v[kSlots];
#pragma unroll
        for (int s = 0; s < kSlots; ++s) v[s] = s < slots ? feat_part[s * feat_elems + i] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) acc += v[s];                  // fixed order (absent slots add 0)
        g_feat[i] = acc;
    }
    const int64_t j = i - ((feat_elems + 255) / 256) * 256;
    if (j >= 0 && j < (int64_t)N * P) {
        const int n = (int)(j / P), q = (int)(j % P);
        const float* src = param_part + (int64_t)n * T * P + q;
        float acc = 0.f;
        int t = 0;
        for (; t + 8 <= T; t += 8) {                         // 8 loads in flight, summed in tile order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(t + u) * P];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; t < T; ++t) acc += src[(int64_t)t * P];
        g_params[j] = acc;
    }
}

int fill_dyn(const float* feat, int B, int C, int H, int W, const float* params, int N, const float* coors,
                    const int64_t* level, const int64_t* img, const float* soi, int n_levels, int in_stride, int factor,
                    int disable_rel, DynArgs& a) {
    if (B <= 0 || H <= 0 || W <= 0 || N < 0 || n_levels <= 0) return BXI_ERR_BAD_SHAPE;
    if (in_stride < 1 || factor < 1) return BXI_ERR_BAD_ARGUMENT;
    if (C != 8 && C != 16) return BXI_ERR_UNSUPPORTED;
    if (N > 0 && (!feat || !params || !coors || !level || !img || !soi)) return BXI_ERR_NULL_POINTER;
    if (!fits_i32((int64_t)N * H * W * factor * factor)) return BXI_ERR_BAD_SHAPE;
    a.feat = feat; a.params = params; a.coors = coors; a.level = level; a.img = img; a.soi = soi;
    a.B = B; a.H = H; a.W = W; a.N = N; a.n_levels = n_levels; a.in_stride = in_stride; a.factor = factor;
    a.rel = disable_rel ? 0 : 1;
    return BXI_OK;
}

static inline int dyn_tiles(int H, int W) { return ((H + kYR - 1) / kYR) * ((W + kYC - 1) / kYC); }
static inline int dyn_params(int C, int rel) { return (C + (rel ? 2 : 0)) * kDC + kDC * kDC + kDC + 2 * kDC + 1; }

// the kernels are instantiated for C in {8,16} x rel x factor in {1,2,4,other}
#define BXI_DYN_DISPATCH_F(KC_, KR_, F_, ...)                                   \
    do {                                                                       \
        constexpr int KC = KC_; constexpr bool KR = KR_;                       \
        if ((F_) == 2) { constexpr int KF = 2; __VA_ARGS__; }                         \
        else if ((F_) == 1) { constexpr int KF = 1; __VA_ARGS__; }                    \
        else if ((F_) == 4) { constexpr int KF = 4; __VA_ARGS__; }                    \
        else { constexpr int KF = 0; __VA_ARGS__; }                                   \
    } while (0)
#define BXI_DYN_DISPATCH(C_, REL_, F_, ...)                                    \
    do {                                                                       \
        if ((C_) == 16 && (REL_)) BXI_DYN_DISPATCH_F(16, true, F_, __VA_ARGS__);      \
        else if ((C_) == 16) BXI_DYN_DISPATCH_F(16, false, F_, __VA_ARGS__);          \
        else if ((REL_)) BXI_DYN_DISPATCH_F(8, true, F_, __VA_ARGS__);                \
        else BXI_DYN_DISPATCH_F(8, false, F_, __VA_ARGS__);                           \
    } while (0)

}  // namespace bxi

extern "C" {

int bxi_dynamic_mask_forward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N,
                                 const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                 const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                 int disable_rel_coors, float* logits, void* stream) {
    bxi::DynArgs a;
    int rc = bxi::fill_dyn(feat, B, C, H, W, params, N, coors, level_inds, img_inds, sizes_of_interest, n_levels, in_stride,
                           factor, disable_rel_coors, a);
    if (rc != BXI_OK) return rc;
    if (N == 0) return BXI_OK;
    if (!logits) return BXI_ERR_NULL_POINTER;
    hipStream_t s = bxi::as_stream(stream);
    const int fwd_rows = (factor == 1 || factor == 2) ? bxi::kFwdR : bxi::kYR;          // as dyn_fwd_kernel chooses
    const unsigned grid = (unsigned)(N * ((H + fwd_rows - 1) / fwd_rows) * ((W + bxi::kFwdC - 1) / bxi::kFwdC));
    BXI_DYN_DISPATCH(C, a.rel, factor, BXI_LAUNCH("dyn_fwd", s, (bxi::dyn_fwd_kernel<KC, KR, KF>), dim3(grid), dim3(256), 0, s, a, params, logits));
    return bxi::check_launch();
}

size_t bxi_dynamic_mask_backward_workspace_bytes(int B, int C, int H, int W, int N, int disable_rel_coors) {
    if (B <= 0 || H <= 0 || W <= 0 || N < 0 || (C != 8 && C != 16)) return 0;
    const size_t feat = sizeof(float) * (size_t)bxi::kSlots * B * C * H * W;
    const size_t par = sizeof(float) * (size_t)(N > 0 ? N : 1) * bxi::dyn_tiles(H, W) * bxi::dyn_params(C, !disable_rel_coors);
    return (feat + 255) / 256 * 256 + (par + 255) / 256 * 256;
}

int bxi_dynamic_mask_backward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N,
                                  const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                  const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                  int disable_rel_coors, const float* g_logits, float* g_feat, float* g_params,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    bxi::DynArgs a;
    int rc = bxi::fill_dyn(feat, B, C, H, W, params, N, coors, level_inds, img_inds, sizes_of_interest, n_levels, in_stride,
                           factor, disable_rel_coors, a);
    if (rc != BXI_OK) return rc;
    if (!g_feat || (N > 0 && (!g_logits || !g_params))) return BXI_ERR_NULL_POINTER;
    const size_t need = bxi_dynamic_mask_backward_workspace_bytes(B, C, H, W, N, disable_rel_coors);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    hipStream_t s = bxi::as_stream(stream);
    const int T = bxi::dyn_tiles(H, W), P = bxi::dyn_params(C, a.rel);
    const int64_t feat_elems = (int64_t)B * C * H * W;
    float* feat_part = (float*)workspace;
    float* param_part = (float*)((char*)workspace + (sizeof(float) * (size_t)bxi::kSlots * feat_elems + 255) / 256 * 256);
    const int cin = C + (a.rel ? 2 : 0);
    const size_t lds = sizeof(float) * ((size_t)(1 + 4 * bxi::kDC + cin + 2) * bxi::kRowPad);
    // Workgroups per (image, tile): each walks every slots-th instance of its image.
#ifdef BXI_DEV
    static const int form = getenv("BXI_DYN_BWD_FORM") ? atoi(getenv("BXI_DYN_BWD_FORM")) : 2;      // A/B of the two forms (developer build only)
    static const int env_slots = getenv("BXI_DYN_BWD_SLOTS") ? atoi(getenv("BXI_DYN_BWD_SLOTS")) : 0;
#else
    constexpr int form = 2, env_slots = 0;
#endif
    int slots;
    if (form == 2 && (factor == 1 || factor == 2)) {
        // dyn_bwd2_kernel, four workgroups per CU (1024 slots on 256 CUs): as many slots per (image, tile) as keep the launch resident at
        // once -- 2 x 52 x 8 = 832 workgroups at 2 x 100 x 128 --, all of them when there are many instances.  Measured against dyn_bwd_kernel
        // (rocprofv3, same box, 2 x 16 x 100 x 128 -> 200 x 256): 32 instances 28.2-28.8 -> 27.2-27.8 us (the reduction over 8 instead of 4
        // feature partials: 5.1 -> 5.6 us), 128 instances 78.6-79.4 -> 70.0-70.8 us; 4 / 6 / 7 slots at 32 instances: 31.3 / 29.8 / 29.9 us.
        const int cap = 4 * bxi::device_cus() / (B * T);
        slots = N > 16 * B || cap >= bxi::kSlots ? bxi::kSlots : (cap < 1 ? 1 : cap);
        if (env_slots > 0) slots = env_slots;
        const size_t lds2 = sizeof(float) * ((size_t)(cin + 1 + 16 + 2) * bxi::kRowPad);
        const unsigned grid2 = (unsigned)(B * T * slots);
#define BXI_DYN2(KC, KR, KF) BXI_LAUNCH("dyn_bwd", s, (bxi::dyn_bwd2_kernel<KC, KR, KF>), dim3(grid2), dim3(256), lds2, s, a, params, params, g_logits, feat_part, param_part, slots)
        if (factor == 2) {
            if (C == 16 && a.rel) BXI_DYN2(16, true, 2); else if (C == 16) BXI_DYN2(16, false, 2); else if (a.rel) BXI_DYN2(8, true, 2); else BXI_DYN2(8, false, 2);
        } else {
            if (C == 16 && a.rel) BXI_DYN2(16, true, 1); else if (C == 16) BXI_DYN2(16, false, 1); else if (a.rel) BXI_DYN2(8, true, 1); else BXI_DYN2(8, false, 1);
        }
#undef BXI_DYN2
    } else {
        // dyn_bwd_kernel (factor 4 and the run-time factor: their (2f-1)^2 tap windows do not fit dyn_bwd2_kernel's registers), two workgroups per
        // CU.  Few instances per image: 4 slots, so that the launch is resident at once (2 x 52 x 4 = 416 workgroups on 512 slots); many: 8.
        slots = N <= 16 * B ? 4 : bxi::kSlots;
        if (env_slots > 0) slots = env_slots;
        const unsigned grid = (unsigned)(B * T * slots);
#define BXI_DYN1(KC, KR, KF) BXI_LAUNCH("dyn_bwd", s, (bxi::dyn_bwd_kernel<KC, KR, KF>), dim3(grid), dim3(256), lds, s, a, params, params, g_logits, feat_part, param_part, slots)
#define BXI_DYN1_F(KF) do { if (C == 16 && a.rel) BXI_DYN1(16, true, KF); else if (C == 16) BXI_DYN1(16, false, KF); else if (a.rel) BXI_DYN1(8, true, KF); else BXI_DYN1(8, false, KF); } while (0)
        if (factor == 4) BXI_DYN1_F(4);
#ifdef BXI_DEV
        else if (factor == 2) BXI_DYN1_F(2);
        else if (factor == 1) BXI_DYN1_F(1);
#endif
        else BXI_DYN1_F(0);
#undef BXI_DYN1_F
#undef BXI_DYN1
    }
    rc = bxi::check_launch();
    if (rc != BXI_OK) return rc;
    const int64_t nb = (feat_elems + 255) / 256 + ((int64_t)N * P + 255) / 256;
    BXI_LAUNCH("dyn_reduce", s, bxi::dyn_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, feat_part, feat_elems, g_feat,
               param_part, N, T, P, g_params, slots);
    return bxi::check_launch();
}

#ifdef BXI_TRACE
int bxi_debug_set_trace_dyn(long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_trace), &buf, sizeof(buf)); }
#endif

}  // extern "C"
