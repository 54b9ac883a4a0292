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
// dyn_fwd_kernel   grid = N x tiles(8x32 of y).  233 weights of the instance in LDS (broadcast reads),
//                  y tile + 1 halo in LDS, upsample from LDS, one coalesced store per output row.
//                  Reads the feature tile (L2-resident, 4*C B per y pixel per instance), writes
//                  4*f^2 B per y pixel per instance: write bound.
// dyn_bwd_kernel   grid = B x tiles x kSlots.  A workgroup owns one 8x32 tile of one image and every
//                  kSlots-th instance of that image:
//                    phase 1 (thread = pixel): dy by the transposed interpolation (gather), forward
//                       recomputed, MLP backward; accumulates d feat over its instances in registers
//                       and stages 51 operand rows of 256 pixels in LDS;
//                    phase 2 (thread = parameter, 233 of 256): the parameter gradients are 233 dot
//                       products of two staged rows (dW = dH^T X, a [8..18] x 256 contraction too thin
//                       for MFMA) -> one partial per (instance, tile).
//                  No atomics: partials are reduced in fixed order by dyn_reduce_kernel.
// dyn_reduce_kernel  g_params[n,q] = sum over tiles ; g_feat[b,c,p] = sum over slots.
#include "common.hpp"

namespace bxi {

constexpr int kDC = 8;            // dynamic_channels (configs/boxinst: 8)
constexpr int kYR = 8, kYC = 32;  // y tile (pixels at in_stride resolution) per workgroup
constexpr int kSlots = 8;         // instance slots per (image, tile) in the backward
constexpr int kRowPad = kYR * kYC + 4;   // LDS row stride of the staged operand rows (conflict-free b128)

struct DynArgs {
    const float* feat;        // [B,C,H,W]
    const float* params;      // [N,P]  P = (C+2)*8 + 64 + 8 + 8 + 8 + 1
    const float* coors;       // [N,2]  (x,y) of the generating location, image pixels
    const int64_t* level;     // [N]
    const int64_t* img;       // [N]
    const float* soi;         // [n_levels]
    int B, H, W, N, n_levels, in_stride, factor, rel;   // rel = !disable_rel_coors
};

template <int C> struct DynLayout {
    static constexpr int CIN = C + 2;             // with relative coordinates (the first two channels)
    static constexpr int W0 = 0;                  // [8][CIN]
    static constexpr int W1 = CIN * kDC;          // [8][8]
    static constexpr int W2 = W1 + kDC * kDC;     // [1][8]
    static constexpr int B0 = W2 + kDC;
    static constexpr int B1 = B0 + kDC;
    static constexpr int B2 = B1 + kDC;
    static constexpr int P = B2 + 1;
};

// one pixel through the three dynamic layers.  `wts` = the instance's parameters in LDS.
// With rel == 0 the first layer has C inputs ([8][C] weights) and in[0..1] are unused.
template <int C>
__device__ __forceinline__ float mlp_forward(const float* wts, const float (&in)[C + 2], int rel, float (&h1)[kDC],
                                             float (&h2)[kDC]) {
    const int cin = rel ? C + 2 : C, off = rel ? 0 : 2;
    const int w1 = cin * kDC, w2 = w1 + kDC * kDC, b0 = w2 + kDC, b1 = b0 + kDC, b2 = b1 + kDC;
#pragma unroll
    for (int o = 0; o < kDC; ++o) {
        float acc = wts[b0 + o];
        for (int i = 0; i < cin; ++i) acc += wts[o * cin + i] * in[i + off];
        h1[o] = fmaxf(acc, 0.f);
    }
#pragma unroll
    for (int o = 0; o < kDC; ++o) {
        float acc = wts[b1 + o];
#pragma unroll
        for (int i = 0; i < kDC; ++i) acc += wts[w1 + o * kDC + i] * h1[i];
        h2[o] = fmaxf(acc, 0.f);
    }
    float y = wts[b2];
#pragma unroll
    for (int i = 0; i < kDC; ++i) y += wts[w2 + i] * h2[i];
    return y;
}

template <int C>
__device__ __forceinline__ void load_inputs(const DynArgs& a, int n, int b, int r, int c, float cx, float cy, float inv_soi,
                                            float (&in)[C + 2]) {
    const int64_t HW = (int64_t)a.H * a.W;
    const float* f = a.feat + (int64_t)b * C * HW + (int64_t)r * a.W + c;
    // locations = arange(0, W*stride, stride) + stride // 2   (:1143-1150)
    in[0] = (cx - (float)(c * a.in_stride + a.in_stride / 2)) * inv_soi;
    in[1] = (cy - (float)(r * a.in_stride + a.in_stride / 2)) * inv_soi;
#pragma unroll
    for (int k = 0; k < C; ++k) in[2 + k] = f[k * HW];
}

// ---- forward ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void dyn_fwd_kernel(DynArgs a, float* __restrict__ logits) {
    __shared__ float wts[DynLayout<C>::P];
    __shared__ float ytile[(kYR + 2) * (kYC + 2)];
    const int tiles_x = (a.W + kYC - 1) / kYC, tiles_y = (a.H + kYR - 1) / kYR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int tid = threadIdx.x;
    const int P = a.rel ? DynLayout<C>::P : DynLayout<C>::P - 2 * kDC;
    for (int i = tid; i < P; i += 256) wts[i] = a.params[(int64_t)n * P + i];
    const int b = (int)a.img[n];
    const float cx = a.coors[2 * n], cy = a.coors[2 * n + 1];
    const float soi = a.soi[a.level[n]];
    __syncthreads();
    // y on the tile plus one pixel of halo on every side (rows r0-1 .. r0+kYR)
    const int r0 = ty * kYR, c0 = tx * kYC;
    for (int i = tid; i < (kYR + 2) * (kYC + 2); i += 256) {
        const int r = r0 - 1 + i / (kYC + 2), c = c0 - 1 + i % (kYC + 2);
        float y = 0.f;
        if (r >= 0 && r < a.H && c >= 0 && c < a.W) {
            float in[C + 2], h1[kDC], h2[kDC];
            // the reference divides: rel_coors / soi (:1153)
            load_inputs<C>(a, n, b, r, c, cx, cy, 1.f, in);
            in[0] = in[0] / soi; in[1] = in[1] / soi;
            y = mlp_forward<C>(wts, in, a.rel, h1, h2);
        }
        ytile[i] = y;
    }
    __syncthreads();
    // aligned_bilinear (:146-167): z[R][Cc] = I[max(R - f/2, 0)][max(Cc - f/2, 0)],
    // I[i][j] = bilinear sample of (y padded by one replicated row/column) at (i/f, j/f)
    const int f = a.factor, OH = a.H * f, OW = a.W * f, half = f / 2;
    const int R0 = r0 * f, C0 = c0 * f;
    float* out = logits + (int64_t)n * OH * OW;
    for (int i = tid; i < kYR * f * kYC * f; i += 256) {
        const int R = R0 + i / (kYC * f), Cc = C0 + i % (kYC * f);
        if (R >= OH || Cc >= OW) continue;
        const int ii = max(R - half, 0), jj = max(Cc - half, 0);
        const int yi = ii / f, xj = jj / f;
        const float fy = (float)(ii % f) / (float)f, fx = (float)(jj % f) / (float)f;
        const int yi1 = min(yi + 1, a.H - 1), xj1 = min(xj + 1, a.W - 1);   // replicate pad (:156)
        auto Y = [&](int r, int c) { return ytile[(r - r0 + 1) * (kYC + 2) + (c - c0 + 1)]; };
        const float top = (1.f - fx) * Y(yi, xj) + fx * Y(yi, xj1);
        const float bot = (1.f - fx) * Y(yi1, xj) + fx * Y(yi1, xj1);
        out[(int64_t)R * OW + Cc] = (1.f - fy) * top + fy * bot;
    }
}

// ---- backward --------------------------------------------------------------------------------------
// d y[r][c] = sum over the output pixels that sampled y[r][c], with their interpolation weights
__device__ __forceinline__ float upsample_weight(int Rout, int r, int f, int Hin) {
    const int ii = max(Rout - f / 2, 0);
    const int yi = ii / f;
    const float fy = (float)(ii % f) / (float)f;
    float wgt = 0.f;
    if (yi == r) wgt += 1.f - fy;
    if (min(yi + 1, Hin - 1) == r) wgt += fy;
    return wgt;
}

template <int C>
__global__ __launch_bounds__(256) void dyn_bwd_kernel(DynArgs a, const float* __restrict__ g_logits,
                                                      float* __restrict__ feat_part /*[kSlots,B,C,H,W]*/,
                                                      float* __restrict__ param_part /*[N,T,P]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CIN = C + 2;
    constexpr int NROW = 1 + kDC + kDC + kDC + kDC + CIN;   // dout, dh2, dh1, h2, h1, in
    float* rows = lds;                                      // [NROW][kRowPad]
    float* wts = lds + NROW * kRowPad;                      // [P]
    __shared__ int imgs[1024];
    const int tiles_x = (a.W + kYC - 1) / kYC, tiles_y = (a.H + kYR - 1) / kYR, T = tiles_x * tiles_y;
    int t = blockIdx.x;
    const int slot = t % kSlots; t /= kSlots;
    const int tile = t % T;
    const int b = t / T;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x;
    const int lr = tid / kYC, lc = tid % kYC;
    const int r = ty * kYR + lr, c = tx * kYC + lc;
    const bool valid = r < a.H && c < a.W;
    const int f = a.factor, OH = a.H * f, OW = a.W * f;
    const int P = a.rel ? DynLayout<C>::P : DynLayout<C>::P - 2 * kDC;
    const int cin = a.rel ? CIN : C, off = a.rel ? 0 : 2;
    const int w1 = cin * kDC, w2 = w1 + kDC * kDC;
    const int64_t HW = (int64_t)a.H * a.W;

    float dfeat[C];
#pragma unroll
    for (int k = 0; k < C; ++k) dfeat[k] = 0.f;

    int seen = 0;   // instances of image b met so far (uniform)
    for (int nb = 0; nb < a.N; nb += 1024) {
        __syncthreads();
        for (int i = tid; i < min(1024, a.N - nb); i += 256) imgs[i] = (int)a.img[nb + i];
        __syncthreads();
        for (int k = 0; k < min(1024, a.N - nb); ++k) {
            if (imgs[k] != b) continue;                      // uniform
            const bool mine = (seen % kSlots) == slot;
            ++seen;
            if (!mine) continue;
            const int n = nb + k;
            // ---- phase 1: thread = pixel --------------------------------------------------------------
            for (int i = tid; i < P; i += 256) wts[i] = a.params[(int64_t)n * P + i];
            __syncthreads();
            float in[CIN], h1[kDC], h2[kDC], dh2[kDC], dh1[kDC], dout = 0.f;
#pragma unroll
            for (int i = 0; i < CIN; ++i) in[i] = 0.f;
#pragma unroll
            for (int i = 0; i < kDC; ++i) { h1[i] = h2[i] = dh1[i] = dh2[i] = 0.f; }
            if (valid) {
                const float soi = a.soi[a.level[n]];
                load_inputs<C>(a, n, b, r, c, a.coors[2 * n], a.coors[2 * n + 1], 1.f, in);
                in[0] = in[0] / soi; in[1] = in[1] / soi;
                (void)mlp_forward<C>(wts, in, a.rel, h1, h2);
                // transposed aligned_bilinear: candidates are the output rows/cols around f*r, f*c
                const float* gz = g_logits + (int64_t)n * OH * OW;
                const int Ra = max(f * (r - 1) + f / 2, 0), Rb = min(f * (r + 1) + f / 2, OH);
                const int Ca = max(f * (c - 1) + f / 2, 0), Cb = min(f * (c + 1) + f / 2, OW);
                const int Ra0 = r == 0 ? 0 : Ra, Ca0 = c == 0 ? 0 : Ca;   // rows/cols clamped to source index 0
                for (int R = Ra0; R < Rb; ++R) {
                    const float wy = upsample_weight(R, r, f, a.H);
                    if (wy == 0.f) continue;
                    for (int Cc = Ca0; Cc < Cb; ++Cc) {
                        const float wx = upsample_weight(Cc, c, f, a.W);
                        if (wx != 0.f) dout += wy * wx * gz[(int64_t)R * OW + Cc];
                    }
                }
                // MLP backward
#pragma unroll
                for (int i = 0; i < kDC; ++i) dh2[i] = h2[i] > 0.f ? dout * wts[w2 + i] : 0.f;
#pragma unroll
                for (int i = 0; i < kDC; ++i) {
                    float acc = 0.f;
#pragma unroll
                    for (int o = 0; o < kDC; ++o) acc += dh2[o] * wts[w1 + o * kDC + i];
                    dh1[i] = h1[i] > 0.f ? acc : 0.f;
                }
#pragma unroll
                for (int kk = 0; kk < C; ++kk) {
                    float acc = 0.f;
#pragma unroll
                    for (int o = 0; o < kDC; ++o) acc += dh1[o] * wts[o * cin + (kk + 2 - off)];
                    dfeat[kk] += acc;
                }
            }
            // stage the operand rows (zeros for pixels outside the map)
            rows[0 * kRowPad + tid] = dout;
#pragma unroll
            for (int i = 0; i < kDC; ++i) {
                rows[(1 + i) * kRowPad + tid] = dh2[i];
                rows[(1 + kDC + i) * kRowPad + tid] = dh1[i];
                rows[(1 + 2 * kDC + i) * kRowPad + tid] = h2[i];
                rows[(1 + 3 * kDC + i) * kRowPad + tid] = h1[i];
            }
#pragma unroll
            for (int i = 0; i < CIN; ++i) rows[(1 + 4 * kDC + i) * kRowPad + tid] = in[i];
            __syncthreads();
            // ---- phase 2: thread = parameter q: a dot product of two staged rows (or a row sum) -----------
            if (tid < P) {
                int ra, rb = -1;     // rb < 0: bias -> plain row sum
                const int q = tid;
                if (q < w1) { ra = 1 + kDC + q / cin; rb = 1 + 4 * kDC + (q % cin) + off; }                 // dW0[o][i] = dh1[o] . in[i]
                else if (q < w2) { ra = 1 + (q - w1) / kDC; rb = 1 + 3 * kDC + (q - w1) % kDC; }         // dW1[o][i] = dh2[o] . h1[i]
                else if (q < w2 + kDC) { ra = 0; rb = 1 + 2 * kDC + (q - w2); }                            // dW2[i]    = dout . h2[i]
                else if (q < w2 + 2 * kDC) ra = 1 + kDC + (q - w2 - kDC);                                   // db0[o] = sum dh1[o]
                else if (q < w2 + 3 * kDC) ra = 1 + (q - w2 - 2 * kDC);                                     // db1[o] = sum dh2[o]
                else ra = 0;                                                                                // db2 = sum dout
                const float4* A = reinterpret_cast<const float4*>(rows + ra * kRowPad);
                float acc = 0.f;
                if (rb >= 0) {
                    const float4* Bv = reinterpret_cast<const float4*>(rows + rb * kRowPad);
#pragma unroll 8
                    for (int j = 0; j < kYR * kYC / 4; ++j) {
                        const float4 x = A[j], y = Bv[j];
                        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                    }
                } else {
#pragma unroll 8
                    for (int j = 0; j < kYR * kYC / 4; ++j) { const float4 x = A[j]; acc += (x.x + x.y) + (x.z + x.w); }
                }
                param_part[((int64_t)n * T + tile) * P + q] = acc;
            }
            __syncthreads();
        }
    }
    if (valid) {
        float* o = feat_part + (((int64_t)slot * a.B + b) * C) * HW + (int64_t)r * a.W + c;
#pragma unroll
        for (int k = 0; k < C; ++k) o[k * HW] = dfeat[k];
    }
}

__global__ __launch_bounds__(256) void dyn_reduce_kernel(const float* __restrict__ feat_part, int64_t feat_elems,
                                                         float* __restrict__ g_feat, const float* __restrict__ param_part,
                                                         int N, int T, int P, float* __restrict__ g_params) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < feat_elems) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) acc += feat_part[s * feat_elems + i];
        g_feat[i] = acc;
    }
    const int64_t j = i - ((feat_elems + 255) / 256) * 256;
    if (j >= 0 && j < (int64_t)N * P) {
        const int n = (int)(j / P), q = (int)(j % P);
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += param_part[((int64_t)n * T + t) * P + q];
        g_params[j] = acc;
    }
}

static int fill_dyn(const float* feat, int B, int C, int H, int W, const float* params, int N, const float* coors,
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
    const unsigned grid = (unsigned)(N * bxi::dyn_tiles(H, W));
    if (C == 16) BXI_LAUNCH("dyn_fwd", s, (bxi::dyn_fwd_kernel<16>), dim3(grid), dim3(256), 0, s, a, logits);
    else BXI_LAUNCH("dyn_fwd", s, (bxi::dyn_fwd_kernel<8>), dim3(grid), dim3(256), 0, s, a, logits);
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
    const int cin = C + 2;
    const size_t lds = sizeof(float) * ((size_t)(1 + 4 * bxi::kDC + cin) * bxi::kRowPad + bxi::dyn_params(C, 1));
    const unsigned grid = (unsigned)(B * T * bxi::kSlots);
    if (C == 16) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::dyn_bwd_kernel<16>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        BXI_LAUNCH("dyn_bwd", s, (bxi::dyn_bwd_kernel<16>), dim3(grid), dim3(256), lds, s, a, g_logits, feat_part, param_part);
    } else {
        BXI_LAUNCH("dyn_bwd", s, (bxi::dyn_bwd_kernel<8>), dim3(grid), dim3(256), lds, s, a, g_logits, feat_part, param_part);
    }
    rc = bxi::check_launch();
    if (rc != BXI_OK) return rc;
    const int64_t nb = (feat_elems + 255) / 256 + ((int64_t)N * P + 255) / 256;
    BXI_LAUNCH("dyn_reduce", s, bxi::dyn_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, feat_part, feat_elems, g_feat,
               param_part, N, T, P, g_params);
    return bxi::check_launch();
}

}  // extern "C"
