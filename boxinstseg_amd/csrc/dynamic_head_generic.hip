// dynamic_head_generic.hip -- the dynamic mask head (CondInstMaskHead.forward, condinst_head.py:1139-1164; parse_dynamic_params
// :1120-1137; aligned_bilinear :146-167) for EVERY head shape the reference's constructor admits (:1079-1089 leaves dynamic_convs,
// dynamic_channels and in_channels free), forward and backward.  dynamic_head.hip holds the kernels tuned for the shape every
// shipped config uses (3 layers x 8 channels on 8 / 16 mask-feature channels); this file is the same arithmetic with
//   layers   1 .. 4                      (run-time loop over the middle layers; the kernels are instantiated per layer count)
//   channels 1 .. 16                     (padded to 4 / 8 / 16 with zero weights: a padded unit is relu(0) = 0 and feeds nothing)
//   in_channels (+2 relative coordinates) <= 34   (run-time loop)
// and any up-sampling factor.  Parameter layout per instance, as parse_dynamic_params splits it: all weights layer by layer
// (rows = output channels), then all biases.
//   dyn_fwd_generic_kernel<DC>     grid (tiles of 8 x 32 y-pixels, N): weights staged (padded) in LDS, a thread evaluates its halo
//                                  pixels, the tile is up-sampled from LDS.
//   dyn_dy_generic_kernel          d loss / d y = the transposed aligned_bilinear applied to d loss / d logits, by gather.
//   dyn_bwd_generic_kernel<DC, L>  grid (tiles, B): a workgroup owns a tile of ONE image and walks that image's instances: forward
//                                  recomputed, MLP backward per pixel, d feat accumulated in registers over the instances (written
//                                  once, no atomics), d params per (instance, tile) by staging operand rows in LDS -> partials.
//   dyn_param_reduce_kernel        g_params[n, q] = sum over tiles, in tile order (run-to-run identical).
// Not tuned to the last instruction -- the shipped shapes never come here -- but every byte and FLOP stays on the GPU in HIP.
#include "dynamic_head_device.hpp"
#include "loss_common.hpp"

namespace bxi {

constexpr int kGMaxL = 4, kGMaxDC = 16, kGMaxCin = 34;

struct GenShape {
    int L, Dc, Cin, C, P;                // layers, dynamic channels, inputs of the first layer (C + 2 with relative coordinates)
    int w_off[kGMaxL], b_off[kGMaxL];    // offsets into an instance's parameters (parse_dynamic_params' split)
    int n_in[kGMaxL], n_out[kGMaxL];
};

static inline bool gen_shape(int L, int Dc, int C, int rel, GenShape& s) {
    if (L < 1 || L > kGMaxL || Dc < 1 || Dc > kGMaxDC || C < 1 || C + (rel ? 2 : 0) > kGMaxCin) return false;
    s.L = L; s.Dc = Dc; s.C = C; s.Cin = C + (rel ? 2 : 0);
    int off = 0;
    for (int l = 0; l < L; ++l) {
        s.n_in[l] = l == 0 ? s.Cin : Dc;
        s.n_out[l] = l == L - 1 ? 1 : Dc;
        s.w_off[l] = off; off += s.n_in[l] * s.n_out[l];
    }
    for (int l = 0; l < L; ++l) { s.b_off[l] = off; off += s.n_out[l]; }
    s.P = off;
    return true;
}

// padded LDS layout (DC = padded width): layer 0  W[DC][Cin] b[DC] ; middle layers  W[DC][DC] b[DC] ; last  w[DC] b
template <int DC> struct GenLds {
    __device__ __forceinline__ static int w0(const GenShape&) { return 0; }
    __device__ __forceinline__ static int b0(const GenShape& s) { return DC * s.Cin; }
    __device__ __forceinline__ static int wm(const GenShape& s, int l) { return DC * s.Cin + DC + (l - 1) * (DC * DC + DC); }   // 1 <= l <= L-2
    __device__ __forceinline__ static int bm(const GenShape& s, int l) { return wm(s, l) + DC * DC; }
    __device__ __forceinline__ static int wl(const GenShape& s) { return DC * s.Cin + DC + (s.L > 2 ? (s.L - 2) * (DC * DC + DC) : 0); }
    __device__ __forceinline__ static int bl(const GenShape& s) { return wl(s) + DC; }
    __device__ __forceinline__ static int size(const GenShape& s) { return bl(s) + 1; }
};
static inline size_t gen_lds_floats(int DC, const GenShape& s) { return (size_t)DC * s.Cin + DC + (s.L > 2 ? (size_t)(s.L - 2) * (DC * DC + DC) : 0) + DC + 1; }

// stage instance n's parameters, padded; L == 1: the single layer is the "last" layer with Cin inputs (wl sized DC >= ... see below)
template <int DC>
__device__ __forceinline__ void stage_weights(const GenShape& s, const float* __restrict__ p, float* wts) {
    const int total = GenLds<DC>::size(s);
    for (int i = threadIdx.x; i < total; i += blockDim.x) wts[i] = 0.f;
    __syncthreads();
    if (s.L == 1) return;                // handled straight from global memory (Cin weights + one bias)
    for (int i = threadIdx.x; i < s.Dc * s.Cin; i += blockDim.x) wts[GenLds<DC>::w0(s) + i] = p[s.w_off[0] + i];      // rows o < Dc are contiguous
    for (int i = threadIdx.x; i < s.Dc; i += blockDim.x) wts[GenLds<DC>::b0(s) + i] = p[s.b_off[0] + i];
    for (int l = 1; l + 1 < s.L; ++l) {
        for (int i = threadIdx.x; i < s.Dc * s.Dc; i += blockDim.x) wts[GenLds<DC>::wm(s, l) + (i / s.Dc) * DC + i % s.Dc] = p[s.w_off[l] + i];
        for (int i = threadIdx.x; i < s.Dc; i += blockDim.x) wts[GenLds<DC>::bm(s, l) + i] = p[s.b_off[l] + i];
    }
    for (int i = threadIdx.x; i < s.Dc; i += blockDim.x) wts[GenLds<DC>::wl(s) + i] = p[s.w_off[s.L - 1] + i];
    if (threadIdx.x == 0) wts[GenLds<DC>::bl(s)] = p[s.b_off[s.L - 1]];
}

// input i of pixel (r, c) of image b for instance n: the relative coordinates first (:1143-1153), then the mask features
__device__ __forceinline__ float gen_input(const DynArgs& a, const GenShape& s, int n, int b, int r, int c, int i) {
    const int rel = s.Cin - s.C;
    if (i < rel) {
        const float soi = a.soi[a.level[n]];
        return i == 0 ? (a.coors[2 * n] - (float)(c * a.in_stride + a.in_stride / 2)) / soi
                      : (a.coors[2 * n + 1] - (float)(r * a.in_stride + a.in_stride / 2)) / soi;
    }
    return a.feat[(((int64_t)b * s.C + (i - rel)) * a.H + r) * a.W + c];
}

// one pixel through the head.  act[l][o] = output of layer l after ReLU (l < L-1), kept for the backward when KEEP.
template <int DC, bool KEEP>
__device__ __forceinline__ float gen_mlp(const DynArgs& a, const GenShape& s, const float* wts, const float* __restrict__ p, int n, int b, int r, int c,
                                         float (&act)[kGMaxL - 1][DC]) {
    if (s.L == 1) {
        float y = p[s.b_off[0]];
        for (int i = 0; i < s.Cin; ++i) y += p[s.w_off[0] + i] * gen_input(a, s, n, b, r, c, i);
        return y;
    }
    float h[DC];
#pragma unroll
    for (int o = 0; o < DC; ++o) h[o] = wts[GenLds<DC>::b0(s) + o];
    for (int i = 0; i < s.Cin; ++i) {
        const float x = gen_input(a, s, n, b, r, c, i);
#pragma unroll
        for (int o = 0; o < DC; ++o) h[o] += wts[GenLds<DC>::w0(s) + o * s.Cin + i] * x;
    }
#pragma unroll
    for (int o = 0; o < DC; ++o) { h[o] = fmaxf(h[o], 0.f); if (KEEP) act[0][o] = h[o]; }
#pragma unroll
    for (int l = 1; l < kGMaxL - 1; ++l) {
        if (l + 1 < s.L) {
            float t[DC];
            const float* W = wts + GenLds<DC>::wm(s, l);
#pragma unroll
            for (int o = 0; o < DC; ++o) {
                float acc = W[DC * DC + o];
#pragma unroll
                for (int i = 0; i < DC; ++i) acc += W[o * DC + i] * h[i];
                t[o] = fmaxf(acc, 0.f);
            }
#pragma unroll
            for (int o = 0; o < DC; ++o) { h[o] = t[o]; if (KEEP) act[l][o] = t[o]; }
        }
    }
    float y = wts[GenLds<DC>::bl(s)];
#pragma unroll
    for (int i = 0; i < DC; ++i) y += wts[GenLds<DC>::wl(s) + i] * h[i];
    return y;
}

template <int DC>
__global__ __launch_bounds__(256) void dyn_fwd_generic_kernel(DynArgs a, GenShape s, const float* __restrict__ params, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int kHalo = (kYR + 2) * (kYC + 2);
    float* ytile = gsm;
    float* wts = gsm + kHalo;
    const int tiles_x = (a.W + kYC - 1) / kYC;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
    const float* p = params + (int64_t)n * s.P;
    const int b = (int)a.img[n];
    stage_weights<DC>(s, p, wts);
    __syncthreads();
    const int r0 = ty * kYR, c0 = tx * kYC;
    for (int e = threadIdx.x; e < kHalo; e += 256) {
        const int r = r0 - 1 + e / (kYC + 2), c = c0 - 1 + e % (kYC + 2);
        const bool v = r >= 0 && r < a.H && c >= 0 && c < a.W;
        float act[kGMaxL - 1][DC];
        ytile[e] = v ? gen_mlp<DC, false>(a, s, wts, p, n, b, r, c, act) : 0.f;
    }
    __syncthreads();
    const int f = a.factor, OH = a.H * f, OW = a.W * f;
    const int R0 = r0 * f, C0 = c0 * f;
    float* out = logits + (int64_t)n * OH * OW;
    for (int i = threadIdx.x; i < kYR * f * kYC * f; i += 256) {
        const int R = R0 + i / (kYC * f), Cc = C0 + i % (kYC * f);
        if (R >= OH || Cc >= OW) continue;
        int y0, y1, x0, x1; float fy, fx;
        upsample_src(R, f, a.H, y0, y1, fy);
        upsample_src(Cc, f, a.W, x0, x1, fx);
        auto Y = [&](int r, int c) { return ytile[(r - r0 + 1) * (kYC + 2) + (c - c0 + 1)]; };
        const float top = (1.f - fx) * Y(y0, x0) + fx * Y(y0, x1);
        const float bot = (1.f - fx) * Y(y1, x0) + fx * Y(y1, x1);
        out[(int64_t)R * OW + Cc] = (1.f - fy) * top + fy * bot;
    }
}

// d loss / d y[n, r, c] = sum over the outputs (R, Cc) that sample y(r, c) of their weight x d loss / d logits (transposed :146-167)
__global__ __launch_bounds__(256) void dyn_dy_generic_kernel(const float* __restrict__ g_logits, int N, int H, int W, int f, float* __restrict__ dy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int c = (int)(i % W), r = (int)((i / W) % H);
    const int64_t n = i / ((int64_t)H * W);
    const int OH = H * f, OW = W * f;
    const float* g = g_logits + n * OH * OW;
    // outputs R with source rows {r0, r0 + 1} containing r:  (r - 1) f + f/2 <= R < (r + 1) f + f/2, plus R < f/2 for r == 0
    const int Ra = max((r - 1) * f + f / 2, 0), Rb = min((r + 1) * f + f / 2, OH);
    const int Ca = max((c - 1) * f + f / 2, 0), Cb = min((c + 1) * f + f / 2, OW);
    float acc = 0.f;
    for (int R = (r == 0 ? 0 : Ra); R < Rb; ++R) {
        int y0, y1; float fy;
        upsample_src(R, f, H, y0, y1, fy);
        const float wy = (y0 == r ? 1.f - fy : 0.f) + (y1 == r ? fy : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int Cc = (c == 0 ? 0 : Ca); Cc < Cb; ++Cc) {
            int x0, x1; float fx;
            upsample_src(Cc, f, W, x0, x1, fx);
            const float wx = (x0 == c ? 1.f - fx : 0.f) + (x1 == c ? fx : 0.f);
            row += wx * g[(int64_t)R * OW + Cc];
        }
        acc += wy * row;
    }
    dy[i] = acc;
}

// rows of 256 pixel values in LDS, padded so that different rows of one pixel column fall into different banks
constexpr int kGRow = 257;

template <int DC, int L>
__global__ __launch_bounds__(256) void dyn_bwd_generic_kernel(DynArgs a, GenShape s, const float* __restrict__ params, const float* __restrict__ dy,
                                                              float* __restrict__ g_feat, float* __restrict__ part /* [N][T][P] */, int T) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* wts = gsm;
    float* rin = gsm + ((GenLds<DC>::size(s) + 3) & ~3);          // [max n_in][kGRow]
    float* rdl = rin + (size_t)kGMaxCin * kGRow;                   // [DC][kGRow]
    const int tiles_x = (a.W + kYC - 1) / kYC;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int r = ty * kYR + tid / kYC, c = tx * kYC + tid % kYC;
    const bool live = r < a.H && c < a.W;
    const int rc = min(r, a.H - 1), cc = min(c, a.W - 1);
    float gf[kGMaxCin];                                            // d feat of this pixel, summed over the image's instances
#pragma unroll
    for (int i = 0; i < kGMaxCin; ++i) gf[i] = 0.f;
    for (int n = 0; n < a.N; ++n) {
        if ((int)a.img[n] != b) continue;                          // uniform
        const float* p = params + (int64_t)n * s.P;
        __syncthreads();                                           // the previous instance's rows and weights are done with
        stage_weights<DC>(s, p, wts);
        __syncthreads();
        float act[kGMaxL - 1][DC];
        (void)gen_mlp<DC, true>(a, s, wts, p, n, b, rc, cc, act);
        const float gy = live ? dy[((int64_t)n * a.H + rc) * a.W + cc] : 0.f;
        float* pn = part + ((int64_t)n * T + tile) * s.P;
        // ---- the last layer: y = b + w . h_{L-2}  (L == 1: y = b + w . x)
        if constexpr (L == 1) {
            for (int i = 0; i < s.Cin; ++i) rin[i * kGRow + tid] = live ? gen_input(a, s, n, b, rc, cc, i) : 0.f;
            rdl[tid] = gy;
            __syncthreads();
            for (int q = tid; q <= s.Cin; q += 256) {              // Cin weights + the bias
                float acc = 0.f;
                if (q < s.Cin) for (int px = 0; px < 256; ++px) acc += rdl[px] * rin[q * kGRow + px];
                else for (int px = 0; px < 256; ++px) acc += rdl[px];
                pn[q < s.Cin ? s.w_off[0] + q : s.b_off[0]] = acc;
            }
            const int rel = s.Cin - s.C;
#pragma unroll
            for (int i = 0; i < kGMaxCin; ++i) if (i >= rel && i < s.Cin) gf[i] += p[s.w_off[0] + i] * gy;
        } else {
        float d[DC];                                               // delta of the layer below (after its ReLU mask)
#pragma unroll
        for (int o = 0; o < DC; ++o) {
            rin[o * kGRow + tid] = act[L - 2][o];
            d[o] = act[L - 2][o] > 0.f ? wts[GenLds<DC>::wl(s) + o] * gy : 0.f;
        }
        rdl[tid] = gy;
        __syncthreads();
        for (int q = tid; q <= s.Dc; q += 256) {
            float acc = 0.f;
            if (q < s.Dc) for (int px = 0; px < 256; ++px) acc += rdl[px] * rin[q * kGRow + px];
            else for (int px = 0; px < 256; ++px) acc += rdl[px];
            pn[q < s.Dc ? s.w_off[L - 1] + q : s.b_off[L - 1]] = acc;
        }
        // ---- the middle layers, top down: h_l = relu(W_l h_{l-1} + b_l), delta_l = d
#pragma unroll
        for (int l = L - 2; l >= 1; --l) {
            __syncthreads();
#pragma unroll
            for (int o = 0; o < DC; ++o) { rin[o * kGRow + tid] = act[l - 1][o]; rdl[o * kGRow + tid] = d[o]; }
            __syncthreads();
            for (int q = tid; q < s.Dc * s.Dc + s.Dc; q += 256) {
                float acc = 0.f;
                if (q < s.Dc * s.Dc) {
                    const int o = q / s.Dc, i = q % s.Dc;
                    for (int px = 0; px < 256; ++px) acc += rdl[o * kGRow + px] * rin[i * kGRow + px];
                    pn[s.w_off[l] + q] = acc;
                } else {
                    const int o = q - s.Dc * s.Dc;
                    for (int px = 0; px < 256; ++px) acc += rdl[o * kGRow + px];
                    pn[s.b_off[l] + o] = acc;
                }
            }
            float dn[DC];
            const float* W = wts + GenLds<DC>::wm(s, l);
#pragma unroll
            for (int i = 0; i < DC; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int o = 0; o < DC; ++o) acc += W[o * DC + i] * d[o];
                dn[i] = act[l - 1][i] > 0.f ? acc : 0.f;
            }
#pragma unroll
            for (int o = 0; o < DC; ++o) d[o] = dn[o];
        }
        // ---- the first layer: h_0 = relu(W_0 x + b_0), delta_0 = d
        __syncthreads();
        for (int i = 0; i < s.Cin; ++i) rin[i * kGRow + tid] = live ? gen_input(a, s, n, b, rc, cc, i) : 0.f;
#pragma unroll
        for (int o = 0; o < DC; ++o) rdl[o * kGRow + tid] = live ? d[o] : 0.f;
        __syncthreads();
        for (int q = tid; q < s.Dc * s.Cin + s.Dc; q += 256) {
            float acc = 0.f;
            if (q < s.Dc * s.Cin) {
                const int o = q / s.Cin, i = q % s.Cin;
                for (int px = 0; px < 256; ++px) acc += rdl[o * kGRow + px] * rin[i * kGRow + px];
                pn[s.w_off[0] + q] = acc;
            } else {
                const int o = q - s.Dc * s.Cin;
                for (int px = 0; px < 256; ++px) acc += rdl[o * kGRow + px];
                pn[s.b_off[0] + o] = acc;
            }
        }
        const int rel = s.Cin - s.C;
#pragma unroll
        for (int i = 0; i < kGMaxCin; ++i)
            if (i >= rel && i < s.Cin) {
                float acc = 0.f;
#pragma unroll
                for (int o = 0; o < DC; ++o) acc += wts[GenLds<DC>::w0(s) + o * s.Cin + i] * d[o];
                gf[i] += acc;
            }
        }
    }
    if (live) {
        const int rel = s.Cin - s.C;
#pragma unroll
        for (int i = 0; i < kGMaxCin; ++i)
            if (i >= rel && i < s.Cin) g_feat[(((int64_t)b * s.C + (i - rel)) * a.H + r) * a.W + c] = gf[i];
    }
}

__global__ __launch_bounds__(256) void dyn_param_reduce_kernel(const float* __restrict__ part, int N, int T, int P, float* __restrict__ g_params) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= (int64_t)N * P) return;
    const int n = (int)(j / P), q = (int)(j % P);
    const float* src = part + (int64_t)n * T * P + q;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += src[(int64_t)t * P];      // tile order: run-to-run identical
    g_params[j] = acc;
}

static int fill_dyn_generic(const float* feat, int B, int C, int H, int W, const float* params, int N, const float* coors, const int64_t* level,
                            const int64_t* img, const float* soi, int n_levels, int in_stride, int factor, int disable_rel, DynArgs& a) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || N < 0 || n_levels <= 0) return BXI_ERR_BAD_SHAPE;
    if (in_stride < 1 || factor < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N > 0 && (!feat || !params || !coors || !level || !img || !soi)) return BXI_ERR_NULL_POINTER;
    if (!fits_i32((int64_t)N * H * W * factor * factor) || !fits_i32((int64_t)B * C * H * W)) return BXI_ERR_BAD_SHAPE;
    a.feat = feat; a.params = params; a.coors = coors; a.level = level; a.img = img; a.soi = soi;
    a.B = B; a.H = H; a.W = W; a.N = N; a.n_levels = n_levels; a.in_stride = in_stride; a.factor = factor;
    a.rel = disable_rel ? 0 : 1;
    return BXI_OK;
}

static inline int pad_dc(int Dc) { return Dc <= 4 ? 4 : (Dc <= 8 ? 8 : 16); }
static inline size_t gen_dy_bytes(int N, int H, int W) { return align_up(sizeof(float) * (size_t)(N > 0 ? N : 1) * H * W, 256); }

}  // namespace bxi

extern "C" {

int bxi_dynamic_mask_generic_forward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N, int layers, int channels,
                                         const float* coors, const int64_t* level_inds, const int64_t* img_inds, const float* sizes_of_interest,
                                         int n_levels, int in_stride, int factor, int disable_rel_coors, float* logits, void* stream) {
    using namespace bxi;
    GenShape s;
    if (!gen_shape(layers, channels, C, !disable_rel_coors, s)) return BXI_ERR_UNSUPPORTED;
    DynArgs a;
    const int rc = fill_dyn_generic(feat, B, C, H, W, params, N, coors, level_inds, img_inds, sizes_of_interest, n_levels, in_stride, factor, disable_rel_coors, a);
    if (rc != BXI_OK) return rc;
    if (N == 0) return BXI_OK;
    if (!logits) return BXI_ERR_NULL_POINTER;
    if (N > 65535) return BXI_ERR_BAD_SHAPE;
    hipStream_t st = as_stream(stream);
    const int DC = pad_dc(channels);
    const dim3 grid((unsigned)(((H + kYR - 1) / kYR) * ((W + kYC - 1) / kYC)), (unsigned)N);
    const size_t lds = sizeof(float) * ((size_t)(kYR + 2) * (kYC + 2) + gen_lds_floats(DC, s));
    if (DC == 4) BXI_LAUNCH("dyn_fwd_generic", st, dyn_fwd_generic_kernel<4>, grid, dim3(256), lds, st, a, s, params, logits);
    else if (DC == 8) BXI_LAUNCH("dyn_fwd_generic", st, dyn_fwd_generic_kernel<8>, grid, dim3(256), lds, st, a, s, params, logits);
    else BXI_LAUNCH("dyn_fwd_generic", st, dyn_fwd_generic_kernel<16>, grid, dim3(256), lds, st, a, s, params, logits);
    return check_launch();
}

size_t bxi_dynamic_mask_generic_backward_workspace_bytes(int B, int C, int H, int W, int N, int layers, int channels, int disable_rel_coors) {
    using namespace bxi;
    GenShape s;
    (void)B;
    if (!gen_shape(layers, channels, C, !disable_rel_coors, s) || H <= 0 || W <= 0 || N < 0) return 0;
    const size_t T = (size_t)((H + kYR - 1) / kYR) * ((W + kYC - 1) / kYC);
    return gen_dy_bytes(N, H, W) + align_up(sizeof(float) * (size_t)(N > 0 ? N : 1) * T * s.P, 256);
}

int bxi_dynamic_mask_generic_backward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N, int layers, int channels,
                                          const float* coors, const int64_t* level_inds, const int64_t* img_inds, const float* sizes_of_interest,
                                          int n_levels, int in_stride, int factor, int disable_rel_coors, const float* g_logits, float* g_feat,
                                          float* g_params, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace bxi;
    GenShape s;
    if (!gen_shape(layers, channels, C, !disable_rel_coors, s)) return BXI_ERR_UNSUPPORTED;
    DynArgs a;
    const int rc = fill_dyn_generic(feat, B, C, H, W, params, N, coors, level_inds, img_inds, sizes_of_interest, n_levels, in_stride, factor, disable_rel_coors, a);
    if (rc != BXI_OK) return rc;
    if (!g_feat || (N > 0 && (!g_params || !g_logits))) return BXI_ERR_NULL_POINTER;
    if (B > 65535) return BXI_ERR_BAD_SHAPE;
    hipStream_t st = as_stream(stream);
    const size_t need = bxi_dynamic_mask_generic_backward_workspace_bytes(B, C, H, W, N, layers, channels, disable_rel_coors);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    float* dy = reinterpret_cast<float*>(workspace);
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + gen_dy_bytes(N, H, W));
    const int T = ((H + kYR - 1) / kYR) * ((W + kYC - 1) / kYC);
    if (N > 0) {
        const int64_t npx = (int64_t)N * H * W;
        if (!fits_i32((npx + 255) / 256)) return BXI_ERR_BAD_SHAPE;
        BXI_LAUNCH("dyn_dy_generic", st, dyn_dy_generic_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, g_logits, N, H, W, factor, dy);
    }
    const int DC = pad_dc(channels);
    const size_t lds = sizeof(float) * (((gen_lds_floats(DC, s) + 3) & ~(size_t)3) + (size_t)(kGMaxCin + DC) * kGRow);
    const dim3 grid((unsigned)T, (unsigned)B);
#define BXI_GEN_BWD(DCC, LL)                                                                                                               \
    {                                                                                                                                      \
        if (lds > 64 * 1024) {                                                                                                             \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dyn_bwd_generic_kernel<DCC, LL>),                              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
            if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }                                                    \
        }                                                                                                                                  \
        BXI_LAUNCH("dyn_bwd_generic", st, (dyn_bwd_generic_kernel<DCC, LL>), grid, dim3(256), lds, st, a, s, params, (const float*)dy, g_feat, part, T); \
    }
#define BXI_GEN_BWD_L(DCC)                                                                                                                 \
    switch (layers) { case 1: BXI_GEN_BWD(DCC, 1) break; case 2: BXI_GEN_BWD(DCC, 2) break; case 3: BXI_GEN_BWD(DCC, 3) break; default: BXI_GEN_BWD(DCC, 4) break; }
    if (DC == 4) BXI_GEN_BWD_L(4) else if (DC == 8) BXI_GEN_BWD_L(8) else BXI_GEN_BWD_L(16)
#undef BXI_GEN_BWD_L
#undef BXI_GEN_BWD
    int rc2 = check_launch();
    if (rc2 != BXI_OK) return rc2;
    if (N > 0) {
        const int64_t np = (int64_t)N * s.P;
        BXI_LAUNCH("dyn_param_reduce", st, dyn_param_reduce_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, (const float*)part, N, T, s.P, g_params);
    }
    return check_launch();
}

}  // extern "C"
