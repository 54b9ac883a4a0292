// mask_loss.hip -- BoxInst projection + pairwise loss, forward AND backward, on gfx950.
//
// Replaces (reference, LiWentomng/BoxInstSeg) CondInstMaskHead.loss with boxinst_enabled,
// condinst_head.py:1288-1343:
//   mask_logits.sigmoid() / compute_project_term      :1300, :134-143 (+ dice_coefficient :117-131)
//   pairwise_nlog (CUDA op, pairwise.cu:68-149)       :1321
//   weights = (sim >= thresh) * bitmask, normalise    :1324-1328
//   warm-up                                           :1330-1332
// and everything autograd does behind them (max backward, sigmoid backward, the op's atomicAdd
// backward, ~1.3 GB of elementwise temporaries at 2x800x1024x32) with two launches:
//
// Kernel C  loss_main  (grid = N x ceil(h/TR) row tiles, 256 threads = 4 wave64)
//   streams the logits once (float4 per lane, a wave = one 1 KiB row segment):
//     - per-row max/arg-max of the logits by wave shuffles (complete inside the workgroup),
//       per-column max/arg-max over the tile's rows -> one partial per tile (no atomics);
//       max_p sigmoid(x_p) = sigmoid(max_p x_p), so sigmoid is applied to h+w maxima only;
//     - only where the tile meets the instance's box dilated by `dilation` (~10 % of the map):
//       sigmoid pairs (p, 1-p) staged in LDS with a `dilation` halo, 8-neighbour pairwise
//       -log(p_i p_j + q_i q_j) and its gradient in gather form (symmetric pair, channel 7-k =
//       opposite offset, so no atomics and a fixed summation order), weighted by the K-bit
//       colour-affinity mask of the centre (bit k) and of the neighbour (bit 7-k);
//     - writes the UN-normalised pairwise gradient (zeros outside the dilated box): every
//       element of g_logits is written exactly once, as float4.
//   HBM roofline: 4 B read + 4 B written per instance-pixel (+ <1 B of partials).
// Kernel D  loss_finalize (grid = S x N)
//   dice per instance from the h+w maxima, unit projection gradients at the arg-max positions,
//   sum of the weight counts -> 1/max(sum W,1), rescale of the dilated-box region in place, the
//   two loss scalars by the last-arriving workgroup (ticket), deterministic order.
#include "common.hpp"

namespace bxi {

constexpr int kTR = 8;          // rows per tile of loss_main
constexpr int kChunk = 256;     // columns per pass: 64 lanes x float4
constexpr int kSlices = 4;      // row slices per instance in loss_finalize
constexpr int kMaxDil = 8;

struct InstArgs {
    const float* logits;
    const int64_t* gt_inds;
    int N, h, w;
    int Hc, Wc, stride;
    GtTable gt;
};

struct LossWs {               // carved from the caller's workspace
    float* colv;              // [N,T,w] per-tile column max (logit)
    uint8_t* colr;            // [N,T,w] row offset of that max inside the tile
    unsigned long long* rowkey;  // [N,h] packed (max logit, first column)
    float* part_num;          // [N*T]
    int* part_cnt;            // [N*T]
    float* dice;              // [N]
    unsigned int* ticket;     // [1]
};

struct LossState {            // kept for bxi_boxinst_loss_rescale_f32
    int* colarg;              // [N,w] arg-max row of column c
    int* rowarg;              // [N,h] arg-max column of row r
    float* gcol;              // [N,w] unit d loss_prj / d logit at (colarg[c], c)
    float* grow;              // [N,h] unit d loss_prj / d logit at (r, rowarg[r])
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int tiles_of(int h) { return (h + kTR - 1) / kTR; }

static size_t carve_ws(void* base, int N, int h, int w, LossWs* ws) {
    const size_t T = (size_t)tiles_of(h);
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    float* colv = (float*)take(sizeof(float) * N * T * w);
    uint8_t* colr = (uint8_t*)take((size_t)N * T * w);
    unsigned long long* rowkey = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)N * h);
    float* part_num = (float*)take(sizeof(float) * N * T);
    int* part_cnt = (int*)take(sizeof(int) * N * T);
    float* dice = (float*)take(sizeof(float) * (size_t)(N > 0 ? N : 1));
    unsigned int* ticket = (unsigned int*)take(sizeof(unsigned int));
    if (ws) { ws->colv = colv; ws->colr = colr; ws->rowkey = rowkey; ws->part_num = part_num;
              ws->part_cnt = part_cnt; ws->dice = dice; ws->ticket = ticket; }
    return off;
}

static size_t carve_state(void* base, int N, int h, int w, LossState* st) {
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    int* colarg = (int*)take(sizeof(int) * (size_t)N * w);
    int* rowarg = (int*)take(sizeof(int) * (size_t)N * h);
    float* gcol = (float*)take(sizeof(float) * (size_t)N * w);
    float* grow = (float*)take(sizeof(float) * (size_t)N * h);
    if (st) { st->colarg = colarg; st->rowarg = rowarg; st->gcol = gcol; st->grow = grow; }
    return off;
}

// ---- device helpers ----------------------------------------------------------------------------
struct InstBox {
    Rect box;   // cells whose sample lies in the GT box            (bitmask == 1)
    Rect dil;   // box grown by `dilation`, clipped                  (pairwise gradient != 0)
    int img;
    bool any;
};

__device__ __forceinline__ InstBox inst_box(const InstArgs& a, int n, int dil) {
    InstBox ib;
    ib.img = 0;
    ib.box.r0 = ib.box.r1 = ib.box.c0 = ib.box.c1 = 0;
    const int64_t g = a.gt_inds[n];
    if (g >= 0 && g < a.gt.first[a.gt.B]) {
        const float* bx = gt_box(a.gt, (int)g, ib.img);
        ib.box = box_rect(bx, a.Hc, a.Wc, a.stride, a.stride / 2, a.h, a.w);
    }
    ib.any = ib.box.r1 > ib.box.r0 && ib.box.c1 > ib.box.c0;
    ib.dil = ib.box;
    if (ib.any) {
        ib.dil.r0 = max(ib.box.r0 - dil, 0); ib.dil.r1 = min(ib.box.r1 + dil, a.h);
        ib.dil.c0 = max(ib.box.c0 - dil, 0); ib.dil.c1 = min(ib.box.c1 + dil, a.w);
    }
    return ib;
}

// (p, q) = (sigmoid(x), sigmoid(-x)), both accurate relatively (no 1-p cancellation)
__device__ __forceinline__ float2 sig_pair(float x) {
    const float e = __expf(-fabsf(x));
    const float r = __frcp_rn(1.f + e);
    const float er = e * r;
    return x >= 0.f ? make_float2(r, er) : make_float2(er, r);
}

__device__ __forceinline__ float4 load4(const float* row, int c, int w, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(row + c);
    float4 v;
    v.x = c + 0 < w ? row[c + 0] : -INFINITY;
    v.y = c + 1 < w ? row[c + 1] : -INFINITY;
    v.z = c + 2 < w ? row[c + 2] : -INFINITY;
    v.w = c + 3 < w ? row[c + 3] : -INFINITY;
    return v;
}
__device__ __forceinline__ void store4(float* row, int c, int w, bool vec, float4 v) {
    if (vec) { *reinterpret_cast<float4*>(row + c) = v; return; }
    if (c + 0 < w) row[c + 0] = v.x;
    if (c + 1 < w) row[c + 1] = v.y;
    if (c + 2 < w) row[c + 2] = v.z;
    if (c + 3 < w) row[c + 3] = v.w;
}

// ---- Kernel C ----------------------------------------------------------------------------------
// LDS: pq   [(kTR+2d)][w] float2   sigmoid pairs, tile rows + halo   (only when the tile meets the box)
//      aff  [(kTR+2d)][w] uint8    affinity bits
//      gt   [kTR][w]      float    gradient tile
//      cbv  [4][kChunk]   float    per-wave column maxima, cbr [4][kChunk] int
__global__ __launch_bounds__(256) void loss_main_kernel(InstArgs a, const uint8_t* __restrict__ affinity, int dil,
                                                        LossWs ws, float* __restrict__ g_logits, int vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = (a.h + kTR - 1) / kTR;
    const int n = blockIdx.x / T, t = blockIdx.x % T;
    const int r0 = t * kTR, r1 = min(a.h, r0 + kTR);
    const int w = a.w, h = a.h;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int64_t P = (int64_t)h * w;
    const float* L = a.logits + (int64_t)n * P;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;

    const InstBox ib = inst_box(a, n, dil);
    const bool hit = ib.any && r0 < ib.dil.r1 && r1 > ib.dil.r0;   // wave-uniform (workgroup-uniform)

    const int HR = kTR + 2 * dil;                                   // staged rows
    float2* pq = reinterpret_cast<float2*>(smem);
    float* gtile = reinterpret_cast<float*>(smem + sizeof(float2) * (size_t)HR * w);
    float* cbv = gtile + (size_t)kTR * w;
    int* cbr = reinterpret_cast<int*>(cbv + 4 * kChunk);
    uint8_t* afl = reinterpret_cast<uint8_t*>(cbr + 4 * kChunk);

    // ---- phase 1: stream the tile rows -----------------------------------------------------
    unsigned long long rkey[kTR / 4];
#pragma unroll
    for (int i = 0; i < kTR / 4; ++i) rkey[i] = 0ull;

    for (int cb = 0; cb < w; cb += kChunk) {
        const int c = cb + lane * 4;
        float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int crow[4] = {r0, r0, r0, r0};
        if (c < w) {
#pragma unroll
            for (int i = 0; i < kTR / 4; ++i) {
                const int r = r0 + wv + 4 * i;
                if (r < r1) {
                    const float4 v = load4(L + (int64_t)r * w, c, w, vec);
                    // row arg-max, first column wins ties
                    float m = v.x; int mc = c;
                    if (v.y > m) { m = v.y; mc = c + 1; }
                    if (v.z > m) { m = v.z; mc = c + 2; }
                    if (v.w > m) { m = v.w; mc = c + 3; }
                    const unsigned long long key = pack_max(m, (uint32_t)mc);
                    rkey[i] = key > rkey[i] ? key : rkey[i];
                    // column arg-max over this wave's rows (ascending r, strict > keeps the first)
                    if (v.x > cmax[0]) { cmax[0] = v.x; crow[0] = r; }
                    if (v.y > cmax[1]) { cmax[1] = v.y; crow[1] = r; }
                    if (v.z > cmax[2]) { cmax[2] = v.z; crow[2] = r; }
                    if (v.w > cmax[3]) { cmax[3] = v.w; crow[3] = r; }
                    if (hit) {
                        const int lr = r - r0 + dil;
                        float2* dst = pq + (size_t)lr * w + c;
                        if (c + 0 < w) dst[0] = sig_pair(v.x);
                        if (c + 1 < w) dst[1] = sig_pair(v.y);
                        if (c + 2 < w) dst[2] = sig_pair(v.z);
                        if (c + 3 < w) dst[3] = sig_pair(v.w);
                    } else if (G) {
                        store4(G + (int64_t)r * w, c, w, vec, make_float4(0.f, 0.f, 0.f, 0.f));
                    }
                }
            }
        }
        // combine the 4 waves' column maxima through LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) { cbv[wv * kChunk + lane * 4 + j] = cmax[j]; cbr[wv * kChunk + lane * 4 + j] = crow[j]; }
        __syncthreads();
        {
            const int cc = cb + tid;
            if (cc < w) {
                float m = cbv[tid]; int mr = cbr[tid];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float v = cbv[q * kChunk + tid]; const int vr = cbr[q * kChunk + tid];
                    if (v > m || (v == m && vr < mr)) { m = v; mr = vr; }
                }
                const int64_t o = ((int64_t)n * T + t) * w + cc;
                ws.colv[o] = m;
                ws.colr[o] = (uint8_t)(mr - r0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kTR / 4; ++i) {
        const int r = r0 + wv + 4 * i;
        const unsigned long long k = wave_max_u64(rkey[i]);
        if (lane == 0 && r < r1) ws.rowkey[(int64_t)n * h + r] = k;
    }

    float num = 0.f;
    int cnt = 0;
    if (hit) {
        // ---- phase 2a: halo rows + affinity bits + zero the gradient tile ---------------------
        const uint8_t* AF = affinity + (int64_t)ib.img * P;
        for (int i = tid; i < HR * (w / 4 + ((w & 3) ? 1 : 0)); i += 256) {
            const int wq = (w + 3) / 4;
            const int lr = i / wq, c = (i % wq) * 4;
            const int r = r0 - dil + lr;
            const bool own = lr >= dil && lr < dil + kTR;
            if (r >= 0 && r < h) {
                if (!own || r >= r1) {
                    const float4 v = load4(L + (int64_t)r * w, c, w, vec);
                    float2* dst = pq + (size_t)lr * w + c;
                    if (c + 0 < w) dst[0] = sig_pair(v.x);
                    if (c + 1 < w) dst[1] = sig_pair(v.y);
                    if (c + 2 < w) dst[2] = sig_pair(v.z);
                    if (c + 3 < w) dst[3] = sig_pair(v.w);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < w) afl[(size_t)lr * w + c + j] = AF[(int64_t)r * w + c + j];
            }
        }
        for (int i = tid; i < kTR * w; i += 256) gtile[i] = 0.f;
        __syncthreads();

        // ---- phase 2b: pairwise term on (tile rows) x (dilated box columns) -------------------
        const int ra = max(r0, ib.dil.r0), rb = min(r1, ib.dil.r1);
        const int cw = ib.dil.c1 - ib.dil.c0;
        const int npx = (rb - ra) * cw;
        for (int i = tid; i < npx; i += 256) {
            const int r = ra + i / cw, c = ib.dil.c0 + i % cw;
            const int lr = r - r0 + dil;
            const float2 pp = pq[(size_t)lr * w + c];
            const bool in_p = r >= ib.box.r0 && r < ib.box.r1 && c >= ib.box.c0 && c < ib.box.c1;
            const uint32_t bits_p = in_p ? afl[(size_t)lr * w + c] : 0u;   // W[k,p] = bit k of p, p in box
            float acc = 0.f;
            int k = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dx == 0 && dy == 0) continue;
                    const int r2 = r + dy * dil, c2 = c + dx * dil;
                    const uint32_t wp = (bits_p >> k) & 1u;
                    cnt += (int)wp;                                          // weights.sum(), :1328
                    if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w) {
                        const int lr2 = lr + dy * dil;
                        const bool in_q = r2 >= ib.box.r0 && r2 < ib.box.r1 && c2 >= ib.box.c0 && c2 < ib.box.c1;
                        const uint32_t wq = in_q ? ((uint32_t)afl[(size_t)lr2 * w + c2] >> (7 - k)) & 1u : 0u;
                        const uint32_t ws2 = wp + wq;
                        if (ws2) {
                            const float2 qq = pq[(size_t)lr2 * w + c2];
                            const float S = pp.x * qq.x + pp.y * qq.y;      // P(y_p == y_q)
                            float nl, coef;
                            if (S > 1e-30f) {
                                nl = -__logf(S);
                                coef = -(qq.x - qq.y) * (pp.x * pp.y) * __frcp_rn(S);
                            } else {   // |logit| beyond ~69: log-space evaluation as pairwise.cu:38-61
                                const float xa = L[(int64_t)r * w + c], xb = L[(int64_t)r2 * w + c2];
                                const float ax = logsig(xa), bx = logsig(-xa), ay = logsig(xb), by = logsig(-xb);
                                const float e1 = ax + ay, e0 = bx + by;
                                const float mx = fmaxf(e1, e0), df = fabsf(e1 - e0);
                                nl = logsig(df) - mx;
                                coef = -(expf(ay) - expf(by)) * expf(ax + bx + nl);
                            }
                            num += (float)wp * nl;
                            acc += (float)ws2 * coef;
                        }
                    }
                    ++k;
                }
            gtile[(size_t)(r - r0) * w + c] = acc;
        }
        __syncthreads();

        // ---- phase 3: write the gradient tile ---------------------------------------------------
        if (G) {
            const int wq = (w + 3) / 4;
            for (int i = tid; i < (r1 - r0) * wq; i += 256) {
                const int lr = i / wq, c = (i % wq) * 4;
                float4 v;
                const float* src = gtile + (size_t)lr * w + c;
                v.x = src[0];
                v.y = c + 1 < w ? src[1] : 0.f;
                v.z = c + 2 < w ? src[2] : 0.f;
                v.w = c + 3 < w ? src[3] : 0.f;
                store4(G + (int64_t)(r0 + lr) * w, c, w, vec, v);
            }
        }
    }

    // ---- block partials (fixed order: lanes by shuffle tree, waves 0..3) -------------------------
    num = wave_sum_f32(num);
    cnt = wave_sum_i32(cnt);
    __syncthreads();
    if (lane == 0) { cbv[wv] = num; cbr[wv] = cnt; }
    __syncthreads();
    if (tid == 0) {
        ws.part_num[blockIdx.x] = (cbv[0] + cbv[1]) + (cbv[2] + cbv[3]);
        ws.part_cnt[blockIdx.x] = cbr[0] + cbr[1] + cbr[2] + cbr[3];
        if (blockIdx.x == 0) *ws.ticket = 0u;   // consumed by loss_finalize after the kernel boundary
    }
}

// ---- Kernel D ----------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_f32(float v, float* red) {
    v = wave_sum_f32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// MODE 0: finalize after loss_main.  MODE 1: rescale with device upstream gradients.
template <int MODE>
__global__ __launch_bounds__(256) void loss_finalize_kernel(InstArgs a, int dil, float warmup, LossWs ws, LossState st,
                                                            const float* __restrict__ up_prj,
                                                            const float* __restrict__ up_pw,
                                                            float* __restrict__ losses, float* __restrict__ g_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red64[4];
    __shared__ float red32[4];
    __shared__ int last_flag;
    __shared__ float dbuf[256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int h = a.h, w = a.w, tid = threadIdx.x;
    const int64_t P = (int64_t)h * w;
    float* gcol = reinterpret_cast<float*>(smem);
    float* grow = gcol + w;
    int* carg = reinterpret_cast<int*>(grow + h);
    int* rarg = carg + w;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;
    const InstBox ib = inst_box(a, n, dil);

    float dense_scale, sparse_scale;
    if (MODE == 0) {
        const int T = (h + kTR - 1) / kTR;
        // column maxima: reduce the per-tile partials in tile order (first row wins ties)
        float ix = 0.f, ux = 0.f, iy = 0.f, uy = 0.f;
        for (int c = tid; c < w; c += 256) {
            float m = -INFINITY; int mr = 0;
            for (int t = 0; t < T; ++t) {
                const int64_t o = ((int64_t)n * T + t) * w + c;
                const float v = ws.colv[o];
                if (v > m) { m = v; mr = t * kTR + ws.colr[o]; }
            }
            const float X = sigmoid_acc(m);
            const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
            gcol[c] = X; carg[c] = mr;
            ix += X * TX; ux += X * X + TX * TX;
        }
        for (int r = tid; r < h; r += 256) {
            const unsigned long long k = ws.rowkey[(int64_t)n * h + r];
            const float Y = sigmoid_acc(unpack_val(k));
            const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
            grow[r] = Y; rarg[r] = (int)unpack_idx(k);
            iy += Y * TY; uy += Y * Y + TY * TY;
        }
        const float Ix = block_sum_f32(ix, red32), Ux = block_sum_f32(ux, red32) + 1e-5f;
        const float Iy = block_sum_f32(iy, red32), Uy = block_sum_f32(uy, red32) + 1e-5f;
        // dice = 1 - 2I/U ; d dice/d u_j = (-2 t_j U + 4 I u_j) / U^2 ; chain through sigmoid ; mean over N
        const float invN = 1.f / (float)a.N;
        for (int c = tid; c < w; c += 256) {
            const float X = gcol[c];
            const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
            gcol[c] = invN * ((-2.f * TX * Ux + 4.f * Ix * X) / (Ux * Ux)) * X * (1.f - X);
        }
        for (int r = tid; r < h; r += 256) {
            const float Y = grow[r];
            const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
            grow[r] = invN * ((-2.f * TY * Uy + 4.f * Iy * Y) / (Uy * Uy)) * Y * (1.f - Y);
        }
        // sum of weights over the whole batch (integer, exact)
        double cnt = 0.0;                                      // exact: integers well below 2^53
        for (int i = tid; i < a.N * T; i += 256) cnt += (double)ws.part_cnt[i];
        const float total = (float)block_sum_f64(cnt, red64);  // weights.sum() is an f32 in the reference
        const float denom = fmaxf(total, 1.f);                 // .clamp(min=1.0), :1328
        dense_scale = warmup / denom;
        sparse_scale = 1.f;
        __syncthreads();
        if (s == 0) {
            if (st.colarg) {
                for (int c = tid; c < w; c += 256) { st.colarg[(int64_t)n * w + c] = carg[c]; st.gcol[(int64_t)n * w + c] = gcol[c]; }
                for (int r = tid; r < h; r += 256) { st.rowarg[(int64_t)n * h + r] = rarg[r]; st.grow[(int64_t)n * h + r] = grow[r]; }
            }
            // loss scalars: the last-arriving instance workgroup sums in index order (deterministic)
            double numd = 0.0;
            if (n == 0) for (int i = tid; i < a.N * T; i += 256) numd += (double)ws.part_num[i];
            const double numt = block_sum_f64(numd, red64);
            if (tid == 0) {
                const float dice = (1.f - 2.f * Ix / Ux) + (1.f - 2.f * Iy / Uy);
                __hip_atomic_store(&ws.dice[n], dice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n == 0) __hip_atomic_store(&losses[1], (float)(numt / (double)denom) * warmup, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned int old = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_flag = (old == (unsigned int)a.N - 1u);
            }
            __syncthreads();
            if (last_flag) {
                if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                float acc = 0.f;
                for (int base = 0; base < a.N; base += 256) {
                    if (base + tid < a.N)
                        dbuf[tid] = __hip_atomic_load(&ws.dice[base + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    if (tid == 0)
                        for (int i = 0; i < min(256, a.N - base); ++i) acc += dbuf[i];
                    __syncthreads();
                }
                if (tid == 0) losses[0] = acc / (float)a.N;   // .mean(), :143
            }
        }
    } else {
        const float gp = *up_prj, gw = *up_pw;
        if (gp == 1.f && gw == 1.f) return;   // mmdet's _parse_losses sum: nothing to do
        for (int c = tid; c < w; c += 256) { carg[c] = st.colarg[(int64_t)n * w + c]; gcol[c] = st.gcol[(int64_t)n * w + c]; }
        for (int r = tid; r < h; r += 256) { rarg[r] = st.rowarg[(int64_t)n * h + r]; grow[r] = st.grow[(int64_t)n * h + r]; }
        dense_scale = gw;          // G <- gw*G + (gp-gw)*prj   (G currently = 1*pw + 1*prj)
        sparse_scale = gp - gw;
        __syncthreads();
    }
    if (!G) return;
    __syncthreads();

    // ---- dense pass over this slice of the dilated box --------------------------------------------
    const float sp_out = MODE == 0 ? 1.f : (sparse_scale + dense_scale);   // value outside the box: gp*prj
    if (ib.any) {
        const int rows = ib.dil.r1 - ib.dil.r0;
        const int per = (rows + gridDim.x - 1) / gridDim.x;
        const int ra = ib.dil.r0 + s * per, rb = min(ib.dil.r1, ra + per);
        const int cw = ib.dil.c1 - ib.dil.c0;
        const int npx = (rb - ra) * cw;
        for (int i = tid; i < npx; i += 256) {
            const int r = ra + i / cw, c = ib.dil.c0 + i % cw;
            float v = G[(int64_t)r * w + c] * dense_scale;
            float sp = 0.f;
            if (carg[c] == r) sp += gcol[c];
            if (rarg[r] == c) sp += grow[r];
            G[(int64_t)r * w + c] = v + sp * sparse_scale;
        }
    }
    // ---- sparse pass: arg-max positions outside the dilated box (the rest of the map is zero) ----
    if (s == 0) {
        for (int c = tid; c < w; c += 256) {
            const int r = carg[c];
            const bool in_d = ib.any && r >= ib.dil.r0 && r < ib.dil.r1 && c >= ib.dil.c0 && c < ib.dil.c1;
            if (!in_d) {
                float v = gcol[c];
                if (rarg[r] == c) v += grow[r];
                G[(int64_t)r * w + c] = v * sp_out;
            }
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg[r];
            const bool in_d = ib.any && r >= ib.dil.r0 && r < ib.dil.r1 && c >= ib.dil.c0 && c < ib.dil.c1;
            if (!in_d && carg[c] != r) G[(int64_t)r * w + c] = grow[r] * sp_out;
        }
    }
}

__global__ void zero_losses_kernel(float* losses) { losses[0] = 0.f; losses[1] = 0.f; }

// ---- host side ---------------------------------------------------------------------------------
int fill_gt_table(const float* const* boxes_per_img_host, const int* gt_count_host, int B, GtTable& gt, int& G);

static int fill_inst(const bxi_instances* in, InstArgs& a) {
    if (!in) return BXI_ERR_NULL_POINTER;
    if (in->N < 0 || in->h <= 0 || in->w <= 0 || in->stride < 1) return BXI_ERR_BAD_SHAPE;
    if (in->Hc != in->h * in->stride || in->Wc != in->w * in->stride) return BXI_ERR_BAD_SHAPE;
    if (!fits_i32((int64_t)in->N * in->h * in->w)) return BXI_ERR_BAD_SHAPE;
    int G = 0;
    int st = fill_gt_table(in->boxes_per_img_host, in->gt_count_host, in->B, a.gt, G);
    if (st != BXI_OK) return st;
    a.logits = in->logits; a.gt_inds = in->gt_inds;
    a.N = in->N; a.h = in->h; a.w = in->w; a.Hc = in->Hc; a.Wc = in->Wc; a.stride = in->stride;
    if (in->N > 0 && (!in->logits || !in->gt_inds)) return BXI_ERR_NULL_POINTER;
    return BXI_OK;
}

static size_t main_lds_bytes(int w, int dil) {
    const size_t HR = kTR + 2 * dil;
    return sizeof(float2) * HR * w + sizeof(float) * kTR * w + sizeof(float) * 4 * kChunk + sizeof(int) * 4 * kChunk +
           HR * w;
}

int launch_loss(const bxi_instances* in, const uint8_t* affinity, int size, int dil, float warmup, float* losses,
                float* g_logits, void* state, void* workspace, size_t workspace_bytes, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || dil > kMaxDil) return BXI_ERR_UNSUPPORTED;
    if (!losses) return BXI_ERR_NULL_POINTER;
    hipStream_t s = as_stream(stream);
    if (a.N == 0) {
        hipLaunchKernelGGL(zero_losses_kernel, dim3(1), dim3(1), 0, s, losses);
        return check_launch();
    }
    if (!affinity) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    const size_t need = carve_ws(nullptr, a.N, a.h, a.w, nullptr);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    LossWs ws;
    carve_ws(workspace, a.N, a.h, a.w, &ws);
    LossState st = {nullptr, nullptr, nullptr, nullptr};
    if (state) {
        if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
        carve_state(state, a.N, a.h, a.w, &st);
    }
    const size_t lds = main_lds_bytes(a.w, dil);
    if (lds > 160 * 1024) return BXI_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(loss_main_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    const int vec = ((a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0 &&
                     (!g_logits || (reinterpret_cast<uintptr_t>(g_logits) & 15) == 0)) ? 1 : 0;
    const int T = tiles_of(a.h);
    hipLaunchKernelGGL(loss_main_kernel, dim3((unsigned)(a.N * T)), dim3(256), lds, s, a, affinity, dil, ws, g_logits, vec);
    rc = check_launch();
    if (rc != BXI_OK) return rc;
    const size_t lds_d = (sizeof(float) + sizeof(int)) * (size_t)(a.h + a.w);
    hipLaunchKernelGGL((loss_finalize_kernel<0>), dim3(kSlices, a.N), dim3(256), lds_d, s, a, dil, warmup, ws, st,
                       (const float*)nullptr, (const float*)nullptr, losses, g_logits);
    return check_launch();
}

int launch_rescale(const bxi_instances* in, const float* g_prj, const float* g_pw, int dil, const void* state,
                   float* g_logits, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (dil < 1 || dil > kMaxDil) return BXI_ERR_BAD_ARGUMENT;
    if (a.N == 0) return BXI_OK;
    if (!g_prj || !g_pw || !state || !g_logits) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    LossState st;
    carve_state(const_cast<void*>(state), a.N, a.h, a.w, &st);
    LossWs ws = {};
    const size_t lds_d = (sizeof(float) + sizeof(int)) * (size_t)(a.h + a.w);
    hipLaunchKernelGGL((loss_finalize_kernel<1>), dim3(kSlices, a.N), dim3(256), lds_d, as_stream(stream), a, dil, 1.f,
                       ws, st, g_prj, g_pw, (float*)nullptr, g_logits);
    return check_launch();
}

size_t loss_ws_bytes(int N, int h, int w) { return carve_ws(nullptr, N, h, w, nullptr); }
size_t loss_state_bytes(int N, int h, int w) { return carve_state(nullptr, N, h, w, nullptr); }

}  // namespace bxi

extern "C" {

size_t bxi_boxinst_loss_workspace_bytes(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    return bxi::loss_ws_bytes(N, h, w);
}
size_t bxi_boxinst_loss_state_bytes(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    return bxi::loss_state_bytes(N, h, w);
}

int bxi_boxinst_loss_fwd_bwd_f32(const bxi_instances* inst_host, const uint8_t* affinity, int size, int dilation,
                                 float warmup, float* losses, float* g_logits, void* state, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return bxi::launch_loss(inst_host, affinity, size, dilation, warmup, losses, g_logits, state, workspace,
                            workspace_bytes, stream);
}

int bxi_boxinst_loss_rescale_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw, int dilation,
                                 const void* state, float* g_logits, void* stream) {
    return bxi::launch_rescale(inst_host, g_prj, g_pw, dilation, state, g_logits, stream);
}

}  // extern "C"
