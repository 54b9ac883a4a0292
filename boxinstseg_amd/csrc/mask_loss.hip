// mask_loss.hip -- BoxInst projection + pairwise loss, forward AND backward, on gfx950.
//
// Replaces (reference, LiWentomng/BoxInstSeg) CondInstMaskHead.loss with boxinst_enabled,
// condinst_head.py:1288-1343:
//   mask_logits.sigmoid() / compute_project_term      :1300, :134-143 (+ dice_coefficient :117-131)
//   pairwise_nlog (CUDA op, pairwise.cu:68-149)       :1321
//   weights = (sim >= thresh) * bitmask, normalise    :1324-1328
//   warm-up                                           :1330-1332
// and everything autograd does behind them (max backward, sigmoid backward, the op's atomicAdd
// backward, ~1.3 GB of elementwise temporaries at 2x800x1024x32) with three launches, each
// documented at its kernel below:
//   stage1        { image pool + Lab } || { logit streaming: row/column maxima, zero-fill }   HBM stream
//   box_kernel    colour-affinity bits + pairwise term and its gradient on the box tiles     latency bound
//   loss_finalize dice, projection gradient at the arg-max positions, normalisation, scalars latency bound
// Data layout in HBM: everything NCHW / row-major as the reference; per-pixel colour affinity is
// never materialised as [N,8,h,w] -- box_kernel derives the 8-bit word it needs from Lab [B,3,h,w].
#include "image_device.hpp"

namespace bxi {

constexpr int kSR = 16;         // rows per streaming tile (stage1): 4 rows per wave
constexpr int kRW = kSR / 4;
constexpr int kChunk = 256;     // columns per pass: 64 lanes x float4
constexpr int kBR = 8;          // box tile rows    (box_kernel)
constexpr int kBC = 64;         // box tile columns
constexpr int kSlices = 4;      // row slices per instance in loss_finalize
constexpr int kMaxDil = 8;
constexpr int kMaxT = 32;       // per-instance column partials reduced per unrolled batch in loss_finalize

struct InstArgs {
    const float* logits;
    const int64_t* gt_inds;
    int N, h, w;
    int Hc, Wc, stride;
    GtTable gt;
};

struct LossWs {               // carved from the caller's workspace
    float* colv;              // [N,Ts,w] per-streaming-tile column max (logit)
    uint8_t* colr;            // [N,Ts,w] row offset of that max inside the tile
    unsigned long long* rowkey;  // [N,h] packed (max logit, first column)
    float* part_num;          // [N*Tr*Tc] per box tile: sum of W*pw
    int* part_cnt;            // [N*Tr*Tc] per box tile: sum of W
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

static inline int stream_tiles(int h) { return (h + kSR - 1) / kSR; }
static inline int box_tiles(int h, int w) { return ((h + kBR - 1) / kBR) * ((w + kBC - 1) / kBC); }

static size_t carve_ws(void* base, int N, int h, int w, LossWs* ws) {
    const size_t T = (size_t)stream_tiles(h);
    const size_t TB = (size_t)box_tiles(h, w);
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    float* colv = (float*)take(sizeof(float) * N * T * w);
    uint8_t* colr = (uint8_t*)take((size_t)N * T * w);
    unsigned long long* rowkey = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)N * h);
    float* part_num = (float*)take(sizeof(float) * N * TB);
    int* part_cnt = (int*)take(sizeof(int) * N * TB);
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

// ================================================================================================
// Kernel 1: stage1 = { pool_rgb + Lab workgroups }  ||  { logit streaming workgroups }
// ================================================================================================
// The two halves are independent (image side / logit side), so they share one launch: the grid is
// n_pool workgroups of image work followed by N*Ts streaming workgroups, all resident at once.
// Streaming workgroup (n, tile of kSR rows), 4 wave64, each wave owns kRW rows, a lane owns 4
// consecutive columns (float4, a wave = 1 KiB contiguous):
//   - issues all its row loads first, then (scalar path) looks the instance's box up;
//   - row max / first arg-max: lane-local, then a 64-lane shuffle tree on a packed 64-bit key;
//   - column max / first arg-max over the tile's rows: registers, then 4 waves through LDS,
//     one partial per (tile, column) -- no atomics;
//   - zero-fills d loss / d logits wherever no box tile of box_kernel will write.
__device__ __forceinline__ bool seg_hit(const InstBox& ib, int r, int c) {
    const int tr = r & ~(kBR - 1), tc = c & ~(kBC - 1);
    return ib.any && tr < ib.dil.r1 && tr + kBR > ib.dil.r0 && tc < ib.dil.c1 && tc + kBC > ib.dil.c0;
}

__device__ __forceinline__ void stream_tile(const InstArgs& a, int dil, const LossWs& ws, float* __restrict__ g_logits,
                                            int vec, int sb, float* cbv, int* cbr) {
    const int h = a.h, w = a.w;
    const int Ts = (h + kSR - 1) / kSR;
    const int n = sb / Ts, t = sb % Ts;
    const int r0 = t * kSR, r1 = min(h, r0 + kSR);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int64_t P = (int64_t)h * w;
    const float* L = a.logits + (int64_t)n * P;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);

    float4 v[kRW];
    {
        const int c = lane * 4;
#pragma unroll
        for (int i = 0; i < kRW; ++i) {
            const int r = r0 + wv + 4 * i;
            v[i] = (r < r1 && c < w) ? load4(L + (int64_t)r * w, c, w, vec) : ninf;
        }
    }
    const InstBox ib = inst_box(a, n, dil);   // two dependent scalar loads, overlapped with the row loads

    unsigned long long rkey[kRW];
#pragma unroll
    for (int i = 0; i < kRW; ++i) rkey[i] = 0ull;

    for (int cb = 0;;) {
        const int c = cb + lane * 4;
        float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int crow[4] = {r0, r0, r0, r0};
        if (c < w) {
#pragma unroll
            for (int i = 0; i < kRW; ++i) {
                const int r = r0 + wv + 4 * i;
                if (r < r1) {
                    float m = v[i].x; int mc = c;                       // first column wins ties
                    if (v[i].y > m) { m = v[i].y; mc = c + 1; }
                    if (v[i].z > m) { m = v[i].z; mc = c + 2; }
                    if (v[i].w > m) { m = v[i].w; mc = c + 3; }
                    const unsigned long long key = pack_max(m, (uint32_t)mc);
                    rkey[i] = key > rkey[i] ? key : rkey[i];
                    if (v[i].x > cmax[0]) { cmax[0] = v[i].x; crow[0] = r; }   // ascending r, strict >
                    if (v[i].y > cmax[1]) { cmax[1] = v[i].y; crow[1] = r; }
                    if (v[i].z > cmax[2]) { cmax[2] = v[i].z; crow[2] = r; }
                    if (v[i].w > cmax[3]) { cmax[3] = v[i].w; crow[3] = r; }
                    if (G && !seg_hit(ib, r, c)) store4(G + (int64_t)r * w, c, w, vec, make_float4(0.f, 0.f, 0.f, 0.f));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { cbv[wv * kChunk + lane * 4 + j] = cmax[j]; cbr[wv * kChunk + lane * 4 + j] = crow[j]; }
        __syncthreads();
        {
            const int cc = cb + tid;
            if (cc < w) {
                float m = cbv[tid]; int mr = cbr[tid];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float x = cbv[q * kChunk + tid]; const int xr = cbr[q * kChunk + tid];
                    if (x > m || (x == m && xr < mr)) { m = x; mr = xr; }
                }
                const int64_t o = ((int64_t)n * Ts + t) * w + cc;
                ws.colv[o] = m;
                ws.colr[o] = (uint8_t)(mr - r0);
            }
        }
        cb += kChunk;
        if (cb >= w) break;
        __syncthreads();
        {
            const int c2 = cb + lane * 4;
#pragma unroll
            for (int i = 0; i < kRW; ++i) {
                const int r = r0 + wv + 4 * i;
                v[i] = (r < r1 && c2 < w) ? load4(L + (int64_t)r * w, c2, w, vec) : ninf;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kRW; ++i) {
        const int r = r0 + wv + 4 * i;
        const unsigned long long k = wave_max_u64(rkey[i]);
        if (lane == 0 && r < r1) ws.rowkey[(int64_t)n * h + r] = k;
    }
    if (sb == 0 && tid == 0) *ws.ticket = 0u;   // consumed by loss_finalize two kernel boundaries later
}

__global__ __launch_bounds__(256) void stage1_kernel(PoolArgs pa, int n_pool, InstArgs a, int dil, LossWs ws,
                                                     float* __restrict__ g_logits, int vec) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[sizeof(float) * 4 * kChunk + sizeof(int) * 4 * kChunk];
    if ((int)blockIdx.x < n_pool) {
        double* lut = reinterpret_cast<double*>(sm);
        lut[threadIdx.x] = kSrgbLut[threadIdx.x];
        const int64_t total = (int64_t)pa.B * (pa.Hc >> 2) * (pa.Wc >> 2);
        const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
        __syncthreads();
        if (o < total) pool_pixel_s4(pa, o, lut);
    } else {
        float* cbv = reinterpret_cast<float*>(sm);
        int* cbr = reinterpret_cast<int*>(cbv + 4 * kChunk);
        stream_tile(a, dil, ws, g_logits, vec, (int)blockIdx.x - n_pool, cbv, cbr);
    }
}

// ================================================================================================
// Kernel 2: box_kernel -- pairwise term on the instance's (dilated) box, 8 x 64 pixel tiles
// ================================================================================================
// grid = N x ceil(h/8) x ceil(w/64); a workgroup whose tile misses the dilated box writes two
// zero partials and exits (~85 % of them).  Otherwise:
//   LDS  pq   [8+2d][64+2P]  (sigmoid(x), sigmoid(-x)) of the tile + halo            (P = d rounded up to 4)
//        bits [8+2d][64+2P]  K-bit colour-affinity word of every in-box pixel of that region
//        lab  [3][8+4d][64+2P2]  CIE-Lab of the tile + 2d halo (FROM_LAB)            (P2 = 2d rounded up to 4)
//   1. all global loads are issued before the first LDS store (float4, aligned);
//   2. affinity words from Lab (same arithmetic as affinity_kernel) or copied from `bits_in`;
//   3. per pixel, 8 neighbours: S = p_i p_j + q_i q_j, -log S and its gradient, weighted by
//      bit k of the pixel + bit 7-k of the neighbour (gather form: no atomics, fixed order);
//   4. writes the UN-normalised pairwise gradient of the whole tile (zeros outside the dilated box).
template <bool FROM_LAB>
__global__ __launch_bounds__(256) void box_kernel(InstArgs a, const float* __restrict__ lab, ImageMeta meta,
                                                  const uint8_t* __restrict__ bits_in, float thresh, int dil, LossWs ws,
                                                  float* __restrict__ g_logits, int vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float rnum[4];
    __shared__ int rcnt[4];
    const int h = a.h, w = a.w;
    const int Tr = (h + kBR - 1) / kBR, Tc = (w + kBC - 1) / kBC;
    int bi = blockIdx.x;
    const int tcx = bi % Tc; bi /= Tc;
    const int trx = bi % Tr;
    const int n = bi / Tr;
    const int r0 = trx * kBR, c0 = tcx * kBC;
    const int tid = threadIdx.x;

    const InstBox ib = inst_box(a, n, dil);
    const bool hit = ib.any && r0 < ib.dil.r1 && r0 + kBR > ib.dil.r0 && c0 < ib.dil.c1 && c0 + kBC > ib.dil.c0;
    if (!hit) {   // workgroup-uniform
        if (tid == 0) { ws.part_num[blockIdx.x] = 0.f; ws.part_cnt[blockIdx.x] = 0; }
        return;
    }
    const int64_t P = (int64_t)h * w;
    const float* L = a.logits + (int64_t)n * P;
    const int d = dil;
    const int PAD = (d + 3) & ~3, PAD2 = (2 * d + 3) & ~3;
    const int PR = kBR + 2 * d, PC = kBC + 2 * PAD;          // pq / bits region
    const int LR = kBR + 4 * d, LC = kBC + 2 * PAD2;         // lab region
    float2* pq = reinterpret_cast<float2*>(smem);
    float* labs = reinterpret_cast<float*>(smem + sizeof(float2) * (size_t)PR * PC);
    uint8_t* bits = smem + sizeof(float2) * (size_t)PR * PC + (FROM_LAB ? sizeof(float) * 3 * (size_t)LR * LC : 0);

    // ---- 1. loads: logits region -> sigmoid pairs; Lab region (or bits) ---------------------------
    {
        const int q4 = PC / 4, items = PR * q4;
        for (int base = tid; base < items; base += 256 * 2) {
            float4 tmp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = base + u * 256;
                tmp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < items) {
                    const int lr = i / q4, r = r0 - d + lr, c = c0 - PAD + (i % q4) * 4;
                    if (r >= 0 && r < h && c >= 0 && c < w) tmp[u] = load4(L + (int64_t)r * w, c, w, vec);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = base + u * 256;
                if (i < items) {
                    float2* dst = pq + (size_t)(i / q4) * PC + (i % q4) * 4;
                    dst[0] = sig_pair(tmp[u].x); dst[1] = sig_pair(tmp[u].y);
                    dst[2] = sig_pair(tmp[u].z); dst[3] = sig_pair(tmp[u].w);
                }
            }
        }
    }
    if (FROM_LAB) {
        const float* LB = lab + (int64_t)ib.img * 3 * P;
        const int q4 = LC / 4, per = LR * q4, items = 3 * per;
        for (int base = tid; base < items; base += 256 * 4) {
            float4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256;
                tmp[u] = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding of F.unfold
                if (i < items) {
                    const int ch = i / per, j = i % per;
                    const int r = r0 - 2 * d + j / q4, c = c0 - PAD2 + (j % q4) * 4;
                    if (r >= 0 && r < h && c >= 0 && c < w) {
                        const float* row = LB + ch * P + (int64_t)r * w;
                        if (vec) tmp[u] = *reinterpret_cast<const float4*>(row + c);
                        else {
                            tmp[u].x = row[c];
                            tmp[u].y = c + 1 < w ? row[c + 1] : 0.f;
                            tmp[u].z = c + 2 < w ? row[c + 2] : 0.f;
                            tmp[u].w = c + 3 < w ? row[c + 3] : 0.f;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256;
                if (i < items) *reinterpret_cast<float4*>(labs + (size_t)i * 4) = tmp[u];
            }
        }
    }
    __syncthreads();

    // ---- 2. affinity words of the in-box pixels of the pq region -----------------------------------
    {
        const int bw = kBC + 2 * d, items = PR * bw;
        const uint8_t* AF = FROM_LAB ? nullptr : bits_in + (int64_t)ib.img * P;
        for (int i = tid; i < items; i += 256) {
            const int lr = i / bw, lc = i % bw;
            const int r = r0 - d + lr, c = c0 - d + lc;
            uint32_t word = 0;
            if (r >= ib.box.r0 && r < ib.box.r1 && c >= ib.box.c0 && c < ib.box.c1) {   // bitmask == 1, :1324-1325
                if (FROM_LAB) {
                    const int li = (lr + d) * LC + (lc - d + PAD2);     // same pixel in the lab region
                    const float L0 = labs[li], A0 = labs[LR * LC + li], B0 = labs[2 * LR * LC + li];
                    int k = 0;
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (dx == 0 && dy == 0) continue;
                            const int r2 = r + dy * d, c2 = c + dx * d;
                            float s = 0.f;                                // zero-padded mask => 0
                            if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w) {
                                const int qi = li + dy * d * LC + dx * d;
                                s = color_sim(L0, A0, B0, labs[qi], labs[LR * LC + qi], labs[2 * LR * LC + qi],
                                              geom_mask(meta, ib.img, r2, c2, a.stride));
                            }
                            word |= (s >= thresh ? 1u : 0u) << k;
                            ++k;
                        }
                } else {
                    word = AF[(int64_t)r * w + c];
                }
            }
            bits[(size_t)lr * PC + (lc - d + PAD)] = (uint8_t)word;
        }
    }
    __syncthreads();

    // ---- 3. pairwise term, 2 pixels per thread -------------------------------------------------------
    const int lr = tid >> 5, lc = (tid & 31) * 2;
    const int r = r0 + lr;
    float num = 0.f;
    int cnt = 0;
    float out[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = c0 + lc + e;
        if (r < ib.dil.r0 || r >= ib.dil.r1 || c < ib.dil.c0 || c >= ib.dil.c1) continue;
        const int pi = (lr + d) * PC + (lc + e + PAD);
        const float2 pp = pq[pi];
        const uint32_t bits_p = bits[pi];
        float acc = 0.f;
        int k = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0) continue;
                const int r2 = r + dy * d, c2 = c + dx * d;
                const uint32_t wp = (bits_p >> k) & 1u;
                cnt += (int)wp;                                          // weights.sum(), :1328
                if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w) {
                    const int qi = pi + dy * d * PC + dx * d;
                    const uint32_t wq = ((uint32_t)bits[qi] >> (7 - k)) & 1u;
                    const uint32_t ws2 = wp + wq;
                    if (ws2) {
                        const float2 qq = pq[qi];
                        const float S = pp.x * qq.x + pp.y * qq.y;      // P(y_p == y_q)
                        float nl, coef;
                        if (S > 1e-30f) {
                            nl = -__logf(S);
                            coef = -(qq.x - qq.y) * (pp.x * pp.y) * __frcp_rn(S);
                        } else {   // |logit| beyond ~69: log-space evaluation as pairwise.cu:38-61
                            const float xa = L[(int64_t)r * w + c], xb = L[(int64_t)r2 * w + c2];
                            const float ax = logsig(xa), bx = logsig(-xa), ay = logsig(xb), by = logsig(-xb);
                            const float e1 = ax + ay, e0 = bx + by;
                            nl = logsig(fabsf(e1 - e0)) - fmaxf(e1, e0);
                            coef = -(expf(ay) - expf(by)) * expf(ax + bx + nl);
                        }
                        num += (float)wp * nl;
                        acc += (float)ws2 * coef;
                    }
                }
                ++k;
            }
        out[e] = acc;
    }
    if (g_logits && r < h) {
        float* G = g_logits + (int64_t)n * P + (int64_t)r * w + c0 + lc;
        if (vec) {
            if (c0 + lc < w) *reinterpret_cast<float2*>(G) = make_float2(out[0], out[1]);
        } else {
            if (c0 + lc < w) G[0] = out[0];
            if (c0 + lc + 1 < w) G[1] = out[1];
        }
    }
    // ---- block partials (fixed order) -------------------------------------------------------------------
    num = wave_sum_f32(num);
    cnt = wave_sum_i32(cnt);
    if ((tid & 63) == 0) { rnum[tid >> 6] = num; rcnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        ws.part_num[blockIdx.x] = (rnum[0] + rnum[1]) + (rnum[2] + rnum[3]);
        ws.part_cnt[blockIdx.x] = rcnt[0] + rcnt[1] + rcnt[2] + rcnt[3];
    }
}

// ================================================================================================
// Kernel 3: loss_finalize (MODE 0) / loss_rescale (MODE 1), grid = kSlices x N
// ================================================================================================
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

// MODE 0: dice per instance from the h+w maxima, unit projection gradients at the arg-max
//         positions, sum W -> normaliser, in-place rescale of the dilated-box region, loss scalars
//         (ticket: the last-arriving instance sums the N dice values in index order).
// MODE 1: fold arbitrary upstream gradients (device scalars) into the unit gradient; all
//         workgroups exit at once when both are exactly 1.
// Every global load of a phase is issued before its first use (the kernel is latency-, not
// bandwidth-bound: it touches ~1 MB).
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

    float dense_scale, sparse_scale;
    InstBox ib;
    unsigned int ticket_old = 0;
    if (MODE == 0) {
        const int Ts = (h + kSR - 1) / kSR;
        const int NB = a.N * ((h + kBR - 1) / kBR) * ((w + kBC - 1) / kBC);
        // ---- round 1: every load this phase needs, back to back ------------------------------------
        double cnt = 0.0, numd = 0.0;
        for (int base = tid; base < NB; base += 256 * 8) {
            int pc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = base + u * 256; pc[u] = i < NB ? ws.part_cnt[i] : 0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt += (double)pc[u];            // exact (integers < 2^53)
        }
        if (s == 0 && n == 0)
            for (int base = tid; base < NB; base += 256 * 8) {
                float pn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = base + u * 256; pn[u] = i < NB ? ws.part_num[i] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) numd += (double)pn[u];
            }
        ib = inst_box(a, n, dil);
        float ix = 0.f, ux = 0.f, iy = 0.f, uy = 0.f;
        for (int c = tid; c < w; c += 256) {
            float m = -INFINITY; int mr = 0;
            for (int t0 = 0; t0 < Ts; t0 += kMaxT) {
                float cv[kMaxT]; uint8_t cr[kMaxT];
#pragma unroll
                for (int u = 0; u < kMaxT; ++u) {
                    const int t = t0 + u;
                    const int64_t o = ((int64_t)n * Ts + (t < Ts ? t : 0)) * w + c;
                    cv[u] = t < Ts ? ws.colv[o] : -INFINITY;
                    cr[u] = t < Ts ? ws.colr[o] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < kMaxT; ++u)
                    if (cv[u] > m) { m = cv[u]; mr = (t0 + u) * kSR + cr[u]; }   // tile order: first row wins ties
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
        const float total = (float)block_sum_f64(cnt, red64);  // weights.sum() is an f32 in the reference
        const float denom = fmaxf(total, 1.f);                 // .clamp(min=1.0), :1328
        if (s == 0) {   // publish this instance's dice early: the ticket's round trip overlaps the passes below
            const double numt = n == 0 ? block_sum_f64(numd, red64) : 0.0;
            if (tid == 0) {
                const float dice = (1.f - 2.f * Ix / Ux) + (1.f - 2.f * Iy / Uy);
                __hip_atomic_store(&ws.dice[n], dice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
                if (n == 0) losses[1] = (float)(numt / (double)denom) * warmup;                      // :1327-1332
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ticket_old = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
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
        dense_scale = warmup / denom;
        sparse_scale = 1.f;
        __syncthreads();
        if (s == 0 && st.colarg) {
            for (int c = tid; c < w; c += 256) { st.colarg[(int64_t)n * w + c] = carg[c]; st.gcol[(int64_t)n * w + c] = gcol[c]; }
            for (int r = tid; r < h; r += 256) { st.rowarg[(int64_t)n * h + r] = rarg[r]; st.grow[(int64_t)n * h + r] = grow[r]; }
        }
    } else {
        const float gp = *up_prj, gw = *up_pw;
        if (gp == 1.f && gw == 1.f) return;   // mmdet's _parse_losses sum: nothing to do
        ib = inst_box(a, n, dil);
        for (int c = tid; c < w; c += 256) { carg[c] = st.colarg[(int64_t)n * w + c]; gcol[c] = st.gcol[(int64_t)n * w + c]; }
        for (int r = tid; r < h; r += 256) { rarg[r] = st.rowarg[(int64_t)n * h + r]; grow[r] = st.grow[(int64_t)n * h + r]; }
        dense_scale = gw;          // G <- gw*G + (gp-gw)*prj   (G currently = 1*pw + 1*prj)
        sparse_scale = gp - gw;
        __syncthreads();
    }

    if (G) {
        // ---- dense pass over this slice of the dilated box (8 loads in flight per thread) -------------
        const float sp_out = MODE == 0 ? 1.f : (sparse_scale + dense_scale);   // outside the box: gp * prj
        if (ib.any) {
            const int rows = ib.dil.r1 - ib.dil.r0;
            const int per = (rows + gridDim.x - 1) / gridDim.x;
            const int ra = ib.dil.r0 + s * per, rb = min(ib.dil.r1, ra + per);
            const int cw = ib.dil.c1 - ib.dil.c0;
            const int npx = (rb - ra) * cw;
            for (int base = tid; base < npx; base += 256 * 8) {
                float gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * 256;
                    gv[u] = i < npx ? G[(int64_t)(ra + i / cw) * w + ib.dil.c0 + i % cw] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * 256;
                    if (i < npx) {
                        const int r = ra + i / cw, c = ib.dil.c0 + i % cw;
                        float sp = 0.f;
                        if (carg[c] == r) sp += gcol[c];
                        if (rarg[r] == c) sp += grow[r];
                        G[(int64_t)r * w + c] = gv[u] * dense_scale + sp * sparse_scale;
                    }
                }
            }
        }
        // ---- sparse pass: arg-max positions outside the dilated box (the rest of the map is zero) ------
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

    if (MODE == 0 && s == 0) {
        // ---- loss_prj: the last-arriving instance sums dice[0..N) in index order (deterministic) ---------
        if (tid == 0) last_flag = (ticket_old == (unsigned int)a.N - 1u);
        __syncthreads();
        if (last_flag) {   // dice[] was stored write-through (sc1) and is read with agent-scope loads
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

static size_t box_lds_bytes(int dil, bool from_lab) {
    const size_t PAD = (dil + 3) & ~3, PAD2 = (2 * dil + 3) & ~3;
    const size_t PR = kBR + 2 * dil, PC = kBC + 2 * PAD, LR = kBR + 4 * dil, LC = kBC + 2 * PAD2;
    return sizeof(float2) * PR * PC + (from_lab ? sizeof(float) * 3 * LR * LC : 0) + PR * PC;
}

int fill_pool_args(const bxi_image_batch* bt, uint8_t* rgb_small, float* lab, PoolArgs& pa);
int fill_image_meta(const bxi_image_batch* bt, ImageMeta& meta, Denorm& dn);
bool pool_vec_ok(const bxi_image_batch* bt, int stride);
int launch_pool(const bxi_image_batch* bt, int stride, uint8_t* rgb_small, float* lab, hipStream_t s);

// One evaluation.  batch != NULL: image side included (lab is a [B,3,h,w] f32 scratch the pool
// workgroups fill and box_kernel reads).  batch == NULL: `affinity` bits are given.
int launch_loss(const bxi_image_batch* batch, float* lab, float color_thresh, const bxi_instances* in,
                const uint8_t* affinity, int size, int dil, float warmup, float* losses, float* g_logits, void* state,
                void* workspace, size_t workspace_bytes, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || dil > kMaxDil) return BXI_ERR_UNSUPPORTED;
    if (!losses) return BXI_ERR_NULL_POINTER;
    hipStream_t s = as_stream(stream);
    PoolArgs pa = {};
    ImageMeta meta = {};
    const bool from_lab = batch != nullptr;
    if (from_lab) {
        if (batch->Hc != in->Hc || batch->Wc != in->Wc || batch->B != in->B) return BXI_ERR_BAD_SHAPE;
        rc = fill_pool_args(batch, nullptr, lab, pa);
        if (rc != BXI_OK) return rc;
        meta = pa.meta;
        if (batch->B > 0 && (!batch->imgs || !lab)) return BXI_ERR_NULL_POINTER;
        if (batch->image_masks) return BXI_ERR_UNSUPPORTED;   // explicit masks: use bxi_color_affinity_f32 + bits
    }
    if (a.N == 0) {
        BXI_LAUNCH("zero_losses", s, zero_losses_kernel, dim3(1), dim3(1), 0, s, losses);
        return check_launch();
    }
    if (!from_lab && !affinity) return BXI_ERR_NULL_POINTER;
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
    const int vec = ((a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0 &&
                     (!g_logits || (reinterpret_cast<uintptr_t>(g_logits) & 15) == 0) &&
                     (!lab || (reinterpret_cast<uintptr_t>(lab) & 15) == 0)) ? 1 : 0;

    // ---- kernel 1: image pooling/Lab workgroups + logit streaming workgroups in one launch -----------
    int n_pool = 0;
    if (from_lab && batch->B > 0) {
        if (pool_vec_ok(batch, a.stride)) {
            const int64_t total = (int64_t)batch->B * a.h * a.w;
            n_pool = (int)((total + 255) / 256);
            n_pool = (n_pool + 7) & ~7;   // keep streaming tile i on XCD i % 8, like box tile i
        } else {                           // unaligned canvas / other strides: separate scalar pooling launch
            rc = launch_pool(batch, a.stride, nullptr, lab, s);
            if (rc != BXI_OK) return rc;
        }
    }
    const int n_stream = a.N * stream_tiles(a.h);
    BXI_LAUNCH("stage1", s, stage1_kernel, dim3((unsigned)(n_pool + n_stream)), dim3(256), 0, s, pa, n_pool, a, dil, ws,
               g_logits, vec);
    rc = check_launch();
    if (rc != BXI_OK) return rc;

    // ---- kernel 2: pairwise term on the box tiles --------------------------------------------------------
    const size_t lds = box_lds_bytes(dil, from_lab);
    const int n_box = a.N * box_tiles(a.h, a.w);
    if (from_lab)
        BXI_LAUNCH("box", s, (box_kernel<true>), dim3((unsigned)n_box), dim3(256), lds, s, a, (const float*)lab, meta,
                   (const uint8_t*)nullptr, color_thresh, dil, ws, g_logits, vec);
    else
        BXI_LAUNCH("box", s, (box_kernel<false>), dim3((unsigned)n_box), dim3(256), lds, s, a, (const float*)nullptr,
                   meta, affinity, 0.f, dil, ws, g_logits, vec);
    rc = check_launch();
    if (rc != BXI_OK) return rc;

    // ---- kernel 3: dice, normalisation, loss scalars -----------------------------------------------------
    const size_t lds_d = (sizeof(float) + sizeof(int)) * (size_t)(a.h + a.w);
    BXI_LAUNCH("loss_finalize", s, (loss_finalize_kernel<0>), dim3(kSlices, a.N), dim3(256), lds_d, s, a, dil, warmup,
               ws, st, (const float*)nullptr, (const float*)nullptr, losses, g_logits);
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
    BXI_LAUNCH("loss_rescale", as_stream(stream), (loss_finalize_kernel<1>), dim3(kSlices, a.N), dim3(256), lds_d,
               as_stream(stream), a, dil, 1.f, ws, st, g_prj, g_pw, (float*)nullptr, g_logits);
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
    return bxi::launch_loss(nullptr, nullptr, 0.f, inst_host, affinity, size, dilation, warmup, losses, g_logits, state,
                            workspace, workspace_bytes, stream);
}

int bxi_boxinst_loss_rescale_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw, int dilation,
                                 const void* state, float* g_logits, void* stream) {
    return bxi::launch_rescale(inst_host, g_prj, g_pw, dilation, state, g_logits, stream);
}

}  // extern "C"
