// pairwise_op.hip -- op-level drop-in for the reference's `pairwise_ext` (bind.cpp:15-36):
//   pairwise_nlog_forward  (pairwise.cu:68-104, launcher :154-175)
//   pairwise_nlog_backward (pairwise.cu:106-149, launcher :177-202)
// written for gfx950: wave64, x-fastest coalesced planes, no atomics (gather backward).
//
// Roofline: HBM.  Forward moves 4*(1+K) B per pixel (K = size^2-1 output planes, write-bound);
// backward 4*(1+K+1) B per pixel (g_pairwise read once through L2: every element is used by the
// pixel itself (channel k) and by one neighbour (channel K-1-k)).  size == 3 runs the tiled kernels
// further down (12.3 us / 14.0 us at 32x200x256 f32, cold, against 12.7 for a linear copy of the backward's bytes); the two kernels below serve
// the other window sizes.
#include "common.hpp"

namespace bxi {

// f(x,y) = -log(s(x)s(y) + s(-x)s(-y)) evaluated in log space exactly as pairwise.cu:38-50.
template <typename T>
__device__ __forceinline__ T pair_nlog(T ax, T bx, T ay, T by) {
    T e1 = ax + ay, e0 = bx + by;
    T mx = e1 > e0 ? e1 : e0;
    T df = e1 > e0 ? e1 - e0 : e0 - e1;
    return logsig(df) - mx;
}

template <typename T>
__global__ __launch_bounds__(256) void pairwise_fwd_kernel(const T* __restrict__ logits, int N, int H, int W,
                                                           int size, int dil, T* __restrict__ out) {
    const int K = size * size - 1;
    const int R = size / 2 * dil;
    const int64_t P = (int64_t)H * W;
    const int64_t total = (int64_t)N * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int64_t n = idx / P;
        const T* L = logits + n * P;
        const T here = L[(int64_t)y * W + x];
        const T ax = logsig(here), bx = logsig(-here);
        T* o = out + n * K * P + (int64_t)y * W + x;
        int k = 0;
        for (int dy = -R; dy <= R; dy += dil)
            for (int dx = -R; dx <= R; dx += dil) {
                if (dx == 0 && dy == 0) continue;
                const int x2 = x + dx, y2 = y + dy;
                T v = T(0);  // padded neighbour: log-probs 0 => pair == 0 (pairwise.cu:43-44)
                if (x2 >= 0 && x2 < W && y2 >= 0 && y2 < H) {
                    const T there = L[(int64_t)y2 * W + x2];
                    v = pair_nlog(ax, bx, logsig(there), logsig(-there));
                }
                o[(int64_t)k * P] = v;
                ++k;
            }
    }
}

// d f(a,b) / d a = -(s(b) - s(-b)) * exp(logs(a) + logs(-a) + f(a,b))      (pairwise.cu:56-58)
template <typename T>
__global__ __launch_bounds__(256) void pairwise_bwd_kernel(const T* __restrict__ logits,
                                                           const T* __restrict__ g_pair, int N, int H, int W,
                                                           int size, int dil, T* __restrict__ g_logits) {
    const int K = size * size - 1;
    const int R = size / 2 * dil;
    const int64_t P = (int64_t)H * W;
    const int64_t total = (int64_t)N * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int64_t n = idx / P;
        const T* L = logits + n * P;
        const T* GP = g_pair + n * K * P;
        const int64_t p = (int64_t)y * W + x;
        const T here = L[p];
        const T ax = logsig(here), bx = logsig(-here);
        T acc = T(0);
        int k = 0;
        for (int dy = -R; dy <= R; dy += dil)
            for (int dx = -R; dx <= R; dx += dil) {
                if (dx == 0 && dy == 0) continue;
                const int x2 = x + dx, y2 = y + dy;
                if (x2 >= 0 && x2 < W && y2 >= 0 && y2 < H) {
                    const int64_t q = (int64_t)y2 * W + x2;
                    const T there = L[q];
                    const T ay = logsig(there), by = logsig(-there);
                    const T pair = pair_nlog(ax, bx, ay, by);
                    // channel k at p is the pair (p,q); channel K-1-k at q is the pair (q,p)
                    const T g = GP[(int64_t)k * P + p] + GP[(int64_t)(K - 1 - k) * P + q];
                    acc += -(t_exp(ay) - t_exp(by)) * t_exp(ax + bx + pair) * g;
                }
                ++k;
            }
        g_logits[n * P + p] = acc;
    }
}

// ---- size == 3 (every shipped config): tiled kernels ---------------------------------------------------------------------
// The kernels above evaluate log-sigmoid of both endpoints for every pair: ~3 exp/log pairs per neighbour, 24 per
// pixel -- ALU bound at 1.7 TB/s.  Here a workgroup stages (log s(x), log s(-x)) of its 16x64 tile + halo in LDS once
// (2 per pixel), so a pair costs one exp/log (pair_nlog), and the 8 output planes are written as full 64-wide rows.
constexpr int kPwTR = 16, kPwTC = 64, kPwMaxDil = 8;

template <typename T> struct LogPair { T a, b; };

template <typename T>
__device__ __forceinline__ void pw_stage(const T* __restrict__ L, int H, int W, int r0, int c0, int d, LogPair<T>* tile) {
    const int PR = kPwTR + 2 * d, PC = kPwTC + 2 * d;
    for (int i = threadIdx.x; i < PR * PC; i += 256) {
        const int r = r0 - d + i / PC, c = c0 - d + i % PC;
        LogPair<T> v{T(0), T(0)};
        if (r >= 0 && r < H && c >= 0 && c < W) { const T x = L[(int64_t)r * W + c]; v.a = logsig(x); v.b = logsig(-x); }
        tile[i] = v;
    }
}

// the exact body: log space as pairwise.cu:38-50 (every dtype; for f32 only when a logit of the tile is beyond +-34)
template <typename T>
__device__ __forceinline__ void pairwise3_fwd_exact(const T* __restrict__ logits, int H, int W, int d, T* __restrict__ out, unsigned char* pw_raw) {
    LogPair<T>* tile = reinterpret_cast<LogPair<T>*>(pw_raw);
    const int tiles_x = (W + kPwTC - 1) / kPwTC, tiles_y = (H + kPwTR - 1) / kPwTR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int64_t n = t / tiles_y;
    const int64_t P = (int64_t)H * W;
    const int r0 = ty * kPwTR, c0 = tx * kPwTC, PC = kPwTC + 2 * d;
    pw_stage(logits + n * P, H, W, r0, c0, d, tile);
    __syncthreads();
    const int lc = threadIdx.x & 63, lr0 = threadIdx.x >> 6;
    const int c = c0 + lc;
    if (c >= W) return;
    T* o = out + n * 8 * P;
#pragma unroll
    for (int j = 0; j < kPwTR / 4; ++j) {
        const int lr = lr0 + 4 * j, r = r0 + lr;
        if (r >= H) break;
        const LogPair<T> p = tile[(lr + d) * PC + lc + d];
        // f(p,q) == f(q,p) bit for bit (both sums commute): a pair is evaluated once, by its earlier pixel, and written
        // to channel k of p and channel 7-k of q; a pixel writes the zeros of its own padded (out-of-map) neighbours.
#pragma unroll
        for (int k = 0; k < 4; ++k) {                         // neighbours before p in raster order
            const int dy = k == 3 ? 0 : -1, dx = k == 3 ? -1 : k - 1;
            const int r2 = r + dy * d, c2 = c + dx * d;
            if (!(r2 >= 0 && r2 < H && c2 >= 0 && c2 < W)) o[(int64_t)k * P + (int64_t)r * W + c] = T(0);   // pairwise.cu:43-44
        }
#pragma unroll
        for (int k = 4; k < 8; ++k) {                         // neighbours after p
            const int dy = k == 4 ? 0 : 1, dx = k == 4 ? 1 : k - 6;
            const int r2 = r + dy * d, c2 = c + dx * d;
            T v = T(0);
            const bool in = r2 >= 0 && r2 < H && c2 >= 0 && c2 < W;
            if (in) {
                const LogPair<T> q = tile[(lr + d + dy * d) * PC + lc + d + dx * d];
                v = pair_nlog(p.a, p.b, q.a, q.b);
                o[(int64_t)(7 - k) * P + (int64_t)r2 * W + c2] = v;
            }
            o[(int64_t)k * P + (int64_t)r * W + c] = v;
        }
    }
}

template <typename T> struct ProbPair { T s, m; };      // sigmoid(x), sigmoid(-x)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // 1 ulp
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
// exp(-a), a >= 0: v_exp_f32 (2^x, ~1 ulp) on a * log2(e) -- the staged probabilities need 1e-6, not the last bit
__device__ __forceinline__ float fast_exp_neg(float a) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * a); }
__device__ __forceinline__ double fast_exp_neg(double a) { return exp(-a); }

template <typename T>
__global__ __launch_bounds__(256) void pairwise3_fwd_kernel(const T* __restrict__ logits, int H, int W, int d, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_raw[];
    pairwise3_fwd_exact<T>(logits, H, W, d, out, pw_raw);
}

// f32: the log-space body spends ~270 instructions per pixel (two accurate exp/log pairs per staged pixel and one per pair, a
// 64-bit address and a branch per store) and ran with the VALU 92 % busy -- 20.6 us for 59 MB.  While every |logit| of the tile
// + halo is <= 34 (block-uniform; otherwise the exact body above) probabilities are staged, (s, s') = (sigmoid(x), sigmoid(-x))
// with one v_exp_f32 and one v_rcp_f32 per pixel, and a pair is S = s_p s_q + s'_p s'_q (>= 3e-15), f = -ln2 log2 S: two
// multiply-adds and one v_log_f32.  Stores: a wave-uniform plane base + the pixel's 32-bit byte offset; a pair whose later pixel
// lies outside the map is 0 and its second store lands on the first one's address (same value) instead of behind a branch.
template <int D>      // dilation (compile time: the staged tile's row length divides by a constant); 0 = run time
__global__ __launch_bounds__(256) void pairwise3_fwd_fast_kernel(const float* __restrict__ logits, int H, int W, int d_, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_raw[];
    const int d = D ? D : d_;
    const int PR = kPwTR + 2 * d, PC = kPwTC + 2 * d;
    const int tiles_x = (W + kPwTC - 1) / kPwTC, tiles_y = (H + kPwTR - 1) / kPwTR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int64_t n = t / tiles_y;
    const int64_t P = (int64_t)H * W;
    const int r0 = ty * kPwTR, c0 = tx * kPwTC;
    const float* L = logits + n * P;
    // tile + halo logits, element i = threadIdx + 256 e of the staged tile (row-major): every lane of every load is used (rows
    // taken by waves with the columns in two passes issued half of their instructions for the 2d columns beyond 64), all
    // loads of a thread in flight together
    constexpr int kMaxE = ((kPwTR + 2 * (D ? D : kPwMaxDil)) * (kPwTC + 2 * (D ? D : kPwMaxDil)) + 255) / 256;
    float xv[kMaxE];
    bool sat = false;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int i = threadIdx.x + 256 * e;
        const int r = r0 - d + i / PC, cq = c0 - d + i % PC;
        xv[e] = 0.f;          // outside the map: never used as a neighbour (those pairs are 0)
        if (i < PR * PC && (unsigned)r < (unsigned)H && (unsigned)cq < (unsigned)W) { xv[e] = L[(uint32_t)(r * W + cq)]; sat |= !(fabsf(xv[e]) <= 34.f); }
    }
    if (__syncthreads_or(sat ? 1 : 0)) {       // rare
        pairwise3_fwd_exact<float>(logits, H, W, d, out, pw_raw);
        return;
    }
    ProbPair<float>* tile = reinterpret_cast<ProbPair<float>*>(pw_raw);
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int i = threadIdx.x + 256 * e;
        if (i < PR * PC) {
            const float x = xv[e], en = fast_exp_neg(fabsf(x)), big = fast_rcp(1.f + en), small = en * big;   // sigmoid(|x|), sigmoid(-|x|)
            tile[i] = x >= 0.f ? ProbPair<float>{big, small} : ProbPair<float>{small, big};
        }
    }
    __syncthreads();
    const int lc = threadIdx.x & 63, lr0 = threadIdx.x >> 6;
    const int c = c0 + lc;
    if (c >= W) return;
    char* ob = reinterpret_cast<char*>(out + n * 8 * P);                 // wave-uniform; 8 planes of one instance < 2^31 bytes (launcher)
    const uint32_t plane = (uint32_t)P * 4u;
    const bool c_lo = c - d >= 0, c_hi = c + d < W;
    const bool edge_tile = r0 < d || c0 < d || c0 + kPwTC + d > W;       // block-uniform: some pixel has an earlier neighbour outside the map
#pragma unroll
    for (int j = 0; j < kPwTR / 4; ++j) {
        const int lr = lr0 + 4 * j, r = r0 + lr;
        if (r >= H) break;
        const uint32_t pix = (uint32_t)(r * W + c) * 4u;
        const bool r_lo = r - d >= 0, r_hi = r + d < H;
        if (edge_tile) {
            // earlier neighbours outside the map: this pixel writes their zeros (pairwise.cu:43-44); inside, the neighbour writes
            if (!(r_lo && c_lo)) *reinterpret_cast<float*>(ob + pix) = 0.f;
            if (!r_lo) *reinterpret_cast<float*>(ob + plane + pix) = 0.f;
            if (!(r_lo && c_hi)) *reinterpret_cast<float*>(ob + 2u * plane + pix) = 0.f;
            if (!c_lo) *reinterpret_cast<float*>(ob + 3u * plane + pix) = 0.f;
        }
        const ProbPair<float> p = tile[(lr + d) * PC + lc + d];
#pragma unroll
        for (int k = 4; k < 8; ++k) {                         // later neighbours: (0,+d) (+d,-d) (+d,0) (+d,+d)
            const int dy = k == 4 ? 0 : 1, dx = k == 4 ? 1 : k - 6;
            const bool in = (dy ? r_hi : true) && (dx < 0 ? c_lo : (dx > 0 ? c_hi : true));
            const ProbPair<float> q = tile[(lr + d + dy * d) * PC + lc + d + dx * d];      // staged whether in the map or not
            const float S = p.s * q.s + p.m * q.m;
            const float v = in ? -0.69314718055994531f * __builtin_amdgcn_logf(S) : 0.f;
            const uint32_t own = (uint32_t)k * plane + pix;
            const uint32_t other = (uint32_t)(7 - k) * plane + pix + (uint32_t)((dy * d * W + dx * d) * 4);   // channel 7-k of q
            *reinterpret_cast<float*>(ob + own) = v;
            *reinterpret_cast<float*>(ob + (in ? other : own)) = v;
        }
    }
}

// backward: a pair (p,q) is evaluated once by its earlier pixel p, which keeps its own share and deposits q's share into slot
// [q][k-4] of an LDS plane (one writer per slot: no atomics); pairs whose earlier pixel lies outside the tile are evaluated
// by the later pixel itself.  Two bodies:
//   fast   every |logit| of the tile + halo <= 34: probabilities are staged, (s, s') = (sigmoid(x), sigmoid(-x)), one exp and one
//          division per pixel; with t = s - s', u = s s' and S = s_p s_q + s'_p s'_q (= exp(-f), >= 3e-15, cannot underflow)
//              d f / d x_p = -t_q u_p / S ,  d f / d x_q = -t_p u_q / S
//          -- the same quantity as pairwise.cu:56-58's -(s(b) - s(-b)) exp(logs(a) + logs(-a) + f), a dozen instructions per pair
//          instead of five exp / log evaluations (the kernel was bound by those: 43 us, 0.2 of the HBM peak);
//   exact  otherwise (block-uniform choice): log space exactly as pairwise.cu:38-58, (log s, log s', s - s') staged.
template <typename T> struct LogTriple { T a, b, dd; };


template <typename T, int D>      // D: dilation at compile time (the staged tile's row length divides by a constant); 0 = run time
__global__ __launch_bounds__(256) void pairwise3_bwd_kernel(const T* __restrict__ logits, const T* __restrict__ g_pair, int H, int W, int d_,
                                                            T* __restrict__ g_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_raw[];
    const int d = D ? D : d_;
    const int PR = kPwTR + 2 * d, PC = kPwTC + 2 * d;
    const int tiles_x = (W + kPwTC - 1) / kPwTC, tiles_y = (H + kPwTR - 1) / kPwTR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int64_t n = t / tiles_y;
    const int64_t P = (int64_t)H * W;
    const int r0 = ty * kPwTR, c0 = tx * kPwTC;
    const T* L = logits + n * P;
    // the tile + halo logits, element i = threadIdx + 256 e of the staged tile (row-major), all loads of a thread in flight
    // together (out-of-map positions: 0, never used as a neighbour).  Every lane of every load is used: rows taken by waves
    // with the columns in two passes issued half of their instructions for the 2d columns beyond 64.
    constexpr int kMaxE = ((kPwTR + 2 * (D ? D : kPwMaxDil)) * (kPwTC + 2 * (D ? D : kPwMaxDil)) + 255) / 256;
    T xv[kMaxE];
    bool sat = false;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int i = threadIdx.x + 256 * e;
        const int r = r0 - d + i / PC, cq = c0 - d + i % PC;
        xv[e] = T(0);
        if (i < PR * PC && (unsigned)r < (unsigned)H && (unsigned)cq < (unsigned)W) { xv[e] = L[(uint32_t)(r * W + cq)]; sat |= !(t_abs(xv[e]) <= T(34)); }
    }
    const int lc = threadIdx.x & 63, lr0 = threadIdx.x >> 6;
    const int c = c0 + lc;
    const T* GP = g_pair + n * 8 * P;
    T own[kPwTR / 4];
    // the gradient sums G = g[k][p] + g[7-k][q] of all four pixels of the thread: 64 loads, requested before the block decides
    // on its body and stages the tile, so that they fly meanwhile.  Addresses: a wave-uniform base (instance, plane, tap
    // offset) + the pixel's 32-bit index; a neighbour outside the map gets weight 0 later, so its address only has to stay
    // inside the instance's 8 planes -- one v_med3 on the index, no row / column clamps and no selects in front of the loads
    // (that arithmetic, ~5 instructions per load, was a quarter of the kernel).
    T G[kPwTR / 4][8];
    const int cc = min(c, W - 1);
    const int lim = (int)(8 * P - 1) * (int)sizeof(T);            // bytes; fits: the launcher takes this kernel only while 8 P sizeof(T) < 2^31
    const char* gb = reinterpret_cast<const char*>(GP);             // wave-uniform base + 32-bit byte offset: no 64-bit address arithmetic per load
    const int plane = (int)P * (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < kPwTR / 4; ++j) {
        const int r = min(r0 + lr0 + 4 * j, H - 1);
        const int pix = (r * W + cc) * (int)sizeof(T);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = k < 4 ? k : k + 1, dy = kk / 3 - 1, dx = kk % 3 - 1;
            const int tap = (7 - k) * plane + ((dy * d) * W + dx * d) * (int)sizeof(T);     // wave-uniform
            const int nb = min(max(tap + pix, 0), lim);
            G[j][k] = *reinterpret_cast<const T*>(gb + (uint32_t)(k * plane + pix)) + *reinterpret_cast<const T*>(gb + (uint32_t)nb);
        }
    }
    // (sat is known once the logit loads -- issued first -- have returned; the G loads stay in flight across the barrier)
    const bool exact = __syncthreads_or(sat ? 1 : 0) != 0;
    if (!exact) {
        ProbPair<T>* tile = reinterpret_cast<ProbPair<T>*>(pw_raw);
        T* slots = reinterpret_cast<T*>(pw_raw + sizeof(LogTriple<T>) * (size_t)PR * PC);     // same place in both bodies
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int i = threadIdx.x + 256 * e;
            if (i < PR * PC) {
                const T x = xv[e], en = fast_exp_neg(t_abs(x)), big = fast_rcp(T(1) + en), small = en * big;   // sigmoid(|x|), sigmoid(-|x|)
                tile[i] = x >= T(0) ? ProbPair<T>{big, small} : ProbPair<T>{small, big};
            }
        }
        for (int i = threadIdx.x; i < kPwTR * kPwTC * 4; i += 256) slots[i] = T(0);
        __syncthreads();
        // Branch-free: every tap of every pixel is evaluated (its staged neighbour always exists: the halo is d wide) and
        // weighted by 0 where it does not count -- neighbour outside the map, or a pair that the earlier pixel evaluates
        // (k < 4 with the neighbour inside the tile).  Per-pair branches cost more than the three wasted evaluations per
        // pixel: they keep the independent division chains of a thread from overlapping.  A share for a neighbour outside
        // the tile goes to a scratch cell.
        T* trash = slots + 4 * kPwTR * kPwTC + (threadIdx.x & 63);
#pragma unroll
        for (int j = 0; j < kPwTR / 4; ++j) {
            const int lr = lr0 + 4 * j, r = r0 + lr;
            const ProbPair<T> p = tile[(lr + d) * PC + lc + d];
            const T tp = p.s - p.m, up = p.s * p.m;
            const bool r_lo = r - d >= 0, r_hi = r + d < H, c_lo2 = c - d >= 0, c_hi2 = c + d < W;
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = k < 4 ? k : k + 1, dy = kk / 3 - 1, dx = kk % 3 - 1;
                const bool in_map = (dy < 0 ? r_lo : (dy > 0 ? r_hi : true)) && (dx < 0 ? c_lo2 : (dx > 0 ? c_hi2 : true));
                const int lr2 = lr + dy * d, lc2 = lc + dx * d;
                const bool q_in_tile = lr2 >= 0 && lr2 < kPwTR && lc2 >= 0 && lc2 < kPwTC;
                const bool use = in_map && (k >= 4 || !q_in_tile);
                const ProbPair<T> q = tile[(lr2 + d) * PC + lc2 + d];
                const T S = p.s * q.s + p.m * q.m;                 // >= 3e-15: every |logit| <= 34
                const T m = use ? G[j][k] * fast_rcp(S) : T(0);    // channel k at p is the pair (p,q); channel 7-k at q is the pair (q,p)
                acc += -(q.s - q.m) * up * m;
                if (k >= 4) *(q_in_tile ? &slots[(k - 4) * (kPwTR * kPwTC) + lr2 * kPwTC + lc2] : trash) = -tp * (q.s * q.m) * m;
            }
            own[j] = acc;
        }
    } else {
        LogTriple<T>* tile = reinterpret_cast<LogTriple<T>*>(pw_raw);
        T* slots = reinterpret_cast<T*>(tile + PR * PC);                     // [4][kPwTR*kPwTC]: shares deposited by earlier pixels
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int i = threadIdx.x + 256 * e;
            const int r = r0 - d + i / PC, cq = c0 - d + i % PC;
            if (i < PR * PC) {
                LogTriple<T> v{T(0), T(0), T(0)};
                if ((unsigned)r < (unsigned)H && (unsigned)cq < (unsigned)W) { const T x = xv[e]; v.a = logsig(x); v.b = logsig(-x); v.dd = t_exp(v.a) - t_exp(v.b); }
                tile[i] = v;
            }
        }
        for (int i = threadIdx.x; i < kPwTR * kPwTC * 4; i += 256) slots[i] = T(0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPwTR / 4; ++j) {
            const int lr = lr0 + 4 * j, r = r0 + lr;
            own[j] = T(0);
            if (c >= W || r >= H) continue;
            const LogTriple<T> p = tile[(lr + d) * PC + lc + d];
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = k < 4 ? k : k + 1, dy = kk / 3 - 1, dx = kk % 3 - 1;
                const int r2 = r + dy * d, c2 = c + dx * d;
                if (!(r2 >= 0 && r2 < H && c2 >= 0 && c2 < W)) continue;
                const int lr2 = lr + dy * d, lc2 = lc + dx * d;
                const bool q_in_tile = lr2 >= 0 && lr2 < kPwTR && lc2 >= 0 && lc2 < kPwTC;
                if (k < 4 && q_in_tile) continue;                 // that pair is evaluated by q (the earlier pixel) and deposited
                const LogTriple<T> q = tile[(lr2 + d) * PC + lc2 + d];
                const T pair = pair_nlog(p.a, p.b, q.a, q.b);
                const T Gk = G[j][k];
                acc += -q.dd * t_exp(p.a + p.b + pair) * Gk;
                if (k >= 4 && q_in_tile) slots[(k - 4) * (kPwTR * kPwTC) + lr2 * kPwTC + lc2] = -p.dd * t_exp(q.a + q.b + pair) * Gk;
            }
            own[j] = acc;
        }
    }
    __syncthreads();
    const T* slots = reinterpret_cast<const T*>(pw_raw + sizeof(LogTriple<T>) * (size_t)PR * PC);
#pragma unroll
    for (int j = 0; j < kPwTR / 4; ++j) {
        const int lr = lr0 + 4 * j, r = r0 + lr;
        if (c >= W || r >= H) continue;
        const T* sl = slots + lr * kPwTC + lc;             // four planes [k - 4][pixel]: a wave's accesses fall on distinct banks
        constexpr int kPl = kPwTR * kPwTC;
        // summation order: the four earlier neighbours (k = 3,2,1,0 deposit into slots 0..3), then the later ones
        g_logits[n * P + (int64_t)r * W + c] = (((sl[3 * kPl] + sl[2 * kPl]) + sl[kPl]) + sl[0]) + own[j];
    }
}

// ---- f32, size == 3, rows of whole float4 (W % 4 == 0, 16-byte aligned planes): the "wide" forward, the "pair" backward ---------
// The kernels above give a lane one column and four rows: 32 (forward) / 64 (backward) four-byte memory instructions per thread,
// and at 52 MB of output / input the launch is bound by ISSUING them (MI355X_MICROARCH.md: epilogue store tails are store-issue
// bound; 16-byte accesses halve them).  Here a thread owns FOUR ADJACENT PIXELS of one row of the 16 x 64 tile:
//   forward   every pixel evaluates all eight of its pairs itself (f(p,q) = f(q,p) bit for bit, so the partner would get the same
//             number): 8 x float4 stores per thread, aligned, no partner stores, no edge cases; the second evaluation of a pair
//             is two multiply-adds and one v_log_f32 against 4 x fewer store instructions;
//   backward  every UNORDERED pair once (further down).  Rounds 3-5 summed d f/d x_p over all eight taps in the pixel itself: 16 float4
//             loads per thread, every gradient element fetched twice (tools/micro/pw_bwd_wide_ref.inc keeps that kernel as the
//             measuring stick: 19.4 us where the pair kernel takes 13.8 and a copy of the bytes 12.7, same box, cold).
// The neighbours' probabilities of the four pixels overlap: 3 rows x (4 + 2d) staged entries are read once per thread.
// A tile with a logit beyond +-34 (S could underflow) takes a per-pixel log-space path straight from global memory, exactly
// pairwise.cu:38-58 (block-uniform choice, no extra LDS).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // a float4 at any dword address

// staged tile: two planes of floats, s = sigmoid(x) and m = sigmoid(-x), rows padded to a multiple of 4 floats so that a thread
// reads its 4 + 2 D neighbour columns of a row as whole float4 (ds_read_b128 at a 16-byte lane stride: conflict-free; 8-byte
// (s, m) pairs at a 32-byte lane stride put 32 lanes on 8 banks)
template <int D, int TC> struct PwGeom { static constexpr int PC = (TC + 2 * D + 3) & ~3, NW4 = (4 + 2 * D + 3) / 4; };

template <int D, int TR, int TC>
__device__ __forceinline__ void pw3_stage_probs(const float* __restrict__ L, int H, int W, int r0, int c0, float* ts, float* tm, bool& sat_out) {
    constexpr int PR = TR + 2 * D, PCs = TC + 2 * D, PC = PwGeom<D, TC>::PC;
    constexpr int kMaxE = (PR * PCs + 255) / 256;
    float xv[kMaxE];
    bool sat = false;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int i = threadIdx.x + 256 * e;
        const int r = r0 - D + i / PCs, cq = c0 - D + i % PCs;
        xv[e] = 0.f;          // outside the map: never used as a neighbour (those taps weigh 0)
        if (i < PR * PCs && (unsigned)r < (unsigned)H && (unsigned)cq < (unsigned)W) { xv[e] = L[(uint32_t)(r * W + cq)]; sat |= !(fabsf(xv[e]) <= 34.f); }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int i = threadIdx.x + 256 * e;
        if (i < PR * PCs) {
            const float x = xv[e], en = fast_exp_neg(fabsf(x)), big = fast_rcp(1.f + en), small = en * big;   // sigmoid(|x|), sigmoid(-|x|)
            const int o = (i / PCs) * PC + i % PCs;
            ts[o] = x >= 0.f ? big : small;
            tm[o] = x >= 0.f ? small : big;
        }
    }
    sat_out = __syncthreads_or(sat ? 1 : 0) != 0;       // also: the staged tile is complete
}

// rows r - D, r, r + D ; columns c - D .. : the thread's window of one plane
template <int D, int TC>
__device__ __forceinline__ void pw3_window(const float* plane, int lr, int lc, float (&q)[3][4 * PwGeom<D, TC>::NW4]) {
    constexpr int PC = PwGeom<D, TC>::PC, NW4 = PwGeom<D, TC>::NW4;
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int x4 = 0; x4 < NW4; ++x4) {
            const float4 v = *reinterpret_cast<const float4*>(plane + (lr + y * D) * PC + lc + 4 * x4);
            q[y][4 * x4] = v.x; q[y][4 * x4 + 1] = v.y; q[y][4 * x4 + 2] = v.z; q[y][4 * x4 + 3] = v.w;
        }
}

template <int D, int TR, int TC>
__global__ __launch_bounds__(256) void pairwise3_fwd_wide_kernel(const float* __restrict__ logits, int H, int W, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_raw[];
    constexpr int PC = PwGeom<D, TC>::PC, NWp = 4 * PwGeom<D, TC>::NW4;
    static_assert(TR * TC == 1024 && TC % 4 == 0, "256 threads x four adjacent pixels");
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int64_t n = t / tiles_y;
    const int64_t P = (int64_t)H * W;
    const int r0 = ty * TR, c0 = tx * TC;
    const float* L = logits + n * P;
    float* ts = reinterpret_cast<float*>(pw_raw);
    float* tm = ts + (TR + 2 * D) * PC;
    bool sat;
    pw3_stage_probs<D, TR, TC>(L, H, W, r0, c0, ts, tm, sat);
    const int lr = threadIdx.x / (TC / 4), lc = (threadIdx.x % (TC / 4)) * 4;
    const int r = r0 + lr, c = c0 + lc;
    if (r >= H || c >= W) return;                                        // W % 4 == 0: the four pixels are in the map together
    char* ob = reinterpret_cast<char*>(out + n * 8 * P);                 // wave-uniform; 8 planes of one instance < 2^31 bytes (launcher)
    const uint32_t plane = (uint32_t)P * 4u, pix = (uint32_t)(r * W + c) * 4u;
    if (sat) {                                                           // rare: log space, straight from global memory
        for (int i = 0; i < 4; ++i) {
            const float here = L[r * W + c + i];
            const float ax = logsig(here), bx = logsig(-here);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = k < 4 ? k : k + 1, r2 = r + (kk / 3 - 1) * D, c2 = c + i + (kk % 3 - 1) * D;
                float v = 0.f;
                if (r2 >= 0 && r2 < H && c2 >= 0 && c2 < W) { const float there = L[r2 * W + c2]; v = pair_nlog(ax, bx, logsig(there), logsig(-there)); }
                *reinterpret_cast<float*>(ob + (uint32_t)k * plane + pix + 4u * i) = v;
            }
        }
        return;
    }
    float qs[3][NWp], qm[3][NWp];                                        // rows r - D, r, r + D ; columns c - D .. c + 3 + D
    pw3_window<D, TC>(ts, lr, lc, qs);
    pw3_window<D, TC>(tm, lr, lc, qm);
    const bool r_lo = r - D >= 0, r_hi = r + D < H;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int kk = k < 4 ? k : k + 1, dy = kk / 3 - 1, dx = kk % 3 - 1;
        const bool row_in = dy < 0 ? r_lo : (dy > 0 ? r_hi : true);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = row_in && (dx < 0 ? c + i - D >= 0 : (dx > 0 ? c + i + D < W : true));
            const float S = qs[1][D + i] * qs[1 + dy][D + i + dx * D] + qm[1][D + i] * qm[1 + dy][D + i + dx * D];
            v[i] = in ? -0.69314718055994531f * __builtin_amdgcn_logf(S) : 0.f;                 // pairwise.cu:43-44: padded pairs are 0
        }
        // (non-temporal: 52 MB nobody in this launch reads -- 12.7 -> 12.3 us cold, same box, interleaved four times)
        typedef float f4s __attribute__((ext_vector_type(4)));
        const f4s ov = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(ov, reinterpret_cast<f4s*>(ob + (uint32_t)k * plane + pix));
    }
}

// developer trace of these kernels (-DBXI_PW_TRACE; tools/micro/pw_bwd.hip): wall-clock stamps collected in scalar registers, written by the
// workgroup's first lane at the end -- no memory wait in the middle (BXI_T's pointer load would drain the requests in flight)
#ifdef BXI_PW_TRACE
static __device__ long long* g_pw_trace = nullptr;
#define PWT_DECL long long pwt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PWT(ph) do { long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pwt_[ph] = t_; } while (0)
#define PWT_FLUSH(kid) do { if (threadIdx.x == 0 && g_pw_trace) for (int ph_ = 0; ph_ < 8; ++ph_) g_pw_trace[((size_t)(kid) * 8192 + blockIdx.x) * 8 + ph_] = pwt_[ph_]; } while (0)
#else
#define PWT_DECL do {} while (0)
#define PWT(ph) do {} while (0)
#define PWT_FLUSH(kid) do {} while (0)
#endif
// ---- f32, size == 3, W % 4 == 0: every UNORDERED pair once ("pair" kernel) ---------------------------------------------------------
// Letting every pixel evaluate all eight of its taps evaluates each pair {p, q} twice and fetches each element of the upstream gradient
// twice (once as g[k][p], once as the partner term g[7-k][.] of a neighbour).  But the pair has ONE
//     S = s_p s_q + s'_p s'_q ,  ONE 1/S ,  ONE G = g[k][p] + g[7-k][q]                                              (pairwise.cu:52-66)
// and feeds d/dx_p = -u_p (t_q G/S) and d/dx_q = -u_q (t_p G/S)   (t = s - s', u = s s').  Here the EARLIER pixel p of a pair (k = 4..7:
// q to the right in the row, or in the row D below) evaluates it, keeps t_q G/S and hands t_p G/S to q; the factor -u is applied by the
// owner of a pixel at the end.  A thread owns four adjacent pixels of a row (16-byte loads, as above): 8 float4 loads (g[4..7] at p,
// g[3..0] at the four later partners) instead of 16, 16 pair evaluations instead of 32.  How a share reaches q:
//   same row (k = 4)      q = p + D columns: the thread's own later pixel, or the next lane's (DPP row_shr:1 -- a tile row is 16 lanes = one DPP row)
//   row + D (k = 5, 6, 7) the three shares aimed at one column are summed in registers; the D columns that spill to the left / right belong to
//                         the pixels of the neighbouring lanes' threads (DPP row_shl:1 / row_shr:1); the thread then owns the COMPLETE downward
//                         share of the four pixels below it and writes them, one writer per cell, to an LDS plane V[row + D]
// Pairs whose earlier pixel lies OUTSIDE the tile (top D rows, D columns at the left / right edge) cannot be handed over by another workgroup:
// in an "edge" pass every thread takes one such pixel x of the tile and evaluates only its share of the backward taps that leave the tile
// (k = 0..3, G = g[k][x] + g[7-k][q]) into a second plane E -- D (TC + 2 TRT - 2D) pixels, <= 4 taps each: ~4 % more evaluations instead of 100 %.
// XR > 0: XR more rows of ONE pixel per thread (wave = row, lane = column; wave_shr / wave_shl DPP for the column shifts), so that
// 32 x 200 x 256 is 1280 workgroups = one residency round at five per CU, exactly as the wide kernel's second phase -- but its loads go out
// with the first phase's (8 + 8 + 8 dwords more in flight fit now that the first phase needs 32 registers of gradient, not 64).
// Out-of-map neighbours are staged as x = 0 (s = s' = 1/2, t = 0 exactly): their pairs contribute 0 by arithmetic, G is forced to 0 by a select
// (a NaN / inf elsewhere in the upstream gradient must not leak through a clamped address).  Sums per pixel: own four taps, then the row share,
// then V, then E -- a fixed order, run-to-run identical.
#ifndef BXI_PWP_OCC
#define BXI_PWP_OCC 5
#endif

#define BXI_DPP_ROW_SHL1 0x101
#define BXI_DPP_ROW_SHR1 0x111
#define BXI_DPP_WAVE_SHL1 0x130
#define BXI_DPP_WAVE_SHR1 0x138
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float v) {       // lanes without a source get 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int N>
__device__ __forceinline__ float dpp_zero_n(float v) {
#pragma unroll
    for (int s = 0; s < N; ++s) v = dpp_zero<CTRL>(v);
    return v;
}

// one pixel's gradient in log space straight from global memory, every tap (pairwise.cu:56-58): tiles with a logit beyond +-34
__device__ __forceinline__ float pw3_bwd_exact_px(const float* __restrict__ L, const float* __restrict__ gp, int64_t P, int H, int W, int r, int c, int d) {
    const float here = L[r * W + c];
    const float ax = logsig(here), bx = logsig(-here);
    float acc = 0.f;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int kk = k < 4 ? k : k + 1, r2 = r + (kk / 3 - 1) * d, c2 = c + (kk % 3 - 1) * d;
        if (r2 >= 0 && r2 < H && c2 >= 0 && c2 < W) {
            const float there = L[r2 * W + c2];
            const float ay = logsig(there), by = logsig(-there);
            const float pair = pair_nlog(ax, bx, ay, by);
            const float g = gp[k * P + (int64_t)r * W + c] + gp[(7 - k) * P + (int64_t)r2 * W + c2];
            acc += -(expf(ay) - expf(by)) * expf(ax + bx + pair) * g;
        }
    }
    return acc;
}

template <int D, int XR> struct PwPairGeom {
    static constexpr int TR = 16, TC = 64, TRT = TR + XR, PC = PwGeom<D, TC>::PC, PR = TRT + 2 * D;
    // edge items: (tap k = 0..2, top row a < D, quad) -- 4 pixels each; then the strips, D pixels each: (left, tap 0, row a >= D), (left, tap 3, any row), (right, tap 2, row a >= D)
    static constexpr int kTopItems = 3 * D * (TC / 4), kStripRows = TRT - D, kStripItems = 2 * kStripRows + TRT;   // (the left tap 3 leaves the tile in EVERY row)
    static constexpr int kStripLane0 = (kTopItems + 63) / 64 * 64;      // strip items start at a wave boundary: they are another load instruction
    static_assert(kStripLane0 + kStripItems <= 256, "one edge item per thread");
    // staging: B = the quads of the rows no thread owns as its four pixels (extra rows, halo rows above and below), C = the halo columns
    static constexpr int kQuadsB = (XR + 2 * D) * (TC / 4), kColsC = PR * 2 * D;
    static_assert(kQuadsB <= 256 && kColsC <= 256, "one of each per thread");
    static constexpr int kEtop = 3 * D * TC, kEstrip = 3 * TRT * 4;      // floats
    static constexpr size_t lds_bytes = sizeof(float) * (2 * (size_t)PR * PC + (size_t)TRT * TC + kEtop + kEstrip + 4);
};

template <int D, int XR>
__global__ __launch_bounds__(256, BXI_PWP_OCC) void pairwise3_bwd_pair_kernel(const float* __restrict__ logits, const float* __restrict__ g_pair, int H, int W,
                                                                 float* __restrict__ g_logits, int xcd_swizzle) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_raw[];
    typedef PwPairGeom<D, XR> Gm;
    constexpr int TR = Gm::TR, TC = Gm::TC, TRT = Gm::TRT, PC = Gm::PC, PR = Gm::PR, NWp = 4 * PwGeom<D, TC>::NW4;
    static_assert(D >= 1 && D <= 4, "the column spill of a thread reaches the adjacent lane only");
    static_assert(XR == 0 || XR * TC == 256, "second phase: one pixel per thread, wave = row");
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TRT - 1) / TRT;
    int t = (int)blockIdx.x;                                             // XCD-aware tile order: see pairwise3_bwd_wide_kernel
    if (xcd_swizzle == 1) {
        const unsigned x = blockIdx.x % 8u, q = gridDim.x / 8u, r = gridDim.x % 8u;
        t = (int)(x * q + (x < r ? x : r) + blockIdx.x / 8u);
    } else if (xcd_swizzle >= 2) {                                       // groups of (xcd_swizzle - 1) tile rows stay on one XCD; consecutive groups on consecutive XCDs
        const unsigned gsz = (unsigned)tiles_x * (unsigned)(xcd_swizzle - 1), full = gridDim.x / (8u * gsz) * (8u * gsz);
        if (blockIdx.x < full) { const unsigned x = blockIdx.x % 8u, j = blockIdx.x / 8u; t = (int)((8u * (j / gsz) + x) * gsz + j % gsz); }
    }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int64_t n = t / tiles_y;
    const int64_t P = (int64_t)H * W;
    const int r0 = ty * TRT, c0 = tx * TC;
    const float* L = logits + n * P;
    const int tid = (int)threadIdx.x;
    // Addresses: a wave-uniform base in scalar registers -- the instance's logits, its gradient planes, plane k shifted by D rows -- + ONE
    // per-lane byte offset per pixel (+ the column shift as the instruction's immediate offset).  The bases are made opaque so that the
    // compiler does not re-associate them into per-lane 64-bit pointers (two address registers per load, where the loads of a thread all
    // have to be in flight together).  A tap that leaves the map gets weight 0 later; its address only has to stay inside the instance's
    // 8 planes: it does, |tap offset| < one plane (launcher).
    typedef const __attribute__((address_space(1))) char* gptr;        // global address space, kept through the asm statements below
    typedef const __attribute__((address_space(1))) float* gf1;
    typedef float f4a __attribute__((ext_vector_type(4)));                // a float4 at a 16-byte address (a plain vector: HIP's float4 class does not load from an address space)
    typedef const __attribute__((address_space(1))) f4a* gf4;
    typedef const __attribute__((address_space(1))) f4u* gf4u;
    typedef float fDv __attribute__((ext_vector_type(D == 1 ? 2 : D), aligned(4)));   // D adjacent pixels at any dword address: one load instruction (D = 1: a plain float)
    typedef const __attribute__((address_space(1))) fDv* gfD;
    struct fD { float v[D]; };
    auto loadD = [](gptr p_) { fD r_; if constexpr (D == 1) r_.v[0] = *(gf1)p_; else { const fDv t_ = *(gfD)p_; for (int i = 0; i < D; ++i) r_.v[i] = t_[i]; } return r_; };
    gptr pb[8];                                                          // plane k
    gptr pdn[3];                                                         // plane k (0..2) + D rows: the partners below
    gptr gb = (gptr)(g_pair + n * 8 * P), lb = (gptr)L;
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) { pb[k] = gb + k * P * 4; asm volatile("" : "+s"(pb[k])); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { pdn[k] = gb + k * P * 4 + (int64_t)D * W * 4; asm volatile("" : "+s"(pdn[k])); }
        asm volatile("" : "+s"(gb)); asm volatile("" : "+s"(lb));
    }
    const int plane = (int)P * 4;
    PWT_DECL;
    PWT(0);
    // ---- requests, in the order their data is needed (loads return in order), as few and as wide as they can be: beyond its bytes a launch
    // pays ~0.16 us per load instruction of a workgroup (profiles/NOTES.md R6-1) -- the first version of this kernel asked for the staged logits
    // and the edge pixels one dword at a time (15 + 16 instructions per thread) and ran 21.4 us where the same kernel without them took 13.7.
    // 1: the logits.  A: the thread's own four pixels; B: a quad of the rows nobody owns as four pixels (extra rows, halo rows); C: a halo column cell
    const int lrA = tid / (TC / 4), lcA = (tid % (TC / 4)) * 4;
    auto rowB = [&](int th) { const int rb = th / (TC / 4); return rb < XR ? TR + rb : (rb < XR + D ? rb - XR - D : TRT + rb - XR - D); };   // tile row of B quad `th`
    f4a xa, xb;           // (xb, xc and the edge registers below are written and read under the same thread-index conditions: no
    float xc;             //  zero defaults -- a default's v_mov would have to wait for the other branch's load into the same register)
    xa = *(gf4)(lb + (uint32_t)((min(r0 + lrA, H - 1) * W + min(c0 + lcA, W - 4)) * 4));
    if (tid < Gm::kQuadsB) xb = *(gf4)(lb + (uint32_t)((min(max(r0 + rowB(tid), 0), H - 1) * W + min(c0 + (tid % (TC / 4)) * 4, W - 4)) * 4));
    if (tid >= 256 - Gm::kColsC) {
        const int ci = tid - (256 - Gm::kColsC), pr = ci / (2 * D), hc = ci % (2 * D);
        xc = *(gf1)(lb + (uint32_t)((min(max(r0 - D + pr, 0), H - 1) * W + min(max(c0 - D + (hc < D ? hc : TC + hc), 0), W - 1)) * 4));
    }
    // 2: the edge item -- pixels of the tile with an EARLIER partner outside it (taps k = 0..3 = (-D,-D) (-D,0) (-D,+D) (0,-D)): g[k] at the pixels, g[7-k] at the partners
    f4a eo4; f4u ep4;
    fD eoD, epD;
    auto top_item = [&](int th, int& k, int& a, int& b) { k = th / (D * (TC / 4)); a = (th % (D * (TC / 4))) / (TC / 4); b = (th % (TC / 4)) * 4; };
    auto strip_item = [&](int th, int& k, int& a, int& b, int& dy, int& dx) -> int {
        const int s_ = th - Gm::kStripLane0, w = s_ < Gm::kStripRows ? 0 : (s_ < Gm::kStripRows + TRT ? 1 : 2);   // 0: left, tap 0 ; 1: left, tap 3 ; 2: right, tap 2
        a = w == 0 ? D + s_ : (w == 1 ? s_ - Gm::kStripRows : D + s_ - Gm::kStripRows - TRT);
        k = w == 0 ? 0 : (w == 1 ? 3 : 2); b = w == 2 ? TC - D : 0; dy = w == 1 ? 0 : -1; dx = w == 2 ? 1 : -1;
        return w;
    };
    if (tid < Gm::kTopItems) {
        int k, a, b;
        top_item(tid, k, a, b);
        const int pixe = (min(r0 + a, H - 1) * W + min(c0 + b, W - 4)) * 4;
        eo4 = *(gf4)(gb + (uint32_t)(k * plane + pixe));
        ep4 = *(gf4u)(gb + (uint32_t)((7 - k) * plane + pixe - D * W * 4 + (k - 1) * D * 4));
    } else if (tid >= Gm::kStripLane0 && tid < Gm::kStripLane0 + Gm::kStripItems) {
        int k, a, b, dy, dx;
        strip_item(tid, k, a, b, dy, dx);
        const int pixe = (min(r0 + a, H - 1) * W + min(c0 + b, W - D)) * 4;
        eoD = loadD(gb + (uint32_t)(k * plane + pixe));
        epD = loadD(gb + (uint32_t)((7 - k) * plane + pixe + (dy * D * W + dx * D) * 4));
    }
    // 3: the one pixel of the extra rows: g[4..7] at p, g[3..0] at the later partner (k = 4 + j: (0,+D) (+D,-D) (+D,0) (+D,+D); its channel is 7 - k)
    float xo[4], xp[4];
    if (XR > 0) {
        const int pix2 = (min(r0 + TR + tid / TC, H - 1) * W + min(c0 + tid % TC, W - 1)) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = j == 0 ? 0 : 1, dx = j == 0 ? 1 : j - 2;
            xo[j] = *(gf1)(pb[4 + j] + (uint32_t)pix2);
            xp[j] = *(gf1)((dy ? pdn[3 - j] : pb[3 - j]) + (uint32_t)pix2 + dx * D * 4);
        }
    }
    // 4: four adjacent pixels of a row, the same planes as sixteen-byte loads
    f4a own[4];
    f4u part[4];
    {
        uint32_t pix = (uint32_t)((min(r0 + lrA, H - 1) * W + min(c0 + lcA, W - 4)) * 4);
        asm volatile("" : "+v"(pix));                                    // (its own register: shared with the logits' offset the loads below came out with 64-bit per-lane addresses)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = j == 0 ? 0 : 1, dx = j == 0 ? 1 : j - 2;
            // non-temporal: every byte of these planes is wanted once by this launch (the few lines a neighbouring tile's edge items ask for again
            // come from the Infinity Cache); kept out of the L2's way they leave it to the logits and the halo lines.  Same box, cold, us:
            // plain 15.5 -> output stores nt 14.85 -> + these loads nt 13.77 (the copy: 12.65); nt on the extra rows' dwords +0.3, on the edge items 0,
            // on the logits +0.35 (profiles/NOTES.md R6-2)
            own[j] = __builtin_nontemporal_load((gf4)(pb[4 + j] + (uint32_t)pix));
            part[j] = __builtin_nontemporal_load((gf4u)((dy ? pdn[3 - j] : pb[3 - j]) + (uint32_t)pix + dx * D * 4));
        }
    }
    // (everything below is derived from a thread index the compiler cannot see through: nothing of it is computed -- and kept in registers --
    // while the registers of the requests above are being filled)
    int th = tid;
    asm volatile("" : "+v"(th));
    PWT(1);
    const int lr = th / (TC / 4), lc = (th % (TC / 4)) * 4;
    const int r = r0 + lr, c = c0 + lc;
    const bool live = r < H && c < W;                                   // W % 4 == 0: the four pixels are in the map together
    const int lr2 = TR + th / TC, lc2 = th % TC;
    const int r2 = r0 + lr2, c2 = c0 + lc2;
    const bool live2 = XR > 0 && r2 < H && c2 < W;
    // ---- the staged tile: s = sigmoid(x), s' = sigmoid(-x), two planes (rows padded to whole float4; staged row / column = tile row / column + D);
    // out-of-map positions are x = 0 (s = s' = 1/2, t = 0: their pairs vanish by arithmetic)
    float* ts = reinterpret_cast<float*>(pw_raw);
    float* tm = ts + PR * PC;
    float* Vp = tm + PR * PC;                                            // [TRT][TC]: the complete share from the row D above
    float* Et = Vp + TRT * TC;                                           // [3][D][TC]: top rows' shares from above the tile, by tap
    float* Es = Et + Gm::kEtop;                                          // [3][TRT][4]: left tap 0, left tap 3, right tap 2 -- D pixels per row
    int* satw = reinterpret_cast<int*>(Es + Gm::kEstrip);                // [4]: one flag per wave
    bool sat = false;
    auto stage = [&](float x, int o) {
        sat |= !(fabsf(x) <= 34.f);
        const float en = fast_exp_neg(fabsf(x)), big = fast_rcp(1.f + en), small = en * big;       // sigmoid(|x|), sigmoid(-|x|)
        ts[o] = x >= 0.f ? big : small;
        tm[o] = x >= 0.f ? small : big;
    };
    {
        const bool col_in = c < W;
        const bool inA = r < H && col_in;
        const float a4[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) stage(inA ? a4[i] : 0.f, (lr + D) * PC + lc + D + i);
        if (th < Gm::kQuadsB) {
            const int trB = rowB(th);
            const bool inB = (unsigned)(r0 + trB) < (unsigned)H && col_in;
            const float b4[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) stage(inB ? b4[i] : 0.f, (trB + D) * PC + lc + D + i);
        }
        if (th >= 256 - Gm::kColsC) {
            const int ci = th - (256 - Gm::kColsC), pr = ci / (2 * D), hc = ci % (2 * D), sc = hc < D ? hc : TC + hc;
            const bool inC = (unsigned)(r0 - D + pr) < (unsigned)H && (unsigned)(c0 - D + sc) < (unsigned)W;
            stage(inC ? xc : 0.f, pr * PC + sc);
        }
    }
    if ((th & 63) == 0) satw[th >> 6] = __any(sat) ? 1 : 0;
    PWT(2);
    lds_barrier();                                                       // the staged tile is complete (the gradient requests stay in flight)
    PWT(3);
    if (satw[0] | satw[1] | satw[2] | satw[3]) {                         // rare, block-uniform: log space, straight from global memory
        if (live) {
            float a4[4];
            for (int i = 0; i < 4; ++i) a4[i] = pw3_bwd_exact_px(L, g_pair + n * 8 * P, P, H, W, r, c + i, D);
            *reinterpret_cast<float4*>(g_logits + n * P + (int64_t)r * W + c) = make_float4(a4[0], a4[1], a4[2], a4[3]);
        }
        if (live2) g_logits[n * P + (int64_t)r2 * W + c2] = pw3_bwd_exact_px(L, g_pair + n * 8 * P, P, H, W, r2, c2, D);
        return;
    }
    // ---- edge pass: the shares that come from outside the tile, one writer per cell: Et[tap][row][column], Es[which][row][pixel]
    if (th < Gm::kTopItems) {
        int k, a, b;
        top_item(th, k, a, b);
        const int dx = k - 1;
        const bool rows_in = r0 + a < H && r0 + a - D >= 0 && c0 + b < W;
        const float o4[4] = {eo4.x, eo4.y, eo4.z, eo4.w}, p4[4] = {ep4.x, ep4.y, ep4.z, ep4.w};
        float sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (a + D) * PC + b + D + i;
            const float sp = ts[o], mp = tm[o], sq = ts[o - D * PC + dx * D], mq = tm[o - D * PC + dx * D];
            const bool in = rows_in && (unsigned)(c0 + b + i + dx * D) < (unsigned)W;
            const float G = in ? o4[i] + p4[i] : 0.f;
            sh[i] = (sq - mq) * (G * fast_rcp(sp * sq + mp * mq));
        }
        *reinterpret_cast<float4*>(Et + (k * D + a) * TC + b) = make_float4(sh[0], sh[1], sh[2], sh[3]);
    } else if (th >= Gm::kStripLane0 && th < Gm::kStripLane0 + Gm::kStripItems) {
        int k, a, b, dy, dx;
        const int w = strip_item(th, k, a, b, dy, dx);
        const bool rows_in = r0 + a < H && (unsigned)(c0 + b + dx * D) < (unsigned)W && c0 + b < W;      // D <= 4, W % 4 == 0, tiles start at multiples of 64: the D partners are in the map together
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int o = (a + D) * PC + b + D + i;
            const float sp = ts[o], mp = tm[o], sq = ts[o + dy * D * PC + dx * D], mq = tm[o + dy * D * PC + dx * D];
            const float G = rows_in ? eoD.v[i] + epD.v[i] : 0.f;
            Es[(w * TRT + a) * 4 + i] = (sq - mq) * (G * fast_rcp(sp * sq + mp * mq));
        }
    }
    PWT(4);
    // ---- the extra rows: one pixel per thread (wave = row, lane = column)
    float acc2 = 0.f;
    if (XR > 0) {
        const float* s_ = ts + (lr2 + D) * PC + lc2 + D;
        const float* m_ = tm + (lr2 + D) * PC + lc2 + D;
        const float sp = s_[0], mp = m_[0], tp = sp - mp;
        const bool row_d = live2 && r2 + D < H, c_lo = c2 - D >= 0, c_hi = c2 + D < W;
        float sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = j == 0 ? D : D * PC + (j - 2) * D;
            const float sq = s_[off], mq = m_[off];
            const bool in = j == 0 ? (live2 && c_hi) : (row_d && (j == 1 ? c_lo : (j == 3 ? c_hi : true)));
            const float G = in ? xo[j] + xp[j] : 0.f;
            const float cc = G * fast_rcp(sp * sq + mp * mq);
            acc2 += (sq - mq) * cc;
            sh[j] = tp * cc;
        }
        acc2 += dpp_zero_n<BXI_DPP_WAVE_SHR1, D>(sh[0]);
        const float v2 = sh[2] + dpp_zero_n<BXI_DPP_WAVE_SHL1, D>(sh[1]) + dpp_zero_n<BXI_DPP_WAVE_SHR1, D>(sh[3]);
        if (lr2 + D < TRT) Vp[(lr2 + D) * TC + lc2] = v2;                 // wave-uniform
    }
    PWT(5);
    // ---- four adjacent pixels, their four later pairs each.  Two stages -- the pairs within the row, then those with the row D below
    float acc[4];
    {
        auto window = [&](const float* pl, int row, float (&v)[NWp]) {  // staged row `row`, columns c - D .. c + 3 + D of one plane
#pragma unroll
            for (int x4 = 0; x4 < NWp / 4; ++x4) {
                const float4 a = *reinterpret_cast<const float4*>(pl + row * PC + lc + 4 * x4);
                v[4 * x4] = a.x; v[4 * x4 + 1] = a.y; v[4 * x4 + 2] = a.z; v[4 * x4 + 3] = a.w;
            }
        };
        float sp[4], mp[4], tp[4], hz[4];
        {
            float sr[NWp], mr[NWp];
            window(ts, lr + D, sr); window(tm, lr + D, mr);
            const float o4[4] = {own[0].x, own[0].y, own[0].z, own[0].w}, p4[4] = {part[0].x, part[0].y, part[0].z, part[0].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sp[i] = sr[D + i]; mp[i] = mr[D + i]; tp[i] = sp[i] - mp[i];
                const float sq = sr[2 * D + i], mq = mr[2 * D + i];
                const float G = live && c + i + D < W ? o4[i] + p4[i] : 0.f;
                const float cc = G * fast_rcp(sp[i] * sq + mp[i] * mq);  // S >= 3e-15: every |logit| <= 34
                acc[i] = (sq - mq) * cc;                                 // p's share of the pair
                hz[i] = tp[i] * cc;                                      // q's
            }
        }
        float dn[4 + 2 * D];
#pragma unroll
        for (int j = 0; j < 4 + 2 * D; ++j) dn[j] = 0.f;
        {
            float sd[NWp], md[NWp];
            window(ts, lr + 2 * D, sd); window(tm, lr + 2 * D, md);
            const bool row_d = live && r + D < H;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const float o4[4] = {own[j].x, own[j].y, own[j].z, own[j].w}, p4[4] = {part[j].x, part[j].y, part[j].z, part[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = (j - 1) * D + i;                       // the partner's column in the window
                    const bool in = row_d && (j == 1 ? c + i - D >= 0 : (j == 3 ? c + i + D < W : true));
                    const float G = in ? o4[i] + p4[i] : 0.f;
                    const float cc = G * fast_rcp(sp[i] * sd[q] + mp[i] * md[q]);
                    acc[i] += (sd[q] - md[q]) * cc;
                    dn[q] += tp[i] * cc;
                }
            }
        }
        // the row share: from the thread's own earlier pixel, or the previous lane's
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += i >= D ? hz[i - D] : dpp_zero<BXI_DPP_ROW_SHR1>(hz[i + 4 - D]);
        // the share for the row below: own columns + the next lane's left spill + the previous lane's right spill
        float vd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vd[i] = dn[D + i];
            if (i >= 4 - D) vd[i] += dpp_zero<BXI_DPP_ROW_SHL1>(dn[i - (4 - D)]);
            if (i < D) vd[i] += dpp_zero<BXI_DPP_ROW_SHR1>(dn[D + 4 + i]);
        }
        if (XR >= D || lr + D < TRT) *reinterpret_cast<float4*>(Vp + (lr + D) * TC + lc) = make_float4(vd[0], vd[1], vd[2], vd[3]);
    }
    PWT(6);
    lds_barrier();
    // ---- every pixel: own taps + the row share (above), + the share from the row above, + the shares from outside the tile (taps in order) ; x -u
    // (positions derived again from the thread index: nothing but the sums above stays in registers across the passes)
    asm volatile("" : "+v"(th));
    {
        const int gr = th / (TC / 4), gc = (th % (TC / 4)) * 4;
        if (r0 + gr < H && c0 + gc < W) {
            float e4[4] = {0.f, 0.f, 0.f, 0.f};
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= D) {
                v = *reinterpret_cast<const float4*>(Vp + gr * TC + gc);
                if (gc == 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) e4[i] = Es[(0 * TRT + gr) * 4 + i] + Es[(1 * TRT + gr) * 4 + i];
                }
                if (gc == TC - 4) {                                      // (D == 4: a 64-column tile's first and last quad differ)
#pragma unroll
                    for (int i = 4 - D; i < 4; ++i) e4[i] = Es[(2 * TRT + gr) * 4 + i - (4 - D)];
                }
            } else {
                const float4 e0 = *reinterpret_cast<const float4*>(Et + (0 * D + gr) * TC + gc), e1 = *reinterpret_cast<const float4*>(Et + (1 * D + gr) * TC + gc),
                             e2 = *reinterpret_cast<const float4*>(Et + (2 * D + gr) * TC + gc);
                e4[0] = (e0.x + e1.x) + e2.x; e4[1] = (e0.y + e1.y) + e2.y; e4[2] = (e0.z + e1.z) + e2.z; e4[3] = (e0.w + e1.w) + e2.w;
                if (gc == 0) {                                           // the left neighbour (tap 3) of the first D columns is outside the tile too
#pragma unroll
                    for (int i = 0; i < D; ++i) e4[i] += Es[(1 * TRT + gr) * 4 + i];
                }
            }
            const float v4[4] = {v.x, v.y, v.z, v.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = ts[(gr + D) * PC + gc + D + i] * tm[(gr + D) * PC + gc + D + i];
                o[i] = -u * ((acc[i] + v4[i]) + e4[i]);
            }
            const f4a ov = {o[0], o[1], o[2], o[3]};                     // (non-temporal: nobody in this launch reads the output)
            __builtin_nontemporal_store(ov, reinterpret_cast<f4a*>(g_logits + n * P + (int64_t)(r0 + gr) * W + c0 + gc));
        }
    }
    if (XR > 0) {
        const int gr = TR + th / TC, gc = th % TC;
        if (r0 + gr < H && c0 + gc < W) {
            const float v = Vp[gr * TC + gc];
            float e = 0.f;
            if (gc < D) e = Es[(0 * TRT + gr) * 4 + gc] + Es[(1 * TRT + gr) * 4 + gc];
            if (gc >= TC - D) e = Es[(2 * TRT + gr) * 4 + gc - (TC - D)];
            const float u = ts[(gr + D) * PC + gc + D] * tm[(gr + D) * PC + gc + D];
            __builtin_nontemporal_store(-u * ((acc2 + v) + e), g_logits + n * P + (int64_t)(r0 + gr) * W + c0 + gc);
        }
    }
    PWT(7);
    PWT_FLUSH(1);
}

template <typename T>
static size_t pw3_lds(int d) { return sizeof(LogPair<T>) * (size_t)(kPwTR + 2 * d) * (kPwTC + 2 * d); }
template <typename T>
static size_t pw3_bwd_lds(int d) { return sizeof(LogTriple<T>) * (size_t)(kPwTR + 2 * d) * (kPwTC + 2 * d) + sizeof(T) * (kPwTR * kPwTC * 4 + 64); }

template <typename T>
static int launch_fwd(const T* logits, int N, int H, int W, int size, int dil, T* out, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0 || size == 1) return BXI_OK;
    if (!logits || !out) return BXI_ERR_NULL_POINTER;
    const int64_t total = (int64_t)N * H * W;
    if (!fits_i32(total * (size * size - 1))) return BXI_ERR_BAD_SHAPE;
    const int block = 256;
    if (size == 3 && dil <= kPwMaxDil) {
        const int64_t tiles = (int64_t)N * ((H + kPwTR - 1) / kPwTR) * ((W + kPwTC - 1) / kPwTC);
        if (fits_i32(tiles)) {
            if (sizeof(T) == 4 && (int64_t)8 * H * W * 4 < ((int64_t)1 << 31)) {      // the fast kernels address an instance's planes by 32-bit byte offsets
                const dim3 g((unsigned)tiles), b(block);
                const size_t lds = pw3_lds<T>(dil);
                hipStream_t st = as_stream(stream);
                const bool wide = dil <= 4 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
                if (wide) {
#define BXI_PWF(DD)                                                                                                                             \
                    {                                                                                                                           \
                        const size_t ldw = 2 * sizeof(float) * (size_t)(kPwTR + 2 * DD) * PwGeom<DD, kPwTC>::PC;                             \
                        BXI_LAUNCH("pairwise_fwd", st, (pairwise3_fwd_wide_kernel<DD, kPwTR, kPwTC>), g, b, ldw, st, (const float*)logits, H, W, (float*)out); \
                    }
                    switch (dil) { case 1: BXI_PWF(1) break; case 2: BXI_PWF(2) break; case 3: BXI_PWF(3) break; default: BXI_PWF(4) break; }
#undef BXI_PWF
                    return check_launch();
                }
                switch (dil) {
                    case 1: BXI_LAUNCH("pairwise_fwd", st, pairwise3_fwd_fast_kernel<1>, g, b, lds, st, (const float*)logits, H, W, dil, (float*)out); break;
                    case 2: BXI_LAUNCH("pairwise_fwd", st, pairwise3_fwd_fast_kernel<2>, g, b, lds, st, (const float*)logits, H, W, dil, (float*)out); break;
                    case 3: BXI_LAUNCH("pairwise_fwd", st, pairwise3_fwd_fast_kernel<3>, g, b, lds, st, (const float*)logits, H, W, dil, (float*)out); break;
                    case 4: BXI_LAUNCH("pairwise_fwd", st, pairwise3_fwd_fast_kernel<4>, g, b, lds, st, (const float*)logits, H, W, dil, (float*)out); break;
                    default: BXI_LAUNCH("pairwise_fwd", st, pairwise3_fwd_fast_kernel<0>, g, b, lds, st, (const float*)logits, H, W, dil, (float*)out); break;
                }
            } else
                BXI_LAUNCH("pairwise_fwd", as_stream(stream), (pairwise3_fwd_kernel<T>), dim3((unsigned)tiles), dim3(block), pw3_lds<T>(dil),
                           as_stream(stream), logits, H, W, dil, out);
            return check_launch();
        }
    }
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 64) grid = 256 * 64;  // 256 CUs x 8 waves/SIMD; grid-stride the rest
    BXI_LAUNCH("pairwise_fwd", as_stream(stream), (pairwise_fwd_kernel<T>), dim3((unsigned)grid), dim3(block), 0, as_stream(stream), logits,
                       N, H, W, size, dil, out);
    return check_launch();
}

template <typename T>
static int launch_bwd(const T* logits, const T* g_pair, int N, int H, int W, int size, int dil, T* g_logits,
                      void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0) return BXI_OK;
    if (!logits || !g_logits || (size > 1 && !g_pair)) return BXI_ERR_NULL_POINTER;
    const int64_t total = (int64_t)N * H * W;
    if (!fits_i32(total * (size * size - 1 > 0 ? size * size - 1 : 1))) return BXI_ERR_BAD_SHAPE;
    const int block = 256;
    if (size == 3 && dil <= kPwMaxDil && (int64_t)8 * H * W * (int64_t)sizeof(T) < ((int64_t)1 << 31)) {      // the tiled kernel addresses an instance's planes by 32-bit byte offsets
        const int64_t tiles = (int64_t)N * ((H + kPwTR - 1) / kPwTR) * ((W + kPwTC - 1) / kPwTC);
        if (fits_i32(tiles)) {
            hipStream_t st = as_stream(stream);
            const size_t lds = pw3_bwd_lds<T>(dil);
            const dim3 g((unsigned)tiles), b(block);
#define BXI_PW3_BWD(DD)                                                                                                                        \
            {                                                                                                                                  \
                if (lds > 64 * 1024) {                                                                                                         \
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise3_bwd_kernel<T, DD>),                             \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
                    if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }                                                \
                }                                                                                                                              \
                BXI_LAUNCH("pairwise_bwd", st, (pairwise3_bwd_kernel<T, DD>), g, b, lds, st, logits, g_pair, H, W, dil, g_logits);             \
            }
            if constexpr (sizeof(T) == 4) {
                // (H >= 2: a tap's offset stays below one plane, so that every address of the pair kernel lies inside the instance's 8 planes)
                if (dil <= 4 && (W & 3) == 0 && H >= 2 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(g_pair) | reinterpret_cast<uintptr_t>(g_logits)) & 15) == 0) {
                    // 16-row tiles, or 20-row tiles (16 rows of four pixels per thread + 4 rows of one): whichever needs fewer residency
                    // rounds x rows (five workgroups per CU), then whichever pads the map's height less.  32 x 200 x 256: 1664 workgroups
                    // = 2 rounds of 16 rows against 1280 = ONE round of 20.
                    constexpr int kTR = 16, kTC = 64, kXR = 256 / kTC;
                    const int64_t cols_b = (W + kTC - 1) / kTC;
                    const int64_t t16 = (int64_t)N * ((H + kTR - 1) / kTR) * cols_b, t20 = (int64_t)N * ((H + kTR + kXR - 1) / (kTR + kXR)) * cols_b;
                    const int64_t slots_b = (int64_t)BXI_PWP_OCC * device_cus();
                    const int64_t c16 = ((t16 + slots_b - 1) / slots_b) * kTR, c20 = ((t20 + slots_b - 1) / slots_b) * (kTR + kXR);
                    const int64_t p16 = (int64_t)((H + kTR - 1) / kTR) * kTR, p20 = (int64_t)((H + kTR + kXR - 1) / (kTR + kXR)) * (kTR + kXR);
                    const bool tall = c20 < c16 || (t16 <= slots_b && t20 <= slots_b && p20 < p16);
                    const int64_t tiles_b = tall ? t20 : t16;
                    if (!fits_i32(tiles_b)) return BXI_ERR_BAD_SHAPE;
                    const dim3 gb((unsigned)tiles_b);
                    // tile order: groups of two tile rows on one XCD, consecutive groups on consecutive XCDs (kernel: xcd_swizzle = 1 + rows per group)
                    constexpr int kSwz = 3;
#define BXI_PWB(DD)                                                                                                                             \
                    {                                                                                                                           \
                        if (tall)                                                                                                               \
                            BXI_LAUNCH("pairwise_bwd", st, (pairwise3_bwd_pair_kernel<DD, kXR>), gb, b, (PwPairGeom<DD, kXR>::lds_bytes), st,   \
                                       (const float*)logits, (const float*)g_pair, H, W, (float*)g_logits, kSwz);                               \
                        else                                                                                                                    \
                            BXI_LAUNCH("pairwise_bwd", st, (pairwise3_bwd_pair_kernel<DD, 0>), gb, b, (PwPairGeom<DD, 0>::lds_bytes), st,       \
                                       (const float*)logits, (const float*)g_pair, H, W, (float*)g_logits, kSwz);                               \
                    }
                    switch (dil) { case 1: BXI_PWB(1) break; case 2: BXI_PWB(2) break; case 3: BXI_PWB(3) break; default: BXI_PWB(4) break; }
#undef BXI_PWB
                    return check_launch();
                }
                if (dil == 1) BXI_PW3_BWD(1)
                else if (dil == 2) BXI_PW3_BWD(2)
                else if (dil == 3) BXI_PW3_BWD(3)
                else if (dil == 4) BXI_PW3_BWD(4)
                else BXI_PW3_BWD(0)
            } else BXI_PW3_BWD(0)
#undef BXI_PW3_BWD
            return check_launch();
        }
    }
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 64) grid = 256 * 64;
    BXI_LAUNCH("pairwise_bwd", as_stream(stream), (pairwise_bwd_kernel<T>), dim3((unsigned)grid), dim3(block), 0, as_stream(stream), logits,
                       g_pair, N, H, W, size, dil, g_logits);
    return check_launch();
}

}  // namespace bxi

extern "C" {

int bxi_pairwise_nlog_forward_f32(const float* logits, int N, int H, int W, int size, int dilation,
                                  float* pairwise, void* stream) {
    return bxi::launch_fwd<float>(logits, N, H, W, size, dilation, pairwise, stream);
}
int bxi_pairwise_nlog_forward_f64(const double* logits, int N, int H, int W, int size, int dilation,
                                  double* pairwise, void* stream) {
    return bxi::launch_fwd<double>(logits, N, H, W, size, dilation, pairwise, stream);
}
int bxi_pairwise_nlog_backward_f32(const float* logits, const float* pairwise, const float* g_pairwise, int N,
                                   int H, int W, int size, int dilation, float* g_logits, void* stream) {
    (void)pairwise;
    return bxi::launch_bwd<float>(logits, g_pairwise, N, H, W, size, dilation, g_logits, stream);
}
int bxi_pairwise_nlog_backward_f64(const double* logits, const double* pairwise, const double* g_pairwise, int N,
                                   int H, int W, int size, int dilation, double* g_logits, void* stream) {
    (void)pairwise;
    return bxi::launch_bwd<double>(logits, g_pairwise, N, H, W, size, dilation, g_logits, stream);
}

}  // extern "C"
