// levelset.hip -- SURVEY 8(f-4): Box2Mask / BoxLevelSet loss pieces on gfx950
// (mmdet/models/losses/levelset_loss.py: LevelsetLoss :7-18, region_levelset :21-45,
//  LocalConsistencyModule :63-126; BoxProjectionLoss lives with mil_loss in meanfield.hip).
//
// region_levelset: per instance and side (foreground / background score) the region mean of every target channel,
//   a_c = sum(f T_c) / max(sum f, 1e-5), and the energy sum_c sum_p (T_c - a_c)^2 f_p.  One pass (8 workgroups per
//   instance, partial sums combined in index order by a second small launch): the energy is
//   Q_c - 2 a_c A_c + a_c^2 S with A = sum f T, Q = sum f T^2, accumulated in fp64 (the reference makes two passes and
//   materialises [N,C,H,W] temporaries).  The backward is elementwise from the saved sums.
// LocalConsistencyModule: an 8-neighbour (dilated, replicate-padded) affinity from the image and `iters` applications
//   of the linear operator phi <- sum_k aff_k * phi(neighbour_k).  A map that fits LDS (96x96 in the reference) is
//   refined by ONE workgroup in ONE launch, ping-ponging between two LDS planes; the backward is the adjoint operator,
//   written as a gather (every (p,k) that lands on q after clamping is enumerated) so that it needs no atomics.
#include "common.hpp"

namespace bxi {

constexpr int kLsMaxC = 8;        // target channels a launch of the partial sums keeps in registers; more channels = more launches (groups of 8)

__device__ __forceinline__ double block_sum_f64_ls(double v, double* red /*[16]*/) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];   // fixed order
    return s;
}

// state per instance: S[2] | a[2][C] | R[2][C]   (doubles; R = A - a S, zero unless the clamp of :34-35 is active),
// followed (after all instances) by the partial sums of the slices: [N][kLsSlices][2 + 4C]  = S[2] | A[2][C] | Q[2][C]
__device__ __forceinline__ int ls_stride(int C) { return 2 + 4 * C; }
constexpr int kLsSlices = 8;      // workgroups per instance in the forward (a single one per instance is latency bound)

// (512 threads: 2 + 4 C double-precision running sums per thread, 34 at C = 8, next to the C + 2 float4 in flight do not fit the 128 registers of a
// 1024-thread workgroup)
constexpr int kLsThreads = 512;
template <bool VEC>
__global__ __launch_bounds__(kLsThreads) void levelset_partial_kernel(const float* __restrict__ ms, const float* __restrict__ tg, int N, int C,
                                                                int H, int W, double* __restrict__ state, int c0) {
    __shared__ double red[16 * (2 + 4 * kLsMaxC)];
    const int n = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    const int64_t chunk = ((HW + kLsSlices - 1) / kLsSlices + 3) & ~(int64_t)3;
    const int64_t lo = sl * chunk, hi = lo + chunk < HW ? lo + chunk : HW;
    const float* f0 = ms + (int64_t)n * 2 * HW;
    const float* T = tg + ((int64_t)n * C + c0) * HW;              // this launch: channels c0 .. c0 + Cg - 1
    const int Cg = C - c0 < kLsMaxC ? C - c0 : kLsMaxC;
    double S[2] = {0.0, 0.0}, A[2][kLsMaxC], Q[2][kLsMaxC];
#pragma unroll
    for (int c = 0; c < kLsMaxC; ++c) { A[0][c] = A[1][c] = Q[0][c] = Q[1][c] = 0.0; }
    if (VEC) {
        // H*W a multiple of 4 and 16-byte aligned planes (host check): 4 consecutive pixels per thread and trip as float4
        // loads (2 + C wave loads of 1 KiB instead of 8 + 4C of 256 B); `chunk` and `hi` are multiples of 4 then
        for (int64_t p0 = lo + 4 * tid; p0 < hi; p0 += 4 * kLsThreads) {
            const float4 f4 = *reinterpret_cast<const float4*>(f0 + p0), g4 = *reinterpret_cast<const float4*>(f0 + HW + p0);
            float4 t4[kLsMaxC];
#pragma unroll
            for (int c = 0; c < kLsMaxC; ++c)
                if (c < Cg) t4[c] = *reinterpret_cast<const float4*>(T + (int64_t)c * HW + p0);
            const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) { S[0] += (double)fv[u]; S[1] += (double)gv[u]; }
#pragma unroll
            for (int c = 0; c < kLsMaxC; ++c)
                if (c < Cg) {
                    const float tv[4] = {t4[c].x, t4[c].y, t4[c].z, t4[c].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double f = fv[u], g = gv[u], t = tv[u];
                        A[0][c] += f * t; A[1][c] += g * t;
                        Q[0][c] += f * t * t; Q[1][c] += g * t * t;
                    }
                }
        }
    } else
    for (int64_t p0 = lo + tid; p0 < hi; p0 += 4 * kLsThreads) {        // 4 pixels per trip: their loads are independent
        // unconditional loads at a clamped index (a guarded load becomes a branch, and branches serialise the loads);
        // a pixel past the end gets zero scores, which add nothing to any sum
        float fv[4], gv[4];
        int64_t pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t p = p0 + u * kLsThreads; pc[u] = p < hi ? p : hi - 1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { fv[u] = f0[pc[u]]; gv[u] = f0[HW + pc[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u * kLsThreads >= hi) { fv[u] = 0.f; gv[u] = 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { S[0] += (double)fv[u]; S[1] += (double)gv[u]; }
#pragma unroll
        for (int c = 0; c < kLsMaxC; ++c)
            if (c < Cg) {
                float tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) tv[u] = T[(int64_t)c * HW + pc[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double f = fv[u], g = gv[u], t = tv[u];
                    A[0][c] += f * t; A[1][c] += g * t;
                    Q[0][c] += f * t * t; Q[1][c] += g * t * t;
                }
            }
    }
    // All 2 + 4C sums are reduced TOGETHER: their wave butterflies are independent chains (one after another, each is six
    // dependent cross-lane steps and two barriers: ~1 us per sum, 14 sums at C = 3), one LDS hand-over, one barrier, then
    // thread q adds the 16 wave partials of sum q in wave order -- the same additions in the same order as before.
    double* part = state + (int64_t)N * ls_stride(C) + ((int64_t)n * kLsSlices + sl) * ls_stride(C);
    const int n_sums = ls_stride(Cg);
    S[0] = wave_sum_f64(S[0]); S[1] = wave_sum_f64(S[1]);
#pragma unroll
    for (int c = 0; c < kLsMaxC; ++c)
        if (c < Cg) {
            A[0][c] = wave_sum_f64(A[0][c]); A[1][c] = wave_sum_f64(A[1][c]);
            Q[0][c] = wave_sum_f64(Q[0][c]); Q[1][c] = wave_sum_f64(Q[1][c]);
        }
    if ((tid & 63) == 0) {
        double* r = red + (tid >> 6) * (2 + 4 * kLsMaxC);
        r[0] = S[0]; r[1] = S[1];
#pragma unroll
        for (int c = 0; c < kLsMaxC; ++c)
            if (c < Cg) {
                r[2 + c] = A[0][c]; r[2 + Cg + c] = A[1][c];
                r[2 + 2 * Cg + c] = Q[0][c]; r[2 + 3 * Cg + c] = Q[1][c];
            }
    }
    __syncthreads();
    if (tid < n_sums) {
        double s = 0.0;
        for (int wv = 0; wv < kLsThreads / 64; ++wv) s += red[wv * (2 + 4 * kLsMaxC) + tid];     // fixed order
        // sum `tid` of this group -> its place among all C channels (S[2] is the same in every group's launch: same data, same order)
        const int q = tid - 2;
        part[tid < 2 ? tid : 2 + (q / Cg) * C + c0 + q % Cg] = s;
    }
}

// slices summed in index order -> region means, energy, loss; one thread per instance
__global__ __launch_bounds__(64) void levelset_finish_kernel(const float* __restrict__ pixel_num, int N, int C, double weight,
                                                             float* __restrict__ loss, double* __restrict__ state) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const double* part = state + (int64_t)N * ls_stride(C) + (int64_t)n * kLsSlices * ls_stride(C);
    double* st = state + (int64_t)n * ls_stride(C);
    double total = 0.0;
    for (int side = 0; side < 2; ++side) {
        double s = 0.0;
        for (int k = 0; k < kLsSlices; ++k) s += part[k * ls_stride(C) + side];
        const double sc = s > 1e-5 ? s : 1e-5;                   // .clamp(min=0.00001)
        st[side] = s;
        for (int c = 0; c < C; ++c) {
            double a_sum = 0.0, q_sum = 0.0;
            for (int k = 0; k < kLsSlices; ++k) { a_sum += part[k * ls_stride(C) + 2 + side * C + c]; q_sum += part[k * ls_stride(C) + 2 + 2 * C + side * C + c]; }
            const double a = a_sum / sc;
            total += q_sum - 2.0 * a * a_sum + a * a * s;
            st[2 + side * C + c] = a; st[2 + 2 * C + side * C + c] = a_sum - a * s;
        }
    }
    loss[n] = (float)(weight * total / ((double)C * (double)pixel_num[n]));
}

__global__ __launch_bounds__(256) void levelset_bwd_kernel(const float* __restrict__ ms, const float* __restrict__ tg,
                                                           const float* __restrict__ pixel_num, int N, int C, int H, int W,
                                                           double weight, const double* __restrict__ state,
                                                           const float* __restrict__ g_loss, float* __restrict__ g_ms,
                                                           float* __restrict__ g_tg) {
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * HW) return;
    const int n = (int)(i / HW);
    const int64_t p = i % HW;
    const double* st = state + (int64_t)n * ls_stride(C);
    const double w = (double)g_loss[n] * weight / ((double)C * (double)pixel_num[n]);
    const float* f0 = ms + (int64_t)n * 2 * HW;
    double gt[kLsMaxC];
#pragma unroll
    for (int c = 0; c < kLsMaxC; ++c) gt[c] = 0.0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const double f = f0[side * HW + p];
        const double s = st[side], sc = s > 1e-5 ? s : 1e-5;
        // one fp64 division per side instead of two per side and channel (twenty per pixel at C = 5: the kernel was bound by them, not by its bytes);
        // the quotients differ from x / sc by an ulp of a double at most -- far inside the fp32 results' 1e-4
        const double inv = 1.0 / sc, fi = f * inv;
        double gm = 0.0;
#pragma unroll
        for (int c = 0; c < kLsMaxC; ++c)
            if (c < C) {
                const double t = tg[((int64_t)n * C + c) * HW + p];
                const double a = st[2 + side * C + c], R = st[2 + 2 * C + side * C + c];
                const double d = t - a;
                const double da_df = (t - (s >= 1e-5 ? a : 0.0)) * inv;
                gm += d * d - 2.0 * R * da_df;
                gt[c] += 2.0 * d * f - 2.0 * R * fi;
            }
        g_ms[((int64_t)n * 2 + side) * HW + p] = (float)(w * gm);
    }
    if (g_tg) {
#pragma unroll
        for (int c = 0; c < kLsMaxC; ++c)
            if (c < C) g_tg[((int64_t)n * C + c) * HW + p] = (float)(w * gt[c]);
    }
}

// more than kLsMaxC target channels: the same sums in the same order (per side over c; per channel side 0 then side 1), channel by channel
__global__ __launch_bounds__(256) void levelset_bwd_anyc_kernel(const float* __restrict__ ms, const float* __restrict__ tg,
                                                                const float* __restrict__ pixel_num, int N, int C, int H, int W,
                                                                double weight, const double* __restrict__ state,
                                                                const float* __restrict__ g_loss, float* __restrict__ g_ms,
                                                                float* __restrict__ g_tg) {
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * HW) return;
    const int n = (int)(i / HW);
    const int64_t p = i % HW;
    const double* st = state + (int64_t)n * ls_stride(C);
    const double w = (double)g_loss[n] * weight / ((double)C * (double)pixel_num[n]);
    const float* f0 = ms + (int64_t)n * 2 * HW;
    const double f[2] = {(double)f0[p], (double)f0[HW + p]};
    const double s[2] = {st[0], st[1]};
    const double sc[2] = {s[0] > 1e-5 ? s[0] : 1e-5, s[1] > 1e-5 ? s[1] : 1e-5};
    const double inv[2] = {1.0 / sc[0], 1.0 / sc[1]}, fi[2] = {f[0] * inv[0], f[1] * inv[1]};     // (one division per side: levelset_bwd_kernel)
    double gm[2] = {0.0, 0.0};
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const double t = tg[((int64_t)n * C + c) * HW + p];
        double gt = 0.0;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const double a = st[2 + side * C + c], R = st[2 + 2 * C + side * C + c];
            const double d = t - a;
            const double da_df = (t - (s[side] >= 1e-5 ? a : 0.0)) * inv[side];
            gm[side] += d * d - 2.0 * R * da_df;
            gt += 2.0 * d * f[side] - 2.0 * R * fi[side];
        }
        if (g_tg) g_tg[((int64_t)n * C + c) * HW + p] = (float)(w * gt);
    }
    g_ms[((int64_t)n * 2 + 0) * HW + p] = (float)(w * gm[0]);
    g_ms[((int64_t)n * 2 + 1) * HW + p] = (float)(w * gm[1]);
}

// ---------------------------------------------------------------------------------------------------
// LocalConsistencyModule
__device__ __forceinline__ void lcm_offset(int k, int& dy, int& dx) {     // get_kernel (:82-92): row-major 3x3 without the centre
    const int kk = k < 4 ? k : k + 1;
    dy = kk / 3 - 1; dx = kk % 3 - 1;
}

__global__ __launch_bounds__(256) void lcm_affinity_kernel(const float* __restrict__ imgs, int N, int C, int h, int w, int d,
                                                           float alpha, float* __restrict__ aff) {
    const int64_t hw = (int64_t)h * w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * hw) return;
    const int n = (int)(i / hw), p = (int)(i % hw), r = p / w, c = p % w;
    // The deviation of a near-constant window is a difference of nearly equal numbers, and the affinity divides by it: in f32 the
    // rounding of mean / variance is amplified into the 5th digit of the softmax (an off-suite fuzz case, 120 x 197 at dilation 3, sat
    // 2.2e-5 from the fp64 oracle).  Eight values per pixel and channel: the statistics are taken in double (fp64 runs at the f32 rate
    // on this part), the exponent and the softmax stay f32 as in the reference (levelset_loss.py:113-120).
    double ed[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ed[k] = 0.0;
    for (int ch = 0; ch < C; ++ch) {
        const float* I = imgs + ((int64_t)n * C + ch) * hw;
        const double ip = (double)I[p];
        double v[8], mean = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int dy, dx; lcm_offset(k, dy, dx);
            const int r2 = min(max(r + dy * d, 0), h - 1), c2 = min(max(c + dx * d, 0), w - 1);   // replicate padding (:99)
            v[k] = (double)I[(int64_t)r2 * w + c2];
            mean += v[k];
        }
        mean *= 0.125;
        double var = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) var += (v[k] - mean) * (v[k] - mean);
        const double sd = sqrt(var / 7.0) + 1e-8;                  // torch.std: unbiased (:116)
        // ONE double-precision division per pixel and channel instead of sixteen (each is ~40 instructions: the kernel took 49 us for 4 MB):
        // |v - ip| / sd / alpha = |v - ip| * (1 / (sd alpha)) to within two roundings of a double, nine digits below the f32 the result becomes
        const double inv = 1.0 / (sd * (double)alpha);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double z = fabs(v[k] - ip) * inv; ed[k] -= z * z; }
    }
    float e[8];
    float m = -INFINITY;
    const double invC = 1.0 / (double)C;
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = (float)(ed[k] * invC); m = fmaxf(m, e[k]); }      // .mean(dim=1)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = expf(e[k] - m); s += e[k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) aff[((int64_t)n * 8 + k) * hw + p] = e[k] / s;
}

// one application of the operator (or of its adjoint) at pixel (r,c); src: a plane of h*w floats (LDS or global)
__device__ __forceinline__ float lcm_apply(const float* __restrict__ A /*[8,h,w] of the instance*/, const float* src, int h, int w,
                                           int d, int r, int c) {
    const int64_t hw = (int64_t)h * w;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int dy, dx; lcm_offset(k, dy, dx);
        const int r2 = min(max(r + dy * d, 0), h - 1), c2 = min(max(c + dx * d, 0), w - 1);
        acc += A[k * hw + (int64_t)r * w + c] * src[r2 * w + c2];
    }
    return acc;
}

// positions p in [0,n) with clamp(p + delta, 0, n-1) == q
__device__ __forceinline__ void lcm_sources(int q, int delta, int n, int& lo, int& hi) {
    if (n == 1) { lo = 0; hi = 0; return; }
    if (q == 0) { lo = 0; hi = min(-delta, n - 1); }
    else if (q == n - 1) { lo = max(n - 1 - delta, 0); hi = n - 1; }
    else { lo = hi = q - delta; if (lo < 0 || lo >= n) { lo = 1; hi = 0; } }
}

__device__ __forceinline__ float lcm_apply_adjoint(const float* __restrict__ A, const float* src, int h, int w, int d, int r, int c) {
    const int64_t hw = (int64_t)h * w;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int dy, dx; lcm_offset(k, dy, dx);
        int rlo, rhi, clo, chi;
        lcm_sources(r, dy * d, h, rlo, rhi);
        lcm_sources(c, dx * d, w, clo, chi);
        for (int pr = rlo; pr <= rhi; ++pr)
            for (int pc = clo; pc <= chi; ++pc) acc += A[k * hw + (int64_t)pr * w + pc] * src[pr * w + pc];
    }
    return acc;
}

// whole map in LDS: all iterations in one launch, one workgroup per instance
template <bool ADJ>
__global__ __launch_bounds__(1024) void lcm_refine_lds_kernel(const float* __restrict__ aff, const float* __restrict__ phi, int h, int w,
                                                              int d, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lcm_planes[];   // [2][h*w]
    const int n = blockIdx.x, tid = threadIdx.x, hw = h * w;
    float* cur = lcm_planes;
    float* nxt = lcm_planes + hw;
    const float* A = aff + (int64_t)n * 8 * hw;
    for (int p = tid; p < hw; p += 1024) cur[p] = phi[(int64_t)n * hw + p];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int p = tid; p < hw; p += 1024) {
            const int r = p / w, c = p % w;
            nxt[p] = ADJ ? lcm_apply_adjoint(A, cur, h, w, d, r, c) : lcm_apply(A, cur, h, w, d, r, c);
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    for (int p = tid; p < hw; p += 1024) out[(int64_t)n * hw + p] = cur[p];
}

// The adjoint, all iterations in one launch.  The forward operator reads a replicate-padded plane:
// out[p] = sum_k aff_k[p] * pad(phi)[p + delta_k + d].  Its adjoint is therefore (a) an 8-tap gather on the PADDED
// domain with no clamping, gp[s] = sum_k aff_k[p_k] g[p_k], p_k = s - d - delta_k (skipped outside the map), followed by
// (b) folding the padding back, g'[q] = sum of gp over the padded positions that replicate q.  Both steps are
// branch-light, read only LDS (plus the 8 coefficient loads, issued together) and have a fixed summation order.
__global__ __launch_bounds__(1024) void lcm_adjoint_lds_kernel(const float* __restrict__ aff, const float* __restrict__ gout, int h, int w,
                                                               int d, int iters, float* __restrict__ gphi) {
    extern __shared__ __attribute__((aligned(16))) float lcm_planes[];   // g [h*w] | gp [(h+2d)*(w+2d)]
    const int n = blockIdx.x, tid = threadIdx.x, hw = h * w, hp = h + 2 * d, wp = w + 2 * d;
    float* g = lcm_planes;
    float* gp = lcm_planes + hw;
    const float* A = aff + (int64_t)n * 8 * hw;
    for (int p = tid; p < hw; p += 1024) g[p] = gout[(int64_t)n * hw + p];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int s = tid; s < hp * wp; s += 1024) {
            const int sr = s / wp, sc = s % wp;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int dy, dx; lcm_offset(k, dy, dx);
                const int pr = sr - d - dy * d, pc = sc - d - dx * d;
                const bool in = pr >= 0 && pr < h && pc >= 0 && pc < w;
                const int pi = in ? pr * w + pc : 0;                      // clamped: the loads carry no branch
                acc += (in ? 1.f : 0.f) * (A[(int64_t)k * hw + pi] * g[pi]);
            }
            gp[s] = acc;
        }
        __syncthreads();
        for (int q = tid; q < hw; q += 1024) {
            const int r = q / w, c = q % w;
            const int r_lo = r == 0 ? 0 : r + d, r_hi = r == h - 1 ? hp - 1 : r + d;      // padded rows that replicate row r
            const int c_lo = c == 0 ? 0 : c + d, c_hi = c == w - 1 ? wp - 1 : c + d;
            float acc = 0.f;
            for (int sr = r_lo; sr <= r_hi; ++sr)
                for (int sc = c_lo; sc <= c_hi; ++sc) acc += gp[sr * wp + sc];
            g[q] = acc;
        }
        __syncthreads();
    }
    for (int p = tid; p < hw; p += 1024) gphi[(int64_t)n * hw + p] = g[p];
}

// ---- padded maps of at most kLcmPadSPT * 1024 positions (96 x 96, d = 2 in the reference), all iterations in one launch ----
// Both directions work on REPLICATE-PADDED planes in LDS, so a tap is `plane[base + constant]`: no clamping arithmetic in
// the iterations (the clamps were most of the instructions: 8 taps x 18 pixels x ~8 integer operations a thread and
// iteration).  The 8 coefficients of a thread's positions stay in registers for all iterations.  H, W, D > 0 fix the shape
// at compile time (the tap offsets become instruction immediates); 0 = run-time shape.
constexpr int kLcmPadThreads = 1024;
constexpr int kLcmPadSPT = 10;         // padded positions per thread: 10 x 1024 >= 100 x 100

// the padded positions that replicate pixel (r, c): rows [rlo, rhi] x columns [clo, chi] of the (h + 2d) x (w + 2d) plane
__device__ __forceinline__ void lcm_replicas(int r, int c, int h, int w, int d, int& rlo, int& rhi, int& clo, int& chi) {
    rlo = r == 0 ? 0 : r + d; rhi = r == h - 1 ? h + 2 * d - 1 : r + d;
    clo = c == 0 ? 0 : c + d; chi = c == w - 1 ? w + 2 * d - 1 : c + d;
}

// Forward: EVERY position of the padded plane is computed (a padding cell repeats the arithmetic of the pixel it replicates:
// 8.5 % more work at 96 x 96, d = 2), so the iterations carry no border case and one barrier each.
template <int H, int W, int D>
__global__ __launch_bounds__(kLcmPadThreads) void lcm_refine_pad_kernel(const float* __restrict__ aff, const float* __restrict__ phi,
                                                                        int h_, int w_, int d_, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lcm_planes[];   // [2][(h + 2d) * (w + 2d)]
    const int h = H ? H : h_, w = W ? W : w_, d = D ? D : d_;
    const int n = blockIdx.x, tid = threadIdx.x, hw = h * w, hp = h + 2 * d, wp = w + 2 * d, hpwp = hp * wp;
    float* cur = lcm_planes;
    float* nxt = lcm_planes + hpwp;
    const float* A = aff + (int64_t)n * 8 * hw;
    float coef[kLcmPadSPT][8];
    int base[kLcmPadSPT];           // padded index of the top-left tap of the pixel this position replicates; -1: no position
#pragma unroll
    for (int j = 0; j < kLcmPadSPT; ++j) {
        const int s = tid + j * kLcmPadThreads;
        base[j] = -1;
        if (s < hpwp) {
            const int r = min(max(s / wp - d, 0), h - 1), c = min(max(s % wp - d, 0), w - 1);     // replicate padding (:99)
            base[j] = r * wp + c;
#pragma unroll
            for (int k = 0; k < 8; ++k) coef[j][k] = A[(int64_t)k * hw + r * w + c];
            cur[s] = phi[(int64_t)n * hw + r * w + c];
        }
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kLcmPadSPT; ++j) {
            if (base[j] < 0) continue;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int dy, dx; lcm_offset(k, dy, dx);
                acc += coef[j][k] * cur[base[j] + (dy + 1) * d * wp + (dx + 1) * d];
            }
            nxt[tid + j * kLcmPadThreads] = acc;
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    for (int p = tid; p < hw; p += kLcmPadThreads) out[(int64_t)n * hw + p] = cur[(p / w + d) * wp + p % w + d];
}

// The adjoint.  The forward reads pad(phi)[p + delta_k] with coefficient aff_k[p]; its adjoint is (a) a gather on the PADDED
// domain, gp[s] = sum_k aff_k[p_k] g[p_k] with p_k = s - d - delta_k (nothing where p_k falls outside the map), and (b)
// folding the padding back, g'[q] = sum of gp over the padded positions that replicate q (rows, then columns, ascending).
// The transposed coefficients aff_k[p_k(s)] of a thread's padded positions are fetched once; g lives in a ZERO-padded plane
// (2d on every side), so (a) is again `plane[base + constant]` without a branch.  Summation orders = lcm_adjoint_lds_kernel's.
template <int H, int W, int D>
__global__ __launch_bounds__(kLcmPadThreads) void lcm_adjoint_pad_kernel(const float* __restrict__ aff, const float* __restrict__ gout,
                                                                         int h_, int w_, int d_, int iters, float* __restrict__ gphi) {
    extern __shared__ __attribute__((aligned(16))) float lcm_planes[];   // gz [(h + 4d) * (w + 4d)] | gp [(h + 2d) * (w + 2d)]
    const int h = H ? H : h_, w = W ? W : w_, d = D ? D : d_;
    const int n = blockIdx.x, tid = threadIdx.x, hw = h * w, hp = h + 2 * d, wp = w + 2 * d, wq = w + 4 * d, nq = (h + 4 * d) * wq;
    float* gz = lcm_planes;
    float* gp = lcm_planes + nq;
    const float* A = aff + (int64_t)n * 8 * hw;
    for (int i = tid; i < nq; i += kLcmPadThreads) gz[i] = 0.f;
    // The transposed coefficients live in registers for the reference's shape (96 x 96, d = 2: everything about an index is a compile-time
    // constant).  The run-time-shape variant would need 80 registers for them next to run-time strides: it re-reads them (L2-resident,
    // 32 B per pixel and instance) in every iteration instead of spilling them -- the same values in the same order.
    constexpr bool kRegCoef = H != 0;
    float coef[kRegCoef ? kLcmPadSPT : 1][8];
    int sbase[kLcmPadSPT];          // index into gz of the top-left-most source of padded position s; -1: no position
#pragma unroll
    for (int j = 0; j < kLcmPadSPT; ++j) {
        const int s = tid + j * kLcmPadThreads;
        sbase[j] = -1;
        if (s < hp * wp) {
            const int sr = s / wp, sc = s % wp;
            // source of tap k: pixel (sr - (dy + 1) d, sc - (dx + 1) d); in gz (shifted by 2d): (sr + (1 - dy) d, sc + (1 - dx) d)
            sbase[j] = sr * wq + sc;
            if constexpr (kRegCoef) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int dy, dx; lcm_offset(k, dy, dx);
                    const int pr = sr - (dy + 1) * d, pc = sc - (dx + 1) * d;
                    const bool in = pr >= 0 && pr < h && pc >= 0 && pc < w;
                    coef[j][k] = in ? A[(int64_t)k * hw + pr * w + pc] : 0.f;
                }
            }
        }
    }
    // fold-back bookkeeping, once: an interior pixel copies its one padded position; perimeter pixel number tid of the walk
    // (top row, bottom row, the two side columns) sums the padded positions that replicate it
    constexpr int kPix = kLcmPadSPT - 1;                 // pixels per thread: hw <= hp * wp - ... <= 9 x 1024 when the padded plane fits
    int gsrc[kPix], gdst[kPix];                          // gp / gz index of an interior pixel; -1: none or perimeter
#pragma unroll
    for (int j = 0; j < kPix; ++j) {
        const int p = tid + j * kLcmPadThreads;
        gsrc[j] = -1; gdst[j] = 0;
        if (p < hw) {
            const int r = p / w, c = p % w;
            if (!(r == 0 || r == h - 1 || c == 0 || c == w - 1)) { gsrc[j] = (r + d) * wp + c + d; gdst[j] = (r + 2 * d) * wq + c + 2 * d; }
        }
    }
    int bdst = -1, b_r = 0, b_c = 0;
    {
        const int n_border = 2 * w + 2 * max(h - 2, 0);
        if (tid < n_border && (h > 1 || tid < w)) {
            int r, c;
            if (tid < w) { r = 0; c = tid; }
            else if (tid < 2 * w) { r = h - 1; c = tid - w; }
            else { const int u = tid - 2 * w; r = 1 + (u >> 1); c = (u & 1) ? w - 1 : 0; }
            if (!(w == 1 && tid >= 2 * w && (tid & 1))) {               // a one-column map: the two side columns coincide
                int rlo, rhi, clo, chi;
                lcm_replicas(r, c, h, w, d, rlo, rhi, clo, chi);
                bdst = (r + 2 * d) * wq + c + 2 * d; b_r = rlo | (rhi << 16); b_c = clo | (chi << 16);
            }
        }
    }
    __syncthreads();                 // the zero-fill is complete before the map goes in
    for (int p = tid; p < hw; p += kLcmPadThreads) gz[(p / w + 2 * d) * wq + p % w + 2 * d] = gout[(int64_t)n * hw + p];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (kRegCoef) {
#pragma unroll
            for (int j = 0; j < kLcmPadSPT; ++j) {
                if (sbase[j] < 0) continue;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int dy, dx; lcm_offset(k, dy, dx);
                    acc += coef[j][k] * gz[sbase[j] + (1 - dy) * d * wq + (1 - dx) * d];
                }
                gp[tid + j * kLcmPadThreads] = acc;
            }
        } else {
#pragma unroll 1
            for (int s = tid; s < hp * wp; s += kLcmPadThreads) {        // (nothing per position is kept across iterations: see kRegCoef)
                const int sr = s / wp, sc = s % wp, sb = sr * wq + sc;
                float cf[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int dy, dx; lcm_offset(k, dy, dx);
                    const int pr = sr - (dy + 1) * d, pc = sc - (dx + 1) * d;
                    const bool in = pr >= 0 && pr < h && pc >= 0 && pc < w;
                    cf[k] = in ? A[(int64_t)k * hw + pr * w + pc] : 0.f;
                }
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int dy, dx; lcm_offset(k, dy, dx);
                    acc += cf[k] * gz[sb + (1 - dy) * d * wq + (1 - dx) * d];
                }
                gp[s] = acc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPix; ++j)
            if (gsrc[j] >= 0) gz[gdst[j]] = 0.f + gp[gsrc[j]];
        if (bdst >= 0) {
            float acc = 0.f;
            if constexpr (D > 0) {
                // at most (D + 1) x (D + 1) replicas: all loads issued together, added in row / column order (an absent one adds
                // +0, which changes nothing); the run-time loops below made six waves walk dependent LDS reads while the
                // other ten waited at the barrier
                float v[(D + 1) * (D + 1)];
#pragma unroll
                for (int i = 0; i <= D; ++i)
#pragma unroll
                    for (int j = 0; j <= D; ++j) {
                        const int sr = (b_r & 0xffff) + i, sc = (b_c & 0xffff) + j;
                        const bool ok = sr <= (b_r >> 16) && sc <= (b_c >> 16);
                        v[i * (D + 1) + j] = ok ? gp[sr * wp + sc] : 0.f;
                    }
#pragma unroll
                for (int k = 0; k < (D + 1) * (D + 1); ++k) acc += v[k];
            } else {
                for (int sr = b_r & 0xffff; sr <= (b_r >> 16); ++sr)
                    for (int sc = b_c & 0xffff; sc <= (b_c >> 16); ++sc) acc += gp[sr * wp + sc];
            }
            gz[bdst] = acc;
        }
        __syncthreads();
    }
    for (int p = tid; p < hw; p += kLcmPadThreads) gphi[(int64_t)n * hw + p] = gz[(p / w + 2 * d) * wq + p % w + 2 * d];
}

// any size: one launch per iteration, planes in global memory
template <bool ADJ>
__global__ __launch_bounds__(256) void lcm_refine_step_kernel(const float* __restrict__ aff, const float* __restrict__ src, int N, int h,
                                                              int w, int d, float* __restrict__ dst) {
    const int64_t hw = (int64_t)h * w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * hw) return;
    const int n = (int)(i / hw), p = (int)(i % hw), r = p / w, c = p % w;
    const float* A = aff + (int64_t)n * 8 * hw;
    const float* s = src + (int64_t)n * hw;
    dst[i] = ADJ ? lcm_apply_adjoint(A, s, h, w, d, r, c) : lcm_apply(A, s, h, w, d, r, c);
}

}  // namespace bxi

extern "C" {

size_t bxi_levelset_state_bytes(int N, int C) {
    if (N < 0 || C <= 0 || C > 4096) return 0;
    return sizeof(double) * (size_t)(N > 0 ? N : 1) * (2 + 4 * C) * (1 + bxi::kLsSlices);
}

int bxi_levelset_loss_forward_f32(const float* mask_score, const float* target, const float* pixel_num, int N, int C, int H,
                                  int W, float loss_weight, float* loss, void* state, void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0) return BXI_ERR_BAD_SHAPE;
    if (C > 4096) return BXI_ERR_UNSUPPORTED;
    if (N == 0) return BXI_OK;
    if (!mask_score || !target || !pixel_num || !loss || !state) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * H * W * (C > 2 ? C : 2))) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 7) return BXI_ERR_WORKSPACE;
    hipStream_t s = bxi::as_stream(stream);
    if (N > 65535) return BXI_ERR_UNSUPPORTED;
    const bool vec = ((int64_t)H * W) % 4 == 0 && ((reinterpret_cast<uintptr_t>(mask_score) | reinterpret_cast<uintptr_t>(target)) & 15) == 0;
    int rc = BXI_OK;
    for (int c0 = 0; c0 < C; c0 += bxi::kLsMaxC) {                  // the shipped losses have C = 3 and C = 5: one launch
        if (vec)
            BXI_LAUNCH("levelset_partial", s, bxi::levelset_partial_kernel<true>, dim3(bxi::kLsSlices, N), dim3(bxi::kLsThreads), 0, s, mask_score, target,
                       N, C, H, W, reinterpret_cast<double*>(state), c0);
        else
            BXI_LAUNCH("levelset_partial", s, bxi::levelset_partial_kernel<false>, dim3(bxi::kLsSlices, N), dim3(bxi::kLsThreads), 0, s, mask_score, target,
                       N, C, H, W, reinterpret_cast<double*>(state), c0);
        rc = bxi::check_launch();
        if (rc != BXI_OK) return rc;
    }
    BXI_LAUNCH("levelset_finish", s, bxi::levelset_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, s, pixel_num, N, C, (double)loss_weight,
               loss, reinterpret_cast<double*>(state));
    return bxi::check_launch();
}

int bxi_levelset_loss_backward_f32(const float* mask_score, const float* target, const float* pixel_num, int N, int C, int H,
                                   int W, float loss_weight, const void* state, const float* g_loss, float* g_mask_score,
                                   float* g_target, void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0) return BXI_ERR_BAD_SHAPE;
    if (C > 4096) return BXI_ERR_UNSUPPORTED;
    if (N == 0) return BXI_OK;
    if (!mask_score || !target || !pixel_num || !state || !g_loss || !g_mask_score) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * H * W * (C > 2 ? C : 2))) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)N * H * W + 255) / 256);
    if (C <= bxi::kLsMaxC)
        BXI_LAUNCH("levelset_bwd", s, bxi::levelset_bwd_kernel, dim3(grid), dim3(256), 0, s, mask_score, target, pixel_num, N, C, H, W,
                   (double)loss_weight, reinterpret_cast<const double*>(state), g_loss, g_mask_score, g_target);
    else
        BXI_LAUNCH("levelset_bwd", s, bxi::levelset_bwd_anyc_kernel, dim3(grid), dim3(256), 0, s, mask_score, target, pixel_num, N, C, H, W,
                   (double)loss_weight, reinterpret_cast<const double*>(state), g_loss, g_mask_score, g_target);
    return bxi::check_launch();
}

int bxi_lcm_affinity_f32(const float* imgs, int N, int C, int h, int w, int dilation, float alpha, float* aff, void* stream) {
    if (N < 0 || C <= 0 || h <= 0 || w <= 0) return BXI_ERR_BAD_SHAPE;
    if (dilation < 1 || !(alpha > 0.f)) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0) return BXI_OK;
    if (!imgs || !aff) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * h * w * (C > 8 ? C : 8))) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)N * h * w + 255) / 256);
    BXI_LAUNCH("lcm_affinity", s, bxi::lcm_affinity_kernel, dim3(grid), dim3(256), 0, s, imgs, N, C, h, w, dilation, alpha, aff);
    return bxi::check_launch();
}

size_t bxi_lcm_workspace_bytes(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    return sizeof(float) * (size_t)(N > 0 ? N : 1) * h * w;
}

int bxi_lcm_refine_f32(const float* aff, const float* phi, int N, int h, int w, int dilation, int iters, int transpose,
                       float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || h <= 0 || w <= 0 || iters < 0) return BXI_ERR_BAD_SHAPE;
    if (dilation < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0) return BXI_OK;
    if (!aff || !phi || !out) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * h * w * 8)) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const size_t lds = sizeof(float) * 2 * (size_t)h * w;
    auto allow_lds = [&](const void* fn, size_t bytes) -> int {
        if (bytes <= 64 * 1024) return BXI_OK;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
        return BXI_OK;
    };
    const int64_t hp = h + 2 * (int64_t)dilation, wpad = w + 2 * (int64_t)dilation;
    const size_t lds_fwd = sizeof(float) * 2 * (size_t)(hp * wpad);
    const size_t lds_adj2 = sizeof(float) * (size_t)((h + 4 * (int64_t)dilation) * (w + 4 * (int64_t)dilation) + hp * wpad);
    const bool ref_shape = h == 96 && w == 96 && dilation == 2;           // Box2Mask's LCM (levelset_loss.py:64-66, 96 x 96 targets)
    if (!transpose) {
        if (hp * wpad <= bxi::kLcmPadSPT * bxi::kLcmPadThreads && lds_fwd <= 128 * 1024) {
            const void* fn = ref_shape ? reinterpret_cast<const void*>(bxi::lcm_refine_pad_kernel<96, 96, 2>)
                                       : reinterpret_cast<const void*>(bxi::lcm_refine_pad_kernel<0, 0, 0>);
            const int rc = allow_lds(fn, lds_fwd);
            if (rc != BXI_OK) return rc;
            if (ref_shape)
                BXI_LAUNCH("lcm_refine", s, (bxi::lcm_refine_pad_kernel<96, 96, 2>), dim3(N), dim3(bxi::kLcmPadThreads), lds_fwd, s, aff, phi, h, w, dilation, iters, out);
            else
                BXI_LAUNCH("lcm_refine", s, (bxi::lcm_refine_pad_kernel<0, 0, 0>), dim3(N), dim3(bxi::kLcmPadThreads), lds_fwd, s, aff, phi, h, w, dilation, iters, out);
            return bxi::check_launch();
        }
        if (lds <= 128 * 1024) {
            const int rc = allow_lds(reinterpret_cast<const void*>(bxi::lcm_refine_lds_kernel<false>), lds);
            if (rc != BXI_OK) return rc;
            BXI_LAUNCH("lcm_refine", s, bxi::lcm_refine_lds_kernel<false>, dim3(N), dim3(1024), lds, s, aff, phi, h, w, dilation, iters, out);
            return bxi::check_launch();
        }
    } else {
        const size_t lds_adj = sizeof(float) * ((size_t)h * w + (size_t)(h + 2 * dilation) * (w + 2 * dilation));
        if (hp * wpad <= bxi::kLcmPadSPT * bxi::kLcmPadThreads && (int64_t)h * w <= (bxi::kLcmPadSPT - 1) * bxi::kLcmPadThreads &&
            2 * (int64_t)w + 2 * (h > 2 ? h - 2 : 0) <= bxi::kLcmPadThreads && lds_adj2 <= 128 * 1024) {
            const void* fn = ref_shape ? reinterpret_cast<const void*>(bxi::lcm_adjoint_pad_kernel<96, 96, 2>)
                                       : reinterpret_cast<const void*>(bxi::lcm_adjoint_pad_kernel<0, 0, 0>);
            const int rc = allow_lds(fn, lds_adj2);
            if (rc != BXI_OK) return rc;
            if (ref_shape)
                BXI_LAUNCH("lcm_adjoint", s, (bxi::lcm_adjoint_pad_kernel<96, 96, 2>), dim3(N), dim3(bxi::kLcmPadThreads), lds_adj2, s, aff, phi, h, w, dilation, iters, out);
            else
                BXI_LAUNCH("lcm_adjoint", s, (bxi::lcm_adjoint_pad_kernel<0, 0, 0>), dim3(N), dim3(bxi::kLcmPadThreads), lds_adj2, s, aff, phi, h, w, dilation, iters, out);
            return bxi::check_launch();
        }
        if (lds_adj <= 128 * 1024) {
            const int rc = allow_lds(reinterpret_cast<const void*>(bxi::lcm_adjoint_lds_kernel), lds_adj);
            if (rc != BXI_OK) return rc;
            BXI_LAUNCH("lcm_adjoint", s, bxi::lcm_adjoint_lds_kernel, dim3(N), dim3(1024), lds_adj, s, aff, phi, h, w, dilation, iters, out);
            return bxi::check_launch();
        }
    }
    // large maps: ping-pong between `out` and the workspace plane, one launch per iteration
    if (!workspace || workspace_bytes < bxi_lcm_workspace_bytes(N, h, w) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return BXI_ERR_WORKSPACE;
    const int64_t total = (int64_t)N * h * w;
    if (iters == 0) {
        hipError_t e = hipMemcpyAsync(out, phi, sizeof(float) * total, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
        return BXI_OK;
    }
    float* ws = reinterpret_cast<float*>(workspace);
    const unsigned grid = (unsigned)((total + 255) / 256);
    const float* src = phi;
    for (int it = 0; it < iters; ++it) {
        float* dst = ((iters - 1 - it) % 2 == 0) ? out : ws;       // the last step writes `out`
        if (transpose) BXI_LAUNCH("lcm_step", s, bxi::lcm_refine_step_kernel<true>, dim3(grid), dim3(256), 0, s, aff, src, N, h, w, dilation, dst);
        else BXI_LAUNCH("lcm_step", s, bxi::lcm_refine_step_kernel<false>, dim3(grid), dim3(256), 0, s, aff, src, N, h, w, dilation, dst);
        const int rc = bxi::check_launch();
        if (rc != BXI_OK) return rc;
        src = dst;
    }
    return BXI_OK;
}

}  // extern "C"
