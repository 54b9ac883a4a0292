// dynamic_head_device.hpp -- device pieces of the dynamic mask head (condinst_head.py:1139-1164) shared by its own kernels
// (dynamic_head.hip) and by the evaluation's head-fused first launch (fused_eval.hip): parameter layout, the per-pixel MLP,
// input assembly, the aligned_bilinear source index.
#pragma once
#include "common.hpp"

namespace bxi {

constexpr int kDC = 8;            // dynamic_channels (configs/boxinst: 8)
constexpr int kYR = 8, kYC = 32;  // y tile (pixels at in_stride resolution) per workgroup: the backward (thread = pixel) and the head-fused launch
constexpr int kFwdR = 14, kFwdC = 32;   // y tile of the forward kernel (factors 1, 2): with its halo 15 x 33 = 495 pixels on the 512 (thread, pixel)
                                        // slots of a workgroup; 1024 workgroups instead of 1664 at 32 x 100 x 128: 13.3 -> 12.9 us
constexpr int kHeadR = 8;               // rows of a head tile inside the evaluation's first launch (any multiple of 2 up to 14; measured with
                                        // 14: head_prep 20.1 -> 21.2 us -- that launch is not bound by the head tiles' arithmetic)
constexpr int kSlots = 8;         // instance slots per (image, tile) in the backward
constexpr int kRowPad = kYR * kYC;       // LDS row stride of the staged operand rows

struct DynArgs {
    const float* feat;        // [B,C,H,W]
    const float* params;      // [N,P]  P = (C+2)*8 + 64 + 8 + 8 + 8 + 1
    const float* coors;       // [N,2]  (x,y) of the generating location, image pixels
    const int64_t* level;     // [N]
    const int64_t* img;       // [N]
    const float* soi;         // [n_levels]
    int B, H, W, N, n_levels, in_stride, factor, rel;   // rel = !disable_rel_coors
};


// one pixel through the three dynamic layers.  `wts` = the instance's parameters in LDS (broadcast reads).
// Every bound is a compile-time constant: a runtime bound would index the register arrays dynamically,
// which the compiler can only serve with select chains (the first version of this file ran 10x slower).
template <int C, bool REL> struct Dyn {
    static constexpr int CIN = REL ? C + 2 : C;
    static constexpr int W1 = CIN * kDC, W2 = W1 + kDC * kDC, B0 = W2 + kDC, B1 = B0 + kDC, B2 = B1 + kDC, P = B2 + 1;
};

// BXI_SEGMENT: the scheduler may not move instructions across it.  The weights arrive by scalar loads; without
// the fences the scheduler hoists all 233 loads to the top, runs out of SGPRs and spills them to VGPR lanes.
#define BXI_SEGMENT() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// Packed math: a dot product runs over input PAIRS, acc2 += (w[2k], w[2k+1]) * (x[2k], x[2k+1]) as one
// v_pk_fma_f32 with the weight pair as an SGPR operand, and ends with acc2.x + acc2.y: half the VALU
// instructions of scalar FMAs (left to itself the compiler emits v_pk_mul + two v_add per pair).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f w2_at(const float* __restrict__ w, int i) { return v2f{w[i], w[i + 1]}; }

// in2/h1/h2 hold consecutive channels as pairs
template <int C, bool REL>
__device__ __forceinline__ float mlp_forward(const float* __restrict__ wts, const v2f (&in2)[Dyn<C, REL>::CIN / 2],
                                             v2f (&h1)[kDC / 2], v2f (&h2)[kDC / 2]) {
    using D = Dyn<C, REL>;
    static_assert(D::CIN % 2 == 0, "packed dot products need an even channel count");
    float t[kDC];
    // First layer: a scheduler fence per PAIR of output channels, the two dot products interleaved.  (i) What may be in flight: with a fence per four
    // channels the 4 x (CIN + 1) = 76 scalar registers of a segment -- next to the ~40 the instance loop keeps live -- exceeded the file and the compiler
    // parked freshly LOADED weights in VGPR lanes: 222 lane moves in dyn_bwd2_kernel's 1883 instructions, 56 in 1666 now (27.6 -> 25.3-25.6 us at 32
    // instances, 66.5 -> 61.2-61.9 at 128; a fence inside the second layer makes it worse again).  (ii) Consecutive v_pk_fma_f32 into ONE accumulator
    // cost a wait state each (a two-pass instruction): two chains side by side need none; each channel's sum keeps its order -- the same bits.
    // profiles/NOTES.md R6-17
#pragma unroll
    for (int o = 0; o < kDC; o += 2) {
        BXI_SEGMENT();
        v2f accA = {wts[D::B0 + o], 0.f}, accB = {wts[D::B0 + o + 1], 0.f};
#pragma unroll
        for (int i = 0; i < D::CIN / 2; ++i) {
            accA = pk_fma(w2_at(wts, o * D::CIN + 2 * i), in2[i], accA);
            accB = pk_fma(w2_at(wts, (o + 1) * D::CIN + 2 * i), in2[i], accB);
        }
        t[o] = fmaxf(accA.x + accA.y, 0.f);
        t[o + 1] = fmaxf(accB.x + accB.y, 0.f);
    }
#pragma unroll
    for (int o = 0; o < kDC / 2; ++o) h1[o] = v2f{t[2 * o], t[2 * o + 1]};
    BXI_SEGMENT();
#pragma unroll
    for (int o = 0; o < kDC; ++o) {
        v2f acc = {wts[D::B1 + o], 0.f};
#pragma unroll
        for (int i = 0; i < kDC / 2; ++i) acc = pk_fma(w2_at(wts, D::W1 + o * kDC + 2 * i), h1[i], acc);
        t[o] = fmaxf(acc.x + acc.y, 0.f);
    }
#pragma unroll
    for (int o = 0; o < kDC / 2; ++o) h2[o] = v2f{t[2 * o], t[2 * o + 1]};
    v2f y = {wts[D::B2], 0.f};
#pragma unroll
    for (int i = 0; i < kDC / 2; ++i) y = pk_fma(w2_at(wts, D::W2 + 2 * i), h2[i], y);
    BXI_SEGMENT();
    return y.x + y.y;
}

// two pixels at once: every weight is used the moment its scalar load lands, so none has to be kept
// (evaluating the pixels one after the other makes the compiler keep all 233 SGPRs and spill them).  The two pixels ride
// in the two halves of packed FMAs, acc(A,B) += (w,w) * (x_A, x_B), the weight broadcast from its SGPR: one instruction per
// weight instead of two (the forward kernel is bound by instruction issue, not by its 6.5 MB of stores).
template <int C, bool REL>
__device__ __forceinline__ void mlp_forward2(const float* __restrict__ wts, const float (&inA)[Dyn<C, REL>::CIN],
                                             const float (&inB)[Dyn<C, REL>::CIN], float& yA, float& yB) {
    using D = Dyn<C, REL>;
    v2f x0[D::CIN], x1[kDC], x2[kDC];
#pragma unroll
    for (int i = 0; i < D::CIN; ++i) x0[i] = v2f{inA[i], inB[i]};
#pragma unroll
    for (int o = 0; o < kDC; ++o) {
        const float b = wts[D::B0 + o];
        v2f acc = {b, b};
#pragma unroll
        for (int i = 0; i < D::CIN; ++i) { const float w = wts[o * D::CIN + i]; acc = pk_fma(v2f{w, w}, x0[i], acc); }
        x1[o] = v2f{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f)};
    }
#pragma unroll
    for (int o = 0; o < kDC; ++o) {
        const float b = wts[D::B1 + o];
        v2f acc = {b, b};
#pragma unroll
        for (int i = 0; i < kDC; ++i) { const float w = wts[D::W1 + o * kDC + i]; acc = pk_fma(v2f{w, w}, x1[i], acc); }
        x2[o] = v2f{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f)};
    }
    const float b2 = wts[D::B2];
    v2f y = {b2, b2};
#pragma unroll
    for (int i = 0; i < kDC; ++i) { const float w = wts[D::W2 + i]; y = pk_fma(v2f{w, w}, x2[i], y); }
    yA = y.x; yB = y.y;
}

// in[] = (relative coordinates when REL,) C feature channels of pixel (r,c) of image b
template <int C, bool REL>
__device__ __forceinline__ void load_inputs(const DynArgs& a, int n, int b, int r, int c, float (&in)[Dyn<C, REL>::CIN]) {
    const int64_t HW = (int64_t)a.H * a.W;
    const float* fb = a.feat + (int64_t)b * C * HW;          // uniform base, 32-bit lane offset
    const unsigned po = (unsigned)(r * a.W + c);
    constexpr int off = REL ? 2 : 0;
    if constexpr (REL) {
        // locations = arange(0, W*stride, stride) + stride // 2 (:1143-1150); (coors - locations) / soi (:1151-1153)
        const float soi = a.soi[a.level[n]];
        in[0] = (a.coors[2 * n] - (float)(c * a.in_stride + a.in_stride / 2)) / soi;
        in[1] = (a.coors[2 * n + 1] - (float)(r * a.in_stride + a.in_stride / 2)) / soi;
    }
#pragma unroll
    for (int k = 0; k < C; ++k) in[off + k] = (fb + (int64_t)k * HW)[po];
}

// source row/column and fraction of output index R (aligned_bilinear :146-167):
// z[R] = I[max(R - f/2, 0)], I[i] = sample of (y padded by one replicated row) at i/f
__device__ __forceinline__ void upsample_src(int R, int f, int n_in, int& i0, int& i1, float& fr) {
    const int ii = max(R - f / 2, 0);
    i0 = ii / f;
    fr = (float)(ii - i0 * f) / (float)f;
    i1 = min(i0 + 1, n_in - 1);                             // replicate pad (:156)
}

// F = the up-sampling factor as a compile-time constant (0: read a.factor; every index division is then a
// real integer division, ~40 instructions each)
template <int F> __device__ __forceinline__ int factor_of(const DynArgs& a) { return F ? F : a.factor; }

// What the head-fused evaluation (fused_eval.hip) asks of a tile besides its logits: the zero-filled gradient tile and the
// tile's share of the projection maxima -- per output row the (value, first column) key over the tile's 64 columns, per output
// column the (value, first row) key over its 16 rows -- as partials the leaders of pair_kernel combine.
struct DynEpi {
    unsigned long long* colpart;    // [N][n_cb][OW]   n_cb = tiles in y
    unsigned long long* rowpart;    // [N][n_rp][OH]   n_rp = tiles in x
    float* g_zero;                  // [N][OH][OW] or nullptr
    int n_cb, n_rp;
    int through;                    // 1: partials and the zero tile are written through (sc1): a workgroup of the SAME launch reads
                                    //    them after an arrival counter (eval3.hip: the instance's last tile is its leader)
};

// One workgroup: y on a TR x TC tile (+ halo) of instance n, up-sampled to the logits tile.  Thread t evaluates pixels 2t and
// 2t+1 of the halo tile (each weight feeds both), y goes to LDS, the up-sampled tile is written with float2 / float4 stores.
// Halo: one row above and one column to the left; below / to the right only when the factor needs it -- for factors 1 and 2 the last
// output row of a tile samples its last y row with weight 1 (upsample_src: fraction 0), so the 8 x 32 tile stages 9 x 33 pixels, not
// 10 x 34, and a 14 x 32 tile's 495 fill the 512 (thread, pixel) slots of a workgroup (8 x 32: 340 of 512).
template <int F> struct DynHalo { static constexpr int after = (F == 1 || F == 2) ? 0 : 1; };
template <int C, bool REL, int F, bool EPI, int TR = kYR, int TC = kYC>
__device__ __forceinline__ void dyn_tile_forward(const DynArgs& a, const float* __restrict__ params, float* __restrict__ logits, int n,
                                                 int ty, int tx, float* ytile /* LDS [(TR+1+after)*(TC+1+after)] */,
                                                 float* otile /* LDS [TR*F][TC*F], EPI */,
                                                 unsigned long long* ckeys /* LDS [4][TC*F], EPI */, const DynEpi& ep) {
    using D = Dyn<C, REL>;
    constexpr int HB = DynHalo<F>::after, PW = TC + 1 + HB, PH = TR + 1 + HB;
    constexpr int kHalo = PH * PW;
    static_assert(kHalo <= 512, "two pixels per thread");
    constexpr int kYR = TR, kYC = TC;                       // (shadow the namespace constants below)
    const int tid = threadIdx.x;
    // the instance's 233 parameters are wave-uniform: scalar loads, SGPR operands of the FMAs (no LDS, no VGPRs)
    const float* __restrict__ wts = params + (int64_t)n * D::P;
    const int b = (int)a.img[n];
    // y on the tile plus its halo (rows r0-1 .. r0+kYR-1+HB): up to 512 pixels on 256 threads.
    // Both pixels of a thread are loaded (clamped coordinates, no branch) before either is evaluated.
    const int r0 = ty * kYR, c0 = tx * kYC;
    const int eA = 2 * tid, eB = 2 * tid + 1;
    const int rA = r0 - 1 + eA / PW, cA = c0 - 1 + eA % PW;
    const int rB = r0 - 1 + eB / PW, cB = c0 - 1 + eB % PW;
    const bool vA = eA < kHalo && rA >= 0 && rA < a.H && cA >= 0 && cA < a.W;
    const bool vB = eB < kHalo && rB >= 0 && rB < a.H && cB >= 0 && cB < a.W;
    if (eA < kHalo) {
        float inA[D::CIN], inB[D::CIN], yA, yB;
        load_inputs<C, REL>(a, n, b, min(max(rA, 0), a.H - 1), min(max(cA, 0), a.W - 1), inA);
        load_inputs<C, REL>(a, n, b, min(max(rB, 0), a.H - 1), min(max(cB, 0), a.W - 1), inB);
#ifdef BXI_TRACE
        if (!EPI) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BXI_T(3, blockIdx.x, 1); }
#endif
        mlp_forward2<C, REL>(wts, inA, inB, yA, yB);
        ytile[eA] = vA ? yA : 0.f;
        if (eB < kHalo) ytile[eB] = vB ? yB : 0.f;
    }
    if (!EPI) BXI_T(3, blockIdx.x, 2);
    __syncthreads();
    if (!EPI) BXI_T(3, blockIdx.x, 3);
    const int f = factor_of<F>(a), OH = a.H * f, OW = a.W * f;
    constexpr int VW = F == 0 ? 1 : (F % 4 == 0 ? 4 : (F % 2 == 0 ? 2 : 1));   // outputs per store
    const int R0 = r0 * f, C0 = c0 * f;
    const int row_w = kYC * f / VW;                          // stores per output row of the tile
    float* out = logits + (int64_t)n * OH * OW;
    auto Y = [&](int r, int c) { return ytile[(r - r0 + 1) * PW + (c - c0 + 1)]; };
    for (int i = tid; i < kYR * f * row_w; i += 256) {
        const int R = R0 + i / row_w, Cc = C0 + (i % row_w) * VW;
        if (R >= OH || Cc >= OW) continue;
        int y0, y1; float fy;
        upsample_src(R, f, a.H, y0, y1, fy);
        if (HB == 0) y1 = min(y1, r0 + kYR - 1);             // past the staged rows only with weight 0: keep the operand finite
        float v[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            int x0, x1; float fx;
            upsample_src(Cc + k, f, a.W, x0, x1, fx);
            if (HB == 0) x1 = min(x1, c0 + kYC - 1);
            const float top = (1.f - fx) * Y(y0, x0) + fx * Y(y0, x1);
            const float bot = (1.f - fx) * Y(y1, x0) + fx * Y(y1, x1);
            v[k] = (1.f - fy) * top + fy * bot;
        }
        float* o = out + (int64_t)R * OW + Cc;
        if constexpr (VW == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else if constexpr (VW == 2) *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
        else o[0] = v[0];
        if constexpr (EPI) {
#pragma unroll
            for (int k = 0; k < VW; ++k) otile[(R - R0) * (kYC * F) + (Cc - C0) + k] = v[k];
        }
    }
    if constexpr (EPI) {
        static_assert(!EPI || (F == 2 && kYC * F == 64 && (kYR * F) % 4 == 0), "the epilogue maps a lane to a column of a 64-wide tile, a wave to a quarter of its rows");
        constexpr int TW = kYC * F, TH = kYR * F;
        __syncthreads();
        const int wv = tid >> 6, lane = tid & 63, c = C0 + lane;
        unsigned long long ck = 0ull;
#pragma unroll
        for (int i = 0; i < TH / 4; ++i) {
            const int lr = wv * (TH / 4) + i, R = R0 + lr;
            const bool ok = R < OH && c < OW;
            const float v = ok ? otile[lr * TW + lane] : -INFINITY;
            const unsigned long long rk = wave_max_u64(ok ? pack_max(v, (uint32_t)c) : 0ull);      // larger value, then smaller column
            if (lane == 0 && R < OH) {
                unsigned long long* dst = ep.rowpart + ((int64_t)n * ep.n_rp + tx) * OH + R;
                if (ep.through) __hip_atomic_store(dst, rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = rk;
            }
            const unsigned long long k = ok ? pack_max(v, (uint32_t)R) : 0ull;                      // larger value, then smaller row
            ck = k > ck ? k : ck;
        }
        ckeys[wv * TW + lane] = ck;
        if (ep.g_zero) {                                                                            // the tile of d loss / d logits
            for (int z = tid; z < TH * (TW / 4); z += 256) {
                const int zr = R0 + z / (TW / 4), zc = C0 + (z % (TW / 4)) * 4;
                if (zr < OH && zc < OW) {
                    float* dst = ep.g_zero + ((int64_t)n * OH + zr) * OW + zc;
                    if (ep.through) store4_through(dst, 0.f, 0.f, 0.f, 0.f); else *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        if (wv == 0 && c < OW) {
            unsigned long long k = ckeys[lane];
#pragma unroll
            for (int u = 1; u < 4; ++u) { const unsigned long long o = ckeys[u * TW + lane]; k = o > k ? o : k; }
            unsigned long long* dst = ep.colpart + ((int64_t)n * ep.n_cb + ty) * OW + c;
            if (ep.through) __hip_atomic_store(dst, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = k;
        }
    }
}

// host side (dynamic_head.hip): argument checks shared with the head-fused evaluation
int fill_dyn(const float* feat, int B, int C, int H, int W, const float* params, int N, const float* coors, const int64_t* level,
             const int64_t* img, const float* soi, int n_levels, int in_stride, int factor, int disable_rel, DynArgs& a);

}  // namespace bxi
