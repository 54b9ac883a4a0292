// loss_common.hpp -- types and device helpers shared by the two builds of the BoxInst loss path:
//   mask_loss.hip   affinity bits given (bxi_boxinst_loss_fwd_bwd_f32) + the dilation > 4 fallback: stage1 / box / loss_apply
//   fused_eval.hip  the evaluation proper (bxi_boxinst_eval_f32): prep / pair, finished gradient in two launches
// Reference semantics: CondInstMaskHead.loss, condinst_head.py:1288-1343 (see the kernels for line-by-line citations).
#pragma once
#include "image_device.hpp"

namespace bxi {

constexpr int kMaxDil = 8;
constexpr int kMaxT = 32;       // per-instance column partials reduced per unrolled batch by the leader workgroups

struct InstArgs {
    const float* logits;
    const int64_t* gt_inds;
    int N, h, w;
    int Hc, Wc, stride;
    GtTable gt;
};

struct InstRec { int r0, r1, c0, c1, img, pad0, pad1, pad2; };   // 32 B: one load per workgroup

// (sim >= thresh) for a valid neighbour, as a compare on the squared Lab distance:
// exp(-0.5*sqrt(n2)) >= thresh  <=>  n2 <= n2max, with n2max found in stage1 by bisecting the exact
// f32 expression of the reference over the float bit patterns (the expression is monotone in n2).
struct Pred { float n2max; int fast; int zero_bit; int pad; };
constexpr float kNumScale = 16777216.f;   // 2^24

struct LossState {            // what the backward / rescale entry points need (forward -> backward)
    int* colarg;              // [N,w] arg-max row of column c
    int* rowarg;              // [N,h] arg-max column of row r
    float* gcol;              // [N,w] unit d loss_prj / d logit at (colarg[c], c)
    float* grow;              // [N,h] unit d loss_prj / d logit at (r, rowarg[r])
    InstRec* inst;            // [N]   box rectangles
    float* scale;             // [1]   warmup / max(sum W, 1)
    float* applied;           // [2]   upstream factors (g_prj, g_pw) folded into a finished gradient (fused_eval.hip)
    unsigned long long* colk; // [N,w] fused_eval.hip: (unit gradient bits << 32) | arg-max row ; low word 0xffffffff = not published yet
    unsigned long long* rowk; // [N,h] same for the rows
    int* status;              // [2]   {0 or a bit mask of protocol time-outs (never expected), tile rows R} (fused_eval.hip)
    float* iter;              // bxi_instances.iter_counter (not part of `state`): + 1.0f by the evaluation's finisher
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline size_t carve_state(void* base, int N, int h, int w, LossState* st) {
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    int* colarg = (int*)take(sizeof(int) * (size_t)N * w);
    int* rowarg = (int*)take(sizeof(int) * (size_t)N * h);
    float* gcol = (float*)take(sizeof(float) * (size_t)N * w);
    float* grow = (float*)take(sizeof(float) * (size_t)N * h);
    InstRec* inst = (InstRec*)take(32 * (size_t)(N > 0 ? N : 1));
    float* scale = (float*)take(2 * sizeof(float));     // [0] warmup / max(sum W, 1); [1] the warm-up factor itself (fused_eval.hip)
    float* applied = (float*)take(2 * sizeof(float));
    unsigned long long* colk = (unsigned long long*)take(8 * (size_t)N * w);
    unsigned long long* rowk = (unsigned long long*)take(8 * (size_t)N * h);
    int* status = (int*)take(2 * sizeof(int));
    if (st) { st->colarg = colarg; st->rowarg = rowarg; st->gcol = gcol; st->grow = grow; st->inst = inst; st->scale = scale;
              st->applied = applied; st->colk = colk; st->rowk = rowk; st->status = status; }
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

__device__ __forceinline__ InstBox inst_from_rec(const InstRec& rc, int dil, int h, int w) {
    InstBox ib;
    ib.box.r0 = rc.r0; ib.box.r1 = rc.r1; ib.box.c0 = rc.c0; ib.box.c1 = rc.c1;
    ib.img = rc.img;
    ib.any = rc.r1 > rc.r0 && rc.c1 > rc.c0;
    ib.dil = ib.box;
    if (ib.any) {
        ib.dil.r0 = max(rc.r0 - dil, 0); ib.dil.r1 = min(rc.r1 + dil, h);
        ib.dil.c0 = max(rc.c0 - dil, 0); ib.dil.c1 = min(rc.c1 + dil, w);
    }
    return ib;
}

// (p, q) = (sigmoid(x), sigmoid(-x)), both accurate relatively (no 1-p cancellation)
__device__ __forceinline__ float2 sig_pair(float x) {
    const float e = __expf(-fabsf(x));
    const float r = __builtin_amdgcn_rcpf(1.f + e);   // v_rcp_f32 (1 ulp); __frcp_rn would expand to a full IEEE division
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

// exact f32 predicate of the reference for a valid neighbour: exp(-||dLab|| * 0.5) >= thresh  (:237, :1324)
__device__ __forceinline__ bool sim_pred(float n2, float thresh) {
    return expf(__fmul_rn(-__fsqrt_rn(n2), 0.5f)) >= thresh;
}

__device__ __forceinline__ Pred make_pred(float thresh) {   // uniform: every lane computes the same value
    Pred p; p.pad = 0; p.fast = 1;
    p.zero_bit = (0.f >= thresh) ? 1 : 0;            // weight of a padded / masked-out neighbour (sim == 0)
    // sim_pred(n2) is non-increasing in n2 >= 0 and positive floats order like their bit patterns:
    // bisect the bit pattern for the largest n2 that still passes.
    if (!sim_pred(0.f, thresh)) p.n2max = -1.f;                         // thresh > 1: never
    else if (sim_pred(3.0e38f, thresh)) p.n2max = INFINITY;             // thresh <= 0 (exp underflows to 0): always
    else {
        uint32_t lo = 0u, hi = __float_as_uint(3.0e38f);                // pred(lo) true, pred(hi) false
        const float dstar = -2.f * logf(thresh);                        // analytic boundary: n2 = (2 ln thresh)^2
        const uint32_t cb = __float_as_uint(dstar * dstar);
        if (cb > 256u && cb < __float_as_uint(3.0e38f) - 256u && sim_pred(__uint_as_float(cb - 128u), thresh) &&
            !sim_pred(__uint_as_float(cb + 128u), thresh)) { lo = cb - 128u; hi = cb + 128u; }   // 8 steps instead of 31
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sim_pred(__uint_as_float(mid), thresh)) lo = mid; else hi = mid;
        }
        p.n2max = __uint_as_float(lo);
    }
    return p;
}

int fill_gt_table(const float* const* boxes_per_img_host, const int* gt_count_host, int B, GtTable& gt, int& G);
int fill_inst(const bxi_instances* in, InstArgs& a);
int fill_pool_args(const bxi_image_batch* bt, uint8_t* rgb_small, float* lab, PoolArgs& pa);
int fill_image_meta(const bxi_image_batch* bt, ImageMeta& meta, Denorm& dn);
bool pool_vec_ok(const bxi_image_batch* bt, int stride);
int launch_pool(const bxi_image_batch* bt, int stride, uint8_t* rgb_small, float* lab, hipStream_t s);

}  // namespace bxi
