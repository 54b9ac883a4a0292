// fused_eval.hip -- the BoxInst loss evaluation (forward AND finished backward) in TWO launches on gfx950.
//
// Replaces (reference, LiWentomng/BoxInstSeg) CondInstMaskHead.loss with boxinst_enabled,
// condinst_head.py:1288-1343, together with everything it calls:
//   get_targets / get_original_image / get_bitmasks_from_boxes   :170-186, :1345-1448   (image side)
//   get_image_color_similarity + unfold_wo_center                :190-246
//   compute_project_term + dice_coefficient                      :117-143
//   pairwise_nlog (CUDA op, pairwise.cu:68-149) + weights / normalise / warm-up   :1315-1332
// and what autograd does behind them, with the upstream factors folded in, so that g_logits leaves the
// second launch FINISHED (no third launch; round 1 needed loss_apply for that).
//
//   prep_kernel   256-thread workgroups, three roles                                   HBM stream
//     table waves   per-instance box rectangle, the compacted list of box tiles (+ colour predicate), zeroed counters
//     stream blocks 4 waves x 8 rows of one instance map: zero-fill of g_logits, row maxima (complete), column
//                   maxima of the block's 32 rows (combined through LDS) -> Sn = ceil(h/32) partials per column
//     pool blocks   4 waves = the 4 input rows of 64 pooled pixels: a lane loads 3 x float4 (1 KiB contiguous per
//                   wave-load), de-normalises 12 values; the 4x4 sums meet in LDS, then three waves take one CIE
//                   channel each (fp64) -> Lab f32.  Four times more, four times lighter waves than one lane per
//                   pooled pixel: their arithmetic overlaps the other waves' loads (tools/micro/dispatch.hip, D).
//   pair_kernel   256-thread workgroups, three roles + a one-wave finisher, dispatched in this order
//     leaders       one per instance: column partials -> maxima -> sigmoid -> dice -> unit projection gradients
//                   (published as self-flagging 8-byte granules, like its dice loss), projection gradient at the arg-max
//                   positions outside the tiles
//     count waves   one per box tile: sum of the pair weights W from Lab alone -> one packed atomic per tile
//     math waves    one per box tile (wave64, no LDS, no barrier): the tile + halo lives in registers, a lane owns a
//                   column; every unordered pair is evaluated ONCE and feeds both of its pixels (neighbour columns by
//                   cross-lane moves); then reads sum W (complete: the count waves precede the math waves in the
//                   grid, so nothing waits for a workgroup that may not have been dispatched), the leader's
//                   coefficients, and stores  g = g_pw * warm/max(sum W,1) * d pw + g_prj * d prj.
//     finisher      last block: one round of polls delivers both the completion check and the data of the two loss values
// Data layout in HBM: everything NCHW / row-major as the reference hands it over; Lab [B,3,h,w] f32 is the only
// materialised intermediate (1.2 MB at 2x800x1024).
#include "loss_common.hpp"
#include "dynamic_head_device.hpp"
#include <atomic>
#include <cstdlib>

namespace bxi {

constexpr int kWaves = 4;                       // waves per workgroup in both launches
constexpr int kSRows = 8;                       // rows per stream wave
constexpr int kSBlk = kWaves * kSRows;          // rows per stream workgroup
constexpr int kChunkC = 256;                    // columns per pass of a stream wave: 64 lanes x float4
constexpr int kMaxDilFused = 4;
constexpr unsigned kSpinLimit = 200000;
constexpr unsigned long long kGranuleInvalid = 0xffffffffull;
// A math wave's arrival + its share of the loss sum is one atomic on its instance's word.  ~20 tile waves per instance arrive
// within a microsecond; atomics on one word are performed one after the other (~0.15 us each): the last arrival became
// visible ~3 us after it was issued, and the launch ends on it.  Eight words per instance, each in its own 128 bytes.
constexpr int kAcc2Split = 8, kAcc2Stride = 16;
// The count waves' sums likewise: only their total over all instances is ever needed (the normaliser is global, :1327-1328),
// so they go to 64 words, one per lane of whoever adds them up; "every tile counted" = the arrivals add up to the list length.
constexpr int kAcc1Words = 64;
__device__ __forceinline__ unsigned long long* acc2_word(unsigned long long* acc2, int n, int sub) {
    return acc2 + ((size_t)n * kAcc2Split + (sub & (kAcc2Split - 1))) * kAcc2Stride;
}         // bounded waits (never reached: see the grid order above)

// developer tracing (-DBXI_TRACE builds only): per-WAVE phase stamps, see tools/trace_eval.py
#ifdef BXI_TRACE
#define BXI_TW(kid, idx, ph)                                                                                  \
    do {                                                                                                      \
        if ((threadIdx.x & 63) == 0 && g_trace && (idx) >= 0 && (idx) < ::bxi::kTraceBlocks)                  \
            g_trace[((size_t)(kid) * ::bxi::kTraceBlocks + (idx)) * ::bxi::kTracePhases + (ph)] = wall_clock64(); \
    } while (0)
#else
#define BXI_TW(kid, idx, ph) do {} while (0)
#endif

#define BXI_RLX __ATOMIC_RELAXED
#define BXI_AGENT __HIP_MEMORY_SCOPE_AGENT

struct WorkRec2 {                               // 64 B: all a tile wave needs, written by the table waves
    int r0, r1, c0, c1;                         // cells whose sample lies in the GT box (bitmask == 1)
    int img, n, tile_r0, tile_c0;
    float n2max; int zero_bit, vr, vc;          // colour predicate; valid(q) <=> y(q) < vr && x(q) < vc (:1354-1369,:1405)
    int hc1, pad0, pad1, pad2;                  // end column of the instance's tile hull (= dilated box)
};

struct EvalWs {                                 // carved from the caller's workspace
    unsigned long long* colpart;                // [N,n_cb,w] packed (max logit, first row) of a band of rows (32: stream blocks; 16: head tiles)
    unsigned long long* rowkey;                 // [N,n_rp,h] packed (max logit, first column); n_rp = 1 (stream blocks: whole rows) or the head's tiles in x
    int n_cb, n_rp;                             // set per launch
    InstRec* inst;                              // [N]
    WorkRec2* work;                             // [cap]
    int* nwork;                                 // [1]
    unsigned int* expect;                       // [N]  tiles of the instance
    // words polled inside pair_kernel; zeroed by prep_kernel's table waves (i.e. before a kernel boundary)
    unsigned long long* acc1;                   // [kAcc1Words] (one per 128 B)  count waves : arrivals << 40 | sum W; all words together: every tile, the whole sum
    unsigned long long* acc2;                   // [N][kAcc2Split] (one per 128 B)  math waves : arrivals << 52 | sum (W pw + 1) in 2^-24 units
    unsigned long long* dice;                   // [N] leader : 1 << 32 | bits of the instance's dice loss (0 = not published; zeroed by the table waves)
};

static inline int tile_width(int dil) { return 64 - 2 * dil; }
static inline int eval_cap(int N, int h, int w, int dil, int R) {
    const int tw = tile_width(dil);
    return (N > 0 ? N : 1) * ((h + R - 1) / R) * ((w + tw - 1) / tw);
}

static size_t carve_eval(void* base, int N, int h, int w, EvalWs* ws) {
    const int N1 = N > 0 ? N : 1;
    const size_t Sn = (size_t)(h + kSBlk - 1) / kSBlk;
    const size_t cb_max = (size_t)(h + kYR * 2 - 1) / (kYR * 2), rp_max = (size_t)(w + kYC * 2 - 1) / (kYC * 2);   // the head-fused launch's tiles
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    EvalWs t;
    t.colpart = (unsigned long long*)take(8 * (size_t)N1 * (cb_max > Sn ? cb_max : Sn) * w);
    t.rowkey = (unsigned long long*)take(8 * (size_t)N1 * h * (rp_max > 1 ? rp_max : 1));
    t.n_cb = (int)Sn; t.n_rp = 1;
    t.inst = (InstRec*)take(sizeof(InstRec) * (size_t)N1);
    t.work = (WorkRec2*)take(sizeof(WorkRec2) * (size_t)eval_cap(N, h, w, 1, 4));   // the largest list any (dil, R) produces
    t.nwork = (int*)take(sizeof(int));
    t.expect = (unsigned int*)take(4 * (size_t)N1);
    t.acc1 = (unsigned long long*)take(8 * (size_t)kAcc1Words * kAcc2Stride);
    t.acc2 = (unsigned long long*)take(8 * (size_t)N1 * kAcc2Split * kAcc2Stride);
    t.dice = (unsigned long long*)take(8 * (size_t)N1);
    if (ws) *ws = t;
    return off;
}

// ================================================================================================
// prep_kernel
// ================================================================================================
// ---- role 1: table waves (one wave per instance) -------------------------------------------------------------
struct LaneBox2 { int r0, r1, c0, c1, img, tr0, ntr, hc0, hc1, ntc; };
__device__ __forceinline__ LaneBox2 lane_box2(const InstArgs& a, int dil, int R, int m) {
    LaneBox2 lb = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t g = a.gt_inds[m];
    const float* bp = nullptr;
    for (int b = 0; b < a.gt.B; ++b)      // uniform loop: the by-value kernel argument is never indexed per lane
        if (g >= a.gt.first[b] && g < a.gt.first[b + 1]) { bp = a.gt.boxes[b] + 4 * (g - a.gt.first[b]); lb.img = b; }
    if (!bp) return lb;
    const Rect rc = box_rect(bp, a.Hc, a.Wc, a.stride, a.stride / 2, a.h, a.w);
    if (rc.r1 <= rc.r0 || rc.c1 <= rc.c0) return lb;
    lb.r0 = rc.r0; lb.r1 = rc.r1; lb.c0 = rc.c0; lb.c1 = rc.c1;
    const int r0 = max(rc.r0 - dil, 0), r1 = min(rc.r1 + dil, a.h);
    lb.hc0 = max(rc.c0 - dil, 0); lb.hc1 = min(rc.c1 + dil, a.w);
    lb.tr0 = r0 / R; lb.ntr = (r1 - 1) / R - r0 / R + 1;
    const int tw = 64 - 2 * dil;
    lb.ntc = (lb.hc1 - lb.hc0 + tw - 1) / tw;
    return lb;
}

__device__ __forceinline__ void table_wave(const InstArgs& a, const ImageMeta& meta, int dil, int R, float thresh,
                                           const EvalWs& ws, const LossState& st, int n) {
    const int lane = threadIdx.x & 63;
    int base = 0, total = 0;
    LaneBox2 mine = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int m0 = 0; m0 < a.N; m0 += 64) {           // exclusive scan of the tile counts: deterministic offsets, no atomics
        const int m = m0 + lane;
        LaneBox2 lb = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (m < a.N) lb = lane_box2(a, dil, R, m);
        const int cm = lb.ntr * lb.ntc;
        int incl = cm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (n >= m0 && n < m0 + 64) {
            const int src = n - m0;
            base = total + __shfl(incl - cm, src, 64);
            mine.r0 = __shfl(lb.r0, src, 64); mine.r1 = __shfl(lb.r1, src, 64);
            mine.c0 = __shfl(lb.c0, src, 64); mine.c1 = __shfl(lb.c1, src, 64); mine.img = __shfl(lb.img, src, 64);
            mine.tr0 = __shfl(lb.tr0, src, 64); mine.ntr = __shfl(lb.ntr, src, 64);
            mine.hc0 = __shfl(lb.hc0, src, 64); mine.hc1 = __shfl(lb.hc1, src, 64); mine.ntc = __shfl(lb.ntc, src, 64);
        }
        total += __shfl(incl, 63, 64);
    }
    const int cnt = mine.ntr * mine.ntc;
    const Pred pr = make_pred(thresh);
    const int img = __builtin_amdgcn_readfirstlane(mine.img);
    const int vr = min(meta.img_h[img], meta.first_removed[img]), vc = meta.img_w[img];
    if (lane == 0) {
        if (n == 0) { *ws.nwork = total; if (st.status) { st.status[0] = 0; st.status[1] = R; } }
        InstRec rc; rc.r0 = mine.r0; rc.r1 = mine.r1; rc.c0 = mine.c0; rc.c1 = mine.c1; rc.img = mine.img;
        rc.pad0 = rc.pad1 = rc.pad2 = 0;
        ws.inst[n] = rc;
        if (st.inst) st.inst[n] = rc;
        ws.expect[n] = (unsigned int)cnt;
    }
    if (n == 0) ws.acc1[lane * kAcc2Stride] = 0ull;
    if (lane < kAcc2Split) *acc2_word(ws.acc2, n, lane) = 0ull;
    if (lane == 0) ws.dice[n] = 0ull;
    if (st.colk) {      // "not published yet" (the leaders of the next launch publish; its math waves poll)
        for (int i = lane; i < a.w; i += 64) st.colk[(int64_t)n * a.w + i] = kGranuleInvalid;
        for (int i = lane; i < a.h; i += 64) st.rowk[(int64_t)n * a.h + i] = kGranuleInvalid;
    }
    for (int i = lane; i < cnt; i += 64) {
        WorkRec2 wr;
        wr.r0 = mine.r0; wr.r1 = mine.r1; wr.c0 = mine.c0; wr.c1 = mine.c1; wr.img = mine.img; wr.n = n;
        wr.tile_r0 = (mine.tr0 + i / mine.ntc) * R; wr.tile_c0 = mine.hc0 + (i % mine.ntc) * (64 - 2 * dil);
        wr.n2max = pr.n2max; wr.zero_bit = pr.zero_bit; wr.vr = vr; wr.vc = vc;
        wr.hc1 = mine.hc1; wr.pad0 = wr.pad1 = wr.pad2 = 0;
        ws.work[base + i] = wr;
    }
}

// ---- role 2: stream block = 4 waves x 8 rows of one instance map -----------------------------------------------
//   - zero-fill of d loss / d logits first (depends on nothing; pair_kernel overwrites the box tiles);
//   - all 8 row loads in flight together; per-row max / first arg-max by 8 interleaved butterflies;
//   - per-column max / first arg-max over the wave's rows in registers, over the block's 4 waves through LDS.
// `src(r, c)` hands over logits (r, c .. c + 3) of instance n: loaded (LogitRows) or produced on the spot by the dynamic mask
// head (HeadRows, head-fused variant); every (r, c) is asked for exactly once.
struct LogitRows {
    const float* L; int w, vec;
    __device__ __forceinline__ float4 operator()(int r, int c) const { return load4(L + (int64_t)r * w, c, w, vec); }
};

template <typename Src>
__device__ __forceinline__ void stream_block(const InstArgs& a, const EvalWs& ws, float* __restrict__ g_logits, int vec, int sb,
                                             unsigned long long* colp /* LDS [kWaves][w] */, const Src& src) {
    const int h = a.h, w = a.w;
    const int Sn = (h + kSBlk - 1) / kSBlk;
    const int n = sb / Sn, s = sb % Sn;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = s * kSBlk + wv * kSRows, r1 = min(h, r0 + kSRows);     // may be empty
    const int64_t P = (int64_t)h * w;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

    if (G)   // written through: see store4_through
        for (int cb = 0; cb < w; cb += kChunkC) {
            const int c = cb + lane * 4;
            if (c < w) {
#pragma unroll
                for (int i = 0; i < kSRows; ++i)
                    if (r0 + i < r1) {
                        if (vec) store4_through(G + (int64_t)(r0 + i) * w + c, 0.f, 0.f, 0.f, 0.f);
                        else store4(G + (int64_t)(r0 + i) * w, c, w, false, zero);
                    }
            }
        }
    float4 v[kSRows];
    {
        const int c = lane * 4;
#pragma unroll
        for (int i = 0; i < kSRows; ++i) v[i] = (r0 + i < r1 && c < w) ? src(r0 + i, c) : ninf;
    }
    BXI_TW(0, (int)blockIdx.x * kWaves + wv, 1);
    float rmax[kSRows]; int rcol[kSRows];
#pragma unroll
    for (int i = 0; i < kSRows; ++i) { rmax[i] = -INFINITY; rcol[i] = 0; }
    for (int cb = 0;;) {
        const int c = cb + lane * 4;
        if (c < w) {
            float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int crow[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < kSRows; ++i) {
                if (r0 + i < r1) {
                    float m = v[i].x; int mc = c;                       // first column wins ties
                    if (v[i].y > m) { m = v[i].y; mc = c + 1; }
                    if (v[i].z > m) { m = v[i].z; mc = c + 2; }
                    if (v[i].w > m) { m = v[i].w; mc = c + 3; }
                    if (m > rmax[i]) { rmax[i] = m; rcol[i] = mc; }     // chunks ascend: strict > keeps the first
                    if (v[i].x > cmax[0]) { cmax[0] = v[i].x; crow[0] = i; }   // ascending row, strict >: first row wins
                    if (v[i].y > cmax[1]) { cmax[1] = v[i].y; crow[1] = i; }
                    if (v[i].z > cmax[2]) { cmax[2] = v[i].z; crow[2] = i; }
                    if (v[i].w > cmax[3]) { cmax[3] = v[i].w; crow[3] = i; }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < w) colp[(size_t)wv * w + c + j] = pack_max(cmax[j], (uint32_t)(r0 + crow[j]));   // absolute row
        }
        cb += kChunkC;
        if (cb >= w) break;
        const int c2 = cb + lane * 4;
#pragma unroll
        for (int i = 0; i < kSRows; ++i) v[i] = (r0 + i < r1 && c2 < w) ? src(r0 + i, c2) : ninf;
    }
    BXI_TW(0, (int)blockIdx.x * kWaves + wv, 2);
    float wmax[kSRows];
#pragma unroll
    for (int i = 0; i < kSRows; ++i) wmax[i] = rmax[i];
    // eight maxima over the wave side by side: within rows of 16 lanes by DPP, the four rows by v_readlane (no LDS crossbar)
    wave_total_steps([&](int c) {
        float o[kSRows];
#pragma unroll
        for (int i = 0; i < kSRows; ++i) o[i] = __int_as_float(dpp_i32(__float_as_int(wmax[i]), c));
#pragma unroll
        for (int i = 0; i < kSRows; ++i) wmax[i] = fmaxf(wmax[i], o[i]);
    });
#pragma unroll
    for (int i = 0; i < kSRows; ++i) {
        const int b = __float_as_int(wmax[i]);
        wmax[i] = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                        fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
    }
    unsigned long long mine = 0ull;
#pragma unroll
    for (int i = 0; i < kSRows; ++i) {
        const unsigned long long who = __ballot(rmax[i] == wmax[i]);
        const int first = who ? __ffsll((long long)who) - 1 : 0;
        const int col = __builtin_amdgcn_readlane(rcol[i], first);
        if (lane == i) mine = pack_max(wmax[i], (uint32_t)col);
    }
    if (lane < kSRows && r0 + lane < r1) ws.rowkey[(int64_t)n * h + r0 + lane] = mine;
    BXI_TW(0, (int)blockIdx.x * kWaves + wv, 3);
    lds_barrier();
    BXI_TW(0, (int)blockIdx.x * kWaves + wv, 4);
    for (int c = threadIdx.x; c < w; c += kWaves * 64) {
        unsigned long long k = colp[c];
#pragma unroll
        for (int u = 1; u < kWaves; ++u) { const unsigned long long o = colp[(size_t)u * w + c]; k = o > k ? o : k; }
        ws.colpart[((int64_t)n * Sn + s) * w + c] = k;       // larger value, then smaller row
    }
}

// ---- role 3: pool block = the 4 input rows of 64 pooled pixels --------------------------------------------------
// Arithmetic identical to pool_finish_s4 / rgb2lab_f32 (image_device.hpp), only distributed differently.
__device__ __forceinline__ double lab_f(const double* lut, int i, int r8, int g8, int b8) {
    const double r = lut[r8], g = lut[g8], b = lut[b8];
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    const double m0 = i == 0 ? M[0][0] : (i == 1 ? M[1][0] : M[2][0]);
    const double m1 = i == 0 ? M[0][1] : (i == 1 ? M[1][1] : M[2][1]);
    const double m2 = i == 0 ? M[0][2] : (i == 1 ? M[1][2] : M[2][2]);
    const double wt = i == 0 ? white[0] : (i == 1 ? white[1] : white[2]);
    const double acc = __dadd_rn(__dadd_rn(__dmul_rn(m0, r), __dmul_rn(m1, g)), __dmul_rn(m2, b));
    const double v = acc / wt;
    return v > 0.008856 ? cbrt(v) : __dadd_rn(__dmul_rn(7.787, v), 16.0 / 116.0);
}

__device__ __forceinline__ void pool_load(const PoolArgs& pa, int item, int segs, int h, int w, float4 (&v)[3]) {
    const int seg = item % segs, r = (item / segs) % h, b = item / (segs * h);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = seg * 64 + lane;
    const int64_t plane = (int64_t)pa.Hc * pa.Wc;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < w) {
        const float* base = pa.imgs + (int64_t)b * 3 * plane + (int64_t)(4 * r + wv) * pa.Wc + 4 * c;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = *reinterpret_cast<const float4*>(base + pa.dn.src_ch[ch] * plane);
    }
}

// items first, first + step, ... < n_items
__device__ __forceinline__ void pool_block(const PoolArgs& pa, int first, int step, int n_items, double* lut /*[256]*/,
                                           int* part /*[4][3][64]*/, double* fch /*[3][64]*/) {
    const int h = pa.Hc >> 2, w = pa.Wc >> 2;
    const int segs = (w + 63) >> 6;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 v[3], nx[3];
    pool_load(pa, first, segs, h, w, v);
    lut[threadIdx.x] = kSrgbLut[threadIdx.x];            // staged while the image loads fly
    for (int item = first; item < n_items; item += step) {
        const bool more = item + step < n_items;         // workgroup-uniform
        if (more) pool_load(pa, item + step, segs, h, w, nx);
        const int seg = item % segs, r = (item / segs) % h, b = item / (segs * h);
        const int c = seg * 64 + lane;
        const int y = 4 * r + wv;
        const bool act = c < w;
        const int ih = pa.meta.img_h[b], iw = pa.meta.img_w[b];
        const int x0 = 4 * c;
        const bool yin = y < ih;
        int sum[3];
        if (__all(!act || (yin && x0 + 3 < iw))) {       // wave-uniform: the whole row segment is image, not canvas padding
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
                sum[ch] = denorm_u8(v[ch].x, s, m) + denorm_u8(v[ch].y, s, m) + denorm_u8(v[ch].z, s, m) + denorm_u8(v[ch].w, s, m);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
                int t = 0;
                t += (yin && x0 + 0 < iw) ? denorm_u8(v[ch].x, s, m) : 0;
                t += (yin && x0 + 1 < iw) ? denorm_u8(v[ch].y, s, m) : 0;
                t += (yin && x0 + 2 < iw) ? denorm_u8(v[ch].z, s, m) : 0;
                t += (yin && x0 + 3 < iw) ? denorm_u8(v[ch].w, s, m) : 0;
                sum[ch] = t;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) part[(wv * 3 + ch) * 64 + lane] = sum[ch];
        BXI_TW(0, (int)blockIdx.x * kWaves + wv, 1);
        lds_barrier();
        BXI_TW(0, (int)blockIdx.x * kWaves + wv, 2);
        if (wv < 3) {                                     // wave-uniform: wave i takes channel i of XYZ -> f_i
            int px[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                px[ch] = (part[(0 * 3 + ch) * 64 + lane] + part[(1 * 3 + ch) * 64 + lane] + part[(2 * 3 + ch) * 64 + lane] +
                          part[(3 * 3 + ch) * 64 + lane]) >> 4;
            fch[wv * 64 + lane] = lab_f(lut, wv, px[0], px[1], px[2]);
        }
        BXI_TW(0, (int)blockIdx.x * kWaves + wv, 3);
        lds_barrier();
        BXI_TW(0, (int)blockIdx.x * kWaves + wv, 4);
        if (wv < 3 && act && pa.lab) {
            const double f1 = fch[64 + lane];
            float o;
            if (wv == 0) o = (float)__dadd_rn(__dmul_rn(116.0, f1), -16.0);
            else if (wv == 1) o = (float)__dmul_rn(500.0, __dadd_rn(fch[lane], -f1));
            else o = (float)__dmul_rn(200.0, __dadd_rn(f1, -fch[128 + lane]));
            const int64_t P = (int64_t)h * w;
            pa.lab[((int64_t)b * 3 + wv) * P + (int64_t)r * w + c] = o;
        }
        // the next trip's `part` / `fch` writes come after barriers every wave has to reach: no extra barrier needed
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = nx[ch];
    }
}

// ---- head-fused first launch ----------------------------------------------------------------------------------------
// CondInstMaskHead.forward (condinst_head.py:1139-1164) and the evaluation's first launch as ONE grid of independent roles:
//   [table blocks][pool blocks][head tiles: instance x 8 x 32 tiles of y -> 16 x 64 logits]
// The head tiles are dyn_fwd_kernel's workgroups (dynamic_head_device.hpp) with an epilogue that does the stream role's job on
// the tile they just produced: zero-fill of the gradient tile, per-row and per-column (value, first index) maxima as partials
// for the leaders.  Nothing in the launch waits for anything else in it: the HBM-bound image pooling and the issue-bound MLP
// simply run side by side, one launch, one boundary and one 6.5 MB read of the logits fewer than dyn_fwd + prep: 20.0 us against
// 13.4 + 11.4 us (rocprofv3, tools/bench_head_fused.py).  (Pool blocks dealt evenly among the head tiles instead of first: 21.1 us.
// The first attempt put the MLP into the stream blocks -- 224 workgroups, 4.5 dependent rounds a wave: 30.9 us.)
template <int C, bool REL>
__global__ __launch_bounds__(256, 7) void head_prep_kernel(PoolArgs pa, int n_pool, int n_items, InstArgs a, int dil, int R, float thresh,
                                                           EvalWs ws, LossState st, float* __restrict__ g_logits, DynArgs da,
                                                           const float* __restrict__ params, float* __restrict__ logits_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_tab = (a.N + kWaves - 1) / kWaves;
    const int blk = (int)blockIdx.x;
    if (blk < n_tab) {
        const int n = blk * kWaves + (int)(threadIdx.x >> 6);
        if (n < a.N) table_wave(a, pa.meta, dil, R, thresh, ws, st, n);
    } else if (blk < n_tab + n_pool) {
        double* lut = reinterpret_cast<double*>(smem);
        double* fch = lut + 256;
        int* part = reinterpret_cast<int*>(fch + 3 * 64);
        pool_block(pa, blk - n_tab, n_pool, n_items, lut, part, fch);
    } else {
        const int tiles_x = (da.W + kYC - 1) / kYC, tiles_y = (da.H + kYR - 1) / kYR;
        int t = blk - n_tab - n_pool;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        unsigned long long* ckeys = reinterpret_cast<unsigned long long*>(smem);          // [4][64]
        float* otile = reinterpret_cast<float*>(ckeys + 4 * 64);                          // [16][64]
        float* ytile = otile + 16 * 64;                                                   // [(kYR+2)*(kYC+2)]
        const DynEpi ep = {ws.colpart, ws.rowkey, g_logits, ws.n_cb, ws.n_rp, 0};
        dyn_tile_forward<C, REL, 2, true>(da, params, logits_out, n, ty, tx, ytile, otile, ckeys, ep);
    }
}

// grid: [ceil(N/4) table blocks][N*Sn stream blocks][pool blocks].  The table waves carry dependent scalar chains, so
// they go first; the stream blocks precede the pool blocks because their data feeds the next launch's first workgroups.
__global__ __launch_bounds__(256, 5) void prep_kernel(PoolArgs pa, int n_pool, int n_items, InstArgs a, int dil, int R, float thresh, EvalWs ws,
                                                   LossState st, float* __restrict__ g_logits, int vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_tab = (a.N + kWaves - 1) / kWaves;
    const int n_stream = a.N * ((a.h + kSBlk - 1) / kSBlk);
    const int blk = (int)blockIdx.x;
    const int tix = blk * kWaves + (int)(threadIdx.x >> 6);
    (void)tix;
    BXI_TW(0, tix, 0);
    if (blk < n_tab) {
        const int n = blk * kWaves + (int)(threadIdx.x >> 6);
        if (n < a.N) table_wave(a, pa.meta, dil, R, thresh, ws, st, n);
    } else if (blk < n_tab + n_stream) {
        const int Sn = (a.h + kSBlk - 1) / kSBlk;
        const LogitRows rows = {a.logits + (int64_t)((blk - n_tab) / Sn) * a.h * a.w, a.w, vec};
        stream_block(a, ws, g_logits, vec, blk - n_tab, reinterpret_cast<unsigned long long*>(smem), rows);
    } else {
        double* lut = reinterpret_cast<double*>(smem);
        double* fch = lut + 256;
        int* part = reinterpret_cast<int*>(fch + 3 * 64);
        pool_block(pa, blk - n_tab - n_stream, n_pool, n_items, lut, part, fch);
    }
    BXI_TW(0, tix, 7);
}

// ================================================================================================
// pair_kernel
// ================================================================================================
template <int D, int R> struct TG { static constexpr int RD = R + 2 * D, TW = 64 - 2 * D; };

struct TileFlags {   // bit j = data row j (map row tile_r0 - D + j) of this lane's column; R = the lane D to the right
    uint32_t ib, vd, ow, ibR, vdR, owR;       // in GT box ; valid image pixel ; owned by this tile
};

__device__ __forceinline__ uint32_t row_bits(int lo, int hi, int base, int n) {   // bits j in [0,n) with lo <= base + j < hi
    const int a = max(lo - base, 0), b = min(hi - base, n);
    if (b <= a) return 0u;
    return ((1u << b) - 1u) & ~((1u << a) - 1u);              // n <= 16
}

template <int D, int R>
__device__ __forceinline__ TileFlags tile_flags(const WorkRec2& wr, int h, int w, int stride, int lane) {
    constexpr int RD = TG<D, R>::RD;
    const int base = wr.tile_r0 - D;
    const int cl = wr.tile_c0 - D + lane;
    const int half = stride / 2;
    const int vrow = wr.vr - half <= 0 ? 0 : (wr.vr - half + stride - 1) / stride;      // r valid <=> r*stride + half < vr
    const int vcol = wr.vc - half <= 0 ? 0 : (wr.vc - half + stride - 1) / stride;
    const uint32_t rows_box = row_bits(wr.r0, wr.r1, base, RD);
    const uint32_t rows_val = row_bits(0, min(h, vrow), base, RD);
    const uint32_t rows_own = row_bits(wr.tile_r0, min(wr.tile_r0 + R, h), base, RD);
    const int cv = min(w, vcol);
    TileFlags f;
    {
        const int c = cl, ln = lane;
        f.ib = (c >= wr.c0 && c < wr.c1) ? rows_box : 0u;
        f.vd = (c >= 0 && c < cv) ? rows_val : 0u;
        f.ow = (ln >= D && ln < 64 - D && c < wr.hc1) ? rows_own : 0u;
    }
    {
        const int c = cl + D, ln = lane + D;
        const bool in = ln < 64;      // lanes without a right neighbour: every pair weight 0 (they receive some other lane's data)
        f.ibR = (in && c >= wr.c0 && c < wr.c1) ? rows_box : 0u;
        f.vdR = (in && c >= 0 && c < cv) ? rows_val : 0u;
        f.owR = (in && ln >= D && ln < 64 - D && c < wr.hc1) ? rows_own : 0u;
    }
    return f;
}

// The four pair directions of a step i (j = i + D), every one between this lane and the lane D to its right or itself, so
// that only right-neighbour values are ever fetched:
//   0: A = (i, l)  B = (i, l + D)   |   1: A = (j, l)  B = (i, l + D)   |   2: A = (i, l)  B = (j, l)   |   3: A = (i, l)  B = (j, l + D)
// masks, bit i = the pair of step i:  W[k, A] = mA, W[7 - k, B] = mB (zero_bit == 0: the fast paths are not taken otherwise),
// and the same restricted to pixels this tile owns.
struct DirMasks { uint32_t mA, mB, nA, nB; };
template <int D>
__device__ __forceinline__ void dir_masks(const TileFlags& f, DirMasks (&m)[4]) {
    m[0].mA = f.ib & f.vdR;          m[0].mB = f.ibR & f.vd;          m[0].nA = m[0].mA & f.ow;        m[0].nB = m[0].mB & f.owR;
    m[1].mA = (f.ib >> D) & f.vdR;   m[1].mB = f.ibR & (f.vd >> D);   m[1].nA = m[1].mA & (f.ow >> D); m[1].nB = m[1].mB & f.owR;
    m[2].mA = f.ib & (f.vd >> D);    m[2].mB = (f.ib >> D) & f.vd;    m[2].nA = m[2].mA & f.ow;        m[2].nB = m[2].mB & (f.ow >> D);
    m[3].mA = f.ib & (f.vdR >> D);   m[3].mB = (f.ibR >> D) & f.vd;   m[3].nA = m[3].mA & f.ow;        m[3].nB = m[3].mB & (f.owR >> D);
}

template <int D, int R>
__device__ __forceinline__ void load_plane(const float* __restrict__ plane, const WorkRec2& wr, int h, int w, int lane,
                                           float (&v)[R + 2 * D]) {
    const uint32_t cc4 = (uint32_t)min(max(wr.tile_c0 - D + lane, 0), w - 1) * 4u;
    const char* pb = reinterpret_cast<const char*>(plane);                       // scalar (the record is): base + 32-bit byte offset,
#pragma unroll                                                                   // no 64-bit address arithmetic per load (one plane < 2^31 bytes)
    for (int j = 0; j < R + 2 * D; ++j) {
        const uint32_t rr = (uint32_t)min(max(wr.tile_r0 - D + j, 0), h - 1);       // clamped: pairs with a pixel outside the map weigh 0
        v[j] = *reinterpret_cast<const float*>(pb + (rr * (uint32_t)w * 4u + cc4));
    }
}

__device__ __forceinline__ uint32_t spread4(uint32_t x4) { return (x4 * 0x00204081u) & 0x01010101u; }   // bits 0..3 -> bytes 0..3

__device__ __forceinline__ float n2_of(float L0, float A0, float B0, float L1, float A1, float B1) {
    const float dL = L0 - L1, dA = A0 - A1, dB = B0 - B1;     // un-fused: the decision must equal the affinity kernel's
    return __fadd_rn(__fadd_rn(__fmul_rn(dL, dL), __fmul_rn(dA, dA)), __fmul_rn(dB, dB));
}

// Generic (slow) evaluation of one tile: ordered pairs per owned pixel straight from global memory, pair value and
// gradient in log space exactly as pairwise.cu:38-61.  Taken for thresh <= 0 (zero_bit: padded / masked-out neighbours
// weigh 1) and for tiles with saturated logits (S underflows).  Returns the lane's sum W (and sum W pw, gradients -> gout).
template <int D, int R>
__device__ __forceinline__ int slow_tile(const float* __restrict__ Lg, const float* __restrict__ lab, int64_t P, int4 box, int img,
                                      int tile_r0, int tile_c0, float n2max, int zero_bit, int vr, int vc, int hc1, int h,
                                      int w, int stride, int lane, bool want_grad, float* gout /* LDS [R + 1][64]: gradients, then sum W pw */) {
    struct { int r0, r1, c0, c1, img, tile_r0, tile_c0; float n2max; int zero_bit, vr, vc, hc1; } wr =
        {box.x, box.y, box.z, box.w, img, tile_r0, tile_c0, n2max, zero_bit, vr, vc, hc1};
    const int half = stride / 2;
    const int c = wr.tile_c0 - D + lane;
    const bool col_owned = lane >= D && lane < 64 - D && c < wr.hc1;
    int cnt = 0;
    float num = 0.f;
    const float* L0p = lab + (int64_t)wr.img * 3 * P;
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
        const int r = wr.tile_r0 + j;
        float gacc = 0.f;
        if (col_owned && r < h) {
            const bool in_p = r >= wr.r0 && r < wr.r1 && c >= wr.c0 && c < wr.c1;
            const bool val_p = r * stride + half < wr.vr && c * stride + half < wr.vc;
            const int64_t pi = (int64_t)r * w + c;
            const float lp0 = L0p[pi], lp1 = L0p[P + pi], lp2 = L0p[2 * P + pi];
            const float xa = want_grad ? Lg[pi] : 0.f;
            const float ax = logsig(xa), bx = logsig(-xa);
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                const int kk = k < 4 ? k : k + 1;
                const int r2 = r + (kk / 3 - 1) * D, c2 = c + (kk % 3 - 1) * D;
                const bool inq = r2 >= 0 && r2 < h && c2 >= 0 && c2 < w;
                uint32_t pn = 0u;
                int64_t qi = 0;
                if (inq) {
                    qi = (int64_t)r2 * w + c2;
                    pn = n2_of(lp0, lp1, lp2, L0p[qi], L0p[P + qi], L0p[2 * P + qi]) <= wr.n2max ? 1u : 0u;
                }
                const bool val_q = inq && r2 * stride + half < wr.vr && c2 * stride + half < wr.vc;
                const bool in_q = inq && r2 >= wr.r0 && r2 < wr.r1 && c2 >= wr.c0 && c2 < wr.c1;
                const uint32_t wp = in_p ? (val_q ? pn : (uint32_t)wr.zero_bit) : 0u;
                const uint32_t wq = in_q ? (val_p ? pn : (uint32_t)wr.zero_bit) : 0u;
                cnt += (int)wp;                                   // weights.sum() counts padded pairs too (:1328)
                if (want_grad && inq && (wp + wq)) {
                    const float xb = Lg[qi];
                    const float ay = logsig(xb), by = logsig(-xb);
                    const float e1 = ax + ay, e0 = bx + by;
                    const float nl2 = logsig(fabsf(e1 - e0)) - fmaxf(e1, e0);
                    num += (float)wp * nl2;
                    gacc += (float)(wp + wq) * (-(expf(ay) - expf(by)) * expf(ax + bx + nl2));
                }
            }
        }
        if (want_grad) gout[j * 64 + lane] = gacc;
    }
    if (want_grad) gout[R * 64 + lane] = num;
    return cnt;
}

// The value of lane + D / lane - D, by D wavefront rotations of one lane on the VALU's data-parallel path instead of a trip
// through the LDS crossbar (ds_bpermute).  A rotation, not a shift: every lane receives something (the last D lanes receive
// lanes 0..D-1: finite data of the same tile), so the instruction needs no "old" operand and no copy in front of it; whatever
// those lanes compute from it is weighted 0 (TileFlags) or discarded by the caller.
template <int D>
__device__ __forceinline__ float lane_plus(float v) {
    int x = __float_as_int(v);
#pragma unroll
    for (int s = 0; s < D; ++s) x = __builtin_amdgcn_mov_dpp(x, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
    return __int_as_float(x);
}
template <int D>
__device__ __forceinline__ float lane_minus(float v) {
    int x = __float_as_int(v);
#pragma unroll
    for (int s = 0; s < D; ++s) x = __builtin_amdgcn_mov_dpp(x, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
    return __int_as_float(x);
}

// ---- count wave: sum over the tile's owned pixels of W[k,p] (Lab only) -------------------------------------------
template <int D, int R>
__device__ __forceinline__ void count_tile(const InstArgs& a, const float* __restrict__ lab, const EvalWs& ws, const WorkRec2& wr, int tix) {
    constexpr int RD = TG<D, R>::RD;
    const int lane = threadIdx.x & 63;
    const int h = a.h, w = a.w;
    const int64_t P = (int64_t)h * w;
    int cnt = 0;
    if (wr.zero_bit) {
        cnt = slow_tile<D, R>(nullptr, lab, P, make_int4(wr.r0, wr.r1, wr.c0, wr.c1), wr.img, wr.tile_r0, wr.tile_c0, wr.n2max, wr.zero_bit,
                                wr.vr, wr.vc, wr.hc1, h, w, a.stride, lane, false, nullptr);
    } else {
        float L[RD], A[RD], B[RD];
        const float* lp = lab + (int64_t)wr.img * 3 * P;
        load_plane<D, R>(lp, wr, h, w, lane, L);
        load_plane<D, R>(lp + P, wr, h, w, lane, A);
        load_plane<D, R>(lp + 2 * P, wr, h, w, lane, B);
        const TileFlags f = tile_flags<D, R>(wr, h, w, a.stride, lane);
        DirMasks m[4];
        dir_masks<D>(f, m);
        BXI_TW(2, tix, 1);
        float LR[RD], AR[RD], BR[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) { LR[i] = lane_plus<D>(L[i]); AR[i] = lane_plus<D>(A[i]); BR[i] = lane_plus<D>(B[i]); }
        uint32_t pb[4] = {0u, 0u, 0u, 0u};        // bit i = the colour predicate of the pair of step i
#pragma unroll
        for (int i = 0; i < R + D; ++i) {
            const int j = i + D;
            if (i == 1) BXI_TW(2, tix, 2);
            if (i >= D) pb[0] |= n2_of(L[i], A[i], B[i], LR[i], AR[i], BR[i]) <= wr.n2max ? 1u << i : 0u;
            pb[1] |= n2_of(L[j], A[j], B[j], LR[i], AR[i], BR[i]) <= wr.n2max ? 1u << i : 0u;
            pb[2] |= n2_of(L[i], A[i], B[i], L[j], A[j], B[j]) <= wr.n2max ? 1u << i : 0u;
            pb[3] |= n2_of(L[i], A[i], B[i], LR[j], AR[j], BR[j]) <= wr.n2max ? 1u << i : 0u;
        }
#pragma unroll
        for (int dir = 0; dir < 4; ++dir) cnt += __popc(pb[dir] & m[dir].nA) + __popc(pb[dir] & m[dir].nB);
    }
    cnt = wave_total_i32(cnt);
    BXI_TW(2, tix, 3);
    if (lane == 0)   // one packed atomic per tile: (arrival, sum W); integer adds commute -> run-to-run identical
        __hip_atomic_fetch_add(&ws.acc1[((wr.n * 7 + wr.tile_r0 / R + wr.tile_c0) & (kAcc1Words - 1)) * kAcc2Stride],
                               (1ull << 40) | (unsigned long long)(unsigned int)cnt, BXI_RLX, BXI_AGENT);
}

// One round over the count words: true when every tile of the list has been counted; then *total = sum W over all instances.
__device__ __forceinline__ bool counts_complete(const EvalWs& ws, int nwork, double* total) {
    const unsigned long long x = __hip_atomic_load(&ws.acc1[(threadIdx.x & 63) * kAcc2Stride], BXI_RLX, BXI_AGENT);
    const double arrived = (double)wave_total_i32((int)(x >> 40)), s = wave_total_f64((double)(x & ((1ull << 40) - 1ull)));   // exact: integers far below 2^31 / 2^53
    *total = s;
    return arrived == (double)nwork;
}

// The finisher's round: everything the two loss values are made of, requested together -- the tile arrivals and sums W pw
// (8 words per instance, arrival count and sum in one word), the dice granules, the count words -- so that the round in
// which everything turns out to be complete is also the round that delivers the data (three dependent rounds past the
// caches, ~1 us each, used to follow the last tile's arrival: check, sum W, then the sums again; the launch ends on them).
// Instances [b0, b0 + 64).  Returns whether all of them are complete; adds their sums.
__device__ __forceinline__ bool finisher_round(const EvalWs& ws, int N, int b0, unsigned int expect, double* num, float* dsum) {
    const int lane = threadIdx.x & 63, i = b0 + lane;
    unsigned long long x = 0ull, dg = 1ull << 32;
    if (i < N) {
        unsigned long long w[kAcc2Split];
#pragma unroll
        for (int sub = 0; sub < kAcc2Split; ++sub) w[sub] = __hip_atomic_load(acc2_word(ws.acc2, i, sub), BXI_RLX, BXI_AGENT);
        dg = __hip_atomic_load(&ws.dice[i], BXI_RLX, BXI_AGENT);
#pragma unroll
        for (int sub = 0; sub < kAcc2Split; ++sub) x += w[sub];
    }
    const bool have = i >= N || ((unsigned int)(x >> 52) == expect && (dg >> 32) != 0ull);
    if (!__all(have)) return false;
    const long long fixed = (long long)(x & ((1ull << 52) - 1ull)) - ((long long)expect << 24);   // the +1 per tile
    *num += wave_total_f64(i < N ? (double)fixed : 0.0);
    const float dv = i < N ? __uint_as_float((unsigned int)dg) : 0.f;
    const int m = min(64, N - b0);
    for (int k = 0; k < m; ++k) *dsum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), k));   // index order
    return true;
}

__device__ __forceinline__ void write_losses(const LossState& st, int N, float warmup, double total_w, double num, float dsum,
                                             float upp, float upw, float* __restrict__ losses) {
    if ((threadIdx.x & 63) == 0) {
        const float denom = fmaxf((float)total_w, 1.f);                      // weights.sum().clamp(min=1.0), :1328
        losses[0] = dsum / (float)N;                                         // .mean(), :143
        losses[1] = (float)((num / (double)kNumScale) / (double)denom) * warmup;   // :1327-1332
        if (st.scale) { *st.scale = warmup / denom; st.applied[0] = upp; st.applied[1] = upw; }
    }
}

// ---- math wave ---------------------------------------------------------------------------------------------------
template <int D, int R>
__device__ __forceinline__ void math_tile(const InstArgs& a, const float* __restrict__ lab, const EvalWs& ws, const LossState& st,
                                          const WorkRec2& wr, float warmup, float upp, float upw, float* __restrict__ losses,
                                          float* __restrict__ g_logits, float* gbuf /* LDS [R + 1][64] of this wave */,
                                          double& total_w, bool& have_total, int nwork, int tix) {
    constexpr int RD = TG<D, R>::RD;
    const int lane = threadIdx.x & 63;
    const int h = a.h, w = a.w, n = wr.n;
    const int64_t P = (int64_t)h * w;
    const float* Lg = a.logits + (int64_t)n * P;
    float g[R];
    float num = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) g[j] = 0.f;
    bool slow = wr.zero_bit != 0;
    if (!slow) {
        float x[RD], L[RD], A[RD], B[RD];
        const float* lp = lab + (int64_t)wr.img * 3 * P;
        load_plane<D, R>(Lg, wr, h, w, lane, x);
        load_plane<D, R>(lp, wr, h, w, lane, L);
        load_plane<D, R>(lp + P, wr, h, w, lane, A);
        load_plane<D, R>(lp + 2 * P, wr, h, w, lane, B);
        const TileFlags f = tile_flags<D, R>(wr, h, w, a.stride, lane);
        DirMasks m[4];
        dir_masks<D>(f, m);
        // Rolling window over the rows: at step i the pairs (row i -> rows i, i + D) are evaluated; what the rows above
        // contributed is final then, so row i's gradient is collected (and its registers die) inside the loop.
        // Per pixel: (a, b) = (sigmoid(x), sigmoid(-x)), t = a - b, u = a b.  Per pair (p, q):
        //   S = a_p a_q + b_p b_q ; pw = -log S ; d pw / d x_p = -t_q u_p / S ; d pw / d x_q = -t_p u_q / S.
        // S cannot underflow while every |x| <= 34 (then min(a, b) >= 1.7e-15 and S >= 3e-15); tiles with a larger logit
        // take the log-space path below.
        float pa_[RD], pb_[RD], pt_[RD], pu_[RD];    // this lane, rows [i, i + D] live
        float aR[RD], bR[RD], tR[RD], uR[RD], LR[RD], AR[RD], BR[RD];   // the lane D to the right, rows [i, i + D] live
        float gq[RD], gR[RD];                        // gradient of this lane's pixels / of lane + D's
        bool sat = false;
#pragma unroll
        for (int j = 0; j < RD; ++j) { gq[j] = 0.f; gR[j] = 0.f; sat |= !(fabsf(x[j]) <= 34.f); }
        // pair weights as bytes, four rows per word: cw = W[k,A] + W[7-k,B] (gradient), dw = the same restricted to
        // pixels this tile owns (loss sum)
        uint32_t cw[4][(R + D + 3) / 4], dw[4][(R + D + 3) / 4];
#pragma unroll
        for (int dir = 0; dir < 4; ++dir)
#pragma unroll
            for (int q4 = 0; q4 < (R + D + 3) / 4; ++q4) {
                cw[dir][q4] = spread4((m[dir].mA >> (4 * q4)) & 15u) + spread4((m[dir].mB >> (4 * q4)) & 15u);
                dw[dir][q4] = spread4((m[dir].nA >> (4 * q4)) & 15u) + spread4((m[dir].nB >> (4 * q4)) & 15u);
            }
        BXI_TW(1, tix, 1);
#define BXI_ROW(j)                                                                                                  \
        {                                                                                                           \
            const float2 s = sig_pair(x[j]); pa_[j] = s.x; pb_[j] = s.y; pt_[j] = s.x - s.y; pu_[j] = s.x * s.y;    \
            aR[j] = lane_plus<D>(pa_[j]); bR[j] = lane_plus<D>(pb_[j]); tR[j] = aR[j] - bR[j]; uR[j] = aR[j] * bR[j]; \
            LR[j] = lane_plus<D>(L[j]); AR[j] = lane_plus<D>(A[j]); BR[j] = lane_plus<D>(B[j]);                     \
        }
#pragma unroll
        for (int j = 0; j < D; ++j) BXI_ROW(j)
        BXI_TW(1, tix, 2);
        // one unordered pair: A = (row ra, this lane) ; B = (row rb of the lane `q` names) ; num collects -log2 S
#define BXI_PAIR(i, ra, rb, qa, qb, qt, qu, qL, qA, qB, dir, GA, GB)                                              \
        {                                                                                                           \
            const bool pn = n2_of(L[ra], A[ra], B[ra], qL[rb], qA[rb], qB[rb]) <= wr.n2max;                         \
            const float gw = pn ? (float)((cw[dir][(i) >> 2] >> (8 * ((i) & 3))) & 255u) : 0.f;                     \
            const float nw = pn ? (float)((dw[dir][(i) >> 2] >> (8 * ((i) & 3))) & 255u) : 0.f;                     \
            const float S = pa_[ra] * qa[rb] + pb_[ra] * qb[rb];                    /* P(y_A == y_B) */            \
            num -= nw * __builtin_amdgcn_logf(S);                                   /* v_log_f32 = log2 */         \
            const float mm = gw * __builtin_amdgcn_rcpf(S);                                                         \
            GA -= mm * qt[rb] * pu_[ra];                                                                            \
            GB -= mm * pt_[ra] * qu[rb];                                                                            \
        }
#pragma unroll
        for (int i = 0; i < R + D; ++i) {
            const int j = i + D;
            BXI_ROW(j)
            if (i >= D) BXI_PAIR(i, i, i, aR, bR, tR, uR, LR, AR, BR, 0, gq[i], gR[i])
            BXI_PAIR(i, j, i, aR, bR, tR, uR, LR, AR, BR, 1, gq[j], gR[i])
            BXI_PAIR(i, i, j, pa_, pb_, pt_, pu_, L, A, B, 2, gq[i], gq[j])
            BXI_PAIR(i, i, j, aR, bR, tR, uR, LR, AR, BR, 3, gq[i], gR[j])
            if (i >= D) {     // row i is complete: collect what the lane D to the left computed for it
                const float fromL = lane_minus<D>(gR[i]);
                g[i - D] = gq[i] + (lane >= D ? fromL : 0.f);
            }
        }
        num *= 0.69314718055994531f;
#undef BXI_ROW
#undef BXI_PAIR
        slow = __any(sat);
    }
    if (slow) {      // workgroup-divergent but wave-uniform; rare
        const int cnt_unused = slow_tile<D, R>(Lg, lab, P, make_int4(wr.r0, wr.r1, wr.c0, wr.c1), wr.img, wr.tile_r0, wr.tile_c0, wr.n2max,
                                               wr.zero_bit, wr.vr, wr.vc, wr.hc1, h, w, a.stride, lane, true, gbuf);
        num = gbuf[R * 64 + lane];
        (void)cnt_unused;
#pragma unroll
        for (int j = 0; j < R; ++j) g[j] = gbuf[j * 64 + lane];
    }
    // ---- epilogue: one round of polls for everything the stores need, the arrival issued before and consumed after them ----
    BXI_TW(1, tix, 3);
    num = wave_total_f32(num);
    long long fx = (long long)(num * kNumScale) + (1ll << 24);           // + 1.0: keeps the packed field non-negative
    {   // converted here, under the polls' latency, not between the stores and the arrival (the compiler sinks it there: lane 0 only)
        int lo = (int)(fx & 0xffffffffll), hi = (int)(fx >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        fx = ((long long)hi << 32) | (long long)(unsigned int)lo;
    }
    const int c = wr.tile_c0 - D + lane;
    const bool col_owned = g_logits && lane >= D && lane < 64 - D && c < wr.hc1;
    const bool row_lane = g_logits && lane < R && wr.tile_r0 + lane < h;
    // sum W: every count wave has arrived (they precede the math waves in the grid); projection coefficients: the leader of
    // this instance publishes one 8-byte granule per column / row (write-through; the leaders precede every tile wave and
    // never wait).  Both are read past this XCD's caches, in the same round.
    bool ok = true;
    unsigned long long ck = 0ull, rk = 0ull;
    for (unsigned spins = 0;; ++spins) {
        bool have = true;
        double sw = 0.0;
        bool counted = true;
        if (!have_total) counted = counts_complete(ws, nwork, &sw);       // wave-uniform
        if (col_owned) { ck = __hip_atomic_load(&st.colk[(int64_t)n * w + c], BXI_RLX, BXI_AGENT); have &= (unsigned int)ck != 0xffffffffu; }
        if (row_lane) { rk = __hip_atomic_load(&st.rowk[(int64_t)n * h + wr.tile_r0 + lane], BXI_RLX, BXI_AGENT); have &= (unsigned int)rk != 0xffffffffu; }
        if (counted && __all(have)) {
            if (!have_total) { total_w = sw; have_total = true; }
            break;
        }
        if (spins > kSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && lane == 0 && st.status) atomicOr(st.status, 1);
    BXI_TW(1, tix, 4);
    if (g_logits) {
        const int carg = col_owned ? (int)(unsigned int)ck : -1;
        const float gc = __uint_as_float((unsigned int)(ck >> 32));
        const int rarg_l = row_lane ? (int)(unsigned int)rk : -1;
        const float gr_l = __uint_as_float((unsigned int)(rk >> 32));
        const float scale = upw * (warmup / fmaxf((float)total_w, 1.f));
        char* G = reinterpret_cast<char*>(g_logits + (int64_t)n * P);      // scalar base + 32-bit byte offset
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = wr.tile_r0 + j;
            const int ra = __builtin_amdgcn_readlane(rarg_l, j);
            const float gr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gr_l), j));
            if (col_owned && r < h) {
                float sp = 0.f;
                if (carg == r) sp += gc;
                if (ra == c) sp += gr;
                *reinterpret_cast<float*>(G + (uint32_t)(r * w + c) * 4u) = g[j] * scale + sp * upp;
            }
        }
    }
    BXI_TW(1, tix, 5);
    // ---- this tile's share of sum W pw + its arrival: one atomic without return, the wave does not wait for it (a
    // returning atomic on these 32 words costs ~3 us here); the finisher block at the end of the grid watches the counts
    if (lane == 0)
        __hip_atomic_fetch_add(acc2_word(ws.acc2, n, wr.tile_r0 / R + wr.tile_c0 / TG<D, R>::TW), (1ull << 52) + (unsigned long long)fx, BXI_RLX, BXI_AGENT);
    BXI_TW(1, tix, 6);
}

// ---- leader workgroup --------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum4(float (&v)[4], float* red /*[16]*/) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_total_f32(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[(threadIdx.x >> 6) * 4 + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
}
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void leader_block(const InstArgs& a, int dil, int R, const EvalWs& ws, const LossState& st, int n,
                                             float upp, float* __restrict__ g_logits, unsigned char* smem, float* red) {
    const int h = a.h, w = a.w, tid = threadIdx.x;
    float* xs = reinterpret_cast<float*>(smem);   // [w] sigmoid of the column maxima, then their unit gradients
    float* ys = xs + w;                           // [h]
    int* carg = reinterpret_cast<int*>(ys + h);   // [w]
    int* rarg = carg + w;                         // [h]
    const InstRec rec = ws.inst[n];
    const InstBox ib = inst_from_rec(rec, dil, h, w);
    float sums[4] = {0.f, 0.f, 0.f, 0.f};   // I_x, U_x, I_y, U_y
    // the partial maxima of a column / row: up to eight loads in flight at once (the band count is launch data, and a loop
    // of load -> compare would walk through L2 once per band: 7 x 0.5 us at 200 rows)
    auto best_key = [](const unsigned long long* __restrict__ part, int n_part, int64_t stride) {
        unsigned long long k = part[0];
        for (int s0 = 0; s0 < n_part; s0 += 8) {
            unsigned long long o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = part[(int64_t)min(s0 + u, n_part - 1) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) k = o[u] > k ? o[u] : k;
        }
        return k;
    };
    for (int i = tid; i < max(w, h); i += 256) {
        const bool is_c = i < w, is_r = i < h;
        // both keys requested before either is used
        const unsigned long long kc = best_key(ws.colpart + (int64_t)n * ws.n_cb * w + (is_c ? i : 0), ws.n_cb, w);
        const unsigned long long kr = best_key(ws.rowkey + (int64_t)n * ws.n_rp * h + (is_r ? i : 0), ws.n_rp, h);
        if (is_c) {
            const int c = i;
            const float X = sigmoid_acc(unpack_val(kc));
            const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
            xs[c] = X; carg[c] = (int)unpack_idx(kc);
            sums[0] += X * TX; sums[1] += X * X + TX * TX;
        }
        if (is_r) {
            const int r = i;
            const float Y = sigmoid_acc(unpack_val(kr));
            const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
            ys[r] = Y; rarg[r] = (int)unpack_idx(kr);
            sums[2] += Y * TY; sums[3] += Y * Y + TY * TY;
        }
    }
    BXI_TW(3, n, 1);
    block_sum4(sums, red);
    BXI_TW(3, n, 2);
    const float Ix = sums[0], Ux = sums[1] + 1e-5f, Iy = sums[2], Uy = sums[3] + 1e-5f;
    if (tid == 0)   // :130, summed over both axes :143
        __hip_atomic_store(&ws.dice[n], (1ull << 32) | (unsigned long long)__float_as_uint((1.f - 2.f * Ix / Ux) + (1.f - 2.f * Iy / Uy)),
                           BXI_RLX, BXI_AGENT);       // the datum is its own flag
    if (g_logits) {
        // dice = 1 - 2I/U ; d dice/d u_j = (-2 t_j U + 4 I u_j) / U^2 ; chain through sigmoid ; mean over N
        const float invN = 1.f / (float)a.N;
        for (int c = tid; c < w; c += 256) {
            const float X = xs[c];
            const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
            const float gv = invN * ((-2.f * TX * Ux + 4.f * Ix * X) / (Ux * Ux)) * X * (1.f - X);
            xs[c] = gv;     // one aligned 8-byte write-through store per column: the datum is its own flag
            __hip_atomic_store(&st.colk[(int64_t)n * w + c], ((unsigned long long)__float_as_uint(gv) << 32) | (unsigned int)carg[c],
                               BXI_RLX, BXI_AGENT);
        }
        for (int r = tid; r < h; r += 256) {
            const float Y = ys[r];
            const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
            const float gv = invN * ((-2.f * TY * Uy + 4.f * Iy * Y) / (Uy * Uy)) * Y * (1.f - Y);
            ys[r] = gv;
            __hip_atomic_store(&st.rowk[(int64_t)n * h + r], ((unsigned long long)__float_as_uint(gv) << 32) | (unsigned int)rarg[r],
                               BXI_RLX, BXI_AGENT);
        }
        BXI_TW(3, n, 3);
        __syncthreads();      // xs / ys now hold the gradients for every thread
        // projection gradient at the arg-max positions OUTSIDE the tiles (prep_kernel left zeros there; the math waves
        // own every pixel of the tile hull: rows of the R-aligned tiles x columns of the dilated box)
        const int hr0 = ib.any ? (ib.dil.r0 / R) * R : 0, hr1 = ib.any ? min(h, ((ib.dil.r1 + R - 1) / R) * R) : 0;
        const int hc0 = ib.dil.c0, hc1 = ib.any ? ib.dil.c1 : 0;
        float* G = g_logits + (int64_t)n * h * w;
        for (int c = tid; c < w; c += 256) {
            const int r = carg[c];
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t) {
                float v = xs[c];
                if (rarg[r] == c) v += ys[r];
                G[(int64_t)r * w + c] = v * upp;
            }
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg[r];
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t && carg[c] != r) G[(int64_t)r * w + c] = ys[r] * upp;
        }
    }
}

// grid: [N leader blocks][n_cb count blocks][n_cb math blocks][finisher]; a count / math block = 4 independent tile waves
// striding through the work list; the finisher (one wave) writes the two loss values once everybody has arrived.  Every wait in a math wave is for a workgroup EARLIER in the grid (leader, count waves), and
// those never wait themselves, so the launch cannot stall on an un-dispatched workgroup whatever its size.
template <int D, int R>
__global__ __launch_bounds__(256, (R == 4 ? 3 : 2)) void pair_kernel(const WorkRec2* __restrict__ work, const int* __restrict__ nwork_p,
                                                   const float* __restrict__ lab, const float* __restrict__ up_prj,
                                                   const float* __restrict__ up_pw, int n_cb, int dil, float warmup,
                                                   float* __restrict__ losses, float* __restrict__ g_logits, InstArgs a, EvalWs ws,
                                                   LossState st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[16];
    const int blk = (int)blockIdx.x;
    const float upp = up_prj ? *up_prj : 1.f, upw = up_pw ? *up_pw : 1.f;
    if (blk < a.N) {                                                   // ---- leader of instance blk
        BXI_TW(3, blk, 0);
        leader_block(a, dil, R, ws, st, blk, upp, g_logits, smem, red);
        BXI_TW(3, blk, 4);
        BXI_TW(3, blk, 5);
        return;
    }
    if (blk == (int)gridDim.x - 1) {                                   // ---- finisher: the last block of the grid, one wave
        if (threadIdx.x >= 64) return;
        // every leader and tile wave precedes this block in the grid and none of them waits for it
        const int lane = threadIdx.x;
        const int nwork = *nwork_p;
        bool ok = false;
        double total_w = 0.0, num = 0.0;
        float dsum = 0.f;
        for (unsigned spins = 0; spins <= kSpinLimit; ++spins) {
            num = 0.0; dsum = 0.f;
            bool all = counts_complete(ws, nwork, &total_w);            // issued with the first pass's loads, used after them
            for (int b0 = 0; b0 < a.N && all; b0 += 64)                    // (> 64 instances: the passes follow one another)
                all = finisher_round(ws, a.N, b0, b0 + lane < a.N ? ws.expect[b0 + lane] : 0u, &num, &dsum) && all;
            if (all) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0 && st.status) atomicOr(st.status, 2);
        write_losses(st, a.N, warmup, total_w, num, dsum, upp, upw, losses);
        return;
    }
    const int nwork = *nwork_p;
    const int wave = (int)(threadIdx.x >> 6);
    const bool counting = blk < a.N + n_cb;
    const int first = ((counting ? blk - a.N : blk - a.N - n_cb) * kWaves) + wave;
    const int stride_w = n_cb * kWaves;
    const int tix = first;
    (void)tix;
    BXI_TW(counting ? 2 : 1, tix, 0);
    float* gbuf = reinterpret_cast<float*>(smem) + wave * ((R + 1) * 64);
    double total_w = 0.0;
    bool have_total = false;
    for (int wi = first; wi < nwork; wi += stride_w) {
        WorkRec2 wr = work[wi];
        // the record is the same in every lane: as scalars, the instance's planes become scalar bases and the tile's 32 loads take
        // (scalar base + one shared 32-bit lane offset per row) instead of a 64-bit address computed per load
#define BXI_SC(f) wr.f = __builtin_amdgcn_readfirstlane(wr.f)
        BXI_SC(r0); BXI_SC(r1); BXI_SC(c0); BXI_SC(c1); BXI_SC(img); BXI_SC(n); BXI_SC(tile_r0); BXI_SC(tile_c0);
        BXI_SC(zero_bit); BXI_SC(vr); BXI_SC(vc); BXI_SC(hc1);
#undef BXI_SC
        wr.n2max = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wr.n2max)));
        if (counting) count_tile<D, R>(a, lab, ws, wr, tix);
        else math_tile<D, R>(a, lab, ws, st, wr, warmup, upp, upw, losses, g_logits, gbuf, total_w, have_total, nwork, tix);
    }
}

// ---- rescale: g_logits finished for the factors recorded in `state` -> finished for (g_prj, g_pw) ----------------
// grid (8, N).  No-op when the factors are the recorded ones (the usual case: loss.backward() seeds both terms with 1).
// The record is not updated (every block reads it): at most one effective rescale per evaluation.
__global__ __launch_bounds__(256) void rescale_kernel(InstArgs a, int dil, LossState st, const float* __restrict__ g_prj,
                                                      const float* __restrict__ g_pw, float* __restrict__ g_logits) {
    const float np = *g_prj, nw = *g_pw, op = st.applied[0], ow = st.applied[1];
    if (np == op && nw == ow) return;
    const int R = st.status[1];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int h = a.h, w = a.w;
    const InstRec rec = st.inst[n];
    const InstBox ib = inst_from_rec(rec, dil, h, w);
    const int hr0 = ib.any ? (ib.dil.r0 / R) * R : 0, hr1 = ib.any ? min(h, ((ib.dil.r1 + R - 1) / R) * R) : 0;
    const int hc0 = ib.dil.c0, hc1 = ib.any ? ib.dil.c1 : 0;
    const float ratio = nw / ow;                      // recorded g_pw == 0 cannot be rescaled (documented)
    float* G = g_logits + (int64_t)n * h * w;
    const unsigned long long* ckp = st.colk + (int64_t)n * w; const unsigned long long* rkp = st.rowk + (int64_t)n * h;
    auto carg = [&](int c) { return (int)(unsigned int)ckp[c]; };
    auto rarg = [&](int r) { return (int)(unsigned int)rkp[r]; };
    auto gcol = [&](int c) { return __uint_as_float((unsigned int)(ckp[c] >> 32)); };
    auto grow = [&](int r) { return __uint_as_float((unsigned int)(rkp[r] >> 32)); };
    const int cw = hc1 - hc0, rows = hr1 - hr0;
    const int per = (rows + gridDim.x - 1) / gridDim.x;
    const int ra = hr0 + s * per, rb = min(hr1, ra + per);
    const int npx = cw > 0 && rb > ra ? (rb - ra) * cw : 0;
    for (int i = tid; i < npx; i += 256) {            // G = ow*s*d + op*sp  ->  nw*s*d + np*sp
        const int r = ra + i / cw, c = hc0 + i % cw;
        float sp = 0.f;
        if (carg(c) == r) sp += gcol(c);
        if (rarg(r) == c) sp += grow(r);
        const float v = G[(int64_t)r * w + c];
        G[(int64_t)r * w + c] = (v - op * sp) * ratio + np * sp;
    }
    if (s == 0) {                                     // arg-max positions outside the hull hold op * sp
        for (int c = tid; c < w; c += 256) {
            const int r = carg(c);
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t) {
                float v = gcol(c);
                if (rarg(r) == c) v += grow(r);
                G[(int64_t)r * w + c] = v * np;
            }
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg(r);
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t && carg(c) != r) G[(int64_t)r * w + c] = grow(r) * np;
        }
    }
}

__global__ void zero_losses2_kernel(float* losses) { losses[0] = 0.f; losses[1] = 0.f; }

// ---- host side ---------------------------------------------------------------------------------------------------
size_t eval_ws_bytes(int N, int h, int w) { return carve_eval(nullptr, N, h, w, nullptr); }

// compute units of the current device (256 on an MI355X in SPX mode, 32 per partition in CPX): the grids are sized so that a
// launch is resident in one round.  Cached per device ordinal; a wrong value costs time, never correctness.
static int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

static int tile_rows_for(int N, int h, int w, int dil) {
    (void)h; (void)w; (void)dil;
    // A tile wave's time is the length of its dependent chain, so 4-row tiles (6 row steps instead of 10, 1.2x the pair
    // evaluations in total, three waves a SIMD at <= 168 VGPRs) win while the tile waves are resident together: measured
    // 11.4 vs 12.2 us at 32 instances, 19.8 vs 20.9 us at 64, 41.8 vs 40.6 us at 128 (2 x 800 x 1024).  (Before the arrival
    // words were split they lost everywhere: twice the atomics per word.)  The number of tiles is device data; the instance
    // count is what the host has.  Developer knob: BXI_TILE_ROWS.
    return N <= 96 ? 4 : 8;
}
int eval_tile_rows(int N, int h, int w, int dil) { return tile_rows_for(N, h, w, dil); }

template <int D, int R>
static void launch_pair(hipStream_t s, int grid, size_t lds, const InstArgs& a, const float* lab, int dil, float warmup,
                        const EvalWs& ws, const LossState& st, float* losses, float* g_logits, const float* up_prj,
                        const float* up_pw, int n_cb) {
    BXI_LAUNCH("pair", s, (pair_kernel<D, R>), dim3((unsigned)grid), dim3(256), lds, s, (const WorkRec2*)ws.work, (const int*)ws.nwork, lab,
               up_prj, up_pw, n_cb, dil, warmup, losses, g_logits, a, ws, st);
}

bool fused_eval_supported(int dil) { return dil >= 1 && dil <= kMaxDilFused; }

// One evaluation, two launches.  lab: [B,3,h,w] f32 scratch (prep fills, pair reads).
int launch_fused_eval(const bxi_image_batch* batch, float* lab, float color_thresh, const bxi_instances* in, int dil, float warmup,
                      const float* up_prj, const float* up_pw, float* losses, float* g_logits, void* state, void* workspace,
                      size_t workspace_bytes, int force_rows, void* stream, const DynArgs* head, int head_C) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (!losses || !batch) return BXI_ERR_NULL_POINTER;
    hipStream_t s = as_stream(stream);
    PoolArgs pa = {};
    if (batch->Hc != in->Hc || batch->Wc != in->Wc || batch->B != in->B) return BXI_ERR_BAD_SHAPE;
    rc = fill_pool_args(batch, nullptr, lab, pa);
    if (rc != BXI_OK) return rc;
    if (batch->B > 0 && (!batch->imgs || !lab)) return BXI_ERR_NULL_POINTER;
    if (batch->image_masks) return BXI_ERR_UNSUPPORTED;   // explicit masks: use bxi_color_affinity_f32 + bits
    if (a.N == 0) {
        BXI_LAUNCH("zero_losses", s, zero_losses2_kernel, dim3(1), dim3(1), 0, s, losses);
        return check_launch();
    }
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    if (g_logits && !state) return BXI_ERR_NULL_POINTER;
    const size_t need = carve_eval(nullptr, a.N, a.h, a.w, nullptr);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    EvalWs ws;
    carve_eval(workspace, a.N, a.h, a.w, &ws);
    LossState st = {};
    if (state) {
        if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
        carve_state(state, a.N, a.h, a.w, &st);
    }
    const int vec = ((a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0 &&
                     (!g_logits || (reinterpret_cast<uintptr_t>(g_logits) & 15) == 0)) ? 1 : 0;
    static const int env_rows = [] { const char* e = getenv("BXI_TILE_ROWS"); return e ? atoi(e) : 0; }();   // developer knob
    if (!force_rows) force_rows = env_rows;
    const int R = force_rows == 4 || force_rows == 8 ? force_rows : tile_rows_for(a.N, a.h, a.w, dil);

    // ---- launch 1 --------------------------------------------------------------------------------------------------
    int n_pool = 0, n_items = 0;
    const int n_tab = (a.N + kWaves - 1) / kWaves;
    const int n_stream = a.N * ((a.h + kSBlk - 1) / kSBlk);
    if (batch->B > 0) {
        if (pool_vec_ok(batch, a.stride)) {
            // one item = the 4 input rows of 64 pooled pixels.  The whole launch should be resident at once (5 workgroups
            // per CU at <= 96 VGPRs): a pool workgroup takes several items, the next one's loads in flight, when it is not.
            n_items = batch->B * a.h * ((a.w + 63) / 64);
            const int room = 5 * device_cus() - n_tab - (head ? 0 : n_stream);
            const int per = room > 0 ? (n_items + room - 1) / room : 8;
            n_pool = (n_items + (per < 1 ? 1 : per) - 1) / (per < 1 ? 1 : per);
        }
        else {                               // unaligned canvas / other strides: separate scalar pooling launch
            rc = launch_pool(batch, a.stride, nullptr, lab, s);
            if (rc != BXI_OK) return rc;
        }
    }
    size_t lds1 = sizeof(double) * (256 + 3 * 64) + sizeof(int) * 4 * 3 * 64;
    if (lds1 < 8 * (size_t)kWaves * a.w) lds1 = 8 * (size_t)kWaves * a.w;
    if (head) {
        // the head-fused first launch (factor 2, vector rows, the pooled fast path): tables, pool blocks, head tiles
        if (head->factor != 2 || !vec || head->H * 2 != a.h || head->W * 2 != a.w || head->N != a.N || head->B != in->B ||
            (batch->B > 0 && !pool_vec_ok(batch, a.stride)))
            return BXI_ERR_UNSUPPORTED;
        const int tiles = ((head->H + kYR - 1) / kYR) * ((head->W + kYC - 1) / kYC);
        ws.n_cb = (head->H + kYR - 1) / kYR;
        ws.n_rp = (head->W + kYC - 1) / kYC;
        const size_t lds_head = 8 * 4 * 64 + sizeof(float) * (16 * 64 + (kYR + 2) * (kYC + 2));
        if (lds1 < lds_head) lds1 = lds_head;
        if (lds1 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
        float* logits_out = const_cast<float*>(a.logits);
        const unsigned grid1 = (unsigned)(n_tab + n_pool + a.N * tiles);
        if (head_C == 16 && head->rel)
            BXI_LAUNCH("head_prep", s, (head_prep_kernel<16, true>), dim3(grid1), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R, color_thresh, ws, st, g_logits, *head, head->params, logits_out);
        else if (head_C == 16)
            BXI_LAUNCH("head_prep", s, (head_prep_kernel<16, false>), dim3(grid1), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R, color_thresh, ws, st, g_logits, *head, head->params, logits_out);
        else if (head_C == 8 && head->rel)
            BXI_LAUNCH("head_prep", s, (head_prep_kernel<8, true>), dim3(grid1), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R, color_thresh, ws, st, g_logits, *head, head->params, logits_out);
        else if (head_C == 8)
            BXI_LAUNCH("head_prep", s, (head_prep_kernel<8, false>), dim3(grid1), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R, color_thresh, ws, st, g_logits, *head, head->params, logits_out);
        else
            return BXI_ERR_UNSUPPORTED;
    } else {
        if (lds1 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
        BXI_LAUNCH("prep", s, prep_kernel, dim3((unsigned)(n_tab + n_stream + n_pool)), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R,
                   color_thresh, ws, st, g_logits, vec);
    }
    rc = check_launch();
    if (rc != BXI_OK) return rc;

    // ---- launch 2 --------------------------------------------------------------------------------------------------
    const int cap = eval_cap(a.N, a.h, a.w, dil, R);
    int n_cb = (cap + kWaves - 1) / kWaves;
    // the list length is device data: the tile waves stride through it.  3 (R = 4: <= 168 VGPRs) or 2 (R = 8) workgroups per
    // CU are resident: leaders + count + math blocks should fit in one round.
    const int room2 = ((R == 4 ? 3 : 2) * device_cus() - a.N) / 2;
    if (n_cb > (room2 > 64 ? room2 : 64)) n_cb = room2 > 64 ? room2 : 64;
    size_t lds2 = sizeof(float) * (size_t)kWaves * (R + 1) * 64;
    const size_t lds_leader = 2 * sizeof(float) * (size_t)(a.h + a.w) + 16;
    if (lds2 < lds_leader) lds2 = lds_leader;
    if (lds2 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
    const int grid = a.N + 2 * n_cb + 1;                 // + the finisher
#define BXI_PAIR_CASE(DD)                                                                                                    \
    case DD:                                                                                                                 \
        if (R == 4) launch_pair<DD, 4>(s, grid, lds2, a, lab, dil, warmup, ws, st, losses, g_logits, up_prj, up_pw, n_cb);   \
        else launch_pair<DD, 8>(s, grid, lds2, a, lab, dil, warmup, ws, st, losses, g_logits, up_prj, up_pw, n_cb);          \
        break;
    switch (dil) {
        BXI_PAIR_CASE(1) BXI_PAIR_CASE(2) BXI_PAIR_CASE(3) BXI_PAIR_CASE(4)
        default: return BXI_ERR_UNSUPPORTED;
    }
#undef BXI_PAIR_CASE
    return check_launch();
}

int launch_rescale(const bxi_instances* in, const float* g_prj, const float* g_pw, int dil, const void* state, float* g_logits,
                   void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (a.N == 0) return BXI_OK;
    if (!g_prj || !g_pw || !state || !g_logits) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
    LossState st;
    carve_state(const_cast<void*>(state), a.N, a.h, a.w, &st);
    hipStream_t s = as_stream(stream);
    BXI_LAUNCH("rescale", s, rescale_kernel, dim3(8, a.N), dim3(256), 0, s, a, dil, st, g_prj, g_pw, g_logits);
    return check_launch();
}

}  // namespace bxi

#ifdef BXI_TRACE
extern "C" int bxi_debug_set_trace2(void* buf) {   // developer builds only
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_trace), &buf, sizeof(buf));
}
#endif
