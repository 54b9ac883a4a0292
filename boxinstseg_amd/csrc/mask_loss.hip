// mask_loss.hip -- BoxInst projection + pairwise loss from PRECOMPUTED colour-affinity bits (bxi_boxinst_loss_fwd_bwd_f32
// + bxi_boxinst_loss_backward_f32): the round-1 three-launch path, kept for callers that already hold the affinity words
// of bxi_color_affinity_f32 (explicit image masks, get_targets()-style use) and as an independent cross-check of
// fused_eval.hip (tests compare the two routes).  The evaluation from the network input is fused_eval.hip.
//
// Replaces (reference, LiWentomng/BoxInstSeg) CondInstMaskHead.loss with boxinst_enabled, condinst_head.py:1288-1343:
//   mask_logits.sigmoid() / compute_project_term      :1300, :134-143 (+ dice_coefficient :117-131)
//   pairwise_nlog (CUDA op, pairwise.cu:68-149)       :1321
//   weights = (sim >= thresh) * bitmask, normalise    :1324-1328
//   warm-up                                           :1330-1332
//   stage1        tables + logit streaming: row/column maxima, zero-fill                                HBM stream
//   box_kernel    leaders: dice + projection gradient of each instance ; tiles: pairwise term and its un-normalised
//                 gradient on the box tiles (ordered pairs, tile staged in LDS)                          latency bound
//   loss_apply    (backward) normalise the box tiles, add the projection gradient at the h+w arg-max positions, fold
//                 the upstream gradients in from device memory.
#include "loss_common.hpp"

namespace bxi {

constexpr int kSR = 8;          // rows per streaming tile (stage1): one wave64 per tile
constexpr int kChunk = 256;     // columns per pass: 64 lanes x float4
constexpr int kBR = 8;          // box tile rows    (box_kernel)
constexpr int kBC = 64;         // box tile columns
constexpr int kSlices = 8;      // row slices per instance in loss_apply


struct LossWs {               // carved from the caller's workspace
    unsigned long long* colkey;  // [N,Ts,w] per-streaming-tile column max: packed (logit, first row in tile)
    unsigned long long* rowkey;  // [N,h] packed (max logit, first column)
    unsigned long long* acc;  // [N,2] per instance: sum of W (integer) ; sum of W*pw in 2^-24 fixed point
    struct InstRec* inst;     // [N]  box rectangle + image of every instance (written by stage1)
    struct WorkRec* work;     // [N*Tr*Tc] compacted box tiles (written by stage1)
    int* nwork;               // [1]
    unsigned int* arrive;     // [N+1] arrivals per instance (tiles + leader), [N] = instances complete (zeroed by stage1)
    unsigned int* expect;     // [N]   box tiles of the instance + 1 (written by stage1)
    float* dice;              // [N]
};

struct WorkRec { int r0, r1, c0, c1, img, n, tile_r0, tile_c0; float n2max; int zero_bit, pad0, pad1; };  // 48 B: all a tile needs




static inline int stream_tiles(int h) { return (h + kSR - 1) / kSR; }
static inline int box_tiles(int h, int w) { return ((h + kBR - 1) / kBR) * ((w + kBC - 1) / kBC); }

static size_t carve_ws(void* base, int N, int h, int w, LossWs* ws) {
    const size_t T = (size_t)stream_tiles(h);
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    unsigned long long* colkey = (unsigned long long*)take(sizeof(unsigned long long) * N * T * w);
    unsigned long long* rowkey = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)N * h);
    unsigned long long* acc = (unsigned long long*)take(sizeof(unsigned long long) * 2 * (size_t)(N > 0 ? N : 1));
    InstRec* inst = (InstRec*)take(sizeof(InstRec) * (size_t)(N > 0 ? N : 1));
    WorkRec* work = (WorkRec*)take(sizeof(WorkRec) * (size_t)(N > 0 ? N : 1) * (size_t)box_tiles(h, w));
    int* nwork = (int*)take(sizeof(int));
    unsigned int* arrive = (unsigned int*)take(sizeof(unsigned int) * (size_t)(N + 1));
    unsigned int* expect = (unsigned int*)take(sizeof(unsigned int) * (size_t)(N > 0 ? N : 1));
    float* dice = (float*)take(sizeof(float) * (size_t)(N > 0 ? N : 1));
    if (ws) { ws->colkey = colkey; ws->rowkey = rowkey; ws->acc = acc; ws->inst = inst;
              ws->work = work; ws->nwork = nwork; ws->arrive = arrive; ws->expect = expect;
              ws->dice = dice; }
    return off;
}



// ================================================================================================
// Kernel 1: stage1 = { pool_rgb + Lab workgroups }  ||  { logit streaming workgroups }
// ================================================================================================
// The two halves are independent (image side / logit side), so they share one launch of one-wave
// workgroups: N*Ts streaming waves followed by B*h*w/64 pooling waves, all resident at once
// (about 9 waves per CU at 2x800x1024x32), every wave issuing all of its loads before anything else.


// per-lane box lookup (lanes hold different instances): the image table is walked with a uniform
// loop so that the by-value kernel argument is never indexed per lane.
struct LaneBox { int r0, r1, c0, c1, img, tr0, ntr, tc0, ntc; };
__device__ __forceinline__ LaneBox lane_box(const InstArgs& a, int dil, int m) {
    LaneBox lb = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t g = a.gt_inds[m];
    const float* bp = nullptr;
    for (int b = 0; b < a.gt.B; ++b)
        if (g >= a.gt.first[b] && g < a.gt.first[b + 1]) { bp = a.gt.boxes[b] + 4 * (g - a.gt.first[b]); lb.img = b; }
    if (!bp) return lb;
    const Rect rc = box_rect(bp, a.Hc, a.Wc, a.stride, a.stride / 2, a.h, a.w);
    if (rc.r1 <= rc.r0 || rc.c1 <= rc.c0) return lb;
    lb.r0 = rc.r0; lb.r1 = rc.r1; lb.c0 = rc.c0; lb.c1 = rc.c1;
    const int r0 = max(rc.r0 - dil, 0), r1 = min(rc.r1 + dil, a.h), c0 = max(rc.c0 - dil, 0), c1 = min(rc.c1 + dil, a.w);
    lb.tr0 = r0 / kBR; lb.ntr = (r1 - 1) / kBR - r0 / kBR + 1;
    lb.tc0 = c0 / kBC; lb.ntc = (c1 - 1) / kBC - c0 / kBC + 1;
    return lb;
}

// One wave64 of instance n's first streaming workgroup: exclusive prefix of the box-tile counts of
// instances 0..n-1 (deterministic order, no atomics, no pre-zeroed counter), then this instance's tiles.
__device__ __forceinline__ void build_work_list(const InstArgs& a, int dil, const LossWs& ws, int n) {
    const int lane = threadIdx.x & 63;
    int base = 0, total = 0;
    LaneBox mine = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int m0 = 0; m0 < a.N; m0 += 64) {
        const int m = m0 + lane;
        LaneBox lb = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (m < a.N) lb = lane_box(a, dil, m);
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
            mine.tc0 = __shfl(lb.tc0, src, 64); mine.ntc = __shfl(lb.ntc, src, 64);
        }
        total += __shfl(incl, 63, 64);
    }
    if (n == 0 && lane == 0) { *ws.nwork = total; ws.arrive[a.N] = 0u; }
    const int cnt = mine.ntr * mine.ntc;
    if (lane == 0) {   // per-instance records and zeroed accumulators for box_kernel / loss_apply (next launches)
        InstRec rc; rc.r0 = mine.r0; rc.r1 = mine.r1; rc.c0 = mine.c0; rc.c1 = mine.c1; rc.img = mine.img;
        rc.pad0 = rc.pad1 = rc.pad2 = 0;
        ws.inst[n] = rc;
        ws.acc[2 * n] = 0ull; ws.acc[2 * n + 1] = 0ull;
        ws.expect[n] = (unsigned int)cnt + 1u; ws.arrive[n] = 0u;
    }
    for (int i = lane; i < cnt; i += 64) {
        WorkRec wr;
        wr.r0 = mine.r0; wr.r1 = mine.r1; wr.c0 = mine.c0; wr.c1 = mine.c1; wr.img = mine.img; wr.n = n;
        wr.tile_r0 = (mine.tr0 + i / mine.ntc) * kBR; wr.tile_c0 = (mine.tc0 + i % mine.ntc) * kBC;
        wr.n2max = 0.f; wr.zero_bit = 0; wr.pad0 = wr.pad1 = 0;       // (colour predicate: only used when Lab is given)
        ws.work[base + i] = wr;
    }
}

// One wave64 = one streaming tile of kSR rows x all columns of one instance map.  A lane owns 4
// consecutive columns (float4): a wave-level load/store instruction moves one 1 KiB row segment.
//   - the zero-fill of d loss / d logits needs nothing, so it is issued first (box_kernel later
//     overwrites the box tiles; ~15 % of the map is written twice, in exchange for no dependency);
//   - all kSR row loads are issued back to back, then consumed: per-row max / first arg-max by a
//     64-lane butterfly on a packed 64-bit key (the kSR butterflies are interleaved), per-column
//     max / first arg-max over the tile's rows in registers -> one partial per (tile, column).
__device__ __forceinline__ void stream_tile(const InstArgs& a, int dil, const LossWs& ws,
                                            float* __restrict__ g_logits, int vec, int sb) {
    const int h = a.h, w = a.w;
    const int Ts = (h + kSR - 1) / kSR;
    const int n = sb / Ts, t = sb % Ts;
    const int r0 = t * kSR, r1 = min(h, r0 + kSR);
    const int lane = threadIdx.x & 63;
    const int64_t P = (int64_t)h * w;
    const float* L = a.logits + (int64_t)n * P;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 v[kSR];
    {
        const int c = lane * 4;
#pragma unroll
        for (int i = 0; i < kSR; ++i) v[i] = (r0 + i < r1 && c < w) ? load4(L + (int64_t)(r0 + i) * w, c, w, vec) : ninf;
    }
    if (G)   // non-temporal: these lines are not read again in this launch and need not stay dirty in L2
        for (int cb = 0; cb < w; cb += kChunk) {
            const int c = cb + lane * 4;
            if (c < w) {
#pragma unroll
                for (int i = 0; i < kSR; ++i)
                    if (r0 + i < r1) {
                        if (vec) { typedef float f4v __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((f4v){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4v*>(G + (int64_t)(r0 + i) * w + c)); }
                        else store4(G + (int64_t)(r0 + i) * w, c, w, false, zero);
                    }
            }
        }
    BXI_T(0, blockIdx.x, 2);
    __builtin_amdgcn_s_setprio(2);   // short tail: do not queue behind the pooling waves' long fp64 work
    BXI_T(0, blockIdx.x, 3);

    float rmax[kSR]; int rcol[kSR];        // per-lane row maximum and its first column (across column chunks)
#pragma unroll
    for (int i = 0; i < kSR; ++i) { rmax[i] = -INFINITY; rcol[i] = 0; }
    const int Tsw = Ts;
    for (int cb = 0;;) {
        const int c = cb + lane * 4;
        float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int crow[4] = {0, 0, 0, 0};
        if (c < w) {
#pragma unroll
            for (int i = 0; i < kSR; ++i) {
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
            const int64_t o = ((int64_t)n * Tsw + t) * w + c;
            unsigned long long k4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) k4[j] = pack_max(cmax[j], (uint32_t)crow[j]);
            if (vec) {
                *reinterpret_cast<ulonglong2*>(ws.colkey + o) = make_ulonglong2(k4[0], k4[1]);
                *reinterpret_cast<ulonglong2*>(ws.colkey + o + 2) = make_ulonglong2(k4[2], k4[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < w) ws.colkey[o + j] = k4[j];
            }
        }
        cb += kChunk;
        if (cb >= w) break;
        const int c2 = cb + lane * 4;
#pragma unroll
        for (int i = 0; i < kSR; ++i) v[i] = (r0 + i < r1 && c2 < w) ? load4(L + (int64_t)(r0 + i) * w, c2, w, vec) : ninf;
    }
    BXI_T(0, blockIdx.x, 4);
    // row maxima: kSR independent 32-bit butterflies advanced in lock step (their cross-lane moves
    // pipeline); the arg-max is the lowest lane holding the maximum (lanes own ascending columns),
    // found with one ballot + readlane per row.
    float wmax[kSR];
#pragma unroll
    for (int i = 0; i < kSR; ++i) wmax[i] = rmax[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float o[kSR];
#pragma unroll
        for (int i = 0; i < kSR; ++i) o[i] = __shfl_xor(wmax[i], off, kWave);
#pragma unroll
        for (int i = 0; i < kSR; ++i) wmax[i] = fmaxf(wmax[i], o[i]);
    }
    unsigned long long mine = 0ull;
#pragma unroll
    for (int i = 0; i < kSR; ++i) {
        const unsigned long long who = __ballot(rmax[i] == wmax[i]);
        const int first = who ? __ffsll((long long)who) - 1 : 0;
        const int col = __builtin_amdgcn_readlane(rcol[i], first);
        if (lane == i) mine = pack_max(wmax[i], (uint32_t)col);
    }
    if (lane < kSR && r0 + lane < r1) ws.rowkey[(int64_t)n * h + r0 + lane] = mine;
}

// grid: [N table waves][N*Ts streaming waves].  The table waves (per-instance records + work list of box_kernel) carry
// dependent load chains, so they go first; nothing in this launch waits for them.
// (Round 1 also ran the image pooling here; the evaluation from the network input now lives in fused_eval.hip, and this file
// serves the precomputed-affinity-bits entry point, bxi_boxinst_loss_fwd_bwd_f32.)
__global__ __launch_bounds__(64) void stage1_kernel(InstArgs a, int dil, LossWs ws, float* __restrict__ g_logits, int vec) {
    BXI_T(0, blockIdx.x, 0);
    const int n_tab = a.N;
    if ((int)blockIdx.x < n_tab) build_work_list(a, dil, ws, (int)blockIdx.x);
    else stream_tile(a, dil, ws, g_logits, vec, (int)blockIdx.x - n_tab);
    BXI_T(0, blockIdx.x, 1);
}

// ================================================================================================
// Kernel 2: box_kernel -- N leader workgroups (projection term) + pairwise term on 8 x 64 box tiles
// ================================================================================================
// grid = N + N x ceil(h/8) x ceil(w/64).
// Leader workgroup n (blockIdx < N): reduces stage1's per-tile column partials and row keys of
//   instance n to the h+w maxima, applies sigmoid to those only (max sigmoid = sigmoid max), forms
//   the two dice terms (condinst_head.py:117-143) and the unit projection gradient at each arg-max
//   position (kept in `state` for the backward).  Runs concurrently with the tile workgroups.
// Tile workgroups take their (instance, tile) from the compacted work list stage1 built (at most 1024
// tile workgroups are launched; each strides through the list), so the working ones start together.
//   LDS  pq   [8+2d][64+2P]     (sigmoid(x), sigmoid(-x)) of the tile + halo         (P = d rounded up to 4)
//        bits [8+2d][64+2P]     affinity words of the in-box pixels
//   1. the logits region is loaded float4, aligned;
//   2. per pixel, 8 neighbours (operands prefetched from LDS, branch-free): colour weights from the staged affinity words,
//      S = p_i p_j + q_i q_j, -log S and its gradient, weighted by W[k,p] + W[7-k,q]
//      (gather form: no atomics on the gradient, fixed summation order);
//   3. writes the UN-normalised pairwise gradient of the whole tile (zeros outside the dilated box)
//      and adds its integer partial sums to the instance's accumulators.
__device__ __forceinline__ void block_sum4(float (&v)[4], float* red /*[16]*/) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum_f32(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[(threadIdx.x >> 6) * 4 + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// Arrival protocol (hierarchical ticket).  Every working workgroup of box_kernel -- the box tiles and
// the leader of an instance -- calls this once, after thread 0 has issued its agent-scope
// contributions (atomicAdd into acc / write-through store of dice).  The last arrival of an instance
// arrives for the instance; the last instance computes the two loss values and the normaliser
// (what a separate one-workgroup launch would otherwise do).  Counters are zeroed by stage1.
__device__ __forceinline__ void arrive_and_finish(const LossWs& ws, const LossState& st, int n, int N, float warmup,
                                                  float* __restrict__ losses, int* flag, double* red64, float* dbuf) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's contributions are performed
        int last = 0;
        const unsigned int o1 = __hip_atomic_fetch_add(&ws.arrive[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o1 + 1u == __hip_atomic_load(&ws.expect[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const unsigned int o2 = __hip_atomic_fetch_add(&ws.arrive[N], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (o2 + 1u == (unsigned int)N);
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;                                         // workgroup-uniform
    // ---- the last workgroup of the launch: loss_prj, loss_pairwise, normaliser (one wave, no barriers) ----
    if (tid >= 64) return;
    double cnt = 0.0, num = 0.0;
    float dsum = 0.f;
    for (int base = 0; base < N; base += 64) {
        const int i = base + tid;
        unsigned long long c = 0ull; long long s = 0; float dv = 0.f;
        if (i < N) {   // three loads in flight per lane; written by agent-scope atomics / write-through stores
            c = __hip_atomic_load(&ws.acc[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s = (long long)__hip_atomic_load(&ws.acc[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dv = __hip_atomic_load(&ws.dice[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cnt += wave_sum_f64((double)c);                 // exact: integers far below 2^53
        num += wave_sum_f64((double)s);
        const int m = min(64, N - base);
        for (int k = 0; k < m; ++k) dsum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), k));   // index order
    }
    if (tid == 0) {
        const float denom = fmaxf((float)cnt, 1.f);                         // weights.sum().clamp(min=1.0), :1328
        losses[0] = dsum / (float)N;                                        // .mean(), :143
        losses[1] = (float)((num / (double)kNumScale) / (double)denom) * warmup;   // :1327-1332
        if (st.scale) *st.scale = warmup / denom;
    }
}

__device__ __forceinline__ void leader_block(const InstArgs& a, int dil, const LossWs& ws, const LossState& st, int n,
                                             unsigned char* smem, float* red) {
    const int h = a.h, w = a.w, tid = threadIdx.x;
    const int Ts = (h + kSR - 1) / kSR;
    float* xs = reinterpret_cast<float*>(smem);   // [w] column maxima (as sigmoid), then their gradients
    float* ys = xs + w;                           // [h]
    // ---- every global load of the first batch is issued before the first use -----------------------------
    const InstRec rec = ws.inst[n];
    unsigned long long ck[kMaxT];
    {
        const int c = tid < w ? tid : 0;
#pragma unroll
        for (int u = 0; u < kMaxT; ++u)     // unconditional (index clamped): kMaxT 8-byte loads in flight, no branches
            ck[u] = ws.colkey[((int64_t)n * Ts + min(u, Ts - 1)) * w + c];
    }
    const unsigned long long rk0 = ws.rowkey[(int64_t)n * h + (tid < h ? tid : 0)];
    const InstBox ib = inst_from_rec(rec, dil, h, w);
    // field by field: a whole-struct copy moves the three padding words through scratch, and a kernel that owns
    // scratch pays for it at every wave launch
    if (st.gcol && tid == 0) st.inst[n] = InstRec{rec.r0, rec.r1, rec.c0, rec.c1, rec.img, 0, 0, 0};
    float sums[4] = {0.f, 0.f, 0.f, 0.f};   // I_x, U_x, I_y, U_y
    for (int c = tid; c < w; c += 256) {
        float m = -INFINITY; int mr = 0;
        for (int t0 = 0; t0 < Ts; t0 += kMaxT) {
            if (c >= 256 || t0 > 0) {                 // beyond the prefetched batch (w > 256 or > kMaxT tiles)
#pragma unroll
                for (int u = 0; u < kMaxT; ++u)
                    ck[u] = ws.colkey[((int64_t)n * Ts + min(t0 + u, Ts - 1)) * w + c];
            }
#pragma unroll
            for (int u = 0; u < kMaxT; ++u) {
                const float v = unpack_val(ck[u]);
                if (t0 + u < Ts && v > m) { m = v; mr = (t0 + u) * kSR + (int)unpack_idx(ck[u]); }   // tile order: first row wins ties
            }
        }
        const float X = sigmoid_acc(m);
        const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
        xs[c] = X;
        if (st.colarg) st.colarg[(int64_t)n * w + c] = mr;
        sums[0] += X * TX; sums[1] += X * X + TX * TX;
    }
    for (int r = tid; r < h; r += 256) {
        const unsigned long long k = r < 256 ? rk0 : ws.rowkey[(int64_t)n * h + r];
        const float Y = sigmoid_acc(unpack_val(k));
        const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
        ys[r] = Y;
        if (st.rowarg) st.rowarg[(int64_t)n * h + r] = (int)unpack_idx(k);
        sums[2] += Y * TY; sums[3] += Y * Y + TY * TY;
    }
    BXI_T(2, blockIdx.x, 1);
    block_sum4(sums, red);
    BXI_T(2, blockIdx.x, 2);
    const float Ix = sums[0], Ux = sums[1] + 1e-5f, Iy = sums[2], Uy = sums[3] + 1e-5f;
    if (tid == 0)   // :130, summed over both axes :143 ; write-through, read by the last workgroup
        __hip_atomic_store(&ws.dice[n], (1.f - 2.f * Ix / Ux) + (1.f - 2.f * Iy / Uy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st.gcol) {
        // dice = 1 - 2I/U ; d dice/d u_j = (-2 t_j U + 4 I u_j) / U^2 ; chain through sigmoid ; mean over N
        const float invN = 1.f / (float)a.N;
        for (int c = tid; c < w; c += 256) {
            const float X = xs[c];
            const float TX = (ib.any && c >= ib.box.c0 && c < ib.box.c1) ? 1.f : 0.f;
            st.gcol[(int64_t)n * w + c] = invN * ((-2.f * TX * Ux + 4.f * Ix * X) / (Ux * Ux)) * X * (1.f - X);
        }
        for (int r = tid; r < h; r += 256) {
            const float Y = ys[r];
            const float TY = (ib.any && r >= ib.box.r0 && r < ib.box.r1) ? 1.f : 0.f;
            st.grow[(int64_t)n * h + r] = invN * ((-2.f * TY * Uy + 4.f * Iy * Y) / (Uy * Uy)) * Y * (1.f - Y);
        }
    }
}

// |logit| beyond ~69 makes S = p_i p_j + q_i q_j underflow: redo that pixel's 8 pairs in log space,
// exactly as pairwise.cu:38-61 does.  Out of line and rolled: it runs for saturated logits only.
__device__ __noinline__ float2 pair_logspace_redo(const float* __restrict__ L, const float2* pq, int h, int w, int d,
                                                  int PC, int r, int c, int pi, float2 pp, uint32_t wps, uint32_t wqs) {
    float acc2 = 0.f, dnum = 0.f;
    const float xa = L[(int64_t)r * w + c];
    const float ax = logsig(xa), bx = logsig(-xa);
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int kk = k < 4 ? k : k + 1;                 // skip the centre of the 3x3 window
        const int dy = kk / 3 - 1, dx = kk % 3 - 1;
        const int r2 = r + dy * d, c2 = c + dx * d;
        const uint32_t wp = (wps >> k) & 1u, wq = (wqs >> k) & 1u;
        if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w && (wp + wq)) {
            const float xb = L[(int64_t)r2 * w + c2];
            const float ay = logsig(xb), by = logsig(-xb);
            const float e1 = ax + ay, e0 = bx + by;
            const float nl2 = logsig(fabsf(e1 - e0)) - fmaxf(e1, e0);
            const float2 qq = pq[pi + dy * d * PC + dx * d];
            const float S = pp.x * qq.x + pp.y * qq.y;
            dnum += (float)wp * (nl2 + __logf(fmaxf(S, 1e-30f)));   // replaces the fast-path term
            acc2 += (float)(wp + wq) * (-(expf(ay) - expf(by)) * expf(ax + bx + nl2));
        }
    }
    return make_float2(dnum, acc2);   // (correction to the pixel's loss sum, its gradient)
}

template <bool PREFETCH>
__device__ __forceinline__ void box_body(const InstArgs& a, const uint8_t* __restrict__ bits_in, int dil, float warmup,
                                         const LossWs& ws, const LossState& st, float* __restrict__ losses,
                                         float* __restrict__ g_logits, int vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red64[8];
    __shared__ float dbuf[256];
    __shared__ float red[16];
    __shared__ int rcnt[4];
    __shared__ int fin_flag;
    const int h = a.h, w = a.w;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.N) {   // workgroup-uniform
        BXI_T(2, blockIdx.x, 0);
        leader_block(a, dil, ws, st, (int)blockIdx.x, smem, red);
        BXI_T(2, blockIdx.x, 3);
        arrive_and_finish(ws, st, (int)blockIdx.x, a.N, warmup, losses, &fin_flag, red64, dbuf);
        BXI_T(2, blockIdx.x, 4);
        return;
    }
    const int nwork = *ws.nwork;
    const int ntile_wg = (int)gridDim.x - a.N;
    BXI_T(1, blockIdx.x, 0);
    // work list built by stage1: the box tiles of all instances, compacted.  The tile workgroups of this
    // launch take item blockIdx - N (+ a multiple of the tile-workgroup count when there are more items: many instances).
    // The record of the NEXT item is requested before the current tile is worked on, so that a looping workgroup does
    // not pay its ~1.5 us again per tile.
    const int cap = a.N * ((h + kBR - 1) / kBR) * ((w + kBC - 1) / kBC);
    const int64_t P = (int64_t)h * w;
    const int d = dil;
    const int PAD = (d + 3) & ~3;
    const int PR = kBR + 2 * d, PC = kBC + 2 * PAD;          // staged region: tile + halo (columns padded to 4)
    const int q4 = PC / 4, items = PR * q4;                  // d = 2: 216 float4 per plane
    const bool one_item = PREFETCH && items <= 256;          // d <= 3: a thread stages at most one float4 per plane
    // the raw logits of a work item's tile + halo, one float4 per thread (zero padding of F.unfold)
    auto load_item = [&](const WorkRec& w_, int i, float4& t) {
        t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= items) return;
        const int lr = i / q4, r = w_.tile_r0 - d + lr, c = w_.tile_c0 - PAD + (i % q4) * 4;
        if (!(r >= 0 && r < h && c >= 0 && c < w)) return;
        t = load4(a.logits + (int64_t)w_.n * P + (int64_t)r * w, c, w, vec);
    };
    WorkRec wr_next = ws.work[min((int)blockIdx.x - a.N, cap - 1)];   // speculative (inside the list's capacity) ...
    float4 pre;                                              // the next tile's data, in flight while the current tile is worked on
    if ((int)blockIdx.x - a.N < nwork && one_item) load_item(wr_next, tid, pre);
    for (int wi = (int)blockIdx.x - a.N; wi < cap; wi += ntile_wg) {
        if (wi >= nwork) break;                  // workgroup-uniform
        const WorkRec wr = wr_next;
        const bool more = wi + ntile_wg < nwork;
        if (more) wr_next = ws.work[wi + ntile_wg];
        const int n = wr.n, r0 = wr.tile_r0, c0 = wr.tile_c0;
        InstRec rc; rc.r0 = wr.r0; rc.r1 = wr.r1; rc.c0 = wr.c0; rc.c1 = wr.c1; rc.img = wr.img;
        const InstBox ib = inst_from_rec(rc, dil, h, w);
        BXI_T(1, blockIdx.x, 1);
        const float* L = a.logits + (int64_t)n * P;
        float2* pq = reinterpret_cast<float2*>(smem);
        uint8_t* bits = smem + sizeof(float2) * (size_t)PR * PC;                            // [PR][PC] affinity words of the in-box pixels

        // ---- 1. the logits region (-> sigmoid pairs) and the affinity words go to LDS ------------------------------
        {
            for (int i = tid; i < items; i += 256) {
                float4 t;
                if (one_item) t = pre;                               // requested during the previous tile
                else load_item(wr, i, t);
                const int lr = i / q4;
                float2* dst = pq + (size_t)lr * PC + (i % q4) * 4;
                dst[0] = sig_pair(t.x); dst[1] = sig_pair(t.y); dst[2] = sig_pair(t.z); dst[3] = sig_pair(t.w);
            }
            if (one_item && more) load_item(wr_next, tid, pre);     // flies during the pair loop of this tile
            const uint8_t* AF = bits_in + (int64_t)ib.img * P;       // those of the in-box pixels (bitmask == 1, :1324-1325)
            for (int i = tid; i < PR * PC; i += 256) {
                const int r = r0 - d + i / PC, c = c0 - PAD + i % PC;
                const bool inbox = r >= ib.box.r0 && r < ib.box.r1 && c >= ib.box.c0 && c < ib.box.c1;
                bits[i] = inbox ? AF[(int64_t)r * w + c] : (uint8_t)0;
            }
        }
        __syncthreads();
        BXI_T(1, blockIdx.x, 2);

        // ---- 2. pairwise term, 2 pixels per thread ---------------------------------------------------------------------
        // weight of the pair (p, q = p + delta_k):  W[k,p] + W[7-k,q], both from the staged affinity words
        // thread -> row lr, columns lcx and lcx + 32: the 32 lanes of a row read consecutive LDS words
        // (conflict-free ds_read_b32 / b64), unlike an adjacent-pixel pairing (2-way conflicts)
        const int lr = tid >> 5, lcx = tid & 31;
        const int r = r0 + lr;
        float num = 0.f;
        int cnt = 0;
        float out[2] = {0.f, 0.f};
        uint32_t rin[3];                                      // the three neighbour rows: inside the map?
    #pragma unroll
        for (int dy = -1; dy <= 1; ++dy) { const int r2 = r + dy * d; rin[dy + 1] = (r2 >= 0 && r2 < h) ? 1u : 0u; }
    #pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + lcx + 32 * e;
            if (r < ib.dil.r0 || r >= ib.dil.r1 || c < ib.dil.c0 || c >= ib.dil.c1) continue;
            uint32_t cin[3];
    #pragma unroll
            for (int dx = -1; dx <= 1; ++dx) { const int c2 = c + dx * d; cin[dx + 1] = (c2 >= 0 && c2 < w) ? 1u : 0u; }
            const int pi = (lr + d) * PC + (lcx + 32 * e + PAD);
            const float2 pp = pq[pi];
            const uint32_t bits_p = bits[pi];
            float2 nq[8]; uint32_t nbw[8];
            {
                int k = 0;
    #pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
    #pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (dx == 0 && dy == 0) continue;
                        const int qi = pi + dy * d * PC + dx * d;
                        nq[k] = pq[qi];
                        nbw[k] = bits[qi];
                        ++k;
                    }
            }
            float acc = 0.f;
            bool tiny = false;
            uint32_t wps = 0, wqs = 0;     // bit k: W[k,p] / W[7-k,q] (kept for the rare log-space redo)
            const float ppq = pp.x * pp.y;
            int k = 0;
    #pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
    #pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dx == 0 && dy == 0) continue;
                    const bool inb = (rin[dy + 1] & cin[dx + 1]) != 0u;
                    const uint32_t wp = (bits_p >> k) & 1u;
                    const uint32_t wq = inb ? (nbw[k] >> (7 - k)) & 1u : 0u;
                    cnt += (int)wp;                                          // weights.sum(), :1328 (counts padded pairs too)
                    wps |= wp << k; wqs |= wq << k;
                    const float fw = inb ? (float)(wp + wq) : 0.f, fp = inb ? (float)wp : 0.f;
                    const float S = pp.x * nq[k].x + pp.y * nq[k].y;        // P(y_p == y_q)
                    tiny |= (fw != 0.f) && !(S > 1e-30f);
                    const float Sc = fmaxf(S, 1e-30f);
                    num += fp * -__logf(Sc);
                    acc += fw * (-(nq[k].x - nq[k].y) * ppq * __builtin_amdgcn_rcpf(Sc));
                    ++k;
                }
            if (tiny) {   // rare
                const float2 fix = pair_logspace_redo(L, pq, h, w, d, PC, r, c, pi, pp, wps, wqs);
                num += fix.x; acc = fix.y;
            }
            out[e] = acc;
        }
        BXI_T(1, blockIdx.x, 4);
        // ---- per-instance accumulators: integers, so the result does not depend on the arrival order --------
        num = wave_sum_f32(num);
        cnt = wave_sum_i32(cnt);
        if ((tid & 63) == 0) { red[tid >> 6] = num; rcnt[tid >> 6] = cnt; }
        __syncthreads();
        if (tid == 0) {
            const float bn = (red[0] + red[1]) + (red[2] + red[3]);
            const int bc = rcnt[0] + rcnt[1] + rcnt[2] + rcnt[3];
            if (bc) atomicAdd(&ws.acc[2 * n], (unsigned long long)bc);
            if (bn != 0.f) atomicAdd(&ws.acc[2 * n + 1], (unsigned long long)(long long)(bn * kNumScale));
        }
        BXI_T(1, blockIdx.x, 5);
        arrive_and_finish(ws, st, n, a.N, warmup, losses, &fin_flag, red64, dbuf);
        BXI_T(1, blockIdx.x, 6);
        // gradient tile last: nothing in this launch waits for these stores (the arrival above only
        // covers the accumulators), they drain while the wave retires
        if (g_logits && r < h) {
            float* G = g_logits + (int64_t)n * P + (int64_t)r * w + c0 + lcx;   // 2 x 128 B contiguous per half-wave
            if (c0 + lcx < w) G[0] = out[0];
            if (c0 + lcx + 32 < w) G[32] = out[1];
        }
        __syncthreads();                     // LDS is reused by the next work item
    }
}

// Two builds of the same body.  The default one (3 waves per SIMD, no spills) serves launches in which a tile workgroup
// handles one tile: there the per-tile chain is the launch.  With hundreds of instances every workgroup loops over many tiles
// and the launch is throughput bound: 4 waves per SIMD hide more of each chain.  The default build instead prefetches the
// next tile's raw data during the pair loop, worth 4-5 % when it loops.
__global__ __launch_bounds__(256) void box_kernel(InstArgs a, const uint8_t* __restrict__ bits_in, int dil, float warmup, LossWs ws,
                                                  LossState st, float* __restrict__ losses, float* __restrict__ g_logits, int vec) {
    box_body<true>(a, bits_in, dil, warmup, ws, st, losses, g_logits, vec);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void box_kernel_dense(InstArgs a, const uint8_t* __restrict__ bits_in, int dil, float warmup, LossWs ws, LossState st,
                      float* __restrict__ losses, float* __restrict__ g_logits, int vec) {
    box_body<false>(a, bits_in, dil, warmup, ws, st, losses, g_logits, vec);
}

// ================================================================================================
// Kernel 3 (backward): loss_apply -- normalise the pairwise gradient, add the projection gradient
// ================================================================================================
// grid = kSlices x N.  g_logits holds zeros + the un-normalised pairwise gradient on the box tiles.
//   dense pass over the box tiles:  G <- g_pw * (warmup / max(sum W,1)) * G + g_prj * prj(r,c)
//   sparse pass elsewhere        :  G <- g_prj * prj(r,c) at the h+w arg-max positions
// g_prj / g_pw are read from device memory (no host sync).
__global__ __launch_bounds__(256) void loss_apply_kernel(InstArgs a, int dil, LossState st,
                                                         const float* __restrict__ up_prj,
                                                         const float* __restrict__ up_pw,
                                                         float* __restrict__ g_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y, s = blockIdx.x;
    const int h = a.h, w = a.w, tid = threadIdx.x;
    const int64_t P = (int64_t)h * w;
    float* gcol = reinterpret_cast<float*>(smem);
    float* grow = gcol + w;
    int* carg = reinterpret_cast<int*>(grow + h);
    int* rarg = carg + w;
    float* G = g_logits + (int64_t)n * P;
    // ---- every global load of the first batch is issued before the first use ---------------------------
    const InstRec rec = st.inst[n];
    const float gp = *up_prj, gw = *up_pw, scale = *st.scale;
    const int ca0 = tid < w ? st.colarg[(int64_t)n * w + tid] : 0, ra0 = tid < h ? st.rowarg[(int64_t)n * h + tid] : 0;
    const float gc0 = tid < w ? st.gcol[(int64_t)n * w + tid] : 0.f, gr0 = tid < h ? st.grow[(int64_t)n * h + tid] : 0.f;
    const InstBox ib = inst_from_rec(rec, dil, h, w);
    // the box tiles of box_kernel: tile-aligned hull of the dilated box
    const int tr0 = ib.dil.r0 & ~(kBR - 1), tr1 = min(h, (ib.dil.r1 + kBR - 1) & ~(kBR - 1));
    const int tc0 = ib.dil.c0 & ~(kBC - 1), tc1 = min(w, (ib.dil.c1 + kBC - 1) & ~(kBC - 1));
    const int rows = ib.any ? tr1 - tr0 : 0;
    const int per = (rows + gridDim.x - 1) / gridDim.x;
    const int ra = tr0 + s * per, rb = min(tr1, ra + per);
    const int cw = tc1 - tc0;
    const int npx = ib.any && rb > ra ? (rb - ra) * cw : 0;
    float gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = tid + u * 256;
        gv[u] = i < npx ? G[(int64_t)(ra + i / cw) * w + tc0 + i % cw] : 0.f;
    }
    if (tid < w) { carg[tid] = ca0; gcol[tid] = gc0; }
    if (tid < h) { rarg[tid] = ra0; grow[tid] = gr0; }
    for (int c = tid + 256; c < w; c += 256) { carg[c] = st.colarg[(int64_t)n * w + c]; gcol[c] = st.gcol[(int64_t)n * w + c]; }
    for (int r = tid + 256; r < h; r += 256) { rarg[r] = st.rowarg[(int64_t)n * h + r]; grow[r] = st.grow[(int64_t)n * h + r]; }
    const float dense_scale = gw * scale;
    __syncthreads();
    for (int base = tid; base < npx; base += 256 * 8) {
        if (base != tid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256;
                gv[u] = i < npx ? G[(int64_t)(ra + i / cw) * w + tc0 + i % cw] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256;
            if (i < npx) {
                const int r = ra + i / cw, c = tc0 + i % cw;
                float sp = 0.f;
                if (carg[c] == r) sp += gcol[c];
                if (rarg[r] == c) sp += grow[r];
                G[(int64_t)r * w + c] = gv[u] * dense_scale + sp * gp;
            }
        }
    }
    if (s == 0) {   // arg-max positions outside the box tiles (the rest of the map stays zero)
        for (int c = tid; c < w; c += 256) {
            const int r = carg[c];
            const bool in_t = ib.any && r >= tr0 && r < tr1 && c >= tc0 && c < tc1;
            if (!in_t) {
                float v = gcol[c];
                if (rarg[r] == c) v += grow[r];
                G[(int64_t)r * w + c] = v * gp;
            }
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg[r];
            const bool in_t = ib.any && r >= tr0 && r < tr1 && c >= tc0 && c < tc1;
            if (!in_t && carg[c] != r) G[(int64_t)r * w + c] = grow[r] * gp;
        }
    }
}

__global__ void zero_losses_kernel(float* losses) { losses[0] = 0.f; losses[1] = 0.f; }

// ---- host side ---------------------------------------------------------------------------------
int fill_inst(const bxi_instances* in, InstArgs& a) {
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

static size_t box_lds_bytes(int dil) {
    const size_t PAD = (dil + 3) & ~3;
    const size_t PR = kBR + 2 * dil, PC = kBC + 2 * PAD;
    return sizeof(float2) * PR * PC + PR * PC;
}

// One evaluation from precomputed affinity bits: stage1 (tables + logit streaming) -> box (leaders + box tiles);
// bxi_boxinst_loss_backward_f32 (loss_apply) finishes the gradient.
int launch_loss(const bxi_instances* in, const uint8_t* affinity, int size, int dil, float warmup, float* losses, float* g_logits,
                void* state, void* workspace, size_t workspace_bytes, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || dil > kMaxDil) return BXI_ERR_UNSUPPORTED;
    if (!losses) return BXI_ERR_NULL_POINTER;
    hipStream_t s = as_stream(stream);
    if (a.N == 0) {
        BXI_LAUNCH("zero_losses", s, zero_losses_kernel, dim3(1), dim3(1), 0, s, losses);
        return check_launch();
    }
    if (!affinity) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    const size_t need = carve_ws(nullptr, a.N, a.h, a.w, nullptr);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    LossWs ws;
    carve_ws(workspace, a.N, a.h, a.w, &ws);
    LossState st = {};
    if (state) {
        if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
        carve_state(state, a.N, a.h, a.w, &st);
    }
    const int vec = ((a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0 &&
                     (!g_logits || (reinterpret_cast<uintptr_t>(g_logits) & 15) == 0)) ? 1 : 0;

    // ---- kernel 1: table waves + logit streaming waves ----------------------------------------------------
    const int n_stream = a.N * stream_tiles(a.h);
    BXI_LAUNCH("stage1", s, stage1_kernel, dim3((unsigned)(a.N + n_stream)), dim3(64), 0, s, a, dil, ws, g_logits, vec);
    rc = check_launch();
    if (rc != BXI_OK) return rc;

    // ---- kernel 2: N leader workgroups (projection term) + the box tiles (pairwise term) ------------------
    size_t lds = box_lds_bytes(dil);
    const size_t lds_leader = sizeof(float) * (size_t)(a.h + a.w);
    if (lds < lds_leader) lds = lds_leader;
    if (lds > 160 * 1024) return BXI_ERR_UNSUPPORTED;
    const int n_tiles = a.N * box_tiles(a.h, a.w);
    const bool dense = n_tiles > 6400;         // more than ~2 tiles per tile workgroup (see box_kernel_dense)
    if (lds > 64 * 1024) {
        const void* fn = dense ? reinterpret_cast<const void*>(box_kernel_dense) : reinterpret_cast<const void*>(box_kernel);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    const int n_box = a.N + (n_tiles < 1024 ? n_tiles : 1024);   // list length is device data: stride through it
    if (dense) BXI_LAUNCH("box", s, box_kernel_dense, dim3((unsigned)n_box), dim3(256), lds, s, a, affinity, dil, warmup, ws, st, losses, g_logits, vec);
    else BXI_LAUNCH("box", s, box_kernel, dim3((unsigned)n_box), dim3(256), lds, s, a, affinity, dil, warmup, ws, st, losses, g_logits, vec);
    return check_launch();
}

int launch_backward(const bxi_instances* in, const float* g_prj, const float* g_pw, int dil, const void* state,
                    float* g_logits, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (dil < 1 || dil > kMaxDil) return BXI_ERR_BAD_ARGUMENT;
    if (a.N == 0) return BXI_OK;
    if (!g_prj || !g_pw || !state || !g_logits) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
    LossState st;
    carve_state(const_cast<void*>(state), a.N, a.h, a.w, &st);
    const size_t lds_d = (sizeof(float) + sizeof(int)) * (size_t)(a.h + a.w);
    BXI_LAUNCH("loss_apply", as_stream(stream), loss_apply_kernel, dim3(kSlices, a.N), dim3(256), lds_d,
               as_stream(stream), a, dil, st, g_prj, g_pw, g_logits);
    return check_launch();
}

size_t loss_ws_bytes(int N, int h, int w) { return carve_ws(nullptr, N, h, w, nullptr); }
size_t loss_state_bytes(int N, int h, int w) { return carve_state(nullptr, N, h, w, nullptr); }

}  // namespace bxi

#ifdef BXI_TRACE
extern "C" int bxi_debug_set_trace(void* buf) {   // developer builds only
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_trace), &buf, sizeof(buf));
}
#endif

extern "C" {

size_t bxi_boxinst_loss_workspace_bytes(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    return bxi::loss_ws_bytes(N, h, w);
}
size_t bxi_boxinst_loss_state_bytes(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    return bxi::loss_state_bytes(N, h, w);
}
size_t bxi_boxinst_loss_state_status_offset(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    bxi::LossState st;
    char base[1];
    bxi::carve_state(base, N, h, w, &st);
    return (size_t)((char*)st.status - base);
}
size_t bxi_boxinst_loss_state_warmup_offset(int N, int h, int w) {
    if (N < 0 || h <= 0 || w <= 0) return 0;
    bxi::LossState st;
    char base[1];
    bxi::carve_state(base, N, h, w, &st);
    return (size_t)((char*)(st.scale + 1) - base);
}

int bxi_boxinst_loss_fwd_bwd_f32(const bxi_instances* inst_host, const uint8_t* affinity, int size, int dilation,
                                 float warmup, float* losses, float* g_logits, void* state, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return bxi::launch_loss(inst_host, affinity, size, dilation, warmup, losses, g_logits, state, workspace, workspace_bytes, stream);
}

int bxi_boxinst_loss_backward_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw, int dilation,
                                  const void* state, float* g_logits, void* stream) {
    return bxi::launch_backward(inst_host, g_prj, g_pw, dilation, state, g_logits, stream);
}

}  // extern "C"
