// tree_filter.hip -- SURVEY 8(f-4): the reference's `tree_filter` extension on gfx950
// (mmdet/ops/tree_filter: src/mst/{mst.cu,boruvka.cpp}, src/bfs/bfs.cu, src/refine/refine.cu).
//
//   mst     minimum spanning tree of the pixel graph.  The reference copies the graph to the host and runs a
//           sequential Boruvka with union-find, one std::thread per image (mst.cu:93-118).  Here: one workgroup per
//           graph, parallel Boruvka entirely in LDS -- per round every edge offers (weight bits, edge index) to the two
//           components it joins with a 64-bit LDS atomic min, components hook onto their cheapest neighbour, pointer
//           jumping flattens the hooks.  Under the total order (weight, index) the minimum spanning tree is unique and
//           is the one the reference's strict '>' comparisons select, so the edge SET is identical (tests compare with
//           the reference's own boruvka.cpp); it is returned in ascending edge order.
//   bfs     breadth-first order from vertex 0.  The reference appends children with atomicAdd from 64 threads
//           (order depends on arrival, bfs.cu:72); here positions come from wave prefix sums: deterministic, and the
//           children of a node are CONTIGUOUS, which the refine kernels use (first child + count per node).
//   refine  the tree filter  out_i = sum_j S(i,j) x_j / sum_j S(i,j),  S = product of edge weights on the tree path,
//           computed as the reference does by a leaf->root aggregation and a root->leaf propagation (refine.cu:17-121),
//           and its two gradients (:123-184, :235-370).  The recurrences are level-sequential (tree depth ~ hundreds of
//           levels of ~20 nodes), so everything a traversal touches (values, weights, child ranges, level offsets) is
//           staged in LDS first and one wave walks the levels without workgroup barriers.
#include "common.hpp"

namespace bxi {

typedef unsigned long long u64;
typedef unsigned short u16;

// One wave walks a tree level by level through LDS: its LDS operations execute in program order, so between two
// levels only the compiler has to be held back and the outstanding LDS operations waited for -- not the global stores
// of the level (a workgroup-scope fence waits for those too: ~1 us per level, hundreds of levels).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kTfMaxV = 10240;     // LDS-resident vertex limit (96x96 = 9216 in the reference's _scale_target)

// ---------------------------------------------------------------------------------------------------
// mst
__global__ __launch_bounds__(1024) void mst_kernel(const int* __restrict__ edge_index, const float* __restrict__ edge_weight, int E,
                                                   int V, int* __restrict__ edge_out, int* __restrict__ n_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mst_raw[];
    u64* best = reinterpret_cast<u64*>(mst_raw);                          // [V]
    u16* comp = reinterpret_cast<u16*>(best + V);                         // [V] component (= root vertex) of a vertex
    u16* link = comp + V;                                                 // [V] hook of a root
    uint32_t* chosen = reinterpret_cast<uint32_t*>(link + V);             // [ceil(E/32)] bitmap of tree edges (8V + 4V bytes in: aligned)
    __shared__ int flag, scan[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* idx = edge_index + (int64_t)b * E * 2;
    const float* wt = edge_weight + (int64_t)b * E;
    const int nwords = (E + 31) / 32;
    for (int v = tid; v < V; v += 1024) comp[v] = (u16)v;
    for (int i = tid; i < nwords; i += 1024) chosen[i] = 0u;
    __syncthreads();
    for (int round = 0; round < 32; ++round) {
        for (int v = tid; v < V; v += 1024) best[v] = ~0ull;
        if (tid == 0) flag = 0;
        __syncthreads();
        for (int e = tid; e < E; e += 1024) {
            const int cu = comp[idx[2 * e]], cv = comp[idx[2 * e + 1]];
            if (cu != cv) {
                const u64 key = ((u64)__float_as_uint(wt[e]) << 32) | (uint32_t)e;   // weights are >= 0: the bits order like the values
                atomicMin(&best[cu], key);
                atomicMin(&best[cv], key);
            }
        }
        __syncthreads();
        for (int c = tid; c < V; c += 1024) {
            if (comp[c] != c) continue;
            const u64 k = best[c];
            int to = c;
            if (k != ~0ull) {
                const uint32_t e = (uint32_t)k;
                atomicOr(&chosen[e >> 5], 1u << (e & 31));
                const int cu = comp[idx[2 * e]], cv = comp[idx[2 * e + 1]];
                to = cu == c ? cv : cu;
                flag = 1;
            }
            link[c] = (u16)to;
        }
        __syncthreads();
        if (!flag) break;                                                 // one component left (or a disconnected graph)
        for (int c = tid; c < V; c += 1024) {                             // two components that chose each other: the smaller id is the root
            if (comp[c] != c) continue;
            const int o = link[c];
            if (o != c && link[o] == c && c < o) link[c] = (u16)c;
        }
        __syncthreads();
        for (int guard = 0; guard < 32; ++guard) {                        // pointer jumping
            if (tid == 0) flag = 0;
            __syncthreads();
            for (int c = tid; c < V; c += 1024) {
                if (comp[c] != c) continue;
                const int p = link[c], gp = link[p];
                if (gp != p) { link[c] = (u16)gp; flag = 1; }
            }
            __syncthreads();
            if (!flag) break;
            __syncthreads();
        }
        for (int v = tid; v < V; v += 1024) comp[v] = link[comp[v]];
        __syncthreads();
    }
    // tree edges in ascending edge order: each thread owns a contiguous run of bitmap words
    const int per = (nwords + 1023) / 1024;
    const int w0 = min(tid * per, nwords), w1 = min(w0 + per, nwords);
    int cnt = 0;
    for (int i = w0; i < w1; ++i) cnt += __popc(chosen[i]);
    const int lane = tid & 63, wave = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    if (tid == 0) { int s = 0; for (int i = 0; i < 16; ++i) { const int t = scan[i]; scan[i] = s; s += t; } scan[16] = s; }
    __syncthreads();
    int pos = scan[wave] + incl - cnt;
    int* out = edge_out + (int64_t)b * (V - 1) * 2;
    for (int i = w0; i < w1; ++i) {
        uint32_t m = chosen[i];
        while (m) {
            const int e = i * 32 + __ffs((int)m) - 1;
            m &= m - 1;
            if (pos < V - 1) { out[2 * pos] = idx[2 * e]; out[2 * pos + 1] = idx[2 * e + 1]; }
            ++pos;
        }
    }
    if (tid == 0) n_out[b] = scan[16];
}

// ---------------------------------------------------------------------------------------------------
// bfs: levels[b] = { D, off_0 = 0, off_1, ..., off_D = V }
__global__ __launch_bounds__(256) void bfs_kernel(const int* __restrict__ tree, int V, int max_adj, int* __restrict__ sorted_index,
                                                  int* __restrict__ sorted_parent, int* __restrict__ sorted_child,
                                                  int* __restrict__ levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bfs_raw[];
    u16* adj = reinterpret_cast<u16*>(bfs_raw);        // [V][4]
    u16* si = adj + (size_t)V * 4;                     // [V] vertex at a position
    u16* pv = si + V;                                  // [V] parent vertex of the vertex at a position
    unsigned char* deg = reinterpret_cast<unsigned char*>(pv + V);   // [V]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* ed = tree + (int64_t)b * (V - 1) * 2;
    int* s_index = sorted_index + (int64_t)b * V;
    int* s_parent = sorted_parent + (int64_t)b * V;
    int* s_child = sorted_child + (int64_t)b * V * max_adj;
    int* lv = levels + (int64_t)b * (V + 2);
    unsigned int* deg32 = reinterpret_cast<unsigned int*>(deg);
    for (int i = tid; i < (V + 3) / 4; i += 256) deg32[i] = 0u;
    for (int i = tid; i < V * max_adj; i += 256) s_child[i] = 0;
    __syncthreads();
    // adjacency (degree <= 4): slots by byte-wise LDS atomics on the packed degree words
    for (int e = tid; e < V - 1; e += 256) {
        const int u = ed[2 * e], v = ed[2 * e + 1];
        const unsigned su = (atomicAdd(&deg32[u >> 2], 1u << (8 * (u & 3))) >> (8 * (u & 3))) & 0xffu;
        const unsigned sv = (atomicAdd(&deg32[v >> 2], 1u << (8 * (v & 3))) >> (8 * (v & 3))) & 0xffu;
        if (su < 4) adj[u * 4 + su] = (u16)v;
        if (sv < 4) adj[v * 4 + sv] = (u16)u;
    }
    __syncthreads();
    for (int v = tid; v < V; v += 256) {                // arrival order of the atomics -> ascending neighbour ids
        const int d = min((int)deg[v], 4);
        u16 a[4];
        for (int k = 0; k < 4; ++k) a[k] = k < d ? adj[v * 4 + k] : (u16)0xffff;
        for (int i = 1; i < 4; ++i) for (int j = i; j > 0 && a[j - 1] > a[j]; --j) { const u16 t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; }
        for (int k = 0; k < 4; ++k) adj[v * 4 + k] = a[k];
    }
    __syncthreads();
    if (tid >= 64) return;                              // one wave walks the levels (no workgroup barrier inside)
    const int lane = tid;
    if (lane == 0) { si[0] = 0; pv[0] = 0xffff; s_index[0] = 0; s_parent[0] = 0; lv[1] = 0; }
    int lo = 0, hi = 1, n = 1, depth = 0;
    while (lo < hi) {
        for (int base = lo; base < hi; base += 64) {
            const int i = base + lane;
            int cur = 0, par = 0xffff, nch = 0;
            u16 ch[4] = {0, 0, 0, 0};
            if (i < hi) {
                cur = si[i]; par = pv[i];
                const u64 a4 = *reinterpret_cast<const u64*>(adj + cur * 4);        // the 4 neighbour slots in one read
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u16 a = (u16)(a4 >> (16 * k));
                    if (a != 0xffff && a != par) {                                   // slot k -> child number nch (static indexing)
                        if (nch == 0) ch[0] = a; else if (nch == 1) ch[1] = a; else if (nch == 2) ch[2] = a; else ch[3] = a;
                        ++nch;
                    }
                }
            }
            // exclusive prefix sum of nch (0..4) over the wave from three ballots: no LDS traffic
            const u64 b0 = __ballot(nch & 1), b1 = __ballot(nch & 2), b2 = __ballot(nch & 4);
            const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
            const int excl = __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below);
            const int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
            int pos = n + excl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nch) {
                    si[pos + k] = ch[k]; pv[pos + k] = (u16)cur;
                    s_index[pos + k] = ch[k]; s_parent[pos + k] = i;
                    if (k < max_adj) s_child[i * max_adj + k] = pos + k;
                }
            }
            n += total;
            wave_lds_fence();                                          // LDS writes of this chunk before the next reads
        }
        ++depth;
        if (lane == 0) lv[1 + depth] = hi;
        lo = hi; hi = n;
    }
    if (lane == 0) lv[0] = depth;
}

// ---------------------------------------------------------------------------------------------------
// refine: one workgroup per (tree, channel); a traversal = leaf->root aggregation then root->leaf propagation
struct TreeLds {
    float* val;        // [V] values in sorted order (x -> U -> D, in place)
    float* w;          // [V] edge weight to the parent
    uint32_t* fc;      // [V] first child position | child count << 16 ... packed: pos (low 16 bits), count (bits 16..18)
    u16* lv;           // [D+1] level offsets
    int D;
};

// U_i = x_i + sum_c w_c U_c (refine.cu:64-121), then D_0 = U_0, D_c = U_c (1 - w_c^2) + D_parent w_c (:17-62), in place.
// Called by wave 0 only; `u_out` (sorted order, may be null) receives U before it is overwritten.
__device__ __forceinline__ void tree_updown(const TreeLds& t, int V, int lane, float* __restrict__ u_out) {
    // A level costs dependent LDS round trips, and there are ~1000 levels: the (up to 4, contiguous) children of a node
    // are read unconditionally at clamped positions so that they travel together, and the bounds of the next level are
    // read while this one is processed.
    int lo = t.lv[t.D - 1], hi = t.lv[t.D];
    for (int l = t.D - 1; l >= 0; --l) {
        const int nlo = l ? t.lv[l - 1] : 0;
        for (int i = lo + lane; i < hi; i += 64) {
            const uint32_t f = t.fc[i];
            const int c0 = f & 0xffffu, nc = f >> 16;
            float v[4], w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = min(c0 + k, V - 1); v[k] = t.val[c]; w[k] = t.w[c]; }
            float acc = t.val[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += k < nc ? v[k] * w[k] : 0.f;
            t.val[i] = acc;
            if (u_out) u_out[i] = acc;
        }
        wave_lds_fence();
        hi = lo; lo = nlo;
    }
    lo = 0; hi = t.lv[1];
    for (int l = 0; l < t.D; ++l) {
        const int nhi = l + 2 <= t.D ? t.lv[l + 2] : hi;
        for (int i = lo + lane; i < hi; i += 64) {
            const uint32_t f = t.fc[i];
            const int c0 = f & 0xffffu, nc = f >> 16;
            const float dp = t.val[i];
            float v[4], w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = min(c0 + k, V - 1); v[k] = t.val[c]; w[k] = t.w[c]; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nc) t.val[c0 + k] = v[k] * (1.f - w[k] * w[k]) + dp * w[k];
        }
        wave_lds_fence();
        lo = hi; hi = nhi;
    }
}

struct RefineArgs {
    const float* in;            // [B,C,V] vertex order
    const float* pre_div;       // [B,V] vertex order or null: in / pre_div      (grad_out / weight_sum, refine.cu:251)
    const float* pre_mul;       // [B,C,V] vertex order or null: ... * pre_mul   (feature_grad = grad_out_norm * feature_out, :324)
    const float* edge_weight;   // [B,V] sorted order
    const int* sorted_index;    // [B,V]
    const int* sorted_child;    // [B,V,max_adj]
    const int* levels;          // [B,V+2]
    float* up_sorted;           // [B,C,V] or null : U
    float* down_sorted;         // [B,C,V] or null : D in sorted order
    float* down_vertex;         // [B,C,V] or null : D in vertex order (feature_aggr / grad_feature)
    float* out_vertex;          // [B,C,V] or null : D / weight_sum in vertex order (feature_out)
    float* wsum_up_sorted;      // [B,V] or null   : the same traversal of ones (weight_sum_up)
    float* wsum_vertex;         // [B,V]           : weight_sum (written when wsum_up_sorted != null, read for out_vertex)
    int B, C, V, max_adj, with_ones;
};

__global__ __launch_bounds__(256) void tree_refine_kernel(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_raw[];
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, V = a.V;
    TreeLds t;
    t.val = reinterpret_cast<float*>(tf_raw);
    t.w = t.val + V;
    t.fc = reinterpret_cast<uint32_t*>(t.w + V);
    t.lv = reinterpret_cast<u16*>(t.fc + V);
    const int* lv = a.levels + (int64_t)b * (V + 2);
    t.D = lv[0];
    const int* si = a.sorted_index + (int64_t)b * V;
    const int* sc = a.sorted_child + (int64_t)b * V * a.max_adj;
    const float* ew = a.edge_weight + (int64_t)b * V;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int i = tid; i <= t.D; i += 256) t.lv[i] = (u16)lv[1 + i];
    for (int i = tid; i < V; i += 256) {
        t.w[i] = i ? ew[i] : 0.f;                                   // weight[0] = 0 (refine.cu:38)
        int c0 = 0, nc = 0;
        for (int k = 0; k < a.max_adj; ++k) {
            const int c = sc[i * a.max_adj + k];
            if (c <= 0) break;
            if (nc == 0) c0 = c; else if (c != c0 + nc) bad = 1;     // children must be contiguous (bxi_bfs_forward_i32 order)
            ++nc;
        }
        t.fc[i] = (uint32_t)c0 | ((uint32_t)nc << 16);
    }
    __syncthreads();
    const float poison = bad ? __builtin_nanf("") : 1.f;              // a foreign ordering fails loudly in the values
    // ---- the traversal of ones: weight_sum_up / weight_sum (refine.cu:223-228) --------------------------------------
    if (a.with_ones) {
        for (int i = tid; i < V; i += 256) t.val[i] = poison;
        __syncthreads();
        float* wu = (ch == 0 && a.wsum_up_sorted) ? a.wsum_up_sorted + (int64_t)b * V : nullptr;
        if (tid < 64) tree_updown(t, V, tid, wu);
        __syncthreads();
        if (ch == 0 && a.wsum_vertex)
            for (int i = tid; i < V; i += 256) a.wsum_vertex[(int64_t)b * V + si[i]] = t.val[i];
        __syncthreads();
    }
    // ---- the channel ------------------------------------------------------------------------------------------------
    const int64_t cb = ((int64_t)b * a.C + ch) * V;
    // weight_sum of this tree in sorted order is in t.val right now (if with_ones): the division needs it per node, so the
    // channel values are staged in registers first
    constexpr int kPer = (kTfMaxV + 255) / 256;
    float ws[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) { const int i = tid + j * 256; ws[j] = (a.with_ones && i < V) ? t.val[i] : 1.f; }
    __syncthreads();
    for (int i = tid; i < V; i += 256) {
        const int p = si[i];
        float x = a.in[cb + p];
        if (a.pre_div) x /= a.pre_div[(int64_t)b * V + p];
        if (a.pre_mul) x *= a.pre_mul[cb + p];
        t.val[i] = x * poison;
    }
    __syncthreads();
    if (tid < 64) tree_updown(t, V, tid, a.up_sorted ? a.up_sorted + cb : nullptr);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = tid + j * 256;
        if (i < V) {
            const float d = t.val[i];
            const int p = si[i];
            if (a.down_sorted) a.down_sorted[cb + i] = d;
            if (a.down_vertex) a.down_vertex[cb + p] = d;
            if (a.out_vertex) a.out_vertex[cb + p] = d / ws[j];
        }
    }
}

// d loss / d edge_weight (refine.cu:284-370 with root_leaf_grad_kernel :123-184), sorted order, [0] = 0.
// With GU = Up(g~), OG = Down(GU), FGU = Up(g~ out), OG2 = Down(FGU) (sorted order) the two root->leaf gradient sweeps
// reduce, node by node, to
//   grad_w[c] = sum_ch ( GU_c D_p + U_c OG_p - 2 w_c U_c GU_c ) - sum_ch ( FGU_c WD_p + WU_c OG2_p - 2 w_c WU_c FGU_c ),  p = parent(c)
__global__ __launch_bounds__(256) void tree_grad_weight_kernel(const float* __restrict__ U, const float* __restrict__ Dv /*vertex order*/,
                                                               const float* __restrict__ WU, const float* __restrict__ WDv /*vertex order*/,
                                                               const float* __restrict__ GU, const float* __restrict__ OG,
                                                               const float* __restrict__ FGU, const float* __restrict__ OG2,
                                                               const float* __restrict__ edge_weight, const int* __restrict__ sorted_index,
                                                               const int* __restrict__ sorted_parent, int B, int C, int V,
                                                               float* __restrict__ grad_w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * V) return;
    const int b = (int)(i / V), c = (int)(i % V);
    if (c == 0) { grad_w[i] = 0.f; return; }
    const int p = sorted_parent[i];
    const int pp = sorted_index[(int64_t)b * V + p];
    const float w = edge_weight[i];
    const float wu = WU[i], wdp = WDv[(int64_t)b * V + pp];
    float acc = 0.f;
    for (int ch = 0; ch < C; ++ch) {
        const int64_t o = ((int64_t)b * C + ch) * V;
        const float u = U[o + c], gu = GU[o + c], fgu = FGU[o + c];
        acc += gu * Dv[o + pp] + u * OG[o + p] - 2.f * w * u * gu;
        acc -= fgu * wdp + wu * OG2[o + p] - 2.f * w * wu * fgu;
    }
    grad_w[i] = acc;
}

static size_t refine_lds_bytes(int V) { return (size_t)V * 12 + 2 * (size_t)(V + 2); }

}  // namespace bxi

extern "C" {

size_t bxi_mst_workspace_bytes(int B) { return sizeof(int) * (size_t)(B > 0 ? B : 1); }

int bxi_mst_forward_i32(const int* edge_index, const float* edge_weight, int B, int E, int V, int* edge_out, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (B < 0 || E <= 0 || V <= 1) return BXI_ERR_BAD_SHAPE;
    if (V > bxi::kTfMaxV || E > 8 * bxi::kTfMaxV) return BXI_ERR_UNSUPPORTED;
    if (B == 0) return BXI_OK;
    if (!edge_index || !edge_weight || !edge_out) return BXI_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < bxi_mst_workspace_bytes(B) || (reinterpret_cast<uintptr_t>(workspace) & 3)) return BXI_ERR_WORKSPACE;
    const size_t lds = (size_t)V * 12 + 4 * (size_t)((E + 31) / 32) + 16;
    if (lds > 150 * 1024) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = bxi::as_stream(stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::mst_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("mst", s, bxi::mst_kernel, dim3(B), dim3(1024), lds, s, edge_index, edge_weight, E, V, edge_out, reinterpret_cast<int*>(workspace));
    return bxi::check_launch();
}

int bxi_bfs_forward_i32(const int* tree_edges, int B, int V, int max_adj, int* sorted_index, int* sorted_parent, int* sorted_child,
                        int* levels, void* stream) {
    if (B < 0 || V <= 1 || max_adj < 1) return BXI_ERR_BAD_SHAPE;
    if (V > bxi::kTfMaxV || max_adj > 8) return BXI_ERR_UNSUPPORTED;
    if (B == 0) return BXI_OK;
    if (!tree_edges || !sorted_index || !sorted_parent || !sorted_child || !levels) return BXI_ERR_NULL_POINTER;
    const size_t lds = (size_t)V * 8 + (size_t)V * 4 + (size_t)((V + 3) / 4) * 4 + 16;
    if (lds > 150 * 1024) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = bxi::as_stream(stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::bfs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("bfs", s, bxi::bfs_kernel, dim3(B), dim3(256), lds, s, tree_edges, V, max_adj, sorted_index, sorted_parent, sorted_child, levels);
    return bxi::check_launch();
}

static int launch_refine(bxi::RefineArgs& a, void* stream) {
    if (a.B < 0 || a.C <= 0 || a.V <= 1 || a.max_adj < 1) return BXI_ERR_BAD_SHAPE;
    if (a.V > bxi::kTfMaxV || a.C > 65535) return BXI_ERR_UNSUPPORTED;
    if (a.B == 0) return BXI_OK;
    if (!a.in || !a.edge_weight || !a.sorted_index || !a.sorted_child || !a.levels) return BXI_ERR_NULL_POINTER;
    const size_t lds = bxi::refine_lds_bytes(a.V);
    if (lds > 150 * 1024) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = bxi::as_stream(stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::tree_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("tree_refine", s, bxi::tree_refine_kernel, dim3(a.B, a.C), dim3(256), lds, s, a);
    return bxi::check_launch();
}

int bxi_tree_refine_forward_f32(const float* feature_in, const float* edge_weight, const int* sorted_index, const int* sorted_child,
                                const int* levels, int B, int C, int V, int max_adj, float* feature_out, float* feature_aggr,
                                float* feature_aggr_up, float* weight_sum, float* weight_sum_up, void* stream) {
    if (B > 0 && (!feature_out || !feature_aggr || !feature_aggr_up || !weight_sum || !weight_sum_up)) return BXI_ERR_NULL_POINTER;
    bxi::RefineArgs a{};
    a.in = feature_in; a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child; a.levels = levels;
    a.up_sorted = feature_aggr_up; a.down_vertex = feature_aggr; a.out_vertex = feature_out;
    a.wsum_up_sorted = weight_sum_up; a.wsum_vertex = weight_sum;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.with_ones = 1;
    return launch_refine(a, stream);
}

int bxi_tree_refine_backward_feature_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                         const int* sorted_child, const int* levels, const float* weight_sum, int B, int C, int V,
                                         int max_adj, float* grad_feature, void* stream) {
    if (B > 0 && (!weight_sum || !grad_feature)) return BXI_ERR_NULL_POINTER;
    bxi::RefineArgs a{};
    a.in = grad_out; a.pre_div = weight_sum; a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child;
    a.levels = levels; a.down_vertex = grad_feature;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.with_ones = 0;
    return launch_refine(a, stream);
}

size_t bxi_tree_refine_backward_weight_workspace_bytes(int B, int C, int V) {
    if (B < 0 || C <= 0 || V <= 0) return 0;
    return sizeof(float) * 4 * (size_t)(B > 0 ? B : 1) * C * V;
}

int bxi_tree_refine_backward_weight_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                        const int* sorted_parent, const int* sorted_child, const int* levels, const float* feature_out,
                                        const float* feature_aggr, const float* feature_aggr_up, const float* weight_sum,
                                        const float* weight_sum_up, int B, int C, int V, int max_adj, float* grad_weight,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || C <= 0 || V <= 1) return BXI_ERR_BAD_SHAPE;
    if (B == 0) return BXI_OK;
    if (!grad_out || !sorted_parent || !feature_out || !feature_aggr || !feature_aggr_up || !weight_sum || !weight_sum_up || !grad_weight)
        return BXI_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < bxi_tree_refine_backward_weight_workspace_bytes(B, C, V) || (reinterpret_cast<uintptr_t>(workspace) & 3))
        return BXI_ERR_WORKSPACE;
    const size_t plane = (size_t)B * C * V;
    float* GU = reinterpret_cast<float*>(workspace); float* OG = GU + plane; float* FGU = OG + plane; float* OG2 = FGU + plane;
    bxi::RefineArgs a{};
    a.in = grad_out; a.pre_div = weight_sum; a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child;
    a.levels = levels; a.up_sorted = GU; a.down_sorted = OG;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.with_ones = 0;
    int rc = launch_refine(a, stream);
    if (rc != BXI_OK) return rc;
    a.pre_mul = feature_out; a.up_sorted = FGU; a.down_sorted = OG2;
    rc = launch_refine(a, stream);
    if (rc != BXI_OK) return rc;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)B * V + 255) / 256);
    BXI_LAUNCH("tree_grad_weight", s, bxi::tree_grad_weight_kernel, dim3(grid), dim3(256), 0, s, feature_aggr_up, feature_aggr, weight_sum_up,
               weight_sum, GU, OG, FGU, OG2, edge_weight, sorted_index, sorted_parent, B, C, V, grad_weight);
    return bxi::check_launch();
}

}  // extern "C"
