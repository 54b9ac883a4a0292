// tree_filter_large.hip -- the tree_filter extension for graphs that do not fit LDS (V > 10200 vertices).
//
// BoxLevelSet filters its mask-feature maps at full resolution (box_solov2_head.py:354-358: e.g. 200 x 304 = 60 800
// vertices); the reference's kernels have no vertex limit (mmdet/ops/tree_filter/src/{mst/boruvka.cpp:20-112,
// bfs/bfs.cu:46-98, refine/refine.cu:70-199}).  tree_filter.hip keeps everything a traversal touches in LDS and stops at
// 10 200 vertices; the kernels here are the same algorithms with the per-graph arrays in a caller-supplied global
// workspace and 32-bit indices, one 1024-thread workgroup per graph (mst, bfs) or per (graph, channel) (refine):
//   mst     parallel Boruvka: per round every edge offers (weight bits, edge index) to both components (64-bit atomic min),
//           roots hook to their cheapest neighbour (mutual pairs keep the smaller id), pointer jumping, relabel.  Same
//           total order (weight, index) as the LDS kernel and as the reference's strict '>' -> the same edge set.
//   bfs     level by level with the whole workgroup: a block-wide prefix sum of the children counts places the next
//           frontier (deterministic, children of a node contiguous, neighbours in ascending vertex order).
//   refine  leaf->root level by level (the workgroup takes a level's nodes; children are contiguous), root->leaf by pointer
//           jumping over affine maps exactly as in tree_filter.hip.
// Workgroup barriers order the global traffic: a workgroup runs on one CU, whose vector L1 is write-through, and every
// array here is private to the workgroup.
#include "common.hpp"
#include <atomic>

namespace bxi {

typedef unsigned long long u64;
constexpr int kLT = 1024;

__host__ __device__ static inline size_t up16(size_t v) { return (v + 15) / 16 * 16; }

// ---------------------------------------------------------------------------------------------------
struct MstLargeWs { u64* best; uint32_t* comp; uint32_t* link; uint32_t* chosen; int* act; };   // act[r]: an edge was offered in round r (r < 32); act[32]: the tree is complete
__host__ __device__ static size_t carve_mst_large(char* base, int E, int V, MstLargeWs* w) {
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += up16(b); return base ? base + o : nullptr; };
    MstLargeWs t;
    t.best = (u64*)take(8 * (size_t)V); t.comp = (uint32_t*)take(4 * (size_t)V); t.link = (uint32_t*)take(4 * (size_t)V);
    t.chosen = (uint32_t*)take(4 * (size_t)((E + 31) / 32));
    t.act = (int*)take(4 * 33);
    if (w) *w = t;
    return off;
}
size_t mst_large_ws_bytes(int E, int V) { return carve_mst_large(nullptr, E, V, nullptr); }

// Boruvka ACROSS THE GPU: every phase of a round is a grid over the edges / vertices of all graphs, a kernel boundary between
// phases is the grid barrier (one 1024-thread workgroup per graph took 2.5 ms at 60 800 vertices / 121 000 edges: 16 rounds of
// ~120 edges per thread, each with two dependent atomics).  The number of rounds is fixed on the host (the component count at least
// halves per round: ceil(log2 V) rounds); the first round that offers no edge marks the tree complete and every later launch returns at
// once (a grid graph needs about nine of the sixteen rounds at 200 x 304).
__device__ __forceinline__ MstLargeWs mst_ws(char* ws_base, size_t ws_stride, int b, int E, int V) {
    MstLargeWs w;
    carve_mst_large(ws_base + (size_t)b * ws_stride, E, V, &w);
    return w;
}
__global__ __launch_bounds__(256) void mstL_init_kernel(int E, int V, char* ws_base, size_t ws_stride) {
    const MstLargeWs w = mst_ws(ws_base, ws_stride, blockIdx.y, E, V);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < V) { w.comp[i] = (uint32_t)i; w.best[i] = ~0ull; }
    if (i < (E + 31) / 32) w.chosen[i] = 0u;
    if (i < 33) w.act[i] = 0;
}
// every edge between two components offers (weight bits, edge index) to both (64-bit atomic min at the memory side).  In the late rounds a
// handful of components take every offer -- 121 000 atomics on a few words: 12-15 us a round -- so a wave whose offers all go to one
// component (neighbouring edges: the usual case) sends their minimum, once.
__device__ __forceinline__ void offer_min(u64* best, uint32_t comp, u64 key, bool act) {
    const unsigned long long m = __ballot(act);
    if (!m) return;
    const int first = __ffsll((long long)m) - 1;
    const uint32_t c0 = (uint32_t)__shfl((int)comp, first, 64);
    if (__all(!act || comp == c0)) {
        const u64 k = ~wave_max_u64(act ? ~key : 0ull);                     // the minimum over the active lanes
        if ((int)(threadIdx.x & 63) == first) atomicMin(&best[c0], k);
    } else if (act) atomicMin(&best[comp], key);
}
__global__ __launch_bounds__(256) void mstL_offer_kernel(const int* __restrict__ edge_index, const float* __restrict__ edge_weight, int E, int V,
                                                         char* ws_base, size_t ws_stride, int round) {
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    if (w.act[32]) return;                                           // (uniform: the whole launch returns)
    const int* idx = edge_index + (int64_t)b * E * 2;
    uint32_t cu = 0u, cv = 0u;
    if (e < E) { cu = w.comp[idx[2 * e]]; cv = w.comp[idx[2 * e + 1]]; }
    const bool act = e < E && cu != cv;
    u64 key = ~0ull;
    if (act) {
        if (w.act[round] == 0) w.act[round] = 1;                      // (every writer writes 1; the word is read by the NEXT kernels only)
        key = ((u64)__float_as_uint(edge_weight[(int64_t)b * E + e]) << 32) | (uint32_t)e;   // weights are >= 0: the bits order like the values
    }
    offer_min(w.best, cu, key, act);
    offer_min(w.best, cv, key, act);
}
// every component root takes its cheapest edge and hooks to the component at its other end
__global__ __launch_bounds__(256) void mstL_hook_kernel(const int* __restrict__ edge_index, int E, int V, char* ws_base, size_t ws_stride, int round) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= V) return;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    if (w.act[round] == 0) { if (c == 0) w.act[32] = 1; return; }   // no edge joins two components: one component, the tree is complete
                                                                     // (act[32] is read by later kernels only; this one decides on act[round] alone)
    if (w.comp[c] != (uint32_t)c) return;
    const int* idx = edge_index + (int64_t)b * E * 2;
    const u64 k = w.best[c];
    uint32_t to = (uint32_t)c;
    if (k != ~0ull) {
        const uint32_t e = (uint32_t)k;
        atomicOr(&w.chosen[e >> 5], 1u << (e & 31));
        const uint32_t cu = w.comp[idx[2 * e]], cv = w.comp[idx[2 * e + 1]];
        to = cu == (uint32_t)c ? cv : cu;
    }
    w.link[c] = to;
}
// two components that chose each other: the smaller id is the root (reads links of the previous kernel, writes its own)
__global__ __launch_bounds__(256) void mstL_mutual_kernel(int E, int V, char* ws_base, size_t ws_stride, int round) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= V) return;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    if (w.act[round] == 0) return;
    if (w.comp[c] != (uint32_t)c) return;
    const uint32_t o = w.link[c];
    // (c < o decides alone: the partner, if it also points back, keeps its link to c)
    if (o != (uint32_t)c && (uint32_t)c < o && w.link[o] == (uint32_t)c) w.link[c] = (uint32_t)c;
}
// comp[v] = root of comp[v]; the next round's offers start from scratch
__global__ __launch_bounds__(256) void mstL_relabel_kernel(int E, int V, char* ws_base, size_t ws_stride, int round) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    if (w.act[round] == 0) return;
    uint32_t* lk = w.link;
    // The hook chains are walked here (a root points to itself), with PATH HALVING: every node a walker passes is re-pointed at its
    // grandparent, so concurrent walkers shorten each other's way and even a chain as long as the graph (a 1 x V strip with
    // increasing weights) costs O(log) steps per thread.  The re-pointing races with other readers, harmlessly: a link only ever
    // moves to an ancestor, and the root a walk ends at is the chain's one self-pointing node whatever it saw on the way.
    uint32_t r = w.comp[v];
    for (;;) {
        const uint32_t p = lk[r];
        if (p == r) break;
        const uint32_t gp = lk[p];
        if (gp != p) lk[r] = gp;
        r = gp;
    }
    w.comp[v] = r;
    w.best[v] = ~0ull;
}
// tree edges in ascending edge order: (1) exclusive prefix of the bitmap words' popcounts (one workgroup per graph: a few thousand
// words), (2) a thread per word writes its edges at its prefix
__global__ __launch_bounds__(kLT) void mstL_scan_kernel(int E, int V, int* __restrict__ n_out, char* ws_base, size_t ws_stride) {
    __shared__ int scan[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    const int nwords = (E + 31) / 32;
    const int per = (nwords + kLT - 1) / kLT;
    const int w0 = min(tid * per, nwords), w1 = min(w0 + per, nwords);
    int cnt = 0;
    for (int i = w0; i < w1; ++i) cnt += __popc(w.chosen[i]);
    const int lane = tid & 63, wave = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    if (tid == 0) { int s = 0; for (int i = 0; i < 16; ++i) { const int t = scan[i]; scan[i] = s; s += t; } scan[16] = s; }
    __syncthreads();
    int pos = scan[wave] + incl - cnt;
    uint32_t* pre = w.link;                               // the links are done with: the words' prefixes live there (nwords <= V)
    for (int i = w0; i < w1; ++i) { pre[i] = (uint32_t)pos; pos += __popc(w.chosen[i]); }
    if (tid == 0) n_out[b] = scan[16];
}
__global__ __launch_bounds__(256) void mstL_emit_kernel(const int* __restrict__ edge_index, int E, int V, int* __restrict__ edge_out, char* ws_base,
                                                        size_t ws_stride) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (E + 31) / 32) return;
    const MstLargeWs w = mst_ws(ws_base, ws_stride, b, E, V);
    const int* idx = edge_index + (int64_t)b * E * 2;
    int* out = edge_out + (int64_t)b * (V - 1) * 2;
    uint32_t m = w.chosen[i];
    int pos = (int)w.link[i];
    while (m) {
        const int e = i * 32 + __ffs((int)m) - 1;
        m &= m - 1;
        if (pos < V - 1) *reinterpret_cast<int2*>(out + 2 * pos) = *reinterpret_cast<const int2*>(idx + 2 * e);
        ++pos;
    }
}

int launch_mst_large(const int* edge_index, const float* edge_weight, int B, int E, int V, int* edge_out, int* n_out, char* ws,
                     hipStream_t s) {
    if (B > 65535) return BXI_ERR_BAD_SHAPE;
    const size_t stride = mst_large_ws_bytes(E, V);
    const int nw = (E + 31) / 32;
    const dim3 gv((unsigned)((V + 255) / 256), (unsigned)B), ge((unsigned)((E + 255) / 256), (unsigned)B);
    const dim3 gi((unsigned)(((V > nw ? V : nw) + 255) / 256), (unsigned)B);
    BXI_LAUNCH("mst_large_init", s, mstL_init_kernel, gi, dim3(256), 0, s, E, V, ws, stride);
    int rounds = 0;
    while ((1 << rounds) < V) ++rounds;
    for (int r = 0; r < rounds; ++r) {
        BXI_LAUNCH("mst_large_offer", s, mstL_offer_kernel, ge, dim3(256), 0, s, edge_index, edge_weight, E, V, ws, stride, r);
        BXI_LAUNCH("mst_large_hook", s, mstL_hook_kernel, gv, dim3(256), 0, s, edge_index, E, V, ws, stride, r);
        BXI_LAUNCH("mst_large_mutual", s, mstL_mutual_kernel, gv, dim3(256), 0, s, E, V, ws, stride, r);
        // (pointer-jumping kernels in front of the relabel kernel's halving walk -- ceil(log2 V) - r, 3, 2, 1 of them -- measured 680,
        // 330-370, 290-305, 235-243 us per MST at 200 x 304 against 195-216 us without: every launch costs more than the steps it saves)
        BXI_LAUNCH("mst_large_relabel", s, mstL_relabel_kernel, gv, dim3(256), 0, s, E, V, ws, stride, r);
    }
    if (nw > V) return BXI_ERR_UNSUPPORTED;               // (the word prefixes reuse the link array; E <= 8 V is checked by the caller)
    BXI_LAUNCH("mst_large_scan", s, mstL_scan_kernel, dim3(B), dim3(kLT), 0, s, E, V, n_out, ws, stride);
    BXI_LAUNCH("mst_large_emit", s, mstL_emit_kernel, dim3((unsigned)((nw + 255) / 256), (unsigned)B), dim3(256), 0, s, edge_index, E, V, edge_out, ws, stride);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
struct BfsLargeWs {
    uint32_t* adj; uint32_t* deg; uint32_t* nodev; uint32_t* nodep; uint32_t* pos_of; int* flag; int* nf; int* gw; unsigned char* gmask;
    // the Euler-tour form (below): two list rankings over the 4V arc slots, per-vertex results, the pairs of the radix sort
    unsigned long long* e1; unsigned long long* e2; int* pk; uint32_t* key0; uint32_t* key1; uint32_t* val0; uint32_t* val1; uint32_t* hist; int* bad;
    // (named pointers, not arrays: a pointer array indexed by the pass parity cost the refine rounds a factor of three, profiles/NOTES.md R3-2e)
};
constexpr int kEulerMaxV = (1 << 20) - 1;          // arc ids (and the end marker 4V) < 2^22 and counts < 2^21 share one 64-bit word
constexpr int kSortTile = 1024, kSortBits = 9, kSortBuckets = 1 << kSortBits;
__host__ __device__ static size_t carve_bfs_large(char* base, int V, BfsLargeWs* w) {
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += up16(b); return base ? base + o : nullptr; };
    BfsLargeWs t;
    t.adj = (uint32_t*)take(16 * (size_t)V); t.deg = (uint32_t*)take(4 * (size_t)V); t.nodev = (uint32_t*)take(4 * (size_t)(V + 1));
    t.nodep = (uint32_t*)take(4 * (size_t)(V + 1)); t.pos_of = (uint32_t*)take(4 * (size_t)V);
    t.flag = (int*)take(4); t.nf = (int*)take(4); t.gw = (int*)take(4); t.gmask = (unsigned char*)take((size_t)V);
    const bool euler = V <= kEulerMaxV;
    const size_t nblk = ((size_t)V + kSortTile - 1) / kSortTile;
    t.e1 = (unsigned long long*)take(euler ? 32 * (size_t)V : 0); t.e2 = (unsigned long long*)take(euler ? 32 * (size_t)V : 0);
    t.pk = (int*)take(euler ? 4 * (size_t)V : 0);
    t.key0 = (uint32_t*)take(euler ? 4 * (size_t)V : 0); t.val0 = (uint32_t*)take(euler ? 4 * (size_t)V : 0);
    t.key1 = (uint32_t*)take(euler ? 4 * (size_t)V : 0); t.val1 = (uint32_t*)take(euler ? 4 * (size_t)V : 0);
    t.hist = (uint32_t*)take(euler ? 4 * (size_t)kSortBuckets * (nblk + 1) : 0);       // [tile][bucket], then the buckets' bases
    t.bad = (int*)take(4);
    if (w) *w = t;
    return off;
}
size_t bfs_large_ws_bytes(int V) { return carve_bfs_large(nullptr, V, nullptr); }

// block-wide exclusive prefix sum of a small per-thread count (0..4); returns the total through `total`
__device__ __forceinline__ int block_excl_scan(int v, int* part /*[17]*/, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
    __syncthreads();                                   // `part` of the previous call has been read by everyone
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kLT / 64; ++i) { const int p = part[i]; if (i < wave) base += p; tot += p; }
    total = tot;
    return base + incl - v;
}

// BFS of a large tree: the passes over the vertices / edges (adjacency, sorting, outputs) run ACROSS THE GPU; only the level walk
// -- a chain through the tree's depth, ~1700 levels at 200 x 304 -- is one workgroup per graph, and while the frontier is at most
// 256 nodes wide (measured on random 200 x 304 trees: mean 35, max ~115) it is ONE WAVE with the frontier in LDS: per level one dependent read of the adjacency of the
// frontier's nodes, four ballots for the children's places, no workgroup barrier.  A tree on a 4-connected grid (what the callers
// build: tree_filter_oracle / MinimumSpanningTree over grid edges) keeps that adjacency as four bits per vertex in LDS, so the walk
// makes no global read at all; any other tree of degree <= 4 reads the 16-byte records from (warmed) L2.
__device__ __forceinline__ BfsLargeWs bfs_ws(char* ws_base, size_t ws_stride, int b, int V) {
    BfsLargeWs w;
    carve_bfs_large(ws_base + (size_t)b * ws_stride, V, &w);
    return w;
}
__global__ __launch_bounds__(256) void bfsL_zero_kernel(int V, int max_adj, int* __restrict__ sorted_child, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    if (i < V) w.deg[i] = 0u;
    if (i == 0) { *w.flag = 0; *w.gw = 0; *w.bad = 0; }
    if (i < (int64_t)V * max_adj) sorted_child[(int64_t)b * V * max_adj + i] = 0;
}
__global__ __launch_bounds__(256) void bfsL_adj_kernel(const int* __restrict__ tree, int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= V - 1) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const int* ed = tree + (int64_t)b * (V - 1) * 2;
    const int u = ed[2 * e], v = ed[2 * e + 1];           // adjacency (a vertex of a grid tree has at most 4 neighbours)
    const unsigned su = atomicAdd(&w.deg[u], 1u), sv = atomicAdd(&w.deg[v], 1u);
    if (su < 4) w.adj[4 * (size_t)u + su] = (uint32_t)v;
    if (sv < 4) w.adj[4 * (size_t)v + sv] = (uint32_t)u;
    atomicMax(w.gw, u > v ? u - v : v - u);              // on a 4-connected grid: its width
}
__global__ __launch_bounds__(256) void bfsL_sort_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    if (w.deg[v] > 4u) atomicOr(w.flag, 1);              // more than 4 neighbours: not a tree on a grid (edges would be dropped)
    const int d = min((int)w.deg[v], 4);                  // arrival order of the atomics -> ascending neighbour ids
    uint32_t a[4];
    for (int k = 0; k < 4; ++k) a[k] = k < d ? w.adj[4 * (size_t)v + k] : 0xffffffffu;
    for (int i = 1; i < 4; ++i) for (int j = i; j > 0 && a[j - 1] > a[j]; --j) { const uint32_t t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; }
    *reinterpret_cast<uint4*>(w.adj + 4 * (size_t)v) = make_uint4(a[0], a[1], a[2], a[3]);
    // a tree on a 4-connected grid of width W (what mst() is given: every edge joins v and v +- 1 or v +- W) has its whole adjacency
    // in four bits per vertex -- v - W, v - 1, v + 1, v + W, which is also the ascending order -- and then fits LDS
    const int W = *w.gw;
    unsigned m = 0u;
    bool grid = W > 1;
    for (int k = 0; k < d; ++k) {
        const int delta = (int)a[k] - v;
        const int bit = delta == -W ? 0 : (delta == -1 ? 1 : (delta == 1 ? 2 : (delta == W ? 3 : -1)));
        if (bit < 0) grid = false; else m |= 1u << bit;
    }
    w.gmask[v] = (unsigned char)m;
    if (!grid) atomicOr(w.flag, 2);
}

constexpr int kBfsNarrow = 256;                    // widest frontier the one-wave form walks (four nodes per lane)
constexpr int kBfsNext = 3 * kBfsNarrow + 4;
constexpr int kBfsMaskCap = 128 * 1024;            // vertices whose 4-bit adjacency (a byte each) the walk keeps in LDS
struct BfsWalkLds {
    int part[17];
    uint2 fr[2][kBfsNext];                              // the frontier of the one-wave form and the next one; <= 3 children of <= 256 nodes (the root: 4).
                                                        // General trees: (vertex, parent) pairs.  Grid trees: the same bytes as two buffers of words.
    int s_lo, s_hi, s_n, s_depth;
};
// ---- grid trees: a node is ONE word, vertex | (one-hot direction of its parent) << 20, in LDS and in nodev (nodep is not used); the
// directions are 0: v - W, 1: v - 1, 2: v + 1, 3: v + W.  The walk's loop has no global LOAD in it, so nothing ever waits for its global
// stores (gfx9 counts loads and stores in one vmcnt: with a load in the loop every level also waits for the previous level's stores),
// and it is short: one wave alone issues an instruction every ~5 cycles, so the level time is the instruction count (profiles/NOTES.md R3-2c).
constexpr uint32_t kBfsVtx = 0xFFFFFu;
static_assert(kBfsMaskCap <= (int)kBfsVtx, "a grid node is vertex | direction << 20");
__device__ __forceinline__ bool bfs_grid_mode(const BfsLargeWs& w, int V) { return !(*w.flag & 2) && V <= kBfsMaskCap; }
__device__ __forceinline__ uint32_t bfs_grid_parent(uint32_t e, int GW) {
    const uint32_t v = e & kBfsVtx, pm = e >> 20;
    return (pm & 1u) ? v - (uint32_t)GW : ((pm & 2u) ? v - 1u : ((pm & 4u) ? v + 1u : v + (uint32_t)GW));
}
__device__ __forceinline__ void bfs_walk_grid(int V, int GW, int* __restrict__ lv, const BfsLargeWs& w, const unsigned char* lmask, BfsWalkLds& L) {
    const int tid = threadIdx.x;
    uint32_t* fr0 = reinterpret_cast<uint32_t*>(&L.fr[0][0]);     // two buffers of kBfsNext words
    const uint32_t off[4] = {(uint32_t)-GW, 0xffffffffu, 1u, (uint32_t)GW};
    int lo = 0, hi = 1, n = 1, depth = 0;
    while (lo < hi) {                                   // workgroup-uniform
        if (hi - lo <= kBfsNarrow) {
            // ---- one wave walks while the frontier stays narrow, 64 nodes at a time in frontier order; a node is written to nodev when it
            // is PROCESSED (one coalesced store per 64 nodes), its children go to the other LDS buffer.  On entry the frontier is in buffer 0.
            if (tid < 64) {
                const int lane = tid;
                int cb = 0;
                while (lo < hi && hi - lo <= kBfsNarrow) {
                    const int width = hi - lo;
                    const uint32_t* cur_f = fr0 + cb * kBfsNext;
                    uint32_t* next_f = fr0 + (cb ^ 1) * kBfsNext;
                    int total = 0;
                    for (int c = 0; c < width; c += 64) {
                        uint32_t e = 0u, m = 0u;
                        if (c + lane < width) {
                            e = cur_f[c + lane];
                            w.nodev[lo + c + lane] = e;
                            m = (uint32_t)lmask[e & kBfsVtx] & ~(e >> 20);
                        }
                        const uint32_t cur = e & kBfsVtx;
                        int slot = total;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned long long mk = __ballot((m >> k) & 1u);
                            slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)slot));
                            total += __popcll(mk);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if ((m >> k) & 1u) { next_f[slot] = (cur + off[k]) | ((8u >> k) << 20); ++slot; }
                    }
                    ++depth;
                    if (lane == 0) lv[1 + depth] = hi;                           // off[depth] = end of this level
                    lo = hi; hi = n + total; n += total;
                    cb ^= 1;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the next frontier is in LDS (one wave: its LDS operations execute in order)
                }
                // a frontier too wide for one wave: the workgroup continues from nodev
                if (lo < hi) { const uint32_t* f = fr0 + cb * kBfsNext; for (int i = lane; i < hi - lo; i += 64) w.nodev[lo + i] = f[i]; }
                if (lane == 0) { L.s_lo = lo; L.s_hi = hi; L.s_n = n; L.s_depth = depth; }
            }
            __syncthreads();                             // also: the walking wave's global stores are done (vmcnt) before others read them
            lo = L.s_lo; hi = L.s_hi; n = L.s_n; depth = L.s_depth;
            __syncthreads();
            continue;
        }
        for (int base = lo; base < hi; base += kLT) {
            const int i = base + tid;
            const uint32_t e = i < hi ? w.nodev[i] : 0u;
            const uint32_t cur = e & kBfsVtx;
            const uint32_t m = i < hi ? ((uint32_t)lmask[cur] & ~(e >> 20)) : 0u;
            int total;
            int slot = n + block_excl_scan(__popc(m), L.part, total);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((m >> k) & 1u) {
                    const uint32_t ne = (cur + off[k]) | ((8u >> k) << 20);
                    w.nodev[slot] = ne;
                    if (slot - hi < kBfsNarrow) fr0[slot - hi] = ne;             // in case the next level is narrow (n == hi when a level starts)
                    ++slot;
                }
            n += total;
        }
        ++depth;
        if (tid == 0) lv[1 + depth] = hi;               // off[depth] = end of this level
        __syncthreads();                                // the next level's nodes are written
        lo = hi; hi = n;
    }
    if (tid == 0) { lv[0] = ((*w.flag & 1) || n < V) ? -1 : depth; *w.nf = n; }
}
// ---- any other tree of degree <= 4: nodes are (vertex, parent) pairs, the adjacency is read from (warmed) L2
__device__ __forceinline__ void bfs_walk_general(int V, int* __restrict__ lv, const BfsLargeWs& w, BfsWalkLds& L) {
    const int tid = threadIdx.x;
    auto neighbours = [&](bool act, uint32_t cur, uint32_t (&nb)[4]) {
        const uint4 q = act ? *reinterpret_cast<const uint4*>(w.adj + 4 * (size_t)cur) : make_uint4(~0u, ~0u, ~0u, ~0u);
        nb[0] = q.x; nb[1] = q.y; nb[2] = q.z; nb[3] = q.w;
    };
    int lo = 0, hi = 1, n = 1, depth = 0;
    while (lo < hi) {                                   // workgroup-uniform
        if (hi - lo <= kBfsNarrow) {
            // ---- one wave walks while the frontier stays narrow (64 nodes at a time, in frontier order; the next frontier goes to the
            // other LDS buffer); the other waves wait at the barrier below.  On entry the frontier is in buffer 0.
            if (tid < 64) {
                const int lane = tid;
                const unsigned long long lt = (1ull << lane) - 1ull;
                int cb = 0;
                while (lo < hi && hi - lo <= kBfsNarrow) {
                    const int width = hi - lo;
                    int total = 0;
                    for (int c = 0; c < width; c += 64) {
                        const bool act = c + lane < width;
                        const uint2 e = act ? L.fr[cb][c + lane] : make_uint2(0u, 0u);
                        const uint32_t cur = e.x, par = e.y;
                        uint32_t nb[4];
                        neighbours(act, cur, nb);
                        bool ok[4];
                        int before = 0, mine = 0, tot_c = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ok[k] = act && nb[k] != 0xffffffffu && nb[k] != par;
                            const unsigned long long m = __ballot(ok[k]);
                            before += __popcll(m & lt);
                            tot_c += __popcll(m);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (ok[k]) {
                                const int slot = total + before + mine;
                                L.fr[cb ^ 1][slot] = make_uint2(nb[k], cur);
                                w.nodev[n + slot] = nb[k]; w.nodep[n + slot] = cur;
                                ++mine;
                            }
                        total += tot_c;
                    }
                    ++depth;
                    if (lane == 0) lv[1 + depth] = hi;                           // off[depth] = end of this level
                    lo = hi; hi = n + total; n += total;
                    cb ^= 1;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the next frontier is in LDS (one wave: its LDS operations execute in order)
                }
                if (lane == 0) { L.s_lo = lo; L.s_hi = hi; L.s_n = n; L.s_depth = depth; }
            }
            __syncthreads();                             // also: the walking wave's global stores of nodev / nodep are done (vmcnt) before others read them
            lo = L.s_lo; hi = L.s_hi; n = L.s_n; depth = L.s_depth;
            __syncthreads();
            continue;
        }
        for (int base = lo; base < hi; base += kLT) {
            const int i = base + tid;
            const bool act = i < hi;
            const uint32_t cur = act ? w.nodev[i] : 0u, par = act ? w.nodep[i] : 0u;
            uint32_t nb[4]; int rank[4]; bool ok[4];
            int nch = 0;
            neighbours(act, cur, nb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ok[k] = act && nb[k] != 0xffffffffu && nb[k] != par;
                rank[k] = nch;
                nch += ok[k] ? 1 : 0;
            }
            int total;
            const int pos = n + block_excl_scan(nch, L.part, total);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) {
                    w.nodev[pos + rank[k]] = nb[k]; w.nodep[pos + rank[k]] = cur;
                    if (pos + rank[k] - hi < kBfsNarrow) L.fr[0][pos + rank[k] - hi] = make_uint2(nb[k], cur);   // in case the next level is narrow (n == hi when a level starts)
                }
            n += total;
        }
        ++depth;
        if (tid == 0) lv[1 + depth] = hi;               // off[depth] = end of this level
        __syncthreads();                                // the next level's nodes are written
        lo = hi; hi = n;
    }
    // loud, not silent: a vertex of degree > 4 or an input that is not connected leaves depth = -1, which makes
    // bxi_tree_refine_* poison its output (the reference's bfs.cu walks whatever it is given; refine.cu then reads garbage)
    if (tid == 0) { lv[0] = ((*w.flag & 1) || n < V) ? -1 : depth; *w.nf = n; }
}
__global__ __launch_bounds__(kLT) void bfsL_walk_kernel(int V, int* __restrict__ levels, char* ws_base, size_t ws_stride) {
    extern __shared__ unsigned char lmask[];            // [V] in grid mode
    __shared__ BfsWalkLds L;
    const int b = blockIdx.x, tid = threadIdx.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    int* lv = levels + (int64_t)b * (V + 2);
    // Grid mode (the tree lives on a 4-connected grid and its byte-per-vertex adjacency fits LDS): the walk never reads global memory.
    // Otherwise every adjacency record is read exactly once, one dependent global read per level: the table (16 B per vertex) is
    // brought into this XCD's L2 first, with the whole workgroup, so that the walk's reads are L2 hits instead of trips to HBM.
    // (readfirstlane: the two words are waited for HERE, not by a vmcnt(0) inside the walk's loop at their first use)
    const bool gridm = __builtin_amdgcn_readfirstlane((int)bfs_grid_mode(w, V)) != 0;           // workgroup-uniform
    const int GW = __builtin_amdgcn_readfirstlane(*w.gw);
    if (tid == 0) {
        lv[1] = 0; w.nodev[0] = 0u; w.nodep[0] = 0xffffffffu;
        if (gridm) reinterpret_cast<uint32_t*>(&L.fr[0][0])[0] = 0u; else L.fr[0][0] = make_uint2(0u, 0xffffffffu);
    }
    if (gridm) {
        const uint32_t* g4 = reinterpret_cast<const uint32_t*>(w.gmask);          // gmask is 16-byte aligned in the workspace, and padded
        uint32_t* l4 = reinterpret_cast<uint32_t*>(lmask);
        for (int v = tid; v < (V + 3) / 4; v += kLT) l4[v] = g4[v];
    } else {
        uint32_t sink = 0u;
        for (int v = tid; v < V; v += kLT) { const uint4 q = *reinterpret_cast<const uint4*>(w.adj + 4 * (size_t)v); sink ^= q.x ^ q.y ^ q.z ^ q.w; }
        asm volatile("" ::"v"(sink));
    }
    __syncthreads();
    if (gridm) bfs_walk_grid(V, GW, lv, w, lmask, L);
    else bfs_walk_general(V, lv, w, L);
}
__global__ __launch_bounds__(256) void bfsL_index_kernel(int V, int* __restrict__ sorted_index, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= V) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const int nf = *w.nf;                               // < V only for a disconnected input
    const uint32_t v = p < nf ? (bfs_grid_mode(w, V) ? (w.nodev[p] & kBfsVtx) : w.nodev[p]) : 0u;
    sorted_index[(int64_t)b * V + p] = (int)v;
    if (p < nf) w.pos_of[v] = (uint32_t)p;
}
__global__ __launch_bounds__(256) void bfsL_parent_kernel(int V, int max_adj, int* __restrict__ sorted_parent, int* __restrict__ sorted_child, char* ws_base,
                                                          size_t ws_stride) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= V) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const int nf = *w.nf;
    int* s_parent = sorted_parent + (int64_t)b * V;
    if (p == 0 || p >= nf) { s_parent[p] = 0; return; }
    const bool gridm = bfs_grid_mode(w, V);
    const int GW = *w.gw;
    auto parent_of = [&](int q) { return gridm ? bfs_grid_parent(w.nodev[q], GW) : w.nodep[q]; };
    const uint32_t pv = parent_of(p);
    const int pp = (int)w.pos_of[pv];
    s_parent[p] = pp;
    int k = 0;                                          // rank among the (contiguous) siblings
    while (k < 3 && p - k - 1 >= 1 && parent_of(p - k - 1) == pv) ++k;
    if (k < max_adj) sorted_child[(int64_t)b * V * max_adj + (size_t)pp * max_adj + k] = p;
}


// ---------------------------------------------------------------------------------------------------
// The Euler-tour form of the BFS: no walk through the tree's depth at all.  The level walk above is ~1700 dependent steps of one wave at
// 200 x 304 (0.61 ms); here every pass is grid-wide and their number depends on log V only:
//   1. every undirected edge is two ARCS, arc (v -> adj[v][k]) = slot 4v + k; the successor "after arriving at u from v, leave for the
//      neighbour that follows v in u's list (cyclically)" strings all arcs into one cycle, cut where it would return to arc 0: a list
//      from arc 0 = (0 -> adj[0][0]) to the last arc back into the root;
//   2. LIST RANKING by pointer jumping gives every arc its distance to the end of the list; of the two arcs of an edge the one earlier
//      in the tour leads AWAY from the root, which tells every vertex its parent;
//   3. a second tour that visits the children of every vertex in ascending order (this library's sibling order; bfs.cu's is the arrival
//      order of its atomics) is ranked with two
//      weights, "leads down" / "leads up": the suffix counts at the arc (parent -> v) give v's PREORDER number and its DEPTH;
//   4. BFS order = by depth, within a depth by preorder (subtrees are contiguous in preorder, so on every level the preorder of the nodes is
//      the preorder of their parents, then the child order -- the queue's order): a stable LSD radix sort of the preorder sequence by depth;
//      level d starts where the sorted key changes to d.
// Pointer jumping is done IN PLACE and asynchronously: a slot holds (successor, weight of the arcs from this one up to, not including, the
// successor) in ONE 64-bit word, so any value a thread reads -- this launch's or a staler one from its XCD's L2 -- is a correct pair, and
// jumping over it keeps the pair correct; every launch makes each thread jump kEulerJumps times over values at least as advanced as the
// previous launch left them, so ceil(log(arcs) / log(kEulerJumps + 1)) launches reach the end from everywhere.  Integer sums: the result is exact and
// the same on every run whatever the interleaving.
// Input that is not a connected tree of degree <= 4 leaves arcs that never reach the end of the list: reported as by the walk (levels[0] = -1).
constexpr int kEulerJumps = 4;
__device__ __forceinline__ unsigned long long e1_pack(uint32_t succ, uint32_t dist) { return ((unsigned long long)dist << 32) | succ; }
__device__ __forceinline__ unsigned long long e2_pack(uint32_t succ, uint32_t down, uint32_t up) {
    return ((unsigned long long)up << 43) | ((unsigned long long)down << 22) | succ;
}
__device__ __forceinline__ uint32_t e2_succ(unsigned long long x) { return (uint32_t)x & 0x3fffffu; }
__device__ __forceinline__ uint32_t e2_down(unsigned long long x) { return (uint32_t)(x >> 22) & 0x1fffffu; }
__device__ __forceinline__ uint32_t e2_up(unsigned long long x) { return (uint32_t)(x >> 43); }
// index of vertex v in u's (sorted) neighbour list, or -1
__device__ __forceinline__ int nb_index(const BfsLargeWs& w, uint32_t u, uint32_t v, int du) {
    const uint4 q = *reinterpret_cast<const uint4*>(w.adj + 4 * (size_t)u);
    return (du > 0 && q.x == v) ? 0 : ((du > 1 && q.y == v) ? 1 : ((du > 2 && q.z == v) ? 2 : ((du > 3 && q.w == v) ? 3 : -1)));
}
__global__ __launch_bounds__(256) void bfsE_succ1_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t a = blockIdx.x * 256 + threadIdx.x, SENT = 4u * (uint32_t)V;
    if (a >= SENT) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t v = a >> 2;
    const int k = (int)(a & 3u), d = (int)min(w.deg[v], 4u);
    unsigned long long out = e1_pack(SENT, 0u);                        // a slot without an arc
    if (k < d) {
        const uint32_t u = w.adj[a];
        if (u < (uint32_t)V) {
            const int du = (int)min(w.deg[u], 4u), j = nb_index(w, u, v, du);
            if (j >= 0) {
                const uint32_t s = 4u * u + (uint32_t)(j + 1 == du ? 0 : j + 1);
                out = e1_pack(s == 0u ? SENT : s, 1u);                  // the arc that would lead back to arc 0 ends the list
            } else atomicOr(w.bad, 1);
        } else atomicOr(w.bad, 1);
    }
    w.e1[a] = out;
}
__global__ __launch_bounds__(256) void bfsE_rank1_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t a = blockIdx.x * 256 + threadIdx.x, SENT = 4u * (uint32_t)V;
    if (a >= SENT) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    unsigned long long p = w.e1[a];
#pragma unroll
    for (int it = 0; it < kEulerJumps; ++it) {
        const uint32_t s = (uint32_t)p;
        if (s >= SENT) break;
        const unsigned long long q = w.e1[s];
        p = e1_pack((uint32_t)q, (uint32_t)(p >> 32) + (uint32_t)(q >> 32));
    }
    w.e1[a] = p;
}
// parent of every vertex: the one neighbour whose arc towards v comes earlier in the tour than v's arc towards it
__global__ __launch_bounds__(256) void bfsE_parent_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= (uint32_t)V) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t SENT = 4u * (uint32_t)V;
    const int d = (int)min(w.deg[v], 4u);
    int pk = -1, ups = 0;
    bool ended = true;
    for (int k = 0; k < d; ++k) {
        const unsigned long long mine = w.e1[4u * v + k];
        const uint32_t u = w.adj[4u * v + k];
        if (u >= (uint32_t)V) continue;
        const int j = nb_index(w, u, v, (int)min(w.deg[u], 4u));
        if (j < 0) continue;
        const unsigned long long theirs = w.e1[4u * u + j];
        ended = ended && (uint32_t)mine >= SENT && (uint32_t)theirs >= SENT;
        if ((uint32_t)(mine >> 32) < (uint32_t)(theirs >> 32)) { pk = k; ++ups; }      // fewer arcs left: later in the tour: this arc leads up
    }
    if (!ended || ups != (v == 0u ? 0 : 1)) atomicOr(w.bad, 1);       // an arc that never reached the end of the list: not one connected tree
    w.pk[v] = pk;
}
__global__ __launch_bounds__(256) void bfsE_succ2_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t a = blockIdx.x * 256 + threadIdx.x, SENT = 4u * (uint32_t)V;
    if (a >= SENT) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t v = a >> 2;
    const int k = (int)(a & 3u), d = (int)min(w.deg[v], 4u);
    unsigned long long out = e2_pack(SENT, 0u, 0u);
    if (k < d) {
        const uint32_t u = w.adj[a];
        if (u < (uint32_t)V) {
            const int du = (int)min(w.deg[u], 4u), pu = w.pk[u];
            if (k != w.pk[v]) {                                        // down: arriving at u from its parent -> its first child, or straight back up
                const int c = pu == 0 ? 1 : 0;
                out = e2_pack(c < du ? 4u * u + c : (pu >= 0 ? 4u * u + pu : SENT), 1u, 0u);
            } else {                                                   // up: arriving at u from its child v -> u's next child, or on up, or the end
                int c = nb_index(w, u, v, du) + 1;
                if (c == pu) ++c;
                out = e2_pack(c < du ? 4u * u + c : (pu >= 0 ? 4u * u + pu : SENT), 0u, 1u);
            }
        }
    }
    w.e2[a] = out;
}
__global__ __launch_bounds__(256) void bfsE_rank2_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t a = blockIdx.x * 256 + threadIdx.x, SENT = 4u * (uint32_t)V;
    if (a >= SENT) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    unsigned long long p = w.e2[a];
#pragma unroll
    for (int it = 0; it < kEulerJumps; ++it) {
        const uint32_t s = e2_succ(p);
        if (s >= SENT) break;
        const unsigned long long q = w.e2[s];
        p = e2_pack(e2_succ(q), e2_down(p) + e2_down(q), e2_up(p) + e2_up(q));
    }
    w.e2[a] = p;
}
// preorder number and depth of every vertex from the suffix counts at its parent's arc towards it -> the preorder sequence (depth, vertex)
__global__ __launch_bounds__(256) void bfsE_place_kernel(int V, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y;
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t SENT = 4u * (uint32_t)V;
    uint32_t pre = 0u, dep = 0u;
    bool ok = v < (uint32_t)V;                                             // (no early return: the whole wave meets at the shuffles below)
    if (ok && v != 0u) {
        const int pk = w.pk[v];
        ok = pk >= 0;
        if (ok) {
            const uint32_t p = w.adj[4u * v + pk];
            const int j = p < (uint32_t)V ? nb_index(w, p, v, (int)min(w.deg[p], 4u)) : -1;
            ok = j >= 0;
            if (ok) {
                const unsigned long long x = w.e2[4u * p + j];
                const uint32_t down = e2_down(x), up = e2_up(x);           // arcs leading down / up from this one to the end, inclusive
                ok = e2_succ(x) >= SENT && down >= 1u && down <= (uint32_t)V - 1u && up <= (uint32_t)V - 1u;
                pre = (uint32_t)V - down;                                   // = (V - 1) - down + 1: down arcs up to and including this one
                dep = pre - ((uint32_t)V - 1u - up);                        // minus the up arcs before it
                ok = ok && dep >= 1u && dep <= pre;
            }
        }
        if (!ok) atomicOr(w.bad, 1);
    }
    if (ok) { w.key0[pre] = dep; w.val0[pre] = v; }                     // (a permutation when the input is a tree)
}
// ---- stable LSD radix sort of the (depth, vertex) pairs, kSortBits bits a pass, one wave per tile of kSortTile pairs -------------------
__device__ __forceinline__ unsigned long long same_digit_lanes(uint32_t digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < kSortBits; ++bit) {
        const unsigned long long m = __ballot((digit >> bit) & 1u);
        peers &= ((digit >> bit) & 1u) ? m : ~m;
    }
    return peers;
}
__global__ __launch_bounds__(64) void bfsE_sort_hist_kernel(int V, int pass, char* ws_base, size_t ws_stride) {
    __shared__ uint32_t h[kSortBuckets];
    const int b = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t* key = (pass & 1) ? w.key1 : w.key0;
    for (int i = lane; i < kSortBuckets; i += 64) h[i] = 0u;
    __syncthreads();
    for (int i = blk * kSortTile + lane; i < min((blk + 1) * kSortTile, V); i += 64) atomicAdd(&h[(key[i] >> (pass * kSortBits)) & (kSortBuckets - 1)], 1u);
    __syncthreads();
    for (int i = lane; i < kSortBuckets; i += 64) w.hist[(size_t)blk * kSortBuckets + i] = h[i];  // tile-major: coalesced here, in the scan and in the scatter
}
// places: thread d runs down bucket d's counts over the tiles (coalesced across the threads; the counts become the exclusive prefix inside
// the bucket), then the bucket totals are scanned: a pair's place = base[bucket] + prefix[tile][bucket] + its rank in the tile.
// (As one flat scan of the 512 x 60 counters in bucket-major order this kernel took 30 us: every load of a thread's run of entries
// touched 64 different lines, on one CU.)
__global__ __launch_bounds__(kLT) void bfsE_sort_scan_kernel(int V, int nblk, char* ws_base, size_t ws_stride) {
    __shared__ int part[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    int run = 0;
    if (tid < kSortBuckets)
        for (int blk0 = 0; blk0 < nblk; blk0 += 16) {                  // sixteen loads in flight, then their stores (the same array: the compiler
            uint32_t c[16];                                            // would otherwise order every load behind the previous store)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = blk0 + e < nblk ? w.hist[(size_t)(blk0 + e) * kSortBuckets + tid] : 0u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (blk0 + e < nblk) w.hist[(size_t)(blk0 + e) * kSortBuckets + tid] = (uint32_t)run;
                run += (int)c[e];
            }
        }
    int total;
    const int base = block_excl_scan(tid < kSortBuckets ? run : 0, part, total);
    if (tid < kSortBuckets) w.hist[(size_t)nblk * kSortBuckets + tid] = (uint32_t)base;
}
__global__ __launch_bounds__(64) void bfsE_sort_scatter_kernel(int V, int pass, char* ws_base, size_t ws_stride) {
    __shared__ uint32_t place[kSortBuckets];
    const int b = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x, nblk = gridDim.x;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    const uint32_t* key = (pass & 1) ? w.key1 : w.key0; const uint32_t* val = (pass & 1) ? w.val1 : w.val0;
    uint32_t* key_o = (pass & 1) ? w.key0 : w.key1; uint32_t* val_o = (pass & 1) ? w.val0 : w.val1;
    for (int i = lane; i < kSortBuckets; i += 64) place[i] = w.hist[(size_t)nblk * kSortBuckets + i] + w.hist[(size_t)blk * kSortBuckets + i];
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int end = min((blk + 1) * kSortTile, V);
    constexpr int NS = kSortTile / 64;
    uint32_t ks[NS], xs[NS];                                            // the whole tile's pairs first: their loads fly together
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int i = blk * kSortTile + 64 * q + lane;
        ks[q] = i < end ? key[i] : 0u; xs[q] = i < end ? val[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {                                      // strips in order: the sort is stable
        const int i = blk * kSortTile + 64 * q + lane;
        const bool valid = i < end;
        const uint32_t k = ks[q], x = xs[q];
        const uint32_t digit = (k >> (pass * kSortBits)) & (kSortBuckets - 1);
        const unsigned long long peers = same_digit_lanes(digit, valid);
        if (valid) {
            const uint32_t at = place[digit] + (uint32_t)__popcll(peers & lt);
            if (at < (uint32_t)V) { key_o[at] = k; val_o[at] = x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // everybody has read `place` before the group leaders advance it
        if (valid && (peers >> lane) == 1ull) place[digit] += (uint32_t)__popcll(peers);   // the highest lane of every group
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
// the sorted pairs -> nodev / nodep (what bfsL_index / bfsL_parent turn into the outputs) and the level offsets: depths are sorted, every
// depth up to the last occurs, so level d starts where the key changes to d
__global__ __launch_bounds__(256) void bfsE_fill_kernel(int V, int src, int* __restrict__ levels, char* ws_base, size_t ws_stride) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= V) return;
    const BfsLargeWs w = bfs_ws(ws_base, ws_stride, b, V);
    int* lv = levels + (int64_t)b * (V + 2);
    const bool bad = (*w.flag & 1) || *w.bad;
    const uint32_t* keys = src ? w.key1 : w.key0;
    const uint32_t v = bad ? 0u : (src ? w.val1 : w.val0)[p];
    const int pk = v < (uint32_t)V ? w.pk[v] : -1;
    w.nodev[p] = v < (uint32_t)V ? v : 0u;
    w.nodep[p] = (p == 0 || pk < 0) ? 0xffffffffu : w.adj[4u * v + pk];
    if (!bad) {
        const uint32_t d = keys[p];
        if (p == 0 || keys[p - 1] != d) { if (d < (uint32_t)V) lv[1 + d] = p; }
        if (p == V - 1) { lv[0] = (int)d + 1; if (d < (uint32_t)V) lv[2 + d] = V; }
    } else if (p == 0) lv[0] = -1;
    if (p == 0) {
        *w.nf = bad ? 0 : V;
        atomicOr(w.flag, 2);                            // nodev / nodep are plain (vertex, parent) arrays here, not the walk's grid words
    }
}

// developer hook (include/boxinst_hip_dev.h, bxi_dev_set_tree_level_walk): the level walk instead of the Euler tour / the doubling rounds
// (tests compare the two); not part of the production ABI
static std::atomic<int> g_level_walk{0};
void dev_set_tree_level_walk(int on) { g_level_walk.store(on ? 1 : 0, std::memory_order_relaxed); }
static inline bool level_walk() { return g_level_walk.load(std::memory_order_relaxed) != 0; }

int launch_bfs_large(const int* tree, int B, int V, int max_adj, int* si, int* sp, int* sc, int* levels, char* ws, hipStream_t s) {
    if (B > 65535) return BXI_ERR_BAD_SHAPE;
    const size_t stride = bfs_large_ws_bytes(V);
    const dim3 gv((unsigned)((V + 255) / 256), (unsigned)B), gz((unsigned)(((int64_t)V * max_adj + 255) / 256), (unsigned)B);
    BXI_LAUNCH("bfs_large_zero", s, bfsL_zero_kernel, gz, dim3(256), 0, s, V, max_adj, sc, ws, stride);
    BXI_LAUNCH("bfs_large_adj", s, bfsL_adj_kernel, gv, dim3(256), 0, s, tree, V, ws, stride);
    BXI_LAUNCH("bfs_large_sort", s, bfsL_sort_kernel, gv, dim3(256), 0, s, V, ws, stride);
    if (V <= kEulerMaxV && V >= 2 && !level_walk()) {
        const dim3 ga((unsigned)((4 * (size_t)V + 255) / 256), (unsigned)B);
        // a launch is only guaranteed to see what the PREVIOUS launch wrote: each of a thread's jumps then adds at least the span its target
        // had when the launch began, so a launch multiplies every span by at least kEulerJumps + 1 (by 2^kEulerJumps when it sees its own)
        int rounds = 0;
        for (long long span = 1; span < 2ll * V; span *= kEulerJumps + 1) ++rounds;
        BXI_LAUNCH("bfs_euler_succ1", s, bfsE_succ1_kernel, ga, dim3(256), 0, s, V, ws, stride);
        for (int r = 0; r < rounds; ++r) BXI_LAUNCH("bfs_euler_rank1", s, bfsE_rank1_kernel, ga, dim3(256), 0, s, V, ws, stride);
        BXI_LAUNCH("bfs_euler_parent", s, bfsE_parent_kernel, gv, dim3(256), 0, s, V, ws, stride);
        BXI_LAUNCH("bfs_euler_succ2", s, bfsE_succ2_kernel, ga, dim3(256), 0, s, V, ws, stride);
        for (int r = 0; r < rounds; ++r) BXI_LAUNCH("bfs_euler_rank2", s, bfsE_rank2_kernel, ga, dim3(256), 0, s, V, ws, stride);
        BXI_LAUNCH("bfs_euler_place", s, bfsE_place_kernel, gv, dim3(256), 0, s, V, ws, stride);
        int key_bits = 1;
        while ((1ll << key_bits) < (long long)V) ++key_bits;                     // depths are < V
        const int passes = (key_bits + kSortBits - 1) / kSortBits;
        const int nblk = (V + kSortTile - 1) / kSortTile;
        for (int pass = 0; pass < passes; ++pass) {
            BXI_LAUNCH("bfs_euler_sort_hist", s, bfsE_sort_hist_kernel, dim3(nblk, B), dim3(64), 0, s, V, pass, ws, stride);
            BXI_LAUNCH("bfs_euler_sort_scan", s, bfsE_sort_scan_kernel, dim3(B), dim3(kLT), 0, s, V, nblk, ws, stride);
            BXI_LAUNCH("bfs_euler_sort_scatter", s, bfsE_sort_scatter_kernel, dim3(nblk, B), dim3(64), 0, s, V, pass, ws, stride);
        }
        BXI_LAUNCH("bfs_euler_fill", s, bfsE_fill_kernel, gv, dim3(256), 0, s, V, passes & 1, levels, ws, stride);
    } else {
        const size_t lds = V <= kBfsMaskCap ? (size_t)(V + 15) / 16 * 16 : 0;
        if (lds > 48 * 1024) {
            static std::atomic<int> attr_set{0};
            if (!attr_set.load(std::memory_order_relaxed)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bfsL_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
                if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
                attr_set.store(1, std::memory_order_relaxed);
            }
        }
        BXI_LAUNCH("bfs_large_walk", s, bfsL_walk_kernel, dim3(B), dim3(kLT), lds, s, V, levels, ws, stride);
    }
    BXI_LAUNCH("bfs_large_index", s, bfsL_index_kernel, gv, dim3(256), 0, s, V, si, ws, stride);
    BXI_LAUNCH("bfs_large_parent", s, bfsL_parent_kernel, gv, dim3(256), 0, s, V, max_adj, sp, sc, ws, stride);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// refine: records {v0, v1, w, first child | count << 28} in global memory, parent in a second array
struct RefinePlaneL { const float* in; const float* pre_div; const float* pre_mul; float* up_sorted; float* down_sorted; float* down_vertex; };
struct RefineArgsL {
    RefinePlaneL pl[2];
    const float* edge_weight; const int* sorted_index; const int* sorted_child; const int* levels;
    float* out_vertex;
    int B, C, V, max_adj, n_planes;
    char* ws; size_t ws_stride;
};
constexpr int kPadL = 4;
constexpr int kWinL = 8192;            // records per LDS window of the leaf->root walk (128 KB); a power of two: slot = i & (kWinL - 1)
constexpr int kWinLv = 2048;           // levels per window at most (8 KB of level offsets)
static size_t refine_large_block_bytes(int V) { return up16(16 * (size_t)(V + kPadL)) + up16(4 * (size_t)V) + up16(16 * (size_t)(V + kPadL)) + 16 + 3 * up16(4 * (size_t)V); }
size_t refine_large_ws_bytes(int B, int C, int V) { return refine_large_block_bytes(V) * (size_t)(B > 0 ? B : 1) * (size_t)C; }

// per (graph, channel) block of the workspace: records, parents, the second buffer of the jump rounds, one flag word
struct RefineBlkL { float4* rec; uint32_t* parent; float4* tmp; int* bad; uint32_t* dep; uint32_t* lo0; uint32_t* lo1; };   // (no arrays: a runtime index would put the struct in scratch)
__device__ __forceinline__ RefineBlkL refine_blk(const RefineArgsL& a, int b, int ch) {
    char* wsb = a.ws + ((size_t)b * a.C + ch) * a.ws_stride;
    RefineBlkL r;
    r.rec = reinterpret_cast<float4*>(wsb);
    r.parent = reinterpret_cast<uint32_t*>(wsb + up16(16 * (size_t)(a.V + kPadL)));
    r.tmp = reinterpret_cast<float4*>(wsb + up16(16 * (size_t)(a.V + kPadL)) + up16(4 * (size_t)a.V));
    r.bad = reinterpret_cast<int*>(wsb + up16(16 * (size_t)(a.V + kPadL)) + up16(4 * (size_t)a.V) + up16(16 * (size_t)(a.V + kPadL)));
    char* more = reinterpret_cast<char*>(r.bad) + 16;                 // the depth-free leaf->root pass: depth and descendant-range starts
    r.dep = reinterpret_cast<uint32_t*>(more);
    r.lo0 = reinterpret_cast<uint32_t*>(more + up16(4 * (size_t)a.V));
    r.lo1 = reinterpret_cast<uint32_t*>(more + 2 * up16(4 * (size_t)a.V));
    return r;
}

// The refinement of a large tree is five kinds of passes.  Four of them touch every node independently -- staging, the preparation of
// the affine maps, the pointer-jumping rounds, the outputs -- and run ACROSS THE GPU (grid over the nodes x graph x channel; as one
// 1024-thread workgroup per (graph, channel) each pass over 60 800 nodes cost ~60 us, fifteen of them most of a 1.3 ms launch); only
// the leaf->root walk is a chain through the tree's depth and stays one workgroup per (graph, channel).

// ---- pass 1: stage the records (sorted order), the parents, and check that the ordering is bxi_bfs_forward_i32's ----------------------
__global__ __launch_bounds__(256) void refineL_stage_kernel(RefineArgsL a) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const RefineBlkL w = refine_blk(a, b, ch);
    const int* lv = a.levels + (int64_t)b * (V + 2);
    const int* si = a.sorted_index + (int64_t)b * V;
    const int* sc = a.sorted_child + (int64_t)b * V * a.max_adj;
    const float* ew = a.edge_weight + (int64_t)b * V;
    const int64_t cb = ((int64_t)b * a.C + ch) * V;
    const RefinePlaneL pl0 = a.pl[0], pl1 = a.pl[1];
    const bool two = a.n_planes > 1;
    if (i >= V + kPadL) return;
    if (i >= V) { w.rec[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    int bad = (i == 0 && lv[0] < 0) ? 1 : 0;                         // bxi_bfs_forward_i32 found the input not to be a connected grid tree
    const int p = si[i];
    float v0 = pl0.in ? pl0.in[cb + p] : 1.f;
    if (pl0.in && pl0.pre_div) v0 /= pl0.pre_div[(int64_t)b * V + p];
    if (pl0.in && pl0.pre_mul) v0 *= pl0.pre_mul[cb + p];
    float v1 = (two && pl1.in) ? pl1.in[cb + p] : (two ? 1.f : 0.f);
    if (two && pl1.in && pl1.pre_div) v1 /= pl1.pre_div[(int64_t)b * V + p];
    if (two && pl1.in && pl1.pre_mul) v1 *= pl1.pre_mul[cb + p];
    int c0 = 0, nc = 0;
    bool open = true;                                                // children = the leading positive slots
    for (int k = 0; k < 4; ++k) {
        const int c = sc[(size_t)i * a.max_adj + min(k, a.max_adj - 1)];
        open = open && k < a.max_adj && c > 0;
        if (open) {
            if (nc == 0) c0 = c; else if (c != c0 + nc) bad = 1;      // children must be contiguous (bxi_bfs_forward_i32 order)
            ++nc;
        }
    }
    if (a.max_adj > 4 && sc[(size_t)i * a.max_adj + 4] > 0) bad = 1;
    if (c0 + nc > V) { bad = 1; nc = 0; }
    w.rec[i] = make_float4(v0, v1, i ? ew[i] : 0.f /* weight[0] = 0 (refine.cu:38) */, __uint_as_float((uint32_t)c0 | ((uint32_t)nc << 28)));
    for (int k = 0; k < nc; ++k) w.parent[c0 + k] = (uint32_t)i;     // one writer per child
    if (i == 0) w.parent[0] = 0u;
    if (bad) atomicOr(w.bad, 1);                                      // (zeroed by launch_refine_large)
}

// ---- pass 2: leaf->root, U_i = x_i + sum_c w_c U_c (refine.cu:64-121): one workgroup per (graph, channel) ------------------------------
// The trees of a 200 x 304 map are ~1500 levels deep and ~40 nodes wide: level by level out of global memory every level costs a
// workgroup barrier and two dependent global round trips (~0.9 us).  Instead the node array is walked in WINDOWS of kWinL records
// held in LDS: the nodes of a level and their children (the next level) are neighbours in BFS order, so a window holds a run of
// consecutive levels; the workgroup loads it (coalesced), ONE wave walks its levels with no barrier and no global read in between
// (its LDS operations execute in order: tree_filter.hip's tree_up), writing every finished record through to global memory.  A level
// pair wider than the window falls back to the level-by-level form.
__global__ __launch_bounds__(kLT) void refineL_up_kernel(RefineArgsL a) {
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, V = a.V;
    const RefineBlkL w = refine_blk(a, b, ch);
    float4* rec = w.rec;
    const int* lv = a.levels + (int64_t)b * (V + 2);
    if (tid == 0 && *w.bad) {                                        // a foreign ordering fails loudly in the values
        float4 r = rec[0]; r.x *= __builtin_nanf(""); r.y *= __builtin_nanf(""); rec[0] = r;
    }
    __syncthreads();
    const int D = max(lv[0], 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char win_raw[];
    float4* win = reinterpret_cast<float4*>(win_raw);                           // [kWinL] record i lives in slot i % kWinL
    int* wlv = reinterpret_cast<int*>(win_raw + sizeof(float4) * kWinL);        // [kWinLv + 2] level offsets of the window's levels
    __shared__ int w_llo;
    for (int cur = D - 1; cur >= 0;) {
        // the window ends behind the children of level `cur` and starts at the lowest level whose first node still fits
        const int whi = cur + 1 <= D - 1 ? lv[3 + cur] : lv[2 + cur];
        const int bound = whi - kWinL;
        if (tid < 64) {                                                          // 64-ary search for the smallest level l <= cur with lv[1 + l] >= bound
            int a_ = max(0, cur - (kWinLv - 2)), b_ = cur;                       // (at most kWinLv levels per window)
            while (b_ > a_) {
                const int step = (b_ - a_ + 63) / 64;
                const int l = min(a_ + tid * step, b_);
                const unsigned long long m = __ballot(lv[1 + l] >= bound);
                const int k = m ? __ffsll((long long)m) - 1 : 63;
                const int nb = min(a_ + k * step, b_);
                a_ = k ? min(a_ + (k - 1) * step + 1, nb) : a_;
                b_ = nb;
            }
            if (tid == 0) w_llo = lv[1 + cur] >= bound ? b_ : cur + 1;         // cur + 1: not even level `cur` and its children fit
        }
        __syncthreads();
        const int llo = w_llo;
        if (llo > cur) {                                                         // workgroup-uniform fallback: this level straight from global memory
            const int lo = lv[1 + cur], hi = lv[2 + cur];
            for (int i = lo + tid; i < hi; i += kLT) {
                float4 r = rec[i];
                const uint32_t f = __float_as_uint(r.w);
                const int c0 = f & 0x0fffffffu, nc = f >> 28;
                for (int k = 0; k < nc; ++k) { const float4 c = rec[c0 + k]; r.x += c.x * c.z; r.y += c.y * c.z; }
                rec[i] = r;
            }
            __syncthreads();
            --cur;
            continue;
        }
        const int wlo = lv[1 + llo];
        for (int i = wlo + tid; i < whi; i += kLT) win[i & (kWinL - 1)] = rec[i];      // levels > cur are final in global memory
        for (int l = llo + tid; l <= cur + 1; l += kLT) wlv[l - llo] = lv[1 + l];
        __syncthreads();
        if (tid < 64) {
            const int lane = tid;
            const float* winf = reinterpret_cast<const float*>(win);
            int lo = wlv[cur - llo], hi = wlv[cur - llo + 1];
            // the child-range word of a node never changes: the next level's words are fetched while this level is computed, so
            // that a level costs ONE dependent LDS round trip (own values + children together), not two
            uint32_t fpre = __float_as_uint(winf[4 * ((lo + lane) & (kWinL - 1)) + 3]);
            for (int l = cur; l >= llo; --l) {
                const int nlo = l > llo ? wlv[l - 1 - llo] : lo;
                const uint32_t fnext = __float_as_uint(winf[4 * ((nlo + lane) & (kWinL - 1)) + 3]);
                bool first = true;
                for (int i = lo + lane; i < hi; i += 64) {
                    const uint32_t f = first ? fpre : __float_as_uint(winf[4 * (i & (kWinL - 1)) + 3]);
                    first = false;
                    const int c0 = f & 0x0fffffffu, nc = f >> 28;
                    float4 r = win[i & (kWinL - 1)];
                    float4 chd[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) chd[k] = win[(c0 + k) & (kWinL - 1)];  // unconditional
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(chd[k].x), "+v"(chd[k].y), "+v"(chd[k].z));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        r.x = k < nc ? r.x + chd[k].x * chd[k].z : r.x;
                        r.y = k < nc ? r.y + chd[k].y * chd[k].z : r.y;
                    }
                    win[i & (kWinL - 1)] = r;
                    rec[i] = r;                                                  // written through; nobody reads it before the next barrier
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the level's LDS writes before the next level's reads (one wave: in order)
                fpre = fnext; hi = lo; lo = nlo;
            }
        }
        __syncthreads();
        cur = llo - 1;
    }
}

// ---- pass 2, depth-free form: U = (I - A)^-1 x with A[parent, child] = w_child, as (I + A)(I + A^2)(I + A^4)... applied to x ----------------
// A^(2^k) links every node to its 2^k-th ancestor with the product of the weights on the way, so round k is
//     S_{k+1}[i] = S_k[i] + sum over the descendants j of i exactly 2^k levels down of P_k[j] S_k[j]
// (S_k[i] = the terms of U_i from descendants less than 2^k levels down; P_k[j] = product of the 2^k weights above j), together with the
// usual doubling of (ancestor, product).  In BFS order the descendants of i at a given distance are a CONTIGUOUS range of one level, and
// the ranges of the nodes of a level tile it in order: lo_k[i] = first position, 2^k levels down, whose 2^k-th ancestor is >= i; the range
// ends at lo_k[i + 1] (or the level's end), and lo_{k+1}[i] = lo_k[lo_k[i]] -- no search after the first round.  ceil(log2 D) <= ceil(log2 V)
// grid-wide rounds instead of D dependent steps of one wave (1 720 at 200 x 304: 0.44 ms a launch); a sum runs over its range in position
// order, so the result is run-to-run identical; its association differs from refine.cu's child-by-child order (rounding only).
__device__ __forceinline__ int rounds_for_depth(const RefineArgsL& a, int b) {      // smallest n with 2^n >= D: what a tree of D levels needs
    const int D = max(a.levels[(int64_t)b * (a.V + 2)], 0);
    int n = 0;
    while ((1ll << n) < (long long)D) ++n;
    return n;
}
constexpr uint32_t kNoAnc = 0xffffffffu;
__device__ __forceinline__ int level_end(const int* lv, int D, int t, int V) { return t < D ? lv[2 + t] : V; }
__global__ __launch_bounds__(256) void refineL_upd_init_kernel(RefineArgsL a) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const RefineBlkL w = refine_blk(a, b, ch);
    const int* lv = a.levels + (int64_t)b * (V + 2);
    const int D = max(lv[0], 0);
    float4 r = w.rec[i];
    if (i == 0 && *w.bad) { r.x *= __builtin_nanf(""); r.y *= __builtin_nanf(""); }     // a foreign ordering fails loudly in the values
    int d = 0;                                                       // depth: the last level that starts at or before i
    for (int lo = 0, hi = D; lo < hi;) { const int mid = (lo + hi + 1) >> 1; if (mid < D && lv[1 + mid] <= i) { lo = mid; d = mid; } else hi = mid - 1; }
    w.dep[i] = (uint32_t)d;
    uint32_t first = (uint32_t)V;                                    // lo_0: the first child position of i or of the next node that has one
    if (d + 1 < D) {
        int lo = lv[2 + d], hi = level_end(lv, D, d + 1, V);         // level d + 1; parents are non-decreasing along it
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (w.parent[mid] >= (uint32_t)i) hi = mid; else lo = mid + 1; }
        first = (uint32_t)lo;
    }
    w.lo0[i] = first;
    r.w = __uint_as_float(i ? w.parent[i] : kNoAnc);
    w.rec[i] = r;
}
constexpr int kUpdCoop = 24;           // a range longer than this is summed by the whole wave
__global__ __launch_bounds__(256) void refineL_upd_round_kernel(RefineArgsL a, int k, int flip) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = i < V;                                        // (no early return: the wave sums long ranges together)
    const RefineBlkL w = refine_blk(a, b, ch);
    const int* lv = a.levels + (int64_t)b * (V + 2);
    const int D = max(lv[0], 0);
    const float4* src = flip ? w.tmp : w.rec;
    float4* dst = flip ? w.rec : w.tmp;
    const uint32_t* lo_s = flip ? w.lo1 : w.lo0;
    uint32_t* lo_d = flip ? w.lo0 : w.lo1;
    const long long step = 1ll << k;
    if (step >= (long long)D) return;                               // nothing is that far apart: the result stays where round ceil(log2 D) - 1 left it
    float4 r = live ? src[i] : make_float4(0.f, 0.f, 0.f, __uint_as_float(kNoAnc));
    const int d = live ? (int)w.dep[i] : 0;
    const long long t = (long long)d + step;
    const bool has = live && t < (long long)D;
    uint32_t lo_next = (uint32_t)V;
    int q0 = 0, q1 = 0;
    if (has) {
        const int t_end = level_end(lv, D, (int)t, V);
        q0 = (int)lo_s[i];
        q1 = i + 1 < level_end(lv, D, d, V) ? (int)lo_s[i + 1] : t_end;
        // the range 2^(k+1) levels down starts where the range of the first node of this one starts
        const long long t2 = (long long)d + 2 * step;
        if (t2 < (long long)D) lo_next = q0 < t_end ? lo_s[q0] : (uint32_t)level_end(lv, D, (int)t2, V);
    }
    float sx = 0.f, sy = 0.f;
    const bool big = q1 - q0 > kUpdCoop;
    if (!big)
        for (int q = q0; q < q1; q += 8) {                           // eight loads in flight (one at a time, a range is one round trip per node)
            float4 c[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) c[e] = src[min(q + e, q1 - 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (q + e < q1) { sx += c[e].z * c[e].x; sy += c[e].z * c[e].y; }
        }
    // the few long ranges of the late rounds (a node near the root owns a whole level slice): lane l takes nodes l, l + 64, ...; the
    // partial sums meet in a butterfly -- a fixed order, as everything here
    for (unsigned long long m = __ballot(big); m; m &= m - 1) {
        const int L = __ffsll((long long)m) - 1;
        const int a0 = __shfl(q0, L, kWave), a1 = __shfl(q1, L, kWave);
        float px = 0.f, py = 0.f;
        for (int q = a0 + lane; q < a1; q += 64) { const float4 c = src[q]; px += c.z * c.x; py += c.z * c.y; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { px += __shfl_xor(px, off, kWave); py += __shfl_xor(py, off, kWave); }
        if (lane == L) { sx = px; sy = py; }
    }
    if (!live) return;
    r.x += sx; r.y += sy;
    const uint32_t an = __float_as_uint(r.w);
    if (an != kNoAnc) { const float4 q = src[an]; r.z *= q.z; r.w = q.w; } else r.z = 0.f;
    dst[i] = r;
    lo_d[i] = lo_next;
}

// ---- pass 3: U of both planes out (sorted order), and the affine maps D_c = A_c + B_c D_anc(c), A = U (1 - w^2), B = w (refine.cu:17-62) ----
__global__ __launch_bounds__(256) void refineL_prep_kernel(RefineArgsL a, int doubled) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const RefineBlkL w = refine_blk(a, b, ch);
    const int64_t cb = ((int64_t)b * a.C + ch) * V;
    const int from_tmp = doubled ? (rounds_for_depth(a, b) & 1) : 0;   // the depth-free pass ran that many effective rounds, rec -> tmp -> rec ...
    float4 r = (from_tmp ? w.tmp : w.rec)[i];
    r.z = i ? a.edge_weight[(int64_t)b * V + i] : 0.f;              // (the depth-free pass turned .z into a product of weights)
    for (int q = 0; q < 2; ++q) {
        const RefinePlaneL& pl = q ? a.pl[1] : a.pl[0];
        const bool per_tree = pl.in == nullptr;
        if (q >= a.n_planes || !pl.up_sorted || (per_tree && ch != 0)) continue;
        pl.up_sorted[(per_tree ? (int64_t)b * V : cb) + i] = q ? r.y : r.x;
    }
    const float wt = r.z, aa = 1.f - wt * wt;
    w.rec[i] = i ? make_float4(r.x * aa, r.y * aa, wt, __uint_as_float(w.parent[i])) : make_float4(r.x, r.y, 0.f, __uint_as_float(0xffffffffu));
}

// ---- pass 4 (x ceil(log2 V)): one pointer-jumping round, src -> dst (a resolved node is copied) -----------------------------------------
__global__ __launch_bounds__(256) void refineL_jump_kernel(RefineArgsL a, int k) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    if (k >= rounds_for_depth(a, b)) return;                         // every node is resolved: the launch count is fixed on the host for any depth
    const int flip = k & 1;
    const RefineBlkL w = refine_blk(a, b, ch);
    const float4* src = flip ? w.tmp : w.rec;
    float4* dst = flip ? w.rec : w.tmp;
    const float4 r = src[i];
    const uint32_t an = __float_as_uint(r.w);
    if (an == 0xffffffffu) { dst[i] = r; return; }
    const float4 q = src[an];
    dst[i] = make_float4(r.x + r.z * q.x, r.y + r.z * q.y, r.z * q.z, q.w);
}

// ---- pass 5: outputs ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refineL_out_kernel(RefineArgsL a) {
    const int b = blockIdx.y, ch = blockIdx.z, V = a.V;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const int flip = rounds_for_depth(a, b) & 1;
    const RefineBlkL w = refine_blk(a, b, ch);
    const int64_t cb = ((int64_t)b * a.C + ch) * V;
    const float4 d = (flip ? w.tmp : w.rec)[i];
    const int p = a.sorted_index[(int64_t)b * V + i];
    for (int q = 0; q < 2; ++q) {
        const RefinePlaneL& pl = q ? a.pl[1] : a.pl[0];
        if (q >= a.n_planes) continue;
        const bool per_tree = pl.in == nullptr;
        if (per_tree && ch != 0) continue;
        const int64_t ob = per_tree ? (int64_t)b * V : cb;
        if (pl.down_sorted) pl.down_sorted[ob + i] = q ? d.y : d.x;
        if (pl.down_vertex) pl.down_vertex[ob + p] = q ? d.y : d.x;
    }
    if (a.out_vertex) a.out_vertex[cb + p] = d.y / d.x;
}

__global__ void refineL_clear_kernel(RefineArgsL a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < a.B * a.C) *refine_blk(a, i / a.C, i % a.C).bad = 0;
}

int launch_refine_large(const void* planes2 /* RefinePlaneL[2] layout */, const float* edge_weight, const int* sorted_index,
                        const int* sorted_child, const int* levels, float* out_vertex, int B, int C, int V, int max_adj, int n_planes,
                        char* ws, hipStream_t s) {
    RefineArgsL a;
    const RefinePlaneL* p = reinterpret_cast<const RefinePlaneL*>(planes2);
    a.pl[0] = p[0]; a.pl[1] = p[1];
    a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child; a.levels = levels;
    a.out_vertex = out_vertex; a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.n_planes = n_planes;
    a.ws = ws; a.ws_stride = refine_large_block_bytes(V);
    if (B > 65535 || C > 65535) return BXI_ERR_BAD_SHAPE;
    const dim3 over_nodes((unsigned)((V + kPadL + 255) / 256), (unsigned)B, (unsigned)C);
    BXI_LAUNCH("tree_refine_large_clear", s, refineL_clear_kernel, dim3((unsigned)((B * C + 63) / 64)), dim3(64), 0, s, a);
    BXI_LAUNCH("tree_refine_large_stage", s, refineL_stage_kernel, over_nodes, dim3(256), 0, s, a);
    int doubled = 0;
    if (!level_walk()) {
        const dim3 nodes((unsigned)((V + 255) / 256), (unsigned)B, (unsigned)C);
        int up_rounds = 0;
        while ((1ll << up_rounds) < (long long)V) ++up_rounds;       // the depth is device data; rounds beyond it copy
        BXI_LAUNCH("tree_refine_large_up_init", s, refineL_upd_init_kernel, nodes, dim3(256), 0, s, a);
        for (int k = 0; k < up_rounds; ++k) BXI_LAUNCH("tree_refine_large_up_round", s, refineL_upd_round_kernel, nodes, dim3(256), 0, s, a, k, k & 1);
        doubled = 1;
    } else {
        const size_t lds = sizeof(float4) * kWinL + sizeof(int) * (kWinLv + 2);
        static std::atomic<int> attr_set{0};
        if (!attr_set.load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(refineL_up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
            attr_set.store(1, std::memory_order_relaxed);
        }
        BXI_LAUNCH("tree_refine_large_up", s, refineL_up_kernel, dim3(B, C), dim3(kLT), lds, s, a);
    }
    BXI_LAUNCH("tree_refine_large_prep", s, refineL_prep_kernel, over_nodes, dim3(256), 0, s, a, doubled);
    // the depth of the tree is device data: ceil(log2 V) rounds resolve any depth <= V (a resolved node is only copied)
    int rounds = 0;
    while ((1 << rounds) < V) ++rounds;
    for (int k = 0; k < rounds; ++k) BXI_LAUNCH("tree_refine_large_jump", s, refineL_jump_kernel, over_nodes, dim3(256), 0, s, a, k);
    BXI_LAUNCH("tree_refine_large_out", s, refineL_out_kernel, over_nodes, dim3(256), 0, s, a);
    return check_launch();
}

}  // namespace bxi
