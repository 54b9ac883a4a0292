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

// One wave walks a tree level by level through LDS.  Its LDS operations execute in program order, so between two levels
// nothing has to be waited for -- neither the LDS writes just issued nor the global stores of the level (a workgroup-scope
// fence waits for those too: ~1 us per level, hundreds of levels); only the compiler has to be held back
// (`asm volatile("" ::: "memory")`).

constexpr int kTfThreads = 1024;   // bfs / refine workgroups: one wave walks, all of them stage the inputs and write the outputs
constexpr int kTfMaxV = 10200;     // LDS-resident vertex limit (96x96 = 9216 in the reference's _scale_target)

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
// The walk (one wave, ~600 dependent levels at 96x96) touches LDS only: a position holds (vertex | parent vertex << 16),
// a level reads its nodes and their 4 neighbour slots, ranks the non-parent neighbours with ballots and writes them
// behind the nodes found so far (invalid slots go to a trash word: no divergent branches).  Level offsets collect in a
// register (lane = level mod 64).  The outputs -- sorted_index, sorted_parent (positions), sorted_child -- are written
// afterwards by the whole workgroup from the LDS arrays.
__global__ __launch_bounds__(kTfThreads) void bfs_kernel(const int* __restrict__ tree, int V, int max_adj, int* __restrict__ sorted_index,
                                                  int* __restrict__ sorted_parent, int* __restrict__ sorted_child,
                                                  int* __restrict__ levels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bfs_raw[];
    u16* adj = reinterpret_cast<u16*>(bfs_raw);                          // [V][4]
    uint32_t* node = reinterpret_cast<uint32_t*>(adj + (size_t)V * 4);   // [V+1] vertex | parent vertex << 16; [V] = trash
    unsigned char* deg = reinterpret_cast<unsigned char*>(node + V + 1); // [V]
    __shared__ int n_found;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* ed = tree + (int64_t)b * (V - 1) * 2;
    int* s_index = sorted_index + (int64_t)b * V;
    int* s_parent = sorted_parent + (int64_t)b * V;
    int* s_child = sorted_child + (int64_t)b * V * max_adj;
    int* lv = levels + (int64_t)b * (V + 2);
    unsigned int* deg32 = reinterpret_cast<unsigned int*>(deg);
    for (int i = tid; i < (V + 3) / 4; i += kTfThreads) deg32[i] = 0u;
    for (int i = tid; i < V * max_adj; i += kTfThreads) s_child[i] = 0;
    __syncthreads();
    // adjacency (degree <= 4): slots by byte-wise LDS atomics on the packed degree words
    for (int e = tid; e < V - 1; e += kTfThreads) {
        const int u = ed[2 * e], v = ed[2 * e + 1];
        const unsigned su = (atomicAdd(&deg32[u >> 2], 1u << (8 * (u & 3))) >> (8 * (u & 3))) & 0xffu;
        const unsigned sv = (atomicAdd(&deg32[v >> 2], 1u << (8 * (v & 3))) >> (8 * (v & 3))) & 0xffu;
        if (su < 4) adj[u * 4 + su] = (u16)v;
        if (sv < 4) adj[v * 4 + sv] = (u16)u;
    }
    __syncthreads();
    for (int v = tid; v < V; v += kTfThreads) {                // arrival order of the atomics -> ascending neighbour ids
        const int d = min((int)deg[v], 4);
        u16 a[4];
        for (int k = 0; k < 4; ++k) a[k] = k < d ? adj[v * 4 + k] : (u16)0xffff;
        for (int i = 1; i < 4; ++i) for (int j = i; j > 0 && a[j - 1] > a[j]; --j) { const u16 t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; }
        for (int k = 0; k < 4; ++k) adj[v * 4 + k] = a[k];
    }
    __syncthreads();
    if (tid < 64) {                                     // one wave walks the levels (no barrier, no wait between levels)
        const int lane = tid;
        if (lane == 0) node[0] = 0u | (0xffffu << 16);
        int lo = 0, hi = 1, n = 1, depth = 0;
        int offs = 0;                                   // lane k: off[64 c + k] of the chunk c being filled (off[0] = 0)
        while (lo < hi) {
            for (int base = lo; base < hi; base += 64) {
                const int i = base + lane;
                const bool act = i < hi;
                const uint32_t nd = node[min(i, V)];
                const uint32_t cur = act ? (nd & 0xffffu) : 0u, par = nd >> 16;
                const u64 a4 = *reinterpret_cast<const u64*>(adj + cur * 4);        // the 4 neighbour slots in one read
                uint32_t nb[4]; int rank[4]; bool ok[4];
                int nch = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nb[k] = (uint32_t)(a4 >> (16 * k)) & 0xffffu;
                    ok[k] = act && nb[k] != 0xffffu && nb[k] != par;
                    rank[k] = nch;
                    nch += ok[k] ? 1 : 0;
                }
                // exclusive prefix sum of nch (0..4) over the wave from three ballots: no LDS traffic
                const u64 b0 = __ballot(nch & 1), b1 = __ballot(nch & 2), b2 = __ballot(nch & 4);
                const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
                const int excl = __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below);
                const int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
                const int pos = n + excl;
#pragma unroll
                for (int k = 0; k < 4; ++k) node[ok[k] ? pos + rank[k] : V] = nb[k] | (cur << 16);
                n += total;
                asm volatile("" ::: "memory");          // a wave's LDS operations execute in order: compiler barrier only
            }
            ++depth;
            if ((depth & 63) == 0) lv[1 + depth - 64 + lane] = offs;                 // chunk complete
            offs = lane == (depth & 63) ? hi : offs;                                 // off[depth] = hi
            lo = hi; hi = n;
        }
        if (lane <= (depth & 63)) lv[1 + (depth & ~63) + lane] = offs;
        if (lane == 0) { lv[0] = depth; n_found = n; }
    }
    __syncthreads();
    // ---- outputs -------------------------------------------------------------------------------------------------------
    const int nf = n_found;                             // < V only for a disconnected input
    u16* pos_of = adj;                                  // the adjacency is no longer needed
    for (int p = tid; p < V; p += kTfThreads) {
        const uint32_t nd = p < nf ? node[p] : 0u;
        s_index[p] = (int)(nd & 0xffffu);
        if (p < nf) pos_of[nd & 0xffffu] = (u16)p;
    }
    __syncthreads();
    for (int p = tid; p < V; p += kTfThreads) {
        if (p == 0 || p >= nf) { s_parent[p] = 0; continue; }
        const uint32_t pv = node[p] >> 16;
        const int pp = pos_of[pv];
        s_parent[p] = pp;
        int k = 0;                                      // rank among the (contiguous) siblings
        while (k < 3 && p - k - 1 >= 1 && (node[p - k - 1] >> 16) == pv) ++k;
        if (k < max_adj) s_child[pp * max_adj + k] = p;
    }
}

// ---------------------------------------------------------------------------------------------------
// refine: one workgroup per (tree, channel); a traversal = leaf->root aggregation then root->leaf propagation.  The walk
// is a chain of ~2 x 600 dependent steps at 96x96 executed by ONE wave, so what counts is the number of instructions and
// LDS round trips per level.  A node is one 16-byte LDS record
//     { value of plane 0, value of plane 1, edge weight to the parent, first child | count << 14 | parent << 17 }
// and two value planes ride the same walk:
//   forward            plane 0 = ones (weight_sum), plane 1 = the channel
//   backward (weight)  plane 0 = g / weight_sum,    plane 1 = g / weight_sum * feature_out
// Level offsets are read 64 at a time into a register (lane k = off[64 c + k]) and picked with readlane; the chunk
// after the current one is already in flight.  DIR = -1 reads off[first], off[first-1], ...; DIR = +1 ascends.
template <int DIR>
struct LevelReader {
    const int* off; int D, lane, idx, chunk, cur, nxt;
    __device__ __forceinline__ int fetch(int c) const { return (c >= 0 && c * 64 <= D) ? off[min(c * 64 + lane, D)] : 0; }
    __device__ __forceinline__ void init(const int* o, int d, int l, int first) {
        off = o; D = d; lane = l; idx = first; chunk = first >> 6;
        cur = fetch(chunk); nxt = fetch(chunk + DIR);
    }
    __device__ __forceinline__ int next() {              // the offset at idx, then idx += DIR (clamped reads past the ends)
        const int j = min(max(idx, 0), D);
        if ((j >> 6) != chunk) { chunk = j >> 6; cur = nxt; nxt = fetch(chunk + DIR); }
        idx += DIR;
        return __builtin_amdgcn_readlane(cur, __builtin_amdgcn_readfirstlane(j & 63));
    }
};

constexpr int kRecPad = 4;       // records past the last node: the 4 child reads of a node are unconditional

// Leaf->root aggregation U_i = x_i + sum_c w_c U_c (refine.cu:64-121; thread = parent, children contiguous), in place.
// Called by ONE wave: its LDS operations execute in order, so nothing is waited for between levels (compiler barrier
// only), and the child ranges of the NEXT level's nodes are fetched while this level is computed -- one LDS round trip
// per level.  The walk issues no global memory operation except the level-offset chunks; the workgroup writes all
// outputs afterwards, coalesced.
__device__ __forceinline__ void tree_up(float4* __restrict__ rec, const int* __restrict__ off, int D, int V, int lane) {
    float* recf = reinterpret_cast<float*>(rec);
    {
        LevelReader<-1> rd; rd.init(off, D, lane, D);
        int hi = rd.next(), lo = rd.next();
        uint32_t fpre = __float_as_uint(recf[4 * min(lo + lane, V - 1) + 3]);
        for (int l = D - 1; l >= 0; --l) {
            const int nlo = rd.next();
            const uint32_t fnext = __float_as_uint(recf[4 * min(nlo + lane, V - 1) + 3]);
            bool first = true;
            for (int i = lo + lane; i < hi; i += 64) {
                const uint32_t f = first ? fpre : __float_as_uint(recf[4 * i + 3]);
                first = false;
                const int c0 = f & 0x3fffu, nc = (f >> 14) & 7u;
                const float2 own = *reinterpret_cast<const float2*>(recf + 4 * i);
                float4 ch[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) ch[k] = rec[c0 + k];
                // all four reads are issued together and unconditionally (left alone, the compiler sinks the first one into
                // an `nc != 0` branch: a second dependent round trip per level)
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ch[k].x), "+v"(ch[k].y), "+v"(ch[k].z));
                float a0 = own.x, a1 = own.y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0 = k < nc ? a0 + ch[k].x * ch[k].z : a0;
                    a1 = k < nc ? a1 + ch[k].y * ch[k].z : a1;
                }
                *reinterpret_cast<float2*>(recf + 4 * i) = make_float2(a0, a1);
            }
            asm volatile("" ::: "memory");
            fpre = fnext; hi = lo; lo = nlo;
        }
    }
}

// Root->leaf propagation D_0 = U_0, D_c = U_c (1 - w_c^2) + D_parent w_c (refine.cu:17-62) WITHOUT walking the levels: the
// recurrence is affine in D_parent, D_c = A_c + B_c D_anc(c) with A = U (1 - w^2), B = w, anc = parent, and affine maps
// compose -- every round each node replaces (A, B, anc) by its composition with its ancestor's map (pointer jumping:
// A += B A_anc, B *= B_anc, anc = anc(anc)), so a node of depth d is resolved against the root after ceil(log2(d+1)) rounds:
// ~10 rounds of the whole workgroup over all nodes instead of ~600 dependent steps of one wave (77 -> 12 us at 96x96).
// The products of weights only ever shrink (w <= 1), so an underflow just means "no longer depends on the ancestor".
// Records come in as { U0, U1, w, links } and leave as { D0, D1, -, - }; read and write phases of a round are separated by
// barriers (a node's ancestor is rewritten by another thread in the same round).
constexpr uint32_t kJumpResolved = 0xffffffffu;
constexpr int kJumpPer = (kTfMaxV + kTfThreads - 1) / kTfThreads;

__device__ __forceinline__ void tree_down_jump(float4* __restrict__ rec, int D, int V, int tid) {
    for (int i = tid; i < V; i += kTfThreads) {
        const float4 r = rec[i];
        const float w = r.z, a = 1.f - w * w;
        const uint32_t par = __float_as_uint(r.w) >> 17;
        rec[i] = i ? make_float4(r.x * a, r.y * a, w, __uint_as_float(par)) : make_float4(r.x, r.y, 0.f, __uint_as_float(kJumpResolved));
    }
    __syncthreads();
    const int rounds = D > 1 ? 32 - __clz(D - 1) : 0;               // deepest node: depth D - 1
    for (int k = 0; k < rounds; ++k) {
        float mx[kJumpPer], my[kJumpPer], mb[kJumpPer];
        float4 an[kJumpPer];
        uint32_t ma[kJumpPer];
#pragma unroll
        for (int j = 0; j < kJumpPer; ++j) {
            const int i = tid + j * kTfThreads;
            ma[j] = kJumpResolved;
            if (i < V) {
                const float4 r = rec[i];
                mx[j] = r.x; my[j] = r.y; mb[j] = r.z; ma[j] = __float_as_uint(r.w);
                if (ma[j] != kJumpResolved) an[j] = rec[ma[j]];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kJumpPer; ++j) {
            const int i = tid + j * kTfThreads;
            if (ma[j] != kJumpResolved)
                rec[i] = make_float4(mx[j] + mb[j] * an[j].x, my[j] + mb[j] * an[j].y, mb[j] * an[j].z, an[j].w);
        }
        __syncthreads();
    }
}

struct RefinePlane {
    const float* in;            // [B,C,V] vertex order; null = ones (a per-tree plane: outputs are [B,V], written by channel 0)
    const float* pre_div;       // [B,V] vertex order or null: in / pre_div      (grad_out / weight_sum, refine.cu:251)
    const float* pre_mul;       // [B,C,V] vertex order or null: ... * pre_mul   (feature_grad = grad_out_norm * feature_out, :324)
    float* up_sorted;           // or null : U
    float* down_sorted;         // or null : D in sorted order
    float* down_vertex;         // or null : D in vertex order
};

struct RefineArgs {
    RefinePlane pl[2];
    const float* edge_weight;   // [B,V] sorted order
    const int* sorted_index;    // [B,V]
    const int* sorted_child;    // [B,V,max_adj]
    const int* levels;          // [B,V+2]
    float* out_vertex;          // [B,C,V] or null : D(plane 1) / D(plane 0) in vertex order (feature_out)
    int B, C, V, max_adj;
    int n_planes;               // 1 or 2
};

__global__ __launch_bounds__(kTfThreads) void tree_refine_kernel(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_raw[];
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, V = a.V;
    float4* rec = reinterpret_cast<float4*>(tf_raw);
    float* recf = reinterpret_cast<float*>(tf_raw);
    uint32_t* recu = reinterpret_cast<uint32_t*>(tf_raw);
    const int* lv = a.levels + (int64_t)b * (V + 2);
    const int* si = a.sorted_index + (int64_t)b * V;
    const int* sc = a.sorted_child + (int64_t)b * V * a.max_adj;
    const float* ew = a.edge_weight + (int64_t)b * V;
    const int64_t cb = ((int64_t)b * a.C + ch) * V;
    // ---- topology + inputs -> records ----------------------------------------------------------------------------------
    // kStage nodes per thread at a time, every independent load of the batch issued before the gathers that depend on them
    // (node by node, each of a thread's ~36 nodes costs a chain of two or three global round trips: ~100 us per launch)
    constexpr int kStage = 5;
    int bad = 0;
    const RefinePlane pl0 = a.pl[0], pl1 = a.pl[1];
    const bool two = a.n_planes > 1;
    for (int base = tid; base < V + kRecPad; base += kTfThreads * kStage) {
        int p[kStage], ch4[kStage][4];
        float w[kStage];
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
            const int i = min(base + j * kTfThreads, V - 1);
            p[j] = si[i];
            w[j] = ew[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) ch4[j][k] = sc[i * a.max_adj + min(k, a.max_adj - 1)];
        }
        float x0[kStage], x1[kStage], d0[kStage], d1[kStage], m0[kStage], m1[kStage];
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
            x0[j] = pl0.in ? pl0.in[cb + p[j]] : 1.f;
            d0[j] = (pl0.in && pl0.pre_div) ? pl0.pre_div[(int64_t)b * V + p[j]] : 1.f;
            m0[j] = (pl0.in && pl0.pre_mul) ? pl0.pre_mul[cb + p[j]] : 1.f;
            x1[j] = (two && pl1.in) ? pl1.in[cb + p[j]] : (two ? 1.f : 0.f);
            d1[j] = (two && pl1.in && pl1.pre_div) ? pl1.pre_div[(int64_t)b * V + p[j]] : 1.f;
            m1[j] = (two && pl1.in && pl1.pre_mul) ? pl1.pre_mul[cb + p[j]] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
            const int i = base + j * kTfThreads;
            if (i >= V + kRecPad) break;
            if (i >= V) { rec[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            float v0 = x0[j], v1 = x1[j];
            if (pl0.in && pl0.pre_div) v0 /= d0[j];
            if (pl0.in && pl0.pre_mul) v0 *= m0[j];
            if (two && pl1.in && pl1.pre_div) v1 /= d1[j];
            if (two && pl1.in && pl1.pre_mul) v1 *= m1[j];
            int c0 = 0, nc = 0;
            bool open = true;                                        // children = the leading positive slots
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = ch4[j][k];
                open = open && k < a.max_adj && c > 0;
                if (open) {
                    if (nc == 0) c0 = c; else if (c != c0 + nc) bad = 1;   // children must be contiguous (bxi_bfs_forward_i32 order)
                    ++nc;
                }
            }
            if (a.max_adj > 4 && sc[i * a.max_adj + 4] > 0) bad = 1;  // more than 4 children: not a tree on a grid
            if (c0 + nc > V) { bad = 1; nc = 0; }
            rec[i] = make_float4(v0, v1, i ? w[j] : 0.f /* weight[0] = 0 (refine.cu:38) */, __uint_as_float((uint32_t)c0 | ((uint32_t)nc << 14)));
        }
    }
    const float poison = __syncthreads_or(bad) ? __builtin_nanf("") : 1.f;   // a foreign ordering fails loudly in the values
    for (int i = tid; i < V; i += kTfThreads) {                              // parent position into the children's records
        const uint32_t f = recu[4 * i + 3];
        const int c0 = f & 0x3fffu, nc = (f >> 14) & 7u;
        for (int k = 0; k < nc; ++k) recu[4 * (c0 + k) + 3] |= (uint32_t)i << 17;   // one writer per child
        if (i == 0) { recf[0] *= poison; recf[1] *= poison; }
    }
    __syncthreads();
    // ---- leaf->root: the walk (one wave) -------------------------------------------------------------------------------
    const int D = lv[0];
    if (tid < 64) tree_up(rec, lv + 1, D, V, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                     // U of both planes (sorted order), coalesced
        const RefinePlane& pl = q ? pl1 : pl0;
        const bool per_tree = pl.in == nullptr;
        if (q >= a.n_planes || !pl.up_sorted || (per_tree && ch != 0)) continue;
        float* uo = pl.up_sorted + (per_tree ? (int64_t)b * V : cb);
        for (int i = tid; i < V; i += kTfThreads) uo[i] = recf[4 * i + q];
    }
    // ---- root->leaf: pointer jumping (the whole workgroup) --------------------------------------------------------------
    tree_down_jump(rec, D, V, tid);
    // ---- results -------------------------------------------------------------------------------------------------------
    for (int base = tid; base < V; base += kTfThreads * kStage) {
        int p[kStage];
#pragma unroll
        for (int j = 0; j < kStage; ++j) p[j] = si[min(base + j * kTfThreads, V - 1)];
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
            const int i = base + j * kTfThreads;
            if (i >= V) break;
            const float2 d = *reinterpret_cast<const float2*>(recf + 4 * i);   // { D0, D1 }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const RefinePlane& pl = q ? pl1 : pl0;
                if (q >= a.n_planes) continue;
                const bool per_tree = pl.in == nullptr;
                if (per_tree && ch != 0) continue;
                const int64_t ob = per_tree ? (int64_t)b * V : cb;
                if (pl.down_sorted) pl.down_sorted[ob + i] = q ? d.y : d.x;
                if (pl.down_vertex) pl.down_vertex[ob + p[j]] = q ? d.y : d.x;
            }
            if (a.out_vertex) a.out_vertex[cb + p[j]] = d.y / d.x;
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

static size_t refine_lds_bytes(int V) { return 16 * (size_t)(V + kRecPad); }

// tree_filter_large.hip: the same algorithms with their arrays in a global workspace (V > kTfMaxV)
size_t mst_large_ws_bytes(int E, int V);
size_t bfs_large_ws_bytes(int V);
size_t refine_large_ws_bytes(int B, int C, int V);
int launch_mst_large(const int* edge_index, const float* edge_weight, int B, int E, int V, int* edge_out, int* n_out, char* ws, hipStream_t s);
int launch_bfs_large(const int* tree, int B, int V, int max_adj, int* si, int* sp, int* sc, int* levels, char* ws, hipStream_t s);
int launch_refine_large(const void* planes2, const float* edge_weight, const int* sorted_index, const int* sorted_child, const int* levels,
                        float* out_vertex, int B, int C, int V, int max_adj, int n_planes, char* ws, hipStream_t s);
static bool tf_fits_lds(int V) { return V <= kTfMaxV; }
constexpr int kTfMaxVLarge = 1 << 24;

}  // namespace bxi

extern "C" {

size_t bxi_mst_workspace_bytes(int B, int E, int V) {
    if (B < 0 || E <= 0 || V <= 1) return 0;
    const size_t head = (sizeof(int) * (size_t)(B > 0 ? B : 1) + 15) / 16 * 16;
    return head + (bxi::tf_fits_lds(V) && E <= 8 * bxi::kTfMaxV ? 0 : bxi::mst_large_ws_bytes(E, V) * (size_t)(B > 0 ? B : 1));
}

int bxi_mst_forward_i32(const int* edge_index, const float* edge_weight, int B, int E, int V, int* edge_out, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (B < 0 || E <= 0 || V <= 1) return BXI_ERR_BAD_SHAPE;
    if (V > bxi::kTfMaxVLarge || E > 8 * bxi::kTfMaxVLarge) return BXI_ERR_UNSUPPORTED;
    if (B == 0) return BXI_OK;
    if (!edge_index || !edge_weight || !edge_out) return BXI_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < bxi_mst_workspace_bytes(B, E, V) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return BXI_ERR_WORKSPACE;
    hipStream_t s = bxi::as_stream(stream);
    const size_t lds = (size_t)V * 12 + 4 * (size_t)((E + 31) / 32) + 16;
    if (!bxi::tf_fits_lds(V) || E > 8 * bxi::kTfMaxV || lds > 150 * 1024) {        // arrays in the workspace instead of LDS
        if (workspace_bytes < (sizeof(int) * (size_t)B + 15) / 16 * 16 + bxi::mst_large_ws_bytes(E, V) * (size_t)B) return BXI_ERR_WORKSPACE;
        return bxi::launch_mst_large(edge_index, edge_weight, B, E, V, edge_out, reinterpret_cast<int*>(workspace),
                                     reinterpret_cast<char*>(workspace) + (sizeof(int) * (size_t)B + 15) / 16 * 16, s);
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::mst_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("mst", s, bxi::mst_kernel, dim3(B), dim3(1024), lds, s, edge_index, edge_weight, E, V, edge_out, reinterpret_cast<int*>(workspace));
    return bxi::check_launch();
}

size_t bxi_bfs_workspace_bytes(int B, int V) {
    if (B < 0 || V <= 1) return 0;
    return bxi::tf_fits_lds(V) ? 0 : bxi::bfs_large_ws_bytes(V) * (size_t)(B > 0 ? B : 1);
}

int bxi_bfs_forward_i32(const int* tree_edges, int B, int V, int max_adj, int* sorted_index, int* sorted_parent, int* sorted_child,
                        int* levels, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || V <= 1 || max_adj < 1) return BXI_ERR_BAD_SHAPE;
    if (V > bxi::kTfMaxVLarge || max_adj > 8) return BXI_ERR_UNSUPPORTED;
    if (B == 0) return BXI_OK;
    if (!tree_edges || !sorted_index || !sorted_parent || !sorted_child || !levels) return BXI_ERR_NULL_POINTER;
    hipStream_t s = bxi::as_stream(stream);
    if (!bxi::tf_fits_lds(V)) {
        if (!workspace || workspace_bytes < bxi_bfs_workspace_bytes(B, V) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return BXI_ERR_WORKSPACE;
        return bxi::launch_bfs_large(tree_edges, B, V, max_adj, sorted_index, sorted_parent, sorted_child, levels,
                                     reinterpret_cast<char*>(workspace), s);
    }
    const size_t lds = (size_t)V * 8 + (size_t)(V + 1) * 4 + (size_t)((V + 3) / 4) * 4 + 16;
    if (lds > 150 * 1024) return BXI_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::bfs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("bfs", s, bxi::bfs_kernel, dim3(B), dim3(bxi::kTfThreads), lds, s, tree_edges, V, max_adj, sorted_index, sorted_parent, sorted_child, levels);
    return bxi::check_launch();
}

size_t bxi_tree_refine_workspace_bytes(int B, int C, int V) {
    if (B < 0 || C <= 0 || V <= 1) return 0;
    return bxi::tf_fits_lds(V) ? 0 : bxi::refine_large_ws_bytes(B, C, V);
}

static int launch_refine(bxi::RefineArgs& a, void* workspace, size_t workspace_bytes, void* stream) {
    if (a.B < 0 || a.C <= 0 || a.V <= 1 || a.max_adj < 1) return BXI_ERR_BAD_SHAPE;
    if (a.V > bxi::kTfMaxVLarge || a.C > 65535) return BXI_ERR_UNSUPPORTED;
    if (a.B == 0) return BXI_OK;
    if (!a.edge_weight || !a.sorted_index || !a.sorted_child || !a.levels) return BXI_ERR_NULL_POINTER;
    for (int q = 0; q < a.n_planes; ++q)
        if (q + 1 == a.n_planes && !a.pl[q].in) return BXI_ERR_NULL_POINTER;
    if (!bxi::tf_fits_lds(a.V)) {
        if (!workspace || workspace_bytes < bxi_tree_refine_workspace_bytes(a.B, a.C, a.V) || (reinterpret_cast<uintptr_t>(workspace) & 15))
            return BXI_ERR_WORKSPACE;
        return bxi::launch_refine_large(a.pl, a.edge_weight, a.sorted_index, a.sorted_child, a.levels, a.out_vertex, a.B, a.C, a.V, a.max_adj,
                                        a.n_planes, reinterpret_cast<char*>(workspace), bxi::as_stream(stream));
    }
    const size_t lds = bxi::refine_lds_bytes(a.V);
    if (lds > 160 * 1024) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = bxi::as_stream(stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bxi::tree_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { bxi::set_last_hip_error((int)e); return BXI_ERR_LAUNCH; }
    }
    BXI_LAUNCH("tree_refine", s, bxi::tree_refine_kernel, dim3(a.B, a.C), dim3(bxi::kTfThreads), lds, s, a);
    return bxi::check_launch();
}

int bxi_tree_refine_forward_f32(const float* feature_in, const float* edge_weight, const int* sorted_index, const int* sorted_child,
                                const int* levels, int B, int C, int V, int max_adj, float* feature_out, float* feature_aggr,
                                float* feature_aggr_up, float* weight_sum, float* weight_sum_up, void* workspace, size_t workspace_bytes,
                                void* stream) {
    if (B > 0 && (!feature_in || !feature_out || !feature_aggr || !feature_aggr_up || !weight_sum || !weight_sum_up)) return BXI_ERR_NULL_POINTER;
    bxi::RefineArgs a{};
    a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child; a.levels = levels;
    a.pl[0].up_sorted = weight_sum_up; a.pl[0].down_vertex = weight_sum;                       // the traversal of ones (refine.cu:223-228)
    a.pl[1].in = feature_in; a.pl[1].up_sorted = feature_aggr_up; a.pl[1].down_vertex = feature_aggr;
    a.out_vertex = feature_out;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.n_planes = 2;
    return launch_refine(a, workspace, workspace_bytes, stream);
}

int bxi_tree_refine_backward_feature_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                         const int* sorted_child, const int* levels, const float* weight_sum, int B, int C, int V,
                                         int max_adj, float* grad_feature, void* workspace, size_t workspace_bytes, void* stream) {
    if (B > 0 && (!grad_out || !weight_sum || !grad_feature)) return BXI_ERR_NULL_POINTER;
    bxi::RefineArgs a{};
    a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child; a.levels = levels;
    a.pl[0].in = grad_out; a.pl[0].pre_div = weight_sum; a.pl[0].down_vertex = grad_feature;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.n_planes = 1;
    return launch_refine(a, workspace, workspace_bytes, stream);
}

size_t bxi_tree_refine_backward_weight_workspace_bytes(int B, int C, int V) {
    if (B < 0 || C <= 0 || V <= 0) return 0;
    return (sizeof(float) * 4 * (size_t)(B > 0 ? B : 1) * C * V + 15) / 16 * 16 + bxi_tree_refine_workspace_bytes(B, C, V);
}

int bxi_tree_refine_backward_weight_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                        const int* sorted_parent, const int* sorted_child, const int* levels, const float* feature_out,
                                        const float* feature_aggr, const float* feature_aggr_up, const float* weight_sum,
                                        const float* weight_sum_up, int B, int C, int V, int max_adj, float* grad_weight,
                                        float* grad_feature, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || C <= 0 || V <= 1) return BXI_ERR_BAD_SHAPE;
    if (B == 0) return BXI_OK;
    if (!grad_out || !sorted_parent || !feature_out || !feature_aggr || !feature_aggr_up || !weight_sum || !weight_sum_up || !grad_weight)
        return BXI_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < bxi_tree_refine_backward_weight_workspace_bytes(B, C, V) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return BXI_ERR_WORKSPACE;
    char* refine_ws = reinterpret_cast<char*>(workspace) + (sizeof(float) * 4 * (size_t)B * C * V + 15) / 16 * 16;
    const size_t plane = (size_t)B * C * V;
    float* GU = reinterpret_cast<float*>(workspace); float* OG = GU + plane; float* FGU = OG + plane; float* OG2 = FGU + plane;
    bxi::RefineArgs a{};
    a.edge_weight = edge_weight; a.sorted_index = sorted_index; a.sorted_child = sorted_child; a.levels = levels;
    // plane 0: g~ = g / weight_sum -> GU, OG (and, in vertex order, refine_backward_feature's result); plane 1: g~ * feature_out
    a.pl[0].in = grad_out; a.pl[0].pre_div = weight_sum; a.pl[0].up_sorted = GU; a.pl[0].down_sorted = OG; a.pl[0].down_vertex = grad_feature;
    a.pl[1].in = grad_out; a.pl[1].pre_div = weight_sum; a.pl[1].pre_mul = feature_out; a.pl[1].up_sorted = FGU; a.pl[1].down_sorted = OG2;
    a.B = B; a.C = C; a.V = V; a.max_adj = max_adj; a.n_planes = 2;
    int rc = launch_refine(a, refine_ws, bxi_tree_refine_workspace_bytes(B, C, V), stream);
    if (rc != BXI_OK) return rc;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)B * V + 255) / 256);
    BXI_LAUNCH("tree_grad_weight", s, bxi::tree_grad_weight_kernel, dim3(grid), dim3(256), 0, s, feature_aggr_up, feature_aggr, weight_sum_up,
               weight_sum, GU, OG, FGU, OG2, edge_weight, sorted_index, sorted_parent, B, C, V, grad_weight);
    return bxi::check_launch();
}

}  // extern "C"
