// The reference's own bfs.cu / refine.cu KERNELS, run on the CPU.  TEST INFRASTRUCTURE ONLY.
// oracle/Makefile (target `ref`, build container only) extracts the `__global__` functions of
//   $(REFERENCE_ROOT)/mmdet/ops/tree_filter/src/bfs/bfs.cu        (adj_vec_kernel, breadth_first_sort_kernel)
//   $(REFERENCE_ROOT)/mmdet/ops/tree_filter/src/refine/refine.cu  (root_leaf_prop_kernel, leaf_root_aggr_kernel, root_leaf_grad_kernel)
// into a temporary file outside the repository (REF_TREE_KERNELS_INC), which is compiled in here through the shim
// cuda_on_cpu.h and deleted; only the .so lands in oracle/_ref/.  Nothing of the reference is copied into the repo.
// The host functions of those files use ATen/THC and `<<< >>>` and are NOT compiled: the launch sequences below restate
// them (grid / block shapes, argument order, the tensor arithmetic between launches), citing the lines.
#include <cmath>
#include <cstring>
#include <vector>
#include "cuda_on_cpu.h"

#define CUDA_NUM_THREADS 64                    /* bfs.cu:16, refine.cu:16 */
#include REF_TREE_KERNELS_INC

using cpu_cuda::launch;

extern "C" {

// bfs_forward (bfs.cu:92-135): all six work tensors zero-initialised, one block of 64 threads per tree, two launches.
int ref_bfs_forward(const int* edge_index /*[B,V-1,2]*/, int B, int V, int max_adj, int* sorted_index /*[B,V]*/,
                    int* sorted_parent /*[B,V]*/, int* sorted_child /*[B,V,max_adj]*/) {
    std::vector<int> edges(edge_index, edge_index + (size_t)B * (V - 1) * 2);
    std::vector<int> adj_vec((size_t)B * V * max_adj, 0), adj_len((size_t)B * V, 0), parent_index((size_t)B * V, 0);
    std::memset(sorted_index, 0, sizeof(int) * (size_t)B * V);
    std::memset(sorted_parent, 0, sizeof(int) * (size_t)B * V);
    std::memset(sorted_child, 0, sizeof(int) * (size_t)B * V * max_adj);
    launch(dim3(B), dim3(CUDA_NUM_THREADS), [&] { adj_vec_kernel(B, edges.data(), V, adj_vec.data(), adj_len.data(), max_adj); });
    launch(dim3(B), dim3(CUDA_NUM_THREADS), [&] {
        breadth_first_sort_kernel(sorted_index, sorted_parent, sorted_child, adj_vec.data(), adj_len.data(), parent_index.data(), B, V, max_adj);
    });
    return 0;
}

// refine_forward (refine.cu:201-249).  Outputs: the five tensors it returns.
int ref_refine_forward(const float* feature_in /*[B,C,V]*/, const float* edge_weight_in /*[B,V]*/, const int* sorted_index,
                       const int* sorted_parent_in, const int* sorted_child, int B, int C, int V, int max_adj, float* feature_out,
                       float* feature_aggr, float* feature_aggr_up, float* weight_sum, float* weight_sum_up) {
    std::vector<float> fin(feature_in, feature_in + (size_t)B * C * V), w(edge_weight_in, edge_weight_in + (size_t)B * V);
    std::vector<int> si(sorted_index, sorted_index + (size_t)B * V), sp(sorted_parent_in, sorted_parent_in + (size_t)B * V);
    std::vector<int> sc(sorted_child, sorted_child + (size_t)B * V * max_adj);
    std::memset(feature_aggr, 0, sizeof(float) * (size_t)B * C * V);
    std::memset(feature_aggr_up, 0, sizeof(float) * (size_t)B * C * V);
    std::memset(weight_sum, 0, sizeof(float) * (size_t)B * V);
    std::memset(weight_sum_up, 0, sizeof(float) * (size_t)B * V);
    const dim3 blk(CUDA_NUM_THREADS), fgrid(B, C), wgrid(B, 1);
    launch(fgrid, blk, [&] { leaf_root_aggr_kernel(fin.data(), feature_aggr_up, w.data(), si.data(), sc.data(), B, C, V, max_adj); });
    launch(fgrid, blk, [&] { root_leaf_prop_kernel(feature_aggr_up, feature_aggr, w.data(), si.data(), sp.data(), B, C, V); });
    launch(wgrid, blk, [&] { leaf_root_aggr_kernel(nullptr, weight_sum_up, w.data(), si.data(), sc.data(), B, 1, V, max_adj); });
    launch(wgrid, blk, [&] { root_leaf_prop_kernel(weight_sum_up, weight_sum, w.data(), si.data(), sp.data(), B, 1, V); });
    for (int b = 0; b < B; ++b)                                     // :245  feature_aggr / weight_sum.unsqueeze(1)
        for (int c = 0; c < C; ++c)
            for (int v = 0; v < V; ++v) feature_out[((size_t)b * C + c) * V + v] = feature_aggr[((size_t)b * C + c) * V + v] / weight_sum[(size_t)b * V + v];
    return 0;
}

// refine_backward_feature (refine.cu:251-300)
int ref_refine_backward_feature(const float* edge_weight_in, const int* sorted_index, const int* sorted_parent_in, const int* sorted_child,
                                const float* weight_sum, const float* grad_out, int B, int C, int V, int max_adj, float* grad_feature) {
    std::vector<float> w(edge_weight_in, edge_weight_in + (size_t)B * V), gn((size_t)B * C * V), gas((size_t)B * C * V, 0.f);
    std::vector<int> si(sorted_index, sorted_index + (size_t)B * V), sp(sorted_parent_in, sorted_parent_in + (size_t)B * V);
    std::vector<int> sc(sorted_child, sorted_child + (size_t)B * V * max_adj);
    for (int b = 0; b < B; ++b)                                     // :268 grad_out / weight_sum.unsqueeze(1)
        for (int c = 0; c < C; ++c)
            for (int v = 0; v < V; ++v) gn[((size_t)b * C + c) * V + v] = grad_out[((size_t)b * C + c) * V + v] / weight_sum[(size_t)b * V + v];
    std::memset(grad_feature, 0, sizeof(float) * (size_t)B * C * V);
    const dim3 blk(CUDA_NUM_THREADS), fgrid(B, C);
    launch(fgrid, blk, [&] { leaf_root_aggr_kernel(gn.data(), gas.data(), w.data(), si.data(), sc.data(), B, C, V, max_adj); });
    launch(fgrid, blk, [&] { root_leaf_prop_kernel(gas.data(), grad_feature, w.data(), si.data(), sp.data(), B, C, V); });
    return 0;
}

// refine_backward_weight (refine.cu:302-370)
int ref_refine_backward_weight(const float* edge_weight_in, const int* sorted_index, const int* sorted_parent_in, const int* sorted_child,
                               const float* feature_out, const float* feature_aggr_in, const float* feature_aggr_up_in,
                               const float* weight_sum_in, const float* weight_sum_up_in, const float* grad_out, int B, int C, int V,
                               int max_adj, float* grad_weight /*[B,V]*/) {
    const size_t n = (size_t)B * C * V;
    std::vector<float> w(edge_weight_in, edge_weight_in + (size_t)B * V), gn(n), fg(n), gall(n, 0.f), gnall(n, 0.f), gas(n, 0.f), fgas(n, 0.f);
    std::vector<float> fa(feature_aggr_in, feature_aggr_in + n), fau(feature_aggr_up_in, feature_aggr_up_in + n);
    std::vector<float> ws(weight_sum_in, weight_sum_in + (size_t)B * V), wsu(weight_sum_up_in, weight_sum_up_in + (size_t)B * V);
    std::vector<int> si(sorted_index, sorted_index + (size_t)B * V), sp(sorted_parent_in, sorted_parent_in + (size_t)B * V);
    std::vector<int> sc(sorted_child, sorted_child + (size_t)B * V * max_adj);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int v = 0; v < V; ++v) {
                const size_t o = ((size_t)b * C + c) * V + v;
                gn[o] = grad_out[o] / weight_sum_in[(size_t)b * V + v];   // :342
                fg[o] = gn[o] * feature_out[o];                           // :343
            }
    const dim3 blk(CUDA_NUM_THREADS), fgrid(B, C);
    launch(fgrid, blk, [&] { leaf_root_aggr_kernel(gn.data(), gas.data(), w.data(), si.data(), sc.data(), B, C, V, max_adj); });
    launch(fgrid, blk, [&] { leaf_root_aggr_kernel(fg.data(), fgas.data(), w.data(), si.data(), sc.data(), B, C, V, max_adj); });
    // root_leaf_grad_kernel sets node_per_thread[tid] = -1 and goes on WITHOUT a __syncthreads() (refine.cu:165-168): each
    // block is preceded by a run on zero vertices, which only performs that initialisation (see cuda_on_cpu.h)
    auto reset = [&] { root_leaf_grad_kernel(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 1, 0); };
    launch(fgrid, blk, [&] {
        root_leaf_grad_kernel(fau.data(), gas.data(), fa.data(), gas.data(), w.data(), gall.data(), si.data(), sp.data(), B, C, C, V);
    }, reset);
    launch(fgrid, blk, [&] {
        root_leaf_grad_kernel(wsu.data(), fgas.data(), ws.data(), fgas.data(), w.data(), gnall.data(), si.data(), sp.data(), B, 1, C, V);
    }, reset);
    for (int b = 0; b < B; ++b)                                     // :367 (grad_all - grad_norm_all).sum(1)
        for (int v = 0; v < V; ++v) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += gall[((size_t)b * C + c) * V + v] - gnall[((size_t)b * C + c) * V + v];
            grad_weight[(size_t)b * V + v] = s;
        }
    return 0;
}

}  // extern "C"
