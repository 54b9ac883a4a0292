// C wrapper around the reference's own Boruvka MST (mmdet/ops/tree_filter/src/mst/boruvka.cpp), which is plain C++.
// TEST INFRASTRUCTURE ONLY.  The reference sources are compiled where they lie under $(REFERENCE_ROOT) by
// oracle/Makefile (target _ref/libboruvka_ref.so, build container only); nothing of them is copied here.  This file
// only restates what mst.cu:49-56,86-91 does around the call: fill the edge array, run, free.
#include "boruvka.hpp"

extern "C" int ref_boruvka_mst(int vertex_count, int edge_count, const int* edge_index /*[E,2]*/,
                               const float* edge_weight /*[E]*/, int* edge_out /*[V-1,2]*/) {
    Graph* g = createGraph(vertex_count, edge_count);
    for (int i = 0; i < edge_count; ++i) {
        g->edge[i].src = edge_index[2 * i];
        g->edge[i].dest = edge_index[2 * i + 1];
        g->edge[i].weight = edge_weight[i];
    }
    boruvkaMST(g, edge_out);
    delete[] g->edge;
    delete g;
    return 0;
}
