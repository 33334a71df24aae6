// Device bodies of the 32-edge block tables over a destination-sorted edge list (hedge.h: EdgeBlocks), shared by the launches
// that build them: hedge.hip (build_edge_blocks: a caller's edge_index) and graph.hip (the rollout step's destination sort, where the
// plan rides in the fill pass and the tables in the segment sort's launch).
#pragma once
#include "common.h"
#include "hedge.h"

namespace gm {

constexpr int kBlockEdges = 32;   // edges per block (hedge.hip: BE)
__host__ __device__ inline int edge_blocks_padded(int n_edges) { return ((n_edges + kBlockEdges - 1) / kBlockEdges + 3) & ~3; }

#if defined(__HIPCC__)
// One thread: per-graph block counts and their prefix.  Every graph is padded to a multiple of 4 blocks = one group (the unit over
// which the scatter-add carries a running sum: a wave's rows in hmlp.hip, 4 ticks in the systolic kernel), so groups never straddle
// graphs and the partial sums of a graph do not depend on what else shares the launch.
__device__ inline void edge_blocks_plan(const int* __restrict__ in_ptr, int n_nodes, const int* n_per_dev, int n_per_host,
                                        EdgeBlockHeader* tab, int* gblk, int max_graphs) {
    int n_per = n_per_dev ? *n_per_dev : n_per_host;
    if (n_per <= 0) n_per = n_nodes > 0 ? n_nodes : 1;
    int G = (n_nodes + n_per - 1) / n_per;
    if (G > max_graphs) G = max_graphs;   // capacity of the prefix array (never hit: it holds n_nodes + 2 entries)
    int pb = 0;
    for (int g = 0; g < G; ++g) {
        const int lo = g * n_per, hi = min(n_nodes, (g + 1) * n_per);
        gblk[g] = pb;
        pb += edge_blocks_padded(in_ptr[hi] - in_ptr[lo]);
    }
    gblk[G] = pb;
    tab->n_blocks = pb;
    tab->n_groups = pb / 4;
    tab->n_graphs = G;
    tab->n_per_graph = n_per;
    tab->n_stitch = 0;   // counted by the fill pass (a later launch)
}

// blk[b] = (first edge, count | flags << 8) with flags 1 = first block of its group, 2 = last;
// head[g] = the destination whose segment continues from group g - 1 into group g (its sum over group g is a "head
// partial", stored to the side buffer), or -1;  stitch[v] = first group of the run of head partials of destination v
// (the caller has set stitch[] to -1).  Thread `t` of `nt` takes blocks t, t + nt, ...
__device__ inline void edge_blocks_fill(const int* __restrict__ in_ptr, const int* __restrict__ dst, int n_nodes, EdgeBlockHeader* tab,
                                        const int* __restrict__ gblk, int2* blk, int2* seg, int* __restrict__ head, int* __restrict__ stitch,
                                        int* __restrict__ stitch_list, int t, int nt) {
    constexpr int BE = kBlockEdges;
    const int nblk = tab->n_blocks, G = tab->n_graphs, n_per = tab->n_per_graph;
    for (int b = t; b < nblk; b += nt) {
        int lo = 0, hi = G;   // graph g with gblk[g] <= b < gblk[g + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (gblk[mid] <= b) lo = mid; else hi = mid;
        }
        const int g = lo, j = b - gblk[g];
        const int e0 = in_ptr[g * n_per], e1 = in_ptr[min(n_nodes, (g + 1) * n_per)];
        const int start = min(e0 + j * BE, e1);          // padding blocks: (end of the graph, 0 edges)
        const int cnt = min(BE, e1 - start);
        const int fl = ((j & 3) == 0 ? 1 : 0) | ((j & 3) == 3 ? 2 : 0);
        blk[b] = make_int2(start, cnt | (fl << 8));
        {
            // Segment structure of the block's rows (destination-sorted): same reasoning as the kernels' scans.  A block shorter
            // than 32 rows ends its graph; the group's last block ends every open piece.
            unsigned cont = 0, last = 0;
            int prev = (cnt > 0 && (j & 3) != 0 && start > e0) ? dst[start - 1] : -1;   // a full block precedes it in the group
            for (int n = 0; n < cnt; ++n) {
                const int d = dst[start + n];
                if (d == prev) cont |= 1u << n;
                prev = d;
                bool is_last;
                if (n + 1 < cnt) is_last = dst[start + n + 1] != d;
                else is_last = (j & 3) == 3 || cnt < BE || start + cnt >= e1 || dst[start + cnt] != d;
                if (is_last) last |= 1u << n;
            }
            seg[b] = make_int2((int)cont, (int)last);
        }
        if ((j & 3) == 0) {
            int h = -1;
            if (cnt > 0 && start > e0 && dst[start - 1] == dst[start]) h = dst[start];
            head[b >> 2] = h;
            if (h >= 0) {
                // the run of head partials of h starts here unless the previous group (then entirely h's) is one too
                const int ps = start - 4 * BE;
                const bool prev_is_head = j >= 4 && dst[ps] == h && ps > e0 && dst[ps - 1] == h;
                if (!prev_is_head) {
                    stitch[h] = b >> 2;
                    stitch_list[atomicAdd(&tab->n_stitch, 1)] = h;   // one entry per destination: a destination has one run
                }
            }
        }
    }
}
#endif

}  // namespace gm
