// Systolic processor edge kernel (hedge.hip): weight image, block / chunk tables, launcher.
#pragma once
#include "common.h"
#include "mlp.h"

namespace gm {

// ---- tables of 32-edge blocks over a destination-sorted edge list (one or more equal-sized graphs back to back).
// The scatter-add keeps a running sum over a GROUP of 4 blocks.  A segment that begins in a group is stored to agg by
// that group (whole, or its first piece); the pieces that later groups hold of it ("head partials") go to a side
// buffer [n_groups][H] and are added in group order by the node kernel: no atomics, one summation order.
struct EdgeBlockHeader {
    int n_blocks;
    int n_groups;       // n_blocks / 4
    int n_graphs;
    int n_per_graph;
    int n_stitch;       // entries of stitch_list
    int pad[3];
};
struct EdgeBlocks {
    EdgeBlockHeader* hdr;
    int* gblk;          // [n_graphs + 1] first block of a graph
    int2* blk;          // [n_blocks] (first edge, count | flags << 8); flags: 1 first block of its group of 4, 2 last
    int2* seg;          // [n_blocks] (continuation bits, last-row bits) of the block's rows for the systolic kernel's scatter-add:
                        // bit n of .x: row n continues the segment of the row before it (bit 0: of the previous block's last
                        // row, inside a group); bit n of .y: row n is the last one of its segment's piece in this group
    int* head;          // [n_groups] destination whose segment continues from the previous group into this one, or -1
    int* stitch;        // [n_nodes] first group of the run of head partials of a destination, or -1
    int* stitch_list;   // [n_nodes] the destinations with stitch[v] >= 0, in no particular order (n_stitch of them): what the systolic
                        // node path's agg_stitch_kernel walks, once per message-passing step, instead of every node
    int64_t max_blocks;
};
size_t edge_blocks_ints(int64_t n_nodes, int64_t edge_capacity);
constexpr int kSinkRows = 1024;   // power of two >= workgroups of a launch
size_t edge_groups_max(int64_t n_nodes, int64_t edge_capacity);   // rows of the side buffer: head partials + kSinkRows sink rows
EdgeBlocks carve_edge_blocks(int* base, int64_t n_nodes, int64_t edge_capacity);
// in_ptr / dst: the destination-sorted edge structure; n_per_graph: device pointer (GraphHeader::n_per_graph of the
// radius-graph build) or, if null, the host value (<= 0: one graph)
int build_edge_blocks(const int* in_ptr, const int* dst, int64_t n_nodes, int64_t edge_capacity, const int* n_per_graph_dev,
                      int n_per_graph_host, const EdgeBlocks& t, hipStream_t s);

// ---- weight image of one processor step's phi_e for the systolic kernel
constexpr int kPackH3Max = 16;
struct PackH3Job {
    const float* W1;   // Linear 1 weight [H][3H]: the block that multiplies e is packed (W_i, W_j live in P)
    int W1_col0;       // first column of that block (2H with the default concat order)
    int W1_ld;         // leading dimension of W1 (0: 3H, phi_e); a node MLP's image packs the agg block of its [H][2H] first Linear
    const float* W2;   // [H][H]
    const float* W3;   // [H][H]
    const float* b1;   // Linear 1 bias (applied through P; here only for the scale estimate)
    const float* b2;
    const float* b3;
    const float* gamma;
    const float* beta;
    float* dst;        // h3_image_floats() floats
    // encoder phi_e (launch_edge_sys_enc): W1 is [H][k1] with k1 raw features (k1 <= 16, leading dimension k1, W1_col0 = 0); its
    // rows are scaled by their own power of two in the kernel, bounded by the image's cap (hmlp.h).  0: a processor step.
    int enc_k1;
};
size_t h3_image_floats();
int pack_h3(const PackH3Job* jobs, int n, hipStream_t s);

int launch_edge_sys(const EdgeArgs& a, const EdgeBlocks& t, int64_t edge_capacity, hipStream_t s);
// device address of the power of two the systolic kernel's accumulators carry through its first Linear (T1 of the step's weight
// image): the node kernel that writes P for that step multiplies it in (NodeArgs::p_scale), the edge kernel then adds P_i + P_j as is
inline const float* edge_sys_p_scale(const float* h3_image) { return h3_image; }

// ---- processor node MLP and the next step's projections in the weight-stationary form (hedge.hip: sys_node_kernel, sys_proj_kernel)
struct NodeSysArgs {
    const float* h;        // [N][128] (residual input)
    const float* agg;      // [N][128], head partials already added (launch_agg_stitch)
    const float* Q;        // [N][128] = (h W_h^T + b1) T1 of this step's image (launch_proj_sys)
    float* h_out;          // [N][128], may be h
    const float* image;    // pack_h3 image of [W_agg | W2 | W3] of this node MLP
    int n;
    int* flags;
    float eps;
    ProfState* prof;
};
int launch_node_sys(const NodeSysArgs& a, hipStream_t s);
struct ProjSysArgs {
    const float* h;        // [N][128]
    float* P;              // [N][256] = h [W_i | W_j]^T + [b1 | 0], times *scale_p
    float* Q;              // [N][128] = h W_h^T + b1 of the next node MLP, times *scale_q
    const float* img_p;    // hmlp.h Linear images (256 / 128 outputs, 128 inputs)
    const float* img_q;
    const float* scale_p;  // device pointers (edge_sys_p_scale of the consuming kernels' images) or nullptr
    const float* scale_q;
    int n;
    int* flags;
    ProfState* prof;
};
int launch_proj_sys(const ProjSysArgs& a, hipStream_t s);
int launch_agg_stitch(float* agg, const float* side, const EdgeBlocks& t, int64_t n, ProfState* prof, hipStream_t s);
#ifndef GM_SYS_NODE_MIN_NODES
#define GM_SYS_NODE_MIN_NODES (32 * 256 * 6)
#endif
// Graphs from this size take the systolic node path (a workgroup needs blocks to pipeline).  The size that counts is ONE GRAPH's
// (the rollout step passes nodes_per_graph of a block-diagonal batch): the systolic and the streamed node kernels add a row's terms in
// different orders, so a choice by the batch's total would make candidate c of a batch of 10 differ in the last bits from candidate c
// of a batch of 8 -- or alone.  (Round 6 measured the systolic path at + 1 % on a C5 batch, 8 scenes x 5k = 40k nodes; the batch
// invariance is worth more than that.)
constexpr int64_t kSysNodeMinNodes = GM_SYS_NODE_MIN_NODES;
// encoder phi_e in the same weight-stationary form: raw rows [E][4] in sorted order -> e [E][128] (LayerNorm output)
int launch_edge_sys_enc(const EdgeArgs& a, hipStream_t s);
// whether the kernel's 32-bit byte offsets cover a graph of this size (P < 4 GiB, agg + side buffer < 4 GiB: about 4M nodes at
// hidden 128); larger launches take the streamed kernel
bool edge_sys_fits(int64_t n_nodes, int64_t edge_capacity);
// rows the latent edge array must own behind edge_capacity (the kernel reads whole 32-row blocks), zeroed by zero_edge_pad_rows
constexpr int kEdgePadRows = 32;
int zero_edge_pad_rows(const CsrHeader* hdr, float* e, int row_floats, hipStream_t s);

}  // namespace gm
