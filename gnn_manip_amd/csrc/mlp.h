// Argument blocks and launchers of the fused MLP kernels (mlp.hip), used by model.hip.
#pragma once
#include "common.h"

namespace gm {

struct EdgeArgs {
    const CsrHeader* hdr;  // device-side edge count (rollout path) or nullptr
    int n_edges_host;
    const int* dst;        // [E] aggregation node per sorted position (processor only)
    const int* src;        // [E]
    const int* eid;        // [E] row of e_in for sorted position p, or nullptr: row = p
    const int* eid_out;    // [E] row of e_out (and of the residual read), or nullptr: row = p
    const float* P;        // [N][2H]  P_i (+b1) | P_j
    const float* e_in;     // processor: [E][H]; encoder: raw edge_attr [E][k1]
    float* e_out;          // [E][H]
    float* agg;            // [N][H] pre-zeroed, or nullptr
    float* side;           // [n_groups][H] head partials of the scatter-add (hedge.h), sys / hm kernels
    const float* wstream_h3; // fp16 hi / lo image of the systolic kernel (hedge.h), or nullptr
    const float* wstream_hm; // fp16 hi / lo Linear images of this MLP for the streamed kernels (hmlp.h), or nullptr
    const int* edge_blocks;  // block / chunk tables of the edge list (carve_edge_blocks), or nullptr
    int64_t n_nodes_tab;     // n_nodes the tables were carved for
    ProfState* prof;         // timing of this launch (gm_model_profile), or nullptr
    int kernel_choice;       // 0 automatic, 5 systolic fp16 x 3, 6 streamed fp16 x 3
    const float* bias;     // processor: biases of layers 2..; encoder: biases of layers 1..
    const float* ln_g;
    const float* ln_b;
    float eps;
    int residual;          // e_out = e' + e_in
    int discard_e_out;     // nobody reads e_out after this launch (the last step of a forward): a kernel may leave it unwritten
    int P_prescaled;       // P was written times the systolic kernel's weight scale T1 (NodeArgs::p_scale): only that kernel may take the launch
    int k1;                // encoder: edge_dim
    int h_valid;           // the model's hidden_size (<= the width H the kernel runs at; LayerNorm statistics are over these features)
    int zero_pad_rows;     // encoder: e_out is a forward's latent array -- keep kEdgePadRows zero rows behind row n_edges (hedge.h)
    int* pad_rows_done;    // encoder: set to 1 when the launch has taken care of zero_pad_rows itself (else the caller launches zero_edge_pad_rows)
};

struct NodeArgs {
    int n_nodes;
    const float* x_in;     // mode 0: raw node features [N][k1]; mode 1/2: h [N][H]
    int k1;
    const float* agg;      // mode 1: [N][H]
    const int* edge_blocks;  // mode 1: block tables whose stitch / head lists say which side-buffer rows to add to agg (hedge.h), or nullptr
    int64_t n_nodes_tab, edge_capacity_tab;
    const float* side;
    int* err_flags;        // error flags of the forward (CsrHeader::error_flags: ERRF_SPLIT_RANGE), or nullptr
    int h_valid;           // the model's hidden_size (<= the width H the kernel runs at)
    float* h_out;          // [N][H] (may alias x_in)
    int residual;
    const float* wstream_hm;  // fp16 hi / lo Linear images of this MLP (hmlp.h), or nullptr
    const float* tail_hm;     // images of the tail
    ProfState* prof;
    int kernel_choice;        // as EdgeArgs (the node MLPs always take the streamed fp16 x 3 kernel)
    const float* bias;     // [NL+1][H]
    const float* ln_g;
    const float* ln_b;
    float eps;
    int tail;              // 0 none, 1 projection, 2 decoder
    const float* proj_bias;  // [H] layer-1 bias of the next edge MLP
    float* P_out;            // [N][2H]
    const float* p_scale;    // device pointer to the power of two P_out is multiplied with (edge_sys_p_scale), or nullptr
    const float* dec_bias;   // [NL][H] then [32] (out bias zero-padded)
    float* dec_out;          // [N][out_dim]
    int out_dim;
};

struct VecJob {
    const float* src;
    size_t dst_off;
    int count;   // floats copied
    int zero_to; // destination floats [count, zero_to) are zeroed (padding), 0 = none
};
constexpr int kVecJobsMax = 128;
struct VecJobs {
    int n;
    VecJob job[kVecJobsMax];
};
int launch_vec_batch(const VecJobs& jobs, float* base, hipStream_t s);
int launch_edge(int H, int NL, bool enc, const EdgeArgs& a, int64_t edge_capacity, hipStream_t s);
// whether launch_edge hands this processor launch to the systolic kernel (which then expects P pre-scaled: NodeArgs::p_scale)
bool edge_launch_is_sys(int H, int NL, const EdgeArgs& a, int64_t edge_capacity);
int launch_node(int H, int NL, int mode, const NodeArgs& a, hipStream_t s);

}  // namespace gm
