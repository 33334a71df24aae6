// Fused MLP kernels on the fp16 matrix pipe with streamed weights (hmlp.hip): every MLP of the model for hidden sizes
// 64 / 128 / 256 and any num_layers >= 2.  Weight images, argument blocks, launchers.
#pragma once
#include "common.h"
#include "hedge.h"

namespace gm {

// ---- image of one Linear: [t, 1/t, 0, 0] | bias * t (out_pad floats) | fp16 hi / lo A-operand fragments
//      [out_pad / 32][k_pad / 16][2 parts][64 lanes][8 halves]   (t: power of two with max|W| t in [0.25, 0.5))
__host__ __device__ static inline size_t hm_lin_floats(int out_pad, int k_pad) { return 4 + (size_t)out_pad + (size_t)out_pad * k_pad; }

struct PackHmJob {
    const float* W;     // row-major [rows][ld]
    int ld;
    const float* bias;  // or nullptr
    int bias_n;         // valid bias entries (the rest of out_pad is zero)
    int out_valid;      // valid rows per output segment
    int out_pad;        // total padded outputs (multiple of 32)
    int out_seg;        // outputs per segment (= out_pad when there is one)
    int k_valid;        // valid inputs per input segment
    int k_pad;          // total padded inputs (multiple of 16)
    int k_seg;          // inputs per segment (= k_pad when there is one)
    int col0[2];        // first column of W for segment 0 / 1 (output segments OR input segments)
    float* dst;
};
constexpr int kPackHmMax = 24;
int pack_hm(const PackHmJob* jobs, int n, hipStream_t s);

struct HmEdgeArgs {
    const CsrHeader* hdr;
    int n_edges_host;
    const int* dst;
    const int* src;
    const int* eid;       // row of the input for sorted position p, or nullptr
    const int* eid_out;   // row of the output (and of the residual read), or nullptr
    const float* P;       // [N][2H]
    const float* e_in;    // processor: [E][H]; encoder: [E][k1]
    float* e_out;
    float* agg;
    const float* w;       // Linear images L0 | L1 | .. | L_NL, back to back
    const float* ln_g;
    const float* ln_b;
    float eps;
    int residual;
    int k1;
    int nl;               // num_layers: nl + 1 Linears
    const int2* blk;      // processor: 32-edge block table (hedge.h)
    const int* head;      // processor: head list of the groups
    float* side;          // processor: [n_groups][H] head partials
    const EdgeBlockHeader* tab;
    ProfState* prof;
};

struct HmNodeArgs {
    int n_nodes;
    const float* x_in;    // mode 0: [N][k1]; mode 1 / 2: h [N][H]
    int k1;
    const float* agg;
    float* agg_clear;
    float* h_out;
    int residual;
    const float* w;       // L0 | .. | L_NL
    const float* ln_g;
    const float* ln_b;
    float eps;
    int nl;
    int tail;             // 0 none, 1 projection P = h [W_i | W_j]^T (+ b1), 2 decoder
    const float* w_tail;  // tail 1: one Linear image (2H outputs); tail 2: nl images H -> H, then H -> 32 (zero-padded)
    float* P_out;
    float* dec_out;
    int out_dim;
    const int* stitch;    // mode 1: stitch / head lists + side buffer of the edge kernel's head partials (hedge.h), or nullptr
    const int* head;
    const float* side;
    const EdgeBlockHeader* tab;
    ProfState* prof;
};

bool hm_supported(int H);
int launch_edge_hm(int H, bool enc, const HmEdgeArgs& a, hipStream_t s);
int launch_node_hm(int H, int mode, const HmNodeArgs& a, hipStream_t s);

}  // namespace gm
