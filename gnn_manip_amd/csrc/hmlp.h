// Fused MLP kernels on the fp16 matrix pipe with streamed weights (hmlp.hip): every MLP of the model for hidden sizes
// 64 / 128 / 256 and any num_layers >= 2.  Weight images, argument blocks, launchers.
#pragma once
#include "common.h"
#include "hedge.h"

namespace gm {

// ---- image of one Linear:  [t, 1/U, U, 0] | bias * U (out_pad floats) | fp16 hi / lo A-operand fragments of t W
//      [out_pad / 32][k_pad / 16][2 parts][64 lanes][8 halves]
// Scales (all powers of two, so nothing is rounded).  The Linears of an MLP form a chain; the activations are never brought
// back to their natural magnitude between them: the accumulators of Linear l are  U_l z_l  with  U_l = t_1 .. t_l  (times the
// scale of the chain's input image, 1 for h / e / agg, a per-row power of two for the encoders' raw features), ReLU passes
// them on, the bias comes pre-multiplied by U_l, and the scale leaves where results are read (LayerNorm: eps U^2; outputs:
// 1 / U).  U_l is chosen from an estimate of the activations' rms m_l -- a ReLU layer maps the second moment
// m^2 -> gain^2 m^2 + rms(b)^2 / 2  with  gain = ||W_l||_F / sqrt(out) / sqrt(2) -- so that the operand image of the next Linear
// has an rms near 2^4 (and the packed weights t W stay where both halves of their split are normal): elements down to 2^-7 of that keep all
// 22 bits of the two-way fp16 split, values up to 2^12 times it fit (a hub node's aggregate, an outlier activation), and scaling (W_l, b_l) by s and W_l+1 by 1/s -- which
// leaves the function unchanged -- leaves every operand image unchanged bit for bit.  A value that does not fit
// (|x| >= 65504 in an operand image) raises ERRF_SPLIT_RANGE in the forward's CSR header (status() reports it).
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
    int pred;           // job (index in the whole list handed to pack_hm) whose output is this Linear's input, or -1: chain head
    float in_rms;       // chain heads: assumed rms of the input operand image (1 for natural-magnitude inputs)
    int gain_cols;      // columns [gain_col0, gain_col0 + gain_cols) of W that make this Linear's pre-activation (0: the packed ones;
    int gain_col0;      //   phi_e's first Linear is packed as its e block, but h_i and h_j feed the same pre-activation)
    int center;         // this Linear feeds a LayerNorm: pack W - 1 mean_rows(W) and b - mean(b) (means over the valid outputs), so that
                        // its outputs have zero mean over the features and the kernels' statistics are a sum of squares
    float* dst;
};
constexpr int kPackHmMax = 20;
// jobs: host list that stays alive until the stream has run the copy; jobs_dev: device room for n jobs; stats: device scratch of
// 4 * n floats (phase 1 fills it with every job's gain / bias rms / bias max / weight max, phase 2 walks the chains)
int pack_hm(const PackHmJob* jobs, int n, PackHmJob* jobs_dev, float* stats, hipStream_t s);
constexpr float kHmTargetRms = 16.0f;       // rms the operand images aim at: full 22 bits from rms / 128 up, values up to 4096 x rms fit
constexpr float kHmRawInputRms = 32.0f;     // encoders: raw feature rows are scaled so that their maximum is in [2^6, 2^7)

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
    int discard_e_out;    // processor with scatter-add: e_out is not read after this launch and may stay unwritten
    int k1;
    int nl;               // num_layers: nl + 1 Linears
    int h_valid;          // features that exist (<= H): the rest of the width is zero padding
    const int2* blk;      // processor: 32-edge block table (hedge.h)
    const int* head;      // processor: head list of the groups
    float* side;          // processor: [n_groups][H] head partials
    const EdgeBlockHeader* tab;
    int* flags;           // error flags of the forward (CsrHeader::error_flags), or nullptr
    ProfState* prof;
};

struct HmNodeArgs {
    int n_nodes;
    const float* x_in;    // mode 0: [N][k1]; mode 1 / 2: h [N][H]
    int k1;
    const float* agg;
    float* h_out;
    int residual;
    const float* w;       // L0 | .. | L_NL
    const float* ln_g;
    const float* ln_b;
    float eps;
    int nl;
    int h_valid;          // features that exist (<= H)
    int tail;             // 0 none, 1 projection P = h [W_i | W_j]^T (+ b1), 2 decoder
    const float* w_tail;  // tail 1: one Linear image (2H outputs); tail 2: nl images H -> H, then H -> 32 (zero-padded)
    float* P_out;
    const float* p_scale; // tail 1: P is written times *p_scale (a power of two: the consuming systolic edge kernel's weight scale), or nullptr
    float* dec_out;
    int out_dim;
    const int* stitch;    // mode 1: stitch / head lists + side buffer of the edge kernel's head partials (hedge.h), or nullptr
    const int* head;
    const float* side;
    const EdgeBlockHeader* tab;
    int* flags;           // error flags of the forward (CsrHeader::error_flags), or nullptr
    ProfState* prof;
};

bool hm_supported(int H);
// width the kernels run a model of hidden size h at: h (1 .. 256) zero-padded to 64 / 128 / 256, else 0
static inline int hm_padded_hidden(int h) { return (h < 1 || h > 256) ? 0 : (h <= 64 ? 64 : (h <= 128 ? 128 : 256)); }
int launch_edge_hm(int H, bool enc, const HmEdgeArgs& a, hipStream_t s);
int launch_node_hm(int H, int mode, const HmNodeArgs& a, hipStream_t s);

}  // namespace gm
