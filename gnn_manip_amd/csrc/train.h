// Argument blocks and launchers of the training kernels (train.hip), used by train_model.hip.
#pragma once
#include "common.h"

namespace gm {

// activation tape of one MLP (all in the row order the MLP ran in)
struct TapePtr {
    float* a;     // [num_layers][rows][H] post-ReLU outputs of Linear 1 .. num_layers (a_l = a + (l - 1) rows H)
    float* xhat;  // [rows][H] normalised, pre-affine LayerNorm output (normed MLPs)
    float* rstd;  // [rows]    1 / sqrt(var + eps)
    uint32_t* mask;  // [num_layers][rows][H / 32] one bit per element: a_l > 0 (what the backward chain needs of a_l; its values are
                     // read by the weight-gradient jobs only), in the chain kernels' register order (train.hip: relu_mask_store)
};

struct TrainFwdArgs {
    int rows;
    const float* x_in;     // encoders: raw features [.][k1]; processor edge: e [E][H]; processor node / decoder: h [N][H]
    const int* rowidx;     // encoders / processor edge: row of x_in (and of `out`, processor edge) for row p, or nullptr
    int k1;
    const float* agg;      // processor node: [N][H]
    const int* dst;        // processor edge
    const int* src;
    const float* P;        // processor edge: [N][2H] = P_i (+ b1) | P_j
    const float* wstream;  // packed forward stream of this MLP
    const float* bias;     // bias of Linear 1 (unused by the processor edge MLP: it sits in P_i)
    const float* bias_tail;  // biases of Linear 2 .. num_layers + 1, H apart (decoder: the last one padded to 32)
    int nl;                // num_layers: the MLP has nl + 1 Linears
    const float* ln_g;
    const float* ln_b;
    float eps;
    TapePtr tape;
    float* out;            // [rows][H]; decoder: [rows][out_dim]
    int residual;          // processor: out = y + x_in
    int out_dim;
    // node MLPs (encoder, processor): tail that projects the new h for the NEXT edge step's factorised layer 1,
    // P_out [N][2H] = [h W_i^T + proj_bias | h W_j^T]; its two H x H images follow the MLP's in `wstream`.  nullptr: no tail.
    float* P_out;
    const float* proj_bias;
};

struct TrainBwdArgs {
    int rows;
    const float* dY;       // upstream gradient rows [.][H] (decoder: [rows][out_dim]); nullptr = zero
    const int* dyidx;      // row of dY for row p, or nullptr
    const float* dagg;     // edge: + dagg[dst[p]]
    const int* dst;
    const float* Gi;       // node MLPs: + W_i^T Gi[p] + W_j^T Gj[p] (next edge step's layer-1 input gradient), or nullptr
    const float* Gj;
    TapePtr tape;
    const float* ln_g;
    const float* wstream;  // packed TRANSPOSED stream: [W_i^T, W_j^T (if Gi)] W_(nl+1)^T .. W_2^T [W_1^T ...]
    float* dz;             // [nl + 1][dz_stride]: pre-activation gradients dz_l = dz + (l - 1) dz_stride of Linear l = 1 .. nl + 1, rows of H
    size_t dz_stride;      //   (the decoder's dz_(nl+1) is dY itself and is not written)
    int nl;
    float* ln_part;        // normed MLPs: scratch [workgroups][2H] for the LayerNorm parameter gradients (column sums of gy xhat | gy)
    float* dgamma;         //   ... which launch_train_bwd then adds, in workgroup order, to dgamma / dbeta  (nullptr: not computed)
    float* dbeta;
    float* dx_resid;       // node: residual path; receives dY before dx adds the MLP's input gradient to it (may alias dY / dx)
    float* dx;             // edge: de_in; node: dh_in; decoder: dh; projection: dh_in
    const int* dxidx;      // edge: row of dx for row p (block API: the caller's edge order), or nullptr
    float* dx_in;          // encoders: gradient w.r.t. the raw input rows [rows][k1], or nullptr (then the chain stops at dz1)
    int k1;
    float* dagg_out;       // node: [rows][H]
    int residual;          // edge: de_in += dY
    int out_dim;
};

// bf16 x 3 operand image (train.hip) of a Linear with `ksub` outputs and `w_rows` inputs taken from the row-major W:
// element (o, k) = W[o][col0 + k] (fwd) or W[k][col0 + o] (transposed block: the backward product's weight)
struct PackTJob {
    const float* W;
    int w_rows, ld, col0, ksub, fwd;
    size_t dst_off;   // floats from the base of the packed buffer
};
constexpr int kPackTJobsMax = 96;
struct PackTJobs {
    int n;
    PackTJob job[kPackTJobsMax];
};
constexpr int kStageFloatsB3 = 6144;      // one stage of the bf16 x 3 weight stream: 8 groups of 3 KiB
int layer_stages_b3(int k, int out);      // stages of a Linear with k inputs and `out` outputs

int train_kernels_init();
int launch_train_fwd(int H, int kind, const TrainFwdArgs& a, hipStream_t s);
struct WgradBatch;
// wb != nullptr: the LayerNorm partial sums go to one of the batch's own regions and are reduced by its next flush (one launch with
// the weight-gradient reductions) instead of a launch of their own -- at most kWgLnMax chains between two flushes.
int launch_train_bwd(int H, int kind, const TrainBwdArgs& a, hipStream_t s, WgradBatch* wb = nullptr);
// Weight gradients: jobs are collected and run a batch per launch pair (GEMM over row chunks + fixed-order reduction).
//   out[m][col0 + k] += sum_r dz[r][m] * X[xidx ? xidx[r] : r][k];  db[m] += sum_r dz[r][m] (db may be nullptr)
// A job reads its operands when the batch is FLUSHED: flush before anything overwrites them.
struct WgJob {
    const float* dz;
    const float* X;
    const int* xidx;
    float* out;
    float* db;
    int ldz, M, ldx, K, rows, ldw, col0;
    int Mp, Kp, KT, tiles, chunk, G;
    size_t part_off;   // floats from the partial buffer: [G][Mp][Kp] tiles, then [G][Mp] bias partials
};
constexpr int kWgJobsMax = 12;
constexpr int kWgLnMax = 2;   // backward chains whose LayerNorm partials may wait for one flush (a node chain and the edge chain after it)
struct WgLnJob {              // dgamma[c] += sum_g part[g][c], dbeta[c] += sum_g part[g][H + c]
    const float* part;
    int G, H;
    float* dgamma;
    float* dbeta;
};
struct WgJobs {
    int n;
    WgJob job[kWgJobsMax];
    int n_ln;                 // LayerNorm parameter gradients of the backward chains that ran since the last flush
    WgLnJob ln[kWgLnMax];
};
struct WgradBatch {
    WgJobs jobs{};
    float* part = nullptr;   // wgrad_partial_floats(H) floats: weight-gradient partials, then kWgLnMax LayerNorm regions
    size_t cap = 0;          // floats of the weight-gradient part
    size_t ln_floats = 0;    // floats of one LayerNorm region
    hipStream_t stream = nullptr;
    float* ln_region(int slot) const { return part + cap + (size_t)slot * ln_floats; }
};
void wgrad_batch_init(WgradBatch& b, float* part, int H, hipStream_t s);
size_t wgrad_partial_floats(int H);
int wgrad_enqueue(WgradBatch& b, const float* dz, int ldz, int M, const float* X, int ldx, int K, const int* xidx, int64_t rows, float* out,
                  int ldw, int col0, float* db);
int wgrad_flush(WgradBatch& b);
size_t train_bwd_ln_part_floats(int H);   // size of TrainBwdArgs.ln_part
int launch_pack_b3_batch(const PackTJobs& jobs, float* base, hipStream_t s);
int launch_segment_sum(int H, const int* ptr, const int* perm, const float* rows, const float* scale, const float* shift, float* out,
                       int64_t n, hipStream_t s);
// two segment sums over the same rows in one launch (out2 = sums over ptr2 / perm2; scale / shift apply to both)
int launch_segment_sum_pair(int H, const int* ptr, const int* perm, const int* ptr2, const int* perm2, const float* rows, const float* scale,
                            const float* shift, float* out, float* out2, int64_t n, hipStream_t s);
int launch_swap_index(const int* src_sorted, int64_t e, int64_t* ei2, hipStream_t s);

enum : int { TK_ENC_EDGE = 0, TK_ENC_NODE = 1, TK_PROC_EDGE = 2, TK_PROC_NODE = 3, TK_DEC = 4, TK_PROJ = 5 };   // TK_PROJ: out [N][2H] = h [W_i | W_j]^T + [b1 | 0]
enum : int { TB_ENC = 0, TB_EDGE = 1, TB_NODE = 2, TB_DEC = 3, TB_PROJ = 4 };  // TB_PROJ: dx = dY + W_i^T Gi + W_j^T Gj only

}  // namespace gm
