// Model handle shared by model.hip (inference) and train_model.hip (training).
#pragma once
#include <mutex>
#include <vector>
#include "common.h"
#include "hmlp.h"

struct gm_model {
    gm_model_desc d;
    int H, NL, M;                 // H: the model's hidden_size
    int Hp = 0;                   // the width the inference kernels run at: H zero-padded to 64 / 128 / 256 (hm_padded_hidden)
    int ci = 0, cj = 1, ce = 2;   // column block of phi_e's first Linear that multiplies h_i, h_j, e (gm_model_desc.col_*)
    int ch = 0, ca = 1;           // column block of phi_v's first Linear for h, agg (gm_model_desc.node_agg_first)
    float* packed_h3 = nullptr;  // fp16 hi / lo images of the systolic kernels (hedge.h), h3_image_floats() each: the M processor edge MLPs, the
                                 // edge encoder, then the M processor node MLPs (agg block of Linear 1 | Linear 2 | Linear 3)
    float* packed_hm = nullptr;  // fp16 hi / lo image of every Linear (hmlp.h)
    std::vector<gm::PackHmJob> hm_jobs;   // the pack job list of the last weight load (host copy of hm_jobs_dev)
    void* hm_jobs_dev = nullptr;
    float* hm_stats = nullptr;
    size_t hm_jobs_cap = 0;
    size_t hm_floats = 0, hm_enc_edge = 0, hm_enc_node = 0, hm_enc_node_tail = 0;
    std::vector<size_t> hm_edge, hm_node, hm_node_tail;
    std::vector<size_t> hm_node_q;   // Linear image of Q = h W_h^T + b1 of node step k (systolic node path: hedge.h)
    bool legacy = false;         // hidden 64 / 128 / 256: the bf16 x 3 streams (packed_t3) of the training kernels exist
    gm::ProfState* prof = nullptr;  // gm_model_profile
    int edge_kernel = 0;         // processor edge kernel of this model: 0 automatic, 5 / 6 see gm_model_set_edge_kernel
    float* vec = nullptr;     // per-MLP contiguous [bias_0..bias_NL, ln_gamma, ln_beta]
    size_t vec_floats = 0;
    // vec offsets (floats): start of MLP block
    size_t v_enc_edge, v_enc_node, v_dec;
    std::vector<size_t> v_edge, v_node;
    // bf16 x 3 weight streams of the training kernels (train.hip): MLP after MLP, stages of kStageFloatsB3
    float* packed_t3 = nullptr;
    size_t packed_t3_floats = 0;
    size_t t_enc_edge = 0, t_enc_node = 0;
    std::vector<size_t> t_edge, t_node;
    int T_HH = 0, T_e0 = 0, T_n0 = 0, T_out = 0;
    // The model's own copy of the raw tensors (device).  A weight update refreshes it and the cheap images (vec, the training
    // streams); the inference images (packed_hm, packed_h3) are re-packed from it by the first inference call that follows
    // (ensure_inference_images): a training loop, which updates the weights every step, never pays for them.
    float* raw = nullptr;
    std::vector<size_t> raw_off;
    size_t raw_floats = 0;
    bool infer_stale = true;
    std::mutex lazy_mu;
    // Every copy / pack of the weights is queued on the stream of the call that triggered it; `ready` is recorded behind the last
    // one.  An entry point that runs on ANOTHER stream waits for it there (weights_ready_on), so a model may be loaded on one
    // stream and used on others.  (The other direction -- a weight update while a forward on another stream still reads the old
    // images -- is the caller's to order, like any write to memory a queued kernel reads.)
    hipEvent_t ready = nullptr;
    hipStream_t ready_stream = nullptr;
};
int ensure_inference_images(const gm_model* m, hipStream_t s);   // model.hip; called by every inference entry point
int weights_ready_on(const gm_model* m, hipStream_t s);          // model.hip; called by every entry point that reads the packed weights

namespace gm {
inline int tensors_per_normed_mlp(int NL) { return 2 * (NL + 1) + 2; }
}  // namespace gm
