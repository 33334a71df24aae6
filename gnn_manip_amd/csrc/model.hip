// Model handle (packed weight image), forward orchestration and the device-resident rollout step.
// Host-side C++ only: every kernel lives in graph.hip / features.hip / mlp.hip.
#include <vector>
#include "common.h"
#include "mlp.h"
#include "model.h"
#include "train.h"
#include "hedge.h"
#include "hmlp.h"

using namespace gm;


namespace {

int check_desc(const gm_model_desc* d, const char* who) {
    GM_REQUIRE(d != nullptr, GM_ERR_INVALID_ARGUMENT, "%s: null model descriptor", who);
    // the reference's two ctor asserts (epd_gnn.py:26-27)
    GM_REQUIRE(d->num_layers >= 2, GM_ERR_INVALID_ARGUMENT, "The number of layers num_layers must be at least 2");
    GM_REQUIRE(d->m_steps >= 1, GM_ERR_INVALID_ARGUMENT, "The number of m_steps message pasting steps must be at least 1");
    GM_REQUIRE(hm_padded_hidden(d->hidden_size) > 0, GM_ERR_UNSUPPORTED,
               "%s: hidden_size=%d: supported are 1 .. 256 (run zero-padded at width 64, 128 or 256)", who, d->hidden_size);
    GM_REQUIRE(d->num_layers <= 16, GM_ERR_UNSUPPORTED, "%s: num_layers=%d: at most 16", who, d->num_layers);
    GM_REQUIRE(d->edge_dim >= 1 && d->edge_dim <= 8, GM_ERR_UNSUPPORTED, "%s: edge_dim=%d unsupported (1..8)", who, d->edge_dim);
    GM_REQUIRE(d->node_dim >= 1 && d->node_dim <= 32, GM_ERR_UNSUPPORTED, "%s: node_dim=%d unsupported (1..32)", who, d->node_dim);
    GM_REQUIRE(d->out_dim >= 1 && d->out_dim <= 4, GM_ERR_UNSUPPORTED, "%s: out_dim=%d unsupported (1..4)", who, d->out_dim);
    GM_REQUIRE(d->ln_eps > 0.f, GM_ERR_INVALID_ARGUMENT, "%s: ln_eps must be > 0", who);
    GM_REQUIRE(d->flow == 0 || d->flow == 1, GM_ERR_INVALID_ARGUMENT, "%s: flow must be 0 (aggregate at edge_index[1]) or 1", who);
    GM_REQUIRE(d->node_agg_first == 0 || d->node_agg_first == 1, GM_ERR_INVALID_ARGUMENT, "%s: node_agg_first must be 0 or 1", who);
    if (d->col_i || d->col_j || d->col_e) {
        const int a = d->col_i, b = d->col_j, c = d->col_e;
        GM_REQUIRE(a >= 0 && a < 3 && b >= 0 && b < 3 && c >= 0 && c < 3 && a != b && a != c && b != c, GM_ERR_INVALID_ARGUMENT,
                   "%s: col_i, col_j, col_e must be a permutation of 0, 1, 2", who);
    }
    return GM_OK;
}

struct FwdWs {
    float *h, *P, *agg, *side, *e;
    float* Q;   // [N][H]: the h half of the next node MLP's first Linear (systolic node path)
    size_t bytes;
};
// cap_e: rows of the latent edge array held here (0: the caller's); cap_side: edge capacity the side buffer of the
// scatter-add's head partials is sized for (hedge.h)
FwdWs carve_fwd(void* ws, int H, int64_t n, int64_t cap_e, int64_t cap_side) {
    FwdWs f;
    Carver c(ws);
    f.h = c.take<float>((size_t)n * H);
    f.P = c.take<float>((size_t)n * 2 * H);
    f.agg = c.take<float>((size_t)n * H);
    f.side = c.take<float>(edge_groups_max(n, cap_side) * H);
    f.e = c.take<float>((size_t)(cap_e > 0 ? cap_e + kEdgePadRows : 0) * H);   // + zero rows behind the list (hedge.h)
    f.Q = c.take<float>((size_t)n * H);
    f.bytes = c.used();
    return f;
}

// element count of tensor ti in the state_dict order (epd_gnn.py:63-84: per MLP  W_0, b_0, .., W_NL, b_NL [, gamma, beta])
size_t tensor_floats(const gm_model* m, int ti) {
    const int H = m->H, NL = m->NL, M = m->M;
    const int PM = tensors_per_normed_mlp(NL);
    const int n_normed = 2 + 2 * M;
    int mlp = ti / PM, r = ti % PM;
    int in_dim, out_last = H;
    if (mlp >= n_normed) { mlp = n_normed; r = ti - n_normed * PM; in_dim = H; out_last = m->d.out_dim; }
    else if (mlp == 0) in_dim = m->d.edge_dim;
    else if (mlp == 1) in_dim = m->d.node_dim;
    else in_dim = (mlp % 2 == 0) ? 3 * H : 2 * H;
    if (r >= 2 * (NL + 1)) return (size_t)H;   // LayerNorm gamma / beta
    const int l = r / 2;
    const int out = l == NL ? out_last : H, in = l == 0 ? in_dim : H;
    return (r % 2) ? (size_t)out : (size_t)out * in;
}

// (re)build the operand images + vec from the caller's tensors (device pointers)
constexpr int kPackCommon = 1;      // vec + the training streams (+ the fp32 images of development builds)
constexpr int kPackInference = 2;   // packed_hm, packed_h3 (+ the bf16 x 6 image of development builds)
int load_weights_device(gm_model* m, const float* const* T, hipStream_t s, int what) {
    // H: the width the kernels run at (hidden_size zero-padded to 64 / 128 / 256: strides, image sizes); Hv: the model's hidden_size
    const int H = m->Hp, Hv = m->H, NL = m->NL, M = m->M;
    const int PM = tensors_per_normed_mlp(NL);
    int rc = GM_OK;
    const int b_enc_edge = 0, b_enc_node = PM, b_dec = (2 + 2 * M) * PM;
    auto b_edge = [&](int k) { return (2 + 2 * k) * PM; };
    auto b_node = [&](int k) { return (3 + 2 * k) * PM; };

    if (m->packed_hm && (what & kPackInference)) {   // fp16 hi / lo images of every Linear (hmlp.h); consecutive Linears of an MLP form a scale chain
        std::vector<PackHmJob>& jobs = m->hm_jobs;
        jobs.clear();
        int prev = -1;   // job whose output feeds the next lin() (-1: the next one heads a chain)
        float head_rms = 1.f;
        auto lin = [&](int ti, int ld, int col0a, int col0b, int out_valid, int out_pad, int out_seg, int k_valid, int k_pad, int k_seg,
                       bool with_bias, int bias_n, size_t& off, int gain_col0 = 0, int gain_cols = 0) {
            PackHmJob j{};
            j.W = T[ti]; j.ld = ld; j.bias = with_bias ? T[ti + 1] : nullptr; j.bias_n = bias_n;
            j.out_valid = out_valid; j.out_pad = out_pad; j.out_seg = out_seg;
            j.k_valid = k_valid; j.k_pad = k_pad; j.k_seg = k_seg; j.col0[0] = col0a; j.col0[1] = col0b;
            j.pred = prev; j.in_rms = head_rms; j.gain_col0 = gain_col0; j.gain_cols = gain_cols;
            j.dst = m->packed_hm + off;
            off += hm_lin_floats(out_pad, k_pad);
            prev = (int)jobs.size();
            jobs.push_back(j);
        };
        auto head = [&](float rms) { prev = -1; head_rms = rms; };
        auto hh = [&](int ti, size_t& off) { lin(ti, Hv, 0, 0, Hv, H, H, Hv, H, H, true, Hv, off); };
        // Linears 1 .. NL of an MLP that ends in a LayerNorm (the decoder's are queued one by one): the last one is packed centred
        auto hidden = [&](int base, size_t& off) {
            for (int l = 1; l <= NL; ++l) hh(base + 2 * l, off);
            jobs.back().center = 1;
        };
        auto proj = [&](int k, size_t& off) {   // P = h [W_i | W_j]^T + [b1 | 0] of processor step k: a chain of its own (input h)
            head(1.f);
            lin(b_edge(k), 3 * Hv, m->ci * Hv, m->cj * Hv, Hv, 2 * H, H, Hv, H, H, true, Hv, off);
        };
        size_t off = m->hm_enc_edge;
        head(kHmRawInputRms);   // raw edge features: per-row power-of-two scale in the kernel
        lin(b_enc_edge, m->d.edge_dim, 0, 0, Hv, H, H, m->d.edge_dim, 16, 16, true, Hv, off);
        hidden(b_enc_edge, off);
        off = m->hm_enc_node;
        head(kHmRawInputRms);
        lin(b_enc_node, m->d.node_dim, 0, 0, Hv, H, H, m->d.node_dim, 32, 32, true, Hv, off);
        hidden(b_enc_node, off);
        proj(0, off);
        for (int k = 0; k < M; ++k) {
            off = m->hm_edge[k];
            head(1.f);
            // the e block; b1 lives in P_i.  Its pre-activation also takes h_i and h_j: the gain is that of the whole [H x 3H] Linear
            lin(b_edge(k), 3 * Hv, m->ce * Hv, 0, Hv, H, H, Hv, H, H, false, 0, off, 0, 3 * Hv);
            hidden(b_edge(k), off);
            off = m->hm_node[k];
            head(1.f);
            lin(b_node(k), 2 * Hv, m->ch * Hv, m->ca * Hv, Hv, H, H, Hv, 2 * H, H, true, Hv, off);
            hidden(b_node(k), off);
            if (k + 1 < M) proj(k + 1, off);
            else {
                head(1.f);
                for (int l = 0; l < NL; ++l) hh(b_dec + 2 * l, off);
                lin(b_dec + 2 * NL, Hv, 0, 0, m->d.out_dim, 32, 32, Hv, H, H, true, m->d.out_dim, off);
            }
        }
        for (int k = 0; k < M; ++k) {   // Q = h W_h^T + b1 of node step k (systolic node path): a chain of its own, input h
            off = m->hm_node_q[k];
            head(1.f);
            lin(b_node(k), 2 * Hv, m->ch * Hv, 0, Hv, H, H, Hv, H, H, true, Hv, off);
        }
        GM_REQUIRE(jobs.size() <= m->hm_jobs_cap, GM_ERR_INVALID_ARGUMENT, "model: %zu pack jobs, room for %zu", jobs.size(), m->hm_jobs_cap);
        rc = pack_hm(jobs.data(), (int)jobs.size(), static_cast<PackHmJob*>(m->hm_jobs_dev), m->hm_stats, s);
        if (rc != GM_OK) return rc;
    }

    if (what & kPackCommon) {
    // biases + LayerNorm vectors (fp32 kernels, training, LayerNorm of every kernel)
    VecJobs vj;
    vj.n = 0;
    auto flush_vec = [&]() {
        if (rc == GM_OK && vj.n > 0) rc = launch_vec_batch(vj, m->vec, s);
        vj.n = 0;
    };
    auto queue_vec = [&](const float* src, size_t off, int count, int zero_to) {
        if (vj.n == kVecJobsMax) flush_vec();
        VecJob& j = vj.job[vj.n++];
        j.src = src; j.dst_off = off; j.count = count; j.zero_to = zero_to;
    };
    auto vecs = [&](int base, bool normed, size_t voff) {  // biases (+ LN) of the MLP whose first tensor is `base`
        for (int l = 0; l <= NL; ++l) {
            const bool dec_out = !normed && l == NL;
            if (dec_out) queue_vec(T[base + 2 * l + 1], voff + (size_t)NL * H, m->d.out_dim, 32);
            else queue_vec(T[base + 2 * l + 1], voff + (size_t)l * H, Hv, Hv < H ? H : 0);
        }
        if (normed) {   // LayerNorm gamma / beta: zero on the padded features, which therefore stay exactly zero
            queue_vec(T[base + 2 * (NL + 1)], voff + (size_t)(NL + 1) * H, Hv, Hv < H ? H : 0);
            queue_vec(T[base + 2 * (NL + 1) + 1], voff + (size_t)(NL + 2) * H, Hv, Hv < H ? H : 0);
        }
    };
    vecs(b_enc_edge, true, m->v_enc_edge);
    vecs(b_enc_node, true, m->v_enc_node);
    for (int k = 0; k < M; ++k) {
        vecs(b_edge(k), true, m->v_edge[k]);
        vecs(b_node(k), true, m->v_node[k]);
    }
    vecs(b_dec, false, m->v_dec);
    flush_vec();
    if (rc != GM_OK) return rc;
    if (m->packed_t3) {   // bf16 x 3 streams of the training kernels: the forward Linears in the order the chains consume them
        PackTJobs tj;
        tj.n = 0;
        auto flush_t = [&]() {
            if (rc == GM_OK && tj.n > 0) rc = launch_pack_b3_batch(tj, m->packed_t3, s);
            tj.n = 0;
        };
        auto pack3 = [&](int ti, int out_rows, int ld, int col0, int k, size_t& off) {
            if (tj.n == kPackTJobsMax) flush_t();
            PackTJob& j = tj.job[tj.n++];
            j.W = T[ti]; j.w_rows = k; j.ld = ld; j.col0 = col0; j.ksub = out_rows; j.fwd = 1; j.dst_off = off;
            off += (size_t)layer_stages_b3(k, out_rows) * kStageFloatsB3;
        };
        auto hidden3 = [&](int base, size_t& off) { for (int l = 1; l <= NL; ++l) pack3(base + 2 * l, H, H, 0, H, off); };
        size_t off = m->t_enc_edge;
        pack3(b_enc_edge, H, m->d.edge_dim, 0, m->d.edge_dim, off);
        hidden3(b_enc_edge, off);
        off = m->t_enc_node;
        pack3(b_enc_node, H, m->d.node_dim, 0, m->d.node_dim, off);
        hidden3(b_enc_node, off);
        pack3(b_edge(0), H, 3 * H, m->ci * H, H, off);  // W_i of processor 0
        pack3(b_edge(0), H, 3 * H, m->cj * H, H, off);  // W_j
        for (int k = 0; k < M; ++k) {
            off = m->t_edge[k];
            pack3(b_edge(k), H, 3 * H, m->ce * H, H, off);  // W_e
            hidden3(b_edge(k), off);
            off = m->t_node[k];
            pack3(b_node(k), H, 2 * H, m->ch * H, H, off);  // W_h
            pack3(b_node(k), H, 2 * H, m->ca * H, H, off);  // W_agg
            hidden3(b_node(k), off);
            if (k + 1 < M) {
                pack3(b_edge(k + 1), H, 3 * H, m->ci * H, H, off);
                pack3(b_edge(k + 1), H, 3 * H, m->cj * H, H, off);
            } else {
                for (int l = 0; l < NL; ++l) pack3(b_dec + 2 * l, H, H, 0, H, off);
                pack3(b_dec + 2 * NL, m->d.out_dim, H, 0, H, off);
            }
        }
        flush_t();
        if (rc != GM_OK) return rc;
    }
    }   // kPackCommon
    if (!(what & kPackInference)) return rc;
    if (m->packed_h3 && rc == GM_OK) {  // fp16 hi / lo images of the systolic kernels: the M processor edge MLPs, then the edge encoder
        std::vector<PackH3Job> jobs((size_t)M + 1);
        for (int k = 0; k <= M; ++k) {
            PackH3Job& j = jobs[(size_t)k];
            const bool encj = k == M;
            const int b = encj ? b_enc_edge : b_edge(k);
            j.W1 = T[b]; j.W1_col0 = encj ? 0 : m->ce * H; j.W2 = T[b + 2]; j.W3 = T[b + 4];
            j.b1 = T[b + 1]; j.b2 = T[b + 3]; j.b3 = T[b + 5];
            j.gamma = T[b + 6]; j.beta = T[b + 7];
            j.enc_k1 = encj ? m->d.edge_dim : 0;
            j.dst = m->packed_h3 + (size_t)k * h3_image_floats();
        }
        rc = pack_h3(jobs.data(), m->d.edge_dim <= 16 ? M + 1 : M, s);
        std::vector<PackH3Job> nj((size_t)M);
        for (int k = 0; k < M; ++k) {   // node MLPs: the agg block of Linear 1 (its h block is Q), Linear 2, Linear 3
            PackH3Job& j = nj[(size_t)k];
            const int b = b_node(k);
            j.W1 = T[b]; j.W1_col0 = m->ca * H; j.W1_ld = 2 * H; j.W2 = T[b + 2]; j.W3 = T[b + 4];
            j.b1 = T[b + 1]; j.b2 = T[b + 3]; j.b3 = T[b + 5];
            j.gamma = T[b + 6]; j.beta = T[b + 7];
            j.enc_k1 = 0;
            j.dst = m->packed_h3 + (size_t)(M + 1 + k) * h3_image_floats();
        }
        if (rc == GM_OK) rc = pack_h3(nj.data(), M, s);
    }
    return rc;
}

// `ready` = everything queued for the model's images so far, on stream s (model.h)
int mark_ready(gm_model* m, hipStream_t s) {
    if (!m->ready) GM_HIP_CHECK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
    GM_HIP_CHECK(hipEventRecord(m->ready, s));
    m->ready_stream = s;
    return GM_OK;
}

// The raw tensors are first copied into the model's own device buffer (host- or device-resident callers alike), then the cheap
// images are packed from that copy; the inference images follow on first use (ensure_inference_images).
int load_weights(gm_model* m, const float* const* T, int nt, bool on_device, hipStream_t s) {
    GM_REQUIRE(nt == gm_model_num_tensors(&m->d), GM_ERR_INVALID_ARGUMENT, "model: expected %d tensors, got %d",
               gm_model_num_tensors(&m->d), nt);
    for (int i = 0; i < nt; ++i) GM_REQUIRE(T[i] != nullptr, GM_ERR_INVALID_ARGUMENT, "model: tensor %d is null", i);
    std::lock_guard<std::mutex> guard(m->lazy_mu);
    if (!m->raw) {
        m->raw_off.resize((size_t)nt);
        size_t total = 0;
        for (int i = 0; i < nt; ++i) { m->raw_off[(size_t)i] = total; total += (tensor_floats(m, i) + 3) & ~(size_t)3; }
        m->raw_floats = total;
        GM_HIP_CHECK(hipMalloc(&m->raw, total * sizeof(float)));
    }
    int rc = GM_OK;
    if (on_device) {
        VecJobs vj;
        vj.n = 0;
        for (int i = 0; i < nt && rc == GM_OK; ++i) {
            if (vj.n == kVecJobsMax) { rc = launch_vec_batch(vj, m->raw, s); vj.n = 0; }
            VecJob& j = vj.job[vj.n++];
            j.src = T[i]; j.dst_off = m->raw_off[(size_t)i]; j.count = (int)tensor_floats(m, i); j.zero_to = 0;
        }
        if (rc == GM_OK && vj.n > 0) rc = launch_vec_batch(vj, m->raw, s);
    } else {
        for (int i = 0; i < nt && rc == GM_OK; ++i) {
            if (hipMemcpyAsync(m->raw + m->raw_off[(size_t)i], T[i], tensor_floats(m, i) * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) {
                gm::set_error("model: copy of tensor %d to the device failed", i);
                rc = GM_ERR_HIP;
            }
        }
        if (rc == GM_OK) (void)hipStreamSynchronize(s);   // the caller's host buffers may go away when this returns
    }
    if (rc != GM_OK) return rc;
    std::vector<const float*> D((size_t)nt);
    for (int i = 0; i < nt; ++i) D[(size_t)i] = m->raw + m->raw_off[(size_t)i];
    m->infer_stale = true;
    rc = load_weights_device(m, D.data(), s, kPackCommon);
    if (rc == GM_OK) rc = mark_ready(m, s);
    return rc;
}

}  // namespace

int ensure_inference_images(const gm_model* cm, hipStream_t s) {
    gm_model* m = const_cast<gm_model*>(cm);
    std::lock_guard<std::mutex> guard(m->lazy_mu);
    // the copy / common pack (and an earlier inference pack) may have been queued on another stream: this one waits for them
    if (m->ready && s != m->ready_stream) GM_HIP_CHECK(hipStreamWaitEvent(s, m->ready, 0));
    if (!m->infer_stale) return GM_OK;
    GM_REQUIRE(m->raw, GM_ERR_INVALID_ARGUMENT, "model: no weights loaded");
    const size_t nt = m->raw_off.size();
    std::vector<const float*> D(nt);
    for (size_t i = 0; i < nt; ++i) D[i] = m->raw + m->raw_off[i];
    int rc = load_weights_device(m, D.data(), s, kPackInference);
    if (rc == GM_OK) rc = mark_ready(m, s);   // the images are complete once THIS lands: other streams wait for it
    if (rc == GM_OK) m->infer_stale = false;
    return rc;
}

int weights_ready_on(const gm_model* cm, hipStream_t s) {
    gm_model* m = const_cast<gm_model*>(cm);
    std::lock_guard<std::mutex> guard(m->lazy_mu);
    if (m->ready && s != m->ready_stream) GM_HIP_CHECK(hipStreamWaitEvent(s, m->ready, 0));
    return GM_OK;
}

extern "C" {

int gm_padded_hidden_size(int hidden_size) { return hm_padded_hidden(hidden_size); }

int gm_model_num_tensors(const gm_model_desc* d) {
    if (!d) return 0;
    return (2 + 2 * d->m_steps) * tensors_per_normed_mlp(d->num_layers) + 2 * (d->num_layers + 1);
}

int gm_model_create(const gm_model_desc* desc, const float* const* tensors, int n_tensors, int on_device, void* stream,
                    gm_model** out) {
    gm::DevGuard dev_guard((on_device && tensors) ? tensors[0] : nullptr);
    GM_REQUIRE(out && tensors, GM_ERR_INVALID_ARGUMENT, "gm_model_create: null pointer");
    int rc = check_desc(desc, "gm_model_create");
    if (rc != GM_OK) return rc;
    gm_model* m = new gm_model();
    m->d = *desc;
    m->H = desc->hidden_size;
    const int H = m->Hp = hm_padded_hidden(desc->hidden_size);   // width the kernels run at (buffers, images); m->H: the model's
    const int NL = m->NL = desc->num_layers, M = m->M = desc->m_steps;
    if (desc->col_i || desc->col_j || desc->col_e) { m->ci = desc->col_i; m->cj = desc->col_j; m->ce = desc->col_e; }
    if (desc->node_agg_first) { m->ch = 1; m->ca = 0; }
    m->legacy = m->H == 64 || m->H == 128 || m->H == 256;   // widths the fp32 training kernels (and their operand images) exist for
    {   // bf16 x 3 streams of the training kernels, one MLP after the other
        m->T_HH = layer_stages_b3(H, H);
        m->T_e0 = layer_stages_b3(desc->edge_dim, H);
        m->T_n0 = layer_stages_b3(desc->node_dim, H);
        m->T_out = layer_stages_b3(H, desc->out_dim);
        size_t t = 0;
        m->t_enc_edge = t * kStageFloatsB3; t += m->T_e0 + NL * m->T_HH;
        m->t_enc_node = t * kStageFloatsB3; t += m->T_n0 + NL * m->T_HH + 2 * m->T_HH;
        m->t_edge.resize(M);
        m->t_node.resize(M);
        for (int k = 0; k < M; ++k) {
            m->t_edge[k] = t * kStageFloatsB3; t += (NL + 1) * m->T_HH;
            m->t_node[k] = t * kStageFloatsB3; t += (NL + 2) * m->T_HH + (k + 1 < M ? 2 * m->T_HH : NL * m->T_HH + m->T_out);
        }
        m->packed_t3_floats = t * kStageFloatsB3;
    }
    size_t v = 0;
    const size_t VM = (size_t)(NL + 3) * H;
    m->v_enc_edge = v; v += VM;
    m->v_enc_node = v; v += VM;
    m->v_edge.resize(M);
    m->v_node.resize(M);
    for (int k = 0; k < M; ++k) {
        m->v_edge[k] = v; v += VM;
        m->v_node[k] = v; v += VM;
    }
    m->v_dec = v; v += (size_t)NL * H + 32;
    m->vec_floats = v;
    {   // fp16 hi / lo images (hmlp.h): every MLP, any supported size
        const size_t hh = hm_lin_floats(H, H), pj = hm_lin_floats(2 * H, H);
        size_t o = 0;
        m->hm_enc_edge = o; o += hm_lin_floats(H, 16) + NL * hh;
        m->hm_enc_node = o; o += hm_lin_floats(H, 32) + NL * hh;
        m->hm_enc_node_tail = o; o += pj;
        m->hm_edge.resize(M);
        m->hm_node.resize(M);
        m->hm_node_tail.resize(M);
        for (int k = 0; k < M; ++k) {
            m->hm_edge[k] = o; o += (NL + 1) * hh;
            m->hm_node[k] = o; o += hm_lin_floats(H, 2 * H) + NL * hh;
            m->hm_node_tail[k] = o; o += k + 1 < M ? pj : NL * hh + hm_lin_floats(32, H);
        }
        m->hm_node_q.resize(M);
        for (int k = 0; k < M; ++k) { m->hm_node_q[k] = o; o += hh; }
        m->hm_floats = o;
        m->hm_jobs_cap = (size_t)(2 + 2 * M) * (NL + 1) + 2 * M + NL + 1 + 4;
        if (hipMalloc(&m->packed_hm, m->hm_floats * sizeof(float)) != hipSuccess ||
            hipMalloc(&m->hm_jobs_dev, m->hm_jobs_cap * sizeof(PackHmJob)) != hipSuccess ||
            hipMalloc(&m->hm_stats, m->hm_jobs_cap * 4 * sizeof(float)) != hipSuccess) {
            gm::set_error("gm_model_create: hipMalloc failed");
            gm_model_destroy(m);
            return GM_ERR_HIP;
        }
    }
    if (m->H == 128 && NL == 2 && hipMalloc(&m->packed_h3, (size_t)(2 * M + 1) * h3_image_floats() * sizeof(float)) != hipSuccess) {
        gm::set_error("gm_model_create: hipMalloc failed");
        gm_model_destroy(m);
        return GM_ERR_HIP;
    }
    m->edge_kernel = 0;   // automatic; gm_model_set_edge_kernel changes it per handle (no process-wide switch)
    if ((m->legacy && hipMalloc(&m->packed_t3, m->packed_t3_floats * sizeof(float)) != hipSuccess) ||
        hipMalloc(&m->vec, m->vec_floats * sizeof(float)) != hipSuccess) {
        gm::set_error("gm_model_create: hipMalloc failed");
        gm_model_destroy(m);
        return GM_ERR_HIP;
    }
    rc = load_weights(m, tensors, n_tensors, on_device != 0, (hipStream_t)stream);
    if (rc != GM_OK) {
        gm_model_destroy(m);
        return rc;
    }
    *out = m;
    return GM_OK;
}

int gm_model_update(gm_model* m, const float* const* tensors, int n_tensors, int on_device, void* stream) {
    gm::DevGuard dev_guard((on_device && tensors) ? tensors[0] : nullptr);
    GM_REQUIRE(m && tensors, GM_ERR_INVALID_ARGUMENT, "gm_model_update: null pointer");
    return load_weights(m, tensors, n_tensors, on_device != 0, (hipStream_t)stream);
}

void gm_model_destroy(gm_model* m) {
    if (!m) return;
    if (m->packed_t3) hipFree(m->packed_t3);
    if (m->packed_h3) hipFree(m->packed_h3);
    if (m->packed_hm) hipFree(m->packed_hm);
    if (m->hm_jobs_dev) hipFree(m->hm_jobs_dev);
    if (m->hm_stats) hipFree(m->hm_stats);
    if (m->vec) hipFree(m->vec);
    if (m->raw) hipFree(m->raw);
    if (m->ready) (void)hipEventDestroy(m->ready);
    delete m->prof;
    delete m;
}

size_t gm_forward_workspace_bytes(const gm_model_desc* desc, int64_t n, int64_t cap) {
    if (!desc || n < 0 || cap < 0) return 0;
    return carve_fwd(nullptr, hm_padded_hidden(desc->hidden_size), n, cap, cap).bytes;
}

size_t gm_block_workspace_bytes(const gm_model_desc* desc, int64_t n, int64_t cap) {
    if (!desc || n < 0 || cap < 0) return 0;
    return carve_fwd(nullptr, hm_padded_hidden(desc->hidden_size), n, 0, cap).bytes;
}

}  // extern "C"

namespace {

EdgeArgs enc_edge_args(const gm_model* m, const float* edge_attr, const int* eid, const CsrHeader* hdr, int e_host, float* e_out) {
    EdgeArgs a{};
    a.hdr = hdr; a.n_edges_host = e_host; a.eid = eid;
    a.e_in = edge_attr; a.e_out = e_out; a.k1 = m->d.edge_dim;
    a.wstream_hm = m->packed_hm + m->hm_enc_edge;
    a.wstream_h3 = m->packed_h3 ? m->packed_h3 + (size_t)m->M * h3_image_floats() : nullptr;   // the encoder's image follows the steps'
    a.kernel_choice = m->edge_kernel; a.prof = m->prof;
    const float* v = m->vec + m->v_enc_edge;
    a.bias = v; a.ln_g = v + (size_t)(m->NL + 1) * m->Hp; a.ln_b = v + (size_t)(m->NL + 2) * m->Hp; a.eps = m->d.ln_eps;
    a.h_valid = m->H;
    return a;
}
EdgeArgs proc_edge_args(const gm_model* m, int k, const CsrWs& c, int64_t n, const CsrHeader* hdr, int e_host, const int* eid,
                        const float* P, const float* e_in, float* e_out, float* agg, float* side, int residual) {
    EdgeArgs a{};
    a.hdr = hdr; a.n_edges_host = e_host; a.dst = c.dst; a.src = c.src; a.eid = eid; a.eid_out = eid;
    a.P = P; a.e_in = e_in; a.e_out = e_out; a.agg = agg; a.side = side; a.residual = residual;
    a.wstream_hm = m->packed_hm + m->hm_edge[k];
    a.wstream_h3 = m->packed_h3 ? m->packed_h3 + (size_t)k * h3_image_floats() : nullptr;
    a.edge_blocks = c.blocks;
    a.n_nodes_tab = n;
    a.kernel_choice = m->edge_kernel; a.prof = m->prof;
    const float* v = m->vec + m->v_edge[k];
    a.bias = v + m->Hp;  // layer-1 bias lives in P_i
    a.ln_g = v + (size_t)(m->NL + 1) * m->Hp; a.ln_b = v + (size_t)(m->NL + 2) * m->Hp; a.eps = m->d.ln_eps;
    a.h_valid = m->H;
    return a;
}
void set_tail(const gm_model* m, NodeArgs& a, int next_edge_step /* -1: none, M: decoder */, float* P, float* out, bool sys_edge = false) {
    if (next_edge_step < 0) { a.tail = 0; return; }
    // P for a step the systolic edge kernel takes leaves the node kernel at that kernel's weight scale
    a.p_scale = (sys_edge && next_edge_step < m->M && m->packed_h3) ? edge_sys_p_scale(m->packed_h3 + (size_t)next_edge_step * h3_image_floats()) : nullptr;
    a.tail_hm = m->packed_hm + (next_edge_step == 0 ? m->hm_enc_node_tail : m->hm_node_tail[next_edge_step - 1]);
    if (next_edge_step < m->M) {
        a.tail = 1;
        a.proj_bias = m->vec + m->v_edge[next_edge_step];
        a.P_out = P;
    } else {
        a.tail = 2;
        a.dec_bias = m->vec + m->v_dec;
        a.dec_out = out;
        a.out_dim = m->d.out_dim;
    }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// agg_cleared: the caller's first launch has zeroed agg (the rollout step's StepClear); n_per_graph: nodes of ONE graph of a
// block-diagonal batch of equal-sized graphs (the kernel choice must not depend on how many graphs share the call), or n
int epd_forward_impl(const gm_model* m, const float* nodes, int64_t n, const float* edge_attr, int attr_is_csr_order,
                     const void* csr_ws, int64_t cap, float* out, void* fwd_ws, size_t fwd_ws_bytes, void* stream, bool agg_cleared,
                     int64_t n_per_graph);
}

extern "C" {

int gm_epd_forward(const gm_model* m, const float* nodes, int64_t n, const float* edge_attr, int attr_is_csr_order,
                   const void* csr_ws, int64_t cap, float* out, void* fwd_ws, size_t fwd_ws_bytes, void* stream) {
    return epd_forward_impl(m, nodes, n, edge_attr, attr_is_csr_order, csr_ws, cap, out, fwd_ws, fwd_ws_bytes, stream, false, n);
}

}  // extern "C"

namespace {
int epd_forward_impl(const gm_model* m, const float* nodes, int64_t n, const float* edge_attr, int attr_is_csr_order,
                     const void* csr_ws, int64_t cap, float* out, void* fwd_ws, size_t fwd_ws_bytes, void* stream, bool agg_cleared,
                     int64_t n_per_graph) {
    gm::DevGuard dev_guard(out);
    GM_REQUIRE(m && csr_ws && fwd_ws, GM_ERR_INVALID_ARGUMENT, "gm_epd_forward: null pointer");
    GM_REQUIRE(n >= 0 && cap >= 0 && n < ((int64_t)1 << 31) && cap < ((int64_t)1 << 31), GM_ERR_INVALID_ARGUMENT, "gm_epd_forward: sizes out of range");
    if (n == 0) return GM_OK;
    GM_REQUIRE(nodes && out && (edge_attr || cap == 0), GM_ERR_INVALID_ARGUMENT, "gm_epd_forward: null tensor");
    const int H = m->Hp, NL = m->NL, M = m->M;   // the padded width: every latent array has this row stride
    FwdWs f = carve_fwd(fwd_ws, H, n, cap, cap);
    GM_REQUIRE(fwd_ws_bytes >= f.bytes, GM_ERR_WORKSPACE, "gm_epd_forward: workspace %zu < %zu", fwd_ws_bytes, f.bytes);
    CsrWs c = carve_csr(const_cast<void*>(csr_ws), n, cap);
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_inference_images(m, s);
    if (rc != GM_OK) return rc;
    int pad_done = 0;
    {
        EdgeArgs ee = enc_edge_args(m, edge_attr, attr_is_csr_order ? nullptr : c.eid, c.hdr, 0, f.e);
        ee.zero_pad_rows = 1;
        ee.pad_rows_done = &pad_done;
        rc = launch_edge(H, NL, true, ee, cap, s);
    }
    if (rc != GM_OK) return rc;
    if (cap > 0 && !pad_done) {
        ProfScope prof(m->prof, PROF_REST, s);
        rc = zero_edge_pad_rows(c.hdr, f.e, H, s);
        if (rc != GM_OK) return rc;
    }
    NodeArgs na{};
    na.h_valid = m->H;
    na.n_nodes = (int)n; na.x_in = nodes; na.k1 = m->d.node_dim; na.h_out = f.h;
    na.wstream_hm = m->packed_hm + m->hm_enc_node; na.kernel_choice = m->edge_kernel; na.prof = m->prof;
    na.err_flags = &c.hdr->error_flags;
    const float* v = m->vec + m->v_enc_node;
    na.bias = v; na.ln_g = v + (size_t)(NL + 1) * H; na.ln_b = v + (size_t)(NL + 2) * H; na.eps = m->d.ln_eps;
    // which kernel the processor edge launches of this forward take (the same for every step: it depends on sizes and the handle's choice)
    const bool sys_edge = cap > 0 && edge_launch_is_sys(H, NL, proc_edge_args(m, 0, c, n, c.hdr, 0, nullptr, f.P, f.e, f.e, f.agg, f.side, 1), cap);
    // The systolic node path (hedge.h): graphs with enough 32-row blocks per workgroup to pipeline (or the handle's choice 7).  Its
    // node MLP takes the h half of its first Linear as Q, written with P by the projection kernel behind every step.
    const bool sys_node = sys_edge && m->packed_h3 && (m->edge_kernel == 7 || (n_per_graph > 0 ? n_per_graph : n) >= kSysNodeMinNodes);
    const size_t h3f = h3_image_floats();
    auto project = [&](int k) {   // h -> P of edge step k, Q of node step k
        ProjSysArgs pa{};
        pa.h = f.h; pa.P = f.P; pa.Q = f.Q; pa.n = (int)n; pa.flags = &c.hdr->error_flags; pa.prof = m->prof;
        pa.img_p = m->packed_hm + (k == 0 ? m->hm_enc_node_tail : m->hm_node_tail[k - 1]);
        pa.img_q = m->packed_hm + m->hm_node_q[k];
        pa.scale_p = edge_sys_p_scale(m->packed_h3 + (size_t)k * h3f);
        pa.scale_q = edge_sys_p_scale(m->packed_h3 + (size_t)(M + 1 + k) * h3f);
        return launch_proj_sys(pa, s);
    };
    if (sys_node) na.tail = 0;
    else set_tail(m, na, 0, f.P, out, sys_edge);
    rc = launch_node(H, NL, 0, na, s);
    if (rc == GM_OK && sys_node) rc = project(0);
    if (rc != GM_OK) return rc;
    // agg is zeroed once (nodes without in-edges read zeros; rows with in-edges are stored whole by every edge launch)
    if (!agg_cleared) {
        ProfScope prof(m->prof, PROF_REST, s);
        GM_HIP_CHECK(hipMemsetAsync(f.agg, 0, (size_t)n * H * sizeof(float), s));
    }
    for (int k = 0; k < M; ++k) {
        EdgeArgs ea = proc_edge_args(m, k, c, n, c.hdr, 0, nullptr, f.P, f.e, f.e, f.agg, f.side, 1);
        ea.discard_e_out = k + 1 == M;   // the decoder reads h only (epd_gnn.py:96): the last step's e + e' is never looked at
        ea.P_prescaled = sys_edge;
        rc = launch_edge(H, NL, false, ea, cap, s);
        if (rc != GM_OK) return rc;
        if (sys_node) {
            // head partials of the scatter-add into agg, the node MLP (h in place), then the next step's projections -- or the decoder
            rc = launch_agg_stitch(f.agg, f.side, carve_edge_blocks(c.blocks, n, cap), n, m->prof, s);
            NodeSysArgs ns{};
            ns.h = f.h; ns.agg = f.agg; ns.Q = f.Q; ns.h_out = f.h; ns.image = m->packed_h3 + (size_t)(M + 1 + k) * h3f;
            ns.n = (int)n; ns.flags = &c.hdr->error_flags; ns.eps = m->d.ln_eps; ns.prof = m->prof;
            if (rc == GM_OK) rc = launch_node_sys(ns, s);
            if (rc == GM_OK && k + 1 < M) rc = project(k + 1);
            if (rc == GM_OK && k + 1 == M) {
                NodeArgs d{};
                d.h_valid = m->H; d.n_nodes = (int)n; d.x_in = f.h; d.err_flags = &c.hdr->error_flags;
                d.kernel_choice = m->edge_kernel; d.prof = m->prof;
                set_tail(m, d, M, f.P, out);
                rc = launch_node(H, NL, 3, d, s);
            }
            if (rc != GM_OK) return rc;
            continue;
        }
        NodeArgs a{};
        a.h_valid = m->H;
        a.n_nodes = (int)n; a.x_in = f.h; a.agg = f.agg; a.h_out = f.h; a.residual = 1;
        a.edge_blocks = c.blocks; a.n_nodes_tab = n; a.edge_capacity_tab = cap; a.side = f.side;
        a.err_flags = &c.hdr->error_flags;
        a.wstream_hm = m->packed_hm + m->hm_node[k]; a.kernel_choice = m->edge_kernel; a.prof = m->prof;
        const float* vn = m->vec + m->v_node[k];
        a.bias = vn; a.ln_g = vn + (size_t)(NL + 1) * H; a.ln_b = vn + (size_t)(NL + 2) * H; a.eps = m->d.ln_eps;
        set_tail(m, a, k + 1, f.P, out, sys_edge);
        rc = launch_node(H, NL, 1, a, s);
        if (rc != GM_OK) return rc;
    }
    return GM_OK;
}
}  // namespace

extern "C" {

int gm_graph_independent_forward(const gm_model* m, const float* x, int64_t n, const float* edge_attr, int64_t e,
                                 float* h_out, float* e_out, void* stream) {
    gm::DevGuard dev_guard(x ? (const void*)x : (const void*)edge_attr);
    GM_REQUIRE(m, GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_forward: null model");
    GM_REQUIRE(n >= 0 && e >= 0 && n < ((int64_t)1 << 31) && e < ((int64_t)1 << 31), GM_ERR_INVALID_ARGUMENT, "sizes out of range");
    GM_REQUIRE((n == 0 || (x && h_out)) && (e == 0 || (edge_attr && e_out)), GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_forward: null tensor");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_inference_images(m, s);
    if (rc != GM_OK) return rc;
    rc = launch_edge(m->Hp, m->NL, true, enc_edge_args(m, edge_attr, nullptr, nullptr, (int)e, e_out), e, s);
    if (rc != GM_OK) return rc;
    NodeArgs na{};
    na.h_valid = m->H;
    na.n_nodes = (int)n; na.x_in = x; na.k1 = m->d.node_dim; na.h_out = h_out;
    na.wstream_hm = m->packed_hm + m->hm_enc_node; na.kernel_choice = m->edge_kernel; na.prof = m->prof;
    const float* v = m->vec + m->v_enc_node;
    na.bias = v; na.ln_g = v + (size_t)(m->NL + 1) * m->Hp; na.ln_b = v + (size_t)(m->NL + 2) * m->Hp; na.eps = m->d.ln_eps;
    na.tail = 0; na.h_valid = m->H;
    return launch_node(m->Hp, m->NL, 0, na, s);
}

int gm_interaction_network_forward(const gm_model* m, int k, const float* h, int64_t n, const float* e,
                                   const void* csr_ws, int64_t cap, float* h_out, float* e_out, void* fwd_ws,
                                   size_t fwd_ws_bytes, void* stream) {
    gm::DevGuard dev_guard(h);
    GM_REQUIRE(m && csr_ws && fwd_ws, GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_forward: null pointer");
    GM_REQUIRE(k >= 0 && k < m->M, GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_forward: block %d out of range", k);
    GM_REQUIRE(n >= 0 && cap >= 0 && n < ((int64_t)1 << 31) && cap < ((int64_t)1 << 31), GM_ERR_INVALID_ARGUMENT, "sizes out of range");
    if (n == 0) return GM_OK;
    GM_REQUIRE(h && h_out && (cap == 0 || (e && e_out)), GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_forward: null tensor");
    const int H = m->Hp, NL = m->NL;
    FwdWs f = carve_fwd(fwd_ws, H, n, 0, cap);  // P, agg and the side buffer are used
    GM_REQUIRE(fwd_ws_bytes >= f.bytes, GM_ERR_WORKSPACE, "gm_interaction_network_forward: workspace %zu < %zu", fwd_ws_bytes, f.bytes);
    CsrWs c = carve_csr(const_cast<void*>(csr_ws), n, cap);
    hipStream_t s = (hipStream_t)stream;
    {
        const int rc_img = ensure_inference_images(m, s);
        if (rc_img != GM_OK) return rc_img;
    }
    // projection P = h [W_i | W_j]^T (+ b1): the tail section of the preceding node stream
    NodeArgs pa{};
    pa.h_valid = m->H;
    pa.n_nodes = (int)n; pa.x_in = h;
    pa.kernel_choice = m->edge_kernel; pa.prof = m->prof;
    pa.err_flags = &c.hdr->error_flags;
    set_tail(m, pa, k, f.P, nullptr);
    int rc = launch_node(H, NL, 2, pa, s);
    if (rc != GM_OK) return rc;
    GM_HIP_CHECK(hipMemsetAsync(f.agg, 0, (size_t)n * H * sizeof(float), s));
    rc = launch_edge(H, NL, false, proc_edge_args(m, k, c, n, c.hdr, 0, c.eid, f.P, e, e_out, f.agg, f.side, 0), cap, s);
    if (rc != GM_OK) return rc;
    NodeArgs a{};
    a.h_valid = m->H;
    a.n_nodes = (int)n; a.x_in = h; a.agg = f.agg; a.h_out = h_out; a.residual = 0;
    a.edge_blocks = c.blocks; a.n_nodes_tab = n; a.edge_capacity_tab = cap; a.side = f.side;
    a.err_flags = &c.hdr->error_flags;
    a.wstream_hm = m->packed_hm + m->hm_node[k]; a.kernel_choice = m->edge_kernel; a.prof = m->prof;
    const float* vn = m->vec + m->v_node[k];
    a.bias = vn; a.ln_g = vn + (size_t)(NL + 1) * H; a.ln_b = vn + (size_t)(NL + 2) * H; a.eps = m->d.ln_eps;
    a.tail = 0;
    return launch_node(H, NL, 1, a, s);
}

// ------------------------------------------------------------------------------------------
// rollout step
// ------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
struct RolloutWs {
    void *graph, *csr, *fwd;
    float *x, *edge_attr, *pred;
    size_t graph_bytes, csr_bytes, fwd_bytes, bytes;
};
RolloutWs carve_rollout(void* ws, const gm_model_desc* d, int64_t n, int K) {
    RolloutWs r;
    const int64_t cap = n * K;
    r.graph_bytes = gm_graph_workspace_bytes(n, K);
    r.csr_bytes = gm_csr_workspace_bytes(n, cap);
    r.fwd_bytes = gm_forward_workspace_bytes(d, n, cap);
    Carver c(ws);
    r.graph = c.take<char>(r.graph_bytes);
    r.csr = c.take<char>(r.csr_bytes);
    r.fwd = c.take<char>(r.fwd_bytes);
    r.x = c.take<float>((size_t)n * 32);
    r.edge_attr = c.take<float>((size_t)cap * 4);
    r.pred = c.take<float>((size_t)n * 4);
    r.bytes = c.used();
    return r;
}
// scratch of a renumbered rollout (gm_rollout, renumber_every > 0): two copies of the state (a re-ordering reads one and writes the
// other), the row maps that go with them, the order and the rigid ranks of the current copy
struct RenumberWs {
    float* state[2];
    int* total[2];
    int *perm, *rank;
    size_t bytes;
};
RenumberWs carve_renumber(void* ws, const gm_feature_desc* fd, int64_t n) {
    RenumberWs w;
    Carver c(ws);
    const size_t st = (size_t)fd->k_steps * n * fd->data_dim;
    w.state[0] = c.take<float>(st);
    w.state[1] = c.take<float>(st);
    w.total[0] = c.take<int>(n);
    w.total[1] = c.take<int>(n);
    w.perm = c.take<int>(n);
    w.rank = c.take<int>(n);
    w.bytes = c.used();
    return w;
}
}  // namespace

extern "C" {

int gm_model_set_edge_kernel(gm_model* m, int choice) {
    GM_REQUIRE(m, GM_ERR_INVALID_ARGUMENT, "gm_model_set_edge_kernel: null model");
    GM_REQUIRE(choice >= 0 && choice <= 7, GM_ERR_INVALID_ARGUMENT, "gm_model_set_edge_kernel: choice %d out of range", choice);
    GM_REQUIRE(choice == 0 || choice >= 5, GM_ERR_UNSUPPORTED,
               "gm_model_set_edge_kernel: choices 1..4 (the round-1 fp32 / bf16 x 6 kernels) were removed from the library (round 5)");
    GM_REQUIRE((choice != 5 && choice != 7) || m->packed_h3, GM_ERR_UNSUPPORTED, "gm_model_set_edge_kernel: the systolic kernel is for hidden_size 128, num_layers 2");
    m->edge_kernel = choice;
    return GM_OK;
}

int gm_model_profile(gm_model* m, int kind_mask) {
    GM_REQUIRE(m, GM_ERR_INVALID_ARGUMENT, "gm_model_profile: null model");
    if (!m->prof) {
        if (!kind_mask) return GM_OK;
        m->prof = new ProfState();
    }
    for (int k = 0; k < PROF_KINDS; ++k)
        if (((kind_mask & ~m->prof->mask) >> k) & 1) m->prof->count[k] = m->prof->dropped[k] = 0;  // newly enabled kinds start from zero
    m->prof->mask = kind_mask;
    return GM_OK;
}

int gm_model_profile_query(const gm_model* m, int kind, int64_t* launches, double* total_ms) {
    GM_REQUIRE(m && kind >= 0 && kind < PROF_KINDS && launches && total_ms, GM_ERR_INVALID_ARGUMENT, "gm_model_profile_query: bad argument");
    *launches = 0;
    *total_ms = 0.0;
    const ProfState* p = m->prof;
    if (!p) return GM_OK;
    // scopes beyond the kind's PROF_MAX event pairs were not timed: the count comes back NEGATIVE (minus the scopes opened) and the
    // total covers the first PROF_MAX only -- a caller that divides by steps must not use it
    *launches = p->dropped[kind] ? -(int64_t)(p->count[kind] + p->dropped[kind]) : p->count[kind];
    for (int i = 0; i < p->count[kind]; ++i) {
        GM_HIP_CHECK(hipEventSynchronize(p->stop[kind][i]));
        float ms = 0.f;
        GM_HIP_CHECK(hipEventElapsedTime(&ms, p->start[kind][i], p->stop[kind][i]));
        *total_ms += ms;
    }
    return GM_OK;
}

size_t gm_rollout_workspace_bytes(const gm_model_desc* desc, int64_t n, int K) {
    if (!desc || n < 0 || K < 1) return 0;
    return carve_rollout(nullptr, desc, n, K).bytes;
}

int gm_rollout_step(const gm_model* m, float* obs, int64_t n, const gm_feature_desc* fd, int K, const int32_t* rigid_rank,
                    const float* rigid_target, float* pred_acc_out, void* ws, size_t ws_bytes, void* stream) {
    gm::DevGuard dev_guard(obs);
    GM_REQUIRE(m && obs && fd && ws, GM_ERR_INVALID_ARGUMENT, "gm_rollout_step: null pointer");
    const int F = 3 * (fd->k_steps - 1) + 7 + (fd->control_col >= 0 ? 3 : 0);
    GM_REQUIRE(m->d.node_dim == F, GM_ERR_INVALID_ARGUMENT, "gm_rollout_step: model node_dim=%d but features give %d", m->d.node_dim, F);
    GM_REQUIRE(m->d.edge_dim == 4 && m->d.out_dim == 3, GM_ERR_INVALID_ARGUMENT, "gm_rollout_step: needs edge_dim=4, out_dim=3 (3-D scene)");
    GM_REQUIRE(n < ((int64_t)1 << 31) / (K > 0 ? K : 1), GM_ERR_UNSUPPORTED, "gm_rollout_step: n*max_neighbours overflows int32");
    RolloutWs r = carve_rollout(ws, &m->d, n, K);
    GM_REQUIRE(ws_bytes >= r.bytes, GM_ERR_WORKSPACE, "gm_rollout_step: workspace %zu < %zu", ws_bytes, r.bytes);
    if (n == 0) return GM_OK;
    const int64_t cap = n * K;
    int rc;
    hipStream_t hs = (hipStream_t)stream;
    // state_pre + node features in one launch -- which also resets what the step's later launches build on: the graph and
    // destination-sort workspaces (headers, cell counts, in-degrees, scan states, stitch table) and the forward's agg rows
    {
        StepClear clr;
        graph_clear_jobs(clr, carve_graph(r.graph, n, K), n);
        csr_clear_jobs(clr, carve_csr(r.csr, n, cap), n, m->d.flow);
        FwdWs f = carve_fwd(r.fwd, m->Hp, n, cap, cap);
        clr.add(reinterpret_cast<int*>(f.agg), (long long)n * m->Hp, 0);
        ProfScope prof(m->prof, PROF_REST, hs);
        rc = gm::rollout_pre_features(obs, n, fd, rigid_rank, rigid_target, r.x, hs, &clr);
    }
    if (rc != GM_OK) return rc;
    const float* last_pos = obs + (size_t)(fd->k_steps - 1) * n * fd->data_dim + fd->cart_col;
    {   // the in-degree count of the destination sort rides in the neighbour search
        ProfScope prof(m->prof, PROF_GRAPH, hs);
        rc = gm::radius_graph_build_fused(last_pos, fd->data_dim, n, fd->nodes_per_graph > 0 ? fd->nodes_per_graph : n, fd->conn_r, K,
                                          r.graph, r.graph_bytes, r.csr, r.csr_bytes, m->d.flow, hs);
    }
    if (rc != GM_OK) return rc;
    // destination sort; the edge features and the block tables are written by the same pass that fixes each segment's order
    {
        ProfScope prof(m->prof, PROF_REST, hs);
        rc = gm::csr_from_graph_fused(r.graph, n, K, r.csr, r.csr_bytes, last_pos, fd->data_dim, (float)fd->conn_r, r.edge_attr,
                                      m->d.flow, hs);
    }
    if (rc != GM_OK) return rc;
    rc = epd_forward_impl(m, r.x, n, r.edge_attr, 1, r.csr, cap, r.pred, r.fwd, r.fwd_bytes, stream, true,
                          fd->nodes_per_graph > 0 ? fd->nodes_per_graph : n);
    if (rc != GM_OK) return rc;
    // integrator + window shift + write-back (+ copy of the prediction) in one launch
    ProfScope prof(m->prof, PROF_REST, hs);
    return gm::rollout_integrate_post(obs, n, fd, r.pred, rigid_rank, rigid_target, pred_acc_out, hs);
}

size_t gm_rollout_renumber_workspace_bytes(const gm_feature_desc* fd, int64_t n) {
    if (!fd || n < 0 || fd->k_steps < 1 || fd->data_dim < 1) return 0;
    return carve_renumber(nullptr, fd, n).bytes;
}

int gm_rollout(const gm_model* m, float* obs, int64_t n, const gm_feature_desc* fd, int K, const int32_t* rigid_rank,
               const float* rigid_targets, int64_t n_targets, int64_t n_rigid, int64_t steps, float* record_last,
               int64_t renumber_every, void* renumber_ws, size_t renumber_ws_bytes, void* ws, size_t ws_bytes, void* stream) {
    gm::DevGuard dev_guard(obs);
    GM_REQUIRE(m && obs && fd && ws, GM_ERR_INVALID_ARGUMENT, "gm_rollout: null pointer");
    GM_REQUIRE(steps >= 0 && n_targets >= 0 && n_rigid >= 0 && renumber_every >= 0, GM_ERR_INVALID_ARGUMENT, "gm_rollout: negative count");
    // a scene without rigid rows has an empty trajectory ([steps, 0, 3]: a null pointer); the reference's loop runs on it unchanged
    GM_REQUIRE(rigid_targets || n_targets == 0 || n_rigid == 0, GM_ERR_INVALID_ARGUMENT, "gm_rollout: n_targets > 0 without rigid_targets");
    GM_REQUIRE(!rigid_targets || rigid_rank, GM_ERR_INVALID_ARGUMENT, "gm_rollout: rigid_targets need rigid_rank");
    hipStream_t hs = (hipStream_t)stream;
    const size_t frame = (size_t)n * fd->data_dim;
    const bool renum = renumber_every > 0 && n > 0 && steps > 0;
    RenumberWs rw{};
    RolloutWs r{};
    if (renum) {
        GM_REQUIRE(renumber_ws, GM_ERR_INVALID_ARGUMENT, "gm_rollout: renumber_every > 0 needs renumber_ws (gm_rollout_renumber_workspace_bytes)");
        rw = carve_renumber(renumber_ws, fd, n);
        GM_REQUIRE(renumber_ws_bytes >= rw.bytes, GM_ERR_WORKSPACE, "gm_rollout: renumber workspace %zu < %zu", renumber_ws_bytes, rw.bytes);
        GM_REQUIRE(n < ((int64_t)1 << 31) / (K > 0 ? K : 1), GM_ERR_UNSUPPORTED, "gm_rollout: n*max_neighbours overflows int32");
        r = carve_rollout(ws, &m->d, n, K);
        GM_REQUIRE(ws_bytes >= r.bytes, GM_ERR_WORKSPACE, "gm_rollout: workspace %zu < %zu", ws_bytes, r.bytes);
    }
    // The renumbered rollout works on a copy of the state whose rows are in grid-cell order (gm::cell_order: the radius graph's own
    // grid; scenes of a batch stay apart), re-ordered every `renumber_every` steps: state[j] is the caller's row total[j], its rigid
    // rank is the caller's (rank values only index the pose arrays: any order does), records and the final state go back through
    // `total`.  A radius graph does not depend on the numbering and every per-node / per-edge function is numbering-free, so what
    // changes is the order in which a node's incoming messages are summed: float32 rounding (tests: 2e-6 of the plain engine).
    float* state = obs;            // the array the steps run on
    const int* rank = rigid_rank;
    const int* total = nullptr;    // row of the caller's state that row j of `state` is (nullptr: the identity)
    int flip = 0;
    for (int64_t i = 0; i < steps; ++i) {
        if (renum && i % renumber_every == 0) {
            const float* last_pos = state + (size_t)(fd->k_steps - 1) * frame + fd->cart_col;
            int rc = gm::cell_order(last_pos, fd->data_dim, n, fd->nodes_per_graph > 0 ? fd->nodes_per_graph : n, fd->conn_r, K, r.graph,
                                    r.graph_bytes, rw.perm, hs);
            if (rc != GM_OK) return rc;
            rc = gm::renumber_gather(state, rw.state[flip], fd->k_steps, n, fd->data_dim, rw.perm, r.graph, total, rw.total[flip],
                                     rigid_rank, rigid_rank ? rw.rank : nullptr, hs);
            if (rc != GM_OK) return rc;
            state = rw.state[flip];
            total = rw.total[flip];
            rank = rigid_rank ? rw.rank : nullptr;
            flip ^= 1;
        }
        float* last = state + (size_t)(fd->k_steps - 1) * frame;
        // traj_utils.py:126-134: steps past the scripted trajectory keep the rigid body where it is (control = 0 displacement)
        const float* target = i < n_targets ? rigid_targets + (size_t)i * n_rigid * 3 : nullptr;
        if (record_last) {  // the reference records the last frame after the control overwrite (rollout_utils.py:49, traj_utils.py:137)
            int rc = gm_state_pre(state, n, fd, rank, target, stream);
            if (rc != GM_OK) return rc;
            if (total) rc = gm::renumber_scatter(last, record_last + (size_t)i * frame, 1, n, fd->data_dim, total, hs);
            else GM_HIP_CHECK(hipMemcpyAsync(record_last + (size_t)i * frame, last, frame * sizeof(float), hipMemcpyDeviceToDevice, hs));
            if (rc != GM_OK) return rc;
        }
        int rc = gm_rollout_step(m, state, n, fd, K, rank, target, nullptr, ws, ws_bytes, stream);
        if (rc != GM_OK) return rc;
    }
    if (total) return gm::renumber_scatter(state, obs, fd->k_steps, n, fd->data_dim, total, hs);
    return GM_OK;
}

int gm_rollout_status(const void* ws, const gm_model_desc* desc, int64_t n, int K, int64_t* n_edges_host, void* stream) {
    gm::DevGuard dev_guard(ws);
    GM_REQUIRE(ws && desc && n_edges_host, GM_ERR_INVALID_ARGUMENT, "gm_rollout_status: null pointer");
    RolloutWs r = carve_rollout(const_cast<void*>(ws), desc, n, K);
    int rc = gm_radius_graph_num_edges(r.graph, n_edges_host, stream);
    if (rc != GM_OK) return rc;
    int64_t e2 = 0;
    return gm_csr_num_edges(r.csr, &e2, stream);
}

}  // extern "C"
