// Training entry points: forward with activation tape and backward of the whole
// encode-process-decode model (host orchestration; kernels in train.hip / mlp.hip).
// Reference call sites: examples/train_dyn.py:45-72 (model.forward -> loss.backward -> Adam),
// gnn_manip/models/epd_gnn.py:86-105 (forward wiring the gradients flow back through).
#include <vector>
#include "common.h"
#include "mlp.h"
#include "model.h"
#include "train.h"

using namespace gm;

namespace {

struct Tape {
    void *csr_dst, *csr_src;
    size_t csr_bytes;
    int64_t* ei2;
    std::vector<float*> h, e, agg;  // block inputs: h[0..M], e[0..M], agg[0..M-1]
    float* P;
    TapePtr ee, en, dec;
    std::vector<TapePtr> te, tn;
    size_t bytes;
};

TapePtr take_tape(Carver& c, int64_t rows, int H, int NL, bool normed) {
    TapePtr t{};
    t.a = c.take<float>((size_t)NL * rows * H);   // post-ReLU outputs of Linear 1 .. NL, one [rows][H] array each
    t.mask = c.take<uint32_t>((size_t)NL * rows * (H / 32));
    if (normed) {
        t.xhat = c.take<float>((size_t)rows * H);
        t.rstd = c.take<float>((size_t)rows);
    }
    return t;
}

Tape carve_tape(void* ws, const gm_model_desc* d, int64_t n, int64_t e) {
    Tape t;
    const int H = d->hidden_size, M = d->m_steps, NL = d->num_layers;
    Carver c(ws);
    t.csr_bytes = gm_csr_workspace_bytes(n, e);
    t.csr_dst = c.take<char>(t.csr_bytes);
    t.csr_src = c.take<char>(t.csr_bytes);
    t.ei2 = c.take<int64_t>((size_t)2 * e);
    t.h.resize(M + 1);
    t.e.resize(M + 1);
    t.agg.resize(M);
    for (int k = 0; k <= M; ++k) t.h[k] = c.take<float>((size_t)n * H);
    for (int k = 0; k <= M; ++k) t.e[k] = c.take<float>((size_t)e * H);
    for (int k = 0; k < M; ++k) t.agg[k] = c.take<float>((size_t)n * H);
    t.P = c.take<float>((size_t)n * 2 * H);
    t.ee = take_tape(c, e, H, NL, true);
    t.en = take_tape(c, n, H, NL, true);
    t.te.resize(M);
    t.tn.resize(M);
    for (int k = 0; k < M; ++k) {
        t.te[k] = take_tape(c, e, H, NL, true);
        t.tn[k] = take_tape(c, n, H, NL, true);
    }
    t.dec = take_tape(c, n, H, NL, false);
    t.bytes = c.used();
    return t;
}

struct BwdWs {
    float *packT, *dz, *dzn, *de, *dh, *dagg, *Gi, *Gj, *part;
    float* go;          // the upstream gradient as the chains read it: zero when the forward's edge_index was flagged (gate_grad_out_kernel)
    size_t dz_stride;   // floats between dz_l and dz_(l+1) (l = 1 .. NL + 1)
    float* dzl(int l) const { return dz + (size_t)(l - 1) * dz_stride; }
    // the node-sized chains of the model backward (node MLPs, node encoder) leave their dz in a set of their own, so that the
    // weight-gradient jobs over an edge chain's dz and those over the node chain's that follows run as ONE batch
    size_t dzn_stride;
    float* dznl(int l) const { return dzn + (size_t)(l - 1) * dzn_stride; }
    size_t off_dec, off_enc_edge, off_enc_node;
    std::vector<size_t> off_edge, off_node;
    size_t bytes;
};

BwdWs carve_bwd(void* ws, const gm_model_desc* d, int64_t n, int64_t e) {
    BwdWs b;
    const int H = d->hidden_size, M = d->m_steps, NL = d->num_layers;
    const size_t U = (size_t)layer_stages_b3(H, H) * kStageFloatsB3;  // one HxH unit
    size_t off = 0;
    b.off_dec = off; off += (size_t)layer_stages_b3(d->out_dim, H) * kStageFloatsB3 + (size_t)NL * U;
    b.off_edge.resize(M);
    b.off_node.resize(M);
    for (int k = 0; k < M; ++k) {
        b.off_node[k] = off; off += (k + 1 < M ? 2 * U : 0) + (size_t)(NL + 2) * U;
        b.off_edge[k] = off; off += (size_t)(NL + 1) * U;
    }
    const size_t UIN = (size_t)layer_stages_b3(H, 32) * kStageFloatsB3;  // W1^T of an encoder (input gradient, block API)
    b.off_enc_node = off; off += 2 * U + (size_t)NL * U + UIN;
    b.off_enc_edge = off; off += (size_t)NL * U + UIN;
    Carver c(ws);
    b.packT = c.take<float>(off);
    const int64_t R = n > e ? n : e;
    b.dz_stride = align_up((size_t)R * H, 64);
    b.dz = c.take<float>((size_t)(NL + 1) * b.dz_stride);
    b.dzn_stride = align_up((size_t)n * H, 64);
    b.dzn = c.take<float>((size_t)(NL + 1) * b.dzn_stride);
    b.de = c.take<float>((size_t)e * H);
    b.dh = c.take<float>((size_t)n * H);
    b.dagg = c.take<float>((size_t)n * H);
    b.Gi = c.take<float>((size_t)n * H);
    b.Gj = c.take<float>((size_t)n * H);
    b.part = c.take<float>(wgrad_partial_floats(H));
    b.go = c.take<float>((size_t)n * (d->out_dim > 0 ? d->out_dim : 1));
    b.bytes = c.used();
    return b;
}

// A training forward does not synchronise: an edge_index entry outside [0, n) is flagged by the destination sort in the tape's CSR
// headers and reported at a later forward / status() (epd_gnn.py).  Until then the step must not do damage: the flagged forward's
// output is NaN (the loss shows it) and its backward produces exactly zero gradients (the upstream gradient is gated to zero), so
// the optimiser step that runs before the error surfaces leaves the weights where a raise at the forward would have left them.
__global__ void __launch_bounds__(256) poison_if_flagged_kernel(const CsrHeader* a, const CsrHeader* b, float* out, size_t count) {
    if (!((a->error_flags | b->error_flags) & ERRF_BAD_EDGE_INDEX)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) out[i] = __builtin_nanf("");
}
__global__ void __launch_bounds__(256) gate_grad_out_kernel(const CsrHeader* a, const CsrHeader* b, const float* g, float* out, size_t count) {
    const bool bad = ((a->error_flags | b->error_flags) & ERRF_BAD_EDGE_INDEX) != 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) out[i] = bad ? 0.f : g[i];
}
unsigned small_grid(size_t count) {
    const size_t g = (count + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 512 ? 512 : g));
}

int check_sizes(const gm_model* m, int64_t n, int64_t e, const char* who) {
    GM_REQUIRE(m, GM_ERR_INVALID_ARGUMENT, "%s: null model", who);
    GM_REQUIRE(n >= 1 && e >= 0 && n < ((int64_t)1 << 31) && e < ((int64_t)1 << 31) / m->H, GM_ERR_INVALID_ARGUMENT,
               "%s: sizes out of range (n=%lld, e=%lld)", who, (long long)n, (long long)e);
    return GM_OK;
}

const float* mlp_vec(const gm_model* m, size_t voff) { return m->vec + voff; }

// ---- tapes of the two standalone blocks (torch_graphnet API surface)
struct GiTape {
    TapePtr ee, en;
    size_t bytes;
};
GiTape carve_gi_tape(void* ws, int H, int NL, int64_t n, int64_t e) {
    GiTape t;
    Carver c(ws);
    t.ee = take_tape(c, e, H, NL, true);
    t.en = take_tape(c, n, H, NL, true);
    t.bytes = c.used();
    return t;
}
struct InTape {
    void *csr_dst, *csr_src;
    size_t csr_bytes;
    int64_t* ei2;
    float *P, *agg;
    TapePtr te, tn;
    size_t bytes;
};
InTape carve_in_tape(void* ws, int H, int NL, int64_t n, int64_t e) {
    InTape t;
    Carver c(ws);
    t.csr_bytes = gm_csr_workspace_bytes(n, e);
    t.csr_dst = c.take<char>(t.csr_bytes);
    t.csr_src = c.take<char>(t.csr_bytes);
    t.ei2 = c.take<int64_t>((size_t)2 * e);
    t.P = c.take<float>((size_t)n * 2 * H);
    t.agg = c.take<float>((size_t)n * H);
    t.te = take_tape(c, e, H, NL, true);
    t.tn = take_tape(c, n, H, NL, true);
    t.bytes = c.used();
    return t;
}
// backward scratch of a single block: same carve as the whole model with one processor step
BwdWs carve_block_bwd(void* ws, const gm_model_desc* d, int64_t n, int64_t e) {
    gm_model_desc d1 = *d;
    d1.m_steps = 1;
    return carve_bwd(ws, &d1, n, e);
}

}  // namespace

extern "C" {

size_t gm_train_tape_bytes(const gm_model_desc* desc, int64_t n, int64_t e) {
    if (!desc || n < 0 || e < 0) return 0;
    return carve_tape(nullptr, desc, n, e).bytes;
}

size_t gm_train_backward_workspace_bytes(const gm_model_desc* desc, int64_t n, int64_t e) {
    if (!desc || n < 0 || e < 0) return 0;
    return carve_bwd(nullptr, desc, n, e).bytes;
}

int gm_epd_forward_train(const gm_model* m, const float* nodes, int64_t n, const float* edge_attr, const int64_t* edge_index,
                         int64_t e, float* out, void* tape, size_t tape_bytes, void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_epd_forward_train");
    gm::DevGuard dev_guard(nodes);
    int rc = check_sizes(m, n, e, "gm_epd_forward_train");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(nodes && out && tape && (e == 0 || (edge_attr && edge_index)), GM_ERR_INVALID_ARGUMENT, "gm_epd_forward_train: null pointer");
    const int H = m->H, NL = m->NL, M = m->M;
    Tape t = carve_tape(tape, &m->d, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_epd_forward_train: tape %zu < %zu", tape_bytes, t.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    // destination-sorted edges (aggregation index i = edge_index[1]) and the source-grouped view of the sorted list
    rc = gm::csr_from_edge_index(edge_index, n, e, m->d.flow, t.csr_dst, t.csr_bytes, false, (hipStream_t)stream);
    if (rc != GM_OK) return rc;
    CsrWs c = carve_csr(t.csr_dst, n, e);
    rc = launch_swap_index(c.src, e, t.ei2, s);
    if (rc != GM_OK) return rc;
    rc = gm::csr_from_edge_index(t.ei2, n, e, 0, t.csr_src, t.csr_bytes, false, (hipStream_t)stream);
    if (rc != GM_OK) return rc;

    const size_t U = (size_t)m->T_HH * kStageFloatsB3;
    auto normed = [&](TrainFwdArgs& a, size_t voff) {
        const float* v = mlp_vec(m, voff);
        a.bias = v; a.bias_tail = v + H; a.ln_g = v + (size_t)(NL + 1) * H; a.ln_b = v + (size_t)(NL + 2) * H; a.eps = m->d.ln_eps; a.nl = NL;
    };
    {
        TrainFwdArgs a{};
        a.rows = (int)e; a.x_in = edge_attr; a.rowidx = c.eid; a.k1 = m->d.edge_dim; a.wstream = m->packed_t3 + m->t_enc_edge;
        normed(a, m->v_enc_edge);
        a.tape = t.ee; a.out = t.e[0];
        rc = launch_train_fwd(H, TK_ENC_EDGE, a, s);
        if (rc != GM_OK) return rc;
    }
    {
        TrainFwdArgs a{};
        a.rows = (int)n; a.x_in = nodes; a.k1 = m->d.node_dim; a.wstream = m->packed_t3 + m->t_enc_node;
        normed(a, m->v_enc_node);
        a.tape = t.en; a.out = t.h[0];
        // tail: P = h_0 [W_i | W_j]^T (+ b1) of the first edge step -- the projection section that follows this MLP in its stream
        a.P_out = t.P; a.proj_bias = m->vec + m->v_edge[0];
        rc = launch_train_fwd(H, TK_ENC_NODE, a, s);
        if (rc != GM_OK) return rc;
    }
    for (int k = 0; k < M; ++k) {
        const float* ve = mlp_vec(m, m->v_edge[k]);
        {
            TrainFwdArgs a{};
            a.rows = (int)e; a.x_in = t.e[k]; a.dst = c.dst; a.src = c.src; a.P = t.P; a.wstream = m->packed_t3 + m->t_edge[k];
            normed(a, m->v_edge[k]);
            a.tape = t.te[k]; a.out = t.e[k + 1]; a.residual = 1;
            rc = launch_train_fwd(H, TK_PROC_EDGE, a, s);
            if (rc != GM_OK) return rc;
        }
        // agg_i = sum over edges into i of e' = gamma * sum xhat + deg * beta
        rc = launch_segment_sum(H, c.in_ptr, nullptr, t.te[k].xhat, ve + (size_t)(NL + 1) * H, ve + (size_t)(NL + 2) * H, t.agg[k], n, s);
        if (rc != GM_OK) return rc;
        {
            TrainFwdArgs a{};
            a.rows = (int)n; a.x_in = t.h[k]; a.agg = t.agg[k]; a.wstream = m->packed_t3 + m->t_node[k];
            normed(a, m->v_node[k]);
            a.tape = t.tn[k]; a.out = t.h[k + 1]; a.residual = 1;
            if (k + 1 < M) { a.P_out = t.P; a.proj_bias = m->vec + m->v_edge[k + 1]; }   // tail: the next edge step's P (after the decoder's h_M: none)
            rc = launch_train_fwd(H, TK_PROC_NODE, a, s);
            if (rc != GM_OK) return rc;
        }
    }
    {
        TrainFwdArgs a{};
        a.rows = (int)n; a.x_in = t.h[M]; a.wstream = m->packed_t3 + m->t_node[M - 1] + (size_t)(NL + 2) * U;
        const float* v = mlp_vec(m, m->v_dec);
        a.bias = v; a.bias_tail = v + H; a.nl = NL;
        a.tape = t.dec; a.out = out; a.out_dim = m->d.out_dim;
        rc = launch_train_fwd(H, TK_DEC, a, s);
        if (rc != GM_OK) return rc;
    }
    {   // a flagged edge_index: the prediction is NaN, not a plausible number computed on a different graph
        const size_t cnt = (size_t)n * m->d.out_dim;
        hipLaunchKernelGGL(poison_if_flagged_kernel, dim3(small_grid(cnt)), dim3(256), 0, s, c.hdr, carve_csr(t.csr_src, n, e).hdr, out, cnt);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}

int gm_epd_backward(const gm_model* m, const float* const* T, int n_tensors, const float* nodes, const float* edge_attr, int64_t n,
                    int64_t e, const float* grad_out, float* const* grads, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                    void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_epd_backward");
    gm::DevGuard dev_guard(nodes);
    int rc = check_sizes(m, n, e, "gm_epd_backward");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(T && grads && nodes && grad_out && tape && ws && (e == 0 || edge_attr), GM_ERR_INVALID_ARGUMENT, "gm_epd_backward: null pointer");
    GM_REQUIRE(n_tensors == gm_model_num_tensors(&m->d), GM_ERR_INVALID_ARGUMENT, "gm_epd_backward: expected %d tensors, got %d",
               gm_model_num_tensors(&m->d), n_tensors);
    for (int i = 0; i < n_tensors; ++i)
        GM_REQUIRE(T[i] && grads[i], GM_ERR_INVALID_ARGUMENT, "gm_epd_backward: tensor / gradient %d is null", i);
    const int H = m->H, NL = m->NL, M = m->M, OD = m->d.out_dim;
    Tape t = carve_tape(tape, &m->d, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_epd_backward: tape %zu < %zu", tape_bytes, t.bytes);
    BwdWs b = carve_bwd(ws, &m->d, n, e);
    GM_REQUIRE(ws_bytes >= b.bytes, GM_ERR_WORKSPACE, "gm_epd_backward: workspace %zu < %zu", ws_bytes, b.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    WgradBatch wb;   // weight-gradient jobs run a batch per launch; flushed before anything overwrites their operands
    wgrad_batch_init(wb, b.part, H, s);
    CsrWs c = carve_csr(t.csr_dst, n, e);
    CsrWs c2 = carve_csr(t.csr_src, n, e);
    const int PM = tensors_per_normed_mlp(NL);
    const int b_enc_edge = 0, b_enc_node = PM, b_dec = (2 + 2 * M) * PM;
    auto b_edge = [&](int k) { return (2 + 2 * k) * PM; };
    auto b_node = [&](int k) { return (3 + 2 * k) * PM; };
    const size_t U = (size_t)m->T_HH * kStageFloatsB3;

    // ---- transposed operand images of every Linear on the backward path (batched: a few launches)
    {
        PackTJobs jobs;
        jobs.n = 0;
        auto flush = [&]() {
            if (rc == GM_OK && jobs.n > 0) rc = launch_pack_b3_batch(jobs, b.packT, s);
            jobs.n = 0;
        };
        auto packT = [&](const float* W, int w_rows, int ld, int col0, int ksub, size_t& off) {
            if (jobs.n == kPackTJobsMax) flush();
            PackTJob& j = jobs.job[jobs.n++];
            j.W = W; j.w_rows = w_rows; j.ld = ld; j.col0 = col0; j.ksub = ksub; j.fwd = 0; j.dst_off = off;
            off += (size_t)layer_stages_b3(w_rows, ksub) * kStageFloatsB3;
        };
        // the hidden Linears NL + 1 .. 2 of an MLP, in the order its backward chain consumes them
        auto packT_hidden = [&](int base, size_t& off) { for (int l = NL; l >= 1; --l) packT(T[base + 2 * l], H, H, 0, H, off); };
        size_t off = b.off_dec;
        packT(T[b_dec + 2 * NL], OD, H, 0, H, off);
        for (int l = NL - 1; l >= 0; --l) packT(T[b_dec + 2 * l], H, H, 0, H, off);
        for (int k = 0; k < M; ++k) {
            off = b.off_node[k];
            if (k + 1 < M) {
                packT(T[b_edge(k + 1)], H, 3 * H, m->ci * H, H, off);
                packT(T[b_edge(k + 1)], H, 3 * H, m->cj * H, H, off);
            }
            packT_hidden(b_node(k), off);
            packT(T[b_node(k)], H, 2 * H, m->ch * H, H, off);
            packT(T[b_node(k)], H, 2 * H, m->ca * H, H, off);
            off = b.off_edge[k];
            packT_hidden(b_edge(k), off);
            packT(T[b_edge(k)], H, 3 * H, m->ce * H, H, off);
        }
        off = b.off_enc_node;
        packT(T[b_edge(0)], H, 3 * H, m->ci * H, H, off);
        packT(T[b_edge(0)], H, 3 * H, m->cj * H, H, off);
        packT_hidden(b_enc_node, off);
        off = b.off_enc_edge;
        packT_hidden(b_enc_edge, off);
        flush();
        if (rc != GM_OK) return rc;
    }

    auto wgrad = [&](const float* dz, int ldz, int Mo, const float* X, int ldx, int K, const int* xidx, int64_t rows, float* out, int ldw,
                     int col0, float* db) {
        if (rc == GM_OK) rc = wgrad_enqueue(wb, dz, ldz, Mo, X, ldx, K, xidx, rows, out, ldw, col0, db);
    };
    // W3 (+ b3), W2 (+ b2) and the LayerNorm gradients of a normed MLP whose chain kernel has just run over `rows`
    // (b1 comes with the first-layer weight gradient at the call site)
    auto act = [&](const TapePtr& tp, int l, int64_t rows) { return tp.a + (size_t)(l - 1) * rows * H; };   // a_l
    // `node_set`: the chain left its dz in the node-sized set (BwdWs::dzn) -- the node MLPs and the node encoder, so that their
    // jobs and those of the edge chain before them (edge-sized set) wait in one batch: a flush only before each EDGE chain
    auto dz_of = [&](bool node_set, int l) { return node_set ? b.dznl(l) : b.dzl(l); };
    auto normed_tail_grads = [&](int base, const TapePtr& tp, int64_t rows, bool node_set) {
        for (int l = NL; l >= 1; --l)   // Linear l + 1: dW = dz_(l+1)^T a_l
            wgrad(dz_of(node_set, l + 1), H, H, act(tp, l, rows), H, H, nullptr, rows, grads[base + 2 * l], H, 0, grads[base + 2 * l + 1]);
    };
    auto chain = [&](TrainBwdArgs& a) { a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL; };
    // (the LayerNorm parameter gradients are summed inside the chain kernels: TrainBwdArgs.ln_part / dgamma / dbeta)
    auto ln_gamma = [&](size_t voff) { return m->vec + voff + (size_t)(NL + 1) * H; };

    // ---- the upstream gradient as the chains see it: zero for a forward whose edge_index was flagged (see gate_grad_out_kernel)
    {
        const size_t cnt = (size_t)n * OD;
        hipLaunchKernelGGL(gate_grad_out_kernel, dim3(small_grid(cnt)), dim3(256), 0, s, c.hdr, c2.hdr, grad_out, b.go, cnt);
        GM_LAUNCH_CHECK();
        grad_out = b.go;
    }
    // ---- decoder
    {
        TrainBwdArgs a{};
        a.rows = (int)n; a.dY = grad_out; a.out_dim = OD; a.tape = t.dec; a.wstream = b.packT + b.off_dec;
        chain(a); a.dx = b.dh;
        rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_DEC, a, s);
        if (rc != GM_OK) return rc;
        wgrad(grad_out, OD, OD, act(t.dec, NL, n), H, H, nullptr, n, grads[b_dec + 2 * NL], H, 0, grads[b_dec + 2 * NL + 1]);
        for (int l = NL - 1; l >= 1; --l)
            wgrad(b.dzl(l + 1), H, H, act(t.dec, l, n), H, H, nullptr, n, grads[b_dec + 2 * l], H, 0, grads[b_dec + 2 * l + 1]);
        wgrad(b.dzl(1), H, H, t.h[M], H, H, nullptr, n, grads[b_dec + 0], H, 0, grads[b_dec + 1]);
        if (rc != GM_OK) return rc;
    }
    // ---- processor blocks, last to first
    for (int k = M - 1; k >= 0; --k) {
        const bool has_next = k + 1 < M;
        {
            TrainBwdArgs a{};
            a.rows = (int)n; a.dY = b.dh; a.Gi = has_next ? b.Gi : nullptr; a.Gj = has_next ? b.Gj : nullptr;
            a.tape = t.tn[k]; a.ln_g = ln_gamma(m->v_node[k]); a.wstream = b.packT + b.off_node[k];
            a.ln_part = b.part; a.dgamma = grads[b_node(k) + 2 * (NL + 1)]; a.dbeta = grads[b_node(k) + 2 * (NL + 1) + 1]; a.dz = b.dzn; a.dz_stride = b.dzn_stride; a.nl = NL; a.dx_resid = b.dh; a.dx = b.dh; a.dagg_out = b.dagg;
            // no flush: the waiting jobs read the edge-sized dz set, Gi / Gj and tapes -- nothing this chain writes
            rc = launch_train_bwd(H, TB_NODE, a, s, &wb);
            if (rc != GM_OK) return rc;
            normed_tail_grads(b_node(k), t.tn[k], n, true);
            wgrad(b.dznl(1), H, H, t.h[k], H, H, nullptr, n, grads[b_node(k)], 2 * H, m->ch * H, grads[b_node(k) + 1]);
            wgrad(b.dznl(1), H, H, t.agg[k], H, H, nullptr, n, grads[b_node(k)], 2 * H, m->ca * H, nullptr);
            if (rc != GM_OK) return rc;
        }
        {
            TrainBwdArgs a{};
            a.rows = (int)e; a.dY = has_next ? b.de : nullptr; a.dagg = b.dagg; a.dst = c.dst;
            a.tape = t.te[k]; a.ln_g = ln_gamma(m->v_edge[k]); a.wstream = b.packT + b.off_edge[k];
            a.ln_part = b.part; a.dgamma = grads[b_edge(k) + 2 * (NL + 1)]; a.dbeta = grads[b_edge(k) + 2 * (NL + 1) + 1]; a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL; a.dx = b.de; a.residual = 1;
            rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_EDGE, a, s, &wb);
            if (rc != GM_OK) return rc;
            normed_tail_grads(b_edge(k), t.te[k], e, false);
            wgrad(b.dzl(1), H, H, t.e[k], H, H, nullptr, e, grads[b_edge(k)], 3 * H, m->ce * H, grads[b_edge(k) + 1]);
            if (rc != GM_OK) return rc;
            // node-level sums of dz1: everything the factorised layer 1 needs
            rc = launch_segment_sum_pair(H, c.in_ptr, nullptr, c2.in_ptr, c2.eid, b.dzl(1), nullptr, nullptr, b.Gi, b.Gj, n, s);
            wgrad(b.Gi, H, H, t.h[k], H, H, nullptr, n, grads[b_edge(k)], 3 * H, m->ci * H, nullptr);
            wgrad(b.Gj, H, H, t.h[k], H, H, nullptr, n, grads[b_edge(k)], 3 * H, m->cj * H, nullptr);
            if (rc != GM_OK) return rc;
        }
    }
    // ---- encoders
    {
        TrainBwdArgs a{};
        a.rows = (int)n; a.dY = b.dh; a.Gi = b.Gi; a.Gj = b.Gj; a.tape = t.en; a.ln_g = ln_gamma(m->v_enc_node);
        a.wstream = b.packT + b.off_enc_node;
        a.ln_part = b.part; a.dgamma = grads[b_enc_node + 2 * (NL + 1)]; a.dbeta = grads[b_enc_node + 2 * (NL + 1) + 1]; a.dz = b.dzn; a.dz_stride = b.dzn_stride; a.nl = NL;
        rc = launch_train_bwd(H, TB_ENC, a, s, &wb);   // no flush: as the node MLPs
        if (rc != GM_OK) return rc;
        normed_tail_grads(b_enc_node, t.en, n, true);
        wgrad(b.dznl(1), H, H, nodes, m->d.node_dim, m->d.node_dim, nullptr, n, grads[b_enc_node], m->d.node_dim, 0, grads[b_enc_node + 1]);
        if (rc != GM_OK) return rc;
    }
    if (e > 0) {
        TrainBwdArgs a{};
        a.rows = (int)e; a.dY = b.de; a.tape = t.ee; a.ln_g = ln_gamma(m->v_enc_edge); a.wstream = b.packT + b.off_enc_edge;
        a.ln_part = b.part; a.dgamma = grads[b_enc_edge + 2 * (NL + 1)]; a.dbeta = grads[b_enc_edge + 2 * (NL + 1) + 1]; a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL;
        rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_ENC, a, s, &wb);
        if (rc != GM_OK) return rc;
        normed_tail_grads(b_enc_edge, t.ee, e, false);
        wgrad(b.dzl(1), H, H, edge_attr, m->d.edge_dim, m->d.edge_dim, c.eid, e, grads[b_enc_edge], m->d.edge_dim, 0, grads[b_enc_edge + 1]);
        if (rc != GM_OK) return rc;
    }
    return wgrad_flush(wb);
}


// ------------------------------------------------------------------------------------------
// standalone blocks under autograd (the torch_graphnet surface, call sites epd_gnn.py:88,101)
// ------------------------------------------------------------------------------------------
size_t gm_block_tape_bytes(const gm_model_desc* desc, int interaction_network, int64_t n, int64_t e) {
    if (!desc || n < 0 || e < 0) return 0;
    return interaction_network ? carve_in_tape(nullptr, desc->hidden_size, desc->num_layers, n, e).bytes : carve_gi_tape(nullptr, desc->hidden_size, desc->num_layers, n, e).bytes;
}

size_t gm_block_backward_workspace_bytes(const gm_model_desc* desc, int64_t n, int64_t e) {
    if (!desc || n < 0 || e < 0) return 0;
    return carve_block_bwd(nullptr, desc, n, e).bytes;
}

int gm_graph_independent_forward_train(const gm_model* m, const float* x, int64_t n, const float* edge_attr, int64_t e, float* h_out,
                                       float* e_out, void* tape, size_t tape_bytes, void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_graph_independent_forward_train");
    gm::DevGuard dev_guard(x ? (const void*)x : (const void*)edge_attr);
    int rc = check_sizes(m, n, e, "gm_graph_independent_forward_train");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(x && h_out && tape && (e == 0 || (edge_attr && e_out)), GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_forward_train: null pointer");
    const int H = m->H, NL = m->NL;
    GiTape t = carve_gi_tape(tape, H, NL, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_graph_independent_forward_train: tape %zu < %zu", tape_bytes, t.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    auto normed = [&](TrainFwdArgs& a, size_t voff) {
        const float* v = mlp_vec(m, voff);
        a.bias = v; a.bias_tail = v + H; a.ln_g = v + (size_t)(NL + 1) * H; a.ln_b = v + (size_t)(NL + 2) * H; a.eps = m->d.ln_eps; a.nl = NL;
    };
    TrainFwdArgs a{};
    a.rows = (int)e; a.x_in = edge_attr; a.k1 = m->d.edge_dim; a.wstream = m->packed_t3 + m->t_enc_edge;
    normed(a, m->v_enc_edge);
    a.tape = t.ee; a.out = e_out;
    rc = launch_train_fwd(H, TK_ENC_EDGE, a, s);
    if (rc != GM_OK) return rc;
    TrainFwdArgs b{};
    b.rows = (int)n; b.x_in = x; b.k1 = m->d.node_dim; b.wstream = m->packed_t3 + m->t_enc_node;
    normed(b, m->v_enc_node);
    b.tape = t.en; b.out = h_out;
    return launch_train_fwd(H, TK_ENC_NODE, b, s);
}

int gm_graph_independent_backward(const gm_model* m, const float* const* T, int n_tensors, const float* x, const float* edge_attr, int64_t n,
                                  int64_t e, const float* dh, const float* de, float* dx, float* dedge_attr, float* const* grads,
                                  void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_graph_independent_backward");
    gm::DevGuard dev_guard(x ? (const void*)x : (const void*)edge_attr);
    int rc = check_sizes(m, n, e, "gm_graph_independent_backward");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(T && grads && x && dh && tape && ws && (e == 0 || (edge_attr && de)), GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_backward: null pointer");
    GM_REQUIRE(n_tensors == gm_model_num_tensors(&m->d), GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_backward: expected %d tensors, got %d",
               gm_model_num_tensors(&m->d), n_tensors);
    const int H = m->H, NL = m->NL;
    const int PM = tensors_per_normed_mlp(NL);
    for (int i = 0; i < 2 * PM; ++i) GM_REQUIRE(T[i] && grads[i], GM_ERR_INVALID_ARGUMENT, "gm_graph_independent_backward: tensor / gradient %d is null", i);
    GiTape t = carve_gi_tape(tape, H, NL, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_graph_independent_backward: tape %zu < %zu", tape_bytes, t.bytes);
    BwdWs b = carve_block_bwd(ws, &m->d, n, e);
    GM_REQUIRE(ws_bytes >= b.bytes, GM_ERR_WORKSPACE, "gm_graph_independent_backward: workspace %zu < %zu", ws_bytes, b.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    WgradBatch wb;   // weight-gradient jobs run a batch per launch; flushed before anything overwrites their operands
    wgrad_batch_init(wb, b.part, H, s);
    const size_t U = (size_t)m->T_HH * kStageFloatsB3;
    PackTJobs jobs;
    jobs.n = 0;
    auto packT = [&](const float* W, size_t off) {
        PackTJob& j = jobs.job[jobs.n++];
        j.W = W; j.w_rows = H; j.ld = H; j.col0 = 0; j.ksub = H; j.fwd = 0; j.dst_off = off;
    };
    // node stream sits behind the (unused) projection slots of the model layout
    for (int l = NL; l >= 1; --l) {
        packT(T[PM + 2 * l], b.off_enc_node + (size_t)(2 + NL - l) * U);
        packT(T[2 * l], b.off_enc_edge + (size_t)(NL - l) * U);
    }
    auto packT_in = [&](const float* W, int k1, size_t off) {  // (W1 [H, k1])^T as a Linear with k1 outputs, H inputs
        PackTJob& j = jobs.job[jobs.n++];
        j.W = W; j.w_rows = H; j.ld = k1; j.col0 = 0; j.ksub = k1; j.fwd = 0; j.dst_off = off;
    };
    if (dx) packT_in(T[PM], m->d.node_dim, b.off_enc_node + (size_t)(NL + 2) * U);
    if (dedge_attr) packT_in(T[0], m->d.edge_dim, b.off_enc_edge + (size_t)NL * U);
    rc = launch_pack_b3_batch(jobs, b.packT, s);
    if (rc != GM_OK) return rc;
    auto run = [&](int base, const TapePtr& tp, int64_t rows, const float* dY, size_t voff, size_t woff, const float* X, int k1, float* dxin) {
        if (rows <= 0 || rc != GM_OK) return;
        TrainBwdArgs a{};
        a.rows = (int)rows; a.dY = dY; a.tape = tp; a.ln_g = m->vec + voff + (size_t)(NL + 1) * H; a.wstream = b.packT + woff;
        a.dx_in = dxin; a.k1 = k1;
        a.ln_part = b.part; a.dgamma = grads[base + 2 * (NL + 1)]; a.dbeta = grads[base + 2 * (NL + 1) + 1]; a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL;
        rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_ENC, a, s, &wb);
        for (int l = NL; l >= 1 && rc == GM_OK; --l)
            rc = wgrad_enqueue(wb, b.dzl(l + 1), H, H, tp.a + (size_t)(l - 1) * rows * H, H, H, nullptr, rows, grads[base + 2 * l], H, 0,
                              grads[base + 2 * l + 1]);
        if (rc == GM_OK) rc = wgrad_enqueue(wb, b.dzl(1), H, H, X, k1, k1, nullptr, rows, grads[base], k1, 0, grads[base + 1]);
    };
    run(PM, t.en, n, dh, m->v_enc_node, b.off_enc_node + 2 * U, x, m->d.node_dim, dx);
    run(0, t.ee, e, de, m->v_enc_edge, b.off_enc_edge, edge_attr, m->d.edge_dim, dedge_attr);
    if (rc == GM_OK) rc = wgrad_flush(wb);
    return rc;
}

int gm_interaction_network_forward_train(const gm_model* m, int k, const float* h, int64_t n, const float* e_in, const int64_t* edge_index,
                                         int64_t e, float* h_out, float* e_out, void* tape, size_t tape_bytes, void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_interaction_network_forward_train");
    gm::DevGuard dev_guard(h);
    int rc = check_sizes(m, n, e, "gm_interaction_network_forward_train");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(k >= 0 && k < m->M, GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_forward_train: block %d out of range", k);
    GM_REQUIRE(h && h_out && tape && (e == 0 || (e_in && e_out && edge_index)), GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_forward_train: null pointer");
    const int H = m->H, NL = m->NL;
    InTape t = carve_in_tape(tape, H, NL, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_interaction_network_forward_train: tape %zu < %zu", tape_bytes, t.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = gm::csr_from_edge_index(edge_index, n, e, m->d.flow, t.csr_dst, t.csr_bytes, false, (hipStream_t)stream);
    if (rc != GM_OK) return rc;
    CsrWs c = carve_csr(t.csr_dst, n, e);
    rc = launch_swap_index(c.src, e, t.ei2, s);
    if (rc != GM_OK) return rc;
    rc = gm::csr_from_edge_index(t.ei2, n, e, 0, t.csr_src, t.csr_bytes, false, (hipStream_t)stream);
    if (rc != GM_OK) return rc;
    const size_t U = (size_t)m->T_HH * kStageFloatsB3;
    {
        TrainFwdArgs pa{};   // P = [h W_i^T + b1 | h W_j^T] of this block's edge MLP
        pa.rows = (int)n; pa.x_in = h; pa.bias = m->vec + m->v_edge[k]; pa.out = t.P; pa.nl = NL;
        pa.wstream = k == 0 ? m->packed_t3 + m->t_enc_node + (size_t)(m->T_n0 + NL * m->T_HH) * kStageFloatsB3
                            : m->packed_t3 + m->t_node[k - 1] + (size_t)(NL + 2) * U;
        rc = launch_train_fwd(H, TK_PROJ, pa, s);
    }
    if (rc != GM_OK) return rc;
    auto normed = [&](TrainFwdArgs& a, size_t voff) {
        const float* v = mlp_vec(m, voff);
        a.bias = v; a.bias_tail = v + H; a.ln_g = v + (size_t)(NL + 1) * H; a.ln_b = v + (size_t)(NL + 2) * H; a.eps = m->d.ln_eps; a.nl = NL;
    };
    {
        TrainFwdArgs a{};
        a.rows = (int)e; a.x_in = e_in; a.rowidx = c.eid; a.dst = c.dst; a.src = c.src; a.P = t.P; a.wstream = m->packed_t3 + m->t_edge[k];
        normed(a, m->v_edge[k]);
        a.tape = t.te; a.out = e_out; a.residual = 0;
        rc = launch_train_fwd(H, TK_PROC_EDGE, a, s);
        if (rc != GM_OK) return rc;
    }
    const float* ve = mlp_vec(m, m->v_edge[k]);
    rc = launch_segment_sum(H, c.in_ptr, nullptr, t.te.xhat, ve + (size_t)(NL + 1) * H, ve + (size_t)(NL + 2) * H, t.agg, n, s);
    if (rc != GM_OK) return rc;
    TrainFwdArgs a{};
    a.rows = (int)n; a.x_in = h; a.agg = t.agg; a.wstream = m->packed_t3 + m->t_node[k];
    normed(a, m->v_node[k]);
    a.tape = t.tn; a.out = h_out; a.residual = 0;
    return launch_train_fwd(H, TK_PROC_NODE, a, s);
}

int gm_interaction_network_backward(const gm_model* m, int k, const float* const* T, int n_tensors, const float* h, const float* e_in,
                                    int64_t n, int64_t e, const float* dh_out, const float* de_out, float* dh_in, float* de_in,
                                    float* const* grads, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
    GM_REQUIRE(!m || m->legacy, GM_ERR_UNSUPPORTED, "%s: the training kernels are instantiated for hidden_size 64 / 128 / 256", "gm_interaction_network_backward");
    gm::DevGuard dev_guard(h);
    int rc = check_sizes(m, n, e, "gm_interaction_network_backward");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(k >= 0 && k < m->M, GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_backward: block %d out of range", k);
    GM_REQUIRE(T && grads && h && dh_out && dh_in && tape && ws && (e == 0 || (e_in && de_out && de_in)), GM_ERR_INVALID_ARGUMENT,
               "gm_interaction_network_backward: null pointer");
    GM_REQUIRE(n_tensors == gm_model_num_tensors(&m->d), GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_backward: expected %d tensors, got %d",
               gm_model_num_tensors(&m->d), n_tensors);
    const int H = m->H, NL = m->NL;
    const int PM = tensors_per_normed_mlp(NL);
    const int be = (2 + 2 * k) * PM, bn = (3 + 2 * k) * PM;
    for (int i = be; i < bn + PM; ++i) GM_REQUIRE(T[i] && grads[i], GM_ERR_INVALID_ARGUMENT, "gm_interaction_network_backward: tensor / gradient %d is null", i);
    InTape t = carve_in_tape(tape, H, NL, n, e);
    GM_REQUIRE(tape_bytes >= t.bytes, GM_ERR_WORKSPACE, "gm_interaction_network_backward: tape %zu < %zu", tape_bytes, t.bytes);
    BwdWs b = carve_block_bwd(ws, &m->d, n, e);
    GM_REQUIRE(ws_bytes >= b.bytes, GM_ERR_WORKSPACE, "gm_interaction_network_backward: workspace %zu < %zu", ws_bytes, b.bytes);
    rc = train_kernels_init();
    if (rc != GM_OK) return rc;
    rc = weights_ready_on(m, (hipStream_t)stream);   // the weight streams may have been packed on another stream (model.h)
    if (rc != GM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    WgradBatch wb;   // weight-gradient jobs run a batch per launch; flushed before anything overwrites their operands
    wgrad_batch_init(wb, b.part, H, s);
    CsrWs c = carve_csr(t.csr_dst, n, e);
    CsrWs c2 = carve_csr(t.csr_src, n, e);
    const size_t U = (size_t)m->T_HH * kStageFloatsB3;
    {
        PackTJobs jobs;
        jobs.n = 0;
        auto packT = [&](const float* W, int ld, int col0, size_t off) {
            PackTJob& j = jobs.job[jobs.n++];
            j.W = W; j.w_rows = H; j.ld = ld; j.col0 = col0; j.ksub = H; j.fwd = 0; j.dst_off = off;
        };
        for (int l = NL; l >= 1; --l) {
            packT(T[bn + 2 * l], H, 0, b.off_node[0] + (size_t)(NL - l) * U);
            packT(T[be + 2 * l], H, 0, b.off_edge[0] + (size_t)(NL - l) * U);
        }
        packT(T[bn], 2 * H, m->ch * H, b.off_node[0] + (size_t)NL * U);
        packT(T[bn], 2 * H, m->ca * H, b.off_node[0] + (size_t)(NL + 1) * U);
        packT(T[be], 3 * H, m->ce * H, b.off_edge[0] + (size_t)NL * U);
        packT(T[be], 3 * H, m->ci * H, b.off_enc_node);       // W_i^T, W_j^T: projection backward
        packT(T[be], 3 * H, m->cj * H, b.off_enc_node + U);
        rc = launch_pack_b3_batch(jobs, b.packT, s);
        if (rc != GM_OK) return rc;
    }
    auto wgrad = [&](const float* dz, const float* X, int64_t rows, float* out, int ldw, int col0, float* db) {
        if (rc == GM_OK) rc = wgrad_enqueue(wb, dz, H, H, X, H, H, nullptr, rows, out, ldw, col0, db);
    };
    // node MLP: dY = dh_out (no residual inside the block); dx = W_h^T dz1 -> b.dh, dagg -> b.dagg
    {
        TrainBwdArgs a{};
        a.rows = (int)n; a.dY = dh_out; a.tape = t.tn; a.ln_g = m->vec + m->v_node[k] + (size_t)(NL + 1) * H; a.wstream = b.packT + b.off_node[0];
        a.ln_part = b.part; a.dgamma = grads[bn + 2 * (NL + 1)]; a.dbeta = grads[bn + 2 * (NL + 1) + 1]; a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL; a.dx = b.dh; a.dagg_out = b.dagg;
        rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_NODE, a, s, &wb);
        if (rc != GM_OK) return rc;
        for (int l = NL; l >= 1; --l) wgrad(b.dzl(l + 1), t.tn.a + (size_t)(l - 1) * n * H, n, grads[bn + 2 * l], H, 0, grads[bn + 2 * l + 1]);
        wgrad(b.dzl(1), h, n, grads[bn], 2 * H, m->ch * H, grads[bn + 1]);
        wgrad(b.dzl(1), t.agg, n, grads[bn], 2 * H, m->ca * H, nullptr);
        if (rc != GM_OK) return rc;
    }
    if (e > 0) {
        TrainBwdArgs a{};
        a.rows = (int)e; a.dY = de_out; a.dyidx = c.eid; a.dagg = b.dagg; a.dst = c.dst; a.tape = t.te;
        a.ln_g = m->vec + m->v_edge[k] + (size_t)(NL + 1) * H; a.wstream = b.packT + b.off_edge[0];
        a.ln_part = b.part; a.dgamma = grads[be + 2 * (NL + 1)]; a.dbeta = grads[be + 2 * (NL + 1) + 1]; a.dz = b.dz; a.dz_stride = b.dz_stride; a.nl = NL; a.dx = de_in; a.dxidx = c.eid; a.residual = 0;
        rc = wgrad_flush(wb);
        if (rc == GM_OK) rc = launch_train_bwd(H, TB_EDGE, a, s, &wb);
        if (rc != GM_OK) return rc;
        for (int l = NL; l >= 1; --l) wgrad(b.dzl(l + 1), t.te.a + (size_t)(l - 1) * e * H, e, grads[be + 2 * l], H, 0, grads[be + 2 * l + 1]);
        if (rc == GM_OK) rc = wgrad_enqueue(wb, b.dzl(1), H, H, e_in, H, H, c.eid, e, grads[be], 3 * H, m->ce * H, grads[be + 1]);
        if (rc != GM_OK) return rc;
    }
    rc = launch_segment_sum_pair(H, c.in_ptr, nullptr, c2.in_ptr, c2.eid, b.dzl(1), nullptr, nullptr, b.Gi, b.Gj, n, s);
    wgrad(b.Gi, h, n, grads[be], 3 * H, m->ci * H, nullptr);
    wgrad(b.Gj, h, n, grads[be], 3 * H, m->cj * H, nullptr);
    if (rc != GM_OK) return rc;
    // dh_in = W_h^T dz1 (node MLP) + W_i^T G_i + W_j^T G_j (edge MLP, factorised layer 1)
    TrainBwdArgs a{};
    a.rows = (int)n; a.dY = b.dh; a.Gi = b.Gi; a.Gj = b.Gj; a.wstream = b.packT + b.off_enc_node; a.dx = dh_in;
    rc = wgrad_flush(wb);
    if (rc != GM_OK) return rc;
    return launch_train_bwd(H, TB_PROJ, a, s);
}

}  // extern "C"
