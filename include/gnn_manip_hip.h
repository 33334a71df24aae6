/*
 * gnn_manip_hip.h -- C ABI of libgnnmanip_hip.so: the MI355X (gfx950) rollout engine for
 * gnn-manip's encode-process-decode particle simulator.
 *
 * The reference (dblanm/gnn-manip) is pure Python and has no FFI of its own; its boundary is
 * the duck-typed call surface listed in SURVEY.md section 8b.  Every entry point below names
 * the reference function it replaces (paths relative to the reference checkout).  The Python
 * host layer in gnn_manip_amd/ binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory unless the name ends
 *     in _host; the caller (PyTorch) owns every buffer and every workspace;
 *   - nothing is allocated per call; gm_model_create() allocates the packed weight image once;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising, unless documented otherwise;
 *   - return value: GM_OK (0) or a negative gm_status; gm_last_error() returns a thread-local
 *     message for the last failure.  Nothing throws or aborts across this boundary.
 *   - threading / streams: the stateless entry points are re-entrant.  A gm_model handle is NOT: it packs its inference
 *     operand images lazily, on the stream of the first inference call after gm_model_create / gm_model_update, so one
 *     handle is used from one thread and its update and inference calls go to ONE stream (or the caller orders the
 *     streams itself); use one handle per thread / stream otherwise (the reference's callers are single-threaded on one
 *     stream: rollout_utils.py:97-102, train_dyn.py:45-72).
 */
#ifndef GNN_MANIP_HIP_H
#define GNN_MANIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gm_status {
    GM_OK = 0,
    GM_ERR_INVALID_ARGUMENT = -1,
    GM_ERR_UNSUPPORTED = -2,   /* e.g. hidden size the kernels are not instantiated for */
    GM_ERR_HIP = -3,           /* a HIP runtime call failed */
    GM_ERR_WORKSPACE = -4,     /* workspace too small */
    GM_ERR_DATA = -5           /* device-side data error (non-finite position, bad index) */
} gm_status;

typedef struct gm_model gm_model; /* opaque */

const char* gm_last_error(void);
int gm_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Radius graph.   Replaces get_connectivity(pos_nodes, conn_r, max_neighbours)
 *                 gnn_manip/utils/utils.py:64-93  (sklearn KDTree.query_radius semantics:
 *                 float64 squared distance, d2 <= r*r, ascending distance, first max_nb kept,
 *                 ties broken on the smaller index).
 * ------------------------------------------------------------------------------------------ */
size_t gm_graph_workspace_bytes(int64_t n_nodes, int max_neighbours);

/* Build per-query neighbour lists into the workspace.  pos points at x of node 0; node i is at
 * pos + i*pos_stride (floats), so the xyz columns of a [N, D] state row can be passed directly. */
int gm_radius_graph_build(const float* pos, int64_t pos_stride, int64_t n_nodes, double conn_r,
                          int max_neighbours, void* graph_ws, size_t graph_ws_bytes, void* stream);

/* Same for a batch of equal-sized graphs stored back to back (node i belongs to graph i / nodes_per_graph):
 * no edge crosses graphs -- the batch of collate_utils.py:68-87 (edge indices offset by N*i, :76). */
int gm_radius_graph_build_batched(const float* pos, int64_t pos_stride, int64_t n_nodes, int64_t nodes_per_graph,
                                  double conn_r, int max_neighbours, void* graph_ws, size_t graph_ws_bytes, void* stream);

/* Synchronises `stream`; returns E and the device error flags of the last build. */
int gm_radius_graph_num_edges(const void* graph_ws, int64_t* n_edges_host, void* stream);

/* Emit the reference-ordered edge list: grouped by sender (= query node) ascending, distance
 * ascending inside a group.  senders/receivers: int64[capacity], capacity >= E. */
int gm_radius_graph_edges(const void* graph_ws, int64_t n_nodes, int max_neighbours,
                          int64_t* senders, int64_t* receivers, int64_t capacity, void* stream);

/* ------------------------------------------------------------------------------------------
 * Destination-sorted edge structure (the engine's internal edge order).
 * Aggregation index i = edge_index[1], source j = edge_index[0] (PyG source_to_target, see
 * DESIGN.md).  Edges are grouped by i; inside a group ascending original edge id (stable).
 * ------------------------------------------------------------------------------------------ */
size_t gm_csr_workspace_bytes(int64_t n_nodes, int64_t edge_capacity);
int gm_csr_from_graph(const void* graph_ws, int64_t n_nodes, int max_neighbours,
                      void* csr_ws, size_t csr_ws_bytes, void* stream);
int gm_csr_from_edge_index(const int64_t* edge_index /* [2,E] row-major */, int64_t n_nodes,
                           int64_t n_edges, void* csr_ws, size_t csr_ws_bytes, void* stream);
/* The same with the aggregation index chosen by `flow` (gm_model_desc.flow): 0 = edge_index[1] (default), 1 = edge_index[0]. */
int gm_csr_from_graph_flow(const void* graph_ws, int64_t n_nodes, int max_neighbours, int flow,
                           void* csr_ws, size_t csr_ws_bytes, void* stream);
int gm_csr_from_edge_index_flow(const int64_t* edge_index, int64_t n_nodes, int64_t n_edges, int flow,
                                void* csr_ws, size_t csr_ws_bytes, void* stream);
/* Synchronises; copies E and error flags to the host. */
int gm_csr_num_edges(const void* csr_ws, int64_t* n_edges_host, void* stream);
/* The same verdict from a HOST copy of the workspace's first 16 bytes (its header: n_edges, error flags, flow, pad) -- for callers
 * that copy the header asynchronously behind their work and poll for it instead of synchronising.  No device access.  The tape of
 * gm_epd_forward_train and of gm_interaction_network_forward_train begins with the csr workspace of that forward: the same
 * 16 bytes tell whether its edge_index held an entry outside [0, n_nodes) (such edges are left out). */
int gm_csr_header_status(const int32_t* header_host /*[4]*/, int64_t* n_edges_host /* may be NULL */);

/* ------------------------------------------------------------------------------------------
 * Features.
 * gm_edge_features      replaces get_edges_displacement   gnn_manip/utils/utils.py:43-61
 *                       out[e] = [(p[s]-p[r])/conn_r, ||.||], reference edge order.
 * gm_edge_features_csr  same values in the destination-sorted order of a csr workspace (sender = edge_index[0] whatever
 *                       `flow` the workspace was built with: the header records it).
 * gm_node_features      replaces GraphBoundedMultimaterial(Control).compute_nodes
 *                       gnn_manip/utils/collate_utils.py:195-208,217-232 (+ get_nodes_vel,
 *                       utils.py:27-40).  obs: [k, N, D] float32 row-major.
 * ------------------------------------------------------------------------------------------ */
int gm_edge_features(const float* pos, int64_t pos_stride, const int64_t* senders,
                     const int64_t* receivers, int64_t n_edges, float conn_r, float* out /*[E,4]*/,
                     void* stream);
int gm_edge_features_csr(const float* pos, int64_t pos_stride, const void* csr_ws, int64_t n_nodes,
                         int64_t edge_capacity, float conn_r, float* out /*[cap,4]*/, void* stream);

typedef struct gm_feature_desc {
    double conn_r;          /* connectivity radius as the Python float the reference passes: the radius
                               test runs in float64 on it, float32 feature divisions on (float)conn_r */
    int32_t k_steps;        /* frames in the window (6) */
    int32_t data_dim;       /* D: columns per particle row (8 with control, 5 without) */
    int32_t cart_col;       /* first of the 3 contiguous position columns (2) */
    int32_t material_col;   /* (1) */
    int32_t control_col;    /* first of the 3 contiguous control columns (5), or -1: no control */
    int32_t nodes_per_graph; /* rollout of a batch of equal-sized scenes stored back to back (candidates); 0: one scene */
    float vel_mean[3], vel_std[3];
    float acc_mean[3], acc_std[3];
    float lower_bounds[3], upper_bounds[3];
} gm_feature_desc;

int gm_node_features(const float* obs, int64_t n_nodes, const gm_feature_desc* desc,
                     float* out /* [N, 3*(k-1)+6+1(+3)] */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Integrator + rollout state update.
 * gm_integrate     replaces get_position_from_prediction  gnn_manip/utils/rollout_utils.py:145-158
 * gm_state_pre     replaces rollout_utils.py:40-47 == traj_utils.py:126-134: control columns of the
 *                  rigid rows (material == 1) of the last frame <- rigid_target - current xyz.
 * gm_state_post    replaces rollout_utils.py:53-61 == traj_utils.py:146-152: window shift, write
 *                  p_{t+1}, overwrite rigid rows' xyz with the scripted pose.
 * rigid_target: [N_rigid,3] poses in rigid-row order; rigid_rank: int32[N] = rank of row among
 * rigid rows (or -1), built once per scene by gm_rigid_rank.
 * gm_rigid_transform replaces compute_particles_tmatrix    gnn_manip/utils/traj_utils.py:167-194
 * ------------------------------------------------------------------------------------------ */
int gm_integrate(const float* pred_acc /*[N,3]*/, const float* obs, int64_t n_nodes,
                 const gm_feature_desc* desc, float* next_pos /*[N,3]*/, void* stream);
int gm_rigid_rank(const float* obs, int64_t n_nodes, const gm_feature_desc* desc, int32_t* rigid_rank,
                  int32_t* n_rigid_dev, void* stream);
int gm_state_pre(float* obs, int64_t n_nodes, const gm_feature_desc* desc, const int32_t* rigid_rank,
                 const float* rigid_target /* or NULL: control <- current xyz (traj_utils.py:131) */,
                 void* stream);
int gm_state_post(float* obs, int64_t n_nodes, const gm_feature_desc* desc, const float* next_pos,
                  const int32_t* rigid_rank, const float* rigid_target /* or NULL */, void* stream);
int gm_rigid_transform(const float* rigid_init /*[Nr,3]*/, int64_t n_rigid, const float* rot_cs_ty
                       /* [T,3] host-computed (cos, sin, ty_init[1]+translation) float32 */,
                       int64_t n_steps, const float ty_init[3], float* out /*[T,Nr,3]*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * Model.  Replaces EncProcDecGNN.__init__/_build_mlp + load_state_dict
 *         gnn_manip/models/epd_gnn.py:13-49,72-84 ; rollout_utils.py:137-139.
 * tensors_host_or_dev: the model's parameters in state_dict order (encoder.phi_edge.*,
 * encoder.phi_node.*, processor.k.phi_edge.*, processor.k.phi_node.*, decoder.*; inside an MLP:
 * Linear weight [out,in] row-major, bias, ..., LayerNorm weight, bias).  They are copied and
 * repacked into the MFMA operand image; the caller's tensors are not referenced afterwards.
 *
 * Numeric domain.  The reference computes every Linear in float32.  The inference kernels form the same float32 products
 * on the fp16 matrix pipe: each operand is a pair of fp16 numbers (22 significant bits), three exact partial products per
 * multiply, float32 accumulation.  An fp16 pair represents |x| < 65504 and keeps full precision down to 2^-3, so the
 * kernels hold every operand at a power-of-two scale chosen from the weights (nothing is rounded by it): hidden
 * activations ride at an rms near 2^4 (4096-fold headroom) whatever the scale of the weights -- scaling (W_l, b_l) by s and W_(l+1) by 1/s
 * changes no operand bit -- and each raw node / edge feature row is scaled by its own maximum, so features of any
 * magnitude (1e-30 .. 1e30) are as accurate as in float32.  Latents (h, e, agg: LayerNorm outputs and their sums) enter at
 * their natural magnitude: fine from ~2^-8 to 65504.  A value outside the representable range is never clamped: the
 * kernel that meets it sets a flag in the CSR header of that forward and gm_csr_num_edges / gm_rollout_status return
 * GM_ERR_DATA ("fp16 split range exceeded"); the outputs of that forward are then invalid.  A float32 evaluation would
 * also stay finite for magnitudes up to 3.4e38: that part of its domain is not covered.
 * ------------------------------------------------------------------------------------------ */
typedef struct gm_model_desc {
    int32_t node_dim, edge_dim, out_dim;
    int32_t hidden_size;   /* 1 .. 256 (epd_gnn.py:13-14 takes any int); see gm_padded_hidden_size */
    int32_t num_layers;    /* >= 2: hidden layers per MLP (epd_gnn.py:26) */
    int32_t m_steps;       /* >= 1 */
    float ln_eps;          /* 1e-5 */
    /* Convention of the torch_graphnet.InteractionNetwork block, whose source is absent from the reference tree
     * (.gitmodules:1-3).  All zero = the default of DESIGN.md section 2 (BASELINE.json north_star wording, PyG
     * source_to_target).  Exposed so that a checkpoint trained with another convention can be matched. */
    int32_t flow;          /* 0: j = edge_index[0], aggregation index i = edge_index[1]; 1: i = edge_index[0], j = edge_index[1] */
    int32_t col_i, col_j, col_e; /* column block (0, 1, 2) of phi_e's first Linear that multiplies h_i, h_j, e; all zero = (0, 1, 2) */
    int32_t node_agg_first; /* 0: phi_v(cat[h, agg]); 1: phi_v(cat[agg, h]) */
} gm_model_desc;

/* The kernels are instantiated for the widths 64, 128 and 256; a model of another hidden size h <= 256 runs
 * zero-padded at the next of them: weights, biases and LayerNorm vectors are padded with zeros when they are packed, LayerNorm
 * statistics are taken over the h features that exist (the Linear in front of a LayerNorm is packed centred over its outputs,
 * so the statistics are a mean square and zero padding adds nothing), the padded features stay exactly zero.  Returns that width (0:
 * unsupported).  It is the ROW STRIDE of every latent array that crosses this interface: h / e of
 * gm_graph_independent_forward and gm_interaction_network_forward (inputs zero-padded by the caller, outputs padded with
 * zeros); gm_epd_forward and the rollout entry points have no latent arguments.  Training entry points: hidden 64 / 128 / 256. */
int gm_padded_hidden_size(int hidden_size);

int gm_model_num_tensors(const gm_model_desc* desc);
int gm_model_create(const gm_model_desc* desc, const float* const* tensors, int n_tensors,
                    int tensors_on_device, void* stream, gm_model** out);
/* create / update copy the tensors into the model (stream-ordered for device tensors; host tensors are read before the call
 * returns): the caller's buffers are not referenced afterwards.  The operand images of the training kernels are packed by the
 * call itself; those of the inference kernels by the first inference call that follows (on that call's stream) -- a training
 * loop, which updates the weights every step, never pays for them. */
int gm_model_update(gm_model* m, const float* const* tensors, int n_tensors, int tensors_on_device,
                    void* stream);
void gm_model_destroy(gm_model* m);

/* Processor edge kernel of THIS model (no reference counterpart; diagnostics / A-B measurements).  0 = automatic
 * (the systolic fp16 x 3 kernel for hidden 128 / num_layers 2 in the fused forward, else the
 * streamed one); 5 = systolic fp16 x 3 where it applies, else 6; 6 = streamed fp16 x 3 (hmlp.hip; every MLP of the
 * model, any supported size); 7 = as 5, and the processor NODE MLPs in the systolic form too (node kernel + projection kernel,
 * hedge.hip) whatever the graph's size -- 0 takes those for graphs of 49152 nodes or more (one graph of a batch counts, not the batch), where a workgroup has blocks enough to pipeline.  1..4 were round 1's fp32 / bf16 x 6 kernels: removed from the library in round 5
 * (GM_ERR_UNSUPPORTED).  No environment variable is read: the choice belongs to the handle. */
int gm_model_set_edge_kernel(gm_model* m, int choice);

size_t gm_forward_workspace_bytes(const gm_model_desc* desc, int64_t n_nodes, int64_t edge_capacity);
/* workspace of gm_interaction_network_forward (the latent edge arrays are the caller's) */
size_t gm_block_workspace_bytes(const gm_model_desc* desc, int64_t n_nodes, int64_t edge_capacity);

/* EncProcDecGNN.forward(nodes, edge_attr, edge_index)   epd_gnn.py:86-105.
 * edge_attr is in the CALLER's edge order; the csr workspace (built from the same edge_index or
 * from the radius graph) carries the permutation.  edge_attr_is_csr_order != 0 says edge_attr is
 * already destination-sorted (rollout path). */
int gm_epd_forward(const gm_model* m, const float* nodes /*[N,node_dim]*/, int64_t n_nodes,
                   const float* edge_attr /*[E,edge_dim]*/, int edge_attr_is_csr_order,
                   const void* csr_ws, int64_t edge_capacity, float* out /*[N,out_dim]*/,
                   void* fwd_ws, size_t fwd_ws_bytes, void* stream);

/* torch_graphnet.GraphIndependent call site epd_gnn.py:88 -- (phi_node(x), phi_edge(e)).
 * torch_graphnet.InteractionNetwork call site epd_gnn.py:101 -- (h', e'), no residual.
 * `block` = -1: encoder; 0..m_steps-1: processor block.  Outputs in the caller's edge order.
 * Range violations of the fp16 split (see "Numeric domain" above): gm_interaction_network_forward flags them in the header of
 * its csr workspace like the fused forward (gm_csr_num_edges returns GM_ERR_DATA).  gm_graph_independent_forward has no such
 * header: there a violating row -- a non-finite input feature, or a range violation further on -- comes out as NaN in every
 * feature (the kernels carry a per-row poison term from every Linear's accumulators into the LayerNorm), which is how the caller
 * sees it; the other rows are unaffected. */
int gm_graph_independent_forward(const gm_model* m, const float* x, int64_t n_nodes, const float* edge_attr,
                                 int64_t n_edges, float* h_out, float* e_out, void* stream);
int gm_interaction_network_forward(const gm_model* m, int block, const float* h, int64_t n_nodes,
                                   const float* e, const void* csr_ws, int64_t edge_capacity,
                                   float* h_out, float* e_out, void* fwd_ws, size_t fwd_ws_bytes,
                                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Training: what examples/train_dyn.py:45-72 does with the model --
 * `model.forward(x, edge_attr, edge_index)` under autograd, `loss.backward()`, optimiser step.
 * gm_epd_forward_train == EncProcDecGNN.forward (epd_gnn.py:86-105) that also records the activation
 * tape in the caller's buffer (gm_train_tape_bytes); edge_index is the caller's int64 [2, E]
 * (j = row 0, aggregation index i = row 1).  gm_epd_backward consumes the tape and ACCUMULATES the
 * gradient of every parameter into grads[t] (same order and shapes as the `tensors` of
 * gm_model_create; the caller zeroes them), given grad_out = dLoss/d(out) [N, out_dim].  `tensors`
 * are the parameter values the forward ran with (device pointers).  Inputs x / edge_attr get no
 * gradient (they are data in train_dyn.py).  Weight gradients are reduced in a fixed order
 * (deterministic, like the bias and LayerNorm-parameter gradients: no atomics on the backward path). */
size_t gm_train_tape_bytes(const gm_model_desc* desc, int64_t n_nodes, int64_t n_edges);
size_t gm_train_backward_workspace_bytes(const gm_model_desc* desc, int64_t n_nodes, int64_t n_edges);
int gm_epd_forward_train(const gm_model* m, const float* nodes, int64_t n_nodes, const float* edge_attr,
                         const int64_t* edge_index, int64_t n_edges, float* out, void* tape,
                         size_t tape_bytes, void* stream);
int gm_epd_backward(const gm_model* m, const float* const* tensors, int n_tensors, const float* nodes,
                    const float* edge_attr, int64_t n_nodes, int64_t n_edges, const float* grad_out,
                    float* const* grads, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                    void* stream);

/* The two standalone blocks under autograd -- the torch_graphnet surface the reference's own
 * EncProcDecGNN wiring calls (epd_gnn.py:30-33,42-45,88,101).  *_forward_train record a tape
 * (gm_block_tape_bytes; interaction_network = 0 for GraphIndependent, 1 for InteractionNetwork);
 * *_backward accumulate parameter gradients into grads[] (full-model tensor order, only the block's own
 * entries are touched) and, for the InteractionNetwork, return the input gradients dh_in [N,H] and
 * de_in [E,H] (caller's edge order); for the GraphIndependent the input gradients dx / dedge_attr are optional
 * (NULL at the reference's call site, where the inputs are data). */
size_t gm_block_tape_bytes(const gm_model_desc* desc, int interaction_network, int64_t n_nodes, int64_t n_edges);
size_t gm_block_backward_workspace_bytes(const gm_model_desc* desc, int64_t n_nodes, int64_t n_edges);
int gm_graph_independent_forward_train(const gm_model* m, const float* x, int64_t n_nodes, const float* edge_attr,
                                       int64_t n_edges, float* h_out, float* e_out, void* tape, size_t tape_bytes,
                                       void* stream);
int gm_graph_independent_backward(const gm_model* m, const float* const* tensors, int n_tensors, const float* x,
                                  const float* edge_attr, int64_t n_nodes, int64_t n_edges, const float* dh,
                                  const float* de, float* dx /*[N,node_dim] or NULL*/,
                                  float* dedge_attr /*[E,edge_dim] or NULL*/, float* const* grads, void* tape,
                                  size_t tape_bytes, void* ws, size_t ws_bytes, void* stream);
int gm_interaction_network_forward_train(const gm_model* m, int block, const float* h, int64_t n_nodes,
                                         const float* e, const int64_t* edge_index, int64_t n_edges, float* h_out,
                                         float* e_out, void* tape, size_t tape_bytes, void* stream);
int gm_interaction_network_backward(const gm_model* m, int block, const float* const* tensors, int n_tensors,
                                    const float* h, const float* e, int64_t n_nodes, int64_t n_edges,
                                    const float* dh_out, const float* de_out, float* dh_in, float* de_in,
                                    float* const* grads, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                                    void* stream);

/* Planner loss: geomloss.SamplesLoss(loss="sinkhorn", p=2, blur) between the final
 * particle cloud x [N,3] and the desired one y [M,3], uniform weights -- traj_utils.py:69,279.  Debiased
 * Sinkhorn divergence, cost |x-y|^2/2, epsilon-scaling with ratio `scaling` (geomloss default 0.5) from the
 * bounding-box diameter down to blur.  Writes one float to loss_device.  One host synchronisation (the
 * diameter fixes the length of the epsilon schedule). */
size_t gm_sinkhorn_workspace_bytes(int64_t n, int64_t m);
int gm_sinkhorn_divergence(const float* x, int64_t n, const float* y, int64_t m, float blur, float scaling,
                           float* loss_device, void* ws, size_t ws_bytes, void* stream);
/* The losses of a whole block of candidates in one launch sequence (ABI 7) -- the loop of traj_utils.py:279 over the
 * candidates of a CMA-ES generation (traj_utils.py:247-259), `batch` clouds x [batch, N, 3] against y [M, 3]
 * (y_shared != 0: the desired cloud of the planner) or [batch, M, 3].  Pair b gets the epsilon schedule a call of
 * gm_sinkhorn_divergence on it alone would have used (its own bounding-box diameter), and loss_device[b] equals
 * that call's result bit for bit; pairs with shorter schedules idle through the tail of the longest one.
 * diameter > 0: geomloss's `diameter=` keyword -- every pair takes it and NO host synchronisation happens;
 * diameter <= 0 (geomloss default): one host synchronisation per call (the longest schedule = the launch count). */
size_t gm_sinkhorn_batched_workspace_bytes(int64_t batch, int64_t n, int64_t m);
int gm_sinkhorn_divergence_batched(const float* x, int64_t batch, int64_t n, const float* y, int64_t m, int y_shared,
                                   float blur, float scaling, float diameter, float* loss_device /*[batch]*/, void* ws,
                                   size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * One device-resident rollout step = compute_rollout's loop body, rollout_utils.py:38-61 ==
 * cma_objective's, traj_utils.py:123-152:  state_pre -> node features -> radius graph -> csr ->
 * edge features -> forward -> integrate -> state_post.  No host synchronisation.
 * ------------------------------------------------------------------------------------------ */
size_t gm_rollout_workspace_bytes(const gm_model_desc* desc, int64_t n_nodes, int max_neighbours);
int gm_rollout_step(const gm_model* m, float* obs /*[k,N,D] in/out*/, int64_t n_nodes,
                    const gm_feature_desc* fdesc, int max_neighbours, const int32_t* rigid_rank,
                    const float* rigid_target /*[Nr,3] or NULL*/, float* pred_acc_out /*[N,3] or NULL*/,
                    void* rollout_ws, size_t rollout_ws_bytes, void* stream);
/* `steps` device-resident rollout steps without returning to the caller: compute_rollout's loop (rollout_utils.py:38-61)
 * == cma_objective's (traj_utils.py:123-152).  rigid_targets: [n_targets, n_rigid, 3] scripted poses, step i uses pose i;
 * steps beyond n_targets keep the rigid body in place (traj_utils.py:130-131); NULL / 0: no scripted poses.
 * record_last: [steps, N, D] or NULL -- the last frame of the window after the control overwrite of every step, i.e. what
 * the reference appends to `prediction` / `positions` (rollout_utils.py:49, traj_utils.py:137).  No host synchronisation.
 *
 * renumber_every > 0 (ABI 6): the loop runs on a copy of the state whose rows are in grid-cell order -- the radius graph's own
 * cell grid over the last frame, rows of a cell by index, graphs of a batch apart and in order -- re-ordered every
 * `renumber_every` steps (particles move a fraction of a cell per step; 64 is what RolloutEngine uses), so that the per-edge
 * gathers of neighbouring rows share cache lines whatever numbering the caller's simulator emits.  obs, record_last,
 * rigid_rank and rigid_targets stay in the CALLER's numbering: the result is written back through the row map, the records
 * too, and a renumbered rigid row keeps the caller's rank (ranks only index the pose arrays).  The order is computed on the
 * device (a ranking inside the graph build's cells: deterministic, no sort library, no host synchronisation).  A radius graph
 * does not depend on the numbering and every per-node / per-edge function is numbering-free: what changes is the order in
 * which a node's incoming messages are summed (float32 rounding, <= 2e-6 of the plain loop in the tests); the same state
 * gives the same bits every time.  renumber_ws: gm_rollout_renumber_workspace_bytes(fdesc, n_nodes) bytes (two copies of the
 * state + row maps); NULL / 0 with renumber_every == 0, which is the plain loop of ABI 5. */
size_t gm_rollout_renumber_workspace_bytes(const gm_feature_desc* fdesc, int64_t n_nodes);
int gm_rollout(const gm_model* m, float* obs /*[k,N,D] in/out*/, int64_t n_nodes, const gm_feature_desc* fdesc,
               int max_neighbours, const int32_t* rigid_rank, const float* rigid_targets, int64_t n_targets,
               int64_t n_rigid, int64_t steps, float* record_last, int64_t renumber_every, void* renumber_ws,
               size_t renumber_ws_bytes, void* rollout_ws, size_t rollout_ws_bytes, void* stream);
/* Error flags / edge count of the last step (synchronises). */
int gm_rollout_status(const void* rollout_ws, const gm_model_desc* desc, int64_t n_nodes,
                      int max_neighbours, int64_t* n_edges_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart; the reference only wraps model.forward in
 * time.time(), examples/optimise_traj.py:99-103).  Per model handle: when enabled, launches made on behalf of THIS
 * model are bracketed with HIP events on their launch stream; other handles and streams are unaffected.
 * kind: 0 processor edge kernel, 1 processor node kernel, 2 radius-graph build of gm_rollout_step (all its kernels),
 * 3 encoder kernels, 4 the rest of gm_rollout_step (state update + node features, destination sort + block tables + edge
 * features, clears, integration + window shift: kinds 0 .. 4 add up to the step); kind_mask has bit `kind` set for every kind
 * to record (0 disables; a newly enabled kind restarts its counters).  gm_model_profile_query synchronises on the recorded
 * events.  A kind records at most 4096 scopes: once more were opened, *launches comes back NEGATIVE (minus the number opened) and
 * *total_ms covers the first 4096 only.
 * ------------------------------------------------------------------------------------------ */
int gm_model_profile(gm_model* model, int kind_mask);
int gm_model_profile_query(const gm_model* model, int kind, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GNN_MANIP_HIP_H */
