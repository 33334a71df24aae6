// Launch dispatch of the edge / node MLP kernels (+ the batched vector copy of a weight load):
//   processor phi_e (and the edge encoder of the rollout path) at hidden 128 / num_layers 2  -> systolic fp16 x 3 kernels (hedge.hip)
//   everything else                                                                         -> streamed fp16 x 3 kernels (hmlp.hip)
// (The round-1 fp32 / bf16 x 6 kernels behind kernel choices 1 .. 4 were removed in round 5; git history keeps them.)
// Reference semantics: EncProcDecGNN.forward / _process / _build_mlp, gnn_manip/models/epd_gnn.py:72-105.
#include <string.h>
#include "common.h"
#include "mlp.h"

#include "hedge.h"
#include "hmlp.h"

namespace gm {

// ------------------------------------------------------------------------------------------
// copies of small vectors / raw tensors into the model's own buffers (one launch for many)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vec_batch_kernel(VecJobs J, float* __restrict__ base) {
    const VecJob j = J.job[blockIdx.x];   // blockIdx.y: slice of the tensor (a weight matrix is 16k .. 49k floats: not one workgroup's job)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < max(j.count, j.zero_to); i += gridDim.y * blockDim.x)
        base[j.dst_off + i] = i < j.count ? j.src[i] : 0.f;
}

int launch_vec_batch(const VecJobs& jobs, float* base, hipStream_t s) {
    if (jobs.n <= 0) return GM_OK;
    hipLaunchKernelGGL(vec_batch_kernel, dim3(jobs.n, 16), dim3(256), 0, s, jobs, base);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

enum : int { EK_AUTO = 0, EK_SYS = 5, EK_HM = 6, EK_SYS_ALL = 7 };   // gm_model_set_edge_kernel (1 .. 4 were the removed round-1 kernels)


// the systolic fp16 x 3 kernel (hedge.hip): processor step of the fused forward (rows in sorted order, device-side edge count, block
// tables present), hidden 128 / num_layers 2
bool edge_launch_is_sys(int H, int NL, const EdgeArgs& a, int64_t edge_capacity) {
    const int choice = a.kernel_choice;
    const bool sys_ok = H == 128 && (a.h_valid == 0 || a.h_valid == 128) && NL == 2 && a.wstream_h3 && a.edge_blocks && a.hdr && a.agg && !a.eid && !a.eid_out;
    return edge_capacity > 0 && sys_ok && (choice == EK_AUTO || choice == EK_SYS || choice == EK_SYS_ALL) && edge_sys_fits(a.n_nodes_tab, edge_capacity);
}

int launch_edge(int H, int NL, bool enc, const EdgeArgs& a_in, int64_t edge_capacity, hipStream_t s) {
    if (edge_capacity <= 0) return GM_OK;
    const EdgeArgs& a = a_in;
    const int choice = a.kernel_choice;
    if (!enc && edge_launch_is_sys(H, NL, a, edge_capacity))
        return launch_edge_sys(a, carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, edge_capacity), edge_capacity, s);
    GM_REQUIRE(enc || !a.P_prescaled, GM_ERR_INVALID_ARGUMENT, "edge kernel: P carries the systolic kernel's scale, but the launch is not its");
    // the encoder phi_e in the same weight-stationary form: 4 raw features per edge, rows in sorted order (the rollout path)
    if (enc && H == 128 && (a.h_valid == 0 || a.h_valid == 128) && NL == 2 && a.wstream_h3 && a.hdr && !a.eid && a.k1 == 4 &&
        (choice == EK_AUTO || choice == EK_SYS || choice == EK_SYS_ALL))
    {
        if (a.zero_pad_rows && a.pad_rows_done) *a.pad_rows_done = 1;   // the systolic encoder zeroes them in the same launch
        return launch_edge_sys_enc(a, s);
    }
    // the streamed fp16 x 3 kernels (hmlp.hip): every other case
    if (a.wstream_hm && hm_supported(H) && (enc || (a.edge_blocks && (a.side || !a.agg)))) {
        HmEdgeArgs h{};
        h.hdr = a.hdr; h.n_edges_host = a.n_edges_host; h.dst = a.dst; h.src = a.src; h.eid = a.eid; h.eid_out = a.eid_out;
        h.P = a.P; h.e_in = a.e_in; h.e_out = a.e_out; h.agg = a.agg; h.w = a.wstream_hm; h.ln_g = a.ln_g; h.ln_b = a.ln_b;
        h.eps = a.eps; h.residual = a.residual; h.discard_e_out = a.discard_e_out; h.k1 = a.k1; h.nl = NL; h.prof = a.prof;
        h.h_valid = a.h_valid > 0 ? a.h_valid : H;
        h.flags = a.hdr ? const_cast<int*>(&a.hdr->error_flags) : nullptr;
        if (!enc) {
            const EdgeBlocks t = carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, edge_capacity);
            h.blk = t.blk; h.tab = t.hdr; h.head = t.head; h.side = a.side;
        }
        return launch_edge_hm(H, enc, h, s);
    }
    GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "edge kernel: hidden_size=%d, kernel choice %d: no kernel of this library takes this launch", H, choice);
}

int launch_node(int H, int NL, int mode, const NodeArgs& a, hipStream_t s) {
    if (a.n_nodes <= 0) return GM_OK;
    // mode 3: the decoder alone on h (the tail of a step whose node MLP ran in the systolic kernel)
    if (hm_supported(H) && (mode >= 2 ? a.tail_hm != nullptr : a.wstream_hm != nullptr)) {
        HmNodeArgs h{};
        h.n_nodes = a.n_nodes; h.x_in = a.x_in; h.k1 = a.k1; h.agg = a.agg; h.h_out = a.h_out;
        h.residual = a.residual; h.w = a.wstream_hm; h.ln_g = a.ln_g; h.ln_b = a.ln_b; h.eps = a.eps; h.nl = NL;
        h.tail = mode == 2 ? 1 : (mode == 3 ? 2 : a.tail); h.w_tail = a.tail_hm; h.P_out = a.P_out; h.p_scale = a.p_scale; h.dec_out = a.dec_out; h.out_dim = a.out_dim; h.prof = a.prof;
        h.flags = a.err_flags;
        h.h_valid = a.h_valid > 0 ? a.h_valid : H;
        if (mode == 1 && a.edge_blocks && a.side) {   // lists + side buffer of the edge kernel's head partials (hedge.h)
            const EdgeBlocks t = carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, a.edge_capacity_tab);
            h.stitch = t.stitch; h.head = t.head; h.side = a.side; h.tab = t.hdr;
        }
        return launch_node_hm(H, mode == 3 ? 2 : mode, h, s);
    }
    (void)NL;
    GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "node kernel: hidden_size=%d, kernel choice %d: no kernel of this library takes this launch", H, a.kernel_choice);
}

}  // namespace gm
