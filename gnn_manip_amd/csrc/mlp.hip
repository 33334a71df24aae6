// Weight packing shared by the kernel families and the launch dispatch of the edge / node MLP kernels:
//   processor phi_e at hidden 128 / num_layers 2  -> systolic fp16 x 3 kernel (hedge.hip)
//   everything else                               -> streamed fp16 x 3 kernels (hmlp.hip)
//   kernel choices 1 .. 4 (development builds)    -> round-1 fp32 / bf16 x 6 kernels (mlp_dev_kernels.hip, not in the product)
// Reference semantics: EncProcDecGNN.forward / _process / _build_mlp, gnn_manip/models/epd_gnn.py:72-105.
#include <string.h>
#include "common.h"
#include "mlp.h"

#include "mlp_dev.h"
#include "hedge.h"
#include "hmlp.h"

namespace gm {

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
// transpose != 0 packs T = (W[:, col0:col0+out_rows])^T: T[row][col] = W[col][col0 + row], kvalid = rows of W
__global__ void __launch_bounds__(256) pack_linear_kernel(const float* __restrict__ W, int out_rows, int ld, int col0,
                                                           int kvalid, int nkq, int njb, int stages, int transpose,
                                                           float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)stages * STAGE_FLOATS;
    if (idx >= total) return;
    const int p = (int)(idx / PIECE_FLOATS);
    const int within = (int)(idx % PIECE_FLOATS);
    const int lane = within >> 2, t = within & 3;
    const int i = lane & 31, hi = lane >> 5;
    float v = 0.f;
    if (p < nkq * njb) {
        const int kq = p / njb, jb = p % njb;
        const int row = 32 * jb + i, col = 8 * kq + 4 * hi + t;
        if (row < out_rows && col < kvalid) v = transpose ? W[(int64_t)col * ld + col0 + row] : W[(int64_t)row * ld + col0 + col];
    }
    dst[idx] = v;
}

// 16x16x4 operand image: piece(kq, jb)[lane = (i = lane & 15, g = lane >> 4)][r] = W[16 jb + i][16 kq + 4 g + r]
__global__ void __launch_bounds__(256) pack_linear16_kernel(const float* __restrict__ W, int out_rows, int ld, int col0,
                                                             int kvalid, int nkq, int njb, int stages,
                                                             float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)stages * STAGE_FLOATS;
    if (idx >= total) return;
    const int p = (int)(idx / PIECE_FLOATS);
    const int within = (int)(idx % PIECE_FLOATS);
    const int lane = within >> 2, r = within & 3;
    const int i = lane & 15, g = lane >> 4;
    float v = 0.f;
    if (p < nkq * njb) {
        const int kq = p / njb, jb = p % njb;
        const int row = 16 * jb + i, col = 16 * kq + 4 * g + r;
        if (row < out_rows && col < kvalid) v = W[(int64_t)row * ld + col0 + col];
    }
    dst[idx] = v;
}

int layer_stages16(int k, int out) {
    const int nkq = (k + 15) / 16, njb = (out + 15) / 16;
    return (nkq * njb + STAGE_PIECES - 1) / STAGE_PIECES;
}

int pack_linear16(const float* W, int out_rows, int ld, int col0, int kvalid, float* dst, hipStream_t s) {
    const int nkq = (kvalid + 15) / 16, njb = (out_rows + 15) / 16;
    const int stages = layer_stages16(kvalid, out_rows);
    const int64_t total = (int64_t)stages * STAGE_FLOATS;
    hipLaunchKernelGGL(pack_linear16_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, W, out_rows, ld, col0,
                       kvalid, nkq, njb, stages, dst);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int layer_stages(int k, int out) {
    const int nkq = (k + 7) / 8, njb = (out + 31) / 32;
    return (nkq * njb + STAGE_PIECES - 1) / STAGE_PIECES;
}

int pack_linear(const float* W, int out_rows, int ld, int col0, int kvalid, float* dst, hipStream_t s) {
    const int nkq = (kvalid + 7) / 8, njb = (out_rows + 31) / 32;
    const int stages = layer_stages(kvalid, out_rows);
    const int64_t total = (int64_t)stages * STAGE_FLOATS;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, W, out_rows, ld, col0,
                       kvalid, nkq, njb, stages, 0, dst);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// Operand image of the TRANSPOSED sub-block (W[0:w_rows, col0:col0+ksub])^T, i.e. a Linear with ksub outputs and
// w_rows inputs: the weight of the backward (input-gradient) product dX = dZ . W.
int pack_linear_t(const float* W, int w_rows, int ld, int col0, int ksub, float* dst, hipStream_t s) {
    const int nkq = (w_rows + 7) / 8, njb = (ksub + 31) / 32;
    const int stages = layer_stages(w_rows, ksub);
    const int64_t total = (int64_t)stages * STAGE_FLOATS;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, W, ksub, ld, col0,
                       w_rows, nkq, njb, stages, 1, dst);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// one launch for many Linears (blockIdx.y = job): same images as pack_linear_kernel / pack_linear16_kernel
__global__ void __launch_bounds__(256) pack_batch_kernel(PackJobs J, float* __restrict__ base32, float* __restrict__ base16) {
    const PackJob j = J.job[blockIdx.y];
    const int kw = j.layout ? 16 : 8, jw = j.layout ? 16 : 32;
    const int nkq = (j.kvalid + kw - 1) / kw, njb = (j.out_rows + jw - 1) / jw;
    const int64_t total = (int64_t)((nkq * njb + STAGE_PIECES - 1) / STAGE_PIECES) * STAGE_FLOATS;
    float* dst = (j.layout ? base16 : base32) + j.dst_off;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx / PIECE_FLOATS);
        const int within = (int)(idx % PIECE_FLOATS);
        const int lane = within >> 2, t = within & 3;
        float v = 0.f;
        if (p < nkq * njb) {
            const int kq = p / njb, jb = p % njb;
            int row, col;
            if (j.layout) { row = 16 * jb + (lane & 15); col = 16 * kq + 4 * (lane >> 4) + t; }
            else { row = 32 * jb + (lane & 31); col = 8 * kq + 4 * (lane >> 5) + t; }
            if (row < j.out_rows && col < j.kvalid) v = j.W[(int64_t)row * j.ld + j.col0 + col];
        }
        dst[idx] = v;
    }
}

__global__ void __launch_bounds__(256) vec_batch_kernel(VecJobs J, float* __restrict__ base) {
    const VecJob j = J.job[blockIdx.x];   // blockIdx.y: slice of the tensor (a weight matrix is 16k .. 49k floats: not one workgroup's job)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < max(j.count, j.zero_to); i += gridDim.y * blockDim.x)
        base[j.dst_off + i] = i < j.count ? j.src[i] : 0.f;
}

int launch_pack_batch(const PackJobs& jobs, float* base32, float* base16, hipStream_t s) {
    if (jobs.n <= 0) return GM_OK;
    hipLaunchKernelGGL(pack_batch_kernel, dim3(16, jobs.n), dim3(256), 0, s, jobs, base32, base16);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_vec_batch(const VecJobs& jobs, float* base, hipStream_t s) {
    if (jobs.n <= 0) return GM_OK;
    hipLaunchKernelGGL(vec_batch_kernel, dim3(jobs.n, 16), dim3(256), 0, s, jobs, base);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ------------------------------------------------------------------------------------------
// launchers (the round-1 fp32 / bf16 x 6 kernels live in mlp_dev_kernels.hip, development builds only)
// ------------------------------------------------------------------------------------------

enum : int { EK_AUTO = 0, EK_16 = 1, EK_CLASSIC = 2, EK_B3 = 3, EK_B3P = 4, EK_SYS = 5, EK_HM = 6 };


int launch_edge(int H, int NL, bool enc, const EdgeArgs& a_in, int64_t edge_capacity, hipStream_t s) {
    if (edge_capacity <= 0) return GM_OK;
    EdgeArgs a = a_in;
    a.debug = 0;   // timing ablations are a development build's business (the kernels keep the hooks)
    a.stamps = nullptr;
    const int choice = a.kernel_choice;
    const bool fp32_forms = choice >= EK_16 && choice <= EK_B3P;   // explicitly selected fp32 / bf16 x 6 kernels
    // the systolic fp16 x 3 kernel (hedge.hip): processor step of the fused forward (rows in sorted order, device-side
    // edge count, block tables present), hidden 128 / num_layers 2
    const bool sys_ok = H == 128 && (a.h_valid == 0 || a.h_valid == 128) && NL == 2 && !enc && a.wstream_h3 && a.edge_blocks && a.hdr && a.agg && !a.eid && !a.eid_out;
    if (sys_ok && (choice == EK_AUTO || choice == EK_SYS) && edge_sys_fits(a.n_nodes_tab, edge_capacity))
        return launch_edge_sys(a, carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, edge_capacity), s);
    // the encoder phi_e in the same weight-stationary form: 4 raw features per edge, rows in sorted order (the rollout path)
    if (enc && H == 128 && (a.h_valid == 0 || a.h_valid == 128) && NL == 2 && a.wstream_h3 && a.hdr && !a.eid && a.k1 == 4 &&
        (choice == EK_AUTO || choice == EK_SYS))
        return launch_edge_sys_enc(a, s);
    // the streamed fp16 x 3 kernels (hmlp.hip): every other case
    if (!fp32_forms && a.wstream_hm && hm_supported(H) && (enc || (a.edge_blocks && (a.side || !a.agg)))) {
        HmEdgeArgs h{};
        h.hdr = a.hdr; h.n_edges_host = a.n_edges_host; h.dst = a.dst; h.src = a.src; h.eid = a.eid; h.eid_out = a.eid_out;
        h.P = a.P; h.e_in = a.e_in; h.e_out = a.e_out; h.agg = a.agg; h.w = a.wstream_hm; h.ln_g = a.ln_g; h.ln_b = a.ln_b;
        h.eps = a.eps; h.residual = a.residual; h.k1 = a.k1; h.nl = NL; h.prof = a.prof;
        h.h_valid = a.h_valid > 0 ? a.h_valid : H;
        h.flags = a.hdr ? const_cast<int*>(&a.hdr->error_flags) : nullptr;
        if (!enc) {
            const EdgeBlocks t = carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, edge_capacity);
            h.blk = t.blk; h.tab = t.hdr; h.head = t.head; h.side = a.side;
        }
        return launch_edge_hm(H, enc, h, s);
    }
#ifndef GM_DEV_KERNELS
    (void)fp32_forms;
    GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "edge kernel: hidden_size=%d, kernel choice %d: no kernel of this library build takes this launch", H, choice);
#else
    (void)fp32_forms;
    return launch_edge_dev(H, NL, enc, a, edge_capacity, s);
#endif
}

int launch_node(int H, int NL, int mode, const NodeArgs& a, hipStream_t s) {
    if (a.n_nodes <= 0) return GM_OK;
    if (!(a.kernel_choice >= EK_16 && a.kernel_choice <= EK_B3P) && hm_supported(H) && (mode == 2 ? a.tail_hm != nullptr : a.wstream_hm != nullptr)) {
        HmNodeArgs h{};
        h.n_nodes = a.n_nodes; h.x_in = a.x_in; h.k1 = a.k1; h.agg = a.agg; h.agg_clear = a.agg_clear; h.h_out = a.h_out;
        h.residual = a.residual; h.w = a.wstream_hm; h.ln_g = a.ln_g; h.ln_b = a.ln_b; h.eps = a.eps; h.nl = NL;
        h.tail = mode == 2 ? 1 : a.tail; h.w_tail = a.tail_hm; h.P_out = a.P_out; h.dec_out = a.dec_out; h.out_dim = a.out_dim; h.prof = a.prof;
        h.flags = a.err_flags;
        h.h_valid = a.h_valid > 0 ? a.h_valid : H;
        if (mode == 1 && a.edge_blocks && a.side) {   // lists + side buffer of the edge kernel's head partials (hedge.h)
            const EdgeBlocks t = carve_edge_blocks(const_cast<int*>(a.edge_blocks), a.n_nodes_tab, a.edge_capacity_tab);
            h.stitch = t.stitch; h.head = t.head; h.side = a.side; h.tab = t.hdr;
        }
        return launch_node_hm(H, mode, h, s);
    }
#ifndef GM_DEV_KERNELS
    GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "node kernel: hidden_size=%d, kernel choice %d: no kernel of this library build takes this launch", H, a.kernel_choice);
#else
    return launch_node_dev(H, NL, mode, a, s);
#endif
}

}  // namespace gm
