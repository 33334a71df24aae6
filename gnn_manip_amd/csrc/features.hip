// Per-step featurisation, integrator, rollout state update and the scripted rigid-body pose.
// All float32 element-wise work; operations are written with the _rn intrinsics so that no
// multiply-add is contracted: the results are bit-identical to the reference's separate torch
// ops wherever the reference's own op order is defined.
#include "common.h"

namespace gm {

struct FeatParams {
    int k, D, cart, mat, ctrl;
    float r;
    float vm[3], vs[3], am[3], as[3], lo[3], hi[3];
};

static int to_params(const gm_feature_desc* d, FeatParams* p, const char* who) {
    GM_REQUIRE(d != nullptr, GM_ERR_INVALID_ARGUMENT, "%s: null feature descriptor", who);
    GM_REQUIRE(d->k_steps >= 2 && d->k_steps <= 64, GM_ERR_INVALID_ARGUMENT, "%s: k_steps=%d out of range", who, d->k_steps);
    GM_REQUIRE(d->data_dim >= 4 && d->cart_col >= 0 && d->cart_col + 3 <= d->data_dim, GM_ERR_INVALID_ARGUMENT,
               "%s: bad cartesian columns", who);
    GM_REQUIRE(d->material_col >= 0 && d->material_col < d->data_dim, GM_ERR_INVALID_ARGUMENT, "%s: bad material column", who);
    GM_REQUIRE(d->control_col < 0 || d->control_col + 3 <= d->data_dim, GM_ERR_INVALID_ARGUMENT, "%s: bad control columns", who);
    GM_REQUIRE(d->conn_r > 0.0, GM_ERR_INVALID_ARGUMENT, "%s: conn_r must be > 0", who);
    p->k = d->k_steps; p->D = d->data_dim; p->cart = d->cart_col; p->mat = d->material_col; p->ctrl = d->control_col;
    p->r = (float)d->conn_r;
    for (int a = 0; a < 3; ++a) {
        p->vm[a] = d->vel_mean[a]; p->vs[a] = d->vel_std[a];
        p->am[a] = d->acc_mean[a]; p->as[a] = d->acc_std[a];
        p->lo[a] = d->lower_bounds[a]; p->hi[a] = d->upper_bounds[a];
    }
    return GM_OK;
}

// collate_utils.py:217-232 (control) / 195-208; velocities per utils.py:27-40
__global__ void __launch_bounds__(256) node_features_kernel(const float* __restrict__ obs, int64_t n, FeatParams P,
                                                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int F = 3 * (P.k - 1) + 7 + (P.ctrl >= 0 ? 3 : 0);
    float* o = out + i * F;
    const int64_t fs = n * P.D;  // frame stride
    const float* row = obs + i * P.D;
    float prev[3], cur[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) prev[a] = row[P.cart + a];
    for (int t = 1; t < P.k; ++t) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            cur[a] = row[t * fs + P.cart + a];
            o[(t - 1) * 3 + a] = __fdiv_rn(__fsub_rn(__fsub_rn(cur[a], prev[a]), P.vm[a]), P.vs[a]);
            prev[a] = cur[a];
        }
    }
    float* b = o + 3 * (P.k - 1);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = __fdiv_rn(__fsub_rn(cur[a], P.lo[a]), P.r);
        float u = __fdiv_rn(__fsub_rn(P.hi[a], cur[a]), P.r);
        b[a] = fminf(fmaxf(l, -1.f), 1.f);
        b[3 + a] = fminf(fmaxf(u, -1.f), 1.f);
    }
    const float* last = row + (int64_t)(P.k - 1) * fs;
    b[6] = last[P.mat];
    if (P.ctrl >= 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) b[7 + a] = __fdiv_rn(__fsub_rn(last[P.ctrl + a], P.vm[a]), P.vs[a]);
    }
}

__device__ __forceinline__ void edge_feat(const float* __restrict__ pos, int64_t stride, int64_t s, int64_t r, float cr,
                                          float* __restrict__ o) {
    float d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = __fdiv_rn(__fsub_rn(pos[s * stride + a], pos[r * stride + a]), cr);
    float q = __fmul_rn(d[0], d[0]);
    q = __fadd_rn(q, __fmul_rn(d[1], d[1]));
    q = __fadd_rn(q, __fmul_rn(d[2], d[2]));
    *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], __fsqrt_rn(q));
}

// utils.py:43-61, reference edge order
__global__ void __launch_bounds__(256) edge_features_kernel(const float* __restrict__ pos, int64_t stride,
                                                             const int64_t* __restrict__ snd,
                                                             const int64_t* __restrict__ rcv, int64_t e, float cr,
                                                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    edge_feat(pos, stride, snd[i], rcv[i], cr, out + i * 4);
}

// same values, destination-sorted order.  The feature is (p_sender - p_receiver) / r with sender = edge_index[0]: that is
// src[p] -> dst[p] for flow 0 and dst[p] -> src[p] for flow 1 (the header records which row the structure aggregates at)
__global__ void __launch_bounds__(256) edge_features_csr_kernel(const float* __restrict__ pos, int64_t stride,
                                                                 const CsrHeader* __restrict__ hdr,
                                                                 const int* __restrict__ src, const int* __restrict__ dst,
                                                                 float cr, float* __restrict__ out) {
    const int e = hdr->n_edges;
    const bool swap = hdr->flow != 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x)
        edge_feat(pos, stride, swap ? dst[i] : src[i], swap ? src[i] : dst[i], cr, out + i * 4);
}

// rollout_utils.py:145-158
__global__ void __launch_bounds__(256) integrate_kernel(const float* __restrict__ pred, const float* __restrict__ obs,
                                                         int64_t n, FeatParams P, float* __restrict__ next_pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t fs = n * P.D;
    const float* l1 = obs + (int64_t)(P.k - 1) * fs + i * P.D + P.cart;
    const float* l2 = obs + (int64_t)(P.k - 2) * fs + i * P.D + P.cart;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float acc = __fadd_rn(__fmul_rn(pred[i * 3 + a], P.as[a]), P.am[a]);
        float lv = __fsub_rn(l1[a], l2[a]);
        float v = __fadd_rn(lv, acc);
        next_pos[i * 3 + a] = __fadd_rn(l1[a], v);
    }
}

// rollout step, fused: state_pre + node features (one thread owns row i of every frame) -- and, as the step's first launch, the
// resets of everything the step's later launches build on (StepClear: graph / destination-sort workspaces, scan states, agg)
__global__ void __launch_bounds__(256) pre_features_kernel(float* __restrict__ obs, int64_t n, FeatParams P,
                                                            const int* __restrict__ rank, const float* __restrict__ target,
                                                            float* __restrict__ out, StepClear clr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    step_clear_run(clr, i, (long long)gridDim.x * blockDim.x);
    if (i >= n) return;
    const int64_t fs = n * P.D;
    float* row = obs + i * P.D;
    float* last = row + (int64_t)(P.k - 1) * fs;
    if (rank && P.ctrl >= 0) {
        const int rk = rank[i];
        if (rk >= 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float cur = last[P.cart + a];
                last[P.ctrl + a] = target ? __fsub_rn(target[(int64_t)rk * 3 + a], cur) : cur;
            }
        }
    }
    const int F = 3 * (P.k - 1) + 7 + (P.ctrl >= 0 ? 3 : 0);
    float* o = out + i * F;
    float prev[3], cur[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) prev[a] = row[P.cart + a];
    for (int t = 1; t < P.k; ++t) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            cur[a] = row[t * fs + P.cart + a];
            o[(t - 1) * 3 + a] = __fdiv_rn(__fsub_rn(__fsub_rn(cur[a], prev[a]), P.vm[a]), P.vs[a]);
            prev[a] = cur[a];
        }
    }
    float* b = o + 3 * (P.k - 1);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = __fdiv_rn(__fsub_rn(cur[a], P.lo[a]), P.r);
        float u = __fdiv_rn(__fsub_rn(P.hi[a], cur[a]), P.r);
        b[a] = fminf(fmaxf(l, -1.f), 1.f);
        b[3 + a] = fminf(fmaxf(u, -1.f), 1.f);
    }
    b[6] = last[P.mat];
    if (P.ctrl >= 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) b[7 + a] = __fdiv_rn(__fsub_rn(last[P.ctrl + a], P.vm[a]), P.vs[a]);
    }
}

// rollout step, fused: integrator + window shift + write-back (+ optional copy of the prediction)
__global__ void __launch_bounds__(256) integrate_post_kernel(float* __restrict__ obs, int64_t n, FeatParams P,
                                                              const float* __restrict__ pred, const int* __restrict__ rank,
                                                              const float* __restrict__ target, float* __restrict__ pred_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t fs = n * P.D;
    float* row = obs + i * P.D;
    const float* l1 = row + (int64_t)(P.k - 1) * fs + P.cart;
    const float* l2 = row + (int64_t)(P.k - 2) * fs + P.cart;
    float nxt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pa = pred[i * 3 + a];
        if (pred_out) pred_out[i * 3 + a] = pa;
        const float acc = __fadd_rn(__fmul_rn(pa, P.as[a]), P.am[a]);
        const float lv = __fsub_rn(l1[a], l2[a]);
        nxt[a] = __fadd_rn(l1[a], __fadd_rn(lv, acc));
    }
    for (int t = 0; t + 1 < P.k; ++t)
        for (int d = 0; d < P.D; ++d) row[t * fs + d] = row[(t + 1) * fs + d];
    float* last = row + (int64_t)(P.k - 1) * fs;
    const int rk = rank ? rank[i] : -1;
    if (rk < 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) last[P.cart + a] = nxt[a];
    } else if (target) {
#pragma unroll
        for (int a = 0; a < 3; ++a) last[P.cart + a] = target[(int64_t)rk * 3 + a];
    }
}

// rank of each rigid row (material == 1) among the rigid rows; one block, running carry
__global__ void __launch_bounds__(1024) rigid_rank_kernel(const float* __restrict__ obs, int64_t n, FeatParams P,
                                                           int* __restrict__ rank, int* __restrict__ n_rigid) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const float* last = obs + (int64_t)(P.k - 1) * n * P.D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b = 0; b < n; b += 1024) {
        int64_t i = b + threadIdx.x;
        int f = (i < n && last[i * P.D + P.mat] == 1.0f) ? 1 : 0;
        int incl = f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (i < n) rank[i] = f ? base + incl - 1 : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && n_rigid) *n_rigid = carry_s;
}

// rollout_utils.py:40-47 / traj_utils.py:126-134
__global__ void __launch_bounds__(256) state_pre_kernel(float* __restrict__ obs, int64_t n, FeatParams P,
                                                         const int* __restrict__ rank, const float* __restrict__ target) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rk = rank[i];
    if (rk < 0) return;
    float* last = obs + (int64_t)(P.k - 1) * n * P.D + i * P.D;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float cur = last[P.cart + a];
        last[P.ctrl + a] = target ? __fsub_rn(target[(int64_t)rk * 3 + a], cur) : cur;
    }
}

// rollout_utils.py:53-61 / traj_utils.py:146-152
__global__ void __launch_bounds__(256) state_post_kernel(float* __restrict__ obs, int64_t n, FeatParams P,
                                                          const float* __restrict__ next_pos,
                                                          const int* __restrict__ rank, const float* __restrict__ target) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t fs = n * P.D;
    float* row = obs + i * P.D;
    for (int t = 0; t + 1 < P.k; ++t)
        for (int d = 0; d < P.D; ++d) row[t * fs + d] = row[(t + 1) * fs + d];
    float* last = row + (int64_t)(P.k - 1) * fs;
    const int rk = rank ? rank[i] : -1;
    if (rk < 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) last[P.cart + a] = next_pos[i * 3 + a];
    } else if (target) {
#pragma unroll
        for (int a = 0; a < 3; ++a) last[P.cart + a] = target[(int64_t)rk * 3 + a];
    }  // rigid row without a scripted pose keeps its pre-step row (traj_utils.py:150-152)
}

// traj_utils.py:167-194: rotation about X in the cup frame with the y/z axis swap
__global__ void __launch_bounds__(256) rigid_transform_kernel(const float* __restrict__ init, int64_t nr,
                                                               const float* __restrict__ cst, int64_t steps, float tx,
                                                               float ty, float tz, float* __restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nr * steps) return;
    const int64_t t = id / nr, i = id - t * nr;
    const float c = cst[t * 3], s = cst[t * 3 + 1], typ = cst[t * 3 + 2];
    const float i0 = __fsub_rn(tx, init[i * 3 + 0]);
    const float i1 = __fsub_rn(ty, init[i * 3 + 2]);
    const float i2 = __fsub_rn(tz, init[i * 3 + 1]);
    const float p0 = __fadd_rn(i0, tx);
    const float p1 = __fadd_rn(__fmaf_rn(-s, i2, __fmul_rn(c, i1)), typ);
    const float p2 = __fadd_rn(__fmaf_rn(c, i2, __fmul_rn(s, i1)), tz);
    out[id * 3 + 0] = p0;
    out[id * 3 + 2] = p1;
    out[id * 3 + 1] = p2;
}

// ---- renumbered rollout: rows of the [k, N, D] state move as whole rows (D floats), one thread per (frame, row)
__global__ void __launch_bounds__(256) renumber_gather_kernel(const float* __restrict__ in, float* __restrict__ out, int k, int64_t n, int D,
                                                               const int* __restrict__ perm, const GraphHeader* __restrict__ ghdr,
                                                               const int* __restrict__ total_in, int* __restrict__ total_out,
                                                               const int* __restrict__ rank_caller, int* __restrict__ rank_out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)k * n) return;
    const int64_t t = id / n, j = id - t * n;
    const int64_t src = ghdr->order_skip ? j : (int64_t)perm[j];
    const float* a = in + (t * n + src) * D;
    float* b = out + (t * n + j) * D;
    for (int d = 0; d < D; ++d) b[d] = a[d];
    if (t == 0) {
        const int tot = total_in ? total_in[src] : (int)src;
        total_out[j] = tot;
        if (rank_out) rank_out[j] = rank_caller[tot];
    }
}
__global__ void __launch_bounds__(256) renumber_scatter_kernel(const float* __restrict__ in, float* __restrict__ out, int frames, int64_t n, int D,
                                                                const int* __restrict__ total) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)frames * n) return;
    const int64_t t = id / n, j = id - t * n;
    const float* a = in + (t * n + j) * D;
    float* b = out + (t * n + (int64_t)total[j]) * D;
    for (int d = 0; d < D; ++d) b[d] = a[d];
}
int renumber_gather(const float* in, float* out, int k, int64_t n, int D, const int* perm, const void* graph_ws, const int* total_in,
                    int* total_out, const int* rank_caller, int* rank_out, hipStream_t s) {
    if (n <= 0 || k <= 0) return GM_OK;
    hipLaunchKernelGGL(renumber_gather_kernel, dim3((unsigned)cdiv((int64_t)k * n, 256)), dim3(256), 0, s, in, out, k, n, D, perm,
                       static_cast<const GraphHeader*>(graph_ws), total_in, total_out, rank_caller, rank_out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int renumber_scatter(const float* in, float* out, int frames, int64_t n, int D, const int* total, hipStream_t s) {
    if (n <= 0 || frames <= 0) return GM_OK;
    hipLaunchKernelGGL(renumber_scatter_kernel, dim3((unsigned)cdiv((int64_t)frames * n, 256)), dim3(256), 0, s, in, out, frames, n, D, total);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int rollout_pre_features(float* obs, int64_t n, const gm_feature_desc* d, const int* rank, const float* target, float* out,
                         hipStream_t s, const StepClear* clear) {
    FeatParams P;
    int rc = to_params(d, &P, "gm_rollout_step");
    if (rc != GM_OK) return rc;
    hipLaunchKernelGGL(pre_features_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, obs, n, P, rank, target, out, clear ? *clear : StepClear{});
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int rollout_integrate_post(float* obs, int64_t n, const gm_feature_desc* d, const float* pred, const int* rank,
                           const float* target, float* pred_out, hipStream_t s) {
    FeatParams P;
    int rc = to_params(d, &P, "gm_rollout_step");
    if (rc != GM_OK) return rc;
    hipLaunchKernelGGL(integrate_post_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, obs, n, P, pred, rank, target, pred_out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm

using namespace gm;

extern "C" {

int gm_node_features(const float* obs, int64_t n, const gm_feature_desc* desc, float* out, void* stream) {
    gm::DevGuard dev_guard(obs);
    FeatParams P;
    int rc = to_params(desc, &P, "gm_node_features");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(n >= 0 && (n == 0 || (obs && out)), GM_ERR_INVALID_ARGUMENT, "gm_node_features: null pointer");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(node_features_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, obs, n, P, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_edge_features(const float* pos, int64_t pos_stride, const int64_t* senders, const int64_t* receivers,
                     int64_t e, float conn_r, float* out, void* stream) {
    gm::DevGuard dev_guard(pos);
    GM_REQUIRE(e >= 0 && (e == 0 || (pos && senders && receivers && out)), GM_ERR_INVALID_ARGUMENT, "gm_edge_features: null pointer");
    GM_REQUIRE(conn_r > 0.f && pos_stride >= 3, GM_ERR_INVALID_ARGUMENT, "gm_edge_features: bad conn_r / stride");
    if (e == 0) return GM_OK;
    hipLaunchKernelGGL(edge_features_kernel, dim3((unsigned)cdiv(e, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       pos_stride, senders, receivers, e, conn_r, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_edge_features_csr(const float* pos, int64_t pos_stride, const void* csr_ws, int64_t n, int64_t cap,
                         float conn_r, float* out, void* stream) {
    gm::DevGuard dev_guard(pos);
    GM_REQUIRE(pos && csr_ws && out, GM_ERR_INVALID_ARGUMENT, "gm_edge_features_csr: null pointer");
    GM_REQUIRE(conn_r > 0.f && pos_stride >= 3, GM_ERR_INVALID_ARGUMENT, "gm_edge_features_csr: bad conn_r / stride");
    if (cap == 0) return GM_OK;
    CsrWs c = carve_csr(const_cast<void*>(csr_ws), n, cap);
    int64_t nb = cdiv(cap, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(edge_features_csr_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pos, pos_stride,
                       c.hdr, c.src, c.dst, conn_r, out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_integrate(const float* pred, const float* obs, int64_t n, const gm_feature_desc* desc, float* next_pos,
                 void* stream) {
    gm::DevGuard dev_guard(obs);
    FeatParams P;
    int rc = to_params(desc, &P, "gm_integrate");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(n >= 0 && (n == 0 || (pred && obs && next_pos)), GM_ERR_INVALID_ARGUMENT, "gm_integrate: null pointer");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, obs, n, P, next_pos);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_rigid_rank(const float* obs, int64_t n, const gm_feature_desc* desc, int32_t* rank, int32_t* n_rigid_dev,
                  void* stream) {
    gm::DevGuard dev_guard(obs);
    FeatParams P;
    int rc = to_params(desc, &P, "gm_rigid_rank");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(n >= 0 && (n == 0 || (obs && rank)), GM_ERR_INVALID_ARGUMENT, "gm_rigid_rank: null pointer");
    hipLaunchKernelGGL(rigid_rank_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, obs, n, P, rank, n_rigid_dev);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_state_pre(float* obs, int64_t n, const gm_feature_desc* desc, const int32_t* rank, const float* target,
                 void* stream) {
    gm::DevGuard dev_guard(obs);
    FeatParams P;
    int rc = to_params(desc, &P, "gm_state_pre");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(P.ctrl >= 0, GM_ERR_INVALID_ARGUMENT, "gm_state_pre: descriptor has no control columns");
    GM_REQUIRE(n >= 0 && (n == 0 || (obs && rank)), GM_ERR_INVALID_ARGUMENT, "gm_state_pre: null pointer");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(state_pre_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, obs, n, P, rank, target);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_state_post(float* obs, int64_t n, const gm_feature_desc* desc, const float* next_pos, const int32_t* rank,
                  const float* target, void* stream) {
    gm::DevGuard dev_guard(obs);
    FeatParams P;
    int rc = to_params(desc, &P, "gm_state_post");
    if (rc != GM_OK) return rc;
    GM_REQUIRE(n >= 0 && (n == 0 || (obs && next_pos)), GM_ERR_INVALID_ARGUMENT, "gm_state_post: null pointer");
    if (n == 0) return GM_OK;
    hipLaunchKernelGGL(state_post_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, obs, n, P,
                       next_pos, rank, target);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_rigid_transform(const float* rigid_init, int64_t nr, const float* cst, int64_t steps, const float ty_init[3],
                       float* out, void* stream) {
    gm::DevGuard dev_guard(rigid_init);
    GM_REQUIRE(nr >= 0 && steps >= 0 && ty_init, GM_ERR_INVALID_ARGUMENT, "gm_rigid_transform: bad sizes");
    if (nr == 0 || steps == 0) return GM_OK;
    GM_REQUIRE(rigid_init && cst && out, GM_ERR_INVALID_ARGUMENT, "gm_rigid_transform: null pointer");
    hipLaunchKernelGGL(rigid_transform_kernel, dim3((unsigned)cdiv(nr * steps, 256)), dim3(256), 0, (hipStream_t)stream,
                       rigid_init, nr, cst, steps, ty_init[0], ty_init[1], ty_init[2], out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // extern "C"
