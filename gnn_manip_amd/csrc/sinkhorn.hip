// Debiased Sinkhorn divergence between two uniform point clouds in 3-D, on the device -- the planner's
// loss `geomloss.SamplesLoss(loss="sinkhorn", p=2, blur=.05)` (reference call sites traj_utils.py:69,279;
// SURVEY.md section 8f-2).  geomloss is not part of the reference tree (pip dependency, version not pinned,
// environment.yml:25): this restates its published algorithm (Feydy et al., "Interpolating between Optimal
// Transport and MMD using Sinkhorn Divergences", AISTATS 2019; geomloss sinkhorn_divergence.py: log-domain
// symmetric Sinkhorn with epsilon-scaling, debiasing, one final extrapolation):
//     C(x, y) = |x - y|^2 / 2,  eps schedule: diameter^2, then exp(arange(2 log diameter, 2 log blur, 2 log scaling)), blur^2
//     softmin_eps(C, h)_i = -eps log sum_j exp(h_j - C_ij / eps)
//     S = mean_i (b_x - a_x)_i + mean_j (a_y - b_y)_j
// No N x M matrix is ever stored: every softmin recomputes the distances it needs from the coordinates
// ("online" reduction): 6 flops + one exp per pair, two passes (max, then sum) -- exp-throughput bound,
// the clouds and potentials live in L2.
#include <math.h>
#include <vector>
#include "common.h"

namespace gm {

constexpr int SK_RB = 4;  // rows per wave

// Per-pair plan of a batch (written by sinkhorn_plan_kernel, read by every launch that follows): the diameter that fixes the
// pair's epsilon schedule and the schedule's length.  Pairs of a batch have schedules of their own -- exactly what one call per
// pair would have used -- so a launch covers the batch's longest schedule and a pair whose schedule has ended passes its
// potentials through.
struct SkPlan {
    float diameter;   // of the bounding box of x_b u y_b (geomloss max_diameter), or the caller's
    int n_eps;        // length of the epsilon list: 2 + numpy.arange length; 0: both clouds are one point (loss 0); -1: non-finite input
};

// epsilon number `it` of a pair's schedule (p = 2): [diameter^2] + exp(arange(2 log diameter, 2 log blur, 2 log scaling)) + [blur^2]
__device__ __forceinline__ double sk_eps(const SkPlan& pl, int it, double blur, double scaling) {
    const double d = (double)pl.diameter;
    if (it <= 0) return d * d;
    if (it >= pl.n_eps - 1) return blur * blur;
    return exp(2.0 * log(d) + (double)(it - 1) * 2.0 * log(scaling));
}

// Softmin of pair b = blockIdx.y at epsilon number `it` of ITS schedule (it < 0: the last one; the initialisation runs at it = 0):
//   out[i] = w_old * old[i] + w_new * ( -eps * log sum_j exp(logw + f[j] / eps - |p_i - q_j|^2 / (2 eps)) )
// it >= the pair's schedule length: out = old (the pair is done with its symmetrised updates and waits for the batch).
__global__ void __launch_bounds__(256) softmin_kernel(const float* __restrict__ P_all, int R, size_t p_stride, const float* __restrict__ Q_all, int S,
                                                       size_t q_stride, const float* __restrict__ f_all, float logw,
                                                       const SkPlan* __restrict__ plan, int it, float blur, float scaling,
                                                       const float* __restrict__ old_all, float w_old, float w_new, float* __restrict__ out_all) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int r0 = (blockIdx.x * 4 + wave) * SK_RB;
    if (r0 >= R) return;
    const SkPlan pl = plan[b];
    const float* P = P_all + (size_t)b * p_stride;
    const float* Q = Q_all + (size_t)b * q_stride;
    const float* f = f_all ? f_all + (size_t)b * S : nullptr;
    const float* old = old_all ? old_all + (size_t)b * R : nullptr;
    float* out = out_all + (size_t)b * R;
    if (pl.n_eps <= 0 || (it >= 0 && it >= pl.n_eps)) {   // nothing to do for this pair in this launch
        if (lane < SK_RB && r0 + lane < R) out[r0 + lane] = (old && pl.n_eps > 0) ? old[r0 + lane] : 0.f;
        return;
    }
    const double eps_d = sk_eps(pl, it < 0 ? pl.n_eps - 1 : it, (double)blur, (double)scaling);
    const float inv_eps = (float)(1.0 / eps_d), eps = (float)eps_d;
    float px[SK_RB], py[SK_RB], pz[SK_RB];
#pragma unroll
    for (int r = 0; r < SK_RB; ++r) {
        const int rr = min(r0 + r, R - 1);
        px[r] = P[3 * rr];
        py[r] = P[3 * rr + 1];
        pz[r] = P[3 * rr + 2];
    }
    const float hc = 0.5f * inv_eps;
    float m[SK_RB];
#pragma unroll
    for (int r = 0; r < SK_RB; ++r) m[r] = -INFINITY;
    for (int j = lane; j < S; j += 64) {
        const float qx = Q[3 * j], qy = Q[3 * j + 1], qz = Q[3 * j + 2];
        const float h = logw + (f ? f[j] * inv_eps : 0.f);
#pragma unroll
        for (int r = 0; r < SK_RB; ++r) {
            const float dx = px[r] - qx, dy = py[r] - qy, dz = pz[r] - qz;
            m[r] = fmaxf(m[r], h - (dx * dx + dy * dy + dz * dz) * hc);
        }
    }
#pragma unroll
    for (int r = 0; r < SK_RB; ++r)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
    float s[SK_RB];
#pragma unroll
    for (int r = 0; r < SK_RB; ++r) s[r] = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float qx = Q[3 * j], qy = Q[3 * j + 1], qz = Q[3 * j + 2];
        const float h = logw + (f ? f[j] * inv_eps : 0.f);
#pragma unroll
        for (int r = 0; r < SK_RB; ++r) {
            const float dx = px[r] - qx, dy = py[r] - qy, dz = pz[r] - qz;
            s[r] += __expf(h - (dx * dx + dy * dy + dz * dz) * hc - m[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < SK_RB; ++r)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s[r] += __shfl_xor(s[r], o, 64);
    if (lane < SK_RB && r0 + lane < R) {
        float mm = m[0], ss = s[0];
#pragma unroll
        for (int r = 1; r < SK_RB; ++r)
            if (lane == r) { mm = m[r]; ss = s[r]; }
        const float v = -eps * (mm + __logf(ss));
        out[r0 + lane] = (old ? w_old * old[r0 + lane] : 0.f) + w_new * v;
    }
}

// One workgroup per pair: bounding box of x_b u y_b -> diameter (or the caller's) and the length of the pair's epsilon schedule;
// max_eps[0] = the batch's longest schedule (atomicMax), max_eps[1] |= 1 if a coordinate is not finite.
__global__ void __launch_bounds__(256) sinkhorn_plan_kernel(const float* __restrict__ x_all, int n, const float* __restrict__ y_all, int m, size_t y_stride,
                                                             float blur, float scaling, float diameter_given, SkPlan* __restrict__ plan,
                                                             int* __restrict__ max_eps) {
    __shared__ float lo[3][256], hi[3][256];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* x = x_all + (size_t)b * n * 3;
    const float* y = y_all + (size_t)b * y_stride;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    if (!(diameter_given > 0.f)) {
        for (int i = tid; i < n + m; i += 256) {
            const float* p = i < n ? x + 3 * (size_t)i : y + 3 * (size_t)(i - n);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bad |= !(fabsf(p[c]) < INFINITY);
                mn[c] = fminf(mn[c], p[c]);
                mx[c] = fmaxf(mx[c], p[c]);
            }
        }
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (tid == 0) { plan[b] = SkPlan{0.f, -1}; atomicOr(max_eps + 1, 1); }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c][tid] = mn[c]; hi[c][tid] = mx[c]; }
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lo[c][tid] = fminf(lo[c][tid], lo[c][tid + st]);
                hi[c][tid] = fmaxf(hi[c][tid], hi[c][tid + st]);
            }
        __syncthreads();
    }
    if (tid == 0) {
        float diam = diameter_given;
        if (!(diameter_given > 0.f)) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) d2 = __fadd_rn(d2, __fmul_rn(hi[c][0] - lo[c][0], hi[c][0] - lo[c][0]));
            diam = sqrtf(d2);
        }
        SkPlan pl{diam, 0};
        if (diam > 0.f) {
            const double start = 2.0 * log((double)diam), stop = 2.0 * log((double)blur), step = 2.0 * log((double)scaling);
            long long cnt = (long long)ceil((stop - start) / step);   // numpy.arange length
            cnt = cnt < 0 ? 0 : (cnt > 100000 ? 100000 : cnt);
            pl.n_eps = (int)cnt + 2;
        }
        plan[b] = pl;
        atomicMax(max_eps, pl.n_eps);
    }
}

// loss_b = mean(b_x - a_x) + mean(a_y - b_y)   (one workgroup per pair, fixed order)
__global__ void __launch_bounds__(256) sinkhorn_cost_kernel(const float* __restrict__ a_x, const float* __restrict__ b_x, int n,
                                                             const float* __restrict__ a_y, const float* __restrict__ b_y, int m,
                                                             const SkPlan* __restrict__ plan, float* __restrict__ loss) {
    __shared__ double red[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    a_x += (size_t)b * n; b_x += (size_t)b * n; a_y += (size_t)b * m; b_y += (size_t)b * m;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += ((double)b_x[i] - (double)a_x[i]) / n;
    for (int j = tid; j < m; j += 256) s += ((double)a_y[j] - (double)b_y[j]) / m;
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) loss[b] = plan[b].n_eps > 0 ? (float)red[0] : (plan[b].n_eps == 0 ? 0.f : NAN);
}

struct SinkhornWs {
    SkPlan* plan;       // [B]
    int* max_eps;       // 2
    float* pot[2][4];   // ping-pong sets of (a_x[B][n], b_y[B][m], a_y[B][m], b_x[B][n])
    size_t bytes;
};
static SinkhornWs carve_sinkhorn(void* ws, int64_t B, int64_t n, int64_t m) {
    SinkhornWs w;
    Carver c(ws);
    w.plan = c.take<SkPlan>((size_t)B);
    w.max_eps = c.take<int>(4);
    for (int s = 0; s < 2; ++s) {
        w.pot[s][0] = c.take<float>((size_t)(B * n));
        w.pot[s][1] = c.take<float>((size_t)(B * m));
        w.pot[s][2] = c.take<float>((size_t)(B * m));
        w.pot[s][3] = c.take<float>((size_t)(B * n));
    }
    w.bytes = c.used();
    return w;
}

}  // namespace gm

using namespace gm;

extern "C" {

size_t gm_sinkhorn_batched_workspace_bytes(int64_t batch, int64_t n, int64_t m) {
    if (batch < 0 || n < 0 || m < 0) return 0;
    return carve_sinkhorn(nullptr, batch, n, m).bytes;
}
size_t gm_sinkhorn_workspace_bytes(int64_t n, int64_t m) { return gm_sinkhorn_batched_workspace_bytes(1, n, m); }

int gm_sinkhorn_divergence_batched(const float* x, int64_t batch, int64_t n, const float* y, int64_t m, int y_shared, float blur,
                                   float scaling, float diameter, float* loss_device, void* ws, size_t ws_bytes, void* stream) {
    gm::DevGuard dev_guard(x);
    GM_REQUIRE(x && y && loss_device && ws, GM_ERR_INVALID_ARGUMENT, "gm_sinkhorn_divergence: null pointer");
    GM_REQUIRE(batch >= 1 && batch < 65536, GM_ERR_INVALID_ARGUMENT, "gm_sinkhorn_divergence: batch %lld out of range (1 .. 65535)", (long long)batch);
    GM_REQUIRE(n >= 1 && m >= 1 && n < ((int64_t)1 << 30) && m < ((int64_t)1 << 30), GM_ERR_INVALID_ARGUMENT,
               "gm_sinkhorn_divergence: cloud sizes out of range (%lld, %lld)", (long long)n, (long long)m);
    GM_REQUIRE(blur > 0.f && scaling > 0.f && scaling < 1.f, GM_ERR_INVALID_ARGUMENT,
               "gm_sinkhorn_divergence: need blur > 0 and 0 < scaling < 1");
    GM_REQUIRE(!(diameter > 0.f) || isfinite(diameter), GM_ERR_INVALID_ARGUMENT, "gm_sinkhorn_divergence: diameter must be finite");
    SinkhornWs w = carve_sinkhorn(ws, batch, n, m);
    GM_REQUIRE(ws_bytes >= w.bytes, GM_ERR_WORKSPACE, "gm_sinkhorn_divergence: workspace %zu < %zu", ws_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    const size_t y_stride = y_shared ? 0 : (size_t)m * 3;
    // The plan: every pair's diameter (bounding box of its union, geomloss max_diameter -- or the caller's) and the length of its
    // epsilon schedule, on the device.  The host needs ONE number back, the batch's longest schedule = how many launches follow:
    // the only host round trip, once per batch; none when the caller names the diameter (geomloss's `diameter=` keyword).
    GM_HIP_CHECK(hipMemsetAsync(w.max_eps, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(sinkhorn_plan_kernel, dim3((unsigned)batch), dim3(256), 0, s, x, (int)n, y, (int)m, y_stride, blur, scaling,
                       diameter > 0.f ? diameter : 0.f, w.plan, w.max_eps);
    int n_eps = 0;
    if (diameter > 0.f) {   // the same arithmetic as the plan kernel's, on the host
        const double start = 2.0 * log((double)diameter), stop = 2.0 * log((double)blur), step = 2.0 * log((double)scaling);
        long long cnt = (long long)ceil((stop - start) / step);
        cnt = cnt < 0 ? 0 : (cnt > 100000 ? 100000 : cnt);
        n_eps = (int)cnt + 2;
    } else {
        int back[2] = {0, 0};
        GM_HIP_CHECK(hipMemcpyAsync(back, w.max_eps, sizeof(back), hipMemcpyDeviceToHost, s));
        GM_HIP_CHECK(hipStreamSynchronize(s));
        GM_REQUIRE(back[1] == 0, GM_ERR_DATA, "gm_sinkhorn_divergence: non-finite coordinate");
        n_eps = back[0];
    }
    const float logwa = -logf((float)n), logwb = -logf((float)m);
    const dim3 gx((unsigned)cdiv(n, 4 * SK_RB), (unsigned)batch), gy((unsigned)cdiv(m, 4 * SK_RB), (unsigned)batch);
    const size_t xs = (size_t)n * 3;
    // softmin over the second cloud of `f` (supported there), result on the first
    auto softmin = [&](const float* P, int64_t R, size_t ps, dim3 grid, const float* Q, int64_t S, size_t qs, const float* f, float logw, int it,
                       const float* old, float w_old, float w_new, float* out) {
        hipLaunchKernelGGL(softmin_kernel, grid, dim3(256), 0, s, P, (int)R, ps, Q, (int)S, qs, f, logw, w.plan, it, blur, scaling,
                           old, w_old, w_new, out);
    };
    int cur = 0;
    {   // initialisation at the first epsilon
        float** p = w.pot[cur];
        softmin(x, n, xs, gx, x, n, xs, nullptr, logwa, 0, nullptr, 0.f, 1.f, p[0]);             // a_x: OT(a, a)
        softmin(y, m, y_stride, gy, y, m, y_stride, nullptr, logwb, 0, nullptr, 0.f, 1.f, p[1]);   // b_y: OT(b, b)
        softmin(y, m, y_stride, gy, x, n, xs, nullptr, logwa, 0, nullptr, 0.f, 1.f, p[2]);         // a_y
        softmin(x, n, xs, gx, y, m, y_stride, nullptr, logwb, 0, nullptr, 0.f, 1.f, p[3]);         // b_x
    }
    for (int it = 0; it < n_eps; ++it) {  // symmetrised updates, all four from the previous potentials
        float **o = w.pot[cur], **q = w.pot[cur ^ 1];
        softmin(x, n, xs, gx, x, n, xs, o[0], logwa, it, o[0], 0.5f, 0.5f, q[0]);
        softmin(y, m, y_stride, gy, y, m, y_stride, o[1], logwb, it, o[1], 0.5f, 0.5f, q[1]);
        softmin(y, m, y_stride, gy, x, n, xs, o[3], logwa, it, o[2], 0.5f, 0.5f, q[2]);   // a_y from b_x
        softmin(x, n, xs, gx, y, m, y_stride, o[2], logwb, it, o[3], 0.5f, 0.5f, q[3]);   // b_x from a_y
        cur ^= 1;
    }
    {   // last extrapolation at every pair's final epsilon
        float **o = w.pot[cur], **q = w.pot[cur ^ 1];
        softmin(x, n, xs, gx, x, n, xs, o[0], logwa, -1, nullptr, 0.f, 1.f, q[0]);
        softmin(y, m, y_stride, gy, y, m, y_stride, o[1], logwb, -1, nullptr, 0.f, 1.f, q[1]);
        softmin(y, m, y_stride, gy, x, n, xs, o[3], logwa, -1, nullptr, 0.f, 1.f, q[2]);
        softmin(x, n, xs, gx, y, m, y_stride, o[2], logwb, -1, nullptr, 0.f, 1.f, q[3]);
        cur ^= 1;
    }
    float** p = w.pot[cur];
    hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3((unsigned)batch), dim3(256), 0, s, p[0], p[3], (int)n, p[2], p[1], (int)m, w.plan, loss_device);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int gm_sinkhorn_divergence(const float* x, int64_t n, const float* y, int64_t m, float blur, float scaling, float* loss_device,
                           void* ws, size_t ws_bytes, void* stream) {
    return gm_sinkhorn_divergence_batched(x, 1, n, y, m, 1, blur, scaling, 0.f, loss_device, ws, ws_bytes, stream);
}

}  // extern "C"
