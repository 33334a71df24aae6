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

// out[i] = w_old * old[i] + w_new * ( -eps * log sum_j exp(logw + f[j] * inv_eps - |p_i - q_j|^2 * 0.5 * inv_eps) )
__global__ void __launch_bounds__(256) softmin_kernel(const float* __restrict__ P, int R, const float* __restrict__ Q, int S,
                                                       const float* __restrict__ f, float inv_eps, float logw, float eps,
                                                       const float* __restrict__ old, float w_old, float w_new, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * SK_RB;
    if (r0 >= R) return;
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

// mins / maxs of the union of two clouds -> out[0..2] = min xyz, out[3..5] = max xyz (one workgroup)
__global__ void __launch_bounds__(256) cloud_bounds_kernel(const float* __restrict__ x, int n, const float* __restrict__ y, int m,
                                                            float* __restrict__ out) {
    __shared__ float lo[3][256], hi[3][256];
    const int tid = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n + m; i += 256) {
        const float* p = i < n ? x + 3 * (size_t)i : y + 3 * (size_t)(i - n);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mn[c] = fminf(mn[c], p[c]);
            mx[c] = fmaxf(mx[c], p[c]);
        }
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
    if (tid < 3) { out[tid] = lo[tid][0]; out[3 + tid] = hi[tid][0]; }
}

// loss = mean(b_x - a_x) + mean(a_y - b_y)   (one workgroup, fixed order)
__global__ void __launch_bounds__(256) sinkhorn_cost_kernel(const float* __restrict__ a_x, const float* __restrict__ b_x, int n,
                                                             const float* __restrict__ a_y, const float* __restrict__ b_y, int m,
                                                             float* __restrict__ loss) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += ((double)b_x[i] - (double)a_x[i]) / n;
    for (int j = tid; j < m; j += 256) s += ((double)a_y[j] - (double)b_y[j]) / m;
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) *loss = (float)red[0];
}

struct SinkhornWs {
    float* bounds;      // 6
    float* pot[2][4];   // ping-pong sets of (a_x[n], b_y[m], a_y[m], b_x[n])
    size_t bytes;
};
static SinkhornWs carve_sinkhorn(void* ws, int64_t n, int64_t m) {
    SinkhornWs w;
    Carver c(ws);
    w.bounds = c.take<float>(8);
    for (int s = 0; s < 2; ++s) {
        w.pot[s][0] = c.take<float>((size_t)n);
        w.pot[s][1] = c.take<float>((size_t)m);
        w.pot[s][2] = c.take<float>((size_t)m);
        w.pot[s][3] = c.take<float>((size_t)n);
    }
    w.bytes = c.used();
    return w;
}

}  // namespace gm

using namespace gm;

extern "C" {

size_t gm_sinkhorn_workspace_bytes(int64_t n, int64_t m) {
    if (n < 0 || m < 0) return 0;
    return carve_sinkhorn(nullptr, n, m).bytes;
}

int gm_sinkhorn_divergence(const float* x, int64_t n, const float* y, int64_t m, float blur, float scaling, float* loss_device,
                           void* ws, size_t ws_bytes, void* stream) {
    gm::DevGuard dev_guard(x);
    GM_REQUIRE(x && y && loss_device && ws, GM_ERR_INVALID_ARGUMENT, "gm_sinkhorn_divergence: null pointer");
    GM_REQUIRE(n >= 1 && m >= 1 && n < ((int64_t)1 << 30) && m < ((int64_t)1 << 30), GM_ERR_INVALID_ARGUMENT,
               "gm_sinkhorn_divergence: cloud sizes out of range (%lld, %lld)", (long long)n, (long long)m);
    GM_REQUIRE(blur > 0.f && scaling > 0.f && scaling < 1.f, GM_ERR_INVALID_ARGUMENT,
               "gm_sinkhorn_divergence: need blur > 0 and 0 < scaling < 1");
    SinkhornWs w = carve_sinkhorn(ws, n, m);
    GM_REQUIRE(ws_bytes >= w.bytes, GM_ERR_WORKSPACE, "gm_sinkhorn_divergence: workspace %zu < %zu", ws_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    // diameter of the union's bounding box (geomloss max_diameter): the only host round trip
    hipLaunchKernelGGL(cloud_bounds_kernel, dim3(1), dim3(256), 0, s, x, (int)n, y, (int)m, w.bounds);
    float bb[6];
    GM_HIP_CHECK(hipMemcpyAsync(bb, w.bounds, sizeof(bb), hipMemcpyDeviceToHost, s));
    GM_HIP_CHECK(hipStreamSynchronize(s));
    for (int c = 0; c < 6; ++c) GM_REQUIRE(isfinite(bb[c]), GM_ERR_DATA, "gm_sinkhorn_divergence: non-finite coordinate");
    float d2 = 0.f;
    for (int c = 0; c < 3; ++c) d2 += (bb[3 + c] - bb[c]) * (bb[3 + c] - bb[c]);
    const double diameter = (double)sqrtf(d2);
    if (diameter == 0.0) {  // both clouds are the same single point
        GM_HIP_CHECK(hipMemsetAsync(loss_device, 0, sizeof(float), s));
        return GM_OK;
    }
    // epsilon schedule (p = 2): [diameter^2] + exp(arange(2 log diameter, 2 log blur, 2 log scaling)) + [blur^2]
    std::vector<double> eps_s;
    eps_s.push_back(diameter * diameter);
    if (diameter > 0.0) {
        const double start = 2.0 * log(diameter), stop = 2.0 * log((double)blur), step = 2.0 * log((double)scaling);
        const int64_t cnt = (int64_t)ceil((stop - start) / step);  // numpy.arange length
        for (int64_t i = 0; i < cnt; ++i) eps_s.push_back(exp(start + (double)i * step));
    }
    eps_s.push_back((double)blur * (double)blur);

    const float logwa = -logf((float)n), logwb = -logf((float)m);
    const unsigned gx = (unsigned)cdiv(n, 4 * SK_RB), gy = (unsigned)cdiv(m, 4 * SK_RB);
    // softmin over the second cloud of `f` (supported there), result on the first
    auto softmin = [&](const float* P, int64_t R, unsigned grid, const float* Q, int64_t S, const float* f, float logw, double eps,
                       const float* old, float w_old, float w_new, float* out) {
        hipLaunchKernelGGL(softmin_kernel, dim3(grid), dim3(256), 0, s, P, (int)R, Q, (int)S, f, (float)(1.0 / eps), logw, (float)eps,
                           old, w_old, w_new, out);
    };
    int cur = 0;
    float **p = w.pot[cur];
    {   // initialisation at the first epsilon
        const double e0 = eps_s[0];
        softmin(x, n, gx, x, n, nullptr, logwa, e0, nullptr, 0.f, 1.f, p[0]);   // a_x: OT(a, a)
        softmin(y, m, gy, y, m, nullptr, logwb, e0, nullptr, 0.f, 1.f, p[1]);   // b_y: OT(b, b)
        softmin(y, m, gy, x, n, nullptr, logwa, e0, nullptr, 0.f, 1.f, p[2]);   // a_y
        softmin(x, n, gx, y, m, nullptr, logwb, e0, nullptr, 0.f, 1.f, p[3]);   // b_x
    }
    for (size_t it = 0; it < eps_s.size(); ++it) {  // symmetrised updates, all four from the previous potentials
        const double e = eps_s[it];
        float **o = w.pot[cur], **q = w.pot[cur ^ 1];
        softmin(x, n, gx, x, n, o[0], logwa, e, o[0], 0.5f, 0.5f, q[0]);
        softmin(y, m, gy, y, m, o[1], logwb, e, o[1], 0.5f, 0.5f, q[1]);
        softmin(y, m, gy, x, n, o[3], logwa, e, o[2], 0.5f, 0.5f, q[2]);   // a_y from b_x
        softmin(x, n, gx, y, m, o[2], logwb, e, o[3], 0.5f, 0.5f, q[3]);   // b_x from a_y
        cur ^= 1;
    }
    {   // last extrapolation at the final epsilon
        const double e = eps_s.back();
        float **o = w.pot[cur], **q = w.pot[cur ^ 1];
        softmin(x, n, gx, x, n, o[0], logwa, e, nullptr, 0.f, 1.f, q[0]);
        softmin(y, m, gy, y, m, o[1], logwb, e, nullptr, 0.f, 1.f, q[1]);
        softmin(y, m, gy, x, n, o[3], logwa, e, nullptr, 0.f, 1.f, q[2]);
        softmin(x, n, gx, y, m, o[2], logwb, e, nullptr, 0.f, 1.f, q[3]);
        cur ^= 1;
    }
    p = w.pot[cur];
    hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3(1), dim3(256), 0, s, p[0], p[3], (int)n, p[2], p[1], (int)m, loss_device);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // extern "C"
