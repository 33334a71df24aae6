// Microbenchmark: issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950, alone and beside an MFMA-issuing wave on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: scalar fma x 2N, 1: pk fma x N ; waves >= 4 of a block run MFMAs when mfma_waves
__global__ void __launch_bounds__(512) k(float* out, int iters, int mfma_waves) {
    const int wave = threadIdx.x >> 6;
    if (mfma_waves && wave >= 4) {
        floatx16 acc = {};
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0];
        return;
    }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.01f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(c1), "v"(c2));
        } else {
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                float2v v = {x[u], x[u + 1]};
                float2v cc1 = {c1, c1}, cc2 = {c2, c2};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(cc1), "v"(cc2));
                x[u] = v[0]; x[u + 1] = v[1];
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int iters, int mf) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, mf);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, mf);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    for (int mf = 0; mf < 2; ++mf) {
        float t0 = run<0>(d, iters, mf), t1 = run<1>(d, iters, mf);
        // waves 0..3 (or all 8 without MFMA waves: two VALU waves per SIMD) each issue 16 scalar / 8 packed fmas per iteration
        printf("mfma_waves=%d: 16 x v_fma_f32 per iter %.3f ms, 8 x v_pk_fma_f32 per iter %.3f ms  (same flops)\n", mf, t0, t1);
    }
    return 0;
}
