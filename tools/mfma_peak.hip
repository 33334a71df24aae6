// Sustained fp32-MFMA ceiling of the device: every SIMD runs back-to-back v_mfma_f32_16x16x4_f32 (and 32x32x2) on
// independent accumulators, nothing else.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) burn(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if (KIND == 0) {
        floatx4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = floatx4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        floatx16 c[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            const int grid = cus * wgs_per_cu, iters = 200000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(grid), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(burn<1>, dim3(grid), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)grid * 4 /*waves*/ * iters * (kind == 0 ? 8 * 2048.0 : 4 * 4096.0);
                if (rep == 2)
                    printf("%s, %d workgroup(s)/CU: %.1f ms, %.1f TFLOP/s (%.3f of 157.3)\n", kind == 0 ? "16x16x4f32" : "32x32x2f32",
                           wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
            }
        }
    return 0;
}
