// Microbenchmark: a 128 -> 128 Linear + ReLU applied repeatedly to 32 rows per wave, weights resident in LDS, as
//   (a) fp32 MFMAs (v_mfma_f32_32x32x2_f32, 256 per layer and wave), and
//   (b) six bf16 x bf16 products of three-way bf16 splits (v_mfma_f32_32x32x16_bf16, 192 per layer and wave,
//       activations re-split in registers after every layer) -- the "exact fp32 on the bf16 pipe" scheme of
//       tools/bf16_split_study.py.
// Prints the fp32-equivalent TFLOP/s of both.  One workgroup of 8 waves per CU (two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x6_chain.hip -o tools/bf16x6_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int H = 128;

__device__ __forceinline__ void split8(const floatx16& v, int s, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        floatx2 x = {v[8 * s + e], v[8 * s + e + 1]};
        bf16x2 h = __builtin_convertvector(x, bf16x2);
        floatx2 r1 = x - __builtin_convertvector(h, floatx2);
        bf16x2 m = __builtin_convertvector(r1, bf16x2);
        floatx2 r2 = r1 - __builtin_convertvector(m, floatx2);
        bf16x2 l = __builtin_convertvector(r2, bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        mid[e] = m[0]; mid[e + 1] = m[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}

// weights: piece(p, ks, jb) = 64 lanes x 8 bf16; value irrelevant for timing (filled with small numbers)
__global__ void __launch_bounds__(512, 1) chain_bf16x6(const bf16x8* __restrict__ wsrc, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8* W = reinterpret_cast<bf16x8*>(smem);  // [3][8][4][64]
    for (int i = threadIdx.x; i < 3 * 8 * 4 * 64; i += 512) W[i] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    floatx16 acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jb][r] = 0.001f * (lane + r + jb);
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        asm volatile("" ::: "memory");  // the weights are re-read every layer, as a streaming kernel would
        bf16x8 bh[8], bm[8], bl[8];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                floatx16 a = acc[jb];
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
                split8(a, s, bh[2 * jb + s], bm[2 * jb + s], bl[2 * jb + s]);
            }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jb][r] = 0.01f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const bf16x8 ah = W[((0 * 8 + ks) * 4 + jb) * 64 + lane];
                const bf16x8 am = W[((1 * 8 + ks) * 4 + jb) * 64 + lane];
                const bf16x8 al = W[((2 * 8 + ks) * 4 + jb) * 64 + lane];
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks], acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks], acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks], acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc[jb], 0, 0, 0);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) s += acc[jb][0] + acc[jb][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(512, 1) chain_f32(const floatx4* __restrict__ wsrc, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    floatx4* W = reinterpret_cast<floatx4*>(smem);  // [16 kq][4 jb][64 lanes] x 4 k-steps
    for (int i = threadIdx.x; i < 16 * 4 * 64; i += 512) W[i] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    floatx16 acc[4], act[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jb][r] = 0.001f * (lane + r + jb);
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                act[jb][r] = fmaxf(acc[jb][r], 0.f);
                acc[jb][r] = 0.01f;
            }
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const floatx4 a = W[(kq * 4 + jb) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], act[kq >> 2][(kq & 3) * 4 + t], acc[jb], 0, 0, 0);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) s += acc[jb][0] + acc[jb][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t wb = 3 * 8 * 4 * 64 * 16, wf = 16 * 4 * 64 * 16;
    std::vector<unsigned short> hb(wb / 2, 0x3c00);  // small bf16 values
    std::vector<float> hf(wf / 4, 0.001f);
    void *db, *df;
    float* out;
    hipMalloc(&db, wb); hipMalloc(&df, wf); hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMemcpy(db, hb.data(), wb, hipMemcpyHostToDevice);
    hipMemcpy(df, hf.data(), wf, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)chain_bf16x6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wb);
    hipFuncSetAttribute((const void*)chain_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wf);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int layers = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(chain_f32, dim3(cus), dim3(512), wf, 0, (const floatx4*)df, out, layers);
            else hipLaunchKernelGGL(chain_bf16x6, dim3(cus), dim3(512), wb, 0, (const bf16x8*)db, out, layers);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)cus * 8 * layers * 2.0 * H * H * 32;
            if (rep == 2) printf("%s: %.1f ms, %.1f fp32-equivalent TFLOP/s\n", kind == 0 ? "fp32 32x32x2 chain" : "bf16x6 32x32x16 chain", ms, flops / ms / 1e9);
        }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
    return 0;
}
