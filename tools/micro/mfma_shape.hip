// MFMA shape against clock: v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 at the SAME output tile per wave (32 output
// features x 32 rows, K = 128, three partial products per k-group as in the fp16-split kernels: lo*hi, hi*lo, hi*hi), the weights
// (A operand) stationary in 64 registers, every B fragment re-read from a 16 KB LDS image by ds_read_b128, random data.
// Per variant: >= 2 s of back-to-back launches, then the wall time of a launch (HIP events), the wave's own cycle count
// (s_memtime) and the clock it ran at (delta s_memtime / delta s_memrealtime x 100 MHz), median over workgroups
// (MI355X_MICROARCH.md, DVFS give-back items 6 and 7).
//   WAVES = 4: one wave per SIMD, 12: three per SIMD (the systolic kernels' occupancy)
//   FILL  = independent v_fma_f32 per 32x32x16 MFMA (per TWO 16x16x32 MFMAs): the vector work a kernel hides in the gaps
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define SB __builtin_amdgcn_sched_barrier(0)

template <int SHAPE, int FILL>
__global__ void __launch_bounds__(768, 1) burn(const half8* __restrict__ w, const half8* __restrict__ img, float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += blockDim.x) reinterpret_cast<half8*>(smem)[i] = img[i];
    half8 wh[8], wl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        wh[k] = w[((wave * 8 + k) * 2 + 0) * 64 + lane];
        wl[k] = w[((wave * 8 + k) * 2 + 1) * 64 + lane];
    }
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + lane * 1e-3f + i;
    __syncthreads();
    const unsigned a0 = lane * 16;
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float s = 0.f;
    if (SHAPE == 32) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const half8 bh = *reinterpret_cast<const half8*>(smem + a0 + (ks * 2 + 0) * 1024);
                const half8 bl = *reinterpret_cast<const half8*>(smem + a0 + (ks * 2 + 1) * 1024);
                SB;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, acc, 0, 0, 0);
                SB;
#pragma unroll
                for (int i = 0; i < FILL; ++i) f[i & 7] = fmaf(f[i & 7], 1.0000001f, 1e-7f);
                SB;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, acc, 0, 0, 0);
                SB;
#pragma unroll
                for (int i = 0; i < FILL; ++i) f[(i + 3) & 7] = fmaf(f[(i + 3) & 7], 1.0000001f, 1e-7f);
                SB;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, acc, 0, 0, 0);
                SB;
#pragma unroll
                for (int i = 0; i < FILL; ++i) f[(i + 5) & 7] = fmaf(f[(i + 5) & 7], 1.0000001f, 1e-7f);
                SB;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else {
        // the same 32 x 32 output tile as 2 feature tiles x 2 row tiles of 16 x 16; a k-group is K = 32: the A fragment of
        // (feature tile ft, k-group kg) is register set wh[2 kg + ft], the B fragment of (row tile rt, kg) 1 KiB of the image
        floatx4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                half8 bh[2], bl[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    bh[rt] = *reinterpret_cast<const half8*>(smem + a0 + ((kg * 2 + rt) * 2 + 0) * 1024);
                    bl[rt] = *reinterpret_cast<const half8*>(smem + a0 + ((kg * 2 + rt) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            SB;
                            acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part == 0 ? wl[2 * kg + ft] : wh[2 * kg + ft], part == 1 ? bl[rt] : bh[rt], acc[ft][rt], 0, 0, 0);
                            SB;
#pragma unroll
                            for (int i = 0; i < FILL / 2; ++i) f[(i + 2 * (ft * 2 + rt)) & 7] = fmaf(f[(i + 2 * (ft * 2 + rt)) & 7], 1.0000001f, 1e-7f);
                            SB;
                        }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    }
    if (lane == 0) {
        const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[(blockIdx.x * 12 + wave) * 2 + 0] = t1 - t0;
        stamps[(blockIdx.x * 12 + wave) * 2 + 1] = r1 - r0;
    }
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 768 + tid] = s;
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <int SHAPE, int FILL>
static void run(int waves, const half8* w, const half8* img, float* out, unsigned long long* stamps, int cus, const char* name) {
    const int iters = 600 / (waves / 4);   // about 1 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL((burn<SHAPE, FILL>), dim3(cus), dim3(waves * 64), 16384, 0, w, img, out, stamps, iters); };
    // >= 2 s of back-to-back launches before anything is read
    hipEventRecord(e0);
    float ms = 0.f;
    int n = 0;
    do {
        for (int i = 0; i < 100; ++i) launch();
        n += 100;
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 2000.f);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 200;
    std::vector<unsigned long long> st((size_t)cus * 12 * 2);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int b = 0; b < cus; ++b)
        for (int wv = 0; wv < waves; ++wv) {
            const double c = (double)st[(b * 12 + wv) * 2], r = (double)st[(b * 12 + wv) * 2 + 1];
            cyc.push_back(c);
            if (r > 0) clk.push_back(c / r * 0.1);   // GHz: s_memrealtime ticks at 100 MHz
        }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double flop = (double)cus * waves * iters * 24.0 * 2.0 * 32 * 32 * 16;
    const double mfma32 = (double)iters * 24 * (waves / 4);   // 32x32x16-equivalent MFMAs per SIMD
    printf("%-28s %2d waves/CU  fill %d: %.4f ms/launch  %.1f TFLOP/s  wave cycles (median) %.0f = %.1f per 32x32x16-equivalent MFMA of its SIMD  clock %.3f GHz\n",
           name, waves, FILL, ms, flop / ms / 1e9, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / mfma32, clk.empty() ? 0.0 : clk[clk.size() / 2]);
    fflush(stdout);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    srand(1234);
    std::vector<_Float16> hw((size_t)12 * 8 * 2 * 64 * 8), himg(1024 * 8);
    for (size_t i = 0; i < hw.size(); ++i) {
        const bool lo = (i / 512) & 1;     // [wave][k][part][lane][8]
        hw[i] = (_Float16)(frand() * (lo ? 4.8e-4f : 1.0f));
    }
    for (size_t i = 0; i < himg.size(); ++i) {
        const bool lo = (i / 512) & 1;
        himg[i] = (_Float16)(frand() * (lo ? 4.8e-4f : 1.0f) * 16.f);
    }
    half8 *w, *img;
    float* out;
    unsigned long long* stamps;
    hipMalloc(&w, hw.size() * 2);
    hipMalloc(&img, himg.size() * 2);
    hipMalloc(&out, (size_t)cus * 768 * 4);
    hipMalloc(&stamps, (size_t)cus * 12 * 2 * 8);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {   // interleaved rounds (rule 24)
        run<32, 0>(4, w, img, out, stamps, cus, "v_mfma_f32_32x32x16_f16");
        run<16, 0>(4, w, img, out, stamps, cus, "v_mfma_f32_16x16x32_f16");
        run<32, 0>(12, w, img, out, stamps, cus, "v_mfma_f32_32x32x16_f16");
        run<16, 0>(12, w, img, out, stamps, cus, "v_mfma_f32_16x16x32_f16");
        run<32, 4>(12, w, img, out, stamps, cus, "v_mfma_f32_32x32x16_f16");
        run<16, 4>(12, w, img, out, stamps, cus, "v_mfma_f32_16x16x32_f16");
    }
    return 0;
}
