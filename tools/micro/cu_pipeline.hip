// Prototype measurement for the hidden-256 processor edge MLP (C4): ONE Linear per CU, weights stationary in registers, three CUs of an
// XCD chained into a pipeline that hands 32-row operand images (fp16 hi / lo, 32 KB) on through rings in that XCD's L2 with
// counter flags -- the structure DESIGN.md section 6 / 8 names as the replacement of hm_edge_kernel<256, false>, whose tiles stream
// 786 KB of weights per 128 rows and overlap nothing.  Not a product kernel: no LayerNorm / scatter-add, synthetic data, and the
// result is only checked for having passed through every stage in order.  What it measures: rows per second of such a pipeline with
// the real matrix work (256 x 256, three fp16 partial products per multiply), the real memory streams at both ends (stage 0: e rows
// from HBM + two gathered P rows per edge; stage 2: residual rows + e + e' stores) and the hand-off cost in between.
//
//   workgroup w -> XCD w % 8 (checked against HW_REG_XCC_ID and reported), slot w / 8 of that XCD; slots 3 p + s = stage s of the
//   XCD's pipeline p (10 pipelines per XCD, 2 CUs idle); a pipeline walks `blocks` blocks of 32 rows.
//   ring: RING slots of 32 KB per boundary; counters full[] (blocks produced) / done[] (blocks consumed), written by thread 0 behind a
//   workgroup barrier that follows s_waitcnt vmcnt(0); data and counters move with sc1 (agent scope: served by the XCD's L2).
//   Every spin is bounded and watches a global abort word: a lost hand-off ends the launch with an error instead of hanging.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/cu_pipeline.hip -o tools/micro/cu_pipeline      Run: tools/micro/cu_pipeline [blocks [gather 0|1]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));

constexpr int H = 256, BE = 32, KS = H / 16, THREADS = 512;
constexpr int IMG_B = BE * H * 4;          // fp16 hi + lo of 32 rows x 256 features: [KS][2 parts][64 lanes] x 16 B
constexpr int RING = 4;
constexpr int PIPES_PER_XCD = 10;
constexpr unsigned MAXSPIN = 1u << 18;
constexpr int SC1 = 16;                    // buffer-instruction aux bit: sc1 (agent scope)

struct Args {
    const float* e_in;      // [rows][H]
    float* e_out;           // [rows][H]
    const float* P;         // [n_nodes][2 H]
    const int* dst;         // [rows]
    const int* src;         // [rows]
    const half8* w;         // [3 stages][8 jb][KS][2 parts][64 lanes]
    uintx4* rings;          // [pipes][2 boundaries][RING][IMG_B / 16]
    unsigned* flags;        // [pipes][2 boundaries][2: full, done] (64-byte apart)
    unsigned* abort_word;
    unsigned* xcc_wrong;    // workgroups whose XCC_ID is not w % 8
    unsigned* seq_errors;   // blocks that arrived out of order / stale
    unsigned long long* cyc;  // [pipes][3 stages][4]: total, wait input, wait output slot, blocks
    int blocks;
    int gather;             // 0: stage 0 skips the P gathers (what the hand-off stages alone sustain)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The counter behind `flag` has to reach `want`.  `known` is the value last seen (the neighbour usually runs ahead: no memory access
// then).  Otherwise thread 0 spins (bounded; watches the abort word) and the value reaches every thread through LDS + a barrier.
__device__ __forceinline__ bool wait_count(const unsigned* flag, unsigned want, unsigned& known, unsigned* abort_word, unsigned* s_val,
                                           unsigned long long& waited) {
    if (known >= want) return true;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        unsigned v = 0xffffffffu;
        for (unsigned it = 0; it < MAXSPIN; ++it) {
            const unsigned f = ld_flag(flag);
            if (f >= want) { v = f; break; }
            if ((it & 31) == 31 && ld_flag(abort_word)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (v == 0xffffffffu) st_flag(abort_word, 1u);
        *s_val = v;
        waited += __builtin_readcyclecounter() - t0;
    }
    __syncthreads();
    const unsigned v = *s_val;
    __syncthreads();
    if (v == 0xffffffffu) return false;
    known = v;
    return true;
}

__device__ __forceinline__ void split4(floatx4 v, uintx2& h, uintx2& l) {
    half4 hh, ll;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const _Float16 a = (_Float16)v[t];
        hh[t] = a;
        ll[t] = (_Float16)(v[t] - (float)a);
    }
    h = __builtin_bit_cast(uintx2, hh);
    l = __builtin_bit_cast(uintx2, ll);
}

template <int STAGE>
__device__ __forceinline__ void run_stage(const Args& A, int pipe, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hi = lane >> 5;
    const int jb = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* in_img = smem;                    // 2 x IMG_B
    char* out_img = smem + 2 * IMG_B;       // IMG_B
    unsigned* s_val = reinterpret_cast<unsigned*>(smem + 3 * IMG_B);
    unsigned known_in = 0, known_done = 0;
    // weights of this STAGE's Linear: output block jb, all k-groups, hi / lo: 128 registers
    half8 wh[KS], wl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wh[ks] = A.w[(((size_t)(STAGE * 8 + jb) * KS + ks) * 2 + 0) * 64 + lane];
        wl[ks] = A.w[(((size_t)(STAGE * 8 + jb) * KS + ks) * 2 + 1) * 64 + lane];
    }
    unsigned* fl_in = STAGE > 0 ? A.flags + ((size_t)(pipe * 2 + STAGE - 1) * 2) * 16 : nullptr;      // full, done (+16 words)
    unsigned* fl_out = STAGE < 2 ? A.flags + ((size_t)(pipe * 2 + STAGE) * 2) * 16 : nullptr;
    const uintx4* ring_in = STAGE > 0 ? A.rings + (size_t)(pipe * 2 + STAGE - 1) * RING * (IMG_B / 16) : nullptr;
    uintx4* ring_out = STAGE < 2 ? A.rings + (size_t)(pipe * 2 + STAGE) * RING * (IMG_B / 16) : nullptr;
    const int B = A.blocks;
    const size_t row0 = (size_t)pipe * B * BE;          // this pipeline's rows
    unsigned long long waited_in = 0, waited_out = 0;
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned seq_bad = 0;
    bool alive = true;

    // image of block b -> LDS buffer (b & 1).  Stage 0: fp32 rows from HBM, split here; STAGEs 1 / 2: a ring slot, as it is.
    uintx4 nxt[4];
    floatx4 nxtf[4];
    auto request = [&](int b) {
        if (STAGE == 0) {
            const __amdgpu_buffer_rsrc_t r = srd_of(A.e_in + (row0 + (size_t)b * BE) * H, BE * H * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) nxtf[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16 + j * 8192, 0, 0));
        } else {
            const __amdgpu_buffer_rsrc_t r = srd_of(ring_in + (size_t)(b % RING) * (IMG_B / 16), IMG_B);
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16 + j * 8192, 0, SC1));
        }
    };
    auto deposit = [&](int b) {
        char* img = in_img + (b & 1) * IMG_B;
        if (STAGE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int byte = tid * 16 + j * 8192, row = byte >> 10, col = (byte & 1023) >> 2;   // 4 floats: features col .. col + 3
                const int ks = col >> 4, hi2 = (col >> 3) & 1, half = (col >> 2) & 1;
                uintx2 h, l;
                split4(nxtf[j], h, l);
                *reinterpret_cast<uintx2*>(img + ((ks * 2 + 0) * 64 + row + 32 * hi2) * 16 + half * 8) = h;
                *reinterpret_cast<uintx2*>(img + ((ks * 2 + 1) * 64 + row + 32 * hi2) * 16 + half * 8) = l;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uintx4*>(img + tid * 16 + j * 8192) = nxt[j];
        }
    };

    if (STAGE > 0) alive = wait_count(fl_in, 1u, known_in, A.abort_word, s_val, waited_in);
    if (alive) {
        request(0);
        deposit(0);
        __syncthreads();
    }
    int di = 0, si = 0;   // STAGE 0: destination / source node of this lane's row of the block at hand (loaded a block ahead)
    if (STAGE == 0 && alive) { di = A.dst[row0 + n]; si = A.src[row0 + n]; }
    for (int b = 0; alive && b < B; ++b) {
        // ---- requests at the top, consumed behind the Linear: the next block's image (STAGEs 1 / 2: once the producer has published
        // it), STAGE 0: the factorised first layer's node terms -- two gathered P rows per edge -- and the next block's indices,
        // STAGE 2: the residual rows
        if (b + 1 < B) {
            if (STAGE > 0) alive = wait_count(fl_in, (unsigned)(b + 2), known_in, A.abort_word, s_val, waited_in);
            if (!alive) break;
            request(b + 1);
        }
        floatx4 pa[4], pc[4], e0[4];
        if (STAGE == 0 && !A.gather) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { pa[g] = floatx4{0.f, 0.f, 0.f, 0.f}; pc[g] = pa[g]; }
        } else if (STAGE == 0) {
            const float* pi = A.P + (size_t)di * 2 * H + 32 * jb + 4 * hi;
            const float* pj = A.P + (size_t)si * 2 * H + H + 32 * jb + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) { pa[g] = *reinterpret_cast<const floatx4*>(pi + 8 * g); pc[g] = *reinterpret_cast<const floatx4*>(pj + 8 * g); }
            const size_t rn = row0 + (size_t)(b + 1 < B ? b + 1 : b) * BE + n;
            di = A.dst[rn];
            si = A.src[rn];
        } else if (STAGE == 2) {
            const __amdgpu_buffer_rsrc_t ri = srd_of(A.e_in + (row0 + (size_t)b * BE) * H, BE * H * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) e0[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(ri, tid * 16 + j * 8192, 0, 0));
        }
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.01f * jb;
        // ---- the Linear: 16 k-groups x 3 partial products
        const half8* img = reinterpret_cast<const half8*>(in_img + (b & 1) * IMG_B) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 bh = img[(ks * 2 + 0) * 64], bl = img[(ks * 2 + 1) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, acc, 0, 0, 0);
        }
        if (STAGE == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[4 * g + t] += pa[g][t] + pc[g][t];
        }
        // ---- output: STAGEs 0 / 1: ReLU, split, the wave's two k-groups of the next image (through LDS, so that the ring is written
        // in whole lines); STAGE 2: fp32 rows (+ residual) through LDS, whole lines
        if (STAGE < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(acc[4 * g + t], 0.f) * 0.0625f;
                uintx2 h, l;
                split4(v, h, l);
                const int ks = 2 * jb + (g >> 1), hi2 = g & 1;
                *reinterpret_cast<uintx2*>(out_img + ((ks * 2 + 0) * 64 + n + 32 * hi2) * 16 + hi * 8) = h;
                *reinterpret_cast<uintx2*>(out_img + ((ks * 2 + 1) * 64 + n + 32 * hi2) * 16 + hi * 8) = l;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[4 * g + t];
                *reinterpret_cast<floatx4*>(out_img + n * 1024 + (32 * jb + 8 * g + 4 * hi) * 4) = v;
            }
        }
        __syncthreads();
        if (STAGE < 2) {
            // the slot this block goes to is free once the consumer has taken block b - RING
            if (b >= RING) alive = wait_count(fl_out + 16, (unsigned)(b - RING + 1), known_done, A.abort_word, s_val, waited_out);
            if (!alive) break;
            const __amdgpu_buffer_rsrc_t r = srd_of(ring_out + (size_t)(b % RING) * (IMG_B / 16), IMG_B);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uintx4 o = *reinterpret_cast<const uintx4*>(out_img + tid * 16 + j * 8192);
                if (tid == 0 && j == 0) o[0] = (unsigned)b;   // sequence mark (in place of two hi-part values of row 0: harmless here)
                __builtin_amdgcn_raw_buffer_store_b128(o, r, tid * 16 + j * 8192, 0, SC1);
            }
        } else {
            const __amdgpu_buffer_rsrc_t ro = srd_of(A.e_out + (row0 + (size_t)b * BE) * H, BE * H * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const floatx4 y = *reinterpret_cast<const floatx4*>(out_img + tid * 16 + j * 8192);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, e0[j] + y), ro, tid * 16 + j * 8192, 0, 0);
            }
        }
        // ---- the next image into the other LDS buffer.  Its loads were issued at the top of this iteration, behind the ring stores
        // of block b - 1: memory operations of a wave complete in issue order, so once they are here those stores are complete too
        if (b + 1 < B) {
            if (STAGE > 0 && tid == 0 && nxt[0][0] != (unsigned)(b + 1)) ++seq_bad;
            deposit(b + 1);
        }
        __syncthreads();
        if (tid == 0) {
            if (STAGE < 2 && b > 0) st_flag(fl_out, (unsigned)b);                      // blocks 0 .. b - 1 are in the ring
            if (STAGE > 0 && b + 1 < B) st_flag(fl_in + 16, (unsigned)(b + 2));       // blocks 0 .. b + 1 have been taken out of it
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0 && STAGE < 2 && alive) st_flag(fl_out, (unsigned)B);
    if (tid == 0) {
        unsigned long long* c = A.cyc + (size_t)(pipe * 3 + STAGE) * 4;
        c[0] = __builtin_readcyclecounter() - t_start;
        c[1] = waited_in;
        c[2] = waited_out;
        c[3] = alive ? (unsigned long long)B : 0ull;
        if (seq_bad) atomicAdd(A.seq_errors, seq_bad);
    }
}

__global__ void __launch_bounds__(THREADS, 1) pipe3(Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2] input images, output image / fp32 staging, verdict word
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hi = lane >> 5;
    const int jb = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
    if (tid == 0) {
        const unsigned id = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID, bits 3:0
        if ((int)id != xcd) atomicAdd(A.xcc_wrong, 1u);
    }
    if (slot >= 3 * PIPES_PER_XCD) return;
    const int stage = slot % 3, pipe = xcd * PIPES_PER_XCD + slot / 3;
    if (stage == 0) run_stage<0>(A, pipe, smem);
    else if (stage == 1) run_stage<1>(A, pipe, smem);
    else run_stage<2>(A, pipe, smem);
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 768;
    const int gather = argc > 2 ? atoi(argv[2]) : 1;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    if (cus < 256) { printf("needs 256 CUs (8 XCDs x 32), found %d\n", cus); return 1; }
    const int pipes = 8 * PIPES_PER_XCD;
    const size_t rows = (size_t)pipes * blocks * BE;
    const int n_nodes = 100000;
    srand(7);
    std::vector<float> he(rows * H), hP((size_t)n_nodes * 2 * H);
    for (auto& v : he) v = frand();
    for (auto& v : hP) v = frand() * 0.1f;
    std::vector<int> hd(rows), hs(rows);
    for (size_t r = 0; r < rows; ++r) {   // destination-sorted edges of a radius graph: ~20 per node, sources nearby
        hd[r] = (int)(r * (size_t)n_nodes / rows);
        hs[r] = (hd[r] + rand() % 2000 - 1000 + n_nodes) % n_nodes;
    }
    std::vector<_Float16> hw((size_t)3 * 8 * KS * 2 * 64 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(frand() * (((i / 512) & 1) ? 3e-5f : 0.06f));
    Args A{};
    float *e_in, *e_out, *P;
    int *dst, *src;
    half8* w;
    uintx4* rings;
    unsigned *flags, *misc;
    unsigned long long* cyc;
    hipMalloc(&e_in, rows * H * 4);
    hipMalloc(&e_out, rows * H * 4);
    hipMalloc(&P, hP.size() * 4);
    hipMalloc(&dst, rows * 4);
    hipMalloc(&src, rows * 4);
    hipMalloc(&w, hw.size() * 2);
    hipMalloc(&rings, (size_t)pipes * 2 * RING * IMG_B);
    hipMalloc(&flags, (size_t)pipes * 2 * 2 * 16 * 4);
    hipMalloc(&misc, 64);
    hipMalloc(&cyc, (size_t)pipes * 3 * 4 * 8);
    hipMemcpy(e_in, he.data(), rows * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(P, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dst, hd.data(), rows * 4, hipMemcpyHostToDevice);
    hipMemcpy(src, hs.data(), rows * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    A.e_in = e_in; A.e_out = e_out; A.P = P; A.dst = dst; A.src = src; A.w = w; A.rings = rings; A.flags = flags;
    A.abort_word = misc; A.xcc_wrong = misc + 1; A.seq_errors = misc + 2; A.cyc = cyc; A.blocks = blocks; A.gather = gather;
    const size_t lds = 3 * IMG_B + 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pipe3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f, sum = 0.f;
    const int reps = 12;
    unsigned hm[3] = {0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        hipMemsetAsync(flags, 0, (size_t)pipes * 2 * 2 * 16 * 4, 0);
        hipMemsetAsync(misc, 0, 64, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(pipe3, dim3(256), dim3(THREADS), lds, 0, A);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hm, misc, 12, hipMemcpyDeviceToHost);
        if (hm[0]) { printf("rep %d: a hand-off was lost (abort word set): XCC mismatches %u, sequence errors %u\n", rep, hm[1], hm[2]); return 2; }
        if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    std::vector<unsigned long long> hc((size_t)pipes * 3 * 4);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    printf("%d pipelines x 3 CUs, %d blocks of 32 rows each = %zu rows, hidden 256, three fp16 partial products%s\n", pipes, blocks, rows,
           gather ? "" : "; stage 0 WITHOUT its P gathers");
    printf("workgroups not on XCD (id %% 8): %u; blocks that arrived out of order: %u\n", hm[1], hm[2]);
    const double avg = sum / (reps - 2);
    printf("launch: %.3f ms average, %.3f ms best  ->  %.2f us per block and pipeline, %.1f M rows/s\n", avg, best, avg * 1e3 / blocks, rows / avg / 1e3);
    printf("the same rows through hm_edge_kernel<256,false> (profiles/r05_c4_kernel_stats.csv, E = 1.97 M): 3.27 ms\n");
    for (int s = 0; s < 3; ++s) {
        double tot = 0, wi = 0, wo = 0;
        for (int p = 0; p < pipes; ++p) {
            tot += (double)hc[(size_t)(p * 3 + s) * 4 + 0];
            wi += (double)hc[(size_t)(p * 3 + s) * 4 + 1];
            wo += (double)hc[(size_t)(p * 3 + s) * 4 + 2];
        }
        printf("stage %d: %.0f shader-clock counts per block; waiting for its input %.1f %%, for a free output slot %.1f %%\n", s, tot / pipes / blocks,
               100.0 * wi / tot, 100.0 * wo / tot);
    }
    const double flop = (double)rows * 3 * 2.0 * H * H * 3;
    printf("matrix work: %.1f TFLOP/s of issued fp16 MFMA on 240 CUs (dense peak 2500 on 256)\n", flop / avg / 1e9);
    return 0;
}
