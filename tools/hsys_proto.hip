// Prototype / microbenchmark of the systolic fp16x3 processor edge kernel (H = 128, 3 Linears), version 2.
//
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/hsys_proto tools/hsys_proto.hip && tools/hsys_proto [N] [local]
//
// 12 waves per workgroup (one workgroup per CU, three waves per SIMD).  Wave (role, jb): role = Linear 1 / 2 / 3 of
// phi_e, jb = 32-feature output block; a wave keeps only ITS Linear's 32 weight rows in registers (two fp16 parts,
// 64 VGPRs).  Work advances in ticks of one 32-edge block, one workgroup barrier per tick, every buffer double-buffered
// by block parity:
//   role 0: e of block x+1 -> operand image E (split into fp16 hi / lo), Linear 1 of block x on P_i[dst] + P_j[src],
//           image X1; requests the rows of blocks x+2 / x+1;
//   role 1: Linear 2 of block x-1 (X1 -> X2); LayerNorm + e_out = e + e' of block x-3;
//   role 2: Linear 3 of block x-2 (X2 -> Z, LayerNorm partial statistics); aggregation of block x-3 (segmented DPP
//           scan over the destination-sorted edges, carry in registers, one row store per segment).
// Global rows are moved in whole 128-byte lines (8 lanes per line); the register <-> MFMA-fragment re-layouts go
// through swizzled LDS images.  No scale multiplies at run time: the power-of-two weight scales are folded into the
// packed weights, biases and P, and leave through the LayerNorm statistics.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#ifndef DEBUG_DUMP
#define DEBUG_DUMP 0
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 128;
constexpr int BE = 32;    // edges per block
constexpr int CB = 32;    // blocks per chunk (alignment unit of the aggregation: carries reset, atomics at its ends)
constexpr int SYS_THREADS = 768;

struct SysArgs {
    const int* dst; const int* src;
    const float* P;      // [N][2H]  (P_i | P_j) * T1
    const float* e_in; float* e_out; float* agg;
    const half8* wimg;   // [3 layers][4 jb][8 ks][2 parts][64 lanes]
    const float* vec;    // [b2*T2 | b3*T3 | gamma | beta]  (4 x 128)
    const int2* blk;     // [nblk] (first edge, count | flags << 8); flags: 1 first block of a chunk, 2 last
    const int* chunk_first;  // [nchunks + 1]
    int nchunks;
    int E;
    float inv_T, eps;    // 1 / (t1 t2 t3): scale of the Linear-3 accumulators
    int residual;
    unsigned long long* stamps;
};

#ifdef STAMPS
#define STAMP(k) do { if (lane == 0 && blockIdx.x == 7 && t >= 64 && t < 192) A.stamps[(((size_t)(t - 64)) * 12 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k)
#endif
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { unsigned u; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b)); return u; }
__device__ __forceinline__ float sub_lo(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__device__ __forceinline__ float sub_hi(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
// compiler-visible on purpose: it is the first reader of MFMA results, and hipcc pads the MFMA -> VALU hazard only for
// instructions it knows (an inline-asm reader would see stale accumulators)
__device__ __forceinline__ float relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 65504.f); }

// two-way fp16 split of 4 floats: hi / lo as two dwords each (elements in order)
__device__ __forceinline__ void split4(float a, float b, float c, float d, uintx2& hi, uintx2& lo) {
    hi[0] = cvt_pk(a, b);
    hi[1] = cvt_pk(c, d);
    lo[0] = cvt_pk(sub_lo(hi[0], a), sub_hi(hi[0], b));
    lo[1] = cvt_pk(sub_lo(hi[1], c), sub_hi(hi[1], d));
}

// accumulator registers 8q..8q+7 of a wave's 32-feature block are the elements of B fragment ks = 2 jb + q (same lane)
__device__ __forceinline__ void acc_to_image(const floatx16& a, uintx4* img, int jb, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu(a[8 * q + j]);
        uintx2 h0, l0, h1, l1;
        split4(v[0], v[1], v[2], v[3], h0, l0);
        split4(v[4], v[5], v[6], v[7], h1, l1);
        img[((2 * jb + q) * 2 + 0) * 64 + lane] = uintx4{h0[0], h0[1], h1[0], h1[1]};
        img[((2 * jb + q) * 2 + 1) * 64 + lane] = uintx4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// slot of lane (n, kg) inside a fragment of the row-written image E (bank-conflict-free for the 8-byte row-major writes)
__device__ __forceinline__ int eslot(int n, int kg, int ksbit) { return (n ^ (2 * (ksbit + 2 * kg))) + 32 * kg; }

#ifdef EXP_NOSB
#define SB
#else
#define SB __builtin_amdgcn_sched_barrier(0)
#endif
// One Linear for this wave's 32 output features: 8 k-groups x 3 MFMAs (lo*hi, hi*lo, hi*hi).  side(slot), slot = 0..23, is
// executed after each MFMA with the instruction order PINNED (sched_barrier): the wave's MFMAs form one dependent chain,
// so whatever independent work sits between two of them runs in the shadow of the first.  B fragments are fetched one
// k-group ahead.
template <bool SWZ, class F>
__device__ __forceinline__ void mlp_layer(floatx16& acc, const half8 (&wh)[8], const half8 (&wl)[8], const half8* img, int lane, F&& side) {
    const int s0 = SWZ ? eslot(lane & 31, lane >> 5, 0) : lane;
    const int s1 = SWZ ? eslot(lane & 31, lane >> 5, 1) : lane;
    half8 bh = img[0 * 64 + s0], bl = img[1 * 64 + s0];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        half8 nh = bh, nl = bl;
        SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, acc, 0, 0, 0);
        SB;
        if (ks + 1 < 8) {
            nh = img[((ks + 1) * 2 + 0) * 64 + (((ks + 1) & 1) ? s1 : s0)];
            nl = img[((ks + 1) * 2 + 1) * 64 + (((ks + 1) & 1) ? s1 : s0)];
        }
        side(3 * ks);
        SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, acc, 0, 0, 0);
        SB;
        side(3 * ks + 1);
        SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, acc, 0, 0, 0);
        SB;
        side(3 * ks + 2);
        bh = nh;
        bl = nl;
    }
    SB;
}

#define DPP_FMAC(x, f, ctrl) asm volatile("v_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(x) : "v"(f))
#define DPP_FMAC_NOP(x, f, ctrl) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(x) : "v"(f))

// LayerNorm of an edge from the eight 16-feature partials (mean_p, M2_p) of the SCALED accumulators (one per wave and
// lane half): parallel-variance merge; returns k, m with  x_hat = acc * k + m   (k = rstd / T, m = -mean_acc * k)
__device__ __forceinline__ void ln_merge(const float* st, int n, float inv_T, float eps, float& k, float& m) {
    float mw[8], m2 = 0.f, mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float2v s = *reinterpret_cast<const float2v*>(st + (w * BE + n) * 2);
        mw[w] = s[0];
        m2 += s[1];
        mean += s[0];
    }
    mean *= 0.125f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { const float d = mw[w] - mean; m2 = fmaf(16.0f * d, d, m2); }
    const float var = m2 * (1.0f / 128.0f) * inv_T * inv_T;
    k = inv_T / sqrtf(var + eps);
    m = -mean * k;
}

// row-major tile addressing: 32 rows x 8 quads (16 bytes), quad index XORed with the row so that both the row-major
// and the accumulator-layout accesses are conflict-free.  Returns the float4 index inside a 4 KiB tile.
__device__ __forceinline__ int tile_q(int row, int quad) { return row * 8 + (quad ^ (row & 7)); }

__global__ void __launch_bounds__(SYS_THREADS, 1) sys_edge_kernel(SysArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* Eimg = reinterpret_cast<half8*>(smem);        // [2][8][2][64]   2 x 16 KiB, eslot() order
    half8* X1 = Eimg + 2 * 1024;
    half8* X2 = X1 + 2 * 1024;
    floatx4* Z = reinterpret_cast<floatx4*>(X2 + 2 * 1024);  // [2][4 jb][256] float4 (tile_q order)   2 x 16 KiB
    floatx4* PS = Z + 2 * 1024;                               // [4 jb][256] float4: role-0 staging of P_i + P_j
    float* ST = reinterpret_cast<float*>(PS + 1024);          // [2][4 jb][2 halves][32][2]
    float* KM = ST + 2 * 8 * BE * 2;                          // [4 jb][32][2]: role-1 merged statistics
    float* vecs = KM + 4 * BE * 2;                            // 4 x 128
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int lane = lane0, n = lane & 31, hi = lane >> 5;
    const int rr = lane >> 3, cq = lane & 7;             // row-major mapping: row 8 j + rr, quad cq
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, jb = wave & 3;
    const int E = A.E;

    const int c0 = (int)((long long)blockIdx.x * A.nchunks / gridDim.x);
    const int c1 = (int)((long long)(blockIdx.x + 1) * A.nchunks / gridDim.x);
    const int b0 = A.chunk_first[c0], b1 = A.chunk_first[c1];
    const int nb = b1 - b0;
    if (nb <= 0) return;

    half8 wh[8], wl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        wh[ks] = A.wimg[(((role * 4 + jb) * 8 + ks) * 2 + 0) * 64 + lane];
        wl[ks] = A.wimg[(((role * 4 + jb) * 8 + ks) * 2 + 1) * 64 + lane];
    }
    // every buffer starts finite: pipeline fill / drain ticks run on them and must not produce NaN (0 * NaN would leak
    // through the flag-multiplied scan)
    for (int i = tid; i < (int)((reinterpret_cast<char*>(vecs) - smem) / 16); i += SYS_THREADS) reinterpret_cast<uintx4*>(smem)[i] = uintx4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 4 * H; i += SYS_THREADS) vecs[i] = A.vec[i];
    __syncthreads();
    auto ok = [&](int x) { return x >= b0 && x < b1; };
    auto nothing = [](int) {};

    if (role == 0) {
        // ------------------------------------------------------------------ role 0
        floatx4 pi[4], pj[4];   // row-major quads of rows 8 j + rr: P of block x
#pragma unroll
        for (int j = 0; j < 4; ++j) { pi[j] = floatx4{0.f, 0.f, 0.f, 0.f}; pj[j] = pi[j]; }
        floatx16 acc;
        int dl1 = 0, sl1 = 0, dl2 = 0, sl2 = 0;   // destination / source of row (lane & 31): blocks x+1 and x+2
        int st1 = 0, st2 = 0;                      // first edge of blocks x+1, x+2
        int2 bn = ok(b0 + 1) ? A.blk[b0 + 1] : make_int2(0, 0);   // table entry of the block the next fetch() handles
        auto fetch = [&](int x, int2 bi, int& st, int& dl, int& sl) {
            if (!ok(x)) return;
            const int cnt = bi.y & 0xff;
            st = bi.x;
            const int p = bi.x + (n < cnt ? n : cnt - 1);
            dl = A.dst[p];
            sl = A.src[p];
        };
        fetch(b0, A.blk[b0], st2, dl2, sl2);
        floatx4* ps = PS + jb * 256;
#pragma unroll 1
        for (int t = -2; t <= nb + 2; ++t) {
            const int x = b0 + t;
            int lane_t = lane0;
            asm volatile("" : "+v"(lane_t));
            const int lane = lane_t, n = lane & 31, hi = lane >> 5, rr = lane >> 3, cq = lane & 7;
            STAMP(0);
            {
                // accumulator = P_i[dst] + P_j[src] (already scaled): row-major sum -> tile -> accumulator layout
#pragma unroll
                for (int j = 0; j < 4; ++j) ps[tile_q(8 * j + rr, cq)] = pi[j] + pj[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 v = ps[tile_q(n, 2 * g + hi)];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[4 * g + tt] = v[tt];
                }
                STAMP(1);
                mlp_layer<true>(acc, wh, wl, Eimg + (x & 1) * 1024, lane, nothing);
                STAMP(2);
#if DEBUG_DUMP == 3
                if (ok(x)) { const int2 bi = A.blk[x]; if (n < (bi.y & 0xff)) {
                    for (int g = 0; g < 4; ++g) for (int tt = 0; tt < 4; ++tt) A.e_out[(int64_t)(bi.x + n) * H + 32 * jb + 8 * g + 4 * hi + tt] = acc[4 * g + tt] * 0.25f; } }
#endif
                acc_to_image(acc, reinterpret_cast<uintx4*>(X1 + (x & 1) * 1024), jb, lane);
            }
            STAMP(3);
            // requests: e rows of block x+2, P rows of block x+1 (whole 128-byte lines: 8 lanes per row)
#ifndef ABL_NOLOAD
            if (ok(x + 1)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = __builtin_amdgcn_ds_bpermute((8 * j + rr) * 4, dl1);
                    const int s = __builtin_amdgcn_ds_bpermute((8 * j + rr) * 4, sl1);
                    pi[j] = *reinterpret_cast<const floatx4*>(A.P + (unsigned)(d * (2 * H) + 32 * jb + 4 * cq));
                    pj[j] = *reinterpret_cast<const floatx4*>(A.P + (unsigned)(s * (2 * H) + H + 32 * jb + 4 * cq));
                }
            }
#endif
            st1 = st2; dl1 = dl2; sl1 = sl2;
            fetch(x + 3, bn, st2, dl2, sl2);
            if (ok(x + 4)) bn = A.blk[x + 4];
            STAMP(4);
            lds_barrier();
            STAMP(5);
        }
    } else if (role == 1) {
        // ------------------------------------------------------------------ role 1
        floatx16 acc;
        floatx4 er[4];                      // e rows (row-major quads) of block x-3 for the residual
        floatx4 eq[4];                      // e rows of block x+1 on their way into the operand image E
        int st_a = 0, cnt_a = 0, st_b = 0, cnt_b = 0;  // blocks x-3, x-2
        int2 bi_c = make_int2(0, 0);                    // raw table entry of block x-1 (decoded a tick after its load)
        int2 be = make_int2(0, 0);                      // raw table entry of block x+2 (its .x = first edge)
        const float* vgm = vecs + 2 * H + 32 * jb + 4 * cq;
        float* km = KM + jb * BE * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { er[j] = floatx4{0.f, 0.f, 0.f, 0.f}; eq[j] = er[j]; }
        be = A.blk[b0];
#pragma unroll 1
        for (int t = -2; t <= nb + 2; ++t) {
            const int x = b0 + t;
            int lane_t = lane0;
            asm volatile("" : "+v"(lane_t));
            const int lane = lane_t, n = lane & 31, hi = lane >> 5, rr = lane >> 3, cq = lane & 7;
            STAMP(0);
            const bool epi = ok(x - 3);
            const int par3 = (x - 3) & 1;
            {  // merged statistics of block x-3: lane n (both halves) -> km[n]
                float k, m;
                ln_merge(ST + par3 * 8 * BE * 2, n, A.inv_T, A.eps, k, m);
                if (hi == 0) *reinterpret_cast<float2v*>(km + n * 2) = float2v{k, m};
            }
            const floatx4* zt = Z + (par3 * 4 + jb) * 256;
            uintx2* ew = reinterpret_cast<uintx2*>(Eimg + ((x + 1) & 1) * 1024);
            auto conv_e = [&](int j) {   // e of block x+1 -> operand image E (row group j)
                const int r = 8 * j + rr;
                uintx2 h, l;
                split4(eq[j][0], eq[j][1], eq[j][2], eq[j][3], h, l);
                const int kks = 2 * jb + (cq >> 2), kg = cq & 1, half = (cq >> 1) & 1;
                const int slot = eslot(r, kg, cq >> 2);
                ew[((kks * 2 + 0) * 64 + slot) * 2 + half] = h;
                ew[((kks * 2 + 1) * 64 + slot) * 2 + half] = l;
            };
            float2v kmr;
            floatx4 zq, gm, bt;
            auto side = [&](int slot) {   // LayerNorm + e_out of block x-3, row group slot / 6
                const int j = slot / 6, r = 8 * j + rr;
                if (slot % 6 == 0) {
                    const float2v* kp = reinterpret_cast<const float2v*>(km + r * 2);
                    const floatx4* zp = zt + tile_q(r, cq);
                    kmr = *kp;
                    zq = *zp;
#ifdef EXP_KEEPADDR
                    asm volatile("" :: "v"(kp), "v"(zp));
#endif
                    gm = *reinterpret_cast<const floatx4*>(vgm);
                    bt = *reinterpret_cast<const floatx4*>(vgm + H);
                } else if (slot % 6 == 2) {
                    floatx4 o;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const float xh = fmaf(zq[tt], kmr[0], kmr[1]);
#if DEBUG_DUMP == 4
                        o[tt] = kmr[0] / A.inv_T;
#elif DEBUG_DUMP == 5
                        o[tt] = kmr[1];
#elif DEBUG_DUMP == 1
                        o[tt] = zq[tt] * A.inv_T;
#elif DEBUG_DUMP == 2
                        o[tt] = xh;
#else
                        o[tt] = fmaf(xh, gm[tt], bt[tt]) + er[j][tt];
#endif
                    }
#ifndef ABL_NOSTORE
                    if (DEBUG_DUMP != 3 && epi && r < cnt_a) *reinterpret_cast<floatx4*>(A.e_out + (unsigned)((st_a + r) * H + 32 * jb + 4 * cq)) = o;
#else
                    if (epi && r < cnt_a && o[0] == 1234.5f) *reinterpret_cast<floatx4*>(A.e_out + (unsigned)((st_a + r) * H + 32 * jb + 4 * cq)) = o;
#endif
                }
            };
            {
                {
                    const float* vb2 = vecs + 32 * jb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const floatx4 v = *reinterpret_cast<const floatx4*>(vb2 + 8 * g + 4 * hi);
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) acc[4 * g + tt] = v[tt];
                    }
                }
                conv_e(0); conv_e(1); conv_e(2); conv_e(3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // km visible to this wave's own reads
                STAMP(1);
                mlp_layer<false>(acc, wh, wl, X1 + ((x - 1) & 1) * 1024, lane, side);
                STAMP(2);
                acc_to_image(acc, reinterpret_cast<uintx4*>(X2 + ((x - 1) & 1) * 1024), jb, lane);
            }
            STAMP(3);
            st_a = st_b; cnt_a = cnt_b;
            st_b = bi_c.x; cnt_b = bi_c.y & 0xff;
#ifndef ABL_NOLOAD
            if (ok(x - 2) && A.residual) {  // rows of block x-2: consumed next tick
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int row = st_a + 8 * j + rr;
                    row = row < E ? row : E - 1;
                    er[j] = *reinterpret_cast<const floatx4*>(A.e_in + (unsigned)(row * H + 32 * jb + 4 * cq));
                }
            }
#endif
            if (ok(x)) bi_c = A.blk[x];
#ifndef ABL_NOLOAD
            if (ok(x + 2)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int row = be.x + 8 * j + rr;
                    row = row < E ? row : E - 1;
                    eq[j] = *reinterpret_cast<const floatx4*>(A.e_in + (unsigned)(row * H + 32 * jb + 4 * cq));
                }
            }
#endif
            if (ok(x + 3)) be = A.blk[x + 3];
            STAMP(4);
            lds_barrier();
            STAMP(5);
        }
    } else {
        // ------------------------------------------------------------------ role 2
        floatx16 acc, carry;
#pragma unroll
        for (int r = 0; r < 16; ++r) carry[r] = 0.f;
        int dn_a = -1, nx_a = -2, fl_a = 0, cnt_a = 0, dn_b = -1, nx_b = -2, fl_b = 0, cnt_b = 0;  // blocks x-3, x-2
        int head_a = -1, head_b = -1;
        int prev_last_dst = -3;
        const float* vgam = vecs + 2 * H + 32 * jb;
        const float* vbet = vecs + 3 * H + 32 * jb;
        int2 bn = make_int2(0, 0);   // table entry of the block the next fetch() handles (block x of the previous tick)
        auto fetch = [&](int x, int2 bi, int& dn, int& nx, int& fl, int& cnt, int& head) {
            if (!ok(x)) return;
            cnt = bi.y & 0xff;
            fl = bi.y >> 8;
            const int p = bi.x + n;
            dn = n < cnt ? A.dst[p] : -1 - n;
            nx = (n < cnt && p + 1 < E) ? A.dst[p + 1] : -2;
            if (fl & 1) {
                const int first = A.dst[bi.x];
                head = (bi.x > 0 && A.dst[bi.x - 1] == first) ? first : -1;
            }
        };
#pragma unroll 1
        for (int t = -2; t <= nb + 2; ++t) {
            const int x = b0 + t;
            int lane_t = lane0;
            asm volatile("" : "+v"(lane_t));
            const int lane = lane_t, n = lane & 31, hi = lane >> 5, rr = lane >> 3, cq = lane & 7;
            STAMP(0);
            const bool agg_on = ok(x - 3);
            float k = 0.f, m = 0.f, f1 = 0.f, f2 = 0.f, f4 = 0.f, f8 = 0.f, fb = 0.f, fc = 0.f;
            const int dn = dn_a;
            {
                ln_merge(ST + ((x - 3) & 1) * 8 * BE * 2, n, A.inv_T, A.eps, k, m);
                const int p1 = __builtin_amdgcn_update_dpp(-1000000, dn, 0x111, 0xf, 0xf, false);
                const int p2 = __builtin_amdgcn_update_dpp(-1000000, dn, 0x112, 0xf, 0xf, false);
                const int p4 = __builtin_amdgcn_update_dpp(-1000000, dn, 0x114, 0xf, 0xf, false);
                const int p8 = __builtin_amdgcn_update_dpp(-1000000, dn, 0x118, 0xf, 0xf, false);
                const int pb = __builtin_amdgcn_update_dpp(-1000000, dn, 0x142, 0xa, 0xf, false);
                f1 = p1 == dn ? 1.f : 0.f; f2 = p2 == dn ? 1.f : 0.f; f4 = p4 == dn ? 1.f : 0.f; f8 = p8 == dn ? 1.f : 0.f;
                fb = pb == dn ? 1.f : 0.f;
                fc = (n == 0 && !(fl_a & 1) && dn == prev_last_dst) ? 1.f : 0.f;
            }
            // aggregation of block x-3, four registers (one 16-byte piece of the destination row) per call, placed
            // between the MFMAs of block x-2
            bool is_last = false, part = false;
            if (agg_on) {
                const bool lastf = (fl_a & 2) != 0;
                is_last = n < cnt_a && (nx_a != dn || (lastf && n == cnt_a - 1));
                part = dn == head_a || (lastf && n == cnt_a - 1 && nx_a == dn);
            }
            float* arow = A.agg + (unsigned)((dn < 0 ? 0 : dn) * H + 32 * jb + 4 * hi);
            const floatx4* zt3 = Z + (((x - 3) & 1) * 4 + jb) * 256;
            floatx4 zq, gmv, btv;
            float y[4];
            auto side = [&](int slot) {   // chunk g = slot / 6: registers 4g..4g+3 = one 16-byte piece of the destination rows
                const int g = slot / 6;
                switch (slot % 6) {
                case 0:
                    zq = zt3[tile_q(n, 2 * g + hi)];
                    gmv = *reinterpret_cast<const floatx4*>(vgam + 8 * g + 4 * hi);
                    btv = *reinterpret_cast<const floatx4*>(vbet + 8 * g + 4 * hi);
                    break;
                case 1:
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const float xh = fmaf(zq[tt], k, m);
                        y[tt] = fmaf(carry[4 * g + tt], fc, fmaf(xh, gmv[tt], btv[tt]));
                    }
                    break;
                case 2:
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f1, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC(y[tt], f2, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
                    break;
                case 3:
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f4, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC(y[tt], f8, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
                    break;
                case 4:
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], fb, "row_bcast:15 row_mask:0xa bank_mask:0xf");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        carry[4 * g + tt] = __uint_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) | 31) * 4, __float_as_uint(y[tt])));
                    break;
                default:
#ifdef ABL_NOSTORE
                    if (is_last && y[0] == 1234.5f) {
#else
                    if (is_last) {
#endif
                        if (part) {
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) atomicAdd(arow + 8 * g + tt, y[tt]);
                        } else {
                            *reinterpret_cast<floatx4*>(arow + 8 * g) = floatx4{y[0], y[1], y[2], y[3]};
                        }
                    }
                    break;
                }
            };
            const bool l3 = ok(x - 2);
            const int par2 = (x - 2) & 1;
            {
                const float* vb3 = vecs + H + 32 * jb;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 v = *reinterpret_cast<const floatx4*>(vb3 + 8 * g + 4 * hi);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[4 * g + tt] = v[tt];
                }
                STAMP(1);
                mlp_layer<false>(acc, wh, wl, X2 + par2 * 1024, lane, side);
            }
            STAMP(2);
            if (agg_on) prev_last_dst = (cnt_a == BE && !(fl_a & 2)) ? __builtin_amdgcn_readlane(dn, 31) : -3;
            STAMP(3);
            {
                // LayerNorm partial statistics of the scaled accumulators; raw accumulators to Z (tile order)
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[r];
                const float mh = s * (1.0f / 16.0f);
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[r] - mh; q = fmaf(d, d, q); }
                if (l3) *reinterpret_cast<float2v*>(ST + ((par2 * 8 + jb * 2 + hi) * BE + n) * 2) = float2v{mh, q};
                floatx4* zt = Z + (par2 * 4 + jb) * 256;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 z;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) z[tt] = acc[4 * g + tt];
                    zt[tile_q(n, 2 * g + hi)] = z;
                }
            }
            dn_a = dn_b; nx_a = nx_b; fl_a = fl_b; cnt_a = cnt_b;
            if (fl_b & 1) head_a = head_b;
            fetch(x - 1, bn, dn_b, nx_b, fl_b, cnt_b, head_b);
            if (ok(x)) bn = A.blk[x];
            STAMP(4);
            lds_barrier();
            STAMP(5);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
static float f16_round(float x) { return (float)(_Float16)x; }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 100000;
    const bool local = argc > 2 && atoi(argv[2]) != 0;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::normal_distribution<float> G(0.f, 1.f);
    std::vector<int> dst, src;
    for (int i = 0; i < N; ++i) {
        const int deg = 14 + (int)(rng() % 13);
        for (int k = 0; k < deg; ++k) {
            dst.push_back(i);
            int s = local ? (int)((i + (int)(rng() % 2001) - 1000 + N) % N) : (int)(rng() % N);
            src.push_back(s);
        }
    }
    const int E = (int)dst.size();
    std::vector<float> W[3], b[3];
    float sc[3];
    const float bound = 1.0f / sqrtf(128.f);
    for (int l = 0; l < 3; ++l) {
        W[l].resize(H * H); b[l].resize(H);
        float mx = 0.f;
        for (auto& w : W[l]) { w = U(rng) * bound; mx = std::max(mx, fabsf(w)); }
        for (auto& x : b[l]) x = U(rng) * bound;
        int ex; frexpf(mx, &ex);
        sc[l] = ldexpf(1.f, -ex - 1);   // max |W| * t in [0.25, 0.5)
    }
    std::vector<float> gam(H), bet(H);
    for (int i = 0; i < H; ++i) { gam[i] = 1.f + 0.1f * U(rng); bet[i] = 0.1f * U(rng); }
    std::vector<float> P((size_t)N * 2 * H), e((size_t)E * H);
    for (auto& x : P) x = 0.5f * G(rng);
    for (auto& x : e) x = G(rng);
    std::vector<_Float16> wimg((size_t)3 * 4 * 8 * 2 * 64 * 8);
    for (int l = 0; l < 3; ++l)
        for (int w = 0; w < 4; ++w)
            for (int ks = 0; ks < 8; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int i = lane & 31, kg = lane >> 5;
                        const int k = 16 * ks + 8 * (j >> 2) + 4 * kg + (j & 3);
                        const float v = W[l][(32 * w + i) * H + k] * sc[l];
                        const float h = f16_round(v);
                        const float lo = f16_round(v - h);
                        const size_t base = ((((size_t)(l * 4 + w) * 8 + ks) * 2) * 64 + lane) * 8 + j;
                        wimg[base] = (_Float16)h;
                        wimg[base + 64 * 8] = (_Float16)lo;
                    }
    std::vector<float> vec(4 * H);
    const float T1 = sc[0], T2 = T1 * sc[1], T3 = T2 * sc[2];
    for (int i = 0; i < H; ++i) { vec[i] = b[1][i] * T2; vec[H + i] = b[2][i] * T3; vec[2 * H + i] = gam[i]; vec[3 * H + i] = bet[i]; }
    for (int i = 0; i < N; ++i) for (int f = 0; f < H; ++f) P[(size_t)i * 2 * H + f] += b[0][f];
    std::vector<float> Ps(P.size());
    for (size_t i = 0; i < P.size(); ++i) Ps[i] = P[i] * T1;
    // block / chunk tables (one graph)
    const int nblk = (E + BE - 1) / BE;
    std::vector<int2> blk(nblk);
    std::vector<int> chunk_first;
    for (int j = 0; j < nblk; ++j) {
        int fl = 0;
        if (j % CB == 0) { fl |= 1; chunk_first.push_back(j); }
        if (j % CB == CB - 1 || j == nblk - 1) fl |= 2;
        blk[j] = make_int2(j * BE, std::min(BE, E - j * BE) | (fl << 8));
    }
    const int nchunks = (int)chunk_first.size();
    chunk_first.push_back(nblk);

    int *d_dst, *d_src, *d_cf; int2* d_blk; float *d_P, *d_e, *d_eo, *d_agg, *d_vec; half8* d_w;
    CK(hipMalloc(&d_dst, E * 4)); CK(hipMalloc(&d_src, E * 4)); CK(hipMalloc(&d_cf, chunk_first.size() * 4)); CK(hipMalloc(&d_blk, blk.size() * 8));
    CK(hipMalloc(&d_P, P.size() * 4)); CK(hipMalloc(&d_e, e.size() * 4)); CK(hipMalloc(&d_eo, e.size() * 4));
    CK(hipMalloc(&d_agg, (size_t)N * H * 4)); CK(hipMalloc(&d_vec, vec.size() * 4)); CK(hipMalloc(&d_w, wimg.size() * 2));
    CK(hipMemcpy(d_dst, dst.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_src, src.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cf, chunk_first.data(), chunk_first.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_blk, blk.data(), blk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_P, Ps.data(), Ps.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_e, e.data(), e.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vec, vec.data(), vec.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, wimg.data(), wimg.size() * 2, hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 128 * 12 * 8 * 8)); CK(hipMemset(d_st, 0, 128 * 12 * 8 * 8));
    SysArgs A{d_dst, d_src, d_P, d_e, d_eo, d_agg, d_w, d_vec, d_blk, d_cf, nchunks, E, 1.f / T3, 1e-5f, 1, d_st};
    const size_t lds = 3 * 2 * 16384 + 2 * 16384 + 16384 + 2 * 8 * BE * 2 * 4 + 4 * BE * 2 * 4 + 4 * H * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_edge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int ncu = 256; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = std::min(nchunks, ncu);
    CK(hipMemset(d_agg, 0, (size_t)N * H * 4));
    hipLaunchKernelGGL(sys_edge_kernel, dim3(grid), dim3(SYS_THREADS), lds, 0, A);
    CK(hipDeviceSynchronize());
    std::vector<float> eo(e.size()), agg((size_t)N * H);
    CK(hipMemcpy(eo.data(), d_eo, eo.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(agg.data(), d_agg, agg.size() * 4, hipMemcpyDeviceToHost));

    double max_e = 0, max_a = 0, ref_e = 0, ref_a = 0; int nbad = 0;
    std::vector<int> start(N + 1, 0);
    for (int p = 0; p < E; ++p) start[dst[p] + 1]++;
    for (int i = 0; i < N; ++i) start[i + 1] += start[i];
    std::vector<int> samples = {0, 1, 2, 3, 4, 5, 6, 7, 1000, 1001, 1002, N / 2, N / 2 + 1, N - 3, N - 2, N - 1};
    // nodes whose segment straddles a chunk boundary
    for (int c = 1; c < nchunks && samples.size() < 40; c += std::max(1, nchunks / 12)) samples.push_back(dst[chunk_first[c] * BE]);
    for (int node : samples) {
        if (node < 0 || node >= N) continue;
        std::vector<double> asum(H, 0.0);
        for (int p = start[node]; p < start[node + 1]; ++p) {
            std::vector<double> x(H), z(H), z1(H);
            for (int f = 0; f < H; ++f) {
                double s = (double)P[(size_t)dst[p] * 2 * H + f] + (double)P[(size_t)src[p] * 2 * H + H + f];
                for (int k = 0; k < H; ++k) s += (double)W[0][f * H + k] * (double)e[(size_t)p * H + k];
                z1[f] = s;
                x[f] = std::max(s, 0.0);
            }
            for (int l = 1; l < 3; ++l) {
                for (int f = 0; f < H; ++f) {
                    double s = b[l][f];
                    for (int k = 0; k < H; ++k) s += (double)W[l][f * H + k] * x[k];
                    z[f] = s;
                }
                if (l == 1) for (int f = 0; f < H; ++f) x[f] = std::max(z[f], 0.0);
            }
            double mean = 0, var = 0;
            for (int f = 0; f < H; ++f) mean += z[f];
            mean /= H;
            for (int f = 0; f < H; ++f) var += (z[f] - mean) * (z[f] - mean);
            var /= H;
            for (int f = 0; f < H; ++f) {
                const double yv = (z[f] - mean) / sqrt(var + 1e-5) * gam[f] + bet[f];
                asum[f] += yv;
#if DEBUG_DUMP == 4
                const double want = 1.0 / sqrt(var + 1e-5);
#elif DEBUG_DUMP == 5
                const double want = -mean / sqrt(var + 1e-5);
#elif DEBUG_DUMP == 3
                const double want = z1[f];
#elif DEBUG_DUMP == 1
                const double want = z[f];
#elif DEBUG_DUMP == 2
                const double want = (z[f] - mean) / sqrt(var + 1e-5);
#else
                const double want = yv + e[(size_t)p * H + f];
#endif
                if (fabs(want - eo[(size_t)p * H + f]) > 1e-3 && nbad++ < 24) printf("bad: edge %d (block %d row %d) feature %d got %.5f want %.5f\n", p, p / 32, p % 32, f, eo[(size_t)p * H + f], want);
                max_e = std::max(max_e, fabs(want - eo[(size_t)p * H + f]));
                ref_e = std::max(ref_e, fabs(want));
            }
        }
        for (int f = 0; f < H; ++f) {
            max_a = std::max(max_a, fabs(asum[f] - agg[(size_t)node * H + f]));
            ref_a = std::max(ref_a, fabs(asum[f]));
        }
    }
    printf("N=%d E=%d blocks=%d chunks=%d grid=%d  e_out max abs err %.3e (max |ref| %.3f)  agg max abs err %.3e (max |ref| %.3f)\n", N, E, nblk,
           nchunks, grid, max_e, ref_e, max_a, ref_a);
    {
        double bad = 0;
        std::vector<double> ca(H, 0.0), ce(H, 0.0);
        for (int i = 0; i < N; ++i) for (int f = 0; f < H; ++f) ca[f] += agg[(size_t)i * H + f];
        for (int p = 0; p < E; ++p) for (int f = 0; f < H; ++f) ce[f] += (double)eo[(size_t)p * H + f] - (double)e[(size_t)p * H + f];
        for (int f = 0; f < H; ++f) bad = std::max(bad, fabs(ca[f] - ce[f]) / (1.0 + fabs(ce[f])));
        // per-node check of every row: agg row sums vs per-destination sums of (e_out - e_in)
        double worst = 0;
        std::vector<double> rs(H);
        for (int i = 0; i < N; ++i) {
            std::fill(rs.begin(), rs.end(), 0.0);
            for (int p = start[i]; p < start[i + 1]; ++p)
                for (int f = 0; f < H; ++f) rs[f] += (double)eo[(size_t)p * H + f] - (double)e[(size_t)p * H + f];
            for (int f = 0; f < H; ++f) worst = std::max(worst, fabs(rs[f] - agg[(size_t)i * H + f]));
        }
        printf("sum check (agg vs e_out - e_in): per column relative %.3e, every node row max abs %.3e\n", bad, worst);
    }
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(sys_edge_kernel, dim3(grid), dim3(SYS_THREADS), lds, 0, A);
    const int reps = 20;
    CK(hipEventRecord(t0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(sys_edge_kernel, dim3(grid), dim3(SYS_THREADS), lds, 0, A);
    CK(hipEventRecord(t1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, t0, t1)); ms /= reps;
    const double flops = (double)E * 2.0 * 3 * H * H * 3;
    const double bytes = (double)E * H * 4 * 2 + (double)N * H * 4 * 3 + (double)E * 12;
    printf("kernel %.3f ms  | f16 MFMA issued %.1f TF (%.3f of 2.5 PF) | fp32-equivalent %.1f TF | algorithmic %.2f GB -> %.2f TB/s (%.3f of 8 TB/s)\n",
           ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, flops / 3 / ms / 1e9, bytes / 1e9, bytes / ms / 1e9, bytes / ms / 1e9 / 8000.0);
#ifdef STAMPS
    {
        std::vector<unsigned long long> st(128 * 12 * 8);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 12; w += 1) {
            double d[8] = {0}; double tick = 0; int cnt = 0;
            for (int t = 4; t < 120; ++t) {
                const unsigned long long* q = &st[((size_t)t * 12 + w) * 8];
                const unsigned long long* qn = &st[((size_t)(t + 1) * 12 + w) * 8];
                const int last = 5;
                for (int k = 0; k < last; ++k) d[k] += (double)(q[k + 1] - q[k]);
                tick += (double)(qn[0] - q[0]); cnt++;
            }
            printf("wave %2d (role %d): tick %.0f :", w, w / 4, tick / cnt);
            for (int k = 0; k < 5; ++k) printf(" s%d %.0f", k, d[k] / cnt);
            printf("\n");
        }
    }
#endif
    return 0;
}
