// Device building blocks of the training kernels' chains (train.hip): LDS-DMA weight stream, the transposed MFMA layer on the 32 x 32
// accumulator layout, register <-> feature-row helpers.  (Round 1's fp32 inference kernels were built from the same pieces; they
// were removed in round 5.)
#pragma once
#include "common.h"
#include "mlp.h"

namespace gm {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int PIECE_FLOATS = 256;                  // 1 KiB
constexpr int STAGE_PIECES = 16;
constexpr int STAGE_FLOATS = PIECE_FLOATS * STAGE_PIECES;  // 16 KiB
constexpr int TILE = 128;                          // graph elements per workgroup tile
constexpr int THREADS = 256;
constexpr int TS = 68;                             // LDS row stride (floats) of the 64-feature staging tile

// ------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------
struct WStream {
    const float* base;  // stage 0 of this kernel's packed stream (global)
    float* ring;        // LDS, 2 * STAGE_FLOATS
    int total;          // stages per tile
    int cur;            // next stage to consume (index within the tile sequence)
    int parity;         // ring buffer holding stage `cur`
    int lane, wave;
};

__device__ __forceinline__ void issue_stage(const WStream& ws, int stage, int buf) {
#pragma unroll
    for (int c = 0; c < STAGE_PIECES / 4; ++c) {
        const int piece = c * 4 + ws.wave;  // one wave-instruction = one contiguous 1 KiB piece
        const float* g = ws.base + (size_t)stage * STAGE_FLOATS + piece * PIECE_FLOATS + ws.lane * 4;
        float* l = ws.ring + buf * STAGE_FLOATS + piece * PIECE_FLOATS;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
}

// Workgroup barrier that orders LDS traffic only (does not drain global stores / loads in flight).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One Linear: acc[jb] += W(jb-block rows) . act.   NKQ = K/8 input octets, NJB = OUT/32 blocks.
// `more` = another stage will be consumed after this layer's last one (this tile or the next).
// PEND: number of vector-memory instructions (stores of results that nothing in this layer reads) the
// caller has issued, on EVERY path and in every wave, after the previous run_layer returned.  vmcnt
// retires in issue order, so waiting for "at most PEND outstanding" still guarantees the stage's DMA
// (issued earlier) has landed, without draining those stores in front of the MFMAs.
template <int NKQ, int NJB, int NKB, int PEND = 0>
__device__ __forceinline__ void run_layer(floatx16 (&acc)[NJB], const floatx16 (&act)[NKB], WStream& ws, bool more_tiles) {
    constexpr int NP = NKQ * NJB;
    constexpr int NST = (NP + STAGE_PIECES - 1) / STAGE_PIECES;
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (PEND > 0 && s == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PEND) : "memory");
            lds_barrier();  // LDS-only barrier: __syncthreads() would drain the pending stores
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // stage `cur` has landed for every wave; the other buffer is free
        }
        int nxt = ws.cur + 1;
        const bool wrap = nxt == ws.total;
        if (wrap) nxt = 0;
        {
            // launder the stage index: otherwise every stage's DMA addresses are precomputed outside
            // the tile loop and spilled
            int st = nxt;
            asm volatile("" : "+s"(st));
            if (!wrap || more_tiles) issue_stage(ws, st, ws.parity ^ 1);
            // keep the DMA issue HERE, right behind the barrier: the scheduler otherwise sinks it below
            // the stage's MFMAs, next to the wait that needs it, and the copy no longer overlaps them
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* buf = ws.ring + ws.parity * STAGE_FLOATS + ws.lane * 4;
        // A operands are fetched one group of pieces ahead of the MFMAs that consume them, so the LDS
        // latency sits under the matrix pipe instead of in front of it.
        constexpr int GP = NJB >= 2 ? 2 : 1;  // pieces per group
        constexpr int NG = STAGE_PIECES / GP;
        floatx4 a_cur[GP], a_nxt[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q)
            if (s * STAGE_PIECES + q < NP) a_cur[q] = *reinterpret_cast<const floatx4*>(buf + q * PIECE_FLOATS);
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) {
            if (gidx + 1 < NG) {
#pragma unroll
                for (int q = 0; q < GP; ++q) {
                    const int slot = (gidx + 1) * GP + q;
                    if (s * STAGE_PIECES + slot < NP) a_nxt[q] = *reinterpret_cast<const floatx4*>(buf + slot * PIECE_FLOATS);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < GP; ++q) {
                    const int p = s * STAGE_PIECES + gidx * GP + q;
                    if (p < NP) {
                        const int kq = p / NJB, jb = p % NJB;
                        acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q][t], act[kq >> 2][(kq & 3) * 4 + t], acc[jb], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < GP; ++q) a_cur[q] = a_nxt[q];
        }
        ws.cur = nxt;
        ws.parity ^= 1;
    }
}

// registers <-> feature vectors.  v[kb][4g + t] <-> row[32 kb + 8 g + 4 hi + t]
template <int NKB>
__device__ __forceinline__ void load_feat(floatx16 (&v)[NKB], const float* __restrict__ row, int hi) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 x = *reinterpret_cast<const floatx4*>(row + 32 * kb + 8 * g + 4 * hi);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[kb][4 * g + t] = x[t];
        }
}
// v += row, one 32-feature block at a time (bounds the registers held by in-flight loads)
template <int NKB>
__device__ __forceinline__ void add_feat(floatx16 (&v)[NKB], const float* __restrict__ row, int hi) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 x = *reinterpret_cast<const floatx4*>(row + 32 * kb + 8 * g + 4 * hi);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[kb][4 * g + t] += x[t];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// guarded scalar loads for a raw input row of k (< 32*NKB) floats
template <int NKB>
__device__ __forceinline__ void load_feat_guard(floatx16 (&v)[NKB], const float* __restrict__ row, int hi, int k) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3);
            v[kb][r] = f < k ? row[f] : 0.f;
        }
}
template <int NKB>
__device__ __forceinline__ void store_feat(const floatx16 (&v)[NKB], float* __restrict__ row, int hi) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            floatx4 x;
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = v[kb][4 * g + t];
            *reinterpret_cast<floatx4*>(row + 32 * kb + 8 * g + 4 * hi) = x;
        }
}
template <int NKB>
__device__ __forceinline__ void relu_to(floatx16 (&dst)[NKB], const floatx16 (&src)[NKB]) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[kb][r] = fmaxf(src[kb][r], 0.f);
}

}  // namespace gm
