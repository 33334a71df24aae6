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

// The wave's 32 rows x 32 NKB features (accumulator layout) to 32 CONSECUTIVE rows of a row-major array as whole 128-byte lines:
// each 32-feature block takes a turn through a wave-private LDS tile (32 rows of TURN_LD floats; no barrier: one wave's LDS
// instructions execute in order), after which a store instruction covers 8 rows x 128 bytes instead of 32 rows x 32 bytes.
// dst = row 0 of the wave's rows, ld = row stride (floats), rows_left = rows of the array from dst on (any value: rows past
// the end are dropped by the buffer bound, and the instruction count -- 4 NKB, what run_layer's PEND expects -- does not vary).
// a wave-uniform pointer the compiler cannot prove uniform (it would wrap every buffer instruction that uses a resource built from it
// in a waterfall loop): both halves through v_readfirstlane
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
// No barrier or fence separates a turn's writes from its reads, and none is needed -- neither in hardware (one wave's LDS
// instructions execute in order) nor against the compiler: every read instruction of a turn has a lane that reads a piece THAT LANE
// wrote (the transposition's diagonal: lanes 0 / 45 / 18 / 63 for row groups 0 .. 3), and every write of the next turn has a lane
// that overwrites a piece it has just read.  A lane's own read and write addresses therefore do alias, no proof to the contrary
// exists, and the single-thread memory model itself keeps each such pair in program order -- for the whole wave, since all lanes
// execute one instruction stream.  tests/test_turn_tiles.py checks that property of the index maps (here and in hmlp.hip's
// HM_LINES turn).  Explicit fences were measured instead (round 6, tools/ab_train.sh): any form -- all address spaces, LDS-only
// at wavefront scope, volatile tile accesses -- costs the training step 2.8 % (the chains' global prefetches no longer cross a turn).
constexpr int TURN_LD = 36;
constexpr int TURN_FLOATS = 32 * TURN_LD;
typedef unsigned int uintx4_t __attribute__((ext_vector_type(4)));
template <int NKB>
__device__ __forceinline__ void store_feat_lines(const floatx16 (&v)[NKB], float* dst, int ld, int rows_left, float* turn, int lane) {
    const int n = lane & 31, hi = lane >> 5, rr = lane >> 3, cq = lane & 7;
    const int cnt = rows_left < 0 ? 0 : rows_left > 32 ? 32 : rows_left;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(dst), 0, (unsigned)__builtin_amdgcn_readfirstlane(cnt * ld * 4), 0x00020000);
    float* wr = turn + n * TURN_LD + 4 * hi;
    const float* rd = turn + rr * TURN_LD + 4 * cq;
    const unsigned voff = ((unsigned)rr * (unsigned)ld + 4u * cq) * 4u;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            floatx4 x;
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = v[kb][4 * g + t];
            *reinterpret_cast<floatx4*>(wr + 8 * g) = x;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const floatx4 o = *reinterpret_cast<const floatx4*>(rd + 8 * j * TURN_LD);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, o), srd, voff + (unsigned)(8 * j * ld + 32 * kb) * 4u, 0, 0);
        }
    }
}

// The reverse turn: 32 CONSECUTIVE rows of a row-major array into the accumulator layout, fetched as whole lines (a load
// instruction covers 8 rows x 128 bytes).  Rows past the end read as zero.  MODE 0: v = rows, 1: v += rows, 2: v = rows > 0 ? v : 0.
template <int MODE>
__device__ __forceinline__ void turn_in(floatx16& v, const floatx4 (&x)[4], float* turn, int lane) {
    const int n = lane & 31, hi = lane >> 5, rr = lane >> 3, cq = lane & 7;
    float* wr = turn + rr * TURN_LD + 4 * cq;
    const float* rd = turn + n * TURN_LD + 4 * hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<floatx4*>(wr + 8 * j * TURN_LD) = x[j];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 y = *reinterpret_cast<const floatx4*>(rd + 8 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) v[4 * g + t] = y[t];
            else if (MODE == 1) v[4 * g + t] += y[t];
            else v[4 * g + t] = y[t] > 0.f ? v[4 * g + t] : 0.f;
        }
    }
}
template <int MODE, int NKB>
__device__ __forceinline__ void load_feat_lines(floatx16 (&v)[NKB], const float* src, int ld, int rows_left, float* turn, int lane) {
    const int rr = lane >> 3, cq = lane & 7;
    const int cnt = rows_left < 0 ? 0 : rows_left > 32 ? 32 : rows_left;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float*>(src)), 0, (unsigned)__builtin_amdgcn_readfirstlane(cnt * ld * 4), 0x00020000);
    const unsigned voff = ((unsigned)rr * (unsigned)ld + 4u * cq) * 4u;
    if (MODE == 0) {
        // every line of the tile in flight at once, into the registers they end up in; then one block after the other takes its turn
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const floatx4 x = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff + (unsigned)(8 * j * ld + 32 * kb) * 4u, 0, 0));
#pragma unroll
                for (int t = 0; t < 4; ++t) v[kb][4 * j + t] = x[t];
            }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            floatx4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) x[j][t] = v[kb][4 * j + t];
            turn_in<0>(v[kb], x, turn, lane);
        }
    } else {
        // one block ahead (16 registers of lines in flight next to the 16 being turned)
        floatx4 cur[4], nxt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff + (unsigned)(8 * j * ld) * 4u, 0, 0));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff + (unsigned)(8 * j * ld + 32 * (kb + 1)) * 4u, 0, 0));
            }
            turn_in<MODE>(v[kb], cur, turn, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
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
