// DEVELOPMENT BUILDS ONLY (GM_DEV_KERNELS=1 python -m gnn_manip_amd.build --tag=dev): the round-1 inference kernels -- fp32 MFMA
// chains (v_mfma_f32_32x32x2_f32 / 16x16x4_f32 with LDS-DMA weight stages) and the six-product bf16 forms -- kept as A/B and
// accuracy references.  The product library does not compile this file (gnn_manip_amd/build.py); its design notes are the
// comments below and the round-1 revision of DESIGN.md.
#include <string.h>
#include "common.h"
#include "mlp.h"

#include "mlp_dev.h"
#include "hedge.h"
#include "hmlp.h"

namespace gm {

// Round-1 inference kernels (fp32 MFMA chains and the six-product bf16 forms): A/B and accuracy references of DEVELOPMENT builds
// (GM_DEV_KERNELS=1 python -m gnn_manip_amd.build --tag=dev); the product library does not contain them.
// ------------------------------------------------------------------------------------------
// EDGE kernel
// ------------------------------------------------------------------------------------------
// Diagnostic stamps: 100 MHz s_memrealtime per tile phase, written to a buffer that no other code reads.  The
// library always passes a null buffer (launch_edge); round 1's reader script went with the stamp setter of the ABI.
#define GM_STAMP(k)                                                                          \
    do {                                                                                     \
        if (A.stamps && tid == 0) A.stamps[(size_t)tile * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// MODE 0: encoder phi_e on raw edge attributes; 1: processor phi_e with the residual e <- e' + e
// (fused forward); 2: processor phi_e without residual (InteractionNetwork block API).
template <int H, int NL, int MODE>
__global__ void __launch_bounds__(THREADS, H <= 128 ? 2 : 1) edge_kernel(EdgeArgs A) {
    constexpr bool ENC = MODE == 0;
    constexpr bool with_resid = MODE == 1;
    constexpr int NJB = H / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    float* T = ring + 2 * STAGE_FLOATS;
    int* sdst = reinterpret_cast<int*>(T + TILE * TS);                 // 2 x [TILE + 2] (+2 pad): per tile parity
    float* headv = reinterpret_cast<float*>(sdst + 2 * (TILE + 4));   // 2 x (head[4][64], tail[4][64]): per chunk parity
    int tpar = 0;

    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6, n0 = lane0 & 31;
    const int E = A.hdr ? A.hdr->n_edges : A.n_edges_host;

    const int ntiles = (E + TILE - 1) / TILE;

    WStream ws;
    ws.base = A.wstream;
    ws.ring = ring;
    ws.total = ENC ? (1 + NL * (H / 8) * NJB / STAGE_PIECES) : ((NL + 1) * (H / 8) * NJB / STAGE_PIECES);
    ws.cur = 0;
    ws.parity = 0;
    ws.lane = lane0;
    ws.wave = wave0;
    if ((int)blockIdx.x < ntiles) issue_stage(ws, 0, 0);
    // Per-tile indices are fetched one tile ahead (registers), so that a tile's gathers do not wait
    // behind an index load.
    struct TileIdx { int er, d, sr, dq, sd; };
    auto fetch_idx = [&](int tile) {
        TileIdx ix;
        const int p0 = tile * TILE;
        const int p = p0 + wave0 * 32 + n0;
        const int pc = p < E ? p : E - 1;
        ix.er = A.eid ? A.eid[pc] : pc;
        ix.d = ix.sr = 0;
        ix.dq = ix.sd = -1;
        if (!ENC) {
            ix.d = A.dst[pc];
            ix.sr = A.src[pc];
            if (tid0 < TILE + 2) {
                const int pp = p0 - 1 + tid0;
                ix.sd = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
            }
            if (lane0 < 34) {  // lane l: destination of row (32*wave - 1 + l) of the tile; -2 before edge 0, -1 past E
                const int pp = p0 + 32 * wave0 - 1 + lane0;
                ix.dq = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
            }
        }
        return ix;
    };
    TileIdx nx = {0, 0, 0, -1, -1};
    if ((int)blockIdx.x < ntiles) nx = fetch_idx(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int p0 = tile * TILE;
        const TileIdx ix = nx;
        if (!(A.debug & 16)) prio_latency_phase();
        // Per-tile copies of the thread coordinates, laundered so that the prologue / epilogue address
        // arithmetic is recomputed per tile instead of being hoisted out of the loop and spilled (a spill
        // reload is a full memory round trip).  The weight stream keeps the un-laundered lane0 / wave0.
        int tid_t = tid0;
        asm volatile("" : "+v"(tid_t));
        const int tid = tid_t, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
        GM_STAMP(0);
        if (A.stamps && tid == 0) {
            A.stamps[(size_t)tile * 16 + 14] = ((unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) << 32) |
                                             __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_ID, XCC_ID
            A.stamps[(size_t)tile * 16 + 15] = blockIdx.x;
        }
        int* sd = sdst + tpar * (TILE + 4);
        if (!ENC && tid < TILE + 2) sd[tid] = ix.sd;  // ordered before its readers by the stage barriers
        const int dq = ix.dq;
        floatx16 acc[NJB], act[NJB];
        if (ENC) {
            load_feat_guard(act, A.e_in + (int64_t)ix.er * A.k1, hi, A.k1);
            load_feat(acc, A.bias, hi);
            if (more_tiles) nx = fetch_idx(tile + gridDim.x);
            prio_mfma_phase();
            // layer 1: K = edge_dim padded to 8 -> one k-octet (edge_dim <= 8 checked on the host)
            run_layer<1, NJB, NJB>(acc, act, ws, more_tiles);
            mlp_tail_layers<H, NL>(acc, act, A.bias + H, ws, more_tiles, hi);
        } else {
            if (A.debug & 8) {  // ablation: no gathers
                load_feat(acc, A.bias, hi);
                load_feat(act, A.bias, hi);
            } else {
                load_feat(acc, A.P + (int64_t)ix.d * (2 * H), hi);        // P_i[dst] (+ b1)
                add_feat(acc, A.P + (int64_t)ix.sr * (2 * H) + H, hi);    // P_j[src]
                load_feat(act, A.e_in + (int64_t)ix.er * H, hi);
            }
            if (more_tiles) nx = fetch_idx(tile + gridDim.x);
            prio_mfma_phase();
            if (!(A.debug & 1)) {
                run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);  // + W_e e
                GM_STAMP(1);
                mlp_tail_layers<H, NL>(acc, act, A.bias, ws, more_tiles, hi);
            }
            GM_STAMP(2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(A.debug & 16)) prio_latency_phase();
        layer_norm_regs(acc, A.ln_g, A.ln_b, A.eps, hi);
        GM_STAMP(3);
        if (A.debug & 2) {  // ablation: no epilogue (keep the result alive)
            if (acc[0][0] == 123.456f) A.e_out[p0] = acc[1][1] + acc[2][2] + acc[3][3];
            continue;
        }

        // ---- epilogue: 64 features at a time through the LDS tile, which serves the segmented reduction
        // (scatter-add) and whole-row coalesced stores of e_out = e' (+ e_in).  Barriers here order LDS
        // traffic only (lgkmcnt): a __syncthreads() would also drain every global store in flight.
        constexpr int NCH = H / 64;
        constexpr int NPASS = TILE / 16;
        const bool do_agg = !ENC && !(A.debug & 4);
        const int c4 = (tid & 15) * 4;
#pragma unroll
        for (int fh = 0; fh < NCH; ++fh) {
            float* hv = headv + (fh & 1) * 512;  // head / tail partials double-buffered across chunks
            float* tl = hv + 256;
#pragma unroll
            for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 x;
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[t] = acc[2 * fh + jb2][4 * g + t];
                    *reinterpret_cast<floatx4*>(T + (wave * 32 + n) * TS + 32 * jb2 + 8 * g + 4 * hi) = x;
                }
            lds_barrier();
            if (do_agg) {
                // segmented sum over destination-sorted rows: wave q owns rows 32q..32q+31, lane = column.
                // Row destinations come from a lane-held register through v_readlane (scalar, so every
                // branch is a scalar branch); the 32 tile values are fetched with independent LDS reads.
                const int r0 = 32 * wave;
                float tv[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) tv[r] = T[(r0 + r) * TS + lane];
                float run = 0.f;
                int d = __builtin_amdgcn_readlane(dq, 1);
                bool first = d >= 0 && __builtin_amdgcn_readlane(dq, 0) == d;
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int dn = __builtin_amdgcn_readlane(dq, r + 2);
                    if (d >= 0) run += tv[r];
                    if (dn != d && d >= 0) {
                        if (first) hv[wave * 64 + lane] = run;
                        else A.agg[(int64_t)d * H + 64 * fh + lane] = run;
                        run = 0.f;
                        first = false;
                    }
                    d = dn;
                }
                {
                    const int dl = __builtin_amdgcn_readlane(dq, 32);
                    if (dl >= 0 && __builtin_amdgcn_readlane(dq, 33) == dl) tl[wave * 64 + lane] = run;
                }
                lds_barrier();  // T is free again; head / tail partials are visible
                if (wave == 0) {  // stitch segments that cross quarter / tile boundaries
                    float carry = 0.f;
                    bool ext = false;
#pragma unroll 1
                    for (int q = 0; q < 4; ++q) {
                        const int q0 = 32 * q;
                        const int df = sd[q0 + 1];
                        const bool cont_in = df >= 0 && sd[q0] == df;
                        const bool through = cont_in && sd[q0 + 32] == df && sd[q0 + 33] == df;
                        if (cont_in) {
                            if (q == 0) { carry = 0.f; ext = true; }
                            if (through) {
                                carry += tl[q * 64 + lane];
                            } else {
                                const float tot = carry + hv[q * 64 + lane];
                                float* dstp = A.agg + (int64_t)df * H + 64 * fh + lane;
                                if (ext) atomicAdd(dstp, tot); else *dstp = tot;
                                carry = 0.f;
                                ext = false;
                            }
                        }
                        if (!through) {
                            const int dl = sd[q0 + 32];
                            if (dl >= 0 && sd[q0 + 33] == dl) { carry = tl[q * 64 + lane]; ext = false; }
                        }
                    }
                    const int dl = sd[TILE];
                    if (dl >= 0 && sd[TILE + 1] == dl)  // open at the tile end: the rest is in the next tile
                        atomicAdd(A.agg + (int64_t)dl * H + 64 * fh + lane, carry);
                }
            }
            // coalesced row stores: e_out = e' (+ e_in); the residual loads of all passes are in flight together
            {
                int orow[NPASS];
                floatx4 o[NPASS];
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int pr = p0 + pass * 16 + (tid >> 4);
                    const int prc = pr < E ? pr : E - 1;
                    orow[pass] = A.eid_out ? A.eid_out[prc] : prc;
                }
                if (with_resid) {
#pragma unroll
                    for (int pass = 0; pass < NPASS; ++pass)
                        o[pass] = *reinterpret_cast<const floatx4*>(A.e_in + (int64_t)orow[pass] * H + 64 * fh + c4);
                }
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int row = pass * 16 + (tid >> 4);
                    floatx4 v = *reinterpret_cast<const floatx4*>(T + row * TS + c4);
                    if (with_resid) v += o[pass];
                    if (p0 + row < E) *reinterpret_cast<floatx4*>(A.e_out + (int64_t)orow[pass] * H + 64 * fh + c4) = v;
                }
            }
            lds_barrier();  // T may be overwritten by the next chunk / tile
            GM_STAMP(4 + fh);
        }
        tpar ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// EDGE kernel on v_mfma_f32_16x16x4_f32 (the form production launches for H = 128).
//
// Why a second formulation: the 32x32x2 form above holds 32 edges x 128 features per wave (64 + 64
// registers of state), which leaves no room to prefetch, and the measured cost of that is large --
// the two workgroups of a CU run in phase, so their gather / epilogue phases (bound by the CU's own
// ~24 GB/s memory path) serialise with their MFMA phases instead of hiding under them
// (in-kernel stamps, round 1).  16x16x4 runs at the same FLOP rate with HALF the state per wave (16 edges:
// 32 + 32 registers), and the same register-chaining trick holds:
//     D block jb, lane (n = lane & 15, g = lane >> 4), register r  <->  feature 16 jb + 4 g + r
// is exactly the k this lane must supply as B operand in step r of k-block jb of the next layer.
// The freed registers hold (a) the tile's input rows e for the residual (no re-read), and (b) the
// NEXT tile's gathered operands, requested two loads per weight stage so that every wait in the MFMA
// phase finds its loads one stage old: the memory phase runs under the matrix pipe.
// ------------------------------------------------------------------------------------------
constexpr int T16 = 64;   // edges per workgroup tile (4 waves x 16)
constexpr int TS16 = 132; // LDS row stride (floats) of the H-wide staging tile
constexpr int NB16 = 8;   // 16-feature blocks in H = 128

template <int NB>
__device__ __forceinline__ void load_feat16(floatx4 (&v)[NB], const float* __restrict__ row, int g) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) v[kb] = *reinterpret_cast<const floatx4*>(row + 16 * kb + 4 * g);
}

// one weight stage (16 pieces) of an H x H layer: pieces [kq = 2 st + (0,1)][jb = 0..7]
template <int NKQ, int NJB>
__device__ __forceinline__ void stage16(floatx4 (&acc)[NJB], const floatx4 (&act)[NB16], const float* buf, int st) {
    constexpr int KQ_PER_STAGE = STAGE_PIECES / NJB;  // 2 for an H x H layer
    constexpr int NG = STAGE_PIECES / 2;              // groups of 2 pieces (two independent accumulators)
    floatx4 a_cur[2], a_nxt[2];
    a_cur[0] = *reinterpret_cast<const floatx4*>(buf);
    a_cur[1] = *reinterpret_cast<const floatx4*>(buf + PIECE_FLOATS);
#pragma unroll
    for (int gidx = 0; gidx < NG; ++gidx) {
        if (gidx + 1 < NG) {
            a_nxt[0] = *reinterpret_cast<const floatx4*>(buf + (2 * gidx + 2) * PIECE_FLOATS);
            a_nxt[1] = *reinterpret_cast<const floatx4*>(buf + (2 * gidx + 3) * PIECE_FLOATS);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int p = 2 * gidx + q;
                const int kq = st * KQ_PER_STAGE + p / NJB, jb = p % NJB;
                if (kq < NKQ) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[q][r], act[kq][r], acc[jb], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
    }
}

template <int NB>
__device__ __forceinline__ void layer_norm16(floatx4 (&acc)[NB], const float* lgamma, const float* lbeta, float eps, int g) {
    constexpr float INV_H = 1.0f / (16 * NB);
    float s = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[jb][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * INV_H;
    float q = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = acc[jb][r] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * INV_H + eps);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        const floatx4 gm = *reinterpret_cast<const floatx4*>(lgamma + 16 * jb + 4 * g);
        const floatx4 bt = *reinterpret_cast<const floatx4*>(lbeta + 16 * jb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][r] = (acc[jb][r] - mean) * rstd * gm[r] + bt[r];
    }
}

template <int NL, int MODE>
__global__ void __launch_bounds__(THREADS, 2) edge_kernel16(EdgeArgs A) {
    constexpr int H = 128;
    constexpr bool ENC = MODE == 0;
    constexpr bool with_resid = MODE == 1;
    constexpr int NB = NB16;
    constexpr int SL = 4;                                   // stages of an H x H layer (64 pieces)
    constexpr int TOTAL = ENC ? 1 + NL * SL : (NL + 1) * SL;
    constexpr int NBIAS = ENC ? NL + 1 : NL;
    constexpr int NCH = H / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    float* T = ring + 2 * STAGE_FLOATS;                                // [T16][TS16]
    int* sdst = reinterpret_cast<int*>(T + T16 * TS16);                // 2 x [T16 + 4]: per tile parity
    float* headv = reinterpret_cast<float*>(sdst + 2 * (T16 + 4));    // 2 tile parities x 2 halves x (head[4][64] | tail[4][64])
    float* vecs = headv + 2 * 1024;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int E = A.hdr ? A.hdr->n_edges : A.n_edges_host;
    const int ntiles = (E + T16 - 1) / T16;
    if ((int)blockIdx.x >= ntiles) return;

    WStream ws;
    ws.base = A.wstream;
    ws.ring = ring;
    ws.total = TOTAL;
    ws.cur = 0;
    ws.parity = 0;
    ws.lane = lane;
    ws.wave = wave;
    issue_stage(ws, 0, 0);
    for (int i = tid; i < NBIAS * H; i += THREADS) vecs[i] = A.bias[i];
    for (int i = tid; i < H; i += THREADS) {
        vecs[NBIAS * H + i] = A.ln_g[i];
        vecs[(NBIAS + 1) * H + i] = A.ln_b[i];
    }
    const float* lbias = vecs;
    const float* lgamma = vecs + NBIAS * H;
    const float* lbeta = lgamma + H;

    struct TileIdx { int er, d, sr, dq, sd; };
    auto fetch_idx = [&](int tile) {
        TileIdx ix;
        const int p0 = tile * T16;
        const int p = p0 + wave * 16 + n;
        const int pc = p < E ? p : E - 1;
        ix.er = A.eid ? A.eid[pc] : pc;
        ix.d = ix.sr = 0;
        ix.dq = ix.sd = -1;
        if (!ENC) {
            ix.d = A.dst[pc];
            ix.sr = A.src[pc];
            if (tid < T16 + 2) {
                const int pp = p0 - 1 + tid;
                ix.sd = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
            }
            if (lane < 18) {  // lane l: destination of row (16*wave - 1 + l) of the tile; -2 before edge 0, -1 past E
                const int pp = p0 + 16 * wave - 1 + lane;
                ix.dq = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
            }
        }
        return ix;
    };

    floatx4 acc[NB], act[NB], ekeep[NB], nacc[NB], nact[NB];
    // ---- first tile: operands requested directly
    TileIdx ix = fetch_idx(blockIdx.x);
    if (ENC) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * kb + 4 * g + r;
                nact[kb][r] = f < A.k1 ? A.e_in[(int64_t)ix.er * A.k1 + f] : 0.f;
            }
    } else {
        load_feat16(nact, A.e_in + (int64_t)ix.er * H, g);
        load_feat16(nacc, A.P + (int64_t)ix.d * (2 * H), g);
        floatx4 pj[NB];
        load_feat16(pj, A.P + (int64_t)ix.sr * (2 * H) + H, g);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) nacc[kb] += pj[kb];
    }
    TileIdx nx = ix;
    if ((int)(blockIdx.x + gridDim.x) < ntiles) nx = fetch_idx(blockIdx.x + gridDim.x);
    __syncthreads();  // vecs visible
    int tpar = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int p0 = tile * T16;
        // operands of this tile arrive from the prefetch registers; its indices from `ix`
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            act[kb] = nact[kb];
            if (with_resid) ekeep[kb] = nact[kb];
        }
        if (ENC) {
            load_feat16(acc, lbias, g);
        } else {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) acc[kb] = nacc[kb];
        }
        int* sd = sdst + tpar * (T16 + 4);
        if (!ENC && tid < T16 + 2) sd[tid] = ix.sd;  // ordered before its readers by the stage barriers
        const int dq = ix.dq;
        const int er_cur = ix.er;
        const TileIdx jx = nx;                        // indices of the NEXT tile (loaded one tile ago)
        if (more_tiles && tile + 2 * (int)gridDim.x < ntiles) nx = fetch_idx(tile + 2 * gridDim.x);
        floatx4 pj0, pj1;                             // rotating temporaries for the P_j rows of the next tile
        GM_STAMP(0);
        if (A.stamps && tid == 0) {
            A.stamps[(size_t)tile * 16 + 14] = ((unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) << 32) |
                                             __builtin_amdgcn_s_getreg((3 << 11) | 20);
            A.stamps[(size_t)tile * 16 + 15] = blockIdx.x;
        }
        prio_mfma_phase();
        // ---- the MLP: TOTAL weight stages; stage s also requests loads 2s, 2s+1 of the next tile
#pragma unroll
        for (int s = 0; s < TOTAL; ++s) {
            constexpr int L1S = ENC ? 1 : SL;                       // stages of layer 1
            const int layer = s < L1S ? 0 : 1 + (s - L1S) / SL;     // compile-time after unrolling
            const int st_in_layer = s < L1S ? s : (s - L1S) % SL;
            if (s >= L1S && st_in_layer == 0) {                     // layer boundary: ReLU -> B operand, bias -> accumulator
#pragma unroll
                for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[kb][r] = fmaxf(acc[kb][r], 0.f);
                load_feat16(acc, lbias + (ENC ? layer : layer - 1) * H, g);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            {
                int nxt = ws.cur + 1;
                const bool wrap = nxt == ws.total;
                if (wrap) nxt = 0;
                int stl = nxt;
                asm volatile("" : "+s"(stl));
                if (!wrap || more_tiles) issue_stage(ws, stl, ws.parity ^ 1);
            }
            // drip-fed prefetch of the next tile's operands (previous stage's loads have landed: the wait above)
            if (!ENC && more_tiles && !(A.debug & 8)) {
                if (s < 4) {
                    nact[2 * s] = *reinterpret_cast<const floatx4*>(A.e_in + (int64_t)jx.er * H + 16 * (2 * s) + 4 * g);
                    nact[2 * s + 1] = *reinterpret_cast<const floatx4*>(A.e_in + (int64_t)jx.er * H + 16 * (2 * s + 1) + 4 * g);
                } else if (s < 8) {
                    const int kb = 2 * (s - 4);
                    nacc[kb] = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.d * (2 * H) + 16 * kb + 4 * g);
                    nacc[kb + 1] = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.d * (2 * H) + 16 * (kb + 1) + 4 * g);
                } else if (s < 12) {
                    const int kb = 2 * (s - 8);
                    if (s > 8) {
                        nacc[kb - 2] += pj0;
                        nacc[kb - 1] += pj1;
                    }
                    pj0 = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.sr * (2 * H) + H + 16 * kb + 4 * g);
                    pj1 = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.sr * (2 * H) + H + 16 * (kb + 1) + 4 * g);
                }
            }
            if (ENC && more_tiles && s == 0) {
#pragma unroll
                for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * kb + 4 * g + r;
                        nact[kb][r] = f < A.k1 ? A.e_in[(int64_t)jx.er * A.k1 + f] : 0.f;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* buf = ws.ring + ws.parity * STAGE_FLOATS + lane * 4;
            if (ENC && s == 0) stage16<1, NB>(acc, act, buf, 0);
            else if (!(A.debug & 1)) stage16<H / 16, NB>(acc, act, buf, st_in_layer);
            ws.cur = ws.cur + 1 == ws.total ? 0 : ws.cur + 1;
            ws.parity ^= 1;
            if (s == L1S - 1) GM_STAMP(1);
        }
        GM_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        prio_latency_phase();
        if (A.debug & 2) {  // ablation: no epilogue (keep the result alive)
            if (acc[0][0] == 123.456f) A.e_out[p0] = acc[1][1] + acc[2][2] + acc[3][3];
            if (!ENC && more_tiles) { nacc[NB - 2] += pj0; nacc[NB - 1] += pj1; }
            ix = jx;
            tpar ^= 1;
            continue;
        }
        layer_norm16(acc, lgamma, lbeta, A.eps, g);
        GM_STAMP(3);

        // ---- e_out = e' (+ e) straight from the registers: 64 contiguous bytes per row and instruction
        const int p = p0 + wave * 16 + n;
        const bool valid = p < E;
        const int64_t out_row = !valid ? 0 : (!A.eid_out ? (int64_t)p : (A.eid_out == A.eid ? (int64_t)er_cur : (int64_t)A.eid_out[p]));
        if (valid && !(A.debug & 64)) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                floatx4 x = acc[kb];
                if (with_resid) x += ekeep[kb];
                *reinterpret_cast<floatx4*>(A.e_out + out_row * H + 16 * kb + 4 * g) = x;
            }
        }
        GM_STAMP(4);
        // ---- scatter-add.  e' (all H features of the 64 rows) is staged in LDS once; wave q sums the
        // destination segments inside its rows 16q..16q+15 (lane = column, two passes of 64 columns);
        // segments that cross a wave's rows or the tile are stitched from the per-wave head / tail
        // partials by wave 0 (columns 0..63) and wave 1 (columns 64..127) with scalar control flow.
        // Two LDS-only barriers per tile; the stitch is off the other waves' critical path.
        if (!ENC && !(A.debug & 4)) {
            float* part = headv + tpar * 1024;  // [2 halves][head 4x64 | tail 4x64], double-buffered per tile
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
                *reinterpret_cast<floatx4*>(T + (wave * 16 + n) * TS16 + 16 * kb + 4 * g) = acc[kb];
            lds_barrier();
            GM_STAMP(5);
            {   // one pass: lane owns columns 2*lane, 2*lane+1 (the scalar segment logic runs once per row)
                typedef float floatx2 __attribute__((ext_vector_type(2)));
                float* hv = part;         // head[4][128]
                float* tl = part + 512;   // tail[4][128]
                const int r0 = 16 * wave;
                floatx2 tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) tv[r] = *reinterpret_cast<const floatx2*>(T + (r0 + r) * TS16 + 2 * lane);
                floatx2 run = {0.f, 0.f};
                int d = __builtin_amdgcn_readlane(dq, 1);
                bool first = d >= 0 && __builtin_amdgcn_readlane(dq, 0) == d;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dn = __builtin_amdgcn_readlane(dq, r + 2);
                    if (d >= 0) run += tv[r];
                    if (dn != d && d >= 0) {
                        if (first) *reinterpret_cast<floatx2*>(hv + wave * 128 + 2 * lane) = run;
                        else *reinterpret_cast<floatx2*>(A.agg + (int64_t)d * H + 2 * lane) = run;
                        run = floatx2{0.f, 0.f};
                        first = false;
                    }
                    d = dn;
                }
                const int dl = __builtin_amdgcn_readlane(dq, 16);
                if (dl >= 0 && __builtin_amdgcn_readlane(dq, 17) == dl) *reinterpret_cast<floatx2*>(tl + wave * 128 + 2 * lane) = run;
            }
            lds_barrier();  // partials visible; staging tile free for the next tile
            GM_STAMP(6);
            if (wave < NCH) {
                const int fh = wave;
                const float* hv = part + 64 * fh;        // head[q][128], this wave's 64 columns
                const float* tl = part + 512 + 64 * fh;
                // boundary destinations of the four 16-row quarters, fetched once and read back as scalars:
                // lane 4q + j holds sd[16q + {0, 1, 16, 17}[j]]  (row 16q-1, 16q, 16q+15, 16q+16 of the tile)
                const int bl = lane & 15;
                const int bv = sd[16 * (bl >> 2) + ((bl & 3) < 2 ? (bl & 3) : 14 + (bl & 3))];
                float hq[4], tq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hq[q] = hv[q * 128 + lane];
                    tq[q] = tl[q * 128 + lane];
                }
                float carry = 0.f;
                bool ext = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int dprev = __builtin_amdgcn_readlane(bv, 4 * q);
                    const int df = __builtin_amdgcn_readlane(bv, 4 * q + 1);
                    const int dlast = __builtin_amdgcn_readlane(bv, 4 * q + 2);
                    const int dnext = __builtin_amdgcn_readlane(bv, 4 * q + 3);
                    const bool cont_in = df >= 0 && dprev == df;
                    const bool through = cont_in && dlast == df && dnext == df;
                    if (cont_in) {
                        if (q == 0) { carry = 0.f; ext = true; }
                        if (through) {
                            carry += tq[q];
                        } else {
                            const float tot = carry + hq[q];
                            float* dstp = A.agg + (int64_t)df * H + 64 * fh + lane;
                            if (ext) atomicAdd(dstp, tot); else *dstp = tot;
                            carry = 0.f;
                            ext = false;
                        }
                    }
                    if (!through && dlast >= 0 && dnext == dlast) { carry = tq[q]; ext = false; }
                    if (q == 3 && dlast >= 0 && dnext == dlast)  // open at the tile end: the rest is in the next tile
                        atomicAdd(A.agg + (int64_t)dlast * H + 64 * fh + lane, carry);
                }
            }
        }
        GM_STAMP(7);
        GM_STAMP(8); GM_STAMP(9); GM_STAMP(10); GM_STAMP(11);
        // last P_j pair of the next tile
        if (!ENC && more_tiles) {
            nacc[NB - 2] += pj0;
            nacc[NB - 1] += pj1;
        }
        ix = jx;
        tpar ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// EDGE kernel on the bf16 matrix pipe with fp32 accuracy (processor phi_e, H = 128).
//
// A bf16 x bf16 product is exact in fp32, so with x = x_hi + x_mid + x_lo, w = w_hi + w_mid + w_lo (three bf16 parts
// each, 24 mantissa bits) the six products  lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  accumulated in fp32
// reproduce an fp32 FMA chain to fp32 accuracy (measured through the whole model: 4.3e-7 against float64, plain
// float32 9.7e-7; tools/bf16_split_study.py).  v_mfma_f32_32x32x16_bf16 does 32768 flop in the time the fp32
// MFMA does 4096: six of them per fp32-equivalent product block is 2.7x the fp32 matrix rate (measured on an MLP chain:
// 358 vs 146 TFLOP/s, tools/bf16x6_chain.hip).
//
// Structure: 128-edge tiles, one workgroup of 8 waves per CU; wave (eh, fh) owns edges 32 eh .. +31 and output
// features 64 fh .. +63 (two accumulator blocks).  Weights are pre-split (pack_linear_b3) and streamed in 24 KiB
// stages = 2 k-groups of 16 x 4 output blocks x 3 parts; a layer's input lives as fp32 in the LDS tile X (each wave
// writes its ReLU'd half, the stage barrier publishes it) and every wave re-splits the 16 k-values it needs per
// k-group in registers.  K slot (lane >> 5, e) of k-group ks carries feature 16 ks + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
// in BOTH operands, which is exactly where the 32x32 accumulator layout keeps that feature: no cross-lane traffic.
// Gathers (drip-fed), LayerNorm (pair merge), stores and the scatter-add follow the other edge kernels.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2v __attribute__((ext_vector_type(2)));
constexpr int TSP = 132;                 // row stride (floats) of the X tile
constexpr int TE3 = 128;                 // edges per tile
constexpr int B3_THREADS = 512;
constexpr int B3_STAGE_BYTES = 24 * 1024;
constexpr int B3_STAGE_FLOATS = B3_STAGE_BYTES / 4;

__device__ __forceinline__ unsigned short bf16_rne_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_float(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// dst: [4 stages][2 ksl][4 jb][3 parts][64 lanes][8 bf16] of W[0:128, col0:col0+128] (row-major, leading dimension ld)
__global__ void __launch_bounds__(256) pack_linear_b3_kernel(const float* __restrict__ W, int ld, int col0, unsigned short* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one (piece-without-part, lane, e)
    if (idx >= 8 * 4 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, jb = (idx >> 9) & 3, ks = idx >> 11;
    const int i = lane & 31, kg = lane >> 5;
    const int k = 16 * ks + 8 * (e >> 2) + 4 * kg + (e & 3);
    const float w = W[(size_t)(32 * jb + i) * ld + col0 + k];
    const unsigned short hi = bf16_rne_bits(w);
    const float r1 = w - bf16_bits_to_float(hi);
    const unsigned short mid = bf16_rne_bits(r1);
    const unsigned short lo = bf16_rne_bits(r1 - bf16_bits_to_float(mid));
    const size_t piece0 = ((size_t)ks * 4 + jb) * 3;  // (stage = ks / 2, ksl = ks & 1) are contiguous in this order
    dst[(piece0 + 0) * 512 + lane * 8 + e] = hi;
    dst[(piece0 + 1) * 512 + lane * 8 + e] = mid;
    dst[(piece0 + 2) * 512 + lane * 8 + e] = lo;
}

int pack_linear_b3(const float* W, int ld, int col0, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(pack_linear_b3_kernel, dim3(8 * 4 * 64 * 8 / 256), dim3(256), 0, s, W, ld, col0, reinterpret_cast<unsigned short*>(dst));
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// three-way bf16 split of 8 floats (two float4 halves) -> B operand parts
__device__ __forceinline__ void b3_split(const floatx4& a, const floatx4& b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        floatx2v x;
        x[0] = e < 4 ? a[e] : b[e - 4];
        x[1] = e < 4 ? a[e + 1] : b[e - 3];
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const floatx2v r1 = x - __builtin_convertvector(h, floatx2v);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const floatx2v r2 = r1 - __builtin_convertvector(m, floatx2v);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        mid[e] = m[0]; mid[e + 1] = m[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}

template <int NL, int MODE>
__global__ void __launch_bounds__(B3_THREADS, 1) edge_kernel_b3(EdgeArgs A) {
    constexpr int H = 128;
    constexpr bool with_resid = MODE == 1;
    constexpr int SL = 4;                    // stages per layer (2 k-groups of 16 each)
    constexpr int TOTAL = (NL + 1) * SL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);                      // 2 x 24 KiB
    float* X = ring + 2 * B3_STAGE_FLOATS;                             // [TE3][TSP] fp32 layer input / staging tile
    int* sdst = reinterpret_cast<int*>(X + TE3 * TSP);                 // 2 x [TE3 + 4]
    float* headv = reinterpret_cast<float*>(sdst + 2 * (TE3 + 4));    // 2 parities x (head[8][128] | tail[8][128])
    float* vecs = headv + 2 * 2048;                                    // [NL][H] biases, gamma, beta
    float* lnx = vecs + (NL + 2) * H;                                  // [8 waves][32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int eh = wave >> 1, fh = wave & 1;
    const int E = A.hdr ? A.hdr->n_edges : A.n_edges_host;
    const int ntiles = (E + TE3 - 1) / TE3;
    if ((int)blockIdx.x >= ntiles) return;

    const float* wbase = A.wstream_b3;
    int ws_cur = 0, ws_par = 0;  // next stage to consume (0 .. TOTAL-1), ring buffer holding it
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int piece = c * 8 + wave;  // 24 pieces of 1 KiB
            const float* g = wbase + (size_t)stage * B3_STAGE_FLOATS + piece * 256 + lane * 4;
            float* l = ring + buf * B3_STAGE_FLOATS + piece * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    issue(0, 0);
    for (int i = tid; i < NL * H; i += B3_THREADS) vecs[i] = A.bias[i];
    for (int i = tid; i < H; i += B3_THREADS) {
        vecs[NL * H + i] = A.ln_g[i];
        vecs[(NL + 1) * H + i] = A.ln_b[i];
    }
    const float* lbias = vecs + 64 * fh;
    const float* lgamma = vecs + NL * H + 64 * fh;
    const float* lbeta = lgamma + H;
    float* xrow = X + (32 * eh + n) * TSP;

    struct TileIdx { int er, d, sr, dq, sd; };
    auto fetch_idx = [&](int tile) {
        TileIdx ix;
        const int p0 = tile * TE3;
        const int p = p0 + 32 * eh + n;
        const int pc = p < E ? p : E - 1;
        ix.er = A.eid ? A.eid[pc] : pc;
        ix.d = A.dst[pc];
        ix.sr = A.src[pc];
        ix.dq = ix.sd = -1;
        if (tid < TE3 + 2) {
            const int pp = p0 - 1 + tid;
            ix.sd = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
        }
        if (lane < 18) {  // lane l: destination of row (16*wave - 1 + l) of the tile; -2 before edge 0, -1 past E
            const int pp = p0 + 16 * wave - 1 + lane;
            ix.dq = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
        }
        return ix;
    };

    floatx16 acc[2], ekeep[2], nkeep[2], nacc[2];
    TileIdx ix = fetch_idx(blockIdx.x);
    load_feat(nkeep, A.e_in + (int64_t)ix.er * H + 64 * fh, hi);
    load_feat(nacc, A.P + (int64_t)ix.d * (2 * H) + 64 * fh, hi);
    add_feat(nacc, A.P + (int64_t)ix.sr * (2 * H) + H + 64 * fh, hi);
    TileIdx nx = ix;
    if ((int)(blockIdx.x + gridDim.x) < ntiles) nx = fetch_idx(blockIdx.x + gridDim.x);
    __syncthreads();
    int tpar = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int p0 = tile * TE3;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            ekeep[jb] = nkeep[jb];
            acc[jb] = nacc[jb];
        }
        store_feat(ekeep, xrow + 64 * fh, hi);
        int* sd = sdst + tpar * (TE3 + 4);
        if (tid < TE3 + 2) sd[tid] = ix.sd;
        const int dq = ix.dq;
        const int er_cur = ix.er;
        const TileIdx jx = nx;
        if (more_tiles && tile + 2 * (int)gridDim.x < ntiles) nx = fetch_idx(tile + 2 * gridDim.x);
        floatx4 pj[4];  // P_j rows of the next tile: requested at stage s, added at stage s + 2 (the wait above covers them)
        floatx4 xf[4], xn[4];  // fp32 k-values of this stage's two k-groups (two float4 each), and the next stage's
#pragma unroll
        for (int s = 0; s < TOTAL; ++s) {
            const int layer = s / SL, st = s % SL;
            if (st == 0 && layer > 0) {
                floatx16 r[2];
                relu_to(r, acc);
                store_feat(r, xrow + 64 * fh, hi);
                load_feat(acc, lbias + (layer - 1) * H, hi);
            }
            // This stage's DMA was issued one stage ago BEFORE that stage's two gathers (sched_barrier below): waiting for
            // "at most two outstanding" proves it has landed and leaves the gathers another stage to arrive (a stage of
            // this kernel is shorter than an HBM gather).
            if (more_tiles && s > 0) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                lds_barrier();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            {
                int nxt = ws_cur + 1;
                const bool wrap = nxt == TOTAL;
                if (wrap) nxt = 0;
                int stl = nxt;
                asm volatile("" : "+s"(stl));
                if (!wrap || more_tiles) issue(stl, ws_par ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more_tiles) {  // drip-fed prefetch of the next tile's operands: 2 float4 per stage (8 e, 8 P_i, 8 P_j)
                const int q = 2 * (s % 4);
                if (s < 4) {
                    const float* src = A.e_in + (int64_t)jx.er * H + 64 * fh + 4 * hi;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f4 = q + u;
                        const floatx4 x = *reinterpret_cast<const floatx4*>(src + 32 * (f4 >> 2) + 8 * (f4 & 3));
#pragma unroll
                        for (int t = 0; t < 4; ++t) nkeep[f4 >> 2][4 * (f4 & 3) + t] = x[t];
                    }
                } else if (s < 8) {
                    const float* src = A.P + (int64_t)jx.d * (2 * H) + 64 * fh + 4 * hi;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f4 = q + u;
                        const floatx4 x = *reinterpret_cast<const floatx4*>(src + 32 * (f4 >> 2) + 8 * (f4 & 3));
#pragma unroll
                        for (int t = 0; t < 4; ++t) nacc[f4 >> 2][4 * (f4 & 3) + t] = x[t];
                    }
                } else {
                    const float* src = A.P + (int64_t)jx.sr * (2 * H) + H + 64 * fh + 4 * hi;
                    if (s > 9) {  // the pair requested two stages ago
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int f4 = q - 4 + u;
#pragma unroll
                            for (int t = 0; t < 4; ++t) nacc[f4 >> 2][4 * (f4 & 3) + t] += pj[2 * (s & 1) + u][t];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f4 = q + u;
                        pj[2 * (s & 1) + u] = *reinterpret_cast<const floatx4*>(src + 32 * (f4 >> 2) + 8 * (f4 & 3));
                    }
                }
            }
            // fp32 inputs of k-groups ks = 2 st, 2 st + 1: features 16 ks + {4 hi .. +3} and 16 ks + 8 + {4 hi .. +3}
            if (st == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xf[c] = *reinterpret_cast<const floatx4*>(xrow + 16 * (c >> 1) + 8 * (c & 1) + 4 * hi);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) xf[c] = xn[c];
            }
            if (st + 1 < SL) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xn[c] = *reinterpret_cast<const floatx4*>(xrow + 32 * (st + 1) + 16 * (c >> 1) + 8 * (c & 1) + 4 * hi);
            }
            const bf16x8* wst = reinterpret_cast<const bf16x8*>(ring + ws_par * B3_STAGE_FLOATS) + lane;
#pragma unroll
            for (int ksl = 0; ksl < 2; ++ksl) {
                bf16x8 bh, bm, bl;
                b3_split(xf[2 * ksl], xf[2 * ksl + 1], bh, bm, bl);
#pragma unroll
                for (int jbl = 0; jbl < 2; ++jbl) {
                    const bf16x8* pw = wst + ((ksl * 4 + 2 * fh + jbl) * 3) * 64;  // piece stride: 64 lanes x 16 B
                    const bf16x8 ah = pw[0], am = pw[64], al = pw[128];
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[jbl], 0, 0, 0);
                }
            }
            ws_cur = ws_cur + 1 == TOTAL ? 0 : ws_cur + 1;
            ws_par ^= 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LayerNorm (this wave: 64 of the 128 features of its 32 edges; pair merge through LDS)
        {
            float sm = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sm += acc[jb][r];
            sm += __shfl_xor(sm, 32, 64);
            const float mh = sm * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[jb][r] - mh;
                    m2 += d * d;
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (hi == 0) {
                lnx[(wave * 32 + n) * 2] = mh;
                lnx[(wave * 32 + n) * 2 + 1] = m2;
            }
            lds_barrier();
            const float mo = lnx[((wave ^ 1) * 32 + n) * 2], m2o = lnx[((wave ^ 1) * 32 + n) * 2 + 1];
            const float mean = 0.5f * (mh + mo);
            const float dlt = mh - mo;
            const float var = (m2 + m2o + 32.0f * dlt * dlt) * (1.0f / 128.0f);
            const float rstd = 1.0f / sqrtf(var + A.eps);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 gm = *reinterpret_cast<const floatx4*>(lgamma + 32 * jb + 8 * g + 4 * hi);
                    const floatx4 bt = *reinterpret_cast<const floatx4*>(lbeta + 32 * jb + 8 * g + 4 * hi);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[jb][4 * g + t] = (acc[jb][4 * g + t] - mean) * rstd * gm[t] + bt[t];
                }
            }
        }
        const int p = p0 + 32 * eh + n;
        const bool valid = p < E;
        const int64_t out_row = !valid ? 0 : (!A.eid_out ? (int64_t)p : (A.eid_out == A.eid ? (int64_t)er_cur : (int64_t)A.eid_out[p]));
        if (valid) {
            floatx16 o[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                o[jb] = acc[jb];
                if (with_resid) o[jb] += ekeep[jb];
            }
            store_feat(o, A.e_out + out_row * H + 64 * fh, hi);
        }
        // ---- scatter-add: e' staged in X; wave q sums the destination segments inside rows 16q..16q+15 (8 waves = 128
        // rows); waves 0 / 1 stitch the 8 quarters (64 columns each)
        {
            float* part = headv + tpar * 2048;
            store_feat(acc, xrow + 64 * fh, hi);
            lds_barrier();
            {
                typedef float floatx2 __attribute__((ext_vector_type(2)));
                float* hv = part;          // head[8][128]
                float* tl = part + 1024;   // tail[8][128]
                const int r0 = 16 * wave;
                floatx2 tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) tv[r] = *reinterpret_cast<const floatx2*>(X + (r0 + r) * TSP + 2 * lane);
                floatx2 run = {0.f, 0.f};
                int d = __builtin_amdgcn_readlane(dq, 1);
                bool first = d >= 0 && __builtin_amdgcn_readlane(dq, 0) == d;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dn = __builtin_amdgcn_readlane(dq, r + 2);
                    if (d >= 0) run += tv[r];
                    if (dn != d && d >= 0) {
                        if (first) *reinterpret_cast<floatx2*>(hv + wave * 128 + 2 * lane) = run;
                        else *reinterpret_cast<floatx2*>(A.agg + (int64_t)d * H + 2 * lane) = run;
                        run = floatx2{0.f, 0.f};
                        first = false;
                    }
                    d = dn;
                }
                const int dl = __builtin_amdgcn_readlane(dq, 16);
                if (dl >= 0 && __builtin_amdgcn_readlane(dq, 17) == dl) *reinterpret_cast<floatx2*>(tl + wave * 128 + 2 * lane) = run;
            }
            lds_barrier();
            if (wave < 2) {
                const int fc = wave;
                const float* hv = part + 64 * fc;
                const float* tl = part + 1024 + 64 * fc;
                const int bl = lane & 31;  // lane 4q + j holds sd[16q + {0, 1, 16, 17}[j]], q = 0..7
                const int bv = sd[16 * (bl >> 2) + ((bl & 3) < 2 ? (bl & 3) : 14 + (bl & 3))];
                float carry = 0.f;
                bool ext = false;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float hq = hv[q * 128 + lane], tq = tl[q * 128 + lane];
                    const int dprev = __builtin_amdgcn_readlane(bv, 4 * q);
                    const int df = __builtin_amdgcn_readlane(bv, 4 * q + 1);
                    const int dlast = __builtin_amdgcn_readlane(bv, 4 * q + 2);
                    const int dnext = __builtin_amdgcn_readlane(bv, 4 * q + 3);
                    const bool cont_in = df >= 0 && dprev == df;
                    const bool through = cont_in && dlast == df && dnext == df;
                    if (cont_in) {
                        if (q == 0) { carry = 0.f; ext = true; }
                        if (through) {
                            carry += tq;
                        } else {
                            const float tot = carry + hq;
                            float* dstp = A.agg + (int64_t)df * H + 64 * fc + lane;
                            if (ext) atomicAdd(dstp, tot); else *dstp = tot;
                            carry = 0.f;
                            ext = false;
                        }
                    }
                    if (!through && dlast >= 0 && dnext == dlast) { carry = tq; ext = false; }
                    if (q == 7 && dlast >= 0 && dnext == dlast)
                        atomicAdd(A.agg + (int64_t)dlast * H + 64 * fc + lane, carry);
                }
            }
        }
        if (more_tiles) {  // the last two P_j pairs (requested at stages 10 and 11: f4 = 4, 5 and 6, 7)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    nacc[1][4 * u + t] += pj[u][t];
                    nacc[1][4 * (2 + u) + t] += pj[2 + u][t];
                }
        }
        ix = jx;
        tpar ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// The same bf16-pipe scheme on 64-edge tiles: 4 waves (edge half x feature half), TWO workgroups per CU, weight
// stages of one k-group (12 KiB = 4 output blocks x 3 parts), 24 stages per tile.  Twice the rendezvous count of
// edge_kernel_b3, but the second workgroup runs under the first one's epilogue and gather latency again.
// ------------------------------------------------------------------------------------------
constexpr int B3P_STAGE_FLOATS = 3072;  // 12 KiB

template <int NL, int MODE>
__global__ void __launch_bounds__(THREADS, 2) edge_kernel_b3p(EdgeArgs A) {
    constexpr int H = 128;
    constexpr bool with_resid = MODE == 1;
    constexpr int SL = 8;                    // stages (k-groups of 16) per layer
    constexpr int TOTAL = (NL + 1) * SL;     // 24
    constexpr int NCH = H / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);                     // 2 x 12 KiB
    float* X = ring + 2 * B3P_STAGE_FLOATS;                           // [T16][TSP]
    int* sdst = reinterpret_cast<int*>(X + T16 * TSP);                // 2 x [T16 + 4]
    float* headv = reinterpret_cast<float*>(sdst + 2 * (T16 + 4));   // 2 parities x (head[4][128] | tail[4][128])
    float* vecs = headv + 2 * 1024;
    float* lnx = vecs + (NL + 2) * H;                                 // [4 waves][32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int eh = wave >> 1, fh = wave & 1;
    const int E = A.hdr ? A.hdr->n_edges : A.n_edges_host;
    const int ntiles = (E + T16 - 1) / T16;
    if ((int)blockIdx.x >= ntiles) return;

    const float* wbase = A.wstream_b3;
    int ws_cur = 0, ws_par = 0;
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int piece = c * 4 + wave;  // 12 pieces of 1 KiB
            const float* g = wbase + (size_t)stage * B3P_STAGE_FLOATS + piece * 256 + lane * 4;
            float* l = ring + buf * B3P_STAGE_FLOATS + piece * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    issue(0, 0);
    for (int i = tid; i < NL * H; i += THREADS) vecs[i] = A.bias[i];
    for (int i = tid; i < H; i += THREADS) {
        vecs[NL * H + i] = A.ln_g[i];
        vecs[(NL + 1) * H + i] = A.ln_b[i];
    }
    const float* lbias = vecs + 64 * fh;
    const float* lgamma = vecs + NL * H + 64 * fh;
    const float* lbeta = lgamma + H;
    float* xrow = X + (32 * eh + n) * TSP;

    struct TileIdx { int er, d, sr, dq, sd; };
    auto fetch_idx = [&](int tile) {
        TileIdx ix;
        const int p0 = tile * T16;
        const int p = p0 + 32 * eh + n;
        const int pc = p < E ? p : E - 1;
        ix.er = A.eid ? A.eid[pc] : pc;
        ix.d = A.dst[pc];
        ix.sr = A.src[pc];
        ix.dq = ix.sd = -1;
        if (tid < T16 + 2) {
            const int pp = p0 - 1 + tid;
            ix.sd = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
        }
        if (lane < 18) {
            const int pp = p0 + 16 * wave - 1 + lane;
            ix.dq = pp < 0 ? -2 : (pp < E ? A.dst[pp] : -1);
        }
        return ix;
    };

    floatx16 acc[2], ekeep[2], nkeep[2], nacc[2];
    TileIdx ix = fetch_idx(blockIdx.x);
    load_feat(nkeep, A.e_in + (int64_t)ix.er * H + 64 * fh, hi);
    load_feat(nacc, A.P + (int64_t)ix.d * (2 * H) + 64 * fh, hi);
    add_feat(nacc, A.P + (int64_t)ix.sr * (2 * H) + H + 64 * fh, hi);
    TileIdx nx = ix;
    if ((int)(blockIdx.x + gridDim.x) < ntiles) nx = fetch_idx(blockIdx.x + gridDim.x);
    __syncthreads();
    int tpar = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int p0 = tile * T16;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            ekeep[jb] = nkeep[jb];
            acc[jb] = nacc[jb];
        }
        store_feat(ekeep, xrow + 64 * fh, hi);
        int* sd = sdst + tpar * (T16 + 4);
        if (tid < T16 + 2) sd[tid] = ix.sd;
        const int dq = ix.dq;
        const int er_cur = ix.er;
        const TileIdx jx = nx;
        if (more_tiles && tile + 2 * (int)gridDim.x < ntiles) nx = fetch_idx(tile + 2 * gridDim.x);
        floatx4 pj[3];         // P_j float4s of the next tile: requested at stage s, added at stage s + 2
        floatx4 xf[2], xn[2];  // fp32 k-values of this stage's k-group (two float4), and the next stage's
        prio_mfma_phase();
#pragma unroll
        for (int s = 0; s < TOTAL; ++s) {
            const int layer = s / SL, ks = s % SL;
            if (ks == 0 && layer > 0) {
                floatx16 r[2];
                relu_to(r, acc);
                store_feat(r, xrow + 64 * fh, hi);
                load_feat(acc, lbias + (layer - 1) * H, hi);
            }
            // the stage's DMA is older than the one gather issued behind it a stage ago: counted wait
            if (more_tiles && s > 0) {
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                lds_barrier();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            {
                int nxt = ws_cur + 1;
                const bool wrap = nxt == TOTAL;
                if (wrap) nxt = 0;
                int stl = nxt;
                asm volatile("" : "+s"(stl));
                if (!wrap || more_tiles) issue(stl, ws_par ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more_tiles) {  // one float4 of the next tile's operands per stage: 8 e, 8 P_i, 8 P_j
                const int f4 = s % 8;
                const int off = 32 * (f4 >> 2) + 8 * (f4 & 3) + 64 * fh + 4 * hi;
                if (s < 8) {
                    const floatx4 x = *reinterpret_cast<const floatx4*>(A.e_in + (int64_t)jx.er * H + off);
#pragma unroll
                    for (int t = 0; t < 4; ++t) nkeep[f4 >> 2][4 * (f4 & 3) + t] = x[t];
                } else if (s < 16) {
                    const floatx4 x = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.d * (2 * H) + off);
#pragma unroll
                    for (int t = 0; t < 4; ++t) nacc[f4 >> 2][4 * (f4 & 3) + t] = x[t];
                } else {
                    if (s >= 18) {
                        const int g4 = f4 - 2;
#pragma unroll
                        for (int t = 0; t < 4; ++t) nacc[g4 >> 2][4 * (g4 & 3) + t] += pj[(s - 2) % 3][t];
                    }
                    pj[s % 3] = *reinterpret_cast<const floatx4*>(A.P + (int64_t)jx.sr * (2 * H) + H + off);
                }
            }
            if (ks == 0) {
                xf[0] = *reinterpret_cast<const floatx4*>(xrow + 4 * hi);
                xf[1] = *reinterpret_cast<const floatx4*>(xrow + 8 + 4 * hi);
            } else {
                xf[0] = xn[0];
                xf[1] = xn[1];
            }
            if (ks + 1 < SL) {
                xn[0] = *reinterpret_cast<const floatx4*>(xrow + 16 * (ks + 1) + 4 * hi);
                xn[1] = *reinterpret_cast<const floatx4*>(xrow + 16 * (ks + 1) + 8 + 4 * hi);
            }
            const bf16x8* wst = reinterpret_cast<const bf16x8*>(ring + ws_par * B3P_STAGE_FLOATS) + lane;
            {
                bf16x8 bh, bm, bl;
                b3_split(xf[0], xf[1], bh, bm, bl);
#pragma unroll
                for (int jbl = 0; jbl < 2; ++jbl) {
                    const bf16x8* pw = wst + ((2 * fh + jbl) * 3) * 64;
                    const bf16x8 ah = pw[0], am = pw[64], al = pw[128];
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[jbl], 0, 0, 0);
                    acc[jbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[jbl], 0, 0, 0);
                }
            }
            ws_cur = ws_cur + 1 == TOTAL ? 0 : ws_cur + 1;
            ws_par ^= 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        prio_latency_phase();
        {
            float sm = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sm += acc[jb][r];
            sm += __shfl_xor(sm, 32, 64);
            const float mh = sm * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[jb][r] - mh;
                    m2 += d * d;
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (hi == 0) {
                lnx[(wave * 32 + n) * 2] = mh;
                lnx[(wave * 32 + n) * 2 + 1] = m2;
            }
            lds_barrier();
            const float mo = lnx[((wave ^ 1) * 32 + n) * 2], m2o = lnx[((wave ^ 1) * 32 + n) * 2 + 1];
            const float mean = 0.5f * (mh + mo);
            const float dlt = mh - mo;
            const float var = (m2 + m2o + 32.0f * dlt * dlt) * (1.0f / 128.0f);
            const float rstd = 1.0f / sqrtf(var + A.eps);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 gm = *reinterpret_cast<const floatx4*>(lgamma + 32 * jb + 8 * g + 4 * hi);
                    const floatx4 bt = *reinterpret_cast<const floatx4*>(lbeta + 32 * jb + 8 * g + 4 * hi);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[jb][4 * g + t] = (acc[jb][4 * g + t] - mean) * rstd * gm[t] + bt[t];
                }
            }
        }
        const int p = p0 + 32 * eh + n;
        const bool valid = p < E;
        const int64_t out_row = !valid ? 0 : (!A.eid_out ? (int64_t)p : (A.eid_out == A.eid ? (int64_t)er_cur : (int64_t)A.eid_out[p]));
        if (valid) {
            floatx16 o[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                o[jb] = acc[jb];
                if (with_resid) o[jb] += ekeep[jb];
            }
            store_feat(o, A.e_out + out_row * H + 64 * fh, hi);
        }
        {
            float* part = headv + tpar * 1024;
            store_feat(acc, xrow + 64 * fh, hi);
            lds_barrier();
            {
                typedef float floatx2 __attribute__((ext_vector_type(2)));
                float* hv = part;
                float* tl = part + 512;
                const int r0 = 16 * wave;
                floatx2 tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) tv[r] = *reinterpret_cast<const floatx2*>(X + (r0 + r) * TSP + 2 * lane);
                floatx2 run = {0.f, 0.f};
                int d = __builtin_amdgcn_readlane(dq, 1);
                bool first = d >= 0 && __builtin_amdgcn_readlane(dq, 0) == d;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dn = __builtin_amdgcn_readlane(dq, r + 2);
                    if (d >= 0) run += tv[r];
                    if (dn != d && d >= 0) {
                        if (first) *reinterpret_cast<floatx2*>(hv + wave * 128 + 2 * lane) = run;
                        else *reinterpret_cast<floatx2*>(A.agg + (int64_t)d * H + 2 * lane) = run;
                        run = floatx2{0.f, 0.f};
                        first = false;
                    }
                    d = dn;
                }
                const int dl = __builtin_amdgcn_readlane(dq, 16);
                if (dl >= 0 && __builtin_amdgcn_readlane(dq, 17) == dl) *reinterpret_cast<floatx2*>(tl + wave * 128 + 2 * lane) = run;
            }
            lds_barrier();
            if (wave < NCH) {
                const int fc = wave;
                const float* hv = part + 64 * fc;
                const float* tl = part + 512 + 64 * fc;
                const int bl = lane & 15;
                const int bv = sd[16 * (bl >> 2) + ((bl & 3) < 2 ? (bl & 3) : 14 + (bl & 3))];
                float carry = 0.f;
                bool ext = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float hq = hv[q * 128 + lane], tq = tl[q * 128 + lane];
                    const int dprev = __builtin_amdgcn_readlane(bv, 4 * q);
                    const int df = __builtin_amdgcn_readlane(bv, 4 * q + 1);
                    const int dlast = __builtin_amdgcn_readlane(bv, 4 * q + 2);
                    const int dnext = __builtin_amdgcn_readlane(bv, 4 * q + 3);
                    const bool cont_in = df >= 0 && dprev == df;
                    const bool through = cont_in && dlast == df && dnext == df;
                    if (cont_in) {
                        if (q == 0) { carry = 0.f; ext = true; }
                        if (through) {
                            carry += tq;
                        } else {
                            const float tot = carry + hq;
                            float* dstp = A.agg + (int64_t)df * H + 64 * fc + lane;
                            if (ext) atomicAdd(dstp, tot); else *dstp = tot;
                            carry = 0.f;
                            ext = false;
                        }
                    }
                    if (!through && dlast >= 0 && dnext == dlast) { carry = tq; ext = false; }
                    if (q == 3 && dlast >= 0 && dnext == dlast)
                        atomicAdd(A.agg + (int64_t)dlast * H + 64 * fc + lane, carry);
                }
            }
        }
        if (more_tiles) {  // the last two P_j float4s (requested at stages 22, 23: f4 = 6, 7)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                nacc[1][4 * 2 + t] += pj[22 % 3][t];
                nacc[1][4 * 3 + t] += pj[23 % 3][t];
            }
        }
        ix = jx;
        tpar ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// NODE kernel.  MODE 0: encoder MLP on raw node features; 1: processor phi_v on [h | agg];
// 2: projection only (block API).  Tail (runtime, uniform): 0 none, 1 projection P = h'[W_i|W_j]^T
// for the next edge step, 2 decoder.
// ------------------------------------------------------------------------------------------
template <int H, int NL, int MODE>
__global__ void __launch_bounds__(THREADS, H <= 128 ? 2 : 1) node_kernel(NodeArgs A) {
    constexpr int NJB = H / 32;
    constexpr int SL = (H / 8) * NJB / STAGE_PIECES;  // stages of one HxH layer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int N = A.n_nodes;
    const int ntiles = (N + TILE - 1) / TILE;

    WStream ws;
    ws.base = A.wstream;
    ws.ring = ring;
    constexpr int S_IN = (4 * NJB + STAGE_PIECES - 1) / STAGE_PIECES;       // encoder layer 1: K padded to 32
    constexpr int S_OUT = (H / 8 + STAGE_PIECES - 1) / STAGE_PIECES;        // decoder output layer: one 32-row block
    const int mlp_stages = MODE == 0 ? (S_IN + NL * SL) : (MODE == 1 ? (NL + 2) * SL : 0);
    const int tail_stages = A.tail == 1 ? 2 * SL : (A.tail == 2 ? NL * SL + S_OUT : 0);
    ws.total = mlp_stages + tail_stages;
    ws.cur = 0;
    ws.parity = 0;
    ws.lane = lane;
    ws.wave = wave;
    if ((int)blockIdx.x < ntiles) issue_stage(ws, 0, 0);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int p = tile * TILE + wave * 32 + n;
        const bool valid = p < N;
        const int64_t pc = valid ? p : N - 1;
        floatx16 acc[NJB], act[NJB];
        if (MODE == 0) {
            load_feat_guard(act, A.x_in + pc * A.k1, hi, A.k1);
            load_feat(acc, A.bias, hi);
            run_layer<4, NJB, NJB>(acc, act, ws, more_tiles);  // K = node_dim padded to 32
            mlp_tail_layers<H, NL>(acc, act, A.bias + H, ws, more_tiles, hi);
            layer_norm_regs(acc, A.ln_g, A.ln_b, A.eps, hi);
        } else if (MODE == 1) {
            load_feat(act, A.x_in + pc * H, hi);
            load_feat(acc, A.bias, hi);
            run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);  // W_h h
            load_feat(act, A.agg + pc * H, hi);
            if (A.agg_clear && valid) {  // the rows just read become the zeroed target of the next scatter-add
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NJB; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<floatx4*>(A.agg_clear + pc * H + 32 * kb + 8 * g + 4 * hi) = z;
            }
            run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);  // + W_agg agg
            mlp_tail_layers<H, NL>(acc, act, A.bias + H, ws, more_tiles, hi);
            layer_norm_regs(acc, A.ln_g, A.ln_b, A.eps, hi);
            if (A.residual) add_feat(acc, A.x_in + pc * H, hi);  // h <- h' + h (epd_gnn.py:103)
        } else {
            load_feat(acc, A.x_in + pc * H, hi);
        }
        if (MODE != 2 && valid) store_feat(acc, A.h_out + pc * H, hi);

        if (A.tail == 1) {
            // P_i = h' W_i^T + b1_edge ; P_j = h' W_j^T   (layer-1 factorisation of the next edge MLP)
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
            load_feat(acc, A.proj_bias, hi);
            run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);
            if (valid) store_feat(acc, A.P_out + pc * (2 * H), hi);
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jb][r] = 0.f;
            run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);
            if (valid) store_feat(acc, A.P_out + pc * (2 * H) + H, hi);
        } else if (A.tail == 2) {
            // decoder (epd_gnn.py:49,96): Linear ReLU [Linear ReLU]x(NL-1) Linear(H -> out_dim), no LayerNorm
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
            load_feat(acc, A.dec_bias, hi);
            run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                relu_to(act, acc);
                load_feat(acc, A.dec_bias + l * H, hi);
                run_layer<H / 8, NJB, NJB>(acc, act, ws, more_tiles);
            }
            relu_to(act, acc);
            floatx16 o[1];
            load_feat(o, A.dec_bias + NL * H, hi);  // out bias, zero-padded to 32
            run_layer<H / 8, 1, NJB>(o, act, ws, more_tiles);
            if (valid && hi == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < A.out_dim) A.dec_out[pc * A.out_dim + c] = o[0][c];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// NODE kernel, wide form for small graphs.  The form above gives one wave 32 nodes x all H features:
// 6 layer-units = 1536 dependent-chain MFMAs per wave, ~41 us however few nodes there are, and at
// N = 5k only 40 workgroups exist for 256 CUs.  Here a 4-wave workgroup owns 32 nodes and wave w
// computes feature block jb = w of every layer (64 instead of 256 MFMAs per layer); between layers the
// four blocks are exchanged through LDS ([32 nodes][H] tile) and every wave re-reads the full vector as
// its next B operand.  Same packed weight stream (every wave reads its jb piece of each k-octet),
// 4-deep ring so that a whole layer is in flight.  Used when it gives more workgroups than CUs can
// otherwise be offered (launch_node).
// ------------------------------------------------------------------------------------------
constexpr int WTILE = 32;
constexpr int WRING = 4;
constexpr int XS = 132;  // LDS row stride of the exchange tile

template <int H>
struct WideCtx {
    const float* base;   // packed stream, stage 0
    float* ring;         // WRING stages
    int total;           // stages per tile
    int cur;             // next stage to consume (within the tile sequence)
    int slot;            // ring slot of `cur`
    int issued;          // stages issued ahead of `cur` (<= WRING - 1)
    int lane, wave;
    bool more_tiles;
};

template <int H>
__device__ __forceinline__ void wide_issue(const WideCtx<H>& c, int stage, int slot) {
#pragma unroll
    for (int q = 0; q < STAGE_PIECES / 4; ++q) {
        const int piece = q * 4 + c.wave;
        const float* g = c.base + (size_t)stage * STAGE_FLOATS + piece * PIECE_FLOATS + c.lane * 4;
        float* l = c.ring + slot * STAGE_FLOATS + piece * PIECE_FLOATS;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
}

// One Linear for this wave's 32-feature block.  NKQ input octets; NJB_L = output blocks of the LAYER
// (4 for an HxH layer: wave w takes block w; 1 for the decoder output: wave 0 only).
// DRAIN: other vector-memory operations may be in flight -> wait for everything at the first stage;
// otherwise a counted wait leaves the younger DMA stages in flight.
template <int H, int NKQ, int NJB_L, bool DRAIN>
__device__ __forceinline__ void wide_layer(floatx16& acc, const floatx16 (&act)[H / 32], WideCtx<H>& c) {
    constexpr int NP = NKQ * NJB_L;
    constexpr int NST = (NP + STAGE_PIECES - 1) / STAGE_PIECES;
    constexpr int KQ_PER_STAGE = STAGE_PIECES / NJB_L;
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        // stage `cur` landed?  younger stages (4 DMA instructions each) may stay in flight
        if (DRAIN && s == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (c.issued >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (c.issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();  // stage `cur` visible to every wave; the slot consumed before it is free
        const float* buf = c.ring + c.slot * STAGE_FLOATS + c.lane * 4;
        const bool mine = NJB_L > 1 || c.wave == 0;
        floatx4 a[KQ_PER_STAGE > 4 ? 4 : KQ_PER_STAGE];
        // advance the stream state, then top the ring up (`issued` = stages requested and not yet consumed)
        c.issued--;
        c.cur++;
        if (c.cur == c.total) c.cur = 0;
        c.slot = (c.slot + 1) % WRING;
        {
            int st = c.cur + c.issued;
            const bool wrap = st >= c.total;
            if (wrap) st -= c.total;
            // stages past the end of the sequence (and everything requested right after the sequence wrapped)
            // belong to the workgroup's next tile: request them only if there is one
            const bool next_tile_stage = wrap || c.cur == 0;
            if (!next_tile_stage || c.more_tiles) {
                int ss = st;
                asm volatile("" : "+s"(ss));
                wide_issue(c, ss, (c.slot + c.issued) % WRING);
                c.issued++;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mine) {
            if (NJB_L > 1) {
#pragma unroll
                for (int kl = 0; kl < KQ_PER_STAGE; ++kl) {
                    const int kq = s * KQ_PER_STAGE + kl;
                    if (kq < NKQ) a[kl] = *reinterpret_cast<const floatx4*>(buf + (kl * NJB_L + c.wave) * PIECE_FLOATS);
                }
#pragma unroll
                for (int kl = 0; kl < KQ_PER_STAGE; ++kl) {
                    const int kq = s * KQ_PER_STAGE + kl;
                    if (kq < NKQ) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kl][t], act[kq >> 2][(kq & 3) * 4 + t], acc, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int kl = 0; kl < KQ_PER_STAGE; ++kl) {
                    const int kq = s * KQ_PER_STAGE + kl;
                    if (kq < NKQ) {
                        const floatx4 av = *reinterpret_cast<const floatx4*>(buf + kl * PIECE_FLOATS);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], act[kq >> 2][(kq & 3) * 4 + t], acc, 0, 0, 0);
                    }
                }
            }
        }
    }
}

// this wave's 16 registers <-> features 32*wave + 8g + 4hi + t of a row
__device__ __forceinline__ void load_q(floatx16& v, const float* __restrict__ row, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 x = *reinterpret_cast<const floatx4*>(row + 8 * g + 4 * hi);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[4 * g + t] = x[t];
    }
}
__device__ __forceinline__ void store_q(const floatx16& v, float* __restrict__ row, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        floatx4 x;
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = v[4 * g + t];
        *reinterpret_cast<floatx4*>(row + 8 * g + 4 * hi) = x;
    }
}

// exchange: every wave publishes its block (optionally through ReLU), then reads the full vector
template <int H, bool RELU>
__device__ __forceinline__ void wide_exchange(const floatx16& acc, floatx16 (&act)[H / 32], float* X, int n, int hi, int wave) {
    floatx16 v = acc;
    if (RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    store_q(v, X + n * XS + 32 * wave, hi);
    lds_barrier();
    load_feat(act, X + n * XS, hi);
    // the tile is rewritten only after the next layer's stage barriers (all reads above are waited by then)
}

template <int H, int NL, int MODE>
__global__ void __launch_bounds__(THREADS, 2) node_kernel_wide(NodeArgs A) {
    constexpr int NJB = H / 32;
    static_assert(NJB == 4, "wide node kernel: one feature block per wave of a 4-wave workgroup");
    constexpr int SL = (H / 8) * NJB / STAGE_PIECES;
    constexpr int S_IN = (4 * NJB + STAGE_PIECES - 1) / STAGE_PIECES;
    constexpr int S_OUT = (H / 8 + STAGE_PIECES - 1) / STAGE_PIECES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    float* X = ring + WRING * STAGE_FLOATS;
    float* vecs = X + WTILE * XS;  // [NL+1 biases | gamma | beta | proj_bias | dec biases NL x H + 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const int N = A.n_nodes;
    const int ntiles = (N + WTILE - 1) / WTILE;

    float* lbias = vecs;
    float* lgamma = lbias + (NL + 1) * H;
    float* lbeta = lgamma + H;
    float* lproj = lbeta + H;
    float* ldec = lproj + H;
    if (MODE != 2) {
        for (int i = tid; i < (NL + 1) * H; i += THREADS) lbias[i] = A.bias[i];
        for (int i = tid; i < H; i += THREADS) { lgamma[i] = A.ln_g[i]; lbeta[i] = A.ln_b[i]; }
    }
    if (A.tail == 1) for (int i = tid; i < H; i += THREADS) lproj[i] = A.proj_bias[i];
    if (A.tail == 2) for (int i = tid; i < NL * H + 32; i += THREADS) ldec[i] = A.dec_bias[i];

    WideCtx<H> c;
    c.base = A.wstream;
    c.ring = ring;
    const int mlp_stages = MODE == 0 ? (S_IN + NL * SL) : (MODE == 1 ? (NL + 2) * SL : 0);
    const int tail_stages = A.tail == 1 ? 2 * SL : (A.tail == 2 ? NL * SL + S_OUT : 0);
    c.total = mlp_stages + tail_stages;
    c.cur = 0;
    c.slot = 0;
    c.issued = 0;
    c.lane = lane;
    c.wave = wave;
    c.more_tiles = true;
    if ((int)blockIdx.x < ntiles) {
        for (int st = 0; st < WRING - 1 && st < c.total; ++st) {
            wide_issue(c, st, st);
            c.issued++;
        }
    }
    // `issued` counts stages requested but not yet consumed, including `cur` itself
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        c.more_tiles = tile + (int)gridDim.x < ntiles;
        const int p = tile * WTILE + n;
        const bool valid = p < N;
        const int64_t pc = valid ? p : N - 1;
        floatx16 act[NJB];
        floatx16 acc;
        if (MODE == 0) {
            load_feat_guard(act, A.x_in + pc * A.k1, hi, A.k1);
            load_q(acc, lbias + 32 * wave, hi);
            wide_layer<H, 4, NJB, true>(acc, act, c);
        } else if (MODE == 1) {
            load_feat(act, A.x_in + pc * H, hi);
            load_q(acc, lbias + 32 * wave, hi);
            wide_layer<H, H / 8, NJB, true>(acc, act, c);
            load_feat(act, A.agg + pc * H, hi);
            if (A.agg_clear && valid) {  // each wave clears its own block of the row
                floatx16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                store_q(z, A.agg_clear + pc * H + 32 * wave, hi);
            }
            wide_layer<H, H / 8, NJB, true>(acc, act, c);
        }
        if (MODE != 2) {
#pragma unroll
            for (int l = 1; l <= NL; ++l) {
                wide_exchange<H, true>(acc, act, X, n, hi, wave);
                load_q(acc, lbias + l * H + 32 * wave, hi);
                wide_layer<H, H / 8, NJB, false>(acc, act, c);
            }
            wide_exchange<H, false>(acc, act, X, n, hi, wave);   // act = full pre-LayerNorm vector (in every wave)
            layer_norm_regs(act, lgamma, lbeta, A.eps, hi);
            if (MODE == 1 && A.residual) add_feat(act, A.x_in + pc * H, hi);
            if (valid) {  // each wave stores its own block of h'
                float* row = A.h_out + pc * H + 32 * wave;
                if (wave == 0) store_q(act[0], row, hi);
                else if (wave == 1) store_q(act[1], row, hi);
                else if (wave == 2) store_q(act[2], row, hi);
                else store_q(act[3], row, hi);
            }
        } else {
            load_feat(act, A.x_in + pc * H, hi);
        }
        if (A.tail == 1) {
            load_q(acc, lproj + 32 * wave, hi);
            wide_layer<H, H / 8, NJB, true>(acc, act, c);
            if (valid) store_q(acc, A.P_out + pc * (2 * H) + 32 * wave, hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            wide_layer<H, H / 8, NJB, true>(acc, act, c);
            if (valid) store_q(acc, A.P_out + pc * (2 * H) + H + 32 * wave, hi);
        } else if (A.tail == 2) {
            load_q(acc, ldec + 32 * wave, hi);
            wide_layer<H, H / 8, NJB, true>(acc, act, c);
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                wide_exchange<H, true>(acc, act, X, n, hi, wave);
                load_q(acc, ldec + l * H + 32 * wave, hi);
                wide_layer<H, H / 8, NJB, false>(acc, act, c);
            }
            wide_exchange<H, true>(acc, act, X, n, hi, wave);
            load_q(acc, ldec + NL * H, hi);  // out bias, zero-padded to 32 (only wave 0's result is used)
            wide_layer<H, H / 8, 1, false>(acc, act, c);
            if (valid && hi == 0 && wave == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < A.out_dim) A.dec_out[pc * A.out_dim + q] = acc[q];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
size_t edge_lds_bytes() { return (size_t)(2 * STAGE_FLOATS + TILE * TS + 2 * (TILE + 4) + 2 * 2 * 4 * 64 + 8 * 128) * 4; }
size_t edge16_lds_bytes() { return (size_t)(2 * STAGE_FLOATS + T16 * TS16 + 2 * (T16 + 4) + 2 * 1024 + 6 * 128) * 4; }
size_t node_lds_bytes() { return (size_t)(2 * STAGE_FLOATS) * 4; }
size_t node_wide_lds_bytes() { return (size_t)(WRING * STAGE_FLOATS + WTILE * XS + 12 * 128 + 64) * 4; }

static int grid_for(int64_t tiles) {
    if (tiles < 1) tiles = 1;
    return (int)(tiles < 2048 ? tiles : 2048);
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return GM_OK;
}


enum : int { EK_AUTO = 0, EK_16 = 1, EK_CLASSIC = 2, EK_B3 = 3, EK_B3P = 4, EK_SYS = 5, EK_HM = 6 };

template <int H>
static int launch_edge_h(bool enc, const EdgeArgs& a, int grid, size_t lds, hipStream_t s) {
    static PerDeviceOnce attr_done_dev;
    {
        const int rc_attr = attr_done_dev.run([&]() -> int {
        int rc = set_lds(edge_kernel<H, 2, 0>, lds);
        if (rc == GM_OK) rc = set_lds(edge_kernel<H, 2, 1>, lds);
        if (rc == GM_OK) rc = set_lds(edge_kernel<H, 2, 2>, lds);
            return rc;
        });
        if (rc_attr != GM_OK) return rc_attr;
    }
    ProfScope prof(a.prof, enc ? PROF_ENC : PROF_EDGE, s);
    if (enc) hipLaunchKernelGGL((edge_kernel<H, 2, 0>), dim3(grid), dim3(THREADS), lds, s, a);
    else if (a.residual) hipLaunchKernelGGL((edge_kernel<H, 2, 1>), dim3(grid), dim3(THREADS), lds, s, a);
    else hipLaunchKernelGGL((edge_kernel<H, 2, 2>), dim3(grid), dim3(THREADS), lds, s, a);
    return GM_OK;
}


// the fp32 / bf16 x 6 forms of the processor / encoder edge MLP (kernel choices 1 .. 4)
int launch_edge_dev(int H, int NL, bool enc, const EdgeArgs& a_in, int64_t edge_capacity, hipStream_t s) {
    EdgeArgs a = a_in;
    const int choice = a.kernel_choice;
    GM_REQUIRE((H == 128 || H == 256) && NL == 2 && a.wstream, GM_ERR_UNSUPPORTED,
               "edge kernel: hidden_size=%d num_layers=%d: the fp32 kernels are instantiated for 128 / 256 and 2", H, NL);
    const int grid = grid_for(cdiv(edge_capacity, TILE));
    const size_t lds = edge_lds_bytes();
    if (H == 256) {
        int rc = launch_edge_h<256>(enc, a, grid, lds, s);
        if (rc != GM_OK) return rc;
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    int ncu = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    }
    const bool b3_ok = H == 128 && !enc && a.wstream_b3 && a.agg;
    const bool small = cdiv(edge_capacity, TE3) < 4 * (int64_t)ncu;
    const bool use_b3p = b3_ok && (choice == EK_B3P || (choice == EK_AUTO && small));
    const bool use_b3 = b3_ok && (choice == EK_B3 || (choice == EK_AUTO && !small));
    if (use_b3p) {
        const size_t lb = (size_t)(2 * B3P_STAGE_FLOATS + T16 * TSP + 2 * (T16 + 4) + 2 * 1024 + 4 * 128 + 4 * 32 * 2) * 4;
        static PerDeviceOnce donebp_dev;
        {
            const int rc_attr = donebp_dev.run([&]() -> int {
            int rc = set_lds(edge_kernel_b3p<2, 1>, lb);
            if (rc == GM_OK) rc = set_lds(edge_kernel_b3p<2, 2>, lb);
                return rc;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int gb = grid_for(cdiv(edge_capacity, T16));
        if (gb > 2 * cus) gb = 2 * cus;
        {
            ProfScope prof(a.prof, PROF_EDGE, s);
            if (a.residual) hipLaunchKernelGGL((edge_kernel_b3p<2, 1>), dim3(gb), dim3(THREADS), lb, s, a);
            else hipLaunchKernelGGL((edge_kernel_b3p<2, 2>), dim3(gb), dim3(THREADS), lb, s, a);
        }
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    if (use_b3) {
        const size_t lb = (size_t)(2 * B3_STAGE_FLOATS + TE3 * TSP + 2 * (TE3 + 4) + 2 * 2048 + 4 * 128 + 8 * 32 * 2) * 4;
        static PerDeviceOnce doneb_dev;
        {
            const int rc_attr = doneb_dev.run([&]() -> int {
            int rc = set_lds(edge_kernel_b3<2, 1>, lb);
            if (rc == GM_OK) rc = set_lds(edge_kernel_b3<2, 2>, lb);
                return rc;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int gb = grid_for(cdiv(edge_capacity, TE3));
        if (gb > cus) gb = cus;
        {
            ProfScope prof(a.prof, PROF_EDGE, s);
            if (a.residual) hipLaunchKernelGGL((edge_kernel_b3<2, 1>), dim3(gb), dim3(B3_THREADS), lb, s, a);
            else hipLaunchKernelGGL((edge_kernel_b3<2, 2>), dim3(gb), dim3(B3_THREADS), lb, s, a);
        }
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    const bool use16 = choice != EK_CLASSIC;
    if (H == 128 && use16 && a.wstream16) {
        const size_t l16 = edge16_lds_bytes();
        static PerDeviceOnce done16_dev;
        {
            const int rc_attr = done16_dev.run([&]() -> int {
            int rc = set_lds(edge_kernel16<2, 0>, l16);
            if (rc == GM_OK) rc = set_lds(edge_kernel16<2, 1>, l16);
            if (rc == GM_OK) rc = set_lds(edge_kernel16<2, 2>, l16);
                return rc;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        a.wstream = a.wstream16;
        // persistent grid: one workgroup per resident slot (2 per CU), so that every workgroup walks several tiles and
        // the next tile's gathers / first weight stage are always in flight (at N = 5k a one-tile-per-workgroup grid
        // costs 6 % per launch: no prefetch, 3x the prologues)
        static const int grid_cap = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return 2 * cus;
        }();
        int g16 = grid_for(cdiv(edge_capacity, T16));
        if (grid_cap > 0 && g16 > grid_cap) g16 = grid_cap;
        {
            ProfScope prof(a.prof, enc ? PROF_ENC : PROF_EDGE, s);
            if (enc) hipLaunchKernelGGL((edge_kernel16<2, 0>), dim3(g16), dim3(THREADS), l16, s, a);
            else if (a.residual) hipLaunchKernelGGL((edge_kernel16<2, 1>), dim3(g16), dim3(THREADS), l16, s, a);
            else hipLaunchKernelGGL((edge_kernel16<2, 2>), dim3(g16), dim3(THREADS), l16, s, a);
        }
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    {
        int rc = launch_edge_h<128>(enc, a, grid, lds, s);
        if (rc != GM_OK) return rc;
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_node_dev(int H, int NL, int mode, const NodeArgs& a, hipStream_t s) {
    GM_REQUIRE((H == 128 || H == 256) && NL == 2 && a.wstream, GM_ERR_UNSUPPORTED,
               "node kernel: hidden_size=%d num_layers=%d: the fp32 kernels are instantiated for 128 / 256 and 2", H, NL);
    const int grid = grid_for(cdiv(a.n_nodes, TILE));
    const size_t lds = node_lds_bytes();
    const bool wide = H == 128 && cdiv(a.n_nodes, TILE) <= 64;
    if (wide) {
        const size_t wl = node_wide_lds_bytes();
        static PerDeviceOnce attr_done_dev;
        {
            const int rc_attr = attr_done_dev.run([&]() -> int {
            int rc = set_lds(node_kernel_wide<128, 2, 0>, wl);
            if (rc == GM_OK) rc = set_lds(node_kernel_wide<128, 2, 1>, wl);
            if (rc == GM_OK) rc = set_lds(node_kernel_wide<128, 2, 2>, wl);
                return rc;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        const int wg = grid_for(cdiv(a.n_nodes, WTILE));
        ProfScope prof(a.prof, mode == 1 ? PROF_NODE : PROF_ENC, s);
        switch (mode) {
            case 0: hipLaunchKernelGGL((node_kernel_wide<128, 2, 0>), dim3(wg), dim3(THREADS), wl, s, a); break;
            case 1: hipLaunchKernelGGL((node_kernel_wide<128, 2, 1>), dim3(wg), dim3(THREADS), wl, s, a); break;
            default: hipLaunchKernelGGL((node_kernel_wide<128, 2, 2>), dim3(wg), dim3(THREADS), wl, s, a); break;
        }
    } else {
        ProfScope prof(a.prof, mode == 1 ? PROF_NODE : PROF_ENC, s);
        if (H == 128) {
            switch (mode) {
                case 0: hipLaunchKernelGGL((node_kernel<128, 2, 0>), dim3(grid), dim3(THREADS), lds, s, a); break;
                case 1: hipLaunchKernelGGL((node_kernel<128, 2, 1>), dim3(grid), dim3(THREADS), lds, s, a); break;
                default: hipLaunchKernelGGL((node_kernel<128, 2, 2>), dim3(grid), dim3(THREADS), lds, s, a); break;
            }
        } else {
            switch (mode) {
                case 0: hipLaunchKernelGGL((node_kernel<256, 2, 0>), dim3(grid), dim3(THREADS), lds, s, a); break;
                case 1: hipLaunchKernelGGL((node_kernel<256, 2, 1>), dim3(grid), dim3(THREADS), lds, s, a); break;
                default: hipLaunchKernelGGL((node_kernel<256, 2, 2>), dim3(grid), dim3(THREADS), lds, s, a); break;
            }
        }
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm
