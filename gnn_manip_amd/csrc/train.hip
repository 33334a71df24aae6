// Training path of the encode-process-decode model: forward with an activation tape and the full
// backward pass (callers: examples/train_dyn.py:45-72 -> loss.backward()).
//
// Design.  The per-row work (everything that is "one graph element through an MLP") is a transposed MFMA chain: a wave
// carries 32 rows x H features through consecutive Linear layers in registers, weights streamed L2 -> LDS in stages.  The
// backward of  y = LN(W_(L+1) relu(... relu(W_1 x + b_1) ...) + b_(L+1))  is the same kind of chain run with TRANSPOSED weights:
//     dz_(L+1) = LN'(dy);  dz_l = (W_(l+1)^T dz_(l+1)) * [a_l > 0];  dx = W_1^T dz_1
// so one kernel per MLP produces dz_1 .. dz_(L+1), the input gradient and -- summed over its rows -- the LayerNorm parameter
// gradients.  What else reduces over ROWS (weight and bias gradients) is done by wgrad_kernel on the arrays the chain kernels
// leave in HBM: dW = dz^T X with the rows on the K dimension of the MFMA, operands read in their natural row-major layout,
// bias gradient = column sums of the same operand, jobs batched per launch, deterministic two-stage reduction.
// Arithmetic of both: three-way bf16 operand splits, six partial products, fp32 accumulation (below).
//
// The layer-1 factorisation of the edge MLP (P = h [W_i|W_j]^T per node) carries over to the backward:
// by linearity  dh_i = W_i^T sum_{edges into i} dz1  and  dW_i = (sum_{edges into i} dz1)^T h, so the
// per-edge dz1 rows are first segment-summed per destination (G_i) and per source (G_j) node and both
// products run over N rows instead of E.
//
// Tape (saved by the forward, one set per MLP): post-ReLU activations a_1 .. a_L, the normalised
// pre-affine LayerNorm output xhat and 1/std per row; plus the block inputs h_k, e_k, agg_k.
#include <vector>
#include "common.h"
#include "mlp.h"
#include "fchain.h"
#include "train.h"

namespace gm {

// ------------------------------------------------------------------------------------------
// Arithmetic of the chains and of the weight gradients: fp32 products on the bf16 matrix pipe.  Every operand is split into
// three bf16 parts (x = hi + mid + lo: 24 significant bits with the FULL float exponent range -- gradients span many
// orders of magnitude, which rules out the fp16 split of the inference kernels here without a scale search per array) and
// six of the nine partial products are accumulated in fp32 (hh, hm, mh, mm, hl, lh: what is dropped is below 2^-24 of a
// product).  v_mfma_f32_32x32x16_bf16 does 8 x the flop of the fp32-input MFMA per issue slot: 2.7 x the throughput at
// fp32 accuracy.  Weights are split when their images are packed, activations / gradients in registers.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2v __attribute__((ext_vector_type(2)));

struct Bf3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Bf3 split_bf3(const float (&v)[8]) {
    Bf3 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const floatx2v x = floatx2v{v[e], v[e + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const floatx2v r1 = x - __builtin_convertvector(h, floatx2v);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const floatx2v r2 = r1 - __builtin_convertvector(m, floatx2v);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        o.h[e] = h[0]; o.h[e + 1] = h[1];
        o.m[e] = m[0]; o.m[e + 1] = m[1];
        o.l[e] = l[0]; o.l[e + 1] = l[1];
    }
    return o;
}
__device__ __forceinline__ void mfma_bf3(floatx16& acc, const Bf3& a, const Bf3& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// Weight stream of the chains.  Image of a Linear with K inputs and OUT outputs: groups g = kg * NJB + jb (kg = 16-wide
// k-group, jb = 32-row output block), each group = 3 parts (hi, mid, lo) x 64 lanes x 8 bf16 = 3 KiB: the A fragments of
// v_mfma_f32_32x32x16_bf16.  K slot (lane >> 5, e) of k-group kg carries input feature 16 kg + 8 (e >> 2) + 4 (lane >> 5) + (e & 3),
// which is where the 32x32 accumulator layout keeps that feature (registers 8 q .. 8 q + 7 of block jb are the eight
// slots of k-group 2 jb + q on the same lane): a Linear's output becomes the next one's B fragments without leaving the
// registers.  Stages of 8 groups (24 KiB) are streamed L2 -> LDS by DMA one stage ahead, shared by the workgroup's waves.
// ------------------------------------------------------------------------------------------
constexpr int B3_GROUP_FLOATS = 768;
constexpr int B3_STAGE_GROUPS = 8;
constexpr int B3_STAGE_FLOATS = B3_GROUP_FLOATS * B3_STAGE_GROUPS;   // = kStageFloatsB3
static_assert(B3_STAGE_FLOATS == kStageFloatsB3, "stage size");

__device__ __forceinline__ void issue_stage3(const WStream& ws, int stage, int buf) {
#pragma unroll
    for (int c = 0; c < 3 * B3_STAGE_GROUPS / 4; ++c) {
        const int piece = c * 4 + ws.wave;  // one wave-instruction = one contiguous 1 KiB piece
        const float* g = ws.base + (size_t)stage * B3_STAGE_FLOATS + piece * PIECE_FLOATS + ws.lane * 4;
        float* l = ws.ring + buf * B3_STAGE_FLOATS + piece * PIECE_FLOATS;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
}

// One Linear: acc[jb] += W(block jb) . act.  NKG = K / 16 k-groups, NJB = OUT / 32 blocks, act in the accumulator layout.
// more / PEND: as run_layer (fchain.h).
template <int NKG, int NJB, int NKB, int PEND = 0>
__device__ __forceinline__ void run_layer_b3(floatx16 (&acc)[NJB], const floatx16 (&act)[NKB], WStream& ws, bool more_tiles) {
    constexpr int NG = NKG * NJB;
    constexpr int NST = (NG + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;
    Bf3 b;
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
            int st = nxt;
            asm volatile("" : "+s"(st));   // launder: keeps the DMA addresses from being precomputed per stage outside the tile loop
            if (!wrap || more_tiles) issue_stage3(ws, st, ws.parity ^ 1);
            __builtin_amdgcn_sched_barrier(0);   // the DMA issue stays right behind the barrier
        }
        const bf16x8* buf = reinterpret_cast<const bf16x8*>(ws.ring + ws.parity * B3_STAGE_FLOATS) + ws.lane;
#pragma unroll
        for (int u = 0; u < B3_STAGE_GROUPS; ++u) {
            const int g = s * B3_STAGE_GROUPS + u;
            if (g < NG) {
                const int kg = g / NJB, jb = g % NJB;
                if (jb == 0) {   // a new k-group: its B fragments from the activation registers
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act[kg >> 1][8 * (kg & 1) + e];
                    b = split_bf3(v);
                }
                Bf3 a;
                a.h = buf[(u * 3 + 0) * 64];
                a.m = buf[(u * 3 + 1) * 64];
                a.l = buf[(u * 3 + 2) * 64];
                mfma_bf3(acc[jb], a, b);
            }
        }
        ws.cur = nxt;
        ws.parity ^= 1;
    }
}

// ------------------------------------------------------------------------------------------
// small register helpers on the 32x32 feature layout (see fchain.h)
// ------------------------------------------------------------------------------------------
template <int NKB>
__device__ __forceinline__ void zero_feat(floatx16 (&v)[NKB]) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[kb][r] = 0.f;
}
// The ReLU mask of a chain's activations as bits: lane (n, hi) packs its 16 NKB values (value 16 kb + r = feature
// 32 kb + 8 (r >> 2) + 4 hi + (r & 3) of row n) into NKB / 2 words, stored at m[row][hi][NKB / 2] -- one store instruction per
// lane, 512 contiguous bytes per wave at hidden 128.  The backward chain reads these H / 8 bytes per row instead of the
// 4 H bytes of a_l (whose values only the weight-gradient jobs need).
template <int NKB>
__device__ __forceinline__ void relu_mask_store(const floatx16 (&act)[NKB], uint32_t* __restrict__ m) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // value q of a word is shifted in from the right (and the word reversed at the end).  The activations are post-ReLU (>= +0): the
    // integer negation of the bit pattern has its sign bit set exactly when the value is > 0, and v_alignbit shifts that bit in:
    // two vector instructions per value
    uint32_t w[NKB / 2];
#pragma unroll
    for (int q = 0; q < NKB / 2; ++q) w[q] = 0u;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = act[kb][r];   // (a scalar copy: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
            w[kb >> 1] = __builtin_amdgcn_alignbit(w[kb >> 1], 0u - __float_as_uint(x), 31);
        }
#pragma unroll
    for (int q = 0; q < NKB / 2; ++q) w[q] = __builtin_bitreverse32(w[q]);   // value q at bit q
    if (NKB == 2) *m = w[0];
    else if (NKB == 4) *reinterpret_cast<u32x2*>(m) = u32x2{w[0], w[1 % (NKB / 2)]};
    else *reinterpret_cast<u32x4*>(m) = u32x4{w[0], w[1 % (NKB / 2)], w[2 % (NKB / 2)], w[3 % (NKB / 2)]};
}
template <int NKB>
__device__ __forceinline__ void mask_bits(floatx16 (&v)[NKB], const uint32_t* __restrict__ m) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t w[NKB / 2];
    if (NKB == 2) {
        w[0] = *m;
    } else if (NKB == 4) {
        const u32x2 x = *reinterpret_cast<const u32x2*>(m);
        w[0] = x[0];
        w[1 % (NKB / 2)] = x[1];
    } else {
        const u32x4 x = *reinterpret_cast<const u32x4*>(m);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q % (NKB / 2)] = x[q];
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // a sign-extended one-bit field (0 / all ones) and an AND
            const float x = v[kb][r];
            const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe((int)w[kb >> 1], (kb & 1) * 16 + r, 1);
            v[kb][r] = __uint_as_float(__float_as_uint(x) & keep);
        }
}

// LayerNorm forward that also leaves xhat in `xh` (registers) and returns 1/std; acc <- xhat*gamma + beta
template <int NJB>
__device__ __forceinline__ float layer_norm_tape(floatx16 (&acc)[NJB], floatx16 (&xh)[NJB], const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, int hi) {
    constexpr float INV_H = 1.0f / (32 * NJB);
    float s = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[jb][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * INV_H;
    float q = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[jb][r] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * INV_H + eps);
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 gm = *reinterpret_cast<const floatx4*>(gamma + 32 * jb + 8 * g + 4 * hi);
            const floatx4 bt = *reinterpret_cast<const floatx4*>(beta + 32 * jb + 8 * g + 4 * hi);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float x = (acc[jb][4 * g + t] - mean) * rstd;
                xh[jb][4 * g + t] = x;
                acc[jb][4 * g + t] = x * gm[t] + bt[t];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return rstd;
}

// LayerNorm backward: dy in `g` (overwritten), xhat in `xh`; result dz = rstd (g*gamma - mean(g*gamma) - xhat mean(g*gamma*xhat)) -> xh
template <int NJB>
__device__ __forceinline__ void layer_norm_bwd(floatx16 (&g)[NJB], floatx16 (&xh)[NJB], const float* __restrict__ gamma, float rstd, int hi) {
    constexpr float INV_H = 1.0f / (32 * NJB);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const floatx4 gm = *reinterpret_cast<const floatx4*>(gamma + 32 * jb + 8 * q + 4 * hi);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = g[jb][4 * q + t] * gm[t];
                g[jb][4 * q + t] = v;
                s1 += v;
                s2 += v * xh[jb][4 * q + t];
            }
        }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float m1 = s1 * INV_H, m2 = s2 * INV_H;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) xh[jb][r] = rstd * (g[jb][r] - m1 - xh[jb][r] * m2);
}

// LayerNorm parameter gradients inside the backward chain: dgamma[c] += sum_rows gy[row][c] xhat[row][c], dbeta[c] += sum_rows gy[row][c].
// Rows sit on the lanes (32 per half wave), so a column sum is a 32-lane reduction per register: DPP prefix adds inside the
// 16-lane rows (row_shr 1 / 2 / 4 / 8, zeros shifted in) and row_bcast:15 across them leave the totals in lanes 31 / 63, which
// add them to the wave's slots of an LDS accumulator (one writer per address).  ~1.4k vector instructions per 32-row block
// against the ~1000 MFMAs of its chain; it replaces a kernel that re-read gy and xhat (8 H bytes per row) from HBM.
__device__ __forceinline__ float half_wave_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));
    return x;   // lanes 31 and 63: the sum over their 32 lanes
}
template <int NJB>
__device__ __forceinline__ void ln_param_sums(const floatx16 (&g)[NJB], const floatx16 (&xh)[NJB], bool valid, float* acc_w /*LDS [2 H]*/, int n, int hi) {
    constexpr int H = 32 * NJB;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gv = valid ? g[jb][r] : 0.f;
            const float a = half_wave_sum(gv * xh[jb][r]);
            const float b = half_wave_sum(gv);
            if (n == 31) {
                const int f = 32 * jb + 8 * (r >> 2) + 4 * hi + (r & 3);
                acc_w[f] += a;
                acc_w[H + f] += b;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Which of the chains' row-major arrays move as whole 128-byte lines through a wave-private LDS turn (store_feat_lines /
// load_feat_lines) instead of accumulator-layout pieces (32 rows x 32 bytes per instruction).  Forward: 1 tape stores, 2 output
// stores.  Backward: 8 dz stores, 16 input-gradient stores, 32 dY / G / xhat loads.
#ifndef TRAIN_LINES
#define TRAIN_LINES 63
#endif

// ------------------------------------------------------------------------------------------
// forward with tape
// ------------------------------------------------------------------------------------------

template <int H, int KIND>
__global__ void __launch_bounds__(THREADS, H <= 128 ? 2 : 1) train_fwd_kernel(TrainFwdArgs A) {
    constexpr int NJB = H / 32;
    constexpr int SL = ((H / 16) * NJB + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;   // stages of one H x H Linear
    const int NL = A.nl;   // any num_layers >= 2 (epd_gnn.py:72-84): the hidden Linears are a run-time loop
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const int R = A.rows;
    const size_t tstride = (size_t)R * H;   // rows H between the tape's per-layer activation arrays
    const int ntiles = (R + TILE - 1) / TILE;
    WStream ws;
    ws.base = A.wstream;
    ws.ring = ring;
    const bool proj_tail = (KIND == TK_ENC_NODE || KIND == TK_PROC_NODE) && A.P_out != nullptr;
    ws.total = KIND == TK_ENC_EDGE ? 1 + NL * SL
             : KIND == TK_ENC_NODE ? (2 * NJB + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS + NL * SL + (proj_tail ? 2 * SL : 0)
             : KIND == TK_PROC_EDGE ? (NL + 1) * SL
             : KIND == TK_PROC_NODE ? (NL + 2) * SL + (proj_tail ? 2 * SL : 0)
             : KIND == TK_PROJ ? 2 * SL
                                   : NL * SL + (H / 16 + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;
    ws.cur = 0;
    ws.parity = 0;
    ws.lane = lane;
    ws.wave = wave;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float* turn = ring + 2 * B3_STAGE_FLOATS + wave_u * TURN_FLOATS;   // store_feat_lines
    if ((int)blockIdx.x < ntiles) issue_stage3(ws, 0, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more = tile + (int)gridDim.x < ntiles;
        const int p = tile * TILE + wave * 32 + n;
        const bool valid = p < R;
        const int64_t pc = valid ? p : R - 1;
        const int row0 = tile * TILE + wave_u * 32;   // the wave's rows: row0 .. row0 + 31
        floatx16 acc[NJB], act[NJB];
        if (KIND == TK_ENC_EDGE || KIND == TK_ENC_NODE) {
            const int64_t rin = A.rowidx ? A.rowidx[pc] : pc;
            load_feat_guard(act, A.x_in + rin * A.k1, hi, A.k1);
            load_feat(acc, A.bias, hi);
            if (KIND == TK_ENC_EDGE) run_layer_b3<1, NJB, NJB>(acc, act, ws, more);
            else run_layer_b3<2, NJB, NJB>(acc, act, ws, more);
        } else if (KIND == TK_PROC_EDGE) {
            load_feat(acc, A.P + (int64_t)A.dst[pc] * (2 * H), hi);
            add_feat(acc, A.P + (int64_t)A.src[pc] * (2 * H) + H, hi);
            load_feat(act, A.x_in + (A.rowidx ? (int64_t)A.rowidx[pc] : pc) * H, hi);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
        } else if (KIND == TK_PROJ) {
            // the next edge step's factorised layer 1: P = [h W_i^T + b1 | h W_j^T] per node (what the inference node kernels' tail does)
            load_feat(act, A.x_in + pc * H, hi);
            load_feat(acc, A.bias, hi);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
            if (TRAIN_LINES & 2) store_feat_lines(acc, A.out + (size_t)row0 * (2 * H), 2 * H, R - row0, turn, lane);
            else store_feat(acc, A.out + pc * (2 * H), hi);
            zero_feat(acc);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
            if (TRAIN_LINES & 2) store_feat_lines(acc, A.out + (size_t)row0 * (2 * H) + H, 2 * H, R - row0, turn, lane);
            else store_feat(acc, A.out + pc * (2 * H) + H, hi);
            continue;
        } else if (KIND == TK_PROC_NODE) {
            load_feat(act, A.x_in + pc * H, hi);
            load_feat(acc, A.bias, hi);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
            load_feat(act, A.agg + pc * H, hi);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
        } else {
            load_feat(act, A.x_in + pc * H, hi);
            load_feat(acc, A.bias, hi);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
        }
        // hidden layers 2..NL and the output layer; bias_tail = bias of layer 2
        // Tape stores are issued by every lane (rows past the end write a duplicate of the last row: same
        // inputs, same values) so that their count is exact for run_layer<.., PEND>, which then does not
        // drain them; the bias loads go first so that waiting for them does not wait for the stores either.
        constexpr int NST = H / 8;  // vector stores of one store_feat
#pragma unroll 1
        for (int l = 1; l < NL; ++l) {   // Linear l + 1 (hidden)
            relu_to(act, acc);
            load_feat(acc, A.bias_tail + (size_t)(l - 1) * H, hi);
            if (TRAIN_LINES & 1) store_feat_lines(act, A.tape.a + (size_t)(l - 1) * tstride + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(act, A.tape.a + (size_t)(l - 1) * tstride + pc * H, hi);
            relu_mask_store(act, A.tape.mask + ((size_t)(l - 1) * R + pc) * NJB + hi * (NJB / 2));
            run_layer_b3<H / 16, NJB, NJB, NST + 1>(acc, act, ws, more);
        }
        relu_to(act, acc);
        if (KIND != TK_DEC) load_feat(acc, A.bias_tail + (size_t)(NL - 1) * H, hi);
        if (TRAIN_LINES & 1) store_feat_lines(act, A.tape.a + (size_t)(NL - 1) * tstride + (size_t)row0 * H, H, R - row0, turn, lane);
        else store_feat(act, A.tape.a + (size_t)(NL - 1) * tstride + pc * H, hi);
        relu_mask_store(act, A.tape.mask + ((size_t)(NL - 1) * R + pc) * NJB + hi * (NJB / 2));
        if (KIND == TK_DEC) {
            floatx16 o[1];
            load_feat(o, A.bias_tail + (size_t)(NL - 1) * H, hi);  // out bias, zero-padded to 32
            run_layer_b3<H / 16, 1, NJB>(o, act, ws, more);
            if (valid && hi == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < A.out_dim) A.out[pc * A.out_dim + c] = o[0][c];
            }
        } else {
            run_layer_b3<H / 16, NJB, NJB, NST + 1>(acc, act, ws, more);
            const float rstd = layer_norm_tape(acc, act, A.ln_g, A.ln_b, A.eps, hi);
            if (TRAIN_LINES & 1) store_feat_lines(act, A.tape.xhat + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(act, A.tape.xhat + pc * H, hi);
            if (valid && hi == 0) A.tape.rstd[pc] = rstd;
            if (KIND == TK_PROC_EDGE && A.rowidx) {   // the single-block entry point: e rows in the caller's order
                const int64_t orow = A.rowidx[pc];
                if (A.residual) add_feat(acc, A.x_in + orow * H, hi);
                if (valid) store_feat(acc, A.out + orow * H, hi);
            } else {
                if ((KIND == TK_PROC_EDGE || KIND == TK_PROC_NODE) && A.residual) add_feat(acc, A.x_in + pc * H, hi);
                if (TRAIN_LINES & 2) store_feat_lines(acc, A.out + (size_t)row0 * H, H, R - row0, turn, lane);
                else store_feat(acc, A.out + pc * H, hi);
            }
            if ((KIND == TK_ENC_NODE || KIND == TK_PROC_NODE) && proj_tail) {
                // the next edge step's factorised layer 1 on the rows still in registers: P = [h W_i^T + b1 | h W_j^T]
                // (what the inference node kernels' tail does; it was a launch of its own that re-read h)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
                load_feat(acc, A.proj_bias, hi);
                run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
                if (TRAIN_LINES & 2) store_feat_lines(acc, A.P_out + (size_t)row0 * (2 * H), 2 * H, R - row0, turn, lane);
                else store_feat(acc, A.P_out + pc * (2 * H), hi);
                zero_feat(acc);
                run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
                if (TRAIN_LINES & 2) store_feat_lines(acc, A.P_out + (size_t)row0 * (2 * H) + H, 2 * H, R - row0, turn, lane);
                else store_feat(acc, A.P_out + pc * (2 * H) + H, hi);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// backward chain
// ------------------------------------------------------------------------------------------

template <int H, int KIND>
__global__ void __launch_bounds__(THREADS, H <= 128 ? 2 : 1) train_bwd_kernel(TrainBwdArgs A) {
    constexpr int NJB = H / 32;
    constexpr int SL = ((H / 16) * NJB + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;   // stages of one H x H Linear
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
    const int R = A.rows;
    const int NL = A.nl;
    const size_t tstride = (size_t)R * H;
    const int ntiles = (R + TILE - 1) / TILE;
    const bool has_g = (KIND == TB_NODE || KIND == TB_ENC || KIND == TB_PROJ) && A.Gi != nullptr;
    constexpr bool NORMED = KIND == TB_NODE || KIND == TB_ENC || KIND == TB_EDGE;
    float* lnacc = ring + 2 * B3_STAGE_FLOATS;   // [4 waves][2 H]: this workgroup's share of the LayerNorm parameter gradients
    if (NORMED && A.ln_part) {
        for (int i = tid; i < 4 * 2 * H; i += THREADS) lnacc[i] = 0.f;
        __syncthreads();
    }
    WStream ws;
    ws.base = A.wstream;
    ws.ring = ring;
    constexpr int S_IN = (H / 16 + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;  // W1^T of an encoder: one 32-row block of outputs
    ws.total = (has_g ? 2 * SL : 0) +
               (KIND == TB_ENC ? NL * SL + (A.dx_in ? S_IN : 0) : KIND == TB_EDGE ? (NL + 1) * SL : KIND == TB_NODE ? (NL + 2) * SL
                : KIND == TB_PROJ ? 0 : 1 + NL * SL);
    ws.cur = 0;
    ws.parity = 0;
    ws.lane = lane;
    ws.wave = wave;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float* turn = lnacc + 4 * 2 * H + wave_u * TURN_FLOATS;   // store_feat_lines
    if ((int)blockIdx.x < ntiles) issue_stage3(ws, 0, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool more = tile + (int)gridDim.x < ntiles;
        const int p = tile * TILE + wave * 32 + n;
        const bool valid = p < R;
        const int64_t pc = valid ? p : R - 1;
        const int row0 = tile * TILE + wave_u * 32;
        floatx16 acc[NJB], act[NJB];
        if (KIND == TB_DEC) {
            // dz3 = dY [rows][out_dim]; first product has K = out_dim (one k-octet)
            load_feat_guard(act, A.dY + pc * A.out_dim, hi, A.out_dim);
            zero_feat(acc);
            run_layer_b3<1, NJB, NJB>(acc, act, ws, more);
        } else {
            // total upstream gradient of the MLP output row
            if (A.dY) {
                if (A.dyidx || !(TRAIN_LINES & 32)) load_feat(acc, A.dY + (A.dyidx ? (int64_t)A.dyidx[pc] : pc) * H, hi);
                else load_feat_lines<0>(acc, A.dY + (size_t)row0 * H, H, R - row0, turn, lane);
            } else {
                zero_feat(acc);
            }
            if (KIND == TB_EDGE && A.dagg) add_feat(acc, A.dagg + (int64_t)A.dst[pc] * H, hi);
            if (has_g) {  // + W_i^T G_i + W_j^T G_j : input gradient of the NEXT edge step's factorised layer 1
                if (TRAIN_LINES & 32) load_feat_lines<0>(act, A.Gi + (size_t)row0 * H, H, R - row0, turn, lane);
                else load_feat(act, A.Gi + pc * H, hi);
                run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
                if (TRAIN_LINES & 32) load_feat_lines<0>(act, A.Gj + (size_t)row0 * H, H, R - row0, turn, lane);
                else load_feat(act, A.Gj + pc * H, hi);
                run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);
            }
            if (KIND == TB_PROJ) {
                if (TRAIN_LINES & 16) store_feat_lines(acc, A.dx + (size_t)row0 * H, H, R - row0, turn, lane);
                else store_feat(acc, A.dx + pc * H, hi);
                continue;
            }
            if (valid && KIND == TB_NODE && A.dx_resid) store_feat(acc, A.dx_resid + pc * H, hi);  // residual path: dh_in starts as dY
            if (TRAIN_LINES & 32) load_feat_lines<0>(act, A.tape.xhat + (size_t)row0 * H, H, R - row0, turn, lane);
            else load_feat(act, A.tape.xhat + pc * H, hi);
            if (NORMED && A.ln_part) ln_param_sums(acc, act, valid, lnacc + wave * 2 * H, n, hi);
            layer_norm_bwd(acc, act, A.ln_g, A.tape.rstd[pc], hi);
            // dz stores: every lane (duplicates of the last row past the end), counted by run_layer<.., PEND>
            if (TRAIN_LINES & 8) store_feat_lines(act, A.dz + (size_t)NL * A.dz_stride + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(act, A.dz + (size_t)NL * A.dz_stride + pc * H, hi);
            zero_feat(acc);
            run_layer_b3<H / 16, NJB, NJB, H / 8>(acc, act, ws, more);  // W_(NL+1)^T dz_(NL+1)
        }
#pragma unroll 1
        for (int l = NL; l >= 2; --l) {   // dz_l = (W_(l+1)^T dz_(l+1)) [a_l > 0], then on through W_l^T
            mask_bits(acc, A.tape.mask + ((size_t)(l - 1) * R + pc) * NJB + hi * (NJB / 2));
            if (TRAIN_LINES & 8) store_feat_lines(acc, A.dz + (size_t)(l - 1) * A.dz_stride + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(acc, A.dz + (size_t)(l - 1) * A.dz_stride + pc * H, hi);
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
            zero_feat(acc);
            run_layer_b3<H / 16, NJB, NJB, H / 8>(acc, act, ws, more);
        }
        mask_bits(acc, A.tape.mask + (size_t)pc * NJB + hi * (NJB / 2));
        if (TRAIN_LINES & 8) store_feat_lines(acc, A.dz + (size_t)row0 * H, H, R - row0, turn, lane);
        else store_feat(acc, A.dz + pc * H, hi);
        if (KIND == TB_ENC) {
            if (A.dx_in) {  // gradient w.r.t. the raw input features: dX = dz1 . W1  (k1 <= 32 columns)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
                floatx16 o[1];
                zero_feat(o);
                run_layer_b3<H / 16, 1, NJB, H / 8>(o, act, ws, more);
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 8 * (r >> 2) + 4 * hi + (r & 3);
                        if (f < A.k1) A.dx_in[pc * A.k1 + f] = o[0][r];
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) act[jb] = acc[jb];
        zero_feat(acc);
        run_layer_b3<H / 16, NJB, NJB, H / 8>(acc, act, ws, more);  // W1^T dz1 (edge: W_e block; node: W_h block; decoder: W1)
        if (KIND == TB_EDGE) {
            // de_in = W_e^T dz1 (+ de_out through the residual), written over the row it came from
            if (A.residual && A.dY) {
                if (A.dyidx || !(TRAIN_LINES & 32)) add_feat(acc, A.dY + (A.dyidx ? (int64_t)A.dyidx[pc] : pc) * H, hi);
                else load_feat_lines<1>(acc, A.dY + (size_t)row0 * H, H, R - row0, turn, lane);
            }
            if (A.dxidx) {
                if (valid) store_feat(acc, A.dx + (int64_t)A.dxidx[pc] * H, hi);
            } else {
                if (TRAIN_LINES & 16) store_feat_lines(acc, A.dx + (size_t)row0 * H, H, R - row0, turn, lane);
                else store_feat(acc, A.dx + pc * H, hi);
            }
        } else if (KIND == TB_NODE) {
            if (A.dx_resid) add_feat(acc, A.dx_resid + pc * H, hi);  // same thread wrote this row above
            if (TRAIN_LINES & 16) store_feat_lines(acc, A.dx + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(acc, A.dx + pc * H, hi);
            zero_feat(acc);
            run_layer_b3<H / 16, NJB, NJB>(acc, act, ws, more);  // W_agg^T dz1
            if (TRAIN_LINES & 16) store_feat_lines(acc, A.dagg_out + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(acc, A.dagg_out + pc * H, hi);
        } else {
            if (TRAIN_LINES & 16) store_feat_lines(acc, A.dx + (size_t)row0 * H, H, R - row0, turn, lane);
            else store_feat(acc, A.dx + pc * H, hi);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (NORMED && A.ln_part) {
        __syncthreads();
        for (int i = tid; i < 2 * H; i += THREADS)
            A.ln_part[(size_t)blockIdx.x * 2 * H + i] = (lnacc[i] + lnacc[2 * H + i]) + (lnacc[4 * H + i] + lnacc[6 * H + i]);
    }
}

// ------------------------------------------------------------------------------------------
// dW = dz^T X  (+ db = column sums of dz), a batch of independent jobs per launch, each split over rows:
// partial[g][m][k], partial_b[g][m], reduced in fixed order by wgrad_reduce_kernel (no atomics: deterministic).
//
// Arithmetic: the sum over rows is the K dimension of v_mfma_f32_32x32x16_bf16.  Both operands are split in registers
// into three bf16 parts (x = hi + mid + lo, 24 significant bits, the full float exponent range -- gradients span many
// orders of magnitude, which rules the fp16 split of the inference kernels out here) and six of the nine partial
// products are accumulated in fp32 (hh, hm, mh, mm, hl, lh: what is left is below 2^-24 of a product): fp32 accuracy
// at 2.7 x the matrix throughput of the fp32-input MFMA this kernel used before, which makes a job of E rows
// bound by its 8 H E bytes of HBM reads instead.
//
// A wave owns a 64 x 64 block of dW (2 x 2 accumulator tiles).  K slot (lane >> 5, e) of a 16-row step is row
// r0 + 8 (lane >> 5) + e in BOTH operands: lane (i, hi) reads dz[row][m0 + i] and X[row][k0 + i] -- every load
// instruction covers two rows x 32 consecutive floats (two full 128-byte lines), straight from the row-major arrays.
// The operands of the next step are requested before the splits / MFMAs of the current one (register double buffer);
// every load is unconditional from a clamped (valid) address, so the waits are counted vmcnt.
// ------------------------------------------------------------------------------------------
struct WgRaw {
    float a0[8], a1[8], b0[8], b1[8];
};
typedef __amdgpu_buffer_rsrc_t wg_srd_t;
__device__ __forceinline__ wg_srd_t wg_make_srd(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000); }
__device__ __forceinline__ float wg_ld(wg_srd_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }

template <bool HAS_IDX>
__device__ __forceinline__ void wgrad_body(const WgJob& J, float* __restrict__ part) {
    const float* __restrict__ dz = J.dz;
    const float* __restrict__ X = J.X;
    const int* __restrict__ xidx = J.xidx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
    const int M = J.M, K = J.K, ldz = J.ldz, ldx = J.ldx;
    const int mt = blockIdx.y / J.KT, kt = blockIdx.y % J.KT;
    const int m0 = mt * 128 + 64 * (wave >> 1), k0 = kt * 128 + 64 * (wave & 1);
    const int r_begin = blockIdx.x * J.chunk;
    const int r_end = min(J.rows, r_begin + J.chunk);
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // columns past M / K read a clamped (valid) column: their products land in partial entries nobody reads
    const int mc0 = min(m0 + i, M - 1), mc1 = min(m0 + 32 + i, M - 1), kc0 = min(k0 + i, K - 1), kc1 = min(k0 + 32 + i, K - 1);
    float bs0 = 0.f, bs1 = 0.f;  // column sums of dz (bias gradient), kept by the waves of k-block 0
    // Buffer addressing: a resource per operand that covers exactly the chunk's rows (a read past them returns 0: the chunk's last,
    // partial step needs no masking), a lane-constant byte offset (row 8 hi of the step, column), and a SCALAR offset per row of
    // the step -- no per-lane 64-bit address arithmetic in the loop (it was a quarter of the loop's vector instructions).
    const int rows_c = r_end - r_begin;
    const wg_srd_t srd_z = wg_make_srd(dz + (size_t)r_begin * ldz, (unsigned)rows_c * (unsigned)ldz * 4u);
    const wg_srd_t srd_x = wg_make_srd(HAS_IDX ? X : X + (size_t)r_begin * ldx, HAS_IDX ? 0xffffffffu : (unsigned)rows_c * (unsigned)ldx * 4u);
    const unsigned vz0 = (unsigned)(8 * hi * ldz + mc0) * 4u, vz1 = (unsigned)(8 * hi * ldz + mc1) * 4u;
    const unsigned vx0 = (unsigned)(8 * hi * ldx + kc0) * 4u, vx1 = (unsigned)(8 * hi * ldx + kc1) * 4u;
    auto load = [&](WgRaw& o, int r0) {
        const unsigned rr = (unsigned)(r0 - r_begin);   // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned sz = (rr + e) * (unsigned)ldz * 4u;
            o.a0[e] = wg_ld(srd_z, vz0, sz);
            o.a1[e] = wg_ld(srd_z, vz1, sz);
        }
        if (HAS_IDX) {
            // gathered X rows (encoder inputs in the caller's edge order): per-lane row, 32-bit byte offsets; a row past the chunk
            // reads row index r_end - 1's -- its dz is zero
            unsigned xo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[e] = (unsigned)xidx[min(r0 + 8 * hi + e, r_end - 1)] * (unsigned)ldx * 4u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o.b0[e] = wg_ld(srd_x, xo[e] + (unsigned)kc0 * 4u, 0);
                o.b1[e] = wg_ld(srd_x, xo[e] + (unsigned)kc1 * 4u, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned sx = (rr + e) * (unsigned)ldx * 4u;
                o.b0[e] = wg_ld(srd_x, vx0, sx);
                o.b1[e] = wg_ld(srd_x, vx1, sx);
            }
        }
    };
    auto fma = [&](WgRaw& o, int) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs0 += o.a0[e]; bs1 += o.a1[e]; }
        const Bf3 a0 = split_bf3(o.a0), b0 = split_bf3(o.b0);
        mfma_bf3(acc[0][0], a0, b0);
        const Bf3 b1 = split_bf3(o.b1);
        mfma_bf3(acc[0][1], a0, b1);
        const Bf3 a1 = split_bf3(o.a1);
        mfma_bf3(acc[1][0], a1, b0);
        mfma_bf3(acc[1][1], a1, b1);
    };
    WgRaw A, B;
    int r0 = r_begin;
    if (r0 < r_end) {
        load(A, r0);
        while (true) {
            const int r1 = r0 + 16;
            load(B, r1);  // unconditional (clamped): a conditional request would make every wait below conservative
            __builtin_amdgcn_sched_barrier(0);
            fma(A, r0);
            __builtin_amdgcn_sched_barrier(0);
            if (r1 >= r_end) break;
            const int r2 = r1 + 16;
            load(A, r2);
            __builtin_amdgcn_sched_barrier(0);
            fma(B, r1);
            __builtin_amdgcn_sched_barrier(0);
            if (r2 >= r_end) break;
            r0 = r2;
        }
    }
    float* out = part + J.part_off + (size_t)blockIdx.x * J.Mp * J.Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * a + 8 * (r >> 2) + 4 * hi + (r & 3);
                const int k = k0 + 32 * b + i;
                out[(size_t)m * J.Kp + k] = acc[a][b][r];
            }
    if (J.db && kt == 0 && (wave & 1) == 0) {
        float* partb = part + J.part_off + (size_t)J.G * J.Mp * J.Kp;
        bs0 += __shfl_xor(bs0, 32, 64);
        bs1 += __shfl_xor(bs1, 32, 64);
        if (hi == 0) {
            partb[(size_t)blockIdx.x * J.Mp + m0 + i] = bs0;
            partb[(size_t)blockIdx.x * J.Mp + m0 + 32 + i] = bs1;
        }
    }
}

// grid (max G, max tiles, jobs)
__global__ void __launch_bounds__(THREADS, 2) wgrad_kernel(WgJobs Js, float* __restrict__ part) {
    const WgJob& J = Js.job[blockIdx.z];
    if ((int)blockIdx.x >= J.G || (int)blockIdx.y >= J.tiles) return;
    if (J.xidx) wgrad_body<true>(J, part);
    else wgrad_body<false>(J, part);
}

// sum over g = g0, g0 + 8, g0 + 16, ... < G of p[g * stride]: four loads in flight (the reduction kernels are a chain of dependent
// L2 round trips otherwise), four accumulators combined in a fixed order -- the result depends on G only, not on timing
__device__ __forceinline__ float strided_sum4(const float* __restrict__ p, size_t stride, int g0, int G) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = g0;
    for (; g + 24 < G; g += 32) {
        const float v0 = p[(size_t)g * stride], v1 = p[(size_t)(g + 8) * stride], v2 = p[(size_t)(g + 16) * stride], v3 = p[(size_t)(g + 24) * stride];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; g < G; g += 8) s0 += p[(size_t)g * stride];
    return (s0 + s1) + (s2 + s3);
}

// out[m][col0 + k] += sum_g part[g][m][k]; db[m] += sum_g partb[g][m].  32 outputs x 8 partial groups per
// workgroup, fixed summation order (deterministic).  grid (max outputs / 32, jobs)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgJobs Js, const float* __restrict__ part_all) {
    __shared__ float red[8][33];
    const int tid = threadIdx.x, i = tid & 31, gg = tid >> 5;
    const int o = blockIdx.x * 32 + i;
    if ((int)blockIdx.y >= Js.n) {   // the LayerNorm parameter gradients of a backward chain since the last flush (fixed order: deterministic)
        const WgLnJob& Lj = Js.ln[blockIdx.y - Js.n];
        const int H = Lj.H;
        if ((int)blockIdx.x * 32 >= 2 * H) return;
        const float s = o < 2 * H ? strided_sum4(Lj.part + o, (size_t)2 * H, gg, Lj.G) : 0.f;
        red[gg][i] = s;
        __syncthreads();
        if (gg == 0 && o < 2 * H) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q][i];
            if (o < H) Lj.dgamma[o] += t;
            else Lj.dbeta[o - H] += t;
        }
        return;
    }
    const WgJob& J = Js.job[blockIdx.y];
    const int M = J.M, K = J.K, Mp = J.Mp, Kp = J.Kp, G = J.G;
    const int nw = M * K, total = nw + (J.db ? M : 0);
    if ((int)blockIdx.x * 32 >= total) return;
    const float* part = part_all + J.part_off;
    const float* partb = part + (size_t)G * Mp * Kp;
    float s = 0.f;
    if (o < nw) {
        const int m = o / K, k = o % K;
        s = strided_sum4(part + (size_t)m * Kp + k, (size_t)Mp * Kp, gg, G);
    } else if (o < total) {
        s = strided_sum4(partb + (o - nw), (size_t)Mp, gg, G);
    }
    red[gg][i] = s;
    __syncthreads();
    if (gg == 0 && o < total) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][i];
        if (o < nw) J.out[(size_t)(o / K) * J.ldw + J.col0 + (o % K)] += t;
        else J.db[o - nw] += t;
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm parameter gradients, second stage: dgamma[c] += sum_g part[g][c], dbeta[c] += sum_g part[g][H + c]  (fixed order: deterministic)
__global__ void __launch_bounds__(256) ln_grads_reduce_kernel(const float* __restrict__ part, int G, int H, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ float red[8][33];
    const int tid = threadIdx.x, i = tid & 31, gg = tid >> 5;
    const int o = blockIdx.x * 32 + i;  // < 2H
    float s = 0.f;
    for (int g = gg; g < G; g += 8) s += part[(size_t)g * 2 * H + o];
    red[gg][i] = s;
    __syncthreads();
    if (gg == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][i];
        if (o < H) dgamma[o] += t;
        else dbeta[o - H] += t;
    }
}

// bf16 x 3 operand images of many Linears in one launch (blockIdx.y = job).  A job describes a Linear with `ksub` outputs and
// `w_rows` inputs: element (o, k) = W[o][col0 + k] (fwd) or W[k][col0 + o] (the transposed block: the weight of the backward,
// input-gradient product dX = dZ . W).
__device__ __forceinline__ unsigned short bf16_rne_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ void __launch_bounds__(256) pack_b3_batch_kernel(PackTJobs J, float* __restrict__ base) {
    const PackTJob j = J.job[blockIdx.y];
    const int nkg = (j.w_rows + 15) / 16, njb = (j.ksub + 31) / 32;
    const int groups = nkg * njb;
    const int total = ((groups + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS) * B3_STAGE_GROUPS * 512;   // (group, lane, e)
    unsigned short* dst = reinterpret_cast<unsigned short*>(base + j.dst_off);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, g = idx >> 9;
        float v = 0.f;
        if (g < groups) {
            const int kg = g / njb, jb = g % njb;
            const int o = 32 * jb + (lane & 31), k = 16 * kg + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
            if (o < j.ksub && k < j.w_rows) v = j.fwd ? j.W[(int64_t)o * j.ld + j.col0 + k] : j.W[(int64_t)k * j.ld + j.col0 + o];
        }
        const unsigned short hi = bf16_rne_bits(v);
        const float r1 = v - __uint_as_float((unsigned)hi << 16);
        const unsigned short mid = bf16_rne_bits(r1);
        const unsigned short lo = bf16_rne_bits(r1 - __uint_as_float((unsigned)mid << 16));
        const size_t at = ((size_t)(g * 3) * 64 + lane) * 8 + e;
        dst[at] = hi;
        dst[at + 512] = mid;
        dst[at + 1024] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// segment sums over a CSR: out[i] = scale * sum_{p in [ptr[i], ptr[i+1])} rows[perm ? perm[p] : p] + cnt * shift
// ------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(256) segment_sum_kernel(const int* __restrict__ ptr, const int* __restrict__ perm,
                                                           const float* __restrict__ rows, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out, int n,
                                                           const int* __restrict__ ptr2, const int* __restrict__ perm2, float* __restrict__ out2) {
    // blockIdx.y = 1: the second sum of a pair over the same rows (the per-destination and per-source sums of dz1: one launch)
    if (blockIdx.y) { ptr = ptr2; perm = perm2; out = out2; }
    // One wave per node.  A row is read as 16-byte pieces (H / 4 lanes per row, so a wave step covers 64 / (H / 4) rows and every
    // request is a whole line), two steps in flight; the partial sums of the row slots are added in a fixed order at the end
    // (round 4: the 8-byte-per-lane form ran at 0.6 of this rate).  One order of additions: bit-reproducible.
    constexpr int H = 64 * W, LPR = H / 4, RPW = 64 / LPR;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, sub = lane / LPR, col = (lane % LPR) * 4;
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n) return;
    const int b = ptr[node], e = ptr[node + 1];
    f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    auto row = [&](int p) {
        const int64_t r = perm ? perm[p] : p;
        return *reinterpret_cast<const f4*>(rows + r * H + col);
    };
    int p = b + sub;
    for (; p + RPW < e; p += 2 * RPW) {
        const f4 v0 = row(p), v1 = row(p + RPW);
        s0 += v0;
        s1 += v1;
    }
    if (p < e) s0 += row(p);
    s0 += s1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // the RPW row slots, in slot order
        float v = s0[k];
        if (RPW >= 2) v += __shfl_xor(v, LPR, 64);
        if (RPW >= 4) v += __shfl_xor(v, 2 * LPR, 64);
        s0[k] = v;
    }
    if (sub == 0) {
        f4 o = s0;
        if (scale) {
            const f4 sc = *reinterpret_cast<const f4*>(scale + col), sh = *reinterpret_cast<const f4*>(shift + col);
            const float cnt = (float)(e - b);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = s0[k] * sc[k] + cnt * sh[k];
        }
        *reinterpret_cast<f4*>(out + (size_t)node * H + col) = o;
    }
}

__global__ void __launch_bounds__(256) swap_index_kernel(const int* __restrict__ src_sorted, int e, int64_t* __restrict__ ei2) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= e) return;
    ei2[p] = 0;                          // row 0: unused "source" role (any valid node index)
    ei2[(size_t)e + p] = src_sorted[p];  // row 1: segment key = source node of sorted position p
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int grid_tiles(int64_t rows) {
    int64_t t = cdiv(rows, TILE);
    if (t < 1) t = 1;
    return (int)(t < 2048 ? t : 2048);
}

template <typename Kern>
static int set_dyn_lds(Kern k, size_t bytes) {
    GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return GM_OK;
}

template <int H>
static int launch_train_fwd_h(int kind, const TrainFwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * B3_STAGE_FLOATS * 4 + (size_t)4 * TURN_FLOATS * 4;
    const int grid = grid_tiles(a.rows);
    switch (kind) {
        case TK_ENC_EDGE: hipLaunchKernelGGL((train_fwd_kernel<H, TK_ENC_EDGE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TK_ENC_NODE: hipLaunchKernelGGL((train_fwd_kernel<H, TK_ENC_NODE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TK_PROC_EDGE: hipLaunchKernelGGL((train_fwd_kernel<H, TK_PROC_EDGE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TK_PROC_NODE: hipLaunchKernelGGL((train_fwd_kernel<H, TK_PROC_NODE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TK_PROJ: hipLaunchKernelGGL((train_fwd_kernel<H, TK_PROJ>), dim3(grid), dim3(THREADS), lds, s, a); break;
        default: hipLaunchKernelGGL((train_fwd_kernel<H, TK_DEC>), dim3(grid), dim3(THREADS), lds, s, a); break;
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_train_fwd(int H, int kind, const TrainFwdArgs& a, hipStream_t s) {
    if (a.rows <= 0) return GM_OK;
    return H == 64 ? launch_train_fwd_h<64>(kind, a, s) : H == 128 ? launch_train_fwd_h<128>(kind, a, s) : launch_train_fwd_h<256>(kind, a, s);
}

static int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}
size_t train_bwd_ln_part_floats(int H) { return (size_t)2 * 1024 * 2 * H; }   // workgroups of a backward launch: <= 2 per CU

template <int H>
static int launch_train_bwd_h(int kind, const TrainBwdArgs& a_in, hipStream_t s, WgradBatch* wb) {
    const size_t lds = (size_t)2 * B3_STAGE_FLOATS * 4 + (size_t)4 * 2 * H * 4 + (size_t)4 * TURN_FLOATS * 4;
    int grid = grid_tiles(a_in.rows);
    TrainBwdArgs a = a_in;
    const bool batched = wb && a.ln_part && a.dgamma && a.dbeta;
    if (batched) {
        GM_REQUIRE(wb->jobs.n_ln < kWgLnMax, GM_ERR_INVALID_ARGUMENT, "launch_train_bwd: flush the weight-gradient batch first");
        a.ln_part = wb->ln_region(wb->jobs.n_ln);
    }
    const bool ln = a.ln_part && (kind == TB_ENC || kind == TB_EDGE || kind == TB_NODE);
    if (ln) {
        // as many workgroups as are resident at once, each walking its tiles: one partial row of the LayerNorm parameter sums each
        const int resident = device_cus() * (H <= 128 ? 2 : 1);
        if (grid > resident) grid = resident;
        GM_REQUIRE((size_t)grid * 2 * H <= train_bwd_ln_part_floats(H), GM_ERR_WORKSPACE, "launch_train_bwd: %d workgroups", grid);
    }
    switch (kind) {
        case TB_ENC: hipLaunchKernelGGL((train_bwd_kernel<H, TB_ENC>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TB_EDGE: hipLaunchKernelGGL((train_bwd_kernel<H, TB_EDGE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TB_NODE: hipLaunchKernelGGL((train_bwd_kernel<H, TB_NODE>), dim3(grid), dim3(THREADS), lds, s, a); break;
        case TB_PROJ: hipLaunchKernelGGL((train_bwd_kernel<H, TB_PROJ>), dim3(grid), dim3(THREADS), lds, s, a); break;
        default: hipLaunchKernelGGL((train_bwd_kernel<H, TB_DEC>), dim3(grid), dim3(THREADS), lds, s, a); break;
    }
    if (ln && batched) {
        WgLnJob& Lj = wb->jobs.ln[wb->jobs.n_ln++];
        Lj.part = a.ln_part; Lj.G = grid; Lj.H = H; Lj.dgamma = a.dgamma; Lj.dbeta = a.dbeta;
    } else if (ln && a.dgamma && a.dbeta) {
        hipLaunchKernelGGL(ln_grads_reduce_kernel, dim3(2 * H / 32), dim3(256), 0, s, a.ln_part, grid, H, a.dgamma, a.dbeta);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_train_bwd(int H, int kind, const TrainBwdArgs& a, hipStream_t s, WgradBatch* wb) {
    if (a.rows <= 0) return GM_OK;
    return H == 64 ? launch_train_bwd_h<64>(kind, a, s, wb) : H == 128 ? launch_train_bwd_h<128>(kind, a, s, wb) : launch_train_bwd_h<256>(kind, a, s, wb);
}

// Row chunks of a batch.  All jobs of a flush share one chunk length, chosen so that the whole batch is two rounds of resident
// workgroups (two per CU at a time: 194 VGPRs): every workgroup does the same share of the batch's rows, and the partial tiles the
// reduction reads are a third of what 512 chunks per job made them (measured at 2 x N = 5k: one round 99, two 102.4, three 101.8,
// four 100.0, six 98.0 training steps/s).  Chunks are multiples of 16 rows (one MFMA step), at least 64 rows.
static size_t wgrad_job_floats(int G, int Mp, int Kp) { return (size_t)G * Mp * Kp + (size_t)G * Mp; }
constexpr int kWgSlotsMax = 1024;   // resident workgroups the sizing may assume (the partial buffer is sized for it)
size_t wgrad_partial_floats(int H) {
    // a flush: sum over jobs of G * tiles <= kWgSlotsMax + kWgJobsMax * 4 tiles of 128 x 128 + the bias partials; then the LayerNorm region
    return ((size_t)kWgSlotsMax + kWgJobsMax * 4) * (128 * 128 + 128) + kWgLnMax * train_bwd_ln_part_floats(H);
}
void wgrad_batch_init(WgradBatch& b, float* part, int H, hipStream_t s) {
    b.part = part;
    b.ln_floats = train_bwd_ln_part_floats(H);
    b.cap = wgrad_partial_floats(H) - kWgLnMax * b.ln_floats;
    b.stream = s;
    b.jobs.n = 0;
    b.jobs.n_ln = 0;
}

int wgrad_flush(WgradBatch& b) {
    if (b.jobs.n <= 0 && b.jobs.n_ln <= 0) return GM_OK;
    int maxG = 1, maxT = 1, maxO = 1;
    for (int q = 0; q < b.jobs.n_ln; ++q) maxO = 2 * b.jobs.ln[q].H > maxO ? 2 * b.jobs.ln[q].H : maxO;
    if (b.jobs.n > 0) {
        int slots = 4 * device_cus();
        if (slots > kWgSlotsMax) slots = kWgSlotsMax;
        int64_t work = 0;
        for (int q = 0; q < b.jobs.n; ++q) work += (int64_t)b.jobs.job[q].rows * b.jobs.job[q].tiles;
        int64_t chunk = cdiv(cdiv(work, slots), 16) * 16;
        if (chunk < 64) chunk = 64;
        auto total = [&](int64_t c) { int64_t t = 0; for (int q = 0; q < b.jobs.n; ++q) t += cdiv(b.jobs.job[q].rows, c) * b.jobs.job[q].tiles; return t; };
        while (total(chunk) > slots) chunk += 16;    // the ceilings of the jobs' last chunks: a few steps at most
        size_t used = 0;
        for (int q = 0; q < b.jobs.n; ++q) {
            WgJob& j = b.jobs.job[q];
            j.chunk = (int)chunk;
            j.G = (int)cdiv(j.rows, chunk);
            j.part_off = used;
            used += wgrad_job_floats(j.G, j.Mp, j.Kp);
            maxG = j.G > maxG ? j.G : maxG;
            maxT = j.tiles > maxT ? j.tiles : maxT;
            const int outs = j.M * j.K + (j.db ? j.M : 0);
            maxO = outs > maxO ? outs : maxO;
        }
        GM_REQUIRE(used <= b.cap, GM_ERR_WORKSPACE, "wgrad: partial buffer too small (%zu > %zu floats)", used, b.cap);
        hipLaunchKernelGGL(wgrad_kernel, dim3(maxG, maxT, b.jobs.n), dim3(THREADS), 0, b.stream, b.jobs, b.part);
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(maxO, 32), b.jobs.n + b.jobs.n_ln), dim3(256), 0, b.stream, b.jobs, b.part);
    b.jobs.n = 0;
    b.jobs.n_ln = 0;
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int wgrad_enqueue(WgradBatch& b, const float* dz, int ldz, int M, const float* X, int ldx, int K, const int* xidx, int64_t rows, float* out,
                  int ldw, int col0, float* db) {
    if (rows <= 0 || M <= 0 || K <= 0) return GM_OK;
    WgJob j{};
    j.dz = dz; j.X = X; j.xidx = xidx; j.out = out; j.db = db;
    j.ldz = ldz; j.M = M; j.ldx = ldx; j.K = K; j.rows = (int)rows; j.ldw = ldw; j.col0 = col0;
    j.Mp = (int)cdiv(M, 128) * 128; j.Kp = (int)cdiv(K, 128) * 128;
    j.KT = j.Kp / 128; j.tiles = (j.Mp / 128) * j.KT;
    GM_REQUIRE(j.tiles <= 4, GM_ERR_UNSUPPORTED, "wgrad: a %d x %d weight block (at most 4 tiles of 128 x 128 per job)", M, K);
    // Indexed X rows are addressed with 32-bit byte offsets from the array's base (buffer addressing, wgrad_body): every row a
    // valid index can name must lie below 4 GiB.  xidx values are rows of X < `rows` for every caller (an edge permutation).
    GM_REQUIRE(!xidx || (uint64_t)rows * (uint64_t)ldx * 4u < (1ull << 32), GM_ERR_UNSUPPORTED,
               "wgrad: an indexed operand of %lld rows x %d floats exceeds the 4 GiB its 32-bit row offsets cover", (long long)rows, ldx);
    // A full batch is flushed first: the jobs it holds read operands that are still valid (they were produced by launches already
    // on the stream), so an early flush is always safe -- deep MLPs (num_layers >= 4) put more than kWgJobsMax jobs between the
    // model's own flush points.
    if (b.jobs.n >= kWgJobsMax) {
        const int rc = wgrad_flush(b);
        if (rc != GM_OK) return rc;
    }
    b.jobs.job[b.jobs.n++] = j;   // chunk, G and the partial offset are set when the batch is flushed
    return GM_OK;
}

int layer_stages_b3(int k, int out) {
    const int nkg = (k + 15) / 16, njb = (out + 31) / 32;
    return (nkg * njb + B3_STAGE_GROUPS - 1) / B3_STAGE_GROUPS;
}

int launch_pack_b3_batch(const PackTJobs& jobs, float* base, hipStream_t s) {
    if (jobs.n <= 0) return GM_OK;
    hipLaunchKernelGGL(pack_b3_batch_kernel, dim3(16, jobs.n), dim3(256), 0, s, jobs, base);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_segment_sum_pair(int H, const int* ptr, const int* perm, const int* ptr2, const int* perm2, const float* rows, const float* scale,
                            const float* shift, float* out, float* out2, int64_t n, hipStream_t s) {
    if (n <= 0) return GM_OK;
    const dim3 grid((unsigned)cdiv(n, 4), out2 ? 2 : 1);
    if (H == 64) hipLaunchKernelGGL((segment_sum_kernel<1>), grid, dim3(256), 0, s, ptr, perm, rows, scale, shift, out, (int)n, ptr2, perm2, out2);
    else if (H == 128) hipLaunchKernelGGL((segment_sum_kernel<2>), grid, dim3(256), 0, s, ptr, perm, rows, scale, shift, out, (int)n, ptr2, perm2, out2);
    else hipLaunchKernelGGL((segment_sum_kernel<4>), grid, dim3(256), 0, s, ptr, perm, rows, scale, shift, out, (int)n, ptr2, perm2, out2);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
int launch_segment_sum(int H, const int* ptr, const int* perm, const float* rows, const float* scale, const float* shift, float* out,
                       int64_t n, hipStream_t s) {
    return launch_segment_sum_pair(H, ptr, perm, nullptr, nullptr, rows, scale, shift, out, nullptr, n, s);
}

int launch_swap_index(const int* src_sorted, int64_t e, int64_t* ei2, hipStream_t s) {
    if (e <= 0) return GM_OK;
    hipLaunchKernelGGL(swap_index_kernel, dim3((unsigned)cdiv(e, 256)), dim3(256), 0, s, src_sorted, (int)e, ei2);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int train_kernels_init() {
    static PerDeviceOnce done_dev;
    return done_dev.run([]() -> int {
    const size_t lds = (size_t)2 * B3_STAGE_FLOATS * 4 + (size_t)4 * 2 * 256 * 4 + (size_t)4 * TURN_FLOATS * 4;   // weight ring + the backward kernels' LayerNorm sums + the line stores' turn tiles
    int rc = GM_OK;
#define GM_SET(k) if (rc == GM_OK) rc = set_dyn_lds(k, lds)
    GM_SET((train_fwd_kernel<64, TK_ENC_EDGE>)); GM_SET((train_fwd_kernel<64, TK_ENC_NODE>)); GM_SET((train_fwd_kernel<64, TK_PROC_EDGE>));
    GM_SET((train_fwd_kernel<64, TK_PROC_NODE>)); GM_SET((train_fwd_kernel<64, TK_DEC>)); GM_SET((train_fwd_kernel<64, TK_PROJ>));
    GM_SET((train_fwd_kernel<128, TK_PROJ>)); GM_SET((train_fwd_kernel<256, TK_PROJ>));
    GM_SET((train_bwd_kernel<64, TB_ENC>)); GM_SET((train_bwd_kernel<64, TB_EDGE>)); GM_SET((train_bwd_kernel<64, TB_NODE>)); GM_SET((train_bwd_kernel<64, TB_DEC>));
    GM_SET((train_bwd_kernel<64, TB_PROJ>));
    GM_SET((train_fwd_kernel<128, TK_ENC_EDGE>)); GM_SET((train_fwd_kernel<128, TK_ENC_NODE>)); GM_SET((train_fwd_kernel<128, TK_PROC_EDGE>));
    GM_SET((train_fwd_kernel<128, TK_PROC_NODE>)); GM_SET((train_fwd_kernel<128, TK_DEC>));
    GM_SET((train_fwd_kernel<256, TK_ENC_EDGE>)); GM_SET((train_fwd_kernel<256, TK_ENC_NODE>)); GM_SET((train_fwd_kernel<256, TK_PROC_EDGE>));
    GM_SET((train_fwd_kernel<256, TK_PROC_NODE>)); GM_SET((train_fwd_kernel<256, TK_DEC>));
    GM_SET((train_bwd_kernel<128, TB_ENC>)); GM_SET((train_bwd_kernel<128, TB_EDGE>)); GM_SET((train_bwd_kernel<128, TB_NODE>)); GM_SET((train_bwd_kernel<128, TB_DEC>));
    GM_SET((train_bwd_kernel<128, TB_PROJ>)); GM_SET((train_bwd_kernel<256, TB_PROJ>));
    GM_SET((train_bwd_kernel<256, TB_ENC>)); GM_SET((train_bwd_kernel<256, TB_EDGE>)); GM_SET((train_bwd_kernel<256, TB_NODE>)); GM_SET((train_bwd_kernel<256, TB_DEC>));
#undef GM_SET
    return rc;
    });
}

}  // namespace gm
