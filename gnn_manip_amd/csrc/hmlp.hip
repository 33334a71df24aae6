// Fused MLP kernels of the encode-process-decode model on the fp16 matrix pipe with fp32 accuracy, for hidden sizes
// 64 / 128 / 256 and any num_layers >= 2 (epd_gnn.py:63-84 builds num_layers + 1 Linears for any value; train_dyn.py:237-238
// exposes both as options):
//   hm_edge_kernel<H, ENC>   encoder phi_e (ENC) or processor phi_e + LayerNorm + residual + scatter-add (epd_gnn.py:35-46,100-105)
//   hm_node_kernel<H, MODE>  encoder phi_v (0), processor phi_v (1), projection only (2); tails: the next step's
//                            P = h [W_i | W_j]^T + b1, or the decoder MLP (epd_gnn.py:47-48,107)
//
// Arithmetic: hmma_dev.h (two-way fp16 operand split, three exact partial products per multiply, fp32 accumulation).
// Power-of-two scales ride through an MLP's chain of Linears (hmlp.h): accumulators of Linear l are U_l z_l, ReLU passes them on
// as the next operand image, the scale leaves in the LayerNorm statistics / at the outputs.  The encoders scale every raw
// feature row by its own power of two (row maximum to [2^6, 2^7)), which rides along the same way.  Every value written
// to an operand image is range-checked (|x| < 65504): a violation sets ERRF_SPLIT_RANGE in the forward's CSR header.
//
// Structure.  One 8-wave workgroup per CU walks over tiles of M rows with M * H = 32768: the fp16 hi / lo operand image of
// a tile's activations is exactly 128 KiB of LDS.  Wave (rg, jb) owns output block jb (32 features) of the 128 rows of row
// group rg: 4 accumulator tiles.  Per k-group it reads its A fragments (weights: 1 KiB, wave-private, prefetched two
// k-groups ahead straight from L2 -- the weights of a Linear do not fit registers or LDS next to the image at H = 256) and
// 8 B fragments from the image, and issues 12 MFMAs.  Between Linears: barrier, accumulators -> ReLU -> split -> image,
// barrier.  LayerNorm statistics are merged across the waves of a row through LDS (parallel-variance merge); outputs,
// the residual and the scatter-add work directly on the accumulator layout (16-byte pieces of the rows).
//
// The scatter-add is the one of hedge.hip: segmented DPP scan over the destination-sorted rows of a wave's 4-block group, one
// row store per finished segment -- to agg, or, for the piece of a segment that began in an earlier group, to the group's
// row of the side buffer; the node kernel adds those pieces in group order when it reads agg (hedge.h: head / stitch
// lists).  No atomics: one summation order, the same alone or inside a batch (groups are aligned to each graph's first block).
//
// Built with -fno-slp-vectorize like hedge.hip.
#include "common.h"
#include "mlp.h"
#include "hedge.h"
#include "hmlp.h"
#include "hmma_dev.h"

namespace gm {

namespace {

#ifdef HM_STAMPS
// development build only: s_memtime stamps of wave 0 / lane 0 of four workgroups at the phase boundaries of their first four
// tiles -- of the node kernels, or (-DHM_STAMPS_EDGE) of the edge kernels; read back with gm_debug_hm_stamps (tools/hm_stamps.py)
__device__ unsigned long long g_hm_stamps[4 * 4 * 16];
#define HM_STAMP_DO(tile_i, slot)                                                                                      \
    do {                                                                                                               \
        if ((blockIdx.x & 63) == 0 && blockIdx.x < 256 && threadIdx.x == 0 && (tile_i) < 4)                              \
            g_hm_stamps[((blockIdx.x >> 6) * 4 + (tile_i)) * 16 + (slot)] = __builtin_readcyclecounter();             \
    } while (0)
#endif
#if defined(HM_STAMPS) && !defined(HM_STAMPS_EDGE)
#define HM_STAMP(tile_i, slot) HM_STAMP_DO(tile_i, slot)
#else
#define HM_STAMP(tile_i, slot) do { } while (0)
#endif
#if defined(HM_STAMPS) && defined(HM_STAMPS_EDGE)
#define HM_STAMP_E(tile_i, slot) HM_STAMP_DO(tile_i, slot)
#else
#define HM_STAMP_E(tile_i, slot) do { } while (0)
#endif
// HM_LINES: the edge kernel's e + e' rows (and the residual rows they are added to) move as whole 128-byte lines: a wave's
// 32 rows x 32 features take a turn through a wave-private tile in the (then idle) image region, after which an instruction
// covers 8 rows x 128 bytes instead of 32 rows x 32 bytes.  Rows in the caller's order (eid_out) keep the piece form.
#ifndef HM_LINES
#define HM_LINES 1
#endif
constexpr int HM_TURN_LD = 36;                       // floats per row of the turn tile
constexpr int HM_TURN_FLOATS = 32 * HM_TURN_LD;
constexpr int HM_THREADS = 512;
constexpr int HM_WAVES = 8;
constexpr int BE = 32;
constexpr size_t HM_IMG_BYTES = 131072;
constexpr size_t HM_ST_BYTES = 16384;    // [NRB][2 NJB partials][32 rows] float2  (NRB * 2 NJB = 64)
constexpr size_t HM_BLK_BYTES = 2 * 16 * sizeof(int2);
constexpr size_t HM_GB_BYTES = 2 * 256 * sizeof(float);    // gamma | beta
constexpr size_t HM_RS_BYTES = 32 * 32 * sizeof(float);    // encoders: power-of-two scale of every row of the tile
constexpr size_t HM_LDS_BYTES = HM_IMG_BYTES + HM_ST_BYTES + HM_BLK_BYTES + HM_GB_BYTES + HM_RS_BYTES;
// node kernels with RBW row blocks per wave: image and statistics shrink with the tile.  (Two workgroups per CU in the one-block
// form, which then fits twice: measured +-0 at N = 100k; a two-block form needs > 128 VGPRs.)
constexpr size_t hm_node_img_bytes(int rbw) { return HM_IMG_BYTES * rbw / 4; }
constexpr size_t hm_node_lds_bytes(int rbw) { return hm_node_img_bytes(rbw) + HM_ST_BYTES * rbw / 4 + HM_BLK_BYTES + HM_GB_BYTES + HM_RS_BYTES; }

// RBW: 32-row blocks per wave.  4 fills the LDS image (M * H = 32768); the node kernels also come with 1 for small
// graphs (four times the tiles; per-row arithmetic is identical, so results do not depend on the choice).
template <int H, int RBW = 4>
struct Cfg {
    static constexpr int NJB = H / 32;           // output blocks = waves per row group
    static constexpr int KS = H / 16;            // k-groups of an H-wide input
    static constexpr int NRG = HM_WAVES / NJB;   // row groups
    static constexpr int NRB = RBW * NRG;        // 32-row blocks per tile
    static constexpr int M = 32 * NRB;           // rows per tile
};

struct Lin {
    float t, inv_u, u;   // this Linear's weight scale; 1 / U and U, the scale its accumulators carry (hmlp.h)
    float cap;           // encoders' first Linear: largest per-row input scale
    const float* bias;   // pre-multiplied by U
    const half8* frag;
};
__device__ __forceinline__ Lin lin_at(const float* p, int out_pad) {
    Lin L;
    L.t = p[0];
    L.inv_u = p[1];
    L.u = p[2];
    L.cap = p[3];
    L.bias = p + 4;
    L.frag = reinterpret_cast<const half8*>(p + 4 + out_pad);
    return L;
}

// accumulators of the wave's 4 row blocks <- bias (already scaled) of output block jbv
template <int RBW>
__device__ __forceinline__ void init_bias(floatx16 (&acc)[RBW], const float* bias, int jbv, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(bias + 32 * jbv + 8 * g + 4 * hi);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[rb][4 * g + tt] = v[tt];
    }
}
// the same with the per-row power of two of the encoders' raw features (rs[rb]: this lane's row of row block rb)
template <int RBW>
__device__ __forceinline__ void init_bias_rows(floatx16 (&acc)[RBW], const float* bias, int jbv, int hi, const float (&rs)[RBW]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(bias + 32 * jbv + 8 * g + 4 * hi);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[rb][4 * g + tt] = v[tt] * rs[rb];
    }
}

// Range check of the fp16 split (hmma_dev.h).  A value of magnitude >= 65520 becomes (inf, -inf) as a pair, and every accumulator of
// its row that sums over it becomes NaN (inf - inf) -- the whole row, all output features.  So one accumulator per row and
// Linear tells: a comparison per row block instead of a maximum over every value written.  The verdict is wave-uniform and
// lives in a scalar register.  (A non-finite input of the caller's raises the same flag.)
template <int RBW>
__device__ __forceinline__ void check_rows(int& bad, const floatx16 (&acc)[RBW]) {
    bool nan = false;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) nan |= acc[rb][0] != acc[rb][0];
    bad |= __any(nan) ? 1 : 0;
}
// Encoder forms (raw feature rows in; the standalone GraphIndependent entry point has no header to flag): a row whose accumulators
// went NaN in ANY Linear -- a non-finite input, or a range violation further on -- must come out NaN, but the ReLU flushes a NaN
// operand to zero.  pois[rb] stays 0 for a clean row and turns NaN for good (x - x: 0 for finite x, NaN for NaN / inf); it joins
// the LayerNorm's additive term.
template <int RBW>
__device__ __forceinline__ void mark_rows(float (&pois)[RBW], const floatx16 (&acc)[RBW]) {
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) pois[rb] += acc[rb][0] - acc[rb][0];
}
__device__ __forceinline__ void report_range(int bad, int* flags) {
    if (flags && bad && (threadIdx.x & 63) == 0) atomicOr(flags, ERRF_SPLIT_RANGE);
}

// acc[rb] += W[jb block, k-groups ks0 .. ks0 + nks) x image rows of row block rb.
//   wf : this wave's fragments of the first k-group, + lane  (k-group stride 128 half8)
//   im : image of the wave's first row block at k-group 0, + lane  (row-block stride img_ksn * 128, k-group stride 128)
// nrb: row blocks of this wave that hold rows (the others' MFMAs are skipped: the last tile of a workgroup's range may be partial;
// a branch-free copy of the loop for full tiles was measured: -4 %, it costs registers the kernels do not have)
#ifndef HM_SHORT_TILE_ROWS
#define HM_SHORT_TILE_ROWS 1
#endif
#ifndef HM_RING4
#define HM_RING4(RBW) ((RBW) == 1)
#endif
// One k-group: the three partial products of every row block (lo.hi + hi.lo + hi.hi).
template <int RBW>
__device__ __forceinline__ void mfma3(floatx16 (&acc)[RBW], const half8& ah, const half8& al, const half8 (&bh)[RBW], const half8 (&bl)[RBW], int nrb) {
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[rb], acc[rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[rb], acc[rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[rb], acc[rb], 0, 0, 0);
}

// Fragment loads of the GEMM loop.  Every load is UNCONDITIONAL, from a clamped (always valid) k-group: a load inside a
// branch makes hipcc wait with vmcnt(0) at the join, which also drains the prefetches issued since -- the loop then runs at
// one L2 round trip per trip (that is what the round-2 form of this loop did: 3.3 ms per launch at hidden 256).  With
// straight-line loads the waits are counted.  The requests are pinned where they are written (sched_barrier): under
// register pressure the scheduler otherwise sinks every load to its first use.  The weights come from L2 (~1 us under
// load), so the A fragments run a ring of 4 k-groups: a k-group is requested one and a half trips (36 MFMAs of this wave,
// as many of the SIMD's other wave) before its use.  The redundant loads of the last trips re-read the last k-group.
template <int RBW>
__device__ __forceinline__ void gemm(floatx16 (&acc)[RBW], const half8* __restrict__ wf, const half8* im, int img_ksn, int nks, int nrb = RBW) {
    const int last = nks - 1;
    half8 bh[RBW], bl[RBW], ch[RBW], cl[RBW];
    auto load_b = [&](half8 (&h)[RBW], half8 (&l)[RBW], int kg) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) { h[rb] = im[(rb * img_ksn + kg) * 128]; l[rb] = im[(rb * img_ksn + kg) * 128 + 64]; }
    };
    if (HM_RING4(RBW) && nks >= 4 && (nks & 3) == 0) {
        half8 ah[4], al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ah[q] = wf[q * 128]; al[q] = wf[q * 128 + 64]; }
        load_b(bh, bl, 0);
        load_b(ch, cl, 1);
#pragma unroll 1
        for (int ks = 0; ks < nks; ks += 4) {
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                mfma3(acc, ah[u], al[u], bh, bl, nrb);
                __builtin_amdgcn_sched_barrier(0);
                { const int ka = min(ks + u + 4, last); ah[u] = wf[ka * 128]; al[u] = wf[ka * 128 + 64]; }
                load_b(bh, bl, min(ks + u + 2, last));
                __builtin_amdgcn_sched_barrier(0);
                mfma3(acc, ah[u + 1], al[u + 1], ch, cl, nrb);
                __builtin_amdgcn_sched_barrier(0);
                { const int ka = min(ks + u + 5, last); ah[u + 1] = wf[ka * 128]; al[u + 1] = wf[ka * 128 + 64]; }
                load_b(ch, cl, min(ks + u + 3, last));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    // short inputs (the encoders' first Linear: 1 or 2 k-groups)
    half8 ah0 = wf[0], al0 = wf[64];
    half8 ah1 = wf[min(1, last) * 128], al1 = wf[min(1, last) * 128 + 64];
    load_b(bh, bl, 0);
    load_b(ch, cl, min(1, last));
#pragma unroll 1
    for (int ks = 0; ks < nks; ks += 2) {
        const int k2 = min(ks + 2, last), k3 = min(ks + 3, last);
        mfma3(acc, ah0, al0, bh, bl, nrb);
        __builtin_amdgcn_sched_barrier(0);
        ah0 = wf[k2 * 128]; al0 = wf[k2 * 128 + 64];
        load_b(bh, bl, k2);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < nks) mfma3(acc, ah1, al1, ch, cl, nrb);   // MFMAs only inside the branch (odd nks: the 16-wide edge features)
        __builtin_amdgcn_sched_barrier(0);
        ah1 = wf[k3 * 128]; al1 = wf[k3 * 128 + 64];
        load_b(ch, cl, k3);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators -> [ReLU] -> fp16 hi / lo -> the wave's two k-groups of the image of row block rb.  The ReLU form passes the
// accumulators' scale on (no multiply); the other form (a LayerNorm output entering a tail) writes the values as they are.
template <bool RELU>
__device__ __forceinline__ void acc_to_img(const floatx16& a, uintx4* img_rb, int jb, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = RELU ? relu(a[8 * q + j]) : a[8 * q + j];
        uintx2 h0, l0, h1, l1;
        split4(v[0], v[1], v[2], v[3], h0, l0);
        split4(v[4], v[5], v[6], v[7], h1, l1);
        img_rb[((2 * jb + q) * 2 + 0) * 64 + lane] = uintx4{h0[0], h0[1], h1[0], h1[1]};
        img_rb[((2 * jb + q) * 2 + 1) * 64 + lane] = uintx4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// rows of H floats -> image (k-group count KS).  16 RBW units of (16 rows, 2 k-groups) per tile, 2 RBW per wave; lane = (row, k-group
// parity, kg) reads the two 16-byte pieces that make its 8 K slots.  row_of(rbg, n) returns the source row (or -1: zeros).
template <int H, int RBW, int UNR, class R, class Z>
__device__ __forceinline__ void rows_to_image(const float* __restrict__ src, uintx4* img, int wave, int lane, R&& row_of, Z&& adjust) {
    using C = Cfg<H, RBW>;
#pragma unroll UNR
    for (int it = 0; it < 2 * RBW; ++it) {
        const int u = it * 8 + wave;
        const int rowhalf = u % (2 * C::NRB), kspair = u / (2 * C::NRB);
        const int rbg = rowhalf >> 1, nn = 16 * (rowhalf & 1) + (lane & 15), c = lane >> 4, ks = 2 * kspair + (c >> 1), kg = c & 1;
        const long long row = row_of(rbg, nn);
        const float* p = src + row * H + 16 * ks + 4 * kg;
        floatx4 v0 = *reinterpret_cast<const floatx4*>(p);
        floatx4 v1 = *reinterpret_cast<const floatx4*>(p + 8);
        adjust(row, 16 * ks + 4 * kg, v0, v1);
        uintx2 h0, l0, h1, l1;
        split4(v0[0], v0[1], v0[2], v0[3], h0, l0);
        split4(v1[0], v1[1], v1[2], v1[3], h1, l1);
        img[((rbg * C::KS + ks) * 2 + 0) * 64 + nn + 32 * kg] = uintx4{h0[0], h0[1], h1[0], h1[1]};
        img[((rbg * C::KS + ks) * 2 + 1) * 64 + nn + 32 * kg] = uintx4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// The agg rows of a node tile -> image.  Per batch of pieces: every row piece is requested first, together with the piece of the
// first head partial another group holds of the row's segment (hedge.h; SC[slot] = 2 g + more for the first such group g of
// the row, or -1, looked up at the start of the tile and resolved under the first GEMM -- rows without one read side row 0,
// a cache hit, and drop it); then the pieces are combined, in group order, and split into the image.  `more` (the next
// group holds a partial of the same row too: in-degree beyond a group's 128 edges) takes the loop.
// (The first form waited for a stitch[] and a head[] look-up per piece, two pieces at a time: a third of the tile's time.)
template <int H, int RBW, class R>
__device__ __forceinline__ void agg_rows_to_image(const float* __restrict__ agg, uintx4* img, int wave, int lane, R&& row_of, const int* SC,
                                                  const float* __restrict__ side, const int* __restrict__ head, int ng) {
    using C = Cfg<H, RBW>;
    constexpr int NB = RBW >= 2 ? RBW : 2;   // pieces per batch: bounds the registers held by requests in flight
#pragma unroll
    for (int b0 = 0; b0 < 2 * RBW; b0 += NB) {
        floatx4 v0[NB], v1[NB], s0[NB], s1[NB];
        long long rows[NB];
        int gs[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int u = (b0 + q) * 8 + wave;
            const int rowhalf = u % (2 * C::NRB);
            gs[q] = SC ? SC[(rowhalf >> 1) * 32 + 16 * (rowhalf & 1) + (lane & 15)] : -1;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int u = (b0 + q) * 8 + wave;
            const int rowhalf = u % (2 * C::NRB), kspair = u / (2 * C::NRB);
            const int rbg = rowhalf >> 1, nn = 16 * (rowhalf & 1) + (lane & 15), c = lane >> 4, ks = 2 * kspair + (c >> 1), kg = c & 1;
            rows[q] = row_of(rbg, nn);
            const float* p = agg + rows[q] * H + 16 * ks + 4 * kg;
            v0[q] = *reinterpret_cast<const floatx4*>(p);
            v1[q] = *reinterpret_cast<const floatx4*>(p + 8);
            if (SC) {
                const float* sp = side + (size_t)(gs[q] < 0 ? 0 : gs[q] >> 1) * H + 16 * ks + 4 * kg;
                s0[q] = *reinterpret_cast<const floatx4*>(sp);
                s1[q] = *reinterpret_cast<const floatx4*>(sp + 8);
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int u = (b0 + q) * 8 + wave;
            const int rowhalf = u % (2 * C::NRB), kspair = u / (2 * C::NRB);
            const int rbg = rowhalf >> 1, nn = 16 * (rowhalf & 1) + (lane & 15), c = lane >> 4, ks = 2 * kspair + (c >> 1), kg = c & 1;
            if (SC) {
                // rows without a head partial read side row 0 for nothing: dropped by a SELECT, not by a multiplication with zero --
                // that row belongs to no group (group 0 never has a head partial) and holds whatever the workspace held: NaN x 0
                // is NaN, and a NaN operand is then flushed to zero by the ReLU (a wrong, finite result)
                const bool has = gs[q] >= 0;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { v0[q][tt] += has ? s0[q][tt] : 0.f; v1[q][tt] += has ? s1[q][tt] : 0.f; }
                if (gs[q] >= 0 && (gs[q] & 1)) {
                    int g = (gs[q] >> 1) + 1;
                    do {
                        const float* sp = side + (size_t)g * H + 16 * ks + 4 * kg;
                        const floatx4 t0 = *reinterpret_cast<const floatx4*>(sp), t1 = *reinterpret_cast<const floatx4*>(sp + 8);
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) { v0[q][tt] += t0[tt]; v1[q][tt] += t1[tt]; }
                        ++g;
                    } while (g < ng && head[g] == (int)rows[q]);
                }
            }
            uintx2 h0, l0, h1, l1;
            split4(v0[q][0], v0[q][1], v0[q][2], v0[q][3], h0, l0);
            split4(v1[q][0], v1[q][1], v1[q][2], v1[q][3], h1, l1);
            img[((rbg * C::KS + ks) * 2 + 0) * 64 + nn + 32 * kg] = uintx4{h0[0], h0[1], h1[0], h1[1]};
            img[((rbg * C::KS + ks) * 2 + 1) * 64 + nn + 32 * kg] = uintx4{l0[0], l0[1], l1[0], l1[1]};
        }
    }
}

// rows of k1 <= 16 KSN raw features -> image with KSN k-groups (zero-padded), every row scaled by its own power of two so that
// its largest magnitude lands in [2^6, 2^7), but by no more than `cap` (pack_hm_kernel: what keeps the bias in range; an
// all-zero row takes the cap).  RS[rbg * 32 + n] receives the scale.
template <int H, int RBW, int KSN, class R>
__device__ __forceinline__ void narrow_rows_to_image(const float* __restrict__ src, int k1, uintx4* img, float* RS, float cap, int tid, R&& row_of) {
    using C = Cfg<H, RBW>;
    for (int i = tid; i < C::NRB * KSN * 64; i += HM_THREADS) {
        const int rbg = i / (KSN * 64), rem = i % (KSN * 64), ks = rem >> 6, l = rem & 63, nn = l & 31, kg = l >> 5;
        const long long row = row_of(rbg, nn);
        const float* p = src + row * k1;
        float mx = 0.f;
        for (int f = 0; f < k1; ++f) mx = fmaxf(mx, fabsf(p[f]));   // the whole row (L1-resident): every lane of the row agrees
        float sc = cap;                       // zero (or non-finite) row: bias only
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);            // mx = f 2^ex, f in [0.5, 1)  ->  mx 2^(7 - ex) in [2^6, 2^7)
            sc = fminf(ldexpf(1.f, min(7 - ex, 100)), cap);
        }
        if (ks == 0 && kg == 0) RS[rbg * 32 + nn] = sc;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = 16 * ks + 8 * (j >> 2) + 4 * kg + (j & 3);
            v[j] = f < k1 ? p[f] * sc : 0.f;
        }
        uintx2 h0, l0, h1, l1;
        split4(v[0], v[1], v[2], v[3], h0, l0);
        split4(v[4], v[5], v[6], v[7], h1, l1);
        img[((rbg * KSN + ks) * 2 + 0) * 64 + l] = uintx4{h0[0], h0[1], h1[0], h1[1]};
        img[((rbg * KSN + ks) * 2 + 1) * 64 + l] = uintx4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// LayerNorm statistics of the wave's rows.  The Linear in front of a LayerNorm is packed centred over its output features
// (hmlp.h: PackHmJob::center), so a row's accumulators have zero mean and its variance is its mean square -- over the features
// that exist: the zero-padded ones are exactly zero and add nothing, whatever the hidden size.  Each lane publishes the sum of
// squares of its 16 accumulator values per row block; after the barrier every wave adds the 2 NJB partials of its rows:
// x_hat = acc * k (+ m = 0).
template <int H, int RBW>
__device__ __forceinline__ void ln_publish(const floatx16 (&acc)[RBW], float* ST, int rg, int jb, int n, int hi) {
    using C = Cfg<H>;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) q = fmaf(acc[rb][r], acc[rb][r], q);
        ST[(((RBW * rg + rb) * 2 * C::NJB) + 2 * jb + hi) * BE + n] = q;
    }
}
// hv: the features that exist (<= H)
template <int H>
__device__ __forceinline__ void ln_merge(const float* ST, int rbg, int n, float t, float eps, int hv, float& k, float& m) {
    using C = Cfg<H>;
    constexpr int NP = 2 * C::NJB;
    const float* st = ST + rbg * NP * BE + n;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < NP; ++w) m2 += st[w * BE];
    m2 *= 1.0f / (float)hv;
    // accumulators carry the scale t (= U of the chain, times the row's own scale in the encoders):
    // acc / sqrt(var_acc + eps t^2) is the normalised value
    const float v = fmaf(eps * t, t, m2);
    float r = __builtin_amdgcn_rsqf(v);   // 1 ulp + one Newton step
    r = r * fmaf(-0.5f * v * r, r, 1.5f);
    k = r;
    m = 0.f;
}

// ------------------------------------------------------------------------------------------
// edge kernel
// ------------------------------------------------------------------------------------------
template <int H, bool ENC>
__global__ void __launch_bounds__(HM_THREADS, 1) hm_edge_kernel(HmEdgeArgs A) {
    using C = Cfg<H>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uintx4* img = reinterpret_cast<uintx4*>(smem);
    const half8* imgh = reinterpret_cast<const half8*>(smem);
    float* ST = reinterpret_cast<float*>(smem + HM_IMG_BYTES);
    int2* s_blk = reinterpret_cast<int2*>(smem + HM_IMG_BYTES + HM_ST_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jb = wave % C::NJB, rg = wave / C::NJB;
    const int n = lane & 31, hi = lane >> 5;
    const int E = A.hdr ? A.hdr->n_edges : A.n_edges_host;
    if (E <= 0) return;
    const int nblk = ENC ? (E + BE - 1) / BE : A.tab->n_blocks;
    const int ntiles = (nblk + C::NRB - 1) / C::NRB;
    constexpr int KS0 = ENC ? 1 : C::KS;   // k-groups of the first Linear's input image
    const size_t lin0 = hm_lin_floats(H, 16 * KS0), linh = hm_lin_floats(H, H);

    float* GB = reinterpret_cast<float*>(smem + HM_IMG_BYTES + HM_ST_BYTES + HM_BLK_BYTES);   // gamma[H] | beta[H]
    float* RS = reinterpret_cast<float*>(smem + HM_IMG_BYTES + HM_ST_BYTES + HM_BLK_BYTES + HM_GB_BYTES);   // encoder: row scales
    for (int i = tid; i < H; i += HM_THREADS) { GB[i] = A.ln_g[i]; GB[H + i] = A.ln_b[i]; }
    const float* gamp = GB + 32 * jb + 4 * hi;
    const float* betp = GB + H + 32 * jb + 4 * hi;
    int rng = 0;   // range check of the fp16 split: set once a value written to an operand image does not fit

    int iter = 0;
#pragma unroll 1
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++iter) {
        int2* sb = s_blk + (iter & 1) * 16;
        HM_STAMP_E(iter, 0);
        if (tid < C::NRB) {
            const int b = t * C::NRB + tid;
            int2 e = make_int2(E, 0);
            if (b < nblk) {
                if (ENC) e = make_int2(b * BE, min(BE, E - b * BE));
                else { e = A.blk[b]; e.y &= 0xff; }
            }
            sb[tid] = e;
        }
        __syncthreads();
        // source position of row n of row block rbg (rows past a block's end repeat its last row; their results are dropped)
        auto pos_of = [&](int rbg, int nn) -> int {
            const int2 bi = sb[rbg];
            const int p = bi.x + (nn < bi.y ? nn : bi.y - 1);
            return p < 0 ? 0 : p;
        };
        auto in_row = [&](int rbg, int nn) -> long long {
            const int p = pos_of(rbg, nn);
            return A.eid ? A.eid[p] : p;
        };
        if (ENC) narrow_rows_to_image<H, 4, 1>(A.e_in, A.k1, img, RS, A.w[3], tid, in_row);
        else rows_to_image<H, 4, 8>(A.e_in, img, wave, lane, in_row, [](long long, int, floatx4&, floatx4&) {});
        __syncthreads();
        HM_STAMP_E(iter, 1);   // e rows in the image

        const float* wp = A.w;
        Lin L = lin_at(wp, H);
        floatx16 acc[4];
        int pe[4], dn[4];
        float rs[4] = {1.f, 1.f, 1.f, 1.f};   // encoder: scale of this lane's row of every row block (rides through the chain)
        if (ENC) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) rs[rb] = RS[(4 * rg + rb) * 32 + n];
            init_bias_rows(acc, L.bias, jb, hi, rs);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) { pe[rb] = pos_of(4 * rg + rb, n); dn[rb] = 0; }
        } else {
            // layer-1 factorisation: P_i[dst] (+ b1) + P_j[src], brought to the weight scale of the e-block
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int p = pos_of(4 * rg + rb, n);
                pe[rb] = p;
                const int d = A.dst[p], sr = A.src[p];
                dn[rb] = d;
                const float* pi = A.P + (size_t)d * 2 * H + 32 * jb + 4 * hi;
                const float* pj = A.P + (size_t)sr * 2 * H + H + 32 * jb + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 a = *reinterpret_cast<const floatx4*>(pi + 8 * g);
                    const floatx4 b = *reinterpret_cast<const floatx4*>(pj + 8 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[rb][4 * g + tt] = (a[tt] + b[tt]) * L.u;
                }
            }
        }
        HM_STAMP_E(iter, 2);   // P gather issued (accumulators initialised: waits land in the GEMM)
        gemm(acc, L.frag + (size_t)jb * KS0 * 128 + lane, imgh + (size_t)(4 * rg) * KS0 * 128 + lane, KS0, KS0);
        check_rows(rng, acc);
        float pois[4] = {0.f, 0.f, 0.f, 0.f};
        if (ENC) mark_rows(pois, acc);
        wp += lin0;
        HM_STAMP_E(iter, 3);   // GEMM 1
#pragma unroll 1
        for (int l = 1; l <= A.nl; ++l) {
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc_to_img<true>(acc[rb], img + (size_t)(4 * rg + rb) * C::KS * 128, jb, lane);
            __syncthreads();
            L = lin_at(wp, H);
            if (ENC) init_bias_rows(acc, L.bias, jb, hi, rs);
            else init_bias(acc, L.bias, jb, hi);
            gemm(acc, L.frag + (size_t)jb * C::KS * 128 + lane, imgh + (size_t)(4 * rg) * C::KS * 128 + lane, C::KS, C::KS);
            check_rows(rng, acc);
            if (ENC) mark_rows(pois, acc);
            wp += linh;
        }
        HM_STAMP_E(iter, 4);   // hidden Linears
        // LayerNorm
        ln_publish<H, 4>(acc, ST, rg, jb, n, hi);
        __syncthreads();
        HM_STAMP_E(iter, 5);   // statistics exchanged
        float carry[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) carry[r] = 0.f;
        int prev_last = -3, head = -1;
        // the image is idle from here to the next tile's barrier: wave-private turn tiles at its start
        const bool lines = HM_LINES && !ENC && !A.eid_out && !A.discard_e_out;
        float* turn = reinterpret_cast<float*>(smem) + wave * HM_TURN_FLOATS;
        float* t_acc = turn + n * HM_TURN_LD + 4 * hi;                       // this lane's pieces of row n: + 8 g
        float* t_row = turn + (lane >> 3) * HM_TURN_LD + 4 * (lane & 7);      // row-major: row (lane >> 3) + 8 j: + 8 j HM_TURN_LD
        const unsigned v_line = (unsigned)((lane >> 3) * H + 4 * (lane & 7)) * 4u;
        const int grp = t * C::NRG + rg;   // this wave's group of 4 blocks
        if (!ENC && A.agg && grp < A.tab->n_groups) head = A.head[grp];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int rbg = 4 * rg + rb;
            const int cnt = sb[rbg].y;
            float k, m;
            ln_merge<H>(ST, rbg, n, L.u * rs[rb], A.eps, A.h_valid, k, m);
            if (ENC) m += pois[rb];
            const bool valid = n < cnt;
            const int p = pe[rb];
            const long long orow = A.eid_out ? A.eid_out[p] : p;
            float* outp = A.e_out + orow * H + 32 * jb + 4 * hi;
            if (ENC || !A.agg) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 y;
                    const floatx4 gm4 = *reinterpret_cast<const floatx4*>(gamp + 8 * g), bt4 = *reinterpret_cast<const floatx4*>(betp + 8 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) y[tt] = fmaf(fmaf(acc[rb][4 * g + tt], k, m), gm4[tt], bt4[tt]);
                    if (valid) {
                        if (!ENC && A.residual) {
                            const floatx4 e0 = *reinterpret_cast<const floatx4*>(outp + 8 * g);
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) y[tt] += e0[tt];
                        }
                        *reinterpret_cast<floatx4*>(outp + 8 * g) = y;
                    }
                }
            } else {
                const int dnv = valid ? dn[rb] : -1 - n;
                const int nxv = (valid && p + 1 < E) ? A.dst[p + 1] : -2;
                float f1, f2, f4, f8, fb, fc;
                {
                    const int p1 = __builtin_amdgcn_update_dpp(-1000000, dnv, 0x111, 0xf, 0xf, false);
                    const int p2 = __builtin_amdgcn_update_dpp(-1000000, dnv, 0x112, 0xf, 0xf, false);
                    const int p4 = __builtin_amdgcn_update_dpp(-1000000, dnv, 0x114, 0xf, 0xf, false);
                    const int p8 = __builtin_amdgcn_update_dpp(-1000000, dnv, 0x118, 0xf, 0xf, false);
                    const int pb = __builtin_amdgcn_update_dpp(-1000000, dnv, 0x142, 0xa, 0xf, false);
                    f1 = p1 == dnv ? 1.f : 0.f; f2 = p2 == dnv ? 1.f : 0.f; f4 = p4 == dnv ? 1.f : 0.f; f8 = p8 == dnv ? 1.f : 0.f;
                    fb = pb == dnv ? 1.f : 0.f;
                    fc = (n == 0 && dnv == prev_last) ? 1.f : 0.f;
                }
                const bool lastblk = rb == 3;
                const bool is_last = valid && (nxv != dnv || (lastblk && n == cnt - 1));
                __amdgpu_buffer_rsrc_t srd_o;
                if (lines) {
                    const int p0 = __builtin_amdgcn_readfirstlane(sb[rbg].x), cu = __builtin_amdgcn_readfirstlane(cnt);
                    srd_o = __builtin_amdgcn_make_buffer_rsrc(A.e_out + (size_t)(p0 < 0 ? 0 : p0) * H + 32 * jb, 0, (unsigned)cu * (unsigned)H * 4u, 0x00020000);
                    if (A.residual) {
                        floatx4 x[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(srd_o, v_line + (unsigned)(8 * j * H) * 4u, 0, 0));
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<floatx4*>(t_row + 8 * j * HM_TURN_LD) = x[j];
                    }
                }
                // stored exactly once: to its agg row, or -- the piece of a segment that began in an earlier group -- to this
                // group's row of the side buffer
                float* arow = (dnv == head ? A.side + (size_t)grp * H : A.agg + (size_t)(dnv < 0 ? 0 : dnv) * H) + 32 * jb + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y[4];
                    const floatx4 gm4 = *reinterpret_cast<const floatx4*>(gamp + 8 * g), bt4 = *reinterpret_cast<const floatx4*>(betp + 8 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) y[tt] = fmaf(fmaf(acc[rb][4 * g + tt], k, m), gm4[tt], bt4[tt]);
                    if (lines) {
                        floatx4 eo = floatx4{y[0], y[1], y[2], y[3]};
                        if (A.residual) {
                            const floatx4 e0 = *reinterpret_cast<const floatx4*>(t_acc + 8 * g);
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) eo[tt] += e0[tt];
                        }
                        *reinterpret_cast<floatx4*>(t_acc + 8 * g) = eo;
                    } else if (valid && !A.discard_e_out) {   // (the last step of a forward: nobody reads its e + e')
                        floatx4 eo = floatx4{y[0], y[1], y[2], y[3]};
                        if (A.residual) {
                            const floatx4 e0 = *reinterpret_cast<const floatx4*>(outp + 8 * g);
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) eo[tt] += e0[tt];
                        }
                        *reinterpret_cast<floatx4*>(outp + 8 * g) = eo;
                    }
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) y[tt] = fmaf(carry[4 * g + tt], fc, y[tt]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f1, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f2, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f4, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], f8, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) DPP_FMAC_NOP(y[tt], fb, "row_bcast:15 row_mask:0xa bank_mask:0xf");
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        carry[4 * g + tt] = __uint_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) | 31) * 4, __float_as_uint(y[tt])));
                    if (is_last) *reinterpret_cast<floatx4*>(arow + 8 * g) = floatx4{y[0], y[1], y[2], y[3]};
                }
                if (lines) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const floatx4 o = *reinterpret_cast<const floatx4*>(t_row + 8 * j * HM_TURN_LD);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, o), srd_o, v_line + (unsigned)(8 * j * H) * 4u, 0, 0);
                    }
                }
                prev_last = cnt == BE ? __builtin_amdgcn_readlane(dnv, 31) : -3;
            }
        }
        HM_STAMP_E(iter, 6);   // epilogue: LayerNorm, residual, e_out stores, scatter-add
    }
    report_range(rng, A.flags);
}

// ------------------------------------------------------------------------------------------
// node kernel
// ------------------------------------------------------------------------------------------
template <int H, int MODE, int RBW>
__global__ void __launch_bounds__(HM_THREADS, 1) hm_node_kernel(HmNodeArgs A) {
    using C = Cfg<H, RBW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uintx4* img = reinterpret_cast<uintx4*>(smem);
    const half8* imgh = reinterpret_cast<const half8*>(smem);
    constexpr size_t IMGB = hm_node_img_bytes(RBW), STB = HM_ST_BYTES * RBW / 4;
    float* ST = reinterpret_cast<float*>(smem + IMGB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jb = wave % C::NJB, rg = wave / C::NJB;
    const int n = lane & 31, hi = lane >> 5;
    const int N = A.n_nodes;
    if (N <= 0) return;
    // Every workgroup takes a contiguous range of 32-row blocks and walks it in tiles of up to NRB blocks; the blocks of a tile
    // are dealt to the row groups in turn, so a short last tile costs what its rows cost (at N = 100k, hidden 128: 12.2 blocks
    // per workgroup = one full tile + a half one, instead of two rounds of full tiles).
    const int NBLK = (N + 31) / 32;
    const int bq0 = (int)((long long)NBLK * blockIdx.x / gridDim.x), bq1 = (int)((long long)NBLK * (blockIdx.x + 1) / gridDim.x);
    constexpr int K0 = MODE == 0 ? 32 : 2 * H;
    const size_t lin0 = hm_lin_floats(H, K0), linh = hm_lin_floats(H, H);

    float* GB = reinterpret_cast<float*>(smem + IMGB + STB + HM_BLK_BYTES);   // gamma[H] | beta[H]
    float* RS = reinterpret_cast<float*>(smem + IMGB + STB + HM_BLK_BYTES + HM_GB_BYTES);   // encoder: row scales
    int rng = 0;
    // decoder tail: a range violation anywhere earlier in this forward (its kernels have finished: stream order) makes the
    // prediction NaN instead of a plausible wrong number -- a rollout then stops at its next graph build (non-finite position)
    const bool poisoned = A.tail == 2 && A.flags && (*A.flags & ERRF_SPLIT_RANGE) != 0;
    if (MODE != 2) {
        for (int i = tid; i < H; i += HM_THREADS) { GB[i] = A.ln_g[i]; GB[H + i] = A.ln_b[i]; }
    }
    const float* gamp = GB + 32 * jb + 4 * hi;
    const float* betp = GB + H + 32 * jb + 4 * hi;

    int tile_i = 0;
#pragma unroll 1
    for (int bt = bq0; bt < bq1; bt += C::NRB, ++tile_i) {
        HM_STAMP(tile_i, 0);
        const int nbt = min(C::NRB, bq1 - bt);                      // blocks of this tile
        const int nrb = (nbt - rg + C::NRG - 1) / C::NRG;           // ... of which this wave's row group holds (image slots rb < nrb)
        // first row of image slot rbg = RBW rg' + rb': tile block j = rb' NRG + rg'
        auto slot_row0 = [&](int rbg) { return 32 * (bt + (rbg % RBW) * C::NRG + rbg / RBW); };
        // image slots of a short tile that hold no block (tile block j >= nbt) are filled from the tile's FIRST block instead of from
        // the rows that follow the workgroup's range: same instructions (no branch around a load), but cache hits instead of HBM
        // rows nobody uses -- the last, short tile of a workgroup then costs what its blocks cost
        auto row_of = [&](int rbg, int nn) -> long long {
            const int j = (rbg % RBW) * C::NRG + rbg / RBW;
            const int r = (HM_SHORT_TILE_ROWS && j >= nbt ? 32 * bt : slot_row0(rbg)) + nn;
            return r < N ? r : N - 1;
        };
        __syncthreads();   // the previous tile's readers of the image are done
        // head partials of this tile's agg rows (MODE 1): the stitch[] look-up is requested now and resolved under the first GEMM
        int* SC = reinterpret_cast<int*>(RS);   // [NRB * 32] first group holding a head partial of the row, or -1 (RS is the encoders')
        const bool stitched = MODE == 1 && A.stitch != nullptr;
        int sc_row = 0, sc_c = -1;
        if (stitched && tid < C::NRB * 32) {
            sc_row = (int)row_of(tid >> 5, tid & 31);
            sc_c = A.stitch[sc_row];
        }
        floatx16 acc[RBW];
        Lin L;
        float rs[RBW], pois[RBW];   // pois: MODE 0 only (mark_rows)
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) { rs[rb] = 1.f; pois[rb] = 0.f; }
        if (MODE == 0) narrow_rows_to_image<H, RBW, 2>(A.x_in, A.k1, img, RS, A.w[3], tid, row_of);
        else rows_to_image<H, RBW, 2 * RBW>(A.x_in, img, wave, lane, row_of, [](long long, int, floatx4&, floatx4&) {});
        __syncthreads();
        HM_STAMP(tile_i, 1);   // h rows in the image
        if (MODE != 2) {
            const float* wp = A.w;
            L = lin_at(wp, H);
            if (MODE == 0) {
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) rs[rb] = RS[(RBW * rg + rb) * 32 + n];
                init_bias_rows(acc, L.bias, jb, hi, rs);
            } else {
                init_bias(acc, L.bias, jb, hi);
            }
            if (MODE == 0) {
                gemm(acc, L.frag + (size_t)jb * 2 * 128 + lane, imgh + (size_t)(RBW * rg) * 2 * 128 + lane, 2, 2, nrb);
                check_rows(rng, acc);
                mark_rows(pois, acc);
            } else {
                int sc_head = -2;
                const int ng = stitched ? A.tab->n_groups : 0;
                int sc_head1 = -2;
                if (stitched && tid < C::NRB * 32 && sc_c >= 0 && sc_c < ng) {   // both land under the GEMM
                    sc_head = A.head[sc_c];
                    if (sc_c + 1 < ng) sc_head1 = A.head[sc_c + 1];
                }
                gemm(acc, L.frag + (size_t)jb * 2 * C::KS * 128 + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
                if (stitched && tid < C::NRB * 32) SC[tid] = sc_head == sc_row ? 2 * sc_c + (sc_head1 == sc_row ? 1 : 0) : -1;
                __syncthreads();
                HM_STAMP(tile_i, 2);   // GEMM 1a (h part)
                // agg row + the head partials other groups hold of its segment, in group order (hedge.h)
                agg_rows_to_image<H, RBW>(A.agg, img, wave, lane, row_of, stitched ? SC : nullptr, A.side, A.head, ng);
                __syncthreads();
                HM_STAMP(tile_i, 3);   // agg rows (+ head partials) in the image
                gemm(acc, L.frag + ((size_t)jb * 2 * C::KS + C::KS) * 128 + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
            }
            wp += lin0;
            HM_STAMP(tile_i, 4);   // GEMM 1b (agg part)
#pragma unroll 1
            for (int l = 1; l <= A.nl; ++l) {
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc_to_img<true>(acc[rb], img + (size_t)(RBW * rg + rb) * C::KS * 128, jb, lane);
                __syncthreads();
                L = lin_at(wp, H);
                if (MODE == 0) init_bias_rows(acc, L.bias, jb, hi, rs);
                else init_bias(acc, L.bias, jb, hi);
                gemm(acc, L.frag + (size_t)jb * C::KS * 128 + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
                if (MODE == 0) mark_rows(pois, acc);
                wp += linh;
            }
            HM_STAMP(tile_i, 5);   // hidden Linears
            ln_publish<H, RBW>(acc, ST, rg, jb, n, hi);
            __syncthreads();
            HM_STAMP(tile_i, 6);   // LayerNorm statistics published
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (rb >= nrb) continue;
                const int rbg = RBW * rg + rb;
                float k, m;
                ln_merge<H>(ST, rbg, n, L.u * rs[rb], A.eps, A.h_valid, k, m);
                if (MODE == 0) m += pois[rb];
                const int r = slot_row0(rbg) + n;
                const bool valid = rb < nrb && r < N;
                const size_t off = (size_t)(r < N ? r : N - 1) * H + 32 * jb + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 y;
                    const floatx4 gm4 = *reinterpret_cast<const floatx4*>(gamp + 8 * g), bt4 = *reinterpret_cast<const floatx4*>(betp + 8 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) y[tt] = fmaf(fmaf(acc[rb][4 * g + tt], k, m), gm4[tt], bt4[tt]);
                    if (A.residual) {
                        // (requesting these rows ahead of the previous row block's stores, or all of them before the epilogue, was
                        // measured: -4 % / -17 %: the 16 .. 64 registers they hold spill elsewhere in this 256-VGPR kernel)
                        const floatx4 h0 = *reinterpret_cast<const floatx4*>(A.x_in + off + 8 * g);
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) y[tt] += h0[tt];
                    }
                    if (valid) *reinterpret_cast<floatx4*>(A.h_out + off + 8 * g) = y;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[rb][4 * g + tt] = y[tt];
                }
            }
            HM_STAMP(tile_i, 7);   // epilogue: residual read, h stores
            if (A.tail == 0) continue;
            // the new h becomes the tail's input image (every wave has passed the barrier after the last Linear)
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc_to_img<false>(acc[rb], img + (size_t)(RBW * rg + rb) * C::KS * 128, jb, lane);
            __syncthreads();
            HM_STAMP(tile_i, 8);   // h in the image
        }
        if (A.tail == 1) {   // (MODE 2 launches carry tail 1 = projection only, or tail 2 = decoder only)
            const Lin LP = lin_at(A.w_tail, 2 * H);
            // P leaves at the scale its consumer multiplies at: the systolic edge kernel adds P_i + P_j straight into accumulators that
            // carry its first Linear's power-of-two weight scale (exact: a power of two commutes with every rounding on the way)
            const float p_out_scale = LP.inv_u * (A.p_scale ? *A.p_scale : 1.f);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const int jbv = jb + half * C::NJB;
                init_bias(acc, LP.bias, jbv, hi);
                gemm(acc, LP.frag + (size_t)jbv * C::KS * 128 + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) {
                    const int r = slot_row0(RBW * rg + rb) + n;
                    if (rb < nrb && r < N) {
                        float* pp = A.P_out + (size_t)r * 2 * H + 32 * jbv + 4 * hi;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            floatx4 v;
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) v[tt] = acc[rb][4 * g + tt] * p_out_scale;
                            *reinterpret_cast<floatx4*>(pp + 8 * g) = v;
                        }
                    }
                }
                HM_STAMP(tile_i, 9 + half);   // projection half: GEMM + P stores
            }
        } else if (A.tail == 2) {
            const float* wp = A.w_tail;
#pragma unroll 1
            for (int l = 0; l < A.nl; ++l) {
                const Lin LD = lin_at(wp, H);
                init_bias(acc, LD.bias, jb, hi);
                gemm(acc, LD.frag + (size_t)jb * C::KS * 128 + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) if (rb < nrb) acc_to_img<true>(acc[rb], img + (size_t)(RBW * rg + rb) * C::KS * 128, jb, lane);
                __syncthreads();
                wp += linh;
            }
            if (jb == 0) {
                const Lin LO = lin_at(wp, 32);
                init_bias(acc, LO.bias, 0, hi);
                gemm(acc, LO.frag + lane, imgh + (size_t)(RBW * rg) * C::KS * 128 + lane, C::KS, C::KS, nrb);
                check_rows(rng, acc);
                if (hi == 0) {
#pragma unroll
                    for (int rb = 0; rb < RBW; ++rb) {
                        const int r = slot_row0(RBW * rg + rb) + n;
                        if (rb < nrb && r < N) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (c < A.out_dim) A.dec_out[(size_t)r * A.out_dim + c] = (poisoned || rng) ? __builtin_nanf("") : acc[rb][c] * LO.inv_u;
                        }
                    }
                }
            }
        }
    }
    report_range(rng, A.flags);
}

// ------------------------------------------------------------------------------------------
// weight images
// ------------------------------------------------------------------------------------------
struct PackHmJobs {
    int n, first;   // first: index of job[0] in the whole list (stats rows)
    PackHmJob job[kPackHmMax];
};

constexpr int HM_PACK_THREADS = 1024;   // one workgroup per Linear: the packing kernels run on few CUs, so each uses a full one
__device__ __forceinline__ float hm_value(const PackHmJob& j, int o, int k) {
    const int os = o / j.out_seg, ro = o % j.out_seg, is = k / j.k_seg, rk = k % j.k_seg;
    if (ro >= j.out_valid || rk >= j.k_valid) return 0.f;
    return j.W[(size_t)ro * j.ld + j.col0[os + is] + rk];
}

// phase 1: statistics of every Linear -> stats[4 job ..]: gain ||W||_F / sqrt(out) / sqrt(2) over the columns that make its
// pre-activation, rms and maximum of the bias, maximum |W| of the packed block.  Also resets the job's row-scale cap slot.
__global__ void __launch_bounds__(HM_PACK_THREADS) pack_hm_stats_kernel(PackHmJobs J, float* __restrict__ stats) {
    const PackHmJob& j = J.job[blockIdx.x];
    __shared__ float red[HM_PACK_THREADS];
    const int tid = threadIdx.x;
    float ss = 0.f, wm = 0.f;
    if (j.gain_cols > 0) {
        for (int i = tid; i < j.out_valid * j.gain_cols; i += HM_PACK_THREADS) {
            const float v = j.W[(size_t)(i / j.gain_cols) * j.ld + j.gain_col0 + (i % j.gain_cols)];
            ss = fmaf(v, v, ss);
        }
    }
    for (int i = tid; i < j.out_pad * j.k_pad; i += HM_PACK_THREADS) {
        const float v = hm_value(j, i / j.k_pad, i % j.k_pad);
        if (j.gain_cols <= 0) ss = fmaf(v, v, ss);
        wm = fmaxf(wm, fabsf(v));
    }
    auto reduce = [&](float v, bool is_max) {
        red[tid] = v;
        __syncthreads();
        for (int s = HM_PACK_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    ss = reduce(ss, false);
    wm = reduce(wm, true);
    if (tid == 0) {
        float bs = 0.f, bm = 0.f;
        if (j.bias) for (int o = 0; o < j.bias_n; ++o) { bs = fmaf(j.bias[o], j.bias[o], bs); bm = fmaxf(bm, fabsf(j.bias[o])); }
        const int outs = j.out_valid * (j.out_pad / j.out_seg);
        float* st = stats + 4 * (J.first + blockIdx.x);
        st[0] = sqrtf(ss / (float)(outs > 0 ? outs : 1) * 0.5f);
        st[1] = j.bias_n > 0 ? sqrtf(bs / (float)j.bias_n) : 0.f;
        st[2] = bm;
        st[3] = wm;
        j.dst[3] = 1.0e18f;   // row-scale cap (chain heads on raw features): phase 2 takes the minimum over the chain
    }
}

__device__ __forceinline__ float hm_pow2_near(float want) {
    if (!(want > 0.f) || !(want < 3.0e38f)) return 1.f;
    int ex;
    const float f = frexpf(want, &ex);          // want = f 2^ex, f in [0.5, 1): nearest power of two
    int sh = f >= 0.70710678f ? ex : ex - 1;
    sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
    return ldexpf(1.f, sh);
}
__device__ __forceinline__ float hm_pow2_floor(float want) {
    if (!(want > 0.f) || !(want < 3.0e38f)) return 1.f;
    int ex;
    (void)frexpf(want, &ex);
    int sh = ex - 1;
    sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
    return ldexpf(1.f, sh);
}

// phase 2: walk the chain from its head to this Linear (scales of every predecessor from the phase-1 statistics), then pack.
// m = estimated rms of the activations in the chain's units: a ReLU layer maps the second moment  m^2 -> gain^2 m^2 + b_rms^2 / 2
// (zero-mean weights; the bias term is what deep chains settle on).  U_l = power of two nearest kHmTargetRms / m_l.
__global__ void __launch_bounds__(HM_PACK_THREADS) pack_hm_kernel(PackHmJobs J, const PackHmJob* __restrict__ all, const float* __restrict__ stats) {
    const PackHmJob& j = J.job[blockIdx.x];
    __shared__ float sc[2];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int chain[24], nc = 0;
        for (int q = J.first + blockIdx.x; q >= 0 && nc < 24; q = all[q].pred) chain[nc++] = q;
        const int head = chain[nc - 1];
        const bool raw = all[head].in_rms != 1.f;   // rows scaled by their own power of two (encoders): biases are the cap's business
        float m = all[head].in_rms, U = 1.f, t = 1.f;
        if (!(m > 0.f)) m = 1.f;
        for (int c = nc - 1; c >= 0; --c) {
            const float* st = stats + 4 * chain[c];
            float m2 = st[0] * st[0] * m * m;
            if (!raw) m2 = fmaf(0.5f * st[1], st[1], m2);
            m = sqrtf(m2);
            if (!(m > 1.0e-30f) || !(m < 1.0e30f)) m = 1.f;
            float Un = hm_pow2_near(kHmTargetRms / m);
            t = Un / U;
            // keep the packed weights inside the window where both halves of the split are normal: max |W t| in [2^-4, 2^13]
            const float wm = st[3];
            if (wm > 0.f && wm < 3.0e38f) {
                if (wm * t > 8192.0f) t = hm_pow2_floor(8192.0f / wm);
                if (wm * t < 0.0625f) t = 2.f * hm_pow2_floor(0.0625f / wm);
            }
            U = U * t;
            if (!(U > 7.9e-31f)) U = 7.9e-31f;   // float range whatever the depth
            if (!(U < 1.2e30f)) U = 1.2e30f;
            if (raw && st[2] > 0.f) {
                // no row of this chain may be scaled so far that this Linear's bias leaves the range (accumulator units: U rs b)
                const float cap = hm_pow2_floor(4096.0f / (U * st[2]));
                atomicMin(reinterpret_cast<unsigned*>(all[head].dst + 3), __float_as_uint(cap));   // positive floats order like their bits
            }
        }
        sc[0] = t;
        sc[1] = U;
    }
    // centred form (this Linear feeds a LayerNorm): column means of W over the valid outputs and the mean of the bias
    __shared__ float cmean[513];
    for (int k = tid; k < j.k_pad && k < 512; k += HM_PACK_THREADS) {
        float a = 0.f;
        if (j.center) {
            for (int o = 0; o < j.out_valid; ++o) a += hm_value(j, o, k);
            a /= (float)(j.out_valid > 0 ? j.out_valid : 1);
        }
        cmean[k] = a;
    }
    if (tid == 0) {
        float a = 0.f;
        if (j.center && j.bias) {
            for (int o = 0; o < j.bias_n; ++o) a += j.bias[o];
            a /= (float)(j.bias_n > 0 ? j.bias_n : 1);
        }
        cmean[512] = a;
    }
    __syncthreads();
    const float t = sc[0], U = sc[1];
    if (tid == 0) { j.dst[0] = t; j.dst[1] = 1.f / U; j.dst[2] = U; }
    for (int o = tid; o < j.out_pad; o += HM_PACK_THREADS) j.dst[4 + o] = (j.bias && o < j.bias_n) ? (j.bias[o] - cmean[512]) * U : 0.f;
    _Float16* frag = reinterpret_cast<_Float16*>(j.dst + 4 + j.out_pad);
    const int ksn = j.k_pad / 16;
    const int entries = (j.out_pad / 32) * ksn * 64;   // (jb, ks, lane); two parts each
    for (int e = tid; e < entries; e += HM_PACK_THREADS) {
        const int lane = e & 63, ks = (e >> 6) % ksn, jbv = (e >> 6) / ksn;
        const int o = 32 * jbv + (lane & 31), kg = lane >> 5;
        _Float16* hi_p = frag + ((size_t)((jbv * ksn + ks) * 2 + 0) * 64 + lane) * 8;
        _Float16* lo_p = frag + ((size_t)((jbv * ksn + ks) * 2 + 1) * 64 + lane) * 8;
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * ks + 8 * (q >> 2) + 4 * kg + (q & 3);
            const bool valid = (o % j.out_seg) < j.out_valid && (k % j.k_seg) < j.k_valid;   // padding stays exactly zero
            const float v = (hm_value(j, o, k) - (valid ? cmean[k < 512 ? k : 0] : 0.f)) * t;
            const _Float16 h = (_Float16)v;
            hi_p[q] = h;
            lo_p[q] = (_Float16)(v - (float)h);
        }
    }
}

int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}

template <class K>
int set_lds_attr(K kernel, size_t bytes = HM_LDS_BYTES) {
    GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return GM_OK;
}

template <int H>
int launch_edge_h(bool enc, const HmEdgeArgs& a, hipStream_t s) {
    static PerDeviceOnce once;
    const int rc_attr = once.run([]() -> int {
        int rc = set_lds_attr(hm_edge_kernel<H, true>);
        if (rc == GM_OK) rc = set_lds_attr(hm_edge_kernel<H, false>);
        return rc;
    });
    if (rc_attr != GM_OK) return rc_attr;
    ProfScope prof(a.prof, enc ? PROF_ENC : PROF_EDGE, s);
    if (enc) hipLaunchKernelGGL((hm_edge_kernel<H, true>), dim3(device_cus()), dim3(HM_THREADS), HM_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((hm_edge_kernel<H, false>), dim3(device_cus()), dim3(HM_THREADS), HM_LDS_BYTES, s, a);
    return GM_OK;
}

template <int H, int RBW>
int launch_node_hr(int mode, const HmNodeArgs& a, hipStream_t s) {
    constexpr size_t LDS = hm_node_lds_bytes(RBW);
    static PerDeviceOnce once;
    const int rc_attr = once.run([]() -> int {
        int rc = set_lds_attr(hm_node_kernel<H, 0, RBW>, LDS);
        if (rc == GM_OK) rc = set_lds_attr(hm_node_kernel<H, 1, RBW>, LDS);
        if (rc == GM_OK) rc = set_lds_attr(hm_node_kernel<H, 2, RBW>, LDS);
        return rc;
    });
    if (rc_attr != GM_OK) return rc_attr;
    const int tiles = (int)cdiv(a.n_nodes, Cfg<H, RBW>::M);
    int grid = device_cus();
    if (tiles < grid) grid = tiles < 1 ? 1 : tiles;
    ProfScope prof(a.prof, mode == 0 ? PROF_ENC : PROF_NODE, s);
    if (mode == 0) hipLaunchKernelGGL((hm_node_kernel<H, 0, RBW>), dim3(grid), dim3(HM_THREADS), LDS, s, a);
    else if (mode == 1) hipLaunchKernelGGL((hm_node_kernel<H, 1, RBW>), dim3(grid), dim3(HM_THREADS), LDS, s, a);
    else hipLaunchKernelGGL((hm_node_kernel<H, 2, RBW>), dim3(grid), dim3(HM_THREADS), LDS, s, a);
    return GM_OK;
}
// small graphs: one 32-row block per wave, so that the tiles cover the CUs.  HM_NODE_SMALL_ROUNDS: the 4-block form is used
// once its tiles fill the CUs that many times over.  1 is the measured choice: at N = 100k / hidden 128 (391 tiles of 256
// rows on 256 CUs, a half-empty second round) the small form is still 16 % slower (1.80 vs 1.55 ms of node kernels per
// step, A/B on one box) -- it streams the weights four times as often.
// Round 6, hidden 128: the kernel deals 32-row BLOCKS evenly to the workgroups whatever the form, so what counts is blocks per
// workgroup, not whole tiles -- measured on one box (bench.py, -DHM_NODE_SMALL_BLOCKS): 1.2 blocks per workgroup (2 x 5k nodes)
// one-block form 1924 vs four-block form 1503 steps/s; 2.4 (4 x 5k) 2091 vs 2021; 4.9 (a C5 batch, 8 x 5k) 2304 vs 2355 - 2360.
// The four-block form takes over from 4 blocks per workgroup (round 5: from 8).  Results do not depend on the form.
#ifndef HM_NODE_SMALL_ROUNDS
#define HM_NODE_SMALL_ROUNDS 1
#endif
#ifndef HM_NODE_RBW
#define HM_NODE_RBW 4
#endif
template <int H>
int launch_node_h(int mode, const HmNodeArgs& a, hipStream_t s) {
#ifndef HM_NODE_SMALL_BLOCKS
#define HM_NODE_SMALL_BLOCKS (H == 128 ? 4 : HM_NODE_SMALL_ROUNDS * Cfg<H, 4>::NRB)   // 32-row blocks per workgroup below which the one-block form runs
#endif
    if (cdiv(a.n_nodes, 32) < (int64_t)(HM_NODE_SMALL_BLOCKS) * device_cus()) return launch_node_hr<H, 1>(mode, a, s);
    return launch_node_hr<H, HM_NODE_RBW>(mode, a, s);
}

}  // namespace

int pack_hm(const PackHmJob* jobs, int n, PackHmJob* jobs_dev, float* stats, hipStream_t s) {
    if (n <= 0) return GM_OK;
    GM_REQUIRE(stats && jobs_dev, GM_ERR_INVALID_ARGUMENT, "pack_hm: no scratch");
    // the whole job list on the device: phase 2 walks chains across launch batches
    GM_HIP_CHECK(hipMemcpyAsync(jobs_dev, jobs, (size_t)n * sizeof(PackHmJob), hipMemcpyHostToDevice, s));
    for (int phase = 0; phase < 2; ++phase) {
        for (int off = 0; off < n; off += kPackHmMax) {
            PackHmJobs J{};
            J.n = n - off < kPackHmMax ? n - off : kPackHmMax;
            J.first = off;
            for (int i = 0; i < J.n; ++i) J.job[i] = jobs[off + i];
            if (phase == 0) hipLaunchKernelGGL(pack_hm_stats_kernel, dim3(J.n), dim3(HM_PACK_THREADS), 0, s, J, stats);
            else hipLaunchKernelGGL(pack_hm_kernel, dim3(J.n), dim3(HM_PACK_THREADS), 0, s, J, jobs_dev, stats);
            GM_LAUNCH_CHECK();
        }
    }
    return GM_OK;
}

#ifdef HM_STAMPS
extern "C" int gm_debug_hm_stamps(unsigned long long* out) {   // 4 workgroups x 4 tiles x 16 slots, development builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hm_stamps), sizeof(unsigned long long) * 4 * 4 * 16) == hipSuccess ? 0 : -1;
}
#endif

bool hm_supported(int H) { return H == 64 || H == 128 || H == 256; }   // instantiated widths (hm_padded_hidden maps a model onto one)

int launch_edge_hm(int H, bool enc, const HmEdgeArgs& a, hipStream_t s) {
    GM_REQUIRE(a.w && a.e_in && a.e_out && a.ln_g && a.ln_b, GM_ERR_INVALID_ARGUMENT, "launch_edge_hm: null pointer");
    GM_REQUIRE(enc || (a.P && a.dst && a.src && a.blk && a.tab && a.head && (a.side || !a.agg)), GM_ERR_INVALID_ARGUMENT,
               "launch_edge_hm: processor step needs P, dst, src, the block tables and the side buffer");
    GM_REQUIRE(!enc || (a.k1 >= 1 && a.k1 <= 16), GM_ERR_UNSUPPORTED, "launch_edge_hm: edge_dim %d unsupported (1..16)", a.k1);
    GM_REQUIRE(a.nl >= 2, GM_ERR_INVALID_ARGUMENT, "launch_edge_hm: num_layers %d", a.nl);
    if (!a.hdr && a.n_edges_host <= 0) return GM_OK;
    int rc;
    switch (H) {
    case 64: rc = launch_edge_h<64>(enc, a, s); break;
    case 128: rc = launch_edge_h<128>(enc, a, s); break;
    case 256: rc = launch_edge_h<256>(enc, a, s); break;
    default: GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "launch_edge_hm: hidden_size %d (64, 128, 256)", H);
    }
    if (rc != GM_OK) return rc;
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_node_hm(int H, int mode, const HmNodeArgs& a, hipStream_t s) {
    GM_REQUIRE(a.x_in, GM_ERR_INVALID_ARGUMENT, "launch_node_hm: null input");
    GM_REQUIRE(mode == 2 || (a.w && a.h_out && a.ln_g && a.ln_b), GM_ERR_INVALID_ARGUMENT, "launch_node_hm: null pointer");
    GM_REQUIRE(mode != 1 || a.agg, GM_ERR_INVALID_ARGUMENT, "launch_node_hm: processor step needs agg");
    GM_REQUIRE(mode != 0 || (a.k1 >= 1 && a.k1 <= 32), GM_ERR_UNSUPPORTED, "launch_node_hm: node_dim %d unsupported (1..32)", a.k1);
    GM_REQUIRE((a.tail == 0 && mode != 2) || a.w_tail, GM_ERR_INVALID_ARGUMENT, "launch_node_hm: tail without weights");
    GM_REQUIRE(a.tail != 1 || a.P_out, GM_ERR_INVALID_ARGUMENT, "launch_node_hm: projection without P_out");
    GM_REQUIRE(mode != 2 || a.tail == 1 || a.tail == 2, GM_ERR_INVALID_ARGUMENT, "launch_node_hm: mode 2 runs a tail only (1 projection, 2 decoder)");
    GM_REQUIRE(a.tail != 2 || (a.dec_out && a.out_dim >= 1 && a.out_dim <= 4), GM_ERR_INVALID_ARGUMENT, "launch_node_hm: decoder tail arguments");
    if (a.n_nodes <= 0) return GM_OK;
    int rc;
    switch (H) {
    case 64: rc = launch_node_h<64>(mode, a, s); break;
    case 128: rc = launch_node_h<128>(mode, a, s); break;
    case 256: rc = launch_node_h<256>(mode, a, s); break;
    default: GM_REQUIRE(false, GM_ERR_UNSUPPORTED, "launch_node_hm: hidden_size %d (64, 128, 256)", H);
    }
    if (rc != GM_OK) return rc;
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm
