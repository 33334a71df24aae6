// Processor edge kernel, systolic form, for hidden 128 / three Linears (num_layers = 2) on the fp16 matrix pipe with
// fp32 accuracy:  e' = LayerNorm(phi_e([h_i, h_j, e])),  e_out = e + e',  agg_i = sum_{e -> i} e'
// (epd_gnn.py:35-46,100-105; block semantics DESIGN.md section 2).
//
// Arithmetic.  Every fp32 operand is split into two fp16 parts, x = hi + lo (22 significant bits; the low parts may be
// fp16 subnormals, which v_mfma_f32_32x32x16_f16 honours -- checked on gfx950).  A product of two fp16 values is exact in
// fp32, so  lo*hi + hi*lo + hi*hi  accumulated in fp32 reproduces the fp32 product to 2^-22; through the whole model
// this scheme is as accurate against float64 as plain float32 (tools/f16_split_study.py: 1.1e-6 vs 0.9e-6).  Weights are
// pre-scaled by a power of two per Linear (chosen from its gain, hmlp.h: operand images at rms ~2^4, both halves of every
// weight normal); the scales ride on through the activations, biases and P and leave in the LayerNorm statistics: no
// run-time multiplies.  A value that does not fit fp16 is detected (NaN accumulator row) and reported, never clamped.
//
// Structure.  12 waves per workgroup, one workgroup per CU, three waves per SIMD.  Wave (role, jb): role = Linear 1 / 2 /
// 3, jb = 32-feature output block; the wave keeps only ITS Linear's 32 weight rows in registers (64 VGPRs) for the whole
// launch.  Work advances in ticks of one 32-edge block with ONE workgroup barrier per tick; every LDS buffer is
// double-buffered by block parity:
//   role 0: all the tick's requests at its top (P rows and indices of block x+1, e rows of block x+2: unconditional,
//           clamped); P_i[dst] + P_j[src] of block x -> accumulators; e of block x+1 -> operand image E (between the
//           MFMAs); Linear 1 on E, ReLU, image X1; destinations of the block into a small LDS ring for role 2;
//   role 1: Linear 2 of block x-1 (X1 -> X2); LayerNorm + e_out = e + e' of the first EPI_SPLIT row groups of block x-3
//           (row-major, 8 lanes per row; unconditional stores: rows that do not exist lie beyond the store's buffer bound);
//   role 2: Linear 3 of block x-2 (X2 -> Z + LayerNorm partial statistics); the other row groups of that epilogue;
//           aggregation of block x-3 on a TRANSPOSED view (lane = feature, 16 rows of a half block in registers): the
//           segment structure of the destination-sorted rows is the same for every feature, so it lives in scalar
//           registers (continuation / last-row bit masks from the block tables) and the segmented scan is 15 masked adds
//           down the registers; one 128-byte store per finished segment and half wave -- to its agg row, or, for the piece
//           of a segment that began in an earlier group of 4 blocks, to that group's row of the side buffer, which the
//           node kernel adds in group order: no atomics.  Roles run at different s_setprio levels (role 2 first).
// The cache policy of the row stores is chosen per launch by size (STREAM, below).
// hipcc's counted vmcnt waits assume the path with the fewest younger operations and share one in-order counter between
// loads and stores: every global access of the tick loops is therefore branch-free, the weights are waited for before the
// loops, and the block tables come through scalar loads (separate __restrict__ kernel parameters).
// A wave's MFMAs form one dependent chain, so its other work of the tick is placed BETWEEN them with the order pinned
// (one MFMA shadows about three vector instructions of the same wave).  Global rows move as whole 128-byte lines
// (8 lanes per row); the register <-> MFMA-fragment re-layouts go through XOR-swizzled, conflict-free LDS images.
//
// Blocks are aligned to each graph's first edge and grouped in fours (the aggregation's carry resets there), both
// listed by build_edge_blocks(): results do not depend on how many graphs share a launch, nor on the run.
//
// Built with -fno-slp-vectorize: packed fp32 VALU beside MFMAs is slow (guide) and hipcc's v_pk_fma_f32 form of the
// LayerNorm epilogue returned wrong low lanes on gfx950 in this kernel.
#include <type_traits>
#include "common.h"
#include "mlp.h"
#include "hedge.h"
#include "hmlp.h"
#include "hmma_dev.h"
#include "blocks_dev.h"

namespace gm {

namespace {

constexpr int H = 128;
constexpr int BE = 32;             // edges per block
constexpr int SYS_THREADS = 768;
static_assert(BE == kBlockEdges, "block tables and the systolic kernels agree on the block size");
#ifndef SIDE_STRIDE
#define SIDE_STRIDE 6   // MFMA slots per row group of role 1's e_out epilogue (24 slots per tick, 4 row groups)
#endif
#ifdef HEDGE_STAMPS
// development build only: s_memtime stamps of the three roles (workgroup 0, waves jb = 0, lane 0) at the phase boundaries of
// ticks 16..47; read back with gm_debug_sys_stamps (tools/sys_stamps.py)
__device__ unsigned long long g_sys_stamps[3 * 32 * 8 + 32];   // + the 100 MHz s_memrealtime at the start of role 0's ticks
#define SYS_STAMP(tick, slot)                                                                                         \
    do {                                                                                                              \
        if (blockIdx.x == 0 && jb == 0 && lane0 == 0 && (tick) >= 16 && (tick) < 48) {                                \
            g_sys_stamps[(role * 32 + (tick) - 16) * 8 + (slot)] = __builtin_readcyclecounter();                     \
            if (role == 0 && (slot) == 0) g_sys_stamps[3 * 32 * 8 + (tick) - 16] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                             \
    } while (0)
#else
#define SYS_STAMP(tick, slot) do { } while (0)
#endif
#ifndef HEDGE_PRIO0
#define HEDGE_PRIO0 1
#define HEDGE_PRIO1 1
#define HEDGE_PRIO2 3
#endif
#ifndef EPI_SPLIT
#define EPI_SPLIT 1   // row groups (of 8 rows) of a block's LayerNorm + e_out epilogue that role 1 keeps; role 2 takes the others
#endif
#ifndef HEDGE_XCD
#define HEDGE_XCD 1
#endif
// Cache policy of a launch's row stores (aux bits of the buffer instructions: 2 = nt, 16 = sc1).  STREAM launches write e + e' with
// sc1 | nt -- written through and kept out of L2 / the Infinity Cache, which then hold h, P and agg for the kernels that follow -- and
// agg with sc1: right when the rows a launch writes cannot stay resident until the next launch reads them (target: 1 GB per launch,
// + 3 % on the step).  Small graphs keep the default policy: at N = 5k the 49 MB of e live in the 256 MB Infinity Cache from launch to
// launch, and writing them through costs every launch an HBM round trip (round 5 shipped the streaming policy at every size: C2 - 8 %).
// The launcher chooses (launch_edge_sys: kStreamStoreBytes).
constexpr int ST_STREAM_E = 2 | 16, ST_STREAM_AGG = 16;
#ifndef HENC_ST_AUX
#define HENC_ST_AUX 0    // cache policy of the edge encoder's e stores (2 = nt, 16 = sc1)
#endif
constexpr int HW_HEADER_FLOATS = 4;            // T1, 1/T3, cap of the per-row input scale (encoder image), pad
constexpr int HW_VEC_FLOATS = 5 * H;           // b2*T2 | b3*T3 | gamma | beta | b1*T1 (the encoder's; a processor step has b1 in P)
constexpr int HW_IMAGE_HALF8 = 3 * 4 * 8 * 2 * 64;   // [layer][jb][ks][part][lane]


// ---- LDS map (bytes).  The 32 x 32-float tiles (PS, Z) have a row stride of 144 bytes: conflict-free for the row-major
// accesses (8 lanes per row) and for the accumulator-layout ones (lane = row, 16-byte quads), and every access of the tick
// loops is "lane-constant base + immediate".
constexpr int TILE_ROW_B = 144;
constexpr int TILE_B = 32 * TILE_ROW_B;
constexpr int IMG_B = 16384;                      // one operand image: [8 ks][2 parts][64 slots] x 16 B
constexpr int L_E = 0;                            // [2] images of e rows (eslot() order)
constexpr int L_X1 = L_E + 2 * IMG_B;
constexpr int L_X2 = L_X1 + 2 * IMG_B;
constexpr int L_Z = L_X2 + 2 * IMG_B;             // [2][4 jb] tiles: Linear-3 accumulators
constexpr int L_PS = L_Z + 8 * TILE_B;            // [4 jb] tiles: role-0 staging of P_i + P_j
constexpr int L_ST = L_PS + 4 * TILE_B;           // [2][4 jb][32 rows] floats: sum of squares of a row's 32 features of wave jb
constexpr int L_KM = L_ST + 2 * 32 * 4 * 4;       // [4 jb][32] floats: role 1's 1 / (T sigma) per row
constexpr int L_KM2 = L_KM + 4 * 32 * 4;          // [4 jb][32]: role 2's
constexpr int L_ZERO_END = L_KM2 + 4 * 32 * 4;    // everything below starts zeroed
constexpr int L_VEC = L_ZERO_END;                 // 4 x 128 floats: b2 T2 | b3' T3 | gamma | beta
constexpr int DR_SLOTS = 8;                       // ring of per-block destination ids handed from role 0 to role 2
constexpr int L_DR = L_VEC + 4 * H * 4;           // [DR_SLOTS][32] ints
constexpr size_t SYS_LDS_BYTES = L_DR + DR_SLOTS * BE * 4;
static_assert(SYS_LDS_BYTES <= 160 * 1024, "LDS budget");

// slot of lane (n, kg) inside a fragment of the row-written image E (conflict-free for the 8-byte row-major writes)
__device__ __forceinline__ int eslot(int n, int kg, int ksbit) { return (n ^ (2 * (ksbit + 2 * kg))) + 32 * kg; }

// ---- buffer addressing.  Every global access of the tick loops is a buffer instruction: a scalar resource (base, byte
// count), a lane-constant VGPR offset (range-checked against the byte count: a store beyond it is dropped, a load returns 0), a
// scalar offset that moves with the block and an immediate -- no per-lane 64-bit address arithmetic in the loops.
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000); }
__device__ __forceinline__ floatx4 bld4(srd_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
// the same with the non-temporal hint (aux bit 1): rows that are read once per launch
template <int NT>   // NT: cache-policy bits of the instruction (2 = nt, 16 = sc1: served by L2, not kept in the CU's L1)
__device__ __forceinline__ floatx4 bld4s(srd_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NT)); }
template <int NT>
__device__ __forceinline__ void bst4s(srd_t r, unsigned voff, unsigned soff, floatx4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), r, voff, soff, NT); }
__device__ __forceinline__ intx4 bldi4(srd_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(intx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ void bst4(srd_t r, unsigned voff, unsigned soff, floatx4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), r, voff, soff, 0); }
template <int AUX>
__device__ __forceinline__ void bst1s(srd_t r, unsigned voff, unsigned soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX); }
// scalar clamp to [0, hi]: written as SALU so that the values that feed a resource stay in scalar registers
// Workgroups are dealt to the 8 XCDs in turn (workgroup i runs on XCD i % 8), each XCD with its own L2.  The systolic kernels give
// workgroup i the i-th contiguous range of the destination-sorted edge list; numbered this way, the 32 workgroups of an XCD hold one
// contiguous eighth of it, so that -- with particle ids in spatial order (RolloutEngine: grid cells, x fastest) -- the workgroups that
// gather the same neighbouring P rows at the same time (ranges one layer of cells apart advance in step) share an L2.
__device__ __forceinline__ int xcd_major_wg() {
    const int g = (int)gridDim.x, i = (int)blockIdx.x;
    return (HEDGE_XCD && (g & 7) == 0) ? (i & 7) * (g >> 3) + (i >> 3) : i;
}

__device__ __forceinline__ int s_clamp0(int v, int hi) {
    int r;
    asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, %2" : "=s"(r) : "s"(v), "s"(hi) : "scc");
    return r;
}

#define LDS(T, off) (*reinterpret_cast<T*>(smem + (off)))
// A lane-constant LDS / buffer base: made opaque so that hipcc addresses "base register + 16-bit immediate" instead of folding
// every region offset into a register of its own (the DS immediate reaches 64 KiB, the regions lie further apart than that).
__device__ __forceinline__ unsigned opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
#define GM_SB __builtin_amdgcn_sched_barrier(0)

// accumulator registers 8q..8q+7 of a wave's 32-feature block are the elements of B fragment ks = 2 jb + q (same lane):
// K slot (lane >> 5, j) of k-group ks carries feature 16 ks + 8 (j >> 2) + 4 (lane >> 5) + (j & 3) in both operands.
// a: byte address of this lane's slot in fragment (ks = 2 jb, hi part) of the image.
__device__ __forceinline__ void acc_to_image(const floatx16& a, char* smem, unsigned addr) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu(a[8 * q + j]);
        uintx2 h0, l0, h1, l1;
        split4(v[0], v[1], v[2], v[3], h0, l0);
        split4(v[4], v[5], v[6], v[7], h1, l1);
        LDS(uintx4, addr + (q * 2 + 0) * 1024) = uintx4{h0[0], h0[1], h1[0], h1[1]};
        LDS(uintx4, addr + (q * 2 + 1) * 1024) = uintx4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// One Linear for this wave's 32 output features: 8 k-groups x 3 MFMAs (lo*hi, hi*lo, hi*hi).  side(slot), slot = 0..23, runs
// after each MFMA with the instruction order pinned.  B fragments are fetched one k-group (hi part) / two MFMAs (lo part) ahead.  a0 / a1: byte address of this
// lane's slot in fragment (0, hi part) for the even / odd k-groups (they differ in the swizzled image E only); c0: initial
// accumulators (the first MFMA's C operand).
template <class F>
__device__ __forceinline__ void mlp_layer(floatx16& acc, const floatx16& c0, const half8 (&wh)[8], const half8 (&wl)[8], char* smem, unsigned a0, unsigned a1, F&& side) {
    half8 bh = LDS(half8, a0), bl = LDS(half8, a0 + 1024);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        half8 nh = bh;
        const unsigned an = (((ks + 1) & 1) ? a1 : a0) + (ks + 1) * 2048;
        GM_SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, ks == 0 ? c0 : acc, 0, 0, 0);
        GM_SB;
        if (ks + 1 < 8) nh = LDS(half8, an);
        side(3 * ks);
        GM_SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, acc, 0, 0, 0);
        GM_SB;
        if (ks + 1 < 8) bl = LDS(half8, an + 1024);   // the low part is read by the middle MFMA only: its registers are free again
        side(3 * ks + 1);
        GM_SB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, acc, 0, 0, 0);
        GM_SB;
        side(3 * ks + 2);
        bh = nh;
    }
    GM_SB;
}

// 1 / (T sigma) of a row from the four 32-feature sums of squares of its SCALED, CENTRED Linear-3 outputs (the image holds
// W3 and b3 with their mean over the output features removed, so the outputs have zero mean: x_hat = acc * k)
__device__ __forceinline__ float ln_k(char* smem, unsigned st, float inv_T, float eps) {   // st: address of the row's entry for jb = 0
    const float q = (LDS(float, st) + LDS(float, st + 128)) + (LDS(float, st + 256) + LDS(float, st + 384));
    // 1 / sqrt(var + eps): v_rsq_f32 (1 ulp) + one Newton step instead of the ~25 instructions of an IEEE sqrt and divide
    const float v = fmaf(q * (1.0f / 128.0f) * inv_T, inv_T, eps);
    float r = __builtin_amdgcn_rsqf(v);
    r = r * fmaf(-0.5f * v * r, r, 1.5f);
    return inv_T * r;
}

// v_permlane32_swap: the upper half of the first operand and the lower half of the second change places
__device__ __forceinline__ float lower_half_to_both(float v) {   // every lane l gets the value of lane l & 31
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]);
}
__device__ __forceinline__ float upper_half_to_both(float v) {   // every lane l gets the value of lane 32 | l
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_of_halves(float v) {        // every lane l gets v[l & 31] + v[32 | l]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Three rows of the segmented scan down the registers: y[r] += y[r-1] on the lanes whose half has bit r (half 0) / 16 + r
// (half 1) of `cont` set -- the mask is the same for every lane of a half, so it goes into EXEC by two scalar bit-field
// extracts and the add itself is the only vector instruction per row.
template <int R>
__device__ __forceinline__ void scan3_t(float& ya, float& yb, float& yc, float yprev, unsigned cont) {
    asm volatile("s_bfe_i32 exec_lo, %4, %5\n\ts_bfe_i32 exec_hi, %4, %6\n\tv_add_f32 %0, %0, %3\n\t"
                 "s_bfe_i32 exec_lo, %4, %7\n\ts_bfe_i32 exec_hi, %4, %8\n\tv_add_f32 %1, %1, %0\n\t"
                 "s_bfe_i32 exec_lo, %4, %9\n\ts_bfe_i32 exec_hi, %4, %10\n\tv_add_f32 %2, %2, %1\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(ya), "+v"(yb), "+v"(yc)
                 : "v"(yprev), "s"(cont), "n"(0x10000 | R), "n"(0x10000 | (16 + R)), "n"(0x10000 | (R + 1)), "n"(0x10000 | (17 + R)),
                   "n"(0x10000 | (R + 2)), "n"(0x10000 | (18 + R))
                 : "scc");
}
__device__ __forceinline__ void scan3(float& ya, float& yb, float& yc, float yprev, unsigned cont, int r) {
    switch (r) {   // r is a constant wherever this is called (unrolled filler slots)
        case 1: scan3_t<1>(ya, yb, yc, yprev, cont); break;
        case 4: scan3_t<4>(ya, yb, yc, yprev, cont); break;
        case 7: scan3_t<7>(ya, yb, yc, yprev, cont); break;
        case 10: scan3_t<10>(ya, yb, yc, yprev, cont); break;
        default: scan3_t<13>(ya, yb, yc, yprev, cont); break;
    }
}

// WRITE_E = false: the launch whose e_out nobody reads (the last message-passing step of a forward: the decoder takes h only,
// epd_gnn.py:96) -- LayerNorm statistics and the scatter-add run as always, the row-major epilogue (residual read, e + e', store)
// does not exist: 1 GB less written and 1 GB less re-read at the target.
// STREAM: the cache policy of the row stores (ST_STREAM_* above).
template <bool WRITE_E, bool STREAM>
__global__ void __launch_bounds__(SYS_THREADS, 1) sys_edge_kernel(const CsrHeader* a_hdr, const int* __restrict__ a_dst, const int* __restrict__ a_src, const float* __restrict__ a_P,
                                                                       const float* a_e_in, float* a_e_out, float* __restrict__ a_agg, const float* __restrict__ a_hw,
                                                                       const int2* __restrict__ a_blk, const int2* __restrict__ a_seg, const int* __restrict__ a_head,
                                                                       const EdgeBlockHeader* __restrict__ a_tab, unsigned a_side_off, unsigned a_agg_bytes, unsigned a_P_bytes,
                                                                       int* a_flags, float a_eps, int a_residual) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef HEDGE_FORCE_ROLE   // development: register census of one role (tools/kernel_resources.py ... -DHEDGE_FORCE_ROLE=k)
    const int role = HEDGE_FORCE_ROLE, jb = wave & 3;
#else
    const int role = wave >> 2, jb = wave & 3;
#endif
    const int E = a_hdr->n_edges;
    const int nchunks = a_tab->n_groups;   // the workgroup takes a contiguous range of whole groups (4 blocks each)
    const int wg = xcd_major_wg();
    const int c0 = (int)((long long)wg * nchunks / gridDim.x);
    const int c1 = (int)((long long)(wg + 1) * nchunks / gridDim.x);
    if (c1 <= c0) return;
    const int b0 = 4 * c0, b1 = 4 * c1;
    const int nb = b1 - b0;
    if (nb <= 0) return;
    const float inv_T = a_hw[1];
    const float* hvec = a_hw + HW_HEADER_FLOATS;
    const half8* wimg = reinterpret_cast<const half8*>(a_hw + HW_HEADER_FLOATS + HW_VEC_FLOATS);

    half8 wh[8], wl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        wh[ks] = wimg[(((role * 4 + jb) * 8 + ks) * 2 + 0) * 64 + lane0];
        wl[ks] = wimg[(((role * 4 + jb) * 8 + ks) * 2 + 1) * 64 + lane0];
    }
    // every buffer starts finite: the pipeline's fill / drain ticks compute on them
    for (int i = tid; i < L_ZERO_END / 16; i += SYS_THREADS) LDS(uintx4, i * 16) = uintx4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 4 * H; i += SYS_THREADS) LDS(float, L_VEC + 4 * i) = hvec[i];
    // The weight registers must have ARRIVED before the tick loops: otherwise hipcc places their counted vmcnt waits at the first
    // uses inside the loop, where (loads and stores share the counter, in issue order) they wait for the tick's own accesses.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), a form the compiler's wait-count bookkeeping sees
    __syncthreads();
    auto ok = [&](int x) { return x >= b0 && x < b1; };
    auto clampb = [&](int x) { return x < b0 ? b0 : (x < b1 ? x : b1 - 1); };
    // The workgroup's edges are contiguous: its e rows and indices are addressed relative to its first edge, so the scalar
    // offsets stay small whatever the size of the arrays.  Rows past the end of the edge list are READ (the blocks are whole):
    // the forward keeps 32 zero rows behind the list (model.hip: pad rows), index reads are range-checked.
    const int e0 = a_blk[b0].x;
    const srd_t srd_ein = make_srd(a_e_in + (size_t)e0 * H, 0xffffffffu);
    const int n = lane0 & 31, hi = lane0 >> 5, rr = lane0 >> 3, cq = lane0 & 7;
    const unsigned v_eoff = opaque(rr * 512 + jb * 128 + cq * 16);   // row 8 j + rr of a block, this wave's 128-byte slab: + 4096 j by the scalar offset
    std::integral_constant<int, 0> even;
    std::integral_constant<int, 1> odd;

    // Instruction arbitration: role 2 (Linear 3 + statistics + the scatter-add's scan) is the longest instruction stream of
    // a tick and role 0 feeds the pipeline.
    if (role == 2) __builtin_amdgcn_s_setprio(HEDGE_PRIO2);        // the builtin takes an immediate
    else if (role == 0) __builtin_amdgcn_s_setprio(HEDGE_PRIO0);
    else __builtin_amdgcn_s_setprio(HEDGE_PRIO1);
    if (role == 0) {
        // ------------------------------------------------------------------ role 0
        // Rows of a block are dealt to the lanes two ways: e rows as 8 j + rr (the image E's swizzle needs j in the high bits),
        // P rows as 4 rr + j (so that a lane's four row indices are ONE 16-byte load of the index arrays).
        const srd_t srd_P = make_srd(a_P, a_P_bytes);
        const unsigned idx_bytes = (unsigned)s_clamp0(E - e0, 1 << 28) * 4u;
        const srd_t srd_dst = make_srd(a_dst + e0, idx_bytes), srd_src = make_srd(a_src + e0, idx_bytes);
        const unsigned v_poff = opaque(jb * 128 + cq * 16), v_ioff = opaque(rr * 16);
        const unsigned ps_w = opaque(L_PS + jb * TILE_B + 4 * rr * TILE_ROW_B + cq * 16);   // + j rows
        const unsigned ps_r = opaque(L_PS + jb * TILE_B + n * TILE_ROW_B + hi * 16);        // + 32 g
        const int kg = cq & 1, ksb = cq >> 2, half = (cq >> 1) & 1;
        const unsigned e_w = opaque(L_E + ((2 * jb + ksb) * 2 * 64 + ((rr ^ (2 * (ksb + 2 * kg))) + 32 * kg)) * 16 + half * 8);   // + 128 j, + 1024: lo part
        const unsigned e_r0 = opaque(L_E + eslot(n, hi, 0) * 16), e_r1 = opaque(L_E + eslot(n, hi, 1) * 16);
        const unsigned x1_w = opaque(L_X1 + 4 * jb * 1024 + lane0 * 16);
        floatx4 pi[4], pj[4];   // row-major quads of rows 4 rr + j: P_i / P_j of block x
#pragma unroll
        for (int j = 0; j < 4; ++j) { pi[j] = floatx4{0.f, 0.f, 0.f, 0.f}; pj[j] = pi[j]; }
        int rng = 0;            // range check of the fp16 split: set once an accumulator row turns NaN
        floatx4 eq[4];          // e rows of block x+1 on their way into the operand image E (the first block's: requested here)
#pragma unroll
        for (int j = 0; j < 4; ++j) eq[j] = bld4(srd_ein, v_eoff, j * 4096);
        // Two register sets: `acc` = the block's accumulators, c0v = (P_i + P_j) T1 of the next block (the first MFMA's C operand).
        floatx16 acc, c0v;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0v[r] = 0.f; acc[r] = 0.f; }
        intx4 di = bldi4(srd_dst, v_ioff, 0), si = bldi4(srd_src, v_ioff, 0);   // indices of the rows of block b0 ( = "x+1" of the first tick's requests)
        int2 be = a_blk[clampb(b0 + 1)];   // table entry of block x+2 (its .x = first edge): the rows and indices requested this tick
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value;   // parity of x: the images' double buffers are compile-time offsets
            const int x = b0 + t;
            SYS_STAMP(t, 0);
            auto prepare = [&]() {   // accumulator = (P_i[dst] + P_j[src]) * T1: row-major sum -> tile -> accumulator layout
#pragma unroll
                for (int j = 0; j < 4; ++j) LDS(floatx4, ps_w + j * TILE_ROW_B) = pi[j] + pj[j];   // P arrives times T1 (NodeArgs::p_scale)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const floatx4 v = LDS(floatx4, ps_r + 32 * g);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) c0v[4 * g + tt] = v[tt];
                }
            };
            SYS_STAMP(t, 1);
            // Requests, all at the top of the tick so that they have a whole tick to arrive (loads and stores complete in issue
            // order on one counter).  P rows of block x+1 (whole 128-byte lines: 8 lanes per row); its destinations also go to
            // role 2, which needs them in four ticks: its waves issue no vector loads at all.
            if (jb == 0 && cq == 0) LDS(intx4, v_ioff + (L_DR + ((x + 1) & (DR_SLOTS - 1)) * (BE * 4))) = di;
            const unsigned rel = (unsigned)(be.x - e0);
            auto p_loads = [&](int j) {
                pi[j] = bld4(srd_P, (unsigned)(di[j] << 10) + v_poff, 0);
                pj[j] = bld4(srd_P, (unsigned)(si[j] << 10) + v_poff + H * 4, 0);   // P_j: second half of the row
            };
            auto idx_loads = [&]() {
                di = bldi4(srd_dst, v_ioff + rel * 4, 0);     // indices of block x+2
                si = bldi4(srd_src, v_ioff + rel * 4, 0);
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) p_loads(j);
            SYS_STAMP(t, 7);   // (development) the eight P requests issued
            idx_loads();
            const int2 be_next = a_blk[clampb(x + 3)];
            // e of block x+1 -> operand image E (this role reads it next tick), one row group per call, between the MFMAs;
            // then the rows of block x+2 are requested into the same registers
            auto side = [&](int slot) {
                if (slot < 8 && !(slot & 1)) {
                    const int j = slot >> 1;
                    uintx2 h, l;
                    split4(eq[j][0], eq[j][1], eq[j][2], eq[j][3], h, l);
                    LDS(uintx2, e_w + (1 - PAR) * IMG_B + j * 128) = h;
                    LDS(uintx2, e_w + (1 - PAR) * IMG_B + j * 128 + 1024) = l;
                } else if (slot == 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) eq[j] = bld4(srd_ein, v_eoff, rel * 512 + j * 4096);
                    be = be_next;
                }
            };
            SYS_STAMP(t, 2);
            mlp_layer(acc, c0v, wh, wl, smem, e_r0 + PAR * IMG_B, e_r1 + PAR * IMG_B, side);
            SYS_STAMP(t, 3);
            // range check of the fp16 split: a value that does not fit an operand image is (inf, -inf) as a pair and turns every
            // accumulator of its row into NaN (hmlp.hip: check_rows) -- one comparison per tick, wave-uniform verdict
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            GM_SB;
            acc_to_image(acc, smem, x1_w + PAR * IMG_B);
            SYS_STAMP(t, 4);
            // Rotated tick: this wave's MFMAs open the tick -- while roles 1 and 2 merge statistics -- and the accumulators of
            // block x+1 (rows requested at the top of this tick) are prepared here, behind them, and cross the barrier in
            // registers: the three roles' matrix phases spread over the tick instead of piling up in its middle.
            prepare();
            SYS_STAMP(t, 5);
            lds_barrier();
            SYS_STAMP(t, 6);
        };
        // Ticks -1 .. nb + 2: one tick of fill (the e rows and indices of the first block come from the prologue), nb ticks in
        // which blocks enter, three that drain the pipeline: an even count (nb is a multiple of 4), taken as (odd, even) pairs --
        // at N = 5k a workgroup has 12 blocks, and every fill / drain tick counts.
#pragma unroll 1
        for (int t = -1; t <= nb + 1; t += 2) {
            tick(odd, t);
            tick(even, t + 1);
        }
        if (rng && lane0 == 0) atomicOr(a_flags, ERRF_SPLIT_RANGE);
    } else if (role == 1) {
        // ------------------------------------------------------------------ role 1
        // This role runs Linear 2 of block x-1 and its share of the epilogue of block x-3.
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int rng = 0;
        floatx4 er[EPI_SPLIT + 1];          // e rows (row-major quads, rows 8 j + rr) of block x-3 for the residual
        int st_a = e0, cnt_a = 0, st_b = e0, cnt_b = 0;  // blocks x-3, x-2
        int2 bi_c = a_blk[b0];                          // raw table entry of block x-1 (decoded a tick after its load)
        const float res_w = a_residual ? 1.f : 0.f;
        // LayerNorm gamma / beta of this lane's feature quad and the bias of Linear 2 in accumulator layout: constant over the launch
        const floatx4 gm = LDS(floatx4, L_VEC + (2 * H + 32 * jb + 4 * cq) * 4);
        const floatx4 bt = LDS(floatx4, L_VEC + (3 * H + 32 * jb + 4 * cq) * 4);
        floatx16 b2v;   // b2 T2 in accumulator layout: constant over the launch (this role has the registers)
        auto init_acc = [&](floatx16& dstv) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 v = LDS(floatx4, L_VEC + (32 * jb + 4 * hi + 8 * g) * 4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) dstv[4 * g + tt] = v[tt];
            }
        };
        init_acc(b2v);
        const unsigned st_r = opaque(L_ST + n * 4);
        const unsigned km_w = opaque(L_KM + jb * 128 + n * 4), km_r = opaque(L_KM + jb * 128 + rr * 4);
        const unsigned z_r = opaque(L_Z + jb * TILE_B + rr * TILE_ROW_B + cq * 16);
        const unsigned x_in = opaque(L_X1 + lane0 * 16), x_out = opaque(L_X2 + 4 * jb * 1024 + lane0 * 16);
        float* const e_out_wg = a_e_out + (size_t)e0 * H;
#pragma unroll
        for (int j = 0; j < EPI_SPLIT; ++j) er[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, P1 = 1 - PAR, P3 = 1 - PAR;   // parities of blocks x, x-1 (this Linear's), x-3
            const int x = b0 + t;
            SYS_STAMP(t, 0);
            // 1 / (T sigma) of the rows of block x-3: lane n (both halves) -> this wave's table
            if (WRITE_E) LDS(float, km_w) = ln_k(smem, st_r + P3 * 512, inv_T, a_eps);
            const int cnt_st = ok(x - 3) ? cnt_a : 0;   // rows of block x-3 that exist (none in the fill / drain ticks)
            const int2 bi_n = a_blk[clampb(x)];         // requested now, used at the end of the tick: the barrier's wait for the
                                                        // scalar-memory counter then finds it done
            const unsigned rel_a = (unsigned)(st_a - e0), rel_b = (unsigned)(st_b - e0);
            float kr;
            floatx4 zq;
            auto side = [&](int slot) {   // LayerNorm + e_out of block x-3, row group slot / SIDE_STRIDE
                if (!WRITE_E || slot >= EPI_SPLIT * SIDE_STRIDE) return;   // the other row groups are role 2's (balance of the roles' ticks)
                const int j = slot / SIDE_STRIDE;
                if (slot % SIDE_STRIDE == 0) {
                    kr = LDS(float, km_r + j * 32);
                    zq = LDS(floatx4, z_r + P3 * 4 * TILE_B + j * 8 * TILE_ROW_B);
                } else if (slot % SIDE_STRIDE == 2) {
                    floatx4 o;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) o[tt] = fmaf(er[j][tt], res_w, fmaf(zq[tt] * kr, gm[tt], bt[tt]));
                    // rows past the block's end (and every row of a fill / drain tick) lie beyond the resource's byte count: dropped
                    // (the block's position is in the resource's base: gfx9 subtracts a scalar offset from the byte count)
                    bst4s<STREAM ? ST_STREAM_E : 0>(make_srd(e_out_wg + (size_t)(rel_a + 8 * j) * H, (unsigned)s_clamp0(cnt_st - 8 * j, 8) * 512u), v_eoff, 0, o);
                    er[j] = bld4(srd_ein, v_eoff, rel_b * 512 + j * 4096);   // residual rows of block x-2: a whole tick to arrive
                }
            };
            SYS_STAMP(t, 1);
            SYS_STAMP(t, 2);
            mlp_layer(acc, b2v, wh, wl, smem, x_in + P1 * IMG_B, x_in + P1 * IMG_B, side);
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            GM_SB;
            acc_to_image(acc, smem, x_out + P1 * IMG_B);
            SYS_STAMP(t, 3);
            SYS_STAMP(t, 4);
            st_a = st_b; cnt_a = cnt_b;
            st_b = bi_c.x; cnt_b = bi_c.y & 0xff;
            bi_c = bi_n;
            SYS_STAMP(t, 5);
            lds_barrier();
            SYS_STAMP(t, 6);
        };
#pragma unroll 1
        for (int t = -1; t <= nb + 1; t += 2) {
            tick(odd, t);
            tick(even, t + 1);
        }
        if (rng && lane0 == 0) atomicOr(a_flags, ERRF_SPLIT_RANGE);
    } else {
        // ------------------------------------------------------------------ role 2
        // Aggregation of block x-3 on a TRANSPOSED view of its LayerNorm input: lane (f, h) = feature 32 jb + f, rows 16 h ..
        // 16 h + 15 in 16 registers.  The segment structure of the destination-sorted rows is the same for every feature, so it
        // lives in scalar registers (continuation / last-row bits from the block tables): the segmented scan is 15 masked
        // adds down the registers, a finished segment is one 128-byte store per half-wave.  The sum of a segment that is
        // still open at the end of a half (or block) travels on in `carry` and joins the first row stored afterwards.
        floatx16 acc;
        int rng = 0;
        float carry = 0.f;              // open segment's sum from the previous block (this lane's feature; both halves hold it)
        int cnt_a = 0, fl_a = 0, cnt_b = 0, fl_b = 0;    // blocks x-3, x-2 (cnt = 0 while there is none)
        unsigned cont_a = 0, last_a = 0, cont_b = 0, last_b = 0;
        int head_a = -1, head_b = -1;   // destination whose segment began in an earlier group (its sum over this group goes to the side buffer)
        int2 bn = a_blk[b0], sn = make_int2(0, 0);      // table entries of the block the next fetch() handles
        const float gam = LDS(float, L_VEC + (2 * H + 32 * jb + n) * 4), bet = LDS(float, L_VEC + (3 * H + 32 * jb + n) * 4);
        // its share of the LayerNorm + e_out epilogue (row groups EPI_SPLIT .. 3 of block x-3, row-major: 8 lanes per row)
        const float res_w = a_residual ? 1.f : 0.f;
        floatx4 er[4 - EPI_SPLIT + 1];  // e rows of block x-3 for the residual (the groups of this role)
#pragma unroll
        for (int j = 0; j < 4 - EPI_SPLIT; ++j) er[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        int st_a = e0, st_b = e0;       // first edge of blocks x-3, x-2
        const unsigned km_re = opaque(L_KM2 + jb * 128 + rr * 4);                          // + 32 j
        const unsigned z_e = opaque(L_Z + jb * TILE_B + rr * TILE_ROW_B + cq * 16);         // + 8 j rows
        float* const e_out_wg = a_e_out + (size_t)e0 * H;
        // this wave's entry of row n in the statistics tables (ST; KM2 at + L_KM2 - L_ST) and, relative to a resource that starts
        // L_ST bytes before agg, this lane's column of an agg row
        const unsigned v_aoff = opaque(L_ST + jb * 128 + n * 4);
        const unsigned st_r = opaque(L_ST + n * 4);
        const unsigned km_rq = opaque(L_KM2 + jb * 128 + 64 * hi);                         // + 16 c: rows 16 hi + 4 c .. + 3
        const unsigned z_w = opaque(L_Z + jb * TILE_B + n * TILE_ROW_B + hi * 16);          // + 32 g
        const unsigned z_t = opaque(L_Z + jb * TILE_B + 16 * hi * TILE_ROW_B + n * 4);      // + r rows: transposed reads
        const unsigned x_in = opaque(L_X2 + lane0 * 16);
        const unsigned dr_r = opaque(L_DR + 64 * hi);
        const srd_t srd_agg = make_srd(reinterpret_cast<const char*>(a_agg) - L_ST, a_agg_bytes + L_ST);   // see v_aoff
        floatx16 b3v;   // stays in LDS: read at the top of every tick
        auto init_acc = [&]() {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 v = LDS(floatx4, L_VEC + (H + 32 * jb + 4 * hi + 8 * g) * 4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) b3v[4 * g + tt] = v[tt];
            }
        };
        init_acc();
        auto fetch = [&](int x, int2 bi, int2 si, int hd, int& st, int& cnt, int& fl, unsigned& cont, unsigned& last, int& head) {
            st = bi.x;
            if (!ok(x)) { cnt = 0; fl = 0; cont = 0; last = 0; return; }
            cnt = bi.y & 0xff;
            fl = bi.y >> 8;
            cont = (unsigned)si.x;
            last = (unsigned)si.y;
            if (fl & 1) head = hd;
        };
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, P2 = PAR, P3 = 1 - PAR;   // parities of blocks x-2, x-3
            const int x = b0 + t;
            // table entries of block x (decoded next tick): requested at the top, so that the barrier's scalar-memory wait finds them done
            const int2 bn_n = a_blk[clampb(x)], sn_n = a_seg[clampb(x)];
            const int hd_n = a_head[clampb(x - 1) >> 2];   // head of block x-1's group (used if that block opens its group)
            init_acc();                                     // b3' T3 -> the first MFMA's C operand (lands under the statistics merge)
            const bool agg_on = ok(x - 3);
            SYS_STAMP(t, 0);
            // 1 / (T sigma) of the rows of block x-3 (lane = row) -> this wave's table, read back per register row below
            LDS(float, v_aoff + (L_KM2 - L_ST)) = ln_k(smem, st_r + P3 * 512, inv_T, a_eps);
            SYS_STAMP(t, 1);
            const unsigned cont = agg_on ? cont_a : 0u, last = agg_on ? last_a : 0u;
            const int cnt_st = agg_on ? cnt_a : 0;
            const unsigned rel_a = (unsigned)(st_a - e0), rel_b = (unsigned)(st_b - e0);
            floatx16 y;     // a vector: the store loop below indexes it with a (wave-uniform) run-time row
            floatx4 kq;
            float zz[4];
            float ek;
            floatx4 ezq;
            float cpend = 0.f;
            // value of an open segment handed to the next half: half 0 -> half 1 inside the tick, half 1 -> half 0 of the next block
            auto side = [&](int slot) {
                if (slot < 8) {
                    const int c = slot >> 1;
                    if (!(slot & 1)) {   // requests of rows 4c .. 4c+3: their k, their z values of this lane's feature
                        kq = LDS(floatx4, km_rq + 16 * c);
#pragma unroll
                        for (int i = 0; i < 4; ++i) zz[i] = LDS(float, z_t + P3 * 4 * TILE_B + (4 * c + i) * TILE_ROW_B);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[4 * c + i] = fmaf(zz[i] * kq[i], gam, bet);
                    }
                } else if (slot < 13) {
                    // segmented scan down the registers: y[r] += y[r-1] where row r continues row r-1's segment
                    const int r = 3 * (slot - 8) + 1;
                    float ya = y[r], yb = y[r + 1], yc = y[r + 2];
                    const float yp = y[r - 1];
                    scan3(ya, yb, yc, yp, cont, r);
                    y[r] = ya; y[r + 1] = yb; y[r + 2] = yc;
                } else if (slot == 13) {
                    // what enters each half at its first row: half 0 the previous block's open sum, half 1 the sum half 0 leaves open
                    const float c0v = (cont & 1u) ? carry : 0.f;
                    const float open0 = y[15] + ((last & 0xffffu) ? 0.f : c0v);        // meaningful in half 0
                    const float from0 = lower_half_to_both(open0);
                    const float c1v = (cont & 0x10000u) ? from0 : 0.f;
                    cpend = hi ? c1v : c0v;
                } else if (WRITE_E && EPI_SPLIT < 4 && slot >= 14 && slot < 14 + 2 * (4 - EPI_SPLIT)) {
                    // this role's share of the e_out epilogue: row group j, loads at an even slot, arithmetic + store at the next
                    {
                        const int j = EPI_SPLIT + ((slot - 14) >> 1);
                        if (!((slot - 14) & 1)) {
                            ek = LDS(float, km_re + j * 32);
                            ezq = LDS(floatx4, z_e + P3 * 4 * TILE_B + j * 8 * TILE_ROW_B);
                        } else {
                            const floatx4 gmq = LDS(floatx4, L_VEC + (2 * H + 32 * jb + 4 * cq) * 4);   // read per use: registers are scarcer than LDS slots here
                            const floatx4 btq = LDS(floatx4, L_VEC + (3 * H + 32 * jb + 4 * cq) * 4);
                            floatx4 o;
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) o[tt] = fmaf(er[j - EPI_SPLIT][tt], res_w, fmaf(ezq[tt] * ek, gmq[tt], btq[tt]));
                            bst4s<STREAM ? ST_STREAM_E : 0>(make_srd(e_out_wg + (size_t)(rel_a + 8 * j) * H, (unsigned)s_clamp0(cnt_st - 8 * j, 8) * 512u), v_eoff, 0, o);
                            er[j - EPI_SPLIT] = bld4(srd_ein, v_eoff, rel_b * 512 + j * 4096);   // residual rows of block x-2: a whole tick to arrive
                        }
                    }
                }
            };
            SYS_STAMP(t, 2);
            auto stores = [&]() {
                // Stores of the finished segments: one iteration per row index r that ends a segment in either half (about two
                // per tick on a dense graph), ascending, so that the first store of a half takes its pending carry.  A segment's
                // sum goes to its agg row, or -- the piece of a segment that began in an earlier group -- to this group's row of
                // the side buffer (one allocation with agg: byte offsets from a_agg).
                unsigned pend = (last | (last >> 16)) & 0xffffu;
                const unsigned side_row = (a_side_off + (unsigned)(((x - 3) >> 2) * H)) * 4u;
                const unsigned drp = dr_r + ((x - 3) & (DR_SLOTS - 1)) * (BE * 4);
                while (pend) {
                    const int r = __builtin_ctz(pend);
                    pend &= pend - 1;
                    const unsigned long long mk = (unsigned long long)(0u - ((last >> r) & 1u)) | ((unsigned long long)(0u - ((last >> (16 + r)) & 1u)) << 32);
                    if (__builtin_amdgcn_inverse_ballot_w64(mk)) {
                        const int d = LDS(int, drp + 4 * r);
                        const unsigned off = (d == head_a ? side_row : (unsigned)d * (H * 4u)) + v_aoff;
                        bst1s<STREAM ? ST_STREAM_AGG : 0>(srd_agg, off, 0, y[r] + cpend);
                        cpend = 0.f;
                    }
                }
                // the sum left open at the end of the block (half 1's last row) becomes the next block's carry, in both halves
                const float nc = upper_half_to_both(y[15] + cpend);
                carry = (((last >> 31) & 1u) || !agg_on || cnt_a < BE) ? 0.f : nc;
            };
            mlp_layer(acc, b3v, wh, wl, smem, x_in + P2 * IMG_B, x_in + P2 * IMG_B, side);
            SYS_STAMP(t, 3);   // 24 MFMAs with the scatter-add between them
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            {
                // LayerNorm statistics of the scaled accumulators: the image's W3 / b3 are centred over the output features, so a
                // row's outputs have zero mean and its variance is the mean square.  Sum of squares of this wave's 32 features
                // of row n (16 per lane half); raw accumulators to Z.
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q = fmaf(acc[r], acc[r], q);
                LDS(float, v_aoff + P2 * 512) = sum_of_halves(q);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 z;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) z[tt] = acc[4 * g + tt];
                    LDS(floatx4, z_w + P2 * 4 * TILE_B + 32 * g) = z;
                }
            }
            SYS_STAMP(t, 4);   // statistics + Z written
            stores();
            cnt_a = cnt_b; fl_a = fl_b; cont_a = cont_b; last_a = last_b;
            if (fl_b & 1) head_a = head_b;
            st_a = st_b;
            fetch(x - 1, bn, sn, hd_n, st_b, cnt_b, fl_b, cont_b, last_b, head_b);
            bn = bn_n;
            sn = sn_n;
            SYS_STAMP(t, 5);
            lds_barrier();
            SYS_STAMP(t, 6);   // every wave of the workgroup has finished the tick
        };
#pragma unroll 1
        for (int t = -1; t <= nb + 1; t += 2) {
            tick(odd, t);
            tick(even, t + 1);
        }
        if (rng && lane0 == 0) atomicOr(a_flags, ERRF_SPLIT_RANGE);
    }
}

// ------------------------------------------------------------------------------------------
// Encoder phi_e in the same weight-stationary form (hidden 128, three Linears, 4 raw features per edge, rows in sorted order):
//   e = LayerNorm(W3 relu(W2 relu(W1 x + b1) + b2) + b3)      (epd_gnn.py:30-33,72-84,88)
// Role 0: scale of every raw row (its own power of two, hmlp.h), Linear 1 (one k-group: the B fragment is built in registers
// from the 16-byte row), image X1; and -- it has 3 MFMAs per tick against the others' 24 -- the whole LayerNorm + store epilogue
// of block x-3.  Role 1: Linear 2.  Role 2: Linear 3, statistics, raw accumulators to the Z tiles.  The row scales travel
// through a ring in LDS (the biases of all three Linears and the LayerNorm's eps carry them).  Blocks are plain runs of 32 rows.
// ------------------------------------------------------------------------------------------
constexpr int LE_X1 = 0;
constexpr int LE_X2 = LE_X1 + 2 * IMG_B;
constexpr int LE_Z = LE_X2 + 2 * IMG_B;            // [2][4 jb] tiles
constexpr int LE_ST = LE_Z + 8 * TILE_B;           // [2][4 jb][32 rows] floats
constexpr int LE_KM = LE_ST + 2 * 4 * 32 * 4;      // [4 jb][32] floats: 1 / (T sigma) per row (role 0)
constexpr int RS_SLOTS = 8;
constexpr int LE_RS = LE_KM + 4 * 32 * 4;          // [RS_SLOTS][32] floats: power-of-two scale of every row of a block
constexpr int LE_ZERO_END = LE_RS + RS_SLOTS * 32 * 4;
constexpr int LE_VEC = LE_ZERO_END;                // 5 x 128 floats
constexpr size_t ENC_LDS_BYTES = LE_VEC + 5 * H * 4;

__global__ void __launch_bounds__(SYS_THREADS, 1) sys_enc_kernel(const CsrHeader* a_hdr, const float* __restrict__ a_x, float* __restrict__ a_e_out,
                                                                      const float* __restrict__ a_hw, int* a_flags, float a_eps, int a_pad_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, jb = wave & 3;
    const int E = a_hdr->n_edges;
    const int nblk = (E + BE - 1) / BE;
    const int wg = xcd_major_wg();
    const int b0 = (int)((long long)wg * nblk / gridDim.x), b1 = (int)((long long)(wg + 1) * nblk / gridDim.x);
    const int nb = b1 - b0;
    // The processor edge kernel reads whole 32-row blocks: the kEdgePadRows rows behind row E hold zeros (zero_edge_pad_rows does that
    // for the other encoders).  Nobody else writes them: stores past a block's last row are dropped by their buffer bound.
    if (a_pad_rows && blockIdx.x == 0)
        for (int i = tid; i < kEdgePadRows * H / 4; i += SYS_THREADS) reinterpret_cast<floatx4*>(a_e_out + (size_t)E * H)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (nb <= 0) return;
    const float inv_T = a_hw[1], cap = a_hw[2];
    const float* hvec = a_hw + HW_HEADER_FLOATS;
    const half8* wimg = reinterpret_cast<const half8*>(a_hw + HW_HEADER_FLOATS + HW_VEC_FLOATS);
    half8 wh[8], wl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {   // role 0 multiplies one k-group only (its image holds zeros beyond): it loads that one
        const int kq = role == 0 ? 0 : ks;
        wh[ks] = wimg[(((role * 4 + jb) * 8 + kq) * 2 + 0) * 64 + lane0];
        wl[ks] = wimg[(((role * 4 + jb) * 8 + kq) * 2 + 1) * 64 + lane0];
    }
    for (int i = tid; i < LE_ZERO_END / 16; i += SYS_THREADS) LDS(uintx4, i * 16) = uintx4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 5 * H; i += SYS_THREADS) LDS(float, LE_VEC + 4 * i) = hvec[i];
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const int n = lane0 & 31, hi = lane0 >> 5, rr = lane0 >> 3, cq = lane0 & 7;
    const int e0 = b0 * BE;                               // first row of the workgroup
    const int rows_wg = (b1 * BE < E ? b1 * BE : E) - e0;  // its rows
    std::integral_constant<int, 0> even;
    std::integral_constant<int, 1> odd;
    // bias of this role's Linear in accumulator layout (vec slot: role 0 -> b1 T1 (4), role 1 -> b2 T2 (0), role 2 -> b3' T3 (1))
    floatx16 bv;
    {
        const int slot = role == 0 ? 4 : role - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = LDS(floatx4, LE_VEC + (slot * H + 32 * jb + 4 * hi + 8 * g) * 4);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) bv[4 * g + tt] = v[tt];
        }
    }
    const unsigned rs_l = opaque(LE_RS + n * 4);   // + 128 slot
    int rng = 0;
    floatx16 acc;
    auto scaled_bias = [&](float sc) {
        floatx16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = bv[r] * sc;
        return c;
    };
    if (role == 0) {
        // raw rows: lane n of the lower half holds the 16-byte row n of the block; range-checked reads (rows past E read zeros)
        const srd_t srd_x = make_srd(a_x + (size_t)e0 * 4, (unsigned)(rows_wg > 0 ? rows_wg : 0) * 16u);
        float* const e_out_wg = a_e_out + (size_t)e0 * H;
        const unsigned v_xoff = opaque(n * 16 + (hi ? 0x40000000u : 0u));   // the upper half reads nothing (out of range: zeros)
        const unsigned v_eoff = opaque(rr * 512 + jb * 128 + cq * 16);
        const unsigned x1_w = opaque(LE_X1 + 4 * jb * 1024 + lane0 * 16);
        const unsigned st_r = opaque(LE_ST + n * 4);
        const unsigned km_w = opaque(LE_KM + jb * 128 + n * 4), km_r = opaque(LE_KM + jb * 128 + rr * 4);
        const unsigned z_r = opaque(LE_Z + jb * TILE_B + rr * TILE_ROW_B + cq * 16);
        const floatx4 gm = LDS(floatx4, LE_VEC + (2 * H + 32 * jb + 4 * cq) * 4);
        const floatx4 bt = LDS(floatx4, LE_VEC + (3 * H + 32 * jb + 4 * cq) * 4);
        const float T3 = 1.0f / inv_T;
        floatx4 xq = bld4(srd_x, v_xoff, 0);   // rows of the first tick's block
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, P3 = 1 - PAR;
            const int x = b0 + t;
            // ---- this block's rows: scale, B fragment, bias
            float mx = fmaxf(fmaxf(fabsf(xq[0]), fabsf(xq[1])), fmaxf(fabsf(xq[2]), fabsf(xq[3])));
            float sc = cap;                        // zero (or non-finite) row: the bias alone, at the largest scale allowed
            if (mx > 0.f && mx < 3.0e38f) {
                int ex;
                (void)frexpf(mx, &ex);             // mx = f 2^ex, f in [0.5, 1)  ->  mx 2^(7 - ex) in [2^6, 2^7)
                sc = fminf(ldexpf(1.f, min(7 - ex, 100)), cap);
            }
            sc = lower_half_to_both(sc);           // both halves of the wave hold row n's scale
            if (jb == 0 && hi == 0) LDS(float, rs_l + (x & (RS_SLOTS - 1)) * 128) = sc;
            uintx2 h, l;
            split4(xq[0] * sc, xq[1] * sc, xq[2] * sc, xq[3] * sc, h, l);   // upper half: zeros (features 4 .. 7 do not exist)
            const half8 bh = __builtin_bit_cast(half8, uintx4{h[0], h[1], 0u, 0u}), bl = __builtin_bit_cast(half8, uintx4{l[0], l[1], 0u, 0u});
            const floatx16 c0v = scaled_bias(sc);
            // rows of the next block (clamped to the workgroup's range: the fill / drain ticks recompute a block, nothing is stored)
            {
                const int xn = x + 1 < b0 ? b0 : (x + 1 < b1 ? x + 1 : b1 - 1);
                xq = bld4(srd_x, v_xoff, (unsigned)(xn - b0) * 512u);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], bh, c0v, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], bh, acc, 0, 0, 0);
            // ---- LayerNorm + store of block x-3 (its statistics and tiles were written a tick ago)
            {
                const float rs3 = LDS(float, rs_l + ((x - 3) & (RS_SLOTS - 1)) * 128);
                const float q = (LDS(float, st_r + P3 * 512) + LDS(float, st_r + P3 * 512 + 128)) + (LDS(float, st_r + P3 * 512 + 256) + LDS(float, st_r + P3 * 512 + 384));
                const float tt3 = T3 * rs3;        // the accumulators of Linear 3 carry T3 x the row's scale
                const float v = fmaf(a_eps * tt3, tt3, q * (1.0f / 128.0f));
                float r = __builtin_amdgcn_rsqf(v);
                r = r * fmaf(-0.5f * v * r, r, 1.5f);
                LDS(float, km_w) = r;
                const int xb = x - 3;
                const int cnt = (xb >= b0 && xb < b1) ? min(BE, E - xb * BE) : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float kr = LDS(float, km_r + j * 32);
                    const floatx4 zq = LDS(floatx4, z_r + P3 * 4 * TILE_B + j * 8 * TILE_ROW_B);
                    floatx4 o;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) o[tt] = fmaf(zq[tt] * kr, gm[tt], bt[tt]);
                    bst4s<HENC_ST_AUX>(make_srd(e_out_wg + (size_t)((xb - b0) * BE + 8 * j) * H, (unsigned)s_clamp0(cnt - 8 * j, 8) * 512u), v_eoff, 0, o);
                }
            }
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            GM_SB;
            acc_to_image(acc, smem, x1_w + PAR * IMG_B);
            lds_barrier();
        };
#pragma unroll 1
        for (int t = 0; t <= nb + 2; t += 2) {   // ticks 0 .. nb + 2 (one more when nb is even: it drains like the one before it)
            tick(even, t);
            tick(odd, t + 1);
        }
    } else if (role == 1) {
        const unsigned x_in = opaque(LE_X1 + lane0 * 16), x_out = opaque(LE_X2 + 4 * jb * 1024 + lane0 * 16);
        auto nothing = [](int) {};
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, P1 = 1 - PAR;
            const int x = b0 + t;
            const floatx16 c0v = scaled_bias(LDS(float, rs_l + ((x - 1) & (RS_SLOTS - 1)) * 128));
            mlp_layer(acc, c0v, wh, wl, smem, x_in + P1 * IMG_B, x_in + P1 * IMG_B, nothing);
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            GM_SB;
            acc_to_image(acc, smem, x_out + P1 * IMG_B);
            lds_barrier();
        };
#pragma unroll 1
        for (int t = 0; t <= nb + 2; t += 2) {   // ticks 0 .. nb + 2 (one more when nb is even: it drains like the one before it)
            tick(even, t);
            tick(odd, t + 1);
        }
    } else {
        const unsigned x_in = opaque(LE_X2 + lane0 * 16);
        const unsigned st_w = opaque(LE_ST + jb * 128 + n * 4);
        const unsigned z_w = opaque(LE_Z + jb * TILE_B + n * TILE_ROW_B + hi * 16);
        auto nothing = [](int) {};
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, P2 = PAR;
            const int x = b0 + t;
            const floatx16 c0v = scaled_bias(LDS(float, rs_l + ((x - 2) & (RS_SLOTS - 1)) * 128));
            mlp_layer(acc, c0v, wh, wl, smem, x_in + P2 * IMG_B, x_in + P2 * IMG_B, nothing);
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) q = fmaf(acc[r], acc[r], q);
            LDS(float, st_w + P2 * 512) = sum_of_halves(q);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 z;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) z[tt] = acc[4 * g + tt];
                LDS(floatx4, z_w + P2 * 4 * TILE_B + 32 * g) = z;
            }
            lds_barrier();
        };
#pragma unroll 1
        for (int t = 0; t <= nb + 2; t += 2) {   // ticks 0 .. nb + 2 (one more when nb is even: it drains like the one before it)
            tick(even, t);
            tick(odd, t + 1);
        }
    }
    if (rng && lane0 == 0 && a_flags) atomicOr(a_flags, ERRF_SPLIT_RANGE);
}

// ------------------------------------------------------------------------------------------
// Processor node MLP in the same weight-stationary form (hidden 128, three Linears):
//   h' = h + LayerNorm(W3 relu(W2 relu(W1 [h, agg] + b1) + b2) + b3)          (epd_gnn.py:38,44-45,100-105)
// A node MLP's first Linear has 2H inputs: 128 KB of fp16 hi / lo weights, twice what a role's four waves hold.  Its h half is
// therefore taken out the way phi_e's node terms are: the kernel that projects a step's h (sys_proj_kernel: P for the next edge
// step) also writes Q = (W_h h + b1) T1 for the next node step, and Q enters Linear 1 here as the initial accumulator -- 3 H x H
// products per node in this kernel, the roles of the edge encoder's kernel:
//   role 0: Q rows of block x+1 and agg rows of block x+2 requested; agg of block x+1 -> operand image E (between the MFMAs);
//           Linear 1 on E from c0 = Q, ReLU, image X1; Q of block x+1 -> staging tile -> accumulator layout (behind the MFMAs);
//   role 1: Linear 2 (X1 -> X2) and row groups 0 / 1 of the epilogue of block x-3: LayerNorm, h + ., store (whole 128-byte lines);
//   role 2: Linear 3 (X2 -> raw accumulators Z + sums of squares) and row groups 2 / 3 of that epilogue.
// Blocks are plain runs of 32 rows.  agg arrives with the scatter-add's head partials already added (agg_stitch_kernel).  h may be
// updated in place: a block's residual rows are read (role 1 / 2, a tick ahead of their use) before the same waves store them.
// ------------------------------------------------------------------------------------------
constexpr int LN_E = 0;                            // [2] images of agg rows (eslot() order)
constexpr int LN_X1 = LN_E + 2 * IMG_B;
constexpr int LN_X2 = LN_X1 + 2 * IMG_B;
constexpr int LN_Z = LN_X2 + 2 * IMG_B;            // [2][4 jb] tiles
constexpr int LN_PS = LN_Z + 8 * TILE_B;           // [4 jb] tiles: role-0 staging of Q
constexpr int LN_ST = LN_PS + 4 * TILE_B;          // [2][4 jb][32 rows] floats
constexpr int LN_KM = LN_ST + 2 * 4 * 32 * 4;      // [2 roles][4 jb][32] floats: 1 / (T sigma) per row
constexpr int LN_ZERO_END = LN_KM + 2 * 4 * 32 * 4;
constexpr int LN_VEC = LN_ZERO_END;                // 4 x 128 floats: b2 T2 | b3' T3 | gamma | beta
constexpr size_t NODE_LDS_BYTES = LN_VEC + 4 * H * 4;
static_assert(NODE_LDS_BYTES <= 160 * 1024, "LDS budget");

__global__ void __launch_bounds__(SYS_THREADS, 1) sys_node_kernel(const float* a_h, const float* __restrict__ a_agg, const float* __restrict__ a_Q,
                                                                       float* a_h_out, const float* __restrict__ a_hw, int a_n, int* a_flags, float a_eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, jb = wave & 3;
    const int N = a_n;
    const int nblk = (N + BE - 1) / BE;
    const int wg = (int)blockIdx.x;
    const int b0 = (int)((long long)wg * nblk / gridDim.x), b1 = (int)((long long)(wg + 1) * nblk / gridDim.x);
    const int nb = b1 - b0;
    if (nb <= 0) return;
    const float inv_T = a_hw[1];
    const float* hvec = a_hw + HW_HEADER_FLOATS;
    const half8* wimg = reinterpret_cast<const half8*>(a_hw + HW_HEADER_FLOATS + HW_VEC_FLOATS);
    half8 wh[8], wl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        wh[ks] = wimg[(((role * 4 + jb) * 8 + ks) * 2 + 0) * 64 + lane0];
        wl[ks] = wimg[(((role * 4 + jb) * 8 + ks) * 2 + 1) * 64 + lane0];
    }
    for (int i = tid; i < LN_ZERO_END / 16; i += SYS_THREADS) LDS(uintx4, i * 16) = uintx4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 4 * H; i += SYS_THREADS) LDS(float, LN_VEC + 4 * i) = hvec[i];
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const int n = lane0 & 31, hi = lane0 >> 5, rr = lane0 >> 3, cq = lane0 & 7;
    const int r0 = b0 * BE;                                 // first row of the workgroup
    const int rows_wg = (b1 * BE < N ? b1 * BE : N) - r0;   // its rows
    const unsigned wg_bytes = (unsigned)(rows_wg > 0 ? rows_wg : 0) * 512u;   // reads past them return zeros, stores are dropped
    std::integral_constant<int, 0> even;
    std::integral_constant<int, 1> odd;
    const unsigned v_eoff = opaque(rr * 512 + jb * 128 + cq * 16);   // row 8 j + rr of a block, this wave's 128-byte slab: + 4096 j
    int rng = 0;
    if (role == 2) __builtin_amdgcn_s_setprio(HEDGE_PRIO2);
    else if (role == 0) __builtin_amdgcn_s_setprio(HEDGE_PRIO0);
    else __builtin_amdgcn_s_setprio(HEDGE_PRIO1);
    if (role == 0) {
        const srd_t srd_q = make_srd(a_Q + (size_t)r0 * H, wg_bytes), srd_a = make_srd(a_agg + (size_t)r0 * H, wg_bytes);
        const unsigned v_qoff = opaque(rr * 2048 + jb * 128 + cq * 16);   // row 4 rr + j: + 512 j
        const unsigned ps_w = opaque(LN_PS + jb * TILE_B + 4 * rr * TILE_ROW_B + cq * 16);   // + j rows
        const unsigned ps_r = opaque(LN_PS + jb * TILE_B + n * TILE_ROW_B + hi * 16);        // + 32 g
        const int kg = cq & 1, ksb = cq >> 2, half = (cq >> 1) & 1;
        const unsigned e_w = opaque(LN_E + ((2 * jb + ksb) * 2 * 64 + ((rr ^ (2 * (ksb + 2 * kg))) + 32 * kg)) * 16 + half * 8);   // + 128 j, + 1024: lo part
        const unsigned e_r0 = opaque(LN_E + eslot(n, hi, 0) * 16), e_r1 = opaque(LN_E + eslot(n, hi, 1) * 16);
        const unsigned x1_w = opaque(LN_X1 + 4 * jb * 1024 + lane0 * 16);
        floatx4 qv[4], eq[4];   // Q rows of block x+1 (rows 4 rr + j); agg rows of block x+1 (rows 8 j + rr) on their way into E
#pragma unroll
        for (int j = 0; j < 4; ++j) { qv[j] = floatx4{0.f, 0.f, 0.f, 0.f}; eq[j] = bld4(srd_a, v_eoff, j * 4096); }
        floatx16 acc, c0v;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0v[r] = 0.f;
        auto clampb = [&](int x) { return x < b0 ? b0 : (x < b1 ? x : b1 - 1); };
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value;
            const int x = b0 + t;
            const unsigned rel1 = (unsigned)(clampb(x + 1) - b0) * (BE * 512u), rel2 = (unsigned)(clampb(x + 2) - b0) * (BE * 512u);
#pragma unroll
            for (int j = 0; j < 4; ++j) qv[j] = bld4(srd_q, v_qoff, rel1 + j * 512);
            auto side = [&](int slot) {
                if (slot < 8 && !(slot & 1)) {
                    const int j = slot >> 1;
                    uintx2 h, l;
                    split4(eq[j][0], eq[j][1], eq[j][2], eq[j][3], h, l);
                    LDS(uintx2, e_w + (1 - PAR) * IMG_B + j * 128) = h;
                    LDS(uintx2, e_w + (1 - PAR) * IMG_B + j * 128 + 1024) = l;
                } else if (slot == 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) eq[j] = bld4(srd_a, v_eoff, rel2 + j * 4096);
                }
            };
            mlp_layer(acc, c0v, wh, wl, smem, e_r0 + PAR * IMG_B, e_r1 + PAR * IMG_B, side);
            rng |= __any(acc[0] != acc[0]) ? 1 : 0;
            GM_SB;
            acc_to_image(acc, smem, x1_w + PAR * IMG_B);
            // initial accumulators of block x+1: Q rows (already at this kernel's weight scale) row-major -> tile -> accumulator layout
#pragma unroll
            for (int j = 0; j < 4; ++j) LDS(floatx4, ps_w + j * TILE_ROW_B) = qv[j];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 v = LDS(floatx4, ps_r + 32 * g);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) c0v[4 * g + tt] = v[tt];
            }
            lds_barrier();
        };
#pragma unroll 1
        for (int t = -1; t <= nb + 2; t += 2) {   // one tick of fill, nb blocks, three (four when nb is odd) that drain the pipeline
            tick(odd, t);
            tick(even, t + 1);
        }
    } else {
        // roles 1 / 2: Linear 2 of block x-1 / Linear 3 of block x-2; each runs two row groups of the epilogue of block x-3
        const bool r2 = role == 2;
        const int eg0 = r2 ? 2 : 0;                       // first of this role's two epilogue row groups
        const srd_t srd_h = make_srd(a_h + (size_t)r0 * H, wg_bytes);
        float* const h_out_wg = a_h_out + (size_t)r0 * H;
        const floatx4 gm = LDS(floatx4, LN_VEC + (2 * H + 32 * jb + 4 * cq) * 4);
        const floatx4 bt = LDS(floatx4, LN_VEC + (3 * H + 32 * jb + 4 * cq) * 4);
        floatx16 bv;   // bias of this role's Linear in accumulator layout
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 v = LDS(floatx4, LN_VEC + ((r2 ? H : 0) + 32 * jb + 4 * hi + 8 * g) * 4);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) bv[4 * g + tt] = v[tt];
        }
        const unsigned st_r = opaque(LN_ST + n * 4);
        const unsigned km_w = opaque(LN_KM + (r2 ? 512 : 0) + jb * 128 + n * 4), km_r = opaque(LN_KM + (r2 ? 512 : 0) + jb * 128 + rr * 4);
        const unsigned z_r = opaque(LN_Z + jb * TILE_B + rr * TILE_ROW_B + cq * 16);
        const unsigned x_in = opaque((r2 ? LN_X2 : LN_X1) + lane0 * 16), x_out = opaque(LN_X2 + 4 * jb * 1024 + lane0 * 16);
        const unsigned st_w = opaque(LN_ST + jb * 128 + n * 4);
        const unsigned z_w = opaque(LN_Z + jb * TILE_B + n * TILE_ROW_B + hi * 16);
        floatx4 er[2];   // residual rows (h, row-major quads) of block x-3, this role's two row groups
        er[0] = er[1] = floatx4{0.f, 0.f, 0.f, 0.f};
        floatx16 acc;
        auto tick = [&](auto par_c, int t) {
            constexpr int PAR = decltype(par_c)::value, PIN = 1 - PAR, P3 = 1 - PAR;   // role 1: X1 of block x-1; role 2: X2 of block x-2 (PAR)
            const int x = b0 + t;
            LDS(float, km_w) = ln_k(smem, st_r + P3 * 512, inv_T, a_eps);   // 1 / (T sigma) of the rows of block x-3
            const int xb = x - 3;
            const int cnt = (xb >= b0 && xb < b1) ? min(BE, N - xb * BE) : 0;   // rows of block x-3 that exist
            const unsigned rel_a = (unsigned)((xb < b0 ? b0 : xb) - b0) * BE;   // its first row, relative to the workgroup's
            const int xr = x - 2 < b0 ? b0 : (x - 2 < b1 ? x - 2 : b1 - 1);
            const unsigned rel_b = (unsigned)(xr - b0) * (BE * 512u);           // block x-2: the residual rows requested this tick
            float kr;
            floatx4 zq;
            auto side = [&](int slot) {   // LayerNorm + h + . of block x-3, row groups eg0 (slots 0 / 2) and eg0 + 1 (slots 6 / 8)
                if (slot != 0 && slot != 2 && slot != 6 && slot != 8) return;
                const int jj = slot >= 6, j = eg0 + jj;
                if (slot == 0 || slot == 6) {
                    kr = LDS(float, km_r + j * 32);
                    zq = LDS(floatx4, z_r + P3 * 4 * TILE_B + j * 8 * TILE_ROW_B);
                } else {
                    floatx4 o;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) o[tt] = er[jj][tt] + fmaf(zq[tt] * kr, gm[tt], bt[tt]);
                    bst4(make_srd(h_out_wg + (size_t)(rel_a + 8 * j) * H, (unsigned)s_clamp0(cnt - 8 * j, 8) * 512u), v_eoff, 0, o);
                    er[jj] = bld4(srd_h, v_eoff, rel_b + j * 4096);   // residual rows of block x-2: a whole tick to arrive
                }
            };
            if (!r2) {
                mlp_layer(acc, bv, wh, wl, smem, x_in + PIN * IMG_B, x_in + PIN * IMG_B, side);
                rng |= __any(acc[0] != acc[0]) ? 1 : 0;
                GM_SB;
                acc_to_image(acc, smem, x_out + PIN * IMG_B);
            } else {
                mlp_layer(acc, bv, wh, wl, smem, x_in + PAR * IMG_B, x_in + PAR * IMG_B, side);
                rng |= __any(acc[0] != acc[0]) ? 1 : 0;
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q = fmaf(acc[r], acc[r], q);
                LDS(float, st_w + PAR * 512) = sum_of_halves(q);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 z;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) z[tt] = acc[4 * g + tt];
                    LDS(floatx4, z_w + PAR * 4 * TILE_B + 32 * g) = z;
                }
            }
            lds_barrier();
        };
#pragma unroll 1
        for (int t = -1; t <= nb + 2; t += 2) {
            tick(odd, t);
            tick(even, t + 1);
        }
    }
    if (rng && lane0 == 0 && a_flags) atomicOr(a_flags, ERRF_SPLIT_RANGE);
}

// ------------------------------------------------------------------------------------------
// Projections of a step's h for the NEXT step, weight-stationary: twelve waves, wave w = output block w of
//   [ P_i | P_j | Q ] = h [ W_i | W_j | W_h ]^T + [ b1(phi_e) | 0 | b1(phi_v) ]      (3 x 128 outputs, 128 inputs)
// P leaves at the scale of the edge kernel that adds it into its accumulators, Q at the node kernel's (NodeArgs::p_scale).  One
// block of 32 rows per tick and ONE barrier: rows of block x+2 requested, rows of block x+1 -> operand image (waves 0 .. 7, two
// (row group, slab) units each), 24 MFMAs on the image of block x, accumulators -> the wave's own tile -> whole 128-byte lines.
// ------------------------------------------------------------------------------------------
#ifndef HPROJ_Q_AUX
#define HPROJ_Q_AUX 16   // cache policy of the Q stores (16 = sc1)
#endif
constexpr int LP_E = 0;                             // [2] images of h rows
constexpr int LP_T = LP_E + 2 * IMG_B;              // [12 waves] output tiles
constexpr size_t PROJ_LDS_BYTES = LP_T + 12 * TILE_B;

__global__ void __launch_bounds__(SYS_THREADS, 1) sys_proj_kernel(const float* __restrict__ a_h, float* __restrict__ a_P, float* __restrict__ a_Q,
                                                                       const float* __restrict__ a_wp, const float* __restrict__ a_wq,
                                                                       const float* __restrict__ a_sp, const float* __restrict__ a_sq, int a_n, int* a_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a_n;
    const int nblk = (N + BE - 1) / BE;
    const int b0 = (int)((long long)blockIdx.x * nblk / gridDim.x), b1 = (int)((long long)(blockIdx.x + 1) * nblk / gridDim.x);
    const int nb = b1 - b0;
    if (nb <= 0) return;
    // Linear images of hmlp.h: [t, 1/U, U, cap | bias U (out_pad floats) | fragments [out / 32][K / 16][2][64 lanes] half8]
    const bool isq = wave >= 8;
    const float* img = isq ? a_wq : a_wp;
    const int ob = isq ? wave - 8 : wave;                 // output block inside its image
    const int out_pad = isq ? H : 2 * H;
    const float out_scale = img[1] * (isq ? (a_sq ? *a_sq : 1.f) : (a_sp ? *a_sp : 1.f));
    const half8* frag = reinterpret_cast<const half8*>(img + 4 + out_pad);
    half8 wh[8], wl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        wh[ks] = frag[((size_t)(ob * 8 + ks) * 2 + 0) * 64 + lane0];
        wl[ks] = frag[((size_t)(ob * 8 + ks) * 2 + 1) * 64 + lane0];
    }
    const int n = lane0 & 31, hi = lane0 >> 5, rr = lane0 >> 3, cq = lane0 & 7;
    floatx16 bv;   // bias (times U) of this wave's 32 outputs in accumulator layout
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(img + 4 + 32 * ob + 8 * g + 4 * hi);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) bv[4 * g + tt] = v[tt];
    }
    for (int i = tid; i < 2 * IMG_B / 16; i += SYS_THREADS) LDS(uintx4, LP_E + i * 16) = uintx4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const int r0 = b0 * BE;
    const int rows_wg = (b1 * BE < N ? b1 * BE : N) - r0;
    const unsigned wg_bytes = (unsigned)(rows_wg > 0 ? rows_wg : 0) * 512u;
    const srd_t srd_h = make_srd(a_h + (size_t)r0 * H, wg_bytes);
    // image production: waves 0 .. 7 take units u = 2 wave, 2 wave + 1 of the 16 (row group j = u >> 2, slab sb = u & 3) units of a block
    const bool producer = wave < 8;
    unsigned v_in[2], e_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = 2 * (wave & 7) + i, j = u >> 2, sb = u & 3;
        const int kg = cq & 1, ksb = cq >> 2, half = (cq >> 1) & 1;
        v_in[i] = opaque((unsigned)((8 * j + rr) * 512 + sb * 128 + cq * 16));
        e_w[i] = opaque((unsigned)(LP_E + ((2 * sb + ksb) * 2 * 64 + ((rr ^ (2 * (ksb + 2 * kg))) + 32 * kg)) * 16 + half * 8 + j * 128));
    }
    const unsigned e_r0 = opaque(LP_E + eslot(n, hi, 0) * 16), e_r1 = opaque(LP_E + eslot(n, hi, 1) * 16);
    const unsigned t_w = opaque(LP_T + wave * TILE_B + n * TILE_ROW_B + hi * 16);        // + 32 g: accumulator layout
    const unsigned t_r = opaque(LP_T + wave * TILE_B + rr * TILE_ROW_B + cq * 16);       // + 8 j rows: row-major
    float* const out_wg = isq ? a_Q + (size_t)r0 * H + 32 * ob : a_P + (size_t)r0 * 2 * H + 32 * ob;
    const unsigned out_row_b = isq ? 512u : 1024u;
    const unsigned v_out = opaque((unsigned)(rr * out_row_b + cq * 16));                  // row 8 j + rr: + 8 j rows by the resource's base
    std::integral_constant<int, 0> even;
    std::integral_constant<int, 1> odd;
    auto clampb = [&](int x) { return x < b0 ? b0 : (x < b1 ? x : b1 - 1); };
    floatx4 hq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hq[i] = producer ? bld4(srd_h, v_in[i], 0) : floatx4{0.f, 0.f, 0.f, 0.f};   // rows of block b0 ("x+1" of the fill tick)
    int rng = 0;
    floatx16 acc;
    auto tick = [&](auto par_c, int t) {
        constexpr int PAR = decltype(par_c)::value;
        const int x = b0 + t;
        const unsigned rel2 = (unsigned)(clampb(x + 2) - b0) * (BE * 512u);
        const int cnt = (x >= b0 && x < b1) ? min(BE, N - x * BE) : 0;
        auto side = [&](int slot) {
            if (producer && (slot == 0 || slot == 2)) {   // rows of block x+1 -> image
                const int i = slot >> 1;
                uintx2 h, l;
                split4(hq[i][0], hq[i][1], hq[i][2], hq[i][3], h, l);
                LDS(uintx2, e_w[i] + (1 - PAR) * IMG_B) = h;
                LDS(uintx2, e_w[i] + (1 - PAR) * IMG_B + 1024) = l;
            } else if (producer && slot == 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) hq[i] = bld4(srd_h, v_in[i], rel2);
            }
        };
        mlp_layer(acc, bv, wh, wl, smem, e_r0 + PAR * IMG_B, e_r1 + PAR * IMG_B, side);
        rng |= __any(acc[0] != acc[0]) ? 1 : 0;
        // outputs of block x: accumulator layout -> the wave's own tile -> row-major, whole 128-byte lines (rows that do not exist
        // lie beyond the resource's byte count)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            floatx4 z;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) z[tt] = acc[4 * g + tt] * out_scale;
            LDS(floatx4, t_w + 32 * g) = z;
        }
        const unsigned rowx = (unsigned)((x < b0 ? b0 : x) - b0) * BE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const floatx4 o = LDS(floatx4, t_r + j * 8 * TILE_ROW_B);
            // Q is read once, by the node kernel behind the next edge launch (2 GB of streaming later): written through, not kept in L2;
            // P is gathered by that edge launch: default policy
            if (isq) bst4s<HPROJ_Q_AUX>(make_srd(out_wg + (size_t)(rowx + 8 * j) * (out_row_b / 4), (unsigned)s_clamp0(cnt - 8 * j, 8) * out_row_b), v_out, 0, o);
            else bst4(make_srd(out_wg + (size_t)(rowx + 8 * j) * (out_row_b / 4), (unsigned)s_clamp0(cnt - 8 * j, 8) * out_row_b), v_out, 0, o);
        }
        lds_barrier();
    };
#pragma unroll 1
    for (int t = -1; t <= nb; t += 2) {   // one tick of fill, nb blocks (one more when nb is even: it recomputes the last block, stores nothing)
        tick(odd, t);
        tick(even, t + 1);
    }
    if (rng && lane0 == 0 && a_flags) atomicOr(a_flags, ERRF_SPLIT_RANGE);
}

// agg rows of the nodes whose in-edge segment crosses groups of the scatter-add: + the head partials the later groups hold, in
// group order (hedge.h: stitch / head lists) -- what hm_node_kernel does while it reads agg; the systolic node kernel reads plain rows.
// Walks the list of those nodes (EdgeBlocks::stitch_list, written with the tables once per structure: about one node in seven of a
// dense scene), eight threads per entry, instead of asking every node.
__global__ void __launch_bounds__(256) agg_stitch_kernel(float* __restrict__ agg, const float* __restrict__ side, const int* __restrict__ stitch,
                                                          const int* __restrict__ stitch_list, const int* __restrict__ head,
                                                          const EdgeBlockHeader* __restrict__ tab, int n) {
    const int ng = tab->n_groups, ns = tab->n_stitch;
    const int c = threadIdx.x & 7;
    for (int k = (blockIdx.x * 256 + threadIdx.x) >> 3; k < ns; k += (gridDim.x * 256) >> 3) {
        const int v = stitch_list[k];
        if (v < 0 || v >= n) continue;
        const int g0 = stitch[v];
        if (g0 < 0 || g0 >= ng || head[g0] != v) continue;
        floatx4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<floatx4*>(agg + (size_t)v * H + 16 * c + 4 * q);
        int g = g0;
        do {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += *reinterpret_cast<const floatx4*>(side + (size_t)g * H + 16 * c + 4 * q);
            ++g;
        } while (g < ng && head[g] == v);
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<floatx4*>(agg + (size_t)v * H + 16 * c + 4 * q) = a[q];
    }
}

// ------------------------------------------------------------------------------------------
// weight image: [T1, 1/T3, 0, 0 | b2 T2, b3 T3, gamma, beta | fp16 hi / lo fragments of t_l W_l]
// One workgroup per processor step.  t_l: power of two from the Linear's gain (hmlp.h).
// ------------------------------------------------------------------------------------------
struct PackH3Jobs {
    int n;
    const float* W1[kPackH3Max];   // [H][3H]: the e block starts at column c1[]
    int c1[kPackH3Max];
    const float* W2[kPackH3Max];
    const float* W3[kPackH3Max];
    const float* b1[kPackH3Max];
    const float* b2[kPackH3Max];
    const float* b3[kPackH3Max];
    const float* gamma[kPackH3Max];
    const float* beta[kPackH3Max];
    float* dst[kPackH3Max];
    int enc_k1[kPackH3Max];
    int ld1[kPackH3Max];
};

constexpr int H3_PACK_THREADS = 1024;
__global__ void __launch_bounds__(H3_PACK_THREADS) pack_h3_kernel(PackH3Jobs J) {
    __shared__ float red[H3_PACK_THREADS];
    __shared__ float tsc[4];
    __shared__ float cmean[H + 1];   // Linear 3: mean of every column of W3 over the output features, then the mean of b3
    const int job = blockIdx.x, tid = threadIdx.x;
    const float* Wl[3] = {J.W1[job], J.W2[job], J.W3[job]};
    // LayerNorm subtracts the mean over the output features of Linear 3; with W3' = W3 - 1 mean_rows(W3), b3' = b3 - mean(b3)
    // the outputs are z - mean(z) exactly, so the kernel needs no mean: variance = mean square, x_hat = z' / sigma.
    if (tid < H) {
        float a = 0.f;
        for (int o = 0; o < H; ++o) a += J.W3[job][(size_t)o * H + tid];
        cmean[tid] = a * (1.0f / H);
    } else if (tid == H) {
        float a = 0.f;
        for (int o = 0; o < H; ++o) a += J.b3[job][o];
        cmean[H] = a * (1.0f / H);
    }
    __syncthreads();
    const int k1 = J.enc_k1[job];          // > 0: encoder image (Linear 1 takes k1 raw features)
    const bool enc = k1 > 0;
    const int ld[3] = {enc ? k1 : (J.ld1[job] > 0 ? J.ld1[job] : 3 * H), H, H}, c0[3] = {J.c1[job], 0, 0};
    // Scales as in the streamed kernels (hmlp.h / pack_hm_kernel): m = estimated rms of the activations, a ReLU layer maps
    // m^2 -> gain^2 m^2 + rms(b)^2 / 2 with gain = ||W_l||_F / sqrt(out) / sqrt(2); U_l = power of two nearest kHmTargetRms / m_l,
    // t_l = U_l / U_(l-1).  Linear 1's pre-activation takes h_i and h_j too: its gain is that of the whole [H x 3H] matrix; its
    // inputs (h, e) are at their natural magnitude (m_0 = 1).
    const float* bl[3] = {J.b1[job], J.b2[job], J.b3[job]};
    // Encoder: the rows enter scaled by their own power of two (maximum in [2^6, 2^7): rms ~ kHmRawInputRms), the biases ride
    // on that scale and are kept in range by the cap (pack_hm_kernel's rule), so they stay out of the estimate.
    float m_est = enc ? kHmRawInputRms : 1.f, U_prev = 1.f;
    if (tid == 0) tsc[3] = 0.f;
    __shared__ float capv;
    if (tid == 0) capv = 3.0e38f;
    for (int l = 0; l < 3; ++l) {
        float ss = 0.f, wm = 0.f;
        const int cols = l == 0 ? ld[0] : H;
        for (int i = tid; i < H * cols; i += H3_PACK_THREADS) {
            const int col = i % cols;
            float v = Wl[l][(size_t)(i / cols) * ld[l] + (l == 0 ? 0 : c0[l]) + col];
            if (l == 2) v -= cmean[col];
            ss = fmaf(v, v, ss);
            if (l != 0 || enc || (col >= c0[0] && col < c0[0] + H)) wm = fmaxf(wm, fabsf(v));   // the packed block
        }
        red[tid] = ss;
        __syncthreads();
        for (int s = H3_PACK_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const float ss_all = red[0];
        __syncthreads();
        red[tid] = wm;
        __syncthreads();
        for (int s = H3_PACK_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        if (tid == 0) {
            const float wmax = red[0];
            float bs = 0.f, bmax = 0.f;
            for (int o = 0; o < H; ++o) {
                const float bv = bl[l][o] - (l == 2 ? cmean[H] : 0.f);
                bs = fmaf(bv, bv, bs);
                bmax = fmaxf(bmax, fabsf(bv));
            }
            const float g2 = ss_all / (float)H * 0.5f;
            float m = sqrtf(fmaf(enc ? 0.f : 0.5f, bs / (float)H, g2 * m_est * m_est));
            if (!(m > 1.0e-30f) || !(m < 1.0e30f)) m = 1.f;
            auto pow2 = [](float want, bool nearest) {
                if (!(want > 0.f) || !(want < 3.0e38f)) return 1.f;
                int ex;
                const float f = frexpf(want, &ex);
                int sh = (nearest && f >= 0.70710678f) ? ex : ex - 1;
                sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
                return ldexpf(1.f, sh);
            };
            float t = pow2(kHmTargetRms / m, true) / U_prev;
            if (wmax > 0.f && wmax < 3.0e38f) {   // packed weights where both halves of the split are normal
                if (wmax * t > 8192.0f) t = pow2(8192.0f / wmax, false);
                if (wmax * t < 0.0625f) t = 2.f * pow2(0.0625f / wmax, false);
            }
            tsc[l] = t;
            tsc[3] = m;
            if (enc && bmax > 0.f) capv = fminf(capv, pow2(4096.0f / (U_prev * t * bmax), false));   // no row scale may push a bias out of range
        }
        __syncthreads();
        m_est = tsc[3];
        U_prev *= tsc[l];
        __syncthreads();
    }
    const float T1 = tsc[0], T2 = T1 * tsc[1], T3 = T2 * tsc[2];
    float* dst = J.dst[job];
    if (tid == 0) { dst[0] = T1; dst[1] = 1.0f / T3; dst[2] = capv; dst[3] = 0.f; }
    float* vec = dst + HW_HEADER_FLOATS;
    for (int i = tid; i < H; i += H3_PACK_THREADS) {
        vec[4 * H + i] = J.b1[job][i] * T1;
        vec[i] = J.b2[job][i] * T2;
        vec[H + i] = (J.b3[job][i] - cmean[H]) * T3;
        vec[2 * H + i] = J.gamma[job][i];
        vec[3 * H + i] = J.beta[job][i];
    }
    _Float16* img = reinterpret_cast<_Float16*>(dst + HW_HEADER_FLOATS + HW_VEC_FLOATS);
    // element (l, w, ks, lane = (i, kg), j): W_l[32 w + i][16 ks + 8 (j >> 2) + 4 kg + (j & 3)]
    for (int idx = tid; idx < 3 * 4 * 8 * 64 * 8; idx += H3_PACK_THREADS) {
        const int j = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 7, w = (idx >> 12) & 3, l = idx >> 14;
        const int i = lane & 31, kg = lane >> 5;
        const int kcol = 16 * ks + 8 * (j >> 2) + 4 * kg + (j & 3);
        const bool pad = l == 0 && enc && kcol >= k1;   // the encoder's first Linear: zero beyond its k1 inputs
        const float v = pad ? 0.f : (Wl[l][(size_t)(32 * w + i) * ld[l] + c0[l] + kcol] - (l == 2 ? cmean[kcol] : 0.f)) * tsc[l];
        const _Float16 h = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)h);
        const size_t base = ((((size_t)(l * 4 + w) * 8 + ks) * 2) * 64 + lane) * 8 + j;
        img[base] = h;
        img[base + 64 * 8] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// block / chunk tables of a destination-sorted edge list that holds one or more equal-sized graphs back to back
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) edge_blocks_plan_kernel(const int* __restrict__ in_ptr, int n_nodes, const int* n_per_dev, int n_per_host,
                                                               EdgeBlockHeader* tab, int* gblk, int max_graphs) {
    if (threadIdx.x == 0) edge_blocks_plan(in_ptr, n_nodes, n_per_dev, n_per_host, tab, gblk, max_graphs);
}

__global__ void __launch_bounds__(256) edge_blocks_fill_kernel(const int* __restrict__ in_ptr, const int* __restrict__ dst, int n_nodes,
                                                               EdgeBlockHeader* tab, const int* __restrict__ gblk, int2* blk, int2* seg,
                                                               int* __restrict__ head, int* __restrict__ stitch, int* __restrict__ stitch_list) {
    edge_blocks_fill(in_ptr, dst, n_nodes, tab, gblk, blk, seg, head, stitch, stitch_list, blockIdx.x * blockDim.x + threadIdx.x,
                     gridDim.x * blockDim.x);
}

}  // namespace

size_t h3_image_floats() { return (size_t)HW_HEADER_FLOATS + HW_VEC_FLOATS + (size_t)HW_IMAGE_HALF8 * 4; }

int pack_h3(const PackH3Job* jobs, int n, hipStream_t s) {
    for (int off = 0; off < n; off += kPackH3Max) {
        PackH3Jobs J{};
        J.n = n - off < kPackH3Max ? n - off : kPackH3Max;
        for (int i = 0; i < J.n; ++i) {
            const PackH3Job& j = jobs[off + i];
            J.W1[i] = j.W1; J.c1[i] = j.W1_col0; J.W2[i] = j.W2; J.W3[i] = j.W3; J.b1[i] = j.b1; J.b2[i] = j.b2; J.b3[i] = j.b3; J.gamma[i] = j.gamma; J.beta[i] = j.beta;
            J.dst[i] = j.dst;
            J.enc_k1[i] = j.enc_k1;
            J.ld1[i] = j.W1_ld;
        }
        hipLaunchKernelGGL(pack_h3_kernel, dim3(J.n), dim3(H3_PACK_THREADS), 0, s, J);
        GM_LAUNCH_CHECK();
    }
    return GM_OK;
}

static size_t max_blocks_of(int64_t n_nodes, int64_t edge_capacity) {
    return (size_t)cdiv(edge_capacity, BE) + 4 * (size_t)n_nodes + 4;   // each graph: up to 3 padding blocks + 1 partial
}
// rows of the side buffer: one per group, then kSinkRows rows that take the masked-off lanes of the systolic kernel's
// (unconditional) row stores, one per workgroup
size_t edge_groups_max(int64_t n_nodes, int64_t edge_capacity) { return max_blocks_of(n_nodes, edge_capacity) / 4 + 1 + kSinkRows; }

size_t edge_blocks_ints(int64_t n_nodes, int64_t edge_capacity) {
    const size_t nblk = max_blocks_of(n_nodes, edge_capacity);
    // header | gblk[n+2] | blk[nblk] (int2) | seg[nblk] (int2) | head[nblk/4 + 2] | stitch[n] | stitch_list[n]
    return 8 + ((size_t)n_nodes + 2) + 1 + 4 * nblk + (nblk / 4 + 2) + 2 * (size_t)n_nodes;
}

EdgeBlocks carve_edge_blocks(int* base, int64_t n_nodes, int64_t edge_capacity) {
    EdgeBlocks t;
    const size_t nblk = max_blocks_of(n_nodes, edge_capacity);
    t.hdr = reinterpret_cast<EdgeBlockHeader*>(base);
    t.gblk = base + 8;
    int* p = t.gblk + n_nodes + 2;
    p += (reinterpret_cast<uintptr_t>(p) & 4) ? 1 : 0;   // int2 alignment
    t.blk = reinterpret_cast<int2*>(p);
    t.seg = reinterpret_cast<int2*>(p + 2 * nblk);
    t.head = p + 4 * nblk;
    t.stitch = t.head + nblk / 4 + 2;
    t.stitch_list = t.stitch + n_nodes;
    t.max_blocks = (int64_t)nblk;
    return t;
}

static int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}

int build_edge_blocks(const int* in_ptr, const int* dst, int64_t n_nodes, int64_t edge_capacity, const int* n_per_graph_dev,
                      int n_per_graph_host, const EdgeBlocks& t, hipStream_t s) {
    // t.stitch is -1 everywhere: the destination sort's clear pass set it (csr_clear_jobs)
    hipLaunchKernelGGL(edge_blocks_plan_kernel, dim3(1), dim3(64), 0, s, in_ptr, (int)n_nodes, n_per_graph_dev, n_per_graph_host,
                       t.hdr, t.gblk, (int)n_nodes + 1);
    int gb = (int)cdiv(t.max_blocks, 256);
    gb = gb < 1 ? 1 : (gb > 1024 ? 1024 : gb);
    hipLaunchKernelGGL(edge_blocks_fill_kernel, dim3(gb), dim3(256), 0, s, in_ptr, dst, (int)n_nodes, t.hdr, t.gblk, t.blk, t.seg, t.head, t.stitch,
                       t.stitch_list);
    GM_LAUNCH_CHECK();
    (void)edge_capacity;
    return GM_OK;
}

bool edge_sys_fits(int64_t n_nodes, int64_t edge_capacity) {
    // byte offsets of the P gathers and of the agg / side-buffer stores are 32-bit (buffer addressing)
    const uint64_t agg_rows = (uint64_t)n_nodes + edge_groups_max(n_nodes, edge_capacity);
    return (uint64_t)n_nodes * 2 * H * 4 < (1ull << 32) && agg_rows * H * 4 < (1ull << 32);
}

#ifndef HEDGE_STREAM_MB
#define HEDGE_STREAM_MB 128   // A/B builds move it; 0 = always stream (round 5's policy), a huge value = never
#endif
constexpr uint64_t kStreamStoreBytes = (uint64_t)HEDGE_STREAM_MB << 20;
int launch_edge_sys(const EdgeArgs& a, const EdgeBlocks& t, int64_t edge_capacity, hipStream_t s) {
    GM_REQUIRE(a.hdr && a.wstream_h3 && a.agg && a.side && !a.eid && !a.eid_out, GM_ERR_INVALID_ARGUMENT, "launch_edge_sys: unsupported argument combination");
    GM_REQUIRE(a.P_prescaled, GM_ERR_INVALID_ARGUMENT, "launch_edge_sys: P must carry the weight scale of this step (NodeArgs::p_scale = edge_sys_p_scale(image))");
    // the scatter-add addresses agg rows and side rows with 32-bit byte offsets from agg (carve_fwd puts them in one workspace)
    const uint64_t agg_bytes = ((uint64_t)(a.side - a.agg) + (uint64_t)(t.max_blocks / 4 + 1) * H) * 4;
    GM_REQUIRE(a.side >= a.agg && agg_bytes < (1ull << 32) && (uint64_t)a.n_nodes_tab * 2 * H * 4 < (1ull << 32), GM_ERR_INVALID_ARGUMENT,
               "launch_edge_sys: P and agg + side buffer must each stay below 4 GiB (edge_sys_fits)");
    static PerDeviceOnce attr_done;
    const int rc_attr = attr_done.run([]() -> int {
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_edge_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SYS_LDS_BYTES));
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_edge_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SYS_LDS_BYTES));
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_edge_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SYS_LDS_BYTES));
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_edge_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SYS_LDS_BYTES));
        return GM_OK;
    });
    if (rc_attr != GM_OK) return rc_attr;
    {
        ProfScope prof(a.prof, PROF_EDGE, s);
        // pointers as separate __restrict__ parameters (e_in / e_out may be the same array): the table reads are then provably
        // unclobbered and become scalar loads
        // Store policy by size (ST_STREAM_*): the edge rows of the launch's capacity against what the 256 MB Infinity Cache can keep from
        // one launch to the next beside h, P and agg.  The capacity, not the device-side edge count: no host synchronisation.
        const bool stream = (uint64_t)edge_capacity * H * 4 > kStreamStoreBytes;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(device_cus()), dim3(SYS_THREADS), SYS_LDS_BYTES, s, a.hdr, a.dst, a.src, a.P, a.e_in, a.e_out,
                               a.agg, a.wstream_h3, t.blk, t.seg, t.head, t.hdr, (unsigned)(a.side - a.agg), (unsigned)agg_bytes,
                               (unsigned)((uint64_t)a.n_nodes_tab * 2 * H * 4), const_cast<int*>(&a.hdr->error_flags), a.eps, a.residual);
        };
        if (a.discard_e_out) { if (stream) go(sys_edge_kernel<false, true>); else go(sys_edge_kernel<false, false>); }
        else { if (stream) go(sys_edge_kernel<true, true>); else go(sys_edge_kernel<true, false>); }
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_edge_sys_enc(const EdgeArgs& a, hipStream_t s) {
    GM_REQUIRE(a.hdr && a.wstream_h3 && a.e_in && a.e_out && !a.eid && !a.eid_out && a.k1 == 4, GM_ERR_INVALID_ARGUMENT,
               "launch_edge_sys_enc: unsupported argument combination");
    static PerDeviceOnce attr_done;
    const int rc_attr = attr_done.run([]() -> int {
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_enc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ENC_LDS_BYTES));
        return GM_OK;
    });
    if (rc_attr != GM_OK) return rc_attr;
    {
        ProfScope prof(a.prof, PROF_ENC, s);
        hipLaunchKernelGGL(sys_enc_kernel, dim3(device_cus()), dim3(SYS_THREADS), ENC_LDS_BYTES, s, a.hdr, a.e_in, a.e_out, a.wstream_h3,
                           const_cast<int*>(&a.hdr->error_flags), a.eps, a.zero_pad_rows ? 1 : 0);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_node_sys(const NodeSysArgs& a, hipStream_t s) {
    GM_REQUIRE(a.h && a.agg && a.Q && a.h_out && a.image && a.n > 0, GM_ERR_INVALID_ARGUMENT, "launch_node_sys: bad argument");
    static PerDeviceOnce attr_done;
    const int rc_attr = attr_done.run([]() -> int {
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_node_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NODE_LDS_BYTES));
        return GM_OK;
    });
    if (rc_attr != GM_OK) return rc_attr;
    {
        ProfScope prof(a.prof, PROF_NODE, s);
        hipLaunchKernelGGL(sys_node_kernel, dim3(device_cus()), dim3(SYS_THREADS), NODE_LDS_BYTES, s, a.h, a.agg, a.Q, a.h_out, a.image, a.n, a.flags, a.eps);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_proj_sys(const ProjSysArgs& a, hipStream_t s) {
    GM_REQUIRE(a.h && a.P && a.Q && a.img_p && a.img_q && a.n > 0, GM_ERR_INVALID_ARGUMENT, "launch_proj_sys: bad argument");
    static PerDeviceOnce attr_done;
    const int rc_attr = attr_done.run([]() -> int {
        GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sys_proj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PROJ_LDS_BYTES));
        return GM_OK;
    });
    if (rc_attr != GM_OK) return rc_attr;
    {
        ProfScope prof(a.prof, PROF_NODE, s);
        hipLaunchKernelGGL(sys_proj_kernel, dim3(device_cus()), dim3(SYS_THREADS), PROJ_LDS_BYTES, s, a.h, a.P, a.Q, a.img_p, a.img_q, a.scale_p, a.scale_q, a.n, a.flags);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

int launch_agg_stitch(float* agg, const float* side, const EdgeBlocks& t, int64_t n, ProfState* prof_state, hipStream_t s) {
    if (n <= 0) return GM_OK;
    {
        ProfScope prof(prof_state, PROF_NODE, s);
        // sized for the list's worst case (every node) up to two rounds of the chip's workgroup slots; a dense scene's list is n / 7
        int64_t gb = cdiv(n * 8, 256 * 4);
        gb = gb < 1 ? 1 : (gb > 2048 ? 2048 : gb);
        hipLaunchKernelGGL(agg_stitch_kernel, dim3((unsigned)gb), dim3(256), 0, s, agg, side, t.stitch, t.stitch_list, t.head, t.hdr, (int)n);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// The systolic kernel reads whole 32-row blocks: the (up to 31) rows behind the last edge of the list must hold finite values
// (a NaN there would raise the range flag).  One tiny launch per forward keeps kEdgePadRows zero rows behind row n_edges.
__global__ void __launch_bounds__(256) edge_pad_rows_kernel(const CsrHeader* hdr, float* e, int row_floats) {
    const size_t base = (size_t)hdr->n_edges * row_floats;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kEdgePadRows * row_floats; i += gridDim.x * blockDim.x) e[base + i] = 0.f;
}
int zero_edge_pad_rows(const CsrHeader* hdr, float* e, int row_floats, hipStream_t s) {
    hipLaunchKernelGGL(edge_pad_rows_kernel, dim3(4), dim3(256), 0, s, hdr, e, row_floats);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

#ifdef HEDGE_STAMPS
extern "C" int gm_debug_sys_stamps(unsigned long long* out) {   // 3 roles x 32 ticks x 8 slots + 32 real-time stamps, development builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sys_stamps), sizeof(unsigned long long) * (3 * 32 * 8 + 32)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace gm
