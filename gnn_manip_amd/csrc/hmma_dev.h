// Device helpers shared by the fp16-split matrix-pipe kernels (hedge.hip, hmlp.hip).
//
// Arithmetic.  Every fp32 operand is split into two fp16 parts, x = hi + lo (22 significant bits; the low parts may be
// fp16 subnormals, which v_mfma_f32_32x32x16_f16 honours -- checked on gfx950).  A product of two fp16 values is exact in
// fp32, so  lo*hi + hi*lo + hi*hi  accumulated in fp32 reproduces the fp32 product to 2^-22.
//
// Domain.  hi is a normal fp16 number for 2^-14 <= |x| < 65504 and lo keeps all of its 11 bits for |x| >= 2^-3; below that
// the pair carries an absolute error of 2^-25 (lo's subnormal spacing).  The kernels therefore keep the operand images at
// power-of-two scales chosen from the weights (hmlp.h: rms near 2^4, so everything within 2^-7 .. 2^12 of the rms is exact to
// 22 bits) and scale raw feature rows by their own maximum; |x| >= 65504 in an image is reported (ERRF_SPLIT_RANGE), never
// clamped.  What is NOT covered: a row whose h / e / agg entries (LayerNorm outputs and their sums, written at their natural
// magnitude) are all below ~2^-8 in magnitude loses relative precision, where the fp32 reference would not.
//
// Operand layout.  v_mfma_f32_32x32x16_f16 with A = 32 weight rows (output features) and B = 32 rows of the tile (edges or
// nodes): lane (n, kg) = (lane & 31, lane >> 5) holds 8 halves of K.  K slot (kg, j) of k-group ks carries input feature
//     16 ks + 8 (j >> 2) + 4 kg + (j & 3)
// in BOTH operands, which makes the accumulator registers 8q..8q+7 of the wave that owns output block jb exactly the
// 8 elements of the next Linear's B fragment ks = 2 jb + q on the same lane: activations go accumulator -> ReLU -> split
// -> LDS image without any cross-lane movement.  Accumulator register r of lane (n, hi) is output feature
//     32 jb + 8 (r >> 2) + 4 hi + (r & 3)   of row n.
#pragma once
#include <hip/hip_runtime.h>

namespace gm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { unsigned u; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b)); return u; }
// ReLU as ONE instruction: fmaxf (and the med3 builtin) cost two -- hipcc canonicalises the operand first.  hipcc pads the MFMA ->
// VALU read hazard only for instructions it knows, so an inline-asm reader must not be the FIRST reader of MFMA results: every
// caller reads the accumulators with compiler-visible code first (the range check of the rows) and has a barrier -- a workgroup
// barrier or a scheduling barrier -- between that and this.
// No clamp: a value beyond the fp16 range is not saturated silently.  Every value written to an operand image is range-checked
// by its kernel instead (ERRF_SPLIT_RANGE, common.h).
__device__ __forceinline__ float relu(float x) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x)); return r; }

// low part of a pair straight to fp16: x - hi is exact in fp32, so one rounding either way (v_fma_mixlo / mixhi write one half
// of the destination and keep the other)
__device__ __forceinline__ unsigned lo_pair(unsigned h, float a, float b) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r) : "v"(h), "v"(a), "v"(b));
    return r;
}

// two-way fp16 split of 4 floats: hi / lo as two dwords each (elements in order)
__device__ __forceinline__ void split4(float a, float b, float c, float d, uintx2& hi, uintx2& lo) {
    hi[0] = cvt_pk(a, b);
    hi[1] = cvt_pk(c, d);
    lo[0] = lo_pair(hi[0], a, b);
    lo[1] = lo_pair(hi[1], c, d);
}

// x += lanes(x shifted) * f.  The value is produced by compiler code just before: the VALU -> DPP hazard (2 wait states) of
// the first reader is padded by hand, hipcc pads nothing inside asm.
#define DPP_FMAC(x, f, ctrl) asm volatile("v_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(x) : "v"(f))
#define DPP_FMAC_NOP(x, f, ctrl) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(x) : "v"(f))

}  // namespace
}  // namespace gm
