// Packed fp32 arithmetic for the Winograd transforms (device code only; included by conv_wino.hip and conv_wino_wgrad.hip).
#pragma once

namespace w2l {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Packed fp32: one v_pk_add/mul/fma_f32 does two floats per lane and costs the same issue time as one scalar VALU instruction
// next to an MFMA stream (profiles/r01/i_mfma_overlap_microbench.txt: VALU work does not hide behind fp32 MFMAs when a wave is
// alone on its SIMD, and a packed instruction costs what a scalar one does).  The compiler splits packed fp32 operations that
// follow an MFMA back into scalar ones, so the transforms issue them through inline assembly.
#ifndef W2L_PK_NOASM
__device__ __forceinline__ f32x2 pk2_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk2_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk2_half(f32x2 a) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ f32x2 pk2_mul_s(f32x2 a, f32x2 s) {              // s: wave-uniform splat in an SGPR pair
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(s));
    return d;
}
__device__ __forceinline__ f32x2 pk2_fma_s(f32x2 a, f32x2 s, f32x2 c) {    // a * s + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}
#else
__device__ __forceinline__ f32x2 pk2_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk2_sub(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 pk2_half(f32x2 a) { return a * 0.5f; }
__device__ __forceinline__ f32x2 pk2_mul_s(f32x2 a, f32x2 s) { return a * s; }
__device__ __forceinline__ f32x2 pk2_fma_s(f32x2 a, f32x2 s, f32x2 c) { return __builtin_elementwise_fma(a, s, c); }
#endif
__device__ __forceinline__ f32x4 cat(f32x2 l, f32x2 h) { return __builtin_shufflevector(l, h, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b) { return cat(pk2_add(a.lo, b.lo), pk2_add(a.hi, b.hi)); }
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) { return cat(pk2_sub(a.lo, b.lo), pk2_sub(a.hi, b.hi)); }
__device__ __forceinline__ f32x4 pk_half(f32x4 a) { return cat(pk2_half(a.lo), pk2_half(a.hi)); }
__device__ __forceinline__ f32x4 pk_mul(f32x4 a, f32x2 s) { return cat(pk2_mul_s(a.lo, s), pk2_mul_s(a.hi, s)); }
__device__ __forceinline__ f32x4 pk_fma(f32x4 a, f32x2 s, f32x4 c) { return cat(pk2_fma_s(a.lo, s, c.lo), pk2_fma_s(a.hi, s, c.hi)); }

}  // namespace w2l
