// Operand dtype of every GEMM / attention operand buffer (`bf16_t` = raw 16 bits) — one switch for the whole library:
//   default      bf16: the dtype of the V3-class trainers' autocast (train_ddp_v3m2.py:545) and of the sampler
//   -DJAT_FP16   fp16: `torch.amp.autocast('cuda')` of the v3mod2 trainer (train_ddp_v3mod2.py:854) with its GradScaler
//                (:745); built as a second library, libjat_hip_fp16.so, selected by JAT_OPERAND_DTYPE=fp16
// Same MFMA shape (v_mfma_f32_16x16x32_{bf16,f16}), same fragment layouts, same 16-bit transposed LDS reads; what differs
// is the conversion at every store / unpack site, gathered here.  fp32 -> fp16 saturates to inf above 65504: that is
// what the dynamic loss scale reacts to.
#pragma once
#include <hip/hip_runtime.h>

#ifdef JAT_FP16
typedef _Float16 jat_op_t;
#define JAT_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define JAT_OPERAND_DTYPE 1
__device__ __forceinline__ unsigned short jat_f2op(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float jat_op2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ float jat_lo2f(unsigned u) { return jat_op2f((unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float jat_hi2f(unsigned u) { return jat_op2f((unsigned short)(u >> 16)); }
#else
typedef __bf16 jat_op_t;
#define JAT_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define JAT_OPERAND_DTYPE 0
__device__ __forceinline__ unsigned short jat_f2op(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
}
__device__ __forceinline__ float jat_op2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ float jat_lo2f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float jat_hi2f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
#endif
typedef __attribute__((ext_vector_type(8))) jat_op_t jat_opx8;
// two fp32 -> one packed pair of operands (lo in bits 0..15).  fp16: v_cvt_pk_f16_f32 (round-to-nearest-even, one instruction
// instead of two conversions and an OR); bf16: v_cvt_pk_bf16_f32 the same way.
#ifdef JAT_FP16
__device__ __forceinline__ unsigned jat_pack2(float lo, float hi) {
  typedef float jat_f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 jat_f16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(jat_f32x2{lo, hi}, jat_f16x2));
}
#else
__device__ __forceinline__ unsigned jat_pack2(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (the scalar form costs two + fix-ups)
  typedef float jat_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 jat_bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(jat_f32x2{lo, hi}, jat_bf16x2));
}
#endif
// eight fp32 (two accumulator quads) -> one MFMA operand fragment
typedef float jat_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ jat_opx8 jat_pack8(const jat_f32x4 a, const jat_f32x4 b) {
#ifdef JAT_FP16
  typedef unsigned jat_u32x4 __attribute__((ext_vector_type(4)));
  const jat_u32x4 u = {jat_pack2(a[0], a[1]), jat_pack2(a[2], a[3]), jat_pack2(b[0], b[1]), jat_pack2(b[2], b[3])};
  return __builtin_bit_cast(jat_opx8, u);
#else
  jat_opx8 f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f[r] = (jat_op_t)a[r];
    f[4 + r] = (jat_op_t)b[r];
  }
  return f;
#endif
}
