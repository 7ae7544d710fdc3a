// GELU of the GEMM epilogues and of the split-K finishing pass (gemm.hip, elementwise.hip): ONE definition, so that a Linear + GELU
// gives the same bits whether its K loop ran in one block or in slices.
#pragma once
#include <hip/hip_runtime.h>

// GELU (erf form, nn.GELU() default: jat_audiosr_v3.py:223,268) as x * Phi(x), Phi(x) = clamp01(1/2 + x * Q(s)), s = clamp01(x^2 / 4.5^2),
// Q a degree-8 polynomial (weighted minimax fit of (Phi(x) - 1/2) / x on |x| <= 4.5, monomial basis in s; beyond 4.5 the two
// clamps saturate Phi at 0 / 1: Phi(-4.5) = 3.4e-6).  |gelu_fast - gelu| <= 6e-5 for all x in fp32 (tests/test_host_cpu.py
// evaluates the same expression in numpy against scipy's erf), i.e. 1/30 of the half-ulp of the bf16 the result is rounded to at
// |gelu| ~ 1.  No transcendental; the two clamps are the free output modifier of v_fma_f32 (hence scalar fmas there), everything
// else packed fp32.  The fc1 epilogue is VALU-bound (140 values per lane per 224 x 320 tile), so what counts is (a) the issue
// count: 13 fp32 multiply-add slots per element, and (b) that the chains of SEVERAL pairs are interleaved: one pair at a
// time, every Horner step waits for the previous one (the compiler serialised the round-2 form: profiles/r03).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void gelu_erf_n(f32x2 (&x)[N]) {
#define JAT_C2(v) f32x2{v, v}
  f32x2 s[N], q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f32x2 xs = x[i] * JAT_C2(0.22222222f);
    s[i] = f32x2{__builtin_amdgcn_fmed3f(__builtin_fmaf(xs[0], xs[0], 0.f), 0.f, 1.f),
                 __builtin_amdgcn_fmed3f(__builtin_fmaf(xs[1], xs[1], 0.f), 0.f, 1.f)};
  }
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(JAT_C2(0.858849732f), s[i], JAT_C2(-4.62950545f));
  constexpr float cs[7] = {10.9725412f, -15.1604596f, 13.6928242f, -8.61624417f, 3.9305869f, -1.33619357f, 0.398712717f};
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], JAT_C2(cs[k]));
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f32x2 phi = f32x2{__builtin_amdgcn_fmed3f(__builtin_fmaf(x[i][0], q[i][0], 0.5f), 0.f, 1.f),
                            __builtin_amdgcn_fmed3f(__builtin_fmaf(x[i][1], q[i][1], 0.5f), 0.f, 1.f)};
    x[i] = x[i] * phi;
  }
#undef JAT_C2
}
