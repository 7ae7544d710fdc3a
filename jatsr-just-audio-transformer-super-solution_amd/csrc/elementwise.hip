// HBM-bound row-wise and elementwise kernels of the DiT sampling path (gfx950).
// All of them stream 16 B per lane (cdna_hip_programming.md Guideline 13) and reduce with 64-wide
// wavefront shuffles; none is GEMM-shaped.
#include "jat_kernels.h"
#include "jat_gelu.h"
#include "jat_dtype.h"
#include <cstdlib>

__device__ __forceinline__ unsigned short f2bf_e(float f) { return jat_f2op(f); }
__device__ __forceinline__ uint2 pack4_e(float a, float b, float c, float d) {
  uint2 r;
  r.x = jat_pack2(a, b);
  r.y = jat_pack2(c, d);
  return r;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- fused norm + adaLN modulate -> bf16 ----------------------------------------------------------
// y = norm(x) * w * (1 + scale[b]) + shift[b]     (jat_audiosr_v3.py:297-298, 303-304; final norm :384)
// mode 0: RMSNorm(eps 1e-6, weight) ; mode 1: LayerNorm(eps 1e-6, no affine) ; mode 2: plain cast.
// One wave per row; the row (D <= 2048 floats) lives in registers between the statistics and the write.
__global__ void __launch_bounds__(256) norm_modulate_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ scale, int64_t mod_bstride,
                                                            bf16_t* __restrict__ y, int M, int D, int ntok, int mode) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = D >> 8;  // 256 floats per chunk (64 lanes x float4)
  const float* xr = x + (int64_t)row * D + lane * 4;
  float4 v[8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < nch) {
      v[c] = *(const float4*)(xr + c * 256);
      s1 += v[c].x + v[c].y + v[c].z + v[c].w;
      s2 += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
    }
  }
  float mu = 0.f, rstd = 1.f;
  if (mode == 0) {
    s2 = wave_sum(s2);
    rstd = rsqrtf(s2 / (float)D + 1e-6f);
  } else if (mode == 1) {
    s1 = wave_sum(s1);
    mu = s1 / (float)D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nch) {
        const float a = v[c].x - mu, b2 = v[c].y - mu, c2 = v[c].z - mu, d2 = v[c].w - mu;
        var += a * a + b2 * b2 + c2 * c2 + d2 * d2;
      }
    var = wave_sum(var);
    rstd = rsqrtf(var / (float)D + 1e-6f);
  }
  const int b = row / ntok;
  const float* sh = shift ? shift + (int64_t)b * mod_bstride + lane * 4 : nullptr;
  const float* sc = scale ? scale + (int64_t)b * mod_bstride + lane * 4 : nullptr;
  bf16_t* yr = y + (int64_t)row * D + lane * 4;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < nch) {
      float4 t = v[c];
      t.x = (t.x - mu) * rstd; t.y = (t.y - mu) * rstd; t.z = (t.z - mu) * rstd; t.w = (t.w - mu) * rstd;
      if (mode == 0 && w) {
        const float4 ww = *(const float4*)(w + c * 256 + lane * 4);
        t.x *= ww.x; t.y *= ww.y; t.z *= ww.z; t.w *= ww.w;
      }
      if (sc) {
        const float4 a = *(const float4*)(sc + c * 256);
        const float4 s = *(const float4*)(sh + c * 256);
        t.x = t.x * (1.f + a.x) + s.x; t.y = t.y * (1.f + a.y) + s.y;
        t.z = t.z * (1.f + a.z) + s.z; t.w = t.w * (1.f + a.w) + s.w;
      }
      *(uint2*)(yr + c * 256) = pack4_e(t.x, t.y, t.z, t.w);
    }
  }
}

// Persistent variant for the common widths (NCH = D/256 known at compile time): each wave walks rows
// wave, wave + nwaves, ... and issues the NEXT row's loads before it reduces the current one, so the load
// latency of a row hides under the previous row's shuffle reduction and stores.
typedef __attribute__((ext_vector_type(4))) float f32x4_e;
template <int NCH>
__global__ void __launch_bounds__(256) norm_modulate_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ scale, int64_t mod_bstride,
                                                                 bf16_t* __restrict__ y, int M, int ntok, int mode) {
  constexpr int D = NCH * 256;
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4_e ww[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    ww[c] = (mode == 0 && w) ? *(const f32x4_e*)(w + c * 256 + lane * 4) : f32x4_e{1.f, 1.f, 1.f, 1.f};
  f32x4_e cur[NCH], nxt[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) cur[c] = *(const f32x4_e*)(x + (int64_t)row * D + c * 256 + lane * 4);
  while (true) {
    const int next = row + nwaves;
    if (next < M) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) nxt[c] = *(const f32x4_e*)(x + (int64_t)next * D + c * 256 + lane * 4);
    }
    // this row's modulation does not depend on its statistics: request it before the shuffle reduction, not after
    const int b = row / ntok;
    const float* sh = shift ? shift + (int64_t)b * mod_bstride + lane * 4 : nullptr;
    const float* sc = scale ? scale + (int64_t)b * mod_bstride + lane * 4 : nullptr;
    f32x4_e ma[NCH], ms[NCH];
    if (sc) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) { ma[c] = *(const f32x4_e*)(sc + c * 256); ms[c] = *(const f32x4_e*)(sh + c * 256); }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      s1 += cur[c][0] + cur[c][1] + cur[c][2] + cur[c][3];
      s2 += cur[c][0] * cur[c][0] + cur[c][1] * cur[c][1] + cur[c][2] * cur[c][2] + cur[c][3] * cur[c][3];
    }
    float mu = 0.f, rstd = 1.f;
    if (mode == 0) {
      rstd = rsqrtf(wave_sum(s2) / (float)D + 1e-6f);
    } else if (mode == 1) {
      mu = wave_sum(s1) / (float)D;
      float var = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) var += (cur[c][e] - mu) * (cur[c][e] - mu);
      rstd = rsqrtf(wave_sum(var) / (float)D + 1e-6f);
    }
    bf16_t* yr = y + (int64_t)row * D + lane * 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4_e t = (cur[c] - mu) * rstd * ww[c];
      if (sc) t = t * (1.f + ma[c]) + ms[c];
      *(uint2*)(yr + c * 256) = pack4_e(t[0], t[1], t[2], t[3]);
    }
    if (next >= M) break;
    row = next;
#pragma unroll
    for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
  }
}

hipError_t launch_norm_modulate(const float* x, const float* w, const float* shift, const float* scale,
                                int64_t mod_bstride, bf16_t* y, int M, int D, int ntok, int mode, hipStream_t s) {
  if (D % 256 != 0 || D > 2048 || M <= 0 || ntok <= 0) return hipErrorInvalidValue;
  if ((shift == nullptr) != (scale == nullptr)) return hipErrorInvalidValue;
  // rows per wave: the persistent form (a wave walks rows w, w + nwaves, .. with the next row's loads in flight) was written for
  // 4 rows per wave; measured, ONE row per wave is faster at every batch size (more waves in flight beat the amortised weight
  // loads): 50-step run at M = 256: 144 -> 129.6 ms, M = 2760: 264.5 -> 255.2 ms, M = 7168 un-folded: 423 -> 418 ms
  static const int rpw_env = getenv("JAT_NORM_RPW") ? atoi(getenv("JAT_NORM_RPW")) : -1;
  const int rows_per_wave = rpw_env >= 0 ? rpw_env : 1;
  if (rows_per_wave > 0 && (D == 1280 || D == 512 || D == 256)) {
    const int blocks = ((M + 3) / 4 + rows_per_wave - 1) / rows_per_wave;
    if (D == 1280)
      hipLaunchKernelGGL(norm_modulate_rows_kernel<5>, dim3(blocks), dim3(256), 0, s, x, w, shift, scale, mod_bstride, y, M, ntok, mode);
    else if (D == 512)
      hipLaunchKernelGGL(norm_modulate_rows_kernel<2>, dim3(blocks), dim3(256), 0, s, x, w, shift, scale, mod_bstride, y, M, ntok, mode);
    else
      hipLaunchKernelGGL(norm_modulate_rows_kernel<1>, dim3(blocks), dim3(256), 0, s, x, w, shift, scale, mod_bstride, y, M, ntok, mode);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(norm_modulate_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w, shift, scale, mod_bstride, y, M,
                     D, ntok, mode);
  return hipGetLastError();
}

// ---- patchify: [B, C, T] fp32 (T contiguous) -> A[(b,tok)][c*4+p] bf16 ----------------------------------
// Restates pad + cat([x_t, x_cond], 1) + reshape/permute (jat_audiosr_v3.py:435-444, 242-244) without
// materialising any of them: reads are 16 B per lane along T, the transpose goes through LDS, writes are
// 256-B runs along the feature dimension.  Tile = 64 tokens x 32 channels.
__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ x_t, const float* __restrict__ x_c,
                                                       bf16_t* __restrict__ A, int B_src, int cond_zero_from, int C_t,
                                                       int C_c, int T_orig, int ntok, const int* __restrict__ tvalid) {
  constexpr int ROW = 264;  // 32 channels x 8 B + 8 B pad: conflict-free ds_write_b64 down a column
  __shared__ __attribute__((aligned(16))) char tile[64 * ROW];
  const int tid = threadIdx.x;
  const int tok0 = blockIdx.x * 64, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int Ktot = (C_t + C_c) * 4;
  const float* src;
  bool zero = false;
  int cl;
  if (c0 < C_t) {
    src = x_t + (int64_t)(b % B_src) * C_t * T_orig;
    cl = c0;
  } else {
    zero = b >= cond_zero_from;
    src = x_c + (int64_t)b * C_c * T_orig;
    cl = c0 - C_t;
  }
  const int tl = tid & 63, tok = tok0 + tl;
  // frames >= tv read as zero: T_orig, or the row's own length when a short chunk is batched with longer ones (the sampler's
  // state z evolves in the padded frames too; a stand-alone run would re-pad with zeros at every step, :435-439)
  const int tv = tvalid ? min(tvalid[b % B_src], T_orig) : T_orig;
  const bool aligned = (T_orig & 3) == 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = (tid >> 6) + 4 * j;
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (!zero && tok < ntok) {
      const float* p = src + (int64_t)(cl + ci) * T_orig + tok * 4;
      const int t0 = tok * 4;
      if (aligned && t0 + 3 < tv) {
        v = *(const float4*)p;
      } else {
        if (t0 + 0 < tv) v.x = p[0];
        if (t0 + 1 < tv) v.y = p[1];
        if (t0 + 2 < tv) v.z = p[2];
        if (t0 + 3 < tv) v.w = p[3];
      }
    }
    *(uint2*)(tile + tl * ROW + ci * 8) = pack4_e(v.x, v.y, v.z, v.w);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int id = tid + 256 * j, r = id >> 5, part = id & 31;
    if (tok0 + r < ntok) {
      const uint2 v = *(const uint2*)(tile + r * ROW + part * 8);
      *(uint2*)(A + ((int64_t)b * ntok + tok0 + r) * Ktot + (int64_t)c0 * 4 + part * 4) = v;
    }
  }
}

hipError_t launch_patchify(const float* x_t, const float* x_cond, bf16_t* A, int B, int B_src, int cond_zero_from,
                           int C_t, int C_c, int T_orig, int ntok, hipStream_t s, const int* tvalid) {
  if (C_t % 32 != 0 || C_c % 32 != 0 || B <= 0 || B_src <= 0) return hipErrorInvalidValue;
  dim3 grid((ntok + 63) / 64, (C_t + C_c) / 32, B);
  hipLaunchKernelGGL(patchify_kernel, grid, dim3(256), 0, s, x_t, x_cond, A, B_src, cond_zero_from, C_t, C_c, T_orig,
                     ntok, tvalid);
  return hipGetLastError();
}

// ---- time embedding sinusoid (jat_audiosr_v3.py:194-207) ---------------------------------------------
__global__ void time_sinusoid_kernel(const float* __restrict__ t, float* __restrict__ e, int B, int D) {
  const int half = D >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float c = logf(10000.0f) / (float)(half - 1);
  const float f = expf((float)k * -c);
  const float a = t[b] * f;
  e[(int64_t)b * D + k] = sinf(a);
  e[(int64_t)b * D + half + k] = cosf(a);
}
hipError_t launch_time_sinusoid(const float* t, float* e, int B, int D, hipStream_t s) {
  const int n = B * (D / 2);
  hipLaunchKernelGGL(time_sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, e, B, D);
  return hipGetLastError();
}

// ---- small fp32 linear (t_embedder, jat_audiosr_v3.py:364-369): one wave per output feature -------------
template <int NCH>   // K / 256, compile-time: with a run-time chunk count every load sat behind its own branch and its own vmcnt(0)
                     // (28 rows x 5 dependent L2 round trips: 46 us for a 6.5 MB weight)
__global__ void __launch_bounds__(256) linear_f32_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         bf16_t* __restrict__ out_silu, int B, int N, int act_out) {
  constexpr int K = NCH * 256;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float4 w[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) w[c] = *(const float4*)(W + (int64_t)n * K + c * 256 + lane * 4);
  const float bn = bias ? bias[n] : 0.f;
  // four batch rows at a time: their loads and shuffle reductions are independent chains that interleave; each row's own
  // summation order is the same as one row at a time
  for (int b0 = 0; b0 < B; b0 += 4) {
    float4 xv[4][NCH];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xr = in + (int64_t)min(b0 + r, B - 1) * K + lane * 4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) xv[r][c] = *(const float4*)(xr + c * 256);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        acc[r] += xv[r][c].x * w[c].x + xv[r][c].y * w[c].y + xv[r][c].z * w[c].z + xv[r][c].w * w[c].w;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = wave_sum(acc[r]) + bn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + r;
      float v = acc[r];
      if (act_out == 1) v = silu_f(v);
      if (lane == 0 && b < B) {
        out[(int64_t)b * N + n] = v;
        if (out_silu) out_silu[(int64_t)b * N + n] = f2bf_e(silu_f(v));
      }
    }
  }
}
hipError_t launch_linear_f32(const float* in, const float* W, const float* bias, float* out, bf16_t* out_silu_bf16,
                             int B, int N, int K, int act_out, hipStream_t s) {
  if (K % 256 != 0 || K > 2048 || K <= 0) return hipErrorInvalidValue;
  const dim3 grid((N + 3) / 4), block(256);
  switch (K / 256) {
#define JAT_LIN(NCH) case NCH: hipLaunchKernelGGL(linear_f32_kernel<NCH>, grid, block, 0, s, in, W, bias, out, out_silu_bf16, B, N, act_out); break;
    JAT_LIN(1) JAT_LIN(2) JAT_LIN(3) JAT_LIN(4) JAT_LIN(5) JAT_LIN(6) JAT_LIN(7) JAT_LIN(8)
#undef JAT_LIN
  }
  return hipGetLastError();
}

// ---- casts -------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n, int silu) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *(const float4*)(in + i);
    if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    *(uint2*)(out + i) = pack4_e(v.x, v.y, v.z, v.w);
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = f2bf_e(silu ? silu_f(in[j]) : in[j]);
  }
}
hipError_t launch_cast_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, in, out, n, 0);
  return hipGetLastError();
}
hipError_t launch_silu_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, in, out, n, 1);
  return hipGetLastError();
}

__global__ void cast_bf16_rope_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  const int h = r >> 6, q = r & 63;
  const int src = (h << 6) + (q >> 1) + ((q & 1) << 5);
  const float4 v = *(const float4*)(in + (int64_t)src * cols + c);
  *(uint2*)(out + i) = pack4_e(v.x, v.y, v.z, v.w);
}
hipError_t launch_cast_bf16_rope_rows(const float* in, bf16_t* out, int rows, int cols, hipStream_t s) {
  if (rows % 64 != 0 || cols % 4 != 0) return hipErrorInvalidValue;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(cast_bf16_rope_rows_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, in, out, rows,
                     cols);
  return hipGetLastError();
}

// ---- weight folding (sampler creation only): out[r][c] = bf16(in[src(r)][c] * g[c]) -------------------------------------
// RMSNorm + adaLN modulation commute with the following Linear when every row shares the modulation (the sampler: one t
// per step, infer_test_v3m2.py:150):  (x * rstd * w * (1 + scale) + shift) @ W^T = rstd * (x @ (W diag(g))^T) + shift @ W^T,
// g = w * (1 + scale).  rope != 0: rows pair-interleaved per 64-row head as launch_cast_bf16_rope_rows does.
__global__ void fold_weight_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                                   bf16_t* __restrict__ out, int rows, int cols, int rope) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  int src = r;
  if (rope) {
    const int h = r >> 6, q = r & 63;
    src = (h << 6) + (q >> 1) + ((q & 1) << 5);
  }
  const float4 v = *(const float4*)(in + (int64_t)src * cols + c);
  float4 g = *(const float4*)(w + c);
  if (scale) {
    const float4 sc = *(const float4*)(scale + c);
    g.x *= 1.0f + sc.x; g.y *= 1.0f + sc.y; g.z *= 1.0f + sc.z; g.w *= 1.0f + sc.w;
  }
  *(uint2*)(out + i) = pack4_e(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
}
hipError_t launch_fold_weight(const float* in, const float* w, const float* scale, bf16_t* out, int rows, int cols, int rope,
                              hipStream_t s) {
  if ((rope && rows % 64 != 0) || cols % 4 != 0) return hipErrorInvalidValue;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(fold_weight_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, in, w, scale, out, rows, cols, rope);
  return hipGetLastError();
}

// ---- norm-folding table helpers (sampler creation only) ---------------------------------------------------------
__global__ void fold_scale_kernel(const float* __restrict__ w, const float* __restrict__ scale, int64_t in_stride,
                                  float* __restrict__ out, int64_t out_stride, int rows, int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, k = i - r * cols;
  out[(int64_t)r * out_stride + k] = w[k] * (1.0f + scale[(int64_t)r * in_stride + k]);
}
hipError_t launch_fold_scale(const float* w, const float* scale, int64_t in_stride, float* out, int64_t out_stride,
                             int rows, int cols, hipStream_t s) {
  const int n = rows * cols;
  hipLaunchKernelGGL(fold_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, scale, in_stride, out, out_stride, rows, cols);
  return hipGetLastError();
}
__global__ void gather_cast_rows_kernel(const float* __restrict__ in, int64_t in_stride, bf16_t* __restrict__ out, int rows,
                                        int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, k = i - r * cols;
  out[(int64_t)r * cols + k] = f2bf_e(in[(int64_t)r * in_stride + k]);
}
hipError_t launch_gather_cast_rows(const float* in, int64_t in_stride, bf16_t* out, int rows, int cols, hipStream_t s) {
  const int n = rows * cols;
  hipLaunchKernelGGL(gather_cast_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, in_stride, out, rows, cols);
  return hipGetLastError();
}

// ---- CFG combine + Euler step (infer_test_v3m2.py:161-179) ----------------------------------------------
// x = u + s (c - u);  z += (x - z) / (1 - t + 1e-5) * dt   (t < 0.999)   |   z = x   (otherwise)
// The branch depends only on the host-side schedule, so it is a kernel argument, not a device read.
__global__ void __launch_bounds__(256) cfg_euler_kernel(const float* __restrict__ xp, float* __restrict__ z,
                                                        float cfg_scale, float denom, float dt, int use_cfg,
                                                        int direct, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 c = *(const float4*)(xp + i);
      if (use_cfg) {
        const float4 u = *(const float4*)(xp + n + i);
        c.x = u.x + cfg_scale * (c.x - u.x); c.y = u.y + cfg_scale * (c.y - u.y);
        c.z = u.z + cfg_scale * (c.z - u.z); c.w = u.w + cfg_scale * (c.w - u.w);
      }
      if (!direct) {
        const float4 zz = *(const float4*)(z + i);
        c.x = zz.x + __fdiv_rn(c.x - zz.x, denom) * dt; c.y = zz.y + __fdiv_rn(c.y - zz.y, denom) * dt;
        c.z = zz.z + __fdiv_rn(c.z - zz.z, denom) * dt; c.w = zz.w + __fdiv_rn(c.w - zz.w, denom) * dt;
      }
      *(float4*)(z + i) = c;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float c = xp[j];
        if (use_cfg) { const float u = xp[n + j]; c = u + cfg_scale * (c - u); }
        if (!direct) c = z[j] + __fdiv_rn(c - z[j], denom) * dt;
        z[j] = c;
      }
    }
  }
}
hipError_t launch_cfg_euler(const float* xp, float* z, float cfg_scale, float t, float dt, int use_cfg,
                            int64_t n_per_half, hipStream_t s) {
  const int direct = !(t < 0.999f);
  const float denom = 1.0f - t + 1e-5f;
  int64_t blocks = (n_per_half / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xp, z, cfg_scale, denom, dt, use_cfg,
                     direct, n_per_half);
  return hipGetLastError();
}

// ---- per-channel (de)normalisation (infer_test_v3m2.py:381-382, 394) -------------------------------------
__global__ void channel_affine_kernel(const float* __restrict__ in, const float* __restrict__ mean,
                                      const float* __restrict__ sd, float* __restrict__ out, int C, int T,
                                      int inverse, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / T) % C);
  out[i] = inverse ? in[i] * sd[c] + mean[c] : (in[i] - mean[c]) / sd[c];
}
hipError_t launch_channel_affine(const float* in, const float* mean, const float* std, float* out, int B, int C, int T,
                                 int inverse, hipStream_t s) {
  const int64_t n = (int64_t)B * C * T;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(channel_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, mean, std, out, C,
                     T, inverse, n);
  return hipGetLastError();
}

// ---- linear crossfade of two chunks (infer_test_v3m2.py:188-233) -----------------------------------------
// out[r, :Tp-ov] = prev ; out[r, Tp-ov:Tp] = prev*fade_out + cur*fade_in ; out[r, Tp:] = cur[r, ov:]
// fade_out = linspace(1,0,ov), fade_in = linspace(0,1,ov) evaluated as torch.linspace does (two halves).
__device__ __forceinline__ float linspace_at(float a, float b, int n, int i) {
  if (n == 1) return a;
  const float step = (b - a) / (float)(n - 1);
  return i < n / 2 ? fmaf(step, (float)i, a) : fmaf(-step, (float)(n - 1 - i), b);
}
__global__ void crossfade_pair_kernel(const float* __restrict__ prev, int Tp, const float* __restrict__ cur, int Tc,
                                      int ov, float* __restrict__ out, int rows) {
  const int To = Tp + Tc - ov;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * To) return;
  const int r = (int)(i / To), t = (int)(i - (int64_t)r * To);
  float v;
  if (t < Tp - ov) {
    v = prev[(int64_t)r * Tp + t];
  } else if (t < Tp) {
    const int k = t - (Tp - ov);
    v = prev[(int64_t)r * Tp + t] * linspace_at(1.f, 0.f, ov, k) + cur[(int64_t)r * Tc + k] * linspace_at(0.f, 1.f, ov, k);
  } else {
    v = cur[(int64_t)r * Tc + (t - Tp + ov)];
  }
  out[i] = v;
}
hipError_t launch_crossfade_pair(const float* prev, int Tp, const float* cur, int Tc, int overlap, float* out, int rows,
                                 hipStream_t s) {
  if (overlap < 0 || overlap > Tp || overlap > Tc) return hipErrorInvalidValue;
  const int64_t n = (int64_t)rows * (Tp + Tc - overlap);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(crossfade_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prev, Tp, cur, Tc,
                     overlap, out, rows);
  return hipGetLastError();
}

// ---- finish of a split-K gated-residual GEMM (small-M inference): fixed summation order, one pass over x -------------
__global__ void __launch_bounds__(256) splitk_resid_finish_kernel(const float* __restrict__ part, int nsplit, int64_t stride,
                                                                  const float* __restrict__ bias, const float* __restrict__ gate,
                                                                  int64_t gate_bstride, int ntok, float* __restrict__ x, int M,
                                                                  int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 columns
  const int per_row = N / 4;
  if (i >= (int64_t)M * per_row) return;
  const int row = (int)(i / per_row), c = (int)(i % per_row) * 4;
  f32x4_e acc = *(const f32x4_e*)(part + (int64_t)row * N + c);
  for (int z = 1; z < nsplit; ++z) {
    const f32x4_e v = *(const f32x4_e*)(part + (int64_t)z * stride + (int64_t)row * N + c);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  const f32x4_e b = bias ? *(const f32x4_e*)(bias + c) : f32x4_e{0.f, 0.f, 0.f, 0.f};
  const f32x4_e g = *(const f32x4_e*)(gate + (int64_t)(row / ntok) * gate_bstride + c);
  f32x4_e xv = *(const f32x4_e*)(x + (int64_t)row * N + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) xv[j] += g[j] * (acc[j] + b[j]);
  *(f32x4_e*)(x + (int64_t)row * N + c) = xv;
}
// ---- finish of a split-K Linear + GELU (the first patch-embed Linear, jat_audiosr_v3.py:221-223: [M, 8192] x [512, 8192]^T is 224
// tiles of 64 x 128 with 64-128 K-tiles each — one 4-wave block per CU; K slices put two on every CU): slice sum in fixed order,
// bias, GELU (the epilogue's own expression: jat_gelu.h), bf16.  dual_rows > 0 (CFG sampler, split patch embed): row m gets
// gelu(sum + bias + dual_add[m]), row m + dual_rows gets gelu(sum + bias) — what EPI_BF16_GELU's dual output writes.
__global__ void __launch_bounds__(256) splitk_gelu_finish_kernel(const float* __restrict__ part, int nsplit, int64_t stride,
                                                                 const float* __restrict__ bias, const float* __restrict__ dual_add,
                                                                 int dual_rows, bf16_t* __restrict__ out, int64_t ldo, int M, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 columns
  const int per_row = N / 4;
  if (i >= (int64_t)M * per_row) return;
  const int row = (int)(i / per_row), c = (int)(i % per_row) * 4;
  f32x4_e acc = *(const f32x4_e*)(part + (int64_t)row * N + c);
  for (int z = 1; z < nsplit; ++z) {
    const f32x4_e v = *(const f32x4_e*)(part + (int64_t)z * stride + (int64_t)row * N + c);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  const f32x4_e b = bias ? *(const f32x4_e*)(bias + c) : f32x4_e{0.f, 0.f, 0.f, 0.f};
  acc = acc + b;
  if (dual_rows > 0) {
    const f32x4_e d = *(const f32x4_e*)(dual_add + (int64_t)row * N + c);
    f32x2 g[4] = {f32x2{acc[0] + d[0], acc[1] + d[1]}, f32x2{acc[2] + d[2], acc[3] + d[3]}, f32x2{acc[0], acc[1]}, f32x2{acc[2], acc[3]}};
    gelu_erf_n<4>(g);
    *(uint2*)(out + (int64_t)row * ldo + c) = pack4_e(g[0][0], g[0][1], g[1][0], g[1][1]);
    *(uint2*)(out + (int64_t)(row + dual_rows) * ldo + c) = pack4_e(g[2][0], g[2][1], g[3][0], g[3][1]);
  } else {
    f32x2 g[2] = {f32x2{acc[0], acc[1]}, f32x2{acc[2], acc[3]}};
    gelu_erf_n<2>(g);
    *(uint2*)(out + (int64_t)row * ldo + c) = pack4_e(g[0][0], g[0][1], g[1][0], g[1][1]);
  }
}
hipError_t launch_splitk_gelu_finish(const float* part, int nsplit, int64_t stride, const float* bias, const float* dual_add,
                                     int dual_rows, bf16_t* out, int64_t ldo, int M, int N, hipStream_t s) {
  if (N % 4 != 0 || nsplit < 1 || (dual_rows > 0 && !dual_add)) return hipErrorInvalidValue;
  const int64_t n = (int64_t)M * (N / 4);
  hipLaunchKernelGGL(splitk_gelu_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nsplit, stride, bias,
                     dual_add, dual_rows, out, ldo, M, N);
  return hipGetLastError();
}
// ---- finish of a split-K QKV GEMM (one chunk: M = 256, 56 tiles - five K-slices put 280 blocks on the weights): slice sum in
// fixed order, RoPE on the pair-interleaved q / k features (gemm.hip EPI_QKV_ROPE: the pair (2d', 2d'+1) of a head holds
// features d', d'+32), bf16 q / k rows and the transposed, key-padded V^T the attention kernels read.  One thread = 4 columns.
__global__ void __launch_bounds__(256) splitk_qkv_finish_kernel(const float* __restrict__ part, int nsplit, int64_t stride,
                                                                const float* __restrict__ rope_cos,
                                                                const float* __restrict__ rope_sin, bf16_t* __restrict__ q,
                                                                bf16_t* __restrict__ k, bf16_t* __restrict__ vt, int M, int D,
                                                                int kvD, int ntok, int npad) {
  const int Nq = D + 2 * kvD, per_row = Nq / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * per_row) return;
  const int row = (int)(i / per_row), n = (int)(i % per_row) * 4;
  f32x4_e acc = *(const f32x4_e*)(part + (int64_t)row * Nq + n);
  for (int z = 1; z < nsplit; ++z) {
    const f32x4_e v = *(const f32x4_e*)(part + (int64_t)z * stride + (int64_t)row * Nq + n);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  const int b = row / ntok, pos = row - b * ntok;
  if (n < D + kvD) {
    const int d0 = (n & 63) >> 1;
    const float2 c = *(const float2*)(rope_cos + (int64_t)pos * 32 + d0);
    const float2 sn = *(const float2*)(rope_sin + (int64_t)pos * 32 + d0);
    const float o0 = __builtin_fmaf(acc[0], c.x, -(acc[1] * sn.x)), o1 = __builtin_fmaf(acc[1], c.x, acc[0] * sn.x);
    const float o2 = __builtin_fmaf(acc[2], c.y, -(acc[3] * sn.y)), o3 = __builtin_fmaf(acc[3], c.y, acc[2] * sn.y);
    bf16_t* dst = n < D ? q + (int64_t)row * D + n : k + (int64_t)row * kvD + (n - D);
    *(uint2*)dst = pack4_e(o0, o1, o2, o3);
  } else {
    const int nv = n - D - kvD;   // 4 consecutive features of one V head (64 | 4)
    bf16_t* dst = vt + ((int64_t)(b * (kvD >> 6) + (nv >> 6)) * 64 + (nv & 63)) * npad + pos;
    const uint2 pk = pack4_e(acc[0], acc[1], acc[2], acc[3]);
    dst[0] = (bf16_t)(pk.x & 0xffffu); dst[npad] = (bf16_t)(pk.x >> 16);
    dst[2 * (int64_t)npad] = (bf16_t)(pk.y & 0xffffu); dst[3 * (int64_t)npad] = (bf16_t)(pk.y >> 16);
  }
}
hipError_t launch_splitk_qkv_finish(const float* part, int nsplit, int64_t stride, const float* rope_cos, const float* rope_sin,
                                    bf16_t* q, bf16_t* k, bf16_t* vt, int M, int D, int kvD, int ntok, int npad, hipStream_t s) {
  if (D % 64 != 0 || kvD % 64 != 0 || nsplit < 1 || npad < ntok) return hipErrorInvalidValue;
  const int64_t n = (int64_t)M * ((D + 2 * kvD) / 4);
  hipLaunchKernelGGL(splitk_qkv_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nsplit, stride, rope_cos,
                     rope_sin, q, k, vt, M, D, kvD, ntok, npad);
  return hipGetLastError();
}
// The same finish FUSED with the norm + modulation that consumes the new residual row (norm2 of the block, norm1 of the next
// block, or the final norm): one BLOCK per row, one float4 column per thread (D / 4 threads), the row statistics through a
// fixed-order block reduction.  At one chunk (M = 256) a captured kernel node costs ~4 us whatever it does; this keeps the
// wide, short shape of the finishing pass (a one-wave-per-row version with 40 dependent loads per lane was slower than the
// two separate launches).
__global__ void __launch_bounds__(512) splitk_resid_norm_block_kernel(const float* __restrict__ part, int nsplit, int64_t stride,
                                                                      const float* __restrict__ bias, const float* __restrict__ gate,
                                                                      int64_t gate_bstride, float* __restrict__ x,
                                                                      const float* __restrict__ w, const float* __restrict__ shift,
                                                                      const float* __restrict__ scale, int64_t mod_bstride,
                                                                      bf16_t* __restrict__ y, int D, int ntok, int mode) {
  __shared__ float red[2][8];
  const int row = blockIdx.x, col = threadIdx.x * 4, b = row / ntok;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x4_e acc = *(const f32x4_e*)(part + (int64_t)row * D + col);
  for (int z = 1; z < nsplit; ++z) {
    const f32x4_e v = *(const f32x4_e*)(part + (int64_t)z * stride + (int64_t)row * D + col);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  const f32x4_e bb = bias ? *(const f32x4_e*)(bias + col) : f32x4_e{0.f, 0.f, 0.f, 0.f};
  const f32x4_e g = *(const f32x4_e*)(gate + (int64_t)b * gate_bstride + col);
  f32x4_e xv = *(const f32x4_e*)(x + (int64_t)row * D + col);
#pragma unroll
  for (int j = 0; j < 4; ++j) xv[j] += g[j] * (acc[j] + bb[j]);
  *(f32x4_e*)(x + (int64_t)row * D + col) = xv;
  auto block_sum = [&](float v, int slot) {   // fixed order: lanes by shuffles, then the waves in index order
    v = wave_sum(v);
    if (lane == 0) red[slot][wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[slot][i];
    return t;
  };
  // the norm weight and this sample's modulation do not depend on the row statistics: requested before the block reduction
  // (its barrier), not after it — one L2 round trip less on the critical path of a launch that is all latency at small M
  const f32x4_e ww = (mode == 0 && w) ? *(const f32x4_e*)(w + col) : f32x4_e{1.f, 1.f, 1.f, 1.f};
  f32x4_e ma = f32x4_e{0.f, 0.f, 0.f, 0.f}, ms = f32x4_e{0.f, 0.f, 0.f, 0.f};
  if (scale) {
    ma = *(const f32x4_e*)(scale + (int64_t)b * mod_bstride + col);
    ms = *(const f32x4_e*)(shift + (int64_t)b * mod_bstride + col);
  }
  float mu = 0.f, rstd = 1.f;
  if (mode == 0) {
    rstd = rsqrtf(block_sum(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3], 0) / (float)D + 1e-6f);
  } else if (mode == 1) {
    mu = block_sum(xv[0] + xv[1] + xv[2] + xv[3], 0) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) var += (xv[e] - mu) * (xv[e] - mu);
    rstd = rsqrtf(block_sum(var, 1) / (float)D + 1e-6f);
  }
  f32x4_e t = (xv - mu) * rstd * ww;
  if (scale) t = t * (1.f + ma) + ms;
  *(uint2*)(y + (int64_t)row * D + col) = pack4_e(t[0], t[1], t[2], t[3]);
}
// the fused form exists for widths of 256 .. 2048 in steps of 256 (D / 4 threads = whole waves, at most 8)
bool splitk_resid_norm_supported(int D) { return D % 256 == 0 && D >= 256 && D <= 2048; }
hipError_t launch_splitk_resid_norm(const float* part, int nsplit, int64_t stride, const float* bias, const float* gate,
                                    int64_t gate_bstride, float* x, const float* w, const float* shift, const float* scale,
                                    int64_t mod_bstride, bf16_t* y, int M, int D, int ntok, int mode, hipStream_t s) {
  if (!splitk_resid_norm_supported(D) || nsplit < 1 || (shift == nullptr) != (scale == nullptr)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(splitk_resid_norm_block_kernel, dim3(M), dim3(D / 4), 0, s, part, nsplit, stride, bias, gate, gate_bstride, x, w,
                     shift, scale, mod_bstride, y, D, ntok, mode);
  return hipGetLastError();
}
hipError_t launch_splitk_resid_finish(const float* part, int nsplit, int64_t stride, const float* bias, const float* gate,
                                      int64_t gate_bstride, int ntok, float* x, int M, int N, hipStream_t s) {
  if (N % 4 != 0 || nsplit < 1) return hipErrorInvalidValue;
  const int64_t n = (int64_t)M * (N / 4);
  hipLaunchKernelGGL(splitk_resid_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nsplit, stride, bias,
                     gate, gate_bstride, ntok, x, M, N);
  return hipGetLastError();
}
