// Backward-pass and optimiser kernels of the training step (train_ddp_v3m2.py:533-622).
// The three GEMMs of every Linear (y = x W^T, dx = dy W, dW = dy^T x) all run on gemm_bf16_kernel (gemm.hip): its
// operands are K-contiguous, so dx uses a transposed bf16 weight copy and dW uses transposed activation copies made
// by transpose_bf16_kernel.  Everything here is either HBM-bound row/column work or the attention backward.
// All reductions are fixed-order (partials + a finishing kernel): no atomics, a step is bit-reproducible.
#include "jat_kernels.h"
#include "jat_dtype.h"
#include "jat_rng.h"
#include <cstdlib>

typedef jat_opx8 bf16x8_t;   // operand fragment: bf16, or fp16 in the -DJAT_FP16 build (jat_dtype.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

__device__ __forceinline__ unsigned short f2bf_t(float f) { return jat_f2op(f); }
__device__ __forceinline__ float bf2f_t(unsigned short u) { return jat_op2f(u); }
__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = jat_lo2f(v[i]);
    f[2 * i + 1] = jat_hi2f(v[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = jat_pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---- bf16 transpose: out[c][m] = in[m][c], m >= M zero-filled up to Mpad (the K dimension of a dW GEMM) ----------
// 64 x 64 tiles: 16-B global loads -> row-major LDS image (144-B pitch) -> ds_read_b64_tr_b16 hands every lane 8
// consecutive m of one output row -> 16-B global stores.
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int M, int C,
                                                             bf16_t* __restrict__ out, int Mpad) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = pass * 32 + (tid >> 3), ch = tid & 7;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (m0 + r < M) v = *(const u32x4_t*)(in + (int64_t)(m0 + r) * ld_in + c0 + ch * 8);
    *(u32x4_t*)&tile[r][ch * 8] = v;
  }
  __syncthreads();
  // wave w owns output rows c0 + 16 w + fr; lane group fg owns m = 32 ks + 8 fg .. + 7
  const int q = fr >> 2, pp = fr & 3;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)&tile[ks * 32 + 8 * fg + q][wave * 16 + 4 * pp]);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)&tile[ks * 32 + 8 * fg + 4 + q][wave * 16 + 4 * pp]);
    u32x4_t v;
    v[0] = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    v[1] = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    v[2] = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    v[3] = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    *(u32x4_t*)(out + (int64_t)(c0 + wave * 16 + fr) * Mpad + m0 + ks * 32 + 8 * fg) = v;   // Mpad % 64 == 0: in range
  }
}
hipError_t launch_transpose_bf16(const bf16_t* in, int64_t ld_in, int M, int C, bf16_t* out, int Mpad, hipStream_t s) {
  if (C % 64 != 0 || Mpad % 64 != 0 || Mpad < M) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mpad / 64, C / 64), dim3(256), 0, s, in, ld_in, M, C, out, Mpad);
  return hipGetLastError();
}

// out[i] = sum_z part[z * stride + i]   (split-K partials of a dW GEMM, fixed order)
__global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ part, int nsplit, int64_t stride,
                                                           float* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4_t acc = *(const f32x4_t*)(part + i * 4);
  for (int z = 1; z < nsplit; ++z) {
    const f32x4_t v = *(const f32x4_t*)(part + (int64_t)z * stride + i * 4);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  *(f32x4_t*)(out + i * 4) = acc;
}
hipError_t launch_sum_partials(const float* part, int nsplit, int64_t stride, float* out, int64_t n, hipStream_t s) {
  if (n % 4 != 0 || stride % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, nsplit, stride, out, n / 4);
  return hipGetLastError();
}

// ---- row sums of a bf16 matrix [R][ld] over its first n columns -> fp32 [R]   (bias gradients from dY^T) -------
__global__ void __launch_bounds__(256) rowsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int R, int n,
                                                          float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float acc = 0.f;
  for (int c = lane * 8; c < n; c += 512) {   // n % 8 == 0
    float f[8];
    unpack8(*(const u32x4_t*)(x + (int64_t)row * ld + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += f[i];
  }
  acc = wave_sum_t(acc);
  if (lane == 0) out[row] = acc;
}
hipError_t launch_rowsum_bf16(const bf16_t* x, int64_t ld, int R, int n, float* out, hipStream_t s) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, ld, R, n, out);
  return hipGetLastError();
}

// ---- GELU (erf form, nn.GELU default) forward on bf16 and its backward ------------------------------------------
__device__ __forceinline__ float gelu_t(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_t(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// out = dropout(gelu(in)): element e is scaled by jat_drop_mult(drop, e)   (nn.Dropout after nn.GELU, :268-269)
__global__ void __launch_bounds__(256) gelu_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t n8,
                                                        const DropSpec drop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float f[8];
  unpack8(*(const u32x4_t*)(in + i * 8), f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = gelu_t(f[j]);
  if (drop.thresh) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= jat_drop_mult(drop, (uint64_t)(i * 8 + j));
  }
  *(u32x4_t*)(out + i * 8) = pack8(f);
}
// dpre = dpost * mask * gelu'(pre), in place on dpost
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ d, int64_t n8,
                                                       const DropSpec drop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float f[8], g[8];
  unpack8(*(const u32x4_t*)(pre + i * 8), f);
  unpack8(*(const u32x4_t*)(d + i * 8), g);
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] *= dgelu_t(f[j]);
  if (drop.thresh) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= jat_drop_mult(drop, (uint64_t)(i * 8 + j));
  }
  *(u32x4_t*)(d + i * 8) = pack8(g);
}
hipError_t launch_gelu_bf16(const bf16_t* in, bf16_t* out, int64_t n, DropSpec drop, hipStream_t s) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gelu_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, in, out, n / 8, drop);
  return hipGetLastError();
}
hipError_t launch_gelu_bwd(const bf16_t* pre, bf16_t* d, int64_t n, DropSpec drop, hipStream_t s) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, pre, d, n / 8, drop);
  return hipGetLastError();
}

// ---- gated residual forward with the branch output kept: x_out = x_in + gate[b] * y   (jat_audiosr_v3.py:300,306) --
__global__ void __launch_bounds__(256) resid_gate_kernel(const float* __restrict__ x_in, const bf16_t* __restrict__ y,
                                                         const float* __restrict__ gate, int64_t gate_bstride,
                                                         float* __restrict__ x_out, int M, int D, int ntok,
                                                         const DropSpec path, const DropSpec elem) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread = 8 columns
  const int per_row = D / 8;
  if (i >= (int64_t)M * per_row) return;
  const int row = (int)(i / per_row), c = (int)(i % per_row) * 8;
  const float* g = gate + (int64_t)(row / ntok) * gate_bstride + c;
  float f[8];
  unpack8(*(const u32x4_t*)(y + (int64_t)row * D + c), f);
  if (path.thresh | elem.thresh) {   // DropPath: per-sample scale of the branch; Dropout: per-element scale of y
    const float pm = path.thresh ? jat_drop_mult(path, (uint64_t)(row / ntok)) : 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= pm * (elem.thresh ? jat_drop_mult(elem, (uint64_t)row * D + c + j) : 1.0f);
  }
  const f32x4_t a0 = *(const f32x4_t*)(x_in + (int64_t)row * D + c), a1 = *(const f32x4_t*)(x_in + (int64_t)row * D + c + 4);
  const f32x4_t g0 = *(const f32x4_t*)g, g1 = *(const f32x4_t*)(g + 4);
  f32x4_t o0, o1;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o0[j] = a0[j] + g0[j] * f[j]; o1[j] = a1[j] + g1[j] * f[4 + j]; }
  *(f32x4_t*)(x_out + (int64_t)row * D + c) = o0;
  *(f32x4_t*)(x_out + (int64_t)row * D + c + 4) = o1;
}
hipError_t launch_resid_gate(const float* x_in, const bf16_t* y, const float* gate, int64_t gate_bstride, float* x_out,
                             int M, int D, int ntok, DropSpec path, DropSpec elem, hipStream_t s) {
  const int64_t n = (int64_t)M * (D / 8);
  hipLaunchKernelGGL(resid_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x_in, y, gate, gate_bstride,
                     x_out, M, D, ntok, path, elem);
  return hipGetLastError();
}

// ---- backward of the gated residual: dy = bf16(dx * gate[b]);  dgate[b][n] = sum_tok dx * y -----------------------
// grid (chunks of TOKC tokens, B); partial sums part[b][chunk][D]; finished by reduce_chunks_kernel.
constexpr int TOKC = 16;
// blockDim = D/4 threads when that is at most 512 (D = 1280: 320 threads, one float4 column each: with 256 threads the second
// pass over the columns ran a quarter full)
__global__ void __launch_bounds__(512) gate_bwd_kernel(const float* __restrict__ dx, const bf16_t* __restrict__ y,
                                                       const float* __restrict__ gate, int64_t gate_bstride,
                                                       bf16_t* __restrict__ dy, float* __restrict__ part, int D, int ntok,
                                                       const DropSpec path, const DropSpec elem) {
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int t0 = chunk * TOKC, t1 = min(t0 + TOKC, ntok);
  const float pm = path.thresh ? jat_drop_mult(path, (uint64_t)b) : 1.0f;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    f32x4_t g = *(const f32x4_t*)(gate + (int64_t)b * gate_bstride + c);
    g[0] *= pm; g[1] *= pm; g[2] *= pm; g[3] *= pm;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = t0; t < t1; ++t) {
      const int64_t row = (int64_t)b * ntok + t;
      const f32x4_t d = *(const f32x4_t*)(dx + row * D + c);
      const u32x2_t yy = *(const u32x2_t*)(y + row * D + c);
      const float y0 = jat_lo2f(yy[0]), y1 = jat_hi2f(yy[0]);
      const float y2 = jat_lo2f(yy[1]), y3 = jat_hi2f(yy[1]);
      float e0 = pm, e1 = pm, e2 = pm, e3 = pm;   // d(branch)/d(gate*y) multipliers: DropPath x element dropout
      float h0 = 1.f, h1 = 1.f, h2 = 1.f, h3 = 1.f;
      if (elem.thresh) {
        const uint64_t e = (uint64_t)row * D + c;
        h0 = jat_drop_mult(elem, e); h1 = jat_drop_mult(elem, e + 1); h2 = jat_drop_mult(elem, e + 2); h3 = jat_drop_mult(elem, e + 3);
      }
      e0 *= h0; e1 *= h1; e2 *= h2; e3 *= h3;
      acc[0] += d[0] * y0 * e0; acc[1] += d[1] * y1 * e1; acc[2] += d[2] * y2 * e2; acc[3] += d[3] * y3 * e3;
      u32x2_t o;
      o[0] = jat_pack2(d[0] * g[0] * h0, d[1] * g[1] * h1);
      o[1] = jat_pack2(d[2] * g[2] * h2, d[3] * g[3] * h3);
      *(u32x2_t*)(dy + row * D + c) = o;
    }
    *(f32x4_t*)(part + ((int64_t)b * nchunk + chunk) * D + c) = acc;
  }
}
// out[b*out_bstride + c] = sum_chunk part[(b*nchunk + chunk)*chunk_stride + c]   (sum_b: also over b, out has one row)
__global__ void __launch_bounds__(256) reduce_chunks_kernel(const float* __restrict__ part, int nchunk, int64_t chunk_stride,
                                                            float* __restrict__ out, int64_t out_bstride, int B, int ncols,
                                                            int sum_b) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  if (sum_b) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < nchunk; ++k) acc += part[((int64_t)b * nchunk + k) * chunk_stride + c];
    out[c] = acc;
  } else {
    const int b = blockIdx.y;
    float acc = 0.f;
    for (int k = 0; k < nchunk; ++k) acc += part[((int64_t)b * nchunk + k) * chunk_stride + c];
    out[(int64_t)b * out_bstride + c] = acc;
  }
}
int train_nchunk(int ntok) { return (ntok + TOKC - 1) / TOKC; }
hipError_t launch_gate_bwd(const float* dx, const bf16_t* y, const float* gate, int64_t gate_bstride, bf16_t* dy,
                           float* part, float* dgate, int64_t dgate_bstride, int B, int D, int ntok, DropSpec path,
                           DropSpec elem, hipStream_t s) {
  const int nchunk = train_nchunk(ntok);
  const int nthr = (D / 4 <= 512 && D % 256 == 0) ? D / 4 : 256;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(nchunk, B), dim3(nthr), 0, s, dx, y, gate, gate_bstride, dy, part, D, ntok, path,
                     elem);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((D + 255) / 256, B), dim3(256), 0, s, part, nchunk, (int64_t)D, dgate,
                     dgate_bstride, B, D, 0);
  return hipGetLastError();
}

// ---- backward of y = norm(x) * w * (1 + scale[b]) + shift[b] ---------------------------------------------------------
// mode 0 RMSNorm(+w): xh = x * rstd;  mode 1 LayerNorm(no affine): xh = (x - mu) * rstd.   g = dy * w * (1 + scale).
//   dx += rstd * (g - [mode 1: mean(g)] - xh * mean(g * xh));  dshift[b] = sum_tok dy;  dscale[b] = sum_tok dy * xh * w;
//   dw = sum_rows dy * (1 + scale) * xh.
// One wave per row (row in registers), each wave walks the rows of its token chunk and keeps per-column partial sums;
// the four waves of a block are combined through LDS.  part[b][chunk][3][D] -> reduce_chunks_kernel.
template <int NCH>
__global__ void __launch_bounds__(256) norm_bwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       const float* __restrict__ w, const float* __restrict__ scale,
                                                       int64_t mod_bstride, float* __restrict__ dx, float* __restrict__ part,
                                                       int D, int ntok, int mode, int accumulate) {
  extern __shared__ float red[];   // [4 waves][3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int t0 = chunk * TOKC, t1 = min(t0 + TOKC, ntok);
  constexpr int nch = NCH;
  f32x4_t a_sh[NCH], a_sc[NCH], a_w[NCH], ww[NCH], sc1[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    a_sh[c] = a_sc[c] = a_w[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    ww[c] = sc1[c] = f32x4_t{1.f, 1.f, 1.f, 1.f};
    if (c < nch) {
      if (mode == 0 && w) ww[c] = *(const f32x4_t*)(w + c * 256 + lane * 4);
      if (scale) {
        const f32x4_t s4 = *(const f32x4_t*)(scale + (int64_t)b * mod_bstride + c * 256 + lane * 4);
        sc1[c] = f32x4_t{1.f + s4[0], 1.f + s4[1], 1.f + s4[2], 1.f + s4[3]};
      }
    }
  }
  // the next row's loads (x, dy) are issued before the current row's three dependent wave
  // reductions: a wave keeps two rows in flight (one row at a time left the kernel latency-bound at 55 us for 173 MB)
  f32x4_t nx[NCH];
  u32x2_t nd[NCH];
  auto load_row = [&](int t) {
    const int64_t row = (int64_t)b * ntok + t;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      nx[c] = *(const f32x4_t*)(x + row * D + c * 256 + lane * 4);
      nd[c] = *(const u32x2_t*)(dy + row * D + c * 256 + lane * 4);
    }
  };
  if (t0 + wave < t1) load_row(t0 + wave);
  for (int t = t0 + wave; t < t1; t += 4) {
    const int64_t row = (int64_t)b * ntok + t;
    f32x4_t xv[NCH], gv[NCH], dv[NCH], old[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      xv[c] = nx[c];
      if (accumulate) old[c] = *(const f32x4_t*)(dx + row * D + c * 256 + lane * 4);   // needed last: flies under the reductions
      dv[c][0] = jat_lo2f(nd[c][0]); dv[c][1] = jat_hi2f(nd[c][0]);
      dv[c][2] = jat_lo2f(nd[c][1]); dv[c][3] = jat_hi2f(nd[c][1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1 += xv[c][j]; s2 += xv[c][j] * xv[c][j]; }
    }
    if (t + 4 < t1) load_row(t + 4);
    float mu = 0.f, rstd;
    if (mode == 0) {
      rstd = rsqrtf(wave_sum_t(s2) / (float)D + 1e-6f);
    } else {
      mu = wave_sum_t(s1) / (float)D;
      float var = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float e = xv[c][j] - mu; var += e * e; }
      rstd = rsqrtf(wave_sum_t(var) / (float)D + 1e-6f);
    }
    float mg = 0.f, mgx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xv[c][j] - mu) * rstd;
        const float g = dv[c][j] * ww[c][j] * sc1[c][j];
        xv[c][j] = xh; gv[c][j] = g;
        mg += g; mgx += g * xh;
        a_sh[c][j] += dv[c][j];
        a_sc[c][j] += dv[c][j] * xh * ww[c][j];
        a_w[c][j] += dv[c][j] * sc1[c][j] * xh;
      }
    mg = mode == 1 ? wave_sum_t(mg) / (float)D : 0.f;
    mgx = wave_sum_t(mgx) / (float)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd * (gv[c][j] - mg - xv[c][j] * mgx);
      if (accumulate) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += old[c][j];
      }
      *(f32x4_t*)(dx + row * D + c * 256 + lane * 4) = o;
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (c < nch) {
      *(f32x4_t*)(red + (wave * 3 + 0) * D + c * 256 + lane * 4) = a_sh[c];
      *(f32x4_t*)(red + (wave * 3 + 1) * D + c * 256 + lane * 4) = a_sc[c];
      *(f32x4_t*)(red + (wave * 3 + 2) * D + c * 256 + lane * 4) = a_w[c];
    }
  __syncthreads();
  float* po = part + ((int64_t)b * nchunk + chunk) * 3 * D;
  for (int i = threadIdx.x; i < 3 * D; i += 256)
    po[i] = red[i] + red[3 * D + i] + red[6 * D + i] + red[9 * D + i];
}
// which = blockIdx.z: 0 -> dshift[b], 1 -> dscale[b], 2 -> dw_part[b]  (each the sum over the token chunks of sample b)
__global__ void __launch_bounds__(256) norm_bwd_finish_kernel(const float* __restrict__ part, int nchunk, int D,
                                                              float* __restrict__ dshift, float* __restrict__ dscale,
                                                              int64_t dmod_bstride, float* __restrict__ dw_part) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, which = blockIdx.z;
  float* out = which == 0 ? dshift : (which == 1 ? dscale : dw_part);
  if (c >= D || !out) return;
  float acc = 0.f;
  for (int k = 0; k < nchunk; ++k) acc += part[(((int64_t)b * nchunk + k) * 3 + which) * D + c];
  out[(int64_t)b * (which == 2 ? (int64_t)D : dmod_bstride) + c] = acc;
}
// dshift/dscale: [B] rows with stride dmod_bstride (nullptr: skip); dw [D] (nullptr: skip).  part: B*nchunk*3*D floats,
// dw_part: B*D floats.
hipError_t launch_norm_bwd(const float* x, const bf16_t* dy, const float* w, const float* scale, int64_t mod_bstride,
                           float* dx, int accumulate, float* part, float* dw_part, float* dshift, float* dscale,
                           int64_t dmod_bstride, float* dw, int B, int D, int ntok, int mode, hipStream_t s) {
  if (D % 256 != 0 || D > 2048) return hipErrorInvalidValue;
  const int nchunk = train_nchunk(ntok);
#define NORM_BWD_CASE(NCH)                                                                                              \
  case NCH: {                                                                                                           \
    static const hipError_t attr = hipFuncSetAttribute((const void*)norm_bwd_kernel<NCH>,                               \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 12 * NCH * 256 * 4); \
    if (attr != hipSuccess) return attr;                                                                                \
    hipLaunchKernelGGL(norm_bwd_kernel<NCH>, dim3(nchunk, B), dim3(256), (size_t)12 * D * 4, s, x, dy, w, scale,        \
                       mod_bstride, dx, part, D, ntok, mode, accumulate);                                               \
    break;                                                                                                              \
  }
  switch (D >> 8) {
    NORM_BWD_CASE(1) NORM_BWD_CASE(2) NORM_BWD_CASE(3) NORM_BWD_CASE(4) NORM_BWD_CASE(5) NORM_BWD_CASE(6) NORM_BWD_CASE(7)
    NORM_BWD_CASE(8)
  }
#undef NORM_BWD_CASE
  // one launch finishes dshift / dscale per sample and the per-sample part of dw; a second sums dw over the samples
  hipLaunchKernelGGL(norm_bwd_finish_kernel, dim3((D + 255) / 256, B, 3), dim3(256), 0, s, part, nchunk, D, dshift, dscale,
                     dmod_bstride, dw ? dw_part : nullptr);
  if (dw)
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((D + 255) / 256), dim3(256), 0, s, dw_part, 1, (int64_t)D, dw, (int64_t)0, B,
                       D, 1);
  return hipGetLastError();
}

// ---- attention backward (GQA, no mask, no dropout; jat_audiosr_v3.py:164-181) ------------------------------------------
// S = Q K^T / 8, P = softmax(S), O = P V.  With lse2 = log2-domain log-sum-exp saved by the forward and
// delta_i = sum_d dO_id O_id:   P = exp2(S * c - lse2),  dP = dO V^T,  dS = P (dP - delta) / 8,
//   dV = P^T dO,  dK = dS^T Q  (summed over the 5 query heads of the KV group),  dQ = dS K.
// Two kernels so that every output has exactly one writer (no atomics): attn_bwd_dkv (one block per 64-key block of a
// (batch, KV head), looping over its query heads and all query blocks) and attn_bwd_dq (one block per 64-query block
// of a (batch, query head), looping over key blocks).  All five products are v_mfma_f32_16x16x32_bf16 over operand
// tiles staged in LDS as [64][64] bf16 images with a 144-B row pitch (conflict-free ds_read_b128).  The inverse RoPE
// rotation (transpose of jat_audiosr_v3.py:87-108, pair-interleaved feature order) is applied to dQ / dK on the way out.
constexpr int AP = 72;   // LDS row pitch in bf16 elements
struct AttnBwdArgs {
  const bf16_t *q, *k, *vt, *dout;   // q, dout [M, ldq]; k [M, ldk]; vt [B, Hkv, 64, npad]
  const float *lse, *delta;           // [B, Hq, N]
  bf16_t* dqkv;                       // [M, ldg]: dq at column 0, dk at column D, dv at column D + kvD
  const float *rope_cos, *rope_sin;   // [max_pos, 32]
  int64_t ldq, ldk, ldg;
  int B, N, Hq, Hkv, npad, D, kvD;
  float scale_log2e, scale;
  DropSpec drop;   // attention-probability dropout (:175): element ((b*Hq + h)*N + i)*N + j
  // dK/dV: the G query heads of a KV group may be spread over `hsplit` blocks (more, shorter blocks: better balance);
  // each then writes an fp32 partial [(b*Hkv + g)*hsplit + part][N][128] (dK | dV) summed by attn_dkv_reduce_kernel
  float* dkv_part;
  int hsplit;
};

// 64 x 64 bf16 tile staging, split in a global-load half (issued one iteration ahead) and an LDS-store half.
// rows r0 + r of `src` (rows >= nvalid read as zero), 64 contiguous columns.
struct TileRegs { u32x4_t v[2]; };
__device__ __forceinline__ TileRegs tile_load(const bf16_t* __restrict__ src, int64_t ld, int r0, int nvalid, int tid) {
  TileRegs t;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = pass * 32 + (tid >> 3), ch = tid & 7;
    t.v[pass] = u32x4_t{0u, 0u, 0u, 0u};
    if (r0 + r < nvalid) t.v[pass] = *(const u32x4_t*)(src + (int64_t)(r0 + r) * ld + ch * 8);
  }
  return t;
}
__device__ __forceinline__ void tile_store(const TileRegs& t, unsigned short (*dst)[AP], int tid) {
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) *(u32x4_t*)&dst[pass * 32 + (tid >> 3)][(tid & 7) * 8] = t.v[pass];
}
// acc[nt] (16 x 16 tile nt of a 16 x 64 strip) += X[xr0 + 0..15][0..63] * Y[16 nt + 0..15][0..63]^T
__device__ __forceinline__ void strip_mma(unsigned short (*X)[AP], int xr0, unsigned short (*Y)[AP], f32x4_t* acc, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t a = *(const bf16x8_t*)&X[xr0 + fr][ks * 32 + fg * 8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const bf16x8_t bb = *(const bf16x8_t*)&Y[nt * 16 + fr][ks * 32 + fg * 8];
      acc[nt] = JAT_MFMA_16x16x32(a, bb, acc[nt], 0, 0, 0);
    }
  }
}
// A row-read-only image without padding: [64][64] bf16 (128-B rows), 16-B chunks XOR-swizzled by (row & 7) — conflict-free
// ds_read_b128 like the GEMM's tiles, and 1 KiB smaller than the padded form (what lets three dK/dV blocks share a CU).
__device__ __forceinline__ void tile_store_swz(const TileRegs& t, unsigned short (*dst)[64], int tid) {
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = pass * 32 + (tid >> 3), ch = tid & 7;
    *(u32x4_t*)&dst[r][(ch ^ (r & 7)) * 8] = t.v[pass];
  }
}
// acc[nt] += X[xr0 + m][k] * Ysw[16 nt + n][k]   with Ysw a swizzled unpadded image
__device__ __forceinline__ void strip_mma_swzY(unsigned short (*X)[AP], int xr0, unsigned short (*Ysw)[64], f32x4_t* acc, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t a = *(const bf16x8_t*)&X[xr0 + fr][ks * 32 + fg * 8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int r = nt * 16 + fr;
      const bf16x8_t bb = *(const bf16x8_t*)&Ysw[r][((ks * 4 + fg) ^ (r & 7)) * 8];
      acc[nt] = JAT_MFMA_16x16x32(a, bb, acc[nt], 0, 0, 0);
    }
  }
}
// acc[nt] += Xsw[xr0 + m][k] * Yt[k][16 nt + n]      (first operand: swizzled unpadded image; second stored k-major)
__device__ __forceinline__ bf16x8_t tr_frag(unsigned short (*img)[AP], int k0, int c0, int lane);
__device__ __forceinline__ void strip_mma_tr_swzX(unsigned short (*Xsw)[64], int xr0, unsigned short (*Yt)[AP], f32x4_t* acc,
                                                  int lane) {
  const int fr = lane & 15, fg = lane >> 4, r = xr0 + fr;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t a = *(const bf16x8_t*)&Xsw[r][((ks * 4 + fg) ^ (r & 7)) * 8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      acc[nt] = JAT_MFMA_16x16x32(a, tr_frag(Yt, ks * 32, nt * 16, lane), acc[nt], 0, 0, 0);
  }
}
// Operand fragment of a TRANSPOSED image: element j of lane (fg, fr) = img[k0 + 8 fg + j][c0 + fr], j = 0..7, by two
// ds_read_b64_tr_b16 (each hands a 16-lane group a 4-row x 16-column block column-major; lane 4q+p supplies the
// address of row q, columns 4p..4p+3).  Needs all 64 lanes active.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8_t tr_frag(unsigned short (*img)[AP], int k0, int c0, int lane) {
  const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, pp = fr & 3;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)&img[k0 + 8 * fg + q][c0 + 4 * pp]);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)&img[k0 + 8 * fg + 4 + q][c0 + 4 * pp]);
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, both);
}
// acc[nt] += X[xr0 + m][k] * Yt[k][16 nt + n]      (second operand stored k-major: rows k, columns n)
__device__ __forceinline__ void strip_mma_tr(unsigned short (*X)[AP], int xr0, unsigned short (*Yt)[AP], f32x4_t* acc, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t a = *(const bf16x8_t*)&X[xr0 + fr][ks * 32 + fg * 8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      acc[nt] = JAT_MFMA_16x16x32(a, tr_frag(Yt, ks * 32, nt * 16, lane), acc[nt], 0, 0, 0);
  }
}
// acc[nt] += Xt[k][xc0 + m] * Y[16 nt + n][k]      (first operand stored k-major)
__device__ __forceinline__ void strip_mma_trA(unsigned short (*Xt)[AP], int xc0, unsigned short (*Y)[AP], f32x4_t* acc, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t a = tr_frag(Xt, ks * 32, xc0, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const bf16x8_t bb = *(const bf16x8_t*)&Y[nt * 16 + fr][ks * 32 + fg * 8];
      acc[nt] = JAT_MFMA_16x16x32(a, bb, acc[nt], 0, 0, 0);
    }
  }
}
// store a 16 x 64 fp32 strip (lane: rows 4 fg + r, column 16 nt + fr) as bf16 rows of `dst`, optionally through the
// inverse RoPE rotation of the interleaved pair (2d', 2d'+1): x0' = c x0 + s x1, x1' = c x1 - s x0
__device__ __forceinline__ void store_strip(const f32x4_t* acc, bf16_t* __restrict__ dst, int64_t ld, int row0, int nrows,
                                            int pos0, const float* __restrict__ rc, const float* __restrict__ rs, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = fg * 4 + r;
      float v = acc[nt][r];
      if (rc) {
        const float partner = __shfl_xor(v, 1);
        const int pos = min(pos0 + lr, 2047), dp = (nt * 16 + fr) >> 1;
        const float c = rc[pos * 32 + dp], sn = rs[pos * 32 + dp];
        v = (fr & 1) ? c * v - sn * partner : c * v + sn * partner;
      }
      if (lr < nrows) dst[(int64_t)(row0 + lr) * ld + nt * 16 + fr] = f2bf_t(v);
    }
}

template <bool IDX32>   // every dropout element index of the call fits 32 bits (B*Hq*N*N <= 2^32): no 64-bit index arithmetic per element
__global__ void __launch_bounds__(256, 3) attn_bwd_dkv_kernel(const AttnBwdArgs p) {
  // K [j][d], V^T [d][j] (as stored), Q [i][d], dO [i][d] row-major; P^T, dS^T [j][i].  The products that contract over
  // the ROW index of an image (dP over d of V^T, dV / dK over i of dO / Q) read it with ds_read_b64_tr_b16.
  __shared__ __attribute__((aligned(16))) unsigned short sK[64][64], sVt[64][AP], sQ[64][AP], sDO[64][AP], sPT[64][64],
      sDST[64][64];   // 52 224 B: three blocks per CU (K, P^T, dS^T are only ever read by rows: unpadded + swizzled)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int jb = blockIdx.x, b = blockIdx.z, N = p.N, G = p.Hq / p.Hkv;
  const int g = blockIdx.y / p.hsplit, part = blockIdx.y - g * p.hsplit;
  const int h_lo = part * G / p.hsplit, h_hi = (part + 1) * G / p.hsplit;   // this block's query heads of group g
  const int j0 = jb * 64;
  tile_store_swz(tile_load(p.k + (int64_t)b * N * p.ldk + g * 64, p.ldk, j0, N, tid), sK, tid);
  tile_store(tile_load(p.vt + ((int64_t)(b * p.Hkv + g) * 64) * p.npad + j0, p.npad, 0, 64, tid), sVt, tid);
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) dk[nt] = dv[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nib = (N + 63) / 64, niter = (h_hi - h_lo) * nib;
  const bf16_t* qb = p.q + (int64_t)b * N * p.ldq;
  const bf16_t* dob = p.dout + (int64_t)b * N * p.ldq;
  TileRegs rq = tile_load(qb + (g * G + h_lo) * 64, p.ldq, 0, N, tid), rdo = tile_load(dob + (g * G + h_lo) * 64, p.ldq, 0, N, tid);
  for (int it = 0; it < niter; ++it) {
    const int hh = h_lo + it / nib, ib = it - (it / nib) * nib;
    {
      const int h = g * G + hh;
      const float* lse = p.lse + ((int64_t)b * p.Hq + h) * N;
      const float* dl = p.delta + ((int64_t)b * p.Hq + h) * N;
      const int i0 = ib * 64;
      __syncthreads();   // previous iteration's LDS reads are done (also orders the K / V^T staging before first use)
      tile_store(rq, sQ, tid);
      tile_store(rdo, sDO, tid);
      float l2[4], de[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 16 + fg * 4 + r;
        l2[r] = i < N ? lse[i] : 0.f;
        de[r] = i < N ? dl[i] : 0.f;
      }
      __syncthreads();
      if (it + 1 < niter) {   // next (head, query block): global loads fly under this iteration's MFMAs
        const int hn = h_lo + (it + 1) / nib, ibn = (it + 1) - ((it + 1) / nib) * nib;
        rq = tile_load(qb + (g * G + hn) * 64, p.ldq, ibn * 64, N, tid);
        rdo = tile_load(dob + (g * G + hn) * 64, p.ldq, ibn * 64, N, tid);
      }
      f32x4_t sacc[4], pacc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) sacc[nt] = pacc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      strip_mma_swzY(sQ, wave * 16, sK, sacc, lane);   // S[i][j]  = sum_d Q[i][d] K[j][d]
      strip_mma_tr(sDO, wave * 16, sVt, pacc, lane);   // dP[i][j] = sum_d dO[i][d] V^T[d][j]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = __builtin_amdgcn_exp2f(fmaf(sacc[nt][r], p.scale_log2e, -l2[r]));   // v_exp_f32 directly: no denormal-range fix-up ops
          float dp = pacc[nt][r];
          if (p.drop.thresh) {   // O = (P o m) V:  dV uses P o m,  dP = (dO V^T) o m,  delta = rowsum(dO o O) unchanged
            const int64_t i = i0 + wave * 16 + fg * 4 + r, j = j0 + nt * 16 + fr;
            const float mm = IDX32 ? jat_drop_mult32(p.drop, (uint32_t)((b * p.Hq + h) * N + (int)i) * (uint32_t)N + (uint32_t)j)
                                   : jat_drop_mult(p.drop, (uint64_t)((((int64_t)b * p.Hq + h) * N + i) * N + j));
            dp *= mm;
            ds[r] = pr[r] * (dp - de[r]) * p.scale;
            pr[r] *= mm;
          } else {
            ds[r] = pr[r] * (dp - de[r]) * p.scale;
          }
        }
        u32x2_t a, d;
        a[0] = jat_pack2(pr[0], pr[1]);
        a[1] = jat_pack2(pr[2], pr[3]);
        d[0] = jat_pack2(ds[0], ds[1]);
        d[1] = jat_pack2(ds[2], ds[3]);
        {
          const int row = nt * 16 + fr, col = wave * 16 + fg * 4;   // 8-byte store inside the swizzled 16-byte chunk
          const int sw = (((col >> 3) ^ (row & 7)) << 3) + (col & 7);
          *(u32x2_t*)&sPT[row][sw] = a;    // P^T[j][i]
          *(u32x2_t*)&sDST[row][sw] = d;   // dS^T[j][i]
        }
      }
      __syncthreads();
      strip_mma_tr_swzX(sPT, wave * 16, sDO, dv, lane);   // dV[j][d] += sum_i P^T[j][i] dO[i][d]
      strip_mma_tr_swzX(sDST, wave * 16, sQ, dk, lane);   // dK[j][d] += sum_i dS^T[j][i] Q[i][d]
    }
  }
  const int nrows = min(16, N - (j0 + wave * 16));
  if (p.hsplit > 1) {   // fp32 partial of this block's heads; RoPE^T, the head sum and the bf16 cast happen in the reduce
    float* pb = p.dkv_part + ((int64_t)((b * p.Hkv + g) * p.hsplit + part) * N) * 128;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = fg * 4 + r;
        if (lr < nrows) {
          float* rowp = pb + (int64_t)(j0 + wave * 16 + lr) * 128 + nt * 16 + fr;
          rowp[0] = dk[nt][r];
          rowp[64] = dv[nt][r];
        }
      }
    return;
  }
  bf16_t* base = p.dqkv + (int64_t)b * N * p.ldg;
  store_strip(dk, base + p.D + g * 64, p.ldg, j0 + wave * 16, nrows, j0 + wave * 16, p.rope_cos, p.rope_sin, lane);
  store_strip(dv, base + p.D + p.kvD + g * 64, p.ldg, j0 + wave * 16, nrows, 0, nullptr, nullptr, lane);
}

// dK / dV of (b, g): sum of the hsplit fp32 partials in fixed order, inverse RoPE on dK (interleaved pairs), bf16 into dqkv
__global__ void __launch_bounds__(256) attn_dkv_reduce_kernel(const AttnBwdArgs p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (b, g, j, pair): 64 pairs per row (32 dK + 32 dV)
  const int64_t total = (int64_t)p.B * p.Hkv * p.N * 64;
  if (idx >= total) return;
  const int pr = (int)(idx & 63);
  const int64_t rowi = idx >> 6;                 // (b*Hkv + g)*N + j
  const int j = (int)(rowi % p.N);
  const int64_t bg = rowi / p.N;
  const int g = (int)(bg % p.Hkv), b = (int)(bg / p.Hkv);
  float v0 = 0.f, v1 = 0.f;
  for (int z = 0; z < p.hsplit; ++z) {
    const float* src = p.dkv_part + ((bg * p.hsplit + z) * p.N + j) * 128 + pr * 2;
    v0 += src[0]; v1 += src[1];
  }
  bf16_t* dst = p.dqkv + ((int64_t)b * p.N + j) * p.ldg + p.D;
  if (pr < 32) {   // dK pair (2d', 2d'+1): x0' = c x0 + s x1, x1' = c x1 - s x0
    const float c = p.rope_cos[min(j, 2047) * 32 + pr], sn = p.rope_sin[min(j, 2047) * 32 + pr];
    const float o0 = c * v0 + sn * v1, o1 = c * v1 - sn * v0;
    *(unsigned*)(dst + g * 64 + pr * 2) = jat_pack2(o0, o1);
  } else {
    *(unsigned*)(dst + p.kvD + g * 64 + (pr - 32) * 2) = jat_pack2(v0, v1);
  }
}

template <bool IDX32>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned short sQ[64][AP], sDO[64][AP], sK[64][AP], sVt[64][AP], sDS[64][AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int ib = blockIdx.x, h = blockIdx.y, b = blockIdx.z, N = p.N, g = h / (p.Hq / p.Hkv);
  const int i0 = ib * 64;
  tile_store(tile_load(p.q + (int64_t)b * N * p.ldq + h * 64, p.ldq, i0, N, tid), sQ, tid);
  tile_store(tile_load(p.dout + (int64_t)b * N * p.ldq + h * 64, p.ldq, i0, N, tid), sDO, tid);
  const float* lse = p.lse + ((int64_t)b * p.Hq + h) * N;
  const float* dl = p.delta + ((int64_t)b * p.Hq + h) * N;
  float l2[4], de[4];   // per query column i = i0 + 16 nt + fr of the transposed tiles
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int i = i0 + nt * 16 + fr;
    l2[nt] = i < N ? lse[i] : 0.f;
    de[nt] = i < N ? dl[i] : 0.f;
  }
  f32x4_t dq[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) dq[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int njb = (N + 63) / 64;
  const bf16_t* kb = p.k + (int64_t)b * N * p.ldk + g * 64;
  const bf16_t* vb = p.vt + ((int64_t)(b * p.Hkv + g) * 64) * p.npad;
  TileRegs rk = tile_load(kb, p.ldk, 0, N, tid), rv = tile_load(vb, p.npad, 0, 64, tid);
  for (int jb = 0; jb < njb; ++jb) {
    const int j0 = jb * 64;
    __syncthreads();
    tile_store(rk, sK, tid);
    tile_store(rv, sVt, tid);
    __syncthreads();
    if (jb + 1 < njb) {
      rk = tile_load(kb, p.ldk, j0 + 64, N, tid);
      rv = tile_load(vb + j0 + 64, p.npad, 0, 64, tid);
    }
    f32x4_t sacc[4], pacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) sacc[nt] = pacc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    strip_mma(sK, wave * 16, sQ, sacc, lane);          // S^T[j][i]  = sum_d K[j][d] Q[i][d]
    strip_mma_trA(sVt, wave * 16, sDO, pacc, lane);    // dP^T[j][i] = sum_d V^T[d][j] dO[i][d]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[nt][r], p.scale_log2e, -l2[nt]));
        float dp = pacc[nt][r];
        if (p.drop.thresh) {
          const int64_t i = i0 + nt * 16 + fr, j = j0 + wave * 16 + fg * 4 + r;
          dp *= IDX32 ? jat_drop_mult32(p.drop, (uint32_t)((b * p.Hq + h) * N + (int)i) * (uint32_t)N + (uint32_t)j)
                      : jat_drop_mult(p.drop, (uint64_t)((((int64_t)b * p.Hq + h) * N + i) * N + j));
        }
        ds[r] = pr * (dp - de[nt]) * p.scale;
      }
      u32x2_t d;
      d[0] = jat_pack2(ds[0], ds[1]);
      d[1] = jat_pack2(ds[2], ds[3]);
      *(u32x2_t*)&sDS[nt * 16 + fr][wave * 16 + fg * 4] = d;   // dS[i][j]
    }
    __syncthreads();
    strip_mma_tr(sDS, wave * 16, sK, dq, lane);   // dQ[i][d] += sum_j dS[i][j] K[j][d]
  }
  const int nrows = min(16, N - (i0 + wave * 16));
  store_strip(dq, p.dqkv + (int64_t)b * N * p.ldg + h * 64, p.ldg, i0 + wave * 16, nrows, i0 + wave * 16, p.rope_cos,
              p.rope_sin, lane);
}

// delta[b][h][i] = sum_d dO[i][h*64+d] * O[i][h*64+d]
__global__ void __launch_bounds__(256) attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         int64_t ld, float* __restrict__ delta, int B, int N, int Hq) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (row, head), head fastest
  if (idx >= (int64_t)B * N * Hq) return;
  const int h = (int)(idx % Hq);
  const int64_t row = idx / Hq;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float a[8], d[8];
    unpack8(*(const u32x4_t*)(o + row * ld + h * 64 + c * 8), a);
    unpack8(*(const u32x4_t*)(dout + row * ld + h * 64 + c * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] * d[j];
  }
  const int b = (int)(row / N), i = (int)(row % N);
  delta[((int64_t)b * Hq + h) * N + i] = acc;
}

hipError_t launch_attention_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const bf16_t* o, const bf16_t* dout,
                                const float* lse, float* delta, bf16_t* dqkv, const float* rope_cos, const float* rope_sin,
                                int B, int N, int Hq, int Hkv, int npad, DropSpec drop, float* dkv_part, hipStream_t s) {
  if (Hq % Hkv != 0 || npad % 64 != 0 || npad < N || N > 2048) return hipErrorInvalidValue;
  AttnBwdArgs a;
  a.drop = drop;
  a.q = q; a.k = k; a.vt = vt; a.dout = dout; a.lse = lse; a.delta = delta; a.dqkv = dqkv;
  a.rope_cos = rope_cos; a.rope_sin = rope_sin;
  a.D = Hq * 64; a.kvD = Hkv * 64; a.ldq = a.D; a.ldk = a.kvD; a.ldg = a.D + 2 * a.kvD;
  a.B = B; a.N = N; a.Hq = Hq; a.Hkv = Hkv; a.npad = npad;
  a.scale = 0.125f; a.scale_log2e = 0.125f * 1.4426950408889634f;
  const int64_t nd = (int64_t)B * N * Hq;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, o, dout, (int64_t)a.D, delta, B, N, Hq);
  const int nb = (N + 63) / 64;
  // dkv_part (B*Hq*N*128 floats) given: one query head per block, partials reduced afterwards
  a.dkv_part = dkv_part;
  a.hsplit = (dkv_part && Hq > Hkv) ? Hq / Hkv : 1;
  const bool idx32 = (uint64_t)B * Hq * N * N <= 0xffffffffull;
  if (idx32) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3(nb, Hkv * a.hsplit, B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3(nb, Hkv * a.hsplit, B), dim3(256), 0, s, a);
  if (a.hsplit > 1) {
    const int64_t nr = (int64_t)B * Hkv * N * 64;
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, a);
  }
  if (idx32) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(nb, Hq, B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(nb, Hq, B), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---- loss: mse_loss(pred, target) (mean over all elements) and its gradient ------------------------------------------
// dpred = 2 (pred - target) / n * loss_scale; per-block partial sums of (pred - target)^2, finished in fixed order.
__global__ void __launch_bounds__(256) mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       float* __restrict__ dpred, float* __restrict__ part, int64_t n,
                                                       float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4_t a = *(const f32x4_t*)(pred + i), t = *(const f32x4_t*)(target + i);
      f32x4_t d;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = a[j] - t[j]; acc += e * e; d[j] = e * gscale; }
      *(f32x4_t*)(dpred + i) = d;
    } else {
      for (int64_t k = i; k < n; ++k) { const float e = pred[k] - target[k]; acc += e * e; dpred[k] = e * gscale; }
    }
  }
  acc = wave_sum_t(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out[0] = scale * sum(part[0..n)); out[1] = sqrt(sum) (used by the gradient norm)
__global__ void __launch_bounds__(256) finish_sum_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = (float)(red[0] * (double)scale); out[1] = (float)sqrt(red[0] * (double)scale); }
}
// ---- loss: Charbonnier, mean(sqrt((pred - target)^2 + eps)) (train_ddp_v3m2mod1.py:72-101: eps is ADDED to the squared
// difference, not squared) and its gradient  dpred = (pred - target) / sqrt((pred - target)^2 + eps) / n * loss_scale.
__global__ void __launch_bounds__(256) charbonnier_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                               float* __restrict__ dpred, float* __restrict__ part, int64_t n,
                                                               float eps, float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4_t a = *(const f32x4_t*)(pred + i), t = *(const f32x4_t*)(target + i);
      f32x4_t d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = a[j] - t[j], r = sqrtf(e * e + eps);
        acc += r;
        d[j] = e / r * gscale;
      }
      *(f32x4_t*)(dpred + i) = d;
    } else {
      for (int64_t k = i; k < n; ++k) {
        const float e = pred[k] - target[k], r = sqrtf(e * e + eps);
        acc += r;
        dpred[k] = e / r * gscale;
      }
    }
  }
  acc = wave_sum_t(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
constexpr int RED_BLOCKS = 1024;
int train_red_blocks() { return RED_BLOCKS; }
hipError_t launch_charbonnier_grad(const float* pred, const float* target, float* dpred, float* part, float* loss2, int64_t n,
                                   float eps, float loss_scale, hipStream_t s);
hipError_t launch_mse_grad(const float* pred, const float* target, float* dpred, float* part, float* loss2, int64_t n,
                           float loss_scale, hipStream_t s) {
  hipLaunchKernelGGL(mse_grad_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, pred, target, dpred, part, n,
                     2.0f * loss_scale / (float)n);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, part, RED_BLOCKS, 1.0f / (float)n, loss2);
  return hipGetLastError();
}

hipError_t launch_charbonnier_grad(const float* pred, const float* target, float* dpred, float* part, float* loss2, int64_t n,
                                   float eps, float loss_scale, hipStream_t s) {
  hipLaunchKernelGGL(charbonnier_grad_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, pred, target, dpred, part, n, eps,
                     loss_scale / (float)n);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, part, RED_BLOCKS, 1.0f / (float)n, loss2);
  return hipGetLastError();
}

// ---- global gradient norm + clip + AdamW (torch.nn.utils.clip_grad_norm_, torch.optim.AdamW semantics) ---------------
__global__ void __launch_bounds__(256) sqsum_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {   // n % 4 == 0
    const f32x4_t a = *(const f32x4_t*)(g + i);
    acc += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  acc = wave_sum_t(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm2[0] = sum g^2 of the SCALED gradients, norm2[1] = its square root.  inv_scale undoes the loss scaling.
// A non-finite norm skips the update (GradScaler.step semantics, train_ddp_v3m2.py:618).
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ norm2,
                                                    float inv_scale, float max_norm, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt) {
  const float total = norm2[1] * inv_scale;
  if (!(total == total) || total > 3.0e38f) return;
  float coef = inv_scale;
  if (max_norm > 0.f) coef *= fminf(1.0f, max_norm / (total + 1e-6f));
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    f32x4_t pp = *(const f32x4_t*)(p + i), gg = *(const f32x4_t*)(g + i), mm = *(const f32x4_t*)(m + i), vv = *(const f32x4_t*)(v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * coef;   // unscaled, clipped gradient (not written back: nothing reads it after the step)
      pp[j] *= 1.0f - lr * wd;
      mm[j] = beta1 * mm[j] + (1.0f - beta1) * gr;
      vv[j] = beta2 * vv[j] + (1.0f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] -= (lr / bc1) * (mm[j] / denom);
    }
    *(f32x4_t*)(p + i) = pp; *(f32x4_t*)(m + i) = mm; *(f32x4_t*)(v + i) = vv;
  }
}
hipError_t launch_grad_sqsum(const float* g, int64_t n, float* part, float* norm2, hipStream_t s) {
  if (n % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sqsum_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, g, n, part);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, part, RED_BLOCKS, 1.0f, norm2);
  return hipGetLastError();
}
hipError_t launch_adamw(float* p, float* g, float* m, float* v, int64_t n, const float* norm2, float inv_scale,
                        float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s) {
  if (n % 4 != 0 || step < 1) return hipErrorInvalidValue;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  // one float4 of each of the four streams per thread, no grid-stride loop: 2048 looping blocks ran this 21 GB pass at 4.7 TB/s,
  // the one-pass grid is 0.4-0.9 ms faster per step on the boxes measured (profiles/r03/adamw_grid_sweep.log); JAT_ADAMW_BLOCKS: A/B
  static const int blocks_env = getenv("JAT_ADAMW_BLOCKS") ? atoi(getenv("JAT_ADAMW_BLOCKS")) : 0;
  const int64_t full = (n / 4 + 255) / 256;
  const unsigned blocks = blocks_env > 0 ? (unsigned)blocks_env : (unsigned)(full < 1 ? 1 : full);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, norm2, inv_scale, max_norm, lr, beta1, beta2,
                     eps, wd, (float)bc1, (float)sqrt(bc2));
  return hipGetLastError();
}

// ---- small-batch Linear backward (adaLN modulation and t_embedder: B <= 32 rows, weights up to [215040, 1280]) --------
// Both are HBM-bound streams over the weight-sized operand: dW is written once (fp32), W is read once (bf16 or fp32).
// dW[n][k] = sum_b dy[b][n] * x[b][k]   (optionally x -> silu(x));   db[n] = sum_b dy[b][n]
// One block = DW_R output rows; a thread owns 4 consecutive k (float4 stores) and keeps DW_R x 4 accumulators.
constexpr int DW_R = 16;
__global__ void __launch_bounds__(320) small_dw_kernel(const float* __restrict__ dy, int64_t ldy, const float* __restrict__ x,
                                                       int64_t ldx, float* __restrict__ dW, float* __restrict__ db, int B,
                                                       int N, int K, int silu_x) {
  __shared__ __attribute__((aligned(16))) float sdy[64][DW_R];   // this block's dy rows, b-major: broadcast reads in the loop
  const int n0 = blockIdx.x * DW_R;
  for (int i = threadIdx.x; i < B * DW_R; i += blockDim.x) {
    const int b = i / DW_R, r = i - b * DW_R;
    sdy[b][r] = (n0 + r < N) ? dy[(int64_t)b * ldy + n0 + r] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x * 4; k < K; k += blockDim.x * 4) {
    f32x4_t acc[DW_R];
#pragma unroll
    for (int r = 0; r < DW_R; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
      f32x4_t xv = *(const f32x4_t*)(x + (int64_t)b * ldx + k);
      if (silu_x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xv[j] / (1.0f + __expf(-xv[j]));
      }
#pragma unroll
      for (int q = 0; q < DW_R / 4; ++q) {
        const f32x4_t d = *(const f32x4_t*)&sdy[b][q * 4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q * 4 + e][j] += d[e] * xv[j];
      }
    }
#pragma unroll
    for (int r = 0; r < DW_R; ++r)
      if (n0 + r < N) *(f32x4_t*)(dW + (int64_t)(n0 + r) * K + k) = acc[r];
  }
  if (db && threadIdx.x < DW_R && n0 + threadIdx.x < N) {
    float bsum = 0.f;
    for (int b = 0; b < B; ++b) bsum += sdy[b][threadIdx.x];
    db[n0 + threadIdx.x] = bsum;
  }
}
// partial dx: part[slab][b][k] = sum_{n in slab} dy[b][n] * W[n][k]; W bf16 (packed operand copy) or fp32.
// The slab's dy values sit in LDS as [n][32 b]; a thread owns 4 consecutive k and 32 x 4 accumulators (B <= 32).
template <typename WT>
__global__ void __launch_bounds__(320) small_dx_kernel(const float* __restrict__ dy, int64_t ldy, const WT* __restrict__ W,
                                                       float* __restrict__ part, int B, int N, int K, int slab) {
  extern __shared__ float sdy[];   // [slab][32]
  const int n0 = blockIdx.x * slab, n1 = min(n0 + slab, N);
  for (int i = threadIdx.x; i < slab * 32; i += blockDim.x) {
    const int n = n0 + (i >> 5), b = i & 31;
    sdy[i] = (n < N && b < B) ? dy[(int64_t)b * ldy + n] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x * 4; k < K; k += blockDim.x * 4) {
    f32x4_t acc[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
      f32x4_t w;
      if constexpr (sizeof(WT) == 2) {
        const u32x2_t ww = *(const u32x2_t*)(W + (int64_t)n * K + k);
        w[0] = jat_lo2f(ww[0]); w[1] = jat_hi2f(ww[0]);
        w[2] = jat_lo2f(ww[1]); w[3] = jat_hi2f(ww[1]);
      } else {
        w = *(const f32x4_t*)(W + (int64_t)n * K + k);
      }
      const f32x4_t* row = (const f32x4_t*)(sdy + (n - n0) * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4_t d = row[q];   // broadcast read: every lane the same address
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q * 4 + e][j] += d[e] * w[j];
      }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b)
      if (b < B) *(f32x4_t*)(part + ((int64_t)blockIdx.x * B + b) * K + k) = acc[b];
  }
}
// dx[b][k] (+)= sum_split part[split][b][k], optionally times silu'(pre[b][k])
__global__ void __launch_bounds__(256) small_dx_finish_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ dx,
                                                              int64_t BK, int accumulate, const float* __restrict__ silu_pre) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BK) return;
  float acc = 0.f;
  for (int sidx = 0; sidx < nsplit; ++sidx) acc += part[(int64_t)sidx * BK + i];
  if (accumulate) acc += dx[i];
  if (silu_pre) {
    const float u = silu_pre[i], sg = 1.0f / (1.0f + __expf(-u));
    acc *= sg * (1.0f + u * (1.0f - sg));
  }
  dx[i] = acc;
}
hipError_t launch_small_dw(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dW, float* db, int B, int N,
                           int K, int silu_x, hipStream_t s) {
  if (K % 4 != 0 || B > 64) return hipErrorInvalidValue;
  const int threads = K / 4 >= 320 ? 320 : 64 * ((K / 4 + 63) / 64);   // one pass over K for K = 1280
  hipLaunchKernelGGL(small_dw_kernel, dim3((N + DW_R - 1) / DW_R), dim3(threads), 0, s, dy, ldy, x, ldx, dW, db, B, N, K, silu_x);
  return hipGetLastError();
}
int small_dx_slab(int N) { return N >= 16384 ? 256 : 32; }
// part must hold ceil(N / small_dx_slab(N)) * B * K floats.  w_is_bf16: W is the packed bf16 operand copy.
hipError_t launch_small_dx(const float* dy, int64_t ldy, const void* W, int w_is_bf16, float* part, float* dx, int B, int N,
                           int K, int accumulate, const float* silu_pre, hipStream_t s) {
  if (B > 32 || K % 4 != 0) return hipErrorInvalidValue;
  const int slab = small_dx_slab(N), nsplit = (N + slab - 1) / slab;
  const int threads = K / 4 >= 320 ? 320 : 64 * ((K / 4 + 63) / 64);
  if (w_is_bf16)
    hipLaunchKernelGGL(small_dx_kernel<bf16_t>, dim3(nsplit), dim3(threads), (size_t)slab * 32 * 4, s, dy, ldy, (const bf16_t*)W,
                       part, B, N, K, slab);
  else
    hipLaunchKernelGGL(small_dx_kernel<float>, dim3(nsplit), dim3(threads), (size_t)slab * 32 * 4, s, dy, ldy, (const float*)W,
                       part, B, N, K, slab);
  const int64_t BK = (int64_t)B * K;
  hipLaunchKernelGGL(small_dx_finish_kernel, dim3((unsigned)((BK + 255) / 256)), dim3(256), 0, s, part, nsplit, dx, BK,
                     accumulate, silu_pre);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) silu_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float u = in[i]; out[i] = u / (1.0f + __expf(-u)); }
}
hipError_t launch_silu_f32(const float* in, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

// ---- gradient of the fused, pair-interleaved [Wq; Wk; Wv] back to the reference's three tensors -------------------------
// fused row (h*64 + 2d + e) of the q / k part holds reference row (h*64 + d + 32e)   (elementwise.hip cast_bf16_rope_rows)
__global__ void __launch_bounds__(256) unpack_qkv_grad_kernel(const float* __restrict__ fused, float* __restrict__ gq,
                                                              float* __restrict__ gk, float* __restrict__ gv, int D, int kvD,
                                                              int K) {
  const int r = blockIdx.x;   // fused row
  float* dst;
  if (r < D + kvD) {
    const int base = r < D ? 0 : D, lr = r - base, h = lr >> 6, w = lr & 63;
    const int ref = h * 64 + (w >> 1) + 32 * (w & 1);
    dst = (r < D ? gq : gk) + (int64_t)ref * K;
  } else {
    dst = gv + (int64_t)(r - D - kvD) * K;
  }
  for (int k = threadIdx.x * 4; k < K; k += 1024) *(f32x4_t*)(dst + k) = *(const f32x4_t*)(fused + (int64_t)r * K + k);
}
hipError_t launch_unpack_qkv_grad(const float* fused, float* gq, float* gk, float* gv, int D, int kvD, int K, hipStream_t s) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(unpack_qkv_grad_kernel, dim3(D + 2 * kvD), dim3(256), 0, s, fused, gq, gk, gv, D, kvD, K);
  return hipGetLastError();
}

// ---- data preparation of the step (train_ddp_v3m2.py:548-579) -----------------------------------------------------------
// z_t = t[b] * x + (1 - t[b]) * noise
__global__ void __launch_bounds__(256) flow_mix_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                       const float* __restrict__ t, float* __restrict__ z, int64_t per_sample,
                                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float tv = t[i / per_sample];
  z[i] = tv * x[i] + (1.0f - tv) * noise[i];
}
hipError_t launch_flow_mix(const float* x, const float* noise, const float* t, float* z, int B, int64_t per_sample, hipStream_t s) {
  const int64_t n = (int64_t)B * per_sample;
  hipLaunchKernelGGL(flow_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, noise, t, z, per_sample, n);
  return hipGetLastError();
}
// cond = (cond + noise * (ratio * clamp(std, 0.5, 2))) * keep[b];  std2[1] = unbiased std of cond (device scalar pair)
__global__ void __launch_bounds__(256) cond_augment_kernel(float* __restrict__ cond, const float* __restrict__ noise,
                                                           const float* __restrict__ std2, float ratio,
                                                           const float* __restrict__ keep, int64_t per_sample, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = cond[i];
  if (noise) v += noise[i] * (ratio * (std2 ? fminf(fmaxf(std2[1], 0.5f), 2.0f) : 1.0f));
  cond[i] = v * (keep ? keep[i / per_sample] : 1.0f);
}
hipError_t launch_cond_augment(float* cond, const float* noise, const float* std2, float ratio, const float* keep, int B,
                               int64_t per_sample, hipStream_t s) {
  const int64_t n = (int64_t)B * per_sample;
  hipLaunchKernelGGL(cond_augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cond, noise, std2, ratio, keep,
                     per_sample, n);
  return hipGetLastError();
}
// unbiased standard deviation of a tensor (torch.Tensor.std()): two fixed-order passes (mean, then centred squares)
__global__ void __launch_bounds__(256) moment_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ mean2,
                                                     float* __restrict__ part) {
  __shared__ float red[4];
  const float mu = mean2 ? mean2[0] : 0.f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float e = x[i] - mu;
    acc += mean2 ? e * e : e;
  }
  acc = wave_sum_t(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out2[0] = variance (unbiased), out2[1] = std; mean2 is scratch (2 floats)
hipError_t launch_tensor_std(const float* x, int64_t n, float* part, float* mean2, float* out2, hipStream_t s) {
  if (n < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moment_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, x, n, (const float*)nullptr, part);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, part, RED_BLOCKS, 1.0f / (float)n, mean2);
  hipLaunchKernelGGL(moment_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, x, n, (const float*)mean2, part);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, part, RED_BLOCKS, 1.0f / (float)(n - 1), out2);
  return hipGetLastError();
}

// ---- table-driven copies: the fp32 tensors the kernels read directly (biases, norm weights, t_embedder) go from the
// flat master buffer to the model's packed blob in ONE launch after every optimiser step ---------------------------------
__global__ void __launch_bounds__(256) multi_copy_kernel(const CopyJob* __restrict__ jobs) {
  const CopyJob j = jobs[blockIdx.y];
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < j.n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < j.n) *(f32x4_t*)(j.dst + i) = *(const f32x4_t*)(j.src + i);   // tensors start 16-B aligned on both sides
    else for (int64_t k = i; k < j.n; ++k) j.dst[k] = j.src[k];
  }
}
hipError_t launch_multi_copy(const CopyJob* jobs_dev, int njobs, hipStream_t s) {
  if (njobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(32, njobs), dim3(256), 0, s, jobs_dev);
  return hipGetLastError();
}

// ---- v3mod2 training loss: MSE + latent perceptual loss (train_ddp_v3mod2.py:53-321, 889-896) ------------------------------
//   loss = mse + lw * (fw * freq + mw * ms + cw * cons)
//   freq = mean|log(|P|+1e-7) - log(|H|+1e-7)| + 0.1 * mean_{k<low}|P - H|                 (:97-123)   P, H, R = rfft over T
//   ms   = (mean|p-h| + mean|pool2(p)-pool2(h)| + mean|pool4(p)-pool4(h)|) / 3             (:158-171)  of pred, target, clean LR
//   cons = mean_{k<strict}|P - R| + mean_{strict<=k<soft} w_k ||P| - |R||, w = linspace(1,0) (:229-262)
// One block per (b, c) row.  T = 1378 = 2*13*53 is not FFT-friendly and the whole loss is < 2 % of the step's FLOPs, so
// the spectra are direct fp32 DFTs against an exact twiddle table (cos, sin of 2 pi m / T, m < T, index k*n mod T kept
// incrementally), and the gradient is the adjoint sum  dp_n = Re sum_k g_k e^{+i 2 pi k n / T}  over the rfft bins.
// Bound: fp32 VALU (8 FMA per (k, n) pair).  Loss terms: per-row partials, finished in fixed order.
struct LatentLossArgs {
  const float *pred, *target, *lr;   // [rows, T]; lr = clean normalised condition (may be null when cw == 0)
  const float2* tw;                  // [T] (cos, sin)(2 pi m / T)
  float* dpred;                      // [rows, T]   d(total loss * gscale)/d pred
  float* part;                       // [rows, 8] partial sums: se, logmag, low, l1, l1p2, l1p4, strict, trans
  int rows, T, F, low, strict, soft;
  float lw, fw, mw, cw, gscale;      // weights; gscale = loss_scale
  float inv_n;                       // 1 / (rows * T)
};
template <int FB, int NB>   // bins / samples per thread of the blocked DFT loops (the launcher picks the smallest cover)
__global__ void __launch_bounds__(256) latent_loss_kernel(const LatentLossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, F = a.F, tid = threadIdx.x;
  constexpr int RS = 8;    // twiddle re-seed interval (see the forward loop)
  float* sp = sm;                 // pred row
  float* sh = sp + T;             // target row
  float* sr = sh + T;             // clean LR row
  float2* stw = (float2*)(sr + T);   // twiddles
  float2* sg = stw + T;           // spectral gradient g_k
  float* red = (float*)(sg + F);  // [4 waves][8]
  const int64_t row = blockIdx.x;
  const bool use_lr = a.lr != nullptr && a.cw != 0.f;
  for (int n = tid; n < T; n += 256) {
    sp[n] = a.pred[row * T + n];
    sh[n] = a.target[row * T + n];
    sr[n] = use_lr ? a.lr[row * T + n] : 0.f;
    stw[n] = a.tw[n];
  }
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float rows_f = (float)a.rows;
  // mean normalisers of the spectral terms (per element); guarded against empty bands
  const float n_log = 1.0f / (rows_f * F), n_low = a.low > 0 ? 1.0f / (rows_f * a.low) : 0.f;
  const float n_str = a.strict > 0 ? 1.0f / (rows_f * a.strict) : 0.f;
  const int bw = a.soft - a.strict;
  const float n_tr = bw > 0 ? 1.0f / (rows_f * bw) : 0.f;
  // forward DFTs: a thread owns up to FB bins (k = tid + 256 j) and walks n once for all of them, so the three row
  // samples are read from LDS once per n instead of once per (bin, n)
  for (int kbase = 0; kbase < F; kbase += 256 * FB) {
  float spa[FB], spb[FB], sha[FB], shb[FB], sra[FB], srb[FB];
  int kk[FB], idxs[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    spa[j] = spb[j] = sha[j] = shb[j] = sra[j] = srb[j] = 0.f;
    kk[j] = kbase + tid + 256 * j;
    if (kk[j] >= F) kk[j] = 0;   // idle slot: computes bin 0 again, result unused
    idxs[j] = 0;
  }
  // twiddle e^{-i theta_kn} by rotation with the bin's step e^{-i theta_k}, re-seeded from the exact table every RS
  // samples (error growth RS * 2^-24): four FMAs instead of a bank-conflicted LDS gather per (bin, sample)
  float ck[FB], sk[FB];
  int step16[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    const float2 w = stw[kk[j]];
    ck[j] = w.x; sk[j] = w.y;
    step16[j] = (int)(((long long)kk[j] * RS) % T);
  }
  for (int nb = 0; nb < T; nb += RS) {
    float c[FB], sn[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j) { const float2 w = stw[idxs[j]]; c[j] = w.x; sn[j] = w.y; }
    const int nend = min(RS, T - nb);
    for (int u = 0; u < nend; ++u) {
      const float x = sp[nb + u], y = sh[nb + u], zz = sr[nb + u];
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        spa[j] += x * c[j]; spb[j] -= x * sn[j];
        sha[j] += y * c[j]; shb[j] -= y * sn[j];
        sra[j] += zz * c[j]; srb[j] -= zz * sn[j];
        const float cn = c[j] * ck[j] - sn[j] * sk[j];
        sn[j] = sn[j] * ck[j] + c[j] * sk[j];
        c[j] = cn;
      }
    }
#pragma unroll
    for (int j = 0; j < FB; ++j) { idxs[j] += step16[j]; if (idxs[j] >= T) idxs[j] -= T; }
  }
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    const int k = kbase + tid + 256 * j;
    if (k >= F) continue;
    const float pa = spa[j], pb = spb[j], ha = sha[j], hb = shb[j], ra = sra[j], rb = srb[j];
    float ga = 0.f, gb = 0.f;
    const float pm = sqrtf(pa * pa + pb * pb), hm = sqrtf(ha * ha + hb * hb);
    {   // log-magnitude L1
      const float d = __logf(pm + 1e-7f) - __logf(hm + 1e-7f);
      acc[1] += fabsf(d);
      if (pm > 0.f && d != 0.f) {
        const float c = a.fw * (d > 0.f ? 1.f : -1.f) / (pm + 1e-7f) * n_log / pm;
        ga += c * pa; gb += c * pb;
      }
    }
    if (k < a.low) {   // low-frequency complex L1 (weight 0.1 inside freq)
      const float da = pa - ha, db = pb - hb, m = sqrtf(da * da + db * db);
      acc[2] += m;
      if (m > 0.f) { const float c = a.fw * 0.1f * n_low / m; ga += c * da; gb += c * db; }
    }
    if (use_lr) {
      if (k < a.strict) {
        const float da = pa - ra, db = pb - rb, m = sqrtf(da * da + db * db);
        acc[6] += m;
        if (m > 0.f) { const float c = a.cw * n_str / m; ga += c * da; gb += c * db; }
      } else if (k < a.soft) {
        const float wk = bw > 1 ? 1.0f - (float)(k - a.strict) / (float)(bw - 1) : 1.0f;   // torch.linspace(1, 0, bw)
        const float rm = sqrtf(ra * ra + rb * rb), d = pm - rm;
        acc[7] += wk * fabsf(d);
        if (pm > 0.f && d != 0.f) { const float c = a.cw * wk * (d > 0.f ? 1.f : -1.f) * n_tr / pm; ga += c * pa; gb += c * pb; }
      }
    }
    sg[k] = float2{ga, gb};
  }
  }
  __syncthreads();
  const int T2 = T / 2, T4 = T / 4;
  const float lam = a.lw * a.gscale;
  for (int nbase = 0; nbase < T; nbase += 256 * NB) {
  // adjoint DFT: s_n = sum_k ga cos(theta) - gb sin(theta), theta = 2 pi k n / T; g_k is read once per k for NB samples
  float sv[NB];
  int nn[NB], idn[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    sv[j] = 0.f;
    nn[j] = nbase + tid + 256 * j;
    if (nn[j] >= T) nn[j] = 0;
    idn[j] = 0;
  }
  float cnn[NB], snn[NB];
  int stepn[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float2 w = stw[nn[j]];
    cnn[j] = w.x; snn[j] = w.y;
    stepn[j] = (int)(((long long)nn[j] * RS) % T);
  }
  for (int kb = 0; kb < F; kb += RS) {
    float c[NB], sn[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { const float2 w = stw[idn[j]]; c[j] = w.x; sn[j] = w.y; }
    const int kend = min(RS, F - kb);
    for (int u = 0; u < kend; ++u) {
      const float2 g = sg[kb + u];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        sv[j] += g.x * c[j] - g.y * sn[j];
        const float cn = c[j] * cnn[j] - sn[j] * snn[j];
        sn[j] = sn[j] * cnn[j] + c[j] * snn[j];
        c[j] = cn;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) { idn[j] += stepn[j]; if (idn[j] >= T) idn[j] -= T; }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = nbase + tid + 256 * j;
    if (n >= T) continue;
    const float s = sv[j];
    const float e = sp[n] - sh[n];
    acc[0] += e * e;
    acc[3] += fabsf(e);
    float gt = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * a.inv_n;   // d mean|p-h|
    if (n < 2 * T2) {
      const int j = n >> 1;
      const float q = 0.5f * ((sp[2 * j] - sh[2 * j]) + (sp[2 * j + 1] - sh[2 * j + 1]));
      if ((n & 1) == 0) acc[4] += fabsf(q);
      gt += (q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f)) * 0.5f / (rows_f * T2);
    }
    if (n < 4 * T4) {
      const int j = n >> 2;
      float q = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) q += sp[4 * j + u] - sh[4 * j + u];
      q *= 0.25f;
      if ((n & 3) == 0) acc[5] += fabsf(q);
      gt += (q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f)) * 0.25f / (rows_f * T4);
    }
    a.dpred[row * T + n] = a.gscale * 2.0f * e * a.inv_n + lam * (s + a.mw * gt * (1.0f / 3.0f));
  }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = wave_sum_t(acc[i]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(tid >> 6) * 8 + i] = acc[i];
  __syncthreads();
  if (tid < 8) a.part[row * 8 + tid] = red[tid] + red[8 + tid] + red[16 + tid] + red[24 + tid];
}
// ---- the same loss with the DFTs factored T = N1 * N2 (Cooley-Tukey, two stages, any factor pair) -------------------------
// forward, n = N2 n1 + n2:   Y[k1][n2] = sum_n1 x[N2 n1 + n2] W^(k1 N2 n1)      (k1 <= N1/2; Y[N1-k1] = conj Y[k1], x real)
//                            X[k]      = sum_n2 Y[k mod N1][n2] W^(k n2)          (k < F)           W = e^{-2 pi i / T}
// adjoint:                   Z[k1][n2] = sum_{k = k1 (mod N1), k < F} g_k W^(-k n2)
//                            s[N2 n1 + n2] = Re sum_k1 Z[k1][n2] W^(-k1 N2 n1)
// T (N1 + N2) complex MACs per transform instead of T^2 / 2: 10 x fewer at T = 1378 = 26 * 53.  All twiddles come from the
// one exact table W^m (m < T) with incrementally reduced indices.
struct SpecNorm { float n_log, n_low, n_str, n_tr; int bw; bool use_lr; };
__device__ __forceinline__ float2 spectral_terms(const LatentLossArgs& a, const SpecNorm& nm, int k, float pa, float pb, float ha,
                                                 float hb, float ra, float rb, float* acc) {
  float ga = 0.f, gb = 0.f;
  const float pm = sqrtf(pa * pa + pb * pb), hm = sqrtf(ha * ha + hb * hb);
  {
    const float d = __logf(pm + 1e-7f) - __logf(hm + 1e-7f);
    acc[1] += fabsf(d);
    if (pm > 0.f && d != 0.f) {
      const float c = a.fw * (d > 0.f ? 1.f : -1.f) / (pm + 1e-7f) * nm.n_log / pm;
      ga += c * pa; gb += c * pb;
    }
  }
  if (k < a.low) {
    const float da = pa - ha, db = pb - hb, m = sqrtf(da * da + db * db);
    acc[2] += m;
    if (m > 0.f) { const float c = a.fw * 0.1f * nm.n_low / m; ga += c * da; gb += c * db; }
  }
  if (nm.use_lr) {
    if (k < a.strict) {
      const float da = pa - ra, db = pb - rb, m = sqrtf(da * da + db * db);
      acc[6] += m;
      if (m > 0.f) { const float c = a.cw * nm.n_str / m; ga += c * da; gb += c * db; }
    } else if (k < a.soft) {
      const float wk = nm.bw > 1 ? 1.0f - (float)(k - a.strict) / (float)(nm.bw - 1) : 1.0f;
      const float rm = sqrtf(ra * ra + rb * rb), d = pm - rm;
      acc[7] += wk * fabsf(d);
      if (pm > 0.f && d != 0.f) { const float c = a.cw * wk * (d > 0.f ? 1.f : -1.f) * nm.n_tr / pm; ga += c * pa; gb += c * pb; }
    }
  }
  return float2{ga, gb};
}
// time-domain terms of sample n (MSE, multi-scale L1) and the output gradient
__device__ __forceinline__ void time_terms(const LatentLossArgs& a, const float* sp, const float* sh, int n, float s, int64_t row,
                                           float* acc) {
  const int T = a.T, T2 = T / 2, T4 = T / 4;
  const float rows_f = (float)a.rows;
  const float e = sp[n] - sh[n];
  acc[0] += e * e;
  acc[3] += fabsf(e);
  float gt = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * a.inv_n;
  if (n < 2 * T2) {
    const int j = n >> 1;
    const float q = 0.5f * ((sp[2 * j] - sh[2 * j]) + (sp[2 * j + 1] - sh[2 * j + 1]));
    if ((n & 1) == 0) acc[4] += fabsf(q);
    gt += (q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f)) * 0.5f / (rows_f * T2);
  }
  if (n < 4 * T4) {
    const int j = n >> 2;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) q += sp[4 * j + u] - sh[4 * j + u];
    q *= 0.25f;
    if ((n & 3) == 0) acc[5] += fabsf(q);
    gt += (q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f)) * 0.25f / (rows_f * T4);
  }
  a.dpred[row * T + n] = a.gscale * 2.0f * e * a.inv_n + a.lw * a.gscale * (s + a.mw * gt * (1.0f / 3.0f));
}
__global__ void __launch_bounds__(256) latent_loss_fft_kernel(const LatentLossArgs a, int N1, int N2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, F = a.F, tid = threadIdx.x, N1h = N1 / 2 + 1;
  float* sp = sm;
  float* sh = sp + T;
  float* sr = sh + T;
  float2* stw = (float2*)(sr + T);
  float2* sg = stw + T;                 // [F]
  float2* sY = sg + F;                  // [3][N1h][N2]; later Z [N1][N2]
  float* red = (float*)(sY + max(3 * N1h * N2, T));
  const int64_t row = blockIdx.x;
  SpecNorm nm;
  nm.use_lr = a.lr != nullptr && a.cw != 0.f;
  for (int n = tid; n < T; n += 256) {
    sp[n] = a.pred[row * T + n];
    sh[n] = a.target[row * T + n];
    sr[n] = nm.use_lr ? a.lr[row * T + n] : 0.f;
    stw[n] = a.tw[n];
  }
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float rows_f = (float)a.rows;
  nm.n_log = 1.0f / (rows_f * F);
  nm.n_low = a.low > 0 ? 1.0f / (rows_f * a.low) : 0.f;
  nm.n_str = a.strict > 0 ? 1.0f / (rows_f * a.strict) : 0.f;
  nm.bw = a.soft - a.strict;
  nm.n_tr = nm.bw > 0 ? 1.0f / (rows_f * nm.bw) : 0.f;
  __syncthreads();
  // stage 1: N1-point DFTs over the strided samples, three signals at once
  const int plane = N1h * N2;
  for (int o = tid; o < plane; o += 256) {
    const int k1 = o / N2, n2 = o - k1 * N2;
    const int step = (k1 * N2) % T;
    float pa = 0.f, pb = 0.f, ha = 0.f, hb = 0.f, ra = 0.f, rb = 0.f;
    int idx = 0;
#pragma unroll 4
    for (int n1 = 0; n1 < N1; ++n1) {
      const float2 w = stw[idx];
      const int n = N2 * n1 + n2;
      const float x = sp[n], y = sh[n], z = sr[n];
      pa += x * w.x; pb -= x * w.y;
      ha += y * w.x; hb -= y * w.y;
      ra += z * w.x; rb -= z * w.y;
      idx += step; if (idx >= T) idx -= T;
    }
    sY[o] = float2{pa, pb};
    sY[plane + o] = float2{ha, hb};
    sY[2 * plane + o] = float2{ra, rb};
  }
  __syncthreads();
  // stage 2: X[k] = sum_n2 Y[k mod N1][n2] W^(k n2), loss terms, spectral gradient
  for (int k = tid; k < F; k += 256) {
    int k1 = k % N1;
    const bool cj = k1 > N1 / 2;      // Y[k1] = conj(Y[N1 - k1])
    if (cj) k1 = N1 - k1;
    const float sgn = cj ? -1.f : 1.f;
    const float2* yp = sY + k1 * N2;
    float pa = 0.f, pb = 0.f, ha = 0.f, hb = 0.f, ra = 0.f, rb = 0.f;
    int idx = 0;
#pragma unroll 4
    for (int n2 = 0; n2 < N2; ++n2) {
      const float2 w = stw[idx];       // W^m = (cos, -sin): (yr + i yi)(c - i s) = (yr c + yi s) + i (yi c - yr s)
      float2 y = yp[n2];
      y.y *= sgn;
      pa += y.x * w.x + y.y * w.y; pb += y.y * w.x - y.x * w.y;
      y = yp[plane + n2]; y.y *= sgn;
      ha += y.x * w.x + y.y * w.y; hb += y.y * w.x - y.x * w.y;
      y = yp[2 * plane + n2]; y.y *= sgn;
      ra += y.x * w.x + y.y * w.y; rb += y.y * w.x - y.x * w.y;
      idx += k; if (idx >= T) idx -= T;
    }
    sg[k] = spectral_terms(a, nm, k, pa, pb, ha, hb, ra, rb, acc);
  }
  __syncthreads();
  // stage 3: Z[k1][n2] = sum_{k = k1 (mod N1), k < F} g_k W^(-k n2)      (overwrites Y)
  float2* sZ = sY;
  for (int o = tid; o < T; o += 256) {
    const int k1 = o / N2, n2 = o - k1 * N2;
    int idx = (k1 * n2) % T;
    const int step = (int)(((long long)N1 * n2) % T);
    float zr = 0.f, zi = 0.f;
#pragma unroll 4
    for (int k = k1; k < F; k += N1) {
      const float2 w = stw[idx], g = sg[k];   // W^-m = (cos, +sin)
      zr += g.x * w.x - g.y * w.y;
      zi += g.x * w.y + g.y * w.x;
      idx += step; if (idx >= T) idx -= T;
    }
    sZ[o] = float2{zr, zi};
  }
  __syncthreads();
  // stage 4: s[N2 n1 + n2] = Re sum_k1 Z[k1][n2] W^(-k1 N2 n1), then the time-domain terms and the output
  for (int n = tid; n < T; n += 256) {
    const int n1 = n / N2, n2 = n - n1 * N2;
    const int step = (N2 * n1) % T;
    float sv = 0.f;
    int idx = 0;
#pragma unroll 4
    for (int k1 = 0; k1 < N1; ++k1) {
      const float2 w = stw[idx], z = sZ[k1 * N2 + n2];
      sv += z.x * w.x - z.y * w.y;
      idx += step; if (idx >= T) idx -= T;
    }
    time_terms(a, sp, sh, n, sv, row, acc);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = wave_sum_t(acc[i]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(tid >> 6) * 8 + i] = acc[i];
  __syncthreads();
  if (tid < 8) a.part[row * 8 + tid] = red[tid] + red[8 + tid] + red[16 + tid] + red[24 + tid];
}
// out[0] = total, [1] = mse, [2] = freq, [3] = ms, [4] = cons, [5] = fw*freq + mw*ms + cw*cons
__global__ void __launch_bounds__(256) latent_loss_finish_kernel(const float* __restrict__ part, const LatentLossArgs a,
                                                                 float* __restrict__ out) {
  __shared__ double red[8][256];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < a.rows; r += 256)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (double)part[(int64_t)r * 8 + i];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[i][threadIdx.x] = acc[i];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double rows = a.rows, T = a.T, n = rows * T;
    const double mse = red[0][0] / n;
    const double freq = red[1][0] / (rows * a.F) + (a.low > 0 ? 0.1 * red[2][0] / (rows * a.low) : 0.0);
    const int T2 = a.T / 2, T4 = a.T / 4;
    const double ms = (red[3][0] / n + (T2 > 0 ? red[4][0] / (rows * T2) : 0.0) + (T4 > 0 ? red[5][0] / (rows * T4) : 0.0)) / 3.0;
    const int bw = a.soft - a.strict;
    const double cons = (a.strict > 0 ? red[6][0] / (rows * a.strict) : 0.0) + (bw > 0 ? red[7][0] / (rows * bw) : 0.0);
    const double lat = a.fw * freq + a.mw * ms + a.cw * cons;
    out[0] = (float)(mse + a.lw * lat); out[1] = (float)mse; out[2] = (float)freq; out[3] = (float)ms;
    out[4] = (float)cons; out[5] = (float)lat;
  }
}
hipError_t launch_latent_loss(const float* pred, const float* target, const float* lr, const float2* tw, float* dpred,
                              float* part, float* out6, int rows, int T, float lw, float fw, float mw, float cw, int low,
                              int strict, int soft, float loss_scale, hipStream_t s) {
  LatentLossArgs a;
  a.pred = pred; a.target = target; a.lr = lr; a.tw = tw; a.dpred = dpred; a.part = part;
  a.rows = rows; a.T = T; a.F = T / 2 + 1;
  a.low = low; a.strict = strict; a.soft = soft;
  if (low < 0 || low > a.F || strict < 0 || soft < strict || soft > a.F) return hipErrorInvalidValue;
  a.lw = lw; a.fw = fw; a.mw = mw; a.cw = cw; a.gscale = loss_scale;
  a.inv_n = 1.0f / ((float)rows * (float)T);
  const size_t lds = (size_t)(3 * T) * 4 + (size_t)(T + a.F) * 8 + 32 * 4;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  int N1 = 1;                       // largest divisor of T not above sqrt(T)
  for (int d = 2; d * d <= T; ++d)
    if (T % d == 0) N1 = d;
  static const int direct_env = getenv("JAT_LOSS_DIRECT_DFT") ? atoi(getenv("JAT_LOSS_DIRECT_DFT")) : 0;   // A/B
  if (N1 >= 2 && !direct_env) {
    const int N2 = T / N1, N1h = N1 / 2 + 1;
    const int ybuf = 3 * N1h * N2 > T ? 3 * N1h * N2 : T;
    const size_t lds2 = (size_t)(3 * T) * 4 + (size_t)(T + a.F + ybuf) * 8 + 32 * 4;
    if (lds2 <= 160 * 1024) {
      static size_t attr2 = 0;
      if (lds2 > attr2) {
        hipError_t e = hipFuncSetAttribute((const void*)latent_loss_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        attr2 = lds2;
      }
      hipLaunchKernelGGL(latent_loss_fft_kernel, dim3(rows), dim3(256), lds2, s, a, N1, N2);
      hipLaunchKernelGGL(latent_loss_finish_kernel, dim3(1), dim3(256), 0, s, part, a, out6);
      return hipGetLastError();
    }
  }
  const int fb = (a.F + 255) / 256, nb = (T + 255) / 256;
#define LL_CASE(FBv, NBv)                                                                                              \
  {                                                                                                                    \
    static size_t attr_set = 0;                                                                                        \
    if (lds > attr_set) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)latent_loss_kernel<FBv, NBv>,                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
      attr_set = lds;                                                                                                  \
    }                                                                                                                  \
    hipLaunchKernelGGL((latent_loss_kernel<FBv, NBv>), dim3(rows), dim3(256), lds, s, a);                              \
  }
  if (fb <= 1 && nb <= 2) LL_CASE(1, 2)
  else if (fb <= 2 && nb <= 4) LL_CASE(2, 4)
  else LL_CASE(3, 6)   // larger T: the kernel loops in chunks of 256 * FB bins / 256 * NB samples
#undef LL_CASE
  hipLaunchKernelGGL(latent_loss_finish_kernel, dim3(1), dim3(256), 0, s, part, a, out6);
  return hipGetLastError();
}
