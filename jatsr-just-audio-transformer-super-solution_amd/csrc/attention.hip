// Grouped-query attention forward for gfx950:  O = softmax(Q K^T / sqrt(64)) V, no mask, head_dim 64
// (reference: GroupedQueryAttention.forward, src/models/jat_audiosr_v3.py:164-181; q-head h reads
// kv-head h / (Hq/Hkv), the repeat_interleave of :164-165 is never materialised).
//
// One workgroup = 4 waves = 64*QT query rows of one (batch, q-head); each wave owns QT 16-row q tiles.
// Keys/values stream through LDS in blocks of 64 keys (flash-style online softmax, so any N <= 2048
// works; N = 128 is two blocks).  Layout choices that keep everything in registers between the MFMAs:
//   - scores are computed TRANSPOSED, S^T = K Q^T (K fragment as MFMA A operand, Q fragment as B):
//     the accumulator then has the query on the lane (col = lane&15) and keys down the registers, so a
//     softmax row reduction is 15 in-lane max/add + 2 cross-lane shuffles (xor 16, 32);
//   - the K rows fed to tile kt are permuted (row i <- key 32*(kt/2) + 8*(i/4) + (i%4) + 4*(kt%2)) so
//     that the 8 probabilities a lane holds for a 32-key group are 8 CONSECUTIVE keys: exactly the B-operand
//     fragment of the next MFMA (O^T = V^T P^T) with V^T rows read by one ds_read_b128 — P never touches LDS;
//   - V arrives already transposed (vt[b][kvh][d][key], written by the QKV GEMM epilogue) and zero padded
//     to a multiple of 64 keys; K rows past N are clamped on load and masked to -1e30 before the softmax;
//   - K and V^T LDS images are [64 rows][128 B] with the 16-B chunk XOR-swizzled (V^T by row & 7, K by the
//     row bits its permuted read order exercises) — conflict-free ds_read_b128, cdna_hip_programming.md T2.
// Softmax statistics, the running output and the 1/l normalisation are fp32; P is rounded to bf16 for PV.
#include "jat_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned short f2bf_a(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// K-tile chunk swizzle: the permuted row order {0-3, 8-11, 16-19, 24-27}(+4) only exercises bits 0,1,3 of the
// row, so XOR with those (simulated conflict-free for ds_read_b128; `row & 7` would be 2-way here).
__device__ __forceinline__ int kswz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int QT>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8192];
  char* sK = smem;
  char* sV = smem + 8192;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.Hq / p.Hkv);
  const int N = p.N;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);

  // Q fragments (MFMA B operand): lane holds Q[q0 + qt*16 + (lane&15)][s*32 + 8*(lane>>4) .. +7]
  bf16x8 qf[QT][2];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = min(q0 + qt * 16 + frow, N - 1);
    const bf16_t* qp = p.q + ((int64_t)b * N + q) * p.ldq + h * 64 + fg * 8;
    qf[qt][0] = *(const bf16x8*)(qp);
    qf[qt][1] = *(const bf16x8*)(qp + 32);
  }

  float m_run[QT], l_run[QT];
  f32x4 o[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -1e30f;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const bf16_t* kbase = p.k + (int64_t)b * N * p.ldk + hk * 64;
  const bf16_t* vbase = p.vt + ((int64_t)(b * p.Hkv + hk) * 64) * p.npad;

  const int nkb = (N + 63) >> 6;
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * 64;
    __syncthreads();  // all waves done reading the previous block
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j, row = c >> 3, ch = c & 7;
      const int dst = row * 128 + ((ch ^ (row & 7)) << 4);        // V^T image: rows read in natural order
      const int dstk = row * 128 + ((ch ^ kswz(row)) << 4);       // K image: rows read in permuted order
      const uint4 kv = *(const uint4*)(kbase + (int64_t)min(key0 + row, N - 1) * p.ldk + ch * 8);
      const uint4 vv = *(const uint4*)(vbase + (int64_t)row * p.npad + key0 + ch * 8);
      *(uint4*)(sK + dstk) = kv;
      *(uint4*)(sV + dst) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T ---------------------------------------------------------------------------
    f32x4 st[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) st[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int r = 32 * (kt >> 1) + 8 * (frow >> 2) + (frow & 3) + 4 * (kt & 1);  // permuted key row
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kf = *(const bf16x8*)(sK + r * 128 + (((s * 4 + fg) ^ kswz(r)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          st[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], st[qt][kt], 0, 0, 0);
      }
    }

    // ---- online softmax (per query = per lane column; keys across 16 regs and the 4 lane groups) --
    bf16x8 pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + 32 * (kt >> 1) + 8 * fg + 4 * (kt & 1) + r;
          const float sv = key < N ? st[qt][kt][r] * p.scale_log2e : -1e30f;
          st[qt][kt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = exp2f(st[qt][kt][r] - m_new);
          st[qt][kt][r] = pv;
          sum += pv;
        }
      l_run[qt] = l_run[qt] * alpha + sum;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        o[qt][dt][0] *= alpha; o[qt][dt][1] *= alpha; o[qt][dt][2] *= alpha; o[qt][dt][3] *= alpha;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = (__bf16)st[qt][2 * kk][r];
          f[4 + r] = (__bf16)st[qt][2 * kk + 1][r];
        }
        pf[qt][kk] = f;
      }
    }

    // ---- O^T += V^T P^T --------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + frow;
        const bf16x8 vf = *(const bf16x8*)(sV + d * 128 + (((kk * 4 + fg) ^ (d & 7)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][kk], o[qt][dt], 0, 0, 0);
      }
  }

  // ---- normalise and store: lane owns O[q][dt*16 + 4*fg .. +3] ----------------------------------------
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int q = q0 + qt * 16 + frow;
    if (q < N) {
      bf16_t* op = p.o + ((int64_t)b * N + q) * p.ldo + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 r;
        r.x = (unsigned)f2bf_a(o[qt][dt][0] * inv) | ((unsigned)f2bf_a(o[qt][dt][1] * inv) << 16);
        r.y = (unsigned)f2bf_a(o[qt][dt][2] * inv) | ((unsigned)f2bf_a(o[qt][dt][3] * inv) << 16);
        *(uint2*)(op + dt * 16) = r;
      }
    }
  }
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.B <= 0 || a.Hq % a.Hkv != 0 || a.npad % 64 != 0 || a.npad < a.N) return hipErrorInvalidValue;
  constexpr int QT = 2;
  dim3 grid((a.N + 64 * QT - 1) / (64 * QT), a.Hq, a.B);
  hipLaunchKernelGGL(attn_fwd_kernel<QT>, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
