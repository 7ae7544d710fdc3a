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
#include "jat_dtype.h"
#include <cstdlib>

typedef jat_opx8 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned short f2bf_a(float f) { return jat_f2op(f); }

// K-tile chunk swizzle: the permuted row order {0-3, 8-11, 16-19, 24-27}(+4) only exercises bits 0,1,3 of the
// row, so XOR with those (simulated conflict-free for ds_read_b128; `row & 7` would be 2-way here).
__device__ __forceinline__ int kswz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int QT, int KVB>   // QT 16-row query tiles per wave; KVB keys per LDS block (64 or 128)
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnArgs p) {
  constexpr int NKT = KVB / 16;   // 16-key MFMA tiles per block
  constexpr int NKK = KVB / 32;   // 32-key k-steps of the PV product
  constexpr int NCH = KVB / 32;   // 16-B staging chunks per thread per operand (KVB*8 chunks / 256 threads)
  __shared__ __attribute__((aligned(16))) char smem[2 * KVB * 128];
  char* sK = smem;                // [KVB keys][128 B]
  char* sV = smem + KVB * 128;    // [64 d][KVB * 2 B]  (row stride KVB*2 bytes, 128-B sub-rows swizzled)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.Hq / p.Hkv);
  const int N = p.N;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);

  const bf16_t* kbase = p.k + (int64_t)b * N * p.ldk + hk * 64;
  const bf16_t* vbase = p.vt + ((int64_t)(b * p.Hkv + hk) * 64) * p.npad;

  // register staging (issued one block ahead of the LDS write: T14 split)
  u32x4 kreg[NCH], vreg[NCH];
  auto load_regs = [&](int key0) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j;
      const int krow = c >> 3, kch = c & 7;                 // K: KVB rows x 8 chunks
      kreg[j] = *(const u32x4*)(kbase + (int64_t)min(key0 + krow, N - 1) * p.ldk + kch * 8);
      const int vrow = c / (KVB / 8), vch = c % (KVB / 8);  // V^T: 64 rows x KVB/8 chunks
      const int vkey = min(key0 + vch * 8, p.npad - 8);
      vreg[j] = *(const u32x4*)(vbase + (int64_t)vrow * p.npad + vkey);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j;
      const int krow = c >> 3, kch = c & 7;
      *(u32x4*)(sK + krow * 128 + ((kch ^ kswz(krow)) << 4)) = kreg[j];
      const int vrow = c / (KVB / 8), vch = c % (KVB / 8);
      // V^T row = 2 (KVB=128) or 1 (KVB=64) sub-rows of 128 B; swizzle the chunk inside its sub-row
      *(u32x4*)(sV + vrow * (KVB * 2) + (vch >> 3) * 128 + (((vch & 7) ^ (vrow & 7)) << 4)) = vreg[j];
    }
  };
  load_regs(0);

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

  const int Nb = p.lens ? min(p.lens[b], N) : N;   // keys this sample attends to (block-uniform)
  const int nkb = (Nb + KVB - 1) / KVB;            // same key blocks as a stand-alone run of Nb tokens
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * KVB;
    __syncthreads();  // all waves done reading the previous block
    write_lds();
    __syncthreads();
    if (kb + 1 < nkb) load_regs(key0 + KVB);  // next block's loads fly under this block's MFMAs

    // ---- S^T = K Q^T ---------------------------------------------------------------------------
    f32x4 st[QT][NKT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) st[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int r = 32 * (kt >> 1) + 8 * (frow >> 2) + (frow & 3) + 4 * (kt & 1);  // permuted key row
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kf = *(const bf16x8*)(sK + r * 128 + (((s * 4 + fg) ^ kswz(r)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          st[qt][kt] = JAT_MFMA_16x16x32(kf, qf[qt][s], st[qt][kt], 0, 0, 0);
      }
    }

    // ---- online softmax (per query = per lane column; keys across regs and the 4 lane groups) ------
    bf16x8 pf[QT][NKK];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // max over the RAW scores (the scale is positive), keys past the end masked in the last key block only; the scale and
      // the running maximum then enter one FMA per score: exp2(s * c - m)
      float mx = -1e30f;
      if (key0 + KVB > Nb) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 32 * (kt >> 1) + 8 * fg + 4 * (kt & 1) + r;
            if (key >= Nb) st[qt][kt][r] = -1e30f;
          }
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        mx = fmaxf(mx, fmaxf(fmaxf(st[qt][kt][0], st[qt][kt][1]), fmaxf(st[qt][kt][2], st[qt][kt][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qt], mx * p.scale_log2e);
      const float alpha = exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(st[qt][kt][r], p.scale_log2e, -m_new));
          st[qt][kt][r] = pv;
          sum += pv;
        }
      if (p.drop.thresh) {   // training: dropout on the normalised probabilities = mask the numerators, keep the row sum
        const int64_t qrow = (((int64_t)b * p.Hq + h) * N + min(q0 + qt * 16 + frow, N - 1)) * N;
        if ((uint64_t)p.B * p.Hq * N * N <= 0xffffffffull) {   // uniform: every element index of this call fits 32 bits
          const uint32_t q32 = (uint32_t)qrow + key0 + 8 * fg;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[qt][kt][r] *= jat_drop_mult32(p.drop, q32 + 32 * (kt >> 1) + 4 * (kt & 1) + r);
        } else {
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = key0 + 32 * (kt >> 1) + 8 * fg + 4 * (kt & 1) + r;
              st[qt][kt][r] *= jat_drop_mult(p.drop, (uint64_t)(qrow + key));
            }
        }
      }
      l_run[qt] = l_run[qt] * alpha + sum;
      if (kb > 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          o[qt][dt][0] *= alpha; o[qt][dt][1] *= alpha; o[qt][dt][2] *= alpha; o[qt][dt][3] *= alpha;
        }
      }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        pf[qt][kk] = jat_pack8(st[qt][2 * kk], st[qt][2 * kk + 1]);
      }
    }

    // ---- O^T += V^T P^T --------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + frow;
        const int ch = kk * 4 + fg;  // 16-B chunk (8 keys) inside the V^T row
        const bf16x8 vf = *(const bf16x8*)(sV + d * (KVB * 2) + (ch >> 3) * 128 + (((ch & 7) ^ (d & 7)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          o[qt][dt] = JAT_MFMA_16x16x32(vf, pf[qt][kk], o[qt][dt], 0, 0, 0);
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
      // training forward: log2-domain log-sum-exp of the scaled scores, P = exp2(s * scale_log2e - lse) in the backward
      if (p.lse && fg == 0) p.lse[((int64_t)b * p.Hq + h) * N + q] = m_run[qt] + log2f(l);
      bf16_t* op = p.o + ((int64_t)b * N + q) * p.ldo + h * 64 + fg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 r;
        r.x = jat_pack2(o[qt][dt][0] * inv, o[qt][dt][1] * inv);
        r.y = jat_pack2(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
        *(uint2*)(op + dt * 16) = r;
      }
    }
  }
}

// ---- short-sequence variant (N <= 128, the sampler's N = 128): one workgroup per (batch, KV head) --------------
// K [128 keys][64] and V^T [64][128 keys] are staged in LDS ONCE and reused by the G = Hq/Hkv query heads that share
// them (the repeat_interleave of jat_audiosr_v3.py:164-165); no online-softmax state is carried because all keys
// fit one block.  The next head's Q fragments are prefetched under the current head's MFMAs.
template <int QT, int NWV>   // NWV waves per block, each owning QT 16-row query tiles (NWV*QT*16 = 128 rows)
__global__ void __launch_bounds__(NWV * 64) attn_group_kernel(const AttnArgs p) {
  constexpr int KVB = 128, NKT = 8, NKK = 4, NT = NWV * 64, NCH = 1024 / NT;
  static_assert(NWV * QT * 16 == 128, "one block covers 128 query rows");
  __shared__ __attribute__((aligned(16))) char smem[2 * KVB * 128];
  char* sK = smem;
  char* sV = smem + KVB * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int hk = blockIdx.x, b = blockIdx.y;
  const int G = p.Hq / p.Hkv, N = p.N;
  const int Nk = p.lens ? min(p.lens[b], N) : N;   // keys this sample attends to (a short chunk padded into a longer bucket)
  const int q0 = wave * (16 * QT);
  const bf16_t* kbase = p.k + (int64_t)b * N * p.ldk + hk * 64;
  const bf16_t* vbase = p.vt + ((int64_t)(b * p.Hkv + hk) * 64) * p.npad;
  {
    u32x4 kreg[NCH], vreg[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + NT * j;
      kreg[j] = *(const u32x4*)(kbase + (int64_t)min(c >> 3, N - 1) * p.ldk + (c & 7) * 8);
      vreg[j] = *(const u32x4*)(vbase + (int64_t)(c >> 4) * p.npad + min((c & 15) * 8, p.npad - 8));
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + NT * j;
      const int krow = c >> 3, kch = c & 7, vrow = c >> 4, vch = c & 15;
      *(u32x4*)(sK + krow * 128 + ((kch ^ kswz(krow)) << 4)) = kreg[j];
      *(u32x4*)(sV + vrow * 256 + (vch >> 3) * 128 + (((vch & 7) ^ (vrow & 7)) << 4)) = vreg[j];
    }
  }
  auto load_q = [&](bf16x8(&qf)[QT][2], int h) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = min(q0 + qt * 16 + frow, N - 1);
      const bf16_t* qp = p.q + ((int64_t)b * N + q) * p.ldq + h * 64 + fg * 8;
      qf[qt][0] = *(const bf16x8*)(qp);
      qf[qt][1] = *(const bf16x8*)(qp + 32);
    }
  };
  bf16x8 qa[QT][2], qb[QT][2];
  load_q(qa, hk * G);
  __syncthreads();

  auto head = [&](const bf16x8(&qf)[QT][2], int h) {
    f32x4 st[QT][NKT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) st[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int r = 32 * (kt >> 1) + 8 * (frow >> 2) + (frow & 3) + 4 * (kt & 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kf = *(const bf16x8*)(sK + r * 128 + (((s * 4 + fg) ^ kswz(r)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          st[qt][kt] = JAT_MFMA_16x16x32(kf, qf[qt][s], st[qt][kt], 0, 0, 0);
      }
    }
    bf16x8 pf[QT][NKK];
    float linv[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (Nk < KVB) {  // block-uniform: mask the padded keys only when there are any
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (32 * (kt >> 1) + 8 * fg + 4 * (kt & 1) + r >= Nk) st[qt][kt][r] = -1e30f;
      }
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        mx = fmaxf(mx, fmaxf(fmaxf(st[qt][kt][0], st[qt][kt][1]), fmaxf(st[qt][kt][2], st[qt][kt][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float nb = -mx * p.scale_log2e;   // scale > 0: max commutes with the scaling
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(st[qt][kt][r], p.scale_log2e, nb));  // one FMA + v_exp_f32
          st[qt][kt][r] = pv;
          sum += pv;
        }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      linv[qt] = 1.0f / sum;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        pf[qt][kk] = jat_pack8(st[qt][2 * kk], st[qt][2 * kk + 1]);
      }
    }
    f32x4 o[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + frow, ch = kk * 4 + fg;
        const bf16x8 vf = *(const bf16x8*)(sV + d * 256 + (ch >> 3) * 128 + (((ch & 7) ^ (d & 7)) << 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          o[qt][dt] = JAT_MFMA_16x16x32(vf, pf[qt][kk], o[qt][dt], 0, 0, 0);
      }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = q0 + qt * 16 + frow;
      if (q < N) {
        bf16_t* op = p.o + ((int64_t)b * N + q) * p.ldo + h * 64 + fg * 4;
        const float inv = linv[qt];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 r;
          r.x = jat_pack2(o[qt][dt][0] * inv, o[qt][dt][1] * inv);
          r.y = jat_pack2(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
          *(uint2*)(op + dt * 16) = r;
        }
      }
    }
  };
  // heads two at a time so that the Q double buffer is addressed statically (rule 20)
  for (int g = 0; g < G; g += 2) {
    const int h = hk * G + g;
    if (g + 1 < G) load_q(qb, h + 1);
    head(qa, h);
    if (g + 1 < G) {
      if (g + 2 < G) load_q(qa, h + 2);
      head(qb, h + 1);
    }
  }
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.B <= 0 || a.Hq % a.Hkv != 0 || a.npad % 64 != 0 || a.npad < a.N) return hipErrorInvalidValue;
  static const int kvb_env = getenv("JAT_ATTN_KVB") ? atoi(getenv("JAT_ATTN_KVB")) : 64;
  static const int group_env = getenv("JAT_ATTN_GROUP") ? atoi(getenv("JAT_ATTN_GROUP")) : 1;
  static const int qt_env = getenv("JAT_ATTN_QT") ? atoi(getenv("JAT_ATTN_QT")) : 0;   // 0: by block count
  // 16 queries per wave (QT = 1).  32 per wave halve the K/V staging per query, but the kernel is bound by the per-element softmax
  // (and dropout-hash) VALU work of a wave, not by staging, and twice the blocks hide each other's barriers better: measured
  // one chunk (B = 2 with CFG, N = 345: 120 -> 240 blocks) 140.8 -> 134.6 ms, a four-chunk file 243.8 -> 241.4 ms, the training
  // step (B = 28, N = 345) 62.05 -> 61.39 ms (profiles/r03/single_chunk_attention_qt_sweep.log, attention_qt_long_train.log)
  const int qt = qt_env ? qt_env : 1;
  dim3 grid((a.N + 64 * qt - 1) / (64 * qt), a.Hq, a.B);
  const bool kvb64 = kvb_env == 64 || a.N <= 64;
  if (group_env && a.N <= 128 && a.npad >= 128 && !a.lse && !a.drop.thresh) {   // the sampler's shape: K/V staged once per KV head (lens honoured)
    hipLaunchKernelGGL((attn_group_kernel<1, 8>), dim3(a.Hkv, a.B), dim3(512), 0, s, a);
  } else if (qt == 1) {
    if (kvb64) hipLaunchKernelGGL((attn_fwd_kernel<1, 64>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<1, 128>), grid, dim3(256), 0, s, a);
  } else {
    if (kvb64) hipLaunchKernelGGL((attn_fwd_kernel<2, 64>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<2, 128>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}
