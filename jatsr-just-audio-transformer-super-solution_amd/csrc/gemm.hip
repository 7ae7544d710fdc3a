// bf16 MFMA GEMM for gfx950 with fused epilogues:  C[M,N] = A[M,K] * W[N,K]^T.
//
// Both operands are K-contiguous (activations [tokens, features], nn.Linear weights [out, in]), so A and
// W fragments are fetched the same way.  Structure (cdna_hip_programming.md §5 / T2 / T3 "minimum 2-phase"):
//   - BK = 64; LDS image per operand = [rows][128 B], 16-B chunks XOR-swizzled by (row & 7) so that the
//     ds_read_b128 fragment reads are bank-conflict free;
//   - global -> LDS by global_load_lds_dwordx4 (1 KiB per wave-instruction, LDS image lane-linear, the
//     swizzle applied on the per-lane SOURCE address and again on the read: rule 21);
//   - 2 LDS stages, one barrier per K-tile, the next tile's DMA in flight under the current tile's MFMAs;
//   - v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W fragment as A, activation fragment as B):
//     the accumulator tile is then C^T, i.e. each lane owns 4 CONSECUTIVE output features of ONE row,
//     which makes every epilogue store 8-16 B wide and lets RoPE pair (d, d+32) in registers;
//   - XCD-aware block remap (bijective) + grouped-M tile order for L2 reuse.
// M may be ragged (rows clamped on load, masked on store); N % BN == 0 and K % 64 == 0 are required.
#include "jat_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
  r.y = (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16);
  return r;
}
// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the bf16 output rounding): nn.GELU() erf form.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  poly *= t;
  const float e = 1.0f - poly * __expf(-z * z);
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(WM * WN * 64) gemm_bf16_kernel(const GemmArgs p) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 64;
  constexpr int NW = WM * WN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // 1-KiB DMA pieces per wave per K-tile
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  static_assert(EPI != EPI_QKV_ROPE || TN == 4, "RoPE epilogue needs a 64-wide wave tile (one head)");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> tile: XCD-contiguous chunks, then grouped-M order -------------------------------
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int first_m = (id / per_group) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int in_g = id % per_group;
  const int m0 = (first_m + in_g % gsz) * BM;
  const int n0 = (in_g / gsz) * BN;

  // ---- staging addresses (source-side swizzle) ---------------------------------------------------
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const bf16_t* a_src[AI];
  const bf16_t* b_src[BI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int r = (wave * AI + j) * 8 + srow;
    const int gm = min(m0 + r, p.M - 1);
    a_src[j] = p.A + (int64_t)gm * p.lda + schunk * 8;
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = (wave * BI + j) * 8 + srow;
    b_src[j] = p.W + (int64_t)(n0 + r) * p.ldw + schunk * 8;
  }
  auto stage = [&](int st, int kt) {
    char* sA = smem + st * STAGE;
    char* sB = sA + A_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < AI; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(a_src[j] + koff), (lds_ptr_t)(sA + (wave * AI + j) * 1024),
                                       16, 0, 0);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(b_src[j] + koff), (lds_ptr_t)(sB + (wave * BI + j) * 1024),
                                       16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fg = lane >> 4;
  const int a_row_off = (wm * TM * 16 + frow) * 128;
  const int b_row_off = (wn * TN * 16 + frow) * 128;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((s * 4 + fg) ^ (frow & 7)) * 16;
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(sA + a_row_off + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(sB + b_row_off + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns C[m][n..n+3], m = tile row (lane&15), n = 4*(lane>>4) + reg -----------
  const int nw0 = n0 + wn * TN * 16;  // wave-uniform first column
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * TM * 16 + i * 16 + frow;
    if (m >= p.M) continue;
    int b = 0, pos = m;
    if constexpr (EPI == EPI_RESID || EPI == EPI_QKV_ROPE || EPI == EPI_UNPATCH) {
      b = m / p.ntok;
      pos = m - b * p.ntok;
    }
    if constexpr (EPI == EPI_QKV_ROPE) {
      // wave tile = one 64-wide head; RoPE pairs (d, d+32) = accumulator tiles (j, j+2), same lane/reg
      const int dl = fg * 4;
      if (nw0 < p.D + p.kvD) {  // q or k head: rotate (jat_audiosr_v3.py:87-108)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float4 c = *(const float4*)(p.rope_cos + (int64_t)pos * 32 + j * 16 + dl);
          const float4 s = *(const float4*)(p.rope_sin + (int64_t)pos * 32 + j * 16 + dl);
          const f32x4 x1 = acc[i][j], x2 = acc[i][j + 2];
          acc[i][j] = f32x4{x1[0] * c.x - x2[0] * s.x, x1[1] * c.y - x2[1] * s.y, x1[2] * c.z - x2[2] * s.z,
                            x1[3] * c.w - x2[3] * s.w};
          acc[i][j + 2] = f32x4{x2[0] * c.x + x1[0] * s.x, x2[1] * c.y + x1[1] * s.y,
                                x2[2] * c.z + x1[2] * s.z, x2[3] * c.w + x1[3] * s.w};
        }
        bf16_t* dst = (nw0 < p.D) ? ((bf16_t*)p.out + (int64_t)m * p.D + nw0)
                                  : (p.k_out + (int64_t)m * p.kvD + (nw0 - p.D));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          *(uint2*)(dst + j * 16 + dl) = pack4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {  // v head: store transposed vt[b][hv][d][pos]
        const int hv = (nw0 - p.D - p.kvD) >> 6;
        bf16_t* dst = p.vt_out + ((int64_t)(b * (p.kvD >> 6) + hv) * 64) * p.npad + pos;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(int64_t)(j * 16 + dl + r) * p.npad] = f2bf(acc[i][j][r]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw0 + j * 16 + fg * 4;
        f32x4 v = acc[i][j];
        if (p.bias) {
          const float4 bb = *(const float4*)(p.bias + n);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if constexpr (EPI == EPI_F32) {
          *(float4*)((float*)p.out + (int64_t)m * p.ldo + n) = float4{v[0], v[1], v[2], v[3]};
        } else if constexpr (EPI == EPI_BF16) {
          *(uint2*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EPI == EPI_BF16_GELU) {
          *(uint2*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) =
              pack4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else if constexpr (EPI == EPI_RESID) {
          const float4 g = *(const float4*)(p.gate + (int64_t)b * p.gate_bstride + n);
          float4* xp = (float4*)((float*)p.out + (int64_t)m * p.ldo + n);
          float4 x = *xp;
          x.x += g.x * v[0]; x.y += g.y * v[1]; x.z += g.z * v[2]; x.w += g.w * v[3];
          *xp = x;
        } else if constexpr (EPI == EPI_UNPATCH) {
          const int c = n >> 2, t0 = pos * 4;
          float* dst = (float*)p.out + ((int64_t)b * p.C_out + c) * p.T_orig + t0;
          if ((p.T_orig & 3) == 0) {
            *(float4*)dst = float4{v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t0 + r < p.T_orig) dst[r] = v[r];
          }
        }
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int LDS = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  auto kern = gemm_bf16_kernel<WM, WN, TM, TN, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (a.N % BN != 0 || a.K % 64 != 0 || a.M <= 0) return hipErrorInvalidValue;
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), LDS, s, a);
  return hipGetLastError();
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_epi(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_F32: return launch_one<WM, WN, TM, TN, EPI_F32>(a, s);
    case EPI_BF16: return launch_one<WM, WN, TM, TN, EPI_BF16>(a, s);
    case EPI_BF16_GELU: return launch_one<WM, WN, TM, TN, EPI_BF16_GELU>(a, s);
    case EPI_RESID: return launch_one<WM, WN, TM, TN, EPI_RESID>(a, s);
    case EPI_QKV_ROPE: return launch_one<WM, WN, TM, TN, EPI_QKV_ROPE>(a, s);
    case EPI_UNPATCH: return launch_one<WM, WN, TM, TN, EPI_UNPATCH>(a, s);
  }
  return hipErrorInvalidValue;
}

int gemm_num_variants() { return 3; }

hipError_t launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
  // fall back to the 128-wide tile whenever N is not a multiple of 256
  if (variant == 2 && a.N % 256 != 0) variant = 0;
  switch (variant) {
    case 0: return launch_epi<2, 2, 4, 4>(a, epi, s);  // 128 x 128, 4 waves
    case 1: return launch_epi<4, 2, 4, 4>(a, epi, s);  // 256 x 128, 8 waves
    case 2: return launch_epi<2, 4, 8, 4>(a, epi, s);  // 256 x 256, 8 waves
  }
  return hipErrorInvalidValue;
}
