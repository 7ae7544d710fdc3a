// Weight-gradient GEMM of the training step (the backward of every nn.Linear of
// models/JaT_V3.py under train_ddp_v3m2.py:601 scaler.scale(loss).backward()):
//
//     dW[o][i] = sum_tok dY[tok][o] * X[tok][i]          db[o] = sum_tok dY[tok][o]
//
// Both operands are stored token-major, i.e. the reduction index is the ROW of either matrix ("TN" in BLAS terms), so
// gemm_bf16_kernel (K-contiguous operands) needed a transposed copy of each: 22 GB of HBM traffic and 4.9 ms of a
// 65 ms step (profiles/r02/train_T1378_kernel_stats.txt).  This kernel reads the row-major activations directly:
//   - a K-tile is 64 token rows of a 128-column panel of each operand: 64 x 256 B, staged global -> LDS by
//     global_load_lds_dwordx4 (one wave-instruction = 4 rows x 256 B, lane-linear in LDS), 16-B chunk c of row r landing
//     at chunk position c ^ sw(r), sw(r) = ((r & 3) << 2) | ((r >> 2) & 3)  (source-side swizzle, whole rows still read
//     contiguously);
//   - MFMA operand fragments (8 consecutive k of one column per lane) come out of that k-major image by
//     ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block column-major; with the swizzle above the two
//     groups of a 32-lane half (rows 8 apart, same columns) and the four rows of a group fall in disjoint banks;
//   - 128 x 128 output tile, 4 waves (2 x 2) of 64 x 64 (16 accumulator tiles), 2 LDS stages of 32 KiB, 2 blocks per CU;
//     fragments double-buffered in registers, one barrier per K-tile, the DMA of tile t+2 in flight under the MFMAs of t+1.
//     Used for the small weights; the big ones take the 256 x 256 form further down (this one is bound by the ~90 GB/s a
//     CU can pull out of L2: 64 KiB per K-tile for two co-resident blocks);
//   - token rows past the end (the last K-tile of a ragged batch) are zeroed in the LDS image of ONE operand, the
//     source rows clamped in range: no padded copies, no assumption about what follows the buffers;
//   - split-K over blockIdx.y writes fp32 partial slices, summed in order by sum_partials_kernel (train.hip): no
//     atomics, a step stays bit-reproducible.
// Output lane mapping: D = mfma(X-fragment, dY-fragment) leaves lane (g, fr) with dW[o = fr][i = 4g .. 4g+3] of a
// 16 x 16 tile, stored as one float4.
#include "jat_kernels.h"
#include "jat_dtype.h"
#include <cstdlib>

typedef jat_opx8 opx8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct GemmTnArgs {
  const bf16_t* A; int64_t lda;   // dY [K tokens][M]
  const bf16_t* B; int64_t ldb;   // X  [K tokens][N]
  float* out; int64_t ldo;        // dW [M][N] (+ z * split_stride)
  int M, N, K;
  int ksplit; int64_t split_stride;
  const void* zeros;              // >= 16 zero bytes (the 256 x 256 kernel's source for token rows past the end)
};

namespace {
constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int IMG = TBK * 256;     // one operand image: 64 token rows x 256 B
constexpr int STAGE = 2 * IMG;     // dY image, then X image

__device__ __forceinline__ opx8 tr_pair(const unsigned char* img, int off_lo, int off_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + off_hi));
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(opx8, both);
}
}  // namespace

__global__ void __launch_bounds__(256, 2) gemm_tn_kernel(GemmTnArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_tn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // block -> tile: XCD-contiguous chunks (each XCD's L2 sees few dY panels and every X panel), N fastest
  const int tiles_n = p.N / TBN;
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (id / tiles_n) * TBM, n0 = (id % tiles_n) * TBN;

  const int nkt = (p.K + TBK - 1) / TBK;
  int kt0 = 0, kt1 = nkt;
  float* out = p.out;
  if (p.ksplit > 1) {
    const int z = blockIdx.y;
    kt0 = (int)((int64_t)z * nkt / p.ksplit);
    kt1 = (int)((int64_t)(z + 1) * nkt / p.ksplit);
    out += (int64_t)z * p.split_stride;
  }

  // ---- staging: wave w moves pieces w, w+4, w+8, w+12 (4 token rows each) of both images --------------------------
  const int srow = lane >> 4;                                   // row inside the piece
  int soff[4];                                                  // element offset of my 16 B inside the 128-column panel
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave + 4 * j) * 4 + srow;
    soff[j] = ((lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3))) * 8;
  }
  auto stage = [&](int st, int kt) {
    unsigned char* sA = smem_tn + st * STAGE;
    unsigned char* sB = sA + IMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = wave + 4 * j;
      const int64_t tok = min(kt * TBK + pc * 4 + srow, p.K - 1);
      __builtin_amdgcn_global_load_lds((const void*)(p.A + tok * p.lda + m0 + soff[j]), (lds_ptr_t)(sA + pc * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(p.B + tok * p.ldb + n0 + soff[j]), (lds_ptr_t)(sB + pc * 1024), 16, 0, 0);
    }
  };

  // ---- fragment addresses: element j of lane (fg, fr) = img[ks*32 + 8 fg + j][c0 + fr]; the lane supplies the address of
  // row 8 fg + q (+ 4), columns c0 + 4 pp .. + 3 (q = fr >> 2, pp = fr & 3) ---------------------------------------------
  const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, pp = fr & 3;
  int offA[4][2], offB[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * fg + q + 4 * h;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);          // the same for rows + 32 (second k-step)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ca = (wm * 64 + t * 16) / 8 + (pp >> 1), cb = (wn * 64 + t * 16) / 8 + (pp >> 1);
      offA[t][h] = row * 256 + ((ca ^ sw) << 4) + (pp & 1) * 8;
      offB[t][h] = row * 256 + ((cb ^ sw) << 4) + (pp & 1) * 8;
    }
  }

  f32x4 acc[4][4];   // [dY tile (o)][X tile (i)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline (two k-steps per K-tile, fragments double-buffered in registers):
  //   phase A: read F1(tile t) | MFMAs on F0(tile t)        -> s_waitcnt vmcnt(0), barrier: tile t+1 landed, stage of t free
  //   DMA tile t+2 into the stage of t
  //   phase B: read F0(tile t+1) | MFMAs on F1(tile t)
  opx8 p0[4], q0[4], p1[4], q1[4];
  auto read_frags = [&](opx8 (&pf)[4], opx8 (&qf)[4], int st, int ks) {
    const unsigned char* sA = smem_tn + st * STAGE + ks * 32 * 256;
    const unsigned char* sB = sA + IMG;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pf[t] = tr_pair(sA, offA[t][0], offA[t][1]);
      qf[t] = tr_pair(sB, offB[t][0], offB[t][1]);
    }
  };
  auto mma = [&](const opx8 (&pf)[4], const opx8 (&qf)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = JAT_MFMA_16x16x32(qf[j], pf[i], acc[i][j], 0, 0, 0);
  };
  auto interleave = [&]() {   // one fragment read (2 ds_read_b64_tr_b16), then 2 MFMAs (sched_group_barrier masks: MFMA 0x8, DS read 0x100)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };
  // ragged tile: zero the dY rows past the last token (tile `kt` sits in stage `st`, landed and published by a barrier)
  auto zero_tail = [&](int st, int kt) {
    if ((kt + 1) * TBK > p.K) {
      unsigned char* sA = smem_tn + st * STAGE;
      const int rv = p.K - kt * TBK;
      for (int i = tid; i < (TBK - rv) * 16; i += 256) *(u32x4*)(sA + rv * 256 + i * 16) = u32x4{0u, 0u, 0u, 0u};
      __syncthreads();
    }
  };
  const int nk = kt1 - kt0;
  if (nk > 0) {
    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) stage(1, kt0 + 1);
    zero_tail(0, kt0);
    read_frags(p0, q0, 0, 0);
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
      read_frags(p1, q1, cur, 1);
      mma(p0, q0);
      interleave();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of tile t+1 have landed
      __syncthreads();                                    // every wave holds F1 in registers: stage `cur` is free
      if (t + 2 < nk) stage(cur, kt0 + t + 2);
      if (t + 1 < nk) {
        zero_tail(cur ^ 1, kt0 + t + 1);
        read_frags(p0, q0, cur ^ 1, 0);
      }
      mma(p1, q1);
      interleave();
    }
  }

  // ---- epilogue: lane (fg, fr) of tile (i, j) owns dW[o = fr][n = 4 fg .. + 3] -------------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * fg;
      *(f32x4*)(out + (int64_t)o * p.ldo + n) = acc[i][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 x 4 of 128 x 64), one block per CU: the form the big weights use.  A CU pulls at most ~90 GB/s
// out of its XCD's L2 (profiles/r02/gemm_tn_ablation.log: the 128 x 128 kernel's DMA alone takes 113 of its 190 us), so the tile
// has to be large enough that a K-tile's 64 KiB feed 128 MFMAs per wave.
//   - LDS image per operand: 64 token rows at a pitch of 544 B = 17 x 32 B.  An ODD number of 32-B units makes consecutive rows
//     start 8 banks apart, and the two 16-lane groups of a half-wave read rows 12 apart instead of 8 (odd groups fetch their
//     upper four k first): the 8 rows x 32 B of one ds_read_b64_tr_b16 half cover all 64 banks with NO address swizzle, so every
//     tile / k-step offset is an instruction immediate (2 address registers per operand instead of 24).  Both operands use
//     the same k order inside a fragment, so the products still pair up; only the summation order inside an MFMA changes.
//   - DMA in image order: a stage is 68 pieces of 1 KiB (34 per operand), lanes that fall into the 32-B row pad fetch a dummy;
//     rows past the last token of the dY operand are fetched from a zeroed 16-B cell instead of being zeroed afterwards.
//   - ping-pong: waves w and w+4 share a SIMD and run ONE barrier apart; a phase is the 24 transposed reads of one k-step plus
//     4-5 DMA pieces, s_barrier, its 32 MFMAs, s_barrier - one group's MFMAs run under the other's LDS reads and DMA issue.
//     The DMA of a tile is issued per k-step half, each half one phase after its last read (see the main loop).
namespace tn256 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int PITCH = 544;
constexpr int IMG2 = BK * PITCH;        // 34 KiB
constexpr int STAGE2 = 2 * IMG2;        // 68 pieces of 1 KiB
constexpr int NPIECE = STAGE2 / 1024, HPIECE = NPIECE / 2, PPH = (HPIECE + 7) / 8;   // 68 per stage, 34 per half, <= 5 per wave and half
static_assert(IMG2 % 2048 == 0, "no DMA piece may straddle the two operand images");
}  // namespace tn256

__global__ void __launch_bounds__(512, 1) gemm_tn256_kernel(GemmTnArgs p) {
  using namespace tn256;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_tn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 2) & 1, wn = wave & 3, grp = wave >> 2;   // w and w+4 (same SIMD) differ in wm: one per group
  const int tiles_n = p.N / BN;
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
  const int nkt = (p.K + BK - 1) / BK;
  int kt0 = 0, kt1 = nkt;
  float* out = p.out;
  if (p.ksplit > 1) {
    const int z = blockIdx.y;
    kt0 = (int)((int64_t)z * nkt / p.ksplit);
    kt1 = (int)((int64_t)(z + 1) * nkt / p.ksplit);
    out += (int64_t)z * p.split_stride;
  }
  const int nk = kt1 - kt0;

  // ---- DMA: a stage = two halves (token rows 0..31 / 32..63 of both images = the two k-steps), 34 pieces each: piece i of a
  // half is piece i (+17 for the second half) of the dY image for i < 17, of the X image otherwise; wave w moves i = w, w+8, ..
  int prow[2][PPH];
  const char* psrc[2][PPH];  // the lane's source address for K-tile kt0, advanced by 64 token rows per tile
  int pdst[2][PPH];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < PPH; ++j) {
      const int i = min(wave + 8 * j, HPIECE - 1);
      const bool isB = i >= HPIECE / 2;
      const int pc = (isB ? i - HPIECE / 2 : i) + h * (HPIECE / 2);       // piece inside its image
      const int bb = pc * 1024 + lane * 16;
      const int row = bb / PITCH, x = min(bb - row * PITCH, BM * 2 - 16);   // pad lanes re-fetch the row's last chunk
      prow[h][j] = row;
      pdst[h][j] = (isB ? IMG2 : 0) + pc * 1024;
      const int64_t tok = min(kt0 * BK + row, p.K - 1);
      psrc[h][j] = isB ? (const char*)(p.B + tok * p.ldb + n0) + x : (const char*)(p.A + tok * p.lda + m0) + x;
    }
  const int64_t stepA = (int64_t)BK * p.lda * 2, stepB = (int64_t)BK * p.ldb * 2;
  const bool last_slot = wave + 8 * (PPH - 1) < HPIECE;   // waves 0, 1 move 5 pieces per half, the others 4
  auto dma = [&](int st, int kt, int h) {   // kt: absolute K-tile index
    unsigned char* dst = smem_tn + st * STAGE2;
    const int64_t adv = kt - kt0;
    if ((kt + 1) * BK <= p.K) {
#pragma unroll
      for (int j = 0; j < PPH; ++j)
        if (j < PPH - 1 || last_slot) {
          const bool isB = wave + 8 * j >= HPIECE / 2;
          __builtin_amdgcn_global_load_lds((const void*)(psrc[h][j] + adv * (isB ? stepB : stepA)), (lds_ptr_t)(dst + pdst[h][j]), 16, 0, 0);
        }
    } else {                         // ragged last tile: dY rows past the last token come from the zero cell, X rows are clamped
#pragma unroll
      for (int j = 0; j < PPH; ++j)
        if (j < PPH - 1 || last_slot) {
          const bool isB = wave + 8 * j >= HPIECE / 2;
          const int tok = kt * BK + prow[h][j];
          const char* src = psrc[h][j] + adv * (isB ? stepB : stepA);
          if (tok >= p.K)
            src = isB ? psrc[h][j] + ((int64_t)(p.K - 1) - min(kt0 * BK + prow[h][j], p.K - 1)) * p.ldb * 2 : (const char*)p.zeros;
          __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr_t)(dst + pdst[h][j]), 16, 0, 0);
        }
    }
  };

  // ---- fragment addresses: lane (fg, fr) supplies the address of row 8 fg + q (+4), columns c0 + 4 pp ..; odd fg swap the halves
  const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, pp = fr & 3;
  const int row_lo = 8 * fg + q + 4 * (fg & 1), row_hi = 8 * fg + q + 4 * (1 - (fg & 1));
  const int aLo = row_lo * PITCH + wm * 256 + pp * 8, aHi = row_hi * PITCH + wm * 256 + pp * 8;
  const int bLo = IMG2 + row_lo * PITCH + wn * 128 + pp * 8, bHi = IMG2 + row_hi * PITCH + wn * 128 + pp * 8;

  f32x4 acc[8][4];   // [dY tile (o)][X tile (i)]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  opx8 pf[8], qf[4];
  auto rd = [&](int st, int ks) {
    const unsigned char* sb = smem_tn + st * STAGE2 + ks * 32 * PITCH;
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = tr_pair(sb + t * 32, bLo, bHi);
#pragma unroll
    for (int t = 0; t < 8; ++t) pf[t] = tr_pair(sb + t * 32, aLo, aHi);
  };
#define JAT_TN_LOAD_END()                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);                \
  __builtin_amdgcn_s_barrier();                     \
  __builtin_amdgcn_s_setprio(1);
#define JAT_TN_MMA()                                                              \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                                   \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                   \
    acc[i][j] = JAT_MFMA_16x16x32(qf[j], pf[i], acc[i][j], 0, 0, 0);              \
  __builtin_amdgcn_s_setprio(0);                                                  \
  __builtin_amdgcn_sched_barrier(0);                                              \
  __builtin_amdgcn_s_barrier();
  if (nk > 0) {
    dma(0, kt0, 0); dma(0, kt0, 1);
    if (nk > 1) { dma(1, kt0 + 1, 0); dma(1, kt0 + 1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    // P1(t): k-step 0 of tile t | DMA rows 32..63 of tile t+1 (their stage half was last read in P2(t-1))
    // P2(t): k-step 1 of tile t | DMA rows 0..31 of tile t+2 (last read in P1(t)) | counted vmcnt: tile t+1 has landed
    auto ktile = [&](int t, int st) {
      rd(st, 0);
      if (t >= 1 && t + 1 < nk) dma(st ^ 1, kt0 + t + 1, 1);
      JAT_TN_LOAD_END()
      JAT_TN_MMA()
      rd(st, 1);
      if (t + 2 < nk) {
        dma(st, kt0 + t + 2, 0);
        if (last_slot) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPH) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPH - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      JAT_TN_LOAD_END()
      JAT_TN_MMA()
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(t, 0);
      if (t + 1 < nk) ktile(t + 1, 1);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }
#undef JAT_TN_LOAD_END
#undef JAT_TN_MMA
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = m0 + wm * 128 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * fg;
      *(f32x4*)(out + (int64_t)o * p.ldo + n) = acc[i][j];
    }
  }
}

bool gemm_tn_supports(int M, int N) { return M % TBM == 0 && N % TBN == 0; }
// split-K slices that fill the chip: one 256 x 256 tile per CU when both sides allow it, else the 128 x 128 kernel (2 blocks/CU)
int gemm_tn_ksplit(int M, int N, int K) {
  const int nkt = (K + 63) / 64;
  const bool big = M % 256 == 0 && N % 256 == 0 && (int64_t)M * N >= 1024 * 1024;
  const int tiles = big ? (M / 256) * (N / 256) : (M / TBM) * (N / TBN), slots = big ? 256 : 512;
  int s = slots / tiles;
  if (s > nkt / 8) s = nkt / 8;     // at least 8 K-tiles per slice: the prologue, epilogue and the partial sums are not free
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}

hipError_t launch_gemm_tn(const bf16_t* dY, int64_t ldy, const bf16_t* X, int64_t ldx, float* dW, int64_t ldo, int M, int N, int K,
                          int ksplit, int64_t split_stride, const void* zeros, hipStream_t s) {
  if (!gemm_tn_supports(M, N) || K <= 0 || ldy % 8 != 0 || ldx % 8 != 0 || ldo % 4 != 0) return hipErrorInvalidValue;
  if (ksplit < 1 || ksplit > (K + TBK - 1) / TBK) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * tn256::STAGE2);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  GemmTnArgs a{};
  a.A = dY; a.lda = ldy; a.B = X; a.ldb = ldx; a.out = dW; a.ldo = ldo; a.M = M; a.N = N; a.K = K;
  a.ksplit = ksplit; a.split_stride = split_stride; a.zeros = zeros;
  static const int force = getenv("JAT_TN_TILE") ? atoi(getenv("JAT_TN_TILE")) : 0;   // 128 / 256: tests and tools
  const bool big = M % 256 == 0 && N % 256 == 0 && zeros && (force ? force == 256 : (int64_t)M * N >= 1024 * 1024);
  if (big)
    hipLaunchKernelGGL(gemm_tn256_kernel, dim3((M / 256) * (N / 256), ksplit), dim3(512), 2 * tn256::STAGE2, s, a);
  else
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((M / TBM) * (N / TBN), ksplit), dim3(256), 2 * STAGE, s, a);
  return hipGetLastError();
}

// ---- db[c] = sum_tok dY[tok][c]: column sums of a token-major bf16 matrix, fixed order ------------------------------------
// grid (C / 128, row slices): a block sums rows slice, slice + nslice, ... of 128 columns (16 lanes x 8 columns per row,
// 16 rows per pass), reduces its 16 row-lanes through LDS in order and writes part[slice][c]; colsum_finish adds the slices.
__global__ void __launch_bounds__(256) colsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int R, float* __restrict__ part,
                                                          int C) {
  __shared__ float red[16][129];
  const int c0 = blockIdx.x * 128 + (threadIdx.x & 15) * 8, rl = threadIdx.x >> 4;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = blockIdx.y * 16 + rl; r < R; r += gridDim.y * 16) {
    const u32x4 v = *(const u32x4*)(x + (int64_t)r * ld + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] += jat_lo2f(v[i]);
      acc[2 * i + 1] += jat_hi2f(v[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][(threadIdx.x & 15) * 8 + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 128) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
    part[(int64_t)blockIdx.y * C + blockIdx.x * 128 + threadIdx.x] = s;
  }
}
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float* __restrict__ part, int nslice, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nslice; ++k) s += part[(int64_t)k * C + c];
  out[c] = s;
}
int colsum_slices(int R) { return R >= 2048 ? 32 : (R >= 256 ? 8 : 1); }
// part: colsum_slices(R) * C floats of scratch
hipError_t launch_colsum_bf16(const bf16_t* x, int64_t ld, int R, int C, float* part, float* out, hipStream_t s) {
  if (C % 128 != 0 || ld % 8 != 0 || R <= 0) return hipErrorInvalidValue;
  const int ns = colsum_slices(R);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(C / 128, ns), dim3(256), 0, s, x, ld, R, part, C);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, ns, C, out);
  return hipGetLastError();
}
