// bf16 MFMA GEMM for gfx950 with fused epilogues:  C[M,N] = A[M,K] * W[N,K]^T.
//
// Both operands are K-contiguous (activations [tokens, features], nn.Linear weights [out, in]), so A and
// W fragments are fetched the same way.  Common to every variant (cdna_hip_programming.md §5, T1, T2):
//   - BK = 64; LDS image per operand = [rows][128 B], 16-B chunks XOR-swizzled by (row & 7) so that the
//     ds_read_b128 fragment reads are bank-conflict free;
//   - global -> LDS by global_load_lds_dwordx4 (1 KiB per wave-instruction, LDS image lane-linear, the
//     swizzle applied on the per-lane SOURCE address and again on the read: rule 21);
//   - v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W fragment as A, activation fragment as B):
//     the accumulator tile is then C^T, i.e. each lane owns 4 CONSECUTIVE output features of ONE row,
//     which makes every epilogue store 8-16 B wide and keeps RoPE pairs in one lane (see EPI_QKV_ROPE);
//   - XCD-aware block remap (bijective) + grouped-M tile order for L2 reuse;
//   - tile shape is a template parameter (WM x WN waves, TM x TN 16x16 MFMA tiles per wave): the model
//     picks, per GEMM, the shape whose tile count quantises best onto 256 CUs.
// Main-loop structures (template parameter PIPE; the ones measured and retired in rounds 1-2 — one barrier per K-tile
// without register double buffering, 3-stage ring, K-tile-granular ping-pong, DMA waves without ping-pong — are in the
// history and in DESIGN.md "what was tried"):
//   PIPE 1/2 4-wave blocks, two per CU: one barrier per K-tile placed MID-tile (1: MFMA clusters fenced by s_setprio,
//           2: ds_read/MFMA interleave requested with sched_group_barrier), fragments double-buffered in registers and
//           always read one k-step ahead of the MFMAs that use them (also across the tile boundary):
//             phase A: MFMA(F0: tile t, k-step 0)  ||  ds_read F1 <- tile t, k-step 1
//             s_waitcnt (F1 in regs, my DMA pieces of tile t+1 landed) ; barrier ; DMA tile t+2 -> stage of t
//             phase B: MFMA(F1)                    ||  ds_read F0 <- tile t+1, k-step 0
//           The small and medium tiles (64x128 ... 256x256) whose two co-resident blocks hide each other's epilogue.
//   PIPE 6  8 MFMA waves in K-tile-granular ping-pong + 4 DMA-only waves, 3 LDS stages (256x160 / 256x128).
//   PIPE 8  quadrant ping-pong, 8 waves, 2 LDS stages: the large tiles (224x320, 256x256, 224x256, 128x448, 256x160).
// M may be ragged (rows clamped on load, masked on store); N % BN == 0 and K % 64 == 0 are required.
#include "jat_kernels.h"
#include "jat_dtype.h"

#include <type_traits>

typedef jat_opx8 bf16x8;   // 8 operand elements (bf16, or fp16 in the -DJAT_FP16 build): one MFMA fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned short f2bf(float f) { return jat_f2op(f); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = jat_pack2(a, b);
  r.y = jat_pack2(c, d);
  return r;
}
#include "jat_gelu.h"
__device__ __forceinline__ uint2 gelu_pack4(float a, float b, float c, float d) {
  f32x2 g[2] = {f32x2{a, b}, f32x2{c, d}};
  gelu_erf_n<2>(g);
  return pack4(g[0][0], g[0][1], g[1][0], g[1][1]);
}

// RoPE rotation of one pair (a, b) by the angle with cosine c and sine s, multiply-add contraction spelled out: left to
// -ffp-contract=fast the compiler may fuse a*c - b*s either way, differently at each call site, and the fused and the separate
// QKV paths would then differ in the last fp32 bit (visible after rounding to fp16; caught by the bit-identity test).
__device__ __forceinline__ float2 rope_rot(float a, float b, float c, float s) {
  return float2{__builtin_fmaf(a, c, -(b * s)), __builtin_fmaf(b, c, a * s)};
}

// consumer side of the norm folding: 1/rms of row m of the A operand from the producer's partial sums (fixed order)
// The partials of a wave tile's rows are one contiguous block of part[M][np]: read it lane-linear (16 B per lane, whole
// cache lines) and finish the row sums with shuffles instead of gathering np floats per row (a 16-line gather per
// instruction cost ~10 us per launch).  np in {4, 8, 16}; result: rstd[i] for row i*16 + (lane&15).
// Three steps so that nothing of it sits on the kernel's critical path (round 2 ran it whole at kernel entry: 7 dependent L2
// round trips + 63 ds_bpermute BEFORE the first operand DMA was issued, 5-6 us of a 47 us block: profiles/r03/timeline_*):
//   issue   at kernel entry: the loads, as inline asm (the compiler would drain the operand DMA issued behind them at the
//           first use of a compiler-visible load result: cdna_hip_programming.md "Pipelining across barriers");
//   reduce  after the prologue DMA has been issued, behind a counted vmcnt: one value per load (in-lane sum + lane-group sum);
//   finish  after the K loop: ONE shuffle per 16-row tile (the load that holds row r = 16 i + lane % 16 is (16 i) / rpi for
//           every lane: r % 16 < 16 <= rpi) + rsqrt.
template <int TM>
__device__ __forceinline__ void rows_rstd_issue(const GemmArgs& p, int row0, int lane, f32x4 (&rsv)[TM]) {
  const int lg = p.rs_np >> 3;                     // log2 of the lanes per row (np 4 / 8 / 16 -> 1 / 2 / 4 lanes): 0, 1, 2
  const int rpi = 64 >> lg;                        // rows per 64-lane load
  // every one of the TM loads is issued, also those past the wave tile's rows when a load covers 32 or 64 rows (row clamped,
  // value unused): a load under a condition would merge with a constant, and the compiler may copy an asm load's destination
  // registers at such a merge BEFORE the data has landed (it counts them as written at the end of the asm statement)
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = min(row0 + t * rpi + (lane >> lg), p.M - 1);
    const float* ptr = p.rs_part + (int64_t)row * p.rs_np + (lane & ((1 << lg) - 1)) * 4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rsv[t]) : "v"(ptr) : "memory");
  }
}
// `younger`: vector-memory operations issued after rows_rstd_issue that may still be in flight (a compile-time count)
template <int TM, int YOUNGER>
__device__ __forceinline__ void rows_rstd_reduce(const GemmArgs& p, f32x4 (&rsv)[TM], float (&sums)[TM]) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
#pragma unroll
  for (int t = 0; t < TM; ++t) asm volatile("" : "+v"(rsv[t]));   // no use of a destination is scheduled above the wait
  const int lpr = p.rs_np >> 2;
#pragma unroll
  for (int t = 0; t < TM; ++t) {                   // sums[t]: row sum held by the lpr lanes of row t*rpi + lane/lpr
    float sq = (rsv[t][0] + rsv[t][1]) + (rsv[t][2] + rsv[t][3]);
    if (lpr >= 2) sq += __shfl_xor(sq, 1);
    if (lpr >= 4) sq += __shfl_xor(sq, 2);
    sums[t] = sq;
  }
}
template <int TM>
__device__ __forceinline__ void rows_rstd_finish(const GemmArgs& p, int lane, const float (&sums)[TM], float (&rstd)[TM]) {
  const int lg = p.rs_np >> 3, rlog = 6 - lg, frow = lane & 15;   // lanes per row = 1 << lg, rows per load = 1 << rlog
  const float invk = 1.0f / (float)p.K;
  float sq[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ti = (i * 16) >> rlog;               // wave-uniform: which load holds the rows of tile i
    float sel = sums[0];
#pragma unroll
    for (int t = 1; t < TM; ++t) sel = (t == ti) ? sums[t] : sel;   // static register index, uniform select
    sq[i] = __shfl(sel, ((i * 16 + frow) & ((1 << rlog) - 1)) << lg);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) rstd[i] = rsqrtf(sq[i] * invk + 1e-6f);
}
// 4-wave blocks with <= 20 accumulator tiles per wave are meant to run two per CU (2 waves per SIMD): cap the
// register allocation accordingly (2nd launch-bounds argument = waves per SIMD).
// PIPE 6 adds 4 DMA-only waves (one per SIMD) to the 8 MFMA waves: 768 threads, three waves per SIMD.
template <int WM, int WN, int TM, int TN, int PIPE, int CE, int EPI>
__global__ void __launch_bounds__((WM * WN + (PIPE == 6 ? 4 : 0)) * 64, (WM * WN == 4 && TM * TN <= 20) ? 2 : 1)
    gemm_bf16_kernel(const GemmArgs p_in) {
  GemmArgs p = p_in;
  // whole-kernel timeline (-DJAT_TIMELINE diagnostic build only, tools/tl_probe.py): s_memtime at entry / K loop start / K loop
  // end / after the epilogue's first barrier / exit, s_memrealtime at entry and exit; 8 words per wave, written at exit
#ifdef JAT_TIMELINE
  unsigned long long tl_[7] = {0, 0, 0, 0, 0, 0, 0}, tlx_[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tlx_: extra stamps inside an epilogue
#define JAT_TL(i)                                                                          \
  if (p.dbg_out) {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_[i])::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
#define JAT_TLX(i)                                                                         \
  if (p.dbg_out) {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlx_[i])::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
#define JAT_TLR(i)                                                                         \
  if (p.dbg_out) {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_[i])::"memory");   \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
#define JAT_TL_FLUSH()                                                                     \
  {                                                                                        \
    JAT_TL(4) JAT_TLR(6)                                                                   \
    if (p.dbg_out && (threadIdx.x & 63) == 0) {                                            \
      unsigned long long* o_ = p.dbg_out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; \
      for (int i_ = 0; i_ < 7; ++i_) o_[i_] = tl_[i_];                                     \
      o_[7] = blockIdx.x;                                                                  \
      for (int i_ = 0; i_ < 8; ++i_) o_[8 + i_] = tlx_[i_];                                \
    }                                                                                      \
  }
  JAT_TL(0) JAT_TLR(5)
#else
#define JAT_TL(i)
#define JAT_TLX(i)
#define JAT_TLR(i)
#define JAT_TL_FLUSH()
#endif
  if (p.ksplit > 1) {   // split-K slice of this block (uniform): shift the operands along K and the output to its partial
    const int z = blockIdx.y;
    p.A += (int64_t)z * p.K;
    p.W += (int64_t)z * p.K;
    p.out = (float*)p.out + (int64_t)z * p.split_stride;
  }
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 64;
  constexpr int NW = WM * WN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8, PB = BN / 8;                       // 1-KiB DMA pieces per operand per K-tile
  constexpr int AI = (PA + NW - 1) / NW, BI = (PB + NW - 1) / NW;  // pieces per wave (last may be idle)
  static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows must be a multiple of 8");
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
    const int r = min(wave + j * NW, PA - 1) * 8 + srow;
    const int gm = min(m0 + r, p.M - 1);
    a_src[j] = p.A + (int64_t)gm * p.lda + schunk * 8;
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = min(wave + j * NW, PB - 1) * 8 + srow;
    b_src[j] = p.W + (int64_t)(n0 + r) * p.ldw + schunk * 8;
  }
  auto stage = [&](int st, int kt) {
    char* sA = smem + st * STAGE;
    char* sB = sA + A_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < AI; ++j)
      if (PA % NW == 0 || wave + j * NW < PA)
        __builtin_amdgcn_global_load_lds((const void*)(a_src[j] + koff),
                                         (lds_ptr_t)(sA + min(wave + j * NW, PA - 1) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      if (PB % NW == 0 || wave + j * NW < PB)
        __builtin_amdgcn_global_load_lds((const void*)(b_src[j] + koff),
                                         (lds_ptr_t)(sB + min(wave + j * NW, PB - 1) * 1024), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fg = lane >> 4;
  const int a_row_off = (wm * TM * 16 + frow) * 128;
  const int b_row_off = (wn * TN * 16 + frow) * 128;
  const int coff0 = ((0 + fg) ^ (frow & 7)) * 16, coff1 = ((4 + fg) ^ (frow & 7)) * 16;
  // Which 16-row tile of the W image wave column w reads as its column tile j.  Plain GEMMs: wave-column-major.  Fused QKV +
  // attention (W rows group-major: 20 q tiles, 4 k tiles, 4 v tiles): every wave column gets five q tiles, ONE k tile (j = 5) and
  // ONE v tile (j = 6), so that the RoPE / operand-image phase is balanced over the waves and "which kind of tile" is a
  // compile-time property of j (the v tile is computed with swapped MFMA operands: see the epilogue).
  constexpr bool QA = EPI == EPI_QKV_ATTN;
  static_assert(!QA || (TN == 7 && WN == 4), "fused QKV + attention: 5 q + k + v tiles per wave column");
  auto ctile = [](int w, int j) { return QA ? (j < 5 ? w * 5 + j : j == 5 ? 20 + w : 24 + w) : w * TN + j; };
  auto b_frag_off = [&](int j) {   // byte offset of my fragment row of column tile j inside the W image (j: compile-time after unrolling)
    return QA ? ctile(wn, j) * 2048 + frow * 128 : b_row_off + j * 2048;
  };

  auto read_frags = [&](bf16x8(&af)[TM], bf16x8(&wf)[TN], int st, int coff) {
    const char* sA = smem + st * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(sA + a_row_off + i * 2048 + coff);
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(sB + b_row_off + j * 2048 + coff);
  };
  auto mma = [&](const bf16x8(&af)[TM], const bf16x8(&wf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = JAT_MFMA_16x16x32(wf[j], af[i], acc[i][j], 0, 0, 0);
  };

  // norm folding, consumer side: 1/rms of this lane's TM rows, loaded before the K loop so the latency is hidden
  float rstd_rows[TM], rstd_sums[TM];
  [[maybe_unused]] f32x4 rstd_raw[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rstd_rows[i] = 1.0f;
  if (p.rs_part) {
    rows_rstd_issue<TM>(p, m0 + wm * TM * 16, lane, rstd_raw);
    if constexpr (PIPE != 8) rows_rstd_reduce<TM, 0>(p, rstd_raw, rstd_sums);   // PIPE 8: behind its prologue DMA, below
  }

  const int nk = p.K / BK;
  if constexpr (PIPE == 6) {
    // Wave specialisation: waves 0-7 are the ping-pong MFMA waves of PIPE 4 but never touch global memory in the
    // K loop; waves 8-11 (one per SIMD) only issue the global->LDS DMA, spread evenly over every slot
    // (A pieces of tile j+2 in slot 2j, W pieces in slot 2j+1), each completing two slots before its first reader.
    // The per-CU DMA path moves one 1-KiB piece per ~22 cycles (tools/dma_probe.hip), so 52 pieces per K-tile fit
    // under the 2 x 40 MFMAs of the two slots only if their issue is continuous instead of bursty.
    static_assert(NW == 8 && PA % 4 == 0 && PB % 4 == 0, "PIPE 6: 8 MFMA waves + 4 DMA waves");
    constexpr int NDW = 4, AI6 = PA / NDW, BI6 = PB / NDW;
    if (wave >= NW) {
      const int dw = wave - NW;
      const bf16_t* asrc[AI6];
      const bf16_t* bsrc[BI6];
#pragma unroll
      for (int j = 0; j < AI6; ++j)
        asrc[j] = p.A + (int64_t)min(m0 + (dw + j * NDW) * 8 + srow, p.M - 1) * p.lda + schunk * 8;
#pragma unroll
      for (int j = 0; j < BI6; ++j) bsrc[j] = p.W + (int64_t)(n0 + (dw + j * NDW) * 8 + srow) * p.ldw + schunk * 8;
      auto issue_a = [&](int st, int kt) {
#pragma unroll
        for (int j = 0; j < AI6; ++j)
          __builtin_amdgcn_global_load_lds((const void*)(asrc[j] + kt * BK),
                                           (lds_ptr_t)(smem + st * STAGE + (dw + j * NDW) * 1024), 16, 0, 0);
      };
      auto issue_b = [&](int st, int kt) {
#pragma unroll
        for (int j = 0; j < BI6; ++j)
          __builtin_amdgcn_global_load_lds((const void*)(bsrc[j] + kt * BK),
                                           (lds_ptr_t)(smem + st * STAGE + A_BYTES + (dw + j * NDW) * 1024), 16, 0, 0);
      };
      issue_a(0, 0);
      issue_b(0, 0);
      if (nk > 1) {
        issue_a(1, 1);
        issue_b(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI6 + BI6) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // tile 0 visible
      int st = 2;                    // stage of tile j+2
      for (int sl = 0; sl <= 2 * nk; ++sl) {
        const int t = (sl >> 1) + 2;
        if (t < nk) {
          if (sl & 1) issue_b(st, t); else issue_a(st, t);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI6 + BI6) : "memory");  // everything older than 2 slots landed
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (sl & 1) st = st == 2 ? 0 : st + 1;
      }
      if constexpr (CE && EPI <= EPI_QKV_ROPE) __builtin_amdgcn_s_barrier();  // pairs with the epilogue's barrier
      return;
    }
    const int grp = wave >> 2;
    bf16x8 a0[TM], w0[TN], a1[TM], w1[TN];
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int cur = 0;
    // optional timeline (diagnostic build path, taken only when dbg_out is set): cycles per slot phase
    unsigned long long tl_load = 0, tl_b1 = 0, tl_mma = 0, tl_b2 = 0, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    const bool tl = p.dbg_out != nullptr;
#define JAT_STAMP(x)                                                                  \
  if (tl) {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                \
  }
    JAT_STAMP(ts0)
    for (int kt = 0; kt < nk; ++kt) {
      read_frags(a0, w0, cur, coff0);
      read_frags(a1, w1, cur, coff1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      JAT_STAMP(ts1)
      __builtin_amdgcn_s_barrier();
      JAT_STAMP(ts2)
      __builtin_amdgcn_s_setprio(1);
      mma(a0, w0);
      mma(a1, w1);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      JAT_STAMP(ts3)
      __builtin_amdgcn_s_barrier();
      if (tl) {
        unsigned long long ts4;
        JAT_STAMP(ts4)
        tl_load += ts1 - ts0; tl_b1 += ts2 - ts1; tl_mma += ts3 - ts2; tl_b2 += ts4 - ts3;
        ts0 = ts4;
      }
      cur = cur == 2 ? 0 : cur + 1;
    }
#undef JAT_STAMP
    if (tl && lane == 0) {
      unsigned long long* o = p.dbg_out + ((size_t)blockIdx.x * NW + wave) * 4;
      o[0] = tl_load; o[1] = tl_b1; o[2] = tl_mma; o[3] = tl_b2;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else if constexpr (PIPE == 8) {
    // Quadrant ping-pong (the "8-phase" structure of cdna_hip_programming.md §5, generalised to any TM x TN wave tile and to
    // uneven halves): the wave tile is cut into an A split (rows: TMa + TMb 16-row MFMA tiles; one part when SPLIT_M is off)
    // and a B split (columns: TNa + TNb); one PHASE = the fragment reads of ONE part (both k-steps of the K-tile) + one
    // unit of global->LDS DMA, s_barrier, the MFMAs of one quadrant, s_barrier.  The two waves that share a SIMD (w, w+4)
    // run ONE barrier apart, so a quadrant's MFMAs always overlap the partner's LDS reads / DMA issue, and a wave holds
    // one A part + one B part of fragments (<= 56 VGPRs) instead of a double-buffered K-tile (96): a 7 x 5 wave tile
    // (224 x 320 block, two exact rounds of 256 CUs for M = 7168, N = 5120) fits 256 registers.
    //   quadrant order (serpentine, one operand part changes per phase):  (A0,B0) (A0,B1) (A1,B1) (A1,B0)
    //   2 LDS stages (K-tile parity); a part's DMA for K-tile t+2 is issued ONE phase after its last read in K-tile t:
    //     P1(t): B0(t+1)   P2(t): A0(t+2)   P3(t): B1(t+2)   P4(t): A1(t+2)  + counted vmcnt: K-tile t+1 has landed
    //   (SPLIT_M off:  P1(t): B1(t+1)   P2(t): A(t+2), B0(t+2) + counted vmcnt.)
    // Hazards (MI355X_MICROARCH "Two waves per SIMD" item 7; guide "Read a staged buffer one phase AFTER the wait"):
    //   RAW  every wave's counted vmcnt sits before the first barrier of the K-tile's last phase; the first read of K-tile
    //        t+1 is after that phase's second barrier (one more barrier for the trailing wave group);
    //   WAR  fragment reads are retired (lgkmcnt(0)) BEFORE the phase's first barrier, the overwrite is issued after its
    //        second barrier by either group.
    static_assert(NW == 8, "quadrant ping-pong needs two waves per SIMD in one block");
    constexpr bool SPLIT_M = TM * TN > 20;
    constexpr int TMa = SPLIT_M ? (TM + 1) / 2 : TM, TMb = TM - TMa, TNa = (TN + 1) / 2, TNb = TN - TNa;
    // DMA units: 8-row pieces of each part, dealt round-robin to the 8 waves (the last pieces are duplicated so that every
    // wave issues the same number of DMAs per unit: the vmcnt immediates are then compile-time constants)
    constexpr int PA0 = WM * TMa * 2, PA1 = WM * TMb * 2, PB0 = WN * TNa * 2, PB1 = WN * TNb * 2;
    constexpr int CA0 = (PA0 + 7) / 8, CA1 = (PA1 + 7) / 8, CB0 = (PB0 + 7) / 8, CB1 = (PB1 + 7) / 8;
    constexpr int CTILE = CA0 + CA1 + CB0 + CB1;
    const int grp = wave >> 2;
    const char* a_base = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* b_base = (const char*)(p.W + (int64_t)n0 * p.ldw);
    // per-lane byte offsets (32-bit, relative to the tile's first row) and wave-uniform LDS offsets of my pieces
    unsigned oa0[CA0], oa1[CA1 > 0 ? CA1 : 1], ob0[CB0], ob1[CB1];
    int la0[CA0], la1[CA1 > 0 ? CA1 : 1], lb0[CB0], lb1[CB1];
    auto piece_row = [&](int q, int per, int tiles, int first) {   // piece q of a part: per = 2 * tiles-in-part pieces per wave row/col
      const int w = q / per, in = q - w * per;
      return w * tiles * 16 + first * 16 + in * 8;
    };
#pragma unroll
    for (int j = 0; j < CA0; ++j) {
      const int r = piece_row(min(wave + 8 * j, PA0 - 1), TMa * 2, TM, 0);
      oa0[j] = (unsigned)((min(m0 + r + srow, p.M - 1) - m0) * (int)p.lda * 2 + schunk * 16);
      la0[j] = r * 128;
    }
#pragma unroll
    for (int j = 0; j < CA1; ++j) {
      const int r = piece_row(min(wave + 8 * j, PA1 - 1), TMb * 2, TM, TMa);
      oa1[j] = (unsigned)((min(m0 + r + srow, p.M - 1) - m0) * (int)p.lda * 2 + schunk * 16);
      la1[j] = r * 128;
    }
    auto piece_row_b = [&](int q, int per, int first) {   // the same for a B part, through the column-tile map
      const int w = q / per, in = q - w * per;
      return ctile(w, first + (in >> 1)) * 16 + (in & 1) * 8;
    };
#pragma unroll
    for (int j = 0; j < CB0; ++j) {
      const int r = piece_row_b(min(wave + 8 * j, PB0 - 1), TNa * 2, 0);
      ob0[j] = (unsigned)((r + srow) * (int)p.ldw * 2 + schunk * 16);
      lb0[j] = A_BYTES + r * 128;
    }
#pragma unroll
    for (int j = 0; j < CB1; ++j) {
      const int r = piece_row_b(min(wave + 8 * j, PB1 - 1), TNb * 2, TNa);
      ob1[j] = (unsigned)((r + srow) * (int)p.ldw * 2 + schunk * 16);
      lb1[j] = A_BYTES + r * 128;
    }
    auto dma_a0 = [&](int st, int kt) {
#pragma unroll
      for (int j = 0; j < CA0; ++j)
        __builtin_amdgcn_global_load_lds((const void*)(a_base + kt * 128 + oa0[j]), (lds_ptr_t)(smem + st * STAGE + la0[j]), 16, 0, 0);
    };
    auto dma_a1 = [&](int st, int kt) {
#pragma unroll
      for (int j = 0; j < CA1; ++j)
        __builtin_amdgcn_global_load_lds((const void*)(a_base + kt * 128 + oa1[j]), (lds_ptr_t)(smem + st * STAGE + la1[j]), 16, 0, 0);
    };
    auto dma_b0 = [&](int st, int kt) {
#pragma unroll
      for (int j = 0; j < CB0; ++j)
        __builtin_amdgcn_global_load_lds((const void*)(b_base + kt * 128 + ob0[j]), (lds_ptr_t)(smem + st * STAGE + lb0[j]), 16, 0, 0);
    };
    auto dma_b1 = [&](int st, int kt) {
#pragma unroll
      for (int j = 0; j < CB1; ++j)
        __builtin_amdgcn_global_load_lds((const void*)(b_base + kt * 128 + ob1[j]), (lds_ptr_t)(smem + st * STAGE + lb1[j]), 16, 0, 0);
    };
    bf16x8 fa[2][TMa], fb[2][TNa];
    auto rd_a = [&](int st, int i0, int cnt) {
      const char* sA = smem + st * STAGE + a_row_off + i0 * 2048;
#pragma unroll
      for (int i = 0; i < TMa; ++i)
        if (i < cnt) {
          fa[0][i] = *(const bf16x8*)(sA + i * 2048 + coff0);
          fa[1][i] = *(const bf16x8*)(sA + i * 2048 + coff1);
        }
    };
    auto rd_b = [&](int st, int j0, int cnt) {
      if constexpr (QA) {
        const char* sB = smem + st * STAGE + A_BYTES;
#pragma unroll
        for (int j = 0; j < TNa; ++j)
          if (j < cnt) {
            fb[0][j] = *(const bf16x8*)(sB + b_frag_off(j0 + j) + coff0);
            fb[1][j] = *(const bf16x8*)(sB + b_frag_off(j0 + j) + coff1);
          }
      } else {
        const char* sB = smem + st * STAGE + A_BYTES + b_row_off + j0 * 2048;
#pragma unroll
        for (int j = 0; j < TNa; ++j)
          if (j < cnt) {
            fb[0][j] = *(const bf16x8*)(sB + j * 2048 + coff0);
            fb[1][j] = *(const bf16x8*)(sB + j * 2048 + coff1);
          }
      }
    };
    // fused QKV + attention: the v tile (j = TN - 1) with the operands swapped, D[token][feature]: a lane then holds 4 consecutive
    // KEYS of one feature, which is 8 contiguous bytes of the V^T operand image
#define JAT_Q(I0, IC, J0, JC)                                                                                   \
  if (!abl_mma) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
  _Pragma("unroll") for (int i = 0; i < (IC); ++i)                                                              \
  _Pragma("unroll") for (int j = 0; j < (JC); ++j)                                                              \
    acc[(I0) + i][(J0) + j] = (QA && (J0) + j == TN - 1)                                                        \
        ? JAT_MFMA_16x16x32(fa[ks][i], fb[ks][j], acc[(I0) + i][(J0) + j], 0, 0, 0)                             \
        : JAT_MFMA_16x16x32(fb[ks][j], fa[ks][i], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define JAT_LOAD_END()                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);                \
  __builtin_amdgcn_s_barrier();                     \
  __builtin_amdgcn_s_setprio(1);
#define JAT_MMA_END()                 \
  __builtin_amdgcn_s_setprio(0);      \
  __builtin_amdgcn_sched_barrier(0);  \
  __builtin_amdgcn_s_barrier();
    // prologue: K-tiles 0 and 1 in flight, tile 0 landed
    dma_a0(0, 0); dma_b0(0, 0); dma_b1(0, 0); dma_a1(0, 0);
    if (nk > 1) {
      dma_a0(1, 1); dma_b0(1, 1); dma_b1(1, 1); dma_a1(1, 1);
      if (p.rs_part) rows_rstd_reduce<TM, 2 * CTILE>(p, rstd_raw, rstd_sums);   // the row-statistic loads are older than both tiles
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CTILE) : "memory");
    } else {
      if (p.rs_part) rows_rstd_reduce<TM, CTILE>(p, rstd_raw, rstd_sums);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    static_assert(2 * CTILE <= 63, "vmcnt immediate");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    JAT_TL(1)
#ifdef JAT_ABLATE   // timing ablations (wrong results; -DJAT_ABLATE builds only): dbg bit 1: no DMA after the prologue, bit 2: no fragment reads
    const bool abl_dma = p.dbg & 2, abl_rd = p.dbg & 4, abl_mma = p.dbg & 8;   // after K-tile 0, bit 3: no MFMAs
#else
    constexpr bool abl_dma = false, abl_rd = false, abl_mma = false;
#endif
    auto ktile = [&](int t, int st) {
      const bool more2 = t + 2 < nk && !abl_dma;
      const bool more1 = t >= 1 && t + 1 < nk && !abl_dma;
      const bool rd = !abl_rd || t == 0;
      if constexpr (SPLIT_M) {
        // P1 (A0, B0)
        if (rd) rd_b(st, 0, TNa);
        __builtin_amdgcn_sched_barrier(0);
        if (rd) rd_a(st, 0, TMa);
        if (more1) dma_b0(st ^ 1, t + 1);
        JAT_LOAD_END()
        JAT_Q(0, TMa, 0, TNa)
        JAT_MMA_END()
        // P2 (A0, B1)
        if (rd) rd_b(st, TNa, TNb);
        if (more2) dma_a0(st, t + 2);
        JAT_LOAD_END()
        JAT_Q(0, TMa, TNa, TNb)
        JAT_MMA_END()
        // P3 (A1, B1)
        if (rd) rd_a(st, TMa, TMb);
        if (more2) dma_b1(st, t + 2);
        JAT_LOAD_END()
        JAT_Q(TMa, TMb, TNa, TNb)
        JAT_MMA_END()
        // P4 (A1, B0)
        if (rd) rd_b(st, 0, TNa);
        if (more2) {
          dma_a1(st, t + 2);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA0 + CB1 + CA1) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        JAT_LOAD_END()
        JAT_Q(TMa, TMb, 0, TNa)
        JAT_MMA_END()
      } else {
        // P1 (A, B0)
        if (rd) rd_b(st, 0, TNa);
        __builtin_amdgcn_sched_barrier(0);
        if (rd) rd_a(st, 0, TMa);
        if (more1) dma_b1(st ^ 1, t + 1);
        JAT_LOAD_END()
        JAT_Q(0, TMa, 0, TNa)
        JAT_MMA_END()
        // P2 (A, B1)
        if (rd) rd_b(st, TNa, TNb);
        if (more2) {
          dma_a0(st, t + 2);
          dma_b0(st, t + 2);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA0 + CB0) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        JAT_LOAD_END()
        JAT_Q(0, TMa, TNa, TNb)
        JAT_MMA_END()
      }
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(t, 0);
      if (t + 1 < nk) ktile(t + 1, 1);
    }
#undef JAT_Q
#undef JAT_LOAD_END
#undef JAT_MMA_END
    JAT_TL(2)
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
    // interleave hint: one fragment read, then MPR MFMAs, ... (sched_group_barrier masks: MFMA 0x8, DS read 0x100)
    constexpr int NREAD = TM + TN, NMMA = TM * TN, MPR = NMMA / NREAD;
    auto interleave = [&]() {
#pragma unroll
      for (int r = 0; r < NREAD; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMMA - MPR * NREAD, 0);
    };
    bf16x8 a0[TM], w0[TN], a1[TM], w1[TN];
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) stage(1, 1);
    read_frags(a0, w0, 0, coff0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      // phase A: MFMAs of k-step 0 while the fragments of k-step 1 stream in
      read_frags(a1, w1, cur, coff1);
      if constexpr (PIPE == 1) __builtin_amdgcn_s_setprio(1);
      mma(a0, w0);
      if constexpr (PIPE == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (PIPE == 2) interleave();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of tile kt+1 have landed
      __syncthreads();                                   // everyone holds F1 in registers: stage `cur` is free
      if (kt + 2 < nk) stage(cur, kt + 2);
      // phase B: MFMAs of k-step 1 while the first fragments of tile kt+1 stream in (harmless when kt+1 == nk)
      read_frags(a0, w0, cur ^ 1, coff0);
      if constexpr (PIPE == 1) __builtin_amdgcn_s_setprio(1);
      mma(a1, w1);
      if constexpr (PIPE == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (PIPE == 2) interleave();
    }
  }

  if (p.rs_part) rows_rstd_finish<TM>(p, lane, rstd_sums, rstd_rows);
  const int nw0 = n0 + wn * TN * 16;  // wave-uniform first column

  // ---- split-residual epilogue (sampler with folded norms).  The residual stream lives as TWO bf16 planes, x = hi + lo with
  // hi = bf16(x), lo = bf16(x - hi): 16 significant bits (an update costs 2^-17 relative, two orders below the bf16 rounding of
  // the GEMM operands), the same 4 bytes per element as fp32 — and `hi` IS the bf16 A operand of the next GEMM (whose
  // folded weights carry the norm weight and the adaLN scale), so the epilogue writes no extra copy and no norm kernel runs.
  //   EPI_RESID: x_new = (hi + lo) + gate[b][n] * (acc + bias)      EPI_F32: x_new = acc + bias
  // plus the row partial sums of x_new^2 over this wave's columns (the consumer's rstd), in fixed order.
  // Accumulators go through the wave-private LDS slab so that every global access is 16 B per lane along a row.
  if constexpr (CE && (EPI == EPI_RESID || EPI == EPI_F32)) {
    if (p.fold_out != nullptr) {
      constexpr int RS = TN * 64 + 16;         // slab row: TN*16 fp32 + pad
      constexpr int CPR8 = TN * 2;             // 8-element chunks per row
      constexpr int NCH8 = 32 * CPR8 / 64;     // chunks per lane per 32-row group (= TN)
      static_assert(NW * 32 * RS <= 2 * STAGE, "epilogue slab does not fit the staging buffers");
      __builtin_amdgcn_s_barrier();            // every wave is done reading the staging buffers
      JAT_TL(3)
      if (p.dbg & 1) { JAT_TL_FLUSH() return; }
      char* wbuf = smem + wave * (32 * RS);
      const int mw0 = m0 + wm * TM * 16;
      float4 bb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bb[j] = p.bias ? *(const float4*)(p.bias + nw0 + j * 16 + fg * 4) : float4{0.f, 0.f, 0.f, 0.f};
      auto up = [](unsigned u, float& a, float& b) { a = jat_lo2f(u); b = jat_hi2f(u); };
#pragma unroll
      for (int ig = 0; ig < (TM + 1) / 2; ++ig) {
        const int grows = (2 * ig + 1 < TM) ? 32 : 16;
        // residual / gate loads of the whole group first (independent of the slab): in flight under the slab writes
        [[maybe_unused]] uint4 hi[NCH8], lo[NCH8];
        [[maybe_unused]] float4 g0[NCH8], g1[NCH8];
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
          for (int t = 0; t < NCH8; ++t) {
            const int c = lane + 64 * t, row = min(c / CPR8, grows - 1), cc = c - (c / CPR8) * CPR8;
            const int m = min(mw0 + ig * 32 + row, p.M - 1), n = nw0 + cc * 8;
            hi[t] = *(const uint4*)(p.fold_out + (int64_t)m * p.ldo + n);
            lo[t] = *(const uint4*)(p.fold_lo + (int64_t)m * p.ldo + n);
            const float* gp = p.gate + (int64_t)(m / p.ntok) * p.gate_bstride + n;
            g0[t] = *(const float4*)gp;
            g1[t] = *(const float4*)(gp + 4);
          }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < TN; ++j) if (2 * ig + ii < TM) {
            f32x4 v = acc[2 * ig + ii < TM ? 2 * ig + ii : 0][j] * rstd_rows[2 * ig + ii < TM ? 2 * ig + ii : 0];
            *(float4*)(wbuf + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * 4) =
                float4{v[0] + bb[j].x, v[1] + bb[j].y, v[2] + bb[j].z, v[3] + bb[j].w};
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NCH8; ++t) {
          const int c = lane + 64 * t, row = c / CPR8, cc = c - row * CPR8;
          const int m = mw0 + ig * 32 + row, n = nw0 + cc * 8;
          char* slot = wbuf + (row < 32 ? row : 0) * RS + cc * 32;
          const float4 a0 = *(const float4*)slot, a1 = *(const float4*)(slot + 16);
          float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          if constexpr (EPI == EPI_RESID) {
            const unsigned hw[4] = {hi[t].x, hi[t].y, hi[t].z, hi[t].w}, lw[4] = {lo[t].x, lo[t].y, lo[t].z, lo[t].w};
            const float gg[8] = {g0[t].x, g0[t].y, g0[t].z, g0[t].w, g1[t].x, g1[t].y, g1[t].z, g1[t].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float h0, h1, l0, l1;
              up(hw[e], h0, h1);
              up(lw[e], l0, l1);
              x[2 * e] = __builtin_fmaf(gg[2 * e], x[2 * e], h0 + l0);          // spelled out: left to -ffp-contract the two forms
              x[2 * e + 1] = __builtin_fmaf(gg[2 * e + 1], x[2 * e + 1], h1 + l1);  // of this epilogue could round differently
            }
          }
          unsigned ho[4], lw2[4];
          float sq = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned short ha = f2bf(x[2 * e]), hb = f2bf(x[2 * e + 1]);
            ho[e] = (unsigned)ha | ((unsigned)hb << 16);
            const float ra = x[2 * e] - jat_op2f(ha), rb = x[2 * e + 1] - jat_op2f(hb);
            lw2[e] = jat_pack2(ra, rb);
            sq += x[2 * e] * x[2 * e] + x[2 * e + 1] * x[2 * e + 1];
          }
          const bool live = row < grows && m < p.M;
          if (live && !(p.dbg & 128)) {
            *(uint4*)(p.fold_out + (int64_t)m * p.ldo + n) = uint4{ho[0], ho[1], ho[2], ho[3]};
            *(uint4*)(p.fold_lo + (int64_t)m * p.ldo + n) = uint4{lw2[0], lw2[1], lw2[2], lw2[3]};
          }
          if (row < 32) *(float*)slot = live ? sq : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {  // row partial sums over this wave's columns: two lanes per row, fixed order
          constexpr int HALF = CPR8 / 2;
          const int r = lane >> 1, h = lane & 1;
          float sq = 0.f;
#pragma unroll
          for (int cc = 0; cc < HALF; ++cc) sq += *(const float*)(wbuf + r * RS + (h * HALF + cc) * 32);
          sq += __shfl_xor(sq, 1);
          const int m = mw0 + ig * 32 + r;
          if (h == 0 && m < p.M && r < grows) p.fold_part[(int64_t)m * p.fold_np + nw0 / (TN * 16)] = sq;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      JAT_TL_FLUSH()
      return;
    }
  }

  // ---- coalesced epilogue (CE): accumulators -> wave-private LDS slab (32 rows at a time) -> each lane
  // reads back 16 B that are CONTIGUOUS along n, so global loads/stores cover whole rows of the wave tile
  // (full 128-B lines) instead of 8-16 B per lane at a row stride.
  if constexpr (CE && EPI <= EPI_RESID) {
    // the wave tile is walked 32 rows at a time; an odd TM leaves a 16-row tail group (rows >= grows are skipped)
    constexpr bool OUT32 = (EPI == EPI_F32 || EPI == EPI_RESID);
    constexpr int EB = OUT32 ? 4 : 2;        // bytes per output element
    constexpr int EPC = 16 / EB;             // elements per 16-B chunk
    constexpr int RS = TN * 16 * EB + 16;    // padded LDS row stride (bytes)
    constexpr int CPR = TN * 16 / EPC;       // chunks per row
    constexpr int NCH = (32 * CPR + 63) / 64;  // chunks per lane per 32-row group
    static_assert(NW * 32 * RS <= 2 * STAGE, "epilogue slab does not fit the staging buffers");
    __builtin_amdgcn_s_barrier();             // every wave is done reading the staging buffers
    JAT_TL(3)
    if (p.dbg & 1) { JAT_TL_FLUSH() return; }
    char* wbuf = smem + wave * (32 * RS);
    const int mw0 = m0 + wm * TM * 16;
    float4 bb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bb[j] = p.bias ? *(const float4*)(p.bias + nw0 + j * 16 + fg * 4) : float4{0.f, 0.f, 0.f, 0.f};
    const int npass = (EPI == EPI_BF16_GELU && p.dual_rows > 0) ? 2 : 1;
    // ---- software-pipelined form (CE == 2; bf16 / GELU outputs): the epilogue above runs per 32-row group as
    //   [VALU: bias, GELU, pack -> slab] [slab -> 16-B global stores]
    // and an in-order wave that meets a full store queue cannot start the next group's VALU: the vector work (GELU: 15
    // multiply-adds per element) and the store drain add up instead of overlapping (profiles/r02/epilogue_cost.log: +11.6 us
    // for the stores alone, +22.9 us with GELU).  Here the stores of group g-1 are issued ONE AT A TIME between the column
    // tiles of group g's vector work (two slabs per wave, alternating), so the store path drains under the VALU of the same
    // wave and of its SIMD partner.  Same arithmetic, same bytes, same addresses.
    if constexpr (CE == 2 && (EPI == EPI_BF16_GELU || EPI == EPI_BF16)) {
      if (npass == 1) {
        constexpr int G = (TM + 1) / 2;
        static_assert(NCH == TN && NW * 64 * RS <= 2 * STAGE, "pipelined epilogue: two slabs per wave, TN chunks per lane");
        char* const slab0 = wbuf;
        char* const slab1 = smem + (NW + wave) * (32 * RS);
        int soff[NCH], goff[NCH], rowt[NCH];
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
          const int c = lane + 64 * t;
          rowt[t] = c / CPR;
          const int cc = c - rowt[t] * CPR;
          soff[t] = rowt[t] * RS + cc * 16;
          goff[t] = rowt[t] * (int)p.ldo * 2 + cc * 16;
        }
        u32x4 raw[NCH];
        // compile-time group / tile indices throughout (static_for): a runtime index into acc / raw / slab would put them in scratch
        auto vwrite = [&](auto gc, auto jc) __attribute__((always_inline)) {   // vector work of column tile j of group g -> slab[g & 1]
          constexpr int g = decltype(gc)::value, j = decltype(jc)::value;
          constexpr int NI = (2 * g + 1 < TM) ? 2 : 1;   // 16-row tiles in this group
          f32x2 h[2 * NI];
#pragma unroll
          for (int ii = 0; ii < NI; ++ii) {
            f32x4 v = acc[2 * g + ii][j] * rstd_rows[2 * g + ii];
            h[2 * ii] = f32x2{v[0] + bb[j].x, v[1] + bb[j].y};
            h[2 * ii + 1] = f32x2{v[2] + bb[j].z, v[3] + bb[j].w};
          }
          if constexpr (EPI == EPI_BF16_GELU) gelu_erf_n<2 * NI>(h);   // the chains of both tiles interleaved
#pragma unroll
          for (int ii = 0; ii < NI; ++ii)
            *(uint2*)((g & 1 ? slab1 : slab0) + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * 2) =
                pack4(h[2 * ii][0], h[2 * ii][1], h[2 * ii + 1][0], h[2 * ii + 1][1]);
        };
        auto sread = [&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          static_for<0, NCH>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            raw[t] = *(const u32x4*)((g & 1 ? slab1 : slab0) + (rowt[t] < 32 ? soff[t] : 0));
          });
        };
        auto gstore = [&](auto gc, auto tc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value, t = decltype(tc)::value;
          constexpr int grows = (2 * g + 1 < TM) ? 32 : 16;
          const int m = mw0 + g * 32 + rowt[t];
          if (rowt[t] < grows && m < p.M && !(p.dbg & 128))
            *(u32x4*)((char*)p.out + ((int64_t)(mw0 + g * 32) * p.ldo + nw0) * 2 + goff[t]) = raw[t];
        };
        static_for<0, TN>([&](auto jc) __attribute__((always_inline)) { vwrite(std::integral_constant<int, 0>{}, jc); });
        static_for<1, G>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          // same-wave LDS accesses execute in order: the slab of group g-1 is complete, and slab[g & 1] (read in iteration g-1,
          // consumed by its stores) is free to be rewritten
          sread(std::integral_constant<int, g - 1>{});
          static_for<0, TN>([&](auto jc) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            vwrite(gc, jc);
            __builtin_amdgcn_sched_barrier(0);
            gstore(std::integral_constant<int, g - 1>{}, jc);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        sread(std::integral_constant<int, G - 1>{});
        static_for<0, NCH>([&](auto tc) __attribute__((always_inline)) { gstore(std::integral_constant<int, G - 1>{}, tc); });
        JAT_TL_FLUSH()
        return;
      }
    }
    for (int pass = 0; pass < npass; ++pass)
#pragma unroll
    for (int ig = 0; ig < (TM + 1) / 2; ++ig) {
      const int grows = (2 * ig + 1 < TM) ? 32 : 16;   // folds: ig is an unrolled constant
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j) if (2 * ig + ii < TM) {
          f32x4 v = acc[2 * ig + ii < TM ? 2 * ig + ii : 0][j] * rstd_rows[2 * ig + ii < TM ? 2 * ig + ii : 0];
          v[0] += bb[j].x; v[1] += bb[j].y; v[2] += bb[j].z; v[3] += bb[j].w;
          if constexpr (EPI == EPI_BF16_GELU) {
            if (npass == 2 && pass == 0) {
              const int mrow = min(mw0 + (2 * ig + ii) * 16 + frow, p.M - 1);
              const float4 da = *(const float4*)(p.dual_add + (int64_t)mrow * p.N + nw0 + j * 16 + fg * 4);
              v[0] += da.x; v[1] += da.y; v[2] += da.z; v[3] += da.w;
            }
          }
          char* dst = wbuf + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * EB;
          if constexpr (OUT32) {
            *(float4*)dst = float4{v[0], v[1], v[2], v[3]};
          } else if constexpr (EPI == EPI_BF16_GELU) {
            *(uint2*)dst = gelu_pack4(v[0], v[1], v[2], v[3]);
          } else {
            *(uint2*)dst = pack4(v[0], v[1], v[2], v[3]);
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // EPI_RESID: issue ALL residual / gate loads of the group first.  Written chunk by chunk, the store of chunk t
      // and the load of chunk t+1 hit the same buffer, the compiler cannot prove them disjoint and serialises
      // 10 global round trips per group (measured: ~13 us of a 34 us out_proj launch).
      constexpr int NHALF = (EPI == EPI_RESID) ? 2 : 1, HCH = NCH / NHALF;   // two batches: 40 VGPRs of loads in flight
      static_assert(NCH % NHALF == 0, "chunk count must split evenly");
#pragma unroll
      for (int hh = 0; hh < NHALF; ++hh) {
      [[maybe_unused]] f32x4 xs[HCH], gs[HCH];
      if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int tt = 0; tt < HCH; ++tt) {
          const int c = lane + 64 * (hh * HCH + tt), row = min(c / CPR, grows - 1), cc = c - (c / CPR) * CPR;
          const int m = min(mw0 + ig * 32 + row, p.M - 1), n = nw0 + cc * EPC;
          xs[tt] = *(const f32x4*)((const float*)p.out + (int64_t)m * p.ldo + n);
          gs[tt] = *(const f32x4*)(p.gate + (int64_t)(m / p.ntok) * p.gate_bstride + n);
        }
      }
#pragma unroll
      for (int tt = 0; tt < HCH; ++tt) {
        const int t = hh * HCH + tt;
        const int c = lane + 64 * t, row = c / CPR, cc = c - row * CPR;
        const int m = mw0 + ig * 32 + row, n = nw0 + cc * EPC;
        const uint4 raw = *(const uint4*)(wbuf + (row < 32 ? row : 0) * RS + cc * 16);
        if (row >= grows) continue;
        if (m < p.M && !(p.dbg & 128)) {
          if constexpr (EPI == EPI_RESID) {
            float4 x;
            x.x = xs[tt][0] + gs[tt][0] * __uint_as_float(raw.x); x.y = xs[tt][1] + gs[tt][1] * __uint_as_float(raw.y);
            x.z = xs[tt][2] + gs[tt][2] * __uint_as_float(raw.z); x.w = xs[tt][3] + gs[tt][3] * __uint_as_float(raw.w);
            *(float4*)((float*)p.out + (int64_t)m * p.ldo + n) = x;
          } else if constexpr (EPI == EPI_F32) {
            *(uint4*)((float*)p.out + (int64_t)m * p.ldo + n) = raw;
          } else {
            *(uint4*)((bf16_t*)p.out + (int64_t)(m + pass * p.dual_rows) * p.ldo + n) = raw;
          }
        }
      }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    JAT_TL_FLUSH()
    return;
  }

  // ---- coalesced QKV epilogue: RoPE in registers, q/k through the LDS slab (16-B row-contiguous stores into
  // the q / k buffers), v tiles stored transposed directly (runs of 16 tokens per feature).
  if constexpr (CE && EPI == EPI_QKV_ROPE && TM % 2 == 0) {   // odd TM: direct epilogue below
    constexpr int RS = TN * 32 + 16, CPR = TN * 2, NCH = 32 * CPR / 64;
    static_assert(NW * 32 * RS <= 2 * STAGE, "epilogue slab does not fit the staging buffers");
    __builtin_amdgcn_s_barrier();
    if (p.dbg & 1) return;
    char* wbuf = smem + wave * (32 * RS);
    const int mw0 = m0 + wm * TM * 16;
    const int nqk = p.D + p.kvD;
    const bool vfast = (p.ntok & 7) == 0;  // 8-token runs stay inside one sample and are 16-B aligned in vt
    // RoPE angles in registers: a lane needs (pos, d) for ONE row and two frequencies per MFMA tile, so a table
    // lookup is a 16-cache-line gather per instruction (measured 10 us of a 53 us launch); v_sin/v_cos of
    // fract(pos * inv_freq / 2pi) costs ~1 us instead.  inv_freq is the reference's fp32 value (:77).
    float2 invf[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int d0 = (((nw0 + j * 16) & 63) >> 1) + fg * 2;
      invf[j] = *(const float2*)(p.rope_inv_freq + d0);
    }

#pragma unroll
    for (int ig = 0; ig < TM / 2; ++ig) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = min(mw0 + (2 * ig + ii) * 16 + frow, p.M - 1);
        const int b = m / p.ntok, pos = m - b * p.ntok;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nt = nw0 + j * 16;
          f32x4 v = acc[2 * ig + ii][j] * rstd_rows[2 * ig + ii];
          if (p.bias) {
            const float4 bq = *(const float4*)(p.bias + nt + fg * 4);
            v[0] += bq.x; v[1] += bq.y; v[2] += bq.z; v[3] += bq.w;
          }
          if (nt < nqk) {
            const float r0 = __builtin_amdgcn_fractf((float)pos * invf[j].x * 0.15915494309189535f);
            const float r1 = __builtin_amdgcn_fractf((float)pos * invf[j].y * 0.15915494309189535f);
            const float2 c = float2{__builtin_amdgcn_cosf(r0), __builtin_amdgcn_cosf(r1)};
            const float2 s = float2{__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1)};
            const float2 r01 = rope_rot(v[0], v[1], c.x, s.x), r23 = rope_rot(v[2], v[3], c.y, s.y);
            *(uint2*)(wbuf + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * 2) = pack4(r01.x, r01.y, r23.x, r23.y);
          } else if (vfast) {  // v tile: plain bf16 into the slab, transposed out below
            *(uint2*)(wbuf + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * 2) = pack4(v[0], v[1], v[2], v[3]);
          } else if (mw0 + (2 * ig + ii) * 16 + frow < p.M) {  // ragged token count: element-wise transposed store
            const int nv = nt + fg * 4 - nqk;
            bf16_t* dst = p.vt_out + ((int64_t)(b * (p.kvD >> 6) + (nv >> 6)) * 64 + (nv & 63)) * p.npad + pos;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.npad] = f2bf(v[r]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // v tiles: lane = (feature d = lane&15, token group tg = lane>>4): gather 8 consecutive tokens of one
      // feature down a slab column and store them as ONE 16-B run of vt[b][hv][d][pos..pos+7]
      if (vfast) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nt = nw0 + j * 16;
          if (nt >= nqk) {
            const int mg = mw0 + ig * 32 + fg * 8;  // first of this lane's 8 tokens
            const char* col = wbuf + (fg * 8) * RS + (j * 16 + frow) * 2;
            uint4 pk;
            pk.x = (unsigned)*(const unsigned short*)(col + 0 * RS) | ((unsigned)*(const unsigned short*)(col + 1 * RS) << 16);
            pk.y = (unsigned)*(const unsigned short*)(col + 2 * RS) | ((unsigned)*(const unsigned short*)(col + 3 * RS) << 16);
            pk.z = (unsigned)*(const unsigned short*)(col + 4 * RS) | ((unsigned)*(const unsigned short*)(col + 5 * RS) << 16);
            pk.w = (unsigned)*(const unsigned short*)(col + 6 * RS) | ((unsigned)*(const unsigned short*)(col + 7 * RS) << 16);
            if (mg < p.M && !(p.dbg & 16)) {  // M % 8 == 0 on this path, so the 8 tokens are valid together
              const int b = mg / p.ntok, pos = mg - b * p.ntok, nv = nt + frow - nqk;
              *(uint4*)(p.vt_out + ((int64_t)(b * (p.kvD >> 6) + (nv >> 6)) * 64 + (nv & 63)) * p.npad + pos) = pk;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NCH; ++t) {
        const int c = lane + 64 * t, row = c / CPR, cc = c - row * CPR;
        const int m = mw0 + ig * 32 + row, n = nw0 + cc * 8;
        const uint4 raw = *(const uint4*)(wbuf + row * RS + cc * 16);
        if (m < p.M && n < nqk) {
          bf16_t* dst = (n < p.D) ? ((bf16_t*)p.out + (int64_t)m * p.D + n) : (p.k_out + (int64_t)m * p.kvD + (n - p.D));
          *(uint4*)dst = raw;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }

  // ---- fused QKV projection + RoPE + GQA attention (ntok == 128): the 128 x 448 tile is ALL of q (5 heads), k, v
  // of one (sample, KV group).  Accumulators -> RoPE -> bf16 operand images in LDS (Q [5][128][64], K [128][64],
  // V^T [64][128], the layouts of attention.hip), then each of the 8 waves runs 16 query rows x 5 heads of
  // softmax(QK^T/8)V straight from LDS.  q, k, v never touch HBM and the attention launch disappears.
  if constexpr (EPI == EPI_QKV_ATTN) {
    static_assert(BM == 128 && BN == 448 && NW == 8, "one block = one sample x one KV group");
    constexpr int SQ = 0, SK = 5 * 16384, SV = SK + 16384;
    static_assert(SV + 16384 <= 2 * STAGE, "operand images do not fit the staging buffers");
    __builtin_amdgcn_s_barrier();
    JAT_TL(3)
    if (p.dbg & 64) { JAT_TL_FLUSH() return; }    // timing aid: K loop only
    // Column tiles of this wave (ctile): j < 5 the q tiles 5 wn + j, j = 5 its k tile, j = 6 its v tile (swapped operands).
    // RoPE angles: tile c covers the pair-interleaved features (16 c) % 64 ... of a head, i.e. frequencies 8 (c % 4) + 2 fg, +1;
    // c % 4 = (wn + j) % 4 for the q tiles and wn % 4 for the k tile: FOUR angle sets per row tile serve its six RoPE tiles
    // (set s: frequencies of (wn + s) % 4; tile j uses set j % 4, the k tile set 0).  v_sin / v_cos of fract(pos * inv_freq / 2 pi),
    // evaluated in registers (a table lookup is a 16-cache-line gather per instruction).
    float2 invf[4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) invf[sidx] = *(const float2*)(p.rope_inv_freq + ((wn + sidx) & 3) * 8 + fg * 2);
    float4 bqk[6];
    float bvs = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      bqk[j] = p.bias ? *(const float4*)(p.bias + n0 + ctile(wn, j) * 16 + fg * 4) : float4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bvs = p.bias[n0 + ctile(wn, 6) * 16 + frow];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * TM * 16 + i * 16 + frow;  // token position inside the sample (m0 = b * 128)
      float cs[4][4];                              // [set]: cos0, cos1, sin0, sin1
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const float r0 = __builtin_amdgcn_fractf((float)r * invf[sidx].x * 0.15915494309189535f);
        const float r1 = __builtin_amdgcn_fractf((float)r * invf[sidx].y * 0.15915494309189535f);
        cs[sidx][0] = __builtin_amdgcn_cosf(r0); cs[sidx][1] = __builtin_amdgcn_cosf(r1);
        cs[sidx][2] = __builtin_amdgcn_sinf(r0); cs[sidx][3] = __builtin_amdgcn_sinf(r1);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int nl = ctile(wn, j) * 16;          // tile-local first column of this MFMA tile (wave-uniform)
        f32x4 v = acc[i][j] * rstd_rows[i];
        if (p.bias) { v[0] += bqk[j].x; v[1] += bqk[j].y; v[2] += bqk[j].z; v[3] += bqk[j].w; }   // folded norm: shift @ W^T
        constexpr int NSET[6] = {0, 1, 2, 3, 0, 0};
        const float* a4 = cs[NSET[j]];
        const float2 r01 = rope_rot(v[0], v[1], a4[0], a4[2]), r23 = rope_rot(v[2], v[3], a4[1], a4[3]);
        const uint2 pk = pack4(r01.x, r01.y, r23.x, r23.y);
        const int colb = ((nl & 63) + fg * 4) * 2, chunk = colb >> 4, off = colb & 15;
        if (j < 5) *(uint2*)(smem + SQ + (nl >> 6) * 16384 + r * 128 + ((chunk ^ (r & 7)) << 4) + off) = pk;
        else *(uint2*)(smem + SK + r * 128 + ((chunk ^ ((r & 3) | (((r >> 3) & 1) << 2))) << 4) + off) = pk;
      }
      {  // v tile, D[token][feature]: this lane holds keys rk .. rk + 3 of feature d -> 8 contiguous bytes of V^T [64][128]
        const int rk = wm * TM * 16 + i * 16 + fg * 4, d = (ctile(wn, 6) - 24) * 16 + frow, kc = rk >> 3;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vv[e] = acc[i][6][e] * __shfl(rstd_rows[i], fg * 4 + e);   // rstd of token rk + e sits in the lane whose frow is 4 fg + e
          if (p.bias) vv[e] += bvs;
        }
        *(uint2*)(smem + SV + d * 256 + (kc >> 3) * 128 + (((kc & 7) ^ (d & 7)) << 4) + (rk & 7) * 2) = pack4(vv[0], vv[1], vv[2], vv[3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    JAT_TLX(0)
    __builtin_amdgcn_s_barrier();
    JAT_TLX(1)
    if (p.dbg & 32) { JAT_TL_FLUSH() return; }    // timing aid: no attention phase
    const int q = wave * 16 + frow;                // this lane's query row (B-operand column)
    constexpr int G = 5;
    const int hbase = (n0 / 448) * G;
    // The five query heads of the group are processed TOGETHER, phase by phase, instead of one head after the other:
    // 80 independent QK^T MFMAs (each K fragment read from LDS once and used by all five heads), then five softmaxes whose
    // exp2 / shuffles overlap, then 80 independent PV MFMAs (each V fragment read once).  One head at a time the phase was
    // a chain of dependent LDS-read -> MFMA -> VALU steps on two waves per SIMD: 13.9 us per launch for 2.4 us of MFMA
    // work (profiles/r02/fused_attention_phase.log).  Per-element arithmetic and accumulation order are unchanged.
    bf16x8 qf[G][2];
#pragma unroll
    for (int h = 0; h < G; ++h) {
      qf[h][0] = *(const bf16x8*)(smem + SQ + h * 16384 + q * 128 + (((0 + fg) ^ (q & 7)) << 4));
      qf[h][1] = *(const bf16x8*)(smem + SQ + h * 16384 + q * 128 + (((4 + fg) ^ (q & 7)) << 4));
    }
    f32x4 st[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) st[h][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const int r = 32 * (kt >> 1) + 8 * (frow >> 2) + (frow & 3) + 4 * (kt & 1);
      const int ks = (r & 3) | (((r >> 3) & 1) << 2);
      const bf16x8 kf0 = *(const bf16x8*)(smem + SK + r * 128 + (((0 + fg) ^ ks) << 4));
      const bf16x8 kf1 = *(const bf16x8*)(smem + SK + r * 128 + (((4 + fg) ^ ks) << 4));
#pragma unroll
      for (int h = 0; h < G; ++h) {
        st[h][kt] = JAT_MFMA_16x16x32(kf0, qf[h][0], st[h][kt], 0, 0, 0);
        st[h][kt] = JAT_MFMA_16x16x32(kf1, qf[h][1], st[h][kt], 0, 0, 0);
      }
    }
    JAT_TLX(2)
    float inv[G];
    bf16x8 pf[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h) {
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(st[h][kt][0], st[h][kt][1]), fmaxf(st[h][kt][2], st[h][kt][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float nb = -mx * p.attn_scale_log2e;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(st[h][kt][e], p.attn_scale_log2e, nb));
          st[h][kt][e] = pv;
          sum += pv;
        }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      inv[h] = 1.0f / sum;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        pf[h][kk] = jat_pack8(st[h][2 * kk], st[h][2 * kk + 1]);
      }
    }
    JAT_TLX(3)
    f32x4 o[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[h][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + frow, ch = kk * 4 + fg;
        const bf16x8 vf = *(const bf16x8*)(smem + SV + d * 256 + (ch >> 3) * 128 + (((ch & 7) ^ (d & 7)) << 4));
#pragma unroll
        for (int h = 0; h < G; ++h) o[h][dt] = JAT_MFMA_16x16x32(vf, pf[h][kk], o[h][dt], 0, 0, 0);
      }
    JAT_TLX(4)
    if (m0 + q < p.M) {
#pragma unroll
      for (int h = 0; h < G; ++h) {
        bf16_t* op = (bf16_t*)p.out + (int64_t)(m0 + q) * p.ldo + (hbase + h) * 64 + fg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *(uint2*)(op + dt * 16) = pack4(o[h][dt][0] * inv[h], o[h][dt][1] * inv[h], o[h][dt][2] * inv[h], o[h][dt][3] * inv[h]);
      }
    }
    JAT_TL_FLUSH()
    return;
  }

  // ---- direct epilogue: lane owns C[m][n..n+3], m = tile row (lane&15), n = 4*(lane>>4) + reg -----------
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * TM * 16 + i * 16 + frow;
    if (m >= p.M) continue;
    int b = 0, pos = m;
    if constexpr (EPI == EPI_RESID || EPI == EPI_QKV_ROPE || EPI == EPI_UNPATCH) {
      b = m / p.ntok;
      pos = m - b * p.ntok;
    }
    const float rstd_d = rstd_rows[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nt = nw0 + j * 16;          // wave-uniform first column of this 16-wide MFMA tile
      const int n = nt + fg * 4;
      f32x4 v = acc[i][j] * rstd_d;
      if ((p.dbg & 1) && v[0] != 12345.678f) continue;
      if constexpr (EPI == EPI_QKV_ROPE) {
        // Wq / Wk rows are packed pair-interleaved per head (position 2d <- feature d, 2d+1 <- feature d+32),
        // so the RoPE pair (d, d+32) (jat_audiosr_v3.py:87-108) sits in adjacent registers of one lane.
        // q.k is invariant under this shared permutation; V is not permuted.
        if (nt < p.D + p.kvD) {
          const int d0 = ((nt & 63) >> 1) + fg * 2;
          const float2 c = *(const float2*)(p.rope_cos + (int64_t)pos * 32 + d0);
          const float2 s = *(const float2*)(p.rope_sin + (int64_t)pos * 32 + d0);
          const float o0 = v[0] * c.x - v[1] * s.x, o1 = v[1] * c.x + v[0] * s.x;
          const float o2 = v[2] * c.y - v[3] * s.y, o3 = v[3] * c.y + v[2] * s.y;
          bf16_t* dst = (nt < p.D) ? ((bf16_t*)p.out + (int64_t)m * p.D + n)
                                   : (p.k_out + (int64_t)m * p.kvD + (n - p.D));
          *(uint2*)dst = pack4(o0, o1, o2, o3);
        } else {  // v head: store transposed vt[b][hv][d][pos]
          const int nv = n - p.D - p.kvD;
          bf16_t* dst = p.vt_out + ((int64_t)(b * (p.kvD >> 6) + (nv >> 6)) * 64 + (nv & 63)) * p.npad + pos;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.npad] = f2bf(v[r]);
        }
      } else {
        if (p.bias) {
          const float4 bb = *(const float4*)(p.bias + n);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if constexpr (EPI == EPI_F32) {
          *(float4*)((float*)p.out + (int64_t)m * p.ldo + n) = float4{v[0], v[1], v[2], v[3]};
        } else if constexpr (EPI == EPI_BF16) {
          *(uint2*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EPI == EPI_BF16_GELU) {
          *(uint2*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) = gelu_pack4(v[0], v[1], v[2], v[3]);
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
// Persistent two-tile form of the quadrant ping-pong GEMM (PIPE 8 above, same K loop) for a CONSUMER GEMM with more tiles than
// CUs: the MLP fc1 of the sampler, M = 7168, N = 5120 -> 512 tiles of 224 x 320 on 256 CUs.  As two rounds of one-tile blocks
// (variant 36) every block pays, per tile (profiles/r03/timeline_*): ~5 us from entry to the first MFMA (cold first K-tiles,
// 35 MB requested by 256 CUs at once), ~33 us of K loop, ~7 us of epilogue, and the second round starts 2-5 us after the first
// exits.  Here ONE block per CU walks two tiles, and while it runs the first tile's epilogue the first K-tile of the second
// tile (+ its bias slice and row statistics) is already streaming into LDS:
//     [X0, K-tiles 0/1 of tile 0] K loop | read rstd / bias from LDS | barrier | DMA: X1, K-tile 0 of tile 1 -> stage 0 |
//     epilogue of tile 0 (slabs in stage 1; its stores drain under the next K loop) | barrier | DMA: K-tile 1 -> stage 1 |
//     counted vmcnt | K loop of tile 1 | epilogue
// What makes the overlap work:
//   * nothing the epilogue waits for is a vector-memory LOAD (vmcnt counts loads and stores in issue order: one compiler-
//     inserted vmcnt(0) for a bias load would drain the prefetch first): the tile's bias slice and its rows of the producer's
//     row partial sums arrive by LDS-DMA with the first K-tile ("X": 2 + 14 KiB beside the two stages) and are read with
//     ds_read; the 28 registers of raw partials and the shuffles of the one-tile kernels are gone;
//   * the epilogue issues EXACTLY NST stores per wave (M % BM == 0, whole tiles only), so "tile 1's first K-tile has landed"
//     is the counted wait vmcnt(NST + CTILE) behind the DMA of its second K-tile: the stores stay in flight;
//   * the epilogue works on 16-row groups (two 2.8 KB slabs per wave, alternating) with the stores of group g-1 issued between
//     the column-tile pairs of group g's vector work (4 interleaved GELU chains per step).
// Grid = min(tiles, 256 * k) blocks with tiles <= 2 * grid; virtual tile v of a block b: b, b + grid.
template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(WM * WN * 64, 1) gemm_persist_kernel(const GemmArgs p) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int NW = WM * WN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(NW == 8 && TM * TN > 20, "the row-split quadrant ping-pong of the large wave tiles");
  static_assert(EPI == EPI_BF16 || EPI == EPI_BF16_GELU, "consumer epilogues only");
  constexpr int TMa = (TM + 1) / 2, TMb = TM - TMa, TNa = (TN + 1) / 2, TNb = TN - TNa;
  constexpr int PA0 = WM * TMa * 2, PA1 = WM * TMb * 2, PB0 = WN * TNa * 2, PB1 = WN * TNb * 2;
  constexpr int CA0 = (PA0 + 7) / 8, CA1 = (PA1 + 7) / 8, CB0 = (PB0 + 7) / 8, CB1 = (PB1 + 7) / 8;
  constexpr int CTILE = CA0 + CA1 + CB0 + CB1;
  // X: bias slice of the tile (BN floats, as whole 1-KiB pieces) + its rows of the row partial sums (BM rows x 16 floats)
  constexpr int XB_PIECES = (BN * 4 + 1023) / 1024, XP_PIECES = BM * 64 / 1024;
  constexpr int XB_OFF = 2 * STAGE, XP_OFF = XB_OFF + XB_PIECES * 1024;
  constexpr int CXP = (XP_PIECES + 7) / 8, CX = 1 + CXP;          // DMA instructions per wave: one bias piece + its partials pieces
  static_assert(BM * 64 % 1024 == 0, "whole pieces of row partials");
  // epilogue: 16-row groups, RS-byte slab rows, chunks of 16 B
  constexpr int RS = TN * 32 + 16, SLAB = 16 * RS, CPR = TN * 2, NCH = (16 * CPR + 63) / 64;
  constexpr int NST = TM * NCH;                                   // stores per wave per tile
  static_assert(NW * 2 * SLAB <= STAGE, "slabs live in stage 1");
  static_assert(NST + CTILE <= 63 && 2 * CTILE + CX <= 63, "vmcnt immediates");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, grp = wave >> 2;
  const int tiles_m = p.M / BM, tiles_n = p.N / BN, ntiles = tiles_m * tiles_n;
  const int nk = p.K / 64;
  auto coords = [&](int v, int& m0, int& n0) {   // XCD-contiguous chunks of the VIRTUAL tile range, then grouped-M order (as above)
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int first_m = (id / per_group) * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int in_g = id % per_group;
    m0 = (first_m + in_g % gsz) * BM;
    n0 = (in_g / gsz) * BN;
  };
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const int frow = lane & 15, fg = lane >> 4;
  const int a_row_off = (wm * TM * 16 + frow) * 128;
  const int b_row_off = (wn * TN * 16 + frow) * 128;
  const int coff0 = ((0 + fg) ^ (frow & 7)) * 16, coff1 = ((4 + fg) ^ (frow & 7)) * 16;
  // per-lane source offsets (whole tiles: no row clamp, the same for every tile) and wave-uniform LDS offsets of my DMA pieces
  unsigned oa0[CA0], oa1[CA1], ob0[CB0], ob1[CB1];
  int la0[CA0], la1[CA1], lb0[CB0], lb1[CB1];
  auto piece_row = [&](int q, int per, int tiles, int first) {
    const int w = q / per, in = q - w * per;
    return w * tiles * 16 + first * 16 + in * 8;
  };
#pragma unroll
  for (int j = 0; j < CA0; ++j) {
    const int r = piece_row(min(wave + 8 * j, PA0 - 1), TMa * 2, TM, 0);
    oa0[j] = (unsigned)((r + srow) * (int)p.lda * 2 + schunk * 16);
    la0[j] = r * 128;
  }
#pragma unroll
  for (int j = 0; j < CA1; ++j) {
    const int r = piece_row(min(wave + 8 * j, PA1 - 1), TMb * 2, TM, TMa);
    oa1[j] = (unsigned)((r + srow) * (int)p.lda * 2 + schunk * 16);
    la1[j] = r * 128;
  }
#pragma unroll
  for (int j = 0; j < CB0; ++j) {
    const int r = piece_row(min(wave + 8 * j, PB0 - 1), TNa * 2, TN, 0);
    ob0[j] = (unsigned)((r + srow) * (int)p.ldw * 2 + schunk * 16);
    lb0[j] = A_BYTES + r * 128;
  }
#pragma unroll
  for (int j = 0; j < CB1; ++j) {
    const int r = piece_row(min(wave + 8 * j, PB1 - 1), TNb * 2, TN, TNa);
    ob1[j] = (unsigned)((r + srow) * (int)p.ldw * 2 + schunk * 16);
    lb1[j] = A_BYTES + r * 128;
  }
  const char *a_base, *b_base;      // current tile's first A row / W row (wave-uniform)
  auto dma_a0 = [&](int st, int kt) {
#pragma unroll
    for (int j = 0; j < CA0; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(a_base + kt * 128 + oa0[j]), (lds_ptr_t)(smem + st * STAGE + la0[j]), 16, 0, 0);
  };
  auto dma_a1 = [&](int st, int kt) {
#pragma unroll
    for (int j = 0; j < CA1; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(a_base + kt * 128 + oa1[j]), (lds_ptr_t)(smem + st * STAGE + la1[j]), 16, 0, 0);
  };
  auto dma_b0 = [&](int st, int kt) {
#pragma unroll
    for (int j = 0; j < CB0; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(b_base + kt * 128 + ob0[j]), (lds_ptr_t)(smem + st * STAGE + lb0[j]), 16, 0, 0);
  };
  auto dma_b1 = [&](int st, int kt) {
#pragma unroll
    for (int j = 0; j < CB1; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(b_base + kt * 128 + ob1[j]), (lds_ptr_t)(smem + st * STAGE + lb1[j]), 16, 0, 0);
  };
  auto dma_tile = [&](int st, int kt) { dma_a0(st, kt); dma_b0(st, kt); dma_b1(st, kt); dma_a1(st, kt); };
  // X of the tile at (m0, n0): every wave issues CX pieces (duplicates rewrite the same bytes).  The bias slice is BN floats:
  // its last piece is shifted back so that it ends with the slice (pieces overlap instead of reading past it).
  auto dma_x = [&](int m0, int n0) {
    {
      const int q = wave % XB_PIECES;
      const int foff = min(q * 256, BN - 256);                       // first float of this piece
      const float* src = (p.bias ? p.bias + n0 : (const float*)p.W) + foff + lane * 4;   // no bias: any readable bytes (unused)
      __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr_t)(smem + XB_OFF + foff * 4), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < CXP; ++j) {
      const int q = min(wave + 8 * j, XP_PIECES - 1);
      const float* src = (p.rs_part ? p.rs_part + (int64_t)m0 * 16 : (const float*)p.A) + q * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr_t)(smem + XP_OFF + q * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
  bf16x8 fa[2][TMa], fb[2][TNa];
  auto rd_a = [&](int st, int i0, int cnt) {
    const char* sA = smem + st * STAGE + a_row_off + i0 * 2048;
#pragma unroll
    for (int i = 0; i < TMa; ++i)
      if (i < cnt) {
        fa[0][i] = *(const bf16x8*)(sA + i * 2048 + coff0);
        fa[1][i] = *(const bf16x8*)(sA + i * 2048 + coff1);
      }
  };
  auto rd_b = [&](int st, int j0, int cnt) {
    const char* sB = smem + st * STAGE + A_BYTES + b_row_off + j0 * 2048;
#pragma unroll
    for (int j = 0; j < TNa; ++j)
      if (j < cnt) {
        fb[0][j] = *(const bf16x8*)(sB + j * 2048 + coff0);
        fb[1][j] = *(const bf16x8*)(sB + j * 2048 + coff1);
      }
  };
#define JAT_Q(I0, IC, J0, JC)                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
  _Pragma("unroll") for (int i = 0; i < (IC); ++i)                                                              \
  _Pragma("unroll") for (int j = 0; j < (JC); ++j)                                                              \
    acc[(I0) + i][(J0) + j] = JAT_MFMA_16x16x32(fb[ks][j], fa[ks][i], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define JAT_LOAD_END()                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);                \
  __builtin_amdgcn_s_barrier();                     \
  __builtin_amdgcn_s_setprio(1);
#define JAT_MMA_END()                 \
  __builtin_amdgcn_s_setprio(0);      \
  __builtin_amdgcn_sched_barrier(0);  \
  __builtin_amdgcn_s_barrier();
  // the K loop of PIPE 8 (row-split form): on entry K-tile 0 has landed in stage 0 and K-tile 1 is in flight into stage 1
  auto kloop = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    auto ktile = [&](int t, int st) __attribute__((always_inline)) {
      const bool more2 = t + 2 < nk;
      const bool more1 = t >= 1 && t + 1 < nk;
      rd_b(st, 0, TNa);                       // P1 (A0, B0)
      __builtin_amdgcn_sched_barrier(0);
      rd_a(st, 0, TMa);
      if (more1) dma_b0(st ^ 1, t + 1);
      JAT_LOAD_END()
      JAT_Q(0, TMa, 0, TNa)
      JAT_MMA_END()
      rd_b(st, TNa, TNb);                     // P2 (A0, B1)
      if (more2) dma_a0(st, t + 2);
      JAT_LOAD_END()
      JAT_Q(0, TMa, TNa, TNb)
      JAT_MMA_END()
      rd_a(st, TMa, TMb);                     // P3 (A1, B1)
      if (more2) dma_b1(st, t + 2);
      JAT_LOAD_END()
      JAT_Q(TMa, TMb, TNa, TNb)
      JAT_MMA_END()
      rd_b(st, 0, TNa);                       // P4 (A1, B0)
      if (more2) {
        dma_a1(st, t + 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CA0 + CB1 + CA1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      JAT_LOAD_END()
      JAT_Q(TMa, TMb, 0, TNa)
      JAT_MMA_END()
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(t, 0);
      if (t + 1 < nk) ktile(t + 1, 1);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  };
#undef JAT_Q
#undef JAT_LOAD_END
#undef JAT_MMA_END

  // ---- epilogue of the tile at (m0, n0); `prefetch()` runs right after the barrier that frees the stages and X --------------
  const float invk = 1.0f / (float)p.K;
  const unsigned sl0 = (unsigned)(uintptr_t)(lds_ptr_t)(smem + STAGE + wave * (2 * SLAB));   // LDS byte addresses of my two slabs
  const unsigned sl1 = sl0 + SLAB;
  const unsigned wslab = frow * RS + fg * 8;                                                 // my 4 packed values of column tile 0
  int rowt[NCH], soff[NCH], goff[NCH];
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    const int c = lane + 64 * t;
    rowt[t] = c / CPR;
    const int cc = c - rowt[t] * CPR;
    soff[t] = min(rowt[t], 15) * RS + cc * 16;
    goff[t] = rowt[t] * (int)p.ldo * 2 + cc * 16;
  }
  auto epilogue = [&](int m0, int n0, auto&& prefetch) __attribute__((always_inline)) {
    float rstd[TM];
    float4 bb[TN];
    // 1/rms of my rows from the 16 partial sums per row staged in X, in the order of the one-tile kernels:
    // chunk sums (x + y) + (z + w), then (c0 + c1) + (c2 + c3)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rstd[i] = 1.0f;
      if (p.rs_part) {
        const char* rp = smem + XP_OFF + (wm * TM * 16 + i * 16 + frow) * 64;
        const float4 c0 = *(const float4*)rp, c1 = *(const float4*)(rp + 16), c2 = *(const float4*)(rp + 32), c3 = *(const float4*)(rp + 48);
        const float s0 = (c0.x + c0.y) + (c0.z + c0.w), s1 = (c1.x + c1.y) + (c1.z + c1.w);
        const float s2 = (c2.x + c2.y) + (c2.z + c2.w), s3 = (c3.x + c3.y) + (c3.z + c3.w);
        rstd[i] = rsqrtf(((s0 + s1) + (s2 + s3)) * invk + 1e-6f);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bb[j] = p.bias ? *(const float4*)(smem + XB_OFF + (wn * TN * 16 + j * 16 + fg * 4) * 4) : float4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();            // every wave is done with the stages and with X
    prefetch();
    char* const obase = (char*)p.out + ((int64_t)(m0 + wm * TM * 16) * p.ldo + n0 + wn * TN * 16) * 2;
    u32x4 raw[NCH];
    auto vwrite = [&](auto gc, auto kc) __attribute__((always_inline)) {   // column tiles 2k, 2k+1 of row tile g -> slab[g & 1]
      constexpr int g = decltype(gc)::value, k = decltype(kc)::value;
      constexpr int NJ = (2 * k + 1 < TN) ? 2 : 1;
      f32x2 h[2 * NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const f32x4 v = acc[g][2 * k + jj] * rstd[g];
        h[2 * jj] = f32x2{v[0] + bb[2 * k + jj].x, v[1] + bb[2 * k + jj].y};
        h[2 * jj + 1] = f32x2{v[2] + bb[2 * k + jj].z, v[3] + bb[2 * k + jj].w};
      }
      if constexpr (EPI == EPI_BF16_GELU) gelu_erf_n<2 * NJ>(h);
      // The slab accesses are inline asm: the compiler puts s_waitcnt vmcnt(0) in front of every LDS access it can see while
      // an LDS-DMA is in flight (it cannot tell the slabs from the DMA's destination), which would hold the whole epilogue
      // until the prefetched K-tile has landed.  Same-wave LDS accesses execute in order; the reads are waited for below.
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const uint2 pk = pack4(h[2 * jj][0], h[2 * jj][1], h[2 * jj + 1][0], h[2 * jj + 1][1]);
        const unsigned long long pk64 = (unsigned long long)pk.x | ((unsigned long long)pk.y << 32);
        const unsigned waddr = (g & 1 ? sl1 : sl0) + wslab + ((2 * k + jj) * 32);   // (asm operands must be locals of this lambda)
        asm volatile("ds_write_b64 %0, %1" ::"v"(waddr), "v"(pk64) : "memory");
      }
    };
    auto sread = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      static_for<0, NCH>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const unsigned raddr = (g & 1 ? sl1 : sl0) + soff[t];
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(raddr) : "memory");
        raw[t] = r;
      });
    };
    auto swait = [&]() __attribute__((always_inline)) {   // the reads above have returned; no use of raw[] is scheduled above this
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      static_for<0, NCH>([&](auto tc) __attribute__((always_inline)) {
        u32x4 r = raw[decltype(tc)::value];
        asm volatile("" : "+v"(r));
        raw[decltype(tc)::value] = r;
      });
    };
    auto gstore = [&](auto gc, auto tc) __attribute__((always_inline)) {   // issued by EVERY wave for every (g, t): NST is exact
      constexpr int g = decltype(gc)::value, t = decltype(tc)::value;
      if (rowt[t] < 16) *(u32x4*)(obase + (int64_t)g * 16 * p.ldo * 2 + goff[t]) = raw[t];
    };
    constexpr int NK2 = (TN + 1) / 2;                                    // vector-work steps per group
    static_assert(NK2 >= NCH, "one store slot per vector-work step");
    static_for<0, NK2>([&](auto kc) __attribute__((always_inline)) { vwrite(std::integral_constant<int, 0>{}, kc); });
    static_for<1, TM>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      sread(std::integral_constant<int, g - 1>{});   // same-wave LDS accesses execute in order
      static_for<0, NK2>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        __builtin_amdgcn_sched_barrier(0);
        vwrite(gc, kc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k == 0) swait();
        if constexpr (k < NCH) gstore(std::integral_constant<int, g - 1>{}, kc);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    sread(std::integral_constant<int, TM - 1>{});
    swait();
    static_for<0, NCH>([&](auto tc) __attribute__((always_inline)) { gstore(std::integral_constant<int, TM - 1>{}, tc); });
  };

  // ---- tile 0 ---------------------------------------------------------------------------------------------------------------
  const int v0 = blockIdx.x, v1 = blockIdx.x + gridDim.x;
  const bool two = v1 < ntiles;
  int m0, n0, m1 = 0, n1 = 0;
  coords(v0, m0, n0);
  if (two) coords(v1, m1, n1);
  a_base = (const char*)(p.A + (int64_t)m0 * p.lda);
  b_base = (const char*)(p.W + (int64_t)n0 * p.ldw);
  dma_x(m0, n0);
  dma_tile(0, 0);
  if (nk > 1) {
    dma_tile(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CTILE) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  kloop();
  epilogue(m0, n0, [&]() __attribute__((always_inline)) {
    if (two) {   // the second tile's X and first K-tile stream in under this epilogue (stage 0 and X are free; the slabs are in stage 1)
      a_base = (const char*)(p.A + (int64_t)m1 * p.lda);
      b_base = (const char*)(p.W + (int64_t)n1 * p.ldw);
      dma_x(m1, n1);
      dma_tile(0, 0);
    }
  });
  if (!two) return;
  // ---- tile 1: its first K-tile (+ X) was issued BEFORE this wave's NST epilogue stores, its second K-tile goes out now --------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();              // every wave has read its slabs (stage 1) for the last time
  if (nk > 1) {
    dma_tile(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + CTILE) : "memory");   // all but my stores and K-tile 1: K-tile 0 and X have landed
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
  }
  kloop();
  epilogue(m1, n1, [&]() __attribute__((always_inline)) {});
}

// -----------------------------------------------------------------------------------------------------
// k-step-pair GEMM for the gated-residual PRODUCER GEMMs of the sampler (out_proj, MLP fc2: N = 1280): a 224 x 160 block tile so
// that M = 7168 gives 32 x 8 = 256 tiles = every CU (the 256 x 160 tile of variants 25 / 32 fills 224 of 256 CUs, and no 8-wave
// tiling of 224 x 160 exists: 140 MFMA tiles do not divide by 8).  Here the 8 waves are TWO groups of 2 x 2 waves that compute
// the SAME 224 x 160 tile (wave tile 112 x 80 = 7 x 5 MFMA tiles, as the fc1 kernels) over different halves of K: group g takes
// k-step g (32 columns) of every 64-column K-tile.  The two waves of a SIMD (w, w + 4) run one barrier apart as in PIPE 8 —
//     interval 2k   : group 0 reads its fragments of K-tile k      | group 1: 35 MFMAs on K-tile k-1
//     interval 2k+1 : group 0: 35 MFMAs on K-tile k                | group 1 reads its fragments of K-tile k
// — a wave holds ONE k-step of fragments (48 VGPRs, no quadrant cut), 2 phases and 4 barriers per K-tile instead of 4 and 8,
// 12 instead of 18 fragment reads per 35 / 40 MFMAs.  THREE LDS stages (K-tile k in stage k % 3, 48 KiB each): the DMA of K-tile
// k+3 is issued in the load phase of K-tile k+1 (by then both groups have read K-tile k) and has two K-tiles to land; each
// wave waits for its own pieces of K-tile k+1 with a counted vmcnt at the end of its load phase of K-tile k, one barrier
// before anyone reads them.  After the K loop the groups exchange halves of their partial accumulators through LDS (group 0
// keeps row tiles 0-3 of its wave tile, group 1 row tiles 4-6; fp32 a + b in either order is the same number), and all 8 waves
// run the split-residual epilogue of the one-tile kernels on their half.
template <int EPI, int LONGK, bool PART = false>   // LONGK: a name tag only (K >= 4096: the MLP fc2; else out_proj / patch embed) so that
                                                   // the two call sites of the sampler show as separate rows of a kernel trace
// PART: a split-K slice (blockIdx.y) of an un-folded GEMM — the tile's fp32 sums go to slice z of the partial workspace, no bias, no
// planes; the finishing pass (elementwise.hip: splitk_resid_*) sums the slices in fixed order.  This is how a half-size batch
// (M = 3584: 128 tiles) still puts one fat tile on each of the 256 CUs.
__global__ void __launch_bounds__(512, 1) gemm_kpair_kernel(const GemmArgs p) {
  constexpr int TM = 7, TN = 5, BM = 224, BN = 160, NW = 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = (BM + BN) / 8, CP = PIECES / NW;            // 48 one-KiB pieces per K-tile, 6 per wave
  static_assert(PIECES % NW == 0, "every wave issues the same number of DMA pieces");
  static_assert(EPI == EPI_RESID || EPI == EPI_F32, "split-residual producer epilogues");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef JAT_TIMELINE
  unsigned long long tl_[7] = {0, 0, 0, 0, 0, 0, 0}, tlx_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  JAT_TL(0) JAT_TLR(5)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
  const int frow = lane & 15, fg = lane >> 4;
  const int tiles_m = p.M / BM, tiles_n = p.N / BN;
  int m0, n0;
  {  // XCD-contiguous chunks, grouped-M order (as gemm_bf16_kernel)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int first_m = (id / per_group) * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int in_g = id % per_group;
    m0 = (first_m + in_g % gsz) * BM;
    n0 = (in_g / gsz) * BN;
  }
  const int nk = p.K / 64;
  const int64_t kz = PART ? (int64_t)blockIdx.y * p.K : 0;      // my slice's first column of A and W (p.K = the slice depth)
  // my DMA pieces: piece q = wave + 8 j covers image rows 8 q .. 8 q + 7 (A rows first, then W rows); whole tiles: no clamp
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const char* psrc[CP];
  int pdst[CP];
#pragma unroll
  for (int j = 0; j < CP; ++j) {
    const int q = wave + NW * j, r = q * 8;
    pdst[j] = r * 128;
    psrc[j] = r < BM ? (const char*)(p.A + (int64_t)(m0 + r + srow) * p.lda + kz) + schunk * 16
                     : (const char*)(p.W + (int64_t)(n0 + r - BM + srow) * p.ldw + kz) + schunk * 16;
  }
  auto dma_tile = [&](int kt) {
    char* st = smem + (kt % 3) * STAGE;
#pragma unroll
    for (int j = 0; j < CP; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(psrc[j] + kt * 128), (lds_ptr_t)(st + pdst[j]), 16, 0, 0);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[TM], fb[TN];
  const int a_off = (wm * TM * 16 + frow) * 128 + ((4 * grp + fg) ^ (frow & 7)) * 16;               // my k-step's chunk, swizzled
  const int b_off = A_BYTES + (wn * TN * 16 + frow) * 128 + ((4 * grp + fg) ^ (frow & 7)) * 16;
  auto rd = [&](int kt) {
    const char* st = smem + (kt % 3) * STAGE;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8*)(st + b_off + j * 2048);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8*)(st + a_off + i * 2048);
  };
  // prologue: K-tiles 0, 1, 2 in flight; tile 0 landed
  dma_tile(0);
  if (nk > 1) dma_tile(1);
  if (nk > 2) dma_tile(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CP) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  JAT_TL(1)
  // (All CP pieces go out in the load phase.  Spreading half of them between the MFMAs of the compute phase was measured SLOWER,
  // 1420 -> 1570 cycles per K-tile: the tile needs 44 B/clk of the ~46 B/clk the per-CU L2->LDS path delivers, and a DMA that
  // stalls at issue inside the compute phase stalls the MFMAs behind it.)
  for (int kt = 0; kt < nk; ++kt) {
    // load phase of K-tile kt (my pieces of K-tile kt+1 were waited for at the end of the previous load phase)
    rd(kt);
    if (kt >= 1 && kt + 2 < nk) dma_tile(kt + 2);          // stage of K-tile kt-1: both groups finished reading it an interval ago
    if (kt + 1 < nk) {
      if (kt >= 1 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CP) : "memory");   // all but the pieces just issued
      else if (kt == 0 && nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CP) : "memory");   // prologue: tile 2 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = JAT_MFMA_16x16x32(fb[j], fa[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  JAT_TL(2)
  if (grp == 0) __builtin_amdgcn_s_barrier();
  // ---- exchange: every wave is past its last fragment read (the barrier above) -------------------------------------------
  // slot (wq, direction): group 1 -> group 0 row tiles 0..3 (20 accumulator tiles), group 0 -> group 1 row tiles 4..6 (15)
  constexpr int XA = 20 * 1024, XB = 15 * 1024;
  char* xa = smem + wq * (XA + XB);
  char* xb = xa + XA;
  static_assert(4 * (XA + XB) <= 3 * STAGE, "exchange buffers fit the stages");
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) *(f32x4*)(xa + ((i * TN + j) * 64 + lane) * 16) = acc[i][j];
  } else {
#pragma unroll
    for (int i = 4; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) *(f32x4*)(xb + (((i - 4) * TN + j) * 64 + lane) * 16) = acc[i][j];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += *(const f32x4*)(xa + ((i * TN + j) * 64 + lane) * 16);
  } else {
#pragma unroll
    for (int i = 4; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += *(const f32x4*)(xb + (((i - 4) * TN + j) * 64 + lane) * 16);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the exchange buffers are consumed: the slabs below may overwrite them
  JAT_TL(3)
  if (p.dbg & 1) { JAT_TL_FLUSH() return; }
  // ---- split-residual epilogue (see gemm_bf16_kernel) on my half: row tiles [I0, I0 + NT) of the wave tile ---------------------
  constexpr int RS = TN * 64 + 16, CPR8 = TN * 2, NCH8 = TN;
  static_assert(NW * 32 * RS <= 3 * STAGE, "epilogue slabs fit the stages");
  char* wbuf = smem + wave * (32 * RS);
  const int nw0 = n0 + wn * TN * 16;
  float4 bb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bb[j] = (!PART && p.bias) ? *(const float4*)(p.bias + nw0 + j * 16 + fg * 4) : float4{0.f, 0.f, 0.f, 0.f};
  auto up = [](unsigned u, float& a, float& b) { a = jat_lo2f(u); b = jat_hi2f(u); };
  [[maybe_unused]] float* const pout = PART ? (float*)p.out + (int64_t)blockIdx.y * p.split_stride : nullptr;
  auto half_epilogue = [&](auto i0c, auto ntc) __attribute__((always_inline)) {
    constexpr int I0 = decltype(i0c)::value, NT = decltype(ntc)::value, NG = (NT + 1) / 2;
    const int mw0 = m0 + wm * TM * 16 + I0 * 16;
    // ONE register set for the residual planes and the gate: chunk t of group ig+1 is requested right after chunk t of group ig has
    // been consumed and BEFORE that chunk's stores are issued — a load issued behind a store would wait for the store to be
    // acknowledged (vmcnt counts loads and stores in issue order), which used to expose the store latency once per group
    [[maybe_unused]] uint4 hi[NCH8], lo[NCH8];
    [[maybe_unused]] float4 g0[NCH8], g1[NCH8];
    auto load_chunk = [&](int ig, int t) __attribute__((always_inline)) {
      const int grows = (2 * ig + 1 < NT) ? 32 : 16;
      const int c = lane + 64 * t, row = min(c / CPR8, grows - 1), cc = c - (c / CPR8) * CPR8;
      const int m = mw0 + ig * 32 + row, n = nw0 + cc * 8;
      const unsigned off = (unsigned)(m * (int)p.ldo + n) * 2u;   // uniform base + 32-bit lane offset (planes < 4 GB)
      hi[t] = *(const uint4*)((const char*)p.fold_out + off);
      lo[t] = *(const uint4*)((const char*)p.fold_lo + off);
      const float* gp = p.gate + (int64_t)(m / p.ntok) * p.gate_bstride + n;
      g0[t] = *(const float4*)gp;
      g1[t] = *(const float4*)(gp + 4);
    };
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int t = 0; t < NCH8; ++t) load_chunk(0, t);
    }
#pragma unroll
    for (int ig = 0; ig < NG; ++ig) {
      const int grows = (2 * ig + 1 < NT) ? 32 : 16;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j) if (2 * ig + ii < NT) {
          const f32x4 v = acc[I0 + (2 * ig + ii < NT ? 2 * ig + ii : 0)][j];
          *(float4*)(wbuf + (ii * 16 + frow) * RS + (j * 16 + fg * 4) * 4) =
              float4{v[0] + bb[j].x, v[1] + bb[j].y, v[2] + bb[j].z, v[3] + bb[j].w};
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NCH8; ++t) {
        const int c = lane + 64 * t, row = c / CPR8, cc = c - row * CPR8;
        const int m = mw0 + ig * 32 + row, n = nw0 + cc * 8;
        char* slot = wbuf + (row < 32 ? row : 0) * RS + cc * 32;
        const float4 a0 = *(const float4*)slot, a1 = *(const float4*)(slot + 16);
        float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if constexpr (PART) {      // the slice's sums as they are: 32 B per lane, 320 B per row and wave
          if (row < grows) {
            float* o = pout + (int64_t)m * p.ldo + n;
            *(float4*)o = a0;
            *(float4*)(o + 4) = a1;
          }
          continue;
        }
        if constexpr (EPI == EPI_RESID) {
          const unsigned hw[4] = {hi[t].x, hi[t].y, hi[t].z, hi[t].w}, lw[4] = {lo[t].x, lo[t].y, lo[t].z, lo[t].w};
          const float gg[8] = {g0[t].x, g0[t].y, g0[t].z, g0[t].w, g1[t].x, g1[t].y, g1[t].z, g1[t].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float h0, h1, l0, l1;
            up(hw[e], h0, h1);
            up(lw[e], l0, l1);
            x[2 * e] = __builtin_fmaf(gg[2 * e], x[2 * e], h0 + l0);
            x[2 * e + 1] = __builtin_fmaf(gg[2 * e + 1], x[2 * e + 1], h1 + l1);
          }
          if (ig + 1 < NG) load_chunk(ig + 1, t);        // this chunk's registers are free: the next group's chunk, before my stores
        }
        unsigned ho[4], lw2[4];
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned short ha = f2bf(x[2 * e]), hb = f2bf(x[2 * e + 1]);
          ho[e] = (unsigned)ha | ((unsigned)hb << 16);
          const float ra = x[2 * e] - jat_op2f(ha), rb = x[2 * e + 1] - jat_op2f(hb);
          lw2[e] = jat_pack2(ra, rb);
          sq += x[2 * e] * x[2 * e] + x[2 * e + 1] * x[2 * e + 1];
        }
        const bool live = row < grows;
        if (live && !(p.dbg & 128)) {
          const unsigned off = (unsigned)(m * (int)p.ldo + n) * 2u;
          *(uint4*)((char*)p.fold_out + off) = uint4{ho[0], ho[1], ho[2], ho[3]};
          *(uint4*)((char*)p.fold_lo + off) = uint4{lw2[0], lw2[1], lw2[2], lw2[3]};
        }
        if (row < 32) *(float*)slot = live ? sq : 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (!PART) {  // row partial sums over this wave's columns: two lanes per row, fixed order
        constexpr int HALF = CPR8 / 2;
        const int r = lane >> 1, h = lane & 1;
        float sq = 0.f;
#pragma unroll
        for (int cc = 0; cc < HALF; ++cc) sq += *(const float*)(wbuf + r * RS + (h * HALF + cc) * 32);
        sq += __shfl_xor(sq, 1);
        const int m = mw0 + ig * 32 + r;
        if (h == 0 && r < grows) p.fold_part[(int64_t)m * p.fold_np + nw0 / (TN * 16)] = sq;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  if (grp == 0) half_epilogue(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
  else half_epilogue(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
  JAT_TL_FLUSH()
}

static bool gemm_kpair_eligible(const GemmArgs& a, int epi) {
  return (epi == EPI_RESID || epi == EPI_F32) && a.fold_out && a.fold_lo && a.fold_part && a.M > 0 && a.M % 224 == 0 &&
         a.N % 160 == 0 && a.K % 64 == 0 && a.ksplit <= 1 && !a.rs_part &&
         (int64_t)a.M * a.ldo * 2 < (1ll << 32);   // the epilogue addresses the planes with 32-bit byte offsets
}
// split-K slices of an un-folded GEMM on the same tile (a.K = the slice depth, jat_gemm)
static bool gemm_kpair_part_eligible(const GemmArgs& a, int epi) {
  return epi == EPI_F32 && a.ksplit > 1 && !a.fold_out && !a.rs_part && !a.dual_rows && a.out && a.M > 0 && a.M % 224 == 0 &&
         a.N % 160 == 0 && a.K % 64 == 0 && a.K >= 192;
}
template <int EPI, int LONGK, bool PART = false>
static hipError_t launch_kpair(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = 3 * (224 + 160) * 128;
  static_assert(LDS <= 160 * 1024, "three stages must fit the 160 KiB LDS");
  static bool attr_set = false;
  auto kern = gemm_kpair_kernel<EPI, LONGK, PART>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M / 224) * (a.N / 160), PART ? a.ksplit : 1), dim3(512), LDS, s, a);
  return hipGetLastError();
}

static bool gemm_persist_eligible(const GemmArgs& a, int epi) {
  return (epi == EPI_BF16 || epi == EPI_BF16_GELU) && a.M > 0 && a.M % 224 == 0 && a.N % 320 == 0 && a.K % 64 == 0 && a.ksplit <= 1 &&
         a.dual_rows == 0 && !a.fold_out && (!a.rs_part || a.rs_np == 16) && a.ldo * 2 * 16 < (1ll << 31) && !(a.dbg & 129);
}
template <int EPI>
static hipError_t launch_persist(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 224, BN = 320, STAGE = (BM + BN) * 128;
  constexpr int LDS = 2 * STAGE + 2048 + BM * 64;
  static_assert(LDS <= 160 * 1024, "stages + bias slice + row partials must fit the 160 KiB LDS");
  static bool attr_set = false;
  auto kern = gemm_persist_kernel<2, 4, 7, 5, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles = (a.M / BM) * (a.N / BN);
  int grid = tiles <= 256 ? tiles : (tiles + 1) / 2;            // two tiles per block once the tiles outnumber the CUs
  if (grid < 256 && tiles > 256) grid = 256;
  // experiment (tools/two_stream_ab.py): two tiles per block already from `half` blocks on, so that two half-batch launches on two
  // streams occupy 128 CUs each
  static const int half_env = getenv("JAT_PERSIST_HALF") ? atoi(getenv("JAT_PERSIST_HALF")) : 0;
  if (half_env > 0 && tiles > half_env && tiles <= 2 * half_env) grid = (tiles + 1) / 2;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
  return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int PIPE, int CE, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int LDS = (PIPE == 6 ? 3 : 2) * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  static bool attr_set = false;
  auto kern = gemm_bf16_kernel<WM, WN, TM, TN, PIPE, CE, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (a.N % BN != 0 || a.K % 64 != 0 || a.M <= 0) return hipErrorInvalidValue;
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  if (a.ksplit > 1 && EPI != EPI_F32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(tiles, a.ksplit > 1 ? a.ksplit : 1), dim3((WM * WN + (PIPE == 6 ? 4 : 0)) * 64), LDS, s, a);
  return hipGetLastError();
}

template <int WM, int WN, int TM, int TN, int PIPE, int CE = 0>
static hipError_t launch_epi(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_F32: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_F32>(a, s);
    case EPI_BF16: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_BF16>(a, s);
    case EPI_BF16_GELU: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_BF16_GELU>(a, s);
    case EPI_RESID: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_RESID>(a, s);
    case EPI_QKV_ROPE: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_QKV_ROPE>(a, s);
    case EPI_UNPATCH: return launch_one<WM, WN, TM, TN, PIPE, CE, EPI_UNPATCH>(a, s);
  }
  return hipErrorInvalidValue;
}

// variant table: {BM, BN} per id (for the host-side chooser) and the dispatch below must stay in sync
// Variant ids are stable across rounds (profiles and DESIGN.md cite them); retired ids have a {0, 0} tile and are rejected.
static const int kVariantTile[][2] = {
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},   // 0-9: retired (PIPE 0 / 1)
    {128, 64},                                                                          // 10: PIPE 1, the always-valid fallback (N % 64)
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},                             // 11-17: retired (PIPE 2 without the coalesced epilogue, PIPE 3)
    {128, 160}, {0, 0}, {128, 128}, {256, 256},      // 18, 20, 21: PIPE 2 + coalesced epilogue (19 retired)
    {0, 0}, {0, 0}, {0, 0},                          // 22-24: retired (PIPE 4 / 5)
    {256, 160}, {256, 128},                          // 25-26: PIPE 6 (8 MFMA waves + 4 DMA waves)
    {64, 160}, {64, 128},                            // 27-28: small-M tiles (PIPE 2 + coalesced epilogue)
    {0, 0}, {0, 0},                                  // 29-30: retired (PIPE 7)
    {224, 320}, {256, 160}, {256, 256}, {128, 448},  // 31-34: PIPE 8 (quadrant ping-pong, 2 LDS stages) + coalesced epilogue
    {224, 256},                                      // 35: PIPE 8
    {224, 320},                                      // 36: the tile of 31 with the software-pipelined bf16 / GELU epilogue (CE == 2)
    {0, 0},                                          // 37: retired (pipelined split-residual epilogue: slower, profiles/r03)
    {224, 320},                                      // 38: persistent two-tile form of 36 (gemm_persist_kernel); falls back to 36
    {256, 160},                                      // 39: k-step-pair 224 x 160 tile for the split-residual producers (gemm_kpair_kernel);
                                                     //     falls back to 32 (whose tile this entry reports)
};
static const int kVariantWaveN[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32, 0, 0, 0, 0, 0, 0, 0,
                                    80, 0, 64, 64, 0, 0, 0, 80, 64, 80, 64, 0, 0, 80, 80, 64, 112, 64, 80, 0, 80, 80};
int gemm_variant_wave_n(int variant) { return kVariantWaveN[variant]; }
bool gemm_variant_coalesced(int variant) { return variant >= 18; }
int gemm_num_variants() { return (int)(sizeof(kVariantTile) / sizeof(kVariantTile[0])); }

hipError_t launch_qkv_attn(const GemmArgs& a, hipStream_t s) {
  if (a.ntok != 128 || a.N % 448 != 0 || a.M % 128 != 0) return hipErrorInvalidValue;
  return launch_one<2, 4, 4, 7, 8, 0, EPI_QKV_ATTN>(a, s);
}
void gemm_variant_tile(int variant, int* bm, int* bn) {
  *bm = kVariantTile[variant][0];
  *bn = kVariantTile[variant][1];
}

bool gemm_variant_exists(int variant) {
  return variant >= 0 && variant < gemm_num_variants() && kVariantTile[variant][0] != 0;
}

hipError_t launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
  if (!gemm_variant_exists(variant)) return hipErrorInvalidValue;
  if (a.N % kVariantTile[variant][1] != 0) variant = (a.N % 128 == 0) ? 20 : 10;  // always-valid fallbacks
  switch (variant) {
    case 10: return launch_epi<2, 2, 4, 2, 1>(a, epi, s);
    case 18: return launch_epi<2, 2, 4, 5, 2, 1>(a, epi, s);
    case 20: return launch_epi<2, 2, 4, 4, 2, 1>(a, epi, s);
    case 21: return launch_epi<2, 4, 8, 4, 2, 1>(a, epi, s);
    case 25: return launch_epi<4, 2, 4, 5, 6, 1>(a, epi, s);
    case 26: return launch_epi<4, 2, 4, 4, 6, 1>(a, epi, s);
    case 27: return launch_epi<2, 2, 2, 5, 2, 1>(a, epi, s);
    case 28: return launch_epi<2, 2, 2, 4, 2, 1>(a, epi, s);
    case 31: return launch_epi<2, 4, 7, 5, 8, 1>(a, epi, s);
    case 39:
      if (gemm_kpair_part_eligible(a, epi)) return launch_kpair<EPI_F32, 0, true>(a, s);
      if (gemm_kpair_eligible(a, epi))
        return epi != EPI_RESID ? launch_kpair<EPI_F32, 0>(a, s) : a.K >= 4096 ? launch_kpair<EPI_RESID, 1>(a, s) : launch_kpair<EPI_RESID, 0>(a, s);
      [[fallthrough]];
    case 32: return launch_epi<4, 2, 4, 5, 8, 1>(a, epi, s);
    case 33: return launch_epi<2, 4, 8, 4, 8, 1>(a, epi, s);
    case 34: return launch_epi<2, 4, 4, 7, 8, 1>(a, epi, s);
    case 35: return launch_epi<2, 4, 7, 4, 8, 1>(a, epi, s);
    case 38:
      if (gemm_persist_eligible(a, epi)) return epi == EPI_BF16 ? launch_persist<EPI_BF16>(a, s) : launch_persist<EPI_BF16_GELU>(a, s);
      [[fallthrough]];
    case 36: return launch_epi<2, 4, 7, 5, 8, 2>(a, epi, s);
  }
  return hipErrorInvalidValue;
}
