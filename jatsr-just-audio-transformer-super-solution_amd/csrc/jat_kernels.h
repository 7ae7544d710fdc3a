// Internal launch interface between the C-ABI host code (jat_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jat_rng.h"

typedef uint16_t bf16_t;  // raw bf16 bits

// ---- GEMM: C[M,N] = A[M,K] * W[N,K]^T with a fused epilogue ------------------------------------
enum GemmEpi : int {
  EPI_F32 = 0,       // out fp32 [M,ldo] = acc + bias
  EPI_BF16 = 1,      // out bf16 [M,ldo] = acc + bias
  EPI_BF16_GELU = 2, // out bf16 = gelu_erf(acc + bias)        (jat_audiosr_v3.py:221-225, 266-268)
  EPI_RESID = 3,     // out fp32 [M,ldo] += gate[b,n]*(acc+bias) (jat_audiosr_v3.py:300,306)
  EPI_QKV_ROPE = 4,  // RoPE on q,k heads; q->[M,D], k->[M,kvD], v->vt[B,Hkv,64,Npad] (:154-160)
  EPI_QKV_ATTN = 6,  // fused: QKV projection + RoPE + GQA attention of one (sample, KV group) per block; out = attn_out
  EPI_UNPATCH = 5,   // out fp32 [B,C,T_orig]: feature c*4+p of token n -> [b,c,4n+p] (:406-420,465-469)
};

struct GemmArgs {
  const bf16_t* A;  // [M, lda] bf16, K contiguous
  const bf16_t* W;  // [N, ldw] bf16, K contiguous (nn.Linear weight layout)
  int64_t lda, ldw;
  int M, N, K;
  // epilogue
  void* out;
  int64_t ldo;
  const float* bias;     // [N] or nullptr
  const float* gate;     // EPI_RESID: gate + b*gate_bstride + n
  int64_t gate_bstride;
  int ntok;              // rows per batch sample (b = m / ntok, pos = m % ntok)
  // EPI_QKV_ROPE
  bf16_t* k_out;
  bf16_t* vt_out;
  int D, kvD, npad;
  const float* rope_cos; // [max_pos, 32]
  const float* rope_sin;
  const float* rope_inv_freq;  // [32] fp32: 1/10000^(2i/64), for in-register sin/cos (coalesced epilogue)
  // EPI_UNPATCH
  int C_out, T_orig;
  int dbg;  // profiling aid (bit0: suppress epilogue stores); 0 in production
  // ---- norm folding (sampler path; coalesced-epilogue variants only) ------------------------------------------
  // RMSNorm commutes with the matmul, and in the sampler every row shares the modulation (one t per step), so
  //   (x * rstd * w * (1 + scale) + shift) @ W^T  =  rstd[m] * (bf16(x) @ W'^T) + (shift @ W^T),   W' = W diag(w (1 + scale))
  // with W' and shift @ W^T precomputed per (step, layer) at sampler creation (jat_api.cpp FoldTable).
  // producer (EPI_RESID / EPI_F32): the residual stream is kept as two bf16 planes x = hi + lo (fold_out / fold_lo, ldo
  //   elements per row; `out` is not touched): read-modify-write them and emit the row partial sums of x_new^2 of this
  //   wave's column tile into fold_part[m][nw0 / wave_tile_n]  (plain stores, fixed order);
  // consumer (any epilogue): scale the accumulator row m by rsqrt(sum_j rs_part[m][j] / K + 1e-6) before the bias.
  bf16_t* fold_out;   // hi plane of the split residual stream (bf16(x)): also the next GEMM's A operand
  bf16_t* fold_lo;    // lo plane: bf16(x - hi)
  float* fold_part;
  int fold_np;
  const float* rs_part;
  int rs_np;
  float attn_scale_log2e;  // EPI_QKV_ATTN: (1/sqrt(64)) * log2(e)
  // EPI_BF16_GELU dual output (sampler patch embed): rows m get gelu(acc + bias + dual_add[m][n]), rows m + dual_rows
  // get gelu(acc + bias): the CFG cond / uncond halves share the z contribution of the first patch-embed Linear.
  const float* dual_add;
  int dual_rows;
  unsigned long long* dbg_out;  // profiling aid (tools/gemm_timeline.py): per-wave cycle sums of the PIPE 6 slot phases
  // split-K (EPI_F32 only; the dW GEMMs of the training step whose M x N is too small to fill 256 CUs): grid.y = ksplit
  // blocks each contract K columns starting at blockIdx.y * K and write their partial to out + blockIdx.y * split_stride
  // (fp32 elements); the caller sums the partials in fixed order (launch_sum_partials).  0 / 1: off.
  int ksplit;
  int64_t split_stride;
  int variant_hint;   // host side only (jat_gemm): > 0 = the tile variant the caller's split plan was made for
};

// variant: index into the tile/pipeline table of gemm.hip (gemm_variant_tile gives its BM x BN)
hipError_t launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s);
int gemm_num_variants();
bool gemm_variant_exists(int variant);   // ids of retired variants are rejected by launch_gemm
// Fused QKV projection + RoPE + attention for ntok == 128 (W = group-major fused weight [Hkv][5*64+64+64][K]):
// one block per (sample, KV group); a.out = attention output bf16 [M, D]; a.N = Hkv * 448.
hipError_t launch_qkv_attn(const GemmArgs& a, hipStream_t s);
void gemm_variant_tile(int variant, int* bm, int* bn);
int gemm_variant_wave_n(int variant);   // columns per wave tile (fold_part slot width)
bool gemm_variant_coalesced(int variant);

// ---- attention -----------------------------------------------------------------------------------
struct AttnArgs {
  const bf16_t* q;   // [B*N, ldq]   head h at column h*64
  const bf16_t* k;   // [B*N, ldk]   kv head g at column g*64
  const bf16_t* vt;  // [B, Hkv, 64, npad]  (V transposed, zero padded for key >= N)
  bf16_t* o;         // [B*N, ldo]
  int64_t ldq, ldk, ldo;
  int B, N, Hq, Hkv, npad;
  float scale_log2e; // (1/sqrt(64)) * log2(e)
  float* lse;        // optional [B, Hq, N] fp32: log2-domain log-sum-exp per query row (training forward), else nullptr
  DropSpec drop;     // training: dropout on the attention probabilities (thresh == 0: off); element ((b*Hq+h)*N + q)*N + key
  const int* lens;   // optional [B] (device): sample b attends to keys < lens[b] only (a short chunk padded into a batch of
                     // longer ones, infer_test_v3m2.py:370-398: the reference runs it alone, unpadded); nullptr: all N keys
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// ---- row-wise / elementwise ------------------------------------------------------------------------
// y = norm(x)*w*(1+scale[b]) + shift[b] -> bf16.  mode: 0 RMS(+w), 1 LayerNorm no affine, 2 none (cast).
hipError_t launch_norm_modulate(const float* x, const float* w, const float* shift, const float* scale,
                                int64_t mod_bstride, bf16_t* y, int M, int D, int ntok, int mode,
                                hipStream_t s);
// A[m=(b,tok)][k=c*4+p] = bf16(x[b][c][4*tok+p]) for the concatenated [x_t ; x_cond] channels.
// x_t batch index = b % B_src; x_cond batch index = b (b < cond_zero_from) else zeros (CFG uncond half).
// tvalid: optional device [B_src]: frames >= tvalid[b % B_src] of batch row b read as zero (rows shorter than T_orig).
hipError_t launch_patchify(const float* x_t, const float* x_cond, bf16_t* A, int B, int B_src,
                           int cond_zero_from, int C_t, int C_c, int T_orig, int ntok, hipStream_t s,
                           const int* tvalid = nullptr);
// sinusoidal embedding: e[b][i] = sin(t[b]*f_i), e[b][half+i] = cos(t[b]*f_i)  (jat_audiosr_v3.py:194-207)
hipError_t launch_time_sinusoid(const float* t, float* e, int B, int D, hipStream_t s);
// out[b][n] = act_out(sum_k in[b][k]*W[n][k] + bias[n]) in fp32; act_out: 0 none, 1 SiLU.
// Optionally also writes bf16(silu(out)) to out_silu_bf16 (the adaLN GEMM's A operand).
hipError_t launch_linear_f32(const float* in, const float* W, const float* bias, float* out,
                             bf16_t* out_silu_bf16, int B, int N, int K, int act_out, hipStream_t s);
// x[m][n] += gate[b][n] * (sum_z part[z*stride + m*N + n] + bias[n])   — finish of a split-K gated-residual GEMM
// finish of a split-K Linear + GELU: out bf16 = gelu(sum_z part[z] + bias (+ dual_add)); dual_rows > 0: rows m and m + dual_rows
hipError_t launch_splitk_gelu_finish(const float* part, int nsplit, int64_t stride, const float* bias, const float* dual_add,
                                     int dual_rows, bf16_t* out, int64_t ldo, int M, int N, hipStream_t s);
// finish of a split-K QKV GEMM: slice sum, RoPE, bf16 q / k rows and transposed V^T (what EPI_QKV_ROPE writes)
hipError_t launch_splitk_qkv_finish(const float* part, int nsplit, int64_t stride, const float* rope_cos, const float* rope_sin,
                                    bf16_t* q, bf16_t* k, bf16_t* vt, int M, int D, int kvD, int ntok, int npad, hipStream_t s);
// the same + the norm / modulation that consumes the updated rows (one launch instead of two)
bool splitk_resid_norm_supported(int D);
hipError_t launch_splitk_resid_norm(const float* part, int nsplit, int64_t stride, const float* bias, const float* gate,
                                    int64_t gate_bstride, float* x, const float* w, const float* shift, const float* scale,
                                    int64_t mod_bstride, bf16_t* y, int M, int D, int ntok, int mode, hipStream_t s);
hipError_t launch_splitk_resid_finish(const float* part, int nsplit, int64_t stride, const float* bias, const float* gate,
                                      int64_t gate_bstride, int ntok, float* x, int M, int N, hipStream_t s);
hipError_t launch_silu_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s);
hipError_t launch_cast_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s);
// fp32 [rows, cols] -> bf16 with the rows of every 64-row head pair-interleaved for in-lane RoPE:
// out row (h*64 + 2d + e) <- in row (h*64 + d + 32e), e in {0,1}   (rows % 64 == 0)
hipError_t launch_cast_bf16_rope_rows(const float* in, bf16_t* out, int rows, int cols, hipStream_t s);
// weight folding (sampler): out[r][c] = bf16(in[src(r)][c] * w[c] * (1 + scale[c]))  (scale may be null; rope: rows
// pair-interleaved per 64-row head like launch_cast_bf16_rope_rows)
hipError_t launch_fold_weight(const float* in, const float* w, const float* scale, bf16_t* out, int rows, int cols, int rope,
                              hipStream_t s);
// norm-folding table helpers: out[r][k] = w[k] * (1 + scale[r*in_stride + k]) ; out[r][k] = bf16(in[r*in_stride + k])
hipError_t launch_fold_scale(const float* w, const float* scale, int64_t in_stride, float* out, int64_t out_stride,
                             int rows, int cols, hipStream_t s);
hipError_t launch_gather_cast_rows(const float* in, int64_t in_stride, bf16_t* out, int rows, int cols, hipStream_t s);
// z += ((u + s(c-u)) - z)/(1-t+1e-5)*dt  (or z = x when t >= 0.999)  (infer_test_v3m2.py:161-179)
hipError_t launch_cfg_euler(const float* xp, float* z, float cfg_scale, float t, float dt, int use_cfg,
                            int64_t n_per_half, hipStream_t s);
hipError_t launch_channel_affine(const float* in, const float* mean, const float* std, float* out, int B,
                                 int C, int T, int inverse, hipStream_t s);
hipError_t launch_crossfade_pair(const float* prev, int Tp, const float* cur, int Tc, int overlap,
                                 float* out, int rows, hipStream_t s);

// ---- training step (train.hip): backward, loss, optimiser, data preparation -------------------------------------------
hipError_t launch_sum_partials(const float* part, int nsplit, int64_t stride, float* out, int64_t n, hipStream_t s);
hipError_t launch_transpose_bf16(const bf16_t* in, int64_t ld_in, int M, int C, bf16_t* out, int Mpad, hipStream_t s);
hipError_t launch_rowsum_bf16(const bf16_t* x, int64_t ld, int R, int n, float* out, hipStream_t s);
// gemm_tn.hip: dW[M,N] fp32 = dY[K,M]^T X[K,N] straight from the token-major operands (no transposed copies), optional
// split-K partial slices; column sums of a token-major matrix (bias gradients) with colsum_slices(R) * C floats of scratch
bool gemm_tn_supports(int M, int N);
// zeros: >= 16 zero bytes of device memory (needed by the 256 x 256 form; nullptr keeps the 128 x 128 kernel);
// gemm_tn_ksplit: the slice count that fills the chip for this shape
int gemm_tn_ksplit(int M, int N, int K);
hipError_t launch_gemm_tn(const bf16_t* dY, int64_t ldy, const bf16_t* X, int64_t ldx, float* dW, int64_t ldo, int M, int N, int K,
                          int ksplit, int64_t split_stride, const void* zeros, hipStream_t s);
int colsum_slices(int R);
hipError_t launch_colsum_bf16(const bf16_t* x, int64_t ld, int R, int C, float* part, float* out, hipStream_t s);
hipError_t launch_gelu_bf16(const bf16_t* in, bf16_t* out, int64_t n, DropSpec drop, hipStream_t s);
hipError_t launch_gelu_bwd(const bf16_t* pre, bf16_t* d, int64_t n, DropSpec drop, hipStream_t s);
hipError_t launch_resid_gate(const float* x_in, const bf16_t* y, const float* gate, int64_t gate_bstride, float* x_out,
                             int M, int D, int ntok, DropSpec path, DropSpec elem, hipStream_t s);
int train_nchunk(int ntok);     // token chunks per sample of the column-reduction partials
int train_red_blocks();         // blocks (= partial sums) of the scalar reductions
hipError_t launch_gate_bwd(const float* dx, const bf16_t* y, const float* gate, int64_t gate_bstride, bf16_t* dy,
                           float* part, float* dgate, int64_t dgate_bstride, int B, int D, int ntok, DropSpec path,
                           DropSpec elem, hipStream_t s);
hipError_t launch_norm_bwd(const float* x, const bf16_t* dy, const float* w, const float* scale, int64_t mod_bstride,
                           float* dx, int accumulate, float* part, float* dw_part, float* dshift, float* dscale,
                           int64_t dmod_bstride, float* dw, int B, int D, int ntok, int mode, hipStream_t s);
hipError_t launch_attention_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const bf16_t* o, const bf16_t* dout,
                                const float* lse, float* delta, bf16_t* dqkv, const float* rope_cos, const float* rope_sin,
                                int B, int N, int Hq, int Hkv, int npad, DropSpec drop, float* dkv_part, hipStream_t s);
hipError_t launch_mse_grad(const float* pred, const float* target, float* dpred, float* part, float* loss2, int64_t n,
                           float loss_scale, hipStream_t s);
// Charbonnier loss mean(sqrt((pred - target)^2 + eps)) (train_ddp_v3m2mod1.py:72-101) and d(loss * loss_scale)/d pred;
// part: train_red_blocks() floats, loss2[0] = loss
hipError_t launch_charbonnier_grad(const float* pred, const float* target, float* dpred, float* part, float* loss2, int64_t n,
                                   float eps, float loss_scale, hipStream_t s);
// v3mod2 loss (MSE + lw * (fw*freq + mw*ms + cw*cons)): dpred = d(loss * loss_scale)/d pred, out6 = {total, mse, freq, ms,
// cons, fw*freq + mw*ms + cw*cons}; part: rows*8 floats; tw: [T] (cos, sin)(2 pi m / T); lr may be null when cw == 0.
// low / strict / soft band edges (in rfft bins) are computed by the caller exactly as the reference does (int(F * ratio)).
hipError_t launch_latent_loss(const float* pred, const float* target, const float* lr, const float2* tw, float* dpred,
                              float* part, float* out6, int rows, int T, float lw, float fw, float mw, float cw, int low,
                              int strict, int soft, float loss_scale, hipStream_t s);
hipError_t launch_grad_sqsum(const float* g, int64_t n, float* part, float* norm2, hipStream_t s);
hipError_t launch_adamw(float* p, float* g, float* m, float* v, int64_t n, const float* norm2, float inv_scale,
                        float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s);
hipError_t launch_small_dw(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dW, float* db, int B, int N,
                           int K, int silu_x, hipStream_t s);
int small_dx_slab(int N);   // rows of W per partial-sum slab
hipError_t launch_small_dx(const float* dy, int64_t ldy, const void* W, int w_is_bf16, float* part, float* dx, int B, int N,
                           int K, int accumulate, const float* silu_pre, hipStream_t s);
// n independent fp32 copies described by a device table of (src, dst, count) triples, one launch
struct CopyJob { const float* src; float* dst; int64_t n; };
hipError_t launch_multi_copy(const CopyJob* jobs_dev, int njobs, hipStream_t s);
hipError_t launch_silu_f32(const float* in, float* out, int64_t n, hipStream_t s);
hipError_t launch_unpack_qkv_grad(const float* fused, float* gq, float* gk, float* gv, int D, int kvD, int K, hipStream_t s);
hipError_t launch_flow_mix(const float* x, const float* noise, const float* t, float* z, int B, int64_t per_sample, hipStream_t s);
hipError_t launch_cond_augment(float* cond, const float* noise, const float* std2, float ratio, const float* keep, int B,
                               int64_t per_sample, hipStream_t s);
hipError_t launch_tensor_std(const float* x, int64_t n, float* part, float* mean2, float* out2, hipStream_t s);
