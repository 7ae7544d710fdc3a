/*
 * jat_hip.h — C ABI of libjat_hip.so: the MI355X (gfx950) DiT flow-matching sampling path of JaTSR.
 *
 * The reference (HUSRCF/JaTSR) has no plugin/FFI layer; the interface this library replaces is the
 * Python module API of src/models/jat_audiosr_v3.py and the sampler in infer_test_v3m2.py.
 * Each entry point cites the reference symbol it stands in for (file:line relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host];
 *   - tensors are dense row-major fp32 unless a comment says otherwise;
 *   - every call enqueues its work on the caller's hipStream_t (passed as void*) and returns without
 *     synchronising; no allocation happens after *_create / *_load_weights;
 *   - return value: 0 = ok, negative = error (JAT_E_*), message via jat_last_error() (thread-local);
 *   - nothing throws across the ABI; a handle is used from one host thread at a time (one per device).
 */
#ifndef JAT_HIP_H
#define JAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAT_OK 0
#define JAT_E_INVALID (-1)   /* bad argument / unsupported shape (mirrors ValueError / AssertionError) */
#define JAT_E_HIP (-2)       /* a HIP runtime call failed */
#define JAT_E_STATE (-3)     /* weights not loaded, workspace too small, ... */
#define JAT_E_SEQLEN (-4)    /* N = ceil(T/4) > max_len (2048): jat_audiosr_v3.py:451-452 raises ValueError */

#define JAT_NORM_RMS_W 0        /* nn.RMSNorm(D, eps=1e-6) with weight: jat_audiosr_v3.py:261,264,384 */
#define JAT_NORM_LN_NOAFFINE 1  /* nn.LayerNorm(D, elementwise_affine=False, eps=1e-6): jat_audiosr_v2.py:242,245,361 */

typedef struct jat_model jat_model;
typedef struct jat_sampler jat_sampler;

/* Constructor arguments of JaT_AudioSR_V3 (jat_audiosr_v3.py:320-331); dropout / drop_path are
 * training-only and have no effect on this (eval) path. */
typedef struct jat_config {
  int32_t input_channels;  /* 1024 */
  int32_t cond_channels;   /* 1024 */
  int32_t patch_len;       /* 4 (the only supported value) */
  int32_t hidden_size;     /* D, multiple of 256 */
  int32_t depth;
  int32_t num_q_heads;     /* hidden_size / num_q_heads must be 64 */
  int32_t num_kv_heads;
  int32_t bottleneck_dim;  /* multiple of 128 */
  int32_t mlp_hidden;      /* int(hidden_size * mlp_ratio), multiple of 128 */
  int32_t norm_mode;       /* JAT_NORM_* */
} jat_config;

/* One named fp32 parameter of the reference state_dict (device pointer, [out,in] row-major for Linear
 * weights).  Names are the reference's keys, e.g. "blocks.3.attn.q_proj.weight". */
typedef struct jat_tensor_ref {
  const char* name;   /* [host] */
  const float* data;  /* device */
  int64_t numel;
} jat_tensor_ref;

const char* jat_last_error(void);
int jat_version(void);
/* Operand dtype this library was built for: 0 = bf16 (libjat_hip.so: the sampler and the V3-class trainers' autocast,
 * train_ddp_v3m2.py:545), 1 = fp16 (libjat_hip_fp16.so: torch.amp.autocast('cuda') of train_ddp_v3mod2.py:854). */
int jat_operand_dtype(void);

/* ---- model: JaT_AudioSR_V3 / _V2 (jat_audiosr_v3.py:311-471) ------------------------------------ */
int jat_model_create(const jat_config* cfg, jat_model** out);
void jat_model_destroy(jat_model* m);
/* == load_state_dict (infer_test_v3m2.py:61-74): borrows the fp32 tensors for the duration of the call
 * (synchronises `stream` before returning), packs them once into bf16 (fused [Wq;Wk;Wv], all layers'
 * adaLN weights concatenated) plus fp32 copies of biases / norm weights / t_embedder.  Unknown names
 * (e.g. the persistent RoPE buffers) are ignored; missing names are an error unless norm weights in
 * LN_NOAFFINE mode. */
int jat_model_load_weights(jat_model* m, const jat_tensor_ref* named, int32_t n, void* stream);
int jat_model_workspace_bytes(const jat_model* m, int32_t B, int32_t T, size_t* out);
/* Behaviour switches of this handle.  Their defaults come from the JAT_* environment variables, which are read ONCE, in
 * jat_model_create (INTEGRATION.md "Environment switches"); nothing on the enqueue path reads the environment.  Names:
 * "fuse_qkv_attn" (0 / 1 / 2), "qkv_split", "fuse_finish", "fold_norm" (0 / 1 / 2), "split_patch", "gemm_dbg", "fold_cap_mb".
 * Takes effect for forwards enqueued and samplers created afterwards. */
int jat_model_set_switch(jat_model* m, const char* name, int32_t value);

/* == JaT_AudioSR_V3.forward(x_t, t, x_cond) in eval mode (jat_audiosr_v3.py:422-471).
 * x_t, x_cond, x_pred: [B, input_channels, T]; t: [B].  Pads T to a multiple of 4 internally (:435-439),
 * trims on output (:468-469).  Inputs are not modified. */
int jat_forward(jat_model* m, const float* x_t, const float* t, const float* x_cond, float* x_pred,
                int32_t B, int32_t T, void* workspace, size_t workspace_bytes, void* stream);

/* == DiTBlock_GQA.forward(x, t_emb) (jat_audiosr_v3.py:284-308): x,y [B,N,D], t_emb [B,D]. */
int jat_block_forward(jat_model* m, int32_t layer, const float* x, const float* t_emb, float* y,
                      int32_t B, int32_t N, void* workspace, size_t workspace_bytes, void* stream);
/* == GroupedQueryAttention.forward(x) (jat_audiosr_v3.py:144-184): x,y [B,N,D]. */
int jat_attn_forward(jat_model* m, int32_t layer, const float* x, float* y, int32_t B, int32_t N,
                     void* workspace, size_t workspace_bytes, void* stream);
/* == t_embedder(t) (jat_audiosr_v3.py:364-369,455): t [B] -> t_emb [B,D]. */
int jat_time_embed(jat_model* m, const float* t, float* t_emb, int32_t B, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- sampler: flow_matching_sample (infer_test_v3m2.py:107-185) ------------------------------------ */
/* Builds, for a fixed (B, T, steps, cfg_scale): the schedule linspace(0,1,steps+1) (:136) as host floats,
 * the [steps, depth, 6D] adaLN modulation table (all rows of a step share one t, :150), private state
 * buffers, and ONE hipGraph holding all `steps` CFG double-batch forwards + Euler updates. */
int jat_sampler_create(jat_model* m, int32_t B, int32_t T, int32_t steps, float cfg_scale,
                       jat_sampler** out);
void jat_sampler_destroy(jat_sampler* s);
/* What the sampler's captured graph runs: *folded != 0 = per-step folded weights (0 after the fallback to the
 * norm kernels: not an RMSNorm model, over the "fold_cap_mb" switch, or the table did not fit the device), *fused_attn != 0 = the
 * fused QKV + RoPE + attention kernel, *fold_bytes = size of the folded-weight table shared through the model.  Any pointer
 * may be NULL. */
int jat_sampler_info(const jat_sampler* sampler, int32_t* folded, int32_t* fused_attn, int64_t* fold_bytes);
/* Rows of the bucket that are SHORTER than T (the last chunk of a file, infer_test_v3m2.py:353-361,370-398, batched with
 * the full-length chunks instead of sampled alone): frames[b] in (0, T] valid latent frames of row b; the caller zero-pads
 * lr_latent / z0 beyond them and ignores z_out there.  Attention masks the padded keys, every other operator is row-wise,
 * so the valid frames equal a stand-alone run of frames[b] frames.  Sticky until set again.  Not available for buckets of
 * exactly 128 tokens that run the fused QKV+attention kernel (JAT_E_STATE). */
int jat_sampler_set_lengths(jat_sampler* sampler, const int32_t* frames, int32_t n, void* stream);

/* lr_latent, z0_noise, z_out: [B, C, T].  z0_noise replaces torch.randn at :133 (caller-supplied so that
 * results are reproducible).  use_graph=0 replays the same kernels eagerly (debug / A-B timing). */
int jat_sampler_run(jat_sampler* s, const float* lr_latent, const float* z0_noise, float* z_out,
                    int32_t use_graph, void* stream);
/* One CFG combine + Euler update (infer_test_v3m2.py:161-179) in place on z [B,C,T];
 * x_pred_2B = [cond; uncond] is [2B,C,T] when cfg_scale != 1, else [B,C,T]. */
int jat_cfg_euler_step(const float* x_pred_2B, float* z, float cfg_scale, float t, float dt, int32_t B,
                       int32_t C, int32_t T, void* stream);

/* ---- chunk driver pieces (infer_test_v3m2.py:188-233, 381-394) ------------------------------------- */
/* out[c,t] = (in[c,t] - mean[c]) / std[c]   (inverse=0)  |  in[c,t]*std[c] + mean[c]   (inverse=1) */
int jat_channel_affine(const float* in, const float* mean, const float* std, float* out, int32_t B,
                       int32_t C, int32_t T, int32_t inverse, void* stream);
/* Linear crossfade of `prev` tail with `cur` head over `overlap` frames into out [rows, Tp+Tc-overlap]. */
int jat_crossfade_pair(const float* prev, int32_t Tp, const float* cur, int32_t Tc, int32_t overlap,
                       float* out, int32_t rows, void* stream);

/* ---- training step (train_ddp_v3m2.py:533-622) ------------------------------------ */
/* The step is split at its only cross-device boundary, the DDP gradient all-reduce (train_ddp_v3m2.py:486,610):
 *   jat_trainer_prepare   z_t = t x + (1-t) eps, cond noise, CFG condition dropout            (:548-579)
 *   jat_trainer_fwd_bwd   pred = model(z_t, t, cond); loss = mse_loss(pred, target); backward  (:582-610)
 *   [caller: all-reduce(grads_flat) / world_size over RCCL when world_size > 1]
 *   jat_trainer_optim     unscale, clip_grad_norm_(max_norm), AdamW, re-pack bf16 operands     (:613-619)
 * Parameters, gradients and the two AdamW moments are four caller-owned flat fp32 device buffers of `total` floats
 * (total % 4 == 0); `params` names the reference state_dict tensors as 16-byte aligned slices of params_flat, and the
 * gradient / moment of a tensor lives at the same offset of its buffer.  Gaps between tensors must be zero-filled.
 * Dropout (attention probabilities :175, MLP :269,271) and DropPath (:38-64, :300,306) draw their masks from a
 * counter-based generator keyed by (rng_seed of the step, layer, site, element index) — the same Bernoulli(1-p) / (1-p)
 * semantics as nn.Dropout / drop_path, a different random stream than torch's Philox.  Per-rank batch B <= 32. */
typedef struct jat_trainer jat_trainer;
int jat_trainer_create(jat_model* m, const jat_tensor_ref* params, int32_t n_params, float* params_flat,
                       float* grads_flat, float* exp_avg, float* exp_avg_sq, int64_t total, int32_t B, int32_t T,
                       void* stream, jat_trainer** out);
void jat_trainer_destroy(jat_trainer* tr);
/* Overlap of the gradient all-reduce with the backward (what DDP's bucket hooks do, train_ddp_v3m2.py:486): `hook` is
 * called on the host thread, during enqueue, each time the last kernel writing a contiguous slice grads_flat[off, off+n)
 * has been enqueued on the step's stream — the final layer first, then blocks depth-1 .. 0 (one slice per block,
 * 109 MB for v3mod2), then patch_embed + t_embedder; the slices tile [0, total) exactly once per jat_trainer_fwd_bwd.
 * The callee orders a collective behind the work enqueued so far (event on the stream) and returns.  NULL: off. */
int jat_trainer_set_grad_hook(jat_trainer* tr, void (*hook)(int64_t off, int64_t n, void* user), void* user);
/* Per-layer rates (host arrays [depth]): dropout[l] = the block's nn.Dropout p (jat_audiosr_v3.py:262,269,271),
 * drop_path[l] = linspace(0, drop_path_rate, depth)[l] (:372-377).  Default: all zero. */
int jat_trainer_set_regularisers(jat_trainer* tr, const float* dropout, const float* drop_path);
/* Loss of the v3mod2 trainer (train_ddp_v3mod2.py:53-321,889-896):
 *   loss = mse + latent_weight * (freq_weight * FrequencyDomainLatentLoss + ms_weight * MultiScaleLatentLoss
 *                                 + consistency_weight * HybridConsistencyLoss(pred, clean LR latent))
 * with the reference's defaults 0.3 / 0.5 / 0.5 / 0.1 and band ratios 0.3 / 0.30 / 0.36 (TrainConfig :362-372); fp32 rFFT
 * over T as in the reference (:90-95).  latent_weight == 0 (the default) is the MSE-only loss of train_ddp_v3m2.py:585. */
int jat_trainer_set_latent_loss(jat_trainer* tr, double latent_weight, double freq_weight, double ms_weight,
                                double consistency_weight, double low_freq_phase_ratio, double strict_cutoff,
                                double soft_cutoff);
/* Reconstruction loss of the V3M2-MOD1 trainer (train_ddp_v3m2mod1.py:72-101,150-151,666-672):
 *   charbonnier_loss(pred, target, eps) = mean(sqrt((pred - target)^2 + eps)),  eps = 1e-6 (ADDED to the squared difference)
 * eps > 0 selects it, eps == 0 returns to F.mse_loss.  Not combinable with the latent perceptual loss (JAT_E_STATE). */
int jat_trainer_set_charbonnier(jat_trainer* tr, double eps);
/* out6 (device): {total, mse, freq, ms, consistency, weighted latent sum} of the latest jat_trainer_fwd_bwd. */
int jat_trainer_loss_terms(jat_trainer* tr, float* out6, void* stream);
int jat_trainer_workspace_bytes(const jat_trainer* tr, size_t* out);
/* Re-derive every operand copy (bf16 weights and their transposes, fp32 operand tensors) from params_flat after the
 * caller overwrote parameters — checkpoint resume, train_ddp_v3m2.py:443-500. */
int jat_trainer_repack(jat_trainer* tr, void* stream);
/* hr_norm, noise, z_t: [B,C,T]; cond [B,Cc,T] is modified in place: cond = (cond + cond_noise * ratio *
 * (adaptive ? clamp(std(cond), 0.5, 2) : 1)) * keep[b]   (cond_noise / keep may be NULL); t [B]. */
int jat_trainer_prepare(jat_trainer* tr, const float* hr_norm, float* cond, const float* noise,
                        const float* cond_noise, float cond_noise_ratio, int32_t adaptive, const float* keep,
                        const float* t, float* z_t, void* stream);
/* Overwrites grads_flat with d(loss * loss_scale)/d(param); loss_out (device, 1 float, nullable) = unscaled loss;
 * x_pred_out (device [B,C,T], nullable) = the prediction.  cond_clean [B,C,T]: the normalised LR latent before the
 * condition noise (lr_norm_original, train_ddp_v3mod2.py:861), read only by the consistency loss (may be NULL otherwise). */
int jat_trainer_fwd_bwd(jat_trainer* tr, const float* z_t, const float* t, const float* x_cond, const float* target,
                        const float* cond_clean, float loss_scale, uint64_t rng_seed, float* loss_out, float* x_pred_out,
                        void* stream);
/* grads_flat is read, not modified (clip_grad_norm_'s in-place scaling of .grad is not reproduced: nothing reads it).
 * grad_norm_out (device, 1 float, nullable) = L2 norm of the loss-SCALED gradients (divide by loss_scale).  A
 * non-finite norm leaves parameters and moments untouched (GradScaler.step).  `step` is 1-based (bias correction). */
int jat_trainer_optim(jat_trainer* tr, float lr, float beta1, float beta2, float eps, float weight_decay,
                      float max_grad_norm, float loss_scale, int32_t step, float* grad_norm_out, void* stream);

/* ---- per-kernel entry points (unit parity tests; bench roofline leg) --------------------------------- */
/* y_bf16[M,D] = norm(x[M,D]) (* w) * (1 + scale[b]) + shift[b], b = row / rows_per_batch;
 * shift/scale may be NULL (no modulation); mod_bstride = element stride between batches (0 = shared). */
int jat_k_norm_modulate(const float* x, const float* w, const float* shift, const float* scale,
                        int64_t mod_bstride, uint16_t* y_bf16, int32_t M, int32_t D, int32_t rows_per_batch,
                        int32_t norm_mode, void* stream);
/* C[M,N] (+bias) = A_bf16[M,K] * W_bf16[N,K]^T ; epilogue: 0 = fp32 out, 1 = bf16 out, 2 = bf16 GELU(erf),
 * 3 = fp32 out += gate[b]*(acc+bias) (gate [B, N] with stride gate_bstride).  variant selects the tile
 * configuration (0 = default). */
int jat_k_gemm(const uint16_t* A, const uint16_t* W, const float* bias, void* C, int32_t M, int32_t N,
               int32_t K, int32_t epilogue, const float* gate, int64_t gate_bstride, int32_t rows_per_batch,
               int32_t variant, void* stream);
/* The same GEMM with the sampler's norm folding (computes jat_audiosr_v3.py:297-306 for rows that share one
 * modulation): the residual stream lives as two 16-bit planes x = hi + lo.
 *   producer (hi != NULL; epilogue 0 or 3): x_new = acc + bias (0) | (hi + lo) + gate[b] * (acc + bias) (3), written back
 *     as hi = round(x_new), lo = round(x_new - hi), plus part_out[M, N / jat_k_gemm_wave_n(variant)]: partial row sums of
 *     x_new^2 in fixed order.  C is not touched.
 *   consumer (part_in != NULL; any epilogue): accumulator row m scaled by rsqrt(sum_j part_in[m][j] / K + 1e-6) before the
 *     bias; part_in_np in {4, 8, 16}.
 * Variants with the coalesced epilogue only (>= 18). */
int jat_k_gemm_fold(const uint16_t* A, const uint16_t* W, const float* bias, void* C, int32_t M, int32_t N, int32_t K,
                    int32_t epilogue, const float* gate, int64_t gate_bstride, int32_t rows_per_batch, uint16_t* hi,
                    uint16_t* lo, float* part_out, const float* part_in, int32_t part_in_np, int32_t variant, void* stream);
/* Split-K slices of the same product: parts[z][M][N] fp32 = A[:, z K/ksplit : (z+1) K/ksplit] W[:, same]^T, z < ksplit, no
 * bias; summed in order by the caller / the finishing pass.  This is what the un-folded forward's fc2 and out_proj
 * (jat_audiosr_v3.py:300,306) launch when their tiles would leave CUs idle; variant 39 = the 224 x 160 k-step-pair tile
 * (M % 224 == 0, N % 160 == 0, K / ksplit a multiple of 64 and >= 192). */
int jat_k_gemm_splitk(const uint16_t* A, const uint16_t* W, float* parts, int32_t M, int32_t N, int32_t K, int32_t ksplit,
                      int32_t variant, void* stream);
/* Fused QKV projection + RoPE + GQA attention for 128-token samples (the sampler's form of jat_audiosr_v3.py:154-181 when
 * B * Hkv blocks fill the chip): A [M, K] (M % 128 == 0), Wg = the group-major fused weight [Hkv][5*64 + 64 + 64][K] with q / k rows
 * pair-interleaved per head (as jat_model_load_weights packs it), out [M, Hkv*320] = attention output; bias / part_in as in
 * jat_k_gemm_fold's consumer side (folded norms). */
int jat_k_qkv_attn(const uint16_t* A, const uint16_t* Wg, const float* bias, uint16_t* out, int32_t M, int32_t Hkv, int32_t K,
                   const float* rope_inv_freq, const float* part_in, int32_t part_in_np, void* stream);
/* columns per wave tile of a GEMM tile variant (the slot width of part_out); 0 for an unknown variant */
int jat_k_gemm_wave_n(int32_t variant);
/* Weight gradient of y = x W^T + b from token-major operands: dW[out,in] = dY[tokens,out]^T X[tokens,in] (fp32), db[out] =
 * column sums of dY (db may be NULL).  out and in multiples of 128; ksplit >= 1 slices of the token axis summed in order
 * (0 = the count that fills the chip, at most 16); work: 256 + (ksplit > 1 ? ksplit*out*in*4 : 0) + 32*out*4 bytes.  (The backward of every nn.Linear of
 * models/JaT_V3.py under train_ddp_v3m2.py:601.) */
int jat_k_weight_grad(const uint16_t* dY, const uint16_t* X, float* dW, float* db, int32_t tokens, int32_t out, int32_t in,
                      int32_t ksplit, void* work, size_t work_bytes, void* stream);
/* GQA attention on bf16 q[M,Hq*64], k[M,Hkv*64], vt[B,Hkv,64,Npad] -> o[M,Hq*64]; softmax(q k^T / 8) v. */
int jat_k_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* o, int32_t B,
                    int32_t N, int32_t Hq, int32_t Hkv, int32_t Npad, void* stream);
/* The v3mod2 loss (see jat_trainer_set_latent_loss) on pred / target / clean-LR tensors [rows, T]: dpred = d(total *
 * loss_scale)/d pred, out6 = {total, mse, freq, ms, consistency, weighted latent sum}; work: T*8 (rounded up to 256)
 * + rows*32 bytes of device scratch. */
int jat_k_latent_loss(const float* pred, const float* target, const float* lr, float* dpred, float* out6,
                      int32_t rows, int32_t T, double latent_weight, double freq_weight, double ms_weight,
                      double consistency_weight, double low_freq_phase_ratio, double strict_cutoff, double soft_cutoff,
                      float loss_scale, void* work, size_t work_bytes, void* stream);
/* Reconstruction loss on n elements: eps == 0: F.mse_loss (train_ddp_v3m2.py:585), eps > 0: charbonnier_loss
 * (train_ddp_v3m2mod1.py:72-101); dpred = d(loss * loss_scale)/d pred, loss_out: 1 float; work: >= 4104 bytes. */
int jat_k_recon_loss(const float* pred, const float* target, float* dpred, float* loss_out, int64_t n, double eps,
                     float loss_scale, void* work, size_t work_bytes, void* stream);
/* fp32 -> bf16 (round-to-nearest-even) */
int jat_k_cast_bf16(const float* in, uint16_t* out, int64_t n, void* stream);

/* ---- measurement aid (bench.py roofline leg; no reference counterpart) ------------------------------------ */
/* (state lives in the model handle: two models in one process do not share a bracket)
 * Bracket every GEMM launch of one call site (0 qkv, 1 out_proj, 2 MLP fc1, 3 MLP fc2, 4 other; -1 = off) with a
 * HIP event pair on the launch stream, for at most max_launches launches.  Eager calls only. */
int jat_prof_gemm_site(jat_model* m, int32_t site, int32_t max_launches);
/* Sum of the bracketed launch durations [host ms], their count, algorithmic FLOPs and the tile variant used;
 * synchronises on the recorded events and switches the bracket off. */
int jat_prof_collect(jat_model* m, double* total_ms, int32_t* launches, double* flops, int32_t* variant);

#ifdef __cplusplus
}
#endif
#endif /* JAT_HIP_H */
