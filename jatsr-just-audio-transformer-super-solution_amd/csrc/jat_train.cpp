// Training step of the DiT: the body of train_ddp_v3m2.py:533-622 split at its only
// cross-device boundary.  jat_trainer_fwd_bwd = model(z_t, t, cond) -> mse_loss -> backward into a flat fp32 gradient
// buffer; the caller all-reduces that buffer (RCCL) when world_size > 1; jat_trainer_optim = clip_grad_norm_ + AdamW +
// re-pack of the bf16 operand copies.  Sequencing only: the arithmetic lives in gemm.hip / attention.hip /
// elementwise.hip / train.hip.
//
// Every Linear runs its three GEMMs on gemm_bf16_kernel (C = A W^T, both operands K-contiguous):
//   y  = x  W^T        A = x [M,in]          W = W   [out,in]
//   dx = dy W          A = dy [M,out]        W = W^T [in,out]     (transposed bf16 copy, rebuilt after every update)
//   dW = dy^T x        A = dy^T [out,Mpad]   W = x^T [in,Mpad]    (transpose_bf16_kernel, token axis zero-padded to 64)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "jat_internal.h"

namespace {

struct TLayer {
  float *x_mid, *lse;
  bf16_t *xn1, *q, *k, *vt, *ao, *y_attn, *xn2, *h_pre, *h_post, *y_mlp;
  bf16_t *wqkvT, *woT, *w1T, *w2T;
  // gradient / master-parameter offsets (floats) into the flat buffers
  int64_t o_n1, o_n2, o_q, o_k, o_v, o_o, o_w1, o_b1, o_w2, o_b2, o_ada_w, o_ada_b;
};

}  // namespace

struct jat_trainer {
  jat_model* m = nullptr;
  int B = 0, T = 0, ntok = 0, M = 0, Mpad = 0, npad = 0;
  float *P = nullptr, *G = nullptr, *m1 = nullptr, *m2 = nullptr;
  int64_t total = 0;
  std::vector<std::string> names;
  std::vector<jat_tensor_ref> prefs;   // name -> pointer into P (for the re-pack)
  char* blob = nullptr;
  size_t blob_bytes = 0;
  std::vector<float*> x;               // depth + 1 residual-stream snapshots [M, D]
  std::vector<TLayer> L;
  bf16_t *a_patch, *pe_pre, *pe_h, *xnf, *t_silu, *pe_w2T, *wfinalT;
  float *e_sin, *u1, *t_h, *t_emb, *mod, *pred;
  // backward scratch
  float *dx, *dpred, *dmod, *part, *red_part, *scal, *delta, *dwqkv, *dt_emb, *du1, *small_part;
  bf16_t *dy, *dh, *dxn, *dao, *dqkv, *tA, *tB, *dyf;
  int64_t o_pe_w1, o_pe_b1, o_pe_w2, o_pe_b2, o_te_w1, o_te_b1, o_te_w2, o_te_b2, o_fn, o_wf, o_bf;
  bool rms = true;
  // gradient-ready hook (DDP overlap): called on the host, during enqueue, once the last kernel writing the slice
  // grads_flat[off, off + n) has been enqueued — final layer first, then blocks depth-1 .. 0, then patch embed + t_embedder
  void (*hook)(int64_t off, int64_t n, void* user) = nullptr;
  void* hook_user = nullptr;
  std::vector<int64_t> block_lo;       // first float of block l's parameters (depth + 1 entries: [depth] = final layer)
  std::vector<float> p_drop, p_path;   // per-layer Dropout / DropPath rates (jat_trainer_set_regularisers); default 0
  uint64_t seed = 0;                   // RNG seed of the step in flight (masks are recomputed by the backward)
  CopyJob* copy_jobs = nullptr;        // device table: fp32 master slices -> the model's fp32 operand tensors
  int n_copy_jobs = 0;
  float* dw_part = nullptr;
  // v3mod2 latent perceptual loss (jat_trainer_set_latent_loss); lw == 0: plain MSE (the V3 trainer)
  double lw = 0.0, fw = 0.5, mw = 0.5, cw = 0.1, phase_ratio = 0.3, strict_cut = 0.30, soft_cut = 0.36;
  double charb_eps = 0.0;              // > 0: Charbonnier reconstruction loss (train_ddp_v3m2mod1.py:72-101) instead of MSE
  float2* tw = nullptr;                // [T] twiddles
  float *ll_part = nullptr, *terms = nullptr;
  float* dw_split = nullptr;           // split-K partials of the small dW GEMMs
  const void* zero_cell = nullptr;     // 256 zero bytes (the workspace is zeroed once and this cell is never written)
  float* colsum_part = nullptr;        // row-slice partials of the bias-gradient column sums
  bool tn_dw = true;                   // dW straight from token-major operands (JAT_TN_DW=0: transposed copies + gemm_bf16_kernel)
  float* dkv_part = nullptr;           // per-query-head fp32 partials of dK / dV (attention backward)
  int64_t split4_area = 0, split2_area = 0;
  // Weight gradients on a second stream (JAT_DW_STREAM=1).  dW = dY^T X of a Linear is off the critical path of the backward (nothing
  // reads it before the gradient norm), while the dX chain in front of it alternates MFMA-bound GEMMs with HBM- / VALU-bound
  // passes (norm and gate backward, GELU', attention backward): with the dW GEMMs queued beside that chain the chip works on a
  // GEMM while the chain is in a memory pass.  The four gradient operands a layer hands to its dW GEMMs (dy of the MLP branch,
  // dh, dy of the attention branch, dqkv) are double-buffered by layer parity, so the chain runs up to one layer ahead of the dW
  // stream; events order producer -> dW (ready) and dW -> next writer of the same buffer (done).
  bool dw_async = false;
  hipStream_t dw_stream = nullptr;
  hipEvent_t ev_ready[4][2] = {}, ev_done[4][2] = {}, ev_mod[2] = {}, ev_join = nullptr, ev_wt = nullptr;
  std::vector<hipEvent_t> ev_layer;    // block l's weight gradients are complete (gates the gradient-ready hook)
  bf16_t *dy_m[2] = {}, *dh_b[2] = {}, *dy_a[2] = {}, *dq_b[2] = {};
};

namespace {

int build_transposes(jat_trainer* tr, hipStream_t s) {
  jat_model* m = tr->m;
  const int D = m->D, Nqkv = D + 2 * m->kvD;
  for (int l = 0; l < m->depth; ++l) {
    const LayerW& W = m->layers[l];
    TLayer& L = tr->L[l];
    KCHK(launch_transpose_bf16(W.wqkv, D, Nqkv, D, L.wqkvT, Nqkv, s));
    KCHK(launch_transpose_bf16(W.wo, D, D, D, L.woT, D, s));
    KCHK(launch_transpose_bf16(W.w1, D, m->mlp, D, L.w1T, m->mlp, s));
    KCHK(launch_transpose_bf16(W.w2, m->mlp, D, m->mlp, L.w2T, D, s));
  }
  KCHK(launch_transpose_bf16(m->pe_w2, m->bott, D, m->bott, tr->pe_w2T, D, s));
  KCHK(launch_transpose_bf16(m->wfinal, D, m->Fout, D, tr->wfinalT, m->Fout, s));
  return JAT_OK;
}

// After an optimiser step: fp32 master -> the operand copies the kernels read.  No host synchronisation; the small
// fp32 tensors go in one table-driven launch, the GEMM weights in one cast per matrix, then the transposed copies.
// The group-major QKV copy of the fused inference kernel is NOT rebuilt (m->group_copy_stale).
int repack(jat_trainer* tr, hipStream_t s) {
  jat_model* m = tr->m;
  const float* P = tr->P;
  const int D = m->D, kvD = m->kvD, mlp = m->mlp;
  m->group_copy_stale = true;
  m->fold_src_ok = false;    // the sampler's folded-weight tables (jat_api.cpp FoldTable) describe the previous weights:
  m->fold_cache.clear();     // samplers created from now on run the norm kernels until the next full load
  KCHK(launch_multi_copy(tr->copy_jobs, tr->n_copy_jobs, s));
  KCHK(launch_cast_bf16(P + tr->o_pe_w1, m->pe_w1, (int64_t)m->bott * m->Kp, s));
  KCHK(launch_cast_bf16(P + tr->o_pe_w2, m->pe_w2, (int64_t)D * m->bott, s));
  KCHK(launch_cast_bf16(P + tr->o_wf, m->wfinal, (int64_t)m->Fout * D, s));
  for (int l = 0; l < m->depth; ++l) {
    const LayerW& W = m->layers[l];
    const TLayer& L = tr->L[l];
    KCHK(launch_cast_bf16_rope_rows(P + L.o_q, W.wqkv, D, D, s));
    KCHK(launch_cast_bf16_rope_rows(P + L.o_k, W.wqkv + (int64_t)D * D, kvD, D, s));
    KCHK(launch_cast_bf16(P + L.o_v, W.wqkv + (int64_t)(D + kvD) * D, (int64_t)kvD * D, s));
    KCHK(launch_cast_bf16(P + L.o_o, W.wo, (int64_t)D * D, s));
    KCHK(launch_cast_bf16(P + L.o_w1, W.w1, (int64_t)mlp * D, s));
    KCHK(launch_cast_bf16(P + L.o_w2, W.w2, (int64_t)D * mlp, s));
    KCHK(launch_cast_bf16(P + L.o_ada_w, m->wada + (int64_t)l * 6 * D * D, (int64_t)6 * D * D, s));
  }
  if (tr->dw_async) {
    // the transposed copies are read by the NEXT backward only (dX = dY W): rebuilt on the second stream, under the next forward
    // (HBM-bound copies beside MFMA-bound GEMMs); backward_train waits for ev_wt before its first dX GEMM
    HIPCHK(hipEventRecord(tr->ev_join, s));
    HIPCHK(hipStreamWaitEvent(tr->dw_stream, tr->ev_join, 0));
    JCHK(build_transposes(tr, tr->dw_stream));
    HIPCHK(hipEventRecord(tr->ev_wt, tr->dw_stream));
    return JAT_OK;
  }
  return build_transposes(tr, s);
}

// dW[out,in] = dY^T X and (optionally) db[out] = column sums of dY, from dY bf16 [M,out] and X bf16 [M,in]
int weight_grad(jat_trainer* tr, const bf16_t* dY, int out, const bf16_t* X, int in, float* dW, float* db, hipStream_t s) {
  // M x N tiles of a small weight do not fill 256 CUs while K = all tokens is long: split K, sum the partials in order
  const int64_t area = (int64_t)out * in;
  const int split = area <= tr->split4_area ? 4 : (area <= tr->split2_area ? 2 : 1);
  if (tr->tn_dw && gemm_tn_supports(out, in)) {   // straight from the token-major operands (gemm_tn.hip)
    const int ks = gemm_tn_ksplit(out, in, tr->M);
    KCHK(launch_gemm_tn(dY, out, X, in, ks > 1 ? tr->dw_split : dW, in, out, in, tr->M, ks, area, tr->zero_cell, s));
    if (ks > 1) KCHK(launch_sum_partials(tr->dw_split, ks, area, dW, area, s));
    if (db) KCHK(launch_colsum_bf16(dY, out, tr->M, out, tr->colsum_part, db, s));
    return JAT_OK;
  }
  KCHK(launch_transpose_bf16(dY, out, tr->M, out, tr->tA, tr->Mpad, s));
  KCHK(launch_transpose_bf16(X, in, tr->M, in, tr->tB, tr->Mpad, s));
  GemmArgs e{};
  e.ldo = in; e.ntok = out;
  if (split > 1) {
    e.out = tr->dw_split; e.ksplit = split; e.split_stride = area;
    JCHK(jat_gemm(tr->m, G_OTHER, tr->tA, tr->Mpad, tr->tB, tr->Mpad, out, in, tr->Mpad, EPI_F32, e, s));
    KCHK(launch_sum_partials(tr->dw_split, split, area, dW, area, s));
  } else {
    e.out = dW;
    JCHK(jat_gemm(tr->m, G_OTHER, tr->tA, tr->Mpad, tr->tB, tr->Mpad, out, in, tr->Mpad, EPI_F32, e, s));
  }
  if (db) KCHK(launch_rowsum_bf16(tr->tA, tr->Mpad, out, tr->Mpad, db, s));
  return JAT_OK;
}

// dX bf16 [M,in] = dY [M,out] W, with WT = W^T [in,out]
int input_grad(jat_trainer* tr, const bf16_t* dY, int out, const bf16_t* WT, int in, bf16_t* dX, hipStream_t s) {
  GemmArgs e{};
  e.out = dX; e.ldo = in; e.ntok = tr->ntok;
  return jat_gemm(tr->m, G_OTHER, dY, out, WT, out, tr->M, in, out, EPI_BF16, e, s);
}

// mask sites of layer l (jat_rng.h): 0 attention probabilities, 1 DropPath(attn), 2 MLP after GELU, 3 MLP out, 4 DropPath(MLP)
DropSpec site(const jat_trainer* tr, int l, int kind) {
  const float p = (kind == 1 || kind == 4) ? tr->p_path[l] : tr->p_drop[l];
  return jat_drop_spec(tr->seed, (uint32_t)(l * 8 + kind), p);
}
const DropSpec kNoDrop = {0u, 0u, 0u, 1.0f};

int forward_train(jat_trainer* tr, const float* z_t, const float* t, const float* x_cond, hipStream_t s) {
  jat_model* m = tr->m;
  const int B = tr->B, T = tr->T, ntok = tr->ntok, M = tr->M, D = m->D, Nqkv = D + 2 * m->kvD;
  const int64_t mstride = (int64_t)m->depth * 6 * D;
  // t_embedder (fp32; jat_audiosr_v3.py:364-369) with the pre-activation kept for the backward
  KCHK(launch_time_sinusoid(t, tr->e_sin, B, D, s));
  KCHK(launch_linear_f32(tr->e_sin, m->te_w1, m->te_b1, tr->u1, nullptr, B, D, D, 0, s));
  KCHK(launch_silu_f32(tr->u1, tr->t_h, (int64_t)B * D, s));
  KCHK(launch_linear_f32(tr->t_h, m->te_w2, m->te_b2, tr->t_emb, tr->t_silu, B, D, D, 0, s));
  {
    GemmArgs e{};
    e.out = tr->mod; e.ldo = mstride; e.bias = m->bada; e.ntok = 1;
    JCHK(jat_gemm(m, G_OTHER, tr->t_silu, D, m->wada, D, B, (int)mstride, D, EPI_F32, e, s));
  }
  KCHK(launch_patchify(z_t, x_cond, tr->a_patch, B, B, B, m->Cin, m->Cc, T, ntok, s));
  {
    GemmArgs e{};
    e.out = tr->pe_pre; e.ldo = m->bott; e.bias = m->pe_b1; e.ntok = ntok;
    JCHK(jat_gemm(m, G_OTHER, tr->a_patch, m->Kp, m->pe_w1, m->Kp, M, m->bott, m->Kp, EPI_BF16, e, s));
    KCHK(launch_gelu_bf16(tr->pe_pre, tr->pe_h, (int64_t)M * m->bott, kNoDrop, s));
    GemmArgs f{};
    f.out = tr->x[0]; f.ldo = D; f.bias = m->pe_b2; f.ntok = ntok;
    JCHK(jat_gemm(m, G_OTHER, tr->pe_h, m->bott, m->pe_w2, m->bott, M, D, m->bott, EPI_F32, f, s));
  }
  for (int l = 0; l < m->depth; ++l) {
    const LayerW& W = m->layers[l];
    TLayer& L = tr->L[l];
    const float* mod = tr->mod + (int64_t)l * 6 * D;
    KCHK(launch_norm_modulate(tr->x[l], W.norm1, mod + 0 * D, mod + 1 * D, mstride, L.xn1, M, D, ntok, m->cfg.norm_mode, s));
    {
      GemmArgs e{};
      e.out = L.q; e.k_out = L.k; e.vt_out = L.vt; e.D = D; e.kvD = m->kvD; e.npad = tr->npad; e.ntok = ntok;
      e.rope_cos = m->rope_cos; e.rope_sin = m->rope_sin; e.rope_inv_freq = m->rope_invf;
      JCHK(jat_gemm(m, G_QKV, L.xn1, D, W.wqkv, D, M, Nqkv, D, EPI_QKV_ROPE, e, s));
    }
    {
      AttnArgs a{};
      a.q = L.q; a.k = L.k; a.vt = L.vt; a.o = L.ao; a.ldq = D; a.ldk = m->kvD; a.ldo = D;
      a.B = B; a.N = ntok; a.Hq = m->Hq; a.Hkv = m->Hkv; a.npad = tr->npad;
      a.scale_log2e = 0.125f * 1.4426950408889634f;
      a.lse = L.lse;
      a.drop = site(tr, l, 0);
      KCHK(launch_attention(a, s));
    }
    {
      GemmArgs e{};
      e.out = L.y_attn; e.ldo = D; e.ntok = ntok;
      JCHK(jat_gemm(m, G_OUT, L.ao, D, W.wo, D, M, D, D, EPI_BF16, e, s));
    }
    KCHK(launch_resid_gate(tr->x[l], L.y_attn, mod + 2 * D, mstride, L.x_mid, M, D, ntok, site(tr, l, 1), kNoDrop, s));
    KCHK(launch_norm_modulate(L.x_mid, W.norm2, mod + 3 * D, mod + 4 * D, mstride, L.xn2, M, D, ntok, m->cfg.norm_mode, s));
    {
      GemmArgs e{};
      e.out = L.h_pre; e.ldo = m->mlp; e.bias = W.b1; e.ntok = ntok;
      JCHK(jat_gemm(m, G_FC1, L.xn2, D, W.w1, D, M, m->mlp, D, EPI_BF16, e, s));
      KCHK(launch_gelu_bf16(L.h_pre, L.h_post, (int64_t)M * m->mlp, site(tr, l, 2), s));
      GemmArgs f{};
      f.out = L.y_mlp; f.ldo = D; f.bias = W.b2; f.ntok = ntok;
      JCHK(jat_gemm(m, G_FC2, L.h_post, m->mlp, W.w2, m->mlp, M, D, m->mlp, EPI_BF16, f, s));
    }
    KCHK(launch_resid_gate(L.x_mid, L.y_mlp, mod + 5 * D, mstride, tr->x[l + 1], M, D, ntok, site(tr, l, 4), site(tr, l, 3), s));
  }
  KCHK(launch_norm_modulate(tr->x[m->depth], m->final_norm, nullptr, nullptr, 0, tr->xnf, M, D, ntok, m->cfg.norm_mode, s));
  {
    GemmArgs e{};
    e.out = tr->pred; e.bias = m->bfinal; e.ntok = ntok; e.C_out = m->Cin; e.T_orig = T;
    JCHK(jat_gemm(m, G_OTHER, tr->xnf, D, m->wfinal, D, M, m->Fout, D, EPI_UNPATCH, e, s));
  }
  return JAT_OK;
}

int backward_train(jat_trainer* tr, const float* target, const float* cond_clean, float loss_scale, hipStream_t s) {
  jat_model* m = tr->m;
  const int B = tr->B, T = tr->T, ntok = tr->ntok, M = tr->M, D = m->D, Nqkv = D + 2 * m->kvD, mode = m->cfg.norm_mode;
  const int64_t mstride = (int64_t)m->depth * 6 * D;
  float* G = tr->G;
  if (tr->lw != 0.0) {
    const int F = T / 2 + 1;   // band edges exactly as the reference computes them: int(freq_bins * ratio) in double
    KCHK(launch_latent_loss(tr->pred, target, cond_clean, tr->tw, tr->dpred, tr->ll_part, tr->terms, B * m->Cin, T,
                            (float)tr->lw, (float)tr->fw, (float)tr->mw, (float)tr->cw, (int)((double)F * tr->phase_ratio),
                            (int)((double)F * tr->strict_cut), (int)((double)F * tr->soft_cut), loss_scale, s));
    HIPCHK(hipMemcpyAsync(tr->scal, tr->terms, 4, hipMemcpyDeviceToDevice, s));
  } else if (tr->charb_eps > 0.0) {
    KCHK(launch_charbonnier_grad(tr->pred, target, tr->dpred, tr->red_part, tr->scal, (int64_t)B * m->Cin * T,
                                 (float)tr->charb_eps, loss_scale, s));
  } else {
    KCHK(launch_mse_grad(tr->pred, target, tr->dpred, tr->red_part, tr->scal, (int64_t)B * m->Cin * T, loss_scale, s));
  }
  // Where a weight gradient runs: on `s` itself, or (dw_async) on the second stream behind an event of the kernel that
  // finished its gradient operand; `done` is what the next writer of that operand buffer waits for.
  hipStream_t ws = tr->dw_async ? tr->dw_stream : s;
  auto dw_begin = [&](hipEvent_t ready) -> int {
    if (!tr->dw_async) return JAT_OK;
    HIPCHK(hipEventRecord(ready, s));
    HIPCHK(hipStreamWaitEvent(ws, ready, 0));
    return JAT_OK;
  };
  auto dw_end = [&](hipEvent_t done) -> int {
    if (tr->dw_async) HIPCHK(hipEventRecord(done, ws));
    return JAT_OK;
  };
  auto before_write = [&](hipEvent_t done) -> int {   // a never-recorded event does not block
    if (tr->dw_async) HIPCHK(hipStreamWaitEvent(s, done, 0));
    return JAT_OK;
  };
  // final layer: Linear (unpatchify^T is a patchify of dpred) and the un-modulated norm
  KCHK(launch_patchify(tr->dpred, nullptr, tr->dyf, B, B, B, m->Cin, 0, T, ntok, s));
  if (tr->dw_async) HIPCHK(hipStreamWaitEvent(s, tr->ev_wt, 0));   // the transposed weight copies of the last re-pack (repack())
  JCHK(input_grad(tr, tr->dyf, m->Fout, tr->wfinalT, D, tr->dxn, s));
  if (tr->dw_async) { HIPCHK(hipEventRecord(tr->ev_join, s)); HIPCHK(hipStreamWaitEvent(ws, tr->ev_join, 0)); }
  JCHK(weight_grad(tr, tr->dyf, m->Fout, tr->xnf, D, G + tr->o_wf, G + tr->o_bf, ws));
  if (tr->dw_async) HIPCHK(hipEventRecord(tr->ev_layer[m->depth], ws));
  KCHK(launch_norm_bwd(tr->x[m->depth], tr->dxn, m->final_norm, nullptr, 0, tr->dx, 0, tr->part, tr->dw_part, nullptr, nullptr, 0,
                       tr->rms ? G + tr->o_fn : nullptr, B, D, ntok, mode, s));
  // gradient-ready hook of a block: with the second stream its weight gradients may still be in flight when the chain moves on,
  // so the hook of block l fires one block later, behind an event of the dW stream (the exchange stream orders itself after `s`)
  int pending_hook = -1;
  auto fire_hook = [&](int blk) -> int {
    if (!tr->hook) return JAT_OK;
    if (tr->dw_async) HIPCHK(hipStreamWaitEvent(s, tr->ev_layer[blk], 0));
    tr->hook(tr->block_lo[blk], (blk == m->depth ? tr->total : tr->block_lo[blk + 1]) - tr->block_lo[blk], tr->hook_user);
    return JAT_OK;
  };
  if (tr->dw_async) pending_hook = m->depth; else JCHK(fire_hook(m->depth));
  for (int l = m->depth - 1; l >= 0; --l) {
    TLayer& L = tr->L[l];
    const int par = l & 1;
    bf16_t *dy_m = tr->dy_m[par], *dh = tr->dh_b[par], *dy_a = tr->dy_a[par], *dq = tr->dq_b[par];
    const float* mod = tr->mod + (int64_t)l * 6 * D;
    float* dmod = tr->dmod + (int64_t)l * 6 * D;
    // x_out = x_mid + gate_mlp * mlp(norm2(x_mid) * (1 + scale_mlp) + shift_mlp)          jat_audiosr_v3.py:303-306
    JCHK(before_write(tr->ev_done[0][par]));
    KCHK(launch_gate_bwd(tr->dx, L.y_mlp, mod + 5 * D, mstride, dy_m, tr->part, dmod + 5 * D, mstride, B, D, ntok,
                         site(tr, l, 4), site(tr, l, 3), s));
    JCHK(before_write(tr->ev_done[1][par]));
    JCHK(input_grad(tr, dy_m, D, L.w2T, m->mlp, dh, s));
    JCHK(dw_begin(tr->ev_ready[0][par]));
    JCHK(weight_grad(tr, dy_m, D, L.h_post, m->mlp, G + L.o_w2, G + L.o_b2, ws));
    JCHK(dw_end(tr->ev_done[0][par]));
    KCHK(launch_gelu_bwd(L.h_pre, dh, (int64_t)M * m->mlp, site(tr, l, 2), s));
    JCHK(input_grad(tr, dh, m->mlp, L.w1T, D, tr->dxn, s));
    JCHK(dw_begin(tr->ev_ready[1][par]));
    JCHK(weight_grad(tr, dh, m->mlp, L.xn2, D, G + L.o_w1, G + L.o_b1, ws));
    JCHK(dw_end(tr->ev_done[1][par]));
    KCHK(launch_norm_bwd(L.x_mid, tr->dxn, m->layers[l].norm2, mod + 4 * D, mstride, tr->dx, 1, tr->part, tr->dw_part, dmod + 3 * D,
                         dmod + 4 * D, mstride, tr->rms ? G + L.o_n2 : nullptr, B, D, ntok, mode, s));
    // x_mid = x_in + gate_msa * out_proj(attn(norm1(x_in) * (1 + scale_msa) + shift_msa))   :297-300
    JCHK(before_write(tr->ev_done[2][par]));
    KCHK(launch_gate_bwd(tr->dx, L.y_attn, mod + 2 * D, mstride, dy_a, tr->part, dmod + 2 * D, mstride, B, D, ntok,
                         site(tr, l, 1), kNoDrop, s));
    JCHK(input_grad(tr, dy_a, D, L.woT, D, tr->dao, s));
    JCHK(dw_begin(tr->ev_ready[2][par]));
    JCHK(weight_grad(tr, dy_a, D, L.ao, D, G + L.o_o, nullptr, ws));
    JCHK(dw_end(tr->ev_done[2][par]));
    JCHK(before_write(tr->ev_done[3][par]));
    KCHK(launch_attention_bwd(L.q, L.k, L.vt, L.ao, tr->dao, L.lse, tr->delta, dq, m->rope_cos, m->rope_sin, B, ntok,
                              m->Hq, m->Hkv, tr->npad, site(tr, l, 0), tr->dkv_part, s));
    JCHK(input_grad(tr, dq, Nqkv, L.wqkvT, D, tr->dxn, s));
    JCHK(dw_begin(tr->ev_ready[3][par]));
    JCHK(weight_grad(tr, dq, Nqkv, L.xn1, D, tr->dwqkv, nullptr, ws));
    KCHK(launch_unpack_qkv_grad(tr->dwqkv, G + L.o_q, G + L.o_k, G + L.o_v, D, m->kvD, D, ws));
    JCHK(dw_end(tr->ev_done[3][par]));
    KCHK(launch_norm_bwd(tr->x[l], tr->dxn, m->layers[l].norm1, mod + 1 * D, mstride, tr->dx, 1, tr->part, tr->dw_part, dmod + 0 * D,
                         dmod + 1 * D, mstride, tr->rms ? G + L.o_n1 : nullptr, B, D, ntok, mode, s));
    // adaLN modulation Linear(SiLU(t_emb)) of this block (:275-278): its six dmod slices are complete now (a weight gradient
    // like the others: nothing in the chain reads it)
    JCHK(dw_begin(tr->ev_mod[par]));
    KCHK(launch_small_dw(dmod, mstride, tr->t_emb, D, G + L.o_ada_w, G + L.o_ada_b, B, 6 * D, D, 1, ws));
    if (tr->dw_async) HIPCHK(hipEventRecord(tr->ev_layer[l], ws));
    if (tr->dw_async) {
      if (pending_hook >= 0) JCHK(fire_hook(pending_hook));
      pending_hook = l;
    } else {
      JCHK(fire_hook(l));
    }
  }
  if (tr->dw_async) {   // join: everything below (and the optimiser) runs on `s` alone again and may reuse the dW scratch
    HIPCHK(hipEventRecord(tr->ev_join, ws));
    HIPCHK(hipStreamWaitEvent(s, tr->ev_join, 0));
    if (pending_hook >= 0) JCHK(fire_hook(pending_hook));
  }
  // patch embed: Linear(Kp -> bott) - GELU - Linear(bott -> D)   (jat_audiosr_v3.py:221-225); no gradient to the input
  KCHK(launch_cast_bf16(tr->dx, tr->dy, (int64_t)M * D, s));
  JCHK(input_grad(tr, tr->dy, D, tr->pe_w2T, m->bott, tr->dh, s));
  JCHK(weight_grad(tr, tr->dy, D, tr->pe_h, m->bott, G + tr->o_pe_w2, G + tr->o_pe_b2, s));
  KCHK(launch_gelu_bwd(tr->pe_pre, tr->dh, (int64_t)M * m->bott, kNoDrop, s));
  JCHK(weight_grad(tr, tr->dh, m->bott, tr->a_patch, m->Kp, G + tr->o_pe_w1, G + tr->o_pe_b1, s));
  // the t_embedder MLP (:364-369) behind all adaLN Linears; fp32, B rows
  // d silu(t_emb) = dmod [B, depth*6D] @ W_ada (the packed bf16 copy the forward multiplied with), all layers at once
  KCHK(launch_small_dx(tr->dmod, mstride, m->wada, 1, tr->small_part, tr->dt_emb, B, (int)mstride, D, 0, tr->t_emb, s));
  KCHK(launch_small_dw(tr->dt_emb, D, tr->t_h, D, G + tr->o_te_w2, G + tr->o_te_b2, B, D, D, 0, s));
  KCHK(launch_small_dx(tr->dt_emb, D, tr->P + tr->o_te_w2, 0, tr->small_part, tr->du1, B, D, D, 0, tr->u1, s));
  KCHK(launch_small_dw(tr->du1, D, tr->e_sin, D, G + tr->o_te_w1, G + tr->o_te_b1, B, D, D, 0, s));
  if (tr->hook) tr->hook(0, tr->block_lo[0], tr->hook_user);
  return JAT_OK;
}

}  // namespace

extern "C" void jat_trainer_destroy(jat_trainer* tr) {
  if (!tr) return;
  if (tr->dw_stream) { (void)hipStreamSynchronize(tr->dw_stream); (void)hipStreamDestroy(tr->dw_stream); }
  for (int k = 0; k < 4; ++k)
    for (int q = 0; q < 2; ++q) {
      if (tr->ev_ready[k][q]) (void)hipEventDestroy(tr->ev_ready[k][q]);
      if (tr->ev_done[k][q]) (void)hipEventDestroy(tr->ev_done[k][q]);
    }
  if (tr->ev_join) (void)hipEventDestroy(tr->ev_join);
  if (tr->ev_wt) (void)hipEventDestroy(tr->ev_wt);
  for (int q = 0; q < 2; ++q) if (tr->ev_mod[q]) (void)hipEventDestroy(tr->ev_mod[q]);
  for (hipEvent_t e : tr->ev_layer) if (e) (void)hipEventDestroy(e);
  if (tr->blob) (void)hipFree(tr->blob);
  delete tr;
}

extern "C" int jat_trainer_create(jat_model* m, const jat_tensor_ref* params, int32_t n, float* params_flat,
                                  float* grads_flat, float* exp_avg, float* exp_avg_sq, int64_t total, int32_t B,
                                  int32_t T, void* stream, jat_trainer** out) {
  if (!m || !params || !params_flat || !grads_flat || !exp_avg || !exp_avg_sq || !out)
    return fail(JAT_E_INVALID, "null argument");
  if (B <= 0 || T <= 0) return fail(JAT_E_INVALID, "B and T must be positive");
  if (B > 32) return fail(JAT_E_INVALID, "per-rank batch %d > 32 is not supported by the adaLN backward", B);
  if (total <= 0 || total % 4 != 0) return fail(JAT_E_INVALID, "flat buffer length must be a positive multiple of 4");
  const int ntok = (T + 3) / 4;
  if (ntok > MAX_LEN) return fail(JAT_E_SEQLEN, "Sequence length %d exceeds max_len %d", ntok, MAX_LEN);
  hipStream_t s = (hipStream_t)stream;
  jat_trainer* tr = new jat_trainer();
  tr->m = m; tr->B = B; tr->T = T; tr->ntok = ntok; tr->M = B * ntok;
  tr->Mpad = (int)align_up((size_t)tr->M, 512);   // K of the dW GEMMs: divisible by 64 * (split-K factor <= 8)
  tr->npad = (int)align_up((size_t)ntok, 64);
  tr->P = params_flat; tr->G = grads_flat; tr->m1 = exp_avg; tr->m2 = exp_avg_sq; tr->total = total;
  tr->rms = m->cfg.norm_mode == JAT_NORM_RMS_W;
  tr->p_drop.assign(m->depth, 0.f);
  tr->p_path.assign(m->depth, 0.f);
  const int D = m->D, depth = m->depth, mlp = m->mlp, bott = m->bott, kvD = m->kvD, Nqkv = D + 2 * kvD;
  const int M = tr->M, Mpad = tr->Mpad;

  // ---- parameter table: every tensor must lie inside the flat buffer, 16-B aligned, with the expected size ----
  std::unordered_map<std::string, int64_t> off;
  tr->names.reserve(n);
  for (int i = 0; i < n; ++i) {
    const int64_t o = params[i].data - params_flat;
    if (o < 0 || o + params[i].numel > total || o % 4 != 0) {
      delete tr;
      return fail(JAT_E_INVALID, "parameter '%s' is not a 16-byte aligned slice of the flat buffer", params[i].name);
    }
    tr->names.emplace_back(params[i].name);
    off[tr->names.back()] = o;
  }
  tr->prefs.resize(n);
  for (int i = 0; i < n; ++i) tr->prefs[i] = jat_tensor_ref{tr->names[i].c_str(), params[i].data, params[i].numel};
  int rc = JAT_OK;
  size_t used = 0;
  auto need = [&](const std::string& name) -> int64_t {
    auto it = off.find(name);
    if (it == off.end()) { rc = fail(JAT_E_STATE, "missing parameter '%s'", name.c_str()); return 0; }
    ++used;
    return it->second;
  };
  tr->o_pe_w1 = need("patch_embed.proj.0.weight"); tr->o_pe_b1 = need("patch_embed.proj.0.bias");
  tr->o_pe_w2 = need("patch_embed.proj.2.weight"); tr->o_pe_b2 = need("patch_embed.proj.2.bias");
  tr->o_te_w1 = need("t_embedder.1.weight"); tr->o_te_b1 = need("t_embedder.1.bias");
  tr->o_te_w2 = need("t_embedder.3.weight"); tr->o_te_b2 = need("t_embedder.3.bias");
  tr->o_fn = tr->rms ? need("final_layer.0.weight") : 0;
  tr->o_wf = need("final_layer.1.weight"); tr->o_bf = need("final_layer.1.bias");
  tr->L.resize(depth);
  for (int l = 0; l < depth; ++l) {
    const std::string p = "blocks." + std::to_string(l) + ".";
    TLayer& L = tr->L[l];
    L.o_n1 = tr->rms ? need(p + "norm1.weight") : 0; L.o_n2 = tr->rms ? need(p + "norm2.weight") : 0;
    L.o_q = need(p + "attn.q_proj.weight"); L.o_k = need(p + "attn.k_proj.weight"); L.o_v = need(p + "attn.v_proj.weight");
    L.o_o = need(p + "attn.out_proj.weight");
    L.o_w1 = need(p + "mlp.0.weight"); L.o_b1 = need(p + "mlp.0.bias");
    L.o_w2 = need(p + "mlp.3.weight"); L.o_b2 = need(p + "mlp.3.bias");
    L.o_ada_w = need(p + "adaLN_modulation.1.weight"); L.o_ada_b = need(p + "adaLN_modulation.1.bias");
  }
  // contiguous parameter ranges per block for the gradient-ready hook: block l = [block_lo[l], block_lo[l+1])
  tr->block_lo.assign(depth + 1, 0);
  for (int l = 0; l < depth && rc == JAT_OK; ++l) {
    const TLayer& L = tr->L[l];
    int64_t lo = std::min(std::min(L.o_q, L.o_k), std::min(L.o_v, L.o_o));
    lo = std::min(lo, std::min(std::min(L.o_w1, L.o_b1), std::min(L.o_w2, L.o_b2)));
    lo = std::min(lo, std::min(L.o_ada_w, L.o_ada_b));
    if (tr->rms) lo = std::min(lo, std::min(L.o_n1, L.o_n2));
    tr->block_lo[l] = lo;
  }
  tr->block_lo[depth] = std::min(tr->o_wf, tr->o_bf);
  if (tr->rms) tr->block_lo[depth] = std::min(tr->block_lo[depth], tr->o_fn);
  for (int l = 0; l < depth && rc == JAT_OK; ++l)
    if (tr->block_lo[l] >= tr->block_lo[l + 1])
      rc = fail(JAT_E_INVALID, "parameters of block %d are not laid out before those of block %d / the final layer", l, l + 1);
  {
    const int64_t head_hi = std::max(std::max(tr->o_pe_w1, tr->o_pe_b1), std::max(std::max(tr->o_pe_w2, tr->o_pe_b2),
                            std::max(std::max(tr->o_te_w1, tr->o_te_b1), std::max(tr->o_te_w2, tr->o_te_b2))));
    if (rc == JAT_OK && head_hi >= tr->block_lo[0])
      rc = fail(JAT_E_INVALID, "patch_embed / t_embedder parameters must precede the blocks in the flat buffer");
  }
  if (rc == JAT_OK && used != (size_t)n)
    rc = fail(JAT_E_INVALID, "%d parameters given, %zu belong to this model: every trainable tensor must be known", n, used);
  if (rc != JAT_OK) { delete tr; return rc; }

  if (const char* e = getenv("JAT_TN_DW")) tr->tn_dw = atoi(e) != 0;
  tr->dw_async = true;   // measured: 61.2 -> 58.5 ms per step at T = 1378, 33.3 -> 31.5 ms at T = 512 (profiles/r03/train_dw_stream_ab.log)
  if (const char* e = getenv("JAT_DW_STREAM")) tr->dw_async = atoi(e) != 0;
  if (tr->dw_async) {
    // lowest priority: when both queues have a GEMM ready, the chain's (critical path) goes first
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    static const int prio_env = getenv("JAT_DW_STREAM_PRIO") ? atoi(getenv("JAT_DW_STREAM_PRIO")) : 1;
    bool ok = (prio_env ? hipStreamCreateWithPriority(&tr->dw_stream, hipStreamNonBlocking, prio_least)
                        : hipStreamCreateWithFlags(&tr->dw_stream, hipStreamNonBlocking)) == hipSuccess;
    auto mk = [&](hipEvent_t* e) { ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; };
    for (int k = 0; k < 4; ++k)
      for (int q = 0; q < 2; ++q) { mk(&tr->ev_ready[k][q]); mk(&tr->ev_done[k][q]); }
    mk(&tr->ev_join); mk(&tr->ev_wt); mk(&tr->ev_mod[0]); mk(&tr->ev_mod[1]);
    tr->ev_layer.assign((size_t)depth + 1, nullptr);
    for (auto& e : tr->ev_layer) mk(&e);
    if (!ok) { jat_trainer_destroy(tr); return fail(JAT_E_HIP, "stream / event creation for the weight-gradient stream failed"); }
  }
  // ---- one allocation: transposed weights, saved activations, backward scratch ----
  for (int pass = 0; pass < 2; ++pass) {
    size_t o = 0;
    char* base = tr->blob;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align_up(bytes, 256); return p; };
    const size_t MD2 = (size_t)M * D * 2, MD4 = (size_t)M * D * 4;
    tr->x.resize(depth + 1);
    for (int l = 0; l <= depth; ++l) tr->x[l] = (float*)take(MD4);
    for (int l = 0; l < depth; ++l) {
      TLayer& L = tr->L[l];
      L.x_mid = (float*)take(MD4);
      L.lse = (float*)take((size_t)B * m->Hq * ntok * 4);
      L.xn1 = (bf16_t*)take(MD2); L.q = (bf16_t*)take(MD2); L.k = (bf16_t*)take((size_t)M * kvD * 2);
      L.vt = (bf16_t*)take((size_t)B * m->Hkv * HEAD_DIM * tr->npad * 2);
      L.ao = (bf16_t*)take(MD2); L.y_attn = (bf16_t*)take(MD2); L.xn2 = (bf16_t*)take(MD2);
      L.h_pre = (bf16_t*)take((size_t)M * mlp * 2); L.h_post = (bf16_t*)take((size_t)M * mlp * 2);
      L.y_mlp = (bf16_t*)take(MD2);
      L.wqkvT = (bf16_t*)take((size_t)D * Nqkv * 2); L.woT = (bf16_t*)take((size_t)D * D * 2);
      L.w1T = (bf16_t*)take((size_t)D * mlp * 2); L.w2T = (bf16_t*)take((size_t)mlp * D * 2);
    }
    tr->a_patch = (bf16_t*)take((size_t)M * m->Kp * 2);
    tr->pe_pre = (bf16_t*)take((size_t)M * bott * 2); tr->pe_h = (bf16_t*)take((size_t)M * bott * 2);
    tr->xnf = (bf16_t*)take(MD2);
    tr->t_silu = (bf16_t*)take((size_t)B * D * 2);
    tr->pe_w2T = (bf16_t*)take((size_t)bott * D * 2); tr->wfinalT = (bf16_t*)take((size_t)D * m->Fout * 2);
    tr->e_sin = (float*)take((size_t)B * D * 4); tr->u1 = (float*)take((size_t)B * D * 4);
    tr->t_h = (float*)take((size_t)B * D * 4); tr->t_emb = (float*)take((size_t)B * D * 4);
    tr->mod = (float*)take((size_t)B * depth * 6 * D * 4);
    tr->pred = (float*)take((size_t)B * m->Cin * T * 4);
    tr->dx = (float*)take(MD4); tr->dpred = (float*)take((size_t)B * m->Cin * T * 4);
    tr->dmod = (float*)take((size_t)B * depth * 6 * D * 4);
    tr->part = (float*)take((size_t)B * train_nchunk(ntok) * 3 * D * 4);
    tr->red_part = (float*)take((size_t)train_red_blocks() * 4);
    tr->scal = (float*)take(64);
    tr->terms = (float*)take(64);
    tr->tw = (float2*)take((size_t)T * sizeof(float2));
    tr->ll_part = (float*)take((size_t)B * m->Cin * 8 * 4);
    tr->delta = (float*)take((size_t)B * m->Hq * ntok * 4);
    tr->dkv_part = (float*)take((size_t)B * m->Hq * ntok * 128 * 4);
    tr->dwqkv = (float*)take((size_t)Nqkv * D * 4);
    tr->dt_emb = (float*)take((size_t)B * D * 4);
    tr->du1 = (float*)take((size_t)B * D * 4);
    {
      const int Nall = depth * 6 * D;
      const size_t slabs = std::max((Nall + small_dx_slab(Nall) - 1) / small_dx_slab(Nall), (D + small_dx_slab(D) - 1) / small_dx_slab(D));
      tr->small_part = (float*)take(slabs * B * D * 4);
    }
    tr->dw_part = (float*)take((size_t)B * D * 4);
    tr->split4_area = (int64_t)Nqkv * D;            // q/k/v, out_proj, patch-embed proj.2: 4 slices
    tr->split2_area = (int64_t)m->Fout * D;         // final Linear: 2 slices; the MLP weights fill the chip unsplit
    if (tr->split2_area < tr->split4_area) tr->split2_area = tr->split4_area;
    {
      size_t need = (size_t)std::max(4 * tr->split4_area, 2 * tr->split2_area);
      const int shapes[][2] = {{m->Fout, D}, {D, mlp}, {mlp, D}, {D, D}, {Nqkv, D}, {D, m->bott}, {m->bott, m->Kp}};
      for (auto& sh : shapes)
        if (gemm_tn_supports(sh[0], sh[1])) need = std::max(need, (size_t)gemm_tn_ksplit(sh[0], sh[1], M) * sh[0] * sh[1]);
      tr->dw_split = (float*)take(need * 4);
    }
    tr->zero_cell = take(256);
    tr->copy_jobs = (CopyJob*)take((size_t)(8 + 5 * depth + 2) * sizeof(CopyJob));
    tr->dy = (bf16_t*)take(MD2); tr->dh = (bf16_t*)take((size_t)M * std::max(mlp, bott) * 2);
    tr->dxn = (bf16_t*)take(MD2); tr->dao = (bf16_t*)take(MD2); tr->dqkv = (bf16_t*)take((size_t)M * Nqkv * 2);
    tr->dy_m[0] = tr->dy_a[0] = tr->dy; tr->dy_m[1] = tr->dy_a[1] = tr->dy;   // one stream: every operand has one home
    tr->dh_b[0] = tr->dh_b[1] = tr->dh; tr->dq_b[0] = tr->dq_b[1] = tr->dqkv;
    if (tr->dw_async) {   // second home per operand (layer parity) + a separate one for the attention branch's dy: + ~0.4 GB at T = 1378
      tr->dy_m[1] = (bf16_t*)take(MD2); tr->dy_a[0] = (bf16_t*)take(MD2); tr->dy_a[1] = (bf16_t*)take(MD2);
      tr->dh_b[1] = (bf16_t*)take((size_t)M * std::max(mlp, bott) * 2);
      tr->dq_b[1] = (bf16_t*)take((size_t)M * Nqkv * 2);
    }
    tr->dyf = (bf16_t*)take((size_t)M * m->Fout * 2);
    const int rowsA = std::max(std::max(m->Fout, Nqkv), std::max(mlp, std::max(D, bott)));
    const int rowsB = std::max(std::max(m->Kp, mlp), std::max(D, bott));
    {   // transposed activation copies: only for the weights gemm_tn.hip does not take (widths that are not multiples of 128)
      const int shapes[][2] = {{m->Fout, D}, {D, mlp}, {mlp, D}, {D, D}, {Nqkv, D}, {D, bott}, {bott, m->Kp}};
      bool need = !tr->tn_dw;
      for (auto& sh : shapes) need = need || !gemm_tn_supports(sh[0], sh[1]);
      tr->tA = need ? (bf16_t*)take((size_t)rowsA * Mpad * 2) : nullptr;
      tr->tB = need ? (bf16_t*)take((size_t)rowsB * Mpad * 2) : nullptr;
    }
    tr->colsum_part = (float*)take((size_t)colsum_slices(M) * rowsA * 4);
    if (pass == 0) {
      tr->blob_bytes = o;
      if (hipMalloc((void**)&tr->blob, o) != hipSuccess) {
        delete tr;
        return fail(JAT_E_HIP, "hipMalloc of %zu bytes for the training workspace failed", o);
      }
      // zero once: the key padding of every V^T buffer must be 0 and is never written afterwards
      if (hipMemsetAsync(tr->blob, 0, o, s) != hipSuccess) { jat_trainer_destroy(tr); return fail(JAT_E_HIP, "memset failed"); }
    }
  }
  {
    std::vector<CopyJob> jobs;
    auto job = [&](int64_t o, float* dst, int64_t cnt) { jobs.push_back(CopyJob{params_flat + o, dst, cnt}); };
    job(tr->o_pe_b1, m->pe_b1, bott); job(tr->o_pe_b2, m->pe_b2, D);
    job(tr->o_te_w1, m->te_w1, (int64_t)D * D); job(tr->o_te_b1, m->te_b1, D);
    job(tr->o_te_w2, m->te_w2, (int64_t)D * D); job(tr->o_te_b2, m->te_b2, D);
    job(tr->o_bf, m->bfinal, m->Fout);
    if (tr->rms) job(tr->o_fn, m->final_norm, D);
    for (int l = 0; l < depth; ++l) {
      const TLayer& L = tr->L[l];
      if (tr->rms) { job(L.o_n1, m->layers[l].norm1, D); job(L.o_n2, m->layers[l].norm2, D); }
      job(L.o_b1, m->layers[l].b1, mlp); job(L.o_b2, m->layers[l].b2, D);
      job(L.o_ada_b, m->bada + (int64_t)l * 6 * D, (int64_t)6 * D);
    }
    tr->n_copy_jobs = (int)jobs.size();
    if (hipMemcpyAsync(tr->copy_jobs, jobs.data(), jobs.size() * sizeof(CopyJob), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      jat_trainer_destroy(tr);
      return fail(JAT_E_HIP, "copy-table upload failed");
    }
  }
  {
    std::vector<float2> tw(T);
    for (int i = 0; i < T; ++i) {
      const double th = 2.0 * M_PI * (double)i / (double)T;
      tw[i] = float2{(float)cos(th), (float)sin(th)};
    }
    if (hipMemcpyAsync(tr->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      jat_trainer_destroy(tr);
      return fail(JAT_E_HIP, "twiddle upload failed");
    }
  }
  rc = build_transposes(tr, s);   // the model's own copies were packed from these very tensors by jat_model_load_weights
  if (rc != JAT_OK) { jat_trainer_destroy(tr); return rc; }
  if (hipStreamSynchronize(s) != hipSuccess) { jat_trainer_destroy(tr); return fail(JAT_E_HIP, "trainer setup failed"); }
  *out = tr;
  return JAT_OK;
}

extern "C" int jat_trainer_set_grad_hook(jat_trainer* tr, void (*hook)(int64_t, int64_t, void*), void* user) {
  if (!tr) return fail(JAT_E_INVALID, "null argument");
  tr->hook = hook;
  tr->hook_user = user;
  return JAT_OK;
}

extern "C" int jat_trainer_set_regularisers(jat_trainer* tr, const float* dropout, const float* drop_path) {
  if (!tr || !dropout || !drop_path) return fail(JAT_E_INVALID, "null argument");
  for (int l = 0; l < tr->m->depth; ++l) {
    if (!(dropout[l] >= 0.f && dropout[l] < 1.f) || !(drop_path[l] >= 0.f && drop_path[l] < 1.f))
      return fail(JAT_E_INVALID, "layer %d: rates must be in [0, 1)", l);
    tr->p_drop[l] = dropout[l];
    tr->p_path[l] = drop_path[l];
  }
  return JAT_OK;
}

extern "C" int jat_trainer_set_latent_loss(jat_trainer* tr, double latent_weight, double freq_weight, double ms_weight,
                                           double consistency_weight, double low_freq_phase_ratio, double strict_cutoff,
                                           double soft_cutoff) {
  if (!tr) return fail(JAT_E_INVALID, "null argument");
  if (!(low_freq_phase_ratio >= 0 && low_freq_phase_ratio <= 1 && strict_cutoff >= 0 && soft_cutoff >= strict_cutoff &&
        soft_cutoff <= 1))
    return fail(JAT_E_INVALID, "band ratios must satisfy 0 <= strict <= soft <= 1 and 0 <= phase ratio <= 1");
  if (latent_weight != 0.0 && tr->charb_eps > 0.0)
    return fail(JAT_E_STATE, "the latent perceptual loss is defined on top of the MSE loss only (train_ddp_v3mod2.py:889-896)");
  tr->lw = latent_weight; tr->fw = freq_weight; tr->mw = ms_weight; tr->cw = consistency_weight;
  tr->phase_ratio = low_freq_phase_ratio; tr->strict_cut = strict_cutoff; tr->soft_cut = soft_cutoff;
  return JAT_OK;
}

extern "C" int jat_trainer_set_charbonnier(jat_trainer* tr, double eps) {
  if (!tr) return fail(JAT_E_INVALID, "null argument");
  if (!(eps >= 0.0)) return fail(JAT_E_INVALID, "eps must be >= 0 (0 selects the MSE loss)");
  if (eps > 0.0 && tr->lw != 0.0)
    return fail(JAT_E_STATE, "the latent perceptual loss is defined on top of the MSE loss only (train_ddp_v3mod2.py:889-896)");
  tr->charb_eps = eps;
  return JAT_OK;
}

extern "C" int jat_trainer_loss_terms(jat_trainer* tr, float* out6, void* stream) {
  if (!tr || !out6) return fail(JAT_E_INVALID, "null argument");
  if (tr->lw == 0.0) return fail(JAT_E_STATE, "the latent perceptual loss is off (jat_trainer_set_latent_loss)");
  HIPCHK(hipMemcpyAsync(out6, tr->terms, 6 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JAT_OK;
}

extern "C" int jat_trainer_workspace_bytes(const jat_trainer* tr, size_t* out) {
  if (!tr || !out) return fail(JAT_E_INVALID, "null argument");
  *out = tr->blob_bytes;
  return JAT_OK;
}

extern "C" int jat_trainer_prepare(jat_trainer* tr, const float* hr_norm, float* cond, const float* noise,
                                   const float* cond_noise, float cond_noise_ratio, int32_t adaptive, const float* keep,
                                   const float* t, float* z_t, void* stream) {
  if (!tr || !hr_norm || !cond || !noise || !t || !z_t) return fail(JAT_E_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t per = (int64_t)tr->m->Cin * tr->T;
  if (cond_noise && cond_noise_ratio > 0.f && adaptive)   // lr_norm.std().clamp(0.5, 2.0)  (train_ddp_v3m2.py:553-556)
    KCHK(launch_tensor_std(cond, (int64_t)tr->B * tr->m->Cc * tr->T, tr->red_part, tr->scal + 8, tr->scal + 4, s));
  if ((cond_noise && cond_noise_ratio > 0.f) || keep)
    KCHK(launch_cond_augment(cond, cond_noise_ratio > 0.f ? cond_noise : nullptr, adaptive ? tr->scal + 4 : nullptr,
                             cond_noise_ratio, keep, tr->B, (int64_t)tr->m->Cc * tr->T, s));
  KCHK(launch_flow_mix(hr_norm, noise, t, z_t, tr->B, per, s));
  return JAT_OK;
}

extern "C" int jat_trainer_fwd_bwd(jat_trainer* tr, const float* z_t, const float* t, const float* x_cond,
                                   const float* target, const float* cond_clean, float loss_scale, uint64_t rng_seed,
                                   float* loss_out, float* x_pred_out, void* stream) {
  if (!tr || !z_t || !t || !x_cond || !target) return fail(JAT_E_INVALID, "null argument");
  if (tr->lw != 0.0 && tr->cw != 0.0 && !cond_clean)
    return fail(JAT_E_INVALID, "the consistency loss needs the clean condition latent (cond_clean)");
  if (!tr->m->loaded) return fail(JAT_E_STATE, "weights not loaded");
  hipStream_t s = (hipStream_t)stream;
  tr->seed = rng_seed;
  JCHK(forward_train(tr, z_t, t, x_cond, s));
  JCHK(backward_train(tr, target, cond_clean, loss_scale, s));
  if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, tr->scal, 4, hipMemcpyDeviceToDevice, s));
  if (x_pred_out)
    HIPCHK(hipMemcpyAsync(x_pred_out, tr->pred, (size_t)tr->B * tr->m->Cin * tr->T * 4, hipMemcpyDeviceToDevice, s));
  return JAT_OK;
}

extern "C" int jat_trainer_optim(jat_trainer* tr, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 float max_grad_norm, float loss_scale, int32_t step, float* grad_norm_out, void* stream) {
  if (!tr) return fail(JAT_E_INVALID, "null argument");
  if (step < 1 || loss_scale <= 0.f) return fail(JAT_E_INVALID, "step must be >= 1 and loss_scale > 0");
  hipStream_t s = (hipStream_t)stream;
  KCHK(launch_grad_sqsum(tr->G, tr->total, tr->red_part, tr->scal + 2, s));
  KCHK(launch_adamw(tr->P, tr->G, tr->m1, tr->m2, tr->total, tr->scal + 2, 1.0f / loss_scale, max_grad_norm, lr, beta1, beta2,
                    eps, weight_decay, step, s));
  if (grad_norm_out) HIPCHK(hipMemcpyAsync(grad_norm_out, tr->scal + 3, 4, hipMemcpyDeviceToDevice, s));
  return repack(tr, s);
}

// per-kernel entry point (unit parity, validation): reconstruction loss and d(loss * loss_scale)/d pred on n elements;
// eps > 0: Charbonnier (train_ddp_v3m2mod1.py:72-101), eps == 0: MSE (train_ddp_v3m2.py:585); work: 1024 floats
extern "C" int jat_k_recon_loss(const float* pred, const float* target, float* dpred, float* loss_out, int64_t n, double eps,
                                float loss_scale, void* work, size_t work_bytes, void* stream) {
  if (!pred || !target || !dpred || !loss_out || !work || n <= 0 || !(eps >= 0.0)) return fail(JAT_E_INVALID, "bad argument");
  const size_t need = ((size_t)train_red_blocks() + 2) * 4;
  if (work_bytes < need) return fail(JAT_E_STATE, "work buffer too small: %zu < %zu bytes", work_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)work;
  float* loss2 = part + train_red_blocks();
  if (eps > 0.0) KCHK(launch_charbonnier_grad(pred, target, dpred, part, loss2, n, (float)eps, loss_scale, s));
  else KCHK(launch_mse_grad(pred, target, dpred, part, loss2, n, loss_scale, s));
  HIPCHK(hipMemcpyAsync(loss_out, loss2, 4, hipMemcpyDeviceToDevice, s));
  return JAT_OK;
}

// per-kernel entry point (unit parity): the v3mod2 loss on [rows, T] tensors; `work` holds T*8 + rows*32 bytes
extern "C" int jat_k_latent_loss(const float* pred, const float* target, const float* lr, float* dpred, float* out6,
                                 int32_t rows, int32_t T, double latent_weight, double freq_weight, double ms_weight,
                                 double consistency_weight, double low_freq_phase_ratio, double strict_cutoff,
                                 double soft_cutoff, float loss_scale, void* work, size_t work_bytes, void* stream) {
  if (!pred || !target || !dpred || !out6 || !work || rows <= 0 || T <= 0) return fail(JAT_E_INVALID, "bad argument");
  const size_t need = align_up((size_t)T * sizeof(float2), 256) + (size_t)rows * 8 * 4;
  if (work_bytes < need) return fail(JAT_E_STATE, "work buffer too small: %zu < %zu bytes", work_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  float2* tw = (float2*)work;
  float* part = (float*)((char*)work + align_up((size_t)T * sizeof(float2), 256));
  std::vector<float2> h(T);
  for (int i = 0; i < T; ++i) {
    const double th = 2.0 * M_PI * (double)i / (double)T;
    h[i] = float2{(float)cos(th), (float)sin(th)};
  }
  HIPCHK(hipMemcpyAsync(tw, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  const int F = T / 2 + 1;
  KCHK(launch_latent_loss(pred, target, lr, tw, dpred, part, out6, rows, T, (float)latent_weight, (float)freq_weight,
                          (float)ms_weight, (float)consistency_weight, (int)((double)F * low_freq_phase_ratio),
                          (int)((double)F * strict_cutoff), (int)((double)F * soft_cutoff), loss_scale, s));
  return JAT_OK;
}

// Re-derive every operand copy (bf16 weights, transposed copies, fp32 operand tensors) from the flat master buffer,
// e.g. after the caller overwrote parameters (checkpoint resume, train_ddp_v3m2.py:443-500).
extern "C" int jat_trainer_repack(jat_trainer* tr, void* stream) {
  if (!tr) return fail(JAT_E_INVALID, "null argument");
  return repack(tr, (hipStream_t)stream);
}
