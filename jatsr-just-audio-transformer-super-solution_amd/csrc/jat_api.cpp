// C-ABI host side of libjat_hip.so (declarations and contracts: include/jat_hip.h).
// Owns the packed weights, lays out the caller's workspace, sequences the gfx950 kernels of one DiT forward
// on the caller's stream, and captures the 50-step CFG sampler into a hipGraph.  No arithmetic happens here.
#include "../../include/jat_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "jat_internal.h"
#include "jat_dtype.h"

static thread_local char g_err[512] = "";
int jat_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// workspace carve-up for a forward over `B` batch rows of `ntok` tokens
struct Workspace {
  bf16_t *a_patch, *h_patch, *xn, *xlo, *q, *k, *vt, *ao, *hm, *t_silu;
  float *x, *mod, *e_sin, *t_h, *t_emb, *part;
  float* kpart;   // split-K partials of the fc2 GEMM when M is too small to fill the chip (nullptr for large M)
  const int* lens = nullptr;   // sampler: per-batch-row key counts (tokens) for the attention kernel, or nullptr
  const int* tvalid = nullptr; // sampler: valid frames per source row (patchify reads zeros beyond them), or nullptr
  int npad;
  size_t vt_bytes, total;
};

// Small-M inference (the reference's own B = 1 chunk loop gives M = 690 rows with CFG; a file's short last chunk M = 240):
// the K = 5120 fc2 GEMM has a few dozen tiles x 80 K-steps — split K over otherwise idle CUs, finish in fixed order.
static constexpr int kSplitMaxRows = 2304, kSplitMax = 8;
static constexpr int kSplitWsRows = 4096;   // split-K partial workspace exists up to here (un-folded buckets: see resid_split)

static Workspace carve(const jat_model* m, int B, int ntok, char* base) {
  Workspace w;
  size_t off = 0;
  const size_t M = (size_t)B * ntok;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  w.npad = (int)align_up((size_t)ntok, 64);
  w.a_patch = (bf16_t*)take(M * m->Kp * 2);
  w.h_patch = (bf16_t*)take(M * m->bott * 2);
  w.x = (float*)take(M * m->D * 4);
  w.xn = (bf16_t*)take(M * m->D * 2);
  w.xlo = (bf16_t*)take(M * m->D * 2);   // lo plane of the split residual stream (sampler with folded norms; hi = xn)
  w.q = (bf16_t*)take(M * m->D * 2);
  w.k = (bf16_t*)take(M * m->kvD * 2);
  w.vt_bytes = (size_t)B * m->Hkv * HEAD_DIM * w.npad * 2;
  w.vt = (bf16_t*)take(w.vt_bytes);
  w.ao = (bf16_t*)take(M * m->D * 2);
  w.hm = (bf16_t*)take(M * m->mlp * 2);
  w.mod = (float*)take((size_t)B * m->depth * 6 * m->D * 4);
  w.e_sin = (float*)take((size_t)B * m->D * 4);
  w.t_h = (float*)take((size_t)B * m->D * 4);
  w.t_emb = (float*)take((size_t)B * m->D * 4);
  w.t_silu = (bf16_t*)take((size_t)B * m->D * 2);
  w.part = (float*)take(M * 32 * 4);  // row partial sums of x^2 (norm folding), <= 32 wave column tiles
  w.kpart = M <= kSplitWsRows ? (float*)take((size_t)(M <= kSplitMaxRows ? kSplitMax : 2) * M * (m->D + 2 * m->kvD) * 4) : nullptr;   // widest user: the QKV GEMM
  w.total = off;
  return w;
}

extern "C" const char* jat_last_error(void) { return g_err; }
extern "C" int jat_version(void) { return 2; }
extern "C" int jat_operand_dtype(void) { return JAT_OPERAND_DTYPE; }   // 0 bf16, 1 fp16 (jat_dtype.h)

// ---------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------
extern "C" int jat_model_create(const jat_config* c, jat_model** out) {
  if (!c || !out) return fail(JAT_E_INVALID, "null argument");
  // same checks as the reference constructor asserts (jat_audiosr_v3.py:119-120) plus kernel limits
  if (c->num_q_heads <= 0 || c->hidden_size % c->num_q_heads != 0)
    return fail(JAT_E_INVALID, "hidden_size must be divisible by num_q_heads");
  if (c->num_kv_heads <= 0 || c->num_q_heads % c->num_kv_heads != 0)
    return fail(JAT_E_INVALID, "num_q_heads must be divisible by num_kv_heads");
  if (c->hidden_size / c->num_q_heads != HEAD_DIM) return fail(JAT_E_INVALID, "head_dim must be 64");
  if (c->patch_len != 4) return fail(JAT_E_INVALID, "patch_len must be 4");
  if (c->hidden_size % 256 != 0 || c->hidden_size > 2048)
    return fail(JAT_E_INVALID, "hidden_size must be a multiple of 256, <= 2048");
  if (c->bottleneck_dim % 128 != 0 || c->mlp_hidden % 128 != 0)
    return fail(JAT_E_INVALID, "bottleneck_dim and mlp_hidden must be multiples of 128");
  if (c->input_channels % 32 != 0 || c->cond_channels % 32 != 0 || c->input_channels <= 0 || c->cond_channels <= 0)
    return fail(JAT_E_INVALID, "channel counts must be positive multiples of 32");
  if (c->depth <= 0) return fail(JAT_E_INVALID, "depth must be positive");
  if (c->norm_mode != JAT_NORM_RMS_W && c->norm_mode != JAT_NORM_LN_NOAFFINE)
    return fail(JAT_E_INVALID, "unknown norm_mode");
  jat_model* m = new jat_model();
  m->cfg = *c;
  m->D = c->hidden_size; m->depth = c->depth; m->Hq = c->num_q_heads; m->Hkv = c->num_kv_heads;
  m->kvD = m->Hkv * HEAD_DIM; m->mlp = c->mlp_hidden; m->bott = c->bottleneck_dim;
  m->Cin = c->input_channels; m->Cc = c->cond_channels; m->P = 4;
  m->Kp = m->P * (m->Cin + m->Cc); m->Fout = m->P * m->Cin;
  auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
  m->sw.fuse_qkv_attn = env_int("JAT_FUSE_QKV_ATTN", 1); m->sw.qkv_split = env_int("JAT_QKV_SPLIT", 1);
  m->sw.fuse_finish = env_int("JAT_FUSE_FINISH", 1); m->sw.fold_norm = env_int("JAT_FOLD_NORM", 1);
  m->sw.split_patch = env_int("JAT_SPLIT_PATCH", 1); m->sw.gemm_dbg = env_int("JAT_GEMM_DBG", 0);
  m->sw.fold_cap_mb = env_int("JAT_FOLD_CAP_MB", 0); m->sw.patch_split = env_int("JAT_PATCH_SPLIT", 1);
  if (const char* v = getenv("JAT_GEMM_VARIANT"))
    for (int i = 0; i < 5; ++i) m->variants[i] = atoi(v);
  if (const char* v = getenv("JAT_GEMM_VARIANTS")) {  // "qkv,out,fc1,fc2,other"
    int x[5];
    if (sscanf(v, "%d,%d,%d,%d,%d", &x[0], &x[1], &x[2], &x[3], &x[4]) == 5)
      for (int i = 0; i < 5; ++i) m->variants[i] = x[i];
  }
  *out = m;
  return JAT_OK;
}

extern "C" int jat_model_set_switch(jat_model* m, const char* name, int32_t value) {
  if (!m || !name) return fail(JAT_E_INVALID, "null argument");
  const std::string n(name);
  int* slot = n == "fuse_qkv_attn" ? &m->sw.fuse_qkv_attn : n == "qkv_split" ? &m->sw.qkv_split : n == "fuse_finish" ? &m->sw.fuse_finish
            : n == "fold_norm" ? &m->sw.fold_norm : n == "split_patch" ? &m->sw.split_patch : n == "gemm_dbg" ? &m->sw.gemm_dbg
            : n == "fold_cap_mb" ? &m->sw.fold_cap_mb : n == "patch_split" ? &m->sw.patch_split : nullptr;
  if (!slot) return fail(JAT_E_INVALID, "unknown switch '%s'", name);
  *slot = value;
  return JAT_OK;
}

extern "C" void jat_model_destroy(jat_model* m) {
  if (!m) return;
  for (hipEvent_t e : m->prof.ev) (void)hipEventDestroy(e);
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

extern "C" int jat_model_load_weights(jat_model* m, const jat_tensor_ref* named, int32_t n, void* stream_) {
  if (!m || !named) return fail(JAT_E_INVALID, "null argument");
  return jat_pack_weights(m, named, n, (hipStream_t)stream_, true);
}

// build_tables == false: re-pack after an optimiser step (jat_train.cpp) — conversions only, no host synchronisation
int jat_pack_weights(jat_model* m, const jat_tensor_ref* named, int32_t n, hipStream_t s, bool build_tables) {
  std::unordered_map<std::string, const jat_tensor_ref*> by_name;
  for (int i = 0; i < n; ++i) by_name[named[i].name] = &named[i];
  const int D = m->D, kvD = m->kvD, mlp = m->mlp, bott = m->bott, depth = m->depth;
  const bool rms = m->cfg.norm_mode == JAT_NORM_RMS_W;

  // ---- layout of the packed blob ----
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t o_pe_w1 = take((size_t)bott * m->Kp * 2), o_pe_w2 = take((size_t)D * bott * 2);
  const size_t o_wada = take((size_t)depth * 6 * D * D * 2), o_wfinal = take((size_t)m->Fout * D * 2);
  std::vector<size_t> o_qkv(depth), o_wo(depth), o_w1(depth), o_w2(depth), o_n1(depth), o_n2(depth), o_b1(depth),
      o_b2(depth), o_qkvg(depth), o_q32(depth), o_k32(depth), o_v32(depth), o_w132(depth);
  const bool keep32 = rms;   // fold sources (RMSNorm models only)
  const bool group5 = m->Hq / m->Hkv == 5;
  for (int l = 0; l < depth; ++l) {
    o_qkv[l] = take((size_t)(D + 2 * kvD) * D * 2); o_wo[l] = take((size_t)D * D * 2);
    o_qkvg[l] = group5 ? take((size_t)(D + 2 * kvD) * D * 2) : 0;
    o_w1[l] = take((size_t)mlp * D * 2); o_w2[l] = take((size_t)D * mlp * 2);
    o_n1[l] = take((size_t)D * 4); o_n2[l] = take((size_t)D * 4);
    o_b1[l] = take((size_t)mlp * 4); o_b2[l] = take((size_t)D * 4);
    if (keep32) {
      o_q32[l] = take((size_t)D * D * 4); o_k32[l] = take((size_t)kvD * D * 4); o_v32[l] = take((size_t)kvD * D * 4);
      o_w132[l] = take((size_t)mlp * D * 4);
    }
  }
  const size_t o_wfinal32 = keep32 ? take((size_t)m->Fout * D * 4) : 0;
  const size_t o_pe_b1 = take((size_t)bott * 4), o_pe_b2 = take((size_t)D * 4);
  const size_t o_te_w1 = take((size_t)D * D * 4), o_te_b1 = take((size_t)D * 4);
  const size_t o_te_w2 = take((size_t)D * D * 4), o_te_b2 = take((size_t)D * 4);
  const size_t o_bada = take((size_t)depth * 6 * D * 4), o_fn = take((size_t)D * 4), o_bfinal = take((size_t)m->Fout * 4);
  const size_t o_cos = take((size_t)MAX_LEN * 32 * 4), o_sin = take((size_t)MAX_LEN * 32 * 4), o_invf = take(32 * 4);
  if (!m->blob) {
    HIPCHK(hipMalloc((void**)&m->blob, off));
    m->blob_bytes = off;
  }
  char* base = m->blob;
  m->pe_w1 = (bf16_t*)(base + o_pe_w1); m->pe_w2 = (bf16_t*)(base + o_pe_w2);
  m->wada = (bf16_t*)(base + o_wada); m->wfinal = (bf16_t*)(base + o_wfinal);
  m->pe_b1 = (float*)(base + o_pe_b1); m->pe_b2 = (float*)(base + o_pe_b2);
  m->te_w1 = (float*)(base + o_te_w1); m->te_b1 = (float*)(base + o_te_b1);
  m->te_w2 = (float*)(base + o_te_w2); m->te_b2 = (float*)(base + o_te_b2);
  m->bada = (float*)(base + o_bada); m->final_norm = (float*)(base + o_fn); m->bfinal = (float*)(base + o_bfinal);
  m->rope_cos = (float*)(base + o_cos); m->rope_sin = (float*)(base + o_sin); m->rope_invf = (float*)(base + o_invf);
  m->layers.resize(depth);
  for (int l = 0; l < depth; ++l) {
    LayerW& L = m->layers[l];
    L.wqkv = (bf16_t*)(base + o_qkv[l]); L.wo = (bf16_t*)(base + o_wo[l]);
    L.wqkv_g = group5 ? (bf16_t*)(base + o_qkvg[l]) : nullptr;
    L.w1 = (bf16_t*)(base + o_w1[l]); L.w2 = (bf16_t*)(base + o_w2[l]);
    L.norm1 = (float*)(base + o_n1[l]); L.norm2 = (float*)(base + o_n2[l]);
    L.b1 = (float*)(base + o_b1[l]); L.b2 = (float*)(base + o_b2[l]);
    if (keep32) {
      L.q32 = (float*)(base + o_q32[l]); L.k32 = (float*)(base + o_k32[l]); L.v32 = (float*)(base + o_v32[l]);
      L.w132 = (float*)(base + o_w132[l]);
    }
  }
  m->wfinal32 = keep32 ? (float*)(base + o_wfinal32) : nullptr;
  m->fold_cache.clear();   // folded tables belong to the previous weights (samplers that still hold one keep it alive)
  m->fold_src_ok = false;

  // ---- copy / convert ----
  int rc = JAT_OK;
  auto find = [&](const std::string& name, int64_t numel) -> const float* {
    auto it = by_name.find(name);
    if (it == by_name.end()) { rc = fail(JAT_E_STATE, "missing parameter '%s'", name.c_str()); return nullptr; }
    if (it->second->numel != numel) {
      rc = fail(JAT_E_INVALID, "parameter '%s' has %lld elements, expected %lld", name.c_str(),
                (long long)it->second->numel, (long long)numel);
      return nullptr;
    }
    return it->second->data;
  };
  auto to_bf16 = [&](const std::string& name, bf16_t* dst, int64_t numel) {
    const float* src = find(name, numel);
    if (!src) return;
    if (launch_cast_bf16(src, dst, numel, s) != hipSuccess) rc = fail(JAT_E_HIP, "cast kernel launch failed");
  };
  auto to_bf16_rope = [&](const std::string& name, bf16_t* dst, int rows, int cols) {
    const float* src = find(name, (int64_t)rows * cols);
    if (!src) return;
    if (launch_cast_bf16_rope_rows(src, dst, rows, cols, s) != hipSuccess) rc = fail(JAT_E_HIP, "cast kernel launch failed");
  };
  auto to_f32 = [&](const std::string& name, float* dst, int64_t numel) {
    const float* src = find(name, numel);
    if (!src) return;
    if (hipMemcpyAsync(dst, src, numel * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
      rc = fail(JAT_E_HIP, "memcpy failed for '%s'", name.c_str());
  };
  auto ones = [&](float* dst, int64_t numel) {
    if (!build_tables) return;  // constants: written once at load time
    std::vector<float> h(numel, 1.0f);
    if (hipMemcpyAsync(dst, h.data(), numel * 4, hipMemcpyHostToDevice, s) != hipSuccess) rc = fail(JAT_E_HIP, "memcpy");
    (void)hipStreamSynchronize(s);
  };
  to_bf16("patch_embed.proj.0.weight", m->pe_w1, (int64_t)bott * m->Kp);
  to_f32("patch_embed.proj.0.bias", m->pe_b1, bott);
  to_bf16("patch_embed.proj.2.weight", m->pe_w2, (int64_t)D * bott);
  to_f32("patch_embed.proj.2.bias", m->pe_b2, D);
  to_f32("t_embedder.1.weight", m->te_w1, (int64_t)D * D);
  to_f32("t_embedder.1.bias", m->te_b1, D);
  to_f32("t_embedder.3.weight", m->te_w2, (int64_t)D * D);
  to_f32("t_embedder.3.bias", m->te_b2, D);
  for (int l = 0; l < depth && rc == JAT_OK; ++l) {
    const std::string p = "blocks." + std::to_string(l) + ".";
    LayerW& L = m->layers[l];
    if (rms) { to_f32(p + "norm1.weight", L.norm1, D); to_f32(p + "norm2.weight", L.norm2, D); }
    else { ones(L.norm1, D); ones(L.norm2, D); }
    // fused [Wq; Wk; Wv]; q/k rows pair-interleaved per head so that RoPE pairs share a lane (gemm.hip)
    to_bf16_rope(p + "attn.q_proj.weight", L.wqkv, D, D);
    to_bf16_rope(p + "attn.k_proj.weight", L.wqkv + (int64_t)D * D, kvD, D);
    to_bf16(p + "attn.v_proj.weight", L.wqkv + (int64_t)(D + kvD) * D, (int64_t)kvD * D);
    if (group5 && rc == JAT_OK) {  // group-major copy: per KV head g: 5 q heads, its k head, its v head
      const float* wq = find(p + "attn.q_proj.weight", (int64_t)D * D);
      const float* wk = find(p + "attn.k_proj.weight", (int64_t)kvD * D);
      const float* wv = find(p + "attn.v_proj.weight", (int64_t)kvD * D);
      for (int g = 0; g < m->Hkv && wq && wk && wv; ++g) {
        bf16_t* dst = L.wqkv_g + (int64_t)g * 448 * D;
        if (launch_cast_bf16_rope_rows(wq + (int64_t)g * 320 * D, dst, 320, D, s) != hipSuccess ||
            launch_cast_bf16_rope_rows(wk + (int64_t)g * 64 * D, dst + (int64_t)320 * D, 64, D, s) != hipSuccess ||
            launch_cast_bf16(wv + (int64_t)g * 64 * D, dst + (int64_t)384 * D, (int64_t)64 * D, s) != hipSuccess)
          rc = fail(JAT_E_HIP, "group-major qkv pack failed");
      }
    }
    if (keep32 && build_tables) {   // fold sources: refreshed on a full load only (a training re-pack leaves them stale)
      to_f32(p + "attn.q_proj.weight", L.q32, (int64_t)D * D);
      to_f32(p + "attn.k_proj.weight", L.k32, (int64_t)kvD * D);
      to_f32(p + "attn.v_proj.weight", L.v32, (int64_t)kvD * D);
      to_f32(p + "mlp.0.weight", L.w132, (int64_t)mlp * D);
    }
    to_bf16(p + "attn.out_proj.weight", L.wo, (int64_t)D * D);
    to_bf16(p + "mlp.0.weight", L.w1, (int64_t)mlp * D);
    to_f32(p + "mlp.0.bias", L.b1, mlp);
    to_bf16(p + "mlp.3.weight", L.w2, (int64_t)D * mlp);
    to_f32(p + "mlp.3.bias", L.b2, D);
    to_bf16(p + "adaLN_modulation.1.weight", m->wada + (int64_t)l * 6 * D * D, (int64_t)6 * D * D);
    to_f32(p + "adaLN_modulation.1.bias", m->bada + (int64_t)l * 6 * D, (int64_t)6 * D);
  }
  if (rms) to_f32("final_layer.0.weight", m->final_norm, D); else ones(m->final_norm, D);
  to_bf16("final_layer.1.weight", m->wfinal, (int64_t)m->Fout * D);
  if (keep32 && build_tables) to_f32("final_layer.1.weight", m->wfinal32, (int64_t)m->Fout * D);
  to_f32("final_layer.1.bias", m->bfinal, m->Fout);
  if (rc != JAT_OK) return rc;

  // RoPE tables in fp32 exactly as RoPE.__init__ builds them (jat_audiosr_v3.py:77-85); only the first half
  // of `emb = cat([freqs, freqs])` is distinct.
  if (build_tables) {
    std::vector<float> hc((size_t)MAX_LEN * 32), hs((size_t)MAX_LEN * 32), hf(32);
    for (int i = 0; i < 32; ++i) {
      const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * i) / 64.0f);
      hf[i] = inv_freq;
      for (int pos = 0; pos < MAX_LEN; ++pos) {
        const float a = (float)pos * inv_freq;
        hc[(size_t)pos * 32 + i] = cosf(a);
        hs[(size_t)pos * 32 + i] = sinf(a);
      }
    }
    HIPCHK(hipMemcpyAsync(m->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->rope_invf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  if (build_tables) HIPCHK(hipStreamSynchronize(s));
  m->group_copy_stale = false;
  m->fold_src_ok = keep32 && build_tables;
  m->loaded = true;
  return JAT_OK;
}

extern "C" int jat_model_workspace_bytes(const jat_model* m, int32_t B, int32_t T, size_t* out) {
  if (!m || !out || B <= 0 || T <= 0) return fail(JAT_E_INVALID, "bad argument");
  const int ntok = (T + 3) / 4;
  *out = carve(m, B, ntok, nullptr).total;
  return JAT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// forward pieces
// ---------------------------------------------------------------------------------------------------------

// Tile choice by shape (gemm.hip variant table; measured on MI355X, profiles/r01/gemm_variants.md).  What
// decides is how the tile count quantises onto 256 CUs (one 8-wave block or two 4-wave blocks per CU) and how
// many bytes are staged per MFMA: 256x160 with DMA waves (variant 25; 224 tiles at M=7168, N=1280: one round) >
// 128x160 > 256x256 > 128x128.
static int pick_variant(int M, int N, int nbatch = 1) {
  // score = in-tile efficiency factor x tile-quantisation efficiency on the slots the variant occupies
  // (256 CUs x 1 eight/twelve-wave block, or x 2 four-wave blocks); factors calibrated on the measured block GEMMs
  // at M = 7168 and M = 3584 (profiles/r01/gemm_variants_*.log).  Multi-round 1-block-per-CU variants pay 15 %:
  // their prologue/epilogue is not overlapped by a co-resident block.
  struct Cand { int id, bm, bn, slots; double f; };
  // 31-35: quadrant ping-pong (PIPE 8), one 8-wave block per CU; calibrated on profiles/r02/gemm_variants_*.log
  static const Cand cands[] = {
      {20, 128, 128, 512, 0.95}, {18, 128, 160, 512, 0.95}, {25, 256, 160, 256, 1.00},
      {26, 256, 128, 256, 0.90}, {21, 256, 256, 256, 1.00}, {27, 64, 160, 512, 0.60}, {28, 64, 128, 512, 0.62},
      {31, 224, 320, 256, 1.12}, {32, 256, 160, 256, 1.01}, {33, 256, 256, 256, 1.06},
      {35, 224, 256, 256, 1.06},
  };
  int best = 20;
  double best_score = -1.0;
  for (const Cand& c : cands) {
    if (N % c.bn != 0) continue;
    const long t = (long)((M + c.bm - 1) / c.bm) * (N / c.bn) * nbatch;
    const long rounds = (t + c.slots - 1) / c.slots;
    double score = c.f * (double)t / (double)(rounds * c.slots);
    // multi-round penalty: the 12-wave DMA-wave variants (25, 26) pay their un-overlapped prologue/epilogue per round;
    // the 8-wave tiles (21, 31-35) less so (M = 9660, N = 5120: 110 us vs 124 us for 128x160, tools/gemm_shapes_bench.py)
    if (c.slots == 256 && rounds > 1 && (c.id == 25 || c.id == 26)) score *= 0.85;
    // padding waste of a ragged last row tile counts against big tiles
    score *= (double)M / (double)(((M + c.bm - 1) / c.bm) * c.bm);
    if (score > best_score) { best_score = score; best = c.id; }
  }
  // a half-size batch's QKV GEMM (M = 3584, N = 1792): 392 tiles of 128 x 128 fill 77 % of the 512 four-wave slots; 196 tiles of
  // 256 x 128 with the DMA-wave pipeline are one block on 196 CUs and measured faster (forward 5.530 -> 5.455 ms,
  // profiles/r03/forward_B28_out_qkv_tile_sweep.log).  Only where those tiles make one nearly full round.
  if (best == 20 && nbatch == 1 && N % 128 == 0) {
    const long t26 = (long)((M + 255) / 256) * (N / 128);
    if (t26 >= 192 && t26 <= 256 && M % 256 == 0) best = 26;
  }
  // 36: the tile of 31 with the software-pipelined bf16 / GELU epilogue (JAT_EPI_PIPE=0 keeps the plain one: A/B)
  static const int epi_pipe = getenv("JAT_EPI_PIPE") ? atoi(getenv("JAT_EPI_PIPE")) : 1;
  if (epi_pipe && best == 31) best = 36;
  // 39: the k-step-pair 224 x 160 tile for the N = 1280 class when it fills more of the chip than 256 x 160 (M = 7168: 256 tiles
  // against 224); launch_gemm falls back to 32 for anything but the split-residual producer epilogues.  JAT_KPAIR=0: A/B
  static const int kpair = getenv("JAT_KPAIR") ? atoi(getenv("JAT_KPAIR")) : 1;
  if (kpair && best == 32 && nbatch == 1 && M % 224 == 0 && N % 160 == 0) {
    auto eff = [](long t) { return (double)t / (double)(((t + 255) / 256) * 256); };
    if (eff((long)(M / 224) * (N / 160)) > eff((long)((M + 255) / 256) * (N / 160))) best = 39;
  }
  // 38: the persistent two-tile form of 36 (launch_gemm falls back to 36 for shapes / epilogues it does not take); JAT_PERSIST=0: A/B
  static const int persist = getenv("JAT_PERSIST") ? atoi(getenv("JAT_PERSIST")) : 1;
  if (persist && best == 36 && M % 224 == 0 && (long)(M / 224) * (N / 320) * nbatch > 256) best = 38;
  return best;
}

static int kTileN(int variant) { int bm, bn; gemm_variant_tile(variant, &bm, &bn); return bn; }

int jat_gemm(const jat_model* m, int site, const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, int M, int N,
             int K, int epi, GemmArgs extra, hipStream_t s) {
  GemmArgs a = extra;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.M = M; a.N = N; a.K = K;
  if (a.ksplit > 1) a.K = K / a.ksplit;   // per-slice depth; the kernel shifts A / W / out by blockIdx.y
  int variant = m->variants[site] >= 0 ? m->variants[site] : a.variant_hint > 0 ? a.variant_hint : pick_variant(M, N, a.ksplit > 1 ? a.ksplit : 1);
  if ((a.fold_out || a.rs_part) && !gemm_variant_coalesced(variant)) variant = 20;  // folding lives in the CE epilogues
  if (N % kTileN(variant) != 0) variant = 20;  // 128 x 128, always valid
  if (a.fold_out) { a.fold_np = N / gemm_variant_wave_n(variant); m->last_fold_np = a.fold_np; }
  if (a.rs_part) a.rs_np = m->last_fold_np;
  a.dbg = m->sw.gemm_dbg;   // profiling aid (0 in production)
  // measurement aid (bench.py roofline leg): bracket the launches of one call site with HIP events on the
  // launch stream.  Never active during graph capture (the bench enables it around eager forwards only).
  const bool timed = m->prof.site == site && m->prof.n < (int)m->prof.ev.size() / 2;
  if (timed) { m->prof.stream = s; (void)hipEventRecord(m->prof.ev[2 * m->prof.n], s); }
  hipError_t e = launch_gemm(a, epi, variant, s);
  if (timed) {
    (void)hipEventRecord(m->prof.ev[2 * m->prof.n + 1], s);
    m->prof.flops += 2.0 * M * N * K;
    m->prof.variant = variant;
    ++m->prof.n;
  }
  if (e != hipSuccess) return fail(JAT_E_HIP, "gemm launch (M=%d N=%d K=%d epi=%d): %s", M, N, K, epi, hipGetErrorString(e));
  return JAT_OK;
}
#define gemm jat_gemm

// t [B] -> t_emb [B,D] fp32 (+ bf16 silu(t_emb))  (t_embedder, jat_audiosr_v3.py:364-369)
static int time_path(const jat_model* m, const Workspace& w, const float* t, int B, hipStream_t s) {
  KCHK(launch_time_sinusoid(t, w.e_sin, B, m->D, s));
  KCHK(launch_linear_f32(w.e_sin, m->te_w1, m->te_b1, w.t_h, nullptr, B, m->D, m->D, 1, s));
  KCHK(launch_linear_f32(w.t_h, m->te_w2, m->te_b2, w.t_emb, w.t_silu, B, m->D, m->D, 0, s));
  return JAT_OK;
}
// silu(t_emb) bf16 [B,D] -> mod [B, nlayers*6D] for layers [l0, l0+nl)  (adaLN_modulation, :275-278)
static int adaln_path(const jat_model* m, const bf16_t* t_silu, float* mod, int B, int l0, int nl, hipStream_t s) {
  GemmArgs e{};
  e.out = mod; e.ldo = (int64_t)nl * 6 * m->D; e.bias = m->bada + (int64_t)l0 * 6 * m->D; e.ntok = 1;
  return gemm(m, G_OTHER, t_silu, m->D, m->wada + (int64_t)l0 * 6 * m->D * m->D, m->D, B, nl * 6 * m->D, m->D, EPI_F32, e, s);
}

// Norm folding (sampler path, RMSNorm only): this step's slice of the FoldTable (jat_internal.h); layer offsets are applied
// in run_block.  wqkv_g != nullptr selects the fused QKV+attention kernel (group-major weights), else wqkv_i.
struct Fold {
  const bf16_t *wqkv_g, *wqkv_i, *w1, *wfinal;
  const float *bq_g, *bq_i, *bf;
};

// K-slices for a gated-residual GEMM [M, D] = A[M, K] W^T whose tiles do not fill the chip (small-M inference), 1 = none
static int resid_split(const jat_model* m, const Workspace& w, int site, int M, int K, bool folding, int* variant = nullptr) {
  if (variant) *variant = -1;
  if (!w.kpart || folding || K < 1024 || m->variants[site] >= 0) return 1;
  auto slices = [&](int v, int cap) {
    int bm, bn;
    gemm_variant_tile(v, &bm, &bn);
    const int tiles = ((M + bm - 1) / bm) * (m->D / bn), slots = (v == 18 || v == 20 || v == 27 || v == 28) ? 512 : 256;
    int split = slots / tiles < cap ? slots / tiles : cap;
    while (split > 1 && ((K / 64) % split != 0 || K / split < 256)) --split;   // >= 4 K-tiles per slice
    return split > 1 ? split : 1;
  };
  // Which tile the slices are cut for: 64 x 128 tiles (what pick_variant takes un-split) are bound by the per-CU L2->LDS rate
  // (24 KB per K-tile and block, two blocks per CU); 128 x 128 tiles move 2/3 of the bytes per flop and, cut into more slices,
  // give as many blocks.  Measured per 50-step run (B = 2 / 4 / 8): fc2 133.0 -> 130.3, 169.0 -> 154.5, 253.6 -> 220.1 ms (B = 1:
  // neutral); out_proj only pays from M = 2048 (B = 8: 219.7 -> 212.8 ms).
  if (M <= kSplitMaxRows) return slices((K >= 4096 || M >= 1536) ? 20 : pick_variant(M, m->D), kSplitMax);
  // a mid-size un-folded bucket (a T = 4096 file: M = 2760): the 64 x 128 tiles that fill the chip un-split are bound by the
  // per-CU L2->LDS rate (24 KB per K-tile and block, two blocks per CU); for the long-K fc2 two slices of 128 x 128 tiles
  // (the same 440 blocks, 2/3 of the bytes per flop) + the finishing pass are faster: 70 -> 45 us
  // A half-size batch (configs[1]'s single forward, M = 3584): the 224 x 160 k-step-pair tile makes 128 tiles — two K slices
  // put one on every CU (tile bytes per flop: 0.011 against 0.022 for the 64 x 160 tiles that fill the chip un-split); the
  // finishing pass also applies the norm that follows, which saves the separate norm launch.  JAT_KPAIR_SPLIT=0: A/B
  static const int kpair_split = getenv("JAT_KPAIR_SPLIT") ? atoi(getenv("JAT_KPAIR_SPLIT")) : 1;
  if (kpair_split && variant && M % 224 == 0 && m->D % 160 == 0) {
    const int tiles = (M / 224) * (m->D / 160);
    int split = tiles <= 128 ? 256 / tiles : 1;
    if (split > 2) split = 2;                      // the workspace of this bucket holds two slices (carve)
    while (split > 1 && ((K / 64) % split != 0 || K / split < 512)) --split;
    // only where the slices fill the chip (M = 3136 ... 3584: 224 ... 256 blocks); below that the 128 x 128 slices stay (measured
    // at M = 3584 only: profiles/r03/forward_B28_kernel_table_kpair_split.txt)
    if (split > 1 && tiles * split >= 224 && (K >= 4096 || kpair_split >= 2)) { *variant = 39; return split; }
  }
  return K >= 4096 ? slices(20, 2) : 1;
}

// K-slices for the QKV GEMM of a small bucket (M <= kSplitMaxRows, un-folded, separate attention kernel): its 56 tiles at one
// chunk leave 200 CUs without weights to pull; the slices are summed, rotated and laid out by splitk_qkv_finish_kernel
static int qkv_split(const jat_model* m, const Workspace& w, int M, int K, bool folding) {
  if (!m->sw.qkv_split || m->D % 64 != 0 || m->kvD % 64 != 0 || !w.kpart || folding || M > kSplitMaxRows || m->variants[G_QKV] >= 0) return 1;
  const int N = m->D + 2 * m->kvD;
  int bm, bn;
  const int v = pick_variant(M, N);
  gemm_variant_tile(v, &bm, &bn);
  if (N % bn != 0) return 1;
  const int tiles = ((M + bm - 1) / bm) * (N / bn), slots = (v == 18 || v == 20 || v == 27 || v == 28) ? 512 : 256;
  int split = slots / tiles < kSplitMax ? slots / tiles : kSplitMax;
  while (split > 1 && ((K / 64) % split != 0 || K / split < 256)) --split;
  return split > 1 ? split : 1;
}

// one DiTBlock_GQA on the residual stream w.x  (jat_audiosr_v3.py:284-308); mod_l = this layer's 6D row of batch 0
// The norm that consumes a block's output (norm1 of the next block, or the final norm: shift == nullptr) can ride in the
// finishing pass of a split-K fc2 (small-M buckets): `next` describes it, *next_done reports that w.xn already holds it.
struct NextNorm { const float *w, *shift, *scale; };
static int run_block(const jat_model* m, const Workspace& w, int l, int B, int ntok, const float* mod_l,
                     int64_t bstride, hipStream_t s, const Fold* f = nullptr, bool xn_ready = false,
                     const NextNorm* next = nullptr, bool* next_done = nullptr) {
  const int D = m->D, M = B * ntok, Nqkv = D + 2 * m->kvD;
  const LayerW& L = m->layers[l];
  if (next_done) *next_done = false;
  const bool can_fuse = m->sw.fuse_finish && splitk_resid_norm_supported(D);
  if (!f && !xn_ready) KCHK(launch_norm_modulate(w.x, L.norm1, mod_l + 0 * D, mod_l + 1 * D, bstride, w.xn, M, D, ntok, m->cfg.norm_mode, s));
  const int fuse_env = m->sw.fuse_qkv_attn;   // per-handle switch (jat_model_set_switch): tests A/B the two paths on one model
  // one block per (sample, KV group): worth it only when B * Hkv blocks fill the 256 CUs (measured: +1.8 % at
  // B = 56, -5 % at B = 28); fuse_env = 2 forces it (tests).  With folded weights the sampler decided at creation.
  const bool fused_attn = f ? f->wqkv_g != nullptr
                            : (fuse_env && L.wqkv_g && !m->group_copy_stale && ntok == 128 && (B * m->Hkv >= 192 || fuse_env == 2));
  if (fused_attn) {
    // q/k/v projection + RoPE + attention of one (sample, KV group) per block: q, k, v stay in LDS
    GemmArgs a{};
    a.A = w.xn; a.lda = D; a.W = f ? f->wqkv_g + (int64_t)l * Nqkv * D : L.wqkv_g; a.ldw = D; a.M = M; a.N = m->Hkv * 448; a.K = D;
    a.out = w.ao; a.ldo = D; a.ntok = ntok; a.rope_inv_freq = m->rope_invf;
    a.attn_scale_log2e = 0.125f * 1.4426950408889634f;
    if (f) { a.rs_part = w.part; a.rs_np = m->last_fold_np; a.bias = f->bq_g + (int64_t)l * Nqkv; }
    a.dbg = m->sw.gemm_dbg;   // profiling aid (0 in production)
    KCHK(launch_qkv_attn(a, s));
  } else {
    GemmArgs e{};
    e.out = w.q; e.k_out = w.k; e.vt_out = w.vt; e.D = D; e.kvD = m->kvD; e.npad = w.npad; e.ntok = ntok;
    e.rope_cos = m->rope_cos; e.rope_sin = m->rope_sin; e.rope_inv_freq = m->rope_invf;
    if (f) { e.rs_part = w.part; e.bias = f->bq_i + (int64_t)l * Nqkv; }
    const int qs = qkv_split(m, w, M, D, f != nullptr);
    if (qs > 1) {
      GemmArgs p{};
      p.out = w.kpart; p.ldo = Nqkv; p.ntok = ntok; p.ksplit = qs; p.split_stride = (int64_t)M * Nqkv;
      JCHK(gemm(m, G_QKV, w.xn, D, L.wqkv, D, M, Nqkv, D, EPI_F32, p, s));
      KCHK(launch_splitk_qkv_finish(w.kpart, qs, (int64_t)M * Nqkv, m->rope_cos, m->rope_sin, w.q, w.k, w.vt, M, D, m->kvD, ntok,
                                    w.npad, s));
    } else {
      JCHK(gemm(m, G_QKV, w.xn, D, f ? f->wqkv_i + (int64_t)l * Nqkv * D : L.wqkv, D, M, Nqkv, D, EPI_QKV_ROPE, e, s));
    }
  }
  if (!fused_attn) {
    AttnArgs a{};
    a.q = w.q; a.k = w.k; a.vt = w.vt; a.o = w.ao; a.ldq = D; a.ldk = m->kvD; a.ldo = D;
    a.B = B; a.N = ntok; a.Hq = m->Hq; a.Hkv = m->Hkv; a.npad = w.npad;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    a.lens = w.lens;
    KCHK(launch_attention(a, s));
  }
  bool norm2_done = false;
  {
    GemmArgs e{};
    e.out = w.x; e.ldo = D; e.gate = mod_l + 2 * D; e.gate_bstride = bstride; e.ntok = ntok;
    if (f) { e.fold_out = w.xn; e.fold_lo = w.xlo; e.fold_part = w.part; }
    int sv = -1;
    const int split = resid_split(m, w, G_OUT, M, D, f != nullptr, &sv);
    if (split > 1) {
      GemmArgs p{};
      p.out = w.kpart; p.ldo = D; p.ntok = ntok; p.ksplit = split; p.split_stride = (int64_t)M * D; p.variant_hint = sv;
      JCHK(gemm(m, G_OUT, w.ao, D, L.wo, D, M, D, D, EPI_F32, p, s));
      if (can_fuse && !f) {   // slice sum + gate + residual + norm2 in one launch
        KCHK(launch_splitk_resid_norm(w.kpart, split, (int64_t)M * D, nullptr, mod_l + 2 * D, bstride, w.x, L.norm2, mod_l + 3 * D,
                                      mod_l + 4 * D, bstride, w.xn, M, D, ntok, m->cfg.norm_mode, s));
        norm2_done = true;
      } else {
        KCHK(launch_splitk_resid_finish(w.kpart, split, (int64_t)M * D, nullptr, mod_l + 2 * D, bstride, ntok, w.x, M, D, s));
      }
    } else {
      JCHK(gemm(m, G_OUT, w.ao, D, L.wo, D, M, D, D, EPI_RESID, e, s));
    }
  }
  if (!f && !norm2_done) KCHK(launch_norm_modulate(w.x, L.norm2, mod_l + 3 * D, mod_l + 4 * D, bstride, w.xn, M, D, ntok, m->cfg.norm_mode, s));
  {
    GemmArgs e{};
    e.out = w.hm; e.ldo = m->mlp; e.bias = L.b1; e.ntok = ntok;
    if (f) { e.rs_part = w.part; e.bias = f->bf + (int64_t)l * m->mlp; }
    JCHK(gemm(m, G_FC1, w.xn, D, f ? f->w1 + (int64_t)l * m->mlp * D : L.w1, D, M, m->mlp, D, EPI_BF16_GELU, e, s));
  }
  {
    GemmArgs e{};
    e.out = w.x; e.ldo = D; e.bias = L.b2; e.gate = mod_l + 5 * D; e.gate_bstride = bstride; e.ntok = ntok;
    if (f) { e.fold_out = w.xn; e.fold_lo = w.xlo; e.fold_part = w.part; }   // feeds the next layer's norm1, or the final norm
    int sv = -1;
    const int split = resid_split(m, w, G_FC2, M, m->mlp, f != nullptr, &sv);
    if (split > 1) {
      GemmArgs p{};
      p.out = w.kpart; p.ldo = D; p.ntok = ntok; p.ksplit = split; p.split_stride = (int64_t)M * D; p.variant_hint = sv;
      JCHK(gemm(m, G_FC2, w.hm, m->mlp, L.w2, m->mlp, M, D, m->mlp, EPI_F32, p, s));
      if (can_fuse && !f && next) {   // + the norm that reads this block's output
        KCHK(launch_splitk_resid_norm(w.kpart, split, (int64_t)M * D, L.b2, mod_l + 5 * D, bstride, w.x, next->w, next->shift,
                                      next->scale, bstride, w.xn, M, D, ntok, m->cfg.norm_mode, s));
        if (next_done) *next_done = true;
      } else {
        KCHK(launch_splitk_resid_finish(w.kpart, split, (int64_t)M * D, L.b2, mod_l + 5 * D, bstride, ntok, w.x, M, D, s));
      }
    } else {
      JCHK(gemm(m, G_FC2, w.hm, m->mlp, L.w2, m->mlp, M, D, m->mlp, EPI_RESID, e, s));
    }
  }
  return JAT_OK;
}

// Whole forward over B batch rows.  x_t rows are read modulo B_src and the condition is zero from batch row
// cond_zero_from on: this is how the CFG double batch [z;z],[lr;0] (infer_test_v3m2.py:154-156) is fed
// without materialising the concatenations.  mod == nullptr: compute the modulation from t [B].
static int forward_impl(const jat_model* m, const Workspace& w, const float* x_t, int B_src, const float* x_cond,
                        int cond_zero_from, const float* t, const float* mod, int64_t mod_bstride, float* x_pred,
                        int B, int T, hipStream_t s, const Fold* f = nullptr, const float* pc = nullptr) {
  const int ntok = (T + 3) / 4, M = B * ntok, D = m->D;
  if (!mod) {
    JCHK(time_path(m, w, t, B, s));
    // (Running this 0.55 GB weight stream on a second stream beside the patch embed was measured: 5.68 -> 5.71 ms at B = 28 —
    // its 1344 blocks and the patch GEMM's share the same CUs, nothing is gained.)
    JCHK(adaln_path(m, w.t_silu, w.mod, B, 0, m->depth, s));
    mod = w.mod;
    mod_bstride = (int64_t)m->depth * 6 * D;
  }
  // The key padding of V^T must be zero (0 * garbage could be NaN).  Nothing ever writes the padding, so the
  // sampler zeroes its private buffer once at creation (mod != nullptr path) and only the generic entry point,
  // whose workspace belongs to the caller, clears it per call.
  if (t) HIPCHK(hipMemsetAsync(w.vt, 0, w.vt_bytes, s));
  // The first patch-embed Linear is narrow and deep ([rows, 4096 or 8192] x [512, .]^T: 64 x 128 tiles make at most one 4-wave
  // block per CU at the bench's batch, each walking 64-128 K-tiles): K slices put two blocks on every CU, the finishing pass
  // adds bias and GELU (same expression as the epilogue).  The partials live in the MLP hidden buffer, idle until block 0's fc1.
  auto patch_split = [&](int rows, int K) {
    if (!m->sw.patch_split || m->variants[G_OTHER] >= 0 || m->bott % 128 != 0) return 1;
    const int tiles = ((rows + 63) / 64) * (m->bott / 128);
    // measured (profiles/r03/patch_embed_split_ab.log): pays for the single forward (K = 8192: 5.45 -> 5.41 ms) and for one chunk
    // (24 tiles: 133.4 -> 131.9 ms), not for the sampler's half-depth form at the bench's batch (224 tiles, K = 4096: 353.1 vs 353.6 ms)
    if (tiles > 128 && K < 8192) return 1;
    int split = 512 / tiles, cap = m->mlp / (2 * m->bott);   // partial slices must fit w.hm: split * bott * 4 <= mlp * 2 bytes per row
    if (cap > 4) cap = 4;
    if (split > cap) split = cap;
    while (split > 1 && ((K / 64) % split != 0 || K / split < 1024)) --split;
    return split > 1 ? split : 1;
  };
  if (pc) {
    // CFG sampler: the first patch-embed Linear is linear in [z ; cond], the z part is the same for the cond and
    // uncond halves and the cond part (pc = patch(lr) @ W1[:, cond]^T, fp32) does not change over the 50 steps:
    // h_cond = gelu(S_z + pc + b1), h_uncond = gelu(S_z + b1) from ONE quarter-size GEMM (M/2 rows, K/2 deep).
    const int Mh = M / 2, Kz = m->P * m->Cin;
    KCHK(launch_patchify(x_t, nullptr, w.a_patch, B_src, B_src, B_src, m->Cin, 0, T, ntok, s, w.tvalid));
    const int ps = patch_split(Mh, Kz);
    if (ps > 1) {
      GemmArgs e{};
      e.out = w.hm; e.ldo = m->bott; e.ntok = ntok; e.ksplit = ps; e.split_stride = (int64_t)Mh * m->bott;
      JCHK(gemm(m, G_OTHER, w.a_patch, Kz, m->pe_w1, m->Kp, Mh, m->bott, Kz, EPI_F32, e, s));
      KCHK(launch_splitk_gelu_finish((const float*)w.hm, ps, (int64_t)Mh * m->bott, m->pe_b1, pc, Mh, w.h_patch, m->bott, Mh, m->bott, s));
    } else {
      GemmArgs e{};
      e.out = w.h_patch; e.ldo = m->bott; e.bias = m->pe_b1; e.ntok = ntok; e.dual_add = pc; e.dual_rows = Mh;
      JCHK(gemm(m, G_OTHER, w.a_patch, Kz, m->pe_w1, m->Kp, Mh, m->bott, Kz, EPI_BF16_GELU, e, s));
    }
  } else {
    KCHK(launch_patchify(x_t, x_cond, w.a_patch, B, B_src, cond_zero_from, m->Cin, m->Cc, T, ntok, s, w.tvalid));
    const int ps = patch_split(M, m->Kp);
    if (ps > 1) {
      GemmArgs e{};
      e.out = w.hm; e.ldo = m->bott; e.ntok = ntok; e.ksplit = ps; e.split_stride = (int64_t)M * m->bott;
      JCHK(gemm(m, G_OTHER, w.a_patch, m->Kp, m->pe_w1, m->Kp, M, m->bott, m->Kp, EPI_F32, e, s));
      KCHK(launch_splitk_gelu_finish((const float*)w.hm, ps, (int64_t)M * m->bott, m->pe_b1, nullptr, 0, w.h_patch, m->bott, M, m->bott, s));
    } else {
      GemmArgs e{};
      e.out = w.h_patch; e.ldo = m->bott; e.bias = m->pe_b1; e.ntok = ntok;
      JCHK(gemm(m, G_OTHER, w.a_patch, m->Kp, m->pe_w1, m->Kp, M, m->bott, m->Kp, EPI_BF16_GELU, e, s));
    }
  }
  {
    GemmArgs e{};
    e.out = w.x; e.ldo = D; e.bias = m->pe_b2; e.ntok = ntok;
    if (f) { e.fold_out = w.xn; e.fold_lo = w.xlo; e.fold_part = w.part; }
    JCHK(gemm(m, G_OTHER, w.h_patch, m->bott, m->pe_w2, m->bott, M, D, m->bott, EPI_F32, e, s));
  }
  bool xn_ready = false;
  for (int l = 0; l < m->depth; ++l) {
    const float* mod_n = mod + (int64_t)(l + 1) * 6 * D;
    const NextNorm nn = l + 1 < m->depth ? NextNorm{m->layers[l + 1].norm1, mod_n + 0 * D, mod_n + 1 * D}
                                         : NextNorm{m->final_norm, nullptr, nullptr};
    bool done = false;
    JCHK(run_block(m, w, l, B, ntok, mod + (int64_t)l * 6 * D, mod_bstride, s, f, xn_ready, &nn, &done));
    xn_ready = done;
  }
  if (!f && !xn_ready) KCHK(launch_norm_modulate(w.x, m->final_norm, nullptr, nullptr, 0, w.xn, M, D, ntok, m->cfg.norm_mode, s));
  {
    GemmArgs e{};
    e.out = x_pred; e.bias = m->bfinal; e.ntok = ntok; e.C_out = m->Cin; e.T_orig = T;
    if (f) e.rs_part = w.part;
    JCHK(gemm(m, G_OTHER, w.xn, D, f ? f->wfinal : m->wfinal, D, M, m->Fout, D, EPI_UNPATCH, e, s));
  }
  return JAT_OK;
}

static int check_ready(const jat_model* m, int B, int ntok, void* ws, size_t ws_bytes, Workspace* w) {
  if (!m) return fail(JAT_E_INVALID, "null model");
  if (!m->loaded) return fail(JAT_E_STATE, "weights not loaded");
  if (B <= 0 || ntok <= 0) return fail(JAT_E_INVALID, "B and T must be positive");
  if (ntok > MAX_LEN) return fail(JAT_E_SEQLEN, "Sequence length %d exceeds max_len %d", ntok, MAX_LEN);
  *w = carve(m, B, ntok, (char*)ws);
  if (!ws || ws_bytes < w->total)
    return fail(JAT_E_STATE, "workspace too small: %zu < %zu bytes", ws_bytes, w->total);
  return JAT_OK;
}

extern "C" int jat_forward(jat_model* m, const float* x_t, const float* t, const float* x_cond, float* x_pred,
                           int32_t B, int32_t T, void* ws, size_t ws_bytes, void* stream) {
  if (T <= 0) return fail(JAT_E_INVALID, "T must be positive");
  Workspace w;
  JCHK(check_ready(m, B, (T + 3) / 4, ws, ws_bytes, &w));
  return forward_impl(m, w, x_t, B, x_cond, B, t, nullptr, 0, x_pred, B, T, (hipStream_t)stream);
}

extern "C" int jat_time_embed(jat_model* m, const float* t, float* t_emb, int32_t B, void* ws, size_t ws_bytes,
                              void* stream) {
  Workspace w;
  JCHK(check_ready(m, B, 1, ws, ws_bytes, &w));
  hipStream_t s = (hipStream_t)stream;
  JCHK(time_path(m, w, t, B, s));
  HIPCHK(hipMemcpyAsync(t_emb, w.t_emb, (size_t)B * m->D * 4, hipMemcpyDeviceToDevice, s));
  return JAT_OK;
}

extern "C" int jat_block_forward(jat_model* m, int32_t layer, const float* x, const float* t_emb, float* y, int32_t B,
                                 int32_t N, void* ws, size_t ws_bytes, void* stream) {
  Workspace w;
  JCHK(check_ready(m, B, N, ws, ws_bytes, &w));
  if (layer < 0 || layer >= m->depth) return fail(JAT_E_INVALID, "layer out of range");
  hipStream_t s = (hipStream_t)stream;
  const size_t xb = (size_t)B * N * m->D * 4;
  KCHK(launch_silu_bf16(t_emb, w.t_silu, (int64_t)B * m->D, s));
  JCHK(adaln_path(m, w.t_silu, w.mod, B, layer, 1, s));
  HIPCHK(hipMemsetAsync(w.vt, 0, w.vt_bytes, s));
  HIPCHK(hipMemcpyAsync(w.x, x, xb, hipMemcpyDeviceToDevice, s));
  JCHK(run_block(m, w, layer, B, N, w.mod, (int64_t)6 * m->D, s));
  HIPCHK(hipMemcpyAsync(y, w.x, xb, hipMemcpyDeviceToDevice, s));
  return JAT_OK;
}

extern "C" int jat_attn_forward(jat_model* m, int32_t layer, const float* x, float* y, int32_t B, int32_t N, void* ws,
                                size_t ws_bytes, void* stream) {
  Workspace w;
  JCHK(check_ready(m, B, N, ws, ws_bytes, &w));
  if (layer < 0 || layer >= m->depth) return fail(JAT_E_INVALID, "layer out of range");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->D, M = B * N;
  const LayerW& L = m->layers[layer];
  HIPCHK(hipMemsetAsync(w.vt, 0, w.vt_bytes, s));
  KCHK(launch_norm_modulate(x, nullptr, nullptr, nullptr, 0, w.xn, M, D, N, 2, s));  // plain bf16 cast
  {
    GemmArgs e{};
    e.out = w.q; e.k_out = w.k; e.vt_out = w.vt; e.D = D; e.kvD = m->kvD; e.npad = w.npad; e.ntok = N;
    e.rope_cos = m->rope_cos; e.rope_sin = m->rope_sin; e.rope_inv_freq = m->rope_invf;
    JCHK(gemm(m, G_QKV, w.xn, D, L.wqkv, D, M, D + 2 * m->kvD, D, EPI_QKV_ROPE, e, s));
  }
  {
    AttnArgs a{};
    a.q = w.q; a.k = w.k; a.vt = w.vt; a.o = w.ao; a.ldq = D; a.ldk = m->kvD; a.ldo = D;
    a.B = B; a.N = N; a.Hq = m->Hq; a.Hkv = m->Hkv; a.npad = w.npad;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    KCHK(launch_attention(a, s));
  }
  {
    GemmArgs e{};
    e.out = y; e.ldo = D; e.ntok = N;
    JCHK(gemm(m, G_OUT, w.ao, D, L.wo, D, M, D, D, EPI_F32, e, s));
  }
  return JAT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// sampler
// ---------------------------------------------------------------------------------------------------------
struct jat_sampler {
  jat_model* m;
  int B, T, steps, Bf;  // Bf = batch rows per forward (2B with CFG)
  float cfg_scale;
  bool use_cfg;
  std::vector<float> ts;  // [host] linspace(0,1,steps+1)
  char* blob = nullptr;   // private device allocation
  float *z, *lr, *xpred, *mod_table, *ts_dev;
  std::shared_ptr<FoldTable> fold;   // per-step folded weights (RMSNorm models), shared through the model's cache
  bool fused_attn = false;           // this bucket runs the fused QKV+attention kernel (group-major folded weights)
  float* pc = nullptr;   // CFG: patch(lr) @ W1[:, cond]^T, recomputed once per run (split patch embed)
  int* lens_dev = nullptr;   // [Bf] valid tokens per batch row (default: ntok), read by the attention kernel of the graph
  int* frames_dev = nullptr; // [B] valid frames per row (default: T), read by the patchify kernels
  bool folded = false;
  void* ws;
  size_t ws_bytes;
  Workspace w;
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

// torch.linspace(0, 1, n) in fp32 as PyTorch evaluates it (step = fp32(1/(n-1)); first half start+step*i,
// second half end-step*(n-1-i), one rounding each) — pinned by tests/golden/misc.npz `linspace51`.
static void linspace01(int n, std::vector<float>& out) {
  out.resize(n);
  const float step = 1.0f / (float)(n - 1);
  for (int i = 0; i < n; ++i)
    out[i] = i < n / 2 ? (float)((double)step * i) : (float)(1.0 - (double)step * (n - 1 - i));
}

// pc = patch(lr) @ W1[:, cond columns]^T (fp32, no bias): once per run, before the captured steps
static int sampler_cond_part(jat_sampler* sp, hipStream_t s) {
  if (!sp->pc) return JAT_OK;
  jat_model* m = sp->m;
  const int ntok = (sp->T + 3) / 4, Kc = m->P * m->Cc;
  KCHK(launch_patchify(sp->lr, nullptr, sp->w.a_patch, sp->B, sp->B, sp->B, m->Cc, 0, sp->T, ntok, s, sp->w.tvalid));
  GemmArgs e{};
  e.out = sp->pc; e.ldo = m->bott; e.ntok = ntok;
  return gemm(m, G_OTHER, sp->w.a_patch, Kc, m->pe_w1 + (int64_t)m->P * m->Cin, m->Kp, sp->B * ntok, m->bott, Kc, EPI_F32, e, s);
}

static int sampler_steps(jat_sampler* sp, hipStream_t s) {
  jat_model* m = sp->m;
  const int64_t n_half = (int64_t)sp->B * m->Cin * sp->T;
  const int64_t row = (int64_t)m->depth * 6 * m->D;
  for (int i = 0; i < sp->steps; ++i) {
    const float t_curr = sp->ts[i], dt = sp->ts[i + 1] - sp->ts[i];
    Fold f{};
    if (sp->folded) {
      const FoldTable& ft = *sp->fold;
      const int64_t Nqkv = m->D + 2 * m->kvD, dl = m->depth;
      if (sp->fused_attn) { f.wqkv_g = ft.qkv_g + (int64_t)i * dl * Nqkv * m->D; f.bq_g = ft.bq_g + (int64_t)i * dl * Nqkv; }
      else { f.wqkv_i = ft.qkv_i + (int64_t)i * dl * Nqkv * m->D; f.bq_i = ft.bq_i + (int64_t)i * dl * Nqkv; }
      f.w1 = ft.w1 + (int64_t)i * dl * m->mlp * m->D;
      f.bf = ft.bf + (int64_t)i * dl * m->mlp;
      f.wfinal = ft.wfinal;
    }
    JCHK(forward_impl(m, sp->w, sp->z, sp->B, sp->lr, sp->B, nullptr, sp->mod_table + i * row, 0, sp->xpred, sp->Bf,
                      sp->T, s, sp->folded ? &f : nullptr, sp->pc));
    KCHK(launch_cfg_euler(sp->xpred, sp->z, sp->cfg_scale, t_curr, dt, sp->use_cfg ? 1 : 0, n_half, s));
  }
  return JAT_OK;
}

// Build (or extend with the missing QKV layout) the folded-weight table of `steps` steps from the modulation table
// mod [steps][depth*6D] (fp32, device).  Everything is enqueued on s; the caller synchronises.  On an allocation
// failure the sampler simply runs un-folded (norm kernels).
static int fold_alloc(FoldTable& ft, void** p, size_t bytes) {
  if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(JAT_E_HIP, "hipMalloc(%zu) for the folded weights", bytes); }
  ft.allocs.push_back(*p);
  return JAT_OK;
}
static int build_fold_table_impl(jat_model* m, FoldTable& ft, int steps, const float* mod, bool group_major, bf16_t* sh_bf16,
                                 hipStream_t s, bool need_w1, bool need_qkv);
static int build_fold_table(jat_model* m, FoldTable& ft, int steps, const float* mod, bool group_major, bf16_t* sh_bf16,
                            hipStream_t s) {
  const bool need_w1 = ft.w1 == nullptr;
  const bool need_qkv = group_major ? ft.qkv_g == nullptr : ft.qkv_i == nullptr;
  const size_t keep = ft.allocs.size();
  const int rc = build_fold_table_impl(m, ft, steps, mod, group_major, sh_bf16, s, need_w1, need_qkv);
  if (rc != JAT_OK) {
    // undo THIS call's allocations and pointers: what earlier calls built (and other samplers hold) stays valid, and no
    // half-built part is left looking finished
    (void)hipStreamSynchronize(s);
    while (ft.allocs.size() > keep) { (void)hipFree(ft.allocs.back()); ft.allocs.pop_back(); }
    if (need_w1) { ft.w1 = nullptr; ft.bf = nullptr; ft.wfinal = nullptr; }
    if (need_qkv) { if (group_major) { ft.qkv_g = nullptr; ft.bq_g = nullptr; } else { ft.qkv_i = nullptr; ft.bq_i = nullptr; } }
  }
  return rc;
}
static int build_fold_table_impl(jat_model* m, FoldTable& ft, int steps, const float* mod, bool group_major, bf16_t* sh_bf16,
                                 hipStream_t s, bool need_w1, bool need_qkv) {
  const int D = m->D, kvD = m->kvD, mlp = m->mlp, depth = m->depth, Nqkv = D + 2 * kvD;
  const int64_t mrow = (int64_t)depth * 6 * D;
  if (need_w1) {
    JCHK(fold_alloc(ft, (void**)&ft.w1, (size_t)steps * depth * mlp * D * 2));
    JCHK(fold_alloc(ft, (void**)&ft.bf, (size_t)steps * depth * mlp * 4));
    JCHK(fold_alloc(ft, (void**)&ft.wfinal, (size_t)m->Fout * D * 2));
    KCHK(launch_fold_weight(m->wfinal32, m->final_norm, nullptr, ft.wfinal, m->Fout, D, 0, s));
  }
  bf16_t* qkv = nullptr;
  float* bq = nullptr;
  if (need_qkv) {
    JCHK(fold_alloc(ft, (void**)&qkv, (size_t)steps * depth * Nqkv * D * 2));
    JCHK(fold_alloc(ft, (void**)&bq, (size_t)steps * depth * Nqkv * 4));
    if (group_major) { ft.qkv_g = qkv; ft.bq_g = bq; } else { ft.qkv_i = qkv; ft.bq_i = bq; }
  }
  ft.steps = steps;
  for (int l = 0; l < depth; ++l) {
    const LayerW& L = m->layers[l];
    for (int i = 0; i < steps; ++i) {
      const float* mod_il = mod + i * mrow + (int64_t)l * 6 * D;   // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      if (need_qkv) {
        bf16_t* dst = qkv + ((int64_t)i * depth + l) * Nqkv * D;
        if (group_major) {   // per KV head g: its 5 q heads, its k head, its v head (the layout of L.wqkv_g)
          for (int g = 0; g < m->Hkv; ++g) {
            bf16_t* dg = dst + (int64_t)g * 448 * D;
            KCHK(launch_fold_weight(L.q32 + (int64_t)g * 320 * D, L.norm1, mod_il + D, dg, 320, D, 1, s));
            KCHK(launch_fold_weight(L.k32 + (int64_t)g * 64 * D, L.norm1, mod_il + D, dg + (int64_t)320 * D, 64, D, 1, s));
            KCHK(launch_fold_weight(L.v32 + (int64_t)g * 64 * D, L.norm1, mod_il + D, dg + (int64_t)384 * D, 64, D, 0, s));
          }
        } else {
          KCHK(launch_fold_weight(L.q32, L.norm1, mod_il + D, dst, D, D, 1, s));
          KCHK(launch_fold_weight(L.k32, L.norm1, mod_il + D, dst + (int64_t)D * D, kvD, D, 1, s));
          KCHK(launch_fold_weight(L.v32, L.norm1, mod_il + D, dst + (int64_t)(D + kvD) * D, kvD, D, 0, s));
        }
      }
      if (need_w1) KCHK(launch_fold_weight(L.w132, L.norm2, mod_il + 4 * D, ft.w1 + ((int64_t)i * depth + l) * mlp * D, mlp, D, 0, s));
    }
    // shift @ W^T for all steps of this layer in one skinny GEMM each (un-folded packed weights, fp32 accumulate)
    if (need_qkv) {
      KCHK(launch_gather_cast_rows(mod + (int64_t)l * 6 * D, mrow, sh_bf16, steps, D, s));
      GemmArgs e{};
      e.out = bq + (int64_t)l * Nqkv; e.ldo = (int64_t)depth * Nqkv; e.ntok = 1;
      JCHK(gemm(m, G_OTHER, sh_bf16, D, group_major ? L.wqkv_g : L.wqkv, D, steps, Nqkv, D, EPI_F32, e, s));
    }
    if (need_w1) {
      KCHK(launch_gather_cast_rows(mod + (int64_t)l * 6 * D + 3 * D, mrow, sh_bf16, steps, D, s));
      GemmArgs e{};
      e.out = ft.bf + (int64_t)l * mlp; e.ldo = (int64_t)depth * mlp; e.bias = L.b1; e.ntok = 1;
      JCHK(gemm(m, G_OTHER, sh_bf16, D, L.w1, D, steps, mlp, D, EPI_F32, e, s));
    }
  }
  return JAT_OK;
}

extern "C" void jat_sampler_destroy(jat_sampler* sp) {
  if (!sp) return;
  if (sp->exec) (void)hipGraphExecDestroy(sp->exec);
  if (sp->graph) (void)hipGraphDestroy(sp->graph);
  if (sp->cap_stream) (void)hipStreamDestroy(sp->cap_stream);
  if (sp->blob) (void)hipFree(sp->blob);
  delete sp;
}

extern "C" int jat_sampler_create(jat_model* m, int32_t B, int32_t T, int32_t steps, float cfg_scale,
                                  jat_sampler** out) {
  if (!m || !out) return fail(JAT_E_INVALID, "null argument");
  if (!m->loaded) return fail(JAT_E_STATE, "weights not loaded");
  if (B <= 0 || T <= 0 || steps <= 0) return fail(JAT_E_INVALID, "B, T, steps must be positive");
  const int ntok = (T + 3) / 4;
  if (ntok > MAX_LEN) return fail(JAT_E_SEQLEN, "Sequence length %d exceeds max_len %d", ntok, MAX_LEN);
  // The tables and the captured graph are built on a private stream: order them after everything already enqueued on
  // the caller's streams (e.g. an asynchronous weight re-pack).  Creation is a slow path; a device sync is the simple order.
  HIPCHK(hipDeviceSynchronize());
  jat_sampler* sp = new jat_sampler();
  sp->m = m; sp->B = B; sp->T = T; sp->steps = steps; sp->cfg_scale = cfg_scale;
  sp->use_cfg = cfg_scale != 1.0f;  // infer_test_v3m2.py:139
  sp->Bf = sp->use_cfg ? 2 * B : B;
  linspace01(steps + 1, sp->ts);

  const size_t lat = (size_t)B * m->Cin * T * 4;
  const size_t row = (size_t)m->depth * 6 * m->D;
  const int Bws = sp->Bf > steps ? sp->Bf : steps;  // the table precompute runs the time path on `steps` rows
  const size_t ws_bytes = carve(m, Bws, ntok, nullptr).total;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t o_z = take(lat), o_lr = take(lat), o_xp = take((size_t)sp->Bf * m->Cin * T * 4);
  const size_t o_tab = take((size_t)steps * row * 4), o_ts = take((size_t)steps * 4), o_ws = take(ws_bytes);
  // Norm folding (default on for RMSNorm models; JAT_FOLD_NORM=0 keeps the norm kernels): decided per sampler.
  const int fold_env = m->sw.fold_norm;
  {  // the consumer side reads the row partials lane-linear: needs 4, 8 or 16 slots per row; the three producers of the
     // residual stream (patch embed, out_proj, fc2: all [M, D]) must agree on the slot count
    const int M = sp->Bf * ntok;
    auto var = [&](int site) { return m->variants[site] >= 0 ? m->variants[site] : pick_variant(M, m->D); };
    const int np = m->D / gemm_variant_wave_n(var(G_OUT));
    sp->folded = fold_env > 0 && m->cfg.norm_mode == JAT_NORM_RMS_W && m->fold_src_ok && (np == 4 || np == 8 || np == 16) &&
                 gemm_variant_coalesced(var(G_OUT)) && gemm_variant_coalesced(var(G_FC2)) && gemm_variant_coalesced(var(G_OTHER)) &&
                 gemm_variant_wave_n(var(G_OUT)) == gemm_variant_wave_n(var(G_FC2)) &&
                 gemm_variant_wave_n(var(G_OUT)) == gemm_variant_wave_n(var(G_OTHER)) &&
                 (M > kSplitMaxRows || fold_env >= 2);   // small-M buckets finish fc2 / out_proj with split-K instead (2: force, tests)
    const int fuse_env = m->sw.fuse_qkv_attn;
    sp->fused_attn = fuse_env && m->Hq / m->Hkv == 5 && !m->group_copy_stale && ntok == 128 && (sp->Bf * m->Hkv >= 192 || fuse_env == 2);
  }
  const size_t o_sh = take((size_t)steps * m->D * 2);
  const size_t o_lens = take((size_t)sp->Bf * 4), o_frames = take((size_t)B * 4);
  const size_t o_pc = take((size_t)B * ntok * m->bott * 4);
  hipError_t e = hipMalloc((void**)&sp->blob, off);
  if (e != hipSuccess) { delete sp; return fail(JAT_E_HIP, "hipMalloc(%zu): %s", off, hipGetErrorString(e)); }
  sp->z = (float*)(sp->blob + o_z); sp->lr = (float*)(sp->blob + o_lr); sp->xpred = (float*)(sp->blob + o_xp);
  sp->mod_table = (float*)(sp->blob + o_tab); sp->ts_dev = (float*)(sp->blob + o_ts);
  sp->ws = sp->blob + o_ws; sp->ws_bytes = ws_bytes;
  bf16_t* sh_bf16 = (bf16_t*)(sp->blob + o_sh);
  sp->lens_dev = (int*)(sp->blob + o_lens);
  sp->frames_dev = (int*)(sp->blob + o_frames);
  if (sp->use_cfg && m->sw.split_patch) sp->pc = (float*)(sp->blob + o_pc);

  int rc = JAT_OK;
  auto bail = [&](int code) { jat_sampler_destroy(sp); return code; };
  if (hipStreamCreateWithFlags(&sp->cap_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(JAT_E_HIP, "hipStreamCreate failed"));
  hipStream_t s = sp->cap_stream;

  // modulation table [steps, depth*6D]: every row of step i shares t = ts[i] (infer_test_v3m2.py:150)
  {
    Workspace wt = carve(m, steps, ntok, (char*)sp->ws);
    if (hipMemcpyAsync(sp->ts_dev, sp->ts.data(), (size_t)steps * 4, hipMemcpyHostToDevice, s) != hipSuccess)
      return bail(fail(JAT_E_HIP, "memcpy ts"));
    if ((rc = time_path(m, wt, sp->ts_dev, steps, s)) != JAT_OK) return bail(rc);
    if ((rc = adaln_path(m, wt.t_silu, sp->mod_table, steps, 0, m->depth, s)) != JAT_OK) return bail(rc);
    if (sp->folded) {
      // per-step folded weights: shared by every sampler of this model with the same step count (model-level cache)
      // The cache keeps at most TWO step counts alive on its own (a table is ~0.5 GB per step for v3mod2: 25 GB at 50 steps,
      // 50 GB at the 100 steps the reference README also offers); tables that a live sampler still holds stay until it is destroyed.
      for (auto it = m->fold_cache.begin(); it != m->fold_cache.end() && m->fold_cache.size() >= 2;)
        it = (it->first != steps && it->second.use_count() == 1) ? m->fold_cache.erase(it) : std::next(it);
      const size_t Nq = (size_t)m->D + 2 * m->kvD;
      const size_t need = (size_t)steps * m->depth * ((Nq + m->mlp) * m->D * 2 + (Nq + m->mlp) * 4) + (size_t)m->Fout * m->D * 2;
      const bool over_cap = m->sw.fold_cap_mb > 0 && need > (size_t)m->sw.fold_cap_mb * 1048576;   // "fold_cap_mb": operator's bound
      std::shared_ptr<FoldTable>& slot = m->fold_cache[steps];
      if (!slot) slot = std::make_shared<FoldTable>();
      if (over_cap || build_fold_table(m, *slot, steps, sp->mod_table, sp->fused_attn, sh_bf16, s) != JAT_OK) {
        (void)hipStreamSynchronize(s);
        // over the cap or out of memory: run this sampler with the norm kernels.  What was already built stays for the samplers
        // that hold it; an EMPTY entry is dropped so that a later, smaller request starts clean
        if (!slot->w1 && !slot->qkv_g && !slot->qkv_i) m->fold_cache.erase(steps);
        sp->folded = false;
      } else {
        sp->fold = slot;
      }
    }
    if (hipStreamSynchronize(s) != hipSuccess) return bail(fail(JAT_E_HIP, "sync after table build"));
  }
  sp->w = carve(m, sp->Bf, ntok, (char*)sp->ws);
  if (hipMemsetAsync(sp->w.vt, 0, sp->w.vt_bytes, s) != hipSuccess) return bail(fail(JAT_E_HIP, "memset vt"));
  {  // every row attends to all of its ntok keys until jat_sampler_set_lengths says otherwise; the fused QKV+attention
     // kernel (ntok == 128) has no key mask and is never combined with lengths
    std::vector<int> full((size_t)sp->Bf, ntok), fullT((size_t)B, T);
    if (hipMemcpyAsync(sp->lens_dev, full.data(), full.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(sp->frames_dev, fullT.data(), fullT.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return bail(fail(JAT_E_HIP, "lengths upload"));
    if (!sp->fused_attn) { sp->w.lens = sp->lens_dev; sp->w.tvalid = sp->frames_dev; }
  }

  // one eager pass first: sets every kernel's function attributes outside of capture and validates launches
  if (hipMemsetAsync(sp->z, 0, lat, s) != hipSuccess || hipMemsetAsync(sp->lr, 0, lat, s) != hipSuccess)
    return bail(fail(JAT_E_HIP, "memset"));
  {
    const int saved = sp->steps;
    sp->steps = 1;
    if ((rc = sampler_cond_part(sp, s)) != JAT_OK) return bail(rc);
    rc = sampler_steps(sp, s);
    sp->steps = saved;
    if (rc != JAT_OK) return bail(rc);
    if (hipStreamSynchronize(s) != hipSuccess)
      return bail(fail(JAT_E_HIP, "eager warm-up step failed: %s", hipGetErrorString(hipGetLastError())));
  }
  // capture all `steps` forwards + Euler updates into one graph
  if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess)
    return bail(fail(JAT_E_HIP, "hipStreamBeginCapture failed"));
  rc = sampler_steps(sp, s);
  e = hipStreamEndCapture(s, &sp->graph);
  if (rc != JAT_OK) return bail(rc);
  if (e != hipSuccess) return bail(fail(JAT_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e)));
  e = hipGraphInstantiate(&sp->exec, sp->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) return bail(fail(JAT_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)));
  *out = sp;
  return JAT_OK;
}

// what this sampler's captured graph runs: folded != 0 = per-step folded weights (no norm kernels), fused_attn != 0 = the fused
// QKV + RoPE + attention kernel, fold_bytes = size of the folded-weight table it shares through the model (0 when not folded)
extern "C" int jat_sampler_info(const jat_sampler* sp, int32_t* folded, int32_t* fused_attn, int64_t* fold_bytes) {
  if (!sp) return fail(JAT_E_INVALID, "null sampler");
  if (folded) *folded = sp->folded ? 1 : 0;
  if (fused_attn) *fused_attn = sp->fused_attn ? 1 : 0;
  if (fold_bytes) {
    const jat_model* m = sp->m;
    const size_t Nq = (size_t)m->D + 2 * m->kvD;
    *fold_bytes = sp->folded ? (int64_t)((size_t)sp->steps * m->depth * ((Nq + m->mlp) * m->D * 2 + (Nq + m->mlp) * 4) + (size_t)m->Fout * m->D * 2) : 0;
  }
  return JAT_OK;
}

extern "C" int jat_sampler_set_lengths(jat_sampler* sp, const int32_t* frames, int32_t n, void* stream) {
  if (!sp || !frames) return fail(JAT_E_INVALID, "null argument");
  if (n != sp->B) return fail(JAT_E_INVALID, "need one length per batch row (%d), got %d", sp->B, n);
  std::vector<int> tok((size_t)sp->Bf);
  bool all_full = true;
  for (int b = 0; b < sp->B; ++b) {
    if (frames[b] <= 0 || frames[b] > sp->T) return fail(JAT_E_INVALID, "length %d of row %d outside (0, %d]", frames[b], b, sp->T);
    tok[b] = (frames[b] + 3) / 4;                       // the reference pads a chunk to a multiple of 4 frames (:435-439)
    if (sp->use_cfg) tok[sp->B + b] = tok[b];           // CFG double batch [cond ; uncond]
    all_full = all_full && frames[b] == sp->T;          // per FRAME: the fused buckets wire no frame mask into patchify either
  }
  if (sp->fused_attn && !all_full)
    return fail(JAT_E_STATE, "this bucket runs the fused QKV+attention kernel (128 tokens), which has no key mask");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(sp->lens_dev, tok.data(), tok.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(sp->frames_dev, frames, (size_t)sp->B * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));                      // `tok` / `frames` are host memory of the caller
  return JAT_OK;
}

extern "C" int jat_sampler_run(jat_sampler* sp, const float* lr_latent, const float* z0, float* z_out,
                               int32_t use_graph, void* stream) {
  if (!sp || !lr_latent || !z0 || !z_out) return fail(JAT_E_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  const size_t lat = (size_t)sp->B * sp->m->Cin * sp->T * 4;
  HIPCHK(hipMemcpyAsync(sp->lr, lr_latent, lat, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(sp->z, z0, lat, hipMemcpyDeviceToDevice, s));
  JCHK(sampler_cond_part(sp, s));
  if (use_graph) {
    HIPCHK(hipGraphLaunch(sp->exec, s));
  } else {
    JCHK(sampler_steps(sp, s));
  }
  HIPCHK(hipMemcpyAsync(z_out, sp->z, lat, hipMemcpyDeviceToDevice, s));
  return JAT_OK;
}

extern "C" int jat_cfg_euler_step(const float* x_pred_2B, float* z, float cfg_scale, float t, float dt, int32_t B,
                                  int32_t C, int32_t T, void* stream) {
  KCHK(launch_cfg_euler(x_pred_2B, z, cfg_scale, t, dt, cfg_scale != 1.0f ? 1 : 0, (int64_t)B * C * T,
                        (hipStream_t)stream));
  return JAT_OK;
}

extern "C" int jat_channel_affine(const float* in, const float* mean, const float* std, float* out, int32_t B, int32_t C,
                                  int32_t T, int32_t inverse, void* stream) {
  KCHK(launch_channel_affine(in, mean, std, out, B, C, T, inverse, (hipStream_t)stream));
  return JAT_OK;
}
extern "C" int jat_crossfade_pair(const float* prev, int32_t Tp, const float* cur, int32_t Tc, int32_t overlap,
                                  float* out, int32_t rows, void* stream) {
  KCHK(launch_crossfade_pair(prev, Tp, cur, Tc, overlap, out, rows, (hipStream_t)stream));
  return JAT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// per-kernel entry points
// ---------------------------------------------------------------------------------------------------------
extern "C" int jat_k_norm_modulate(const float* x, const float* w, const float* shift, const float* scale,
                                   int64_t mod_bstride, uint16_t* y, int32_t M, int32_t D, int32_t rows_per_batch,
                                   int32_t norm_mode, void* stream) {
  KCHK(launch_norm_modulate(x, w, shift, scale, mod_bstride, y, M, D, rows_per_batch, norm_mode, (hipStream_t)stream));
  return JAT_OK;
}
extern "C" int jat_k_gemm(const uint16_t* A, const uint16_t* W, const float* bias, void* C, int32_t M, int32_t N,
                          int32_t K, int32_t epilogue, const float* gate, int64_t gate_bstride, int32_t rows_per_batch,
                          int32_t variant, void* stream) {
  if (epilogue < 0 || epilogue > EPI_RESID) return fail(JAT_E_INVALID, "epilogue must be 0..3");
  GemmArgs a{};
  a.A = A; a.W = W; a.lda = K; a.ldw = K; a.M = M; a.N = N; a.K = K;
  a.out = C; a.ldo = N; a.bias = bias; a.gate = gate; a.gate_bstride = gate_bstride;
  a.ntok = rows_per_batch > 0 ? rows_per_batch : 1;
  if (const char* d = getenv("JAT_GEMM_DBG")) a.dbg = atoi(d);
  if (const char* d = getenv("JAT_GEMM_TIMELINE")) a.dbg_out = (unsigned long long*)strtoull(d, nullptr, 0);  // tools/gemm_timeline.py
  KCHK(launch_gemm(a, epilogue, variant, (hipStream_t)stream));
  return JAT_OK;
}
// split-K slices of C = A W^T: parts[z][M][N] fp32 = the sum over K columns [z K/ksplit, (z+1) K/ksplit); the caller (or the
// finishing passes of elementwise.hip) adds the slices in order.  The un-folded forward's fc2 / out_proj use it when their
// tiles leave CUs idle (resid_split); variant 39 = the 224 x 160 k-step-pair tile (M % 224 == 0, N % 160 == 0).
extern "C" int jat_k_gemm_splitk(const uint16_t* A, const uint16_t* W, float* parts, int32_t M, int32_t N, int32_t K,
                                 int32_t ksplit, int32_t variant, void* stream) {
  if (!A || !W || !parts || M <= 0 || N <= 0 || ksplit < 2 || K % (64 * ksplit) != 0) return fail(JAT_E_INVALID, "bad argument");
  if (!gemm_variant_exists(variant)) return fail(JAT_E_INVALID, "unknown variant");
  GemmArgs a{};
  a.A = A; a.W = W; a.lda = K; a.ldw = K; a.M = M; a.N = N; a.K = K / ksplit;
  a.out = parts; a.ldo = N; a.ntok = 1; a.ksplit = ksplit; a.split_stride = (int64_t)M * N;
  KCHK(launch_gemm(a, EPI_F32, variant, (hipStream_t)stream));
  return JAT_OK;
}
extern "C" int jat_k_gemm_wave_n(int32_t variant) { return gemm_variant_exists(variant) ? gemm_variant_wave_n(variant) : 0; }
extern "C" int jat_k_gemm_fold(const uint16_t* A, const uint16_t* W, const float* bias, void* C, int32_t M, int32_t N, int32_t K,
                               int32_t epilogue, const float* gate, int64_t gate_bstride, int32_t rows_per_batch, uint16_t* hi,
                               uint16_t* lo, float* part_out, const float* part_in, int32_t part_in_np, int32_t variant,
                               void* stream) {
  if (epilogue < 0 || epilogue > EPI_RESID) return fail(JAT_E_INVALID, "epilogue must be 0..3");
  if (!gemm_variant_exists(variant) || !gemm_variant_coalesced(variant)) return fail(JAT_E_INVALID, "needs a coalesced-epilogue variant");
  if (hi && (epilogue != EPI_F32 && epilogue != EPI_RESID)) return fail(JAT_E_INVALID, "producer epilogue must be 0 or 3");
  if (hi && (!lo || !part_out)) return fail(JAT_E_INVALID, "producer needs hi, lo and part_out");
  if (part_in && part_in_np != 4 && part_in_np != 8 && part_in_np != 16) return fail(JAT_E_INVALID, "part_in_np must be 4, 8 or 16");
  GemmArgs a{};
  a.A = A; a.W = W; a.lda = K; a.ldw = K; a.M = M; a.N = N; a.K = K;
  a.out = C; a.ldo = N; a.bias = bias; a.gate = gate; a.gate_bstride = gate_bstride;
  a.ntok = rows_per_batch > 0 ? rows_per_batch : 1;
  a.fold_out = hi; a.fold_lo = lo; a.fold_part = part_out;
  if (hi) a.fold_np = N / gemm_variant_wave_n(variant);
  a.rs_part = part_in; a.rs_np = part_in_np;
  if (const char* d = getenv("JAT_GEMM_DBG")) a.dbg = atoi(d);
  if (const char* d = getenv("JAT_GEMM_TIMELINE")) a.dbg_out = (unsigned long long*)strtoull(d, nullptr, 0);  // tools/tl_probe.py
  KCHK(launch_gemm(a, epilogue, variant, (hipStream_t)stream));
  return JAT_OK;
}
// per-kernel entry point (unit parity, tools/tl_probe.py): fused QKV projection + RoPE + GQA attention of 128-token samples,
// W = group-major fused weight [Hkv][5*64 + 64 + 64][K] (q / k rows pair-interleaved per head), out = attention output [M, Hkv*320]
extern "C" int jat_k_qkv_attn(const uint16_t* A, const uint16_t* Wg, const float* bias, uint16_t* out, int32_t M, int32_t Hkv,
                              int32_t K, const float* rope_inv_freq, const float* part_in, int32_t part_in_np, void* stream) {
  if (!A || !Wg || !out || !rope_inv_freq || M <= 0 || M % 128 != 0 || K % 64 != 0) return fail(JAT_E_INVALID, "bad argument");
  GemmArgs a{};
  a.A = A; a.lda = K; a.W = Wg; a.ldw = K; a.M = M; a.N = Hkv * 448; a.K = K;
  a.out = out; a.ldo = (int64_t)Hkv * 320; a.ntok = 128; a.rope_inv_freq = rope_inv_freq; a.bias = bias;
  a.attn_scale_log2e = 0.125f * 1.4426950408889634f;
  a.rs_part = part_in; a.rs_np = part_in_np;
  if (const char* d = getenv("JAT_GEMM_DBG")) a.dbg = atoi(d);
  if (const char* d = getenv("JAT_GEMM_TIMELINE")) a.dbg_out = (unsigned long long*)strtoull(d, nullptr, 0);
  KCHK(launch_qkv_attn(a, (hipStream_t)stream));
  return JAT_OK;
}
extern "C" int jat_k_weight_grad(const uint16_t* dY, const uint16_t* X, float* dW, float* db, int32_t tokens, int32_t out,
                                 int32_t in, int32_t ksplit, void* work, size_t work_bytes, void* stream) {
  if (!gemm_tn_supports(out, in)) return fail(JAT_E_INVALID, "out and in must be multiples of 128");
  if (tokens <= 0) return fail(JAT_E_INVALID, "tokens must be positive");
  if (ksplit == 0) ksplit = gemm_tn_ksplit(out, in, tokens);
  if (ksplit < 1 || ksplit > (tokens + 63) / 64) return fail(JAT_E_INVALID, "bad ksplit");
  const int64_t area = (int64_t)out * in;
  const size_t part_f = ksplit > 1 ? (size_t)ksplit * area : 0, col_f = db ? (size_t)colsum_slices(tokens) * out : 0;
  const size_t need = 256 + (part_f + col_f) * 4;
  if (!work || work_bytes < need) return fail(JAT_E_INVALID, "work needs %zu bytes", need);
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(work, 0, 256, s));   // the zero cell ragged token tiles are padded from
  float* part = (float*)((char*)work + 256);
  KCHK(launch_gemm_tn(dY, out, X, in, ksplit > 1 ? part : dW, in, out, in, tokens, ksplit, area, work, s));
  if (ksplit > 1) KCHK(launch_sum_partials(part, ksplit, area, dW, area, s));
  if (db) KCHK(launch_colsum_bf16(dY, out, tokens, out, part + part_f, db, s));
  return JAT_OK;
}
extern "C" int jat_k_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* o, int32_t B,
                               int32_t N, int32_t Hq, int32_t Hkv, int32_t Npad, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.vt = vt; a.o = o; a.ldq = (int64_t)Hq * 64; a.ldk = (int64_t)Hkv * 64; a.ldo = (int64_t)Hq * 64;
  a.B = B; a.N = N; a.Hq = Hq; a.Hkv = Hkv; a.npad = Npad;
  a.scale_log2e = 0.125f * 1.4426950408889634f;
  KCHK(launch_attention(a, (hipStream_t)stream));
  return JAT_OK;
}
extern "C" int jat_prof_gemm_site(jat_model* m, int32_t site, int32_t max_launches) {
  if (!m) return fail(JAT_E_INVALID, "null model");
  if (site < -1 || site > G_OTHER) return fail(JAT_E_INVALID, "site must be -1 (off) or 0..4");
  m->prof.site = site; m->prof.n = 0; m->prof.flops = 0.0; m->prof.variant = -1;
  while ((int)m->prof.ev.size() < 2 * max_launches) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    m->prof.ev.push_back(e);
  }
  return JAT_OK;
}
extern "C" int jat_prof_collect(jat_model* m, double* total_ms, int32_t* launches, double* flops, int32_t* variant) {
  if (!m) return fail(JAT_E_INVALID, "null model");
  double tot = 0.0;
  for (int i = 0; i < m->prof.n; ++i) {
    HIPCHK(hipEventSynchronize(m->prof.ev[2 * i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, m->prof.ev[2 * i], m->prof.ev[2 * i + 1]));
    tot += ms;
  }
  // calibrate: an EMPTY event pair on the same stream measures the marker-to-marker overhead the bracket adds to
  // every launch; subtract it so that the mean agrees with the rocprofv3 kernel-trace duration
  if (m->prof.n > 0 && m->prof.ev.size() >= 2) {
    double ovh = 0.0;
    const int reps = 32;
    for (int i = 0; i < reps; ++i) {
      HIPCHK(hipEventRecord(m->prof.ev[0], m->prof.stream));
      HIPCHK(hipEventRecord(m->prof.ev[1], m->prof.stream));
      HIPCHK(hipEventSynchronize(m->prof.ev[1]));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, m->prof.ev[0], m->prof.ev[1]));
      ovh += ms;
    }
    tot -= ovh / reps * m->prof.n;
    if (tot < 0.0) tot = 0.0;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = m->prof.n;
  if (flops) *flops = m->prof.flops;
  if (variant) *variant = m->prof.variant;
  m->prof.site = -1;
  return JAT_OK;
}
extern "C" int jat_k_cast_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
  KCHK(launch_cast_bf16(in, out, n, (hipStream_t)stream));
  return JAT_OK;
}
