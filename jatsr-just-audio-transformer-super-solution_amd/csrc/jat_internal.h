// Internals shared by the C-ABI translation units (jat_api.cpp: model / forward / sampler; jat_train.cpp: training step).
#pragma once
#include "../../include/jat_hip.h"

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "jat_kernels.h"

int jat_fail(int code, const char* fmt, ...);
#define fail jat_fail
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) return fail(JAT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                       __FILE__, __LINE__);                                            \
  } while (0)
#define JCHK(expr)            \
  do {                        \
    int r__ = (expr);         \
    if (r__ != JAT_OK) return r__; \
  } while (0)
#define KCHK(expr)                                                                            \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) return fail(JAT_E_HIP, "%s: %s", #expr, hipGetErrorString(e__));   \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static constexpr int MAX_LEN = 2048;  // jat_audiosr_v3.py:361
static constexpr int HEAD_DIM = 64;

struct LayerW {
  bf16_t *wqkv, *wo, *w1, *w2;  // [D+2kvD, D], [D, D], [mlp, D], [D, mlp]
  bf16_t* wqkv_g = nullptr;     // group-major copy [Hkv][5*64 + 64 + 64][D] for the fused QKV+attention kernel (Hq/Hkv == 5)
  float *norm1, *norm2, *b1, *b2;
  float *q32 = nullptr, *k32 = nullptr, *v32 = nullptr, *w132 = nullptr;   // fp32 copies: sources of the sampler's folded weights
};

// Per-step folded weights of the sampler (RMSNorm models): W' = W diag(w_norm (1 + scale)) and shift @ W^T for every
// (step, layer) of a fixed schedule — input independent, built once per (model weights, steps) and shared by every sampler
// bucket (B, T) with that step count.  ~25 GB for v3mod2 at 50 steps: HBM capacity (288 GB) spent to delete the two norm
// kernels of every block from the captured graph.
struct FoldTable {
  int steps = 0;
  bf16_t *qkv_g = nullptr, *qkv_i = nullptr;   // [steps][depth][D+2kvD][D]: group-major (fused QKV+attention) / pair-interleaved
  bf16_t *w1 = nullptr, *wfinal = nullptr;     // [steps][depth][mlp][D], [Fout][D]
  float *bq_g = nullptr, *bq_i = nullptr, *bf = nullptr;   // [steps][depth][D+2kvD] x2, [steps][depth][mlp]
  std::vector<void*> allocs;
  ~FoldTable() { for (void* p : allocs) (void)hipFree(p); }
};

// Behaviour switches of one model handle: defaults from the JAT_* environment variables, read ONCE in jat_model_create (nothing on the
// per-call enqueue path touches the environment), changed per handle with jat_model_set_switch.
struct jat_switches {
  int fuse_qkv_attn = 1;   // JAT_FUSE_QKV_ATTN: 0 separate QKV GEMM + attention, 1 fused when B * Hkv blocks fill the chip, 2 always (128 tokens)
  int qkv_split = 1;       // JAT_QKV_SPLIT:     K-slices for the QKV GEMM of small buckets
  int fuse_finish = 1;     // JAT_FUSE_FINISH:   split-K finishing pass fused with the norm that follows it
  int fold_norm = 1;       // JAT_FOLD_NORM:     sampler norm folding (0 off, 1 buckets above kSplitMaxRows, 2 every bucket)
  int split_patch = 1;     // JAT_SPLIT_PATCH:   CFG sampler: condition half of the first patch-embed Linear computed once per run
  int gemm_dbg = 0;        // JAT_GEMM_DBG:      timing aids of gemm.hip (wrong results); 0 in production
  int fold_cap_mb = 0;     // JAT_FOLD_CAP_MB:   upper bound on the folded-weight table (0 = none); beyond it the sampler keeps the norm kernels
  int patch_split = 1;     // JAT_PATCH_SPLIT:   first patch-embed Linear as K slices + bias/GELU finishing pass when its tiles leave CU slots empty
};
// measurement aid (bench.py roofline leg): HIP-event brackets around the launches of one GEMM call site of THIS model
struct GemmProf {
  int site = -1, n = 0, variant = -1;
  double flops = 0.0;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;
};

struct jat_model {
  jat_config cfg;
  jat_switches sw;
  mutable GemmProf prof;
  int D, depth, Hq, Hkv, kvD, mlp, bott, Cin, Cc, P, Kp, Fout;
  bool loaded = false;
  char* blob = nullptr;  // one device allocation holding every packed tensor
  size_t blob_bytes = 0;
  bf16_t *pe_w1, *pe_w2, *wada, *wfinal;
  float *pe_b1, *pe_b2, *te_w1, *te_b1, *te_w2, *te_b2, *bada, *final_norm, *bfinal, *rope_cos, *rope_sin, *rope_invf;
  std::vector<LayerW> layers;
  // GEMM tile/pipeline variant per call site: qkv, out_proj, fc1, fc2, everything else (gemm.hip table)
  int variants[5] = {-1, -1, -1, -1, -1};  // -1: choose by shape (pick_variant)
  mutable int last_fold_np = 0;            // partial-sum slots per row written by the latest folding producer
  float* wfinal32 = nullptr;               // fp32 copy of final_layer.1.weight (fold source)
  bool fold_src_ok = false;                // fp32 copies match the packed weights (false after a training re-pack)
  std::map<int, std::shared_ptr<FoldTable>> fold_cache;   // by step count; dropped whenever the weights change
  bool group_copy_stale = false;           // the training re-pack skips wqkv_g: the fused QKV+attention kernel is off until
                                           // the next full jat_model_load_weights
};
enum { G_QKV = 0, G_OUT = 1, G_FC1 = 2, G_FC2 = 3, G_OTHER = 4 };

// C[M,N] = A[M,K] W[N,K]^T through the variant chooser / profiling bracket of jat_api.cpp
int jat_gemm(const jat_model* m, int site, const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, int M, int N, int K,
             int epi, GemmArgs extra, hipStream_t s);
// (re)pack the bf16 / fp32 device copies of the model from named fp32 tensors; sync_tables: also (re)build the RoPE tables
int jat_pack_weights(jat_model* m, const jat_tensor_ref* named, int32_t n, hipStream_t s, bool build_tables);
