// Counter-based dropout masks for the training step (jat_audiosr_v3.py:38-64,139,175,269-271).
// A mask element is a pure function of (step seed, site, element index): the backward recomputes it, nothing is stored.
// The test suite carries a numpy mirror of this file (bit-exact integer arithmetic) to rebuild the masks on the CPU.
#pragma once
#include <stdint.h>

struct DropSpec {
  uint32_t k0, k1;    // site keys derived from the step seed on the host
  uint32_t thresh;    // element is dropped when its 32-bit draw is < thresh (= p * 2^32); 0 disables the site
  float inv_keep;     // 1 / (1 - p)
};

#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t jat_hash32(uint32_t x) {   // "lowbias32" integer finaliser
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline float jat_drop_mult(const DropSpec d, uint64_t idx) {
  // one finaliser round per element (two quarter-rate 32-bit multiplies): the site keys enter before (k0, together with
  // the rotated high half of the index) and after (k1) the mixing
  const uint32_t hi = (uint32_t)(idx >> 32);
  const uint32_t r = jat_hash32((uint32_t)idx ^ d.k0 ^ ((hi << 13) | (hi >> 19))) ^ d.k1;
  return r < d.thresh ? 0.0f : d.inv_keep;
}
// the same draw for an element index known to fit 32 bits (hi = 0 above): no 64-bit index arithmetic per element
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline float jat_drop_mult32(const DropSpec d, uint32_t idx) {
  return (jat_hash32(idx ^ d.k0) ^ d.k1) < d.thresh ? 0.0f : d.inv_keep;
}
// site = layer * 8 + kind;  kind: 0 attention probabilities, 1 DropPath(attention branch), 2 MLP after GELU,
// 3 MLP output, 4 DropPath(MLP branch)
static inline DropSpec jat_drop_spec(uint64_t seed, uint32_t site, float p) {
  DropSpec d;
  d.k0 = jat_hash32((uint32_t)seed ^ (site * 0x9e3779b9U));
  d.k1 = jat_hash32((uint32_t)(seed >> 32) + site * 0x85ebca6bU + 1U);
  if (p <= 0.f) { d.thresh = 0; d.inv_keep = 1.0f; return d; }
  const double t = (double)p * 4294967296.0;
  d.thresh = t >= 4294967295.0 ? 4294967295U : (uint32_t)t;
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}
