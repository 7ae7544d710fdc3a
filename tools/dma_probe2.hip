// Probe 2: is the ~22 cycles per 1-KiB piece of tools/dma_probe.hip a per-CU limit (TA / L1 path) or a shared
// per-XCD L2 limit?  Same access pattern, but (a) the number of active CUs is varied (8 = one per XCD ... 256),
// (b) global_load_lds_dwordx4 is compared with plain global_load_dwordx4 into registers, (c) an L1-resident
// pattern (every k-step re-reads the same 8 KiB per wave) isolates the TA/L1 issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe2 dma_probe2.hip && ./dma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PER_STEP, int DEPTH>   // MODE 0 = glds, 1 = plain loads to VGPRs, 2 = glds L1-resident
__global__ void __launch_bounds__(512) probe(const unsigned short* A, int lda, int nk, int rows_total, unsigned long long* cyc, u32x4* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned short* src[PER_STEP];
  for (int j = 0; j < PER_STEP; ++j) {
    int r = ((blockIdx.x * 8 + wave) * PER_STEP + j) * 8 + srow;
    r %= rows_total;
    src[j] = A + (long)r * lda + schunk * 8;
  }
  constexpr int RING = DEPTH + 1;
  char* base = smem + wave * (RING * PER_STEP * 1024);
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ko = (MODE == 2 ? (kt & 1) : kt) * 64;
    if (MODE == 1) {
      u32x4 v[PER_STEP];
#pragma unroll
      for (int j = 0; j < PER_STEP; ++j) v[j] = __builtin_nontemporal_load((const u32x4*)(src[j] + ko));
#pragma unroll
      for (int j = 0; j < PER_STEP; ++j) acc ^= v[j];
    } else {
#pragma unroll
      for (int j = 0; j < PER_STEP; ++j)
        __builtin_amdgcn_global_load_lds((const void*)(src[j] + ko), (lds_ptr_t)(base + (slot * PER_STEP + j) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER_STEP) : "memory");
      slot = slot + 1 == RING ? 0 : slot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE == 1 && acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int MODE, int PER_STEP, int DEPTH>
void run(const unsigned short* A, int lda, int nk, int rows, int nblocks, unsigned long long* dcyc, u32x4* sink, const char* tag) {
  constexpr int RING = DEPTH + 1;
  const int lds = MODE == 1 ? 1024 : 8 * RING * PER_STEP * 1024;
  hipFuncSetAttribute((const void*)probe<MODE, PER_STEP, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, PER_STEP, DEPTH>), dim3(nblocks), dim3(512), lds, 0, A, lda, nk, rows, dcyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(nblocks);
  hipMemcpy(h.data(), dcyc, nblocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= nblocks;   // s_memtime ticks at 100 MHz
  const double bytes_per_cu = 8.0 * PER_STEP * 1024 * nk;
  const double us_in = avg / 100.0;                              // in-kernel time of the streaming loop
  printf("%-22s mode=%d blocks=%3d per_step=%d depth=%d : %.1f us wall, %.1f us in-loop -> %.1f GB/s per CU (in-loop), chip %.2f TB/s\n",
         tag, MODE, nblocks, PER_STEP, DEPTH, best * 1e3, us_in, bytes_per_cu / (us_in * 1e-6) / 1e9,
         bytes_per_cu * nblocks / (us_in * 1e-6) / 1e12);
}

int main() {
  const int rows = 7168, K = 5120, nk = K / 64;
  unsigned short* A; hipMalloc(&A, (size_t)rows * K * 2); hipMemset(A, 1, (size_t)rows * K * 2);
  unsigned long long* dcyc; hipMalloc(&dcyc, 256 * 8);
  u32x4* sink; hipMalloc(&sink, 512 * 16);
  const int grids[] = {8, 32, 64, 128, 256};
  for (int g : grids) run<0, 4, 3>(A, K, nk, rows, g, dcyc, sink, "glds L2/MALL");
  for (int g : grids) run<0, 4, 3>(A, K, nk, 1024, g, dcyc, sink, "glds rows=1024");
  for (int g : grids) run<1, 4, 4>(A, K, nk, rows, g, dcyc, sink, "plain L2/MALL");
  for (int g : grids) run<1, 8, 4>(A, K, nk, rows, g, dcyc, sink, "plain x8");
  for (int g : grids) run<2, 4, 3>(A, K, nk, rows, g, dcyc, sink, "glds L1-resident");
  run<0, 2, 8>(A, K, nk, rows, 256, dcyc, sink, "glds deep");
  run<0, 6, 2>(A, K, nk, rows, 256, dcyc, sink, "glds 6/step");
  return 0;
}
