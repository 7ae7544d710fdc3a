// Probe: how many bytes per clock can one CU pull from L2/MALL into LDS with global_load_lds_dwordx4,
// as a function of the number of 1-KiB pieces kept in flight per wave?  (8 waves per block, 1 block per CU)
// Access pattern = the GEMM's: a piece is 8 rows x 128 B of a K-contiguous bf16 matrix, walking along K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int INFLIGHT, int PER_STEP>
__global__ void __launch_bounds__(512) probe(const unsigned short* A, int lda, int nk, int rows_total, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  // every wave owns PER_STEP pieces per k-step: rows block*... spread like a 256+160 panel
  const unsigned short* src[PER_STEP];
  for (int j = 0; j < PER_STEP; ++j) {
    int r = ((blockIdx.x * 8 + wave) * PER_STEP + j) * 8 + srow;
    r %= rows_total;
    src[j] = A + (long)r * lda + schunk * 8;
  }
  constexpr int RING = INFLIGHT / PER_STEP + 1;  // ring slots (each = PER_STEP KiB per wave)
  char* base = smem + wave * (RING * PER_STEP * 1024);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int j = 0; j < PER_STEP; ++j)
      __builtin_amdgcn_global_load_lds((const void*)(src[j] + kt * 64), (lds_ptr_t)(base + (slot * PER_STEP + j) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT - PER_STEP) : "memory");
    slot = slot + 1 == RING ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int INFLIGHT, int PER_STEP>
void run(const unsigned short* A, int lda, int nk, int rows, unsigned long long* dcyc, const char* tag) {
  constexpr int RING = INFLIGHT / PER_STEP + 1;
  const int lds = 8 * RING * PER_STEP * 1024;
  if (lds > 160 * 1024) { printf("%s: skip (LDS %d)\n", tag, lds); return; }
  hipFuncSetAttribute((const void*)probe<INFLIGHT, PER_STEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<INFLIGHT, PER_STEP>), dim3(256), dim3(512), lds, 0, A, lda, nk, rows, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= 256;
  const double bytes_per_cu = 8.0 * PER_STEP * 1024 * nk;
  printf("%s inflight/wave=%2d KiB per_step=%d : %.1f us, %.0f cyc (memtime @100MHz ticks=%.0f) -> %.1f GB/s per CU, chip %.2f TB/s\n", tag, INFLIGHT,
         PER_STEP, ms * 1e3, avg, avg, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
  const int rows = 7168, K = 5120, nk = K / 64;
  unsigned short* A; hipMalloc(&A, (size_t)rows * K * 2); hipMemset(A, 1, (size_t)rows * K * 2);
  unsigned long long* dcyc; hipMalloc(&dcyc, 256 * 8);
  run<7, 7>(A, K, nk, rows, dcyc, "K=5120");
  run<14, 7>(A, K, nk, rows, dcyc, "K=5120");
  run<4, 2>(A, K, nk, rows, dcyc, "K=5120");
  run<8, 2>(A, K, nk, rows, dcyc, "K=5120");
  run<16, 2>(A, K, nk, rows, dcyc, "K=5120");
  run<8, 4>(A, K, nk, rows, dcyc, "K=5120");
  run<16, 4>(A, K, nk, rows, dcyc, "K=5120");
  // small panel (L2-resident): 1024 rows reused by all blocks
  run<14, 7>(A, K, nk, 1024, dcyc, "rows=1024 (L2-resident)");
  run<16, 4>(A, K, nk, 1024, dcyc, "rows=1024 (L2-resident)");
  return 0;
}
