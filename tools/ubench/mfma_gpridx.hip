// micro-benchmark (round 6): does the VGPR index mode (s_set_gpr_idx_on) reach the C / D operands of v_mfma_f64_4x4x4_4b_f64?
// If it does, a wave that owns 16 (or 32) 6 x 6 accumulator blocks can pick the block of a pair with two scalar instructions
// instead of four indexed moves around the product (k_schur_rows: ~60 cycles per pair and SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_gpridx.hip -o /tmp/gpridx && /tmp/gpridx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d16_t __attribute__((ext_vector_type(16)));

// one product into accumulator `idx2 / 2` of the tuple pinned to v[64:95]; NOPS wait states after it (a dependent product may follow)
#define MFMA_IDX(ACC, A, B, IDX2)                                                                                          \
  asm volatile("s_set_gpr_idx_on %3, 0xc\n\t"                                                                              \
               "v_mfma_f64_4x4x4_4b_f64 v[64:65], %1, %2, v[64:65]\n\t"                                                    \
               "s_set_gpr_idx_off\n\t"                                                                                     \
               "s_nop 4"                                                                                                   \
               : "+{v[64:95]}"(ACC)                                                                                        \
               : "v"(A), "v"(B), "s"(IDX2)                                                                                 \
               : "m0")

__global__ void k_check(const int* seq, int n, double* out) {
  const int lane = threadIdx.x & 63;
  const double a = (lane >> 4) == 0 ? 1.0 : 0.0, b = a;   // k = 0 only: every product adds 1 to all 64 result lanes
  d16_t acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int idx2 = __builtin_amdgcn_readfirstlane(2 * seq[i]);
    MFMA_IDX(acc, a, b, idx2);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int k = 0; k < 16; ++k) out[k * 64 + lane] = acc[k];
}

template <int MODE>
__global__ void k_rate(double* out, long long* cyc, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  d16_t acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pb = __builtin_amdgcn_readfirstlane((5 * i + 7 * u + 3 * wave) & 15);
      if (MODE == 0) {
        MFMA_IDX(acc, a, b, 2 * pb);
      } else {   // what the compiler makes of a runtime index (indexed moves around the product)
        acc[pb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[pb], 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  double s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int n = 4096;
  std::vector<int> seq(n);
  std::vector<double> expect(16, 0.0);
  unsigned r = 12345u;
  for (int i = 0; i < n; ++i) {
    r = r * 1664525u + 1013904223u;
    seq[i] = (i % 7 == 0 && i) ? seq[i - 1] : (int)((r >> 16) & 15u);   // (some back-to-back products into the same accumulator)
    expect[seq[i]] += 1.0;
  }
  int* dSeq; double* dOut; long long* dCyc;
  hipMalloc(&dSeq, n * sizeof(int)); hipMalloc(&dOut, 1 << 20); hipMalloc(&dCyc, 64);
  hipMemcpy(dSeq, seq.data(), n * sizeof(int), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dSeq, n, dOut);
  std::vector<double> out(16 * 64);
  hipMemcpy(out.data(), dOut, out.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int k = 0; k < 16; ++k)
    for (int l = 0; l < 64; ++l)
      if (out[k * 64 + l] != expect[k]) { if (bad < 8) printf("acc %d lane %d: %g, expected %g\n", k, l, out[k * 64 + l], expect[k]); ++bad; }
  printf("index mode on the C / D operands of v_mfma_f64_4x4x4_4b_f64: %s (%d mismatches)\n", bad ? "DOES NOT WORK" : "works", bad);
  for (int waves : {1, 2, 4, 8, 16}) {
    long long c0 = 0, c1 = 0;
    const int iters = 2000;
    hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64 * waves), 0, 0, dOut, dCyc, iters);
    hipMemcpy(&c0, dCyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64 * waves), 0, 0, dOut, dCyc, iters);
    hipMemcpy(&c1, dCyc, 8, hipMemcpyDeviceToHost);
    printf("%2d waves on one CU: indexed product %.1f cycles per pair and wave (%.1f per pair and CU); indexed moves %.1f (%.1f)\n", waves,
           (double)c0 / (4.0 * iters), (double)c0 / (4.0 * iters * waves), (double)c1 / (4.0 * iters), (double)c1 / (4.0 * iters * waves));
  }
  return 0;
}
