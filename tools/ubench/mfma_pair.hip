// micro-benchmark: dependent v_mfma_f64_16x16x4_f64 accumulator chains when 1, 2 (one per SIMD pair) ... 8 waves of ONE
// workgroup run them at the same time (2 waves share a SIMD's matrix pipe from 5 waves on), with and without an LDS operand read
// per product -- the shape of the trailing update of k_chol_solve_lds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n, int activeMask, int useLds) {
  __shared__ double lds[8 * 16 * 17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 16 * 17; i += 512) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  d4_t acc = {0, 0, 0, 0};
  const int lop = (lane & 15) * 17 + (lane >> 4);
  long long t0 = __builtin_readcyclecounter();
  if ((activeMask >> wave) & 1) {
    const double* X = lds + wave * 272;
    for (int i = 0; i < n; ++i) {
      double x0 = a, x1 = a, x2 = a, x3 = a;
      if (useLds) { x0 = X[lop]; x1 = X[lop + 4]; x2 = X[lop + 8]; x3 = X[lop + 12]; }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b, acc, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = acc[0] + acc[3];
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 64);
  const int n = 2048;
  const int masks[] = {0x01, 0x11, 0x0f, 0xff, 0xee, 0xfe};
  for (int useLds = 0; useLds < 2; ++useLds)
    for (int m : masks) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, n, m, useLds);
      hipDeviceSynchronize();
      long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      printf("lds %d mask %02x: cycles per 4-product tile:", useLds, m);
      for (int w = 0; w < 8; ++w) if ((m >> w) & 1) printf(" w%d %.0f", w, (double)h[w] / n);
      printf("\n");
    }
  return 0;
}
