// micro-benchmark: N workgroups each ADD a private d x d tile set (a Schur slab) into ONE shared matrix with global f64 atomics
// (device scope) vs writing private slabs -- is "atomics instead of slabs + a reduction kernel" an option for k_schur_dense?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_atomic(double* S, int n) {
  for (int i = threadIdx.x; i < n; i += 256) atomicAdd(&S[i], 1.0 + 1e-9 * blockIdx.x);
}
__global__ __launch_bounds__(256) void k_slab(double* slabs, int n) {
  double* s = slabs + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) s[i] = 1.0 + 1e-9 * blockIdx.x;
}
__global__ __launch_bounds__(256) void k_reduce(const double* slabs, double* S, int n, int nSlabs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a = 0;
  for (int k = 0; k < nSlabs; ++k) a += slabs[(size_t)k * n + i];
  S[i] = a;
}
int main() {
  const int cfg[][2] = {{136, 150 * 150 + 3 * 150}, {136, 60 * 60 + 3 * 60}, {250, 270 * 270 + 3 * 270}, {250, 180 * 180}};
  for (auto& c : cfg) {
    const int nB = c[0], n = c[1];
    double *S, *slabs;
    hipMalloc(&S, n * 8); hipMalloc(&slabs, (size_t)nB * n * 8);
    hipMemset(S, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_atomic, dim3(nB), dim3(256), 0, 0, S, n);
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_atomic, dim3(nB), dim3(256), 0, 0, S, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const float atomicUs = ms * 50;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k_slab, dim3(nB), dim3(256), 0, 0, slabs, n); hipLaunchKernelGGL(k_reduce, dim3((n + 255) / 256), dim3(256), 0, 0, slabs, S, n, nB); }
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) { hipLaunchKernelGGL(k_slab, dim3(nB), dim3(256), 0, 0, slabs, n); hipLaunchKernelGGL(k_reduce, dim3((n + 255) / 256), dim3(256), 0, 0, slabs, S, n, nB); }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%d workgroups x %d doubles: atomics into one matrix %.1f us per launch; private slabs + reduction %.1f us per pair of launches\n", nB, n, atomicUs, ms * 50);
    hipFree(S); hipFree(slabs);
  }
  return 0;
}
