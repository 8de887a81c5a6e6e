// micro-benchmark: cost of handing a 64x64 f64 block (32 KB) from one workgroup to another inside one launch on
// gfx950 -- what the one-launch tile Cholesky pays per dependency on its critical path.
//   mode 0: agent-scope relaxed atomic (sc1) loads / stores of the data, workgroup-scope fences, relaxed flag
//   mode 1: plain loads / stores, __threadfence() (buffer_wbl2) before the flag, agent acquire fence (buffer_inv) after
// Two workgroups (placed far apart in the grid so that they land on different XCDs) play ping-pong n times.
// Also times the pieces in isolation inside one workgroup: block load, block store, 64 MFMAs, one flag round trip.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__device__ void putBlock(double* g, const double* lds) {
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
    if (MODE == 0) __hip_atomic_store(g + e, lds[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else g[e] = lds[e];
  }
}
template <int MODE>
__device__ void getBlock(const double* g, double* lds) {
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
    if (MODE == 0) lds[e] = __hip_atomic_load(g + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else lds[e] = g[e];
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void pingpong(double* buf, int* flags, int n, int partner, long long* cyc) {
  __shared__ double lds[4096];
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
  if (me < 0) return;
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) lds[e] = e;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < n; ++it) {
    // round 2*it: 0 -> 1 ; round 2*it+1: 1 -> 0
    for (int half = 0; half < 2; ++half) {
      const int round = 2 * it + half;
      if (me == half) {  // sender
        putBlock<MODE>(buf + (size_t)(round & 1) * 4096, lds);
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + round, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {           // receiver
        if (threadIdx.x == 0) {
          int spins = 0;
          while (__hip_atomic_load(flags + round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        getBlock<MODE>(buf + (size_t)(round & 1) * 4096, lds);
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0 && me == 0) cyc[0] = wall_clock64() - t0;
}
template <int MODE>
__global__ __launch_bounds__(256) void pieces(double* buf, int* flags, int n, long long* cyc) {
  __shared__ double lds[4096];
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) lds[e] = e;
  __syncthreads();
  long long t0 = wall_clock64();
  for (int it = 0; it < n; ++it) { getBlock<MODE>(buf + (size_t)(it & 7) * 4096, lds); __syncthreads(); }
  long long t1 = wall_clock64();
  for (int it = 0; it < n; ++it) {
    putBlock<MODE>(buf + (size_t)(it & 7) * 4096, lds);
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __threadfence();
    __syncthreads();
  }
  long long t2 = wall_clock64();
  for (int it = 0; it < n; ++it) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(flags + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flags + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
    }
    __syncthreads();
  }
  long long t3 = wall_clock64();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}

int main() {
  double* buf; int* flags; long long* cyc;
  const int n = 200;
  OK(hipMalloc(&buf, 8 * 4096 * sizeof(double)));
  OK(hipMalloc(&flags, 4 * n * sizeof(int)));
  OK(hipMalloc(&cyc, 8 * sizeof(long long)));
  long long h[8];
  const double tick = 1e6 / 100e6;  // wall_clock64: 100 MHz -> us per tick
  for (int mode = 0; mode < 2; ++mode) {
    for (int partner : {1, 2, 7, 8, 9}) {
      OK(hipMemset(flags, 0, 4 * n * sizeof(int)));
      OK(hipMemset(buf, 0, 8 * 4096 * sizeof(double)));
      if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(256), 0, 0, buf, flags, n, partner, cyc);
      else hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(256), 0, 0, buf, flags, n, partner, cyc);
      OK(hipDeviceSynchronize());
      OK(hipMemcpy(h, cyc, sizeof(long long), hipMemcpyDeviceToHost));
      printf("mode %d (%s) partner workgroup %d: %.2f us per one-way hand-over of a 32 KB block\n", mode,
             mode == 0 ? "sc1 atomics" : "plain + wbl2/inv", partner, h[0] * tick / (2.0 * n));
    }
    OK(hipMemset(flags, 0, 4 * n * sizeof(int)));
    if (mode == 0) hipLaunchKernelGGL(pieces<0>, dim3(1), dim3(256), 0, 0, buf, flags, n, cyc);
    else hipLaunchKernelGGL(pieces<1>, dim3(1), dim3(256), 0, 0, buf, flags, n, cyc);
    OK(hipDeviceSynchronize());
    OK(hipMemcpy(h, cyc, 3 * sizeof(long long), hipMemcpyDeviceToHost));
    printf("mode %d pieces: block load %.2f us, block store + completion %.2f us, flag store+load by one thread %.2f us\n", mode,
           h[0] * tick / n, h[1] * tick / n, h[2] * tick / n);
  }
  return 0;
}
