// Host cost of issuing one batched round: a small pinned->device copy and seven short kernels, as individual launches and as
// one hipGraphLaunch (tools/ubench; hipcc --offload-arch=gfx950 -O2 graph_launch.hip -o /tmp/graph_launch)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(const int* in, int* out, int n) {
  int v = in[0];
  for (int i = 0; i < n; ++i) v = v * 3 + 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s;
  OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int *d, *h;
  OK(hipMalloc(&d, 8192));
  OK(hipHostMalloc(&h, 8192, hipHostMallocDefault));
  const int rounds = 2000;
  auto issue = [&]() {
    (void)hipMemcpyAsync(d, h, 6528, hipMemcpyHostToDevice, s);
    for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, d, d + 1024, 10);
  };
  for (int i = 0; i < 20; ++i) issue();
  OK(hipStreamSynchronize(s));
  double t0 = now();
  for (int i = 0; i < rounds; ++i) issue();
  double t1 = now();
  OK(hipStreamSynchronize(s));
  double t2 = now();
  printf("individual: issue %.2f us per round (8 submissions), drained after %.2f us per round\n", 1e6 * (t1 - t0) / rounds, 1e6 * (t2 - t0) / rounds);
  // kernels only
  t0 = now();
  for (int i = 0; i < rounds; ++i) for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, d, d + 1024, 10);
  t1 = now();
  OK(hipStreamSynchronize(s));
  t2 = now();
  printf("kernels only: issue %.2f us per round, drained %.2f\n", 1e6 * (t1 - t0) / rounds, 1e6 * (t2 - t0) / rounds);
  hipGraph_t g; hipGraphExec_t ge;
  OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  issue();
  OK(hipStreamEndCapture(s, &g));
  OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) OK(hipGraphLaunch(ge, s));
  OK(hipStreamSynchronize(s));
  t0 = now();
  for (int i = 0; i < rounds; ++i) (void)hipGraphLaunch(ge, s);
  t1 = now();
  OK(hipStreamSynchronize(s));
  t2 = now();
  printf("graph: issue %.2f us per round, drained after %.2f us per round\n", 1e6 * (t1 - t0) / rounds, 1e6 * (t2 - t0) / rounds);
  // latency: one round, then wait
  t0 = now();
  for (int i = 0; i < 200; ++i) { issue(); (void)hipStreamSynchronize(s); }
  t1 = now();
  for (int i = 0; i < 200; ++i) { (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s); }
  t2 = now();
  printf("round trip: individual %.2f us, graph %.2f us\n", 1e6 * (t1 - t0) / 200, 1e6 * (t2 - t1) / 200);
  return 0;
}
