// micro-benchmark (round 6): v_mfma_f64_4x4x4_4b_f64 -- operand / result lane layout and issue rate -- and the LDS side of a
// "one 6x6 block pair per instruction" Schur accumulation: ds_add_f64 of a wave's 64 results, alone and in the full loop
// (operand read from LDS, product with C = 0, atomic add into the workgroup's accumulator image).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_4x4.hip -o /tmp/mfma44 && /tmp/mfma44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_layout(int* outLane, int* outCount) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(c != 0.0);
      if (lane == 0) { outLane[la * 64 + lb] = m ? __builtin_ctzll(m) : -1; outCount[la * 64 + lb] = __builtin_popcountll(m); }
    }
}

template <int NACC>
__global__ void k_rate(double* out, long long* cyc, int n) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  double acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// mode 0: ds_add_f64 only; 1: ds_read_b64 + ds_write_b64 (read-modify-write by the owner); 2: ds_read operand + mfma (C = 0) + ds_add;
// 3: like 2 with 16x16x4 (four results per lane, four ds_add)
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k_lds(double* out, long long* cyc, int n, int nSlots, const double* __restrict__ gops) {
  extern __shared__ double sm[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < nSlots * 64 + 4096; i += blockDim.x) sm[i] = 1e-3 * i;
  __syncthreads();
  const double* ops = sm + nSlots * 64;
  double a = 1.0 + lane * 1e-9, keep = 0;
  (void)gops;
  // (slot numbers from the loop counter: scalar arithmetic only, two instructions per operation)
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int slot = (5 * i + 11 * u + 3 * wave) & (nSlots - 1);
      const int osl = (3 * i + 7 * u + wave) & 63;
      double* dst = sm + slot * 64 + lane;
      if (MODE == 0) {
        atomicAdd(dst, a);
      } else if (MODE == 1) {
        const double c = *dst;
        *dst = c + a;
      } else if (MODE == 2) {
        const double bo = ops[osl * 64 + lane];
        const double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bo, 0.0, 0, 0, 0);
        atomicAdd(dst, c);
      } else if (MODE == 3) {
        const double bo = ops[osl * 64 + lane];
        const d4_t z = {0, 0, 0, 0};
        const d4_t c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bo, z, 0, 0, 0);
        atomicAdd(dst, c[0]);
        atomicAdd(sm + ((slot + 1) & (nSlots - 1)) * 64 + lane, c[1]);
        atomicAdd(sm + ((slot + 2) & (nSlots - 1)) * 64 + lane, c[2]);
        atomicAdd(sm + ((slot + 3) & (nSlots - 1)) * 64 + lane, c[3]);
      } else if (MODE == 5) {   // the Schur kernel's add: 36 valid result lanes, 6 x 6 block of 36 consecutive doubles
        const int kk = lane >> 4, bq = (lane >> 2) & 3, ij = lane & 3;
        const int dRow = 4 * (bq >> 1) + kk, dCol = 4 * (bq & 1) + ij;
        if (dRow < 6 && dCol < 6) atomicAdd(sm + ((5 * i + 11 * u + 3 * wave) & 63) * 36 + dRow * 6 + dCol, a);
      } else if (MODE == 6) {   // the same 36 lanes, lane l -> double l of the block (no bank shared by two lanes beyond the 32nd)
        if (lane < 36) atomicAdd(sm + ((5 * i + 11 * u + 3 * wave) & 63) * 36 + lane, a);
      } else if (MODE == 7) {   // 32 lanes only
        if (lane < 32) atomicAdd(sm + ((5 * i + 11 * u + 3 * wave) & 63) * 36 + lane, a);
      } else if (MODE == 8) {   // the Schur kernel's operand read: 24 distinct doubles of a record + a zero, by 64 lanes
        const int kk = lane >> 4, bq = (lane >> 2) & 3, ij = lane & 3;
        const int off = kk < 3 ? 8 * kk + 4 * (bq >> 1) + ij : 24;
        keep += ops[((3 * i + 7 * u + wave) & 63) * 25 + off];
      } else {   // 4: operand from global memory (L2-resident table), product, ds_add
        const double bo = gops[(size_t)(((5 * i + u) * 37 + 11 * wave + 64 * blockIdx.x) & 16383) * 64 + lane];
        const double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bo, 0.0, 0, 0, 0);
        atomicAdd(dst, c);
      }
    }
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  for (int i = t; i < nSlots * 64; i += blockDim.x) keep += sm[i];
  out[blockIdx.x * blockDim.x + t] = keep;
  if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  int *dLane, *dCount;
  hipMalloc(&dLane, 4096 * 4); hipMalloc(&dCount, 4096 * 4);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dLane, dCount);
  std::vector<int> L(4096), C(4096);
  hipMemcpy(L.data(), dLane, 4096 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(C.data(), dCount, 4096 * 4, hipMemcpyDeviceToHost);
  printf("v_mfma_f64_4x4x4_4b_f64 layout: for A lane la, the B lanes lb that meet it and the D lane that receives a[la] * b[lb]\n");
  for (int la = 0; la < 64; ++la) {
    printf("A %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (L[la * 64 + lb] >= 0) printf("  B%2d->D%2d%s", lb, L[la * 64 + lb], C[la * 64 + lb] == 1 ? "" : "(!)");
    printf("\n");
  }
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
  const int n = 4096;
  double* gops; hipMalloc(&gops, (size_t)16384 * 64 * 8); hipMemset(gops, 0, (size_t)16384 * 64 * 8);
  auto report = [&](const char* what, float ms, long long c, double opsPerWave, int waves, int blocks) {
    printf("%-62s %8.1f cycles per op and wave (%lld cycles, %.3f ms, %d waves x %d blocks)\n", what, (double)c / opsPerWave, c, ms, waves, blocks);
  };
  for (int rep = 0; rep < 2; ++rep) {
    for (int threads : {64, 512, 768, 1024}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms; long long h;
#define RUN(K, label, opsPerIter, blocks, shm, ...) \
      hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), shm, 0, __VA_ARGS__); hipEventRecord(e1); hipEventSynchronize(e1); \
      hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); report(label, ms, h, (double)n * (opsPerIter), threads / 64, blocks);
      RUN((k_rate<1>), "4x4x4_4b: one dependent accumulator chain", 1, 1, 0, out, cyc, n)
      RUN((k_rate<4>), "4x4x4_4b: four independent accumulators", 4, 1, 0, out, cyc, n)
      RUN((k_rate<8>), "4x4x4_4b: eight independent accumulators", 8, 1, 0, out, cyc, n)
      RUN((k_rate<8>), "4x4x4_4b: eight independent accumulators, 256 blocks", 8, 256, 0, out, cyc, n)
      const int nSlots = 64;   // 64 accumulator blocks of 64 doubles: 32 KB
      const size_t shm = (size_t)(nSlots * 64 + 4096) * 8;
      RUN((k_lds<0>), "ds_add_f64, 64 distinct addresses per instruction", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<1>), "ds_read_b64 + add + ds_write_b64", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<2>), "ds_read operand + 4x4x4_4b + ds_add_f64", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<3>), "ds_read operand + 16x16x4 + 4 x ds_add_f64", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<2>), "ds_read operand + 4x4x4_4b + ds_add_f64, 256 blocks", 8, 256, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<5>), "ds_add_f64, 36 result lanes of a 6 x 6 block (the kernel's map)", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<6>), "ds_add_f64, 36 lanes, lane l -> double l", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<7>), "ds_add_f64, 32 lanes, lane l -> double l", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<8>), "ds_read_b64 operand pattern (24 doubles + zero, broadcast)", 8, 1, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<4>), "global (8 MB table) operand + 4x4x4_4b + ds_add_f64, 256 blocks", 8, 256, shm, out, cyc, n, nSlots, gops)
      RUN((k_lds<4>), "global (8 MB table) operand + 4x4x4_4b + ds_add_f64, 512 blocks", 8, 512, shm, out, cyc, n, nSlots, gops)
    }
  }
  return 0;
}
