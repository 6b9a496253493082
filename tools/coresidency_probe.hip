// coresidency_probe.hip -- experiment (not product code): can a small kernel on a second stream start while a
// persistent one-workgroup-per-CU kernel with a large LDS allocation occupies every CU?
//   hipcc -O3 --offload-arch=gfx950 tools/coresidency_probe.hip -o gpurun_out/coresidency_probe
// For each configuration of the big kernel (threads, LDS bytes, grid) prints when a chain of 20 small kernels
// (threads/regs configurable) launched 0.2 ms after it finished, relative to the big kernel's start and end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NREG>
__global__ void __launch_bounds__(768) big(double *out, long long cycles, int lds_doubles)
{
  extern __shared__ double sm[];
  double r[NREG];
  for (int q = 0; q < NREG; q++) r[q] = threadIdx.x * 1e-3 + q;
  for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x) sm[i] = i;
  __syncthreads();
  long long t0 = wall_clock64();
  int it = 0;
  while (wall_clock64() - t0 < cycles) {
    const int base = (threadIdx.x + it) % (lds_doubles - NREG);
#pragma unroll
    for (int q = 0; q < NREG; q++) r[q] = r[q] * 1.0000001 + sm[base + q];
    it++;
  }
  double s = 0;
  for (int q = 0; q < NREG; q++) s += r[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NREG>
__global__ void small(double *out, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double r[NREG];
  for (int q = 0; q < NREG; q++) r[q] = i * 1e-3 + q;
  for (int it = 0; it < 200; it++)
#pragma unroll
    for (int q = 0; q < NREG; q++) r[q] = r[q] * 1.0000001 + r[(q + 1) % NREG];
  double s = 0;
  for (int q = 0; q < NREG; q++) s += r[q];
  if (i < n) out[i] = s;
}

int main()
{
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  double *o1, *o2; CK(hipMalloc(&o1, 1 << 24)); CK(hipMalloc(&o2, 1 << 24));
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  CK(hipFuncSetAttribute((const void *) big<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *) big<70>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  long long hz = 100000000;   // wall_clock64 = 100 MHz
  struct Cfg { int threads, lds, grid, regs; };
  std::vector<Cfg> cfgs = {{704, 137000, 256, 70}, {704, 137000, 240, 70}, {704, 137000, 224, 70}, {704, 137000, 192, 70},
                           {704, 64000, 256, 70},  {704, 137000, 256, 32}, {256, 137000, 256, 70}, {704, 1024, 256, 70},
                           {640, 137000, 256, 70}, {512, 137000, 256, 70}, {704, 100000, 256, 32}, {704, 1024, 256, 32}};
  for (int sm_threads : {64, 256}) for (auto c : cfgs) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s1));
    if (c.regs == 70) hipLaunchKernelGGL(big<70>, dim3(c.grid), dim3(c.threads), c.lds, s1, o1, 2 * hz / 1000, c.lds / 8);
    else hipLaunchKernelGGL(big<12>, dim3(c.grid), dim3(c.threads), c.lds, s1, o1, 2 * hz / 1000, c.lds / 8);
    CK(hipEventRecord(e1, s1));
    CK(hipEventRecord(f0, s2));
    for (int k = 0; k < 20; k++) hipLaunchKernelGGL(small<12>, dim3(60000 / sm_threads), dim3(sm_threads), 0, s2, o2, 60000);
    CK(hipEventRecord(f1, s2));
    CK(hipDeviceSynchronize());
    float big_ms, small_end, small_start;
    CK(hipEventElapsedTime(&big_ms, e0, e1)); CK(hipEventElapsedTime(&small_end, e0, f1)); CK(hipEventElapsedTime(&small_start, e0, f0));
    printf("small wg %3d | big: threads %3d lds %6d grid %3d regs~%d | big %.3f ms, small chain %.3f -> %.3f ms  %s\n", sm_threads,
           c.threads, c.lds, c.grid, c.regs * 2, big_ms, small_start, small_end, small_end < 0.8 * big_ms ? "OVERLAPPED" : "blocked");
  }
  return 0;
}
