// lds_bank_probe.hip -- experiment (not product code): cycles per ds_read_b64 / ds_read_b128 of one wave for chosen
// lane -> address patterns, to learn how gfx950 groups the 64 lanes into passes and what counts as a bank conflict.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_bank_probe.hip -o gpurun_out/lds_bank_probe && gpurun_out/lds_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP 256

// idx[lane] in units of 8 bytes (b64) or 16 bytes (b128); waves = co-resident waves on the CU issuing the same pattern
__global__ void k_b64(const int *idx, long long *out, double *sink)
{
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int a = idx[threadIdx.x & 63];
  double acc = 0;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = lds[(a + u * 64 * 0 + r * 0) & 8191];   // same addresses every time: isolates the bank pattern
#pragma unroll
    for (int u = 0; u < 8; u++) asm volatile("" :: "v"(v[u]));           // consumed, no arithmetic: the LDS pipe is the only work
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 12345.678) sink[0] = acc;
}
__global__ void k_b128(const int *idx, long long *out, double *sink)
{
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int a = idx[threadIdx.x & 63];
  const double2 *l2 = (const double2 *) lds;
  double acc = 0;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = l2[a & 4095];
#pragma unroll
    for (int u = 0; u < 8; u++) asm volatile("" :: "v"(v[u].x), "v"(v[u].y));
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 12345.678) sink[0] = acc;
}

// no-return fp64 atomics (ds_add_f64) of one wave: what an in-tile Newton-3 scatter would cost
__global__ void k_atom(const int *idx, long long *out, double *sink)
{
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int a = idx[threadIdx.x & 63];
  const double v = 1.0 + threadIdx.x;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int u = 0; u < 8; u++) unsafeAtomicAdd(&lds[a & 8191], v);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  __syncthreads();
  if (lds[threadIdx.x] == 12345.678) sink[0] = 1;
}
static int run_atom(const char *name, const std::vector<int> &idx, int waves, int *d_idx, long long *d_out, double *d_sink)
{
  CK(hipMemcpy(d_idx, idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
  for (int it = 0; it < 2; it++) {
    hipLaunchKernelGGL(k_atom, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, d_sink);
    CK(hipDeviceSynchronize());
  }
  long long h[32];
  CK(hipMemcpy(h, d_out, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int w = 0; w < waves; w++) worst = h[w] > worst ? (double) h[w] : worst;
  printf("add64 %-44s waves %2d: %7.2f clock64 ticks per atomic per wave, %7.2f per atomic of the CU\n", name, waves,
         worst / (REP * 8.0), worst / (REP * 8.0 * waves));
  return 0;
}

static int run(const char *name, const std::vector<int> &idx, int waves, bool b128, int *d_idx, long long *d_out, double *d_sink)
{
  CK(hipMemcpy(d_idx, idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
  for (int it = 0; it < 2; it++) {
    if (b128) hipLaunchKernelGGL(k_b128, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, d_sink);
    else hipLaunchKernelGGL(k_b64, dim3(1), dim3(64 * waves), 65536, 0, d_idx, d_out, d_sink);
    CK(hipDeviceSynchronize());
  }
  long long h[32];
  CK(hipMemcpy(h, d_out, sizeof(long long) * waves, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int w = 0; w < waves; w++) worst = h[w] > worst ? (double) h[w] : worst;
  printf("%-5s %-44s waves %2d: %7.2f clock64 ticks per read per wave, %7.2f per read of the CU\n", b128 ? "b128" : "b64", name, waves,
         worst / (REP * 8.0), worst / (REP * 8.0 * waves));
  return 0;
}

int main()
{
  int *d_idx; long long *d_out; double *d_sink;
  CK(hipMalloc(&d_idx, 64 * sizeof(int))); CK(hipMalloc(&d_out, 64 * sizeof(long long))); CK(hipMalloc(&d_sink, 8));
  CK(hipFuncSetAttribute((const void *) k_b64, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void *) k_b128, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void *) k_atom, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<int> v(64);
  srand(12345);
  if (getenv("PROBE_ATOMICS")) {
    for (int waves : {1, 11, 16}) {
      for (int l = 0; l < 64; l++) v[l] = l;                 run_atom("lane (consecutive)", v, waves, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = l / 2;             run_atom("lane pairs share an address", v, waves, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = 0;                 run_atom("all lanes one address", v, waves, d_idx, d_out, d_sink);
      for (int t = 0; t < 3; t++) { for (int l = 0; l < 64; l++) v[l] = rand() % 704; run_atom("random slots of 704", v, waves, d_idx, d_out, d_sink); }
      for (int t = 0; t < 2; t++) { for (int l = 0; l < 64; l++) v[l] = rand() % 3136; run_atom("random slots of 3136", v, waves, d_idx, d_out, d_sink); }
      for (int t = 0; t < 2; t++) { for (int l = 0; l < 64; l++) v[l] = rand() % 3136; run("random slots of 3136", v, waves, false, d_idx, d_out, d_sink); }
    }
    return 0;
  }
  for (int waves : {1, 16}) {
    for (int b128 = 0; b128 < 2; b128++) {
      const int nb = b128 ? 16 : 32;       // granules per 256 B
      for (int l = 0; l < 64; l++) v[l] = l;                                run("lane (consecutive)", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = 0;                                run("all lanes one address (broadcast)", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = l / 2;                            run("lane pairs share an address", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = (l % nb) + nb * 3 * (l / nb);     run("groups of nb lanes: distinct banks, other rows", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = (l % (nb / 2)) + nb * (l / (nb / 2));   run("2 lanes per bank inside each group of nb", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = (l % (nb / 4)) + nb * (l / (nb / 4));   run("4 lanes per bank inside each group of nb", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = (l % (2 * nb)) / 2 + nb * 5 * (l & 1) + nb * 11 * (l / (2 * nb));
                                                                              run("2 lanes per bank inside each group of 2 nb", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = nb * l;                           run("all lanes one bank, 64 addresses", v, waves, b128, d_idx, d_out, d_sink);
      for (int l = 0; l < 64; l++) v[l] = (l * 2) % (2 * nb) + (l * 2) / (2 * nb) * 2 * nb;  run("stride 2 granules", v, waves, b128, d_idx, d_out, d_sink);
      double avg = 0;
      for (int t = 0; t < 3; t++) { for (int l = 0; l < 64; l++) v[l] = rand() % 3136; run("random slots of 3136", v, waves, b128, d_idx, d_out, d_sink); }
      (void) avg;
      // an interleaving test: lanes l and l+32 on the same bank (different rows), lanes inside a half distinct
      for (int l = 0; l < 64; l++) v[l] = (l % 32) % nb + nb * (l / nb) * 7; run("halves collide, inside halves distinct (b64)", v, waves, b128, d_idx, d_out, d_sink);
    }
  }
  return 0;
}
