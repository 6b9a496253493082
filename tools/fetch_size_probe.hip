// fetch_size_probe.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the force
// kernels ("validate the FETCH_SIZE x2 correction on a gather-shaped micro-kernel before quoting traffic").
// Every kernel reads a KNOWN number of bytes exactly once from buffers far larger than the 256 MiB Infinity Cache:
//   k_stream16   16 B per lane, coalesced                      (the guide's calibration case: reports 1/2)
//   k_stream8     8 B per lane, coalesced                      (the u16 neighbour-list words of the tile kernels)
//   k_gather32   32-B records (double4) through an int index list, every record once, indices shuffled inside
//                windows of 64 records (the staging gathers of the tile kernels: nearby, not consecutive)
//   k_gather32r  the same with indices shuffled over the whole buffer (no locality at all)
//   k_write8      8 B per lane coalesced stores                (force planes)
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_size_probe.hip -o /tmp/fetch_size_probe
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o run -- /tmp/fetch_size_probe   (then --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_stream16(const double2 *a, size_t n, double *out)
{
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_stream8(const double *a, size_t n, double *out)
{
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_gather32(const double4 *a, const int *idx, size_t n, double *out)
{
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { double4 v = a[idx[i]]; s += v.x + v.w; }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_gather32r(const double4 *a, const int *idx, size_t n, double *out)
{
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { double4 v = a[idx[i]]; s += v.x + v.w; }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_write8(double *a, size_t n)
{
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) a[i] = (double) i;
}

int main()
{
  const size_t N = (size_t) 1 << 25;                 // 33.5 M records: 1 GiB of double4, 512 MiB of double2, 256 MiB of double
  double4 *a4; double2 *a2; double *a1, *out; int *idx, *idxr;
  CK(hipMalloc(&a4, N * 32)); CK(hipMalloc(&a2, 2 * N * 16)); CK(hipMalloc(&a1, 4 * N * 8)); CK(hipMalloc(&out, 64));
  CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&idxr, N * 4));
  CK(hipMemset(a4, 0, N * 32)); CK(hipMemset(a2, 0, 2 * N * 16)); CK(hipMemset(a1, 0, 4 * N * 8));
  std::vector<int> h(N);
  std::iota(h.begin(), h.end(), 0);
  std::mt19937 rng(1);
  for (size_t w = 0; w < N; w += 64) std::shuffle(h.begin() + w, h.begin() + w + 64, rng);
  CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
  std::shuffle(h.begin(), h.end(), rng);
  CK(hipMemcpy(idxr, h.data(), N * 4, hipMemcpyHostToDevice));
  const int G = 2048, B = 256;
  hipLaunchKernelGGL(k_stream16, dim3(G), dim3(B), 0, 0, a2, 2 * N, out);
  hipLaunchKernelGGL(k_stream8, dim3(G), dim3(B), 0, 0, a1, 4 * N, out);
  hipLaunchKernelGGL(k_gather32, dim3(G), dim3(B), 0, 0, a4, idx, N, out);
  hipLaunchKernelGGL(k_gather32r, dim3(G), dim3(B), 0, 0, a4, idxr, N, out);
  hipLaunchKernelGGL(k_write8, dim3(G), dim3(B), 0, 0, a1, 4 * N);
  CK(hipDeviceSynchronize());
  printf("expected bytes read: k_stream16 %zu  k_stream8 %zu  k_gather32 %zu (+%zu index)  k_gather32r %zu (+%zu index); written: k_write8 %zu\n",
         2 * N * 16, 4 * N * 8, N * 32, N * 4, N * 32, N * 4, 4 * N * 8);
  return 0;
}
