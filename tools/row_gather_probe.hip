// row_gather_probe.hip -- experiment (not product code): what one wave-"neighbour" of pass 1 costs when the three table rows
// (48 contiguous bytes at a random row of a 64 KB table) come from the LDS (ds_read_b128 x 3), from the vector L1 / L2
// (global_load_dwordx4 x 3), or from both pipes in turns, beside the three position reads (ds_read_b64 x 3 at random slots)
// that always stay in the LDS.  704 threads per workgroup, one workgroup per CU, like the force passes.
//   hipcc -O3 --offload-arch=gfx950 tools/row_gather_probe.hip -o gpurun_out/row_gather_probe && gpurun_out/row_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define NROWS 4002
#define NSLOT 3136
#define ITER 2048

// MIX: of every 8 neighbours, this many take their rows from global memory (0 = all LDS, 8 = all global)
template <int MIX, bool POS>
__global__ __launch_bounds__(704) void k_probe(const double2 *tab, const unsigned *rnd, double *sink)
{
  extern __shared__ double lds[];
  double *sp = lds;                                   // 3 planes of NSLOT
  double2 *tm = (double2 *) (lds + 3 * NSLOT);        // NROWS rows
  for (int i = threadIdx.x; i < 3 * NSLOT; i += blockDim.x) sp[i] = i * 1e-3;
  for (int i = threadIdx.x; i < NROWS; i += blockDim.x) tm[i] = tab[i];
  __syncthreads();
  unsigned s = rnd[blockIdx.x * blockDim.x + threadIdx.x];
  double acc = 0;
  for (int it = 0; it < ITER; it += 8) {
    int k[8], j[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s = s * 1664525u + 1013904223u;
      k[u] = (int) ((s >> 8) % (NROWS - 2));
      j[u] = (int) ((s >> 4) % NSLOT);
    }
    double2 A[8], B[8], C[8];
    double x[8], y[8], z[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (POS) { x[u] = sp[j[u]]; y[u] = sp[NSLOT + j[u]]; z[u] = sp[2 * NSLOT + j[u]]; }
      if (u < MIX) { const double2 *g = tab + k[u]; A[u] = g[0]; B[u] = g[1]; C[u] = g[2]; }
      else { const double2 *l = tm + k[u]; A[u] = l[0]; B[u] = l[1]; C[u] = l[2]; }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      acc += A[u].x * B[u].y + C[u].x;
      if (POS) acc += x[u] * y[u] + z[u];
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int MIX, bool POS>
static int run(const char *name, const double2 *d_tab, const unsigned *d_rnd, double *d_sink, int ncu)
{
  const size_t lds = (3 * NSLOT + 2 * NROWS) * sizeof(double);
  CK(hipFuncSetAttribute((const void *) k_probe<MIX, POS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_probe<MIX, POS>), dim3(ncu), dim3(704), lds, 0, d_tab, d_rnd, d_sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  // 11 waves per CU, ITER neighbours each
  printf("%-58s %8.3f ms  = %6.1f ns per wave-neighbour per CU\n", name, best, best * 1e6 / (11.0 * ITER));
  return 0;
}

int main()
{
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs, %d MHz\n", p.name, ncu, p.clockRate / 1000);
  std::vector<double2> tab(NROWS);
  for (int i = 0; i < NROWS; i++) tab[i] = make_double2(1.0 / (i + 1), 2.0 / (i + 3));
  std::vector<unsigned> rnd((size_t) ncu * 704);
  for (size_t i = 0; i < rnd.size(); i++) rnd[i] = (unsigned) (i * 2654435761u + 12345u);
  double2 *d_tab; unsigned *d_rnd; double *d_sink;
  CK(hipMalloc((void **) &d_tab, NROWS * sizeof(double2))); CK(hipMalloc((void **) &d_rnd, rnd.size() * 4)); CK(hipMalloc((void **) &d_sink, 8));
  CK(hipMemcpy(d_tab, tab.data(), NROWS * sizeof(double2), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
  if (run<0, false>("rows: LDS only", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<8, false>("rows: global only", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<0, true>("positions LDS + rows LDS (pass 1 today)", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<1, true>("positions LDS + rows 7/8 LDS, 1/8 global", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<2, true>("positions LDS + rows 6/8 LDS, 2/8 global", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<3, true>("positions LDS + rows 5/8 LDS, 3/8 global", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<4, true>("positions LDS + rows 4/8 LDS, 4/8 global", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<6, true>("positions LDS + rows 2/8 LDS, 6/8 global", d_tab, d_rnd, d_sink, ncu)) return 1;
  if (run<8, true>("positions LDS + rows global only", d_tab, d_rnd, d_sink, ncu)) return 1;
  return 0;
}
