// valu_rate_probe.hip -- experiment (not product code): issue cost in cycles of the fp64/int VALU instructions
// the force kernels are made of, one wave per SIMD, 8 independent chains, gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o gpurun_out/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP 64

#define KERNEL(NAME, ASM, CONSTR_T)                                                                         \
  __global__ void NAME(long long *out, double seed)                                                         \
  {                                                                                                         \
    CONSTR_T r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; \
    double c = 1.0000001;                                                                                   \
    long long t0 = clock64();                                                                               \
    for (int it = 0; it < REP; it++) {                                                                      \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                  \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c)); \
    }                                                                                                       \
    long long t1 = clock64();                                                                               \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                        \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678) out[1000] = 1;                                  \
  }

#define A_FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %8\n"
#define A_ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_MIN(i) "v_min_f64 %" #i ", %" #i ", %8\n"
#define A_FRACT(i) "v_fract_f64 %" #i ", %" #i "\n"
#define A_FLOOR(i) "v_floor_f64 %" #i ", %" #i "\n"
#define A_RCP(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_CMP(i) "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define A_CVTRT(i) "v_cvt_i32_f64 %" #i ", %" #i "\n v_cvt_f64_i32 %" #i ", %" #i "\n"
#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"

KERNEL(k_fma, A_FMA, double)
KERNEL(k_add, A_ADD, double)
KERNEL(k_mul, A_MUL, double)
KERNEL(k_min, A_MIN, double)
KERNEL(k_fract, A_FRACT, double)
KERNEL(k_floor, A_FLOOR, double)
KERNEL(k_rcp, A_RCP, double)
KERNEL(k_cmp, A_CMP, double)

__global__ void k_cvt(long long *out, double seed)
{
  double r[8];
  for (int q = 0; q < 8; q++) r[q] = seed + q;
  long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) { int k; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(r[q])); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[q]) : "v"(k)); }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  double s = 0; for (int q = 0; q < 8; q++) s += r[q];
  if (s == 12345.678) out[1000] = 1;
}
__global__ void k_i32(long long *out, int seed)
{
  int r[8];
  for (int q = 0; q < 8; q++) r[q] = seed + q + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[q]) : "v"(seed));
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  int s = 0; for (int q = 0; q < 8; q++) s += r[q];
  if (s == 12345678) out[1000] = 1;
}
__global__ void k_f32(long long *out, float seed)
{
  float r[8];
  for (int q = 0; q < 8; q++) r[q] = seed + q + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[q]) : "v"(seed));
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float s = 0; for (int q = 0; q < 8; q++) s += r[q];
  if (s == 12345678.f) out[1000] = 1;
}
__global__ void k_lds(long long *out, int seed)
{
  __shared__ double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = i;
  __syncthreads();
  int idx[8]; double acc = 0;
  for (int q = 0; q < 8; q++) idx[q] = (threadIdx.x * 3 + q * 67 + seed) & 4095;
  long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = sm[idx[q]];
#pragma unroll
    for (int q = 0; q < 8; q++) { acc += v[q]; idx[q] = (idx[q] + 193) & 4095; }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) out[1000] = 1;
}

int main()
{
  long long *d; CK(hipMalloc(&d, 8192 * 8));
  long long h[4];
#define RUN(K, ARG, NINS, LABEL) \
  for (int waves : {1, 2, 4}) { hipLaunchKernelGGL(K, dim3(1), dim3(256 * waves), 0, 0, d, ARG); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost)); \
    printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave (%.2f per SIMD issue)\n", LABEL, waves, (double) h[0] / (REP * 8.0 * NINS), (double) h[0] / (REP * 8.0 * NINS * waves)); }
  RUN(k_fma, 1.0, 1, "v_fma_f64")
  RUN(k_add, 1.0, 1, "v_add_f64")
  RUN(k_mul, 1.0, 1, "v_mul_f64")
  RUN(k_min, 1.0, 1, "v_min_f64")
  RUN(k_fract, 1.5, 1, "v_fract_f64")
  RUN(k_floor, 1.5, 1, "v_floor_f64")
  RUN(k_rcp, 1.5, 1, "v_rcp_f64")
  RUN(k_cmp, 1.5, 1, "v_cmp_lt_f64")
  RUN(k_cvt, 1.5, 2, "v_cvt_i32_f64+v_cvt_f64_i32")
  RUN(k_i32, 3, 1, "v_mad_u32_u24")
  RUN(k_f32, 3.f, 1, "v_fma_f32")
  RUN(k_lds, 3, 1, "ds_read_b64 (+add,+and)")
  return 0;
}
