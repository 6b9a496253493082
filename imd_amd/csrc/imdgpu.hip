// imdgpu.hip -- MI355X (gfx950, CDNA4) kernels + C ABI for the neighbour-list
// pair/EAM force path of IMD/lasMD.  Written for wave64 / 256 CUs / 160 KB LDS;
// no CUDA compatibility paths.  Public ABI and reference cross-references:
// include/imdgpu.h.  Design notes: DESIGN.md.
//
// Arithmetic: fp64 throughout.  The file is compiled with -ffp-contract=off so
// that squared distances, cut-off tests and the table interpolation evaluate
// with exactly the roundings of the reference CPU build (the reference tests
// r2 = dx*dx+dy*dy+dz*dz, makros.h:361); only summation ORDER differs from the
// reference (full-list gather instead of half-list scatter).
#include "imdgpu_internal.h"
#include <chrono>

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

// ---------------------------------------------------------------------------
// errors / profiling helpers
// ---------------------------------------------------------------------------
int imdgpu_fail(imdgpu_ctx *c, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return 1;
}

ProfScope::ProfScope(imdgpu_ctx *ctx, int id_) : c(ctx), id(id_)
{
  if (!c->prof_on) return;
  auto get = [&]() {
    hipEvent_t e;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else (void) hipEventCreate(&e);
    return e;
  };
  a = get(); b = get();
  (void) hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope()
{
  if (!a) return;
  (void) hipEventRecord(b, c->stream);
  c->ev_pending.push_back({a, b, id});
  if (c->ev_pending.size() > 8192) imdgpu_prof_resolve(c);
}
void imdgpu_prof_resolve(imdgpu_ctx *c)
{
  for (auto &p : c->ev_pending) {
    float ms = 0.f;
    (void) hipEventSynchronize(p.b);
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->prof_ms[p.id] += ms; c->prof_n[p.id]++; }
    c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
  }
  c->ev_pending.clear();
}

void imdgpu_tl_mark(imdgpu_ctx *c, const char *name)
{
  if (!c->tl_on) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void) hipEventRecord(e, c->stream);
  c->tl.push_back({name, e});
}
void imdgpu_tl_flush(imdgpu_ctx *c, int steps)
{
  if (!c->tl_on || c->tl.empty()) return;
  (void) hipDeviceSynchronize();
  fprintf(stderr, "[imdgpu timeline] step %d:", steps);
  for (auto &m : c->tl) {
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, c->tl[0].second, m.second);
    fprintf(stderr, " %s=%.3f", m.first, ms);
  }
  fprintf(stderr, "\n");
  for (auto &m : c->tl) (void) hipEventDestroy(m.second);
  c->tl.clear();
}

static const char *prof_names[PROF_COUNT] = {
  "pass1", "pass2", "pass3", "ghosts", "nblist", "fix_cells", "integrate", "check", "reduce",
  "ttm_update_fd", "ttm_diff", "ttm_fillmesh", "pass1b"};

Geo imdgpu_make_geo(const imdgpu_ctx *c)
{
  Geo g;
  memcpy(g.box, c->box, sizeof(g.box)); memcpy(g.tbox, c->tbox, sizeof(g.tbox));
  for (int d = 0; d < 3; d++) { g.pbc[d] = c->pbc[d]; g.gdim[d] = c->gdim[d]; g.dim[d] = c->dim[d]; g.my_coord[d] = c->my_coord[d]; }
  g.nx = c->dim[0] - 2; g.ny = c->dim[1] - 2; g.nz = c->dim[2] - 2;
  g.bx = c->tile[0]; g.by = c->tile[1]; g.bz = c->tile[2];
  g.ntx = c->ntile[0]; g.nty = c->ntile[1]; g.ntz = c->ntile[2];
  g.CPT = c->CPT; g.nkeys = c->nkeys;
  return g;
}
#define make_geo imdgpu_make_geo

// ---------------------------------------------------------------------------
// table interpolation (device).  Same expression tree as PAIR_INT2 /
// VAL_FUNC2 / DERIV_FUNC2 of the reference (potaccess.h:323-354,465-495,
// 591-621): clamp, truncate, three consecutive rows, quadratic form.
// ---------------------------------------------------------------------------
struct Tab3 { double p0, dv, d2v, chi, istep; };

__device__ __forceinline__ Tab3 tab3(const double *t, int inc, int col, double begin, double end,
                                     double istep, double arg, int &is_short)
{
  Tab3 r;
  double a = (arg < end ? arg : end) - begin;
  if (a < 0) { a = 0; is_short = 1; }
  a = a * istep;
  int k = (int) a;
  r.chi = a - k;
  const double *p = t + (size_t) k * inc + col;
  double p1 = p[inc], p2 = p[2 * inc];
  r.p0 = p[0];
  r.dv = p1 - r.p0;
  r.d2v = p2 - 2 * p1 + r.p0;
  r.istep = istep;
  return r;
}
__device__ __forceinline__ double tab_val(const Tab3 &r) { return r.p0 + r.chi * r.dv + 0.5 * r.chi * (r.chi - 1) * r.d2v; }
__device__ __forceinline__ double tab_grad(const Tab3 &r) { return 2 * r.istep * (r.dv + (r.chi - 0.5) * r.d2v); }

// wave64 / block reductions ----------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// sums `nv` values per thread over the block; result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch /* >= NV*16 doubles */)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int q = 0; q < NV; q++) v[q] = wave_sum(v[q]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int q = 0; q < NV; q++) scratch[q * 16 + w] = v[q];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int q = 0; q < NV; q++) {
      double s = 0;
      for (int i = 0; i < nw; i++) s += scratch[q * 16 + i];
      v[q] = s;
    }
}

// ---------------------------------------------------------------------------
// re-binning kernels  [fix_cells + do_boundaries]
// ---------------------------------------------------------------------------
// Wrap every local atom into the box direction by direction (do_boundaries,
// imd_main_3d.c:2018-2110), then compute its global cell (cell_coord,
// imd_geom_3d.c:1077-1097) and from it the rank of its inner cell on this GPU.
__global__ void k_wrap_key(int n, double4 *pos, Geo g, int *keys, int *perm, int *counters)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double4 r = pos[i];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if (g.pbc[d] != 1) continue;
    const double *t = g.tbox + 3 * d, *b = g.box + 3 * d;
    double s = -floor(r.x * t[0] + r.y * t[1] + r.z * t[2]);
    r.x += s * b[0]; r.y += s * b[1]; r.z += s * b[2];
  }
  pos[i] = r;
  int c[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double *t = g.tbox + 3 * d;
    int q = (int) (g.gdim[d] * (r.x * t[0] + r.y * t[1] + r.z * t[2]));
    if (q >= g.gdim[d]) q = g.gdim[d] - 1; else if (q < 0) q = 0;
    q -= g.my_coord[d] * (g.dim[d] - 2);            // local_cell_coord - 1
    if (q < 0 || q >= g.dim[d] - 2) {               // left this rank's domain
      atomicAdd(&counters[CNT_OUTSIDE], 1);
      q = q < 0 ? 0 : g.dim[d] - 3;
    }
    c[d] = q;
  }
  keys[i] = xyz_to_key(c[0], c[1], c[2], g);
  perm[i] = i;
}

__global__ void k_gather_sorted(int n, int cap, const int *perm, const int *keys, const double4 *pos, double4 *pos2,
                                const int *type, int *type2, const int *id, int *id2, const double *p, double *p2,
                                const double *mass, double *mass2, int *acell, int *icnt, const int *fdc, int *fdc2)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = perm[i];
  pos2[i] = pos[s]; type2[i] = type[s]; id2[i] = id[s]; mass2[i] = mass[s]; fdc2[i] = fdc[s];
  p2[i] = p[s]; p2[cap + i] = p[cap + s]; p2[2 * (size_t) cap + i] = p[2 * (size_t) cap + s];
  int k = keys[i];
  acell[i] = k;
  // atoms per key: the keys are sorted, so a key's atoms are one run of consecutive threads -- its first thread subtracts its own
  // index, its last one adds the index behind it: two atomics per key instead of one per atom (a cell's ~21 atoms adding to ONE
  // word serialise in the L2's atomic unit: the kernel took 0.70 ms at 10 M atoms, most of it waiting for those)
  const int kp = i > 0 ? keys[i - 1] : -1, kn = i + 1 < n ? keys[i + 1] : -1;
  if (kp != k) atomicSub(&icnt[k], i);
  if (kn != k) atomicAdd(&icnt[k], i + 1);
}

// cell key -> linear index in the ghost-including cell array
__global__ void k_cells_inner(int nkeys, Geo g, const int *icnt, const int *istart, int *cell_start, int *cell_cnt)
{
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nkeys) return;
  int ix, iy, iz;
  if (!key_to_xyz(r, g, ix, iy, iz)) return;
  int lin = ((ix + 1) * g.dim[1] + (iy + 1)) * g.dim[2] + (iz + 1);
  cell_start[lin] = istart[r];
  cell_cnt[lin] = icnt[r];
}

__global__ void k_ghost_cnt(int ng, const int *gsrc, const int *icnt, int *gcnt)
{
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  int s = gsrc[g];
  gcnt[g] = s >= 0 ? icnt[s] : 0;
}

__global__ void k_ghost_map(int ng, int nlocal, const int *glin, const int *gsrc, const int *gshift, const int *gcnt,
                            const int *gstart, const int *istart, int *cell_start, int *cell_cnt, int *ghost_src,
                            int *ghost_code)
{
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  int n = gcnt[g], s = gstart[g], lin = glin[g];
  cell_start[lin] = nlocal + s;
  cell_cnt[lin] = n;
  if (n == 0) return;
  int base = istart[gsrc[g]], code = gshift[g];
  for (int t = 0; t < n; t++) { ghost_src[s + t] = base + t; ghost_code[s + t] = code; }
}

// Ghost atom = periodic image of a local atom: position + box vectors, added in
// the order of the reference's z, y, x stages (copy_cell, imd_comm_force_3d.c:
// 726-755, shift vectors :247-266).  Also carries dF (w) and optionally type.
__global__ void k_ghost_update(int nghost, int nlocal, const int *src, const int *code, double4 *pos, int *type,
                               Geo g, int copy_type)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nghost) return;
  int s = src[t], c = code[t];
  double4 r = pos[s];
  int sx = c % 3 - 1, sy = (c / 3) % 3 - 1, sz = c / 9 - 1;
  if (sz) { r.x += sz * g.box[6]; r.y += sz * g.box[7]; r.z += sz * g.box[8]; }
  if (sy) { r.x += sy * g.box[3]; r.y += sy * g.box[4]; r.z += sy * g.box[5]; }
  if (sx) { r.x += sx * g.box[0]; r.y += sx * g.box[1]; r.z += sx * g.box[2]; }
  pos[nlocal + t] = r;
  if (copy_type) type[nlocal + t] = type[s];
}

// ---------------------------------------------------------------------------
// Verlet list  [make_nblist]
// ---------------------------------------------------------------------------
// Full (symmetric) list, one thread per local atom, ELL-transposed so that the
// force kernels read nbr[k*stride+i] coalesced.  Membership is the reference's
// predicate r2 < cellsz on the unfused r2 (imd_forces_nbl.c:256-263); (i,j) and
// (j,i) evaluate bit-identical r2, so the list is exactly the symmetrisation of
// the reference's half list.  Also stores the reference positions (nbl_pos).
__global__ void k_nbl_build(int nlocal, const double4 *pos, const int *acell, const int *cell_start,
                            const int *cell_cnt, Geo g, double cellsz, int *nbr, size_t stride, int W, int *nbr_cnt,
                            double *ref, int cap, int *counters)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int m = 0;
  if (i < nlocal) {
  double4 pi = pos[i];
  ref[i] = pi.x; ref[cap + i] = pi.y; ref[2 * (size_t) cap + i] = pi.z;
  int ix, iy, iz;
  key_to_xyz(acell[i], g, ix, iy, iz);
  for (int dx = -1; dx <= 1; dx++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dz = -1; dz <= 1; dz++) {
        int c = ((ix + 1 + dx) * g.dim[1] + (iy + 1 + dy)) * g.dim[2] + (iz + 1 + dz);
        int s = cell_start[c], n = cell_cnt[c];
        for (int j = s; j < s + n; j++) {
          if (j == i) continue;
          double4 pj = pos[j];
          double ex = pj.x - pi.x, ey = pj.y - pi.y, ez = pj.z - pi.z;
          double r2 = ex * ex + ey * ey + ez * ez;
          if (r2 < cellsz) {
            if (m < W) nbr[(size_t) m * stride + i] = j;
            m++;
          }
        }
      }
  nbr_cnt[i] = m < W ? m : W;
  }
  int mw = m;                                   // one atomic per wave, not per atom
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int v = __shfl_down(mw, o, 64); mw = v > mw ? v : mw; }
  if ((threadIdx.x & 63) == 0) atomicMax(&counters[CNT_MAXNBR], mw);
}

// Reference-format half list (export only): same traversal as the reference
// (14-cell stencil order of make_cell_lists, imd_geom_3d.c:907-975; j>i in the
// own cell), numbering n = ref_cl_off[cell]+slot.
__device__ __forceinline__ bool stencil_cell(int m, int ix, int iy, int iz, const Geo &g, int &c)
{
  // slot 0 = self, then (l,m,n) = (0,0,1),(0,1,-1..1),(1,-1..1,-1..1)
  int l, mm, n;
  if (m == 0) { l = 0; mm = 0; n = 0; }
  else if (m == 1) { l = 0; mm = 0; n = 1; }
  else if (m < 5) { l = 0; mm = 1; n = m - 3; }
  else { int q = m - 5; l = 1; mm = q / 3 - 1; n = q % 3 - 1; }
  int r[3] = {ix + 1 + l, iy + 1 + mm, iz + 1 + n};
  for (int d = 0; d < 3; d++) {
    int gl = r[d] - 1 + g.my_coord[d] * (g.dim[d] - 2);
    if ((gl < 0 || gl > g.gdim[d] - 1) && g.pbc[d] != 1) return false;
  }
  c = (r[0] * g.dim[1] + r[1]) * g.dim[2] + r[2];
  return true;
}

template <bool FILL>
__global__ void k_export_half(int nlocal, const double4 *pos, const int *acell, const int *cell_start,
                              const int *cell_cnt, Geo g, double cellsz, const int *ref_cl_off,
                              const int *refstart, const int *istart, int *tl_or_cnt, int *tb)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  double4 pi = pos[i];
  int ix, iy, iz;
  const int key = acell[i];
  key_to_xyz(key, g, ix, iy, iz);
  const int ri = refstart[key] + (i - istart[key]);      // index of this atom in the reference's `cells` order
  int m = 0, out = FILL ? tl_or_cnt[ri] : 0;
  for (int q = 0; q < 14; q++) {
    int c;
    if (!stencil_cell(q, ix, iy, iz, g, c)) continue;
    int s = cell_start[c], n = cell_cnt[c];
    int j0 = (q == 0) ? i + 1 : s;
    for (int j = j0; j < s + n; j++) {
      double4 pj = pos[j];
      double ex = pj.x - pi.x, ey = pj.y - pi.y, ez = pj.z - pi.z;
      double r2 = ex * ex + ey * ey + ez * ez;
      if (r2 < cellsz) {
        if (FILL) tb[out + m] = ref_cl_off[c] + (j - s);
        m++;
      }
    }
  }
  if (!FILL) tl_or_cnt[ri] = m;
}

// ---------------------------------------------------------------------------
// force passes  [calc_forces]
// ---------------------------------------------------------------------------
struct ForceArgs {
  int nlocal, ntypes, cap;
  const double4 *pos;
  const int *type;
  const int *nbr, *nbr_cnt;
  size_t stride;
  TabPar pair, rho;
  double *f, *epot, *rhoh;
  int *numneighs;
  double *partials;
  int *counters;
  const int *guard;               // see TileArgs.guard
};

__device__ __forceinline__ void lds_load_table(double *dst, const TabPar &t)
{
  const int n = t.nrows * t.ncols, n2 = n >> 1;
  const double2 *src2 = (const double2 *) t.tab;          // hipMalloc'ed: 16-byte aligned
  double2 *dst2 = (double2 *) dst;
  for (int k = threadIdx.x; k < n2; k += 4 * blockDim.x) {
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < n2) v[u] = src2[q]; }
#pragma unroll
    for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < n2) dst2[q] = v[u]; }
  }
  if ((n & 1) && threadIdx.x == 0) dst[n - 1] = t.tab[n - 1];
}

// Pass 1: pair energy/force, host electron density, TTM neighbour count
// (imd_forces_nbl.c:418-672), gather form: every local atom sums over its full
// list, nothing is scattered, ghost atoms are read-only.
template <bool PAIR, bool EAM, bool TTMN, bool NT1>
__device__ __forceinline__ void pass1_body(const ForceArgs &a, const double *tp, const double *tr, double *scratch)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int inc = a.ntypes * a.ntypes;
  double sums[2] = {0.0, 0.0};             // pair energy, virial
  int is_short = 0;
  if (i < a.nlocal) {
    const double4 pi = a.pos[i];
    const int it = NT1 ? 0 : a.type[i];
    const int cnt = a.nbr_cnt[i];
    double fx = 0, fy = 0, fz = 0, ee = 0, rr = 0, vir = 0;
    int nb = 0;
    const int *np = a.nbr + i;
    for (int k = 0; k < cnt; k++) {
      const int j = np[(size_t) k * a.stride];
      const double4 pj = a.pos[j];
      const double dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
      const double r2 = dx * dx + dy * dy + dz * dz;
      const int jt = NT1 ? 0 : a.type[j];
      const int col = NT1 ? 0 : it * a.ntypes + jt;
      if (TTMN) { if (r2 <= a.pair.end[0] && it == 0 && jt == 0) nb++; }
      if (PAIR) {
        if (r2 <= a.pair.end[col]) {
          Tab3 q = tab3(tp, inc, col, a.pair.begin[col], a.pair.end[col], a.pair.invstep[col], r2, is_short);
          double pot = tab_val(q), grad = tab_grad(q);
          fx += dx * grad; fy += dy * grad; fz += dz * grad;
          ee += 0.5 * pot;
          vir -= 0.5 * (r2 * grad);
        }
      }
      if (EAM) {
        if (r2 < a.rho.end[col]) {
          Tab3 q = tab3(tr, inc, col, a.rho.begin[col], a.rho.end[col], a.rho.invstep[col], r2, is_short);
          rr += tab_val(q);
        }
      }
    }
    a.f[i] = fx; a.f[a.cap + i] = fy; a.f[2 * (size_t) a.cap + i] = fz;
    a.epot[i] = ee;
    if (EAM) a.rhoh[i] = rr;
    if (TTMN) a.numneighs[i] = nb;
    sums[0] = ee; sums[1] = vir;
  }
  block_sum<2>(sums, scratch);
  if (threadIdx.x == 0) {
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 0] = sums[0];
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 1] = sums[1];
  }
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}
// tables copied into LDS (they fit: one atom type always does) ...
template <bool PAIR, bool EAM, bool TTMN, bool NT1>
__global__ __launch_bounds__(512) void k_pass1(ForceArgs a)
{
  if (a.guard && *a.guard) return;
  extern __shared__ double lds[];
  double *tp = lds;
  double *tr = lds + (PAIR ? a.pair.nrows * a.pair.ncols : 0);
  double *scratch = tr + (EAM ? a.rho.nrows * a.rho.ncols : 0);
  if (PAIR) lds_load_table(tp, a.pair);
  if (EAM) lds_load_table(tr, a.rho);
  __syncthreads();
  pass1_body<PAIR, EAM, TTMN, NT1>(a, tp, tr, scratch);
}
// ... or read where they are (ntypes^2 columns of a few thousand rows exceed the LDS): the rows come through L2
template <bool PAIR, bool EAM, bool TTMN>
__global__ __launch_bounds__(512) void k_pass1_gt(ForceArgs a)
{
  if (a.guard && *a.guard) return;
  __shared__ double scratch[64];
  pass1_body<PAIR, EAM, TTMN, false>(a, a.pair.tab, a.rho.tab, scratch);
}

// Pass 2: embedding energy F(rho_i) and its derivative (stored as twice the
// derivative exactly like EAM_DF, imd_forces_nbl.c:748-783; the factor is taken
// out again in pass 3).  dF goes into pos[i].w so that pass 3 gathers it with
// the position.
__global__ void k_pass2(int nlocal, int ntypes, double4 *pos, const int *type, const double *rhoh, double *epot,
                        TabPar emb, double *partials)
{
  __shared__ double scratch[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s[1] = {0.0};
  if (i < nlocal) {
    int col = ntypes == 1 ? 0 : type[i], dummy = 0;
    Tab3 q = tab3(emb.tab, ntypes, col, emb.begin[col], emb.end[col], emb.invstep[col], rhoh[i], dummy);
    double pot = tab_val(q);
    pos[i].w = tab_grad(q);
    epot[i] += pot;
    s[0] = pot;
  }
  block_sum<1>(s, scratch);
  if (threadIdx.x == 0) partials[(size_t) blockIdx.x * IMDGPU_NPART + 2] = s[0];
}

// Pass 3: EAM force, grad = 0.5*(dF_i*rho'_j + dF_j*rho'_i)
// (imd_forces_nbl.c:789-985), gather form.
template <bool NT1>
__device__ __forceinline__ void pass3_body(const ForceArgs &a, const double *tr, double *scratch)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int inc = a.ntypes * a.ntypes;
  double sums[1] = {0.0};
  int is_short = 0;
  if (i < a.nlocal) {
    const double4 pi = a.pos[i];
    const int it = NT1 ? 0 : a.type[i];
    const int cnt = a.nbr_cnt[i];
    double fx = 0, fy = 0, fz = 0, vir = 0;
    const int *np = a.nbr + i;
    for (int k = 0; k < cnt; k++) {
      const int j = np[(size_t) k * a.stride];
      const double4 pj = a.pos[j];
      const double dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
      const double r2 = dx * dx + dy * dy + dz * dz;
      const int jt = NT1 ? 0 : a.type[j];
      const int col1 = NT1 ? 0 : jt * a.ntypes + it, col2 = NT1 ? 0 : it * a.ntypes + jt;
      if (r2 < a.rho.end[col1] || r2 < a.rho.end[col2]) {
        Tab3 q = tab3(tr, inc, col1, a.rho.begin[col1], a.rho.end[col1], a.rho.invstep[col1], r2, is_short);
        double ri = tab_grad(q), rj = ri;
        if (!NT1 && col1 != col2) {
          Tab3 q2 = tab3(tr, inc, col2, a.rho.begin[col2], a.rho.end[col2], a.rho.invstep[col2], r2, is_short);
          rj = tab_grad(q2);
        }
        double grad = 0.5 * (pi.w * rj + pj.w * ri);
        double gx = dx * grad, gy = dy * grad, gz = dz * grad;
        fx += gx; fy += gy; fz += gz;
        vir -= 0.5 * (dx * gx + dy * gy + dz * gz);
      }
    }
    a.f[i] += fx; a.f[a.cap + i] += fy; a.f[2 * (size_t) a.cap + i] += fz;
    sums[0] = vir;
  }
  block_sum<1>(sums, scratch);
  if (threadIdx.x == 0) a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 3] = sums[0];
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}
template <bool NT1>
__global__ __launch_bounds__(512) void k_pass3(ForceArgs a)
{
  extern __shared__ double lds[];
  double *tr = lds;
  double *scratch = tr + a.rho.nrows * a.rho.ncols;
  lds_load_table(tr, a.rho);
  __syncthreads();
  pass3_body<NT1>(a, tr, scratch);
}
__global__ __launch_bounds__(512) void k_pass3_gt(ForceArgs a)
{
  __shared__ double scratch[64];
  pass3_body<false>(a, a.rho.tab, scratch);
}

// ===========================================================================
// LDS-tiled force path (mode 1).  One workgroup per tile of bx*by*bz inner cells:
// the positions (pass 3: + dF) of the tile and its one-cell halo are staged once
// into LDS with coalesced loads, the potential tables sit next to them, and the
// neighbour list holds 16-bit LDS slot numbers (4 per 8-byte load, ELL-transposed
// per tile).  HBM sees: list (2 B/entry), own-atom outputs, one staging pass.
// ===========================================================================
#define IMDGPU_KERNEL_REV 13
#define LDS_LIMIT (160 * 1024)
#define TABLE_LDS_LIMIT (150 * 1024)                 // global-gather kernels: tables beyond this stay in global memory
#define NH_MAX 512                                   // halo cells per tile (<= 6*6*6 = 216 for tiles up to 4^3)

struct TileArgs {
  int nlocal, ntypes, cap, W4, tile_cap, slot_cap, ntiles;
  Geo g;
  const double4 *pos;
  const int *type, *acell, *cell_start, *cell_cnt, *istart;
  ushort4 *nbr16;
  int *nbr_cnt;
  // staging plan written by k_nbl_build_tile, valid until the next re-binning: per tile the global index of the atom
  // in every LDS slot, per own atom (slot | neighbour count << 16), per tile {slots, first own atom, own atoms, 0}
  int *tsrc; unsigned *ownw, *ownseg; int4 *tinfo;         // ownseg: several atom types, see NBR_SCALE
  int dbg;                        // experiments (IMDGPU_DBG, read only by a library built with -DIMDGPU_EXPERIMENTS; see DBG()): 1 = skip the neighbour loops, 4 / 8 = the list build's candidate loops / appends
  TabPar pair, rho, emb;
  double4 *posw;                  // pos again, writable: the fused embedding pass stores F'(rho) in .w
  const double *merged;           // interleaved (phi,rho) table + 3 zero rows, or nullptr
  const double2 *coef3;           // gradient coefficients of rho_h_tab (ctx->d_coef3): ncoef rows, the last one zero
  int ncoef, klo;                 // k_pass3_tile_coef keeps rows [klo, ncoef) in LDS and steps aside when a pair fell below klo
  int p3_fallback;                // k_pass3_tile: 1 = run only when k_pass3_tile_coef steps aside
  // several atom types with tables too large for the LDS (k_pass1_tile_rec / k_pass3_tile_rec): per column and row one aligned record
  // of interpolation coefficients, read in place through L2 -- recP / recR[col * nrec + k] = (p0, dv, d2v, 0) of pair_pot / rho_h_tab,
  // rec3[col * nrec3 + k] = invstep * (dv - d2v / 2, d2v) of rho_h_tab; the last record of every column is zero
  const double4 *recP, *recR; const double2 *rec3; int nrecP, nrecR;
  // ... and the compact form of the same tables for the LDS (every distinct column with its own rows, three zero rows at the end):
  // comp*[ncomp*] doubles, loff*[col] = first row of column col
  const double *compP, *compR; int ncompP, ncompR; int loffP[IMDGPU_MAXCOL], loffR[IMDGPU_MAXCOL];
  double *f, *epot, *rhoh, *ref;
  int *numneighs;
  double *partials;
  int *counters;
  double cellsz;
  const int *guard;               // pass 1 launched ahead of the host's look at the list-validity flag: *guard != 0 -> return at once
  // k_pass1_tile_merged<.., DYN = true> (multi-rank step with the position halo in flight, imdgpu_dist.hip): tiles are taken from
  // a work queue, tile_map[position] = tile, positions [0, n_interior) = INTERIOR tiles (no cell of another rank in their halo),
  // the rest boundary tiles; dyn[0..7] / dyn[8..15] = per-XCD counters of the two classes; a launch takes classes <= dyn_last
  const int *tile_map; int n_interior; int *dyn; int dyn_last, part_off;
  float nbl_eps;                  // k_nbl_build_tile32: |r2 - cellsz| below which the single-precision test is not trusted (host: make_tile_args)
  int balance;                    // block ranges cut by cost (boxes with vacuum, moving surfaces) instead of by tiles
  const long long *wprefix;       // [ntiles + 1] cost prefix (k_tile_wprefix), read when balance is set
};

#ifdef IMDGPU_EXPERIMENTS
#define DBG(a, bit) ((a).dbg & (bit))
#else
#define DBG(a, bit) 0                   // the default build has no switch that skips work
#endif
struct TileCtx { int tile, hx, hy, hz, nh, own_start, nown; };

// Persistent workgroups: block b works on a contiguous range of tiles [t0,t1) (consecutive tiles are
// neighbours along z, so their halos overlap in L2), and the blocks that share an XCD (b % 8 under the
// observed round-robin placement; speed only) get adjacent ranges.  With uneven occupancy the ranges are cut by COST, not
// by tiles -- in a slab with vacuum (config 4, the Maxwell-solver slab) half of the tiles are empty and equal tile counts
// would idle half the chip: two binary searches per block over a cost prefix.
// wprefix[t] = cost of the tiles before t, cost(t) = own atoms + TILE_FIXED_COST if the tile has any (the per-tile staging and
// barriers of a force pass are worth several hundred atoms' neighbour loops there: 512 measured best, flat up to 1024): a slab whose surface has just moved into the next layer of
// tiles has hundreds of sparse tiles there, and ranges cut by atoms alone hand all of them to ONE block (measured on the
// 10 M-atom slab: passes 2.0 -> 4.1 ms once 800 tiles hold 16..127 atoms each)
#ifndef TILE_FIXED_COST
#define TILE_FIXED_COST 512
#endif
__device__ __forceinline__ int tile_of_cost(const long long *wprefix, int ntiles, long long c)
{
  int lo = 0, hi = ntiles;                      // smallest t with wprefix[t + 1] > c
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (wprefix[mid + 1] > c) hi = mid; else lo = mid + 1; }
  return lo;
}
__device__ __forceinline__ void block_tile_range(int ntiles, const long long *wprefix, int balance, int &t0, int &t1)
{
  const int nwg = gridDim.x, per = nwg >> 3, b = blockIdx.x;
  const int p = (nwg & 7) == 0 ? (b & 7) * per + (b >> 3) : b;
  const long long W = balance ? wprefix[ntiles] : 0;
  if (!balance || W <= 0) {                     // equal tile counts (dense periodic boxes: identical result, no search)
    const int chunk = (ntiles + nwg - 1) / nwg;
    t0 = p * chunk; t1 = t0 + chunk;
    if (t0 > ntiles) t0 = ntiles;
    if (t1 > ntiles) t1 = ntiles;
    return;
  }
  // block p: from the tile holding cost unit ceil(W p / G) up to (not including) the tile holding unit ceil(W (p+1) / G); the
  // last block ends behind the last tile with atoms.  Empty tiles in front of the first and behind the last atom belong to nobody.
  const long long a0 = (W * p + nwg - 1) / nwg, a1 = (W * (p + 1) + nwg - 1) / nwg;
  t0 = a0 >= W ? ntiles : tile_of_cost(wprefix, ntiles, a0);
  t1 = a1 >= W ? (a0 >= W ? ntiles : tile_of_cost(wprefix, ntiles, W - 1) + 1) : tile_of_cost(wprefix, ntiles, a1);
}

// one workgroup: cost prefix over the tiles (a chunk per thread, then a scan of the chunk sums)
__global__ __launch_bounds__(1024) void k_tile_wprefix(int ntiles, int CPT, const int *istart, long long *wprefix)
{
  __shared__ long long part[1024];
  const int chunk = (ntiles + 1023) / 1024, t0 = threadIdx.x * chunk, t1 = min(ntiles, t0 + chunk);
  long long sum = 0;
  for (int t = t0; t < t1; t++) { const int o = istart[(size_t) (t + 1) * CPT] - istart[(size_t) t * CPT]; sum += o > 0 ? o + TILE_FIXED_COST : 0; }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { long long acc = 0; for (int q = 0; q < 1024; q++) { const long long v = part[q]; part[q] = acc; acc += v; } wprefix[ntiles] = acc; }
  __syncthreads();
  long long acc = part[threadIdx.x];
  for (int t = t0; t < t1; t++) { wprefix[t] = acc; const int o = istart[(size_t) (t + 1) * CPT] - istart[(size_t) t * CPT]; acc += o > 0 ? o + TILE_FIXED_COST : 0; }
}

#define SLOTCAP(a) ((a).slot_cap)
// FAST_CAP: the plane stride the neighbour loops of the force kernels are ALSO compiled for as a constant (template
// parameter CAP; 0 = the run-time stride a.slot_cap).  With a constant stride the y / z (/ dF) reads of a neighbour are
// immediate offsets of the x read's address -- two (pass 3: three) address computations fewer per neighbour -- and being
// volatile LDS reads they stay three ds_read_b64 (left to itself the compiler pairs them into ds_read2st64_b64, which
// costs more under random slots than it saves: +4..7 %, profiles/r03_kernel_variants.json).  Measured, 10 M atoms + TTM:
// pass 1 1.896 -> 1.854 ms, pass 3 1.656 -> 1.624, the lattice chain beside it 1.57 -> 1.42, the step 4.57 -> 4.49 ms.
// make_nblist rounds slot_cap up to FAST_CAP whenever the largest tile stages fewer atoms and the LDS holds the wider planes
// (3136 = 49 x 64 slots is what a 2 x 4 x 4 tile of a condensed sample needs: ~21 atoms in each of its 4 x 6 x 6 cells).
#define FAST_CAP 3136
// The 16-bit entries of the tile lists are the neighbour's slot TIMES 8: its byte offset inside a plane of staged doubles
// (slot_cap <= 6826 by the LDS size, so 8 * slot < 65536).  With the planes at the start of the workgroup's LDS the
// extracted half-word IS the address of the x read -- one shift-and-add per neighbour less in each force pass (measured on one
// box, three runs each: step 4.523 -> 4.509 ms; within the noise of a single run, kept because it is free).
#define NBR_SCALE 8
// ... and with several atom types a row is SEGMENTED by the neighbour's type: first the neighbours of type 0 (in the reference's order),
// padded with dummies to a whole word, then those of type 1, ... -- ownseg[tile * tile_cap + o] holds the end word of segment k in
// byte k, k < ntypes - 1 (the other bytes stay 0xff: the last segment runs to the row's end, and so do the dummy words behind it).  The record kernels then know the column of a whole list word (four neighbours) from its index
// and fetch the column's parameters once per word instead of once per neighbour; an entry stays the plain address.  The sums of an
// atom run over its neighbours type by type instead of cell by cell (rounding-level difference, bar 1e-9).
typedef __attribute__((address_space(3))) const volatile double lds_cvdouble;
// halo-cell table + prefix (hstart/hsrc) and staging of positions into LDS as x, y, z planes of slot_cap
// doubles (NC = 4: + dF plane sw).  Used by the list build; the force passes stage from the plan it writes.  The last slot (slot_cap-1) is a far-away dummy that pads the
// neighbour list, so the inner loops need no "valid entry" test.  All threads of the block call it.
template <int NC, bool WITH_T>
__device__ __forceinline__ void tile_stage(const TileArgs &a, TileCtx &T, double *sp, int *ts, int *hstart, int *hsrc,
                                           double *sw = nullptr)
{
  const Geo &g = a.g;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int tz = T.tile % g.ntz, q = T.tile / g.ntz, ty = q % g.nty, tx = q / g.nty;
  T.hx = g.bx + 2; T.hy = g.by + 2; T.hz = g.bz + 2; T.nh = T.hx * T.hy * T.hz;
  for (int h = tid; h < T.nh; h += nthr) {
    int c = h % T.hz, r = h / T.hz, b = r % T.hy, aa = r / T.hy;
    int fx = tx * g.bx + aa, fy = ty * g.by + b, fz = tz * g.bz + c;
    int cnt = 0, src = 0;
    if (fx < g.dim[0] && fy < g.dim[1] && fz < g.dim[2]) {
      int lin = (fx * g.dim[1] + fy) * g.dim[2] + fz;
      cnt = a.cell_cnt[lin]; src = a.cell_start[lin];
    }
    hstart[h + 1] = cnt; hsrc[h] = src;
  }
  if (tid == 0) {
    sp[SLOTCAP(a) - 1] = 1e150; sp[2 * SLOTCAP(a) - 1] = 1e150; sp[3 * SLOTCAP(a) - 1] = 1e150;
    if (NC == 4) sw[SLOTCAP(a) - 1] = 0.0;
    if (WITH_T) ts[SLOTCAP(a) - 1] = 0;
  }
  __syncthreads();
  if (tid < 64) {
    int carry = 0;
    for (int base = 0; base < T.nh; base += 64) {
      int idx = base + tid, v = idx < T.nh ? hstart[idx + 1] : 0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (tid >= o) v += t; }
      if (idx < T.nh) hstart[idx + 1] = v + carry;
      carry += __shfl(v, 63, 64);
    }
    if (tid == 0) hstart[0] = 0;
  }
  __syncthreads();
  // one half-wave per halo cell, four cells in flight per half-wave
  const int hw = tid >> 5, l = tid & 31, nhw = nthr >> 5;
  for (int h0 = hw; h0 < T.nh; h0 += 4 * nhw) {
    double4 p[4]; int sl[4], tt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int h = h0 + u * nhw;
      sl[u] = -1;
      if (h < T.nh) {
        int s = hstart[h], n = hstart[h + 1] - s;
        if (l < n && s + l < SLOTCAP(a) - 1) { sl[u] = s + l; p[u] = a.pos[hsrc[h] + l]; if (WITH_T) tt[u] = a.type[hsrc[h] + l]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (sl[u] >= 0) {
        sp[sl[u]] = p[u].x; sp[SLOTCAP(a) + sl[u]] = p[u].y; sp[2 * SLOTCAP(a) + sl[u]] = p[u].z;
        if (NC == 4) sw[sl[u]] = p[u].w;
        if (WITH_T) ts[sl[u]] = tt[u];
      }
  }
  for (int h = hw; h < T.nh; h += nhw) {                    // rare: cells with more than 32 atoms
    int s = hstart[h], n = hstart[h + 1] - s, src = hsrc[h];
    for (int e = l + 32; e < n; e += 32)
      if (s + e < SLOTCAP(a) - 1) {
        double4 q = a.pos[src + e];
        sp[s + e] = q.x; sp[SLOTCAP(a) + s + e] = q.y; sp[2 * SLOTCAP(a) + s + e] = q.z;
        if (NC == 4) sw[s + e] = q.w;
        if (WITH_T) ts[s + e] = a.type[src + e];
      }
  }
  T.own_start = a.istart[T.tile * g.CPT];
  T.nown = a.istart[(T.tile + 1) * g.CPT] - T.own_start;
  __syncthreads();
}

// LDS slot of own atom i (global index) and the halo coordinates of its cell
__device__ __forceinline__ int own_slot(const TileArgs &a, const TileCtx &T, int i, const int *hstart, const int *hsrc,
                                        int &lx, int &ly, int &lz)
{
  const Geo &g = a.g;
  int cit = a.acell[i] - T.tile * g.CPT;
  lz = cit % g.bz; int q = cit / g.bz; ly = q % g.by; lx = q / g.by;
  int h = ((lx + 1) * T.hy + (ly + 1)) * T.hz + (lz + 1);
  return hstart[h] + (i - hsrc[h]);
}

// ---- staging from the plan (force passes): see "staging AHEAD" below ------------------------------------
// LDS layout of the staged positions.  Default: x, y, z (, dF) planes of 8-byte records.  make EXTRA=-DPOSXY: (x, y) pairs in
// one plane of 16-byte records and z in a plane of its own -- pass 3: (z, dF) pairs in the second plane --, i.e. one
// ds_read_b128 + one ds_read_b64 per neighbour (pass 3: two ds_read_b128) instead of three (four) ds_read_b64.  Measured in
// round 3 (profiles/r03_kernel_variants.json, 10 M atoms): without TTM pass 1 -0.9 %, pass 3 -1.7 %; but the 128-bit loads
// need aligned register quads -- pass 3 grows from 138 to 148 VGPRs, three of its waves then leave no SIMD with the 80
// registers a wave of the lattice chain needs, and the step WITH TTM gets 33 % slower (6.76 vs 5.07 ms).  Not the default.
#ifdef POSXY
#define LPOS_RD(sp, sw, cap, j, X, Y, Z) { const double2 q_ = ((const double2 *) (sp))[j]; X = q_.x; Y = q_.y; Z = (sp)[2 * (size_t) (cap) + (j)]; }
#ifndef P3_SPLITZW
#define LPOS_RDW(sp, sw, cap, j, X, Y, Z, W) { const double2 q_ = ((const double2 *) (sp))[j], r_ = ((const double2 *) (sp) + (cap))[j]; X = q_.x; Y = q_.y; Z = r_.x; W = r_.y; }
#define LPOS_STW(sp, sw, cap, s, X, Y, Z, W) { ((double2 *) (sp))[s] = make_double2(X, Y); ((double2 *) (sp) + (cap))[s] = make_double2(Z, W); }
#else
#define LPOS_RDW(sp, sw, cap, j, X, Y, Z, W) { const double2 q_ = ((const double2 *) (sp))[j]; X = q_.x; Y = q_.y; Z = (sp)[2 * (size_t) (cap) + (j)]; W = (sw)[j]; }
#define LPOS_STW(sp, sw, cap, s, X, Y, Z, W) { ((double2 *) (sp))[s] = make_double2(X, Y); (sp)[2 * (size_t) (cap) + (s)] = (Z); (sw)[s] = (W); }
#endif
#define LPOS_ST(sp, sw, cap, s, X, Y, Z) { ((double2 *) (sp))[s] = make_double2(X, Y); (sp)[2 * (size_t) (cap) + (s)] = (Z); }
#else
#define LPOS_RD(sp, sw, cap, j, X, Y, Z) { X = (sp)[j]; Y = (sp)[(cap) + (j)]; Z = (sp)[2 * (size_t) (cap) + (j)]; }
#define LPOS_RDW(sp, sw, cap, j, X, Y, Z, W) { X = (sp)[j]; Y = (sp)[(cap) + (j)]; Z = (sp)[2 * (size_t) (cap) + (j)]; W = (sw)[j]; }
#define LPOS_ST(sp, sw, cap, s, X, Y, Z) { (sp)[s] = (X); (sp)[(cap) + (s)] = (Y); (sp)[2 * (size_t) (cap) + (s)] = (Z); }
#define LPOS_STW(sp, sw, cap, s, X, Y, Z, W) { (sp)[s] = (X); (sp)[(cap) + (s)] = (Y); (sp)[2 * (size_t) (cap) + (s)] = (Z); (sw)[s] = (W); }
#endif
// a neighbour's staged record in the loops of the force kernels, by its list entry jb = slot * NBR_SCALE (a byte offset):
// CAP = 0 the run-time plane stride, CAP > 0 the constant stride (FAST_CAP)
template <int CAP>
__device__ __forceinline__ void nbr_rd(const double *sp, int cap, int jb, double &X, double &Y, double &Z)
{
#ifdef POSXY
  LPOS_RD(sp, sp, cap, jb / NBR_SCALE, X, Y, Z)
#else
  if constexpr (CAP > 0) { lds_cvdouble *q = (lds_cvdouble *) ((const char *) sp + jb); X = q[0]; Y = q[CAP]; Z = q[2 * CAP]; }
  else { const double *q = (const double *) ((const char *) sp + jb); X = q[0]; Y = q[cap]; Z = q[2 * (size_t) cap]; }
#endif
}
template <int CAP>
__device__ __forceinline__ void nbr_rdw(const double *sp, const double *sw, int cap, int jb, double &X, double &Y, double &Z, double &W)
{
#ifdef POSXY
  LPOS_RDW(sp, sw, cap, jb / NBR_SCALE, X, Y, Z, W)
#else
  if constexpr (CAP > 0) { lds_cvdouble *q = (lds_cvdouble *) ((const char *) sp + jb); X = q[0]; Y = q[CAP]; Z = q[2 * CAP]; W = q[3 * CAP]; }
  else { const double *q = (const double *) ((const char *) sp + jb); X = q[0]; Y = q[cap]; Z = q[2 * (size_t) cap]; W = *(const double *) ((const char *) sw + jb); }
#endif
}
// the type of the atom in a neighbour's slot (several atom types)
__device__ __forceinline__ int nbr_type(const int *ts, int jb) { return *(const int *) ((const char *) ts + jb / (NBR_SCALE / 4)); }
// per tile: own atoms and staged atoms, for sizing tile_cap / slot_cap
__global__ void k_tile_stats(int ntiles, Geo g, const int *cell_cnt, const int *istart, int *counters)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int own = 0, halo = 0;
  if (t < ntiles) {
    own = istart[(t + 1) * g.CPT] - istart[t * g.CPT];
    int tz = t % g.ntz, q = t / g.ntz, ty = q % g.nty, tx = q / g.nty;
    for (int aa = 0; aa < g.bx + 2; aa++) for (int b = 0; b < g.by + 2; b++) for (int c = 0; c < g.bz + 2; c++) {
      int fx = tx * g.bx + aa, fy = ty * g.by + b, fz = tz * g.bz + c;
      if (fx < g.dim[0] && fy < g.dim[1] && fz < g.dim[2]) halo += cell_cnt[(fx * g.dim[1] + fy) * g.dim[2] + fz];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int v = __shfl_down(own, o, 64); own = v > own ? v : own;
    v = __shfl_down(halo, o, 64); halo = v > halo ? v : halo;
  }
  if ((threadIdx.x & 63) == 0) { atomicMax(&counters[CNT_MAXOWN], own); atomicMax(&counters[CNT_MAXHALO], halo); }
}

// Verlet list build, tile form.  Same predicate / same candidate order as k_nbl_build.
#ifndef NBL_GROUP
#define NBL_GROUP 4               // candidates in flight per thread (a power of two <= 32)
#endif
__global__ __launch_bounds__(1024) void k_nbl_build_tile(TileArgs a)
{
  extern __shared__ double lds[];
  double *sp = lds;
  int *hstart = (int *) (sp + 3 * (size_t) SLOTCAP(a)), *hsrc = hstart + NH_MAX + 1;
  unsigned char *ft = (unsigned char *) (hsrc + NH_MAX);      // several atom types: the type of the atom in every slot
  const bool typed = a.ntypes > 1;
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  const unsigned short dummy = (unsigned short) ((SLOTCAP(a) - 1) * NBR_SCALE);      // list entries: slot * NBR_SCALE (see NBR_SCALE)
  int mmax = 0;
  for (T.tile = t0; T.tile < t1; T.tile++) {
  __syncthreads();
  tile_stage<3, false>(a, T, sp, nullptr, hstart, hsrc);
  {                                                  // the staging plan of this tile for the force passes
    int *src = a.tsrc + (size_t) T.tile * SLOTCAP(a);
    const int hw = threadIdx.x >> 5, l = threadIdx.x & 31, nhw = blockDim.x >> 5;
    for (int h = hw; h < T.nh; h += nhw) {
      const int s0 = hstart[h], n = hstart[h + 1] - s0, g0 = hsrc[h];
      for (int e = l; e < n; e += 32) if (s0 + e < SLOTCAP(a) - 1) { src[s0 + e] = g0 + e; if (typed) ft[s0 + e] = (unsigned char) a.type[g0 + e]; }
    }
    if (typed) __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = hstart[T.nh];
      a.tinfo[T.tile] = make_int4(tot < SLOTCAP(a) - 1 ? tot : SLOTCAP(a) - 1, T.own_start, T.nown, 0);
    }
  }
  for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
    const int i = T.own_start + o;
    int lx, ly, lz;
    const int slot = own_slot(a, T, i, hstart, hsrc, lx, ly, lz);
    const double *sx = sp, *sy = sp + SLOTCAP(a), *sz = sp + 2 * SLOTCAP(a);
    const double xi = sx[slot], yi = sy[slot], zi = sz[slot];
    a.ref[i] = xi; a.ref[a.cap + i] = yi; a.ref[2 * (size_t) a.cap + i] = zi;
    unsigned long long *lp = (unsigned long long *) (a.nbr16 + (size_t) T.tile * a.W4 * a.tile_cap + o);
    // four 16-bit slots per list word, filled through a shift register (first entry ends up in the low bits;
    // a partial last word keeps dummies in its low slots)
    const unsigned long long dummy64 = (unsigned long long) dummy * 0x0001000100010001ull;
    unsigned long long w = dummy64;
    int m = 0, mreal = 0;
    unsigned segw = 0xffffffffu;
    for (int pt = 0; pt < (typed ? a.ntypes : 1); pt++) {      // several types: one sweep per neighbour type (NBR segments)
    for (int dx = 0; dx <= (DBG(a, 4) ? -1 : 2); dx++)
      for (int dy = 0; dy <= 2; dy++)
        for (int dz = 0; dz <= 2; dz++) {
          const int h = ((lx + dx) * T.hy + (ly + dy)) * T.hz + (lz + dz);
          const int s = hstart[h], e = hstart[h + 1];
          // two steps per neighbour cell: (1) branch-free distance tests, four candidates in flight, accepted ones
          // recorded as bits; (2) the append loop runs over the set bits only (~10 % of the candidates pass)
          for (int j0 = s; j0 < e; j0 += 32) {
            const int cnt = e - j0 < 32 ? e - j0 : 32;
            unsigned mask = 0;
            // reads may run up to NBL_GROUP - 1 slots past the cell (the next cell's atoms or the planes' padding): their bits are
            // cleared below, so the loop needs no per-candidate bounds or self tests
            for (int u0 = 0; u0 < cnt; u0 += NBL_GROUP) {
              const double *px = sx + j0 + u0, *py = sy + j0 + u0, *pz = sz + j0 + u0;
              double r2[NBL_GROUP];
#pragma unroll
              for (int u = 0; u < NBL_GROUP; u++) {
                const double ex = px[u] - xi, ey = py[u] - yi, ez = pz[u] - zi;
                r2[u] = ex * ex + ey * ey + ez * ez;          // unfused, as the reference (imd_forces_nbl.c:218-222)
              }
              unsigned grp = 0;
#pragma unroll
              for (int u = 0; u < NBL_GROUP; u++) grp |= (r2[u] < a.cellsz) ? (1u << u) : 0u;
              mask |= grp << u0;
            }
            mask &= cnt == 32 ? 0xffffffffu : ((1u << cnt) - 1u);
            if (slot >= j0 && slot < j0 + 32) mask &= ~(1u << (slot - j0));   // not its own neighbour
            if (DBG(a, 8)) mask = 0;
            while (mask) {                                   // ascending j: the reference's order inside a cell
              const int j = j0 + __builtin_ctz(mask);
              mask &= mask - 1;
              if (typed && ft[j] != pt) continue;
              w = (w >> 16) | ((unsigned long long) (j * NBR_SCALE) << 48);
              m++; mreal++;
              if ((m & 3) == 0) {
                if ((m >> 2) <= a.W4) lp[(size_t) ((m >> 2) - 1) * a.tile_cap] = w;
                w = dummy64;
              }
            }
          }
        }
    if ((m & 3) != 0) {                                      // the segment's (the row's) last word: dummies in the remaining slots
      for (int q = m & 3; q < 4; q++) w = (w >> 16) | ((unsigned long long) dummy << 48);   // entries to the low slots
      if ((m >> 2) < a.W4) lp[(size_t) (m >> 2) * a.tile_cap] = w;
      m = (m + 3) & ~3; w = dummy64;
    }
    if (pt < a.ntypes - 1) segw = (segw & ~(0xffu << (8 * pt))) | ((unsigned) (m >> 2 < 255 ? m >> 2 : 255) << (8 * pt));
    }
    a.nbr_cnt[i] = mreal < 4 * a.W4 ? mreal : 4 * a.W4;
    a.ownw[(size_t) T.tile * a.tile_cap + o] = (unsigned) slot | ((unsigned) (m < 4 * a.W4 ? m : 4 * a.W4) << 16);
    if (typed) a.ownseg[(size_t) T.tile * a.tile_cap + o] = segw;
    mmax = m > mmax ? m : mmax;
  }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int v = __shfl_down(mmax, o, 64); mmax = v > mmax ? v : mmax; }
  if ((threadIdx.x & 63) == 0 && mmax > 0) atomicMax(&a.counters[CNT_MAXNBR], mmax);
}

// The list build again, with the distance tests in SINGLE precision and an exact stand-in where that could matter
// (kernel_rev 8; VERDICT r3 item 8: the build issued 2.1e9 VALU wave-instructions per rebuild, four times pass 1).
// The membership test is the reference's `r2 < cellsz` on its unfused double-precision r2 (imd_forces_nbl.c:218-224), and the
// list must come out bit for bit.  Almost every candidate is far from the threshold: positions are staged as FLOATS
// relative to the tile's first staged atom (|coordinate| < 64 A: error <= 2^-24 * 64 per coordinate, <= 2e-4 in r2 at the list
// radius), four consecutive candidates come with three ds_read_b128, the squared distances with packed fp32 arithmetic (two
// candidates per instruction), and the accepted ones are shifted into the lane's bit mask through the SIGN of r2 - cellsz (one
// v_alignbit per candidate).  A chunk of 32 candidates in which ANY lane of the wave saw |r2 - cellsz| < eps (a rigorous bound of
// the single-precision error from the tile's extent, ~5e-4: one chunk in ~100) is done again by the whole wave with the reference's expression on the double-precision positions (global
// memory, through the staging plan this workgroup has just written) -- the decision is then the reference's, exactly.
// Same candidate order as k_nbl_build_tile (neighbour cells in the order of cnbrs[c].nq[], j ascending), same outputs.
typedef float nbl_f2 __attribute__((ext_vector_type(2)));
// TY: 1 = one atom type (plain rows), 2 = two types (segmented rows from one sweep), 0 = three or four types (a sweep per type)
template <int TY>
__global__ __launch_bounds__(1024) void k_nbl_build_tile32(TileArgs a)
{
  extern __shared__ double lds[];
  float *fx = (float *) lds, *fy = fx + SLOTCAP(a), *fz = fy + SLOTCAP(a);
  int *hstart = (int *) (fz + SLOTCAP(a)), *hsrc = hstart + NH_MAX + 1;
  unsigned char *ft = (unsigned char *) (hsrc + NH_MAX);      // several atom types: the type of the atom in every slot
  constexpr bool typed = TY != 1;
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  const unsigned short dummy = (unsigned short) ((SLOTCAP(a) - 1) * NBR_SCALE);
  const float cth = (float) a.cellsz;
  int mmax = 0;
  for (T.tile = t0; T.tile < t1; T.tile++) {
    __syncthreads();
    // ---- halo-cell table and prefix (as tile_stage) ----
    const Geo &g = a.g;
    const int tid = threadIdx.x, nthr = blockDim.x;
    {
      const int tz = T.tile % g.ntz, q = T.tile / g.ntz, ty = q % g.nty, tx = q / g.nty;
      T.hx = g.bx + 2; T.hy = g.by + 2; T.hz = g.bz + 2; T.nh = T.hx * T.hy * T.hz;
      for (int h = tid; h < T.nh; h += nthr) {
        int c = h % T.hz, r = h / T.hz, b = r % T.hy, aa = r / T.hy;
        int fxc = tx * g.bx + aa, fyc = ty * g.by + b, fzc = tz * g.bz + c;
        int cnt = 0, src = 0;
        if (fxc < g.dim[0] && fyc < g.dim[1] && fzc < g.dim[2]) {
          int lin = (fxc * g.dim[1] + fyc) * g.dim[2] + fzc;
          cnt = a.cell_cnt[lin]; src = a.cell_start[lin];
        }
        hstart[h + 1] = cnt; hsrc[h] = src;
      }
      __syncthreads();
      if (tid < 64) {
        int carry = 0;
        for (int base = 0; base < T.nh; base += 64) {
          int idx = base + tid, v = idx < T.nh ? hstart[idx + 1] : 0;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (tid >= o) v += t; }
          if (idx < T.nh) hstart[idx + 1] = v + carry;
          carry += __shfl(v, 63, 64);
        }
        if (tid == 0) hstart[0] = 0;
      }
      __syncthreads();
    }
    T.own_start = a.istart[T.tile * g.CPT];
    T.nown = a.istart[(T.tile + 1) * g.CPT] - T.own_start;
    const int tot = hstart[T.nh] < SLOTCAP(a) - 1 ? hstart[T.nh] : SLOTCAP(a) - 1;
    // origin: the tile's first own atom (any point inside the staged region does)
    const double4 org = a.pos[T.nown > 0 ? T.own_start : 0];
    int *src = a.tsrc + (size_t) T.tile * SLOTCAP(a);
    {   // positions as floats relative to the origin; the staging plan of the force passes
      const int hw = tid >> 5, l = tid & 31, nhw = nthr >> 5;
      for (int h = hw; h < T.nh; h += nhw) {
        const int s0 = hstart[h], n = hstart[h + 1] - s0, g0 = hsrc[h];
        for (int e = l; e < n; e += 32)
          if (s0 + e < SLOTCAP(a) - 1) {
            const double4 q = a.pos[g0 + e];
            fx[s0 + e] = (float) (q.x - org.x); fy[s0 + e] = (float) (q.y - org.y); fz[s0 + e] = (float) (q.z - org.z);
            src[s0 + e] = g0 + e;
            if (typed) ft[s0 + e] = (unsigned char) a.type[g0 + e];
          }
      }
      for (int s = tot + tid; s < ((tot + 3) & ~3) + 4 && s < SLOTCAP(a); s += nthr) { fx[s] = 1e30f; fy[s] = 1e30f; fz[s] = 1e30f; }   // what an aligned group may read behind the last atom
      if (tid == 0) a.tinfo[T.tile] = make_int4(tot, T.own_start, T.nown, 0);
    }
    __syncthreads();                                   // planes and plan complete (the plan is read back by the exact stand-in)
    for (int o = tid; o < T.nown; o += nthr) {
      const int i = T.own_start + o;
      int lx, ly, lz;
      const int slot = own_slot(a, T, i, hstart, hsrc, lx, ly, lz);
      const double4 pi = a.pos[i];
      a.ref[i] = pi.x; a.ref[a.cap + i] = pi.y; a.ref[2 * (size_t) a.cap + i] = pi.z;
      const float xi = fx[slot], yi = fy[slot], zi = fz[slot];
      const nbl_f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi}, cth2 = {cth, cth};
      unsigned long long *lp = (unsigned long long *) (a.nbr16 + (size_t) T.tile * a.W4 * a.tile_cap + o);
      const unsigned long long dummy64 = (unsigned long long) dummy * 0x0001000100010001ull;
      unsigned long long w = dummy64;
      int m = 0, mreal = 0;
      unsigned segw = 0xffffffffu;
      // several types, the rows segmented by neighbour type (see NBR_SCALE).  Two types: ONE sweep -- the neighbours of type 1 are
      // collected from the row's END downwards and moved behind the segment of type 0 afterwards (a handful of words per atom);
      // three and four types: one sweep per type
      constexpr bool two = TY == 2;
      unsigned long long w1 = dummy64;
      int m1 = 0;
      for (int pt = 0; pt < (typed && !two ? a.ntypes : 1); pt++) {
      // the three neighbour cells dz = 0, 1, 2 of a (dx, dy) pair are consecutive halo cells, so their atoms are ONE run of
      // slots: nine runs of ~63 candidates instead of 27 cells of ~21 -- the same candidates in the same order (cells by dz,
      // atoms ascending inside a cell), a third of the per-run bookkeeping and fuller chunks
      for (int dx = 0; dx <= (DBG(a, 4) ? -1 : 2); dx++)
        for (int dy = 0; dy <= 2; dy++) {
            const int h = ((lx + dx) * T.hy + (ly + dy)) * T.hz + lz;
            const int s = hstart[h], e = hstart[h + 3];
            for (int j0 = s & ~3; j0 < e; j0 += 32) {
              // bit 31 - u of `mask` = candidate j0 + u (shifted in through the sign of r2 - cellsz, first candidate highest)
              unsigned mask = 0;
              float near = 1e30f;
              const int ngrp = (e - j0 + 3) >> 2 < 8 ? (e - j0 + 3) >> 2 : 8;
              for (int q = 0; q < ngrp; q++) {
                const float4 X = *(const float4 *) (fx + j0 + 4 * q), Y = *(const float4 *) (fy + j0 + 4 * q), Z = *(const float4 *) (fz + j0 + 4 * q);
                nbl_f2 ax = {X.x, X.y}, bx = {X.z, X.w}, ay = {Y.x, Y.y}, by = {Y.z, Y.w}, az = {Z.x, Z.y}, bz = {Z.z, Z.w};
                ax -= xi2; bx -= xi2; ay -= yi2; by -= yi2; az -= zi2; bz -= zi2;
                nbl_f2 ra = ax * ax, rb = bx * bx;
                ra = __builtin_elementwise_fma(ay, ay, ra); rb = __builtin_elementwise_fma(by, by, rb);
                ra = __builtin_elementwise_fma(az, az, ra); rb = __builtin_elementwise_fma(bz, bz, rb);
                ra -= cth2; rb -= cth2;                                  // negative = inside
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(ra.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(ra.y), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(rb.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(rb.y), 31);
                near = fminf(fminf(near, fabsf(ra.x)), fabsf(ra.y));
                near = fminf(fminf(near, fabsf(rb.x)), fabsf(rb.y));
              }
              mask <<= 32 - 4 * ngrp;                                    // first candidate of the chunk in bit 31
              // candidates of this chunk that exist and are not the atom itself
              const int lo = s > j0 ? s - j0 : 0, hi = e - j0 < 32 ? e - j0 : 32;          // [lo, hi) within the chunk
              unsigned valid = (lo == 0 ? 0xffffffffu : (0xffffffffu >> lo)) & (hi == 32 ? 0xffffffffu : ~(0xffffffffu >> hi));
              if (slot >= j0 && slot < j0 + 32) valid &= ~(0x80000000u >> (slot - j0));
              if (__any(near < a.nbl_eps)) {
                // some lane of the wave is close to the threshold in this chunk: the reference's own expression for everybody
                mask = 0;
                for (int u = lo; u < hi; u++) {
                  const double4 pj = a.pos[src[j0 + u]];
                  const double ex = pj.x - pi.x, ey = pj.y - pi.y, ez = pj.z - pi.z;
                  const double r2 = ex * ex + ey * ey + ez * ez;        // unfused (the file is compiled -ffp-contract=off)
                  mask |= (r2 < a.cellsz) ? (0x80000000u >> u) : 0u;
                }
              }
              mask &= valid;
              if (DBG(a, 8)) mask = 0;
              while (mask) {                                             // ascending j: the reference's order inside a cell
                const int b = __clz((int) mask);
                mask &= ~(0x80000000u >> b);
                if (two) {                                   // one code path for both segments (selects, no divergent branch)
                  const bool t1 = ft[j0 + b] != 0;
                  unsigned long long wc = t1 ? w1 : w;
                  int mc = t1 ? m1 : m;
                  wc = (wc >> 16) | ((unsigned long long) ((j0 + b) * NBR_SCALE) << 48);
                  mc++; mreal++;
                  if ((mc & 3) == 0) {
                    if ((mc >> 2) <= a.W4) lp[(size_t) (t1 ? a.W4 - (mc >> 2) : (mc >> 2) - 1) * a.tile_cap] = wc;
                    wc = dummy64;
                  }
                  w1 = t1 ? wc : w1; m1 = t1 ? mc : m1;
                  w = t1 ? w : wc; m = t1 ? m : mc;
                  continue;
                }
                if (typed && ft[j0 + b] != pt) continue;
                w = (w >> 16) | ((unsigned long long) ((j0 + b) * NBR_SCALE) << 48);
                m++; if (typed) mreal++;
                if ((m & 3) == 0) {
                  if ((m >> 2) <= a.W4) lp[(size_t) ((m >> 2) - 1) * a.tile_cap] = w;
                  w = dummy64;
                }
              }
            }
          }
      if (!typed) mreal = m;
      if ((m & 3) != 0) {                                      // the segment's (the row's) last word: dummies in the remaining slots
        for (int q = m & 3; q < 4; q++) w = (w >> 16) | ((unsigned long long) dummy << 48);
        if ((m >> 2) < a.W4) lp[(size_t) (m >> 2) * a.tile_cap] = w;
        m = (m + 3) & ~3; w = dummy64;
      }
      if (typed && pt < a.ntypes - 1) segw = (segw & ~(0xffu << (8 * pt))) | ((unsigned) (m >> 2 < 255 ? m >> 2 : 255) << (8 * pt));
      }
      if (two) {
        if ((m1 & 3) != 0) {
          for (int q = m1 & 3; q < 4; q++) w1 = (w1 >> 16) | ((unsigned long long) dummy << 48);
          m1 = (m1 + 3) & ~3;
          if ((m1 >> 2) <= a.W4) lp[(size_t) (a.W4 - (m1 >> 2)) * a.tile_cap] = w1;
        }
        // the move must not run into words it has yet to read: the row has to hold the second segment twice (else it counts as
        // full: make_nblist grows W and builds again)
        const int b0 = m >> 2, n1 = m1 >> 2;
        if (b0 + 2 * n1 <= a.W4) {
          for (int q = 0; q < n1; q++) lp[(size_t) (b0 + q) * a.tile_cap] = lp[(size_t) (a.W4 - 1 - q) * a.tile_cap];
          m += m1;
        } else m += 2 * m1;
      }
      a.nbr_cnt[i] = mreal < 4 * a.W4 ? mreal : 4 * a.W4;
      a.ownw[(size_t) T.tile * a.tile_cap + o] = (unsigned) slot | ((unsigned) (m < 4 * a.W4 ? m : 4 * a.W4) << 16);
      if (typed) a.ownseg[(size_t) T.tile * a.tile_cap + o] = segw;
      mmax = m > mmax ? m : mmax;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int v = __shfl_down(mmax, o, 64); mmax = v > mmax ? v : mmax; }
  if ((threadIdx.x & 63) == 0 && mmax > 0) atomicMax(&a.counters[CNT_MAXNBR], mmax);
}

// ---- inner loops of the tiled kernels ------------------------------------------------------------
// Branch-free and written in phases over UNR neighbours (all LDS reads of a phase are issued
// back-to-back, then the arithmetic), because with one ~140 KB workgroup per CU only ~3 waves per
// SIMD are resident and instruction-level parallelism has to hide the LDS latency.
// Table interpolation = PAIR_INT2 / VAL_FUNC2 / DERIV_FUNC2 (potaccess.h:323-354,465-495,591-621) with
// the argument clamped into the table (2 pad rows, imd_potential.c:1272) and the result masked by the
// cut-off test.  Products may fuse into FMAs here (relative change ~1e-16); the squared distance and
// every cut-off comparison use the reference's unfused expression.
#ifndef TILE_MAXTHREADS
#define TILE_MAXTHREADS 768
#endif
#ifdef IMDGPU_NOSCHED
#define SCHED_BARRIER
#else
#define SCHED_BARRIER __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef UNR
#define UNR 4                     // neighbours in flight per wave and phase (4 = one list word; 2 = half a word: fewer registers)
#endif
// experiment (make EXTRA=-DIMDGPU_LDSPRIO): raised wave priority while a wave issues its LDS gathers
#ifdef IMDGPU_LDSPRIO
#define PRIO_HI __builtin_amdgcn_s_setprio(2)
#define PRIO_LO __builtin_amdgcn_s_setprio(0)
#else
#define PRIO_HI
#define PRIO_LO
#endif

// Tile loop plumbing shared by the force kernels (staging: "staging AHEAD" below).  The first list words and the
// (slot, count) word of the thread's own atom do not depend on the staged positions: they are requested before
// the barriers, so their latency is hidden as well.
#define OWN_PROLOGUE                                                                                             \
      const uint2 *lp = (const uint2 *) (a.nbr16 + lbase + o);                                                   \
      unsigned ow = ow0, c0x = w0.x, c0y = w0.y, c1x = w1.x, c1y = w1.y;   /* scalars: uint2 selects end up in scratch */ \
      if (o != (int) threadIdx.x) {                                                                              \
        ow = a.ownw[(size_t) T.tile * a.tile_cap + o];                                                           \
        const uint2 q0 = lp[0], q1 = lp[(size_t) a.tile_cap];                                                    \
        c0x = q0.x; c0y = q0.y; c1x = q1.x; c1y = q1.y;                                                          \
      }                                                                                                          \
      const int slot = (int) (ow & 0xffffu), n4 = DBG(a, 1) ? 0 : ((int) (ow >> 16) + 3) >> 2;                \
      if (n4 < 1) { c0x = dw; c0y = dw; }                                                                        \
      if (n4 < 2) { c1x = dw; c1y = dw; }
// The list walk.  LIST_WORDS_BEGIN opens the per-word body (a lambda over one list word = four slots, defines j[UNR]);
// LIST_WORDS_END(RING) closes it and drives it over the words of the lane's row in one of two ways:
//  RING = 0  two words (8 slots) are consumed per pair of trips while the next two are in flight and handed over every
//            second trip; per-lane trip counts.  Fewest registers: pass 3 beside the TTM chain (<= 144 VGPRs, see there).
//  RING = 4  the words live in a ring of four registers with static names: word w is consumed from ring[w % 4] and that
//            register is re-loaded at once with word w + 4, so three loads of 8 B per lane are in flight per wave and no
//            hand-over waits for the newest one.  The loop runs to the wave's longest list (uniform branches; lanes past
//            their own count walk dummy words, which costs the SIMD nothing).  Measured -8 % (pass 1) / -4 % (pass 3):
//            the list is a 2 GB stream per pass and its latency, not only the LDS gathers, is what the waves wait for.
// Lanes past their own count get the dummy word either way.
// experiment (make EXTRA=-DWAVE_FAIR): a wave's priority falls as it gets through its list (3, 2, 1, 0 by quarters), so the waves
// a SIMD would serve last (it prefers its oldest wave) catch up and the workgroup's waves reach the barrier together
#if defined(WAVE_FAIR) && WAVE_FAIR >= 2
// WAVE_FAIR = n >= 2: the first n waves keep the top priority (they finish early and do the staging ahead), the others share fairly
#define WAVE_FAIR_BEGIN const bool wf_early = (int) (threadIdx.x >> 6) < WAVE_FAIR; const int wf1 = (nwu + 2) / 3, wf2 = (2 * nwu + 2) / 3; if (wf_early) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
#define WAVE_FAIR_STEP if (!wf_early && w >= wf1) { if (w >= wf2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
#define WAVE_FAIR_END __builtin_amdgcn_s_setprio(0);
#elif defined(WAVE_FAIR)
#define WAVE_FAIR_BEGIN const int wf1 = (nwu + 3) >> 2, wf2 = (nwu + 1) >> 1, wf3 = (3 * nwu + 3) >> 2; __builtin_amdgcn_s_setprio(3);
#define WAVE_FAIR_STEP if (w >= wf1) { if (w >= wf3) __builtin_amdgcn_s_setprio(0); else if (w >= wf2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
#define WAVE_FAIR_END __builtin_amdgcn_s_setprio(0);
#else
#define WAVE_FAIR_BEGIN
#define WAVE_FAIR_STEP
#define WAVE_FAIR_END
#endif
template <int S> struct RingSlot { static constexpr int value = S; };
#define LIST_WORDS_BEGIN                                                                                         \
      auto body = [&](const unsigned curx, const unsigned cury, auto ring_slot) __attribute__((always_inline)) { \
        constexpr int RSLOT = decltype(ring_slot)::value;       /* which of the four ring registers this word came from */ \
        (void) RSLOT;                                                                                            \
        const int j4[4] = {(int) (curx & 0xffffu), (int) (curx >> 16), (int) (cury & 0xffffu), (int) (cury >> 16)};     \
        HALF_BEGIN
#define RING_LOAD(DX, DY, W) { DX = dw; DY = dw; if ((W) < n4) { const uint2 q = lp[(size_t) (W) * a.tile_cap]; DX = q.x; DY = q.y; } }
#define RING_STEP(RX, RY, S) if (w >= nwu) break; body(RX, RY, RingSlot<S>{}); RING_LOAD(RX, RY, w + 4) w++;
#define LIST_WORDS_END(RING)                                                                                     \
      };                                                                                                         \
      if constexpr ((RING) == 0) {                                                                               \
        unsigned m0x = dw, m0y = dw, m1x = dw, m1y = dw;                                                         \
        for (int k4 = 0; k4 < n4; k4++) {                                                                        \
          if ((k4 & 1) == 0) {                    /* even trip: issue the next pair of loads */                  \
            RING_LOAD(m0x, m0y, k4 + 2) RING_LOAD(m1x, m1y, k4 + 3)                                              \
          }                                                                                                      \
          const unsigned curx = (k4 & 1) ? c1x : c0x, cury = (k4 & 1) ? c1y : c0y;                               \
          if (k4 & 1) { c0x = m0x; c0y = m0y; c1x = m1x; c1y = m1y; }   /* odd trip: hand over (loads are ~8 neighbours old) */ \
          body(curx, cury, RingSlot<0>{});                                                                       \
        }                                                                                                        \
      } else {                                                                                                   \
        int nwu = n4;                                                                                            \
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(nwu, o, 64); nwu = t > nwu ? t : nwu; }  \
        nwu = __builtin_amdgcn_readfirstlane(nwu);                                                               \
        unsigned r2x, r2y, r3x, r3y;                                                                             \
        RING_LOAD(r2x, r2y, 2) RING_LOAD(r3x, r3y, 3)                                                            \
        int w = 0;                                                                                               \
        WAVE_FAIR_BEGIN                                                                                          \
        for (;;) { WAVE_FAIR_STEP RING_STEP(c0x, c0y, 0) RING_STEP(c1x, c1y, 1) RING_STEP(r2x, r2y, 2) RING_STEP(r3x, r3y, 3) } \
        WAVE_FAIR_END                                                                                            \
      }
#if UNR == 4
#define HALF_BEGIN { const int *j = j4;
#define HALF_END }
#else
#define HALF_BEGIN _Pragma("unroll") for (int hh = 0; hh < 4 / UNR; hh++) { const int j[2] = {j4[2 * hh], j4[2 * hh + 1]};
#define HALF_END SCHED_BARRIER; }
#endif
// ---- staging AHEAD, by the waves that are through first (the one-type force kernels) --------------------------------------------
// The waves of a workgroup do not finish a tile together: the SIMDs serve their oldest wave first, wave 0 is through with its atoms
// a third earlier than the last one (phase timing, make EXTRA=-DPHASE_TIMING: 0.94 vs 1.37 ms of the pass) and used to idle at the
// barrier, after which ALL waves went through the staging's two dependent round trips (slot -> atom index, atom -> position)
// together: 3.2 us of a 30-us tile with the LDS pipe idle.  Now a wave that is through takes the next SA chunks of 64 slots of the
// NEXT tile from a counter in the LDS, loads their indices and positions into registers (free once its neighbour loop is over)
// and goes to the barrier; the first ceil(slots / 64 / SA) waves to finish cover the whole tile, the last ones find the counter
// exhausted and arrive at the barrier with nothing to wait for.  Behind the barrier the held records are written to the planes;
// chunks nobody took (more than SA * waves: tiles beyond autotile's shapes) are staged by a plain loop.  Two counters, used in
// turns: the one the next grab will use is zeroed between the barriers of the tile before.
#ifndef SA
#define SA 8
#endif
#define AHEAD_DECL(SCRATCH)                                                                                      \
  int ah_c0 = 0, ah_par = 0; int ah_idx[SA], ah_T[SA]; double4 ah_P[SA];                                          \
  _Pragma("unroll") for (int u = 0; u < SA; u++) { ah_idx[u] = -1; ah_T[u] = 0; ah_P[u] = make_double4(0, 0, 0, 0); } \
  int *ah_cnt = (int *) ((SCRATCH) + 60);                                                                        \
  if (threadIdx.x == 0) { ah_cnt[0] = 0; ah_cnt[1] = 0; }                                                         \
  __syncthreads();
#define AHEAD_GRAB(NT, NSL, WITH_T)                                                                              \
  {                                                                                                              \
    int c_ = 0;                                                                                                  \
    if ((threadIdx.x & 63) == 0) c_ = atomicAdd(&ah_cnt[ah_par], SA);                                            \
    ah_c0 = __builtin_amdgcn_readfirstlane(c_);                                                                  \
    /* every record is assigned on every path (a guarded load, or zeros): otherwise the registers count as live through the */ \
    /* neighbour loop, which has none to spare.  A wave that finds the counter exhausted -- the LAST waves to finish, the ones */ \
    /* the workgroup waits for -- issues no load at all */                                                       \
    if (ah_c0 * 64 < (NSL)) {                                                                                    \
      const int *src_ = a.tsrc + (size_t) (NT) * SLOTCAP(a);                                                     \
      _Pragma("unroll") for (int u = 0; u < SA; u++) { const int s_ = (ah_c0 + u) * 64 + (int) (threadIdx.x & 63); ah_idx[u] = s_ < (NSL) ? src_[s_] : -1; } \
      _Pragma("unroll") for (int u = 0; u < SA; u++) ah_P[u] = a.pos[ah_idx[u] >= 0 ? ah_idx[u] : 0];           \
      if (WITH_T) { _Pragma("unroll") for (int u = 0; u < SA; u++) ah_T[u] = a.type[ah_idx[u] >= 0 ? ah_idx[u] : 0]; } \
    } else { _Pragma("unroll") for (int u = 0; u < SA; u++) { ah_idx[u] = -1; ah_T[u] = 0; ah_P[u] = make_double4(0, 0, 0, 0); } } \
  }
#define TILE_PROLOGUE_A(WITH_W, WITH_T, SW, TS, WSCALE, PRE, BETWEEN, MID, NEXT)                                 \
    T.own_start = info.y; T.nown = info.z;                                                                       \
    const unsigned dw = (unsigned) ((SLOTCAP(a) - 1) * NBR_SCALE) * 0x10001u;                                    \
    const uint2 dummyw = make_uint2(dw, dw);                                                                     \
    const size_t lbase = (size_t) T.tile * a.W4 * a.tile_cap;                                                    \
    unsigned ow0 = 0; uint2 w0 = dummyw, w1 = dummyw;                                                            \
    if ((int) threadIdx.x < T.nown) {                                                                            \
      const uint2 *lq = (const uint2 *) (a.nbr16 + lbase + threadIdx.x);                                         \
      ow0 = a.ownw[(size_t) T.tile * a.tile_cap + threadIdx.x]; w0 = lq[0]; w1 = lq[(size_t) a.tile_cap];       \
    }                                                                                                            \
    PRE                                                                                                          \
    __syncthreads();                              /* every wave is done with the previous tile's LDS */          \
    BETWEEN                                                                                                      \
    {                                                                                                            \
      double *const sw_ = (double *) (SW); int *const ts_ = (int *) (TS);                                        \
      _Pragma("unroll") for (int u = 0; u < SA; u++)                                                             \
        if (ah_idx[u] >= 0) {                                                                                    \
          const int s_ = (ah_c0 + u) * 64 + (int) (threadIdx.x & 63);                                            \
          if (WITH_W) LPOS_STW(sp, sw_, SLOTCAP(a), s_, ah_P[u].x, ah_P[u].y, ah_P[u].z, ah_P[u].w * (WSCALE))    \
          else LPOS_ST(sp, sw_, SLOTCAP(a), s_, ah_P[u].x, ah_P[u].y, ah_P[u].z)                                  \
          if (WITH_T) ts_[s_] = ah_T[u];                                                                         \
        }                                                                                                        \
      const int cfin_ = ah_cnt[ah_par];           /* every wave has grabbed: chunks from here on are nobody's */ \
      const int *src_ = a.tsrc + (size_t) T.tile * SLOTCAP(a);                                                   \
      for (int s_ = cfin_ * 64 + (int) threadIdx.x; s_ < info.x; s_ += blockDim.x) {                             \
        const int gi_ = src_[s_];                                                                                \
        const double4 q_ = a.pos[gi_];                                                                           \
        if (WITH_W) LPOS_STW(sp, sw_, SLOTCAP(a), s_, q_.x, q_.y, q_.z, q_.w * (WSCALE))                          \
        else LPOS_ST(sp, sw_, SLOTCAP(a), s_, q_.x, q_.y, q_.z)                                                   \
        if (WITH_T) ts_[s_] = a.type[gi_];                                                                       \
      }                                                                                                          \
      if (threadIdx.x == 0) {                     /* the far-away dummy that pads the neighbour rows */          \
        if (WITH_W) LPOS_STW(sp, sw_, SLOTCAP(a), SLOTCAP(a) - 1, 1e150, 1e150, 1e150, 0.0)                       \
        else LPOS_ST(sp, sw_, SLOTCAP(a), SLOTCAP(a) - 1, 1e150, 1e150, 1e150)                                    \
        if (WITH_T) ts_[SLOTCAP(a) - 1] = 0;                                                                     \
        ah_cnt[ah_par ^ 1] = 0;                                                                                  \
      }                                                                                                          \
    }                                                                                                            \
    MID                                                                                                          \
    __syncthreads();                                                                                             \
    ah_par ^= 1;                                                                                                 \
    const int next_tile = NEXT;                                                                                  \
    int4 info_next = info;                                                                                       \
    if (next_tile >= 0) info_next = a.tinfo[next_tile];
#define TILE_EPILOGUE_A(WITH_T)                                                                                  \
    if (next_tile >= 0) AHEAD_GRAB(next_tile, info_next.x, WITH_T)                                               \
    else { _Pragma("unroll") for (int u = 0; u < SA; u++) { ah_idx[u] = -1; ah_T[u] = 0; ah_P[u] = make_double4(0, 0, 0, 0); } }  \
    info = info_next;

struct TabK { double begin, end, istep, istep2; };      // one column of a table (NT1 fast path)

__device__ __forceinline__ void tab_index(const TabK &c, double arg, int &k, double &chi)
{
  double a = fmax(fmin(arg, c.end) - c.begin, 0.0) * c.istep;
  k = (int) a;
  chi = a - (double) k;
}

// the same index for callers that send every argument beyond `end` to the zero rows themselves (no upper clamp needed);
// chi = a - trunc(a) is exact, and so is v_fract_f64 for a >= 0: the same bits with one instruction less
// sgn collects the sign bits of (arg - begin): negative once any argument lay below the table (is_short, potaccess.h:331) --
// one integer OR per neighbour instead of a running fp64 minimum
__device__ __forceinline__ void tab_index_open(const TabK &c, double arg, int &k, double &chi, int &sgn)
{
  const double d = arg - c.begin;
  sgn |= __double2hiint(d);
  const double a = fmax(d, 0.0) * c.istep;
  k = (int) a;
  chi = __builtin_amdgcn_fract(a);
}

// ... and with (arg - begin) * istep as ONE fused multiply-add, a = arg * istep + c0, c0 = -begin * istep: one instruction less per
// look-up.  `a` differs from the reference's product by a rounding error; it matters only where a lies within that error of an
// integer, i.e. of a table node, where the three-point rule is continuous (value) or jumps by a third difference (gradient):
// ~1e-16 relative, seven orders inside the 1e-9 bar.  sgn collects the sign of a: the is_short flag may differ from the
// reference's for a pair whose r^2 lies within one rounding error of the table's begin.  (The force passes' neighbour loops are
// bound by the NUMBER of vector instructions, not by the LDS: the distinct-grids form has 50 % more LDS reads and 29 % more vector
// instructions than the shared-grid form and takes 31 % longer -- profiles/r05_pass1_forms.json.)
// r^2 in the one-type pass 1.  EXACT = true: the reference's unfused expression (SPROD3D, makros.h:361) -- needed when the pass
// counts neighbours (numneighs of the TTM, an integer decided by r2 <= pair_pot.end[0], imd_forces_nbl.c:473).  Without that count
// nothing in the pass is a decision on r2 that is not continuous in it (a pair a rounding error beyond the cut-off contributes a
// term that vanishes there), and two fused multiply-adds save two of the loop's 40 vector instructions.
template <bool EXACT>
__device__ __forceinline__ double r2_of(double dx, double dy, double dz)
{
  if constexpr (EXACT) return dx * dx + dy * dy + dz * dz;
  else return fma(dz, dz, fma(dy, dy, dx * dx));
}
struct TabF { double c0, istep, end; };
__device__ __forceinline__ TabF tabf(const TabPar &t, int col) { return {-t.begin[col] * t.invstep[col], t.invstep[col], t.end[col]}; }
__device__ __forceinline__ void tab_index_fma(const TabF &c, double arg, int &k, double &chi, int &sgn)
{
  double a = fma(arg, c.istep, c.c0);
  sgn |= __double2hiint(a);
  a = fmax(a, 0.0);
  k = (int) a;
  chi = __builtin_amdgcn_fract(a);
}

template <bool PAIR, bool EAM, bool TTMN, bool NT1>
__global__ __launch_bounds__(TILE_MAXTHREADS) void k_pass1_tile(TileArgs a)
{
  extern __shared__ double lds[];
  if (a.guard && *a.guard) return;
  double *tp = lds;
  double *tr = tp + (PAIR ? a.pair.nrows * a.pair.ncols : 0);
  double *sp = tr + (EAM ? a.rho.nrows * a.rho.ncols : 0);
  sp += ((size_t) sp >> 3) & 1;                         // 16-byte records (LPOS_*): tables may hold an odd number of doubles
  double *scratch = sp + 3 * (size_t) SLOTCAP(a);
  int *ts = (int *) (scratch + 64);
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  double sums[2] = {0.0, 0.0};
  int is_short = 0;
  if (t0 < t1) {                                 // tables are loaded once per (persistent) workgroup
    if (PAIR) lds_load_table(tp, a.pair);
    if (EAM) lds_load_table(tr, a.rho);
  }
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, !NT1) }
  for (T.tile = t0; T.tile < t1; T.tile++) {     // uniform per block
    TILE_PROLOGUE_A(false, !NT1, nullptr, ts, 1.0, , , , (T.tile + 1 < t1 ? T.tile + 1 : -1))
    const int nt = a.ntypes, inc = NT1 ? 1 : nt * nt;
    const double pend0 = PAIR ? a.pair.end[0] : 0.0;
    const TabK cp = {a.pair.begin[0], a.pair.end[0], a.pair.invstep[0], 2 * a.pair.invstep[0]};
    const TabK cr = {a.rho.begin[0], a.rho.end[0], a.rho.invstep[0], 2 * a.rho.invstep[0]};
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      OWN_PROLOGUE
      double xi, yi, zi;
      LPOS_RD(sp, sp, SLOTCAP(a), slot, xi, yi, zi)
      const int it = NT1 ? 0 : ts[slot];
      double fx = 0, fy = 0, fz = 0, ee = 0, rr = 0, vir = 0, r2min = 1e300;
      int nb = 0;
      // list words: PFD groups of four slots in flight (only ~3 waves/SIMD are resident, so the HBM stream
      // of the list needs memory-level parallelism from each wave)
      // two list words (8 slots) are consumed per trip while the next two are in flight; lanes past their
      // own count get the dummy word
      LIST_WORDS_BEGIN
        double dx[UNR], dy[UNR], dz[UNR], r2[UNR];
#pragma unroll
        for (int e = 0; e < UNR; e++) {                                   // phase 1: positions from LDS, all reads first
          nbr_rd<0>(sp, SLOTCAP(a), j[e], dx[e], dy[e], dz[e]);
        }
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
          r2[e] = dx[e] * dx[e] + dy[e] * dy[e] + dz[e] * dz[e];          // unfused: the reference's r2
        }
        if (NT1) {
          int kp[UNR], kr[UNR];
          double cp_[UNR], cr_[UNR], P0[UNR], P1[UNR], P2[UNR], R0[UNR], R1[UNR], R2[UNR];
#pragma unroll
          for (int e = 0; e < UNR; e++) {                                 // phase 2: table indices, then all row reads
            r2min = fmin(r2min, r2[e]);
            if (PAIR) tab_index(cp, r2[e], kp[e], cp_[e]);
            if (EAM) tab_index(cr, r2[e], kr[e], cr_[e]);
          }
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < UNR; e++) {
            if (PAIR) { P0[e] = tp[kp[e]]; P1[e] = tp[kp[e] + 1]; P2[e] = tp[kp[e] + 2]; }
            if (EAM) { R0[e] = tr[kr[e]]; R1[e] = tr[kr[e] + 1]; R2[e] = tr[kr[e] + 2]; }
          }
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < UNR; e++) {                                 // phase 3: arithmetic
#pragma clang fp contract(fast)
            if (TTMN) nb += (r2[e] <= pend0) ? 1 : 0;
            if (PAIR) {
              const double chi = cp_[e], dv = P1[e] - P0[e], d2v = (P2[e] - 2 * P1[e]) + P0[e];
              const double pot = P0[e] + chi * dv + (0.5 * chi) * (chi - 1) * d2v;     // = 0 at/after the cut-off (shifted table)
              double grad = cp.istep2 * (dv + (chi - 0.5) * d2v);
              grad = (r2[e] <= cp.end) ? grad : 0.0;
              fx += dx[e] * grad; fy += dy[e] * grad; fz += dz[e] * grad;
              ee += (r2[e] <= cp.end) ? 0.5 * pot : 0.0;
              vir -= 0.5 * (r2[e] * grad);
            }
            if (EAM) {
              const double chi = cr_[e], dv = R1[e] - R0[e], d2v = (R2[e] - 2 * R1[e]) + R0[e];
              const double val = R0[e] + chi * dv + (0.5 * chi) * (chi - 1) * d2v;
              rr += (r2[e] < cr.end) ? val : 0.0;
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
            const int jt = nbr_type(ts, j[e]), col = it * nt + jt;
            if (TTMN) nb += (r2[e] <= pend0 && it == 0 && jt == 0) ? 1 : 0;
            if (PAIR) {
              const TabK c = {a.pair.begin[col], a.pair.end[col], a.pair.invstep[col], 2 * a.pair.invstep[col]};
              int k; double chi;
              tab_index(c, r2[e], k, chi);
              is_short |= (r2[e] < c.begin);
              const double *q = tp + k * inc + col;
              const double p0 = q[0], p1 = q[inc], p2 = q[2 * inc], dv = p1 - p0, d2v = (p2 - 2 * p1) + p0;
              const double pot = p0 + chi * dv + (0.5 * chi) * (chi - 1) * d2v;
              double grad = c.istep2 * (dv + (chi - 0.5) * d2v);
              grad = (r2[e] <= c.end) ? grad : 0.0;
              fx += dx[e] * grad; fy += dy[e] * grad; fz += dz[e] * grad;
              ee += (r2[e] <= c.end) ? 0.5 * pot : 0.0;
              vir -= 0.5 * (r2[e] * grad);
            }
            if (EAM) {
              const TabK c = {a.rho.begin[col], a.rho.end[col], a.rho.invstep[col], 2 * a.rho.invstep[col]};
              int k; double chi;
              tab_index(c, r2[e], k, chi);
              is_short |= (r2[e] < c.begin);
              const double *q = tr + k * inc + col;
              const double p0 = q[0], p1 = q[inc], p2 = q[2 * inc], dv = p1 - p0, d2v = (p2 - 2 * p1) + p0;
              const double val = p0 + chi * dv + (0.5 * chi) * (chi - 1) * d2v;
              rr += (r2[e] < c.end) ? val : 0.0;
            }
          }
        }
        HALF_END
      LIST_WORDS_END(0)
      if (NT1) is_short |= ((PAIR && r2min < cp.begin) || (EAM && r2min < cr.begin)) ? 1 : 0;
      a.f[i] = fx; a.f[a.cap + i] = fy; a.f[2 * (size_t) a.cap + i] = fz;
      a.epot[i] = ee;
      if (EAM) a.rhoh[i] = rr;
      if (TTMN) a.numneighs[i] = nb;
      sums[0] += ee; sums[1] += vir;
    }
    TILE_EPILOGUE_A(!NT1)
  }
  block_sum<2>(sums, scratch);
  if (threadIdx.x == 0) {
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 0] = sums[0];
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 1] = sums[1];
  }
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}

// Pass 1, one atom type, pair_pot and rho_h_tab tabulated on the SAME r^2 grid (the usual case for EAM
// files written by one tool): the two tables are interleaved in LDS as (phi[k], rho[k]) so that one index
// computation and three aligned 16-byte reads serve both; three all-zero rows follow the table and
// neighbours beyond the cut-off (list skin, padding) are sent there, which replaces every mask.
// FEMB: the embedding pass (pass 2, imd_forces_nbl.c:748-783) is folded into the epilogue -- with the full list rho of an
// own atom is complete when its neighbour loop ends, so F(rho) and F'(rho) can be taken right there: no k_pass2 launch,
// rho is not re-read, epot is written once.
// DYN: the tiles come from a work queue instead of a fixed range per workgroup (see TileArgs::tile_map).  The position halo of a
// multi-rank step is still in flight when this pass starts: launch A takes the interior tiles and ends; launch B, behind the
// halo's event, takes the boundary tiles.  (A first version let A's workgroups go on with boundary tiles once a device word said
// the halo had landed: the few workgroups still busy at that moment then did ALL boundary tiles alone -- 16 ms instead of 2 --
// while B waited behind A in stream order.  A kernel boundary also makes the unpacked ghost atoms visible without any fence.)
// Chunks of consecutive positions (tiles that are neighbours along z: their halos overlap in L1/L2), DYN_CHUNK at first, fewer
// as a range runs empty (a chunk is 0.03 ms of work per tile: the last chunks decide how long the slowest workgroup runs); one
// counter per XCD and class, other XCDs' ranges when the own one is empty.  Energies are summed per workgroup in the order the
// queue hands the tiles out: they differ from run to run in the last bits (forces, rho, list checks do not).
// make EXTRA="-DPHASE_TIMING [-DPH_THREAD=640]": where one thread of every workgroup of k_pass1_tile_merged spends its time (100 MHz
// clock, summed over the workgroups and the launches of a run, printed by imdgpu_destroy): 0 loop top, 1 first loads issued + the previous
// atom's embedding finished, 2 first barrier, 3 staging, 4 second barrier (+ next tile's info), 5 the own atom, 6 epilogue
#ifdef PHASE_TIMING
#ifndef PH_THREAD
#define PH_THREAD 0
#endif
__device__ unsigned long long g_phase[8];
#define PH_DECL unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, ph_last = wall_clock64();
#define PH(n) if (threadIdx.x == PH_THREAD) { const unsigned long long t_ = wall_clock64(); ph[n] += t_ - ph_last; ph_last = t_; }
#define PH_END if (threadIdx.x == PH_THREAD) { for (int q_ = 0; q_ < 7; q_++) atomicAdd(&g_phase[q_], ph[q_]); atomicAdd(&g_phase[7], 1ull); }
#else
#define PH_DECL
#define PH(n)
#define PH_END
#endif
#define DYN_CHUNK 8
__device__ __forceinline__ int dyn_next_tile(const int *tile_map, int pos) { return pos >= 0 ? __builtin_amdgcn_readfirstlane(tile_map[pos]) : -1; }
// DG ("distinct grids"): pair_pot and rho_h_tab each on its OWN r^2 grid (begin, end, step and row count per table,
// types.h:537-549; three files with three headers, read_pot_table2, imd_potential.c:394-466).  The two tables then sit one behind the
// other in the LDS, each followed by its three zero rows; a neighbour costs two index computations and six 8-byte row reads instead
// of one and three 16-byte ones, everything else -- staging ahead, the folded embedding pass, the row report for the EAM force
// pass (in rows of rho_h_tab, the table that pass reads) -- is shared.  Cut-offs as the reference has them: pair r2 <= end
// (imd_forces_nbl.c:489), density r2 < end (:550), neighbour count r2 <= pair_pot.end[0] (:473).
template <bool TTMN, bool FEMB, int CAP, bool DYN = false, bool DG = false>
__global__ __launch_bounds__(TILE_MAXTHREADS) void k_pass1_tile_merged(TileArgs a)
{
  extern __shared__ double lds[];
  if (a.guard && *a.guard) return;
  static_assert(!(DG && DYN), "the tile queue exists for the shared-grid form only");
  const int nrows = a.pair.nrows, kz = nrows;
  const int nrows_r = a.rho.nrows, kzr = nrows_r;
  double *sp = lds;                                             // the planes first: a list entry is the address of its x read (NBR_SCALE)
  double *scratch = sp + 3 * (size_t) SLOTCAP(a);
  double2 *tm = (double2 *) (scratch + 64);                     // (nrows + 3) rows; slot_cap is a multiple of 64: 16-byte aligned
  double *tp = (double *) tm, *tr = tp + nrows + 3;             // DG: pair_pot's rows + 3 zero rows, then rho_h_tab's + 3
  (void) tp; (void) tr; (void) kzr;
  TileCtx T;
  int t0 = 0, t1 = 0;
  // ---- DYN: the queue ------------------------------------------------------------------------------------------------
  // the chunk thread 0 has just taken: [begin, end) positions, or -1.  In the tail of the reduction scratch (block_sum<3> uses 48 of
  // its 64 doubles, and only after the loop) -- NOT a static __shared__ array: that would move the dynamic region by 8 bytes and
  // every 16-byte table read (ds_read_b128) off its alignment, which costs the pass a factor of three
  int *s_dyn = (int *) (scratch + 62);
  int q_phase = 0, q_k = 0;                // thread 0: class being drained (0 interior, 1 boundary), XCDs tried so far in it
  auto grab = [&]() {                      // thread 0 only
    const int x0 = blockIdx.x & 7, nI = a.n_interior, nB = a.ntiles - a.n_interior;
    for (;;) {
      const int n = q_phase ? nB : nI, base = q_phase ? nI : 0;
      for (; q_k < 8; q_k++) {
        const int x = (x0 + q_k) & 7;
        const int lo = (int) ((long long) n * x / 8), hi = (int) ((long long) n * (x + 1) / 8);
        int *cnt = &a.dyn[8 * q_phase + x];
        const int left = hi - lo - __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // a hint: may be stale
        if (left <= 0) continue;
        const int c = left >= 64 * DYN_CHUNK ? DYN_CHUNK : (left >= 64 ? left >> 6 : 1);                      // ~ left / (2 x workgroups of an XCD)
        const int p = lo + atomicAdd(cnt, c);
        if (p < hi) { s_dyn[0] = base + p; s_dyn[1] = base + (p + c < hi ? p + c : hi); return; }
      }
      if (q_phase >= a.dyn_last) break;
      q_phase++; q_k = 0;
    }
    s_dyn[0] = -1; s_dyn[1] = -1;
  };
  int pos = -1, pos_end = -1;
  if constexpr (DYN) {
    if (threadIdx.x == 0) grab();
    __syncthreads();
    pos = __builtin_amdgcn_readfirstlane(s_dyn[0]); pos_end = __builtin_amdgcn_readfirstlane(s_dyn[1]);     // uniform: scalar registers
    t1 = pos >= 0 ? 1 : 0;                 // "this workgroup has work": the table is loaded below
  } else
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  double sums[3] = {0.0, 0.0, 0.0};        // pair energy, virial, embedding energy
  int is_short = 0;
  int kmn = 0x7fffffff;                    // lowest table row of any pair: tells the EAM force pass which rows it needs
  if (t0 < t1) {
    if constexpr (DG) {
      for (int k = threadIdx.x; k < nrows + 3; k += blockDim.x) tp[k] = k < nrows ? a.pair.tab[k] : 0.0;
      for (int k = threadIdx.x; k < nrows_r + 3; k += blockDim.x) tr[k] = k < nrows_r ? a.rho.tab[k] : 0.0;
    } else {
    const double2 *src = (const double2 *) a.merged;
    for (int k = threadIdx.x; k < nrows + 3; k += 4 * blockDim.x) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < nrows + 3) v[u] = src[q]; }
#pragma unroll
      for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < nrows + 3) tm[q] = v[u]; }
    }
    }
  }
  const TabK c = {a.pair.begin[0], a.pair.end[0], a.pair.invstep[0], 2 * a.pair.invstep[0]};
  const TabF cf = tabf(a.pair, 0), crf = tabf(a.rho, 0);
  (void) crf;
  // embedding energy and its derivative of the atom just finished (FEMB), completed one tile later: see the end of the atom loop
  const double emb_begin = FEMB ? a.emb.begin[0] : 0, emb_end = FEMB ? a.emb.end[0] : 0, emb_istep = FEMB ? a.emb.invstep[0] : 0;
  int pend_i = -1;
  double pend_chi = 0, pend_p0 = 0, pend_p1 = 0, pend_p2 = 0, pend_ee = 0;
  auto emb_finish = [&]() {
    if (FEMB && pend_i >= 0) {
      Tab3 q;
      q.p0 = pend_p0; q.dv = pend_p1 - pend_p0; q.d2v = pend_p2 - 2 * pend_p1 + pend_p0; q.chi = pend_chi; q.istep = emb_istep;
      const double pe = tab_val(q);
      a.posw[pend_i].w = tab_grad(q);
      a.epot[pend_i] = pend_ee + pe;
      sums[2] += pe;
      pend_i = -1;
    }
  };
  PH_DECL
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if constexpr (DYN) { T.tile = pos >= 0 ? __builtin_amdgcn_readfirstlane(a.tile_map[pos]) : 0; t0 = 0; t1 = pos >= 0 ? 1 : 0; if (pos >= 0) { info = a.tinfo[T.tile]; AHEAD_GRAB(T.tile, info.x, false) } }
  else { if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, false) } T.tile = t0; }
  while (DYN ? pos >= 0 : T.tile < t1) {
    // DYN: on the last tile of its chunk thread 0 takes the next chunk between the prologue's barriers (everybody is through
    // with the previous iteration's read of s_dyn by then), all threads read it behind the second barrier
    const bool last_of_chunk = DYN && pos + 1 >= pos_end;
    PH(0)
    TILE_PROLOGUE_A(false, false, nullptr, nullptr, 1.0,
                    PH(1),
                    PH(2) if (DYN && last_of_chunk && threadIdx.x == 0) grab();,
                    PH(3),
                    (DYN ? dyn_next_tile(a.tile_map, last_of_chunk ? __builtin_amdgcn_readfirstlane(s_dyn[0]) : pos + 1) : (T.tile + 1 < t1 ? T.tile + 1 : -1)))
    emb_finish();                        // the previous tile's atom: its rows have had both barriers' time to arrive
    PH(4)
    int npos = pos + 1, npos_end = pos_end;
    if (DYN && last_of_chunk) { npos = __builtin_amdgcn_readfirstlane(s_dyn[0]); npos_end = __builtin_amdgcn_readfirstlane(s_dyn[1]); }
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      OWN_PROLOGUE
      double xi, yi, zi;
      LPOS_RD(sp, sp, SLOTCAP(a), slot, xi, yi, zi)
      emb_finish();                                        // (a second own atom of this thread in one tile: never at autotile's shapes)
      double fx = 0, fy = 0, fz = 0, ee = 0, rr = 0, vir = 0, ee0 = 0, rr0 = 0;
      int nb = 0, sgn = 0;
      LIST_WORDS_BEGIN
        double dx[UNR], dy[UNR], dz[UNR], r2[UNR], chi[UNR];
        int k[UNR];
        PRIO_HI;
#pragma unroll
        for (int e = 0; e < UNR; e++) nbr_rd<CAP>(sp, SLOTCAP(a), j[e], dx[e], dy[e], dz[e]);
        PRIO_LO;
        SCHED_BARRIER;
        if constexpr (DG) {
          int kr[UNR];
          double chr[UNR], P0[UNR], P1[UNR], P2[UNR], R0[UNR], R1[UNR], R2[UNR];
#pragma unroll
          for (int e = 0; e < UNR; e++) {
            dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
            r2[e] = r2_of<TTMN>(dx[e], dy[e], dz[e]);
            tab_index_fma(cf, r2[e], k[e], chi[e], sgn);
            tab_index_fma(crf, r2[e], kr[e], chr[e], sgn);
          }
#if UNR == 4
          kmn = min(min(kmn, kr[0]), kr[1]); kmn = min(min(kmn, kr[2]), kr[3]);   // in rows of rho_h_tab: the EAM force pass's table
#else
          kmn = min(min(kmn, kr[0]), kr[1]);
#endif
#pragma unroll
          for (int e = 0; e < UNR; e++) {
            const bool in = r2[e] <= c.end;                                 // pair: <= end (imd_forces_nbl.c:489)
            if (TTMN) nb += in ? 1 : 0;
            k[e] = in ? k[e] : kz;
            kr[e] = r2[e] < crf.end ? kr[e] : kzr;                          // density: < end (:550)
          }
          SCHED_BARRIER;
          PRIO_HI;
#pragma unroll
          for (int e = 0; e < UNR; e++) {                                   // volatile: six ds_read_b64, not three ds_read2_b64 (FAST_CAP's note)
            lds_cvdouble *qp = (lds_cvdouble *) (tp + k[e]), *qr = (lds_cvdouble *) (tr + kr[e]);
            P0[e] = qp[0]; P1[e] = qp[1]; P2[e] = qp[2];
            R0[e] = qr[0]; R1[e] = qr[1]; R2[e] = qr[2];
          }
          PRIO_LO;
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
            // value = p0 + chi (dv + (chi - 1) / 2 d2v); gradient / (2 invstep) = dv + (chi - 1/2) d2v = gr; value = p0 + chi (gr - chi d2v / 2)
            const double dvp = P1[e] - P0[e], d2p = (P2[e] - 2 * P1[e]) + P0[e];
            const double dvr = R1[e] - R0[e], d2r = (R2[e] - 2 * R1[e]) + R0[e];
            const double gr = fma(chi[e] - 0.5, d2p, dvp);                    // scaled by 2 invstep once per atom
            ee = fma(chi[e], fma(-0.5 * chi[e], d2p, gr), ee); ee0 += P0[e];  // halved once per atom
            rr = fma(chr[e], fma(fma(0.5, chr[e], -0.5), d2r, dvr), rr); rr0 += R0[e];
            fx += dx[e] * gr; fy += dy[e] * gr; fz += dz[e] * gr;
            vir += r2[e] * gr;
          }
        } else {
        double2 A[UNR], B[UNR], C[UNR];
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
          r2[e] = r2_of<TTMN>(dx[e], dy[e], dz[e]);
          tab_index_fma(cf, r2[e], k[e], chi[e], sgn);
        }
#if UNR == 4
        kmn = min(min(kmn, k[0]), k[1]); kmn = min(min(kmn, k[2]), k[3]);  // v_min3_i32
#else
        kmn = min(min(kmn, k[0]), k[1]);
#endif
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          const bool in = r2[e] <= c.end;                                 // pair: <= end (imd_forces_nbl.c:489); rho(end) = 0 exactly
          if (TTMN) nb += in ? 1 : 0;
          k[e] = in ? k[e] : kz;
        }
        SCHED_BARRIER;
        PRIO_HI;
#pragma unroll
        for (int e = 0; e < UNR; e++) { A[e] = tm[k[e]]; B[e] = tm[k[e] + 1]; C[e] = tm[k[e] + 2]; }
        PRIO_LO;
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
          const double g = chi[e] - 0.5, h = fma(0.5 * g, g, -0.125);      // h = chi (chi - 1) / 2
          const double dvp = B[e].x - A[e].x, d2p = (C[e].x - 2 * B[e].x) + A[e].x;
          const double dvr = B[e].y - A[e].y, d2r = (C[e].y - 2 * B[e].y) + A[e].y;
          const double gr = dvp + g * d2p;                                  // grad / (2 invstep): scaled once per atom
          rr += A[e].y + chi[e] * dvr + h * d2r;
          fx += dx[e] * gr; fy += dy[e] * gr; fz += dz[e] * gr;
          ee += A[e].x + chi[e] * dvp + h * d2p;                            // halved once per atom
          vir += r2[e] * gr;
        }
        }
        HALF_END
      LIST_WORDS_END(4)
      if constexpr (DG) { ee += ee0; rr += rr0; }
      fx *= c.istep2; fy *= c.istep2; fz *= c.istep2; ee *= 0.5; vir *= -0.5 * c.istep2;
      is_short |= sgn < 0 ? 1 : 0;
      a.f[i] = fx; a.f[a.cap + i] = fy; a.f[2 * (size_t) a.cap + i] = fz;
      a.rhoh[i] = rr;
      if (FEMB) {                                           // PAIR_INT on embed_pot, :757-760; EAM_DF into pos.w
        // tab3's index and its three row loads now; value, gradient and their stores in emb_finish, in front of the next tile's
        // first barrier: the rows travel together with that tile's first loads instead of in a round trip of their own
        double ea = (rr < emb_end ? rr : emb_end) - emb_begin;
        ea = (ea < 0 ? 0 : ea) * emb_istep;
        const int ek = (int) ea;
        const double *ep = a.emb.tab + ek;
        pend_chi = ea - ek; pend_p0 = ep[0]; pend_p1 = ep[1]; pend_p2 = ep[2]; pend_ee = ee; pend_i = i;
      } else a.epot[i] = ee;
      if (TTMN) a.numneighs[i] = nb;
      sums[0] += ee; sums[1] += vir;
    }
    PH(5)
    TILE_EPILOGUE_A(false)
    PH(6)
    if constexpr (DYN) { pos = next_tile >= 0 ? npos : -1; pos_end = npos_end; T.tile = next_tile >= 0 ? next_tile : 0; }
    else T.tile++;
  }
  emb_finish();
  PH_END
  block_sum<3>(sums, scratch);
  if (threadIdx.x == 0) {
    const size_t pb = (size_t) blockIdx.x + (DYN ? (size_t) a.part_off : 0);
    a.partials[pb * IMDGPU_NPART + 0] = sums[0];
    a.partials[pb * IMDGPU_NPART + 1] = sums[1];
    if (FEMB) a.partials[pb * IMDGPU_NPART + 2] = sums[2];
  }
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(kmn, o, 64); kmn = t < kmn ? t : kmn; }
  if ((threadIdx.x & 63) == 0 && kmn != 0x7fffffff) atomicMax(&a.counters[CNT_KNEG], 0x7fffffff - kmn);
}

// 144 VGPRs: three waves per SIMD then leave 80 registers free on EVERY SIMD, which is what one wave of the TTM
// sub-step kernel needs -- the lattice chain that runs beside this pass gets four SIMDs per CU instead of one
// RING = 0 beside the TTM chain (142 VGPRs), 4 otherwise (148): see LIST_WORDS_END
#ifdef P3_VGPRS
#define P3_ATTR __attribute__((amdgpu_num_vgpr(P3_VGPRS)))
#else
#define P3_ATTR
#endif
template <bool NT1, int RING, int CAP>
__global__ __launch_bounds__(TILE_MAXTHREADS) P3_ATTR void k_pass3_tile(TileArgs a)
{
  extern __shared__ double lds[];
  if (a.p3_fallback) {                                   // did k_pass3_tile_coef do this step?  (same test as there)
    const int kneg = a.counters[CNT_KNEG];
    if (kneg != 0 && 0x7fffffff - kneg >= a.klo + 1) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.counters[CNT_P3_STANDIN], 1);    // this pass is the stand-in's (get_info "p3_standin_passes")
  }
  double *tr = lds;                                      // rho_h_tab, one type: + zero rows for r2 >= end (no mask)
  double *sp = tr + a.rho.nrows * a.rho.ncols + 4;      // (x, y) records, then (z, dF) records (LPOS_*) ...
  sp += ((size_t) sp >> 3) & 1;
  double *sw = sp + 3 * (size_t) SLOTCAP(a);             // ... and dF in its own array (fewer bank conflicts than 32 B records)
  double *scratch = sw + SLOTCAP(a);
  int *ts = (int *) (scratch + 64);
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  double sums[1] = {0.0};
  int is_short = 0;
  if (t0 < t1) {
    lds_load_table(tr, a.rho);
    if (threadIdx.x < 4) tr[a.rho.nrows * a.rho.ncols + threadIdx.x] = 0.0;
  }
  // one type: dF is staged as dF * invstep (= 0.5 * 2 invstep), so grad = (w_i + w_j) * (dv + (chi - 1/2) d2v)
  const double wscale = NT1 ? a.rho.invstep[0] : 1.0;
  const int kz = a.rho.nrows * a.rho.ncols;
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, !NT1) }
  int pf_i = -1;                          // f += (this pass's force): read issued when an atom is through, completed one tile later
  double pf_x = 0, pf_y = 0, pf_z = 0, pf_ox = 0, pf_oy = 0, pf_oz = 0;
  auto f_finish = [&]() {
    if (pf_i >= 0) { a.f[pf_i] = pf_ox + pf_x; a.f[a.cap + pf_i] = pf_oy + pf_y; a.f[2 * (size_t) a.cap + pf_i] = pf_oz + pf_z; pf_i = -1; }
  };
  for (T.tile = t0; T.tile < t1; T.tile++) {
    TILE_PROLOGUE_A(true, !NT1, sw, ts, wscale, , , , (T.tile + 1 < t1 ? T.tile + 1 : -1))
    f_finish();                          // the previous tile's atom (see k_pass3_tile_coef)
    const int nt = a.ntypes, inc = NT1 ? 1 : nt * nt;
    const TabK cr = {a.rho.begin[0], a.rho.end[0], a.rho.invstep[0], 2 * a.rho.invstep[0]};
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      OWN_PROLOGUE
      double xi, yi, zi, wi;
      LPOS_RDW(sp, sw, SLOTCAP(a), slot, xi, yi, zi, wi)
      f_finish();
      const int it = NT1 ? 0 : ts[slot];
      double fx = 0, fy = 0, fz = 0, vir = 0;
      int sgn = 0;
      // list words: PFD groups of four slots in flight (only ~3 waves/SIMD are resident, so the HBM stream
      // of the list needs memory-level parallelism from each wave)
      // two list words (8 slots) are consumed per trip while the next two are in flight; lanes past their
      // own count get the dummy word
      LIST_WORDS_BEGIN
        double dx[UNR], dy[UNR], dz[UNR], wj[UNR], r2[UNR];
        PRIO_HI;
#pragma unroll
        for (int e = 0; e < UNR; e++) nbr_rdw<CAP>(sp, sw, SLOTCAP(a), j[e], dx[e], dy[e], dz[e], wj[e]);
        PRIO_LO;
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
          r2[e] = dx[e] * dx[e] + dy[e] * dy[e] + dz[e] * dz[e];
        }
        if (NT1) {
          int kr[UNR];
          double ch[UNR], R0[UNR], R1[UNR], R2[UNR];
#pragma unroll
          for (int e = 0; e < UNR; e++) {
            tab_index_open(cr, r2[e], kr[e], ch[e], sgn);
            kr[e] = (r2[e] < cr.end) ? kr[e] : kz;                        // rho': r2 < end (imd_forces_nbl.c:868)
          }
          SCHED_BARRIER;
          PRIO_HI;
#pragma unroll
          for (int e = 0; e < UNR; e++) { R0[e] = tr[kr[e]]; R1[e] = tr[kr[e] + 1]; R2[e] = tr[kr[e] + 2]; }
          PRIO_LO;
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
            const double dv = R1[e] - R0[e], d2v = (R2[e] - 2 * R1[e]) + R0[e];
            const double grad = (wi + wj[e]) * (dv + (ch[e] - 0.5) * d2v);           // rho'_i = rho'_j (one type)
            fx += dx[e] * grad; fy += dy[e] * grad; fz += dz[e] * grad;
            vir += r2[e] * grad;                                                      // d.f = grad r2
          }
        } else {
#pragma unroll
          for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
            const int jt = nbr_type(ts, j[e]), col1 = jt * nt + it, col2 = it * nt + jt;
            const TabK c1 = {a.rho.begin[col1], a.rho.end[col1], a.rho.invstep[col1], 2 * a.rho.invstep[col1]};
            const TabK c2 = {a.rho.begin[col2], a.rho.end[col2], a.rho.invstep[col2], 2 * a.rho.invstep[col2]};
            int k; double chi;
            tab_index(c1, r2[e], k, chi);
            const double *q = tr + k * inc + col1;
            double p0 = q[0], p1 = q[inc], p2 = q[2 * inc], dv = p1 - p0, d2v = (p2 - 2 * p1) + p0;
            const double ri = c1.istep2 * (dv + (chi - 0.5) * d2v);
            tab_index(c2, r2[e], k, chi);
            q = tr + k * inc + col2;
            p0 = q[0]; p1 = q[inc]; p2 = q[2 * inc]; dv = p1 - p0; d2v = (p2 - 2 * p1) + p0;
            const double rj = c2.istep2 * (dv + (chi - 0.5) * d2v);
            const bool in = (r2[e] < c1.end) || (r2[e] < c2.end);
            is_short |= (in && (r2[e] < c1.begin || r2[e] < c2.begin));
            const double grad = in ? 0.5 * (wi * rj + wj[e] * ri) : 0.0;
            const double gx = dx[e] * grad, gy = dy[e] * grad, gz = dz[e] * grad;
            fx += gx; fy += gy; fz += gz;
            vir -= 0.5 * (dx[e] * gx + dy[e] * gy + dz[e] * gz);
          }
        }
        HALF_END
      LIST_WORDS_END(RING)
      if (NT1) { is_short |= sgn < 0 ? 1 : 0; vir *= -0.5; }
      pf_ox = a.f[i]; pf_oy = a.f[a.cap + i]; pf_oz = a.f[2 * (size_t) a.cap + i]; pf_x = fx; pf_y = fy; pf_z = fz; pf_i = i;
      sums[0] += vir;
    }
    TILE_EPILOGUE_A(!NT1)
  }
  f_finish();
  block_sum<1>(sums, scratch);
  if (threadIdx.x == 0) a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 3] = sums[0];
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}

// The EAM force pass for one atom type on a table grid shared with pair_pot (the merged pass 1 ran): the gradient of
// rho_h_tab needs two numbers per table row, A_k = dv_k - d2v_k / 2 and B_k = d2v_k (grad = A_k + chi B_k is
// DERIV_FUNC2's dv + (chi - 1/2) d2v, potaccess.h:591-621, up to one rounding), so one aligned 16-byte LDS read per
// neighbour replaces three 8-byte reads of consecutive rows, and three VALU instructions go with them.  Twice the bytes
// per row do not fit beside the four position planes for a 4000-row table: the kernel keeps rows [klo, ncoef) -- the
// lowest rows belong to distances no pair of a condensed sample reaches -- and pass 1 reports the lowest row any pair of
// THIS step fell into (CNT_KNEG): when that lies below klo -- or when no pass 1 has reported since the last force pass --
// this kernel returns at once and k_pass3_tile, launched right behind it with p3_fallback set, does the step.  is_short is pass 1's business here (same pairs, same grid).
template <int RING, int CAP>
// four waves per SIMD's worth of registers (<= 128; two spilled with the register ring): three waves of this kernel then leave
// every SIMD the 128 registers a wave of the lattice chain's spill-free build needs (k_ttm_chain_persist<.., 4, ..>).  Measured
// at 10 M atoms + TTM: 132 registers beside the 80-register chain 4.82 ms per step, beside the 96-register chain 4.67,
// 128 beside the 128-register chain 4.62 (profiles/r03_kernel_variants.json)
__global__ __launch_bounds__(TILE_MAXTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pass3_tile_coef(TileArgs a)
{
  extern __shared__ double lds[];
  {                                                      // no report from a pass 1 of this step (0), or a pair below the rows held here
    const int kneg = a.counters[CNT_KNEG];
    if (kneg == 0 || 0x7fffffff - kneg < a.klo + 1) return;
  }
  double *sp = lds;
  double *sw = sp + 3 * (size_t) SLOTCAP(a);
  double *scratch = sw + SLOTCAP(a);
  double2 *tc = (double2 *) (scratch + 64);              // rows klo .. ncoef-1
  tc += ((size_t) tc >> 3) & 1;
  const double2 *tcb = tc - a.klo;                       // indexed by the row number itself
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  double sums[1] = {0.0};
  if (t0 < t1) {
    const double2 *src = a.coef3 + a.klo;
    const int nr = a.ncoef - a.klo;
    for (int k = threadIdx.x; k < nr; k += 4 * blockDim.x) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < nr) v[u] = src[q]; }
#pragma unroll
      for (int u = 0; u < 4; u++) { int q = k + u * blockDim.x; if (q < nr) tc[q] = v[u]; }
    }
  }
  const double wscale = a.rho.invstep[0];                // dF staged as dF * invstep: grad = (w_i + w_j) (A + chi B)
  const int kz = a.ncoef - 1;
  const double end = a.rho.end[0], istep = a.rho.invstep[0], c0 = -a.rho.begin[0] * a.rho.invstep[0];
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, false) }
  // f += (this pass's force) is a read-modify-write: the read is issued when an atom is through, the sum and the store follow in
  // front of the next tile's first barrier, under that tile's first loads (like the embedding look-up of pass 1)
  int pf_i = -1;
  double pf_x = 0, pf_y = 0, pf_z = 0, pf_ox = 0, pf_oy = 0, pf_oz = 0;
  auto f_finish = [&]() {
    if (pf_i >= 0) { a.f[pf_i] = pf_ox + pf_x; a.f[a.cap + pf_i] = pf_oy + pf_y; a.f[2 * (size_t) a.cap + pf_i] = pf_oz + pf_z; pf_i = -1; }
  };
  for (T.tile = t0; T.tile < t1; T.tile++) {
    TILE_PROLOGUE_A(true, false, sw, nullptr, wscale, , , , (T.tile + 1 < t1 ? T.tile + 1 : -1))
    f_finish();                          // the previous tile's atom: its old force has had both barriers' time to arrive
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      OWN_PROLOGUE
      double xi, yi, zi, wi;
      LPOS_RDW(sp, sw, SLOTCAP(a), slot, xi, yi, zi, wi)
      f_finish();
      double fx = 0, fy = 0, fz = 0, vir = 0;
      LIST_WORDS_BEGIN
        double dx[UNR], dy[UNR], dz[UNR], wj[UNR], r2[UNR], ch[UNR];
        int kr[UNR];
        double2 AB[UNR];
        PRIO_HI;
#pragma unroll
        for (int e = 0; e < UNR; e++) nbr_rdw<CAP>(sp, sw, SLOTCAP(a), j[e], dx[e], dy[e], dz[e], wj[e]);
        PRIO_LO;
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
#ifdef P3C_UNFUSED_R2
          r2[e] = dx[e] * dx[e] + dy[e] * dy[e] + dz[e] * dz[e];
#else
          // fused: r2 may differ from pass 1's (the reference's unfused expression) in the last bit.  Nothing here is an
          // integer or a list decision -- the cut-off test gates a term that is continuous in r2 -- and a row index that
          // lands one below pass 1's is covered by the guard's margin of one row
          r2[e] = fma(dz[e], dz[e], fma(dy[e], dy[e], dx[e] * dx[e]));
#endif
          // tab_index_fma without the sign and without the clamp at 0: this kernel runs only when pass 1 has seen no pair below row
          // klo + 1 >= 1 of this table (the guard above), so q >= 1 up to the roundings the guard's margin of one row covers
          const double q = fma(r2[e], istep, c0);
          kr[e] = (int) q;
          ch[e] = __builtin_amdgcn_fract(q);
          kr[e] = (r2[e] < end) ? kr[e] : kz;                              // rho': r2 < end (imd_forces_nbl.c:868)
        }
        SCHED_BARRIER;
        PRIO_HI;
#pragma unroll
        for (int e = 0; e < UNR; e++) AB[e] = tcb[kr[e]];
        PRIO_LO;
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
          const double grad = (wi + wj[e]) * fma(ch[e], AB[e].y, AB[e].x);
          fx += dx[e] * grad; fy += dy[e] * grad; fz += dz[e] * grad;
          vir += r2[e] * grad;
        }
        HALF_END
      LIST_WORDS_END(RING)
      pf_ox = a.f[i]; pf_oy = a.f[a.cap + i]; pf_oz = a.f[2 * (size_t) a.cap + i]; pf_x = fx; pf_y = fy; pf_z = fz; pf_i = i;
      sums[0] += -0.5 * vir;
    }
    TILE_EPILOGUE_A(false)
  }
  f_finish();
  block_sum<1>(sums, scratch);
  if (threadIdx.x == 0) a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 3] = sums[0];
}

// ---- several atom types: the record kernels --------------------------------------------------------------------------------------
// Positions in the LDS like everywhere; the neighbour's type rides in the list entry (NBR_TYPE_BITS), the own atom's comes from the
// atom array; the per-column parameters (begin, step, end: types.h:537-549, one set per column) sit in a small LDS array.  Column as
// the reference has it: col = it * ntypes + jt for the pair term and for the density atom i receives (imd_forces_nbl.c:457-470,550);
// cut-offs <= / < per column (:489,:550); the embedding pass of the own atom (:748-783, column = its type) is folded in as in
// k_pass1_tile_merged.  Where a table's rows come from is decided per table and per launch (make_nblist: rec_lds):
//   *LDS = true   the table sits in the LDS in COMPACT form -- every distinct column (columns with equal parameters and equal rows
//                 are kept once: phi_01 = phi_10 of a physical pair potential) with its own len + 2 rows, one behind the other, three
//                 zero rows at the end; a look-up is three 8-byte reads, dv = p1 - p0 and d2v = p2 - 2 p1 + p0 follow (potaccess.h:338).
//   *LDS = false  the table does not fit beside the tile (two types at production size: 2 x 4 columns x 4002 rows = 256 KB): a
//                 look-up is ONE aligned record (p0, dv, d2v) -- the same differences formed on the host -- read in place through
//                 L2: two loads of one sector instead of three rows 32 bytes apart.  The texture path serves a wave's 64 different
//                 lines in ~55 cycles per instruction (the LDS in ~7): this form is bound by it.
// Look-ups beyond their cut-off read zero rows (LDS) or load nothing (L2).
// A record look-up through L2 is a raw BUFFER load: the look-ups beyond their cut-off (list skin, padding: a quarter of the entries)
// carry the record index -1, i.e. an offset beyond the buffer's size -- the hardware returns zeros for it and sends nothing to the
// texture path (whose cost goes with the number of lanes that ask), without a branch around the load.
typedef unsigned rec_u4 __attribute__((ext_vector_type(4)));
typedef unsigned rec_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rec_rsrc(const void *p, size_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, (int) bytes, 0x00020000);        // raw buffer (stride 0), 32-bit data format
}
__device__ __forceinline__ double2 rec_ld16(__amdgpu_buffer_rsrc_t r, int byteoff)
{
  const rec_u4 q = __builtin_amdgcn_raw_buffer_load_b128(r, byteoff, 0, 0);
  return make_double2(__hiloint2double((int) q.y, (int) q.x), __hiloint2double((int) q.w, (int) q.z));
}
__device__ __forceinline__ double rec_ld8(__amdgpu_buffer_rsrc_t r, int byteoff)
{
  const rec_u2 q = __builtin_amdgcn_raw_buffer_load_b64(r, byteoff, 0, 0);
  return __hiloint2double((int) q.y, (int) q.x);
}
struct ColPar { double c0, istep, end; int boff, pad; };    // c0 = -begin * invstep; boff: first row / record of the column
template <bool TTMN, int CAP, bool PLDS, bool RLDS>
__global__ __launch_bounds__(TILE_MAXTHREADS) void k_pass1_tile_rec(TileArgs a)
{
  extern __shared__ double lds[];
  if (a.guard && *a.guard) return;
  double *sp = lds;
  double *scratch = sp + 3 * (size_t) SLOTCAP(a);
  ColPar *parP = (ColPar *) (scratch + 64), *parR = parP + IMDGPU_MAXCOL;      // slot_cap is a multiple of 64: 16-byte aligned
  double *tp = (double *) (parR + IMDGPU_MAXCOL), *tr = tp + (PLDS ? a.ncompP : 0);
  (void) tp; (void) tr;
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  const int nt = a.ntypes, ncol = nt * nt;
  if ((int) threadIdx.x < ncol) {
    const int c = threadIdx.x;
    parP[c] = {-a.pair.begin[c] * a.pair.invstep[c], a.pair.invstep[c], a.pair.end[c], PLDS ? a.loffP[c] : c * a.nrecP, 0};
    parR[c] = {-a.rho.begin[c] * a.rho.invstep[c], a.rho.invstep[c], a.rho.end[c], RLDS ? a.loffR[c] : c * a.nrecR, 0};
  }
  if (t0 < t1) {
    if (PLDS) for (int k = threadIdx.x; k < a.ncompP; k += blockDim.x) tp[k] = a.compP[k];
    if (RLDS) for (int k = threadIdx.x; k < a.ncompR; k += blockDim.x) tr[k] = a.compR[k];
  }
  const double pend0 = a.pair.end[0];
  const int kzP = a.ncompP - 3, kzR = a.ncompR - 3;      // the zero rows of the compact tables
  (void) kzP; (void) kzR;
  const __amdgpu_buffer_rsrc_t rsP = rec_rsrc(a.recP, (size_t) ncol * a.nrecP * 32), rsR = rec_rsrc(a.recR, (size_t) ncol * a.nrecR * 32);
  (void) rsP; (void) rsR;
  double sums[3] = {0.0, 0.0, 0.0};        // pair energy, virial, embedding energy
  int is_short = 0;
  int pend_i = -1, pend_t = 0;
  double pend_chi = 0, pend_p0 = 0, pend_p1 = 0, pend_p2 = 0, pend_ee = 0;
  auto emb_finish = [&]() {
    if (pend_i >= 0) {
      Tab3 q;
      q.p0 = pend_p0; q.dv = pend_p1 - pend_p0; q.d2v = pend_p2 - 2 * pend_p1 + pend_p0; q.chi = pend_chi; q.istep = a.emb.invstep[pend_t];
      const double pe = tab_val(q);
      a.posw[pend_i].w = tab_grad(q);
      a.epot[pend_i] = pend_ee + pe;
      sums[2] += pe;
      pend_i = -1;
    }
  };
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, false) }
  for (T.tile = t0; T.tile < t1; T.tile++) {
    TILE_PROLOGUE_A(false, false, nullptr, nullptr, 1.0, , , , (T.tile + 1 < t1 ? T.tile + 1 : -1))
    emb_finish();
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      const int it = a.type[i], itn = it * nt;
      OWN_PROLOGUE
      double xi, yi, zi;
      LPOS_RD(sp, sp, SLOTCAP(a), slot, xi, yi, zi)
      emb_finish();
      double fx = 0, fy = 0, fz = 0, ee = 0, rr = 0, vir = 0, ee0 = 0, rr0 = 0;
      int nb = 0, sgn = 0, wi = 0;
      const unsigned segw = a.ownseg[(size_t) T.tile * a.tile_cap + o];
      const int sg0 = segw & 0xff, sg1 = (segw >> 8) & 0xff, sg2 = (segw >> 16) & 0xff;
      LIST_WORDS_BEGIN
        double dx[UNR], dy[UNR], dz[UNR], r2[UNR], chp[UNR], chr[UNR];
        int kp[UNR], kr[UNR];
        double P0[UNR], Pv[UNR], Pw[UNR], R0[UNR], Rv[UNR], Rw[UNR];              // (p0, dv, d2v) of the two look-ups
        const int col = itn + (wi >= sg0) + (wi >= sg1) + (wi >= sg2);            // this word's neighbours are all of one type
        wi++;
        const ColPar cp = parP[col], cr = parR[col];
#pragma unroll
        for (int e = 0; e < UNR; e++) nbr_rd<CAP>(sp, SLOTCAP(a), j[e], dx[e], dy[e], dz[e]);
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
          r2[e] = r2_of<TTMN>(dx[e], dy[e], dz[e]);
          double ap = fma(r2[e], cp.istep, cp.c0), ar = fma(r2[e], cr.istep, cr.c0);
          sgn |= __double2hiint(ap) | __double2hiint(ar);                 // r2 below a table's begin (is_short; see tab_index_fma)
          ap = fmax(ap, 0.0); ar = fmax(ar, 0.0);
          const bool inp = r2[e] <= cp.end;                               // pair: <= end (imd_forces_nbl.c:489)
          if (TTMN) nb += (r2[e] <= pend0 && col == 0) ? 1 : 0;           // both of type 0 (:473-476)
          kp[e] = inp ? cp.boff + (int) ap : (PLDS ? kzP : -1); chp[e] = __builtin_amdgcn_fract(ap);
          kr[e] = r2[e] < cr.end ? cr.boff + (int) ar : (RLDS ? kzR : -1); chr[e] = __builtin_amdgcn_fract(ar);      // density: < end (:550)
        }
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
          if constexpr (PLDS) { lds_cvdouble *q = (lds_cvdouble *) (tp + kp[e]); P0[e] = q[0]; Pv[e] = q[1]; Pw[e] = q[2]; }
          else {
            const double2 q = rec_ld16(rsP, kp[e] * 32); P0[e] = q.x; Pv[e] = q.y; Pw[e] = rec_ld8(rsP, kp[e] * 32 + 16);
          }
          if constexpr (RLDS) { lds_cvdouble *q = (lds_cvdouble *) (tr + kr[e]); R0[e] = q[0]; Rv[e] = q[1]; Rw[e] = q[2]; }
          else {
            const double2 q = rec_ld16(rsR, kr[e] * 32); R0[e] = q.x; Rv[e] = q.y; Rw[e] = rec_ld8(rsR, kr[e] * 32 + 16);
          }
        }
        SCHED_BARRIER;
#pragma unroll
        for (int e = 0; e < UNR; e++) {
#pragma clang fp contract(fast)
          if constexpr (PLDS) { const double p1 = Pv[e], p2 = Pw[e]; Pv[e] = p1 - P0[e]; Pw[e] = (p2 - 2 * p1) + P0[e]; }
          if constexpr (RLDS) { const double p1 = Rv[e], p2 = Rw[e]; Rv[e] = p1 - R0[e]; Rw[e] = (p2 - 2 * p1) + R0[e]; }
          const double gr = fma(chp[e] - 0.5, Pw[e], Pv[e]) * cp.istep;     // grad / 2: doubled once per atom
          ee = fma(chp[e], fma(chp[e] - 1.0, 0.5 * Pw[e], Pv[e]), ee); ee0 += P0[e];      // halved once per atom
          rr = fma(chr[e], fma(chr[e] - 1.0, 0.5 * Rw[e], Rv[e]), rr); rr0 += R0[e];
          fx += dx[e] * gr; fy += dy[e] * gr; fz += dz[e] * gr;
          vir += r2[e] * gr;
        }
        HALF_END
      LIST_WORDS_END(4)
      ee = 0.5 * (ee + ee0); rr += rr0;
      fx *= 2.0; fy *= 2.0; fz *= 2.0; vir = -vir;                        // vir -= r2 grad / 2, grad = 2 gr
      is_short |= sgn < 0 ? 1 : 0;
      a.f[i] = fx; a.f[a.cap + i] = fy; a.f[2 * (size_t) a.cap + i] = fz;
      a.rhoh[i] = rr;
      {                                                     // PAIR_INT on embed_pot, column = the atom's type (:757-760); EAM_DF into pos.w
        double ea = (rr < a.emb.end[it] ? rr : a.emb.end[it]) - a.emb.begin[it];
        ea = (ea < 0 ? 0 : ea) * a.emb.invstep[it];
        const int ek = (int) ea;
        const double *ep = a.emb.tab + (size_t) ek * nt + it;
        pend_chi = ea - ek; pend_p0 = ep[0]; pend_p1 = ep[nt]; pend_p2 = ep[2 * nt]; pend_ee = ee; pend_i = i; pend_t = it;
      }
      if (TTMN) a.numneighs[i] = nb;
      sums[0] += ee; sums[1] += vir;
    }
    TILE_EPILOGUE_A(false)
  }
  emb_finish();
  block_sum<3>(sums, scratch);
  if (threadIdx.x == 0) {
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 0] = sums[0];
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 1] = sums[1];
    a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 2] = sums[2];
  }
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}

// The EAM force pass of the same family: grad = (dF_i rho'_j + dF_j rho'_i) / 2 with rho'_i from column jt * ntypes + it and rho'_j
// from column it * ntypes + jt, both evaluated -- at min(r2, end) like DERIV_FUNC2 does (potaccess.h:601) -- whenever r2 lies inside
// EITHER column's range (imd_forces_nbl.c:843-877).  RLDS = false: a look-up is one 16-byte record invstep * (dv - d2v / 2, d2v);
// RLDS = true: three rows of the compact rho_h_tab in the LDS.
// RING = 0 (beside the lattice chain): 144 registers, so that three waves per SIMD leave the 80 a wave of the chain needs (see k_pass3_tile)
template <int CAP, bool RLDS, int RING>
__device__ __forceinline__ void pass3_tile_rec_body(const TileArgs &a)
{
  extern __shared__ double lds[];
  double *sp = lds;
  double *sw = sp + 3 * (size_t) SLOTCAP(a);
  double *scratch = sw + SLOTCAP(a);
  ColPar *par = (ColPar *) (scratch + 64);
  double *tr = (double *) (par + IMDGPU_MAXCOL);
  (void) tr;
  TileCtx T;
  int t0, t1;
  block_tile_range(a.ntiles, a.wprefix, a.balance, t0, t1);
  const int nt = a.ntypes, ncol = nt * nt;
  if ((int) threadIdx.x < ncol) {
    const int c = threadIdx.x;
    par[c] = {-a.rho.begin[c] * a.rho.invstep[c], a.rho.invstep[c], a.rho.end[c], RLDS ? a.loffR[c] : c * a.nrecR, 0};
  }
  if (RLDS && t0 < t1) for (int k = threadIdx.x; k < a.ncompR; k += blockDim.x) tr[k] = a.compR[k];
  const int kzR = a.ncompR - 3;                          // the zero rows of the compact table
  (void) kzR;
  const __amdgpu_buffer_rsrc_t rs3 = rec_rsrc(a.rec3, (size_t) ncol * a.nrecR * 16);
  (void) rs3;
  double sums[1] = {0.0};
  int is_short = 0;
  int4 info = make_int4(0, 0, 0, 0);
  AHEAD_DECL(scratch)
  if (t0 < t1) { info = a.tinfo[t0]; AHEAD_GRAB(t0, info.x, false) }
  int pf_i = -1;
  double pf_x = 0, pf_y = 0, pf_z = 0, pf_ox = 0, pf_oy = 0, pf_oz = 0;
  auto f_finish = [&]() {
    if (pf_i >= 0) { a.f[pf_i] = pf_ox + pf_x; a.f[a.cap + pf_i] = pf_oy + pf_y; a.f[2 * (size_t) a.cap + pf_i] = pf_oz + pf_z; pf_i = -1; }
  };
  for (T.tile = t0; T.tile < t1; T.tile++) {
    TILE_PROLOGUE_A(true, false, sw, nullptr, 1.0, , , , (T.tile + 1 < t1 ? T.tile + 1 : -1))
    f_finish();
    for (int o = threadIdx.x; o < T.nown; o += blockDim.x) {
      const int i = T.own_start + o;
      const int it = a.type[i], itn = it * nt;
      OWN_PROLOGUE
      double xi, yi, zi, wi;
      LPOS_RDW(sp, sw, SLOTCAP(a), slot, xi, yi, zi, wi)
      f_finish();
      double fx = 0, fy = 0, fz = 0, vir = 0;
      int sgn = 0, wd = 0;
      constexpr int HB = RING == 0 ? 2 : UNR;
      const unsigned segw = a.ownseg[(size_t) T.tile * a.tile_cap + o];
      const int sg0 = segw & 0xff, sg1 = (segw >> 8) & 0xff, sg2 = (segw >> 16) & 0xff;
      LIST_WORDS_BEGIN
        const int jt = (wd >= sg0) + (wd >= sg1) + (wd >= sg2);                   // this word's neighbours are all of one type
        wd++;
        const ColPar p1 = par[jt * nt + it], p2 = par[itn + jt];
        const bool other = jt != it;
        // HB neighbours at a time: all four where registers are plenty, two and two in the lean form
#pragma unroll
        for (int h = 0; h < UNR; h += HB) {
          double dx[HB], dy[HB], dz[HB], wj[HB], r2[HB], c1[HB], c2[HB], g1[HB], g2[HB];
          int k1[HB], k2[HB];
#pragma unroll
          for (int e = 0; e < HB; e++) nbr_rdw<CAP>(sp, sw, SLOTCAP(a), j[h + e], dx[e], dy[e], dz[e], wj[e]);
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < HB; e++) {
            dx[e] -= xi; dy[e] -= yi; dz[e] -= zi;
            r2[e] = fma(dz[e], dz[e], fma(dy[e], dy[e], dx[e] * dx[e]));   // fused (nothing here is a decision that is not continuous in r2)
            const bool in = (r2[e] < p1.end) || (r2[e] < p2.end);
            double a1 = fma(fmin(r2[e], p1.end), p1.istep, p1.c0), a2 = fma(fmin(r2[e], p2.end), p2.istep, p2.c0);
            sgn |= in ? (__double2hiint(a1) | __double2hiint(a2)) : 0;
            a1 = fmax(a1, 0.0); a2 = fmax(a2, 0.0);
            // beyond both cut-offs: zero rows / nothing is loaded; atoms of one type: one look-up serves both terms (col1 == col2, :858-860)
            k1[e] = in ? p1.boff + (int) a1 : (RLDS ? kzR : -1); c1[e] = __builtin_amdgcn_fract(a1);
            k2[e] = (in && (RLDS || other)) ? p2.boff + (int) a2 : (RLDS ? kzR : -1); c2[e] = __builtin_amdgcn_fract(a2);
          }
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < HB; e++) {
#pragma clang fp contract(fast)
            if constexpr (RLDS) {                     // rho' / 2 = invstep (dv + (chi - 1/2) d2v) from three rows (potaccess.h:601-617)
              lds_cvdouble *q1 = (lds_cvdouble *) (tr + k1[e]), *q2 = (lds_cvdouble *) (tr + k2[e]);
              const double u0 = q1[0], u1 = q1[1], u2 = q1[2], v0 = q2[0], v1 = q2[1], v2 = q2[2];
              const double du = u1 - u0, d2u = (u2 - 2 * u1) + u0, dv = v1 - v0, d2v = (v2 - 2 * v1) + v0;
              g1[e] = p1.istep * fma(c1[e] - 0.5, d2u, du); g2[e] = p2.istep * fma(c2[e] - 0.5, d2v, dv);
            } else {                                  // the records carry invstep (dv - d2v / 2, d2v)
              const double2 A1 = rec_ld16(rs3, k1[e] * 16), A2 = rec_ld16(rs3, k2[e] * 16);
              g1[e] = fma(c1[e], A1.y, A1.x); g2[e] = other ? fma(c2[e], A2.y, A2.x) : g1[e];
            }
          }
          SCHED_BARRIER;
#pragma unroll
          for (int e = 0; e < HB; e++) {
#pragma clang fp contract(fast)
            const double grad = wi * g2[e] + wj[e] * g1[e];                 // the 2 of rho' cancels the 1/2 of the force term
            fx += dx[e] * grad; fy += dy[e] * grad; fz += dz[e] * grad;
            vir += r2[e] * grad;                                            // d . f = grad r2
          }
          if (HB != UNR) SCHED_BARRIER;
        }
        HALF_END
      LIST_WORDS_END(RING)
      is_short |= sgn < 0 ? 1 : 0;
      pf_ox = a.f[i]; pf_oy = a.f[a.cap + i]; pf_oz = a.f[2 * (size_t) a.cap + i]; pf_x = fx; pf_y = fy; pf_z = fz; pf_i = i;
      sums[0] += -0.5 * vir;
    }
    TILE_EPILOGUE_A(false)
  }
  f_finish();
  block_sum<1>(sums, scratch);
  if (threadIdx.x == 0) a.partials[(size_t) blockIdx.x * IMDGPU_NPART + 3] = sums[0];
  if (is_short) a.counters[CNT_IS_SHORT] = 1;
}

template <int CAP, bool RLDS>
__global__ __launch_bounds__(TILE_MAXTHREADS) void k_pass3_tile_rec(TileArgs a) { pass3_tile_rec_body<CAP, RLDS, 4>(a); }
template <int CAP, bool RLDS>
__global__ __launch_bounds__(TILE_MAXTHREADS) void k_pass3_tile_rec_lean(TileArgs a) { pass3_tile_rec_body<CAP, RLDS, 0>(a); }

// Deterministic final reduction of the per-block partial sums
// (tot_pot_energy, virial: imd_forces_nbl.c:319-328,:496,:521,:760,:943).
// nb1: workgroups of pass 1 (slots 0, 1), nb2: of the embedding pass (slot 2), nb3: of the EAM force pass (slot 3)
__global__ void k_reduce_force_sums(int nb1, int nb2, int nb3, const double *partials, double *sums, int *counters)
{
  if (threadIdx.x == 0) counters[CNT_KNEG] = 0;          // both guarded EAM force kernels of this step are through
  __shared__ double scratch[4 * 16];
  double v[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb1; b += blockDim.x) {
    v[0] += partials[(size_t) b * IMDGPU_NPART + 0];
    v[1] += partials[(size_t) b * IMDGPU_NPART + 1];
  }
  for (int b = threadIdx.x; b < nb3; b += blockDim.x) v[3] += partials[(size_t) b * IMDGPU_NPART + 3];
  for (int b = threadIdx.x; b < nb2; b += blockDim.x) v[2] += partials[(size_t) b * IMDGPU_NPART + 2];
  block_sum<4>(v, scratch);
  if (threadIdx.x == 0) {
    sums[SUM_EPOT] = v[0] + v[2];
    sums[SUM_VIRIAL] = v[1] + v[3];
    for (int q = 2; q < 8; q++) sums[q] = 0.0;
  }
}

// ---------------------------------------------------------------------------
// integrator + list check  [move_atoms_nve, check_nblist]
// ---------------------------------------------------------------------------
template <bool CHECK>
__global__ void k_move_nve(int n, int cap, double4 *pos, const int *type, double *p, double *f, const double *mass,
                           const double *restr, int ntypes, double dt, const double *ref, double *partials)
{
  __shared__ double scratch[2 * 16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s[2] = {0.0, 0.0};
  if (i < n) {
    const size_t c1 = cap, c2 = 2 * (size_t) cap;
    int t = ntypes == 1 ? 0 : type[i];
    double fx = f[i], fy = f[c1 + i], fz = f[c2 + i];
    double px = p[i], py = p[c1 + i], pz = p[c2 + i], m = mass[i];
    double e1 = px * px + py * py + pz * pz;
    if (restr) {                                 // KRAFT *= restrictions (imd_integrate.c); nullptr = every factor is 1: nothing to write back
      fx *= restr[3 * t]; fy *= restr[3 * t + 1]; fz *= restr[3 * t + 2];
      f[i] = fx; f[c1 + i] = fy; f[c2 + i] = fz;
    }
    px += dt * fx; py += dt * fy; pz += dt * fz;
    double e2 = px * px + py * py + pz * pz;
    s[0] = (e1 + e2) / (4 * m);
    double tmp = dt / m;
    double4 r = pos[i];
    r.x += tmp * px; r.y += tmp * py; r.z += tmp * pz;
    pos[i] = r;
    p[i] = px; p[c1 + i] = py; p[c2 + i] = pz;
    if (CHECK) {
      double dx = r.x - ref[i], dy = r.y - ref[c1 + i], dz = r.z - ref[c2 + i];
      s[1] = dx * dx + dy * dy + dz * dz;
    }
  }
  s[0] = wave_sum(s[0]);
  if (CHECK) s[1] = wave_max(s[1]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { scratch[w] = s[0]; scratch[16 + w] = s[1]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double e = 0, mx = 0;
    for (int q = 0; q < nw; q++) { e += scratch[q]; mx = fmax(mx, scratch[16 + q]); }
    partials[(size_t) blockIdx.x * 2] = e;
    partials[(size_t) blockIdx.x * 2 + 1] = mx;
  }
}

__global__ void k_check_only(int n, int cap, const double4 *pos, const double *ref, double *partials)
{
  __shared__ double scratch[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    double4 r = pos[i];
    double dx = r.x - ref[i], dy = r.y - ref[cap + i], dz = r.z - ref[2 * (size_t) cap + i];
    v = dx * dx + dy * dy + dz * dz;
  }
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = 0;
    for (int q = 0; q < nw; q++) mx = fmax(mx, scratch[q]);
    partials[(size_t) blockIdx.x * 2 + 1] = mx;
    partials[(size_t) blockIdx.x * 2] = 0.0;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int nblk(long long n, int b) { return n <= 0 ? 1 : (int) ((n + b - 1) / b); }   // never an empty grid: a rank may own no atoms (vacuum)

static int setup_tiles(imdgpu_ctx *ctx);
static int build_ref_perm(imdgpu_ctx *ctx, std::vector<int> *perm, std::vector<int> *refstart);

static void invert_box(const double b[9], double t[9])
{
  double cx[3] = {b[4] * b[8] - b[5] * b[7], b[5] * b[6] - b[3] * b[8], b[3] * b[7] - b[4] * b[6]};
  double cy[3] = {b[7] * b[2] - b[8] * b[1], b[8] * b[0] - b[6] * b[2], b[6] * b[1] - b[7] * b[0]};
  double cz[3] = {b[1] * b[5] - b[2] * b[4], b[2] * b[3] - b[0] * b[5], b[0] * b[4] - b[1] * b[3]};
  double det = b[0] * cx[0] + b[1] * cx[1] + b[2] * cx[2];
  for (int i = 0; i < 3; i++) { t[i] = cx[i] / det; t[3 + i] = cy[i] / det; t[6 + i] = cz[i] / det; }
}

extern "C" int imdgpu_create(imdgpu_ctx **out, int device)
{
  imdgpu_ctx *ctx = new imdgpu_ctx();
  *out = ctx;
  ctx->device = device;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) ctx->num_cus = pr.multiProcessorCount; }
  ctx->tl_on = cfg_env("IMDGPU_TIMELINE") != nullptr;
  {
    // the TTM lattice chain (100 tiny dependent launches) runs beside the force passes: give it the queue priority
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, greatest));
  }
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_ufd, hipEventDisableTiming));
  ctx->own_stream = true;
  HIPCHK(hipMalloc((void **) &ctx->d_counters, CNT_N * sizeof(int)));
  HIPCHK(hipMemset(ctx->d_counters, 0, CNT_N * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &ctx->h_counters, CNT_N * sizeof(int)));
  HIPCHK(hipMalloc((void **) &ctx->d_sums, SUM_N * sizeof(double)));
  HIPCHK(hipMemset(ctx->d_sums, 0, SUM_N * sizeof(double)));
  HIPCHK(hipHostMalloc((void **) &ctx->h_sums, SUM_N * sizeof(double)));
  return 0;
}

extern "C" void imdgpu_destroy(imdgpu_ctx *c)
{
#ifdef PHASE_TIMING
  {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) == hipSuccess && h[7]) {
      fprintf(stderr, "[imdgpu phases] thread %d of k_pass1_tile_merged, us per workgroup-launch:", PH_THREAD);
      for (int q = 0; q < 7; q++) fprintf(stderr, " %d=%.1f", q, h[q] * 0.01 / (double) h[7]);
      fprintf(stderr, "  (%llu workgroup-launches)\n", h[7]);
    }
  }
#endif
  if (!c) return;
  (void) hipSetDevice(c->device);
  (void) hipDeviceSynchronize();
  imdgpu_prof_resolve(c);
  for (auto e : c->ev_pool) (void) hipEventDestroy(e);
  imdgpu_ttm_free(c);
  imdgpu_ttm1d_free(c);
  imdgpu_dist_free(c);
  if (c->d_refperm) (void) hipFree(c->d_refperm);
  if (c->d_stage) (void) hipFree(c->d_stage);
  if (c->d_wprefix) (void) hipFree(c->d_wprefix);
  if (c->d_tile_map) (void) hipFree(c->d_tile_map);
  if (c->d_dyn) (void) hipFree(c->d_dyn);
  void *ptrs[] = {c->pos, c->type, c->id, c->acell, c->p, c->f, c->ref, c->mass, c->epot, c->rhoh, c->numneighs,
                  c->pos2, c->type2, c->id2, c->p2, c->mass2, c->nn2, c->fdc, c->fdc2, c->keys, c->keys2, c->perm, c->perm2,
                  c->cub_tmp, c->cell_start, c->cell_cnt, c->icnt, c->istart, c->gcell_lin, c->gcell_src,
                  c->gcell_shift, c->gcnt, c->gstart, c->ghost_src, c->ghost_code, c->d_counters, c->nbr,
                  c->nbr_cnt, c->nbr16, c->tsrc, c->ownw, c->tinfo, c->d_merged, c->d_coef3, c->d_recP, c->d_recR, c->d_rec3, c->partials, c->d_sums, c->pair.d_tab, c->rho.d_tab, c->emb.d_tab, c->d_restr};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  if (c->h_counters) (void) hipHostFree(c->h_counters);
  if (c->h_sums) (void) hipHostFree(c->h_sums);
  if (c->own_stream && c->stream) (void) hipStreamDestroy(c->stream);
  if (c->stream2) (void) hipStreamDestroy(c->stream2);
  if (c->ev_fork) (void) hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void) hipEventDestroy(c->ev_join);
  if (c->ev_ufd) (void) hipEventDestroy(c->ev_ufd);
  delete c;
}

extern "C" const char *imdgpu_last_error(const imdgpu_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int imdgpu_set_option(imdgpu_ctx *ctx, const char *name, int value)
{
  imdgpu_dist_cancel_ahead(ctx);
  if (!strcmp(name, "mode")) {
    if (value != 0 && value != 1) return imdgpu_fail(ctx, "mode must be 0 (global gather) or 1 (LDS tiles)");
    ctx->mode = value; ctx->have_valid_nbl = 0;
    return 0;
  }
  if (!strcmp(name, "tile_x") || !strcmp(name, "tile_y") || !strcmp(name, "tile_z")) {
    if (ctx->nlocal > 0) return imdgpu_fail(ctx, "tile shape must be set before atoms are uploaded");
    ctx->tile[name[5] - 'x'] = value; ctx->autotiled = 1;
    return ctx->have_geometry ? setup_tiles(ctx) : 0;
  }
  // (cells_epoch: the atoms' FD-cell memory may be stale when the bookkeeping comes back on, see imdgpu_ttm_update_fd)
  if (!strcmp(name, "ttm_flux")) { ctx->ttm_flux = value ? 1 : 0; if (!value) ctx->ttm_adv = 0; ctx->cells_epoch++; return 0; }
  if (!strcmp(name, "ttm_adv")) { ctx->ttm_adv = value ? 1 : 0; if (value) ctx->ttm_flux = 1; ctx->cells_epoch++; return 0; }
  if (!strcmp(name, "p3_coef")) { ctx->p3_coef = value ? 1 : 0; return 0; }
  // tests, before upload_tables: pair_pot and rho_h_tab are kept apart even on a shared grid (the distinct-grids form of pass 1)
  if (!strcmp(name, "no_merge")) { ctx->no_merge = value ? 1 : 0; return 0; }
  if (!strcmp(name, "df_halo_beside_head")) { ctx->df_halo_beside_head = value < 0 ? -1 : (value ? 1 : 0); return 0; }   // see imdgpu_dist_md_step
  if (!strcmp(name, "act_ranksort_max")) { ctx->act_ranksort_max = value; return 0; }       // test hook (do_cell_activation)
  if (!strcmp(name, "ttm_remote")) { ctx->ttm_remote_opt = value ? 1 : 0; return 0; }        // before the first multi-rank rebuild
  if (!strcmp(name, "ttm_remote_fail")) { ctx->ttm_remote_fail = value; return 0; }          // test hook
  // several atom types: 0 = never the record kernels (tables beyond the LDS then select the global-gather family), 1 = when the tables
  // do not fit the LDS beside a tile (default), 2 = whenever there are several types
  if (!strcmp(name, "tile_rec")) { ctx->tile_rec_opt = value > 0; ctx->have_valid_nbl = 0; ctx->autotiled = 0; return 0; }
  if (!strcmp(name, "tile_rec_lds")) { ctx->rec_lds_opt = value < 0 ? -1 : (value & 7); ctx->have_valid_nbl = 0; ctx->autotiled = 0; return 0; }
  if (!strcmp(name, "p3_klo")) { ctx->p3_klo = value > 0 ? value : 0; return 0; }           // tests: which rows k_pass3_tile_coef holds
  if (!strcmp(name, "eos_bilinear")) { ctx->eos_bilinear = value ? 1 : 0; return 0; }    // before ttm_init / ttm1d_init
  if (!strcmp(name, "ttm_persist")) { imdgpu_ttm_set_persist(ctx, value < 0 ? -1 : (value ? 1 : 0)); return 0; }
  // several ranks: pass 1 of the next step starts on interior tiles while its position halo travels (imdgpu_dist.hip); the value is
  // the number of CUs the first launch leaves to the transport's kernels
  if (!strcmp(name, "halo_overlap")) { ctx->halo_overlap = value < 0 ? 0 : (value > 64 ? 64 : value); ctx->dyn_ntiles = -1; imdgpu_dist_cancel_ahead(ctx); return 0; }
  if (!strcmp(name, "ttm_persist_miss")) { ctx->ttm_persist_miss = value; return 0; }      // tests: the MD step whose persistent launch cannot complete its census
  if (!strcmp(name, "ttm_decomposed")) {         // before ttm_init
    if (ctx->ttm) return imdgpu_fail(ctx, "ttm_decomposed must be set before ttm_init");
    ctx->ttm_decomposed = value ? 1 : 0;
    return 0;
  }
  if (!strcmp(name, "external_halo")) {          // before set_geometry: the host layer owns the ghost cells even on one rank
    if (ctx->have_geometry) return imdgpu_fail(ctx, "external_halo must be set before set_geometry");
    ctx->force_external_halo = value ? 1 : 0;
    return 0;
  }
  return imdgpu_fail(ctx, "unknown option %s", name);
}

static bool fuse_emb(const imdgpu_ctx *ctx);
static bool p1_onetype(const imdgpu_ctx *ctx);
static size_t p3_coef_lds(const imdgpu_ctx *ctx, int &klo);
// which experiment macros this library was built with (0 = the product build): bit 0 IMDGPU_EXPERIMENTS (environment switches
// live), 1 PHASE_TIMING, 2 WAVE_FAIR, 3 POSXY, 4 IMDGPU_LDSPRIO, 5 non-default TILE_MAXTHREADS / UNR / SA, 6 IMDGPU_NOSCHED,
// 7 P3_VGPRS / P3C_UNFUSED_R2, 8 PERSIST_PRIO (imdgpu_ttm.hip)
int imdgpu_ttm_experiment_mask();
static long long imdgpu_experiment_mask()
{
  long long m = 0;
#ifdef IMDGPU_EXPERIMENTS
  m |= 1;
#endif
#ifdef PHASE_TIMING
  m |= 2;
#endif
#ifdef WAVE_FAIR
  m |= 4;
#endif
#ifdef POSXY
  m |= 8;
#endif
#ifdef IMDGPU_LDSPRIO
  m |= 16;
#endif
  if (TILE_MAXTHREADS != 768 || UNR != 4 || SA != 8) m |= 32;
#ifdef IMDGPU_NOSCHED
  m |= 64;
#endif
#if defined(P3_VGPRS) || defined(P3C_UNFUSED_R2)
  m |= 128;
#endif
  return m | imdgpu_ttm_experiment_mask();
}
extern "C" int imdgpu_get_info(imdgpu_ctx *ctx, const char *name, long long *value)
{
  int klo_now = 0;
  const bool coef_on = p3_coef_lds(ctx, klo_now) != 0;
  struct { const char *n; long long v; } tab[] = {
    // rows of the gradient-coefficient table the EAM force pass holds in LDS and the first of them (0 / -1: three-row form)
    {"p3_coef_rows", coef_on ? ctx->ncoef - klo_now : 0}, {"p3_coef_klo", coef_on ? klo_now : -1},
    // plane stride the neighbour loops of the running force kernels are compiled for (0: the run-time stride slot_cap)
    {"fast_cap", (ctx->list_mode == 1 && ctx->slot_cap == FAST_CAP && exp_env_int("IMDGPU_FAST_CAP", 1)) ? FAST_CAP : 0},
    {"list_mode", ctx->list_mode}, {"mode", ctx->mode}, {"tile_cap", ctx->tile_cap}, {"slot_cap", ctx->slot_cap},
    {"max_halo", ctx->max_halo}, {"W", ctx->W}, {"tile_x", ctx->tile[0]}, {"tile_y", ctx->tile[1]}, {"tile_z", ctx->tile[2]},
    {"ntiles", ctx->ntiles}, {"tile_threads", ctx->tile_threads}, {"nlocal", ctx->nlocal}, {"nghost", ctx->nghost},
    {"lds_pass1", (long long) ctx->lds_need[1]}, {"lds_pass3", (long long) ctx->lds_need[3]}, {"nbl_count", ctx->nbl_count},
    {"fused_embedding", fuse_emb(ctx) ? 1 : 0},
    // which pass 1 runs: 2 = one type, pair_pot and rho_h_tab interleaved on their shared grid; 1 = one type, each table on its own grid;
    // 0 = the general kernels (several types, pair only, ...)
    {"experiments", imdgpu_experiment_mask()},
    {"rec_lds", ctx->list_mode == 1 && ctx->tile_rec ? ctx->rec_lds : 0},
    {"pass1_form", p1_onetype(ctx) ? (ctx->d_merged ? 2 : 1) : (ctx->list_mode == 1 && ctx->tile_rec ? 3 : 0)},
    // global-gather family: the tables are too large for the LDS and are read in place through L2 (k_pass1_gt / k_pass3_gt)
    {"global_tables_in_place", ctx->list_mode == 0 && ((size_t) (ctx->pair.present ? ctx->pair.par.nrows * ctx->pair.par.ncols : 0) +
                                                       (ctx->rho.present ? ctx->rho.par.nrows * ctx->rho.par.ncols : 0) + 64) * sizeof(double) > TABLE_LDS_LIMIT}, {"ttm_flux", ctx->ttm_flux}, {"ttm_adv", ctx->ttm_adv}, {"ttm_block_width", ctx->ttm_flux ? 10 : 6},
    // bumped whenever a force / list kernel or its data layout changes: committed PMC traffic figures carry the revision
    // they were measured on (profiles/*hbm_traffic*.json "_kernel_rev"), bench.py refuses figures of another revision
    {"kernel_rev", IMDGPU_KERNEL_REV}, {"pass3_ring", ctx->ttm ? 0 : 4}, {"have_valid_nbl", ctx->have_valid_nbl},
    {"ttm_persist_mode", imdgpu_ttm_persist_info(ctx, 0)}, {"ttm_persist_runs", imdgpu_ttm_persist_info(ctx, 1)},
    // decomposed lattice: 0 = faces through the host mailbox, 1 / 2 = ghost records in the neighbours' device memory (coarse- / fine-grained)
    {"ttm_remote", imdgpu_ttm_persist_info(ctx, 2)},
    // MD steps whose sub-steps the host issued again through the launch chain because the persistent kernel had stepped aside
    {"ttm_persist_redone", imdgpu_ttm_persist_info(ctx, 3)},
    // EAM force passes so far in which the three-row kernel stood in for the coefficient form (as of the last md_step / calc_forces)
    {"p3_standin_passes", ctx->h_counters ? ctx->h_counters[CNT_P3_STANDIN] : 0},
    {"rccl_ranks", imdgpu_dist_rccl_info(ctx, 0)}, {"rccl_comms", imdgpu_dist_rccl_info(ctx, 1)}, {"rccl_device", imdgpu_dist_rccl_info(ctx, 2)},
    {"halo_overlap_passes", ctx->dyn_seq}, {"halo_overlap_interior_tiles", ctx->dyn_ntiles == ctx->ntiles ? ctx->n_interior : -1}};       // pass-1 launch pairs under a position halo in flight (option halo_overlap)
  for (auto &e : tab) if (!strcmp(name, e.n)) { *value = e.v; return 0; }
  return imdgpu_fail(ctx, "unknown info %s", name);
}

extern "C" int imdgpu_device_count(int *n)
{
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  if (n) *n = c;
  return c > 0 ? 0 : 1;
}

// identity of the context's GPU as text: "<PCI bus id> uuid <hex> <name>, <CUs> CUs" -- what bench.py gathers from every rank
extern "C" int imdgpu_device_identity(imdgpu_ctx *ctx, char *buf, int len)
{
  if (!buf || len < 1) return imdgpu_fail(ctx, "device_identity: no buffer");
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, ctx->device));
  char bus[64] = "?";
  (void) hipDeviceGetPCIBusId(bus, sizeof(bus), ctx->device);
  char uu[40]; int o = 0;
  for (int q = 0; q < 16; q++) o += snprintf(uu + o, sizeof(uu) - o, "%02x", (unsigned) (unsigned char) p.uuid.bytes[q]);
  snprintf(buf, (size_t) len, "%s uuid %s %s, %d CUs", bus, uu, p.name, p.multiProcessorCount);
  return 0;
}

extern "C" int imdgpu_set_stream(imdgpu_ctx *ctx, void *s)
{
  if (ctx->own_stream && ctx->stream) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipStreamDestroy(ctx->stream)); }
  ctx->stream = (hipStream_t) s;
  ctx->own_stream = false;
  return 0;
}

extern "C" int imdgpu_synchronize(imdgpu_ctx *ctx)
{
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// (re)derive everything that depends on the tile shape: key space, per-key tables, ghost-source keys
static int setup_tiles(imdgpu_ctx *ctx)
{
  const int *D = ctx->dim;
  for (int d = 0; d < 3; d++) {
    if (ctx->tile[d] > D[d] - 2) ctx->tile[d] = D[d] - 2;
    if (ctx->tile[d] < 1) ctx->tile[d] = 1;
    ctx->ntile[d] = (D[d] - 2 + ctx->tile[d] - 1) / ctx->tile[d];
  }
  ctx->CPT = ctx->tile[0] * ctx->tile[1] * ctx->tile[2];
  ctx->ntiles = ctx->ntile[0] * ctx->ntile[1] * ctx->ntile[2];
  ctx->nkeys = ctx->ntiles * ctx->CPT;
  if (dev_alloc(ctx, &ctx->icnt, ctx->nkeys + 1) || dev_alloc(ctx, &ctx->istart, ctx->nkeys + 1)) return 1;
  const Geo g = make_geo(ctx);
  const int ng = ctx->nghost_cells;
  std::vector<int> src(ng);
  for (int q = 0; q < ng; q++) {
    const int *x = &ctx->h_gsrc_xyz[3 * q];
    src[q] = x[0] >= 0 ? xyz_to_key(x[0], x[1], x[2], g) : -1;
  }
  HIPCHK(hipMemcpy(ctx->gcell_src, src.data(), ng * sizeof(int), hipMemcpyHostToDevice));
  int nb = std::max(nblk(ctx->cap_local, 256) + 8, 2 * (((ctx->ntiles + 7) / 8) * 8 + 8));    // twice: pass 1 in two launches (imdgpu_launch_pass1_dyn)
  if (nb > ctx->npart_blocks) {
    if (dev_alloc(ctx, &ctx->partials, (size_t) nb * IMDGPU_NPART)) return 1;
    ctx->npart_blocks = nb;
  }
  ctx->nbr16_cap = 0; ctx->tile_cap = 0;
  ctx->have_valid_nbl = 0;
  return 0;
}

// old cell key -> new cell key after a change of the tile shape (cells themselves are unchanged)
__global__ void k_rekey(int n, const int *acell, Geo gold, Geo gnew, int *keys, int *perm)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int ix, iy, iz;
  key_to_xyz(acell[i], gold, ix, iy, iz);
  keys[i] = xyz_to_key(ix, iy, iz, gnew);
  perm[i] = i;
}

static int sort_and_build(imdgpu_ctx *ctx);

// change the tile shape and re-sort the atoms into the new key order
static int retile(imdgpu_ctx *ctx, const int nt[3])
{
  const Geo gold = make_geo(ctx);
  for (int d = 0; d < 3; d++) ctx->tile[d] = nt[d];
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // keys of the old arrangement are needed by k_rekey: keep acell, rebuild the rest
  if (setup_tiles(ctx)) return 1;
  const Geo gnew = make_geo(ctx);
  hipLaunchKernelGGL(k_rekey, dim3(nblk(ctx->nlocal, 256)), dim3(256), 0, ctx->stream, ctx->nlocal, ctx->acell, gold, gnew,
                     ctx->keys, ctx->perm);
  return sort_and_build(ctx);
}

// shrink the largest tile dimension by one cell
static int shrink_tiles(imdgpu_ctx *ctx)
{
  int d = 0, nt[3] = {ctx->tile[0], ctx->tile[1], ctx->tile[2]};
  for (int q = 1; q < 3; q++) if (nt[q] > nt[d]) d = q;
  if (nt[d] <= 1) return imdgpu_fail(ctx, "tiles cannot shrink further");
  nt[d]--;
  return retile(ctx, nt);
}

static size_t tile_lds_bytes_for(const imdgpu_ctx *ctx, int slot_cap, int which);
static size_t align8(size_t b) { return (b + 7) & ~(size_t) 7; }

// Pick the tile shape from the actual cell occupancies: the largest tile (fewest halo atoms staged per own
// atom, fewest workgroup rounds) whose fullest tile still fits one TILE_MAXTHREADS workgroup and whose
// fullest halo fits the LDS next to the tables.  Runs once per arrangement, on the host.
static int autotile(imdgpu_ctx *ctx)
{
  const int *D = ctx->dim;
  const int nx = D[0] - 2, ny = D[1] - 2, nz = D[2] - 2;
  std::vector<int> cc(ctx->nall_cells);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cc.data(), ctx->cell_cnt, cc.size() * sizeof(int), hipMemcpyDeviceToHost));
  // 3D inclusive prefix sums over the full (ghost-including) cell array, 1-based
  const int X = D[0] + 1, Y = D[1] + 1, Z = D[2] + 1;
  std::vector<long long> P((size_t) X * Y * Z, 0);
  auto at = [&](int i, int j, int k) -> long long & { return P[((size_t) i * Y + j) * Z + k]; };
  for (int i = 1; i < X; i++) for (int j = 1; j < Y; j++) for (int k = 1; k < Z; k++)
    at(i, j, k) = cc[((i - 1) * D[1] + (j - 1)) * D[2] + (k - 1)] + at(i - 1, j, k) + at(i, j - 1, k) + at(i, j, k - 1)
                  - at(i - 1, j - 1, k) - at(i - 1, j, k - 1) - at(i, j - 1, k - 1) + at(i - 1, j - 1, k - 1);
  auto boxsum = [&](int i0, int i1, int j0, int j1, int k0, int k1) {   // full-grid cell ranges [i0,i1) etc., clipped
    i0 = std::max(i0, 0); j0 = std::max(j0, 0); k0 = std::max(k0, 0);
    i1 = std::min(i1, D[0]); j1 = std::min(j1, D[1]); k1 = std::min(k1, D[2]);
    return at(i1, j1, k1) - at(i0, j1, k1) - at(i1, j0, k1) - at(i1, j1, k0) + at(i0, j0, k1) + at(i0, j1, k0) + at(i1, j0, k0) - at(i0, j0, k0);
  };
  int best[3] = {ctx->tile[0], ctx->tile[1], ctx->tile[2]};
  double bestscore = -1;
  struct Cand { double score; int b[3]; };
  std::vector<Cand> cands;
  for (int bx = 1; bx <= 4; bx++) for (int by = 1; by <= 4; by++) for (int bz = 1; bz <= 4; bz++) {
    if (bx > nx || by > ny || bz > nz) continue;
    if ((bx + 2) * (by + 2) * (bz + 2) > NH_MAX) continue;
    long long maxown = 0, maxhalo = 0, ntl = 0, sumhalo = 0;
    for (int tx = 0; tx * bx < nx; tx++) for (int ty = 0; ty * by < ny; ty++) for (int tz = 0; tz * bz < nz; tz++) {
      int i0 = tx * bx + 1, j0 = ty * by + 1, k0 = tz * bz + 1;
      long long own = boxsum(i0, std::min(i0 + bx, nx + 1), j0, std::min(j0 + by, ny + 1), k0, std::min(k0 + bz, nz + 1));
      long long halo = boxsum(i0 - 1, i0 + bx + 1, j0 - 1, j0 + by + 1, k0 - 1, k0 + bz + 1);
      maxown = std::max(maxown, own); maxhalo = std::max(maxhalo, halo);
      if (own > 0) { ntl++; sumhalo += halo; }            // tiles in vacuum cost nothing (block ranges are cut by atoms)
    }
    if (maxown > TILE_MAXTHREADS) continue;
    int slot_cap = (int) ((maxhalo + 1 + 63) / 64 * 64);
    if (slot_cap < FAST_CAP && std::max(tile_lds_bytes_for(ctx, FAST_CAP, 1), tile_lds_bytes_for(ctx, FAST_CAP, 3)) <= LDS_LIMIT) slot_cap = FAST_CAP;
    if (std::max(tile_lds_bytes_for(ctx, slot_cap, 1), tile_lds_bytes_for(ctx, slot_cap, 3)) > LDS_LIMIT) continue;
    // score: atoms per workgroup round, discounted by the wave imbalance over the 4 SIMDs and by the
    // per-tile staging cost (grows with staged/own atoms; weight fitted on MI355X timings)
    int threads = (int) std::min<long long>(TILE_MAXTHREADS, (maxown + 63) / 64 * 64), waves = threads / 64;
    if (ntl == 0) ntl = 1;
    // ties: long in z (contiguous cell keys), then thin in x (measured on a slab: 2x4x4 25 % faster than 4x2x4)
    double score = (double) ctx->nlocal / ntl / ((waves + 3) / 4) / (1.0 + 0.3 * (double) sumhalo / std::max(1, ctx->nlocal))
                   + 1e-3 * bz + 1e-4 * by - 1e-4 * bx;
    // one atom type: a halo so large that the gradient-coefficient table of the EAM force pass no longer fits beside its four planes
    // leaves that pass its three-row form (a quarter slower: a tenth of the step)
    if (ctx->ntypes == 1 && ctx->d_coef3 && ctx->p3_coef) {
      const size_t fixed = align8((4 * (size_t) slot_cap + 64 + 2) * 8);
      const long long fit = fixed + 64 * 16 > LDS_LIMIT ? 0 : (long long) ((LDS_LIMIT - fixed) / 16);
      if (ctx->ncoef - fit > ctx->ncoef / 4) score *= 0.9;
    }
    if (score > bestscore) { bestscore = score; best[0] = bx; best[1] = by; best[2] = bz; }
    cands.push_back({score, {bx, by, bz}});
  }
  // shapes within 6 % of the best score are equivalent by this model: among them thin in x (the slowest index), then long in
  // z, then in y -- measured on a 10 M-atom slab: 2x4x4 6.87 ms per step, 4x2x4 (same score) 8.59; on the Maxwell-solver slab
  // (148 x 78 x 78 cells: 4x2x4 divides evenly and scores 5 % above 2x4x4, whose last tiles along y and z are half empty)
  // 2x4x4 7.20 against 7.47 ms per step
  for (const Cand &c : cands)
    if (c.score >= 0.94 * bestscore) {
      const bool better = c.b[0] != best[0] ? c.b[0] < best[0] : (c.b[2] != best[2] ? c.b[2] > best[2] : c.b[1] > best[1]);
      if (better) { best[0] = c.b[0]; best[1] = c.b[1]; best[2] = c.b[2]; }
    }
  ctx->autotiled = 1;
  if (bestscore < 0 || (best[0] == ctx->tile[0] && best[1] == ctx->tile[1] && best[2] == ctx->tile[2])) return 0;
  return retile(ctx, best);
}

extern "C" int imdgpu_set_geometry(imdgpu_ctx *ctx, const double box[9], const int pbc[3], const int cell_dim[3],
                                   const int gdim[3], const int cpu_dim[3], const int my_coord[3])
{
  HIPCHK(hipSetDevice(ctx->device));
  memcpy(ctx->box, box, sizeof(ctx->box));
  invert_box(ctx->box, ctx->tbox);
  for (int d = 0; d < 3; d++) {
    ctx->pbc[d] = pbc[d]; ctx->dim[d] = cell_dim[d]; ctx->gdim[d] = gdim[d];
    ctx->cpu_dim[d] = cpu_dim[d]; ctx->my_coord[d] = my_coord[d];
    if (cell_dim[d] < 3) return imdgpu_fail(ctx, "cell_dim[%d]=%d: need at least one inner cell", d, cell_dim[d]);
    if ((cell_dim[d] - 2) * cpu_dim[d] != gdim[d])
      return imdgpu_fail(ctx, "cell_dim/global_cell_dim/cpu_dim inconsistent in direction %d", d);
    if (cpu_dim[d] == 1 && pbc[d] == 1 && gdim[d] < 3)
      return imdgpu_fail(ctx, "periodic direction %d needs >= 3 cells (has %d)", d, gdim[d]);
  }
  const int *D = ctx->dim;
  ctx->external_halo = (cpu_dim[0] * cpu_dim[1] * cpu_dim[2] > 1 || ctx->force_external_halo) ? 1 : 0;
  ctx->nall_cells = D[0] * D[1] * D[2];
  ctx->ninner_cells = (D[0] - 2) * (D[1] - 2) * (D[2] - 2);
  ctx->nghost_cells = ctx->nall_cells - ctx->ninner_cells;
  // tiling of the inner cells (one force-kernel workgroup per tile): 3 cells per direction (2 where that
  // avoids a one-cell remainder); make_nblist shrinks tiles that do not fit the LDS.  IMDGPU_TILE="bx,by,bz".
  for (int d = 0; d < 3; d++) {
    int n = D[d] - 2;
    ctx->tile[d] = n >= 3 ? ((n % 3 == 1 && n % 2 == 0) ? 2 : 3) : n;
  }
  if (const char *e = cfg_env("IMDGPU_TILE")) {
    int a, b, c;
    if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { ctx->tile[0] = a; ctx->tile[1] = b; ctx->tile[2] = c; }
  }
  if (const char *e = cfg_env("IMDGPU_MODE")) ctx->mode = atoi(e);
  // static ghost-cell tables: source inner cell (periodic wrap on this rank) and shift code
  std::vector<int> lin, sh;
  ctx->h_gsrc_xyz.clear();
  for (int i = 0; i < D[0]; i++)
    for (int j = 0; j < D[1]; j++)
      for (int k = 0; k < D[2]; k++) {
        int c[3] = {i, j, k}, s[3], sc[3] = {0, 0, 0};
        bool ghost = false, active = true;
        for (int d = 0; d < 3; d++) {
          s[d] = c[d];
          if (c[d] == 0) { ghost = true; s[d] = D[d] - 2; sc[d] = -1; }
          else if (c[d] == D[d] - 1) { ghost = true; s[d] = 1; sc[d] = +1; }
          if (sc[d] != 0) {
            if (cpu_dim[d] > 1) active = false;        // filled by the halo exchange, not by local images
            else if (pbc[d] != 1) active = false;      // open boundary: nothing beyond it
          }
        }
        if (!ghost) continue;
        if (ctx->external_halo) active = false;       // multi-rank: every ghost cell is filled through imdgpu_halo_scatter
        lin.push_back((i * D[1] + j) * D[2] + k);
        for (int d = 0; d < 3; d++) ctx->h_gsrc_xyz.push_back(active ? s[d] - 1 : -1);
        sh.push_back((sc[0] + 1) + 3 * (sc[1] + 1) + 9 * (sc[2] + 1));
      }
  int ng = ctx->nghost_cells;
  if (dev_alloc(ctx, &ctx->gcell_lin, ng) || dev_alloc(ctx, &ctx->gcell_src, ng) || dev_alloc(ctx, &ctx->gcell_shift, ng) ||
      dev_alloc(ctx, &ctx->gcnt, ng + 1) || dev_alloc(ctx, &ctx->gstart, ng + 1) ||
      dev_alloc(ctx, &ctx->cell_start, ctx->nall_cells) || dev_alloc(ctx, &ctx->cell_cnt, ctx->nall_cells))
    return 1;
  HIPCHK(hipMemcpy(ctx->gcell_lin, lin.data(), ng * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->gcell_shift, sh.data(), ng * sizeof(int), hipMemcpyHostToDevice));
  if (setup_tiles(ctx)) return 1;
  HIPCHK(hipMemset(ctx->cell_start, 0, ctx->nall_cells * sizeof(int)));
  HIPCHK(hipMemset(ctx->cell_cnt, 0, ctx->nall_cells * sizeof(int)));
  ctx->have_geometry = true;
  ctx->have_valid_nbl = 0;
  return 0;
}

extern "C" int imdgpu_set_params(imdgpu_ctx *ctx, int ntypes, double cellsz, double nbl_margin, double timestep,
                                 int count_ttm)
{
  if (ntypes < 1 || ntypes * ntypes > IMDGPU_MAXCOL) return imdgpu_fail(ctx, "ntypes=%d unsupported (max 4)", ntypes);
  ctx->ntypes = ntypes; ctx->cellsz = cellsz; ctx->nbl_margin = nbl_margin; ctx->timestep = timestep;
  ctx->count_ttm = count_ttm;
  std::vector<double> ones(3 * ntypes, 1.0);
  if (dev_alloc(ctx, &ctx->d_restr, 3 * ntypes)) return 1;
  HIPCHK(hipMemcpy(ctx->d_restr, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
  ctx->restr_identity = 1;
  ctx->have_valid_nbl = 0;
  return 0;
}

extern "C" int imdgpu_set_restrictions(imdgpu_ctx *ctx, const double *r)
{
  if (!ctx->d_restr) return imdgpu_fail(ctx, "set_params first");
  HIPCHK(hipMemcpy(ctx->d_restr, r, 3 * ctx->ntypes * sizeof(double), hipMemcpyHostToDevice));
  ctx->restr_identity = 1;
  for (int q = 0; q < 3 * ctx->ntypes; q++) if (r[q] != 1.0) ctx->restr_identity = 0;
  return 0;
}

static int upload_table(imdgpu_ctx *ctx, DevTable &T, const imdgpu_pot_table *pt, const char *name)
{
  if (T.d_tab) { (void) hipFree(T.d_tab); T.d_tab = nullptr; }
  T.present = false;
  if (!pt) return 0;
  if (pt->ncols < 1 || pt->ncols > IMDGPU_MAXCOL) return imdgpu_fail(ctx, "%s: ncols=%d unsupported", name, pt->ncols);
  size_t n = (size_t) (pt->maxsteps + 2) * pt->ncols;
  HIPCHK(hipMalloc((void **) &T.d_tab, n * sizeof(double)));
  HIPCHK(hipMemcpy(T.d_tab, pt->table, n * sizeof(double), hipMemcpyHostToDevice));
  T.par.tab = T.d_tab; T.par.ncols = pt->ncols; T.par.nrows = pt->maxsteps + 2;
  for (int c = 0; c < pt->ncols; c++) { T.par.begin[c] = pt->begin[c]; T.par.end[c] = pt->end[c]; T.par.invstep[c] = pt->invstep[c]; }
  T.present = true;
  return 0;
}

extern "C" int imdgpu_upload_tables(imdgpu_ctx *ctx, const imdgpu_pot_table *pair, const imdgpu_pot_table *rho,
                                    const imdgpu_pot_table *emb)
{
  HIPCHK(hipSetDevice(ctx->device));
  if ((rho == nullptr) != (emb == nullptr)) return imdgpu_fail(ctx, "rho_h_tab and embed_pot must be given together");
  if (upload_table(ctx, ctx->pair, pair, "pair_pot") || upload_table(ctx, ctx->rho, rho, "rho_h_tab") ||
      upload_table(ctx, ctx->emb, emb, "embed_pot")) return 1;
  int nn = ctx->ntypes * ctx->ntypes;
  if (pair && pair->ncols != nn) return imdgpu_fail(ctx, "pair_pot has %d columns, expected ntypes^2=%d", pair->ncols, nn);
  if (rho && rho->ncols != nn) return imdgpu_fail(ctx, "rho_h_tab has %d columns, expected %d", rho->ncols, nn);
  if (emb && emb->ncols != ctx->ntypes) return imdgpu_fail(ctx, "embed_pot has %d columns, expected %d", emb->ncols, ctx->ntypes);
  if (ctx->d_merged) { (void) hipFree(ctx->d_merged); ctx->d_merged = nullptr; }
  if (ctx->d_coef3) { (void) hipFree(ctx->d_coef3); ctx->d_coef3 = nullptr; ctx->ncoef = 0; }
  if (pair && rho && ctx->ntypes == 1 && !ctx->no_merge && pair->maxsteps == rho->maxsteps && pair->begin[0] == rho->begin[0] &&
      pair->end[0] == rho->end[0] && pair->invstep[0] == rho->invstep[0]) {
    const int R = pair->maxsteps + 2;
    std::vector<double> m(2 * (size_t) (R + 3), 0.0);
    for (int k = 0; k < R; k++) { m[2 * k] = pair->table[k]; m[2 * k + 1] = rho->table[k]; }
    HIPCHK(hipMalloc((void **) &ctx->d_merged, m.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_merged, m.data(), m.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (pair && rho && ctx->ntypes == 1) {
    // gradient coefficients of rho_h_tab for k_pass3_tile_coef: rows 0 .. maxsteps-1 (row k needs table rows k .. k+2), then a zero row
    const int NC = rho->maxsteps + 1;
    std::vector<double> cf(2 * (size_t) NC, 0.0);
    for (int k = 0; k + 1 < NC; k++) {
      const double p0 = rho->table[k], p1 = rho->table[k + 1], p2 = rho->table[k + 2];
      const double dv = p1 - p0, d2v = (p2 - 2 * p1) + p0;
      cf[2 * k] = dv - 0.5 * d2v; cf[2 * k + 1] = d2v;
    }
    HIPCHK(hipMalloc((void **) &ctx->d_coef3, cf.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_coef3, cf.data(), cf.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->ncoef = NC;
  }
  // several atom types: coefficient records per column and row for the tile kernels that read the tables through L2
  // (k_pass1_tile_rec / k_pass3_tile_rec; make_nblist picks them when the tables do not fit the LDS beside a tile)
  for (void **q : {(void **) &ctx->d_recP, (void **) &ctx->d_recR, (void **) &ctx->d_rec3, (void **) &ctx->d_compP, (void **) &ctx->d_compR})
    if (*q) { (void) hipFree(*q); *q = nullptr; }
  ctx->nrecP = ctx->nrecR = ctx->ncompP = ctx->ncompR = 0;
  if (pair && rho && ctx->ntypes > 1) {
    auto records = [&](const imdgpu_pot_table *t, std::vector<double> &rec, std::vector<double> *rec3, int &nrec) {
      const int nc = t->ncols, R = t->maxsteps + 2;
      nrec = R - 1;                                   // rows 0 .. R-3 (row k needs table rows k .. k+2) + one zero record
      rec.assign((size_t) nc * nrec * 4, 0.0);
      if (rec3) rec3->assign((size_t) nc * nrec * 2, 0.0);
      for (int c = 0; c < nc; c++)
        for (int k = 0; k + 2 < R; k++) {
          const double p0 = t->table[(size_t) k * nc + c], p1 = t->table[(size_t) (k + 1) * nc + c], p2 = t->table[(size_t) (k + 2) * nc + c];
          const double dv = p1 - p0, d2v = (p2 - 2 * p1) + p0;            // potaccess.h:338-344
          double *r = &rec[((size_t) c * nrec + k) * 4];
          r[0] = p0; r[1] = dv; r[2] = d2v;
          if (rec3) { double *q = &(*rec3)[((size_t) c * nrec + k) * 2]; q[0] = t->invstep[c] * (dv - 0.5 * d2v); q[1] = t->invstep[c] * d2v; }
        }
    };
    std::vector<double> rp, rr, r3;
    records(pair, rp, nullptr, ctx->nrecP);
    records(rho, rr, &r3, ctx->nrecR);
    HIPCHK(hipMalloc((void **) &ctx->d_recP, rp.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_recP, rp.data(), rp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **) &ctx->d_recR, rr.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_recR, rr.data(), rr.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **) &ctx->d_rec3, r3.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_rec3, r3.data(), r3.size() * sizeof(double), hipMemcpyHostToDevice));
    // the compact form for the LDS: a column that repeats an earlier one (same begin / end / step / length, same rows -- phi_AB = phi_BA
    // of a pair potential, a density that depends on the neighbour's type only) shares its rows
    auto compact = [&](const imdgpu_pot_table *t, std::vector<double> &out, int *loff) {
      const int nc = t->ncols;
      out.clear();
      for (int c = 0; c < nc; c++) {
        const int rows = t->len[c] + 2;
        int same = -1;
        for (int b = 0; b < c && same < 0; b++) {
          if (t->len[b] != t->len[c] || t->begin[b] != t->begin[c] || t->end[b] != t->end[c] || t->invstep[b] != t->invstep[c]) continue;
          same = b;
          for (int k = 0; k < rows; k++) if (t->table[(size_t) k * nc + b] != t->table[(size_t) k * nc + c]) { same = -1; break; }
        }
        if (same >= 0) { loff[c] = loff[same]; continue; }
        loff[c] = (int) out.size();
        for (int k = 0; k < rows; k++) out.push_back(t->table[(size_t) k * nc + c]);
      }
      out.insert(out.end(), 3, 0.0);
    };
    std::vector<double> cp, cr;
    compact(pair, cp, ctx->loffP);
    compact(rho, cr, ctx->loffR);
    ctx->ncompP = (int) cp.size(); ctx->ncompR = (int) cr.size();
    HIPCHK(hipMalloc((void **) &ctx->d_compP, cp.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_compP, cp.data(), cp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **) &ctx->d_compR, cr.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_compR, cr.data(), cr.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return 0;
}

static int ensure_atom_capacity(imdgpu_ctx *ctx, int nlocal, int ntot)
{
  if (nlocal > ctx->cap_local || ctx->cap_local == 0) {
    int cap = (int) (nlocal * 1.1) + 1024;
    cap = (cap + 63) / 64 * 64;
    if (dev_alloc(ctx, &ctx->id, cap) || dev_alloc(ctx, &ctx->acell, cap) || dev_alloc(ctx, &ctx->p, 3 * (size_t) cap) ||
        dev_alloc(ctx, &ctx->f, 3 * (size_t) cap) || dev_alloc(ctx, &ctx->ref, 3 * (size_t) cap) ||
        dev_alloc(ctx, &ctx->mass, cap) || dev_alloc(ctx, &ctx->epot, cap) || dev_alloc(ctx, &ctx->rhoh, cap) ||
        dev_alloc(ctx, &ctx->numneighs, cap) || dev_alloc(ctx, &ctx->id2, cap) || dev_alloc(ctx, &ctx->p2, 3 * (size_t) cap) ||
        dev_alloc(ctx, &ctx->mass2, cap) || dev_alloc(ctx, &ctx->keys, cap) || dev_alloc(ctx, &ctx->keys2, cap) ||
        dev_alloc(ctx, &ctx->perm, cap) || dev_alloc(ctx, &ctx->perm2, cap) || dev_alloc(ctx, &ctx->nbr_cnt, cap) ||
        dev_alloc(ctx, &ctx->fdc, cap) || dev_alloc(ctx, &ctx->fdc2, cap))
      return 1;
    HIPCHK(hipMemset(ctx->fdc, 0xff, cap * sizeof(int)));
    HIPCHK(hipMemset(ctx->f, 0, 3 * (size_t) cap * sizeof(double)));
    HIPCHK(hipMemset(ctx->epot, 0, cap * sizeof(double)));
    HIPCHK(hipMemset(ctx->rhoh, 0, cap * sizeof(double)));
    HIPCHK(hipMemset(ctx->numneighs, 0, cap * sizeof(int)));
    HIPCHK(hipMemset(ctx->ref, 0, 3 * (size_t) cap * sizeof(double)));
    ctx->cap_local = cap;
    int nb = std::max(nblk(cap, 256) + 8, 2 * (((ctx->ntiles + 7) / 8) * 8 + 8));
    if (dev_alloc(ctx, &ctx->partials, (size_t) nb * IMDGPU_NPART)) return 1;
    ctx->npart_blocks = nb;
    ctx->nbr_cap = 0;                        // list must be re-allocated for the new stride
  }
  if (ntot > ctx->cap_tot || ctx->cap_tot == 0) {
    int cap = (int) (ntot * 1.1) + 4096;
    double4 *np = nullptr, *np2 = nullptr; int *nt = nullptr, *nt2 = nullptr;
    HIPCHK(hipMalloc((void **) &np, (size_t) cap * sizeof(double4)));
    HIPCHK(hipMalloc((void **) &np2, (size_t) cap * sizeof(double4)));
    HIPCHK(hipMalloc((void **) &nt, (size_t) cap * sizeof(int)));
    HIPCHK(hipMalloc((void **) &nt2, (size_t) cap * sizeof(int)));
    if (ctx->pos && ctx->nlocal > 0) {
      HIPCHK(hipMemcpy(np, ctx->pos, (size_t) ctx->nlocal * sizeof(double4), hipMemcpyDeviceToDevice));
      HIPCHK(hipMemcpy(nt, ctx->type, (size_t) ctx->nlocal * sizeof(int), hipMemcpyDeviceToDevice));
    }
    if (ctx->pos) { (void) hipFree(ctx->pos); (void) hipFree(ctx->pos2); (void) hipFree(ctx->type); (void) hipFree(ctx->type2); }
    ctx->pos = np; ctx->pos2 = np2; ctx->type = nt; ctx->type2 = nt2;
    ctx->cap_tot = cap;
  }
  return 0;
}

// scan + ghost bookkeeping from per-inner-cell counts already in ctx->icnt
static int build_cells_from_counts(imdgpu_ctx *ctx)
{
  ctx->cells_epoch++;                              // the host-order permutation of the level-1 boundary is stale now
  Geo g = make_geo(ctx);
  const int ni = ctx->nkeys, ng = ctx->nghost_cells;
  size_t need = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, ctx->icnt, ctx->istart, ni + 1, ctx->stream));
  size_t need2 = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need2, ctx->gcnt, ctx->gstart, ng + 1, ctx->stream));
  if (need2 > need) need = need2;
  if (need > ctx->cub_tmp_bytes) {
    if (ctx->cub_tmp) (void) hipFree(ctx->cub_tmp);
    HIPCHK(hipMalloc(&ctx->cub_tmp, need * 2 + 1024));
    ctx->cub_tmp_bytes = need * 2 + 1024;
  }
  size_t tb = ctx->cub_tmp_bytes;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cub_tmp, tb, ctx->icnt, ctx->istart, ni + 1, ctx->stream));
  hipLaunchKernelGGL(k_cells_inner, dim3(nblk(ni, 256)), dim3(256), 0, ctx->stream, ni, g, ctx->icnt, ctx->istart,
                     ctx->cell_start, ctx->cell_cnt);
  hipLaunchKernelGGL(k_ghost_cnt, dim3(nblk(ng, 256)), dim3(256), 0, ctx->stream, ng, ctx->gcell_src, ctx->icnt, ctx->gcnt);
  tb = ctx->cub_tmp_bytes;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cub_tmp, tb, ctx->gcnt, ctx->gstart, ng + 1, ctx->stream));
  int nghost = 0;
  HIPCHK(hipMemcpyAsync(&nghost, ctx->gstart + ng, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->nghost = nghost;
  if (ensure_atom_capacity(ctx, ctx->nlocal, ctx->nlocal + nghost)) return 1;
  if (nghost > ctx->cap_ghost) {
    int cap = (int) (nghost * 1.1) + 4096;
    if (dev_alloc(ctx, &ctx->ghost_src, cap) || dev_alloc(ctx, &ctx->ghost_code, cap)) return 1;
    ctx->cap_ghost = cap;
  }
  hipLaunchKernelGGL(k_ghost_map, dim3(nblk(ng, 256)), dim3(256), 0, ctx->stream, ng, ctx->nlocal, ctx->gcell_lin,
                     ctx->gcell_src, ctx->gcell_shift, ctx->gcnt, ctx->gstart, ctx->istart, ctx->cell_start,
                     ctx->cell_cnt, ctx->ghost_src, ctx->ghost_code);
  if (nghost > 0)
    hipLaunchKernelGGL(k_ghost_update, dim3(nblk(nghost, 256)), dim3(256), 0, ctx->stream, nghost, ctx->nlocal,
                       ctx->ghost_src, ctx->ghost_code, ctx->pos, ctx->type, g, 1);
  HIPCHK(hipGetLastError());
  ctx->have_valid_nbl = 0;
  ctx->need_fix_cells = 0;
  return 0;
}

static int upload_common(imdgpu_ctx *ctx, int n, const double *ort, const double *impuls, const double *masse,
                         const int *sorte, const int *nummer)
{
  imdgpu_dist_cancel_ahead(ctx);
  if (!ctx->have_geometry) return imdgpu_fail(ctx, "imdgpu_set_geometry must be called first");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->nlocal = 0;
  if (ensure_atom_capacity(ctx, n, n)) return 1;
  HIPCHK(hipMemset(ctx->fdc, 0xff, (size_t) ctx->cap_local * sizeof(int)));      // no FD cell known yet for any atom
  const int cap = ctx->cap_local;
  std::vector<double4> hp(n);
  std::vector<double> hm(3 * (size_t) n);
  std::vector<int> ht(n), hid(n);
  for (int i = 0; i < n; i++) {
    hp[i] = make_double4(ort[3 * i], ort[3 * i + 1], ort[3 * i + 2], 0.0);
    hm[i] = impuls ? impuls[3 * i] : 0.0;
    hm[(size_t) n + i] = impuls ? impuls[3 * i + 1] : 0.0;
    hm[2 * (size_t) n + i] = impuls ? impuls[3 * i + 2] : 0.0;
    ht[i] = sorte ? sorte[i] : 0;
    hid[i] = nummer ? nummer[i] : i;
    if (ht[i] < 0 || ht[i] >= ctx->ntypes) return imdgpu_fail(ctx, "atom %d has type %d outside [0,%d)", i, ht[i], ctx->ntypes);
  }
  HIPCHK(hipMemcpy(ctx->pos, hp.data(), n * sizeof(double4), hipMemcpyHostToDevice));
  for (int d = 0; d < 3; d++)
    HIPCHK(hipMemcpy(ctx->p + (size_t) d * cap, hm.data() + (size_t) d * n, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->mass, masse, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->type, ht.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->id, hid.data(), n * sizeof(int), hipMemcpyHostToDevice));
  ctx->nlocal = n;
  ctx->have_atoms = 1;
  return 0;
}

extern "C" int imdgpu_upload_cells(imdgpu_ctx *ctx, const int *cell_n, int natoms, const double *ort,
                                   const double *impuls, const double *masse, const int *sorte, const int *nummer)
{
  if (upload_common(ctx, natoms, ort, impuls, masse, sorte, nummer)) return 1;
  // keys from the host's cell assignment (NOT from positions: atoms may have drifted out of the cell
  // they are stored in since the host's last fix_cells); the stable device sort keeps the slot order
  const int *D = ctx->dim;
  const Geo g = make_geo(ctx);
  std::vector<int> keys(natoms), perm(natoms);
  int at = 0;
  for (int i = 1; i < D[0] - 1; i++)
    for (int j = 1; j < D[1] - 1; j++)
      for (int k = 1; k < D[2] - 1; k++) {
        int n = cell_n[(i * D[1] + j) * D[2] + k];
        if (n < 0 || at + n > natoms) return imdgpu_fail(ctx, "cell_n does not add up to natoms=%d", natoms);
        int key = xyz_to_key(i - 1, j - 1, k - 1, g);
        for (int t = 0; t < n; t++) { keys[at + t] = key; perm[at + t] = at + t; }
        at += n;
      }
  if (at != natoms) return imdgpu_fail(ctx, "cell_n sums to %d, natoms=%d", at, natoms);
  HIPCHK(hipMemcpy(ctx->keys, keys.data(), natoms * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->perm, perm.data(), natoms * sizeof(int), hipMemcpyHostToDevice));
  return sort_and_build(ctx);
}


// ---- host-buffer boundary: reorder on the device, move whole arrays over PCIe --------------------------------------------
// perm[r] = device index of the atom that is r-th in the reference's cell order (build_ref_perm), kept on the device
static int ensure_refperm(imdgpu_ctx *ctx)
{
  if (ctx->refperm_epoch == ctx->cells_epoch && ctx->d_refperm) return 0;
  std::vector<int> perm;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (build_ref_perm(ctx, &perm, nullptr)) return 1;
  if ((size_t) ctx->nlocal > ctx->refperm_cap) {
    if (dev_alloc(ctx, &ctx->d_refperm, (size_t) ctx->cap_local)) return 1;
    ctx->refperm_cap = (size_t) ctx->cap_local;
  }
  if (ctx->nlocal > 0) HIPCHK(hipMemcpy(ctx->d_refperm, perm.data(), (size_t) ctx->nlocal * sizeof(int), hipMemcpyHostToDevice));
  ctx->refperm_epoch = ctx->cells_epoch;
  return 0;
}
static int ensure_stage(imdgpu_ctx *ctx, size_t bytes)
{
  if (bytes <= ctx->stage_bytes) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->d_stage) (void) hipFree(ctx->d_stage);
  ctx->stage_bytes = bytes + bytes / 8 + 4096;
  HIPCHK(hipMalloc(&ctx->d_stage, ctx->stage_bytes));
  return 0;
}
__global__ void k_l1_put_pos(int n, const int *perm, const double *in, double4 *pos)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int i = perm[r];
  pos[i].x = in[3 * (size_t) r]; pos[i].y = in[3 * (size_t) r + 1]; pos[i].z = in[3 * (size_t) r + 2];
}
__global__ void k_l1_put_planes(int n, int cap, const int *perm, const double *in, double *planes)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int i = perm[r];
  planes[i] = in[3 * (size_t) r]; planes[cap + i] = in[3 * (size_t) r + 1]; planes[2 * (size_t) cap + i] = in[3 * (size_t) r + 2];
}
__global__ void k_l1_get_planes(int n, int cap, const int *perm, const double *planes, double *out)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int i = perm[r];
  out[3 * (size_t) r] = planes[i]; out[3 * (size_t) r + 1] = planes[cap + i]; out[3 * (size_t) r + 2] = planes[2 * (size_t) cap + i];
}
__global__ void k_l1_get_pos(int n, const int *perm, const double4 *pos, double *ort, double *dF)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double4 q = pos[perm[r]];
  if (ort) { ort[3 * (size_t) r] = q.x; ort[3 * (size_t) r + 1] = q.y; ort[3 * (size_t) r + 2] = q.z; }
  if (dF) dF[r] = q.w;
}
template <typename T>
__global__ void k_l1_get_scalar(int n, const int *perm, const T *in, T *out)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = in[perm[r]];
}

// page-locked host memory for the arrays a host hands to imdgpu_update_* / imdgpu_download_atoms every step: transfers
// from / to such memory run at PCIe speed and asynchronously (pageable memory is staged by the runtime at a fraction of it)
extern "C" int imdgpu_host_alloc(void **p, size_t bytes)
{
  return hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : 1;
}
extern "C" int imdgpu_host_free(void *p) { return p == nullptr || hipHostFree(p) == hipSuccess ? 0 : 1; }

// new positions for the CURRENT arrangement (cell ordered like imdgpu_download_atoms returns them); the
// neighbour list stays valid -- what a host integrator calls between two list rebuilds
extern "C" int imdgpu_update_positions(imdgpu_ctx *ctx, int natoms, const double *ort)
{
  imdgpu_dist_cancel_ahead(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  if (natoms != ctx->nlocal) return imdgpu_fail(ctx, "update_positions: %d atoms given, %d on the device", natoms, ctx->nlocal);
  if (natoms == 0) return 0;
  if (ensure_refperm(ctx) || ensure_stage(ctx, (size_t) natoms * 24)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->d_stage, ort, (size_t) natoms * 24, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_l1_put_pos, dim3(nblk(natoms, 256)), dim3(256), 0, ctx->stream, natoms, ctx->d_refperm, (const double *) ctx->d_stage, ctx->pos);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));            // the caller may reuse `ort` at once
  return 0;
}

// new momenta for the current arrangement (cell ordered): a host integrator that keeps calc_ttm on the device
extern "C" int imdgpu_update_momenta(imdgpu_ctx *ctx, int natoms, const double *impuls)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (natoms != ctx->nlocal) return imdgpu_fail(ctx, "update_momenta: %d atoms given, %d on the device", natoms, ctx->nlocal);
  if (natoms == 0) return 0;
  if (ensure_refperm(ctx) || ensure_stage(ctx, (size_t) natoms * 24)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->d_stage, impuls, (size_t) natoms * 24, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_l1_put_planes, dim3(nblk(natoms, 256)), dim3(256), 0, ctx->stream, natoms, ctx->cap_local, ctx->d_refperm,
                     (const double *) ctx->d_stage, ctx->p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int imdgpu_upload_atoms(imdgpu_ctx *ctx, int natoms, const double *ort, const double *impuls,
                                   const double *masse, const int *sorte, const int *nummer)
{
  if (upload_common(ctx, natoms, ort, impuls, masse, sorte, nummer)) return 1;
  return imdgpu_fix_cells(ctx);
}

extern "C" int imdgpu_num_atoms(imdgpu_ctx *ctx, int *nlocal, int *nghost)
{
  if (nlocal) *nlocal = ctx->nlocal;
  if (nghost) *nghost = ctx->nghost;
  return 0;
}

// stable sort of the local atoms by cell key (ctx->keys / ctx->perm filled), gather, cell tables
static int sort_and_build(imdgpu_ctx *ctx)
{
  const int n = ctx->nlocal, cap = ctx->cap_local;
  {
    ProfScope ps(ctx, PROF_FIXCELLS);
    int bits = 1;
    while ((1 << bits) < ctx->nkeys) bits++;
    size_t need = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, ctx->keys, ctx->keys2, ctx->perm, ctx->perm2, n, 0, bits, ctx->stream));
    if (need > ctx->cub_tmp_bytes) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (ctx->cub_tmp) (void) hipFree(ctx->cub_tmp);
      HIPCHK(hipMalloc(&ctx->cub_tmp, need + 1024));
      ctx->cub_tmp_bytes = need + 1024;
    }
    size_t tb = ctx->cub_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp, tb, ctx->keys, ctx->keys2, ctx->perm, ctx->perm2, n, 0, bits, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->icnt, 0, (ctx->nkeys + 1) * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, cap, ctx->perm2, ctx->keys2,
                       ctx->pos, ctx->pos2, ctx->type, ctx->type2, ctx->id, ctx->id2, ctx->p, ctx->p2, ctx->mass,
                       ctx->mass2, ctx->acell, ctx->icnt, ctx->fdc, ctx->fdc2);
    std::swap(ctx->pos, ctx->pos2); std::swap(ctx->type, ctx->type2); std::swap(ctx->id, ctx->id2);
    std::swap(ctx->p, ctx->p2); std::swap(ctx->mass, ctx->mass2); std::swap(ctx->fdc, ctx->fdc2);
  }
  return build_cells_from_counts(ctx);
}

extern "C" int imdgpu_fix_cells(imdgpu_ctx *ctx)
{
  imdgpu_dist_cancel_ahead(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal;                     // 0 is legal (a rank in vacuum)
  if (!ctx->have_atoms) return imdgpu_fail(ctx, "no atoms uploaded");
  Geo g = make_geo(ctx);
  HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_OUTSIDE, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_wrap_key, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, ctx->pos, g, ctx->keys, ctx->perm,
                     ctx->d_counters);
  if (sort_and_build(ctx)) return 1;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost));
  if (ctx->h_counters[CNT_OUTSIDE] > 0 && (ctx->cpu_dim[0] * ctx->cpu_dim[1] * ctx->cpu_dim[2] > 1))
    return imdgpu_fail(ctx, "%d atoms left this rank's domain: migrate them with the halo layer before fix_cells",
                       ctx->h_counters[CNT_OUTSIDE]);
  return 0;
}

extern "C" int imdgpu_fill_ghosts(imdgpu_ctx *ctx)
{
  if (ctx->external_halo) return 0;            // ghost atoms are written by imdgpu_halo_scatter (multi-rank)
  if (ctx->nghost <= 0) return 0;
  ProfScope ps(ctx, PROF_GHOSTS);
  Geo g = make_geo(ctx);
  hipLaunchKernelGGL(k_ghost_update, dim3(nblk(ctx->nghost, 256)), dim3(256), 0, ctx->stream, ctx->nghost, ctx->nlocal,
                     ctx->ghost_src, ctx->ghost_code, ctx->pos, ctx->type, g, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

// persistent tile kernels: one workgroup per CU (their ~140 KB of LDS admit no second one)
static int tile_grid(const imdgpu_ctx *ctx)
{
  static const int forced = exp_env_int("IMDGPU_TILE_GRID", 0);
  if (forced > 0) return std::min(ctx->ntiles, forced);
  // reserve_cus: CUs left free by the persistent force kernels for a multi-wave workgroup of the lattice stream that cannot
  // sit beside them on one CU (the single-workgroup Maxwell-solver loop, imdgpu_ttm_fdtd_init)
  return std::min(ctx->ntiles, (ctx->num_cus > 0 ? ctx->num_cus : 256) - ctx->reserve_cus);
}

static TileArgs make_tile_args(imdgpu_ctx *ctx)
{
  TileArgs a;
  a.nlocal = ctx->nlocal; a.ntypes = ctx->ntypes; a.cap = ctx->cap_local; a.W4 = ctx->W / 4;
  a.tile_cap = ctx->tile_cap; a.slot_cap = ctx->slot_cap; a.ntiles = ctx->ntiles;
  a.g = make_geo(ctx);
  a.pos = ctx->pos; a.type = ctx->type; a.acell = ctx->acell; a.cell_start = ctx->cell_start; a.cell_cnt = ctx->cell_cnt;
  a.istart = ctx->istart; a.nbr16 = (ushort4 *) ctx->nbr16; a.nbr_cnt = ctx->nbr_cnt;
  a.tsrc = ctx->tsrc; a.ownw = ctx->ownw; a.tinfo = (int4 *) ctx->tinfo;
  a.ownseg = ctx->ownw ? ctx->ownw + (size_t) ctx->ntiles * ctx->tile_cap : nullptr;        // allocated behind ownw (several types)
  { static const int dbg = exp_env_int("IMDGPU_DBG", 0); a.dbg = dbg; }
  a.pair = ctx->pair.par; a.rho = ctx->rho.par; a.emb = ctx->emb.par; a.posw = ctx->pos; a.merged = ctx->d_merged;
  a.coef3 = (const double2 *) ctx->d_coef3; a.ncoef = ctx->ncoef; a.klo = 0; a.p3_fallback = 0;
  a.recP = (const double4 *) ctx->d_recP; a.recR = (const double4 *) ctx->d_recR; a.rec3 = (const double2 *) ctx->d_rec3;
  a.nrecP = ctx->nrecP; a.nrecR = ctx->nrecR;
  a.compP = ctx->d_compP; a.compR = ctx->d_compR; a.ncompP = ctx->ncompP; a.ncompR = ctx->ncompR;
  memcpy(a.loffP, ctx->loffP, sizeof(a.loffP)); memcpy(a.loffR, ctx->loffR, sizeof(a.loffR));
  a.f = ctx->f; a.epot = ctx->epot; a.rhoh = ctx->rhoh; a.ref = ctx->ref; a.numneighs = ctx->numneighs;
  a.partials = ctx->partials; a.counters = ctx->d_counters; a.cellsz = ctx->cellsz; a.guard = nullptr;
  a.balance = ctx->tile_balance; a.wprefix = ctx->d_wprefix;
  {
    // k_nbl_build_tile32: how far the single-precision r2 can be from the reference's.  Staged coordinates are relative to an
    // atom of the tile, so |coordinate| <= M = the extent of tile + halo (+ 1 A); converting costs <= 2^-23 M each, a
    // difference of two of them 2^-22 M + its own rounding, and r2 = dx^2 + dy^2 + dz^2 near the list radius R moves by
    // 2 (|dx| + |dy| + |dz|) times that (<= 2 sqrt(3) R) plus the roundings of its six operations.  Half again as margin.
    double M = 0;
    for (int c = 0; c < 3; c++) {
      double e = 1.0;
      for (int d = 0; d < 3; d++) e += fabs(ctx->box[3 * d + c]) * (double) (ctx->tile[d] + 2) / (double) std::max(1, ctx->gdim[d]);
      M = std::max(M, e);
    }
    const double u = 1.0 / 8388608.0, R = sqrt(ctx->cellsz), ddx = 2 * M * u + R * u;
    a.nbl_eps = (float) (1.5 * (2 * 1.7320508 * R * ddx + 6 * 0.5 * u * ctx->cellsz));
  }
  return a;
}

static size_t tile_lds_bytes(const imdgpu_ctx *ctx, int which /*0 build, 1 pass1, 3 pass3*/);
static size_t tile_lds_bytes_for(const imdgpu_ctx *ctx, int slot_cap, int which)
{
  imdgpu_ctx tmp;                       // only the sizing fields are read
  tmp.ntypes = ctx->ntypes; tmp.slot_cap = slot_cap; tmp.tile_rec = ctx->tile_rec;
  tmp.rec_lds = ctx->rec_lds; tmp.ncompP = ctx->ncompP; tmp.ncompR = ctx->ncompR;
  tmp.pair.present = ctx->pair.present; tmp.pair.par = ctx->pair.par;
  tmp.rho.present = ctx->rho.present; tmp.rho.par = ctx->rho.par;
  return tile_lds_bytes(&tmp, which);
}
static size_t tile_lds_bytes(const imdgpu_ctx *ctx, int which /*0 build, 1 pass1, 3 pass3*/)
{
  size_t ints = (ctx->ntypes == 1 ? 0 : (size_t) ctx->slot_cap * sizeof(int));      // per-slot types (several types only)
  size_t tp = ctx->pair.present ? (size_t) ctx->pair.par.nrows * ctx->pair.par.ncols : 0;
  size_t tr = ctx->rho.present ? (size_t) ctx->rho.par.nrows * ctx->rho.par.ncols : 0;
  if (which == 0)                       // list build: planes, halo-cell table, per-slot type bytes (several types)
    return align8((size_t) 3 * ctx->slot_cap * 8 + (size_t) (2 * NH_MAX + 1) * sizeof(int) + (ctx->ntypes == 1 ? 0 : (size_t) ctx->slot_cap));
  if (ctx->tile_rec) {                  // record kernels: planes, scratch, per-column parameters, the tables this list keeps in the LDS
    const size_t tabs = which == 1 ? ((ctx->rec_lds & 1) ? ctx->ncompP : 0) + (size_t) ((ctx->rec_lds & 2) ? ctx->ncompR : 0)
                                   : (size_t) ((ctx->rec_lds & 4) ? ctx->ncompR : 0);
    return align8(((which == 1 ? 3 : 4) * (size_t) ctx->slot_cap + 64 + tabs) * 8 + (which == 1 ? 2 : 1) * IMDGPU_MAXCOL * sizeof(ColPar));
  }
  if (which == 1) return align8((tp + tr + 6 + 2 + 3 * (size_t) ctx->slot_cap + 64) * 8 + ints);
  return align8((tr + 4 + 2 + 4 * (size_t) ctx->slot_cap + 64) * 8 + ints);
}

// Multi-rank set-ups call this between the ghost layout and make_nblist: it may re-sort the local atoms
// (new tile shape), after which the caller has to rebuild its halo index lists.  *changed tells.
extern "C" int imdgpu_autotile(imdgpu_ctx *ctx, int *changed)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (changed) *changed = 0;
  if (ctx->mode != 1 || ctx->autotiled || cfg_env("IMDGPU_TILE")) { ctx->autotiled = 1; return 0; }
  const int before[3] = {ctx->tile[0], ctx->tile[1], ctx->tile[2]};
  if (autotile(ctx)) return 1;
  if (changed) *changed = before[0] != ctx->tile[0] || before[1] != ctx->tile[1] || before[2] != ctx->tile[2];
  return 0;
}

extern "C" int imdgpu_make_nblist(imdgpu_ctx *ctx)
{
  imdgpu_dist_cancel_ahead(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal;                     // 0 is legal: a rank whose domain is vacuum (laser-ablation slabs split along x)
  if (!ctx->have_atoms) return imdgpu_fail(ctx, "no atoms uploaded");
  if (ctx->cellsz <= 0) return imdgpu_fail(ctx, "cellsz not set");
  Geo g = make_geo(ctx);
  if (imdgpu_fill_ghosts(ctx)) return 1;
  ctx->list_mode = ctx->mode;
  // the tile kernels keep the potential tables in LDS next to the staged positions: tables that leave no room for
  // even a one-cell tile (several atom types) select the global-gather kernels (reported by get_info("list_mode"))
  ctx->tile_rec = 0; ctx->rec_lds = 0;
  if (ctx->list_mode == 1 && ctx->d_recP && ctx->d_recR && ctx->d_rec3 && ctx->tile_rec_opt) {
    // several atom types with pair_pot and rho_h_tab: the record kernels.  Which tables they keep in the LDS is settled here, before
    // the tile shape (whose LDS budget it enters): what fits beside the planes of a FAST_CAP tile -- in pass 3 rho_h_tab beside four
    // planes, in pass 1 both tables beside three, else the one that fits (pair_pot first)
    ctx->tile_rec = 1;
    const size_t free1 = LDS_LIMIT - tile_lds_bytes_for(ctx, FAST_CAP, 1), free3 = LDS_LIMIT - tile_lds_bytes_for(ctx, FAST_CAP, 3);
    const size_t bP = (size_t) ctx->ncompP * 8, bR = (size_t) ctx->ncompR * 8;
    int m = 0;
    if (bR <= free3) m |= 4;
    if (bP + bR <= free1) m |= 3; else if (bP <= free1) m |= 1; else if (bR <= free1) m |= 2;
    if (ctx->rec_lds_opt >= 0) {          // forced: keep what can fit at all (beside a 1024-slot tile)
      m = ctx->rec_lds_opt & 7;
      ctx->rec_lds = m;
      if (tile_lds_bytes_for(ctx, 1024, 1) > LDS_LIMIT) m &= ~2;
      ctx->rec_lds = m;
      if (tile_lds_bytes_for(ctx, 1024, 1) > LDS_LIMIT) m &= ~1;
      ctx->rec_lds = m;
      if (tile_lds_bytes_for(ctx, 1024, 3) > LDS_LIMIT) m &= ~4;
    }
    ctx->rec_lds = m;
  } else if (ctx->list_mode == 1 && std::max(tile_lds_bytes_for(ctx, 1024, 1), tile_lds_bytes_for(ctx, 1024, 3)) > LDS_LIMIT)
    ctx->list_mode = 0;                  // tables that leave no room for a tile and no record form: the global-gather kernels
  if (ctx->list_mode == 1 && !ctx->autotiled && !cfg_env("IMDGPU_TILE") && !ctx->external_halo) {
    if (autotile(ctx)) return 1;
    g = make_geo(ctx);
    if (imdgpu_fill_ghosts(ctx)) return 1;
  }
  for (int retile = 0; ctx->list_mode == 1 && retile < 8; retile++) {
    // size the tiles: largest own-atom count (thread count, list row length) and largest staged count (LDS)
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_MAXOWN, 0, 2 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_tile_stats, dim3(nblk(ctx->ntiles, 64)), dim3(64), 0, ctx->stream, ctx->ntiles, g, ctx->cell_cnt,
                       ctx->istart, ctx->d_counters);
    HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int tile_cap = (std::max(ctx->h_counters[CNT_MAXOWN], 1) + 63) / 64 * 64;
    {   // uneven occupancy (vacuum, slabs): cut the persistent blocks' tile ranges by atoms, not by tiles
      static const char *fb = exp_env("IMDGPU_TILE_BALANCE");
      ctx->tile_balance = fb ? atoi(fb) : ((double) ctx->h_counters[CNT_MAXOWN] * ctx->ntiles > 1.25 * (double) std::max(n, 1));
    }
    if (ctx->tile_balance) {
      if (ctx->wprefix_cap < (size_t) ctx->ntiles + 1) { if (dev_alloc(ctx, &ctx->d_wprefix, (size_t) ctx->ntiles + 1)) return 1; ctx->wprefix_cap = (size_t) ctx->ntiles + 1; }
      hipLaunchKernelGGL(k_tile_wprefix, dim3(1), dim3(1024), 0, ctx->stream, ctx->ntiles, g.CPT, ctx->istart, ctx->d_wprefix);
    }
    ctx->slot_cap = (ctx->h_counters[CNT_MAXHALO] + 1 + 63) / 64 * 64;      // + far-away dummy slot
    // ... rounded up to the stride the force kernels' neighbour loops are also compiled for, if the LDS holds the wider planes
    if (ctx->slot_cap < FAST_CAP && std::max(tile_lds_bytes_for(ctx, FAST_CAP, 1), tile_lds_bytes_for(ctx, FAST_CAP, 3)) <= LDS_LIMIT)
      ctx->slot_cap = FAST_CAP;
    ctx->max_halo = ctx->h_counters[CNT_MAXHALO];
    if (tile_cap != ctx->tile_cap) { ctx->tile_cap = tile_cap; ctx->nbr16_cap = 0; }
    ctx->tile_threads = std::min(TILE_MAXTHREADS, ctx->tile_cap);
    ctx->lds_need[1] = tile_lds_bytes(ctx, 1); ctx->lds_need[3] = tile_lds_bytes(ctx, 3);
    size_t need = std::max(ctx->lds_need[1], ctx->lds_need[3]);
    if (need <= LDS_LIMIT && ctx->slot_cap <= 65535 && (g.bx + 2) * (g.by + 2) * (g.bz + 2) <= NH_MAX) break;
    if (ctx->tile[0] * ctx->tile[1] * ctx->tile[2] > 1 && !ctx->external_halo) {   // tile + halo does not fit the 160 KB LDS: smaller tiles
      if (shrink_tiles(ctx)) return 1;
      g = make_geo(ctx);
      if (imdgpu_fill_ghosts(ctx)) return 1;
    } else ctx->list_mode = 0;                                 // even one cell + halo too big: global-gather kernels
  }
  size_t stride = (size_t) ctx->cap_local;
  if (ctx->W == 0) ctx->W = 96;
  for (int attempt = 0; attempt < 3; attempt++) {
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_MAXNBR, 0, sizeof(int), ctx->stream));
    if (ctx->list_mode == 1) {
      size_t need = (size_t) ctx->ntiles * (ctx->W / 4) * ctx->tile_cap * 4;           // in ushorts
      if (ctx->nbr16_cap < need) {
        if (dev_alloc(ctx, &ctx->nbr16, need)) return 1;
        ctx->nbr16_cap = need;
      }
      const size_t need_src = (size_t) ctx->ntiles * ctx->slot_cap, need_own = (size_t) ctx->ntiles * ctx->tile_cap * (ctx->ntypes > 1 ? 2 : 1);
      if (ctx->tsrc_cap < need_src) { if (dev_alloc(ctx, &ctx->tsrc, need_src)) return 1; ctx->tsrc_cap = need_src; }
      if (ctx->ownw_cap < need_own) { if (dev_alloc(ctx, &ctx->ownw, need_own)) return 1; ctx->ownw_cap = need_own; }
      if (ctx->tinfo_cap < (size_t) ctx->ntiles * 4) { if (dev_alloc(ctx, &ctx->tinfo, (size_t) ctx->ntiles * 4)) return 1; ctx->tinfo_cap = (size_t) ctx->ntiles * 4; }
      TileArgs a = make_tile_args(ctx);
      size_t lds = tile_lds_bytes(ctx, 0);
      HIPCHK(hipFuncSetAttribute((const void *) k_nbl_build_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      ProfScope ps(ctx, PROF_NBLIST);
      // the build needs only the position planes (about half the LDS of a force pass): two workgroups per CU
      const int bgrid = 2 * lds <= LDS_LIMIT ? std::min(ctx->ntiles, 2 * tile_grid(ctx)) : tile_grid(ctx);
      static const int nbl32 = exp_env_int("IMDGPU_NBL32", 1);     // 0: the all-double build kernel
      if (nbl32) {
#define LB(TY)                                                                                                       \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_nbl_build_tile32<TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
    hipLaunchKernelGGL(k_nbl_build_tile32<TY>, dim3(bgrid), dim3(ctx->tile_threads), lds, ctx->stream, a);           \
  } while (0)
        if (ctx->ntypes == 1) LB(1); else if (ctx->ntypes == 2) LB(2); else LB(0);
#undef LB
      } else
      hipLaunchKernelGGL(k_nbl_build_tile, dim3(bgrid), dim3(ctx->tile_threads), lds, ctx->stream, a);
    } else {
      if (ctx->nbr_cap < (size_t) ctx->W * stride) {
        if (dev_alloc(ctx, &ctx->nbr, (size_t) ctx->W * stride)) return 1;
        ctx->nbr_cap = (size_t) ctx->W * stride;
      }
      ctx->stride = stride;
      ProfScope ps(ctx, PROF_NBLIST);
      hipLaunchKernelGGL(k_nbl_build, dim3(nblk(n, 128)), dim3(128), 0, ctx->stream, n, ctx->pos, ctx->acell,
                         ctx->cell_start, ctx->cell_cnt, g, ctx->cellsz, ctx->nbr, stride, ctx->W, ctx->nbr_cnt, ctx->ref,
                         ctx->cap_local, ctx->d_counters);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int mx = ctx->h_counters[CNT_MAXNBR];
    if (mx <= ctx->W) {
      if (ctx->tl_on && ctx->list_mode == 1) {               // diagnostics: how the own atoms are spread over the tiles
        std::vector<int> is((size_t) ctx->ntiles * ctx->CPT + 1);
        (void) hipMemcpy(is.data(), ctx->istart, is.size() * sizeof(int), hipMemcpyDeviceToHost);
        int h[5] = {0, 0, 0, 0, 0};
        for (int t = 0; t < ctx->ntiles; t++) { const int o = is[(size_t) (t + 1) * ctx->CPT] - is[(size_t) t * ctx->CPT]; h[o == 0 ? 0 : o < 16 ? 1 : o < 128 ? 2 : o < 512 ? 3 : 4]++; }
        fprintf(stderr, "[imdgpu timeline] tiles by own atoms: empty %d, 1-15: %d, 16-127: %d, 128-511: %d, >=512: %d\n", h[0], h[1], h[2], h[3], h[4]);
      }
      if (ctx->tl_on) fprintf(stderr, "[imdgpu timeline] list build %d: mode %d tile %d %d %d ntiles %d tile_cap %d slot_cap %d W %d max_nbr %d balance %d nlocal %d nghost %d\n",
                                 ctx->nbl_count + 1, ctx->list_mode, ctx->tile[0], ctx->tile[1], ctx->tile[2], ctx->ntiles, ctx->tile_cap, ctx->slot_cap, ctx->W, mx, ctx->tile_balance, ctx->nlocal, ctx->nghost);
      ctx->have_valid_nbl = 1;
      ctx->nbl_count++;
      HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_INVALID, 0, sizeof(int), ctx->stream));
      return 0;
    }
    ctx->W = ((int) (mx * 1.15) + 8 + 3) / 4 * 4;     // "neighbor table full": grow and rebuild
    ctx->nbr_cap = 0; ctx->nbr16_cap = 0;
  }
  return imdgpu_fail(ctx, "neighbor table full");
}

extern "C" int imdgpu_nblist_stats(imdgpu_ctx *ctx, long long *entries, int *maxn)
{
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<int> c(ctx->nlocal);
  HIPCHK(hipMemcpy(c.data(), ctx->nbr_cnt, ctx->nlocal * sizeof(int), hipMemcpyDeviceToHost));
  long long s = 0; int m = 0;
  for (int v : c) { s += v; if (v > m) m = v; }
  if (entries) *entries = s;
  if (maxn) *maxn = m;
  return 0;
}

extern "C" int imdgpu_check_nblist(imdgpu_ctx *ctx, int *valid)
{
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal, nb = nblk(n, 256);
  double h = 0.5 * ctx->nbl_margin;
  {
    ProfScope ps(ctx, PROF_CHECK);
    hipLaunchKernelGGL(k_check_only, dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, ctx->pos, ctx->ref, ctx->partials);
    hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, h * h,
                       ctx->d_counters, 0);
  }
  HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->h_counters[CNT_INVALID]) { ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1; }
  if (valid) *valid = ctx->have_valid_nbl;
  return 0;
}

// launches the three passes; assumes ghosts/list are current
// phase 0: all three passes; 1 / 2 / 3: only that pass (3 includes the final reduction) -- the split form is
// used when an external halo exchange (multi-rank) has to run between the passes
// the embedding pass is folded into pass 1 whenever the merged-table tile kernel runs (one atom type, pair and rho on the
// same grid); IMDGPU_FUSE_EMB=0 keeps the separate k_pass2 (experiments)
static bool fuse_emb(const imdgpu_ctx *ctx);
// LDS bytes of k_pass3_tile_coef and the first coefficient row it holds; 0 = not applicable (several types, tables on
// different grids, the option off, or more than a quarter of the rows would have to stay out)
// pass 1 is k_pass1_tile_merged: one atom type, pair_pot and rho_h_tab both present, tile kernels -- on a shared grid (d_merged) or
// each table on its own (the DG form).  That kernel folds the embedding pass in and reports the lowest row of rho_h_tab a pair fell into
static bool p1_onetype(const imdgpu_ctx *ctx)
{
  return ctx->list_mode == 1 && ctx->ntypes == 1 && ctx->pair.present && ctx->rho.present;
}
static size_t p3_coef_lds(const imdgpu_ctx *ctx, int &klo)
{
  static const int env = exp_env_int("IMDGPU_P3_COEF", 1);
  if (!env || !ctx->p3_coef || !p1_onetype(ctx) || !ctx->d_coef3) return 0;
  if (imdgpu_ttm_chain_needs_lds(ctx)) return 0;      // this kernel leaves no LDS to anything beside it
  const size_t fixed = align8((4 * (size_t) ctx->slot_cap + 64 + 2) * 8);
  if (fixed + 64 * 16 > LDS_LIMIT) return 0;
  const int fit = (int) ((LDS_LIMIT - fixed) / 16);
  klo = ctx->ncoef > fit ? ctx->ncoef - fit : 0;
  if (klo > ctx->ncoef / 4) return 0;
  if (ctx->p3_klo > klo) klo = ctx->p3_klo < ctx->ncoef - 1 ? ctx->p3_klo : ctx->ncoef - 1;
  return fixed + (size_t) (ctx->ncoef - klo) * 16;
}
// does the EAM force pass of this context run in its coefficient form (128 registers)?  The lattice chain forked beside it
// picks its register budget by this
int imdgpu_p3_coef_active(const imdgpu_ctx *ctx) { int klo; return p3_coef_lds(ctx, klo) != 0; }
static bool fuse_emb(const imdgpu_ctx *ctx)
{
  static const int on = exp_env_int("IMDGPU_FUSE_EMB", 1);
  return on && (p1_onetype(ctx) || (ctx->list_mode == 1 && ctx->tile_rec));
}

static int launch_forces(imdgpu_ctx *ctx, int phase = 0)
{
  const int n = ctx->nlocal;
  const bool PAIR = ctx->pair.present, EAM = ctx->rho.present, NT1 = ctx->ntypes == 1, TT = ctx->count_ttm != 0;
  if (!PAIR && !EAM) return imdgpu_fail(ctx, "no potential tables uploaded");
  if (TT && !PAIR) return imdgpu_fail(ctx, "TTM neighbour count needs pair_pot.end[0]");
  ForceArgs a;
  a.nlocal = n; a.ntypes = ctx->ntypes; a.cap = ctx->cap_local;
  a.pos = ctx->pos; a.type = ctx->type; a.nbr = ctx->nbr; a.nbr_cnt = ctx->nbr_cnt; a.stride = ctx->stride;
  a.pair = ctx->pair.par; a.rho = ctx->rho.par;
  a.f = ctx->f; a.epot = ctx->epot; a.rhoh = ctx->rhoh; a.numneighs = ctx->numneighs;
  a.partials = ctx->partials; a.counters = ctx->d_counters; a.guard = ctx->pass1_guard;
  const bool TILE = ctx->list_mode == 1;
  const int B = TILE ? ctx->tile_threads : 512, nb = TILE ? tile_grid(ctx) : nblk(n, B);
  if (nb > ctx->npart_blocks) return imdgpu_fail(ctx, "internal: partials too small");
  TileArgs ta;
  if (TILE) { ta = make_tile_args(ctx); ta.guard = ctx->pass1_guard; }
  static const int fastcap_env = exp_env_int("IMDGPU_FAST_CAP", 1);
  const bool fastcap = fastcap_env && TILE && ctx->slot_cap == FAST_CAP;       // the neighbour loops compiled for this plane stride
  size_t lds1 = ((PAIR ? a.pair.nrows * a.pair.ncols : 0) + (EAM ? a.rho.nrows * a.rho.ncols : 0) + 64) * sizeof(double);
  if (phase == 0 || phase == 1) {
    ctx->dyn_nb[0] = ctx->dyn_nb[1] = 0;
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_IS_SHORT, 0, sizeof(int), ctx->stream));
    ProfScope ps(ctx, PROF_PASS1);
#define L1(P, E, T, N)                                                                                          \
  do {                                                                                                          \
    if (TILE) {                                                                                                 \
      size_t l = tile_lds_bytes(ctx, 1);                                                                        \
      HIPCHK(hipFuncSetAttribute((const void *) k_pass1_tile<P, E, T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
      hipLaunchKernelGGL((k_pass1_tile<P, E, T, N>), dim3(nb), dim3(B), l, ctx->stream, ta);                     \
    } else if (lds1 > TABLE_LDS_LIMIT) hipLaunchKernelGGL((k_pass1_gt<P, E, T>), dim3(nb), dim3(B), 0, ctx->stream, a); \
    else hipLaunchKernelGGL((k_pass1<P, E, T, N>), dim3(nb), dim3(B), lds1, ctx->stream, a);                    \
  } while (0)
#define L1T(P, E, T) do { if (NT1) L1(P, E, T, true); else L1(P, E, T, false); } while (0)
#define L1E(P, E) do { if (TT) L1T(P, E, true); else L1T(P, E, false); } while (0)
    if (TILE && ctx->tile_rec) {
      size_t l = tile_lds_bytes(ctx, 1);
#define LR(T, C, PL, RL)                                                                                             \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_pass1_tile_rec<T, C, PL, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
    hipLaunchKernelGGL((k_pass1_tile_rec<T, C, PL, RL>), dim3(nb), dim3(B), l, ctx->stream, ta);                     \
  } while (0)
#define LRC(T, PL, RL) do { if (fastcap) LR(T, FAST_CAP, PL, RL); else LR(T, 0, PL, RL); } while (0)
#define LRT(PL, RL) do { if (TT) LRC(true, PL, RL); else LRC(false, PL, RL); } while (0)
      switch (ctx->rec_lds & 3) {
        case 3: LRT(true, true); break;
        case 1: LRT(true, false); break;
        case 2: LRT(false, true); break;
        default: LRT(false, false);
      }
#undef LRT
#undef LRC
#undef LR
    } else if (p1_onetype(ctx)) {
      size_t l = tile_lds_bytes(ctx, 1);
      const bool dg = ctx->d_merged == nullptr;         // each table on its own grid
#define LMCD(T, F, C, D)                                                                                             \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_pass1_tile_merged<T, F, C, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
    hipLaunchKernelGGL((k_pass1_tile_merged<T, F, C, false, D>), dim3(nb), dim3(B), l, ctx->stream, ta);             \
  } while (0)
#define LMC(T, F, C) do { if (dg) LMCD(T, F, C, true); else LMCD(T, F, C, false); } while (0)
#define LM(T, F) do { if (fastcap) LMC(T, F, FAST_CAP); else LMC(T, F, 0); } while (0)
      if (TT) { if (fuse_emb(ctx)) LM(true, true); else LM(true, false); }
      else { if (fuse_emb(ctx)) LM(false, true); else LM(false, false); }
#undef LM
#undef LMC
#undef LMCD
    } else if (PAIR && EAM) L1E(true, true); else if (PAIR) L1E(true, false); else L1E(false, true);
#undef L1E
#undef L1T
#undef L1
  }
  imdgpu_tl_mark(ctx, "pass1");
  // fork point of the TTM lattice chain.  Forking later (after the embedding pass) was tried: update_fd then starts
  // under the EAM force pass, whose waves leave it one SIMD per CU, takes 1.5 ms instead of 0.4 and the chain
  // outlasts the pass; beside the short, register-light embedding pass it runs at streaming speed.
  if ((phase == 0 || phase == 1) && ctx->after_pass1) { if (ctx->after_pass1(ctx, ctx->after_pass1_arg)) return 1; }
  const bool FUSED = fuse_emb(ctx);                    // the embedding pass ran inside pass 1: its partials sit in pass 1's blocks
  const int nb2 = EAM ? (FUSED ? nb : nblk(n, 256)) : 0;
  if (EAM && (phase == 0 || phase == 2)) {
    if (!FUSED) {
      ProfScope ps(ctx, PROF_PASS2);
      hipLaunchKernelGGL(k_pass2, dim3(nb2), dim3(256), 0, ctx->stream, n, ctx->ntypes, ctx->pos, ctx->type, ctx->rhoh,
                         ctx->epot, ctx->emb.par, ctx->partials);
    }
    if (imdgpu_fill_ghosts(ctx)) return 1;            // dF of the images  [send_cells(copy_dF), :786]
    imdgpu_tl_mark(ctx, "pass2");
  }
  if (EAM && (phase == 0 || phase == 3)) {
    // without the embedding pass between them, update_fd of the forked TTM chain would start beside the persistent EAM
    // force kernel and crawl (see above): the force pass waits for it (0.15 ms alone), the rest of the chain overlaps
    if (ctx->ufd_pending) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_ufd, 0)); ctx->ufd_pending = 0; }
    size_t lds3 = ((size_t) a.rho.nrows * a.rho.ncols + 64) * sizeof(double);
    ProfScope ps(ctx, PROF_PASS3);
    if (TILE && ctx->tile_rec) {
      size_t l = tile_lds_bytes(ctx, 3);
#define LR3(C, RL, K)                                                                                                \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) K<C, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l));       \
    hipLaunchKernelGGL((K<C, RL>), dim3(nb), dim3(B), l, ctx->stream, ta);                                           \
  } while (0)
      // beside the lattice chain the leaner list walk (see k_pass3_tile)
#define LR3C(RL, RG) do { if (fastcap) LR3(FAST_CAP, RL, RG); else LR3(0, RL, RG); } while (0)
      if (ctx->rec_lds & 4) { if (ctx->ttm) LR3C(true, k_pass3_tile_rec_lean); else LR3C(true, k_pass3_tile_rec); }
      else { if (ctx->ttm) LR3C(false, k_pass3_tile_rec_lean); else LR3C(false, k_pass3_tile_rec); }
#undef LR3C
#undef LR3
    } else if (TILE) {
      size_t l = tile_lds_bytes(ctx, 3);
      // with a lattice attached its sub-step chain runs beside this pass and needs 80 registers on every SIMD: the
      // leaner list walk (142 VGPRs) then; the register ring (148) otherwise
#define L3K(N, R, C)                                                                                                 \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_pass3_tile<N, R, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
    hipLaunchKernelGGL((k_pass3_tile<N, R, C>), dim3(nb), dim3(B), l, ctx->stream, ta);                              \
  } while (0)
#define L3(N, R) do { if (fastcap) L3K(N, R, FAST_CAP); else L3K(N, R, 0); } while (0)
      // one type behind the merged pass 1: the coefficient form first, the three-row form behind it as its stand-in for
      // steps in which a pair fell below the rows the LDS holds (both look at CNT_KNEG and one of them returns at once)
      int klo = 0;
      const size_t lc = p3_coef_lds(ctx, klo);
      if (lc) {
        ta.klo = klo;
#define L3CK(R, C)                                                                                                   \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_pass3_tile_coef<R, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lc)); \
    hipLaunchKernelGGL((k_pass3_tile_coef<R, C>), dim3(nb), dim3(B), lc, ctx->stream, ta);                           \
  } while (0)
#define L3C(R) do { if (fastcap) L3CK(R, FAST_CAP); else L3CK(R, 0); } while (0)
        // 132 VGPRs with the register ring: three waves per SIMD still leave the 80 a wave of the lattice chain needs
        static const int ring = exp_env_int("IMDGPU_P3_RING", 4);
        if (ring == 0) L3C(0); else L3C(4);
#undef L3C
#undef L3CK
        ta.p3_fallback = 1;
      }
      if (ctx->ttm) { if (NT1) L3(true, 0); else L3(false, 0); }
      else { if (NT1) L3(true, 4); else L3(false, 4); }
#undef L3
#undef L3K
    } else if (lds3 > TABLE_LDS_LIMIT) hipLaunchKernelGGL(k_pass3_gt, dim3(nb), dim3(B), 0, ctx->stream, a);
    else if (NT1) hipLaunchKernelGGL((k_pass3<true>), dim3(nb), dim3(B), lds3, ctx->stream, a);
    else hipLaunchKernelGGL((k_pass3<false>), dim3(nb), dim3(B), lds3, ctx->stream, a);
  }
  if (phase == 0 || phase == 3) {
    ProfScope ps(ctx, PROF_REDUCE);
    const int nbd = ctx->dyn_nb[0] + ctx->dyn_nb[1];          // pass 1 ran in its two-launch form (imdgpu_launch_pass1_dyn): embedding fused
    hipLaunchKernelGGL(k_reduce_force_sums, dim3(1), dim3(1024), 0, ctx->stream, nbd ? nbd : nb, EAM ? (nbd ? nbd : nb2) : 0, nb, ctx->partials,
                       ctx->d_sums, ctx->d_counters);
  }
  imdgpu_tl_mark(ctx, "pass3");
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- pass 1 from a tile queue (k_pass1_tile_merged<.., DYN>) -----------------------------------------------------------------
int imdgpu_pass1_dyn_ready(const imdgpu_ctx *ctx)
{
  return ctx->halo_overlap > 0 && ctx->list_mode == 1 && ctx->pair.present && ctx->rho.present && ctx->ntypes == 1 && ctx->d_merged &&
         fuse_emb(ctx) && ctx->d_tile_map && ctx->dyn_ntiles == ctx->ntiles && !ctx->after_pass1;
}
int imdgpu_pass1_dyn_prepare(imdgpu_ctx *ctx, const int remote[3])
{
  ctx->dyn_ntiles = -1;
  if (ctx->halo_overlap <= 0 || ctx->list_mode != 1) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const int *nt = ctx->ntile;
  std::vector<int> map((size_t) ctx->ntiles);
  size_t nI = 0;
  for (int pass = 0; pass < 2; pass++)              // interior tiles in tile order (z fastest), then the boundary tiles
    for (int t = 0, w = (int) nI; t < ctx->ntiles; t++) {
      const int c[3] = {t / (nt[1] * nt[2]), (t / nt[2]) % nt[1], t % nt[2]};
      bool bnd = false;
      for (int d = 0; d < 3; d++) bnd = bnd || (remote[d] && (c[d] == 0 || c[d] == nt[d] - 1));
      if (pass == 0 && !bnd) map[nI++] = t;
      if (pass == 1 && bnd) map[(size_t) w++] = t;
    }
  if (ctx->tile_map_cap < map.size()) { if (dev_alloc(ctx, &ctx->d_tile_map, map.size())) return 1; ctx->tile_map_cap = map.size(); }
  if (!ctx->d_dyn) { if (dev_alloc(ctx, &ctx->d_dyn, 32)) return 1; HIPCHK(hipMemset(ctx->d_dyn, 0, 32 * sizeof(int))); }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(ctx->d_tile_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
  ctx->n_interior = (int) nI; ctx->dyn_ntiles = ctx->ntiles;
  return 0;
}
int imdgpu_launch_pass1_dyn(imdgpu_ctx *ctx, int part)
{
  if (!imdgpu_pass1_dyn_ready(ctx)) return imdgpu_fail(ctx, "internal: pass 1 has no tile queue");
  const bool TT = ctx->count_ttm != 0;
  static const int fastcap_env = exp_env_int("IMDGPU_FAST_CAP", 1);
  const bool fastcap = fastcap_env && ctx->slot_cap == FAST_CAP;
  const int B = ctx->tile_threads, full = tile_grid(ctx);
  // the first launch leaves halo_overlap CUs alone: the transport's kernels (RCCL's, the unpack) need somewhere to run
  const int nb = part == 0 ? std::max(1, full - ctx->halo_overlap) : full;
  if (part == 0) {
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_IS_SHORT, 0, sizeof(int), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_dyn, 0, 16 * sizeof(int), ctx->stream));
    ctx->dyn_seq++;
    ctx->dyn_nb[0] = nb; ctx->dyn_nb[1] = 0;
  } else ctx->dyn_nb[1] = nb;
  if (ctx->dyn_nb[0] + ctx->dyn_nb[1] > ctx->npart_blocks) return imdgpu_fail(ctx, "internal: partials too small");
  TileArgs ta = make_tile_args(ctx);
  ta.guard = ctx->pass1_guard;
  ta.tile_map = ctx->d_tile_map; ta.n_interior = ctx->n_interior; ta.dyn = ctx->d_dyn;
  ta.dyn_last = part; ta.part_off = part == 0 ? 0 : ctx->dyn_nb[0];
  const size_t l = tile_lds_bytes(ctx, 1);
  {
    ProfScope ps(ctx, part == 0 ? PROF_PASS1 : PROF_PASS1B);
#define LD(T, C)                                                                                                     \
  do {                                                                                                               \
    HIPCHK(hipFuncSetAttribute((const void *) k_pass1_tile_merged<T, true, C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
    hipLaunchKernelGGL((k_pass1_tile_merged<T, true, C, true>), dim3(nb), dim3(B), l, ctx->stream, ta);              \
  } while (0)
    if (TT) { if (fastcap) LD(true, FAST_CAP); else LD(true, 0); }
    else { if (fastcap) LD(false, FAST_CAP); else LD(false, 0); }
#undef LD
  }
  imdgpu_tl_mark(ctx, part == 0 ? "pass1A" : "pass1B");
  HIPCHK(hipGetLastError());
  return 0;
}

static int zero_eam_partials(imdgpu_ctx *ctx)
{
  // slots 2 and 3 are only written by the EAM passes
  HIPCHK(hipMemsetAsync(ctx->partials, 0, (size_t) ctx->npart_blocks * IMDGPU_NPART * sizeof(double), ctx->stream));
  return 0;
}

extern "C" int imdgpu_calc_forces(imdgpu_ctx *ctx, int steps, double sums[8], int *is_short)
{
  (void) steps;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->have_valid_nbl) {
    if (ctx->need_fix_cells && imdgpu_fix_cells(ctx)) return 1;    // :300-308 (after an upload: arrangement as given)
    if (imdgpu_make_nblist(ctx)) return 1;                         // :315
  } else if (imdgpu_fill_ghosts(ctx)) return 1;                    // :311
  if (!ctx->rho.present && zero_eam_partials(ctx)) return 1;
  if (launch_forces(ctx)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->h_sums, ctx->d_sums, SUM_N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (sums) for (int q = 0; q < 8; q++) sums[q] = ctx->h_sums[q];
  if (is_short) *is_short = ctx->h_counters[CNT_IS_SHORT];
  return 0;
}

extern "C" int imdgpu_move_atoms_nve(imdgpu_ctx *ctx, double *ekin)
{
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal, nb = nblk(n, 256);
  double h = 0.5 * ctx->nbl_margin;
  {
    ProfScope ps(ctx, PROF_INTEGRATE);
    hipLaunchKernelGGL((k_move_nve<false>), dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, ctx->pos, ctx->type,
                       ctx->p, ctx->f, ctx->mass, ctx->restr_identity ? nullptr : ctx->d_restr, ctx->ntypes, ctx->timestep, ctx->ref, ctx->partials);
    hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, h * h + 1e300,
                       ctx->d_counters, 1);
  }
  HIPCHK(hipMemcpyAsync(ctx->h_sums, ctx->d_sums, SUM_N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ekin) *ekin = ctx->h_sums[SUM_EKIN];
  return 0;
}

static int fork_ttm(imdgpu_ctx *ctx, void *)
{
  static const int serial = exp_env_int("IMDGPU_TTM_SERIAL", 0);   // experiment: calc_ttm alone on the chip
  if (serial) {
    if (ctx->ttm1d ? imdgpu_ttm1d_calc(ctx, ctx->ttm_steps) : imdgpu_calc_ttm(ctx, ctx->ttm_steps)) return 1;
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream));
    return 0;
  }
  HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  hipStream_t main = ctx->stream;
  ctx->stream = ctx->stream2;
  int rc = ctx->ttm1d ? imdgpu_ttm1d_calc(ctx, ctx->ttm_steps) : imdgpu_calc_ttm(ctx, ctx->ttm_steps);
  ctx->stream = main;
  if (rc) return 1;
  HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
  return 0;
}

int imdgpu_ensure_capacity(imdgpu_ctx *ctx, int nlocal, int ntot) { return ensure_atom_capacity(ctx, nlocal, ntot); }
int imdgpu_launch_forces_phase(imdgpu_ctx *ctx, int phase)
{
  if (phase == 1 && !ctx->rho.present && zero_eam_partials(ctx)) return 1;
  return launch_forces(ctx, phase);
}
int imdgpu_launch_integrate(imdgpu_ctx *ctx)
{
  if (ctx->ttm1d) return imdgpu_ttm1d_move_internal(ctx);
  if (ctx->ttm) return imdgpu_ttm_move_internal(ctx);
  const int n = ctx->nlocal, nb = nblk(n, 256);
  double h = 0.5 * ctx->nbl_margin;
  ProfScope ps(ctx, PROF_INTEGRATE);
  hipLaunchKernelGGL((k_move_nve<true>), dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, ctx->pos, ctx->type,
                     ctx->p, ctx->f, ctx->mass, ctx->restr_identity ? nullptr : ctx->d_restr, ctx->ntypes, ctx->timestep, ctx->ref, ctx->partials);
  hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, h * h,
                     ctx->d_counters, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

// The MD step in two halves for hosts that keep their own main loop (level-2 integration, imd_shim.c): what the
// reference's loop does between calc_forces and check_nblist (imd_main_3d.c:398-812) happens between the two calls.
//   begin: list maintenance (fix_cells + make_nblist when the list is stale, else the ghost refresh), the force passes,
//          calc_ttm forked beside them -- everything enqueued, nothing waited for;
//   end:   move_atoms_{nve,ttm} + check_nblist, then the one host round trip of a step (is the list still valid?).
extern "C" int imdgpu_md_step_begin(imdgpu_ctx *ctx, int steps)
{
  HIPCHK(hipSetDevice(ctx->device));
  imdgpu_tl_mark(ctx, "begin");
  if (!ctx->have_valid_nbl) {
    static const bool rt = cfg_env("IMDGPU_REBUILD_TIMING") != nullptr;     // diagnostics: host wall clock of a rebuild's two halves
    const auto t0 = std::chrono::steady_clock::now();
    if (ctx->need_fix_cells && imdgpu_fix_cells(ctx)) return 1;
    const auto t1 = std::chrono::steady_clock::now();
    if (imdgpu_make_nblist(ctx)) return 1;
    if (rt) {
      const auto t2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[imdgpu rebuild] step %d: fix_cells %.3f ms, make_nblist %.3f ms (host wall clock, synchronisations included)\n", steps,
              std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
  } else if (imdgpu_fill_ghosts(ctx)) return 1;
  imdgpu_tl_mark(ctx, "list");
  if (!ctx->rho.present && zero_eam_partials(ctx)) return 1;
  if (ctx->ttm || ctx->ttm1d) {
    // calc_ttm needs only pass 1's neighbour counts and the momenta: it runs on the second stream,
    // concurrently with the embedding pass and the EAM force pass; move_atoms_ttm joins the two
    ctx->ttm_steps = steps;
    ctx->after_pass1 = fork_ttm; ctx->after_pass1_arg = nullptr;
    int rc = launch_forces(ctx);
    ctx->after_pass1 = nullptr;
    if (rc) return 1;
  } else if (launch_forces(ctx)) return 1;
  ctx->step_open = 1;
  return 0;
}
extern "C" int imdgpu_md_step_end(imdgpu_ctx *ctx, int steps)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->step_open) return imdgpu_fail(ctx, "md_step_end without md_step_begin");
  ctx->step_open = 0;
  if (ctx->ttm1d) {
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    if (imdgpu_ttm1d_move_internal(ctx)) return 1;
  } else if (ctx->ttm) {
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    if (imdgpu_ttm_move_internal(ctx)) return 1;
  } else {
    const int n = ctx->nlocal, nb = nblk(n, 256);
    double h = 0.5 * ctx->nbl_margin;
    ProfScope ps(ctx, PROF_INTEGRATE);
    hipLaunchKernelGGL((k_move_nve<true>), dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, ctx->pos, ctx->type,
                       ctx->p, ctx->f, ctx->mass, ctx->restr_identity ? nullptr : ctx->d_restr, ctx->ntypes, ctx->timestep, ctx->ref, ctx->partials);
    hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, h * h,
                       ctx->d_counters, 1);
  }
  // the only per-step host round trip: one int that says whether the list survived
  HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  imdgpu_tl_mark(ctx, "move");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  imdgpu_tl_flush(ctx, steps);
  if (ctx->ttm) {
    // the persistent sub-step kernel of this step stepped aside (status 1: nothing of the lattice touched, the integrator above
    // returned at once on the same word): its sub-steps have been issued through the launch chain; now the integrator again
    const int r = imdgpu_ttm_persist_settle(ctx);
    if (r < 0) return 1;
    if (r == 1) {
      if (imdgpu_ttm_move_internal(ctx)) return 1;
      HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  if (ctx->h_counters[CNT_INVALID]) { ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1; }
  if (ctx->h_counters[CNT_TTM_TIMEOUT]) {
    // reported once: the context stays usable -- e.g. with option ttm_persist = 0 and a lattice uploaded again
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_TTM_TIMEOUT, 0, sizeof(int), ctx->stream));
    return imdgpu_fail(ctx, "TTM lattice: a sub-step hand-off timed out inside the loop; the lattice of this step is not valid (option ttm_persist = 0 / IMDGPU_TTM_PERSIST=0 disables the persistent sub-step kernel)");
  }
  if (ctx->h_counters[CNT_TMM_SHORT]) return imdgpu_fail(ctx, "Sample is too short for TMM. Consider reducing tmm_threshold.");      // imd_tmm.c:362: the reference aborts
  return 0;
}
extern "C" int imdgpu_md_step(imdgpu_ctx *ctx, int steps)
{
  return imdgpu_md_step_begin(ctx, steps) || imdgpu_md_step_end(ctx, steps);
}

// ---- the reference's cell-pair entry points on host arrays (do_forces, do_forces_eam2, do_embedding_energy) ------------
// For hosts that keep their own cell-pair loop (imd_main_risc_3d.c:31-182) and call the kernels of imd_forces.c /
// imd_forces_eam2.c per pair of cells: the same contract -- results are ADDED to the arrays passed in, each pair is visited
// once, the partner gets the opposite force -- on a pair of small host arrays.  One workgroup per call (cells hold a few
// dozen atoms); a compatibility path, three orders of magnitude away from the list kernels' throughput.
struct CellPairArgs {
  int np, nq, same, ntypes;
  const double *pos_p, *pos_q, *dF_p, *dF_q;     // xyz interleaved
  const int *typ_p, *typ_q;
  double sx, sy, sz;
  double *f_p, *f_q, *e_p, *e_q, *rho_p, *rho_q;
  int *nn_p, *nn_q;
  double *sums;                                  // [0] Epot, [1] virial
  int *is_short;
  TabPar pair, rho;
};
// which = 1: do_forces (imd_forces.c:33-326: pair term, host electron density, TTM neighbour count);
// which = 2: do_forces_eam2 (imd_forces_eam2.c:236-369: the EAM force from dF = EAM_DF of both atoms)
template <int WHICH>
__global__ void k_cellpair(CellPairArgs a)
{
  const int inc = a.ntypes * a.ntypes;
  double epot = 0, vir = 0;
  int shortd = 0;
  for (int i = threadIdx.x; i < a.np; i += blockDim.x) {
    const double tx = a.pos_p[3 * i] - a.sx, ty = a.pos_p[3 * i + 1] - a.sy, tz = a.pos_p[3 * i + 2] - a.sz;
    const int pt = a.typ_p[i];
    double fx = 0, fy = 0, fz = 0, ei = 0, ri = 0;
    int ni = 0;
    for (int j = a.same ? i + 1 : 0; j < a.nq; j++) {
      const double dx = a.pos_q[3 * j] - tx, dy = a.pos_q[3 * j + 1] - ty, dz = a.pos_q[3 * j + 2] - tz;
      const double r2 = dx * dx + dy * dy + dz * dz;
      const int qt = a.typ_q[j];
      if (WHICH == 1) {
        const int col = pt * a.ntypes + qt, col2 = qt * a.ntypes + pt;
        if (a.pair.tab && r2 <= a.pair.end[col]) {
          const Tab3 q = tab3(a.pair.tab, inc, col, a.pair.begin[col], a.pair.end[col], a.pair.invstep[col], r2, shortd);
          const double pot = tab_val(q), grad = tab_grad(q);
          fx += dx * grad; fy += dy * grad; fz += dz * grad;
          atomicAdd(&a.f_q[3 * j], -dx * grad); atomicAdd(&a.f_q[3 * j + 1], -dy * grad); atomicAdd(&a.f_q[3 * j + 2], -dz * grad);
          epot += pot; ei += 0.5 * pot; atomicAdd(&a.e_q[j], 0.5 * pot);
          vir -= r2 * grad;
          if (a.nn_p) { ni++; atomicAdd(&a.nn_q[j], 1); }
        }
        if (a.rho.tab) {
          double rh = 0;
          if (r2 < a.rho.end[col]) {
            const Tab3 q = tab3(a.rho.tab, inc, col, a.rho.begin[col], a.rho.end[col], a.rho.invstep[col], r2, shortd);
            rh = tab_val(q); ri += rh;
          }
          if (pt == qt) { if (r2 < a.rho.end[col]) atomicAdd(&a.rho_q[j], rh); }
          else if (r2 < a.rho.end[col2]) {
            const Tab3 q = tab3(a.rho.tab, inc, col2, a.rho.begin[col2], a.rho.end[col2], a.rho.invstep[col2], r2, shortd);
            atomicAdd(&a.rho_q[j], tab_val(q));
          }
        }
      } else {
        const int col1 = qt * a.ntypes + pt, col2 = pt * a.ntypes + qt;
        if (r2 < a.rho.end[col1] || r2 < a.rho.end[col2]) {
          const Tab3 q1 = tab3(a.rho.tab, inc, col1, a.rho.begin[col1], a.rho.end[col1], a.rho.invstep[col1], r2, shortd);
          const double rsi = tab_grad(q1);
          double rsj = rsi;
          if (col1 != col2) { const Tab3 q2 = tab3(a.rho.tab, inc, col2, a.rho.begin[col2], a.rho.end[col2], a.rho.invstep[col2], r2, shortd); rsj = tab_grad(q2); }
          const double g = 0.5 * (a.dF_p[i] * rsj + a.dF_q[j] * rsi);
          fx += dx * g; fy += dy * g; fz += dz * g;
          atomicAdd(&a.f_q[3 * j], -dx * g); atomicAdd(&a.f_q[3 * j + 1], -dy * g); atomicAdd(&a.f_q[3 * j + 2], -dz * g);
          vir -= r2 * g;
        }
      }
    }
    atomicAdd(&a.f_p[3 * i], fx); atomicAdd(&a.f_p[3 * i + 1], fy); atomicAdd(&a.f_p[3 * i + 2], fz);
    if (WHICH == 1) { atomicAdd(&a.e_p[i], ei); if (a.rho.tab) atomicAdd(&a.rho_p[i], ri); if (a.nn_p) atomicAdd(&a.nn_p[i], ni); }
  }
  atomicAdd(&a.sums[0], epot); atomicAdd(&a.sums[1], vir);
  if (shortd) *a.is_short = 1;
}

// staging: [pos_p 3np][pos_q 3nq][dF_p np][dF_q nq][f_p 3np][f_q 3nq][e_p][e_q][rho_p][rho_q][sums 2] doubles, then ints
static int cellpair_call(imdgpu_ctx *ctx, int which, int np, const double *ort_p, const int *sorte_p, const double *dF_p, int nq,
                         const double *ort_q, const int *sorte_q, const double *dF_q, const double pbc[3], int same_cell,
                         double *kraft_p, double *kraft_q, double *poteng_p, double *poteng_q, double *rho_p, double *rho_q,
                         int *nn_p, int *nn_q, double *epot, double *virial, int *is_short)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (np <= 0 || nq <= 0) return 0;
  if (which == 2 && !ctx->rho.present) return imdgpu_fail(ctx, "do_forces_eam2: no rho_h_tab uploaded");
  const bool same_arr = same_cell != 0;
  const size_t nd = (size_t) 3 * (np + nq) + (np + nq) + 3 * (np + nq) + 4 * (size_t) (np + nq) + 2, ni = (size_t) 2 * (np + nq) + 1;
  const size_t bytes = nd * sizeof(double) + ni * sizeof(int);
  if (bytes > ctx->stage_bytes) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) (void) hipFree(ctx->d_stage);
    ctx->stage_bytes = bytes * 2 + 4096;
    HIPCHK(hipMalloc(&ctx->d_stage, ctx->stage_bytes));
  }
  std::vector<double> h(nd, 0.0); std::vector<int> hi(ni, 0);
  double *d = (double *) ctx->d_stage; int *di = (int *) (d + nd);
  size_t o = 0;
  const size_t o_pp = o; memcpy(&h[o], ort_p, 24 * (size_t) np); o += 3 * (size_t) np;
  const size_t o_pq = o; memcpy(&h[o], ort_q, 24 * (size_t) nq); o += 3 * (size_t) nq;
  const size_t o_dp = o; if (dF_p) memcpy(&h[o], dF_p, 8 * (size_t) np); o += np;
  const size_t o_dq = o; if (dF_q) memcpy(&h[o], dF_q, 8 * (size_t) nq); o += nq;
  const size_t o_fp = o; o += 3 * (size_t) np;
  const size_t o_fq = o; o += 3 * (size_t) nq;
  const size_t o_ep = o; o += np; const size_t o_eq = o; o += nq;
  const size_t o_rp = o; o += np; const size_t o_rq = o; o += nq;
  const size_t o_s = o;
  for (int i = 0; i < np; i++) hi[i] = sorte_p ? sorte_p[i] : 0;
  for (int j = 0; j < nq; j++) hi[np + j] = sorte_q ? sorte_q[j] : 0;
  HIPCHK(hipMemcpyAsync(d, h.data(), nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(di, hi.data(), ni * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  CellPairArgs a;
  a.np = np; a.nq = nq; a.same = same_cell; a.ntypes = ctx->ntypes;
  a.pos_p = d + o_pp; a.pos_q = d + o_pq; a.dF_p = d + o_dp; a.dF_q = d + o_dq;
  a.typ_p = di; a.typ_q = di + np;
  a.sx = pbc[0]; a.sy = pbc[1]; a.sz = pbc[2];
  // the same cell passed twice (p == q): ONE set of output arrays, as the reference's pointers alias
  a.f_p = d + o_fp; a.f_q = same_arr ? d + o_fp : d + o_fq; a.e_p = d + o_ep; a.e_q = same_arr ? d + o_ep : d + o_eq;
  a.rho_p = d + o_rp; a.rho_q = same_arr ? d + o_rp : d + o_rq;
  a.nn_p = nn_p ? di + np + nq : nullptr; a.nn_q = nn_p ? (same_arr ? di + np + nq : di + 2 * np + nq) : nullptr;
  a.sums = d + o_s; a.is_short = di + ni - 1;
  a.pair = ctx->pair.present ? ctx->pair.par : TabPar{}; a.rho = ctx->rho.present ? ctx->rho.par : TabPar{};
  if (!ctx->pair.present) a.pair.tab = nullptr;
  if (!ctx->rho.present) a.rho.tab = nullptr;
  if (which == 1) hipLaunchKernelGGL((k_cellpair<1>), dim3(1), dim3(64), 0, ctx->stream, a);
  else hipLaunchKernelGGL((k_cellpair<2>), dim3(1), dim3(64), 0, ctx->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h.data(), d, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(hi.data(), di, ni * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3 * np; i++) kraft_p[i] += h[o_fp + i];
  if (!same_arr) for (int j = 0; j < 3 * nq; j++) kraft_q[j] += h[o_fq + j];
  if (which == 1) {
    for (int i = 0; i < np; i++) { if (poteng_p) poteng_p[i] += h[o_ep + i]; if (rho_p) rho_p[i] += h[o_rp + i]; if (nn_p) nn_p[i] += hi[np + nq + i]; }
    if (!same_arr) for (int j = 0; j < nq; j++) { if (poteng_q) poteng_q[j] += h[o_eq + j]; if (rho_q) rho_q[j] += h[o_rq + j]; if (nn_q) nn_q[j] += hi[2 * np + nq + j]; }
    if (epot) *epot += h[o_s];
  }
  if (virial) *virial += h[o_s + 1];
  if (is_short && hi[ni - 1]) *is_short = 1;
  return 0;
}
extern "C" int imdgpu_do_forces(imdgpu_ctx *ctx, int np, const double *ort_p, const int *sorte_p, int nq, const double *ort_q,
                                const int *sorte_q, const double pbc[3], int same_cell, double *kraft_p, double *kraft_q,
                                double *poteng_p, double *poteng_q, double *eam_rho_p, double *eam_rho_q, int *numneighs_p,
                                int *numneighs_q, double *epot, double *virial, int *is_short)
{
  return cellpair_call(ctx, 1, np, ort_p, sorte_p, nullptr, nq, ort_q, sorte_q, nullptr, pbc, same_cell, kraft_p, kraft_q, poteng_p, poteng_q,
                       eam_rho_p, eam_rho_q, numneighs_p, numneighs_q, epot, virial, is_short);
}
extern "C" int imdgpu_do_forces_eam2(imdgpu_ctx *ctx, int np, const double *ort_p, const int *sorte_p, const double *eam_dF_p, int nq,
                                     const double *ort_q, const int *sorte_q, const double *eam_dF_q, const double pbc[3], int same_cell,
                                     double *kraft_p, double *kraft_q, double *virial, int *is_short)
{
  return cellpair_call(ctx, 2, np, ort_p, sorte_p, eam_dF_p, nq, ort_q, sorte_q, eam_dF_q, pbc, same_cell, kraft_p, kraft_q, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, virial, is_short);
}
// do_embedding_energy (imd_forces_eam2.c:202-230): F(rho) is ADDED to pot_eng, EAM_DF = 2 F'(rho) written, the sum returned
__global__ void k_embed_host(int n, int ntypes, const double *rho, const int *type, TabPar emb, double *epot, double *dF, double *sum)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int col = ntypes == 1 ? 0 : type[i], dummy = 0;
  const Tab3 q = tab3(emb.tab, ntypes, col, emb.begin[col], emb.end[col], emb.invstep[col], rho[i], dummy);
  epot[i] = tab_val(q); dF[i] = tab_grad(q);
  (void) sum;
}
extern "C" int imdgpu_do_embedding_energy(imdgpu_ctx *ctx, int n, const double *eam_rho, const int *sorte, double *pot_eng, double *eam_dF,
                                          double *tot_pot_energy)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->emb.present) return imdgpu_fail(ctx, "do_embedding_energy: no embed_pot uploaded");
  if (n <= 0) return 0;
  const size_t bytes = (size_t) n * (3 * sizeof(double) + sizeof(int));
  if (bytes > ctx->stage_bytes) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) (void) hipFree(ctx->d_stage);
    ctx->stage_bytes = bytes + 4096;
    HIPCHK(hipMalloc(&ctx->d_stage, ctx->stage_bytes));
  }
  double *d = (double *) ctx->d_stage; int *di = (int *) (d + 3 * (size_t) n);
  std::vector<int> t(n, 0);
  if (sorte) for (int i = 0; i < n; i++) t[i] = sorte[i];
  HIPCHK(hipMemcpyAsync(d, eam_rho, (size_t) n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(di, t.data(), (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_embed_host, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, ctx->ntypes, d, di, ctx->emb.par, d + n, d + 2 * (size_t) n, (double *) nullptr);
  HIPCHK(hipGetLastError());
  std::vector<double> e(n), g(n);
  HIPCHK(hipMemcpyAsync(e.data(), d + n, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(g.data(), d + 2 * (size_t) n, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double s = 0;
  for (int i = 0; i < n; i++) { pot_eng[i] += e[i]; eam_dF[i] = g[i]; s += e[i]; }      // in atom order, like the reference's loop
  if (tot_pot_energy) *tot_pot_energy += s;
  return 0;
}

extern "C" int imdgpu_get_sums(imdgpu_ctx *ctx, double sums[8], double *ekin, int *nbl_count)
{
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(ctx->h_sums, ctx->d_sums, SUM_N * sizeof(double), hipMemcpyDeviceToHost));
  if (sums) for (int q = 0; q < 8; q++) sums[q] = ctx->h_sums[q];
  if (ekin) *ekin = ctx->h_sums[SUM_EKIN];
  if (nbl_count) *nbl_count = ctx->nbl_count;
  return 0;
}

// perm[r] = device index of the atom that is r-th in the reference's cell order (`cells` list: x outer,
// z inner; slots inside a cell as on the device); refstart[key] = first r of that cell
static int build_ref_perm(imdgpu_ctx *ctx, std::vector<int> *perm, std::vector<int> *refstart)
{
  const Geo g = make_geo(ctx);
  std::vector<int> ic(ctx->nkeys + 1), is(ctx->nkeys + 1);
  HIPCHK(hipMemcpy(ic.data(), ctx->icnt, ic.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(is.data(), ctx->istart, is.size() * sizeof(int), hipMemcpyDeviceToHost));
  if (perm) perm->resize(ctx->nlocal);
  if (refstart) refstart->assign(ctx->nkeys + 1, 0);
  int r = 0;
  for (int ix = 0; ix < g.nx; ix++) for (int iy = 0; iy < g.ny; iy++) for (int iz = 0; iz < g.nz; iz++) {
    int key = xyz_to_key(ix, iy, iz, g);
    if (refstart) (*refstart)[key] = r;
    for (int t = 0; t < ic[key]; t++, r++) if (perm) (*perm)[r] = is[key] + t;
  }
  if (r != ctx->nlocal) return imdgpu_fail(ctx, "internal: cell tables hold %d atoms, nlocal=%d", r, ctx->nlocal);
  return 0;
}

extern "C" int imdgpu_download_atoms(imdgpu_ctx *ctx, int *cell_n, double *ort, double *impuls, double *kraft,
                                     double *pot_eng, double *eam_rho, double *eam_dF, int *numneighs, int *nummer,
                                     int *sorte, double *nbl_pos)
{
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal, cap = ctx->cap_local;
  if (cell_n) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<int> cc(ctx->nall_cells);
    HIPCHK(hipMemcpy(cc.data(), ctx->cell_cnt, cc.size() * sizeof(int), hipMemcpyDeviceToHost));
    const int *D = ctx->dim;
    for (int c = 0; c < ctx->nall_cells; c++) cell_n[c] = 0;
    for (int i = 1; i < D[0] - 1; i++) for (int j = 1; j < D[1] - 1; j++) for (int k = 1; k < D[2] - 1; k++) {
      int c = (i * D[1] + j) * D[2] + k; cell_n[c] = cc[c];
    }
  }
  if (n == 0) return 0;
  // the fields are put into the host's cell order ON THE DEVICE (one staging buffer, field after field) and leave in
  // whole-array transfers on the stream; one synchronisation at the end
  size_t need = 0;
  need += (ort ? 24 : 0) + (eam_dF ? 8 : 0) + (impuls ? 24 : 0) + (kraft ? 24 : 0) + (nbl_pos ? 24 : 0) + (pot_eng ? 8 : 0) +
          (eam_rho ? 8 : 0) + (numneighs ? 8 : 0) + (nummer ? 8 : 0) + (sorte ? 8 : 0);
  if (need == 0) return 0;
  if (ensure_refperm(ctx) || ensure_stage(ctx, need * (size_t) n)) return 1;
  char *st = (char *) ctx->d_stage;
  const dim3 G(nblk(n, 256)), B(256);
  auto take = [&](size_t bytes_per_atom) { char *p = st; st += bytes_per_atom * (size_t) n; return p; };
  if (ort || eam_dF) {
    double *so = ort ? (double *) take(24) : nullptr, *sd = eam_dF ? (double *) take(8) : nullptr;
    hipLaunchKernelGGL(k_l1_get_pos, G, B, 0, ctx->stream, n, ctx->d_refperm, ctx->pos, so, sd);
    if (ort) HIPCHK(hipMemcpyAsync(ort, so, (size_t) n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (eam_dF) HIPCHK(hipMemcpyAsync(eam_dF, sd, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  auto planes = [&](const double *dev, double *out) -> int {
    double *so = (double *) take(24);
    hipLaunchKernelGGL(k_l1_get_planes, G, B, 0, ctx->stream, n, cap, ctx->d_refperm, dev, so);
    HIPCHK(hipMemcpyAsync(out, so, (size_t) n * 24, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  };
  auto scal_d = [&](const double *dev, double *out) -> int {
    double *so = (double *) take(8);
    hipLaunchKernelGGL((k_l1_get_scalar<double>), G, B, 0, ctx->stream, n, ctx->d_refperm, dev, so);
    HIPCHK(hipMemcpyAsync(out, so, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  };
  auto scal_i = [&](const int *dev, int *out) -> int {
    int *so = (int *) take(8);
    hipLaunchKernelGGL((k_l1_get_scalar<int>), G, B, 0, ctx->stream, n, ctx->d_refperm, dev, so);
    HIPCHK(hipMemcpyAsync(out, so, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  };
  if (impuls && planes(ctx->p, impuls)) return 1;
  if (kraft && planes(ctx->f, kraft)) return 1;
  if (nbl_pos && planes(ctx->ref, nbl_pos)) return 1;
  if (pot_eng && scal_d(ctx->epot, pot_eng)) return 1;
  if (eam_rho && scal_d(ctx->rhoh, eam_rho)) return 1;
  if (numneighs && scal_i(ctx->numneighs, numneighs)) return 1;
  if (nummer && scal_i(ctx->id, nummer)) return 1;
  if (sorte && scal_i(ctx->type, sorte)) return 1;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// Reference cl_off: prefix sum over ALL cells in linear order where a ghost cell
// holds a copy of the inner cell it images, except the x=0 wall which the
// reference's AR mode leaves empty (send_cells, imd_comm_force_3d.c:356-377).
extern "C" int imdgpu_export_nblist(imdgpu_ctx *ctx, int *tl, int *tb, int *cl_off, int *ntb)
{
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->cpu_dim[0] * ctx->cpu_dim[1] * ctx->cpu_dim[2] != 1)
    return imdgpu_fail(ctx, "export_nblist: single-rank arrangements only");
  const int n = ctx->nlocal, nall = ctx->nall_cells;
  const int *D = ctx->dim;
  std::vector<int> cc(nall), off(nall);
  HIPCHK(hipMemcpy(cc.data(), ctx->cell_cnt, nall * sizeof(int), hipMemcpyDeviceToHost));
  int at = 0;
  for (int i = 0; i < D[0]; i++) for (int j = 0; j < D[1]; j++) for (int k = 0; k < D[2]; k++) {
    int c = (i * D[1] + j) * D[2] + k;
    int si = i == 0 ? D[0] - 2 : (i == D[0] - 1 ? 1 : i), sj = j == 0 ? D[1] - 2 : (j == D[1] - 1 ? 1 : j),
        sk = k == 0 ? D[2] - 2 : (k == D[2] - 1 ? 1 : k);
    int cnt = (i == 0) ? 0 : cc[(si * D[1] + sj) * D[2] + sk];
    off[c] = at; at += cnt;
  }
  if (cl_off) memcpy(cl_off, off.data(), nall * sizeof(int));
  int *d_off = nullptr, *d_tl = nullptr, *d_tb = nullptr, *d_rs = nullptr;
  Geo g = make_geo(ctx);
  std::vector<int> refstart;
  if (build_ref_perm(ctx, nullptr, &refstart)) return 1;
  HIPCHK(hipMalloc((void **) &d_rs, refstart.size() * sizeof(int)));
  HIPCHK(hipMemcpy(d_rs, refstart.data(), refstart.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void **) &d_off, nall * sizeof(int)));
  HIPCHK(hipMalloc((void **) &d_tl, (n + 1) * sizeof(int)));
  HIPCHK(hipMemcpy(d_off, off.data(), nall * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_tl, 0, (n + 1) * sizeof(int)));
  hipLaunchKernelGGL((k_export_half<false>), dim3(nblk(n, 128)), dim3(128), 0, ctx->stream, n, ctx->pos, ctx->acell,
                     ctx->cell_start, ctx->cell_cnt, g, ctx->cellsz, d_off, d_rs, ctx->istart, d_tl, (int *) nullptr);
  size_t need = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_tl, d_tl, n + 1, ctx->stream));
  void *tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, need + 16));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp, need, d_tl, d_tl, n + 1, ctx->stream));
  int total = 0;
  HIPCHK(hipMemcpyAsync(&total, d_tl + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ntb) *ntb = total;
  if (tl) HIPCHK(hipMemcpy(tl, d_tl, (n + 1) * sizeof(int), hipMemcpyDeviceToHost));
  if (tb) {
    HIPCHK(hipMalloc((void **) &d_tb, (size_t) (total + 1) * sizeof(int)));
    hipLaunchKernelGGL((k_export_half<true>), dim3(nblk(n, 128)), dim3(128), 0, ctx->stream, n, ctx->pos, ctx->acell,
                       ctx->cell_start, ctx->cell_cnt, g, ctx->cellsz, d_off, d_rs, ctx->istart, d_tl, d_tb);
    HIPCHK(hipStreamSynchronize(ctx->stream));       // ctx->stream is non-blocking w.r.t. the null stream
    HIPCHK(hipMemcpy(tb, d_tb, (size_t) total * sizeof(int), hipMemcpyDeviceToHost));
    (void) hipFree(d_tb);
  }
  (void) hipFree(tmp); (void) hipFree(d_off); (void) hipFree(d_tl); (void) hipFree(d_rs);
  return 0;
}

extern "C" int imdgpu_profile_enable(imdgpu_ctx *ctx, int on)
{
  if (!on) imdgpu_prof_resolve(ctx);
  ctx->prof_on = on != 0;
  return 0;
}
extern "C" int imdgpu_profile_reset(imdgpu_ctx *ctx)
{
  imdgpu_prof_resolve(ctx);
  for (int i = 0; i < PROF_COUNT; i++) { ctx->prof_n[i] = 0; ctx->prof_ms[i] = 0; }
  return 0;
}
extern "C" int imdgpu_profile_get(imdgpu_ctx *ctx, const char *name, long long *launches, double *ms)
{
  imdgpu_prof_resolve(ctx);
  for (int i = 0; i < PROF_COUNT; i++)
    if (!strcmp(name, prof_names[i])) {
      if (launches) *launches = ctx->prof_n[i];
      if (ms) *ms = ctx->prof_ms[i];
      return 0;
    }
  return imdgpu_fail(ctx, "unknown profile name %s", name);
}
