// imdgpu_halo.hip -- device side of the multi-GPU spatial decomposition (one rank per GPU).
//
// The reference exchanges ghost cells with Plimpton's 3-stage scheme over MPI
// (send_cells / send_forces, imd_comm_force_3d.c:222-396,569-715) and migrates
// atoms in fix_cells -> send_atoms (imd_fix_cells_3d.c:43-306,436-542).  Here the
// decomposition logic (which cells go where, in which order, with which periodic
// shift) lives in the host layer (imd_amd/parallel.py) and the transport is
// RCCL/xGMI through torch.distributed; this file provides the generic device
// operations that layer is built from:
//   - cell tables out / ghost layout in,
//   - gather(index list, shift) -> contiguous send buffer, scatter(recv buffer) -> ghost atoms,
//   - migration: mark atoms that left the rank's domain along one direction,
//     pack them, drop them, append the arrivals,
//   - the force passes and the integrator as separate asynchronous launches, so
//     that exchanges can be placed between them.
// Full-list gather kernels need no reverse (force / rho) halo: only positions
// and dF travel, both forward.
#include "imdgpu_internal.h"

#include <hipcub/hipcub.hpp>

#include <vector>

static inline int nblk(long long n, int b) { return n <= 0 ? 1 : (int) ((n + b - 1) / b); }   // never an empty grid: a rank may own no atoms (vacuum)

extern "C" int imdgpu_get_cell_tables(imdgpu_ctx *ctx, int *cell_start, int *cell_cnt)
{
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (cell_start) HIPCHK(hipMemcpy(cell_start, ctx->cell_start, ctx->nall_cells * sizeof(int), hipMemcpyDeviceToHost));
  if (cell_cnt) HIPCHK(hipMemcpy(cell_cnt, ctx->cell_cnt, ctx->nall_cells * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

// Host decides where the ghost atoms of every ghost cell live (start >= nlocal) -- the inner cells'
// entries of the two arrays must be the ones returned by imdgpu_get_cell_tables.
extern "C" int imdgpu_set_ghost_layout(imdgpu_ctx *ctx, const int *cell_start, const int *cell_cnt, int nghost)
{
  imdgpu_dist_cancel_ahead(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->external_halo) return imdgpu_fail(ctx, "set_ghost_layout: only with a multi-rank geometry (cpu_dim > 1)");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (imdgpu_ensure_capacity(ctx, ctx->nlocal, ctx->nlocal + nghost)) return 1;
  HIPCHK(hipMemcpy(ctx->cell_start, cell_start, ctx->nall_cells * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->cell_cnt, cell_cnt, ctx->nall_cells * sizeof(int), hipMemcpyHostToDevice));
  ctx->nghost = nghost;
  ctx->have_valid_nbl = 0;
  return 0;
}

// ---- host-buffer variants for a C host that runs its OWN halo exchange (level-1 MPI builds of the reference: after its
// send_cells the buffer cells of the host hold the ghost atoms).  ghost_cell_n: atoms per cell for all nallcells cells,
// 0 for the inner ones; ort / sorte / dF: the ghost atoms in cell order (cell index ascending, slots as on the host).
extern "C" int imdgpu_upload_ghosts(imdgpu_ctx *ctx, const int *ghost_cell_n, int nghost, const double *ort, const int *sorte)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->external_halo) return imdgpu_fail(ctx, "upload_ghosts: only with a multi-rank geometry (cpu_dim > 1)");
  const int nall = ctx->nall_cells;
  std::vector<int> cs(nall), cc(nall);
  if (imdgpu_get_cell_tables(ctx, cs.data(), cc.data())) return 1;
  long long off = 0;
  for (int c = 0; c < nall; c++)
    if (ghost_cell_n[c] > 0) {
      if (cc[c] != 0 && cs[c] < ctx->nlocal) return imdgpu_fail(ctx, "upload_ghosts: cell %d holds local atoms", c);
      cs[c] = ctx->nlocal + (int) off; cc[c] = ghost_cell_n[c]; off += ghost_cell_n[c];
    } else if (cs[c] >= ctx->nlocal) { cs[c] = ctx->nlocal; cc[c] = 0; }       // a ghost cell that is empty now
  if (off != nghost) return imdgpu_fail(ctx, "upload_ghosts: ghost_cell_n sums to %lld, nghost = %d", off, nghost);
  if (imdgpu_set_ghost_layout(ctx, cs.data(), cc.data(), nghost)) return 1;
  if (nghost == 0) return 0;
  std::vector<double4> hp(nghost);
  for (int g = 0; g < nghost; g++) hp[g] = make_double4(ort[3 * g], ort[3 * g + 1], ort[3 * g + 2], 0.0);
  HIPCHK(hipMemcpy(ctx->pos + ctx->nlocal, hp.data(), (size_t) nghost * sizeof(double4), hipMemcpyHostToDevice));
  if (sorte) HIPCHK(hipMemcpy(ctx->type + ctx->nlocal, sorte, (size_t) nghost * sizeof(int), hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(ctx->type + ctx->nlocal, 0, (size_t) nghost * sizeof(int)));
  return 0;
}

// new positions (ort != NULL) and/or F'(rho) (dF != NULL) of the ghost atoms, same order as in imdgpu_upload_ghosts
extern "C" int imdgpu_update_ghosts(imdgpu_ctx *ctx, int nghost, const double *ort, const double *dF)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (nghost != ctx->nghost) return imdgpu_fail(ctx, "update_ghosts: %d ghost atoms given, %d on the device", nghost, ctx->nghost);
  if (nghost == 0 || (!ort && !dF)) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<double4> hp(nghost);
  HIPCHK(hipMemcpy(hp.data(), ctx->pos + ctx->nlocal, (size_t) nghost * sizeof(double4), hipMemcpyDeviceToHost));
  for (int g = 0; g < nghost; g++) {
    if (ort) { hp[g].x = ort[3 * g]; hp[g].y = ort[3 * g + 1]; hp[g].z = ort[3 * g + 2]; }
    if (dF) hp[g].w = dF[g];
  }
  HIPCHK(hipMemcpy(ctx->pos + ctx->nlocal, hp.data(), (size_t) nghost * sizeof(double4), hipMemcpyHostToDevice));
  return 0;
}

// what: 0 = position record (x,y,z,dF) as 4 doubles, shifted; 1 = dF only (1 double); 2 = type (int);
//       3 = shifted position only (3 doubles: the per-step halo, 24 instead of 32 bytes per atom on the wire)
__global__ void k_halo_gather(int what, int n, const int *idx, const double4 *pos, const int *type, double sx,
                              double sy, double sz, void *out)
{
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  int i = idx[m];
  if (what == 0) { double4 r = pos[i]; r.x += sx; r.y += sy; r.z += sz; ((double4 *) out)[m] = r; }
  else if (what == 1) ((double *) out)[m] = pos[i].w;
  else if (what == 3) { double4 r = pos[i]; double *o = (double *) out + 3 * (size_t) m; o[0] = r.x + sx; o[1] = r.y + sy; o[2] = r.z + sz; }
  else ((int *) out)[m] = type[i];
}
__global__ void k_halo_scatter(int what, int n, const int *idx, double4 *pos, int *type, const void *in)
{
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  int i = idx[m];
  if (what == 0) pos[i] = ((const double4 *) in)[m];
  else if (what == 1) pos[i].w = ((const double *) in)[m];
  else if (what == 3) { const double *q = (const double *) in + 3 * (size_t) m; pos[i].x = q[0]; pos[i].y = q[1]; pos[i].z = q[2]; }
  else type[i] = ((const int *) in)[m];
}

extern "C" int imdgpu_halo_gather(imdgpu_ctx *ctx, int what, const int *dev_idx, int n, const double shift[3], void *dev_out)
{
  if (n <= 0) return 0;
  ProfScope ps(ctx, PROF_GHOSTS);
  hipLaunchKernelGGL(k_halo_gather, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, what, n, dev_idx, ctx->pos, ctx->type,
                     shift ? shift[0] : 0.0, shift ? shift[1] : 0.0, shift ? shift[2] : 0.0, dev_out);
  HIPCHK(hipGetLastError());
  return 0;
}
// both faces of one exchange stage in one launch: entries [0, n_lo) take shift_lo, [n_lo, n) shift_hi
__global__ void k_halo_gather2(int what, int n_lo, int n, const int *idx, const double4 *pos, const int *type, double lx,
                               double ly, double lz, double hx, double hy, double hz, void *out)
{
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  int i = idx[m];
  const bool lo = m < n_lo;
  if (what == 0) { double4 r = pos[i]; r.x += lo ? lx : hx; r.y += lo ? ly : hy; r.z += lo ? lz : hz; ((double4 *) out)[m] = r; }
  else if (what == 1) ((double *) out)[m] = pos[i].w;
  else if (what == 3) {
    double4 r = pos[i]; double *o = (double *) out + 3 * (size_t) m;
    o[0] = r.x + (lo ? lx : hx); o[1] = r.y + (lo ? ly : hy); o[2] = r.z + (lo ? lz : hz);
  }
  else ((int *) out)[m] = type[i];
}
extern "C" int imdgpu_halo_gather2(imdgpu_ctx *ctx, int what, const int *dev_idx, int n_lo, int n, const double shift_lo[3],
                                   const double shift_hi[3], void *dev_out)
{
  if (n <= 0) return 0;
  ProfScope ps(ctx, PROF_GHOSTS);
  hipLaunchKernelGGL(k_halo_gather2, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, what, n_lo, n, dev_idx, ctx->pos, ctx->type,
                     shift_lo[0], shift_lo[1], shift_lo[2], shift_hi[0], shift_hi[1], shift_hi[2], dev_out);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int imdgpu_halo_scatter(imdgpu_ctx *ctx, int what, const int *dev_idx, int n, const void *dev_in)
{
  if (n <= 0) return 0;
  ProfScope ps(ctx, PROF_GHOSTS);
  hipLaunchKernelGGL(k_halo_scatter, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, what, n, dev_idx, ctx->pos, ctx->type,
                     dev_in);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- migration -----------------------------------------------------------------
#define MIG_REC 10     // doubles per migrating atom: x y z px py pz mass type id pad

// Which rank (along dimension d) owns the cell that contains the atom?  Signed rank distance, wrapped over
// the periodic rank grid; <0: leaves through the low face, >0: through the high face.
__global__ void k_migrate_mark(int n, int d, const double4 *pos, Geo g, int ncpu, int *flag, int *idx_lo, int *idx_hi,
                               int *counters)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double4 r = pos[i];
  const double *t = g.tbox + 3 * d;
  double s = r.x * t[0] + r.y * t[1] + r.z * t[2];
  if (g.pbc[d] == 1) s -= floor(s);
  int c = (int) (g.gdim[d] * s);
  if (c >= g.gdim[d]) c = g.gdim[d] - 1; else if (c < 0) c = 0;
  int per = g.dim[d] - 2, dr = c / per - g.my_coord[d];
  if (g.pbc[d] == 1) { if (2 * dr > ncpu) dr -= ncpu; else if (2 * dr < -ncpu) dr += ncpu; }
  int f = dr < 0 ? 1 : (dr > 0 ? 2 : 0);
  flag[i] = f;
  if (f == 1) idx_lo[atomicAdd(&counters[0], 1)] = i;
  else if (f == 2) idx_hi[atomicAdd(&counters[1], 1)] = i;
}

__global__ void k_migrate_gather(int n, const int *idx, int cap, const double4 *pos, const double *p, const double *mass,
                                 const int *type, const int *id, const int *fdc, double *out)
{
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  int i = idx[m];
  double4 r = pos[i];
  double *o = out + (size_t) m * MIG_REC;
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
  o[3] = p[i]; o[4] = p[cap + i]; o[5] = p[2 * (size_t) cap + i];
  o[6] = mass[i]; o[7] = (double) type[i]; o[8] = (double) id[i]; o[9] = (double) fdc[i];   // fdi/fdj/fdk travel too (imd_mpi_util.c:399)
}

__global__ void k_migrate_compact(int n, int cap, const int *perm, const double4 *pos, double4 *pos2, const int *type,
                                  int *type2, const int *id, int *id2, const double *p, double *p2, const double *mass,
                                  double *mass2, const int *fdc, int *fdc2)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = perm[i];
  pos2[i] = pos[s]; type2[i] = type[s]; id2[i] = id[s]; mass2[i] = mass[s]; fdc2[i] = fdc[s];
  p2[i] = p[s]; p2[cap + i] = p[cap + s]; p2[2 * (size_t) cap + i] = p[2 * (size_t) cap + s];
}

__global__ void k_migrate_append(int n, int at, int cap, const double *in, double4 *pos, int *type, int *id, double *p,
                                 double *mass, int *fdc)
{
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const double *o = in + (size_t) m * MIG_REC;
  int i = at + m;
  pos[i] = make_double4(o[0], o[1], o[2], 0.0);
  p[i] = o[3]; p[cap + i] = o[4]; p[2 * (size_t) cap + i] = o[5];
  mass[i] = o[6]; type[i] = (int) o[7]; id[i] = (int) o[8]; fdc[i] = (int) o[9];
}

__global__ void k_iota_stay(int n, const int *flag, int *iota, char *stay)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  iota[i] = i; stay[i] = flag[i] == 0;
}

extern "C" int imdgpu_migrate_scan(imdgpu_ctx *ctx, int dim, int *n_lo, int *n_hi)
{
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal;
  if (ctx->mig_cap < ctx->cap_local) {
    if (dev_alloc(ctx, &ctx->mig_idx[0], ctx->cap_local) || dev_alloc(ctx, &ctx->mig_idx[1], ctx->cap_local) ||
        dev_alloc(ctx, &ctx->mig_flag, ctx->cap_local)) return 1;
    ctx->mig_cap = ctx->cap_local;
  }
  Geo g = imdgpu_make_geo(ctx);
  int *cnt = ctx->d_counters + 8;                 // two scratch counters
  HIPCHK(hipMemsetAsync(cnt, 0, 2 * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_migrate_mark, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, dim, ctx->pos, g, ctx->cpu_dim[dim],
                     ctx->mig_flag, ctx->mig_idx[0], ctx->mig_idx[1], cnt);
  int h[2];
  HIPCHK(hipMemcpyAsync(h, cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->mig_n[0] = h[0]; ctx->mig_n[1] = h[1];
  if (n_lo) *n_lo = h[0];
  if (n_hi) *n_hi = h[1];
  return 0;
}

extern "C" int imdgpu_migrate_gather(imdgpu_ctx *ctx, int dir, void *dev_out)
{
  const int n = ctx->mig_n[dir];
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_migrate_gather, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, ctx->mig_idx[dir], ctx->cap_local,
                     ctx->pos, ctx->p, ctx->mass, ctx->type, ctx->id, ctx->fdc, (double *) dev_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// drop the atoms marked by the last imdgpu_migrate_scan, append the arrivals (records as written by
// imdgpu_migrate_gather on the neighbour ranks)
extern "C" int imdgpu_migrate_apply(imdgpu_ctx *ctx, const void *in_lo, int n_in_lo, const void *in_hi, int n_in_hi)
{
  imdgpu_dist_cancel_ahead(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nlocal, nout = ctx->mig_n[0] + ctx->mig_n[1], nstay = n - nout, nnew = nstay + n_in_lo + n_in_hi;
  if (nout > 0) {
    // stable partition of the stayers (keeps the deterministic cell order)
    char *stay = (char *) ctx->keys2;                      // scratch: cap_local ints >= cap_local chars
    int *iota = ctx->perm, *sel = ctx->perm2, *dnum = ctx->d_counters + 10;
    hipLaunchKernelGGL(k_iota_stay, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, ctx->mig_flag, iota, stay);
    size_t need = 0;
    HIPCHK(hipcub::DeviceSelect::Flagged(nullptr, need, iota, stay, sel, dnum, n, ctx->stream));
    if (need > ctx->cub_tmp_bytes) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (ctx->cub_tmp) (void) hipFree(ctx->cub_tmp);
      HIPCHK(hipMalloc(&ctx->cub_tmp, need + 1024));
      ctx->cub_tmp_bytes = need + 1024;
    }
    size_t tb = ctx->cub_tmp_bytes;
    HIPCHK(hipcub::DeviceSelect::Flagged(ctx->cub_tmp, tb, iota, stay, sel, dnum, n, ctx->stream));
    hipLaunchKernelGGL(k_migrate_compact, dim3(nblk(nstay, 256)), dim3(256), 0, ctx->stream, nstay, ctx->cap_local, sel,
                       ctx->pos, ctx->pos2, ctx->type, ctx->type2, ctx->id, ctx->id2, ctx->p, ctx->p2, ctx->mass, ctx->mass2,
                       ctx->fdc, ctx->fdc2);
    std::swap(ctx->pos, ctx->pos2); std::swap(ctx->type, ctx->type2); std::swap(ctx->id, ctx->id2);
    std::swap(ctx->p, ctx->p2); std::swap(ctx->mass, ctx->mass2); std::swap(ctx->fdc, ctx->fdc2);
  }
  ctx->nlocal = nstay;
  if (nnew > ctx->cap_local) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return imdgpu_fail(ctx, "migration: %d atoms exceed the local capacity %d (10%% head-room over the initial count)", nnew, ctx->cap_local);
  }
  if (nnew > ctx->cap_tot && imdgpu_ensure_capacity(ctx, nstay, nnew)) return 1;
  if (n_in_lo > 0)
    hipLaunchKernelGGL(k_migrate_append, dim3(nblk(n_in_lo, 256)), dim3(256), 0, ctx->stream, n_in_lo, nstay, ctx->cap_local,
                       (const double *) in_lo, ctx->pos, ctx->type, ctx->id, ctx->p, ctx->mass, ctx->fdc);
  if (n_in_hi > 0)
    hipLaunchKernelGGL(k_migrate_append, dim3(nblk(n_in_hi, 256)), dim3(256), 0, ctx->stream, n_in_hi, nstay + n_in_lo,
                       ctx->cap_local, (const double *) in_hi, ctx->pos, ctx->type, ctx->id, ctx->p, ctx->mass, ctx->fdc);
  HIPCHK(hipGetLastError());
  ctx->nlocal = nnew;
  ctx->cells_epoch++;
  ctx->mig_n[0] = ctx->mig_n[1] = 0;
  ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1;
  return 0;
}

// ---- split step ------------------------------------------------------------------
extern "C" int imdgpu_forces_phase(imdgpu_ctx *ctx, int phase)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (phase < 1 || phase > 3) return imdgpu_fail(ctx, "forces_phase: phase must be 1, 2 or 3");
  if (!ctx->have_valid_nbl) return imdgpu_fail(ctx, "forces_phase: neighbour list invalid (imdgpu_make_nblist first)");
  return imdgpu_launch_forces_phase(ctx, phase);
}

extern "C" int imdgpu_integrate(imdgpu_ctx *ctx)
{
  HIPCHK(hipSetDevice(ctx->device));
  // A host that drives the step piece by piece (imdgpu_calc_ttm / imdgpu_ttm_lattice_step, then this call) has nobody behind it who
  // would run the integrator AGAIN: the persistent sub-step launch of this step is settled first -- a launch that stepped aside
  // (status 1) gets its sub-steps through the launch chain now -- so the integrator below is never guarded by its status word and
  // never skipped (before, one MD step of motion was silently lost).  Costs this entry point one wait for the status
  // word per step; imdgpu_md_step_end and imdgpu_dist_md_step keep the guarded form and re-run the integrator themselves.
  if (ctx->ttm && !ctx->step_open && imdgpu_ttm_persist_settle(ctx) < 0) return 1;
  return imdgpu_launch_integrate(ctx);
}

extern "C" int imdgpu_get_invalid(imdgpu_ctx *ctx, int *flag)
{
  HIPCHK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, CNT_N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (flag) *flag = ctx->h_counters[CNT_INVALID];
  if (ctx->h_counters[CNT_TTM_TIMEOUT])
    return imdgpu_fail(ctx, "TTM mailbox: a neighbour rank did not publish its lattice faces in time (rank gone or stalled)");
  return 0;
}

extern "C" int imdgpu_invalidate(imdgpu_ctx *ctx)
{
  imdgpu_dist_cancel_ahead(ctx);
  ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1;
  return 0;
}
