// imdgpu_internal.h -- context and helpers shared by the HIP translation units
// of libimdgpu.so (gfx950 only).  Public ABI: include/imdgpu.h.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/imdgpu.h"

#define IMDGPU_MAXCOL 16          // ntypes <= 4

// Environment names, two classes.
//  cfg_env: CONFIGURATION, read by every build -- choices among implementations that are all held against the goldens (IMDGPU_TILE,
//           IMDGPU_MODE), diagnostics that change no result (IMDGPU_TIMELINE, IMD_DIST_TIMING) and the transport of the multi-rank step
//           (IMDGPU_DIST_TRANSPORT, IMDGPU_DIST_SELF_COMM).
//  exp_env: EXPERIMENT switches (timing splits that skip work, scheduling knobs, forced failure paths).  They exist only in a library
//           built with `make EXTRA=-DIMDGPU_EXPERIMENTS`; the default build does not look at the environment for them, so a stray
//           IMDGPU_DBG=1 cannot turn the force loops off.  imdgpu_get_info("experiments") tells how a library was built (a bit mask,
//           see imdgpu_experiment_mask), bench.py refuses anything but 0.  Test hooks are options (imdgpu_set_option), not names here.
#include <cstdlib>
static inline const char *cfg_env(const char *n) { return getenv(n); }
#ifdef IMDGPU_EXPERIMENTS
static inline const char *exp_env(const char *n) { return getenv(n); }
#else
static inline const char *exp_env(const char *) { return nullptr; }
#endif
static inline int exp_env_int(const char *n, int dflt) { const char *e = exp_env(n); return e ? atoi(e) : dflt; }
#define IMDGPU_NPART 4            // partial sums per block: e_pair, vir1, e_embed / vir3, spare

struct TabPar {                   // one potential table as kernel argument
  const double *tab;              // device, (maxsteps+2)*ncols
  int ncols, nrows;               // nrows = maxsteps+2
  double begin[IMDGPU_MAXCOL], end[IMDGPU_MAXCOL], invstep[IMDGPU_MAXCOL];
};

struct DevTable {
  double *d_tab = nullptr;
  TabPar par{};
  bool present = false;
};

enum ProfId {
  PROF_PASS1 = 0, PROF_PASS2, PROF_PASS3, PROF_GHOSTS, PROF_NBLIST, PROF_FIXCELLS,
  PROF_INTEGRATE, PROF_CHECK, PROF_REDUCE, PROF_TTM_UPDATE, PROF_TTM_DIFF,
  PROF_TTM_FILL, PROF_PASS1B, PROF_COUNT
};

struct ProfPending { hipEvent_t a, b; int id; };

struct imdgpu_ctx {
  int device = 0, num_cus = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: TTM lattice work overlapped with passes 2-3
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_ufd = nullptr;
  int ufd_pending = 0;            // the forked TTM chain recorded ev_ufd after update_fd: the EAM force pass waits for it
  int (*after_pass1)(imdgpu_ctx *, void *) = nullptr; void *after_pass1_arg = nullptr;
  int ttm_steps = 0;
  bool own_stream = false;
  std::string err;

  // geometry -------------------------------------------------------------
  double box[9]{}, tbox[9]{};
  int pbc[3]{1, 1, 1}, dim[3]{}, gdim[3]{}, cpu_dim[3]{1, 1, 1}, my_coord[3]{};
  int nall_cells = 0, ninner_cells = 0, nghost_cells = 0;
  int tile[3]{3, 3, 3}, ntile[3]{}, ntiles = 0, CPT = 0, nkeys = 0;   // tiling of the inner cells
  bool have_geometry = false;
  std::vector<int> h_gsrc_xyz;      // per ghost cell: inner coords of its periodic source (or -1)

  // parameters -------------------------------------------------------------
  int ntypes = 1, count_ttm = 0;
  double cellsz = 0, nbl_margin = 0, timestep = 0;
  DevTable pair, rho, emb;
  double *d_merged = nullptr;     // (phi,rho) interleaved + 3 zero rows when both share one r^2 grid
  // one atom type, merged tables: the gradient of rho_h_tab as two coefficients per row, (dv - d2v/2, d2v), + one zero row:
  // what the EAM force pass reads instead of three table rows (k_pass3_tile_coef)
  double *d_coef3 = nullptr; int ncoef = 0;
  // several atom types: coefficient records of pair_pot / rho_h_tab per column and row (k_pass1_tile_rec / k_pass3_tile_rec)
  double *d_recP = nullptr, *d_recR = nullptr, *d_rec3 = nullptr; int nrecP = 0, nrecR = 0;
  int tile_rec = 0;               // this list's force kernels are the record kernels (set by make_nblist)
  int tile_rec_opt = 1;           // option "tile_rec": 0 never (the older several-type tile kernels / global gather), 1 whenever several types
  // ... and the compact form for the LDS: the distinct columns of each table one behind the other (len + 2 rows each), three zero rows last
  double *d_compP = nullptr, *d_compR = nullptr; int ncompP = 0, ncompR = 0; int loffP[IMDGPU_MAXCOL] = {0}, loffR[IMDGPU_MAXCOL] = {0};
  int rec_lds = 0;                // which tables this list's record kernels keep in the LDS: 1 pair_pot (pass 1), 2 rho_h_tab (pass 1), 4 rho_h_tab (pass 3)
  int rec_lds_opt = -1;           // option "tile_rec_lds": that mask forced (tables that do not fit are dropped from it); -1 = chosen by make_nblist
  int act_ranksort_max = 1024;    // option "act_ranksort_max" (test hook): above this many simultaneous fallback activations the ordered walk replaces the rank sort
  int ttm_remote_opt = 1;         // option "ttm_remote": 0 keeps the host mailbox for the decomposed lattice's faces
  int ttm_remote_fail = 0;        // option "ttm_remote_fail" (test hook): rank + 1 that pretends it cannot set the remote records up
  int df_halo_beside_head = -1;   // option: decomposed lattice, dF halo beside the chain's head (1), in front of the fork (0), by transport (-1)
  int no_merge = 0;               // option "no_merge" (tests, before upload_tables): keep the tables apart even on a shared grid
  int p3_coef = 1;                // option "p3_coef": 0 keeps k_pass3_tile for every step
  int p3_klo = 0;                 // option "p3_klo" (tests): keep at least this many of the lowest rows out of the LDS
  double *d_restr = nullptr;      // ntypes*3
  int restr_identity = 1;         // every factor is 1: the integrators need not write the (unchanged) forces back

  // atoms (device).  local atoms [0,nlocal) sorted by inner-cell rank, ghost
  // atoms [nlocal,ntot) grouped by ghost cell in linear cell order ----------
  int nlocal = 0, nghost = 0, cap_local = 0, cap_tot = 0;
  double4 *pos = nullptr;         // x,y,z,dF   [cap_tot]
  int *type = nullptr;            // [cap_tot]
  int *id = nullptr;              // nummer [cap_local]
  int *acell = nullptr;           // inner-cell rank of each local atom [cap_local]
  double *p = nullptr;            // momenta 3 planes [3*cap_local]
  double *f = nullptr;            // forces  3 planes
  double *ref = nullptr;          // nbl_pos 3 planes
  double *mass = nullptr, *epot = nullptr, *rhoh = nullptr;
  int *numneighs = nullptr;
  int *fdc = nullptr, *fdc2 = nullptr;   // FD cell of the atom at the last update_fd, packed i | j<<10 | k<<20 (-1: none yet);
                                         // travels with the atom like the reference's cell.fdi/fdj/fdk (imd_alloc.c:210,645)
  // double buffers for re-binning
  double4 *pos2 = nullptr; int *type2 = nullptr, *id2 = nullptr; double *p2 = nullptr, *mass2 = nullptr;
  int *nn2 = nullptr;
  int *keys = nullptr, *keys2 = nullptr, *perm = nullptr, *perm2 = nullptr;
  void *cub_tmp = nullptr; size_t cub_tmp_bytes = 0;

  // cells -----------------------------------------------------------------
  int *cell_start = nullptr, *cell_cnt = nullptr;   // [nall_cells], linear (ghost incl.) index
  int *icnt = nullptr, *istart = nullptr;           // [ninner_cells+1], inner rank order
  int *gcell_lin = nullptr, *gcell_src = nullptr, *gcell_shift = nullptr;  // [nghost_cells] static
  int *gcnt = nullptr, *gstart = nullptr;           // [nghost_cells+1]
  int *ghost_src = nullptr, *ghost_code = nullptr;  // [cap ghost]
  int cap_ghost = 0;
  int have_atoms = 0;                               // an upload has happened (nlocal may be 0: a rank in vacuum)
  int *d_counters = nullptr;                        // misc device ints (see kernels)
  int *h_counters = nullptr;                        // pinned mirror

  // neighbour list, full (symmetric).  mode 0: ELL transposed int32 global indices
  // nbr[k*stride+i]; mode 1: per tile, uint16 LDS-slot indices tl16[(tile*W+k)*tile_cap+a]
  int mode = 1, list_mode = 0, autotiled = 0;
  int force_external_halo = 0;
  int ttm_flux = 0;               // update_fd also counts the atoms that changed FD cells (imd_ttm_3D.c:335-370)
  int ttm_adv = 0;                // calc_ttm runs do_ADV after update_fd (imd_ttm_3D.c:57-60); implies ttm_flux
  int ttm_persist_opt = -1;       // option ttm_persist: the persistent sub-step kernel on (1) / off (0) / default (-1)
  int ttm_persist_miss = -1;      // option ttm_persist_miss (tests): the persistent launch of this MD step waits for one workgroup too many
  int ttm_decomposed = 0;         // several ranks: one TTM lattice block per rank, faces through the shared-memory mailbox
  int external_halo = 0;          // ghost atoms are supplied through imdgpu_halo_scatter (multi-rank decomposition)
  int *mig_idx[2]{nullptr, nullptr}; int mig_n[2]{0, 0}, mig_cap = 0; int *mig_flag = nullptr;
  int *nbr = nullptr, *nbr_cnt = nullptr;
  unsigned short *nbr16 = nullptr; size_t nbr16_cap = 0;
  int *tsrc = nullptr; unsigned *ownw = nullptr; int *tinfo = nullptr;   // staging plan of the tiles (k_nbl_build_tile)
  size_t tsrc_cap = 0, ownw_cap = 0, tinfo_cap = 0;
  int W = 0; size_t stride = 0; size_t nbr_cap = 0;
  int tile_cap = 0, max_halo = 0, slot_cap = 0, tile_threads = 0;
  size_t lds_need[4]{};
  int have_valid_nbl = 0, nbl_count = 0, need_fix_cells = 0;

  // reductions -----------------------------------------------------------------
  double *partials = nullptr; int npart_blocks = 0;
  double *d_sums = nullptr;        // [16]: 0..7 force sums, 8 ekin, 9 maxdisp2 bits
  double *h_sums = nullptr;        // pinned

  // profiling ------------------------------------------------------------------
  bool prof_on = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<ProfPending> ev_pending;
  long long prof_n[PROF_COUNT]{};
  double prof_ms[PROF_COUNT]{};

  // debugging aid: IMDGPU_TIMELINE=1 prints where each stage of a step ended on the device clock
  bool tl_on = false; std::vector<std::pair<const char *, hipEvent_t>> tl;

  // TTM state lives in imdgpu_ttm.hip
  struct TtmState *ttm = nullptr;
  int eos_bilinear = 0;              // option "eos_bilinear": plain bilinear EOS look-up instead of the reference's triangles
  void *ttm1d = nullptr;             // the reference's own 1D variant (imd_ttm.c), imdgpu_ttm1d_* (end of imdgpu_ttm.hip)
  // native multi-GPU step (imdgpu_dist.hip)
  // host-buffer boundary (level-1 integration): device-side reordering between the device's tile-major atom order and the
  // host's cell order, staged through one device buffer; the permutation is rebuilt when the cell tables change
  int *d_refperm = nullptr; size_t refperm_cap = 0; long long refperm_epoch = -1, cells_epoch = 0;
  void *d_stage = nullptr; size_t stage_bytes = 0;
  int reserve_cus = 0;
  int tile_balance = 0;             // set by make_nblist when some tiles are (nearly) empty: see block_tile_range
  long long *d_wprefix = nullptr; size_t wprefix_cap = 0;   // [ntiles + 1] cost prefix of the tiles (k_tile_wprefix)
  void *dist = nullptr;
  int step_open = 0;                  // imdgpu_md_step_begin has enqueued a step that imdgpu_md_step_end has not closed yet
  const int *pass1_guard = nullptr;   // device int: a pass-1 kernel launched AHEAD of the host's look at the list-validity flag returns at once when it is set
  // pass 1 under the position halo (option "halo_overlap", imdgpu_dist.hip): tiles from a work queue, interior ones first
  int halo_overlap = 0;               // option: 0 off, n > 0: on, the first launch leaves n CUs to the transport's kernels
  int *d_tile_map = nullptr; size_t tile_map_cap = 0; int dyn_ntiles = -1, n_interior = 0;   // queue order of the tiles (imdgpu_pass1_dyn_prepare)
  int *d_dyn = nullptr;               // [0..15] queue counters
  int dyn_seq = 0;                    // launch pairs so far
  int dyn_nb[2]{0, 0};                // workgroups of the two launches of the pass 1 now in d_partials (0, 0: the one-launch form)
};

int imdgpu_fail(imdgpu_ctx *c, const char *fmt, ...);

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return imdgpu_fail(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct ProfScope {
  imdgpu_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(imdgpu_ctx *ctx, int id_);
  ~ProfScope();
};
void imdgpu_prof_resolve(imdgpu_ctx *c);
void imdgpu_tl_mark(imdgpu_ctx *c, const char *name);     // no-op unless IMDGPU_TIMELINE is set
void imdgpu_tl_flush(imdgpu_ctx *c, int steps);            // prints and clears the marks of a step (synchronises the device)

template <typename T>
static inline int dev_alloc(imdgpu_ctx *ctx, T **p, size_t n)
{
  if (*p) { (void) hipFree(*p); *p = nullptr; }
  HIPCHK(hipMalloc((void **) p, (n ? n : 1) * sizeof(T)));
  return 0;
}

// ---------------------------------------------------------------------------
// geometry passed to kernels by value.  Local atoms are kept sorted by a
// TILE-MAJOR cell key: the inner cells are grouped into tiles of bx*by*bz cells
// (one workgroup of the force kernels per tile); key = tile*CPT + cell-in-tile,
// both z fastest.  Partial tiles at the upper faces leave some keys unused.
// ---------------------------------------------------------------------------
struct Geo {
  double box[9], tbox[9];
  int pbc[3], gdim[3], dim[3], my_coord[3];
  int nx, ny, nz;                   // inner cells per direction (dim-2)
  int bx, by, bz, ntx, nty, ntz, CPT, nkeys;
};

__host__ __device__ __forceinline__ int xyz_to_key(int ix, int iy, int iz, const Geo &g)
{
  int tx = ix / g.bx, ty = iy / g.by, tz = iz / g.bz;
  int lx = ix - tx * g.bx, ly = iy - ty * g.by, lz = iz - tz * g.bz;
  return ((tx * g.nty + ty) * g.ntz + tz) * g.CPT + (lx * g.by + ly) * g.bz + lz;
}
// returns false for keys of cells that do not exist (partial tiles)
__host__ __device__ __forceinline__ bool key_to_xyz(int key, const Geo &g, int &ix, int &iy, int &iz)
{
  int tile = key / g.CPT, cit = key - tile * g.CPT;
  int lz = cit % g.bz, q = cit / g.bz, ly = q % g.by, lx = q / g.by;
  int tz = tile % g.ntz, r = tile / g.ntz, ty = r % g.nty, tx = r / g.nty;
  ix = tx * g.bx + lx; iy = ty * g.by + ly; iz = tz * g.bz + lz;
  return ix < g.nx && iy < g.ny && iz < g.nz;
}
Geo imdgpu_make_geo(const imdgpu_ctx *c);

// device counters (ints) and d_sums slots shared by the translation units
enum { CNT_MAXNBR = 0, CNT_IS_SHORT = 1, CNT_OUTSIDE = 2, CNT_NGHOST = 3, CNT_INVALID = 4, CNT_MAXOWN = 5,
       CNT_MAXHALO = 6, CNT_TTM_TIMEOUT = 7, CNT_TMM_SHORT = 12, CNT_KNEG = 13, CNT_P3_STANDIN = 14, CNT_N = 16 };     // 8..10: scratch of the migration
// CNT_KNEG: INT_MAX - (lowest table row any pair of this step's pass 1 fell into); zeroed by k_reduce_force_sums
// CNT_P3_STANDIN: force passes in which k_pass3_tile stood in for k_pass3_tile_coef (a pair below the rows it holds); never zeroed
enum { SUM_EPOT = 0, SUM_VIRIAL = 1, SUM_EKIN = 8, SUM_MAXDISP = 9, SUM_N = 16 };

// sum of kinetic partials, max of displacement partials; decides list validity
// (max > (0.5*margin)^2  =>  invalid; imd_forces_nbl.c:1091).  One block.
[[maybe_unused]] static __global__ void k_reduce_move(int nb, const double *partials, double *sums, double limit2, int *counters,
                                     int write_ekin)
{
  __shared__ double scratch[2 * 16];
  double e = 0, mx = 0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) { e += partials[(size_t) b * 2]; mx = fmax(mx, partials[(size_t) b * 2 + 1]); }
  for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o, 64); mx = fmax(mx, __shfl_down(mx, o, 64)); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { scratch[w] = e; scratch[16 + w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double E = 0, M = 0;
    for (int q = 0; q < nw; q++) { E += scratch[q]; M = fmax(M, scratch[16 + q]); }
    if (write_ekin) sums[SUM_EKIN] = E;
    sums[SUM_MAXDISP] = M;
    counters[CNT_INVALID] = (M > limit2) ? 1 : 0;
  }
}

int imdgpu_ensure_capacity(imdgpu_ctx *ctx, int nlocal, int ntot);
int imdgpu_launch_forces_phase(imdgpu_ctx *ctx, int phase);
int imdgpu_launch_integrate(imdgpu_ctx *ctx);
// pass 1 in two launches from a tile queue: remote[d] != 0 = the ghost cells at the two ends of dimension d arrive through the
// transport (everything else is in place when the first launch starts).  prepare: once per list; ready: this context's pass 1 has
// the queue form; launch(part 0): the interior tiles; launch(part 1): the rest
int imdgpu_pass1_dyn_prepare(imdgpu_ctx *ctx, const int remote[3]);
int imdgpu_pass1_dyn_ready(const imdgpu_ctx *ctx);
int imdgpu_launch_pass1_dyn(imdgpu_ctx *ctx, int part);
int imdgpu_p3_coef_active(const imdgpu_ctx *ctx);
int imdgpu_ttm_chain_needs_lds(const imdgpu_ctx *ctx);

// TTM entry points called from imdgpu_md_step
int imdgpu_ttm_step_internal(imdgpu_ctx *ctx, int steps);
int imdgpu_ttm_move_internal(imdgpu_ctx *ctx);
void imdgpu_ttm_free(imdgpu_ctx *ctx);
void imdgpu_ttm1d_free(imdgpu_ctx *ctx);
int imdgpu_ttm1d_move_internal(imdgpu_ctx *ctx);
// sum over the ranks of n doubles in device memory, in place, on ctx->stream (the lattice stream when forked); nothing on one rank
int imdgpu_dist_sum_dev(imdgpu_ctx *ctx, double *d, int n);
void imdgpu_ttm_set_persist(imdgpu_ctx *ctx, int value);
int imdgpu_ttm_persist_settle(imdgpu_ctx *ctx);            // 0 nothing, 1 the stepped-aside launch's sub-steps were re-issued (launch the integrator again), -1 error
const int *imdgpu_ttm_persist_guard(imdgpu_ctx *ctx);      // status word a guarded integrator looks at, or null
long long imdgpu_ttm_persist_info(imdgpu_ctx *ctx, int which);
void imdgpu_dist_free(imdgpu_ctx *ctx);
long long imdgpu_dist_rccl_info(imdgpu_ctx *ctx, int which);
void imdgpu_dist_cancel_ahead(imdgpu_ctx *ctx);
