// imdgpu_ttm.hip -- 3D two-temperature model on the device (gfx950).
// Reference: imd_ttm_3D.c (update_fd, do_FILLMESH, do_DIFF, CFL_maxdt,
// ttm_fill_ghost_layers, do_cell_activation) and move_atoms_ttm
// (imd_integrate.c:581-811).  Public ABI: include/imdgpu.h.
//
// Lattice layout: SoA over local_fd_dim = (cell_dim-2)/fd_ext + 2 cells per
// direction including the ghost shell, C order [i][j][k] (k fastest) like the
// reference's l1[i][j][k].  The two lattices l1/l2 of the reference become
// ping-pong pairs for the fields do_DIFF double-buffers (temp, U) plus fd_k in
// the fused sub-step kernel; `cur` selects l1.
#include "imdgpu_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

struct EosDev { int nx, ny; double xmin, xmax, ymin, ymax; const double *z; const unsigned char *diag; };   // diag: see eos_lookup

// the reference's triangulation of a table grid (imdgpu_eos_triangulation, further down) on the device, attached to the table
extern "C" int imdgpu_eos_triangulation(int nx, int ny, unsigned char *diag);
static int eos_attach_triangulation(imdgpu_ctx *ctx, EosDev &e, unsigned char **store)
{
  std::vector<unsigned char> d((size_t) (e.nx - 1) * (e.ny - 1));
  if (e.nx < 2 || e.ny < 2 || imdgpu_eos_triangulation(e.nx, e.ny, d.data())) return imdgpu_fail(ctx, "EOS table: no triangulation for a %d x %d grid", e.nx, e.ny);
  if (*store) { (void) hipFree(*store); *store = nullptr; }
  HIPCHK(hipMalloc((void **) store, d.size()));
  HIPCHK(hipMemcpy(*store, d.data(), d.size(), hipMemcpyHostToDevice));
  e.diag = *store;
  return 0;
}

// descriptor of the shared-memory mailbox of a decomposed lattice (explained above the kernels that use it)
struct MailboxDev {
  char *base;                    // device-visible address of the shared region (all ranks)
  size_t stride;                 // bytes per rank
  size_t o_nat, o_dat;           // offsets inside a rank's part: natoms faces, data faces [2][6][maxface][2]
  int rank, nranks, maxface;
  int nbr[6];                    // neighbour rank per direction 2*d+side (side 0 = low), -1 = open boundary
  unsigned long long *gbase;     // device: index of the first publication of the current MD step
  int *counters;                 // ctx->d_counters
};

struct TtmState {
  int dim[3]{}, ext[3]{}, md_dim[3]{}, pbc[3]{};
  int ncell = 0, ninner = 0;
  int off[3]{}, ldim[3]{};          // this rank's block of the (replicated, global) lattice: offset and inner extent
  int goff[3]{};                    // GLOBAL FD index of this rank's first inner cell (= off, except on the decomposed lattice where off = 0)
  int fd_min_atoms = 0, nsub = 0, cur = 0, kcur = 0;
  double h[3]{}, vol = 0, neighvol = 0, atomic_weight = 0, atomic_charge = 0, omega_laser = 0, Temin = 1e-3;
  double max_dt_ttm = 0;
  int have_source = 0;              // a source term was uploaded (imdgpu_ttm_upload with `source`) or a laser is set: the sub-steps read it
  int laser_active = 0;             // laser_rescale_ttm's flag (source above 1e-3 of its peak): do_DIFF then books the absorbed energy
  int *d_cand = nullptr;            // do_cell_activation: [ninner] candidates, [ninner] sorted, [1] count
  double *d_eabs = nullptr;         // Eabs_local [eV] (imd_ttm_3D.c:1647-1651), device resident
  int laser_on = 0; imdgpu_laser_params laser{};   // imdgpu_ttm_set_laser: imdgpu_calc_ttm applies laser_rescale_ttm every step
  int *natoms = nullptr, *natoms_old = nullptr;
  int *flux = nullptr;              // 8 planes: atoms that entered the FD cell since the last update_fd (option ttm_flux)
  double *temp[2]{}, *U[2]{}, *fd_k[2]{};
  double *xi = nullptr, *md_temp = nullptr, *source = nullptr, *vcom = nullptr /* 3 planes */, *dens = nullptr,
         *fd_g = nullptr, *Z = nullptr, *Ce = nullptr, *ne = nullptr;
  double *cve_z = nullptr, *ee_z = nullptr;
  unsigned char *eos_diag[2] = {nullptr, nullptr};   // triangulation maps of the two EOS tables (imdgpu_ttm_set_eos_triangulation)
  EosDev cve{}, ee{};
  double *d_red = nullptr;       // reduction scratch
  // one captured sub-step chain per value of `cur` at its start (do_ADV's extra swap alternates it from step to step)
  hipGraphExec_t graph[2] = {nullptr, nullptr}; int graph_kcur[2] = {-1, -1}, graph_src[2] = {-1, -1}; double graph_tau[2] = {0, 0};
  // decomposed lattice (several ranks): faces travel through a host shared-memory mailbox (see Mailbox below)
  void *mb_host = nullptr; size_t mb_bytes = 0; int mb_rank = -1, mb_nranks = 0; bool mb_registered = false;
  MailboxDev mbd{};
  unsigned long long *d_gbase = nullptr; // index of the next face publication (monotone over the whole run)
  int *d_facecnt = nullptr;              // per face: workgroups that finished publishing (monotone)
  struct FdtdState *fdtd = nullptr;      // 1D Maxwell solver on a one-row lattice (imdgpu_ttm_fdtd_*, end of this file)
  int fdtd_on = 0;
  // persistent sub-step kernel (k_ttm_chain_persist): 0 = off, 1 = trial (the captured launch chain follows every launch,
  // guarded by the kernel's status word), 2 = trusted (launch chain dropped)
  int persist = -1, persist_ok = 0, persist_tiles = 0, pstat_pending = 0;
  int pstat_mode = 0;                    // mode of the launch whose status is pending: 1 trial, 2 trusted, 3 trusted with remote records
  int persist_miss_run = 0, persist_redone = 0;   // launches in a row that stepped aside; steps re-issued through the launch chain
  int *d_pbbox = nullptr;                // dim[0] + dim[1] + dim[2] marks of the active cells' extent (k_ttm_fillmesh -> k_ttm_chain_persist)
  unsigned *d_pflags = nullptr;          // census word, gate word
  unsigned long long *d_prec = nullptr;  // hand-off records [2][ncell][4]
  unsigned persist_base = 0;             // tag of state 0 of the next launch
  int *d_pstat = nullptr, *h_pstat = nullptr;   // status of the last launch (device word: the four ints behind d_pflags' sixteen / pinned copy)
  hipEvent_t ev_pstat = nullptr;
  const int *graph_guard[2] = {nullptr, nullptr};
  double *d_ufd_part = nullptr; size_t ufd_part_cap = 0;   // partial sums of the chunked update_fd
  long long fdc_epoch = -1;         // ctx->cells_epoch at the last update_fd that did the flux bookkeeping (imdgpu_ttm_update_fd)
  int flux_zero = 0;                // the flux counters are known to be all zero
  // decomposed lattice, faces through ghost RECORDS in the neighbours' device memory (imdgpu_ttm_remote_*)
  unsigned long long *d_xrec = nullptr; size_t xrec_bytes = 0; int xoff[6]{}, xcells = 0; bool xrec_fine = false;
  unsigned long long *peer[6]{}; int xremote = 0; bool xrec_ready = false;
};

// physical constants of the reference, globals.h:324-331,1190
#define C_ECONST 8.841941282883075e-12
#define C_HBAR 1.054571800139113e-34
#define C_EMASS 9.109383560000000e-31
#define C_BOLTZMAN 1.380648520000000e-23
#define C_ECHARGE 1.602176620000000e-19
#define C_AMU 1.660539040200000e-27
#define C_J2EV 6.2415090744607626077762409809304458998869658961709711e18
#define C_PI 3.14159265358979323846

struct Lat {                      // lattice geometry + parameters by value
  int dim[3], ext[3], md_dim[3], pbc[3], off[3], ldim[3], goff[3];
  int fd_min_atoms;
  double h[3], vol, neighvol, atomic_weight, atomic_charge, omega_laser, Temin;
  // uniform reciprocals, evaluated once on the host (an fp64 division is ~20 VALU instructions on the device)
  double inv_z0_43;               // 1e-3 / atomic_charge^(4/3)        (MeanCharge t0)
  double inv_z0am;                // 1e-3 / (atomic_charge*atomic_weight) (MeanCharge r0)
  double inv_awamu;               // 1 / (atomic_weight * AMU)          (ne)
  double invhsq[3];               // 1 / fd_h^2
  double xi_fac;                  // 26.9815*AMU/3*1e30 / diff_substeps  (do_DIFF :1614)
  int wave_prio;                  // wave priority of the sub-step kernel (IMDGPU_TTM_PRIO, default 0)
};

__device__ __forceinline__ int fidx(const Lat &L, int i, int j, int k) { return (i * L.dim[1] + j) * L.dim[2] + k; }

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
}
__device__ __forceinline__ int wsumi(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
}

// ---------------------------------------------------------------------------
// Decomposed lattice (one block per rank): the ghost faces of {natoms, temp, fd_k} that the reference exchanges with
// MPI_Sendrecv after every diffusion sub-step (ttm_fill_ghost_layers, imd_ttm_3D.c:3314-3504) travel through a
// MAILBOX in host shared memory that the GPUs of all ranks of the node read and write directly: after every sub-step
// a six-wave kernel of rank A stores A's boundary layers into the slots of its neighbours and raises their flags;
// before the next sub-step a six-wave kernel of rank B polls B's flags and copies the slots into B's ghost cells.
// No collective and no CU held by a communication kernel (RCCL's workgroups cannot be co-resident with the persistent
// force kernels); the data is ~150 KB per sub-step, so PCIe-attached host memory is plenty.
// Publications are numbered by one monotone index; two slot parities make a writer that is one sub-step ahead
// harmless.  Every spin is bounded.
// ---------------------------------------------------------------------------
#define MB_SPIN_MAX 2000000L
// All mailbox traffic uses RELAXED system-scope accesses (cache-bypassing loads/stores of fine-grained host memory)
// and, on the writer, an AGENT-scope release (wait for the stores' acknowledgements) before the flag goes out.  A
// system-scope fence or acquire would write back / invalidate the whole L2 every time -- measured: 158 us per
// sub-step and the concurrent force pass 2.5x slower.  The reader needs no fence: it branches on the flag value
// before it issues the data loads, and those bypass the caches.
__device__ __forceinline__ unsigned long long mb_ld(const unsigned long long *p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mb_st(unsigned long long *p, unsigned long long v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mb_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ unsigned long long *mb_flag(const MailboxDev &M, int rank, int dir)
{ return (unsigned long long *) (M.base + (size_t) rank * M.stride) + dir; }
__device__ __forceinline__ unsigned long long *mb_abort(const MailboxDev &M) { return (unsigned long long *) (M.base + 64); }
__device__ __forceinline__ double *mb_dat(const MailboxDev &M, int rank, int parity, int dir, int idx)
{ return (double *) (M.base + (size_t) rank * M.stride + M.o_dat) + (((size_t) parity * 6 + dir) * M.maxface + idx) * 2; }
__device__ __forceinline__ double mb_ldd(const double *p)      // data written by another GPU: never from a cache
{ return __longlong_as_double((long long) __hip_atomic_load((const unsigned long long *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
__device__ __forceinline__ int mb_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ int *mb_nat(const MailboxDev &M, int rank, int dir, int idx)
{ return (int *) (M.base + (size_t) rank * M.stride + M.o_nat) + (size_t) dir * M.maxface + idx; }
// wait until my flag for direction dir shows `need` publications; false on abort/time-out (results are then invalid,
// CNT_TTM_TIMEOUT is set and reported by the host)
__device__ bool mb_wait(const MailboxDev &M, int dir, unsigned long long need)
{
  const unsigned long long *f = mb_flag(M, M.rank, dir);
  for (long it = 0; it < MB_SPIN_MAX; it++) {
    if (mb_ld(f) >= need) return true;
    if (mb_ld(mb_abort(M))) return false;
    __builtin_amdgcn_s_sleep(16);
  }
  mb_st(mb_abort(M), 1ull);
  M.counters[CNT_TTM_TIMEOUT] = 1;
  return false;
}
// publish my six boundary layers as publication `*gbase + off` (off = 0: after FILLMESH, with natoms; off = s + 1: after
// sub-step s).  ONE wave per face (six pollers / six flag writers in total: a flood of PCIe polls from many
// workgroups was measured to cost more than the exchange itself); the face is walked with eight independent
// accesses in flight per lane.
#define MB_UNR 8
__global__ __launch_bounds__(64) void k_ttm_push_faces(Lat L, MailboxDev M, int off, int with_nat, const int *natoms,
                                                        const double *temp, const double *fd_k)
{
  const int b = blockIdx.x, d = b >> 1, side = b & 1, lane = threadIdx.x;
  const int d1 = (d + 1) % 3, d2 = (d + 2) % 3, n1 = L.dim[d1] - 2, n2 = L.dim[d2] - 2, nf = n1 * n2;
  if (M.nbr[b] < 0) return;
  const unsigned long long g = *M.gbase + (unsigned long long) off;
  // first publication of an MD step: the neighbour must have taken my previous step's last publication (it says so
  // with its own last publication) before its slot is overwritten; later ones are ordered by the pull before them
  int ok = 1;
  if (off == 0 && g > 0) { if (lane == 0) ok = mb_wait(M, b, g) ? 1 : 0; ok = __shfl(ok, 0, 64); }
  if (!ok) return;
  double2 *o = (double2 *) mb_dat(M, M.nbr[b], (int) (g & 1), b ^ 1, 0);
  int *on = mb_nat(M, M.nbr[b], b ^ 1, 0);
  for (int base = 0; base < nf; base += 64 * MB_UNR) {
    double2 v[MB_UNR]; int nn[MB_UNR];
#pragma unroll
    for (int u = 0; u < MB_UNR; u++) {
      const int idx = base + u * 64 + lane;
      if (idx < nf) {
        int c[3];
        c[d] = side ? L.dim[d] - 2 : 1; c[d1] = idx / n2 + 1; c[d2] = idx % n2 + 1;
        const int cell = fidx(L, c[0], c[1], c[2]);
        v[u] = make_double2(temp[cell], fd_k[cell]);
        if (with_nat) nn[u] = natoms[cell];
      }
    }
#pragma unroll
    for (int u = 0; u < MB_UNR; u++) {
      const int idx = base + u * 64 + lane;
      if (idx < nf) {           // payload: system-scope (cache-bypassing) atomic stores, like every other mailbox access
        mb_st((unsigned long long *) &o[idx].x, (unsigned long long) __double_as_longlong(v[u].x));
        mb_st((unsigned long long *) &o[idx].y, (unsigned long long) __double_as_longlong(v[u].y));
        if (with_nat) __hip_atomic_store(on + idx, nn[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  // every lane's stores are acknowledged (vmcnt 0: inline asm, the compiler may not drop it -- MI355X_MICROARCH.md,
  // "Compiler hazard") before the one wave of this face lets the flag go out
  mb_release();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) mb_st(mb_flag(M, M.nbr[b], b ^ 1), g + 1);
}
// take the neighbours' publication `*gbase + off` into the ghost cells of the arrays (natoms only with the first of a step)
__global__ __launch_bounds__(64) void k_ttm_pull_ghosts(Lat L, MailboxDev M, int off, int with_nat, int *natoms, double *temp,
                                                         double *fd_k)
{
  const int b = blockIdx.x, d = b >> 1, side = b & 1, lane = threadIdx.x;
  const int d1 = (d + 1) % 3, d2 = (d + 2) % 3, n1 = L.dim[d1] - 2, n2 = L.dim[d2] - 2, nf = n1 * n2;
  const unsigned long long g = *M.gbase + (unsigned long long) off;
  if (M.nbr[b] < 0) {                                // open boundary: the ghost layer is empty (reflecting)
    if (with_nat)
      for (int idx = lane; idx < nf; idx += 64) {
        int c[3];
        c[d] = side ? L.dim[d] - 1 : 0; c[d1] = idx / n2 + 1; c[d2] = idx % n2 + 1;
        natoms[fidx(L, c[0], c[1], c[2])] = 0;
      }
    return;
  }
  int ok = 0;
  if (lane == 0) ok = mb_wait(M, b, g + 1) ? 1 : 0;
  ok = __shfl(ok, 0, 64);
  if (!ok) return;
  const double *q = mb_dat(M, M.rank, (int) (g & 1), b, 0);
  const int *qn = mb_nat(M, M.rank, b, 0);
  for (int base = 0; base < nf; base += 64 * MB_UNR) {
    double v0[MB_UNR], v1[MB_UNR]; int nn[MB_UNR];
#pragma unroll
    for (int u = 0; u < MB_UNR; u++) {
      const int idx = base + u * 64 + lane;
      if (idx < nf) { v0[u] = mb_ldd(q + 2 * idx); v1[u] = mb_ldd(q + 2 * idx + 1); if (with_nat) nn[u] = mb_ldi(qn + idx); }
    }
#pragma unroll
    for (int u = 0; u < MB_UNR; u++) {
      const int idx = base + u * 64 + lane;
      if (idx < nf) {
        int c[3];
        c[d] = side ? L.dim[d] - 1 : 0; c[d1] = idx / n2 + 1; c[d2] = idx % n2 + 1;
        const int cell = fidx(L, c[0], c[1], c[2]);
        temp[cell] = v0[u]; fd_k[cell] = v1[u];
        if (with_nat) natoms[cell] = nn[u];
      }
    }
  }
}
__global__ void k_ttm_advance_base(unsigned long long *gbase, int nsub) { *gbase += (unsigned long long) nsub + 1ull; }

// ---------------------------------------------------------------------------
// update_fd: one wavefront per FD cell gathers its fd_ext^3 MD cells
// (imd_ttm_3D.c:233-488).  First sweep: natoms, sum numneighs, sum p, sum m;
// second sweep: sum m*(p/m - vcom)^2.  __shfl reductions, no atomics.
// ---------------------------------------------------------------------------
// One rank: results go straight into the lattice.  Several ranks (lattice replicated on every rank, each
// rank owns the block of FD cells over its MD cells): `blockbuf` != nullptr receives 6 doubles per local
// FD cell {natoms, md_temp, dens, vcomx, vcomy, vcomz}; k_ttm_apply_blocks folds all ranks' blocks in.
__global__ __launch_bounds__(64) void k_ttm_update_fd(Lat L, int steps, int cap, const int *cell_start, const int *cell_cnt,      // launched with TTM_WG = 64 threads
                                const int *numneighs, const double *p, const double *mass, int *natoms,
                                int *natoms_old, double *temp0, double *temp1, double *xi, double *md_temp,
                                double *source, double *vcom, double *dens, int ncell, double *blockbuf, int *fdc, int *flux)
{
  // fdc/flux != nullptr: also the atom-flux bookkeeping of the advection step (imd_ttm_3D.c:335-370): every atom
  // remembers the global FD cell it was counted in; an atom found in another (i, j) column than last time is counted
  // by the direction it came from (0: +x, 1: -x, 2: +y, 3: +x-y, 4: +x+y, 5: -y, 6: -x+y, 7: -x-y).
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int ni = L.ldim[0], nj = L.ldim[1], nk = L.ldim[2];
  if (w >= ni * nj * nk) return;
  const int k = w % nk + 1, j = (w / nk) % nj + 1, i = w / (nk * nj) + 1;     // local FD coordinates (1-based)
  const int c = fidx(L, i + L.off[0], j + L.off[1], k + L.off[2]);
  int nat = 0, neigh = 0;
  int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double sx = 0, sy = 0, sz = 0, sm = 0;
  // Up to eight MD cells of at most 64 atoms per FD cell (fd_ext 2 2 2 and smaller: the usual case): every lane meets
  // at most one atom per MD cell and keeps its momentum and mass in registers for the second sweep, which then reads
  // nothing -- the kernel is bound by exactly those loads.  Same operations in the same order either way.
  const int e0 = L.ext[0], e1 = L.ext[1], e2 = L.ext[2], ne = e0 * e1 * e2;
  const bool small = ne <= 8;
  bool cached = small;
  double cx[8], cy[8], cz[8], cm[8];
  // one MD cell: first-sweep sums (+ the flux bookkeeping); returns this lane's first atom through (px, py, pz, pm)
  auto sweep1 = [&](int xc, int yc, int zc, double &px, double &py, double &pz, double &pm) {
    const int mc = (xc * L.md_dim[1] + yc) * L.md_dim[2] + zc;
    const int s = cell_start[mc], n = cell_cnt[mc];
    nat += n;
    if (n > 64) cached = false;
    for (int a = s + lane; a < s + n; a += 64) {
      neigh += numneighs[a];
      const double ax = p[a], ay = p[cap + a], az = p[2 * (size_t) cap + a], am = mass[a];
      sx += ax; sy += ay; sz += az;
      sm += am;
      if (a == s + lane) { px = ax; py = ay; pz = az; pm = am; }
      if (fdc) {
        const int ig = i + L.goff[0] - 1, jg = j + L.goff[1] - 1, kg = k + L.goff[2] - 1, old = fdc[a];
        if (steps > 0 && old >= 0) {
          const int di = old & 1023, dj = (old >> 10) & 1023;
          if (di != ig || dj != jg) {
            const int q = di > ig ? (dj > jg ? 4 : (dj < jg ? 3 : 0)) : (di < ig ? (dj < jg ? 7 : (dj > jg ? 6 : 1)) : (dj > jg ? 2 : 5));
            fl[q]++;
          }
        }
        fdc[a] = ig | (jg << 10) | (kg << 20);
      }
    }
  };
  int cs[8], cn[8];
  if (small) {                                   // the cells' (start, count) first: eight independent loads
#pragma unroll
    for (int q = 0; q < 8; q++) {
      cs[q] = 0; cn[q] = 0; cx[q] = 0.0; cy[q] = 0.0; cz[q] = 0.0; cm[q] = 0.0;
      if (q < ne) {
        const int xc = (i - 1) * e0 + 1 + q / (e2 * e1), yc = (j - 1) * e1 + 1 + (q / e2) % e1, zc = (k - 1) * e2 + 1 + q % e2;
        const int mc = (xc * L.md_dim[1] + yc) * L.md_dim[2] + zc;
        cs[q] = cell_start[mc]; cn[q] = cell_cnt[mc];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) if (cn[q] > 64) cached = false;
  }
  // The usual fd_ext 2 2 2 on a list whose tiles are even along z: the MD cells 2r and 2r+1 of the FD cell are neighbours
  // along z, i.e. consecutive in the atom arrays, and hold ~42 atoms together: four runs of up to 64 atoms instead of eight
  // cells of ~21 -- half the load instructions, two thirds of the lanes busy instead of one third.  (Lane sums are taken over
  // other atoms than in the eight-cell walk: md_temp and v_com differ from it at rounding level, integers not at all.)
  bool paired = cached && ne == 8;
  if (paired) {
#pragma unroll
    for (int r = 0; r < 4; r++) if (cs[2 * r + 1] != cs[2 * r] + cn[2 * r] || cn[2 * r] + cn[2 * r + 1] > 64) paired = false;
  }
  if (paired) {
    int nn[4], old[4], rn[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      nn[r] = 0; old[r] = -1; rn[r] = cn[2 * r] + cn[2 * r + 1];
      if (lane < rn[r]) {
        const int a = cs[2 * r] + lane;
        nn[r] = numneighs[a];
        cx[r] = p[a]; cy[r] = p[cap + a]; cz[r] = p[2 * (size_t) cap + a]; cm[r] = mass[a];
        if (fdc) old[r] = fdc[a];
      }
    }
    const int ig = i + L.goff[0] - 1, jg = j + L.goff[1] - 1, kg = k + L.goff[2] - 1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      nat += rn[r];
      if (lane < rn[r]) {
        neigh += nn[r];
        sx += cx[r]; sy += cy[r]; sz += cz[r];
        sm += cm[r];
        if (fdc) {
          if (steps > 0 && old[r] >= 0) {
            const int di = old[r] & 1023, dj = (old[r] >> 10) & 1023;
            if (di != ig || dj != jg) {
              const int w8 = di > ig ? (dj > jg ? 4 : (dj < jg ? 3 : 0)) : (di < ig ? (dj < jg ? 7 : (dj > jg ? 6 : 1)) : (dj > jg ? 2 : 5));
#pragma unroll
              for (int u = 0; u < 8; u++) fl[u] += (u == w8) ? 1 : 0;
            }
          }
          fdc[cs[2 * r] + lane] = ig | (jg << 10) | (kg << 20);
        }
      }
    }
  } else if (cached) {
    // straight-line: all loads of the (at most eight) atoms this lane meets are issued before the first use
    int nn[8], old[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      nn[q] = 0; old[q] = -1;
      if (lane < cn[q]) {
        const int a = cs[q] + lane;
        nn[q] = numneighs[a];
        cx[q] = p[a]; cy[q] = p[cap + a]; cz[q] = p[2 * (size_t) cap + a]; cm[q] = mass[a];
        if (fdc) old[q] = fdc[a];
      }
    }
    const int ig = i + L.goff[0] - 1, jg = j + L.goff[1] - 1, kg = k + L.goff[2] - 1;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      nat += cn[q];
      if (lane < cn[q]) {
        neigh += nn[q];
        sx += cx[q]; sy += cy[q]; sz += cz[q];
        sm += cm[q];
        if (fdc) {
          if (steps > 0 && old[q] >= 0) {
            const int di = old[q] & 1023, dj = (old[q] >> 10) & 1023;
            if (di != ig || dj != jg) {
              const int w8 = di > ig ? (dj > jg ? 4 : (dj < jg ? 3 : 0)) : (di < ig ? (dj < jg ? 7 : (dj > jg ? 6 : 1)) : (dj > jg ? 2 : 5));
#pragma unroll
              for (int u = 0; u < 8; u++) fl[u] += (u == w8) ? 1 : 0;
            }
          }
          fdc[cs[q] + lane] = ig | (jg << 10) | (kg << 20);
        }
      }
    }
  } else if (small) {
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (q < ne) sweep1((i - 1) * e0 + 1 + q / (e2 * e1), (j - 1) * e1 + 1 + (q / e2) % e1, (k - 1) * e2 + 1 + q % e2,
                         cx[q], cy[q], cz[q], cm[q]);
  } else {
    double d0, d1, d2, d3;
    for (int xc = (i - 1) * e0 + 1; xc < i * e0 + 1; xc++)
      for (int yc = (j - 1) * e1 + 1; yc < j * e1 + 1; yc++)
        for (int zc = (k - 1) * e2 + 1; zc < k * e2 + 1; zc++) sweep1(xc, yc, zc, d0, d1, d2, d3);
  }
  neigh = wsumi(neigh);
  if (fdc) {
    // atoms change FD columns rarely: the eight wave reductions only when some lane of this wave saw a crossing
    int anyf = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) anyf |= fl[q];
    const bool crossing = __any(anyf != 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (crossing) fl[q] = wsumi(fl[q]);
      if (lane == 0 && flux) flux[(size_t) q * ncell + c] = crossing ? fl[q] : 0;
    }
  }
  sx = wsum(sx); sy = wsum(sy); sz = wsum(sz); sm = wsum(sm);
  double vx = 0, vy = 0, vz = 0, d = 0;
  if (nat > 0) {
    d = (double) neigh / ((double) nat) * L.atomic_weight / L.neighvol * 1660.53907;
    if (d == 0) d = (double) nat * L.atomic_weight / L.vol * 1660.53907;
    vx = sx / sm; vy = sy / sm; vz = sz / sm;
  }
  double mt = 0;
  if (cached) {
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (cm[q] != 0.0) {                        // this lane met an atom in MD cell q (masses are positive)
        const double m = cm[q];
        double u;
        u = cx[q] / m - vx; mt += m * (u * u);
        u = cy[q] / m - vy; mt += m * (u * u);
        u = cz[q] / m - vz; mt += m * (u * u);
      }
  } else
  for (int xc = (i - 1) * L.ext[0] + 1; xc < i * L.ext[0] + 1; xc++)
    for (int yc = (j - 1) * L.ext[1] + 1; yc < j * L.ext[1] + 1; yc++)
      for (int zc = (k - 1) * L.ext[2] + 1; zc < k * L.ext[2] + 1; zc++) {
        int mc = (xc * L.md_dim[1] + yc) * L.md_dim[2] + zc;
        int s = cell_start[mc], n = cell_cnt[mc];
        for (int a = s + lane; a < s + n; a += 64) {
          double m = mass[a], q;
          q = p[a] / m - vx; mt += m * (q * q);
          q = p[cap + a] / m - vy; mt += m * (q * q);
          q = p[2 * (size_t) cap + a] / m - vz; mt += m * (q * q);
        }
      }
  mt = wsum(mt);
  if (lane == 0 && blockbuf) {
    if (nat >= L.fd_min_atoms) mt /= 3.0 * nat; else mt = 0.0;
    double *o = blockbuf + (size_t) w * (fdc ? 10 : 6);
    o[0] = (double) nat; o[1] = mt; o[2] = d; o[3] = vx; o[4] = vy; o[5] = vz;
    if (fdc) {                                  // the eight flux counters, two per double
      int *oi = (int *) (o + 6);
#pragma unroll
      for (int q = 0; q < 8; q++) oi[q] = fl[q];
    }
  } else if (lane == 0) {
    if (nat >= L.fd_min_atoms) mt /= 3.0 * nat; else mt = 0.0;
    int old = natoms[c];
    natoms_old[c] = (steps < 1) ? nat : old;
    natoms[c] = nat;
    xi[c] = 0.0; source[c] = 0.0;
    dens[c] = d; md_temp[c] = mt;
    vcom[c] = vx; vcom[ncell + c] = vy; vcom[2 * ncell + c] = vz;
    if (steps < 1 && nat >= L.fd_min_atoms) { temp0[c] = mt; temp1[c] = mt; }
  }
}

// update_fd for FD cells spanning MANY MD cells (fd_ext product > 64, e.g. the one-row lattice of the Maxwell solver:
// 1 x 78 x 78 MD cells = 130 000 atoms per FD cell): one 1024-thread workgroup per FD cell, a WAVE per MD cell (strided over
// the cells, lanes over the cell's atoms: coalesced), block-wide sums through LDS in a fixed order (deterministic; the order
// differs from the reference's sequential loop at rounding level, integers are exact).  No flux bookkeeping here.
__global__ __launch_bounds__(1024) void k_ttm_update_fd_wide(Lat L, int steps, int cap, const int *cell_start, const int *cell_cnt,
                                const int *numneighs, const double *p, const double *mass, int *natoms,
                                int *natoms_old, double *temp0, double *temp1, double *xi, double *md_temp,
                                double *source, double *vcom, double *dens, int ncell)
{
  __shared__ double sd[5][16];
  __shared__ int si[2][16];
  __shared__ double bc[4];
  const int w = blockIdx.x, ni = L.ldim[0], nj = L.ldim[1], nk = L.ldim[2];
  if (w >= ni * nj * nk) return;
  const int k = w % nk + 1, j = (w / nk) % nj + 1, i = w / (nk * nj) + 1;
  const int c = fidx(L, i + L.off[0], j + L.off[1], k + L.off[2]);
  const int e0 = L.ext[0], e1 = L.ext[1], e2 = L.ext[2], ne = e0 * e1 * e2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  auto md_cell = [&](int q) {
    const int xc = (i - 1) * e0 + 1 + q / (e2 * e1), yc = (j - 1) * e1 + 1 + (q / e2) % e1, zc = (k - 1) * e2 + 1 + q % e2;
    return (xc * L.md_dim[1] + yc) * L.md_dim[2] + zc;
  };
  int nat = 0, neigh = 0;
  double sx = 0, sy = 0, sz = 0, sm = 0;
  for (int q0 = wv; q0 < ne; q0 += 2 * nw) {                     // two MD cells in flight per wave
    const int q1 = q0 + nw;
    const int m0 = md_cell(q0), s0 = cell_start[m0], n0 = cell_cnt[m0];
    int s1 = 0, n1 = 0;
    if (q1 < ne) { const int m1 = md_cell(q1); s1 = cell_start[m1]; n1 = cell_cnt[m1]; }
    if (lane == 0) nat += n0 + n1;
    for (int a = lane; a < n0 || a < n1; a += 64) {
      if (a < n0) { const int g = s0 + a; neigh += numneighs[g]; sx += p[g]; sy += p[cap + g]; sz += p[2 * (size_t) cap + g]; sm += mass[g]; }
      if (a < n1) { const int g = s1 + a; neigh += numneighs[g]; sx += p[g]; sy += p[cap + g]; sz += p[2 * (size_t) cap + g]; sm += mass[g]; }
    }
  }
  nat = wsumi(nat); neigh = wsumi(neigh); sx = wsum(sx); sy = wsum(sy); sz = wsum(sz); sm = wsum(sm);
  if (lane == 0) { si[0][wv] = nat; si[1][wv] = neigh; sd[0][wv] = sx; sd[1][wv] = sy; sd[2][wv] = sz; sd[3][wv] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int N = 0, NN = 0; double X = 0, Y = 0, Z = 0, M = 0;
    for (int q = 0; q < nw; q++) { N += si[0][q]; NN += si[1][q]; X += sd[0][q]; Y += sd[1][q]; Z += sd[2][q]; M += sd[3][q]; }
    double d = 0, vx = 0, vy = 0, vz = 0;
    if (N > 0) {
      d = (double) NN / ((double) N) * L.atomic_weight / L.neighvol * 1660.53907;
      if (d == 0) d = (double) N * L.atomic_weight / L.vol * 1660.53907;
      vx = X / M; vy = Y / M; vz = Z / M;
    }
    si[0][0] = N; bc[0] = d; bc[1] = vx; bc[2] = vy; bc[3] = vz;
  }
  __syncthreads();
  nat = si[0][0];
  const double d = bc[0], vx = bc[1], vy = bc[2], vz = bc[3];
  double mt = 0;
  for (int q0 = wv; q0 < ne; q0 += 2 * nw) {
    const int q1 = q0 + nw;
    const int m0 = md_cell(q0), s0 = cell_start[m0], n0 = cell_cnt[m0];
    int s1 = 0, n1 = 0;
    if (q1 < ne) { const int m1 = md_cell(q1); s1 = cell_start[m1]; n1 = cell_cnt[m1]; }
    for (int a = lane; a < n0 || a < n1; a += 64) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int n = h ? n1 : n0, g = (h ? s1 : s0) + a;
        if (a < n) {
          const double m = mass[g];
          double u;
          u = p[g] / m - vx; mt += m * (u * u);
          u = p[cap + g] / m - vy; mt += m * (u * u);
          u = p[2 * (size_t) cap + g] / m - vz; mt += m * (u * u);
        }
      }
    }
  }
  mt = wsum(mt);
  if (lane == 0) sd[4][wv] = mt;
  __syncthreads();
  if (threadIdx.x == 0) {
    double T = 0;
    for (int q = 0; q < nw; q++) T += sd[4][q];
    if (nat >= L.fd_min_atoms) T /= 3.0 * nat; else T = 0.0;
    const int old = natoms[c];
    natoms_old[c] = (steps < 1) ? nat : old;
    natoms[c] = nat;
    xi[c] = 0.0; source[c] = 0.0;
    dens[c] = d; md_temp[c] = T;
    vcom[c] = vx; vcom[ncell + c] = vy; vcom[2 * ncell + c] = vz;
    if (steps < 1 && nat >= L.fd_min_atoms) { temp0[c] = T; temp1[c] = T; }
  }
}

__global__ void k_ttm_apply_blocks(Lat L, int steps, int nblocks, const int *offs /* 3 per block */, int bi, int bj, int bk,
                                   const double *in, int *natoms, int *natoms_old, double *temp0, double *temp1,
                                   double *xi, double *md_temp, double *source, double *vcom, double *dens, int ncell,
                                   int *flux /* nullptr: 6 doubles per cell, else 10 */)
{
  const int per = bi * bj * bk, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nblocks * per) return;
  const int b = t / per, w = t - b * per;
  const int k = w % bk + 1, j = (w / bk) % bj + 1, i = w / (bk * bj) + 1;
  const int c = fidx(L, i + offs[3 * b], j + offs[3 * b + 1], k + offs[3 * b + 2]);
  const double *o = in + (size_t) t * (flux ? 10 : 6);
  if (flux) {
    const int *oi = (const int *) (o + 6);
#pragma unroll
    for (int q = 0; q < 8; q++) flux[(size_t) q * ncell + c] = oi[q];
  }
  const int nat = (int) o[0], old = natoms[c];
  natoms_old[c] = (steps < 1) ? nat : old;
  natoms[c] = nat;
  xi[c] = 0.0; source[c] = 0.0;
  md_temp[c] = o[1]; dens[c] = o[2];
  vcom[c] = o[3]; vcom[ncell + c] = o[4]; vcom[2 * ncell + c] = o[5];
  if (steps < 1 && nat >= L.fd_min_atoms) { temp0[c] = o[1]; temp1[c] = o[1]; }
}

// ---------------------------------------------------------------------------
// wide-range model + EOS (do_FILLMESH and helpers, imd_ttm_3D.c:492-661,
// :2352-2592, :2694-2783)
// ---------------------------------------------------------------------------
// Table look-up.  The reference interpolates LINEARLY ON A TRIANGULATION of the table's nodes (nn-c lpi_interpolate_point,
// imd_ttm_3D.c:2722,2759); which diagonal splits a grid cell is a property of its triangulator.  With a
// triangulation map (e.diag, one byte per grid cell: 1 = diagonal (i+1,j)-(i,j+1), 2 = diagonal (i,j)-(i+1,j+1); derived by
// imdgpu_eos_triangulation at init -- the default --, or read off the reference's own interpolator by the level-1 shim and
// handed over, imdgpu_ttm_set_eos_triangulation) the look-up is that triangle's plane; without one (0) it is bilinear, which
// agrees at nodes and on cell edges and differs inside a cell by at most a quarter of the cell's twist.
__device__ double eos_lookup(const EosDev &e, double x, double y)
{
  double fx = (x - e.xmin) / (e.xmax - e.xmin) * (e.nx - 1);
  double fy = (y - e.ymin) / (e.ymax - e.ymin) * (e.ny - 1);
  if (fx < 0) fx = 0;
  if (fx > e.nx - 1) fx = e.nx - 1;
  if (fy < 0) fy = 0;
  if (fy > e.ny - 1) fy = e.ny - 1;
  int ix = (int) fx, iy = (int) fy;
  if (ix > e.nx - 2) ix = e.nx - 2;
  if (iy > e.ny - 2) iy = e.ny - 2;
  double tx = fx - ix, ty = fy - iy;
  double z00 = e.z[(size_t) ix * e.ny + iy], z01 = e.z[(size_t) ix * e.ny + iy + 1];
  double z10 = e.z[(size_t) (ix + 1) * e.ny + iy], z11 = e.z[(size_t) (ix + 1) * e.ny + iy + 1];
  const int dg = e.diag ? e.diag[(size_t) ix * (e.ny - 1) + iy] : 0;
  if (dg == 1) return tx + ty <= 1.0 ? z00 + tx * (z10 - z00) + ty * (z01 - z00) : z11 + (1.0 - tx) * (z01 - z11) + (1.0 - ty) * (z10 - z11);
  if (dg == 2) return tx >= ty ? z00 + tx * (z10 - z00) + ty * (z11 - z10) : z00 + tx * (z11 - z01) + ty * (z01 - z00);
  return (1 - tx) * ((1 - ty) * z00 + ty * z01) + tx * ((1 - ty) * z10 + ty * z11);
}
__device__ double eos_ee(const EosDev &e, double r, double tK) { return eos_lookup(e, r, sqrt(tK)); }

// EOS_te_from_r_ee (imd_ttm_3D.c:2786-2811): T_e [K] from density and internal energy: minimise |ee(r,T) - e| over
// the temperature range of the ee(r, sqrt T) table with Brent's bracketing minimiser as the reference configures
// it (fminbnd3.c:6-107: x tolerance 1e-5 whatever is passed, eps weight, 500 iterations, golden first probe, and
// the sign its unparenthesised SIGNUM macro gives the parabolic numerator: +sign(q), makros.h:338).
__device__ __forceinline__ double d_sgn(double v) { return (double) ((v > 0) - (v < 0)); }
__device__ double d_te_from_r_ee(const EosDev &ee, double r, double e)
{
  const double tolx = 1e-5, eps = 2.220446049250313e-16, gold = 0.5 * (3.0 - sqrt(5.0));
  double lo = ee.ymin * ee.ymin, hi = ee.ymax * ee.ymax;
  double x = lo + gold * (hi - lo), w = x, v = x;
  double fx = fabs(eos_ee(ee, r, x) - e), fw = fx, fv = fx;
  double step = 0.0, prev = 0.0;
  for (int it = 0; it <= 500; it++) {
    const double mid = 0.5 * (lo + hi), tol = 2.0 * eps * fabs(x) + tolx / 3.0;
    if (fabs(x - mid) <= 2.0 * tol - 0.5 * (hi - lo) || it == 500) return x;
    bool golden = true;
    double u;
    if (fabs(prev) > tol) {
      double rr = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * rr;
      q = 2.0 * (q - rr);
      p *= d_sgn(q);
      q = fabs(q);
      rr = prev;
      prev = step;
      if (fabs(p) < fabs(0.5 * q * rr) && p > q * (lo - x) && p < q * (hi - x)) {
        step = p / q;
        u = x + step;
        if (fmin(u - lo, hi - u) < 2.0 * tol) step = tol * (d_sgn(mid - x) + (mid == x ? 1.0 : 0.0));
        golden = false;
      }
    }
    if (golden) {
      prev = (x >= mid) ? lo - x : hi - x;
      step = gold * prev;
    }
    u = x + fmax(fabs(step), tol) * (d_sgn(step) + (step == 0 ? 1.0 : 0.0));
    const double fu = fabs(eos_ee(ee, r, u) - e);
    if (fu < fx) {
      if (u < x) hi = x; else lo = x;
      v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
    } else {
      if (u < x) lo = u; else hi = u;
      if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    }
  }
  return -1.0;
}
__global__ void k_ttm_te_from_r_ee(int n, EosDev ee, const double *r, const double *e, double *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d_te_from_r_ee(ee, r[i], e[i]);
}

__global__ void k_ttm_eos_lookup(int n, EosDev tab, const double *x, const double *y, double *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = eos_lookup(tab, x[i], y[i]);
}

// do_ADV (imd_ttm_3D.c:1321-1458) on a lattice that is not split over ranks -- there do_COMMFLUX (:667-995) has
// nothing to move: the ghost columns it fills on a rank boundary are the neighbouring inner columns themselves, and
// the ghost columns at the two global x faces keep flux 0 / U 0 from init_ttm (:1136-1157), periodic or not.
// As compiled in the reference (ADVMODE 1, ADV2D undefined): upwind in x only, lattice plane k = 1 only (:1327); the
// new energy per atom goes to l2, an active cell gets its temperature from the inverse EOS, and the host swaps
// l1 <-> l2 afterwards.  One thread per cell of the plane; the Brent minimiser dominates (about 40 EOS look-ups).
// A lattice replicated on a grid of ranks holds the blocks of all of them: every rank of the reference advects the plane
// k = 1 of ITS block, i.e. the global planes 1 + m ldim_z, m < cpu_dim.z -- nplanes of them, kstride apart.
__global__ __launch_bounds__(64) void k_ttm_adv(Lat L, EosDev ee, double tau, const int *natoms, const int *natoms_old,
                                                const int *flux, int ncell, const double *dens, const double *U1,
                                                double *U2, double *T2, int nplanes, int kstride)
{
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nj = L.dim[1] - 2, per = (L.dim[0] - 2) * nj;
  if (t0 >= per * nplanes) return;
  const int t = t0 % per, k = 1 + (t0 / per) * kstride;
  const int i = t / nj + 1, j = t % nj + 1;
  const int c = fidx(L, i, j, k), xp = fidx(L, i + 1, j, k), xm = fidx(L, i - 1, j, k);
  const double Nold = (double) natoms_old[c], Nnew = (double) natoms[c];
  if (Nnew > 0) {
    const double u = U1[c];
    const double u2 = u * Nold / Nnew + tau * (
                        + (double) flux[c] * U1[xp]                       // flux[0]: received from +x
                        - (double) flux[(size_t) ncell + xp] * u          // +x neighbour's flux[1]: left to +x
                        + (double) flux[(size_t) ncell + c] * U1[xm]      // flux[1]: received from -x
                        - (double) flux[xm] * u                           // -x neighbour's flux[0]: left to -x
                      ) / Nnew;
    U2[c] = u2;
    if (natoms[c] >= L.fd_min_atoms) T2[c] = d_te_from_r_ee(ee, dens[c], u2 / (26.9815 * C_AMU * C_J2EV)) / 11604.5;
  } else if (Nnew < 1) { U2[c] = 0.0; T2[c] = 0.0; }
}

// laser_rescale_ttm (imd_laser.c:715-788, the branch without LASERYZ) with ttm_calc_depth (:699-710): Beer-Lambert
// source term exp(-mu depth) * [exp(-(t - t_0)^2 / sigma_t^2 / 2) * p_peak] in every ACTIVE cell; depth = distance of
// the cell's low corner along laser_dir behind laser_offset, clamped at 0.  `e` = the bracket, evaluated on the host.
__global__ void k_ttm_laser(Lat L, int g0, int g1, int g2, int d0, int d1, int d2, double offset, double mu, double e,
                            const int *natoms, double *source)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ni = L.dim[0] - 2, nj = L.dim[1] - 2, nk = L.dim[2] - 2;
  if (t >= ni * nj * nk) return;
  const int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1, c = fidx(L, i, j, k);
  if (natoms[c] < L.fd_min_atoms) return;
  double depth = (L.h[0] * ((i - 1) + g0)) * d0 + (L.h[1] * ((j - 1) + g1)) * d1 + (L.h[2] * ((k - 1) + g2)) * d2;
  depth -= offset;
  if (depth < 0) depth = 0;
  source[c] = exp(-mu * depth) * e;
}

// do_DIFF's bookkeeping of the absorbed laser energy (imd_ttm_3D.c:1647-1651): Eabs_local += source * fd_vol * tau for
// every inner cell in every sub-step.  The source does not change within an MD step, so `rep` sub-steps are booked at once.
__global__ void k_ttm_eabs(Lat L, const double *source, double vol_tau_rep, double *eabs)
{
  __shared__ double sh[16];
  const int ni = L.dim[0] - 2, nj = L.dim[1] - 2, nk = L.dim[2] - 2, n = ni * nj * nk;
  double s = 0.0;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1;
    s += source[fidx(L, i, j, k)];
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) tot += sh[w];
    *eabs += tot * vol_tau_rep;
  }
}

// do_COMMFLUX (imd_ttm_3D.c:667-995, the part compiled without ADV2D) for a lattice split over ranks: the columns
// i = 1 and i = lx-2 of {flux[8], U} for every inner (j, k) go to the x neighbours' ghost columns
// i = lx-1 / i = 0.  pack: side 0 = my column i = 1, side 1 = my column lx-2; 5 doubles per (j, k): the eight counters
// as ints, then U.  unpack: side 0 = into my ghost column i = 0, side 1 = into i = lx-1.
__global__ void k_ttm_commflux(Lat L, int side, int unpack, int *flux, int ncell, double *U, double *buf)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x, nj = L.dim[1] - 2, nk = L.dim[2] - 2;
  if (t >= nj * nk) return;
  const int j = t / nk + 1, k = t % nk + 1;
  const int i = unpack ? (side ? L.dim[0] - 1 : 0) : (side ? L.dim[0] - 2 : 1), c = fidx(L, i, j, k);
  double *o = buf + (size_t) t * 5;
  int *oi = (int *) o;
  if (unpack) {
#pragma unroll
    for (int q = 0; q < 8; q++) flux[(size_t) q * ncell + c] = oi[q];
    U[c] = o[4];
  } else {
#pragma unroll
    for (int q = 0; q < 8; q++) oi[q] = flux[(size_t) q * ncell + c];
    o[4] = U[c];
  }
}

// do_FILLMESH per cell: MeanCharge (More's Thomas-Fermi fit, imd_ttm_3D.c:2399-2473), getKappa (:2549-2592),
// getGamma (:2503-2528) and their helpers fermi_E, fermi_T, omega_pl, r0, coulomb_log, numet, numax, nupl
// (:2352-2498), written as ONE function so that everything the reference recomputes per helper call is shared:
// the Fermi energy, the Wigner-Seitz radius, the Coulomb logarithm and the thermal speed are evaluated once
// for kappa and gamma; pow(a,b) with run-time exponents is exp(b*log(a)) with the logarithms shared (7 pow ->
// 4 log + 6 exp), x^(2/3) and x^(1/3) use cbrt.  Differences to the reference are rounding-level (<=1e-13).
struct MeshOut { double Z, ne, Ce, k, g; };
// What of a cell's FILLMESH depends on its density and lattice temperature only: constant over the sub-steps of an MD step, so the
// sub-step loops (persistent kernel, Maxwell-solver loop) take it once per launch (one of the four logarithms among it).
// mesh_props = mesh_pre + mesh_eval: the same operations in the same order either way, the same bits.
struct MeshPre { double dens, TiK, r0, lr; };
__device__ __forceinline__ MeshPre mesh_pre(const Lat &L, double Tl, double dens)
{
  MeshPre q;
  q.dens = dens; q.TiK = Tl * 11604.5;
  q.r0 = dens * L.inv_z0am;
  q.lr = log(q.r0);
  return q;
}
__device__ __forceinline__ MeshOut mesh_eval(const Lat &L, const EosDev &cve, const MeshPre &q, double Te)
{
  MeshOut o;
  const double TeK = Te * 11604.5, TiK = q.TiK, dens = q.dens;
  {
    const double YA1 = 0.003323, YA2 = 0.971832, YA3 = 0.926148e-5, YA4 = 3.10165, YB0 = -1.7630, YB1 = 1.43175,
                 YB2 = 0.315463, YC1 = 0.036666667, YC2 = 0.983333333, YAL = 0.143139e2, YBE = 0.6624e0;
    const double t0 = TeK * L.inv_z0_43;
    const double lt = log(t0), lr = q.lr;
    const double ytf = t0 / (1.0 + t0);
    const double ya = YA1 * exp(YA2 * lt) + YA3 * exp(YA4 * lt);
    const double yb = -exp(YB0 + YB1 * ytf + YB2 * ytf * ytf);
    const double yc = YC1 * ytf + YC2;
    const double lq1 = log(ya) + yb * lr;                      // log(ya * r0^yb)
    const double sum = exp(yc * lr) + exp(yc * lq1);           // r0^yc + yq1^yc
    const double yx = YAL * exp(YBE / yc * log(sum));          // YAL * (sum^(1/yc))^YBE
    o.Z = L.atomic_charge * yx / (1.0 + yx + sqrt(1.0 + 2.0 * yx));
  }
  o.ne = o.Z * dens * L.inv_awamu;
  {
    double z = eos_lookup(cve, dens, TeK);                     // EOS_cve_from_r_te (:2736-2783)
    z *= dens; z *= 11604.5; z *= 1e-30; z *= C_J2EV;
    o.Ce = z;
  }
  constexpr double INV_EMASS = 1.0 / C_EMASS;
  const double Ne = o.ne, Z = o.Z;
  const double cb = cbrt(3.0 * C_PI * C_PI * Ne);
  const double EF = (C_HBAR * C_HBAR * 0.5 * INV_EMASS) * (cb * cb);            // fermi_E
  const double TF = 4.828648689433765e+22 * EF, VF = 1.481734876966785e+15 * sqrt(EF);
  const double inv_TF = 1.0 / TF;
  const double rws = cbrt((3.0 / (4 * C_PI)) * Z / Ne);                         // r0(ni), ni = Ne/Z
  const double kT = C_BOLTZMAN * TeK, sqkT = sqrt(kT), inv_kT = 1.0 / kT;
  const double bmin = fmax(Z * (C_ECHARGE * C_ECHARGE) * inv_kT, (C_HBAR / __builtin_sqrt(2 * C_EMASS)) * sqkT * inv_kT);
  const double ft = (2.0 / (3.0 * C_BOLTZMAN)) * EF;                            // fermi_T
  const double teff = sqrt(TeK * TeK + ft * ft);
  const double wpl = sqrt(Ne * (C_ECHARGE * C_ECHARGE * INV_EMASS / C_ECONST)); // omega_pl
  const double bmax = fmax(rws, sqrt((C_BOLTZMAN * INV_EMASS) * teff) / fmax(L.omega_laser, wpl));
  const double coulomb = fmax(2.0, log(bmax / bmin));
  const double vth_over_r = sqrt(VF * VF + kT * INV_EMASS) / rws;               // numax / A3
  const double te2_tf = TeK * TeK * inv_TF;
  {
    const double A1t = 2.95, A2t = 0.5, A3t = 0.16, A4t = 1.2;
    const double nu_eff = fmin(1.309202957843054e+11 * (A1t * TiK + A2t * te2_tf), A3t * vth_over_r);
    const double kappa_met = 6.884236239621913e-16 * Ne * TeK / nu_eff;
    const double kappa_pl = 4.428788911416808e+43 * (kT * kT * sqkT) / (Z * coulomb);
    o.k = (kappa_pl + (kappa_met - kappa_pl) * exp(-A4t * TeK * inv_TF)) * 0.073768115942029;
  }
  {
    const double A1g = 50, A2g = 20, A3g = 0.25, m_atom = 26.9815;
    const double nupl = 1.863839656495274e-40 * Z * Ne * coulomb * (inv_kT * inv_kT * sqkT);
    const double nu_eff = fmin(1.309202957843054e+11 * (A1g * TiK + A2g * te2_tf), fmin(A3g * vth_over_r, nupl));
    o.g = 2.783313120645740e-74 / (m_atom * C_AMU) * Ne * nu_eff;
  }
  return o;
}
__device__ __forceinline__ MeshOut mesh_props(const Lat &L, const EosDev &cve, double Te, double Tl, double dens)
{
  return mesh_eval(L, cve, mesh_pre(L, Tl, dens), Te);
}

// occ (may be null): dim[0] + dim[1] + dim[2] ints, zeroed before the launch, in which every wave MARKS the lowest and the highest
// inner coordinate of its active cells per dimension (plain stores of 1 spread over ~130 words: six atomics per wave on six
// words cost 60 us at config 3's lattice, serialised in L2).  The persistent sub-step kernel reads the bounding box of the active
// cells off the marks -- first and last marked coordinate per dimension -- and anchors its blocks there.
__global__ void k_ttm_fillmesh(Lat L, int steps, EosDev cve, EosDev ee, const int *natoms, const double *temp,
                               const double *md_temp, const double *dens, double *Z, double *ne, double *Ce,
                               double *fd_k, double *fd_g, double *U0, double *U1, int *occ)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ni = L.dim[0] - 2, nj = L.dim[1] - 2, nk = L.dim[2] - 2;
  const int BIG = 1 << 20;
  int v[6] = {0, 0, 0, 0, 0, 0};            // max of (BIG - coordinate) = the lowest coordinate, max of the coordinate
  if (t < ni * nj * nk) {
    const int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1;
    const int c = fidx(L, i, j, k);
    if (natoms[c] >= L.fd_min_atoms) {
      MeshOut o = mesh_props(L, cve, temp[c], md_temp[c], dens[c]);
      Z[c] = o.Z; ne[c] = o.ne; Ce[c] = o.Ce; fd_k[c] = o.k; fd_g[c] = o.g;
      if (steps < 1) { double u = eos_ee(ee, dens[c], temp[c] * 11604.5) * 26.9815 * C_AMU * C_J2EV; U0[c] = u; U1[c] = u; }
      v[0] = BIG - i; v[1] = BIG - j; v[2] = BIG - k; v[3] = i; v[4] = j; v[5] = k;
    } else { fd_k[c] = 0.0; fd_g[c] = 0.0; Z[c] = 0.0; ne[c] = 0.0; Ce[c] = 0.0; }
  }
  if (occ) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_down(v[q], o, 64); v[q] = w > v[q] ? w : v[q]; }
    }
    if ((threadIdx.x & 63) == 0 && v[3] > 0) {
      int off = 0;
#pragma unroll
      for (int d = 0; d < 3; d++) { occ[off + BIG - v[d]] = 1; occ[off + v[3 + d]] = 1; off += L.dim[d]; }
    }
  }
}

// do_cell_activation (imd_ttm_3D.c:1468-1596).  The reference walks the cells in order (i outer, k inner) and writes
// a freshly activated cell's fallback temperature IN PLACE, so a cell activated in the same step as a neighbour that
// comes earlier in that order sees the neighbour's NEW temperature.  Two kernels keep that: the first, parallel, does
// what is order independent (xi = 0 of deactivated cells) and collects the cells that need the fallback (rare:
// usually none); the second, one wave, puts them into the reference's order and processes them one after the other.
// X (decomposed lattice with ghost records, steps the persistent sub-step kernel will take): every inner cell on a face with
// another rank beyond it tells that rank its atom count, tagged with the coming launch's base -- active or not: the count
// decides between mirror and flux on the other side (the natoms plane of ttm_fill_ghost_layers).  Done HERE, by the kernel
// that walks all inner cells anyway: a launch of its own for it was measured to sit 1.2 ms in the queue beside the freshly
// started force pass (profiles/r04_persist_rem_variants.json).
struct XPub { unsigned long long *peer[6]; int xoff[6], xcells; unsigned base; };
__global__ void k_ttm_activation(Lat L, const int *natoms, const int *natoms_old, const double *T1, double *xi,
                                 int *cand, int *ncand, XPub X)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ni = L.dim[0] - 2, nj = L.dim[1] - 2, nk = L.dim[2] - 2;
  if (t >= ni * nj * nk) return;
  const int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1, c = fidx(L, i, j, k), m = L.fd_min_atoms;
  if (X.xcells > 0) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int d = q >> 1, me = d == 0 ? i : d == 1 ? j : k;
      if (X.peer[q] == nullptr || me != ((q & 1) ? 1 : L.dim[d] - 2)) continue;
      const int f = d == 0 ? (j - 1) * nk + (k - 1) : d == 1 ? (i - 1) * nk + (k - 1) : (i - 1) * nj + (j - 1);      // face_index
      __hip_atomic_store(X.peer[q] + (size_t) 2 * X.xcells * 4 + X.xoff[q ^ 1] + f, ((unsigned long long) X.base << 32) | (unsigned) natoms[c],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (natoms_old[c] >= m && natoms[c] < m) xi[c] = 0.0;
  else if (natoms_old[c] < m && natoms[c] >= m) {
    const double tc = T1[c];
    if (isnan(tc) || tc <= L.Temin) cand[atomicAdd(ncand, 1)] = t;       // t = position in the reference's loop order
  }
}

__device__ void d_activate_cell(const Lat &L, const EosDev &ee, int t, const int *natoms, double *T1, double *T2,
                                double *U1, double *U2, const double *md_temp, const double *dens)
{
  const int nj = L.dim[1] - 2, nk = L.dim[2] - 2;
  const int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1, c = fidx(L, i, j, k), m = L.fd_min_atoms;
  const int nb[6] = {fidx(L, i + 1, j, k), fidx(L, i - 1, j, k), fidx(L, i, j + 1, k), fidx(L, i, j - 1, k),
                     fidx(L, i, j, k + 1), fidx(L, i, j, k - 1)};
  int nn = 0; double E = 0, tc;
  for (int q = 0; q < 6; q++) if (natoms[nb[q]] >= m) { E += T1[nb[q]] * T1[nb[q]]; nn++; }
  if (nn != 0) { tc = sqrt(E / ((double) nn)); if (tc < L.Temin) tc = md_temp[c]; }
  else tc = md_temp[c];
  T1[c] = tc; T2[c] = tc;
  const double u = eos_ee(ee, dens[c], tc * 11604.5) * 26.9815 * C_AMU * C_J2EV;
  U1[c] = u; U2[c] = u;
}

// one wave (a multi-wave workgroup would not be placed beside the persistent force kernels)
__global__ __launch_bounds__(64) void k_ttm_activation_seq(Lat L, EosDev ee, const int *natoms, double *T1, double *T2,
                                                           double *U1, double *U2, const double *md_temp,
                                                           const double *dens, int *cand, int *sorted, int *ncand,
                                                           int rank_max)
{
  const int n = *ncand, lane = threadIdx.x;
  if (n == 0) return;
  if (n <= rank_max) {
    // rank sort (the positions are distinct): O(n^2 / 64), n is a handful of cells in practice
    for (int a = lane; a < n; a += 64) {
      const int v = cand[a];
      int r = 0;
      for (int b = 0; b < n; b++) r += cand[b] < v ? 1 : 0;
      sorted[r] = v;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0)
      for (int q = 0; q < n; q++) d_activate_cell(L, ee, sorted[q], natoms, T1, T2, U1, U2, md_temp, dens);
  } else {
    // many at once (e.g. a lattice uploaded cold): mark them and walk all cells in order, 64 per trip
    const int ninner = (L.dim[0] - 2) * (L.dim[1] - 2) * (L.dim[2] - 2);
    for (int a = lane; a < ninner; a += 64) sorted[a] = 0;
    __threadfence_block();
    __syncthreads();
    for (int a = lane; a < n; a += 64) sorted[cand[a]] = 1;
    __threadfence_block();
    __syncthreads();
    for (int base = 0; base < ninner; base += 64) {
      const int m = base + lane < ninner ? sorted[base + lane] : 0;
      unsigned long long mask = __ballot(m != 0);
      if (lane == 0)
        while (mask) {
          const int b = __ffsll((long long) mask) - 1;
          mask &= mask - 1;
          d_activate_cell(L, ee, base + b, natoms, T1, T2, U1, U2, md_temp, dens);
        }
    }
  }
  if (lane == 0) *ncand = 0;
}

// ttm_fill_ghost_layers, one rank: FACE cells only (plane datatypes of
// ttm_create_mpi_datatypes, imd_ttm_3D.c:2972-3312); periodic: copy from the
// opposite inner plane; open: ghost face empty (natoms = 0).
// remote: bit q (2d: the + face, 2d+1: the - face) = that ghost face belongs to another rank and is filled by it (decomposed
// lattice with remote records: the persistent sub-step kernel has written it); such faces are left alone
__global__ void k_ttm_ghosts(Lat L, int *natoms, double *temp, double *fd_k, int remote = 0)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int *D = L.dim;
  int off = 0;
  for (int d = 0; d < 3; d++) {
    int d1 = (d + 1) % 3, d2 = (d + 2) % 3, n1 = D[d1] - 2, n2 = D[d2] - 2, nf = n1 * n2;
    if (t >= off && t < off + nf) {
      int q = t - off, idx[3];
      idx[d1] = q / n2 + 1; idx[d2] = q % n2 + 1;
      idx[d] = 0;        int lo = fidx(L, idx[0], idx[1], idx[2]);
      idx[d] = D[d] - 1; int hi = fidx(L, idx[0], idx[1], idx[2]);
      idx[d] = 1;        int ilo = fidx(L, idx[0], idx[1], idx[2]);
      idx[d] = D[d] - 2; int ihi = fidx(L, idx[0], idx[1], idx[2]);
      const bool rhi = remote >> (2 * d) & 1, rlo = remote >> (2 * d + 1) & 1;
      if (L.pbc[d] == 1) {
        if (!rhi) { natoms[hi] = natoms[ilo]; temp[hi] = temp[ilo]; fd_k[hi] = fd_k[ilo]; }
        if (!rlo) { natoms[lo] = natoms[ihi]; temp[lo] = temp[ihi]; fd_k[lo] = fd_k[ihi]; }
      } else { if (!rlo) natoms[lo] = 0; if (!rhi) natoms[hi] = 0; }
      return;
    }
    off += nf;
  }
}

// do_DIFF: explicit 7-point variable-kappa step with mirror at inactive
// neighbours (imd_ttm_3D.c:1603-1829), LDS-blocked: a workgroup stages its
// (TX+2)x(TY+2)x(TZ+2) tile of {temp, fd_k, natoms} once.
// one wavefront per workgroup: a 39^3 lattice gives 1000 workgroups = one per SIMD of the chip, a single round
#define DT_X 2
#define DT_Y 4
#define DT_Z 8
// WRAP: ghost-shell cells are not read from memory but resolved on the fly (periodic direction: the
// opposite inner cell; open direction: empty), i.e. ttm_fill_ghost_layers is folded into the load --
// valid when the lattice is not split over ranks.
template <bool FUSE_MESH, bool WRAP>
__global__ __launch_bounds__(DT_X *DT_Y *DT_Z) void k_ttm_diff(
    Lat L, double tau, int nsub, int steps, const int *natoms, const double *T1, double *T2, double *U1, double *U2,
    const double *fd_k, double *fd_k_out, double *Ce, double *fd_g, const double *md_temp, const double *source,
    const double *dens, double *xi, double *Zo, double *neo, EosDev cve, EosDev ee, int flags, const int *run_if)
{
  // run_if: status word of the persistent sub-step kernel that ran in front of this launch chain -- 0 = it did the work
  if (run_if && *run_if == 0) return;
  // flags: 1 = store Z and ne (diagnostic fields nothing in the sub-step loop reads: only the last sub-step of an MD
  // step has to), 2 = there is a source term to read.  A large (replicated) lattice is bound by its bytes per cell.
  // NOLDS (the fused single-rank chain, <true, true>): no LDS tile -- every thread reads its six neighbours where they are (the
  // lattice is cache resident: 168 B per cell, L1 / L2 hits).  A kernel without LDS can sit beside the coefficient form of the EAM
  // force pass, which fills the CU's 160 KB: lattices too large for the persistent kernel (80 M atoms on one GPU: 4.7e5 FD cells)
  // then keep that force pass (imdgpu_ttm_chain_needs_lds).  Same values from the same cells: same bits as the tiled form.
  constexpr bool NOLDS = FUSE_MESH && WRAP;
  __shared__ double sT[NOLDS ? 1 : DT_X + 2][NOLDS ? 1 : DT_Y + 2][NOLDS ? 1 : DT_Z + 2];
  __shared__ double sK[NOLDS ? 1 : DT_X + 2][NOLDS ? 1 : DT_Y + 2][NOLDS ? 1 : DT_Z + 2];
  __shared__ int sN[NOLDS ? 1 : DT_X + 2][NOLDS ? 1 : DT_Y + 2][NOLDS ? 1 : DT_Z + 2];
  // chain of tiny launches sharing the CUs with the force passes.  Raising the wave priority (1..3, all the same)
  // finishes the chain in 1.4 ms instead of 2.2 but costs the EAM force pass 0.22 ms: measured best is NOT to raise it
  // (180 vs 177 steps/s); IMDGPU_TTM_PRIO keeps the experiment available.
  if (L.wave_prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (L.wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (L.wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  const int tz = threadIdx.x % DT_Z, ty = (threadIdx.x / DT_Z) % DT_Y, tx = threadIdx.x / (DT_Z * DT_Y);
  const int nbz = (L.dim[2] - 2 + DT_Z - 1) / DT_Z, nby = (L.dim[1] - 2 + DT_Y - 1) / DT_Y;
  const int bz = blockIdx.x % nbz, by = (blockIdx.x / nbz) % nby, bx = blockIdx.x / (nbz * nby);
  const int i0 = bx * DT_X, j0 = by * DT_Y, k0 = bz * DT_Z;     // tile origin in ghost-including coordinates
  // one cell of the (ghost-including) lattice as the stencil sees it: ghost-shell cells resolved on the fly when WRAP
  auto cell_of = [&](int gi, int gj, int gk, double &t, double &kk, int &n) {
    t = 0; kk = 0; n = 0;
    if (gi < L.dim[0] && gj < L.dim[1] && gk < L.dim[2]) {
      bool empty = false;
      if (WRAP) {
        int g[3] = {gi, gj, gk};
#pragma unroll
        for (int d = 0; d < 3; d++) {
          if (g[d] == 0) { if (L.pbc[d] == 1) g[d] = L.dim[d] - 2; else empty = true; }
          else if (g[d] == L.dim[d] - 1) { if (L.pbc[d] == 1) g[d] = 1; else empty = true; }
        }
        gi = g[0]; gj = g[1]; gk = g[2];
      }
      if (!empty) { int c = fidx(L, gi, gj, gk); t = T1[c]; kk = fd_k[c]; n = natoms[c]; }
    }
  };
  if constexpr (!NOLDS) {
    for (int s = threadIdx.x; s < (DT_X + 2) * (DT_Y + 2) * (DT_Z + 2); s += blockDim.x) {
      int lz = s % (DT_Z + 2), ly = (s / (DT_Z + 2)) % (DT_Y + 2), lx = s / ((DT_Z + 2) * (DT_Y + 2));
      double t, kk; int n;
      cell_of(i0 + lx, j0 + ly, k0 + lz, t, kk, n);
      sT[lx][ly][lz] = t; sK[lx][ly][lz] = kk; sN[lx][ly][lz] = n;
    }
    __syncthreads();
  }
  const int i = i0 + tx + 1, j = j0 + ty + 1, k = k0 + tz + 1;
  if (i > L.dim[0] - 2 || j > L.dim[1] - 2 || k > L.dim[2] - 2) return;
  const int c = fidx(L, i, j, k), m = L.fd_min_atoms;
  const int lx = tx + 1, ly = ty + 1, lz = tz + 1;
  double Te, kk; int nat;
  if constexpr (NOLDS) cell_of(i, j, k, Te, kk, nat);
  else { nat = sN[lx][ly][lz]; Te = sT[lx][ly][lz]; kk = sK[lx][ly][lz]; }
  if (nat < m) {
    if (FUSE_MESH) { fd_k_out[c] = 0.0; fd_g[c] = 0.0; Ce[c] = 0.0; if (flags & 1) { Zo[c] = 0.0; neo[c] = 0.0; } }
    return;
  }
  const double invxsq = L.invhsq[0], invysq = L.invhsq[1], invzsq = L.invhsq[2];
#define NB(dx, dy, dz, TT, KK) \
  double TT, KK; { int nn_; if constexpr (NOLDS) cell_of(i + dx, j + dy, k + dz, TT, KK, nn_); \
                   else { nn_ = sN[lx + dx][ly + dy][lz + dz]; TT = sT[lx + dx][ly + dy][lz + dz]; KK = sK[lx + dx][ly + dy][lz + dz]; } \
                   if (nn_ < m) { TT = Te; KK = kk; } }
  NB(1, 0, 0, xaT, xaK) NB(-1, 0, 0, xbT, xbK) NB(0, 1, 0, yaT, yaK) NB(0, -1, 0, ybT, ybK) NB(0, 0, 1, zaT, zaK) NB(0, 0, -1, zbT, zbK)
#undef NB
  const double ce = Ce[c], g = fd_g[c], Tl = md_temp[c];
  const double Tn = tau / ce *
      ( ( ((kk + xaK) / 2 * (xaT - Te) * invxsq)
         - ((kk + xbK) / 2 * (Te - xbT) * invxsq)
         + ((kk + yaK) / 2 * (yaT - Te) * invysq)
         - ((kk + ybK) / 2 * (Te - ybT) * invysq)
         + ((kk + zaK) / 2 * (zaT - Te) * invzsq)
         - ((kk + zbK) / 2 * (Te - zbT) * invzsq) )
        - g * (Te - Tl)
        + ((flags & 2) ? source[c] : 0.0)
      ) + Te;
  T2[c] = Tn;
  U2[c] = U1[c] + (Tn - Te) * ce * L.vol / ((double) nat);
  xi[c] += (Tn - Tl) * L.xi_fac * g / (Tl * dens[c]);
  if (FUSE_MESH) {                 // do_FILLMESH of the same cell with the new temperature
    MeshOut o = mesh_props(L, cve, Tn, Tl, dens[c]);
    if (flags & 1) { Zo[c] = o.Z; neo[c] = o.ne; }
    Ce[c] = o.Ce; fd_k_out[c] = o.k; fd_g[c] = o.g;
    if (steps < 1) {               // do_FILLMESH re-initialises U from the EOS on every call of step 0 (:614-623)
      double u = eos_ee(ee, dens[c], Tn * 11604.5) * 26.9815 * C_AMU * C_J2EV;
      U1[c] = u; U2[c] = u;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// All diffusion sub-steps of one MD step in ONE launch (the loop of calc_ttm, imd_ttm_3D.c:62-77, without a return to
// the host or to the command processor between sub-steps).  One wave per 4x4x4 block of inner FD cells, every block
// resident at once (a 39^3 lattice: 1000 one-wave workgroups, one per SIMD beside the EAM force pass).  A lane keeps its
// cell's state (Te, kappa, Ce, gamma, U, xi) in registers for the whole loop.  What the stencil needs from other cells:
//   * neighbours inside the block: (Te, kappa) by lane shuffle;
//   * neighbours in another block: through RECORDS in device memory, four 8-byte granules per cell and parity
//     {tag | hi(Te)}, {tag | lo(Te)}, {tag | hi(kappa)}, {tag | lo(kappa)}, each written by ONE agent-scope relaxed atomic
//     store (`sc1`, write-through) and read by agent-scope relaxed atomic loads until its tag shows the wanted sub-step:
//     the data is the flag (cdna_hip_programming.md, Guideline 16, form R2) -- no drain, no flag store, no fence, one
//     memory round trip per hand-off.  tag = base + s for state s (after s sub-steps), base advancing by nsub + 1 per
//     launch, so records never have to be cleared; state s lives in parity s & 1.
//   * overwriting parity (s+1) & 1 (state s-1) is safe once every neighbour record this block reads carries tag s: the
//     neighbour block computed its state s from this block's state s-1 before it stored that tag.
// State 0 is read from temp[] / fd_k[] (written by the kernels in front), the final state goes back there.
// The arithmetic is k_ttm_diff<true, true>'s, statement by statement: results are bit-identical to the launch chain.
// Liveness: a census word counts the workgroups that have started; the LAST arriver opens the gate, a workgroup that
// waits longer than PERSIST_CENSUS_TICKS closes it for good (one CAS decides) and every workgroup then leaves WITHOUT having
// touched the lattice (status 1: the guarded launch chain behind this kernel does the step instead).  Spins inside the
// loop are bounded as well (status 2 + CNT_TTM_TIMEOUT: the lattice of this step is not valid, reported by the host).
#define PT 4
// Blocks that can be resident at once.  Beside a force pass of eleven waves per CU (three SIMDs with three of its waves, one
// with two) every SIMD has the registers for one wave of this kernel and the fourth for two: 5 x 256.  (Fewer fit beside a
// larger force workgroup, more when the kernel runs alone; a launch whose blocks do not all arrive steps aside, see below.)
#define PERSIST_MAX_TILES 1280
#define PERSIST_CENSUS_TICKS 20000000ull       // 0.2 s of the 100 MHz s_memrealtime counter
#define PERSIST_SPIN_TICKS 200000000ull        // 2 s
struct PersistDev {
  unsigned *census, *gate;              // zeroed before every launch
  unsigned long long *rec;              // [2][ncell][4] granules
  int *status, *counters;
  int trusted, ncell;
  unsigned base;                        // tag of state 0 of this launch
  const int *bbox;                      // marks of the active cells' lowest / highest coordinates, as k_ttm_fillmesh leaves them (device; this step's)
  int extra_census;                     // test hook (option ttm_persist_miss): workgroups the census waits for in vain
  // decomposed lattice (REM): the cells beyond a rank boundary live on ANOTHER GPU.  Every rank owns a buffer of ghost
  // records -- per face cell and parity the same four granules as `rec`, then one granule {tag | natoms} per face cell --
  // which its neighbours write straight into (system-scope stores over xGMI, or the same memory when ranks share a GPU)
  // and which it polls locally: the push direction keeps the polling off the link.
  unsigned long long *xrec;             // mine: [2][xcells][4], then [xcells] natoms granules
  unsigned long long *peer[6];          // the buffer of the rank beyond face q (2d: +, 2d+1: -), nullptr = no other rank there
  int xoff[6], xcells;                  // first record of face q in every rank's buffer; records per parity
};
__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rec_store(unsigned long long *r, unsigned tag, double v)
{
  const unsigned long long b = (unsigned long long) __double_as_longlong(v), t = (unsigned long long) tag << 32;
  __hip_atomic_store(r, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(r + 1, t | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same granules across GPUs: system scope (fine-grained memory; over xGMI a store is posted, the poll stays local)
__device__ __forceinline__ void xrec_store(unsigned long long *r, unsigned tag, double v)
{
  const unsigned long long b = (unsigned long long) __double_as_longlong(v), t = (unsigned long long) tag << 32;
  __hip_atomic_store(r, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(r + 1, t | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long xrec_load(const unsigned long long *r) { return __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// index of inner cell (i, j, k) within face q of its rank's block: the two other inner coordinates, larger dimension fastest
__device__ __forceinline__ int face_index(const Lat &L, int q, int i, int j, int k)
{
  const int d = q >> 1;
  return d == 0 ? (j - 1) * (L.dim[2] - 2) + (k - 1) : d == 1 ? (i - 1) * (L.dim[2] - 2) + (k - 1) : (i - 1) * (L.dim[1] - 2) + (j - 1);
}


// WPE = waves per SIMD the register budget is cut for: 6 (80 VGPRs, 52 of them spilled to scratch) is what fits beside three
// 144-register waves of the three-row EAM force pass; 4 (128 VGPRs, no spills: 0.76 instead of 0.83 ms alone) is what a
// stand-alone calc_ttm runs and what fits beside the 128-register coefficient form of the force pass (the usual case); 5 (96
// VGPRs) is kept for experiments
// REM: the lattice is this rank's BLOCK of a decomposed lattice; faces towards other ranks go through PersistDev::xrec / peer
template <bool SRC, int WPE, bool REM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_ttm_chain_persist(
    Lat L, double tau, int nsub, int *natoms, double *T0, double *T1, double *U0, double *U1, double *K0, double *K1,
    double *Ce, double *fd_g, const double *md_temp, const double *source, const double *dens, double *xi, double *Zo,
    double *neo, EosDev cve, PersistDev P)
{
  const int lane = threadIdx.x, tile = blockIdx.x;
#ifdef PERSIST_PRIO
  // experiment: a sub-step's arithmetic at a raised priority (its few hundred instructions queue behind the force pass's three
  // waves on the SIMD otherwise), the polls for the neighbours' records at priority 0 (a spinning wave must not take issue slots)
#define PRIO_WORK __builtin_amdgcn_s_setprio(PERSIST_PRIO)
#define PRIO_POLL __builtin_amdgcn_s_setprio(0)
#else
#define PRIO_WORK
#define PRIO_POLL
#endif
  // The blocks are anchored on the bounding box of the ACTIVE cells of this step (k_ttm_fillmesh): a slab with vacuum, or the
  // share of it a rank holds, needs blocks for its filled layers only, and the box moves with the surfaces.  Per dimension
  // the box of n layers is cut into ceil(n / 4) blocks of four layers -- with n = 4 q + 1 the last two are three and two
  // layers wide, so no block is ONE layer wide (a cell of such a block would have both its neighbours of that dimension
  // in other blocks; the hand-off below knows one per dimension).  Cells outside the box are inactive: FILLMESH has
  // zeroed what the launch chain would zero for them.  The kernel steps aside (status 1, nothing touched: the launch chain
  // does the step) when the box needs more blocks than the launch brought -- all of them must be resident at once -- or is
  // one layer thin.
  int lo[3], nn[3], nb[3];
  bool aside = false, none = false;
  {
    const int *occ = P.bbox;                                                       // the marks k_ttm_fillmesh left, dimension after dimension
#pragma unroll
    for (int d = 0; d < 3; d++) {
      int a = 1 << 20, b = 0;
      for (int base = 1; base <= L.dim[d] - 2; base += 64) {
        const int x = base + lane;
        const unsigned long long mk = __ballot(x <= L.dim[d] - 2 && occ[x] != 0);
        if (mk) { const int f = base + __ffsll((long long) mk) - 1, l = base + 63 - __clzll((long long) mk); a = f < a ? f : a; b = l > b ? l : b; }
      }
      occ += L.dim[d];
      if (b == 0) { a = 1; none = true; }                                          // no active cell at all
      // one layer thin (the first filled layer of a rank's block: material expanding or ablating into an empty block): the
      // box takes the inactive layer next to it along -- in-box cells without atoms are mirrors like cells outside the box
      if (b == a && L.dim[d] - 2 >= 2) { if (a > 1) a--; else b++; }
      lo[d] = a; nn[d] = b - a + 1; nb[d] = (nn[d] + PT - 1) / PT;
      if (nn[d] == 1) aside = true;                                                // a lattice of one layer in this dimension
    }
  }
  if (none) return;                                                                // nothing active, nothing to do: status stays 0
  const int ntiles = nb[0] * nb[1] * nb[2];
  if (ntiles > (int) gridDim.x) aside = true;
  if (aside) {
    if (tile == 0 && lane == 0) {
      __hip_atomic_store(P.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // other ranks' kernels are waiting for this one's records: an error on every rank (the flag all-reduce takes the maximum).
      // 2 = "this rank's active region does not fit the launch", told apart from a time-out (1) by the host's message
      if (REM) __hip_atomic_store(P.counters + CNT_TTM_TIMEOUT, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (tile >= ntiles) return;
  const int tb[3] = {tile / (nb[2] * nb[1]), (tile / nb[2]) % nb[1], tile % nb[2]};
  const int l3[3] = {lane >> 4, (lane >> 2) & 3, lane & 3};
  int bs[3], bw[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if ((nn[d] & 3) == 1 && tb[d] >= nb[d] - 2) { bs[d] = PT * (nb[d] - 2) + (tb[d] == nb[d] - 1 ? 3 : 0); bw[d] = tb[d] == nb[d] - 1 ? 2 : 3; }
    else { bs[d] = PT * tb[d]; bw[d] = nn[d] - bs[d] < PT ? nn[d] - bs[d] : PT; }
  }
  const int i = lo[0] + bs[0] + l3[0], j = lo[1] + bs[1] + l3[1], k = lo[2] + bs[2] + l3[2];
  const bool inside = l3[0] < bw[0] && l3[1] < bw[1] && l3[2] < bw[2];
  const int m = L.fd_min_atoms, c = inside ? fidx(L, i, j, k) : 0;
  const int nat = inside ? natoms[c] : 0;
  const bool active = inside && nat >= m;
  // ---- census and gate -------------------------------------------------------------------------------------------
  int go = 0;
  if (lane == 0) {
    const unsigned long long t0 = wall_clock64();
    const unsigned n = __hip_atomic_fetch_add(P.census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (n == (unsigned) (ntiles + P.extra_census)) {
      unsigned expect = 0;
      __hip_atomic_compare_exchange_strong(P.gate, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (;;) {
      const unsigned g = ld_flag(P.gate);
      if (g == 1u) { go = 1; break; }
      if (g >= 2u) break;
      // trusted (no guarded launch chain behind this kernel any more; earlier launches have shown that all workgroups fit): a
      // late arrival can only mean that something else holds compute units for the moment -- wait ten times longer before
      // giving the step up
      if (wall_clock64() - t0 > (P.trusted ? PERSIST_SPIN_TICKS : PERSIST_CENSUS_TICKS)) {
        unsigned expect = 0;
        if (__hip_atomic_compare_exchange_strong(P.gate, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          // status 1 = nothing of the lattice has been touched: the guarded launch chain behind this kernel (trial) or the
          // host (trusted: the integrator is guarded by the same word and the step's lattice part is issued again) does the step.
          // With faces through other ranks' records (REM) their kernels are waiting for this one: an error on every rank
          __hip_atomic_store(P.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (REM) __hip_atomic_store(P.counters + CNT_TTM_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if (expect == 1u) go = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  go = __shfl(go, 0, 64);
  if (!go) return;                                     // nothing of the lattice has been touched
  if (L.wave_prio) __builtin_amdgcn_s_setprio(2);      // IMDGPU_TTM_PRIO: experiment
  // ---- neighbours ------------------------------------------------------------------------------------------------
  // per direction: nc = cell index of the neighbour, or -1 = mirror (outside an open boundary or an inactive cell; the
  // ghost shell is resolved on the fly as in k_ttm_diff<.., WRAP = true>); intile bit = the neighbour is a lane of this wave
  int nc[6];
  unsigned intile = 0, rem = 0;                        // rem bit q: the neighbour in direction q is a cell of another rank
  const unsigned long long tloop = wall_clock64();
  if (REM) {
    // every ACTIVE inner cell on a face towards another rank gives that rank its state 0; the atom counts of all face cells
    // (activity decides mirror or flux) have gone out before this launch (k_ttm_activation: the blocks of this kernel cover
    // the active cells only) -- all of it tagged with this launch's base; the ghost cells' counts are awaited below
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int d = q >> 1, me = d == 0 ? i : d == 1 ? j : k;
      if (!inside || P.peer[q] == nullptr || me != ((q & 1) ? 1 : L.dim[d] - 2)) continue;
      rem |= 1u << q;
      if (active) {
        unsigned long long *r = P.peer[q] + (size_t) (P.xoff[q ^ 1] + face_index(L, q, i, j, k)) * 4;      // my cell in the peer's face q ^ 1, parity 0
        xrec_store(r, P.base, T0[c]); xrec_store(r + 2, P.base, K0[c]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; q++) {
    int g[3] = {i, j, k};
    const int d = q >> 1, dir = (q & 1) ? -1 : 1;
    g[d] += dir;
    bool empty = !inside;
    int cell = -1;
    if (REM && (rem >> q & 1u)) {                      // ghost cell filled by the rank beyond: wait for its atom count
      const int gc = fidx(L, g[0], g[1], g[2]);
      const unsigned long long *r = P.xrec + (size_t) 2 * P.xcells * 4 + P.xoff[q] + face_index(L, q, i, j, k);
      int gn = -1;
      for (;;) {
        const unsigned long long v = xrec_load(r);
        if ((unsigned) (v >> 32) == P.base) { gn = (int) (unsigned) v; break; }
        if (wall_clock64() - tloop > PERSIST_SPIN_TICKS || ld_flag(P.gate) == 3u) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (gn < 0) {                                    // the neighbour rank never started its launch: give up, loudly
        __hip_atomic_store(P.gate, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.counters + CNT_TTM_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gn = 0;
      }
      if (gn >= m) cell = gc;                          // (the ghost shell of the arrays follows in k_ttm_xrec_final)
      nc[q] = cell;
      continue;
    }
    if (g[d] == 0) { if (L.pbc[d] == 1) g[d] = L.dim[d] - 2; else empty = true; }
    else if (g[d] == L.dim[d] - 1) { if (L.pbc[d] == 1) g[d] = 1; else empty = true; }
    if (!empty) { cell = fidx(L, g[0], g[1], g[2]); if (natoms[cell] < m) cell = -1; }
    nc[q] = cell;
    const int ln = l3[d] + dir;
    if (ln >= 0 && ln < bw[d] && g[d] == (d == 0 ? i : d == 1 ? j : k) + dir) intile |= 1u << q;    // same block and not wrapped
  }
  if (inside && !active) { K0[c] = 0.0; K1[c] = 0.0; fd_g[c] = 0.0; Ce[c] = 0.0; Zo[c] = 0.0; neo[c] = 0.0; }   // what the chain's launches write (nobody reads an inactive cell)
  const bool any_active = __ballot(active) != 0ull;
  if (!any_active) return;                             // a block without an active cell: nothing to compute, never read
  // at most ONE neighbour per dimension lies in another block or on another rank (the host does not use this kernel for
  // lattices with a one-cell-wide last block): xc[d] = its cell -- for a cell of another rank: its record in xrec -- or -1
  // (bit d of xany: there is one; bit d of xrm: it is a record of xrec)
  int xc[3]; unsigned xany = 0, xrm = 0;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const bool cp = nc[2 * d] >= 0 && !(intile >> (2 * d) & 1u), cm = nc[2 * d + 1] >= 0 && !(intile >> (2 * d + 1) & 1u);
    xc[d] = cp ? nc[2 * d] : (cm ? nc[2 * d + 1] : -1);
    if (cp || cm) xany |= 1u << d;
    if (REM && (cp || cm)) {
      const int q = cp ? 2 * d : 2 * d + 1;
      if (rem >> q & 1u) { xrm |= 1u << d; xc[d] = P.xoff[q] + face_index(L, q, i, j, k); }
    }
  }
  // an inactive cell of a partly active block computes nothing and is a mirror for its neighbours, which therefore
  // publish no record on its account: it must not wait for any (rough surfaces, voids, craters across a block face)
  if (!active) xany = 0;
  // what the loop keeps of all this (registers are what this kernel is short of: one wave per SIMD beside the force pass, and a
  // spilled value reloaded inside the sub-step's dependent chain costs a memory round trip nothing hides): one bit per
  // direction "there is a neighbour" instead of its cell number, and -- REM -- per dimension the record of this cell in the
  // buffer of the rank beyond (a cell has at most one face per dimension towards another rank: the lattice is at least two
  // cells wide) with the side it lies on
  unsigned nvalid = 0;
#pragma unroll
  for (int q = 0; q < 6; q++) if (nc[q] >= 0) nvalid |= 1u << q;
  // a cell on a face towards another rank publishes to that rank whether the cell beyond is active or not: pidx[d] = its
  // record in that rank's buffer (-1: no such face in dimension d), bit 19 + d of the flags = the face is the low one
  int pidx[3] = {-1, -1, -1};
  unsigned pside = 0;
  if (REM) {
#pragma unroll
    for (int q = 0; q < 6; q++) if (rem >> q & 1u) { pidx[q >> 1] = P.xoff[q ^ 1] + face_index(L, q, i, j, k); pside |= (unsigned) (q & 1) << (q >> 1); }
  }
  // the loop's flags in one word: bits 0-5 "there is a neighbour in direction q", 6-11 "... and it is a lane of this wave",
  // 12-14 "dimension d has a neighbour in another block / on another rank", 15-17 "... on another rank", 18 "this cell publishes
  // to a rank", 19-21 "on its low face of dimension d"
  const unsigned fl = nvalid | (intile << 6) | (xany << 12) | (xrm << 15) | ((REM && rem != 0) ? 1u << 18 : 0u) | (pside << 19);
  // ---- the sub-step loop -------------------------------------------------------------------------------------------
  // U and xi are accumulated in the arrays, not in registers: nothing of the temperature chain waits for them (their loads are
  // issued at the top of a sub-step and used at its end), and the four registers go to the chain -- same operations, same bits
  double Te = 0, kk = 0, ce = 1, g = 0, Tl = 1, dn = 1, src = 0;
  if (active) {
    Te = T0[c]; kk = K0[c]; ce = Ce[c]; g = fd_g[c]; Tl = md_temp[c]; dn = dens[c];
    if (SRC) src = source[c];
  }
  const double invsq[3] = {L.invhsq[0], L.invhsq[1], L.invhsq[2]};
  const MeshPre mpre = mesh_pre(L, Tl, dn);              // density and lattice temperature do not change inside an MD step
  if (!REM) { (void) tloop; }
  for (int s = 0; any_active && s < nsub; s++) {
    double xT[3], xK[3];
    double u = 0, x = 0;
    if (active) { u = ((s & 1) ? U1 : U0)[c]; x = xi[c]; }
    // (1) the neighbours in other blocks: state 0 from the arrays, later states from their records once the tags are there
    //     (cells of other ranks: every state from their records in xrec)
    const unsigned fxany = fl >> 12 & 7u, fxrm = fl >> 15 & 7u;
    unsigned todo = (s == 0) ? (fxany & fxrm) : fxany;
    if (s == 0) {
#pragma unroll
      for (int d = 0; d < 3; d++) { xT[d] = Te; xK[d] = kk; if ((fxany & ~fxrm) >> d & 1u) { xT[d] = T0[xc[d]]; xK[d] = K0[xc[d]]; } }
    } else {
#pragma unroll
      for (int d = 0; d < 3; d++) { xT[d] = Te; xK[d] = kk; }
    }
    if (s > 0 || REM) {
      const unsigned long long want = (unsigned long long) (P.base + (unsigned) s);
      const unsigned long long *R = P.rec + (size_t) (s & 1) * P.ncell * 4;
      const unsigned long long *X = REM ? P.xrec + (size_t) (s & 1) * P.xcells * 4 : nullptr;
      PRIO_POLL;
      for (int spin = 0;; spin++) {
#pragma unroll
        for (int d = 0; d < 3; d++)
          if (todo >> d & 1u) {
            const unsigned long long *r = (REM && (fxrm >> d & 1u)) ? X + (size_t) xc[d] * 4 : R + (size_t) xc[d] * 4;
            unsigned long long a0, a1, a2, a3;
            if (REM && (fxrm >> d & 1u)) { a0 = xrec_load(r); a1 = xrec_load(r + 1); a2 = xrec_load(r + 2); a3 = xrec_load(r + 3); }
            else {
              a0 = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a1 = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              a2 = __hip_atomic_load(r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a3 = __hip_atomic_load(r + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((a0 >> 32) == want && (a1 >> 32) == want && (a2 >> 32) == want && (a3 >> 32) == want) {
              xT[d] = __longlong_as_double((long long) ((a0 << 32) | (a1 & 0xffffffffull)));
              xK[d] = __longlong_as_double((long long) ((a2 << 32) | (a3 & 0xffffffffull)));
              todo &= ~(1u << d);
            }
          }
        if (__all(todo == 0)) break;
        if ((spin & 31) == 31 && __any(wall_clock64() - tloop > PERSIST_SPIN_TICKS || ld_flag(P.gate) == 3u)) {   // a neighbour never came: give up, loudly
          if (lane == 0) {
            __hip_atomic_store(P.gate, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.counters + CNT_TTM_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      PRIO_WORK;
    }
    // (2) the six flux terms in the reference's order; neighbours inside the block by lane shuffle, mirror = own values
    double flux = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int d = q >> 1, st = d == 0 ? 16 : d == 1 ? 4 : 1, srcl = lane + ((q & 1) ? -st : st);
      const double sT = __shfl(Te, srcl & 63, 64), sK = __shfl(kk, srcl & 63, 64);
      double nT = Te, nK = kk;
      if (fl >> q & 1u) {
        if (fl >> (6 + q) & 1u) { nT = sT; nK = sK; } else { nT = xT[d]; nK = xK[d]; }
      }
      const double term = (q & 1) ? ((kk + nK) / 2 * (Te - nT) * invsq[d]) : ((kk + nK) / 2 * (nT - Te) * invsq[d]);
      flux = q == 0 ? term : ((q & 1) ? flux - term : flux + term);
    }
    if (active) {
      const double Tn = tau / ce * (flux - g * (Te - Tl) + (SRC ? src : 0.0)) + Te;
      double *Tw = (s & 1) ? T0 : T1, *Kw = (s & 1) ? K0 : K1, *Uw = (s & 1) ? U0 : U1;
      u = u + (Tn - Te) * ce * L.vol / ((double) nat);
      x += (Tn - Tl) * L.xi_fac * g / (Tl * dn);
      // the arrays follow as in the launch chain (ping-pong): nobody in this launch reads them after state 0.  Stored BEFORE
      // FILLMESH, whose registers they would otherwise sit in for nothing
      Tw[c] = Tn; Uw[c] = u; xi[c] = x;
      const MeshOut o = mesh_eval(L, cve, mpre, Tn);
      ce = o.Ce; kk = o.k; g = o.g; Te = Tn;
      if ((fxany & ~fxrm) != 0 && s + 1 < nsub) {       // a lane with a neighbour in another block publishes state s + 1
        unsigned long long *r = P.rec + ((size_t) ((s + 1) & 1) * P.ncell + c) * 4;
        rec_store(r, P.base + (unsigned) (s + 1), Te);
        rec_store(r + 2, P.base + (unsigned) (s + 1), kk);
      }
      if (REM && (fl >> 18 & 1u)) {                     // ... and to the ranks beyond its faces, the final state included
#pragma unroll
        for (int d = 0; d < 3; d++)
          if (pidx[d] >= 0) {
            unsigned long long *base = (fl >> (19 + d) & 1u) ? P.peer[2 * d + 1] : P.peer[2 * d];
            unsigned long long *r = base + ((size_t) ((s + 1) & 1) * P.xcells + pidx[d]) * 4;
            xrec_store(r, P.base + (unsigned) (s + 1), Te);
            xrec_store(r + 2, P.base + (unsigned) (s + 1), kk);
          }
      }
      Kw[c] = kk;
      if (s == nsub - 1) { Zo[c] = o.Z; neo[c] = o.ne; }
    }
  }
  if (active) { Ce[c] = ce; fd_g[c] = g; }
}

// Decomposed lattice with ghost records, the two ends of a launch that do not follow the blocks of k_ttm_chain_persist (which
// cover the ACTIVE cells only).  One thread per face cell of the faces with another rank beyond them.
//  counts: every inner face cell's atom count goes out with k_ttm_activation (see there).
//  final:  ttm_fill_ghost_layers after the last sub-step (imd_ttm_3D.c:70,3314-3504): count and, for active cells, final
//          (Te, kappa) of every cell beyond my faces go into the ghost shell of the arrays (what the next step's cell
//          activation, CFL_maxdt and downloads read).  Runs behind the persistent kernel; skipped when that stepped aside.
__device__ __forceinline__ bool xface_cell(const Lat &L, const PersistDev &P, int t, int &q, int &f, int (&g)[3])
{
  q = -1;
#pragma unroll
  for (int u = 0; u < 6; u++) {
    const int d = u >> 1, nf = (L.dim[(d + 1) % 3] - 2) * (L.dim[(d + 2) % 3] - 2);
    if (q < 0 && t >= P.xoff[u] && t < P.xoff[u] + nf) q = u;
  }
  if (q < 0 || P.peer[q] == nullptr) return false;
  f = t - P.xoff[q];
  const int d = q >> 1, n2 = d == 2 ? L.dim[1] - 2 : L.dim[2] - 2, a = f / n2 + 1, b = f % n2 + 1;
  g[d] = (q & 1) ? 1 : L.dim[d] - 2;
  if (d == 0) { g[1] = a; g[2] = b; } else if (d == 1) { g[0] = a; g[2] = b; } else { g[0] = a; g[1] = b; }
  return true;
}
__global__ __launch_bounds__(64) void k_ttm_xrec_final(Lat L, PersistDev P, int nsub, int *natoms, double *Tf, double *Kf)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int q, f, g[3];
  if (t >= P.xcells || !xface_cell(L, P, t, q, f, g)) return;
  if (__hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  g[q >> 1] += (q & 1) ? -1 : 1;
  const int gc = fidx(L, g[0], g[1], g[2]);
  const unsigned long long t0 = wall_clock64();
  const unsigned long long *rn = P.xrec + (size_t) 2 * P.xcells * 4 + P.xoff[q] + f;
  const unsigned long long *r = P.xrec + ((size_t) (nsub & 1) * P.xcells + P.xoff[q] + f) * 4;
  const unsigned long long want = (unsigned long long) (P.base + (unsigned) nsub);
  int gn = -1;
  for (;;) {
    if (gn < 0) { const unsigned long long v = xrec_load(rn); if ((unsigned) (v >> 32) == P.base) { gn = (int) (unsigned) v; natoms[gc] = gn; } }
    if (gn >= 0) {
      if (gn < L.fd_min_atoms) return;
      const unsigned long long a0 = xrec_load(r), a1 = xrec_load(r + 1), a2 = xrec_load(r + 2), a3 = xrec_load(r + 3);
      if ((a0 >> 32) == want && (a1 >> 32) == want && (a2 >> 32) == want && (a3 >> 32) == want) {
        Tf[gc] = __longlong_as_double((long long) ((a0 << 32) | (a1 & 0xffffffffull)));
        Kf[gc] = __longlong_as_double((long long) ((a2 << 32) | (a3 & 0xffffffffull)));
        return;
      }
    }
    if (wall_clock64() - t0 > PERSIST_SPIN_TICKS || ld_flag(P.gate) == 3u) {          // the rank beyond never came this far: loudly
      __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.counters + CNT_TTM_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// CFL_maxdt (imd_ttm_3D.c:2637-2691): min over cells and +x,+y,+z faces
__global__ void k_ttm_cfl(Lat L, const int *natoms, const double *fd_k, const double *Ce, double *out)
{
  __shared__ double sm[16];
  const int ni = L.dim[0] - 2, nj = L.dim[1] - 2, nk = L.dim[2] - 2, n = ni * nj * nk, m = L.fd_min_atoms;
  double mx = 1e9;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    int i = t / (nk * nj) + 1, j = (t / nk) % nj + 1, k = t % nk + 1, c = fidx(L, i, j, k);
    if (!(natoms[c] > m)) continue;
    int nb[3] = {fidx(L, i + 1, j, k), fidx(L, i, j + 1, k), fidx(L, i, j, k + 1)};
    for (int d = 0; d < 3; d++) {
      int q = natoms[nb[d]] >= m ? nb[d] : c;
      double khalf = fd_k[c] + fd_k[q];
      mx = fmin(mx, Ce[c] * (L.h[d] * L.h[d]) / khalf);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmin(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) { for (int q = 1; q < (int) (blockDim.x >> 6); q++) mx = fmin(mx, sm[q]); out[0] = mx; }
}

// move_atoms_ttm (imd_integrate.c:581-811, 3D branch) + check_nblist fused:
// xi of the atom's FD cell if that cell is active, p += dt (F + xi m (p/m - vcom))
__global__ void k_move_ttm(int n, int cap, Lat L, Geo g, double4 *pos, const int *type, const int *acell,
                           double *p, double *f, const double *mass, const double *restr, int ntypes, double dt,
                           const double *ref, const int *natoms, const double *xi, const double *vcom, int ncell,
                           double *partials, const int *skip_if)
{
  // skip_if: status word of the persistent sub-step kernel of this step -- non-zero = it stepped aside and xi is not this
  // step's yet: nothing moves, the host issues the sub-steps through the launch chain and this kernel again
  if (skip_if && *skip_if != 0) return;
  __shared__ double scratch[2 * 16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s0 = 0, s1 = 0;
  if (i < n) {
    const size_t c1 = cap, c2 = 2 * (size_t) cap;
    int ix, iy, iz;
    key_to_xyz(acell[i], g, ix, iy, iz);
    int fc = fidx(L, ix / L.ext[0] + 1 + L.off[0], iy / L.ext[1] + 1 + L.off[1], iz / L.ext[2] + 1 + L.off[2]);
    double fxi = natoms[fc] >= L.fd_min_atoms ? xi[fc] : 0.0;
    double vx = vcom[fc], vy = vcom[ncell + fc], vz = vcom[2 * ncell + fc];
    int t = ntypes == 1 ? 0 : type[i];
    double fx = f[i], fy = f[c1 + i], fz = f[c2 + i];
    double px = p[i], py = p[c1 + i], pz = p[c2 + i], m = mass[i];
    double e1 = px * px + py * py + pz * pz;
    if (restr) {                                 // nullptr = every restriction factor is 1: the forces need no write-back
      fx *= restr[3 * t]; fy *= restr[3 * t + 1]; fz *= restr[3 * t + 2];
      f[i] = fx; f[c1 + i] = fy; f[c2 + i] = fz;
    }
    px += dt * (fx + fxi * m * (px / m - vx));
    py += dt * (fy + fxi * m * (py / m - vy));
    pz += dt * (fz + fxi * m * (pz / m - vz));
    double e2 = px * px + py * py + pz * pz;
    s0 = (e1 + e2) / (4 * m);
    double tmp = dt / m;
    double4 q = pos[i];
    q.x += tmp * px; q.y += tmp * py; q.z += tmp * pz;
    pos[i] = q;
    p[i] = px; p[c1 + i] = py; p[c2 + i] = pz;
    double dx = q.x - ref[i], dy = q.y - ref[c1 + i], dz = q.z - ref[c2 + i];
    s1 = dx * dx + dy * dy + dz * dz;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 = fmax(s1, __shfl_down(s1, o, 64)); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { scratch[w] = s0; scratch[16 + w] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double e = 0, mx = 0;
    for (int q = 0; q < nw; q++) { e += scratch[q]; mx = fmax(mx, scratch[16 + q]); }
    partials[(size_t) blockIdx.x * 2] = e;
    partials[(size_t) blockIdx.x * 2 + 1] = mx;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int nblk(long long n, int b) { return n <= 0 ? 1 : (int) ((n + b - 1) / b); }   // never an empty grid: a rank may own no atoms (vacuum)
// Every kernel of the lattice chain uses ONE-WAVE workgroups.  The chain runs on the second stream beside the
// force passes, whose persistent 11-wave workgroups leave no CU with room for a multi-wave workgroup (three
// waves of 150-170 VGPRs fill three of the four SIMDs): measured with tools/coresidency_probe.hip, a 256-thread
// workgroup waits for the force pass to END, a 64-thread one is placed at once on the fourth SIMD.
#define TTM_WG 64

static Lat make_lat(const TtmState *t)
{
  Lat L;
  for (int d = 0; d < 3; d++) { L.dim[d] = t->dim[d]; L.ext[d] = t->ext[d]; L.md_dim[d] = t->md_dim[d]; L.pbc[d] = t->pbc[d]; L.h[d] = t->h[d];
                                L.off[d] = t->off[d]; L.ldim[d] = t->ldim[d]; L.goff[d] = t->goff[d]; }
  L.fd_min_atoms = t->fd_min_atoms; L.vol = t->vol; L.neighvol = t->neighvol; L.atomic_weight = t->atomic_weight;
  L.atomic_charge = t->atomic_charge; L.omega_laser = t->omega_laser; L.Temin = t->Temin;
  L.inv_z0_43 = 1e-3 / pow(t->atomic_charge, 4.0 / 3.0);
  L.inv_z0am = 1e-3 / (t->atomic_charge * t->atomic_weight);
  L.inv_awamu = 1.0 / (t->atomic_weight * C_AMU);
  for (int d = 0; d < 3; d++) L.invhsq[d] = 1.0 / (t->h[d] * t->h[d]);
  L.xi_fac = 26.9815 * C_AMU / 3.0 * 1e30 / ((double) t->nsub);
  { static const int prio = exp_env_int("IMDGPU_TTM_PRIO", 0); L.wave_prio = prio; }
  return L;
}

static void imdgpu_fdtd_free(TtmState *t);
static int fdtd_lattice_loop(imdgpu_ctx *ctx, TtmState *t, const Lat &L, int steps);
void imdgpu_ttm_free(imdgpu_ctx *ctx)
{
  TtmState *t = ctx->ttm;
  if (!t) return;
  void *ptrs[] = {t->natoms, t->natoms_old, t->temp[0], t->temp[1], t->U[0], t->U[1], t->fd_k[0], t->fd_k[1], t->xi,
                  t->md_temp, t->source, t->vcom, t->dens, t->fd_g, t->Z, t->Ce, t->ne, t->cve_z, t->ee_z, t->d_red, t->flux, t->d_eabs, t->d_cand};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  for (int q = 0; q < 2; q++) if (t->graph[q]) (void) hipGraphExecDestroy(t->graph[q]);
  if (t->mb_host && t->mb_registered) (void) hipHostUnregister(t->mb_host);
  if (t->d_gbase) (void) hipFree(t->d_gbase);
  if (t->d_facecnt) (void) hipFree(t->d_facecnt);
  if (t->d_pflags) (void) hipFree(t->d_pflags);
  if (t->d_pbbox) (void) hipFree(t->d_pbbox);
  if (t->d_ufd_part) (void) hipFree(t->d_ufd_part);
  if (t->d_prec) (void) hipFree(t->d_prec);
  if (t->d_xrec) (void) hipFree(t->d_xrec);
  if (t->h_pstat) (void) hipHostFree(t->h_pstat);
  if (t->ev_pstat) (void) hipEventDestroy(t->ev_pstat);
  imdgpu_fdtd_free(t);
  for (int q = 0; q < 2; q++) if (t->eos_diag[q]) (void) hipFree(t->eos_diag[q]);
  delete t;
  ctx->ttm = nullptr;
}

extern "C" int imdgpu_ttm_init(imdgpu_ctx *ctx, const imdgpu_ttm_params *par)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->have_geometry) return imdgpu_fail(ctx, "ttm_init: set_geometry first");
  if (!ctx->pair.present) return imdgpu_fail(ctx, "ttm_init: pair_pot needed (neighvol from pair_pot.end[0])");
  if (!par->cve_z || !par->ee_z) return imdgpu_fail(ctx, "ttm_init: EOS tables missing");
  imdgpu_ttm_free(ctx);
  TtmState *t = new TtmState();
  ctx->ttm = t;
  for (int d = 0; d < 3; d++) {
    if (par->fd_ext[d] < 1 || (ctx->dim[d] - 2) % par->fd_ext[d] != 0)
      return imdgpu_fail(ctx, "cell_dim and fd_ext are not commensurate: cell_dim[%d]=%d fd_ext=%d", d, ctx->dim[d], par->fd_ext[d]);
    t->ext[d] = par->fd_ext[d]; t->md_dim[d] = ctx->dim[d]; t->pbc[d] = ctx->pbc[d];
    // The lattice is GLOBAL and replicated on every rank (each rank fills the block over its own MD cells,
    // blocks are all-gathered once per MD step, the cheap diffusion sub-steps run redundantly everywhere:
    // no per-sub-step ghost exchange).  One rank: global == local.
    t->ldim[d] = (ctx->dim[d] - 2) / par->fd_ext[d];
    t->off[d] = ctx->ttm_decomposed ? 0 : ctx->my_coord[d] * t->ldim[d];
    t->goff[d] = ctx->my_coord[d] * t->ldim[d];
    t->dim[d] = ctx->ttm_decomposed ? t->ldim[d] + 2 : t->ldim[d] * ctx->cpu_dim[d] + 2;
    // fd_h = fd_ext * (box/global_cell_dim), imd_geom_3d.c:243-246 (orthorhombic diagonal like the reference)
    t->h[d] = par->fd_ext[d] * ctx->box[4 * d] * (1.0 / ctx->gdim[d]);
  }
  t->ncell = t->dim[0] * t->dim[1] * t->dim[2];
  t->ninner = (t->dim[0] - 2) * (t->dim[1] - 2) * (t->dim[2] - 2);
  t->fd_min_atoms = par->fd_min_atoms; t->nsub = par->fd_n_timesteps;
  if (t->nsub < 1) return imdgpu_fail(ctx, "fd_n_timesteps must be >= 1");
  t->atomic_weight = par->atomic_weight; t->atomic_charge = par->atomic_charge; t->omega_laser = par->omega_laser;
  t->vol = t->h[0] * t->h[1] * t->h[2];
  t->neighvol = pow(sqrt(ctx->pair.par.end[0]), 3.0) * 4.0 / 3.0 * M_PI;
  t->max_dt_ttm = ctx->timestep / ((double) t->nsub);
  const size_t n = t->ncell;
  if (dev_alloc(ctx, &t->natoms, n) || dev_alloc(ctx, &t->natoms_old, n) || dev_alloc(ctx, &t->temp[0], n) ||
      dev_alloc(ctx, &t->temp[1], n) || dev_alloc(ctx, &t->U[0], n) || dev_alloc(ctx, &t->U[1], n) ||
      dev_alloc(ctx, &t->fd_k[0], n) || dev_alloc(ctx, &t->fd_k[1], n) || dev_alloc(ctx, &t->xi, n) ||
      dev_alloc(ctx, &t->md_temp, n) || dev_alloc(ctx, &t->source, n) || dev_alloc(ctx, &t->vcom, 3 * n) ||
      dev_alloc(ctx, &t->dens, n) || dev_alloc(ctx, &t->fd_g, n) || dev_alloc(ctx, &t->Z, n) ||
      dev_alloc(ctx, &t->Ce, n) || dev_alloc(ctx, &t->ne, n) || dev_alloc(ctx, &t->d_red, 16) || dev_alloc(ctx, &t->flux, 8 * n) || dev_alloc(ctx, &t->d_eabs, 1) ||
      dev_alloc(ctx, &t->d_cand, 2 * (size_t) t->ninner + 1))
    return 1;
  int *ip[] = {t->natoms, t->natoms_old};
  for (int *p : ip) HIPCHK(hipMemset(p, 0, n * sizeof(int)));
  double *dp[] = {t->temp[0], t->temp[1], t->U[0], t->U[1], t->fd_k[0], t->fd_k[1], t->xi, t->md_temp, t->source,
                  t->dens, t->fd_g, t->Z, t->Ce, t->ne};
  for (double *p : dp) HIPCHK(hipMemset(p, 0, n * sizeof(double)));
  HIPCHK(hipMemset(t->vcom, 0, 3 * n * sizeof(double)));
  HIPCHK(hipMemset(t->flux, 0, 8 * n * sizeof(int)));
  HIPCHK(hipMemset(t->d_eabs, 0, sizeof(double)));
  HIPCHK(hipMemset(t->d_cand, 0, (2 * (size_t) t->ninner + 1) * sizeof(int)));
  if (ctx->ttm_flux && (t->dim[0] > 1024 || t->dim[1] > 1024 || t->dim[2] > 1024))
    return imdgpu_fail(ctx, "ttm_flux: the packed FD cell index holds 10 bits per direction");
  size_t nc = (size_t) par->cve_nx * par->cve_ny, ne = (size_t) par->ee_nx * par->ee_ny;
  if (dev_alloc(ctx, &t->cve_z, nc) || dev_alloc(ctx, &t->ee_z, ne)) return 1;
  HIPCHK(hipMemcpy(t->cve_z, par->cve_z, nc * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t->ee_z, par->ee_z, ne * sizeof(double), hipMemcpyHostToDevice));
  t->cve = {par->cve_nx, par->cve_ny, par->cve_xmin, par->cve_xmax, par->cve_ymin, par->cve_ymax, t->cve_z, nullptr};
  t->ee = {par->ee_nx, par->ee_ny, par->ee_xmin, par->ee_xmax, par->ee_ymin, par->ee_ymax, t->ee_z, nullptr};
  // the look-up is the reference's: linear on the triangles nn-c builds over the table nodes (option eos_bilinear: in the grid cell)
  if (!ctx->eos_bilinear && (eos_attach_triangulation(ctx, t->cve, &t->eos_diag[0]) || eos_attach_triangulation(ctx, t->ee, &t->eos_diag[1]))) return 1;
  ctx->count_ttm = 1;
  return 0;
}

#define NEED_TTM() TtmState *t = ctx->ttm; if (!t) return imdgpu_fail(ctx, "TTM not initialised (imdgpu_ttm_init)"); \
  HIPCHK(hipSetDevice(ctx->device)); Lat L = make_lat(t); (void) L

extern "C" int imdgpu_ttm_dims(imdgpu_ctx *ctx, int dim[3], double h[3])
{
  NEED_TTM();
  for (int d = 0; d < 3; d++) { if (dim) dim[d] = t->dim[d]; if (h) h[d] = t->h[d]; }
  return 0;
}

// decomposed lattice: a face that a neighbour rank never published leaves the sticky time-out flag behind and every later
// pull/push returns early -- so every call that hands lattice results to the host (download, CFL_maxdt) reports it
static int ttm_check_mailbox(imdgpu_ctx *ctx)
{
  TtmState *t = ctx->ttm;
  if (!t || (!t->mb_host && t->persist <= 0)) return 0;
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, ctx->d_counters + CNT_TTM_TIMEOUT, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (flag) {
    // reported once: the caller may switch the persistent kernel off (option ttm_persist) or detach the remote records and
    // go on from a lattice it uploads again -- the flag must not fail every later call of this context
    (void) hipMemsetAsync(ctx->d_counters + CNT_TTM_TIMEOUT, 0, sizeof(int), ctx->stream);
    return imdgpu_fail(ctx, "TTM lattice: a sub-step hand-off timed out (mailbox: a neighbour rank did not publish its faces; or the "
                            "persistent sub-step kernel's hand-offs stalled, option ttm_persist = 0 disables it); the lattice of this step is not valid");
  }
  return 0;
}

static int ttm_xfer(imdgpu_ctx *ctx, const imdgpu_ttm_fields *f, bool down)
{
  NEED_TTM();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->step_open) HIPCHK(hipStreamSynchronize(ctx->stream2));   // between md_step_begin and _end the lattice chain runs on the side stream
  if (!ctx->step_open && imdgpu_ttm_persist_settle(ctx) < 0) return 1;
  if (down && ttm_check_mailbox(ctx)) return 1;
  const size_t n = t->ncell;
  auto X = [&](void *host, void *dev, size_t bytes) -> int {
    if (!host) return 0;
    if (down) HIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    else HIPCHK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (X(f->natoms, t->natoms, n * 4) || X(f->natoms_old, t->natoms_old, n * 4) || X(f->temp, t->temp[t->cur], n * 8) ||
      X(f->temp2, t->temp[t->cur ^ 1], n * 8) || X(f->xi, t->xi, n * 8) || X(f->md_temp, t->md_temp, n * 8) ||
      X(f->source, t->source, n * 8) || X(f->dens, t->dens, n * 8) || X(f->fd_k, t->fd_k[t->kcur], n * 8) ||
      X(f->fd_g, t->fd_g, n * 8) || X(f->Z, t->Z, n * 8) || X(f->Ce, t->Ce, n * 8) || X(f->ne, t->ne, n * 8) ||
      X(f->U, t->U[t->cur], n * 8) || X(f->U2, t->U[t->cur ^ 1], n * 8))
    return 1;
  if (f->flux) {                       // 8 planes <-> [cell][8] like ttm_Element.flux
    std::vector<int> h(8 * n);
    if (down) {
      HIPCHK(hipMemcpy(h.data(), t->flux, 8 * n * 4, hipMemcpyDeviceToHost));
      for (size_t c = 0; c < n; c++) for (int q = 0; q < 8; q++) f->flux[8 * c + q] = h[q * n + c];
    } else {
      for (size_t c = 0; c < n; c++) for (int q = 0; q < 8; q++) h[q * n + c] = f->flux[8 * c + q];
      HIPCHK(hipMemcpy(t->flux, h.data(), 8 * n * 4, hipMemcpyHostToDevice));
      t->flux_zero = 0;
    }
  }
  if (f->vcom) {                       // planes <-> interleaved
    std::vector<double> h(3 * n);
    if (down) {
      HIPCHK(hipMemcpy(h.data(), t->vcom, 3 * n * 8, hipMemcpyDeviceToHost));
      for (size_t c = 0; c < n; c++) for (int d = 0; d < 3; d++) f->vcom[3 * c + d] = h[d * n + c];
    } else {
      for (size_t c = 0; c < n; c++) for (int d = 0; d < 3; d++) h[d * n + c] = f->vcom[3 * c + d];
      HIPCHK(hipMemcpy(t->vcom, h.data(), 3 * n * 8, hipMemcpyHostToDevice));
    }
  }
  return 0;
}
extern "C" int imdgpu_ttm_download(imdgpu_ctx *ctx, const imdgpu_ttm_fields *f) { return ttm_xfer(ctx, f, true); }
extern "C" int imdgpu_ttm_upload(imdgpu_ctx *ctx, const imdgpu_ttm_fields *f)
{
  if (f->source && ctx->ttm) ctx->ttm->have_source = 1;
  return ttm_xfer(ctx, f, false);
}

// update_fd for FD cells over THOUSANDS of MD cells (a one-row lattice spans the whole cross-section of the sample:
// 78 x 78 MD cells per FD cell at config 5): one workgroup per FD cell (k_ttm_update_fd_wide) leaves most of the chip idle
// and walks 6000 cells per workgroup, latency bound (1.1 ms at 10 M atoms).  Here every FD cell is cut into UFD_CHUNKS
// chunks of MD cells, a four-wave workgroup per chunk writes partial sums, and a one-wave kernel per FD cell adds the
// chunks in a fixed order (deterministic); two sweeps as in the reference (the second needs v_com), so four launches.
#define UFD_CHUNKS 64
template <int PHASE>
__global__ __launch_bounds__(256) void k_ttm_ufd_part(Lat L, int cap, const int *cell_start, const int *cell_cnt, const int *numneighs,
                                                      const double *p, const double *mass, const double *vcom, int ncell, double *part)
{
  __shared__ double sd[6][4];
  const int w = blockIdx.x / UFD_CHUNKS, ch = blockIdx.x % UFD_CHUNKS, nj = L.ldim[1], nk = L.ldim[2];
  const int k = w % nk + 1, j = (w / nk) % nj + 1, i = w / (nk * nj) + 1;
  const int c = fidx(L, i + L.off[0], j + L.off[1], k + L.off[2]);
  const int e0 = L.ext[0], e1 = L.ext[1], e2 = L.ext[2], ne = e0 * e1 * e2;
  const int per = (ne + UFD_CHUNKS - 1) / UFD_CHUNKS, qa = ch * per, qb = min(ne, qa + per);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto md_cell = [&](int q) {
    const int xc = (i - 1) * e0 + 1 + q / (e2 * e1), yc = (j - 1) * e1 + 1 + (q / e2) % e1, zc = (k - 1) * e2 + 1 + q % e2;
    return (xc * L.md_dim[1] + yc) * L.md_dim[2] + zc;
  };
  double v[6] = {0, 0, 0, 0, 0, 0};                     // phase 1: natoms, sum numneighs, sum p (3), sum m; phase 2: sum m |v - v_com|^2
  double vx = 0, vy = 0, vz = 0;
  if (PHASE == 2) { vx = vcom[c]; vy = vcom[ncell + c]; vz = vcom[2 * ncell + c]; }
  // a wave walks three cells at a time: lane l takes atom l % 21 of cell l / 21 (cells hold ~21 atoms; lane 63 idles), longer
  // cells finish in extra trips
  for (int q0 = qa + wv * 3; q0 < qb; q0 += 4 * 3) {
    const int sub = lane / 21, a0 = lane % 21, q = q0 + sub;
    if (sub < 3 && q < qb) {
      const int m = md_cell(q), s0 = cell_start[m], n0 = cell_cnt[m];
      if (PHASE == 1 && a0 == 0) v[0] += n0;
      for (int a = a0; a < n0; a += 21) {
        const int g = s0 + a;
        if (PHASE == 1) { v[1] += numneighs[g]; v[2] += p[g]; v[3] += p[cap + g]; v[4] += p[2 * (size_t) cap + g]; v[5] += mass[g]; }
        else {
          const double mm = mass[g];
          double u;
          u = p[g] / mm - vx; v[0] += mm * (u * u);
          u = p[cap + g] / mm - vy; v[0] += mm * (u * u);
          u = p[2 * (size_t) cap + g] / mm - vz; v[0] += mm * (u * u);
        }
      }
    }
  }
  const int NV = PHASE == 1 ? 6 : 1;
#pragma unroll
  for (int q = 0; q < NV; q++) { v[q] = wsum(v[q]); if (lane == 0) sd[q][wv] = v[q]; }
  __syncthreads();
  if (threadIdx.x < NV) part[((size_t) blockIdx.x) * 6 + threadIdx.x] = ((sd[threadIdx.x][0] + sd[threadIdx.x][1]) + sd[threadIdx.x][2]) + sd[threadIdx.x][3];
}

template <int PHASE>
__global__ __launch_bounds__(64) void k_ttm_ufd_combine(Lat L, int steps, const double *part, int *natoms, int *natoms_old, double *temp0,
                                                       double *temp1, double *xi, double *md_temp, double *source, double *vcom,
                                                       double *dens, int ncell)
{
  const int w = blockIdx.x, lane = threadIdx.x, nj = L.ldim[1], nk = L.ldim[2];
  const int k = w % nk + 1, j = (w / nk) % nj + 1, i = w / (nk * nj) + 1;
  const int c = fidx(L, i + L.off[0], j + L.off[1], k + L.off[2]);
  double v[6];
#pragma unroll
  for (int q = 0; q < 6; q++) v[q] = (lane < UFD_CHUNKS && (PHASE == 1 || q == 0)) ? part[((size_t) w * UFD_CHUNKS + lane) * 6 + q] : 0.0;
#pragma unroll
  for (int q = 0; q < (PHASE == 1 ? 6 : 1); q++) v[q] = wsum(v[q]);
  if (lane != 0) return;
  if (PHASE == 1) {
    const int N = (int) v[0];
    double d = 0, vx = 0, vy = 0, vz = 0;
    if (N > 0) {
      d = v[1] / ((double) N) * L.atomic_weight / L.neighvol * 1660.53907;
      if (d == 0) d = (double) N * L.atomic_weight / L.vol * 1660.53907;
      vx = v[2] / v[5]; vy = v[3] / v[5]; vz = v[4] / v[5];
    }
    const int old = natoms[c];
    natoms_old[c] = (steps < 1) ? N : old;
    natoms[c] = N;
    xi[c] = 0.0; source[c] = 0.0; dens[c] = d;
    vcom[c] = vx; vcom[ncell + c] = vy; vcom[2 * ncell + c] = vz;
  } else {
    const int nat = natoms[c];
    double T = v[0];
    if (nat >= L.fd_min_atoms) T /= 3.0 * nat; else T = 0.0;
    md_temp[c] = T;
    if (steps < 1 && nat >= L.fd_min_atoms) { temp0[c] = T; temp1[c] = T; }
  }
}

extern "C" int imdgpu_ttm_update_fd(imdgpu_ctx *ctx, int steps)
{
  NEED_TTM();
  if (imdgpu_ttm_persist_settle(ctx) < 0) return 1;      // the previous step's persistent launch, if nobody has looked yet (no-op inside imdgpu_md_step)
  ProfScope ps(ctx, PROF_TTM_UPDATE);
  if (t->ext[0] * t->ext[1] * t->ext[2] > 64) {             // FD cells over many MD cells: a workgroup per FD cell
    if (ctx->ttm_flux) return imdgpu_fail(ctx, "update_fd: the flux bookkeeping (ttm_flux / ttm_adv) is not available for FD cells of more than 64 MD cells");
    const int nfd = t->ldim[0] * t->ldim[1] * t->ldim[2];
    if (t->ext[0] * t->ext[1] * t->ext[2] >= 2048) {         // thousands of MD cells per FD cell: chunked, four launches
      if (t->ufd_part_cap < (size_t) nfd * UFD_CHUNKS * 6) {
        if (dev_alloc(ctx, &t->d_ufd_part, (size_t) nfd * UFD_CHUNKS * 6)) return 1;
        t->ufd_part_cap = (size_t) nfd * UFD_CHUNKS * 6;
      }
      hipLaunchKernelGGL((k_ttm_ufd_part<1>), dim3(nfd * UFD_CHUNKS), dim3(256), 0, ctx->stream, L, ctx->cap_local, ctx->cell_start,
                         ctx->cell_cnt, ctx->numneighs, ctx->p, ctx->mass, t->vcom, t->ncell, t->d_ufd_part);
      hipLaunchKernelGGL((k_ttm_ufd_combine<1>), dim3(nfd), dim3(64), 0, ctx->stream, L, steps, t->d_ufd_part, t->natoms, t->natoms_old,
                         t->temp[0], t->temp[1], t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell);
      hipLaunchKernelGGL((k_ttm_ufd_part<2>), dim3(nfd * UFD_CHUNKS), dim3(256), 0, ctx->stream, L, ctx->cap_local, ctx->cell_start,
                         ctx->cell_cnt, ctx->numneighs, ctx->p, ctx->mass, t->vcom, t->ncell, t->d_ufd_part);
      hipLaunchKernelGGL((k_ttm_ufd_combine<2>), dim3(nfd), dim3(64), 0, ctx->stream, L, steps, t->d_ufd_part, t->natoms, t->natoms_old,
                         t->temp[0], t->temp[1], t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell);
      HIPCHK(hipGetLastError());
      return 0;
    }
    hipLaunchKernelGGL(k_ttm_update_fd_wide, dim3(t->ldim[0] * t->ldim[1] * t->ldim[2]), dim3(1024), 0, ctx->stream, L, steps,
                       ctx->cap_local, ctx->cell_start, ctx->cell_cnt, ctx->numneighs, ctx->p, ctx->mass, t->natoms,
                       t->natoms_old, t->temp[0], t->temp[1], t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // The flux bookkeeping (imd_ttm_3D.c:335-370) compares the FD cell an atom is counted in with the one it was counted in last
  // time.  Atoms are counted by the MD cell they are binned in, and between two re-binnings nobody changes cells: every counter
  // is zero and every atom's memory would be rewritten with what it holds.  So only the first update_fd behind a re-binning (or
  // an upload, a migration, a change of the option) does that part -- 12 of the 52 bytes per atom the kernel moves, and its
  // eight reductions per FD cell; the others leave the counters at zero.
  const bool track = ctx->ttm_flux != 0, redo = track && (t->fdc_epoch != ctx->cells_epoch || steps < 1);
  if (track && !redo && !t->flux_zero) {
    HIPCHK(hipMemsetAsync(t->flux, 0, 8 * (size_t) t->ncell * sizeof(int), ctx->stream));
    t->flux_zero = 1;
  }
  hipLaunchKernelGGL(k_ttm_update_fd, dim3(nblk((long long) t->ldim[0] * t->ldim[1] * t->ldim[2] * 64, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, steps,
                     ctx->cap_local, ctx->cell_start, ctx->cell_cnt, ctx->numneighs, ctx->p, ctx->mass, t->natoms,
                     t->natoms_old, t->temp[0], t->temp[1], t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell,
                     (double *) nullptr, redo ? ctx->fdc : (int *) nullptr, redo ? t->flux : (int *) nullptr);
  HIPCHK(hipGetLastError());
  if (redo) { t->fdc_epoch = ctx->cells_epoch; t->flux_zero = 0; }
  return 0;
}

// multi-rank: this rank's block of update_fd results -> dev_out (6 doubles per local FD cell; 10 with option ttm_flux:
// the eight flux counters follow as ints; imdgpu_get_info "ttm_block_width")
extern "C" int imdgpu_ttm_local_block(imdgpu_ctx *ctx, int steps, double *dev_out, int *ncells_block)
{
  NEED_TTM();
  const int nb = t->ldim[0] * t->ldim[1] * t->ldim[2];
  if (ncells_block) *ncells_block = nb;
  if (!dev_out) return 0;
  if (imdgpu_ttm_persist_settle(ctx) < 0) return 1;
  ProfScope ps(ctx, PROF_TTM_UPDATE);
  hipLaunchKernelGGL(k_ttm_update_fd, dim3(nblk((long long) nb * 64, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, steps,
                     ctx->cap_local, ctx->cell_start, ctx->cell_cnt, ctx->numneighs, ctx->p, ctx->mass, t->natoms,
                     t->natoms_old, t->temp[0], t->temp[1], t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell, dev_out,
                     ctx->ttm_flux ? ctx->fdc : (int *) nullptr, (int *) nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// multi-rank: fold the all-gathered blocks of all ranks into the replicated lattice.
// dev_in: nblocks * ncells_block * 6 (10 with ttm_flux) doubles in rank order; dev_offs: 3 ints per block (FD-cell offset of that rank)
extern "C" int imdgpu_ttm_apply_blocks(imdgpu_ctx *ctx, int steps, const double *dev_in, int nblocks, const int *dev_offs)
{
  NEED_TTM();
  const int per = t->ldim[0] * t->ldim[1] * t->ldim[2];
  hipLaunchKernelGGL(k_ttm_apply_blocks, dim3(nblk((long long) nblocks * per, 256)), dim3(256), 0, ctx->stream, L, steps, nblocks,
                     dev_offs, t->ldim[0], t->ldim[1], t->ldim[2], dev_in, t->natoms, t->natoms_old, t->temp[0], t->temp[1],
                     t->xi, t->md_temp, t->source, t->vcom, t->dens, t->ncell, ctx->ttm_flux ? t->flux : (int *) nullptr);
  HIPCHK(hipGetLastError());
  t->flux_zero = 0;
  return 0;
}

// EOS entry points on host arrays (EOS_ee_from_r_te is part of FILLMESH/activation; the inverse is used by do_ADV)
extern "C" int imdgpu_ttm_eos_te_from_r_ee(imdgpu_ctx *ctx, int n, const double *r, const double *e, double *te_out)
{
  NEED_TTM();
  if (n <= 0) return 0;
  double *d = nullptr;
  HIPCHK(hipMalloc((void **) &d, (size_t) 3 * n * sizeof(double)));
  HIPCHK(hipMemcpy(d, r, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + n, e, n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ttm_te_from_r_ee, dim3(nblk(n, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, n, t->ee, d, d + n, d + 2 * (size_t) n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(te_out, d + 2 * (size_t) n, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipFree(d));
  return 0;
}

// ---- which diagonal the reference's triangulation puts into every cell of a table grid -------------------------------------
// The reference interpolates its EOS tables with nn-c (v1.86.2, shipped as the binary nn_interpol/libnn.a only): linear on a
// Delaunay triangulation that nn-c's delaunay_build gets from J. R. Shewchuk's Triangle -- divide-and-conquer (Guibas &
// Stolfi) with alternating cuts (Dwyer), exact predicates.  On a rectilinear grid every cell is a cocircular quadrilateral, so
// the diagonals are decided by that algorithm's tie rules alone -- not by the coordinates: any spacing, uniform or not, gives
// the same map (measured against libnn.a on 96 grids, tests/golden/eos_triangulation.npz).  Restated here from the published
// algorithm on the integer grid (i, j):
//   * vertices sorted by (x, y); the array is cut at n/2 BY COUNT, the halves re-partitioned along the other axis (ties by the
//     first), recursively, down to sets of two or three, which are ordered by (x, y) again [Triangle: alternateaxes];
//   * two vertices: an edge; three: a triangle, or two edges when collinear;
//   * merge of a left / right pair (cut along x): handles = the (x, y)-smallest and -largest hull vertex of each half; of a
//     lower / upper pair (cut along y): the (y, x)-smallest and -largest [Triangle shifts its handles from the x extremes to the
//     y extremes by walking the hull while the next vertex is STRICTLY lower / higher: the first end of the bottom / top edge];
//     lower common tangent with strict orientation tests; zip upwards; a candidate edge goes only if the next candidate's far
//     vertex is STRICTLY inside the circle (incircle > 0), and of two valid candidates the right one is taken only if the left
//     candidate's circle STRICTLY contains the right candidate's vertex -- a tie goes to the left candidate.
// Quad-edge structure after Guibas & Stolfi; predicates in integers (exact).  diag as imdgpu_ttm_set_eos_triangulation takes it.
namespace {
struct GridDelaunay {
  typedef long long I;
  int ny;
  std::vector<int> onext, org;            // directed edge e = 4 * quad + r; org of the primal edges (r = 0, 2) = vertex id i * ny + j
  std::vector<char> alive;
  static int rot(int e) { return (e & ~3) | ((e + 1) & 3); }
  static int sym(int e) { return (e & ~3) | ((e + 2) & 3); }
  static int irot(int e) { return (e & ~3) | ((e + 3) & 3); }
  int dest(int e) const { return org[sym(e)]; }
  int oprev(int e) const { return rot(onext[rot(e)]); }
  int lnext(int e) const { return rot(onext[irot(e)]); }
  int rprev(int e) const { return onext[sym(e)]; }
  int make_edge(int a, int b)
  {
    const int e = (int) onext.size();
    onext.resize(e + 4); org.resize(e + 4, -1); alive.push_back(1);
    onext[e] = e; onext[e + 2] = e + 2; onext[e + 1] = e + 3; onext[e + 3] = e + 1;
    org[e] = a; org[e + 2] = b;
    return e;
  }
  void splice(int a, int b)
  {
    const int al = rot(onext[a]), be = rot(onext[b]);
    std::swap(onext[a], onext[b]);
    std::swap(onext[al], onext[be]);
  }
  int connect(int a, int b)
  {
    const int e = make_edge(dest(a), org[b]);
    splice(e, lnext(a));
    splice(sym(e), b);
    return e;
  }
  void kill(int e)
  {
    splice(e, oprev(e));
    splice(sym(e), oprev(sym(e)));
    alive[e >> 2] = 0;
  }
  I X(int v) const { return v / ny; }
  I Y(int v) const { return v % ny; }
  I ccw(int a, int b, int c) const { return (X(b) - X(a)) * (Y(c) - Y(a)) - (Y(b) - Y(a)) * (X(c) - X(a)); }
  __int128 incircle(int a, int b, int c, int d) const
  {
    const I adx = X(a) - X(d), ady = Y(a) - Y(d), bdx = X(b) - X(d), bdy = Y(b) - Y(d), cdx = X(c) - X(d), cdy = Y(c) - Y(d);
    const __int128 ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return (__int128) adx * (bdy * cd - bd * cdy) - (__int128) ady * (bdx * cd - bd * cdx) + ad * (__int128) (bdx * cdy - bdy * cdx);
  }
  bool rightof(int p, int e) const { return ccw(p, dest(e), org[e]) > 0; }
  bool leftof(int p, int e) const { return ccw(p, org[e], dest(e)) > 0; }
  bool less_axis(int a, int b, int axis) const
  {
    const I ka = axis ? Y(a) : X(a), kb = axis ? Y(b) : X(b);
    if (ka != kb) return ka < kb;
    return (axis ? X(a) : Y(a)) < (axis ? X(b) : Y(b));
  }
  void arrange(int *p, int n, int axis)
  {
    if (n <= 3) { std::sort(p, p + n, [&](int a, int b) { return less_axis(a, b, 0); }); return; }
    std::sort(p, p + n, [&](int a, int b) { return less_axis(a, b, axis); });
    const int d = n >> 1;
    if (n - d >= 2) { if (d >= 2) arrange(p, d, 1 - axis); arrange(p + d, n - d, 1 - axis); }
  }
  // ccw hull edge out of the frame's smallest hull vertex, cw hull edge out of its largest, from any ccw hull edge h
  void handles(int h, int axis, int *lo, int *hi) const
  {
    int best_lo = h, best_hi = h, e = h;
    do {
      if (less_axis(org[e], org[best_lo], axis)) best_lo = e;
      if (less_axis(dest(best_hi), dest(e), axis)) best_hi = e;
      e = rprev(e);
    } while (e != h);
    *lo = best_lo; *hi = sym(best_hi);
  }
  int build(const int *p, int n, int axis)          // returns a ccw hull edge (the outside on its right)
  {
    if (n == 2) return make_edge(p[0], p[1]);
    if (n == 3) {
      const int a = make_edge(p[0], p[1]), b = make_edge(p[1], p[2]);
      splice(sym(a), b);
      const I area = ccw(p[0], p[1], p[2]);
      if (area > 0) { connect(b, a); return a; }
      if (area < 0) { const int c = connect(b, a); return sym(c); }
      return a;
    }
    const int d = n >> 1;
    const int hl = build(p, d, 1 - axis), hr = build(p + d, n - d, 1 - axis);
    int ldo, ldi, rdi, rdo;
    handles(hl, axis, &ldo, &ldi);
    handles(hr, axis, &rdi, &rdo);
    for (;;) {                                      // lower common tangent
      if (leftof(org[rdi], ldi)) ldi = lnext(ldi);
      else if (rightof(org[ldi], rdi)) rdi = rprev(rdi);
      else break;
    }
    int basel = connect(sym(rdi), ldi);
    if (org[ldi] == org[ldo]) ldo = sym(basel);
    for (;;) {
      int lcand = onext[sym(basel)];
      if (rightof(dest(lcand), basel))
        while (incircle(dest(basel), org[basel], dest(lcand), dest(onext[lcand])) > 0) { const int t2 = onext[lcand]; kill(lcand); lcand = t2; }
      int rcand = oprev(basel);
      if (rightof(dest(rcand), basel))
        while (incircle(dest(basel), org[basel], dest(rcand), dest(oprev(rcand))) > 0) { const int t2 = oprev(rcand); kill(rcand); rcand = t2; }
      const bool lv = rightof(dest(lcand), basel), rv = rightof(dest(rcand), basel);
      if (!lv && !rv) break;
      if (!lv || (rv && incircle(dest(lcand), org[lcand], org[rcand], dest(rcand)) > 0)) basel = connect(rcand, sym(basel));
      else basel = connect(sym(basel), sym(lcand));
    }
    return ldo;
  }
};
}
// diag[(nx-1)*(ny-1)], cell (ix, iy) at ix*(ny-1)+iy: 1 = split along (ix+1,iy)-(ix,iy+1), 2 = along (ix,iy)-(ix+1,iy+1).  Host only.
extern "C" int imdgpu_eos_triangulation(int nx, int ny, unsigned char *diag)
{
  if (nx < 2 || ny < 2 || !diag || (long long) nx * ny > 50000000LL) return 1;
  GridDelaunay G;
  G.ny = ny;
  const int n = nx * ny;
  G.onext.reserve(16 * (size_t) n); G.org.reserve(16 * (size_t) n); G.alive.reserve(4 * (size_t) n);
  std::vector<int> p(n);
  for (int q = 0; q < n; q++) p[q] = q;             // (x, y) order already
  const int d = n >> 1;
  if (n - d >= 2) { if (d >= 2) G.arrange(p.data(), d, 1); G.arrange(p.data() + d, n - d, 1); }
  G.build(p.data(), n, 0);
  memset(diag, 0, (size_t) (nx - 1) * (ny - 1));
  for (size_t k = 0; k < G.alive.size(); k++) {
    if (!G.alive[k]) continue;
    int a = G.org[4 * k], b = G.org[4 * k + 2];
    if (a / ny > b / ny) std::swap(a, b);
    const int ia = a / ny, ja = a % ny, ib = b / ny, jb = b % ny;
    if (ib == ia + 1 && jb == ja + 1) diag[(size_t) ia * (ny - 1) + ja] = 2;
    else if (ib == ia + 1 && jb == ja - 1) diag[(size_t) ia * (ny - 1) + jb] = 1;
  }
  for (size_t q = 0; q < (size_t) (nx - 1) * (ny - 1); q++) if (!diag[q]) return 2;      // every cell of a Delaunay triangulation of a grid is split
  return 0;
}

// which = 0: cve table, 1: ee table; diag: (nx-1)*(ny-1) bytes, cell (ix, iy) at ix*(ny-1)+iy: 1 = the reference's triangulation
// splits the cell along (ix+1,iy)-(ix,iy+1), 2 = along (ix,iy)-(ix+1,iy+1), 0 = unknown (bilinear); NULL: back to bilinear
extern "C" int imdgpu_ttm_set_eos_triangulation(imdgpu_ctx *ctx, int which, const unsigned char *diag)
{
  NEED_TTM();
  if (which != 0 && which != 1) return imdgpu_fail(ctx, "ttm_set_eos_triangulation: which must be 0 (cve) or 1 (ee)");
  EosDev &e = which ? t->ee : t->cve;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (t->eos_diag[which]) { (void) hipFree(t->eos_diag[which]); t->eos_diag[which] = nullptr; }
  e.diag = nullptr;
  if (diag) {
    const size_t n = (size_t) (e.nx - 1) * (e.ny - 1);
    for (size_t q = 0; q < n; q++) if (diag[q] > 2) return imdgpu_fail(ctx, "ttm_set_eos_triangulation: entries must be 0, 1 or 2");
    HIPCHK(hipMalloc((void **) &t->eos_diag[which], n));
    HIPCHK(hipMemcpy(t->eos_diag[which], diag, n, hipMemcpyHostToDevice));
    e.diag = t->eos_diag[which];
  }
  for (int q = 0; q < 2; q++) if (t->graph[q]) { (void) hipGraphExecDestroy(t->graph[q]); t->graph[q] = nullptr; }   // captured kernel arguments
  return 0;
}

// the raw table look-up both EOS functions are built on (the reference: lpi_interpolate_point, imd_ttm_3D.c:2722,2759),
// without unit factors; which = 0: cve(r, T) table, 1: ee(r, sqrt T) table; (x, y) in the table's own coordinates
extern "C" int imdgpu_ttm_eos_lookup(imdgpu_ctx *ctx, int which, int n, const double *x, const double *y, double *z_out)
{
  NEED_TTM();
  if (which != 0 && which != 1) return imdgpu_fail(ctx, "ttm_eos_lookup: which must be 0 (cve) or 1 (ee)");
  if (n <= 0) return 0;
  double *d = nullptr;
  HIPCHK(hipMalloc((void **) &d, (size_t) 3 * n * sizeof(double)));
  HIPCHK(hipMemcpy(d, x, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + n, y, n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ttm_eos_lookup, dim3(nblk(n, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, n, which ? t->ee : t->cve, d, d + n,
                     d + 2 * (size_t) n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(z_out, d + 2 * (size_t) n, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipFree(d));
  return 0;
}

// [laser_rescale_ttm, imd_laser.c:715-788]: writes the source term of the current MD step (time = timestep * steps)
extern "C" int imdgpu_ttm_laser_rescale(imdgpu_ctx *ctx, const imdgpu_laser_params *p, int steps)
{
  NEED_TTM();
  if (!p) return imdgpu_fail(ctx, "ttm_laser_rescale: no parameters");
  double g = ctx->timestep * steps - p->t_0;
  g *= g;
  const double e = exp(-g / p->sigma_t_squared / 2.0) * p->p_peak;
  int go[3];
  for (int d = 0; d < 3; d++) go[d] = ctx->ttm_decomposed ? t->goff[d] : 0;
  hipLaunchKernelGGL(k_ttm_laser, dim3(nblk(t->ninner, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, go[0], go[1], go[2],
                     p->dir[0], p->dir[1], p->dir[2], p->offset, p->mu, e, t->natoms, t->source);
  HIPCHK(hipGetLastError());
  t->have_source = 1;
  t->laser_active = !(e < p->p_peak * 0.001);          // imd_laser.c:744-747
  return 0;
}

// imdgpu_calc_ttm / imdgpu_ttm_lattice_step apply the laser after update_fd from now on (p == NULL: off)
extern "C" int imdgpu_ttm_set_laser(imdgpu_ctx *ctx, const imdgpu_laser_params *p)
{
  NEED_TTM();
  t->laser_on = p != nullptr;
  if (p) { t->laser = *p; t->have_source = 1; }
  return 0;
}

// [do_COMMFLUX, imd_ttm_3D.c:667-995] one x face of {flux[8], U} of the current lattice: dev_buf holds 5 doubles per
// inner (j, k).  unpack == 0: side 0 packs my column i = 1, side 1 my column lx-2; unpack == 1: side 0 fills my ghost
// column i = 0, side 1 the ghost column lx-1.  *ncells receives the number of (j, k) pairs.
extern "C" int imdgpu_ttm_commflux(imdgpu_ctx *ctx, int side, int unpack, double *dev_buf, int *ncells)
{
  NEED_TTM();
  const int n = (t->dim[1] - 2) * (t->dim[2] - 2);
  if (ncells) *ncells = n;
  if (!dev_buf) return 0;
  if (!ctx->ttm_flux) return imdgpu_fail(ctx, "ttm_commflux: option ttm_flux / ttm_adv is off");
  hipLaunchKernelGGL(k_ttm_commflux, dim3(nblk(n, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, side, unpack, t->flux, t->ncell,
                     t->U[t->cur], dev_buf);
  HIPCHK(hipGetLastError());
  if (unpack) t->flux_zero = 0;               // a neighbour's counters in the ghost column
  return 0;
}

// [do_ADV(tau), imd_ttm_3D.c:1321-1458; calc_ttm calls it with tau = 1.0 right after update_fd (:57-60)]
extern "C" int imdgpu_ttm_adv(imdgpu_ctx *ctx, double tau, int steps)
{
  NEED_TTM();
  if (!ctx->ttm_flux) return imdgpu_fail(ctx, "ttm_adv: needs the atom-flux counters of update_fd (option ttm_flux or ttm_adv)");
  if (t->dim[2] < 3) return imdgpu_fail(ctx, "ttm_adv: lattice has no plane k = 1");
  if (steps < 1) return 0;                               // :1323
  // replicated lattice of several ranks: the plane k = 1 of every rank's block along z (see k_ttm_adv)
  const int nplanes = (ctx->external_halo && !ctx->ttm_decomposed) ? ctx->cpu_dim[2] : 1;
  const int n = (t->dim[0] - 2) * (t->dim[1] - 2) * nplanes;
  hipLaunchKernelGGL(k_ttm_adv, dim3(nblk(n, 64)), dim3(64), 0, ctx->stream, L, t->ee, tau, t->natoms, t->natoms_old,
                     t->flux, t->ncell, t->dens, t->U[t->cur], t->U[t->cur ^ 1], t->temp[t->cur ^ 1], nplanes, t->ldim[2]);
  HIPCHK(hipGetLastError());
  t->cur ^= 1;                                           // l1 <-> l2 as a whole, :1454-1456
  return 0;
}

static int ttm_cell_activation(imdgpu_ctx *ctx, TtmState *t, const Lat &L, bool publish_counts)
{
  XPub X{};
  if (publish_counts) {
    for (int q = 0; q < 6; q++) { X.peer[q] = t->peer[q]; X.xoff[q] = t->xoff[q]; }
    X.xcells = t->xcells; X.base = t->persist_base;        // the base of the persistent launch that follows in this step
  }
  hipLaunchKernelGGL(k_ttm_activation, dim3(nblk(t->ninner, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, t->natoms,
                     t->natoms_old, t->temp[t->cur], t->xi, t->d_cand, t->d_cand + 2 * (size_t) t->ninner, X);
  hipLaunchKernelGGL(k_ttm_activation_seq, dim3(1), dim3(64), 0, ctx->stream, L, t->ee, t->natoms, t->temp[t->cur],
                     t->temp[t->cur ^ 1], t->U[t->cur], t->U[t->cur ^ 1], t->md_temp, t->dens, t->d_cand,
                     t->d_cand + t->ninner, t->d_cand + 2 * (size_t) t->ninner,
                     ctx->act_ranksort_max);
  HIPCHK(hipGetLastError());
  return 0;
}
int imdgpu_ttm_experiment_mask()
{
  int m = 0;
#ifdef PERSIST_PRIO
  m |= 256;
#endif
  return m;
}
extern "C" int imdgpu_ttm_cell_activation(imdgpu_ctx *ctx)
{
  NEED_TTM();
  return ttm_cell_activation(ctx, t, L, false);
}

static int ttm_fillmesh(imdgpu_ctx *ctx, TtmState *t, const Lat &L, int steps, int *bbox)
{
  ProfScope ps(ctx, PROF_TTM_FILL);
  hipLaunchKernelGGL(k_ttm_fillmesh, dim3(nblk(t->ninner, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, steps, t->cve, t->ee,
                     t->natoms, t->temp[t->cur], t->md_temp, t->dens, t->Z, t->ne, t->Ce, t->fd_k[t->kcur], t->fd_g,
                     t->U[t->cur], t->U[t->cur ^ 1], bbox);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int imdgpu_ttm_fillmesh(imdgpu_ctx *ctx, int steps)
{
  NEED_TTM();
  return ttm_fillmesh(ctx, t, L, steps, nullptr);
}

extern "C" int imdgpu_ttm_fill_ghost_layers(imdgpu_ctx *ctx)
{
  NEED_TTM();
  int nf = (t->dim[1] - 2) * (t->dim[2] - 2) + (t->dim[2] - 2) * (t->dim[0] - 2) + (t->dim[0] - 2) * (t->dim[1] - 2);
  hipLaunchKernelGGL(k_ttm_ghosts, dim3(nblk(nf, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, t->natoms, t->temp[t->cur],
                     t->fd_k[t->kcur]);
  HIPCHK(hipGetLastError());
  return 0;
}

static int diff_grid(const TtmState *t)
{
  return ((t->dim[0] - 2 + DT_X - 1) / DT_X) * ((t->dim[1] - 2 + DT_Y - 1) / DT_Y) * ((t->dim[2] - 2 + DT_Z - 1) / DT_Z);
}

static int book_eabs(imdgpu_ctx *ctx, TtmState *t, const Lat &L, double tau, int rep)
{
  if (!t->laser_active) return 0;
  hipLaunchKernelGGL(k_ttm_eabs, dim3(1), dim3(64), 0, ctx->stream, L, t->source, t->vol * tau * (double) rep, t->d_eabs);
  HIPCHK(hipGetLastError());
  return 0;
}

// Eabs_local [eV]: absorbed laser energy booked by the diffusion sub-steps so far (synchronises)
extern "C" int imdgpu_ttm_get_eabs(imdgpu_ctx *ctx, double *eabs_local)
{
  NEED_TTM();
  HIPCHK(hipMemcpyAsync(eabs_local, t->d_eabs, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int imdgpu_ttm_diff(imdgpu_ctx *ctx, double tau)
{
  NEED_TTM();
  if (book_eabs(ctx, t, L, tau, 1)) return 1;
  ProfScope ps(ctx, PROF_TTM_DIFF);
  hipLaunchKernelGGL((k_ttm_diff<false, false>), dim3(diff_grid(t)), dim3(DT_X * DT_Y * DT_Z), 0, ctx->stream, L, tau, t->nsub, 1,
                     t->natoms, t->temp[t->cur], t->temp[t->cur ^ 1], t->U[t->cur], t->U[t->cur ^ 1], t->fd_k[t->kcur],
                     t->fd_k[t->kcur], t->Ce, t->fd_g, t->md_temp, t->source, t->dens, t->xi, t->Z, t->ne, t->cve, t->ee, 3, nullptr);
  HIPCHK(hipGetLastError());
  t->cur ^= 1;                                        // l1 <-> l2
  return 0;
}

extern "C" int imdgpu_ttm_cfl_maxdt(imdgpu_ctx *ctx, int steps, double *out)
{
  NEED_TTM();
  if (ctx->step_open) HIPCHK(hipStreamSynchronize(ctx->stream2));
  else if (imdgpu_ttm_persist_settle(ctx) < 0) return 1;
  hipLaunchKernelGGL(k_ttm_cfl, dim3(1), dim3(1024), 0, ctx->stream, L, t->natoms, t->fd_k[t->kcur], t->Ce, t->d_red);
  double mx = 0;
  HIPCHK(hipMemcpyAsync(&mx, t->d_red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ttm_check_mailbox(ctx)) return 1;
  t->max_dt_ttm = (steps > 0 ? mx : ctx->timestep / ((double) t->nsub)) * 0.9;
  if (out) *out = t->max_dt_ttm;
  return 0;
}

// One fused sub-step: do_DIFF + do_FILLMESH (own cell, new temperature) in one
// launch, then the ghost faces.  fd_k is double buffered because the stencil of
// the same launch still needs the neighbours' old kappa.
static int launch_substep_fused(imdgpu_ctx *ctx, TtmState *t, const Lat &L, double tau, int steps, int sub, const int *guard)
{
  const int flags = (sub == t->nsub - 1 ? 1 : 0) | (t->have_source ? 2 : 0);
  if (t->mb_host) {
    // decomposed lattice: pull the neighbours' publication `*gbase + sub` into the ghost cells, one sub-step on the
    // arrays (ghost cells read like any other), publish my new boundary layers as `*gbase + sub + 1`
    hipLaunchKernelGGL(k_ttm_pull_ghosts, dim3(6), dim3(64), 0, ctx->stream, L, t->mbd, sub, sub == 0 ? 1 : 0, t->natoms,
                       t->temp[t->cur], t->fd_k[t->kcur]);
    hipLaunchKernelGGL((k_ttm_diff<true, false>), dim3(diff_grid(t)), dim3(DT_X * DT_Y * DT_Z), 0, ctx->stream, L, tau, t->nsub, steps,
                       t->natoms, t->temp[t->cur], t->temp[t->cur ^ 1], t->U[t->cur], t->U[t->cur ^ 1], t->fd_k[t->kcur],
                       t->fd_k[t->kcur ^ 1], t->Ce, t->fd_g, t->md_temp, t->source, t->dens, t->xi, t->Z, t->ne, t->cve, t->ee, flags, guard);
    hipLaunchKernelGGL(k_ttm_push_faces, dim3(6), dim3(64), 0, ctx->stream, L, t->mbd, sub + 1, 0, t->natoms,
                       t->temp[t->cur ^ 1], t->fd_k[t->kcur ^ 1]);
  } else
  hipLaunchKernelGGL((k_ttm_diff<true, true>), dim3(diff_grid(t)), dim3(DT_X * DT_Y * DT_Z), 0, ctx->stream, L, tau, t->nsub, steps,
                     t->natoms, t->temp[t->cur], t->temp[t->cur ^ 1], t->U[t->cur], t->U[t->cur ^ 1], t->fd_k[t->kcur],
                     t->fd_k[t->kcur ^ 1], t->Ce, t->fd_g, t->md_temp, t->source, t->dens, t->xi, t->Z, t->ne, t->cve, t->ee, flags, guard);
  t->cur ^= 1; t->kcur ^= 1;
  return 0;
}

// The persistent sub-step kernel, when it applies: no host mailbox, no Maxwell solver, an even number of sub-steps, steps >= 1,
// and a lattice whose ACTIVE region fits PERSIST_MAX_TILES blocks (all of them must be resident at once; the kernel anchors its
// blocks on the bounding box of the active cells and decides per launch, on the device).  IMDGPU_TTM_PERSIST=0 switches it off.
//   trial   (1): the captured launch chain follows, each of its kernels guarded by the status word (0 = the persistent kernel
//                did the step), until two launches in a row came through;
//   trusted (2): the chain is dropped.  A launch that steps aside (status 1: census not complete in time because co-tenants
//                hold compute units, or an active region that has outgrown the launch) has touched nothing: the integrator is
//                guarded by the same status word, and the host issues that step's sub-steps through the launch chain and the
//                integrator again (imdgpu_ttm_persist_settle).  Three misses in a row switch the kernel off for the context.
//   A time-out INSIDE the loop (status 2) is an error at any time (CNT_TTM_TIMEOUT, reported once by the host).
// option "ttm_persist" (0 / 1; -1 = the default, IMDGPU_TTM_PERSIST or on) and the info words "ttm_persist_mode" (0 off,
// 1 trial, 2 trusted, -1 not decided yet) / "ttm_persist_runs" (launches that came through) / "ttm_persist_redone"
void imdgpu_ttm_set_persist(imdgpu_ctx *ctx, int value)
{
  ctx->ttm_persist_opt = value;
  if (ctx->ttm) {
    TtmState *t = ctx->ttm;
    (void) imdgpu_ttm_persist_settle(ctx);
    t->persist = value == 0 ? 0 : -1; t->persist_ok = 0; t->persist_miss_run = 0;
    (void) hipMemsetAsync(ctx->d_counters + CNT_TTM_TIMEOUT, 0, sizeof(int), ctx->stream);    // a reported time-out does not outlive the switch
  }
}
long long imdgpu_ttm_persist_info(imdgpu_ctx *ctx, int which)
{
  TtmState *t = ctx->ttm;
  if (!t) return -1;
  if (!ctx->step_open) (void) imdgpu_ttm_persist_settle(ctx);
  if (which == 2) return t->xrec_ready ? (t->xrec_fine ? 2 : 1) : 0;
  if (which == 3) return t->persist_redone;
  return which == 0 ? t->persist : t->persist_ok;
}

static int persist_mode(imdgpu_ctx *ctx, TtmState *t)
{
  if (t->persist < 0) {
    static const int envv = exp_env_int("IMDGPU_TTM_PERSIST", 1);
    const int env = ctx->ttm_persist_opt >= 0 ? ctx->ttm_persist_opt : envv;
    const int nt[3] = {(t->dim[0] - 2 + PT - 1) / PT, (t->dim[1] - 2 + PT - 1) / PT, (t->dim[2] - 2 + PT - 1) / PT};
    const long long full = (long long) nt[0] * nt[1] * nt[2];
    bool shape_ok = true;                                 // a lattice one layer thin stays on the launch chain (the kernel would step aside every time)
    for (int d = 0; d < 3; d++) if (t->dim[d] - 2 < 2) shape_ok = false;
    // the launch brings at most PERSIST_MAX_TILES blocks; a lattice with vacuum may need fewer blocks than it has (the
    // kernel decides, per launch); one that is four times the limit is not worth the attempts
    t->persist_tiles = (int) std::min<long long>(full, PERSIST_MAX_TILES);
    t->persist = (env && shape_ok && full <= 4LL * PERSIST_MAX_TILES) ? 1 : 0;
    if (t->persist && !t->d_pflags) {
      // census / gate words, the status words and the active cells' bounding box in one allocation
      if (dev_alloc(ctx, &t->d_pflags, 16 + 4) || dev_alloc(ctx, &t->d_prec, (size_t) 2 * t->ncell * 4) ||
          dev_alloc(ctx, &t->d_pbbox, (size_t) t->dim[0] + t->dim[1] + t->dim[2])) return -1;
      t->d_pstat = (int *) (t->d_pflags + 16);
      if (hipMemsetAsync(t->d_prec, 0, (size_t) 2 * t->ncell * 4 * sizeof(unsigned long long), ctx->stream) != hipSuccess) return -1;
      t->persist_base = 1;
      if (hipHostMalloc((void **) &t->h_pstat, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&t->ev_pstat, hipEventDisableTiming) != hipSuccess) { imdgpu_fail(ctx, "persistent TTM kernel: host allocations failed"); return -1; }
      if (hipMemsetAsync(t->d_pstat, 0, 4 * sizeof(int), ctx->stream) != hipSuccess) return -1;
      t->h_pstat[0] = 0;
    }
  }
  return t->persist;
}

// How did the last launch of the persistent sub-step kernel go?  Waits for its status word (the launch itself has long
// ended wherever this is called: after the step's synchronisation, or at the next call that touches the lattice).
// Returns 0 = nothing to do, 1 = it had stepped aside in trusted mode and the sub-steps of that step have now been issued
// through the launch chain on ctx->stream (a caller whose integrator was guarded by the status word launches it again),
// -1 = error.
int imdgpu_ttm_persist_settle(imdgpu_ctx *ctx)
{
  TtmState *t = ctx->ttm;
  if (!t || !t->pstat_pending) return 0;
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(t->ev_pstat) != hipSuccess) { imdgpu_fail(ctx, "persistent TTM kernel: waiting for its status failed"); return -1; }
  t->pstat_pending = 0;
  const int st = t->h_pstat[0], mode = t->pstat_mode;
  if (st == 0) { t->persist_miss_run = 0; if (++t->persist_ok >= 2 && t->persist == 1) t->persist = 2; return 0; }
  if (st != 1 || mode == 3) return 0;                    // a time-out inside the loop, or with other ranks waiting: CNT_TTM_TIMEOUT tells
  t->persist_ok = 0;
  if (++t->persist_miss_run >= 3 && t->persist > 0) t->persist = 0;   // co-tenants for good, or a lattice that outgrew the launch: the chain from now on
  if (mode != 2) return 0;                               // trial: the guarded chain behind the kernel has done the step
  const Lat L = make_lat(t);
  const double tau = ctx->timestep / ((double) t->nsub);
  if (hipMemsetAsync(t->d_pstat, 0, sizeof(int), ctx->stream) != hipSuccess) { imdgpu_fail(ctx, "persistent TTM kernel: re-issue failed"); return -1; }
  for (int i = 0; i < t->nsub; i++) if (launch_substep_fused(ctx, t, L, tau, 1, i, nullptr)) return -1;
  if (imdgpu_ttm_fill_ghost_layers(ctx)) return -1;
  t->persist_redone++;
  return 1;
}
// device word an integrator enqueued behind the lattice chain must look at (see k_move_ttm), or null
const int *imdgpu_ttm_persist_guard(imdgpu_ctx *ctx)
{
  TtmState *t = ctx->ttm;
  return (t && t->pstat_pending && t->pstat_mode == 2) ? t->d_pstat : nullptr;
}

// Does the lattice work that is forked beside the EAM force pass need LDS of its own?  The coefficient form of that pass
// (k_pass3_tile_coef) fills the CU's 160 KB, so kernels with LDS tiles would have to wait for its workgroups to end (measured in
// round 3 with the tiled launch chain on the 10 M-atom slab: 6.12 -> 6.94 ms per step).  Since round 5 the single-rank launch
// chain reads its neighbours in place (k_ttm_diff<true, true>: no LDS), like the persistent kernel and the Maxwell-solver loop;
// what still needs LDS is the 1D model's single workgroup and the mailbox form of the decomposed lattice.
int imdgpu_ttm_chain_needs_lds(const imdgpu_ctx *ctx)
{
  if (ctx->ttm1d) return 1;
  const TtmState *t = ctx->ttm;
  if (!t) return 0;
  if (t->fdtd && t->fdtd_on) return 0;
  if (ctx->ttm_decomposed && !t->xrec_ready) return 1;      // faces through the host mailbox: the tiled k_ttm_diff<true, false> per sub-step
  return 0;                                                 // the persistent kernel, and the single-rank launch chain (k_ttm_diff<true, true>: no LDS)
}

// the descriptor of the NEXT launch (its tag base is taken, not advanced: launch_persist does that)
static PersistDev persist_dev(imdgpu_ctx *ctx, TtmState *t, int mode, int steps, bool rem)
{
  PersistDev P;
  P.xrec = rem ? t->d_xrec : nullptr; P.xcells = t->xcells;
  for (int q = 0; q < 6; q++) { P.peer[q] = rem ? t->peer[q] : nullptr; P.xoff[q] = t->xoff[q]; }
  P.census = t->d_pflags; P.gate = t->d_pflags + 1; P.rec = t->d_prec; P.ncell = t->ncell;
  P.base = t->persist_base;
  P.status = t->d_pstat; P.counters = ctx->d_counters;
  P.trusted = mode == 2; P.bbox = t->d_pbbox;
  P.extra_census = (ctx->ttm_persist_miss >= 0 && ctx->ttm_persist_miss == steps) ? 1 : 0;   // test hook: a census that cannot complete
  return P;
}
static int launch_persist(imdgpu_ctx *ctx, TtmState *t, const Lat &L, double tau, int mode, int steps, bool rem = false)
{
  const PersistDev P = persist_dev(ctx, t, mode, steps, rem);
  t->persist_base += (unsigned) t->nsub + 1u;
  HIPCHK(hipMemsetAsync(t->d_pflags, 0, (16 + 4) * sizeof(unsigned), ctx->stream));     // d_pstat follows d_pflags
  double *T0 = t->temp[t->cur], *T1 = t->temp[t->cur ^ 1], *U0 = t->U[t->cur], *U1 = t->U[t->cur ^ 1];
  double *K0 = t->fd_k[t->kcur], *K1 = t->fd_k[t->kcur ^ 1];
  const bool alone = ctx->stream != ctx->stream2;         // not forked beside the force pass: a stand-alone calc_ttm
#define PERSIST_LAUNCH(S, W, R)                                                                                                  \
  hipLaunchKernelGGL((k_ttm_chain_persist<S, W, R>), dim3(t->persist_tiles), dim3(64), 0, ctx->stream, L, tau, t->nsub, t->natoms, T0, T1, \
                     U0, U1, K0, K1, t->Ce, t->fd_g, t->md_temp, t->source, t->dens, t->xi, t->Z, t->ne, t->cve, P)
  // register budget: 128 (no spills) alone on the chip; beside the coefficient form of the EAM force pass, whose three
  // 128-register waves leave 128 on every SIMD, 96 -- measured with the final force kernels: 128 ends the chain 0.27 ms before
  // the pass (1.33 / 1.59 ms), 96 costs the pass less and both end together (1.47 / 1.58; step 4.49 -> 4.48 ms), 80 makes the
  // chain the longer one again (1.62 / 1.52; 4.58); 80 beside the three-row form (138 registers).  IMDGPU_TTM_WPE=4|5|6 overrides.
  static const int wpe_env = exp_env_int("IMDGPU_TTM_WPE", 0);
  const int wpe = alone ? 4 : (wpe_env >= 4 && wpe_env <= 6 ? wpe_env : (imdgpu_p3_coef_active(ctx) ? 5 : 6));
#define PERSIST_W(S, R) { if (wpe == 4) PERSIST_LAUNCH(S, 4, R); else if (wpe == 5) PERSIST_LAUNCH(S, 5, R); else PERSIST_LAUNCH(S, 6, R); }
  if (rem) { if (t->have_source) PERSIST_W(true, true) else PERSIST_W(false, true) }
  else if (t->have_source) PERSIST_W(true, false) else PERSIST_W(false, false)
#undef PERSIST_W
#undef PERSIST_LAUNCH
  if (rem)                                                // ttm_fill_ghost_layers after the last sub-step, for the faces with a rank beyond them
    hipLaunchKernelGGL(k_ttm_xrec_final, dim3(nblk(t->xcells, 64)), dim3(64), 0, ctx->stream, L, P, t->nsub, t->natoms,
                       (t->nsub & 1) ? T1 : T0, (t->nsub & 1) ? K1 : K0);
  HIPCHK(hipGetLastError());
  // the status of every launch goes to the host; it is looked at after the step's synchronisation (imdgpu_ttm_persist_settle)
  HIPCHK(hipMemcpyAsync(t->h_pstat, t->d_pstat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipEventRecord(t->ev_pstat, ctx->stream));
  t->pstat_pending = 1; t->pstat_mode = rem ? 3 : mode;
  return 0;
}

// FD cells per direction of this rank's block (what imdgpu_ttm_local_block packs)
void imdgpu_ttm_block_dims(imdgpu_ctx *ctx, int ldim[3])
{
  for (int d = 0; d < 3; d++) ldim[d] = ctx->ttm ? ctx->ttm->ldim[d] : 0;
}

extern "C" int imdgpu_calc_ttm(imdgpu_ctx *ctx, int steps)
{
  if (ctx->external_halo && !ctx->ttm_decomposed) return imdgpu_fail(ctx, "calc_ttm: with several ranks use ttm_local_block / apply_blocks / ttm_lattice_step");
  if (imdgpu_ttm_update_fd(ctx, steps)) return 1;
  if (ctx->stream == ctx->stream2) {                  // forked by imdgpu_md_step: tell the main stream that update_fd is through
    HIPCHK(hipEventRecord(ctx->ev_ufd, ctx->stream));
    ctx->ufd_pending = 1;
  }
  if (ctx->ttm_adv && imdgpu_ttm_adv(ctx, 1.0, steps)) return 1;
  return imdgpu_ttm_lattice_step(ctx, steps);
}

// everything of calc_ttm after update_fd: activation, FILLMESH, ghosts, the diffusion sub-steps
extern "C" int imdgpu_ttm_lattice_step(imdgpu_ctx *ctx, int steps)
{
  NEED_TTM();
  const double tau = ctx->timestep / ((double) t->nsub);
  const bool MBX = t->mb_host != nullptr;
  const bool fdtd = t->fdtd && t->fdtd_on;
  if (imdgpu_ttm_persist_settle(ctx) < 0) return 1;       // a stand-alone caller's previous launch (no-op inside imdgpu_md_step)
  // the sub-step loop never returns to the host: nsub even => the ping-pong state after the loop equals the state before
  // it, so one persistent launch / one captured hipGraph serves every MD step; step 0 re-initialises U: run it un-captured
  const bool can_graph = (t->nsub % 2 == 0) && steps >= 1;
  // The persistent kernel does the whole loop whenever it applies: one rank's whole lattice, or this rank's block of a
  // decomposed lattice with remote records attached (faces included; the same decision on every rank: it depends on the
  // lattice shape, the sub-step count and the step number only).  Decided BEFORE FILLMESH, which leaves it the bounding box
  // of the active cells.
  int pmode = 0;
  if (can_graph && !fdtd && (ctx->ttm_decomposed ? t->xrec_ready : !MBX)) { pmode = persist_mode(ctx, t); if (pmode < 0) return 1; }
  const bool REMK = ctx->ttm_decomposed && pmode > 0;
  if (ctx->ttm_decomposed && !MBX && !REMK) return imdgpu_fail(ctx, "decomposed TTM lattice: imdgpu_ttm_mailbox_attach first (needed whenever the persistent kernel with remote records does not apply: step 0, odd sub-step counts)");
  imdgpu_tl_mark(ctx, "ttm_update_fd");
  static const bool force_rem = exp_env("IMDGPU_TTM_FORCE_REM") != nullptr;        // experiment: the remote-record build of the kernel on one rank's whole lattice
  if (t->laser_on && imdgpu_ttm_laser_rescale(ctx, &t->laser, steps)) return 1;   // update_fd has just zeroed the source
  if (ttm_cell_activation(ctx, t, L, REMK)) return 1;      // with remote records: also every face cell's atom count to the rank beyond
  if (pmode > 0) HIPCHK(hipMemsetAsync(t->d_pbbox, 0, ((size_t) t->dim[0] + t->dim[1] + t->dim[2]) * sizeof(int), ctx->stream));
  if (ttm_fillmesh(ctx, t, L, steps, pmode > 0 ? t->d_pbbox : nullptr)) return 1;
  if (REMK) {
    // nothing: the kernel publishes state 0 of its faces itself
  } else if (MBX) {                // first publication of this MD step: natoms, temp, fd_k of my boundary layers
    hipLaunchKernelGGL(k_ttm_push_faces, dim3(6), dim3(64), 0, ctx->stream, L, t->mbd, 0, 1, t->natoms,
                       t->temp[t->cur], t->fd_k[t->kcur]);
  } else if (imdgpu_ttm_fill_ghost_layers(ctx)) return 1;
  imdgpu_tl_mark(ctx, "ttm_fillmesh");
  if (fdtd) {                                                // config 5: the Maxwell solver drives the source inside the loop
    if (fdtd_lattice_loop(ctx, t, L, steps)) return 1;
    imdgpu_tl_mark(ctx, "ttm_diff");
    return 0;
  }
  if (book_eabs(ctx, t, L, tau, t->nsub)) return 1;          // one 64-thread workgroup, once per MD step
  {
    ProfScope ps(ctx, PROF_TTM_DIFF);
    const int gi = t->cur;
    if (REMK) { if (launch_persist(ctx, t, L, tau, 2, steps, true)) return 1; }     // no trial: other ranks wait for this one's records
    else if (pmode > 0 && launch_persist(ctx, t, L, tau, pmode, steps, force_rem)) return 1;
    const int *guard = (pmode == 1 && !REMK) ? t->d_pstat : nullptr;
    if (pmode == 2 || REMK) {
      // trusted: the persistent kernel alone
    } else if (can_graph && t->graph[gi] && t->graph_kcur[gi] == t->kcur && t->graph_tau[gi] == tau && t->graph_src[gi] == t->have_source &&
               t->graph_guard[gi] == guard) {
      HIPCHK(hipGraphLaunch(t->graph[gi], ctx->stream));
    } else if (can_graph) {
      if (t->graph[gi]) { (void) hipGraphExecDestroy(t->graph[gi]); t->graph[gi] = nullptr; }
      hipGraph_t g;
      const int k0 = t->kcur;
      HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      // the first few sub-steps run at raised wave priority: beside the EAM force pass a sub-step takes 22 us at normal
      // priority and 14 us raised (at 0.22 ms cost to the pass if all 100 were); a fifth of them raised lets the chain
      // end just before the pass does (IMDGPU_TTM_HIGH = how many)
      // (a lattice of 2x the cells and more -- the replicated lattice of several ranks -- outlasts the pass anyway: every
      // sub-step raised then; measured with 2x / 4x the cells: step 5.72 -> 5.53 / 6.66 -> 6.50 ms)
      static const int nhigh_env = exp_env_int("IMDGPU_TTM_HIGH", -1);
      const int nhigh = nhigh_env >= 0 ? nhigh_env : (t->ninner >= 90000 ? t->nsub : 20);
      for (int i = 0; i < t->nsub; i++) {
        Lat Li = L;
        if (i < nhigh) Li.wave_prio = 3;
        launch_substep_fused(ctx, t, Li, tau, 1, i, guard);
      }
      HIPCHK(hipStreamEndCapture(ctx->stream, &g));
      HIPCHK(hipGraphInstantiate(&t->graph[gi], g, nullptr, nullptr, 0));
      (void) hipGraphDestroy(g);
      t->graph_kcur[gi] = k0; t->graph_tau[gi] = tau; t->graph_src[gi] = t->have_source; t->graph_guard[gi] = guard;
      HIPCHK(hipGraphLaunch(t->graph[gi], ctx->stream));
    } else {
      for (int i = 0; i < t->nsub; i++) launch_substep_fused(ctx, t, L, tau, steps, i, nullptr);
    }
  }
  imdgpu_tl_mark(ctx, "ttm_substeps");
  if (REMK) {                      // ghost faces of other ranks: written by k_ttm_xrec_final; faces of periodic images / open faces: here
    int nf = (t->dim[1] - 2) * (t->dim[2] - 2) + (t->dim[2] - 2) * (t->dim[0] - 2) + (t->dim[0] - 2) * (t->dim[1] - 2);
    hipLaunchKernelGGL(k_ttm_ghosts, dim3(nblk(nf, TTM_WG)), dim3(TTM_WG), 0, ctx->stream, L, t->natoms, t->temp[t->cur], t->fd_k[t->kcur], t->xremote);
  } else if (MBX) {                // neighbours' final layers into the ghost cells; next MD step = next block of indices
    hipLaunchKernelGGL(k_ttm_pull_ghosts, dim3(6), dim3(64), 0, ctx->stream, L, t->mbd, t->nsub, 0, t->natoms,
                       t->temp[t->cur], t->fd_k[t->kcur]);
    hipLaunchKernelGGL(k_ttm_advance_base, dim3(1), dim3(1), 0, ctx->stream, t->d_gbase, t->nsub);
  } else if (imdgpu_ttm_fill_ghost_layers(ctx)) return 1;     // sub-steps resolve ghosts on the fly; leave the shell consistent
  HIPCHK(hipGetLastError());
  // CFL_maxdt: value is informational for the host (diff_substeps is fixed by fd_n_timesteps in the
  // sequence implemented here); computed on demand by imdgpu_ttm_cfl_maxdt to keep the step sync-free
  return 0;
}

// ---- decomposed lattice: ghost records in the neighbours' device memory ------------------------------------------
// Every rank owns one buffer: per parity one record of four granules for every cell of its six ghost faces, then one
// {tag | natoms} granule per face cell.  The ranks beyond its faces write into it (k_ttm_chain_persist<.., REM = true>).
static void xrec_layout(TtmState *t)
{
  int o = 0;
  for (int q = 0; q < 6; q++) {
    const int d = q >> 1;
    t->xoff[q] = o;
    o += (t->dim[(d + 1) % 3] - 2) * (t->dim[(d + 2) % 3] - 2);
  }
  t->xcells = o;
}
extern "C" int imdgpu_ttm_remote_init(imdgpu_ctx *ctx, int coarse, void **dev_ptr, size_t *bytes)
{
  NEED_TTM();
  if (!ctx->ttm_decomposed) return imdgpu_fail(ctx, "ttm_remote_init: set option ttm_decomposed before ttm_init");
  xrec_layout(t);
  const size_t n = (size_t) 2 * t->xcells * 4 + t->xcells + 8;          // + one spare granule per face (imdgpu_ttm_remote_ping)
  if (t->d_xrec) { (void) hipFree(t->d_xrec); t->d_xrec = nullptr; }
  t->xrec_ready = false;
  // fine-grained device memory: stores of other GPUs become visible to this GPU's polling loads while the kernels run
  // (coarse-grained memory is only coherent at kernel boundaries); `coarse` != 0 asks for plain hipMalloc -- enough when all
  // ranks share ONE GPU (rehearsals), where every access goes through the same L2
  if (coarse || hipExtMallocWithFlags((void **) &t->d_xrec, n * sizeof(unsigned long long), hipDeviceMallocFinegrained) != hipSuccess) {
    (void) hipGetLastError();
    HIPCHK(hipMalloc((void **) &t->d_xrec, n * sizeof(unsigned long long)));
    t->xrec_fine = false;
  } else t->xrec_fine = true;
  t->xrec_bytes = n * sizeof(unsigned long long);
  HIPCHK(hipMemset(t->d_xrec, 0, t->xrec_bytes));
  if (dev_ptr) *dev_ptr = t->d_xrec;
  if (bytes) *bytes = t->xrec_bytes;
  return 0;
}
int imdgpu_ttm_remote_is_fine(imdgpu_ctx *ctx) { return (ctx->ttm && ctx->ttm->d_xrec && ctx->ttm->xrec_fine) ? 1 : 0; }
// the 64-byte IPC handle of my buffer for ranks in OTHER processes (hipIpcMemHandle_t); *fine = 1: fine-grained memory
extern "C" int imdgpu_ttm_remote_ipc_handle(imdgpu_ctx *ctx, void *handle64, int *fine)
{
  NEED_TTM();
  if (!t->d_xrec) return imdgpu_fail(ctx, "ttm_remote_ipc_handle: imdgpu_ttm_remote_init first");
  hipIpcMemHandle_t h;
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t");
  if (hipIpcGetMemHandle(&h, t->d_xrec) != hipSuccess) { (void) hipGetLastError(); return imdgpu_fail(ctx, "hipIpcGetMemHandle failed for the ghost-record buffer (%s memory)", t->xrec_fine ? "fine-grained" : "coarse-grained"); }
  memcpy(handle64, &h, 64);
  if (fine) *fine = t->xrec_fine ? 1 : 0;
  return 0;
}
extern "C" int imdgpu_ttm_remote_open(imdgpu_ctx *ctx, const void *handle64, void **dev_ptr)
{
  HIPCHK(hipSetDevice(ctx->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  HIPCHK(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}
// peer[q]: the ghost-record buffer of the rank beyond face q (2d: +, 2d+1: -) as THIS process addresses it (its own
// pointer when that rank lives in this process, else what imdgpu_ttm_remote_open returned); NULL = no other rank there
// (open boundary, or a periodic direction with one rank: resolved locally)
extern "C" int imdgpu_ttm_remote_attach(imdgpu_ctx *ctx, void *const peer[6])
{
  NEED_TTM();
  if (!t->d_xrec) return imdgpu_fail(ctx, "ttm_remote_attach: imdgpu_ttm_remote_init first");
  t->xremote = 0;
  for (int q = 0; q < 6; q++) { t->peer[q] = (unsigned long long *) peer[q]; if (peer[q]) t->xremote |= 1 << q; }
  t->xrec_ready = true;
  t->persist = -1; t->persist_ok = 0;           // decide again (the shape test is the same, the kernel variant is not)
  return 0;
}

extern "C" int imdgpu_ttm_remote_detach(imdgpu_ctx *ctx)
{
  NEED_TTM();
  for (int q = 0; q < 6; q++) t->peer[q] = nullptr;
  t->xremote = 0; t->xrec_ready = false;
  t->persist = -1; t->persist_ok = 0;
  return 0;
}
// phase 0: store `word` (tagged with the face) into the spare granule of every neighbour's buffer, with the store the sub-step
// kernel uses; phase 1 (after all ranks met): read my own spare granules with its polling load; *bad = faces whose word is
// missing.  Run once at set-up: a peer mapping that cannot be written, or stores that never become visible, show up here and
// not as a time-out in the middle of a run.
__global__ void k_xrec_ping(int phase, unsigned word, unsigned mask, size_t spare, unsigned long long *mine, unsigned long long *p0,
                            unsigned long long *p1, unsigned long long *p2, unsigned long long *p3, unsigned long long *p4,
                            unsigned long long *p5, int *bad)
{
  unsigned long long *peer[6] = {p0, p1, p2, p3, p4, p5};
  const int q = threadIdx.x;
  if (q >= 6 || !(mask >> q & 1u)) return;
  if (phase == 0) __hip_atomic_store(peer[q] + spare + (q ^ 1), ((unsigned long long) word << 32) | (unsigned) (q ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if (xrec_load(mine + spare + q) != (((unsigned long long) word << 32) | (unsigned) q)) atomicOr(bad, 1 << q);
}
extern "C" int imdgpu_ttm_remote_ping(imdgpu_ctx *ctx, int phase, unsigned word, int *bad)
{
  NEED_TTM();
  if (!t->d_xrec || !t->xrec_ready) return imdgpu_fail(ctx, "ttm_remote_ping: imdgpu_ttm_remote_attach first");
  int *d_bad = nullptr;
  HIPCHK(hipMalloc((void **) &d_bad, sizeof(int)));
  HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  const size_t spare = (size_t) 2 * t->xcells * 4 + t->xcells;
  hipLaunchKernelGGL(k_xrec_ping, dim3(1), dim3(64), 0, ctx->stream, phase, word, (unsigned) t->xremote, spare, t->d_xrec, t->peer[0], t->peer[1],
                     t->peer[2], t->peer[3], t->peer[4], t->peer[5], d_bad);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  int h = 0;
  if (e == hipSuccess) e = hipMemcpy(&h, d_bad, sizeof(int), hipMemcpyDeviceToHost);
  (void) hipFree(d_bad);
  if (e != hipSuccess) return imdgpu_fail(ctx, "ttm_remote_ping: %s", hipGetErrorString(e));
  if (bad) *bad = h;
  return 0;
}

// ---- decomposed lattice: the shared-memory mailbox -----------------------------------------------------------
static void mailbox_layout(const TtmState *t, int *maxface, size_t *o_nat, size_t *o_dat, size_t *bytes)
{
  int mf = 0;
  for (int d = 0; d < 3; d++) mf = std::max(mf, (t->dim[(d + 1) % 3] - 2) * (t->dim[(d + 2) % 3] - 2));
  *maxface = mf;
  *o_nat = 128;
  *o_dat = 128 + (((size_t) 6 * mf * sizeof(int) + 63) / 64) * 64;
  *bytes = ((*o_dat + (size_t) 2 * 6 * mf * 2 * sizeof(double) + 4095) / 4096) * 4096;
}
extern "C" int imdgpu_ttm_mailbox_bytes(imdgpu_ctx *ctx, size_t *bytes_per_rank)
{
  NEED_TTM();
  if (!ctx->ttm_decomposed) return imdgpu_fail(ctx, "mailbox: set option ttm_decomposed before ttm_init");
  int mf; size_t a, b;
  mailbox_layout(t, &mf, &a, &b, bytes_per_rank);
  return 0;
}
// host_base: a region of nranks * bytes_per_rank bytes shared by all ranks of the node (e.g. a mmap'ed /dev/shm file),
// zero-filled before any rank attaches.  nbr[2*d+side]: rank of the neighbour block (may be this rank), -1 = open.
extern "C" int imdgpu_ttm_mailbox_attach(imdgpu_ctx *ctx, void *host_base, size_t bytes_per_rank, int rank, int nranks,
                                         const int nbr[6])
{
  NEED_TTM();
  if (!ctx->ttm_decomposed) return imdgpu_fail(ctx, "mailbox: set option ttm_decomposed before ttm_init");
  int mf; size_t o_nat, o_dat, need;
  mailbox_layout(t, &mf, &o_nat, &o_dat, &need);
  if (bytes_per_rank < need) return imdgpu_fail(ctx, "mailbox: %zu bytes per rank given, %zu needed", bytes_per_rank, need);
  // memory that is page-locked and mapped already (imdgpu_local_group_shared: ranks as threads of one process) is used as it
  // is; anything else (a mmap'ed /dev/shm file shared by processes) is registered here
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, host_base, 0) != hipSuccess || !dev) {
    (void) hipGetLastError();
    HIPCHK(hipHostRegister(host_base, bytes_per_rank * (size_t) nranks, hipHostRegisterMapped | hipHostRegisterPortable));
    t->mb_registered = true;
    HIPCHK(hipHostGetDevicePointer(&dev, host_base, 0));
  }
  t->mb_host = host_base; t->mb_bytes = bytes_per_rank; t->mb_rank = rank; t->mb_nranks = nranks;
  if (dev_alloc(ctx, &t->d_gbase, 1) || dev_alloc(ctx, &t->d_facecnt, 12)) return 1;
  HIPCHK(hipMemset(t->d_gbase, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemset(t->d_facecnt, 0, 12 * sizeof(int)));
  MailboxDev &M = t->mbd;
  M.base = (char *) dev; M.stride = bytes_per_rank; M.o_nat = o_nat; M.o_dat = o_dat;
  M.rank = rank; M.nranks = nranks; M.maxface = mf;
  M.gbase = t->d_gbase; M.counters = ctx->d_counters;
  for (int d = 0; d < 3; d++)
    for (int side = 0; side < 2; side++) {
      const int b = 2 * d + side;
      if (nbr[b] >= nranks) return imdgpu_fail(ctx, "mailbox: neighbour rank %d out of range", nbr[b]);
      M.nbr[b] = nbr[b];
    }
  for (int q = 0; q < 2; q++) if (t->graph[q]) { (void) hipGraphExecDestroy(t->graph[q]); t->graph[q] = nullptr; }
  return 0;
}
// calc_ttm of one rank's block: update_fd, activation, FILLMESH, face publication, the sub-steps
extern "C" int imdgpu_ttm_local_step(imdgpu_ctx *ctx, int steps)
{
  if (!ctx->ttm_decomposed) return imdgpu_fail(ctx, "ttm_local_step: decomposed lattice only");
  if (ctx->ttm_adv) return imdgpu_fail(ctx, "ttm_local_step with ttm_adv: call update_fd, the do_COMMFLUX exchange (imdgpu_ttm_commflux), ttm_adv and ttm_lattice_step separately");
  if (imdgpu_ttm_update_fd(ctx, steps)) return 1;
  return imdgpu_ttm_lattice_step(ctx, steps);
}

static int launch_move_ttm(imdgpu_ctx *ctx, double limit2)
{
  NEED_TTM();
  const int n = ctx->nlocal, nb = nblk(n, 256);
  ProfScope ps(ctx, PROF_INTEGRATE);
  hipLaunchKernelGGL(k_move_ttm, dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, L, imdgpu_make_geo(ctx),
                     ctx->pos, ctx->type, ctx->acell, ctx->p, ctx->f, ctx->mass, ctx->restr_identity ? nullptr : ctx->d_restr, ctx->ntypes, ctx->timestep,
                     ctx->ref, t->natoms, t->xi, t->vcom, t->ncell, ctx->partials, imdgpu_ttm_persist_guard(ctx));
  hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, limit2,
                     ctx->d_counters, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int imdgpu_move_atoms_ttm(imdgpu_ctx *ctx, double *ekin)
{
  if (imdgpu_ttm_persist_settle(ctx) < 0) return 1;      // a stand-alone calc_ttm's persistent launch that stepped aside: the chain, now
  if (launch_move_ttm(ctx, 1e300)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->h_sums, ctx->d_sums, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ekin) *ekin = ctx->h_sums[8];
  return 0;
}

int imdgpu_ttm_move_internal(imdgpu_ctx *ctx)
{
  double h = 0.5 * ctx->nbl_margin;
  return launch_move_ttm(ctx, h * h);
}

int imdgpu_ttm_step_internal(imdgpu_ctx *ctx, int steps)
{
  if (imdgpu_calc_ttm(ctx, steps)) return 1;
  double h = 0.5 * ctx->nbl_margin;
  return launch_move_ttm(ctx, h * h);
}


// ===========================================================================================================
// 1D Maxwell solver as laser source (BASELINE config 5): the reference's do_fdtd -- Yee scheme along x,
// z-polarised, Drude + Lorentz poles by auxiliary differential equations, split-field PML at both ends, soft source
// with total-field/scattered-field correction (imd_fdtd.c:63-126 fdtd_softsource, :128-306 do_fdtd, :594-763 init_fdtd,
// :765-859 init_pml; 1D branches) -- on a lattice of ONE row of FD cells (global_fd_dim.y = z = 1, which init_fdtd
// demands, :602-614).  calc_ttm's loop (imd_ttm_3D.c:96-122,163-179) runs per MD step
//     fdtd_substeps x [ do_fdtd(t_SI);  diff_substeps x (do_DIFF, ghosts, do_FILLMESH);  t_SI += tau_FDTD ]
// with fdtd_substeps ~ 440 for 7 A cells: far too many launches for a chain.  The row has at most ~1000 cells, so the
// whole loop of an MD step is ONE launch of ONE workgroup, a thread per cell: the fields and the cell's electron state
// stay in registers, neighbours are read through LDS, two barriers per FDTD update and one per diffusion sub-step.
// The Drude-Lorentz parameters DL[0..4] come from fitDL's tricubic tables in the reference (DL1..5.txt, not shipped):
// here they are constants from the host (fitDL's own commented alternative, imd_ttm_3D.c:2921-2937) or a per-cell array;
// DL[5], the plasma frequency, follows the cell's electron density like :2960.
// ===========================================================================================================
#define F_EPS0 8.841941282883075e-12      // globals.h:497-499: the reference's values
#define F_MU0 1.256637061435917e-06
#define F_C0 3.0e8

// ---- fitDL: the cell's Drude-Lorentz parameters from five tables over (rho, lg Te, lg Ti) ------------------------------
// [do_tricubinterp, imd_interpol.c:740-957; tricub_get_coeff(_stacked) :1088,:1150; tricub_eval :1108; fitDL,
// imd_ttm_3D.c:2919-2968].  The cell search assumes nodes at the cell centres; the differences at the 8 corners are
// un-normalised and clamped at the table's ends; a = A x with A the Lekien-Marsden matrix = the tensor product of the 1D
// cubic Hermite basis (HERM below: coefficients of f0, f1, f0', f1' in powers of t; pinned against the reference's matrix
// through the golden vectors).  tricub_eval passes x^i, y^j, z^k through INT variables, so every power is truncated
// towards zero: with all three scaled coordinates inside (-1, 1) only a[0] = f[i1][j1][k1] survives -- the common case,
// answered by one load; everything else (the upper end of an axis, tables with nodes elsewhere) takes the general path,
// which forms only the coefficients that carry weight, each as the reference's sum over the 64 corner values in order.
struct TricubDev {
  int is, js, ks, same;
  double xmin, xmax, ymin, ymax, zmin, zmax, dx, dy, dz;
  const double *x, *y, *z, *f;                   // f[(i*js + j)*ks + k]
};
struct TricubIdx { int nd[3][2], pm[3][2][2]; };  // [axis][corner bit] = i1 / i2; [axis][corner bit][0 min, 1 max]
__device__ const int HERM[4][4] = {{1, 0, 0, 0}, {0, 0, 1, 0}, {-3, 3, -2, -1}, {2, -2, 1, 1}};

// corner value / difference number q = 8 * type + corner of tricub_get_coeff's 64-vector (nested z( y( x(.) ) ))
__device__ __noinline__ double tricub_corner(const TricubDev &T, const TricubIdx &I, int q)
{
  const int ty = q >> 3, c = q & 7;
  const int der[3] = {(ty == 1 || ty == 4 || ty == 5 || ty == 7) ? 1 : 0, (ty == 2 || ty == 4 || ty == 6 || ty == 7) ? 1 : 0,
                      (ty == 3 || ty == 5 || ty == 6 || ty == 7) ? 1 : 0};
  const int b[3] = {c & 1, (c >> 1) & 1, (c >> 2) & 1};
  double vz[2] = {0.0, 0.0};
  for (int kz = 0; kz < (der[2] ? 2 : 1); kz++) {
    const int kk = der[2] ? I.pm[2][b[2]][1 - kz] : I.nd[2][b[2]];
    double vy[2] = {0.0, 0.0};
    for (int jy = 0; jy < (der[1] ? 2 : 1); jy++) {
      const int jj = der[1] ? I.pm[1][b[1]][1 - jy] : I.nd[1][b[1]];
      const size_t row = (size_t) jj * T.ks + kk, pl = (size_t) T.js * T.ks;
      vy[jy] = der[0] ? (T.f[I.pm[0][b[0]][1] * pl + row] - T.f[I.pm[0][b[0]][0] * pl + row]) : T.f[I.nd[0][b[0]] * pl + row];
    }
    vz[kz] = der[1] ? (vy[0] - vy[1]) : vy[0];
  }
  return der[2] ? (vz[0] - vz[1]) : vz[0];
}

__device__ __noinline__ double tricub_general(const TricubDev &T, const TricubIdx &I, const double *sc)
{
  int w[3][4];
  for (int ax = 0; ax < 3; ax++) {               // t4 = pow(x, i) etc. stored in int (tricub_eval :1134-1136)
    const double v = sc[ax];
    w[ax][0] = 1; w[ax][1] = (int) v; w[ax][2] = (int) (v * v); w[ax][3] = (int) (v * v * v);
  }
  double s4 = 0.0;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) {
    const int wt = w[0][i] * w[1][j] * w[2][k];
    if (wt == 0) continue;                       // a * 0.0 adds nothing (finite tables)
    double acc = 0.0;                            // a[i + 4j + 16k], tricub_get_coeff_stacked :1150-1164
    for (int q = 0; q < 64; q++) {
      const int tq = q >> 3, cq = q & 7;
      const int dxq = (tq == 1 || tq == 4 || tq == 5 || tq == 7), dyq = (tq == 2 || tq == 4 || tq == 6 || tq == 7),
                dzq = (tq == 3 || tq == 5 || tq == 6 || tq == 7);
      const int A = HERM[i][(cq & 1) + 2 * dxq] * HERM[j][((cq >> 1) & 1) + 2 * dyq] * HERM[k][((cq >> 2) & 1) + 2 * dzq];
      if (A != 0) acc += (double) A * tricub_corner(T, I, q);
    }
    s4 = s4 + acc * (double) wt;
  }
  return 0.0 + s4;
}

// cell search of do_tricubinterp (:752-808, :940-951): the node below per axis and the scaled coordinates
struct TricubCell { int lo[3], hi[3]; double sc[3]; bool inside, fast; };
__device__ __forceinline__ TricubCell tricub_locate(const TricubDev &T, double xp, double yp, double zp)
{
  TricubCell c;
  c.inside = !(xp > T.xmax || xp < T.xmin || yp > T.ymax || yp < T.ymin || zp < T.zmin || zp > T.zmax);      // :752-757 (no message)
  c.fast = true;
  if (!c.inside) return c;
  const int n[3] = {T.is, T.js, T.ks};
  const double p[3] = {xp, yp, zp};
  const double *nodes[3] = {T.x, T.y, T.z};
  c.lo[0] = (int) floor((xp - (T.xmin + T.dx / 2)) / (T.xmax - T.xmin) * (double) T.is);                        // :772-774
  c.lo[1] = (int) floor((yp - (T.ymin + T.dy / 2)) / (T.ymax - T.ymin) * (double) T.js);
  c.lo[2] = (int) floor((zp - (T.zmin + T.dz / 2)) / (T.zmax - T.zmin) * (double) T.ks);
  for (int ax = 0; ax < 3; ax++) {
    c.lo[ax] = max(c.lo[ax], 0);
    c.hi[ax] = max(min(c.lo[ax] + 1, n[ax] - 1), 0);
    const double x1 = nodes[ax][c.lo[ax]], x2 = nodes[ax][c.hi[ax]];
    // |(p - x1) / (x2 - x1)| < 1 exactly when |p - x1| < |x2 - x1| (a correctly rounded quotient of doubles cannot reach 1
    // from below): the common case needs no division; the quotient itself only on the general path
    const bool in1 = (x2 - x1 != 0) && fabs(p[ax] - x1) < fabs(x2 - x1);
    c.sc[ax] = in1 ? 0.0 : ((x2 - x1 == 0) ? 1.0 : (p[ax] - x1) / (x2 - x1));                                      // :940-951
    c.fast = c.fast && in1;
  }
  return c;
}

__device__ __forceinline__ double tricub_value(const TricubDev &T, const TricubCell &c)
{
  if (!c.inside) return -1;
  if (c.fast) return 0.0 + T.f[((size_t) c.lo[0] * T.js + c.lo[1]) * T.ks + c.lo[2]];      // a[0] alone: 1 * f[i1][j1][k1]
  const int n[3] = {T.is, T.js, T.ks};
  TricubIdx I;
  for (int ax = 0; ax < 3; ax++) {
    I.nd[ax][0] = c.lo[ax]; I.nd[ax][1] = c.hi[ax];
    for (int b = 0; b < 2; b++) { I.pm[ax][b][0] = max(I.nd[ax][b] - 1, 0); I.pm[ax][b][1] = min(I.nd[ax][b] + 1, n[ax] - 1); }   // :795-808
  }
  return tricub_general(T, I, c.sc);
}

__device__ __forceinline__ double tricub_lookup(const TricubDev &T, double xp, double yp, double zp)
{
  return tricub_value(T, tricub_locate(T, xp, yp, zp));
}

#define F_HBAREV 6.582119487171653e-16      // globals.h:485
// fitDL for one cell: DL[0..4] from the tables, DL[5] the plasma frequency of the cell's electron density (:2939-2960).
// tab[0].same: the five tables share dimensions, ranges and nodes (one grid, five quantities): one cell search for all
__device__ __forceinline__ void fit_dl(const TricubDev *tab, double temp, double md_temp, double dens, double ne, double *DL)
{
  const double Te = fmax(temp, 0.0259), Ti = fmax(md_temp, 0.0259);
  double lgTe = log10(Te), lgTi = log10(Ti);
  lgTi = fmax(lgTi, -1.587504); lgTe = fmax(lgTe, -1.587504);
  lgTe = fmin(lgTe, tab[0].ymax); lgTi = fmin(lgTi, tab[0].zmax);
  if (tab[0].same) {
    const TricubCell c = tricub_locate(tab[0], dens, lgTe, lgTi);
    DL[0] = tricub_value(tab[0], c);
    for (int q = 1; q < 5; q++) DL[q] = tricub_value(tab[q], c) / F_HBAREV;
  } else {
    DL[0] = tricub_lookup(tab[0], dens, lgTe, lgTi);
    for (int q = 1; q < 5; q++) DL[q] = tricub_lookup(tab[q], dens, lgTe, lgTi) / F_HBAREV;
  }
  DL[5] = sqrt(ne * C_ECHARGE * C_ECHARGE / C_ECONST / C_EMASS);
}

struct FdtdDev {
  int nx, fd_min_atoms, srcx_int, containssrc, dl_mode;     // dl_mode 0: constants + plasma frequency from ne; 1: per-cell array as is; 2: fitDL's tables
  double dt, dx, Nlambdax, Imp0, E0, t0, sigma_t_squared, DLc[5];
  double *E, *H, *J, *Jl, *P, *sig, *Jo, *Jlo, *DL;
  const TricubDev *tab;                                     // five tables in device memory (dl_mode 2)
};
struct FdtdState {
  FdtdDev d{};
  double E0_init = 0, omega_laser = 0, t_SI = 0;
  int bw = 0, on = 0;
  TricubDev *d_tab = nullptr;
  TricubDev h_tab[5] = {};
  std::vector<void *> tab_bufs;
};

static void imdgpu_fdtd_free(TtmState *t)
{
  if (!t->fdtd) return;
  void *ptrs[] = {t->fdtd->d.E, t->fdtd->d.H, t->fdtd->d.J, t->fdtd->d.Jl, t->fdtd->d.P, t->fdtd->d.sig, t->fdtd->d.Jo, t->fdtd->d.Jlo, t->fdtd->d.DL};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  for (void *p : t->fdtd->tab_bufs) (void) hipFree(p);
  if (t->fdtd->d_tab) (void) hipFree(t->fdtd->d_tab);
  delete t->fdtd;
  t->fdtd = nullptr; t->fdtd_on = 0;
}

// one do_fdtd(t) for the cell of this thread; all threads of the block call it (two barriers inside).  sH / sE: LDS rows.
struct FdtdCell { double E, H, J, Jl, P, Jo, Jlo, src, sig, DL[6]; int active; };
// K: what of the update depends on the cell's conductivity and on DL[0..4] only -- twelve divisions -- taken once per launch where
// those parameters do not change inside an MD step (dl_mode 0 and 1; with fitDL's tables they follow the electron temperature:
// K = nullptr, everything per sub-step as the reference writes it, imd_fdtd.c:178-286)
struct FdtdCoef { double hy1, hy2, ezx1, ezx2, Cb, CD1, CL1, CD2, CL2, wL2e, w02, qeL, gpe, inv_dx; };
__device__ __forceinline__ FdtdCoef fdtd_coef(const FdtdDev &F, const FdtdCell &c)
{
  FdtdCoef K;
  const double dt = F.dt, dx = F.dx;
  const double sigmastary = c.sig * F_MU0 / F_EPS0;
  K.hy1 = (F_MU0 - 0.5 * dt * sigmastary) / (F_MU0 + 0.5 * dt * sigmastary);
  K.hy2 = (dt / dx) / (F_MU0 + 0.5 * dt * sigmastary);
  K.ezx1 = (F_EPS0 - 0.5 * dt * c.sig) / (F_EPS0 + 0.5 * dt * c.sig);
  K.ezx2 = (dt / dx) / (F_EPS0 + 0.5 * dt * c.sig);
  K.Cb = dt / F_EPS0 / c.DL[0];
  const double gamma_p = c.DL[1], omegapl_L = c.DL[2], Omega0_L = c.DL[3], Gamma_L = c.DL[4];
  K.CD1 = (2 - dt * gamma_p) / (2 + dt * gamma_p); K.CL1 = (2 - dt * Gamma_L) / (2 + dt * Gamma_L);
  K.CD2 = 2 * dt / (2 + dt * gamma_p); K.CL2 = 2 * dt / (2 + dt * Gamma_L);
  const double wL2 = omegapl_L * omegapl_L;
  K.wL2e = wL2 * F_EPS0; K.w02 = Omega0_L * Omega0_L;
  K.qeL = Gamma_L / F_EPS0 / wL2; K.gpe = gamma_p / F_EPS0;
  K.inv_dx = 1.0 / dx;
  return K;
}
// the two source terms of a solver step at time t: what the soft source adds to E at the source cell and what the TF/SF
// correction takes from H in the cell in front of it (fdtd_softsource, imd_fdtd.c:78-83,114-122; :293-303)
__device__ __forceinline__ double fdtd_hinc(const FdtdDev &F, double t, double laser_timefun)
{
  const double wt = 2.0 * C_PI * (F_C0 / (F.dx * F.Nlambdax));
  return laser_timefun * sin(wt * (t + F.dt)) / F.Imp0;
}
__device__ __forceinline__ double fdtd_einc(const FdtdDev &F, double t)
{
  const double wt = 2.0 * C_PI * (F_C0 / (F.dx * F.Nlambdax));
  const double d = t - F.t0;
  return sin(wt * t) * exp(-0.5 * (d * d) / F.sigma_t_squared) * F.E0;
}
// pre (may be null): {Hinc, Einc} of this step, evaluated ahead for all steps of the launch by as many threads (k_ttm_fdtd_chain):
// a sine, and a sine times an exponential, on the critical path of every one of the ~430 steps of an MD step otherwise
__device__ __forceinline__ void fdtd_step(const FdtdDev &F, FdtdCell &c, int i, bool have, bool inner, double t, double laser_timefun,
                                          double *sH, double *sE, const FdtdCoef *K = nullptr, const double *pre = nullptr)
{
  const double dt = F.dt, dx = F.dx;
  const int ilocal = F.srcx_int + 1;
  if (F.containssrc == 1 && i == ilocal) {                                   // fdtd_softsource, :78-83,114-122 (ilocal < nx)
    const double ezy2 = (dt / dx) / (F_EPS0);
    const double Hinc = pre ? pre[0] : fdtd_hinc(F, t, laser_timefun);
    c.E += ezy2 * Hinc;
  }
  if (have) sH[i] = c.H;                     // every thread of the block passes the two barriers (threads past the row idle)
  __syncthreads();
  if (inner) {                                                               // E: n -> n+1, :178-231
    const double Hm = sH[i - 1];
    if (c.active) {
      c.Jo = c.J; c.Jlo = c.Jl;
      if (K) c.E = c.E + K->Cb * (c.H - Hm) * K->inv_dx - K->Cb * c.J - K->Cb * c.Jl;
      else {
        const double Cb = dt / F_EPS0 / c.DL[0];
        c.E = c.E + Cb * (c.H - Hm) / dx - Cb * c.J - Cb * c.Jl;
      }
      c.P = c.P + dt * c.Jl;
    } else if (K) c.E = K->ezx1 * c.E + K->ezx2 * (c.H - Hm);
    else {
      const double ezx1 = (F_EPS0 - 0.5 * dt * c.sig) / (F_EPS0 + 0.5 * dt * c.sig);
      const double ezx2 = (dt / dx) / (F_EPS0 + 0.5 * dt * c.sig);
      c.E = ezx1 * c.E + ezx2 * (c.H - Hm);
    }
  }
  if (have) sE[i] = c.E;
  __syncthreads();
  if (inner && K) {                                                          // H: -> n+3/2, currents, power, :236-286, coefficients from K
    c.H = K->hy1 * c.H + K->hy2 * (sE[i + 1] - c.E);
    if (c.active) {
      const double wD2 = c.DL[5] * c.DL[5];
      c.J = K->CD1 * c.J + K->CD2 * (wD2 * F_EPS0 * c.E);
      c.Jl = K->CL1 * c.Jl + K->CL2 * (K->wL2e * c.E - K->w02 * c.P);
      const double Jz = (c.J + c.Jo) * 0.5, Jlz = 0.5 * (c.Jl + c.Jlo);
      const double qe = K->gpe * (Jz * Jz) / wD2 + K->qeL * (Jlz * Jlz);
      c.src = 6.3538562638e-26 * qe;                                          // W/m^3 -> IMD units
    }
  } else if (inner) {
    const double sigmastary = c.sig * F_MU0 / F_EPS0;
    const double hy1 = (F_MU0 - 0.5 * dt * sigmastary) / (F_MU0 + 0.5 * dt * sigmastary);
    const double hy2 = (dt / dx) / (F_MU0 + 0.5 * dt * sigmastary);
    c.H = hy1 * c.H + hy2 * (sE[i + 1] - c.E);
    if (c.active) {
      const double gamma_p = c.DL[1], omegapl_L = c.DL[2], Omega0_L = c.DL[3], Gamma_L = c.DL[4], omega_plasma = c.DL[5];
      const double CD1 = (2 - dt * gamma_p) / (2 + dt * gamma_p), CL1 = (2 - dt * Gamma_L) / (2 + dt * Gamma_L);
      const double CD2 = 2 * dt / (2 + dt * gamma_p), CL2 = 2 * dt / (2 + dt * Gamma_L);
      const double wD2 = omega_plasma * omega_plasma, wL2 = omegapl_L * omegapl_L, w02 = Omega0_L * Omega0_L;
      c.J = CD1 * c.J + CD2 * (wD2 * F_EPS0 * c.E);
      c.Jl = CL1 * c.Jl + CL2 * (wL2 * F_EPS0 * c.E - w02 * c.P);
      const double Jz = (c.J + c.Jo) * 0.5, Jlz = 0.5 * (c.Jl + c.Jlo);
      const double qe = gamma_p * (Jz * Jz) / F_EPS0 / wD2 + Gamma_L / F_EPS0 / wL2 * (Jlz * Jlz);
      c.src = 6.3538562638e-26 * qe;                                          // W/m^3 -> IMD units
    }
  }
  if (F.containssrc && i == ilocal - 1) {                                    // TF/SF: no wave towards -x, :293-303
    const double Einc = pre ? pre[1] : fdtd_einc(F, t);
    c.H -= (dt / dx) / (F_MU0) * Einc;
  }
}

// nsub x [do_fdtd; ndiff x (do_DIFF, do_FILLMESH)] for the whole row, one workgroup.  WITH_TTM = false: the solver alone.
// MAXT: 256 for rows of up to 254 cells (the register file then holds the cell's whole state: no spills), 1024 beyond
template <bool WITH_TTM, int MAXT>
__global__ __launch_bounds__(MAXT) void k_ttm_fdtd_chain(Lat L, FdtdDev F, int nsub, int ndiff, double t_SI, double tau_fdtd,
    double tau_diff, int laser_active, double laser_timefun, int steps, const int *natoms, double *T1, double *T2, double *U1,
    double *U2, double *fd_k, double *Ce, double *fd_g, const double *md_temp, double *source, const double *dens, double *xi,
    double *Zo, double *neo, EosDev cve, EosDev ee, double *eabs, int npre)
{
  extern __shared__ double fl[];
  const int nx = F.nx, i = threadIdx.x;
  double *sH = fl, *sE = fl + nx, *sT = fl + 2 * nx, *sK = fl + 3 * nx;
  int *sN = (int *) (fl + 4 * nx);
  const bool have = i < nx, inner = i >= 1 && i < nx - 1;
  const int cell = have ? fidx(L, i, 1, 1) : 0, m = L.fd_min_atoms;
  FdtdCell c{};
  int nat = 0;
  double Te = 0, Tprev = 0, U = 0, Uprev = 0, xiv = 0, ce = 0, kk = 0, g = 0, Tl = 0, rho = 0, Zv = 0, nev = 0, eacc = 0;
  if (have) {
    c.E = F.E[i]; c.H = F.H[i]; c.J = F.J[i]; c.Jl = F.Jl[i]; c.P = F.P[i]; c.Jo = F.Jo[i]; c.Jlo = F.Jlo[i]; c.sig = F.sig[i];
    nat = inner ? natoms[cell] : 0;
    c.active = nat >= m;
    c.src = WITH_TTM ? source[cell] : 0.0;
    if (F.dl_mode == 1) for (int q = 0; q < 6; q++) c.DL[q] = F.DL[6 * i + q];
    else { for (int q = 0; q < 5; q++) c.DL[q] = F.DLc[q]; c.DL[5] = 0.0; }
    if (WITH_TTM && inner) {
      Te = T1[cell]; Tprev = T2[cell]; U = U1[cell]; Uprev = U2[cell]; xiv = xi[cell]; ce = Ce[cell]; kk = fd_k[cell]; g = fd_g[cell];
      Tl = md_temp[cell]; rho = dens[cell]; Zv = Zo[cell]; nev = neo[cell];
      if (F.dl_mode == 0 && c.active) c.DL[5] = sqrt(nev * C_ECHARGE * C_ECHARGE / C_ECONST / C_EMASS);      // fitDL, :2960
      if (F.dl_mode == 2) {                       // the state the FILLMESH before the loop left (imd_ttm_3D.c:600)
        for (int q = 0; q < 6; q++) c.DL[q] = F.DL[6 * i + q];
        if (c.active) fit_dl(F.tab, Te, Tl, rho, nev, c.DL);
      }
    }
    sN[i] = nat;
  }
  __syncthreads();
  // diffusion neighbours along x: ghost cells of a periodic row are the opposite inner cells, of an open row empty
  const int ilo = i == 1 ? (L.pbc[0] == 1 ? nx - 2 : -1) : i - 1, ihi = i == nx - 2 ? (L.pbc[0] == 1 ? 1 : -1) : i + 1;
  const bool lo_ok = inner && ilo >= 0 && sN[ilo] >= m, hi_ok = inner && ihi >= 0 && sN[ihi] >= m;
  double t = t_SI;
  // constant inside an MD step: the solver's coefficients unless fitDL's tables move DL[0..4] with the electron temperature,
  // the cell's density and lattice temperature (mesh_pre), two reciprocals of the diffusion update
  const bool hoist = F.dl_mode != 2;
  FdtdCoef K{};
  if (hoist && have) K = fdtd_coef(F, c);
  const MeshPre mpre = mesh_pre(L, (WITH_TTM && inner && c.active) ? Tl : 1.0, (WITH_TTM && inner && c.active) ? rho : 1.0);
  const double inv_nat = nat > 0 ? 1.0 / ((double) nat) : 0.0, inv_tlrho = (WITH_TTM && inner && c.active) ? 1.0 / (Tl * rho) : 0.0;
  // the source terms of all nsub steps ahead: the times as the loop accumulates them (one thread, nsub additions), then one step
  // per thread.  Room for them in the LDS behind the row's arrays when the launch brought it (npre >= nsub).
  double *sPre = (double *) (sN + nx + (nx & 1));
  const bool pre = laser_active && F.containssrc && npre >= nsub;
  if (pre) {
    if (i == 0) { double tq = t_SI; for (int q = 0; q < nsub; q++) { sPre[2 * q] = tq; tq += tau_fdtd; } }
    __syncthreads();
    for (int q = i; q < nsub; q += blockDim.x) {
      const double tq = sPre[2 * q];
      sPre[2 * q] = fdtd_hinc(F, tq, laser_timefun); sPre[2 * q + 1] = fdtd_einc(F, tq);
    }
    __syncthreads();
  }
  for (int s = 0; s < nsub; s++) {
    if (laser_active) fdtd_step(F, c, i, have, inner, t, laser_timefun, sH, sE, hoist ? &K : nullptr, pre ? sPre + 2 * s : nullptr);       // uniform: laser_active is a kernel argument
    if (WITH_TTM) {
      for (int q = 0; q < ndiff; q++) {
        if (have) { sT[i] = Te; sK[i] = kk; }
        __syncthreads();
        if (inner) {
          if (laser_active) eacc += c.src * L.vol * tau_diff;                          // do_DIFF, :1647-1651
          if (c.active) {
            const double xaT = hi_ok ? sT[ihi] : Te, xaK = hi_ok ? sK[ihi] : kk, xbT = lo_ok ? sT[ilo] : Te, xbK = lo_ok ? sK[ilo] : kk;
            // the y and z faces of a one-row lattice see the cell itself (periodic) or nothing (open): zero flux either way
            const double Tn = tau_diff / ce * ((((kk + xaK) / 2 * (xaT - Te) * L.invhsq[0]) - ((kk + xbK) / 2 * (Te - xbT) * L.invhsq[0]))
                                               - g * (Te - Tl) + c.src) + Te;
            Tprev = Te; Uprev = U;
            U = U + (Tn - Te) * ce * L.vol * inv_nat;
            xiv += (Tn - Tl) * L.xi_fac * g * inv_tlrho;
            Te = Tn;
            MeshOut o = mesh_eval(L, cve, mpre, Te);
            Zv = o.Z; nev = o.ne; ce = o.Ce; kk = o.k; g = o.g;
            if (steps < 1) { U = eos_ee(ee, rho, Te * 11604.5) * 26.9815 * C_AMU * C_J2EV; Uprev = U; }
            if (F.dl_mode == 0) c.DL[5] = sqrt(nev * C_ECHARGE * C_ECHARGE / C_ECONST / C_EMASS);
            else if (F.dl_mode == 2) fit_dl(F.tab, Te, Tl, rho, nev, c.DL);
          }
        }
        __syncthreads();
      }
    }
    t += tau_fdtd;
  }
  if (have) {
    F.E[i] = c.E; F.H[i] = c.H; F.J[i] = c.J; F.Jl[i] = c.Jl; F.P[i] = c.P; F.Jo[i] = c.Jo; F.Jlo[i] = c.Jlo;
    if (F.dl_mode != 1 && F.DL) for (int q = 0; q < 6; q++) F.DL[6 * i + q] = c.DL[q];
    if (WITH_TTM && inner) {
      source[cell] = c.src;
      if (c.active) { T1[cell] = Te; T2[cell] = Tprev; U1[cell] = U; U2[cell] = Uprev; xi[cell] = xiv; Ce[cell] = ce; fd_k[cell] = kk;
                      fd_g[cell] = g; Zo[cell] = Zv; neo[cell] = nev; }
    } else if (!WITH_TTM && inner) source[i] = c.src;
  }
  if (WITH_TTM && eabs) {                       // Eabs_local of this MD step, summed in cell order by one thread (deterministic)
    __syncthreads();
    sT[have ? i : 0] = have ? eacc : 0.0;
    __syncthreads();
    if (i == 0) { double e = 0; for (int q = 1; q < nx - 1; q++) e += sT[q]; eabs[0] += e; }
  }
}

extern "C" int imdgpu_ttm_fdtd_init(imdgpu_ctx *ctx, const imdgpu_fdtd_params *p)
{
  NEED_TTM();
  if (!p) return imdgpu_fail(ctx, "ttm_fdtd_init: no parameters");
  if (t->dim[1] != 3 || t->dim[2] != 3) return imdgpu_fail(ctx, "ttm_fdtd_init: the 1D Maxwell solver needs a lattice of one row of FD cells (fd_ext covering all MD cells in y and z), like init_fdtd (imd_fdtd.c:602-614)");
  if (t->pbc[0] == 1) return imdgpu_fail(ctx, "ttm_fdtd_init: x must be open (PML at both ends)");
  if (ctx->external_halo) return imdgpu_fail(ctx, "ttm_fdtd_init: one rank only");
  if (t->dim[0] > 1024) return imdgpu_fail(ctx, "ttm_fdtd_init: at most 1022 cells along x");
  imdgpu_fdtd_free(t);
  FdtdState *S = new FdtdState();
  t->fdtd = S;
  FdtdDev &F = S->d;
  const int nx = t->dim[0];
  F.nx = nx; F.fd_min_atoms = t->fd_min_atoms; F.dl_mode = 0;
  // init_fdtd, :594-652
  F.Imp0 = sqrt(F_MU0 / F_EPS0);
  F.dx = t->h[0] / 1e10;
  F.dt = std::min(1.0, p->Sc) * F.dx / F_C0;
  const double freq = F_C0 / p->lambda;
  S->omega_laser = 2.0 * M_PI * freq;
  F.Nlambdax = F_C0 / (freq * F.dx);
  F.E0 = sqrt(2.0 * p->I0 * F.Imp0);
  S->E0_init = F.E0;
  F.t0 = p->t_0; F.sigma_t_squared = p->sigma_t * p->sigma_t;
  F.srcx_int = (int) round(p->srcx / F.dx);
  if (F.srcx_int < p->bw + 10) F.srcx_int = p->bw + 10;
  F.containssrc = (F.srcx_int <= nx - 2 && F.srcx_int >= 0) ? 1 : 0;
  S->bw = std::max(p->bw, 5);
  for (int q = 0; q < 5; q++) F.DLc[q] = p->DL[q];
  double **arr[] = {&F.E, &F.H, &F.J, &F.Jl, &F.P, &F.sig, &F.Jo, &F.Jlo};
  for (double **a : arr) { HIPCHK(hipMalloc((void **) a, nx * sizeof(double))); HIPCHK(hipMemset(*a, 0, nx * sizeof(double))); }
  HIPCHK(hipMalloc((void **) &F.DL, 6 * (size_t) nx * sizeof(double)));
  HIPCHK(hipMemset(F.DL, 0, 6 * (size_t) nx * sizeof(double)));
  // init_pml, 1D branch :765-858 (evaluated on the host with the reference's expressions)
  {
    const double refl_coeff = 1.0e-12, go = 8.0, bw = (double) S->bw;
    const double sigma_max = (-log10(refl_coeff) * (go + 1.0) * F_EPS0 * F_C0) / (2.0 * bw * F.dx);
    const double bf3 = (1.0 * sigma_max) / (pow(bw, go) * (go + 1.0));
    std::vector<double> sig(nx, 0.0);
    const int gx = nx - 2;
    for (int i = 1; i < nx - 1; i++) {
      const int ig = i - 1;
      if (ig >= gx - S->bw) {
        const double idist = (double) (S->bw - ((gx - 1) - ig));
        sig[i] = bf3 * (pow(idist + 0.5, go + 1.0) - pow(idist - 0.5 * (ig > gx - 1 - S->bw), go + 1.0));
      } else if (ig <= S->bw) {
        const double idist = (double) (S->bw - ig);
        sig[i] = bf3 * (pow(idist + 0.5, go + 1.0) - pow(idist - 0.5 * (ig < S->bw), go + 1.0));
      }
    }
    HIPCHK(hipMemcpy(F.sig, sig.data(), nx * sizeof(double), hipMemcpyHostToDevice));
  }
  S->t_SI = 0.0; S->on = 1; t->fdtd_on = 1;
  ctx->reserve_cus = 1;                            // the solver loop is one multi-wave workgroup: it gets a CU of its own beside the force passes
  t->have_source = 1;
  t->omega_laser = S->omega_laser;                 // init_fdtd sets the global the wide-range model reads (:633)
  return 0;
}

// per-cell Drude-Lorentz parameters (6 per cell of the row, ghost cells included) instead of the constants
extern "C" int imdgpu_ttm_fdtd_set_dl(imdgpu_ctx *ctx, const double *DL)
{
  NEED_TTM();
  if (!t->fdtd) return imdgpu_fail(ctx, "ttm_fdtd_set_dl: imdgpu_ttm_fdtd_init first");
  t->fdtd->d.dl_mode = DL ? 1 : (t->fdtd->d.tab ? 2 : 0);
  if (DL) HIPCHK(hipMemcpy(t->fdtd->d.DL, DL, 6 * (size_t) t->fdtd->d.nx * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int imdgpu_ttm_fdtd_get(imdgpu_ctx *ctx, double *dt_FDTD, double *E0, int *srcx_int, double *t_SI)
{
  NEED_TTM();
  if (!t->fdtd) return imdgpu_fail(ctx, "ttm_fdtd_get: imdgpu_ttm_fdtd_init first");
  if (dt_FDTD) *dt_FDTD = t->fdtd->d.dt;
  if (E0) *E0 = t->fdtd->d.E0;
  if (srcx_int) *srcx_int = t->fdtd->d.srcx_int;
  if (t_SI) *t_SI = t->fdtd->t_SI;
  return 0;
}
extern "C" int imdgpu_ttm_fdtd_set_time(imdgpu_ctx *ctx, double t_SI)
{
  NEED_TTM();
  if (!t->fdtd) return imdgpu_fail(ctx, "ttm_fdtd_set_time: imdgpu_ttm_fdtd_init first");
  t->fdtd->t_SI = t_SI;
  return 0;
}

// the fields of the row (nx values each, ghost cells included; NULL = skip).  which: 0 Ezx, 1 Hy, 2 Jzx, 3 Jlzx, 4 Pzx,
// 5 sigmay, 6 Jzx of the step before, 7 Jlzx of the step before
extern "C" int imdgpu_ttm_fdtd_fields(imdgpu_ctx *ctx, int which, double *out)
{
  NEED_TTM();
  if (!t->fdtd) return imdgpu_fail(ctx, "ttm_fdtd_fields: imdgpu_ttm_fdtd_init first");
  const FdtdDev &F = t->fdtd->d;
  double *src[] = {F.E, F.H, F.J, F.Jl, F.P, F.sig, F.Jo, F.Jlo, F.DL};
  if (which < 0 || which > 8) return imdgpu_fail(ctx, "ttm_fdtd_fields: which = 0..8");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, src[which], (which == 8 ? 6 : 1) * (size_t) F.nx * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// fitDL's five tables (DL1.txt .. DL5.txt as read_tricub_interp keeps them, imd_ttm_3D.c:1245-1249): from now on every
// FILLMESH of the solver loop looks the cell's DL[0..4] up at (dens, lg Te, lg Ti).  tabs = NULL: back to the constants.
extern "C" int imdgpu_ttm_fdtd_set_dl_tables(imdgpu_ctx *ctx, const imdgpu_tricub_table *tabs)
{
  NEED_TTM();
  FdtdState *S = t->fdtd;
  if (!S) return imdgpu_fail(ctx, "ttm_fdtd_set_dl_tables: imdgpu_ttm_fdtd_init first");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (void *p : S->tab_bufs) (void) hipFree(p);
  S->tab_bufs.clear();
  if (!tabs) { S->d.dl_mode = 0; S->d.tab = nullptr; return 0; }
  for (int q = 0; q < 5; q++) {
    const imdgpu_tricub_table &h = tabs[q];
    if (h.is < 1 || h.js < 1 || h.ks < 1 || !h.x || !h.y || !h.z || !h.f) return imdgpu_fail(ctx, "ttm_fdtd_set_dl_tables: empty table");
    TricubDev &d = S->h_tab[q];
    d.is = h.is; d.js = h.js; d.ks = h.ks;
    d.xmin = h.range[0]; d.xmax = h.range[1]; d.ymin = h.range[2]; d.ymax = h.range[3]; d.zmin = h.range[4]; d.zmax = h.range[5];
    d.dx = (d.xmax - d.xmin) / ((double) d.is);                  // read_tricub_interp, imd_interpol.c:1008-1010
    d.dy = (d.ymax - d.ymin) / ((double) d.js);
    d.dz = (d.zmax - d.zmin) / ((double) d.ks);
    const size_t nn[4] = {(size_t) h.is, (size_t) h.js, (size_t) h.ks, (size_t) h.is * h.js * h.ks};
    const double *hs[4] = {h.x, h.y, h.z, h.f};
    const double **ds[4] = {&d.x, &d.y, &d.z, &d.f};
    for (int a = 0; a < 4; a++) {
      void *buf = nullptr;
      HIPCHK(hipMalloc(&buf, nn[a] * sizeof(double)));
      S->tab_bufs.push_back(buf);
      HIPCHK(hipMemcpy(buf, hs[a], nn[a] * sizeof(double), hipMemcpyHostToDevice));
      *ds[a] = (const double *) buf;
    }
  }
  int same = 1;                                                  // one grid for all five quantities?
  for (int q = 1; q < 5 && same; q++) {
    const imdgpu_tricub_table &a = tabs[0], &b = tabs[q];
    same = a.is == b.is && a.js == b.js && a.ks == b.ks && !memcmp(a.range, b.range, sizeof(a.range)) &&
           !memcmp(a.x, b.x, a.is * sizeof(double)) && !memcmp(a.y, b.y, a.js * sizeof(double)) && !memcmp(a.z, b.z, a.ks * sizeof(double));
  }
  for (int q = 0; q < 5; q++) S->h_tab[q].same = same;
  if (!S->d_tab) HIPCHK(hipMalloc((void **) &S->d_tab, 5 * sizeof(TricubDev)));
  HIPCHK(hipMemcpy(S->d_tab, S->h_tab, 5 * sizeof(TricubDev), hipMemcpyHostToDevice));
  S->d.tab = S->d_tab; S->d.dl_mode = 2;
  return 0;
}

__global__ void k_tricub_lookup(const TricubDev *tab, int which, int n, const double *xyz, double *out)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = tricub_lookup(tab[which], xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]);
}
__global__ void k_fit_dl(const TricubDev *tab, int n, const double *temp, const double *md_temp, const double *dens, const double *ne,
                         double *DL)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) { double d[6]; fit_dl(tab, temp[q], md_temp[q], dens[q], ne[q], d); for (int a = 0; a < 6; a++) DL[6 * q + a] = d[a]; }
}

// the device's do_tricubinterp on table `which` (0..4) at n points (x, y, z triples), -1 outside the table like the reference
extern "C" int imdgpu_ttm_fdtd_dl_lookup(imdgpu_ctx *ctx, int which, int n, const double *xyz, double *out)
{
  NEED_TTM();
  if (!t->fdtd || t->fdtd->d.dl_mode != 2) return imdgpu_fail(ctx, "ttm_fdtd_dl_lookup: imdgpu_ttm_fdtd_set_dl_tables first");
  if (which < 0 || which > 4) return imdgpu_fail(ctx, "ttm_fdtd_dl_lookup: table 0..4");
  if (n <= 0) return 0;
  double *d_in = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc((void **) &d_in, 3 * (size_t) n * sizeof(double)));
  HIPCHK(hipMalloc((void **) &d_out, (size_t) n * sizeof(double)));
  HIPCHK(hipMemcpy(d_in, xyz, 3 * (size_t) n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_tricub_lookup, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, t->fdtd->d.tab, which, n, d_in, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d_out, (size_t) n * sizeof(double), hipMemcpyDeviceToHost));
  (void) hipFree(d_in); (void) hipFree(d_out);
  return 0;
}

// the device's fitDL for n cells given (temp, md_temp, dens, ne): DL[6 n] out
extern "C" int imdgpu_ttm_fdtd_fitdl(imdgpu_ctx *ctx, int n, const double *temp, const double *md_temp, const double *dens,
                                     const double *ne, double *DL)
{
  NEED_TTM();
  if (!t->fdtd || t->fdtd->d.dl_mode != 2) return imdgpu_fail(ctx, "ttm_fdtd_fitdl: imdgpu_ttm_fdtd_set_dl_tables first");
  if (n <= 0) return 0;
  double *d_in = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc((void **) &d_in, 4 * (size_t) n * sizeof(double)));
  HIPCHK(hipMalloc((void **) &d_out, 6 * (size_t) n * sizeof(double)));
  const double *hs[4] = {temp, md_temp, dens, ne};
  for (int a = 0; a < 4; a++) HIPCHK(hipMemcpy(d_in + (size_t) a * n, hs[a], (size_t) n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fit_dl, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, t->fdtd->d.tab, n, d_in, d_in + n, d_in + 2 * (size_t) n,
                     d_in + 3 * (size_t) n, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(DL, d_out, 6 * (size_t) n * sizeof(double), hipMemcpyDeviceToHost));
  (void) hipFree(d_in); (void) hipFree(d_out);
  return 0;
}

// what calc_ttm settles once per MD step (imd_ttm_3D.c:96-122): envelope at the start of the step, solver on/off,
// the number of FDTD sub-steps, the diffusion sub-steps inside each, and their lengths
static void fdtd_plan_step(const imdgpu_ctx *ctx, const TtmState *t, const FdtdState *S, int *nsub, int *ndiff, double *tau_f,
                           double *tau_d, int *active, double *timefun)
{
  const FdtdDev &F = S->d;
  const double d = S->t_SI + F.dt - F.t0;
  *timefun = F.E0 * exp(-0.5 * (d * d) / F.sigma_t_squared);
  *active = !(*timefun < S->E0_init * 1e-4);
  const double dt_md = ctx->timestep * 10.18 * 1.0e-15;
  if (*active) {
    *nsub = (int) (dt_md / F.dt + 1.0);
    int nd = std::max(*nsub, t->max_dt_ttm > 0 ? (int) (ctx->timestep / t->max_dt_ttm) : 0);
    nd = std::max(nd, t->nsub);
    nd = (int) (((double) nd) / ((double) *nsub));
    *ndiff = std::max(nd, 1);
    *tau_f = dt_md / ((double) *nsub);
    *tau_d = *tau_f / ((double) *ndiff) / (10.18 * 1.0e-15);
  } else {
    *nsub = std::max(t->nsub, t->max_dt_ttm > 0 ? (int) (ctx->timestep / t->max_dt_ttm) : 0);
    *ndiff = 1;
    *tau_f = dt_md / ((double) *nsub);
    *tau_d = *tau_f / (10.18 * 1.0e-15);
  }
}

// the solver alone: nsub x do_fdtd from time t_SI on with a fixed envelope value (function-level parity with do_fdtd);
// the dissipated power lands in the lattice's source array (row cells i, 1, 1)
extern "C" int imdgpu_ttm_fdtd_substeps(imdgpu_ctx *ctx, double t_SI, int nsub, double tau_FDTD, double laser_timefun)
{
  NEED_TTM();
  if (!t->fdtd) return imdgpu_fail(ctx, "ttm_fdtd_substeps: imdgpu_ttm_fdtd_init first");
  FdtdDev F = t->fdtd->d;
  const int thr = (F.nx + 63) / 64 * 64;
  // the bare solver works on a contiguous row: use a scratch row for the source and copy it into the lattice afterwards
  double *row = nullptr;
  HIPCHK(hipMalloc((void **) &row, F.nx * sizeof(double)));
  HIPCHK(hipMemsetAsync(row, 0, F.nx * sizeof(double), ctx->stream));
  const int npre = nsub <= 4096 ? nsub : 0;               // source terms evaluated ahead (LDS: 16 bytes per step)
  // rows of ~1000 cells with thousands of steps evaluated ahead need more than the 64 KB a launch gets by default
#define FD_BARE(MT) HIPCHK(hipFuncSetAttribute((const void *) k_ttm_fdtd_chain<false, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ((4 * F.nx + F.nx / 2 + 2 + 2 + 2 * npre) * sizeof(double)))); \
  hipLaunchKernelGGL((k_ttm_fdtd_chain<false, MT>), dim3(1), dim3(thr), (4 * F.nx + F.nx / 2 + 2 + 2 + 2 * npre) * sizeof(double), ctx->stream, L, F, nsub, 0, \
                     t_SI, tau_FDTD, 0.0, 1, laser_timefun, 1, t->natoms, (double *) nullptr, (double *) nullptr, (double *) nullptr, \
                     (double *) nullptr, (double *) nullptr, (double *) nullptr, (double *) nullptr, (const double *) nullptr, row, \
                     (const double *) nullptr, (double *) nullptr, (double *) nullptr, (double *) nullptr, t->cve, t->ee, (double *) nullptr, npre)
  if (thr <= 256) { FD_BARE(256); } else { FD_BARE(1024); }
#undef FD_BARE
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<double> h(F.nx), lat(t->ncell);
  HIPCHK(hipMemcpy(h.data(), row, F.nx * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(lat.data(), t->source, t->ncell * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 1; i < F.nx - 1; i++) lat[(i * t->dim[1] + 1) * t->dim[2] + 1] = h[i];
  HIPCHK(hipMemcpy(t->source, lat.data(), t->ncell * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipFree(row));
  t->fdtd->t_SI = t_SI + nsub * tau_FDTD;
  return 0;
}

// the lattice loop of one MD step with the Maxwell solver inside (called by imdgpu_ttm_lattice_step when the solver is on)
static int fdtd_lattice_loop(imdgpu_ctx *ctx, TtmState *t, const Lat &L, int steps)
{
  FdtdState *S = t->fdtd;
  int nsub, ndiff, active; double tau_f, tau_d, timefun;
  fdtd_plan_step(ctx, t, S, &nsub, &ndiff, &tau_f, &tau_d, &active, &timefun);
  Lat L2 = L;
  // do_DIFF's xi_fac divides by the GLOBAL diff_substeps (imd_ttm_3D.c:1614), which inside the solver loop is the number of
  // diffusion sub-steps per FDTD sub-step (:116-118), not their total per MD step: xi therefore accumulates
  // fdtd_substeps times the weight it has without the solver.  Reproduced as the source has it.
  L2.xi_fac = 26.9815 * C_AMU / 3.0 * 1e30 / ((double) ndiff);
  FdtdDev F = S->d;
  const int thr = (F.nx + 63) / 64 * 64;
  t->laser_active = active;
  ProfScope ps(ctx, PROF_TTM_DIFF);
  const int npre = nsub <= 4096 ? nsub : 0;
#define FD_LOOP(MT) HIPCHK(hipFuncSetAttribute((const void *) k_ttm_fdtd_chain<true, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ((4 * F.nx + F.nx / 2 + 2 + 2 + 2 * npre) * sizeof(double)))); \
  hipLaunchKernelGGL((k_ttm_fdtd_chain<true, MT>), dim3(1), dim3(thr), (4 * F.nx + F.nx / 2 + 2 + 2 + 2 * npre) * sizeof(double), ctx->stream, L2, F, nsub, ndiff, \
                     S->t_SI, tau_f, tau_d, active, timefun, steps, t->natoms, t->temp[t->cur], t->temp[t->cur ^ 1], t->U[t->cur], \
                     t->U[t->cur ^ 1], t->fd_k[t->kcur], t->Ce, t->fd_g, t->md_temp, t->source, t->dens, t->xi, t->Z, t->ne, t->cve, \
                     t->ee, t->d_eabs, npre)
  if (thr <= 256) { FD_LOOP(256); } else { FD_LOOP(1024); }
#undef FD_LOOP
  HIPCHK(hipGetLastError());
  S->t_SI += nsub * tau_f;
  return 0;
}

// =====================================================================================================================
// The reference's OWN 1D two-temperature model (imd_ttm.c, the file its Makefile builds for ttm targets, Makefile:185;
// -DTTM1D) -- SURVEY 8(f)4.  What differs from the 3D file: atoms are binned into slabs by POSITION
// ((int)(x / fd_h.x), imd_ttm.c:296-298) and the slab sums are global arrays known to every rank (:325-334); a cell is
// active when natoms >= fd_min_atoms AND dens > RHOMIN (= 2 kg/m^3, :69); a static VIRTUAL LATTICE of vlatdim cells
// continues the row behind the last active cell (:1013-1024, 1595-1650); the number of diffusion sub-steps follows the
// stability limit (:97-99); U is taken from the EOS after the loop (:119-133); move_atoms_ttm takes xi and v_com of the slab
// an atom is in NOW (imd_integrate.c:649-760).  The row is ONE global array here (the reference cuts it by rank id; its
// rank-layout behaviour that shows in results -- every cell of the rank owning the last active cell takes the virtual
// lattice's first cell as +x neighbour, :1535-1540 -- is reproduced through `ref_ranks`).  do_tmm (the transfer-matrix
// Helmholtz solver of imd_tmm.c, the model's laser source) and -DELECPRESS are included (imdgpu_ttm1d_tmm_init,
// imdgpu_ttm1d_set_epress); not included: COLRAD, BALLISTIC.
// Device: two atom sweeps with wave-reduced atomics (slab sums), then the WHOLE lattice part of calc_ttm -- reconstruction,
// do_ADV, activation, FILLMESH, CFL_maxdt, the data-dependent number of sub-steps, the EOS update -- in ONE single-
// workgroup launch, a thread per cell (real and virtual), neighbours through LDS: the row has at most ~1000 cells and the
// loop is sequential in time, like the Maxwell-solver loop of 4.7.
// =====================================================================================================================
#define T1D_MAX 1024
#define T1D_RHOMIN 2.0
struct Ttm1dState {
  int nx = 0, nv = 0, fd_min_atoms = 0, fd_n = 0, vlatbuffer = 0, ref_ranks = 1, adv = 0;
  double hx = 0, vol = 0, neighvol = 0, aw = 0, ac = 0, vlatdens = 0, Temin = 1e-3;
  int *natoms = nullptr, *natoms_old = nullptr;                      // [nx + 2]
  double *temp[2]{}, *U[2]{}, *xi = nullptr, *md_temp = nullptr, *source = nullptr, *dens = nullptr, *fd_k = nullptr, *fd_g = nullptr,
         *Z = nullptr, *Ce = nullptr, *ne = nullptr;                   // [nx + 2]
  int *gi = nullptr;                                                  // [4 nx]: natomsglobal, totneigh, fluxl, fluxr
  double *gd = nullptr;                                               // [5 nx]: vcom x, y, z, mdtemp sums, xiarr
  double *vt[2]{}, *vm[2]{}, *vk = nullptr, *vg = nullptr, *vCe = nullptr, *vZ = nullptr, *vne = nullptr;   // [nv]
  int *sc_i = nullptr; double *sc_d = nullptr;                        // scalars: cur, vcur, last_active, vlat_proc, nsub ; max_dt_ttm
  double *cve_z = nullptr, *ee_z = nullptr;
  EosDev cve{}, ee{};
  Lat L{};
  unsigned char *diag[4]{};                                           // triangulation maps of the cve, ee, phase and pressure tables
  double *sumbuf = nullptr;                                           // several ranks: the slab sums on their way through the all-reduce
  double *part = nullptr, *ovf = nullptr; int *gmin = nullptr; size_t part_cap = 0;   // update_fd's block partials
  EosDev pe{}; double *pe_z = nullptr; int ep_on = 0;                  // -DELECPRESS: EOS_pe_from_r_te's table (imdgpu_ttm1d_set_epress)
  double box_xx = 0;
  struct TmmDev *tmm = nullptr;                                       // transfer-matrix laser source (imdgpu_ttm1d_tmm_init), host copy
  double omega_laser_tmm = 0;
  double *tmm_K12 = nullptr, *tmm_phase_z = nullptr, *tmm_q = nullptr, *tmm_eps = nullptr, *tmm_sc = nullptr, *tmm_bf = nullptr;
  int *tmm_act = nullptr;
};
// do_tmm's parameters and device arrays (imd_tmm.c); see tmm_* below
struct TmmDev {
  int on;
  double I0, t0, sig2, t1, sig1_2, ecut_thresh, laser_threshold, k0, omega_las, omega_laser, ebb_re, ebb_im, lambda;
  const double *K12;                 // [1500][2]
  EosDev phase;
  double *q, *eps, *sc, *bf;         // q[nx]: tmm_Qabs (persists between calls); eps[2 nx]; sc[8]: tmm_time, laser_active, eabs, I_t, refl, trans, abs, ecut; bf[4 nx]
  int *act;                          // [nx]
};
struct T1dDev {
  int nx, nv, fd_min_atoms, fd_n, vlatbuffer, ref_ranks, ep_on;
  double hx, vol, neighvol, aw, vlatdens, Temin;
  EosDev pe;
  TmmDev M;
  int *natoms, *natoms_old;
  double *temp0, *temp1, *U0, *U1, *xi, *md_temp, *source, *dens, *fd_k, *fd_g, *Z, *Ce, *ne;
  int *gi; double *gd;
  double *vt0, *vt1, *vm0, *vm1, *vk, *vg, *vCe, *vZ, *vne;
  int *sc_i; double *sc_d;
  int *counters;                     // the context's step counters (CNT_TMM_SHORT)
};
// ---- complex arithmetic as the reference's compiler does it (C99 double complex under gcc on x86-64: products inline,
// quotients through libgcc's __divdc3 = Smith's method; csqrt like glibc's) ---------------------------------------------
struct cpx { double re, im; };
__device__ __forceinline__ cpx c_mk(double a, double b) { cpx r; r.re = a; r.im = b; return r; }
__device__ __forceinline__ cpx c_add(cpx a, cpx b) { return c_mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cpx c_sub(cpx a, cpx b) { return c_mk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cpx c_neg(cpx a) { return c_mk(-a.re, -a.im); }
__device__ __forceinline__ cpx c_mul(cpx a, cpx b) { return c_mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cpx c_scale(double f, cpx a) { return c_mk(f * a.re, f * a.im); }
__device__ __forceinline__ cpx c_div(cpx n, cpx dd)
{
  const double a = n.re, b = n.im, c = dd.re, d = dd.im;
  if (fabs(c) < fabs(d)) {
    const double ratio = c / d, denom = (c * ratio) + d;
    return c_mk(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
  }
  const double ratio = d / c, denom = (d * ratio) + c;
  return c_mk(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
}
__device__ __forceinline__ double c_abs2(cpx a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ cpx c_sqrt(cpx z)
{
  if (z.re == 0.0 && z.im == 0.0) return c_mk(0.0, z.im);
  const double d = sqrt(z.re * z.re + z.im * z.im);          // permittivities: no overflow to guard against (glibc: hypot)
  double r, sI;
  if (z.re > 0) { r = sqrt(0.5 * (d + z.re)); sI = 0.5 * (z.im / r); }
  else { sI = sqrt(0.5 * (d - z.re)); r = fabs(0.5 * (z.im / sI)); sI = copysign(sI, z.im); }
  return c_mk(r, sI);
}
// eiphi of do_tmm / EE: exp(Re(dx k i)) (cos(Im(dx k i)) + i sin(..)), dx k i = (-dx Im k) + i (dx Re k)
__device__ __forceinline__ cpx tmm_eiphi(double dx, cpx k)
{
  const double phi1 = -(dx * k.im), phi2 = dx * k.re, e = exp(phi1);
  return c_mk(e * cos(phi2), e * sin(phi2));
}
// eiphi and 1 / eiphi = exp(-phi1) (cos - i sin) from one exp and one sincos (the reference divides: last-bit differences)
__device__ __forceinline__ void tmm_eiphi2(double dx, cpx k, cpx *e, cpx *ie)
{
  const double phi1 = -(dx * k.im), phi2 = dx * k.re, m = exp(phi1), im = 1.0 / m;
  double sn, cs;
  sincos(phi2, &sn, &cs);
  *e = c_mk(m * cs, m * sn); *ie = c_mk(im * cs, -(im * sn));
}
// tmm_K1 / tmm_K2 (imd_tmm.c:662-695): `(int) nu * 30 - 1` truncates nu BEFORE the product, as the source has it
__device__ __forceinline__ double tmm_K(const double *K12, int col, double nu)
{
  if (nu < 1.0 / 30.0) return 1;
  if (nu < 50) {
    int i = (int) nu * 30 - 1;
    if (i < 0) i = 0;
    if (i > 1498) i = 1498;
    return K12[2 * i + col] + (30 * nu - i - 1) * (K12[2 * (i + 1) + col] - K12[2 * i + col]);
  }
  return col == 0 ? 315.0 / 8.0 / (nu * nu) - 675675.0 / 64.0 / (nu * nu * nu * nu) : 6.0 / (nu * nu) - 720.0 / (nu * nu * nu * nu);
}
// tmm_get_epsilon (imd_tmm.c:501-555): the wide-range permittivity of a cell at the laser wavelength; Te, Ti in eV.
// The interband term epsilon_bb at the wavelength is a constant of the run (interpolated on the host, :531-540).
__device__ cpx tmm_epsilon(const TmmDev &M, bool laser_active, double Te, double Ti, double Z, double Ne)
{
  Te *= 11604.52500617; Ti *= 11604.52500617;
  const double A1p = 4.41, A2p = 0.8, A3p = 0.7, A4p = 0.2;
  const double omega_las = M.omega_las;
  const double Ncr = C_ECONST * C_EMASS * omega_las * omega_las / C_ECHARGE / C_ECHARGE;                     // ne_cr
  const double Ni = Ne / Z;
  // x^(2/3), x^(1/3), x^1.5 as cbrt / sqrt forms (pow costs ~250 instructions here; the results differ in the last bit)
  const double cbn = cbrt(3.0 * C_PI * C_PI * Ne);
  const double EF = C_HBAR * C_HBAR * (cbn * cbn) / 2.0 / C_EMASS;                                          // fermi_E
  const double TF = 2.0 * EF / (3.0 * C_BOLTZMAN), VF = sqrt(2 * EF / C_EMASS);
  const double nu_met = 1.309202957843054e+11 * (A1p * Ti + A2p * Te * Te / TF);                            // numet
  const double rws = cbrt(3.0 / (4 * C_PI * Ni));                                                           // r0(ni)
  const double nu_max = A3p / rws * sqrt(VF * VF + C_BOLTZMAN * Te / C_EMASS);                              // numax
  double nu_pl;
  {                                                                                                         // nupl -> coulomb_log -> bMax / bMin
    const double ftz = TF;                                                            // fermi_T(ni * Z): ni Z = Ne up to the last bit
    const double teff = sqrt(Te * Te + ftz * ftz);
    const double wpl = sqrt((Ni * Z) * C_ECHARGE * C_ECHARGE / C_EMASS / C_ECONST);
    const double bmax = fmax(rws, sqrt(C_BOLTZMAN * teff / (C_EMASS)) / (laser_active ? fmax(omega_las, wpl) : wpl));
    const double bmin = fmax(Z * C_ECHARGE * C_ECHARGE / (C_BOLTZMAN * Te), C_HBAR / sqrt(2 * C_EMASS * C_BOLTZMAN * Te));
    const double coulomb = fmax(2.0, log(bmax / bmin));
    const double kT = C_BOLTZMAN * Te;
    nu_pl = 1.863839656495274e-40 * Z * Ne * coulomb / (kT * sqrt(kT));
  }
  const double ksi = 0.75 * sqrt(C_PI) * nu_pl / omega_las;
  cpx ebb = c_mk(M.ebb_re, M.ebb_im);
  {                                                                                                         // EOS_phase_from_r_ti (imd_ttm.c:2594-2637)
    const double ph = (double) ((int) round(eos_lookup(M.phase, Ni * 26.9815 * C_AMU, Ti)));
    if (fabs(ph) == 3 || fabs(ph) == 4 || fabs(ph) == 5) ebb = c_mk(0.0, 0.0);
  }
  const cpx un = c_mk(1.0, 0.0);
  // epsilon_met = epsilon_bb + 1 - Ne / Ncr / (1 + i min(nu_met, nu_max) / omega_las): real / complex through the quotient rule
  const cpx dm = c_add(un, c_mk(0.0 * (fmin(nu_met, nu_max) / omega_las), fmin(nu_met, nu_max) / omega_las));
  const cpx emet = c_sub(c_add(ebb, un), c_div(c_mk(Ne / Ncr, 0.0), dm));
  const double K1 = tmm_K(M.K12, 0, ksi), K2 = tmm_K(M.K12, 1, ksi);
  const cpx epl = c_sub(un, c_scale(Ne / Ncr, c_mk(K1, -(nu_pl / omega_las * K2))));
  const double w = exp(-A4p * Te / TF);
  return c_add(epl, c_scale(w, c_sub(emet, epl)));
}
// EE and Runge5 (imd_tmm.c:36-96): |E|^2 inside a layer and its adaptive integral over the layer
__device__ __forceinline__ double tmm_EE(double z, double delta, cpx kl, cpx Bp, cpx Bm)
{
  cpx e, ie;
  tmm_eiphi2(delta * z, kl, &e, &ie);
  const cpx Ez = c_add(c_mul(Bp, e), c_mul(Bm, ie));
  return Ez.re * Ez.re + Ez.im * Ez.im + 0.0 * 0.0 + 0.0 * 0.0;
}
__device__ double tmm_runge5(double delta, cpx kl, cpx Bp, cpx Bm)
{
  double dx = 1, cur_xpos = 0, result = 0;
  const double errval = 1e-5;
  int guard = 0;
  while (cur_xpos < 1 && dx > 1e-5 && guard++ < 100000) {
    const double k1 = dx / 3.0 * tmm_EE(cur_xpos, delta, kl, Bp, Bm), k3 = dx / 3.0 * tmm_EE(cur_xpos + dx / 3.0, delta, kl, Bp, Bm),
                 k4 = dx / 3.0 * tmm_EE(cur_xpos + 0.5 * dx, delta, kl, Bp, Bm), k5 = dx / 3.0 * tmm_EE(cur_xpos + dx, delta, kl, Bp, Bm);
    const double ERR = k1 - 4.5 * k3 + 4 * k4 - 0.5 * k5;
    if (ERR < 5.0 / 32 * errval) { result += 0.5 * (k1 + 4 * k4 + k5); cur_xpos += dx; dx *= 1.1; }
    else if (ERR > 5 * errval) { dx *= 0.5; continue; }
    else { result += 0.5 * (k1 + 4 * k4 + k5); cur_xpos += dx; }
    if (cur_xpos + dx > 1) dx = 1 - cur_xpos;
  }
  return result;
}

__device__ __forceinline__ int t1d_slab(double x, double hx, int nx)
{
  int g = (int) (x / hx);
  g = g > nx - 1 ? nx - 1 : g;
  return g < 0 ? 0 : g;
}
// the two atom sweeps of update_fd (:271-324 and :376-403), without atomics and in a fixed order (the same sums bit for bit
// from run to run).  Atoms are sorted by MD cell, x slowest: a block's run of consecutive atoms lies in a few neighbouring
// slabs.  Sweep kernel: every wave sums its atoms per slab by shuffles -- one round per slab present in the wave -- into its
// own LDS row of a T1D_WIN-slab window that starts at the block's lowest slab; the block adds its waves' rows in wave order
// and writes ONE partial per (block, window slab, quantity).  Combine kernel: a workgroup per slab adds the partials of
// the blocks whose window covers it, in block order.  (The first version added per lane with compare-and-swap loops: 1.1 ms
// for 370 000 atoms; per wave and slab with hardware fp64 atomics: 2.6 ms for 10 million, every wave of the chip on ~250
// addresses.)  An atom beyond its block's window (tiny systems: a block spans many cell layers) falls back to an atomic.
#define T1D_WIN 16
#define T1D_NQ(PHASE) ((PHASE) == 1 ? 7 : 1)
template <int PHASE>
__global__ __launch_bounds__(256) void k_ttm1d_bin(int n, int apt, int cap, int ntypes, int steps, double hx, int nx, const double4 *pos, const int *type,
                                                   const double *p, const double *mass, const int *numneighs, int *fdc, const double *gd, double *part,
                                                   int *gmin_out, double *ovf)
{
  constexpr int NQ = T1D_NQ(PHASE);
  __shared__ double acc[4][T1D_WIN][NQ];
  __shared__ int smin[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int base = blockIdx.x * 256 * apt;
  for (int q = tid; q < 4 * T1D_WIN * NQ; q += 256) (&acc[0][0][0])[q] = 0.0;
  // the block's lowest slab
  int gm = 1 << 30;
  for (int j = 0; j < apt; j++) {
    const int a = base + j * 256 + tid;
    if (a < n && (ntypes == 1 || type[a] == 0)) { const int g = t1d_slab(pos[a].x, hx, nx); gm = g < gm ? g : gm; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t2 = __shfl_xor(gm, o, 64); gm = t2 < gm ? t2 : gm; }
  if (lane == 0) smin[wv] = gm;
  __syncthreads();
  gm = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
  for (int j = 0; j < apt; j++) {
    const int a = base + j * 256 + tid;
    const bool on = a < n && (ntypes == 1 || type[a] == 0);
    int g = -1;
    double v[NQ];
    for (int q = 0; q < NQ; q++) v[q] = 0.0;
    if (on) {
      g = t1d_slab(pos[a].x, hx, nx);
      const double px = p[a], py = p[(size_t) cap + a], pz = p[2 * (size_t) cap + a];
      if constexpr (PHASE == 1) {                     // 0 atoms, 1 neighbours, 2 / 3 arrived from the left / right, 4..6 momentum
        v[0] = 1.0; v[1] = (double) numneighs[a]; v[4] = px; v[5] = py; v[6] = pz;
        const int old = fdc[a];
        if (steps > 0 && old != g) v[old > g ? 3 : 2] = 1.0;
        fdc[a] = g;
      } else {
        const double m = mass[a], ux = px / m - gd[g], uy = py / m - gd[nx + g], uz = pz / m - gd[2 * nx + g];
        v[0] = m * (ux * ux); v[0] += m * (uy * uy); v[0] += m * (uz * uz);
      }
    }
    unsigned long long todo = __ballot(on);
    while (todo) {                                   // one round per slab present in the wave
      const int lead = __ffsll((long long) todo) - 1;
      const int g0 = __shfl(g, lead, 64);
      const bool mine = on && g == g0;
      double w[NQ];
      for (int q = 0; q < NQ; q++) w[q] = mine ? v[q] : 0.0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        for (int q = 0; q < NQ; q++) w[q] += __shfl_xor(w[q], o, 64);
      if (lane == lead) {
        const int l = g0 - gm;
        if (l < T1D_WIN) for (int q = 0; q < NQ; q++) acc[wv][l][q] += w[q];
        else for (int q = 0; q < NQ; q++) unsafeAtomicAdd(&ovf[(size_t) q * nx + g0], w[q]);
      }
      todo &= ~__ballot(mine);
    }
  }
  __syncthreads();
  for (int q = tid; q < T1D_WIN * NQ; q += 256) {
    const int l = q / NQ, k = q % NQ;
    part[((size_t) blockIdx.x * T1D_WIN + l) * NQ + k] = ((acc[0][l][k] + acc[1][l][k]) + acc[2][l][k]) + acc[3][l][k];
  }
  if (tid == 0) gmin_out[blockIdx.x] = gm;
}
// slab g = the partials of the blocks whose window covers it, in block order; PHASE 1 -> gi (natoms, neighbours, flux from the
// left / right) and gd (momentum sums), PHASE 2 -> gd[3 nx + g]
template <int PHASE>
__global__ __launch_bounds__(256) void k_ttm1d_combine(int nblocks, int nx, const double *part, const int *gmin, const double *ovf, int *gi, double *gd)
{
  constexpr int NQ = T1D_NQ(PHASE);
  __shared__ double red[256];
  const int g = blockIdx.x, tid = threadIdx.x;
  double s[NQ];
  for (int q = 0; q < NQ; q++) s[q] = 0.0;
  for (int b = tid; b < nblocks; b += 256) {
    const int l = g - gmin[b];
    if (l >= 0 && l < T1D_WIN) for (int q = 0; q < NQ; q++) s[q] += part[((size_t) b * T1D_WIN + l) * NQ + q];
  }
  for (int q = 0; q < NQ; q++) {
    red[tid] = s[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
      const double r = red[0] + ovf[(size_t) q * nx + g];
      if (PHASE == 1) { if (q < 4) gi[q * nx + g] = (int) r; else gd[(q - 4) * nx + g] = r; }
      else gd[3 * nx + g] = r;
    }
    __syncthreads();
  }
}
// v_com = sum p / (natoms * atomic_weight) (:350-372)
__global__ void k_ttm1d_vcom(int nx, double aw, const int *gi, double *gd)
{
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nx) return;
  const double tm = (double) gi[g] * aw;
  for (int d = 0; d < 3; d++) gd[d * nx + g] = tm > 0 ? gd[d * nx + g] / tm : 0.0;
}

// stages (bit mask): 1 the tail of update_fd (lattice from the slab sums, step-0 initialisation, virtual-lattice bookkeeping),
// 2 do_ADV, 4 do_cell_activation + do_FILLMESH + ghost cells, 8 CFL_maxdt and the sub-step count, 16 ONE do_DIFF,
// 32 do_FILLMESH + ghost cells, 64 the loop (diff_substeps x (do_DIFF, do_FILLMESH, ghosts)), 128 the EOS update of U
// MAXT: threads of the one workgroup (>= ttmdimx + vlatdim): 256 / 512 leave every wave 512 / 256 registers -- the cell's
// state, the wide-range model and do_tmm's complex arithmetic stay out of scratch memory --, 1024 spills (the loop is
// latency-bound: 29 us per sub-step with 319 spilled registers against 10 without, 135 active cells)
template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_ttm1d_calc(T1dDev D, Lat L, EosDev cve, EosDev ee, int steps, double timestep, int stages)
{
  // stage 256: ONE do_tmm (the transfer-matrix laser source) on the lattice as it stands
  __shared__ double sT[MAXT + 2], sK[MAXT + 2], sD[MAXT + 2], sU[MAXT + 2], vT[MAXT], vK[MAXT], red[16];
  __shared__ int sN[MAXT + 2], ired[16], s_last, s_nsub, s_ecut;
  __shared__ double s_maxdt, s_tmm[8];
  extern __shared__ double sM[];                 // do_tmm: the interface matrices Cinv P (8 doubles each), then Bplus / Bminus in their place
  const int tid = threadIdx.x, nx = D.nx, nv = D.nv, lane = tid & 63, wv = tid >> 6;
  // the virtual lattice's threads start at a wave boundary: a wave that held cells of both kinds would walk the wide-range
  // model twice per sub-step (once per branch) and every barrier would wait for it (25 -> 13 us per sub-step)
  const int vbase = (nx + 63) & ~63;
  const bool cellt = tid < nx, vlatt = tid >= vbase && tid < vbase + nv;
  const int c = tid + 1, g = tid, vi = tid - vbase;                    // lattice index / slab index / virtual-lattice index
  int cur = D.sc_i[0], vcur = D.sc_i[1], last_active = D.sc_i[2], vlat_proc = D.sc_i[3], nsub = D.sc_i[4];
  double max_dt = D.sc_d[0];
  double *T[2] = {D.temp0, D.temp1}, *U[2] = {D.U0, D.U1}, *VT[2] = {D.vt0, D.vt1}, *VM[2] = {D.vm0, D.vm1};
  const int amin = D.fd_min_atoms;
  auto act = [&](int n, double d) { return n >= amin && d > T1D_RHOMIN; };
  auto act_ge = [&](int n, double d) { return !(n < amin || d < T1D_RHOMIN); };
  const double AMU_J2EV = 26.9815 * C_AMU * C_J2EV;
  // ---- stage 1: update_fd's reconstruction (:409-470), step-0 initialisation (:493-503), VLATTICE (:507-542) -----------
  if (stages & 1) {
    if (tid == 0) { T[cur][0] = 0.0; T[cur][nx + 1] = 0.0; }                         // :245-246 at the global ends
    int cand = -5000;
    if (cellt) {
      D.source[c] = 0.0; D.natoms_old[c] = D.natoms[c]; D.xi[c] = 0.0;
      const int n = D.gi[g];
      double dn = 0.0, mt = 0.0;
      if (n > 0) {
        dn = ((double) D.gi[nx + g]) / ((double) n) * D.aw / D.neighvol * 1660.53907;
        if (dn == 0) dn = (double) n * D.aw / D.vol * 1660.53907;
        mt = D.gd[3 * nx + g] / (3.0 * (double) n);
        if (act(n, dn)) cand = g - D.vlatbuffer;
      }
      D.natoms[c] = n; D.dens[c] = dn; D.md_temp[c] = mt;
      if (steps < 1) { D.natoms_old[c] = n; if (act(n, dn)) { T[0][c] = mt; T[1][c] = mt; } }
      D.gd[4 * nx + g] = 0.0;                                                         // xiarr
    }
    if (steps == 0) {                                                                 // VLATTICE_STATIC: once
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const int t2 = __shfl_down(cand, o, 64); cand = t2 > cand ? t2 : cand; }
      if (lane == 0) ired[wv] = cand;
      __syncthreads();
      if (tid == 0) { int m = -5000; for (int q = 0; q < (int) (blockDim.x >> 6); q++) m = ired[q] > m ? ired[q] : m; s_last = m; }
      __syncthreads();
      last_active = s_last;
    }
    vlat_proc = (int) (((double) last_active * D.ref_ranks) / ((double) nx));
    if (cellt && g > last_active) D.natoms[c] = -1;
    __syncthreads();
  }
  // ---- stage 2: do_ADV (:1208-1309; do_COMMFLUX is nothing on one global row) -----------------------------------------
  if ((stages & 2) && steps >= 1) {
    if (cellt) sU[c] = U[cur][c];
    if (tid == 0) { sU[0] = U[cur][0]; sU[nx + 1] = U[cur][nx + 1]; }
    __syncthreads();
    if (cellt) {
      const double Nold = (double) D.natoms_old[c], Nnew = (double) D.natoms[c];
      if (Nnew > 0) {
        const double fr_here = D.gi[3 * nx + g], fl_next = g + 1 < nx ? D.gi[2 * nx + g + 1] : 0.0, fl_here = D.gi[2 * nx + g],
                     fr_prev = g > 0 ? D.gi[3 * nx + g - 1] : 0.0;
        const double u2 = sU[c] * Nold / Nnew + 1.0 * (+fr_here * sU[c + 1] - fl_next * sU[c] + fl_here * sU[c - 1] - fr_prev * sU[c]) / Nnew;
        U[cur ^ 1][c] = u2;
        if (act(D.natoms[c], D.dens[c])) T[cur ^ 1][c] = d_te_from_r_ee(ee, D.dens[c], u2 / AMU_J2EV) / 11604.5;
      } else if (Nnew < 1) { U[cur ^ 1][c] = 0.0; T[cur ^ 1][c] = 0.0; }
    }
    cur ^= 1;
    __syncthreads();
  }
  // own cell's state in registers from here on
  int nat = 0; double Te = 0, Te2 = 0, u = 0, u2 = 0, x = 0, Tl = 0, dn = 0, src = 0, kk = 0, gg = 0, ce = 0, Zc = 0, nec = 0;
  double vTe = 0, vTe2 = 0, vTl = 0, vTl2 = 0, vkk = 0, vgg = 0, vce = 0, vZc = 0, vnec = 0;
  if (cellt) { nat = D.natoms[c]; Te = T[cur][c]; Te2 = T[cur ^ 1][c]; u = U[cur][c]; u2 = U[cur ^ 1][c]; x = D.xi[c]; Tl = D.md_temp[c]; dn = D.dens[c];
               src = D.source[c]; kk = D.fd_k[c]; gg = D.fd_g[c]; ce = D.Ce[c]; Zc = D.Z[c]; nec = D.ne[c]; }
  if (vlatt) { vTe = VT[vcur][vi]; vTe2 = VT[vcur ^ 1][vi]; vTl = VM[vcur][vi]; vTl2 = VM[vcur ^ 1][vi]; vkk = D.vk[vi]; vgg = D.vg[vi]; vce = D.vCe[vi];
               vZc = D.vZ[vi]; vnec = D.vne[vi]; }
  double xiarr = cellt ? D.gd[4 * nx + g] : 0.0;
  // natoms / dens of the row do not change below: in LDS once (ghost cells: the global ends)
  if (cellt) { sN[c] = nat; sD[c] = dn; }
  if (tid == 0) { sN[0] = D.natoms[0]; sN[nx + 1] = D.natoms[nx + 1]; sD[0] = D.dens[0]; sD[nx + 1] = D.dens[nx + 1]; sT[0] = T[cur][0]; sT[nx + 1] = T[cur][nx + 1];
                  sK[0] = D.fd_k[0]; sK[nx + 1] = D.fd_k[nx + 1]; }
  __syncthreads();
  // transfer-matrix laser source (imd_tmm.c), state across the launches of a run
  const TmmDev &M = D.M;
  double tmm_time = 0.0, qab = 0.0, eacc = 0.0;
  bool laser_active = false;
  if (M.on) { tmm_time = M.sc[0]; laser_active = M.sc[1] != 0.0; if (cellt) qab = M.q[g]; }
  auto fillmesh = [&]() {                                                              // do_FILLMESH (:629-850), own cell
    L.omega_laser = laser_active ? M.omega_laser : 0.0;                                // bMax (:2207-2213): omega_las only while the laser is active
    if (cellt && !(g > last_active)) {
      if (act(nat, dn)) {
        const MeshOut o = mesh_props(L, cve, Te, Tl, dn);
        Zc = o.Z; nec = o.ne; ce = o.Ce; kk = o.k; gg = o.g;
        if (steps < 1) { u = eos_ee(ee, dn, Te * 11604.5) * AMU_J2EV; u2 = u; }
      } else { kk = 0; gg = 0; Zc = 0; nec = 0; ce = 0; Te = 0.0; }
    }
    if (vlatt) { const MeshOut o = mesh_props(L, cve, vTe, vTl, D.vlatdens); vZc = o.Z; vnec = o.ne; vkk = o.k; vce = o.Ce; vgg = o.g; }
  };
  auto publish = [&]() {                                                               // what the neighbours read: temp, fd_k (ttm_fill_ghost_layers: natoms 0 at the ends)
    __syncthreads();
    if (cellt) { sT[c] = Te; sK[c] = kk; }
    if (vlatt) { vT[vi] = vTe; vK[vi] = vkk; }
    if (tid == 0) { sN[0] = 0; sN[nx + 1] = 0; }
    __syncthreads();
  };
  publish();                        // the row's temp / fd_k as they stand, for whichever stage comes first
  // ---- stage 4: do_cell_activation (:1311-1425) + FILLMESH + ghosts ---------------------------------------------------
  if (stages & 4) {
    if (cellt) {
      const int nold = D.natoms_old[c];
      if (nold >= amin && nat < amin && dn > T1D_RHOMIN) x = 0.0;
      else if (nold < amin && nat >= amin && dn > T1D_RHOMIN && !(g > last_active)) {
        if (isnan(Te) || Te <= D.Temin) {
          int nn = 0; double e = 0.0;
          if (sN[c + 1] >= amin && sD[c + 1] >= T1D_RHOMIN) { e += sT[c + 1] * sT[c + 1]; nn++; }
          if (sN[c - 1] >= amin && sD[c - 1] >= T1D_RHOMIN) { e += sT[c - 1] * sT[c - 1]; nn++; }
          if (nn != 0) { Te = sqrt(e / (double) nn); if (Te < D.Temin) Te = Tl; } else Te = Tl;
          Te2 = Te;
          u = eos_ee(ee, dn, Te * 11604.5) * AMU_J2EV; u2 = u;
        }
      }
    }
    fillmesh();
    publish();
  }
  // ---- stage 8: CFL_maxdt (:2444-2481) and the sub-step count (:97-99) ------------------------------------------------
  if (stages & 8) {
    double mx = 1e9;
    if (cellt && act(nat, dn)) { const int m = (sN[c + 1] >= amin && sD[c + 1] >= T1D_RHOMIN) ? c + 1 : c; mx = ce * (D.hx * D.hx) / (kk + sK[m]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmin(mx, __shfl_down(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    if (tid == 0) {
      double m = 1e9;
      for (int q = 0; q < (int) (blockDim.x >> 6); q++) m = fmin(m, red[q]);
      m = (steps > 0 ? m : timestep / (double) D.fd_n) * 0.9;
      int ns = (int) (timestep / m);
      s_maxdt = m; s_nsub = ns > D.fd_n ? ns : D.fd_n;
    }
    __syncthreads();
    max_dt = s_maxdt; nsub = s_nsub;
  }
  // ---- stages 16 / 32 / 64: do_DIFF (:1473-1680), FILLMESH, the loop ---------------------------------------------------
  const int per = nx / D.ref_ranks;
  double Ci = 8.589449886e+02; Ci *= D.vlatdens; Ci *= 11604.5; Ci *= 1e-30; Ci *= C_J2EV;
  const double invxsq = 1.0 / (D.hx * D.hx);
  auto diff = [&](double tau) {
    const double xi_fac = 26.9815 * C_AMU / 3.0 * 1e30 / ((double) nsub);
    if (laser_active && cellt && !(g > last_active)) eacc += src * D.vol * tau;       // :1515-1518
    if (cellt && !(g > last_active) && !(nat < amin || dn < T1D_RHOMIN)) {
      const int xmin = act_ge(sN[c - 1], sD[c - 1]) ? c - 1 : c, xmax = act_ge(sN[c + 1], sD[c + 1]) ? c + 1 : c;
      double xmaxTe = sT[xmax], xminTe = sT[xmin], xmaxk = sK[xmax], xmink = sK[xmin];
      if (g / per == vlat_proc) { xmaxTe = vT[0]; xmaxk = vK[0]; }                     // :1535-1540: every cell of that rank of the reference
      const double Tn = tau / ce * ((((kk + xmaxk) / 2 * (xmaxTe - Te) * invxsq) - ((kk + xmink) / 2 * (Te - xminTe) * invxsq)) - gg * (Te - Tl) + src) + Te;
      u2 = u + (Tn - Te) * ce * D.vol / ((double) nat);
      const double dxi = (Tn - Tl) * xi_fac * gg / Tl / dn;
      x += dxi; xiarr += dxi;
      Te2 = Tn;
    }
    if (vlatt) {
      const int cl = last_active + 1;
      const double xminTe = vi == 0 ? sT[cl] : vT[vi - 1], xmink = vi == 0 ? sK[cl] : vK[vi - 1];
      const double xmaxTe = vi < nv - 1 ? vT[vi + 1] : vT[vi], xmaxk = vi < nv - 1 ? vK[vi + 1] : vK[vi];
      vTe2 = tau / vce * (((vkk + xmaxk) / 2 * (xmaxTe - vTe) * invxsq) - ((vkk + xmink) / 2 * (vTe - xminTe) * invxsq) - vgg * (vTe - vTl)) + vTe;
      vTl2 = vTl + tau / Ci * vgg * (vTe - vTl);
    }
    // l1 <-> l2, vlattice1 <-> vlattice2 (:1654-1662): the registers change places
    { double t2 = Te; Te = Te2; Te2 = t2; t2 = u; u = u2; u2 = t2; t2 = vTe; vTe = vTe2; vTe2 = t2; t2 = vTl; vTl = vTl2; vTl2 = t2; }
    cur ^= 1; vcur ^= 1;
  };
  // do_tmm (imd_tmm.c:193-455): permittivity per cell, the stack's transfer matrices up to the absorption depth, the field
  // amplitudes of every layer, the absorbed power per layer -> source.  The reference solves on rank 0 and scatters; here the
  // per-cell parts (permittivity, interface matrix, Runge5 quadrature) are the cells' threads, the two recursions over the
  // layers -- the matrix product that finds the absorption depth, the amplitude vector -- run on thread 0 in the reference's order.
  auto tmm = [&]() {
    const double Imp = 3.769911184307751e+02;
    double I_t = M.I0 * exp(-((tmm_time - M.t0) * (tmm_time - M.t0)) / M.sig2);
    I_t += M.I0 * exp(-((tmm_time - M.t1) * (tmm_time - M.t1)) / M.sig1_2);
    if (tid == 0) M.sc[3] = I_t;
    if (steps < 2) return;
    if (sqrt(2.0 * I_t * Imp) < M.laser_threshold * sqrt(2 * M.I0 * Imp)) { laser_active = false; return; }
    laser_active = true;
    cpx km = c_mk(0.0, 0.0);                                                         // km[nx - 1] is never set (:296-306)
    if (cellt) {
      const bool a = !(nat < amin || dn < T1D_RHOMIN);
      cpx e = c_mk(0.0, 0.0);
      if (a) e = tmm_epsilon(M, true, Te, Tl, Zc, nec);
      M.eps[g] = e.re; M.eps[nx + g] = e.im; M.act[g] = a ? 1 : 0;
      if (g < nx - 1) {                                                               // :296-306
        if (g == 0 || !a) km = c_mk(M.k0, 0.0);
        else { const cpx sq = c_sqrt(e); km = c_mk(M.k0 * sq.re, M.k0 * sq.im); }       // k0 * csqrt(eps): real x complex
      }
      sM[8 * g] = km.re; sM[8 * g + 1] = km.im;
    }
    __syncthreads();
    cpx kr = c_mk(0.0, 0.0);
    if (cellt && g < nx - 1) kr = c_mk(sM[8 * (g + 1)], sM[8 * (g + 1) + 1]);
    __syncthreads();
    cpx p0 = c_mk(1.0, 0.0), p1 = c_mk(0.0, 0.0), p2 = c_mk(0.0, 0.0), p3 = c_mk(1.0, 0.0);
    const bool iface = cellt && g < nx - 1;
    if (iface) {                                                                      // tmpmat = Cinv P of interface g -> g + 1 (:318-336)
      const double dx = g == 0 ? 1e10 : D.hx;
      cpx e, ie;
      tmm_eiphi2(dx, km, &e, &ie);
      const cpx r = c_div(c_scale(0.5, km), kr);
      const cpx P0 = e, P2 = ie, P1 = e, P3 = c_neg(ie), C0 = c_mk(0.5, 0.0), C2 = r, C1 = c_mk(0.5, 0.0), C3 = c_neg(r);
      p0 = c_add(c_mul(C0, P0), c_mul(C2, P1)); p1 = c_add(c_mul(C1, P0), c_mul(C3, P1));
      p2 = c_add(c_mul(C0, P2), c_mul(C2, P3)); p3 = c_add(c_mul(C1, P2), c_mul(C3, P3));
    }
    // E_g = M_g M_(g-1) ... M_0 for every interface at once: an inclusive scan of 2x2 complex products (the reference multiplies
    // them one after the other, :310-345; the product is associative, the roundings differ in the last bits)
    double *mine = sM + 8 * g;
    if (iface) { mine[0] = p0.re; mine[1] = p0.im; mine[2] = p1.re; mine[3] = p1.im; mine[4] = p2.re; mine[5] = p2.im; mine[6] = p3.re; mine[7] = p3.im; }
    __syncthreads();
    for (int d = 1; d < nx - 1; d <<= 1) {
      cpx q0 = c_mk(1.0, 0.0), q1 = c_mk(0.0, 0.0), q2 = c_mk(0.0, 0.0), q3 = c_mk(1.0, 0.0);
      const bool on = iface && g >= d;
      if (on) { const double *o = sM + 8 * (g - d); q0 = c_mk(o[0], o[1]); q1 = c_mk(o[2], o[3]); q2 = c_mk(o[4], o[5]); q3 = c_mk(o[6], o[7]); }
      __syncthreads();
      if (on) {
        const cpx n0 = c_add(c_mul(p0, q0), c_mul(p2, q1)), n1 = c_add(c_mul(p1, q0), c_mul(p3, q1));
        const cpx n2 = c_add(c_mul(p0, q2), c_mul(p2, q3)), n3 = c_add(c_mul(p1, q2), c_mul(p3, q3));
        p0 = n0; p1 = n1; p2 = n2; p3 = n3;
        mine[0] = p0.re; mine[1] = p0.im; mine[2] = p1.re; mine[3] = p1.im; mine[4] = p2.re; mine[5] = p2.im; mine[6] = p3.re; mine[7] = p3.im;
      }
      __syncthreads();
    }
    // reflected / transmitted amplitude behind every interface (:347-348, B0 = 1) and the first one below the threshold (:350-356)
    cpx BR = c_mk(0.0, 0.0), BT = c_mk(0.0, 0.0);
    int first = 1 << 30;
    if (iface) {
      BR = c_div(c_neg(p1), p3);
      BT = c_add(p0, c_mul(p2, BR));
      if (c_abs2(BT) < M.ecut_thresh) first = g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t2 = __shfl_down(first, o, 64); first = t2 < first ? t2 : first; }
    if (lane == 0) ired[wv] = first;
    __syncthreads();
    if (tid == 0) {
      int m = 1 << 30;
      for (int q = 0; q < (int) (blockDim.x >> 6); q++) m = ired[q] < m ? ired[q] : m;
      s_ecut = m == (1 << 30) ? 0 : m + 1;
      if (m == (1 << 30)) { D.sc_i[5] = 1; D.counters[CNT_TMM_SHORT] = 1; }           // "Sample is too short for TMM": the host reports it
    }
    __syncthreads();
    const int ecut = s_ecut;
    if (iface && g == ecut - 1) {
      s_tmm[0] = 1.0; s_tmm[1] = 0.0; s_tmm[2] = BR.re; s_tmm[3] = BR.im;
      M.sc[4] = c_abs2(BR); M.sc[5] = c_abs2(BT); M.sc[6] = 1.0 - c_abs2(BR) - c_abs2(BT); M.sc[7] = (double) ecut;      // tmm_refl, tmm_trans, tmm_abs (:366-370)
    }
    __syncthreads();
    if (cellt && g < ecut) {                                                          // amplitudes in layer g: E_(g-1) (B0, BR) (:377-420), then :424-441
      const cpx F0 = c_mk(s_tmm[0], s_tmm[1]), F1 = c_mk(s_tmm[2], s_tmm[3]);
      cpx Bp = F0, Bm = F1;
      if (g > 0) {
        const double *o = sM + 8 * (g - 1);
        const cpx a0 = c_mk(o[0], o[1]), a1 = c_mk(o[2], o[3]), a2 = c_mk(o[4], o[5]), a3 = c_mk(o[6], o[7]);
        Bp = c_add(c_mul(a0, F0), c_mul(a2, F1)); Bm = c_add(c_mul(a1, F0), c_mul(a3, F1));
      }
      if (M.bf) { M.bf[4 * g] = Bp.re; M.bf[4 * g + 1] = Bp.im; M.bf[4 * g + 2] = Bm.re; M.bf[4 * g + 3] = Bm.im; }
      const double dx = g == 0 ? 1e10 : D.hx;
      // cells without absorption (vacuum: eps_imag = 0) get 0 x a finite integral; the quadrature of layer 0 (1e10 A of vacuum:
      // the integrand oscillates 1e6 times) would otherwise halve its step 17 times on one thread for that zero
      const double ei = M.eps[nx + g];
      qab = ei == 0.0 ? 0.0 : I_t * M.k0 * ei * tmm_runge5(dx, km, Bp, Bm);
      qab *= 1e10; qab *= 6.3538562638e-26;
    }
    if (ecut > 0 && cellt) src = qab;                                                 // the scatter of tmm_Qabs, entries beyond ecut as earlier calls left them (:448-453)
    __syncthreads();
  };
  const double tau = timestep / ((double) (nsub > 0 ? nsub : 1));
  if (stages & 256) tmm();                                                            // one do_tmm by itself (function level)
  if (stages & 16) { diff(tau); if (!(stages & 32)) publish(); }
  if (stages & 32) { fillmesh(); publish(); }
  if (stages & 64) for (int s = 0; s < nsub; s++) {
    if (M.on) { tmm(); tmm_time += tau * 10.18 * 1.0e-15; }                           // imd_ttm.c:113-117
    diff(tau); fillmesh(); publish();
  }
  // ---- -DELECPRESS: do_electronic_pressure after the loop (:127-129, 553-623), part of stage 128 --------------------------
  // pressure of every active cell from the EOS, its centred difference per atom density -> gd[6 nx + g] for move_atoms_ttm
  if ((stages & 128) && D.ep_on) {
    __syncthreads();
    double pr = 0.0;
    if (cellt && !(dn < T1D_RHOMIN || nat < amin) && !(g > last_active)) pr = eos_lookup(D.pe, dn, Te * 11604.5) * 6.242e-12;
    if (cellt) { sT[c] = pr; D.gd[5 * nx + g] = pr; }
    __syncthreads();
    if (cellt) {
      double dv = 0.0;
      if (g >= 1 && g < nx - 1 && !(g > last_active) && steps > 1) dv = (sT[c + 1] - sT[c - 1]) / 2.0 / D.hx / ((double) nat / D.vol);
      D.gd[6 * nx + g] = dv;
    }
    __syncthreads();
  }
  // ---- stage 128: U from the EOS after the loop (:119-133) ------------------------------------------------------------
  if ((stages & 128) && cellt && nat >= amin && dn >= T1D_RHOMIN) { u = eos_ee(ee, dn, Te * 11604.5) * AMU_J2EV; u2 = u; }
  // ---- back to memory -------------------------------------------------------------------------------------------------
  if (cellt) { T[cur][c] = Te; T[cur ^ 1][c] = Te2; U[cur][c] = u; U[cur ^ 1][c] = u2; D.xi[c] = x; D.fd_k[c] = kk; D.fd_g[c] = gg; D.Ce[c] = ce; D.Z[c] = Zc;
               D.ne[c] = nec; D.gd[4 * nx + g] = xiarr; }
  if (M.on) {
    if (cellt) { D.source[c] = src; M.q[g] = qab; }
    __syncthreads();                                                                  // Eabs_local of this launch, summed in cell order by one thread
    sT[cellt ? c : 0] = cellt ? eacc : 0.0;
    __syncthreads();
    if (tid == 0) { double e = 0.0; for (int q = 1; q <= nx; q++) e += sT[q]; M.sc[2] += e; M.sc[0] = tmm_time; M.sc[1] = laser_active ? 1.0 : 0.0; }
  }
  if (vlatt) { VT[vcur][vi] = vTe; VT[vcur ^ 1][vi] = vTe2; VM[vcur][vi] = vTl; VM[vcur ^ 1][vi] = vTl2; D.vk[vi] = vkk; D.vg[vi] = vgg; D.vCe[vi] = vce;
               D.vZ[vi] = vZc; D.vne[vi] = vnec; }
  if (tid == 0) { D.natoms[0] = sN[0]; D.natoms[nx + 1] = sN[nx + 1];
                  D.sc_i[0] = cur; D.sc_i[1] = vcur; D.sc_i[2] = last_active; D.sc_i[3] = vlat_proc; D.sc_i[4] = nsub; D.sc_d[0] = max_dt; }
}

// move_atoms_ttm, the TTM1D branch (imd_integrate.c:649-760) + check_nblist: xi and v_com of the slab the atom is in now
__global__ void k_move_ttm1d(int n, int cap, int ntypes, double hx, int nx, double4 *pos, const int *type, double *p, double *f, const double *mass,
                             const double *restr, double dt, const double *ref, const double *gd, double *partials, int ep_ranks, double box_xx)
{
  // ep_ranks > 0: -DELECPRESS, the pressure force of the atom's slab (imd_integrate.c:657-659,741-743).  Every rank of the
  // reference fills epress_deriv for ITS share of the row only (imd_ttm.c:596-620), so an atom feels the force if the rank that
  // owns it also owns its slab: ep_ranks says how the run to be reproduced was cut (1: every atom)
  __shared__ double scratch[2 * 16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s0 = 0, s1 = 0;
  if (i < n) {
    const size_t c1 = cap, c2 = 2 * (size_t) cap;
    double4 q = pos[i];
    const int t = ntypes == 1 ? 0 : type[i];
    double fxi = 0, vx = 0, vy = 0, vz = 0, epf = 0;
    if (t == 0) {
      const int g = t1d_slab(q.x, hx, nx); fxi = gd[4 * nx + g]; vx = gd[g]; vy = gd[nx + g]; vz = gd[2 * nx + g];
      if (ep_ranks > 0) {
        int mdr = (int) (q.x / (box_xx / ep_ranks));
        mdr = mdr < 0 ? 0 : mdr > ep_ranks - 1 ? ep_ranks - 1 : mdr;
        if (mdr == g / (nx / ep_ranks)) epf = gd[6 * nx + g];
      }
    }
    double fx = f[i], fy = f[c1 + i], fz = f[c2 + i];
    double px = p[i], py = p[c1 + i], pz = p[c2 + i], m = mass[i];
    const double e1 = px * px + py * py + pz * pz;
    if (restr) { fx *= restr[3 * t]; fy *= restr[3 * t + 1]; fz *= restr[3 * t + 2]; f[i] = fx; f[c1 + i] = fy; f[c2 + i] = fz; }
    px += dt * (fx + fxi * m * (px / m - vx));
    py += dt * (fy + fxi * m * (py / m - vy));
    pz += dt * (fz + fxi * m * (pz / m - vz));
    if (ep_ranks > 0 && t == 0) px -= dt * epf;
    const double e2 = px * px + py * py + pz * pz;
    s0 = (e1 + e2) / (4 * m);
    const double tmp = dt / m;
    q.x += tmp * px; q.y += tmp * py; q.z += tmp * pz;
    pos[i] = q;
    p[i] = px; p[c1 + i] = py; p[c2 + i] = pz;
    const double dx = q.x - ref[i], dy = q.y - ref[c1 + i], dz = q.z - ref[c2 + i];
    s1 = dx * dx + dy * dy + dz * dz;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 = fmax(s1, __shfl_down(s1, o, 64)); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { scratch[w] = s0; scratch[16 + w] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double e = 0, mx = 0;
    for (int q2 = 0; q2 < nw; q2++) { e += scratch[q2]; mx = fmax(mx, scratch[16 + q2]); }
    partials[(size_t) blockIdx.x * 2] = e; partials[(size_t) blockIdx.x * 2 + 1] = mx;
  }
}

#define T1D(ctx) ((Ttm1dState *) (ctx)->ttm1d)
void imdgpu_ttm1d_free(imdgpu_ctx *ctx)
{
  Ttm1dState *t = T1D(ctx);
  if (!t) return;
  void *p[] = {t->natoms, t->natoms_old, t->temp[0], t->temp[1], t->U[0], t->U[1], t->xi, t->md_temp, t->source, t->dens, t->fd_k, t->fd_g, t->Z, t->Ce,
               t->ne, t->gi, t->gd, t->vt[0], t->vt[1], t->vm[0], t->vm[1], t->vk, t->vg, t->vCe, t->vZ, t->vne, t->sc_i, t->sc_d, t->cve_z, t->ee_z};
  for (void *q : p) if (q) (void) hipFree(q);
  void *p2[] = {t->tmm_K12, t->tmm_phase_z, t->tmm_q, t->tmm_eps, t->tmm_sc, t->tmm_bf, t->tmm_act, t->pe_z, t->sumbuf, t->part, t->ovf, t->gmin, t->diag[0], t->diag[1], t->diag[2], t->diag[3]};
  for (void *q : p2) if (q) (void) hipFree(q);
  delete t->tmm;
  delete t;
  ctx->ttm1d = nullptr;
}
static T1dDev t1d_dev(const Ttm1dState *t)
{
  T1dDev D;
  D.nx = t->nx; D.nv = t->nv; D.fd_min_atoms = t->fd_min_atoms; D.fd_n = t->fd_n; D.vlatbuffer = t->vlatbuffer; D.ref_ranks = t->ref_ranks;
  D.hx = t->hx; D.vol = t->vol; D.neighvol = t->neighvol; D.aw = t->aw; D.vlatdens = t->vlatdens; D.Temin = t->Temin;
  D.natoms = t->natoms; D.natoms_old = t->natoms_old; D.temp0 = t->temp[0]; D.temp1 = t->temp[1]; D.U0 = t->U[0]; D.U1 = t->U[1]; D.xi = t->xi;
  D.md_temp = t->md_temp; D.source = t->source; D.dens = t->dens; D.fd_k = t->fd_k; D.fd_g = t->fd_g; D.Z = t->Z; D.Ce = t->Ce; D.ne = t->ne;
  D.gi = t->gi; D.gd = t->gd; D.vt0 = t->vt[0]; D.vt1 = t->vt[1]; D.vm0 = t->vm[0]; D.vm1 = t->vm[1]; D.vk = t->vk; D.vg = t->vg; D.vCe = t->vCe;
  D.vZ = t->vZ; D.vne = t->vne; D.sc_i = t->sc_i; D.sc_d = t->sc_d;
  if (t->tmm) D.M = *t->tmm; else { memset(&D.M, 0, sizeof D.M); }
  D.counters = nullptr;
  D.ep_on = t->ep_on; D.pe = t->pe;
  return D;
}
static T1dDev t1d_dev_ctx(const imdgpu_ctx *ctx, const Ttm1dState *t) { T1dDev D = t1d_dev(t); D.counters = ctx->d_counters; return D; }
extern "C" int imdgpu_ttm1d_init(imdgpu_ctx *ctx, const imdgpu_ttm1d_params *par)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->have_geometry) return imdgpu_fail(ctx, "ttm1d_init: set_geometry first");
  if (!ctx->pair.present) return imdgpu_fail(ctx, "ttm1d_init: pair_pot needed (neighvol from pair_pot.end[0])");
  if (ctx->ttm) return imdgpu_fail(ctx, "ttm1d_init: a 3D lattice is attached to this context already");
  if (par->nx < 2 || ((par->nx + 63) & ~63) + par->vlatdim > T1D_MAX || par->vlatdim < 1)
    return imdgpu_fail(ctx, "ttm1d_init: 2 <= ttmdimx, 1 <= vlatdim, ttmdimx (rounded up to 64) + vlatdim <= %d", T1D_MAX);
  if (par->ref_ranks < 1 || par->nx % par->ref_ranks) return imdgpu_fail(ctx, "ttm1d_init: ttmdimx must be a multiple of ref_ranks (imd_ttm.c:962-965)");
  if (!par->cve_z || !par->ee_z) return imdgpu_fail(ctx, "ttm1d_init: EOS tables missing");
  imdgpu_ttm1d_free(ctx);
  Ttm1dState *t = new Ttm1dState();
  ctx->ttm1d = t;
  t->nx = par->nx; t->nv = par->vlatdim; t->fd_min_atoms = par->fd_min_atoms; t->fd_n = par->fd_n_timesteps; t->vlatbuffer = par->vlatbuffer;
  t->ref_ranks = par->ref_ranks; t->adv = par->adv; t->aw = par->atomic_weight; t->ac = par->atomic_charge; t->vlatdens = par->vlatdens;
  t->hx = ctx->box[0] / par->nx; t->vol = t->hx * ctx->box[4] * ctx->box[8];                          // init_ttm :972-976
  t->box_xx = ctx->box[0];
  t->neighvol = pow(sqrt(ctx->pair.par.end[0]), 3.0) * 4.0 / 3.0 * M_PI;                              // :979
  const size_t n2 = (size_t) t->nx + 2, nv = t->nv;
  if (dev_alloc(ctx, &t->natoms, n2) || dev_alloc(ctx, &t->natoms_old, n2) || dev_alloc(ctx, &t->xi, n2) || dev_alloc(ctx, &t->md_temp, n2) ||
      dev_alloc(ctx, &t->source, n2) || dev_alloc(ctx, &t->dens, n2) || dev_alloc(ctx, &t->fd_k, n2) || dev_alloc(ctx, &t->fd_g, n2) ||
      dev_alloc(ctx, &t->Z, n2) || dev_alloc(ctx, &t->Ce, n2) || dev_alloc(ctx, &t->ne, n2) || dev_alloc(ctx, &t->gi, 4 * (size_t) t->nx) ||
      dev_alloc(ctx, &t->gd, 7 * (size_t) t->nx) || dev_alloc(ctx, &t->vk, nv) || dev_alloc(ctx, &t->vg, nv) || dev_alloc(ctx, &t->vCe, nv) ||
      dev_alloc(ctx, &t->vZ, nv) || dev_alloc(ctx, &t->vne, nv) || dev_alloc(ctx, &t->sc_i, 8) || dev_alloc(ctx, &t->sc_d, 8)) return 1;
  for (int q = 0; q < 2; q++) if (dev_alloc(ctx, &t->temp[q], n2) || dev_alloc(ctx, &t->U[q], n2) || dev_alloc(ctx, &t->vt[q], nv) || dev_alloc(ctx, &t->vm[q], nv)) return 1;
  void *zi[] = {t->natoms, t->natoms_old}; for (void *p : zi) HIPCHK(hipMemset(p, 0, n2 * sizeof(int)));
  void *zd[] = {t->temp[0], t->temp[1], t->U[0], t->U[1], t->xi, t->md_temp, t->source, t->dens, t->fd_k, t->fd_g, t->Z, t->Ce, t->ne};
  for (void *p : zd) HIPCHK(hipMemset(p, 0, n2 * sizeof(double)));
  HIPCHK(hipMemset(t->gi, 0, 4 * (size_t) t->nx * sizeof(int))); HIPCHK(hipMemset(t->gd, 0, 5 * (size_t) t->nx * sizeof(double)));
  void *zv[] = {t->vk, t->vg, t->vCe, t->vZ, t->vne}; for (void *p : zv) HIPCHK(hipMemset(p, 0, nv * sizeof(double)));
  { std::vector<double> v(nv, 0.0264);                                                                    // :1017-1023
    for (int q = 0; q < 2; q++) { HIPCHK(hipMemcpy(t->vt[q], v.data(), nv * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(t->vm[q], v.data(), nv * 8, hipMemcpyHostToDevice)); } }
  { int si[8] = {0, 0, -5000, 0, par->fd_n_timesteps, 0, 0, 0}; double sd[8] = {ctx->timestep / (double) par->fd_n_timesteps, 0, 0, 0, 0, 0, 0, 0};   // :978
    HIPCHK(hipMemcpy(t->sc_i, si, sizeof si, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(t->sc_d, sd, sizeof sd, hipMemcpyHostToDevice)); }
  const size_t nc = (size_t) par->cve_nx * par->cve_ny, ne = (size_t) par->ee_nx * par->ee_ny;
  if (dev_alloc(ctx, &t->cve_z, nc) || dev_alloc(ctx, &t->ee_z, ne)) return 1;
  HIPCHK(hipMemcpy(t->cve_z, par->cve_z, nc * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(t->ee_z, par->ee_z, ne * 8, hipMemcpyHostToDevice));
  t->cve = EosDev{par->cve_nx, par->cve_ny, par->cve_xmin, par->cve_xmax, par->cve_ymin, par->cve_ymax, t->cve_z, nullptr};
  t->ee = EosDev{par->ee_nx, par->ee_ny, par->ee_xmin, par->ee_xmax, par->ee_ymin, par->ee_ymax, t->ee_z, nullptr};
  if (!ctx->eos_bilinear && (eos_attach_triangulation(ctx, t->cve, &t->diag[0]) || eos_attach_triangulation(ctx, t->ee, &t->diag[1]))) return 1;
  Lat &L = t->L;
  memset(&L, 0, sizeof L);
  L.fd_min_atoms = t->fd_min_atoms; L.vol = t->vol; L.neighvol = t->neighvol; L.atomic_weight = t->aw; L.atomic_charge = t->ac; L.omega_laser = 0.0;
  L.Temin = t->Temin; L.inv_z0_43 = 1e-3 / pow(t->ac, 4.0 / 3.0); L.inv_z0am = 1e-3 / (t->ac * t->aw); L.inv_awamu = 1.0 / (t->aw * C_AMU);
  ctx->count_ttm = 1;
  ctx->reserve_cus = 1;                           // the single-workgroup lattice launch needs a CU beside the persistent force kernels (4.7)
  HIPCHK(hipMemset(ctx->fdc, 0xff, (size_t) ctx->cap_local * sizeof(int)));
  return 0;
}
#define NEED_T1D()                                                                   \
  Ttm1dState *t = T1D(ctx);                                                          \
  if (!t) return imdgpu_fail(ctx, "imdgpu_ttm1d_init first");                       \
  HIPCHK(hipSetDevice(ctx->device))
// several ranks: the integer slab sums travel as doubles beside the momentum sums (exact far beyond any atom count)
__global__ void k_ttm1d_pack(int nx, int dir, int *gi, double *gd, double *buf)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 7 * nx) return;
  if (dir == 0) buf[q] = q < 4 * nx ? (double) gi[q] : gd[q - 4 * nx];
  else if (q < 4 * nx) gi[q] = (int) buf[q];
  else gd[q - 4 * nx] = buf[q];
}
static int t1d_update_fd_sums(imdgpu_ctx *ctx, Ttm1dState *t, int steps)
{
  const int n = ctx->nlocal, nx = t->nx;
  const bool many = ctx->dist != nullptr && ctx->cpu_dim[0] * ctx->cpu_dim[1] * ctx->cpu_dim[2] > 1;
  // atoms per block: 2048 where a block then still lies within a few cell layers (>= 64 blocks per... small systems: 256)
  const int apt = n >= 256 * 64 * 8 ? 8 : (n >= 256 * 64 * 2 ? 2 : 1);
  const int nb = n > 0 ? nblk(n, 256 * apt) : 0;
  if ((size_t) nb > t->part_cap) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (t->part) (void) hipFree(t->part);
    if (t->gmin) (void) hipFree(t->gmin);
    t->part = nullptr; t->gmin = nullptr;
    t->part_cap = (size_t) nb + nb / 8 + 16;
    if (dev_alloc(ctx, &t->part, t->part_cap * T1D_WIN * 7) || dev_alloc(ctx, &t->gmin, t->part_cap)) return 1;
  }
  if (!t->ovf && dev_alloc(ctx, &t->ovf, 7 * (size_t) nx)) return 1;
  HIPCHK(hipMemsetAsync(t->ovf, 0, 7 * (size_t) nx * sizeof(double), ctx->stream));
  ProfScope ps(ctx, PROF_TTM_UPDATE);
  if (nb > 0)
    hipLaunchKernelGGL((k_ttm1d_bin<1>), dim3(nb), dim3(256), 0, ctx->stream, n, apt, ctx->cap_local, ctx->ntypes, steps, t->hx, nx, ctx->pos, ctx->type,
                       ctx->p, ctx->mass, ctx->numneighs, ctx->fdc, t->gd, t->part, t->gmin, t->ovf);
  hipLaunchKernelGGL((k_ttm1d_combine<1>), dim3(nx), dim3(256), 0, ctx->stream, nb, nx, t->part, t->gmin, t->ovf, t->gi, t->gd);
  if (many) {                                       // [MPI_Allreduce of the slab arrays, imd_ttm.c:325-334,340-347]
    if (!t->sumbuf && dev_alloc(ctx, &t->sumbuf, 7 * (size_t) nx)) return 1;
    hipLaunchKernelGGL(k_ttm1d_pack, dim3(nblk(7 * nx, 256)), dim3(256), 0, ctx->stream, nx, 0, t->gi, t->gd, t->sumbuf);
    if (imdgpu_dist_sum_dev(ctx, t->sumbuf, 7 * nx)) return 1;
    hipLaunchKernelGGL(k_ttm1d_pack, dim3(nblk(7 * nx, 256)), dim3(256), 0, ctx->stream, nx, 1, t->gi, t->gd, t->sumbuf);
  }
  hipLaunchKernelGGL(k_ttm1d_vcom, dim3(nblk(nx, 256)), dim3(256), 0, ctx->stream, nx, t->aw, t->gi, t->gd);
  HIPCHK(hipMemsetAsync(t->ovf, 0, (size_t) nx * sizeof(double), ctx->stream));
  if (nb > 0)
    hipLaunchKernelGGL((k_ttm1d_bin<2>), dim3(nb), dim3(256), 0, ctx->stream, n, apt, ctx->cap_local, ctx->ntypes, steps, t->hx, nx, ctx->pos, ctx->type,
                       ctx->p, ctx->mass, ctx->numneighs, ctx->fdc, t->gd, t->part, t->gmin, t->ovf);
  hipLaunchKernelGGL((k_ttm1d_combine<2>), dim3(nx), dim3(256), 0, ctx->stream, nb, nx, t->part, t->gmin, t->ovf, t->gi, t->gd);
  if (many && imdgpu_dist_sum_dev(ctx, t->gd + 3 * (size_t) nx, nx)) return 1;             // [:404-407] the md_temp sums
  HIPCHK(hipGetLastError());
  return 0;
}
// function-level entry: the stages of k_ttm1d_calc (bit 0 = update_fd incl. the atom sweeps)
extern "C" int imdgpu_ttm1d_stage(imdgpu_ctx *ctx, int steps, int stages)
{
  NEED_T1D();
  if ((stages & 1) && t1d_update_fd_sums(ctx, t, steps)) return 1;
  ProfScope ps(ctx, PROF_TTM_DIFF);
  if ((stages & 256) && !t->tmm) return imdgpu_fail(ctx, "ttm1d_stage: stage 256 (do_tmm) needs imdgpu_ttm1d_tmm_init");
  const size_t dyn = t->tmm ? (size_t) 8 * t->nx * sizeof(double) : 0;            // the interface matrices of do_tmm
  const int need = ((t->nx + 63) & ~63) + t->nv;
  if (need <= 256) hipLaunchKernelGGL((k_ttm1d_calc<256>), dim3(1), dim3(256), dyn, ctx->stream, t1d_dev_ctx(ctx, t), t->L, t->cve, t->ee, steps, ctx->timestep, stages);
  else if (need <= 512) hipLaunchKernelGGL((k_ttm1d_calc<512>), dim3(1), dim3(512), dyn, ctx->stream, t1d_dev_ctx(ctx, t), t->L, t->cve, t->ee, steps, ctx->timestep, stages);
  else hipLaunchKernelGGL((k_ttm1d_calc<T1D_MAX>), dim3(1), dim3(T1D_MAX), dyn, ctx->stream, t1d_dev_ctx(ctx, t), t->L, t->cve, t->ee, steps, ctx->timestep, stages);
  HIPCHK(hipGetLastError());
  return 0;
}
// [calc_ttm, imd_ttm.c:75-155; do_tmm at the top of every sub-step once imdgpu_ttm1d_tmm_init has been called]
extern "C" int imdgpu_ttm1d_calc(imdgpu_ctx *ctx, int steps)
{
  NEED_T1D();
  return imdgpu_ttm1d_stage(ctx, steps, 1 | (t->adv ? 2 : 0) | 4 | 8 | 64 | 128);
}

// [tmm_init, imd_tmm.c:98-191] the transfer-matrix Helmholtz solver as laser source of the 1D model.  eps_bb: the rows of
// alu_eps_bb.dat (lambda [um], Re, Im), K12: the 1500 rows of K12.dat, phase: EOS_phase_from_r_ti's table.
extern "C" int imdgpu_ttm1d_tmm_init(imdgpu_ctx *ctx, const imdgpu_tmm_params *p)
{
  NEED_T1D();
  if (!p || !p->eps_bb || !p->K12 || !p->phase.z) return imdgpu_fail(ctx, "ttm1d_tmm_init: parameters and the three tables are needed");
  if (p->K12_rows < 1500) return imdgpu_fail(ctx, "ttm1d_tmm_init: K12 needs 1500 rows (tmm_K1 / tmm_K2 index up to row 1499, imd_tmm.c:662-695)");
  if (p->eps_bb_rows < 2 || !(p->lambda * 1e6 > p->eps_bb[0]) || !(p->lambda * 1e6 <= p->eps_bb[3 * (p->eps_bb_rows - 1)]))
    return imdgpu_fail(ctx, "ttm1d_tmm_init: lambda = %g um lies outside the interband table (%g .. %g um)", p->lambda * 1e6, p->eps_bb[0],
                       p->eps_bb[3 * (p->eps_bb_rows - 1)]);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int nx = t->nx;
  if (!t->tmm) t->tmm = new TmmDev();
  TmmDev &M = *t->tmm;
  void *old[] = {t->tmm_K12, t->tmm_phase_z, t->tmm_q, t->tmm_eps, t->tmm_sc, t->tmm_bf, t->tmm_act};
  for (void *q : old) if (q) (void) hipFree(q);
  const size_t nph = (size_t) p->phase.nx * p->phase.ny;
  if (dev_alloc(ctx, &t->tmm_K12, 2 * (size_t) p->K12_rows) || dev_alloc(ctx, &t->tmm_phase_z, nph) || dev_alloc(ctx, &t->tmm_q, nx) ||
      dev_alloc(ctx, &t->tmm_eps, 2 * (size_t) nx) || dev_alloc(ctx, &t->tmm_sc, 8) || dev_alloc(ctx, &t->tmm_bf, 4 * (size_t) nx) ||
      dev_alloc(ctx, &t->tmm_act, nx)) return 1;
  HIPCHK(hipMemcpy(t->tmm_K12, p->K12, 2 * (size_t) p->K12_rows * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t->tmm_phase_z, p->phase.z, nph * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(t->tmm_q, 0, nx * sizeof(double))); HIPCHK(hipMemset(t->tmm_eps, 0, 2 * (size_t) nx * sizeof(double)));
  HIPCHK(hipMemset(t->tmm_sc, 0, 8 * sizeof(double))); HIPCHK(hipMemset(t->tmm_bf, 0, 4 * (size_t) nx * sizeof(double)));
  HIPCHK(hipMemset(t->tmm_act, 0, nx * sizeof(int)));
  M.on = 1; M.I0 = p->I0; M.t0 = p->laser_t_0; M.sig2 = p->laser_sigma_t * p->laser_sigma_t; M.t1 = p->laser_t_1;
  M.sig1_2 = p->laser_sigma_t1 * p->laser_sigma_t1;
  M.ecut_thresh = exp(-p->tmm_threshold); M.laser_threshold = p->tmm_laser_threshold; M.lambda = p->lambda;
  M.k0 = 2.0 * M_PI / p->lambda; M.k0 /= 1e10;                                     // :154-155
  M.omega_laser = 2.0 * M_PI * (3.0e8 / p->lambda);                                // :152-153 with c0 = 3e8 (globals.h:499): what FILLMESH's Coulomb logarithm sees
  M.omega_las = (2.0 * M_PI * 2.998e8 / p->lambda);                                // tmm_get_epsilon's own, with LIGHTSPEED (:508, globals.h:325)
  {                                                                                // epsilon_bb at the wavelength (:519-540), evaluated once
    int il = 0, ir = 0;
    for (int id = 0; p->lambda * 1e6 > p->eps_bb[3 * id]; id++) { il = id; ir = id + 1; }
    const std::complex<double> l(p->eps_bb[3 * il + 1], p->eps_bb[3 * il + 2]), r(p->eps_bb[3 * ir + 1], p->eps_bb[3 * ir + 2]);
    const double dxl = p->eps_bb[3 * ir] - p->eps_bb[3 * il], w = p->lambda * 1e6 - p->eps_bb[3 * il];
    const std::complex<double> d = r - l;
    const std::complex<double> e = l + std::complex<double>(d.real() / dxl * w, d.imag() / dxl * w);
    M.ebb_re = e.real(); M.ebb_im = e.imag();
  }
  M.K12 = t->tmm_K12;
  M.phase.nx = p->phase.nx; M.phase.ny = p->phase.ny; M.phase.xmin = p->phase.xmin; M.phase.xmax = p->phase.xmax;
  M.phase.ymin = p->phase.ymin; M.phase.ymax = p->phase.ymax; M.phase.z = t->tmm_phase_z; M.phase.diag = nullptr;
  if (!ctx->eos_bilinear && eos_attach_triangulation(ctx, M.phase, &t->diag[2])) return 1;
  M.q = t->tmm_q; M.eps = t->tmm_eps; M.sc = t->tmm_sc; M.bf = t->tmm_bf; M.act = t->tmm_act;
  const size_t dyn = (size_t) 8 * nx * sizeof(double);
  HIPCHK(hipFuncSetAttribute((const void *) k_ttm1d_calc<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn));
  HIPCHK(hipFuncSetAttribute((const void *) k_ttm1d_calc<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn));
  HIPCHK(hipFuncSetAttribute((const void *) k_ttm1d_calc<T1D_MAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn));
  t->omega_laser_tmm = M.omega_laser;
  return 0;
}
// do_tmm's state: scalars[8] = tmm_time [s], laser_active, Eabs_local [eV], I_t, reflectivity, transmission, absorption, ecut;
// eps_re / eps_im / qabs [nx] (tmm_Qabs as the last call left it), active [nx]; NULL = skip.  too_short: a call found the
// sample too short for the absorption threshold (the reference aborts there)
extern "C" int imdgpu_ttm1d_tmm_get(imdgpu_ctx *ctx, double *scalars, double *eps_re, double *eps_im, double *qabs, int *active, int *too_short)
{
  NEED_T1D();
  if (!t->tmm) return imdgpu_fail(ctx, "ttm1d_tmm_get: imdgpu_ttm1d_tmm_init first");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->step_open) HIPCHK(hipStreamSynchronize(ctx->stream2));
  const size_t nx = t->nx;
  if (scalars) HIPCHK(hipMemcpy(scalars, t->tmm_sc, 8 * sizeof(double), hipMemcpyDeviceToHost));
  if (eps_re) HIPCHK(hipMemcpy(eps_re, t->tmm_eps, nx * sizeof(double), hipMemcpyDeviceToHost));
  if (eps_im) HIPCHK(hipMemcpy(eps_im, t->tmm_eps + nx, nx * sizeof(double), hipMemcpyDeviceToHost));
  if (qabs) HIPCHK(hipMemcpy(qabs, t->tmm_q, nx * sizeof(double), hipMemcpyDeviceToHost));
  if (active) HIPCHK(hipMemcpy(active, t->tmm_act, nx * sizeof(int), hipMemcpyDeviceToHost));
  if (too_short) { int si[8]; HIPCHK(hipMemcpy(si, t->sc_i, sizeof si, hipMemcpyDeviceToHost)); *too_short = si[5]; }
  return 0;
}
extern "C" int imdgpu_ttm1d_tmm_set_time(imdgpu_ctx *ctx, double tmm_time)
{
  NEED_T1D();
  if (!t->tmm) return imdgpu_fail(ctx, "ttm1d_tmm_set_time: imdgpu_ttm1d_tmm_init first");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(t->tmm_sc, &tmm_time, sizeof(double), hipMemcpyHostToDevice));
  return 0;
}
int imdgpu_ttm1d_move_internal(imdgpu_ctx *ctx)
{
  NEED_T1D();
  const int n = ctx->nlocal, nb = nblk(n, 256);
  const double h = 0.5 * ctx->nbl_margin;
  ProfScope ps(ctx, PROF_INTEGRATE);
  hipLaunchKernelGGL(k_move_ttm1d, dim3(nb), dim3(256), 0, ctx->stream, n, ctx->cap_local, ctx->ntypes, t->hx, t->nx, ctx->pos, ctx->type, ctx->p, ctx->f,
                     ctx->mass, ctx->restr_identity ? nullptr : ctx->d_restr, ctx->timestep, ctx->ref, t->gd, ctx->partials,
                     t->ep_on ? t->ref_ranks : 0, t->box_xx);
  hipLaunchKernelGGL(k_reduce_move, dim3(1), dim3(1024), 0, ctx->stream, nb, ctx->partials, ctx->d_sums, h * h, ctx->d_counters, 1);
  HIPCHK(hipGetLastError());
  return 0;
}
// -DELECPRESS [EOS_pe_from_r_te's table, imd_ttm.c:1133-1135,2638-2693, in Pa]: from now on calc_ttm ends with
// do_electronic_pressure (:127-129) and move_atoms_ttm adds the pressure force; z = NULL switches it off again
extern "C" int imdgpu_ttm1d_set_epress(imdgpu_ctx *ctx, int nx, int ny, double xmin, double xmax, double ymin, double ymax, const double *z)
{
  NEED_T1D();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (t->pe_z) { (void) hipFree(t->pe_z); t->pe_z = nullptr; }
  t->ep_on = 0;
  if (!z) return 0;
  if (nx < 2 || ny < 2) return imdgpu_fail(ctx, "ttm1d_set_epress: a table of at least 2 x 2 nodes");
  if (dev_alloc(ctx, &t->pe_z, (size_t) nx * ny)) return 1;
  HIPCHK(hipMemcpy(t->pe_z, z, (size_t) nx * ny * sizeof(double), hipMemcpyHostToDevice));
  t->pe.nx = nx; t->pe.ny = ny; t->pe.xmin = xmin; t->pe.xmax = xmax; t->pe.ymin = ymin; t->pe.ymax = ymax; t->pe.z = t->pe_z; t->pe.diag = nullptr;
  if (!ctx->eos_bilinear && eos_attach_triangulation(ctx, t->pe, &t->diag[3])) return 1;
  HIPCHK(hipMemset(t->gd + 5 * (size_t) t->nx, 0, 2 * (size_t) t->nx * sizeof(double)));
  t->ep_on = 1;
  return 0;
}
// epress_global and epress_deriv [ttmdimx values each] as the last calc_ttm left them
extern "C" int imdgpu_ttm1d_epress_get(imdgpu_ctx *ctx, double *epress, double *epress_deriv)
{
  NEED_T1D();
  if (!t->ep_on) return imdgpu_fail(ctx, "ttm1d_epress_get: imdgpu_ttm1d_set_epress first");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->step_open) HIPCHK(hipStreamSynchronize(ctx->stream2));
  if (epress) HIPCHK(hipMemcpy(epress, t->gd + 5 * (size_t) t->nx, t->nx * sizeof(double), hipMemcpyDeviceToHost));
  if (epress_deriv) HIPCHK(hipMemcpy(epress_deriv, t->gd + 6 * (size_t) t->nx, t->nx * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int imdgpu_ttm1d_move_atoms(imdgpu_ctx *ctx, double *ekin)
{
  if (imdgpu_ttm1d_move_internal(ctx)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->h_sums, ctx->d_sums, SUM_N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ekin) *ekin = ctx->h_sums[SUM_EKIN];
  return 0;
}
static int t1d_xfer(imdgpu_ctx *ctx, const imdgpu_ttm1d_fields *f, bool down)
{
  NEED_T1D();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->step_open) HIPCHK(hipStreamSynchronize(ctx->stream2));
  int si[8]; double sd[8];
  HIPCHK(hipMemcpy(si, t->sc_i, sizeof si, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(sd, t->sc_d, sizeof sd, hipMemcpyDeviceToHost));
  const int cur = si[0], vcur = si[1];
  const size_t n2 = (size_t) t->nx + 2, nx = t->nx, nv = t->nv;
  auto X = [&](void *host, void *dev, size_t bytes) -> int {
    if (!host) return 0;
    if (down) HIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); else HIPCHK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (X(f->natoms, t->natoms, n2 * 4) || X(f->natoms_old, t->natoms_old, n2 * 4) || X(f->temp, t->temp[cur], n2 * 8) || X(f->temp2, t->temp[cur ^ 1], n2 * 8) ||
      X(f->U, t->U[cur], n2 * 8) || X(f->U2, t->U[cur ^ 1], n2 * 8) || X(f->xi, t->xi, n2 * 8) || X(f->md_temp, t->md_temp, n2 * 8) || X(f->source, t->source, n2 * 8) ||
      X(f->dens, t->dens, n2 * 8) || X(f->fd_k, t->fd_k, n2 * 8) || X(f->fd_g, t->fd_g, n2 * 8) || X(f->Z, t->Z, n2 * 8) || X(f->Ce, t->Ce, n2 * 8) || X(f->ne, t->ne, n2 * 8) ||
      X(f->natomsglobal, t->gi, nx * 4) || X(f->fluxfromleft, t->gi + 2 * nx, nx * 4) || X(f->fluxfromright, t->gi + 3 * nx, nx * 4) ||
      X(f->vcomglobal, t->gd, 3 * nx * 8) || X(f->xiarr_global, t->gd + 4 * nx, nx * 8) ||
      X(f->vlat_temp, t->vt[vcur], nv * 8) || X(f->vlat_md_temp, t->vm[vcur], nv * 8) || X(f->vlat_fd_k, t->vk, nv * 8) || X(f->vlat_fd_g, t->vg, nv * 8) ||
      X(f->vlat_Ce, t->vCe, nv * 8))
    return 1;
  if (f->scalars) {
    if (down) { f->scalars[0] = si[2]; f->scalars[1] = si[3]; f->scalars[2] = si[4]; f->scalars[3] = sd[0]; }
    else { si[2] = (int) f->scalars[0]; si[3] = (int) f->scalars[1]; si[4] = (int) f->scalars[2]; sd[0] = f->scalars[3];
           HIPCHK(hipMemcpy(t->sc_i, si, sizeof si, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(t->sc_d, sd, sizeof sd, hipMemcpyHostToDevice)); }
  }
  return 0;
}
extern "C" int imdgpu_ttm1d_download(imdgpu_ctx *ctx, const imdgpu_ttm1d_fields *f) { return t1d_xfer(ctx, f, true); }
extern "C" int imdgpu_ttm1d_upload(imdgpu_ctx *ctx, const imdgpu_ttm1d_fields *f) { return t1d_xfer(ctx, f, false); }
