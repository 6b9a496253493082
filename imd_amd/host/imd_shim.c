/*
 * imd_shim.c -- drop-in replacement for imd_forces_nbl.o of IMD/lasMD.
 *
 * Compile this file INSIDE the IMD source tree (it includes imd.h, so it sees the
 * build's own `cell`, `pot_table_t` and globals) with the flags of your target
 * (e.g. -DNBL -DEAM2 -DNVE) and link it, together with libimdgpu.so, in place of
 * imd_forces_nbl.o.  It defines the reference's own entry points
 *
 *     void calc_forces(int steps);          (prototypes.h:399, imd_forces_nbl.c:295)
 *     void make_nblist(void);               (prototypes.h:421, imd_forces_nbl.c:136)
 *     void check_nblist(void);              (prototypes.h:422, imd_forces_nbl.c:1062)
 *     void deallocate_nblist(void);         (prototypes.h:423)
 *     int  estimate_nblist_size(void);      (prototypes.h:424)
 *
 * with the reference's semantics: calc_forces overwrites kraft, pot_eng, eam_rho,
 * eam_dF (and numneighs with TTM) of the inner cells and sets tot_pot_energy and
 * virial; list validity follows have_valid_nbl; errors go through error().
 * The host keeps integrating (move_atoms) and re-binning (fix_cells) as before;
 * this is integration level 1 (forces only).  Level 2, the device-resident loop
 * imdgpu_md_step, needs no IMD code at all.
 *
 * Builds WITHOUT -DNBL (cell-pair force loop: calc_forces of imd_main_risc_3d.c:31 calling do_forces,
 * imd_forces.c:33, do_embedding_energy and do_forces_eam2, imd_forces_eam2.c:202,239) are served by the
 * second half of this file: the same device kernels with a zero list margin, i.e. every call re-bins and
 * finds all pairs with r2 < cellsz exactly like the loop over cell pairs does.  There the host's cell array
 * has no buffer layer (periodic images come from the pair list's shift vectors), so the shim presents the
 * cells to the library with one added.  Compile imd_main_risc_3d.c with -Dcalc_forces=calc_forces_host
 * (or guard that definition) so that this calc_forces is the one that links.
 *
 * Builds with -DTTM (3D lattice, imd_ttm_3D.c) additionally get
 *
 *     void calc_ttm(void);                  (prototypes.h:1020, imd_ttm_3D.c:35)
 *
 * the first block of the reference's calc_ttm (:50-77: update_fd, [do_ADV,] do_cell_activation, do_FILLMESH,
 * ttm_fill_ghost_layers, diff_substeps x {do_DIFF, do_FILLMESH, ttm_fill_ghost_layers}) plus CFL_maxdt on the
 * device; the lattice comes back into l1/l2 so that move_atoms_ttm, ttm_writeout and everything else of the host
 * program read it as before.  init_ttm stays the reference's (it allocates l1/l2 and reads the EOS tables, which
 * the shim takes over from intp_cve_from_r_te / intp_ee_from_r_tesqrt -- they must be regular grids).  Compile
 * imd_ttm_3D.c with -Dcalc_ttm=calc_ttm_host so that this calc_ttm is the one that links.  With -DLASER and
 * laser_rescale_mode 4 the Beer-Lambert source of laser_rescale_ttm is applied on the device too.  do_ADV: the reference
 * cannot run it on one rank (do_COMMFLUX's self-Sendrecv never matches); IMD_SHIM_TTM_ADV=1 turns it on here.
 *
 * MPI builds with several ranks (one GPU per rank, or ranks sharing GPUs): the host keeps its own halo exchange --
 * send_cells(copy_cell, ...) before the forces and send_cells(copy_dF, ...) between the embedding and the EAM force
 * pass exactly where imd_forces_nbl.c has them (:311, :786), plus the transfer into the x = 0 buffer layer that AR
 * builds leave out (send_cells_west below) -- and hands the buffer cells to the library
 * (imdgpu_upload_ghosts / imdgpu_update_ghosts).  The two reverse exchanges of the reference (send_forces for rho
 * and for the forces, :745, :1052) are not needed: the device list is symmetric, every rank computes complete
 * forces and densities for its own atoms.  Sums are reduced with MPI_Allreduce like :1007.  calc_ttm on several ranks:
 * one lattice block per rank (option ttm_decomposed), the per-sub-step ghost faces through the library's mailbox in
 * host shared memory (the ranks of one node).
 */
#include "imd.h"
#include "imdgpu.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(TTM) && defined(MPI)
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#endif

static imdgpu_ctx *G = NULL;
static int   g_cap = 0, g_have_cells = 0;
static double *g_ort, *g_imp, *g_mas, *g_kraft, *g_epot, *g_rho, *g_dF;
static int    *g_sorte, *g_num, *g_nn, *g_cn;
static int    g_multi = 0;                    /* several ranks: ghosts come from the host's send_cells */
static int    g_gcap = 0, g_nghost = 0, *g_gcn = NULL, *g_gsorte = NULL;
static char   *g_inner = NULL;
static double *g_gort = NULL, *g_gdF = NULL;

static void gpu_check(int rc)
{
  if (rc) { char msg[1200]; sprintf(msg, "imdgpu: %s", imdgpu_last_error(G)); error(msg); }
}

static void gpu_init(void)
{
  double box[9];
  int pbc[3], cd[3], gd[3], cpu[3] = {1, 1, 1}, me[3] = {0, 0, 0}, dev = 0;
  if (G) return;
#ifdef MPI
  g_multi = cpu_dim.x * cpu_dim.y * cpu_dim.z != 1;
#ifndef NBL
  if (g_multi) error("imd_shim: several ranks need the neighbour-list build (-DNBL)");
#endif
  if (g_multi) {
    int ndev = 1;
    cpu[0] = cpu_dim.x; cpu[1] = cpu_dim.y; cpu[2] = cpu_dim.z;
    me[0] = my_coord.x; me[1] = my_coord.y; me[2] = my_coord.z;
    if (imdgpu_device_count(&ndev) || ndev < 1) error("imd_shim: no GPU");
    dev = getenv("IMD_SHIM_DEVICE") ? atoi(getenv("IMD_SHIM_DEVICE")) : myid % ndev;
  }
#else
  if (cpu_dim.x * cpu_dim.y * cpu_dim.z != 1) error("imd_shim: several ranks need an MPI build");
#endif
  gpu_check(imdgpu_create(&G, dev));
#ifdef _OPENMP
  /* copy loops: the cores of this rank's share (a one-GPU job: 16), not the whole host; IMD_SHIM_THREADS overrides */
  {
    int nranks = cpu_dim.x * cpu_dim.y * cpu_dim.z, nt = getenv("IMD_SHIM_THREADS") ? atoi(getenv("IMD_SHIM_THREADS")) : 16 / (nranks > 16 ? 16 : nranks);
    omp_set_num_threads(nt < 1 ? 1 : nt);
  }
#endif
  box[0] = box_x.x; box[1] = box_x.y; box[2] = box_x.z;
  box[3] = box_y.x; box[4] = box_y.y; box[5] = box_y.z;
  box[6] = box_z.x; box[7] = box_z.y; box[8] = box_z.z;
  pbc[0] = pbc_dirs.x; pbc[1] = pbc_dirs.y; pbc[2] = pbc_dirs.z;
#ifdef NBL
  cd[0] = cell_dim.x; cd[1] = cell_dim.y; cd[2] = cell_dim.z;
#else
  cd[0] = cell_dim.x + 2; cd[1] = cell_dim.y + 2; cd[2] = cell_dim.z + 2;      /* library cells carry a ghost layer */
#endif
  gd[0] = global_cell_dim.x; gd[1] = global_cell_dim.y; gd[2] = global_cell_dim.z;
  gpu_check(imdgpu_set_geometry(G, box, pbc, cd, gd, cpu, me));
  {
#ifdef NBL
    double margin = nbl_margin;
#else
    double margin = 0.0;
#endif
#ifdef TTM
    gpu_check(imdgpu_set_params(G, ntypes, cellsz, margin, timestep, 1));
#else
    gpu_check(imdgpu_set_params(G, ntypes, cellsz, margin, timestep, 0));
#endif
  }
  {
    /* pot_table_t and imdgpu_pot_table have the same members in the same order (types.h:537-549) */
    const imdgpu_pot_table *pp = NULL, *pr = NULL, *pe = NULL;
#ifdef PAIR
    pp = (const imdgpu_pot_table *) &pair_pot;
#endif
#ifdef EAM2
    pr = (const imdgpu_pot_table *) &rho_h_tab;
    pe = (const imdgpu_pot_table *) &embed_pot;
#endif
    gpu_check(imdgpu_upload_tables(G, pp, pr, pe));
  }
}

#ifdef NBL
#define SHIM_NALL nallcells
#define SHIM_CELLIDX(p) ((int) ((p) - cell_array))
#else
#define SHIM_NALL ((cell_dim.x + 2) * (cell_dim.y + 2) * (cell_dim.z + 2))
static int shim_padded_index(int k)       /* host cell k = (i*dim.y + j)*dim.z + l  ->  (i+1, j+1, l+1) in the padded array */
{
  int l = k % cell_dim.z, r = k / cell_dim.z, j = r % cell_dim.y, i = r / cell_dim.y;
  return ((i + 1) * (cell_dim.y + 2) + (j + 1)) * (cell_dim.z + 2) + (l + 1);
}
#define SHIM_CELLIDX(p) shim_padded_index((int) ((p) - cell_array))
#endif

static int count_atoms(void)
{
  int k, n = 0;
  for (k = 0; k < NCELLS; k++) n += CELLPTR(k)->n;
  return n;
}

/* the per-step staging arrays are page-locked (imdgpu_host_alloc): they cross PCIe whole, at link speed, and the library
 * reorders between its atom order and this cell order on the device */
static void *pin_realloc(void *old, size_t bytes)
{
  void *p = NULL;
  if (old) imdgpu_host_free(old);                 /* contents are rewritten before every use */
  if (imdgpu_host_alloc(&p, bytes)) error("imd_shim: cannot allocate page-locked staging arrays");
  return p;
}
static void ensure_host(int n)
{
  if (n <= g_cap && g_cn != NULL) return;        /* g_cn == NULL: first call, also for a rank without atoms (vacuum) */
  g_cap = n + n / 8 + 1024;
  g_ort = (double *) pin_realloc(g_ort, 3 * (size_t) g_cap * sizeof(double)); g_imp = (double *) pin_realloc(g_imp, 3 * (size_t) g_cap * sizeof(double));
  g_kraft = (double *) pin_realloc(g_kraft, 3 * (size_t) g_cap * sizeof(double));
  g_mas = (double *) pin_realloc(g_mas, g_cap * sizeof(double)); g_epot = (double *) pin_realloc(g_epot, g_cap * sizeof(double));
  g_rho = (double *) pin_realloc(g_rho, g_cap * sizeof(double)); g_dF = (double *) pin_realloc(g_dF, g_cap * sizeof(double));
  g_sorte = (int *) pin_realloc(g_sorte, g_cap * sizeof(int)); g_num = (int *) pin_realloc(g_num, g_cap * sizeof(int));
  g_nn = (int *) pin_realloc(g_nn, g_cap * sizeof(int));
  g_cn = (int *) realloc(g_cn, SHIM_NALL * sizeof(int));
  if (!g_cn) error("imd_shim: cannot allocate staging arrays");
}

/* walk the inner cells in the order of the `cells` list: that IS the ABI's cell order */
static int *g_off = NULL; static int g_off_cap = 0;
/* first flat index of every inner cell (cells list order); returns the number of atoms */
static int cell_offsets(void)
{
  int k, n = 0;
  if (NCELLS + 1 > g_off_cap) {
    g_off_cap = NCELLS + 1;
    g_off = (int *) realloc(g_off, g_off_cap * sizeof(int));
    if (!g_off) error("imd_shim: cannot allocate cell offsets");
  }
  for (k = 0; k < NCELLS; k++) { g_off[k] = n; n += CELLPTR(k)->n; }
  g_off[NCELLS] = n;
  return n;
}
/* The copy loops between the reference's per-cell arrays and the flat staging arrays run on all the cores of the rank's
 * share (OpenMP, when the shim is compiled with -fopenmp: the drop-in builds of oracle/build_ref.sh are); cells are
 * independent, every atom has its fixed flat index g_off[k] + i. */
static int gather_cells(int with_all)
{
  int k, n;
  for (k = 0; k < SHIM_NALL; k++) g_cn[k] = 0;
  n = cell_offsets();
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    int i, m = g_off[k];
    g_cn[SHIM_CELLIDX(p)] = p->n;
    for (i = 0; i < p->n; i++, m++) {
      g_ort[3 * m] = ORT(p, i, X); g_ort[3 * m + 1] = ORT(p, i, Y); g_ort[3 * m + 2] = ORT(p, i, Z);
      if (with_all) {
        g_imp[3 * m] = IMPULS(p, i, X); g_imp[3 * m + 1] = IMPULS(p, i, Y); g_imp[3 * m + 2] = IMPULS(p, i, Z);
        g_mas[m] = MASSE(p, i); g_sorte[m] = SORTE(p, i); g_num[m] = NUMMER(p, i);
      }
    }
  }
  return n;
}

#if defined(NBL) && defined(MPI)
/* With AR (actio = reactio; forced for NBL builds, config.h:150-153) the reference's send_cells only fills the buffer
 * layer at cell_dim.x-1: its half list lets every rank compute the pairs across the other face and send the forces
 * back.  The device list is symmetric and needs the layer at x = 0 as well: this is the transfer send_cells itself
 * makes without AR (imd_comm_force_3d.c:356-392), with the reference's own pack/unpack functions and buffers, after
 * the y and z stages like there. */
static void send_cells_west(void (*copy_func)(int, int, int, int, int, int, vektor),
                            void (*pack_func)(msgbuf *, int, int, int, vektor), void (*unpack_func)(msgbuf *, int, int, int))
{
  int i, j;
  vektor wvec = {0, 0, 0};
  if (pbc_dirs.x == 1 && my_coord.x == cpu_dim.x - 1) { wvec.x = -box_x.x; wvec.y = -box_x.y; wvec.z = -box_x.z; }
  if (cpu_dim.x == 1) {
    for (i = 0; i < cell_dim.y; ++i)
      for (j = 0; j < cell_dim.z; ++j) (*copy_func)(cell_dim.x - 2, i, j, 0, i, j, wvec);
  } else {
    MPI_Request req[2];
    MPI_Status  st[2];
    send_buf_west.n = 0; recv_buf_east.n = 0;
    for (i = 0; i < cell_dim.y; ++i)
      for (j = 0; j < cell_dim.z; ++j) (*pack_func)(&send_buf_west, cell_dim.x - 2, i, j, wvec);
    irecv_buf(&recv_buf_east, nbeast, &req[1]);
    isend_buf(&send_buf_west, nbwest, &req[0]);
    MPI_Waitall(2, req, st);
    recv_buf_east.n = 0;
    for (i = 0; i < cell_dim.y; ++i)
      for (j = 0; j < cell_dim.z; ++j) (*unpack_func)(&recv_buf_east, 0, i, j);
  }
}

/* the buffer cells as send_cells left them: counts per cell (inner cells 0), atoms in cell order */
static void gather_ghosts(int with_dF)
{
  int c, i, n = 0;
  if (NULL == g_inner) {
    g_inner = (char *) calloc(nallcells, 1); g_gcn = (int *) calloc(nallcells, sizeof(int));
    if (!g_inner || !g_gcn) error("imd_shim: cannot allocate ghost tables");
  }
  memset(g_inner, 0, nallcells);
  for (c = 0; c < NCELLS; c++) g_inner[SHIM_CELLIDX(CELLPTR(c))] = 1;
  for (c = 0; c < nallcells; c++) { g_gcn[c] = g_inner[c] ? 0 : cell_array[c].n; n += g_gcn[c]; }
  if (n > g_gcap) {
    g_gcap = n + n / 8 + 1024;
    g_gort = (double *) realloc(g_gort, 3 * g_gcap * sizeof(double)); g_gdF = (double *) realloc(g_gdF, g_gcap * sizeof(double));
    g_gsorte = (int *) realloc(g_gsorte, g_gcap * sizeof(int));
    if (!g_gort || !g_gdF || !g_gsorte) error("imd_shim: cannot allocate ghost staging arrays");
  }
  g_nghost = n;
  for (c = 0, n = 0; c < nallcells; c++) {
    cell *p = cell_array + c;
    if (g_inner[c]) continue;
    for (i = 0; i < p->n; i++, n++) {
      g_gort[3 * n] = ORT(p, i, X); g_gort[3 * n + 1] = ORT(p, i, Y); g_gort[3 * n + 2] = ORT(p, i, Z);
      g_gsorte[n] = SORTE(p, i);
#ifdef EAM2
      if (with_dF) g_gdF[n] = EAM_DF(p, i);
#endif
    }
  }
}
#endif

#ifdef NBL
/* ---- LEVEL 2 (IMD_SHIM_LEVEL=2; one rank, neighbour-list build): the atoms LIVE on the device ---------------------------
 * The reference's main loop stays what it is (imd_main_3d.c:398-812) -- calc_forces, [calc_ttm,] move_atoms, the writers,
 * check_nblist -- but its three hot calls only enqueue device work:
 *   calc_forces(steps) = imdgpu_md_step_begin  (fix_cells + make_nblist when the list is stale, the force passes, calc_ttm
 *                        forked beside them),
 *   calc_ttm()         = nothing left to do,
 *   move_atoms()       = imdgpu_md_step_end    (move_atoms_nve / move_atoms_ttm + check_nblist on the device); the global
 *                        function pointer `move_atoms` (globals.h:1777) is re-pointed here at the first calc_forces,
 *   check_nblist()     = have_valid_nbl from the device's flag.
 * The host's cell arrays are brought up to date only when the host is about to READ them: after move_atoms on checkpoint
 * steps (write_config, imd_main_3d.c:702-705) and on the last step, after calc_forces on force_int steps, and on the
 * steps listed in IMD_SHIM_SYNC_STEPS (comma separated; "host looks at the atoms after calc_forces of these steps").
 * tot_pot_energy / virial / tot_kin_energy follow on eng_int steps (write_eng_file), the lattice l1/l2 on ttm_int steps.
 * Atoms change cells on the device only; a sync rebuilds the host cells from the device's arrangement (alloc_cell as
 * needed).  Host-side edits of atoms between syncs are not seen by the device. */
static int g_level2 = -1, g_dev_atoms = 0;
static long g_moves = 0, g_synced_moves = -1;    /* integrator calls on the device / the one after which the host cells were last refreshed */
static void (*g_host_move_atoms)(void) = NULL;
#ifdef TTM
static int g_ttm_ready, g_fdn;                    /* defined with the TTM part below */
static void lattice_to_host(void);
#endif
static int l2_on(void)
{
  if (g_level2 < 0) {
    const char *e = getenv("IMD_SHIM_LEVEL");
    g_level2 = (e && atoi(e) == 2) ? 1 : 0;
    if (g_level2 && cpu_dim.x * cpu_dim.y * cpu_dim.z != 1) error("imd_shim: IMD_SHIM_LEVEL=2 serves one rank (several ranks: level 1, or the library's own multi-rank loop imdgpu_dist_step)");
#ifndef IMD_SHIM_WRAP_WRITERS
    /* Without the writer wrappers (below) the host cells are refreshed on checkpt_int / eng_int / ttm_int / force_int steps and at
     * steps_max only: a run that ends on maxwalltime or a stop file, a `write` file, or any of the other periodic writers would
     * put the atoms of the LAST refresh into its file (imd_main_3d.c:715-722,838-860,1964-2007) -- refused rather than wrong */
    if (g_level2 && (maxwalltime > 0 || stop_int > 0 || watch_int > 0 || dist_int > 0 || pic_int > 0))
      error("imd_shim: IMD_SHIM_LEVEL=2 with maxwalltime / stop_int / watch_int / dist_int / pic_int needs the shim built with "
            "-DIMD_SHIM_WRAP_WRITERS and linked with -Wl,--wrap=write_config,--wrap=write_config_select,--wrap=write_distrib,--wrap=write_pictures (INTEGRATION.md)");
#endif
  }
  return g_level2;
}
static int l2_listed(const char *var, int s)
{
  const char *e = getenv(var);
  if (e == NULL) return 0;
  while (*e) {
    char *end; long v = strtol(e, &end, 10);
    if (end == e) break;
    if (v == s) return 1;
    e = end; while (*e == ',' || *e == ' ') e++;
  }
  return 0;
}
/* device -> host cells: the device's arrangement replaces the host's */
static void l2_sync_down(void)
{
  int nl = 0, ng = 0, k, nc = 0;
  double sums[8], ek = 0;
  gpu_check(imdgpu_num_atoms(G, &nl, &ng));
  ensure_host(nl);
  gpu_check(imdgpu_download_atoms(G, g_cn, g_ort, g_imp, g_kraft, g_epot, g_rho, g_dF, g_nn, g_num, g_sorte, NULL));
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    int c = g_cn[SHIM_CELLIDX(p)];
    if (c > p->n_max) alloc_cell(p, c + 8);
    p->n = c;
  }
  (void) cell_offsets();
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    int i, m = g_off[k];
    for (i = 0; i < p->n; i++, m++) {
      ORT(p, i, X) = g_ort[3 * m]; ORT(p, i, Y) = g_ort[3 * m + 1]; ORT(p, i, Z) = g_ort[3 * m + 2];
      IMPULS(p, i, X) = g_imp[3 * m]; IMPULS(p, i, Y) = g_imp[3 * m + 1]; IMPULS(p, i, Z) = g_imp[3 * m + 2];
      KRAFT(p, i, X) = g_kraft[3 * m]; KRAFT(p, i, Y) = g_kraft[3 * m + 1]; KRAFT(p, i, Z) = g_kraft[3 * m + 2];
      POTENG(p, i) = g_epot[m]; NUMMER(p, i) = g_num[m]; SORTE(p, i) = g_sorte[m]; VSORTE(p, i) = g_sorte[m];
      MASSE(p, i) = masses[g_sorte[m]];
#ifdef EAM2
      EAM_RHO(p, i) = g_rho[m]; EAM_DF(p, i) = g_dF[m];
#endif
#ifdef TTM
      NUMNEIGHS(p, i) = g_nn[m];
#endif
    }
  }
  gpu_check(imdgpu_get_sums(G, sums, &ek, &nc));
  nbl_count = nc;
  g_synced_moves = g_moves;
  /* the list globals tl/tb/cl_off (globals.h:184-188) of the arrangement just handed over, for modules that read them */
  {
    int ntb = 0;
    const int with_list = getenv("IMD_SHIM_NO_NBL_EXPORT") == NULL;
    if (with_list) gpu_check(imdgpu_export_nblist(G, NULL, NULL, NULL, &ntb));
    tl = (int *) realloc(tl, (nl + 1) * sizeof(int));
    tb = (int *) realloc(tb, (ntb + 1) * sizeof(int));
    cl_off = (int *) realloc(cl_off, nallcells * sizeof(int));
    if (!tl || !tb || !cl_off) error("imd_shim: cannot allocate neighbor table");
    if (with_list) { gpu_check(imdgpu_export_nblist(G, tl, tb, cl_off, &ntb)); nb_max = ntb; }
    else { memset(tl, 0, (nl + 1) * sizeof(int)); tb[0] = 0; memset(cl_off, 0, nallcells * sizeof(int)); }
  }
}
static void l2_move_atoms(void)
{
  long long valid = 1;
  gpu_check(imdgpu_md_step_end(G, steps));
  g_moves++;
  gpu_check(imdgpu_get_info(G, "have_valid_nbl", &valid));
  have_valid_nbl = (int) valid;
  if ((eng_int > 0 && 0 == steps % eng_int) || steps == steps_max) {
    double ek = 0; int nc = 0;
    gpu_check(imdgpu_get_sums(G, NULL, &ek, &nc));
    tot_kin_energy = ek; nbl_count = nc;
  }
  if ((checkpt_int > 0 && 0 == steps % checkpt_int && steps > 0) || steps == steps_max) l2_sync_down();   /* write_config reads the cells */
#ifdef TTM
  if ((ttm_int > 0 && 0 == steps % ttm_int) || steps == steps_max) {                                      /* ttm_writeout reads l1 */
#ifndef TTM1D
    double mdt = 0.0;
    gpu_check(imdgpu_ttm_cfl_maxdt(G, steps, &mdt));
    max_dt_ttm = mdt;
#endif
    lattice_to_host();
  }
#endif
}
#ifdef TTM
static void ttm_gpu_init(void);
#endif
static void l2_calc_forces(int steps_)
{
  double sums[8];
  int n, is_short = 0;
  if (!g_dev_atoms) {                             /* first call: the host's arrangement goes to the device once */
    n = count_atoms();
    ensure_host(n);
    gather_cells(1);
    gpu_check(imdgpu_upload_cells(G, g_cn, n, g_ort, g_imp, g_mas, g_sorte, g_num));
    g_have_cells = 1; g_dev_atoms = 1;
    g_host_move_atoms = move_atoms;
    move_atoms = l2_move_atoms;                   /* the active integrator routine, globals.h:1777 */
#ifdef TTM
    if (!g_ttm_ready) ttm_gpu_init();
#endif
  }
  gpu_check(imdgpu_md_step_begin(G, steps_));
  have_valid_nbl = 1;
  if ((eng_int > 0 && 0 == steps_ % eng_int) || steps_ == steps_max) {
    gpu_check(imdgpu_get_sums(G, sums, NULL, NULL));
    tot_pot_energy = sums[0]; virial = sums[1];
    vir_xx = vir_yy = vir_zz = vir_xy = vir_yz = vir_zx = 0.0;
  }
  (void) is_short;
  nfc++;
  if (l2_listed("IMD_SHIM_SYNC_STEPS", steps_)
#ifdef FORCE
      || (force_int > 0 && 0 == steps_ % force_int)
#endif
     ) l2_sync_down();
}

#ifdef IMD_SHIM_WRAP_WRITERS
/* Level 2: EVERY writer of the reference that walks the host cells first gets the device's atoms (and, with the TTM, its
 * lattice) -- whatever decides that it runs: checkpt_int, the last step, a `write` file (check_write, imd_main_3d.c:838-843),
 * a stop file or maxwalltime (check_stop / check_walltime call write_config(-1, steps), :1964-2007), dist_int, pic_int, the
 * *_int of write_config_select's callers.  Link with -Wl,--wrap=write_config,--wrap=write_config_select,--wrap=write_distrib,
 * --wrap=write_pictures; nothing is copied when the cells are already those of the last integrator call.   */
static void l2_fresh(void)
{
  if (g_level2 != 1 || !g_dev_atoms || g_synced_moves == g_moves) return;
  l2_sync_down();
#ifdef TTM
  if (g_ttm_ready) lattice_to_host();
#endif
}
void __real_write_config(int fzhlr, int steps_);
void __wrap_write_config(int fzhlr, int steps_) { l2_fresh(); __real_write_config(fzhlr, steps_); }
void __real_write_config_select(int fzhlr, char *suffix, void (*write_atoms_fun)(FILE *), void (*write_header_fun)(FILE *));
void __wrap_write_config_select(int fzhlr, char *suffix, void (*write_atoms_fun)(FILE *), void (*write_header_fun)(FILE *))
{ l2_fresh(); __real_write_config_select(fzhlr, suffix, write_atoms_fun, write_header_fun); }
void __real_write_distrib(int steps_);
void __wrap_write_distrib(int steps_) { l2_fresh(); __real_write_distrib(steps_); }
void __real_write_pictures(int steps_);
void __wrap_write_pictures(int steps_) { l2_fresh(); __real_write_pictures(steps_); }
#endif
#endif /* NBL: level 2 */

#ifdef NBL
void make_nblist(void)
{
  int n = count_atoms(), k, i;
  gpu_init();
  ensure_host(n);
  gather_cells(1);
  gpu_check(imdgpu_upload_cells(G, g_cn, n, g_ort, g_imp, g_mas, g_sorte, g_num));
#ifdef MPI
  if (g_multi) {                                  /* the caller has just run send_cells(copy_cell, ...) */
    int changed = 0;
    gather_ghosts(0);
    gpu_check(imdgpu_upload_ghosts(G, g_gcn, g_nghost, g_gort, g_gsorte));
    gpu_check(imdgpu_autotile(G, &changed));      /* first list: tile shape from the real occupancies */
    if (changed) gpu_check(imdgpu_upload_ghosts(G, g_gcn, g_nghost, g_gort, g_gsorte));
  }
#endif
  gpu_check(imdgpu_make_nblist(G));
  g_have_cells = 1;
  /* this fork keeps tl/tb/cl_off in globals.h:184-188 for other modules (imd_nrb.c): publish the list in the
   * reference's own half-list format unless IMD_SHIM_NO_NBL_EXPORT is set */
  if (getenv("IMD_SHIM_NO_NBL_EXPORT") == NULL && !g_multi) {
    int ntb = 0;
    gpu_check(imdgpu_export_nblist(G, NULL, NULL, NULL, &ntb));
    tl = (int *) realloc(tl, (n + 1) * sizeof(int));
    tb = (int *) realloc(tb, (ntb + 1) * sizeof(int));
    cl_off = (int *) realloc(cl_off, nallcells * sizeof(int));
    if (!tl || !tb || !cl_off) error("imd_shim: cannot allocate neighbor table");
    gpu_check(imdgpu_export_nblist(G, tl, tb, cl_off, &ntb));
    nb_max = ntb;
  }
  if (g_multi) {                                 /* several ranks: the globals exist but hold an empty list */
    tl = (int *) realloc(tl, (n + 1) * sizeof(int)); tb = (int *) realloc(tb, sizeof(int));
    cl_off = (int *) realloc(cl_off, nallcells * sizeof(int));
    if (!tl || !tb || !cl_off) error("imd_shim: cannot allocate neighbor table");
    memset(tl, 0, (n + 1) * sizeof(int)); tb[0] = 0; memset(cl_off, 0, nallcells * sizeof(int));
  }
  for (k = 0; k < ncells; k++) {                 /* reference positions, imd_forces_nbl.c:143-155 */
    cell *p = cell_array + cnbrs[k].np;
    for (i = 0; i < p->n; i++) {
      NBL_POS(p, i, X) = ORT(p, i, X); NBL_POS(p, i, Y) = ORT(p, i, Y); NBL_POS(p, i, Z) = ORT(p, i, Z);
    }
  }
  have_valid_nbl = 1;
  nbl_count++;
}

void deallocate_nblist(void) { have_valid_nbl = 0; }

int estimate_nblist_size(void)
{
  long long e = 0; int m = 0;
  if (!G || !g_have_cells) return 100000;
  gpu_check(imdgpu_nblist_stats(G, &e, &m));
  return (int) (e / 2) + 1;
}

void check_nblist(void)
{
  /* same rule on the host copies the integrator just advanced (imd_forces_nbl.c:1062-1092) */
  real max1 = 0.0;
  int k, i;
  if (G && l2_on() && g_dev_atoms) return;        /* level 2: the device integrator has checked; have_valid_nbl is set by move_atoms */
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    for (i = 0; i < p->n; i++) {
      real dx = ORT(p, i, X) - NBL_POS(p, i, X), dy = ORT(p, i, Y) - NBL_POS(p, i, Y), dz = ORT(p, i, Z) - NBL_POS(p, i, Z);
      real r2 = dx * dx + dy * dy + dz * dz;
      if (r2 > max1) max1 = r2;
    }
  }
#ifdef MPI
  { real max2; MPI_Allreduce(&max1, &max2, 1, REAL, MPI_MAX, cpugrid); max1 = max2; }      /* :1087 */
#endif
  if (max1 > SQR(0.5 * nbl_margin)) have_valid_nbl = 0;
}

#endif /* NBL */

void calc_forces(int steps)
{
  double sums[8];
  int n, k, i, m = 0, is_short = 0;
  gpu_init();
#ifdef NBL
  if (l2_on()) { l2_calc_forces(steps); return; }
  if (0 == have_valid_nbl) {
    fix_cells();                                  /* the host's own re-binning, imd_forces_nbl.c:300-308 */
#ifdef MPI
    if (g_multi) {
      if (0 == nbl_count % BUFSTEP) setup_buffers();                       /* :304 */
      send_cells(copy_cell, pack_cell, unpack_cell);                        /* :311 */
      send_cells_west(copy_cell, pack_cell, unpack_cell);
    }
#endif
    make_nblist();
  } else {
#ifdef MPI
    if (g_multi) { send_cells(copy_cell, pack_cell, unpack_cell); send_cells_west(copy_cell, pack_cell, unpack_cell); }   /* :311 */
#endif
    n = count_atoms();
    ensure_host(n);
    gather_cells(0);
    gpu_check(imdgpu_update_positions(G, n, g_ort));
#ifdef MPI
    if (g_multi) { gather_ghosts(0); gpu_check(imdgpu_update_ghosts(G, g_nghost, g_gort, NULL)); }
#endif
  }
#else
  /* cell-pair variant: the main loop has re-binned (fix_cells, imd_main_3d.c:815); all pairs of this arrangement */
  n = count_atoms();
  ensure_host(n);
  gather_cells(1);
  gpu_check(imdgpu_upload_cells(G, g_cn, n, g_ort, g_imp, g_mas, g_sorte, g_num));
  gpu_check(imdgpu_make_nblist(G));
  g_have_cells = 1;
#endif
  n = count_atoms();
#if defined(NBL) && defined(MPI)
  if (g_multi) {
    double ek = 0; int nc = 0;
    gpu_check(imdgpu_forces_phase(G, 1));                                   /* pair + rho */
    gpu_check(imdgpu_forces_phase(G, 2));                                   /* embedding */
#ifdef EAM2
    gpu_check(imdgpu_download_atoms(G, NULL, NULL, NULL, NULL, NULL, NULL, g_dF, NULL, NULL, NULL, NULL));
    for (k = 0, m = 0; k < NCELLS; k++) { cell *p = CELLPTR(k); for (i = 0; i < p->n; i++, m++) EAM_DF(p, i) = g_dF[m]; }
    m = 0;
    send_cells(copy_dF, pack_dF, unpack_dF);                                /* :786 */
    send_cells_west(copy_dF, pack_dF, unpack_dF);
    gather_ghosts(1);
    gpu_check(imdgpu_update_ghosts(G, g_nghost, NULL, g_gdF));
#endif
    gpu_check(imdgpu_forces_phase(G, 3));                                   /* EAM forces + sums */
    gpu_check(imdgpu_get_sums(G, sums, &ek, &nc));
    { double loc[2], glob[2]; loc[0] = sums[0]; loc[1] = sums[1];
      MPI_Allreduce(loc, glob, 2, MPI_DOUBLE, MPI_SUM, cpugrid); sums[0] = glob[0]; sums[1] = glob[1]; }      /* :1007 */
  } else
#endif
  gpu_check(imdgpu_calc_forces(G, steps, sums, &is_short));
  gpu_check(imdgpu_download_atoms(G, NULL, NULL, NULL, g_kraft, g_epot, g_rho, g_dF, g_nn, NULL, NULL, NULL));
  (void) cell_offsets();
#ifdef _OPENMP
#pragma omp parallel for schedule(static) private(i, m)
#endif
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    for (i = 0, m = g_off[k]; i < p->n; i++, m++) {
      KRAFT(p, i, X) = g_kraft[3 * m]; KRAFT(p, i, Y) = g_kraft[3 * m + 1]; KRAFT(p, i, Z) = g_kraft[3 * m + 2];
      POTENG(p, i) = g_epot[m];
#ifdef EAM2
      EAM_RHO(p, i) = g_rho[m]; EAM_DF(p, i) = g_dF[m];
#endif
#ifdef TTM
      NUMNEIGHS(p, i) = g_nn[m];
#endif
    }
  }
  tot_pot_energy = sums[0]; virial = sums[1];
  vir_xx = vir_yy = vir_zz = vir_xy = vir_yz = vir_zx = 0.0;
  nfc++;
  if (is_short) fprintf(stderr, "Short distance, step %d!\n", steps);
}

#if !defined(NBL) && defined(IMD_SHIM_CELLPAIR_FUNCTIONS)
/* ---- the cell-pair kernels one by one, under the reference's own names and signatures ----------------------------------
 *     void do_forces(cell *p, cell *q, vektor pbc, real *Epot, real *Virial, real *Vir_xx, ... real *Vir_xy);
 *                                                              (prototypes.h:412, imd_forces.c:33)
 *     void do_forces_eam2(cell *p, cell *q, vektor pbc, real *Virial, real *Vir_xx, ... real *Vir_xy);
 *                                                              (prototypes.h:417, imd_forces_eam2.c:236)
 *     void do_embedding_energy(void);                          (prototypes.h:418, imd_forces_eam2.c:202)
 * for hosts that keep the cell-pair loop of imd_main_risc_3d.c:31-182 (its calc_forces) and call these per pair of cells:
 * link the shim INSTEAD of imd_forces.o / imd_forces_eam2.o (and compile it with -Dcalc_forces=<unused name>, the host's
 * own calc_forces stays).  Each call = the two cells' arrays to the device, one small launch, results added on the host:
 * the contract of the originals at the price of two PCIe round trips per cell pair -- the whole-step calc_forces above is
 * the fast path. */
static int *cp_ints(const shortint *a, int n, int **buf, int *cap)
{
  int i;
  if (n > *cap) { *cap = n + 64; *buf = (int *) realloc(*buf, *cap * sizeof(int)); if (!*buf) error("imd_shim: out of memory"); }
  for (i = 0; i < n; i++) (*buf)[i] = a ? a[i] : 0;
  return *buf;
}
void do_forces(cell *p, cell *q, vektor pbc, real *Epot, real *Virial, real *Vir_xx, real *Vir_yy, real *Vir_zz,
               real *Vir_yz, real *Vir_zx, real *Vir_xy)
{
  static int *tp = NULL, *tq = NULL, ctp = 0, ctq = 0, *np_ = NULL, *nq_ = NULL, cnp = 0, cnq = 0;
  double sh[3]; int same, is_short = 0, i;
  gpu_init();
  sh[0] = pbc.x; sh[1] = pbc.y; sh[2] = pbc.z;
  same = (p == q) && pbc.x == 0 && pbc.y == 0 && pbc.z == 0;
  cp_ints(p->sorte, p->n, &tp, &ctp); cp_ints(q->sorte, q->n, &tq, &ctq);
#ifdef TTM
  cp_ints(NULL, p->n, &np_, &cnp); cp_ints(NULL, q->n, &nq_, &cnq);
#endif
  gpu_check(imdgpu_do_forces(G, p->n, p->ort, tp, q->n, q->ort, tq, sh, same, p->kraft, q->kraft, p->pot_eng, q->pot_eng,
#ifdef EAM2
                             p->eam_rho, q->eam_rho,
#else
                             NULL, NULL,
#endif
#ifdef TTM
                             np_, same ? np_ : nq_,
#else
                             NULL, NULL,
#endif
                             Epot, Virial, &is_short));
#ifdef TTM
  for (i = 0; i < p->n; i++) NUMNEIGHS(p, i) += np_[i];
  if (!same) for (i = 0; i < q->n; i++) NUMNEIGHS(q, i) += nq_[i];
#endif
  if (is_short) printf("\n Short distance!\n");
  (void) Vir_xx; (void) Vir_yy; (void) Vir_zz; (void) Vir_yz; (void) Vir_zx; (void) Vir_xy; (void) i; (void) np_; (void) nq_; (void) cnp; (void) cnq;
}
#ifdef EAM2
void do_forces_eam2(cell *p, cell *q, vektor pbc, real *Virial, real *Vir_xx, real *Vir_yy, real *Vir_zz,
                    real *Vir_yz, real *Vir_zx, real *Vir_xy)
{
  static int *tp = NULL, *tq = NULL, ctp = 0, ctq = 0;
  double sh[3]; int same, is_short = 0;
  gpu_init();
  sh[0] = pbc.x; sh[1] = pbc.y; sh[2] = pbc.z;
  same = (p == q) && pbc.x == 0 && pbc.y == 0 && pbc.z == 0;
  cp_ints(p->sorte, p->n, &tp, &ctp); cp_ints(q->sorte, q->n, &tq, &ctq);
  gpu_check(imdgpu_do_forces_eam2(G, p->n, p->ort, tp, p->eam_dF, q->n, q->ort, tq, q->eam_dF, sh, same, p->kraft, q->kraft, Virial, &is_short));
  if (is_short) printf("\n Short distance!\n");
  (void) Vir_xx; (void) Vir_yy; (void) Vir_zz; (void) Vir_yz; (void) Vir_zx; (void) Vir_xy;
}
void do_embedding_energy(void)
{
  static int *tp = NULL, ctp = 0;
  int k;
  gpu_init();
  for (k = 0; k < NCELLS; k++) {
    cell *p = CELLPTR(k);
    if (p->n == 0) continue;
    cp_ints(p->sorte, p->n, &tp, &ctp);
    gpu_check(imdgpu_do_embedding_energy(G, p->n, p->eam_rho, tp, p->pot_eng, p->eam_dF, &tot_pot_energy));
  }
}
#endif
#endif /* IMD_SHIM_CELLPAIR_FUNCTIONS */

#if defined(TTM) && defined(TTM1D)

/* ---- imd_ttm.c's own 1D model (-DTTM1D, what the reference's Makefile builds; with imd_tmm.c as laser source) ------------
 * Build: the reference's sources with -Dcalc_ttm=calc_ttm_host on imd_ttm.c and its ttm_fill_ghost_layers made a weak symbol
 * in the object file (objcopy; the file compiled -fPIC so that init_ttm's own calls bind at link time) -- no source change --,
 * imd_forces_nbl.c left out; this file provides calc_forces, calc_ttm and ttm_fill_ghost_layers.  The reference's 1D program cannot even initialise on one rank (init_ttm ends with
 * ttm_fill_ghost_layers, whose one-rank case receives from a rank that does not exist, imd_ttm.c:1193-1199,3264-3360); with the
 * lattice on the device one rank is all there is to it (the row is ONE array), so the ghost fill of the host
 * lattice has nothing to do.  init_ttm / tmm_init stay the reference's: they read the tables, which go to the library from the
 * reference's own memory.  IMD_SHIM_REF_RANKS=n reproduces what a run cut into n ranks does differently (the virtual lattice
 * as +x neighbour of every cell of its rank, imd_ttm.c:1535-1540; who feels the electron-pressure force, :596-620). */
static int     g_ttm_ready = 0, g_fdn = 0;
static int    *g_fi[5];                               /* natoms, natoms_old, natomsglobal, fluxfromleft, fluxfromright */
static double *g_fd[22];

static void eos_grid(const nn_interp *nn, const char *what, int *nx, int *ny, double **z)
{
  int n = nn->npoints, js = 1, q;
  while (js < n && nn->points[js].x == nn->points[0].x) js++;
  if (n <= 0 || n % js != 0) error_str("imd_shim: EOS table %s is not a regular grid", (char *) what);
  *ny = js; *nx = n / js;
  *z = (double *) malloc(n * sizeof(double));
  if (NULL == *z) error("imd_shim: cannot allocate EOS grid");
  for (q = 0; q < n; q++) {
    if (nn->points[q].x != nn->points[(q / js) * js].x || nn->points[q].y != nn->points[q % js].y)
      error_str("imd_shim: EOS table %s is not a regular grid", (char *) what);
    (*z)[q] = nn->points[q].z;
  }
}

void ttm_fill_ghost_layers(void) { }

/* the device row -> l1/l2 (index i = global cell + 1), the global slab arrays move_atoms_ttm reads (xiarr_global, vcom*global:
 * imd_integrate.c:651-656,737-739), the bookkeeping scalars, and what the reference prints per step */
static void lattice_to_host(void)
{
  imdgpu_ttm1d_fields F;
  const int nx = global_fd_dim.x;
  int i;
  double sc[4];
  memset(&F, 0, sizeof(F));
  F.natoms = g_fi[0]; F.natoms_old = g_fi[1]; F.natomsglobal = g_fi[2]; F.fluxfromleft = g_fi[3]; F.fluxfromright = g_fi[4];
  F.temp = g_fd[0]; F.temp2 = g_fd[1]; F.U = g_fd[2]; F.U2 = g_fd[3]; F.xi = g_fd[4]; F.md_temp = g_fd[5]; F.source = g_fd[6];
  F.dens = g_fd[7]; F.fd_k = g_fd[8]; F.fd_g = g_fd[9]; F.Z = g_fd[10]; F.Ce = g_fd[11]; F.ne = g_fd[12];
  F.vcomglobal = g_fd[13]; F.xiarr_global = g_fd[14];
  F.vlat_temp = g_fd[15]; F.vlat_md_temp = g_fd[16]; F.vlat_fd_k = g_fd[17]; F.vlat_fd_g = g_fd[18]; F.vlat_Ce = g_fd[19];
  F.scalars = sc;
  gpu_check(imdgpu_ttm1d_download(G, &F));
  for (i = 0; i < nx + 2; i++) {
    ttm_Element *e = &l1[i], *e2 = &l2[i];
    e->natoms = e2->natoms = F.natoms[i]; e->natoms_old = e2->natoms_old = F.natoms_old[i];
    e->temp = F.temp[i]; e2->temp = F.temp2[i]; e->U = F.U[i]; e2->U = F.U2[i];
    e->xi = e2->xi = F.xi[i]; e->md_temp = e2->md_temp = F.md_temp[i]; e->source = e2->source = F.source[i];
    e->dens = e2->dens = F.dens[i]; e->fd_k = F.fd_k[i]; e->fd_g = F.fd_g[i]; e->Z = F.Z[i]; e->Ce = F.Ce[i]; e->ne = F.ne[i];
  }
  for (i = 0; i < nx; i++) {
    natomsglobal[i] = F.natomsglobal[i]; fluxfromleftglobal[i] = F.fluxfromleft[i]; fluxfromrightglobal[i] = F.fluxfromright[i];
    vcomxglobal[i] = F.vcomglobal[i]; vcomyglobal[i] = F.vcomglobal[nx + i]; vcomzglobal[i] = F.vcomglobal[2 * nx + i];
    xiarr_global[i] = F.xiarr_global[i];
    l1[i + 1].vcomx = l2[i + 1].vcomx = vcomxglobal[i]; l1[i + 1].vcomy = l2[i + 1].vcomy = vcomyglobal[i];
    l1[i + 1].vcomz = l2[i + 1].vcomz = vcomzglobal[i];
  }
  for (i = 0; i < vlatdim; i++) {
    vlattice1[i].temp = F.vlat_temp[i]; vlattice1[i].md_temp = F.vlat_md_temp[i]; vlattice1[i].fd_k = F.vlat_fd_k[i];
    vlattice1[i].fd_g = F.vlat_fd_g[i]; vlattice1[i].Ce = F.vlat_Ce[i];
  }
  last_active_cell_global = (int) sc[0]; cur_vlattice_proc = (int) sc[1]; diff_substeps = (int) sc[2]; max_dt_ttm = sc[3];
#ifdef ELECPRESS
  gpu_check(imdgpu_ttm1d_epress_get(G, epress_global, epress_deriv));
#endif
#ifdef TMM
  if (I0 > 0) {
    double ts[8];
    int too_short = 0;
    gpu_check(imdgpu_ttm1d_tmm_get(G, ts, NULL, NULL, tmm_Qabs, NULL, &too_short));
    if (too_short) error("Sample is too short for TMM.Consider reducing tmm_threshold.");       /* imd_tmm.c:362 */
    tmm_time = ts[0]; laser_active = ts[1] != 0.0; Eabs_local = ts[2]; I_t = ts[3]; tmm_refl = ts[4]; tmm_trans = ts[5]; tmm_abs = ts[6];
  }
#endif
}

static void ttm_gpu_init(void)
{
  imdgpu_ttm1d_params par;
  double *zc, *ze;
  const int nx = global_fd_dim.x;
  int k;
  const char *rr = getenv("IMD_SHIM_REF_RANKS");
  if (cpu_dim.x * cpu_dim.y * cpu_dim.z != 1) error("imd_shim: the 1D model on the device serves one rank (it IS the whole row)");
  memset(&par, 0, sizeof(par));
  par.nx = nx; par.fd_min_atoms = fd_min_atoms; par.fd_n_timesteps = fd_n_timesteps; par.vlatdim = vlatdim; par.vlatbuffer = vlatbuffer;
  par.ref_ranks = rr ? atoi(rr) : 1;
  par.adv = getenv("IMD_SHIM_TTM_NOADV") ? 0 : 1;          /* calc_ttm calls do_ADV(1.0) (ADVMODE 1, imd_ttm.c:96-98) */
  par.vlatdens = vlatdens; par.atomic_weight = atomic_weight; par.atomic_charge = atomic_charge;
  eos_grid(&intp_cve_from_r_te, "EOS_cve_from_r_te", &par.cve_nx, &par.cve_ny, &zc);
  eos_grid(&intp_ee_from_r_tesqrt, "EOS_ee_from_r_tesqrt", &par.ee_nx, &par.ee_ny, &ze);
  par.cve_xmin = intp_cve_from_r_te.xmin; par.cve_xmax = intp_cve_from_r_te.xmax;
  par.cve_ymin = intp_cve_from_r_te.ymin; par.cve_ymax = intp_cve_from_r_te.ymax; par.cve_z = zc;
  par.ee_xmin = intp_ee_from_r_tesqrt.xmin; par.ee_xmax = intp_ee_from_r_tesqrt.xmax;
  par.ee_ymin = intp_ee_from_r_tesqrt.ymin; par.ee_ymax = intp_ee_from_r_tesqrt.ymax; par.ee_z = ze;
  gpu_check(imdgpu_ttm1d_init(G, &par));
  free(zc); free(ze);
#ifdef TMM
  if (I0 > 0) {                                            /* tmm_init has read the tables (imd_tmm.c:117-118) */
    imdgpu_tmm_params tp;
    double *eb = (double *) malloc(3 * (size_t) eps_bb_rows * sizeof(double)), *k12 = (double *) malloc(2 * (size_t) K12_rows * sizeof(double)), *pz;
    int i;
    memset(&tp, 0, sizeof(tp));
    for (i = 0; i < eps_bb_rows; i++) { eb[3 * i] = eps_bb_data[i][0]; eb[3 * i + 1] = eps_bb_data[i][1]; eb[3 * i + 2] = eps_bb_data[i][2]; }
    for (i = 0; i < K12_rows; i++) { k12[2 * i] = K12[i][0]; k12[2 * i + 1] = K12[i][1]; }
    tp.I0 = I0; tp.lambda = lambda; tp.laser_sigma_t = laser_sigma_t; tp.laser_t_0 = laser_t_0; tp.laser_sigma_t1 = laser_sigma_t1;
    tp.laser_t_1 = laser_t_1; tp.tmm_threshold = tmm_absorption_threshold; tp.tmm_laser_threshold = tmm_laser_threshold;
    tp.eps_bb_rows = eps_bb_rows; tp.eps_bb = eb; tp.K12_rows = K12_rows; tp.K12 = k12;
    eos_grid(&intp_phase_from_r_ti, "EOS_phase_from_r_ti", &tp.phase.nx, &tp.phase.ny, &pz);
    tp.phase.xmin = intp_phase_from_r_ti.xmin; tp.phase.xmax = intp_phase_from_r_ti.xmax;
    tp.phase.ymin = intp_phase_from_r_ti.ymin; tp.phase.ymax = intp_phase_from_r_ti.ymax; tp.phase.z = pz;
    gpu_check(imdgpu_ttm1d_tmm_init(G, &tp));
    gpu_check(imdgpu_ttm1d_tmm_set_time(G, tmm_time));     /* a restart sets it in init_ttm (:1184) */
    free(eb); free(k12); free(pz);
  }
#endif
#ifdef ELECPRESS
  {
    double *pz;
    int pnx, pny;
    eos_grid(&intp_pe_from_r_te, "EOS_pe_from_r_te", &pnx, &pny, &pz);
    gpu_check(imdgpu_ttm1d_set_epress(G, pnx, pny, intp_pe_from_r_te.xmin, intp_pe_from_r_te.xmax, intp_pe_from_r_te.ymin,
                                      intp_pe_from_r_te.ymax, pz));
    free(pz);
  }
#endif
  g_fdn = nx + 2;
  for (k = 0; k < 2; k++) g_fi[k] = (int *) malloc((nx + 2) * sizeof(int));
  for (k = 2; k < 5; k++) g_fi[k] = (int *) malloc(nx * sizeof(int));
  for (k = 0; k < 22; k++) g_fd[k] = (double *) malloc((k == 13 ? 3 : 1) * (size_t) (nx + 2 + vlatdim) * sizeof(double));
  g_ttm_ready = 1;
}

void calc_ttm(void)
{
  int n;
  gpu_init();
  if (!g_have_cells) error("imd_shim: calc_ttm before calc_forces");
#ifdef NBL
  if (l2_on()) {                                   /* level 2: forked beside the force passes by calc_forces */
    if (l2_listed("IMD_SHIM_SYNC_STEPS", steps)) lattice_to_host();
    return;
  }
#endif
  if (!g_ttm_ready) ttm_gpu_init();
  n = count_atoms();
  ensure_host(n);
  gather_cells(1);
  gpu_check(imdgpu_update_momenta(G, n, g_imp));    /* positions are the ones of this step's calc_forces; the momenta the host integrator's */
  gpu_check(imdgpu_ttm1d_calc(G, steps));
  lattice_to_host();                                /* l1/l2 and the global arrays move_atoms_ttm reads (xiarr_global, vcom*global, epress_deriv) */
}

#elif defined(TTM)

/* ---- calc_ttm on the device ------------------------------------------------------------------------------- */
static int     g_ttm_ready = 0, g_fdn = 0;
static int    *g_fi[2];                               /* natoms, natoms_old */
static double *g_fd[15];                              /* temp temp2 xi md_temp source vcom(3n) dens fd_k fd_g Z Ce ne U U2 */

/* the reference keeps an EOS table as the point list nn_read_table read (imd_interpol.c:1174-1232: x outer, y inner);
 * the library takes the same values on their regular grid */
static void eos_grid(const nn_interp *nn, const char *what, int *nx, int *ny, double **z)
{
  int n = nn->npoints, js = 1, q;
  while (js < n && nn->points[js].x == nn->points[0].x) js++;
  if (n <= 0 || n % js != 0) error_str("imd_shim: EOS table %s is not a regular grid", (char *) what);
  *ny = js; *nx = n / js;
  *z = (double *) malloc(n * sizeof(double));
  if (NULL == *z) error("imd_shim: cannot allocate EOS grid");
  for (q = 0; q < n; q++) {
    if (nn->points[q].x != nn->points[(q / js) * js].x || nn->points[q].y != nn->points[q % js].y)
      error_str("imd_shim: EOS table %s is not a regular grid", (char *) what);
    (*z)[q] = nn->points[q].z;
  }
}

/* Which diagonal did nn-c's triangulation put into every cell of the table grid?  The reference's own interpolator tells:
 * at a cell centre lpi_interpolate_point (the call of EOS_cve_from_r_te / EOS_ee_from_r_te, imd_ttm_3D.c:2722,2759) returns
 * the mean over the two nodes of the diagonal that splits the cell.  The library then interpolates on the same triangles
 * (imdgpu_ttm_set_eos_triangulation) -- the EOS look-up of the device IS the reference's, not a bilinear approximation. */
static unsigned char *eos_triangulation(nn_interp *nn, int nx, int ny, const double *z)
{
  unsigned char *d = (unsigned char *) malloc((size_t) (nx - 1) * (ny - 1));
  int i, j;
  if (NULL == d) error("imd_shim: cannot allocate the EOS triangulation map");
  for (i = 0; i < nx - 1; i++)
    for (j = 0; j < ny - 1; j++) {
      const point *p00 = &nn->points[i * ny + j], *p11 = &nn->points[(i + 1) * ny + j + 1];
      const double z00 = z[i * ny + j], z01 = z[i * ny + j + 1], z10 = z[(i + 1) * ny + j], z11 = z[(i + 1) * ny + j + 1];
      const double da = 0.5 * (z10 + z01), db = 0.5 * (z00 + z11);
      point c;
      c.x = 0.5 * (p00->x + p11->x); c.y = 0.5 * (p00->y + p11->y); c.z = 0.0;
      lpi_interpolate_point(nn->interpolator, &c);
      d[i * (ny - 1) + j] = (c.z != c.z) ? 0 : (fabs(c.z - da) <= fabs(c.z - db) ? 1 : 2);     /* NaN (outside the hull): bilinear */
    }
  return d;
}

/* the host lattice l1/l2 -> the device lattice.  Used once when the device lattice is created: it starts from the HOST
 * lattice as init_ttm left it -- on a restart (imdrestart > 0) ttm_read has filled l1/l2 with the checkpointed electron
 * state (imd_ttm_3D.c:1284-1290, :2128-2330: natoms, temp, md_temp, U, xi, source, dens, vcom, fd_k, fd_g, Z, Ce); on a
 * fresh start the same upload carries init_ttm's zeros.  Later host-side edits of l1/l2 between steps are NOT seen by the
 * whole-step calc_ttm (level 1); the per-function entry points below (IMD_SHIM_TTM_FUNCTIONS) upload
 * before every call. */
static void lattice_to_device(int with_source)
{
  imdgpu_ttm_fields F;
  int i, j, kk, c = 0;
  for (i = 0; i < local_fd_dim.x; i++)
    for (j = 0; j < local_fd_dim.y; j++)
      for (kk = 0; kk < local_fd_dim.z; kk++, c++) {
        const ttm_Element *e = &l1[i][j][kk], *e2 = &l2[i][j][kk];
        g_fi[0][c] = e->natoms; g_fi[1][c] = e->natoms_old;
        g_fd[0][c] = e->temp; g_fd[1][c] = e2->temp; g_fd[2][c] = e->xi; g_fd[3][c] = e->md_temp; g_fd[4][c] = e->source;
        g_fd[5][3 * c] = e->vcomx; g_fd[5][3 * c + 1] = e->vcomy; g_fd[5][3 * c + 2] = e->vcomz;
        g_fd[6][c] = e->dens; g_fd[7][c] = e->fd_k; g_fd[8][c] = e->fd_g; g_fd[9][c] = e->Z; g_fd[10][c] = e->Ce;
        g_fd[11][c] = e->ne; g_fd[12][c] = e->U; g_fd[13][c] = e2->U;
      }
  memset(&F, 0, sizeof(F));
  F.natoms = g_fi[0]; F.natoms_old = g_fi[1];
  F.temp = g_fd[0]; F.temp2 = g_fd[1]; F.xi = g_fd[2]; F.md_temp = g_fd[3]; F.vcom = g_fd[5];
  if (with_source) F.source = g_fd[4];          /* a host that calls the pieces may have written l1.source itself */
  F.dens = g_fd[6]; F.fd_k = g_fd[7]; F.fd_g = g_fd[8]; F.Z = g_fd[9]; F.Ce = g_fd[10]; F.ne = g_fd[11];
  F.U = g_fd[12]; F.U2 = g_fd[13];
  gpu_check(imdgpu_ttm_upload(G, &F));
}
/* the device lattice -> l1/l2 (l1 = the current lattice, l2 = the other half of the ping-pong pairs) */
static void lattice_to_host(void)
{
  imdgpu_ttm_fields F;
  int i, j, k, c = 0;
  memset(&F, 0, sizeof(F));
  F.natoms = g_fi[0]; F.natoms_old = g_fi[1];
  F.temp = g_fd[0]; F.temp2 = g_fd[1]; F.xi = g_fd[2]; F.md_temp = g_fd[3]; F.source = g_fd[4]; F.vcom = g_fd[5];
  F.dens = g_fd[6]; F.fd_k = g_fd[7]; F.fd_g = g_fd[8]; F.Z = g_fd[9]; F.Ce = g_fd[10]; F.ne = g_fd[11];
  F.U = g_fd[12]; F.U2 = g_fd[13];
  gpu_check(imdgpu_ttm_download(G, &F));
  for (i = 0; i < local_fd_dim.x; i++)
    for (j = 0; j < local_fd_dim.y; j++)
      for (k = 0; k < local_fd_dim.z; k++, c++) {
        ttm_Element *e = &l1[i][j][k], *e2 = &l2[i][j][k];
        e->natoms = e2->natoms = F.natoms[c]; e->natoms_old = e2->natoms_old = F.natoms_old[c];
        e->temp = F.temp[c]; e2->temp = F.temp2[c];
        e->U = F.U[c]; e2->U = F.U2[c];
        e->xi = e2->xi = F.xi[c]; e->md_temp = e2->md_temp = F.md_temp[c]; e->source = e2->source = F.source[c];
        e->vcomx = e2->vcomx = F.vcom[3 * c]; e->vcomy = e2->vcomy = F.vcom[3 * c + 1]; e->vcomz = e2->vcomz = F.vcom[3 * c + 2];
        e->dens = e2->dens = F.dens[c];
        e->fd_k = F.fd_k[c]; e->fd_g = F.fd_g[c]; e->Z = F.Z[c]; e->Ce = F.Ce[c]; e->ne = F.ne[c];
      }
}

static void ttm_gpu_init(void)
{
  imdgpu_ttm_params par;
  double *zc, *ze;
  int dim[3], k;
  double h[3];
  memset(&par, 0, sizeof(par));
  par.fd_ext[0] = fd_ext.x; par.fd_ext[1] = fd_ext.y; par.fd_ext[2] = fd_ext.z;
  par.fd_min_atoms = fd_min_atoms;
  par.fd_n_timesteps = 100;                            /* calc_ttm hard-codes it, imd_ttm_3D.c:51-52 */
  par.atomic_weight = atomic_weight; par.atomic_charge = atomic_charge; par.omega_laser = omega_laser;
  eos_grid(&intp_cve_from_r_te, "EOS_cve_from_r_te", &par.cve_nx, &par.cve_ny, &zc);
  eos_grid(&intp_ee_from_r_tesqrt, "EOS_ee_from_r_tesqrt", &par.ee_nx, &par.ee_ny, &ze);
  par.cve_xmin = intp_cve_from_r_te.xmin; par.cve_xmax = intp_cve_from_r_te.xmax;
  par.cve_ymin = intp_cve_from_r_te.ymin; par.cve_ymax = intp_cve_from_r_te.ymax; par.cve_z = zc;
  par.ee_xmin = intp_ee_from_r_tesqrt.xmin; par.ee_xmax = intp_ee_from_r_tesqrt.xmax;
  par.ee_ymin = intp_ee_from_r_tesqrt.ymin; par.ee_ymax = intp_ee_from_r_tesqrt.ymax; par.ee_z = ze;
  if (getenv("IMD_SHIM_TTM_ADV")) {
    if (g_multi) error("imd_shim: IMD_SHIM_TTM_ADV on several ranks is not wired into the shim");
    gpu_check(imdgpu_set_option(G, "ttm_adv", 1));
  }
#ifdef MPI
  if (g_multi) gpu_check(imdgpu_set_option(G, "ttm_decomposed", 1));    /* one lattice block per rank, like l1/l2 */
#endif
  gpu_check(imdgpu_ttm_init(G, &par));
  if (getenv("IMD_SHIM_EOS_BILINEAR") == NULL) {       /* default: the reference's own triangulation of the two tables */
    unsigned char *dc = eos_triangulation(&intp_cve_from_r_te, par.cve_nx, par.cve_ny, zc);
    unsigned char *de = eos_triangulation(&intp_ee_from_r_tesqrt, par.ee_nx, par.ee_ny, ze);
    gpu_check(imdgpu_ttm_set_eos_triangulation(G, 0, dc));
    gpu_check(imdgpu_ttm_set_eos_triangulation(G, 1, de));
    free(dc); free(de);
  } else {                                             /* experiment: the plain bilinear rule (the library's default is its own restatement of the triangulation) */
    gpu_check(imdgpu_ttm_set_eos_triangulation(G, 0, NULL));
    gpu_check(imdgpu_ttm_set_eos_triangulation(G, 1, NULL));
  }
#ifdef MPI
  if (g_multi) {
    /* the ghost faces of {natoms, temp, fd_k} that ttm_fill_ghost_layers moves with MPI_Sendrecv after every sub-step
     * (imd_ttm_3D.c:3314-3504) go through the library's mailbox in host shared memory, written and read by the GPUs
     * themselves (one node): rank 0 creates the file, everybody maps it */
    size_t per = 0;
    char name[256];
    int fd, nbr[6], nranks = cpu_dim.x * cpu_dim.y * cpu_dim.z;
    void *base;
    gpu_check(imdgpu_ttm_mailbox_bytes(G, &per));
    memset(name, 0, sizeof(name));
    if (0 == myid) {
      sprintf(name, "/dev/shm/imdgpu_ttm_%d", (int) getpid());
      fd = open(name, O_CREAT | O_RDWR | O_TRUNC, 0600);
      if (fd < 0 || ftruncate(fd, (off_t) (per * nranks)) != 0) error("imd_shim: cannot create the TTM mailbox in /dev/shm");
      close(fd);
    }
    MPI_Bcast(name, sizeof(name), MPI_CHAR, 0, cpugrid);
    fd = open(name, O_RDWR);
    if (fd < 0) error("imd_shim: cannot open the TTM mailbox");
    base = mmap(NULL, per * nranks, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (MAP_FAILED == base) error("imd_shim: cannot map the TTM mailbox");
    /* east = -x, west = +x, north = -y, south = +y, up = -z, down = +z (imd_ttm_3D.c:680-687); -1 = open boundary */
    nbr[0] = nbeast; nbr[1] = nbwest; nbr[2] = nbnorth; nbr[3] = nbsouth; nbr[4] = nbup; nbr[5] = nbdown;
    if (pbc_dirs.x != 1) { if (my_coord.x == 0) nbr[0] = -1; if (my_coord.x == cpu_dim.x - 1) nbr[1] = -1; }
    if (pbc_dirs.y != 1) { if (my_coord.y == 0) nbr[2] = -1; if (my_coord.y == cpu_dim.y - 1) nbr[3] = -1; }
    if (pbc_dirs.z != 1) { if (my_coord.z == 0) nbr[4] = -1; if (my_coord.z == cpu_dim.z - 1) nbr[5] = -1; }
    gpu_check(imdgpu_ttm_mailbox_attach(G, base, per, myid, nranks, nbr));
    MPI_Barrier(cpugrid);
    if (0 == myid) unlink(name);
  }
#endif
#ifdef LASER
  if (laser_rescale_mode == 4) {                       /* laser_rescale_ttm (imd_laser.c:715-788) inside calc_ttm */
    imdgpu_laser_params lp;
    lp.dir[0] = laser_dir.x; lp.dir[1] = laser_dir.y; lp.dir[2] = laser_dir.z;
    lp.offset = laser_offset; lp.mu = laser_mu; lp.p_peak = laser_p_peak; lp.t_0 = laser_t_0;
    lp.sigma_t_squared = laser_sigma_t_squared;
    gpu_check(imdgpu_ttm_set_laser(G, &lp));
  }
#endif
  free(zc); free(ze);
  gpu_check(imdgpu_ttm_dims(G, dim, h));
  if (dim[0] != local_fd_dim.x || dim[1] != local_fd_dim.y || dim[2] != local_fd_dim.z)
    error("imd_shim: FD lattice of the library differs from local_fd_dim");
  g_fdn = dim[0] * dim[1] * dim[2];
  for (k = 0; k < 2; k++) g_fi[k] = (int *) malloc(g_fdn * sizeof(int));
  for (k = 0; k < 15; k++) g_fd[k] = (double *) malloc((k == 5 ? 3 : 1) * (size_t) g_fdn * sizeof(double));
  lattice_to_device(0);
  g_ttm_ready = 1;
}

void calc_ttm(void)
{
  int n;
  double mdt = 0.0;
  gpu_init();
  if (!g_have_cells) error("imd_shim: calc_ttm before calc_forces");
#ifdef NBL
  if (l2_on()) {                                   /* level 2: forked beside the force passes by calc_forces */
    fd_n_timesteps = 100; diff_substeps = fd_n_timesteps;
    if (l2_listed("IMD_SHIM_SYNC_STEPS", steps)) {  /* the host wants to look at l1/l2 after this calc_ttm */
      double mdt2 = 0.0;
      gpu_check(imdgpu_ttm_cfl_maxdt(G, steps, &mdt2));
      max_dt_ttm = mdt2;
      lattice_to_host();
    }
    return;
  }
#endif
  if (!g_ttm_ready) ttm_gpu_init();
  fd_n_timesteps = 100; diff_substeps = fd_n_timesteps;              /* imd_ttm_3D.c:51-52 */
  /* same arrangement as the calc_forces of this step; the momenta are the host integrator's */
  n = count_atoms();
  ensure_host(n);
  gather_cells(1);
  gpu_check(imdgpu_update_momenta(G, n, g_imp));
  if (g_multi) gpu_check(imdgpu_ttm_local_step(G, steps));      /* this rank's block; faces through the mailbox */
  else gpu_check(imdgpu_calc_ttm(G, steps));
  /* CFL_maxdt and the download below fail (-> error() on this rank, MPI_Abort for all) when a neighbour's lattice
   * faces never arrived in the mailbox: a stale-ghost lattice is never copied into l1/l2 */
  gpu_check(imdgpu_ttm_cfl_maxdt(G, steps, &mdt));
#ifdef MPI
  if (g_multi) { double m2; MPI_Allreduce(&mdt, &m2, 1, MPI_DOUBLE, MPI_MIN, cpugrid); mdt = m2; }   /* :2683-2687 */
#endif
  max_dt_ttm = mdt;
  lattice_to_host();
}

#ifdef IMD_SHIM_TTM_FUNCTIONS
/* ---- the lattice functions one by one, under the reference's own names and signatures -------------------------------
 *     void update_fd(void);              (prototypes.h:1021, imd_ttm_3D.c:233)
 *     void do_ADV(double tau);           (prototypes.h:143,  imd_ttm_3D.c:1321)   [option ttm_adv: IMD_SHIM_TTM_ADV=1]
 *     void do_cell_activation(void);     (prototypes.h:147,  imd_ttm_3D.c:1468)
 *     void do_FILLMESH(void);            (prototypes.h:141,  imd_ttm_3D.c:492)
 *     void ttm_fill_ghost_layers(void);  (prototypes.h:1018, imd_ttm_3D.c:3314)
 *     void do_DIFF(double tau);          (prototypes.h:145,  imd_ttm_3D.c:1603)
 *     void CFL_maxdt(void);              (prototypes.h:153,  imd_ttm_3D.c:2637)
 * for hosts that keep their own calc_ttm (or a variant of it) and call the pieces.  Each call is
 *   l1/l2 -> device, the matching imdgpu_ttm_* call, device -> l1/l2,
 * i.e. it honours whatever the host did to the lattice in between (level 1: two 10-MB copies per call at config 3's size;
 * the whole-step calc_ttm above is the fast path).  Compile imd_ttm_3D.c with -Dupdate_fd=update_fd_host ... (one -D per
 * name, no source change) so that these definitions are the ones that link; one rank only (on several ranks the faces go
 * through the mailbox inside the whole-step calc_ttm). */
static void fn_begin(const char *who)
{
  gpu_init();
  if (g_multi) error("imd_shim: the per-function TTM entry points serve one rank; use calc_ttm on several");
  if (!g_have_cells) error("imd_shim: TTM function called before calc_forces");
  if (!g_ttm_ready) ttm_gpu_init();
  lattice_to_device(1);
  (void) who;
}
void update_fd(void)
{
  int n;
  fn_begin("update_fd");
  n = count_atoms();
  ensure_host(n);
  gather_cells(1);
  gpu_check(imdgpu_update_momenta(G, n, g_imp));          /* update_fd reads IMPULS (:323-325,431-436) and NUMNEIGHS of the last calc_forces */
  gpu_check(imdgpu_ttm_update_fd(G, steps));
  lattice_to_host();
}
void do_ADV(double tau)
{
  fn_begin("do_ADV");
  gpu_check(imdgpu_ttm_adv(G, tau, steps));
  lattice_to_host();
}
void do_cell_activation(void) { fn_begin("do_cell_activation"); gpu_check(imdgpu_ttm_cell_activation(G)); lattice_to_host(); }
void do_FILLMESH(void) { fn_begin("do_FILLMESH"); gpu_check(imdgpu_ttm_fillmesh(G, steps)); lattice_to_host(); }
void ttm_fill_ghost_layers(void) { fn_begin("ttm_fill_ghost_layers"); gpu_check(imdgpu_ttm_fill_ghost_layers(G)); lattice_to_host(); }
void do_DIFF(double tau) { fn_begin("do_DIFF"); gpu_check(imdgpu_ttm_diff(G, tau)); lattice_to_host(); }
void CFL_maxdt(void)
{
  double mdt = 0.0;
  fn_begin("CFL_maxdt");
  gpu_check(imdgpu_ttm_cfl_maxdt(G, steps, &mdt));
  max_dt_ttm = mdt;
}
#endif /* IMD_SHIM_TTM_FUNCTIONS */
#endif /* TTM */
