/*
 * imdgpu.h -- C ABI of libimdgpu.so: MI355X (gfx950) implementation of the
 * force + TTM inner loop of IMD/lasMD (fmqeisfeld/IMD).
 *
 * The reference has no plugin interface: its hot path is entered through
 * link-time C symbols that work on global state.  Every
 * entry point below names the reference function it replaces; the shim in
 * imd_amd/host/imd_shim.c gives them the reference's own names and signatures
 * (calc_forces(int), make_nblist(void), ...) so that the library can be linked
 * in place of imd_forces_nbl.o / imd_ttm_3D.o.
 *
 * Conventions
 *   - plain C types only; every call returns 0 on success, non-zero on error,
 *     and imdgpu_last_error() returns the message (the reference calls error()
 *     -> exit; a library must not).
 *   - host arrays handed in/out are "cell ordered": atoms of the inner cells
 *     concatenated in the order of the reference's `cells` list (x outer, z
 *     inner; imd_geom_3d.c:898-900), which is also the numbering of the
 *     reference's `tl` (imd_forces_nbl.c:36-41).  Coordinates are xyz
 *     interleaved like cell.ort (types.h:189, makros.h:143).
 *   - all floating point data is fp64 (reference `real` = double, types.h:31).
 */
#ifndef IMDGPU_H
#define IMDGPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct imdgpu_ctx imdgpu_ctx;

/* Same members, same order as the reference's pot_table_t (types.h:537-549,
 * non-SPLINE build): a reference pot_table_t* may be passed as is. */
typedef struct {
  double *begin;      /* first argument value per column               */
  double *end;        /* last argument value per column                */
  double *step;
  double *invstep;
  int    *len;
  int     ncols;
  int     maxsteps;
  double *table;      /* table[k*ncols+col], (maxsteps+2)*ncols values */
} imdgpu_pot_table;

/* ---- life cycle ---------------------------------------------------------- */
int  imdgpu_create(imdgpu_ctx **ctx, int device);
void imdgpu_destroy(imdgpu_ctx *ctx);
const char *imdgpu_last_error(const imdgpu_ctx *ctx);
/* run on a caller-owned HIP stream (hipStream_t) instead of the context's own */
int  imdgpu_set_stream(imdgpu_ctx *ctx, void *hip_stream);
int  imdgpu_synchronize(imdgpu_ctx *ctx);
/* tuning knobs: "mode" = 1 LDS-tiled force kernels (default), 0 global-gather kernels;
 * "ttm_persist" = 1 (default) all diffusion sub-steps of an MD step in one persistent launch where that applies,
 * 0 the captured chain of one launch per sub-step (same results bit for bit), -1 back to the default (a persistent launch
 * that cannot run -- its workgroups not all resident in time, or an active region beyond its 1280 blocks -- touches nothing
 * and the step is issued again through the chain: "ttm_persist_redone" counts those; setting the option also clears a
 * reported hand-off time-out); "ttm_persist_miss" = MD step whose persistent launch is made to step aside (tests);
 * "halo_overlap" = N > 0 (several ranks; default 0): pass 1 of the next step is enqueued as two launches from a tile queue -- the
 * tiles with no cell of another rank in their halo on all but N CUs while the position halo [send_cells(copy_cell),
 * imd_forces_nbl.c:311] travels on its own stream, the boundary tiles behind it; same results, measured slower on one GPU
 * (DESIGN 5), needs the third communicator id on the RCCL transport;
 * "p3_coef" = 1 (default) the EAM force pass (do_forces_eam2, imd_forces_nbl.c:789-985) reads DERIV_FUNC2's gradient
 * (potaccess.h:591-621) as one coefficient pair per table row where that applies (one atom type, pair_pot and rho_h_tab on
 * one r^2 grid, nothing with LDS tiles of its own beside it), 0 always as three table rows; the two agree to rounding;
 * "tile_rec" = 1 (default) several atom types with pair_pot and rho_h_tab run the record kernels (list rows segmented by the
 * neighbour's type; PAIR_INT2 / VAL_FUNC2 / DERIV_FUNC2 with the per-column begin / end / step of pot_table_t, potaccess.h:323-354,
 * 465-495,591-621), 0 the older general kernels (tables and a type per slot in the LDS when they fit, global gather when not);
 * "tile_rec_lds" = -1 (default) the library decides which tables the record kernels keep in the LDS (what fits beside a tile), else
 * a mask: 1 pair_pot and 2 rho_h_tab in pass 1, 4 rho_h_tab in the EAM force pass; the rest are read in place through L2 as
 * coefficient records; tables that cannot fit are dropped from the mask (get_info "rec_lds" says what runs); same results */
int  imdgpu_set_option(imdgpu_ctx *ctx, const char *name, int value);
/* introspection: "list_mode" (kernel family of the current list), "tile_cap", "slot_cap", "W", "ntiles", ...,
 * "fused_embedding" (1: the embedding pass runs inside the pass-1 kernel of the current list),
 * "ttm_flux", "ttm_adv", "ttm_block_width" (doubles per FD cell in imdgpu_ttm_local_block's records),
 * "ttm_persist_mode" (0 launch chain, 1 persistent kernel on trial with the chain behind it, 2 persistent kernel alone,
 * -1 not decided yet), "ttm_persist_runs" (persistent launches in a row that came through), "ttm_persist_redone" (MD steps
 * re-issued through the launch chain), "p3_standin_passes" (EAM force passes in which the three-row kernel stood in for the
 * coefficient form), "kernel_rev",
 * "p3_coef_rows" / "p3_coef_klo" (rows of the gradient-coefficient table the EAM force pass holds in LDS and the first
 * of them; 0 / -1: the three-row form), "pass1_form" (2 one type on a shared grid, 1 one type with pair_pot and rho_h_tab on
 * their own grids, 3 several types: record kernels, 0 the general kernels), "rec_lds" (see option "tile_rec_lds") */
int  imdgpu_get_info(imdgpu_ctx *ctx, const char *name, long long *value);

/* ---- set-up -------------------------------------------------------------- */
/* Geometry of one rank: box_x/y/z rows (globals.h:613), pbc_dirs
 * (globals.h:221), local cell array incl. ghost layer `cell_dim`
 * (imd_geom_3d.c:253-255), global_cell_dim, cpu_dim, my_coord
 * (imd_geom_mpi_3d.c:32-90). */
int imdgpu_set_geometry(imdgpu_ctx *ctx, const double box[9], const int pbc_dirs[3],
                        const int cell_dim[3], const int global_cell_dim[3],
                        const int cpu_dim[3], const int my_coord[3]);
/* cellsz = (r_cut+nbl_margin)^2 as after init_cells (imd_geom_3d.c:127-131) */
int imdgpu_set_params(imdgpu_ctx *ctx, int ntypes, double cellsz, double nbl_margin,
                      double timestep, int count_ttm_neighbours);
/* pair_pot / rho_h_tab / embed_pot (globals.h:701,1414-1415).  pair==NULL:
 * build without PAIR; rho==embed==NULL: build without EAM2. */
int imdgpu_upload_tables(imdgpu_ctx *ctx, const imdgpu_pot_table *pair,
                         const imdgpu_pot_table *rho_h, const imdgpu_pot_table *embed);
/* per-type force restrictions (globals.h `restrictions`, imd_integrate.c:209-213) */
int imdgpu_set_restrictions(imdgpu_ctx *ctx, const double *restr /* ntypes*3 */);

/* ---- atoms --------------------------------------------------------------- */
/* Take over the host's cell arrangement as it stands (no re-binning):
 * cell_n[nallcells] in linear cell order (ghost entries ignored), per-atom
 * arrays cell ordered.  Invalidates the neighbour list.  Replaces the
 * AoS-of-arrays walk over cell_array (types.h:188-349). */
int imdgpu_upload_cells(imdgpu_ctx *ctx, const int *cell_n, int natoms,
                        const double *ort, const double *impuls, const double *masse,
                        const int *sorte, const int *nummer);
/* Unsorted atoms (e.g. a generator): binned on the device by imdgpu_fix_cells. */
int imdgpu_upload_atoms(imdgpu_ctx *ctx, int natoms, const double *ort,
                        const double *impuls, const double *masse,
                        const int *sorte, const int *nummer);
/* new positions in the current arrangement (cell ordered), list stays valid: a host integrator between rebuilds */
int imdgpu_update_positions(imdgpu_ctx *ctx, int natoms, const double *ort);
/* new momenta in the current arrangement (cell ordered): what a host integrator hands over before imdgpu_calc_ttm
 * (update_fd reads IMPULS, imd_ttm_3D.c:323-325,431-436) */
int imdgpu_update_momenta(imdgpu_ctx *ctx, int natoms, const double *impuls);
int imdgpu_num_atoms(imdgpu_ctx *ctx, int *nlocal, int *nghost);
/* Any output pointer may be NULL.  cell_n: nallcells ints (inner cells only
 * are non-zero).  Arrays are cell ordered in the CURRENT device arrangement. */
int imdgpu_download_atoms(imdgpu_ctx *ctx, int *cell_n, double *ort, double *impuls,
                          double *kraft, double *pot_eng, double *eam_rho, double *eam_dF,
                          int *numneighs, int *nummer, int *sorte, double *nbl_pos);

/* ---- hot path (reference names in brackets) ----------------------------- */
/* [fix_cells, imd_fix_cells_3d.c:43-306 + do_boundaries, imd_main_3d.c:2018]
 * wrap into the box, re-bin (stable counting order), invalidate the list */
int imdgpu_fix_cells(imdgpu_ctx *ctx);
/* [send_cells(copy_cell,...), imd_comm_force_3d.c:222-396] refresh the ghost
 * layer from the periodic images (all 26 directions, i.e. the non-AR shell) */
int imdgpu_fill_ghosts(imdgpu_ctx *ctx);
/* [make_nblist, imd_forces_nbl.c:136-287] */
int imdgpu_make_nblist(imdgpu_ctx *ctx);
/* [check_nblist, imd_forces_nbl.c:1062-1092] *have_valid_nbl as the global */
int imdgpu_check_nblist(imdgpu_ctx *ctx, int *have_valid_nbl);
/* [calc_forces, imd_forces_nbl.c:295-1054].  sums[8] = tot_pot_energy, virial,
 * vir_xx, vir_yy, vir_zz, vir_xy, vir_yz, vir_zx (order of tmpvec1, :999-1006;
 * the vir_?? entries are 0 as in the non-P_AXIAL build).  is_short as :673. */
int imdgpu_calc_forces(imdgpu_ctx *ctx, int steps, double sums[8], int *is_short);
/* [move_atoms_nve, imd_integrate.c:35-561] */
int imdgpu_move_atoms_nve(imdgpu_ctx *ctx, double *tot_kin_energy);
/* one full MD step kept on the device: calc_forces, [calc_ttm], move_atoms,
 * check_nblist (main_loop body, imd_main_3d.c:406-812).  Results of the
 * reductions stay on the device until imdgpu_get_sums. */
int imdgpu_md_step(imdgpu_ctx *ctx, int steps);
/* The same step in two halves for a host that keeps its OWN main loop (level 2): begin = what calc_forces
 * [+ calc_ttm] do (imd_main_3d.c:398-534; list maintenance included, nothing waited for), end = move_atoms + check_nblist
 * (:564,812) and the step's one host round trip.  Between the two the host may do whatever its loop does there; device
 * results stay on the device until it asks for them (imdgpu_get_sums, imdgpu_download_atoms, imdgpu_ttm_download). */
int imdgpu_md_step_begin(imdgpu_ctx *ctx, int steps);
int imdgpu_md_step_end(imdgpu_ctx *ctx, int steps);
int imdgpu_get_sums(imdgpu_ctx *ctx, double sums[8], double *tot_kin_energy,
                    int *nbl_count);

/* The reference's CELL-PAIR kernels on host arrays, for hosts that keep the cell-pair loop of imd_main_risc_3d.c:31-182 and
 * call them per pair of cells.  Same contract as the originals: results are ADDED to the arrays passed in; every pair is
 * visited once and the partner gets the opposite force / the same energy half; pbc = the periodic shift of cell q
 * (the reference's `pbc` vector); same_cell != 0 means p == q with zero shift (j starts at i + 1; pass the same arrays for
 * p and q).  ort xyz interleaved, sorte may be NULL (all type 0), numneighs NULL unless the TTM neighbour count is wanted.
 *   [do_forces, imd_forces.c:33-326: pair_pot term (r2 <= end), host electron density (r2 < end), Epot, virial]
 *   [do_forces_eam2, imd_forces_eam2.c:236-369: EAM force 0.5 (dF_i rho'_j + dF_j rho'_i), virial]
 *   [do_embedding_energy, imd_forces_eam2.c:202-230: pot_eng += F(rho), eam_dF = 2 F'(rho), tot_pot_energy += sum]
 * A compatibility path (one small launch and two PCIe round trips per call); the list kernels are the fast path. */
int imdgpu_do_forces(imdgpu_ctx *ctx, int np, const double *ort_p, const int *sorte_p, int nq, const double *ort_q,
                     const int *sorte_q, const double pbc[3], int same_cell, double *kraft_p, double *kraft_q,
                     double *poteng_p, double *poteng_q, double *eam_rho_p, double *eam_rho_q, int *numneighs_p,
                     int *numneighs_q, double *epot, double *virial, int *is_short);
int imdgpu_do_forces_eam2(imdgpu_ctx *ctx, int np, const double *ort_p, const int *sorte_p, const double *eam_dF_p, int nq,
                          const double *ort_q, const int *sorte_q, const double *eam_dF_q, const double pbc[3], int same_cell,
                          double *kraft_p, double *kraft_q, double *virial, int *is_short);
int imdgpu_do_embedding_energy(imdgpu_ctx *ctx, int n, const double *eam_rho, const int *sorte, double *pot_eng, double *eam_dF,
                               double *tot_pot_energy);

/* The Verlet list in the REFERENCE's own format and numbering (half list,
 * tl/tb/cl_off, imd_forces_nbl.c:24-42), rebuilt from the current arrangement.
 * Pass tb==NULL to query sizes: *ntb receives the entry count.  tl has
 * nlocal+1 entries, cl_off nallcells entries. */
int imdgpu_export_nblist(imdgpu_ctx *ctx, int *tl, int *tb, int *cl_off, int *ntb);
/* full (symmetric) device list statistics: total entries and max per atom */
int imdgpu_nblist_stats(imdgpu_ctx *ctx, long long *entries, int *max_per_atom);

/* Page-locked host memory for the arrays a level-1 host passes to imdgpu_update_positions / _update_momenta /
 * _download_atoms every step: the reordering between the device's atom order and the host's cell order
 * happens on the device, the arrays cross PCIe whole; from pinned memory that runs at link speed. */
int imdgpu_host_alloc(void **p, size_t bytes);
int imdgpu_host_free(void *p);

/* ---- two-temperature model (3D FD lattice, imd_ttm_3D.c) ------------------- */
/* Parameters the reference reads from the param file / derives in init_ttm
 * (imd_ttm_3D.c:1001-1314): fd_ext, fd_min_atoms, fd_n_timesteps (used as
 * diff_substeps, :51-52), atomic_weight, atomic_charge, omega_laser; and the two
 * EOS tables of nn_read_table (imd_interpol.c:1174-1232) given on their regular
 * grid z[ix*ny+iy]: cve(rho [kg/m^3], T [K]) in J/(K kg) and
 * ee(rho, sqrt(T [K])) in J/kg.  Interpolation is bilinear on that grid. */
typedef struct {
  int    fd_ext[3];
  int    fd_min_atoms;
  int    fd_n_timesteps;
  double atomic_weight, atomic_charge, omega_laser;
  int    cve_nx, cve_ny;
  double cve_xmin, cve_xmax, cve_ymin, cve_ymax;
  const double *cve_z;
  int    ee_nx, ee_ny;
  double ee_xmin, ee_xmax, ee_ymin, ee_ymax;
  const double *ee_z;
} imdgpu_ttm_params;

/* All lattice arrays are SoA over the local FD lattice INCLUDING its ghost shell,
 * C order [i][j][k] like l1[i][j][k] (imd_ttm_3D.c:1071-1083). */
typedef struct {
  int    *natoms, *natoms_old;
  double *temp, *temp2;          /* l1.temp, l2.temp (do_DIFF ping-pong, :1819-1821) */
  double *xi, *md_temp, *source;
  double *vcom;                  /* 3 per cell: vcomx, vcomy, vcomz */
  double *dens, *fd_k, *fd_g, *Z, *Ce, *ne;
  double *U, *U2;
  int    *flux;                  /* 8 per cell: atoms that entered the cell since the last update_fd, by direction
                                    (ttm_Element.flux, imd_ttm_3D.c:335-370); counted with option "ttm_flux" */
} imdgpu_ttm_fields;

/* [init_ttm] needs geometry, params, pair_pot (neighvol from pair_pot.end[0], :1019) */
int imdgpu_ttm_init(imdgpu_ctx *ctx, const imdgpu_ttm_params *par);
int imdgpu_ttm_dims(imdgpu_ctx *ctx, int local_fd_dim[3], double fd_h[3]);
/* any member of `f` may be NULL */
int imdgpu_ttm_download(imdgpu_ctx *ctx, const imdgpu_ttm_fields *f);
int imdgpu_ttm_upload(imdgpu_ctx *ctx, const imdgpu_ttm_fields *f);
/* [update_fd, :233-488] uses numneighs/impuls/masse of the last calc_forces */
int imdgpu_ttm_update_fd(imdgpu_ctx *ctx, int steps);
/* [do_ADV(tau), :1321-1458, with do_COMMFLUX, :667-995] advection of the electron energy with the atoms that changed
 * FD cells, as the reference compiles it (ADVMODE 1 without ADV2D): x direction, lattice plane k = 1, then l1 <-> l2.
 * Needs update_fd's flux counters (option "ttm_flux"; option "ttm_adv" sets it and makes imdgpu_calc_ttm call
 * imdgpu_ttm_adv(1.0) after update_fd like calc_ttm :57-60).  On a lattice held as a whole (one rank, or the
 * replicated lattice of several ranks: there every rank block's plane k = 1 is advected, as the ranks of the reference
 * do between them) do_COMMFLUX has nothing to move; on the decomposed lattice imdgpu_dist_md_step exchanges the x faces
 * of {flux[8], U} itself between update_fd and do_ADV (imdgpu_ttm_commflux is the pack / unpack half for hosts with a
 * transport of their own; ranks at the ends of the x range do not wrap around, :724-760). */
int imdgpu_ttm_adv(imdgpu_ctx *ctx, double tau, int steps);
/* [do_COMMFLUX :667-995] dev_buf: 5 doubles per inner (j,k) = the eight counters as ints, then U.  unpack 0: side 0 packs
 * my column i = 1, side 1 my column lx-2; unpack 1: side 0 fills my ghost column i = 0, side 1 the ghost column lx-1 */
int imdgpu_ttm_commflux(imdgpu_ctx *ctx, int side, int unpack, double *dev_buf, int *ncells);
/* [laser_rescale_ttm, imd_laser.c:715-788 (-DLASER, laser_rescale_mode 4), with ttm_calc_depth :699-710]
 * Beer-Lambert source term of the electron heat equation: in every active FD cell
 *   source = exp(-mu * depth) * exp(-(timestep*steps - t_0)^2 / sigma_t_squared / 2) * p_peak,
 * depth = position of the cell's low corner along `dir` minus `offset`, clamped at 0.  The members are the reference's
 * globals laser_dir, laser_offset, laser_mu, laser_p_peak, laser_t_0, laser_sigma_t_squared after init_laser
 * (imd_laser.c:283-360: p_peak = mu * sigma_e / sigma_t / sqrt(2 pi), offset = get_surface()).
 * imdgpu_ttm_laser_rescale writes the term once (update_fd zeroes it every step); imdgpu_ttm_set_laser makes
 * imdgpu_calc_ttm / imdgpu_ttm_lattice_step do that after update_fd in every step (NULL switches it off). */
typedef struct {
  int    dir[3];
  double offset, mu, p_peak, t_0, sigma_t_squared;
} imdgpu_laser_params;
int imdgpu_ttm_laser_rescale(imdgpu_ctx *ctx, const imdgpu_laser_params *par, int steps);
int imdgpu_ttm_set_laser(imdgpu_ctx *ctx, const imdgpu_laser_params *par);
/* [Eabs_local, imd_ttm_3D.c:1647-1651] absorbed laser energy in eV booked by the diffusion sub-steps so far
 * (while the source is above 1e-3 of its peak: laser_active, imd_laser.c:744-747) */
int imdgpu_ttm_get_eabs(imdgpu_ctx *ctx, double *eabs_local);
/* [do_cell_activation, :1468-1596] */
int imdgpu_ttm_cell_activation(imdgpu_ctx *ctx);
/* [do_FILLMESH, :492-661] */
int imdgpu_ttm_fillmesh(imdgpu_ctx *ctx, int steps);
/* [ttm_fill_ghost_layers, :3314-3504] */
int imdgpu_ttm_fill_ghost_layers(imdgpu_ctx *ctx);
/* [do_DIFF(tau), :1603-1829] */
int imdgpu_ttm_diff(imdgpu_ctx *ctx, double tau);
/* [EOS_te_from_r_ee, :2786-2811 + fminbnd3.c:6-107] electron temperature in K from density (kg/m^3) and internal
 * energy (table units) for n points on host arrays: the reference's Brent minimisation over the ee(r, sqrt T) table */
int imdgpu_ttm_eos_te_from_r_ee(imdgpu_ctx *ctx, int n, const double *r, const double *e, double *te_out);
/* The reference interpolates its EOS tables linearly on a Delaunay TRIANGULATION of the table nodes (nn-c, binary
 * nn_interpol/libnn.a); on a regular grid which diagonal splits a cell is a property of that triangulator.  A host that
 * has the reference's interpolators (the level-1 shim does) reads the choice off them -- lpi_interpolate_point at every
 * cell centre equals the mean over one of the two diagonals -- and hands it over here; the look-up is then the same
 * triangle's plane as the reference's.  diag: (nx-1)*(ny-1) bytes, cell (ix, iy) at ix*(ny-1)+iy: 1 = diagonal
 * (ix+1,iy)-(ix,iy+1), 2 = diagonal (ix,iy)-(ix+1,iy+1), 0 = unknown (bilinear in that cell); NULL = bilinear everywhere. */
int imdgpu_ttm_set_eos_triangulation(imdgpu_ctx *ctx, int which /*0 cve, 1 ee*/, const unsigned char *diag);
/* The same map WITHOUT the reference's interpolator: nn-c builds its triangulation with Shewchuk's Triangle (divide-and-conquer
 * with alternating cuts, exact predicates); on a rectilinear grid the diagonals follow from that algorithm's tie rules alone,
 * whatever the node spacing.  Restated from the published algorithm (imdgpu_ttm.hip: GridDelaunay) and equal to libnn.a's
 * result on every grid tried (tests/golden/eos_triangulation.npz).  Host only, no context.  imdgpu_ttm_init and
 * imdgpu_ttm1d_init apply it to their tables by default (option "eos_bilinear" = 1 before init: plain bilinear look-up). */
int imdgpu_eos_triangulation(int nx, int ny, unsigned char *diag /* (nx-1)*(ny-1) bytes, coding as above */);
/* [lpi_interpolate_point as called at :2722 (which = 1: ee(r, sqrt T) table) and :2759 (which = 0: cve(r, T) table)]
 * the raw table look-up for n points (x, y) in the table's own coordinates on host arrays, no unit factors.  Rule: the
 * reference's -- linear on the triangle of nn-c's Delaunay triangulation of the table nodes (imdgpu_eos_triangulation, applied
 * by imdgpu_ttm_init).  After imdgpu_ttm_set_eos_triangulation(.., NULL) or with option eos_bilinear: bilinear inside the grid
 * cell, which agrees at nodes and on cell edges and differs inside a cell by at most |z00 + z11 - z10 - z01| / 4 */
int imdgpu_ttm_eos_lookup(imdgpu_ctx *ctx, int which, int n, const double *x, const double *y, double *z_out);
/* ---- 1D Maxwell solver as laser source (BASELINE config 5; imd_fdtd.c) ---------------------------------------------
 * [init_fdtd :594-763, init_pml :765-859] on a lattice of ONE row of FD cells (fd_ext spans all MD cells in y and z),
 * open in x.  SI units like the reference's parameters: I0 [W/m^2], lambda [m], sigma_t and t_0 [s], srcx [m] position
 * of the soft source, bw PML cells, Sc Courant number (<= 1).  DL: Drude-Lorentz constants eps_inf, gamma_p [1/s],
 * omegapl_L, Omega0_L, Gamma_L [rad/s] (fitDL's constant alternative, imd_ttm_3D.c:2921-2937; its tricubic tables
 * DL1..5.txt are not part of the reference tree); the Drude plasma frequency follows the cells' electron density.
 * After this call imdgpu_calc_ttm / imdgpu_ttm_lattice_step run calc_ttm's solver loop (imd_ttm_3D.c:96-122,163-179):
 * fdtd_substeps x [do_fdtd(t_SI); diff_substeps x (do_DIFF, do_FILLMESH); t_SI += tau_FDTD] in one launch. */
typedef struct imdgpu_fdtd_params { double I0, lambda, sigma_t, t_0, srcx; int bw; double Sc; double DL[5]; } imdgpu_fdtd_params;
int imdgpu_ttm_fdtd_init(imdgpu_ctx *ctx, const imdgpu_fdtd_params *p);
/* per-cell parameters DL[6] for every cell of the row (ghost cells included) instead of the constants; NULL: back */
int imdgpu_ttm_fdtd_set_dl(imdgpu_ctx *ctx, const double *DL);
/* [fitDL, imd_ttm_3D.c:2919-2968 (imd_ttm.c:2867-2916); read_tricub_interp / do_tricubinterp, imd_interpol.c:959-1084,
 * :740-957] the five tables DL1.txt .. DL5.txt over (rho [kg/m^3], lg Te [eV], lg Ti [eV]) as read_tricub_interp keeps
 * them: is x js x ks nodes, range = xmin xmax ymin ymax zmin zmax (line 2 of the file), node coordinates x[is], y[js],
 * z[ks], values f[(i*js + j)*ks + k].  With tables set, every do_FILLMESH of the solver loop takes the cell's
 * DL[0] = eps_inf and DL[1..4] (table value / hbar[eV s]) from them at (dens, lg max(Te, 0.0259), lg max(Ti, 0.0259)),
 * clamped like fitDL; a look-up outside a table yields -1 as in the reference.  tabs = NULL: back to the constants. */
typedef struct imdgpu_tricub_table { int is, js, ks; double range[6]; const double *x, *y, *z, *f; } imdgpu_tricub_table;
int imdgpu_ttm_fdtd_set_dl_tables(imdgpu_ctx *ctx, const imdgpu_tricub_table *tabs /* [5] */);
/* the device's do_tricubinterp on table `which` (0..4) at n points (x, y, z triples); its fitDL for n cells -> DL[6 n] */
int imdgpu_ttm_fdtd_dl_lookup(imdgpu_ctx *ctx, int which, int n, const double *xyz, double *out);
int imdgpu_ttm_fdtd_fitdl(imdgpu_ctx *ctx, int n, const double *temp, const double *md_temp, const double *dens, const double *ne,
                          double *DL);
/* [do_fdtd(t), imd_fdtd.c:128-306] nsub times from t_SI on (t += tau_FDTD after each) with the envelope value
 * laser_timefun held fixed as calc_ttm does within an MD step (:98); the dissipated power goes to the lattice's source */
int imdgpu_ttm_fdtd_substeps(imdgpu_ctx *ctx, double t_SI, int nsub, double tau_FDTD, double laser_timefun);
/* which: 0 Ezx, 1 Hy, 2 Jzx, 3 Jlzx, 4 Pzx, 5 sigmay, 6/7 Jzx/Jlzx of the sub-step before: nx values; 8: the cells' DL[6], 6 nx values */
int imdgpu_ttm_fdtd_fields(imdgpu_ctx *ctx, int which, double *out);
int imdgpu_ttm_fdtd_get(imdgpu_ctx *ctx, double *dt_FDTD, double *E0, int *srcx_int, double *t_SI);
int imdgpu_ttm_fdtd_set_time(imdgpu_ctx *ctx, double t_SI);
/* [CFL_maxdt, :2637-2691] */
int imdgpu_ttm_cfl_maxdt(imdgpu_ctx *ctx, int steps, double *max_dt_ttm);
/* [calc_ttm, first block :50-77 without do_tmm/do_ADV]: update_fd, cell
 * activation, FILLMESH, ghosts, diff_substeps x (DIFF, FILLMESH, ghosts), CFL */
int imdgpu_calc_ttm(imdgpu_ctx *ctx, int steps);
/* [move_atoms_ttm, imd_integrate.c:581-811] */
int imdgpu_move_atoms_ttm(imdgpu_ctx *ctx, double *tot_kin_energy);

/* ---- the reference's OWN 1D two-temperature model (imd_ttm.c; the file its Makefile builds, -DTTM1D) ------------------
 * Atoms binned into ttmdimx slabs by position ((int)(x / fd_h.x), imd_ttm.c:296-298), slab sums as global arrays, activity =
 * natoms >= fd_min_atoms && dens > RHOMIN (2 kg/m^3, :69), a static virtual lattice of vlatdim cells behind the last active
 * cell (:1013-1024,1595-1650), diffusion sub-steps from the stability limit (:97-99), U from the EOS after the loop
 * (:119-133); move_atoms_ttm takes xi / v_com of the slab an atom is in now (imd_integrate.c:649-760).  One rank; no do_tmm
 * (its input files are not part of the tree).  ref_ranks: how many ranks the REFERENCE run to be reproduced cut the row
 * into (every cell of the piece that holds the last active cell takes the virtual lattice as +x neighbour, :1535-1540).
 * Lattice arrays: ttmdimx + 2 entries (ghost cells at both ends), global slab arrays: ttmdimx, virtual lattice: vlatdim. */
typedef struct {
  int    nx;                      /* ttmdimx (globals.h:1172) */
  int    fd_min_atoms, fd_n_timesteps, vlatdim, vlatbuffer, ref_ranks, adv;
  double vlatdens, atomic_weight, atomic_charge;
  int    cve_nx, cve_ny;
  double cve_xmin, cve_xmax, cve_ymin, cve_ymax;
  const double *cve_z;
  int    ee_nx, ee_ny;
  double ee_xmin, ee_xmax, ee_ymin, ee_ymax;
  const double *ee_z;
} imdgpu_ttm1d_params;
typedef struct {
  int    *natoms, *natoms_old;
  double *temp, *temp2, *U, *U2, *xi, *md_temp, *source, *dens, *fd_k, *fd_g, *Z, *Ce, *ne;
  int    *natomsglobal, *fluxfromleft, *fluxfromright;
  double *vcomglobal;             /* 3 x ttmdimx: x, y, z */
  double *xiarr_global;
  double *vlat_temp, *vlat_md_temp, *vlat_fd_k, *vlat_fd_g, *vlat_Ce;
  double *scalars;                /* 4: last_active_cell_global, cur_vlattice_proc, diff_substeps, max_dt_ttm */
} imdgpu_ttm1d_fields;
/* [init_ttm, imd_ttm.c:955-1206] */
int imdgpu_ttm1d_init(imdgpu_ctx *ctx, const imdgpu_ttm1d_params *par);
/* [calc_ttm, imd_ttm.c:75-155]: update_fd, do_ADV, do_cell_activation, do_FILLMESH, CFL_maxdt, the loop, the EOS update of U;
 * imdgpu_md_step runs it beside the force passes like the 3D calc_ttm when a 1D lattice is attached */
int imdgpu_ttm1d_calc(imdgpu_ctx *ctx, int steps);
/* function level: bit 0 [update_fd :161-548], 1 [do_ADV :1208-1309], 2 [do_cell_activation :1311-1425 + do_FILLMESH :629-850 +
 * ghosts], 3 [CFL_maxdt :2444-2481 + the sub-step count :97-99], 4 [one do_DIFF :1473-1680], 5 [do_FILLMESH + ghosts],
 * 6 [the loop], 7 [U from the EOS, :119-133], 8 [one do_tmm, imd_tmm.c:193-455: needs imdgpu_ttm1d_tmm_init] */
int imdgpu_ttm1d_stage(imdgpu_ctx *ctx, int steps, int stages);
/* ---- the transfer-matrix Helmholtz solver as laser source of the 1D model (imd_tmm.c; the reference's Makefile target
 * imd_mpi_eam_ttm_tmm_nbl): [tmm_init :98-191].  SI units like the parameter file: I0 [W/m^2], lambda [m], laser_sigma_t /
 * laser_t_0 and the second pulse laser_sigma_t1 / laser_t_1 [s], tmm_threshold (absorption depth: stop where the transmitted
 * intensity falls below exp(-tmm_threshold)), tmm_laser_threshold (field strength relative to the peak below which the laser
 * counts as off; globals.h:453,479).  Tables: eps_bb = rows of ../alu_eps_bb.dat (lambda [um], Re, Im of the interband
 * permittivity, ascending), K12 = the 1500 rows of ../K12.dat (K1, K2 at nu = (row + 1) / 30), phase = the grid of
 * ../EOS_phase_from_r_ti.txt (density [kg/m^3] x Ti [K], x slowest).  After this call imdgpu_ttm1d_calc (and imdgpu_md_step
 * with a 1D lattice) runs do_tmm [:193-455, tmm_get_epsilon :501-586] at the top of every diffusion sub-step as calc_ttm does
 * (imd_ttm.c:113): the absorbed power per cell becomes the source of do_DIFF, Eabs_local is booked (:1515-1518). */
typedef struct imdgpu_tmm_params {
  double I0, lambda, laser_sigma_t, laser_t_0, laser_sigma_t1, laser_t_1, tmm_threshold, tmm_laser_threshold;
  int eps_bb_rows; const double *eps_bb;      /* [rows][3] */
  int K12_rows;    const double *K12;         /* [rows][2] */
  struct { int nx, ny; double xmin, xmax, ymin, ymax; const double *z; } phase;
} imdgpu_tmm_params;
int imdgpu_ttm1d_tmm_init(imdgpu_ctx *ctx, const imdgpu_tmm_params *p);
/* scalars[8]: tmm_time [s], laser_active, Eabs_local [eV], I_t, tmm_refl, tmm_trans, tmm_abs, absorption depth in cells;
 * eps_re / eps_im / qabs (tmm_Qabs) / active: nx values each as the last do_tmm left them; too_short: a call found the sample
 * too short for tmm_threshold (the reference aborts there).  NULL = skip */
int imdgpu_ttm1d_tmm_get(imdgpu_ctx *ctx, double *scalars, double *eps_re, double *eps_im, double *qabs, int *active, int *too_short);
int imdgpu_ttm1d_tmm_set_time(imdgpu_ctx *ctx, double tmm_time);
/* -DELECPRESS of the reference (commented out at imd_ttm.c:60 / imd_integrate.c:25; a build flag there, a call here): the table of
 * EOS_pe_from_r_te [Pa over density x Te[K], x slowest; imd_ttm.c:1133-1135,2638-2693].  calc_ttm then ends with
 * [do_electronic_pressure :553-623] and move_atoms_ttm subtracts timestep * epress_deriv[slab] from p_x [imd_integrate.c:657-659,
 * 741-743].  With ref_ranks > 1 an atom feels the force only if the rank that owns it (by position) also owns its slab, as in
 * the reference, where every rank fills epress_deriv for its own share of the row.  z = NULL: off */
int imdgpu_ttm1d_set_epress(imdgpu_ctx *ctx, int nx, int ny, double xmin, double xmax, double ymin, double ymax, const double *z);
int imdgpu_ttm1d_epress_get(imdgpu_ctx *ctx, double *epress_global, double *epress_deriv);
/* [move_atoms_ttm, TTM1D branch, imd_integrate.c:649-760] */
int imdgpu_ttm1d_move_atoms(imdgpu_ctx *ctx, double *tot_kin_energy);
int imdgpu_ttm1d_download(imdgpu_ctx *ctx, const imdgpu_ttm1d_fields *f);
int imdgpu_ttm1d_upload(imdgpu_ctx *ctx, const imdgpu_ttm1d_fields *f);

/* ---- multi-GPU spatial decomposition (one rank per GPU) --------------------- */
/* The reference's Plimpton exchange (send_cells / send_forces, imd_comm_force_3d.c:
 * 222-396,569-715) and atom migration (fix_cells -> send_atoms, imd_fix_cells_3d.c:
 * 43-306,436-542) become: decomposition logic on the host (imd_amd/parallel.py or an
 * MPI host), transport over RCCL/xGMI, and these generic device operations.  With
 * cpu_dim > 1 in imdgpu_set_geometry every ghost cell is owned by the host layer.
 * All dev_* pointers are device memory of the same GPU. */
int imdgpu_get_cell_tables(imdgpu_ctx *ctx, int *cell_start /*nallcells*/, int *cell_cnt /*nallcells*/);
/* pick the LDS tile shape from the current cell occupancies (ghost cells included); *changed != 0 means the
 * local atoms were re-sorted and the halo index lists must be rebuilt */
int imdgpu_autotile(imdgpu_ctx *ctx, int *changed);
int imdgpu_set_ghost_layout(imdgpu_ctx *ctx, const int *cell_start, const int *cell_cnt, int nghost);
/* Host-buffer variants for a C host that keeps the reference's OWN halo exchange (level-1 shim in an MPI build: after
 * send_cells(copy_cell, ...), imd_forces_nbl.c:311, the host's buffer cells hold the ghost atoms; after
 * send_cells(copy_dF, ...), :786, their EAM_DF).  ghost_cell_n: atoms per cell for all nallcells cells, 0 for inner
 * cells; ort / sorte / dF: the ghost atoms in cell order (cell index ascending, slots as on the host). */
int imdgpu_upload_ghosts(imdgpu_ctx *ctx, const int *ghost_cell_n, int nghost, const double *ort, const int *sorte);
int imdgpu_update_ghosts(imdgpu_ctx *ctx, int nghost, const double *ort /*or NULL*/, const double *dF /*or NULL*/);
/* number of visible GPUs (to spread the ranks of a node) */
int imdgpu_device_count(int *n);
/* what: 0 = (x,y,z,dF) record of 4 doubles, positions shifted by `shift` (periodic image,
 *       imd_comm_force_3d.c:247-266); 1 = dF only (1 double) [copy_dF]; 2 = type (int);
 *       3 = shifted position only (3 doubles): the per-step position halo */
int imdgpu_halo_gather(imdgpu_ctx *ctx, int what, const int *dev_idx, int n, const double shift[3], void *dev_out);
/* both faces of a stage in one launch: entries [0, n_lo) of dev_idx are the low face (shift_lo), the rest the high face */
int imdgpu_halo_gather2(imdgpu_ctx *ctx, int what, const int *dev_idx, int n_lo, int n, const double shift_lo[3],
                        const double shift_hi[3], void *dev_out);
int imdgpu_halo_scatter(imdgpu_ctx *ctx, int what, const int *dev_idx, int n, const void *dev_in);
/* migration along one direction: mark + count atoms whose cell belongs to the lower / higher neighbour rank,
 * pack them (10 doubles each: x y z px py pz mass type id pad), drop them and append the arrivals */
int imdgpu_migrate_scan(imdgpu_ctx *ctx, int dim, int *n_lo, int *n_hi);
int imdgpu_migrate_gather(imdgpu_ctx *ctx, int dir /*0 low, 1 high*/, void *dev_out);
int imdgpu_migrate_apply(imdgpu_ctx *ctx, const void *dev_in_lo, int n_in_lo, const void *dev_in_hi, int n_in_hi);
/* the MD step in pieces, all asynchronous on the context's stream: exchanges go between them.
 * phase 1 = pair + rho pass, 2 = embedding pass, 3 = EAM force pass + sums (calc_forces, imd_forces_nbl.c).
 * With one atom type and pair/rho tables on one grid the embedding pass runs inside phase 1 (get_info
 * "fused_embedding"): F'(rho) of the local atoms is complete after phase 1, phase 2 then only refreshes periodic images. */
/* Identity of the context's GPU as text ("<PCI bus id> uuid <hex> <name>, <n> CUs"): what a multi-rank host prints so that a reader
 * sees N ranks on N different devices (no counterpart in the reference, which has no devices; cf. its rank banner, imd.c:80-90). */
int imdgpu_device_identity(imdgpu_ctx *ctx, char *buf, int len);
int imdgpu_forces_phase(imdgpu_ctx *ctx, int phase);
int imdgpu_integrate(imdgpu_ctx *ctx);                 /* move_atoms_{nve|ttm} + displacement check */
int imdgpu_get_invalid(imdgpu_ctx *ctx, int *flag);    /* check_nblist result of this rank (synchronises) */
int imdgpu_invalidate(imdgpu_ctx *ctx);                /* some rank asked for a rebuild (MPI_Allreduce MAX, :1087) */
/* TTM with several ranks: the FD lattice is replicated; each rank computes update_fd for the block over
 * its own MD cells (6 doubles per FD cell: natoms, md_temp, dens, vcom; with option "ttm_flux" 10: the eight flux
 * counters follow as ints), blocks are all-gathered and folded in, the sub-steps run everywhere.  With option
 * "ttm_adv" the host calls imdgpu_ttm_adv(1.0) between apply_blocks and lattice_step (calc_ttm's order, :57-60). */
int imdgpu_ttm_local_block(imdgpu_ctx *ctx, int steps, double *dev_out, int *ncells_block);
int imdgpu_ttm_apply_blocks(imdgpu_ctx *ctx, int steps, const double *dev_in, int nblocks, const int *dev_offs);
int imdgpu_ttm_lattice_step(imdgpu_ctx *ctx, int steps);
/* Alternative that keeps the per-rank TTM cost constant as ranks are added: imdgpu_set_option("ttm_decomposed", 1)
 * before imdgpu_ttm_init gives every rank only ITS block of the lattice (as in the reference); the ghost faces of
 * {natoms, temp, fd_k} that the reference moves with MPI_Sendrecv after every sub-step (ttm_fill_ghost_layers,
 * imd_ttm_3D.c:3314-3504) go through a mailbox in host shared memory which the GPUs of all ranks of the node read
 * and write from inside the sub-step kernels (no collective, no launch).  host_base: nranks * bytes_per_rank bytes
 * mapped by every rank, zero-filled before the first attach; nbr[2*d+side] = rank of the neighbouring block
 * (this rank itself for a periodic direction with one rank), -1 = open boundary. */
int imdgpu_ttm_mailbox_bytes(imdgpu_ctx *ctx, size_t *bytes_per_rank);
int imdgpu_ttm_mailbox_attach(imdgpu_ctx *ctx, void *host_base, size_t bytes_per_rank, int rank, int nranks, const int nbr[6]);
/* Decomposed lattice without the host in the loop: every rank owns a buffer of GHOST RECORDS in device memory (per cell of
 * its six ghost faces and parity four tagged 8-byte granules {tag | half of Te or kappa}, plus {tag | natoms}); the ranks
 * beyond its faces write their boundary cells' state straight into it from inside the persistent sub-step kernel (stores
 * over xGMI; system-scope atomics on fine-grained memory) and it polls locally -- ttm_fill_ghost_layers' six MPI_Sendrecv
 * per sub-step (imd_ttm_3D.c:3314-3504) without a launch, a collective or the host.  Used whenever the persistent kernel
 * applies (steps >= 1, even sub-step count, block shape); other steps fall back to the mailbox above, which therefore
 * stays attached.  imdgpu_dist_rebuild sets all of this up by itself (imdgpu_dist_ttm_remote_setup; IMDGPU_TTM_REMOTE=0
 * keeps the mailbox); the pieces for hosts with their own launcher: */
int imdgpu_ttm_remote_init(imdgpu_ctx *ctx, int coarse, void **dev_ptr, size_t *bytes);
int imdgpu_ttm_remote_ipc_handle(imdgpu_ctx *ctx, void *handle64, int *fine_grained);
int imdgpu_ttm_remote_open(imdgpu_ctx *ctx, const void *handle64, void **dev_ptr);
/* peer[q]: the buffer of the rank beyond face q (2d: +, 2d+1: -) as this process addresses it, NULL = none */
int imdgpu_ttm_remote_attach(imdgpu_ctx *ctx, void *const peer[6]);
int imdgpu_ttm_remote_detach(imdgpu_ctx *ctx);
/* set-up self-test: phase 0 stores `word` into a spare granule of every neighbour's buffer (the sub-step kernel's store),
 * phase 1 -- after all ranks met -- reads the own spare granules with its polling load; *bad = bit mask of faces without it */
int imdgpu_ttm_remote_ping(imdgpu_ctx *ctx, int phase, unsigned word, int *bad);
/* collective and failure-safe: if any rank cannot provide, open or reach a buffer, EVERY rank returns the error (and stays on
 * the mailbox); imdgpu_dist_rebuild then prints the reason and goes on with the mailbox */
int imdgpu_dist_ttm_remote_setup(imdgpu_ctx *ctx);
/* [calc_ttm] for this rank's block: update_fd, activation, FILLMESH, face publication, the sub-steps */
int imdgpu_ttm_local_step(imdgpu_ctx *ctx, int steps);

/* ---- native multi-GPU step: C host + RCCL over xGMI (imd_amd/csrc/imdgpu_dist.hip) ------------------------------
 * Replaces, per step, what the reference does with MPI: send_cells(copy_cell) before pass 1 and send_cells(copy_dF)
 * before the EAM force pass (imd_forces_nbl.c:311,786; imd_comm_force_3d.c:222-396) and the MPI_Allreduce of the list
 * validity (imd_forces_nbl.c:1087).  The 26 ghost regions of a rank are filled by 26 DIRECT messages in one RCCL group
 * (every GPU of the node has its own xGMI link to every other one); the validity flag is all-reduced on the device and
 * guards the next step's pass 1, which is enqueued before the host reads the flag.  Rebuilds (migration, ghost layout,
 * plan, list) stay with the host layer between two calls.  All of it runs on the context's stream. */
/* host transport for rehearsals without RCCL (several ranks on ONE GPU): op 0 = point-to-point segments (element
 * offsets / counts into dev_send / dev_recv, elem_bytes each; messages between one pair of ranks match in the order
 * they are listed), 1 = all-reduce MAX of nsend int32 in place, 2 = all-gather of nsend doubles per rank into dev_recv,
 * 3 = all-reduce SUM of nsend doubles in place.  Called with the stream drained; returns 0 on success. */
typedef int (*imdgpu_exchange_fn)(void *user, int op, void *dev_send, void *dev_recv, int nsend, const int *send_peer,
                                  const long long *send_off, const long long *send_cnt, int nrecv, const int *recv_peer,
                                  const long long *recv_off, const long long *recv_cnt, int elem_bytes);
/* rank 0 makes the communicator ids -- nbytes = 256: two ncclUniqueId (halos + flags; lattice stream), nbytes = 384: a third one
 * for the position halo on its own stream (option "halo_overlap") --, the launcher hands them to every rank; init takes 256 or 384 */
int imdgpu_dist_unique_id(void *out, int nbytes);
int imdgpu_dist_init_rccl(imdgpu_ctx *ctx, int nranks, int rank, const void *unique_id, int nbytes);
int imdgpu_dist_init_callback(imdgpu_ctx *ctx, int nranks, int rank, imdgpu_exchange_fn fn, void *user);
/* after imdgpu_set_ghost_layout: the send segments (peer rank, atoms, periodic shift added by the sender
 * [imd_comm_force_3d.c:247-266], atom indices on the device, segment after segment) and the receive segments; the
 * ghost atoms [nlocal, nlocal + nghost) lie in the order of the receive segments */
int imdgpu_dist_set_plan(imdgpu_ctx *ctx, int nsend, const int *send_peer, const int *send_count, const double *send_shift,
                         const int *dev_send_idx, int nrecv, const int *recv_peer, const int *recv_count);
/* one halo exchange: what = 0 (x,y,z,dF), 1 dF [copy_dF], 2 type, 3 (x,y,z) [copy_cell per step] */
int imdgpu_dist_halo(imdgpu_ctx *ctx, int what);
/* replicated TTM lattice: FD-cell offset of every rank's update_fd block, 3 ints per rank */
int imdgpu_dist_set_ttm_blocks(imdgpu_ctx *ctx, const int *offs);
/* one MD step (forces, calc_ttm, integrator, list check); *rebuild_needed != 0: some rank's list became invalid in this
 * step -- migrate / re-bin / re-plan / make_nblist before the next call.  After the call the force arrays already
 * belong to the NEXT step's pass 1 (enqueued ahead); positions, momenta and sums are those of this step. */
int imdgpu_dist_md_step(imdgpu_ctx *ctx, int steps, int *rebuild_needed);
/* sum of n <= 16 host doubles over the ranks (energies) */
int imdgpu_dist_allreduce_sum(imdgpu_ctx *ctx, double *vals, int n);
/* The REBUILD, native too (a C host needs no other layer): what fix_cells -> send_atoms (imd_fix_cells_3d.c:43-306,
 * 436-542; atom records of imd_mpi_util.c:349-650) and the first send_cells of a new list (imd_comm_force_3d.c:222-396,
 * buffers of imd_mpi_util.c:850-1059) do in the reference: migration along x, y, z (counts, then 10 doubles per atom),
 * re-binning, per-cell counts to the 26 peers, ghost layout and send index lists (built on the device), tile choice,
 * first halo, imdgpu_make_nblist.  Collective: every rank calls it; ranks are numbered x slowest, z fastest
 * (rank = (cx * cpu_dim.y + cy) * cpu_dim.z + cz as in imd_geom_mpi_3d.c:32-90 with MPI_Cart_create's row-major order). */
int imdgpu_dist_rebuild(imdgpu_ctx *ctx);
int imdgpu_dist_rebuild_count(imdgpu_ctx *ctx, int *n);
/* one MD step of a multi-rank run for a C host: imdgpu_dist_rebuild when the list is stale, then imdgpu_dist_md_step
 * (the main_loop body, imd_main_3d.c:406-812, with calc_forces / calc_ttm / move_atoms / check_nblist on the device) */
int imdgpu_dist_step(imdgpu_ctx *ctx, int steps);
/* complete forces of the CURRENT positions (three passes with their halos), e.g. before imdgpu_download_atoms */
int imdgpu_dist_calc_forces(imdgpu_ctx *ctx);
/* the static part of the plan, host only (inspection, tests): the 26 directions in the fixed order of the exchange
 * (dx slowest; (0,0,0) skipped).  seg_out: 6 ints per direction {dx, dy, dz, rank the boundary layer goes to or -1, rank
 * whose layer fills my ghost region or -1, cells of the layer}; shift_out: 3 doubles per direction (periodic image shift
 * the SENDER adds, imd_comm_force_3d.c:247-266); send_cells / recv_cells (may be NULL): linear cell indices of the layers
 * that are really sent / received, direction after direction. */
int imdgpu_dist_plan_geometry(const int cpu_dim[3], const int my_coord[3], const int cell_dim[3], const int pbc_dirs[3],
                              const double box[9], int *seg_out, double *shift_out, int *send_cells, int *nsend_cells,
                              int *recv_cells, int *nrecv_cells);
/* In-process transport: the ranks are contexts of ONE process, each driven by its own host thread (rehearsals of many
 * ranks on one GPU -- a GPU box admits only a few processes --, or one process driving all GPUs of a node with peer
 * copies).  Same contract as the host callback; every collective call must be made by all ranks of the group. */
int imdgpu_local_group_create(int nranks, void **group);
void imdgpu_local_group_destroy(void *group);
int imdgpu_local_group_barrier(void *group);
/* a zero-filled page-locked host region that all ranks of the group see under one name (e.g. the TTM face mailbox of
 * imdgpu_ttm_mailbox_attach); the first caller allocates, later callers get the same pointer; freed with the group */
int imdgpu_local_group_shared(void *group, const char *name, size_t bytes, void **ptr);
int imdgpu_dist_init_local(imdgpu_ctx *ctx, void *group, int rank);

/* ---- profiling ------------------------------------------------------------ */
/* Accumulate HIP-event timings per kernel class on the context's stream. */
int imdgpu_profile_enable(imdgpu_ctx *ctx, int on);
int imdgpu_profile_reset(imdgpu_ctx *ctx);
/* names: "pass1","pass2","pass3","ghosts","nblist","fix_cells","integrate",
 * "check","reduce","ttm_update_fd","ttm_diff","ttm_fillmesh" */
int imdgpu_profile_get(imdgpu_ctx *ctx, const char *name, long long *launches, double *ms);

#ifdef __cplusplus
}
#endif
#endif
