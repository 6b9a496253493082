/*
 * ref_probe_ttm.c -- TEST INFRASTRUCTURE (oracle side), not product code.
 *
 * Probe for the reference's 3D two-temperature model (imd_ttm_3D.c), attached
 * with ld --wrap by oracle/build_ref.sh (target mpi_eam_ttm3d_nbl).
 *
 * Why a probe and not the plain program: the shipped 3D TTM cannot be built as
 * a whole program -- imd_ttm_3D.c only compiles with -DTMM (it uses tmm_time,
 * laser_active, ... unconditionally, :73,:82,:161) but the TMM solver the
 * reference ships (imd_tmm.c) is written for the 1D lattice and does not
 * compile against the 3D one.  The probe therefore
 *   - answers the two TMM entry points (tmm_init, do_tmm) with "no laser"
 *     (return 0, nothing deposited; all fixtures run with the laser off), and
 *   - drives, from __wrap_calc_ttm, the reference's OWN compiled functions in
 *     the order of the first block of calc_ttm (imd_ttm_3D.c:50-77):
 *       update_fd -> do_cell_activation -> do_FILLMESH -> ttm_fill_ghost_layers
 *       -> diff_substeps x { do_DIFF(tau); do_FILLMESH; ttm_fill_ghost_layers }
 *       -> CFL_maxdt
 *     leaving out do_tmm (laser) and, unless IMD_PROBE_ADV is set, do_ADV (advection: see DESIGN.md 9 for
 *     what the reference does there), dumping the lattice in raw fp64 between the stages.
 *     IMD_PROBE_EOSINV=1 additionally evaluates the reference's EOS_ee_from_r_te / EOS_te_from_r_ee on sample points.
 * Every number in the TTM fixtures is thus computed by the reference's code.
 */
#include "imd.h"

void __real_calc_forces(int steps);

int __wrap_tmm_init(void) { return 0; }
int __wrap_do_tmm(real dt) { (void) dt; return 0; }

static int want(int s)
{
  const char *e = getenv("IMD_PROBE_STEPS");
  if (e == NULL) return s == 0;
  while (*e) {
    char *end; long v = strtol(e, &end, 10);
    if (end == e) break;
    if (v == s) return 1;
    e = end; while (*e == ',' || *e == ' ') e++;
  }
  return 0;
}

static FILE *tf = NULL;
static void wi(int v)    { fwrite(&v, sizeof(int), 1, tf); }
static void wd(double v) { fwrite(&v, sizeof(double), 1, tf); }

static void dump_lattice(const char *tag)
{
  int i, j, k;
  char t[16];
  memset(t, 0, sizeof(t)); strncpy(t, tag, 15);
  fwrite(t, 1, 16, tf);
  wd(max_dt_ttm);
  for (i = 0; i < local_fd_dim.x; i++)
    for (j = 0; j < local_fd_dim.y; j++)
      for (k = 0; k < local_fd_dim.z; k++) {
        ttm_Element *e = &l1[i][j][k], *e2 = &l2[i][j][k];
        wi(e->natoms); wi(e->natoms_old);
        wd(e->temp); wd(e->xi); wd(e->md_temp); wd(e->source);
        wd(e->vcomx); wd(e->vcomy); wd(e->vcomz);
        wd(e->dens); wd(e->fd_k); wd(e->fd_g); wd(e->Z); wd(e->Ce); wd(e->ne); wd(e->U);
        wd(e2->temp); wd(e2->U);
      }
}

/* IMD_PROBE_EOSINV=1: at step 0 evaluate the reference's own EOS_ee_from_r_te and EOS_te_from_r_ee (Brent minimiser,
 * imd_ttm_3D.c:2694-2811, fminbnd3.c) on a fixed set of (density, temperature) points and dump inputs and results */
static void dump_eos_inverse(void)
{
  static const double rs[] = {1200.0, 1900.0, 2350.0, 2699.0, 2950.0, 3300.0};
  static const double ts[] = {310.0, 555.5, 1000.0, 2345.0, 7000.0, 15000.0, 42000.0};
  static const double fe[] = {1.0, 0.97, 1.0004};         /* on the curve and slightly off it */
  char fn[1024];
  const char *dir = getenv("IMD_PROBE_DIR");
  FILE *f;
  int a, b, c, n = 0;
  sprintf(fn, "%s/eosinv.%d.bin", dir ? dir : ".", myid);
  f = fopen(fn, "wb");
  fwrite("IMDEOSIN", 1, 8, f);
  n = 6 * 7 * 3;
  fwrite(&n, sizeof(int), 1, f);
  for (a = 0; a < 6; a++) for (b = 0; b < 7; b++) for (c = 0; c < 3; c++) {
    double r = rs[a], t = ts[b], e = EOS_ee_from_r_te(r, t) * fe[c], tout = EOS_te_from_r_ee(r, e);
    fwrite(&r, sizeof(double), 1, f); fwrite(&t, sizeof(double), 1, f);
    fwrite(&e, sizeof(double), 1, f); fwrite(&tout, sizeof(double), 1, f);
  }
  fclose(f);
}

/* IMD_PROBE_EOSGRID=1: at step 0 read points "x y" from eos_samples.txt (working directory) and evaluate the reference's
 * interpolators on both EOS tables at them: raw lpi_interpolate_point results, no unit factors */
static void dump_eos_grid(void)
{
  char fn[1024];
  const char *dir = getenv("IMD_PROBE_DIR");
  FILE *in = fopen("eos_samples.txt", "r"), *f;
  double x, y;
  int n = 0;
  if (in == NULL) return;
  sprintf(fn, "%s/eosgrid.%d.bin", dir ? dir : ".", myid);
  f = fopen(fn, "wb");
  while (2 == fscanf(in, "%lf %lf", &x, &y)) {
    point a, b;
    a.x = x; a.y = y; a.z = 0; b = a;
    lpi_interpolate_point(intp_cve_from_r_te.interpolator, &a);
    lpi_interpolate_point(intp_ee_from_r_tesqrt.interpolator, &b);
    fwrite(&x, sizeof(double), 1, f); fwrite(&y, sizeof(double), 1, f);
    fwrite(&a.z, sizeof(double), 1, f); fwrite(&b.z, sizeof(double), 1, f);
    n++;
  }
  fclose(in); fclose(f);
}

/* advection bookkeeping of update_fd: the eight atom-flux counters per FD cell (imd_ttm_3D.c:335-370) */
static void dump_flux(const char *tag)
{
  int i, j, k, q;
  char t[16];
  memset(t, 0, sizeof(t)); strncpy(t, tag, 15);
  fwrite(t, 1, 16, tf);
  wd(-1.0);                                  /* marks an int record: 8 ints per cell follow */
  for (i = 0; i < local_fd_dim.x; i++)
    for (j = 0; j < local_fd_dim.y; j++)
      for (k = 0; k < local_fd_dim.z; k++)
        for (q = 0; q < 8; q++) wi(l1[i][j][k].flux[q]);
}

static void open_dump(double tau)
{
  char fn[1024];
  const char *dir = getenv("IMD_PROBE_DIR");
  sprintf(fn, "%s/ttm.%d.%d.bin", dir ? dir : ".", myid, steps);
  tf = fopen(fn, "wb");
  fwrite("IMDTTMPR", 1, 8, tf);
  wi(1); wi(steps);
  wi(local_fd_dim.x); wi(local_fd_dim.y); wi(local_fd_dim.z);
  wi(fd_ext.x); wi(fd_ext.y); wi(fd_ext.z);
  wi(fd_min_atoms); wi(diff_substeps); wi(fd_n_timesteps);
  wd(fd_h.x); wd(fd_h.y); wd(fd_h.z); wd(fd_vol); wd(neighvol);
  wd(atomic_weight); wd(atomic_charge); wd(timestep); wd(tau); wd(omega_laser);
}

#ifdef PROBE_TTM_DROPIN
/* Drop-in build (oracle/build_ref.sh dropin: imd_mpi_eam_ttm3d_nbl_gpu): calc_ttm is the one of
 * imd_amd/host/imd_shim.c (device); only the dump of the lattice it left in l1/l2 is added here. */
void __real_calc_ttm(void);
void __wrap_calc_ttm(void)
{
  int dumping = want(steps);
  __real_calc_ttm();
  if (dumping) {
    open_dump(timestep / ((double) diff_substeps));
    dump_lattice("cfl");
    fwrite("ENDTTMPR", 1, 8, tf);
    fclose(tf); tf = NULL;
  }
}
#else
#ifdef LASER
/* the globals laser_rescale_ttm reads (after init_laser, imd_laser.c:283-360) */
static void dump_laser_params(const char *tag)
{
  char t[16];
  memset(t, 0, sizeof(t)); strncpy(t, tag, 15);
  fwrite(t, 1, 16, tf);
  wd(-2.0);                                  /* marks a parameter record: 10 doubles follow */
  wd((double) laser_dir.x); wd((double) laser_dir.y); wd((double) laser_dir.z);
  wd(laser_offset); wd(laser_mu); wd(laser_p_peak); wd(laser_t_0); wd(laser_sigma_t_squared);
  wd(laser_sigma_e); wd(laser_sigma_t);
}
#endif

void __wrap_calc_ttm(void)
{
  int i, dumping = want(steps);
  if (steps == 0 && getenv("IMD_PROBE_EOSINV")) dump_eos_inverse();
  if (steps == 0 && getenv("IMD_PROBE_EOSGRID")) dump_eos_grid();
  double tau;
  diff_substeps = fd_n_timesteps;
  tau = timestep / ((double) diff_substeps);
  if (dumping) {
    open_dump(tau);
    dump_lattice("entry");
  }
  update_fd();
  if (dumping) dump_lattice("update_fd");
  if (dumping && getenv("IMD_PROBE_FLUX")) dump_flux("flux");   /* update_fd's atom-flux counters (:335-370) */
#ifdef LASER
  if (getenv("IMD_PROBE_LASER")) {         /* Beer-Lambert source term (imd_laser.c:715-788; calc_ttm's second block calls it
                                              before every do_DIFF, :192-197, with the same value within an MD step:
                                              here once for the dump, then in the sub-step loop below) */
    laser_rescale_ttm();
    if (dumping) { dump_laser_params("laserpar"); dump_lattice("laser"); }
  }
#endif
  if (getenv("IMD_PROBE_ADV")) {           /* advection step of calc_ttm (imd_ttm_3D.c:58-60); returns at once for steps < 1 */
    if (dumping && !getenv("IMD_PROBE_FLUX")) dump_flux("flux");
    do_ADV(1.0);
    if (dumping) dump_lattice("adv");
  }
  do_cell_activation();
  do_FILLMESH();
  ttm_fill_ghost_layers();
  if (dumping) dump_lattice("fillmesh0");
  for (i = 0; i < diff_substeps; i++) {
#ifdef LASER
    if (getenv("IMD_PROBE_LASER")) laser_rescale_ttm();   /* before EVERY do_DIFF as in :192-197: it writes l1 only, and
                                                              do_DIFF has just swapped the lattices */
#endif
    do_DIFF(tau);
    if (dumping && i == 0) dump_lattice("diff_only1");
    do_FILLMESH();
    ttm_fill_ghost_layers();
    if (dumping && (i == 0 || i == 1)) dump_lattice(i == 0 ? "substep1" : "substep2");
  }
  if (dumping) dump_lattice("substeps_done");
#ifdef LASER
  if (dumping && getenv("IMD_PROBE_LASER")) {              /* do_DIFF's bookkeeping of the absorbed energy, :1647-1651 */
    char t[16];
    memset(t, 0, sizeof(t)); strncpy(t, "eabs", 15);
    fwrite(t, 1, 16, tf);
    wd(-3.0);                                              /* marks a 2-double record */
    wd(Eabs_local); wd((double) laser_active);
  }
#endif
  CFL_maxdt();
  if (dumping) {
    dump_lattice("cfl");
    fwrite("ENDTTMPR", 1, 8, tf);
    fclose(tf); tf = NULL;
  }
}
#endif /* PROBE_TTM_DROPIN */
