/* ref_probe_fdtd.c -- TEST INFRASTRUCTURE.  Harness around the reference's OWN 1D Maxwell solver.
 *
 * imd_fdtd.c (do_fdtd, fdtd_softsource, init_fdtd, init_pml: 1D Yee scheme, Drude-Lorentz ADE, split-field PML,
 * TF/SF source; /root/reference/imd_fdtd.c:63-126,128-306,594-859) compiles from its own source file with -DFDTD;
 * its only caller, the second block of calc_ttm in imd_ttm_3D.c (:163-179), does NOT: with -DFDTD that file stops at a
 * typo in ttm_create_mpi_datatypes (imd_ttm_3D.c:3022, `blockcountselements+=10,` -- a comma for a semicolon), and
 * the rules of this repo forbid patching or copying reference sources.  So the solver is pinned at FUNCTION level:
 * this translation unit instantiates the reference's globals (it is the MAIN of this little program, like imd.c is
 * of IMD), allocates the FD lattice rows l1/l2 the way init_ttm does, fills the cells' material state with
 * deterministic inputs, then calls the reference's init_fdtd() and do_fdtd() in the order of calc_ttm's loop
 * (:163-179: per FDTD sub-step do_fdtd(t_SI); t_SI += tau_FDTD, laser_timefun fixed per MD step, :98) and dumps
 * the fields as raw doubles.  Nothing of the reference is replaced: no stand-in header, library or function.
 *
 * The Drude-Lorentz parameters DL[0..4] of a cell come from fitDL's tricubic tables in the reference (DL1..5.txt, not
 * shipped); here they are inputs (the constant-parameter variant fitDL itself keeps in a comment, imd_ttm_3D.c:2921-
 * 2937), DL[5] = plasma frequency from the cell's electron density as fitDL computes it (:2960).
 *
 * usage: ref_probe_fdtd <out.bin> nx hx_A natoms_lo natoms_hi I0 lambda sigma_t t0 srcx_m bw Sc nmd dt_md_imd
 */
#define MAIN
#include "imd.h"

static void wd(FILE *f, double v) { fwrite(&v, sizeof(double), 1, f); }
static void wi(FILE *f, int v) { fwrite(&v, sizeof(int), 1, f); }

static ttm_Element ***alloc_lattice(int nx, int ny, int nz, ttm_Element **flat)
{
  int i, j;
  ttm_Element ***l;
  *flat = (ttm_Element *) calloc((size_t) nx * ny * nz, sizeof(ttm_Element));      /* init_ttm, imd_ttm_3D.c:1066-1090 */
  l = (ttm_Element ***) malloc(nx * sizeof(ttm_Element **));
  for (i = 0; i < nx; i++) {
    l[i] = (ttm_Element **) malloc(ny * sizeof(ttm_Element *));
    for (j = 0; j < ny; j++) l[i][j] = *flat + ((size_t) i * ny + j) * nz;
  }
  return l;
}

static void dump(FILE *f, const char *tag, int nx)
{
  char t[16];
  int i;
  memset(t, 0, sizeof(t)); strncpy(t, tag, 15);
  fwrite(t, 1, 16, f);
  wd(f, t_SI); wd(f, laser_timefun);
  for (i = 0; i < nx; i++) {
    const ttm_Element *e = &l1[i][1][1], *e2 = &l2[i][1][1];
    wd(f, e->Ezx); wd(f, e->Hy); wd(f, e->Jzx); wd(f, e->Jlzx); wd(f, e->Pzx); wd(f, e->source); wd(f, e->sigmay);
    wd(f, e2->Jzx); wd(f, e2->Jlzx);
  }
}

int main(int argc, char **argv)
{
  int nx, lo, hi, nmd, i, s, sub;
  double hx, dtmd;
  FILE *f;
  MPI_Init(&argc, &argv);
  if (argc < 15) { fprintf(stderr, "usage: see the header of oracle/ref_probe_fdtd.c\n"); return 2; }
  nx = atoi(argv[2]); hx = atof(argv[3]); lo = atoi(argv[4]); hi = atoi(argv[5]);
  I0 = atof(argv[6]); lambda = atof(argv[7]); laser_sigma_t = atof(argv[8]); laser_t_0 = atof(argv[9]);
  srcx = atof(argv[10]); bw = atoi(argv[11]); Sc = atof(argv[12]); nmd = atoi(argv[13]); dtmd = atof(argv[14]);
  /* one rank, open x: what setup_mpi_topology / init_ttm leave behind for cpu_dim 1 1 1, pbc_dirs 0 1 1 */
  myid = 0; num_cpus = 1; cpugrid = MPI_COMM_WORLD;
  cpu_dim.x = cpu_dim.y = cpu_dim.z = 1; my_coord.x = my_coord.y = my_coord.z = 0;
  pbc_dirs.x = 0; pbc_dirs.y = 1; pbc_dirs.z = 1;
  local_fd_dim.x = nx; local_fd_dim.y = 3; local_fd_dim.z = 3;
  global_fd_dim.x = nx - 2; global_fd_dim.y = 1; global_fd_dim.z = 1;
  fd_h.x = hx; fd_h.y = 60.0; fd_h.z = 60.0;
  fd_min_atoms = 5;
  timestep = dtmd;
  l1 = alloc_lattice(nx, 3, 3, &lattice1);
  l2 = alloc_lattice(nx, 3, 3, &lattice2);
  if (init_fdtd() != 0) { fprintf(stderr, "init_fdtd failed\n"); return 1; }
  /* material state of the cells: matter in [lo, hi) (inner index i-1), vacuum elsewhere; a density / temperature ramp
   * so that every cell has its own plasma frequency */
  for (i = 1; i < nx - 1; i++) {
    ttm_Element *e = &l1[i][1][1];
    int ig = i - 1;
    if (ig >= lo && ig < hi) {
      double x = (double) (ig - lo) / (double) (hi - lo);
      e->natoms = 400; e->dens = 2700.0 * (1.0 - 0.15 * x); e->temp = 0.03 + 0.4 * x * (1.0 - x); e->md_temp = 0.03;
      e->ne = 3.0 * e->dens / (26.9815 * AMU);
      e->DL[0] = 2.73; e->DL[1] = 1.1174e+15; e->DL[2] = 7.6595e+15; e->DL[3] = 2.4024e+15; e->DL[4] = 4.5199e+14;
      e->DL[5] = sqrt(e->ne * ECHARGE * ECHARGE / ECONST / EMASS);
    } else e->natoms = 0;
    l2[i][1][1] = *e;
  }
  f = fopen(argv[1], "wb");
  fwrite("IMDFDTD1", 1, 8, f);
  wi(f, nx); wi(f, bw); wi(f, srcx_int); wi(f, containssrc);
  wd(f, dt_FDTD); wd(f, fd_dx); wd(f, E0); wd(f, E0_init); wd(f, Nlambdax); wd(f, Imp0); wd(f, laser_sigma_t_squared);
  wd(f, omega_laser);
  for (i = 0; i < nx; i++) {
    const ttm_Element *e = &l1[i][1][1];
    wi(f, e->natoms); wd(f, e->dens); wd(f, e->temp); wd(f, e->ne);
    for (s = 0; s < 6; s++) wd(f, e->DL[s]);
  }
  dump(f, "init", nx);
  t_SI = 0.0;
  for (s = 0; s < nmd; s++) {
    /* calc_ttm, imd_ttm_3D.c:96-122 */
    int fdtd_substeps;
    double tau_FDTD;
    laser_timefun = E0 * exp(-0.5 * pow(t_SI + dt_FDTD - laser_t_0, 2.0) / laser_sigma_t_squared);
    laser_active = !(laser_timefun < E0_init * 1e-4);
    fdtd_substeps = (int) ((timestep * 10.18 * 1.0e-15) / (dt_FDTD) + 1.0);
    tau_FDTD = timestep * 10.18 * 1.0e-15 / ((double) fdtd_substeps);
    wi(f, fdtd_substeps); wi(f, laser_active);
    for (sub = 0; sub < fdtd_substeps; sub++) {                      /* :163-179 */
      if (laser_active) do_fdtd(t_SI);
      t_SI += tau_FDTD;
      if (s == 0 && (sub == 0 || sub == 1 || sub == 7)) { char tag[16]; sprintf(tag, "s0_sub%d", sub); dump(f, tag, nx); }
    }
    { char tag[16]; sprintf(tag, "md%d", s); dump(f, tag, nx); }
  }
  fwrite("ENDFDTD1", 1, 8, f);
  fclose(f);
  MPI_Finalize();
  return 0;
}
