/* fdtd_oracle.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference's 1D Maxwell solver (row f2 of SURVEY 8:
 * config 5's laser source).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Follows /root/reference/imd_fdtd.c: init_fdtd :594-763 (1D branch), init_pml :765-859 (1D branch), fdtd_softsource
 * :63-126 (1D branch), do_fdtd :128-306; the calling sequence is calc_ttm's (imd_ttm_3D.c:96-122,163-179).
 * One row of cells along x (index 0 and nx-1 are the ghost cells of a one-rank, open-x lattice: never updated, fields
 * stay zero -- fdtd_comm_ghost_cells :1074-1126 does nothing for cpu_dim.x = 1, pbc_dirs.x = 0).
 * Pinned against the reference's own functions through oracle/ref_probe_fdtd.c -> tests/golden/fdtd_1d.npz. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fdtd_oracle.h"

#define EPS0 8.841941282883075e-12      /* globals.h:497-499 (the reference's values, not CODATA) */
#define MU0 1.256637061435917e-06
#define C0 3.0e8

static double mind(double a, double b) { return a < b ? a : b; }

orc_fdtd *orc_fdtd_new(int nx, double fd_h_x_angstrom, double I0, double lambda, double sigma_t, double t0, double srcx,
                       int bw, double Sc, int fd_min_atoms)
{
  orc_fdtd *f = (orc_fdtd *) calloc(1, sizeof(orc_fdtd));
  int i;
  double freq, sigma_max_x, sigma_max_y, bf3, refl_coeff = 1.0e-12, gradingorder = 8.0;
  f->nx = nx; f->fd_min_atoms = fd_min_atoms;
  f->Ezx = (double *) calloc(nx, 8); f->Hy = (double *) calloc(nx, 8); f->Jzx = (double *) calloc(nx, 8);
  f->Jlzx = (double *) calloc(nx, 8); f->Pzx = (double *) calloc(nx, 8); f->source = (double *) calloc(nx, 8);
  f->sigmay = (double *) calloc(nx, 8); f->Jzx_old = (double *) calloc(nx, 8); f->Jlzx_old = (double *) calloc(nx, 8);
  f->natoms = (int *) calloc(nx, sizeof(int)); f->DL = (double *) calloc(6 * (size_t) nx, 8);
  /* init_fdtd, :594-652 */
  f->Imp0 = sqrt(MU0 / EPS0);
  f->fd_dx = fd_h_x_angstrom / 1e10;
  f->Sc_FDTD = mind(1.0, Sc);
  f->dt = f->Sc_FDTD * f->fd_dx / C0;
  freq = C0 / lambda;
  f->omega_laser = 2.0 * M_PI * freq;
  f->Nlambdax = C0 / (freq * f->fd_dx);
  f->E0 = sqrt(2.0 * I0 * f->Imp0);
  f->E0_init = f->E0;
  f->t0 = t0; f->sigma_t_squared = sigma_t * sigma_t;
  f->bw = bw > 5 ? bw : 5;                                            /* init_pml :770 (applied after srcx_int, below) */
  f->srcx_int = (int) round(srcx / f->fd_dx);
  if (f->srcx_int < bw + 10) f->srcx_int = bw + 10;                   /* :648, with the bw of the parameter file */
  f->containssrc = (f->srcx_int <= nx - 1 - 1 && f->srcx_int >= 0) ? 1 : 0;       /* :649-651, my_coord.x = 0 */
  /* init_pml, 1D branch :765-858: conductivity profile of the two absorbing layers, vacuum assumed there */
  sigma_max_x = (-log10(refl_coeff) * (gradingorder + 1.0) * EPS0 * C0) / (2.0 * (double) f->bw * f->fd_dx);
  sigma_max_y = sigma_max_x;
  bf3 = (1.0 * sigma_max_y) / (pow(f->bw, gradingorder) * (gradingorder + 1.0));
  for (i = 1; i < nx - 1; i++) {
    const int ig = i - 1, gx = nx - 2;
    double idist;
    if (ig >= gx - f->bw) {
      idist = (double) (f->bw - ((gx - 1) - ig));
      f->sigmay[i] = bf3 * (pow(idist + 0.5, gradingorder + 1.0) - pow(idist - 0.5 * (ig > gx - 1 - f->bw), gradingorder + 1.0));
    } else if (ig <= f->bw) {
      idist = (double) (f->bw - ig);
      f->sigmay[i] = bf3 * (pow(idist + 0.5, gradingorder + 1.0) - pow(idist - 0.5 * (ig < f->bw), gradingorder + 1.0));
    }
  }
  return f;
}

void orc_fdtd_free(orc_fdtd *f)
{
  if (!f) return;
  free(f->Ezx); free(f->Hy); free(f->Jzx); free(f->Jlzx); free(f->Pzx); free(f->source); free(f->sigmay);
  free(f->Jzx_old); free(f->Jlzx_old); free(f->natoms); free(f->DL); free(f);
}

/* what calc_ttm evaluates once per MD step before the loop (imd_ttm_3D.c:96-122): the envelope at the start of the
 * step, whether the solver runs at all, the number of FDTD sub-steps and their length */
int orc_fdtd_begin_md_step(orc_fdtd *f, double t_SI, double timestep_imd, double *tau_FDTD)
{
  int n;
  f->laser_timefun = f->E0 * exp(-0.5 * pow(t_SI + f->dt - f->t0, 2.0) / f->sigma_t_squared);
  f->laser_active = !(f->laser_timefun < f->E0_init * 1e-4);
  n = (int) ((timestep_imd * 10.18 * 1.0e-15) / (f->dt) + 1.0);
  if (tau_FDTD) *tau_FDTD = timestep_imd * 10.18 * 1.0e-15 / ((double) n);
  return n;
}

/* one do_fdtd(t), :128-306 */
void orc_fdtd_step(orc_fdtd *f, double t)
{
  const double dt = f->dt, dx = f->fd_dx;
  int i;
  /* fdtd_softsource, 1D branch :78-83,114-122 */
  {
    const double ezy2 = (dt / dx) / (EPS0);
    const double sine = sin(2.0 * M_PI * (C0 / (dx * f->Nlambdax)) * (t + dt));
    if (f->containssrc == 1) {
      const int ilocal = f->srcx_int + 1;
      const double Hinc = f->laser_timefun * sine / f->Imp0;
      f->Ezx[ilocal] += ezy2 * Hinc;
    }
  }
  /* E update n -> n+1, :178-231 */
  for (i = 1; i < f->nx - 1; i++) {
    if (f->natoms[i] >= f->fd_min_atoms) {
      const double epsinf = f->DL[6 * i + 0];
      const double Cb = dt / EPS0 / epsinf;
      f->Jzx_old[i] = f->Jzx[i]; f->Jlzx_old[i] = f->Jlzx[i];
      f->Ezx[i] = f->Ezx[i] + Cb * (f->Hy[i] - f->Hy[i - 1]) / dx - Cb * f->Jzx[i] - Cb * f->Jlzx[i];
      f->Pzx[i] = f->Pzx[i] + dt * f->Jlzx[i];
    } else {
      const double eps = EPS0, sigmay = f->sigmay[i];
      const double ezx1 = (eps - 0.5 * dt * sigmay) / (eps + 0.5 * dt * sigmay);
      const double ezx2 = (dt / dx) / (eps + 0.5 * dt * sigmay);
      f->Ezx[i] = ezx1 * f->Ezx[i] + ezx2 * (f->Hy[i] - f->Hy[i - 1]);
    }
  }
  /* H update to n+3/2, polarisation currents, dissipated power, :236-286 */
  for (i = 1; i < f->nx - 1; i++) {
    const double mu = MU0, eps = EPS0, sigmay = f->sigmay[i];
    const double sigmastary = sigmay * mu / eps;
    const double hy1 = (mu - 0.5 * dt * sigmastary) / (mu + 0.5 * dt * sigmastary);
    const double hy2 = (dt / dx) / (mu + 0.5 * dt * sigmastary);
    f->Hy[i] = hy1 * f->Hy[i] + hy2 * (f->Ezx[i + 1] - f->Ezx[i]);
    if (f->natoms[i] >= f->fd_min_atoms) {
      const double *DL = f->DL + 6 * i;
      const double omega_plasma = DL[5], gamma_p = DL[1], omegapl_L = DL[2], Omega0_L = DL[3], Gamma_L = DL[4];
      const double CD1 = (2 - dt * gamma_p) / (2 + dt * gamma_p), CL1 = (2 - dt * Gamma_L) / (2 + dt * Gamma_L);
      const double CD2 = 2 * dt / (2 + dt * gamma_p), CL2 = 2 * dt / (2 + dt * Gamma_L);
      const double omegapl_D_sq = omega_plasma * omega_plasma, omegapl_L_sq = omegapl_L * omegapl_L;
      const double omega0_L_sq = Omega0_L * Omega0_L;
      double Jz_present, Jlz_present, qe_Drude, qe_Lorentz;
      f->Jzx[i] = CD1 * f->Jzx[i] + CD2 * (omegapl_D_sq * EPS0 * f->Ezx[i]);
      f->Jlzx[i] = CL1 * f->Jlzx[i] + CL2 * (omegapl_L_sq * EPS0 * f->Ezx[i] - omega0_L_sq * f->Pzx[i]);
      Jz_present = (f->Jzx[i] + f->Jzx_old[i]) * 0.5;
      Jlz_present = 0.5 * (f->Jlzx[i] + f->Jlzx_old[i]);
      qe_Drude = gamma_p * (Jz_present * Jz_present) / EPS0 / omegapl_D_sq;
      qe_Lorentz = Gamma_L / EPS0 / omegapl_L_sq * (Jlz_present * Jlz_present);
      f->source[i] = 6.3538562638e-26 * (qe_Drude + qe_Lorentz);          /* W/m^3 -> IMD units */
    }
  }
  /* total-field / scattered-field correction: no wave towards -x, :293-303 */
  if (f->containssrc) {
    const int ilocal = f->srcx_int + 1;
    const double sine = sin(2.0 * M_PI * (C0 / (dx * f->Nlambdax)) * t);
    const double temporal = exp(-0.5 * pow(t - f->t0, 2.0) / f->sigma_t_squared);
    const double Einc = sine * temporal * f->E0;
    const double hy2 = (dt / dx) / (MU0);
    f->Hy[ilocal - 1] -= hy2 * Einc;
  }
}
