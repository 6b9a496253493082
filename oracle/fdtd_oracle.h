/* TEST INFRASTRUCTURE -- see fdtd_oracle.c */
#ifndef FDTD_ORACLE_H
#define FDTD_ORACLE_H
typedef struct {
  int nx, fd_min_atoms, bw, srcx_int, containssrc, laser_active;
  double Imp0, fd_dx, Sc_FDTD, dt, omega_laser, Nlambdax, E0, E0_init, t0, sigma_t_squared, laser_timefun;
  double *Ezx, *Hy, *Jzx, *Jlzx, *Pzx, *source, *sigmay, *Jzx_old, *Jlzx_old, *DL;
  int *natoms;
} orc_fdtd;
orc_fdtd *orc_fdtd_new(int nx, double fd_h_x_angstrom, double I0, double lambda, double sigma_t, double t0, double srcx,
                       int bw, double Sc, int fd_min_atoms);
void orc_fdtd_free(orc_fdtd *f);
int orc_fdtd_begin_md_step(orc_fdtd *f, double t_SI, double timestep_imd, double *tau_FDTD);
void orc_fdtd_step(orc_fdtd *f, double t);
#endif
