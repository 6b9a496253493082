/* ttm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See ttm_oracle.c. */
#ifndef TTM_ORACLE_H
#define TTM_ORACLE_H
#include <stddef.h>
#include "imd_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int nx, ny; double xmin, xmax, ymin, ymax; double *z; unsigned char *diag; } orc_eos;   /* diag: triangulation map or NULL */

/* SoA restatement of the ttm_Element lattices l1/l2 (types.h:441-492): fields
 * that do_DIFF double-buffers (temp, U) are ping-pong pairs, `cur` = l1. */
typedef struct {
  int dim[3], ext[3], pbc[3];
  size_t ncell;
  int fd_min_atoms, cur;
  double h[3], vol, neighvol, atomic_weight, atomic_charge, omega_laser, Temin, max_dt_ttm;
  int *natoms, *natoms_old;
  double *temp[2], *U[2];
  double *xi, *md_temp, *source, *vcom, *dens, *fd_k, *fd_g, *Z, *Ce, *ne;
  orc_eos cve, ee;
  int *flux;                 /* [8*ncell]: atoms that entered the FD cell since the last update_fd, by direction (:335-370) */
  int adv;                   /* 1: orc_ttm_calc runs do_ADV after update_fd as calc_ttm does (:57-60) */
  int laser_active;          /* set by orc_ttm_laser_rescale like the global of that name (imd_laser.c:744-747) */
  double eabs;               /* Eabs_local: absorbed laser energy [eV] accumulated by do_DIFF (:1647-1651) */
} orc_ttm;

orc_ttm *orc_ttm_new(const int md_dim[3], const int fd_ext[3], const double fd_h[3], const int pbc[3],
                     int fd_min_atoms, double atomic_weight, double atomic_charge, double r2_cut_pair);
void orc_ttm_free(orc_ttm *t);
void orc_ttm_set_eos(orc_ttm *t, int which /*0 cve(r,T), 1 ee(r,sqrt T)*/, int nx, int ny, double xmin, double xmax,
                     double ymin, double ymax, const double *z);
double orc_eos_lookup(const orc_eos *e, double x, double y);
void orc_ttm_set_eos_diag(orc_ttm *t, int which, const unsigned char *diag);
double orc_eos_cve(const orc_ttm *t, double r, double tK);
double orc_eos_ee(const orc_ttm *t, double r, double tK);
double orc_eos_te_from_r_ee(const orc_ttm *t, double r, double e);
double orc_mean_charge(double temp, double rho, double z0, double am);
double orc_get_gamma(double Te, double Ti, double Ne, double Z, double omega_laser);
double orc_get_kappa(double Te, double Ti, double Ne, double Z, double omega_laser);
void orc_wr_tmm_rates(double omega_las, double omega_coul, double Te, double Ti, double Z, double Ne, double *Ncr, double *TF,
                      double *nu_met, double *nu_max, double *nu_pl);
void orc_ttm_update_fd(orc_ttm *t, orc_sys *s, int steps);
void orc_ttm_adv(orc_ttm *t, double tau, int steps);
/* laser_rescale_ttm (imd_laser.c:715-788), non-LASERYZ branch; the globals it reads are the arguments */
void orc_ttm_laser_rescale(orc_ttm *t, const int dir[3], double offset, double mu, double p_peak, double t_0,
                           double sigma_t_squared, double time);
void orc_ttm_cell_activation(orc_ttm *t);
void orc_ttm_fillmesh(orc_ttm *t, int steps);
void orc_ttm_fill_ghosts(orc_ttm *t);
void orc_ttm_diff(orc_ttm *t, double tau, int diff_substeps);
void orc_ttm_cfl(orc_ttm *t, int steps, double timestep, int fd_n_timesteps);
void orc_ttm_calc(orc_ttm *t, orc_sys *s, int steps, int diff_substeps);
void orc_move_atoms_ttm(orc_sys *s, const orc_ttm *t);
/* the reference's own 1D variant (imd_ttm.c): oracle/ttm1d_oracle.c; the struct is defined there (and mirrored in pyoracle.py) */
#ifdef __cplusplus
}
#endif
#endif
