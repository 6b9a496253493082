/*
 * ttm1d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's OWN 1D two-temperature model, imd_ttm.c (the file the reference's Makefile builds for ttm
 * targets, Makefile:185; -DTTM1D): atoms binned into slabs by POSITION, slab sums held as global arrays on every rank, the
 * activity criterion natoms >= fd_min_atoms && dens > RHOMIN, a static virtual lattice behind the sample, a sub-step count
 * from the stability limit.  Citations relative to /root/reference/imd_ttm.c unless a file is named.
 *
 * The reference cuts the row of FD cells into num_cpus pieces by rank id (:277-283) and keeps rank-local ghost cells; here the
 * row is ONE global array (index g = i_global, ghost cells at -1 and nx).  Pinned against the reference's own functions through
 * tests/golden/ttm1d_slab.npz (probe: oracle/ref_probe_ttm1d.c, a two-rank run: the reference's ttm_fill_ghost_layers and
 * do_COMMFLUX have no one-rank branch).  Rank-layout behaviour of the reference that is reproduced: every cell on the rank that
 * owns the last active cell takes the virtual lattice's first cell as its +x neighbour (:1535-1540; `nranks` says how the
 * reference was cut).  Not reproduced (undefined or layout-dependent, cannot occur in the fixtures): reads of natomsglobal[-1] /
 * [nx] when the sample touches the box ends (:441,452, 1232-1235); freshly activated cells AT a rank boundary see a ghost
 * temperature that update_fd has just zeroed (:245-246).  do_tmm (imd_tmm.c) and -DELECPRESS are at the end of this
 * file.  Left out: COLRAD, BALLISTIC, FEG EOS mode (EOSMODE 0).
 */
#include "ttm_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RHOMIN 2.0                      /* :69 */
static const double AMU = 1.660539040200000e-27, J2EV = 6.2415090744607626077762409809304458998869658961709711e18;

typedef struct {
  int nx, nranks, vlatdim, vlatbuffer, fd_min_atoms, fd_n_timesteps, cur, vcur, last_active, vlat_proc, diff_substeps, adv;
  double hx, vol, neighvol, aw, ac, vlatdens, Temin, max_dt_ttm, omega_laser;
  /* lattice row, index g + 1 (ghost cells at 0 and nx + 1) */
  int *natoms, *natoms_old;
  double *temp[2], *U[2], *xi, *md_temp, *source, *dens, *fd_k, *fd_g, *Z, *Ce, *ne;
  /* the global slab arrays of update_fd (:170-230), index g */
  int *fluxl, *fluxr, *totneigh, *natomsglobal;
  double *vcom[3], *xiarr;
  /* virtual lattice (:1013-1024) */
  double *vtemp[2], *vmd[2], *vk, *vg, *vCe, *vZ, *vne;
  orc_ttm *eos;                         /* holder of the two EOS tables */
  /* transfer-matrix laser source (imd_tmm.c), see the end of this file */
  int tmm_on, laser_active, ecut, eps_bb_rows, K12_rows;
  double I0, lambda, t0, sigma_t, t1, sigma_t1, abs_threshold, laser_threshold, k0, tmm_time, I_t, refl, trans, absorbed, eabs;
  double *eps_re, *eps_im, *qabs, *eps_bb, *K12;
  int *tmm_active;
  double complex *km;
  orc_eos phase;
  /* electron pressure (-DELECPRESS): do_electronic_pressure (:553-623) and move_atoms_ttm's extra force */
  int ep_on;
  double box_xx;
  double *epress, *epress_deriv;
  orc_eos pe;
} orc_ttm1d;

static void *za(size_t n, size_t s) { void *p = calloc(n ? n : 1, s); if (!p) abort(); return p; }

orc_ttm1d *orc_ttm1d_new(int nx, double box_xx, double box_yy, double box_zz, int nranks, int vlatdim, int vlatbuffer, double vlatdens,
                         int fd_min_atoms, int fd_n_timesteps, double atomic_weight, double atomic_charge, double r2_cut_pair,
                         double timestep, orc_ttm *eos_holder)
{
  orc_ttm1d *t = za(1, sizeof *t);
  int q, i;
  t->nx = nx; t->nranks = nranks; t->vlatdim = vlatdim; t->vlatbuffer = vlatbuffer; t->vlatdens = vlatdens;
  t->fd_min_atoms = fd_min_atoms; t->fd_n_timesteps = fd_n_timesteps; t->aw = atomic_weight; t->ac = atomic_charge;
  t->hx = box_xx / nx; t->vol = t->hx * box_yy * box_zz;                                    /* init_ttm :972-976 */
  t->box_xx = box_xx;
  t->neighvol = pow(sqrt(r2_cut_pair), 3.0) * 4.0 / 3.0 * M_PI;                           /* :979 */
  t->max_dt_ttm = timestep / (double) fd_n_timesteps;                                     /* :978 */
  t->Temin = 1e-3; t->eos = eos_holder; t->last_active = -5000; t->vlat_proc = 0;
  t->natoms = za(nx + 2, sizeof(int)); t->natoms_old = za(nx + 2, sizeof(int));
  for (q = 0; q < 2; q++) { t->temp[q] = za(nx + 2, 8); t->U[q] = za(nx + 2, 8); t->vtemp[q] = za(vlatdim, 8); t->vmd[q] = za(vlatdim, 8); }
  t->xi = za(nx + 2, 8); t->md_temp = za(nx + 2, 8); t->source = za(nx + 2, 8); t->dens = za(nx + 2, 8); t->fd_k = za(nx + 2, 8);
  t->fd_g = za(nx + 2, 8); t->Z = za(nx + 2, 8); t->Ce = za(nx + 2, 8); t->ne = za(nx + 2, 8);
  t->fluxl = za(nx, sizeof(int)); t->fluxr = za(nx, sizeof(int)); t->totneigh = za(nx, sizeof(int)); t->natomsglobal = za(nx, sizeof(int));
  for (q = 0; q < 3; q++) t->vcom[q] = za(nx, 8);
  t->xiarr = za(nx, 8);
  t->vk = za(vlatdim, 8); t->vg = za(vlatdim, 8); t->vCe = za(vlatdim, 8); t->vZ = za(vlatdim, 8); t->vne = za(vlatdim, 8);
  for (i = 0; i < vlatdim; i++) for (q = 0; q < 2; q++) { t->vtemp[q][i] = 0.0264; t->vmd[q][i] = 0.0264; }   /* :1017-1023 */
  return t;
}
void orc_ttm1d_free(orc_ttm1d *t)
{
  int q;
  void *p[] = {t->natoms, t->natoms_old, t->xi, t->md_temp, t->source, t->dens, t->fd_k, t->fd_g, t->Z, t->Ce, t->ne, t->fluxl, t->fluxr,
               t->totneigh, t->natomsglobal, t->xiarr, t->vk, t->vg, t->vCe, t->vZ, t->vne};
  for (q = 0; q < (int) (sizeof p / sizeof p[0]); q++) free(p[q]);
  for (q = 0; q < 2; q++) { free(t->temp[q]); free(t->U[q]); free(t->vtemp[q]); free(t->vmd[q]); }
  for (q = 0; q < 3; q++) free(t->vcom[q]);
  if (t->ep_on) { free(t->epress); free(t->epress_deriv); free(t->pe.z); }
  if (t->tmm_on) { free(t->eps_re); free(t->eps_im); free(t->qabs); free(t->tmm_active); free(t->km); free(t->eps_bb); free(t->K12); free(t->phase.z); }
  free(t);
}
static int active(const orc_ttm1d *t, int c) { return t->natoms[c] >= t->fd_min_atoms && t->dens[c] > RHOMIN; }          /* e.g. :662 */
static int active_ge(const orc_ttm1d *t, int c) { return !(t->natoms[c] < t->fd_min_atoms || t->dens[c] < RHOMIN); }     /* do_DIFF / CFL neighbours, :1525-1527 */
static int slab(const orc_ttm1d *t, double x)
{
  int g = (int) (x / t->hx);                                                              /* :296-298 */
  if (g > t->nx - 1) g = t->nx - 1;
  if (g < 0) g = 0;
  return g;
}

/* update_fd (:161-548).  Atoms as flat arrays; fdi = the slab of each atom at the previous call (cell.fdi), updated here. */
void orc_ttm1d_update_fd(orc_ttm1d *t, int steps, int n, const double *ort, const int *sorte, const double *impuls, const double *masse,
                         const int *numneighs, int *fdi)
{
  const int nx = t->nx;
  int g, a, d;
  double *mdt = za(nx, 8);
  for (g = 0; g < nx; g++) { t->natomsglobal[g] = t->totneigh[g] = t->fluxl[g] = t->fluxr[g] = 0; t->vcom[0][g] = t->vcom[1][g] = t->vcom[2][g] = 0; t->xiarr[g] = 0; }
  t->temp[t->cur][0] = t->temp[t->cur][nx + 1] = 0;                                       /* :245-246 (the global ends) */
  for (g = 1; g <= nx; g++) {                                                              /* :252-266 */
    t->source[g] = 0; t->natoms_old[g] = t->natoms[g]; t->natoms[g] = 0; t->xi[g] = 0; t->dens[g] = 0;
  }
  for (a = 0; a < n; a++) {                                                                /* :271-324 */
    if (sorte && sorte[a] != 0) continue;
    g = slab(t, ort[3 * a]);
    t->natomsglobal[g]++; t->totneigh[g] += numneighs[a];
    for (d = 0; d < 3; d++) t->vcom[d][g] += impuls[3 * a + d];
    if (steps > 0 && fdi[a] != g) { if (fdi[a] > g) t->fluxr[g]++; else t->fluxl[g]++; }   /* :307-318 */
    fdi[a] = g;
  }
  for (g = 0; g < nx; g++) {                                                               /* :350-372 */
    const double totmass = (double) t->natomsglobal[g] * t->aw;
    for (d = 0; d < 3; d++) t->vcom[d][g] = totmass > 0 ? t->vcom[d][g] / totmass : 0.0;
  }
  for (a = 0; a < n; a++) {                                                                /* :376-403 */
    if (sorte && sorte[a] != 0) continue;
    g = slab(t, ort[3 * a]);
    for (d = 0; d < 3; d++) { const double u = impuls[3 * a + d] / masse[a] - t->vcom[d][g]; mdt[g] += masse[a] * (u * u); }
  }
  t->last_active = steps == 0 ? -5000 : t->last_active;
  {
    int last_local = -5000;
    for (g = 0; g < nx; g++) {                                                             /* :409-470 */
      const int c = g + 1;
      t->natoms[c] = t->natomsglobal[g];
      if (t->natoms[c] > 0) {
        t->dens[c] = ((double) t->totneigh[g]) / ((double) t->natoms[c]) * t->aw / t->neighvol * 1660.53907;
        if (t->dens[c] == 0) t->dens[c] = (double) t->natoms[c] * t->aw / t->vol * 1660.53907;
        t->md_temp[c] = mdt[g] / (3.0 * (double) t->natoms[c]);
        if (active(t, c) && g - t->vlatbuffer > last_local) last_local = g - t->vlatbuffer;   /* :431-435 */
      } else { t->dens[c] = 0.0; t->md_temp[c] = 0.0; t->natoms[c] = 0; }
    }
    if (steps < 1)                                                                         /* :493-503 */
      for (g = 1; g <= nx; g++) { t->natoms_old[g] = t->natoms[g]; if (active(t, g)) t->temp[0][g] = t->temp[1][g] = t->md_temp[g]; }
    if (steps == 0) t->last_active = last_local;                                           /* VLATTICE_STATIC, :515-519 */
  }
  t->vlat_proc = (int) (((double) t->last_active * t->nranks) / ((double) nx));          /* :522 */
  for (g = 0; g < nx; g++) if (g > t->last_active) t->natoms[g + 1] = -1;                  /* :537-542 */
  free(mdt);
}

/* do_ADV (:1208-1309) with do_COMMFLUX folded in (one global row: the neighbours' U is simply there; the ghost cells at
 * the global ends keep U = 0 like the end ranks' ghost cells, :858-953) */
void orc_ttm1d_adv(orc_ttm1d *t, double tau, int steps)
{
  const int nx = t->nx, c0 = t->cur, c1 = c0 ^ 1;
  int g;
  if (steps < 1) return;
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    const double Nold = (double) t->natoms_old[c], Nnew = (double) t->natoms[c];
    if (Nnew > 0) {
      const double fr_here = t->fluxr[g], fl_next = g + 1 < nx ? t->fluxl[g + 1] : 0.0, fl_here = t->fluxl[g], fr_prev = g > 0 ? t->fluxr[g - 1] : 0.0;
      t->U[c1][c] = t->U[c0][c] * Nold / Nnew + tau * (+fr_here * t->U[c0][c + 1] - fl_next * t->U[c0][c] + fl_here * t->U[c0][c - 1] - fr_prev * t->U[c0][c]) / Nnew;
      if (active(t, c)) t->temp[c1][c] = orc_eos_te_from_r_ee(t->eos, t->dens[c], t->U[c1][c] / (26.9815 * AMU * J2EV)) / 11604.5;
    } else if (Nnew < 1) { t->U[c1][c] = 0.0; t->temp[c1][c] = 0.0; }
  }
  t->cur ^= 1;                                                                             /* l1 <-> l2, :1304-1306 */
}

/* do_cell_activation (:1311-1425) */
void orc_ttm1d_cell_activation(orc_ttm1d *t)
{
  const int nx = t->nx, c0 = t->cur;
  int g;
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    if (t->natoms_old[c] >= t->fd_min_atoms && t->natoms[c] < t->fd_min_atoms && t->dens[c] > RHOMIN) t->xi[c] = 0.0;
    else if (t->natoms_old[c] < t->fd_min_atoms && t->natoms[c] >= t->fd_min_atoms && t->dens[c] > RHOMIN) {
      if (g > t->last_active) continue;
      if (isnan(t->temp[c0][c]) || t->temp[c0][c] <= t->Temin) {
        int nn = 0; double e = 0.0;
        if (t->natoms[c + 1] >= t->fd_min_atoms && t->dens[c + 1] >= RHOMIN) { e += t->temp[c0][c + 1] * t->temp[c0][c + 1]; nn++; }
        if (t->natoms[c - 1] >= t->fd_min_atoms && t->dens[c - 1] >= RHOMIN) { e += t->temp[c0][c - 1] * t->temp[c0][c - 1]; nn++; }
        if (nn != 0) {
          t->temp[c0][c] = sqrt(e / (double) nn);
          if (t->temp[c0][c] < t->Temin) t->temp[c0][c] = t->md_temp[c];
        } else t->temp[c0][c] = t->md_temp[c];
        t->temp[c0 ^ 1][c] = t->temp[c0][c];
        t->U[0][c] = t->U[1][c] = orc_eos_ee(t->eos, t->dens[c], t->temp[c0][c] * 11604.5) * 26.9815 * AMU * J2EV;
      }
    }
  }
}

/* do_FILLMESH (:629-850): wide-range properties of the active cells and of the virtual lattice */
void orc_ttm1d_fillmesh(orc_ttm1d *t, int steps)
{
  const int nx = t->nx, c0 = t->cur;
  const double om = t->laser_active ? t->omega_laser : 0.0;       /* bMax (:2207-2213): omega_las only while the laser is active */
  int g, i;
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    if (g > t->last_active) continue;                                                     /* :649-652 */
    if (active(t, c)) {
      const double TeK = t->temp[c0][c] * 11604.5;
      t->Z[c] = orc_mean_charge(TeK, t->dens[c], t->ac, t->aw);
      t->ne[c] = t->Z[c] * t->dens[c] / (t->aw * AMU);
      { double z = orc_eos_cve(t->eos, t->dens[c], TeK); t->Ce[c] = z; }
      t->fd_k[c] = orc_get_kappa(t->temp[c0][c], t->md_temp[c], t->ne[c], t->Z[c], om);
      t->fd_g[c] = orc_get_gamma(t->temp[c0][c], t->md_temp[c], t->ne[c], t->Z[c], om);
      if (steps < 1) t->U[0][c] = t->U[1][c] = orc_eos_ee(t->eos, t->dens[c], TeK) * 26.9815 * AMU * J2EV;      /* :760-764 */
    } else { t->fd_k[c] = t->fd_g[c] = t->Z[c] = t->ne[c] = t->Ce[c] = 0.0; t->temp[c0][c] = 0.0; }          /* :806-814 */
  }
  for (i = 0; i < t->vlatdim; i++) {                                                      /* :819-843 (on the rank that owns the last active cell) */
    const double Te = t->vtemp[t->vcur][i], Ti = t->vmd[t->vcur][i];
    t->vZ[i] = orc_mean_charge(Te * 11604.5, t->vlatdens, t->ac, t->aw);
    t->vne[i] = t->vZ[i] * t->vlatdens / (t->aw * AMU);
    t->vk[i] = orc_get_kappa(Te, Ti, t->vne[i], t->vZ[i], om);
    t->vCe[i] = orc_eos_cve(t->eos, t->vlatdens, Te * 11604.5);
    t->vg[i] = orc_get_gamma(Te, Ti, t->vne[i], t->vZ[i], om);
  }
}

/* ttm_fill_ghost_layers (:3264-3360) for the global ends: the end ranks set their outer ghost cell's natoms to 0 */
void orc_ttm1d_fill_ghosts(orc_ttm1d *t) { t->natoms[0] = 0; t->natoms[t->nx + 1] = 0; }

/* CFL_maxdt (:2444-2481) */
void orc_ttm1d_cfl(orc_ttm1d *t, int steps, double timestep)
{
  const int nx = t->nx;
  double mx = 1e9;
  int g;
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    if (!active(t, c)) continue;
    { const int m = (t->natoms[c + 1] >= t->fd_min_atoms && t->dens[c + 1] >= RHOMIN) ? c + 1 : c;
      const double v = t->Ce[c] * (t->hx * t->hx) / (t->fd_k[c] + t->fd_k[m]);
      if (v < mx) mx = v; }
  }
  t->max_dt_ttm = (steps > 0 ? mx : timestep / (double) t->fd_n_timesteps) * 0.9;
}

/* do_DIFF (:1473-1680) */
void orc_ttm1d_diff(orc_ttm1d *t, double tau)
{
  const int nx = t->nx, c0 = t->cur, c1 = c0 ^ 1, v0 = t->vcur, v1 = v0 ^ 1;
  const double xi_fac = 26.9815 * AMU / 3.0 * 1e30 / ((double) t->diff_substeps), invxsq = 1.0 / (t->hx * t->hx);
  const int per = nx / t->nranks;                 /* cells per rank of the reference run */
  int g, i;
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    int xmin, xmax;
    double xmaxTe, xminTe, xmaxk, xmink, Ce;
    if (g > t->last_active) continue;
    if (t->laser_active) t->eabs += t->source[c] * t->vol * tau;                           /* :1515-1518 */
    if (t->natoms[c] < t->fd_min_atoms || t->dens[c] < RHOMIN) continue;
    xmin = active_ge(t, c - 1) ? c - 1 : c;
    xmax = active_ge(t, c + 1) ? c + 1 : c;
    xmaxTe = t->temp[c0][xmax]; xminTe = t->temp[c0][xmin]; xmaxk = t->fd_k[xmax]; xmink = t->fd_k[xmin];
    if (g / per == t->vlat_proc) { xmaxTe = t->vtemp[v0][0]; xmaxk = t->vk[0]; }           /* :1535-1540: EVERY cell of that rank */
    Ce = t->Ce[c];
    t->temp[c1][c] = tau / Ce * ((((t->fd_k[c] + xmaxk) / 2 * (xmaxTe - t->temp[c0][c]) * invxsq)
                                  - ((t->fd_k[c] + xmink) / 2 * (t->temp[c0][c] - xminTe) * invxsq))
                                 - t->fd_g[c] * (t->temp[c0][c] - t->md_temp[c]) + t->source[c]) + t->temp[c0][c];
    t->U[c1][c] = t->U[c0][c] + (t->temp[c1][c] - t->temp[c0][c]) * Ce * t->vol / ((double) t->natoms[c]);
    t->xi[c] += (t->temp[c1][c] - t->md_temp[c]) * xi_fac * t->fd_g[c] / t->md_temp[c] / t->dens[c];
    t->xiarr[g] += (t->temp[c1][c] - t->md_temp[c]) * xi_fac * t->fd_g[c] / t->md_temp[c] / t->dens[c];
  }
  {                                                                                        /* the virtual lattice, :1595-1650 */
    double Ci = 8.589449886e+02;
    const int cl = t->last_active + 1;             /* lattice index of the last active cell */
    Ci *= t->vlatdens; Ci *= 11604.5; Ci *= 1e-30; Ci *= J2EV;
    for (i = 0; i < t->vlatdim; i++) {
      double xminTe, xmink, xmaxTe, xmaxk, Ce, dT;
      if (i == 0) { xminTe = t->temp[c0][cl]; xmink = t->fd_k[cl]; } else { xminTe = t->vtemp[v0][i - 1]; xmink = t->vk[i - 1]; }
      if (i < t->vlatdim - 1) { xmaxk = t->vk[i + 1]; xmaxTe = t->vtemp[v0][i + 1]; } else { xmaxk = t->vk[i]; xmaxTe = t->vtemp[v0][i]; }
      Ce = t->vCe[i];
      t->vtemp[v1][i] = tau / Ce * (((t->vk[i] + xmaxk) / 2 * (xmaxTe - t->vtemp[v0][i]) * invxsq)
                                    - ((t->vk[i] + xmink) / 2 * (t->vtemp[v0][i] - xminTe) * invxsq)
                                    - t->vg[i] * (t->vtemp[v0][i] - t->vmd[v0][i])) + t->vtemp[v0][i];
      dT = tau / Ci * t->vg[i] * (t->vtemp[v0][i] - t->vmd[v0][i]);
      t->vmd[v1][i] = t->vmd[v0][i] + dT;
    }
  }
  t->cur ^= 1; t->vcur ^= 1;                                                               /* :1654-1662 */
}

int orc_ttm1d_do_tmm(orc_ttm1d *t, int steps);
void orc_ttm1d_elec_pressure(orc_ttm1d *t, int steps);
/* calc_ttm (:75-155); returns the number of sub-steps */
int orc_ttm1d_calc(orc_ttm1d *t, int steps, double timestep, int n, const double *ort, const int *sorte, const double *impuls,
                   const double *masse, const int *numneighs, int *fdi)
{
  int i, g;
  double tau;
  orc_ttm1d_update_fd(t, steps, n, ort, sorte, impuls, masse, numneighs, fdi);
  if (t->adv) orc_ttm1d_adv(t, 1.0, steps);
  orc_ttm1d_cell_activation(t);
  orc_ttm1d_fillmesh(t, steps);
  orc_ttm1d_fill_ghosts(t);
  orc_ttm1d_cfl(t, steps, timestep);
  t->diff_substeps = (int) (timestep / t->max_dt_ttm);                                    /* :98-99 */
  if (t->fd_n_timesteps > t->diff_substeps) t->diff_substeps = t->fd_n_timesteps;
  tau = timestep / ((double) t->diff_substeps);
  for (i = 0; i < t->diff_substeps; i++) {
    if (t->tmm_on) orc_ttm1d_do_tmm(t, steps);                                             /* :113-117 */
    t->tmm_time += tau * 10.18 * 1.0e-15;
    orc_ttm1d_diff(t, tau); orc_ttm1d_fillmesh(t, steps); orc_ttm1d_fill_ghosts(t);
  }
  if (t->ep_on) orc_ttm1d_elec_pressure(t, steps);                                        /* :127-129 */
  for (g = 1; g <= t->nx; g++)                                                             /* :119-133 */
    if (t->natoms[g] >= t->fd_min_atoms && t->dens[g] >= RHOMIN)
      t->U[0][g] = t->U[1][g] = orc_eos_ee(t->eos, t->dens[g], t->temp[t->cur][g] * 11604.5) * 26.9815 * AMU * J2EV;
  return t->diff_substeps;
}

/* move_atoms_ttm, the TTM1D branch (imd_integrate.c:649-760): xi and v_com of the slab the atom is in NOW */
double orc_ttm1d_move_atoms(const orc_ttm1d *t, double timestep, int n, double *ort, const int *sorte, double *impuls, const double *kraft,
                            const double *masse)
{
  double ekin = 0.0;
  int a, d;
  for (a = 0; a < n; a++) {
    double xi = 0.0, vc[3] = {0, 0, 0}, e1 = 0, e2 = 0, tmp;
    if (!sorte || sorte[a] == 0) { const int g = slab(t, ort[3 * a]); xi = t->xiarr[g]; for (d = 0; d < 3; d++) vc[d] = t->vcom[d][g]; }
    for (d = 0; d < 3; d++) e1 += impuls[3 * a + d] * impuls[3 * a + d];
    for (d = 0; d < 3; d++) impuls[3 * a + d] += timestep * (kraft[3 * a + d] + xi * masse[a] * (impuls[3 * a + d] / masse[a] - vc[d]));
    if (t->ep_on && (!sorte || sorte[a] == 0)) {                                           /* imd_integrate.c:657-659,741-743 */
      /* every rank of the reference fills epress_deriv for ITS share of the row only (:596-620): an atom feels the force if
         the rank that owns it also owns its slab (nranks = how the reference run was cut; ownership taken from the position) */
      const int g = slab(t, ort[3 * a]);
      int md_rank = (int) (ort[3 * a] / (t->box_xx / t->nranks));
      md_rank = md_rank < 0 ? 0 : md_rank > t->nranks - 1 ? t->nranks - 1 : md_rank;
      if (md_rank == g / (t->nx / t->nranks)) impuls[3 * a] -= timestep * t->epress_deriv[g];
    }
    for (d = 0; d < 3; d++) e2 += impuls[3 * a + d] * impuls[3 * a + d];
    ekin += (e1 + e2) / (4 * masse[a]);
    tmp = timestep / masse[a];
    for (d = 0; d < 3; d++) ort[3 * a + d] += tmp * impuls[3 * a + d];
  }
  return ekin;
}


/* ---- the transfer-matrix laser source of the 1D model: imd_tmm.c (tmm_init :98-191, do_tmm :193-455, tmm_get_epsilon
 * :501-586, tmm_K1 / tmm_K2 :662-695, Runge5 / EE :36-96) -----------------------------------------------------------------
 * Pinned against the reference's own functions by tests/golden/ttm1d_tmm_slab.npz (oracle/_ref/imd_mpi_eam_ttm1d_tmm_nbl on
 * two ranks with synthetic alu_eps_bb.dat / K12.dat / EOS_phase_from_r_ti.txt).  Complex arithmetic is C99's, compiled by the
 * same compiler as the reference (multiplication inline, division through libgcc's __divdc3), so the restatement can be
 * bit-exact.  Reproduced as the source has it: `(int) nu * 30 - 1` in tmm_K1/K2 truncates nu BEFORE the product; tmm_Qabs keeps
 * entries of cells beyond the current absorption depth from earlier calls; nothing happens before step 2. */
void orc_ttm1d_tmm_init(orc_ttm1d *t, double I0, double lambda, double sigma_t, double t0, double sigma_t1, double t1,
                        double abs_threshold, double laser_threshold, int eps_bb_rows, const double *eps_bb, int K12_rows,
                        const double *K12, int pnx, int pny, const double *prange, const double *pz)
{
  const int nx = t->nx;
  t->tmm_on = 1; t->I0 = I0; t->lambda = lambda; t->sigma_t = sigma_t; t->t0 = t0; t->sigma_t1 = sigma_t1; t->t1 = t1;
  t->abs_threshold = abs_threshold; t->laser_threshold = laser_threshold;
  t->eps_re = za(nx, 8); t->eps_im = za(nx, 8); t->qabs = za(nx, 8); t->tmm_active = za(nx, sizeof(int)); t->km = za(nx, sizeof(double complex));
  t->eps_bb_rows = eps_bb_rows; t->eps_bb = za(3 * (size_t) eps_bb_rows, 8); memcpy(t->eps_bb, eps_bb, 24 * (size_t) eps_bb_rows);
  t->K12_rows = K12_rows; t->K12 = za(2 * (size_t) K12_rows, 8); memcpy(t->K12, K12, 16 * (size_t) K12_rows);
  t->phase.nx = pnx; t->phase.ny = pny; t->phase.xmin = prange[0]; t->phase.xmax = prange[1]; t->phase.ymin = prange[2]; t->phase.ymax = prange[3];
  t->phase.z = za((size_t) pnx * pny, 8); memcpy(t->phase.z, pz, 8 * (size_t) pnx * pny); t->phase.diag = NULL;
  t->omega_laser = 2.0 * M_PI * (3.0e8 / lambda);                                         /* :152-153 (c0 = 3e8, globals.h:499) */
  t->k0 = 2.0 * M_PI / lambda; t->k0 /= 1e10;                                              /* :154-155 */
  t->tmm_time = 0.0; t->laser_active = 0;
}

static double tmm_K(const orc_ttm1d *t, int col, double nu)
{
  if (nu < 1.0 / 30.0) return 1;
  if (nu < 50) {
    int i = (int) nu * 30 - 1;
    if (i < 0) i = 0;
    if (i > 1498) i = 1498;
    return t->K12[2 * i + col] + (30 * nu - i - 1) * (t->K12[2 * (i + 1) + col] - t->K12[2 * i + col]);
  }
  return col == 0 ? 315.0 / 8.0 / (nu * nu) - 675675.0 / 64.0 / (nu * nu * nu * nu) : 6.0 / (nu * nu) - 720.0 / (nu * nu * nu * nu);
}

static void tmm_get_epsilon(const orc_ttm1d *t, double Te, double Ti, double Z, double Ne, double *re, double *img)
{
  const double LIGHTSPEED = 2.998e8, AMU_ = 1.660539040200000e-27, A4p = 0.2;            /* globals.h:325 */
  const double complex un = 1, im = I;
  double omega_las, Ncr, Ni, TF, nu_met, nu_max, nu_pl, ksi, phase;
  double complex ebb_l, ebb_r, epsilon_bb, epsilon_met, epsilon_pl, epsilon_wr;
  int il = 0, ir = 0, id;
  Te *= 11604.52500617; Ti *= 11604.52500617;
  omega_las = (2.0 * M_PI * LIGHTSPEED / t->lambda);
  Ni = Ne / Z;
  for (id = 0; t->lambda * 1e6 > t->eps_bb[3 * id]; id++) { il = id; ir = id + 1; }
  orc_wr_tmm_rates(omega_las, t->laser_active ? omega_las : 0.0, Te, Ti, Z, Ne, &Ncr, &TF, &nu_met, &nu_max, &nu_pl);
  ksi = 0.75 * sqrt(M_PI) * nu_pl / omega_las;
  ebb_l = t->eps_bb[3 * il + 1] + I * t->eps_bb[3 * il + 2]; ebb_r = t->eps_bb[3 * ir + 1] + I * t->eps_bb[3 * ir + 2];
  epsilon_bb = ebb_l + (ebb_r - ebb_l) / (t->eps_bb[3 * ir] - t->eps_bb[3 * il]) * (t->lambda * 1e6 - t->eps_bb[3 * il]);
  {                                                                                        /* EOS_phase_from_r_ti (imd_ttm.c:2594-2637): bounds, round */
    double r = Ni * 26.9815 * AMU_, tt = Ti;
    r = r > t->phase.xmax ? t->phase.xmax : r; r = r < t->phase.xmin ? t->phase.xmin : r;
    tt = tt > t->phase.ymax ? t->phase.ymax : tt; tt = tt < t->phase.ymin ? t->phase.ymin : tt;
    phase = (double) ((int) round(orc_eos_lookup(&t->phase, r, tt)));
  }
  if (fabs(phase) == 3 || fabs(phase) == 4 || fabs(phase) == 5) epsilon_bb = 0 * I * 0;
  epsilon_met = epsilon_bb + un - Ne / Ncr / (un + im * (nu_met < nu_max ? nu_met : nu_max) / omega_las);
  epsilon_pl = un - Ne / Ncr * (tmm_K(t, 0, ksi) - im * nu_pl / omega_las * tmm_K(t, 1, ksi));
  epsilon_wr = epsilon_pl + (epsilon_met - epsilon_pl) * exp(-A4p * Te / TF);
  *re = creal(epsilon_wr); *img = cimag(epsilon_wr);
}

static void cmatmul(const double complex *a, const double complex *b, double complex *p)
{
  p[0] = a[0] * b[0] + a[2] * b[1]; p[1] = a[1] * b[0] + a[3] * b[1]; p[2] = a[0] * b[2] + a[2] * b[3]; p[3] = a[1] * b[2] + a[3] * b[3];
}
static double tmm_EE(double z, double delta, double complex kl, double complex Bplus, double complex Bminus)
{
  const double complex im = 0 + I * 1;
  const double phi1 = creal(delta * z * kl * im), phi2 = cimag(delta * z * kl * im);
  const double complex eiphi = exp(phi1) * cos(phi2) + I * exp(phi1) * sin(phi2);
  const double complex Ez = (Bplus * eiphi + Bminus / eiphi), Ey = 0;
  return creal(Ez) * creal(Ez) + cimag(Ez) * cimag(Ez) + creal(Ey) * creal(Ey) + cimag(Ey) * cimag(Ey);
}
static double tmm_Runge5(double delta, double complex kl, double complex Bplus, double complex Bminus)
{
  double dx = 1, cur_xpos = 0, result = 0, k1, k3, k4, k5;
  const double errval = 1e-5;
  while (cur_xpos < 1 && dx > 1e-5) {
    double ERR;
    k1 = dx / 3.0 * tmm_EE(cur_xpos, delta, kl, Bplus, Bminus);
    k3 = dx / 3.0 * tmm_EE(cur_xpos + dx / 3.0, delta, kl, Bplus, Bminus);
    k4 = dx / 3.0 * tmm_EE(cur_xpos + 0.5 * dx, delta, kl, Bplus, Bminus);
    k5 = dx / 3.0 * tmm_EE(cur_xpos + dx, delta, kl, Bplus, Bminus);
    ERR = k1 - 4.5 * k3 + 4 * k4 - 0.5 * k5;
    if (ERR < 5.0 / 32 * errval) { result += 0.5 * (k1 + 4 * k4 + k5); cur_xpos += dx; dx *= 1.1; }
    else if (ERR > 5 * errval) { dx *= 0.5; continue; }
    else { result += 0.5 * (k1 + 4 * k4 + k5); cur_xpos += dx; }
    if (cur_xpos + dx > 1) dx = 1 - cur_xpos;
  }
  return result;
}

/* do_tmm (:193-455).  Returns 0, or 1 when the sample is too short for the absorption threshold (the reference aborts). */
int orc_ttm1d_do_tmm(orc_ttm1d *t, int steps)
{
  const int nx = t->nx, c0 = t->cur;
  const double Imp = 3.769911184307751e+02;
  double complex Cinv[4], Pmat[4], tmpmat[4], tmpmat2[4], Emat[4], kl, kr, eiphi, BR, BT, B0;
  double phi1, phi2, dx, ecut_thresh = exp(-t->abs_threshold);
  int g, j, ecut = 0, tooshort = 1;
  t->I_t = t->I0 * exp(-pow(t->tmm_time - t->t0, 2) / (t->sigma_t * t->sigma_t));
  t->I_t += t->I0 * exp(-pow(t->tmm_time - t->t1, 2) / (t->sigma_t1 * t->sigma_t1));
  if (steps < 2) return 0;
  if (sqrt(2.0 * t->I_t * Imp) < t->laser_threshold * sqrt(2 * t->I0 * Imp)) { t->laser_active = 0; return 0; }
  t->laser_active = 1;
  for (g = 0; g < nx; g++) {                                                               /* :212-254 */
    const int c = g + 1;
    if (t->natoms[c] < t->fd_min_atoms || t->dens[c] < RHOMIN) { t->eps_re[g] = 0.0; t->eps_im[g] = 0.0; t->tmm_active[g] = 0; continue; }
    tmm_get_epsilon(t, t->temp[c0][c], t->md_temp[c], t->Z[c], t->ne[c], &t->eps_re[g], &t->eps_im[g]);
    t->tmm_active[g] = 1;
  }
  for (g = 0; g < nx / t->nranks; g++) if (t->eps_re[g] == 0) t->eps_re[g] = 1.0;         /* :264-273: rank 0's cells only */
  BR = BT = 0.0 + I * 0; B0 = 1.0 + I * 0;
  t->km[0] = t->k0;                                                                        /* :296-306 */
  for (g = 1; g < nx - 1; g++) t->km[g] = t->tmm_active[g] == 0 ? t->k0 : t->k0 * csqrt(t->eps_re[g] + I * t->eps_im[g]);
  Emat[0] = 1.0 + I * 0; Emat[1] = 0; Emat[2] = 0; Emat[3] = 1.0 + I * 0.0;
  for (g = 0; g < nx - 1; g++) {                                                           /* :310-375 */
    dx = g == 0 ? 1e10 : t->hx;
    kl = t->km[g]; kr = t->km[g + 1];
    phi1 = creal(dx * kl * I); phi2 = cimag(dx * kl * I);
    eiphi = exp(phi1) * cos(phi2) + I * exp(phi1) * sin(phi2);
    Pmat[0] = eiphi; Pmat[2] = 1.0 / eiphi; Pmat[1] = eiphi; Pmat[3] = -Pmat[2];
    Cinv[0] = 0.5; Cinv[2] = 0.5 * kl / kr; Cinv[1] = 0.5; Cinv[3] = -Cinv[2];
    cmatmul(Cinv, Pmat, tmpmat); cmatmul(tmpmat, Emat, tmpmat2);
    for (j = 0; j < 4; j++) Emat[j] = tmpmat2[j];
    BR = -Emat[1] * B0 / Emat[3];
    BT = Emat[0] * B0 + Emat[2] * BR;
    if (creal(BT) * creal(BT) + cimag(BT) * cimag(BT) < ecut_thresh) { tooshort = 0; ecut = g + 1; break; }
  }
  if (tooshort) return 1;
  {
    const double complex tmpr0 = BR / B0;
    t->refl = creal(tmpr0) * creal(tmpr0) + cimag(tmpr0) * cimag(tmpr0);
    t->trans = creal(BT) * creal(BT) + cimag(BT) * cimag(BT);
    t->absorbed = 1.0 - t->refl - t->trans;
  }
  {                                                                                        /* :377-445 */
    double complex *Bplus = za(ecut + 1, sizeof(double complex)), *Bminus = za(ecut + 1, sizeof(double complex)), Fvec[2], tmpvec[2];
    Fvec[0] = B0; Fvec[1] = BR; Bplus[0] = B0; Bminus[0] = BR;
    for (g = 0; g < ecut; g++) {
      dx = g == 0 ? 1e10 : t->hx;
      kl = t->km[g]; kr = t->km[g + 1];
      phi1 = creal(dx * kl * I); phi2 = cimag(dx * kl * I);
      eiphi = exp(phi1) * cos(phi2) + I * exp(phi1) * sin(phi2);
      Pmat[0] = eiphi; Pmat[2] = 1.0 / eiphi; Pmat[1] = eiphi; Pmat[3] = -Pmat[2];
      Cinv[0] = 0.5; Cinv[2] = 0.5 * kl / kr; Cinv[1] = 0.5; Cinv[3] = -Cinv[2];
      cmatmul(Cinv, Pmat, tmpmat);
      tmpvec[0] = tmpmat[0] * Fvec[0] + tmpmat[2] * Fvec[1]; tmpvec[1] = tmpmat[1] * Fvec[0] + tmpmat[3] * Fvec[1];   /* matmul(tmpmat, Fvec, tmpvec): the first column of matmul's product */
      Fvec[0] = tmpvec[0]; Fvec[1] = tmpvec[1];
      Bplus[g + 1] = Fvec[0]; Bminus[g + 1] = Fvec[1];
    }
    for (g = 0; g < ecut; g++) {
      dx = g == 0 ? 1e10 : t->hx;
      t->qabs[g] = t->I_t * t->k0 * t->eps_im[g] * tmm_Runge5(dx, t->km[g], Bplus[g], Bminus[g]);
      t->qabs[g] *= 1e10; t->qabs[g] *= 6.3538562638e-26;
    }
    free(Bplus); free(Bminus);
  }
  t->ecut = ecut;
  for (g = 0; g < nx; g++) t->source[g + 1] = t->qabs[g];                                   /* scatter, :448-453 */
  return 0;
}


/* ---- electron pressure (-DELECPRESS): do_electronic_pressure, :553-623; EOS_pe_from_r_te, :2638-2693 (table in Pa) ---------- */
void orc_ttm1d_set_epress(orc_ttm1d *t, int pnx, int pny, const double *prange, const double *pz)
{
  t->ep_on = 1;
  t->epress = za(t->nx, 8); t->epress_deriv = za(t->nx, 8);
  t->pe.nx = pnx; t->pe.ny = pny; t->pe.xmin = prange[0]; t->pe.xmax = prange[1]; t->pe.ymin = prange[2]; t->pe.ymax = prange[3];
  t->pe.z = za((size_t) pnx * pny, 8); memcpy(t->pe.z, pz, 8 * (size_t) pnx * pny); t->pe.diag = NULL;
}
void orc_ttm1d_elec_pressure(orc_ttm1d *t, int steps)
{
  const int nx = t->nx, c0 = t->cur;
  int g;
  for (g = 0; g < nx; g++) { t->epress[g] = 0.0; t->epress_deriv[g] = 0.0; }
  for (g = 0; g < nx; g++) {
    const int c = g + 1;
    double r, tt;
    if (t->dens[c] < RHOMIN || t->natoms[c] < t->fd_min_atoms) continue;
    if (g > t->last_active) continue;
    r = t->dens[c]; tt = t->temp[c0][c] * 11604.5;
    r = r > t->pe.xmax ? t->pe.xmax : r; r = r < t->pe.xmin ? t->pe.xmin : r;
    tt = tt > t->pe.ymax ? t->pe.ymax : tt; tt = tt < t->pe.ymin ? t->pe.ymin : tt;
    t->epress[g] = orc_eos_lookup(&t->pe, r, tt) * 6.242e-12;                               /* Pa -> eV/A^3, :2692 */
  }
  for (g = 1; g < nx - 1; g++) {
    if (g > t->last_active) continue;
    if (steps > 1) t->epress_deriv[g] = (t->epress[g + 1] - t->epress[g - 1]) / 2.0 / t->hx / (t->natoms[g + 1] / t->vol);
  }
}
