/*
 * ref_probe_ttm1d.c -- TEST INFRASTRUCTURE (oracle side), not product code.
 *
 * The reference's OWN 1D two-temperature model (imd_ttm.c, the file its Makefile builds for ttm targets, Makefile:185)
 * driven function by function in the order of its calc_ttm (imd_ttm.c:75-155) and dumped as raw doubles.  Linked into a
 * build of the real reference (oracle/build_ref.sh mpi_eam_ttm1d_nbl: -DTTM -DTMM -DTTM1D with imd_ttm.c) with
 * ld --wrap=calc_ttm, like ref_probe_ttm.c for the 3D file.  What is left out of the sequence is do_tmm, the transfer-
 * matrix Helmholtz solver for the laser field (imd_tmm.c; its input files alu_eps_bb.dat / K12.dat are not part of the
 * tree): in the build mpi_eam_ttm1d_nbl tmm_init / do_tmm are answered with "no laser", so node.source stays 0; in the
 * build mpi_eam_ttm1d_tmm_nbl imd_tmm.c is linked and, with IMD_PROBE_TMM set, do_tmm runs at the top of every sub-step as
 * in calc_ttm (:113) on synthetic input files, its arrays dumped to tmm.<rank>.<step>.bin.  Everything else is the reference's code:
 * update_fd, do_ADV (IMD_PROBE_ADV; needs two ranks, do_COMMFLUX has no one-rank branch), do_cell_activation, do_FILLMESH,
 * ttm_fill_ghost_layers, CFL_maxdt, do_DIFF with the virtual lattice, the xi all-gather and the EOS update of U that
 * calc_ttm itself does after the loop (:110-133, restated here because it is not a function).
 *
 * Environment: IMD_PROBE_DIR, IMD_PROBE_STEPS as in ref_probe.c.
 */
#include "imd.h"

void __real_calc_ttm(void);
int  __wrap_tmm_init(void) { return 0; }
int  __wrap_do_tmm(real dt) { return 0; }

static FILE *tf = NULL;
static void wi(int v) { fwrite(&v, sizeof(int), 1, tf); }
static void wd(double v) { fwrite(&v, sizeof(double), 1, tf); }

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

/* one stage: tag[16], then per local lattice cell 17 doubles (l1 fields, l2.temp, l2.U), the global arrays, the virtual
 * lattice, scalars */
static void dump_stage(const char *tag)
{
  char t[16];
  int i;
  memset(t, 0, sizeof(t)); strncpy(t, tag, 15);
  fwrite(t, 1, 16, tf);
  wi(local_fd_dim.x); wi(global_fd_dim.x); wi(vlatdim);
  for (i = 0; i < local_fd_dim.x; i++) {
    const ttm_Element *e = &l1[i], *e2 = &l2[i];
    wd((double) e->natoms); wd((double) e->natoms_old); wd(e->temp); wd(e->xi); wd(e->md_temp); wd(e->source);
    wd(e->vcomx); wd(e->vcomy); wd(e->vcomz); wd(e->dens); wd(e->fd_k); wd(e->fd_g); wd(e->Z); wd(e->Ce); wd(e->ne); wd(e->U);
    wd(e2->temp); wd(e2->U);
  }
  for (i = 0; i < global_fd_dim.x; i++) {
    wd((double) natomsglobal[i]); wd((double) fluxfromleftglobal[i]); wd((double) fluxfromrightglobal[i]);
    wd(vcomxglobal[i]); wd(vcomyglobal[i]); wd(vcomzglobal[i]); wd(xiarr_global[i]);
  }
  for (i = 0; i < vlatdim; i++) {
    const ttm_Element *e = &vlattice1[i];
    wd(e->temp); wd(e->md_temp); wd(e->dens); wd(e->fd_k); wd(e->fd_g); wd(e->Z); wd(e->Ce); wd(e->ne);
  }
  wd((double) last_active_cell_global); wd((double) cur_vlattice_proc); wd((double) diff_substeps); wd(max_dt_ttm);
}

/* do_tmm's state after a call: what rank 0 solved (eps, active cells, Qabs over the global row), my cells' source, scalars */
static FILE *mf = NULL;
static void dump_tmm(int sub)
{
  int i;
  double v;
  if (!mf) return;
  fwrite(&sub, sizeof(int), 1, mf);
  for (i = 0; i < global_fd_dim.x; i++) { v = tmm_eps_real_arr_global[i]; fwrite(&v, 8, 1, mf); }
  for (i = 0; i < global_fd_dim.x; i++) { v = tmm_eps_imag_arr_global[i]; fwrite(&v, 8, 1, mf); }
  for (i = 0; i < global_fd_dim.x; i++) { v = (double) tmm_active_cell_global[i]; fwrite(&v, 8, 1, mf); }
  for (i = 0; i < global_fd_dim.x; i++) { v = tmm_Qabs[i]; fwrite(&v, 8, 1, mf); }
  for (i = 0; i < local_fd_dim.x; i++) { v = l1[i].source; fwrite(&v, 8, 1, mf); }
  v = I_t; fwrite(&v, 8, 1, mf); v = tmm_time; fwrite(&v, 8, 1, mf); v = tmm_refl; fwrite(&v, 8, 1, mf);
  v = tmm_trans; fwrite(&v, 8, 1, mf); v = tmm_abs; fwrite(&v, 8, 1, mf); v = (double) laser_active; fwrite(&v, 8, 1, mf);
  v = Eabs_local; fwrite(&v, 8, 1, mf);
}

void __wrap_calc_ttm(void)
{
  int i, dumping = want(steps);
  const int with_tmm = getenv("IMD_PROBE_TMM") != NULL;
  double tau_DIFF;
  if (dumping) {
    char fn[1024];
    const char *dir = getenv("IMD_PROBE_DIR");
    sprintf(fn, "%s/ttm1d.%d.%d.bin", dir ? dir : ".", myid, steps);
    tf = fopen(fn, "wb");
    fwrite("IMDTT1DP", 1, 8, tf);
    wi(1); wi(steps); wi(fd_min_atoms); wi(fd_n_timesteps); wi(vlatbuffer);
    wd(fd_h.x); wd(fd_h.y); wd(fd_h.z); wd(fd_vol); wd(neighvol); wd(atomic_weight); wd(atomic_charge); wd(timestep);
    wd(vlatdens); wd(Temin);
  }
  update_fd();
  if (dumping) dump_stage("update_fd");
  if (getenv("IMD_PROBE_ADV")) { do_ADV(1.0); if (dumping) dump_stage("adv"); }
  do_cell_activation();
  do_FILLMESH();
  ttm_fill_ghost_layers();
  if (dumping) dump_stage("fillmesh0");
  CFL_maxdt();
  diff_substeps = MAX(fd_n_timesteps, (int) (timestep / max_dt_ttm));                 /* imd_ttm.c:98-99 */
  tau_DIFF = timestep / ((double) diff_substeps);
  if (dumping) dump_stage("cfl");
  if (dumping && with_tmm) {
    char fn[1024];
    const char *dir = getenv("IMD_PROBE_DIR");
    double hd[8];
    sprintf(fn, "%s/tmm.%d.%d.bin", dir ? dir : ".", myid, steps);
    mf = fopen(fn, "wb");
    fwrite("IMDTMM1P", 1, 8, mf);
    hd[0] = global_fd_dim.x; hd[1] = local_fd_dim.x; hd[2] = diff_substeps; hd[3] = k0; hd[4] = omega_laser; hd[5] = tmm_time;
    hd[6] = fd_h.x; hd[7] = tmm_absorption_threshold;
    fwrite(hd, 8, 8, mf);
  }
  for (i = 0; i < diff_substeps; i++) {
    if (with_tmm) {
      do_tmm(tau_DIFF);                                                                  /* imd_ttm.c:113 */
      if (dumping && (i < 2 || i == diff_substeps - 1)) dump_tmm(i);
    }
    tmm_time += tau_DIFF * 10.18 * 1.0e-15;
    do_DIFF(tau_DIFF);
    if (dumping && i == 0) dump_stage("diff_only1");
    do_FILLMESH();
    ttm_fill_ghost_layers();
    if (dumping && i == 0) dump_stage("substep1");
  }
  MPI_Allgather(&xiarr_local[1], local_fd_dim.x - 2, MPI_DOUBLE, &xiarr_global[0], local_fd_dim.x - 2, MPI_DOUBLE, cpugrid);   /* :110-111 */
#ifdef ELECPRESS
  do_electronic_pressure();                                                                                                        /* :127-129 */
  if (dumping) {                                      /* epress_global (all-reduced) and this rank's share of epress_deriv */
    char fn[1024];
    const char *dir = getenv("IMD_PROBE_DIR");
    FILE *ef;
    sprintf(fn, "%s/epress.%d.%d.bin", dir ? dir : ".", myid, steps);
    ef = fopen(fn, "wb");
    fwrite(epress_global, sizeof(double), global_fd_dim.x, ef);
    fwrite(epress_deriv, sizeof(double), global_fd_dim.x, ef);
    fclose(ef);
  }
#endif
  tot_elec_energy_local = 0;                                                                                                       /* :119-133 */
  for (i = 1; i < local_fd_dim.x - 1; ++i)
    if (l1[i].natoms >= fd_min_atoms && l1[i].dens >= 2 /* RHOMIN */) {
      l1[i].U = EOS_ee_from_r_te(l1[i].dens, l1[i].temp * 11604.5) * 26.9815 * AMU * J2eV;
      l2[i].U = l1[i].U;
      tot_elec_energy_local += l1[i].U * ((double) l1[i].natoms);
    }
  if (mf) { int end = -1; fwrite(&end, sizeof(int), 1, mf); fclose(mf); mf = NULL; }
  if (dumping) {
    dump_stage("substeps_done");
    fwrite("ENDTT1DP", 1, 8, tf);
    fclose(tf); tf = NULL;
  }
}
