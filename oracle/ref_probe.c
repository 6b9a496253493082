/*
 * ref_probe.c -- TEST INFRASTRUCTURE (oracle side), not product code.
 *
 * Observation probe linked into a build of the *real* reference (IMD/lasMD,
 * /root/reference) by oracle/build_ref.sh via `ld --wrap`.  It changes no
 * reference behaviour: each wrapper calls the real function and then dumps the
 * reference's own global state (cells, neighbour list, potential tables, sums)
 * as raw binary, so that golden fixtures carry full fp64 precision instead of
 * the text formats of the reference's writers.
 *
 *   __wrap_calc_forces(steps)  -> __real_calc_forces(steps); dump if selected
 *
 * Environment:
 *   IMD_PROBE_DIR    directory for dumps (default ".")
 *   IMD_PROBE_STEPS  comma separated list of step numbers to dump (default "0")
 *
 * Reference interfaces observed: calc_forces (imd_forces_nbl.c:295,
 * imd_main_risc_3d.c:31), globals in globals.h:99-188,670-711, cell layout
 * types.h:188-349, pot_table_t types.h:537-549.
 */
#include "imd.h"

void __real_calc_forces(int steps);

static int probe_want(int s)
{
  const char *e = getenv("IMD_PROBE_STEPS");
  if (e == NULL) return s == 0;
  while (*e) {
    char *end;
    long v = strtol(e, &end, 10);
    if (end == e) break;
    if (v == s) return 1;
    e = end;
    while (*e == ',' || *e == ' ') e++;
  }
  return 0;
}

static void wi(FILE *f, int v)          { fwrite(&v, sizeof(int), 1, f); }
static void wd(FILE *f, double v)       { fwrite(&v, sizeof(double), 1, f); }
static void wv(FILE *f, vektor v)       { wd(f, v.x); wd(f, v.y); wd(f, v.z); }
static void wiv(FILE *f, ivektor v)     { wi(f, v.x); wi(f, v.y); wi(f, v.z); }

static void dump_table(FILE *f, pot_table_t *pt)
{
  int i, n;
  wi(f, pt->ncols);
  wi(f, pt->maxsteps);
  for (i = 0; i < pt->ncols; i++) wd(f, pt->begin[i]);
  for (i = 0; i < pt->ncols; i++) wd(f, pt->end[i]);
  for (i = 0; i < pt->ncols; i++) wd(f, pt->step[i]);
  for (i = 0; i < pt->ncols; i++) wd(f, pt->invstep[i]);
  for (i = 0; i < pt->ncols; i++) wi(f, pt->len[i]);
  n = (pt->maxsteps + 2) * pt->ncols;
  for (i = 0; i < n; i++) wd(f, pt->table[i]);
}

static void probe_dump(int step)
{
  char  fn[1024];
  const char *dir = getenv("IMD_PROBE_DIR");
  FILE *f;
  int   k, i, flags = 0, ninner = 0;
  pot_table_t empty;
  memset(&empty, 0, sizeof(empty));

  sprintf(fn, "%s/probe.%d.%d.bin", dir ? dir : ".", myid, step);
  f = fopen(fn, "wb");
  if (f == NULL) { fprintf(stderr, "probe: cannot open %s\n", fn); return; }

#ifdef EAM2
  flags |= 1;
#endif
#ifdef NBLIST
  flags |= 2;
#endif
#ifdef TTM
  flags |= 4;
#endif
#ifdef MPI
  flags |= 8;
#endif
  fwrite("IMDPROBE", 1, 8, f);
  wi(f, 1);           /* format version */
  wi(f, flags);
  wi(f, step);
  wi(f, myid);
  wi(f, ntypes);
  wi(f, nallcells);
  wi(f, ncells);
  wiv(f, cell_dim);
  wiv(f, global_cell_dim);
  wiv(f, pbc_dirs);
  wiv(f, cpu_dim);
  wiv(f, my_coord);
  wi(f, (int) natoms);
#ifdef NBLIST
  wi(f, have_valid_nbl);
  wi(f, nbl_count);
  wi(f, last_nbl_len);
#else
  wi(f, 0); wi(f, 0); wi(f, 0);
#endif
  wv(f, box_x); wv(f, box_y); wv(f, box_z);
  wd(f, cellsz);
#ifdef NBLIST
  wd(f, nbl_margin);
#else
  wd(f, 0.0);
#endif
  wd(f, timestep);
  wd(f, tot_pot_energy);
  wd(f, virial);
  wd(f, tot_kin_energy);

  /* potential tables as the reference holds them after init_threepoint */
#ifdef PAIR
  dump_table(f, &pair_pot);
#else
  wi(f, 0); wi(f, 0);
#endif
#ifdef EAM2
  dump_table(f, &rho_h_tab);
  dump_table(f, &embed_pot);
#else
  wi(f, 0); wi(f, 0);
  wi(f, 0); wi(f, 0);
#endif

  /* all cells, in linear cell-array order: occupancy, ids, types, positions */
  for (k = 0; k < nallcells; k++) wi(f, (cell_array + k)->n);
  for (k = 0; k < nallcells; k++) {
    cell *p = cell_array + k;
    for (i = 0; i < p->n; i++) {
      wi(f, (int) NUMMER(p, i));
      wi(f, (int) SORTE(p, i));
      wd(f, ORT(p, i, X)); wd(f, ORT(p, i, Y)); wd(f, ORT(p, i, Z));
    }
  }

  /* inner cells, in the order of the reference's `cells` list */
  for (k = 0; k < ncells; k++) { wi(f, (int) (CELLPTR(k) - cell_array)); ninner += CELLPTR(k)->n; }
  wi(f, ninner);
  for (k = 0; k < ncells; k++) {
    cell *p = CELLPTR(k);
    for (i = 0; i < p->n; i++) {
      wd(f, MASSE(p, i));
      wd(f, IMPULS(p, i, X)); wd(f, IMPULS(p, i, Y)); wd(f, IMPULS(p, i, Z));
      wd(f, KRAFT(p, i, X));  wd(f, KRAFT(p, i, Y));  wd(f, KRAFT(p, i, Z));
      wd(f, POTENG(p, i));
#ifdef EAM2
      wd(f, EAM_RHO(p, i)); wd(f, EAM_DF(p, i));
#else
      wd(f, 0.0); wd(f, 0.0);
#endif
#ifdef NBLIST
      wd(f, NBL_POS(p, i, X)); wd(f, NBL_POS(p, i, Y)); wd(f, NBL_POS(p, i, Z));
#else
      wd(f, 0.0); wd(f, 0.0); wd(f, 0.0);
#endif
#ifdef TTM
      wi(f, (int) NUMNEIGHS(p, i));
#else
      wi(f, 0);
#endif
    }
  }

#ifdef NBLIST
  /* Verlet list exactly as make_nblist left it (imd_forces_nbl.c:136-287) */
  fwrite(cl_off, sizeof(int), nallcells, f);
  fwrite(tl, sizeof(int), ninner + 1, f);
  fwrite(tb, sizeof(int), tl[ninner], f);
  /* neighbour cell stencil (imd_geom_3d.c:846-975) */
  for (k = 0; k < ncells; k++) {
    wi(f, (int) cnbrs[k].np);
    for (i = 0; i < NNBCELL; i++) wi(f, (int) cnbrs[k].nq[i]);
  }
#endif
  fwrite("ENDPROBE", 1, 8, f);
  fclose(f);
}

void __wrap_calc_forces(int st)
{
  __real_calc_forces(st);
  if (probe_want(st)) probe_dump(st);
}
