/*
 * dist_host.c -- a C host (no Python, no torch) driving the multi-rank MD loop of libimdgpu.so.
 *
 * What a C program like IMD itself needs per rank is three calls: imdgpu_dist_init_* once, then imdgpu_dist_step() per MD
 * step -- the library rebuilds (migration, re-binning, halo plan, neighbour list: imdgpu_dist_rebuild) whenever some
 * rank's list went stale, like main_loop does with fix_cells / make_nblist (imd_main_3d.c:406-812, imd_forces_nbl.c:300-311).
 * Here the ranks are THREADS of one process on one GPU (imdgpu_dist_init_local); with one process per GPU the only
 * difference is imdgpu_dist_init_rccl(ctx, nranks, rank, id) with the id of imdgpu_dist_unique_id broadcast by the launcher.
 *
 * The program runs the same 4 000-atom FCC block (synthetic tabulated EAM) on one rank through
 * imdgpu_md_step and on PX x PY x PZ ranks through imdgpu_dist_step, and compares the trajectories atom by atom.
 *
 * build:  gcc -O2 -o dist_host dist_host.c -I../../include -L../csrc -limdgpu -lpthread -lm -Wl,-rpath,$PWD/../csrc
 * usage:  dist_host [px py pz [steps [ttm 0|1]]]        exit code 0 = trajectories agree to 1e-9
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "imdgpu.h"

#define NTAB 4000
static double cutf(double r) { double x = (r - 6.5) / 0.6, x4 = x * x * x * x; return r < 6.5 ? x4 / (1.0 + x4) : 0.0; }
static double f_phi(double r) { return 0.27 * (exp(-2 * 1.16 * (r - 2.86)) - 2 * exp(-1.16 * (r - 2.86))) * cutf(r); }
static double f_rho(double r) { return exp(-3.0 * (r / 2.86 - 1.0)) * cutf(r); }
static double f_emb(double p) { return -1.6 * sqrt(p) + 0.02 * p * p; }

typedef struct { double begin, end, step, invstep; int len; double tab[NTAB + 2]; imdgpu_pot_table t; } table1;
/* one column, two rows appended that continue the last parabola (init_threepoint, imd_potential.c:1260-1276) */
static void make_table(table1 *T, double b, double e, double (*fn)(double), int radial)
{
  int k;
  T->begin = b; T->end = e; T->step = (e - b) / (NTAB - 1); T->invstep = 1.0 / T->step; T->len = NTAB;
  for (k = 0; k < NTAB; k++) { double a = b + T->step * k; T->tab[k] = fn(radial ? sqrt(a) : a); }
  if (radial && T->tab[NTAB - 1] != 0.0) { double s = T->tab[NTAB - 1]; for (k = 0; k < NTAB; k++) T->tab[k] -= s; }
  T->tab[NTAB] = 3 * T->tab[NTAB - 1] - 3 * T->tab[NTAB - 2] + T->tab[NTAB - 3];
  T->tab[NTAB + 1] = 6 * T->tab[NTAB - 1] - 8 * T->tab[NTAB - 2] + 3 * T->tab[NTAB - 3];
  T->t.begin = &T->begin; T->t.end = &T->end; T->t.step = &T->step; T->t.invstep = &T->invstep; T->t.len = &T->len;
  T->t.ncols = 1; T->t.maxsteps = NTAB; T->t.table = T->tab;
}

static table1 PHI, RHO, EMB;
static double BOX[9], CELLSZ;
static int GDIM[3], PBC[3] = {1, 1, 1}, NSTEPS = 25, WITH_TTM = 0;
static int NATOMS; static double *POS, *MOM, *MASS; static int *IDS;
static double EOS_CVE[24 * 40], EOS_EE[24 * 40];

typedef struct { int rank, cpu[3], coord[3]; void *group; int n; double *pos; int *ids; int rc; char err[512]; } rank_t;

static int CK(rank_t *R, imdgpu_ctx *c, int rc) { if (rc) { snprintf(R->err, sizeof R->err, "%s", imdgpu_last_error(c)); R->rc = 1; } return rc; }

static void *rank_main(void *arg)
{
  rank_t *R = (rank_t *) arg;
  imdgpu_ctx *c = NULL;
  int d, i, n = 0, dim[3], multi = R->cpu[0] * R->cpu[1] * R->cpu[2] > 1;
  double *p, *m, *ms; int *id;
  if (imdgpu_create(&c, 0)) { snprintf(R->err, sizeof R->err, "imdgpu_create failed"); R->rc = 1; return NULL; }
  for (d = 0; d < 3; d++) dim[d] = GDIM[d] / R->cpu[d] + 2;
  if (CK(R, c, imdgpu_set_geometry(c, BOX, PBC, dim, GDIM, R->cpu, R->coord))) return NULL;
  if (CK(R, c, imdgpu_set_params(c, 1, CELLSZ, 0.5, 0.1, WITH_TTM))) return NULL;
  if (CK(R, c, imdgpu_upload_tables(c, &PHI.t, &RHO.t, &EMB.t))) return NULL;
  if (WITH_TTM) {
    imdgpu_ttm_params tp;
    memset(&tp, 0, sizeof tp);
    tp.fd_ext[0] = tp.fd_ext[1] = tp.fd_ext[2] = 1; tp.fd_min_atoms = 5; tp.fd_n_timesteps = 100;
    tp.atomic_weight = 26.9815; tp.atomic_charge = 13.0;
    tp.cve_nx = tp.ee_nx = 24; tp.cve_ny = tp.ee_ny = 40;
    tp.cve_xmin = tp.ee_xmin = 500.0; tp.cve_xmax = tp.ee_xmax = 4500.0;
    tp.cve_ymin = 1.0; tp.cve_ymax = 60000.0; tp.ee_ymin = 1.0; tp.ee_ymax = 245.0;
    tp.cve_z = EOS_CVE; tp.ee_z = EOS_EE;
    if (CK(R, c, imdgpu_ttm_init(c, &tp))) return NULL;
  }
  /* my atoms: cell coordinate of the position, owner = cell / cells per rank (imd_geom_mpi_3d.c:119-128) */
  p = malloc(sizeof(double) * 3 * NATOMS); m = malloc(sizeof(double) * 3 * NATOMS); ms = malloc(sizeof(double) * NATOMS);
  id = malloc(sizeof(int) * NATOMS);
  for (i = 0; i < NATOMS; i++) {
    int mine = 1;
    for (d = 0; d < 3; d++) {
      int cc = (int) (GDIM[d] * (POS[3 * i + d] / BOX[4 * d]));
      if (cc >= GDIM[d]) cc = GDIM[d] - 1;
      if (cc / (GDIM[d] / R->cpu[d]) != R->coord[d]) mine = 0;
    }
    if (!mine) continue;
    memcpy(p + 3 * n, POS + 3 * i, 24); memcpy(m + 3 * n, MOM + 3 * i, 24); ms[n] = MASS[i]; id[n] = IDS[i]; n++;
  }
  if (CK(R, c, imdgpu_upload_atoms(c, n, p, m, ms, NULL, id))) return NULL;
  if (multi && CK(R, c, imdgpu_dist_init_local(c, R->group, R->rank))) return NULL;
  for (i = 0; i < NSTEPS; i++)
    if (CK(R, c, multi ? imdgpu_dist_step(c, i) : imdgpu_md_step(c, i))) return NULL;
  if (CK(R, c, imdgpu_synchronize(c))) return NULL;
  {
    int nl = 0, ng = 0;
    if (CK(R, c, imdgpu_num_atoms(c, &nl, &ng))) return NULL;
    R->n = nl; R->pos = malloc(sizeof(double) * 3 * (nl + 1)); R->ids = malloc(sizeof(int) * (nl + 1));
    if (CK(R, c, imdgpu_download_atoms(c, NULL, R->pos, NULL, NULL, NULL, NULL, NULL, NULL, R->ids, NULL, NULL))) return NULL;
  }
  if (multi) { int nreb = 0; imdgpu_dist_rebuild_count(c, &nreb); if (R->rank == 0) printf("rank 0: %d atoms at the end, %d rebuilds\n", R->n, nreb); }
  if (multi) imdgpu_local_group_barrier(R->group);       /* nobody destroys its context while others still run */
  imdgpu_destroy(c);
  free(p); free(m); free(ms); free(id);
  return NULL;
}

int main(int argc, char **argv)
{
  int cpu[3] = {2, 1, 1}, nb = 10, i, d, x, y, z, world;
  unsigned long long s = 88172645463325252ULL;
  const double a = 4.05;
  if (argc >= 4) for (d = 0; d < 3; d++) cpu[d] = atoi(argv[1 + d]);
  if (argc >= 5) NSTEPS = atoi(argv[4]);
  if (argc >= 6) WITH_TTM = atoi(argv[5]);
  world = cpu[0] * cpu[1] * cpu[2];
  make_table(&PHI, 1.5 * 1.5, 6.5 * 6.5, f_phi, 1);
  make_table(&RHO, 1.5 * 1.5, 6.5 * 6.5, f_rho, 1);
  make_table(&EMB, 0.0, 40.0, f_emb, 0);
  for (x = 0; x < 24; x++) for (y = 0; y < 40; y++) {       /* planar synthetic EOS as in imd_amd/ttm.py */
    EOS_CVE[x * 40 + y] = 0.05 * (1.0 + (60000.0 - 1.0) * y / 39.0);
    EOS_EE[x * 40 + y] = 4000.0 * (1.0 + (245.0 - 1.0) * y / 39.0);
  }
  CELLSZ = (6.5 + 0.5) * (6.5 + 0.5);
  memset(BOX, 0, sizeof BOX);
  for (d = 0; d < 3; d++) {
    BOX[4 * d] = nb * a;
    GDIM[d] = (int) (1.0 / sqrt(CELLSZ / (BOX[4 * d] * BOX[4 * d])));      /* init_cells, imd_geom_3d.c:141-188 */
    GDIM[d] -= GDIM[d] % cpu[d];
    if (GDIM[d] / cpu[d] < 2) { fprintf(stderr, "box too small for %d ranks along %d\n", cpu[d], d); return 2; }
  }
  NATOMS = 4 * nb * nb * nb;
  POS = malloc(sizeof(double) * 3 * NATOMS); MOM = malloc(sizeof(double) * 3 * NATOMS); MASS = malloc(sizeof(double) * NATOMS);
  IDS = malloc(sizeof(int) * NATOMS);
  i = 0;
  for (x = 0; x < 2 * nb; x++) for (y = 0; y < 2 * nb; y++) for (z = 0; z < 2 * nb; z++) {
    if ((x + y + z) & 1) continue;
    int q[3] = {x, y, z};
    for (d = 0; d < 3; d++) {
      double u;
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;                 /* xorshift: deterministic thermal noise */
      u = (double) (s >> 11) / 9007199254740992.0 - 0.5;
      POS[3 * i + d] = (q[d] + 0.5) * a / 2 + 0.16 * u;
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      u = (double) (s >> 11) / 9007199254740992.0 - 0.5;
      MOM[3 * i + d] = 26.9815 * 0.25 * u;                     /* ~0.05 eV */
    }
    MASS[i] = 26.9815; IDS[i] = i; i++;
  }
  rank_t one; memset(&one, 0, sizeof one);
  one.cpu[0] = one.cpu[1] = one.cpu[2] = 1;
  rank_main(&one);
  if (one.rc) { fprintf(stderr, "single rank failed: %s\n", one.err); return 1; }
  void *group = NULL;
  if (imdgpu_local_group_create(world, &group)) { fprintf(stderr, "local group failed\n"); return 1; }
  rank_t *R = calloc(world, sizeof(rank_t));
  pthread_t *th = calloc(world, sizeof(pthread_t));
  for (i = 0; i < world; i++) {
    R[i].rank = i; R[i].group = group;
    memcpy(R[i].cpu, cpu, sizeof cpu);
    R[i].coord[0] = i / (cpu[1] * cpu[2]); R[i].coord[1] = (i / cpu[2]) % cpu[1]; R[i].coord[2] = i % cpu[2];
    pthread_create(&th[i], NULL, rank_main, &R[i]);
  }
  for (i = 0; i < world; i++) pthread_join(th[i], NULL);
  for (i = 0; i < world; i++) if (R[i].rc) { fprintf(stderr, "rank %d failed: %s\n", i, R[i].err); return 1; }
  /* compare by atom number */
  double *ref = malloc(sizeof(double) * 3 * NATOMS), maxd = 0;
  int seen = 0, k;
  for (k = 0; k < one.n; k++) memcpy(ref + 3 * one.ids[k], one.pos + 3 * k, 24);
  for (i = 0; i < world; i++)
    for (k = 0; k < R[i].n; k++, seen++)
      for (d = 0; d < 3; d++) {
        double dx = R[i].pos[3 * k + d] - ref[3 * R[i].ids[k] + d];
        dx -= BOX[4 * d] * floor(dx / BOX[4 * d] + 0.5);
        if (fabs(dx) > maxd) maxd = fabs(dx);
      }
  printf("%d x %d x %d ranks, %d atoms, %d steps%s: max |dx| vs the single-rank run %.3e A (%d atoms seen)\n", cpu[0], cpu[1], cpu[2],
         NATOMS, NSTEPS, WITH_TTM ? " with TTM" : "", maxd, seen);
  imdgpu_local_group_destroy(group);
  return (seen == NATOMS && maxd < 1e-9) ? 0 : 1;
}
