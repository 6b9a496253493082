/*
 * imd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see imd_oracle.h).
 *
 * CPU restatement of the neighbour-list force path of IMD/lasMD.  Every
 * function names the reference lines it follows (paths relative to
 * /root/reference).  Operation order follows the reference so that, built
 * with `gcc -O2 -ffp-contract=off`, results agree with the reference binary
 * to the last bit on the same cell arrangement (tests/test_oracle_vs_reference.py).
 */
#include "imd_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NNB 14                      /* half stencil incl. self, config.h:157-161 */
#define CIDX(s,i,j,k) (((i)*(s)->dim[1]+(j))*(s)->dim[2]+(k))

static void *xcalloc(size_t n, size_t sz)
{
  void *p = calloc(n ? n : 1, sz);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}

/* ------------------------------------------------------------------------ */
/* construction                                                              */
/* ------------------------------------------------------------------------ */

static void invert_box(const double b[9], double t[9])
{
  /* reciprocal vectors tbox_* = (box_y x box_z)/det etc., imd_geom_3d.c:52-110 */
  double cx[3] = { b[4]*b[8]-b[5]*b[7], b[5]*b[6]-b[3]*b[8], b[3]*b[7]-b[4]*b[6] };
  double cy[3] = { b[7]*b[2]-b[8]*b[1], b[8]*b[0]-b[6]*b[2], b[6]*b[1]-b[7]*b[0] };
  double cz[3] = { b[1]*b[5]-b[2]*b[4], b[2]*b[3]-b[0]*b[5], b[0]*b[4]-b[1]*b[3] };
  double det = b[0]*cx[0] + b[1]*cx[1] + b[2]*cx[2];
  int i;
  for (i = 0; i < 3; i++) { t[i] = cx[i]/det; t[3+i] = cy[i]/det; t[6+i] = cz[i]/det; }
}

orc_sys *orc_new(const int dim[3], int cap, int ntypes, const double box[9],
                 const int pbc[3], double cellsz, double nbl_margin,
                 double timestep)
{
  orc_sys *s = (orc_sys *) xcalloc(1, sizeof(orc_sys));
  size_t slots;
  int i;
  for (i = 0; i < 3; i++) { s->dim[i] = dim[i]; s->gdim[i] = dim[i]-2; s->pbc[i] = pbc[i]; }
  memcpy(s->box, box, sizeof(s->box));
  invert_box(s->box, s->tbox);
  s->ntypes = ntypes; s->cap = cap;
  s->nall   = dim[0]*dim[1]*dim[2];
  s->ninner = (dim[0]-2)*(dim[1]-2)*(dim[2]-2);
  s->cellsz = cellsz; s->nbl_margin = nbl_margin; s->timestep = timestep;
  slots = (size_t) s->nall * cap;
  s->n      = (int *) xcalloc(s->nall, sizeof(int));
  s->inner  = (int *) xcalloc(s->ninner, sizeof(int));
  s->nq     = (int *) xcalloc((size_t) s->ninner*NNB, sizeof(int));
  s->ort    = (double *) xcalloc(slots*3, sizeof(double));
  s->impuls = (double *) xcalloc(slots*3, sizeof(double));
  s->kraft  = (double *) xcalloc(slots*3, sizeof(double));
  s->nbl_pos= (double *) xcalloc(slots*3, sizeof(double));
  s->masse  = (double *) xcalloc(slots, sizeof(double));
  s->pot_eng= (double *) xcalloc(slots, sizeof(double));
  s->rho    = (double *) xcalloc(slots, sizeof(double));
  s->dF     = (double *) xcalloc(slots, sizeof(double));
  s->sorte  = (int *) xcalloc(slots, sizeof(int));
  s->nummer = (int *) xcalloc(slots, sizeof(int));
  s->numneighs = (int *) xcalloc(slots, sizeof(int));
  s->fdi = (int *) xcalloc(slots, sizeof(int)); s->fdj = (int *) xcalloc(slots, sizeof(int)); s->fdk = (int *) xcalloc(slots, sizeof(int));
  s->cl_off = (int *) xcalloc(s->nall+1, sizeof(int));
  s->restrictions[0] = s->restrictions[1] = s->restrictions[2] = 1.0;
  orc_make_cell_lists(s);
  return s;
}

static void free_table(orc_table *t)
{
  free(t->begin); free(t->end); free(t->step); free(t->invstep);
  free(t->len); free(t->table);
  memset(t, 0, sizeof(*t));
}

void orc_free(orc_sys *s)
{
  if (!s) return;
  free(s->n); free(s->inner); free(s->nq);
  free(s->ort); free(s->impuls); free(s->kraft); free(s->nbl_pos);
  free(s->masse); free(s->pot_eng); free(s->rho); free(s->dF);
  free(s->sorte); free(s->nummer); free(s->numneighs); free(s->fdi); free(s->fdj); free(s->fdk);
  free(s->cl_off); free(s->tl); free(s->tb);
  free_table(&s->pair); free_table(&s->rhotab); free_table(&s->embed);
  free(s);
}

static double *dupd(const double *p, int n)
{ double *q = (double *) xcalloc(n, sizeof(double)); memcpy(q, p, n*sizeof(double)); return q; }

void orc_set_table(orc_sys *s, int which, int ncols, int maxsteps,
                   const double *begin, const double *end, const double *step,
                   const double *invstep, const int *len, const double *table)
{
  orc_table *t = which == 0 ? &s->pair : which == 1 ? &s->rhotab : &s->embed;
  free_table(t);
  t->ncols = ncols; t->maxsteps = maxsteps;
  t->begin = dupd(begin, ncols); t->end = dupd(end, ncols);
  t->step  = dupd(step, ncols);  t->invstep = dupd(invstep, ncols);
  t->len   = (int *) xcalloc(ncols, sizeof(int));
  memcpy(t->len, len, ncols*sizeof(int));
  t->table = dupd(table, (maxsteps+2)*ncols);
  if (which == 0) s->have_pair = 1;
  if (which == 2) s->have_eam  = 1;
}

/* ------------------------------------------------------------------------ */
/* potential tables                                                          */
/* ------------------------------------------------------------------------ */

/* Three-point (quadratic) table interpolation in the table argument
 * (r^2 for radial tables, rho for the embedding table).
 * Reference: PAIR_INT2, potaccess.h:323-354.  Returns value and *twice* the
 * derivative w.r.t. the argument (= (1/r) dV/dr for an r^2 table). */
static inline void tab3(const orc_table *pt, int col, int inc, double arg,
                        double *p0, double *dv, double *d2v, double *chi,
                        double *istep, int *is_short)
{
  double a = arg < pt->end[col] ? arg : pt->end[col];   /* MIN(r2,end) */
  const double *ptr;
  double p1, p2;
  int k;
  a = a - pt->begin[col];
  if (a < 0) { a = 0; *is_short = 1; }
  *istep = pt->invstep[col];
  a = a * *istep;
  k = (int) a;                                          /* POS_TRUNC, potaccess.h:43 */
  *chi = a - k;
  ptr = pt->table + (size_t) k*inc + col;               /* PTR_2D(table,k,col,.,inc) */
  *p0 = ptr[0]; p1 = ptr[inc]; p2 = ptr[2*inc];
  *dv  = p1 - *p0;
  *d2v = p2 - 2 * p1 + *p0;
}

void orc_pair_int(const orc_table *pt, int col, int inc, double r2,
                  double *pot, double *grad, int *is_short)
{
  double p0, dv, d2v, chi, istep;
  tab3(pt, col, inc, r2, &p0, &dv, &d2v, &chi, &istep, is_short);
  *pot  = p0 + chi * dv + 0.5 * chi * (chi - 1) * d2v;
  *grad = 2 * istep * (dv + (chi - 0.5) * d2v);
}

/* VAL_FUNC2, potaccess.h:465-495 */
double orc_val_func(const orc_table *pt, int col, int inc, double r2, int *is_short)
{
  double p0, dv, d2v, chi, istep;
  tab3(pt, col, inc, r2, &p0, &dv, &d2v, &chi, &istep, is_short);
  return p0 + chi * dv + 0.5 * chi * (chi - 1) * d2v;
}

/* DERIV_FUNC2, potaccess.h:591-621 */
double orc_deriv_func(const orc_table *pt, int col, int inc, double r2, int *is_short)
{
  double p0, dv, d2v, chi, istep;
  tab3(pt, col, inc, r2, &p0, &dv, &d2v, &chi, &istep, is_short);
  return 2 * istep * (dv + (chi - 0.5) * d2v);
}

/* Continue the last interpolation parabola by two rows so that k+2 is always
 * readable.  Reference: init_threepoint, imd_potential.c:1260-1276. */
void orc_init_threepoint(double *table, int ncols, const int *len)
{
  int col;
  for (col = 0; col < ncols; col++) {
    double *y = table + col;
    int n = len[col];
    double a = y[(size_t)(n-1)*ncols], b = y[(size_t)(n-2)*ncols], c = y[(size_t)(n-3)*ncols];
    y[(size_t) n   *ncols] = 3*a - 3*b +   c;
    y[(size_t)(n+1)*ncols] = 6*a - 8*b + 3*c;
  }
}

/* ------------------------------------------------------------------------ */
/* geometry                                                                  */
/* ------------------------------------------------------------------------ */

/* Number of cells per direction: floor(1/sqrt(cellsz/height)) rounded down to
 * a multiple of cd (= cpu_dim [* fd_ext with TTM]).  height_x = 1/|tbox_x|^2.
 * Reference: init_cells, imd_geom_3d.c:141-188; heights imd_geom_3d.c:52-110. */
void orc_global_cell_dim(const double box[9], double cellsz, const int cd[3], int out[3])
{
  double t[9];
  int d;
  invert_box(box, t);
  for (d = 0; d < 3; d++) {
    double tt = t[3*d]*t[3*d] + t[3*d+1]*t[3*d+1] + t[3*d+2]*t[3*d+2];
    double height = 1.0 / tt;
    double scale = sqrt(1.0 * cellsz / height);
    int g = (int) (1.0 / scale);
    if (g % cd[d] != 0) g = (g / cd[d]) * cd[d];
    out[d] = g;
  }
}

/* Inner-cell list and the 14-cell half stencil in the reference's slot order.
 * Reference: make_cell_lists (NBLIST variant, AR), imd_geom_3d.c:846-975.
 * Single rank: my_coord = 0, so the global coordinate of local cell r is r-1. */
void orc_make_cell_lists(orc_sys *s)
{
  int i, j, k, l, m, n, c = 0;
  for (i = 1; i < s->dim[0]-1; i++)
    for (j = 1; j < s->dim[1]-1; j++)
      for (k = 1; k < s->dim[2]-1; k++) {
        int np = CIDX(s, i, j, k), nn = 1;
        int *q = s->nq + (size_t) c*NNB;
        s->inner[c] = np;
        q[0] = np;
        for (l = 0; l <= 1; l++)
          for (m = -l; m <= 1; m++)
            for (n = (l == 0 ? -m : -l); n <= 1; n++) {
              int r = i+l, t2 = j+m, u = k+n, qq = CIDX(s, r, t2, u), ok = 1, g[3], d;
              if (qq == np) continue;
              g[0] = r-1; g[1] = t2-1; g[2] = u-1;
              for (d = 0; d < 3; d++)
                if ((g[d] < 0 || g[d] > s->gdim[d]-1) && s->pbc[d] != 1) ok = 0;
              q[nn++] = ok ? qq : -1;
            }
        c++;
      }
}

/* Reference: cell_coord, imd_geom_3d.c:1077-1097 (global cell coordinates). */
void orc_cell_coord(const orc_sys *s, double x, double y, double z, int out[3])
{
  int d;
  for (d = 0; d < 3; d++) {
    const double *t = s->tbox + 3*d;
    int c = (int) (s->gdim[d] * (x*t[0] + y*t[1] + z*t[2]));
    if (c >= s->gdim[d]) c = s->gdim[d]-1; else if (c < 0) c = 0;
    out[d] = c;
  }
}

/* Append an atom to the cell that contains it (generation order = slot order;
 * reference: generate_fcc -> move_atom appends, imd_generate.c:370-560,
 * imd_alloc.c:111-127).  Returns the slot index or -1 if the cell is full. */
int orc_insert_atom(orc_sys *s, int nummer, int sorte, double masse,
                    const double pos[3], const double imp[3])
{
  int g[3], c, sl;
  size_t a;
  orc_cell_coord(s, pos[0], pos[1], pos[2], g);
  c = CIDX(s, g[0]+1, g[1]+1, g[2]+1);      /* local_cell_coord, imd_geom_mpi_3d.c:119-128 */
  sl = s->n[c];
  if (sl >= s->cap) return -1;
  a = (size_t) c*s->cap + sl;
  s->nummer[a] = nummer; s->sorte[a] = sorte; s->masse[a] = masse;
  memcpy(s->ort + 3*a, pos, 3*sizeof(double));
  memcpy(s->impuls + 3*a, imp, 3*sizeof(double));
  s->n[c]++;
  return sl;
}

/* ------------------------------------------------------------------------ */
/* atom bookkeeping                                                          */
/* ------------------------------------------------------------------------ */

static void copy_slot(orc_sys *s, size_t to, size_t from)
{
  /* copy_atom_cell_cell, imd_alloc.c:136-... (fields of the EAM2+NBL+TTM build) */
  memcpy(s->ort+3*to,     s->ort+3*from,     3*sizeof(double));
  memcpy(s->impuls+3*to,  s->impuls+3*from,  3*sizeof(double));
  memcpy(s->kraft+3*to,   s->kraft+3*from,   3*sizeof(double));
  memcpy(s->nbl_pos+3*to, s->nbl_pos+3*from, 3*sizeof(double));
  s->masse[to] = s->masse[from]; s->pot_eng[to] = s->pot_eng[from];
  s->rho[to] = s->rho[from];     s->dF[to] = s->dF[from];
  s->sorte[to] = s->sorte[from]; s->nummer[to] = s->nummer[from];
  s->numneighs[to] = s->numneighs[from];
  s->fdi[to] = s->fdi[from]; s->fdj[to] = s->fdj[from]; s->fdk[to] = s->fdk[from];   /* imd_alloc.c:210-212 */
}

/* Wrap atoms of inner cells back into the box, direction by direction.
 * Reference: do_boundaries, imd_main_3d.c:2018-2110. */
void orc_do_boundaries(orc_sys *s)
{
  int c, l, d;
  for (c = 0; c < s->ninner; c++) {
    int cell = s->inner[c];
    for (d = 0; d < 3; d++) {
      const double *t = s->tbox + 3*d, *b = s->box + 3*d;
      if (s->pbc[d] != 1) continue;
      for (l = 0; l < s->n[cell]; l++) {
        double *x = s->ort + 3*((size_t) cell*s->cap + l);
        double i = -floor(x[0]*t[0] + x[1]*t[1] + x[2]*t[2]);
        x[0] += i * b[0];
        x[1] += i * b[1];
        x[2] += i * b[2];
      }
    }
  }
}

/* Re-bin atoms: visit inner cells x-outer/z-inner, and for every atom whose
 * cell changed append it to the new cell and fill the hole with the last atom
 * of the old one (so the hole's new occupant is examined next).
 * Reference: fix_cells, imd_fix_cells_3d.c:43-306; move_atom, imd_alloc.c:111-127. */
void orc_fix_cells(orc_sys *s)
{
  int i, j, k;
  orc_do_boundaries(s);
  for (i = 1; i < s->dim[0]-1; i++)
    for (j = 1; j < s->dim[1]-1; j++)
      for (k = 1; k < s->dim[2]-1; k++) {
        int p = CIDX(s, i, j, k), l = 0;
        while (l < s->n[p]) {
          size_t a = (size_t) p*s->cap + l;
          int g[3];
          orc_cell_coord(s, s->ort[3*a], s->ort[3*a+1], s->ort[3*a+2], g);
          if (g[0]+1 == i && g[1]+1 == j && g[2]+1 == k) { l++; continue; }
          {
            int q = CIDX(s, g[0]+1, g[1]+1, g[2]+1);
            if (s->n[q] >= s->cap) { fprintf(stderr, "oracle: cell capacity exceeded\n"); abort(); }
            copy_slot(s, (size_t) q*s->cap + s->n[q], a);
            s->n[q]++;
            s->n[p]--;
            if (l < s->n[p]) copy_slot(s, a, (size_t) p*s->cap + s->n[p]);
          }
        }
      }
  s->have_valid_nbl = 0;
}

/* ------------------------------------------------------------------------ */
/* ghost layer (single rank: in-process copies)                              */
/* ------------------------------------------------------------------------ */

static void copy_cell_pos(orc_sys *s, int from, int to, const double v[3])
{
  /* copy_cell, imd_comm_force_3d.c:726-818 */
  int i;
  s->n[to] = s->n[from];
  for (i = 0; i < s->n[to]; i++) {
    size_t a = (size_t) from*s->cap + i, b = (size_t) to*s->cap + i;
    s->ort[3*b]   = s->ort[3*a]   + v[0];
    s->ort[3*b+1] = s->ort[3*a+1] + v[1];
    s->ort[3*b+2] = s->ort[3*a+2] + v[2];
    s->sorte[b] = s->sorte[a];
    s->numneighs[b] = s->numneighs[a];
    s->nummer[b] = s->nummer[a];
  }
}

static void copy_cell_dF(orc_sys *s, int from, int to)
{
  /* copy_dF, imd_comm_force_3d.c (EAM2 callbacks) */
  int i;
  for (i = 0; i < s->n[to]; i++)
    s->dF[(size_t) to*s->cap + i] = s->dF[(size_t) from*s->cap + i];
}

/* Plimpton forward exchange z, y, x for cpu_dim = 1 1 1: ghost cells receive
 * copies of the opposite inner layer, shifted by the box vector where the
 * direction is periodic.  In the reference's AR mode the x=0 wall is not
 * filled (full_shell=0); full_shell=1 fills it too (the non-AR variant).
 * Reference: send_cells, imd_comm_force_3d.c:222-396. */
void orc_send_cells(orc_sys *s, int what, int full_shell)
{
  int i, j, d;
  double up[3] = {0,0,0}, dn[3] = {0,0,0}, no[3] = {0,0,0}, so[3] = {0,0,0},
         ea[3] = {0,0,0}, we[3] = {0,0,0};
  const int *D = s->dim;
  for (d = 0; d < 3; d++) {
    if (s->pbc[0] == 1) { ea[d] = s->box[d];   we[d] = -s->box[d]; }
    if (s->pbc[1] == 1) { no[d] = s->box[3+d]; so[d] = -s->box[3+d]; }
    if (s->pbc[2] == 1) { up[d] = s->box[6+d]; dn[d] = -s->box[6+d]; }
  }
#define CP(fi,fj,fk,ti,tj,tk,v) do { \
    if (what == 0) copy_cell_pos(s, CIDX(s,fi,fj,fk), CIDX(s,ti,tj,tk), v); \
    else           copy_cell_dF (s, CIDX(s,fi,fj,fk), CIDX(s,ti,tj,tk)); } while (0)
  for (i = 1; i < D[0]-1; i++)
    for (j = 1; j < D[1]-1; j++) {
      CP(i, j, 1,      i, j, D[2]-1, up);
      CP(i, j, D[2]-2, i, j, 0,      dn);
    }
  for (i = 1; i < D[0]-1; i++)
    for (j = 0; j < D[2]; j++) {
      CP(i, 1,      j, i, D[1]-1, j, no);
      CP(i, D[1]-2, j, i, 0,      j, so);
    }
  for (i = 0; i < D[1]; i++)
    for (j = 0; j < D[2]; j++) {
      CP(1, i, j, D[0]-1, i, j, ea);
      if (full_shell) CP(D[0]-2, i, j, 0, i, j, we);
    }
#undef CP
}

static void add_cell(orc_sys *s, int what, int from, int to)
{
  /* add_rho / add_forces, imd_comm_force_3d.c:1008-1060 */
  int i;
  for (i = 0; i < s->n[to]; i++) {
    size_t a = (size_t) from*s->cap + i, b = (size_t) to*s->cap + i;
    if (what == 0) s->rho[b] += s->rho[a];
    else {
      s->kraft[3*b]   += s->kraft[3*a];
      s->kraft[3*b+1] += s->kraft[3*a+1];
      s->kraft[3*b+2] += s->kraft[3*a+2];
      s->pot_eng[b]   += s->pot_eng[a];
      s->numneighs[b] += s->numneighs[a];
    }
  }
}

/* Reverse exchange x, y, z: what was scattered onto ghost atoms is added back
 * to the atoms they are images of.  Reference: send_forces,
 * imd_comm_force_3d.c:569-715 (AR: only the x=dim-1 wall carries data). */
void orc_send_forces(orc_sys *s, int what)
{
  int i, j;
  const int *D = s->dim;
  for (i = 0; i < D[1]; i++)
    for (j = 0; j < D[2]; j++)
      add_cell(s, what, CIDX(s, D[0]-1, i, j), CIDX(s, 1, i, j));
  for (i = 1; i < D[0]-1; i++)
    for (j = 0; j < D[2]; j++) {
      add_cell(s, what, CIDX(s, i, 0, j),      CIDX(s, i, D[1]-2, j));
      add_cell(s, what, CIDX(s, i, D[1]-1, j), CIDX(s, i, 1, j));
    }
  for (i = 1; i < D[0]-1; i++)
    for (j = 1; j < D[1]-1; j++) {
      add_cell(s, what, CIDX(s, i, j, 0),      CIDX(s, i, j, D[2]-2));
      add_cell(s, what, CIDX(s, i, j, D[2]-1), CIDX(s, i, j, 1));
    }
}

/* ------------------------------------------------------------------------ */
/* Verlet list                                                               */
/* ------------------------------------------------------------------------ */

/* Half neighbour list in the reference's numbering: neighbour atoms are
 * n = cl_off[cell]+slot over ALL cells (ghosts included); list owners run over
 * the atoms of the inner cells in `cells` order.  Entry test is
 * dx*dx+dy*dy+dz*dz < cellsz with that summation order (SPROD3D, makros.h:361).
 * Reference: make_nblist, imd_forces_nbl.c:136-287. */
void orc_make_nblist(orc_sys *s)
{
  int c, i, k, at = 0, n = 0, tn = 0;

  for (c = 0; c < s->ninner; c++) {             /* reference positions, :143-155 */
    int cell = s->inner[c];
    for (i = 0; i < s->n[cell]; i++) {
      size_t a = (size_t) cell*s->cap + i;
      memcpy(s->nbl_pos+3*a, s->ort+3*a, 3*sizeof(double));
    }
  }
  for (k = 0; k < s->nall; k++) { s->cl_off[k] = at; at += s->n[k]; }   /* :164-170 */
  s->cl_off[s->nall] = at;

  if (at + 1 > s->tl_cap) {
    free(s->tl);
    s->tl_cap = at + 1 + at/2;
    s->tl = (int *) xcalloc(s->tl_cap, sizeof(int));
  }
  s->tl[0] = 0;
  for (c = 0; c < s->ninner; c++) {
    int c1 = s->inner[c];
    const int *q = s->nq + (size_t) c*NNB;
    for (i = 0; i < s->n[c1]; i++) {
      const double *x1 = s->ort + 3*((size_t) c1*s->cap + i);
      int m;
      for (m = 0; m < NNB; m++) {
        int c2 = q[m], j, jstart;
        if (c2 < 0) continue;
        jstart = (c2 == c1) ? i+1 : 0;
        if (tn + s->n[c2] > s->tb_cap) {
          s->tb_cap = 2*s->tb_cap + s->n[c2] + 1024;
          s->tb = (int *) realloc(s->tb, (size_t) s->tb_cap*sizeof(int));
          if (!s->tb) abort();
        }
        for (j = jstart; j < s->n[c2]; j++) {
          const double *x2 = s->ort + 3*((size_t) c2*s->cap + j);
          double dx = x2[0]-x1[0], dy = x2[1]-x1[1], dz = x2[2]-x1[2];
          double r2 = dx*dx + dy*dy + dz*dz;
          if (r2 < s->cellsz) s->tb[tn++] = s->cl_off[c2] + j;
        }
      }
      s->tl[++n] = tn;
    }
  }
  s->tb_len = tn;
  s->have_valid_nbl = 1;
  s->nbl_count++;
}

/* Reference: check_nblist, imd_forces_nbl.c:1062-1092. */
void orc_check_nblist(orc_sys *s)
{
  double max1 = 0.0, h;
  int c, i;
  for (c = 0; c < s->ninner; c++) {
    int cell = s->inner[c];
    for (i = 0; i < s->n[cell]; i++) {
      size_t a = (size_t) cell*s->cap + i;
      double dx = s->ort[3*a]-s->nbl_pos[3*a], dy = s->ort[3*a+1]-s->nbl_pos[3*a+1],
             dz = s->ort[3*a+2]-s->nbl_pos[3*a+2];
      double r2 = dx*dx + dy*dy + dz*dz;
      if (r2 > max1) max1 = r2;
    }
  }
  h = 0.5*s->nbl_margin;
  if (max1 > h*h) s->have_valid_nbl = 0;
}

/* ------------------------------------------------------------------------ */
/* force loop                                                                */
/* ------------------------------------------------------------------------ */

/* flat reference index -> (cell, slot): cl_num[] of the reference */
static int *build_cl_num(const orc_sys *s)
{
  int *cl_num = (int *) xcalloc(s->cl_off[s->nall] + 1, sizeof(int));
  int k, i, n = 0;
  for (k = 0; k < s->nall; k++) for (i = 0; i < s->n[k]; i++) cl_num[n++] = k;
  return cl_num;
}

/* Reference: calc_forces, imd_forces_nbl.c:295-1054 (PAIR + EAM2 [+TTM] build,
 * one rank).  Steps: rebuild cells/list if invalid, ghost positions, zero,
 * pass 1 (pair + host density + TTM neighbour count), reverse-add rho,
 * pass 2 (embedding), forward dF, pass 3 (EAM force), reverse-add forces. */
void orc_calc_forces(orc_sys *s, int steps)
{
  int c, i, k, n, inc = s->ntypes*s->ntypes, idummy = 0;
  int *cl_num;
  (void) steps;

  if (!s->have_valid_nbl) orc_fix_cells(s);                       /* :300-308 */
  orc_send_cells(s, 0, 0);                                        /* :311 */
  if (!s->have_valid_nbl) orc_make_nblist(s);                     /* :315 */

  s->tot_pot_energy = 0.0; s->virial = 0.0; s->is_short = 0;      /* :319-328 */
  for (k = 0; k < s->nall; k++)                                   /* :333-404 */
    for (i = 0; i < s->n[k]; i++) {
      size_t a = (size_t) k*s->cap + i;
      s->kraft[3*a] = s->kraft[3*a+1] = s->kraft[3*a+2] = 0.0;
      s->pot_eng[a] = 0.0; s->rho[a] = 0.0; s->numneighs[a] = 0;
    }
  cl_num = build_cl_num(s);

  /* pass 1, :418-672 */
  n = 0;
  for (c = 0; c < s->ninner; c++) {
    int cell = s->inner[c];
    for (i = 0; i < s->n[cell]; i++) {
      size_t a = (size_t) cell*s->cap + i;
      double fx = 0.0, fy = 0.0, fz = 0.0, ee = 0.0, eam_r = 0.0;
      double x1 = s->ort[3*a], y1 = s->ort[3*a+1], z1 = s->ort[3*a+2];
      int it = s->sorte[a], nbttm = 0, m;
      for (m = s->tl[n]; m < s->tl[n+1]; m++) {
        int c2 = cl_num[s->tb[m]], j = s->tb[m] - s->cl_off[c2];
        size_t b = (size_t) c2*s->cap + j;
        double dx = s->ort[3*b]-x1, dy = s->ort[3*b+1]-y1, dz = s->ort[3*b+2]-z1;
        double r2 = dx*dx + dy*dy + dz*dz, rho_h = 0.0;
        int jt = s->sorte[b], col = it*s->ntypes + jt, col2 = jt*s->ntypes + it;

        if (s->count_ttm_neigh && r2 <= s->pair.end[0])            /* :472-482 */
          if (it == 0 && jt == 0) { nbttm++; s->numneighs[b]++; }

        if (s->have_pair && r2 <= s->pair.end[col]) {              /* :489-544 */
          double pot, grad, gx, gy, gz;
          orc_pair_int(&s->pair, col, inc, r2, &pot, &grad, &s->is_short);
          s->tot_pot_energy += pot;
          gx = dx * grad; gy = dy * grad; gz = dz * grad;
          s->kraft[3*b] -= gx; s->kraft[3*b+1] -= gy; s->kraft[3*b+2] -= gz;
          fx += gx; fy += gy; fz += gz;
          pot *= 0.5;
          ee += pot;
          s->pot_eng[b] += pot;
          s->virial -= r2 * grad;
        }
        if (s->have_eam) {                                         /* :548-573 */
          if (r2 < s->rhotab.end[col]) {
            rho_h = orc_val_func(&s->rhotab, col, inc, r2, &s->is_short);
            eam_r += rho_h;
          }
          if (it == jt) {
            if (r2 < s->rhotab.end[col]) s->rho[b] += rho_h;
          } else if (r2 < s->rhotab.end[col2]) {
            rho_h = orc_val_func(&s->rhotab, col2, inc, r2, &s->is_short);
            s->rho[b] += rho_h;
          }
        }
      }
      s->kraft[3*a] += fx; s->kraft[3*a+1] += fy; s->kraft[3*a+2] += fz;   /* :630-669 */
      s->pot_eng[a] += ee;
      s->rho[a] += eam_r;
      s->numneighs[a] += nbttm;
      n++;
    }
  }

  if (s->have_eam) {
    orc_send_forces(s, 0);                                         /* :745 */
    for (c = 0; c < s->ninner; c++) {                              /* pass 2, :748-783 */
      int cell = s->inner[c];
      for (i = 0; i < s->n[cell]; i++) {
        size_t a = (size_t) cell*s->cap + i;
        double pot;
        orc_pair_int(&s->embed, s->sorte[a], s->ntypes, s->rho[a], &pot, &s->dF[a], &idummy);
        s->pot_eng[a] += pot;
        s->tot_pot_energy += pot;
      }
    }
    orc_send_cells(s, 1, 0);                                       /* :786 */

    n = 0;                                                         /* pass 3, :789-985 */
    for (c = 0; c < s->ninner; c++) {
      int cell = s->inner[c];
      for (i = 0; i < s->n[cell]; i++) {
        size_t a = (size_t) cell*s->cap + i;
        double fx = 0.0, fy = 0.0, fz = 0.0;
        double x1 = s->ort[3*a], y1 = s->ort[3*a+1], z1 = s->ort[3*a+2];
        int it = s->sorte[a], m;
        for (m = s->tl[n]; m < s->tl[n+1]; m++) {
          int c2 = cl_num[s->tb[m]], j = s->tb[m] - s->cl_off[c2];
          size_t b = (size_t) c2*s->cap + j;
          double dx = s->ort[3*b]-x1, dy = s->ort[3*b+1]-y1, dz = s->ort[3*b+2]-z1;
          double r2 = dx*dx + dy*dy + dz*dz;
          int jt = s->sorte[b], col1 = jt*s->ntypes + it, col2 = it*s->ntypes + jt;
          if (r2 < s->rhotab.end[col1] || r2 < s->rhotab.end[col2]) {
            double ri = orc_deriv_func(&s->rhotab, col1, inc, r2, &s->is_short), rj, grad, gx, gy, gz;
            rj = (col1 == col2) ? ri : orc_deriv_func(&s->rhotab, col2, inc, r2, &s->is_short);
            grad = 0.5 * (s->dF[a]*rj + s->dF[b]*ri);              /* :877 */
            gx = dx * grad; gy = dy * grad; gz = dz * grad;
            s->kraft[3*b] -= gx; s->kraft[3*b+1] -= gy; s->kraft[3*b+2] -= gz;
            fx += gx; fy += gy; fz += gz;
            s->virial -= dx*gx + dy*gy + dz*gz;                    /* :943 */
          }
        }
        s->kraft[3*a] += fx; s->kraft[3*a+1] += fy; s->kraft[3*a+2] += fz;
        n++;
      }
    }
  }
  orc_send_forces(s, 1);                                           /* :1052 */
  free(cl_num);
}

/* ------------------------------------------------------------------------ */
/* integrator                                                                */
/* ------------------------------------------------------------------------ */

/* Leap-frog NVE: F *= restrictions; Ekin += (p_old^2+p_new^2)/(4m);
 * p += dt F; x += dt/m p.  Reference: move_atoms_nve, imd_integrate.c:35-561
 * (plain NVE branch :147,:209-213,:261-263,:396,:415-419). */
void orc_move_atoms_nve(orc_sys *s)
{
  int c, i;
  s->tot_kin_energy = 0.0;
  for (c = 0; c < s->ninner; c++) {
    int cell = s->inner[c];
    for (i = 0; i < s->n[cell]; i++) {
      size_t a = (size_t) cell*s->cap + i;
      double *p = s->impuls + 3*a, *f = s->kraft + 3*a, *x = s->ort + 3*a;
      double e1 = p[0]*p[0] + p[1]*p[1] + p[2]*p[2], e2, tmp;
      f[0] *= s->restrictions[0]; f[1] *= s->restrictions[1]; f[2] *= s->restrictions[2];
      p[0] += s->timestep * f[0];
      p[1] += s->timestep * f[1];
      p[2] += s->timestep * f[2];
      e2 = p[0]*p[0] + p[1]*p[1] + p[2]*p[2];
      s->tot_kin_energy += (e1 + e2) / (4 * s->masse[a]);
      tmp = s->timestep / s->masse[a];
      x[0] += tmp * p[0];
      x[1] += tmp * p[1];
      x[2] += tmp * p[2];
    }
  }
}

/* Test helper: install a Verlet list taken from a reference probe dump (same
 * cell arrangement), so the force passes can be checked on their own. */
void orc_set_nblist(orc_sys *s, const int *tl, int ntl, const int *tb, int ntb)
{
  int k, at = 0;
  for (k = 0; k < s->nall; k++) { s->cl_off[k] = at; at += s->n[k]; }
  s->cl_off[s->nall] = at;
  free(s->tl); free(s->tb);
  s->tl = (int *) xcalloc(ntl, sizeof(int)); s->tl_cap = ntl;
  s->tb = (int *) xcalloc(ntb + 1, sizeof(int)); s->tb_cap = ntb + 1;
  memcpy(s->tl, tl, (size_t) ntl*sizeof(int));
  memcpy(s->tb, tb, (size_t) ntb*sizeof(int));
  s->tb_len = ntb;
  s->have_valid_nbl = 1;
}
