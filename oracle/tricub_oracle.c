/* tricub_oracle.c -- TEST INFRASTRUCTURE (CPU oracle).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call this; the product path (libimdgpu.so) never does.
 *
 * Restatement of the tricubic table look-up behind fitDL, the per-cell Drude-Lorentz parameters of the Maxwell solver:
 *   read_tricub_interp   /root/reference/imd_interpol.c:959-1084   (table layout, dx = (xmax-xmin)/is)
 *   do_tricubinterp      :740-957     (cell search, un-normalised central differences clamped at the table's ends)
 *   tricub_get_coeff     :1088-1106   (the 64-vector: f, dfdx, dfdy, dfdz, d2fdxdy, d2fdxdz, d2fdydz, d3fdxdydz x 8 corners)
 *   tricub_get_coeff_stacked :1150-1164  (a = A x, A the 64x64 integer matrix of Lekien & Marsden, tricub_coeffmat.h)
 *   tricub_eval          :1108-1148   (the powers x^i, y^j, z^k pass through INT variables there, see below)
 *   fitDL                /root/reference/imd_ttm_3D.c:2919-2968 (= imd_ttm.c:2867-2916)
 * Pinned bit for bit by tests/golden/tricub.npz, outputs of the reference's own functions (oracle/ref_probe_tricub.c);
 * fitDL's own dozen lines cannot be run (neither TTM file compiles with -DFDTD as shipped) and are restated from the text.
 *
 * The matrix A is not copied: it is the tensor product of the 1D cubic Hermite basis, A[i+4j+16k][8 t + c] =
 * H[i][cx] H[j][cy] H[k][cz] with H the coefficients of (f0, f1, f0', f1') in powers of t, column = corner bit + 2 x
 * (derivative taken along that axis); the golden vectors include queries in which all 64 coefficients carry weight.
 *
 * What tricub_eval really computes: `t4=pow(x,i); t5=pow(y,j); t6=pow(z,k); s2=t4*t5*t6;` with t4..t6 declared int --
 * every power is truncated towards zero before use.  For scaled coordinates inside (-1, 1), which is what a table with
 * nodes at its cell centres produces, only i=j=k=0 survives and the look-up returns a[0] = f[i1][j1][k1], the node below;
 * at the upper end of an axis (i2 == i1, scaled coordinate set to 1.0) all four powers are 1.  Reproduced as is.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_tricub {
  int is, js, ks;
  double xmin, xmax, ymin, ymax, zmin, zmax, dx, dy, dz;
  double *x, *y, *z, *f;                 /* f[(i*js + j)*ks + k] = the reference's f[i][j][k] */
} orc_tricub;

orc_tricub *orc_tricub_new(int is, int js, int ks, const double *range, const double *x, const double *y, const double *z,
                           const double *f)
{
  orc_tricub *t = (orc_tricub *) calloc(1, sizeof(orc_tricub));
  size_t n = (size_t) is * js * ks;
  t->is = is; t->js = js; t->ks = ks;
  t->xmin = range[0]; t->xmax = range[1]; t->ymin = range[2]; t->ymax = range[3]; t->zmin = range[4]; t->zmax = range[5];
  t->dx = (t->xmax - t->xmin) / ((double) is);                         /* :1008-1010 */
  t->dy = (t->ymax - t->ymin) / ((double) js);
  t->dz = (t->zmax - t->zmin) / ((double) ks);
  t->x = (double *) malloc(is * sizeof(double)); memcpy(t->x, x, is * sizeof(double));
  t->y = (double *) malloc(js * sizeof(double)); memcpy(t->y, y, js * sizeof(double));
  t->z = (double *) malloc(ks * sizeof(double)); memcpy(t->z, z, ks * sizeof(double));
  t->f = (double *) malloc(n * sizeof(double)); memcpy(t->f, f, n * sizeof(double));
  return t;
}

void orc_tricub_free(orc_tricub *t)
{
  if (!t) return;
  free(t->x); free(t->y); free(t->z); free(t->f); free(t);
}

static const int HERM[4][4] = {{1, 0, 0, 0}, {0, 0, 1, 0}, {-3, 3, -2, -1}, {2, -2, 1, 1}};

static int imax2(int a, int b) { return a > b ? a : b; }
static int imin2(int a, int b) { return a < b ? a : b; }

double orc_tricub_eval(const orc_tricub *t, double xp, double yp, double zp)
{
  int lo[3], hi[3], n[3], ax, c, ty, q, i, j, k;
  int pm[3][2][2];                       /* [axis][corner bit][0 = min, 1 = max]: the clamped neighbours of the differences */
  int nd[3][2];                          /* [axis][corner bit] = i1 / i2 */
  double X[64], a[64], s[3], s4 = 0.0;
  const double *ax_nodes[3];
  if (xp > t->xmax || xp < t->xmin || yp > t->ymax || yp < t->ymin || zp < t->zmin || zp > t->zmax) return -1;      /* :752-757 */
  n[0] = t->is; n[1] = t->js; n[2] = t->ks;
  lo[0] = (int) floor((xp - (t->xmin + t->dx / 2)) / (t->xmax - t->xmin) * (double) t->is);                            /* :772-774 */
  lo[1] = (int) floor((yp - (t->ymin + t->dy / 2)) / (t->ymax - t->ymin) * (double) t->js);
  lo[2] = (int) floor((zp - (t->zmin + t->dz / 2)) / (t->zmax - t->zmin) * (double) t->ks);
  ax_nodes[0] = t->x; ax_nodes[1] = t->y; ax_nodes[2] = t->z;
  for (ax = 0; ax < 3; ax++) {
    lo[ax] = imax2(lo[ax], 0);
    hi[ax] = imax2(imin2(lo[ax] + 1, n[ax] - 1), 0);                                                                     /* :780-792 */
    nd[ax][0] = lo[ax]; nd[ax][1] = hi[ax];
    for (c = 0; c < 2; c++) {
      pm[ax][c][0] = imax2(nd[ax][c] - 1, 0);                                                                            /* :795-808 */
      pm[ax][c][1] = imin2(nd[ax][c] + 1, n[ax] - 1);
    }
    {
      const double x1 = ax_nodes[ax][lo[ax]], x2 = ax_nodes[ax][hi[ax]], p = ax == 0 ? xp : ax == 1 ? yp : zp;
      s[ax] = (x2 - x1 == 0) ? 1.0 : (p - x1) / (x2 - x1);                                                               /* :940-951 */
    }
  }
#define FV(i, j, k) t->f[((size_t) (i) * t->js + (j)) * t->ks + (k)]
  /* :825-932: per corner the value and the differences, nested z( y( x(.) ) ) where an axis is differentiated */
  for (ty = 0; ty < 8; ty++) {
    static const int DER[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    for (c = 0; c < 8; c++) {
      const int b[3] = {c & 1, (c >> 1) & 1, (c >> 2) & 1};
      double vz[2];
      int kz, nz = DER[ty][2] ? 2 : 1;
      for (kz = 0; kz < nz; kz++) {
        const int kk = DER[ty][2] ? pm[2][b[2]][1 - kz] : nd[2][b[2]];       /* kz = 0: kmax, 1: kmin */
        double vy[2];
        int jy, ny = DER[ty][1] ? 2 : 1;
        for (jy = 0; jy < ny; jy++) {
          const int jj = DER[ty][1] ? pm[1][b[1]][1 - jy] : nd[1][b[1]];
          vy[jy] = DER[ty][0] ? (FV(pm[0][b[0]][1], jj, kk) - FV(pm[0][b[0]][0], jj, kk)) : FV(nd[0][b[0]], jj, kk);
        }
        vz[kz] = DER[ty][1] ? (vy[0] - vy[1]) : vy[0];
      }
      X[8 * ty + c] = DER[ty][2] ? (vz[0] - vz[1]) : vz[0];
    }
  }
#undef FV
  /* tricub_get_coeff_stacked, :1150-1164: a[n] = sum over q = 0..63 in order of A[n][q] * X[q] */
  for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) for (k = 0; k < 4; k++) {
    static const int DER[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    double acc = 0.0;
    for (q = 0; q < 64; q++) {
      const int tq = q >> 3, cq = q & 7;
      const int A = HERM[i][(cq & 1) + 2 * DER[tq][0]] * HERM[j][((cq >> 1) & 1) + 2 * DER[tq][1]] * HERM[k][((cq >> 2) & 1) + 2 * DER[tq][2]];
      acc += A * X[q];
    }
    a[i + 4 * j + 16 * k] = acc;
  }
  /* tricub_eval, :1108-1148: i outermost, k innermost; the powers go through int */
  for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) for (k = 0; k < 4; k++) {
    const int t4 = pow(s[0], i), t5 = pow(s[1], j), t6 = pow(s[2], k);
    const double s2 = t4 * t5 * t6;
    s4 = s4 + a[i + 4 * j + 16 * k] * s2;
  }
  return 0.0 + s4;
}

void orc_tricub_eval_n(const orc_tricub *t, int n, const double *xyz, double *out)
{
  int q;
  for (q = 0; q < n; q++) out[q] = orc_tricub_eval(t, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]);
}

/* fitDL for one FD cell: tab[0..4] = DL1..DL5; DL[0..5] out; returns -1 when a look-up left a table (DL[q] == -1) */
int orc_fitdl(const orc_tricub *const *tab, double temp, double md_temp, double dens, double ne, double *DL)
{
  const double hbarev = 6.582119487171653e-16;                          /* globals.h:485 */
  const double ECHARGE = 1.602176620000000e-19, ECONST = 8.841941282883075e-12, EMASS = 9.109383560000000e-31;   /* globals.h:324-329 */
  const double Te = temp > 0.0259 ? temp : 0.0259, Ti = md_temp > 0.0259 ? md_temp : 0.0259;
  double lgTe = log10(Te), lgTi = log10(Ti);
  int q;
  lgTi = lgTi > -1.587504 ? lgTi : -1.587504;
  lgTe = lgTe > -1.587504 ? lgTe : -1.587504;
  lgTe = lgTe < tab[0]->ymax ? lgTe : tab[0]->ymax;
  lgTi = lgTi < tab[0]->zmax ? lgTi : tab[0]->zmax;
  DL[0] = orc_tricub_eval(tab[0], dens, lgTe, lgTi);
  for (q = 1; q < 5; q++) DL[q] = orc_tricub_eval(tab[q], dens, lgTe, lgTi) / hbarev;
  DL[5] = sqrt(ne * ECHARGE * ECHARGE / ECONST / EMASS);
  for (q = 0; q < 6; q++) if (DL[q] == -1) return -1;
  return 0;
}
