/* ref_probe_tricub.c -- TEST INFRASTRUCTURE.  Harness around the reference's OWN tricubic table look-up.
 *
 * fitDL (imd_ttm_3D.c:2919-2968, imd_ttm.c:2867-2916) takes the Drude-Lorentz parameters of an FD cell from five
 * tables over (rho, lg Te, lg Ti) through read_tricub_interp / do_tricubinterp (imd_interpol.c:959-1084, :740-957,
 * tricub_get_coeff :1088, tricub_eval :1108, tricub_get_coeff_stacked :1150 with the 64x64 matrix of
 * tricub_coeffmat.h).  Neither TTM file compiles with -DFDTD as shipped (imd_ttm_3D.c:3022 a comma for a semicolon;
 * imd_ttm.c:2889 indexes the 1D lattice with three subscripts), so fitDL itself cannot be run; the interpolation it
 * calls compiles from its own source file and is pinned here at FUNCTION level: this translation unit instantiates
 * the reference's globals (it is the MAIN of this little program), lets the reference READ a table file with its own
 * read_tricub_interp and evaluates its own do_tricubinterp on a list of query points.  Nothing of the reference is
 * replaced: no stand-in header, library or function (libnn.a is the archive the reference ships).
 *
 * usage: ref_probe_tricub <table.txt> <queries.bin> <out.bin>
 *   queries.bin: int n, then n x (x, y, z) doubles;  out.bin: header (is js ks as doubles, the six range values, dx dy dz),
 *   then n doubles.  The "ERROR: Interpolation table exceeded" lines of out-of-range queries go to stdout.
 */
#define MAIN
#include "imd.h"

int main(int argc, char **argv)
{
  struct tricubinterp tri;
  FILE *f;
  int n, q;
  double *xyz, *out, hd[12];
  MPI_Init(&argc, &argv);
  if (argc < 4) { fprintf(stderr, "usage: ref_probe_tricub <table.txt> <queries.bin> <out.bin>\n"); return 2; }
  MPI_Comm_rank(MPI_COMM_WORLD, &myid);
  cpugrid = MPI_COMM_WORLD;                       /* read_tricub_interp broadcasts over cpugrid */
  read_tricub_interp(&tri, argv[1]);
  f = fopen(argv[2], "rb");
  if (!f || fread(&n, sizeof(int), 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  xyz = (double *) malloc(3 * (size_t) n * sizeof(double));
  out = (double *) malloc((size_t) n * sizeof(double));
  if (fread(xyz, sizeof(double), 3 * (size_t) n, f) != 3 * (size_t) n) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);
  for (q = 0; q < n; q++) out[q] = do_tricubinterp(&tri, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]);
  hd[0] = tri.is; hd[1] = tri.js; hd[2] = tri.ks;
  hd[3] = tri.xmin; hd[4] = tri.xmax; hd[5] = tri.ymin; hd[6] = tri.ymax; hd[7] = tri.zmin; hd[8] = tri.zmax;
  hd[9] = tri.dx; hd[10] = tri.dy; hd[11] = tri.dz;
  f = fopen(argv[3], "wb");
  fwrite(hd, sizeof(double), 12, f);
  fwrite(out, sizeof(double), (size_t) n, f);
  fclose(f);
  MPI_Finalize();
  return 0;
}
