/* ref_probe_tri.c -- TEST INFRASTRUCTURE.  Which diagonal does the reference's triangulation put into every cell of a table grid?
 *
 * The reference interpolates its EOS tables with nn-c (lpi_interpolate_point on the triangulation delaunay_build returns;
 * imd_interpol.c:1263-1267, imd_ttm_3D.c:2722,2759), shipped only as the binary /root/reference/nn_interpol/libnn.a with its
 * two headers.  This little program links that archive where it lies and asks it directly: the nodes of a rectilinear grid
 * go through the reference's own delaunay_build, the triangles come back, and every cell reports its diagonal.
 * Nothing of the reference is replaced or copied.
 *
 * usage: ref_probe_tri <grid.txt>      grid.txt: nx ny, then the nx x-coordinates, then the ny y-coordinates
 * output: nx-1 lines of ny-1 characters: '/' = diagonal (i+1,j)-(i,j+1), '\' = diagonal (i,j)-(i+1,j+1), '?' = cell not split
 */
#include <stdio.h>
#include <stdlib.h>
#include "nn.h"
#include "delaunay.h"

int main(int argc, char **argv)
{
  int nx, ny, i, j, t;
  double *x, *y;
  point *p;
  char *map;
  delaunay *d;
  FILE *f = argc > 1 ? fopen(argv[1], "r") : NULL;
  if (!f || fscanf(f, "%d %d", &nx, &ny) != 2) { fprintf(stderr, "usage: ref_probe_tri <grid.txt>\n"); return 2; }
  x = malloc(sizeof(double) * nx); y = malloc(sizeof(double) * ny);
  for (i = 0; i < nx; i++) if (fscanf(f, "%lf", &x[i]) != 1) return 2;
  for (j = 0; j < ny; j++) if (fscanf(f, "%lf", &y[j]) != 1) return 2;
  fclose(f);
  p = malloc(sizeof(point) * nx * ny);
  map = calloc((size_t) (nx - 1) * (ny - 1), 1);
  for (i = 0; i < nx; i++)
    for (j = 0; j < ny; j++) { p[i * ny + j].x = x[i]; p[i * ny + j].y = y[j]; p[i * ny + j].z = 0; }      /* x outer, y inner like nn_read_table */
  d = delaunay_build(nx * ny, p, 0, NULL, 0, NULL);
  for (t = 0; t < d->ntriangles; t++) {
    const int *v = d->triangles[t].vids;
    int a, b;
    for (a = 0; a < 3; a++)
      for (b = 0; b < 3; b++) {
        const int ia = v[a] / ny, ja = v[a] % ny, ib = v[b] / ny, jb = v[b] % ny;
        if (ib == ia + 1 && jb == ja + 1) map[ia * (ny - 1) + ja] = '\\';
        if (ib == ia + 1 && jb == ja - 1) map[ia * (ny - 1) + jb] = '/';
      }
  }
  for (i = 0; i < nx - 1; i++) {
    for (j = 0; j < ny - 1; j++) putchar(map[i * (ny - 1) + j] ? map[i * (ny - 1) + j] : '?');
    putchar('\n');
  }
  return 0;
}
