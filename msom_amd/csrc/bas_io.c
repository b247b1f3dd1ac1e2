/* bas_io.c -- ".bas" gnuplot-binary matrix IO for layered fields (host C).
 *
 * Format written by output_matrixl / write_field (msqg/auxiliar_input.h:101-167) and read by
 * input_matrixl (:24-59) and by msqg/scripts/read_data.py:44-46: for every layer, an
 * (n+1) x (n+1) float32 matrix; element [0][0] = n, first row = y coordinates of the cell
 * centres, then one row per x index: x coordinate followed by f(x_i, y_j), j = 0..n-1.
 * I.e. on disk the layer is stored [x][y]; in memory the library uses [layer][y][x].
 */
#include <stdio.h>
#include <stdlib.h>

#include "msom_params.h"

int msom_bas_write(const char *path, const double *a, int nl, int n, double L0) {
  FILE *fp = fopen(path, "w");
  if (!fp) {
    msom_set_error("cannot open %s for writing", path);
    return -2;
  }
  const size_t n1 = (size_t)n + 1;
  float *frame = (float *)malloc(n1 * n1 * sizeof(float));
  if (!frame) {
    fclose(fp);
    msom_set_error("out of memory writing %s (frame %d)", path, n);
    return -2;
  }
  const float fn = (float)n, delta = (float)L0 / fn; /* float arithmetic as in the reference */
  for (int l = 0; l < nl; l++) {
    const double *al = a + (size_t)l * n * n;
    frame[0] = fn;
    for (int j = 0; j < n; j++) frame[1 + j] = delta * j + 0.f + delta / 2.f; /* Y0 = 0 */
    for (int i = 0; i < n; i++) {
      float *row = frame + (size_t)(i + 1) * n1;
      row[0] = delta * i + 0.f + delta / 2.f; /* X0 = 0 */
      for (int j = 0; j < n; j++) row[1 + j] = (float)al[(size_t)j * n + i];
    }
    if (fwrite(frame, sizeof(float), n1 * n1, fp) != n1 * n1) {
      free(frame);
      fclose(fp);
      msom_set_error("short write on %s", path);
      return -2;
    }
  }
  free(frame);
  fclose(fp);
  return 0;
}

/* The file resolution is taken from the file itself and sampled onto the model grid at the
 * cell that contains each model cell centre (msqg/auxiliar_input.h:44-56); points outside
 * the file's extent get 0. */
int msom_bas_read(const char *path, double *a, int nl, int n, double L0) {
  FILE *fp = fopen(path, "r");
  if (!fp) {
    msom_set_error("file %s not found", path);
    return -2;
  }
  const double delta = L0 / n;
  for (int l = 0; l < nl; l++) {
    float width;
    if (fread(&width, sizeof(float), 1, fp) != 1) goto shortread;
    const int m = (int)width;
    if (m <= 0 || m > (1 << 16) || (float)m != width) goto shortread; /* the frame size comes from the file: bound it */
    const size_t m1 = (size_t)m + 1;
    float *frame = (float *)malloc(m1 * m1 * sizeof(float));
    if (!frame) {
      fclose(fp);
      msom_set_error("out of memory reading %s (frame %d)", path, m);
      return -2;
    }
    /* rest of the first row, then m full rows */
    if (fread(frame + 1, sizeof(float), m1 * m1 - 1, fp) != m1 * m1 - 1) {
      free(frame);
      goto shortread;
    }
    double *al = a + (size_t)l * n * n;
    for (int jj = 0; jj < n; jj++)
      for (int ii = 0; ii < n; ii++) {
        const double x = (ii + 0.5) * delta, y = (jj + 0.5) * delta;
        const int i = (int)((x - 0.) * width / L0), j = (int)((y - 0.) * width / L0);
        al[(size_t)jj * n + ii] =
            (i >= 0 && i < m && j >= 0 && j < m) ? (double)frame[(size_t)(i + 1) * m1 + (j + 1)] : 0.;
      }
    free(frame);
  }
  fclose(fp);
  return 0;
shortread:
  fclose(fp);
  msom_set_error("short read on %s", path);
  return -2;
}
