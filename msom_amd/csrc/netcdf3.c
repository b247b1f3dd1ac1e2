/* netcdf3.c -- minimal NetCDF-3 "classic" (CDF-1 / CDF-2) writer and reader (host C).
 *
 * Reproduces the files the reference writes through libnetcdf (newqg/netcdf_bas.h:42-244,
 * qg-node/netcdf_vertex_bas.h:53-308), which is not available here: dimensions defined in the
 * order level(nl), y(ny), x(nx), time(UNLIMITED); float coordinate variables time(time),
 * y(y), x(x) with y_i = Y0 + (i + 1/2) Delta; one float variable per field with dimensions
 * (time, level, y, x); one record appended per output event with time = t.  The reference's
 * cell-centred write_nc stores level 0 only (count[1] = 1, newqg/netcdf_bas.h:184-187); like
 * the vertex version (qg-node/netcdf_vertex_bas.h:225-227) this writer stores all levels.
 * Restart: read_nc matches variables by name and reads one record (qg-node/netcdf_vertex_bas.h:350-410).
 *
 * File format: "The NetCDF Classic Format Specification" (big-endian header:
 * magic numrecs dim_list gatt_list var_list, then fixed-size data, then records).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msom_params.h"

#define NC_DIMENSION 10
#define NC_VARIABLE 11
#define NC_ATTRIBUTE 12
#define NC_FLOAT 5
#define NC_DOUBLE 6

static void put32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static uint32_t get32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put64(unsigned char *p, uint64_t v) { put32(p, (uint32_t)(v >> 32)); put32(p + 4, (uint32_t)v); }
static uint64_t get64(const unsigned char *p) { return ((uint64_t)get32(p) << 32) | get32(p + 4); }
static void putf(unsigned char *p, float f) { uint32_t u; memcpy(&u, &f, 4); put32(p, u); }
static float getf(const unsigned char *p) { uint32_t u = get32(p); float f; memcpy(&f, &u, 4); return f; }

typedef struct { unsigned char *b; size_t n, cap; } buf;
static void bput(buf *o, const void *p, size_t n) {
  if (o->n + n > o->cap) { o->cap = (o->n + n) * 2 + 256; o->b = (unsigned char *)realloc(o->b, o->cap); }
  memcpy(o->b + o->n, p, n);
  o->n += n;
}
static void b32(buf *o, uint32_t v) { unsigned char t[4]; put32(t, v); bput(o, t, 4); }
static void boff(buf *o, uint64_t v, int v2) { unsigned char t[8]; if (v2) { put64(t, v); bput(o, t, 8); } else { put32(t, (uint32_t)v); bput(o, t, 4); } }
static void bname(buf *o, const char *s) {
  size_t n = strlen(s), pad = (4 - n % 4) % 4;
  b32(o, (uint32_t)n);
  bput(o, s, n);
  bput(o, "\0\0\0", pad);
}

/* layout shared by create and append: everything follows from (nl, ny, nx, nvars, names) */
typedef struct { int v2; uint64_t header, y_begin, x_begin, rec_begin, recsize, fieldsize; } layout;

static buf make_header(int nl, int ny, int nx, int nvars, const char *const *names, uint32_t numrecs, layout *L) {
  const uint64_t fsz = (uint64_t)nl * ny * nx * 4;
  L->fieldsize = fsz;
  L->recsize = 4 + (uint64_t)nvars * fsz;
  /* CDF-1 keeps 32-bit offsets; switch to CDF-2 (64-bit offsets) for big grids / long runs */
  L->v2 = (L->recsize * 64 + (uint64_t)(ny + nx) * 4 + 4096) > 0x7fffffffull;
  buf h = {0, 0, 0};
  for (int pass = 0; pass < 2; pass++) { /* pass 0 measures the header, pass 1 writes the offsets */
    h.n = 0;
    bput(&h, L->v2 ? "CDF\002" : "CDF\001", 4);
    b32(&h, numrecs);
    b32(&h, NC_DIMENSION); b32(&h, 4);
    bname(&h, "level"); b32(&h, (uint32_t)nl);
    bname(&h, "y"); b32(&h, (uint32_t)ny);
    bname(&h, "x"); b32(&h, (uint32_t)nx);
    bname(&h, "time"); b32(&h, 0); /* UNLIMITED */
    b32(&h, 0); b32(&h, 0);        /* no global attributes */
    b32(&h, NC_VARIABLE); b32(&h, (uint32_t)(3 + nvars));
    /* time(time): record variable */
    bname(&h, "time"); b32(&h, 1); b32(&h, 3); b32(&h, 0); b32(&h, 0); b32(&h, NC_FLOAT); b32(&h, 4); boff(&h, L->rec_begin, L->v2);
    bname(&h, "y"); b32(&h, 1); b32(&h, 1); b32(&h, 0); b32(&h, 0); b32(&h, NC_FLOAT); b32(&h, (uint32_t)ny * 4); boff(&h, L->y_begin, L->v2);
    bname(&h, "x"); b32(&h, 1); b32(&h, 2); b32(&h, 0); b32(&h, 0); b32(&h, NC_FLOAT); b32(&h, (uint32_t)nx * 4); boff(&h, L->x_begin, L->v2);
    for (int v = 0; v < nvars; v++) {
      bname(&h, names[v]);
      b32(&h, 4); b32(&h, 3); b32(&h, 0); b32(&h, 1); b32(&h, 2); /* (time, level, y, x) */
      b32(&h, 0); b32(&h, 0); b32(&h, NC_FLOAT);
      b32(&h, fsz > 0xffffffffull ? 0xffffffffu : (uint32_t)fsz);
      boff(&h, L->rec_begin + 4 + (uint64_t)v * fsz, L->v2);
    }
    L->header = h.n;
    L->y_begin = L->header;
    L->x_begin = L->y_begin + (uint64_t)ny * 4;
    L->rec_begin = L->x_begin + (uint64_t)nx * 4;
  }
  return h;
}

/* create_nc, newqg/netcdf_bas.h:42-134 */
int msom_nc_create(const char *path, int nl, int ny, int nx, double L0, int nvars, const char *const *names) {
  return msom_nc_create2(path, nl, ny, nx, L0, nvars, names, 0);
}
/* vertex = 1: ny = nx = N + 1 vertex coordinates i * L0 / N (qg-node/netcdf_vertex_bas.h:145-150) */
int msom_nc_create2(const char *path, int nl, int ny, int nx, double L0, int nvars, const char *const *names, int vertex) {
  layout L;
  memset(&L, 0, sizeof L);
  buf h = make_header(nl, ny, nx, nvars, names, 0, &L);
  FILE *fp = fopen(path, "wb");
  if (!fp) { free(h.b); msom_set_error("cannot create %s", path); return -2; }
  fwrite(h.b, 1, h.n, fp);
  free(h.b);
  const double Delta = L0 * 1.0 / (vertex ? nx - 1 : nx), off = vertex ? 0. : 0.5;
  unsigned char t[4];
  for (int i = 0; i < ny; i++) { putf(t, (float)(0. + (i + off) * Delta)); fwrite(t, 1, 4, fp); }
  for (int i = 0; i < nx; i++) { putf(t, (float)(0. + (i + off) * Delta)); fwrite(t, 1, 4, fp); }
  fclose(fp);
  return 0;
}

/* write_nc, newqg/netcdf_bas.h:144-244: append one record (time + every field, all levels).
 * fields[v] is [level][y][x] fp64. */
int msom_nc_append(const char *path, int nl, int ny, int nx, int nvars, const char *const *names, double time, const double *const *fields) {
  layout L;
  memset(&L, 0, sizeof L);
  buf h = make_header(nl, ny, nx, nvars, names, 0, &L);
  FILE *fp = fopen(path, "r+b");
  if (!fp) { free(h.b); msom_set_error("file %s not found", path); return -2; }
  unsigned char *cur = (unsigned char *)malloc(h.n);
  if (fread(cur, 1, h.n, fp) != h.n || memcmp(cur, h.b, 4) || memcmp(cur + 8, h.b + 8, h.n - 8)) {
    free(cur); free(h.b); fclose(fp);
    msom_set_error("%s was not created with the same grid and variables", path);
    return -2;
  }
  const uint32_t numrecs = get32(cur + 4);
  free(cur); free(h.b);
  const size_t nf = (size_t)nl * ny * nx;
  unsigned char *rec = (unsigned char *)malloc(4 + nf * 4);
  fseeko(fp, (off_t)(L.rec_begin + (uint64_t)numrecs * L.recsize), SEEK_SET);
  putf(rec, (float)time);
  fwrite(rec, 1, 4, fp);
  for (int v = 0; v < nvars; v++) {
    for (size_t k = 0; k < nf; k++) putf(rec + 4 * k, (float)fields[v][k]);
    fwrite(rec, 1, nf * 4, fp);
  }
  free(rec);
  unsigned char t[4];
  put32(t, numrecs + 1);
  fseeko(fp, 4, SEEK_SET);
  fwrite(t, 1, 4, fp);
  fclose(fp);
  return (int)numrecs;
}

/* read_nc: variable `name`, record `rec` (-1 = last) -> out[level][y][x] fp64.  Generic header
 * parser (any classic file whose variable is float or double with dims (time, level, y, x)). */
int msom_nc_read(const char *path, const char *name, int rec, int nl, int ny, int nx, double *out, double *time_out) {
  FILE *fp = fopen(path, "rb");
  if (!fp) { msom_set_error("file %s not found", path); return -2; }
  fseeko(fp, 0, SEEK_END);
  off_t fsize = ftello(fp);
  size_t hmax = fsize < (1 << 20) ? (size_t)fsize : (1u << 20);
  unsigned char *h = (unsigned char *)malloc(hmax);
  fseeko(fp, 0, SEEK_SET);
  if (fread(h, 1, hmax, fp) != hmax || memcmp(h, "CDF", 3) || (h[3] != 1 && h[3] != 2)) {
    free(h); fclose(fp); msom_set_error("%s is not a NetCDF classic file", path); return -2;
  }
  const int v2 = h[3] == 2;
  const uint32_t numrecs = get32(h + 4);
  size_t p = 8;
#define NEED(n) if (p + (n) > hmax) goto bad
  uint32_t dimlen[64]; int ndims = 0, recdim = -1;
  NEED(8);
  uint32_t tag = get32(h + p), cnt = get32(h + p + 4); p += 8;
  if (tag == NC_DIMENSION) {
    for (uint32_t d = 0; d < cnt; d++) {  /* every entry is walked so that the header offset stays right */
      NEED(4); uint32_t n = get32(h + p); p += 4 + ((n + 3) & ~3u);
      NEED(4); const uint32_t len = get32(h + p); p += 4;
      if (d >= 64) continue;            /* dimensions beyond the table are skipped, not misparsed */
      dimlen[ndims] = len;
      if (len == 0) recdim = ndims;
      ndims++;
    }
  }
  /* skip an attribute list */
#define SKIP_ATTS()                                                                        \
  do {                                                                                     \
    NEED(8); uint32_t at = get32(h + p), an = get32(h + p + 4); p += 8;                    \
    if (at == NC_ATTRIBUTE)                                                                \
      for (uint32_t q = 0; q < an; q++) {                                                  \
        NEED(4); uint32_t n = get32(h + p); p += 4 + ((n + 3) & ~3u);                      \
        NEED(8); uint32_t ty = get32(h + p), ne = get32(h + p + 4); p += 8;                \
        static const int tsz[7] = {0, 1, 1, 2, 4, 4, 8};                                   \
        p += ((size_t)ne * tsz[ty < 7 ? ty : 0] + 3) & ~(size_t)3;                         \
      }                                                                                    \
  } while (0)
  SKIP_ATTS();
  NEED(8);
  tag = get32(h + p); cnt = get32(h + p + 4); p += 8;
  uint64_t recsize = 0, vbegin = 0, tbegin = 0;
  int found = 0, vtype = 0, have_t = 0;
  for (uint32_t v = 0; tag == NC_VARIABLE && v < cnt; v++) {
    NEED(4); uint32_t n = get32(h + p);
    NEED(4 + n);
    const int is_target = n == strlen(name) && !memcmp(h + p + 4, name, n);
    const int is_time = n == 4 && !memcmp(h + p + 4, "time", 4);
    p += 4 + ((n + 3) & ~3u);
    NEED(4); uint32_t nd = get32(h + p); p += 4;
    int isrec = 0; uint32_t dl[8] = {0};
    for (uint32_t d = 0; d < nd; d++) { NEED(4); uint32_t id = get32(h + p); p += 4; if ((int)id == recdim && d == 0) isrec = 1; if (d < 8 && id < 64) dl[d] = dimlen[id]; }
    SKIP_ATTS();
    NEED(8 + (v2 ? 8 : 4));
    uint32_t ty = get32(h + p), vsize = get32(h + p + 4); p += 8;
    uint64_t begin = v2 ? get64(h + p) : get32(h + p); p += v2 ? 8 : 4;
    if (isrec) recsize += vsize == 0xffffffffu ? (uint64_t)dl[1] * dl[2] * dl[3] * (ty == NC_DOUBLE ? 8 : 4) : ((vsize + 3) & ~3u);
    if (is_time && isrec) { tbegin = begin; have_t = 1; }
    if (is_target) {
      if (!isrec || nd != 4 || (int)dl[1] != nl || (int)dl[2] != ny || (int)dl[3] != nx || (ty != NC_FLOAT && ty != NC_DOUBLE)) {
        free(h); fclose(fp);
        msom_set_error("variable %s in %s does not have shape (time,%d,%d,%d)", name, path, nl, ny, nx);
        return -1;
      }
      found = 1; vbegin = begin; vtype = ty;
    }
  }
  if (!found) { free(h); fclose(fp); msom_set_error("variable %s not in %s", name, path); return -1; }
  if (rec < 0) rec = (int)numrecs - 1;
  if (rec < 0 || (uint32_t)rec >= numrecs) { free(h); fclose(fp); msom_set_error("record %d not in %s", rec, path); return -1; }
  {
    const size_t nf = (size_t)nl * ny * nx, es = vtype == NC_DOUBLE ? 8 : 4;
    unsigned char *raw = (unsigned char *)malloc(nf * es);
    fseeko(fp, (off_t)(vbegin + (uint64_t)rec * recsize), SEEK_SET);
    if (fread(raw, es, nf, fp) != nf) { free(raw); goto bad; }
    for (size_t k = 0; k < nf; k++) {
      if (es == 4) out[k] = getf(raw + 4 * k);
      else { uint64_t u = get64(raw + 8 * k); double d; memcpy(&d, &u, 8); out[k] = d; }
    }
    free(raw);
    if (time_out && have_t) {
      unsigned char t[4];
      fseeko(fp, (off_t)(tbegin + (uint64_t)rec * recsize), SEEK_SET);
      if (fread(t, 1, 4, fp) == 4) *time_out = getf(t);
    }
  }
  free(h); fclose(fp);
  return 0;
bad:
  free(h); fclose(fp);
  msom_set_error("truncated NetCDF file %s", path);
  return -2;
}
