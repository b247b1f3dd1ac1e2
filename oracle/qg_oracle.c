/*
 * qg_oracle.c -- CPU restatement ("oracle") of the msom multi-layer QG hot path
 * (msqg/qg.h, msqg/poisson_layer.h, msqg/layer.h, msqg/qg.c; MG driver text from
 * mspg/elliptic.h; dt limiter text from newqg/qg.h).
 *
 * TEST INFRASTRUCTURE ONLY -- see qg_oracle.h.  PARITY UNPINNED (no reference golden
 * vectors exist and the Basilisk-C reference cannot be built here); pinned by analytic
 * known-answer tests only.
 *
 * Conventions: cell (i,j), i = x index, j = y index, centre ((i+1/2)D, (j+1/2)D);
 * layer l = 0 is the top layer.  API arrays are numpy C-order [layer][y][x], interior
 * points only (msqg/qg.h:1177-1188).  Internal storage keeps one ghost ring.
 * Citations are file:line relative to /root/reference.
 */
#include "qg_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum { BC_DIRICHLET0 = 0, BC_NEUMANN = 1, BC_PERIODIC = 2, BC_DIRICHLET_LIN = 3 };

typedef struct fld_s {
  int nx, ny, nl, bc;
  double *d;
  /* BC_DIRICHLET_LIN: value vpg[l]*x - upg[l]*y imposed on the wall faces (msqg/qg.h:1105-1114) */
  const double *lin_u, *lin_v;
  double lin_D;
} fld;

#define IDX(f, l, i, j) ((((size_t)(l) * ((f)->ny + 2)) + (size_t)((j) + 1)) * ((f)->nx + 2) + (size_t)((i) + 1))
#define V(f, l, i, j) ((f)->d[IDX(f, l, i, j)])

struct orc {
  /* parameters, msqg/qg.h:63-106 */
  int nx, ny, nl;
  double L0, Rom, Ekb, Eks, tau0, Re, Re4, iRe, iRe4, sbc, beta, DT, CFL, tend, dtout;
  int varRo, flsrv, flag_topo, nptr, ediag, nme_ft;
  double afilt, Lfmax, dtflt;   /* wavelet filter, msqg/qg.h:53,83,85 */
  int wnlev;                    /* wavelet pyramid: level 0 = finest ... wnlev-1 = 1 cell on the short side */
  struct fld_s *ws, *ww, *wsig, *wsf; /* s, w, sig_lev, sig_filt on every level */
  double ptr_r[ORC_MAXNL], ptr_ir[ORC_MAXNL], Pe[ORC_MAXNL], iPe[ORC_MAXNL];
  double Frm[ORC_MAXNL], dhu[ORC_MAXNL], upg[ORC_MAXNL], vpg[ORC_MAXNL];
  double dhf[ORC_MAXNL], dhc[ORC_MAXNL], idh0[ORC_MAXNL], idh1[ORC_MAXNL];
  /* stochastic variant msqg/qg_stochastic.h */
  int stochastic, corrector_step;
  double tr_stoch, itr_stoch, amp_stoch;
  /* solver controls, mspg/elliptic.h:111-112, msqg/qg.h:159 */
  double tolerance;
  int nitermax, nitermin, smoother, quiet, mglevels;
  /* fields */
  fld f[ORC_NFIELDS];
  /* multigrid hierarchy: level 0 = finest */
  int nlev;
  fld *da, *res, *S;
  double *delta;
  /* time loop state */
  double t, dt, tnext_event, previous;
  int iter;
  orc_mgstats mg;
};

/* ------------------------------------------------------------------ fields */

static void fld_alloc(fld *f, int nx, int ny, int nl, int bc) {
  f->nx = nx; f->ny = ny; f->nl = nl; f->bc = bc;
  f->d = (double *)calloc((size_t)nl * (nx + 2) * (ny + 2), sizeof(double));
}
static void fld_free(fld *f) { free(f->d); f->d = NULL; }
static void fld_zero(fld *f) { memset(f->d, 0, (size_t)f->nl * (f->nx + 2) * (f->ny + 2) * sizeof(double)); }

/* [BASILISK RULE] boundary(): box BCs applied direction by direction (right, left, then
 * top, bottom); the loop along the wall for the y-direction spans the x-ghosts, so corner
 * ghosts are the y-BC applied to the x-ghost column.  dirichlet(0): ghost = -interior (wall
 * on the cell face, msqg/layer.h:13-21); default: ghost = interior; periodic: wrap. */
static void boundary(fld *f) {
  const int nx = f->nx, ny = f->ny;
  if (f->bc == BC_DIRICHLET_LIN) {
    /* [BASILISK RULE] dirichlet(expr): ghost = 2*expr - interior, expr evaluated at the
     * centre of the boundary face; same direction order / corner rule as below */
    const double D = f->lin_D, Lx = nx * D, Ly = ny * D;
    for (int l = 0; l < f->nl; l++) {
      const double u = f->lin_u[l], v = f->lin_v[l];
      for (int j = 0; j < ny; j++) {
        const double y = (j + 0.5) * D;
        V(f, l, nx, j) = 2. * (v * Lx - u * y) - V(f, l, nx - 1, j);
        V(f, l, -1, j) = 2. * (v * 0. - u * y) - V(f, l, 0, j);
      }
      for (int i = -1; i <= nx; i++) {
        const double x = (i + 0.5) * D;
        V(f, l, i, ny) = 2. * (v * x - u * Ly) - V(f, l, i, ny - 1);
        V(f, l, i, -1) = 2. * (v * x - u * 0.) - V(f, l, i, 0);
      }
    }
    return;
  }
  for (int l = 0; l < f->nl; l++) {
    for (int j = 0; j < ny; j++) {
      if (f->bc == BC_PERIODIC) {
        V(f, l, nx, j) = V(f, l, 0, j);
        V(f, l, -1, j) = V(f, l, nx - 1, j);
      } else {
        double s = f->bc == BC_DIRICHLET0 ? -1. : 1.;
        V(f, l, nx, j) = s * V(f, l, nx - 1, j);
        V(f, l, -1, j) = s * V(f, l, 0, j);
      }
    }
    for (int i = -1; i <= nx; i++) {
      if (f->bc == BC_PERIODIC) {
        V(f, l, i, ny) = V(f, l, i, 0);
        V(f, l, i, -1) = V(f, l, i, ny - 1);
      } else {
        double s = f->bc == BC_DIRICHLET0 ? -1. : 1.;
        V(f, l, i, ny) = s * V(f, l, i, ny - 1);
        V(f, l, i, -1) = s * V(f, l, i, 0);
      }
    }
  }
}

static void fld_from_array(fld *f, const double *a) {
  for (int l = 0; l < f->nl; l++)
    for (int j = 0; j < f->ny; j++)
      for (int i = 0; i < f->nx; i++)
        V(f, l, i, j) = a[((size_t)l * f->ny + j) * f->nx + i];
  boundary(f);
}
static void fld_to_array(const fld *f, double *a) {
  for (int l = 0; l < f->nl; l++)
    for (int j = 0; j < f->ny; j++)
      for (int i = 0; i < f->nx; i++)
        a[((size_t)l * f->ny + j) * f->nx + i] = V(f, l, i, j);
}

/* ------------------------------------------------------------------ params */

/* msqg/qg.h:668-675 */
static void trim_whitespace(char *s) {
  const char *d = s;
  do {
    while (*d == ' ') ++d;
  } while ((*s++ = *d++));
}
/* msqg/qg.h:678-687 */
static void str2array(char *s, double *array) {
  int n = 0;
  char *p = strtok(s, "[,]");
  while (p != NULL && n < ORC_MAXNL) {
    array[n++] = atof(p);
    p = strtok(NULL, ",");
  }
}

static void parse_line(orc_t *o, char *buf, int *N, int *Ny) {
  trim_whitespace(buf);
  char *eq = strchr(buf, '=');
  if (!eq) return;
  *eq = '\0';
  char *k = buf, *v = eq + 1;
  char *e2 = strchr(v, '=');  /* strtok(NULL,"=") stops at a second '=' */
  if (e2) *e2 = '\0';
  size_t n = strlen(v);
  while (n && (v[n - 1] == '\n' || v[n - 1] == '\r')) v[--n] = '\0';
  if (!*v) return;
  /* msqg/qg.h:698-731 */
  if (!strcmp(k, "N")) *N = atoi(v);
  else if (!strcmp(k, "Ny")) *Ny = atoi(v);          /* extension: non-square domain */
  else if (!strcmp(k, "nl")) o->nl = atoi(v);
  else if (!strcmp(k, "varRo")) o->varRo = atoi(v);
  else if (!strcmp(k, "flsrv")) o->flsrv = atoi(v);
  else if (!strcmp(k, "nptr")) o->nptr = atoi(v);
  else if (!strcmp(k, "ptr_r")) str2array(v, o->ptr_r);
  else if (!strcmp(k, "Pe")) str2array(v, o->Pe);
  else if (!strcmp(k, "L0")) o->L0 = atof(v);
  else if (!strcmp(k, "Rom")) o->Rom = atof(v);
  else if (!strcmp(k, "Ekb")) o->Ekb = atof(v);
  else if (!strcmp(k, "Eks")) o->Eks = atof(v);
  else if (!strcmp(k, "tau0")) o->tau0 = atof(v);
  else if (!strcmp(k, "Re")) o->Re = atof(v);
  else if (!strcmp(k, "Re4")) o->Re4 = atof(v);
  else if (!strcmp(k, "sbc")) o->sbc = atof(v);
  else if (!strcmp(k, "beta")) o->beta = atof(v);
  else if (!strcmp(k, "DT")) o->DT = atof(v);
  else if (!strcmp(k, "tend")) o->tend = atof(v);
  else if (!strcmp(k, "dtout")) o->dtout = atof(v);
  else if (!strcmp(k, "CFL")) o->CFL = atof(v);
  else if (!strcmp(k, "Fr")) str2array(v, o->Frm);
  else if (!strcmp(k, "dh")) str2array(v, o->dhu);
  else if (!strcmp(k, "upg")) str2array(v, o->upg);
  else if (!strcmp(k, "vpg")) str2array(v, o->vpg);
  else if (!strcmp(k, "tr_stoch")) o->tr_stoch = atof(v);
  else if (!strcmp(k, "amp_stoch")) o->amp_stoch = atof(v);
  else if (!strcmp(k, "ediag")) o->ediag = atoi(v);
  else if (!strcmp(k, "afilt")) o->afilt = atof(v);
  else if (!strcmp(k, "Lfmax")) o->Lfmax = atof(v);
  else if (!strcmp(k, "dtflt")) o->dtflt = atof(v);
  else if (!strcmp(k, "MGLEVELS")) o->mglevels = atoi(v); /* extension: cap on MG levels */
}

/* wavelet pyramid: Basilisk levels depth() ... 0 (1 x 1 cell) */
static void build_wavelet_levels(orc_t *o) {
  int n = 1, nx = o->nx, ny = o->ny;
  while ((nx >> n) >= 1 && (ny >> n) >= 1 && ((nx >> n) << n) == nx && ((ny >> n) << n) == ny) n++;
  o->wnlev = n;
  o->ws = (fld *)calloc(n, sizeof(fld)); o->ww = (fld *)calloc(n, sizeof(fld));
  o->wsig = (fld *)calloc(n, sizeof(fld)); o->wsf = (fld *)calloc(n, sizeof(fld));
  int bc = o->sbc == -1 ? BC_PERIODIC : BC_DIRICHLET0, bcn = o->sbc == -1 ? BC_PERIODIC : BC_NEUMANN;
  for (int k = 0; k < n; k++) {
    fld_alloc(&o->ws[k], nx >> k, ny >> k, o->nl, bc); fld_alloc(&o->ww[k], nx >> k, ny >> k, o->nl, bcn);
    fld_alloc(&o->wsig[k], nx >> k, ny >> k, 1, bcn); fld_alloc(&o->wsf[k], nx >> k, ny >> k, 1, bcn);
  }
}

static void build_levels(orc_t *o) {
  int n = 0, nx = o->nx, ny = o->ny;
  /* minlevel = 1 (msqg/poisson_layer.h:296-297): coarsest grid has 2 cells on its short side */
  while ((nx >> n) >= 2 && (ny >> n) >= 2 && ((nx >> n) << n) == nx && ((ny >> n) << n) == ny) n++;
  if (o->mglevels > 0 && o->mglevels < n) n = o->mglevels;
  o->nlev = n;
  o->da = (fld *)calloc(n, sizeof(fld));
  o->res = (fld *)calloc(n, sizeof(fld));
  o->S = (fld *)calloc(n, sizeof(fld));
  o->delta = (double *)calloc(n, sizeof(double));
  int bc = o->sbc == -1 ? BC_PERIODIC : BC_DIRICHLET0;
  int bcn = o->sbc == -1 ? BC_PERIODIC : BC_NEUMANN;
  for (int k = 0; k < n; k++) {
    fld_alloc(&o->da[k], nx >> k, ny >> k, o->nl, bc); /* homogeneous version of a's BC */
    fld_alloc(&o->res[k], nx >> k, ny >> k, o->nl, bc);
    fld_alloc(&o->S[k], nx >> k, ny >> k, o->nl > 1 ? o->nl - 1 : 1, bcn);
    o->delta[k] = o->L0 / (double)(nx >> k);
  }
}

orc_t *orc_create_str(const char *text) {
  orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
  int N = 64, Ny = 0;
  /* defaults msqg/qg.h:63-106 and Basilisk globals */
  o->nl = 1; o->L0 = 1.; o->beta = 0.5; o->DT = 1e10; o->CFL = 0.5; o->tend = 1; o->dtout = 1;
  o->amp_stoch = 1; o->afilt = 10.; o->Lfmax = 1e10; o->dtflt = -1; o->ediag = -1;
  o->tolerance = 1e-3; o->nitermax = 100; o->nitermin = 1; o->smoother = ORC_GS_LEX;
  char *copy = strdup(text), *save = NULL;
  for (char *line = strtok_r(copy, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    char buf[300];
    strncpy(buf, line, 299); buf[299] = '\0';
    parse_line(o, buf, &N, &Ny);
  }
  free(copy);
  o->nx = N; o->ny = Ny > 0 ? Ny : N;
  /* msqg/qg.h:739-746 */
  o->iRe = o->Re == 0 ? 0. : 1 / o->Re;
  o->iRe4 = o->Re4 == 0 ? 0. : -1 / o->Re4;
  double D = o->L0 / o->nx;
  if (o->Re != 0) o->DT = 0.5 * fmin(o->DT, D * D * o->Re / 4.);
  if (o->Re4 != 0) o->DT = 0.5 * fmin(o->DT, (D * D) * (D * D) * o->Re4 / 32.);
  if (o->tr_stoch != 0) o->itr_stoch = 1 / o->tr_stoch;   /* qg.h:757 */
  for (int nt = 0; nt < o->nptr && nt < ORC_MAXNL; nt++) { /* qg.h:751-754 */
    o->ptr_ir[nt] = o->ptr_r[nt] == 0 ? 0. : 1 / o->ptr_r[nt];
    o->iPe[nt] = o->Pe[nt] == 0 ? 0. : 1 / o->Pe[nt];
  }
  if (o->nl < 1 || o->nl > ORC_MAXNL) { free(o); return NULL; }

  /* set_vars msqg/qg.h:837-925 */
  int bc = o->sbc == -1 ? BC_PERIODIC : BC_DIRICHLET0;
  int bcn = o->sbc == -1 ? BC_PERIODIC : BC_NEUMANN;
  int nl = o->nl, nlm = nl > 1 ? nl - 1 : 1;
  for (int k = 0; k < ORC_NFIELDS; k++) {
    int layers = nl, b = bc;
    if (k == ORC_FR || k == ORC_S) { layers = nlm; b = bcn; }
    if (k == ORC_RO || k == ORC_TOPO) { layers = 1; b = bcn; }
    if (k >= ORC_PTR && k <= ORC_PTR_PRED) { layers = nl * (o->nptr > 0 ? o->nptr : 1); b = bcn; } /* qg.h:867-870 */
    if (k == ORC_RD) { layers = 1; b = bcn; }
    fld_alloc(&o->f[k], o->nx, o->ny, layers, b);
  }
  if (o->sbc == -1) { /* msqg/qg.h:1105-1114: the large-scale stream function is not periodic */
    fld *pg = &o->f[ORC_PSIPG];
    pg->bc = BC_DIRICHLET_LIN; pg->lin_u = o->upg; pg->lin_v = o->vpg; pg->lin_D = D;
  }
  for (int l = 0; l < nl; l++) o->dhf[l] = o->dhu[l];
  fld *Fr = &o->f[ORC_FR], *pp = &o->f[ORC_PSIPG], *Ro = &o->f[ORC_RO];
  for (int l = 0; l < nl - 1; l++)
    for (int j = 0; j < o->ny; j++)
      for (int i = 0; i < o->nx; i++) V(Fr, l, i, j) = o->Frm[l];
  for (int l = 0; l < nl; l++)
    for (int j = 0; j < o->ny; j++)
      for (int i = 0; i < o->nx; i++) {
        double x = (i + 0.5) * D, y = (j + 0.5) * D;
        V(pp, l, i, j) = o->vpg[l] * x - o->upg[l] * y;   /* qg.h:907 */
      }
  boundary(pp);
  for (int j = 0; j < o->ny; j++)
    for (int i = 0; i < o->nx; i++) { V(Ro, 0, i, j) = o->Rom; V(&o->f[ORC_RD], 0, i, j) = 1.; /* qg.h:913 */ }
  build_levels(o);
  build_wavelet_levels(o);
  o->t = 0; o->iter = 0; o->dt = 1.; o->tnext_event = HUGE_VAL; o->previous = 0.;
  return o;
}

void orc_destroy(orc_t *o) {
  if (!o) return;
  for (int k = 0; k < ORC_NFIELDS; k++) fld_free(&o->f[k]);
  for (int k = 0; k < o->nlev; k++) { fld_free(&o->da[k]); fld_free(&o->res[k]); fld_free(&o->S[k]); }
  for (int k = 0; k < o->wnlev; k++) { fld_free(&o->ws[k]); fld_free(&o->ww[k]); fld_free(&o->wsig[k]); fld_free(&o->wsf[k]); }
  free(o->ws); free(o->ww); free(o->wsig); free(o->wsf);
  free(o->da); free(o->res); free(o->S); free(o->delta);
  free(o);
}

int orc_set_option(orc_t *o, const char *key, double v) {
  if (!strcmp(key, "smoother")) o->smoother = (int)v;
  else if (!strcmp(key, "TOLERANCE")) o->tolerance = v;
  else if (!strcmp(key, "NITERMAX")) o->nitermax = (int)v;
  else if (!strcmp(key, "NITERMIN")) o->nitermin = (int)v;
  else if (!strcmp(key, "stochastic")) o->stochastic = (int)v;
  else if (!strcmp(key, "quiet")) o->quiet = (int)v;
  else if (!strcmp(key, "flag_topo")) o->flag_topo = (int)v;
  else if (!strcmp(key, "DT")) o->DT = v;
  else return -1;
  return 0;
}

double orc_get_param(orc_t *o, const char *key) {
  if (!strcmp(key, "N") || !strcmp(key, "nx")) return o->nx;
  if (!strcmp(key, "ny")) return o->ny;
  if (!strcmp(key, "nl")) return o->nl;
  if (!strcmp(key, "nptr")) return o->nptr;
  if (!strcmp(key, "L0")) return o->L0;
  if (!strcmp(key, "DT")) return o->DT;
  if (!strcmp(key, "iRe")) return o->iRe;
  if (!strcmp(key, "iRe4")) return o->iRe4;
  if (!strcmp(key, "CFL")) return o->CFL;
  if (!strcmp(key, "Rom")) return o->Rom;
  if (!strcmp(key, "tend")) return o->tend;
  if (!strcmp(key, "dtout")) return o->dtout;
  if (!strcmp(key, "beta")) return o->beta;
  if (!strcmp(key, "tau0")) return o->tau0;
  if (!strcmp(key, "Ekb")) return o->Ekb;
  if (!strcmp(key, "sbc")) return o->sbc;
  if (!strncmp(key, "idh0_", 5)) return o->idh0[atoi(key + 5)];
  if (!strncmp(key, "idh1_", 5)) return o->idh1[atoi(key + 5)];
  if (!strncmp(key, "Fr_", 3)) return o->Frm[atoi(key + 3)];
  if (!strncmp(key, "dh_", 3)) return o->dhf[atoi(key + 3)];
  return NAN;
}

int orc_nlayers_of(orc_t *o, int field) { return o->f[field].nl; }
void orc_set_field(orc_t *o, int field, const double *a) { fld_from_array(&o->f[field], a); }
void orc_get_field(orc_t *o, int field, double *a) { fld_to_array(&o->f[field], a); }

/* msqg/qg.c:65-70; [BASILISK RULE] statsf: sum = sum f D^2, volume = sum D^2 */
void orc_remove_mean(orc_t *o, int field) {
  fld *f = &o->f[field];
  double D2 = (o->L0 / o->nx) * (o->L0 / o->nx);
  for (int l = 0; l < f->nl; l++) {
    double sum = 0, vol = 0;
    for (int i = 0; i < f->nx; i++)
      for (int j = 0; j < f->ny; j++) { sum += V(f, l, i, j) * D2; vol += D2; }
    double m = sum / vol;
    for (int i = 0; i < f->nx; i++)
      for (int j = 0; j < f->ny; j++) V(f, l, i, j) -= m;
  }
  boundary(f);
}

/* ------------------------------------------------------------------ operators */

#define LAP(p, l, i, j, D2) ((V(p, l, (i) + 1, j) + V(p, l, (i) - 1, j) + V(p, l, i, (j) + 1) + V(p, l, i, (j) - 1) - 4 * V(p, l, i, j)) / (D2))

/* msqg/qg.h:185-198: partial-slip override of the zeta ghosts */
static void slip_bc(orc_t *o, fld *po, fld *zeta) {
  if (!(o->sbc > 0)) return;
  double D = o->L0 / o->nx, c = o->sbc / ((0.5 * o->sbc + 1) * D * D);
  int nx = po->nx, ny = po->ny;
  for (int l = 0; l < po->nl; l++) {
    for (int j = 0; j < ny; j++) {
      V(zeta, l, -1, j) = c * (V(po, l, 0, j) - V(po, l, -1, j));
      V(zeta, l, nx, j) = c * (V(po, l, nx - 1, j) - V(po, l, nx, j));
    }
    for (int i = 0; i < nx; i++) {
      V(zeta, l, i, ny) = c * (V(po, l, i, ny - 1) - V(po, l, i, ny));
      V(zeta, l, i, -1) = c * (V(po, l, i, 0) - V(po, l, i, -1));
    }
  }
}

/* msqg/qg.h:172-200 */
static void comp_del2(orc_t *o, fld *po, fld *zeta, double add, double fac) {
  double D = o->L0 / o->nx, D2 = D * D;
#pragma omp parallel for collapse(2)
  for (int l = 0; l < po->nl; l++)
    for (int j = 0; j < po->ny; j++)
      for (int i = 0; i < po->nx; i++)
        V(zeta, l, i, j) = add * V(zeta, l, i, j) + fac * LAP(po, l, i, j, D2);
  boundary(zeta);
  slip_bc(o, po, zeta);
}
void orc_comp_del2(orc_t *o, int in, int out, double add, double fac) { comp_del2(o, &o->f[in], &o->f[out], add, fac); }

/* msqg/qg.h:203-246 */
static void comp_stretch(orc_t *o, fld *po, fld *st, double add, double fac) {
  const int nl = o->nl;
  fld *S = &o->f[ORC_S];
  const double *idh0 = o->idh0, *idh1 = o->idh1;
#pragma omp parallel for
  for (int j = 0; j < po->ny; j++)
    for (int i = 0; i < po->nx; i++) {
      if (nl > 1) {
        int l = 0;
        V(st, l, i, j) = add * V(st, l, i, j) + fac * V(S, l, i, j) * (V(po, l + 1, i, j) - V(po, l, i, j)) * idh1[l];
        for (l = 1; l < nl - 1; l++)
          V(st, l, i, j) = add * V(st, l, i, j) +
                           fac * (V(S, l - 1, i, j) * (V(po, l - 1, i, j) - V(po, l, i, j)) * idh0[l] +
                                  V(S, l, i, j) * (V(po, l + 1, i, j) - V(po, l, i, j)) * idh1[l]);
        l = nl - 1;
        V(st, l, i, j) = add * V(st, l, i, j) + fac * V(S, l - 1, i, j) * (V(po, l - 1, i, j) - V(po, l, i, j)) * idh0[l];
      } else
        /* reference: stretch[] = 0 (qg.h:239-242), which makes nl == 1 degenerate.
         * Build-defined: Gamma = 0 contributes nothing, so nl == 1 is a barotropic model. */
        V(st, 0, i, j) = add * V(st, 0, i, j);
    }
  boundary(st);
}
void orc_comp_stretch(orc_t *o, int in, int out, double add, double fac) { comp_stretch(o, &o->f[in], &o->f[out], add, fac); }

/* msqg/qg.h:252-262: this is -J(p,q) */
static inline double jacobian(const fld *po, int lp, const fld *qo, int lq, int i, int j, double D) {
#define P(a, b) V(po, lp, i + (a), j + (b))
#define Q(a, b) V(qo, lq, i + (a), j + (b))
  return (((Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0)) +
           Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1)) - Q(0, 1) * (P(1, 1) - P(-1, 1)) +
           Q(0, -1) * (P(1, -1) - P(-1, -1)) + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1)) -
           P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1))) /
          (12. * D * D));
#undef P
#undef Q
}
/* msqg/qg.h:269 */
#define BETA_EFFECT(po, l, i, j) (o->beta * (V(po, l, (i) - 1, j) - V(po, l, (i) + 1, j)) / (2 * D))

/* [BASILISK RULE] timestep() (timestep.h; algorithm text in-tree at newqg/qg.h:202-219) */
double orc_timestep_limiter(orc_t *o, double dtmin_faces, double dtmax) {
  dtmax /= o->CFL;
  if (dtmin_faces < dtmax) dtmax = dtmin_faces;
  dtmax *= o->CFL;
  if (dtmax > o->previous) dtmax = (o->previous + 0.1 * dtmax) / 1.1;
  o->previous = dtmax;
  return dtmax;
}
void orc_reset_limiter(orc_t *o) { o->previous = 0.; }

/* msqg/qg.h:276-283 (comp_vel) + min over faces of Delta/|u| */
static double min_dt_faces(orc_t *o, fld *po, int l) {
  double D = o->L0 / o->nx, dtmin = HUGE_VAL;
  int nx = po->nx, ny = po->ny;
#pragma omp parallel for reduction(min : dtmin)
  for (int j = 0; j <= ny; j++)
    for (int i = 0; i <= nx; i++) {
      if (j < ny) { /* x-face between (i-1,j) and (i,j) */
        double u = -1. * 0.25 * (V(po, l, i, j + 1) - V(po, l, i, j - 1) + V(po, l, i - 1, j + 1) - V(po, l, i - 1, j - 1)) / D;
        if (u != 0.) { double dt = D / fabs(u); if (dt < dtmin) dtmin = dt; }
      }
      if (i < nx) { /* y-face between (i,j-1) and (i,j) */
        double v = 1. * 0.25 * (V(po, l, i + 1, j) - V(po, l, i - 1, j) + V(po, l, i + 1, j - 1) - V(po, l, i - 1, j - 1)) / D;
        if (v != 0.) { double dt = D / fabs(v); if (dt < dtmin) dtmin = dt; }
      }
    }
  return dtmin;
}

/* msqg/qg.h:288-393 (and msqg/qg_stochastic.h:17-111 when o->stochastic) */
static double advection_pv(orc_t *o, fld *qo, fld *qot, fld *po, fld *dqo, double dtmax) {
  const int nl = o->nl;
  const double D = o->L0 / o->nx;
  fld *pp = &o->f[ORC_PSIPG], *qp = &o->f[ORC_ZETAPG], *S = &o->f[ORC_S];
  const double *idh0 = o->idh0, *idh1 = o->idh1;
  const int st = o->stochastic;
#pragma omp parallel for
  for (int j = 0; j < po->ny; j++)
    for (int i = 0; i < po->nx; i++) {
      double ju, jd = 0;
      if (nl > 1) {
        int l = 0;
        if (!st) {
          jd = jacobian(po, l, po, l + 1, i, j, D) + jacobian(pp, l, po, l + 1, i, j, D) + jacobian(po, l, pp, l + 1, i, j, D);
          V(dqo, l, i, j) += jacobian(po, l, qo, l, i, j, D) + jacobian(pp, l, qo, l, i, j, D) + BETA_EFFECT(po, l, i, j) +
                             V(S, l, i, j) * jd * idh1[l];
        } else { /* qg_stochastic.h:39-40: top layer has no J(po,qo) */
          jd = jacobian(pp, l, po, l + 1, i, j, D) + jacobian(po, l, pp, l + 1, i, j, D);
          V(dqo, l, i, j) += jacobian(pp, l, qo, l, i, j, D) + BETA_EFFECT(po, l, i, j) + V(S, l, i, j) * jd * idh1[l];
        }
        V(dqo, l, i, j) += jacobian(po, l, qp, l, i, j, D);
        if (st) V(dqo, l, i, j) += -V(qot, l, i, j) * o->itr_stoch;
        for (l = 1; l < nl - 1; l++) {
          ju = -jd;
          if (!st)
            jd = jacobian(po, l, po, l + 1, i, j, D) + jacobian(pp, l, po, l + 1, i, j, D) + jacobian(po, l, pp, l + 1, i, j, D);
          else
            jd = jacobian(pp, l, po, l + 1, i, j, D) + jacobian(po, l, pp, l + 1, i, j, D);
          V(dqo, l, i, j) += jacobian(po, l, qo, l, i, j, D) + jacobian(pp, l, qo, l, i, j, D) + BETA_EFFECT(po, l, i, j) +
                             V(S, l - 1, i, j) * ju * idh0[l] + V(S, l, i, j) * jd * idh1[l];
          V(dqo, l, i, j) += jacobian(po, l, qp, l, i, j, D);
          if (st) V(dqo, l, i, j) += -V(qot, l, i, j) * o->itr_stoch;
        }
        l = nl - 1;
        ju = -jd;
        V(dqo, l, i, j) += jacobian(po, l, qo, l, i, j, D) + jacobian(pp, l, qo, l, i, j, D) + BETA_EFFECT(po, l, i, j) +
                           V(S, l - 1, i, j) * ju * idh0[l];
        V(dqo, l, i, j) += jacobian(po, l, qp, l, i, j, D);
        if (st) V(dqo, l, i, j) += -V(qot, l, i, j) * o->itr_stoch;
      } else {
        /* reference: dqo[] = 0 (qg.h:376-379), degenerate; build-defined barotropic tendency
         * = the nl > 1 formula without the cross-layer terms (cf. newqg/qg.h:197-200) */
        V(dqo, 0, i, j) += jacobian(po, 0, qo, 0, i, j, D) + jacobian(pp, 0, qo, 0, i, j, D) + BETA_EFFECT(po, 0, i, j);
        V(dqo, 0, i, j) += jacobian(po, 0, qp, 0, i, j, D);
      }
    }
  /* qg.h:383-391: 2*nl sequential limiter calls sharing one static `previous` */
  for (int l = 0; l < nl; l++) {
    dtmax = orc_timestep_limiter(o, min_dt_faces(o, po, l), dtmax);
    dtmax = orc_timestep_limiter(o, min_dt_faces(o, pp, l), dtmax);
  }
  return dtmax;
}
double orc_advection_pv(orc_t *o, int zeta, int q, int psi, int dq, double dtmax) {
  return advection_pv(o, &o->f[zeta], &o->f[q], &o->f[psi], &o->f[dq], dtmax);
}

/* msqg/qg.h:397-403 */
static void comp_q(orc_t *o, fld *po, fld *qo) {
  comp_del2(o, po, qo, 0., 1.);
  comp_stretch(o, po, qo, 1., 1.);
  boundary(qo);
}
void orc_comp_q(orc_t *o, int psi, int q) { comp_q(o, &o->f[psi], &o->f[q]); }

/* msqg/qg.h:407-422 */
static void dissip(orc_t *o, fld *zeta, fld *dqo) {
  fld *tmp = &o->f[ORC_TMP];
  comp_stretch(o, zeta, dqo, 1., o->iRe);
  comp_del2(o, zeta, tmp, 0., 1.);
#pragma omp parallel for collapse(2)
  for (int l = 0; l < o->nl; l++)
    for (int j = 0; j < zeta->ny; j++)
      for (int i = 0; i < zeta->nx; i++) V(dqo, l, i, j) += V(tmp, l, i, j) * o->iRe;
  comp_stretch(o, tmp, dqo, 1., o->iRe4);
  comp_del2(o, tmp, dqo, 1., o->iRe4);
}
void orc_dissip(orc_t *o, int zeta, int dq) { dissip(o, &o->f[zeta], &o->f[dq]); }

/* msqg/qg.h:429-440 ekman_friction, :447-459 surface_forcing, :466-474 qforcing,
 * :481-488 bottom_topography, call order :626-630 */
static void forcing_terms(orc_t *o, fld *zeta, fld *po, fld *dqo, int with_qforcing) {
  const int nl = o->nl, nx = zeta->nx, ny = zeta->ny;
  const double D = o->L0 / o->nx;
  fld *qf = &o->f[ORC_QFORC], *topo = &o->f[ORC_TOPO], *Ro = &o->f[ORC_RO];
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      V(dqo, 0, i, j) -= o->Eks / (o->Rom * 2 * o->dhf[0]) * V(zeta, 0, i, j);
      V(dqo, nl - 1, i, j) -= o->Ekb / (o->Rom * 2 * o->dhf[nl - 1]) * V(zeta, nl - 1, i, j);
    }
  for (int j = 0; j < ny; j++) {
    double y = (j + 0.5) * D;
    for (int i = 0; i < nx; i++)
      V(dqo, 0, i, j) -= o->tau0 / (o->Rom * o->dhf[0]) * sin(2 * M_PI * y / o->L0) * sin(M_PI * y / o->L0);
  }
  if (with_qforcing)
    for (int l = 0; l < nl; l++)
      for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) V(dqo, l, i, j) += V(qf, l, i, j);
  if (o->flag_topo)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        V(dqo, nl - 1, i, j) += jacobian(po, nl - 1, topo, 0, i, j, D) / (V(Ro, 0, i, j) * o->dhf[nl - 1]);
}
void orc_forcing_terms(orc_t *o, int zeta, int psi, int dq) { forcing_terms(o, &o->f[zeta], &o->f[psi], &o->f[dq], 1); }

/* ------------------------------------------------------------------ elliptic solver */

/* one column of relax_layer, msqg/poisson_layer.h:75-149 (nl == 1: the reference body is
 * empty; the build defines it as the plain Poisson relaxation of Basilisk poisson.h,
 * cf. newqg/qg.h:148-156 with lambda = 0) */
static inline void relax_column(const orc_t *o, fld *al, const fld *bl, const fld *S, double D, int i, int j) {
  const int nl = al->nl;
  double t0[ORC_MAXNL], t1[ORC_MAXNL], t2[ORC_MAXNL], rhs[ORC_MAXNL];
  const double sqD = D * D;
  if (nl > 1) {
    int ll = 0;
    rhs[ll] = -sqD * V(bl, ll, i, j);
    t2[ll] = -sqD * V(S, ll, i, j) * o->idh1[ll];
    t1[ll] = -t2[ll];
    rhs[ll] += 1. * V(al, ll, i + 1, j) + 1. * V(al, ll, i - 1, j);
    t1[ll] += 1. + 1.;
    rhs[ll] += 1. * V(al, ll, i, j + 1) + 1. * V(al, ll, i, j - 1);
    t1[ll] += 1. + 1.;
    for (ll = 1; ll < nl - 1; ll++) {
      rhs[ll] = -sqD * V(bl, ll, i, j);
      t0[ll] = -sqD * V(S, ll - 1, i, j) * o->idh0[ll];
      t2[ll] = -sqD * V(S, ll, i, j) * o->idh1[ll];
      t1[ll] = -t0[ll] - t2[ll];
      rhs[ll] += 1. * V(al, ll, i + 1, j) + 1. * V(al, ll, i - 1, j);
      t1[ll] += 1. + 1.;
      rhs[ll] += 1. * V(al, ll, i, j + 1) + 1. * V(al, ll, i, j - 1);
      t1[ll] += 1. + 1.;
    }
    ll = nl - 1;
    rhs[ll] = -sqD * V(bl, ll, i, j);
    t0[ll] = -sqD * V(S, ll - 1, i, j) * o->idh0[ll];
    t1[ll] = -t0[ll];
    rhs[ll] += 1. * V(al, ll, i + 1, j) + 1. * V(al, ll, i - 1, j);
    t1[ll] += 1. + 1.;
    rhs[ll] += 1. * V(al, ll, i, j + 1) + 1. * V(al, ll, i, j - 1);
    t1[ll] += 1. + 1.;
    /* Thomas :137-146 */
    for (ll = 1; ll < nl; ll++) {
      t1[ll] -= t0[ll] * t2[ll - 1] / t1[ll - 1];
      rhs[ll] -= t0[ll] * rhs[ll - 1] / t1[ll - 1];
    }
    V(al, nl - 1, i, j) = t0[nl - 1] = rhs[nl - 1] / t1[nl - 1];
    for (ll = nl - 2; ll >= 0; ll--) V(al, ll, i, j) = t0[ll] = (rhs[ll] - t2[ll] * t0[ll + 1]) / t1[ll];
  } else {
    double n = -sqD * V(bl, 0, i, j), d = 0;
    n += V(al, 0, i + 1, j) + V(al, 0, i - 1, j); d += 2.;
    n += V(al, 0, i, j + 1) + V(al, 0, i, j - 1); d += 2.;
    V(al, 0, i, j) = n / d;
  }
}

/* relax on one level.  ORC_GS_LEX: the reference traversal, x outer / y inner, in place
 * (poisson_layer.h:75; loop nesting confirmed by qg-node/inner-vertex.h:26-29).
 * ORC_GS_RB: red ((i+j) even) then black, the ordering the GPU uses.  In both, wall ghosts
 * lag: they hold -a as of the last boundary_level(). */
static void relax_level(orc_t *o, int k, fld *al, const fld *bl) {
  const fld *S = &o->S[k];
  const double D = o->delta[k];
  if (o->smoother == ORC_GS_LEX) {
    for (int i = 0; i < al->nx; i++)
      for (int j = 0; j < al->ny; j++) relax_column(o, al, bl, S, D, i, j);
  } else {
    for (int c = 0; c < 2; c++) {
#pragma omp parallel for
      for (int j = 0; j < al->ny; j++)
        for (int i = (j + c) & 1; i < al->nx; i += 2) relax_column(o, al, bl, S, D, i, j);
      /* red-black is the build's ordering; its definition refreshes the ghosts after every
       * colour (what a halo exchange between tiles does).  With walls this equals the
       * reference's once-per-sweep boundary_level(); with periodic BCs the black half-sweep
       * then sees the new red values across the seam. */
      if (c == 0) boundary(al);
    }
  }
}

/* msqg/poisson_layer.h:157-258 */
static double residual_layer(orc_t *o, fld *al, fld *bl, fld *resl) {
  const int nl = o->nl, nx = al->nx, ny = al->ny;
  const double D = o->delta[0];
  const fld *S = &o->S[0];
  double maxres = 0.;
#define FGX(a, l, i, j, s) ((V(a, l, (i) + (s), j) - V(a, l, (i) + (s) - 1, j)) / D)
#define FGY(a, l, i, j, s) ((V(a, l, i, (j) + (s)) - V(a, l, i, (j) + (s) - 1)) / D)
#pragma omp parallel for reduction(max : maxres)
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      for (int l = 0; l < nl; l++) {
        double r;
        if (nl == 1) r = V(bl, l, i, j);
        else if (l == 0) r = V(bl, l, i, j) + V(S, l, i, j) * (V(al, l, i, j) - V(al, l + 1, i, j)) * o->idh1[l];
        else if (l < nl - 1)
          r = V(bl, l, i, j) + V(S, l - 1, i, j) * (V(al, l, i, j) - V(al, l - 1, i, j)) * o->idh0[l] -
              V(S, l, i, j) * (V(al, l + 1, i, j) - V(al, l, i, j)) * o->idh1[l];
        else r = V(bl, l, i, j) + V(S, l - 1, i, j) * (V(al, l, i, j) - V(al, l - 1, i, j)) * o->idh0[l];
        r += (1. * FGX(al, l, i, j, 0) - 1. * FGX(al, l, i, j, 1)) / D;
        r += (1. * FGY(al, l, i, j, 0) - 1. * FGY(al, l, i, j, 1)) / D;
        V(resl, l, i, j) = r;
        if (fabs(r) > maxres) maxres = fabs(r);
      }
    }
  boundary(resl);
  return maxres;
}

/* [BASILISK RULE] restriction of cell scalars: mean of the 4 children, summed in
 * foreach_child order (x outer, y inner) */
static void restrict_fld(const fld *fine, fld *coarse) {
#pragma omp parallel for collapse(2)
  for (int l = 0; l < coarse->nl; l++)
    for (int j = 0; j < coarse->ny; j++)
      for (int i = 0; i < coarse->nx; i++) {
        double sum = 0.;
        sum += V(fine, l, 2 * i, 2 * j);
        sum += V(fine, l, 2 * i, 2 * j + 1);
        sum += V(fine, l, 2 * i + 1, 2 * j);
        sum += V(fine, l, 2 * i + 1, 2 * j + 1);
        V(coarse, l, i, j) = sum / 4;
      }
}

/* [BASILISK RULE] bilinear prolongation:
 * (9 c + 3 (c[child.x] + c[0,child.y]) + c[child.x,child.y]) / 16 */
static void prolong_fld(const fld *coarse, fld *fine) {
#pragma omp parallel for collapse(2)
  for (int l = 0; l < fine->nl; l++)
    for (int j = 0; j < fine->ny; j++)
      for (int i = 0; i < fine->nx; i++) {
        int I = i >> 1, J = j >> 1, cx = (i & 1) ? 1 : -1, cy = (j & 1) ? 1 : -1;
        V(fine, l, i, j) = (9. * V(coarse, l, I, J) + 3. * (V(coarse, l, I + cx, J) + V(coarse, l, I, J + cy)) + V(coarse, l, I + cx, J + cy)) / 16.;
      }
}

/* restrict S to all levels, msqg/poisson_layer.h:284 (done on every poisson_layer call) */
static void restrict_S(orc_t *o) {
  fld *S0 = &o->f[ORC_S];
  memcpy(o->S[0].d, S0->d, (size_t)S0->nl * (S0->nx + 2) * (S0->ny + 2) * sizeof(double));
  for (int k = 1; k < o->nlev; k++) { restrict_fld(&o->S[k - 1], &o->S[k]); boundary(&o->S[k]); }
}

/* mspg/elliptic.h:43-99 with minlevel = 1 */
static void mg_cycle(orc_t *o, fld *a, int nrelax) {
  for (int k = 1; k < o->nlev; k++) restrict_fld(&o->res[k - 1], &o->res[k]);
  for (int k = o->nlev - 1; k >= 0; k--) {
    fld *da = &o->da[k];
    if (k == o->nlev - 1) fld_zero(da);
    else prolong_fld(&o->da[k + 1], da);
    boundary(da);
    for (int it = 0; it < nrelax; it++) {
      relax_level(o, k, da, &o->res[k]);
      boundary(da);
    }
  }
  fld *da = &o->da[0];
#pragma omp parallel for collapse(2)
  for (int l = 0; l < a->nl; l++)
    for (int j = 0; j < a->ny; j++)
      for (int i = 0; i < a->nx; i++) V(a, l, i, j) += V(da, l, i, j);
  boundary(a);
}

/* mspg/elliptic.h:145-229 driven as msqg/poisson_layer.h:263-306 */
static orc_mgstats mg_solve(orc_t *o, fld *a, fld *b) {
  orc_mgstats s = {0, 0, 0, 0, 0};
  restrict_S(o);
  double sum = 0.;
  for (int l = 0; l < b->nl; l++)
    for (int i = 0; i < b->nx; i++)
      for (int j = 0; j < b->ny; j++) sum += V(b, l, i, j);
  s.sum = sum;
  s.nrelax = 4;
  double resb;
  resb = s.resb = s.resa = residual_layer(o, a, b, &o->res[0]);
  for (s.i = 0; s.i < o->nitermax && (s.i < o->nitermin || s.resa > o->tolerance); s.i++) {
    mg_cycle(o, a, s.nrelax);
    s.resa = residual_layer(o, a, b, &o->res[0]);
    if (s.resa > o->tolerance) {
      if (resb / s.resa < 1.2 && s.nrelax < 100) s.nrelax++;
      else if (resb / s.resa > 10 && s.nrelax > 2) s.nrelax--;
    }
    resb = s.resa;
  }
  if (s.resa > o->tolerance && !o->quiet)
    fprintf(stderr, "WARNING: convergence not reached after %d iterations\n  res: %g sum: %g nrelax: %d\n", s.i, s.resa, s.sum, s.nrelax);
  return s;
}

/* msqg/qg.h:114-163 */
static orc_mgstats invertq(orc_t *o, fld *po, fld *qo) {
  o->mg = mg_solve(o, po, qo);
  boundary(po);
  return o->mg;
}
orc_mgstats orc_invertq(orc_t *o, int psi, int q) { return invertq(o, &o->f[psi], &o->f[q]); }
orc_mgstats orc_last_mgstats(orc_t *o) { return o->mg; }

int orc_nlevels(orc_t *o) { return o->nlev; }
void orc_level_dims(orc_t *o, int lev, int *nx, int *ny) { *nx = o->nx >> lev; *ny = o->ny >> lev; }

void orc_relax_raw(orc_t *o, int lev, double *da, const double *res, int nsweeps) {
  restrict_S(o);
  fld *a = &o->da[lev], *b = &o->res[lev];
  fld_from_array(a, da);
  fld_from_array(b, res);
  for (int it = 0; it < nsweeps; it++) { relax_level(o, lev, a, b); boundary(a); }
  fld_to_array(a, da);
}
double orc_residual_raw(orc_t *o, const double *a, const double *b, double *res) {
  restrict_S(o);
  fld fa, fb;
  fld_alloc(&fa, o->nx, o->ny, o->nl, o->f[ORC_PSI].bc);
  fld_alloc(&fb, o->nx, o->ny, o->nl, o->f[ORC_PSI].bc);
  fld_from_array(&fa, a); fld_from_array(&fb, b);
  double m = residual_layer(o, &fa, &fb, &o->res[0]);
  fld_to_array(&o->res[0], res);
  fld_free(&fa); fld_free(&fb);
  return m;
}
void orc_restrict_raw(orc_t *o, int lev_fine, const double *fine, double *coarse, int nlay) {
  fld ff, fc;
  fld_alloc(&ff, o->nx >> lev_fine, o->ny >> lev_fine, nlay, BC_NEUMANN);
  fld_alloc(&fc, o->nx >> (lev_fine + 1), o->ny >> (lev_fine + 1), nlay, BC_NEUMANN);
  fld_from_array(&ff, fine);
  restrict_fld(&ff, &fc);
  fld_to_array(&fc, coarse);
  fld_free(&ff); fld_free(&fc);
}
void orc_prolong_raw(orc_t *o, int lev_coarse, const double *coarse, double *fine) {
  fld *c = &o->da[lev_coarse], *f = &o->da[lev_coarse - 1];
  fld_from_array(c, coarse); /* includes boundary_level(da): homogeneous BC */
  prolong_fld(c, f);
  fld_to_array(f, fine);
}

/* ------------------------------------------------------------------ set_const */

/* msqg/qg.h:931-1116 without the cwd file discovery (tests set fields through the API) */
/* filter length scale and wavelet coefficients, msqg/qg.h:1059-1090 (MODE_PV_INVERT off) */
static void build_siglev(orc_t *o) {
  fld *Rd = &o->f[ORC_RD];
  const int K = o->wnlev;
  for (int j = 0; j < o->ny; j++)
    for (int i = 0; i < o->nx; i++) V(&o->wsf[0], 0, i, j) = fmin(o->afilt * V(Rd, 0, i, j), o->Lfmax);
  boundary(&o->wsf[0]);
  for (int k = 1; k < K; k++) { restrict_fld(&o->wsf[k - 1], &o->wsf[k]); boundary(&o->wsf[k]); }   /* restriction({sig_filt}) */
  for (int k = 0; k < K; k++) {  /* low pass: for (l = depth(); l >= 0; l--) */
    fld *sl = &o->wsig[k], *sf = &o->wsf[k];
    const double Delta = o->L0 / (double)(o->nx >> k);
    for (int j = 0; j < sl->ny; j++)
      for (int i = 0; i < sl->nx; i++) {
        double ref_flag = 0;
        if (k > 0) {  /* foreach_child(): x outer, y inner */
          ref_flag += V(&o->wsig[k - 1], 0, 2 * i, 2 * j); ref_flag += V(&o->wsig[k - 1], 0, 2 * i, 2 * j + 1);
          ref_flag += V(&o->wsig[k - 1], 0, 2 * i + 1, 2 * j); ref_flag += V(&o->wsig[k - 1], 0, 2 * i + 1, 2 * j + 1);
        }
        const double s = V(sf, 0, i, j);
        if (ref_flag > 0) V(sl, 0, i, j) = 1;
        else if (s > 2 * Delta) V(sl, 0, i, j) = 0;
        else if (s <= 2 * Delta && s > Delta) V(sl, 0, i, j) = 1 - (s - Delta) / Delta;
        else V(sl, 0, i, j) = 1;
      }
    boundary(sl);
  }
  for (int k = 0; k < K; k++) {  /* high pass */
    fld *sl = &o->wsig[k];
    for (int j = 0; j < sl->ny; j++) for (int i = 0; i < sl->nx; i++) V(sl, 0, i, j) = 1 - V(sl, 0, i, j);
    boundary(sl);
  }
}

/* [BASILISK RULE] wavelet() / inverse_wavelet() of grid/multigrid-common.h (not in the tree; the reference's
 * own masked copies wavelet_mask() / inverse_wavelet_mask(), qg-node/wavelet_vertex.h:10-46, have exactly this
 * structure with an extra factor mask_c): s restricted to all levels (mean of 4 children, BC on every level);
 * detail w_k = s_k - bilinear(s_{k+1}) on levels finer than the root, w_root = s_root; inverse:
 * s_root = w_root, s_k = bilinear(s_{k+1}) + w_k with boundary_level after each level.
 * Here with the scaling by sig_lev in between (msqg/qg.h:532-538), all layers at once. */
static void wavelet_apply(orc_t *o, fld *po) {
  const int K = o->wnlev;
  const size_t sz = (size_t)po->nl * (po->nx + 2) * (po->ny + 2) * sizeof(double);
  memcpy(o->ws[0].d, po->d, sz);
  boundary(&o->ws[0]);
  for (int k = 1; k < K; k++) { restrict_fld(&o->ws[k - 1], &o->ws[k]); boundary(&o->ws[k]); }
  for (int k = 0; k < K - 1; k++) {
    fld *w = &o->ww[k], *s = &o->ws[k];
    prolong_fld(&o->ws[k + 1], w);   /* sp */
    for (int l = 0; l < w->nl; l++) for (int j = 0; j < w->ny; j++) for (int i = 0; i < w->nx; i++) {
      double d = V(s, l, i, j); d -= V(w, l, i, j);      /* w[] = s[]; w[] -= sp */
      V(w, l, i, j) = d * V(&o->wsig[k], 0, i, j);       /* w[] *= sig_lev[] */
    }
  }
  { fld *w = &o->ww[K - 1], *s = &o->ws[K - 1];
    for (int l = 0; l < w->nl; l++) for (int j = 0; j < w->ny; j++) for (int i = 0; i < w->nx; i++) V(w, l, i, j) = V(s, l, i, j) * V(&o->wsig[K - 1], 0, i, j); }
  memcpy(o->ws[K - 1].d, o->ww[K - 1].d, (size_t)po->nl * (o->ws[K - 1].nx + 2) * (o->ws[K - 1].ny + 2) * sizeof(double));
  boundary(&o->ws[K - 1]);
  for (int k = K - 2; k >= 0; k--) {
    fld *s = &o->ws[k], *w = &o->ww[k];
    prolong_fld(&o->ws[k + 1], s);
    for (int l = 0; l < s->nl; l++) for (int j = 0; j < s->ny; j++) for (int i = 0; i < s->nx; i++) V(s, l, i, j) += V(w, l, i, j);
    boundary(s);
  }
  memcpy(po->d, o->ws[0].d, sz);
}
static orc_mgstats invertq(orc_t *o, fld *po, fld *qo);
static void comp_q(orc_t *o, fld *po, fld *qo);
/* wavelet_filter msqg/qg.h:509-560.  `nbar` is passed by value there, so the caller's counter
 * never advances: qof = (tmp - q) / dtflt on every call. */
static void wavelet_filter(orc_t *o, fld *qof, double dtflt) {
  fld *q = &o->f[ORC_Q], *po = &o->f[ORC_PSI], *tmp = &o->f[ORC_TMP];
  const int nbar = 0;
  const size_t sz = (size_t)q->nl * (q->nx + 2) * (q->ny + 2) * sizeof(double);
  for (int l = 0; l < q->nl; l++) for (int j = 0; j < q->ny; j++) for (int i = 0; i < q->nx; i++) V(tmp, l, i, j) = V(q, l, i, j);
  invertq(o, po, q);
  wavelet_apply(o, po);
  comp_q(o, po, q);
  for (int l = 0; l < q->nl; l++) for (int j = 0; j < q->ny; j++) for (int i = 0; i < q->nx; i++)
    V(qof, l, i, j) = (V(qof, l, i, j) * nbar + (V(tmp, l, i, j) - V(q, l, i, j)) / dtflt) / (nbar + 1);
  if (dtflt < 0.0) { memcpy(q->d, tmp->d, sz); }   /* list_copy_deep(tmpl, qol) incl. ghosts */
  boundary(qof);
}
void orc_wavelet_filter(orc_t *o, double dtflt) { wavelet_filter(o, &o->f[ORC_QOF], dtflt); }
int orc_wavelet_levels(orc_t *o) { return o->wnlev; }
void orc_get_siglev(orc_t *o, int lev, double *a) { fld_to_array(&o->wsig[lev], a); }
void orc_wavelet_apply(orc_t *o, int field) { wavelet_apply(o, &o->f[field]); }

void orc_set_const(orc_t *o) {
  const int nl = o->nl;
  for (int l = 0; l < nl - 1; l++) o->dhc[l] = 0.5 * (o->dhf[l] + o->dhf[l + 1]);
  if (nl > 1) {
    o->idh0[0] = 0.;
    o->idh1[0] = 1. / (o->dhc[0] * o->dhf[0]);
    for (int l = 1; l < nl - 1; l++) {
      o->idh0[l] = 1. / (o->dhc[l - 1] * o->dhf[l]);
      o->idh1[l] = 1. / (o->dhc[l] * o->dhf[l]);
    }
    o->idh0[nl - 1] = 1. / (o->dhc[nl - 2] * o->dhf[nl - 1]);
    o->idh1[nl - 1] = 0.;
  }
  fld *Ro = &o->f[ORC_RO], *Fr = &o->f[ORC_FR], *S = &o->f[ORC_S];
  const double D = o->L0 / o->nx;
  for (int j = 0; j < o->ny; j++)
    for (int i = 0; i < o->nx; i++) {
      double y = (j + 0.5) * D;
      V(Ro, 0, i, j) = o->varRo > 0 ? o->Rom / (1 + o->Rom * o->beta * (y - 0.5 * o->L0)) : o->Rom;
    }
  for (int l = 0; l < nl - 1; l++)
    for (int j = 0; j < o->ny; j++)
      for (int i = 0; i < o->nx; i++) {
        double r = V(Fr, l, i, j) / V(Ro, 0, i, j);
        V(S, l, i, j) = r * r;
      }
  boundary(Ro); boundary(Fr); boundary(S);
  build_siglev(o);                                               /* :1059-1090 */
  comp_q(o, &o->f[ORC_PSI], &o->f[ORC_Q]);                       /* :1092 */
  if (o->flsrv == 1) comp_del2(o, &o->f[ORC_PSIPG], &o->f[ORC_ZETAPG], 0., 1.0); /* :1094-1097 */
  for (int k = 0; k < ORC_NFIELDS; k++) boundary(&o->f[k]);      /* :1103 */
}

/* ------------------------------------------------------------------ energy diagnostics, msqg/qg_energy.h */
#define EW(po, l, i, j) (-V(po, l, i, j) * (1 - ediag) + ediag)
static void comp_del2(orc_t *o, fld *po, fld *zeta, double add, double fac);
static void comp_stretch(orc_t *o, fld *po, fld *st, double add, double fac);
/* advection_de :28-158 (called with qol = zetal); ENERGY_CONSERV off; the _LS_RV term uses zeta_pg, which
 * is identically 0 unless flsrv = 1 */
static void advection_de(orc_t *o, fld *qo, fld *po, fld *j1, fld *j2, fld *j3, double dt, double ediag) {
  const int nl = o->nl;
  const double D = o->L0 / o->nx;
  fld *pp = &o->f[ORC_PSIPG], *qp = &o->f[ORC_ZETAPG], *S = &o->f[ORC_S];
  const double *idh0 = o->idh0, *idh1 = o->idh1;
#pragma omp parallel for
  for (int j = 0; j < po->ny; j++)
    for (int i = 0; i < po->nx; i++) {
      if (nl > 1) {
        double ju_1, jd_1, ju_2, jd_2, ju_3, jd_3, jc;
        int l = 0;
        jd_1 = jacobian(po, l, po, l + 1, i, j, D);
        jd_2 = jacobian(pp, l, po, l + 1, i, j, D);
        jd_3 = jacobian(po, l, pp, l + 1, i, j, D);
        jc = jacobian(po, l, pp, l, i, j, D);
        V(j1, l, i, j) += (jacobian(po, l, qo, l, i, j, D) + V(S, l, i, j) * jd_1 * idh1[l]) * dt * EW(po, l, i, j);
        V(j2, l, i, j) += (jacobian(pp, l, qo, l, i, j, D) + V(S, l, i, j) * (jd_2 + jc) * idh1[l]) * dt * EW(po, l, i, j);
        V(j3, l, i, j) += (BETA_EFFECT(po, l, i, j) + V(S, l, i, j) * (jd_3 - jc) * idh1[l]) * dt * EW(po, l, i, j);
        V(j3, l, i, j) += jacobian(po, l, qp, l, i, j, D) * dt * EW(po, l, i, j);
        for (l = 1; l < nl - 1; l++) {
          ju_1 = -jd_1; ju_2 = -jd_3; ju_3 = -jd_2; /* swap, :96-98 */
          jd_1 = jacobian(po, l, po, l + 1, i, j, D);
          jd_2 = jacobian(pp, l, po, l + 1, i, j, D);
          jd_3 = jacobian(po, l, pp, l + 1, i, j, D);
          jc = jacobian(po, l, pp, l, i, j, D);
          V(j1, l, i, j) += (jacobian(po, l, qo, l, i, j, D) + V(S, l - 1, i, j) * ju_1 * idh0[l] + V(S, l, i, j) * jd_1 * idh1[l]) * dt * EW(po, l, i, j);
          V(j2, l, i, j) += (jacobian(pp, l, qo, l, i, j, D) + V(S, l - 1, i, j) * (ju_2 + jc) * idh0[l] + V(S, l, i, j) * (jd_2 + jc) * idh1[l]) * dt * EW(po, l, i, j);
          V(j3, l, i, j) += (BETA_EFFECT(po, l, i, j) + V(S, l - 1, i, j) * (ju_3 - jc) * idh0[l] + V(S, l, i, j) * (jd_3 - jc) * idh1[l]) * dt * EW(po, l, i, j);
          V(j3, l, i, j) += jacobian(po, l, qp, l, i, j, D) * dt * EW(po, l, i, j);
        }
        l = nl - 1;
        ju_1 = -jd_1; ju_2 = -jd_3; ju_3 = -jd_2;
        jc = jacobian(po, l, pp, l, i, j, D);
        V(j1, l, i, j) += (jacobian(po, l, qo, l, i, j, D) + V(S, l - 1, i, j) * ju_1 * idh0[l]) * dt * EW(po, l, i, j);
        V(j2, l, i, j) += (jacobian(pp, l, qo, l, i, j, D) + V(S, l - 1, i, j) * (ju_2 + jc) * idh0[l]) * dt * EW(po, l, i, j);
        V(j3, l, i, j) += (BETA_EFFECT(po, l, i, j) + V(S, l - 1, i, j) * (ju_3 - jc) * idh0[l]) * dt * EW(po, l, i, j);
        V(j3, l, i, j) += jacobian(po, l, qp, l, i, j, D) * dt * EW(po, l, i, j);
      } else { V(j1, 0, i, j) = 0; V(j2, 0, i, j) = 0; V(j3, 0, i, j) = 0; }
    }
}
/* dissip_de :161-191 */
static void dissip_de(orc_t *o, fld *zeta, fld *dqo, fld *po, double dt, double ediag) {
  fld *tmp = &o->f[ORC_TMP], *tmp2 = &o->f[ORC_TMP2];
  const double D = o->L0 / o->nx, D2 = D * D;
  comp_del2(o, zeta, tmp, 0., 1.);
  comp_stretch(o, zeta, tmp2, 0., 1.);
  for (int l = 0; l < o->nl; l++) for (int j = 0; j < po->ny; j++) for (int i = 0; i < po->nx; i++) {
    V(dqo, l, i, j) += (V(tmp, l, i, j) + V(tmp2, l, i, j)) * o->iRe * dt * EW(po, l, i, j);
    V(dqo, l, i, j) += o->iRe4 * LAP(tmp, l, i, j, D2) * dt * EW(po, l, i, j);
  }
  comp_stretch(o, tmp, tmp2, 0., 1.);
  for (int l = 0; l < o->nl; l++) for (int j = 0; j < po->ny; j++) for (int i = 0; i < po->nx; i++)
    V(dqo, l, i, j) += o->iRe4 * (V(tmp2, l, i, j)) * dt * EW(po, l, i, j);
}
/* ekman_friction_de :193-206 */
static void ekman_friction_de(orc_t *o, fld *zeta, fld *dqo, fld *po, double dt, double ediag) {
  const int b = o->nl - 1;
  for (int j = 0; j < po->ny; j++) for (int i = 0; i < po->nx; i++) {
    V(dqo, 0, i, j) -= o->Eks / (o->Rom * 2 * o->dhf[0]) * V(zeta, 0, i, j) * dt * EW(po, 0, i, j);
    V(dqo, b, i, j) -= o->Ekb / (o->Rom * 2 * o->dhf[b]) * V(zeta, b, i, j) * dt * EW(po, b, i, j);
  }
}
/* filter_de :208-225 */
static void filter_de(orc_t *o, fld *pm, double dtflt, double ediag) {
  fld *tmp2 = &o->f[ORC_TMP2], *ft = &o->f[ORC_DE_FT];
  wavelet_filter(o, tmp2, -dtflt);
  for (int l = 0; l < o->nl; l++) for (int j = 0; j < ft->ny; j++) for (int i = 0; i < ft->nx; i++) {
    V(ft, l, i, j) += V(tmp2, l, i, j) * dtflt * (-V(pm, l, i, j) * (1 - ediag) + ediag);
    V(pm, l, i, j) = 0;
  }
  o->nme_ft = 0;
}
void orc_filter_de(orc_t *o, int pm_field, double dtflt) { filter_de(o, &o->f[pm_field], dtflt, (double)o->ediag); }
/* energy_tend :227-241 */
void orc_energy_tend(orc_t *o, double dt) {
  fld *po = &o->f[ORC_PSI], *zeta = &o->f[ORC_ZETA], *pm = &o->f[ORC_PO_MFT];
  const double ediag = (double)o->ediag;
  comp_del2(o, po, zeta, 0., 1.0);
  advection_de(o, zeta, po, &o->f[ORC_DE_J1], &o->f[ORC_DE_J2], &o->f[ORC_DE_J3], dt, ediag);
  dissip_de(o, zeta, &o->f[ORC_DE_VD], po, dt, ediag);
  ekman_friction_de(o, zeta, &o->f[ORC_DE_BF], po, dt, ediag);
  for (int l = 0; l < o->nl; l++) for (int j = 0; j < po->ny; j++) for (int i = 0; i < po->nx; i++)
    V(pm, l, i, j) = (V(pm, l, i, j) * o->nme_ft + V(po, l, i, j)) / (o->nme_ft + 1);
  o->nme_ft += 1;
}
void orc_reset_de(orc_t *o) { for (int k = ORC_DE_BF; k <= ORC_DE_FT; k++) fld_zero(&o->f[k]); }
/* pystep_de :296-349: ediag = 1 and dt = 1 are locals there; filter_de runs with po_mft = pol (so pol is zeroed)
 * and the global dtflt */
void orc_pystep_de(orc_t *o, const double *po_in, double *bf, double *vd, double *j1, double *j2, double *j3, double *ft, int onlyKE) {
  fld *po = &o->f[ORC_PSI], *zeta = &o->f[ORC_ZETA];
  const double ediag = 1., dt = 1.;
  fld_from_array(po, po_in);
  orc_reset_de(o);
  comp_del2(o, po, zeta, 0., 1.0);
  comp_q(o, po, &o->f[ORC_Q]);
  if (onlyKE == 1) { fld *S = &o->f[ORC_S]; for (int l = 0; l < o->nl - 1; l++) for (int j = 0; j < S->ny; j++) for (int i = 0; i < S->nx; i++) V(S, l, i, j) = 0.; }
  advection_de(o, zeta, po, &o->f[ORC_DE_J1], &o->f[ORC_DE_J2], &o->f[ORC_DE_J3], dt, ediag);
  dissip_de(o, zeta, &o->f[ORC_DE_VD], po, dt, ediag);
  ekman_friction_de(o, zeta, &o->f[ORC_DE_BF], po, dt, ediag);
  filter_de(o, po, o->dtflt, ediag);
  fld_to_array(&o->f[ORC_DE_BF], bf); fld_to_array(&o->f[ORC_DE_VD], vd); fld_to_array(&o->f[ORC_DE_J1], j1);
  fld_to_array(&o->f[ORC_DE_J2], j2); fld_to_array(&o->f[ORC_DE_J3], j3); fld_to_array(&o->f[ORC_DE_FT], ft);
}

/* ------------------------------------------------------------------ time stepping */

/* ptr_rhs, msqg/qg.h:574-588 */
static void ptr_rhs(orc_t *o, fld *c, fld *po, fld *dp) {
  fld *rel = &o->f[ORC_PTR_RELAX];
  const double D = o->L0 / o->nx, D2 = D * D;
  const int np = o->nptr;
#pragma omp parallel for
  for (int j = 0; j < po->ny; j++)
    for (int i = 0; i < po->nx; i++)
      for (int l = 0; l < o->nl; l++)
        for (int nt = 0; nt < np; nt++) {
          const int k = l * np + nt;
          V(dp, k, i, j) += jacobian(po, l, c, k, i, j, D) + o->iPe[nt] * LAP(c, k, i, j, D2) + o->ptr_ir[nt] * (V(rel, k, i, j) - V(c, k, i, j));
        }
}

/* msqg/qg.h:609-650 */
static double update_qg(orc_t *o, fld *q, fld *dq, double dtmax) {
  fld_zero(dq);
  invertq(o, &o->f[ORC_PSI], q);
  comp_del2(o, &o->f[ORC_PSI], &o->f[ORC_ZETA], 0., 1.0);
  dtmax = advection_pv(o, &o->f[ORC_ZETA], q, &o->f[ORC_PSI], dq, dtmax);
  dissip(o, &o->f[ORC_ZETA], dq);
  forcing_terms(o, &o->f[ORC_ZETA], &o->f[ORC_PSI], dq, 1);
  if (o->nptr > 0) { /* :634-647: the tracers are the part of `evolving` that goes with q */
    fld *c = q == &o->f[ORC_QPRED] ? &o->f[ORC_PTR_PRED] : &o->f[ORC_PTR];
    fld_zero(&o->f[ORC_DPTR]);
    ptr_rhs(o, c, &o->f[ORC_PSI], &o->f[ORC_DPTR]);
  }
  return dtmax;
}
double orc_update(orc_t *o, int q, int dq, double dtmax) { return update_qg(o, &o->f[q], &o->f[dq], dtmax); }

/* msqg/qg_stochastic.h:9 and :117-126; serial rand() in foreach order (x outer, y inner) */
static double normal_noise(void) {
  double a = sqrt(-2. * log(((double)(rand()) + 1.) / ((double)(RAND_MAX) + 2.)));
  return a * cos(2 * M_PI * rand() / (double)RAND_MAX);
}
static void generate_noise(orc_t *o) {
  fld *n = &o->f[ORC_NOISE], *s = &o->f[ORC_SIGMA];
  for (int i = 0; i < n->nx; i++)
    for (int j = 0; j < n->ny; j++)
      for (int l = 0; l < o->nl; l++) V(n, l, i, j) = o->amp_stoch * V(s, l, i, j) * normal_noise();
}

/* advance_qg for the tracer part of the lists, msqg/qg.h:597-605 */
static void advance_plain(fld *out, fld *in, fld *d, double dt) {
  for (int l = 0; l < out->nl; l++)
    for (int j = 0; j < out->ny; j++)
      for (int i = 0; i < out->nx; i++) V(out, l, i, j) = V(in, l, i, j) + V(d, l, i, j) * dt;
  boundary(out);
}

/* msqg/qg.h:594-606; stochastic: msqg/qg_stochastic.h:128-149 */
static void advance_qg(orc_t *o, fld *out, fld *in, fld *dq, double dt) {
  if (!o->stochastic) {
#pragma omp parallel for collapse(2)
    for (int l = 0; l < out->nl; l++)
      for (int j = 0; j < out->ny; j++)
        for (int i = 0; i < out->nx; i++) V(out, l, i, j) = V(in, l, i, j) + V(dq, l, i, j) * dt;
  } else {
    o->corrector_step = (o->corrector_step + 1) % 2;
    float dts = sqrt(dt);
    if (o->corrector_step) {
      generate_noise(o);
      dts = dts / sqrt(2);
    }
    fld *n = &o->f[ORC_NOISE];
    for (int l = 0; l < out->nl; l++)
      for (int j = 0; j < out->ny; j++)
        for (int i = 0; i < out->nx; i++) V(out, l, i, j) = V(in, l, i, j) + V(dq, l, i, j) * dt + V(n, l, i, j) * dts;
  }
  boundary(out);
}
void orc_advance(orc_t *o, int out, int in, int dq, double dt) { advance_qg(o, &o->f[out], &o->f[in], &o->f[dq], dt); }

/* msqg/qg.c:101-109 */
double orc_ke(orc_t *o) {
  fld *po = &o->f[ORC_PSI];
  double D = o->L0 / o->nx, D2 = D * D, ke = 0;
  for (int i = 0; i < po->nx; i++)
    for (int j = 0; j < po->ny; j++) ke -= 0.5 * V(po, 0, i, j) * LAP(po, 0, i, j, D2) * D2;
  return ke;
}

/* [BASILISK RULE] dtnext(): shorten dt so the next t-scheduled event is hit exactly */
static double dtnext(orc_t *o, double dt, double *tnext_out) {
  double tnext = o->tnext_event, t = o->t;
  if (tnext != HUGE_VAL && tnext > t) {
    unsigned int n = (unsigned int)((tnext - t) / dt);
    if (n == 0) dt = tnext - t;
    else {
      double dt1 = (tnext - t) / n;
      if (dt1 > dt * (1. + 1e-9)) dt = (tnext - t) / (n + 1);
      else if (dt1 < dt) dt = dt1;
      tnext = t + dt;
    }
  } else
    tnext = t + dt;
  *tnext_out = tnext;
  return dt;
}

/* [BASILISK RULE] one iteration of predictor-corrector.h run() */
int orc_step(orc_t *o) {
  fld *q = &o->f[ORC_Q], *dq = &o->f[ORC_DQ], *pred = &o->f[ORC_QPRED];
  double tnext;
  o->dt = dtnext(o, update_qg(o, q, dq, o->DT), &tnext);
  advance_qg(o, pred, q, dq, o->dt / 2.);
  if (o->nptr > 0) advance_plain(&o->f[ORC_PTR_PRED], &o->f[ORC_PTR], &o->f[ORC_DPTR], o->dt / 2.);
  update_qg(o, pred, dq, o->dt);
  advance_qg(o, q, q, dq, o->dt);
  if (o->nptr > 0) advance_plain(&o->f[ORC_PTR], &o->f[ORC_PTR], &o->f[ORC_DPTR], o->dt);
  o->t = tnext;
  o->iter++;
  return 0;
}
double orc_time(orc_t *o) { return o->t; }
double orc_dt(orc_t *o) { return o->dt; }
int orc_iter(orc_t *o) { return o->iter; }
void orc_set_tnext(orc_t *o, double tnext) { o->tnext_event = tnext; }

/* ------------------------------------------------------------------ pystep_bfn & co */

/* msqg/qg_bfn.h:21-80 (vartype == 1 only; vartype 0 is disabled in the reference) */
void orc_pystep_bfn(orc_t *o, const double *q_in, double *tend, double direction, int vartype) {
  double dtmax = o->DT;
  if (direction > 0) {
    o->iRe = o->Re == 0 ? 0. : 1 / o->Re;
    o->iRe4 = o->Re4 == 0 ? 0. : -1 / o->Re4;
    o->Eks = fabs(o->Eks); o->Ekb = fabs(o->Ekb);
  } else {
    o->iRe = o->Re == 0 ? 0. : -1 / o->Re;
    o->iRe4 = o->Re4 == 0 ? 0. : 1 / o->Re4;
    o->Eks = -fabs(o->Eks); o->Ekb = -fabs(o->Ekb);
  }
  fld *tl = &o->f[ORC_DQ];
  fld_zero(tl);
  if (vartype != 1) { fld_to_array(tl, tend); return; }
  fld_from_array(&o->f[ORC_Q], q_in);
  invertq(o, &o->f[ORC_PSI], &o->f[ORC_Q]);
  comp_del2(o, &o->f[ORC_PSI], &o->f[ORC_ZETA], 0., 1.0);
  dtmax = advection_pv(o, &o->f[ORC_ZETA], &o->f[ORC_Q], &o->f[ORC_PSI], tl, dtmax);
  dissip(o, &o->f[ORC_ZETA], tl);
  forcing_terms(o, &o->f[ORC_ZETA], &o->f[ORC_PSI], tl, 0); /* bfn path has no qforcing */
  fld_to_array(tl, tend);
}
/* msqg/qg_bfn.h:85-93 */
void orc_pyq2p(orc_t *o, double *psi_out, const double *q_in) {
  fld_zero(&o->f[ORC_PSI]);
  fld_from_array(&o->f[ORC_Q], q_in);
  invertq(o, &o->f[ORC_PSI], &o->f[ORC_Q]);
  fld_to_array(&o->f[ORC_PSI], psi_out);
}
/* msqg/qg_bfn.h:95-103 */
void orc_pyp2q(orc_t *o, const double *psi_in, double *q_out) {
  fld_zero(&o->f[ORC_Q]);
  fld_from_array(&o->f[ORC_PSI], psi_in);
  comp_q(o, &o->f[ORC_PSI], &o->f[ORC_Q]);
  fld_to_array(&o->f[ORC_Q], q_out);
}

/* ------------------------------------------------------------------ .bas IO */

/* msqg/auxiliar_input.h:101-149: per layer (float)n, n y-coords, then n rows of
 * x-coord + n values f(x_i, y_j) */
int orc_write_bas(orc_t *o, int field, const char *path) {
  fld *f = &o->f[field];
  if (f->nx != f->ny) return -2;
  FILE *fp = fopen(path, "w");
  if (!fp) return -1;
  int n = f->nx;
  float fn = n, Delta = o->L0 / fn;
  for (int l = 0; l < f->nl; l++) {
    fwrite(&fn, sizeof(float), 1, fp);
    for (int j = 0; j < n; j++) { float yp = Delta * j + 0. + Delta / 2.; fwrite(&yp, sizeof(float), 1, fp); }
    for (int i = 0; i < n; i++) {
      float xp = Delta * i + 0. + Delta / 2.;
      fwrite(&xp, sizeof(float), 1, fp);
      for (int j = 0; j < n; j++) { float v = (float)V(f, l, i, j); fwrite(&v, sizeof(float), 1, fp); }
    }
  }
  fclose(fp);
  return 0;
}

/* msqg/auxiliar_input.h:24-59: n from the file, nearest-cell sampling onto the model grid */
int orc_read_bas(orc_t *o, int field, const char *path) {
  fld *f = &o->f[field];
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  const double D = o->L0 / o->nx;
  for (int l = 0; l < f->nl; l++) {
    float width = 0;
    if (fread(&width, sizeof(float), 1, fp) != 1) { fclose(fp); return -3; }
    int n = (int)width;
    float *v = (float *)malloc((size_t)n * n * sizeof(float)), *yp = (float *)malloc(n * sizeof(float)), xp;
    if (fread(yp, sizeof(float), n, fp) != (size_t)n) { fclose(fp); return -3; }
    for (int i = 0; i < n; i++) {
      if (fread(&xp, sizeof(float), 1, fp) != 1) { fclose(fp); return -3; }
      if (fread(v + (size_t)i * n, sizeof(float), n, fp) != (size_t)n) { fclose(fp); return -3; }
    }
    for (int jj = 0; jj < f->ny; jj++)
      for (int ii = 0; ii < f->nx; ii++) {
        double x = (ii + 0.5) * D, y = (jj + 0.5) * D;
        int i = (x - 0.) * width / o->L0, j = (y - 0.) * width / o->L0;
        V(f, l, ii, jj) = (i >= 0 && i < width && j >= 0 && j < width) ? v[(size_t)i * n + j] : 0.;
      }
    free(v); free(yp);
  }
  fclose(fp);
  boundary(f);
  return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
