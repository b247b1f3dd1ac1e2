/*
 * qgnode_oracle.c -- CPU restatement ("oracle") of the VERTEX-grid QG model of the reference
 * (qg-node/qg.h, qg-node/qg_baroclinic_ms.h [-DLAYERS=1, nl >= 2], qg-node/qg_barotropic.h
 * [nl = 1], qg-node/nodal-poisson.h, qg-node/my_vertex.h, driver qg-node/qg.c).
 *
 * TEST INFRASTRUCTURE ONLY (see qg_oracle.h).  PARITY UNPINNED: the reference is Basilisk-C,
 * cannot be built here and ships no golden vectors; pinned by analytic known-answer tests
 * (tests/test_oracle_node_kat.py).
 *
 * Unknowns live on the (N+1)^2 vertices x = i*D, y = j*D, i, j = 0..N; arrays are
 * [layer][j][i] (qg-node/netcdf_vertex_bas.h:253).  [BASILISK RULE, SURVEY App. B (11)]: a
 * vertex-scalar boundary condition writes the boundary vertex itself, evaluated with the
 * first interior vertex next to it ("First interior point minus boundary point in vertex
 * convention", qg-node/qg.h:207-214); x-direction boundaries are applied first, then y.
 * Everything is multiplied by `mask` (1 inside, 0 on the boundary vertices and on land).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define ORN_MAXNL 16

enum { ORN_PSI = 0, ORN_Q, ORN_ZETA, ORN_TMP, ORN_PSIPG, ORN_S2 /* nl-1 */, ORN_TOPO /* 1 */, ORN_QFORC /* 1 */, ORN_MASK /* 1 */, ORN_DQ,
       ORN_QPRED, ORN_QFORC3D /* q_forcing_3d, -DFORCING_3D, qg_baroclinic_ms.h:25,179-185 */,
       /* surface-QG variant (params key sqg = 1), the finished parts of qg-node/sqg_baroclinic_ms.h */
       ORN_BS /* bs: surface buoyancy, 1 layer (:77-86, argument `bs` of comp_stretch) */,
       ORN_S2S /* S2 of the surface "layer": N2[0] before, f/N2[0] after set_const (:545), 1 layer */,
       ORN_PSIF /* psi_f: running mean of the filtered part, qg_baroclinic_ms.h:30,384 */,
       ORN_QEFF /* scratch: rhs of the elliptic problem = q minus the surface term */, ORN_D2BS /* scratch: laplacian(bs), :160-168 */,
       ORN_NFIELDS };

typedef struct { int n, nl; double *d; } vf; /* (n+3)^2 per layer: vertices -1..n+1 */
#define VI(f, l, i, j) ((((size_t)(l) * ((f)->n + 3)) + (size_t)((j) + 1)) * ((f)->n + 3) + (size_t)((i) + 1))
#define W(f, l, i, j) ((f)->d[VI(f, l, i, j)])

typedef struct { int i; double resb, resa, sum; int nrelax; } orn_mgstats;

typedef struct {
  int N, nl, flag_ms, sqg;
  double L0, f0, beta, hEkb, tau0, tau1, tf1, tf2, dy_ws, forc_mode, nu, nu4, gp_low, iRd2_low, scale_topo, bc_fac, psi_bc;
  double DT, tend, dtout, CFL, TOLERANCE, noise_init;
  double Lfmax, Lfmin, fac_filt_Rd, dtflt; int nbar; /* wavelet filter of the vertex model, qg-node/qg.h:118-122 */
  double **wsig, **wmc, **ws, **ww; int wv_ready; /* cell pyramids of the filter: coefficients, mask_c, s, w (nl layers) */
  double dh[ORN_MAXNL], N2[ORN_MAXNL], idh0[ORN_MAXNL], idh1[ORN_MAXNL];
  int nitermax, nitermin, nrelax, smoother, quiet;
  vf f[ORN_NFIELDS];
  int nlev;           /* level k (0 = finest) has (N >> k) + 1 vertices per side; coarsest: 2 cells */
  vf *da, *res, *mask, *S2;
  double t, dt, tnext_event, previous;
  int iter;
  orn_mgstats mg;
  /* stochastic forcing (-D_STOCHASTIC): qg-node/qg_stochastic.h, qg-node/qg.h:306-320.  n_stoch and sig_lev are
   * CELL scalars ("random noise define on scalar instead of vertex in order to use the wavelet transform") */
  int stochastic, corrector_step, cnlev, forcing_3d;
  double amp_stoch, L_filt;
  double **cs, **cw, **csig; /* cell pyramids, level k: (N >> k)^2 cells with one ghost ring */
} orn_t;

static void vf_alloc(vf *f, int n, int nl) { f->n = n; f->nl = nl; f->d = (double *)calloc((size_t)nl * (n + 3) * (n + 3), sizeof(double)); }
static void vf_zero(vf *f) { memset(f->d, 0, (size_t)f->nl * (f->n + 3) * (f->n + 3) * sizeof(double)); }

/* ---------------------------------------------------------------- params (qg-node/qg.c:72-107, extra.h:83-116) */
static void trim(char *s) { const char *d = s; do { while (*d == ' ') ++d; } while ((*s++ = *d++)); }
static void arr(char *s, double *a) { int n = 0; for (char *p = strtok(s, "[,]"); p && n < ORN_MAXNL; p = strtok(NULL, ",")) a[n++] = atof(p); }

orn_t *orn_create_str(const char *text) {
  orn_t *o = (orn_t *)calloc(1, sizeof(orn_t));
  /* defaults qg-node/qg.h:104-127, qg.c:61-66, Basilisk globals */
  o->N = 64; o->nl = 1; o->L0 = 1; o->f0 = 1.; o->tend = 100; o->dtout = 1; o->dh[0] = 1.; o->N2[0] = 1.; o->scale_topo = 1.;
  o->tf1 = 1; o->tf2 = 1; o->dy_ws = 1; o->forc_mode = 2.0; o->DT = 1e10; o->CFL = 0.5; o->TOLERANCE = 1e-3;
  o->nitermax = 100; o->nitermin = 1; o->nrelax = 5; o->Lfmax = 1e30; o->Lfmin = 1e30; o->dtflt = -1; /* HUGE, qg-node/qg.h:119-120 */
  char *copy = strdup(text), *save = NULL;
  for (char *line = strtok_r(copy, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    char b[300]; strncpy(b, line, 299); b[299] = 0; trim(b);
    char *eq = strchr(b, '='); if (!eq) continue; *eq = 0; char *k = b, *v = eq + 1;
    char *e2 = strchr(v, '='); if (e2) *e2 = 0; if (!*v) continue;
#define KD(name, field) else if (!strcmp(k, name)) o->field = atof(v)
#define KI(name, field) else if (!strcmp(k, name)) o->field = atoi(v)
    if (0) {}
    KI("N", N); KI("nl", nl); KI("flag_ms", flag_ms); KI("sqg", sqg); KD("L0", L0); KD("f0", f0); KD("beta", beta); KD("nu", nu); KD("nu4", nu4);
    KD("hEkb", hEkb); KD("gp_low", gp_low); KD("scale_topo", scale_topo); KD("tau0", tau0); KD("tau1", tau1); KD("tf1", tf1);
    KD("tf2", tf2); KD("dy_ws", dy_ws); KD("forc_mode", forc_mode); KD("noise_init", noise_init); KD("bc_fac", bc_fac); KD("DT", DT);
    KD("tend", tend); KD("dtout", dtout); KD("CFL", CFL); KD("TOLERANCE", TOLERANCE); KD("amp_stoch", amp_stoch); KD("L_filt", L_filt);
    KD("Lfmax", Lfmax); KD("Lfmin", Lfmin); KD("fac_filt_Rd", fac_filt_Rd); KD("dtflt", dtflt);
    else if (!strcmp(k, "dh")) arr(v, o->dh);
    else if (!strcmp(k, "N2")) arr(v, o->N2);
  }
  free(copy);
  if (o->nl < 1 || o->nl > ORN_MAXNL || o->N < 2 || (o->N & (o->N - 1))) { free(o); return NULL; }
  const int N = o->N, nl = o->nl, nlm = nl > 1 ? nl - 1 : 1;
  if (o->sqg && nl < 2) { free(o); return NULL; }
  for (int k = 0; k < ORN_NFIELDS; k++)
    vf_alloc(&o->f[k], N, (k == ORN_S2) ? nlm : (k == ORN_TOPO || k == ORN_QFORC || k == ORN_MASK || k == ORN_BS || k == ORN_S2S || k == ORN_D2BS) ? 1 : nl);
  int n = 0; while ((N >> n) >= 2) n++;
  o->nlev = n;
  o->da = (vf *)calloc(n, sizeof(vf)); o->res = (vf *)calloc(n, sizeof(vf)); o->mask = (vf *)calloc(n, sizeof(vf)); o->S2 = (vf *)calloc(n, sizeof(vf));
  for (int k = 0; k < n; k++) { vf_alloc(&o->da[k], N >> k, nl); vf_alloc(&o->res[k], N >> k, nl); vf_alloc(&o->mask[k], N >> k, 1); vf_alloc(&o->S2[k], N >> k, nlm); }
  /* set_vars qg-node/qg.h:404-450: mask = 1 on every vertex, then its BC (0 on the 4 walls) */
  vf *mk = &o->f[ORN_MASK];
  for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) W(mk, 0, i, j) = (i == 0 || i == N || j == 0 || j == N) ? 0. : 1.;
  /* S2[] = N2[l] (qg_baroclinic_ms.h:471-476) */
  /* sqg: S2 has nl layers there, layer 0 = the surface (sqg_baroclinic_ms.h:519-523); its layers 1..nl-1 are the
   * interfaces below layers 0..nl-2, i.e. exactly S2[0..nl-2] of the baroclinic model: N2 = [surface, interfaces...] */
  for (int l = 0; l < nl - 1; l++) for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) W(&o->f[ORN_S2], l, i, j) = o->N2[l + (o->sqg ? 1 : 0)];
  if (o->sqg) for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) W(&o->f[ORN_S2S], 0, i, j) = o->N2[0];
  { int c = 1; while ((N >> c) >= 1) c++; o->cnlev = c; /* Basilisk levels depth() ... 0 */
    o->cs = (double **)calloc(c, sizeof(double *)); o->cw = (double **)calloc(c, sizeof(double *)); o->csig = (double **)calloc(c, sizeof(double *));
    for (int k = 0; k < c; k++) { size_t sz = (size_t)((N >> k) + 2) * ((N >> k) + 2);
      o->cs[k] = (double *)calloc(sz, sizeof(double)); o->cw[k] = (double *)calloc(sz, sizeof(double)); o->csig[k] = (double *)calloc(sz, sizeof(double)); } }
  o->dt = 1.; o->tnext_event = HUGE_VAL;
  return o;
}
void orn_destroy(orn_t *o) {
  for (int k = 0; k < ORN_NFIELDS; k++) free(o->f[k].d);
  for (int k = 0; k < o->nlev; k++) { free(o->da[k].d); free(o->res[k].d); free(o->mask[k].d); free(o->S2[k].d); }
  for (int k = 0; k < o->cnlev; k++) { free(o->cs[k]); free(o->cw[k]); free(o->csig[k]); }
  free(o->cs); free(o->cw); free(o->csig);
  free(o->da); free(o->res); free(o->mask); free(o->S2); free(o);
}
int orn_set_option(orn_t *o, const char *k, double v) {
  if (!strcmp(k, "smoother")) o->smoother = (int)v; else if (!strcmp(k, "TOLERANCE")) o->TOLERANCE = v;
  else if (!strcmp(k, "NITERMAX")) o->nitermax = (int)v; else if (!strcmp(k, "NITERMIN")) o->nitermin = (int)v;
  else if (!strcmp(k, "quiet")) o->quiet = (int)v; else if (!strcmp(k, "DT")) o->DT = v;
  else if (!strcmp(k, "forcing_3d")) o->forcing_3d = (int)v;
  else if (!strcmp(k, "stochastic")) o->stochastic = (int)v; else if (!strcmp(k, "seed")) srand((unsigned)v); else return -1;
  return 0;
}
double orn_get_param(orn_t *o, const char *k) {
  if (!strcmp(k, "N")) return o->N;
  if (!strcmp(k, "nl")) return o->nl;
  if (!strcmp(k, "L0")) return o->L0;
  if (!strcmp(k, "DT")) return o->DT;
  if (!strcmp(k, "nlevels")) return o->nlev;
  if (!strcmp(k, "iRd2_low")) return o->iRd2_low;
  if (!strcmp(k, "bc_fac")) return o->bc_fac;
  if (!strcmp(k, "sqg")) return o->sqg;
  if (!strcmp(k, "tend")) return o->tend;
  if (!strcmp(k, "dtout")) return o->dtout;
  if (!strncmp(k, "idh0_", 5)) return o->idh0[atoi(k + 5)];
  if (!strncmp(k, "idh1_", 5)) return o->idh1[atoi(k + 5)];
  return NAN;
}
int orn_nlayers_of(orn_t *o, int f) { return o->f[f].nl; }
void orn_set_field(orn_t *o, int fi, const double *a) {
  vf *f = &o->f[fi]; const int n1 = f->n + 1;
  for (int l = 0; l < f->nl; l++) for (int j = 0; j < n1; j++) for (int i = 0; i < n1; i++) W(f, l, i, j) = a[((size_t)l * n1 + j) * n1 + i];
}
void orn_get_field(orn_t *o, int fi, double *a) {
  vf *f = &o->f[fi]; const int n1 = f->n + 1;
  for (int l = 0; l < f->nl; l++) for (int j = 0; j < n1; j++) for (int i = 0; i < n1; i++) a[((size_t)l * n1 + j) * n1 + i] = W(f, l, i, j);
}

/* ---------------------------------------------------------------- boundary conditions */
/* f_bnd = c * (g_first_interior - g_bnd_value): q, zeta from psi (g_bnd = psi_bc), qg-node/qg.h:206-214,
 * qg_baroclinic_ms.h:61-64; tmp from zeta (g_bnd = zeta at the boundary vertex), :66-69 */
static void bnd_from(vf *f, const vf *g, double c, int use_g_bnd, double gbc) {
  const int n = f->n;
  for (int l = 0; l < f->nl; l++) {
    for (int j = 0; j <= n; j++) {
      W(f, l, 0, j) = c * (W(g, l, 1, j) - (use_g_bnd ? W(g, l, 0, j) : gbc));
      W(f, l, n, j) = c * (W(g, l, n - 1, j) - (use_g_bnd ? W(g, l, n, j) : gbc));
    }
    for (int i = 0; i <= n; i++) {
      W(f, l, i, 0) = c * (W(g, l, i, 1) - (use_g_bnd ? W(g, l, i, 0) : gbc));
      W(f, l, i, n) = c * (W(g, l, i, n - 1) - (use_g_bnd ? W(g, l, i, n) : gbc));
    }
  }
}
static void bnd_const(vf *f, double v) {
  const int n = f->n;
  for (int l = 0; l < f->nl; l++) {
    for (int j = 0; j <= n; j++) { W(f, l, 0, j) = v; W(f, l, n, j) = v; }
    for (int i = 0; i <= n; i++) { W(f, l, i, 0) = v; W(f, l, i, n) = v; }
  }
}
static double bcc(const orn_t *o, int lev) { double D = o->L0 / (o->N >> lev); return 2 * o->bc_fac / (D * D); }
/* boundary({psi}), boundary({q}), boundary({zeta}), boundary({tmp}) with set_bc_ms() in force */
static void bnd_psi(orn_t *o) { bnd_const(&o->f[ORN_PSI], o->psi_bc); }
static void bnd_q(orn_t *o, vf *q) { bnd_from(q, &o->f[ORN_PSI], bcc(o, 0), 0, o->psi_bc); }
/* sqg_baroclinic_ms.h:64-67: the tmp rule subtracts psi_bc instead of the boundary value of zeta */
static void bnd_tmp(orn_t *o) { bnd_from(&o->f[ORN_TMP], &o->f[ORN_ZETA], bcc(o, 0), o->sqg ? 0 : 1, o->sqg ? o->psi_bc : 0.); }

/* ---------------------------------------------------------------- operators (qg-node/qg.h:176-190) */
#define LAPN(p, l, i, j, D2) ((W(p, l, (i) + 1, j) + W(p, l, (i) - 1, j) + W(p, l, i, (j) + 1) + W(p, l, i, (j) - 1) - 4 * W(p, l, i, j)) / (D2))
static inline double jac(const vf *p, int lp, const vf *q, int lq, int i, int j, double D) { /* +J(p,q), qg.h:178-188 */
#define P(a, b) W(p, lp, i + (a), j + (b))
#define Q(a, b) W(q, lq, i + (a), j + (b))
  return (((P(1, 0) - P(-1, 0)) * (Q(0, 1) - Q(0, -1)) + (P(0, -1) - P(0, 1)) * (Q(1, 0) - Q(-1, 0)) + P(1, 0) * (Q(1, 1) - Q(1, -1)) -
           P(-1, 0) * (Q(-1, 1) - Q(-1, -1)) - P(0, 1) * (Q(1, 1) - Q(-1, 1)) + P(0, -1) * (Q(1, -1) - Q(-1, -1)) + Q(0, 1) * (P(1, 1) - P(-1, 1)) -
           Q(0, -1) * (P(1, -1) - P(-1, -1)) - Q(1, 0) * (P(1, 1) - P(1, -1)) + Q(-1, 0) * (P(-1, 1) - P(-1, -1))) /
          (12. * D * D));
#undef P
#undef Q
}
#define BETAV(p, l, i, j) (o->beta * (W(p, l, (i) + 1, j) - W(p, l, (i) - 1, j)) / (2 * D))

/* comp_del2 qg-node/qg.h:230-241: all vertices, then boundary(out) according to what `out` is */
enum { OUT_ZETA, OUT_TMP, OUT_NONE };
static void comp_del2(orn_t *o, vf *in, vf *out, double add, double fac, int kind) {
  const int n = o->N; const double D = o->L0 / n, D2 = D * D;
  for (int l = 0; l < in->nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++)
    W(out, l, i, j) = add * W(out, l, i, j) + fac * LAPN(in, l, i, j, D2);
  if (kind == OUT_ZETA) bnd_q(o, out); else if (kind == OUT_TMP) bnd_tmp(o);
}
/* comp_stretch qg_baroclinic_ms.h:79-100 (the boundary(stretch) of the target is irrelevant: masked later) */
static void comp_stretch(orn_t *o, vf *in, vf *out, double add, double fac) {
  const int n = o->N, nl = o->nl; vf *S2 = &o->f[ORN_S2];
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) {
    int l = 0;
    W(out, l, i, j) = add * W(out, l, i, j) + fac * W(S2, l, i, j) * (W(in, l + 1, i, j) - W(in, l, i, j)) * o->idh1[l];
    for (l = 1; l < nl - 1; l++)
      W(out, l, i, j) = add * W(out, l, i, j) + fac * (W(S2, l - 1, i, j) * (W(in, l - 1, i, j) - W(in, l, i, j)) * o->idh0[l] + W(S2, l, i, j) * (W(in, l + 1, i, j) - W(in, l, i, j)) * o->idh1[l]);
    l = nl - 1;
    W(out, l, i, j) = add * W(out, l, i, j) + fac * W(S2, l - 1, i, j) * (W(in, l - 1, i, j) - W(in, l, i, j)) * o->idh0[l];
  }
}

/* comp_stretch(psi, bs, stretch, add, fac) of sqg_baroclinic_ms.h:77-98: S2 there has nl layers with layer 0 at the
 * surface, so its S2[] / S2[0,0,1] of layer l are S2S / S2[0] for l = 0 and S2[l-1] / S2[l] below; idh0[0] = 1/dh[0] (:502) */
static void comp_stretch_sqg(orn_t *o, vf *in, vf *bs, vf *out, double add, double fac) {
  const int n = o->N, nl = o->nl; vf *S2 = &o->f[ORN_S2], *S2S = &o->f[ORN_S2S];
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) {
    int l = 0;
    W(out, l, i, j) = add * W(out, l, i, j) + fac * (W(S2S, 0, i, j) * W(bs, 0, i, j) * o->idh0[0] - W(S2, l, i, j) * (W(in, l, i, j) - W(in, l + 1, i, j)) * o->idh1[l]);
    for (l = 1; l < nl - 1; l++)
      W(out, l, i, j) = add * W(out, l, i, j) + fac * (W(S2, l - 1, i, j) * (W(in, l - 1, i, j) - W(in, l, i, j)) * o->idh0[l] - W(S2, l, i, j) * (W(in, l, i, j) - W(in, l + 1, i, j)) * o->idh1[l]);
    l = nl - 1;
    W(out, l, i, j) = add * W(out, l, i, j) + fac * (-W(S2, l - 1, i, j) * (W(in, l, i, j) - W(in, l - 1, i, j))) * o->idh0[l];
  }
}
/* del2_bs / del4_bs of sqg_baroclinic_ms.h:160-168, 187-195: laplacian(bs) on every vertex, then neumann(0) on the four
 * sides ([BASILISK RULE] a vertex-scalar BC writes the boundary vertex: it takes the first interior value; x sides first) */
static void lap_bs(orn_t *o) {
  const int n = o->N; const double D = o->L0 / n, D2 = D * D; vf *bs = &o->f[ORN_BS], *d = &o->f[ORN_D2BS];
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(d, 0, i, j) = LAPN(bs, 0, i, j, D2);
  for (int j = 0; j <= n; j++) { W(d, 0, 0, j) = W(d, 0, 1, j); W(d, 0, n, j) = W(d, 0, n - 1, j); }
  for (int i = 0; i <= n; i++) { W(d, 0, i, 0) = W(d, 0, i, 1); W(d, 0, i, n) = W(d, 0, i, n - 1); }
}

/* rhs_pv_baroclinic qg_baroclinic_ms.h:104-196 / rhs_pv_barotropic qg_barotropic.h:16-29 */
static void rhs_pv(orn_t *o, vf *q, vf *dq) {
  const int n = o->N, nl = o->nl; const double D = o->L0 / n, D2 = D * D;
  vf *psi = &o->f[ORN_PSI], *zeta = &o->f[ORN_ZETA], *tmp = &o->f[ORN_TMP], *pg = &o->f[ORN_PSIPG], *S2 = &o->f[ORN_S2];
  vf *topo = &o->f[ORN_TOPO], *qf = &o->f[ORN_QFORC], *mk = &o->f[ORN_MASK];
  if (nl == 1) {
    for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++)
      W(dq, 0, i, j) = -jac(psi, 0, q, 0, i, j, D) - BETAV(psi, 0, i, j) - o->hEkb * o->f0 / (2 * o->dh[nl - 1]) * W(q, 0, i, j) + W(qf, 0, i, j) + o->nu * LAPN(q, 0, i, j, D2);
    return;
  }
  for (int l = 0; l < nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) { W(q, l, i, j) *= W(mk, 0, i, j); W(psi, l, i, j) *= W(mk, 0, i, j); }
  comp_del2(o, psi, zeta, 0., 1., OUT_ZETA);
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) {
    int l = 0; double ju, jd;
    jd = jac(psi, l, psi, l + 1, i, j, D) + jac(pg, l, psi, l + 1, i, j, D) + jac(psi, l, pg, l + 1, i, j, D);
    W(dq, l, i, j) = -jac(psi, l, zeta, l, i, j, D) - jac(pg, l, zeta, l, i, j, D) - W(S2, l, i, j) * jd * o->idh1[l] - BETAV(psi, l, i, j);
    for (l = 1; l < nl - 1; l++) {
      ju = -jd;
      jd = jac(psi, l, psi, l + 1, i, j, D) + jac(pg, l, psi, l + 1, i, j, D) + jac(psi, l, pg, l + 1, i, j, D);
      W(dq, l, i, j) = -jac(psi, l, zeta, l, i, j, D) - jac(pg, l, zeta, l, i, j, D) - W(S2, l, i, j) * jd * o->idh1[l] - W(S2, l - 1, i, j) * ju * o->idh0[l] - BETAV(psi, l, i, j);
    }
    l = nl - 1; ju = -jd;
    W(dq, l, i, j) = -jac(psi, l, zeta, l, i, j, D) - jac(pg, l, zeta, l, i, j, D) - W(S2, l - 1, i, j) * ju * o->idh0[l] - BETAV(psi, l, i, j);
    W(dq, l, i, j) += -o->hEkb * o->f0 / (2 * o->dh[nl - 1]) * W(zeta, l, i, j) - jac(psi, l, topo, 0, i, j, D) * o->f0 / o->dh[nl - 1];
  }
  if (o->sqg) { lap_bs(o); comp_stretch_sqg(o, zeta, &o->f[ORN_D2BS], dq, 1., o->nu); }   /* sqg_baroclinic_ms.h:160-174 */
  else comp_stretch(o, zeta, dq, 1., o->nu);
  comp_del2(o, zeta, tmp, 0., 1.0, OUT_TMP);
  for (int l = 0; l < nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(dq, l, i, j) += o->nu * W(tmp, l, i, j);
  const double minus_nu4 = -o->nu4;
  if (o->sqg) comp_stretch_sqg(o, tmp, &o->f[ORN_D2BS], dq, 1., minus_nu4);   /* del4_bs is laplacian(bs) again, :187-201 */
  else comp_stretch(o, tmp, dq, 1., minus_nu4);
  comp_del2(o, tmp, dq, 1., minus_nu4, OUT_NONE);
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(dq, 0, i, j) += W(qf, 0, i, j);
  if (o->forcing_3d) for (int l = 0; l < nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(dq, l, i, j) += W(&o->f[ORN_QFORC3D], l, i, j);
  for (int l = 0; l < nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(dq, l, i, j) *= W(mk, 0, i, j);
}

/* comp_q_baroclinic qg_baroclinic_ms.h:199-211 / comp_q_barotropic qg_barotropic.h:32-39 */
static void comp_q(orn_t *o, vf *psi, vf *q) {
  const int n = o->N; const double D = o->L0 / n, D2 = D * D;
  if (o->nl == 1) {
    for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(q, 0, i, j) = LAPN(psi, 0, i, j, D2) - o->iRd2_low * W(psi, 0, i, j);
  } else {
    for (int l = 0; l < o->nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(q, l, i, j) = LAPN(psi, l, i, j, D2);
    /* sqg: the reference's comp_q_baroclinic (:232-243) still calls the 4-argument comp_stretch and does not compile;
     * completed here with the surface buoyancy bs as the `bs` argument */
    if (o->sqg) comp_stretch_sqg(o, psi, &o->f[ORN_BS], q, 1., 1.);
    else comp_stretch(o, psi, q, 1., 1.);
  }
  bnd_q(o, q);
}

/* ---------------------------------------------------------------- vertex multigrid */
static void relax_col(const orn_t *o, vf *a, const vf *b, const vf *mk, const vf *S2, double D, int i, int j) {
  const int nl = o->nl; const double sq = D * D, m = W(mk, 0, i, j);
  if (nl == 1) { /* relax_barotropic qg_barotropic.h:57-76 */
    double d = -(-o->iRd2_low) * sq, v = -W(b, 0, i, j) * sq;
    v += (W(a, 0, i + 1, j) + W(a, 0, i - 1, j)) * m; d += 2.;
    v += (W(a, 0, i, j + 1) + W(a, 0, i, j - 1)) * m; d += 2.;
    W(a, 0, i, j) = v / d;
    return;
  }
  double t0[ORN_MAXNL], t1[ORN_MAXNL], t2[ORN_MAXNL], rhs[ORN_MAXNL]; /* relax_baroclinic qg_baroclinic_ms.h:228-291 */
  int l = 0;
  rhs[l] = -sq * W(b, l, i, j) * m; t2[l] = -sq * W(S2, l, i, j) * o->idh1[l] * m; t1[l] = -t2[l];
  rhs[l] += (W(a, l, i + 1, j) + W(a, l, i - 1, j)) * m; t1[l] += 2;
  rhs[l] += (W(a, l, i, j + 1) + W(a, l, i, j - 1)) * m; t1[l] += 2;
  for (l = 1; l < nl - 1; l++) {
    rhs[l] = -sq * W(b, l, i, j) * m; t0[l] = -sq * W(S2, l - 1, i, j) * o->idh0[l] * m; t2[l] = -sq * W(S2, l, i, j) * o->idh1[l] * m; t1[l] = -t0[l] - t2[l];
    rhs[l] += (W(a, l, i + 1, j) + W(a, l, i - 1, j)) * m; t1[l] += 2;
    rhs[l] += (W(a, l, i, j + 1) + W(a, l, i, j - 1)) * m; t1[l] += 2;
  }
  l = nl - 1;
  rhs[l] = -sq * W(b, l, i, j) * m; t0[l] = -sq * W(S2, l - 1, i, j) * o->idh0[l]; /* not masked, :267 */ t1[l] = -t0[l];
  rhs[l] += (W(a, l, i + 1, j) + W(a, l, i - 1, j)) * m; t1[l] += 2;
  rhs[l] += (W(a, l, i, j + 1) + W(a, l, i, j - 1)) * m; t1[l] += 2;
  for (l = 1; l < nl; l++) { t1[l] -= t0[l] * t2[l - 1] / t1[l - 1]; rhs[l] -= t0[l] * rhs[l - 1] / t1[l - 1]; }
  W(a, nl - 1, i, j) = t0[nl - 1] = rhs[nl - 1] / t1[nl - 1];
  for (l = nl - 2; l >= 0; l--) W(a, l, i, j) = t0[l] = (rhs[l] - t2[l] * t0[l + 1]) / t1[l];
}
/* one sweep over ALL vertices of level k (foreach_vertex_level, inner-vertex.h:107-138) followed by
 * boundary_level(a): the correction's BC is the homogeneous psi BC (0 on the boundary vertices).
 * smoother 0: x outer / y inner, in place (reference); 1: red ((i+j) even) then black. */
static void relax_level(orn_t *o, int k) {
  vf *a = &o->da[k]; const int n = a->n; const double D = o->L0 / n;
  if (o->smoother == 0) { for (int i = 0; i <= n; i++) for (int j = 0; j <= n; j++) relax_col(o, a, &o->res[k], &o->mask[k], &o->S2[k], D, i, j); }
  else for (int c = 0; c < 2; c++) { for (int j = 0; j <= n; j++) for (int i = (j + c) & 1; i <= n; i += 2) relax_col(o, a, &o->res[k], &o->mask[k], &o->S2[k], D, i, j); bnd_const(a, 0.); }
  bnd_const(a, 0.);
}
/* residual_baroclinic :295-341 / residual_barotropic qg_barotropic.h:78-97 */
static double residual(orn_t *o, vf *a, vf *b, vf *res) {
  const int n = o->N, nl = o->nl; const double D = o->L0 / n, sq = D * D; vf *mk = &o->mask[0], *S2 = &o->S2[0]; double maxres = 0.;
  for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) {
    const double m = W(mk, 0, i, j);
    for (int l = 0; l < nl; l++) {
      double r;
      if (nl == 1) r = (W(b, l, i, j) - (-o->iRd2_low * W(a, l, i, j))) * m;
      else if (l == 0) r = (W(b, l, i, j) + W(S2, l, i, j) * (W(a, l, i, j) - W(a, l + 1, i, j)) * o->idh1[l]) * m;
      else if (l < nl - 1) r = (W(b, l, i, j) + W(S2, l - 1, i, j) * (W(a, l, i, j) - W(a, l - 1, i, j)) * o->idh0[l] - W(S2, l, i, j) * (W(a, l + 1, i, j) - W(a, l, i, j)) * o->idh1[l]) * m;
      else r = (W(b, l, i, j) + W(S2, l - 1, i, j) * (W(a, l, i, j) - W(a, l - 1, i, j)) * o->idh0[l]) * m;
      r -= (W(a, l, i - 1, j) - 2. * W(a, l, i, j) + W(a, l, i + 1, j)) / sq * m;
      r -= (W(a, l, i, j - 1) - 2. * W(a, l, i, j) + W(a, l, i, j + 1)) / sq * m;
      W(res, l, i, j) = r;
      if (fabs(r) > maxres) maxres = fabs(r);
    }
  }
  return maxres;
}
/* restriction_coarsen_vert my_vertex.h:55-62 */
static void restrict_res(const vf *f, vf *c) {
  for (int l = 0; l < c->nl; l++) for (int J = 0; J <= c->n; J++) for (int I = 0; I <= c->n; I++)
    W(c, l, I, J) = (W(f, l, 2 * I + 1, 2 * J) + 2 * W(f, l, 2 * I, 2 * J) + W(f, l, 2 * I - 1, 2 * J) + W(f, l, 2 * I, 2 * J + 1) + W(f, l, 2 * I, 2 * J - 1)) / 6.;
}
/* restriction_coarsen_vert2 my_vertex.h:65-75 (mask), restriction_vert :49-51 (S2: injection) */
static void restrict_mask(const vf *f, vf *c) {
  for (int J = 0; J <= c->n; J++) for (int I = 0; I <= c->n; I++)
    W(c, 0, I, J) = (4 * W(f, 0, 2 * I, 2 * J) + 2 * W(f, 0, 2 * I + 1, 2 * J) + 2 * W(f, 0, 2 * I - 1, 2 * J) + 2 * W(f, 0, 2 * I, 2 * J + 1) + 2 * W(f, 0, 2 * I, 2 * J - 1) +
                     W(f, 0, 2 * I + 1, 2 * J + 1) + W(f, 0, 2 * I - 1, 2 * J + 1) + W(f, 0, 2 * I + 1, 2 * J - 1) + W(f, 0, 2 * I - 1, 2 * J - 1)) / 16.;
}
static void inject(const vf *f, vf *c) { for (int l = 0; l < c->nl; l++) for (int J = 0; J <= c->n; J++) for (int I = 0; I <= c->n; I++) W(c, l, I, J) = W(f, l, 2 * I, 2 * J); }
/* refine_vert my_vertex.h:82-105, for every coarse vertex */
static void prolong(const vf *c, vf *f) {
  for (int l = 0; l < c->nl; l++) for (int J = 0; J <= c->n; J++) for (int I = 0; I <= c->n; I++) {
    W(f, l, 2 * I, 2 * J) = W(c, l, I, J);
    W(f, l, 2 * I + 1, 2 * J) = (W(c, l, I, J) + W(c, l, I + 1, J)) / 2.;
    W(f, l, 2 * I, 2 * J + 1) = (W(c, l, I, J) + W(c, l, I, J + 1)) / 2.;
    W(f, l, 2 * I + 1, 2 * J + 1) = (W(c, l, I, J) + W(c, l, I + 1, J) + W(c, l, I, J + 1) + W(c, l, I + 1, J + 1)) / 4.;
  }
}
/* set_const qg-node/qg.h:467-472 (mask on all levels) + init qg_baroclinic_ms.h:501-510 (S2) */
static void build_levels(orn_t *o) {
  const size_t sz = (size_t)(o->N + 3) * (o->N + 3) * sizeof(double);
  bnd_const(&o->f[ORN_MASK], 0.);
  memcpy(o->mask[0].d, o->f[ORN_MASK].d, sz);
  memcpy(o->S2[0].d, o->f[ORN_S2].d, sz * o->f[ORN_S2].nl);
  for (int k = 1; k < o->nlev; k++) { restrict_mask(&o->mask[k - 1], &o->mask[k]); bnd_const(&o->mask[k], 0.); inject(&o->S2[k - 1], &o->S2[k]); }
}
/* vpoisson nodal-poisson.h:19-143 */
static orn_mgstats vpoisson(orn_t *o, vf *a, vf *b) {
  orn_mgstats mg; mg.sum = HUGE_VAL; mg.resa = HUGE_VAL; mg.resb = 0; mg.nrelax = o->nrelax;
  for (mg.i = 0; mg.i < o->nitermax; mg.i++) {
    const double max = residual(o, a, b, &o->res[0]);
    mg.resa = max;
    if (mg.i == 0) mg.resb = max;
    if (max < o->TOLERANCE && mg.i >= o->nitermin) break;
    bnd_const(&o->res[0], 0.);
    for (int k = 1; k < o->nlev; k++) restrict_res(&o->res[k - 1], &o->res[k]);
    for (int k = 0; k < o->nlev; k++) bnd_const(&o->res[k], 0.);
    vf_zero(&o->da[o->nlev - 1]);
    for (int k = o->nlev - 1; k >= 0; k--) {
      bnd_const(&o->da[k], 0.);
      for (int r = 0; r < mg.nrelax; r++) relax_level(o, k);
      if (k > 0) { prolong(&o->da[k], &o->da[k - 1]); bnd_const(&o->da[k - 1], 0.); }
    }
    for (int l = 0; l < a->nl; l++) for (int j = 0; j <= a->n; j++) for (int i = 0; i <= a->n; i++) W(a, l, i, j) += W(&o->da[0], l, i, j);
    bnd_const(a, o->psi_bc);
  }
  if (mg.resa > o->TOLERANCE && !o->quiet) fprintf(stderr, "Convergence for psi not reached.\nmg.i = %d, mg.resb: %g mg.resa: %g\n", mg.i, mg.resb, mg.resa);
  return mg;
}
/* invert_q_baroclinic :217-225 / invert_q_barotropic qg_barotropic.h:45-54 */
static void invert_q(orn_t *o, vf *q) {
  if (o->sqg) {
    /* the elliptic operator of relax_ / residual_baroclinic (sqg_baroclinic_ms.h:248-400) is the baroclinic one with the
     * shifted S2 index and has no bs term; the reference passes q itself (:250, unfinished file).  Completed so that
     * invert_q inverts comp_q: the known surface term S2S*bs*idh0[0] of the top layer moves to the right-hand side */
    const int n = o->N; vf *e = &o->f[ORN_QEFF];
    memcpy(e->d, q->d, (size_t)q->nl * (n + 3) * (n + 3) * sizeof(double));
    for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) W(e, 0, i, j) = W(q, 0, i, j) - W(&o->f[ORN_S2S], 0, i, j) * W(&o->f[ORN_BS], 0, i, j) * o->idh0[0];
    o->mg = vpoisson(o, &o->f[ORN_PSI], e);
    bnd_psi(o);
    bnd_q(o, q);
    return;
  }
  o->mg = vpoisson(o, &o->f[ORN_PSI], q);
  bnd_psi(o);
  bnd_q(o, q);
}

/* ---------------------------------------------------------------- stochastic forcing (cell scalars) */
#define CI(n, i, j) ((size_t)((j) + 1) * ((n) + 2) + (size_t)((i) + 1))
/* [BASILISK RULE] default BC of a cell scalar: ghost = interior (x direction first, then y over the x-ghosts) */
static void cell_bc(double *f, int n) {
  for (int j = 0; j < n; j++) { f[CI(n, n, j)] = f[CI(n, n - 1, j)]; f[CI(n, -1, j)] = f[CI(n, 0, j)]; }
  for (int i = -1; i <= n; i++) { f[CI(n, i, n)] = f[CI(n, i, n - 1)]; f[CI(n, i, -1)] = f[CI(n, i, 0)]; }
}
/* event init_stoch qg-node/qg_stochastic.h:15-47: wavelet coefficients of the uniform filter length L_filt */
static void init_stoch(orn_t *o) {
  const int K = o->cnlev;
  for (int k = 0; k < K; k++) {
    const int n = o->N >> k; const double Delta = o->L0 / n;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
      double ref_flag = 0;
      if (k > 0) { const int m = n * 2; const double *c = o->csig[k - 1];
        ref_flag += c[CI(m, 2 * i, 2 * j)]; ref_flag += c[CI(m, 2 * i, 2 * j + 1)]; ref_flag += c[CI(m, 2 * i + 1, 2 * j)]; ref_flag += c[CI(m, 2 * i + 1, 2 * j + 1)]; }
      double v;
      if (ref_flag > 0) v = 1;
      else if (o->L_filt > 2 * Delta) v = 0;
      else if (o->L_filt <= 2 * Delta && o->L_filt > Delta) v = 1 - (o->L_filt - Delta) / Delta;
      else v = 1;
      o->csig[k][CI(n, i, j)] = v;
    }
    cell_bc(o->csig[k], n);
  }
  for (int k = 0; k < K; k++) { const int n = o->N >> k;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) o->csig[k][CI(n, i, j)] = 1 - o->csig[k][CI(n, i, j)];
    cell_bc(o->csig[k], n); }
}
static double normal_noise(void) { /* qg_stochastic.h:13 */
  double a = sqrt(-2. * log(((double)(rand()) + 1.) / ((double)(RAND_MAX) + 2.)));
  return a * cos(2 * M_PI * rand() / (double)RAND_MAX);
}
static double cell_bilinear(const double *c, int n, int i, int j) { /* fine cell (i, j) from the level with n cells */
  const int I = i >> 1, J = j >> 1, cx = (i & 1) ? 1 : -1, cy = (j & 1) ? 1 : -1;
  return (9. * c[CI(n, I, J)] + 3. * (c[CI(n, I + cx, J)] + c[CI(n, I, J + cy)]) + c[CI(n, I + cx, J + cy)]) / 16.;
}
/* wavelet -> scale by sig_lev -> inverse_wavelet of the cell field cs[0] ([BASILISK RULE], as in qg_oracle.c) */
static void cell_wavelet_filter(orn_t *o) {
  const int K = o->cnlev, N = o->N;
  cell_bc(o->cs[0], N);
  for (int k = 1; k < K; k++) { const int n = N >> k, m = n * 2; const double *f = o->cs[k - 1];
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
      double sum = 0.; sum += f[CI(m, 2 * i, 2 * j)]; sum += f[CI(m, 2 * i, 2 * j + 1)]; sum += f[CI(m, 2 * i + 1, 2 * j)]; sum += f[CI(m, 2 * i + 1, 2 * j + 1)];
      o->cs[k][CI(n, i, j)] = sum / 4; }
    cell_bc(o->cs[k], n); }
  for (int k = 0; k < K - 1; k++) { const int n = N >> k;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
      double d = o->cs[k][CI(n, i, j)]; d -= cell_bilinear(o->cs[k + 1], n >> 1, i, j);
      o->cw[k][CI(n, i, j)] = d * o->csig[k][CI(n, i, j)]; } }
  { const int n = N >> (K - 1);
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) o->cs[K - 1][CI(n, i, j)] = o->cs[K - 1][CI(n, i, j)] * o->csig[K - 1][CI(n, i, j)];
    cell_bc(o->cs[K - 1], n); }
  for (int k = K - 2; k >= 0; k--) { const int n = N >> k;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { double r = cell_bilinear(o->cs[k + 1], n >> 1, i, j); r += o->cw[k][CI(n, i, j)]; o->cw[k][CI(n, i, j)] = r; }
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) o->cs[k][CI(n, i, j)] = o->cw[k][CI(n, i, j)];
    cell_bc(o->cs[k], n); }
}
/* generate_noise qg_stochastic.h:49-65: serial rand() stream in foreach order (x outer, y inner) */
static void generate_noise(orn_t *o) {
  const int N = o->N;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) o->cs[0][CI(N, i, j)] = o->amp_stoch * normal_noise();
  cell_wavelet_filter(o);
}
void orn_get_noise(orn_t *o, double *a) { const int N = o->N; for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) a[(size_t)j * N + i] = o->cs[0][CI(N, i, j)]; }
void orn_set_noise(orn_t *o, const double *a) { const int N = o->N; for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) o->cs[0][CI(N, i, j)] = a[(size_t)j * N + i]; cell_bc(o->cs[0], N); }
void orn_filter_noise(orn_t *o) { cell_wavelet_filter(o); }
void orn_get_csig(orn_t *o, int k, double *a) { const int n = o->N >> k; for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) a[(size_t)j * n + i] = o->csig[k][CI(n, i, j)]; }
int orn_cell_levels(orn_t *o) { return o->cnlev; }

/* ---------------------------------------------------------------- wavelet filter of the vertex model
 * wavelet_filter qg_baroclinic_ms.h:346-400 ("temporary fix": the transform runs on the CELL average of psi), its
 * coefficients sig_lev and the cell mask mask_c :525-578, wavelet_mask / inverse_wavelet_mask qg-node/wavelet_vertex.h:10-46.
 * Cell pyramids: level k has (N >> k)^2 cells and one ghost ring (CI); nl layers one after the other. */
#define CL(n) ((size_t)((n) + 2) * ((n) + 2))
static void cell_bc_dirichlet(double *f, int n) { /* dirichlet(0): ghost = -interior, x sides first ([BASILISK RULE]) */
  for (int j = 0; j < n; j++) { f[CI(n, n, j)] = -f[CI(n, n - 1, j)]; f[CI(n, -1, j)] = -f[CI(n, 0, j)]; }
  for (int i = -1; i <= n; i++) { f[CI(n, i, n)] = -f[CI(n, i, n - 1)]; f[CI(n, i, -1)] = -f[CI(n, i, 0)]; }
}
static void wv_setup(orn_t *o) {
  const int K = o->cnlev, N = o->N, nl = o->nl;
  if (!o->wsig) {
    o->wsig = (double **)calloc(K, sizeof(double *)); o->wmc = (double **)calloc(K, sizeof(double *));
    o->ws = (double **)calloc(K, sizeof(double *)); o->ww = (double **)calloc(K, sizeof(double *));
    for (int k = 0; k < K; k++) { const size_t sz = CL(N >> k);
      o->wsig[k] = (double *)calloc(sz, sizeof(double)); o->wmc[k] = (double *)calloc(sz, sizeof(double));
      o->ws[k] = (double *)calloc(sz * nl, sizeof(double)); o->ww[k] = (double *)calloc(sz * nl, sizeof(double)); }
  }
  /* sig_lev :527-552, a VERTEX scalar used at the cell with the same index (foreach_vertex_level ... w[] *= sig_lev[],
   * :373-376): only the vertices (i, j) < n of a level are ever read, and their children (2i + a, 2j + b) are in range.
   * Low pass only (the high-pass flip is commented out, :554-559).  L_filt2 = min(fac_filt_Rd dh[0] / sqrt(S2[]), Lfmax)
   * with S2 of layer 0 at the vertex (injected on the coarse levels) when fac_filt_Rd > 0, else
   * L_filt = Lfmax + (y / L0)(Lfmin - Lfmax), qg_baroclinic_ms.h:52 */
  for (int k = 0; k < K; k++) {
    const int n = N >> k; const double Delta = o->L0 / n;
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
      double ref_flag = 0;
      if (k > 0) { const int m = n * 2; const double *c = o->wsig[k - 1];
        ref_flag += c[CI(m, 2 * i, 2 * j)]; ref_flag += c[CI(m, 2 * i, 2 * j + 1)]; ref_flag += c[CI(m, 2 * i + 1, 2 * j)]; ref_flag += c[CI(m, 2 * i + 1, 2 * j + 1)]; }
      double v;
      if (ref_flag > 0) v = 1;
      else {
        double L2;
        if (o->fac_filt_Rd > 0) L2 = fmin(o->fac_filt_Rd * o->dh[0] / sqrt(W(&o->f[ORN_S2], 0, i << k, j << k)), o->Lfmax);
        else L2 = o->Lfmax + (j * Delta / o->L0) * (o->Lfmin - o->Lfmax);
        if (L2 > 2 * Delta) v = 0;
        else if (L2 <= 2 * Delta && L2 > Delta) v = 1 - (L2 - Delta) / Delta;
        else v = 1;
      }
      o->wsig[k][CI(n, i, j)] = v;
    }
  }
  /* mask_c :567-575: cell average of the vertex mask, restricted (mean of the 4 children) */
  { vf *mk = &o->f[ORN_MASK];
    for (int j = 0; j < N; j++) for (int i = 0; i < N; i++)
      o->wmc[0][CI(N, i, j)] = 0.25 * (W(mk, 0, i, j) + W(mk, 0, i + 1, j) + W(mk, 0, i, j + 1) + W(mk, 0, i + 1, j + 1));
    for (int k = 1; k < K; k++) { const int n = N >> k, m = n * 2; const double *f = o->wmc[k - 1];
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        double sum = 0.; sum += f[CI(m, 2 * i, 2 * j)]; sum += f[CI(m, 2 * i, 2 * j + 1)]; sum += f[CI(m, 2 * i + 1, 2 * j)]; sum += f[CI(m, 2 * i + 1, 2 * j + 1)];
        o->wmc[k][CI(n, i, j)] = sum / 4; } } }
  o->wv_ready = 1;
}
/* psi_i <- inverse_wavelet_mask(sig_lev * wavelet_mask(psi_i)), all layers; ws[0] holds psi_i (interior) on entry */
static void wv_masked_apply(orn_t *o) {
  const int K = o->cnlev, N = o->N, nl = o->nl;
  for (int l = 0; l < nl; l++) {
    double *s0 = o->ws[0] + l * CL(N);
    cell_bc_dirichlet(s0, N);
    for (int k = 1; k < K; k++) { const int n = N >> k, m = n * 2; const double *f = o->ws[k - 1] + l * CL(m); double *c = o->ws[k] + l * CL(n);
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        double sum = 0.; sum += f[CI(m, 2 * i, 2 * j)]; sum += f[CI(m, 2 * i, 2 * j + 1)]; sum += f[CI(m, 2 * i + 1, 2 * j)]; sum += f[CI(m, 2 * i + 1, 2 * j + 1)];
        c[CI(n, i, j)] = sum / 4; }
      cell_bc_dirichlet(c, n); }
    /* w = (s - prolongation(s coarse)) * mask_c, then * sig_lev (wavelet_vertex.h:17-26, qg_baroclinic_ms.h:373-376) */
    for (int k = 0; k < K - 1; k++) { const int n = N >> k; const double *s = o->ws[k] + l * CL(n), *sc = o->ws[k + 1] + l * CL(n >> 1); double *w = o->ww[k] + l * CL(n);
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        double d = s[CI(n, i, j)]; d -= cell_bilinear(sc, n >> 1, i, j);
        d = d * o->wmc[k][CI(n, i, j)];
        w[CI(n, i, j)] = d * o->wsig[k][CI(n, i, j)]; } }
    { const int n = N >> (K - 1); double *s = o->ws[K - 1] + l * CL(n); /* root: w = s mask_c; w *= sig_lev; s = w mask_c */
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        double w = s[CI(n, i, j)] * o->wmc[K - 1][CI(n, i, j)]; w = w * o->wsig[K - 1][CI(n, i, j)];
        s[CI(n, i, j)] = w * o->wmc[K - 1][CI(n, i, j)]; }
      cell_bc_dirichlet(s, n); }
    for (int k = K - 2; k >= 0; k--) { const int n = N >> k; double *s = o->ws[k] + l * CL(n), *w = o->ww[k] + l * CL(n); const double *sc = o->ws[k + 1] + l * CL(n >> 1);
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { double r = cell_bilinear(sc, n >> 1, i, j); r += w[CI(n, i, j)]; w[CI(n, i, j)] = r * o->wmc[k][CI(n, i, j)]; }
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) s[CI(n, i, j)] = w[CI(n, i, j)];
      cell_bc_dirichlet(s, n); }
  }
}
static void invert_q(orn_t *o, vf *q);
static void comp_q(orn_t *o, vf *psi, vf *q);
void orn_wavelet_filter(orn_t *o, double dtflt) { /* qg_baroclinic_ms.h:346-400 */
  const int N = o->N, nl = o->nl; vf *psi = &o->f[ORN_PSI], *pf = &o->f[ORN_PSIF], *mk = &o->f[ORN_MASK];
  if (!o->wv_ready) wv_setup(o);
  invert_q(o, &o->f[ORN_Q]);
  for (int l = 0; l < nl; l++) { double *s = o->ws[0] + l * CL(N);
    for (int j = 0; j < N; j++) for (int i = 0; i < N; i++)
      s[CI(N, i, j)] = 0.25 * (W(psi, l, i, j) + W(psi, l, i + 1, j) + W(psi, l, i, j + 1) + W(psi, l, i + 1, j + 1)); }
  wv_masked_apply(o);
  if (o->Lfmax < 1e30) /* `if (Lfmax < HUGE)`, :380 */
    for (int l = 0; l < nl; l++) { const double *s = o->ws[0] + l * CL(N);
      for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) {
        const double psi_loc = 0.25 * (s[CI(N, i, j)] + s[CI(N, i - 1, j)] + s[CI(N, i, j - 1)] + s[CI(N, i - 1, j - 1)]);
        W(pf, l, i, j) = (W(pf, l, i, j) * o->nbar + psi_loc / dtflt) / (o->nbar + 1);
        W(psi, l, i, j) = (W(psi, l, i, j) - psi_loc) * W(mk, 0, i, j); } }
  bnd_psi(o);
  comp_q(o, psi, &o->f[ORN_Q]);
  o->nbar++;
}
void orn_wv_get(orn_t *o, int what, int k, double *a) { /* 0: sig_lev (as used at the cells), 1: mask_c; level k, [n][n] */
  if (!o->wv_ready) wv_setup(o);
  const int n = o->N >> k; const double *f = what ? o->wmc[k] : o->wsig[k];
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) a[(size_t)j * n + i] = f[CI(n, i, j)];
}
void orn_wv_apply(orn_t *o, const double *in, double *out) { /* the masked transform pair alone on a cell field [nl][N][N] */
  const int N = o->N, nl = o->nl;
  if (!o->wv_ready) wv_setup(o);
  for (int l = 0; l < nl; l++) for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) (o->ws[0] + l * CL(N))[CI(N, i, j)] = in[((size_t)l * N + j) * N + i];
  wv_masked_apply(o);
  for (int l = 0; l < nl; l++) for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) out[((size_t)l * N + j) * N + i] = (o->ws[0] + l * CL(N))[CI(N, i, j)];
}

/* ---------------------------------------------------------------- set_const, time stepping */
void orn_set_const(orn_t *o) { /* qg-node/qg.h:465-524 + qg_baroclinic_ms.h:449-510 + qg_barotropic.h:115-118 */
  const int nl = o->nl, N = o->N; const double D = o->L0 / N;
  if (nl > 1) {
    double dhc[ORN_MAXNL];
    for (int l = 0; l < nl - 1; l++) dhc[l] = 0.5 * (o->dh[l] + o->dh[l + 1]);
    o->idh0[0] = o->sqg ? 1. / o->dh[0] : 0.; /* sqg_baroclinic_ms.h:502 "surface layer: 1/h" */
    o->idh1[0] = 1. / (dhc[0] * o->dh[0]);
    for (int l = 1; l < nl - 1; l++) { o->idh0[l] = 1. / (dhc[l - 1] * o->dh[l]); o->idh1[l] = 1. / (dhc[l] * o->dh[l]); }
    o->idh0[nl - 1] = 1. / (dhc[nl - 2] * o->dh[nl - 1]); o->idh1[nl - 1] = 0.;
    /* S2: N^2 -> f^2/N^2 with f = f0 + flag_ms*beta*(y - L0/2), :501-505 (input: the field holds N^2) */
    for (int l = 0; l < nl - 1; l++) for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) {
      const double f = o->f0 + o->flag_ms * o->beta * (j * D - 0.5 * o->L0);
      W(&o->f[ORN_S2], l, i, j) = f * f / W(&o->f[ORN_S2], l, i, j);
    }
    if (o->sqg) for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) { /* :545 surface layer f/N^2 (f, not f^2) */
      const double f = o->f0 + o->flag_ms * o->beta * (j * D - 0.5 * o->L0);
      W(&o->f[ORN_S2S], 0, i, j) = f / W(&o->f[ORN_S2S], 0, i, j);
    }
    for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) W(&o->f[ORN_TOPO], 0, i, j) *= o->scale_topo;
  } else if (o->gp_low != 0.) o->iRd2_low = o->f0 * o->f0 / (o->gp_low * o->dh[nl - 1]);
  build_levels(o);
  init_stoch(o);
  o->wv_ready = 0;
  bnd_psi(o);
  if (o->nu != 0) o->DT = 0.5 * fmin(o->DT, D * D / o->nu / 4.);          /* qg-node/qg.h:511-512 */
  if (o->beta != 0) o->DT = fmin(o->DT, 1 / (2. * o->beta * o->L0));
  comp_q(o, &o->f[ORN_PSI], &o->f[ORN_Q]);
}
/* adjust_dt qg-node/qg.h:258-284: u = (psi[0,1] - psi[])/D on x-faces, (psi[1,0] - psi[])/D on y-faces */
static double adjust_dt(orn_t *o, double dtmax) {
  const int n = o->N; const double D = o->L0 / n; vf *psi = &o->f[ORN_PSI];
  dtmax /= o->CFL;
  for (int l = 0; l < o->nl; l++) for (int j = 0; j <= n; j++) for (int i = 0; i <= n; i++) {
    if (j < n) { double u = (W(psi, l, i, j + 1) - W(psi, l, i, j)) / D; if (u != 0.) { double dt = D / fabs(u); if (dt < dtmax) dtmax = dt; } }
    if (i < n) { double u = (W(psi, l, i + 1, j) - W(psi, l, i, j)) / D; if (u != 0.) { double dt = D / fabs(u); if (dt < dtmax) dtmax = dt; } }
  }
  dtmax *= o->CFL;
  if (dtmax > o->previous) dtmax = (o->previous + 0.1 * dtmax) / 1.1;
  o->previous = dtmax;
  return dtmax;
}
double orn_update(orn_t *o, int qf, int dqf, double dtmax) { /* update_qg qg-node/qg.h:334-354 */
  invert_q(o, &o->f[qf]);
  rhs_pv(o, &o->f[qf], &o->f[dqf]);
  return adjust_dt(o, dtmax);
}
void orn_advance(orn_t *o, int out, int in, int dq, double dt) { /* advance_qg qg-node/qg.h:291-302 */
  vf *a = &o->f[out], *b = &o->f[in], *d = &o->f[dq];
  for (int l = 0; l < a->nl; l++) for (int j = 0; j <= a->n; j++) for (int i = 0; i <= a->n; i++) W(a, l, i, j) = W(b, l, i, j) + W(d, l, i, j) * dt;
  if (o->stochastic) { /* qg-node/qg.h:306-320; the vertex (i, j) takes the value of the cell (i, j) (ghost cells at i, j = N) */
    const int N = o->N;
    o->corrector_step = (o->corrector_step + 1) % 2;
    double dts = sqrt(dt);
    if (o->corrector_step) { generate_noise(o); dts = dts / sqrt(2); }
    for (int j = 0; j <= N; j++) for (int i = 0; i <= N; i++) W(a, 0, i, j) += o->cs[0][CI(N, i, j)] * dts;
  }
}
/* event forcing (i++), qg-node/qg.c:136-145 */
void orn_forcing(orn_t *o) {
  const int n = o->N; const double D = o->L0 / n, L0 = o->L0, t = o->t;
  for (int j = 0; j <= n; j++) { const double y = j * D;
    const double v = -(o->tau0 + o->tau1 * cos(2 * M_PI * t / o->tf1)) / o->dh[0] * o->forc_mode * M_PI / L0 *
                     sin(o->forc_mode * M_PI * (y + y * (y - L0) * 2 / (L0 * L0) * o->dy_ws * sin(2 * M_PI * t / o->tf2)) / L0);
    for (int i = 0; i <= n; i++) W(&o->f[ORN_QFORC], 0, i, j) = v; }
}
static double dtnext(orn_t *o, double dt, double *tn) { /* [BASILISK RULE] */
  double tnext = o->tnext_event, t = o->t;
  if (tnext != HUGE_VAL && tnext > t) {
    unsigned int n = (unsigned int)((tnext - t) / dt);
    if (n == 0) dt = tnext - t;
    else { double dt1 = (tnext - t) / n; if (dt1 > dt * (1. + 1e-9)) dt = (tnext - t) / (n + 1); else if (dt1 < dt) dt = dt1; tnext = t + dt; }
  } else tnext = t + dt;
  *tn = tnext; return dt;
}
int orn_step(orn_t *o, int with_forcing_event) { /* events, then one iteration of run() */
  double tn;
  if (with_forcing_event) orn_forcing(o);
  o->dt = dtnext(o, orn_update(o, ORN_Q, ORN_DQ, o->DT), &tn);
  orn_advance(o, ORN_QPRED, ORN_Q, ORN_DQ, o->dt / 2.);
  orn_update(o, ORN_QPRED, ORN_DQ, o->dt);
  orn_advance(o, ORN_Q, ORN_Q, ORN_DQ, o->dt);
  o->t = tn; o->iter++;
  return 0;
}
double orn_ke(orn_t *o) { /* qg-node/qg.c:172-178 */
  const int n = o->N; const double D = o->L0 / n, D2 = D * D; double ke = 0; vf *psi = &o->f[ORN_PSI];
  for (int i = 0; i <= n; i++) for (int j = 0; j <= n; j++) ke -= 0.5 * W(psi, 0, i, j) * LAPN(psi, 0, i, j, D2) * D2;
  return ke;
}
/* event write_1d_diag qg-node/qg.h:361-399: sums over the CELL loop (vertices i, j = 0..N-1), top layer */
void orn_diag1d(orn_t *o, double *out) {
  const int n = o->N; const double D = o->L0 / n, D2 = D * D; vf *psi = &o->f[ORN_PSI], *q = &o->f[ORN_Q], *qf = &o->f[ORN_QFORC];
  double ke = 0, d_ke = 0, f_ke = 0;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
    ke -= 0.5 * W(psi, 0, i, j) * LAPN(psi, 0, i, j, D2) * D2;
    d_ke -= o->nu * W(psi, 0, i, j) * LAPN(q, 0, i, j, D2) * D2;
    f_ke -= W(psi, 0, i, j) * W(qf, 0, i, j) * D2;
  }
  out[0] = ke; out[1] = d_ke; out[2] = f_ke;
}
double orn_time(orn_t *o) { return o->t; }
double orn_dt(orn_t *o) { return o->dt; }
void orn_set_tnext(orn_t *o, double t) { o->tnext_event = t; }
orn_mgstats orn_last_mgstats(orn_t *o) { return o->mg; }
orn_mgstats orn_invert_q(orn_t *o, int qf) { invert_q(o, &o->f[qf]); return o->mg; }
void orn_comp_q(orn_t *o, int psif, int qf) { comp_q(o, &o->f[psif], &o->f[qf]); }
void orn_rhs_pv(orn_t *o, int qf, int dqf) { rhs_pv(o, &o->f[qf], &o->f[dqf]); }
void orn_comp_del2_zeta(orn_t *o) { comp_del2(o, &o->f[ORN_PSI], &o->f[ORN_ZETA], 0., 1., OUT_ZETA); }

/* raw multigrid pieces on level arrays [layer][j][i] of (n_k + 1)^2 vertices */
static void lv_from(vf *f, const double *a) { const int n1 = f->n + 1; for (int l = 0; l < f->nl; l++) for (int j = 0; j < n1; j++) for (int i = 0; i < n1; i++) W(f, l, i, j) = a[((size_t)l * n1 + j) * n1 + i]; }
static void lv_to(const vf *f, double *a) { const int n1 = f->n + 1; for (int l = 0; l < f->nl; l++) for (int j = 0; j < n1; j++) for (int i = 0; i < n1; i++) a[((size_t)l * n1 + j) * n1 + i] = W(f, l, i, j); }
void orn_relax_raw(orn_t *o, int k, double *da, const double *res, int nsweeps) {
  lv_from(&o->da[k], da); lv_from(&o->res[k], res); bnd_const(&o->da[k], 0.);
  for (int r = 0; r < nsweeps; r++) relax_level(o, k);
  lv_to(&o->da[k], da);
}
double orn_residual_raw(orn_t *o, const double *a, const double *b, double *res) {
  vf fa, fb; vf_alloc(&fa, o->N, o->nl); vf_alloc(&fb, o->N, o->nl); lv_from(&fa, a); lv_from(&fb, b);
  double m = residual(o, &fa, &fb, &o->res[0]); lv_to(&o->res[0], res); free(fa.d); free(fb.d); return m;
}
void orn_restrict_raw(orn_t *o, int k, const double *fine, double *coarse) { lv_from(&o->res[k], fine); bnd_const(&o->res[k], 0.); restrict_res(&o->res[k], &o->res[k + 1]); bnd_const(&o->res[k + 1], 0.); lv_to(&o->res[k + 1], coarse); }
void orn_prolong_raw(orn_t *o, int k, const double *coarse, double *fine) { lv_from(&o->da[k], coarse); bnd_const(&o->da[k], 0.); prolong(&o->da[k], &o->da[k - 1]); bnd_const(&o->da[k - 1], 0.); lv_to(&o->da[k - 1], fine); }
void orn_get_level_mask(orn_t *o, int k, double *a) { lv_to(&o->mask[k], a); }
