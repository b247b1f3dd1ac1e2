/*
 * qg_oracle.h -- CPU restatement ("oracle") of the msom multi-layer QG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under msom_amd/ (the product) may include,
 * link, import or execute anything in oracle/.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the reference (bderembl/msom, Basilisk-C) cannot be compiled
 * here (needs Basilisk's qcc + runtime, not vendored, no pinned version) and ships
 * no golden vectors, tests or fixtures for this path.  This restatement follows the
 * reference source text line by line (citations `file:line` are relative to
 * /root/reference) and is pinned only by analytic known-answer tests
 * (tests/test_oracle_kat.py).  Basilisk runtime rules that are not in the tree are
 * isolated in single functions marked [BASILISK RULE].
 */
#ifndef QG_ORACLE_H
#define QG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXNL 64

/* field ids for orc_set_field / orc_get_field (arrays are [layer][y][x], interior only) */
enum {
  ORC_PSI = 0, ORC_Q = 1, ORC_ZETA = 2, ORC_PSIPG = 3, ORC_ZETAPG = 4, ORC_QFORC = 5,
  ORC_TMP = 6, ORC_FR = 7 /* nl-1 layers */, ORC_S = 8 /* nl-1 layers */, ORC_DQ = 9,
  ORC_RO = 10 /* 1 layer */, ORC_TOPO = 11 /* 1 layer */, ORC_QPRED = 12,
  ORC_NOISE = 13, ORC_SIGMA = 14,
  /* passive tracers, nl*nptr layers, index l*nptr + nt (msqg/qg.h:100-101,574-588) */
  ORC_PTR = 15, ORC_PTR_RELAX = 16, ORC_DPTR = 17, ORC_PTR_PRED = 18,
  ORC_RD = 19 /* Rd, 1 layer (msqg/qg.h:47,913) */, ORC_QOF = 20 /* qofl: filter mean (qg.h:27) */,
  /* energy diagnostics msqg/qg_energy.h:7-15, nl layers each */
  ORC_DE_BF = 21, ORC_DE_VD = 22, ORC_DE_J1 = 23, ORC_DE_J2 = 24, ORC_DE_J3 = 25, ORC_DE_FT = 26, ORC_TMP2 = 27, ORC_PO_MFT = 28,
  ORC_NFIELDS = 29
};

enum { ORC_GS_LEX = 0, ORC_GS_RB = 1 };

typedef struct {
  int i;               /* number of cycles done                 mspg/elliptic.h:118-123 */
  double resb, resa;   /* max residual before / after                                  */
  double sum;          /* sum of rhs                                                   */
  int nrelax;          /* final number of relaxations per level                        */
} orc_mgstats;

typedef struct orc orc_t;

/* lifecycle: read_params -> set_vars -> (user sets fields) -> set_const */
orc_t *orc_create_str(const char *params_text);   /* msqg/qg.h:689-761 + set_vars :837-925 */
void   orc_destroy(orc_t *o);
int    orc_set_option(orc_t *o, const char *key, double v);  /* smoother, TOLERANCE, NITERMAX, ... */
double orc_get_param(orc_t *o, const char *key);
void   orc_set_const(orc_t *o);                   /* msqg/qg.h:931-1116 (without file discovery) */

int  orc_nlayers_of(orc_t *o, int field);
void orc_set_field(orc_t *o, int field, const double *a);  /* then BC fill, cf. pyset_field :1164 */
void orc_get_field(orc_t *o, int field, double *a);
void orc_remove_mean(orc_t *o, int field);        /* msqg/qg.c:65-70 */

/* operators on the internal fields (ids above) */
void   orc_comp_del2(orc_t *o, int in, int out, double add, double fac);     /* qg.h:172-200 */
void   orc_comp_stretch(orc_t *o, int in, int out, double add, double fac);  /* qg.h:203-246 */
void   orc_comp_q(orc_t *o, int psi, int q);                                 /* qg.h:397-403 */
double orc_advection_pv(orc_t *o, int zeta, int q, int psi, int dq, double dtmax); /* qg.h:288-393 */
void   orc_dissip(orc_t *o, int zeta, int dq);                               /* qg.h:407-422 */
void   orc_forcing_terms(orc_t *o, int zeta, int psi, int dq);               /* qg.h:429-488, 626-630 */
orc_mgstats orc_invertq(orc_t *o, int psi, int q);                           /* qg.h:114-163 */
double orc_update(orc_t *o, int q, int dq, double dtmax);                    /* qg.h:609-650 */
void   orc_advance(orc_t *o, int out, int in, int dq, double dt);            /* qg.h:594-606 */
double orc_ke(orc_t *o);                                                     /* qg.c:101-109 */
double orc_timestep_limiter(orc_t *o, double dtmin_faces, double dtmax);     /* newqg/qg.h:202-219 */
void   orc_reset_limiter(orc_t *o);

/* multigrid pieces on raw arrays (level 0 = finest); arrays [layer][y][x] of that level */
int    orc_nlevels(orc_t *o);
void   orc_level_dims(orc_t *o, int lev, int *nx, int *ny);
void   orc_relax_raw(orc_t *o, int lev, double *da, const double *res, int nsweeps); /* poisson_layer.h:48-150 */
double orc_residual_raw(orc_t *o, const double *a, const double *b, double *res);  /* poisson_layer.h:157-258 */
void   orc_restrict_raw(orc_t *o, int lev_fine, const double *fine, double *coarse, int nlay); /* [BASILISK RULE] */
void   orc_prolong_raw(orc_t *o, int lev_coarse, const double *coarse, double *fine); /* [BASILISK RULE] */

/* time loop: Basilisk predictor-corrector run() [BASILISK RULE] */
int    orc_step(orc_t *o);            /* one RK2 step incl. dtnext; returns 0 */
double orc_time(orc_t *o);
double orc_dt(orc_t *o);
int    orc_iter(orc_t *o);
orc_mgstats orc_last_mgstats(orc_t *o);
void   orc_set_tnext(orc_t *o, double tnext); /* time of next t-scheduled event (HUGE if none) */

/* pystep_bfn / pyq2p / pyp2q  msqg/qg_bfn.h:21-103 */
void orc_pystep_bfn(orc_t *o, const double *q_in, double *tend, double direction, int vartype);
void orc_pyq2p(orc_t *o, double *psi_out, const double *q_in);
void orc_pyp2q(orc_t *o, const double *psi_in, double *q_out);

/* wavelet scale filter msqg/qg.h:509-560 (filter coefficients sig_lev: qg.h:1059-1090); dtflt < 0: q restored */
void orc_wavelet_filter(orc_t *o, double dtflt);
int  orc_wavelet_levels(orc_t *o);                       /* depth() + 1; level 0 = finest here */
void orc_get_siglev(orc_t *o, int lev, double *a);       /* [ny>>lev][nx>>lev] */
void orc_wavelet_apply(orc_t *o, int field);             /* field <- inverse_wavelet(sig_lev * wavelet(field)) */

/* energy / PV budgets msqg/qg_energy.h (ediag: 0 = terms x (-psi), 1 = PV terms) */
void orc_energy_tend(orc_t *o, double dt);               /* :227-241, event comp_diag :289-291 */
void orc_filter_de(orc_t *o, int pm_field, double dtflt);/* :208-225 */
void orc_reset_de(orc_t *o);                             /* reset_layer_var of the six budgets, qg.c:153-159 */
void orc_pystep_de(orc_t *o, const double *po, double *bf, double *vd, double *j1, double *j2, double *j3, double *ft, int onlyKE); /* :296-349 */

/* .bas IO  msqg/auxiliar_input.h:24-59,101-149 (square grids only) */
int orc_write_bas(orc_t *o, int field, const char *path);
int orc_read_bas(orc_t *o, int field, const char *path);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
