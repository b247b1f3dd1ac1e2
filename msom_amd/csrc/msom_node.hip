// msom_node.hip -- host side of the vertex-grid QG variant (C ABI msomn_*, include/msom.h).
//
// Restates the control flow of qg-node/qg.h (update_qg :334-354, advance_qg :291-302, adjust_dt
// :258-284, set_vars :404-450, set_const :465-524), qg_baroclinic_ms.h (rhs_pv_baroclinic :104-196,
// comp_q :199-211, invert_q :217-225, init :449-510), qg_barotropic.h and the nodal multigrid
// vpoisson (nodal-poisson.h:19-143) on top of the kernels of kernels_node.hip.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <string>
#include <vector>

#include "kernels.h"

#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      msom_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e__), __FILE__, __LINE__, #x); \
      return MSOM_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

struct NLevel {
  int n;        // cells per side: (n + 1)^2 vertices
  double D;     // L0 / n
  NatGeom g;
  double *da, *res, *mask, *S2;  // level 0: mask and S2 alias the model fields
  double *da2;                   // second correction buffer (the tiled smoother works out of place)
  // wide levels (option node_split): da and res in the x-parity split layout of geometry ga, with split copies of mask and S2
  int sp = 0;
  NatGeom ga;                    // geometry of da / res: g, or the split geometry
  double *mask_s = nullptr, *S2_s = nullptr;
  double *S2row_buf = nullptr;   // allocation behind S2row
  double *S2row = nullptr;       // [nlm][n + 1] when S2 does not depend on x (and the option s2_rows is on), else null
};
enum { NSC_RES = 0, NSC_UMAX = 1, NSC_KE = 2, NSC_DIAG = 3 /* 3 slots */, NSC_COUNT = 8 };

// option profile: HIP-event pairs around the finest-level launches of the vertex model (bench.py's per-kernel entries)
struct NProf {
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  double total_ms = 0;
  long launches = 0;
};
enum { NP_RELAX, NP_RELAX_PROLONG, NP_RESIDUAL, NP_CORRECT, NP_RHS, NP_COARSE, NP_MARCH, NP_CORR_RES, NP_COUNT };
static const char *const NP_NAMES[NP_COUNT] = {"relax_fine", "relax_prolong_fine", "residual", "correct", "rhs", "coarse", "march_fine", "correct_residual"};

struct msomn {
  NodeParams p;
  int profile = 0;
  NProf prof[NP_COUNT];
  std::string params_text;
  int N = 0, nl = 1, nlm = 1;
  double D = 0, psi_bc = 0., iRd2_low = 0.;
  double tolerance = 1e-3;
  int nitermax = 100, nitermin = 1, nrelax = 5, quiet = 0;  // nodal-poisson.h:19-23
  // stochastic forcing (-D_STOCHASTIC of the reference): cell-scalar noise n_stoch, wavelet-filtered (qg-node/qg_stochastic.h)
  int stochastic = 0, corrector_step = 0, cnlev = 0;
  int pg_set = 0, topo_set = 0;   // psi_pg / topo have been set (zero until then: their terms of the tendency are skipped by k_n_rhs_all)
  int forcing_3d = 0;  // -DFORCING_3D: switched on by setting MSOMN_QFORC3D
  int sqg = 0;         // surface-QG variant (params key sqg): sqg_baroclinic_ms.h:77-98,502,545-547
  double *qeff = nullptr, *d2bs = nullptr;  // sqg scratch: rhs of the inversion, laplacian(bs)
  // wavelet filter of the vertex model (qg_baroclinic_ms.h:346-400): cell pyramids of s, r (nl layers), sig_lev, mask_c
  std::vector<NatGeom> wg;
  std::vector<double *> ws, wr, wsig, wmc;
  int wv_ready = 0, nbar = 0;
  std::vector<NatGeom> cg;
  std::vector<double *> cs, cr, csig;  // cs[0] = n_stoch
  int mg_coarse = 32;   // the levels of <= (mg_coarse + 1)^2 vertices of a cycle in one launch (k_n_mg_coarse); 0: off
  int node_march = 0;   // levels of >= node_march vertices per side: K = 4 chained colour half-sweeps per pass (k_n_relax_march, rows
                        // software-prefetched one step ahead).  Measured at 2049^2 x 3: 27.1 vs 26.6 ms per step, 513^2 x 3: 4.2 vs 3.1 --
                        // half the bytes, but in the natural layout half of the lanes idle in every half-sweep and the vertex column
                        // solve (vertex-dependent coefficients, one reciprocal per layer) is arithmetic-bound: off
  int node_pfused = 1;   // option: prolongation folded into the first colour pass of the split levels
  int node_tile_max = 513;  // option: ... and of <= node_tile_max vertices per side (wider levels: the colour passes are no longer launch-bound)
  int node_march_tail1 = 1; // option: 9 half-sweeps as passes of 4 + 4 + a colour pass (measured 13.96 -> 13.43 ms per step) instead of 4 + 3 + 2 (0)
  int node_tile_k = 8;      // option: half-sweeps per LDS-tiled pass (2..8)
  int node_tile_s = 65;     // option: split levels of >= node_tile_s (and < node_march_s) vertices per side: LDS-tiled passes of up to 4 colour half-sweeps (0: off)
  int node_rhs_fused = 1;   // option: the baroclinic tendency in three passes (k_n_rhs_pre, k_n_del2_bnd, k_n_rhs_all) instead of twelve
  int node_corr_fused = 2;  // option: the correction a += da rides in the next cycle's residual pass (k_n_correct_residual)
  double *psi_alt = nullptr;
  hipEvent_t ev_res = nullptr;   // marks the copy of max |res| to the host inside vpoisson
  int node_march_s = 2049;  // option: split levels of >= node_march_s vertices per side chain up to 4 colour half-sweeps per pass (k_n_relax_march_s,
                            // round 3).  2049^2 x 3, 9 cycles per solve (tools/ab_node_prof.py): 3 passes of 83 us replace 9 colour launches of 33 us:
                            // 17.3 -> 16.6 ms per step; on the 1025^2 and 513^2 levels the pass loses (17.4 / 18.4): too few chunks for a
                            // marching wavefront; three steps of software prefetch instead of one and chunk heights 8 .. 24 change nothing
  int s2_xuniform = 0;   // set_const: S2 does not depend on x
  int s2_rows = 1;       // option: use row tables of S2 in the smoother and the residual when S2 does not depend on x
  int node_split = 65;   // option: levels of >= node_split vertices per side keep da / res / mask / S2 copies in the x-parity split layout (0: off)
  int tiled_relax = 0;  // option: LDS-tiled smoother passes (1-2 sweeps per pass) on the wide levels; measured 3 % faster at 4097^2 x 3, 7 % slower at 2049^2 x 3
  NatGeom g;
  double *f[MSOMN_NFIELDS] = {nullptr};
  int fl[MSOMN_NFIELDS] = {0};
  LayerCoef lc;
  int nlev = 0;
  std::vector<NLevel> lev;
  double *d_scal = nullptr, *h_scal = nullptr, *partial = nullptr, *d_row = nullptr, *h_row = nullptr;
  hipStream_t st = nullptr;
  int const_set = 0;
  double t = 0, dt = 1., tnext = HUGE_VAL, previous = 0;
  int iter = 0;
  msom_mgstats mg = {0, 0, 0, 0, 0};
};

static void nprof_begin(msomn *m, int slot) {
  if (!m->profile) return;
  NProf &ps = m->prof[slot];
  if (ps.used + 2 > ps.ev.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    ps.ev.push_back(a); ps.ev.push_back(b);
  }
  hipEventRecord(ps.ev[ps.used], m->st);
}
static void nprof_end(msomn *m, int slot) {
  if (!m->profile) return;
  NProf &ps = m->prof[slot];
  hipEventRecord(ps.ev[ps.used + 1], m->st);
  ps.used += 2;
}
static void nprof_collect(msomn *m, NProf &ps) {
  hipStreamSynchronize(m->st);
  for (size_t k = 0; k + 1 < ps.used; k += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ps.ev[k], ps.ev[k + 1]) == hipSuccess) { ps.total_ms += ms; ps.launches++; }
  }
  ps.used = 0;
}

static NatGeom node_geom(int n) {
  NatGeom g;
  g.nx = g.ny = n + 1;
  g.pitch = ((n + 1 + 15) / 16) * 16 + 2 * MSOM_XP;
  g.rows = n + 1 + 2 * MSOM_YP;
  g.ls = (size_t)g.pitch * g.rows;
  return g;
}
// x-parity split rows [even-i half | odd-i half]: n / 2 + 1 and n / 2 vertices, one pad slot at least after each half
static NatGeom node_geom_split(int n) {
  NatGeom g;
  g.nx = g.ny = n + 1;
  const int hp = ((n / 2 + 2 + 15) / 16) * 16;
  g.pitch = 2 * hp + 2 * MSOM_XP;
  g.rows = n + 1 + 2 * MSOM_YP;
  g.ls = (size_t)g.pitch * g.rows;
  return g;
}
static int dalloc(double **p, size_t n) {
  HIPCHK(hipMalloc((void **)p, n * sizeof(double)));
  HIPCHK(hipMemset(*p, 0, n * sizeof(double)));
  return MSOM_OK;
}
static int field_layers(const msomn *m, int f) {
  if (f == MSOMN_PSIF) return m->nl;
  return f == MSOMN_S2 ? m->nlm : (f == MSOMN_TOPO || f == MSOMN_QFORC || f == MSOMN_MASK || f == MSOMN_BS || f == MSOMN_S2S) ? 1 : m->nl;
}

// [layers][n+1][n+1] contiguous (host or device) <-> padded natural layout
static int upload_g(msomn *m, double *dst, const NatGeom &g, int nl, const double *a) {
  const size_t n1 = g.nx;
  for (int l = 0; l < nl; l++)
    HIPCHK(hipMemcpy2DAsync(dst + nat_idx(g, l, 0, 0), g.pitch * sizeof(double), a + (size_t)l * n1 * n1, n1 * sizeof(double), n1 * sizeof(double), n1,
                            hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
static int download_g(msomn *m, const double *src, const NatGeom &g, int nl, double *a) {
  const size_t n1 = g.nx;
  for (int l = 0; l < nl; l++)
    HIPCHK(hipMemcpy2DAsync(a + (size_t)l * n1 * n1, n1 * sizeof(double), src + nat_idx(g, l, 0, 0), g.pitch * sizeof(double), n1 * sizeof(double), n1,
                            hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}

extern "C" void msomn_destroy(msomn_t *m) {
  if (!m) return;
  for (int k = 0; k < MSOMN_NFIELDS; k++) if (m->f[k]) (void)hipFree(m->f[k]);
  for (size_t k = 0; k < m->lev.size(); k++) {
    if (m->lev[k].da) (void)hipFree(m->lev[k].da);
    if (m->lev[k].da2) (void)hipFree(m->lev[k].da2);
    if (m->lev[k].res) (void)hipFree(m->lev[k].res);
    if (m->lev[k].mask_s) (void)hipFree(m->lev[k].mask_s);
    if (m->lev[k].S2_s) (void)hipFree(m->lev[k].S2_s);
    if (m->lev[k].S2row_buf) (void)hipFree(m->lev[k].S2row_buf);
    if (k > 0 && m->lev[k].mask) (void)hipFree(m->lev[k].mask);
    if (k > 0 && m->lev[k].S2) (void)hipFree(m->lev[k].S2);
  }
  for (size_t k = 0; k < m->cs.size(); k++) {
    if (m->cs[k]) (void)hipFree(m->cs[k]);
    if (m->cr[k]) (void)hipFree(m->cr[k]);
    if (m->csig[k]) (void)hipFree(m->csig[k]);
  }
  for (auto *v : {&m->ws, &m->wr, &m->wsig, &m->wmc})
    for (double *q : *v) if (q) (void)hipFree(q);
  if (m->psi_alt) (void)hipFree(m->psi_alt);
  if (m->ev_res) (void)hipEventDestroy(m->ev_res);
  if (m->qeff) (void)hipFree(m->qeff);
  if (m->d2bs) (void)hipFree(m->d2bs);
  if (m->d_scal) (void)hipFree(m->d_scal);
  if (m->partial) (void)hipFree(m->partial);
  if (m->d_row) (void)hipFree(m->d_row);
  if (m->h_scal) (void)hipHostFree(m->h_scal);
  if (m->h_row) (void)hipHostFree(m->h_row);
  if (m->st) (void)hipStreamDestroy(m->st);
  delete m;
}

static int node_alloc(msomn *m) {
  HIPCHK(hipStreamCreate(&m->st));
  for (int k = 0; k < MSOMN_NFIELDS; k++) {
    m->fl[k] = field_layers(m, k);
    int r = dalloc(&m->f[k], m->g.ls * m->fl[k]);
    if (r) return r;
  }
  m->lev.resize(m->nlev);
  for (int k = 0; k < m->nlev; k++) {
    NLevel &L = m->lev[k];
    L.n = m->N >> k;
    L.D = m->p.L0 / L.n;
    L.g = node_geom(L.n);
    L.da = L.da2 = L.res = L.mask = L.S2 = nullptr;
    L.ga = L.g;
    int r;
    // da and res are sized for either layout (the layout is chosen in build_levels from the option node_split)
    const size_t lsa = std::max(L.g.ls, node_geom_split(L.n).ls);
    if ((r = dalloc(&L.da, lsa * m->nl)) || (r = dalloc(&L.da2, lsa * m->nl)) || (r = dalloc(&L.res, lsa * m->nl))) return r;
    if (k == 0) { L.mask = m->f[MSOMN_MASK]; L.S2 = m->f[MSOMN_S2]; }
    else if ((r = dalloc(&L.mask, L.g.ls)) || (r = dalloc(&L.S2, L.g.ls * m->nlm))) return r;
  }
  int r;
  if ((r = dalloc(&m->d_scal, NSC_COUNT))) return r;
  const int nblk = ((m->g.nx + 63) / 64) * ((m->g.ny + 3) / 4);
  // three partial arrays, each followed by the chunk sums of launch_sum_final
  if ((r = dalloc(&m->partial, 3 * ((size_t)nblk + 64))) || (r = dalloc(&m->d_row, m->N + 1))) return r;
  HIPCHK(hipHostMalloc((void **)&m->h_scal, NSC_COUNT * sizeof(double)));
  HIPCHK(hipHostMalloc((void **)&m->h_row, (m->N + 1) * sizeof(double)));
  // set_vars qg-node/qg.h:426-430: mask = 1 on every vertex, its BC 0 on the four walls; S2 = N2[l]
  // (qg_baroclinic_ms.h:471-476)
  const size_t n1 = m->N + 1;
  std::vector<double> h(n1 * n1 * (m->nlm > 1 ? m->nlm : 1));
  for (size_t j = 0; j < n1; j++) for (size_t i = 0; i < n1; i++) h[j * n1 + i] = (i == 0 || j == 0 || i == n1 - 1 || j == n1 - 1) ? 0. : 1.;
  if ((r = upload_g(m, m->f[MSOMN_MASK], m->g, 1, h.data()))) return r;
  if (m->nl > 1) {
    // sqg: N2 = [surface, interfaces ...]; S2 layers 1..nl-1 of sqg_baroclinic_ms.h are the interfaces below layers 0..nl-2
    for (int l = 0; l < m->nlm; l++) for (size_t k = 0; k < n1 * n1; k++) h[l * n1 * n1 + k] = m->p.N2[l + (m->sqg ? 1 : 0)];
    if ((r = upload_g(m, m->f[MSOMN_S2], m->g, m->nlm, h.data()))) return r;
    if (m->sqg) {
      for (size_t k = 0; k < n1 * n1; k++) h[k] = m->p.N2[0];
      if ((r = upload_g(m, m->f[MSOMN_S2S], m->g, 1, h.data())) || (r = dalloc(&m->qeff, m->g.ls * m->nl)) || (r = dalloc(&m->d2bs, m->g.ls))) return r;
    }
  }
  return MSOM_OK;
}

static msomn *node_create(const NodeParams &p, const char *text) {
  if (p.nl < 1 || p.nl > MSOM_FASTNL) { msom_set_error("nl = %d outside the supported range 1..%d", p.nl, MSOM_FASTNL); return nullptr; }
  if (p.N < 2 || (p.N & (p.N - 1))) { msom_set_error("N = %d must be a power of two >= 2", p.N); return nullptr; }
  if (p.bc_fac == -1) { msom_set_error("bc_fac = -1 (periodic vertex grid) is not supported"); return nullptr; }
  if (p.sqg && p.nl < 2) { msom_set_error("sqg = 1 needs nl >= 2 (qg-node/sqg_baroclinic_ms.h is the multi-layer model)"); return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    msom_set_error("no HIP device available: libmsomhip has no CPU fallback");
    return nullptr;
  }
  msomn *m = new msomn;
  m->p = p;
  if (text) m->params_text = text;
  m->N = p.N; m->nl = p.nl; m->nlm = p.nl > 1 ? p.nl - 1 : 1;
  m->sqg = p.sqg != 0;
  m->D = p.L0 / p.N;
  m->tolerance = p.TOLERANCE;
  m->g = node_geom(p.N);
  int n = 0;
  while ((p.N >> n) >= 2) n++;
  m->nlev = n;  // coarsest level: 2 x 2 cells, one interior vertex
  memset(&m->lc, 0, sizeof m->lc);
  if (node_alloc(m) != MSOM_OK) { msomn_destroy(m); return nullptr; }
  return m;
}
extern "C" msomn_t *msomn_create_str(const char *text) {
  if (!text) { msom_set_error("null params text"); return nullptr; }
  NodeParams p;
  msom_node_params_defaults(&p);
  msom_node_params_parse_text(&p, text);
  return node_create(p, text);
}
extern "C" msomn_t *msomn_create(const char *path) {
  NodeParams p;
  msom_node_params_defaults(&p);
  const char *pp = path ? path : "params.in";
  if (msom_node_params_parse_file(&p, pp)) return nullptr;
  std::string text;
  if (FILE *fp = fopen(pp, "rb")) {
    char buf[4096];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, k);
    fclose(fp);
  }
  return node_create(p, text.c_str());
}

static int choose_layouts(msomn *m);
extern "C" int msomn_set_option(msomn_t *m, const char *key, double v) {
  if (!m || !key) return MSOM_ERR_ARG;
  if (!strcmp(key, "TOLERANCE")) m->tolerance = v;
  else if (!strcmp(key, "NITERMAX")) m->nitermax = (int)v;
  else if (!strcmp(key, "NITERMIN")) m->nitermin = (int)v;
  else if (!strcmp(key, "DT")) m->p.DT = v;
  else if (!strcmp(key, "quiet")) m->quiet = (int)v;
  else if (!strcmp(key, "profile")) m->profile = (int)v;
  else if (!strcmp(key, "tiled_relax")) m->tiled_relax = (int)v;
  else if (!strcmp(key, "node_pfused")) m->node_pfused = (int)v;
  else if (!strcmp(key, "node_corr_fused")) m->node_corr_fused = (int)v;
  else if (!strcmp(key, "node_rhs_fused")) m->node_rhs_fused = (int)v;
  else if (!strcmp(key, "node_tile_s")) m->node_tile_s = (int)v;
  else if (!strcmp(key, "node_march_tail1")) m->node_march_tail1 = (int)v;
  else if (!strcmp(key, "node_tile_max")) m->node_tile_max = (int)v;
  else if (!strcmp(key, "node_tile_k")) { if (v < 2 || v > 8) return MSOM_ERR_ARG; m->node_tile_k = (int)v; }
  else if (!strcmp(key, "s2_rows")) { m->s2_rows = (int)v; if (m->const_set) return choose_layouts(m); }
  else if (!strcmp(key, "node_split")) { m->node_split = (int)v; if (m->const_set) return choose_layouts(m); }
  else if (!strcmp(key, "node_march")) m->node_march = (int)v;
  else if (!strcmp(key, "node_march_s")) m->node_march_s = (int)v;
  else if (!strcmp(key, "node_march_rows")) { extern int g_node_march_rows; g_node_march_rows = (int)v; }
  else if (!strcmp(key, "mg_coarse")) { m->mg_coarse = (int)v; if (m->const_set) return choose_layouts(m); }
  else if (!strcmp(key, "stochastic")) m->stochastic = (int)v;
  else if (!strcmp(key, "seed")) srand((unsigned)v);
  else { msom_set_error("unknown option %s", key); return MSOM_ERR_ARG; }
  return MSOM_OK;
}
// average duration of a profile slot ("relax_fine", "relax_prolong_fine", "residual", "correct", "rhs", "coarse") since the last reset
extern "C" int msomn_profile_read(msomn_t *m, const char *slot, double *avg_ms, long *launches) {
  if (!m || !slot) return MSOM_ERR_ARG;
  for (int k = 0; k < NP_COUNT; k++)
    if (!strcmp(slot, NP_NAMES[k])) {
      nprof_collect(m, m->prof[k]);
      if (avg_ms) *avg_ms = m->prof[k].launches ? m->prof[k].total_ms / m->prof[k].launches : 0.;
      if (launches) *launches = m->prof[k].launches;
      return MSOM_OK;
    }
  msom_set_error("unknown profile slot %s", slot);
  return MSOM_ERR_ARG;
}
extern "C" int msomn_profile_reset(msomn_t *m) {
  if (!m) return MSOM_ERR_ARG;
  hipStreamSynchronize(m->st);
  for (auto &ps : m->prof) { ps.used = 0; ps.total_ms = 0; ps.launches = 0; }
  return MSOM_OK;
}
extern "C" double msomn_get_param(msomn_t *m, const char *k) {
  if (!m || !k) return NAN;
  if (!strcmp(k, "N")) return m->N;
  if (!strcmp(k, "nl")) return m->nl;
  if (!strcmp(k, "L0")) return m->p.L0;
  if (!strcmp(k, "DT")) return m->p.DT;
  if (!strcmp(k, "tend")) return m->p.tend;
  if (!strcmp(k, "dtout")) return m->p.dtout;
  if (!strcmp(k, "nlevels")) return m->nlev;
  if (!strcmp(k, "iRd2_low")) return m->iRd2_low;
  if (!strcmp(k, "bc_fac")) return m->p.bc_fac;
  if (!strcmp(k, "sqg")) return m->sqg;
  if (!strcmp(k, "s2_xuniform")) return m->s2_xuniform;
  if (!strncmp(k, "split_", 6)) { int l = atoi(k + 6); return l >= 0 && l < m->nlev ? m->lev[l].sp : NAN; }
  if (!strncmp(k, "idh0_", 5)) { int l = atoi(k + 5); return l >= 0 && l < MSOM_MAXNL ? m->lc.idh0[l] : NAN; }
  if (!strncmp(k, "idh1_", 5)) { int l = atoi(k + 5); return l >= 0 && l < MSOM_MAXNL ? m->lc.idh1[l] : NAN; }
  return NAN;
}
#define NEED_FIELD(m, f) if (!(m) || (f) < 0 || (f) >= MSOMN_NFIELDS) { msom_set_error("bad field id %d", (int)(f)); return MSOM_ERR_ARG; }
#define NEED_NCONST(m) if (!(m)) return MSOM_ERR_ARG; if (!(m)->const_set) { msom_set_error("call msomn_set_const first"); return MSOM_ERR_STATE; }

extern "C" int msomn_field_layers(msomn_t *m, int f) { NEED_FIELD(m, f); return m->fl[f]; }
extern "C" int msomn_set_field(msomn_t *m, int f, const double *a) {
  NEED_FIELD(m, f);
  if (!a) return MSOM_ERR_ARG;
  if (f == MSOMN_QFORC3D) m->forcing_3d = 1;
  if (f == MSOMN_PSIPG) m->pg_set = 1;
  if (f == MSOMN_TOPO) m->topo_set = 1;
  return upload_g(m, m->f[f], m->g, m->fl[f], a);
}
extern "C" int msomn_get_field(msomn_t *m, int f, double *a) {
  NEED_FIELD(m, f);
  if (!a) return MSOM_ERR_ARG;
  return download_g(m, m->f[f], m->g, m->fl[f], a);
}

// ---- boundary conditions with set_bc_ms() in force (qg-node/qg.h:197-214, qg_baroclinic_ms.h:55-70)
static double bcc(const msomn *m) { return 2 * m->p.bc_fac / (m->D * m->D); }
static void bnd_psi(msomn *m) { launch_n_bnd_const(m->st, m->f[MSOMN_PSI], m->g, m->nl, m->psi_bc); }
static void bnd_q(msomn *m, double *q) { launch_n_bnd_from(m->st, q, m->f[MSOMN_PSI], m->g, m->nl, bcc(m), 0, m->psi_bc); }
// sqg_baroclinic_ms.h:64-67: the tmp rule subtracts psi_bc instead of the boundary value of zeta
static void bnd_tmp(msomn *m) { launch_n_bnd_from(m->st, m->f[MSOMN_TMP], m->f[MSOMN_ZETA], m->g, m->nl, bcc(m), m->sqg ? 0 : 1, m->sqg ? m->psi_bc : 0.); }

static int read_scalar(msomn *m, int slot, double *out) {
  HIPCHK(hipMemcpyAsync(m->h_scal + slot, m->d_scal + slot, sizeof(double), hipMemcpyDeviceToHost, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  *out = m->h_scal[slot];
  return MSOM_OK;
}

// comp_q_baroclinic / comp_q_barotropic
static int comp_q(msomn *m, const double *psi, double *q) {
  if (m->nl == 1) launch_n_helm(m->st, psi, q, m->g, m->D, m->iRd2_low);
  else {
    launch_n_del2(m->st, psi, q, m->g, m->nl, 0., 1., m->D);
    // sqg: the reference's comp_q_baroclinic (:232-243) still calls the 4-argument comp_stretch; completed with bs
    if (m->sqg) launch_n_stretch_sqg(m->st, psi, m->f[MSOMN_BS], m->f[MSOMN_S2S], q, m->f[MSOMN_S2], m->g, m->nl, 1., 1., m->lc);
    else launch_n_stretch(m->st, psi, q, m->f[MSOMN_S2], m->g, m->nl, 1., 1., m->lc);
  }
  bnd_q(m, q);
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}

// second psi buffer of the out-of-place passes (k_n_rhs_pre, k_n_correct_residual): the pad cells around the grid are written by
// no kernel and no upload, they are zero in both
static int need_psi_alt(msomn *m) {
  if (m->psi_alt) return MSOM_OK;
  HIPCHK(hipMalloc((void **)&m->psi_alt, m->g.ls * m->nl * sizeof(double)));
  HIPCHK(hipMemsetAsync(m->psi_alt, 0, m->g.ls * m->nl * sizeof(double), m->st));   // in the stream of the kernels that follow
  return MSOM_OK;
}
// rhs_pv_baroclinic qg_baroclinic_ms.h:104-196 / rhs_pv_barotropic qg_barotropic.h:16-29
static int rhs_pv_body(msomn *m, double *q, double *dq);
static int rhs_pv(msomn *m, double *q, double *dq) {
  nprof_begin(m, NP_RHS);
  const int r = rhs_pv_body(m, q, dq);
  nprof_end(m, NP_RHS);
  return r;
}
static int rhs_pv_body(msomn *m, double *q, double *dq) {
  const NodeParams &p = m->p;
  const int nl = m->nl;
  const double drag = p.hEkb * p.f0 / (2 * p.dh[nl - 1]);
  if (nl == 1) {
    launch_n_rhs_barotropic(m->st, m->f[MSOMN_PSI], q, m->f[MSOMN_QFORC], dq, m->g, m->D, p.beta, drag, p.nu);
    HIPCHK(hipGetLastError());
    return MSOM_OK;
  }
  double *psi = m->f[MSOMN_PSI], *zeta = m->f[MSOMN_ZETA], *tmp = m->f[MSOMN_TMP], *S2 = m->f[MSOMN_S2];
  if (m->node_rhs_fused) {   // the same arithmetic in three passes (four with the surface term of the SQG variant), kernels_node.hip
    int ra = need_psi_alt(m);
    if (ra) return ra;
    launch_n_rhs_pre(m->st, q, psi, m->psi_alt, zeta, m->f[MSOMN_MASK], m->g, nl, m->D, bcc(m), m->psi_bc);
    std::swap(m->f[MSOMN_PSI], m->psi_alt);
    psi = m->f[MSOMN_PSI];
    launch_n_del2_bnd(m->st, zeta, tmp, m->g, nl, m->D, bcc(m), m->sqg ? 0 : 1, m->sqg ? m->psi_bc : 0.);
    if (m->sqg) launch_n_lap_bs(m->st, m->f[MSOMN_BS], m->d2bs, m->g, m->D);
    launch_n_rhs_all(m->st, psi, zeta, tmp, m->f[MSOMN_PSIPG], S2, m->f[MSOMN_TOPO], m->f[MSOMN_QFORC], m->forcing_3d ? m->f[MSOMN_QFORC3D] : nullptr,
                     m->f[MSOMN_MASK], m->sqg ? m->d2bs : nullptr, m->f[MSOMN_S2S], dq, m->g, nl, m->D, p.beta, drag, p.f0, p.dh[nl - 1], p.nu, -p.nu4, m->lc, m->pg_set, m->topo_set);
    HIPCHK(hipGetLastError());
    return MSOM_OK;
  }
  launch_n_mul_mask(m->st, q, psi, m->f[MSOMN_MASK], m->g, nl);                 // :110-116
  launch_n_del2(m->st, psi, zeta, m->g, nl, 0., 1., m->D);                      // comp_del2(psi, zeta, 0, 1)
  bnd_q(m, zeta);
  launch_n_rhs_main(m->st, psi, zeta, m->f[MSOMN_PSIPG], S2, m->f[MSOMN_TOPO], dq, m->g, nl, 1, 1, m->D, p.beta, drag, p.f0, p.dh[nl - 1], m->lc);
  if (m->sqg) {                                                                 // sqg_baroclinic_ms.h:160-174: del2_bs = laplacian(bs)
    launch_n_lap_bs(m->st, m->f[MSOMN_BS], m->d2bs, m->g, m->D);
    launch_n_stretch_sqg(m->st, zeta, m->d2bs, m->f[MSOMN_S2S], dq, S2, m->g, nl, 1., p.nu, m->lc);
  } else launch_n_stretch(m->st, zeta, dq, S2, m->g, nl, 1., p.nu, m->lc);      // :160
  launch_n_del2(m->st, zeta, tmp, m->g, nl, 0., 1.0, m->D);                     // :162 + boundary(tmp)
  bnd_tmp(m);
  launch_n_axpy(m->st, dq, tmp, m->g, nl, p.nu);                               // :164-167
  const double minus_nu4 = -p.nu4;
  if (m->sqg) launch_n_stretch_sqg(m->st, tmp, m->d2bs, m->f[MSOMN_S2S], dq, S2, m->g, nl, 1., minus_nu4, m->lc);   // del4_bs is laplacian(bs) again, :187-201
  else launch_n_stretch(m->st, tmp, dq, S2, m->g, nl, 1., minus_nu4, m->lc);   // :172
  launch_n_del2(m->st, tmp, dq, m->g, nl, 1., minus_nu4, m->D);                 // :173
  launch_n_add2d(m->st, dq, m->f[MSOMN_QFORC], m->g);                           // :176-180 surface forcing
  if (m->forcing_3d) launch_n_axpy(m->st, dq, m->f[MSOMN_QFORC3D], m->g, nl, 1.);   // :179-185 (FORCING_3D)
  launch_n_mul_mask(m->st, dq, nullptr, m->f[MSOMN_MASK], m->g, nl);            // :186-190
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}

// ---- nodal multigrid
static void relax_level(msomn *m, int k, double *da, const double *res) {
  NLevel &L = m->lev[k];
  for (int c = 0; c < 2; c++) {
    if (k == 0) nprof_begin(m, NP_RELAX);
    if (L.sp) launch_n_relax(m->st, da, res, L.mask_s, L.S2_s, L.ga, m->nl, c, L.D, m->iRd2_low, m->lc, 1, L.S2row);
    else launch_n_relax(m->st, da, res, L.mask, L.S2, L.g, m->nl, c, L.D, m->iRd2_low, m->lc, 0, L.S2row);
    if (k == 0) nprof_end(m, NP_RELAX);
  }
}
// nsweeps red-black sweeps of level k on L.da.  Wide levels: LDS-tiled passes of 2 (or 1) sweeps, out of place
// (L.da <-> L.da2 swap, L.da always the current one); narrow levels: one launch per colour.
// prolong = 1: the correction of level k + 1 has not been prolongated yet; the first colour pass does it on the fly (split levels)
static void relax_sweeps(msomn *m, int k, int nsweeps, int prolong = 0) {
  NLevel &L = m->lev[k];
  if (L.sp && m->node_march_s && L.n + 1 >= m->node_march_s && m->nl <= 6 && (m->nl == 1 || L.S2row)) {
    // chained colour half-sweeps (round 3): the first colour may ride with the prolongation as before; the remaining 2 nsweeps - 1
    // (or 2 nsweeps) half-sweeps go in passes of up to 4, ping-ponging between the two correction buffers; a pass that is
    // followed by more half-sweeps stores only the colour of its last one; a single left-over half-sweep runs in place
    int nh = 2 * nsweeps, c = 0;
    if (prolong && nsweeps >= 1) {
      const NLevel &C = m->lev[k + 1];
      if (k == 0) nprof_begin(m, NP_RELAX_PROLONG);
      launch_n_relax_prolong(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, L.D, m->iRd2_low, m->lc, L.S2row, C.da, C.ga, C.sp);
      if (k == 0) nprof_end(m, NP_RELAX_PROLONG);
      nh--; c = 1;
    }
    while (nh >= 2) {
      const int kmax = m->nl <= 4 ? 4 : (m->nl <= 6 ? 3 : 2);   // windows of K stages x nl layers in registers
      int K = nh < kmax ? nh : kmax;
      if (nh - K == 1 && K > 2 && !m->node_march_tail1) K--;   // leave no single half-sweep behind (node_march_tail1 = 1: do, it then runs as a colour pass)
      if (k == 0) nprof_begin(m, NP_MARCH);
      // a pass stores only the colour of its last half-sweep if ANOTHER PASS follows (which recomputes the other colour before anything
      // reads it); a single colour pass that follows updates interior vertices only, so the pass before it stores both colours
      launch_n_relax_march_s(m->st, L.da, L.da2, L.res, L.mask_s, L.ga, m->nl, c, K, L.D, m->iRd2_low, m->lc, L.S2row, nh - K >= 2);
      if (k == 0) nprof_end(m, NP_MARCH);
      std::swap(L.da, L.da2);
      nh -= K; c = (c + K) & 1;
    }
    if (nh == 1) launch_n_relax(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, c, L.D, m->iRd2_low, m->lc, 1, L.S2row);
    return;
  }
  if (L.sp && m->node_tile_s && L.n + 1 >= m->node_tile_s && L.n + 1 <= m->node_tile_max && m->nl <= 4 && (m->nl == 1 || L.S2row) && nsweeps >= 2) {
    // launch-bound split levels (round 3): first colour with the prolongation as before, then LDS-tiled passes
    // (k_n_relax_tile_s, out of place; up to node_tile_k = 8 half-sweeps each), a single left-over half-sweep as a colour pass
    int nh = 2 * nsweeps, c = 0;
    if (prolong) {
      const NLevel &C = m->lev[k + 1];
      if (k == 0) nprof_begin(m, NP_RELAX_PROLONG);
      launch_n_relax_prolong(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, L.D, m->iRd2_low, m->lc, L.S2row, C.da, C.ga, C.sp);
      if (k == 0) nprof_end(m, NP_RELAX_PROLONG);
      nh--; c = 1;
    }
    while (nh >= 2) {
      const int K = nh < m->node_tile_k ? nh : m->node_tile_k;   // a left-over single half-sweep is cheaper as a colour pass than as a tile pass
      if (k == 0) nprof_begin(m, NP_RELAX);
      launch_n_relax_tile_s(m->st, L.da, L.da2, L.res, L.mask_s, L.ga, m->nl, c, K, L.D, m->iRd2_low, m->lc, L.S2row);
      if (k == 0) nprof_end(m, NP_RELAX);
      std::swap(L.da, L.da2);
      nh -= K; c = (c + K) & 1;
    }
    if (nh == 1) launch_n_relax(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, c, L.D, m->iRd2_low, m->lc, 1, L.S2row);
    return;
  }
  if (L.sp) {  // split layout: a colour pass already moves only the bytes it uses
    for (int s = 0; s < nsweeps; s++) {
      if (prolong && s == 0) {
        const NLevel &C = m->lev[k + 1];
        if (k == 0) nprof_begin(m, NP_RELAX_PROLONG);
        launch_n_relax_prolong(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, L.D, m->iRd2_low, m->lc, L.S2row, C.da, C.ga, C.sp);
        if (k == 0) { nprof_end(m, NP_RELAX_PROLONG); nprof_begin(m, NP_RELAX); }
        launch_n_relax(m->st, L.da, L.res, L.mask_s, L.S2_s, L.ga, m->nl, 1, L.D, m->iRd2_low, m->lc, 1, L.S2row);
        if (k == 0) nprof_end(m, NP_RELAX);
      } else relax_level(m, k, L.da, L.res);
    }
    return;
  }
  if (m->node_march && L.n + 1 >= m->node_march && L.n + 1 >= 64) {
    // 2 nsweeps colour half-sweeps (red, black, red, ...) in passes of up to 4, ping-ponging between the two correction buffers
    int nh = 2 * nsweeps, c = 0;
    while (nh >= 2) {
      int K = nh < 4 ? nh : 4;
      if (nh - K == 1) K--;            // never leave a single half-sweep behind
      launch_n_relax_march(m->st, L.da, L.da2, L.res, L.mask, L.S2, L.g, m->nl, c, K, L.D, m->iRd2_low, m->lc);
      std::swap(L.da, L.da2);
      nh -= K; c = (c + K) & 1;
    }
    if (nh == 1) launch_n_relax(m->st, L.da, L.res, L.mask, L.S2, L.g, m->nl, c, L.D, m->iRd2_low, m->lc);
    return;
  }
  if (!m->tiled_relax || L.n + 1 < 64) {
    for (int s = 0; s < nsweeps; s++) relax_level(m, k, L.da, L.res);
    return;
  }
  for (int s = 0; s < nsweeps;) {
    const int want = nsweeps - s >= 2 ? 2 : 1;
    s += launch_n_relax_tile(m->st, L.da, L.da2, L.res, L.mask, L.S2, L.g, m->nl, want, L.D, m->iRd2_low, m->lc);
    std::swap(L.da, L.da2);
  }
}
// layout of the correction / residual of every level (option node_split; da and res hold nothing between cycles, so the choice
// may change at any time); the levels one workgroup handles (mg_coarse) stay natural
static int choose_layouts(msomn *m) {
  for (int k = 0; k < m->nlev; k++) {   // row tables of an x-independent S2 (levels: injection keeps it x-independent)
    NLevel &L = m->lev[k];
    L.S2row = nullptr;
    if (m->nl > 1 && m->s2_xuniform && m->s2_rows) {
      int r;
      if (!L.S2row_buf && (r = dalloc(&L.S2row_buf, (size_t)m->nlm * (L.n + 1)))) return r;
      launch_n_row_table(m->st, L.S2, L.g, m->nlm, L.S2row_buf);
      L.S2row = L.S2row_buf;
    }
  }
  for (int k = 0; k < m->nlev; k++) {
    NLevel &L = m->lev[k];
    L.sp = m->node_split > 0 && L.n + 1 >= m->node_split && L.n >= 64 && L.n > m->mg_coarse;
    L.ga = L.sp ? node_geom_split(L.n) : L.g;
    if (!L.sp) continue;
    int r;
    if (!L.mask_s && ((r = dalloc(&L.mask_s, L.ga.ls)) || (m->nl > 1 && (r = dalloc(&L.S2_s, L.ga.ls * m->nlm))))) return r;
    launch_n_relayout(m->st, L.mask, L.g, 0, L.mask_s, L.ga, 1, 1);
    if (m->nl > 1) launch_n_relayout(m->st, L.S2, L.g, 0, L.S2_s, L.ga, 1, m->nlm);
  }
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
static int build_levels(msomn *m) {
  launch_n_bnd_const(m->st, m->f[MSOMN_MASK], m->g, 1, 0.);
  for (int k = 1; k < m->nlev; k++) {
    launch_n_restrict(m->st, m->lev[k - 1].mask, m->lev[k - 1].g, m->lev[k].mask, m->lev[k].g, 1, 1);
    launch_n_bnd_const(m->st, m->lev[k].mask, m->lev[k].g, 1, 0.);
    if (m->nl > 1) launch_n_restrict(m->st, m->lev[k - 1].S2, m->lev[k - 1].g, m->lev[k].S2, m->lev[k].g, m->nlm, 2);
  }
  return choose_layouts(m);
}
// host array <-> a level's da / res in whatever layout the level uses (parity tests); da2 is the natural staging buffer
static int upload_lev(msomn *m, NLevel &L, double *dst, const double *a) {
  if (!L.sp) return upload_g(m, dst, L.g, m->nl, a);
  int r = upload_g(m, L.da2, L.g, m->nl, a);
  if (r) return r;
  launch_n_relayout(m->st, L.da2, L.g, 0, dst, L.ga, 1, m->nl);
  return MSOM_OK;
}
static int download_lev(msomn *m, NLevel &L, const double *src, double *a) {
  if (!L.sp) return download_g(m, src, L.g, m->nl, a);
  launch_n_relayout(m->st, src, L.ga, 1, L.da2, L.g, 0, m->nl);
  return download_g(m, L.da2, L.g, m->nl, a);
}
// vpoisson, nodal-poisson.h:19-143: residual first, then (unless converged) one cycle
static int vpoisson(msomn *m, double *&a, const double *b) {
  msom_mgstats mg;
  mg.sum = HUGE_VAL; mg.resa = HUGE_VAL; mg.resb = 0; mg.nrelax = m->nrelax;
  const int nl = m->nl, nlev = m->nlev;
  bool pending_correct = false;
  for (mg.i = 0; mg.i < m->nitermax; mg.i++) {
    HIPCHK(hipMemsetAsync(m->d_scal + NSC_RES, 0, sizeof(double), m->st));
    // the correction of the previous cycle rides in this residual pass (option node_corr_fused; a second psi buffer)
    if (pending_correct && m->node_corr_fused) {
      int ra = need_psi_alt(m);
      if (ra) return ra;
      nprof_begin(m, NP_CORR_RES);
      launch_n_correct_residual(m->st, a, m->psi_alt, m->lev[0].da, m->lev[0].sp ? &m->lev[0].ga : nullptr, m->psi_bc, b, m->lev[0].mask, m->lev[0].S2,
                                m->lev[0].res, m->d_scal + NSC_RES, m->g, nl, m->D, m->iRd2_low, m->lc, m->lev[0].sp ? &m->lev[0].ga : nullptr, m->lev[0].S2row,
                                m->node_corr_fused >= 2);
      nprof_end(m, NP_CORR_RES);
      std::swap(a, m->psi_alt);
    } else {
      if (pending_correct) {
        nprof_begin(m, NP_CORRECT);
        launch_n_correct(m->st, a, m->lev[0].da, m->g, nl, m->psi_bc, m->lev[0].sp ? &m->lev[0].ga : nullptr);
        nprof_end(m, NP_CORRECT);
      }
      nprof_begin(m, NP_RESIDUAL);
      launch_n_residual(m->st, a, b, m->lev[0].mask, m->lev[0].S2, m->lev[0].res, m->d_scal + NSC_RES, m->g, nl, m->D, m->iRd2_low, m->lc,
                        m->lev[0].sp ? &m->lev[0].ga : nullptr, m->lev[0].S2row, 1);
      nprof_end(m, NP_RESIDUAL);
    }
    pending_correct = false;
    // max |res| travels to the host; the restrictions of the cycle that follows unless the solve ends here are queued behind the copy
    // (round 3): the GPU works through them while the host waits for the number and decides -- they write nothing but the coarse levels'
    // residuals, which nobody reads if the solve is over.  (boundary_level of the residual: the pass above stored 0 on the boundary
    // vertices, zb = 1; the boundary vertices of every coarser level end up 0 as well)
    HIPCHK(hipMemcpyAsync(m->h_scal + NSC_RES, m->d_scal + NSC_RES, sizeof(double), hipMemcpyDeviceToHost, m->st));
    if (!m->ev_res) HIPCHK(hipEventCreateWithFlags(&m->ev_res, hipEventDisableTiming));
    HIPCHK(hipEventRecord(m->ev_res, m->st));
    // kc: first level of the group that one workgroup handles in one launch (<= 33^2 vertices, at least two levels)
    int kc = nlev;
    if (m->mg_coarse) {
      int k0 = 0;
      while (k0 < nlev && m->lev[k0].n > m->mg_coarse) k0++;
      if (nlev - k0 >= 2 && nlev - k0 <= NMGC_MAXLEV) kc = k0;
    }
    for (int k = 1; k < nlev && k <= kc; k++)
      launch_n_restrict(m->st, m->lev[k - 1].res, m->lev[k - 1].ga, m->lev[k].res, m->lev[k].ga, nl, 0, m->lev[k - 1].sp, m->lev[k].sp, 1);
    HIPCHK(hipEventSynchronize(m->ev_res));
    const double max = m->h_scal[NSC_RES];
    mg.resa = max;
    if (mg.i == 0) mg.resb = max;
    if (max < m->tolerance && mg.i >= m->nitermin) break;
    if (kc < nlev) {
      NCoarseArgs ca;
      ca.n = nlev - kc; ca.iRd2 = m->iRd2_low; ca.lc = m->lc;
      for (int k = kc; k < nlev; k++) {
        NLevel &L = m->lev[k];
        ca.lev[k - kc] = NCoarseLev{L.da, L.res, L.mask, L.S2, L.g, L.D * L.D};
      }
      nprof_begin(m, NP_COARSE);
      launch_n_mg_coarse(m->st, ca, mg.nrelax, nl);
      nprof_end(m, NP_COARSE);
    } else HIPCHK(hipMemsetAsync(m->lev[nlev - 1].da, 0, m->lev[nlev - 1].g.ls * nl * sizeof(double), m->st));
    // prolongation into level kf: a launch, or (split levels) left to the first colour pass of that level
    auto prolong_into = [&](int kf) -> int {
      if (m->lev[kf].sp && m->node_pfused && mg.nrelax >= 1) return 1;
      launch_n_prolong(m->st, m->lev[kf + 1].da, m->lev[kf + 1].ga, m->lev[kf].da, m->lev[kf].ga, nl, m->lev[kf + 1].sp, m->lev[kf].sp);
      return 0;
    };
    int pending = (kc < nlev && kc > 0) ? prolong_into(kc - 1) : 0;
    for (int k = (kc < nlev ? kc : nlev) - 1; k >= 0; k--) {
      relax_sweeps(m, k, mg.nrelax, pending);
      pending = k > 0 ? prolong_into(k - 1) : 0;
    }
    pending_correct = true;   // a += da: with the next residual, or below if the iteration count ends the loop
  }
  if (pending_correct) {
    nprof_begin(m, NP_CORRECT);
    launch_n_correct(m->st, a, m->lev[0].da, m->g, nl, m->psi_bc, m->lev[0].sp ? &m->lev[0].ga : nullptr);
    nprof_end(m, NP_CORRECT);
  }
  HIPCHK(hipGetLastError());
  if (mg.resa > m->tolerance && !m->quiet)
    fprintf(stderr, "Convergence for psi not reached.\nmg.i = %d, mg.resb: %g mg.resa: %g\n", mg.i, mg.resb, mg.resa);
  m->mg = mg;
  return MSOM_OK;
}
static int invert_q(msomn *m, double *q) {
  const double *b = q;
  if (m->sqg) {  // the known surface term of the top layer moves to the right-hand side (completion, see include/msom.h)
    launch_n_sqg_rhs(m->st, q, m->f[MSOMN_S2S], m->f[MSOMN_BS], m->qeff, m->g, m->nl, m->lc.idh0[0]);
    b = m->qeff;
  }
  int r = vpoisson(m, m->f[MSOMN_PSI], b);
  if (r) return r;
  bnd_psi(m);
  bnd_q(m, q);
  return MSOM_OK;
}
// adjust_dt qg-node/qg.h:258-284 with the `previous` memory of Basilisk's timestep()
static int adjust_dt(msomn *m, double dtmax, double *out) {
  HIPCHK(hipMemsetAsync(m->d_scal + NSC_UMAX, 0, sizeof(double), m->st));
  launch_n_umax(m->st, m->f[MSOMN_PSI], m->d_scal + NSC_UMAX, m->g, m->nl, m->D);
  double um;
  int r = read_scalar(m, NSC_UMAX, &um);
  if (r) return r;
  dtmax /= m->p.CFL;
  if (um != 0.) { const double dt = m->D / um; if (dt < dtmax) dtmax = dt; }
  dtmax *= m->p.CFL;
  if (dtmax > m->previous) dtmax = (m->previous + 0.1 * dtmax) / 1.1;
  m->previous = dtmax;
  *out = dtmax;
  return MSOM_OK;
}

// ---- stochastic forcing: qg-node/qg_stochastic.h (init_stoch :15-47, generate_noise :49-65), advance qg-node/qg.h:306-320
static NatGeom cell_geom(int n) {
  NatGeom g;
  g.nx = g.ny = n;
  g.pitch = ((n + 15) / 16) * 16 + 2 * MSOM_XP;
  g.rows = n + 2 * MSOM_YP;
  g.ls = (size_t)g.pitch * g.rows;
  return g;
}
static int stoch_setup(msomn *m) {  // pyramids of the cell scalar + wavelet coefficients of the uniform filter length L_filt
  const NodeParams &p = m->p;
  int r;
  if (m->cnlev == 0) {
    int c = 1;
    while ((m->N >> c) >= 1) c++;
    m->cnlev = c;
    m->cg.resize(c); m->cs.assign(c, nullptr); m->cr.assign(c, nullptr); m->csig.assign(c, nullptr);
    for (int k = 0; k < c; k++) {
      m->cg[k] = cell_geom(m->N >> k);
      if ((r = dalloc(&m->cs[k], m->cg[k].ls)) || (r = dalloc(&m->cr[k], m->cg[k].ls)) || (r = dalloc(&m->csig[k], m->cg[k].ls))) return r;
    }
  }
  const int K = m->cnlev;
  std::vector<std::vector<double>> sl(K);
  for (int k = 0; k < K; k++) {  // low pass from the finest level down
    const int n = m->N >> k, fx = 2 * n;
    const double Delta = p.L0 / n;
    sl[k].resize((size_t)n * n);
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        double ref_flag = 0;
        if (k > 0) {
          ref_flag += sl[k - 1][(size_t)(2 * j) * fx + 2 * i]; ref_flag += sl[k - 1][(size_t)(2 * j + 1) * fx + 2 * i];
          ref_flag += sl[k - 1][(size_t)(2 * j) * fx + 2 * i + 1]; ref_flag += sl[k - 1][(size_t)(2 * j + 1) * fx + 2 * i + 1];
        }
        double v;
        if (ref_flag > 0) v = 1;
        else if (p.L_filt > 2 * Delta) v = 0;
        else if (p.L_filt <= 2 * Delta && p.L_filt > Delta) v = 1 - (p.L_filt - Delta) / Delta;
        else v = 1;
        sl[k][(size_t)j * n + i] = v;
      }
  }
  for (int k = 0; k < K; k++) {  // high pass, then to the device
    for (double &v : sl[k]) v = 1 - v;
    const NatGeom &g = m->cg[k];
    HIPCHK(hipMemcpy2DAsync(m->csig[k] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), sl[k].data(), g.nx * sizeof(double), g.nx * sizeof(double), g.ny,
                            hipMemcpyHostToDevice, m->st));
  }
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
static void cell_bc(msomn *m, double *f, const NatGeom &g) { launch_fill_ghost(m->st, f, g, 1, BC_NEUMANN, WALL_ALL); }
// wavelet -> scale by sig_lev -> inverse wavelet of the cell field cs[0] (kernels_wavelet.hip, one layer, default BC)
static int cell_wavelet_filter(msomn *m) {
  const int K = m->cnlev;
  cell_bc(m, m->cs[0], m->cg[0]);
  for (int k = 1; k < K; k++) {
    launch_wv_restrict(m->st, m->cs[k - 1], m->cg[k - 1], m->cs[k], m->cg[k], 1);
    cell_bc(m, m->cs[k], m->cg[k]);
  }
  if (K == 1) launch_wv_root(m->st, m->cs[0], m->csig[0], m->cs[0], m->cg[0], 1);
  else {
    launch_wv_root(m->st, m->cs[K - 1], m->csig[K - 1], m->cr[K - 1], m->cg[K - 1], 1);
    cell_bc(m, m->cr[K - 1], m->cg[K - 1]);
  }
  for (int k = K - 2; k >= 0; k--) {
    double *out = k == 0 ? m->cs[0] : m->cr[k];
    launch_wv_recon(m->st, m->cs[k], m->cs[k + 1], m->cr[k + 1], m->csig[k], out, m->cg[k], m->cg[k + 1], 1);
    cell_bc(m, out, m->cg[k]);
  }
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
static int upload_cells(msomn *m, const double *a) {
  const NatGeom &g = m->cg[0];
  HIPCHK(hipMemcpy2DAsync(m->cs[0] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), a, g.nx * sizeof(double), g.nx * sizeof(double), g.ny, hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  cell_bc(m, m->cs[0], g);
  return MSOM_OK;
}
// reference-exact noise: Box-Muller on the serial rand() stream in foreach order (x outer, y inner), qg_stochastic.h:13,51-53
static int generate_noise(msomn *m) {
  const int N = m->N;
  std::vector<double> h((size_t)N * N);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      const double a = sqrt(-2. * log(((double)(rand()) + 1.) / ((double)(RAND_MAX) + 2.)));
      h[(size_t)j * N + i] = m->p.amp_stoch * (a * cos(2 * M_PI * rand() / (double)RAND_MAX));
    }
  int r = upload_cells(m, h.data());
  return r ? r : cell_wavelet_filter(m);
}
extern "C" int msomn_dbg_noise(msomn_t *m, const double *set, int filter, double *get) {
  NEED_NCONST(m);
  if (m->cnlev == 0) { msom_set_error("stochastic forcing is off"); return MSOM_ERR_STATE; }
  int r;
  if (set && (r = upload_cells(m, set))) return r;
  if (filter && (r = cell_wavelet_filter(m))) return r;
  if (get) {
    const NatGeom &g = m->cg[0];
    HIPCHK(hipMemcpy2DAsync(get, g.nx * sizeof(double), m->cs[0] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), g.nx * sizeof(double), g.ny, hipMemcpyDefault, m->st));
  }
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
extern "C" int msomn_dbg_csig(msomn_t *m, int level, double *out) {
  NEED_NCONST(m);
  if (level < 0 || level >= m->cnlev || !out) { msom_set_error("bad level %d", level); return MSOM_ERR_ARG; }
  const NatGeom &g = m->cg[level];
  HIPCHK(hipMemcpy2DAsync(out, g.nx * sizeof(double), m->csig[level] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), g.nx * sizeof(double), g.ny, hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}

// ---- wavelet filter of the vertex model: wavelet_filter qg_baroclinic_ms.h:346-400, sig_lev / mask_c :525-578,
// wavelet_mask / inverse_wavelet_mask qg-node/wavelet_vertex.h:10-46 (the transform runs on the CELL average of psi)
static int wv_setup(msomn *m) {
  const NodeParams &p = m->p;
  int r;
  if (m->wg.empty()) {
    int c = 1;
    while ((m->N >> c) >= 1) c++;
    m->wg.resize(c); m->ws.assign(c, nullptr); m->wr.assign(c, nullptr); m->wsig.assign(c, nullptr); m->wmc.assign(c, nullptr);
    for (int k = 0; k < c; k++) {
      m->wg[k] = cell_geom(m->N >> k);
      if ((r = dalloc(&m->ws[k], m->wg[k].ls * m->nl)) || (r = dalloc(&m->wr[k], m->wg[k].ls * m->nl)) || (r = dalloc(&m->wsig[k], m->wg[k].ls)) ||
          (r = dalloc(&m->wmc[k], m->wg[k].ls)))
        return r;
    }
  }
  const int K = (int)m->wg.size(), N = m->N;
  const size_t n1 = N + 1;
  // sig_lev :527-552 (low pass only): a vertex scalar read at the cell with the same index
  std::vector<double> s2;
  if (p.fac_filt_Rd > 0) {
    if (m->nl < 2) { msom_set_error("fac_filt_Rd > 0 needs the stratification S2 (nl >= 2)"); return MSOM_ERR_CONFIG; }
    s2.resize(n1 * n1 * m->nlm);
    if ((r = download_g(m, m->f[MSOMN_S2], m->g, m->nlm, s2.data()))) return r;
  }
  std::vector<std::vector<double>> sl(K);
  for (int k = 0; k < K; k++) {
    const int n = N >> k, fx = 2 * n;
    const double Delta = p.L0 / n;
    sl[k].resize((size_t)n * n);
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        double ref_flag = 0;
        if (k > 0) {
          ref_flag += sl[k - 1][(size_t)(2 * j) * fx + 2 * i]; ref_flag += sl[k - 1][(size_t)(2 * j + 1) * fx + 2 * i];
          ref_flag += sl[k - 1][(size_t)(2 * j) * fx + 2 * i + 1]; ref_flag += sl[k - 1][(size_t)(2 * j + 1) * fx + 2 * i + 1];
        }
        double v;
        if (ref_flag > 0) v = 1;
        else {
          double L2;
          if (p.fac_filt_Rd > 0) L2 = fmin(p.fac_filt_Rd * p.dh[0] / sqrt(s2[((size_t)j << k) * n1 + ((size_t)i << k)]), p.Lfmax);
          else L2 = p.Lfmax + (j * Delta / p.L0) * (p.Lfmin - p.Lfmax);
          if (L2 > 2 * Delta) v = 0;
          else if (L2 <= 2 * Delta && L2 > Delta) v = 1 - (L2 - Delta) / Delta;
          else v = 1;
        }
        sl[k][(size_t)j * n + i] = v;
      }
    const NatGeom &g = m->wg[k];
    HIPCHK(hipMemcpy2DAsync(m->wsig[k] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), sl[k].data(), g.nx * sizeof(double), g.nx * sizeof(double), g.ny,
                            hipMemcpyHostToDevice, m->st));
  }
  HIPCHK(hipStreamSynchronize(m->st));
  // mask_c :567-575: cell average of the vertex mask, restricted to every level
  launch_wv_vert2cell(m->st, m->f[MSOMN_MASK], m->g, m->wmc[0], m->wg[0], 1);
  for (int k = 1; k < K; k++) launch_wv_restrict(m->st, m->wmc[k - 1], m->wg[k - 1], m->wmc[k], m->wg[k], 1);
  HIPCHK(hipGetLastError());
  m->wv_ready = 1;
  return MSOM_OK;
}
// ws[0] <- inverse_wavelet_mask(sig_lev * wavelet_mask(ws[0])), all layers at once; dirichlet(0) ghost cells on every level
static int wv_masked_apply(msomn *m) {
  const int K = (int)m->wg.size(), nl = m->nl;
  auto bc = [&](double *f, const NatGeom &g) { launch_fill_ghost(m->st, f, g, nl, BC_DIRICHLET0, WALL_ALL); };
  bc(m->ws[0], m->wg[0]);
  for (int k = 1; k < K; k++) {
    launch_wv_restrict(m->st, m->ws[k - 1], m->wg[k - 1], m->ws[k], m->wg[k], nl);
    bc(m->ws[k], m->wg[k]);
  }
  if (K == 1) launch_wv_root_m(m->st, m->ws[0], m->wsig[0], m->wmc[0], m->ws[0], m->wg[0], nl);
  else {
    launch_wv_root_m(m->st, m->ws[K - 1], m->wsig[K - 1], m->wmc[K - 1], m->wr[K - 1], m->wg[K - 1], nl);
    bc(m->wr[K - 1], m->wg[K - 1]);
  }
  for (int k = K - 2; k >= 0; k--) {
    double *out = k == 0 ? m->ws[0] : m->wr[k];
    launch_wv_recon_m(m->st, m->ws[k], m->ws[k + 1], m->wr[k + 1], m->wsig[k], m->wmc[k], out, m->wg[k], m->wg[k + 1], nl);
    bc(out, m->wg[k]);
  }
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
static int invert_q(msomn *m, double *q);
static int comp_q(msomn *m, const double *psi, double *q);
extern "C" int msomn_wavelet_filter(msomn_t *m, double dtflt) {
  NEED_NCONST(m);
  int r;
  if (!m->wv_ready && (r = wv_setup(m))) return r;
  if ((r = invert_q(m, m->f[MSOMN_Q]))) return r;
  launch_wv_vert2cell(m->st, m->f[MSOMN_PSI], m->g, m->ws[0], m->wg[0], m->nl);
  if ((r = wv_masked_apply(m))) return r;
  if (m->p.Lfmax < 1e30)  // `if (Lfmax < HUGE)`, :380
    launch_wv_vertex_update(m->st, m->f[MSOMN_PSI], m->f[MSOMN_PSIF], m->ws[0], m->f[MSOMN_MASK], m->g, m->wg[0], m->nl, dtflt, m->nbar, 1);
  launch_n_bnd_const(m->st, m->f[MSOMN_PSI], m->g, m->nl, m->psi_bc);
  if ((r = comp_q(m, m->f[MSOMN_PSI], m->f[MSOMN_Q]))) return r;
  m->nbar++;
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
// what = 0: sig_lev as used at the cells, 1: mask_c; level k, out [n][n]
extern "C" int msomn_dbg_wv_get(msomn_t *m, int what, int level, double *out) {
  NEED_NCONST(m);
  int r;
  if (!m->wv_ready && (r = wv_setup(m))) return r;
  if (level < 0 || level >= (int)m->wg.size() || !out) { msom_set_error("bad level %d", level); return MSOM_ERR_ARG; }
  const NatGeom &g = m->wg[level];
  const double *src = what ? m->wmc[level] : m->wsig[level];
  HIPCHK(hipMemcpy2DAsync(out, g.nx * sizeof(double), src + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), g.nx * sizeof(double), g.ny, hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
// the masked transform pair alone on a cell field [nl][N][N]
extern "C" int msomn_dbg_wv_apply(msomn_t *m, const double *in, double *out) {
  NEED_NCONST(m);
  int r;
  if (!m->wv_ready && (r = wv_setup(m))) return r;
  if (!in || !out) return MSOM_ERR_ARG;
  const NatGeom &g = m->wg[0];
  for (int l = 0; l < m->nl; l++)
    HIPCHK(hipMemcpy2DAsync(m->ws[0] + nat_idx(g, l, 0, 0), g.pitch * sizeof(double), in + (size_t)l * g.nx * g.ny, g.nx * sizeof(double), g.nx * sizeof(double), g.ny,
                            hipMemcpyDefault, m->st));
  if ((r = wv_masked_apply(m))) return r;
  for (int l = 0; l < m->nl; l++)
    HIPCHK(hipMemcpy2DAsync(out + (size_t)l * g.nx * g.ny, g.nx * sizeof(double), m->ws[0] + nat_idx(g, l, 0, 0), g.pitch * sizeof(double), g.nx * sizeof(double), g.ny,
                            hipMemcpyDefault, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}

extern "C" int msomn_set_const(msomn_t *m) {
  if (!m) return MSOM_ERR_ARG;
  NodeParams &p = m->p;
  const int nl = m->nl, N = m->N;
  const size_t n1 = N + 1;
  int r;
  if (nl > 1) {
    for (int l = 0; l < nl; l++) if (p.dh[l] == 0.) { msom_set_error("thickness = 0: check the definition of dh in params.in"); return MSOM_ERR_CONFIG; }
    double dhc[MSOM_MAXNL];
    for (int l = 0; l < nl - 1; l++) dhc[l] = 0.5 * (p.dh[l] + p.dh[l + 1]);
    m->lc.idh0[0] = m->sqg ? 1. / p.dh[0] : 0.;  // sqg_baroclinic_ms.h:502 "surface layer: 1/h"
    m->lc.idh1[0] = 1. / (dhc[0] * p.dh[0]);
    for (int l = 1; l < nl - 1; l++) { m->lc.idh0[l] = 1. / (dhc[l - 1] * p.dh[l]); m->lc.idh1[l] = 1. / (dhc[l] * p.dh[l]); }
    m->lc.idh0[nl - 1] = 1. / (dhc[nl - 2] * p.dh[nl - 1]); m->lc.idh1[nl - 1] = 0.;
    // S2: N^2 -> f^2 / N^2 with f = f0 + flag_ms beta (y - L0/2)  (qg_baroclinic_ms.h:501-505); init-time host pass
    std::vector<double> h(n1 * n1 * m->nlm);
    if ((r = download_g(m, m->f[MSOMN_S2], m->g, m->nlm, h.data()))) return r;
    for (int l = 0; l < nl - 1; l++) for (size_t j = 0; j < n1; j++) {
      const double f = p.f0 + p.flag_ms * p.beta * (j * m->D - 0.5 * p.L0);
      for (size_t i = 0; i < n1; i++) { double &s = h[(l * n1 + j) * n1 + i]; s = f * f / s; }
    }
    if ((r = upload_g(m, m->f[MSOMN_S2], m->g, m->nlm, h.data()))) return r;
    m->s2_xuniform = 1;
    for (size_t r0 = 0; r0 < (size_t)(nl - 1) * n1 && m->s2_xuniform; r0++)
      for (size_t i = 1; i < n1; i++) if (h[r0 * n1 + i] != h[r0 * n1]) { m->s2_xuniform = 0; break; }
    if (m->sqg) {  // :545 surface layer: f / N^2 (f, not f^2)
      std::vector<double> hs(n1 * n1);
      if ((r = download_g(m, m->f[MSOMN_S2S], m->g, 1, hs.data()))) return r;
      for (size_t j = 0; j < n1; j++) {
        const double f = p.f0 + p.flag_ms * p.beta * (j * m->D - 0.5 * p.L0);
        for (size_t i = 0; i < n1; i++) hs[j * n1 + i] = f / hs[j * n1 + i];
      }
      if ((r = upload_g(m, m->f[MSOMN_S2S], m->g, 1, hs.data()))) return r;
    }
    if (p.scale_topo != 1.) {
      std::vector<double> tp(n1 * n1);
      if ((r = download_g(m, m->f[MSOMN_TOPO], m->g, 1, tp.data()))) return r;
      for (double &v : tp) v *= p.scale_topo;
      if ((r = upload_g(m, m->f[MSOMN_TOPO], m->g, 1, tp.data()))) return r;
    }
  } else if (p.gp_low != 0.) m->iRd2_low = p.f0 * p.f0 / (p.gp_low * p.dh[nl - 1]);
  if ((r = build_levels(m))) return r;
  if (m->stochastic && (r = stoch_setup(m))) return r;  // event init_stoch
  bnd_psi(m);
  if (p.nu != 0) p.DT = 0.5 * fmin(p.DT, m->D * m->D / p.nu / 4.);  // qg-node/qg.h:511-512
  if (p.beta != 0) p.DT = fmin(p.DT, 1 / (2. * p.beta * p.L0));
  if ((r = comp_q(m, m->f[MSOMN_PSI], m->f[MSOMN_Q]))) return r;
  HIPCHK(hipStreamSynchronize(m->st));
  m->const_set = 1;
  m->wv_ready = 0;
  return MSOM_OK;
}

extern "C" int msomn_update(msomn_t *m, int qf, int dqf, double dtmax, double *dt_out) {
  NEED_NCONST(m); NEED_FIELD(m, qf); NEED_FIELD(m, dqf);
  int r;
  if ((r = invert_q(m, m->f[qf])) || (r = rhs_pv(m, m->f[qf], m->f[dqf]))) return r;
  double dt;
  if ((r = adjust_dt(m, dtmax, &dt))) return r;
  if (dt_out) *dt_out = dt;
  return MSOM_OK;
}
extern "C" int msomn_advance(msomn_t *m, int out, int in, int dq, double dt) {
  NEED_FIELD(m, out); NEED_FIELD(m, in); NEED_FIELD(m, dq);
  launch_advance(m->st, m->f[out], m->f[in], m->f[dq], nullptr, m->g, m->nl, dt, 0.);
  if (m->stochastic) {  // qg-node/qg.h:306-320
    if (m->cnlev == 0) { msom_set_error("stochastic: call msomn_set_const after switching it on"); return MSOM_ERR_STATE; }
    m->corrector_step = (m->corrector_step + 1) % 2;
    double dts = sqrt(dt);
    if (m->corrector_step) {
      int r = generate_noise(m);
      if (r) return r;
      dts = dts / sqrt(2);  // to get sqrt(dt)/2 (in the predictor step, dt = dt/2)
    }
    if (m->nl > 1 && !m->quiet) fprintf(stdout, "Stochastic not ready for multilayer yet \n");
    launch_n_add_noise(m->st, m->f[out], m->cs[0], m->g, m->cg[0], dts);
  }
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
extern "C" int msomn_invert_q(msomn_t *m, int qf, msom_mgstats *st) {
  NEED_NCONST(m); NEED_FIELD(m, qf);
  int r = invert_q(m, m->f[qf]);
  if (st) *st = m->mg;
  if (!r) HIPCHK(hipStreamSynchronize(m->st));
  return r;
}
extern "C" int msomn_comp_q(msomn_t *m, int pf, int qf) { NEED_NCONST(m); NEED_FIELD(m, pf); NEED_FIELD(m, qf); return comp_q(m, m->f[pf], m->f[qf]); }
extern "C" int msomn_rhs_pv(msomn_t *m, int qf, int dqf) { NEED_NCONST(m); NEED_FIELD(m, qf); NEED_FIELD(m, dqf); return rhs_pv(m, m->f[qf], m->f[dqf]); }
extern "C" int msomn_dbg_del2_zeta(msomn_t *m) {
  NEED_NCONST(m);
  launch_n_del2(m->st, m->f[MSOMN_PSI], m->f[MSOMN_ZETA], m->g, m->nl, 0., 1., m->D);
  bnd_q(m, m->f[MSOMN_ZETA]);
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
// event forcing (i++), qg-node/qg.c:136-145: q_forcing depends on y and t only -> one row on the host
extern "C" int msomn_forcing(msomn_t *m) {
  if (!m) return MSOM_ERR_ARG;
  const NodeParams &p = m->p;
  const double L0 = p.L0, t = m->t;
  HIPCHK(hipStreamSynchronize(m->st));
  for (int j = 0; j <= m->N; j++) {
    const double y = j * m->D;
    m->h_row[j] = -(p.tau0 + p.tau1 * cos(2 * M_PI * t / p.tf1)) / p.dh[0] * p.forc_mode * M_PI / L0 *
                  sin(p.forc_mode * M_PI * (y + y * (y - L0) * 2 / (L0 * L0) * p.dy_ws * sin(2 * M_PI * t / p.tf2)) / L0);
  }
  HIPCHK(hipMemcpyAsync(m->d_row, m->h_row, (m->N + 1) * sizeof(double), hipMemcpyHostToDevice, m->st));
  launch_n_rowfill(m->st, m->f[MSOMN_QFORC], m->d_row, m->g);
  HIPCHK(hipGetLastError());
  return MSOM_OK;
}
static double dtnext(msomn *m, double dt, double *tnext_out) {  // dtnext() [Basilisk, SURVEY App. B]
  double tnext = m->tnext, t = m->t;
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
extern "C" int msomn_step(msomn_t *m, int with_forcing_event) {
  NEED_NCONST(m);
  int r;
  double d, tn;
  if (with_forcing_event && (r = msomn_forcing(m))) return r;
  if ((r = msomn_update(m, MSOMN_Q, MSOMN_DQ, m->p.DT, &d))) return r;
  m->dt = dtnext(m, d, &tn);
  if ((r = msomn_advance(m, MSOMN_QPRED, MSOMN_Q, MSOMN_DQ, m->dt / 2.))) return r;
  if ((r = msomn_update(m, MSOMN_QPRED, MSOMN_DQ, m->dt, &d))) return r;
  if ((r = msomn_advance(m, MSOMN_Q, MSOMN_Q, MSOMN_DQ, m->dt))) return r;
  m->t = tn;
  m->iter++;
  return MSOM_OK;
}
extern "C" int msomn_set_tnext(msomn_t *m, double t) { if (!m) return MSOM_ERR_ARG; m->tnext = t; return MSOM_OK; }
extern "C" double msomn_time(msomn_t *m) { return m ? m->t : NAN; }
extern "C" double msomn_dt(msomn_t *m) { return m ? m->dt : NAN; }
extern "C" int msomn_iter(msomn_t *m) { return m ? m->iter : MSOM_ERR_ARG; }
extern "C" int msomn_last_mgstats(msomn_t *m, msom_mgstats *s) { if (!m || !s) return MSOM_ERR_ARG; *s = m->mg; return MSOM_OK; }
extern "C" int msomn_ke(msomn_t *m, double *ke) {
  if (!m || !ke) return MSOM_ERR_ARG;
  launch_n_ke(m->st, m->f[MSOMN_PSI], m->partial, m->d_scal + NSC_KE, m->g, m->D);
  double v;
  int r = read_scalar(m, NSC_KE, &v);
  if (r) return r;
  *ke = -v;
  return MSOM_OK;
}

// event write_1d_diag (qg-node/qg.h:361-399): out = {ke, dissipation, forcing}
extern "C" int msomn_diag1d(msomn_t *m, double *out3) {
  if (!m || !out3) return MSOM_ERR_ARG;
  launch_n_diag1d(m->st, m->f[MSOMN_PSI], m->f[MSOMN_Q], m->f[MSOMN_QFORC], m->partial, m->d_scal + NSC_DIAG, m->g, m->p.nu, m->D);
  HIPCHK(hipMemcpyAsync(m->h_scal + NSC_DIAG, m->d_scal + NSC_DIAG, 3 * sizeof(double), hipMemcpyDeviceToHost, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  for (int k = 0; k < 3; k++) out3[k] = -m->h_scal[NSC_DIAG + k];
  return MSOM_OK;
}

// ---- raw multigrid pieces (parity tests)
#define NEED_LEVEL(m, k) if (!(m) || (k) < 0 || (k) >= (m)->nlev) { msom_set_error("bad level %d", (int)(k)); return MSOM_ERR_ARG; }
extern "C" int msomn_dbg_relax(msomn_t *m, int k, double *da, const double *res, int nsweeps) {
  NEED_NCONST(m); NEED_LEVEL(m, k);
  NLevel &L = m->lev[k];
  int r;
  if ((r = upload_lev(m, L, L.da, da)) || (r = upload_lev(m, L, L.res, res))) return r;
  launch_n_bnd_const(m->st, L.da, L.ga, m->nl, 0., L.sp);
  relax_sweeps(m, k, nsweeps);
  HIPCHK(hipGetLastError());
  return download_lev(m, L, L.da, da);
}
extern "C" int msomn_dbg_residual(msomn_t *m, const double *a, const double *b, double *res, double *maxres) {
  NEED_NCONST(m);
  int r;
  if ((r = upload_g(m, m->f[MSOMN_TMP], m->g, m->nl, a)) || (r = upload_g(m, m->f[MSOMN_DQ], m->g, m->nl, b))) return r;
  HIPCHK(hipMemsetAsync(m->d_scal + NSC_RES, 0, sizeof(double), m->st));
  launch_n_residual(m->st, m->f[MSOMN_TMP], m->f[MSOMN_DQ], m->lev[0].mask, m->lev[0].S2, m->lev[0].res, m->d_scal + NSC_RES, m->g, m->nl, m->D, m->iRd2_low,
                    m->lc, m->lev[0].sp ? &m->lev[0].ga : nullptr, m->lev[0].S2row);
  double mx;
  if ((r = read_scalar(m, NSC_RES, &mx))) return r;
  if (maxres) *maxres = mx;
  return download_lev(m, m->lev[0], m->lev[0].res, res);
}
extern "C" int msomn_dbg_restrict(msomn_t *m, int k, const double *fine, double *coarse) {
  NEED_NCONST(m); NEED_LEVEL(m, k); NEED_LEVEL(m, k + 1);
  int r;
  NLevel &Lf = m->lev[k], &Lc = m->lev[k + 1];
  if ((r = upload_lev(m, Lf, Lf.res, fine))) return r;
  launch_n_bnd_const(m->st, Lf.res, Lf.ga, m->nl, 0., Lf.sp);
  launch_n_restrict(m->st, Lf.res, Lf.ga, Lc.res, Lc.ga, m->nl, 0, Lf.sp, Lc.sp);
  launch_n_bnd_const(m->st, Lc.res, Lc.ga, m->nl, 0., Lc.sp);
  return download_lev(m, Lc, Lc.res, coarse);
}
extern "C" int msomn_dbg_prolong(msomn_t *m, int k, const double *coarse, double *fine) {
  NEED_NCONST(m); NEED_LEVEL(m, k); NEED_LEVEL(m, k - 1);
  int r;
  NLevel &Lc = m->lev[k], &Lf = m->lev[k - 1];
  if ((r = upload_lev(m, Lc, Lc.da, coarse))) return r;
  launch_n_bnd_const(m->st, Lc.da, Lc.ga, m->nl, 0., Lc.sp);
  launch_n_prolong(m->st, Lc.da, Lc.ga, Lf.da, Lf.ga, m->nl, Lc.sp, Lf.sp);
  return download_lev(m, Lf, Lf.da, fine);
}
extern "C" int msomn_dbg_level_mask(msomn_t *m, int k, double *out) {
  NEED_NCONST(m); NEED_LEVEL(m, k);
  return download_g(m, m->lev[k].mask, m->lev[k].g, 1, out);
}

// ---- NetCDF-3 IO of vertex fields (qg-node/netcdf_vertex_bas.h:95-424)
extern "C" int msomn_write_nc(msomn_t *m, const char *path) {
  if (!m || !path) return MSOM_ERR_ARG;
  const size_t n1 = m->N + 1, sz = n1 * n1 * m->nl;
  std::vector<double> psi(sz), q(sz);
  int r;
  if ((r = download_g(m, m->f[MSOMN_PSI], m->g, m->nl, psi.data())) || (r = download_g(m, m->f[MSOMN_Q], m->g, m->nl, q.data()))) return r;
  const char *names[2] = {"psi", "q"};
  struct stat sb;
  if (stat(path, &sb) != 0 && msom_nc_create2(path, m->nl, (int)n1, (int)n1, m->p.L0, 2, names, 1)) return MSOM_ERR_IO;
  const double *fields[2] = {psi.data(), q.data()};
  return msom_nc_append(path, m->nl, (int)n1, (int)n1, 2, names, m->t, fields) < 0 ? MSOM_ERR_IO : MSOM_OK;
}
extern "C" int msomn_read_nc(msomn_t *m, int field, const char *path, const char *varname, int record) {
  NEED_FIELD(m, field);
  if (!path || !varname) return MSOM_ERR_ARG;
  const size_t n1 = m->N + 1;
  std::vector<double> h(n1 * n1 * m->fl[field]);
  if (msom_nc_read(path, varname, record, m->fl[field], (int)n1, (int)n1, h.data(), nullptr)) return MSOM_ERR_IO;
  if (field == MSOMN_PSIPG) m->pg_set = 1;
  if (field == MSOMN_TOPO) m->topo_set = 1;
  return upload_g(m, m->f[field], m->g, m->fl[field], h.data());
}

// main() + events of qg-node/qg.c:58-181
extern "C" int msomn_run(msomn_t *m, const char *workdir, long nsteps_max) {
  if (!m) return MSOM_ERR_ARG;
  const NodeParams &p = m->p;
  const char *wd = workdir ? workdir : ".";
  char dpath[600] = "", name[800];
  int r;
  const size_t n1 = m->N + 1;
  for (int i = 1; i < 10000; i++) {  // create_outdir, extra.h:122-135
    snprintf(dpath, sizeof dpath, "%s/outdir_%04d/", wd, i);
    if (mkdir(dpath, 0777) == 0) { fprintf(stdout, "Writing output in %s\n", dpath); break; }
  }
  snprintf(name, sizeof name, "%sparams.in", dpath);  // backup_config
  if (FILE *fp = fopen(name, "w")) { fwrite(m->params_text.data(), 1, m->params_text.size(), fp); fclose(fp); }
  if (!m->const_set) {
    // init (qg_baroclinic_ms.h:478-492): optional input file, variables N2, psi_pg, mask, topo, q_forcing
    if (m->nl > 1) {
      snprintf(name, sizeof name, "%s/input_vars_%dl_N%d.nc", wd, m->nl, m->N);
      struct stat sb;
      if (stat(name, &sb) == 0) {
        fprintf(stdout, "Read input files:\n");
        const struct { int f; const char *v; } in[] = {{MSOMN_S2, "N2"}, {MSOMN_PSIPG, "psi_pg"}, {MSOMN_MASK, "mask"}, {MSOMN_TOPO, "topo"}, {MSOMN_QFORC, "q_forcing"}};
        for (auto &e : in) (void)msomn_read_nc(m, e.f, name, e.v, 0);  // absent variables keep their defaults
        {  // q_forcing_3d: only when the file has it (msomn_read_nc would switch FORCING_3D on)
          std::vector<double> h3((size_t)(m->N + 1) * (m->N + 1) * m->nl);
          if (!msom_nc_read(name, "q_forcing_3d", 0, m->nl, m->N + 1, m->N + 1, h3.data(), nullptr)) (void)msomn_set_field(m, MSOMN_QFORC3D, h3.data());
        }
        fprintf(stdout, "%s .. ok\n", name);
      }
    }
    // set_const qg-node/qg.h:475-479: psi = noise_init (noise() + sin(2 pi y / L0)); noise() in [-1, 1]
    std::vector<double> h(n1 * n1 * m->nl);
    for (size_t j = 0; j < n1; j++) for (size_t i = 0; i < n1; i++) for (int l = 0; l < m->nl; l++)
      h[(l * n1 + j) * n1 + i] = p.noise_init * ((1. - 2. * rand() / (double)RAND_MAX) + sin(2 * M_PI * (j * m->D) / p.L0));
    if ((r = upload_g(m, m->f[MSOMN_PSI], m->g, m->nl, h.data()))) return r;
    snprintf(name, sizeof name, "%s/restart.nc", wd);
    struct stat sb;
    if (stat(name, &sb) == 0) {
      fprintf(stdout, "Read restart file:\n");
      if ((r = msomn_read_nc(m, MSOMN_PSI, name, "psi", -1))) return r;
      fprintf(stdout, "%s .. ok\n", name);
    }
    if ((r = msomn_set_const(m))) return r;
  }
  snprintf(name, sizeof name, "%svars.nc", dpath);
  double tout = 0., tdiag = 0., tflt = p.dtflt;   // event filter (t = dtflt; t <= tend + 1e-10; t += dtflt), qg_baroclinic_ms.h:405-408
  const bool diag = p.dtdiag > 0, filt = p.dtflt > 0;
  char dname[800];
  snprintf(dname, sizeof dname, "%sdiag_1d.dat", dpath);
  long steps = 0;
  for (;;) {
    if (diag && tdiag <= p.tend + 1e-10 && m->t >= tdiag - 1e-12 * fmax(1., fabs(tdiag))) {  // write_1d_diag (qg.h:361-399)
      if (FILE *fp = fopen(dname, "a")) {
        if (m->iter == 0) fprintf(fp, "# time, ke, dissipation, forcing\n");
        else {
          double d3[3];
          if ((r = msomn_diag1d(m, d3))) { fclose(fp); return r; }
          fprintf(fp, "%e, %e, %e, %e\n", m->t, d3[0], d3[1], d3[2]);
        }
        fclose(fp);
      }
      tdiag += p.dtdiag;
    }
    if ((r = msomn_forcing(m))) return r;  // forcing (i++)
    if (filt && tflt <= p.tend + 1e-10 && m->t >= tflt - 1e-12 * fmax(1., fabs(tflt))) {
      fprintf(stdout, "Filter solution\n");
      if ((r = msomn_wavelet_filter(m, p.dtflt))) return r;
      tflt += p.dtflt;
    }
    bool pending = tout <= p.tend + 1e-10;
    if (pending && m->t >= tout - 1e-12 * fmax(1., fabs(tout))) {  // output (t = 0; t <= tend + 1e-10; t += dtout)
      fprintf(stdout, "write file\n");
      if (m->iter == 0 && (r = invert_q(m, m->f[MSOMN_Q]))) return r;
      if ((r = msomn_write_nc(m, name))) return r;
      fprintf(stdout, "file written \n");
      tout += p.dtout;
      pending = tout <= p.tend + 1e-10;
    }
    double ke;
    if ((r = msomn_ke(m, &ke))) return r;
    fprintf(stdout, "i = %i, dt = %g, t = %g, ke_1 = %g\n", m->iter, m->dt, m->t, ke);  // writestdout
    if (!pending) break;
    if (nsteps_max >= 0 && steps >= nsteps_max) break;
    m->tnext = diag && tdiag <= p.tend + 1e-10 ? fmin(tout, tdiag) : tout;
    if (filt && tflt <= p.tend + 1e-10) m->tnext = fmin(m->tnext, tflt);
    if ((r = msomn_step(m, 0))) return r;
    steps++;
  }
  fflush(stdout);
  return m->iter;
}
