// kernels.h -- launch wrappers of the HIP kernels (internal).
#ifndef MSOM_KERNELS_H
#define MSOM_KERNELS_H

#include "msom_internal.h"

// layer metrics idh0/idh1 (msqg/qg.h:1017-1027), passed by value to kernels
struct LayerCoef {
  double idh0[MSOM_MAXNL], idh1[MSOM_MAXNL];
};

// per-level constants of the column solver
struct RelaxCoef {
  double D, sqD;              // Delta and Delta^2 of the level
  double idh0[MSOM_MAXNL], idh1[MSOM_MAXNL];
  // uniform-S fast path: the tridiagonal is the same in every column, so the Thomas
  // factorisation is done once on the host (msqg/poisson_layer.h:137-140)
  double t2[MSOM_MAXNL];      // super-diagonal
  double w[MSOM_MAXNL];       // t0[l] / t1'[l-1]
  double it1[MSOM_MAXNL];     // 1 / t1'[l]
  double S[MSOM_MAXNL];       // uniform S_l (residual, stretching)
};

// ---- kernels_rhs.hip
void launch_fill_ghost(hipStream_t st, double *f, const NatGeom &g, int nl, int bc, int walls, int depth = 1);
void launch_fill_periodic(hipStream_t st, double *f, const NatGeom &g, int nl, int depth);
void launch_fill_lin_dirichlet(hipStream_t st, double *f, const NatGeom &g, int nl, const double *upg, const double *vpg, double D, double Lx,
                               double Ly, int ox = 0, int oy = 0, int sides = WALL_ALL);
void launch_slip_bc(hipStream_t st, const double *po, double *zeta, const NatGeom &g, int nl, double c, int walls);
void launch_pack(hipStream_t st, const double *src, double *dst, const NatGeom &g, int nl);
void launch_unpack(hipStream_t st, const double *src, double *dst, const NatGeom &g, int nl);
void launch_del2(hipStream_t st, const double *po, double *zeta, const NatGeom &g, int nl, double add, double fac, double D);
void launch_stretch(hipStream_t st, const double *po, double *out, const double *S, const NatGeom &g, int nl, double add, double fac,
                    const LayerCoef &lc);
void launch_advection(hipStream_t st, const double *zeta, const double *psi, const double *psipg, const double *zetapg, const double *S,
                      const double *qot, double *dq, const NatGeom &g, int nl, int have_pg, int have_zpg, int stochastic, double D,
                      double beta, double itr_stoch, const LayerCoef &lc);
void launch_umax(hipStream_t st, const double *f, double *partial, double *out, const NatGeom &g, int nl, double D);
void launch_axpy(hipStream_t st, double *dq, const double *x, const NatGeom &g, int nl, double c);
void launch_forcing(hipStream_t st, const double *zeta, const double *psi, const double *qforc, const double *topo, const double *Ro,
                    const double *wind, double *dq, const NatGeom &g, int nl, int have_qforc, int flag_topo, double cs, double cb,
                    double D, double dhb);
void launch_advance(hipStream_t st, double *qo, const double *qi, const double *dq, const double *noise, const NatGeom &g, int nl, double dt,
                    double dts);
int partial_count(const NatGeom &g);
void launch_sum_final(hipStream_t st, const double *partial, double *out, int n);
void launch_ke(hipStream_t st, const double *po, double *partial, double *out, const NatGeom &g, double D);
void launch_sum_layers(hipStream_t st, const double *f, double *partial, double *out, const NatGeom &g, int nl);
void launch_sub_layer_const(hipStream_t st, double *f, const double *sums, const NatGeom &g, int nl, double inv_count);
void launch_make_S(hipStream_t st, const double *Fr, const double *Ro, double *S, const NatGeom &g, int nlm);

void launch_noise(hipStream_t st, double *n, const double *sigma, const NatGeom &g, int nl, double amp, unsigned seed, unsigned draw, int gx0,
                  int gy0, int gnx);

void launch_ptr_rhs(hipStream_t st, const double *psi, const double *c, const double *rel, double *dp, const NatGeom &g, int nl, int np,
                    const double *iPe, const double *ptr_ir, double D);

// energy / PV budgets (msqg/qg_energy.h)
void launch_advection_de(hipStream_t st, const double *zeta, const double *psi, const double *psipg, const double *zetapg, const double *S, double *j1,
                         double *j2, double *j3, const NatGeom &g, int nl, double D, double beta, double dt, double ediag, const LayerCoef &lc);
void launch_dissip_de(hipStream_t st, const double *p4, const double *str, const double *po, double *dq, const NatGeom &g, int nl, double iRe,
                      double iRe4, double dt, double ediag, double D, int stage);
void launch_ekman_de(hipStream_t st, const double *zeta, const double *po, double *dq, const NatGeom &g, int nl, double cs, double cb, double dt,
                     double ediag);
void launch_running_mean(hipStream_t st, double *pm, const double *po, const NatGeom &g, int nl, int n);
void launch_filter_de(hipStream_t st, double *ft, const double *tmp2, double *pm, const NatGeom &g, int nl, double dtflt, double ediag);

// ---- kernels_fused.hip
// optional by-product of the fused tendency + advance pass: the first residual of the next inversion
struct RhsResid {
  double *res, *res_c;       // level-0 residual (split layout) and its restriction to level 1 (or null)
  double *res_max;           // device scalar: max|res| (zeroed by the launcher)
  double *bsum_partial;      // one partial sum of q_out per workgroup (rhs_pipe_blocks entries)
  SplitGeom sg, cg;
};
int rhs_fused_blocks(const NatGeom &g);
void launch_rhs_fused(hipStream_t st, const double *psi, const double *S, const double *qforc, const double *wind, double *dq,
                      double *umax_partial, double *umax_out, const NatGeom &g, int nl, int walls, int uniformS, const double *Su,
                      int have_qforc, double D, double beta, double iRe, double iRe4, double cs, double cb, double slip_c,
                      const LayerCoef &lc, int variant, const double *q_in = nullptr, double *q_out = nullptr, double dt = 0.,
                      const RhsResid *rr = nullptr, int region = 0, const double *dt_ptr = nullptr);
int rhs_pipe_blocks(const NatGeom &g);
// ---- kernels_lpw.hip: same pass, one layer per wavefront, register windows + DPP (chunk_rows <= 0: automatic)
void launch_rhs_lpw(hipStream_t st, const double *psi, const double *S, const double *qforc, const double *wind, double *dq, const NatGeom &g,
                    int nl, int walls, int uniformS, const double *Su, int have_qforc, double D, double beta, double iRe, double iRe4, double cs,
                    double cb, double slip_c, const LayerCoef &lc, const double *q_in, double *q_out, double dt, int chunk_rows, int stoch = 0,
                    const double *q_stage = nullptr, const double *noise = nullptr, double crelax = 0., double dts = 0., int region = 0,
                    const double *dt_ptr = nullptr);

// ---- kernels_mg.hip
// coarse part of the multigrid cycle in one launch (k_mg_coarse): lev[0] = finest of the group
#define MGC_MAXLEV 8
#define MGC_MAXDIM 32
#define MGC_NT 512
struct CoarseLev { double *da, *res; const double *S; SplitGeom g; RelaxCoef rc; };
struct CoarseArgs { CoarseLev lev[MGC_MAXLEV]; int n, walls, prolong_fused, lds; };
void launch_mg_coarse(hipStream_t st, const CoarseArgs *d_args, int nrelax, int nl, int uniformS, int lean = 0);
size_t mg_coarse_lean_doubles(const CoarseArgs &h, int nl);   // LDS doubles the lean form needs (<= MGC_POOL)
#define MGC_POOL 19200  // doubles of LDS pool of the one-launch coarse kernels: 150 KB
size_t mg_coarse_static_lds();  // bytes of static LDS k_mg_coarse declares (its launch needs a device that grants them)
void launch_nat_to_split(hipStream_t st, const double *nat, const NatGeom &g, double *sp, const SplitGeom &sg, int nl);
void launch_split_to_nat(hipStream_t st, const double *sp, const SplitGeom &sg, double *nat, const NatGeom &g, int nl);
void launch_split_pack(hipStream_t st, const double *src, double *sp, const SplitGeom &sg, int nl, int bc, int walls);
void launch_split_unpack(hipStream_t st, const double *sp, const SplitGeom &sg, double *dst, int nl);
void launch_residual(hipStream_t st, const double *a, const double *b, const double *S, const NatGeom &g, double *res, const SplitGeom &sg,
                     int nl, const RelaxCoef &rc, int uniformS, double *maxres, double *sum_partial, int want_sum);
int residual2_blocks(const NatGeom &g);
void launch_residual2(hipStream_t st, int mode, const double *a, const double *da, double *a_out, const double *b, const double *S,
                      const NatGeom &g, double *res, const SplitGeom &sg, double *res_c, const SplitGeom &cg, int nl, const RelaxCoef &rc,
                      int uniformS, int walls, double *maxres, double *sum_partial, int want_sum, double *umax_partial, double *umax_out,
                      int umax_clean = 0,
                      double *res_c2 = nullptr, const SplitGeom *cg2 = nullptr);
void launch_restrict(hipStream_t st, const double *fine, const SplitGeom &fg, double *coarse, const SplitGeom &cg, int nl);
void launch_restrict_pyramid(hipStream_t st, const double *fine, double *const *out, const SplitGeom *g, int n, int nl);   // g[0] fine, g[1..n] outputs; n <= 5
void launch_prolong(hipStream_t st, const double *coarse, const SplitGeom &cg, double *fine, const SplitGeom &fg, int nl, int walls);
void launch_relax_color(hipStream_t st, double *da, const double *res, const double *S, const SplitGeom &sg, int nl, const RelaxCoef &rc,
                        int uniformS, int color, int walls, int fine, int region = 0);
void launch_relax_ring(hipStream_t st, double *da, const double *res, const double *S, const SplitGeom &sg, int nl, const RelaxCoef &rc,
                       int uniformS, int color, int walls);
int launch_relax_block8(hipStream_t st, const double *da_in, const double *coarse, const SplitGeom &cg, const double *res, double *da_out,
                        const SplitGeom &sg, int nl, const RelaxCoef &rc, int walls, int nh, int c0, const double *S = nullptr);
void launch_relax_block2(hipStream_t st, const double *da_in, const double *coarse, const SplitGeom &cg, const double *res, double *da_out,
                         const SplitGeom &sg, int nl, const RelaxCoef &rc, int walls, int fine);
void launch_relax_red_prolong(hipStream_t st, double *da, const double *coarse, const SplitGeom &cg, const double *res, const double *S,
                              const SplitGeom &sg, int nl, const RelaxCoef &rc, int uniformS, int walls);
void launch_correct(hipStream_t st, double *a, const NatGeom &g, const double *da, const SplitGeom &sg, int nl, int walls);

// ---- kernels_march.hip: K = 2..4 consecutive half-sweeps in one pass (register windows, out of place)
// tiles: the `rows` nearest rows of the S / N neighbour tiles of the input and of the residual (null at a wall), laid
// out like `rows` rows of the level (ls doubles per layer)
// last pass of the finest level: psi_out = psi + da instead of storing da
// more_follow: further half-sweeps of the level come after this pass, so it only stores the colour of its last half-sweep
// (the other colour is recomputed by the next half-sweep before anything reads it): w/2 fewer bytes written
struct MarchCorrect { const double *psi; double *psi_out; NatGeom g; };
struct MarchHalo { const double *in_s, *in_n, *res_s, *res_n; size_t ls; int rows; };
int launch_relax_march(hipStream_t st, const double *in, double *out, const double *res, const SplitGeom &sg, int nl, const RelaxCoef &rc, int c1,
                       int K, int walls, int chunk_rows = 0, const MarchHalo *h = nullptr, const double *coarse = nullptr, const SplitGeom *cg = nullptr,
                       const MarchCorrect *mc = nullptr, int more_follow = 0, const MarchHalo *coarse_halo = nullptr, int region = 0);

// ---- kernels_wavelet.hip
void launch_wv_restrict(hipStream_t st, const double *f, const NatGeom &fg, double *c, const NatGeom &cg, int nl);
void launch_wv_recon(hipStream_t st, const double *s, const double *sc, const double *rc, const double *sig, double *out, const NatGeom &fg,
                     const NatGeom &cg, int nl);
void launch_wv_root(hipStream_t st, const double *s, const double *sig, double *r, const NatGeom &g, int nl);
void launch_wv_recon_m(hipStream_t st, const double *s, const double *sc, const double *rc, const double *sig, const double *mc, double *out,
                       const NatGeom &fg, const NatGeom &cg, int nl);
void launch_wv_root_m(hipStream_t st, const double *s, const double *sig, const double *mc, double *r, const NatGeom &g, int nl);
void launch_wv_vert2cell(hipStream_t st, const double *v, const NatGeom &vg, double *c, const NatGeom &cg, int nl);
void launch_wv_vertex_update(hipStream_t st, double *psi, double *psif, const double *c, const double *mask, const NatGeom &vg, const NatGeom &cg, int nl,
                             double dtflt, int nbar, int update_mean);
void launch_wv_qof(hipStream_t st, double *qof, double *q, const double *tmp, const NatGeom &g, int nl, double dtflt, int nbar, int restore);

// ---- kernels_node.hip (vertex-grid variant, qg-node/)
void launch_n_bnd_from(hipStream_t st, double *f, const double *g, const NatGeom &ge, int nl, double c, int use_g_bnd, double gbc);
void launch_n_bnd_const(hipStream_t st, double *f, const NatGeom &ge, int nl, double v, int sp = 0);
void launch_n_mul_mask(hipStream_t st, double *a, double *b, const double *mk, const NatGeom &g, int nl);
void launch_n_del2(hipStream_t st, const double *in, double *out, const NatGeom &g, int nl, double add, double fac, double D);
void launch_n_stretch(hipStream_t st, const double *in, double *out, const double *S2, const NatGeom &g, int nl, double add, double fac,
                      const LayerCoef &lc);
void launch_n_stretch_sqg(hipStream_t st, const double *in, const double *bs, const double *S2S, double *out, const double *S2, const NatGeom &g, int nl,
                          double add, double fac, const LayerCoef &lc);
void launch_n_lap_bs(hipStream_t st, const double *bs, double *out, const NatGeom &g, double D);
void launch_n_sqg_rhs(hipStream_t st, const double *q, const double *S2S, const double *bs, double *qeff, const NatGeom &g, int nl, double idh00);
void launch_n_rhs_main(hipStream_t st, const double *psi, const double *zeta, const double *pg, const double *S2, const double *topo, double *dq,
                       const NatGeom &g, int nl, int have_pg, int have_topo, double D, double beta, double drag, double f0, double dhb, const LayerCoef &lc);
void launch_n_axpy(hipStream_t st, double *dq, const double *x, const NatGeom &g, int nl, double c);
void launch_n_rhs_pre(hipStream_t st, double *q, const double *psi, double *psi_out, double *zeta, const double *mk, const NatGeom &g, int nl, double D,
                      double bc, double gbc);
void launch_n_del2_bnd(hipStream_t st, const double *in, double *out, const NatGeom &g, int nl, double D, double bc, int use_bnd, double gbc);
void launch_n_rhs_all(hipStream_t st, const double *psi, const double *zeta, const double *tmp, const double *pg, const double *S2, const double *topo,
                      const double *qf, const double *qf3d, const double *mk, const double *d2bs, const double *S2S, double *dq, const NatGeom &g, int nl,
                      double D, double beta, double drag, double f0, double dhb, double nu, double mnu4, const LayerCoef &lc, int have_pg, int have_topo);
void launch_n_add2d(hipStream_t st, double *dq, const double *qf, const NatGeom &g);
void launch_n_rhs_barotropic(hipStream_t st, const double *psi, const double *q, const double *qf, double *dq, const NatGeom &g, double D, double beta,
                             double drag, double nu);
void launch_n_helm(hipStream_t st, const double *psi, double *q, const NatGeom &g, double D, double iRd2);
void launch_n_rowfill(hipStream_t st, double *f, const double *row, const NatGeom &g);
// sp = 1: a, b, mk, S2 in the x-parity split layout (g = their geometry)
void launch_n_relax(hipStream_t st, double *a, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl, int color, double D,
                    double iRd2, const LayerCoef &lc, int sp = 0, const double *S2row = nullptr);
int launch_n_relax_march(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl,
                         int color, int K, double D, double iRd2, const LayerCoef &lc);
int launch_n_relax_march_s(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const NatGeom &g, int nl, int color, int K,
                           double D, double iRd2, const LayerCoef &lc, const double *S2row, int partial);
int launch_n_relax_tile_s(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const NatGeom &g, int nl, int color, int K,
                          double D, double iRd2, const LayerCoef &lc, const double *S2row);
int launch_n_relax_tile(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const double *S2, const NatGeom &g,
                        int nl, int ns, double D, double iRd2, const LayerCoef &lc);
void launch_n_residual(hipStream_t st, const double *a, const double *b, const double *mk, const double *S2, double *res, double *maxres,
                       const NatGeom &g, int nl, double D, double iRd2, const LayerCoef &lc, const NatGeom *gres = nullptr,
                       const double *S2row = nullptr, int zb = 0);  // gres: res in the split layout; S2row: row table of an x-independent S2; zb: 0 on the boundary vertices
// coarse levels of a vpoisson cycle in one launch (k_n_mg_coarse): lev[0] = finest of the group
#define NMGC_MAXLEV 8
#define NMGC_NT 1024
struct NCoarseLev { double *da, *res; const double *mask, *S2; NatGeom g; double sqD; };
struct NCoarseArgs { NCoarseLev lev[NMGC_MAXLEV]; int n; double iRd2; LayerCoef lc; };
void launch_n_mg_coarse(hipStream_t st, const NCoarseArgs &a, int nrelax, int nl);
void launch_n_restrict(hipStream_t st, const double *f, const NatGeom &fg, double *c, const NatGeom &cg, int nl, int kind, int fsp = 0, int csp = 0, int zb = 0);
void launch_n_prolong(hipStream_t st, const double *c, const NatGeom &cg, double *f, const NatGeom &fg, int nl, int csp = 0, int fsp = 0);
void launch_n_correct(hipStream_t st, double *a, const double *da, const NatGeom &g, int nl, double bcv, const NatGeom *gda = nullptr);
void launch_n_correct_residual(hipStream_t st, const double *a, double *a_out, const double *da, const NatGeom *gda, double bcv, const double *b, const double *mk,
                               const double *S2, double *res, double *maxres, const NatGeom &g, int nl, double D, double iRd2, const LayerCoef &lc,
                               const NatGeom *gres, const double *S2row, int march);
void launch_n_row_table(hipStream_t st, const double *f, const NatGeom &g, int nl, double *out);
void launch_n_relax_prolong(hipStream_t st, double *a, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl, double D,
                            double iRd2, const LayerCoef &lc, const double *S2row, const double *coarse, const NatGeom &cg, int csp);
void launch_n_relayout(hipStream_t st, const double *src, const NatGeom &sg, int ssp, double *dst, const NatGeom &dg, int dsp, int nl);
void launch_n_umax(hipStream_t st, const double *psi, double *out, const NatGeom &g, int nl, double D);
void launch_n_add_noise(hipStream_t st, double *q, const double *n, const NatGeom &g, const NatGeom &cg, double dts);
void launch_n_diag1d(hipStream_t st, const double *psi, const double *q, const double *qf, double *partial, double *out3, const NatGeom &g, double nu, double D);
void launch_n_ke(hipStream_t st, const double *psi, double *partial, double *out, const NatGeom &g, double D);

#endif
