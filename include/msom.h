/*
 * msom.h -- C ABI of libmsomhip: the MI355X-native multi-layer quasi-geostrophic
 * timestepper that replaces the PV-advection + streamfunction-inversion hot path of
 * bderembl/msom (msqg/qg.h, msqg/poisson_layer.h, msqg/layer.h, driver msqg/qg.c).
 *
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference tree).  Plain pointers and sizes only.  All field arrays are fp64, numpy
 * C-order [layer][y][x], interior points only (msqg/qg.h:1164-1188); every `double*`
 * field argument may be a host pointer or a HIP device pointer (the library copies with
 * hipMemcpyDefault).  The caller owns every buffer it passes.
 *
 * Error convention: the reference prints to stdout and calls exit(0) on bad input
 * (msqg/qg.h:735-738,990-1012).  The library never exits: functions return MSOM_OK (0) or
 * a negative code and msom_last_error() holds the message.  Multigrid non-convergence is
 * not an error (reference: stderr warning, mspg/elliptic.h:215-219); it is reported in
 * msom_mgstats.
 */
#ifndef MSOM_H
#define MSOM_H

#ifdef __cplusplus
extern "C" {
#endif

#define MSOM_OK 0
#define MSOM_ERR_ARG (-1)      /* bad argument / unknown key / unknown field          */
#define MSOM_ERR_IO (-2)       /* file not found or short read                         */
#define MSOM_ERR_CONFIG (-3)   /* dh == 0, Rom <= 0, N not a power of two, nl too big  */
#define MSOM_ERR_HIP (-4)      /* HIP runtime error (no device, out of memory, ...)    */
#define MSOM_ERR_COMM (-5)     /* RCCL error                                           */
#define MSOM_ERR_STATE (-6)    /* call order violated (e.g. step before set_const)     */

#define MSOM_MAXNL 16          /* layers supported (msqg/poisson_layer.h:77 sizes its column arrays by nl) */
#define MSOM_FASTNL 8          /* up to here: the register-resident kernels (chained smoother, one-launch coarse levels, fused
                                  tendency pass); 9 .. MSOM_MAXNL: one kernel per reference loop, same arithmetic */

/* field ids, mirroring the reference's global layer lists (msqg/qg.h:22-57) */
enum {
  MSOM_PSI = 0,    /* pol    stream function                 */
  MSOM_Q = 1,      /* qol    potential vorticity (evolving)  */
  MSOM_ZETA = 2,   /* zetal  relative vorticity              */
  MSOM_PSIPG = 3,  /* ppl    large-scale stream function     */
  MSOM_ZETAPG = 4, /* zetapl large-scale relative vorticity  */
  MSOM_QFORC = 5,  /* q_forcl 3-D PV forcing                 */
  MSOM_TMP = 6,    /* tmpl                                   */
  MSOM_FR = 7,     /* Frl    Froude number, nl-1 layers      */
  MSOM_S = 8,      /* strl   (Fr/Ro)^2, nl-1 layers          */
  MSOM_DQ = 9,     /* updates                                */
  MSOM_RO = 10,    /* Ro, 1 layer                            */
  MSOM_TOPO = 11,  /* topo, 1 layer                          */
  MSOM_QPRED = 12, /* predictor                              */
  MSOM_NOISE = 13, /* n_stochl (msqg/qg_stochastic.h:13)     */
  MSOM_SIGMA = 14, /* s_stochl (msqg/qg_stochastic.h:14)     */
  /* passive tracers (nptr > 0): nl*nptr layers, index l*nptr + nt (msqg/qg.h:100-101) */
  MSOM_PTR = 15,       /* ptracersl                          */
  MSOM_PTR_RELAX = 16, /* ptr_relaxl                         */
  MSOM_DPTR = 17,      /* tracer part of `updates`           */
  MSOM_PTR_PRED = 18,  /* tracer part of the predictor       */
  MSOM_RD = 19,        /* Rd, deformation radius of the filter scale, 1 layer (msqg/qg.h:47,913,963-968) */
  MSOM_QOF = 20,       /* qofl, filter mean (msqg/qg.h:27,549); allocated on first use           */
  /* energy / PV budgets (msqg/qg_energy.h:7-15), nl layers each, allocated on first use */
  MSOM_DE_BF = 21, MSOM_DE_VD = 22, MSOM_DE_J1 = 23, MSOM_DE_J2 = 24, MSOM_DE_J3 = 25, MSOM_DE_FT = 26,
  MSOM_TMP2 = 27, MSOM_PO_MFT = 28,
  MSOM_NFIELDS = 29
};

/* mgstats of Basilisk (text: mspg/elliptic.h:118-123), kept by the reference in `mgpsi`
 * (msqg/qg.h:61) */
typedef struct {
  int i;             /* number of multigrid cycles                */
  double resb, resa; /* max |residual| before and after           */
  double sum;        /* sum of the right-hand side                */
  int nrelax;        /* relaxations per level at exit             */
} msom_mgstats;

typedef struct msom msom_t;

const char *msom_last_error(void);
const char *msom_version(void);

/* ---- lifecycle: read_params -> init_grid -> set_vars   (msqg/qg.c:34-47, qg.h:689-761,837-925)
 * msom_create parses a params.in file; msom_create_str parses the same text from memory.
 * Extension keys (ignored by the reference parser, qg.h:698-731): Ny (non-square domain),
 * TOLERANCE, NITERMAX, NITERMIN.  Returns NULL on error. */
msom_t *msom_create(const char *params_path);
msom_t *msom_create_str(const char *params_text);
/* trash_vars, msqg/qg.h:1130-1154 */
int msom_destroy(msom_t *m);

/* run-time counterparts of the reference's compile-time flags and Basilisk globals:
 * "TOLERANCE" "NITERMAX" "NITERMIN" (mspg/elliptic.h:111-112, qg.h:159), "DT", "quiet",
 * "stochastic" (-D_STOCHASTIC), "seed", "noise_mode" (0: the reference's serial rand() stream generated on
 * the host, 1: counter-based Philox on the device), "flag_topo", "uniform_S" (0 forces the general
 * S-field kernels), "profile" (HIP-event timing of the finest-level launches: 1 every kernel, 2 only the chained smoother passes;
 * an event pair costs ~10 us of stream time).
 * Implementation switches, all result-preserving in the strict build (defaults in brackets):
 * "fused" [1] one-pass tendency kernel, "adv_fused" [1] advance folded into it, "stoch_fused" [1] (product build only) the
 * stochastic variant rides in that kernel too: -q/tau and the noise are read in its finalisation next to q_in, "rhs_variant" [6: one layer per
 * wavefront with register windows; 1: LDS tiles],
 * "rhs_resid" [0] first residual of the next inversion as its by-product, "mg_fused" [1] fused
 * residual/restriction and correction/residual passes, "prolong_fused" [1], "mg_coarse" [4] coarse levels
 * in one launch (1: their arrays in global memory, 2: resident in LDS, 3: as 2 with the LDS pool pre-filled with NaN -- test aid; round 3,
 * default 4: the lean LDS form k_mg_coarse_lean where it applies -- uniform S or one layer, walls or doubly periodic -- else as 2),
 * "resmax_rows" [0 = 32] rows per chunk of the marching max-only residual pass (-1: the LDS-tiled kernel), "march" [1] chained half-sweep smoother on HBM-bound single-GPU levels (2: on every level that is
 * wide enough), "march_k" [4] half-sweeps per pass, "march_rows" [0 = auto] chunk height, "march_min" [23] log2 of the cell-layers a level needs, "march_prolong" [1] prolongation folded
 * into the first pass, "march_dma" [2] memory side of the pass (0: register-window loads, 1: LDS-DMA prefetch with one strip per
 * workgroup, 2: four strips per workgroup marching in step; PROCESS-WIDE tuning knob like march_rows / march_xcd / march_flip /
 * march_dbg / rhs_dbg / block_variant, which are globals of the library rather than fields of the handle), "graph" [0] replay the launches of a multigrid cycle from a captured hipGraph on the launch-bound grids
 * (measured neutral), "march_partial" [1], "march_correct" [1] correction folded into the last pass, "march_xcd" [1] XCD-contiguous block numbering, "march_flip" [1] odd chunks march downwards, "block_sweeps" [0] LDS-tiled blocked smoother (2 sweeps per launch, every level), "restrict2" [1] the pre-cycle residual pass restricts two levels down, "restrict_pyr" [1] the rest of the restriction chain in launches of up to five levels, "step_sync" [-1] (see msom_sync), "block8" [1] / "block8_max" [1024] round 3: prolongation + up to 8 half-sweeps of a
 * visit of a launch-bound level (64 .. block8_max cells a side, not marched; one GPU, walls or doubly periodic, nl <= 8; uniform or general S) in one launch of that kernel with a halo of 8, "agglomerate" [1] / "agg_size" [256]
 * gathered coarse levels of tiled runs, "mg_global_sum" [0]; "rhs_dbg", "block_variant": timing
 * experiments of tools/. */
int msom_set_option(msom_t *m, const char *key, double value);
/* parsed / derived parameters: N nx ny nl L0 DT iRe iRe4 CFL Rom tend dtout beta tau0 Ekb Eks
 * sbc idh0_<l> idh1_<l> Fr_<l> dh_<l> nlevels */
double msom_get_param(msom_t *m, const char *key);

/* pyset_field / pyget_field, msqg/qg.h:1164-1188 (array [layer][y][x]; BC applied after set) */
int msom_set_field(msom_t *m, int field, const double *a);
int msom_get_field(msom_t *m, int field, double *a);
int msom_field_layers(msom_t *m, int field);
/* mean removal of the initial condition, msqg/qg.c:65-70 */
int msom_remove_mean(msom_t *m, int field);
/* set_const, msqg/qg.h:931-1116: layer metrics, Ro, S = (Fr/Ro)^2, q = comp_q(psi), BCs.
 * Input files (psipg_, frpg_, topo, qforc_, p0.bas ...) are read by msom_read_inputs. */
int msom_set_const(msom_t *m);
/* optional input-file discovery in `dir` (msqg/qg.h:940-984, msqg/qg.c:55-59) */
int msom_read_inputs(msom_t *m, const char *dir);

/* ---- the two hooks Basilisk's run() calls (installed at msqg/qg.h:922-923) */
/* update_qg, msqg/qg.h:609-650: dq/dt = F(q); returns the CFL-limited dtmax (< 0 on error) */
double msom_update(msom_t *m, const double *q, double *dqdt, double dtmax);
/* advance_qg, msqg/qg.h:594-606 (stochastic: msqg/qg_stochastic.h:128-149) */
int msom_advance(msom_t *m, double *qout, const double *qin, const double *dqdt, double dt);

/* ---- elliptic plug-in (poisson_layer -> mg_solve(relax_layer, residual_layer),
 * msqg/poisson_layer.h:263-306; invertq msqg/qg.h:114-163).  psi is warm start and result. */
int msom_invertq(msom_t *m, const double *q, double *psi, msom_mgstats *st);
/* comp_q, msqg/qg.h:397-403 */
int msom_comp_q(msom_t *m, const double *psi, double *q);

/* ---- the array-level operator API of the reference's Python module, same argument order
 * (msqg/qg_bfn.h:21-103; SWIG typemaps msqg/qg_bfn.i) */
int pystep_bfn(msom_t *m, double *varin_py, int len1, int len2, int len3, double *tend_py, int len4, int len5,
               int len6, double direction, int vartype);
int pyq2p(msom_t *m, double *po_py, int len7, int len8, int len9, double *qo_py, int len10, int len11, int len12);
int pyp2q(msom_t *m, double *po_py, int len13, int len14, int len15, double *qo_py, int len16, int len17, int len18);

/* ---- time loop of Basilisk predictor-corrector run() as driven by msqg/qg.c
 * msom_step: one RK2 step on the internal state (update, dtnext, advance dt/2, update,
 * advance dt).  msom_set_tnext gives the time of the next t-scheduled event (output). */
int msom_step(msom_t *m, double *dt_used);
int msom_set_tnext(msom_t *m, double tnext);
double msom_time(msom_t *m);
int msom_iter(msom_t *m);
/* kinetic-energy diagnostic of the per-step stdout line, msqg/qg.c:101-109 */
double msom_ke(msom_t *m);
int msom_last_mgstats(msom_t *m, msom_mgstats *st);
/* msom_run: whole main() loop of msqg/qg.c:34-173 (stdout line, po/qo .bas every dtout,
 * outdir_%04d creation, params.in backup).  nsteps_max < 0: run to tend. */
int msom_run(msom_t *m, const char *workdir, long nsteps_max);

/* ---- .bas IO, msqg/auxiliar_input.h:24-59 (input_matrixl), :101-167 (output_matrixl, write_field) */
/* tiled models: msom_write_* / msom_read_* / msom_read_inputs / msom_run are collective (every rank calls them): the
 * global field is gathered and rank 0 writes; every rank reads the shared file and keeps its tile */
int msom_write_bas(msom_t *m, int field, const char *path);
int msom_read_bas(msom_t *m, int field, const char *path);

/* ---- NetCDF-3 classic output / restart (libnetcdf-free): create_nc + write_nc + read_nc of
 * newqg/netcdf_bas.h:42-244 and qg-node/netcdf_vertex_bas.h:315-424.  msom_write_nc appends one
 * record (time = t; variables "psi" and "q", all levels, float) to `path`, creating the file
 * with dims level,y,x,time(UNLIMITED) and coordinate variables time,y,x when it does not exist.
 * msom_read_nc loads record `record` (-1 = last) of variable `varname` into `field`
 * (restart: "psi" -> MSOM_PSI, then msom_set_const). */
int msom_write_nc(msom_t *m, const char *path);
int msom_read_nc(msom_t *m, int field, const char *path, const char *varname, int record);

/* ---- wavelet scale filter, the "multiple scale" part of msom (msqg/qg.h:509-560; event filter :655-658;
 * coefficients sig_lev from sig_filt = min(afilt * Rd, Lfmax), :1059-1090).  Saves q in tmp, inverts
 * q -> psi, removes from every layer of psi the wavelet details selected by sig_lev, recomputes q and
 * sets qof = (q_before - q_after) / dtflt; dtflt < 0 (energy diagnostics, qg_energy.h:213) restores q.
 * Single tile only. */
int msom_wavelet_filter(msom_t *m, double dtflt);
/* ---- energy / PV budgets, msqg/qg_energy.h (params key ediag: -1 off, 0: terms x (-psi), 1: PV terms).
 * msom_energy_tend = energy_tend (:227-241, event comp_diag :289-291): accumulates de_j1/j2/j3, de_vd,
 * de_bf from the current psi with weight dt and updates the running mean po_mft; msom_filter_de =
 * filter_de (:208-225) with po_mft = field `pm_field`; msom_reset_de zeroes the six budgets
 * (msqg/qg.c:153-159); pystep_de = the Python entry point (:296-349, msqg/qg_energy.i:31), same
 * argument order (+ handle).  msom_run drives the events and writes de_*%09d.bas (qg.c:131-160). */
int msom_energy_tend(msom_t *m, double dt);
int msom_filter_de(msom_t *m, int pm_field, double dtflt);
int msom_reset_de(msom_t *m);
int pystep_de(msom_t *m, const double *po_py, int len1, int len2, int len3, double *de_bf_py, int len4, int len5, int len6,
              double *de_vd_py, int len7, int len8, int len9, double *de_j1_py, int len10, int len11, int len12,
              double *de_j2_py, int len13, int len14, int len15, double *de_j3_py, int len16, int len17, int len18,
              double *de_ft_py, int len19, int len20, int len21, int onlyKE);

/* pieces for the parity tests: number of pyramid levels (level 0 = finest ... 1 x 1 cell), sig_lev of
 * one level [ny>>level][nx>>level], and the bare transform-scale-inverse applied to a field */
int msom_dbg_wavelet_levels(msom_t *m);
int msom_dbg_siglev(msom_t *m, int level, double *out);
int msom_dbg_wavelet_apply(msom_t *m, int field);

/* select the HIP device of the calling thread before msom_create* (one process per GPU:
 * device = LOCAL_RANK) */
int msom_set_device(int device);

/* ---- multi-GPU tiling (replaces Basilisk's MPI layer: boundary() halo exchange and
 * foreach(reduction)).  One process per GPU; the 2-D domain is a px x py grid of
 * equal tiles; rank r owns tile (r % px, r / px).  id128 is the 128-byte ncclUniqueId made by
 * msom_comm_unique_id on rank 0 and distributed by the caller (e.g. torch.distributed). */
int msom_comm_unique_id(void *id128);
msom_t *msom_create_tiled(const char *params_text, int px, int py, int rank, const void *id128);
int msom_tile_info(msom_t *m, int *px, int *py, int *ix, int *iy, int *nx_local, int *ny_local);

/* ---- debug / test hooks (used by tests/ to compare single kernels with the oracle) */
int msom_dbg_nlevels(msom_t *m);
int msom_dbg_level_dims(msom_t *m, int lev, int *nx, int *ny);
int msom_dbg_relax(msom_t *m, int lev, double *da, const double *res, int nsweeps);
int msom_dbg_residual(msom_t *m, const double *a, const double *b, double *res, double *maxres);
int msom_dbg_restrict(msom_t *m, int lev_fine, const double *fine, double *coarse);
int msom_dbg_prolong(msom_t *m, int lev_coarse, const double *coarse, double *fine);
int msom_dbg_op(msom_t *m, const char *op, int f_in, int f_out, double add, double fac);

/* ---- measurement: average HIP-event duration (events recorded on the library's own stream while the option
 * "profile" is on, i.e. inside the timed steps) of the finest-level launches named
 *   "sweep" (red + black half-sweep pair), "march2" / "march3" / "march4" (passes of K chained half-sweeps),
 *   "march_pl" (first pass of a level visit: prolongation + K half-sweeps), "march_corr" (last pass of the cycle: K half-sweeps
 *   + correction), "resid_max" (max|res|, max|u| of the corrected psi),
 *   "red_prolong" (first red half-sweep + prolongation), "resid_restrict" (pre-cycle residual + restriction),
 *   "resid_correct" (correction + residual + max|u|), "residual" (both of the former), "rhs" (fused PV tendency
 *   [+ advance] pass), "block2";
 * and a back-to-back kernel microbenchmark outside any step */
int msom_profile_read(msom_t *m, const char *kernel, double *avg_ms, long *launches);
int msom_profile_reset(msom_t *m);
/* waits for everything the handle has queued on its stream.  With option "step_sync" = 0 (default -1: on grids below 2^23 cell-layers;
 * 1: never) and "async_solve", msom_step returns while its last tendency pass is still running; every call that returns device data
 * to the host synchronises by itself, so this is for timing and for callers that share the fields' device pointers with their own streams. */
int msom_sync(msom_t *m);
int msom_bench_kernel(msom_t *m, const char *kernel, int reps, double *avg_ms);
/* one-rank RCCL communicator on the current device: grouped send/recv to self, all-reduce and
 * all-gather through the library's transport code (wiring check on a single GPU) */
int msom_dbg_rccl_selftest(void);

/* ======================================================================================
 * Vertex-grid (masked-domain) variant: qg-node/qg.h + qg_baroclinic_ms.h (nl >= 2) /
 * qg_barotropic.h (nl = 1) + nodal-poisson.h + my_vertex.h, driver qg-node/qg.c.
 * Unknowns on the (N+1)^2 vertices; field arrays are fp64 [layer][N+1][N+1]
 * (qg-node/netcdf_vertex_bas.h:253), host or device pointers.  Single GPU.
 * ====================================================================================== */
enum {
  MSOMN_PSI = 0,   /* psi       stream function                           qg-node/qg.h:130 */
  MSOMN_Q = 1,     /* q         potential vorticity (evolving)            qg.h:131         */
  MSOMN_ZETA = 2,  /* zeta      relative vorticity                qg_baroclinic_ms.h:32   */
  MSOMN_TMP = 3,   /* tmp                                                                  */
  MSOMN_PSIPG = 4, /* psi_pg    large-scale stream function                                */
  MSOMN_S2 = 5,    /* S2        N^2 before, f^2/N^2 after set_const; nl-1 layers           */
  MSOMN_TOPO = 6,  /* topo      1 layer                                                    */
  MSOMN_QFORC = 7, /* q_forcing 1 layer                                   qg.h:133         */
  MSOMN_MASK = 8,  /* mask      1 inside / 0 land and boundary, 1 layer   qg.h:134         */
  MSOMN_DQ = 9,    /* updates                                                              */
  MSOMN_QPRED = 10,/* predictor                                                            */
  MSOMN_QFORC3D = 11, /* q_forcing_3d (-DFORCING_3D, qg_baroclinic_ms.h:25,179-185): added to every layer once set */
  /* surface-QG variant (params key sqg = 1): the finished parts of qg-node/sqg_baroclinic_ms.h -- comp_stretch with the
   * surface buoyancy :77-98, idh0[0] = 1/dh[0] :502, S2 of the surface = f/N2[0] :545, tmp boundary rule :64-67,
   * laplacian(bs) in both dissipation operators :160-201.  That file stops at "TODO: STOPPED HERE" (:222) and does not
   * compile; bs is therefore a prescribed field (its tendency rhs_bs is unfinished there), comp_q / invert_q are completed
   * consistently with comp_stretch */
  MSOMN_BS = 12,   /* bs        surface buoyancy, 1 layer                                  */
  MSOMN_S2S = 13,  /* S2 of the surface: N2[0] before, f/N2[0] after set_const, 1 layer    */
  MSOMN_PSIF = 14, /* psi_f     running mean of the part the wavelet filter removes (qg_baroclinic_ms.h:30,384) */
  MSOMN_NFIELDS = 15
};
typedef struct msomn msomn_t;

/* read_params (qg-node/extra.h:83-116, key list qg.c:72-107) + init_grid + set_bc/set_vars
 * (qg.h:404-459, qg_baroclinic_ms.h:400-447): mask = 1 inside and 0 on the boundary
 * vertices, S2 = N2[l], everything else 0.  Extension: none.  NULL on error. */
msomn_t *msomn_create(const char *params_path);
msomn_t *msomn_create_str(const char *params_text);
void msomn_destroy(msomn_t *m);                                 /* trash_vars qg.h:537-544 */
/* keys: TOLERANCE NITERMAX NITERMIN (nodal-poisson.h:19-23) DT quiet stochastic seed; implementation switches (result-preserving in
 * the strict build): node_split [65] levels of >= that many vertices a side keep correction / residual / mask / S2 copies in the
 * x-parity split layout (0: off), s2_rows [1] row tables for an S2 that does not depend on x, node_pfused [1] prolongation folded into the first colour pass of the split levels, mg_coarse [32] levels of at most that
 * many cells a side in one launch, tiled_relax [0], node_march [0] (measured slower, kept for the tests); round 3: node_march_s [2049] split
 * levels of >= that many vertices a side chain up to 4 colour half-sweeps per pass (node_march_tail1 [1]: 9 half-sweeps as 4 + 4 + a colour
 * launch, 0: 4 + 3 + 2), node_tile_s [65] / node_tile_max [513] / node_tile_k [8]
 * the split levels between those sizes run up to node_tile_k colour half-sweeps per LDS-tiled launch (nl <= 4), node_rhs_fused [1] the
 * baroclinic tendency in three passes instead of the twelve loops of the reference, node_corr_fused [2] the correction of a cycle applied
 * inside the residual pass of the next (2: rows marched, 1: one thread per vertex, 0: separate passes), profile [0] */
int msomn_set_option(msomn_t *m, const char *key, double value);
/* keys: N nl L0 DT tend dtout nlevels iRd2_low bc_fac idh0_<l> idh1_<l>; NaN if unknown */
double msomn_get_param(msomn_t *m, const char *key);
int msomn_field_layers(msomn_t *m, int field);
int msomn_set_field(msomn_t *m, int field, const double *a);   /* a: [layers][N+1][N+1] */
int msomn_get_field(msomn_t *m, int field, double *a);
/* init events: layer metrics + S2 = f^2/N^2 + topo scaling (qg_baroclinic_ms.h:449-510),
 * iRd2_low (qg_barotropic.h:115-118), mask and S2 on every multigrid level, DT limits and
 * q = comp_q(psi) (set_const, qg.h:465-524).  psi, S2 (= N^2), mask, topo, psi_pg must be
 * set before; restart / input files are the driver's business (msomn_run). */
int msomn_set_const(msomn_t *m);
/* update_qg (qg.h:334-354): invert_q + rhs_pv + adjust_dt; *dt_out = new time step */
int msomn_update(msomn_t *m, int qfield, int dqfield, double dtmax, double *dt_out);
int msomn_advance(msomn_t *m, int out, int in, int dq, double dt);        /* advance_qg qg.h:291-302 */
int msomn_invert_q(msomn_t *m, int qfield, msom_mgstats *stats);           /* qg_baroclinic_ms.h:217-225 */
int msomn_comp_q(msomn_t *m, int psifield, int qfield);                    /* :199-211 / qg_barotropic.h:32-39 */
int msomn_rhs_pv(msomn_t *m, int qfield, int dqfield);                     /* :104-196 / qg_barotropic.h:16-29 */
int msomn_forcing(msomn_t *m);                                             /* event forcing, qg.c:136-145 */
/* events of iteration i (forcing if with_forcing_event), then one predictor-corrector step of run() */
int msomn_step(msomn_t *m, int with_forcing_event);
int msomn_set_tnext(msomn_t *m, double tnext);
double msomn_time(msomn_t *m);
double msomn_dt(msomn_t *m);
int msomn_iter(msomn_t *m);
int msomn_ke(msomn_t *m, double *ke);                                      /* event writestdout qg.c:171-178 */
int msomn_last_mgstats(msomn_t *m, msom_mgstats *stats);
/* measurement only (option "profile" = 1): HIP-event timing of the finest-level launches; slots "relax_fine", "relax_prolong_fine",
 * "residual", "correct", "rhs" (the whole rhs_pv chain), "coarse" (the one-launch coarse levels) */
int msomn_profile_read(msomn_t *m, const char *slot, double *avg_ms, long *launches);
int msomn_profile_reset(msomn_t *m);
int msomn_diag1d(msomn_t *m, double *out3);                                /* event write_1d_diag qg.h:361-399: ke, dissipation, forcing */
/* NetCDF-3 output / restart of vertex fields (qg-node/netcdf_vertex_bas.h:95-424): one record of
 * "psi" and "q" appended to `path`; msomn_read_nc loads variable `varname` into `field` */
int msomn_write_nc(msomn_t *m, const char *path);
int msomn_read_nc(msomn_t *m, int field, const char *path, const char *varname, int record);
/* main() + events of qg-node/qg.c: psi = noise_init (noise + sin(2 pi y / L0)) (qg.h:475-479),
 * restart.nc and input_vars_<nl>l_N<N>.nc when present in the working directory, vars.nc in
 * <workdir>/outdir_%04d/, one stdout line per iteration; nsteps_max < 0: run to tend.
 * Returns the iteration count or < 0. */
int msomn_run(msomn_t *m, const char *workdir, long nsteps_max);
/* wavelet_filter of the vertex model, qg_baroclinic_ms.h:346-400 (event filter :405-408, driven by msomn_run when dtflt > 0):
 * invert q, masked wavelet transform (qg-node/wavelet_vertex.h:10-46) of the cell average of psi scaled by sig_lev
 * (:525-552; keys Lfmax, Lfmin, fac_filt_Rd), psi_f running mean, psi -= filtered part, q = comp_q(psi) */
int msomn_wavelet_filter(msomn_t *m, double dtflt);
int msomn_dbg_wv_get(msomn_t *m, int what /* 0 sig_lev, 1 mask_c */, int level, double *out /* [n][n], n = N >> level */);
int msomn_dbg_wv_apply(msomn_t *m, const double *in, double *out /* cell fields [nl][N][N] */);
/* multigrid pieces on level arrays [layer][n_k+1][n_k+1] (level 0 = finest), for the parity tests */
int msomn_dbg_relax(msomn_t *m, int level, double *da, const double *res, int nsweeps);
int msomn_dbg_residual(msomn_t *m, const double *a, const double *b, double *res, double *maxres);
int msomn_dbg_restrict(msomn_t *m, int level_fine, const double *fine, double *coarse);
int msomn_dbg_prolong(msomn_t *m, int level_coarse, const double *coarse, double *fine);
int msomn_dbg_level_mask(msomn_t *m, int level, double *out);
int msomn_dbg_del2_zeta(msomn_t *m);
/* stochastic forcing of the vertex model (-D_STOCHASTIC: qg-node/qg_stochastic.h, qg-node/qg.h:306-320; params keys
 * amp_stoch, L_filt; option "stochastic" before msomn_set_const, option "seed" = srand).  The noise is a CELL
 * scalar (N x N), wavelet-filtered with the coefficients of the uniform length L_filt.  msomn_dbg_noise: optionally
 * set n_stoch, optionally filter it, optionally read it back; msomn_dbg_csig: sig_lev of one level. */
int msomn_dbg_noise(msomn_t *m, const double *set, int filter, double *get);
int msomn_dbg_csig(msomn_t *m, int level, double *out);

#ifdef __cplusplus
}
#endif
#endif
