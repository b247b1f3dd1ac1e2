/* msom_params.h -- plain-C part shared by the host C files (params.c, bas_io.c) and the HIP
 * translation units: parameters of params.in (msqg/qg.h:63-106) and their derived values
 * (:739-758). */
#ifndef MSOM_PARAMS_H
#define MSOM_PARAMS_H

#define MSOM_MAXARR 64 /* max entries of the Fr/dh/upg/vpg arrays in params.in */

struct Params {
  int N, Ny, nl;
  int ediag, varRo, nptr, flsrv;
  double L0, Rom, Ekb, Eks, tau0, Re, Re4, iRe, iRe4, sbc, beta, afilt, Lfmax;
  double DT, tend, dtout, dtflt, CFL;
  double Frm[MSOM_MAXARR], dhu[MSOM_MAXARR], upg[MSOM_MAXARR], vpg[MSOM_MAXARR];
  double ptr_r[MSOM_MAXARR], ptr_ir[MSOM_MAXARR], Pe[MSOM_MAXARR], iPe[MSOM_MAXARR]; /* passive tracers, qg.h:103-106 */
  double tr_stoch, itr_stoch, amp_stoch;
  double tolerance; /* extension key TOLERANCE (reference: 1e-3, msqg/qg.h:159) */
  int nitermax, nitermin;
  int mglevels; /* extension key MGLEVELS: cap on the number of multigrid levels (0 = all) */
};

#ifdef __cplusplus
extern "C" {
#endif
/* params.c: restates read_params, msqg/qg.h:668-761 */
void msom_params_defaults(struct Params *p);
int msom_params_parse_text(struct Params *p, const char *text);
int msom_params_parse_file(struct Params *p, const char *path);
void msom_params_derive(struct Params *p);
/* bas_io.c: restates input_matrixl/output_matrixl, msqg/auxiliar_input.h:24-59,101-149.
 * `a` is [layer][y][x] fp64, n x n cells per layer. */
int msom_bas_write(const char *path, const double *a, int nl, int n, double L0);
int msom_bas_read(const char *path, double *a, int nl, int n, double L0);
/* netcdf3.c: NetCDF-3 classic writer/reader replacing libnetcdf's create_nc/write_nc/read_nc
 * (newqg/netcdf_bas.h:42-244, qg-node/netcdf_vertex_bas.h:315-424).  Fields are [level][y][x] fp64. */
int msom_nc_create(const char *path, int nl, int ny, int nx, double L0, int nvars, const char *const *names);
int msom_nc_create2(const char *path, int nl, int ny, int nx, double L0, int nvars, const char *const *names, int vertex);
int msom_nc_append(const char *path, int nl, int ny, int nx, int nvars, const char *const *names, double time, const double *const *fields);
int msom_nc_read(const char *path, const char *name, int rec, int nl, int ny, int nx, double *out, double *time_out);
void msom_set_error(const char *fmt, ...);

/* parameters of the vertex-grid variant: qg-node/qg.c:72-107 (add_param list), defaults
 * qg-node/qg.h:104-127,164 and qg.c:61-66 */
struct NodeParams {
  int N, nl, flag_ms;
  int sqg; /* 1: surface-QG variant, the finished parts of qg-node/sqg_baroclinic_ms.h (N2 then has nl entries, N2[0] = surface) */
  double L0, f0, beta, nu, nu4, hEkb, gp_low, scale_topo, tau0, tau1, tf1, tf2, dy_ws, forc_mode, noise_init;
  double Lfmax, Lfmin, fac_filt_Rd, dtflt, bc_fac, DT, tend, dtout, CFL, TOLERANCE, dtdiag;
  double amp_stoch, L_filt; /* -D_STOCHASTIC keys, qg-node/qg.c:104-107 */
  double dh[MSOM_MAXARR], N2[MSOM_MAXARR];
};
/* params.c: restates read_params of qg-node/extra.h:83-116 */
void msom_node_params_defaults(struct NodeParams *p);
int msom_node_params_parse_text(struct NodeParams *p, const char *text);
int msom_node_params_parse_file(struct NodeParams *p, const char *path);
#ifdef __cplusplus
}
#endif
#endif
