// kernels_wavelet.hip -- wavelet scale filter of msom (wavelet_filter, msqg/qg.h:509-560).
//
// Basilisk's wavelet()/inverse_wavelet() (grid/multigrid-common.h; not in the reference tree,
// restated in oracle/qg_oracle.c [BASILISK RULE]) on natural-layout pyramids, all layers at once:
//   s_{k+1} = mean of the 4 children of s_k (+ boundary on every level)
//   w_k     = (s_k - bilinear(s_{k+1})) * sig_lev_k,   w_root = s_root * sig_lev_root
//   r_root  = w_root,  r_k = bilinear(r_{k+1}) + w_k    (+ boundary on every level)
// The detail coefficients are never stored: the reconstruction kernel recomputes w_k from the
// unfiltered pyramid s, so the filter costs one restriction pass down and one fused pass up.
#include "kernels.h"

#define BX 64
#define BY 4
static inline dim3 grid2d(int nx, int ny) { return dim3((nx + BX - 1) / BX, (ny + BY - 1) / BY); }

// [BASILISK RULE] restriction: mean of the 4 children, summed in foreach_child order (x outer, y inner)
__global__ void k_wv_restrict(const double *__restrict__ f, NatGeom fg, double *c, NatGeom cg, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= cg.nx || j >= cg.ny) return;
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(fg, l, 2 * j, 2 * i);
    double sum = 0.;
    sum += f[k]; sum += f[k + fg.pitch]; sum += f[k + 1]; sum += f[k + 1 + fg.pitch];
    c[nat_idx(cg, l, j, i)] = sum / 4;
  }
}
void launch_wv_restrict(hipStream_t st, const double *f, const NatGeom &fg, double *c, const NatGeom &cg, int nl) {
  hipLaunchKernelGGL(k_wv_restrict, grid2d(cg.nx, cg.ny), dim3(BX, BY), 0, st, f, fg, c, cg, nl);
}

// [BASILISK RULE] bilinear: (9 c + 3 (c[child.x] + c[0,child.y]) + c[child.x,child.y]) / 16
__device__ __forceinline__ double bilin(const double *__restrict__ c, size_t k, int cx, int cyp) {
  return (9. * c[k] + 3. * (c[k + cx] + c[k + cyp]) + c[k + cx + cyp]) / 16.;
}
// out_k = bilinear(r_{k+1}) + (s_k - bilinear(s_{k+1})) * sig_k ; out may alias s_k (each thread reads
// only its own cell of s_k)
__global__ void k_wv_recon(const double *s, const double *__restrict__ sc, const double *__restrict__ rc, const double *__restrict__ sig,
                           double *out, NatGeom fg, NatGeom cg, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= fg.nx || j >= fg.ny) return;
  const int cx = (i & 1) ? 1 : -1, cyp = (j & 1) ? cg.pitch : -cg.pitch;
  const double sg = sig[nat_idx(fg, 0, j, i)];
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(fg, l, j, i), kc = nat_idx(cg, l, j >> 1, i >> 1);
    double d = s[k];
    d -= bilin(sc, kc, cx, cyp);
    const double w = d * sg;
    double r = bilin(rc, kc, cx, cyp);
    r += w;
    out[k] = r;
  }
}
void launch_wv_recon(hipStream_t st, const double *s, const double *sc, const double *rc, const double *sig, double *out, const NatGeom &fg,
                     const NatGeom &cg, int nl) {
  hipLaunchKernelGGL(k_wv_recon, grid2d(fg.nx, fg.ny), dim3(BX, BY), 0, st, s, sc, rc, sig, out, fg, cg, nl);
}
__global__ void k_wv_root(const double *__restrict__ s, const double *__restrict__ sig, double *r, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const double sg = sig[nat_idx(g, 0, j, i)];
  for (int l = 0; l < nl; l++) r[nat_idx(g, l, j, i)] = s[nat_idx(g, l, j, i)] * sg;
}
void launch_wv_root(hipStream_t st, const double *s, const double *sig, double *r, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_wv_root, grid2d(g.nx, g.ny), dim3(BX, BY), 0, st, s, sig, r, g, nl);
}
// qof = (qof * nbar + (tmp - q) / dtflt) / (nbar + 1); restore != 0: q = tmp   (msqg/qg.h:543-555)
__global__ void k_wv_qof(double *qof, double *q, const double *__restrict__ tmp, NatGeom g, int nl, double dtflt, int nbar, int restore) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(g, l, j, i);
    qof[k] = (qof[k] * nbar + (tmp[k] - q[k]) / dtflt) / (nbar + 1);
    if (restore) q[k] = tmp[k];
  }
}
void launch_wv_qof(hipStream_t st, double *qof, double *q, const double *tmp, const NatGeom &g, int nl, double dtflt, int nbar, int restore) {
  hipLaunchKernelGGL(k_wv_qof, grid2d(g.nx, g.ny), dim3(BX, BY), 0, st, qof, q, tmp, g, nl, dtflt, nbar, restore);
}

// ---- masked transform pair of the vertex model: wavelet_mask / inverse_wavelet_mask, qg-node/wavelet_vertex.h:10-46, with the
// scaling by sig_lev in between (qg_baroclinic_ms.h:373-376):
//   w_k = ((s_k - bilinear(s_{k+1})) * mask_c_k) * sig_k,   r_k = (bilinear(r_{k+1}) + w_k) * mask_c_k
//   root: r = ((s * mask_c) * sig) * mask_c
__global__ void k_wv_recon_m(const double *s, const double *__restrict__ sc, const double *__restrict__ rc, const double *__restrict__ sig,
                             const double *__restrict__ mc, double *out, NatGeom fg, NatGeom cg, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= fg.nx || j >= fg.ny) return;
  const int cx = (i & 1) ? 1 : -1, cyp = (j & 1) ? cg.pitch : -cg.pitch;
  const double sg = sig[nat_idx(fg, 0, j, i)], mk = mc[nat_idx(fg, 0, j, i)];
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(fg, l, j, i), kc = nat_idx(cg, l, j >> 1, i >> 1);
    double d = s[k];
    d -= bilin(sc, kc, cx, cyp);
    d = d * mk;
    const double w = d * sg;
    double r = bilin(rc, kc, cx, cyp);
    r += w;
    out[k] = r * mk;
  }
}
void launch_wv_recon_m(hipStream_t st, const double *s, const double *sc, const double *rc, const double *sig, const double *mc, double *out,
                       const NatGeom &fg, const NatGeom &cg, int nl) {
  hipLaunchKernelGGL(k_wv_recon_m, grid2d(fg.nx, fg.ny), dim3(BX, BY), 0, st, s, sc, rc, sig, mc, out, fg, cg, nl);
}
__global__ void k_wv_root_m(const double *__restrict__ s, const double *__restrict__ sig, const double *__restrict__ mc, double *r, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const double sg = sig[nat_idx(g, 0, j, i)], mk = mc[nat_idx(g, 0, j, i)];
  for (int l = 0; l < nl; l++) {
    double w = s[nat_idx(g, l, j, i)] * mk;
    w = w * sg;
    r[nat_idx(g, l, j, i)] = w * mk;
  }
}
void launch_wv_root_m(hipStream_t st, const double *s, const double *sig, const double *mc, double *r, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_wv_root_m, grid2d(g.nx, g.ny), dim3(BX, BY), 0, st, s, sig, mc, r, g, nl);
}
// cell average of a vertex field (qg_baroclinic_ms.h:369-370): cell (i, j) <- vertices (i, j), (i+1, j), (i, j+1), (i+1, j+1)
__global__ void k_wv_vert2cell(const double *__restrict__ v, NatGeom vg, double *c, NatGeom cg, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= cg.nx || j >= cg.ny) return;
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(vg, l, j, i);
    c[nat_idx(cg, l, j, i)] = 0.25 * (v[k] + v[k + 1] + v[k + vg.pitch] + v[k + 1 + vg.pitch]);
  }
}
void launch_wv_vert2cell(hipStream_t st, const double *v, const NatGeom &vg, double *c, const NatGeom &cg, int nl) {
  hipLaunchKernelGGL(k_wv_vert2cell, grid2d(cg.nx, cg.ny), dim3(BX, BY), 0, st, v, vg, c, cg, nl);
}
// :381-386: psi_loc = vertex average of the filtered cell field (ghost cells: dirichlet), psi_f running mean, psi = (psi - psi_loc) mask
__global__ void k_wv_vertex_update(double *psi, double *psif, const double *__restrict__ c, const double *__restrict__ mask, NatGeom vg, NatGeom cg, int nl,
                                   double dtflt, int nbar, int update_mean) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= vg.nx || j >= vg.ny) return;
  const double mk = mask[nat_idx(vg, 0, j, i)];
  for (int l = 0; l < nl; l++) {
    const size_t kc = nat_idx(cg, l, j, i), k = nat_idx(vg, l, j, i);
    const double psi_loc = 0.25 * (c[kc] + c[kc - 1] + c[kc - cg.pitch] + c[kc - 1 - cg.pitch]);
    if (update_mean) psif[k] = (psif[k] * nbar + psi_loc / dtflt) / (nbar + 1);
    psi[k] = (psi[k] - psi_loc) * mk;
  }
}
void launch_wv_vertex_update(hipStream_t st, double *psi, double *psif, const double *c, const double *mask, const NatGeom &vg, const NatGeom &cg, int nl,
                             double dtflt, int nbar, int update_mean) {
  hipLaunchKernelGGL(k_wv_vertex_update, grid2d(vg.nx, vg.ny), dim3(BX, BY), 0, st, psi, psif, c, mask, vg, cg, nl, dtflt, nbar, update_mean);
}
