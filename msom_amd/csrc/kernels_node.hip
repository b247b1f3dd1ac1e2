// kernels_node.hip -- HIP kernels of the VERTEX-grid (masked) QG variant of the reference:
// qg-node/qg.h, qg-node/qg_baroclinic_ms.h (-DLAYERS=1, nl >= 2), qg-node/qg_barotropic.h
// (nl = 1), qg-node/nodal-poisson.h, qg-node/my_vertex.h.
//
// Unknowns on the (N+1)^2 vertices x = i D, y = j D (i, j = 0..N), stored in the natural padded
// layout with nx = ny = N + 1 (pad cells stay 0: they are the never-defined values outside the
// boundary vertices).  One thread per vertex walks the layers.  Everything is multiplied by
// `mask` (1 inside, 0 on boundary vertices and land; fractional on coarse multigrid levels).
// Boundary vertices carry the boundary conditions (psi = psi_bc, q = zeta =
// 2 bc_fac / D^2 (psi_first_interior - psi_bc), qg-node/qg.h:197-214).
// Smoother: the reference's relax_baroclinic is a lexicographic Gauss-Seidel over all
// vertices with a Thomas solve per column (qg_baroclinic_ms.h:228-291); here the same column
// solve in red-black order ((i + j) even first), as for the cell-centred model.
#include "kernels.h"
#include "rhs_inl.h"

#ifdef MSOM_STRICT
#define DIVC(x, c, rc) ((x) / (c))
#else
#define DIVC(x, c, rc) ((x) * (rc))
#endif
#define BX 64
#define BY 4
static inline dim3 grid2d(int nx, int ny) { return dim3((nx + BX - 1) / BX, (ny + BY - 1) / BY); }
static inline dim3 block2d() { return dim3(BX, BY); }
static inline dim3 grid_capped(int nx, int ny) {
  dim3 g = grid2d(nx, ny);
  const unsigned cap = 2048 / g.x > 0 ? 2048 / g.x : 1;
  if (g.y > cap) g.y = cap;
  return g;
}
#define VTX(g, i, j) const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y; if (i >= (g).nx || j >= (g).ny) return

// Wide multigrid levels keep the correction, the residual and copies of mask / S2 in an x-parity split layout (round 2): a row is
// stored as [even-i half | odd-i half], hp = (pitch - 2 XP) / 2 doubles each, so the vertices of one red-black colour are contiguous
// in every row and a colour pass moves the bytes it uses (natural layout: whole lines, half used).  i = -1 and i = n + 1 land in
// the pad of a half.  sp = 0: natural layout.
__host__ __device__ __forceinline__ size_t gidx(const NatGeom &g, int sp, int l, int j, int i) {
  const size_t row = (size_t)l * g.ls + (size_t)(j + MSOM_YP) * g.pitch + MSOM_XP;
  return sp ? row + (size_t)((i & 1) * ((g.pitch - 2 * MSOM_XP) >> 1) + (i >> 1)) : row + i;
}

__device__ __forceinline__ double wave_max_n(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64)); return v; }
__device__ __forceinline__ double wave_sum_n(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }

#define LAPN(p, c, pitch) ((p)[(c) + 1] + (p)[(c)-1] + (p)[(c) + (pitch)] + (p)[(c) - (pitch)] - 4 * (p)[c])
// +J(p,q), qg-node/qg.h:178-188
__device__ __forceinline__ double jacn(const double *__restrict__ p, const double *__restrict__ q, size_t c, int pitch, double D12, double rD12) {
#define P(a, b) p[c + (a) + (ptrdiff_t)(b)*pitch]
#define Q(a, b) q[c + (a) + (ptrdiff_t)(b)*pitch]
  const double s = (P(1, 0) - P(-1, 0)) * (Q(0, 1) - Q(0, -1)) + (P(0, -1) - P(0, 1)) * (Q(1, 0) - Q(-1, 0)) + P(1, 0) * (Q(1, 1) - Q(1, -1)) -
                   P(-1, 0) * (Q(-1, 1) - Q(-1, -1)) - P(0, 1) * (Q(1, 1) - Q(-1, 1)) + P(0, -1) * (Q(1, -1) - Q(-1, -1)) +
                   Q(0, 1) * (P(1, 1) - P(-1, 1)) - Q(0, -1) * (P(1, -1) - P(-1, -1)) - Q(1, 0) * (P(1, 1) - P(1, -1)) +
                   Q(-1, 0) * (P(-1, 1) - P(-1, -1));
#undef P
#undef Q
  return DIVC(s, D12, rD12);
}

// ---------------------------------------------------------------- boundary vertices
// f_bnd = c * (g_first_interior - (use_g_bnd ? g_bnd : gbc)); x sides first, then y sides (corners
// end up with the y rule).  One thread per boundary vertex and layer.
__global__ void k_n_bnd_from(double *f, const double *g, NatGeom ge, int nl, double c, int use_g_bnd, double gbc) {
  const int n = ge.nx - 1, per = 4 * n;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  int i, j, ii, jj;  // boundary vertex and its first interior neighbour
  if (r < 2 * (n + 1)) {  // y sides, all i (corners included -> y rule)
    i = ii = r >> 1;
    if (r & 1) { j = n; jj = n - 1; } else { j = 0; jj = 1; }
  } else {  // x sides without the corners
    const int q = r - 2 * (n + 1);
    j = jj = 1 + (q >> 1);
    if (q & 1) { i = n; ii = n - 1; } else { i = 0; ii = 1; }
  }
  const size_t b = nat_idx(ge, l, j, i);
  // corner: the y rule reads g at (i, jj), which is an x-side boundary vertex; for use_g_bnd its
  // value must be the one the x pass gave it -- g is a different field here (zeta for tmp), so ok
  f[b] = c * (g[nat_idx(ge, l, jj, ii)] - (use_g_bnd ? g[b] : gbc));
}
void launch_n_bnd_from(hipStream_t st, double *f, const double *g, const NatGeom &ge, int nl, double c, int use_g_bnd, double gbc) {
  const int n = 4 * (ge.nx - 1) * nl;
  hipLaunchKernelGGL(k_n_bnd_from, dim3((n + 255) / 256), dim3(256), 0, st, f, g, ge, nl, c, use_g_bnd, gbc);
}
__global__ void k_n_bnd_const(double *f, NatGeom ge, int nl, double v, int sp) {
  const int n = ge.nx - 1, per = 4 * n;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  int i, j;
  if (r < 2 * (n + 1)) { i = r >> 1; j = (r & 1) ? n : 0; } else { const int q = r - 2 * (n + 1); j = 1 + (q >> 1); i = (q & 1) ? n : 0; }
  f[gidx(ge, sp, l, j, i)] = v;
}
void launch_n_bnd_const(hipStream_t st, double *f, const NatGeom &ge, int nl, double v, int sp) {
  const int n = 4 * (ge.nx - 1) * nl;
  hipLaunchKernelGGL(k_n_bnd_const, dim3((n + 255) / 256), dim3(256), 0, st, f, ge, nl, v, sp);
}

// ---------------------------------------------------------------- pointwise / stencil operators
__global__ void k_n_mul_mask(double *a, double *b, const double *__restrict__ mk, NatGeom g, int nl) {
  VTX(g, i, j);
  const size_t c0 = nat_idx(g, 0, j, i);
  const double m = mk[c0];
  for (int l = 0; l < nl; l++) { a[c0 + l * g.ls] *= m; if (b) b[c0 + l * g.ls] *= m; }
}
void launch_n_mul_mask(hipStream_t st, double *a, double *b, const double *mk, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_n_mul_mask, grid2d(g.nx, g.ny), block2d(), 0, st, a, b, mk, g, nl);
}
__global__ void k_n_del2(const double *__restrict__ in, double *out, NatGeom g, int nl, double add, double fac, double D2, double rD2) {
  VTX(g, i, j);
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) {
    const double lap = DIVC(LAPN(in, c, g.pitch), D2, rD2);
    out[c] = add == 0. ? fac * lap : add * out[c] + fac * lap;
  }
}
void launch_n_del2(hipStream_t st, const double *in, double *out, const NatGeom &g, int nl, double add, double fac, double D) {
  hipLaunchKernelGGL(k_n_del2, grid2d(g.nx, g.ny), block2d(), 0, st, in, out, g, nl, add, fac, D * D, 1. / (D * D));
}
// comp_stretch qg_baroclinic_ms.h:79-100
__global__ void k_n_stretch(const double *__restrict__ in, double *out, const double *__restrict__ S2, NatGeom g, int nl, double add, double fac,
                            LayerCoef lc) {
  VTX(g, i, j);
  const size_t c0 = nat_idx(g, 0, j, i);
  double pm = 0., pc = in[c0], pp = 0., s0 = 0., s1 = 0.;
  for (int l = 0; l < nl; l++) {
    const size_t c = c0 + (size_t)l * g.ls;
    if (l < nl - 1) { pp = in[c + g.ls]; s1 = S2[c]; }
    double v;
    if (l == 0) v = fac * s1 * (pp - pc) * lc.idh1[l];
    else if (l < nl - 1) v = fac * (s0 * (pm - pc) * lc.idh0[l] + s1 * (pp - pc) * lc.idh1[l]);
    else v = fac * s0 * (pm - pc) * lc.idh0[l];
    out[c] = add == 0. ? v : add * out[c] + v;
    pm = pc; pc = pp; s0 = s1;
  }
}
void launch_n_stretch(hipStream_t st, const double *in, double *out, const double *S2, const NatGeom &g, int nl, double add, double fac,
                      const LayerCoef &lc) {
  hipLaunchKernelGGL(k_n_stretch, grid2d(g.nx, g.ny), block2d(), 0, st, in, out, S2, g, nl, add, fac, lc);
}

// comp_stretch(psi, bs, stretch, add, fac) of the surface-QG variant, sqg_baroclinic_ms.h:77-98.  S2 there has nl layers
// with layer 0 at the surface: its S2[] / S2[0,0,1] of layer l are S2S / S2[0] for l = 0 and S2[l-1] / S2[l] below.
__global__ void k_n_stretch_sqg(const double *__restrict__ in, const double *__restrict__ bs, const double *__restrict__ S2S, double *out,
                                const double *__restrict__ S2, NatGeom g, int nl, double add, double fac, LayerCoef lc) {
  VTX(g, i, j);
  const size_t c0 = nat_idx(g, 0, j, i);
  double pm = 0., pc = in[c0], pp = 0., s0 = 0., s1 = 0.;
  for (int l = 0; l < nl; l++) {
    const size_t c = c0 + (size_t)l * g.ls;
    if (l < nl - 1) { pp = in[c + g.ls]; s1 = S2[c]; }
    double v;
    if (l == 0) v = fac * (S2S[c0] * bs[c0] * lc.idh0[0] - s1 * (pc - pp) * lc.idh1[l]);
    else if (l < nl - 1) v = fac * (s0 * (pm - pc) * lc.idh0[l] - s1 * (pc - pp) * lc.idh1[l]);
    else v = fac * (-s0 * (pc - pm)) * lc.idh0[l];
    out[c] = add == 0. ? v : add * out[c] + v;
    pm = pc; pc = pp; s0 = s1;
  }
}
void launch_n_stretch_sqg(hipStream_t st, const double *in, const double *bs, const double *S2S, double *out, const double *S2, const NatGeom &g, int nl,
                          double add, double fac, const LayerCoef &lc) {
  hipLaunchKernelGGL(k_n_stretch_sqg, grid2d(g.nx, g.ny), block2d(), 0, st, in, bs, S2S, out, S2, g, nl, add, fac, lc);
}
// del2_bs of sqg_baroclinic_ms.h:160-168: laplacian(bs) on the inner vertices, neumann(0) on the four sides (the boundary
// vertex takes the first interior value, x sides first, so a corner takes the diagonal neighbour)
__global__ void k_n_lap_bs(const double *__restrict__ bs, double *out, NatGeom g, double D) {
  VTX(g, i, j);
  const int n = g.nx - 1;
  const int ii = i == 0 ? 1 : (i == n ? n - 1 : i), jj = j == 0 ? 1 : (j == n ? n - 1 : j);
  const size_t c = nat_idx(g, 0, jj, ii);
  const double D2 = D * D;
  out[nat_idx(g, 0, j, i)] = DIVC(LAPN(bs, c, g.pitch), D2, 1. / D2);
}
void launch_n_lap_bs(hipStream_t st, const double *bs, double *out, const NatGeom &g, double D) {
  hipLaunchKernelGGL(k_n_lap_bs, grid2d(g.nx, g.ny), block2d(), 0, st, bs, out, g, D);
}
// right-hand side of the surface-QG inversion: q with the known surface term of the top layer removed
__global__ void k_n_sqg_rhs(const double *__restrict__ q, const double *__restrict__ S2S, const double *__restrict__ bs, double *qeff, NatGeom g, int nl,
                            double idh00) {
  VTX(g, i, j);
  size_t k = nat_idx(g, 0, j, i);
  qeff[k] = q[k] - S2S[k] * bs[k] * idh00;
  for (int l = 1; l < nl; l++) { k += g.ls; qeff[k] = q[k]; }
}
void launch_n_sqg_rhs(hipStream_t st, const double *q, const double *S2S, const double *bs, double *qeff, const NatGeom &g, int nl, double idh00) {
  hipLaunchKernelGGL(k_n_sqg_rhs, grid2d(g.nx, g.ny), block2d(), 0, st, q, S2S, bs, qeff, g, nl, idh00);
}

// advective part of rhs_pv_baroclinic, qg_baroclinic_ms.h:116-155
struct NRhsArgs {
  const double *psi, *zeta, *pg, *S2, *topo;
  double *dq;
  NatGeom g;
  int nl, have_pg, have_topo;
  double D, beta, drag, f0, dhb;  // drag = hEkb*f0/(2*dh_b)
  LayerCoef lc;
};
__global__ void k_n_rhs_main(NRhsArgs a) {
  VTX(a.g, i, j);
  const int nl = a.nl, pitch = a.g.pitch;
  const size_t ls = a.g.ls;
  const double D12 = 12. * a.D * a.D, rD12 = 1. / D12, D2x = 2 * a.D, rD2x = 1. / D2x;
  size_t c = nat_idx(a.g, 0, j, i);
  const size_t c0 = c;
  double ju = 0., jd = 0.;
  for (int l = 0; l < nl; l++, c += ls) {
    ju = -jd;
    if (l < nl - 1) {
      jd = jacn(a.psi, a.psi + ls, c, pitch, D12, rD12);
      if (a.have_pg) jd = jd + jacn(a.pg, a.psi + ls, c, pitch, D12, rD12) + jacn(a.psi, a.pg + ls, c, pitch, D12, rD12);
    }
    double d = -jacn(a.psi, a.zeta, c, pitch, D12, rD12);
    if (a.have_pg) d = d - jacn(a.pg, a.zeta, c, pitch, D12, rD12);
    if (l < nl - 1) d = d - a.S2[c] * jd * a.lc.idh1[l];
    if (l > 0) d = d - a.S2[c - ls] * ju * a.lc.idh0[l];
    d = d - DIVC(a.beta * (a.psi[c + 1] - a.psi[c - 1]), D2x, rD2x);
    if (l == nl - 1) {  // bottom friction and topography, :150
      double e = -a.drag * a.zeta[c];
      if (a.have_topo) {
        const double jt = jacn(a.psi + (size_t)l * ls, a.topo, c0, pitch, D12, rD12);
#ifdef MSOM_STRICT
        e = e - jt * a.f0 / a.dhb;
#else
        e = e - jt * (a.f0 / a.dhb);
#endif
      }
      d += e;
    }
    a.dq[c] = d;
  }
}
void launch_n_rhs_main(hipStream_t st, const double *psi, const double *zeta, const double *pg, const double *S2, const double *topo, double *dq,
                       const NatGeom &g, int nl, int have_pg, int have_topo, double D, double beta, double drag, double f0, double dhb, const LayerCoef &lc) {
  NRhsArgs a;
  a.psi = psi; a.zeta = zeta; a.pg = pg; a.S2 = S2; a.topo = topo; a.dq = dq; a.g = g; a.nl = nl; a.have_pg = have_pg; a.have_topo = have_topo;
  a.D = D; a.beta = beta; a.drag = drag; a.f0 = f0; a.dhb = dhb; a.lc = lc;
  hipLaunchKernelGGL(k_n_rhs_main, grid2d(g.nx, g.ny), block2d(), 0, st, a);
}
// dq += c * x   /   dq_0 += qf   /   rhs_pv_barotropic (qg_barotropic.h:16-29)
__global__ void k_n_axpy(double *dq, const double *__restrict__ x, NatGeom g, int nl, double c) {
  VTX(g, i, j);
  size_t k = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, k += g.ls) dq[k] += c * x[k];
}
void launch_n_axpy(hipStream_t st, double *dq, const double *x, const NatGeom &g, int nl, double c) {
  hipLaunchKernelGGL(k_n_axpy, grid2d(g.nx, g.ny), block2d(), 0, st, dq, x, g, nl, c);
}
__global__ void k_n_add2d(double *dq, const double *__restrict__ qf, NatGeom g) {
  VTX(g, i, j);
  const size_t k = nat_idx(g, 0, j, i);
  dq[k] += qf[k];
}
void launch_n_add2d(hipStream_t st, double *dq, const double *qf, const NatGeom &g) { hipLaunchKernelGGL(k_n_add2d, grid2d(g.nx, g.ny), block2d(), 0, st, dq, qf, g); }
// ---- the baroclinic tendency in three passes (round 3) instead of the twelve launches of the reference's loop sequence.  The
// expressions and their order per vertex are those of the kernels above, so the strict build gives the same bits.
// Pass 1: q *= mask; psi_out = psi * mask (second buffer: the neighbours of other threads read the unmasked one and mask it
// themselves); zeta = laplacian of the masked psi inside, the set_bc_ms rule c (psi_first_interior - psi_bc) on the sides
// (k_n_mul_mask + k_n_del2 + k_n_bnd_from).
__global__ void k_n_rhs_pre(double *q, const double *__restrict__ psi, double *__restrict__ psi_out, double *__restrict__ zeta, const double *__restrict__ mk,
                            NatGeom g, int nl, double D2, double rD2, double bc, double gbc) {
  VTX(g, i, j);
  const int n = g.nx - 1, pitch = g.pitch;
  const size_t c0 = nat_idx(g, 0, j, i);
  const double m = mk[c0];
  const bool ys = j == 0 || j == n, xs = i == 0 || i == n;
  // the vertex whose masked psi the side rule reads (y sides first: a corner takes the y rule)
  const ptrdiff_t off = ys ? (j == 0 ? pitch : -pitch) : (i == 0 ? 1 : -1);
  const double me = mk[c0 + 1], mw = mk[c0 - 1], mn = mk[c0 + pitch], ms = mk[c0 - pitch];
  size_t c = c0;
  for (int l = 0; l < nl; l++, c += g.ls) {
    q[c] *= m;
    const double pc = psi[c] * m;
    psi_out[c] = pc;
    if (ys || xs) zeta[c] = bc * (psi[c + off] * mk[c0 + off] - gbc);
    else {
      const double lap = DIVC(psi[c + 1] * me + psi[c - 1] * mw + psi[c + pitch] * mn + psi[c - pitch] * ms - 4 * pc, D2, rD2);
      zeta[c] = 1. * lap;
    }
  }
}
void launch_n_rhs_pre(hipStream_t st, double *q, const double *psi, double *psi_out, double *zeta, const double *mk, const NatGeom &g, int nl, double D,
                      double bc, double gbc) {
  hipLaunchKernelGGL(k_n_rhs_pre, grid2d(g.nx, g.ny), block2d(), 0, st, q, psi, psi_out, zeta, mk, g, nl, D * D, 1. / (D * D), bc, gbc);
}
// Pass 2: out = laplacian(in) inside, c (in_first_interior - (use_bnd ? in_boundary : gbc)) on the sides (k_n_del2 + k_n_bnd_from)
__global__ void k_n_del2_bnd(const double *__restrict__ in, double *__restrict__ out, NatGeom g, int nl, double D2, double rD2, double bc, int use_bnd, double gbc) {
  VTX(g, i, j);
  const int n = g.nx - 1, pitch = g.pitch;
  const bool ys = j == 0 || j == n, xs = i == 0 || i == n;
  const ptrdiff_t off = ys ? (j == 0 ? pitch : -pitch) : (i == 0 ? 1 : -1);
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) {
    if (ys || xs) out[c] = bc * (in[c + off] - (use_bnd ? in[c] : gbc));
    else out[c] = 1. * DIVC(LAPN(in, c, pitch), D2, rD2);
  }
}
void launch_n_del2_bnd(hipStream_t st, const double *in, double *out, const NatGeom &g, int nl, double D, double bc, int use_bnd, double gbc) {
  hipLaunchKernelGGL(k_n_del2_bnd, grid2d(g.nx, g.ny), block2d(), 0, st, in, out, g, nl, D * D, 1. / (D * D), bc, use_bnd, gbc);
}
// Pass 3: k_n_rhs_main, + nu stretch(zeta) (k_n_stretch / k_n_stretch_sqg), + nu tmp (k_n_axpy), - nu4 stretch(tmp), - nu4 laplacian(tmp)
// (k_n_del2 with add = 1), + surface forcing (k_n_add2d), + 3-d forcing, times the mask (k_n_mul_mask) -- one thread per vertex, the
// layers in sequence as in every one of those kernels, the partial sums in a register instead of dq.
struct NRhsTailArgs {
  const double *tmp, *qf, *qf3d, *mk, *d2bs, *S2S;
  double nu, mnu4;
  int sqg;
};
__device__ __forceinline__ double n_stretch_val(int sqg, int l, int nl, double fac, double pm, double pc, double pp, double s0, double s1, double sb,
                                                const LayerCoef &lc) {
  if (sqg) {
    if (l == 0) return fac * (sb * lc.idh0[0] - s1 * (pc - pp) * lc.idh1[l]);
    if (l < nl - 1) return fac * (s0 * (pm - pc) * lc.idh0[l] - s1 * (pc - pp) * lc.idh1[l]);
    return fac * (-s0 * (pc - pm)) * lc.idh0[l];
  }
  if (l == 0) return fac * s1 * (pp - pc) * lc.idh1[l];
  if (l < nl - 1) return fac * (s0 * (pm - pc) * lc.idh0[l] + s1 * (pp - pc) * lc.idh1[l]);
  return fac * s0 * (pm - pc) * lc.idh0[l];
}
// 3 x 3 neighbourhoods in registers: every value of psi and zeta is loaded once per vertex and layer (the psi window of layer
// l + 1 becomes that of layer l), jacn's expression on them
__device__ __forceinline__ void n_ld9(const double *__restrict__ f, size_t c, int pitch, double (&w)[9]) {
#pragma unroll
  for (int b = -1; b <= 1; b++)
#pragma unroll
    for (int x = -1; x <= 1; x++) w[(b + 1) * 3 + x + 1] = f[c + x + (ptrdiff_t)b * pitch];
}
__device__ __forceinline__ double n_jacw(const double (&p)[9], const double (&q)[9], double D12, double rD12) {
#define P(a, b) p[((b) + 1) * 3 + (a) + 1]
#define Q(a, b) q[((b) + 1) * 3 + (a) + 1]
  const double s = (P(1, 0) - P(-1, 0)) * (Q(0, 1) - Q(0, -1)) + (P(0, -1) - P(0, 1)) * (Q(1, 0) - Q(-1, 0)) + P(1, 0) * (Q(1, 1) - Q(1, -1)) -
                   P(-1, 0) * (Q(-1, 1) - Q(-1, -1)) - P(0, 1) * (Q(1, 1) - Q(-1, 1)) + P(0, -1) * (Q(1, -1) - Q(-1, -1)) +
                   Q(0, 1) * (P(1, 1) - P(-1, 1)) - Q(0, -1) * (P(1, -1) - P(-1, -1)) - Q(1, 0) * (P(1, 1) - P(1, -1)) +
                   Q(-1, 0) * (P(-1, 1) - P(-1, -1));
#undef P
#undef Q
  return DIVC(s, D12, rD12);
}
// PG / TOPO: psi_pg / topo have been set (all-zero fields add +-0 to every term: skipped)
template <bool PG, bool TOPO>
__global__ void k_n_rhs_all(NRhsArgs a, NRhsTailArgs t) {
  VTX(a.g, i, j);
  const int nl = a.nl, pitch = a.g.pitch;
  const size_t ls = a.g.ls;
  const double D12 = 12. * a.D * a.D, rD12 = 1. / D12, D2x = 2 * a.D, rD2x = 1. / D2x, D2 = a.D * a.D, rD2 = 1. / D2;
  size_t c = nat_idx(a.g, 0, j, i);
  const size_t c0 = c;
  const double m = t.mk[c0];
  const double sb = t.sqg ? t.S2S[c0] * t.d2bs[c0] : 0.;
  double ju = 0., jd = 0.;
  double zm = 0., zp = 0., tm = 0., tc = t.tmp[c0], tp = 0., s0 = 0., s1 = 0.;
  double P0[9], P1[9], Z[9];
  n_ld9(a.psi, c0, pitch, P0);
  for (int l = 0; l < nl; l++, c += ls) {
    ju = -jd;
    if (l < nl - 1) {
      n_ld9(a.psi, c + ls, pitch, P1);
      jd = n_jacw(P0, P1, D12, rD12);
      if (PG) jd = jd + jacn(a.pg, a.psi + ls, c, pitch, D12, rD12) + jacn(a.psi, a.pg + ls, c, pitch, D12, rD12);
      zp = a.zeta[c + ls]; tp = t.tmp[c + ls]; s1 = a.S2[c];
    }
    n_ld9(a.zeta, c, pitch, Z);
    const double zc = Z[4];
    double d = -n_jacw(P0, Z, D12, rD12);
    if (PG) d = d - jacn(a.pg, a.zeta, c, pitch, D12, rD12);
    if (l < nl - 1) d = d - s1 * jd * a.lc.idh1[l];
    if (l > 0) d = d - s0 * ju * a.lc.idh0[l];
    d = d - DIVC(a.beta * (P0[5] - P0[3]), D2x, rD2x);
    if (l == nl - 1) {  // bottom friction and topography, :150
      double e = -a.drag * zc;
      if (TOPO) {
        double T[9];
        n_ld9(a.topo, c0, pitch, T);
        const double jt = n_jacw(P0, T, D12, rD12);
#ifdef MSOM_STRICT
        e = e - jt * a.f0 / a.dhb;
#else
        e = e - jt * (a.f0 / a.dhb);
#endif
      }
      d += e;
    }
    d = 1. * d + n_stretch_val(t.sqg, l, nl, t.nu, zm, zc, zp, s0, s1, sb, a.lc);
    d += t.nu * tc;
    d = 1. * d + n_stretch_val(t.sqg, l, nl, t.mnu4, tm, tc, tp, s0, s1, sb, a.lc);
    d = 1. * d + t.mnu4 * DIVC(t.tmp[c + 1] + t.tmp[c - 1] + t.tmp[c + pitch] + t.tmp[c - pitch] - 4 * tc, D2, rD2);
    if (l == 0) d += t.qf[c0];
    if (t.qf3d) d += 1. * t.qf3d[c];
    d *= m;
    a.dq[c] = d;
    zm = zc; tm = tc; tc = tp; s0 = s1;
#pragma unroll
    for (int k = 0; k < 9; k++) P0[k] = P1[k];
  }
}
void launch_n_rhs_all(hipStream_t st, const double *psi, const double *zeta, const double *tmp, const double *pg, const double *S2, const double *topo,
                      const double *qf, const double *qf3d, const double *mk, const double *d2bs, const double *S2S, double *dq, const NatGeom &g, int nl,
                      double D, double beta, double drag, double f0, double dhb, double nu, double mnu4, const LayerCoef &lc, int have_pg, int have_topo) {
  NRhsArgs a;
  a.psi = psi; a.zeta = zeta; a.pg = pg; a.S2 = S2; a.topo = topo; a.dq = dq; a.g = g; a.nl = nl; a.have_pg = have_pg; a.have_topo = have_topo;
  a.D = D; a.beta = beta; a.drag = drag; a.f0 = f0; a.dhb = dhb; a.lc = lc;
  NRhsTailArgs t;
  t.tmp = tmp; t.qf = qf; t.qf3d = qf3d; t.mk = mk; t.d2bs = d2bs; t.S2S = S2S; t.nu = nu; t.mnu4 = mnu4; t.sqg = d2bs != nullptr;
  auto k = have_pg ? (have_topo ? k_n_rhs_all<true, true> : k_n_rhs_all<true, false>) : (have_topo ? k_n_rhs_all<false, true> : k_n_rhs_all<false, false>);
  hipLaunchKernelGGL(k, grid2d(g.nx, g.ny), block2d(), 0, st, a, t);
}
__global__ void k_n_rhs_barotropic(const double *__restrict__ psi, const double *__restrict__ q, const double *__restrict__ qf, double *dq, NatGeom g,
                                   double D, double beta, double drag, double nu) {
  VTX(g, i, j);
  const size_t c = nat_idx(g, 0, j, i);
  const double D12 = 12. * D * D, D2 = D * D, D2x = 2 * D;
  dq[c] = -jacn(psi, q, c, g.pitch, D12, 1. / D12) - DIVC(beta * (psi[c + 1] - psi[c - 1]), D2x, 1. / D2x) - drag * q[c] + qf[c] +
          nu * DIVC(LAPN(q, c, g.pitch), D2, 1. / D2);
}
void launch_n_rhs_barotropic(hipStream_t st, const double *psi, const double *q, const double *qf, double *dq, const NatGeom &g, double D, double beta,
                             double drag, double nu) {
  hipLaunchKernelGGL(k_n_rhs_barotropic, grid2d(g.nx, g.ny), block2d(), 0, st, psi, q, qf, dq, g, D, beta, drag, nu);
}
// q = lap(psi) - iRd2 psi (nl = 1)   qg_barotropic.h:32-39
__global__ void k_n_helm(const double *__restrict__ psi, double *q, NatGeom g, double D2, double rD2, double iRd2) {
  VTX(g, i, j);
  const size_t c = nat_idx(g, 0, j, i);
  q[c] = DIVC(LAPN(psi, c, g.pitch), D2, rD2) - iRd2 * psi[c];
}
void launch_n_helm(hipStream_t st, const double *psi, double *q, const NatGeom &g, double D, double iRd2) {
  hipLaunchKernelGGL(k_n_helm, grid2d(g.nx, g.ny), block2d(), 0, st, psi, q, g, D * D, 1. / (D * D), iRd2);
}
__global__ void k_n_rowfill(double *f, const double *__restrict__ row, NatGeom g) {
  VTX(g, i, j);
  f[nat_idx(g, 0, j, i)] = row[j];
}
void launch_n_rowfill(hipStream_t st, double *f, const double *row, const NatGeom &g) { hipLaunchKernelGGL(k_n_rowfill, grid2d(g.nx, g.ny), block2d(), 0, st, f, row, g); }

// ---------------------------------------------------------------- vertex multigrid
struct NRelaxArgs {
  double *a;
  const double *b, *mk, *S2;
  NatGeom g;
  int color;
  double sqD, iRd2;
  LayerCoef lc;
  // S2 independent of x (always so in the reference: S2 = f(y)^2 / N2[l], qg_baroclinic_ms.h:471-476, 501-505): S2row[l * g.ny + j]
  // replaces the field read (a third of the bytes of a colour pass at nl = 3)
  const double *S2row = nullptr;
};
// one colour of relax_baroclinic (qg_baroclinic_ms.h:228-291) / relax_barotropic (qg_barotropic.h:57-76)
// on the interior vertices; boundary vertices of the correction stay 0 (homogeneous psi BC).
// column solve of one vertex: ew[l] = a_E + a_W, ns[l] = a_N + a_S of layer l; c = index of the vertex in layer 0
// bv[l] = b of layer l, m = mask, sv[l] = S2 of layer l (l < NL - 1) at the vertex
template <int NL>
__device__ __forceinline__ void n_col_solve_vals(const NRelaxArgs &p, const double (&bv)[NL], double m, const double (&sv)[NL], const double (&ew)[NL],
                                                 const double (&ns)[NL], double (&x)[NL]) {
  const double sq = p.sqD;
  if (NL == 1) {
    double d = -(-p.iRd2) * sq, v = -bv[0] * sq;
    v += ew[0] * m; d += 2.;
    v += ns[0] * m; d += 2.;
    x[0] = v / d;
    return;
  }
  double t0[NL], t1[NL], t2[NL], rhs[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    rhs[l] = -sq * bv[l] * m;
    t0[l] = l == 0 ? 0. : (l < NL - 1 ? -sq * sv[l - 1] * p.lc.idh0[l] * m : -sq * sv[l - 1] * p.lc.idh0[l]);  // bottom t0 not masked, :267
    t2[l] = l < NL - 1 ? -sq * sv[l] * p.lc.idh1[l] * m : 0.;
    t1[l] = l == 0 ? -t2[l] : (l < NL - 1 ? -t0[l] - t2[l] : -t0[l]);
    rhs[l] += ew[l] * m; t1[l] += 2;
    rhs[l] += ns[l] * m; t1[l] += 2;
  }
#ifdef MSOM_STRICT
#pragma unroll
  for (int l = 1; l < NL; l++) { t1[l] -= t0[l] * t2[l - 1] / t1[l - 1]; rhs[l] -= t0[l] * rhs[l - 1] / t1[l - 1]; }
  x[NL - 1] = rhs[NL - 1] / t1[NL - 1];
#pragma unroll
  for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - t2[l] * x[l + 1]) / t1[l];
#else
  // product build: one reciprocal per layer instead of three divisions (the smoother is bound by the fp64
  // division sequences, not by HBM)
  double r[NL];
  r[0] = 1. / t1[0];
#pragma unroll
  for (int l = 1; l < NL; l++) {
    const double w = t0[l] * r[l - 1];
    t1[l] -= w * t2[l - 1];
    rhs[l] -= w * rhs[l - 1];
    r[l] = 1. / t1[l];
  }
  x[NL - 1] = rhs[NL - 1] * r[NL - 1];
#pragma unroll
  for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - t2[l] * x[l + 1]) * r[l];
#endif
}
template <int NL>
__device__ __forceinline__ void n_col_solve(const NRelaxArgs &p, size_t c, const double (&ew)[NL], const double (&ns)[NL], double (&x)[NL], int j = -1) {
  const size_t ls = p.g.ls;
  double bv[NL], sv[NL];
  const bool rowS = p.S2row != nullptr && j >= 0;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    bv[l] = p.b[c + l * ls];
    sv[l] = (NL > 1 && l < NL - 1) ? (rowS ? p.S2row[l * p.g.ny + j] : p.S2[c + l * ls]) : 0.;
  }
  n_col_solve_vals<NL>(p, bv, p.mk[c], sv, ew, ns, x);
}
template <int NL>
__device__ __forceinline__ void n_relax_pt(const NRelaxArgs &p, int i, int j) {
  const int n = p.g.nx - 1;
  if (i >= n || j >= n) return;
  const int pitch = p.g.pitch;
  const size_t ls = p.g.ls, c = nat_idx(p.g, 0, j, i);
  double ew[NL], ns[NL], x[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const size_t k = c + l * ls;
    ew[l] = p.a[k + 1] + p.a[k - 1];
    ns[l] = p.a[k + pitch] + p.a[k - pitch];
  }
  n_col_solve<NL>(p, c, ew, ns, x, j);
#pragma unroll
  for (int l = 0; l < NL; l++) p.a[c + l * ls] = x[l];
}
template <int NL>
__global__ void __launch_bounds__(BX *BY) k_n_relax(NRelaxArgs p) {
  const int j = 1 + blockIdx.y * BY + threadIdx.y;
  const int i = 1 + 2 * (blockIdx.x * BX + threadIdx.x) + ((j + p.color + 1) & 1);  // (i + j) & 1 == color
  n_relax_pt<NL>(p, i, j);
}
// the same colour pass with a, b, mask and S2 in the split layout: own colour and both x neighbours are contiguous runs
template <int NL>
__global__ void __launch_bounds__(BX *BY) k_n_relax_s(NRelaxArgs p) {
  const int j = 1 + blockIdx.y * BY + threadIdx.y;
  const int i = 1 + 2 * (blockIdx.x * BX + threadIdx.x) + ((j + p.color + 1) & 1);
  const int n = p.g.nx - 1;
  if (i >= n || j >= n) return;
  const int pitch = p.g.pitch;
  const size_t ls = p.g.ls, c = gidx(p.g, 1, 0, j, i), e = gidx(p.g, 1, 0, j, i + 1), w = gidx(p.g, 1, 0, j, i - 1);
  double ew[NL], ns[NL], x[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    ew[l] = p.a[e + l * ls] + p.a[w + l * ls];
    ns[l] = p.a[c + l * ls + pitch] + p.a[c + l * ls - pitch];
  }
  n_col_solve<NL>(p, c, ew, ns, x, j);
#pragma unroll
  for (int l = 0; l < NL; l++) p.a[c + l * ls] = x[l];
}
// Prolongation folded into the first colour pass of a split level (round 2): refine_vert (my_vertex.h:82-105) + boundary_level,
// then colour 0 of the smoother.  In red-black order the first pass reads only the prolongated values of the OTHER colour, and a
// vertex of the other colour has one odd and one even coordinate: its value is the mean of two coarse vertices (the four-vertex
// and the injected cases are colour-0 vertices, whose prolongated values nothing reads).  One thread per colour-0 vertex (i + j
// even, boundary included): it forms its four neighbours from 4 - 5 coarse values, relaxes (interior) or writes 0 (boundary), and
// stores its east neighbour (and the west one at i = 1), so that every vertex of the level is written exactly once -- what the
// prolongation launch (w written, then half of it read back) and the first pass did before.  Same expressions, same values.
template <int NL>
__global__ void __launch_bounds__(BX *BY) k_n_relax_prolong_s(NRelaxArgs p, const double *__restrict__ co, NatGeom cg, int csp) {
  const int j = blockIdx.y * BY + threadIdx.y;
  const int i = 2 * (blockIdx.x * BX + threadIdx.x) + (j & 1);
  const int n = p.g.nx - 1;
  if (i > n || j > n) return;
  const size_t ls = p.g.ls, cls = cg.ls;
  const bool odd = j & 1;  // (odd, odd) vertex: centre of coarse cell (I, J); (even, even): coarse vertex (I, J)
  const int I = i >> 1, J = j >> 1;
  // coarse vertices used: centre C = (I, J); (even, even): E (I+1, J), W (I-1, J), N (I, J+1), S (I, J-1);
  // (odd, odd): the other three corners of the cell
  const size_t kC = gidx(cg, csp, 0, J, I);
  const size_t kE = gidx(cg, csp, 0, J, min(I + 1, cg.nx - 1)), kN = gidx(cg, csp, 0, min(J + 1, cg.ny - 1), I);
  const size_t kW = gidx(cg, csp, 0, J, max(I - 1, 0)), kS = gidx(cg, csp, 0, max(J - 1, 0), I);
  const size_t kNE = gidx(cg, csp, 0, min(J + 1, cg.ny - 1), min(I + 1, cg.nx - 1));
  // boundary vertices of the level are 0 (boundary_level)
  const bool bE = i + 1 >= n || j == 0 || j == n, bW = i - 1 <= 0 || j == 0 || j == n;
  const bool bN = j + 1 >= n || i == 0 || i == n, bS = j - 1 <= 0 || i == 0 || i == n;
  const bool interior = i > 0 && j > 0 && i < n && j < n;
  double aE[NL], aW[NL], ew[NL], ns[NL], x[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const double cC = co[kC + l * cls];
    double e, w, nn, ss;
    if (!odd) {
      e = (cC + co[kE + l * cls]) / 2.; w = (co[kW + l * cls] + cC) / 2.;
      nn = (cC + co[kN + l * cls]) / 2.; ss = (co[kS + l * cls] + cC) / 2.;
    } else {
      const double cE = co[kE + l * cls], cN = co[kN + l * cls], cNE = co[kNE + l * cls];
      e = (cE + cNE) / 2.; w = (cC + cN) / 2.; nn = (cN + cNE) / 2.; ss = (cC + cE) / 2.;
    }
    aE[l] = bE ? 0. : e; aW[l] = bW ? 0. : w;
    ew[l] = aE[l] + aW[l];
    ns[l] = (bN ? 0. : nn) + (bS ? 0. : ss);
  }
  const size_t c = gidx(p.g, 1, 0, j, i);
  if (interior) n_col_solve<NL>(p, c, ew, ns, x, j);
#pragma unroll
  for (int l = 0; l < NL; l++) p.a[c + l * ls] = interior ? x[l] : 0.;
  if (i + 1 <= n) {
    const size_t e = gidx(p.g, 1, 0, j, i + 1);
#pragma unroll
    for (int l = 0; l < NL; l++) p.a[e + l * ls] = aE[l];
  }
  if (i == 1) {
    const size_t w = gidx(p.g, 1, 0, j, 0);
#pragma unroll
    for (int l = 0; l < NL; l++) p.a[w + l * ls] = aW[l];
  }
}
void launch_n_relax_prolong(hipStream_t st, double *a, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl, double D,
                            double iRd2, const LayerCoef &lc, const double *S2row, const double *coarse, const NatGeom &cg, int csp) {
  NRelaxArgs p;
  p.a = a; p.b = b; p.mk = mk; p.S2 = S2; p.g = g; p.color = 0; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc; p.S2row = S2row;
  const int n = g.nx - 1;
  dim3 gr = grid2d(n / 2 + 1, n + 1);
  switch (nl) {
    case 1: hipLaunchKernelGGL(k_n_relax_prolong_s<1>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 2: hipLaunchKernelGGL(k_n_relax_prolong_s<2>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 3: hipLaunchKernelGGL(k_n_relax_prolong_s<3>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 4: hipLaunchKernelGGL(k_n_relax_prolong_s<4>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 5: hipLaunchKernelGGL(k_n_relax_prolong_s<5>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 6: hipLaunchKernelGGL(k_n_relax_prolong_s<6>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 7: hipLaunchKernelGGL(k_n_relax_prolong_s<7>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    case 8: hipLaunchKernelGGL(k_n_relax_prolong_s<8>, gr, block2d(), 0, st, p, coarse, cg, csp); break;
    default: break;
  }
}
// ---- K consecutive colour half-sweeps of the vertex smoother in ONE pass (round 2), the marching scheme of kernels_march.hip in
// the natural layout.  A colour pass of k_n_relax touches every cache line of `a` twice (it reads the other colour and writes
// its own: half of each line both ways) and half of every line of b / mask / S2: ~1.9 w of line traffic per pass, 10 passes per
// level visit (5 sweeps).  Here a wavefront (= a workgroup) marches up a strip of 64 columns, lane = column, both colours of
// a row in the same registers: half-sweep s runs one row behind half-sweep s - 1 and reads its 3-row window of the values
// after half-sweep s - 1 (N / S: own lane, E / W: whole-wave DPP shifts); lanes of the other colour carry their value
// through.  Only `a` lives in windows (K x 3 x NL doubles); b, mask and S2 are read by the column solve where it needs them
// (second and later uses of a row hit L2).  Out of place (a_in -> a_out), K halo lanes per side and K halo rows per chunk
// re-compute the cone of dependence.  Boundary vertices (and everything outside) are never relaxed and carry their value.
// Same column solve (n_col_solve) on the same inputs => bit-identical to the pass-per-colour path.
template <int NL, int K>
__global__ void __launch_bounds__(64) k_n_relax_march(NRelaxArgs p, const double *__restrict__ a_in, int H) {
  constexpr int OW = 64 - 2 * K;
  const int lane = threadIdx.x;
  const int n = p.g.nx - 1;                      // vertices 0 .. n
  const int gi = (int)blockIdx.x * OW - K + lane;
  const int y0 = blockIdx.y * H, y1 = min(n + 1, y0 + H);
  const size_t ls = p.g.ls;
  const int gic = min(max(gi, -1), n + 1);       // pads beyond the grid: read as they are, never used by valid vertices
  const bool own = lane >= K && lane < 64 - K && gi <= n;
  const bool col_in = gi >= 1 && gi <= n - 1;    // an inner column
  auto rowp = [&](const double *f, int r) -> const double * { return f + nat_idx(p.g, 0, min(max(r, -1), n + 1), gic); };
  double W[K][3][NL];
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int l = 0; l < NL; l++) W[s][q][l] = 0.;
  // software pipeline: the rows of step t + 1 (input row t + 2; b, mask, S2 of row t + 1 for the first half-sweep) are
  // requested at the top of step t into `n*` and move to the working registers at the top of step t + 1
  double na[NL], nb[NL], ns2[NL], nm;
  {
    const double *pa = rowp(a_in, y0 - K), *pb = rowp(a_in, y0 - K + 1);
#pragma unroll
    for (int l = 0; l < NL; l++) { W[0][1][l] = pa[l * ls]; W[0][2][l] = pb[l * ls]; }
    const int t = y0 - K + 1;
    const double *qa = rowp(a_in, t + 1), *qb = rowp(p.b, t);
    const double *qs = NL > 1 ? rowp(p.S2, t) : qb;
#pragma unroll
    for (int l = 0; l < NL; l++) { na[l] = qa[l * ls]; nb[l] = qb[l * ls]; ns2[l] = (NL > 1 && l < NL - 1) ? qs[l * ls] : 0.; }
    nm = *rowp(p.mk, t);
  }
  for (int t = y0 - K + 1; t <= y1 + K - 2; t++) {
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
      for (int l = 0; l < NL; l++) { W[s][0][l] = W[s][1][l]; W[s][1][l] = W[s][2][l]; }
    double cb[NL], cs2[NL];
    const double cm = nm;
#pragma unroll
    for (int l = 0; l < NL; l++) { W[0][2][l] = na[l]; cb[l] = nb[l]; cs2[l] = ns2[l]; }
    {
      const double *qa = rowp(a_in, t + 2), *qb = rowp(p.b, t + 1);
      const double *qs = NL > 1 ? rowp(p.S2, t + 1) : qb;
#pragma unroll
      for (int l = 0; l < NL; l++) { na[l] = qa[l * ls]; nb[l] = qb[l * ls]; ns2[l] = (NL > 1 && l < NL - 1) ? qs[l * ls] : 0.; }
      nm = *rowp(p.mk, t + 1);
    }
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = t - (s - 1);                 // row of half-sweep s
      const int col = (p.color + s - 1) & 1;     // its colour
      double x[NL], ew[NL], ns[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double a = W[s - 1][1][l];
        x[l] = a;                                // carried unless this vertex is relaxed now
        ew[l] = lane_above(a) + lane_below(a);   // a_E + a_W (k_n_relax: p.a[k + 1] + p.a[k - 1])
        ns[l] = W[s - 1][2][l] + W[s - 1][0][l]; // a_N + a_S
      }
      if (r >= 1 && r <= n - 1 && col_in && ((gi + r) & 1) == col) {
        double xn[NL];
        if (s == 1) n_col_solve_vals<NL>(p, cb, cm, cs2, ew, ns, xn);           // row t: prefetched one step ago
        else n_col_solve<NL>(p, nat_idx(p.g, 0, r, gi), ew, ns, xn);            // rows t - 1 ...: second use, L2
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = xn[l];
      }
      if (s < K) {
#pragma unroll
        for (int l = 0; l < NL; l++) W[s][2][l] = x[l];
      } else if (r >= y0 && r < y1 && own) {
        double *dst = p.a + nat_idx(p.g, 0, r, gi);
#pragma unroll
        for (int l = 0; l < NL; l++) dst[l * ls] = x[l];
      }
    }
  }
}
int g_node_march_rows = 0;  // tuning knob (option node_march_rows)
template <int NL>
static int n_relax_march_dispatch(hipStream_t st, const NRelaxArgs &p, const double *a_in, int K) {
  const int n1 = p.g.nx;
  // chunk height: about 2 rounds of the resident wavefronts, never below 16 rows (2 K of them are re-computed)
  auto launch = [&](auto kern, int ow) {
    const int strips = (n1 + ow - 1) / ow;
    extern int g_node_march_rows;
    int chunks = 2 * 256 * 8 / strips;
    if (chunks < 1) chunks = 1;
    int H = (n1 + chunks - 1) / chunks;
    if (H < 16) H = 16;
    if (g_node_march_rows > 0) H = g_node_march_rows;
    hipLaunchKernelGGL(kern, dim3(strips, (n1 + H - 1) / H), dim3(64), 0, st, p, a_in, H);
  };
  switch (K) {
    case 2: launch(k_n_relax_march<NL, 2>, 60); return 0;
    case 3: launch(k_n_relax_march<NL, 3>, 58); return 0;
    case 4: launch(k_n_relax_march<NL, 4>, 56); return 0;
  }
  return -1;
}
// ---------------------------------------------------------------------------------------------------------------------
// k_n_relax_march_s (round 3): K colour half-sweeps of a SPLIT level chained in one pass, the scheme of kernels_march.hip on
// the vertex grid.  A colour pass of the finest level already runs at 0.63 of the HBM peak on its own bytes (bench.py,
// C5_vertex_sqg.kernels.relax_fine), so what is left is the bytes: 10 passes of 1.67 w per cycle against ~2.7 w per 4 chained
// half-sweeps.  Lane k holds the vertex pair (2 k, 2 k + 1) of every row -- one vertex of each colour, ALL lanes busy in every
// half-sweep (the natural-layout kernel above kept both colours per lane and idled half of them); windows, whole-wave shifts
// and the one-row lag between half-sweeps as in k_relax_march; the column solve is n_col_solve_vals, the very function of the
// colour-per-launch kernels (mask, unmasked bottom sub-diagonal, one reciprocal per layer in the product build) => bit-identical.
// Residual and mask of the updated colour are read ONCE and wait in register delay lines; S2 comes from the row tables (the
// pass is only used when S2 does not depend on x, which is always so in the reference).  Boundary vertices (i, j = 0, n) are
// never relaxed: the correction's boundary value is 0 before and after every colour.  Rows are software-prefetched one step ahead.
template <int NL, int K>
__global__ void __launch_bounds__(64) k_n_relax_march_s(NRelaxArgs p, const double *__restrict__ a_in, int H, int partial) {
  constexpr int HL = (K + 1) / 2, OW = 64 - 2 * HL;
  constexpr int D1 = K >= 3 ? 3 : 1, D2 = K >= 4 ? 3 : 1;
  const int lane = threadIdx.x;
  const int n = p.g.nx - 1;                                  // vertices 0 .. n, n even
  const int kx = (int)blockIdx.x * OW - HL + lane;           // the lane's pair: vertices 2 kx, 2 kx + 1
  const int y0 = blockIdx.y * H, y1 = min(n + 1, y0 + H);
  const int hp = (p.g.pitch - 2 * MSOM_XP) >> 1;
  const size_t ls = p.g.ls;
  const int kxc = min(max(kx, -2), (n >> 1) + 2);            // spare slots of the halves beyond the grid: read, never used
  const bool own = lane >= HL && lane < 64 - HL && 2 * kx <= n;
  auto off = [&](int half, int r) -> size_t { return (size_t)(min(max(r, -1), n + 1) + MSOM_YP) * p.g.pitch + MSOM_XP + (size_t)half * hp + kxc; };
  const int c1 = p.color, c0 = 1 - c1;                       // colours of half-sweep 1 and of the input values
  double W[K][3][NL];
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int l = 0; l < NL; l++) W[s][q][l] = 0.;
  double R1[D1][NL], R2[D2][NL], M1[D1], M2[D2];
#pragma unroll
  for (int d = 0; d < D1; d++) { M1[d] = 0.;
#pragma unroll
    for (int l = 0; l < NL; l++) R1[d][l] = 0.; }
#pragma unroll
  for (int d = 0; d < D2; d++) { M2[d] = 0.;
#pragma unroll
    for (int l = 0; l < NL; l++) R2[d][l] = 0.; }
  // software pipeline, PF steps deep: what step t + PF consumes is requested at the top of step t (a marching wavefront pays a
  // memory round trip per step otherwise: 86 us per pass at 2049^2 x 3 with one step of lead whatever the chunk height);
  // the loop is unrolled by PF so that the slot of a step is a compile-time index
  constexpr int PF = NL <= 4 ? 3 : 2;
  double na[PF][NL], nr1[PF][NL], nr2[PF][NL], nm1[PF], nm2[PF];
  const int tb = y1 + K - 2;
  auto request = [&](int t, int u) {
    const size_t oa = off((t + 1 + c0) & 1, t + 1), o1 = off((t + c1) & 1, t), o2 = off((t - 1 + c0) & 1, t - 1);
#pragma unroll
    for (int l = 0; l < NL; l++) { na[u][l] = a_in[oa + l * ls]; nr1[u][l] = p.b[o1 + l * ls]; nr2[u][l] = p.b[o2 + l * ls]; }
    nm1[u] = p.mk[o1]; nm2[u] = p.mk[o2];
  };
  {
    const int ta = y0 - K + 1;
    const size_t oA = off((ta - 1 + c0) & 1, ta - 1), oB = off((ta + c0) & 1, ta);
#pragma unroll
    for (int l = 0; l < NL; l++) { W[0][1][l] = a_in[oA + l * ls]; W[0][2][l] = a_in[oB + l * ls]; }
#pragma unroll
    for (int u = 0; u < PF; u++) request(ta + u, u);
  }
  for (int tt = y0 - K + 1; tt <= tb; tt += PF)
#pragma unroll
  for (int u = 0; u < PF; u++) {
    const int t = tt + u;
    if (t > tb) break;
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
      for (int l = 0; l < NL; l++) { W[s][0][l] = W[s][1][l]; W[s][1][l] = W[s][2][l]; }
#pragma unroll
    for (int d = D1 - 1; d > 0; d--) { M1[d] = M1[d - 1];
#pragma unroll
      for (int l = 0; l < NL; l++) R1[d][l] = R1[d - 1][l]; }
#pragma unroll
    for (int d = D2 - 1; d > 0; d--) { M2[d] = M2[d - 1];
#pragma unroll
      for (int l = 0; l < NL; l++) R2[d][l] = R2[d - 1][l]; }
#pragma unroll
    for (int l = 0; l < NL; l++) { W[0][2][l] = na[u][l]; R1[0][l] = nr1[u][l]; R2[0][l] = nr2[u][l]; }
    M1[0] = nm1[u]; M2[0] = nm2[u];
    if (t + PF <= tb) request(t + PF, u);
    const int px = (t + c1) & 1;                             // x parity of the vertices updated in this step (all K half-sweeps)
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = t - (s - 1);                             // row of half-sweep s
      const int i = 2 * kx + px;
      double ew[NL], ns[NL], bv[NL], sv[NL], x[NL];
      // residual / mask of this half-sweep's row: loaded in this step (slot 0) or two steps ago (slot 2 of the delay line)
      const int i1 = (D1 == 3 && s == 3) ? 2 : 0, i2 = (D2 == 3 && s == 4) ? 2 : 0;
      const double mv = (s & 1) ? M1[i1] : M2[i2];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double a = W[s - 1][1][l];
        ew[l] = px ? lane_above(a) + a : a + lane_below(a);  // a_E + a_W
        ns[l] = W[s - 1][2][l] + W[s - 1][0][l];             // a_N + a_S
        bv[l] = (s & 1) ? R1[i1][l] : R2[i2][l];
        sv[l] = (NL > 1 && l < NL - 1) ? p.S2row[l * p.g.ny + min(max(r, 0), n)] : 0.;
      }
      n_col_solve_vals<NL>(p, bv, mv, sv, ew, ns, x);
      const bool inner = r >= 1 && r <= n - 1 && i >= 1 && i <= n - 1;
#pragma unroll
      for (int l = 0; l < NL; l++) x[l] = inner ? x[l] : 0.;  // boundary vertices (and everything beyond the grid) stay 0
      if (s < K) {
#pragma unroll
        for (int l = 0; l < NL; l++) W[s][2][l] = x[l];
      }
      if (s >= K - 1 && r >= y0 && r < y1 && own && i <= n && (s == K || !partial)) {
        double *dst = p.a + off(px, r);
#pragma unroll
        for (int l = 0; l < NL; l++) dst[l * ls] = x[l];
      }
    }
  }
}
template <int NL>
static int n_relax_march_s_dispatch(hipStream_t st, const NRelaxArgs &p, const double *a_in, int K, int partial) {
  extern int g_node_march_rows;
  const int n = p.g.nx - 1, H = g_node_march_rows > 0 ? g_node_march_rows : 12;
  auto launch = [&](auto kern, int ow) { hipLaunchKernelGGL(kern, dim3(((n >> 1) + 1 + ow - 1) / ow, (n + 1 + H - 1) / H), dim3(64), 0, st, p, a_in, H, partial); };
  switch (K) {
    case 2: launch(k_n_relax_march_s<NL, 2>, 62); return 0;
    case 3: launch(k_n_relax_march_s<NL, 3>, 60); return 0;
    case 4: launch(k_n_relax_march_s<NL, 4>, 60); return 0;
  }
  return -1;
}
// K (2..4) half-sweeps of a split level starting with colour `color`, a_in -> a_out (both in the split layout g); S2 by row tables
int launch_n_relax_march_s(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const NatGeom &g, int nl, int color, int K,
                           double D, double iRd2, const LayerCoef &lc, const double *S2row, int partial) {
  NRelaxArgs p;
  p.a = a_out; p.b = b; p.mk = mk; p.S2 = nullptr; p.g = g; p.color = color; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc; p.S2row = S2row;
  if (nl > 1 && !S2row) return -1;
  switch (nl) {
    case 1: return n_relax_march_s_dispatch<1>(st, p, a_in, K, partial);
    case 2: return n_relax_march_s_dispatch<2>(st, p, a_in, K, partial);
    case 3: return n_relax_march_s_dispatch<3>(st, p, a_in, K, partial);
    case 4: return n_relax_march_s_dispatch<4>(st, p, a_in, K, partial);
    case 5: return n_relax_march_s_dispatch<5>(st, p, a_in, K, partial);
    case 6: return n_relax_march_s_dispatch<6>(st, p, a_in, K, partial);
  }
  return -1;
}

// K (2..4) half-sweeps starting with colour `color`, a_in -> a_out; returns -1 if K is not supported
int launch_n_relax_march(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl,
                         int color, int K, double D, double iRd2, const LayerCoef &lc) {
  NRelaxArgs p;
  p.a = a_out; p.b = b; p.mk = mk; p.S2 = S2; p.g = g; p.color = color; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc;
  switch (nl) {
    case 1: return n_relax_march_dispatch<1>(st, p, a_in, K);
    case 2: return n_relax_march_dispatch<2>(st, p, a_in, K);
    case 3: return n_relax_march_dispatch<3>(st, p, a_in, K);
    case 4: return n_relax_march_dispatch<4>(st, p, a_in, K);
    case 5: return n_relax_march_dispatch<5>(st, p, a_in, K);
    case 6: return n_relax_march_dispatch<6>(st, p, a_in, K);
    case 7: return n_relax_march_dispatch<7>(st, p, a_in, K);
    case 8: return n_relax_march_dispatch<8>(st, p, a_in, K);
  }
  return -1;
}

// K (2..8) colour half-sweeps of a split level in one launch, for the levels whose colour passes are launch-bound (65 .. 1025 vertices
// a side at 2049^2: ~4.5 us per pass whatever the size; round 3).  One 1024-thread workgroup per tile of T = 40 - 2 H vertices a side
// (H = K rounded up to even: 32^2 for K <= 4, 24^2 for K = 8): the tile of the correction with a halo of H goes to LDS (40^2 vertices
// per layer), half-sweep h updates the vertices of its colour inside tile +- (K - 1 - h) -- a vertex is
// updated only where its four neighbours carry the values of the previous half-sweep, so every update is the one of the
// colour-per-launch pass, bit for bit (the argument of k_n_relax_tile) -- then the tile is stored (out of place: a_in -> p.a).
// A thread owns the vertex pair (2k, 2k + 1) of one row, one vertex of each colour: residual, mask and the S2 row values stay in
// its registers for all K half-sweeps; the column solve is n_col_solve_vals.
#define NTS_L 40
#define NTS_KMAX 8
#define NTS_NT 1024
template <int NL>
__global__ void __launch_bounds__(NTS_NT) k_n_relax_tile_s(NRelaxArgs p, const double *__restrict__ a_in, int K) {
  __shared__ double A[NL][NTS_L][NTS_L + 1];
  const int tid = threadIdx.x, n = p.g.nx - 1;
  const int NTS_H = (K + 1) & ~1, NTS_T = NTS_L - 2 * NTS_H;
  const int x0 = blockIdx.x * NTS_T - NTS_H, y0 = blockIdx.y * NTS_T - NTS_H;   // first vertex of the LDS region (x0 even)
  for (int t = tid; t < NL * NTS_L * NTS_L; t += NTS_NT) {
    const int l = t / (NTS_L * NTS_L), r = (t / NTS_L) % NTS_L, c = t % NTS_L;
    const int gi = x0 + c, gj = y0 + r;
    A[l][r][c] = (gi >= 0 && gi <= n && gj >= 0 && gj <= n) ? a_in[gidx(p.g, 1, l, gj, gi)] : 0.;
  }
  // this thread's vertex pair: row r, columns 2k, 2k + 1 of the region
  const int r = tid / (NTS_L / 2), k2 = 2 * (tid % (NTS_L / 2));
  const int gj = y0 + r;
  const bool rowok = tid < NTS_L * (NTS_L / 2) && gj >= 1 && gj <= n - 1;
  double bv[2][NL], mv[2], sv[NL];
#pragma unroll
  for (int v = 0; v < 2; v++) {
    const int gi = x0 + k2 + v;
    const bool ok = rowok && gi >= 1 && gi <= n - 1;
    const size_t c = gidx(p.g, 1, 0, ok ? gj : 1, ok ? gi : 1);
    mv[v] = p.mk[c];
#pragma unroll
    for (int l = 0; l < NL; l++) bv[v][l] = p.b[c + l * p.g.ls];
  }
#pragma unroll
  for (int l = 0; l < NL; l++) sv[l] = (NL > 1 && l < NL - 1) ? p.S2row[l * p.g.ny + (rowok ? gj : 1)] : 0.;
  __syncthreads();
  for (int h = 0; h < K; h++) {
    const int e = K - 1 - h, col = (p.color + h) & 1;
    const int v = (gj + col) & 1;           // x0 + k2 is even: the vertex of colour col of the pair
    const int lc = k2 + v, gi = x0 + lc;
    const bool act = rowok && gi >= 1 && gi <= n - 1 && r >= NTS_H - e && r < NTS_H + NTS_T + e && lc >= NTS_H - e && lc < NTS_H + NTS_T + e;
    if (act) {
      double ew[NL], ns[NL], x[NL], b1[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        ew[l] = A[l][r][lc + 1] + A[l][r][lc - 1];
        ns[l] = A[l][r + 1][lc] + A[l][r - 1][lc];
        b1[l] = v ? bv[1][l] : bv[0][l];
      }
      n_col_solve_vals<NL>(p, b1, v ? mv[1] : mv[0], sv, ew, ns, x);
#pragma unroll
      for (int l = 0; l < NL; l++) A[l][r][lc] = x[l];
    }
    __syncthreads();
  }
  for (int t = tid; t < NL * NTS_T * NTS_T; t += NTS_NT) {
    const int l = t / (NTS_T * NTS_T), rr = (t / NTS_T) % NTS_T, c = t % NTS_T;
    const int gi = x0 + NTS_H + c, gj2 = y0 + NTS_H + rr;
    if (gi <= n && gj2 <= n) p.a[gidx(p.g, 1, l, gj2, gi)] = A[l][rr + NTS_H][c + NTS_H];
  }
}
// K (2..4) half-sweeps of a split level starting with colour `color`, a_in -> a_out; -1: not available for this nl / S2 (use the colour passes)
int launch_n_relax_tile_s(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const NatGeom &g, int nl, int color, int K,
                          double D, double iRd2, const LayerCoef &lc, const double *S2row) {
  if (K < 2 || K > NTS_KMAX || (nl > 1 && !S2row)) return -1;
  const int NTS_T = NTS_L - 2 * ((K + 1) & ~1);
  NRelaxArgs p;
  p.a = a_out; p.b = b; p.mk = mk; p.S2 = nullptr; p.g = g; p.color = color; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc; p.S2row = S2row;
  const dim3 grid((g.nx + NTS_T - 1) / NTS_T, (g.ny + NTS_T - 1) / NTS_T);
  switch (nl) {
    case 1: hipLaunchKernelGGL(k_n_relax_tile_s<1>, grid, dim3(NTS_NT), 0, st, p, a_in, K); return 0;
    case 2: hipLaunchKernelGGL(k_n_relax_tile_s<2>, grid, dim3(NTS_NT), 0, st, p, a_in, K); return 0;
    case 3: hipLaunchKernelGGL(k_n_relax_tile_s<3>, grid, dim3(NTS_NT), 0, st, p, a_in, K); return 0;
    case 4: hipLaunchKernelGGL(k_n_relax_tile_s<4>, grid, dim3(NTS_NT), 0, st, p, a_in, K); return 0;
  }
  return -1;
}

// NS full red-black sweeps in ONE pass over HBM, out of place (a_in -> a_out; neighbouring workgroups read each
// other's tiles): a 64 x TH tile of the correction with a 2 NS halo goes to LDS, half-sweep h updates the
// cells of colour h & 1 inside the region tile +- (2 NS - 1 - h) -- a cell is updated only when its four
// neighbours carry the values of the previous half-sweep, so every update equals the one of the
// kernel-per-colour sweep bit for bit -- and the tile is stored.  In the natural (colour-interleaved)
// layout a colour pass moves whole cache lines of which it uses half; this pass reads and writes every
// line once per NS sweeps.
template <int NL, int NS, int TH>
__global__ void __launch_bounds__(BX *BY) k_n_relax_tile(NRelaxArgs p, const double *a_in) {
  constexpr int TW = 64, H = 2 * NS, LW = TW + 2 * H, LH = TH + 2 * H;
  __shared__ double A[NL][LH][LW];
  const int tid = threadIdx.y * BX + threadIdx.x;
  const int n = p.g.nx - 1, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  for (int t = tid; t < NL * LH * LW; t += BX * BY) {
    const int l = t / (LH * LW), r = (t / LW) % LH, c = t % LW;
    const int gi = x0 - H + c, gj = y0 - H + r;
    A[l][r][c] = (gi >= 0 && gi <= n && gj >= 0 && gj <= n) ? a_in[nat_idx(p.g, l, gj, gi)] : 0.;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2 * NS; h++) {
    const int e = H - 1 - h, RW = TW + 2 * e, RH = TH + 2 * e, half = RW / 2, col = h & 1;
    for (int t = tid; t < RH * half; t += BX * BY) {
      const int r = t / half, k = t % half;
      const int gj = y0 - e + r, xs = x0 - e;
      const int gi = xs + ((col + xs + gj) & 1) + 2 * k;
      if (gi < 1 || gi >= n || gj < 1 || gj >= n) continue;  // boundary vertices and beyond: never relaxed
      const int lr = gj - (y0 - H), lc = gi - (x0 - H);
      double ew[NL], ns[NL], x[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        ew[l] = A[l][lr][lc + 1] + A[l][lr][lc - 1];
        ns[l] = A[l][lr + 1][lc] + A[l][lr - 1][lc];
      }
      n_col_solve<NL>(p, nat_idx(p.g, 0, gj, gi), ew, ns, x);
#pragma unroll
      for (int l = 0; l < NL; l++) A[l][lr][lc] = x[l];
    }
    __syncthreads();
  }
  for (int t = tid; t < NL * TH * TW; t += BX * BY) {
    const int l = t / (TH * TW), r = (t / TW) % TH, c = t % TW;
    const int gi = x0 + c, gj = y0 + r;
    if (gi <= n && gj <= n) p.a[nat_idx(p.g, l, gj, gi)] = A[l][r + H][c + H];
  }
}
template <int NL>
static void n_relax_tile_dispatch(hipStream_t st, const NRelaxArgs &p, const double *a_in, int ns) {
  const int n1 = p.g.nx;
  if constexpr (NL <= 6) {
    if (ns == 2) {
      hipLaunchKernelGGL((k_n_relax_tile<NL, 2, 32>), dim3((n1 + 63) / 64, (n1 + 31) / 32), block2d(), 0, st, p, a_in);
      return;
    }
  }
  hipLaunchKernelGGL((k_n_relax_tile<NL, 1, 16>), dim3((n1 + 63) / 64, (n1 + 15) / 16), block2d(), 0, st, p, a_in);
}
// returns the number of sweeps done by this pass (ns asked; 1 when the 2-sweep tile does not fit in LDS)
int launch_n_relax_tile(hipStream_t st, const double *a_in, double *a_out, const double *b, const double *mk, const double *S2, const NatGeom &g,
                        int nl, int ns, double D, double iRd2, const LayerCoef &lc) {
  NRelaxArgs p;
  p.a = a_out; p.b = b; p.mk = mk; p.S2 = S2; p.g = g; p.color = 0; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc;
  if (ns == 2 && nl > 6) ns = 1;
  switch (nl) {
    case 1: n_relax_tile_dispatch<1>(st, p, a_in, ns); break;
    case 2: n_relax_tile_dispatch<2>(st, p, a_in, ns); break;
    case 3: n_relax_tile_dispatch<3>(st, p, a_in, ns); break;
    case 4: n_relax_tile_dispatch<4>(st, p, a_in, ns); break;
    case 5: n_relax_tile_dispatch<5>(st, p, a_in, ns); break;
    case 6: n_relax_tile_dispatch<6>(st, p, a_in, ns); break;
    case 7: n_relax_tile_dispatch<7>(st, p, a_in, ns); break;
    case 8: n_relax_tile_dispatch<8>(st, p, a_in, ns); break;
    default: break;
  }
  return ns;
}
void launch_n_relax(hipStream_t st, double *a, const double *b, const double *mk, const double *S2, const NatGeom &g, int nl, int color, double D,
                    double iRd2, const LayerCoef &lc, int sp, const double *S2row) {
  NRelaxArgs p;
  p.a = a; p.b = b; p.mk = mk; p.S2 = S2; p.g = g; p.color = color; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc; p.S2row = S2row;
  const int n = g.nx - 1;
  dim3 gr = grid2d((n + 1) / 2, n - 1);
  if (sp) {
    switch (nl) {
      case 1: hipLaunchKernelGGL(k_n_relax_s<1>, gr, block2d(), 0, st, p); break;
      case 2: hipLaunchKernelGGL(k_n_relax_s<2>, gr, block2d(), 0, st, p); break;
      case 3: hipLaunchKernelGGL(k_n_relax_s<3>, gr, block2d(), 0, st, p); break;
      case 4: hipLaunchKernelGGL(k_n_relax_s<4>, gr, block2d(), 0, st, p); break;
      case 5: hipLaunchKernelGGL(k_n_relax_s<5>, gr, block2d(), 0, st, p); break;
      case 6: hipLaunchKernelGGL(k_n_relax_s<6>, gr, block2d(), 0, st, p); break;
      case 7: hipLaunchKernelGGL(k_n_relax_s<7>, gr, block2d(), 0, st, p); break;
      case 8: hipLaunchKernelGGL(k_n_relax_s<8>, gr, block2d(), 0, st, p); break;
      default: break;
    }
    return;
  }
  switch (nl) {
    case 1: hipLaunchKernelGGL(k_n_relax<1>, gr, block2d(), 0, st, p); break;
    case 2: hipLaunchKernelGGL(k_n_relax<2>, gr, block2d(), 0, st, p); break;
    case 3: hipLaunchKernelGGL(k_n_relax<3>, gr, block2d(), 0, st, p); break;
    case 4: hipLaunchKernelGGL(k_n_relax<4>, gr, block2d(), 0, st, p); break;
    case 5: hipLaunchKernelGGL(k_n_relax<5>, gr, block2d(), 0, st, p); break;
    case 6: hipLaunchKernelGGL(k_n_relax<6>, gr, block2d(), 0, st, p); break;
    case 7: hipLaunchKernelGGL(k_n_relax<7>, gr, block2d(), 0, st, p); break;
    case 8: hipLaunchKernelGGL(k_n_relax<8>, gr, block2d(), 0, st, p); break;
    default: break;
  }
}
// residual_baroclinic qg_baroclinic_ms.h:295-341 / residual_barotropic qg_barotropic.h:78-97; max -> *maxres
struct NResArgs {
  const double *a, *b, *mk, *S2;
  double *res, *maxres;
  NatGeom g, gr;  // gr, sp: geometry and layout of res
  const double *S2row;  // S2 independent of x: [l * g.ny + j], else null
  int nl, sp;
  int zb;   // 1: the boundary vertices of res get 0 after the maximum took their value (boundary_level; formerly a launch of k_n_bnd_const)
  double sqD, iRd2;
  LayerCoef lc;
};
// at most ~2048 workgroups (rows strided over gridDim.y): one atomicMax per workgroup on a single word
// saturates at ~88 per microsecond, which would dominate a pass launched with one workgroup per 64 x 4 cells
__global__ void k_n_residual(NResArgs p) {
  const int i = blockIdx.x * BX + threadIdx.x;
  double mx = 0.;
  for (int j = blockIdx.y * BY + threadIdx.y; i < p.g.nx && j < p.g.ny; j += gridDim.y * BY) {
    const int nl = p.nl, pitch = p.g.pitch;
    const size_t ls = p.g.ls, c0 = nat_idx(p.g, 0, j, i);
    const double m = p.mk[c0], sq = p.sqD, rsq = 1. / sq;
    const bool zb = p.zb && (i == 0 || j == 0 || i == p.g.nx - 1 || j == p.g.ny - 1);
    for (int l = 0; l < nl; l++) {
      const size_t c = c0 + l * ls;
      const double a1 = p.a[c];
      double r;
      const double s2m = (nl > 1 && l > 0) ? (p.S2row ? p.S2row[(l - 1) * p.g.ny + j] : p.S2[c - ls]) : 0.;
      const double s2c = (nl > 1 && l < nl - 1) ? (p.S2row ? p.S2row[l * p.g.ny + j] : p.S2[c]) : 0.;
      if (nl == 1) r = (p.b[c] - (-p.iRd2 * a1)) * m;
      else if (l == 0) r = (p.b[c] + s2c * (a1 - p.a[c + ls]) * p.lc.idh1[l]) * m;
      else if (l < nl - 1) r = (p.b[c] + s2m * (a1 - p.a[c - ls]) * p.lc.idh0[l] - s2c * (p.a[c + ls] - a1) * p.lc.idh1[l]) * m;
      else r = (p.b[c] + s2m * (a1 - p.a[c - ls]) * p.lc.idh0[l]) * m;
      r -= DIVC(p.a[c - 1] - 2. * a1 + p.a[c + 1], sq, rsq) * m;
      r -= DIVC(p.a[c - pitch] - 2. * a1 + p.a[c + pitch], sq, rsq) * m;
      p.res[p.sp ? gidx(p.gr, 1, l, j, i) : c] = zb ? 0. : r;
      mx = fmax(mx, fabs(r));
    }
  }
  __shared__ double sm[BY];
  mx = wave_max_n(mx);
  if (threadIdx.x == 0) sm[threadIdx.y] = mx;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double v = sm[0];
    for (int k = 1; k < BY; k++) v = fmax(v, sm[k]);
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(v));
  }
}
void launch_n_residual(hipStream_t st, const double *a, const double *b, const double *mk, const double *S2, double *res, double *maxres,
                       const NatGeom &g, int nl, double D, double iRd2, const LayerCoef &lc, const NatGeom *gres, const double *S2row, int zb) {
  NResArgs p;
  p.zb = zb;
  p.S2row = S2row;
  p.a = a; p.b = b; p.mk = mk; p.S2 = S2; p.res = res; p.maxres = maxres; p.g = g; p.nl = nl; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc;
  p.sp = gres != nullptr; p.gr = gres ? *gres : g;
  hipLaunchKernelGGL(k_n_residual, grid_capped(g.nx, g.ny), block2d(), 0, st, p);
}
// Correction of cycle i and residual of cycle i + 1 in one pass (round 3): a_new = a + da (boundary vertices: the boundary value,
// k_n_correct) is formed where the residual needs it -- at the vertex, its four neighbours and the layers above / below -- by the
// same addition, the residual is k_n_residual's expression on those values, a_new goes to a second psi buffer (the neighbours of
// other threads still read the old one).  Saves the read-modify-write pass of k_n_correct: 107 + 47 -> ~115 us per cycle at 2049^2 x 3.
__global__ void k_n_correct_residual(NResArgs p, const double *__restrict__ da, NatGeom gd, int dsp, double *__restrict__ a_out, double bcv) {
  const int i = blockIdx.x * BX + threadIdx.x;
  const int n = p.g.nx - 1;
  double mx = 0.;
  for (int j = blockIdx.y * BY + threadIdx.y; i < p.g.nx && j < p.g.ny; j += gridDim.y * BY) {
    const int nl = p.nl;
    const size_t ls = p.g.ls, c0 = nat_idx(p.g, 0, j, i);
    // vertices beyond the grid (pad cells next to a boundary vertex) are not corrected: they keep what the field holds there
    auto anew = [&](int l, int jj, int ii) -> double {
      const double av = p.a[nat_idx(p.g, l, jj, ii)];
      if (ii < 0 || jj < 0 || ii > n || jj > n) return av;
      if (ii == 0 || jj == 0 || ii == n || jj == n) return bcv;
      return av + da[gidx(gd, dsp, l, jj, ii)];
    };
    const double m = p.mk[c0], sq = p.sqD, rsq = 1. / sq;
    const bool zb = p.zb && (i == 0 || j == 0 || i == p.g.nx - 1 || j == p.g.ny - 1);
    for (int l = 0; l < nl; l++) {
      const size_t c = c0 + l * ls;
      const double a1 = anew(l, j, i);
      a_out[c] = a1;
      double r;
      const double s2m = (nl > 1 && l > 0) ? (p.S2row ? p.S2row[(l - 1) * p.g.ny + j] : p.S2[c - ls]) : 0.;
      const double s2c = (nl > 1 && l < nl - 1) ? (p.S2row ? p.S2row[l * p.g.ny + j] : p.S2[c]) : 0.;
      if (nl == 1) r = (p.b[c] - (-p.iRd2 * a1)) * m;
      else if (l == 0) r = (p.b[c] + s2c * (a1 - anew(l + 1, j, i)) * p.lc.idh1[l]) * m;
      else if (l < nl - 1) r = (p.b[c] + s2m * (a1 - anew(l - 1, j, i)) * p.lc.idh0[l] - s2c * (anew(l + 1, j, i) - a1) * p.lc.idh1[l]) * m;
      else r = (p.b[c] + s2m * (a1 - anew(l - 1, j, i)) * p.lc.idh0[l]) * m;
      r -= DIVC(anew(l, j, i - 1) - 2. * a1 + anew(l, j, i + 1), sq, rsq) * m;
      r -= DIVC(anew(l, j - 1, i) - 2. * a1 + anew(l, j + 1, i), sq, rsq) * m;
      p.res[p.sp ? gidx(p.gr, 1, l, j, i) : c] = zb ? 0. : r;
      mx = fmax(mx, fabs(r));
    }
  }
  __shared__ double sm[BY];
  mx = wave_max_n(mx);
  if (threadIdx.x == 0) sm[threadIdx.y] = mx;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double v = sm[0];
    for (int k = 1; k < BY; k++) v = fmax(v, sm[k]);
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(v));
  }
}
// The same pass marching down chunks of NCR_H rows (round 3): lane = one vertex column (lanes 1..62 own a vertex, 0 and 63 carry the
// x neighbours of the strip: 62 vertices per wavefront), the corrected psi of rows j - 1, j, j + 1 in registers, x neighbours by
// whole-wave DPP shifts: psi and the correction are read once (+ 2 rows per chunk), not five times through the L1.
#define NCR_WPB 4
#define NCR_H 16
template <int NL>
__global__ void __launch_bounds__(64 * NCR_WPB) k_n_correct_residual_m(NResArgs p, const double *__restrict__ da, NatGeom gd, int dsp, double *__restrict__ a_out,
                                                                       double bcv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = p.g.nx - 1;
  const int strip = blockIdx.x * NCR_WPB + wave;
  const int i = strip * 62 - 1 + lane;
  double mx = 0.;
  if (strip * 62 <= n) {   // wave-uniform: the DPP shifts below run with every lane of a live wavefront
    const int il = i < n + 1 ? i : n + 1;       // column loaded (the pad columns -1 and n + 1 are memory of the field)
    const bool own = lane >= 1 && lane <= 62 && i <= n;
    const bool xin = il >= 1 && il <= n - 1, xout = il < 0 || il > n;
    const size_t ls = p.g.ls;
    const int j0 = blockIdx.y * NCR_H, j1 = j0 + NCR_H < n + 1 ? j0 + NCR_H : n + 1;
    // corrected psi of row j at column il: psi itself beyond the grid, the boundary value on the sides, psi + da inside (k_n_correct)
    auto row = [&](int j, double (&v)[NL]) {
      const bool yout = j < 0 || j > n, yin = j >= 1 && j <= n - 1;
      const size_t k = nat_idx(p.g, 0, j, il), kd = gidx(gd, dsp, 0, yin ? j : 1, xin ? il : 1);
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double av = p.a[k + l * ls], d = da[kd + l * gd.ls];
        v[l] = (xout || yout) ? av : ((xin && yin) ? av + d : bcv);
      }
    };
    double prev[NL], cur[NL], next[NL];
    row(j0 - 1, prev);
    row(j0, cur);
    const double sq = p.sqD, rsq = 1. / sq;
    for (int j = j0; j < j1; j++) {
      row(j + 1, next);
      const size_t c0 = nat_idx(p.g, 0, j, il);
      const double m = p.mk[c0];
      const bool zb = i == 0 || j == 0 || i == n || j == n;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double a1 = cur[l];
        const double aw = lane_below(a1), ae = lane_above(a1);
        if (own) {
          const size_t c = c0 + l * ls;
          a_out[c] = a1;
          double r;
          const double s2m = (NL > 1 && l > 0) ? (p.S2row ? p.S2row[(l - 1) * p.g.ny + j] : p.S2[c - ls]) : 0.;
          const double s2c = (NL > 1 && l < NL - 1) ? (p.S2row ? p.S2row[l * p.g.ny + j] : p.S2[c]) : 0.;
          if (NL == 1) r = (p.b[c] - (-p.iRd2 * a1)) * m;
          else if (l == 0) r = (p.b[c] + s2c * (a1 - cur[l + 1 < NL ? l + 1 : l]) * p.lc.idh1[l]) * m;
          else if (l < NL - 1) r = (p.b[c] + s2m * (a1 - cur[l - 1]) * p.lc.idh0[l] - s2c * (cur[l + 1 < NL ? l + 1 : l] - a1) * p.lc.idh1[l]) * m;
          else r = (p.b[c] + s2m * (a1 - cur[l > 0 ? l - 1 : 0]) * p.lc.idh0[l]) * m;
          r -= DIVC(aw - 2. * a1 + ae, sq, rsq) * m;
          r -= DIVC(prev[l] - 2. * a1 + next[l], sq, rsq) * m;
          p.res[p.sp ? gidx(p.gr, 1, l, j, i) : c] = zb ? 0. : r;
          mx = fmax(mx, fabs(r));
        }
      }
#pragma unroll
      for (int l = 0; l < NL; l++) { prev[l] = cur[l]; cur[l] = next[l]; }
    }
  }
  __shared__ double sm[NCR_WPB];
  mx = wave_max_n(mx);
  if (lane == 0) sm[wave] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = sm[0];
    for (int k = 1; k < NCR_WPB; k++) v = fmax(v, sm[k]);
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(v));
  }
}
template <int NL>
static void ncr_launch(hipStream_t st, const NResArgs &p, const double *da, const NatGeom &gd, int dsp, double *a_out, double bcv) {
  const int strips = (p.g.nx + 61) / 62;
  hipLaunchKernelGGL(k_n_correct_residual_m<NL>, dim3((strips + NCR_WPB - 1) / NCR_WPB, (p.g.ny + NCR_H - 1) / NCR_H), dim3(64 * NCR_WPB), 0, st, p, da, gd, dsp,
                     a_out, bcv);
}
void launch_n_correct_residual(hipStream_t st, const double *a, double *a_out, const double *da, const NatGeom *gda, double bcv, const double *b, const double *mk,
                               const double *S2, double *res, double *maxres, const NatGeom &g, int nl, double D, double iRd2, const LayerCoef &lc,
                               const NatGeom *gres, const double *S2row, int march) {
  NResArgs p;
  p.zb = 1;
  p.S2row = S2row;
  p.a = a; p.b = b; p.mk = mk; p.S2 = S2; p.res = res; p.maxres = maxres; p.g = g; p.nl = nl; p.sqD = D * D; p.iRd2 = iRd2; p.lc = lc;
  p.sp = gres != nullptr; p.gr = gres ? *gres : g;
  const NatGeom gd = gda ? *gda : g;
  const int dsp = gda != nullptr;
  switch (march ? nl : 0) {
    case 1: ncr_launch<1>(st, p, da, gd, dsp, a_out, bcv); break;
    case 2: ncr_launch<2>(st, p, da, gd, dsp, a_out, bcv); break;
    case 3: ncr_launch<3>(st, p, da, gd, dsp, a_out, bcv); break;
    case 4: ncr_launch<4>(st, p, da, gd, dsp, a_out, bcv); break;
    case 5: ncr_launch<5>(st, p, da, gd, dsp, a_out, bcv); break;
    case 6: ncr_launch<6>(st, p, da, gd, dsp, a_out, bcv); break;
    case 7: ncr_launch<7>(st, p, da, gd, dsp, a_out, bcv); break;
    case 8: ncr_launch<8>(st, p, da, gd, dsp, a_out, bcv); break;
    default: hipLaunchKernelGGL(k_n_correct_residual, grid_capped(g.nx, g.ny), block2d(), 0, st, p, da, gd, dsp, a_out, bcv);
  }
}
// restriction_coarsen_vert (residual), restriction_coarsen_vert2 (mask), restriction_vert (injection), my_vertex.h:49-75
__device__ __forceinline__ void n_restrict_pt(const double *__restrict__ f, const NatGeom &fg, double *c, const NatGeom &cg, int nl, int kind, int I, int J) {
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(fg, l, 2 * J, 2 * I);
    const int p = fg.pitch;
    double v;
    if (kind == 0) v = (f[k + 1] + 2 * f[k] + f[k - 1] + f[k + p] + f[k - p]) / 6.;
    else if (kind == 1)
      v = (4 * f[k] + 2 * f[k + 1] + 2 * f[k - 1] + 2 * f[k + p] + 2 * f[k - p] + f[k + 1 + p] + f[k - 1 + p] + f[k + 1 - p] + f[k - 1 - p]) / 16.;
    else v = f[k];
    c[nat_idx(cg, l, J, I)] = v;
  }
}
// zb: the boundary vertices of the coarse level get 0 (boundary_level of the residual, formerly a launch of k_n_bnd_const)
__global__ void k_n_restrict(const double *__restrict__ f, NatGeom fg, double *c, NatGeom cg, int nl, int kind, int zb) {
  VTX(cg, I, J);
  if (zb && (I == 0 || J == 0 || I == cg.nx - 1 || J == cg.ny - 1)) {
    for (int l = 0; l < nl; l++) c[nat_idx(cg, l, J, I)] = 0.;
    return;
  }
  n_restrict_pt(f, fg, c, cg, nl, kind, I, J);
}
// residual restriction (kind 0) with either side in the split layout: the five fine vertices of a coarse one are unit-stride
// runs of the even half (2I) and of the odd half (2I +- 1)
__global__ void k_n_restrict_s(const double *__restrict__ f, NatGeom fg, int fsp, double *c, NatGeom cg, int csp, int nl, int zb) {
  VTX(cg, I, J);
  if (zb && (I == 0 || J == 0 || I == cg.nx - 1 || J == cg.ny - 1)) {
    for (int l = 0; l < nl; l++) c[gidx(cg, csp, l, J, I)] = 0.;
    return;
  }
  for (int l = 0; l < nl; l++) {
    const double v = (f[gidx(fg, fsp, l, 2 * J, 2 * I + 1)] + 2 * f[gidx(fg, fsp, l, 2 * J, 2 * I)] + f[gidx(fg, fsp, l, 2 * J, 2 * I - 1)] +
                      f[gidx(fg, fsp, l, 2 * J + 1, 2 * I)] + f[gidx(fg, fsp, l, 2 * J - 1, 2 * I)]) / 6.;
    c[gidx(cg, csp, l, J, I)] = v;
  }
}
void launch_n_restrict(hipStream_t st, const double *f, const NatGeom &fg, double *c, const NatGeom &cg, int nl, int kind, int fsp, int csp, int zb) {
  if (fsp || csp) {
    if (kind != 0) { fprintf(stderr, "msom: launch_n_restrict: split layout only for the residual\n"); abort(); }
    hipLaunchKernelGGL(k_n_restrict_s, grid2d(cg.nx, cg.ny), block2d(), 0, st, f, fg, fsp, c, cg, csp, nl, zb);
    return;
  }
  hipLaunchKernelGGL(k_n_restrict, grid2d(cg.nx, cg.ny), block2d(), 0, st, f, fg, c, cg, nl, kind, zb);
}
// refine_vert my_vertex.h:82-105 followed by boundary_level(da) = 0 on the boundary vertices; one thread per FINE vertex
__device__ __forceinline__ void n_prolong_pt(const double *__restrict__ c, const NatGeom &cg, double *f, const NatGeom &fg, int nl, int i, int j) {
  const int n = fg.nx - 1, I = i >> 1, J = j >> 1;
  const bool bnd = i == 0 || j == 0 || i == n || j == n;
  for (int l = 0; l < nl; l++) {
    const size_t k = nat_idx(cg, l, J, I);
    double v;
    if (bnd) v = 0.;
    else if (!(i & 1) && !(j & 1)) v = c[k];
    else if ((i & 1) && !(j & 1)) v = (c[k] + c[k + 1]) / 2.;
    else if (!(i & 1)) v = (c[k] + c[k + cg.pitch]) / 2.;
    else v = (c[k] + c[k + 1] + c[k + cg.pitch] + c[k + 1 + cg.pitch]) / 4.;
    f[nat_idx(fg, l, j, i)] = v;
  }
}
__global__ void k_n_prolong(const double *__restrict__ c, NatGeom cg, double *f, NatGeom fg, int nl) {
  VTX(fg, i, j);
  n_prolong_pt(c, cg, f, fg, nl, i, j);
}
__global__ void k_n_prolong_s(const double *__restrict__ c, NatGeom cg, int csp, double *f, NatGeom fg, int fsp, int nl) {
  VTX(fg, i, j);
  const int n = fg.nx - 1, I = i >> 1, J = j >> 1;
  const bool bnd = i == 0 || j == 0 || i == n || j == n;
  for (int l = 0; l < nl; l++) {
    double v;
    if (bnd) v = 0.;
    else if (!(i & 1) && !(j & 1)) v = c[gidx(cg, csp, l, J, I)];
    else if ((i & 1) && !(j & 1)) v = (c[gidx(cg, csp, l, J, I)] + c[gidx(cg, csp, l, J, I + 1)]) / 2.;
    else if (!(i & 1)) v = (c[gidx(cg, csp, l, J, I)] + c[gidx(cg, csp, l, J + 1, I)]) / 2.;
    else v = (c[gidx(cg, csp, l, J, I)] + c[gidx(cg, csp, l, J, I + 1)] + c[gidx(cg, csp, l, J + 1, I)] + c[gidx(cg, csp, l, J + 1, I + 1)]) / 4.;
    f[gidx(fg, fsp, l, j, i)] = v;
  }
}
// natural <-> split copy of a level array (mask / S2 copies of the wide levels, the parity tests' upload / download)
__global__ void k_n_relayout(const double *__restrict__ src, NatGeom sg, int ssp, double *dst, NatGeom dg, int dsp, int nl) {
  VTX(dg, i, j);
  for (int l = 0; l < nl; l++) dst[gidx(dg, dsp, l, j, i)] = src[gidx(sg, ssp, l, j, i)];
}
// out[l * ny + j] = f(l, j, i = 1): the row table of a field that does not depend on x
__global__ void k_n_row_table(const double *__restrict__ f, NatGeom g, int nl, double *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nl * g.ny) return;
  out[t] = f[nat_idx(g, t / g.ny, t % g.ny, 1)];
}
void launch_n_row_table(hipStream_t st, const double *f, const NatGeom &g, int nl, double *out) {
  hipLaunchKernelGGL(k_n_row_table, dim3((nl * g.ny + 255) / 256), dim3(256), 0, st, f, g, nl, out);
}
void launch_n_relayout(hipStream_t st, const double *src, const NatGeom &sg, int ssp, double *dst, const NatGeom &dg, int dsp, int nl) {
  hipLaunchKernelGGL(k_n_relayout, grid2d(dg.nx, dg.ny), block2d(), 0, st, src, sg, ssp, dst, dg, dsp, nl);
}

// ---- the coarse levels of one vpoisson cycle in ONE launch (one workgroup, __syncthreads() where the separate launches had
// kernel boundaries; the per-vertex device functions are the ones of the stand-alone kernels => identical arithmetic):
// restriction of the residual from lev[0] down (boundary vertices 0), zero first guess on the coarsest level, nrelax
// red-black sweeps per level, prolongation to the next finer level -- up to and including the sweeps of lev[0].
// At 2049^2 x 3 the levels <= 33^2 vertices are 6 of 11 and ~80 of the ~140 launches of a cycle, each ~5 us long.
template <int NL>
__global__ void __launch_bounds__(NMGC_NT) k_n_mg_coarse(NCoarseArgs a, int nrelax) {
  const int tid = threadIdx.x;
  for (int k = 1; k < a.n; k++) {
    const NCoarseLev &F = a.lev[k - 1], &C = a.lev[k];
    const int n1 = C.g.nx;
    for (int t = tid; t < n1 * n1; t += NMGC_NT) n_restrict_pt(F.res, F.g, C.res, C.g, NL, 0, t % n1, t / n1);
    __syncthreads();
    for (int t = tid; t < n1 * n1; t += NMGC_NT) {   // boundary_level(res) = 0 on the boundary vertices
      const int i = t % n1, j = t / n1;
      if (i == 0 || j == 0 || i == n1 - 1 || j == n1 - 1)
        for (int l = 0; l < NL; l++) C.res[nat_idx(C.g, l, j, i)] = 0.;
    }
    __syncthreads();
  }
  {
    const NCoarseLev &L = a.lev[a.n - 1];
    for (size_t t = tid; t < L.g.ls * NL; t += NMGC_NT) L.da[t] = 0.;
    __syncthreads();
  }
  for (int k = a.n - 1; k >= 0; k--) {
    const NCoarseLev &L = a.lev[k];
    NRelaxArgs p;
    p.a = L.da; p.b = L.res; p.mk = L.mask; p.S2 = L.S2; p.g = L.g; p.sqD = L.sqD; p.iRd2 = a.iRd2; p.lc = a.lc;
    const int n = L.g.nx - 1, half = (n + 1) / 2;
    for (int s = 0; s < nrelax; s++)
      for (int c = 0; c < 2; c++) {
        p.color = c;
        for (int t = tid; t < half * (n - 1); t += NMGC_NT) {
          const int j = 1 + t / half;
          n_relax_pt<NL>(p, 1 + 2 * (t % half) + ((j + c + 1) & 1), j);
        }
        __syncthreads();
      }
    if (k > 0) {
      const NCoarseLev &Fn = a.lev[k - 1];
      const int n1 = Fn.g.nx;
      for (int t = tid; t < n1 * n1; t += NMGC_NT) n_prolong_pt(L.da, L.g, Fn.da, Fn.g, NL, t % n1, t / n1);
      __syncthreads();
    }
  }
}
void launch_n_mg_coarse(hipStream_t st, const NCoarseArgs &a, int nrelax, int nl) {
  switch (nl) {
    case 1: hipLaunchKernelGGL(k_n_mg_coarse<1>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 2: hipLaunchKernelGGL(k_n_mg_coarse<2>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 3: hipLaunchKernelGGL(k_n_mg_coarse<3>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 4: hipLaunchKernelGGL(k_n_mg_coarse<4>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 5: hipLaunchKernelGGL(k_n_mg_coarse<5>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 6: hipLaunchKernelGGL(k_n_mg_coarse<6>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 7: hipLaunchKernelGGL(k_n_mg_coarse<7>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    case 8: hipLaunchKernelGGL(k_n_mg_coarse<8>, dim3(1), dim3(NMGC_NT), 0, st, a, nrelax); break;
    default: break;
  }
}
void launch_n_prolong(hipStream_t st, const double *c, const NatGeom &cg, double *f, const NatGeom &fg, int nl, int csp, int fsp) {
  if (csp || fsp) hipLaunchKernelGGL(k_n_prolong_s, grid2d(fg.nx, fg.ny), block2d(), 0, st, c, cg, csp, f, fg, fsp, nl);
  else hipLaunchKernelGGL(k_n_prolong, grid2d(fg.nx, fg.ny), block2d(), 0, st, c, cg, f, fg, nl);
}
// a += da, then boundary(a): psi_bc on the boundary vertices (nodal-poisson.h:119-128)
__global__ void k_n_correct(double *a, const double *__restrict__ da, NatGeom g, NatGeom gd, int sp, int nl, double bcv) {
  VTX(g, i, j);
  const int n = g.nx - 1;
  const bool bnd = i == 0 || j == 0 || i == n || j == n;
  size_t k = nat_idx(g, 0, j, i), kd = gidx(gd, sp, 0, j, i);
  for (int l = 0; l < nl; l++, k += g.ls, kd += gd.ls) a[k] = bnd ? bcv : a[k] + da[kd];
}
void launch_n_correct(hipStream_t st, double *a, const double *da, const NatGeom &g, int nl, double bcv, const NatGeom *gda) {
  hipLaunchKernelGGL(k_n_correct, grid2d(g.nx, g.ny), block2d(), 0, st, a, da, g, gda ? *gda : g, gda != nullptr, nl, bcv);
}
// adjust_dt qg-node/qg.h:258-284: max |psi[0,1] - psi[]| / D and |psi[1,0] - psi[]| / D over faces and layers
__global__ void k_n_umax(const double *__restrict__ psi, double *out, NatGeom g, int nl, double D) {
  const int i = blockIdx.x * BX + threadIdx.x;
  double m = 0.;
  for (int j = blockIdx.y * BY + threadIdx.y; i < g.nx && j < g.ny; j += gridDim.y * BY) {
    const double rD = 1. / D;
    for (int l = 0; l < nl; l++) {
      const size_t c = nat_idx(g, l, j, i);
      if (j < g.ny - 1) m = fmax(m, fabs(DIVC(psi[c + g.pitch] - psi[c], D, rD)));
      if (i < g.nx - 1) m = fmax(m, fabs(DIVC(psi[c + 1] - psi[c], D, rD)));
    }
  }
  __shared__ double sm[BY];
  m = wave_max_n(m);
  if (threadIdx.x == 0) sm[threadIdx.y] = m;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double v = sm[0];
    for (int k = 1; k < BY; k++) v = fmax(v, sm[k]);
    atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));
  }
}
void launch_n_umax(hipStream_t st, const double *psi, double *out, const NatGeom &g, int nl, double D) {
  hipLaunchKernelGGL(k_n_umax, grid_capped(g.nx, g.ny), block2d(), 0, st, psi, out, g, nl, D);
}
// KE diagnostic qg-node/qg.c:172-178 (per-block partials, summed by launch_sum_final)
__global__ void k_n_ke(const double *__restrict__ psi, double *partial, NatGeom g, double D2, double rD2) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny) {
    const size_t c = nat_idx(g, 0, j, i);
    v = 0.5 * psi[c] * DIVC(LAPN(psi, c, g.pitch), D2, rD2) * D2;
  }
  __shared__ double sm[BY];
  v = wave_sum_n(v);
  if (threadIdx.x == 0) sm[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double s = 0.;
    for (int k = 0; k < BY; k++) s += sm[k];
    partial[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}
void launch_n_ke(hipStream_t st, const double *psi, double *partial, double *out, const NatGeom &g, double D) {
  dim3 gr = grid2d(g.nx, g.ny);
  hipLaunchKernelGGL(k_n_ke, gr, block2d(), 0, st, psi, partial, g, D * D, 1. / (D * D));
  launch_sum_final(st, partial, out, (int)(gr.x * gr.y));
}

// stochastic forcing, qg-node/qg.h:316-317: q_0[vertex (i, j)] += n_stoch[cell (i, j)] * dts; the cell field has N x N
// cells, vertices i, j = N read its ghost cells
__global__ void k_n_add_noise(double *q, const double *__restrict__ n, NatGeom g, NatGeom cg, double dts) {
  VTX(g, i, j);
  q[nat_idx(g, 0, j, i)] += n[nat_idx(cg, 0, j, i)] * dts;
}
void launch_n_add_noise(hipStream_t st, double *q, const double *n, const NatGeom &g, const NatGeom &cg, double dts) {
  hipLaunchKernelGGL(k_n_add_noise, grid2d(g.nx, g.ny), block2d(), 0, st, q, n, g, cg, dts);
}

// event write_1d_diag qg-node/qg.h:361-399: ke, dissipation and forcing sums over the cell loop (vertices i, j < N),
// top layer; per-block partials [3][nblk], summed by launch_sum_final
__global__ void k_n_diag1d(const double *__restrict__ psi, const double *__restrict__ q, const double *__restrict__ qf, double *partial, NatGeom g, double nu,
                           double D2, double rD2) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  double v[3] = {0., 0., 0.};
  if (i < g.nx - 1 && j < g.ny - 1) {
    const size_t c = nat_idx(g, 0, j, i);
    v[0] = 0.5 * psi[c] * DIVC(LAPN(psi, c, g.pitch), D2, rD2) * D2;
    v[1] = nu * psi[c] * DIVC(LAPN(q, c, g.pitch), D2, rD2) * D2;
    v[2] = psi[c] * qf[c] * D2;
  }
  __shared__ double sm[3][BY];
  for (int k = 0; k < 3; k++) {
    const double w = wave_sum_n(v[k]);
    if (threadIdx.x == 0) sm[k][threadIdx.y] = w;
  }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < 3) {
    double s = 0.;
    for (int k = 0; k < BY; k++) s += sm[threadIdx.x][k];
    partial[(size_t)threadIdx.x * (gridDim.x * gridDim.y + 64) + blockIdx.y * gridDim.x + blockIdx.x] = s;  // + 64: chunk sums of launch_sum_final
  }
}
void launch_n_diag1d(hipStream_t st, const double *psi, const double *q, const double *qf, double *partial, double *out3, const NatGeom &g, double nu, double D) {
  dim3 gr = grid2d(g.nx, g.ny);
  hipLaunchKernelGGL(k_n_diag1d, gr, block2d(), 0, st, psi, q, qf, partial, g, nu, D * D, 1. / (D * D));
  const int nb = (int)(gr.x * gr.y);
  for (int k = 0; k < 3; k++) launch_sum_final(st, partial + (size_t)k * (nb + 64), out3 + k, nb);
}
