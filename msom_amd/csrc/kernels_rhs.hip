// kernels_rhs.hip -- PV-tendency stencil kernels of libmsomhip (gfx950 / CDNA4, fp64).
//
// One thread owns one (x, y) column and walks the nl layers, so the vertical coupling
// (stretching, ju = -jd carry of the Arakawa cross-layer Jacobians) stays in registers and
// every global access is coalesced along x (64-wide wavefronts, rows 128-B aligned).
// Reference loops restated here (file:line in the reference tree):
//   comp_del2 msqg/qg.h:172-200, comp_stretch :203-246, jacobian :252-262, beta_effect :269,
//   comp_vel :276-283, advection_pv :288-393, dissip :407-422, ekman_friction :429-440,
//   surface_forcing :447-459, qforcing :466-474, bottom_topography :481-488,
//   advance_qg :594-606, KE diagnostic msqg/qg.c:101-109.
//
// Arithmetic: with -DMSOM_STRICT (validation build, -ffp-contract=off) every expression is
// evaluated in the reference's order with true divisions, which makes the kernels bit-exact
// against the CPU oracle; the product build multiplies by precomputed reciprocals and lets
// the compiler contract to FMA.
#include "kernels.h"

#ifdef MSOM_STRICT
#define DIVC(x, c, rc) ((x) / (c))
#else
#define DIVC(x, c, rc) ((x) * (rc))
#endif

#define BX 64
#define BY 4

static inline dim3 grid2d(int nx, int ny) { return dim3((nx + BX - 1) / BX, (ny + BY - 1) / BY); }
static inline dim3 block2d() { return dim3(BX, BY); }

// ------------------------------------------------------------------ block reductions

__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// max of non-negative doubles: IEEE order == unsigned integer order
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
  atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void block_max_to(double *addr, double v) {
  __shared__ double sm[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, w = tid >> 6, nw = (blockDim.x * blockDim.y + 63) >> 6;
  v = wave_max(v);
  if ((tid & 63) == 0) sm[w] = v;
  __syncthreads();
  if (tid == 0) {
    double m = sm[0];
    for (int k = 1; k < nw; k++) m = fmax(m, sm[k]);
    atomic_max_nonneg(addr, m);
  }
}

// ------------------------------------------------------------------ ghost fill

// boundary(): box BCs direction by direction (x walls first, then y walls over the x-ghosts,
// so corner ghosts are the y-BC applied to the x-ghost column).  Only sides flagged in
// `walls` are physical walls; the others are tile edges filled by the halo exchange.
__global__ void k_fill_ghost(double *f, NatGeom g, int nl, int bc, int walls, int d) {
  const int per = 2 * g.ny + 2 * (g.nx + 2 * d);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  const double s = bc == BC_DIRICHLET0 ? -1. : 1.;
  if (r < 2 * g.ny) {
    const int j = r >> 1, east = r & 1;
    if (east) {
      if (walls & WALL_E) f[nat_idx(g, l, j, g.nx)] = bc == BC_PERIODIC ? f[nat_idx(g, l, j, 0)] : s * f[nat_idx(g, l, j, g.nx - 1)];
    } else {
      if (walls & WALL_W) f[nat_idx(g, l, j, -1)] = bc == BC_PERIODIC ? f[nat_idx(g, l, j, g.nx - 1)] : s * f[nat_idx(g, l, j, 0)];
    }
  } else {
    // y walls over i in [-d, nx + d): the x-ghost columns take part (corner ghosts = y-BC of the
    // x-ghost).  d > 1 only matters on tile edges, where those columns hold exchanged values.
    const int q = r - 2 * g.ny, i = (q >> 1) - d, north = q & 1;
    if (!(walls & (north ? WALL_N : WALL_S))) return;
    const int jsrc = bc == BC_PERIODIC ? (north ? 0 : g.ny - 1) : (north ? g.ny - 1 : 0);
    double v;
    if (i < 0 && (walls & WALL_W)) {
      if (i != -1) return;
      v = bc == BC_PERIODIC ? f[nat_idx(g, l, jsrc, g.nx - 1)] : s * f[nat_idx(g, l, jsrc, 0)];
    } else if (i >= g.nx && (walls & WALL_E)) {
      if (i != g.nx) return;
      v = bc == BC_PERIODIC ? f[nat_idx(g, l, jsrc, 0)] : s * f[nat_idx(g, l, jsrc, g.nx - 1)];
    } else
      v = f[nat_idx(g, l, jsrc, i)];
    f[nat_idx(g, l, north ? g.ny : -1, i)] = bc == BC_PERIODIC ? v : s * v;
  }
}

void launch_fill_ghost(hipStream_t st, double *f, const NatGeom &g, int nl, int bc, int walls, int depth) {
  const int n = (2 * g.ny + 2 * (g.nx + 2 * depth)) * nl;
  hipLaunchKernelGGL(k_fill_ghost, dim3((n + 255) / 256), dim3(256), 0, st, f, g, nl, bc, walls, depth);
}

// doubly periodic ghost cells up to depth d (sbc = -1: periodic(right), periodic(top),
// msqg/qg.h:842-846): every cell of the d-wide frame is the wrapped copy of an interior cell
__global__ void k_fill_periodic(double *f, NatGeom g, int nl, int d) {
  const int wide = g.nx + 2 * d, per = 2 * d * wide + 2 * d * g.ny;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  int i, j;
  if (r < 2 * d * wide) {  // rows below and above
    const int row = r / wide;
    i = r % wide - d;
    j = row < d ? row - d : g.ny + (row - d);
  } else {
    const int q = r - 2 * d * wide;
    j = q / (2 * d);
    const int c = q % (2 * d);
    i = c < d ? c - d : g.nx + (c - d);
  }
  const int si = ((i % g.nx) + g.nx) % g.nx, sj = ((j % g.ny) + g.ny) % g.ny;
  f[nat_idx(g, l, j, i)] = f[nat_idx(g, l, sj, si)];
}
void launch_fill_periodic(hipStream_t st, double *f, const NatGeom &g, int nl, int depth) {
  const int n = (2 * depth * (g.nx + 2 * depth) + 2 * depth * g.ny) * nl;
  hipLaunchKernelGGL(k_fill_periodic, dim3((n + 255) / 256), dim3(256), 0, st, f, g, nl, depth);
}

// large-scale stream function in a periodic domain: dirichlet(vpg*x - upg*y) on the four wall
// faces (msqg/qg.h:1105-1114): ghost = 2 * value(face centre) - interior; corners = y-BC of
// the x-ghost column
struct LinBC { double u[MSOM_MAXNL], v[MSOM_MAXNL]; };
// Tiles (ox, oy = global index of the tile's first cell, sides = which tile edges lie on the domain boundary): the other ghosts
// were filled by the halo exchange and act as interior values here, so that ghost cell (i, j) gets the same number on every tile
// that holds it: an x-side ghost of the rows j = -1, ny (ghost rows of a neighbour tile, not of the domain) from the exchanged
// value next to it, a y-side ghost of the columns i = -1, nx from the exchanged value when that x side is not a domain edge.
__global__ void k_fill_lin_dirichlet(double *f, NatGeom g, int nl, LinBC b, double D, int ox, int oy, double Lx, double Ly, int sides) {
  const int per = 2 * (g.ny + 2) + 2 * (g.nx + 2);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  const double u = b.u[l], v = b.v[l];
  if (r < 2 * (g.ny + 2)) {
    const int j = (r >> 1) - 1, east = r & 1;
    if (!(sides & (east ? WALL_E : WALL_W))) return;
    if ((j == -1 && (sides & WALL_S)) || (j == g.ny && (sides & WALL_N))) return;  // corner of the domain: the y rule below
    const double y = (oy + j + 0.5) * D;
    if (east) f[nat_idx(g, l, j, g.nx)] = 2. * (v * Lx - u * y) - f[nat_idx(g, l, j, g.nx - 1)];
    else f[nat_idx(g, l, j, -1)] = 2. * (v * 0. - u * y) - f[nat_idx(g, l, j, 0)];
  } else {
    const int q = r - 2 * (g.ny + 2), i = (q >> 1) - 1, north = q & 1;
    if (!(sides & (north ? WALL_N : WALL_S))) return;
    const int jsrc = north ? g.ny - 1 : 0;
    const double x = (ox + i + 0.5) * D;
    double inner;  // value of the x-extended row next to the wall
    if (i == -1 && (sides & WALL_W)) inner = 2. * (v * 0. - u * ((oy + jsrc + 0.5) * D)) - f[nat_idx(g, l, jsrc, 0)];
    else if (i == g.nx && (sides & WALL_E)) inner = 2. * (v * Lx - u * ((oy + jsrc + 0.5) * D)) - f[nat_idx(g, l, jsrc, g.nx - 1)];
    else inner = f[nat_idx(g, l, jsrc, i)];
    f[nat_idx(g, l, north ? g.ny : -1, i)] = 2. * (v * x - u * (north ? Ly : 0.)) - inner;
  }
}
void launch_fill_lin_dirichlet(hipStream_t st, double *f, const NatGeom &g, int nl, const double *upg, const double *vpg, double D, double Lx,
                               double Ly, int ox, int oy, int sides) {
  LinBC b;
  for (int l = 0; l < MSOM_MAXNL; l++) { b.u[l] = l < nl ? upg[l] : 0.; b.v[l] = l < nl ? vpg[l] : 0.; }
  const int n = (2 * (g.ny + 2) + 2 * (g.nx + 2)) * nl;
  hipLaunchKernelGGL(k_fill_lin_dirichlet, dim3((n + 255) / 256), dim3(256), 0, st, f, g, nl, b, D, ox, oy, Lx, Ly, sides);
}

// partial-slip override of the zeta ghosts, msqg/qg.h:185-198
__global__ void k_slip_bc(const double *po, double *zeta, NatGeom g, int nl, double c, int walls) {
  const int per = 2 * g.ny + 2 * g.nx;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nl) return;
  const int l = t / per, r = t % per;
  int i, j, gi, gj, side;
  if (r < 2 * g.ny) { j = gj = r >> 1; if (r & 1) { i = g.nx - 1; gi = g.nx; side = WALL_E; } else { i = 0; gi = -1; side = WALL_W; } }
  else { const int q = r - 2 * g.ny; i = gi = q >> 1; if (q & 1) { j = g.ny - 1; gj = g.ny; side = WALL_N; } else { j = 0; gj = -1; side = WALL_S; } }
  if (!(walls & side)) return;
  zeta[nat_idx(g, l, gj, gi)] = c * (po[nat_idx(g, l, j, i)] - po[nat_idx(g, l, gj, gi)]);
}
void launch_slip_bc(hipStream_t st, const double *po, double *zeta, const NatGeom &g, int nl, double c, int walls) {
  const int n = (2 * g.ny + 2 * g.nx) * nl;
  hipLaunchKernelGGL(k_slip_bc, dim3((n + 255) / 256), dim3(256), 0, st, po, zeta, g, nl, c, walls);
}

// ------------------------------------------------------------------ pack / unpack

// contiguous [layer][y][x] <-> padded natural layout
__global__ void k_pack(const double *src, double *dst, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) dst[nat_idx(g, l, j, i)] = src[((size_t)l * g.ny + j) * g.nx + i];
}
__global__ void k_unpack(const double *src, double *dst, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) dst[((size_t)l * g.ny + j) * g.nx + i] = src[nat_idx(g, l, j, i)];
}
void launch_pack(hipStream_t st, const double *src, double *dst, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_pack, grid2d(g.nx, g.ny), block2d(), 0, st, src, dst, g, nl);
}
void launch_unpack(hipStream_t st, const double *src, double *dst, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_unpack, grid2d(g.nx, g.ny), block2d(), 0, st, src, dst, g, nl);
}

// ------------------------------------------------------------------ K1 comp_del2

#define LAPV(p, c, pitch) ((p)[(c) + 1] + (p)[(c)-1] + (p)[(c) + (pitch)] + (p)[(c) - (pitch)] - 4 * (p)[c])

__global__ void k_del2(const double *__restrict__ po, double *zeta, NatGeom g, int nl, double add, double fac, double D2, double rD2) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) {
    const double lap = DIVC(LAPV(po, c, g.pitch), D2, rD2);
    zeta[c] = add == 0. ? fac * lap : add * zeta[c] + fac * lap;
  }
}
void launch_del2(hipStream_t st, const double *po, double *zeta, const NatGeom &g, int nl, double add, double fac, double D) {
  const double D2 = D * D;
  hipLaunchKernelGGL(k_del2, grid2d(g.nx, g.ny), block2d(), 0, st, po, zeta, g, nl, add, fac, D2, 1. / D2);
}

// ------------------------------------------------------------------ K2 comp_stretch

__global__ void k_stretch(const double *__restrict__ po, double *st, const double *__restrict__ S, NatGeom g, int nl, double add,
                          double fac, LayerCoef lc) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const size_t c0 = nat_idx(g, 0, j, i);
  if (nl == 1) { st[c0] = add == 0. ? 0. : add * st[c0]; return; }  // Gamma = 0 (reference: degenerate, qg.h:239-242)
  double pm = 0., pc = po[c0], pp, s0 = 0., s1;
  for (int l = 0; l < nl; l++) {
    const size_t c = c0 + (size_t)l * g.ls;
    double v;
    if (l < nl - 1) { pp = po[c + g.ls]; s1 = S[c]; }
    if (l == 0) v = fac * s1 * (pp - pc) * lc.idh1[l];
    else if (l < nl - 1) v = fac * (s0 * (pm - pc) * lc.idh0[l] + s1 * (pp - pc) * lc.idh1[l]);
    else v = fac * s0 * (pm - pc) * lc.idh0[l];
    st[c] = add == 0. ? v : add * st[c] + v;
    pm = pc; pc = pp; s0 = s1;
  }
}
void launch_stretch(hipStream_t st, const double *po, double *out, const double *S, const NatGeom &g, int nl, double add, double fac,
                    const LayerCoef &lc) {
  hipLaunchKernelGGL(k_stretch, grid2d(g.nx, g.ny), block2d(), 0, st, po, out, S, g, nl, add, fac, lc);
}

// ------------------------------------------------------------------ K3 advection_pv

// -J(p,q) of msqg/qg.h:252-262, evaluated in the reference's order
__device__ __forceinline__ double mjac(const double *__restrict__ p, const double *__restrict__ q, size_t c, int pitch, double D12, double rD12) {
#define P(a, b) p[c + (a) + (ptrdiff_t)(b)*pitch]
#define Q(a, b) q[c + (a) + (ptrdiff_t)(b)*pitch]
  const double s = (Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0)) +
                   Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1)) - Q(0, 1) * (P(1, 1) - P(-1, 1)) +
                   Q(0, -1) * (P(1, -1) - P(-1, -1)) + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1)) -
                   P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1));
#undef P
#undef Q
  return DIVC(s, D12, rD12);
}

struct AdvArgs {
  const double *zeta, *psi, *psipg, *zetapg, *S, *qot;
  double *dq;
  NatGeom g;
  int nl, have_pg, have_zpg, stochastic;
  double D, beta, itr_stoch;
  LayerCoef lc;
};

__global__ void k_advection(AdvArgs a) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= a.g.nx || j >= a.g.ny) return;
  const int pitch = a.g.pitch, nl = a.nl;
  const size_t ls = a.g.ls;
  const double D12 = 12. * a.D * a.D, rD12 = 1. / D12, D2x = 2 * a.D, rD2x = 1. / D2x;
  size_t c = nat_idx(a.g, 0, j, i);
  // nl == 1: the reference zeroes dq (msqg/qg.h:376-379, degenerate); the build defines it as
  // the barotropic tendency = this loop without cross-layer terms
  double ju = 0., jd = 0.;
  for (int l = 0; l < nl; l++, c += ls) {
    const double *po = a.psi, *qo = a.zeta, *pp = a.psipg;
    ju = -jd;
    if (l < nl - 1) {
      jd = a.stochastic ? 0. : mjac(po, po + ls, c, pitch, D12, rD12);
      if (a.have_pg) {
        const double j2 = mjac(pp, po + ls, c, pitch, D12, rD12), j3 = mjac(po, pp + ls, c, pitch, D12, rD12);
        jd = a.stochastic ? j2 + j3 : jd + j2 + j3;
      }
    }
    double t = (a.stochastic && l == 0) ? 0. : mjac(po, qo, c, pitch, D12, rD12);
    if (a.have_pg) {
      const double jp = mjac(pp, qo, c, pitch, D12, rD12);
      t = (a.stochastic && l == 0) ? jp : t + jp;
    }
    const double be = DIVC(a.beta * (po[c - 1] - po[c + 1]), D2x, rD2x);
    t = (a.stochastic && l == 0 && !a.have_pg) ? be : t + be;
    if (l > 0) t = t + a.S[c - ls] * ju * a.lc.idh0[l];
    if (l < nl - 1) t = t + a.S[c] * jd * a.lc.idh1[l];
    double d = a.dq[c] + t;
    if (a.have_zpg) d += mjac(po, a.zetapg, c, pitch, D12, rD12);
    if (a.stochastic) d += -a.qot[c] * a.itr_stoch;
    a.dq[c] = d;
  }
}

// ------------------------------------------------------------------ K4 comp_vel: max |u| on faces

// max |u| over the faces of every layer of one layered field -> out[l].  Persistent-style
// grid (<= 2048 blocks, row-strided): each thread keeps its nl running maxima in registers,
// one block reduction per layer at the end, per-block partials, then a 1-block final pass.
#define UMAX_MAXBLOCKS 2048
struct UmaxArgs {
  const double *f;
  double *partial;  // [gridDim.x][nl]
  NatGeom g;
  int nl;
  double D;
};
__global__ void __launch_bounds__(256) k_umax(UmaxArgs a) {
  const double rD = 1. / a.D;
  const int pitch = a.g.pitch;
  double m[MSOM_MAXNL];
#pragma unroll
  for (int l = 0; l < MSOM_MAXNL; l++) m[l] = 0.;
  // rows j in [0, ny], columns i in [0, nx]: x-face (i,j) needs j < ny, y-face (i,j) needs i < nx
  for (int j = blockIdx.x; j <= a.g.ny; j += gridDim.x)
    for (int i = threadIdx.x; i <= a.g.nx; i += blockDim.x) {
      size_t c = nat_idx(a.g, 0, j, i);
#pragma unroll
      for (int l = 0; l < MSOM_MAXNL; l++) {
        if (l < a.nl) {
          const double *p = a.f;
          double v = 0.;
          if (j < a.g.ny)  // msqg/qg.h:280
            v = fabs(DIVC(0.25 * (p[c + pitch] - p[c - pitch] + p[c - 1 + pitch] - p[c - 1 - pitch]), a.D, rD));
          if (i < a.g.nx)
            v = fmax(v, fabs(DIVC(0.25 * (p[c + 1] - p[c - 1] + p[c + 1 - pitch] - p[c - 1 - pitch]), a.D, rD)));
          m[l] = fmax(m[l], v);
          c += a.g.ls;
        }
      }
    }
  __shared__ double sm[4][MSOM_MAXNL];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < MSOM_MAXNL; l++) {
    if (l < a.nl) {
      const double v = wave_max(m[l]);
      if ((threadIdx.x & 63) == 0) sm[w][l] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < a.nl)
    a.partial[(size_t)blockIdx.x * a.nl + threadIdx.x] =
        fmax(fmax(sm[0][threadIdx.x], sm[1][threadIdx.x]), fmax(sm[2][threadIdx.x], sm[3][threadIdx.x]));
}
// out[l] = max_b partial[b][l]
__global__ void k_max_final(const double *partial, double *out, int nb, int nl) {
  __shared__ double sm[256];
  const int l = blockIdx.x;
  double v = 0.;
  for (int b = threadIdx.x; b < nb; b += 256) v = fmax(v, partial[(size_t)b * nl + l]);
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[l] = sm[0];
}
void launch_umax(hipStream_t st, const double *f, double *partial, double *out, const NatGeom &g, int nl, double D) {
  UmaxArgs a;
  a.f = f; a.partial = partial; a.g = g; a.nl = nl; a.D = D;
  const int nb = g.ny + 1 < UMAX_MAXBLOCKS ? g.ny + 1 : UMAX_MAXBLOCKS;
  hipLaunchKernelGGL(k_umax, dim3(nb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_max_final, dim3(nl), dim3(256), 0, st, partial, out, nb, nl);
}

void launch_advection(hipStream_t st, const double *zeta, const double *psi, const double *psipg, const double *zetapg, const double *S,
                      const double *qot, double *dq, const NatGeom &g, int nl, int have_pg, int have_zpg, int stochastic, double D,
                      double beta, double itr_stoch, const LayerCoef &lc) {
  AdvArgs a;
  a.zeta = zeta; a.psi = psi; a.psipg = psipg; a.zetapg = zetapg; a.S = S; a.qot = qot; a.dq = dq;
  a.g = g; a.nl = nl; a.have_pg = have_pg; a.have_zpg = have_zpg; a.stochastic = stochastic;
  a.D = D; a.beta = beta; a.itr_stoch = itr_stoch; a.lc = lc;
  hipLaunchKernelGGL(k_advection, grid2d(g.nx, g.ny), block2d(), 0, st, a);
}

// ------------------------------------------------------------------ K5 pieces, K6, K7

// dq += c * tmp, msqg/qg.h:413-418
__global__ void k_axpy(double *dq, const double *__restrict__ x, NatGeom g, int nl, double c) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t k = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, k += g.ls) dq[k] += x[k] * c;
}
void launch_axpy(hipStream_t st, double *dq, const double *x, const NatGeom &g, int nl, double c) {
  hipLaunchKernelGGL(k_axpy, grid2d(g.nx, g.ny), block2d(), 0, st, dq, x, g, nl, c);
}

struct ForcArgs {
  const double *zeta, *psi, *qforc, *topo, *Ro, *wind;  // wind: per-row forcing profile (host libm)
  double *dq;
  NatGeom g;
  int nl, have_qforc, flag_topo;
  double cs, cb, D, dhb;  // cs = Eks/(Rom*2*dh0), cb = Ekb/(Rom*2*dh_{nl-1})
};
__global__ void k_forcing(ForcArgs a) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= a.g.nx || j >= a.g.ny) return;
  const size_t c0 = nat_idx(a.g, 0, j, i), cb = c0 + (size_t)(a.nl - 1) * a.g.ls;
  // ekman_friction :437-438 (top first, then bottom; for nl == 1 both hit the same cell)
  a.dq[c0] -= a.cs * a.zeta[c0];
  a.dq[cb] -= a.cb * a.zeta[cb];
  // surface_forcing :451
  a.dq[c0] -= a.wind[j];
  if (a.have_qforc)
    for (int l = 0; l < a.nl; l++) a.dq[c0 + (size_t)l * a.g.ls] += a.qforc[c0 + (size_t)l * a.g.ls];
  if (a.flag_topo) {  // bottom_topography :486
    const double D12 = 12. * a.D * a.D;
    a.dq[cb] += mjac(a.psi + (size_t)(a.nl - 1) * a.g.ls, a.topo, c0, a.g.pitch, D12, 1. / D12) / (a.Ro[c0] * a.dhb);
  }
}
void launch_forcing(hipStream_t st, const double *zeta, const double *psi, const double *qforc, const double *topo, const double *Ro,
                    const double *wind, double *dq, const NatGeom &g, int nl, int have_qforc, int flag_topo, double cs, double cb,
                    double D, double dhb) {
  ForcArgs a;
  a.zeta = zeta; a.psi = psi; a.qforc = qforc; a.topo = topo; a.Ro = Ro; a.wind = wind; a.dq = dq;
  a.g = g; a.nl = nl; a.have_qforc = have_qforc; a.flag_topo = flag_topo; a.cs = cs; a.cb = cb; a.D = D; a.dhb = dhb;
  hipLaunchKernelGGL(k_forcing, grid2d(g.nx, g.ny), block2d(), 0, st, a);
}

// advance_qg :597-604; stochastic variant msqg/qg_stochastic.h:139-147.  dt is read from
// device memory so the stage-1 dt never has to travel through the host.
__global__ void k_advance(double *qo, const double *qi, const double *__restrict__ dq, const double *__restrict__ noise, NatGeom g, int nl,
                          double dt, double dts) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t k = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, k += g.ls) {
    double v = qi[k] + dq[k] * dt;
    if (noise) v = v + noise[k] * dts;
    qo[k] = v;
  }
}
void launch_advance(hipStream_t st, double *qo, const double *qi, const double *dq, const double *noise, const NatGeom &g, int nl, double dt,
                    double dts) {
  hipLaunchKernelGGL(k_advance, grid2d(g.nx, g.ny), block2d(), 0, st, qo, qi, dq, noise, g, nl, dt, dts);
}

// ------------------------------------------------------------------ sums (deterministic two-stage)

// stage 1: one partial per block; stage 2: a single block adds the partials in index order
__global__ void k_ke_partial(const double *__restrict__ po, double *partial, NatGeom g, double D2, double rD2) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny) {
    const size_t c = nat_idx(g, 0, j, i);
    v = 0.5 * po[c] * DIVC(LAPV(po, c, g.pitch), D2, rD2) * D2;  // msqg/qg.c:106
  }
  __shared__ double sm[BY];
  v = wave_sum(v);
  if (threadIdx.x == 0) sm[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double s = 0.;
    for (int k = 0; k < BY; k++) s += sm[k];
    partial[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}
__global__ void k_sum_field_partial(const double *__restrict__ f, double *partial, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  for (int l = 0; l < nl; l++) {
    double v = (i < g.nx && j < g.ny) ? f[nat_idx(g, l, j, i)] : 0.;
    __shared__ double sm[BY];
    v = wave_sum(v);
    if (threadIdx.x == 0) sm[threadIdx.y] = v;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
      double s = 0.;
      for (int k = 0; k < BY; k++) s += sm[k];
      partial[(size_t)l * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
    __syncthreads();
  }
}
// out[l] = sum of partial[l*n .. l*n+n)
__global__ void k_sum_final(const double *partial, double *out, int n) {
  __shared__ double sm[256];
  const double *p = partial + (size_t)blockIdx.x * n;
  double s = 0.;
  for (int k = threadIdx.x; k < n; k += 256) s += p[k];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}
// n > 4096: two stages (SUM_CHUNKS blocks over contiguous chunks, then one block over the chunk sums),
// fixed shape => deterministic.  The chunk sums live behind the partials (the caller's array holds
// at least n + SUM_CHUNKS entries).
#define SUM_CHUNKS 64
__global__ void k_sum_chunks(double *partial, int n) {
  __shared__ double sm[256];
  const int per = (n + SUM_CHUNKS - 1) / SUM_CHUNKS, k0 = blockIdx.x * per, k1 = min(n, k0 + per);
  double s = 0.;
  for (int k = k0 + threadIdx.x; k < k1; k += 256) s += partial[k];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[n + blockIdx.x] = sm[0];
}
void launch_sum_final(hipStream_t st, const double *partial, double *out, int n) {
  if (n > 4096) {
    hipLaunchKernelGGL(k_sum_chunks, dim3(SUM_CHUNKS), dim3(256), 0, st, const_cast<double *>(partial), n);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, st, partial + n, out, SUM_CHUNKS);
    return;
  }
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, st, partial, out, n);
}
int partial_count(const NatGeom &g) {
  dim3 gr = grid2d(g.nx, g.ny);
  return gr.x * gr.y;
}
void launch_ke(hipStream_t st, const double *po, double *partial, double *out, const NatGeom &g, double D) {
  const double D2 = D * D;
  dim3 gr = grid2d(g.nx, g.ny);
  hipLaunchKernelGGL(k_ke_partial, gr, block2d(), 0, st, po, partial, g, D2, 1. / D2);
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, st, partial, out, (int)(gr.x * gr.y));
}
void launch_sum_layers(hipStream_t st, const double *f, double *partial, double *out, const NatGeom &g, int nl) {
  dim3 gr = grid2d(g.nx, g.ny);
  hipLaunchKernelGGL(k_sum_field_partial, gr, block2d(), 0, st, f, partial, g, nl);
  hipLaunchKernelGGL(k_sum_final, dim3(nl), dim3(256), 0, st, partial, out, (int)(gr.x * gr.y));
}
// f[l] -= mean[l]
__global__ void k_sub_layer_const(double *f, const double *__restrict__ sums, NatGeom g, int nl, double inv_count) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) f[nat_idx(g, l, j, i)] -= sums[l] * inv_count;
}
void launch_sub_layer_const(hipStream_t st, double *f, const double *sums, const NatGeom &g, int nl, double inv_count) {
  hipLaunchKernelGGL(k_sub_layer_const, grid2d(g.nx, g.ny), block2d(), 0, st, f, sums, g, nl, inv_count);
}

// S = (Fr/Ro)^2, msqg/qg.h:1043-1048
__global__ void k_make_S(const double *__restrict__ Fr, const double *__restrict__ Ro, double *S, NatGeom g, int nlm) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const size_t c0 = nat_idx(g, 0, j, i);
  for (int l = 0; l < nlm; l++) {
    const double r = Fr[c0 + (size_t)l * g.ls] / Ro[c0];
    S[c0 + (size_t)l * g.ls] = r * r;
  }
}
void launch_make_S(hipStream_t st, const double *Fr, const double *Ro, double *S, const NatGeom &g, int nlm) {
  hipLaunchKernelGGL(k_make_S, grid2d(g.nx, g.ny), block2d(), 0, st, Fr, Ro, S, g, nlm);
}

// ------------------------------------------------------------------ K14 stochastic forcing noise

// n = amp * sigma(x) * N(0,1) per point-layer (generate_noise, msqg/qg_stochastic.h:117-126).
// The reference draws from the serial rand() stream in foreach order, which no parallel
// machine can reproduce; noise_mode 0 replays exactly that stream on the host (parity tests),
// noise_mode 1 (this kernel) uses a counter-based generator: Philox-4x32-10 keyed by the seed,
// counter = (global cell index, layer, draw number), then the reference's Box-Muller formula
// on two uniforms quantised to rand()'s 31 bits.  Results are independent of tiling and of
// the launch geometry; parity with the reference is statistical only.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__global__ void k_noise(double *n, const double *__restrict__ sigma, NatGeom g, int nl, double amp, unsigned seed, unsigned draw, int gx0, int gy0,
                        int gnx) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const uint32_t cell = (uint32_t)((size_t)(gy0 + j) * gnx + (gx0 + i));
  for (int l = 0; l < nl; l++) {
    uint32_t c[4] = {cell, (uint32_t)l, draw, 0x6d736f6du};
    philox4x32_10(c, seed, 0x4d493335u);
    const double r1 = (double)(c[0] >> 1), r2 = (double)(c[1] >> 1), RM = 2147483647.;  // RAND_MAX
    const double a = sqrt(-2. * log((r1 + 1.) / (RM + 2.)));
    const size_t k = nat_idx(g, l, j, i);
    n[k] = amp * sigma[k] * (a * cos(2 * 3.14159265358979323846 * r2 / RM));
  }
}
void launch_noise(hipStream_t st, double *n, const double *sigma, const NatGeom &g, int nl, double amp, unsigned seed, unsigned draw, int gx0,
                  int gy0, int gnx) {
  hipLaunchKernelGGL(k_noise, grid2d(g.nx, g.ny), block2d(), 0, st, n, sigma, g, nl, amp, seed, draw, gx0, gy0, gnx);
}

// ------------------------------------------------------------------ passive tracers

// ptr_rhs, msqg/qg.h:574-588: dpdt += -J(psi_l, c) + c-diffusion + relaxation, tracer fields
// stored [l * nptr + nt]
struct PtrCoef { double iPe[MSOM_MAXNL], ptr_ir[MSOM_MAXNL]; };
__global__ void k_ptr_rhs(const double *__restrict__ psi, const double *__restrict__ c, const double *__restrict__ rel, double *dp, NatGeom g,
                          int nl, int np, PtrCoef pc, double D) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const double D2 = D * D, rD2 = 1. / D2, D12 = 12. * D * D, rD12 = 1. / D12;
  const size_t c0 = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++)
    for (int nt = 0; nt < np; nt++) {
      const size_t k = c0 + (size_t)(l * np + nt) * g.ls;
      const double *cl = c + (size_t)(l * np + nt) * g.ls;
      dp[k] += mjac(psi + (size_t)l * g.ls, cl, c0, g.pitch, D12, rD12) + pc.iPe[nt] * DIVC(LAPV(cl, c0, g.pitch), D2, rD2) +
               pc.ptr_ir[nt] * (rel[k] - c[k]);
    }
}
void launch_ptr_rhs(hipStream_t st, const double *psi, const double *c, const double *rel, double *dp, const NatGeom &g, int nl, int np,
                    const double *iPe, const double *ptr_ir, double D) {
  PtrCoef pc;
  for (int k = 0; k < MSOM_MAXNL; k++) { pc.iPe[k] = k < np ? iPe[k] : 0.; pc.ptr_ir[k] = k < np ? ptr_ir[k] : 0.; }
  hipLaunchKernelGGL(k_ptr_rhs, grid2d(g.nx, g.ny), block2d(), 0, st, psi, c, rel, dp, g, nl, np, pc, D);
}

// ------------------------------------------------------------------ energy / PV budgets, msqg/qg_energy.h
// Every term of the PV equation times dt * w, w = -psi (1 - ediag) + ediag, accumulated into
// de_j1 / de_j2 / de_j3 (advection_de :28-158), de_vd (dissip_de :161-191), de_bf
// (ekman_friction_de :193-206).  Diagnostics: straightforward one-thread-per-column kernels.
#define EWGT(po, c) (-(po)[c] * (1 - ediag) + ediag)
struct AdvDeArgs {
  const double *zeta, *psi, *psipg, *zetapg, *S;
  double *j1, *j2, *j3;
  NatGeom g;
  int nl;
  double D, beta, dt, ediag;
  LayerCoef lc;
};
__global__ void k_advection_de(AdvDeArgs a) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= a.g.nx || j >= a.g.ny) return;
  const int pitch = a.g.pitch, nl = a.nl;
  const size_t ls = a.g.ls;
  const double D12 = 12. * a.D * a.D, rD12 = 1. / D12, D2x = 2 * a.D, rD2x = 1. / D2x, dt = a.dt, ediag = a.ediag;
  size_t c = nat_idx(a.g, 0, j, i);
  const double *po = a.psi, *qo = a.zeta, *pp = a.psipg, *qp = a.zetapg;
  if (nl == 1) { a.j1[c] = 0; a.j2[c] = 0; a.j3[c] = 0; return; }  // :150-157
  double ju_1, jd_1 = 0., ju_2, jd_2 = 0., ju_3, jd_3 = 0.;
  for (int l = 0; l < nl; l++, c += ls) {
    ju_1 = -jd_1; ju_2 = -jd_3; ju_3 = -jd_2;  // swap, :96-98
    if (l < nl - 1) {
      jd_1 = mjac(po, po + ls, c, pitch, D12, rD12);
      jd_2 = mjac(pp, po + ls, c, pitch, D12, rD12);
      jd_3 = mjac(po, pp + ls, c, pitch, D12, rD12);
    }
    const double jc = mjac(po, pp, c, pitch, D12, rD12), w = EWGT(po, c);
    const double be = DIVC(a.beta * (po[c - 1] - po[c + 1]), D2x, rD2x);
    double t1 = mjac(po, qo, c, pitch, D12, rD12), t2 = mjac(pp, qo, c, pitch, D12, rD12), t3 = be;
    if (l > 0) {
      const double s0 = a.S[c - ls], i0 = a.lc.idh0[l];
      t1 = t1 + s0 * ju_1 * i0; t2 = t2 + s0 * (ju_2 + jc) * i0; t3 = t3 + s0 * (ju_3 - jc) * i0;
    }
    if (l < nl - 1) {
      const double s1 = a.S[c], i1 = a.lc.idh1[l];
      t1 = t1 + s1 * jd_1 * i1; t2 = t2 + s1 * (jd_2 + jc) * i1; t3 = t3 + s1 * (jd_3 - jc) * i1;
    }
    a.j1[c] += t1 * dt * w;
    a.j2[c] += t2 * dt * w;
    double v3 = a.j3[c];
    v3 += t3 * dt * w;
    v3 += mjac(po, qp, c, pitch, D12, rD12) * dt * w;
    a.j3[c] = v3;
  }
}
void launch_advection_de(hipStream_t st, const double *zeta, const double *psi, const double *psipg, const double *zetapg, const double *S, double *j1,
                         double *j2, double *j3, const NatGeom &g, int nl, double D, double beta, double dt, double ediag, const LayerCoef &lc) {
  AdvDeArgs a;
  a.zeta = zeta; a.psi = psi; a.psipg = psipg; a.zetapg = zetapg; a.S = S; a.j1 = j1; a.j2 = j2; a.j3 = j3; a.g = g; a.nl = nl; a.D = D; a.beta = beta;
  a.dt = dt; a.ediag = ediag; a.lc = lc;
  hipLaunchKernelGGL(k_advection_de, grid2d(g.nx, g.ny), block2d(), 0, st, a);
}
// stage 0: dq += (p4 + str) iRe dt w; dq += iRe4 lap(p4) dt w   stage 1: dq += iRe4 str dt w
__global__ void k_dissip_de(const double *__restrict__ p4, const double *__restrict__ str, const double *__restrict__ po, double *dq, NatGeom g, int nl,
                            double iRe, double iRe4, double dt, double ediag, double D2, double rD2, int stage) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) {
    const double w = EWGT(po, c);
    double d = dq[c];
    if (stage == 0) {
      d += (p4[c] + str[c]) * iRe * dt * w;
      d += iRe4 * DIVC(LAPV(p4, c, g.pitch), D2, rD2) * dt * w;
    } else
      d += iRe4 * (str[c]) * dt * w;
    dq[c] = d;
  }
}
void launch_dissip_de(hipStream_t st, const double *p4, const double *str, const double *po, double *dq, const NatGeom &g, int nl, double iRe,
                      double iRe4, double dt, double ediag, double D, int stage) {
  hipLaunchKernelGGL(k_dissip_de, grid2d(g.nx, g.ny), block2d(), 0, st, p4, str, po, dq, g, nl, iRe, iRe4, dt, ediag, D * D, 1. / (D * D), stage);
}
__global__ void k_ekman_de(const double *__restrict__ zeta, const double *__restrict__ po, double *dq, NatGeom g, int nl, double cs, double cb, double dt,
                           double ediag) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  const size_t c0 = nat_idx(g, 0, j, i), cbt = c0 + (size_t)(nl - 1) * g.ls;
  dq[c0] -= cs * zeta[c0] * dt * EWGT(po, c0);
  dq[cbt] -= cb * zeta[cbt] * dt * EWGT(po, cbt);
}
void launch_ekman_de(hipStream_t st, const double *zeta, const double *po, double *dq, const NatGeom &g, int nl, double cs, double cb, double dt,
                     double ediag) {
  hipLaunchKernelGGL(k_ekman_de, grid2d(g.nx, g.ny), block2d(), 0, st, zeta, po, dq, g, nl, cs, cb, dt, ediag);
}
// pm = (pm * n + po) / (n + 1)   energy_tend :234-239
__global__ void k_running_mean(double *pm, const double *__restrict__ po, NatGeom g, int nl, int n) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) pm[c] = (pm[c] * n + po[c]) / (n + 1);
}
void launch_running_mean(hipStream_t st, double *pm, const double *po, const NatGeom &g, int nl, int n) {
  hipLaunchKernelGGL(k_running_mean, grid2d(g.nx, g.ny), block2d(), 0, st, pm, po, g, nl, n);
}
// de_ft += tmp2 dtflt (-pm (1 - ediag) + ediag); pm = 0   filter_de :214-223
__global__ void k_filter_de(double *ft, const double *__restrict__ tmp2, double *pm, NatGeom g, int nl, double dtflt, double ediag) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  size_t c = nat_idx(g, 0, j, i);
  for (int l = 0; l < nl; l++, c += g.ls) {
    ft[c] += tmp2[c] * dtflt * (-pm[c] * (1 - ediag) + ediag);
    pm[c] = 0;
  }
}
void launch_filter_de(hipStream_t st, double *ft, const double *tmp2, double *pm, const NatGeom &g, int nl, double dtflt, double ediag) {
  hipLaunchKernelGGL(k_filter_de, grid2d(g.nx, g.ny), block2d(), 0, st, ft, tmp2, pm, g, nl, dtflt, ediag);
}
