// kernels_mg.hip -- multigrid kernels of the layered elliptic solver (gfx950 / CDNA4, fp64).
//
// Solves lap(a)_l + Gamma(a)_l = b_l (msqg/poisson_layer.h) with the reference's cycle
// (mspg/elliptic.h:43-99): residual restricted to all levels, then coarse -> fine:
// bilinear prolongation of the correction + nrelax relaxations per level.
//
// Smoother: the reference's relax_layer (poisson_layer.h:48-150) is an in-place
// lexicographic Gauss-Seidel with an exact tridiagonal (Thomas) solve over the nl layers of
// each column.  Lexicographic order has no parallelism, so the GPU uses the same column
// solve in red-black order: red = (i + j) even, then black.  One thread owns one column and
// keeps the whole tridiagonal system in registers (template on NL).
//
// Layout: da / res / S on every level use the x-parity split layout (msom_internal.h): in
// row j the points of colour c are exactly the half row ((j + c) & 1), so a colour
// half-sweep reads the other colour's half rows (neighbours), its own half rows of res (and
// S), and writes its own half rows of da -- all contiguous, coalesced 64-lane accesses, and
// a full sweep moves the compulsory (3 + (nl-1)/nl) * 8 * nx*ny*nl bytes.
//
// Ghost cells: the homogeneous Dirichlet ghost of a wall cell is -value as of the last
// boundary_level() call.  A ghost is read only by the cell it mirrors, so the thread that
// updates a wall cell also rewrites its ghost(s); no separate boundary kernel is launched.
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

__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// writes the homogeneous-Dirichlet ghosts that mirror cell (i, j) (edges: -v, corners: +v)
__device__ __forceinline__ void split_write_ghosts(double *f, const SplitGeom &g, int l, int j, int i, double v, int walls) {
  const bool w = i == 0 && (walls & WALL_W), e = i == g.nx - 1 && (walls & WALL_E);
  const bool s = j == 0 && (walls & WALL_S), n = j == g.ny - 1 && (walls & WALL_N);
  if (!(w | e | s | n)) return;
  if (w) f[split_idx(g, l, j, -1)] = -v;
  if (e) f[split_idx(g, l, j, g.nx)] = -v;
  if (s) f[split_idx(g, l, -1, i)] = -v;
  if (n) f[split_idx(g, l, g.ny, i)] = -v;
  if (w && s) f[split_idx(g, l, -1, -1)] = v;
  if (w && n) f[split_idx(g, l, g.ny, -1)] = v;
  if (e && s) f[split_idx(g, l, -1, g.nx)] = v;
  if (e && n) f[split_idx(g, l, g.ny, g.nx)] = v;
}
__device__ __forceinline__ void nat_write_ghosts(double *f, const NatGeom &g, int l, int j, int i, double v, int walls) {
  const bool w = i == 0 && (walls & WALL_W), e = i == g.nx - 1 && (walls & WALL_E);
  const bool s = j == 0 && (walls & WALL_S), n = j == g.ny - 1 && (walls & WALL_N);
  if (!(w | e | s | n)) return;
  if (w) f[nat_idx(g, l, j, -1)] = -v;
  if (e) f[nat_idx(g, l, j, g.nx)] = -v;
  if (s) f[nat_idx(g, l, -1, i)] = -v;
  if (n) f[nat_idx(g, l, g.ny, i)] = -v;
  if (w && s) f[nat_idx(g, l, -1, -1)] = v;
  if (w && n) f[nat_idx(g, l, g.ny, -1)] = v;
  if (e && s) f[nat_idx(g, l, -1, g.nx)] = v;
  if (e && n) f[nat_idx(g, l, g.ny, g.nx)] = v;
}

// ------------------------------------------------------------------ layout conversion

__global__ void k_nat_to_split(const double *__restrict__ nat, NatGeom g, double *sp, SplitGeom sg, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) sp[split_idx(sg, l, j, i)] = nat[nat_idx(g, l, j, i)];
}
__global__ void k_split_to_nat(const double *__restrict__ sp, SplitGeom sg, double *nat, NatGeom g, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) nat[nat_idx(g, l, j, i)] = sp[split_idx(sg, l, j, i)];
}
void launch_nat_to_split(hipStream_t st, const double *nat, const NatGeom &g, double *sp, const SplitGeom &sg, int nl) {
  hipLaunchKernelGGL(k_nat_to_split, grid2d(g.nx, g.ny), block2d(), 0, st, nat, g, sp, sg, nl);
}
void launch_split_to_nat(hipStream_t st, const double *sp, const SplitGeom &sg, double *nat, const NatGeom &g, int nl) {
  hipLaunchKernelGGL(k_split_to_nat, grid2d(g.nx, g.ny), block2d(), 0, st, sp, sg, nat, g, nl);
}
// contiguous [layer][y][x] of a level -> split layout (+ homogeneous Dirichlet ghosts)
__global__ void k_split_pack(const double *__restrict__ src, double *sp, SplitGeom sg, int nl, int bc, int walls) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= sg.nx || j >= sg.ny) return;
  for (int l = 0; l < nl; l++) {
    const double v = src[((size_t)l * sg.ny + j) * sg.nx + i];
    sp[split_idx(sg, l, j, i)] = v;
    if (bc == BC_DIRICHLET0) split_write_ghosts(sp, sg, l, j, i, v, walls);
  }
}
__global__ void k_split_unpack(const double *__restrict__ sp, SplitGeom sg, double *dst, int nl) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= sg.nx || j >= sg.ny) return;
  for (int l = 0; l < nl; l++) dst[((size_t)l * sg.ny + j) * sg.nx + i] = sp[split_idx(sg, l, j, i)];
}
void launch_split_pack(hipStream_t st, const double *src, double *sp, const SplitGeom &sg, int nl, int bc, int walls) {
  hipLaunchKernelGGL(k_split_pack, grid2d(sg.nx, sg.ny), block2d(), 0, st, src, sp, sg, nl, bc, walls);
}
void launch_split_unpack(hipStream_t st, const double *sp, const SplitGeom &sg, double *dst, int nl) {
  hipLaunchKernelGGL(k_split_unpack, grid2d(sg.nx, sg.ny), block2d(), 0, st, sp, sg, dst, nl);
}

// ------------------------------------------------------------------ K9 residual_layer

// res_l = b_l - Gamma(a)_l - lap(a)_l in the reference's form (poisson_layer.h:182-255);
// a, b natural (finest level), res split.  max |res| -> *maxres (atomic), optional
// deterministic per-block partial sums of b (mgstats.sum, mspg/elliptic.h:171-176).
struct ResArgs {
  const double *a, *b, *S;
  double *res, *maxres, *sum_partial;
  NatGeom g;
  SplitGeom sg;
  int nl, uniformS, want_sum;
  double D;
  RelaxCoef rc;
};
__global__ void k_residual(ResArgs p) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const bool in = i < p.g.nx && j < p.g.ny;
  double m = 0., bs = 0.;
  if (in) {
    const int pitch = p.g.pitch, nl = p.nl;
    const double D = p.D, rD = 1. / D;
    size_t c = nat_idx(p.g, 0, j, i);
    double a0 = 0., a1 = p.a[c], a2 = 0., s0 = 0., s1 = 0.;
    for (int l = 0; l < nl; l++, c += p.g.ls) {
      const double b = p.b[c];
      bs += b;
      double r = b;
      if (l < nl - 1) { a2 = p.a[c + p.g.ls]; s1 = p.uniformS ? p.rc.S[l] : p.S[c]; }
      if (nl > 1) {
        if (l == 0) r = b + s1 * (a1 - a2) * p.rc.idh1[l];
        else if (l < nl - 1) r = b + s0 * (a1 - a0) * p.rc.idh0[l] - s1 * (a2 - a1) * p.rc.idh1[l];
        else r = b + s0 * (a1 - a0) * p.rc.idh0[l];
      }
      const double aw = p.a[c - 1], ae = p.a[c + 1], as = p.a[c - pitch], an = p.a[c + pitch];
      r += DIVC(DIVC(a1 - aw, D, rD) - DIVC(ae - a1, D, rD), D, rD);
      r += DIVC(DIVC(a1 - as, D, rD) - DIVC(an - a1, D, rD), D, rD);
      p.res[split_idx(p.sg, l, j, i)] = r;
      m = fmax(m, fabs(r));
      a0 = a1; a1 = a2; s0 = s1;
    }
  }
  __shared__ double smm[BY], sms[BY];
  m = wave_max(m);
  if (p.want_sum) bs = wave_sum(bs);
  if (threadIdx.x == 0) { smm[threadIdx.y] = m; sms[threadIdx.y] = bs; }
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double mm = smm[0], ss = sms[0];
    for (int k = 1; k < BY; k++) { mm = fmax(mm, smm[k]); ss += sms[k]; }
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(mm));
    if (p.want_sum) p.sum_partial[blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}
void launch_residual(hipStream_t st, const double *a, const double *b, const double *S, const NatGeom &g, double *res, const SplitGeom &sg,
                     int nl, const RelaxCoef &rc, int uniformS, double *maxres, double *sum_partial, int want_sum) {
  ResArgs p;
  p.a = a; p.b = b; p.S = S; p.res = res; p.maxres = maxres; p.sum_partial = sum_partial;
  p.g = g; p.sg = sg; p.nl = nl; p.uniformS = uniformS; p.want_sum = want_sum; p.D = rc.D; p.rc = rc;
  hipLaunchKernelGGL(k_residual, grid2d(g.nx, g.ny), block2d(), 0, st, p);
}

// ------------------------------------------------------------------ K10 restriction, K11 prolongation

// coarse = mean of the 4 children, summed in Basilisk's foreach_child order
__global__ void k_restrict(const double *__restrict__ fine, SplitGeom fg, double *coarse, SplitGeom cg, int nl) {
  const int I = blockIdx.x * BX + threadIdx.x, J = blockIdx.y * BY + threadIdx.y;
  if (I >= cg.nx || J >= cg.ny) return;
  for (int l = 0; l < nl; l++) {
    double sum = 0.;
    sum += fine[split_idx(fg, l, 2 * J, 2 * I)];
    sum += fine[split_idx(fg, l, 2 * J + 1, 2 * I)];
    sum += fine[split_idx(fg, l, 2 * J, 2 * I + 1)];
    sum += fine[split_idx(fg, l, 2 * J + 1, 2 * I + 1)];
    coarse[split_idx(cg, l, J, I)] = sum / 4;
  }
}
void launch_restrict(hipStream_t st, const double *fine, const SplitGeom &fg, double *coarse, const SplitGeom &cg, int nl) {
  hipLaunchKernelGGL(k_restrict, grid2d(cg.nx, cg.ny), block2d(), 0, st, fine, fg, coarse, cg, nl);
}

// bilinear: (9 c + 3 (c[child.x] + c[0,child.y]) + c[child.x,child.y]) / 16, then
// boundary_level(da) on the fine level (ghosts written by the wall threads)
__global__ void k_prolong(const double *__restrict__ coarse, SplitGeom cg, double *fine, SplitGeom fg, int nl, int walls) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= fg.nx || j >= fg.ny) return;
  const int I = i >> 1, J = j >> 1, cx = (i & 1) ? 1 : -1, cy = (j & 1) ? 1 : -1;
  for (int l = 0; l < nl; l++) {
    const double v = (9. * coarse[split_idx(cg, l, J, I)] +
                      3. * (coarse[split_idx(cg, l, J, I + cx)] + coarse[split_idx(cg, l, J + cy, I)]) +
                      coarse[split_idx(cg, l, J + cy, I + cx)]) / 16.;
    fine[split_idx(fg, l, j, i)] = v;
    split_write_ghosts(fine, fg, l, j, i, v, walls);
  }
}
void launch_prolong(hipStream_t st, const double *coarse, const SplitGeom &cg, double *fine, const SplitGeom &fg, int nl, int walls) {
  hipLaunchKernelGGL(k_prolong, grid2d(fg.nx, fg.ny), block2d(), 0, st, coarse, cg, fine, fg, nl, walls);
}

// ------------------------------------------------------------------ K8 relax_layer (red-black column solve)

struct RelaxArgs {
  double *da;
  const double *res, *S;
  SplitGeom g;
  int color, walls;
  RelaxCoef rc;
};

// FINE tags the instantiation launched on the finest level (level 0): a distinct kernel
// symbol, so that profiler statistics of the HBM-bound fine sweep are not averaged with the
// launch-latency-bound coarse levels.
template <int NL, bool UNIFORM, bool FINE>
__global__ void __launch_bounds__(BX *BY) k_relax_color(RelaxArgs p) {
  const int kx = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (kx >= p.g.hk || j >= p.g.ny) return;
  const int px = (j + p.color) & 1;  // x parity of this colour's points in row j
  const int i = 2 * kx + px;
  const int hp = p.g.hp, rp = p.g.rp;
  const size_t ls = p.g.ls;
  // own cell, and the four neighbours (other colour): W/E live in the other half of row j,
  // S/N in the same half of rows j -+ 1
  const size_t own = (size_t)(j + 1) * rp + px * hp + MSOM_SP + kx;
  const size_t oth = (size_t)(j + 1) * rp + (1 - px) * hp + MSOM_SP + kx;
  const size_t iw = oth - 1 + px, ie = oth + px, is = own - rp, in = own + rp;
  const double sqD = p.rc.sqD;
  double rhs[NL], x[NL];
  if (NL == 1) {
    // nl == 1: reference body is empty (poisson_layer.h:80); plain Poisson relaxation
    double n = -sqD * p.res[own], d = 0.;
    n += p.da[ie] + p.da[iw]; d += 2.;
    n += p.da[in] + p.da[is]; d += 2.;
    x[0] = n / d;
  } else if (UNIFORM) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      double r = -sqD * p.res[own + l * ls];
      r += p.da[ie + l * ls] + p.da[iw + l * ls];
      r += p.da[in + l * ls] + p.da[is + l * ls];
      rhs[l] = r;
    }
#pragma unroll
    for (int l = 1; l < NL; l++) rhs[l] -= p.rc.w[l] * rhs[l - 1];
    x[NL - 1] = rhs[NL - 1] * p.rc.it1[NL - 1];
#pragma unroll
    for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - p.rc.t2[l] * x[l + 1]) * p.rc.it1[l];
  } else {
    double t0[NL], t1[NL], t2[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      rhs[l] = -sqD * p.res[own + l * ls];
      t0[l] = l > 0 ? -sqD * p.S[own + (l - 1) * ls] * p.rc.idh0[l] : 0.;
      t2[l] = l < NL - 1 ? -sqD * p.S[own + l * ls] * p.rc.idh1[l] : 0.;
      t1[l] = l == 0 ? -t2[l] : (l < NL - 1 ? -t0[l] - t2[l] : -t0[l]);
      rhs[l] += 1. * p.da[ie + l * ls] + 1. * p.da[iw + l * ls];
      t1[l] += 1. + 1.;
      rhs[l] += 1. * p.da[in + l * ls] + 1. * p.da[is + l * ls];
      t1[l] += 1. + 1.;
    }
#pragma unroll
    for (int l = 1; l < NL; l++) {
      t1[l] -= t0[l] * t2[l - 1] / t1[l - 1];
      rhs[l] -= t0[l] * rhs[l - 1] / t1[l - 1];
    }
    x[NL - 1] = rhs[NL - 1] / t1[NL - 1];
#pragma unroll
    for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - t2[l] * x[l + 1]) / t1[l];
  }
#pragma unroll
  for (int l = 0; l < NL; l++) p.da[own + l * ls] = x[l];
  const bool edge = (i == 0) | (i == p.g.nx - 1) | (j == 0) | (j == p.g.ny - 1);
  if (edge && p.walls) {
#pragma unroll
    for (int l = 0; l < NL; l++) split_write_ghosts(p.da, p.g, l, j, i, x[l], p.walls);
  }
}

template <int NL>
static void relax_dispatch(hipStream_t st, const RelaxArgs &p, int uniformS, int fine) {
  dim3 gr = grid2d(p.g.hk, p.g.ny);
  if (uniformS) {
    if (fine) hipLaunchKernelGGL((k_relax_color<NL, true, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_color<NL, true, false>), gr, block2d(), 0, st, p);
  } else {
    if (fine) hipLaunchKernelGGL((k_relax_color<NL, false, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_color<NL, false, false>), gr, block2d(), 0, st, p);
  }
}
void launch_relax_color(hipStream_t st, double *da, const double *res, const double *S, const SplitGeom &sg, int nl, const RelaxCoef &rc,
                        int uniformS, int color, int walls, int fine) {
  RelaxArgs p;
  p.da = da; p.res = res; p.S = S; p.g = sg; p.color = color; p.walls = walls; p.rc = rc;
  switch (nl) {
    case 1: relax_dispatch<1>(st, p, uniformS, fine); break;
    case 2: relax_dispatch<2>(st, p, uniformS, fine); break;
    case 3: relax_dispatch<3>(st, p, uniformS, fine); break;
    case 4: relax_dispatch<4>(st, p, uniformS, fine); break;
    case 5: relax_dispatch<5>(st, p, uniformS, fine); break;
    case 6: relax_dispatch<6>(st, p, uniformS, fine); break;
    case 7: relax_dispatch<7>(st, p, uniformS, fine); break;
    case 8: relax_dispatch<8>(st, p, uniformS, fine); break;
    default: break;  // rejected at create time (MSOM_MAXNL)
  }
}

// ------------------------------------------------------------------ K12 correction a += da (+ boundary(a))

__global__ void k_correct(double *a, NatGeom g, const double *__restrict__ da, SplitGeom sg, int nl, int walls) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= g.nx || j >= g.ny) return;
  for (int l = 0; l < nl; l++) {
    const size_t c = nat_idx(g, l, j, i);
    const double v = a[c] + da[split_idx(sg, l, j, i)];
    a[c] = v;
    nat_write_ghosts(a, g, l, j, i, v, walls);
  }
}
void launch_correct(hipStream_t st, double *a, const NatGeom &g, const double *da, const SplitGeom &sg, int nl, int walls) {
  hipLaunchKernelGGL(k_correct, grid2d(g.nx, g.ny), block2d(), 0, st, a, g, da, sg, nl, walls);
}
