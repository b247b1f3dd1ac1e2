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
#include "mg_inl.h"
#include "rhs_inl.h"  // DIVC, whole-wave DPP shifts


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
    if (bc != BC_NEUMANN) split_write_ghosts(sp, sg, l, j, i, v, walls);
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

// ------------------------------------------------------------------ fused residual variants

// Two adjacent cells (x = 2kx, 2kx+1) per thread: 16-byte loads from the natural arrays and
// fully coalesced stores into the two halves of the split layout.  Template flags fuse the
// neighbouring steps of the cycle into the same pass:
//   CORRECT  : a_new = a + da (mg_cycle's "a += da; boundary(a)", mspg/elliptic.h:92-98) is
//              applied on the fly; a_new goes to a second buffer (no in-place race with the
//              neighbours' reads) and the residual is that of a_new;
//   WRITE    : store the residual (split layout);  without it only max|res| is produced
//              (the post-cycle convergence check needs nothing else when it passes);
//   RESTRICT : also store the residual restricted to the next coarser level (mean of the 4
//              children in foreach_child order), saving the first restriction pass.
struct Res2Args {
  const double *a, *b, *S, *da;
  double *a_out, *res, *res_c, *maxres, *sum_partial, *umax_partial;
  NatGeom g;
  SplitGeom sg, cg;
  int nl, uniformS, want_sum, walls, dbg;
  double *res_c2;   // RESTRICT: non-null = the restriction of res_c to the next level as well (geometry cg2); needs ny % 4 == 0, hk even
  SplitGeom cg2;
  RelaxCoef rc;
};

// UMAX (without CORRECT): max |u| of a itself, for the pass that follows a correction done elsewhere (kernels_march.hip)
template <bool CORRECT, bool WRITE, bool RESTRICT, bool UMAX = false>
__global__ void __launch_bounds__(BX *BY) k_residual2(Res2Args p) {
  // RESTRICT: the residuals of the block's 4 x 128 cells, nl layers (dynamic LDS, nl x 4 KB: a static array for MSOM_MAXNL layers
  // was 64 KB and held the kernel at two workgroups per CU), then its level-1 means [nl][2][BX]
  extern __shared__ double sr_dyn[];
  auto sr = [&](int l, int y, int x, int c) -> double & { return sr_dyn[((l * BY + y) * BX + x) * 2 + c]; };
  auto s1 = [&](int l, int y, int x) -> double & { return sr_dyn[p.nl * BY * BX * 2 + (l * 2 + y) * BX + x]; };
  __shared__ double smm[BY], sms[BY];
  constexpr bool VEL = CORRECT || UMAX;
  __shared__ double smu[VEL ? MSOM_MAXNL : 1][BY];
  const int kx = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const bool in = kx < p.sg.hk && j < p.g.ny;
  const int nl = p.nl;
  double m = 0., bs = 0.;
  double um[VEL ? MSOM_MAXNL : 1];
#pragma unroll
  for (int l = 0; l < (VEL ? MSOM_MAXNL : 1); l++) um[l] = 0.;
  if (in) {
    const int pitch = p.g.pitch, i = 2 * kx;
    const double D = p.rc.D, rD = 1. / D;
    size_t c = nat_idx(p.g, 0, j, i);
    // split-layout offsets of da: even/odd half entries of this pair and their neighbours
    const size_t se = split_idx(p.sg, 0, j, i), so = se + p.sg.hp;
    auto ld2 = [&](const double *f, size_t k, double &x, double &y) {
      const double2 v = *reinterpret_cast<const double2 *>(f + k);
      x = v.x; y = v.y;
    };
    // value of a (+ da) at layer offset lo: centre pair, W, E, S pair, N pair
    auto fetch = [&](size_t cl, size_t sl, double &ae, double &ao, double &aw, double &aee, double &se_, double &so_, double &ne_, double &no_) {
      ld2(p.a, cl, ae, ao);
      aw = p.a[cl - 1]; aee = p.a[cl + 2];
      ld2(p.a, cl - pitch, se_, so_);
      ld2(p.a, cl + pitch, ne_, no_);
      if (CORRECT) {
        const double *d = p.da + sl;
        ae = ae + d[se]; ao = ao + d[so];
        aw = aw + d[so - 1]; aee = aee + d[se + 1];
        se_ = se_ + d[se - p.sg.rp]; so_ = so_ + d[so - p.sg.rp];
        ne_ = ne_ + d[se + p.sg.rp]; no_ = no_ + d[so + p.sg.rp];
      }
    };
    double a0e = 0., a0o = 0., a1e, a1o, a1w, a1ee, a1se, a1so, a1ne, a1no;
    double a2e = 0., a2o = 0., a2w = 0., a2ee = 0., a2se = 0., a2so = 0., a2ne = 0., a2no = 0.;
    double s0e = 0., s0o = 0., s1e = 0., s1o = 0.;
    fetch(c, 0, a1e, a1o, a1w, a1ee, a1se, a1so, a1ne, a1no);
    for (int l = 0; l < nl; l++, c += p.g.ls) {
      double be, bo;
      ld2(p.b, c, be, bo);
      bs += be + bo;
      if (l < nl - 1) {
        fetch(c + p.g.ls, (size_t)(l + 1) * p.sg.ls, a2e, a2o, a2w, a2ee, a2se, a2so, a2ne, a2no);
        if (p.uniformS) s1e = s1o = p.rc.S[l];
        else ld2(p.S, c, s1e, s1o);
      }
      double re = be, ro = bo;
      if (nl > 1) {
        if (l == 0) {
          re = be + s1e * (a1e - a2e) * p.rc.idh1[l];
          ro = bo + s1o * (a1o - a2o) * p.rc.idh1[l];
        } else if (l < nl - 1) {
          re = be + s0e * (a1e - a0e) * p.rc.idh0[l] - s1e * (a2e - a1e) * p.rc.idh1[l];
          ro = bo + s0o * (a1o - a0o) * p.rc.idh0[l] - s1o * (a2o - a1o) * p.rc.idh1[l];
        } else {
          re = be + s0e * (a1e - a0e) * p.rc.idh0[l];
          ro = bo + s0o * (a1o - a0o) * p.rc.idh0[l];
        }
      }
      re += DIVC(DIVC(a1e - a1w, D, rD) - DIVC(a1o - a1e, D, rD), D, rD);
      re += DIVC(DIVC(a1e - a1se, D, rD) - DIVC(a1ne - a1e, D, rD), D, rD);
      ro += DIVC(DIVC(a1o - a1e, D, rD) - DIVC(a1ee - a1o, D, rD), D, rD);
      ro += DIVC(DIVC(a1o - a1so, D, rD) - DIVC(a1no - a1o, D, rD), D, rD);
      if (CORRECT) {
        *reinterpret_cast<double2 *>(p.a_out + c) = make_double2(a1e, a1o);
        nat_write_ghosts(p.a_out, p.g, l, j, i, a1e, p.walls);
        nat_write_ghosts(p.a_out, p.g, l, j, i + 1, a1o, p.walls);
      }
      if (WRITE) {
        p.res[se + (size_t)l * p.sg.ls] = re;
        p.res[so + (size_t)l * p.sg.ls] = ro;
      }
      if (RESTRICT) { sr(l, threadIdx.y, threadIdx.x, 0) = re; sr(l, threadIdx.y, threadIdx.x, 1) = ro; }
      m = fmax(m, fmax(fabs(re), fabs(ro)));
      if (VEL && !(p.dbg & 64)) {
        // face velocities of the corrected psi (comp_vel, msqg/qg.h:276-283): west and south face
        // of both cells; feeds the dt limiter so that dt is known before the tendency pass
        double nw = p.a[c - 1 + pitch], sw = p.a[c - 1 - pitch], se2 = p.a[c + 2 - pitch];
        if (CORRECT) {
          const double *d = p.da + (size_t)l * p.sg.ls;
          nw = nw + d[so - 1 + p.sg.rp]; sw = sw + d[so - 1 - p.sg.rp]; se2 = se2 + d[se + 1 - p.sg.rp];
        }
        const double ue = fabs(DIVC(0.25 * (a1ne - a1se + nw - sw), D, rD)), ve = fabs(DIVC(0.25 * (a1o - a1w + a1so - sw), D, rD));
        const double uo = fabs(DIVC(0.25 * (a1no - a1so + a1ne - a1se), D, rD)), vo = fabs(DIVC(0.25 * (a1ee - a1e + se2 - a1se), D, rD));
        const double uu = fmax(fmax(ue, ve), fmax(uo, vo));
#pragma unroll
        for (int q = 0; q < MSOM_MAXNL; q++)
          if (q == l) um[q] = uu;
      }
      a0e = a1e; a0o = a1o; s0e = s1e; s0o = s1o;
      a1e = a2e; a1o = a2o; a1w = a2w; a1ee = a2ee; a1se = a2se; a1so = a2so; a1ne = a2ne; a1no = a2no;
    }
  }
  if (RESTRICT) {
    __syncthreads();
    if (in && (threadIdx.y & 1) == 0) {
      const int J = j >> 1;
      for (int l = 0; l < nl; l++) {
        double sum = 0.;
        sum += sr(l, threadIdx.y, threadIdx.x, 0);
        sum += sr(l, threadIdx.y + 1, threadIdx.x, 0);
        sum += sr(l, threadIdx.y, threadIdx.x, 1);
        sum += sr(l, threadIdx.y + 1, threadIdx.x, 1);
        const double v = sum / 4;
        p.res_c[split_idx(p.cg, l, J, kx)] = v;
        if (p.res_c2) s1(l, threadIdx.y >> 1, threadIdx.x) = v;
      }
    }
    // round 3: the next restriction as well (was a launch of k_restrict reading the level-1 residual back): the block holds the
    // 2 x 64 level-1 cells of 1 x 32 level-2 cells; same sum order (restrict_pt), same values
    if (p.res_c2) {
      __syncthreads();
      const int II = threadIdx.x, JJ = j >> 2;
      if (threadIdx.y == 0 && II < BX / 2 && 2 * II + 1 + blockIdx.x * BX < p.sg.hk && j < p.g.ny) {
        for (int l = 0; l < nl; l++) {
          double sum = 0.;
          sum += s1(l, 0, 2 * II);
          sum += s1(l, 1, 2 * II);
          sum += s1(l, 0, 2 * II + 1);
          sum += s1(l, 1, 2 * II + 1);
          p.res_c2[split_idx(p.cg2, l, JJ, blockIdx.x * (BX / 2) + II)] = sum / 4;
        }
      }
    }
  }
  m = wave_max(m);
  if (p.want_sum) bs = wave_sum(bs);
  if (VEL) {
#pragma unroll
    for (int l = 0; l < MSOM_MAXNL; l++)
      if (l < nl) {
        const double v = wave_max(um[l]);
        if (threadIdx.x == 0) smu[l][threadIdx.y] = v;
      }
  }
  if (threadIdx.x == 0) { smm[threadIdx.y] = m; sms[threadIdx.y] = bs; }
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    double mm = smm[0], ss = sms[0];
    for (int k = 1; k < BY; k++) { mm = fmax(mm, smm[k]); ss += sms[k]; }
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(mm));
    if (p.want_sum) p.sum_partial[blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
  if (VEL && threadIdx.y == 0 && threadIdx.x < nl) {
    double v = smu[threadIdx.x][0];
    for (int k = 1; k < BY; k++) v = fmax(v, smu[threadIdx.x][k]);
    p.umax_partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * nl + threadIdx.x] = v;
  }
}

// LDS-tiled form of k_residual2<CORRECT = true, WRITE = false, RESTRICT = false> for the finest level:
// a_new = a + da is formed ONCE per cell (own cells + a 1-cell halo ring) in a (CR_TR + 2) x 130 LDS tile,
// double-buffered over the layers, and the 5-point residual, the vertical coupling and the face
// velocities (which need the diagonal neighbours) all read it from there: 4 global loads per pair and
// layer instead of ~20.  The arithmetic per cell is the same expression sequence as k_residual2, so the
// results are identical bit for bit.  Thread (tx, ty) owns the column pair 2 tx, 2 tx + 1 of rows
// ty, ty + 4, ...; one barrier per layer.
#define CR_TR 16
#define CR_TW 128
#define CR_LP 132  // LDS row pitch (130 used)
// CORR = false: a is already the corrected field (kernels_march.hip applied a += da): only max |res|, max |u|
template <bool UNIFORM, bool CORR = true>
__global__ void __launch_bounds__(BX *BY, UNIFORM ? 4 : 3) k_correct_residual(Res2Args p) {   // general S: 128 VGPRs spilled 37 registers (round 3: 3 waves per SIMD)
  __shared__ __align__(16) double T[2][CR_TR + 2][CR_LP];
  __shared__ double smm[BY];
  __shared__ double smu[MSOM_MAXNL][BY];
  constexpr int NR = CR_TR / BY;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * BX + tx;
  const int x0 = blockIdx.x * CR_TW, y0 = blockIdx.y * CR_TR;
  const int nl = p.nl, nx = p.g.nx, ny = p.g.ny;
  const int i = x0 + 2 * tx;
  const double D = p.rc.D, rD = 1. / D;
  const bool inx = i < nx;
  // row k of this thread: j = y0 + ty + BY k; natural / split offsets advance by BY rows
  const size_t c0 = nat_idx(p.g, 0, y0 + ty, inx ? i : 0), se0 = split_idx(p.sg, 0, y0 + ty, inx ? i : 0);
  const size_t cstep = (size_t)BY * p.g.pitch, sstep = (size_t)BY * p.sg.rp;
  // halo ring: rows -1 and CR_TR over columns -1..128, columns -1 and 128 over rows 0..CR_TR-1
  constexpr int NH = 2 * (CR_TW + 2) + 2 * CR_TR;
  double a0e[NR], a0o[NR], a1e[NR], a1o[NR], a2e[NR], a2o[NR];
  // own cells of layer l -> registers (xe, xo) and LDS buffer b; halo cells -> LDS
  auto stage = [&](int l, int b, double *xe, double *xo) {
    const double *al = p.a + (size_t)l * p.g.ls, *dl = p.da + (size_t)l * p.sg.ls;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      double e = 0., o = 0.;
      if (inx && y0 + ty + BY * k < ny) {
        const double2 v = *reinterpret_cast<const double2 *>(al + c0 + k * cstep);
        e = v.x; o = v.y;
        if (CORR) { e = e + dl[se0 + k * sstep]; o = o + dl[se0 + k * sstep + p.sg.hp]; }
      }
      xe[k] = e; xo[k] = o;
      *reinterpret_cast<double2 *>(&T[b][ty + BY * k + 1][2 * tx + 2]) = make_double2(e, o);
    }
    for (int h = tid; h < NH; h += BX * BY) {
      int r, cc;
      if (h < CR_TW + 2) { r = -1; cc = h - 1; }
      else if (h < 2 * (CR_TW + 2)) { r = CR_TR; cc = h - (CR_TW + 2) - 1; }
      else if (h < 2 * (CR_TW + 2) + CR_TR) { r = h - 2 * (CR_TW + 2); cc = -1; }
      else { r = h - 2 * (CR_TW + 2) - CR_TR; cc = CR_TW; }
      const int gx = x0 + cc, gy = y0 + r;
      double v = 0.;
      if (gx <= nx && gy <= ny) {
        v = al[nat_idx(p.g, 0, gy, gx)];
        if (CORR) v = v + dl[split_idx(p.sg, 0, gy, gx)];
      }
      T[b][r + 1][cc + 2] = v;
    }
  };
  double m = 0.;
  double s1e[NR], s1o[NR], s0e[NR], s0o[NR];  // UNIFORM: unused (the layer constants are scalars)
#pragma unroll
  for (int k = 0; k < NR; k++) { a0e[k] = a0o[k] = a2e[k] = a2o[k] = 0.; s0e[k] = s0o[k] = s1e[k] = s1o[k] = 0.; }
  stage(0, 0, a1e, a1o);
  __syncthreads();
  for (int l = 0; l < nl; l++) {
    const int b = l & 1;
    if (l < nl - 1) stage(l + 1, b ^ 1, a2e, a2o);
    const double su0 = l > 0 ? p.rc.S[l - 1] : 0., su1 = l < nl - 1 ? p.rc.S[l] : 0.;
    const double i0 = p.rc.idh0[l], i1 = p.rc.idh1[l];
    double uu = 0.;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int j = y0 + ty + BY * k;
      if (inx && j < ny) {
        const int r = ty + BY * k + 1, cx = 2 * tx + 2;  // LDS coordinates of the even cell
        const size_t cl = c0 + k * cstep + (size_t)l * p.g.ls;
        const double2 bv = *reinterpret_cast<const double2 *>(p.b + cl);
        const double be = bv.x, bo = bv.y;
        double z0e, z0o, z1e, z1o;
        if (UNIFORM) { z0e = z0o = su0; z1e = z1o = su1; }
        else {
          if (l < nl - 1) { const double2 sv = *reinterpret_cast<const double2 *>(p.S + cl); s1e[k] = sv.x; s1o[k] = sv.y; }
          z0e = s0e[k]; z0o = s0o[k]; z1e = s1e[k]; z1o = s1o[k];
        }
        const double e1 = a1e[k], o1 = a1o[k];
        const double a1w = T[b][r][cx - 1], a1ee = T[b][r][cx + 2];
        const double2 sv2 = *reinterpret_cast<const double2 *>(&T[b][r - 1][cx]), nv2 = *reinterpret_cast<const double2 *>(&T[b][r + 1][cx]);
        const double a1se = sv2.x, a1so = sv2.y, a1ne = nv2.x, a1no = nv2.y;
        double re = be, ro = bo;
        if (nl > 1) {
          if (l == 0) {
            re = be + z1e * (e1 - a2e[k]) * i1;
            ro = bo + z1o * (o1 - a2o[k]) * i1;
          } else if (l < nl - 1) {
            re = be + z0e * (e1 - a0e[k]) * i0 - z1e * (a2e[k] - e1) * i1;
            ro = bo + z0o * (o1 - a0o[k]) * i0 - z1o * (a2o[k] - o1) * i1;
          } else {
            re = be + z0e * (e1 - a0e[k]) * i0;
            ro = bo + z0o * (o1 - a0o[k]) * i0;
          }
        }
        re += DIVC(DIVC(e1 - a1w, D, rD) - DIVC(o1 - e1, D, rD), D, rD);
        re += DIVC(DIVC(e1 - a1se, D, rD) - DIVC(a1ne - e1, D, rD), D, rD);
        ro += DIVC(DIVC(o1 - e1, D, rD) - DIVC(a1ee - o1, D, rD), D, rD);
        ro += DIVC(DIVC(o1 - a1so, D, rD) - DIVC(a1no - o1, D, rD), D, rD);
        if (CORR) {
          *reinterpret_cast<double2 *>(p.a_out + cl) = make_double2(e1, o1);
          nat_write_ghosts(p.a_out, p.g, l, j, i, e1, p.walls);
          nat_write_ghosts(p.a_out, p.g, l, j, i + 1, o1, p.walls);
        }
        m = fmax(m, fmax(fabs(re), fabs(ro)));
        // face velocities of the corrected psi (comp_vel, msqg/qg.h:276-283): west and south faces of both cells
        const double nw = T[b][r + 1][cx - 1], sw = T[b][r - 1][cx - 1], se2 = T[b][r - 1][cx + 2];
        const double ue = fabs(DIVC(0.25 * (a1ne - a1se + nw - sw), D, rD)), ve = fabs(DIVC(0.25 * (o1 - a1w + a1so - sw), D, rD));
        const double uo = fabs(DIVC(0.25 * (a1no - a1so + a1ne - a1se), D, rD)), vo = fabs(DIVC(0.25 * (a1ee - e1 + se2 - a1se), D, rD));
        uu = fmax(uu, fmax(fmax(ue, ve), fmax(uo, vo)));
        a0e[k] = e1; a0o[k] = o1;
        if (!UNIFORM) { s0e[k] = s1e[k]; s0o[k] = s1o[k]; }
        a1e[k] = a2e[k]; a1o[k] = a2o[k];
      }
    }
    uu = wave_max(uu);
    if (tx == 0) smu[l][ty] = uu;
    __syncthreads();
  }
  m = wave_max(m);
  if (tx == 0) smm[ty] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = smm[0];
    for (int k = 1; k < BY; k++) mm = fmax(mm, smm[k]);
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(mm));
  }
  if (ty == 0 && tx < nl) {
    double v = smu[tx][0];
    for (int k = 1; k < BY; k++) v = fmax(v, smu[tx][k]);
    p.umax_partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * nl + tx] = v;
  }
}

// Max-only pass of the finest level as a marching kernel (round 2): max |res(a)| and max |u(a)| of an a that is already corrected
// (mode 8).  One wavefront per strip of 62 columns (+1 halo lane each side), lane = column, marching up a chunk of rows with a
// 3-row register window of every layer: x neighbours by whole-wave DPP shifts, the diagonal neighbours of the face velocities
// from the shifts of the previous row, rows prefetched two steps ahead.  No LDS tile, no barrier per layer, halo 2 lanes in 64
// and 2 rows per chunk: the pass reads psi and q once.  Uniform S only; the expressions are those of k_correct_residual, so
// both maxima are the same numbers.
#define RM_OW 62
int g_resmax_rows = 0;  // option resmax_rows: rows per chunk of k_resmax_march (0: 32), -1: the LDS-tiled kernel instead
template <int NL>
__global__ void __launch_bounds__(256) k_resmax_march(Res2Args p, int H) {
  __shared__ double smm[4];
  __shared__ double smu[MSOM_MAXNL][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int strip = blockIdx.x * 4 + wv;
  const int nx = p.g.nx, ny = p.g.ny, pitch = p.g.pitch;
  const int nstrips = (nx + RM_OW - 1) / RM_OW;
  const double D = p.rc.D, rD = 1. / D;
  double m = 0., uu[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) uu[l] = 0.;
  const int y0 = blockIdx.y * H, y1 = min(y0 + H, ny);
  if (strip < nstrips && y0 < ny) {
    const int x0 = strip * RM_OW, gi = min(x0 - 1 + lane, nx);
    const bool own = lane >= 1 && lane <= RM_OW && x0 - 1 + lane < nx;
    const size_t ls = p.g.ls;
    const double *pa = p.a + nat_idx(p.g, 0, 0, gi), *pb = p.b + nat_idx(p.g, 0, 0, min(gi, nx - 1) < 0 ? 0 : min(gi, nx - 1));
    double Wm[NL], Wc[NL], Wp[NL], Wn[NL], Em[NL], Mm[NL], qc[NL], qn[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      Wm[l] = pa[l * ls + (ptrdiff_t)(y0 - 1) * pitch];
      Wc[l] = pa[l * ls + (ptrdiff_t)y0 * pitch];
      Wp[l] = pa[l * ls + (ptrdiff_t)min(y0 + 1, ny) * pitch];
      qc[l] = pb[l * ls + (ptrdiff_t)y0 * pitch];
    }
#pragma unroll
    for (int l = 0; l < NL; l++) { Em[l] = lane_above(Wm[l]); Mm[l] = lane_below(Wm[l]); }
    for (int j = y0; j < y1; j++) {
      const ptrdiff_t rn = (ptrdiff_t)min(j + 2, ny) * pitch, rq = (ptrdiff_t)min(j + 1, ny - 1) * pitch;
#pragma unroll
      for (int l = 0; l < NL; l++) { Wn[l] = pa[l * ls + rn]; qn[l] = pb[l * ls + rq]; }
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double c = Wc[l], E = lane_above(c), W = lane_below(c), N = Wp[l], S = Wm[l], NW = lane_below(N), SE = Em[l], SW = Mm[l];
        const double be = qc[l];
        double re = be;
        if (NL > 1) {
          const double i0 = p.rc.idh0[l], i1 = p.rc.idh1[l];
          const double z0 = l > 0 ? p.rc.S[l - 1] : 0., z1 = l < NL - 1 ? p.rc.S[l] : 0.;
          if (l == 0) re = be + z1 * (c - Wc[l + 1 < NL ? l + 1 : l]) * i1;
          else if (l < NL - 1) re = be + z0 * (c - Wc[l - 1]) * i0 - z1 * (Wc[l + 1 < NL ? l + 1 : l] - c) * i1;
          else re = be + z0 * (c - Wc[l - 1]) * i0;
        }
        re += DIVC(DIVC(c - W, D, rD) - DIVC(E - c, D, rD), D, rD);
        re += DIVC(DIVC(c - S, D, rD) - DIVC(N - c, D, rD), D, rD);
        const double u = fabs(DIVC(0.25 * (N - S + NW - SW), D, rD)), v = fabs(DIVC(0.25 * (E - W + SE - SW), D, rD));
        if (own) { m = fmax(m, fabs(re)); uu[l] = fmax(uu[l], fmax(u, v)); }
        Em[l] = E; Mm[l] = W;
      }
#pragma unroll
      for (int l = 0; l < NL; l++) { Wm[l] = Wc[l]; Wc[l] = Wp[l]; Wp[l] = Wn[l]; qc[l] = qn[l]; }
    }
  }
  m = wave_max(m);
  if (lane == 0) smm[wv] = m;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const double w = wave_max(uu[l]);
    if (lane == 0) smu[l][wv] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double mm = fmax(fmax(smm[0], smm[1]), fmax(smm[2], smm[3]));
    atomicMax((unsigned long long *)p.maxres, (unsigned long long)__double_as_longlong(mm));
  }
  if (threadIdx.x < NL)
    p.umax_partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NL + threadIdx.x] =
        fmax(fmax(smu[threadIdx.x][0], smu[threadIdx.x][1]), fmax(smu[threadIdx.x][2], smu[threadIdx.x][3]));
}
template <int NL>
static dim3 resmax_march_launch(hipStream_t st, const Res2Args &p, int H) {
  const int nstrips = (p.g.nx + RM_OW - 1) / RM_OW;
  dim3 gr((nstrips + 3) / 4, (p.g.ny + H - 1) / H);
  hipLaunchKernelGGL(k_resmax_march<NL>, gr, dim3(256), 0, st, p, H);
  return gr;
}
static dim3 launch_resmax_march(hipStream_t st, const Res2Args &p, int H) {
  switch (p.nl) {
    case 1: return resmax_march_launch<1>(st, p, H);
    case 2: return resmax_march_launch<2>(st, p, H);
    case 3: return resmax_march_launch<3>(st, p, H);
    case 4: return resmax_march_launch<4>(st, p, H);
    case 5: return resmax_march_launch<5>(st, p, H);
    case 6: return resmax_march_launch<6>(st, p, H);
    case 7: return resmax_march_launch<7>(st, p, H);
    default: return resmax_march_launch<8>(st, p, H);
  }
}

// out[l] = max_b partial[b][l]
__global__ void k_max_final_mg(const double *partial, double *out, int nb, int nl) {
  // MAXF_BLOCKS blocks, each over a contiguous chunk of partial rows; thread = row, the nl values of a
  // row are contiguous.  max of non-negative doubles is order-independent: atomicMax on the bit pattern
  // (out[] zeroed by the launcher).
  __shared__ double sm[MSOM_MAXNL][4];
  const int per = (nb + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = min(nb, b0 + per);
  double v[MSOM_MAXNL];
#pragma unroll
  for (int l = 0; l < MSOM_MAXNL; l++) v[l] = 0.;
  for (int b = b0 + threadIdx.x; b < b1; b += 256) {
    const double *row = partial + (size_t)b * nl;
#pragma unroll
    for (int l = 0; l < MSOM_MAXNL; l++)
      if (l < nl) v[l] = fmax(v[l], row[l]);
  }
#pragma unroll
  for (int l = 0; l < MSOM_MAXNL; l++) {
    double w = v[l];
    for (int o = 32; o > 0; o >>= 1) w = fmax(w, __shfl_down(w, o, 64));
    if ((threadIdx.x & 63) == 0) sm[l][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if (threadIdx.x < nl) {
    const double w = fmax(fmax(sm[threadIdx.x][0], sm[threadIdx.x][1]), fmax(sm[threadIdx.x][2], sm[threadIdx.x][3]));
    atomicMax((unsigned long long *)(out + threadIdx.x), (unsigned long long)__double_as_longlong(w));
  }
}

int residual2_blocks(const NatGeom &g) {
  dim3 gr = grid2d(g.nx / 2, g.ny);
  return gr.x * gr.y;
}
// mode bits: 1 = CORRECT, 2 = WRITE, 4 = RESTRICT
void launch_residual2(hipStream_t st, int mode, const double *a, const double *da, double *a_out, const double *b, const double *S,
                      const NatGeom &g, double *res, const SplitGeom &sg, double *res_c, const SplitGeom &cg, int nl, const RelaxCoef &rc,
                      int uniformS, int walls, double *maxres, double *sum_partial, int want_sum, double *umax_partial, double *umax_out, int umax_clean,
                      double *res_c2, const SplitGeom *cg2) {
  Res2Args p;
  p.res_c2 = res_c2;
  if (cg2) p.cg2 = *cg2; else p.cg2 = cg;
  extern int g_rhs_dbg;
  p.dbg = g_rhs_dbg;
  p.umax_partial = umax_partial;
  p.a = a; p.b = b; p.S = S; p.da = da; p.a_out = a_out; p.res = res; p.res_c = res_c; p.maxres = maxres; p.sum_partial = sum_partial;
  p.g = g; p.sg = sg; p.cg = cg; p.nl = nl; p.uniformS = uniformS; p.want_sum = want_sum; p.walls = walls; p.rc = rc;
  dim3 gr = grid2d(g.nx / 2, g.ny);
  switch (mode) {
    case 1:
      if (g.nx % 2 == 0 && g.nx >= CR_TW && g.ny >= CR_TR && !(p.dbg & 128)) {
        gr = dim3((g.nx + CR_TW - 1) / CR_TW, (g.ny + CR_TR - 1) / CR_TR);
        if (uniformS) hipLaunchKernelGGL((k_correct_residual<true, true>), gr, block2d(), 0, st, p);
        else hipLaunchKernelGGL((k_correct_residual<false, true>), gr, block2d(), 0, st, p);
      } else
        hipLaunchKernelGGL((k_residual2<true, false, false>), gr, block2d(), 0, st, p);
      if (!umax_clean) (void)hipMemsetAsync(umax_out, 0, nl * sizeof(double), st);  // else zeroed with the solve's other accumulators
      hipLaunchKernelGGL(k_max_final_mg, dim3(64), dim3(256), 0, st, umax_partial, umax_out, (int)(gr.x * gr.y), nl);
      break;
    case 2: hipLaunchKernelGGL((k_residual2<false, true, false>), gr, block2d(), 0, st, p); break;
    case 6: hipLaunchKernelGGL((k_residual2<false, true, true>), gr, block2d(), (size_t)nl * (BY * BX * 2 + 2 * BX) * sizeof(double), st, p); break;
    case 0: hipLaunchKernelGGL((k_residual2<false, false, false>), gr, block2d(), 0, st, p); break;
    case 8:  // max |res(a)| and max |u(a)| of an a that is already corrected
      if (uniformS && g.nx >= 64 && g.ny >= 16 && g_resmax_rows >= 0 && nl <= MSOM_FASTNL) {
        gr = launch_resmax_march(st, p, g_resmax_rows ? g_resmax_rows : 32);
      } else if (g.nx % 2 == 0 && g.nx >= CR_TW && g.ny >= CR_TR && !(p.dbg & 128)) {
        gr = dim3((g.nx + CR_TW - 1) / CR_TW, (g.ny + CR_TR - 1) / CR_TR);
        if (uniformS) hipLaunchKernelGGL((k_correct_residual<true, false>), gr, block2d(), 0, st, p);
        else hipLaunchKernelGGL((k_correct_residual<false, false>), gr, block2d(), 0, st, p);
      } else
        hipLaunchKernelGGL((k_residual2<false, false, false, true>), gr, block2d(), 0, st, p);
      if (!umax_clean) (void)hipMemsetAsync(umax_out, 0, nl * sizeof(double), st);  // else zeroed with the solve's other accumulators
      hipLaunchKernelGGL(k_max_final_mg, dim3(64), dim3(256), 0, st, umax_partial, umax_out, (int)(gr.x * gr.y), nl);
      break;
    default: break;
  }
}

// ------------------------------------------------------------------ K10 restriction, K11 prolongation

// coarse = mean of the 4 children, summed in Basilisk's foreach_child order
__device__ __forceinline__ void restrict_pt(const double *__restrict__ fine, const SplitGeom &fg, double *coarse, const SplitGeom &cg, int nl, int I, int J) {
  for (int l = 0; l < nl; l++) {
    double sum = 0.;
    sum += fine[split_idx(fg, l, 2 * J, 2 * I)];
    sum += fine[split_idx(fg, l, 2 * J + 1, 2 * I)];
    sum += fine[split_idx(fg, l, 2 * J, 2 * I + 1)];
    sum += fine[split_idx(fg, l, 2 * J + 1, 2 * I + 1)];
    coarse[split_idx(cg, l, J, I)] = sum / 4;
  }
}
__global__ void k_restrict(const double *__restrict__ fine, SplitGeom fg, double *coarse, SplitGeom cg, int nl) {
  const int I = blockIdx.x * BX + threadIdx.x, J = blockIdx.y * BY + threadIdx.y;
  if (I >= cg.nx || J >= cg.ny) return;
  restrict_pt(fine, fg, coarse, cg, nl, I, J);
}
void launch_restrict(hipStream_t st, const double *fine, const SplitGeom &fg, double *coarse, const SplitGeom &cg, int nl) {
  hipLaunchKernelGGL(k_restrict, grid2d(cg.nx, cg.ny), block2d(), 0, st, fine, fg, coarse, cg, nl);
}

// Several restrictions in one launch (round 3): the chain below the level the residual pass restricts to is 4-5 launches of ~5 us.
// A workgroup takes a tile of 2^n x 2^n cells of the finest level of the chain and produces its 2^(n-1) x 2^(n-1), ..., 1 x 1 means
// on the n coarser levels, level by level through LDS (the mean of four children needs nothing outside the tile); sums in restrict_pt's
// order from the values just stored, so every level gets the bits of the launch-per-level chain.
#define RPYR_MAX 5
struct RestrictPyramidArgs {
  const double *fine;
  double *out[RPYR_MAX];
  SplitGeom g[RPYR_MAX + 1];   // g[0]: the fine level, g[k]: output level k
  int n, nl;
};
__global__ void __launch_bounds__(256) k_restrict_pyramid(RestrictPyramidArgs a) {
  extern __shared__ double rp[];   // two buffers of nl x (T/2)^2 and nl x (T/4)^2 doubles
  const int tid = threadIdx.x, n = a.n, nl = a.nl, T = 1 << n;
  const int X0 = blockIdx.x * T, Y0 = blockIdx.y * T;
  double *cur = rp, *nxt = rp + nl * (T / 2) * (T / 2);
  {
    const int s = T >> 1;
    for (int t = tid; t < s * s * nl; t += 256) {
      const int I = t % s, J = (t / s) % s, l = t / (s * s);
      const int fi = X0 + 2 * I, fj = Y0 + 2 * J;
      double sum = 0.;
      sum += a.fine[split_idx(a.g[0], l, fj, fi)];
      sum += a.fine[split_idx(a.g[0], l, fj + 1, fi)];
      sum += a.fine[split_idx(a.g[0], l, fj, fi + 1)];
      sum += a.fine[split_idx(a.g[0], l, fj + 1, fi + 1)];
      const double v = sum / 4;
      cur[(l * s + J) * s + I] = v;
      a.out[0][split_idx(a.g[1], l, (Y0 >> 1) + J, (X0 >> 1) + I)] = v;
    }
  }
  for (int k = 2; k <= n; k++) {
    __syncthreads();
    const int s = T >> k, sf = s << 1;
    for (int t = tid; t < s * s * nl; t += 256) {
      const int I = t % s, J = (t / s) % s, l = t / (s * s);
      const double *f = cur + l * sf * sf;
      double sum = 0.;
      sum += f[(2 * J) * sf + 2 * I];
      sum += f[(2 * J + 1) * sf + 2 * I];
      sum += f[(2 * J) * sf + 2 * I + 1];
      sum += f[(2 * J + 1) * sf + 2 * I + 1];
      const double v = sum / 4;
      nxt[(l * s + J) * s + I] = v;
      a.out[k - 1][split_idx(a.g[k], l, (Y0 >> k) + J, (X0 >> k) + I)] = v;
    }
    double *tmp = cur; cur = nxt; nxt = tmp;
  }
}
// levels[0] -> levels[1 .. n] (n <= RPYR_MAX); every level must halve exactly (cell-centred grids do)
void launch_restrict_pyramid(hipStream_t st, const double *fine, double *const *out, const SplitGeom *g, int n, int nl) {
  RestrictPyramidArgs a;
  a.fine = fine; a.n = n; a.nl = nl;
  for (int k = 0; k <= n; k++) a.g[k] = g[k];
  for (int k = 0; k < n; k++) a.out[k] = out[k];
  const int T = 1 << n;
  const size_t lds = (size_t)nl * ((T / 2) * (T / 2) + (T / 4 > 0 ? (T / 4) * (T / 4) : 1)) * sizeof(double);
  hipLaunchKernelGGL(k_restrict_pyramid, dim3(g[0].nx / T, g[0].ny / T), dim3(256), lds, st, a);
}

// bilinear: (9 c + 3 (c[child.x] + c[0,child.y]) + c[child.x,child.y]) / 16, then
// boundary_level(da) on the fine level (ghosts written by the wall threads)
__device__ __forceinline__ void prolong_pt(const double *__restrict__ coarse, const SplitGeom &cg, double *fine, const SplitGeom &fg, int nl, int walls, int i, int j) {
  const int I = i >> 1, J = j >> 1, cx = (i & 1) ? 1 : -1, cy = (j & 1) ? 1 : -1;
  for (int l = 0; l < nl; l++) {
    const double v = BILINEAR(coarse[split_idx(cg, l, J, I)], coarse[split_idx(cg, l, J, I + cx)], coarse[split_idx(cg, l, J + cy, I)],
                              coarse[split_idx(cg, l, J + cy, I + cx)]);
    fine[split_idx(fg, l, j, i)] = v;
    split_write_ghosts(fine, fg, l, j, i, v, walls);
  }
}
__global__ void k_prolong(const double *__restrict__ coarse, SplitGeom cg, double *fine, SplitGeom fg, int nl, int walls) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (i >= fg.nx || j >= fg.ny) return;
  prolong_pt(coarse, cg, fine, fg, nl, walls, i, j);
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
  int region;  // 0: all cells of the colour, 1: all but the outermost ring of the tile (the ring goes first, k_relax_ring,
               // so that its halo exchange overlaps this launch)
  RelaxCoef rc;
};

// FINE tags the instantiation launched on the finest level (level 0): a distinct kernel
// symbol, so that profiler statistics of the HBM-bound fine sweep are not averaged with the
// launch-latency-bound coarse levels.
template <int NL, bool UNIFORM>
__device__ __forceinline__ void relax_color_pt(const RelaxArgs &p, int kx, int j);
template <int NL, bool UNIFORM, bool FINE>
__global__ void __launch_bounds__(BX *BY) k_relax_color(RelaxArgs p) {
  const int kx = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  if (kx >= p.g.hk || j >= p.g.ny) return;
  if (p.region == 1) {
    const int i = 2 * kx + ((j + p.color) & 1);
    if (i == 0 || i == p.g.nx - 1 || j == 0 || j == p.g.ny - 1) return;
  }
  relax_color_pt<NL, UNIFORM>(p, kx, j);
}
// the outermost ring of the tile, colour p.color: rows 0 and ny-1 (hk cells each) and, on every other row,
// the one end cell that has this colour (i = 0 has x parity 0, i = nx-1 parity 1)
template <int NL, bool UNIFORM>
__global__ void __launch_bounds__(256) k_relax_ring(RelaxArgs p) {
  const int t = blockIdx.x * 256 + threadIdx.x, hk = p.g.hk, ny = p.g.ny;
  int kx, j;
  if (t < hk) { kx = t; j = 0; }
  else if (t < 2 * hk) { kx = t - hk; j = ny - 1; }
  else if (t < 2 * hk + ny - 2) { j = 1 + t - 2 * hk; kx = ((j + p.color) & 1) ? hk - 1 : 0; }
  else return;
  if (ny == 1 && t >= hk) return;
  relax_color_pt<NL, UNIFORM>(p, kx, j);
}
template <int NL, bool UNIFORM>
__device__ __forceinline__ void relax_color_pt(const RelaxArgs &p, int kx, int j) {
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

// Two adjacent same-colour points per thread: every access is a 16-byte load/store
// (4 x double2 + 1 scalar load and 1 double2 store per layer for two columns), which is what
// the HBM path of gfx950 wants.  Used on levels whose half rows are even and >= 128 wide.
template <int NL, bool UNIFORM, bool FINE>
__global__ void __launch_bounds__(BX *BY) k_relax_color_x2(RelaxArgs p) {
  const int kx = 2 * (blockIdx.x * BX + threadIdx.x), j = blockIdx.y * BY + threadIdx.y;
  if (kx >= p.g.hk || j >= p.g.ny) return;
  if (p.region == 1 && (j == 0 || j == p.g.ny - 1)) return;
  const int px = (j + p.color) & 1;
  const int i = 2 * kx + px;  // x of the first point; the second is i + 2
  const int hp = p.g.hp, rp = p.g.rp;
  const size_t ls = p.g.ls;
  const size_t own = (size_t)(j + 1) * rp + px * hp + MSOM_SP + kx;
  const size_t oth = (size_t)(j + 1) * rp + (1 - px) * hp + MSOM_SP + kx;
  const size_t ixs = px ? oth + 2 : oth - 1;
  const double sqD = p.rc.sqD;
  double xa[NL], xb[NL];
  auto ld2 = [&](const double *f, size_t k) { return *reinterpret_cast<const double2 *>(f + k); };
  if (NL == 1) {
    const double2 b = ld2(p.res, own), d = ld2(p.da, oth), n = ld2(p.da, own + rp), s = ld2(p.da, own - rp);
    const double xs = p.da[ixs];
    const double w1 = px ? d.x : xs, e1 = px ? d.y : d.x, w2 = px ? d.y : d.x, e2 = px ? xs : d.y;
    double n1 = -sqD * b.x, n2 = -sqD * b.y, dd = 0.;
    n1 += e1 + w1; n2 += e2 + w2; dd += 2.;
    n1 += n.x + s.x; n2 += n.y + s.y; dd += 2.;
    xa[0] = n1 / dd; xb[0] = n2 / dd;
  } else {
    double ra[NL], rb[NL], t0a[NL], t1a[NL], t2a[NL], t0b[NL], t1b[NL], t2b[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double2 b = ld2(p.res, own + l * ls), d = ld2(p.da, oth + l * ls), n = ld2(p.da, own + rp + l * ls), s = ld2(p.da, own - rp + l * ls);
      const double xs = p.da[ixs + l * ls];
      const double w1 = px ? d.x : xs, e1 = px ? d.y : d.x, w2 = px ? d.y : d.x, e2 = px ? xs : d.y;
      ra[l] = -sqD * b.x; rb[l] = -sqD * b.y;
      if (!UNIFORM) {
        double2 sm = make_double2(0., 0.), sc = make_double2(0., 0.);
        if (l > 0) sm = ld2(p.S, own + (l - 1) * ls);
        if (l < NL - 1) sc = ld2(p.S, own + l * ls);
        t0a[l] = l > 0 ? -sqD * sm.x * p.rc.idh0[l] : 0.; t0b[l] = l > 0 ? -sqD * sm.y * p.rc.idh0[l] : 0.;
        t2a[l] = l < NL - 1 ? -sqD * sc.x * p.rc.idh1[l] : 0.; t2b[l] = l < NL - 1 ? -sqD * sc.y * p.rc.idh1[l] : 0.;
        t1a[l] = l == 0 ? -t2a[l] : (l < NL - 1 ? -t0a[l] - t2a[l] : -t0a[l]);
        t1b[l] = l == 0 ? -t2b[l] : (l < NL - 1 ? -t0b[l] - t2b[l] : -t0b[l]);
        ra[l] += 1. * e1 + 1. * w1; rb[l] += 1. * e2 + 1. * w2;
        t1a[l] += 1. + 1.; t1b[l] += 1. + 1.;
        ra[l] += 1. * n.x + 1. * s.x; rb[l] += 1. * n.y + 1. * s.y;
        t1a[l] += 1. + 1.; t1b[l] += 1. + 1.;
      } else {
        ra[l] += e1 + w1; rb[l] += e2 + w2;
        ra[l] += n.x + s.x; rb[l] += n.y + s.y;
      }
    }
    if (UNIFORM) {
#pragma unroll
      for (int l = 1; l < NL; l++) { ra[l] -= p.rc.w[l] * ra[l - 1]; rb[l] -= p.rc.w[l] * rb[l - 1]; }
      xa[NL - 1] = ra[NL - 1] * p.rc.it1[NL - 1]; xb[NL - 1] = rb[NL - 1] * p.rc.it1[NL - 1];
#pragma unroll
      for (int l = NL - 2; l >= 0; l--) {
        xa[l] = (ra[l] - p.rc.t2[l] * xa[l + 1]) * p.rc.it1[l];
        xb[l] = (rb[l] - p.rc.t2[l] * xb[l + 1]) * p.rc.it1[l];
      }
    } else {
#pragma unroll
      for (int l = 1; l < NL; l++) {
        t1a[l] -= t0a[l] * t2a[l - 1] / t1a[l - 1]; ra[l] -= t0a[l] * ra[l - 1] / t1a[l - 1];
        t1b[l] -= t0b[l] * t2b[l - 1] / t1b[l - 1]; rb[l] -= t0b[l] * rb[l - 1] / t1b[l - 1];
      }
      xa[NL - 1] = ra[NL - 1] / t1a[NL - 1]; xb[NL - 1] = rb[NL - 1] / t1b[NL - 1];
#pragma unroll
      for (int l = NL - 2; l >= 0; l--) {
        xa[l] = (ra[l] - t2a[l] * xa[l + 1]) / t1a[l];
        xb[l] = (rb[l] - t2b[l] * xb[l + 1]) / t1b[l];
      }
    }
  }
  if (p.region == 1 && (i == 0 || i + 2 == p.g.nx - 1)) {  // one cell of the pair is on the ring: store the other one only
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (i == 0) p.da[own + 1 + l * ls] = xb[l];
      else p.da[own + l * ls] = xa[l];
    }
    return;
  }
#pragma unroll
  for (int l = 0; l < NL; l++) *reinterpret_cast<double2 *>(p.da + own + l * ls) = make_double2(xa[l], xb[l]);
  const bool edge = (i == 0) | (i + 2 >= p.g.nx - 1) | (j == 0) | (j == p.g.ny - 1);
  if (edge && p.walls && p.region == 0) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      split_write_ghosts(p.da, p.g, l, j, i, xa[l], p.walls);
      split_write_ghosts(p.da, p.g, l, j, i + 2, xb[l], p.walls);
    }
  }
}

// First red half-sweep of a level fused with the prolongation: instead of reading the black
// neighbours from da (which would first have to be written by a prolongation pass), each of
// them is interpolated on the fly from the 3x3 coarse cells around the red cell (bilinear
// rule of k_prolong).  In red-black order the interpolated values are needed nowhere else:
// red values are overwritten without being read, black values are only read here -- except
// as lagged wall ghosts, which this kernel writes (ghost = -interpolated wall cell).
// All lanes of a wave share the parities of i and j, so the case analysis is wave-uniform.
struct RelaxPArgs {
  double *da;
  const double *res, *S, *coarse;
  SplitGeom g, cg;
  int walls;
  RelaxCoef rc;
};
template <int NL, bool UNIFORM, int PJ>
__device__ __forceinline__ void relax_red_prolong_body(const RelaxPArgs &p) {
  // rows of one parity per workgroup half (blockIdx.z): the parities are compile-time, so the
  // choice of coarse cells is a static register selection
  const int kx = blockIdx.x * BX + threadIdx.x, j = 2 * (blockIdx.y * BY + threadIdx.y) + PJ;
  if (kx >= p.g.hk || j >= p.g.ny) return;
  constexpr int pj = PJ, pi = PJ;  // red: (i + j) even
  const int i = 2 * kx + pi, I0 = kx, J0 = j >> 1;
  const int nx = p.g.nx, ny = p.g.ny;
  const size_t ls = p.g.ls;
  const size_t own = split_idx(p.g, 0, j, i);
  // coarse window offsets [dj+1][di+1] (layer 0)
  size_t cw[3][3];
#pragma unroll
  for (int dj = -1; dj <= 1; dj++)
#pragma unroll
    for (int di = -1; di <= 1; di++) cw[dj + 1][di + 1] = split_idx(p.cg, 0, J0 + dj, I0 + di);
  // x columns (near, far) and y rows (near, far) of the four neighbours, window coordinates
  constexpr int wxn = pi ? 1 : 0, wxf = pi ? 0 : 1;   // W = (i-1, j): columns {I0-1, I0}
  constexpr int exn = pi ? 2 : 1, exf = pi ? 1 : 2;   // E = (i+1, j): columns {I0, I0+1}
  constexpr int cxf = pi ? 2 : 0;                     // S, N: x near = I0, far = I0 + (pi ? 1 : -1)
  constexpr int cyf = pj ? 2 : 0;                     // W, E: y near = J0, far = J0 + (pj ? 1 : -1)
  constexpr int syn = pj ? 1 : 0, syf = pj ? 0 : 1;   // S = (i, j-1): rows {J0-1, J0}
  constexpr int nyn = pj ? 2 : 1, nyf = pj ? 1 : 2;   // N = (i, j+1): rows {J0, J0+1}
  const double sqD = p.rc.sqD;
  double rhs[NL], x[NL], t0[NL], t1[NL], t2[NL];
  const bool wl = i - 1 == 0 && (p.walls & WALL_W), el = i + 1 == nx - 1 && (p.walls & WALL_E);
  const bool sl = j - 1 == 0 && (p.walls & WALL_S), nl_ = j + 1 == ny - 1 && (p.walls & WALL_N);
  const bool ow = i == 0 && (p.walls & WALL_W), oe = i == nx - 1 && (p.walls & WALL_E);
  const bool os = j == 0 && (p.walls & WALL_S), on = j == ny - 1 && (p.walls & WALL_N);
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const double *cc = p.coarse + (size_t)l * p.cg.ls;
    double c[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) c[a][b] = cc[cw[a][b]];
#define BIL(yn, yf, xn, xf) BILINEAR(c[yn][xn], c[yn][xf], c[yf][xn], c[yf][xf])
    // interpolated neighbours; beyond a wall the ghost is -own interpolated value (lagged ghost)
    const double vo = (ow | oe | os | on) ? BIL(1, cyf, 1, cxf) : 0.;
    const double vw = ow ? -vo : BIL(1, cyf, wxn, wxf);
    const double ve = oe ? -vo : BIL(1, cyf, exn, exf);
    const double vs = os ? -vo : BIL(syn, syf, 1, cxf);
    const double vn = on ? -vo : BIL(nyn, nyf, 1, cxf);
#undef BIL
    // black wall cells whose wall-normal interior neighbour is this red cell: write their lagged
    // ghosts for the black half-sweep (exactly one writer per ghost)
    if (wl) p.da[split_idx(p.g, l, j, -1)] = -vw;
    if (el) p.da[split_idx(p.g, l, j, nx)] = -ve;
    if (sl) p.da[split_idx(p.g, l, -1, i)] = -vs;
    if (nl_) p.da[split_idx(p.g, l, ny, i)] = -vn;
    rhs[l] = -sqD * p.res[own + l * ls];
    if (NL > 1 && !UNIFORM) {
      t0[l] = l > 0 ? -sqD * p.S[own + (l - 1) * ls] * p.rc.idh0[l] : 0.;
      t2[l] = l < NL - 1 ? -sqD * p.S[own + l * ls] * p.rc.idh1[l] : 0.;
      t1[l] = l == 0 ? -t2[l] : (l < NL - 1 ? -t0[l] - t2[l] : -t0[l]);
      rhs[l] += 1. * ve + 1. * vw;
      t1[l] += 1. + 1.;
      rhs[l] += 1. * vn + 1. * vs;
      t1[l] += 1. + 1.;
    } else {
      rhs[l] += ve + vw;
      rhs[l] += vn + vs;
    }
  }
  if (NL == 1) {
    double d = 0.;
    d += 2.; d += 2.;
    x[0] = rhs[0] / d;
  } else if (UNIFORM) {
#pragma unroll
    for (int l = 1; l < NL; l++) rhs[l] -= p.rc.w[l] * rhs[l - 1];
    x[NL - 1] = rhs[NL - 1] * p.rc.it1[NL - 1];
#pragma unroll
    for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - p.rc.t2[l] * x[l + 1]) * p.rc.it1[l];
  } else {
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
  if ((ow | oe | os | on) || ((p.walls & WALL_PER) && (i == 0 || i == nx - 1 || j == 0 || j == ny - 1))) {
#pragma unroll
    for (int l = 0; l < NL; l++) split_write_ghosts(p.da, p.g, l, j, i, x[l], p.walls);
  }
}
// Both row parities in one thread: the red cells (2 kx, 2 J) and (2 kx + 1, 2 J + 1) interpolate their
// neighbours from the SAME 3 x 3 coarse window around (kx, J), so the window is loaded once for two
// cells (9 instead of 18 coarse loads per pair and layer).  Per-cell arithmetic as in
// relax_red_prolong_body, hence identical results.
template <int NL, bool UNIFORM, int PJ>
struct RedCell {
  int i, j;
  size_t own;
  bool valid, wl, el, sl, nl_, ow, oe, os, on;
  double rhs[NL], t0[NL], t1[NL], t2[NL];
  __device__ __forceinline__ void init(const RelaxPArgs &p, int kx, int J) {
    i = 2 * kx + PJ; j = 2 * J + PJ;
    valid = kx < p.g.hk && j < p.g.ny;
    const int nx = p.g.nx, ny = p.g.ny;
    own = split_idx(p.g, 0, valid ? j : 0, valid ? i : 0);
    wl = i - 1 == 0 && (p.walls & WALL_W); el = i + 1 == nx - 1 && (p.walls & WALL_E);
    sl = j - 1 == 0 && (p.walls & WALL_S); nl_ = j + 1 == ny - 1 && (p.walls & WALL_N);
    ow = i == 0 && (p.walls & WALL_W); oe = i == nx - 1 && (p.walls & WALL_E);
    os = j == 0 && (p.walls & WALL_S); on = j == ny - 1 && (p.walls & WALL_N);
  }
  __device__ __forceinline__ void layer(const RelaxPArgs &p, int l, const double (&c)[3][3]) {
    if (!valid) return;
    constexpr int pj = PJ, pi = PJ;
    constexpr int wxn = pi ? 1 : 0, wxf = pi ? 0 : 1, exn = pi ? 2 : 1, exf = pi ? 1 : 2, cxf = pi ? 2 : 0, cyf = pj ? 2 : 0;
    constexpr int syn = pj ? 1 : 0, syf = pj ? 0 : 1, nyn = pj ? 2 : 1, nyf = pj ? 1 : 2;
    const size_t ls = p.g.ls;
    const double sqD = p.rc.sqD;
#define BIL(yn, yf, xn, xf) BILINEAR(c[yn][xn], c[yn][xf], c[yf][xn], c[yf][xf])
    const double vo = (ow | oe | os | on) ? BIL(1, cyf, 1, cxf) : 0.;
    const double vw = ow ? -vo : BIL(1, cyf, wxn, wxf);
    const double ve = oe ? -vo : BIL(1, cyf, exn, exf);
    const double vs = os ? -vo : BIL(syn, syf, 1, cxf);
    const double vn = on ? -vo : BIL(nyn, nyf, 1, cxf);
#undef BIL
    if (wl) p.da[split_idx(p.g, l, j, -1)] = -vw;
    if (el) p.da[split_idx(p.g, l, j, p.g.nx)] = -ve;
    if (sl) p.da[split_idx(p.g, l, -1, i)] = -vs;
    if (nl_) p.da[split_idx(p.g, l, p.g.ny, i)] = -vn;
    rhs[l] = -sqD * p.res[own + l * ls];
    if (NL > 1 && !UNIFORM) {
      t0[l] = l > 0 ? -sqD * p.S[own + (l - 1) * ls] * p.rc.idh0[l] : 0.;
      t2[l] = l < NL - 1 ? -sqD * p.S[own + l * ls] * p.rc.idh1[l] : 0.;
      t1[l] = l == 0 ? -t2[l] : (l < NL - 1 ? -t0[l] - t2[l] : -t0[l]);
      rhs[l] += 1. * ve + 1. * vw;
      t1[l] += 1. + 1.;
      rhs[l] += 1. * vn + 1. * vs;
      t1[l] += 1. + 1.;
    } else {
      rhs[l] += ve + vw;
      rhs[l] += vn + vs;
    }
  }
  __device__ __forceinline__ void finish(const RelaxPArgs &p) {
    if (!valid) return;
    const size_t ls = p.g.ls;
    double x[NL];
    if (NL == 1) {
      double d = 0.;
      d += 2.; d += 2.;
      x[0] = rhs[0] / d;
    } else if (UNIFORM) {
#pragma unroll
      for (int l = 1; l < NL; l++) rhs[l] -= p.rc.w[l] * rhs[l - 1];
      x[NL - 1] = rhs[NL - 1] * p.rc.it1[NL - 1];
#pragma unroll
      for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - p.rc.t2[l] * x[l + 1]) * p.rc.it1[l];
    } else {
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
    if ((ow | oe | os | on) || ((p.walls & WALL_PER) && (i == 0 || i == p.g.nx - 1 || j == 0 || j == p.g.ny - 1))) {
#pragma unroll
      for (int l = 0; l < NL; l++) split_write_ghosts(p.da, p.g, l, j, i, x[l], p.walls);
    }
  }
};
template <int NL, bool UNIFORM>
__device__ __forceinline__ void red_prolong2_pt(const RelaxPArgs &p, int kx, int J);
template <int NL, bool UNIFORM>
__global__ void __launch_bounds__(BX *BY) k_relax_red_prolong2(RelaxPArgs p) {
  const int kx = blockIdx.x * BX + threadIdx.x, J = blockIdx.y * BY + threadIdx.y;
  if (kx >= p.g.hk || 2 * J >= p.g.ny) return;
  red_prolong2_pt<NL, UNIFORM>(p, kx, J);
}
template <int NL, bool UNIFORM>
__device__ __forceinline__ void red_prolong2_pt(const RelaxPArgs &p, int kx, int J) {
  size_t cw[3][3];
#pragma unroll
  for (int dj = -1; dj <= 1; dj++)
#pragma unroll
    for (int di = -1; di <= 1; di++) cw[dj + 1][di + 1] = split_idx(p.cg, 0, J + dj, kx + di);
  // one cell after the other (the second pass re-reads the window, L1 hits): fewer live registers, one more wave per SIMD
  {
    RedCell<NL, UNIFORM, 0> c0;
    c0.init(p, kx, J);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double *cc = p.coarse + (size_t)l * p.cg.ls;
      double c[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) c[a][b] = cc[cw[a][b]];
      c0.layer(p, l, c);
    }
    c0.finish(p);
  }
  {
    RedCell<NL, UNIFORM, 1> c1;
    c1.init(p, kx, J);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double *cc = p.coarse + (size_t)l * p.cg.ls;
      double c[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) c[a][b] = cc[cw[a][b]];
      c1.layer(p, l, c);
    }
    c1.finish(p);
  }
}

// Same cells per thread as k_relax_red_prolong2, with the coarse correction staged through LDS: a 64 x 4
// thread block needs the (64 + 2) x (4 + 2) coarse cells around it on every layer; loaded once (~9 loads
// per thread for all layers) instead of 9 per thread and layer, the 3 x 3 windows are then LDS reads.
// Used on wide levels, where the per-layer window loads made the pass instruction-bound.
template <int NL, bool UNIFORM>
__global__ void __launch_bounds__(BX *BY) k_relax_red_prolong3(RelaxPArgs p) {
  constexpr int CW = BX + 2, CH = BY + 2, CP = CW + 2;
  __shared__ double C[NL][CH][CP];
  const int tid = threadIdx.y * BX + threadIdx.x;
  const int kx0 = blockIdx.x * BX, J0 = blockIdx.y * BY;
  for (int t = tid; t < NL * CH * CW; t += BX * BY) {
    const int l = t / (CH * CW), r = (t / CW) % CH, c = t % CW;
    const int I = kx0 - 1 + c, J = J0 - 1 + r;
    C[l][r][c] = (I <= p.cg.nx && J <= p.cg.ny) ? p.coarse[split_idx(p.cg, l, J, I)] : 0.;
  }
  __syncthreads();
  const int kx = kx0 + threadIdx.x, J = J0 + threadIdx.y;
  if (kx >= p.g.hk || 2 * J >= p.g.ny) return;
  // one cell after the other (the window is read from LDS twice): half the live registers of the two-cell loop
  {
    RedCell<NL, UNIFORM, 0> c0;
    c0.init(p, kx, J);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      double c[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) c[a][b] = C[l][threadIdx.y + a][threadIdx.x + b];
      c0.layer(p, l, c);
    }
    c0.finish(p);
  }
  {
    RedCell<NL, UNIFORM, 1> c1;
    c1.init(p, kx, J);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      double c[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) c[a][b] = C[l][threadIdx.y + a][threadIdx.x + b];
      c1.layer(p, l, c);
    }
    c1.finish(p);
  }
}

// ------------------------------------------------------------------ coarse levels in ONE launch
// Levels of at most MGC_MAXDIM x MGC_MAXDIM (32 x 32; measured: 64 loses, the single workgroup becomes the bottleneck) cells are launch-latency bound (a colour half-sweep of a 32^2
// level takes ~1 us of work and ~4.5 us of launch): one 512-thread workgroup (256 VGPRs per wave: the fused prolongation needs ~150) runs the whole coarse part
// of the cycle -- restrictions down, zero first guess on the coarsest level, nrelax red-black sweeps per
// level, prolongation folded into the first red half-sweep on the way up -- with __syncthreads() where
// the separate launches had kernel boundaries.  The per-point device functions are the ones of the
// stand-alone kernels, so the arithmetic is identical.
// LDS residency (round 2, lds = 1 and uniform S): a phase of a coarse level is one dependent round trip -- read the neighbours the previous
// phase wrote, solve the column, write, barrier -- and through global memory that round trip costs 2 - 3.7 us however small the level
// (measured with wall_clock64: 165 us for the 45 phases of 32^2 ... 2^2 at nl = 6).  The correction and the residual of the levels
// therefore live in a 150-KB LDS pool, in the same x-parity split layout with one pad slot per half instead of 16 (the per-point
// functions take (pointer, geometry) pairs: the pool pointer is shifted by MSOM_SP - 1 so that split_idx lands on the compact
// rows).  The pointers are formed as pool + offset under compile-time flags, so the inlined per-point code uses ds_read / ds_write
// (a generic pointer would go through the flat path: 1.4 us per phase instead of ~0.6 at nl = 3).
// The group's finest level is materialised late: its residual is restricted straight from global memory on the way down and
// copied in only when the cycle comes back up, over the residuals of the coarser levels (dead by then), which is what lets
// 32^2 x 6 layers fit.  Only that level's correction (ghosts included) goes back to global memory.  Same per-point functions,
// same values as the separate launches.
__device__ __forceinline__ SplitGeom mgc_compact(const SplitGeom &g) {
  SplitGeom c = g;
  c.hp = g.hk + 2; c.rp = 2 * c.hp; c.rows = g.ny + 2; c.ls = (size_t)c.rp * c.rows;
  return c;
}
struct MgcShared {
  CoarseLev lev[MGC_MAXLEV];          // geometry compact for the LDS-resident levels
  int da[MGC_MAXLEV], res[MGC_MAXLEV];  // pool offsets of the shifted array bases; < 0: the level's global arrays
};
template <int NL, bool FL, bool CL>
__device__ __forceinline__ void mgc_restrict(double *pool, const MgcShared &sh, const CoarseArgs &a, int kf, int tid) {
  // a global array goes with the global geometry (sh.lev[0].g is already the compact one when level 0 comes in late)
  const SplitGeom &fg = FL ? sh.lev[kf].g : a.lev[kf].g, &cg = CL ? sh.lev[kf + 1].g : a.lev[kf + 1].g;
  const double *fr = FL ? pool + sh.res[kf] : a.lev[kf].res;
  double *cr = CL ? pool + sh.res[kf + 1] : a.lev[kf + 1].res;
  for (int t = tid; t < cg.nx * cg.ny; t += MGC_NT) restrict_pt(fr, fg, cr, cg, NL, t % cg.nx, t / cg.nx);
  __syncthreads();
}
// first guess (prolongation, fused into the first red half-sweep where possible) and the nrelax sweeps of level k
template <int NL, bool UNIFORM, bool LL, bool CL>
__device__ __forceinline__ void mgc_level(double *pool, const MgcShared &sh, const CoarseArgs &a, int k, int nrelax, int tid) {
  // the fused first phase (prolongation inside the first red half-sweep) needs ~210 VGPRs at nl = 6: inside this 512-thread
  // workgroup (256 per lane) it spilled, and a prolongation phase of its own costs the same 2 us; compiled out from nl = 5 on
  constexpr bool MGC_FUSE = NL <= 4;
  const CoarseLev &L = sh.lev[k];
  double *da = LL ? pool + sh.da[k] : L.da;
  const double *res = LL ? pool + sh.res[k] : L.res;
  const bool coarsest = k == a.n - 1;
  const CoarseLev &C = sh.lev[coarsest ? k : k + 1];
  const double *cda = coarsest ? nullptr : (CL ? pool + sh.da[k + 1] : C.da);
  bool fused = false;
  if (coarsest) {  // first guess 0 (ghosts included)
    for (size_t t = tid; t < L.g.ls * NL; t += MGC_NT) da[(LL ? MSOM_SP - 1 : 0) + t] = 0.;
  } else if (MGC_FUSE && a.prolong_fused && nrelax >= 1 && L.g.nx >= 4 && L.g.ny >= 4) fused = true;
  else
    for (int t = tid; t < L.g.nx * L.g.ny; t += MGC_NT) prolong_pt(cda, C.g, da, L.g, NL, a.walls, t % L.g.nx, t / L.g.nx);
  __syncthreads();
  // a phase of a small level is one wavefront's serial instruction stream (~4.5 cycles per instruction with nothing to overlap):
  // the arguments and the (column, row) of the thread's first point are formed once per level, not once per phase
  RelaxArgs p;
  p.da = da; p.res = res; p.S = L.S; p.g = L.g; p.color = 0; p.walls = a.walls; p.rc = L.rc; p.region = 0;
  const int hk = p.g.hk, cnt = hk * p.g.ny, kx0 = tid % hk, j0 = tid / hk;
  for (int it = 0; it < nrelax; it++)
    for (int c = 0; c < 2; c++) {
      if (MGC_FUSE && fused && it == 0 && c == 0) {
        RelaxPArgs q;
        q.da = da; q.res = res; q.S = L.S; q.coarse = cda; q.g = p.g; q.cg = C.g; q.walls = a.walls; q.rc = p.rc;
        const int nj = (p.g.ny + 1) / 2;
        for (int t = tid; t < hk * nj; t += MGC_NT) red_prolong2_pt<NL, UNIFORM>(q, t % hk, t / hk);
      } else {
        p.color = c;
        if (tid < cnt) relax_color_pt<NL, UNIFORM>(p, kx0, j0);
        for (int t = tid + MGC_NT; t < cnt; t += MGC_NT) relax_color_pt<NL, UNIFORM>(p, t % hk, t / hk);
      }
      __syncthreads();
    }
}
template <int NL, bool UNIFORM>
__global__ void __launch_bounds__(MGC_NT) k_mg_coarse(const CoarseArgs *pa, int nrelax) {
  __shared__ double pool[MGC_POOL];
  __shared__ MgcShared sh;
  __shared__ int s_lds_from, s_top_late;  // levels >= s_lds_from are resident from the start; s_top_late: level 0 comes in on the way up
  const CoarseArgs &a = *pa;
  const int tid = threadIdx.x, n = a.n;
  {  // the level table: copied by all threads, patched by one
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.lev);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(sh.lev);
    for (int t = tid; t < (int)(n * sizeof(CoarseLev) / 8); t += MGC_NT) dst[t] = src[t];
  }
  __syncthreads();
  if (tid == 0) {
    int from = n, late = 0;
    for (int k = 0; k < n; k++) sh.da[k] = sh.res[k] = -1;
    if (UNIFORM && a.lds) {
      size_t sum = 0, top = 2 * mgc_compact(a.lev[0].g).ls * NL;
      for (int k = 1; k < n; k++) sum += mgc_compact(a.lev[k].g).ls * NL;
      if (MSOM_SP + 2 * sum <= MGC_POOL) {  // da of levels n-1 .. 1, then their residuals
        from = n > 1 ? 1 : n;
        size_t o = MSOM_SP;
        for (int k = n - 1; k >= 1; k--) { sh.lev[k].g = mgc_compact(a.lev[k].g); sh.da[k] = (int)(o - (MSOM_SP - 1)); o += sh.lev[k].g.ls * NL; }
        const size_t r0 = o;
        for (int k = n - 1; k >= 1; k--) { sh.res[k] = (int)(o - (MSOM_SP - 1)); o += sh.lev[k].g.ls * NL; }
        if (r0 + top <= MGC_POOL) {  // level 0 over the residual region
          late = 1;
          sh.lev[0].g = mgc_compact(a.lev[0].g);
          sh.da[0] = (int)(r0 - (MSOM_SP - 1));
          sh.res[0] = (int)(r0 + top / 2 - (MSOM_SP - 1));
        }
      }
    }
    s_lds_from = from; s_top_late = late;
  }
  __syncthreads();
  const int lds_from = s_lds_from, late = s_top_late;
  // nothing in the pool is read before this launch wrote it, so it is not cleared; lds = 2 (option mg_coarse = 3) fills it with
  // NaN first, which the parity tests use to prove exactly that
  if (a.lds == 2) {
    for (int t = tid; t < MGC_POOL; t += MGC_NT) pool[t] = __longlong_as_double(0x7ff8000000000000LL);
    __syncthreads();
  }
  // restrictions of the residual, level by level (level 0 always from global memory)
  if (n > 1) {
    if (lds_from <= 1) mgc_restrict<NL, false, true>(pool, sh, a, 0, tid);
    else mgc_restrict<NL, false, false>(pool, sh, a, 0, tid);
  }
  for (int k = 2; k < n; k++) {
    if (lds_from <= 1) mgc_restrict<NL, true, true>(pool, sh, a, k - 1, tid);
    else mgc_restrict<NL, false, false>(pool, sh, a, k - 1, tid);
  }
  for (int k = n - 1; k >= 1; k--) {
    if (lds_from <= 1) mgc_level<NL, UNIFORM, true, true>(pool, sh, a, k, nrelax, tid);
    else mgc_level<NL, UNIFORM, false, false>(pool, sh, a, k, nrelax, tid);
  }
  if (late) {  // the finest level of the group: residual in (interior cells), sweeps, correction out (ghosts included)
    const CoarseLev &G = a.lev[0], &L = sh.lev[0];
    double *lres = pool + sh.res[0], *lda = pool + sh.da[0];
    for (int t = tid; t < G.g.nx * G.g.ny * NL; t += MGC_NT) {
      const int i = t % G.g.nx, j = (t / G.g.nx) % G.g.ny, l = t / (G.g.nx * G.g.ny);
      lres[split_idx(L.g, l, j, i)] = G.res[split_idx(G.g, l, j, i)];
    }
    __syncthreads();
    mgc_level<NL, UNIFORM, true, true>(pool, sh, a, 0, nrelax, tid);
    const int w = G.g.nx + 2, h = G.g.ny + 2;
    for (int t = tid; t < w * h * NL; t += MGC_NT) {
      const int i = t % w - 1, j = (t / w) % h - 1, l = t / (w * h);
      G.da[split_idx(G.g, l, j, i)] = lda[split_idx(L.g, l, j, i)];
    }
  } else if (lds_from <= 1) mgc_level<NL, UNIFORM, false, true>(pool, sh, a, 0, nrelax, tid);
  else mgc_level<NL, UNIFORM, false, false>(pool, sh, a, 0, nrelax, tid);
}
// ---- the same group of levels, lean form (round 3; option mg_coarse = 4, the default where it applies): uniform S (or one layer), a
// single tile with walls or doubly periodic.  k_mg_coarse above runs the device functions of the stand-alone kernels on LDS copies in the
// padded split layout (generic pointers: flat loads and stores next to ds ones, 2300 branches of ghost and layout cases) and takes
// 65 us at nl = 3 / 96 us at nl = 6 for ~50 phases -- a third of a 512^2 x 3 step.  Here every level is a plain [layer][row][column]
// array in LDS with a one-cell ghost ring (correction) or none (residual), the phases are spelled out with the expressions of
// restrict_pt, prolong_pt, relax_color_pt<.., UNIFORM> and split_write_ghosts in their order (validation build: the same bits).
size_t mg_coarse_lean_doubles(const CoarseArgs &h, int nl) {
  size_t o = 0;
  for (int k = 0; k < h.n; k++) o += (size_t)nl * ((size_t)(h.lev[k].g.nx + 2) * (h.lev[k].g.ny + 2) + (size_t)h.lev[k].g.nx * h.lev[k].g.ny);
  return o;
}
#define MGC_NOGHOST (-(1 << 30))   // (ghost offsets are relative to cell (0, 0): -1 and below are real targets)
template <int NL>
__global__ void __launch_bounds__(MGC_NT) k_mg_coarse_lean(const CoarseArgs *pa, int nrelax) {
  __shared__ double pool[MGC_POOL];
  const CoarseArgs &a = *pa;
  const int tid = threadIdx.x;
  // The argument block lives in global memory (7 KB: too large for the kernel-argument segment) and every barrier is a fence: what the
  // phases need of it -- level sizes, pool offsets, the column-solver constants -- is copied to LDS once, and each level takes its
  // constants into registers before its sweeps (a phase then waits for LDS only)
  __shared__ int s_geo[MGC_MAXLEV][4];               // nx, ny, offset of the correction's cell (0, 0), offset of the residual
  __shared__ double s_co[MGC_MAXLEV][1 + 3 * NL];    // sqD, w[], it1[], t2[]
  __shared__ int s_n, s_walls;
  if (tid == 0) {
    int o = 0;
    for (int q = 0; q < a.n; q++) {
      const int nx = a.lev[q].g.nx, ny = a.lev[q].g.ny;
      s_geo[q][0] = nx; s_geo[q][1] = ny;
      s_geo[q][2] = o + (nx + 2) + 1;
      o += NL * (nx + 2) * (ny + 2);
      s_geo[q][3] = o;
      o += NL * nx * ny;
    }
    s_n = a.n; s_walls = a.walls;
  }
  if (tid < MGC_MAXLEV && tid < a.n) {
    const RelaxCoef &rc = a.lev[tid].rc;
    s_co[tid][0] = rc.sqD;
    for (int l = 0; l < NL; l++) { s_co[tid][1 + l] = rc.w[l]; s_co[tid][1 + NL + l] = rc.it1[l]; s_co[tid][1 + 2 * NL + l] = rc.t2[l]; }
  }
  __syncthreads();
  const int n = s_n;
  const bool per = (s_walls & WALL_PER) != 0;
  // offsets of level k: correction [NL][ny + 2][nx + 2] (od points at cell (0, 0) of layer 0), residual [NL][ny][nx]
  auto geom = [&](int k, int &nx, int &ny, int &od, int &orr) { nx = s_geo[k][0]; ny = s_geo[k][1]; od = s_geo[k][2]; orr = s_geo[k][3]; };
  // boundary_level: the ghosts an edge cell owns (split_write_ghosts)
  auto ghosts = [&](double *d, int nx, int ny, int j, int i, double v) {
    const int pw = nx + 2;
    const bool w = i == 0, e = i == nx - 1, s_ = j == 0, nn = j == ny - 1;
    if (!(w | e | s_ | nn)) return;
    if (per) {
      if (w) d[j * pw + nx] = v;
      if (e) d[j * pw - 1] = v;
      if (s_) d[ny * pw + i] = v;
      if (nn) d[-pw + i] = v;
      if (w && s_) d[ny * pw + nx] = v;
      if (w && nn) d[-pw + nx] = v;
      if (e && s_) d[ny * pw - 1] = v;
      if (e && nn) d[-pw - 1] = v;
    } else {
      if (w) d[j * pw - 1] = -v;
      if (e) d[j * pw + nx] = -v;
      if (s_) d[-pw + i] = -v;
      if (nn) d[ny * pw + i] = -v;
      if (w && s_) d[-pw - 1] = v;
      if (w && nn) d[ny * pw - 1] = v;
      if (e && s_) d[-pw + nx] = v;
      if (e && nn) d[ny * pw + nx] = v;
    }
  };
  int nx0, ny0, od0, or0;
  geom(0, nx0, ny0, od0, or0);
  {  // residual of the group's finest level from memory
    const CoarseLev &G = a.lev[0];
    for (int t = tid; t < nx0 * ny0 * NL; t += MGC_NT) {
      const int i = t % nx0, j = (t / nx0) % ny0, l = t / (nx0 * ny0);
      pool[or0 + (l * ny0 + j) * nx0 + i] = G.res[split_idx(G.g, l, j, i)];
    }
  }
  __syncthreads();
  for (int k = 0; k + 1 < n; k++) {   // restrictions (restrict_pt)
    int fx, fy, fd, fr, cx, cy, cd, cr;
    geom(k, fx, fy, fd, fr);
    geom(k + 1, cx, cy, cd, cr);
    for (int t = tid; t < cx * cy; t += MGC_NT) {
      const int I = t % cx, J = t / cx;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double *f = pool + fr + l * fy * fx;
        double sum = 0.;
        sum += f[(2 * J) * fx + 2 * I];
        sum += f[(2 * J + 1) * fx + 2 * I];
        sum += f[(2 * J) * fx + 2 * I + 1];
        sum += f[(2 * J + 1) * fx + 2 * I + 1];
        pool[cr + (l * cy + J) * cx + I] = sum / 4;
      }
    }
    __syncthreads();
  }
  for (int k = n - 1; k >= 0; k--) {
    int nx, ny, od, orr;
    geom(k, nx, ny, od, orr);
    const int pw = nx + 2, lsd = pw * (ny + 2);
    if (k == n - 1) {   // first guess 0, ghosts included
      for (int t = tid; t < NL * lsd; t += MGC_NT) pool[od - pw - 1 + t] = 0.;
    } else {            // bilinear prolongation + boundary_level (prolong_pt)
      int cx, cy, cd, cr;
      geom(k + 1, cx, cy, cd, cr);
      const int cpw = cx + 2, cls = cpw * (cy + 2);
      for (int t = tid; t < nx * ny; t += MGC_NT) {
        const int i = t % nx, j = t / nx;
        const int I = i >> 1, J = j >> 1, sx = (i & 1) ? 1 : -1, sy = (j & 1) ? 1 : -1;
#pragma unroll
        for (int l = 0; l < NL; l++) {
          const double *c = pool + cd + l * cls;
          double *d = pool + od + l * lsd;
          const double v = BILINEAR(c[J * cpw + I], c[J * cpw + I + sx], c[(J + sy) * cpw + I], c[(J + sy) * cpw + I + sx]);
          d[j * pw + i] = v;
          ghosts(d, nx, ny, j, i, v);
        }
      }
    }
    __syncthreads();
    double cw[NL], cit1[NL], ct2[NL];
    const double sqD = s_co[k][0];
#pragma unroll
    for (int l = 0; l < NL; l++) { cw[l] = s_co[k][1 + l]; cit1[l] = s_co[k][1 + NL + l]; ct2[l] = s_co[k][1 + 2 * NL + l]; }
    // a phase of a small level is one wavefront's serial instruction stream: the thread's cell of either colour (a level has at most
    // 32 x 32 / 2 = MGC_NT cells per colour), its offsets and whether it owns ghosts are formed once per level, not once per phase
    const int hk = nx >> 1, cells = nx * ny, lhk = hk > 0 ? 31 - __clz(hk) : 0;
    int oc[2], rc_[2], gt[2][3];
    double gsn[2][3];
    bool act[2], edge[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int i, j;
      if (nx & 1) { i = tid % nx; j = tid / nx; act[c] = tid < cells && ((i + j) & 1) == c; }   // a 1-wide level
      else { j = tid >> lhk; i = 2 * (tid - (j << lhk)) + ((j + c) & 1); act[c] = tid < hk * ny; }   // hk is a power of two (grid sizes are)
      oc[c] = od + j * pw + i;
      rc_[c] = orr + j * nx + i;
      edge[c] = act[c] && (i == 0 || i == nx - 1 || j == 0 || j == ny - 1);
      // the ghosts this cell owns, as offsets from the layer's cell (0, 0) and signs (levels of >= 2 x 2 cells: at most the two edge
      // images and the corner image; degenerate levels go through ghosts())
      gt[c][0] = gt[c][1] = gt[c][2] = MGC_NOGHOST;
      gsn[c][0] = gsn[c][1] = gsn[c][2] = 1.;
      if (edge[c] && nx >= 2 && ny >= 2) {
        const bool w = i == 0, e = i == nx - 1, s_ = j == 0, nn = j == ny - 1;
        int q = 0;
        if (per) {
          if (w) gt[c][q++] = j * pw + nx;
          if (e) gt[c][q++] = j * pw - 1;
          if (s_) gt[c][q++] = ny * pw + i;
          if (nn) gt[c][q++] = -pw + i;
          if (w && s_) gt[c][q++] = ny * pw + nx;
          if (w && nn) gt[c][q++] = -pw + nx;
          if (e && s_) gt[c][q++] = ny * pw - 1;
          if (e && nn) gt[c][q++] = -pw - 1;
        } else {
          if (w) { gsn[c][q] = -1.; gt[c][q++] = j * pw - 1; }
          if (e) { gsn[c][q] = -1.; gt[c][q++] = j * pw + nx; }
          if (s_) { gsn[c][q] = -1.; gt[c][q++] = -pw + i; }
          if (nn) { gsn[c][q] = -1.; gt[c][q++] = ny * pw + i; }
          if (w && s_) gt[c][q++] = -pw - 1;
          if (w && nn) gt[c][q++] = ny * pw - 1;
          if (e && s_) gt[c][q++] = -pw + nx;
          if (e && nn) gt[c][q++] = ny * pw + nx;
        }
      }
    }
    const bool degenerate = nx < 2 || ny < 2;
    // one half-sweep of colour c by this thread's cell
    auto phase = [&](int c) {
      if (!act[c]) return;
      double rhs[NL], x[NL];
      const int o = oc[c];
      if (NL == 1) {
        double nn = -sqD * pool[rc_[c]], dd = 0.;
        nn += pool[o + 1] + pool[o - 1]; dd += 2.;
        nn += pool[o + pw] + pool[o - pw]; dd += 2.;
        x[0] = nn / dd;
      } else {
#pragma unroll
        for (int l = 0; l < NL; l++) {
          double r = -sqD * pool[rc_[c] + l * cells];
          r += pool[o + l * lsd + 1] + pool[o + l * lsd - 1];
          r += pool[o + l * lsd + pw] + pool[o + l * lsd - pw];
          rhs[l] = r;
        }
#pragma unroll
        for (int l = 1; l < NL; l++) rhs[l] -= cw[l] * rhs[l - 1];
        x[NL - 1] = rhs[NL - 1] * cit1[NL - 1];
#pragma unroll
        for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - ct2[l] * x[l + 1]) * cit1[l];
      }
#pragma unroll
      for (int l = 0; l < NL; l++) pool[o + l * lsd] = x[l];
      if (edge[c]) {
        if (degenerate) {
          const int j = (o - od) / pw, i = (o - od) - j * pw;
#pragma unroll
          for (int l = 0; l < NL; l++) ghosts(pool + od + l * lsd, nx, ny, j, i, x[l]);
        } else {
#pragma unroll
          for (int q = 0; q < 3; q++)
            if (gt[c][q] != MGC_NOGHOST) {
#pragma unroll
              for (int l = 0; l < NL; l++) pool[od + l * lsd + gt[c][q]] = gsn[c][q] * x[l];
            }
        }
      }
    };
    const int per_colour = (nx & 1) ? cells : hk * ny;
    if (per_colour <= 64) {
      // a level of at most 64 cells per colour is one wavefront's work: its LDS operations execute in order, so the half-sweeps need no
      // workgroup barrier between them (the other wavefronts wait once, at the end of the level)
      if (tid < 64) {
        for (int it = 0; it < nrelax; it++)
#pragma unroll
          for (int c = 0; c < 2; c++) {
            phase(c);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
      }
      __syncthreads();
    } else {
      for (int it = 0; it < nrelax; it++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
          phase(c);
          __syncthreads();
        }
    }
  }
  {  // correction of the group's finest level to memory, ghosts included (the next finer level interpolates from it)
    const CoarseLev &G = a.lev[0];
    const int w = nx0 + 2, h = ny0 + 2;
    for (int t = tid; t < w * h * NL; t += MGC_NT) {
      const int i = t % w - 1, j = (t / w) % h - 1, l = t / (w * h);
      G.da[split_idx(G.g, l, j, i)] = pool[od0 + l * w * h + j * w + i];
    }
  }
}
template <int NL>
static void mg_coarse_dispatch(hipStream_t st, const CoarseArgs *d_args, int nrelax, int uniformS, int lean) {
  if (lean) { hipLaunchKernelGGL((k_mg_coarse_lean<NL>), dim3(1), dim3(MGC_NT), 0, st, d_args, nrelax); return; }
  if (uniformS) hipLaunchKernelGGL((k_mg_coarse<NL, true>), dim3(1), dim3(MGC_NT), 0, st, d_args, nrelax);
  else hipLaunchKernelGGL((k_mg_coarse<NL, false>), dim3(1), dim3(MGC_NT), 0, st, d_args, nrelax);
}
size_t mg_coarse_static_lds() { return sizeof(double) * MGC_POOL + sizeof(MgcShared); }

void launch_mg_coarse(hipStream_t st, const CoarseArgs *d_args, int nrelax, int nl, int uniformS, int lean) {
  switch (nl) {
    case 1: mg_coarse_dispatch<1>(st, d_args, nrelax, uniformS, lean); break;
    case 2: mg_coarse_dispatch<2>(st, d_args, nrelax, uniformS, lean); break;
    case 3: mg_coarse_dispatch<3>(st, d_args, nrelax, uniformS, lean); break;
    case 4: mg_coarse_dispatch<4>(st, d_args, nrelax, uniformS, lean); break;
    case 5: mg_coarse_dispatch<5>(st, d_args, nrelax, uniformS, lean); break;
    case 6: mg_coarse_dispatch<6>(st, d_args, nrelax, uniformS, lean); break;
    case 7: mg_coarse_dispatch<7>(st, d_args, nrelax, uniformS, lean); break;
    case 8: mg_coarse_dispatch<8>(st, d_args, nrelax, uniformS, lean); break;
    default: break;
  }
}


template <int NL, bool UNIFORM>
__global__ void __launch_bounds__(BX *BY) k_relax_red_prolong(RelaxPArgs p) {
  if (blockIdx.z == 0) relax_red_prolong_body<NL, UNIFORM, 0>(p);
  else relax_red_prolong_body<NL, UNIFORM, 1>(p);
}
template <int NL>
static void relax_red_prolong_dispatch(hipStream_t st, const RelaxPArgs &p, int uniformS) {
  dim3 gr = grid2d(p.g.hk, (p.g.ny + 1) / 2);
  extern int g_rhs_dbg;
  if (!(g_rhs_dbg & 256) && p.g.hk >= 128 && !(g_rhs_dbg & 512)) {  // wide levels: coarse windows through LDS
    if (uniformS) hipLaunchKernelGGL((k_relax_red_prolong3<NL, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_red_prolong3<NL, false>), gr, block2d(), 0, st, p);
    return;
  }
  if (!(g_rhs_dbg & 256)) {  // both row parities per thread
    if (uniformS) hipLaunchKernelGGL((k_relax_red_prolong2<NL, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_red_prolong2<NL, false>), gr, block2d(), 0, st, p);
    return;
  }
  gr.z = 2;
  if (uniformS) hipLaunchKernelGGL((k_relax_red_prolong<NL, true>), gr, block2d(), 0, st, p);
  else hipLaunchKernelGGL((k_relax_red_prolong<NL, false>), gr, block2d(), 0, st, p);
}
void launch_relax_red_prolong(hipStream_t st, double *da, const double *coarse, const SplitGeom &cg, const double *res, const double *S,
                              const SplitGeom &sg, int nl, const RelaxCoef &rc, int uniformS, int walls) {
  RelaxPArgs p;
  p.da = da; p.res = res; p.S = S; p.coarse = coarse; p.g = sg; p.cg = cg; p.walls = walls; p.rc = rc;
  switch (nl) {
    case 1: relax_red_prolong_dispatch<1>(st, p, uniformS); break;
    case 2: relax_red_prolong_dispatch<2>(st, p, uniformS); break;
    case 3: relax_red_prolong_dispatch<3>(st, p, uniformS); break;
    case 4: relax_red_prolong_dispatch<4>(st, p, uniformS); break;
    case 5: relax_red_prolong_dispatch<5>(st, p, uniformS); break;
    case 6: relax_red_prolong_dispatch<6>(st, p, uniformS); break;
    case 7: relax_red_prolong_dispatch<7>(st, p, uniformS); break;
    case 8: relax_red_prolong_dispatch<8>(st, p, uniformS); break;
    default: break;
  }
}

template <int NL>
static void relax_dispatch(hipStream_t st, const RelaxArgs &p, int uniformS, int fine) {
  if (p.g.hk % 2 == 0 && p.g.hk >= 128) {  // wide levels: two points per thread, 16-byte accesses
    dim3 g2 = grid2d(p.g.hk / 2, p.g.ny);
    if (uniformS) {
      if (fine) hipLaunchKernelGGL((k_relax_color_x2<NL, true, true>), g2, block2d(), 0, st, p);
      else hipLaunchKernelGGL((k_relax_color_x2<NL, true, false>), g2, block2d(), 0, st, p);
    } else {
      if (fine) hipLaunchKernelGGL((k_relax_color_x2<NL, false, true>), g2, block2d(), 0, st, p);
      else hipLaunchKernelGGL((k_relax_color_x2<NL, false, false>), g2, block2d(), 0, st, p);
    }
    return;
  }
  dim3 gr = grid2d(p.g.hk, p.g.ny);
  if (uniformS) {
    if (fine) hipLaunchKernelGGL((k_relax_color<NL, true, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_color<NL, true, false>), gr, block2d(), 0, st, p);
  } else {
    if (fine) hipLaunchKernelGGL((k_relax_color<NL, false, true>), gr, block2d(), 0, st, p);
    else hipLaunchKernelGGL((k_relax_color<NL, false, false>), gr, block2d(), 0, st, p);
  }
}
template <int NL>
static void relax_dispatch_wide(hipStream_t st, const RelaxArgs &p, int uniformS) {
  dim3 gr = grid2d(p.g.hk, p.g.ny);
  if (uniformS) hipLaunchKernelGGL((k_relax_color<NL, true, false>), gr, block2d(), 0, st, p);
  else hipLaunchKernelGGL((k_relax_color<NL, false, false>), gr, block2d(), 0, st, p);
}
template <int NL>
static void relax_ring_dispatch(hipStream_t st, const RelaxArgs &p, int uniformS) {
  const int n = 2 * p.g.hk + p.g.ny - 2;
  if (uniformS) hipLaunchKernelGGL((k_relax_ring<NL, true>), dim3((n + 255) / 256), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_relax_ring<NL, false>), dim3((n + 255) / 256), dim3(256), 0, st, p);
}
void launch_relax_ring(hipStream_t st, double *da, const double *res, const double *S, const SplitGeom &sg, int nl, const RelaxCoef &rc,
                       int uniformS, int color, int walls) {
  RelaxArgs p;
  p.da = da; p.res = res; p.S = S; p.g = sg; p.color = color; p.walls = walls; p.rc = rc; p.region = 0;
  switch (nl) {
    case 1: relax_ring_dispatch<1>(st, p, uniformS); break;
    case 2: relax_ring_dispatch<2>(st, p, uniformS); break;
    case 3: relax_ring_dispatch<3>(st, p, uniformS); break;
    case 4: relax_ring_dispatch<4>(st, p, uniformS); break;
    case 5: relax_ring_dispatch<5>(st, p, uniformS); break;
    case 6: relax_ring_dispatch<6>(st, p, uniformS); break;
    case 7: relax_ring_dispatch<7>(st, p, uniformS); break;
    case 8: relax_ring_dispatch<8>(st, p, uniformS); break;
    case 9: relax_ring_dispatch<9>(st, p, uniformS); break;
    case 10: relax_ring_dispatch<10>(st, p, uniformS); break;
    case 11: relax_ring_dispatch<11>(st, p, uniformS); break;
    case 12: relax_ring_dispatch<12>(st, p, uniformS); break;
    case 13: relax_ring_dispatch<13>(st, p, uniformS); break;
    case 14: relax_ring_dispatch<14>(st, p, uniformS); break;
    case 15: relax_ring_dispatch<15>(st, p, uniformS); break;
    case 16: relax_ring_dispatch<16>(st, p, uniformS); break;
    default: break;
  }
}
void launch_relax_color(hipStream_t st, double *da, const double *res, const double *S, const SplitGeom &sg, int nl, const RelaxCoef &rc,
                        int uniformS, int color, int walls, int fine, int region) {
  RelaxArgs p;
  p.da = da; p.res = res; p.S = S; p.g = sg; p.color = color; p.walls = walls; p.rc = rc; p.region = region;
  switch (nl) {
    case 1: relax_dispatch<1>(st, p, uniformS, fine); break;
    case 2: relax_dispatch<2>(st, p, uniformS, fine); break;
    case 3: relax_dispatch<3>(st, p, uniformS, fine); break;
    case 4: relax_dispatch<4>(st, p, uniformS, fine); break;
    case 5: relax_dispatch<5>(st, p, uniformS, fine); break;
    case 6: relax_dispatch<6>(st, p, uniformS, fine); break;
    case 7: relax_dispatch<7>(st, p, uniformS, fine); break;
    case 8: relax_dispatch<8>(st, p, uniformS, fine); break;
    // nl > MSOM_FASTNL (round 3): the one-column-per-thread kernel only, column systems of up to MSOM_MAXNL layers in registers
    case 9: relax_dispatch_wide<9>(st, p, uniformS); break;
    case 10: relax_dispatch_wide<10>(st, p, uniformS); break;
    case 11: relax_dispatch_wide<11>(st, p, uniformS); break;
    case 12: relax_dispatch_wide<12>(st, p, uniformS); break;
    case 13: relax_dispatch_wide<13>(st, p, uniformS); break;
    case 14: relax_dispatch_wide<14>(st, p, uniformS); break;
    case 15: relax_dispatch_wide<15>(st, p, uniformS); break;
    case 16: relax_dispatch_wide<16>(st, p, uniformS); break;
    default: break;  // rejected at create time (MSOM_MAXNL)
  }
}

// ------------------------------------------------------------------ temporally blocked smoother

// Two full red-black sweeps (4 colour half-sweeps) in ONE pass over HBM.  A workgroup loads
// a BTX x BTY tile of the correction with a 4-cell halo into LDS, keeps the residual of the
// cells it owns in registers, runs the 4 half-sweeps on a region that shrinks by one cell per
// half-sweep (a cell is updated only when its 4 neighbours carry the value of the previous
// half-sweep, so every update equals the one of the global sweep bit for bit), and stores the
// tile interior.  HBM traffic per 2 sweeps drops from 2 x (3 w) to about 3.4 w.
// Out of place (da_in -> da_out): neighbouring workgroups read each other's interiors.
// With PROLONG the tile is not read from da_in but interpolated from the next coarser
// level on the fly (mg_cycle's bilinear prolongation + boundary_level, mspg/elliptic.h:74-82).
// Walls: ghost cells lag exactly as in k_relax_color (the owner rewrites its ghost).
// Round 3: the halo BH is a template parameter and the number of half-sweeps (nh <= BH) and their first colour (c0) are arguments:
// with BH = 8 one launch does the prolongation and all 8 half-sweeps of a level visit (nrelax = 4), which on the launch-bound
// levels (64^2 .. 512^2 cells: ~5 us per colour pass whatever the size) replaces 8 launches.
#define BTX 64
struct BlockArgs {
  const double *da_in, *res, *coarse;
  double *da_out;
  SplitGeom g, cg;
  int walls;
  int nh, c0;   // half-sweeps of this pass (<= BH), colour of the first
  const double *S;   // GENERAL instantiation: the level's S field (split layout, nl - 1 layers), read once per owned cell
  RelaxCoef rc;
};

// GENERAL (round 3): the column system of relax_color_pt<NL, false> -- S per cell, three divisions per layer -- instead of the
// constant-coefficient one; S of the two owned cells stays in registers beside their residuals
template <int NL, int BTY, int BNT, bool PROLONG, bool FINE, int BH = 4, int BTXT = BTX, bool GENERAL = false>
__global__ void __launch_bounds__(BNT) k_relax_block(BlockArgs p) {
  constexpr int NX = BTXT + 2 * BH, NY = BTY + 2 * BH, HX = NX / 2, LS = NY * 2 * HX;
  constexpr int NPOS = (HX * NY + BNT - 1) / BNT;
  __shared__ double sA[NL * LS];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * BTXT, y0 = blockIdx.y * BTY;
  const int nx = p.g.nx, ny = p.g.ny;
  const bool wW = (p.walls & WALL_W) && x0 == 0, wE = (p.walls & WALL_E) && x0 + BTXT >= nx;
  const bool wS = (p.walls & WALL_S) && y0 == 0, wN = (p.walls & WALL_N) && y0 + BTY >= ny;
  const bool edge_tile = wW | wE | wS | wN;
  // doubly periodic single tile (round 3): a cell of the region beyond the domain is the periodic image of a domain cell -- loaded
  // (or interpolated) from its wrapped position and relaxed like any other; nothing but the tile interior is stored, the ghost
  // lines are refreshed by the caller (launch_split_wrap) after the last pass of the visit.  Needs nx, ny >= the region
  const bool per = (p.walls & WALL_PER) != 0;
  const double sqD = p.rc.sqD;

  // ---- load / interpolate the tile (+ halo), fetch the residual of the owned cells.
  // A thread owns NPOS positions (row yy, pair index k) = the cells (yy, 2k) and (yy, 2k+1):
  // exactly one cell of each colour per position, so every lane works in every half-sweep.
  double rres[NPOS][2][NL];
  double rS[GENERAL ? NPOS : 1][2][NL > 1 ? NL - 1 : 1];
  int ob[NPOS];  // LDS index of (layer 0, row yy, parity 0, k)
#pragma unroll
  for (int n = 0; n < NPOS; n++) {
    const int s = tid + n * BNT;
    const int yy = s / HX, k = s - yy * HX;
    const int gy = y0 - BH + yy;
    const bool act = s < HX * NY;
    ob[n] = yy * 2 * HX + k;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int xx = 2 * k + c, gx = x0 - BH + xx;
      const int wx = per ? (gx + nx) % nx : gx, wy = per ? (gy + ny) % ny : gy;
      const bool inb = act && (per || (gx >= -1 && gx <= nx && gy >= -1 && gy <= ny));          // stored cell or ghost line
      const bool ind = act && (per || (gx >= 0 && gx < nx && gy >= 0 && gy < ny));               // cell of the domain (or an image of one)
      const size_t gsrc = split_idx(p.g, 0, wy, wx);
      size_t c00 = 0, c10 = 0, c01 = 0, c11 = 0;
      bool pv = false;
      if (PROLONG) {
        pv = inb && (ind || !((gx < 0 || gx >= nx) && (gy < 0 || gy >= ny)));
        // ghost line: homogeneous Dirichlet image of the interpolated wall cell
        const int mx = per ? wx : (gx < 0 ? 0 : (gx >= nx ? nx - 1 : gx)), my = per ? wy : (gy < 0 ? 0 : (gy >= ny ? ny - 1 : gy));
        const int I = mx >> 1, J = my >> 1, cx = (mx & 1) ? 1 : -1, cy = (my & 1) ? 1 : -1;
        c00 = split_idx(p.cg, 0, J, I); c10 = split_idx(p.cg, 0, J, I + cx);
        c01 = split_idx(p.cg, 0, J + cy, I); c11 = split_idx(p.cg, 0, J + cy, I + cx);
      }
#pragma unroll
      for (int l = 0; l < NL; l++) {
        double v = 0.;
        if (PROLONG) {
          if (pv) {
            const double *cc = p.coarse + (size_t)l * p.cg.ls;
            v = BILINEAR(cc[c00], cc[c10], cc[c01], cc[c11]);
            if (!ind) v = -v;
          }
        } else if (inb)
          v = p.da_in[gsrc + (size_t)l * p.g.ls];
        if (act) sA[l * LS + ob[n] + c * HX] = v;
        rres[n][c][l] = ind ? p.res[gsrc + (size_t)l * p.g.ls] : 0.;
        if (GENERAL && NL > 1 && l < NL - 1) rS[n][c][l] = ind ? p.S[gsrc + (size_t)l * p.g.ls] : 0.;
      }
    }
  }
  __syncthreads();

  // ---- 4 colour half-sweeps on the shrinking region
#pragma unroll 1
  for (int h = 1; h <= p.nh; h++) {
    const int col = (p.c0 + h - 1) & 1;
#pragma unroll
    for (int n = 0; n < NPOS; n++) {
      const int s = tid + n * BNT;
      const int yy = s / HX, k = s - yy * HX;
      const int c = (yy + col) & 1;  // which of the two owned cells has this colour
      const int xx = 2 * k + c, gx = x0 - BH + xx, gy = y0 - BH + yy;
      const bool ok = s < HX * NY && (per || (gx >= 0 && gx < nx && gy >= 0 && gy < ny)) && (xx >= h || wW) && (NX - 1 - xx >= h || wE) &&
                      (yy >= h || wS) && (NY - 1 - yy >= h || wN);
      if (ok) {
        const int o = ob[n];
        const int own = o + c * HX, iw = c ? o : o + HX - 1, ie = c ? o + 1 : o + HX;
        double rhs[NL], x[NL];
        if (NL == 1) {
          double nn = -sqD * (c ? rres[n][1][0] : rres[n][0][0]), d = 0.;
          nn += sA[ie] + sA[iw]; d += 2.;
          nn += sA[own + 2 * HX] + sA[own - 2 * HX]; d += 2.;
          x[0] = nn / d;
        } else if (GENERAL) {   // relax_color_pt<NL, false>, expression for expression
          double t0[NL], t1[NL], t2[NL];
#pragma unroll
          for (int l = 0; l < NL; l++) {
            rhs[l] = -sqD * (c ? rres[n][1][l] : rres[n][0][l]);
            t0[l] = l > 0 ? -sqD * (c ? rS[n][1][l > 0 ? l - 1 : 0] : rS[n][0][l > 0 ? l - 1 : 0]) * p.rc.idh0[l] : 0.;
            t2[l] = l < NL - 1 ? -sqD * (c ? rS[n][1][l < NL - 1 ? l : 0] : rS[n][0][l < NL - 1 ? l : 0]) * p.rc.idh1[l] : 0.;
            t1[l] = l == 0 ? -t2[l] : (l < NL - 1 ? -t0[l] - t2[l] : -t0[l]);
            rhs[l] += 1. * sA[l * LS + ie] + 1. * sA[l * LS + iw];
            t1[l] += 1. + 1.;
            rhs[l] += 1. * sA[l * LS + own + 2 * HX] + 1. * sA[l * LS + own - 2 * HX];
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
        } else {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            double r = -sqD * (c ? rres[n][1][l] : rres[n][0][l]);
            r += sA[l * LS + ie] + sA[l * LS + iw];
            r += sA[l * LS + own + 2 * HX] + sA[l * LS + own - 2 * HX];
            rhs[l] = r;
          }
#pragma unroll
          for (int l = 1; l < NL; l++) rhs[l] -= p.rc.w[l] * rhs[l - 1];
          x[NL - 1] = rhs[NL - 1] * p.rc.it1[NL - 1];
#pragma unroll
          for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - p.rc.t2[l] * x[l + 1]) * p.rc.it1[l];
        }
#pragma unroll
        for (int l = 0; l < NL; l++) sA[l * LS + own] = x[l];
        if (edge_tile) {  // the owner rewrites the lagged wall ghosts it mirrors
          const bool gw = gx == 0 && wW, ge = gx == nx - 1 && wE, gs = gy == 0 && wS, gn = gy == ny - 1 && wN;
#pragma unroll
          for (int l = 0; l < NL; l++) {
            if (gw) sA[l * LS + iw] = -x[l];
            if (ge) sA[l * LS + ie] = -x[l];
            if (gs) sA[l * LS + own - 2 * HX] = -x[l];
            if (gn) sA[l * LS + own + 2 * HX] = -x[l];
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- store the tile interior (+ wall ghosts)
#pragma unroll
  for (int n = 0; n < NPOS; n++) {
    const int s = tid + n * BNT;
    const int yy = s / HX, k = s - yy * HX;
    if (s >= HX * NY || yy < BH || yy >= BH + BTY) continue;
    const int gy = y0 - BH + yy;
    if (gy >= ny) continue;   // (tiles start at multiples of the tile size: gy, gx >= 0 here)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int xx = 2 * k + c, gx = x0 - BH + xx;
      if (xx < BH || xx >= BH + BTXT || gx >= nx) continue;
      const size_t gdst = split_idx(p.g, 0, gy, gx);
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double v = sA[l * LS + ob[n] + c * HX];
        p.da_out[gdst + (size_t)l * p.g.ls] = v;
        if (edge_tile) split_write_ghosts(p.da_out, p.g, l, gy, gx, v, p.walls);
      }
    }
  }
}

int g_block_variant = 0;  // tuning knob (tools/bench_kernels.py)
template <int NL, int BTY, int BNT>
static void block_launch(hipStream_t st, const BlockArgs &p, int prolong, int fine) {
  dim3 gr((p.g.nx + BTX - 1) / BTX, (p.g.ny + BTY - 1) / BTY);
  if (prolong) {
    if (fine) hipLaunchKernelGGL((k_relax_block<NL, BTY, BNT, true, true>), gr, dim3(BNT), 0, st, p);
    else hipLaunchKernelGGL((k_relax_block<NL, BTY, BNT, true, false>), gr, dim3(BNT), 0, st, p);
  } else {
    if (fine) hipLaunchKernelGGL((k_relax_block<NL, BTY, BNT, false, true>), gr, dim3(BNT), 0, st, p);
    else hipLaunchKernelGGL((k_relax_block<NL, BTY, BNT, false, false>), gr, dim3(BNT), 0, st, p);
  }
}
template <int NL>
static void block_dispatch(hipStream_t st, const BlockArgs &p, int prolong, int fine) {
  // LDS: NL * (BTY + 8) * 72 * 8 B; two workgroups per CU need <= 80 KiB each
  if (NL == 6) {
    switch (g_block_variant) {
      case 1: block_launch<NL, 8, 256>(st, p, prolong, fine); return;
      case 2: block_launch<NL, 8, 512>(st, p, prolong, fine); return;
      case 3: block_launch<NL, 16, 1024>(st, p, prolong, fine); return;
      case 4: block_launch<NL, 4, 256>(st, p, prolong, fine); return;
      default: break;
    }
  }
  constexpr int BTY = NL <= 3 ? 32 : 16;
  block_launch<NL, BTY, 512>(st, p, prolong, fine);
}
// two full sweeps: da_out = RB^2(da_in or prolong(coarse)); uniform-S constant-coefficient path
void launch_relax_block2(hipStream_t st, const double *da_in, const double *coarse, const SplitGeom &cg, const double *res, double *da_out,
                         const SplitGeom &sg, int nl, const RelaxCoef &rc, int walls, int fine) {
  BlockArgs p;
  p.da_in = da_in; p.res = res; p.coarse = coarse; p.da_out = da_out; p.g = sg; p.cg = cg; p.walls = walls; p.rc = rc;
  p.nh = 4; p.c0 = 0; p.S = nullptr;
  const int prolong = coarse != nullptr;
  switch (nl) {
    case 1: block_dispatch<1>(st, p, prolong, fine); break;
    case 2: block_dispatch<2>(st, p, prolong, fine); break;
    case 3: block_dispatch<3>(st, p, prolong, fine); break;
    case 4: block_dispatch<4>(st, p, prolong, fine); break;
    case 5: block_dispatch<5>(st, p, prolong, fine); break;
    case 6: block_dispatch<6>(st, p, prolong, fine); break;
    case 7: block_dispatch<7>(st, p, prolong, fine); break;
    case 8: block_dispatch<8>(st, p, prolong, fine); break;
    default: break;
  }
}

// up to 8 half-sweeps starting with colour c0 (+ the prolongation from `coarse` when given): 64 x 16 tile, halo 8, 640 threads
// LDS nl x region doubles (16 x 16 tiles: 32 x 32, 32 x 16 tiles: 48 x 32; 98 KB at nl = 8).  Returns -1 where the kernel does not exist (nl > 8)
template <int NL, int TX, int TY, int NT, bool GEN = false>
static void block8_launch_t(hipStream_t st, const BlockArgs &p, int prolong) {
  dim3 gr((p.g.nx + TX - 1) / TX, (p.g.ny + TY - 1) / TY);
  if (prolong) hipLaunchKernelGGL((k_relax_block<NL, TY, NT, true, false, 8, TX, GEN>), gr, dim3(NT), 0, st, p);
  else hipLaunchKernelGGL((k_relax_block<NL, TY, NT, false, false, 8, TX, GEN>), gr, dim3(NT), 0, st, p);
}
// tile shape: the launch-bound levels have few tiles and the pass lasts as long as ONE workgroup does, so small tiles (16 x 16: four
// times the half-sweep work of the level in halo cells, but a quarter of the serial work per workgroup) win up to 256^2; wider levels
// take 32 x 16.  Measured at nl = 6 / nl = 3 (block_variant 6 = 64 x 16 everywhere, 3 = 16 x 16 everywhere): 4096^2 x 6 6.55 / 6.46 /
// 6.39 ms per step with 64 x 16 / 16 x 16 / this rule, 512^2 x 3 0.447 / 0.400 / 0.388; 64 x 16 on the 1024^2 level only: 6.49 vs 6.43
template <int NL>
static void block8_launch(hipStream_t st, const BlockArgs &p, int prolong) {
  if (p.S) {   // general S field: the default tile shapes only
    if constexpr (NL > 1) {
      if (p.g.nx <= 256) block8_launch_t<NL, 16, 16, 512, true>(st, p, prolong);
      else block8_launch_t<NL, 32, 16, 768, true>(st, p, prolong);
    }
    return;
  }
  switch (g_block_variant) {
    case 1: block8_launch_t<NL, 32, 16, 768>(st, p, prolong); return;
    case 2: block8_launch_t<NL, 32, 8, 576>(st, p, prolong); return;
    case 3: block8_launch_t<NL, 16, 16, 512>(st, p, prolong); return;
    case 6:
      if constexpr (NL <= 6) { block8_launch_t<NL, 64, 16, 640>(st, p, prolong); return; }   // (164 KB of LDS at nl = 8)
      break;
    default: break;
  }
  if (p.g.nx <= 256) block8_launch_t<NL, 16, 16, 512>(st, p, prolong);
  else block8_launch_t<NL, 32, 16, 768>(st, p, prolong);
}
int launch_relax_block8(hipStream_t st, const double *da_in, const double *coarse, const SplitGeom &cg, const double *res, double *da_out,
                        const SplitGeom &sg, int nl, const RelaxCoef &rc, int walls, int nh, int c0, const double *S) {
  if (nh < 1 || nh > 8 || nl > MSOM_FASTNL) return -1;
  BlockArgs p;
  p.S = nl > 1 ? S : nullptr;   // non-null: the general column solver
  p.da_in = da_in; p.res = res; p.coarse = coarse; p.da_out = da_out; p.g = sg; p.cg = cg; p.walls = walls; p.rc = rc;
  p.nh = nh; p.c0 = c0;
  const int prolong = coarse != nullptr;
  switch (nl) {
    case 1: block8_launch<1>(st, p, prolong); break;
    case 2: block8_launch<2>(st, p, prolong); break;
    case 3: block8_launch<3>(st, p, prolong); break;
    case 4: block8_launch<4>(st, p, prolong); break;
    case 5: block8_launch<5>(st, p, prolong); break;
    case 6: block8_launch<6>(st, p, prolong); break;
    case 7: block8_launch<7>(st, p, prolong); break;
    case 8: block8_launch<8>(st, p, prolong); break;
    default: return -1;
  }
  return 0;
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
