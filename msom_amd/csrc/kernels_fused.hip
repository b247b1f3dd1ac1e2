// kernels_fused.hip -- the whole PV tendency dq/dt = F(psi) in ONE pass over psi.
//
// Replaces, for the common configuration (no large-scale flow, no topography, deterministic
// model), the chain comp_del2 -> advection_pv (+ comp_vel/timestep reduction) -> dissip
// (comp_stretch, comp_del2, axpy, comp_stretch, comp_del2) -> ekman_friction ->
// surface_forcing -> qforcing of update_qg (msqg/qg.h:622-630), i.e. 9 kernels and ~25 field
// passes, by one kernel that reads psi once and writes dq once.
//
// Blocking: a workgroup owns a FTX x FTY horizontal tile and marches through the layers.
// LDS holds psi_l and psi_{l+1} with a 3-cell halo (two buffers that swap roles), zeta_l =
// lap(psi_l) on a 2-cell halo and tmp_l = lap(zeta_l) on a 1-cell halo; the biharmonic term
// lap(tmp_l), the Arakawa Jacobians J(psi_l, zeta_l) and J(psi_l, psi_{l+1}), the beta term
// and the face velocities all come out of LDS.  Vertical coupling (stretching of zeta and
// tmp, ju = -jd carry) lives in registers; a layer is finalised one iteration late, when the
// layer below it is known, so that the additions happen in the reference's order.
//
// Boundary conditions are applied in LDS exactly where the reference calls boundary():
// zeta and tmp outside a wall are the Dirichlet ghosts (-interior; corner = +interior),
// optionally overridden by the partial-slip formula (msqg/qg.h:185-198).  Outside a tile edge
// that is not a wall the values are computed from the exchanged 3-cell halo of psi.
#include "rhs_inl.h"

#define FTX 64            // tile width  = one wavefront
// FTY = tile height, FNT = threads per workgroup are template parameters (tuning variants)
#define PW (FTX + 6)
#define ZW (FTX + 4)
#define TW (FTX + 2)

struct RhsArgs {
  const double *psi, *S, *qforc, *wind, *q_in;
  double *dq, *umax_partial, *q_out;  // q_out != 0: advance fused, q_out = q_in + dt * dq (msqg/qg.h:602), dq not stored
  double dt;
  int dbg;  // timing experiments only (tools/bench_kernels.py): 1 skip lap passes, 2 skip centre loop, 4 skip psi fetch
  NatGeom g;
  int nl, walls, uniformS, have_qforc;
  double D, beta, iRe, iRe4, cs, cb, slip_c;
  LayerCoef lc;
  double Su[MSOM_MAXNL];
  // optional by-product of the fused advance (k_rhs_fused_pipe): the multigrid residual of the NEXT
  // inversion, res = q_out - (lap + Gamma) psi (residual_layer, msqg/poisson_layer.h:182-255, same
  // expression sequence as k_residual2), in the split layout, its restriction to level 1, max|res| and the
  // per-block sums of q_out.  psi is the first guess of that inversion and already sits in LDS here.
  double *res, *res_c, *res_max, *bsum_partial;
  SplitGeom sg, cg;
};

// value of lane + 1 (valid in even lanes): DPP row_shl:1, two 32-bit moves, no LDS crossbar traffic
__device__ __forceinline__ double lane_next(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x101, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x101, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_f(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// One element of lap(src) -> dst with the wall boundary conditions.  dst is a W x H LDS tile
// with halo h around the block at (x0, y0); src is an LDS tile of width SW and halo
// hs = h + 1.  Positions outside a wall are the ghost cells of boundary():
// sign * lap(src)(mirror) (edges -, corners +), or the partial-slip value
// c * (src(mirror) - src(ghost)) on the first ghost line (msqg/qg.h:185-198).
template <int W, int H, int h, int SW, int hs>
__device__ __forceinline__ double lap_bc_elem(const double *src, int li, int lj, int x0, int y0, int nx, int ny, int walls, double slip_c,
                                              double D2, double rD2) {
  const int gi = x0 + li - h, gj = y0 + lj - h;
  const bool ox = (gi < 0 && (walls & WALL_W)) || (gi >= nx && (walls & WALL_E));
  const bool oy = (gj < 0 && (walls & WALL_S)) || (gj >= ny && (walls & WALL_N));
  int ci = li + (hs - h), cj = lj + (hs - h);  // same position in src coordinates
  bool neg = false;
  if (ox | oy) {
    const int mi = ox ? (gi < 0 ? -1 - gi : 2 * nx - 1 - gi) : gi;
    const int mj = oy ? (gj < 0 ? -1 - gj : 2 * ny - 1 - gj) : gj;
    const int ti = mi - x0 + hs, tj = mj - y0 + hs;
    if (!(ti >= 1 && ti < SW - 1 && tj >= 1 && tj < H + 2 * (hs - h) - 1)) return 0.;
    if (ox != oy) {
      neg = true;
      if (slip_c > 0. && (ox ? (gi == -1 || gi == nx) : (gj == -1 || gj == ny))) return slip_c * (src[tj * SW + ti] - src[cj * SW + ci]);
    }
    ci = ti; cj = tj;
  }
  const int c = cj * SW + ci;
  const double v = DIVC(src[c + 1] + src[c - 1] + src[c + SW] + src[c - SW] - 4 * src[c], D2, rD2);
  return neg ? -v : v;
}

// lap(src) -> dst for the whole tile.  Row mapping: wave w takes rows w, w + NW, ...; lane =
// column (the W - 64 extra columns are a second short pass), so there is no index division.
// `bc` is block-uniform: only workgroups whose halo touches a wall run the BC variant.
template <int W, int H, int h, int SW, int hs, int FNT>
__device__ __forceinline__ void lds_lap(double *dst, const double *src, bool bc, int x0, int y0, int nx, int ny, int walls, double slip_c,
                                        double D2, double rD2) {
  constexpr int NW = FNT / 64, XE = W - 64, o = hs - h;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (!bc) {
#pragma unroll
    for (int lj = w; lj < H; lj += NW) {
      const int c = (lj + o) * SW + lane + o;
      dst[lj * W + lane] = DIVC(src[c + 1] + src[c - 1] + src[c + SW] + src[c - SW] - 4 * src[c], D2, rD2);
    }
    for (int e = threadIdx.x; e < XE * H; e += FNT) {
      const int lj = e / XE, li = 64 + e % XE;
      const int c = (lj + o) * SW + li + o;
      dst[lj * W + li] = DIVC(src[c + 1] + src[c - 1] + src[c + SW] + src[c - SW] - 4 * src[c], D2, rD2);
    }
  } else {
    for (int lj = w; lj < H; lj += NW) dst[lj * W + lane] = lap_bc_elem<W, H, h, SW, hs>(src, lane, lj, x0, y0, nx, ny, walls, slip_c, D2, rD2);
    for (int e = threadIdx.x; e < XE * H; e += FNT) {
      const int lj = e / XE, li = 64 + e % XE;
      dst[lj * W + li] = lap_bc_elem<W, H, h, SW, hs>(src, li, lj, x0, y0, nx, ny, walls, slip_c, D2, rD2);
    }
  }
}

template <int FTY, int FNT, int MINW>
__global__ void __launch_bounds__(FNT, MINW) k_rhs_fused(RhsArgs a) {
  constexpr int RPT = FTX * FTY / FNT;  // consecutive rows per thread (sliding 3x3 windows)
  constexpr int PH = FTY + 6, ZH = FTY + 4, TH = FTY + 2;
  __shared__ double sP[2][PW * PH];
  __shared__ double sZ[ZW * ZH];
  __shared__ double sT[TW * TH];
  __shared__ double sM[FNT / 64][MSOM_MAXNL];

  const int tid = threadIdx.x, tx = tid & (FTX - 1), ly0 = (tid / FTX) * RPT;
  const int x0 = blockIdx.x * FTX, y0 = blockIdx.y * FTY;
  const int nx = a.g.nx, ny = a.g.ny, nl = a.nl, pitch = a.g.pitch;
  const double D = a.D, D2 = D * D, rD2 = 1. / D2, D12 = 12. * D * D, rD12 = 1. / D12, D2x = 2 * D, rD2x = 1. / D2x, rD = 1. / D;
  const int gi = x0 + tx;

  // per-point registers: values of the previous two layers needed to finalise a layer
  double t_prev[RPT], lapT_prev[RPT], zc0[RPT], zc1[RPT], tc0[RPT], tc1[RPT], jd_prev[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) t_prev[k] = lapT_prev[k] = zc0[k] = zc1[k] = tc0[k] = tc1[k] = jd_prev[k] = 0.;

  // psi tile (3-cell halo) of one layer: global -> registers, registers -> LDS.  Row mapping
  // like lds_lap: wave w loads rows w, w + NW, ... (64 columns), then the 6 extra columns.
  constexpr int NW = FNT / 64, NR = (PH + NW - 1) / NW, NE = (6 * PH + FNT - 1) / FNT;
  const int lane = tid & 63, wv = tid >> 6;
  const bool whole = x0 + FTX + 3 <= nx + 3 && y0 + FTY + 3 <= ny + 3;  // tile + halo inside the padded array
  double pf[NR + NE];
  auto fetch = [&](int l) {
    const double *p = a.psi + (size_t)l * a.g.ls + (ptrdiff_t)(y0 - 3 + MSOM_YP) * pitch + (x0 - 3 + MSOM_XP);
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int lj = wv + r * NW;
      pf[r] = (lj < PH && (whole || (x0 + lane - 3 < nx + 3 && y0 + lj - 3 < ny + 3))) ? p[(ptrdiff_t)lj * pitch + lane] : 0.;
    }
#pragma unroll
    for (int r = 0; r < NE; r++) {
      const int e = tid + r * FNT, lj = e / 6, li = 64 + e % 6;
      pf[NR + r] = (e < 6 * PH && (whole || (x0 + li - 3 < nx + 3 && y0 + lj - 3 < ny + 3))) ? p[(ptrdiff_t)lj * pitch + li] : 0.;
    }
  };
  auto stash = [&](double *dst) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int lj = wv + r * NW;
      if (lj < PH) dst[lj * PW + lane] = pf[r];
    }
#pragma unroll
    for (int r = 0; r < NE; r++) {
      const int e = tid + r * FNT;
      if (e < 6 * PH) dst[(e / 6) * PW + 64 + e % 6] = pf[NR + r];
    }
  };
  // finalise layer l (additions in the order of msqg/qg.h:407-473) and store dq_l
  auto finalize = [&](int l, int gj, double t, double lapT, double zm, double zc, double zp, double tm, double tc, double tp) {
    double dq = t;
    const size_t c = nat_idx(a.g, l, gj, gi);
    double s0 = 0., s1 = 0.;
    if (nl > 1) {
      if (l > 0) s0 = a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls];
      if (l < nl - 1) s1 = a.uniformS ? a.Su[l] : a.S[c];
    }
    auto stretch = [&](double fac, double pm, double pc, double pp) -> double {
      if (l == 0) return fac * s1 * (pp - pc) * a.lc.idh1[l];
      if (l < nl - 1) return fac * (s0 * (pm - pc) * a.lc.idh0[l] + s1 * (pp - pc) * a.lc.idh1[l]);
      return fac * s0 * (pm - pc) * a.lc.idh0[l];
    };
    if (a.iRe != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe, zm, zc, zp);
      dq += tc * a.iRe;
    }
    if (a.iRe4 != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe4, tm, tc, tp);
      dq = 1. * dq + a.iRe4 * lapT;
    }
    if (l == 0) dq -= a.cs * zc;
    if (l == nl - 1) dq -= a.cb * zc;
    if (l == 0) dq -= a.wind[gj];
    if (a.have_qforc) dq += a.qforc[c];
    if (a.q_out) a.q_out[c] = a.q_in[c] + dq * a.dt;
    else a.dq[c] = dq;
  };

  // does the 2-cell halo of this block reach a wall?
  const bool bc = ((a.walls & WALL_W) && x0 - 2 < 0) || ((a.walls & WALL_E) && x0 + FTX + 2 > nx) || ((a.walls & WALL_S) && y0 - 2 < 0) ||
                  ((a.walls & WALL_N) && y0 + FTY + 2 > ny);
  fetch(0);
  stash(sP[0]);
  if (nl > 1) fetch(1);
  for (int l = 0; l < nl; l++) {
    const double *P0 = sP[l & 1], *P1 = sP[(l + 1) & 1];
    // psi_{l+1} -> LDS (its buffer was last read as P0 of layer l-1, before the closing barrier
    // of that iteration); then start the global loads of psi_{l+2}, hidden behind this layer
    if (l + 1 < nl) stash(sP[(l + 1) & 1]);
    if (l + 2 < nl && !(a.dbg & 4)) fetch(l + 2);
    __syncthreads();
    if (!(a.dbg & 1)) lds_lap<ZW, ZH, 2, PW, 3, FNT>(sZ, P0, bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);   // zeta_l
    __syncthreads();
    if (!(a.dbg & 1)) lds_lap<TW, TH, 1, ZW, 2, FNT>(sT, sZ, bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);   // tmp_l = lap(zeta_l)
    __syncthreads();
    // centre points: RPT consecutive rows per thread, 3x3 windows slide down the column
    double um = 0.;
    double p[3][3], z[3][3], p1[3][3], tcol[3];
    if (!(a.dbg & 2)) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        p[b + 1][c] = P0[(ly0 + 2 + b) * PW + (tx + 2 + c)];
        z[b + 1][c] = sZ[(ly0 + 1 + b) * ZW + (tx + 1 + c)];
        p1[b + 1][c] = P1[(ly0 + 2 + b) * PW + (tx + 2 + c)];
      }
      tcol[b + 1] = sT[(ly0 + b) * TW + (tx + 1)];
    }
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int ly = ly0 + k, gj = y0 + ly;
      const bool in = gi < nx && gj < ny;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        p[0][c] = p[1][c]; p[1][c] = p[2][c]; p[2][c] = P0[(ly + 4) * PW + (tx + 2 + c)];
        z[0][c] = z[1][c]; z[1][c] = z[2][c]; z[2][c] = sZ[(ly + 3) * ZW + (tx + 1 + c)];
        p1[0][c] = p1[1][c]; p1[1][c] = p1[2][c]; p1[2][c] = P1[(ly + 4) * PW + (tx + 2 + c)];
      }
      tcol[0] = tcol[1]; tcol[1] = tcol[2]; tcol[2] = sT[(ly + 2) * TW + (tx + 1)];
      // face velocities, msqg/qg.h:276-283 (west and south face of this cell)
      {
        const double u = fabs(DIVC(0.25 * (p[2][1] - p[0][1] + p[2][0] - p[0][0]), D, rD));
        const double v = fabs(DIVC(0.25 * (p[1][2] - p[1][0] + p[0][2] - p[0][0]), D, rD));
        if (in) um = fmax(um, fmax(u, v));
      }
      const double adv = mjac9(p, z, D12, rD12);
      const double be = DIVC(a.beta * (p[1][0] - p[1][2]), D2x, rD2x);
      const double jd = l + 1 < nl ? mjac9(p, p1, D12, rD12) : 0.;
      const double ju = -jd_prev[k];
      double t = adv + be;
      if (in && nl > 1) {
        const size_t c = nat_idx(a.g, l, gj, gi);
        if (l > 0) t = t + (a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls]) * ju * a.lc.idh0[l];
        if (l < nl - 1) t = t + (a.uniformS ? a.Su[l] : a.S[c]) * jd * a.lc.idh1[l];
      }
      t = 0. + t;  // updates were zeroed, then += (msqg/qg.h:611-613, 315)
      const double zc = z[1][1], tc = tcol[1];
      const int ct = (ly + 1) * TW + (tx + 1);
      const double lapT = DIVC(sT[ct + 1] + sT[ct - 1] + tcol[2] + tcol[0] - 4 * tc, D2, rD2);
      // layer l-1 is complete now that zeta_l, tmp_l are known
      if (in && l > 0) finalize(l - 1, gj, t_prev[k], lapT_prev[k], zc0[k], zc1[k], zc, tc0[k], tc1[k], tc);
      if (in && l == nl - 1) finalize(l, gj, t, lapT, zc1[k], zc, 0., tc1[k], tc, 0.);
      t_prev[k] = t; lapT_prev[k] = lapT; jd_prev[k] = jd;
      zc0[k] = zc1[k]; zc1[k] = zc; tc0[k] = tc1[k]; tc1[k] = tc;
    }
    }
    // per-wave maximum of |u| of this layer
    um = wave_max_f(um);
    if ((tid & 63) == 0) sM[tid >> 6][l] = um;
    __syncthreads();
  }
  if (a.umax_partial && tid < nl) {
    double v = sM[0][tid];
    for (int w = 1; w < FNT / 64; w++) v = fmax(v, sM[w][tid]);
    a.umax_partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * nl + tid] = v;
  }
}

__global__ void k_max_final2(const double *partial, double *out, int nb, int nl) {
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

// ------------------------------------------------------------------ software-pipelined variant
//
// Same arithmetic as k_rhs_fused, different schedule.  Measured on MI355X the phases of
// k_rhs_fused do not overlap (fetch 0.15 + lap passes 0.45 + centre 0.75 + barriers 0.25 ms at
// 4096^2 x 6): with one 8-wave workgroup per CU every barrier-delimited phase is either
// latency-bound (the LDS lap passes) or fp64-issue-bound (the centre loop).  Here the lap passes
// of layer l+1 run in the same barrier interval as the centre loop of layer l (double-buffered
// zeta / tmp tiles, three psi buffers), and the two waves that share a SIMD take the two jobs
// in opposite order (waves 0-3: pass first, waves 4-7: centre first), so a SIMD always has one
// wave waiting on LDS and one issuing fp64 work.  2 barriers per layer instead of 4.
template <int FTY, int FNT, bool RES>
__global__ void __launch_bounds__(FNT, 2) k_rhs_fused_pipe(RhsArgs a) {
  constexpr int RPT = FTX * FTY / FNT, HALF = RPT / 2;
  constexpr int PH = FTY + 6, ZH = FTY + 4, TH = FTY + 2;
  __shared__ double sP[3][PW * PH];
  __shared__ double sZ[2][ZW * ZH];
  __shared__ double sT[2][TW * TH];
  // q_in of the rows this wave finalises next, fetched one barrier interval ahead by LDS-DMA
  // (global_load_lds_dwordx4: no VGPR destination, 1 KB = 2 rows x 64 doubles per wave and interval): a
  // plain load of q_in at the end of the dependency chain stalled the wave for a full memory latency per
  // row (the kernel runs 2 waves per SIMD), which made the fused advance slower than a separate pass
  __shared__ __align__(16) double sQ[2][FNT / 64][2][FTX];

  const int tid = threadIdx.x, tx = tid & (FTX - 1), ly0 = (tid / FTX) * RPT;
  const int x0 = blockIdx.x * FTX, y0 = blockIdx.y * FTY;
  const int nx = a.g.nx, ny = a.g.ny, nl = a.nl, pitch = a.g.pitch;
  const double D = a.D, D2 = D * D, rD2 = 1. / D2, D12 = 12. * D * D, rD12 = 1. / D12, D2x = 2 * D, rD2x = 1. / D2x;
  const int gi = x0 + tx;
  const bool late = tid >= FNT / 2;  // second wave of each SIMD: centre first, pass second
  const bool qdma = a.q_out != nullptr && x0 + FTX <= a.g.nx && y0 + FTY <= a.g.ny;  // whole tiles only (block-uniform)

#ifdef MSOM_STRICT
  double t_prev[RPT], lapT_prev[RPT], zc0[RPT], zc1[RPT], tc0[RPT], tc1[RPT], jd_prev[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) t_prev[k] = lapT_prev[k] = zc0[k] = zc1[k] = tc0[k] = tc1[k] = jd_prev[k] = 0.;
#else
  // product build: everything of layer l that is known at iteration l is folded into ONE number per cell
  // (tl: advection + beta + cross-layer Jacobians + iRe lap(zeta) + iRe4 lap(lap(zeta)) + drag + wind), and the
  // two stretching terms share one field X = iRe zeta + iRe4 lap(zeta); 4 carried values instead of 7
  double tl_prev[RPT], xa0[RPT], xa1[RPT], jd_prev[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) tl_prev[k] = xa0[k] = xa1[k] = jd_prev[k] = 0.;
#endif
  // residual by-product: psi_{l-1}, psi_{l-2} at the cell and the two horizontal flux terms of layer l-1
  constexpr bool want_res = RES;  // compile-time: the by-product's carried values cost registers only where it is produced
#ifdef MSOM_STRICT
  // reference order ((q + A) - B) + x) + y: the four pieces are carried separately
  double pm1[RPT], pm2[RPT], xt_prev[RPT], yt_prev[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) pm1[k] = pm2[k] = xt_prev[k] = yt_prev[k] = 0.;
#else
  // product build: -(lap + Gamma) psi of layer l is complete at iteration l (psi_{l+1} is in LDS), so one
  // number per cell is carried to the iteration that finalises q_out of that layer
  double pm1[RPT], g_prev[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) pm1[k] = g_prev[k] = 0.;
#endif
  double res_m = 0., res_bs = 0.;
  const double rD1 = 1. / D;

  constexpr int NW = FNT / 64, NR = (PH + NW - 1) / NW, NE = (6 * PH + FNT - 1) / FNT;
  const int lane = tid & 63, wv = tid >> 6;
  const bool whole = x0 + FTX + 3 <= nx + 3 && y0 + FTY + 3 <= ny + 3;
  double pf[NR + NE];
  auto fetch = [&](int l) {
    const double *p = a.psi + (size_t)l * a.g.ls + (ptrdiff_t)(y0 - 3 + MSOM_YP) * pitch + (x0 - 3 + MSOM_XP);
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int lj = wv + r * NW;
      pf[r] = (lj < PH && (whole || (x0 + lane - 3 < nx + 3 && y0 + lj - 3 < ny + 3))) ? p[(ptrdiff_t)lj * pitch + lane] : 0.;
    }
#pragma unroll
    for (int r = 0; r < NE; r++) {
      const int e = tid + r * FNT, lj = e / 6, li = 64 + e % 6;
      pf[NR + r] = (e < 6 * PH && (whole || (x0 + li - 3 < nx + 3 && y0 + lj - 3 < ny + 3))) ? p[(ptrdiff_t)lj * pitch + li] : 0.;
    }
  };
  auto stash = [&](double *dst) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int lj = wv + r * NW;
      if (lj < PH) dst[lj * PW + lane] = pf[r];
    }
#pragma unroll
    for (int r = 0; r < NE; r++) {
      const int e = tid + r * FNT;
      if (e < 6 * PH) dst[(e / 6) * PW + 64 + e % 6] = pf[NR + r];
    }
  };
  // rows ly0 + 2 h, ly0 + 2 h + 1 of layer l: q_in -> sQ[h][wave]; lane L carries 16 bytes (2 columns)
  auto q_prefetch = [&](int l, int h) {
    const int ln = tid & 63, rr = ln >> 5, col = (ln & 31) * 2;
    const double *gsrc = a.q_in + nat_idx(a.g, l, y0 + ly0 + 2 * h + rr, x0 + col);
    const unsigned lds_dst =
        __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)(&sQ[h][tid >> 6][0][0]));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
  };
  // returns the residual of the cell (0 unless want_res)
  auto finalize = [&](int l, int gj, double t, double lapT, double zm, double zc, double zp, double tm, double tc, double tp, double am, double ac,
                      double ap, double xt, double yt, int qh, int qk) -> double {
    double dq = t;
    const size_t c = nat_idx(a.g, l, gj, gi);
    double s0 = 0., s1 = 0.;
    if (nl > 1) {
      if (l > 0) s0 = a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls];
      if (l < nl - 1) s1 = a.uniformS ? a.Su[l] : a.S[c];
    }
    auto stretch = [&](double fac, double pm, double pc, double pp) -> double {
      if (l == 0) return fac * s1 * (pp - pc) * a.lc.idh1[l];
      if (l < nl - 1) return fac * (s0 * (pm - pc) * a.lc.idh0[l] + s1 * (pp - pc) * a.lc.idh1[l]);
      return fac * s0 * (pm - pc) * a.lc.idh0[l];
    };
    if (a.iRe != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe, zm, zc, zp);
      dq += tc * a.iRe;
    }
    if (a.iRe4 != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe4, tm, tc, tp);
      dq = 1. * dq + a.iRe4 * lapT;
    }
    if (l == 0) dq -= a.cs * zc;
    if (l == nl - 1) dq -= a.cb * zc;
    if (l == 0) dq -= a.wind[gj];
    if (a.have_qforc) dq += a.qforc[c];
    if (!a.q_out) { a.dq[c] = dq; return 0.; }
    const double qn = (qh >= 0 ? sQ[qh][tid >> 6][qk][tx] : a.q_in[c]) + dq * a.dt;
    a.q_out[c] = qn;
    if (!want_res) return 0.;
#ifdef MSOM_STRICT
    double re = qn;
    if (nl > 1) {
      if (l == 0) re = qn + s1 * (ac - ap) * a.lc.idh1[l];
      else if (l < nl - 1) re = qn + s0 * (ac - am) * a.lc.idh0[l] - s1 * (ap - ac) * a.lc.idh1[l];
      else re = qn + s0 * (ac - am) * a.lc.idh0[l];
    }
    re += xt;
    re += yt;
#else
    const double re = qn + xt;  // xt carries the whole psi part
#endif
    a.res[split_idx(a.sg, l, gj, gi)] = re;
    res_m = fmax(res_m, fabs(re));
    res_bs += qn;
    return re;
  };
#ifndef MSOM_STRICT
  // product build: dq_l = tl + Gamma(X)_l (+ q_forc); q_out; residual = q_out + g
  auto finalize_fast = [&](int l, int gj, double tl, double xm, double xc, double xp, double g, int qh, int qk) -> double {
    double dq = tl;
    const size_t c = nat_idx(a.g, l, gj, gi);
    if (nl > 1) {
      if (l > 0) dq += (a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls]) * (xm - xc) * a.lc.idh0[l];
      if (l < nl - 1) dq += (a.uniformS ? a.Su[l] : a.S[c]) * (xp - xc) * a.lc.idh1[l];
    }
    if (a.have_qforc) dq += a.qforc[c];
    if (!a.q_out) { a.dq[c] = dq; return 0.; }
    const double qn = (qh >= 0 ? sQ[qh][tid >> 6][qk][tx] : a.q_in[c]) + dq * a.dt;
    a.q_out[c] = qn;
    if (!want_res) return 0.;
    const double re = qn + g;
    a.res[split_idx(a.sg, l, gj, gi)] = re;
    res_m = fmax(res_m, fabs(re));
    res_bs += qn;
    return re;
  };
#endif
  // centre rows [k0, k0 + HALF) of layer l
  auto centre = [&](int l, int k0, const double *P0, const double *P1, const double *Z, const double *T) {
    double p[3][3], z[3][3], p1[3][3], tcol[3];
    double ra_even = 0., rb_even = 0.;
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        p[b + 1][c] = P0[(ly0 + k0 + 2 + b) * PW + (tx + 2 + c)];
        z[b + 1][c] = Z[(ly0 + k0 + 1 + b) * ZW + (tx + 1 + c)];
        p1[b + 1][c] = P1[(ly0 + k0 + 2 + b) * PW + (tx + 2 + c)];
      }
      tcol[b + 1] = T[(ly0 + k0 + b) * TW + (tx + 1)];
    }
#pragma unroll
    for (int kk = 0; kk < HALF; kk++) {
      const int k = k0 + kk, ly = ly0 + k, gj = y0 + ly;
      const bool in = gi < nx && gj < ny;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        p[0][c] = p[1][c]; p[1][c] = p[2][c]; p[2][c] = P0[(ly + 4) * PW + (tx + 2 + c)];
        z[0][c] = z[1][c]; z[1][c] = z[2][c]; z[2][c] = Z[(ly + 3) * ZW + (tx + 1 + c)];
        p1[0][c] = p1[1][c]; p1[1][c] = p1[2][c]; p1[2][c] = P1[(ly + 4) * PW + (tx + 2 + c)];
      }
      tcol[0] = tcol[1]; tcol[1] = tcol[2]; tcol[2] = T[(ly + 2) * TW + (tx + 1)];
      const double adv = mjac9(p, z, D12, rD12);
      const double be = DIVC(a.beta * (p[1][0] - p[1][2]), D2x, rD2x);
      const double jd = l + 1 < nl ? mjac9(p, p1, D12, rD12) : 0.;
      const double ju = -jd_prev[k];
      double t = adv + be;
      if (in && nl > 1) {
        const size_t c = nat_idx(a.g, l, gj, gi);
        if (l > 0) t = t + (a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls]) * ju * a.lc.idh0[l];
        if (l < nl - 1) t = t + (a.uniformS ? a.Su[l] : a.S[c]) * jd * a.lc.idh1[l];
      }
      t = 0. + t;
      const double zc = z[1][1], tc = tcol[1];
      const int ct = (ly + 1) * TW + (tx + 1);
      const double lapT = DIVC(T[ct + 1] + T[ct - 1] + tcol[2] + tcol[0] - 4 * tc, D2, rD2);
      // horizontal part of the residual of layer l at this cell: ((c - W)/D - (E - c)/D)/D, then y
      const double pc = p[1][1];
      double xt = 0., yt = 0.;
      if (want_res) {
        xt = DIVC(DIVC(pc - p[1][0], D, rD1) - DIVC(p[1][2] - pc, D, rD1), D, rD1);
        yt = DIVC(DIVC(pc - p[0][1], D, rD1) - DIVC(p[2][1] - pc, D, rD1), D, rD1);
      }
      double ra = 0., rb = 0.;
#ifdef MSOM_STRICT
      if (in && l > 0) ra = finalize(l - 1, gj, t_prev[k], lapT_prev[k], zc0[k], zc1[k], zc, tc0[k], tc1[k], tc, pm2[k], pm1[k], pc, xt_prev[k], yt_prev[k], qdma ? k0 / HALF : -1, kk);
      if (in && l == nl - 1) rb = finalize(l, gj, t, lapT, zc1[k], zc, 0., tc1[k], tc, 0., pm1[k], pc, 0., xt, yt, -1, 0);
#else
      double gl = 0.;
      if (want_res) {  // -(lap + Gamma) psi_l at this cell
        gl = xt + yt;
        if (in && nl > 1) {
          const size_t c = nat_idx(a.g, l, gj, gi);
          if (l > 0) gl += (a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls]) * (pc - pm1[k]) * a.lc.idh0[l];
          if (l < nl - 1) gl -= (a.uniformS ? a.Su[l] : a.S[c]) * (p1[1][1] - pc) * a.lc.idh1[l];
        }
      }
      const double xl = a.iRe * zc + a.iRe4 * tc;
      double tl = t + a.iRe4 * lapT + tc * a.iRe;
      if (l == 0) tl -= a.cs * zc + (in ? a.wind[gj] : 0.);
      if (l == nl - 1) tl -= a.cb * zc;
      if (in && l > 0) ra = finalize_fast(l - 1, gj, tl_prev[k], xa0[k], xa1[k], xl, g_prev[k], qdma ? k0 / HALF : -1, kk);
      if (in && l == nl - 1) rb = finalize_fast(l, gj, tl, xa1[k], xl, 0., gl, -1, 0);
      tl_prev[k] = tl; xa0[k] = xa1[k]; xa1[k] = xl; jd_prev[k] = jd;
#endif
#ifdef MSOM_STRICT
      t_prev[k] = t; lapT_prev[k] = lapT; jd_prev[k] = jd;
      zc0[k] = zc1[k]; zc1[k] = zc; tc0[k] = tc1[k]; tc1[k] = tc;
#endif
      if (want_res) {
#ifdef MSOM_STRICT
        pm2[k] = pm1[k]; pm1[k] = pc; xt_prev[k] = xt; yt_prev[k] = yt;
#else
        pm1[k] = pc; g_prev[k] = gl;
#endif
        // restriction to level 1: mean of the 4 children in foreach_child order (rows k, k+1 of this
        // thread are the y pair, lanes tx, tx+1 the x pair); every lane takes part in the shuffles
        if (a.res_c) {
          if ((kk & 1) == 0) { ra_even = ra; rb_even = rb; }
          else {
            const double oa0 = lane_next(ra_even), oa1 = lane_next(ra);
            const double ob0 = lane_next(rb_even), ob1 = lane_next(rb);
            if (in && !(tx & 1)) {
              if (l > 0) {
                double sum = 0.;
                sum += ra_even; sum += ra; sum += oa0; sum += oa1;
                a.res_c[split_idx(a.cg, l - 1, gj >> 1, gi >> 1)] = sum / 4;
              }
              if (l == nl - 1) {
                double sum = 0.;
                sum += rb_even; sum += rb; sum += ob0; sum += ob1;
                a.res_c[split_idx(a.cg, l, gj >> 1, gi >> 1)] = sum / 4;
              }
            }
          }
        }
      }
    }
  };

  const bool bc = ((a.walls & WALL_W) && x0 - 2 < 0) || ((a.walls & WALL_E) && x0 + FTX + 2 > nx) || ((a.walls & WALL_S) && y0 - 2 < 0) ||
                  ((a.walls & WALL_N) && y0 + FTY + 2 > ny);
  // prologue: psi_0, psi_1 in LDS, psi_2 in flight, zeta_0 and tmp_0 ready
  fetch(0);
  stash(sP[0]);
  if (nl > 1) { fetch(1); stash(sP[1]); }
  __syncthreads();
  lds_lap<ZW, ZH, 2, PW, 3, FNT>(sZ[0], sP[0], bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
  __syncthreads();
  lds_lap<TW, TH, 1, ZW, 2, FNT>(sT[0], sZ[0], bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
  __syncthreads();
  for (int l = 0; l < nl; l++) {
    const double *P0 = sP[l % 3], *P1 = sP[(l + 1) % 3];
    const double *Z = sZ[l & 1], *T = sT[l & 1];
    const bool more = l + 1 < nl;
    // the q_in rows of layer l-1 fetched during the previous interval have landed (own wave's DMA)
    if (qdma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (qdma && l > 0) q_prefetch(l - 1, 1);  // second-half rows of layer l-1: finalised in interval 2
    // interval 1: zeta_{l+1} pass || first half of the centre rows of layer l.  psi_{l+2} travels global ->
    // registers -> LDS (buffer of psi_{l-1}, free since the last barrier) around the LDS pass, whose duration
    // hides the load latency; the staging registers are dead during the register-hungry centre loop
    if (!late) {
      if (l + 2 < nl) fetch(l + 2);
      if (more) lds_lap<ZW, ZH, 2, PW, 3, FNT>(sZ[(l + 1) & 1], P1, bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
      if (l + 2 < nl) stash(sP[(l + 2) % 3]);
    }
    centre(l, 0, P0, P1, Z, T);
    if (late) {
      if (l + 2 < nl) fetch(l + 2);
      if (more) lds_lap<ZW, ZH, 2, PW, 3, FNT>(sZ[(l + 1) & 1], P1, bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
      if (l + 2 < nl) stash(sP[(l + 2) % 3]);
    }
    __syncthreads();
    if (qdma) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (more) q_prefetch(l, 0);  // first-half rows of layer l: finalised in interval 1 of the next iteration
    }
    // interval 2: tmp_{l+1} pass || second half of the centre rows
    if (!late && more) lds_lap<TW, TH, 1, ZW, 2, FNT>(sT[(l + 1) & 1], sZ[(l + 1) & 1], bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
    centre(l, HALF, P0, P1, Z, T);
    if (late && more) lds_lap<TW, TH, 1, ZW, 2, FNT>(sT[(l + 1) & 1], sZ[(l + 1) & 1], bc, x0, y0, nx, ny, a.walls, a.slip_c, D2, rD2);
    __syncthreads();
  }
  if (want_res) {  // max|res| -> atomic max, sum of q_out -> one partial per workgroup (fixed order)
    double *red = sT[0];  // free after the last barrier
    const double wm = wave_max_f(res_m);
    double ws = res_bs;
    for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o, 64);
    if (lane == 0) { red[wv] = wm; red[NW + wv] = ws; }
    __syncthreads();
    if (tid == 0) {
      double mm = red[0], ss = red[NW];
      for (int k = 1; k < NW; k++) { mm = fmax(mm, red[k]); ss += red[NW + k]; }
      atomicMax((unsigned long long *)a.res_max, (unsigned long long)__double_as_longlong(mm));
      a.bsum_partial[blockIdx.y * gridDim.x + blockIdx.x] = ss;
    }
  }
}

int g_rhs_dbg = 0;
int rhs_pipe_blocks(const NatGeom &g) { return ((g.nx + FTX - 1) / FTX) * ((g.ny + 31) / 32); }
int rhs_fused_blocks(const NatGeom &g) { return ((g.nx + FTX - 1) / FTX) * ((g.ny + 7) / 8); }  // upper bound (smallest FTY)

void launch_rhs_fused(hipStream_t st, const double *psi, const double *S, const double *qforc, const double *wind, double *dq,
                      double *umax_partial, double *umax_out, const NatGeom &g, int nl, int walls, int uniformS, const double *Su,
                      int have_qforc, double D, double beta, double iRe, double iRe4, double cs, double cb, double slip_c,
                      const LayerCoef &lc, int variant, const double *q_in, double *q_out, double dt, const RhsResid *rr, int region, const double *dt_ptr) {
  extern int g_rhs_dbg;
  if (variant == 6) {
    launch_rhs_lpw(st, psi, S, qforc, wind, dq, g, nl, walls, uniformS, Su, have_qforc, D, beta, iRe, iRe4, cs, cb, slip_c, lc, q_in, q_out, dt,
                   g_rhs_dbg >> 8, 0, nullptr, nullptr, 0., 0., region, dt_ptr);  // tuning: rhs_dbg = rows << 8 overrides the chunk height
    return;
  }
  RhsArgs a;
  a.dbg = g_rhs_dbg;
  a.res = a.res_c = a.res_max = a.bsum_partial = nullptr;
  if (rr && variant == 1 && q_out) {
    a.res = rr->res; a.res_c = rr->res_c; a.res_max = rr->res_max; a.bsum_partial = rr->bsum_partial; a.sg = rr->sg; a.cg = rr->cg;
    (void)hipMemsetAsync(rr->res_max, 0, sizeof(double), st);
  } else {
    a.sg = SplitGeom(); a.cg = SplitGeom();
  }
  a.q_in = q_in; a.q_out = q_out; a.dt = dt;
  a.psi = psi; a.S = S; a.qforc = qforc; a.wind = wind; a.dq = dq; a.umax_partial = umax_partial;
  a.g = g; a.nl = nl; a.walls = walls; a.uniformS = uniformS; a.have_qforc = have_qforc;
  a.D = D; a.beta = beta; a.iRe = iRe; a.iRe4 = iRe4; a.cs = cs; a.cb = cb; a.slip_c = slip_c; a.lc = lc;
  for (int l = 0; l < MSOM_MAXNL; l++) a.Su[l] = Su ? Su[l] : 0.;
  int fty = (variant == 2 || variant == 4) ? 16 : (variant == 3 || variant == 5) ? 8 : 32;
  dim3 gr((g.nx + FTX - 1) / FTX, (g.ny + fty - 1) / fty);
  switch (variant) {
    case 1:
      if (a.res) hipLaunchKernelGGL((k_rhs_fused_pipe<32, 512, true>), gr, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((k_rhs_fused_pipe<32, 512, false>), gr, dim3(512), 0, st, a);
      break;
    case 2: hipLaunchKernelGGL((k_rhs_fused<16, 512, 2>), gr, dim3(512), 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_rhs_fused<8, 256, 2>), gr, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_rhs_fused<16, 512, 4>), gr, dim3(512), 0, st, a); break;
    case 5: hipLaunchKernelGGL((k_rhs_fused<8, 256, 4>), gr, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_rhs_fused<32, 512, 2>), gr, dim3(512), 0, st, a); break;
  }
  if (umax_partial) {
    (void)hipMemsetAsync(umax_out, 0, nl * sizeof(double), st);
    hipLaunchKernelGGL(k_max_final2, dim3(64), dim3(256), 0, st, umax_partial, umax_out, (int)(gr.x * gr.y), nl);
  }
}
