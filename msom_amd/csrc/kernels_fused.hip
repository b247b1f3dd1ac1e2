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
#include "kernels.h"

#ifdef MSOM_STRICT
#define DIVC(x, c, rc) ((x) / (c))
#else
#define DIVC(x, c, rc) ((x) * (rc))
#endif

#define FTX 64
#define FTY 16
#define FNT 256
#define NPT (FTX * FTY / FNT)
#define PW (FTX + 6)
#define PH (FTY + 6)
#define ZW (FTX + 4)
#define ZH (FTY + 4)
#define TW (FTX + 2)
#define TH (FTY + 2)

struct RhsArgs {
  const double *psi, *S, *qforc, *wind;
  double *dq, *umax_partial;
  NatGeom g;
  int nl, walls, uniformS, have_qforc;
  double D, beta, iRe, iRe4, cs, cb, slip_c;
  LayerCoef lc;
  double Su[MSOM_MAXNL];
};

__device__ __forceinline__ double wave_max_f(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// -J(p,q), msqg/qg.h:252-262, from 3x3 register windows [dy+1][dx+1]
__device__ __forceinline__ double mjac9(const double (&p)[3][3], const double (&q)[3][3], double D12, double rD12) {
#define P(a, b) p[(b) + 1][(a) + 1]
#define Q(a, b) q[(b) + 1][(a) + 1]
  const double s = (Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0)) +
                   Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1)) - Q(0, 1) * (P(1, 1) - P(-1, 1)) +
                   Q(0, -1) * (P(1, -1) - P(-1, -1)) + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1)) -
                   P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1));
#undef P
#undef Q
  return DIVC(s, D12, rD12);
}

// Dirichlet ghost fill of an LDS tile `t` (width W, height H, halo h around the FTX x FTY
// block at (x0, y0)): every position that lies outside a wall gets sign * t[mirror];
// with partial slip the edge ghosts are c * (src[mirror] - src[ghost]) instead, where src is
// the field the Laplacian was taken of (halo hs, width SW).
template <int W, int H, int h, int SW, int hs>
__device__ __forceinline__ void lds_wall_ghosts(double *t, const double *src, int x0, int y0, int nx, int ny, int walls, double slip_c) {
  for (int idx = threadIdx.x; idx < W * H; idx += FNT) {
    const int li = idx % W, lj = idx / W;
    const int gi = x0 + li - h, gj = y0 + lj - h;
    const bool ox = (gi < 0 && (walls & WALL_W)) || (gi >= nx && (walls & WALL_E));
    const bool oy = (gj < 0 && (walls & WALL_S)) || (gj >= ny && (walls & WALL_N));
    if (!(ox | oy)) continue;
    const int mi = ox ? (gi < 0 ? -1 - gi : 2 * nx - 1 - gi) : gi;
    const int mj = oy ? (gj < 0 ? -1 - gj : 2 * ny - 1 - gj) : gj;
    const int ti = mi - x0 + h, tj = mj - y0 + h;
    double v = 0.;
    if (ti >= 0 && ti < W && tj >= 0 && tj < H) {
      v = t[tj * W + ti];
      if (ox != oy) {
        const bool depth1 = ox ? (gi == -1 || gi == nx) : (gj == -1 || gj == ny);
        if (slip_c > 0. && depth1) v = slip_c * (src[(mj - y0 + hs) * SW + (mi - x0 + hs)] - src[(gj - y0 + hs) * SW + (gi - x0 + hs)]);
        else v = -v;
      }
    }
    t[idx] = v;
  }
}

__global__ void __launch_bounds__(FNT) k_rhs_fused(RhsArgs a) {
  __shared__ double sP[2][PW * PH];
  __shared__ double sZ[ZW * ZH];
  __shared__ double sT[TW * TH];
  __shared__ double sM[FNT / 64][MSOM_MAXNL];

  const int tid = threadIdx.x, tx = tid & (FTX - 1), ty0 = tid / FTX;
  const int x0 = blockIdx.x * FTX, y0 = blockIdx.y * FTY;
  const int nx = a.g.nx, ny = a.g.ny, nl = a.nl, pitch = a.g.pitch;
  const double D = a.D, D2 = D * D, rD2 = 1. / D2, D12 = 12. * D * D, rD12 = 1. / D12, D2x = 2 * D, rD2x = 1. / D2x, rD = 1. / D;

  // per-point registers: values of the previous two layers needed to finalise a layer
  double t_prev[NPT], lapT_prev[NPT], zc0[NPT], zc1[NPT], tc0[NPT], tc1[NPT], jd_prev[NPT];
#pragma unroll
  for (int k = 0; k < NPT; k++) t_prev[k] = lapT_prev[k] = zc0[k] = zc1[k] = tc0[k] = tc1[k] = jd_prev[k] = 0.;

  auto load_psi = [&](double *dst, int l) {
    const double *p = a.psi + (size_t)l * a.g.ls;
    for (int idx = tid; idx < PW * PH; idx += FNT) {
      const int li = idx % PW, lj = idx / PW;
      const int gi = x0 + li - 3, gj = y0 + lj - 3;
      dst[idx] = (gi >= -3 && gi < nx + 3 && gj >= -3 && gj < ny + 3) ? p[(ptrdiff_t)(gj + MSOM_YP) * pitch + (gi + MSOM_XP)] : 0.;
    }
  };
  // finalise layer l (additions in the order of msqg/qg.h:407-473) and store dq_l
  auto finalize = [&](int l, int k, int gi, int gj, double t, double lapT, double zm, double zc, double zp, double tm, double tc, double tp) {
    double dq = t;
    const size_t c = nat_idx(a.g, l, gj, gi);
    double s0 = 0., s1 = 0.;
    if (nl > 1) {
      if (l > 0) s0 = a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls];
      if (l < nl - 1) s1 = a.uniformS ? a.Su[l] : a.S[c];
    }
    auto stretch = [&](double fac, double pm, double pc, double pp) -> double {
      if (nl == 1) return 0.;
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
    a.dq[c] = dq;
  };

  load_psi(sP[0], 0);
  for (int l = 0; l < nl; l++) {
    double *P0 = sP[l & 1], *P1 = sP[(l + 1) & 1];
    if (l + 1 < nl) load_psi(P1, l + 1);
    __syncthreads();
    // zeta_l on the 2-cell halo
    for (int idx = tid; idx < ZW * ZH; idx += FNT) {
      const int li = idx % ZW, lj = idx / ZW;
      const int c = (lj + 1) * PW + (li + 1);
      sZ[idx] = DIVC(P0[c + 1] + P0[c - 1] + P0[c + PW] + P0[c - PW] - 4 * P0[c], D2, rD2);
    }
    __syncthreads();
    if (a.walls) lds_wall_ghosts<ZW, ZH, 2, PW, 3>(sZ, P0, x0, y0, nx, ny, a.walls, a.slip_c);
    __syncthreads();
    // tmp_l = lap(zeta_l) on the 1-cell halo
    for (int idx = tid; idx < TW * TH; idx += FNT) {
      const int li = idx % TW, lj = idx / TW;
      const int c = (lj + 1) * ZW + (li + 1);
      sT[idx] = DIVC(sZ[c + 1] + sZ[c - 1] + sZ[c + ZW] + sZ[c - ZW] - 4 * sZ[c], D2, rD2);
    }
    __syncthreads();
    if (a.walls) lds_wall_ghosts<TW, TH, 1, ZW, 2>(sT, sZ, x0, y0, nx, ny, a.walls, a.slip_c);
    __syncthreads();
    // centre points
    double um = 0.;
#pragma unroll
    for (int k = 0; k < NPT; k++) {
      const int ly = ty0 + k * (FNT / FTX);
      const int gi = x0 + tx, gj = y0 + ly;
      const bool in = gi < nx && gj < ny;
      double p[3][3], q[3][3];
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          p[b][c] = P0[(ly + 2 + b) * PW + (tx + 2 + c)];
          q[b][c] = sZ[(ly + 1 + b) * ZW + (tx + 1 + c)];
        }
      // face velocities, msqg/qg.h:276-283 (west and south face of this cell)
      {
        const double u = fabs(DIVC(0.25 * (p[2][1] - p[0][1] + p[2][0] - p[0][0]), D, rD));
        const double v = fabs(DIVC(0.25 * (p[1][2] - p[1][0] + p[0][2] - p[0][0]), D, rD));
        if (in) um = fmax(um, fmax(u, v));
      }
      const double adv = mjac9(p, q, D12, rD12);
      const double be = DIVC(a.beta * (p[1][0] - p[1][2]), D2x, rD2x);
      double jd = 0.;
      if (l + 1 < nl) {
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
          for (int c = 0; c < 3; c++) q[b][c] = P1[(ly + 2 + b) * PW + (tx + 2 + c)];
        jd = mjac9(p, q, D12, rD12);
      }
      const double ju = -jd_prev[k];
      double t = adv + be;
      if (in && nl > 1) {
        const size_t c = nat_idx(a.g, l, gj, gi);
        if (l > 0) t = t + (a.uniformS ? a.Su[l - 1] : a.S[c - a.g.ls]) * ju * a.lc.idh0[l];
        if (l < nl - 1) t = t + (a.uniformS ? a.Su[l] : a.S[c]) * jd * a.lc.idh1[l];
      }
      const double zc = sZ[(ly + 2) * ZW + (tx + 2)];
      const int ct = (ly + 1) * TW + (tx + 1);
      const double tc = sT[ct];
      const double lapT = DIVC(sT[ct + 1] + sT[ct - 1] + sT[ct + TW] + sT[ct - TW] - 4 * tc, D2, rD2);
      // layer l-1 is complete now that zeta_l, tmp_l are known
      if (in && l > 0) finalize(l - 1, k, gi, gj, t_prev[k], lapT_prev[k], zc0[k], zc1[k], zc, tc0[k], tc1[k], tc);
      if (in && l == nl - 1) finalize(l, k, gi, gj, 0. + t, lapT, zc1[k], zc, 0., tc1[k], tc, 0.);
      t_prev[k] = 0. + t; lapT_prev[k] = lapT; jd_prev[k] = jd;
      zc0[k] = zc1[k]; zc1[k] = zc; tc0[k] = tc1[k]; tc1[k] = tc;
    }
    // per-wave maximum of |u| of this layer
    um = wave_max_f(um);
    if ((tid & 63) == 0) sM[tid >> 6][l] = um;
    __syncthreads();
  }
  if (tid < nl) {
    double v = sM[0][tid];
    for (int w = 1; w < FNT / 64; w++) v = fmax(v, sM[w][tid]);
    a.umax_partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * nl + tid] = v;
  }
}

__global__ void k_max_final2(const double *partial, double *out, int nb, int nl) {
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

int rhs_fused_blocks(const NatGeom &g) { return ((g.nx + FTX - 1) / FTX) * ((g.ny + FTY - 1) / FTY); }

void launch_rhs_fused(hipStream_t st, const double *psi, const double *S, const double *qforc, const double *wind, double *dq,
                      double *umax_partial, double *umax_out, const NatGeom &g, int nl, int walls, int uniformS, const double *Su,
                      int have_qforc, double D, double beta, double iRe, double iRe4, double cs, double cb, double slip_c,
                      const LayerCoef &lc) {
  RhsArgs a;
  a.psi = psi; a.S = S; a.qforc = qforc; a.wind = wind; a.dq = dq; a.umax_partial = umax_partial;
  a.g = g; a.nl = nl; a.walls = walls; a.uniformS = uniformS; a.have_qforc = have_qforc;
  a.D = D; a.beta = beta; a.iRe = iRe; a.iRe4 = iRe4; a.cs = cs; a.cb = cb; a.slip_c = slip_c; a.lc = lc;
  for (int l = 0; l < MSOM_MAXNL; l++) a.Su[l] = Su ? Su[l] : 0.;
  dim3 gr((g.nx + FTX - 1) / FTX, (g.ny + FTY - 1) / FTY);
  hipLaunchKernelGGL(k_rhs_fused, gr, dim3(FNT), 0, st, a);
  hipLaunchKernelGGL(k_max_final2, dim3(nl), dim3(256), 0, st, umax_partial, umax_out, (int)(gr.x * gr.y), nl);
}
