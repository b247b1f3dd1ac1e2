// kernels_lpw.hip -- the PV tendency dq/dt = F(psi) (+ the fused advance), "one layer per wavefront" schedule.
//
// Same arithmetic as kernels_fused.hip (comp_del2 -> advection_pv -> dissip -> ekman_friction -> surface_forcing ->
// qforcing -> advance_qg, msqg/qg.h:172-246,288-380,407-474,594-606), different mapping.  PMC counters of
// k_rhs_fused_pipe at 4096^2 x 6 (profiles/r01_pmc_sq_rhs.json): ~200 vector instructions per cell and layer, waves
// parked on s_waitcnt / s_barrier 58 % of their life (155 KB of LDS tiles => 2 waves per SIMD).  Here:
//
//  * a workgroup is nl wavefronts; wavefront l owns layer l of a strip of 64 columns (58 of them produce output, 3 on
//    each side are the halo) and marches up the rows of a chunk;
//  * zeta = lap(psi), tmp = lap(zeta), lap(tmp) and the two Arakawa Jacobians come from sliding REGISTER windows:
//    a lane keeps its own column of psi (5 rows), zeta (4), tmp (3) and psi_{l+1} (3); the x +- 1 neighbours are
//    fetched from the adjacent lanes by whole-wave DPP shifts (no LDS tiles, no barriers for the stencils);
//  * wall ghosts of zeta / tmp (Dirichlet mirror, partial slip, corners; msqg/qg.h:185-198) are produced in the lane /
//    row that holds the ghost position, from the neighbouring lane (x walls) or the neighbouring row (y walls);
//  * the only cross-wave traffic is the vertical coupling: every wave publishes 2 numbers per cell (3 in the validation
//    build) in a double-buffered LDS ring and finalises its rows one barrier interval later (1 barrier per LPW_R rows);
//  * psi rows, psi_{l+1} rows and q_in are prefetched one interval ahead into registers.
#include "rhs_inl.h"

// Product build: no automatic contraction in this file -- every fused multiply-add below is written out.  The kernel exists
// in two instantiations (with / without the ghost-line code) and a cell is computed by either, depending on the tiling; left
// to the compiler, contraction follows the basic-block structure, which differs between the two (one cell in 2 x 10^6
// differed by an ulp after three steps of a 2 x 2 tiling, tests/test_gpu_tiled.py).  The validation build never contracts.
#pragma clang fp contract(off)

#define LPW_W 58  // output columns of a strip (64 lanes - 2 x 3 halo)
#define LPW_R 4   // rows per barrier interval
// A workgroup is floor(LPW_MAXW / nl) strips x nl layers.  12 wavefronts = 3 per SIMD is what the register budget of the
// product build allows (<= 168 VGPRs); a 6-wave workgroup (one strip of 6 layers) left every CU with ONE resident
// workgroup (2 + 2 + 1 + 1 waves on the SIMDs: the second one did not fit) -- measured 6 waves per CU
#ifdef MSOM_STRICT
#define LPW_NV 3  // zeta, tmp, jd
#define LPW_MAXW 8
#else
#define LPW_NV 2  // X = iRe zeta + iRe4 tmp, jd
#define LPW_MAXW 12
#endif

struct LpwArgs {
  const double *psi, *S, *qforc, *wind, *q_in;
  const double *q_stage, *noise;  // STOCH: q_out = q_in + crelax * q_stage + dts * noise + dt * dq (msqg/qg_stochastic.h:48-63, 139-147)
  double crelax, dts;
  double *dq, *q_out;  // q_out != 0: q_out = q_in + dt * dq (msqg/qg.h:602), dq not stored
  double dt;
  const double *dt_ptr;  // non-null: dt is read from device memory (written by k_step_dt while the host has not seen max|u| yet)
  NatGeom g;
  int nl, walls, uniformS, have_qforc, H, NS;  // NS strips per workgroup
  int noedge_off;  // every wavefront takes the EDGE instantiation (cross-check)
  int region;  // tiles, overlap with the psi halo exchange: 0 all, 1 only the wavefronts that read no halo cell, 2 only the others
  int dbg;  // timing experiments only (results wrong): 1 = no stores, 2 = every load hits the chunk's first row (no HBM reads)
  double D, beta, iRe, iRe4, cs, cb, slip_c;
  LayerCoef lc;
  double Su[MSOM_MAXNL];
};

// UNI: uniform S (constants), QF: 3-D forcing present, ADV: advance fused.  Compile-time so that the unrolled row body
// is straight-line code: s_waitcnt counters stay exact and a wave never waits for a prefetch it does not need yet
// STOCH (msqg/qg_stochastic.h:36-63): the top layer drops J(psi, zeta), no layer has the interface Jacobian
// EDGE (round 3): a wavefront whose strip touches no x wall and whose chunk touches no y wall takes the instantiation
// without the ghost-line code (wave-uniform choice at kernel entry; same arithmetic for every cell it owns).  The wall
// tests were two branches per Laplacian and four per row in EVERY row of EVERY wavefront: 0.60 -> 0.55 ms per launch at
// 4096^2 x 6 with the tests compiled out everywhere (tools/ab_prof.py lpw_dbg), of a kernel that is issue-bound (0.545 ms
// with no memory traffic at all)
template <int R, bool UNI, bool QF, bool ADV, bool STOCH, bool EDGE>
__device__ __forceinline__ void lpw_body(const LpwArgs &a, double (&ring)[2][LPW_MAXW][R][LPW_NV][64]) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // ring slot; wv +- 1 = neighbouring layers of the strip
  const int nl = a.nl, nx = a.g.nx, ny = a.g.ny;
  const int l = wv % nl;
  const ptrdiff_t pitch = a.g.pitch;
  // strips past the right edge (last workgroup of a row) repeat the last strip and store nothing
  // (the XCD-contiguous block numbering of rhs_inl.h cuts this kernel's HBM reads from 2.05 to 1.70 GB but makes it
  // 6 % SLOWER: it is not bandwidth-bound, and neighbouring strips marching in step crowd the same channels)
  const int strip = blockIdx.x * a.NS + wv / nl, nstrips = (nx + LPW_W - 1) / LPW_W;
  const int x0 = min(strip, nstrips - 1) * LPW_W, y0 = blockIdx.y * a.H, y1 = min(ny, y0 + a.H);
  const int gi = x0 - 3 + lane, gic = min(gi, nx + 2);  // lanes past the padded row re-read its last column (never stored)
  const double dtv = a.dt_ptr ? *a.dt_ptr : a.dt;
  const double D = a.D, D2 = D * D, rD2 = 1. / D2, D12 = 12. * D * D, rD12 = 1. / D12, D2x = 2 * D, rD2x = 1. / D2x;
  const bool lower = l + 1 < nl, upper = l > 0;
  // lanes / rows that hold the first ghost line of a wall
  const int lW = ((a.walls & WALL_W) && x0 == 0) ? 2 : -1;
  const int eL = nx - x0 + 3;
  const int lE = ((a.walls & WALL_E) && eL <= 63) ? eL : -1;
  const bool bcx = EDGE && (lW >= 0 || lE >= 0);
  const bool south = (a.walls & WALL_S) != 0, north = (a.walls & WALL_N) != 0;
  const bool slip = a.slip_c > 0.;
  const bool out_ok = lane >= 3 && lane <= 60 && gi < nx && strip < nstrips;
  const double *pP = a.psi + nat_idx(a.g, l, 0, gic);
  // psi of the layer below; the bottom layer re-reads its own rows (value unused) so that every wave issues the same
  // sequence of loads and the s_waitcnt counters are exact
  const double *pQ = lower ? pP + a.g.ls : pP;
  // per-layer constants of this wavefront (scalar registers for the whole kernel)
  const double idh0 = a.lc.idh0[l], idh1 = a.lc.idh1[l];
  const double su0 = (nl > 1 && upper) ? a.Su[l - 1] : 0., su1 = (nl > 1 && lower) ? a.Su[l] : 0.;
  // wind stress curl of the chunk's rows (layer 0): lane k holds row y0 + k (H <= 64), read back with v_readlane
  const double windv = l == 0 ? a.wind[min(y0 + lane, ny - 1)] : 0.;
  auto wind_row = [&](int j) -> double {
    const long long b = __double_as_longlong(windv);
    const int lo = __builtin_amdgcn_readlane((int)b, j - y0), hi = __builtin_amdgcn_readlane((int)(b >> 32), j - y0);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };

  // register windows; index k of P, Z, T, Q = row j - 1 + k of the current output row j
  double P[5], PL[4], PR[4], Z[4], ZL[4], ZR[4], T[3], TL[2], TR[2], Q[3], QL[3], QR[3];
#pragma unroll
  for (int k = 0; k < 5; k++) P[k] = 0.;
#pragma unroll
  for (int k = 0; k < 4; k++) PL[k] = PR[k] = Z[k] = ZL[k] = ZR[k] = 0.;
#pragma unroll
  for (int k = 0; k < 3; k++) T[k] = Q[k] = QL[k] = QR[k] = 0.;
  TL[0] = TL[1] = TR[0] = TR[1] = 0.;

  // values of a row kept for the finalisation one interval later
#ifdef MSOM_STRICT
  double abA[R], jdA[R], lapTA[R], zcA[R], tcA[R];
#else
  double tlA[R];
#endif
  double pnext[R], qnext[R], qreg[R], fqreg[R], qsreg[STOCH ? R : 1], nzreg[STOCH ? R : 1];
#pragma unroll
  for (int r = 0; r < R; r++) {
#ifdef MSOM_STRICT
    abA[r] = jdA[r] = lapTA[r] = zcA[r] = tcA[r] = 0.;
#else
    tlA[r] = 0.;
#endif
    pnext[r] = qnext[r] = qreg[r] = fqreg[r] = 0.;
  }

  // EDGE = false: every row the wavefront touches, the last interval's prefetch included, lies inside the padded layer: no clamps,
  // so that the row offsets are plain induction variables
  auto ld = [&](const double *base, int j) -> double { return base[(ptrdiff_t)((a.dbg & 2) ? y0 : (EDGE ? min(j, ny + 2) : j)) * pitch]; };
#ifdef MSOM_STRICT
  auto lap5 = [&](double c, double w, double e, double n, double s) -> double { return DIVC(e + w + n + s - 4 * c, D2, rD2); };
#else
  auto lap5 = [&](double c, double w, double e, double n, double s) -> double { return fma(-4., c, e + w + n + s) * rD2; };
#endif
  // x walls: the ghost lane takes -dst(mirror lane) or the partial-slip value c (src(mirror) - src(ghost))
  auto xfix = [&](double raw, double so, double sw, double se) -> double {
    if (bcx) {
      if (slip) {
        if (lane == lW) raw = a.slip_c * (se - so);
        if (lane == lE) raw = a.slip_c * (sw - so);
      } else {
        const double re = lane_above(raw), rw = lane_below(raw);
        if (lane == lW) raw = -re;
        if (lane == lE) raw = -rw;
      }
    }
    return raw;
  };
  // y walls: the ghost row from its mirror row (dm with neighbours dmw, dme); corners take +dst(mirror, mirror)
  auto yghost = [&](double dm, double dmw, double dme, double sm, double sg) -> double {
    double v = slip ? a.slip_c * (sm - sg) : -dm;
    if (lane == lW) v = dme;
    if (lane == lE) v = dmw;
    return v;
  };

  // one marching step: psi row j + 3 and psi_{l+1} row j + 1 enter, zeta row j + 2 and tmp row j + 1 are built, the
  // centre terms of row j are computed and published (slot r of ring buffer b)
  auto row = [&](int j, double pnew, double qnew, bool centre, bool yedge, int b, int r) {
#pragma unroll
    for (int k = 0; k < 4; k++) P[k] = P[k + 1];
    P[4] = pnew;
#pragma unroll
    for (int k = 0; k < 3; k++) { PL[k] = PL[k + 1]; PR[k] = PR[k + 1]; Z[k] = Z[k + 1]; ZL[k] = ZL[k + 1]; ZR[k] = ZR[k + 1]; }
    T[0] = T[1]; T[1] = T[2]; TL[0] = TL[1]; TR[0] = TR[1];
#pragma unroll
    for (int k = 0; k < 2; k++) { Q[k] = Q[k + 1]; QL[k] = QL[k + 1]; QR[k] = QR[k + 1]; }
    Q[2] = qnew; QL[2] = lane_below(qnew); QR[2] = lane_above(qnew);
    PL[3] = lane_below(P[3]); PR[3] = lane_above(P[3]);
    // zeta row j + 2, tmp = lap(zeta) row j + 1.  Ghost ROWS exist only next to a y wall: `yedge` (wave-uniform, set per
    // interval) keeps the row tests out of the interior intervals
    double z = xfix(lap5(P[3], PL[3], PR[3], P[4], P[2]), P[3], PL[3], PR[3]);
    if (yedge && north && j + 2 == ny) z = yghost(Z[2], ZL[2], ZR[2], P[2], P[3]);
    Z[3] = z; ZL[3] = lane_below(z); ZR[3] = lane_above(z);
    if (yedge && south && j + 2 == 0) {  // row -1 is the ghost of row 0, which exists only now
      Z[2] = yghost(Z[3], ZL[3], ZR[3], P[3], P[2]);
      ZL[2] = lane_below(Z[2]); ZR[2] = lane_above(Z[2]);
    }
    double t = xfix(lap5(Z[2], ZL[2], ZR[2], Z[3], Z[1]), Z[2], ZL[2], ZR[2]);
    if (yedge && north && j + 1 == ny) t = yghost(T[1], TL[0], TR[0], Z[1], Z[2]);
    T[2] = t; TL[1] = lane_below(t); TR[1] = lane_above(t);
    if (yedge && south && j + 1 == 0) T[1] = yghost(T[2], TL[1], TR[1], Z[2], Z[1]);
    if (!centre) return;

    double p[3][3], zz[3][3], p1[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      p[k][0] = PL[k]; p[k][1] = P[k]; p[k][2] = PR[k];
      zz[k][0] = ZL[k]; zz[k][1] = Z[k]; zz[k][2] = ZR[k];
      p1[k][0] = QL[k]; p1[k][1] = Q[k]; p1[k][2] = QR[k];
    }
    const double adv = (STOCH && l == 0) ? 0. : mjac9(p, zz, D12, rD12);
    const double be = DIVC(a.beta * (p[1][0] - p[1][2]), D2x, rD2x);
    const double jd = (lower && !STOCH) ? mjac9(p, p1, D12, rD12) : 0.;
    const double zc = Z[1], tc = T[1];
    const double lapT = lap5(tc, TL[0], TR[0], T[2], T[0]);
#ifdef MSOM_STRICT
    abA[r] = adv + be; jdA[r] = jd; lapTA[r] = lapT; zcA[r] = zc; tcA[r] = tc;
    ring[b][wv][r][0][lane] = zc;
    ring[b][wv][r][1][lane] = tc;
    ring[b][wv][r][2][lane] = jd;
#else
    // everything of layer l that does not need the neighbouring layers; the stretching terms share X = iRe zeta + iRe4 tmp
    double tl = fma(tc, a.iRe, fma(a.iRe4, lapT, adv + be));
    if (l == 0) tl -= fma(a.cs, zc, wind_row(j));
    if (l == nl - 1) tl = fma(-a.cb, zc, tl);
    tlA[r] = tl;
    ring[b][wv][r][0][lane] = fma(a.iRe, zc, a.iRe4 * tc);
    ring[b][wv][r][1][lane] = jd;
#endif
  };

  // finalise row j (slot r of ring buffer b): vertical coupling, forcing, advance; additions in the order of
  // msqg/qg.h:315-380,407-473 in the validation build.  Returns the value to store (q_out or dq)
  auto finish = [&](int j, int b, int r, double s0, double s1, double fq) -> double {
#ifdef MSOM_STRICT
    const double zc = zcA[r], tc = tcA[r];
    double zm = 0., tm = 0., zp = 0., tp = 0., ju = 0.;
    if (upper) { zm = ring[b][wv - 1][r][0][lane]; tm = ring[b][wv - 1][r][1][lane]; ju = -ring[b][wv - 1][r][2][lane]; }
    if (lower) { zp = ring[b][wv + 1][r][0][lane]; tp = ring[b][wv + 1][r][1][lane]; }
    double t = abA[r];
    if (nl > 1) {
      if (upper) t = t + s0 * ju * idh0;
      if (lower) t = t + s1 * jdA[r] * idh1;
    }
    t = 0. + t;  // updates were zeroed, then += (msqg/qg.h:611-613, 315)
    double dq = t;
    auto stretch = [&](double fac, double pm, double pc, double pp) -> double {
      if (l == 0) return fac * s1 * (pp - pc) * idh1;
      if (l < nl - 1) return fac * (s0 * (pm - pc) * idh0 + s1 * (pp - pc) * idh1);
      return fac * s0 * (pm - pc) * idh0;
    };
    if (a.iRe != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe, zm, zc, zp);
      dq += tc * a.iRe;
    }
    if (a.iRe4 != 0.) {
      if (nl > 1) dq = 1. * dq + stretch(a.iRe4, tm, tc, tp);
      dq = 1. * dq + a.iRe4 * lapTA[r];
    }
    if (l == 0) dq -= a.cs * zc;
    if (l == nl - 1) dq -= a.cb * zc;
    if (l == 0) dq -= wind_row(j);
#else
    double dq = tlA[r];
    const double xc = ring[b][wv][r][0][lane];
    if (nl > 1) {
      if (upper) dq = fma(s0 * idh0, (ring[b][wv - 1][r][0][lane] - xc) - ring[b][wv - 1][r][1][lane], dq);
      if (lower) dq = fma(s1 * idh1, (ring[b][wv + 1][r][0][lane] - xc) + ring[b][wv][r][1][lane], dq);
    }
#endif
    if (QF) dq += fq;
#ifdef MSOM_STRICT
    if (STOCH) return (qreg[r] + qsreg[r] * a.crelax + nzreg[r] * a.dts) + dq * a.dt;
    return ADV ? qreg[r] + dq * dtv : dq;
#else
    if (STOCH) return fma(dq, a.dt, fma(nzreg[r], a.dts, fma(qsreg[r], a.crelax, qreg[r])));
    return ADV ? fma(dq, dtv, qreg[r]) : dq;
#endif
  };

  // rows of the first interval start travelling before the warm-up
#pragma unroll
  for (int r = 0; r < R; r++) {
    pnext[r] = ld(pP, y0 + r + 3);
    qnext[r] = ld(pQ, y0 + r + 1);
  }
  // warm-up: fill the windows below the chunk (psi rows y0 - 3 .. y0 + 2, psi_{l+1} rows y0 - 1, y0; no centre terms).
  // All eight rows are requested before the first one is used (one memory latency per chunk instead of six)
  {
    double wp[6], wq[2];
#pragma unroll
    for (int k = 0; k < 6; k++) wp[k] = ld(pP, y0 - 3 + k);
    wq[0] = ld(pQ, y0 - 1); wq[1] = ld(pQ, y0);
#pragma unroll
    for (int k = 0; k < 6; k++) row(y0 - 6 + k, wp[k], k >= 4 ? wq[k - 4] : 0., false, EDGE, 0, 0);
  }

  // Rows past the end of a ragged chunk are computed on clamped addresses and never stored: the unrolled body has no
  // row-dependent branch (precise s_waitcnt counters, no register shuffles at control-flow joins).
  double *const outp = ADV ? a.q_out : a.dq;
  const int nblk = (y1 - y0 + R - 1) / R;
  // enter the loop with no load in flight: the compiler merges the pending-load state of the pre-header with that of
  // the back-edge, and a pre-header full of warm-up loads made it drain every prefetch at the top of each interval
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  for (int k = 0; k <= nblk; k++) {
    const int b = k & 1;
    if (k > 0) {
      const int jb = y0 + (k - 1) * R;
      double s0[R], s1[R], val[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        s0[r] = su0; s1[r] = su1;
        if (!UNI && nl > 1) {  // general S field: read per cell (not prefetched)
          const size_t c = nat_idx(a.g, l, min(jb + r, ny - 1), gic);
          if (upper) s0[r] = a.S[c - a.g.ls];
          if (lower) s1[r] = a.S[c];
        }
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        val[r] = finish(jb + r, b ^ 1, r, s0[r], s1[r], fqreg[r]);
        asm volatile("" : "+v"(val[r]));  // keep the last use of the prefetched q_in outside the predicated store
      }
#pragma unroll
      for (int r = 0; r < R; r++)
        if (out_ok && jb + r < y1 && !(a.dbg & 1)) outp[nat_idx(a.g, l, jb + r, gic)] = val[r];
    }
    if (k < nblk) {
      const int j0 = y0 + k * R;
      const bool yedge = EDGE && j0 + R + 2 > ny;  // rows j0 .. j0 + R - 1 build zeta / tmp rows up to j0 + R + 1
      // the inputs of the NEXT finalisation first: they are the oldest loads in flight when it starts
#pragma unroll
      for (int r = 0; r < R; r++) {
        const size_t c = nat_idx(a.g, l, (a.dbg & 2) ? y0 : (EDGE ? min(j0 + r, ny - 1) : j0 + r), gic);
        if (ADV) qreg[r] = a.q_in[c];
        if (QF) fqreg[r] = a.qforc[c];
        if (STOCH) { qsreg[r] = a.q_stage[c]; nzreg[r] = a.noise[c]; }
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int j = j0 + r;
        // a REAL register copy (opaque to the compiler): the prefetch register is dead right here, so the next
        // interval's load can land in the same physical register and no in-flight register is moved at the loop
        // back-edge (a coalesced copy made the allocator shuffle loaded registers there => s_waitcnt vmcnt(0))
        double pn, qn;
        asm volatile("v_mov_b64 %0, %1" : "=v"(pn) : "v"(pnext[r]));
        asm volatile("v_mov_b64 %0, %1" : "=v"(qn) : "v"(qnext[r]));
        pnext[r] = ld(pP, j + R + 3);
        qnext[r] = ld(pQ, j + R + 1);
        row(j, pn, qn, true, yedge, b, r);
      }
    }
    __syncthreads();
  }
}

template <int R, bool UNI, bool QF, bool ADV, bool STOCH = false>
__global__ void __launch_bounds__(64 * LPW_MAXW) k_rhs_lpw(LpwArgs a) {
  __shared__ double ring[2][LPW_MAXW][R][LPW_NV][64];
  // the same geometry as lpw_body (all wavefronts of a workgroup share the chunk, hence the number of barriers)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nx = a.g.nx, ny = a.g.ny;
  const int strip = blockIdx.x * a.NS + wv / a.nl, nstrips = (nx + LPW_W - 1) / LPW_W;
  const int x0 = min(strip, nstrips - 1) * LPW_W, y0 = blockIdx.y * a.H, y1 = min(ny, y0 + a.H);
  const bool xwall = ((a.walls & WALL_W) && x0 == 0) || ((a.walls & WALL_E) && nx - x0 + 3 <= 63);
  const int nblk = (y1 - y0 + R - 1) / R;
  const bool ywall = y0 == 0 || y0 + nblk * R + R > ny;   // (+ R: the rows the last interval prefetches stay inside the pad rows)
  if (a.region) {
    // psi is read on lanes x0 - 3 .. x0 + 60 and rows y0 - 3 .. y0 + nblk R + R + 2: inside the tile = no halo cell.  All wavefronts
    // of a workgroup share the chunk; strips differ, and a wavefront that has ended leaves the workgroup's barriers
    const bool inner = strip < nstrips && x0 - 3 >= 0 && x0 + 60 < nx && y0 - 3 >= 0 && y0 + nblk * R + R + 2 < ny;   // (+ R: the last interval's prefetch)
    if ((a.region == 1) != inner) return;
  }
  if (xwall || ywall || a.noedge_off) lpw_body<R, UNI, QF, ADV, STOCH, true>(a, ring);
  else lpw_body<R, UNI, QF, ADV, STOCH, false>(a, ring);
}

int g_lpw_dbg = 0;  // option lpw_dbg: bits 1, 2 timing experiments; 4: every wavefront takes the instantiation with the ghost-line code

void launch_rhs_lpw(hipStream_t st, const double *psi, const double *S, const double *qforc, const double *wind, double *dq, const NatGeom &g,
                    int nl, int walls, int uniformS, const double *Su, int have_qforc, double D, double beta, double iRe, double iRe4, double cs,
                    double cb, double slip_c, const LayerCoef &lc, const double *q_in, double *q_out, double dt, int chunk_rows, int stoch,
                    const double *q_stage, const double *noise, double crelax, double dts, int region, const double *dt_ptr) {
  LpwArgs a;
  a.region = region;
  a.dt_ptr = dt_ptr;
  a.q_stage = q_stage; a.noise = noise; a.crelax = crelax; a.dts = dts;
  a.psi = psi; a.S = S; a.qforc = qforc; a.wind = wind; a.q_in = q_in; a.dq = dq; a.q_out = q_out; a.dt = dt;
  extern int g_lpw_dbg;
  a.dbg = g_lpw_dbg & 3;
  a.noedge_off = (g_lpw_dbg & 4) != 0;
  a.g = g; a.nl = nl; a.walls = walls; a.uniformS = uniformS; a.have_qforc = have_qforc;
  a.D = D; a.beta = beta; a.iRe = iRe; a.iRe4 = iRe4; a.cs = cs; a.cb = cb; a.slip_c = slip_c; a.lc = lc;
  for (int l = 0; l < MSOM_MAXNL; l++) a.Su[l] = Su ? Su[l] : 0.;
  const int strips = (g.nx + LPW_W - 1) / LPW_W;
  int H = chunk_rows;
  a.NS = LPW_MAXW / nl < 1 ? 1 : LPW_MAXW / nl;
  if (a.NS > strips) a.NS = strips;
  if (H <= 0) {
    // One workgroup (all of a CU's wavefront slots at the product build's register count) per CU at a time: the launch takes
    // ceil(workgroups / CUs) rounds of H + 6 row steps (6 warm-up rows per chunk).  Round 3: the chunk height that minimises that product
    // -- the former rule (about 1024 / strips chunks, at most 64 rows) gave 2048^2 x 3 288 workgroups of 64 rows = 1.125 rounds, i.e. two
    // rounds of 70 steps where 40 rows give two of 46 (0.117 -> 0.09 ms per launch); 4096^2 x 6 keeps its 64 rows (9 rounds)
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t pr;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    const int wgx = (strips + a.NS - 1) / a.NS;
    // (workgroups do not end together: past three rounds the idle tail is taken as half a round rather than a whole one -- with whole
    // rounds 4096^2 x 3 would take 32 rows, measured 3 % slower than 64)
    double best = -1.;
    for (int h = 64; h >= 8; h -= 8) {
      const double r = (double)wgx * ((g.ny + h - 1) / h) / ncu, cost = (r <= 3. ? ceil(r) : r + 0.5) * (h + 6);
      if (best < 0. || cost < best) { best = cost; H = h; }
    }
  }
  if (H < 8) H = 8;
  a.H = H;
  const dim3 gr((strips + a.NS - 1) / a.NS, (g.ny + H - 1) / H), bl(64 * nl * a.NS);
  const int sel = (uniformS ? 4 : 0) | (have_qforc ? 2 : 0) | (q_out ? 1 : 0);
  if (stoch) {  // only with the advance fused (the caller folds -q/tau and the noise into q_in)
    switch (sel) {
      case 1: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, false, true, true>), gr, bl, 0, st, a); break;
      case 3: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, true, true, true>), gr, bl, 0, st, a); break;
      case 5: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, false, true, true>), gr, bl, 0, st, a); break;
      case 7: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, true, true, true>), gr, bl, 0, st, a); break;
      default: fprintf(stderr, "msom: launch_rhs_lpw: stochastic variant needs q_out\n"); abort();
    }
    return;
  }
  switch (sel) {
    case 0: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, false, false>), gr, bl, 0, st, a); break;
    case 1: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, false, true>), gr, bl, 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, true, false>), gr, bl, 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, false, true, true>), gr, bl, 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, false, false>), gr, bl, 0, st, a); break;
    case 5: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, false, true>), gr, bl, 0, st, a); break;
    case 6: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, true, false>), gr, bl, 0, st, a); break;
    default: hipLaunchKernelGGL((k_rhs_lpw<LPW_R, true, true, true>), gr, bl, 0, st, a); break;
  }
}
