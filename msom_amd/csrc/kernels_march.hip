// kernels_march.hip -- K consecutive red-black HALF-sweeps of relax_layer (msqg/poisson_layer.h:75-149) in one pass.
//
// The plain smoother (k_relax_color_x2) is HBM-bound at 3 w / 2 per half-sweep: it reads the other colour, reads the
// residual of its colour and writes its colour.  A half-sweep only needs the values the previous half-sweep produced
// at the four neighbours, so K of them can be chained while the data is in registers: a wavefront marches up the rows
// of a strip, half-sweep s runs one row behind half-sweep s - 1, and only the first one reads da from memory
// (w / 2), only the last two write (w): 2.5 w (+ halo) for K half-sweeps instead of 1.5 K w.
//
//  * split layout: a row is [even-x half | odd-x half]; lane k holds the cell pair (2k, 2k+1) of every row, i.e.
//    one cell of each colour.  N/S neighbours are the lane's own values of rows r +- 1, one of E/W is the lane's
//    own other-colour value of row r, the other one sits in the adjacent lane: one whole-wave DPP shift per
//    value (rhs_inl.h), no LDS, no barriers -- every wavefront is an independent workgroup;
//  * per half-sweep a 3-row window of the previous half-sweep's values (nl doubles per row and lane);
//  * the column solve is the uniform-S Thomas recurrence of relax_color_pt, same expression order => bit-identical;
//  * wall ghosts (homogeneous Dirichlet, lagged: ghost = -value of the wall cell at ITS last update) are recreated in
//    the lane / row that holds the ghost position: after half-sweep s the ghost positions of its colour take
//    -(value of the mirrored wall cell after half-sweep s - 1); the ghosts of the input come from memory;
//  * out of place (in -> out): a chunk re-computes K rows of its neighbours (cone of dependence), which those
//    neighbours overwrite; lanes: 2 x ceil(K/2) halo lanes of 64.
// Used on one GPU (no halo exchange between half-sweeps), walls (not the periodic domain), uniform S, nl >= 2.
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include "mg_inl.h"
#include "rhs_inl.h"

// Product build: no automatic contraction in this file; the fused multiply-adds of the column solve are written out in the two
// helpers below, which every body of the pass calls (register-window kernel, LDS-DMA kernel, its lean interior body).  A
// cell is relaxed by one body or another depending on where the chunk and tile edges fall, and left to the compiler the
// contraction follows the code around the expression: lean and general bodies differed in the last bit
// (tests/test_gpu_fullsize.py), which breaks "tiled = single tile, bit for bit".  The validation build never contracts.
#pragma clang fp contract(off)

// right-hand side of one cell's column system, one layer: -Delta^2 res + (W + E) + (N + S)   (msqg/poisson_layer.h:101-110)
__device__ __forceinline__ double march_rhs(double sqD, double rs, double we, double ns) {
#ifdef MSOM_STRICT
  double v = -sqD * rs;
  v += we;
  v += ns;
  return v;
#else
  return fma(-sqD, rs, we) + ns;
#endif
}
// uniform-S Thomas solve with the factors of RelaxCoef (msqg/poisson_layer.h:137-146; same recurrence as relax_color_pt)
template <int NL>
__device__ __forceinline__ void march_thomas(double (&rhs)[NL], double (&x)[NL], const RelaxCoef &rc) {
#ifdef MSOM_STRICT
#pragma unroll
  for (int l = 1; l < NL; l++) rhs[l] -= rc.w[l] * rhs[l - 1];
  x[NL - 1] = rhs[NL - 1] * rc.it1[NL - 1];
#pragma unroll
  for (int l = NL - 2; l >= 0; l--) x[l] = (rhs[l] - rc.t2[l] * x[l + 1]) * rc.it1[l];
#else
#pragma unroll
  for (int l = 1; l < NL; l++) rhs[l] = fma(-rc.w[l], rhs[l - 1], rhs[l]);
  x[NL - 1] = rhs[NL - 1] * rc.it1[NL - 1];
#pragma unroll
  for (int l = NL - 2; l >= 0; l--) x[l] = fma(-rc.t2[l], x[l + 1], rhs[l]) * rc.it1[l];
#endif
}

struct MarchArgs {
  const double *in, *res;
  double *out;
  // tiles of a multi-GPU run: rows below / above a tile edge that is not a wall come from halo arrays (the KR nearest
  // rows of the neighbour tile, same row layout, hls doubles per layer); the columns beyond W / E edges sit in the row pads
  const double *in_s, *in_n, *res_s, *res_n;
  size_t hls;
  int KR;
  // PL variant: the input of the first half-sweep is the bilinear prolongation of the coarser level's correction
  // (mspg/elliptic.h:74-86), interpolated on the fly and never stored; `in` is not read
  const double *coarse;
  SplitGeom cg;
  // tiles: the cKR nearest rows of the coarse correction beyond a tile edge that is not a wall (same layout, chls doubles per
  // layer); the columns beyond W / E edges sit in the row pads of `coarse` (deep halo exchange of the coarse level)
  const double *coarse_s, *coarse_n;
  size_t chls;
  int cKR;
  // CORR variant (last pass of the finest level): the correction a += da (mspg/elliptic.h:92-98) rides in the pass:
  // psi_out = psi + (value after the last update of each colour), natural layout, wall ghosts included; da is not stored
  const double *psi;
  double *psi_out;
  NatGeom ng;
  SplitGeom g;
  int c1;  // colour of the first half-sweep (0 red, 1 black)
  int walls, H, remap, flip;
  int partial;  // another half-sweep follows this pass: the colour updated by half-sweep K - 1 is overwritten before anybody reads it, only the last colour is stored
  int dbg;  // timing experiments only (results wrong): 1 = no stores, 2 = no loads after the first step
  int lean; // interior chunks take the lean body (march_lean below)
  int region;  // tiles, overlap with the deep halo exchange: 0 all chunks, 1 only those that read no halo cell, 2 only the others
  RelaxCoef rc;
};

template <int NL, int K, bool PL, bool CORR>
__global__ void __launch_bounds__(64, 2) k_relax_march(MarchArgs p) {
  // halo lanes per side: K cells of cone; the prolongation variant needs one more coarse cell (DPP neighbour) beyond them
  constexpr int HL = PL ? (K + 2) / 2 : (K + 1) / 2, OW = 64 - 2 * HL;
  // residual windows: half-sweeps 1, 3 use the residual of colour c1 at rows t, t - 2; half-sweeps 2, 4 that of the
  // other colour at rows t - 1, t - 3.  Every residual value is read from memory ONCE and waits in registers
  constexpr int D1 = K >= 3 ? 3 : 1, D2 = K >= 4 ? 3 : 1;
  const int lane = threadIdx.x;
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (p.remap) xcd_remap(bx, by);
  const int kx = (int)bx * OW - HL + lane;
  const int y0 = by * p.H, y1 = min(p.g.ny, y0 + p.H);
  const int hk = p.g.hk, ny = p.g.ny, hp = p.g.hp;
  if (p.region) {   // see k_relax_march_dma
    const int kxa = (int)bx * OW - HL;
    const bool inner = y0 - K >= 0 && y1 + K <= ny && kxa >= 0 && kxa + 63 <= hk - 1;
    if ((p.region == 1) != inner) return;
  }
  // Odd chunks march DOWN.  A colour half-sweep does not depend on the order of its cells, so the direction changes
  // nothing in the result; but two vertically adjacent chunks now reach their common edge at the same time (both at
  // their start or both at their end), so the 2 K rows they both read there are one HBM read and one L2 hit instead of
  // two HBM reads 24 rows apart.  `ph` maps the marching coordinate t (chunk-local, as if marching up) to the row.
  const bool down = p.flip && (by & 1);
  auto ph = [&](int t) -> int { return down ? y0 + y1 - 1 - t : t; };
  const ptrdiff_t rp = p.g.rp;
  const size_t ls = p.g.ls;
  const int kxc = min(max(kx, -2), hk + 1);  // stays inside the padded half row
  const bool own_lane = lane >= HL && lane < 64 - HL && kx < hk;  // kx >= 0 follows from lane >= HL
  const bool wallW = (p.walls & WALL_W) && kx - lane < 0, wallE = (p.walls & WALL_E) && kx - lane + 63 >= hk;  // wave-uniform
  const bool wallS = (p.walls & WALL_S) != 0, wallN = (p.walls & WALL_N) != 0;
  const double sqD = p.rc.sqD;
  auto off = [&](int half, int r) -> ptrdiff_t { return (ptrdiff_t)(min(max(r, -1), ny) + 1) * rp + half * hp + MSOM_SP + kxc; };
  // row r of a field whose rows beyond the tile live in halo arrays (wave-uniform choice); lstride = doubles per layer
  auto rowsrc = [&](const double *f, const double *fs, const double *fn, int half, int r, size_t &lstride) -> const double * {
    if (r < 0 && fs) { lstride = p.hls; return fs + (ptrdiff_t)(max(r, -p.KR) + p.KR + 1) * rp + half * hp + MSOM_SP + kxc; }
    if (r >= ny && fn) { lstride = p.hls; return fn + (ptrdiff_t)(min(r - ny, p.KR - 1) + 1) * rp + half * hp + MSOM_SP + kxc; }
    lstride = ls;
    return f + off(half, r);
  };

  // PL: colour-c0 cells of fine row r from the coarse level.  Fine cell (x, r): coarse cell (I, J) = (x >> 1, r >> 1) and
  // its neighbours towards the cell's quadrant (prolong_pt, kernels_mg.hip).  Lane k holds x = 2k, 2k + 1 => I = k for
  // both; the x-neighbour I +- 1 is the adjacent lane's coarse cell (DPP).  Ghost rows / columns of the prolongated
  // field are -P(wall cell) (what the red + prolongation kernel writes as lagged ghosts).
  const int chp = p.cg.hp;
  const ptrdiff_t crp = p.cg.rp;
  const size_t cls = p.cg.ls;
  const int Ic = min(max(kx, -1), p.cg.nx);  // coarse column of this lane (ghosts -1 and nx_c included)
  const ptrdiff_t coff = (Ic & 1) * chp + MSOM_SP + (Ic >> 1);
  auto prolong_row = [&](int r, int half, double (&dst)[NL]) {
    bool neg = false;
    int rr = r;
    if (r < 0) { rr = 0; neg = wallS; }
    if (r >= ny) { rr = ny - 1; neg = wallN; }
    const int J = rr >> 1, cy = (rr & 1) ? 1 : -1;
    const double *c0p = p.coarse + (ptrdiff_t)(J + 1) * crp + coff, *c1p = p.coarse + (ptrdiff_t)(J + cy + 1) * crp + coff;
    const bool gw = !neg && wallW && half == 1, ge = !neg && wallE && half == 0;  // a ghost column of this colour in this row
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double a0 = c0p[l * cls], a1 = c1p[l * cls];
      const double b0 = half ? lane_above(a0) : lane_below(a0), b1 = half ? lane_above(a1) : lane_below(a1);
      double v = BILINEAR(a0, b0, a1, b1);
      if (neg) v = -v;
      if (gw | ge) {  // -P(wall cell): the wall cell is the lane's / the neighbour lane's cell of the OTHER half
        const double o0 = half ? lane_below(a0) : lane_above(a0), o1 = half ? lane_below(a1) : lane_above(a1);
        const double po = BILINEAR(a0, o0, a1, o1);
        const double gv = gw ? -lane_above(po) : -lane_below(po);
        if (gw ? kx == -1 : kx == hk) v = gv;
      }
      dst[l] = v;
    }
  };

  double W[K][3][NL];  // W[s]: values after half-sweep s (s = 0: the input) of rows r - 1, r, r + 1 of the half-sweep that reads them
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int l = 0; l < NL; l++) W[s][q][l] = 0.;

  double R1[D1][NL], R2[D2][NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
#pragma unroll
    for (int d = 0; d < D1; d++) R1[d][l] = 0.;
#pragma unroll
    for (int d = 0; d < D2; d++) R2[d][l] = 0.;
  }
  const int c0 = 1 - p.c1;  // colour of the input values
  // rows y0 - K and y0 - K + 1 of the input fill the first window before half-sweep 1 starts at row y0 - K + 1
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = ph(y0 - K + q);
    if (PL) prolong_row(r, (r + c0) & 1, W[0][q + 1]);
    else {
      size_t st;
      const double *src = rowsrc(p.in, p.in_s, p.in_n, (r + c0) & 1, r, st);
#pragma unroll
      for (int l = 0; l < NL; l++) W[0][q + 1][l] = src[l * st];
    }
  }
  for (int t = y0 - K + 1; t <= y1 + K - 2; t++) {
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
      for (int l = 0; l < NL; l++) { W[s][0][l] = W[s][1][l]; W[s][1][l] = W[s][2][l]; }
#pragma unroll
    for (int l = 0; l < NL; l++) {
#pragma unroll
      for (int d = D1 - 1; d > 0; d--) R1[d][l] = R1[d - 1][l];
#pragma unroll
      for (int d = D2 - 1; d > 0; d--) R2[d][l] = R2[d - 1][l];
    }
    {
      size_t s0 = 0, s1, s2;
      const int rn = ph(t + 1), ra = ph(t), rb = ph(t - 1);   // rows of the new input row and of half-sweeps 1 and 2
      const double *src = PL ? p.res : rowsrc(p.in, p.in_s, p.in_n, (rn + c0) & 1, rn, s0);
      // residual rows: beyond a wall they are never used (clamped), beyond a tile edge they come from the halo arrays
      const int t1 = (ra < 0 && !p.res_s) ? 0 : ((ra >= ny && !p.res_n) ? ny - 1 : ra);
      const int t2 = (rb < 0 && !p.res_s) ? 0 : ((rb >= ny && !p.res_n) ? ny - 1 : rb);
      const double *r1 = rowsrc(p.res, p.res_s, p.res_n, (ra + p.c1) & 1, t1, s1);           // colour c1, row of half-sweep 1
      const double *r2 = rowsrc(p.res, p.res_s, p.res_n, (rb + c0) & 1, t2, s2);             // colour c0, row of half-sweep 2
#pragma unroll
      for (int l = 0; l < NL; l++) {
        if (!PL) W[0][2][l] = src[l * s0];
        R1[0][l] = r1[l * s1]; R2[0][l] = r2[l * s2];
      }
      if (PL) prolong_row(ph(t + 1), (ph(t + 1) + c0) & 1, W[0][2]);
    }
    // CORR: psi of the row the last half-sweep finishes in this step, requested before the chain of half-sweeps
    double2 pa[CORR ? NL : 1];
    if (CORR) {
      const int rk = min(max(ph(t - (K - 1)), 0), ny - 1);
      const double *ps = p.psi + nat_idx(p.ng, 0, rk, 2 * min(max(kx, 0), hk - 1));
#pragma unroll
      for (int l = 0; l < NL; l++) pa[l] = *reinterpret_cast<const double2 *>(ps + l * p.ng.ls);
    }
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = ph(t - (s - 1));             // row of half-sweep s
      const int px = (r + p.c1 + s - 1) & 1;     // x parity (= half) of its cells in that row
      double x[NL];
      // ghost row: -(wall row after the previous half-sweep); that row is the next one of the window when the chunk
      // marches towards it, the previous one otherwise
      if (wallS && r == -1) {
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = -W[s - 1][down ? 0 : 2][l];
      } else if (wallN && r == ny) {
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = -W[s - 1][down ? 2 : 0][l];
      } else {
        double rhs[NL], rs[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) rs[l] = (s & 1) ? R1[(s - 1) < D1 ? (s - 1) : 0][l] : R2[(s - 2) >= 0 && (s - 2) < D2 ? (s - 2) : 0][l];
        if (px) {  // odd-half cells: W is the lane's own even-half value, E the next lane's
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double a = W[s - 1][1][l];
            rhs[l] = march_rhs(sqD, rs[l], lane_above(a) + a, W[s - 1][2][l] + W[s - 1][0][l]);
          }
        } else {   // even-half cells: E is the lane's own odd-half value, W the previous lane's
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double a = W[s - 1][1][l];
            rhs[l] = march_rhs(sqD, rs[l], a + lane_below(a), W[s - 1][2][l] + W[s - 1][0][l]);
          }
        }
        march_thomas<NL>(rhs, x, p.rc);
        // ghost columns of this colour in this row: x = -1 is an odd-half position, x = nx an even-half one
        if (wallW && px == 1) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double gv = -lane_above(W[s - 1][1][l]);
            if (kx == -1) x[l] = gv;
          }
        }
        if (wallE && px == 0) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double gv = -lane_below(W[s - 1][1][l]);
            if (kx == hk) x[l] = gv;
          }
        }
      }
      if (s < K) {
#pragma unroll
        for (int l = 0; l < NL; l++) W[s][2][l] = x[l];
      }
      if (!CORR && s >= K - 1 && r >= y0 && r < y1 && own_lane) {  // the last update of each colour is what the level keeps
        double *dst = p.out + off(px, r);
#pragma unroll
        for (int l = 0; l < NL; l++) dst[l * ls] = x[l];
        const int i = 2 * kx + px;
        if (i == 0 || i == p.g.nx - 1 || r == 0 || r == ny - 1) {
#pragma unroll
          for (int l = 0; l < NL; l++) split_write_ghosts(p.out, p.g, l, r, i, x[l], p.walls);
        }
      }
      if (CORR && s == K && r >= y0 && r < y1 && own_lane) {
        // both colours of row r are final now: this half-sweep's cell and the other one, which the previous half-sweep
        // updated one step ago (the centre row of its window): one 16-byte load and store per lane and layer
        const size_t c = nat_idx(p.ng, 0, r, 2 * kx);
        const bool wallcell = kx == 0 || kx == hk - 1 || r == 0 || r == ny - 1;
#pragma unroll
        for (int l = 0; l < NL; l++) {
          const double2 a = pa[l];
          const double de = px ? W[K - 1][1][l] : x[l], dd = px ? x[l] : W[K - 1][1][l];
          const double ve = a.x + de, vo = a.y + dd;
          *reinterpret_cast<double2 *>(p.psi_out + c + l * p.ng.ls) = make_double2(ve, vo);
          if (wallcell) {
            nat_write_ghosts(p.psi_out, p.ng, l, r, 2 * kx, ve, p.walls);
            nat_write_ghosts(p.psi_out, p.ng, l, r, 2 * kx + 1, vo, p.walls);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// march_lean (round 3): the body of k_relax_march_dma for INTERIOR chunks -- every row and every lane the chunk touches
// lies inside the level (no wall, no tile edge), which is all but the outermost ring of chunks.
//
// SQ counters of the round-2 body (profiles/r02_pmc_sq_march.json): ~570 vector + ~445 scalar wave-instructions per
// marching step at nl = 6, K = 4 with the prolongation -- the scalar half (row clamps, wall tests, 64-bit address
// products, an m0 save / restore per LDS-DMA) and ~950 branches made the pass issue-bound (0.30 ms with no memory
// traffic at all).  Here:
//  * all rows of a step derive from ONE byte offset that advances by a constant per step (the colour half alternates, so
//    two constants); streams are (uniform 64-bit base in SGPRs) + (32-bit per-lane offset in a VGPR): no clamps, no
//    multiplies, no per-access address arithmetic in vector registers;
//  * the x parity of the updated cells is the same for all K half-sweeps of a step and alternates from step to step:
//    the loop is unrolled by two with the parity as a compile-time constant (whole-wave shift direction, colour half);
//  * the 3-row windows rotate by renaming inside the pair of steps: two register moves per value and PAIR instead of
//    two per step;
//  * the LDS-DMA requests of a step are one sequence with a single m0 save / restore;
//  * loads, stores and LDS-DMA retire in issue order on the vector-memory counter, the requests of step t + 1 are issued
//    BEFORE the stores of step t, so the wait at the top of a step is vmcnt(number of stores): the write
//    acknowledgements are no longer waited for.
// Arithmetic: the expressions of the general body, same order => bit-identical (tests/test_gpu_march.py).
template <int NL, int K, int HL, int WPB, bool PL, bool CORR>
struct MarchLeanRows {
  static constexpr int NLE = (NL + 1) & ~1;                 // two layer-rows per DMA instruction; odd NL repeats its last layer
  static constexpr int CB = 2 * NLE;                        // PL: coarse ring (4 slots of NLE rows)
  static constexpr int PBL = 3 * NLE;                       // CORR: psi block (NL x 128 doubles)
  static constexpr int XBL = PBL + 2 * NL;                  // CORR: values of half-sweep K waiting one step for their psi row
  static constexpr int ROWS = PL ? CB + 4 * NLE : (CORR ? XBL + NL : 3 * NLE);
  // DEEP (requests two steps ahead; not with CORR, whose psi block and parked values leave no room): two buffers of
  // residual (+ input) rows; PL: then a coarse ring of two slots (slot = J & 1: the row a request overwrites was read
  // for the last time in the step that issues it)
  static constexpr int RB = PL ? 2 * NLE : 3 * NLE;
  static constexpr int CBD = 2 * RB;
  static constexpr int ROWSD = PL ? CBD + 2 * NLE : 2 * RB;
};

#define MARCH_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NL, int K, int HL, int WPB, bool PL, bool CORR, bool DEEP>
__device__ __forceinline__ void march_lean(const MarchArgs &p, double (*ring)[64], const int lane, const int kx0, const int y0, const int y1,
                                           const bool down) {
  using LR = MarchLeanRows<NL, K, HL, WPB, PL, CORR>;
  constexpr int D1 = K >= 3 ? 3 : 1, D2 = K >= 4 ? 3 : 1;
  static_assert(!(DEEP && CORR), "no LDS for a second buffer beside the psi block");
  constexpr int NLE = LR::NLE, ND = NLE / 2, CB = DEEP ? LR::CBD : LR::CB, PBL = LR::PBL, XBL = LR::XBL, RB = LR::RB;
  constexpr int CSLOTS = DEEP ? 2 : 4;
  constexpr int NREQ = (PL ? 2 : 3) * ND;   // LDS-DMA instructions of one request of residual (+ input) rows
  const int kx = kx0 + lane;
  const bool own_lane = lane >= HL && lane < 64 - HL;   // interior strip: every lane of the wave lies inside the level
  const int d = down ? -1 : 1;
  const long long rp8 = (long long)p.g.rp * 8, hp8 = (long long)p.g.hp * 8, ls8 = (long long)p.g.ls * 8;
  const long long drp8 = down ? -rp8 : rp8;
  const double sqD = p.rc.sqD;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)(&ring[0][0]));
  const int sub = lane >> 5;
  // per-lane byte offsets inside a layer-row pair of a split field (requests) and of the lane's own cell (stores)
  unsigned voffL[ND];
#pragma unroll
  for (int q = 0; q < ND; q++) {
    const int lsel = sub ? (2 * q + 1 < NL ? 2 * q + 1 : NL - 1) : 2 * q;
    voffL[q] = (unsigned)(((size_t)lsel * p.g.ls + MSOM_SP + kx0 + 2 * (lane & 31)) * 8);
  }
  const unsigned voffS = (unsigned)((MSOM_SP + kx) * 8);
  // PL: the lane's 16-byte piece of a coarse row pair (see k_relax_march_dma) and where its own coarse cell lands in the ring
  unsigned voffC[PL ? ND : 1];
  if constexpr (PL) {
#pragma unroll
    for (int q = 0; q < ND; q++) {
      const int la = 2 * q + sub < NL ? 2 * q + sub : NL - 1;
      voffC[q] = (unsigned)(((size_t)la * p.cg.ls + ((lane >> 4) & 1) * p.cg.hp + MSOM_SP + (kx0 >> 1) + 2 * (lane & 15)) * 8);
    }
  }
  const int cld = (lane & 1) * 32 + (lane >> 1);
  const long long crp8 = (long long)p.cg.rp * 8;
  // CORR: the lane's cell pair of a natural row
  const unsigned voffN = (unsigned)((MSOM_XP + 2 * kx) * 8);
  const long long np8 = (long long)p.ng.pitch * 8, nls8 = (long long)p.ng.ls * 8;
  const long long dnp8 = down ? -np8 : np8;

  const int tA = y0 - K + 1, tB = y1 + K - 2;
  int ra = down ? y0 + y1 - 1 - tA : tA;                               // row of half-sweep 1 in the current step
  long long rowoff = ((long long)(ra + 1) * p.g.rp + ((ra + p.c1) & 1) * p.g.hp) * 8;   // its colour half, in bytes
  const char *const res8 = reinterpret_cast<const char *>(p.res);
  const char *const resb8 = res8 - drp8;                               // row of half-sweep 2
  const char *const in8 = PL ? nullptr : reinterpret_cast<const char *>(p.in) + drp8;   // the new input row
  char *const outK8 = CORR ? nullptr : reinterpret_cast<char *>(p.out) - (long long)(K - 1) * drp8;   // row of half-sweep K
  // CORR: psi of the row half-sweep K finished one step ago, t - K in marching coordinates
  long long natoff = CORR ? ((long long)(ra - d * K + MSOM_YP) * p.ng.pitch) * 8 : 0;
  const char *const psi8 = reinterpret_cast<const char *>(p.psi);
  char *const pso8 = reinterpret_cast<char *>(p.psi_out);
  // PL: coarse rows of the ring
  const char *const co8 = reinterpret_cast<const char *>(p.coarse);

  unsigned m0keep;
#define MARCH_DMA(voff, base, ldsrow) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds0 + (unsigned)(ldsrow) * 512u) : "memory")
  // rows of the step whose half-sweep 1 sits at byte offset ro; with them (CORR) psi of the row at natural offset no
  auto request = [&](long long ro, long long no, bool want_psi, int buf = 0) {
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0keep));
    const char *b1 = res8 + ro, *b2 = resb8 + ro;
    const int r0 = buf * RB;
#pragma unroll
    for (int q = 0; q < ND; q++) MARCH_DMA(voffL[q], b1, r0 + 2 * q);
#pragma unroll
    for (int q = 0; q < ND; q++) MARCH_DMA(voffL[q], b2, r0 + NLE + 2 * q);
    if constexpr (!PL) {
      const char *b3 = in8 + ro;
#pragma unroll
      for (int q = 0; q < ND; q++) MARCH_DMA(voffL[q], b3, r0 + 2 * NLE + 2 * q);
    }
    if constexpr (CORR) {
      if (want_psi && own_lane) {
        const char *bp = psi8 + no;
#pragma unroll
        for (int l = 0; l < NL; l++) MARCH_DMA(voffN, bp + l * nls8, PBL + 2 * l);
      }
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0keep));
  };
  // PL: coarse row J into its ring slot
  auto request_coarse = [&](int J) {
    const char *cb = co8 + (long long)(J + 1) * crp8;
    const unsigned slotrow = (unsigned)(CB + ((J + 8) & (CSLOTS - 1)) * NLE) * 512u;
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0keep));
#pragma unroll
    for (int q = 0; q < ND; q++)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voffC[q]), "s"(cb), "s"(lds0 + slotrow + (unsigned)(2 * q) * 512u) : "memory");
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0keep));
  };

  double W[K][3][NL];
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int l = 0; l < NL; l++) W[s][q][l] = 0.;
  double R1[D1][NL], R2[D2][NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
#pragma unroll
    for (int q = 0; q < D1; q++) R1[q][l] = 0.;
#pragma unroll
    for (int q = 0; q < D2; q++) R2[q][l] = 0.;
  }
  // the first request; PL: the two coarse rows the first step interpolates from
  int rn = ra + d;   // the new input row of the current step
  if constexpr (PL) {
    const int J = rn >> 1, cy = (rn & 1) ? 1 : -1;
    request_coarse(J);
    request_coarse(J + cy);
  }
  const long long strideA = drp8 + hp8, strideB = drp8 - hp8;   // to the next step's row: from an even half / from an odd half
  request(rowoff, natoff, false, 0);
  // DEEP: vector-memory operations retire in issue order, so "the request of this buffer has landed" = "at most
  // (issued - mark) younger operations are outstanding"; counted at run time, waited for with the nearest constant below
  unsigned issued = NREQ, markR[2] = {NREQ, 0}, markC = 0;
  if constexpr (DEEP) {
    request(rowoff + (((ra + p.c1) & 1) ? strideB : strideA), 0, false, 1);
    issued += NREQ; markR[1] = issued;
  }
  // rows y0 - K and y0 - K + 1 (marching coordinates) of the input fill the first window (plain loads, once per chunk)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = ra + d * (q - 1);
    const int half = (r + 1 - p.c1) & 1;
    if constexpr (PL) {
      const int J = r >> 1, cy = (r & 1) ? 1 : -1;
      const size_t coff = (size_t)(kx & 1) * p.cg.hp + MSOM_SP + (kx >> 1);
      const double *c0p = p.coarse + (size_t)(J + 1) * p.cg.rp + coff, *c1p = p.coarse + (size_t)(J + cy + 1) * p.cg.rp + coff;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double a0 = c0p[l * p.cg.ls], a1 = c1p[l * p.cg.ls];
        const double b0 = half ? lane_above(a0) : lane_below(a0), b1 = half ? lane_above(a1) : lane_below(a1);
        W[0][q][l] = BILINEAR(a0, b0, a1, b1);
      }
    } else {
      const double *src = p.in + (size_t)(r + 1) * p.g.rp + (size_t)half * p.g.hp + MSOM_SP + kx;
#pragma unroll
      for (int l = 0; l < NL; l++) W[0][q][l] = src[l * p.g.ls];
    }
  }
  // stores a full step issues after its requests (what the counted wait at the top of the next step leaves in flight)
  const int nstore = CORR ? NL : (p.partial ? NL : 2 * NL);
  bool full_prev = false;
  const int tK = y0 + K - 1;     // first step whose half-sweep K lands on a row of the chunk (the last one is tB)

  auto step = [&](auto pxc, auto phic, const int t) {
    constexpr int PX = decltype(pxc)::value;      // x parity (= colour half) of the cells updated in this step
    constexpr int PHI = decltype(phic)::value;    // position in the pair of steps: window slots (old, mid, new)
    constexpr int O = PHI ? 1 : 0, M = PHI ? 2 : 1, NW = PHI ? 0 : 2;
    constexpr int RA0 = PHI ? 2 : 0, RA2 = PHI ? 1 : 2;   // residual windows: slots of age 0 (new) and age 2
    if (WPB > 1) __builtin_amdgcn_s_barrier();
    if constexpr (DEEP) {
      const unsigned need = PL ? max(markR[PHI], markC) : markR[PHI];
      const unsigned n = issued - need;
      if (n >= NREQ + 4 * NL) MARCH_VMCNT(NREQ + 4 * NL);
      else if (n >= NREQ + 3 * NL) MARCH_VMCNT(NREQ + 3 * NL);
      else if (n >= NREQ + 2 * NL) MARCH_VMCNT(NREQ + 2 * NL);
      else if (n >= NREQ + NL) MARCH_VMCNT(NREQ + NL);
      else if (n >= NREQ) MARCH_VMCNT(NREQ);
      else MARCH_VMCNT(0);
    } else if (full_prev && !(p.dbg & 8)) {
      if constexpr (CORR) MARCH_VMCNT(NL);
      else if (p.partial) MARCH_VMCNT(NL);
      else MARCH_VMCNT(2 * NL);
    } else MARCH_VMCNT(0);
    double A0[PL ? NL : 1], A1[PL ? NL : 1];
    if constexpr (PL) {
      const int J = rn >> 1, cy = (rn & 1) ? 1 : -1;
      const double *s0 = &ring[CB + ((J + 8) & (CSLOTS - 1)) * NLE][cld], *s1 = &ring[CB + ((J + cy + 8) & (CSLOTS - 1)) * NLE][cld];
#pragma unroll
      for (int l = 0; l < NL; l++) { A0[l] = s0[64 * l]; A1[l] = s1[64 * l]; }
    }
    constexpr int R0 = DEEP ? PHI * RB : 0;   // DEEP: the buffer of this step
#pragma unroll
    for (int l = 0; l < NL; l++) {
      R1[D1 == 3 ? RA0 : 0][l] = ring[R0 + l][lane];
      R2[D2 == 3 ? RA0 : 0][l] = ring[R0 + NLE + l][lane];
      if constexpr (!PL) W[0][NW][l] = ring[R0 + 2 * NLE + l][lane];
    }
    // CORR: the row half-sweep K finished in the previous step: its psi and its values wait in the ring
    const bool corr_now = CORR && t - K >= y0;
    double2 pa[CORR ? NL : 1];
    double xv[CORR ? NL : 1];
    if constexpr (CORR) {
      if (corr_now) {
        const double *pb = &ring[0][0] + PBL * 64 + 2 * lane;
#pragma unroll
        for (int l = 0; l < NL; l++) { pa[l] = *reinterpret_cast<const double2 *>(pb + 128 * l); xv[l] = ring[XBL + l][lane]; }
      }
    }
    // the ring is free again once these reads have returned: the next step's rows have this whole step to arrive
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    asm volatile("" ::: "memory");
    const long long ronext = rowoff + (PX ? strideB : strideA);
    if constexpr (DEEP) {
      if (!(p.dbg & 2)) {
        if constexpr (PL) {
          // rows J, J + cy of the next step's input row: one of them is new on every second step
          if (t < tB && (PX == 1) != down) { request_coarse(((rn + d) >> 1) + d); issued += ND; markC = issued; }
        }
        if (t + 2 <= tB) { request(rowoff + 2 * drp8, 0, false, PHI); issued += NREQ; markR[PHI] = issued; }
      }
    } else if (t < tB) {
      if (!(p.dbg & 2)) {
        if constexpr (PL) {
          if ((PX == 1) != down) request_coarse(((rn + d) >> 1) + d);
        }
        request(ronext, natoff + dnp8, t + 1 - K >= y0);
      }
    } else if constexpr (CORR) {
      if (own_lane) {
        asm volatile("s_mov_b32 %0, m0" : "=s"(m0keep));
        const char *bp = psi8 + natoff + dnp8;
#pragma unroll
        for (int l = 0; l < NL; l++) MARCH_DMA(voffN, bp + l * nls8, PBL + 2 * l);
        asm volatile("s_mov_b32 m0, %0" ::"s"(m0keep));
      }
    }
    if constexpr (CORR) {
      if (corr_now && own_lane && !(p.dbg & 1)) {
        // psi_out = psi + da: this step's parity is the one of half-sweep K's cells one step ago flipped
        char *po = pso8 + natoff;
#pragma unroll
        for (int l = 0; l < NL; l++) {
          const double yv = W[K - 1][O][l];
          const double de = PX ? xv[l] : yv, dd = PX ? yv : xv[l];   // previous step: px = 1 - PX
          v2d o;
          o.x = pa[l].x + de; o.y = pa[l].y + dd;
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voffN), "v"(o), "s"(po + l * nls8) : "memory");
        }
      }
    }
    if constexpr (PL) {
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double a0 = A0[l], a1 = A1[l];
        const double b0 = PX ? lane_above(a0) : lane_below(a0), b1 = PX ? lane_above(a1) : lane_below(a1);
        W[0][NW][l] = BILINEAR(a0, b0, a1, b1);
      }
    }
    double x[NL];
#pragma unroll
    for (int s = 1; s <= K; s++) {
      double rhs[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double rs = (s & 1) ? R1[(s == 3 && D1 == 3) ? RA2 : (D1 == 3 ? RA0 : 0)][l] : R2[(s == 4 && D2 == 3) ? RA2 : (D2 == 3 ? RA0 : 0)][l];
        const double a = W[s - 1][M][l];
        rhs[l] = march_rhs(sqD, rs, PX ? lane_above(a) + a : a + lane_below(a), W[s - 1][NW][l] + W[s - 1][O][l]);
      }
      march_thomas<NL>(rhs, x, p.rc);
      if (s < K) {
#pragma unroll
        for (int l = 0; l < NL; l++) W[s][NW][l] = x[l];
      }
    }
    bool full = false;
    if constexpr (CORR) {
#pragma unroll
      for (int l = 0; l < NL; l++) ring[XBL + l][lane] = x[l];
      full = corr_now && !(p.dbg & 1);
    } else {
      // the last update of each colour is what the level keeps; partial: only the colour of half-sweep K (see the general body)
      const bool stK = t >= tK, stK1 = t >= tK - 1 && t < tB && !p.partial;
      if (own_lane && !(p.dbg & 1)) {
        char *bo = outK8 + rowoff;
        if (stK) {
#pragma unroll
          for (int l = 0; l < NL; l++) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" ::"v"(voffS), "v"(x[l]), "s"(bo + l * ls8) : "memory");
          issued += NL;
        }
        if (stK1) {
          bo += drp8;
#pragma unroll
          for (int l = 0; l < NL; l++) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" ::"v"(voffS), "v"(W[K - 1][NW][l]), "s"(bo + l * ls8) : "memory");
          issued += NL;
        }
      }
      full = stK && (p.partial || stK1) && !(p.dbg & 1);
    }
    full_prev = full;
    if constexpr (PHI == 1) {   // back to the slot names of the first step of a pair
#pragma unroll
      for (int s = 0; s < K; s++)
#pragma unroll
        for (int l = 0; l < NL; l++) { W[s][1][l] = W[s][0][l]; W[s][0][l] = W[s][2][l]; }
#pragma unroll
      for (int l = 0; l < NL; l++) {
        if constexpr (D1 == 3) { R1[1][l] = R1[2][l]; R1[2][l] = R1[0][l]; }
        if constexpr (D2 == 3) { R2[1][l] = R2[2][l]; R2[2][l] = R2[0][l]; }
      }
    }
    rowoff = ronext;
    natoff += dnp8;
    ra += d; rn += d;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if ((ra + p.c1) & 1) {
    for (int t = tA; t < tB; t += 2) { step(I1{}, I0{}, t); step(I0{}, I1{}, t + 1); }
  } else {
    for (int t = tA; t < tB; t += 2) { step(I0{}, I0{}, t); step(I1{}, I1{}, t + 1); }
  }
  if constexpr (CORR) {   // the last row of the chunk
    MARCH_VMCNT(0);
    if (own_lane && !(p.dbg & 1)) {
      const double *pb = &ring[0][0] + PBL * 64 + 2 * lane;
      char *po = pso8 + natoff;
      // the pair of steps ended: slot 0 holds the row of half-sweep K - 1 that half-sweep K visited last; its parity is the last step's
      const int pxl = (ra - d + p.c1) & 1;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const double2 a = *reinterpret_cast<const double2 *>(pb + 128 * l);
        const double xl = ring[XBL + l][lane], yv = W[K - 1][0][l];
        const double de = pxl ? yv : xl, dd = pxl ? xl : yv;
        *reinterpret_cast<double2 *>(po + l * nls8 + voffN) = make_double2(a.x + de, a.y + dd);
      }
    }
  }
#undef MARCH_DMA
}

// ---------------------------------------------------------------------------------------------------------------------
// k_relax_march_dma: the same pass with the memory side rebuilt (round 2).
//
// Measured on the register-window kernel above (tools/ab_march_dbg.py, 4096^2 x 6, K = 4): arithmetic alone 0.17 ms, with
// its loads 0.27 ms, with its stores 0.29 ms, everything 0.53 ms -- the time does not depend on K (K = 2: 0.51 ms) and
// loads and stores ADD instead of overlapping: every wavefront is an independent 60-lane strip whose 480-byte row pieces
// (18 read + 12 written per step, 8 bytes per lane, consumed in the step that requests them) reach the memory system
// one strip at a time.  Changes:
//  * the rows of step t + 1 are requested by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, 32 lanes per 512-byte
//    layer-row, two layer-rows per wave instruction, no VGPR destination) right after step t has moved its own rows
//    from LDS into registers, so a request has a whole step to land; one counted wait per step;
//  * WPB = 4 wavefronts per workgroup take 4 adjacent strips and march IN STEP (one s_barrier per row, no data
//    exchanged): the pieces they request and write at the same moment are 1.9 KB of one row instead of 4 unrelated
//    480-byte pieces (-5 % at K = 4, -14 % at K = 2 on the same box);
//  * PL: the prolongation rides in the pass (input of the first half-sweep interpolated on the fly from two coarse rows
//    per step, which arrive by LDS-DMA as well: the L2-latency loads at the head of every step that made the register
//    version of this variant lose are gone) -- a level visit is (PL + 4) + 4 half-sweeps in two passes instead of
//    (red + prolongation) + 4 + 3 in three.
// Arithmetic, windows, ghost rules and stores are the ones above (same expression order => bit-identical,
// tests/test_gpu_march.py).  LDS per wavefront: 3 NL (PL: 4 NL) layer-rows of 512 B = 9 (12) KB at NL = 6.
//  * CORR (last pass of the finest level): the correction a += da (mspg/elliptic.h:92-98) rides in the pass.  psi of the
//    row that half-sweep K finished in step t - 1 is requested with the rows of step t + 1 (LDS-DMA, 16 bytes per lane =
//    the lane's cell pair in the natural layout) and psi_out = psi + da of that row is written at the top of step t + 1,
//    right after the wait that covers the request: nothing of the chain waits for psi, and da is never stored.
template <int NL, int K, int HL, int WPB, bool PL, bool CORR = false>
__global__ void __launch_bounds__(64 * WPB, 2) k_relax_march_dma(MarchArgs p) {
  static_assert(!(PL && CORR), "the prolongation and the correction never ride in the same pass");
  static_assert(HL % 2 == 0 && (!PL || HL % 4 == 0), "16-byte pieces: strips start at even kx (PL: at kx = 0 mod 4)");
  constexpr int OW = 64 - 2 * HL;
  constexpr int D1 = K >= 3 ? 3 : 1, D2 = K >= 4 ? 3 : 1;
  // ring rows: [0, NL) residual of colour c1 (row t), [NL, 2 NL) residual of colour c0 (row t - 1), then either the input
  // row t + 1 (NL rows) or, PL, a ring of 4 coarse rows (slot = J & 3; NL rows each: [even half | odd half] x 32 cells):
  // a coarse row serves 4 fine rows and is fetched ONCE, one new row every second step
  constexpr int NLE = (NL + 1) & ~1;                       // two layer-rows per DMA instruction
  constexpr int CB = 2 * NLE;                              // first coarse ring row
  constexpr int PB = 2 * ((3 * NL + 1) / 2);                // CORR: first row of the psi block (NL x 128 doubles)
  constexpr int XB = PB + 2 * NL;                           // CORR: values of half-sweep K waiting one step for their psi row
  constexpr int LROWS0 = PL ? CB + 4 * NLE : (CORR ? XB + NL : 2 * ((3 * NL + 1) / 2));
  using LRows = MarchLeanRows<NL, K, HL, WPB, PL, CORR>;
  constexpr int LROWSL = (!CORR && LRows::ROWSD > LRows::ROWS) ? LRows::ROWSD : LRows::ROWS;
  constexpr int LROWS = LROWS0 > LROWSL ? LROWS0 : LROWSL;
  __shared__ __align__(16) double ring_all[WPB][LROWS][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double(*ring)[64] = ring_all[wv];
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (p.remap) xcd_remap(bx, by);
  const int kx0 = ((int)bx * WPB + wv) * OW - HL;
  const int kx = kx0 + lane;
  const int y0 = by * p.H, y1 = min(p.g.ny, y0 + p.H);
  const int hk = p.g.hk, ny = p.g.ny, hp = p.g.hp;
  const bool down = p.flip && (by & 1);
  // interior chunks (no wall, no tile edge within reach of any row or lane; an even number of rows): the lean body.  All
  // wavefronts of a workgroup share by, hence the number of steps and of barriers
  // chunks that read nothing beyond the tile: every row y0 - K .. y1 + K - 1 and every lane inside it.  On tiles they run
  // while the deep halo exchange is in flight (region 1), the others once it has arrived (region 2); a wavefront that has
  // ended no longer takes part in the workgroup's barriers
  const bool inner = y0 - K >= 0 && y1 + K <= ny && kx0 >= 0 && kx0 + 63 <= hk - 1;
  if (p.region && (p.region == 1) != inner) return;
  if (p.lean && inner && !((y1 - y0) & 1)) {
    if constexpr (!CORR) {
      if (p.lean >= 2) { march_lean<NL, K, HL, WPB, PL, CORR, true>(p, ring, lane, kx0, y0, y1, down); return; }
    }
    march_lean<NL, K, HL, WPB, PL, CORR, false>(p, ring, lane, kx0, y0, y1, down);
    return;
  }
  auto ph = [&](int t) -> int { return down ? y0 + y1 - 1 - t : t; };
  const ptrdiff_t rp = p.g.rp;
  const size_t ls = p.g.ls;
  const int kxc = min(max(kx, -2), hk + 1);
  const bool own_lane = lane >= HL && lane < 64 - HL && kx < hk;
  const bool wallW = (p.walls & WALL_W) && kx0 < 0, wallE = (p.walls & WALL_E) && kx0 + 63 >= hk;  // wave-uniform
  const bool wallS = (p.walls & WALL_S) != 0, wallN = (p.walls & WALL_N) != 0;
  const double sqD = p.rc.sqD;
  auto off = [&](int half, int r) -> ptrdiff_t { return (ptrdiff_t)(min(max(r, -1), ny) + 1) * rp + half * hp + MSOM_SP + kxc; };
  // DMA side: lane j fetches the 16-byte piece (cells e, e + 1 of the half row) of layer-row 2 d + (j >> 5)
  const int sub = lane >> 5;
  const int epc = MSOM_SP + min(max(kx0 + 2 * (lane & 31), -2), hk);  // piece start inside the padded half row (doubles)
  // wave-uniform row base of a field whose rows beyond the tile live in halo arrays; lstride = doubles per layer
  auto rowbase = [&](const double *f, const double *fs, const double *fn, int half, int r, size_t &lstride) -> const double * {
    if (r < 0 && fs) { lstride = p.hls; return fs + (ptrdiff_t)(max(r, -p.KR) + p.KR + 1) * rp + half * hp; }
    if (r >= ny && fn) { lstride = p.hls; return fn + (ptrdiff_t)(min(r - ny, p.KR - 1) + 1) * rp + half * hp; }
    lstride = ls;
    return f + (ptrdiff_t)(min(max(r, -1), ny) + 1) * rp + half * hp;
  };
  auto dma16 = [&](const double *gsrc, int ldsrow) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)(&ring[ldsrow][0]));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
  };
  const int c0 = 1 - p.c1;  // colour of the input values
  // PL: coarse cell of this lane = column kx of the coarse level; the lane's 16-byte coarse piece: quarter q = lane >> 4
  // = {row J, row J + cy} x {even half, odd half}, cells (kx0 >> 1) + 2 (lane & 15) (+1) of that half row
  const int chp = p.cg.hp;
  const ptrdiff_t crp = p.cg.rp;
  const size_t cls = p.cg.ls;
  // lanes 0-31 fetch layer 2 d, lanes 32-63 layer 2 d + 1; within each, lanes 0-15 the even half, 16-31 the odd half
  const int cpc = ((lane >> 4) & 1) * chp + MSOM_SP + min((kx0 >> 1) + 2 * (lane & 15), (p.cg.nx >> 1) + MSOM_SP - 2);
  const int cld = (lane & 1) * 32 + (lane >> 1);   // where the lane's own coarse cell sits in a 64-double coarse ring row
  int cjA = -1000, cjB = -1000;                    // the two newest coarse rows in the ring (wave-uniform)
  auto coarse_rows = [&](int r, bool &neg, int &J, int &cy) {
    neg = false;
    int rr = r;
    // beyond a wall: the ghost row mirrors the wall row; beyond a tile edge the rows are the neighbour's (coarse halo arrays)
    if (r < 0 && !p.coarse_s) { rr = 0; neg = wallS; }
    if (r >= ny && !p.coarse_n) { rr = ny - 1; neg = wallN; }
    J = rr >> 1;                      // arithmetic shift: rows -1, -2 -> coarse row -1
    cy = (rr & 1) ? 1 : -1;
  };
  // wave-uniform base of coarse row J (without the column offset); lstride = doubles per layer
  const int cny = p.cg.ny;
  auto crow = [&](int J, size_t &lstride) -> const double * {
    if (J < 0 && p.coarse_s) { lstride = p.chls; return p.coarse_s + (ptrdiff_t)(max(J, -p.cKR) + p.cKR + 1) * crp; }
    if (J >= cny && p.coarse_n) { lstride = p.chls; return p.coarse_n + (ptrdiff_t)(min(J - cny, p.cKR - 1) + 1) * crp; }
    lstride = cls;
    return p.coarse + (ptrdiff_t)(min(max(J, -1), cny) + 1) * crp;
  };
  // CORR: with the rows of step t comes psi of the row half-sweep K finished in step t - 1 (the lane's cell pair, natural layout)
  auto request_psi = [&](int t) {
    const int rk = ph(t - K);
    if (rk < y0 || rk >= y1) return;   // wave-uniform: rows of the chunk's halo are never corrected
    if (own_lane) {                    // nor are the halo lanes (inactive lanes fetch nothing)
      const double *ps = p.psi + nat_idx(p.ng, 0, rk, 2 * kx);
#pragma unroll
      for (int l = 0; l < NL; l++) dma16(ps + l * p.ng.ls, PB + 2 * l);
    }
  };
  // requests the rows of marching step t
  auto request = [&](int t) {
    size_t st[3];
    const double *bp[3];
    const int rn = ph(t + 1), ra = ph(t), rb = ph(t - 1);
    const int t1 = (ra < 0 && !p.res_s) ? 0 : ((ra >= ny && !p.res_n) ? ny - 1 : ra);
    const int t2 = (rb < 0 && !p.res_s) ? 0 : ((rb >= ny && !p.res_n) ? ny - 1 : rb);
    bp[0] = rowbase(p.res, p.res_s, p.res_n, (ra + p.c1) & 1, t1, st[0]);
    bp[1] = rowbase(p.res, p.res_s, p.res_n, (rb + c0) & 1, t2, st[1]);
    if (!PL) bp[2] = rowbase(p.in, p.in_s, p.in_n, (rn + c0) & 1, rn, st[2]);
    else { bp[2] = bp[1]; st[2] = st[1]; }
    constexpr int NPLAIN = PL ? 2 * NL : 3 * NL;
#pragma unroll
    for (int d = 0; d < (NPLAIN + 1) / 2; d++) {
      const int ra0 = 2 * d, ra1 = (2 * d + 1 < NPLAIN) ? 2 * d + 1 : NPLAIN - 1;   // an odd row count repeats the last row
      const double *g0 = bp[ra0 / NL] + (size_t)(ra0 % NL) * st[ra0 / NL];
      const double *g1 = bp[ra1 / NL] + (size_t)(ra1 % NL) * st[ra1 / NL];
      dma16((sub ? g1 : g0) + epc, 2 * d);
    }
    static_assert(!PL || 2 * ((2 * NL + 1) / 2) <= CB, "residual rows end before the coarse ring");
    if constexpr (CORR) request_psi(t);
    if constexpr (PL) {
      bool neg;
      int J, cy;
      coarse_rows(rn, neg, J, cy);
      // the rows J and J + cy must be in the ring when step t reads it; in steady state one of them is new every second step
      const int first = down ? max(J, J + cy) : min(J, J + cy), second = down ? min(J, J + cy) : max(J, J + cy);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int Jn = q ? second : first;
        if (Jn == cjA || Jn == cjB) continue;
        cjB = cjA; cjA = Jn;
        size_t cst;
        const double *cb = crow(Jn, cst) + cpc;
        const int slot = (Jn + 8) & 3;
#pragma unroll
        for (int d = 0; d < NLE / 2; d++) {
          const int la = min(2 * d + sub, NL - 1);
          dma16(cb + (size_t)la * cst, CB + slot * NLE + 2 * d);
        }
      }
    }
  };
  // PL: colour-c0 cells of fine row r, interpolated from the two coarse rows (prolong_pt, kernels_mg.hip); a0 / a1 = the
  // lane's coarse cell in rows J / J + cy.  Ghost rows / columns of the prolongated field are -P(wall cell).
  auto prolong_vals = [&](int r, int half, const double (&A0)[NL], const double (&A1)[NL], double (&dst)[NL]) {
    const bool neg = (r < 0 && wallS) || (r >= ny && wallN);
    const bool gw = !neg && wallW && half == 1, ge = !neg && wallE && half == 0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const double a0 = A0[l], a1 = A1[l];
      const double b0 = half ? lane_above(a0) : lane_below(a0), b1 = half ? lane_above(a1) : lane_below(a1);
      double v = BILINEAR(a0, b0, a1, b1);
      if (neg) v = -v;
      if (gw | ge) {
        const double o0 = half ? lane_below(a0) : lane_above(a0), o1 = half ? lane_below(a1) : lane_above(a1);
        const double po = BILINEAR(a0, o0, a1, o1);
        const double gv = gw ? -lane_above(po) : -lane_below(po);
        if (gw ? kx == -1 : kx == hk) v = gv;
      }
      dst[l] = v;
    }
  };

  double W[K][3][NL];
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int l = 0; l < NL; l++) W[s][q][l] = 0.;
  double R1[D1][NL], R2[D2][NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
#pragma unroll
    for (int d = 0; d < D1; d++) R1[d][l] = 0.;
#pragma unroll
    for (int d = 0; d < D2; d++) R2[d][l] = 0.;
  }
  const int tA = y0 - K + 1, tB = y1 + K - 2;
  // psi_out = psi + da of row r: the cell of parity px from half-sweep K (parked in LDS for one step: no registers), the
  // other one from yv (half-sweep K - 1, one step before it); psi of the row is read from the ring layer by layer
  auto correct_row = [&](int r, const double (&yv)[NL]) {
    if constexpr (CORR) {
      if (r >= y0 && r < y1 && own_lane && !(p.dbg & 1)) {
        const int px = (r + p.c1 + K - 1) & 1;
        double *po = p.psi_out + nat_idx(p.ng, 0, r, 2 * kx);   // wall ghosts of psi_out: one boundary pass after the launch
        const double *pb = &ring[0][0] + PB * 64 + 2 * lane;
#pragma unroll
        for (int l = 0; l < NL; l++) {
          const double2 a = *reinterpret_cast<const double2 *>(pb + 128 * l);
          const double xv = ring[XB + l][lane];
          const double de = px ? yv[l] : xv, dd = px ? xv : yv[l];
          *reinterpret_cast<double2 *>(po + l * p.ng.ls) = make_double2(a.x + de, a.y + dd);
        }
      }
    }
  };
  request(tA);
  // rows y0 - K and y0 - K + 1 of the input fill the first window (plain loads, once per chunk)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int r = ph(y0 - K + q);
    if constexpr (PL) {
      bool neg;
      int J, cy;
      coarse_rows(r, neg, J, cy);
      const int Ic = min(max(kx, -8), p.cg.nx + 7);   // ghosts / halo columns included (row pads)
      const ptrdiff_t coff = (Ic & 1) * chp + MSOM_SP + (Ic >> 1);
      size_t s0, s1;
      const double *c0p = crow(J, s0) + coff, *c1p = crow(J + cy, s1) + coff;
      double A0[NL], A1[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) { A0[l] = c0p[l * s0]; A1[l] = c1p[l * s1]; }
      prolong_vals(r, (r + c0) & 1, A0, A1, W[0][q + 1]);
    } else {
      size_t st;
      const double *src = rowbase(p.in, p.in_s, p.in_n, (r + c0) & 1, r, st) + MSOM_SP + kxc;
#pragma unroll
      for (int l = 0; l < NL; l++) W[0][q + 1][l] = src[l * st];
    }
  }
  for (int t = tA; t <= tB; t++) {
    if (WPB > 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the rows requested one step ago (and the stores issued since)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < K; s++)
#pragma unroll
      for (int l = 0; l < NL; l++) { W[s][0][l] = W[s][1][l]; W[s][1][l] = W[s][2][l]; }
#pragma unroll
    for (int l = 0; l < NL; l++) {
#pragma unroll
      for (int d = D1 - 1; d > 0; d--) R1[d][l] = R1[d - 1][l];
#pragma unroll
      for (int d = D2 - 1; d > 0; d--) R2[d][l] = R2[d - 1][l];
    }
    double A0[PL ? NL : 1], A1[PL ? NL : 1];
    int sJ0 = 0, sJ1 = 0;
    if constexpr (PL) {
      bool neg;
      int J, cy;
      coarse_rows(ph(t + 1), neg, J, cy);
      sJ0 = (J + 8) & 3; sJ1 = (J + cy + 8) & 3;
    }
#pragma unroll
    for (int l = 0; l < NL; l++) {
      R1[0][l] = ring[l][lane];
      R2[0][l] = ring[NL + l][lane];
      if constexpr (!PL) W[0][2][l] = ring[2 * NL + l][lane];
      else { A0[l] = ring[CB + sJ0 * NLE + l][cld]; A1[l] = ring[CB + sJ1 * NLE + l][cld]; }
    }
    // CORR: the row half-sweep K finished in the previous step (its other colour is one row down the window by now)
    correct_row(ph(t - K), W[K - 1][0]);
    // the ring is free again once these reads have returned: the next step's rows have this whole step to arrive
    if (t < tB && !(p.dbg & 2)) {
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      asm volatile("" ::: "memory");
      request(t + 1);
    } else if (CORR && t == tB) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      asm volatile("" ::: "memory");
      request_psi(t + 1);
    }
    if constexpr (PL) prolong_vals(ph(t + 1), (ph(t + 1) + c0) & 1, A0, A1, W[0][2]);
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = ph(t - (s - 1));             // row of half-sweep s
      const int px = (r + p.c1 + s - 1) & 1;     // x parity (= half) of its cells in that row
      double x[NL];
      if (wallS && r == -1) {
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = -W[s - 1][down ? 0 : 2][l];
      } else if (wallN && r == ny) {
#pragma unroll
        for (int l = 0; l < NL; l++) x[l] = -W[s - 1][down ? 2 : 0][l];
      } else {
        double rhs[NL], rs[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) rs[l] = (s & 1) ? R1[(s - 1) < D1 ? (s - 1) : 0][l] : R2[(s - 2) >= 0 && (s - 2) < D2 ? (s - 2) : 0][l];
        if (px) {  // odd-half cells: W is the lane's own even-half value, E the next lane's
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double a = W[s - 1][1][l];
            rhs[l] = march_rhs(sqD, rs[l], lane_above(a) + a, W[s - 1][2][l] + W[s - 1][0][l]);
          }
        } else {   // even-half cells: E is the lane's own odd-half value, W the previous lane's
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double a = W[s - 1][1][l];
            rhs[l] = march_rhs(sqD, rs[l], a + lane_below(a), W[s - 1][2][l] + W[s - 1][0][l]);
          }
        }
        march_thomas<NL>(rhs, x, p.rc);
        if (wallW && px == 1) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double gv = -lane_above(W[s - 1][1][l]);
            if (kx == -1) x[l] = gv;
          }
        }
        if (wallE && px == 0) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            const double gv = -lane_below(W[s - 1][1][l]);
            if (kx == hk) x[l] = gv;
          }
        }
      }
      if (s < K) {
#pragma unroll
        for (int l = 0; l < NL; l++) W[s][2][l] = x[l];
      }
      if constexpr (CORR) {
        if (s == K) {
#pragma unroll
          for (int l = 0; l < NL; l++) ring[XB + l][lane] = x[l];
        }
      }
      if (!CORR && s >= K - 1 && r >= y0 && r < y1 && own_lane && !(p.dbg & 1)) {  // the last update of each colour is what the level keeps
        // partial: the colour of half-sweep K - 1 is recomputed by the next pass before anything reads it -- except through
        // the wall ghosts that mirror its wall cells, which sit at positions of the OTHER colour and are input of that pass
        if (s == K || !p.partial) {
          double *dst = p.out + off(px, r);
#pragma unroll
          for (int l = 0; l < NL; l++) dst[l * ls] = x[l];
        }
        const int i = 2 * kx + px;
        if (i == 0 || i == p.g.nx - 1 || r == 0 || r == ny - 1) {
#pragma unroll
          for (int l = 0; l < NL; l++) split_write_ghosts(p.out, p.g, l, r, i, x[l], p.walls);
        }
      }
    }
  }
  if constexpr (CORR) {   // the last row of the chunk
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    correct_row(ph(tB - K + 1), W[K - 1][1]);
  }
}

int g_march_remap = 1;  // XCD-contiguous block numbering (option march_xcd)
int g_march_flip = 1;   // odd chunks march down (option march_flip)

// Chunk height.  All workgroups of a pass take the same time, so the pass ends with an idle tail unless their number
// is close to a whole number of rounds (resident wavefronts per CU x CUs, from the occupancy calculator for the
// instantiation at hand).  Measured at 4096^2 x 6, K = 4 (2048 resident): 24 rows = 2.92 rounds 0.440 ms, 36 rows = 1.95
// rounds 0.453, 72 rows = 0.97 rounds 0.474, but 28 rows = 2.51 rounds 0.469, 44 rows = 1.6 rounds 0.485, 56 rows = 1.26
// rounds 0.55.  Rule: the largest whole number of rounds (<= 3) whose chunks still have >= 20 rows (2 K of them are
// re-computed), at least one round; never below 16 rows.
template <typename Kern>
static void march_launch(hipStream_t st, Kern kern, MarchArgs a, int ow, int chunk_rows, int nthreads = 64) {
  const int strips = (a.g.hk + ow - 1) / ow;
  int H = chunk_rows;
  // Round 3, lean body: short chunks win although they re-compute more (2 K - 2 of H + 2 K - 2 steps): the halo rows are
  // L2 hits (vertically adjacent chunks reach their common edge together, march_flip), and many short workgroups even out
  // the memory traffic and the tail of the launch.  4096^2 x 6 (tools/ab_prof.py): 12-14 rows PL 0.388 / CORR 0.584 ms,
  // 16: 0.390 / 0.590, 20: 0.399 / 0.592, 26 (the round-2 rule below): 0.414 / 0.596, 32: 0.43 / 0.61, 48: 0.49 / 0.66
  if (H == 0 && a.lean) {
    H = 14;
    // levels with fewer chunks than the chip has wavefront slots (2 per SIMD): shorter chunks until one round is full -- a
    // marching wavefront is latency-bound, so concurrency buys more than the re-computed rows cost
    while (H > 6 && (size_t)strips * ((a.g.ny + H - 1) / H) * (nthreads / 64) < 2048) H -= 2;
  }
  if (H <= 0) {
    // occupancy per (device, instantiation), asked once; tiled tests drive this from several host threads
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, std::pair<int, int>> cache;  // -> (blocks per CU, CUs)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int per_cu = 0, ncu = 0;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = cache.find({dev, (const void *)kern});
      if (it == cache.end()) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nthreads, 0) != hipSuccess || per_cu <= 0) per_cu = 512 / nthreads;
        cache[{dev, (const void *)kern}] = {per_cu, ncu};
      } else {
        per_cu = it->second.first;
        ncu = it->second.second;
      }
    }
    const int slots = per_cu * ncu;
    H = 0;
    for (int m = 3; m >= 1 && !H; m--) {
      const int chunks = m * slots / strips;
      if (chunks < 1) continue;
      const int h = (a.g.ny + chunks - 1) / chunks;
      if (h >= 20 || m == 1) H = h;
    }
    if (H < 16) H = 16;
  }
  if (a.lean) H = (H + 1) & ~1;   // the lean body marches in pairs of steps
  a.H = H;
  hipLaunchKernelGGL(kern, dim3(strips, (a.g.ny + H - 1) / H), dim3(nthreads), 0, st, a);
}

int g_march_dbg = 0;    // timing experiments (option march_dbg)
int g_march_lean = 2;   // interior chunks of the LDS-DMA pass take the lean body (option march_lean; 2: requests two steps ahead where LDS allows)
int g_march_dma = 2;    // LDS-DMA version of the pass (option march_dma: 0 register-window kernel, 1 one strip, 2 four strips per workgroup)

template <int NL>
static int march_dispatch(hipStream_t st, const MarchArgs &a, int K, int rows) {
  if (g_march_dma && !(K == 4 && NL > 6) && !((a.coarse || a.psi_out) && NL > 6)) {
    // 1: one strip per workgroup; 2 (default): four adjacent strips per workgroup, marching in step, for the plain pass
    // (the pass with the prolongation measured faster with one: 7.09 vs 7.25 ms per RK2 step); 3: four for both
    constexpr int NLS = NL;
    const bool four = a.coarse ? g_march_dma >= 3 : g_march_dma >= 2;
#define MARCH_DMA_K(KK)                                                                                                      \
    if constexpr (NL <= 6) {   /* four strips per workgroup: 4 x the LDS ring; at nl = 7, 8 that leaves one workgroup per CU */   \
      if (a.coarse && four) { march_launch(st, k_relax_march_dma<NLS, KK, 4, 4, true>, a, 56 * 4, rows, 256); return 0; }        \
      if (a.psi_out && four) { march_launch(st, k_relax_march_dma<NLS, KK, 2, 4, false, true>, a, 60 * 4, rows, 256); return 0; } \
      if (!a.coarse && !a.psi_out && four) { march_launch(st, k_relax_march_dma<NLS, KK, 2, 4, false>, a, 60 * 4, rows, 256); return 0; } \
    }                                                                                                                        \
    if (a.coarse) march_launch(st, k_relax_march_dma<NLS, KK, 4, 1, true>, a, 56, rows, 64);                                   \
    else if (a.psi_out) march_launch(st, k_relax_march_dma<NLS, KK, 2, 1, false, true>, a, 60, rows, 64);                      \
    else march_launch(st, k_relax_march_dma<NLS, KK, 2, 1, false>, a, 60, rows, 64);                                           \
    return 0;
    if constexpr (NL <= 6) {
      switch (K) {
        case 2: if (!a.coarse) { MARCH_DMA_K(2) } break;
        case 3: { MARCH_DMA_K(3) }
        case 4: { MARCH_DMA_K(4) }
      }
    } else {
      switch (K) {
        case 2: if (!a.coarse) { MARCH_DMA_K(2) } break;
        case 3: { MARCH_DMA_K(3) }
      }
    }
#undef MARCH_DMA_K
  }
  switch (K) {
    case 2:
      if (a.psi_out) march_launch(st, k_relax_march<NL, 2, false, true>, a, 62, rows);
      else march_launch(st, k_relax_march<NL, 2, false, false>, a, 62, rows);
      return 0;
    case 3:
      if (a.coarse) march_launch(st, k_relax_march<NL, 3, true, false>, a, 60, rows);
      else if (a.psi_out) march_launch(st, k_relax_march<NL, 3, false, true>, a, 60, rows);
      else march_launch(st, k_relax_march<NL, 3, false, false>, a, 60, rows);
      return 0;
    case 4:
      if (a.coarse) march_launch(st, k_relax_march<NL, 4, true, false>, a, 58, rows);
      else if (a.psi_out) march_launch(st, k_relax_march<NL, 4, false, true>, a, 60, rows);
      else march_launch(st, k_relax_march<NL, 4, false, false>, a, 60, rows);
      return 0;
  }
  return -1;
}

// K (2..4) half-sweeps starting with colour c1, in -> out; returns -1 if (nl, K) has no instantiation
int launch_relax_march(hipStream_t st, const double *in, double *out, const double *res, const SplitGeom &sg, int nl, const RelaxCoef &rc, int c1,
                       int K, int walls, int chunk_rows, const MarchHalo *h, const double *coarse, const SplitGeom *cg, const MarchCorrect *mc, int more_follow, const MarchHalo *ch,
                       int region) {
  MarchArgs a;
  a.region = region;
  a.coarse_s = ch ? ch->in_s : nullptr; a.coarse_n = ch ? ch->in_n : nullptr; a.chls = ch ? ch->ls : 0; a.cKR = ch ? ch->rows : 0;
  a.partial = more_follow != 0;
  a.psi = mc ? mc->psi : nullptr; a.psi_out = mc ? mc->psi_out : nullptr;
  if (mc) a.ng = mc->g;
  if (mc && coarse) return -1;
  a.coarse = coarse; a.cg = cg ? *cg : sg;
  if (coarse && (K < 3 || (h && !ch))) return -1;  // the prolongation variant exists for K = 3, 4; on tiles it needs the coarse halo
  a.in_s = h ? h->in_s : nullptr; a.in_n = h ? h->in_n : nullptr; a.res_s = h ? h->res_s : nullptr; a.res_n = h ? h->res_n : nullptr;
  a.hls = h ? h->ls : 0; a.KR = h ? h->rows : 0;
  extern int g_march_remap;
  extern int g_march_flip;
  extern int g_march_dbg;
  extern int g_march_lean;
  // lean body: 32-bit per-lane byte offsets span all layers of a field; the prolongation variant is written for c1 = 0
  a.lean = !g_march_lean ? 0 : g_march_lean * (int)((size_t)(nl + 1) * sg.ls * 8 < ((size_t)1 << 32) && (!mc || (size_t)(nl + 1) * mc->g.ls * 8 < ((size_t)1 << 32)) && (!coarse || c1 == 0));
  a.in = in; a.out = out; a.res = res; a.g = sg; a.c1 = c1; a.walls = walls; a.rc = rc; a.remap = g_march_remap; a.flip = g_march_flip; a.dbg = g_march_dbg;
  if (nl >= 7 && K > 3) return -1;  // 4 windows of 7 or 8 layers do not fit 256 VGPRs
  switch (nl) {
    case 1: return march_dispatch<1>(st, a, K, chunk_rows);
    case 2: return march_dispatch<2>(st, a, K, chunk_rows);
    case 3: return march_dispatch<3>(st, a, K, chunk_rows);
    case 4: return march_dispatch<4>(st, a, K, chunk_rows);
    case 5: return march_dispatch<5>(st, a, K, chunk_rows);
    case 6: return march_dispatch<6>(st, a, K, chunk_rows);
    case 7: return march_dispatch<7>(st, a, K, chunk_rows);
    case 8: return march_dispatch<8>(st, a, K, chunk_rows);
  }
  return -1;
}
