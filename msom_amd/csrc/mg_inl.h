// mg_inl.h -- device helpers shared by the multigrid kernels (kernels_mg.hip, kernels_march.hip).
#ifndef MSOM_MG_INL_H
#define MSOM_MG_INL_H

#include "kernels.h"

// Basilisk's bilinear prolongation weight rule.  The product build pins the FMA contraction
// so that every kernel that interpolates (k_prolong, k_relax_red_prolong, k_relax_block) and
// every expansion inside them rounds identically; division by 16 is exact either way.
#ifdef MSOM_STRICT
#define BILINEAR(cn, cfx, cfy, cff) ((9. * (cn) + 3. * ((cfx) + (cfy)) + (cff)) / 16.)
#else
#define BILINEAR(cn, cfx, cfy, cff) (fma(9., (cn), fma(3., (cfx) + (cfy), (cff))) * 0.0625)
#endif

// writes the homogeneous-Dirichlet ghosts that mirror cell (i, j) (edges: -v, corners: +v)
__device__ __forceinline__ void split_write_ghosts(double *f, const SplitGeom &g, int l, int j, int i, double v, int walls) {
  if (walls & WALL_PER) {  // periodic images of an edge cell
    const bool w = i == 0, e = i == g.nx - 1, s = j == 0, n = j == g.ny - 1;
    if (!(w | e | s | n)) return;
    if (w) f[split_idx(g, l, j, g.nx)] = v;
    if (e) f[split_idx(g, l, j, -1)] = v;
    if (s) f[split_idx(g, l, g.ny, i)] = v;
    if (n) f[split_idx(g, l, -1, i)] = v;
    if (w && s) f[split_idx(g, l, g.ny, g.nx)] = v;
    if (w && n) f[split_idx(g, l, -1, g.nx)] = v;
    if (e && s) f[split_idx(g, l, g.ny, -1)] = v;
    if (e && n) f[split_idx(g, l, -1, -1)] = v;
    return;
  }
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

// the same for a field in the natural layout
__device__ __forceinline__ void nat_write_ghosts(double *f, const NatGeom &g, int l, int j, int i, double v, int walls) {
  if (walls & WALL_PER) {
    const bool w = i == 0, e = i == g.nx - 1, s = j == 0, n = j == g.ny - 1;
    if (!(w | e | s | n)) return;
    if (w) f[nat_idx(g, l, j, g.nx)] = v;
    if (e) f[nat_idx(g, l, j, -1)] = v;
    if (s) f[nat_idx(g, l, g.ny, i)] = v;
    if (n) f[nat_idx(g, l, -1, i)] = v;
    if (w && s) f[nat_idx(g, l, g.ny, g.nx)] = v;
    if (w && n) f[nat_idx(g, l, -1, g.nx)] = v;
    if (e && s) f[nat_idx(g, l, g.ny, -1)] = v;
    if (e && n) f[nat_idx(g, l, -1, -1)] = v;
    return;
  }
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

// XCD-aware block numbering.  Workgroups are dealt round-robin over the 8 XCDs (linear ids b and b + 8 share an L2), so
// neighbouring strips of a row-major grid land on 8 different L2s and each of them fetches the cache lines the strips
// share (halo columns, partial 128-B lines at unaligned strip edges).  This renumbering gives every XCD one contiguous
// range of the row-major grid; ids that are dispatched together stay neighbours.  Pure permutation of (bx, by).
__device__ __forceinline__ void xcd_remap(unsigned &bx, unsigned &by) {
  const unsigned nbx = gridDim.x, total = nbx * gridDim.y;
  const unsigned lin = blockIdx.y * nbx + blockIdx.x;
  const unsigned c = lin & 7, q = total >> 3, r = total & 7;
  const unsigned flat = c * q + (c < r ? c : r) + (lin >> 3);
  bx = flat % nbx;
  by = flat / nbx;
}

#endif
