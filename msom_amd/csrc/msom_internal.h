// msom_internal.h -- internal types of libmsomhip (not part of the C ABI).
//
// Data layout in HBM:
//  * "natural" fields (psi, q, zeta, dq, ...): fp64 [layer][row][col], x fastest, one
//    allocation per field list.  Every row has a 16-double (128 B) pad on each side so that
//    interior column 0 is 128-B aligned; 3 pad rows below and above.  The first pad
//    column/row next to the interior holds the ghost cells of Basilisk's boundary().
//  * multigrid-internal fields (correction da, residual res, stretching S on every level)
//    use an x-parity split layout: each row is stored as [even-x half | odd-x half], so the
//    points of one red-black colour are contiguous in every row and a colour half-sweep
//    streams exactly the bytes it needs (coalesced), with no LDS transpose.
#ifndef MSOM_INTERNAL_H
#define MSOM_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/msom.h"
#include "msom_params.h"

#define MSOM_XP 16  // pad columns (doubles) on each side of a natural row
#define MSOM_YP 3   // pad rows below/above a natural layer
#define MSOM_SP 16  // pad columns on each side of a split half-row

enum { BC_DIRICHLET0 = 0, BC_NEUMANN = 1, BC_PERIODIC = 2, BC_DIRICHLET_LIN = 3 };
// WALL_PER: doubly periodic domain (sbc = -1): ghost cells are wrapped copies kept up to date by
// the thread that owns the source cell
enum { WALL_W = 1, WALL_E = 2, WALL_S = 4, WALL_N = 8, WALL_ALL = 15, WALL_PER = 16 };

struct NatGeom {
  int nx, ny;     // interior cells of this tile
  int pitch;      // doubles per row  (nx + 2*XP)
  int rows;       // rows per layer   (ny + 2*YP)
  size_t ls;      // doubles per layer
};

struct SplitGeom {
  int nx, ny;     // interior cells of this level (tile-local)
  int hk;         // cells per half row = nx/2
  int hp;         // doubles per half row incl. pads (hk + 2*SP)
  int rp;         // doubles per row = 2*hp
  int rows;       // ny + 2
  size_t ls;      // doubles per layer
};

#define MSOM_HD __host__ __device__ __forceinline__

MSOM_HD size_t nat_idx(const NatGeom &g, int l, int j, int i) {
  return (size_t)l * g.ls + (size_t)(j + MSOM_YP) * g.pitch + (size_t)(i + MSOM_XP);
}
// (i >> 1) is an arithmetic shift: ghost column i = -1 lives at index -1 of the odd half,
// ghost column i = nx at index hk of the even half.
MSOM_HD size_t split_idx(const SplitGeom &g, int l, int j, int i) {
  return (size_t)l * g.ls + (size_t)(j + 1) * g.rp + (size_t)((i & 1) * g.hp + MSOM_SP + (i >> 1));
}

#endif
