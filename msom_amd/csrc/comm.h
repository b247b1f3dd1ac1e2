// comm.h -- tile-to-tile transport of libmsomhip (internal).  See comm.hip.
#ifndef MSOM_COMM_H
#define MSOM_COMM_H

#include <string.h>

#include "msom_internal.h"

enum { COMM_NONE = 0, COMM_RCCL = 1, COMM_LOCAL = 2 };
enum { RED_MAX = 0, RED_MIN = 1, RED_SUM = 2 };
// directions of the 8 neighbours (buffer slots)
enum { DIR_W = 0, DIR_E = 1, DIR_S = 2, DIR_N = 3, DIR_SW = 4, DIR_SE = 5, DIR_NW = 6, DIR_NE = 7 };

// directions W E S N SW SE NW NE and their opposites
static const int COMM_OPP[8] = {1, 0, 3, 2, 7, 6, 5, 4};
struct Xfer {
  int peer;        // rank of the neighbour
  int tag;         // direction (DIR_*) of this tile's edge = direction the sent message travels in
  double *send, *recv;
  size_t count;    // doubles (same in both directions)
};

struct Comm;
int comm_unique_id(void *id128);
int comm_create(Comm **out, int rank, int n, const void *id128, hipStream_t st, size_t max_count);
void comm_destroy(Comm *c);
int comm_kind(const Comm *c);
double *comm_sendbuf(Comm *c, int dir);
double *comm_recvbuf(Comm *c, int dir);
size_t comm_bufcount(const Comm *c);
int comm_exchange(Comm *c, const Xfer *x, int nx);
int comm_allreduce(Comm *c, double *dvals, double *hout, int n, int op);
int comm_allgather(Comm *c, const double *send, double *recv, size_t count);
void launch_assemble_global(hipStream_t st, const double *recv, double *g, const SplitGeom &gg, int nl, int tnx, int tny, int px);
void launch_extract_tile(hipStream_t st, const double *g, const SplitGeom &gg, double *t, const SplitGeom &tg, int nl, int ox, int oy);

void launch_nat_pack_strip(hipStream_t st, const double *f, const NatGeom &g, int nl, int i0, int j0, int w, int h, double *buf);
void launch_nat_unpack_strip(hipStream_t st, double *f, const NatGeom &g, int nl, int i0, int j0, int w, int h, const double *buf);
void launch_split_pack_strip(hipStream_t st, const double *f, const SplitGeom &g, int nl, int i0, int j0, int w, int h, double *buf);
void launch_split_unpack_strip(hipStream_t st, double *f, const SplitGeom &g, int nl, int i0, int j0, int w, int h, const double *buf);
void launch_split_pack_faces(hipStream_t st, const double *f, const SplitGeom &g, int nl, double *const bufs[4]);
void launch_split_unpack_faces(hipStream_t st, double *f, const SplitGeom &g, int nl, double *const bufs[4]);
void launch_split_wall_corners(hipStream_t st, double *f, const SplitGeom &g, int nl, int walls);
void launch_split_wrap(hipStream_t st, double *f, const SplitGeom &g, double *fs, double *fn, const SplitGeom &hg, int nl, int H, int phase);

#endif
