// comm.hip -- halo exchange and scalar reductions between the tiles of the 2-D domain
// decomposition (one process per GPU).  Replaces what Basilisk's MPI layer does inside
// boundary()/boundary_level() and foreach(reduction(...)): 1-cell (or deeper)
// halo exchange with the face (and corner) neighbours, and max/min/sum all-reduces of a few
// doubles.
//
// Transports:
//   RCCL  : grouped ncclSend/ncclRecv per neighbour on the tile's own HIP stream (no host
//           sync), ncclAllReduce for the scalars.  librccl is dlopen()ed so that a single-GPU
//           process never needs it.  xGMI is point-to-point: in a 2 x 4 layout every face
//           neighbour is a direct link, messages are tens of KB (latency-bound), so all
//           layers of one exchange travel in ONE message per neighbour.
//   LOCAL : several tiles in one process (one host thread each, same or different devices),
//           exchanging through device-to-device copies and a host barrier.  This is the test
//           transport: it runs the complete tiling logic on a single GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "comm.h"

// ------------------------------------------------------------------ strip pack / unpack kernels

// natural layout: region columns [i0, i0+w), rows [j0, j0+h), all layers -> contiguous [l][h][w]
__global__ void k_nat_pack_strip(const double *__restrict__ f, NatGeom g, int nl, int i0, int j0, int w, int h, double *buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w * h * nl) return;
  const int i = t % w, j = (t / w) % h, l = t / (w * h);
  buf[t] = f[nat_idx(g, l, j0 + j, i0 + i)];
}
__global__ void k_nat_unpack_strip(double *f, NatGeom g, int nl, int i0, int j0, int w, int h, const double *__restrict__ buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w * h * nl) return;
  const int i = t % w, j = (t / w) % h, l = t / (w * h);
  f[nat_idx(g, l, j0 + j, i0 + i)] = buf[t];
}
__global__ void k_split_pack_strip(const double *__restrict__ f, SplitGeom g, int nl, int i0, int j0, int w, int h, double *buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w * h * nl) return;
  const int i = t % w, j = (t / w) % h, l = t / (w * h);
  buf[t] = f[split_idx(g, l, j0 + j, i0 + i)];
}
__global__ void k_split_unpack_strip(double *f, SplitGeom g, int nl, int i0, int j0, int w, int h, const double *__restrict__ buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w * h * nl) return;
  const int i = t % w, j = (t / w) % h, l = t / (w * h);
  f[split_idx(g, l, j0 + j, i0 + i)] = buf[t];
}
// y-wall BC (homogeneous Dirichlet) on the exchanged x-ghost columns of a split field:
// corner ghost = -(x-ghost next to the wall), cf. k_fill_ghost for natural fields
__global__ void k_split_wall_corners(double *f, SplitGeom g, int nl, int walls) {
  const int l = threadIdx.x;
  if (l >= nl) return;
  if ((walls & WALL_S) && !(walls & WALL_W)) f[split_idx(g, l, -1, -1)] = -f[split_idx(g, l, 0, -1)];
  if ((walls & WALL_S) && !(walls & WALL_E)) f[split_idx(g, l, -1, g.nx)] = -f[split_idx(g, l, 0, g.nx)];
  if ((walls & WALL_N) && !(walls & WALL_W)) f[split_idx(g, l, g.ny, -1)] = -f[split_idx(g, l, g.ny - 1, -1)];
  if ((walls & WALL_N) && !(walls & WALL_E)) f[split_idx(g, l, g.ny, g.nx)] = -f[split_idx(g, l, g.ny - 1, g.nx)];
  // x-wall BC on the exchanged y-ghost rows
  if ((walls & WALL_W) && !(walls & WALL_S)) f[split_idx(g, l, -1, -1)] = -f[split_idx(g, l, -1, 0)];
  if ((walls & WALL_W) && !(walls & WALL_N)) f[split_idx(g, l, g.ny, -1)] = -f[split_idx(g, l, g.ny, 0)];
  if ((walls & WALL_E) && !(walls & WALL_S)) f[split_idx(g, l, -1, g.nx)] = -f[split_idx(g, l, -1, g.nx - 1)];
  if ((walls & WALL_E) && !(walls & WALL_N)) f[split_idx(g, l, g.ny, g.nx)] = -f[split_idx(g, l, g.ny, g.nx - 1)];
}

// all four face strips (1 cell deep) of a split field in ONE launch: W, E columns [l][j], S, N rows [l][i]
// (the layouts of k_split_pack_strip); faces without a neighbour are skipped.  unpack writes the ghosts.
struct FaceBufs { double *b[4]; };  // DIR_W, DIR_E, DIR_S, DIR_N; null = no neighbour
template <bool PACK>
__global__ void k_split_faces(double *f, SplitGeom g, int nl, FaceBufs fb) {
  const int nW = nl * g.ny, nS = nl * g.nx;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int dir, l, i, j;
  if (t < 2 * nW) { dir = t < nW ? DIR_W : DIR_E; t -= dir == DIR_E ? nW : 0; l = t / g.ny; j = t % g.ny; i = dir == DIR_W ? 0 : g.nx - 1; }
  else if (t < 2 * nW + 2 * nS) { t -= 2 * nW; dir = t < nS ? DIR_S : DIR_N; t -= dir == DIR_N ? nS : 0; l = t / g.nx; i = t % g.nx; j = dir == DIR_S ? 0 : g.ny - 1; }
  else return;
  double *b = fb.b[dir];
  if (!b) return;
  if (PACK) b[t] = f[split_idx(g, l, j, i)];
  else {  // ghost cell beyond the face
    const int gi = dir == DIR_W ? -1 : dir == DIR_E ? g.nx : i, gj = dir == DIR_S ? -1 : dir == DIR_N ? g.ny : j;
    f[split_idx(g, l, gj, gi)] = b[t];
  }
}
void launch_split_pack_faces(hipStream_t st, const double *f, const SplitGeom &g, int nl, double *const bufs[4]) {
  FaceBufs fb;
  for (int d = 0; d < 4; d++) fb.b[d] = bufs[d];
  const int n = 2 * nl * (g.nx + g.ny);
  hipLaunchKernelGGL(k_split_faces<true>, dim3((n + 255) / 256), dim3(256), 0, st, const_cast<double *>(f), g, nl, fb);
}
void launch_split_unpack_faces(hipStream_t st, double *f, const SplitGeom &g, int nl, double *const bufs[4]) {
  FaceBufs fb;
  for (int d = 0; d < 4; d++) fb.b[d] = bufs[d];
  const int n = 2 * nl * (g.nx + g.ny);
  hipLaunchKernelGGL(k_split_faces<false>, dim3((n + 255) / 256), dim3(256), 0, st, f, g, nl, fb);
}

static inline dim3 g1(int n) { return dim3((n + 255) / 256); }

// ------------------------------------------------------------------ RCCL (dlopen)

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return MSOM_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    msom_set_error("cannot dlopen librccl: %s", dlerror());
    return MSOM_ERR_COMM;
  }
#define SYM(field, name)                                              \
  *(void **)(&g_rccl.field) = dlsym(h, name);                         \
  if (!g_rccl.field) {                                                \
    msom_set_error("librccl lacks %s", name);                         \
    return MSOM_ERR_COMM;                                             \
  }
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.h = h;
  return MSOM_OK;
}

#define NCCLCHK(x)                                                                               \
  do {                                                                                           \
    ncclResult_t r__ = (x);                                                                      \
    if (r__ != ncclSuccess) {                                                                    \
      msom_set_error("RCCL error %s at %s:%d", g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
      return MSOM_ERR_COMM;                                                                      \
    }                                                                                            \
  } while (0)
#define HIPCHKC(x)                                                                        \
  do {                                                                                    \
    hipError_t e__ = (x);                                                                 \
    if (e__ != hipSuccess) {                                                              \
      msom_set_error("HIP error %s at %s:%d", hipGetErrorString(e__), __FILE__, __LINE__); \
      return MSOM_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

int comm_unique_id(void *id128) {
  int r = rccl_load();
  if (r) return r;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return MSOM_OK;
}

// ------------------------------------------------------------------ LOCAL hub (threads in one process)

struct LocalHub {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0, gen = 0;
  std::vector<const Xfer *> posted;  // per rank: its message list of the current exchange
  std::vector<int> posted_n;
  std::vector<double> red;           // reduction scratch [rank][n]
  std::vector<const double *> gsend; // all-gather: every rank's send buffer
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    int g = gen;
    if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
static std::mutex g_hub_mu;
static std::vector<std::pair<unsigned long long, LocalHub *>> g_hubs;

static LocalHub *hub_get(unsigned long long key, int n) {
  std::lock_guard<std::mutex> lk(g_hub_mu);
  for (auto &h : g_hubs)
    if (h.first == key) return h.second;
  LocalHub *h = new LocalHub();
  h->n = n;
  h->posted.assign(n, nullptr);
  h->posted_n.assign(n, 0);
  h->gsend.assign(n, nullptr);
  g_hubs.push_back({key, h});
  return h;
}

// ------------------------------------------------------------------ Comm

struct Comm {
  int kind = COMM_NONE, rank = 0, n = 1;
  hipStream_t st = nullptr;
  ncclComm_t nccl = nullptr;
  LocalHub *hub = nullptr;
  double *sendbuf[8] = {0}, *recvbuf[8] = {0};
  size_t bufcount = 0;
};

// id128: RCCL: the ncclUniqueId; LOCAL: first 8 bytes "MSOMLOCL", next 8 bytes a hub key
int comm_create(Comm **out, int rank, int n, const void *id128, hipStream_t st, size_t max_count) {
  Comm *c = new Comm();
  c->rank = rank; c->n = n; c->st = st;
  if (n > 1) {
    if (id128 && !memcmp(id128, "MSOMLOCL", 8)) {
      unsigned long long key;
      memcpy(&key, (const char *)id128 + 8, 8);
      c->kind = COMM_LOCAL;
      c->hub = hub_get(key, n);
    } else {
      int r = rccl_load();
      if (r) { delete c; return r; }
      ncclUniqueId id;
      memcpy(&id, id128, sizeof id);
      c->kind = COMM_RCCL;
      NCCLCHK(g_rccl.CommInitRank(&c->nccl, n, id, rank));
    }
    c->bufcount = max_count;
    for (int d = 0; d < 8; d++) {
      HIPCHKC(hipMalloc(&c->sendbuf[d], max_count * sizeof(double)));
      HIPCHKC(hipMalloc(&c->recvbuf[d], max_count * sizeof(double)));
    }
  }
  *out = c;
  return MSOM_OK;
}

void comm_destroy(Comm *c) {
  if (!c) return;
  for (int d = 0; d < 8; d++) {
    if (c->sendbuf[d]) hipFree(c->sendbuf[d]);
    if (c->recvbuf[d]) hipFree(c->recvbuf[d]);
  }
  if (c->nccl) g_rccl.CommDestroy(c->nccl);
  delete c;
}

int comm_kind(const Comm *c) { return c ? c->kind : COMM_NONE; }
double *comm_sendbuf(Comm *c, int dir) { return c->sendbuf[dir]; }
double *comm_recvbuf(Comm *c, int dir) { return c->recvbuf[dir]; }
size_t comm_bufcount(const Comm *c) { return c->bufcount; }

// all messages of one exchange.  Entry k describes the edge in direction tag = d of this tile: its send buffer travels in
// direction d, its receive buffer takes the peer's message travelling in direction OPP[d] (the peer's entry for OPP[d])
int comm_exchange(Comm *c, const Xfer *x, int nx) {
  if (!c || c->kind == COMM_NONE) return MSOM_OK;
  if (c->kind == COMM_RCCL) {
    if (nx == 0) return MSOM_OK;
    // Within a group the sends of one rank to a peer pair up with that peer's receives in posting order.  When both neighbours
    // of an axis are the same rank (periodic tiling, 1 or 2 tiles per side) two messages go to one peer: sends are posted in
    // the order of their direction, receives in the order of the direction the incoming message travels in (the opposite).
    int so[16], ro[16];
    if (nx > 16) return MSOM_ERR_ARG;
    for (int k = 0; k < nx; k++) so[k] = ro[k] = k;
    for (int a = 1; a < nx; a++)
      for (int b = a; b > 0; b--) {
        if (x[so[b]].tag < x[so[b - 1]].tag) { const int t = so[b]; so[b] = so[b - 1]; so[b - 1] = t; }
        if (COMM_OPP[x[ro[b]].tag & 7] < COMM_OPP[x[ro[b - 1]].tag & 7]) { const int t = ro[b]; ro[b] = ro[b - 1]; ro[b - 1] = t; }
      }
    NCCLCHK(g_rccl.GroupStart());
    for (int k = 0; k < nx; k++) NCCLCHK(g_rccl.Send(x[so[k]].send, x[so[k]].count, ncclDouble, x[so[k]].peer, c->nccl, c->st));
    for (int k = 0; k < nx; k++) NCCLCHK(g_rccl.Recv(x[ro[k]].recv, x[ro[k]].count, ncclDouble, x[ro[k]].peer, c->nccl, c->st));
    NCCLCHK(g_rccl.GroupEnd());
    return MSOM_OK;
  }
  // LOCAL: packs are complete once my stream is idle; then everybody copies what it receives
  LocalHub *h = c->hub;
  HIPCHKC(hipStreamSynchronize(c->st));
  h->posted[c->rank] = x;
  h->posted_n[c->rank] = nx;
  h->barrier();
  int err = MSOM_OK;  // an error must not skip the closing barrier: the other tile threads are waiting in it
  for (int k = 0; k < nx && !err; k++) {
    const Xfer *px = h->posted[x[k].peer];
    const int pn = h->posted_n[x[k].peer];
    const Xfer *src = nullptr;
    for (int q = 0; q < pn; q++)
      if (px[q].peer == c->rank && px[q].tag == COMM_OPP[x[k].tag & 7]) src = &px[q];
    if (!src || src->count != x[k].count || x[k].count > c->bufcount) {
      msom_set_error("local exchange: no matching message from rank %d tag %d", x[k].peer, x[k].tag);
      err = MSOM_ERR_COMM;
    } else if (hipMemcpyAsync(x[k].recv, src->send, x[k].count * sizeof(double), hipMemcpyDeviceToDevice, c->st) != hipSuccess) {
      msom_set_error("local exchange: device copy failed");
      err = MSOM_ERR_HIP;
    }
  }
  if (hipStreamSynchronize(c->st) != hipSuccess && !err) err = MSOM_ERR_HIP;
  h->barrier();  // nobody re-packs a send buffer before every reader is done
  return err;
}

// in-place all-reduce of n doubles in device memory; result also copied to host `hout`
int comm_allreduce(Comm *c, double *dvals, double *hout, int n, int op) {
  if (c && c->kind == COMM_RCCL) {
    const ncclRedOp_t o = op == RED_MAX ? ncclMax : op == RED_MIN ? ncclMin : ncclSum;
    NCCLCHK(g_rccl.AllReduce(dvals, dvals, n, ncclDouble, o, c->nccl, c->st));
  }
  HIPCHKC(hipMemcpyAsync(hout, dvals, n * sizeof(double), hipMemcpyDeviceToHost, c ? c->st : nullptr));
  HIPCHKC(hipStreamSynchronize(c ? c->st : nullptr));
  if (c && c->kind == COMM_LOCAL) {
    LocalHub *h = c->hub;
    {
      std::lock_guard<std::mutex> lk(h->mu);
      if (h->red.size() < (size_t)h->n * n) h->red.resize((size_t)h->n * n);
    }
    h->barrier();
    for (int k = 0; k < n; k++) h->red[(size_t)c->rank * n + k] = hout[k];
    h->barrier();
    for (int k = 0; k < n; k++) {
      double v = h->red[k];
      for (int r = 1; r < h->n; r++) {  // fixed rank order: every tile computes the same value
        const double w = h->red[(size_t)r * n + k];
        v = op == RED_MAX ? (w > v ? w : v) : op == RED_MIN ? (w < v ? w : v) : v + w;
      }
      hout[k] = v;
    }
    h->barrier();
    HIPCHKC(hipMemcpyAsync(dvals, hout, n * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHKC(hipStreamSynchronize(c->st));
  }
  return MSOM_OK;
}

// recv[r * count .. (r+1) * count) = rank r's send[0 .. count)   (device buffers)
int comm_allgather(Comm *c, const double *send, double *recv, size_t count) {
  if (!c || c->kind == COMM_NONE) return MSOM_OK;
  if (c->kind == COMM_RCCL) {
    NCCLCHK(g_rccl.AllGather(send, recv, count, ncclDouble, c->nccl, c->st));
    return MSOM_OK;
  }
  LocalHub *h = c->hub;
  HIPCHKC(hipStreamSynchronize(c->st));
  h->gsend[c->rank] = send;
  h->barrier();
  for (int r = 0; r < h->n; r++)
    HIPCHKC(hipMemcpyAsync(recv + (size_t)r * count, h->gsend[r], count * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  HIPCHKC(hipStreamSynchronize(c->st));
  h->barrier();
  return MSOM_OK;
}

// assemble the gathered tile interiors [rank][l][y][x] into a global split-layout field
__global__ void k_assemble_global(const double *__restrict__ recv, double *g, SplitGeom gg, int nl, int tnx, int tny, int px) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x, gj = blockIdx.y;
  if (gi >= gg.nx || gj >= gg.ny) return;
  const int r = (gj / tny) * px + gi / tnx, i = gi % tnx, j = gj % tny;
  const size_t tile = (size_t)nl * tnx * tny;
  for (int l = 0; l < nl; l++) g[split_idx(gg, l, gj, gi)] = recv[(size_t)r * tile + ((size_t)l * tny + j) * tnx + i];
}
void launch_assemble_global(hipStream_t st, const double *recv, double *g, const SplitGeom &gg, int nl, int tnx, int tny, int px) {
  hipLaunchKernelGGL(k_assemble_global, dim3((gg.nx + 63) / 64, gg.ny), dim3(64), 0, st, recv, g, gg, nl, tnx, tny, px);
}
// tile (ix, iy) of a global split field, with its ghost ring (walls or neighbours' cells), -> tile field
__global__ void k_extract_tile(const double *__restrict__ g, SplitGeom gg, double *t, SplitGeom tg, int nl, int ox, int oy) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)blockIdx.y - 1;
  if (i > tg.nx || j > tg.ny) return;
  for (int l = 0; l < nl; l++) t[split_idx(tg, l, j, i)] = g[split_idx(gg, l, oy + j, ox + i)];
}
void launch_extract_tile(hipStream_t st, const double *g, const SplitGeom &gg, double *t, const SplitGeom &tg, int nl, int ox, int oy) {
  hipLaunchKernelGGL(k_extract_tile, dim3((tg.nx + 2 + 63) / 64, tg.ny + 2), dim3(64), 0, st, g, gg, t, tg, nl, ox, oy);
}

// ------------------------------------------------------------------ strip launchers

void launch_nat_pack_strip(hipStream_t st, const double *f, const NatGeom &g, int nl, int i0, int j0, int w, int h, double *buf) {
  hipLaunchKernelGGL(k_nat_pack_strip, g1(w * h * nl), dim3(256), 0, st, f, g, nl, i0, j0, w, h, buf);
}
void launch_nat_unpack_strip(hipStream_t st, double *f, const NatGeom &g, int nl, int i0, int j0, int w, int h, const double *buf) {
  hipLaunchKernelGGL(k_nat_unpack_strip, g1(w * h * nl), dim3(256), 0, st, f, g, nl, i0, j0, w, h, buf);
}
void launch_split_pack_strip(hipStream_t st, const double *f, const SplitGeom &g, int nl, int i0, int j0, int w, int h, double *buf) {
  hipLaunchKernelGGL(k_split_pack_strip, g1(w * h * nl), dim3(256), 0, st, f, g, nl, i0, j0, w, h, buf);
}
void launch_split_unpack_strip(hipStream_t st, double *f, const SplitGeom &g, int nl, int i0, int j0, int w, int h, const double *buf) {
  hipLaunchKernelGGL(k_split_unpack_strip, g1(w * h * nl), dim3(256), 0, st, f, g, nl, i0, j0, w, h, buf);
}
// Doubly periodic domain on ONE tile (sbc = -1, msqg/qg.h:842-846): the deep halo of the chained smoother is the field's own
// other side.  phase 0: H cells beyond the W / E edges into the row pads (all rows -1 .. ny); phase 1: H rows beyond the
// S / N edges into the halo arrays fs / fn (geometry hg: H rows), pad columns included, so the corner regions follow;
// phase 2: the depth-1 ghost rows / columns of the field itself with their corners (what boundary_level() leaves).
__global__ void k_split_wrap(double *f, SplitGeom g, double *fs, double *fn, SplitGeom hg, int nl, int H, int phase) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (phase == 0) {
    const int rows = g.ny + 2;
    if (t >= rows * H) return;
    const int j = t / H - 1, c = t % H;
    f[split_idx(g, l, j, -H + c)] = f[split_idx(g, l, j, g.nx - H + c)];
    f[split_idx(g, l, j, g.nx + c)] = f[split_idx(g, l, j, c)];
  } else if (phase == 1) {
    const int xw = g.nx + 2 * H;
    if (t >= xw * H) return;
    const int r = t / xw, i = t % xw - H;
    fs[split_idx(hg, l, r, i)] = f[split_idx(g, l, g.ny - H + r, i)];
    fn[split_idx(hg, l, r, i)] = f[split_idx(g, l, r, i)];
  } else {
    const int nx = g.nx, ny = g.ny;
    if (t < ny) {            // ghost columns of the interior rows
      f[split_idx(g, l, t, -1)] = f[split_idx(g, l, t, nx - 1)];
      f[split_idx(g, l, t, nx)] = f[split_idx(g, l, t, 0)];
    } else if (t < ny + nx + 2) {   // ghost rows, corners included (the source columns -1 / nx wrap as well)
      const int i = t - ny - 1, si = i < 0 ? nx - 1 : (i >= nx ? 0 : i);
      f[split_idx(g, l, -1, i)] = f[split_idx(g, l, ny - 1, si)];
      f[split_idx(g, l, ny, i)] = f[split_idx(g, l, 0, si)];
    }
  }
}
void launch_split_wrap(hipStream_t st, double *f, const SplitGeom &g, double *fs, double *fn, const SplitGeom &hg, int nl, int H, int phase) {
  const int n = phase == 0 ? (g.ny + 2) * H : (phase == 1 ? (g.nx + 2 * H) * H : g.ny + g.nx + 2);
  hipLaunchKernelGGL(k_split_wrap, dim3((n + 255) / 256, nl), dim3(256), 0, st, f, g, fs, fn, hg, nl, H, phase);
}
void launch_split_wall_corners(hipStream_t st, double *f, const SplitGeom &g, int nl, int walls) {
  hipLaunchKernelGGL(k_split_wall_corners, dim3(1), dim3(64), 0, st, f, g, nl, walls);
}

// ------------------------------------------------------------------ RCCL wiring self-test
// One-rank communicator on the current device: grouped ncclSend/ncclRecv to self (two messages,
// as one exchange posts them), max/sum all-reduce and all-gather, through exactly the code paths
// above.  Checks the dlopen()ed entry points, enum values and stream ordering on a single GPU,
// where a 2-rank communicator cannot be formed.
extern "C" int msom_dbg_rccl_selftest(void) {
  int r = rccl_load();
  if (r) return r;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  // same stream set-up as a tiled model: compute stream + high-priority non-blocking communication stream,
  // ordered by event pairs; all transport calls on the communication stream
  hipStream_t st, st2;
  hipEvent_t e1, e2;
  int lo = 0, hi = 0;
  HIPCHKC(hipStreamCreate(&st));
  HIPCHKC(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIPCHKC(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, hi));
  HIPCHKC(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  HIPCHKC(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  Comm c;
  c.kind = COMM_RCCL; c.rank = 0; c.n = 1; c.st = st2;
  NCCLCHK(g_rccl.CommInitRank(&c.nccl, 1, id, 0));
  const int n = 1000;
  double *d = nullptr, *busy = nullptr;
  const size_t nbusy = (size_t)32 << 20;
  HIPCHKC(hipMalloc(&d, 6 * n * sizeof(double)));
  HIPCHKC(hipMalloc(&busy, nbusy * sizeof(double)));
  std::vector<double> h(6 * n, 0.);
  for (int k = 0; k < 2 * n; k++) h[k] = 1.5 * k - 7.;
  int bad = 0;
  for (int rep = 0; rep < 3 && !bad; rep++) {
    HIPCHKC(hipMemcpyAsync(d, h.data(), 6 * n * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHKC(hipEventRecord(e1, st));              // comm_begin
    HIPCHKC(hipStreamWaitEvent(st2, e1, 0));
    Xfer x[2] = {{0, DIR_W, d, d + 2 * n, (size_t)n}, {0, DIR_E, d + n, d + 3 * n, (size_t)n}};
    if ((r = comm_exchange(&c, x, 2))) return r;
    HIPCHKC(hipMemsetAsync(busy, rep, nbusy * sizeof(double), st));  // "interior" work beside the exchange
    if ((r = comm_allgather(&c, d, d + 4 * n, n))) return r;
    HIPCHKC(hipEventRecord(e2, st2));             // comm_end
    HIPCHKC(hipStreamWaitEvent(st, e2, 0));
    double hs[2];
    HIPCHKC(hipEventRecord(e1, st));
    HIPCHKC(hipStreamWaitEvent(st2, e1, 0));
    if ((r = comm_allreduce(&c, d + 1, hs, 2, RED_MAX))) return r;
    if ((r = comm_allreduce(&c, d + 1, hs, 2, RED_SUM))) return r;
    HIPCHKC(hipEventRecord(e2, st2));
    HIPCHKC(hipStreamWaitEvent(st, e2, 0));
    std::vector<double> out(6 * n, 0.);
    HIPCHKC(hipMemcpyAsync(out.data(), d, 6 * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHKC(hipStreamSynchronize(st));
    // a tile that is its own W and E neighbour (periodic, one tile per side): the W edge receives what the E edge sent and
    // vice versa -- the pairing by direction of travel and posting order of comm_exchange, on the real library
    for (int k = 0; k < n; k++) bad += out[2 * n + k] != 1.5 * (n + k) - 7. || out[3 * n + k] != 1.5 * k - 7.;
    for (int k = 0; k < n; k++) bad += out[4 * n + k] != 1.5 * k - 7.;
    bad += hs[0] != 1.5 * 1 - 7. || hs[1] != 1.5 * 2 - 7.;
  }
  g_rccl.CommDestroy(c.nccl);
  c.nccl = nullptr;
  (void)hipFree(d);
  (void)hipFree(busy);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  (void)hipStreamDestroy(st2);
  (void)hipStreamDestroy(st);
  if (bad) { msom_set_error("RCCL self-test: %d wrong values", bad); return MSOM_ERR_COMM; }
  return MSOM_OK;
}
