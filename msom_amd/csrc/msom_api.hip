// msom_api.hip -- C ABI of libmsomhip: lifecycle, fields, elliptic solver driver, RHS
// evaluation and the predictor-corrector time loop.  Kernels live in kernels_rhs.hip and
// kernels_mg.hip; params.in parsing and .bas IO in params.c / bas_io.c (host C).
//
// Reference call stacks restated here (file:line in the reference tree):
//   set_vars msqg/qg.h:837-925, set_const :931-1116, invertq :114-163,
//   poisson_layer msqg/poisson_layer.h:263-306, mg_solve/mg_cycle mspg/elliptic.h:43-99,145-229,
//   update_qg msqg/qg.h:609-650, advance_qg :594-606, run() of Basilisk predictor-corrector.h
//   (SURVEY App. B), writestdout/output events msqg/qg.c:101-173, pystep_bfn msqg/qg_bfn.h:21-103.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>

#include <algorithm>
#include <string>
#include <utility>
#include <functional>
#include <map>
#include <vector>

#include "comm.h"
#include "kernels.h"

#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      msom_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e__), __FILE__, __LINE__, #x); \
      return MSOM_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

// device scalar slots
// RES0, RES1 and UMAX[nl] are contiguous: one max all-reduce / one copy brings them to the host
enum { SC_BSUM = 2, SC_KE = 3, SC_SCRATCH = 4, SC_RES0 = 6, SC_RES1 = 7, SC_UMAX = 8 /* MAXNL */, SC_RESF = 8 + MSOM_MAXNL /* max|res| from the fused tendency pass */, SC_LSUM = 32 /* MAXNL */, SC_SPEC = 48 /* k_step_dt: dt limit, dt, tnext, dt / 2 */, SC_COUNT = 64 };
static_assert(SC_RESF < SC_LSUM && SC_LSUM + MSOM_MAXNL <= SC_SPEC && SC_SPEC + 4 <= SC_COUNT, "scalar slots overlap");

static int g_dbg_interleave = 0;  // timing experiment of msom_bench_kernel (march passes)
static int g_march_rows = 0;  // tuning knob: chunk height of k_relax_march (0 = automatic)
#define MARCH_HALO 4    // rows (= cells in x) of neighbour data a pass of up to 4 chained half-sweeps reads
struct ProfSlot {
  std::vector<hipEvent_t> ev;  // pairs (start, stop)
  size_t used = 0;
  double total_ms = 0;
  long launches = 0;
};

struct msom {
  Params p;
  std::string params_text;  // raw params.in text (backup copy in the output directory)
  // decomposition (single tile unless created with msom_create_tiled)
  int px = 1, py = 1, ix = 0, iy = 0, rank = 0, nranks = 1;
  int gnx = 0, gny = 0;  // global cells
  int nx = 0, ny = 0;    // local cells
  int walls = WALL_ALL;
  Comm *comm = nullptr;
  int sticky = MSOM_OK;  // first error of a void helper (exchange inside fill_bc / mg_cycle)
  int comm_hold = 0;     // see comm_begin
  int dbg_nosync = 0;    // option dbg_nosync (timing experiment)
  // Speculative tendency pass (round 3, one tile): right after the first multigrid cycle of a solve the tendency kernel is queued
  // BEFORE the host has read max|res| and max|u| -- with dt computed on the device (k_step_dt) in the first RK stage -- so the
  // GPU works through the host round trip instead of idling (0.49 -> 0.43 ms per step at 256^2 x 3 with the read skipped
  // altogether, tools/ab_nosync.py).  If the solve turns out to need another cycle the pass is simply run again afterwards:
  // its output went to the predictor (stage 1) or to a spare buffer that only replaces q when the solve had converged (stage 2).
  int async_solve = 1;               // option
  int step_sync = -1;                // option: 0 = msom_step returns without waiting for its last tendency pass (async_solve only), 1 = it waits,
                                     // -1 (default) = it waits on grids of >= 2^23 cell-layers.  Measured (same process): 128^2 x 1 0.203 -> 0.185 ms per
                                     // step, 512^2 x 3 0.374 -> 0.352, 2048^2 x 3 1.238 -> 1.226, 4096^2 x 6 6.15 -> 6.20 (worse)
  std::function<void(const std::function<void()> &)> spec_hook;   // queued by mg_solve after the first cycle's residual pass; calls its
                                     // argument (the host's read of the scalars) between the dt kernel and the tendency pass
  int spec_launched = 0, spec_valid = 0;
  hipEvent_t ev_spec = nullptr;
  double *h_pub = nullptr, *d_pub = nullptr;   // coherent pinned host block the device publishes the scalars into (+ a sequence word), and its device address
  long pub_seq = 0;
  double *q_alt = nullptr;           // stage 2 writes q + dt dq here
  double *adv_out_override = nullptr;
  const double *dt_dev = nullptr;
  int nb[8];  // neighbour ranks by direction (DIR_*), -1 = none
  int nl = 1, nlm = 1;
  int bc = BC_DIRICHLET0;
  // layer metrics (msqg/qg.h:1017-1027)
  double dhf[MSOM_MAXARR], dhc[MSOM_MAXARR];
  LayerCoef lc;
  // natural fields
  NatGeom g;
  double *f[MSOM_NFIELDS];
  int flayers[MSOM_NFIELDS];
  int fbc[MSOM_NFIELDS];
  // multigrid hierarchy (level 0 = finest)
  int nlev = 0;
  std::vector<SplitGeom> sg;
  std::vector<double *> da, da_alt, res, S;
  // chained smoother on tiles: the MARCH_HALO nearest rows of the S / N neighbour tiles (correction, residual), per level
  std::vector<double *> mh_da_s, mh_da_n, mh_res_s, mh_res_n;
  std::vector<RelaxCoef> rc;
  size_t max_split = 0;
  // agglomerated coarse levels (tiled mode): levels >= agg_level live on the gathered global grid
  int agg_level = -1, agglomerate = 1, agg_size = 256;
  int mg_global_sum = 0;
  std::vector<SplitGeom> gsg;
  std::vector<double *> gda, gda_alt, gres;
  double *agg_send = nullptr, *agg_recv = nullptr;
  // scratch
  double *staging = nullptr;   // contiguous nl*ny*nx
  double *partial = nullptr;   // per-block partial sums
  double *partial_rr = nullptr;  // per-workgroup sums of q_out from the fused tendency pass (consumed by the next mg_solve)
  double *partial_umax = nullptr;  // per-block partial maxima of k_umax
  double *d_scal = nullptr, *h_scal = nullptr;
  double *d_wind = nullptr;    // per-row surface forcing profile
  double umax_pg[MSOM_MAXNL];
  int umax_ready = 0;  // h_scal[SC_UMAX..] holds max|u| of the current psi (from the last solve)
  hipStream_t st = nullptr;
  // tiled runs: every transport call (RCCL / LOCAL), with its pack and unpack kernels, is issued on the
  // communication stream st2 (higher priority); event pairs order it against the compute stream st, so a
  // halo exchange can run beside an interior kernel launched in between (option "overlap")
  hipStream_t st2 = nullptr;
  hipEvent_t ev_c2x = nullptr, ev_x2c = nullptr;
  int overlap = 1;
  // flags
  int const_set = 0, flag_topo = 0, have_pg = 0, have_zpg = 0, have_qforc = 0;
  int fr_uniform = 1, uniformS = 0, uniform_opt = -1 /* auto */;
  int stochastic = 0, corrector_step = 0, noise_mode = 0, stoch_fused = 1;
  int prolong_fused = 1;  // first red half-sweep of a level interpolates its neighbours from the coarser level
  int block_sweeps = 0;  // experimental temporally blocked smoother (2 sweeps per pass); measured not faster at nl = 6
  int restrict_pyr = 1;  // round 3: the restriction chain below that in launches of up to 5 levels (k_restrict_pyramid)
  int restrict2 = 1;     // round 3: the pre-cycle residual pass restricts two levels down (k_residual2, res_c2)
  int block8 = 1;        // round 3: prolongation + up to 8 half-sweeps of a launch-bound level in one launch (k_relax_block, halo 8)
  int block8_max = 1024; // ... on levels of at most this many cells a side (and not marched)
  int block_small = 0;   // the same kernel on the launch-bound levels only (not marched, <= block_small cells wide): 2 launches per level visit instead of 8
  int march = 1;         // chained half-sweeps in register windows (kernels_march.hip) on wide single-GPU levels
  int march_k = 4;       // at most this many half-sweeps per pass (2..4)
  // launch-bound grids (no level takes the marching or the blocked smoother, single tile): the launches of one multigrid cycle
  // are captured once per (nrelax, first_restrict) into a hipGraph and replayed -- the host then issues 1 launch instead of ~45
  int use_graph = 0;   // measured: 512^2 x 3 0.645 vs 0.647 ms per step, 1024^2 x 3 0.860 vs 0.879, 128^2 x 1 0.316 vs 0.305 -- these grids are
                       // bound by the duration of their ~5-us kernels on the GPU, not by the host's launch rate: option "graph", off
  std::map<long, hipGraphExec_t> cyc_graph;
  int march_partial = 1; // a pass that is followed by more half-sweeps stores only the colour of its last half-sweep
  int march_min_tiled = 22;  // the same threshold on tiles (see march_ok)
  int march_min = 23;    // log2 of the cell-layers a level needs for the chained pass (2^23: 2048^2 x 3 1.83 -> 1.78 ms/step, and the 2048 x 1024 x 6 tiles of BASELINE's 2 x 4 layout qualify; 2^22 loses: 1024^2 x 6 2.76 -> 2.87)
  int march_correct = 1; // the last pass of the finest level writes psi + da instead of da (psi rows by LDS-DMA, deferred write): 7.02 -> 6.86 ms per step at 4096^2 x 6
  int corr_req = 0, corr_done = 0;  // set around mg_cycle_levels by mg_solve / by the pass that did it
  int march_prolong = 1; // whole levels: prolongation folded into the first pass ((PL + 4) + 4 half-sweeps; coarse rows by LDS-DMA, kernels_march.hip): 7.63 -> 7.09 ms per step at 4096^2 x 6
  int mg_fused = 1;  // fused residual+restriction and correction+residual passes of the multigrid cycle
  double *psi_alt = nullptr;  // second psi buffer (the fused correction writes out of place)
  int rhs_variant = 6;  // 6: one layer per wavefront, register windows (kernels_lpw.hip, default); 1: LDS tiles, software-pipelined; 0: LDS tiles, phase by phase
  int fused = 1;  // one-pass PV tendency kernel (kernels_fused.hip) when the configuration allows
  unsigned seed = 1, noise_draw = 0;
  int quiet = 0;
  // wavelet scale filter (msqg/qg.h:509-560): pyramids s (restricted psi), r (filtered), sig_lev
  int wv_nlev = 0, wv_ready = 0;
  // tiles: levels 0 .. wv_kt live on the tile (halo exchange per level), the levels above on a gathered top grid that every
  // rank transforms on the host (a handful of cells); wv_top_sig[k - wv_kt]: sig_lev of the gathered levels
  int wv_kt = 0, wv_gx = 0, wv_gy = 0;
  std::vector<std::vector<double>> wv_top_sig;
  double *wv_gsend = nullptr, *wv_grecv = nullptr;
  int nme_ft = 0;  // msqg/qg_energy.h:17
  // coarse levels (<= MGC_MAXDIM cells a side) solved by ONE launch (k_mg_coarse)
  CoarseArgs *d_cargs = nullptr;
  int mgc_dim = MGC_MAXDIM;  // widest level of that group
  int mgc_pfused = 1;   // option: prolongation fused into the first red phase inside k_mg_coarse (nl <= 4)
  int umax_clean = 0;  // the max|u| accumulators were zeroed with the solve's scalars and not used since
  int mgc_first = -1, mgc_opt = 4;  // first (finest) level of the group, -1: none; option "mg_coarse" (1: through global memory, 2: levels resident in LDS,
                                    // 3: as 2 with a NaN-filled pool, 4: the lean LDS form k_mg_coarse_lean where it applies, else 2)
  int mgc_lean = 0;
  int res_ready = -1;  // field id whose first multigrid residual (levels 0, 1; SC_RESF; partial sums) the last tendency pass already produced
  int adv_fused = 1;   // fold q_out = q_in + dt dq into the tendency pass
  int rhs_resid = 0;   // let the fused tendency + advance pass produce it: measured slower (23 spilled VGPRs in the 256-VGPR kernel: 2.21 ms vs 1.63 + 0.50 ms), kept as an option
  int s_zero = 0;  // pystep_de(onlyKE = 1) zeroed the stretching field S (msqg/qg_energy.h:319-325); undone by msom_set_const
  std::vector<NatGeom> wv_g;
  std::vector<double *> wv_s, wv_r, wv_sig;
  // time loop
  double t = 0, dt = 1., tnext = HUGE_VAL, previous = 0;
  int iter = 0;
  msom_mgstats mg = {0, 0, 0, 0, 0};
  // profiling of the finest-level smoother sweep
  int profile = 0;
  ProfSlot prof_sweep, prof_resid, prof_block, prof_march[5];  // prof_march[K]: passes of K chained half-sweeps
  ProfSlot prof_march_pl;  // first pass of a level with the prolongation folded in
  ProfSlot prof_march_corr, prof_resmax;  // last pass of the finest level with the correction folded in; max-only residual pass after it
  ProfSlot prof_rhs, prof_redprol, prof_rescorr, prof_respre;   // tendency pass, finest red+prolongation, post- / pre-cycle residual passes
};

static void free_agglomeration(msom *m);

extern "C" const char *msom_version(void) {
#ifdef MSOM_STRICT
  return "msomhip 0.1 (strict: -ffp-contract=off, reference expression order, bit-exact vs oracle)";
#else
  return "msomhip 0.1 (fast: FMA contraction, reciprocal multiplies)";
#endif
}

// ------------------------------------------------------------------ helpers

static NatGeom make_nat(int nx, int ny) {
  NatGeom g;
  g.nx = nx; g.ny = ny;
  g.pitch = ((nx + 15) / 16) * 16 + 2 * MSOM_XP;
  g.rows = ny + 2 * MSOM_YP;
  g.ls = (size_t)g.pitch * g.rows;
  return g;
}
static SplitGeom make_split(int nx, int ny) {
  SplitGeom s;
  s.nx = nx; s.ny = ny;
  s.hk = nx / 2;
  s.hp = ((s.hk + 15) / 16) * 16 + 2 * MSOM_SP;
  s.rp = 2 * s.hp;
  s.rows = ny + 2;
  s.ls = (size_t)s.rp * s.rows;
  return s;
}

static int sync_stream(msom *m) {
  HIPCHK(hipStreamSynchronize(m->st));
  return m->sticky;
}
#define STICKY(m, call)                                  \
  do {                                                   \
    int r__ = (call);                                    \
    if (r__ && !(m)->sticky) (m)->sticky = r__;          \
  } while (0)

// profile = 1: every slot; 2: only the chained smoother passes (the dominant kernel of the bench line) -- an event pair costs
// ~10 us of stream time, 1.4 % of a 4096^2 x 6 step and 20 % of a 512^2 x 3 step with every slot on
static bool prof_on(const msom *m, const ProfSlot &ps) {
  if (m->profile == 2) return &ps == &m->prof_march_corr || (&ps >= &m->prof_march[0] && &ps <= &m->prof_march[4]);
  return m->profile != 0;
}
static void prof_begin(msom *m, ProfSlot &ps) {
  if (!prof_on(m, ps)) return;
  if (ps.used + 2 > ps.ev.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    ps.ev.push_back(a); ps.ev.push_back(b);
  }
  hipEventRecord(ps.ev[ps.used], m->st);
}
static void prof_end(msom *m, ProfSlot &ps) {
  if (!prof_on(m, ps)) return;
  hipEventRecord(ps.ev[ps.used + 1], m->st);
  ps.used += 2;
}
static void prof_collect(msom *m, ProfSlot &ps) {
  hipStreamSynchronize(m->st);
  for (size_t k = 0; k + 1 < ps.used; k += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ps.ev[k], ps.ev[k + 1]) == hipSuccess) { ps.total_ms += ms; ps.launches++; }
  }
  ps.used = 0;
}

static double wall_seconds() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }


// ------------------------------------------------------------------ tile halo exchange / reductions

static const int OPP[8] = {DIR_E, DIR_W, DIR_N, DIR_S, DIR_NE, DIR_NW, DIR_SE, DIR_SW};
// both ends of a neighbour pair label their message with the same axis id
// a message is labelled with the direction it travels in (the sender's); the receiving end of direction d pairs with the peer's
// message of direction OPP[d] (comm.hip).  On a periodic tiling with 1 or 2 tiles per side both neighbours of an axis are the
// same rank (or the tile itself): the direction tells the two messages apart.
#define AXIS(dir) (dir)

// communication stream <-> compute stream ordering (tiled runs)
// comm_hold: the caller has opened the window itself (comm_begin), queues independent kernels on the compute stream while
// the exchanges it calls run on the communication stream, and closes it (comm_end) before the kernels that need the halos
static void comm_begin(msom *m) {  // st2 continues after everything queued on st so far
  if (m->comm_hold) return;
  hipEventRecord(m->ev_c2x, m->st);
  hipStreamWaitEvent(m->st2, m->ev_c2x, 0);
}
static void comm_end(msom *m) {    // st continues after everything queued on st2 so far
  if (m->comm_hold) return;
  hipEventRecord(m->ev_x2c, m->st2);
  hipStreamWaitEvent(m->st, m->ev_x2c, 0);
}

// all-reduce n device scalars starting at `slot` over the tiles; result in h_scal (and d_scal)
static int reduce_scal(msom *m, int slot, int n, int op) {
  if (m->nranks > 1) {
    comm_begin(m);
    const int r = comm_allreduce(m->comm, m->d_scal + slot, m->h_scal + slot, n, op);
    comm_end(m);
    return r;
  }
  if (m->dbg_nosync) return MSOM_OK;   // timing experiment: the host keeps the numbers of the last real read (results meaningless)
  HIPCHK(hipMemcpyAsync(m->h_scal + slot, m->d_scal + slot, n * sizeof(double), hipMemcpyDeviceToHost, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}

// every message of an exchange must fit the per-direction staging buffers of the communicator
static bool fits_comm(msom *m, size_t count) {
  if (count <= comm_bufcount(m->comm)) return true;
  msom_set_error("halo message of %zu doubles exceeds the communication buffers (%zu)", count, comm_bufcount(m->comm));
  return false;
}

// halo exchange of a natural field, `depth` ghost columns/rows, corners included: two phases
// (x, then y over the x-ghost columns) with the wall BCs applied in between, exactly the
// order of Basilisk's boundary() (x direction first).
static int exch_nat_g(msom *m, double *f, const NatGeom &g, int nl, int bc, int depth);
static int exch_nat(msom *m, double *f, int nl, int bc, int depth) { return exch_nat_g(m, f, m->g, nl, bc, depth); }
// the same on any level of a natural-layout pyramid of the tile (wavelet filter)
static int exch_nat_g(msom *m, double *f, const NatGeom &g, int nl, int bc, int depth) {
  const int d = depth;
  if (m->nranks > 1 && !fits_comm(m, (size_t)d * ((g.nx > g.ny ? g.nx : g.ny) + 2 * d) * nl)) return MSOM_ERR_ARG;
  hipStream_t cs = m->nranks > 1 ? m->st2 : m->st;  // packs, wall BCs and unpacks ride on the communication stream
  if (m->nranks > 1) comm_begin(m);
  if (m->nranks > 1) {
    Xfer x[2];
    int n = 0;
    for (int dir : {DIR_W, DIR_E}) {
      if (m->nb[dir] < 0) continue;
      launch_nat_pack_strip(cs, f, g, nl, dir == DIR_W ? 0 : g.nx - d, 0, d, g.ny, comm_sendbuf(m->comm, dir));
      x[n++] = {m->nb[dir], AXIS(dir), comm_sendbuf(m->comm, dir), comm_recvbuf(m->comm, dir), (size_t)d * g.ny * nl};
    }
    int r = comm_exchange(m->comm, x, n);
    if (r) { comm_end(m); return r; }
    for (int dir : {DIR_W, DIR_E}) {
      if (m->nb[dir] < 0) continue;
      launch_nat_unpack_strip(cs, f, g, nl, dir == DIR_W ? -d : g.nx, 0, d, g.ny, comm_recvbuf(m->comm, dir));
    }
  }
  launch_fill_ghost(cs, f, g, nl, bc, m->walls, d);
  if (m->nranks > 1) {
    Xfer x[2];
    int n = 0;
    const int w = g.nx + 2 * d;
    for (int dir : {DIR_S, DIR_N}) {
      if (m->nb[dir] < 0) continue;
      launch_nat_pack_strip(cs, f, g, nl, -d, dir == DIR_S ? 0 : g.ny - d, w, d, comm_sendbuf(m->comm, dir));
      x[n++] = {m->nb[dir], AXIS(dir), comm_sendbuf(m->comm, dir), comm_recvbuf(m->comm, dir), (size_t)d * w * nl};
    }
    int r = comm_exchange(m->comm, x, n);
    if (r) { comm_end(m); return r; }
    for (int dir : {DIR_S, DIR_N}) {
      if (m->nb[dir] < 0) continue;
      launch_nat_unpack_strip(cs, f, g, nl, -d, dir == DIR_S ? -d : g.ny, w, d, comm_recvbuf(m->comm, dir));
    }
  }
  if (m->nranks > 1) comm_end(m);
  return MSOM_OK;
}

// halo exchange (depth 1) of a multigrid field in split layout.  corners = 0: the 4 face
// neighbours in one phase (enough for the 5-point smoother); corners = 1: two phases so that
// the corner ghosts needed by the bilinear prolongation are valid.
// everything on the communication stream; the caller orders it against the compute stream
static int exch_split_raw(msom *m, double *f, const SplitGeom &sg, int nl, int corners) {
  hipStream_t cs = m->st2;
  if (!fits_comm(m, (size_t)nl * ((sg.nx > sg.ny ? sg.nx : sg.ny) + 2))) return MSOM_ERR_ARG;
  Xfer x[4];
  int n = 0;
  if (!corners) {  // one pack launch, one message per neighbour, one unpack launch
    double *sb[4], *rb[4];
    for (int dir : {DIR_W, DIR_E, DIR_S, DIR_N}) {
      const bool on = m->nb[dir] >= 0;
      sb[dir] = on ? comm_sendbuf(m->comm, dir) : nullptr;
      rb[dir] = on ? comm_recvbuf(m->comm, dir) : nullptr;
      if (on) x[n++] = {m->nb[dir], AXIS(dir), sb[dir], rb[dir], (size_t)nl * (dir == DIR_W || dir == DIR_E ? sg.ny : sg.nx)};
    }
    launch_split_pack_faces(cs, f, sg, nl, sb);
    int r = comm_exchange(m->comm, x, n);
    if (r) return r;
    launch_split_unpack_faces(cs, f, sg, nl, rb);
    return MSOM_OK;
  }
  auto pack = [&](int dir, int i0, int j0, int w, int h) {
    launch_split_pack_strip(cs, f, sg, nl, i0, j0, w, h, comm_sendbuf(m->comm, dir));
    x[n++] = {m->nb[dir], AXIS(dir), comm_sendbuf(m->comm, dir), comm_recvbuf(m->comm, dir), (size_t)w * h * nl};
  };
  const int x0 = -1, xw = sg.nx + 2;
  if (m->nb[DIR_W] >= 0) pack(DIR_W, 0, 0, 1, sg.ny);
  if (m->nb[DIR_E] >= 0) pack(DIR_E, sg.nx - 1, 0, 1, sg.ny);
  int r = comm_exchange(m->comm, x, n);
  if (r) return r;
  if (m->nb[DIR_W] >= 0) launch_split_unpack_strip(cs, f, sg, nl, -1, 0, 1, sg.ny, comm_recvbuf(m->comm, DIR_W));
  if (m->nb[DIR_E] >= 0) launch_split_unpack_strip(cs, f, sg, nl, sg.nx, 0, 1, sg.ny, comm_recvbuf(m->comm, DIR_E));
  launch_split_wall_corners(cs, f, sg, nl, m->walls);
  n = 0;
  if (m->nb[DIR_S] >= 0) pack(DIR_S, x0, 0, xw, 1);
  if (m->nb[DIR_N] >= 0) pack(DIR_N, x0, sg.ny - 1, xw, 1);
  if ((r = comm_exchange(m->comm, x, n))) return r;
  if (m->nb[DIR_S] >= 0) launch_split_unpack_strip(cs, f, sg, nl, x0, -1, xw, 1, comm_recvbuf(m->comm, DIR_S));
  if (m->nb[DIR_N] >= 0) launch_split_unpack_strip(cs, f, sg, nl, x0, sg.ny, xw, 1, comm_recvbuf(m->comm, DIR_N));
  launch_split_wall_corners(cs, f, sg, nl, m->walls);
  return MSOM_OK;
}
static int exch_split(msom *m, double *f, const SplitGeom &sg, int nl, int corners) {
  if (m->nranks == 1) return MSOM_OK;
  comm_begin(m);
  const int r = exch_split_raw(m, f, sg, nl, corners);
  comm_end(m);
  return r;
}

// Deep halo of a split field for the chained smoother (kernels_march.hip): MARCH_HALO cells beyond the W / E tile edges
// go into the row pads of the field itself, MARCH_HALO rows beyond the S / N edges into the halo arrays fs / fn
// (geometry hg: MARCH_HALO rows of the level).  Two phases, the second one carries the freshly received pad columns,
// so the corner regions arrive too.  Edges without a neighbour (walls) are left alone.
static int exch_split_deep(msom *m, double *f, const SplitGeom &sg, double *fs, double *fn, const SplitGeom &hg, int nl) {
  hipStream_t cs = m->st2;
  const int H = MARCH_HALO;
  Xfer x[2];
  int n = 0;
  if (!fits_comm(m, (size_t)H * ((sg.nx > sg.ny ? sg.nx : sg.ny) + 2 * H) * nl)) return MSOM_ERR_ARG;
  comm_begin(m);
  auto pack = [&](int dir, int i0, int j0, int w, int h) {
    launch_split_pack_strip(cs, f, sg, nl, i0, j0, w, h, comm_sendbuf(m->comm, dir));
    x[n++] = {m->nb[dir], AXIS(dir), comm_sendbuf(m->comm, dir), comm_recvbuf(m->comm, dir), (size_t)w * h * nl};
  };
  // rows -1 and ny ride along: next to a y wall they are the neighbour's wall ghosts, which the cells of the halo
  // columns read like any other row
  if (m->nb[DIR_W] >= 0) pack(DIR_W, 0, -1, H, sg.ny + 2);
  if (m->nb[DIR_E] >= 0) pack(DIR_E, sg.nx - H, -1, H, sg.ny + 2);
  int r = comm_exchange(m->comm, x, n);
  if (r) { comm_end(m); return r; }   // the compute stream is re-ordered after the communication stream on every exit
  if (m->nb[DIR_W] >= 0) launch_split_unpack_strip(cs, f, sg, nl, -H, -1, H, sg.ny + 2, comm_recvbuf(m->comm, DIR_W));
  if (m->nb[DIR_E] >= 0) launch_split_unpack_strip(cs, f, sg, nl, sg.nx, -1, H, sg.ny + 2, comm_recvbuf(m->comm, DIR_E));
  n = 0;
  const int x0 = -H, xw = sg.nx + 2 * H;
  if (m->nb[DIR_S] >= 0) pack(DIR_S, x0, 0, xw, H);
  if (m->nb[DIR_N] >= 0) pack(DIR_N, x0, sg.ny - H, xw, H);
  if ((r = comm_exchange(m->comm, x, n))) { comm_end(m); return r; }
  if (m->nb[DIR_S] >= 0) launch_split_unpack_strip(cs, fs, hg, nl, x0, 0, xw, H, comm_recvbuf(m->comm, DIR_S));
  if (m->nb[DIR_N] >= 0) launch_split_unpack_strip(cs, fn, hg, nl, x0, 0, xw, H, comm_recvbuf(m->comm, DIR_N));
  comm_end(m);
  return MSOM_OK;
}

// ------------------------------------------------------------------ lifecycle

static int alloc_all(msom *m) {
  HIPCHK(hipStreamCreate(&m->st));
  HIPCHK(hipEventCreateWithFlags(&m->ev_spec, hipEventDisableTiming));
  HIPCHK(hipHostMalloc(&m->h_pub, (SC_COUNT + 1) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  memset(m->h_pub, 0, (SC_COUNT + 1) * sizeof(double));
  HIPCHK(hipHostGetDevicePointer((void **)&m->d_pub, m->h_pub, 0));
  if (m->nranks > 1) {
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&m->st2, hipStreamNonBlocking, hi));
    HIPCHK(hipEventCreateWithFlags(&m->ev_c2x, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_x2c, hipEventDisableTiming));
  }
  m->g = make_nat(m->nx, m->ny);
  for (int k = 0; k < MSOM_NFIELDS; k++) {
    m->f[k] = nullptr;
    m->flayers[k] = m->nl;
    m->fbc[k] = m->bc;
  }
  m->flayers[MSOM_FR] = m->flayers[MSOM_S] = m->nlm;
  m->flayers[MSOM_RO] = m->flayers[MSOM_TOPO] = 1;
  m->fbc[MSOM_FR] = m->fbc[MSOM_S] = m->fbc[MSOM_RO] = m->fbc[MSOM_TOPO] = m->bc == BC_PERIODIC ? BC_PERIODIC : BC_NEUMANN;
  if (m->bc == BC_PERIODIC) m->fbc[MSOM_PSIPG] = BC_DIRICHLET_LIN;  // msqg/qg.h:1105-1114
  for (int k = MSOM_PTR; k <= MSOM_PTR_PRED; k++) {  // passive tracers: Neumann / periodic (msqg/qg.h:867-870)
    m->flayers[k] = m->nl * (m->p.nptr > 0 ? m->p.nptr : 1);
    m->fbc[k] = m->bc == BC_PERIODIC ? BC_PERIODIC : BC_NEUMANN;
  }
  m->flayers[MSOM_RD] = 1;
  m->fbc[MSOM_RD] = m->bc == BC_PERIODIC ? BC_PERIODIC : BC_NEUMANN;
  for (int k = 0; k < MSOM_NFIELDS; k++) {
    if (k == MSOM_NOISE || k == MSOM_SIGMA || k == MSOM_QOF || k >= MSOM_DE_BF) continue;  // allocated when "stochastic" is switched on / on the first filter call
    if (k >= MSOM_PTR && k <= MSOM_PTR_PRED && m->p.nptr <= 0) continue;
    size_t bytes = m->g.ls * m->flayers[k] * sizeof(double);
    HIPCHK(hipMalloc(&m->f[k], bytes));
    HIPCHK(hipMemsetAsync(m->f[k], 0, bytes, m->st));
  }
  // multigrid levels: minlevel = 1 (msqg/poisson_layer.h:296-297) -> coarsest tile has 2 cells
  // on its short side
  int n = 0;
  while ((m->nx >> n) >= 2 && (m->ny >> n) >= 2 && ((m->nx >> n) << n) == m->nx && ((m->ny >> n) << n) == m->ny) n++;
  if (m->p.mglevels > 0 && m->p.mglevels < n) n = m->p.mglevels;
  m->nlev = n;
  m->sg.resize(n); m->da.resize(n); m->da_alt.resize(n); m->res.resize(n); m->S.resize(n); m->rc.resize(n);
  for (int k = 0; k < n; k++) {
    m->sg[k] = make_split(m->nx >> k, m->ny >> k);
    size_t bytes = m->sg[k].ls * m->nl * sizeof(double);
    HIPCHK(hipMalloc(&m->da[k], bytes));
    HIPCHK(hipMalloc(&m->da_alt[k], bytes));
    HIPCHK(hipMemsetAsync(m->da_alt[k], 0, bytes, m->st));
    HIPCHK(hipMalloc(&m->res[k], bytes));
    HIPCHK(hipMalloc(&m->S[k], m->sg[k].ls * m->nlm * sizeof(double)));
    HIPCHK(hipMemsetAsync(m->da[k], 0, bytes, m->st));
    HIPCHK(hipMemsetAsync(m->res[k], 0, bytes, m->st));
    HIPCHK(hipMemsetAsync(m->S[k], 0, m->sg[k].ls * m->nlm * sizeof(double), m->st));
  }
  HIPCHK(hipMalloc(&m->psi_alt, m->g.ls * m->nl * sizeof(double)));
  HIPCHK(hipMemsetAsync(m->psi_alt, 0, m->g.ls * m->nl * sizeof(double), m->st));
  HIPCHK(hipMalloc(&m->staging, (size_t)m->nl * (m->p.nptr > 1 ? m->p.nptr : 1) * m->nx * m->ny * sizeof(double)));
  HIPCHK(hipMalloc(&m->partial, ((size_t)partial_count(m->g) * m->nl + 64) * sizeof(double)));  // + chunk sums of launch_sum_final
  HIPCHK(hipMalloc(&m->partial_rr, ((size_t)rhs_pipe_blocks(m->g) + 64) * sizeof(double)));
  {
    size_t nb = (size_t)rhs_fused_blocks(m->g);
    if (nb < 2048) nb = 2048;
    HIPCHK(hipMalloc(&m->partial_umax, nb * MSOM_MAXNL * sizeof(double)));
  }
  HIPCHK(hipMalloc(&m->d_scal, SC_COUNT * sizeof(double)));
  HIPCHK(hipMemsetAsync(m->d_scal, 0, SC_COUNT * sizeof(double), m->st));
  HIPCHK(hipHostMalloc(&m->h_scal, SC_COUNT * sizeof(double)));
  HIPCHK(hipMalloc(&m->d_wind, (size_t)m->ny * sizeof(double)));
  return MSOM_OK;
}

// set_vars, msqg/qg.h:837-925: defaults of the large-scale fields
static int set_vars(msom *m) {
  const Params &p = m->p;
  for (int l = 0; l < m->nl; l++) m->dhf[l] = p.dhu[l];
  std::vector<double> h((size_t)m->nl * m->nx * m->ny);
  const double D = p.L0 / m->gnx;
  // Fr[] = Frm[l]  (:898-902)
  for (int l = 0; l < m->nlm; l++)
    for (size_t k = 0; k < (size_t)m->nx * m->ny; k++) h[(size_t)l * m->nx * m->ny + k] = l < m->nl - 1 ? p.Frm[l] : 0.;
  int r = msom_set_field(m, MSOM_FR, h.data());
  if (r) return r;
  m->fr_uniform = 1;
  // pp[] = vpg*x - upg*y  (:904-909)
  m->have_pg = 0;
  for (int l = 0; l < m->nl; l++) {
    if (p.upg[l] != 0 || p.vpg[l] != 0) m->have_pg = 1;
    for (int j = 0; j < m->ny; j++)
      for (int i = 0; i < m->nx; i++) {
        const double x = (m->ix * m->nx + i + 0.5) * D, y = (m->iy * m->ny + j + 0.5) * D;
        h[((size_t)l * m->ny + j) * m->nx + i] = p.vpg[l] * x - p.upg[l] * y;
      }
  }
  int hp = m->have_pg;
  r = msom_set_field(m, MSOM_PSIPG, h.data());
  if (r) return r;
  m->have_pg = hp;
  // Ro[] = Rom, topo = 0  (:911-915)
  for (size_t k = 0; k < (size_t)m->nx * m->ny; k++) h[k] = p.Rom;
  r = msom_set_field(m, MSOM_RO, h.data());
  if (r) return r;
  for (size_t k = 0; k < (size_t)m->nx * m->ny; k++) h[k] = 1.;  // Rd[] = 1. (:913)
  r = msom_set_field(m, MSOM_RD, h.data());
  m->fr_uniform = 1;
  return r;
}

static msom *create_common(const Params &p0, int px, int py, int rank, const void *id128) {
  Params p = p0;
  if (p.Ny <= 0) p.Ny = p.N;
  msom_params_derive(&p);
  if (p.nl < 1 || p.nl > MSOM_MAXNL) {
    msom_set_error("nl = %d outside the supported range 1..%d", p.nl, MSOM_MAXNL);
    return nullptr;
  }
  if (p.N % px || p.Ny % py || !is_pow2(p.N / px) || !is_pow2(p.Ny / py) || p.N / px < 2 || p.Ny / py < 2) {
    msom_set_error("grid %d x %d on %d x %d tiles: every tile edge must be a power of two >= 2", p.N, p.Ny, px, py);
    return nullptr;
  }
  if (p.nptr < 0 || p.nptr > MSOM_MAXNL) {
    msom_set_error("nptr = %d outside 0..%d", p.nptr, MSOM_MAXNL);
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    msom_set_error("no HIP device available: libmsomhip has no CPU fallback");
    return nullptr;
  }
  msom *m = new msom();
  m->p = p;
  m->px = px; m->py = py; m->rank = rank; m->nranks = px * py;
  m->ix = rank % px; m->iy = rank / px;
  m->gnx = p.N; m->gny = p.Ny;
  m->nx = p.N / px; m->ny = p.Ny / py;
  m->nl = p.nl; m->nlm = p.nl > 1 ? p.nl - 1 : 1;
  m->walls = 0;
  if (p.sbc == -1) {  // periodic(right); periodic(top), msqg/qg.h:842-846
    m->bc = BC_PERIODIC;
    // one tile: ghost cells are wrapped copies kept by the owning thread (WALL_PER); tiles: no walls at all, every edge is
    // an exchange with the (wrapped) neighbour
    m->walls = px * py > 1 ? 0 : WALL_PER;
  } else {
    if (m->ix == 0) m->walls |= WALL_W;
    if (m->ix == px - 1) m->walls |= WALL_E;
    if (m->iy == 0) m->walls |= WALL_S;
    if (m->iy == py - 1) m->walls |= WALL_N;
  }
  // neighbour ranks
  for (int d = 0; d < 8; d++) m->nb[d] = -1;
  {
    const int dx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, dy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
    for (int d = 0; d < 8; d++) {
      const int jx = m->ix + dx[d], jy = m->iy + dy[d];
      if (p.sbc == -1 && px * py > 1) m->nb[d] = ((jy + py) % py) * px + (jx + px) % px;
      else if (jx >= 0 && jx < px && jy >= 0 && jy < py) m->nb[d] = jy * px + jx;
    }
  }
  if (alloc_all(m) != MSOM_OK ||
      (m->nranks > 1 && comm_create(&m->comm, rank, m->nranks, id128, m->st2,
                                    // largest message: a MARCH_HALO-deep strip of the widest exchanged field
                                    // (tracer fields carry nl * nptr layers, fill_bc exchanges them whole)
                                    (size_t)MARCH_HALO * ((m->nx > m->ny ? m->nx : m->ny) + 16) * m->nl * (p.nptr > 1 ? p.nptr : 1)) != MSOM_OK) ||
      set_vars(m) != MSOM_OK) {
    msom_destroy(m);
    return nullptr;
  }
  return m;
}

extern "C" msom_t *msom_create_str(const char *text) {
  if (!text) { msom_set_error("null params text"); return nullptr; }
  Params p;
  msom_params_defaults(&p);
  msom_params_parse_text(&p, text);
  msom *m = create_common(p, 1, 1, 0, nullptr);
  if (m) m->params_text = text;
  return m;
}
extern "C" msom_t *msom_create(const char *path) {
  const char *pp = path ? path : "params.in";
  FILE *fp = fopen(pp, "rt");
  if (!fp) {
    msom_set_error("file %s not found", pp);  // reference: message + exit(0), msqg/qg.h:735-738
    return nullptr;
  }
  std::string text;
  char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
  fclose(fp);
  return msom_create_str(text.c_str());
}

static void clear_graphs(msom *m);
extern "C" int msom_destroy(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  if (m->st) hipStreamSynchronize(m->st);
  clear_graphs(m);
  for (int k = 0; k < MSOM_NFIELDS; k++)
    if (m->f[k]) hipFree(m->f[k]);
  for (int k = 0; k < m->nlev; k++) {
    if (m->da[k]) hipFree(m->da[k]);
    if (m->da_alt[k]) hipFree(m->da_alt[k]);
    for (auto *v : {&m->mh_da_s, &m->mh_da_n, &m->mh_res_s, &m->mh_res_n})
      if (k < v->size() && (*v)[k]) hipFree((*v)[k]);
    if (m->res[k]) hipFree(m->res[k]);
    if (m->S[k]) hipFree(m->S[k]);
  }
  free_agglomeration(m);
  for (size_t k = 0; k < m->wv_sig.size(); k++) {
    if (k == 0) { if (m->wv_gsend) hipFree(m->wv_gsend); if (m->wv_grecv) hipFree(m->wv_grecv); }
    if (k > 0 && m->wv_s[k]) hipFree(m->wv_s[k]);
    if (k > 0 && m->wv_r[k]) hipFree(m->wv_r[k]);
    if (m->wv_sig[k]) hipFree(m->wv_sig[k]);
  }
  if (m->psi_alt) hipFree(m->psi_alt);
  if (m->q_alt) hipFree(m->q_alt);
  if (m->ev_spec) hipEventDestroy(m->ev_spec);
  if (m->h_pub) hipHostFree(m->h_pub);
  if (m->staging) hipFree(m->staging);
  if (m->partial) hipFree(m->partial);
  if (m->partial_rr) hipFree(m->partial_rr);
  if (m->d_cargs) hipFree(m->d_cargs);
  if (m->partial_umax) hipFree(m->partial_umax);
  if (m->d_scal) hipFree(m->d_scal);
  if (m->h_scal) hipHostFree(m->h_scal);
  if (m->d_wind) hipFree(m->d_wind);
  for (auto *ps : {&m->prof_sweep, &m->prof_resid, &m->prof_block, &m->prof_march[2], &m->prof_march[3], &m->prof_march[4], &m->prof_rhs, &m->prof_redprol, &m->prof_rescorr, &m->prof_respre, &m->prof_march_pl, &m->prof_march_corr, &m->prof_resmax})
    for (auto e : ps->ev) hipEventDestroy(e);
  if (m->comm) comm_destroy(m->comm);
  if (m->ev_c2x) hipEventDestroy(m->ev_c2x);
  if (m->ev_x2c) hipEventDestroy(m->ev_x2c);
  if (m->st2) hipStreamDestroy(m->st2);
  if (m->st) hipStreamDestroy(m->st);
  delete m;
  return MSOM_OK;
}

static int build_coefs(msom *m);
static void free_agglomeration(msom *m);

static void clear_graphs(msom *m);
extern "C" int msom_set_option(msom_t *m, const char *key, double v) {
  if (!m || !key) return MSOM_ERR_ARG;
  clear_graphs(m);   // captured cycles hold kernel arguments by value: any option may change them
  if (!strcmp(key, "graph")) { m->use_graph = (int)v; return MSOM_OK; }
  if (!strcmp(key, "TOLERANCE")) m->p.tolerance = v;
  else if (!strcmp(key, "NITERMAX")) m->p.nitermax = (int)v;
  else if (!strcmp(key, "NITERMIN")) m->p.nitermin = (int)v;
  else if (!strcmp(key, "DT")) m->p.DT = v;
  else if (!strcmp(key, "flag_topo")) m->flag_topo = (int)v;
  else if (!strcmp(key, "quiet")) m->quiet = (int)v;
  else if (!strcmp(key, "uniform_S")) {
    m->uniform_opt = (int)v;
    if (m->const_set) return build_coefs(m);
  }
  else if (!strcmp(key, "profile")) m->profile = (int)v;
  else if (!strcmp(key, "fused")) m->fused = (int)v;
  else if (!strcmp(key, "mg_fused")) m->mg_fused = (int)v;
  else if (!strcmp(key, "march")) m->march = (int)v;
  else if (!strcmp(key, "march_rows")) g_march_rows = (int)v;
  else if (!strcmp(key, "march_min")) m->march_min = (int)v;
  else if (!strcmp(key, "march_min_tiled")) m->march_min_tiled = (int)v;
  else if (!strcmp(key, "march_partial")) m->march_partial = (int)v;
  else if (!strcmp(key, "march_prolong")) m->march_prolong = (int)v;
  else if (!strcmp(key, "march_correct")) m->march_correct = (int)v;
  else if (!strcmp(key, "march_xcd")) { extern int g_march_remap; g_march_remap = (int)v; }
  else if (!strcmp(key, "march_flip")) { extern int g_march_flip; g_march_flip = (int)v; }
  else if (!strcmp(key, "march_dma")) { extern int g_march_dma; g_march_dma = (int)v; }
  else if (!strcmp(key, "march_dbg")) { extern int g_march_dbg; g_march_dbg = (int)v; }
  else if (!strcmp(key, "dbg_interleave")) g_dbg_interleave = (int)v;
  else if (!strcmp(key, "dbg_nosync")) m->dbg_nosync = (int)v;
  else if (!strcmp(key, "async_solve")) m->async_solve = (int)v;
  else if (!strcmp(key, "step_sync")) m->step_sync = (int)v;
  else if (!strcmp(key, "march_lean")) { extern int g_march_lean; g_march_lean = (int)v; }
  else if (!strcmp(key, "march_k")) m->march_k = (int)v < 2 ? 2 : ((int)v > 4 ? 4 : (int)v);
  else if (!strcmp(key, "block_small")) m->block_small = (int)v;
  else if (!strcmp(key, "block8")) m->block8 = (int)v;
  else if (!strcmp(key, "restrict2")) m->restrict2 = (int)v;
  else if (!strcmp(key, "restrict_pyr")) m->restrict_pyr = (int)v;
  else if (!strcmp(key, "block8_max")) m->block8_max = (int)v;
  else if (!strcmp(key, "block_sweeps")) { m->block_sweeps = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "mg_global_sum")) m->mg_global_sum = (int)v;
  else if (!strcmp(key, "agglomerate")) { m->agglomerate = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "agg_size")) { m->agg_size = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "prolong_fused")) { m->prolong_fused = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "mg_coarse")) { m->mgc_opt = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "mgc_pfused")) { m->mgc_pfused = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "mg_coarse_dim")) { m->mgc_dim = (int)v; if (m->const_set) return build_coefs(m); }
  else if (!strcmp(key, "block_variant")) { extern int g_block_variant; g_block_variant = (int)v; }
  else if (!strcmp(key, "lpw_dbg")) { extern int g_lpw_dbg; g_lpw_dbg = (int)v; }
  else if (!strcmp(key, "rhs_dbg")) { extern int g_rhs_dbg; g_rhs_dbg = (int)v; }
  else if (!strcmp(key, "resmax_rows")) { extern int g_resmax_rows; g_resmax_rows = (int)v; }
  else if (!strcmp(key, "rhs_variant")) m->rhs_variant = (int)v;
  else if (!strcmp(key, "adv_fused")) m->adv_fused = (int)v;
  else if (!strcmp(key, "overlap")) m->overlap = (int)v;
  else if (!strcmp(key, "rhs_resid")) { m->rhs_resid = (int)v; m->res_ready = -1; }
  else if (!strcmp(key, "seed")) { m->seed = (unsigned)v; srand(m->seed); }
  else if (!strcmp(key, "noise_mode")) m->noise_mode = (int)v;
  else if (!strcmp(key, "stoch_fused")) m->stoch_fused = (int)v;
  else if (!strcmp(key, "stochastic")) {
    m->stochastic = (int)v;
    if (m->stochastic && !m->f[MSOM_NOISE]) {
      for (int k : {MSOM_NOISE, MSOM_SIGMA}) {
        size_t bytes = m->g.ls * m->nl * sizeof(double);
        HIPCHK(hipMalloc(&m->f[k], bytes));
        HIPCHK(hipMemsetAsync(m->f[k], 0, bytes, m->st));
      }
    }
  } else {
    msom_set_error("unknown option %s", key);
    return MSOM_ERR_ARG;
  }
  return MSOM_OK;
}

static int march_levels(msom *m);
static bool restrict2_ok(const msom *m);
extern "C" double msom_get_param(msom_t *m, const char *key) {
  if (!m || !key) return NAN;
  const Params &p = m->p;
  if (!strcmp(key, "N") || !strcmp(key, "nx")) return m->gnx;
  if (!strcmp(key, "ny")) return m->gny;
  if (!strcmp(key, "nl")) return m->nl;
  if (!strcmp(key, "nptr")) return p.nptr;
  if (!strcmp(key, "L0")) return p.L0;
  if (!strcmp(key, "DT")) return p.DT;
  if (!strcmp(key, "iRe")) return p.iRe;
  if (!strcmp(key, "iRe4")) return p.iRe4;
  if (!strcmp(key, "CFL")) return p.CFL;
  if (!strcmp(key, "Rom")) return p.Rom;
  if (!strcmp(key, "tend")) return p.tend;
  if (!strcmp(key, "dtout")) return p.dtout;
  if (!strcmp(key, "beta")) return p.beta;
  if (!strcmp(key, "tau0")) return p.tau0;
  if (!strcmp(key, "Ekb")) return p.Ekb;
  if (!strcmp(key, "Eks")) return p.Eks;
  if (!strcmp(key, "sbc")) return p.sbc;
  if (!strcmp(key, "TOLERANCE")) return p.tolerance;
  if (!strcmp(key, "nlevels")) return m->nlev;
  if (!strcmp(key, "uniform_S")) return m->uniformS;
  if (!strcmp(key, "agg_level")) return m->agg_level;
  // which kernels the dispatch picks for this handle (bench.py names what ran from these, not from a table)
  if (!strcmp(key, "resmax_marching")) { extern int g_resmax_rows; return m->uniformS && m->nl <= MSOM_FASTNL && m->g.nx >= 64 && m->g.ny >= 16 && g_resmax_rows >= 0; }
  if (!strcmp(key, "march_lean")) { extern int g_march_lean; return g_march_lean; }
  if (!strcmp(key, "restrict2")) return restrict2_ok(m);   // the pre-cycle residual pass restricts two levels down
  if (!strcmp(key, "mg_coarse_lean")) return m->mgc_first >= 0 && m->mgc_lean;   // the coarse group runs in k_mg_coarse_lean
  if (!strcmp(key, "march_levels")) return march_levels(m);   // tile levels whose half-sweeps are chained (kernels_march.hip)
  auto idx = [](const char *s, int n) { const int k = atoi(s); return k >= 0 && k < n ? k : -1; };
  if (!strncmp(key, "idh0_", 5)) { const int k = idx(key + 5, MSOM_MAXNL); return k < 0 ? NAN : m->lc.idh0[k]; }
  if (!strncmp(key, "idh1_", 5)) { const int k = idx(key + 5, MSOM_MAXNL); return k < 0 ? NAN : m->lc.idh1[k]; }
  if (!strncmp(key, "Fr_", 3)) { const int k = idx(key + 3, MSOM_MAXARR); return k < 0 ? NAN : p.Frm[k]; }
  if (!strncmp(key, "dh_", 3)) { const int k = idx(key + 3, MSOM_MAXARR); return k < 0 ? NAN : m->dhf[k]; }
  return NAN;
}

// ------------------------------------------------------------------ fields

static int check_field(msom *m, int field) {
  if (!m || field < 0 || field >= MSOM_NFIELDS || !m->f[field]) {
    msom_set_error("bad field id %d", field);
    return MSOM_ERR_ARG;
  }
  return MSOM_OK;
}

// boundary(): wall BCs + halo exchange with the neighbour tiles.  The ghosts of q, dq and the
// predictor are never read (b enters the solver at cell centres only), so those fields skip
// the exchange.
static int fill_bc(msom *m, int field) {
  if (m->nranks > 1 && field != MSOM_Q && field != MSOM_DQ && field != MSOM_QPRED && field != MSOM_NOISE && field != MSOM_SIGMA) {
    int r = exch_nat(m, m->f[field], m->flayers[field], m->fbc[field], 1);
    if (!r && m->fbc[field] == BC_DIRICHLET_LIN) {  // the wrapped ghosts on the domain edges give way to dirichlet(vpg x - upg y)
      const int sides = (m->ix == 0 ? WALL_W : 0) | (m->ix == m->px - 1 ? WALL_E : 0) | (m->iy == 0 ? WALL_S : 0) | (m->iy == m->py - 1 ? WALL_N : 0);
      launch_fill_lin_dirichlet(m->st, m->f[field], m->g, m->flayers[field], m->p.upg, m->p.vpg, m->p.L0 / m->gnx, m->p.L0,
                                m->p.L0 * m->gny / m->gnx, m->ix * m->nx, m->iy * m->ny, sides);
    }
    return r;
  }
  if (m->fbc[field] == BC_PERIODIC) launch_fill_periodic(m->st, m->f[field], m->g, m->flayers[field], 1);
  else if (m->fbc[field] == BC_DIRICHLET_LIN)
    launch_fill_lin_dirichlet(m->st, m->f[field], m->g, m->flayers[field], m->p.upg, m->p.vpg, m->p.L0 / m->gnx, m->p.L0,
                              m->p.L0 * m->gny / m->gnx);
  else launch_fill_ghost(m->st, m->f[field], m->g, m->flayers[field], m->fbc[field], m->walls);
  return MSOM_OK;
}

// host or device pointer -> natural field (+ boundary())
static int upload(msom *m, int field, const double *a) {
  m->res_ready = -1;
  const size_t n = (size_t)m->flayers[field] * m->nx * m->ny;
  HIPCHK(hipMemcpyAsync(m->staging, a, n * sizeof(double), hipMemcpyDefault, m->st));
  launch_pack(m->st, m->staging, m->f[field], m->g, m->flayers[field]);
  fill_bc(m, field);
  return MSOM_OK;
}
static int download(msom *m, int field, double *a) {
  const size_t n = (size_t)m->flayers[field] * m->nx * m->ny;
  launch_unpack(m->st, m->f[field], m->staging, m->g, m->flayers[field]);
  HIPCHK(hipMemcpyAsync(a, m->staging, n * sizeof(double), hipMemcpyDefault, m->st));
  return sync_stream(m);
}

extern "C" int msom_field_layers(msom_t *m, int field) {
  if (check_field(m, field)) return MSOM_ERR_ARG;
  return m->flayers[field];
}

static int ensure_field(msom *m, int field);
extern "C" int msom_set_field(msom_t *m, int field, const double *a) {
  if (m && (field == MSOM_QOF || (field >= MSOM_DE_BF && field < MSOM_NFIELDS)) && ensure_field(m, field)) return MSOM_ERR_HIP;
  if (check_field(m, field) || !a) return MSOM_ERR_ARG;
  int r = upload(m, field, a);
  if (r) return r;
  if (field == MSOM_RD) m->wv_ready = 0;
  if (field == MSOM_FR || field == MSOM_RO || field == MSOM_S) { m->fr_uniform = 0; m->const_set = 0; }
  // the background flow feeds values cached by msom_set_const (max |u_pg| of the dt limiter, zeta_pg when flsrv = 1;
  // the reference recomputes them on every step, msqg/qg.h:383-391): a new psi_pg needs a new set_const
  if (field == MSOM_PSIPG) { m->have_pg = 1; m->const_set = 0; }
  if (field == MSOM_ZETAPG) m->have_zpg = 1;
  if (field == MSOM_QFORC) m->have_qforc = 1;
  if (field == MSOM_TOPO) m->flag_topo = 1;
  return sync_stream(m);
}
extern "C" int msom_get_field(msom_t *m, int field, double *a) {
  if (m && (field == MSOM_QOF || (field >= MSOM_DE_BF && field < MSOM_NFIELDS)) && ensure_field(m, field)) return MSOM_ERR_HIP;
  if (check_field(m, field) || !a) return MSOM_ERR_ARG;
  return download(m, field, a);
}

// msqg/qg.c:65-70: po[] -= s.sum/s.volume per layer
extern "C" int msom_remove_mean(msom_t *m, int field) {
  if (check_field(m, field)) return MSOM_ERR_ARG;
  const int nl = m->flayers[field];
  launch_sum_layers(m->st, m->f[field], m->partial, m->d_scal + SC_LSUM, m->g, nl);
  if (m->nranks > 1) {
    int r = reduce_scal(m, SC_LSUM, nl, RED_SUM);
    if (r) return r;
  }
  m->res_ready = -1;
  launch_sub_layer_const(m->st, m->f[field], m->d_scal + SC_LSUM, m->g, nl, 1. / ((double)m->gnx * m->gny));
  fill_bc(m, field);
  return sync_stream(m);
}

// ------------------------------------------------------------------ set_const

// uniform-S Thomas factorisation of one level (msqg/poisson_layer.h:84-140 with constant S)
static void make_relax_coef(msom *m, int k) {
  RelaxCoef &rc = m->rc[k];
  memset(&rc, 0, sizeof rc);
  const int nl = m->nl;
  rc.D = m->p.L0 / (double)(m->gnx >> k);
  rc.sqD = rc.D * rc.D;
  for (int l = 0; l < nl; l++) { rc.idh0[l] = m->lc.idh0[l]; rc.idh1[l] = m->lc.idh1[l]; }
  if (nl < 2) { rc.it1[0] = 0.25; return; }   // one layer: plain Poisson relaxation, x = rhs / 4 (exact either way)
  double t0[MSOM_MAXNL], t1[MSOM_MAXNL], t2[MSOM_MAXNL];
  for (int l = 0; l < nl - 1; l++) {
    const double r = m->s_zero ? 0. : m->p.Frm[l] / m->p.Rom;
    rc.S[l] = r * r;
  }
  for (int l = 0; l < nl; l++) {
    t0[l] = l > 0 ? -rc.sqD * rc.S[l - 1] * rc.idh0[l] : 0.;
    t2[l] = l < nl - 1 ? -rc.sqD * rc.S[l] * rc.idh1[l] : 0.;
    t1[l] = -t0[l] - t2[l] + 4.;
  }
  for (int l = 1; l < nl; l++) t1[l] -= t0[l] * t2[l - 1] / t1[l - 1];
  for (int l = 0; l < nl; l++) {
    rc.t2[l] = t2[l];
    rc.w[l] = l > 0 ? t0[l] / t1[l - 1] : 0.;
    rc.it1[l] = 1. / t1[l];
  }
}

static void free_agglomeration(msom *m) {
  for (auto *v : {&m->gda, &m->gda_alt, &m->gres}) {
    for (double *p : *v)
      if (p) hipFree(p);
    v->clear();
  }
  m->gsg.clear();
  if (m->agg_send) hipFree(m->agg_send);
  if (m->agg_recv) hipFree(m->agg_recv);
  m->agg_send = m->agg_recv = nullptr;
  m->agg_level = -1;
}
static int setup_agglomeration(msom *m) {
  hipStreamSynchronize(m->st);
  free_agglomeration(m);
  if (m->nranks == 1 || !m->agglomerate || (m->nl > 1 && !m->uniformS)) return MSOM_OK;
  int kc = -1;
  for (int k = 0; k < m->nlev; k++)
    if (m->sg[k].nx <= m->agg_size && m->sg[k].ny <= m->agg_size) { kc = k; break; }
  if (kc < 0) return MSOM_OK;
  const int n = m->nlev - kc;
  m->gsg.resize(n); m->gda.assign(n, nullptr); m->gda_alt.assign(n, nullptr); m->gres.assign(n, nullptr);
  for (int q = 0; q < n; q++) {
    m->gsg[q] = make_split(m->sg[kc + q].nx * m->px, m->sg[kc + q].ny * m->py);
    const size_t bytes = m->gsg[q].ls * m->nl * sizeof(double);
    HIPCHK(hipMalloc(&m->gda[q], bytes));
    HIPCHK(hipMalloc(&m->gda_alt[q], bytes));
    HIPCHK(hipMalloc(&m->gres[q], bytes));
    HIPCHK(hipMemsetAsync(m->gda[q], 0, bytes, m->st));
    HIPCHK(hipMemsetAsync(m->gda_alt[q], 0, bytes, m->st));
    HIPCHK(hipMemsetAsync(m->gres[q], 0, bytes, m->st));
  }
  const size_t cnt = (size_t)m->nl * m->sg[kc].nx * m->sg[kc].ny;
  HIPCHK(hipMalloc(&m->agg_send, cnt * sizeof(double)));
  HIPCHK(hipMalloc(&m->agg_recv, cnt * m->nranks * sizeof(double)));
  m->agg_level = kc;
  return MSOM_OK;
}

// group of coarsest levels handled by k_mg_coarse: the gathered global levels (tiles + agglomeration) or
// this tile's own levels (single tile); levels that need halo exchanges stay on the per-kernel path
static int setup_mg_coarse(msom *m) {
  m->mgc_first = -1;
  m->mgc_lean = 0;
  if (!m->mgc_opt || m->block_sweeps || m->nlev < 1 || m->nl > MSOM_FASTNL) return MSOM_OK;
  const bool glob = m->agg_level >= 0;
  if (!glob && m->nranks > 1) return MSOM_OK;
  const int klo = glob ? m->agg_level : 0;
  int k0 = -1;
  for (int k = m->nlev - 1; k >= klo; k--) {
    const SplitGeom &g = glob ? m->gsg[k - m->agg_level] : m->sg[k];
    if (g.nx > m->mgc_dim || g.ny > m->mgc_dim) break;
    k0 = k;
  }
  if (k0 < 0 || m->nlev - k0 > MGC_MAXLEV) return MSOM_OK;
  {
    // k_mg_coarse declares a static LDS pool sized for the 160 KB of gfx950 (MGC_POOL doubles + its bookkeeping): on a
    // device that cannot give a workgroup that much the kernel cannot launch at all, so the levels keep one launch each
    int dev = 0, lds_max = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || (size_t)lds_max < mg_coarse_static_lds()) return MSOM_OK;
  }
  CoarseArgs h;
  memset(&h, 0, sizeof h);
  h.n = m->nlev - k0;
  h.walls = glob ? (m->bc == BC_PERIODIC ? WALL_PER : WALL_ALL) : m->walls;
  h.prolong_fused = m->prolong_fused && m->mgc_pfused;  // the kernel itself compiles the fused phase out from nl = 5 on (registers)
  h.lds = m->mgc_opt == 3 ? 2 : (m->mgc_opt >= 2 ? 1 : 0);
  for (int k = k0; k < m->nlev; k++) {
    CoarseLev &L = h.lev[k - k0];
    if (glob) { const int q = k - m->agg_level; L.da = m->gda[q]; L.res = m->gres[q]; L.S = nullptr; L.g = m->gsg[q]; }
    else { L.da = m->da[k]; L.res = m->res[k]; L.S = m->S[k]; L.g = m->sg[k]; }
    L.rc = m->rc[k];
  }
  // mg_coarse = 4 (default): the lean LDS form of the kernel where it exists -- uniform S or one layer, walls on every side or the
  // doubly periodic single tile (also the gathered levels of a tiled run, which are a whole domain), all levels inside the pool
  m->mgc_lean = m->mgc_opt >= 4 && (m->uniformS || m->nl == 1) && (h.walls == WALL_ALL || h.walls == WALL_PER) &&
                mg_coarse_lean_doubles(h, m->nl) <= (size_t)MGC_POOL;
  if (!m->d_cargs) HIPCHK(hipMalloc(&m->d_cargs, sizeof(CoarseArgs)));
  HIPCHK(hipMemcpyAsync(m->d_cargs, &h, sizeof h, hipMemcpyHostToDevice, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  m->mgc_first = k0;
  return MSOM_OK;
}

// layer metrics, Ro, S on all levels, column-solver constants, forcing profile
static int build_coefs(msom *m) {
  clear_graphs(m);
  const Params &p = m->p;
  const int nl = m->nl;
  // sanity checks :990-1012 (reference: exit(0))
  for (int l = 0; l < nl; l++)
    if (m->dhf[l] == 0) {
      msom_set_error("thickness = 0: check the definition of dh in params.in");
      return MSOM_ERR_CONFIG;
    }
  if (p.Rom <= 0) {
    msom_set_error("Rom <= 0");
    return MSOM_ERR_CONFIG;
  }
  // layer metrics :1017-1027
  memset(&m->lc, 0, sizeof m->lc);
  for (int l = 0; l < nl - 1; l++) m->dhc[l] = 0.5 * (m->dhf[l] + m->dhf[l + 1]);
  if (nl > 1) {
    m->lc.idh0[0] = 0.;
    m->lc.idh1[0] = 1. / (m->dhc[0] * m->dhf[0]);
    for (int l = 1; l < nl - 1; l++) {
      m->lc.idh0[l] = 1. / (m->dhc[l - 1] * m->dhf[l]);
      m->lc.idh1[l] = 1. / (m->dhc[l] * m->dhf[l]);
    }
    m->lc.idh0[nl - 1] = 1. / (m->dhc[nl - 2] * m->dhf[nl - 1]);
    m->lc.idh1[nl - 1] = 0.;
  }
  const double D = p.L0 / m->gnx;
  // variable Rossby number :1032-1037
  if (p.varRo > 0) {
    std::vector<double> h((size_t)m->nx * m->ny);
    for (int j = 0; j < m->ny; j++) {
      const double y = (m->iy * m->ny + j + 0.5) * D;
      for (int i = 0; i < m->nx; i++) h[(size_t)j * m->nx + i] = p.Rom / (1 + p.Rom * p.beta * (y - 0.5 * p.L0));
    }
    int r = upload(m, MSOM_RO, h.data());
    if (r) return r;
    m->fr_uniform = 0;
  }
  // S = (Fr/Ro)^2 :1043-1048, then restricted to every level (hoisted out of the solve:
  // the reference redoes it on every poisson_layer call, msqg/poisson_layer.h:284)
  launch_make_S(m->st, m->f[MSOM_FR], m->f[MSOM_RO], m->f[MSOM_S], m->g, m->nlm);
  if (m->s_zero) HIPCHK(hipMemsetAsync(m->f[MSOM_S], 0, m->g.ls * m->nlm * sizeof(double), m->st));
  fill_bc(m, MSOM_S);
  launch_nat_to_split(m->st, m->f[MSOM_S], m->g, m->S[0], m->sg[0], m->nlm);
  for (int k = 1; k < m->nlev; k++) launch_restrict(m->st, m->S[k - 1], m->sg[k - 1], m->S[k], m->sg[k], m->nlm);
#ifdef MSOM_STRICT
  const int auto_uniform = 0;
#else
  const int auto_uniform = 1;
#endif
  m->uniformS = (m->uniform_opt < 0 ? auto_uniform : m->uniform_opt) && m->fr_uniform && nl > 1;
  for (int k = 0; k < m->nlev; k++) make_relax_coef(m, k);
  // coarse-level agglomeration (tiles): from the first level whose tile is <= agg_size cells
  // wide, every rank holds the whole coarse grid
  {
    int r = setup_agglomeration(m);
    if (r) return r;
    if ((r = setup_mg_coarse(m))) return r;
  }
  // surface forcing profile :451 (host libm so that it matches the CPU formulation bit for bit)
  {
    std::vector<double> w(m->ny);
    for (int j = 0; j < m->ny; j++) {
      const double y = (m->iy * m->ny + j + 0.5) * D;
      w[j] = p.tau0 / (p.Rom * m->dhf[0]) * sin(2 * M_PI * y / p.L0) * sin(M_PI * y / p.L0);
    }
    HIPCHK(hipMemcpyAsync(m->d_wind, w.data(), m->ny * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
  }
  return MSOM_OK;
}

extern "C" int msom_set_const(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  m->s_zero = 0;
  int rr = build_coefs(m);
  if (rr) return rr;
  m->wv_ready = 0;  // sig_filt / sig_lev are rebuilt from Rd on the next filter call (msqg/qg.h:1059-1090)
  const Params &p = m->p;
  const int nl = m->nl;
  const double D = p.L0 / m->gnx;
  // q = comp_q(psi) :1092
  launch_del2(m->st, m->f[MSOM_PSI], m->f[MSOM_Q], m->g, nl, 0., 1., D);
  launch_stretch(m->st, m->f[MSOM_PSI], m->f[MSOM_Q], m->f[MSOM_S], m->g, nl, 1., 1., m->lc);
  fill_bc(m, MSOM_Q);
  // large-scale relative vorticity :1094-1097
  if (p.flsrv == 1) {
    launch_del2(m->st, m->f[MSOM_PSIPG], m->f[MSOM_ZETAPG], m->g, nl, 0., 1., D);
    fill_bc(m, MSOM_ZETAPG);
    m->have_zpg = 1;
  }
  // max |u| of the large-scale flow is constant in time: cache it for the dt limiter
  for (int l = 0; l < MSOM_MAXNL; l++) m->umax_pg[l] = 0.;
  if (m->have_pg) {
    launch_umax(m->st, m->f[MSOM_PSIPG], m->partial_umax, m->d_scal + SC_UMAX, m->g, nl, D);
    int r = reduce_scal(m, SC_UMAX, nl, RED_MAX);
    if (r) return r;
    for (int l = 0; l < nl; l++) m->umax_pg[l] = m->h_scal[SC_UMAX + l];
  }
  m->const_set = 1;
  return sync_stream(m);
}

// ------------------------------------------------------------------ elliptic solver

// One multigrid level as the smoother sees it: either this rank's tile of level k (halo
// exchange with the neighbour tiles after every colour) or, below the agglomeration level, the
// whole coarse grid gathered on every rank (walls only, no communication).
struct Lev {
  double **da, **da_alt;
  double *res;
  const double *S;
  const SplitGeom *sg;
  const RelaxCoef *rc;
  int walls;
  bool tiled;   // needs halo exchanges
  bool fine;    // level 0 (profiling tag)
  int k;        // tile level index (-1: gathered global level)
};
static Lev tile_lev(msom *m, int k) {
  return Lev{&m->da[k], &m->da_alt[k], m->res[k], m->S[k], &m->sg[k], &m->rc[k], m->walls, m->nranks > 1, k == 0, k};
}
static Lev glob_lev(msom *m, int k) {
  const int q = k - m->agg_level;
  return Lev{&m->gda[q], &m->gda_alt[q], m->gres[q], nullptr, &m->gsg[q], &m->rc[k], m->bc == BC_PERIODIC ? WALL_PER : WALL_ALL, false, false, -1};
}

// can the level use the temporally blocked smoother (k_relax_block: 2 sweeps per pass)?
static bool block_ok(msom *m, const Lev &L) {
  const bool want = m->block_sweeps || (m->block_small && L.sg->nx <= m->block_small);
  return want && m->uniformS && m->nl <= MSOM_FASTNL && !L.tiled && !(m->walls & WALL_PER) && L.sg->nx >= 64 && L.sg->ny >= 16;
}
// launch-bound levels (round 3): the prolongation and up to 8 colour half-sweeps of a level visit in ONE launch of the LDS-tiled
// smoother with a halo of 8 (k_relax_block<.., 8>, option block8; levels of 64 .. block8_max cells a side that are not marched).
// One GPU, walls or doubly periodic, nl <= 8 (the fast kernels' limit); uniform S, one layer, or a general S field.  Measured at 4096^2 x 6 / 512^2 x 3: see DESIGN.md section 4
static bool march_ok(msom *m, const Lev &L);
static bool block8_ok(msom *m, const Lev &L) {
  // (a general S field: the kernel's GENERAL instantiation, S of the owned cells in registers; the gathered levels of a tiled run have no S array)
  if (!m->block8 || m->block_sweeps || !(m->uniformS || m->nl == 1 || L.S) || m->nl > MSOM_FASTNL || L.tiled || (L.walls != WALL_ALL && L.walls != WALL_PER)) return false;
  if (L.sg->nx < 64 || L.sg->ny < 16 || L.sg->nx > m->block8_max) return false;
  // doubly periodic single tile: the kernel wraps its loads; the region (<= 48 x 32 cells) must not meet its own image; the gathered
  // coarse levels of tiled runs keep their per-colour launches
  if (L.walls == WALL_PER && (L.k < 0 || L.sg->ny < 64)) return false;
  return !march_ok(m, L);
}
// can the level chain its half-sweeps in registers (k_relax_march)?  One GPU (no halo exchange between half-sweeps),
// walls, uniform S, and a level big enough to be HBM-bound: a marching wavefront pays one memory latency per row, which
// only ~2000 concurrent chunks hide (measured at nl = 6: 4096^2 1.54 -> 1.05 ms per 7 half-sweeps, 2048^2 385 -> 310 us,
// but 1024^2 105 -> 238 us).  march = 2 forces it on every level that is wide enough (tests)
static bool march_ok(msom *m, const Lev &L) {
  // round 3: one layer (no vertical coupling: the column system is x = rhs / 4) and the doubly periodic single tile (deep
  // halo = the field's own other side, launch_split_wrap) take the pass too
  const bool walls_ok = L.tiled || L.walls == WALL_ALL || L.walls == WALL_PER;
  if (!m->march || m->block_sweeps || !(m->uniformS || m->nl == 1) || m->nl > MSOM_FASTNL || !walls_ok || L.sg->nx < 512 || L.sg->ny < 64) return false;
  if (!L.tiled && L.walls == WALL_PER && L.k < 0) return false;   // gathered coarse levels keep their per-colour launches
  // tiles: one level more (2^22 cell-layers: 1024^2 x 6) -- on one GPU marching that level is neutral with the round-3 body (6.69 vs
  // 6.66 ms per step at 4096^2 x 6, 1.57 vs 1.56 at 2048^2 x 3), on tiles it replaces 8 per-colour halo exchanges of a level visit
  // by 3 deep ones; unmeasured on more than one GPU (none available), so the threshold is a separate option
  const int lg = L.tiled ? m->march_min_tiled : m->march_min;
  return m->march >= 2 || (size_t)L.sg->nx * L.sg->ny * m->nl >= ((size_t)1 << lg);
}
static int march_levels(msom *m) {
  int n = 0;
  for (int k = 0; k < m->nlev && (m->agg_level < 0 || k < m->agg_level); k++) {
    Lev L = tile_lev(m, k);
    n += march_ok(m, L);
  }
  return n;
}
// is the prolongation coarse -> L folded into the first smoothing pass of L?
static bool fuse_prolong(msom *m, const Lev &L, int nrelax) {
  if (block_ok(m, L) && nrelax >= 2) return true;
  if (block8_ok(m, L) && nrelax >= 1) return true;
  return m->prolong_fused && m->nl <= MSOM_FASTNL && nrelax >= 1 && L.sg->nx >= 4 && L.sg->ny >= 4;
}

// nrelax red-black relaxations of L.da against L.res (each followed by boundary_level).
// coarse != nullptr: L.da has not been prolongated yet -- the first pass interpolates it from
// the coarser level on the fly.  corners_last: the last exchange also carries corner ghosts.
static void relax_sweeps(msom *m, Lev &L, const Lev *coarse, int nrelax, int corners_last) {
  const bool prof = m->profile && L.fine;
  const int nl = m->nl;
  int it = 0;
  if (march_ok(m, L)) {
    // 2 nrelax half-sweeps; the first red one may carry the prolongation (in place), the others go in passes of
    // up to march_k, ping-ponging between the two correction buffers; a single left-over half-sweep runs in place.
    // Tiles: a pass reads MARCH_HALO cells / rows of its neighbours (exchanged once per pass instead of once per
    // half-sweep; the cone of dependence is re-computed, bit-identically, on both sides of the edge)
    int n = 2 * nrelax, c = 0;
    // doubly periodic single tile: the pass sees a tile without walls whose four neighbours are the tile itself; the deep
    // halo (pads W / E, halo arrays S / N) is filled by local copies (launch_split_wrap) where tiles exchange
    const bool wrap = !L.tiled && (L.walls & WALL_PER);
    const bool deep = L.tiled || wrap;
    const int kwalls = wrap ? 0 : L.walls;
    auto deep_halo = [&](double *f, const SplitGeom &sg, double *fs, double *fn, const SplitGeom &hgeo) {
      if (L.tiled) STICKY(m, exch_split_deep(m, f, sg, fs, fn, hgeo, nl));
      else { launch_split_wrap(m->st, f, sg, fs, fn, hgeo, nl, MARCH_HALO, 0); launch_split_wrap(m->st, f, sg, fs, fn, hgeo, nl, MARCH_HALO, 1); }
    };
    auto has_nb = [&](int dir) { return wrap || m->nb[dir] >= 0; };
    MarchHalo mh{nullptr, nullptr, nullptr, nullptr, 0, MARCH_HALO};
    SplitGeom hg = make_split(L.sg->nx, MARCH_HALO);
    if (deep) {
      const int k = L.k;
      if (m->mh_da_s.size() < (size_t)m->nlev) {
        m->mh_da_s.assign(m->nlev, nullptr); m->mh_da_n.assign(m->nlev, nullptr); m->mh_res_s.assign(m->nlev, nullptr); m->mh_res_n.assign(m->nlev, nullptr);
      }
      if (!m->mh_da_s[k]) {
        for (auto *v : {&m->mh_da_s, &m->mh_da_n, &m->mh_res_s, &m->mh_res_n}) {
          if (hipMalloc(&(*v)[k], hg.ls * nl * sizeof(double)) != hipSuccess) { m->sticky = MSOM_ERR_HIP; return; }
          hipMemsetAsync((*v)[k], 0, hg.ls * nl * sizeof(double), m->st);
        }
      }
      mh.ls = hg.ls;
      mh.in_s = has_nb(DIR_S) ? m->mh_da_s[k] : nullptr; mh.in_n = has_nb(DIR_N) ? m->mh_da_n[k] : nullptr;
      mh.res_s = has_nb(DIR_S) ? m->mh_res_s[k] : nullptr; mh.res_n = has_nb(DIR_N) ? m->mh_res_n[k] : nullptr;
    }
    // Tiles (option overlap): a pass is two launches -- the chunks that read nothing beyond the tile (region 1) are queued on
    // the compute stream FIRST, the deep halo exchanges of the pass then run beside them on the communication stream, the
    // chunks along the tile edges (region 2) follow once the halos have arrived.  Chunks are independent (out of place), so
    // the order changes no bit.  The residual's deep halo is constant during the sweeps: it rides with the first pass.
    const bool ovl = L.tiled && m->overlap;
    bool res_halo_done = !deep;
    auto residual_halo = [&]() {
      if (res_halo_done) return;
      deep_halo(const_cast<double *>(L.res), *L.sg, m->mh_res_s[L.k], m->mh_res_n[L.k], hg);
      res_halo_done = true;
    };
    const int kmax = nl >= 7 && m->march_k > 3 ? 3 : m->march_k;
    // tiles: the pass then needs MARCH_HALO cells / rows of the COARSE correction beyond the tile edges too (LDS-DMA kernel, nl <= 6)
    const bool pl_tiled = deep && coarse && coarse->k >= 0 && nl <= 6 && m->march_prolong >= 1 && coarse->sg->nx >= 2 * MARCH_HALO && coarse->sg->ny >= 2 * MARCH_HALO;
    if (coarse && n >= 3 && kmax >= 3 && (!deep || pl_tiled) && m->march_prolong) {
      // whole levels: the prolongation rides in the first PASS (its input is interpolated from the coarse level on
      // the fly), so the 2 nrelax half-sweeps are 4 + 4 instead of (red + prolongation) + 4 + 3
      int K = n < kmax ? n : kmax;
      if (n - K == 1 && K > 3) K--;
      MarchHalo ch{nullptr, nullptr, nullptr, nullptr, 0, MARCH_HALO};
      if (deep) {   // deep halo of the coarse correction: pads W / E, halo arrays S / N (those of the coarse level's own passes)
        const int ck = coarse->k;
        SplitGeom chg = make_split(coarse->sg->nx, MARCH_HALO);
        if (!m->mh_da_s[ck]) {
          for (auto *v : {&m->mh_da_s, &m->mh_da_n, &m->mh_res_s, &m->mh_res_n}) {
            if (hipMalloc(&(*v)[ck], chg.ls * nl * sizeof(double)) != hipSuccess) { m->sticky = MSOM_ERR_HIP; return; }
            hipMemsetAsync((*v)[ck], 0, chg.ls * nl * sizeof(double), m->st);
          }
        }
        ch.ls = chg.ls;
        ch.in_s = has_nb(DIR_S) ? m->mh_da_s[ck] : nullptr; ch.in_n = has_nb(DIR_N) ? m->mh_da_n[ck] : nullptr;
      }
      auto pl_pass = [&](int region) {
        if (launch_relax_march(m->st, nullptr, *L.da_alt, L.res, *L.sg, nl, *L.rc, 0, K, kwalls, g_march_rows, deep ? &mh : nullptr, *coarse->da, coarse->sg, nullptr,
                               m->march_partial && n - K >= 1, deep ? &ch : nullptr, region))
          m->sticky = MSOM_ERR_ARG;
      };
      auto pl_halos = [&]() {
        residual_halo();
        if (deep) deep_halo(*coarse->da, *coarse->sg, m->mh_da_s[coarse->k], m->mh_da_n[coarse->k], make_split(coarse->sg->nx, MARCH_HALO));
      };
      if (prof) prof_begin(m, m->prof_march_pl);
      if (ovl) {
        comm_begin(m); m->comm_hold = 1;
        pl_pass(1);
        pl_halos();
        m->comm_hold = 0; comm_end(m);
        pl_pass(2);
      } else {
        pl_halos();
        pl_pass(0);
      }
      if (prof) prof_end(m, m->prof_march_pl);
      std::swap(*L.da, *L.da_alt);
      n -= K; c = K & 1;
    } else if (coarse && n > 0) {
      if (prof) prof_begin(m, m->prof_redprol);
      launch_relax_red_prolong(m->st, *L.da, *coarse->da, *coarse->sg, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, L.walls);
      if (prof) prof_end(m, m->prof_redprol);
      n--; c = 1;
    }
    while (n >= 2) {
      int K = n < kmax ? n : kmax;
      if (n - K == 1 && K > 2) K--;
      // the very last pass of the finest level can apply the correction itself: psi_alt = psi + da (mg_solve swaps)
      const bool corr = m->corr_req && L.fine && n == K;
      MarchCorrect mc{m->f[MSOM_PSI], m->psi_alt, m->g};
      auto pass = [&](int region) {
        if (launch_relax_march(m->st, *L.da, *L.da_alt, L.res, *L.sg, nl, *L.rc, c, K, kwalls, g_march_rows, deep ? &mh : nullptr, nullptr, nullptr,
                               corr ? &mc : nullptr, m->march_partial && n - K >= 1, nullptr, region))
          m->sticky = MSOM_ERR_ARG;
      };
      auto halos = [&]() {
        residual_halo();
        if (deep) deep_halo(*L.da, *L.sg, m->mh_da_s[L.k], m->mh_da_n[L.k], hg);
      };
      if (prof) prof_begin(m, corr ? m->prof_march_corr : m->prof_march[K]);
      if (ovl) {
        comm_begin(m); m->comm_hold = 1;
        pass(1);
        halos();
        m->comm_hold = 0; comm_end(m);
        pass(2);
      } else {
        halos();
        pass(0);
      }
      if (prof) prof_end(m, corr ? m->prof_march_corr : m->prof_march[K]);
      n -= K; c = (c + K) & 1;
      if (corr) { m->corr_done = 1; return; }  // da of this level was consumed in registers; nothing reads it any more
      std::swap(*L.da, *L.da_alt);
    }
    // periodic single tile: the passes wrote no ghost cell; boundary_level(da, l) = the wrapped copies, corners included
    if (wrap) launch_split_wrap(m->st, *L.da, *L.sg, nullptr, nullptr, hg, nl, MARCH_HALO, 2);
    if (n == 1) {
      if (L.tiled) STICKY(m, exch_split(m, *L.da, *L.sg, nl, 0));
      launch_relax_color(m->st, *L.da, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, c, L.walls, L.fine);
    }
    // boundary_level(da, l) after the last half-sweep; it also carries the corner ghosts the prolongation reads
    if (L.tiled) STICKY(m, exch_split(m, *L.da, *L.sg, nl, corners_last));
    return;
  }
  if (block8_ok(m, L)) {
    int n = 2 * nrelax, c = 0;
    const Lev *src = coarse;   // first pass: the correction is interpolated from the coarser level on the fly
    while (n > 0) {
      const int K = n < 8 ? n : 8;
      if (K == 1) { launch_relax_color(m->st, *L.da, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, c, L.walls, L.fine); break; }
      if (launch_relax_block8(m->st, *L.da, src ? *src->da : nullptr, src ? *src->sg : *L.sg, L.res, *L.da_alt, *L.sg, nl, *L.rc, L.walls, K, c, m->uniformS ? nullptr : L.S)) m->sticky = MSOM_ERR_ARG;
      std::swap(*L.da, *L.da_alt);
      src = nullptr; n -= K; c = (c + K) & 1;
      // periodic: the pass stored no ghost cell; boundary_level(da, l) = the wrapped copies, corners included (a following colour
      // pass reads them, and so does the prolongation to the next finer level)
      if (L.walls & WALL_PER) launch_split_wrap(m->st, *L.da, *L.sg, nullptr, nullptr, make_split(L.sg->nx, MARCH_HALO), nl, MARCH_HALO, 2);
    }
    return;
  }
  if (block_ok(m, L)) {
    for (; it + 2 <= nrelax; it += 2) {
      const bool pl = coarse && it == 0;
      if (prof && !pl) prof_begin(m, m->prof_block);
      launch_relax_block2(m->st, *L.da, pl ? *coarse->da : nullptr, pl ? *coarse->sg : *L.sg, L.res, *L.da_alt, *L.sg, nl, *L.rc, L.walls, L.fine);
      if (prof && !pl) prof_end(m, m->prof_block);
      std::swap(*L.da, *L.da_alt);
    }
  }
  for (; it < nrelax; it++) {
    const bool pl = coarse && it == 0;  // prolongation rides in the first red half-sweep
    if (prof && !pl) prof_begin(m, m->prof_sweep);
    for (int c = 0; c < 2; c++) {
      const int corners = corners_last && it == nrelax - 1 && c == 1;
      if (L.tiled && m->overlap && !(pl && c == 0) && !corners && L.sg->nx >= 32 && L.sg->ny >= 8) {
        // the outermost ring of the tile first; its halo exchange (communication stream) then runs beside the
        // interior cells (compute stream); same per-cell arithmetic, so the result does not change
        launch_relax_ring(m->st, *L.da, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, c, L.walls);
        comm_begin(m);
        STICKY(m, exch_split_raw(m, *L.da, *L.sg, nl, 0));
        launch_relax_color(m->st, *L.da, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, c, L.walls, L.fine, 1);
        comm_end(m);
        continue;
      }
      if (pl && c == 0) {
        if (prof) prof_begin(m, m->prof_redprol);
        launch_relax_red_prolong(m->st, *L.da, *coarse->da, *coarse->sg, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, L.walls);
        if (prof) prof_end(m, m->prof_redprol);
      } else
        launch_relax_color(m->st, *L.da, L.res, L.S, *L.sg, nl, *L.rc, m->uniformS, c, L.walls, L.fine);
      // boundary_level(da, l): the last exchange of the level also carries the corner ghosts
      // that the bilinear prolongation to the next finer level reads
      if (L.tiled) STICKY(m, exch_split(m, *L.da, *L.sg, nl, corners));
    }
    if (prof && !pl) prof_end(m, m->prof_sweep);
  }
}

// one level of the coarse-to-fine half of mg_cycle: initial guess (zero / prolongation), then
// the relaxations
static void level_solve(msom *m, Lev &L, const Lev *coarse, int nrelax, int corners_last) {
  const int nl = m->nl;
  bool fused = false;
  if (!coarse) hipMemsetAsync(*L.da, 0, L.sg->ls * nl * sizeof(double), m->st);
  else if (fuse_prolong(m, L, nrelax)) fused = true;
  else {
    launch_prolong(m->st, *coarse->da, *coarse->sg, *L.da, *L.sg, nl, L.walls);
    if (L.tiled) STICKY(m, exch_split(m, *L.da, *L.sg, nl, 0));
  }
  relax_sweeps(m, L, fused ? coarse : nullptr, nrelax, corners_last);
}

// coarse-to-fine part of mg_cycle, mspg/elliptic.h:53-89 (minlevel = 1): restriction of the
// residual to all levels (level 1 may already come out of the fused residual kernel), then
// prolongation + nrelax relaxations per level.  With tiles, the levels >= agg_level are solved
// on the gathered global coarse grid by every rank (identical arithmetic, no halo traffic).
static void mg_cycle_levels(msom *m, int nrelax, int first_restrict) {
  const int nl = m->nl, kc = m->agg_level >= 0 ? m->agg_level : m->nlev;
  const int kg = m->mgc_first;  // levels >= kg (of the gathered grid if kc < nlev, else of this tile): one launch
  const bool glob = kc < m->nlev;
  // tile levels: restrict down to the gather level / to the finest level of the one-launch group
  const int rmax = glob ? kc : (kg >= 0 ? kg : m->nlev - 1);
  {
    // the chain in launches of up to 5 levels each (k_restrict_pyramid, round 3) where a tile of 2^n cells a side of the chain's finest
    // level exists on every level down; a single restriction keeps its own kernel
    int k = first_restrict;
    const int kend = rmax < m->nlev - 1 ? rmax : m->nlev - 1;
    while (k <= kend) {
      int n = kend - k + 1;
      if (n > 5) n = 5;
      const SplitGeom &fg = m->sg[k - 1];
      while (n > 1 && (fg.nx % (1 << n) || fg.ny % (1 << n))) n--;
      if (n >= 2 && m->restrict_pyr) {
        SplitGeom g[6];
        double *out[5];
        for (int q = 0; q <= n; q++) g[q] = m->sg[k - 1 + q];
        for (int q = 0; q < n; q++) out[q] = m->res[k + q];
        launch_restrict_pyramid(m->st, m->res[k - 1], out, g, n, nl);
      } else {
        n = 1;
        launch_restrict(m->st, m->res[k - 1], m->sg[k - 1], m->res[k], m->sg[k], nl);
      }
      k += n;
    }
  }
  if (glob) {
    // gather the level-kc residual of all tiles, restrict it further on the global grid
    const SplitGeom &tg = m->sg[kc];
    const size_t cnt = (size_t)nl * tg.nx * tg.ny;
    launch_split_unpack(m->st, m->res[kc], tg, m->agg_send, nl);
    comm_begin(m);
    STICKY(m, comm_allgather(m->comm, m->agg_send, m->agg_recv, cnt));
    comm_end(m);
    launch_assemble_global(m->st, m->agg_recv, m->gres[0], m->gsg[0], nl, tg.nx, tg.ny, m->px);
    const int gtop = kg >= 0 ? kg : m->nlev - 1;  // coarsest level restricted by its own launch
    for (int k = kc + 1; k <= gtop; k++) launch_restrict(m->st, m->gres[k - 1 - kc], m->gsg[k - 1 - kc], m->gres[k - kc], m->gsg[k - kc], nl);
    if (kg >= 0) {
      launch_mg_coarse(m->st, m->d_cargs, nrelax, nl, m->uniformS, m->mgc_lean);
      if (hipGetLastError() != hipSuccess && !m->sticky) m->sticky = MSOM_ERR_HIP;
    }
    for (int k = (kg >= 0 ? kg : m->nlev) - 1; k >= kc; k--) {
      Lev L = glob_lev(m, k);
      if (k == m->nlev - 1) level_solve(m, L, nullptr, nrelax, 0);
      else { Lev C = glob_lev(m, k + 1); level_solve(m, L, &C, nrelax, 0); }
    }
    // this rank's tile of the level-kc correction, with its ghost ring
    launch_extract_tile(m->st, m->gda[0], m->gsg[0], m->da[kc], tg, nl, m->ix * tg.nx, m->iy * tg.ny);
  } else if (kg >= 0) {
    launch_mg_coarse(m->st, m->d_cargs, nrelax, nl, m->uniformS, m->mgc_lean);
    if (hipGetLastError() != hipSuccess && !m->sticky) m->sticky = MSOM_ERR_HIP;
  }
  for (int k = (glob ? kc : (kg >= 0 ? kg : m->nlev)) - 1; k >= 0; k--) {
    Lev L = tile_lev(m, k);
    if (k == m->nlev - 1) level_solve(m, L, nullptr, nrelax, k > 0);
    else { Lev C = tile_lev(m, k + 1); level_solve(m, L, &C, nrelax, k > 0); }
  }
}

static void clear_graphs(msom *m) {
  for (auto &kv : m->cyc_graph) (void)hipGraphExecDestroy(kv.second);
  m->cyc_graph.clear();
}
// the cycle through a captured graph where that is possible (see use_graph); same launches, same order, same arguments
static void mg_cycle(msom *m, int nrelax, int first_restrict) {
  bool ok = m->use_graph && m->nranks == 1 && !m->profile;   // (corr_req only acts inside a marching pass, excluded below)
  for (int k = 0; ok && k < m->nlev; k++) {
    Lev L = tile_lev(m, k);
    if (march_ok(m, L) || block_ok(m, L) || block8_ok(m, L)) ok = false;   // those passes ping-pong between two buffers: pointers differ from cycle to cycle
  }
  if (!ok) { mg_cycle_levels(m, nrelax, first_restrict); return; }
  const long key = (long)nrelax * 8 + first_restrict;
  auto it = m->cyc_graph.find(key);
  if (it == m->cyc_graph.end()) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    if (hipStreamBeginCapture(m->st, hipStreamCaptureModeThreadLocal) != hipSuccess) { m->use_graph = 0; mg_cycle_levels(m, nrelax, first_restrict); return; }
    mg_cycle_levels(m, nrelax, first_restrict);
    if (hipStreamEndCapture(m->st, &g) != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      m->use_graph = 0;                 // nothing was executed during the capture: run the cycle eagerly now
      mg_cycle_levels(m, nrelax, first_restrict);
      return;
    }
    (void)hipGraphDestroy(g);
    it = m->cyc_graph.emplace(key, ex).first;
  }
  if (hipGraphLaunch(it->second, m->st) != hipSuccess && !m->sticky) m->sticky = MSOM_ERR_HIP;
}

static void residual(msom *m, const double *a, const double *b, int slot, int want_sum) {
  if (m->profile) prof_begin(m, m->prof_resid);
  launch_residual(m->st, a, b, m->f[MSOM_S], m->g, m->res[0], m->sg[0], m->nl, m->rc[0], m->uniformS, m->d_scal + slot, m->partial, want_sum);
  if (m->profile) prof_end(m, m->prof_resid);
}
// fused variants (kernels_mg.hip k_residual2); mode bits 1 = CORRECT, 2 = WRITE, 4 = RESTRICT
// the pre-cycle residual pass also restricts to level 2 (round 3: the launch of k_restrict that read the level-1 residual back, 53 us at
// 4096^2 x 6, becomes 1/16 w of extra stores): the block of 4 rows x 128 cells holds whole level-2 cells when ny % 4 == 0 and nx % 4 == 0
static bool restrict2_ok(const msom *m) { return m->restrict2 && m->mg_fused && m->nlev > 2 && m->g.ny % 4 == 0 && m->sg[0].hk % 2 == 0; }
static void residual2(msom *m, int mode, const double *b, int slot, int want_sum) {
  ProfSlot &which = (mode & 8) ? m->prof_resmax : (mode & 1) ? m->prof_rescorr : m->prof_respre;
  if (m->profile) { prof_begin(m, m->prof_resid); prof_begin(m, which); }
  const bool r2 = (mode & 4) && restrict2_ok(m);
  launch_residual2(m->st, mode, m->f[MSOM_PSI], m->da[0], m->psi_alt, b, m->f[MSOM_S], m->g, m->res[0], m->sg[0],
                   m->nlev > 1 ? m->res[1] : nullptr, m->sg[m->nlev > 1 ? 1 : 0], m->nl, m->rc[0], m->uniformS, m->walls, m->d_scal + slot,
                   m->partial, want_sum, m->partial_umax, m->d_scal + SC_UMAX, m->umax_clean, r2 ? m->res[2] : nullptr, r2 ? &m->sg[2] : nullptr);
  if (mode & (1 | 8)) m->umax_clean = 0;
  if (m->profile) { prof_end(m, m->prof_resid); prof_end(m, which); }
}

// max-residual slots (RES0, RES1) and the rhs sum (BSUM) -> host, reduced over the tiles
// (the reference's foreach(reduction(max:maxres)) / reduction(+:sum) are MPI all-reduces)
static int read_residuals(msom *m) {
  if (m->nranks == 1) {
    // one copy of the whole scalar block (sums, residuals, max|u|, the device's dt) and one wait
    if (m->dbg_nosync) return MSOM_OK;
    if (m->spec_launched) {   // the scalars were published by k_step_dt in front of the speculative pass: spin on its sequence word
      volatile long *seq = reinterpret_cast<volatile long *>(m->h_pub + SC_COUNT);
      const double t0 = wall_seconds();
      while (*seq != m->pub_seq) {
        if (wall_seconds() - t0 > 5.) {   // the device is stuck or faulted: let the runtime report it
          HIPCHK(hipStreamSynchronize(m->st));
          if (*seq != m->pub_seq) { msom_set_error("the device did not publish the solver's scalars"); return MSOM_ERR_HIP; }
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      memcpy(m->h_scal, m->h_pub, SC_COUNT * sizeof(double));
      return MSOM_OK;
    }
    HIPCHK(hipMemcpyAsync(m->h_scal, m->d_scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    return MSOM_OK;
  }
  int r = reduce_scal(m, SC_RES0, 2 + m->nl, RED_MAX);  // RES0, RES1, UMAX[nl]
  if (r) return r;
  // mgstats.sum is informational (the reference never prints it, msqg/qg.h:61): with tiles it
  // stays the local tile's sum unless "mg_global_sum" asks for the extra all-reduce
  if (m->nranks > 1 && !m->mg_global_sum) {
    HIPCHK(hipMemcpyAsync(m->h_scal + SC_BSUM, m->d_scal + SC_BSUM, sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    return MSOM_OK;
  }
  return reduce_scal(m, SC_BSUM, 1, RED_SUM);
}

// mg_solve, mspg/elliptic.h:145-229, called as poisson_layer does (msqg/poisson_layer.h:290-303)
// on a = psi (natural field with valid ghosts, warm start) and b (natural field).
// Fused path (default): the pre-cycle residual also produces the level-1 restriction, the
// correction a += da is folded into the post-cycle residual (written to a second psi buffer
// that then becomes psi), and the post-cycle pass only produces max|res|; the residual field
// is regenerated only if another cycle turns out to be needed.
static int mg_solve(msom *m, const double *b, msom_mgstats *s) {
  const Params &p = m->p;
  const bool fused = m->mg_fused && m->nlev > 1;
  s->i = 0; s->nrelax = 4;
  // one fill for every accumulator of the solve: sums, max|res| before / after the first cycle, max|u| per layer
  HIPCHK(hipMemsetAsync(m->d_scal, 0, (SC_UMAX + MSOM_MAXNL) * sizeof(double), m->st));
  m->umax_clean = 1;
  m->umax_ready = 0;
  const bool have_res = fused && m->res_ready >= 0 && b == m->f[m->res_ready];
  m->res_ready = -1;
  if (have_res) {  // res[0], res[1], max|res| and the partial sums of b came out of the tendency pass
    HIPCHK(hipMemcpyAsync(m->d_scal + SC_RES0, m->d_scal + SC_RESF, sizeof(double), hipMemcpyDeviceToDevice, m->st));
    launch_sum_final(m->st, m->partial_rr, m->d_scal + SC_BSUM, rhs_pipe_blocks(m->g));
  } else if (fused) {
    residual2(m, 2 | 4, b, SC_RES0, 1);
    launch_sum_final(m->st, m->partial, m->d_scal + SC_BSUM, residual2_blocks(m->g));
  } else {
    residual(m, m->f[MSOM_PSI], b, SC_RES0, 1);
    launch_sum_final(m->st, m->partial, m->d_scal + SC_BSUM, partial_count(m->g));
  }
  bool have_first = false;
  double resb = 0;
  if (p.nitermin < 1) {  // need the initial residual before deciding on the first cycle
    int rr = read_residuals(m);
    if (rr) return rr;
    resb = s->resb = s->resa = m->h_scal[SC_RES0];
    s->sum = m->h_scal[SC_BSUM];
    have_first = true;
  }
  for (s->i = 0; s->i < p.nitermax && (s->i < p.nitermin || s->resa > p.tolerance); s->i++) {
    m->corr_req = fused && m->march_correct;
    m->corr_done = 0;
    // restrictions still to do: from level 1 (plain path), 2 (the residual pass restricted once) or 3 (twice; not when the residual
    // came out of the tendency pass, option rhs_resid)
    mg_cycle(m, s->nrelax, fused ? ((restrict2_ok(m) && !(have_res && s->i == 0)) ? 3 : 2) : 1);
    m->corr_req = 0;
    if (s->i > 0) HIPCHK(hipMemsetAsync(m->d_scal + SC_RES1, 0, sizeof(double), m->st));
    if (m->corr_done) {  // a_new = a + da already sits in psi_alt (last smoother pass): boundary(a), then max |res|, max |u|
      std::swap(m->f[MSOM_PSI], m->psi_alt);
      if (m->nranks > 1) STICKY(m, exch_nat(m, m->f[MSOM_PSI], m->nl, m->bc, 1));
      else if (m->bc == BC_PERIODIC) launch_fill_periodic(m->st, m->f[MSOM_PSI], m->g, m->nl, 1);
      else launch_fill_ghost(m->st, m->f[MSOM_PSI], m->g, m->nl, m->bc, m->walls);   // the LDS-DMA pass leaves the wall ghosts to this
      residual2(m, 8, b, SC_RES1, 0);
      m->umax_ready = 1;
    } else if (fused) {
      residual2(m, 1, b, SC_RES1, 0);            // a_new = a + da -> psi_alt, max |res(a_new)|, max |u(a_new)|
      std::swap(m->f[MSOM_PSI], m->psi_alt);
      m->umax_ready = 1;
      if (m->nranks > 1) STICKY(m, exch_nat(m, m->f[MSOM_PSI], m->nl, m->bc, 1));
    } else {
      launch_correct(m->st, m->f[MSOM_PSI], m->g, m->da[0], m->sg[0], m->nl, m->walls);
      if (m->nranks > 1) STICKY(m, exch_nat(m, m->f[MSOM_PSI], m->nl, m->bc, 1));  // boundary(a)
      residual(m, m->f[MSOM_PSI], b, SC_RES1, 0);
    }
    if (s->i == 0 && m->spec_hook && m->umax_ready && m->nranks == 1) {   // the tendency pass, queued before the host knows the residual
      m->spec_hook([&]() {});
      m->spec_launched = 1;
    }
    int rr = read_residuals(m);
    if (rr) { m->spec_launched = 0; return rr; }
    if (m->spec_launched) {   // its output counts only if this solve ends here
      const double ra = m->h_scal[SC_RES1];
      m->spec_valid = !(s->i + 1 < p.nitermax && (s->i + 1 < p.nitermin || ra > p.tolerance));
      m->spec_launched = 0;
    }
    if (!have_first) {
      resb = s->resb = m->h_scal[SC_RES0];
      s->sum = m->h_scal[SC_BSUM];
      have_first = true;
    }
    s->resa = m->h_scal[SC_RES1];
    if (s->resa > p.tolerance) {
      if (resb / s->resa < 1.2 && s->nrelax < 100) s->nrelax++;
      else if (resb / s->resa > 10 && s->nrelax > 2) s->nrelax--;
    }
    resb = s->resa;
    // another cycle follows: it needs the residual field of the corrected a on levels 0 and 1
    if (fused && s->i + 1 < p.nitermax && (s->i + 1 < p.nitermin || s->resa > p.tolerance)) residual2(m, 2 | 4, b, SC_SCRATCH, 0);
  }
  if (!have_first) {  // nitermax == 0
    int rr = read_residuals(m);
    if (rr) return rr;
    s->resb = s->resa = m->h_scal[SC_RES0];
    s->sum = m->h_scal[SC_BSUM];
  }
  if (s->resa > p.tolerance && !m->quiet)
    fprintf(stderr, "WARNING: convergence not reached after %d iterations\n  res: %g sum: %g nrelax: %d\n", s->i, s->resa, s->sum, s->nrelax);
  return MSOM_OK;
}

// invertq, msqg/qg.h:114-163 (the trailing boundary(pol) is already done by the correction)
static int invertq(msom *m, const double *q) {
  return mg_solve(m, q, &m->mg);
}

// ------------------------------------------------------------------ RHS

static double limiter(msom *m, double umax, double dtmax) {
  // timestep() [Basilisk; algorithm text newqg/qg.h:202-219]
  const double D = m->p.L0 / m->gnx;
  dtmax /= m->p.CFL;
  if (umax != 0.) {
    const double dt = D / umax;
    if (dt < dtmax) dtmax = dt;
  }
  dtmax *= m->p.CFL;
  if (dtmax > m->previous) dtmax = (m->previous + 0.1 * dtmax) / 1.1;
  m->previous = dtmax;
  return dtmax;
}

static void comp_del2(msom *m, int in, int out, double add, double fac) {
  const double D = m->p.L0 / m->gnx;
  launch_del2(m->st, m->f[in], m->f[out], m->g, m->nl, add, fac, D);
  fill_bc(m, out);
  if (m->p.sbc > 0) launch_slip_bc(m->st, m->f[in], m->f[out], m->g, m->nl, m->p.sbc / ((0.5 * m->p.sbc + 1) * D * D), m->walls);
}
static void comp_stretch(msom *m, int in, int out, double add, double fac) {
  launch_stretch(m->st, m->f[in], m->f[out], m->f[MSOM_S], m->g, m->nl, add, fac, m->lc);
  fill_bc(m, out);
}

// tendency terms after the inversion: comp_del2, advection_pv, dissip, ekman_friction,
// surface_forcing, [qforcing], [bottom_topography]   (msqg/qg.h:622-630 / qg_bfn.h:67-76)
// adv_out >= 0 asks for q[adv_out] = q[adv_in] + adv_dt * dq in the same pass (the corrector's
// advance_qg); *advanced tells the caller whether that happened (fused path) or not.
static int stoch_prepare(msom *m, double dt, double *dts);
static int rhs_terms(msom *m, int qfield, int dqfield, int with_qforcing, double iRe, double iRe4, double Eks, double Ekb, int adv_out = -1,
                     int adv_in = -1, double adv_dt = 0., int *advanced = nullptr) {
  if (advanced) *advanced = 0;
  const Params &p = m->p;
  const double D = p.L0 / m->gnx;
  const int nl = m->nl;
  // stochastic runs (msqg/qg_stochastic.h) ride in the same kernel when the advance does: the relaxation -q/tau and the
  // noise are linear in fields that exist before the pass, so the finalisation of the pass reads them next to q_in.  The
  // validation build keeps the reference's operation order (separate kernels) instead.
#ifdef MSOM_STRICT
  const bool stoch_fused = false;
#else
  const bool stoch_fused = m->stochastic && m->stoch_fused && m->adv_fused && adv_out >= 0 && m->rhs_variant == 6 && !m->rhs_resid;
#endif
  if (m->fused && m->nl <= MSOM_FASTNL && !m->have_pg && !m->have_zpg && !m->flag_topo && (!m->stochastic || stoch_fused) &&
      (m->nranks == 1 || (m->nx >= 4 && m->ny >= 4))) {
    // tiles: the fused kernel needs psi on a 3-cell halo (zeta on 2, lap(zeta) on 1).  With the one-layer-per-wavefront
    // kernel (option overlap) the wavefronts that read no halo cell are queued first and run beside the exchange, the
    // strips and chunks along the tile edges follow once it has arrived (each cell is written by exactly one wavefront)
    if (m->bc == BC_PERIODIC && m->nranks == 1) launch_fill_periodic(m->st, m->f[MSOM_PSI], m->g, nl, 3);
    if (!m->adv_fused) adv_out = -1;
    // the advance rides along: the pass can also emit the first residual of the inversion of q[adv_out]
    // the residual by-product exists only in the LDS-tile kernel: asking for it selects that kernel
    const int variant = (m->rhs_resid && m->rhs_variant == 6) ? 1 : m->rhs_variant;
    const bool use_rr = adv_out >= 0 && m->rhs_resid && variant == 1 && m->mg_fused && m->nlev > 1 && m->bc != BC_PERIODIC;
    RhsResid rr;
    if (use_rr) {
      rr.res = m->res[0]; rr.res_c = m->res[1]; rr.res_max = m->d_scal + SC_RESF; rr.bsum_partial = m->partial_rr;
      rr.sg = m->sg[0]; rr.cg = m->sg[1];
    }
    // the psi halo exchange of a tile around the launch(es) of the pass; splittable: the kernel takes a region argument
    auto with_psi_halo = [&](bool splittable, auto launch) {
      if (m->nranks > 1 && m->overlap && splittable) {
        comm_begin(m); m->comm_hold = 1;
        launch(1);
        STICKY(m, exch_nat(m, m->f[MSOM_PSI], nl, m->bc, 3));
        m->comm_hold = 0; comm_end(m);
        launch(2);
      } else {
        if (m->nranks > 1) STICKY(m, exch_nat(m, m->f[MSOM_PSI], nl, m->bc, 3));
        launch(0);
      }
    };
    // one pass over psi: zeta, Jacobians, beta, dissipation, drag, forcing, max|u| (kernels_fused.hip)
    if (stoch_fused) {
      extern int g_rhs_dbg;
      double dts;
      int r = stoch_prepare(m, adv_dt, &dts);
      if (r) return r;
      m->res_ready = -1;
      // the relaxation -q_stage / tau and the noise are read in the finalisation of the pass: q_out = q_in - (dt / tau) q_stage + dts n + dt dq
      prof_begin(m, m->prof_rhs);
      with_psi_halo(true, [&](int region) {
        launch_rhs_lpw(m->st, m->f[MSOM_PSI], m->f[MSOM_S], m->f[MSOM_QFORC], m->d_wind, nullptr, m->g, nl, m->walls & WALL_ALL, m->uniformS,
                       m->rc[0].S, with_qforcing && m->have_qforc, D, p.beta, iRe, iRe4, Eks / (p.Rom * 2 * m->dhf[0]),
                       Ekb / (p.Rom * 2 * m->dhf[nl - 1]), p.sbc > 0 ? p.sbc / ((0.5 * p.sbc + 1) * D * D) : 0., m->lc, m->f[adv_in],
                       m->f[adv_out], adv_dt, g_rhs_dbg >> 8, 1, m->f[qfield], m->f[MSOM_NOISE], -adv_dt * p.itr_stoch, dts, region);
      });
      prof_end(m, m->prof_rhs);
      if (advanced) *advanced = 1;
      return MSOM_OK;
    }
    prof_begin(m, m->prof_rhs);
    with_psi_halo(variant == 6, [&](int region) {
      launch_rhs_fused(m->st, m->f[MSOM_PSI], m->f[MSOM_S], m->f[MSOM_QFORC], m->d_wind, m->f[dqfield], nullptr,
                       nullptr, m->g, nl, m->walls & WALL_ALL, m->uniformS, m->rc[0].S, with_qforcing && m->have_qforc, D, p.beta, iRe,
                       iRe4, Eks / (p.Rom * 2 * m->dhf[0]), Ekb / (p.Rom * 2 * m->dhf[nl - 1]),
                       p.sbc > 0 ? p.sbc / ((0.5 * p.sbc + 1) * D * D) : 0., m->lc, variant, adv_out >= 0 ? m->f[adv_in] : nullptr,
                       adv_out >= 0 ? (m->adv_out_override ? m->adv_out_override : m->f[adv_out]) : nullptr, adv_dt, use_rr ? &rr : nullptr, region,
                       variant == 6 ? m->dt_dev : nullptr);
    });
    prof_end(m, m->prof_rhs);
    if (use_rr) m->res_ready = adv_out;
    if (advanced && adv_out >= 0) *advanced = 1;
    return MSOM_OK;
  }
  HIPCHK(hipMemsetAsync(m->f[dqfield], 0, m->g.ls * nl * sizeof(double), m->st));  // updates = 0, msqg/qg.h:611-613
  comp_del2(m, MSOM_PSI, MSOM_ZETA, 0., 1.0);
  launch_advection(m->st, m->f[MSOM_ZETA], m->f[MSOM_PSI], m->f[MSOM_PSIPG], m->f[MSOM_ZETAPG], m->f[MSOM_S], m->f[qfield], m->f[dqfield],
                   m->g, nl, m->have_pg, m->have_zpg, m->stochastic, D, p.beta, p.itr_stoch, m->lc);
  // dissip :407-422 (terms with a zero coefficient add exactly 0 and are skipped)
  if (iRe != 0) comp_stretch(m, MSOM_ZETA, dqfield, 1., iRe);
  if (iRe != 0 || iRe4 != 0) comp_del2(m, MSOM_ZETA, MSOM_TMP, 0., 1.);
  if (iRe != 0) launch_axpy(m->st, m->f[dqfield], m->f[MSOM_TMP], m->g, nl, iRe);
  if (iRe4 != 0) {
    comp_stretch(m, MSOM_TMP, dqfield, 1., iRe4);
    comp_del2(m, MSOM_TMP, dqfield, 1., iRe4);
  }
  launch_forcing(m->st, m->f[MSOM_ZETA], m->f[MSOM_PSI], m->f[MSOM_QFORC], m->f[MSOM_TOPO], m->f[MSOM_RO], m->d_wind, m->f[dqfield], m->g,
                 nl, with_qforcing && m->have_qforc, m->flag_topo, Eks / (p.Rom * 2 * m->dhf[0]), Ekb / (p.Rom * 2 * m->dhf[nl - 1]), D,
                 m->dhf[nl - 1]);
  return MSOM_OK;
}

static int tracer_update(msom *m, int cfield);

// first half of update_qg (msqg/qg.h:621-623): invert q -> psi, then the CFL part of
// advection_pv (:383-391): 2*nl limiter calls (psi_l then psipg_l) sharing one static
// `previous`.  max|u| of psi comes out of the solver's last pass (k_residual2<CORRECT>), so dt
// is known BEFORE the tendency pass and the advance can ride in it.
// The same on the device, for the speculative tendency pass of the first RK stage: the 2 nl limiter calls of advection_pv
// (timestep(), newqg/qg.h:202-219) on max|u| of psi (d_scal) and of psi_pg, then dtnext() of run().  Same operations in the
// same order as limiter() / dtnext() on the host, no contraction, correctly rounded divisions: the same numbers.
// The kernel also PUBLISHES the scalar block to the host: 64 threads copy the 64 slots into coherent pinned memory, a system-scope
// fence, then a sequence word the host spins on -- no copy command and no event between the residual pass and the tendency pass.
struct StepDtArgs { double umax_pg[MSOM_MAXNL]; double D, CFL, previous, dtmax, t, tnext; int nl, with_dt; long seq; };
__global__ void __launch_bounds__(64) k_step_dt(double *scal, StepDtArgs a, double *pub) {
#pragma clang fp contract(off)
  if (threadIdx.x == 0 && a.with_dt) {
  const double *umax = scal + SC_UMAX;
  double *out = scal + SC_SPEC;
  double dtmax = a.dtmax, previous = a.previous;
  for (int k = 0; k < 2 * a.nl; k++) {
    const double u = (k & 1) ? a.umax_pg[k >> 1] : umax[k >> 1];
    dtmax /= a.CFL;
    if (u != 0.) {
      const double dt = a.D / u;
      if (dt < dtmax) dtmax = dt;
    }
    dtmax *= a.CFL;
    if (dtmax > previous) dtmax = (previous + 0.1 * dtmax) / 1.1;
    previous = dtmax;
  }
  double dt = dtmax, tnext = a.tnext;
  if (tnext != HUGE_VAL && tnext > a.t) {
    const unsigned int n = (unsigned int)((tnext - a.t) / dt);
    if (n == 0) dt = tnext - a.t;
    else {
      const double dt1 = (tnext - a.t) / n;
      if (dt1 > dt * (1. + 1e-9)) dt = (tnext - a.t) / (n + 1);
      else if (dt1 < dt) dt = dt1;
      tnext = a.t + dt;
    }
  } else
    tnext = a.t + dt;
  out[0] = dtmax; out[1] = dt; out[2] = tnext; out[3] = dt / 2.;
  }
  __syncthreads();
  if (pub) {
    pub[threadIdx.x] = scal[threadIdx.x];      // SC_COUNT = 64 slots
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __atomic_store_n(reinterpret_cast<long *>(pub + SC_COUNT), a.seq, __ATOMIC_RELEASE);
  }
}
static_assert(SC_COUNT == 64, "k_step_dt publishes one slot per thread");

static double solve_and_dt(msom *m, int qfield, double dtmax) {
  const int nl = m->nl;
  m->spec_valid = 0;
  if (invertq(m, m->f[qfield])) return -1;
  if (m->spec_valid && m->dt_dev) {   // the device ran the limiter (first RK stage): adopt its state
    m->previous = m->h_scal[SC_SPEC];
    return m->h_scal[SC_SPEC];
  }
  if (!m->umax_ready) {
    launch_umax(m->st, m->f[MSOM_PSI], m->partial_umax, m->d_scal + SC_UMAX, m->g, nl, m->p.L0 / m->gnx);
    if (reduce_scal(m, SC_UMAX, nl, RED_MAX)) return -1;
  }
  if (m->sticky) return -1;
  for (int l = 0; l < nl; l++) {
    dtmax = limiter(m, m->h_scal[SC_UMAX + l], dtmax);
    dtmax = limiter(m, m->umax_pg[l], dtmax);
  }
  return dtmax;
}

// update_qg, msqg/qg.h:609-650
static double update_qg(msom *m, int qfield, int dqfield, double dtmax) {
  const Params &p = m->p;
  dtmax = solve_and_dt(m, qfield, dtmax);
  if (dtmax < 0) return -1;
  if (rhs_terms(m, qfield, dqfield, 1, p.iRe, p.iRe4, p.Eks, p.Ekb)) return -1;
  if (p.nptr > 0 && tracer_update(m, qfield == MSOM_QPRED ? MSOM_PTR_PRED : MSOM_PTR)) return -1;
  return dtmax;
}

// reference-exact noise: Box-Muller on the serial rand() stream in foreach order
// (x outer, y inner, layer innermost), msqg/qg_stochastic.h:9,117-126
static int generate_noise_host(msom *m) {
  const int nl = m->nl, nx = m->nx, ny = m->ny;
  std::vector<double> sig((size_t)nl * nx * ny), n((size_t)nl * nx * ny);
  int r = download(m, MSOM_SIGMA, sig.data());
  if (r) return r;
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++)
      for (int l = 0; l < nl; l++) {
        const double a = sqrt(-2. * log(((double)(rand()) + 1.) / ((double)(RAND_MAX) + 2.)));
        const double nn = a * cos(2 * M_PI * rand() / (double)RAND_MAX);
        const size_t k = ((size_t)l * ny + j) * nx + i;
        n[k] = m->p.amp_stoch * sig[k] * nn;
      }
  return upload(m, MSOM_NOISE, n.data());
}

// the stochastic part of advance_qg (msqg/qg_stochastic.h:128-138): every other call draws new noise and uses
// sqrt(dt)/sqrt(2), computed in float as the reference does
static int stoch_prepare(msom *m, double dt, double *dts) {
  // refused before any state changes: one rand() stream per process would repeat the same numbers on every tile
  if (m->noise_mode == 0 && m->nranks > 1) {
    msom_set_error("stochastic forcing on tiles: the serial rand() stream (noise_mode = 0) exists on a single tile only, use noise_mode = 1");
    return MSOM_ERR_STATE;
  }
  m->corrector_step = (m->corrector_step + 1) % 2;
  float fdts = sqrt(dt);
  if (m->corrector_step) {
    if (m->noise_mode == 0) {  // reference-exact serial rand() stream
      int r = generate_noise_host(m);
      if (r) return r;
    } else  // counter-based device generator
      launch_noise(m->st, m->f[MSOM_NOISE], m->f[MSOM_SIGMA], m->g, m->nl, m->p.amp_stoch, m->seed, m->noise_draw++, m->ix * m->nx,
                   m->iy * m->ny, m->gnx);
    fdts = fdts / sqrt(2);
  }
  *dts = fdts;
  return MSOM_OK;
}

// advance_qg, msqg/qg.h:594-606 / msqg/qg_stochastic.h:128-149
static int advance_qg(msom *m, int out, int in, int dq, double dt) {
  m->res_ready = -1;
  const double *noise = nullptr;
  double dts = 0;
  if (m->stochastic) {
    int r = stoch_prepare(m, dt, &dts);
    if (r) return r;
    noise = m->f[MSOM_NOISE];
  }
  launch_advance(m->st, m->f[out], m->f[in], m->f[dq], noise, m->g, m->nl, dt, dts);
  fill_bc(m, out);
  return MSOM_OK;
}

#define NEED_CONST(m)                                                    \
  do {                                                                   \
    if (!(m)) return MSOM_ERR_ARG;                                       \
    if (!(m)->const_set) {                                               \
      int r__ = msom_set_const(m);                                       \
      if (r__) return r__;                                               \
    }                                                                    \
  } while (0)

extern "C" double msom_update(msom_t *m, const double *q, double *dqdt, double dtmax) {
  if (!m) return -1;
  if (!m->const_set && msom_set_const(m)) return -1;
  int qf = MSOM_Q;
  if (q) {
    if (upload(m, MSOM_QPRED, q)) return -1;
    qf = MSOM_QPRED;
  }
  double d = update_qg(m, qf, MSOM_DQ, dtmax);
  if (d < 0) return d;
  if (dqdt && download(m, MSOM_DQ, dqdt)) return -1;
  return d;
}

extern "C" int msom_advance(msom_t *m, double *qout, const double *qin, const double *dqdt, double dt) {
  NEED_CONST(m);
  int in = MSOM_Q, out = MSOM_Q, r;
  if (qin) {
    if ((r = upload(m, MSOM_QPRED, qin))) return r;
    in = out = MSOM_QPRED;
  }
  if (dqdt && (r = upload(m, MSOM_DQ, dqdt))) return r;
  if ((r = advance_qg(m, out, in, MSOM_DQ, dt))) return r;
  if (qout) return download(m, out, qout);
  return sync_stream(m);
}

extern "C" int msom_invertq(msom_t *m, const double *q, double *psi, msom_mgstats *st) {
  NEED_CONST(m);
  int qf = MSOM_Q, r;
  if (q) {
    if ((r = upload(m, MSOM_QPRED, q))) return r;
    qf = MSOM_QPRED;
  }
  if (psi && (r = upload(m, MSOM_PSI, psi))) return r;
  if ((r = invertq(m, m->f[qf]))) return r;
  if (st) *st = m->mg;
  if (psi) return download(m, MSOM_PSI, psi);
  return sync_stream(m);
}

extern "C" int msom_comp_q(msom_t *m, const double *psi, double *q) {
  NEED_CONST(m);
  int r;
  if (psi && (r = upload(m, MSOM_PSI, psi))) return r;
  comp_del2(m, MSOM_PSI, MSOM_Q, 0., 1.);
  comp_stretch(m, MSOM_PSI, MSOM_Q, 1., 1.);
  if (q) return download(m, MSOM_Q, q);
  return sync_stream(m);
}

extern "C" int msom_last_mgstats(msom_t *m, msom_mgstats *st) {
  if (!m || !st) return MSOM_ERR_ARG;
  *st = m->mg;
  return MSOM_OK;
}

// ------------------------------------------------------------------ pystep_bfn & co

static int check_shape(msom *m, int a, int b, int c) {
  if (a != m->nl || b != m->ny || c != m->nx) {
    msom_set_error("array shape (%d,%d,%d) != (nl,N,N) = (%d,%d,%d)", a, b, c, m->nl, m->ny, m->nx);
    return MSOM_ERR_ARG;
  }
  return MSOM_OK;
}

// msqg/qg_bfn.h:21-80
extern "C" int pystep_bfn(msom_t *m, double *varin_py, int len1, int len2, int len3, double *tend_py, int len4, int len5, int len6,
                          double direction, int vartype) {
  NEED_CONST(m);
  if (check_shape(m, len1, len2, len3) || check_shape(m, len4, len5, len6)) return MSOM_ERR_ARG;
  Params &p = m->p;
  int r;
  if (direction > 0) {
    p.iRe = p.Re == 0 ? 0. : 1 / p.Re;
    p.iRe4 = p.Re4 == 0 ? 0. : -1 / p.Re4;
    p.Eks = fabs(p.Eks); p.Ekb = fabs(p.Ekb);
  } else {
    p.iRe = p.Re == 0 ? 0. : -1 / p.Re;
    p.iRe4 = p.Re4 == 0 ? 0. : 1 / p.Re4;
    p.Eks = -fabs(p.Eks); p.Ekb = -fabs(p.Ekb);
  }
  if (vartype != 1) HIPCHK(hipMemsetAsync(m->f[MSOM_DQ], 0, m->g.ls * m->nl * sizeof(double), m->st));  // reset_layer_var(bfn_tendl)
  if (vartype == 1) {
    if ((r = upload(m, MSOM_Q, varin_py))) return r;
    if (solve_and_dt(m, MSOM_Q, p.DT) < 0) return MSOM_ERR_HIP;  // invertq + the dt limiter inside advection_pv
    if ((r = rhs_terms(m, MSOM_Q, MSOM_DQ, 0, p.iRe, p.iRe4, p.Eks, p.Ekb))) return r;
  } else if (!m->quiet)
    fprintf(stdout, "temporary disabled psi tendency\n");  // msqg/qg_bfn.h:48
  return download(m, MSOM_DQ, tend_py);
}
// msqg/qg_bfn.h:85-93
extern "C" int pyq2p(msom_t *m, double *po_py, int len7, int len8, int len9, double *qo_py, int len10, int len11, int len12) {
  NEED_CONST(m);
  if (check_shape(m, len7, len8, len9) || check_shape(m, len10, len11, len12)) return MSOM_ERR_ARG;
  int r;
  HIPCHK(hipMemsetAsync(m->f[MSOM_PSI], 0, m->g.ls * m->nl * sizeof(double), m->st));
  if ((r = upload(m, MSOM_Q, qo_py))) return r;
  if ((r = invertq(m, m->f[MSOM_Q]))) return r;
  return download(m, MSOM_PSI, po_py);
}
// msqg/qg_bfn.h:95-103
extern "C" int pyp2q(msom_t *m, double *po_py, int len13, int len14, int len15, double *qo_py, int len16, int len17, int len18) {
  NEED_CONST(m);
  if (check_shape(m, len13, len14, len15) || check_shape(m, len16, len17, len18)) return MSOM_ERR_ARG;
  return msom_comp_q(m, po_py, qo_py);
}

// ------------------------------------------------------------------ time loop

// dtnext() [Basilisk, SURVEY App. B]
static double dtnext(msom *m, double dt, double *tnext_out) {
  double tnext = m->tnext, t = m->t;
  if (tnext != HUGE_VAL && tnext > t) {
    unsigned int n = (unsigned int)((tnext - t) / dt);
    if (n == 0) dt = tnext - t;
    else {
      double dt1 = (tnext - t) / n;
      if (dt1 > dt * (1. + 1e-9)) dt = (tnext - t) / (n + 1);
      else if (dt1 < dt) dt = dt1;
      tnext = t + dt;
    }
  } else
    tnext = t + dt;
  *tnext_out = tnext;
  return dt;
}

// tracer part of update_qg (msqg/qg.h:634-647): dpdt = ptr_rhs(tracers, psi) with `updates` zeroed
static int tracer_update(msom *m, int cfield) {
  const int nt = m->nl * m->p.nptr;
  HIPCHK(hipMemsetAsync(m->f[MSOM_DPTR], 0, m->g.ls * nt * sizeof(double), m->st));
  launch_ptr_rhs(m->st, m->f[MSOM_PSI], m->f[cfield], m->f[MSOM_PTR_RELAX], m->f[MSOM_DPTR], m->g, m->nl, m->p.nptr, m->p.iPe, m->p.ptr_ir,
                 m->p.L0 / m->gnx);
  return MSOM_OK;
}
// tracer part of advance_qg (msqg/qg.h:597-605)
static int tracer_advance(msom *m, int out, int in, double dt) {
  launch_advance(m->st, m->f[out], m->f[in], m->f[MSOM_DPTR], nullptr, m->g, m->nl * m->p.nptr, dt, 0.);
  return fill_bc(m, out);
}

// one iteration of run() [Basilisk predictor-corrector.h]:
//   dt = dtnext(update(evolving, updates, DT)); advance(predictor, evolving, updates, dt/2);
//   update(predictor, updates, dt); advance(evolving, evolving, updates, dt)
// dt is known right after each inversion, so both advances ride in the tendency kernel
// (q_out = q_in + dt * dq) whenever the fused kernel applies; dq is then never stored.
// can a stage's tendency pass be queued speculatively?  One tile, the fused one-layer-per-wavefront kernel with the advance
// folded in, max|u| out of the solver's last pass, nothing with side effects in the pass (noise draws, tracers)
static bool spec_ok(msom *m) {
  return m->async_solve && m->nranks == 1 && m->fused && m->adv_fused && m->rhs_variant == 6 && !m->rhs_resid && m->mg_fused && m->nlev > 1 &&
         m->nl <= MSOM_FASTNL && !m->have_pg && !m->have_zpg && !m->flag_topo && !m->stochastic && m->p.nptr == 0 && m->p.nitermin >= 1 && !m->dbg_nosync;
}

extern "C" int msom_step(msom_t *m, double *dt_used) {
  NEED_CONST(m);
  const Params &p = m->p;
  double tnext;
  int r, advanced = 0;
  const bool spec = spec_ok(m);
  int spec_rc = 0;
  if (spec) {   // stage 1: dt on the device, predictor = q + dt/2 dq
    m->dt_dev = m->d_scal + SC_SPEC + 3;
    m->spec_hook = [&](const std::function<void()> &host_read) {
      StepDtArgs a;
      for (int l = 0; l < MSOM_MAXNL; l++) a.umax_pg[l] = m->umax_pg[l];
      a.D = p.L0 / m->gnx; a.CFL = p.CFL; a.previous = m->previous; a.dtmax = p.DT; a.t = m->t; a.tnext = m->tnext; a.nl = m->nl;
      a.with_dt = 1; a.seq = ++m->pub_seq;
      hipLaunchKernelGGL(k_step_dt, dim3(1), dim3(64), 0, m->st, m->d_scal, a, m->d_pub);
      host_read();
      int adv = 0;
      spec_rc = rhs_terms(m, MSOM_Q, MSOM_DQ, 1, p.iRe, p.iRe4, p.Eks, p.Ekb, MSOM_QPRED, MSOM_Q, 0., &adv);
      if (!adv && !spec_rc) spec_rc = MSOM_ERR_STATE;
    };
  }
  const double d = solve_and_dt(m, MSOM_Q, p.DT);
  m->spec_hook = nullptr;
  const bool hit1 = spec && m->spec_valid && !spec_rc;
  m->dt_dev = nullptr;
  if (d < 0) return m->sticky ? m->sticky : MSOM_ERR_HIP;
  if (hit1) {
    m->dt = m->h_scal[SC_SPEC + 1];
    tnext = m->h_scal[SC_SPEC + 2];
  } else {
    m->dt = dtnext(m, d, &tnext);
    if ((r = rhs_terms(m, MSOM_Q, MSOM_DQ, 1, p.iRe, p.iRe4, p.Eks, p.Ekb, MSOM_QPRED, MSOM_Q, m->dt / 2., &advanced))) return r;
    if (!advanced && (r = advance_qg(m, MSOM_QPRED, MSOM_Q, MSOM_DQ, m->dt / 2.))) return r;
  }
  const bool tracers = m->p.nptr > 0;
  if (tracers && ((r = tracer_update(m, MSOM_PTR)) || (r = tracer_advance(m, MSOM_PTR_PRED, MSOM_PTR, m->dt / 2.)))) return r;
  spec_rc = 0;
  if (spec) {   // stage 2: dt is known; q + dt dq goes to a spare buffer that replaces q only if the solve ends after its first cycle
    if (!m->q_alt) {   // pads and ghost cells as q has them: the pass writes interior cells only
      HIPCHK(hipMalloc(&m->q_alt, m->g.ls * m->nl * sizeof(double)));
      HIPCHK(hipMemcpyAsync(m->q_alt, m->f[MSOM_Q], m->g.ls * m->nl * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    }
    m->spec_hook = [&](const std::function<void()> &host_read) {
      StepDtArgs a;
      memset(&a, 0, sizeof a);
      a.with_dt = 0; a.seq = ++m->pub_seq;
      hipLaunchKernelGGL(k_step_dt, dim3(1), dim3(64), 0, m->st, m->d_scal, a, m->d_pub);   // publish only
      host_read();
      int adv = 0;
      m->adv_out_override = m->q_alt;
      spec_rc = rhs_terms(m, MSOM_QPRED, MSOM_DQ, 1, p.iRe, p.iRe4, p.Eks, p.Ekb, MSOM_Q, MSOM_Q, m->dt, &adv);
      m->adv_out_override = nullptr;
      if (!adv && !spec_rc) spec_rc = MSOM_ERR_STATE;
    };
  }
  const double d2 = solve_and_dt(m, MSOM_QPRED, m->dt);
  m->spec_hook = nullptr;
  if (d2 < 0) return m->sticky ? m->sticky : MSOM_ERR_HIP;
  if (spec && m->spec_valid && !spec_rc) std::swap(m->f[MSOM_Q], m->q_alt);
  else {
    if ((r = rhs_terms(m, MSOM_QPRED, MSOM_DQ, 1, p.iRe, p.iRe4, p.Eks, p.Ekb, MSOM_Q, MSOM_Q, m->dt, &advanced))) return r;
    if (!advanced && (r = advance_qg(m, MSOM_Q, MSOM_Q, MSOM_DQ, m->dt))) return r;
  }
  if (tracers && ((r = tracer_update(m, MSOM_PTR_PRED)) || (r = tracer_advance(m, MSOM_PTR, MSOM_PTR, m->dt)))) return r;
  // With the speculative tendency pass the host already follows the GPU solve by solve (it waits for the published residual of every
  // solve) and everything it returns -- dt, t -- is known: with step_sync = 0 the last tendency pass is left running and the next step's
  // launches queue up behind it (every call that hands device data to the host synchronises the stream itself, msom_sync on request)
  const bool lazy = m->step_sync == 0 || (m->step_sync < 0 && (size_t)m->g.nx * m->g.ny * m->nl < ((size_t)1 << 23));
  if (!spec || !lazy) { if ((r = sync_stream(m))) return r; }
  else if (m->sticky) return m->sticky;
  m->t = tnext;
  m->iter++;
  if (dt_used) *dt_used = m->dt;
  return MSOM_OK;
}
extern "C" int msom_sync(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  return sync_stream(m);
}
extern "C" int msom_set_tnext(msom_t *m, double tnext) {
  if (!m) return MSOM_ERR_ARG;
  m->tnext = tnext;
  return MSOM_OK;
}
extern "C" double msom_time(msom_t *m) { return m ? m->t : NAN; }
extern "C" int msom_iter(msom_t *m) { return m ? m->iter : -1; }

// msqg/qg.c:101-109
extern "C" double msom_ke(msom_t *m) {
  if (!m) return NAN;
  launch_ke(m->st, m->f[MSOM_PSI], m->partial, m->d_scal + SC_KE, m->g, m->p.L0 / m->gnx);
  if (reduce_scal(m, SC_KE, 1, RED_SUM)) return NAN;
  return -m->h_scal[SC_KE];
}

// ------------------------------------------------------------------ wavelet scale filter (msqg/qg.h:509-560)

static int ensure_field(msom *m, int field) {
  if (m->f[field]) return MSOM_OK;
  const size_t bytes = m->g.ls * m->flayers[field] * sizeof(double);
  HIPCHK(hipMalloc(&m->f[field], bytes));
  HIPCHK(hipMemsetAsync(m->f[field], 0, bytes, m->st));
  return MSOM_OK;
}
// pyramid geometry + filter coefficients sig_lev (set_const, msqg/qg.h:1059-1090): init-time host pass
// all-gather of a small tile block [nlay][ny][nx] (host in / host out: [rank][nlay][ny][nx]) through the communicator
static int wv_gather(msom *m, const std::vector<double> &mine, std::vector<double> &all) {
  const size_t cnt = mine.size();
  all.resize(cnt * m->nranks);
  HIPCHK(hipMemcpyAsync(m->wv_gsend, mine.data(), cnt * sizeof(double), hipMemcpyHostToDevice, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  comm_begin(m);
  int r = comm_allgather(m->comm, m->wv_gsend, m->wv_grecv, cnt);
  comm_end(m);
  if (r) return r;
  HIPCHK(hipMemcpyAsync(all.data(), m->wv_grecv, cnt * m->nranks * sizeof(double), hipMemcpyDeviceToHost, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  return MSOM_OK;
}
// gathered tile blocks -> one global array [nlay][gy][gx]
static void wv_assemble(const msom *m, const std::vector<double> &all, int nlay, int bx, int by, std::vector<double> &out) {
  const int gx = bx * m->px, gy = by * m->py;
  out.assign((size_t)nlay * gx * gy, 0.);
  for (int r = 0; r < m->nranks; r++) {
    const int ox = (r % m->px) * bx, oy = (r / m->px) * by;
    for (int l = 0; l < nlay; l++)
      for (int j = 0; j < by; j++)
        for (int i = 0; i < bx; i++) out[((size_t)l * gy + oy + j) * gx + ox + i] = all[(size_t)r * nlay * bx * by + ((size_t)l * by + j) * bx + i];
  }
}
// pyramid geometry + filter coefficients sig_lev (set_const, msqg/qg.h:1059-1090): init-time host pass
static int wavelet_setup(msom *m) {
  if (m->wv_ready) return MSOM_OK;
  const Params &p = m->p;
  int r;
  const bool tiled = m->nranks > 1;
  if (m->wv_nlev == 0) {
    int n = 1;   // Basilisk levels depth() ... 0 of the GLOBAL grid
    while ((m->gnx >> n) >= 1 && (m->gny >> n) >= 1 && ((m->gnx >> n) << n) == m->gnx && ((m->gny >> n) << n) == m->gny) n++;
    m->wv_nlev = n;
    int kt = 0;  // coarsest level that still has a cell of every tile
    while (kt + 1 < n && (m->nx >> (kt + 1)) >= 1 && (m->ny >> (kt + 1)) >= 1) kt++;
    m->wv_kt = tiled ? kt : n - 1;
    const int nt = m->wv_kt + 1;
    m->wv_g.resize(nt); m->wv_s.assign(nt, nullptr); m->wv_r.assign(nt, nullptr); m->wv_sig.assign(nt, nullptr);
    for (int k = 0; k < nt; k++) {
      m->wv_g[k] = make_nat(m->nx >> k, m->ny >> k);
      const size_t ls = m->wv_g[k].ls;
      HIPCHK(hipMalloc(&m->wv_sig[k], ls * sizeof(double)));
      HIPCHK(hipMemsetAsync(m->wv_sig[k], 0, ls * sizeof(double), m->st));
      if (k == 0) continue;  // level 0 works in place on the field
      HIPCHK(hipMalloc(&m->wv_s[k], ls * m->nl * sizeof(double)));
      HIPCHK(hipMalloc(&m->wv_r[k], ls * m->nl * sizeof(double)));
      HIPCHK(hipMemsetAsync(m->wv_s[k], 0, ls * m->nl * sizeof(double), m->st));
      HIPCHK(hipMemsetAsync(m->wv_r[k], 0, ls * m->nl * sizeof(double), m->st));
    }
    if (tiled) {
      const size_t blk = (size_t)(m->nx >> kt) * (m->ny >> kt) * (m->nl > 2 ? m->nl : 2);
      HIPCHK(hipMalloc(&m->wv_gsend, blk * sizeof(double)));
      HIPCHK(hipMalloc(&m->wv_grecv, blk * m->nranks * sizeof(double)));
      m->wv_gx = (m->nx >> kt) * m->px; m->wv_gy = (m->ny >> kt) * m->py;
    }
  }
  const int K = m->wv_kt + 1;   // levels on the tile
  std::vector<std::vector<double>> sf(K), sl(K);
  sf[0].resize((size_t)m->nx * m->ny);
  HIPCHK(hipStreamSynchronize(m->st));
  launch_unpack(m->st, m->f[MSOM_RD], m->staging, m->g, 1);
  HIPCHK(hipMemcpyAsync(sf[0].data(), m->staging, sf[0].size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
  HIPCHK(hipStreamSynchronize(m->st));
  for (double &v : sf[0]) v = fmin(p.afilt * v, p.Lfmax);  // sig_filt = min(afilt * Rd, Lfmax) :1060
  auto restrict_host = [](const std::vector<double> &f, int nx, int ny, std::vector<double> &c) {  // restriction({sig_filt}) :1063
    const int fx = nx * 2;
    c.resize((size_t)nx * ny);
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        double sum = 0.;
        sum += f[(size_t)(2 * j) * fx + 2 * i]; sum += f[(size_t)(2 * j + 1) * fx + 2 * i];
        sum += f[(size_t)(2 * j) * fx + 2 * i + 1]; sum += f[(size_t)(2 * j + 1) * fx + 2 * i + 1];
        c[(size_t)j * nx + i] = sum / 4;
      }
  };
  auto lowpass_host = [&](const std::vector<double> &sfk, const std::vector<double> *child, int nx, int ny, int k, std::vector<double> &out) {  // :1066-1083
    const int fx = nx * 2;
    const double Delta = p.L0 / (double)(m->gnx >> k);
    out.resize((size_t)nx * ny);
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        double ref_flag = 0;
        if (child) {
          ref_flag += (*child)[(size_t)(2 * j) * fx + 2 * i]; ref_flag += (*child)[(size_t)(2 * j + 1) * fx + 2 * i];
          ref_flag += (*child)[(size_t)(2 * j) * fx + 2 * i + 1]; ref_flag += (*child)[(size_t)(2 * j + 1) * fx + 2 * i + 1];
        }
        const double sv = sfk[(size_t)j * nx + i];
        double v;
        if (ref_flag > 0) v = 1;
        else if (sv > 2 * Delta) v = 0;
        else if (sv <= 2 * Delta && sv > Delta) v = 1 - (sv - Delta) / Delta;
        else v = 1;
        out[(size_t)j * nx + i] = v;
      }
  };
  for (int k = 1; k < K; k++) restrict_host(sf[k - 1], m->nx >> k, m->ny >> k, sf[k]);
  for (int k = 0; k < K; k++) lowpass_host(sf[k], k > 0 ? &sl[k - 1] : nullptr, m->nx >> k, m->ny >> k, k, sl[k]);
  if (tiled) {  // the levels above the tiles: gathered sig_filt and low-pass flags of level kt, continued on the global top grid
    const int kt = m->wv_kt, bx = m->nx >> kt, by = m->ny >> kt;
    std::vector<double> mine((size_t)2 * bx * by), all, top;
    std::copy(sf[kt].begin(), sf[kt].end(), mine.begin());
    std::copy(sl[kt].begin(), sl[kt].end(), mine.begin() + (size_t)bx * by);
    if ((r = wv_gather(m, mine, all))) return r;
    wv_assemble(m, all, 2, bx, by, top);
    int gx = m->wv_gx, gy = m->wv_gy;
    std::vector<double> tsf(top.begin(), top.begin() + (size_t)gx * gy), tsl(top.begin() + (size_t)gx * gy, top.end());
    m->wv_top_sig.assign(m->wv_nlev - kt, std::vector<double>());
    m->wv_top_sig[0] = tsl;
    for (int k = kt + 1; k < m->wv_nlev; k++) {
      std::vector<double> csf, csl;
      gx >>= 1; gy >>= 1;
      restrict_host(tsf, gx, gy, csf);
      lowpass_host(csf, &tsl, gx, gy, k, csl);
      m->wv_top_sig[k - kt] = csl;
      tsf.swap(csf); tsl.swap(csl);
    }
    for (auto &v : m->wv_top_sig) for (double &q : v) q = 1 - q;   // high pass :1086-1090
  }
  for (int k = 0; k < K; k++) {  // high pass :1086-1090, then to the device
    for (double &v : sl[k]) v = 1 - v;
    const NatGeom &g = m->wv_g[k];
    HIPCHK(hipMemcpy2DAsync(m->wv_sig[k] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), sl[k].data(), g.nx * sizeof(double), g.nx * sizeof(double), g.ny,
                            hipMemcpyHostToDevice, m->st));
  }
  HIPCHK(hipStreamSynchronize(m->st));
  m->wv_ready = 1;
  return MSOM_OK;
}
static void wv_fill(msom *m, double *f, const NatGeom &g) {
  if (m->nranks > 1) { STICKY(m, exch_nat_g(m, f, g, m->nl, m->bc, 1)); return; }
  if (m->bc == BC_PERIODIC) launch_fill_periodic(m->st, f, g, m->nl, 1);
  else launch_fill_ghost(m->st, f, g, m->nl, m->bc, m->walls);
}
// host arrays of the gathered top levels: [nl][gy + 2][gx + 2] with a ghost ring filled like boundary() does
struct WvTop {
  int gx, gy, nl;
  std::vector<double> v;
  double &at(int l, int j, int i) { return v[((size_t)l * (gy + 2) + (j + 1)) * (gx + 2) + (i + 1)]; }
  void init(int gx_, int gy_, int nl_) { gx = gx_; gy = gy_; nl = nl_; v.assign((size_t)nl * (gx + 2) * (gy + 2), 0.); }
  void fill(int bc) {  // x sides first, then y over the x-ghost columns (corners = y rule of the x ghost)
    for (int l = 0; l < nl; l++) {
      for (int j = 0; j < gy; j++) {
        if (bc == BC_PERIODIC) { at(l, j, -1) = at(l, j, gx - 1); at(l, j, gx) = at(l, j, 0); }
        else if (bc == BC_NEUMANN) { at(l, j, -1) = at(l, j, 0); at(l, j, gx) = at(l, j, gx - 1); }
        else { at(l, j, -1) = -at(l, j, 0); at(l, j, gx) = -at(l, j, gx - 1); }
      }
      for (int i = -1; i <= gx; i++) {
        if (bc == BC_PERIODIC) { at(l, -1, i) = at(l, gy - 1, i); at(l, gy, i) = at(l, 0, i); }
        else if (bc == BC_NEUMANN) { at(l, -1, i) = at(l, 0, i); at(l, gy, i) = at(l, gy - 1, i); }
        else { at(l, -1, i) = -at(l, 0, i); at(l, gy, i) = -at(l, gy - 1, i); }
      }
    }
  }
  double bilin(int l, int i, int j) {  // fine cell (i, j) of the next finer level from this one, as k_wv_recon's bilin()
    const int I = i >> 1, J = j >> 1, cx = (i & 1) ? 1 : -1, cy = (j & 1) ? 1 : -1;
    return (9. * at(l, J, I) + 3. * (at(l, J, I + cx) + at(l, J + cy, I)) + at(l, J + cy, I + cx)) / 16.;
  }
};
// field <- inverse_wavelet(sig_lev * wavelet(field)), all layers (msqg/qg.h:524-539)
static int wavelet_apply(msom *m, double *f) {
  m->res_ready = -1;
  int r = wavelet_setup(m);
  if (r) return r;
  const int K = m->wv_kt + 1, nl = m->nl;   // levels on the tile (all of them on a single tile)
  const bool tiled = m->nranks > 1;
  wv_fill(m, f, m->g);
  for (int k = 1; k < K; k++) {
    launch_wv_restrict(m->st, k == 1 ? f : m->wv_s[k - 1], m->wv_g[k - 1], m->wv_s[k], m->wv_g[k], nl);
    wv_fill(m, m->wv_s[k], m->wv_g[k]);
  }
  if (!tiled) {
    if (K == 1) launch_wv_root(m->st, f, m->wv_sig[0], f, m->g, nl);
    else launch_wv_root(m->st, m->wv_s[K - 1], m->wv_sig[K - 1], m->wv_r[K - 1], m->wv_g[K - 1], nl);
    if (K > 1) wv_fill(m, m->wv_r[K - 1], m->wv_g[K - 1]);
  } else {
    // levels kt .. depth 0 on the gathered top grid, on the host (a few cells), every rank the same arithmetic as the kernels:
    // s_k+1 = mean of the 4 children, w = (s - bilinear(s coarse)) sig, r = bilinear(r coarse) + w, root r = s sig
    // a recorded error of THIS rank must not keep it out of the collective below (the other ranks would wait in it):
    // gather first (whatever the block holds), report afterwards
    const int sticky_before = m->sticky;
    const int kt = m->wv_kt, bx = m->nx >> kt, by = m->ny >> kt, nt = m->wv_nlev - kt;
    const NatGeom &gk = m->wv_g[kt];
    const double *src = kt == 0 ? f : m->wv_s[kt];
    std::vector<double> mine((size_t)nl * bx * by), all, top;
    for (int l = 0; l < nl; l++)
      HIPCHK(hipMemcpy2DAsync(mine.data() + (size_t)l * bx * by, bx * sizeof(double), src + nat_idx(gk, l, 0, 0), gk.pitch * sizeof(double), bx * sizeof(double), by,
                              hipMemcpyDeviceToHost, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
    if ((r = wv_gather(m, mine, all))) return r;
    if (sticky_before) return sticky_before;
    wv_assemble(m, all, nl, bx, by, top);
    std::vector<WvTop> S(nt), R(nt);
    int gx = m->wv_gx, gy = m->wv_gy;
    S[0].init(gx, gy, nl);
    for (int l = 0; l < nl; l++) for (int j = 0; j < gy; j++) for (int i = 0; i < gx; i++) S[0].at(l, j, i) = top[((size_t)l * gy + j) * gx + i];
    S[0].fill(m->bc);
    for (int q = 1; q < nt; q++) {
      S[q].init(S[q - 1].gx >> 1, S[q - 1].gy >> 1, nl);
      for (int l = 0; l < nl; l++) for (int j = 0; j < S[q].gy; j++) for (int i = 0; i < S[q].gx; i++) {
        double sum = 0.;
        sum += S[q - 1].at(l, 2 * j, 2 * i); sum += S[q - 1].at(l, 2 * j + 1, 2 * i); sum += S[q - 1].at(l, 2 * j, 2 * i + 1); sum += S[q - 1].at(l, 2 * j + 1, 2 * i + 1);
        S[q].at(l, j, i) = sum / 4;
      }
      S[q].fill(m->bc);
    }
    R[nt - 1].init(S[nt - 1].gx, S[nt - 1].gy, nl);
    for (int l = 0; l < nl; l++) for (int j = 0; j < R[nt - 1].gy; j++) for (int i = 0; i < R[nt - 1].gx; i++)
      R[nt - 1].at(l, j, i) = S[nt - 1].at(l, j, i) * m->wv_top_sig[nt - 1][(size_t)j * R[nt - 1].gx + i];
    R[nt - 1].fill(m->bc);
    for (int q = nt - 2; q >= 0; q--) {
      R[q].init(S[q].gx, S[q].gy, nl);
      for (int l = 0; l < nl; l++) for (int j = 0; j < R[q].gy; j++) for (int i = 0; i < R[q].gx; i++) {
        double d = S[q].at(l, j, i);
        d -= S[q + 1].bilin(l, i, j);
        const double w = d * m->wv_top_sig[q][(size_t)j * R[q].gx + i];
        double rr = R[q + 1].bilin(l, i, j);
        rr += w;
        R[q].at(l, j, i) = rr;
      }
      R[q].fill(m->bc);
    }
    // this tile's block of the filtered level kt, with its ring (neighbour cells or wall ghosts), back to the device
    double *dst = kt == 0 ? f : m->wv_r[kt];
    std::vector<double> blk((size_t)nl * (bx + 2) * (by + 2));
    for (int l = 0; l < nl; l++) for (int j = -1; j <= by; j++) for (int i = -1; i <= bx; i++)
      blk[((size_t)l * (by + 2) + j + 1) * (bx + 2) + i + 1] = R[0].at(l, m->iy * by + j, m->ix * bx + i);
    for (int l = 0; l < nl; l++)
      HIPCHK(hipMemcpy2DAsync(dst + nat_idx(gk, l, -1, -1), gk.pitch * sizeof(double), blk.data() + (size_t)l * (bx + 2) * (by + 2), (bx + 2) * sizeof(double),
                              (bx + 2) * sizeof(double), by + 2, hipMemcpyHostToDevice, m->st));
    HIPCHK(hipStreamSynchronize(m->st));
  }
  for (int k = K - 2; k >= 0; k--) {
    double *s = k == 0 ? f : m->wv_s[k], *out = k == 0 ? f : m->wv_r[k];
    launch_wv_recon(m->st, s, m->wv_s[k + 1], m->wv_r[k + 1], m->wv_sig[k], out, m->wv_g[k], m->wv_g[k + 1], nl);
    wv_fill(m, out, m->wv_g[k]);
  }
  HIPCHK(hipGetLastError());
  return m->sticky;
}
static int wavelet_filter(msom *m, int qof_field, double dtflt) {
  NEED_CONST(m);
  int r;
  if ((r = wavelet_setup(m)) || (r = ensure_field(m, qof_field))) return r;
  const int nl = m->nl;
  // tmp = q (interior; the saved copy is restored into q when dtflt < 0)
  HIPCHK(hipMemcpyAsync(m->f[MSOM_TMP], m->f[MSOM_Q], m->g.ls * nl * sizeof(double), hipMemcpyDeviceToDevice, m->st));
  if ((r = invertq(m, m->f[MSOM_Q]))) return r;
  if ((r = wavelet_apply(m, m->f[MSOM_PSI]))) return r;
  m->umax_ready = 0;
  comp_del2(m, MSOM_PSI, MSOM_Q, 0., 1.);
  comp_stretch(m, MSOM_PSI, MSOM_Q, 1., 1.);
  // `nbar` is a by-value argument in the reference (msqg/qg.h:510,558): the running mean never advances
  launch_wv_qof(m->st, m->f[qof_field], m->f[MSOM_Q], m->f[MSOM_TMP], m->g, nl, dtflt, 0, dtflt < 0.0);
  if (dtflt < 0.0) fill_bc(m, MSOM_Q);
  fill_bc(m, qof_field);
  return sync_stream(m);
}
extern "C" int msom_wavelet_filter(msom_t *m, double dtflt) { return wavelet_filter(m, MSOM_QOF, dtflt); }

// ------------------------------------------------------------------ energy / PV budgets (msqg/qg_energy.h)

static int ensure_de_fields(msom *m) {
  for (int k = MSOM_DE_BF; k <= MSOM_PO_MFT; k++) {
    int r = ensure_field(m, k);
    if (r) return r;
  }
  return MSOM_OK;
}
// advection_de + dissip_de + ekman_friction_de on the current psi / zeta (energy_tend :229-232, pystep_de :327-329)
static int de_terms(msom *m, double dt, double ediag) {
  const Params &p = m->p;
  const int nl = m->nl;
  const double D = p.L0 / m->gnx;
  launch_advection_de(m->st, m->f[MSOM_ZETA], m->f[MSOM_PSI], m->f[MSOM_PSIPG], m->f[MSOM_ZETAPG], m->f[MSOM_S], m->f[MSOM_DE_J1], m->f[MSOM_DE_J2],
                      m->f[MSOM_DE_J3], m->g, nl, D, p.beta, dt, ediag, m->lc);
  comp_del2(m, MSOM_ZETA, MSOM_TMP, 0., 1.);
  comp_stretch(m, MSOM_ZETA, MSOM_TMP2, 0., 1.);
  launch_dissip_de(m->st, m->f[MSOM_TMP], m->f[MSOM_TMP2], m->f[MSOM_PSI], m->f[MSOM_DE_VD], m->g, nl, p.iRe, p.iRe4, dt, ediag, D, 0);
  comp_stretch(m, MSOM_TMP, MSOM_TMP2, 0., 1.);
  launch_dissip_de(m->st, m->f[MSOM_TMP], m->f[MSOM_TMP2], m->f[MSOM_PSI], m->f[MSOM_DE_VD], m->g, nl, p.iRe, p.iRe4, dt, ediag, D, 1);
  launch_ekman_de(m->st, m->f[MSOM_ZETA], m->f[MSOM_PSI], m->f[MSOM_DE_BF], m->g, nl, p.Eks / (p.Rom * 2 * m->dhf[0]),
                  p.Ekb / (p.Rom * 2 * m->dhf[nl - 1]), dt, ediag);
  HIPCHK(hipGetLastError());
  return m->sticky;
}
extern "C" int msom_energy_tend(msom_t *m, double dt) {
  NEED_CONST(m);
  int r;
  if ((r = ensure_de_fields(m))) return r;
  comp_del2(m, MSOM_PSI, MSOM_ZETA, 0., 1.0);
  if ((r = de_terms(m, dt, (double)m->p.ediag))) return r;
  launch_running_mean(m->st, m->f[MSOM_PO_MFT], m->f[MSOM_PSI], m->g, m->nl, m->nme_ft);
  m->nme_ft += 1;
  return sync_stream(m);
}
static int filter_de(msom *m, int pm_field, double dtflt, double ediag) {
  int r;
  if ((r = ensure_de_fields(m))) return r;
  if ((r = wavelet_filter(m, MSOM_TMP2, -dtflt))) return r;  // tmp2: tmp is used inside wavelet_filter (:210-213)
  launch_filter_de(m->st, m->f[MSOM_DE_FT], m->f[MSOM_TMP2], m->f[pm_field], m->g, m->nl, dtflt, ediag);
  m->nme_ft = 0;
  return sync_stream(m);
}
extern "C" int msom_filter_de(msom_t *m, int pm_field, double dtflt) {
  NEED_CONST(m);
  if (pm_field < 0 || pm_field >= MSOM_NFIELDS || m->flayers[pm_field] != m->nl) { msom_set_error("bad field id %d", pm_field); return MSOM_ERR_ARG; }
  int r = ensure_field(m, pm_field);
  return r ? r : filter_de(m, pm_field, dtflt, (double)m->p.ediag);
}
extern "C" int msom_reset_de(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  int r;
  if ((r = ensure_de_fields(m))) return r;
  for (int k = MSOM_DE_BF; k <= MSOM_DE_FT; k++) HIPCHK(hipMemsetAsync(m->f[k], 0, m->g.ls * m->nl * sizeof(double), m->st));
  return sync_stream(m);
}
// pystep_de, msqg/qg_energy.h:296-349 (SWIG: msqg/qg_energy.i:31; caller msqg/scripts/energy_offline.py:113).
// ediag = 1 and dt = 1 are locals of the reference routine; filter_de runs with po_mft = pol (so the
// stream function is zeroed on exit) and the global dtflt.
extern "C" int pystep_de(msom_t *m, const double *po_py, int len1, int len2, int len3, double *de_bf_py, int len4, int len5, int len6,
                         double *de_vd_py, int len7, int len8, int len9, double *de_j1_py, int len10, int len11, int len12,
                         double *de_j2_py, int len13, int len14, int len15, double *de_j3_py, int len16, int len17, int len18,
                         double *de_ft_py, int len19, int len20, int len21, int onlyKE) {
  NEED_CONST(m);
  if (check_shape(m, len1, len2, len3) || check_shape(m, len4, len5, len6) || check_shape(m, len7, len8, len9) || check_shape(m, len10, len11, len12) ||
      check_shape(m, len13, len14, len15) || check_shape(m, len16, len17, len18) || check_shape(m, len19, len20, len21))
    return MSOM_ERR_ARG;
  if (!po_py) return MSOM_ERR_ARG;
  const double ediag = 1., dt = 1.;
  int r;
  if ((r = upload(m, MSOM_PSI, po_py)) || (r = msom_reset_de(m))) return r;
  m->umax_ready = 0;
  comp_del2(m, MSOM_PSI, MSOM_ZETA, 0., 1.0);
  comp_del2(m, MSOM_PSI, MSOM_Q, 0., 1.);
  comp_stretch(m, MSOM_PSI, MSOM_Q, 1., 1.);
  if (onlyKE == 1 && !m->s_zero) {  // :319-325 strl = 0 (stays so until the next set_const)
    m->s_zero = 1;
    if ((r = build_coefs(m))) return r;
  }
  if ((r = de_terms(m, dt, ediag)) || (r = filter_de(m, MSOM_PSI, m->p.dtflt, ediag))) return r;
  double *outs[6] = {de_bf_py, de_vd_py, de_j1_py, de_j2_py, de_j3_py, de_ft_py};
  for (int k = 0; k < 6; k++)
    if (outs[k] && (r = download(m, MSOM_DE_BF + k, outs[k]))) return r;
  return MSOM_OK;
}

extern "C" int msom_dbg_wavelet_levels(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  int r = wavelet_setup(m);
  return r ? r : m->wv_nlev;
}
extern "C" int msom_dbg_siglev(msom_t *m, int level, double *out) {
  if (!m || !out) return MSOM_ERR_ARG;
  int r = wavelet_setup(m);
  if (r) return r;
  if (level < 0 || level >= (int)m->wv_g.size()) { msom_set_error("bad level %d (tiles: only the levels that live on the tile)", level); return MSOM_ERR_ARG; }
  const NatGeom &g = m->wv_g[level];
  HIPCHK(hipMemcpy2DAsync(out, g.nx * sizeof(double), m->wv_sig[level] + nat_idx(g, 0, 0, 0), g.pitch * sizeof(double), g.nx * sizeof(double), g.ny,
                          hipMemcpyDefault, m->st));
  return sync_stream(m);
}
extern "C" int msom_dbg_wavelet_apply(msom_t *m, int field) {
  NEED_CONST(m);
  if (field < 0 || field >= MSOM_NFIELDS || !m->f[field] || m->flayers[field] != m->nl) { msom_set_error("bad field id %d", field); return MSOM_ERR_ARG; }
  int r = wavelet_apply(m, m->f[field]);
  return r ? r : sync_stream(m);
}

// ------------------------------------------------------------------ .bas IO and the qg.c driver loop

// tiles -> the global array [layers][gny][gnx] on the host of EVERY rank (all-gather of the tile interiors); collective
static int gather_global(msom *m, int field, std::vector<double> &g) {
  const int nlf = m->flayers[field];
  const size_t cnt = (size_t)nlf * m->nx * m->ny;
  g.resize((size_t)nlf * m->gnx * m->gny);
  if (m->nranks == 1) return download(m, field, g.data());
  double *recv = nullptr;
  HIPCHK(hipMalloc(&recv, cnt * m->nranks * sizeof(double)));
  launch_unpack(m->st, m->f[field], m->staging, m->g, nlf);
  comm_begin(m);
  int r = comm_allgather(m->comm, m->staging, recv, cnt);
  comm_end(m);
  std::vector<double> h(cnt * m->nranks);
  if (!r) {
    hipError_t e = hipMemcpyAsync(h.data(), recv, h.size() * sizeof(double), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { msom_set_error("HIP error %s in gather_global", hipGetErrorString(e)); r = MSOM_ERR_HIP; }
  }
  (void)hipFree(recv);
  if (r) return r;
  for (int q = 0; q < m->nranks; q++) {
    const int tx = q % m->px, ty = q / m->px;
    for (int l = 0; l < nlf; l++)
      for (int j = 0; j < m->ny; j++)
        memcpy(&g[((size_t)l * m->gny + ty * m->ny + j) * m->gnx + (size_t)tx * m->nx], &h[q * cnt + ((size_t)l * m->ny + j) * m->nx], m->nx * sizeof(double));
  }
  return MSOM_OK;
}
// the global array (held by every rank) -> this rank's tile; collective (the field's halo exchange)
static int scatter_global(msom *m, int field, const std::vector<double> &g) {
  const int nlf = m->flayers[field];
  if (m->nranks == 1) return msom_set_field(m, field, g.data());
  std::vector<double> t((size_t)nlf * m->nx * m->ny);
  for (int l = 0; l < nlf; l++)
    for (int j = 0; j < m->ny; j++)
      memcpy(&t[((size_t)l * m->ny + j) * m->nx], &g[((size_t)l * m->gny + m->iy * m->ny + j) * m->gnx + (size_t)m->ix * m->nx], m->nx * sizeof(double));
  return msom_set_field(m, field, t.data());
}

// .bas / NetCDF IO.  Tiled runs: the calls are collective; rank 0 writes the global field, every rank reads the (shared)
// file and keeps its tile -- what output_matrix_mpi / input_matrixl do in the reference's MPI build (msqg/auxiliar_input.h)
extern "C" int msom_write_bas(msom_t *m, int field, const char *path) {
  if (check_field(m, field) || !path) return MSOM_ERR_ARG;
  if (m->gnx != m->gny) { msom_set_error(".bas output needs a square grid"); return MSOM_ERR_STATE; }
  std::vector<double> h;
  int r = gather_global(m, field, h);
  if (r) return r;
  if (m->rank != 0) return MSOM_OK;
  return msom_bas_write(path, h.data(), m->flayers[field], m->gnx, m->p.L0) ? MSOM_ERR_IO : MSOM_OK;
}
extern "C" int msom_read_bas(msom_t *m, int field, const char *path) {
  if (check_field(m, field) || !path) return MSOM_ERR_ARG;
  if (m->gnx != m->gny) { msom_set_error(".bas input needs a square grid"); return MSOM_ERR_STATE; }
  std::vector<double> h((size_t)m->flayers[field] * m->gnx * m->gny);
  if (msom_bas_read(path, h.data(), m->flayers[field], m->gnx, m->p.L0)) return MSOM_ERR_IO;
  return scatter_global(m, field, h);
}

extern "C" int msom_write_nc(msom_t *m, const char *path) {
  if (!m || !path) return MSOM_ERR_ARG;
  static const char *names[2] = {"psi", "q"};
  std::vector<double> hp, hq;
  int r;
  if ((r = gather_global(m, MSOM_PSI, hp)) || (r = gather_global(m, MSOM_Q, hq))) return r;
  if (m->rank != 0) return MSOM_OK;
  struct stat sb;
  if (stat(path, &sb) != 0 && msom_nc_create(path, m->nl, m->gny, m->gnx, m->p.L0, 2, names)) return MSOM_ERR_IO;
  const double *f[2] = {hp.data(), hq.data()};
  return msom_nc_append(path, m->nl, m->gny, m->gnx, 2, names, m->t, f) < 0 ? MSOM_ERR_IO : MSOM_OK;
}
extern "C" int msom_read_nc(msom_t *m, int field, const char *path, const char *varname, int record) {
  if (check_field(m, field) || !path || !varname) return MSOM_ERR_ARG;
  std::vector<double> h((size_t)m->flayers[field] * m->gnx * m->gny);
  int r = msom_nc_read(path, varname, record, m->flayers[field], m->gny, m->gnx, h.data(), nullptr);
  if (r) return r == -2 ? MSOM_ERR_IO : MSOM_ERR_ARG;
  return scatter_global(m, field, h);
}

static bool file_exists(const char *path) {
  struct stat sb;
  return stat(path, &sb) == 0;
}

// optional input files, msqg/qg.h:940-984 and msqg/qg.c:55-59 (p0.bas)
extern "C" int msom_read_inputs(msom_t *m, const char *dir) {
  if (!m) return MSOM_ERR_ARG;
  char name[512];
  const char *d = dir ? dir : ".";
  const int nl = m->nl, N = m->gnx;
  int r;
  const bool say = !m->quiet && m->rank == 0;  // tiled runs: rank 0 prints, every rank reads the shared files
  if (say) fprintf(stdout, "Read input files:\n");
  snprintf(name, sizeof name, "%s/dh_%dl.bin", d, nl);
  if (FILE *fp = fopen(name, "r")) {
    std::vector<float> dh(nl);
    size_t got = fread(dh.data(), sizeof(float), nl, fp);
    fclose(fp);
    if (got != (size_t)nl) { msom_set_error("short read on %s", name); return MSOM_ERR_IO; }
    for (int l = 0; l < nl; l++) m->dhf[l] = dh[l];
    if (say) fprintf(stdout, "%s .. ok\n", name);
  }
  struct { const char *fmt; int field; } files[] = {
      {"%s/psipg_%dl_N%d.bas", MSOM_PSIPG}, {"%s/frpg_%dl_N%d.bas", MSOM_FR}, {"%s/rdpg_%dl_N%d.bas", MSOM_RD},
      {"%s/qforc_%dl_N%d.bas", MSOM_QFORC}};
  for (auto &f : files) {
    snprintf(name, sizeof name, f.fmt, d, nl, N);
    if (file_exists(name)) {
      if ((r = msom_read_bas(m, f.field, name))) return r;
      if (say) fprintf(stdout, "%s .. ok\n", name);
    }
  }
  snprintf(name, sizeof name, "%s/topo.bas", d);
  if (file_exists(name)) {
    if ((r = msom_read_bas(m, MSOM_TOPO, name))) return r;
    if (say) fprintf(stdout, "%s .. ok\n", name);
  }
  if (m->p.nptr > 0) {  // msqg/qg.c:75-90
    snprintf(name, sizeof name, "%s/ptr0.bas", d);
    if (file_exists(name)) {
      if ((r = msom_read_bas(m, MSOM_PTR, name))) return r;
      if (say) fprintf(stdout, "%s .. ok\n", name);
    }
    snprintf(name, sizeof name, "%s/ptr_relax.bas", d);
    if (file_exists(name) && (r = msom_read_bas(m, MSOM_PTR_RELAX, name))) return r;
  }
  snprintf(name, sizeof name, "%s/p0.bas", d);
  if (file_exists(name)) {
    if ((r = msom_read_bas(m, MSOM_PSI, name))) return r;
    if (say) fprintf(stdout, "%s .. ok\n", name);
  }
  m->const_set = 0;
  return MSOM_OK;
}

// backup_config, msqg/qg.h:782-835: params.in copy and the constant fields, written into the
// output directory at t = 0 (event write_const, msqg/qg.c:95-97)
static int backup_config(msom *m, const char *dpath) {
  const bool root = m->rank == 0;  // tiled runs: the gathers are collective, rank 0 writes
  if (root) fprintf(stdout, "Backup config\n");
  char name[700];
  const int nl = m->nl, N = m->gnx;
  const size_t n2 = (size_t)m->gnx * m->gny;
  snprintf(name, sizeof name, "%sparams.in", dpath);
  if (root) {
    if (FILE *fp = fopen(name, "w")) {
      fwrite(m->params_text.data(), 1, m->params_text.size(), fp);
      fclose(fp);
    } else {
      msom_set_error("cannot write %s", name);
      return MSOM_ERR_IO;
    }
  }
  // sig_filt = min(afilt * Rd, Lfmax) (msqg/qg.h:1060) and Rd itself (:794-810)
  std::vector<double> h, rd;
  int r;
  if ((r = gather_global(m, MSOM_RD, rd))) return r;
  h.assign(n2, 0.);
  for (size_t k = 0; k < n2; k++) h[k] = fmin(m->p.afilt * rd[k], m->p.Lfmax);
  snprintf(name, sizeof name, "%ssig_filt.bas", dpath);
  if (root && msom_bas_write(name, h.data(), 1, N, m->p.L0)) return MSOM_ERR_IO;
  snprintf(name, sizeof name, "%srdpg_%dl_N%d.bas", dpath, nl, N);
  if (root && msom_bas_write(name, rd.data(), 1, N, m->p.L0)) return MSOM_ERR_IO;
  snprintf(name, sizeof name, "%spsipg_%dl_N%d.bas", dpath, nl, N);
  if ((r = msom_write_bas(m, MSOM_PSIPG, name))) return r;
  // Frl has nl layers in the reference (nl - 1 used, the last one stays 0)
  std::vector<double> fr;
  if (nl > 1 && (r = gather_global(m, MSOM_FR, fr))) return r;
  h.assign(n2 * nl, 0.);
  if (nl > 1) memcpy(h.data(), fr.data(), n2 * (nl - 1) * sizeof(double));
  snprintf(name, sizeof name, "%sfrpg_%dl_N%d.bas", dpath, nl, N);
  if (root && msom_bas_write(name, h.data(), nl, N, m->p.L0)) return MSOM_ERR_IO;
  snprintf(name, sizeof name, "%sqforc_%dl_N%d.bas", dpath, nl, N);
  if ((r = msom_write_bas(m, MSOM_QFORC, name))) return r;
  if (root) {
    std::vector<float> dh(nl);
    for (int l = 0; l < nl; l++) dh[l] = (float)m->dhf[l];
    snprintf(name, sizeof name, "%sdh_%dl.bin", dpath, nl);
    if (FILE *fp = fopen(name, "w")) {
      fwrite(dh.data(), sizeof(float), nl, fp);
      fclose(fp);
    }
  }
  return MSOM_OK;
}

// main loop of msqg/qg.c:34-173: events at the top of every iteration (writestdout i++,
// output t += dtout), then one predictor-corrector step.
extern "C" int msom_run(msom_t *m, const char *workdir, long nsteps_max) {
  NEED_CONST(m);
  const Params &p = m->p;
  // tiled runs (one process / thread per tile): every call below that moves field data is collective; rank 0 alone
  // creates the directory, prints and writes (the reference's pid() == 0 branches, msqg/qg.h:766-780)
  const bool root = m->rank == 0;
  char dpath[600] = "", name[700];
  const char *wd = workdir ? workdir : ".";
  // create_outdir, msqg/qg.h:766-776
  for (int i = 1; root && i < 10000; i++) {
    snprintf(dpath, sizeof dpath, "%s/outdir_%04d/", wd, i);
    if (mkdir(dpath, 0777) == 0) {
      fprintf(stdout, "Writing output in %s\n", dpath);
      break;
    }
  }
  double tout = 0.;  // next output event: t = 0; t <= tend + 1e-10; t += dtout
  double tflt = p.dtflt;  // next filter event: t = dtflt; t <= tend + 1e-10; t += dtflt (msqg/qg.h:655-658)
  const bool filtering = p.dtflt > 0;
  long steps = 0;
  int r;
  if (m->iter == 0 && (r = backup_config(m, dpath))) return r;  // event write_const (t = 0)
  for (;;) {
    if (filtering && tflt <= p.tend + 1e-10 && m->t >= tflt - 1e-12 * fmax(1., fabs(tflt))) {
      // two events named `filter`: the later-defined one (qg_energy.h:269-272) runs first [BASILISK RULE 8]
      if (p.ediag > -1 && (r = filter_de(m, MSOM_PO_MFT, p.dtflt, (double)p.ediag))) return r;
      fprintf(stdout, "Filter solution\n");
      if ((r = msom_wavelet_filter(m, p.dtflt))) return r;
      tflt += p.dtflt;
    }
    if (p.ediag > -1 && (r = msom_energy_tend(m, m->dt))) return r;  // event comp_diag (i++), qg_energy.h:289-291
    // writestdout, msqg/qg.c:101-109
    {
      const double ke = msom_ke(m);
      if (root) fprintf(stdout, "i = %i, dt = %g, t = %g, ke_1 = %g\n", m->iter, m->dt, m->t, ke);
    }
    // output, msqg/qg.c:112-122
    bool out_pending = tout <= p.tend + 1e-10;
    if (out_pending && m->t >= tout - 1e-12 * fmax(1., fabs(tout))) {
      if (root) fprintf(stdout, "write file\n");
      if ((r = invertq(m, m->f[MSOM_Q]))) return r;
      snprintf(name, sizeof name, "%spo%09d.bas", dpath, m->iter);
      if ((r = msom_write_bas(m, MSOM_PSI, name))) return r;
      snprintf(name, sizeof name, "%sqo%09d.bas", dpath, m->iter);
      if ((r = msom_write_bas(m, MSOM_Q, name))) return r;
      if (filtering) {  // msqg/qg.c:124-129: psi of the filter mean, tmpl = invertq(qofl)
        if ((r = ensure_field(m, MSOM_QOF))) return r;
        double *keep = m->f[MSOM_PSI];
        const msom_mgstats mgkeep = m->mg;
        m->f[MSOM_PSI] = m->f[MSOM_TMP];  // invertq(tmpl, qofl): tmp is the unknown (and the first guess)
        r = invertq(m, m->f[MSOM_QOF]);
        double *solved = m->f[MSOM_PSI];
        if (solved != m->f[MSOM_TMP]) { m->psi_alt = m->f[MSOM_TMP]; m->f[MSOM_TMP] = solved; }  // the fused correction swaps buffers
        m->f[MSOM_PSI] = keep;
        m->mg = mgkeep;
        m->umax_ready = 0;
        if (r) return r;
        snprintf(name, sizeof name, "%spf%09d.bas", dpath, m->iter);
        if ((r = msom_write_bas(m, MSOM_TMP, name))) return r;
      }
      if (p.ediag > -1) {  // msqg/qg.c:139-160: budgets scaled by 1/dtout, written, reset
        const char *tags[6] = {"de_bf", "de_vd", "de_j1", "de_j2", "de_j3", "de_ft"};
        std::vector<double> h;
        const double idtout = 1 / p.dtout;
        for (int k = 0; k < 6; k++) {
          if ((r = gather_global(m, MSOM_DE_BF + k, h))) return r;   // collective on tiles; rank 0 writes
          if (!root) continue;
          for (double &v : h) v *= idtout;
          snprintf(name, sizeof name, "%s%s%09d.bas", dpath, tags[k], m->iter);
          if (msom_bas_write(name, h.data(), m->nl, m->gnx, p.L0)) return MSOM_ERR_IO;
        }
        if ((r = msom_reset_de(m))) return r;
      }
      if (p.nptr > 0) {  // msqg/qg.c:168-171
        snprintf(name, sizeof name, "%sptr%09d.bas", dpath, m->iter);
        if ((r = msom_write_bas(m, MSOM_PTR, name))) return r;
      }
      tout += p.dtout;
      out_pending = tout <= p.tend + 1e-10;
    }
    if (!out_pending) break;  // no scheduled event left: run() ends
    if (nsteps_max >= 0 && steps >= nsteps_max) break;
    m->tnext = filtering && tflt <= p.tend + 1e-10 ? fmin(tout, tflt) : tout;
    if ((r = msom_step(m, nullptr))) return r;
    steps++;
  }
  fflush(stdout);
  return MSOM_OK;
}

// ------------------------------------------------------------------ tiling (single tile for now)

extern "C" int msom_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  return MSOM_OK;
}

extern "C" int msom_comm_unique_id(void *id128) {
  if (!id128) return MSOM_ERR_ARG;
  return comm_unique_id(id128);
}
// One tile of a px x py decomposition.  id128 = ncclUniqueId from msom_comm_unique_id (RCCL
// transport, one process per GPU), or "MSOMLOCL" + 8-byte key for the in-process test
// transport (one host thread per tile).
extern "C" msom_t *msom_create_tiled(const char *params_text, int px, int py, int rank, const void *id128) {
  if (!params_text || px < 1 || py < 1 || rank < 0 || rank >= px * py || (px * py > 1 && !id128)) {
    msom_set_error("msom_create_tiled: bad arguments");
    return nullptr;
  }
  Params p;
  msom_params_defaults(&p);
  msom_params_parse_text(&p, params_text);
  msom *m = create_common(p, px, py, rank, id128);
  if (m) m->params_text = params_text;
  return m;
}
extern "C" int msom_tile_info(msom_t *m, int *px, int *py, int *ix, int *iy, int *nx_local, int *ny_local) {
  if (!m) return MSOM_ERR_ARG;
  if (px) *px = m->px;
  if (py) *py = m->py;
  if (ix) *ix = m->ix;
  if (iy) *iy = m->iy;
  if (nx_local) *nx_local = m->nx;
  if (ny_local) *ny_local = m->ny;
  return MSOM_OK;
}

// ------------------------------------------------------------------ debug / test hooks

extern "C" int msom_dbg_nlevels(msom_t *m) { return m ? m->nlev : MSOM_ERR_ARG; }
extern "C" int msom_dbg_level_dims(msom_t *m, int lev, int *nx, int *ny) {
  if (!m || lev < 0 || lev >= m->nlev) return MSOM_ERR_ARG;
  *nx = m->sg[lev].nx; *ny = m->sg[lev].ny;
  return MSOM_OK;
}
static int split_upload(msom *m, double *sp, const SplitGeom &sg, const double *a, int nl, int bc) {
  HIPCHK(hipMemcpyAsync(m->staging, a, (size_t)nl * sg.nx * sg.ny * sizeof(double), hipMemcpyDefault, m->st));
  HIPCHK(hipMemsetAsync(sp, 0, sg.ls * nl * sizeof(double), m->st));
  launch_split_pack(m->st, m->staging, sp, sg, nl, bc, m->walls);
  return MSOM_OK;
}
static int split_download(msom *m, const double *sp, const SplitGeom &sg, double *a, int nl) {
  launch_split_unpack(m->st, sp, sg, m->staging, nl);
  HIPCHK(hipMemcpyAsync(a, m->staging, (size_t)nl * sg.nx * sg.ny * sizeof(double), hipMemcpyDefault, m->st));
  return sync_stream(m);
}
// nsweeps red-black relaxations on level `lev`: da in/out, res in (both [layer][y][x] of that level)
extern "C" int msom_dbg_relax(msom_t *m, int lev, double *da, const double *res, int nsweeps) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  if (lev < 0 || lev >= m->nlev || !da || !res) return MSOM_ERR_ARG;
  int r;
  if ((r = split_upload(m, m->da[lev], m->sg[lev], da, m->nl, BC_DIRICHLET0))) return r;
  if ((r = split_upload(m, m->res[lev], m->sg[lev], res, m->nl, BC_NEUMANN))) return r;
  STICKY(m, exch_split(m, m->da[lev], m->sg[lev], m->nl, 0));
  const int prof = m->profile;
  m->profile = 0;
  {
    Lev L = tile_lev(m, lev);
    relax_sweeps(m, L, nullptr, nsweeps, 0);
  }
  m->profile = prof;
  return split_download(m, m->da[lev], m->sg[lev], da, m->nl);
}
extern "C" int msom_dbg_residual(msom_t *m, const double *a, const double *b, double *res, double *maxres) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  int r;
  if ((r = upload(m, MSOM_TMP, a))) return r;
  if ((r = upload(m, MSOM_QPRED, b))) return r;
  HIPCHK(hipMemsetAsync(m->d_scal, 0, 8 * sizeof(double), m->st));
  residual(m, m->f[MSOM_TMP], m->f[MSOM_QPRED], SC_RES0, 0);
  HIPCHK(hipMemcpyAsync(m->h_scal + SC_RES0, m->d_scal + SC_RES0, sizeof(double), hipMemcpyDeviceToHost, m->st));
  if ((r = split_download(m, m->res[0], m->sg[0], res, m->nl))) return r;
  if (maxres) *maxres = m->h_scal[SC_RES0];
  return MSOM_OK;
}
extern "C" int msom_dbg_restrict(msom_t *m, int lev_fine, const double *fine, double *coarse) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  if (lev_fine < 0 || lev_fine + 1 >= m->nlev) return MSOM_ERR_ARG;
  int r;
  if ((r = split_upload(m, m->res[lev_fine], m->sg[lev_fine], fine, m->nl, BC_NEUMANN))) return r;
  launch_restrict(m->st, m->res[lev_fine], m->sg[lev_fine], m->res[lev_fine + 1], m->sg[lev_fine + 1], m->nl);
  return split_download(m, m->res[lev_fine + 1], m->sg[lev_fine + 1], coarse, m->nl);
}
extern "C" int msom_dbg_prolong(msom_t *m, int lev_coarse, const double *coarse, double *fine) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  if (lev_coarse < 1 || lev_coarse >= m->nlev) return MSOM_ERR_ARG;
  int r;
  if ((r = split_upload(m, m->da[lev_coarse], m->sg[lev_coarse], coarse, m->nl, BC_DIRICHLET0))) return r;
  STICKY(m, exch_split(m, m->da[lev_coarse], m->sg[lev_coarse], m->nl, 1));
  launch_prolong(m->st, m->da[lev_coarse], m->sg[lev_coarse], m->da[lev_coarse - 1], m->sg[lev_coarse - 1], m->nl, m->walls);
  return split_download(m, m->da[lev_coarse - 1], m->sg[lev_coarse - 1], fine, m->nl);
}
// single operators on the internal fields: "del2", "stretch", "advection", "dissip", "forcing"
extern "C" int msom_dbg_op(msom_t *m, const char *op, int f_in, int f_out, double add, double fac) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  if (check_field(m, f_in) || check_field(m, f_out) || !op) return MSOM_ERR_ARG;
  const Params &p = m->p;
  const double D = p.L0 / m->gnx;
  if (!strcmp(op, "del2")) comp_del2(m, f_in, f_out, add, fac);
  else if (!strcmp(op, "stretch")) comp_stretch(m, f_in, f_out, add, fac);
  else if (!strcmp(op, "advection")) {  // f_in = zeta field, psi = PSI, dq = f_out (accumulates)
    launch_advection(m->st, m->f[f_in], m->f[MSOM_PSI], m->f[MSOM_PSIPG], m->f[MSOM_ZETAPG], m->f[MSOM_S], m->f[MSOM_Q], m->f[f_out], m->g,
                     m->nl, m->have_pg, m->have_zpg, m->stochastic, D, p.beta, p.itr_stoch, m->lc);
  } else {
    msom_set_error("unknown op %s", op);
    return MSOM_ERR_ARG;
  }
  return sync_stream(m);
}

// ------------------------------------------------------------------ measurement

extern "C" int msom_profile_reset(msom_t *m) {
  if (!m) return MSOM_ERR_ARG;
  for (auto *ps : {&m->prof_sweep, &m->prof_resid, &m->prof_block, &m->prof_march[2], &m->prof_march[3], &m->prof_march[4], &m->prof_rhs, &m->prof_redprol, &m->prof_rescorr, &m->prof_respre, &m->prof_march_pl, &m->prof_march_corr, &m->prof_resmax}) { ps->used = 0; ps->total_ms = 0; ps->launches = 0; }
  return MSOM_OK;
}
extern "C" int msom_profile_read(msom_t *m, const char *kernel, double *avg_ms, long *launches) {
  if (!m || !kernel) return MSOM_ERR_ARG;
  ProfSlot *ps = !strcmp(kernel, "sweep") ? &m->prof_sweep : !strcmp(kernel, "residual") ? &m->prof_resid : !strcmp(kernel, "block2") ? &m->prof_block :
                 !strcmp(kernel, "march2") ? &m->prof_march[2] : !strcmp(kernel, "march3") ? &m->prof_march[3] : !strcmp(kernel, "march4") ? &m->prof_march[4] :
                 !strcmp(kernel, "march_pl") ? &m->prof_march_pl : !strcmp(kernel, "march_corr") ? &m->prof_march_corr : !strcmp(kernel, "resid_max") ? &m->prof_resmax : !strcmp(kernel, "rhs") ? &m->prof_rhs : !strcmp(kernel, "red_prolong") ? &m->prof_redprol : !strcmp(kernel, "resid_correct") ? &m->prof_rescorr :
                 !strcmp(kernel, "resid_restrict") ? &m->prof_respre : nullptr;
  if (!ps) { msom_set_error("unknown kernel %s", kernel); return MSOM_ERR_ARG; }
  prof_collect(m, *ps);
  if (avg_ms) *avg_ms = ps->launches ? ps->total_ms / ps->launches : 0.;
  if (launches) *launches = ps->launches;
  return MSOM_OK;
}
// back-to-back launches of one kernel on the finest level, HIP-event timed
extern "C" int msom_bench_kernel(msom_t *m, const char *kernel, int reps, double *avg_ms) {
  if (m) m->res_ready = -1;
  NEED_CONST(m);
  if (!kernel || reps < 1 || !avg_ms) return MSOM_ERR_ARG;
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  const double D = m->p.L0 / m->gnx;
  auto one = [&](void) {
    if (!strcmp(kernel, "sweep")) {
      for (int c = 0; c < 2; c++) launch_relax_color(m->st, m->da[0], m->res[0], m->S[0], m->sg[0], m->nl, m->rc[0], m->uniformS, c, m->walls, 1);
    } else if (!strcmp(kernel, "residual")) {
      launch_residual(m->st, m->f[MSOM_PSI], m->f[MSOM_Q], m->f[MSOM_S], m->g, m->res[0], m->sg[0], m->nl, m->rc[0], m->uniformS, m->d_scal + SC_RES1, m->partial, 0);
    } else if (!strcmp(kernel, "advection")) {
      launch_advection(m->st, m->f[MSOM_ZETA], m->f[MSOM_PSI], m->f[MSOM_PSIPG], m->f[MSOM_ZETAPG], m->f[MSOM_S], m->f[MSOM_Q], m->f[MSOM_TMP], m->g,
                       m->nl, m->have_pg, m->have_zpg, m->stochastic, D, m->p.beta, m->p.itr_stoch, m->lc);
    } else if (!strncmp(kernel, "march", 5) && kernel[5] >= '2' && kernel[5] <= '4') {
      const bool rev = kernel[6] == 'r';  // "march3r": da_alt -> da (the direction of the second pass of a level)
      const bool pl = kernel[6] == 'p';   // "march4p": the pass with the prolongation (coarse = level 1)
      // timing experiment (option dbg_interleave; results meaningless): the same buffers addressed as [row][layer][x]
      // instead of [layer][row][x] -- all layers of a row within one 2-MB fragment
      SplitGeom g0 = m->sg[0], g1 = m->sg[m->nlev > 1 ? 1 : 0];
      if (g_dbg_interleave) { g0.ls = g0.rp; g0.rp *= m->nl; g1.ls = g1.rp; g1.rp *= m->nl; }
      if (pl) launch_relax_march(m->st, nullptr, m->da_alt[0], m->res[0], g0, m->nl, m->rc[0], 0, kernel[5] - '0', m->walls, g_march_rows, nullptr, m->da[1], &g1, nullptr, 1);
      else launch_relax_march(m->st, rev ? m->da_alt[0] : m->da[0], rev ? m->da[0] : m->da_alt[0], m->res[0], g0, m->nl, m->rc[0], 1, kernel[5] - '0', m->walls,
                         g_march_rows);
    } else if (!strcmp(kernel, "block2")) {
      launch_relax_block2(m->st, m->da[0], nullptr, m->sg[0], m->res[0], m->da_alt[0], m->sg[0], m->nl, m->rc[0], m->walls, 1);
    } else if (!strcmp(kernel, "block2p")) {
      launch_relax_block2(m->st, m->da[0], m->da[1], m->sg[1], m->res[0], m->da_alt[0], m->sg[0], m->nl, m->rc[0], m->walls, 1);
    } else if (!strcmp(kernel, "rhs")) {
      rhs_terms(m, MSOM_Q, MSOM_DQ, 1, m->p.iRe, m->p.iRe4, m->p.Eks, m->p.Ekb);
    } else if (!strcmp(kernel, "resid_correct")) {
      residual2(m, 1, m->f[MSOM_Q], SC_RES1, 0);
    } else if (!strcmp(kernel, "resid_restrict")) {
      residual2(m, 2 | 4, m->f[MSOM_Q], SC_RES1, 1);
    } else if (!strcmp(kernel, "red_prolong")) {
      launch_relax_red_prolong(m->st, m->da[0], m->da[1], m->sg[1], m->res[0], m->S[0], m->sg[0], m->nl, m->rc[0], m->uniformS, m->walls);
    } else if (!strcmp(kernel, "rhs_adv")) {
      int adv = 0;
      rhs_terms(m, MSOM_Q, MSOM_DQ, 1, m->p.iRe, m->p.iRe4, m->p.Eks, m->p.Ekb, MSOM_QPRED, MSOM_Q, 1e-9, &adv);
    } else if (!strcmp(kernel, "advance")) {
      launch_advance(m->st, m->f[MSOM_QPRED], m->f[MSOM_Q], m->f[MSOM_DQ], nullptr, m->g, m->nl, 1e-9, 0.);
    }
  };
  for (int k = 0; k < 3; k++) one();
  HIPCHK(hipEventRecord(a, m->st));
  for (int k = 0; k < reps; k++) one();
  HIPCHK(hipEventRecord(b, m->st));
  HIPCHK(hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  *avg_ms = ms / reps;
  hipEventDestroy(a);
  hipEventDestroy(b);
  return MSOM_OK;
}
