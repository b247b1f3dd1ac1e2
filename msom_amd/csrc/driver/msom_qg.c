/* msom_qg.c -- host driver in C, the drop-in counterpart of the reference's main()
 * (msqg/qg.c:34-48): read params.in (or argv[1]), create the output directory, run to tend
 * printing the per-step line and writing po/qo .bas files.  Everything numerical happens in
 * libmsomhip through the C ABI (include/msom.h).
 *
 * Multi-GPU (the reference's `mpirun -np 16 ./qg.e`): start one process per GPU with
 *   MSOM_NRANKS (or WORLD_SIZE), MSOM_RANK (or RANK), MSOM_LOCAL_RANK (or LOCAL_RANK, default = rank),
 *   optional MSOM_PX / MSOM_PY (default 1x1, 2x1, 2x2, 2x4) and MSOM_ID_FILE (default ./.msom_comm_id.<MSOM_JOB or MASTER_PORT>)
 * in the environment, e.g. through `python -m torch.distributed.run --no-python` or a shell loop.  Rank 0
 * publishes the 128-byte communicator id in MSOM_ID_FILE (shared directory), the others wait for it. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../../include/msom.h"

static int env_int(const char *a, const char *b, int dflt) {
  const char *v = getenv(a);
  if (!v && b) v = getenv(b);
  return v ? atoi(v) : dflt;
}

static char *slurp(const char *path) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  fseek(fp, 0, SEEK_END);
  long n = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char *b = (char *)malloc(n + 1);
  if (fread(b, 1, n, fp) != (size_t)n) { free(b); fclose(fp); return NULL; }
  b[n] = 0;
  fclose(fp);
  return b;
}

int main(int argc, char *argv[]) {
  const char *params = argc >= 2 ? argv[1] : "params.in";
  long nsteps = argc >= 3 ? atol(argv[2]) : -1; /* extension: stop after nsteps */
  const int nranks = env_int("MSOM_NRANKS", "WORLD_SIZE", 1), rank = env_int("MSOM_RANK", "RANK", 0);
  msom_t *m = NULL;
  if (nranks > 1) {
    static const int grid[9][2] = {{0, 0}, {1, 1}, {2, 1}, {0, 0}, {2, 2}, {0, 0}, {0, 0}, {0, 0}, {2, 4}};
    int px = env_int("MSOM_PX", NULL, nranks <= 8 ? grid[nranks][0] : 0), py = env_int("MSOM_PY", NULL, nranks <= 8 ? grid[nranks][1] : 0);
    if (px * py != nranks) { fprintf(stdout, "MSOM_PX x MSOM_PY = %d x %d does not match %d ranks\n", px, py, nranks); return 1; }
    char idbuf[300];
    const char *job = getenv("MSOM_JOB") ? getenv("MSOM_JOB") : (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
    snprintf(idbuf, sizeof idbuf, ".msom_comm_id.%s", job); /* one name per job: a stale file of a crashed run is never read */
    const char *idfile = getenv("MSOM_ID_FILE") ? getenv("MSOM_ID_FILE") : idbuf;
    unsigned char id[128];
    if (msom_set_device(env_int("MSOM_LOCAL_RANK", "LOCAL_RANK", rank))) { fprintf(stdout, "%s\n", msom_last_error()); return 1; }
    if (rank == 0) {
      char tmp[600];
      if (msom_comm_unique_id(id)) { fprintf(stdout, "%s\n", msom_last_error()); return 1; }
      snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
      FILE *fp = fopen(tmp, "wb");
      if (!fp || fwrite(id, 1, 128, fp) != 128) { fprintf(stdout, "cannot write %s\n", tmp); return 1; }
      fclose(fp);
      rename(tmp, idfile); /* atomic publish */
    } else {
      int ok = 0;
      for (int k = 0; k < 1200 && !ok; k++) { /* up to 2 minutes */
        FILE *fp = fopen(idfile, "rb");
        if (fp) { ok = fread(id, 1, 128, fp) == 128; fclose(fp); }
        if (!ok) usleep(100000);
      }
      if (!ok) { fprintf(stdout, "rank %d: no communicator id in %s\n", rank, idfile); return 1; }
    }
    char *text = slurp(params);
    if (!text) { fprintf(stdout, "file %s not found\n", params); return 1; } /* reference message, msqg/qg.h:736 */
    m = msom_create_tiled(text, px, py, rank, id);
    free(text);
    if (rank == 0) remove(idfile); /* every rank has joined: the id is spent */
  } else
    m = msom_create(params);
  if (!m) {
    fprintf(stdout, "%s\n", msom_last_error());
    return 1;
  }
  if (rank == 0)
    fprintf(stdout, "Config: N = %d, nl = %d, L0 = %g\n", (int)msom_get_param(m, "N"), (int)msom_get_param(m, "nl"), msom_get_param(m, "L0"));
  int r = msom_read_inputs(m, ".");
  if (!r) r = msom_remove_mean(m, MSOM_PSI); /* msqg/qg.c:65-70 */
  if (!r) r = msom_set_const(m);
  if (!r) r = msom_run(m, ".", nsteps);
  if (r) fprintf(stdout, "error %d: %s\n", r, msom_last_error());
  msom_destroy(m);
  return r ? 1 : 0;
}
