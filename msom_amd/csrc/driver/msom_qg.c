/* msom_qg.c -- host driver in C, the drop-in counterpart of the reference's main()
 * (msqg/qg.c:34-48): read params.in (or argv[1]), create the output directory, run to tend
 * printing the per-step line and writing po/qo .bas files.  Everything numerical happens in
 * libmsomhip through the C ABI (include/msom.h). */
#include <stdio.h>
#include <stdlib.h>

#include "../../../include/msom.h"

int main(int argc, char *argv[]) {
  const char *params = argc >= 2 ? argv[1] : "params.in";
  long nsteps = argc >= 3 ? atol(argv[2]) : -1; /* extension: stop after nsteps */
  msom_t *m = msom_create(params);
  if (!m) {
    fprintf(stdout, "%s\n", msom_last_error());
    return 1;
  }
  fprintf(stdout, "Config: N = %d, nl = %d, L0 = %g\n", (int)msom_get_param(m, "N"), (int)msom_get_param(m, "nl"),
          msom_get_param(m, "L0"));
  int r = msom_read_inputs(m, ".");
  if (!r) r = msom_remove_mean(m, MSOM_PSI); /* msqg/qg.c:65-70 */
  if (!r) r = msom_set_const(m);
  if (!r) r = msom_run(m, ".", nsteps);
  if (r) fprintf(stdout, "error %d: %s\n", r, msom_last_error());
  msom_destroy(m);
  return r ? 1 : 0;
}
