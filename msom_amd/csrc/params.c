/* params.c -- params.in reader of libmsomhip (host C).
 *
 * Behaviour follows read_params of the reference (msqg/qg.h:668-761): every blank is
 * deleted from the line, the line is split at '=', the left side is matched against the
 * known keys, anything else (comments, "#!sh", blank lines, unknown keys) is silently
 * ignored; arrays are written [a,b,c]; numbers are read with atoi/atof.  The file stays
 * valid Python (msqg/scripts/read_data.py:14 exec()s it).  Implementation is table driven.
 */
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/msom.h"
#include "msom_params.h"

static __thread char g_err[512] = "";

void msom_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char *msom_last_error(void) { return g_err; }

enum { T_INT, T_DBL, T_ARR };
typedef struct { const char *key; int type; size_t off; } keydef;
#define K(name, type, member) { name, type, offsetof(struct Params, member) }
static const keydef KEYS[] = {
  K("N", T_INT, N), K("nl", T_INT, nl), K("ediag", T_INT, ediag), K("varRo", T_INT, varRo),
  K("nptr", T_INT, nptr), K("flsrv", T_INT, flsrv), K("L0", T_DBL, L0), K("Rom", T_DBL, Rom),
  K("Ekb", T_DBL, Ekb), K("Eks", T_DBL, Eks), K("tau0", T_DBL, tau0), K("Re", T_DBL, Re),
  K("Re4", T_DBL, Re4), K("sbc", T_DBL, sbc), K("beta", T_DBL, beta), K("afilt", T_DBL, afilt),
  K("Lfmax", T_DBL, Lfmax), K("DT", T_DBL, DT), K("tend", T_DBL, tend), K("dtout", T_DBL, dtout),
  K("dtflt", T_DBL, dtflt), K("CFL", T_DBL, CFL), K("Fr", T_ARR, Frm), K("dh", T_ARR, dhu),
  K("upg", T_ARR, upg), K("vpg", T_ARR, vpg), K("ptr_r", T_ARR, ptr_r), K("Pe", T_ARR, Pe), K("tr_stoch", T_DBL, tr_stoch),
  K("amp_stoch", T_DBL, amp_stoch),
  /* extension keys, unknown to (hence ignored by) the reference parser */
  K("Ny", T_INT, Ny), K("TOLERANCE", T_DBL, tolerance), K("NITERMAX", T_INT, nitermax),
  K("NITERMIN", T_INT, nitermin), K("MGLEVELS", T_INT, mglevels),
};

/* defaults: msqg/qg.h:63-106; N, L0, DT, CFL are Basilisk globals (64, 1, 1e10, 0.5) */
void msom_params_defaults(struct Params *p) {
  memset(p, 0, sizeof *p);
  p->N = 64; p->nl = 1; p->ediag = -1;
  p->L0 = 1.; p->beta = 0.5; p->afilt = 10.; p->Lfmax = 1e10;
  p->DT = 1e10; p->tend = 1; p->dtout = 1; p->dtflt = -1; p->CFL = 0.5;
  p->amp_stoch = 1;
  p->tolerance = 1e-3; p->nitermax = 100; p->nitermin = 1;
}

static void parse_array(const char *v, double *out) {
  int n = 0;
  const char *s = v;
  while (*s && n < MSOM_MAXARR) {
    while (*s == '[' || *s == ',' || *s == ']') s++;
    if (!*s) break;
    out[n++] = atof(s);
    while (*s && *s != ',' && *s != ']') s++;
  }
}

static void parse_line_tab(void *p, const keydef *keys, size_t nkeys, const char *line) {
  char buf[300];
  size_t n = 0;
  for (const char *s = line; *s && *s != '\n' && *s != '\r' && n < sizeof buf - 1; s++)
    if (*s != ' ') buf[n++] = *s;
  buf[n] = '\0';
  char *eq = strchr(buf, '=');
  if (!eq || eq == buf) return;
  *eq = '\0';
  char *val = eq + 1, *eq2 = strchr(val, '=');
  if (eq2) *eq2 = '\0';
  if (!*val) return;
  for (size_t k = 0; k < nkeys; k++) {
    if (strcmp(buf, keys[k].key)) continue;
    char *dst = (char *)p + keys[k].off;
    if (keys[k].type == T_INT) *(int *)dst = atoi(val);
    else if (keys[k].type == T_DBL) *(double *)dst = atof(val);
    else parse_array(val, (double *)dst);
    return;
  }
}
static void parse_line(struct Params *p, const char *line) { parse_line_tab(p, KEYS, sizeof KEYS / sizeof KEYS[0], line); }

int msom_params_parse_text(struct Params *p, const char *text) {
  const char *s = text;
  while (*s) {
    const char *e = strchr(s, '\n');
    size_t len = e ? (size_t)(e - s) : strlen(s);
    char line[300];
    if (len > sizeof line - 1) len = sizeof line - 1;
    memcpy(line, s, len);
    line[len] = '\0';
    parse_line(p, line);
    if (!e) break;
    s = e + 1;
  }
  return 0;
}

int msom_params_parse_file(struct Params *p, const char *path) {
  FILE *fp = fopen(path, "rt");
  if (!fp) {
    msom_set_error("file %s not found", path);   /* reference: message + exit(0), qg.h:735-738 */
    return -2;
  }
  char line[300];
  while (fgets(line, sizeof line, fp)) parse_line(p, line);
  fclose(fp);
  return 0;
}

/* derived values, msqg/qg.h:739-758 */
void msom_params_derive(struct Params *p) {
  p->iRe = p->Re == 0 ? 0. : 1 / p->Re;
  p->iRe4 = p->Re4 == 0 ? 0. : -1 / p->Re4;
  double D = p->L0 / p->N, D2 = D * D;
  if (p->Re != 0) p->DT = 0.5 * fmin(p->DT, D2 * p->Re / 4.);
  if (p->Re4 != 0) p->DT = 0.5 * fmin(p->DT, D2 * D2 * p->Re4 / 32.);
  if (p->tr_stoch != 0) p->itr_stoch = 1 / p->tr_stoch;
  for (int nt = 0; nt < p->nptr && nt < MSOM_MAXARR; nt++) { /* msqg/qg.h:751-754 */
    p->ptr_ir[nt] = p->ptr_r[nt] == 0 ? 0. : 1 / p->ptr_r[nt];
    p->iPe[nt] = p->Pe[nt] == 0 ? 0. : 1 / p->Pe[nt];
  }
}

/* ---- vertex-grid variant: parameter list of qg-node/qg.c:72-107, parser qg-node/extra.h:83-116 (same
 * line rules as above: blanks removed, split at '=', atoi/atof, arrays [a,b,c]) */
#define KN(name, type, member) { name, type, offsetof(struct NodeParams, member) }
static const keydef NODE_KEYS[] = {
  KN("N", T_INT, N), KN("nl", T_INT, nl), KN("flag_ms", T_INT, flag_ms), KN("sqg", T_INT, sqg), KN("L0", T_DBL, L0), KN("f0", T_DBL, f0),
  KN("beta", T_DBL, beta), KN("nu", T_DBL, nu), KN("nu4", T_DBL, nu4), KN("hEkb", T_DBL, hEkb), KN("gp_low", T_DBL, gp_low),
  KN("scale_topo", T_DBL, scale_topo), KN("tau0", T_DBL, tau0), KN("tau1", T_DBL, tau1), KN("tf1", T_DBL, tf1), KN("tf2", T_DBL, tf2),
  KN("dy_ws", T_DBL, dy_ws), KN("forc_mode", T_DBL, forc_mode), KN("noise_init", T_DBL, noise_init), KN("Lfmax", T_DBL, Lfmax),
  KN("Lfmin", T_DBL, Lfmin), KN("fac_filt_Rd", T_DBL, fac_filt_Rd), KN("dtflt", T_DBL, dtflt), KN("dh", T_ARR, dh), KN("N2", T_ARR, N2),
  KN("bc_fac", T_DBL, bc_fac), KN("DT", T_DBL, DT), KN("tend", T_DBL, tend), KN("dtout", T_DBL, dtout), KN("CFL", T_DBL, CFL),
  KN("TOLERANCE", T_DBL, TOLERANCE), KN("dtdiag", T_DBL, dtdiag), KN("amp_stoch", T_DBL, amp_stoch), KN("L_filt", T_DBL, L_filt),
};
/* defaults: qg-node/qg.h:104-127,164; qg.c:61-66; Basilisk globals N = 64, L0 = 1, DT = 1e10, CFL = 0.5, TOLERANCE = 1e-3 */
void msom_node_params_defaults(struct NodeParams *p) {
  memset(p, 0, sizeof *p);
  p->N = 64; p->nl = 1; p->L0 = 1.; p->f0 = 1.; p->scale_topo = 1.; p->tf1 = 1.; p->tf2 = 1.; p->dy_ws = 1.; p->forc_mode = 2.0;
  p->dh[0] = 1.; p->N2[0] = 1.; p->DT = 1e10; p->tend = 100; p->dtout = 1; p->CFL = 0.5; p->TOLERANCE = 1e-3; p->dtdiag = -1;
  p->Lfmax = 1e30; p->Lfmin = 1e30; p->dtflt = -1; /* HUGE of Basilisk, qg-node/qg.h:119-120 */
}
int msom_node_params_parse_text(struct NodeParams *p, const char *text) {
  const char *s = text;
  while (*s) {
    const char *e = strchr(s, '\n');
    size_t len = e ? (size_t)(e - s) : strlen(s);
    char line[300];
    if (len > sizeof line - 1) len = sizeof line - 1;
    memcpy(line, s, len);
    line[len] = '\0';
    parse_line_tab(p, NODE_KEYS, sizeof NODE_KEYS / sizeof NODE_KEYS[0], line);
    if (!e) break;
    s = e + 1;
  }
  return 0;
}
int msom_node_params_parse_file(struct NodeParams *p, const char *path) {
  FILE *fp = fopen(path, "rt");
  if (!fp) { msom_set_error("file %s not found", path); return -2; } /* reference: message + exit(0), extra.h:111-114 */
  char line[300];
  while (fgets(line, sizeof line, fp)) parse_line_tab(p, NODE_KEYS, sizeof NODE_KEYS / sizeof NODE_KEYS[0], line);
  fclose(fp);
  return 0;
}
