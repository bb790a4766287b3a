/* oracle/rir.c -- TEST INFRASTRUCTURE ONLY.  See rir.h. */
#include "rir.h"
#include "jmath.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const uint32_t *w; size_t n, pos; int bad; } rd;
static uint32_t r32(rd *r) { if (r->pos >= r->n) { r->bad = 1; return 0; } return r->w[r->pos++]; }
static double r64(rd *r) {
  uint64_t lo = r32(r), hi = r32(r); uint64_t u = lo | (hi << 32); double d; memcpy(&d, &u, 8); return d;
}
#define FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); rir_free(p); free(words); return NULL; } while (0)

/* nodes are topologically ordered (operands have smaller ids), so one reverse sweep
 * propagates reachability from the marked roots. */
static void sweep(const rir_prog *p, uint8_t *seen) {
  for (uint32_t n = p->n_nodes; n-- > 0;) {
    if (!seen[n]) continue;
    const rir_node *nd = &p->nodes[n];
    switch (nd->op) {
    case RIR_CONST: case RIR_INPUT: break;
    case RIR_LOOKUP:
      seen[nd->a] = 1;
      for (uint32_t k = 0; k < nd->count; k++) seen[nd->table[k]] = 1;
      break;
    case RIR_SEQ: seen[nd->a] = 1; seen[nd->b] = 1; break;
    default:
      seen[nd->a] = 1;
      if (nd->op >= RIR_ADD && nd->op <= RIR_COMPARE) seen[nd->b] = 1;
    }
  }
}

rir_prog *rir_parse(const void *buf, size_t len, char *err, size_t errlen) {
  rir_prog *p = calloc(1, sizeof(*p));
  uint32_t *words = malloc(len + 4);
  memcpy(words, buf, len);
  rd r = {words, len / 4, 0, 0};
  if (len % 4) FAIL("RIR length not a multiple of 4");
  if (r32(&r) != 0x31524952u) FAIL("bad RIR magic");
  if (r32(&r) != 1) FAIL("unsupported RIR version");
  p->n_params = r32(&r); p->n_targets = r32(&r); p->n_nodes = r32(&r); (void)r32(&r);
  if (r.bad || p->n_targets == 0 || p->n_targets > 250) FAIL("bad RIR header");
  p->targets = calloc(p->n_targets, sizeof(rir_target));
  uint32_t in = p->n_params;
  for (uint32_t t = 0; t < p->n_targets; t++) {
    rir_target *tg = &p->targets[t];
    tg->n_cols = r32(&r); (void)r32(&r);
    tg->input_start = in; in += tg->n_cols;
    tg->outputs = malloc(sizeof(uint32_t) * (p->n_params + 1));
    for (uint32_t o = 0; o <= p->n_params; o++) {
      tg->outputs[o] = r32(&r);
      if (tg->outputs[o] >= p->n_nodes) FAIL("output node id out of range");
    }
  }
  p->n_inputs = in;
  p->nodes = calloc(p->n_nodes ? p->n_nodes : 1, sizeof(rir_node));
  p->dep_target = calloc(p->n_nodes ? p->n_nodes : 1, 1);
  for (uint32_t i = 0; i < p->n_nodes; i++) {
    rir_node *nd = &p->nodes[i];
    nd->op = r32(&r);
    uint8_t dep = 0;
#define DEP(x) do { if ((x) >= i) FAIL("node %u references later node", i); uint8_t d_ = p->dep_target[x]; if (d_) { if (dep && dep != d_) dep = 255; else dep = d_; } } while (0)
    switch (nd->op) {
    case RIR_CONST: nd->cval = r64(&r); break;
    case RIR_INPUT:
      nd->input = r32(&r);
      if (nd->input >= p->n_inputs) FAIL("input index out of range");
      if (nd->input >= p->n_params) {
        uint32_t t = 0; while (t + 1 < p->n_targets && nd->input >= p->targets[t].input_start + p->targets[t].n_cols) t++;
        dep = (uint8_t)(t + 1);
      }
      break;
    case RIR_ADD: case RIR_SUB: case RIR_MUL: case RIR_DIV: case RIR_POW: case RIR_COMPARE:
      nd->a = r32(&r); nd->b = r32(&r); DEP(nd->a); DEP(nd->b); break;
    case RIR_EXP: case RIR_LOG: case RIR_ABS: case RIR_NOOP: case RIR_SIN: case RIR_COS: case RIR_TAN:
    case RIR_ASIN: case RIR_ACOS: case RIR_ATAN:
      nd->a = r32(&r); DEP(nd->a); break;
    case RIR_LOOKUP:
      nd->a = r32(&r); nd->low = (int32_t)r32(&r); nd->count = r32(&r);
      DEP(nd->a);
      if (nd->count == 0 || nd->count > r.n) FAIL("bad lookup table size");
      nd->table = malloc(sizeof(uint32_t) * nd->count);
      for (uint32_t k = 0; k < nd->count; k++) { nd->table[k] = r32(&r); DEP(nd->table[k]); }
      break;
    case RIR_SEQ: nd->a = r32(&r); nd->b = r32(&r); DEP(nd->a); DEP(nd->b); break;
    default: FAIL("unknown opcode %u at node %u", nd->op, i);
    }
    if (r.bad) FAIL("truncated RIR");
    if (dep == 255) FAIL("node %u mixes columns of two targets", i);
    p->dep_target[i] = dep;
  }
  if (r.pos != r.n) FAIL("trailing bytes in RIR");
  /* evaluation plans */
  uint8_t *seen = malloc(p->n_nodes ? p->n_nodes : 1);
  for (uint32_t t = 0; t < p->n_targets; t++) {
    rir_target *tg = &p->targets[t];
    memset(seen, 0, p->n_nodes);
    for (uint32_t o = 0; o <= p->n_params; o++) seen[tg->outputs[o]] = 1;
    sweep(p, seen);
    tg->once_nodes = malloc(sizeof(uint32_t) * (p->n_nodes + 1));
    tg->row_nodes = malloc(sizeof(uint32_t) * (p->n_nodes + 1));
    for (uint32_t i = 0; i < p->n_nodes; i++) {
      if (!seen[i]) continue;
      if (p->dep_target[i] == 0) tg->once_nodes[tg->n_once++] = i;
      else if (p->dep_target[i] == t + 1) tg->row_nodes[tg->n_row++] = i;
      else { free(seen); FAIL("target %u reads a column of another target", t); }
    }
  }
  free(seen);
  free(words);
  return p;
}

void rir_free(rir_prog *p) {
  if (!p) return;
  if (p->nodes) for (uint32_t i = 0; i < p->n_nodes; i++) free(p->nodes[i].table);
  if (p->targets) for (uint32_t t = 0; t < p->n_targets; t++) { free(p->targets[t].outputs); free(p->targets[t].once_nodes); free(p->targets[t].row_nodes); }
  free(p->nodes); free(p->targets); free(p->dep_target); free(p);
}

rir_density *rir_density_new(const rir_prog *p, const double *const *cols, const int64_t *nrows, int math_mode) {
  rir_density *d = calloc(1, sizeof(*d));
  d->prog = p; d->cols = cols; d->nrows = nrows; d->math_mode = math_mode;
  d->val = calloc(p->n_nodes ? p->n_nodes : 1, sizeof(double));
  return d;
}
void rir_density_free(rir_density *d) { if (d) { free(d->val); free(d); } }

/* java.lang.Math.pow: C99 pow plus the two cases where Java differs (Math.pow Javadoc):
 * second argument NaN -> NaN (unless handled by y == 0 first), |x| == 1 with infinite y -> NaN. */
static double java_pow(double x, double y) {
  if (y == 0.0) return 1.0;
  if (isnan(y)) return NAN;
  if (isinf(y) && fabs(x) == 1.0) return NAN;
  return pow(x, y);
}
/* DCMPL; I2D   (ir/MethodGenerator.scala:62-65) */
static double java_dcmpl(double l, double r) { return l > r ? 1.0 : (l == r ? 0.0 : -1.0); }
/* D2I (JVM spec): NaN -> 0, saturating */
static int32_t java_d2i(double x) {
  if (isnan(x)) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (-2147483647 - 1);
  return (int32_t)x;
}

static inline void eval_node(rir_density *d, uint32_t id, const double *in) {
  const rir_node *n = &d->prog->nodes[id];
  double *v = d->val;
  switch (n->op) {
  case RIR_CONST: v[id] = n->cval; break;
  case RIR_INPUT: v[id] = in[n->input]; break;
  case RIR_ADD: v[id] = v[n->a] + v[n->b]; break;
  case RIR_SUB: v[id] = v[n->a] - v[n->b]; break;
  case RIR_MUL: v[id] = v[n->a] * v[n->b]; break;
  case RIR_DIV: v[id] = v[n->a] / v[n->b]; break;
  case RIR_POW: v[id] = java_pow(v[n->a], v[n->b]); break;
  case RIR_COMPARE: v[id] = java_dcmpl(v[n->a], v[n->b]); break;
  case RIR_EXP: v[id] = jm_exp(d->math_mode, v[n->a]); break;
  case RIR_LOG: v[id] = jm_log(d->math_mode, v[n->a]); break;
  case RIR_ABS: v[id] = fabs(v[n->a]); break;
  case RIR_NOOP: v[id] = v[n->a]; break;
  case RIR_SIN: v[id] = sin(v[n->a]); break;
  case RIR_COS: v[id] = cos(v[n->a]); break;
  case RIR_TAN: v[id] = tan(v[n->a]); break;
  case RIR_ASIN: v[id] = asin(v[n->a]); break;
  case RIR_ACOS: v[id] = acos(v[n->a]); break;
  case RIR_ATAN: v[id] = atan(v[n->a]); break;
  case RIR_LOOKUP: {
    int64_t k = (int64_t)java_d2i(v[n->a]) - (int64_t)n->low;
    if (k < 0 || k >= (int64_t)n->count) { d->lookup_error = 1; v[id] = NAN; }
    else v[id] = v[n->table[k]];
    break;
  }
  case RIR_SEQ: v[id] = v[n->b]; break;
  }
}

enum { ACC_F64 = 0, ACC_ABS = 1, ACC_LD = 2 };
static int update_impl(rir_density *d, const double *q, double *out, int acc_mode) {
  const rir_prog *p = d->prog;
  const uint32_t nout = p->n_params + 1;
  double *in = calloc(p->n_inputs ? p->n_inputs : 1, sizeof(double));
  long double *ld = acc_mode == ACC_LD ? calloc(nout, sizeof(long double)) : NULL;
  memcpy(in, q, sizeof(double) * p->n_params);
  for (uint32_t o = 0; o < nout; o++) out[o] = 0.0;
  d->lookup_error = 0;
  size_t colbase = 0;
  for (uint32_t t = 0; t < p->n_targets; t++) {
    const rir_target *tg = &p->targets[t];
    for (uint32_t i = 0; i < tg->n_once; i++) eval_node(d, tg->once_nodes[i], in);
    int64_t rows = tg->n_cols ? d->nrows[t] : 1;
    for (int64_t k = 0; k < rows; k++) {
      for (uint32_t j = 0; j < tg->n_cols; j++) in[tg->input_start + j] = d->cols[colbase + j][k];
      for (uint32_t i = 0; i < tg->n_row; i++) eval_node(d, tg->row_nodes[i], in);
      for (uint32_t o = 0; o < nout; o++) {
        double c = d->val[tg->outputs[o]];
        if (acc_mode == ACC_F64) out[o] += c;
        else if (acc_mode == ACC_ABS) out[o] += fabs(c);
        else ld[o] += (long double)c;
      }
    }
    colbase += tg->n_cols;
  }
  if (ld) { for (uint32_t o = 0; o < nout; o++) out[o] = (double)ld[o]; free(ld); }
  free(in);
  d->n_updates++;
  return d->lookup_error;
}
/* one pass, two results: the f64 sequential sums (DataFunction order) AND the sums of |term| that the parity bound is
 * stated in -- halves the oracle's cost for the BASELINE-size parity tests */
int rir_density_update_both(rir_density *d, const double *q, double *out, double *abs_out) {
  const rir_prog *p = d->prog;
  const uint32_t nout = p->n_params + 1;
  double *in = calloc(p->n_inputs ? p->n_inputs : 1, sizeof(double));
  memcpy(in, q, sizeof(double) * p->n_params);
  for (uint32_t o = 0; o < nout; o++) { out[o] = 0.0; abs_out[o] = 0.0; }
  d->lookup_error = 0;
  size_t colbase = 0;
  for (uint32_t t = 0; t < p->n_targets; t++) {
    const rir_target *tg = &p->targets[t];
    for (uint32_t i = 0; i < tg->n_once; i++) eval_node(d, tg->once_nodes[i], in);
    int64_t rows = tg->n_cols ? d->nrows[t] : 1;
    for (int64_t k = 0; k < rows; k++) {
      for (uint32_t j = 0; j < tg->n_cols; j++) in[tg->input_start + j] = d->cols[colbase + j][k];
      for (uint32_t i = 0; i < tg->n_row; i++) eval_node(d, tg->row_nodes[i], in);
      for (uint32_t o = 0; o < nout; o++) {
        const double c = d->val[tg->outputs[o]];
        out[o] += c;
        abs_out[o] += fabs(c);
      }
    }
    colbase += tg->n_cols;
  }
  free(in);
  d->n_updates++;
  return d->lookup_error;
}
int rir_density_update(void *d, const double *q, double *out) { return update_impl((rir_density *)d, q, out, ACC_F64); }
int rir_density_abs_sums(rir_density *d, const double *q, double *o) { return update_impl(d, q, o, ACC_ABS); }
int rir_density_update_ld(rir_density *d, const double *q, double *o) { return update_impl(d, q, o, ACC_LD); }

/* Extended-precision evaluation of the same program (x87 long double: 64-bit significand, node by node AND in the row sums):
 * not the reference's arithmetic -- the JVM computes in binary64 -- but the yardstick that tells the reference's own rounding
 * (naive 1 - p, masked gradient columns ...) from the device's when the two differ by more than the stated tolerance.
 * Semantics (compare, D2I, lookup range) as in eval_node. */
static inline void eval_node_x(const rir_prog *p, long double *v, uint32_t id, const long double *in, int *lookup_error) {
  const rir_node *n = &p->nodes[id];
  switch (n->op) {
  case RIR_CONST: v[id] = (long double)n->cval; break;
  case RIR_INPUT: v[id] = in[n->input]; break;
  case RIR_ADD: v[id] = v[n->a] + v[n->b]; break;
  case RIR_SUB: v[id] = v[n->a] - v[n->b]; break;
  case RIR_MUL: v[id] = v[n->a] * v[n->b]; break;
  case RIR_DIV: v[id] = v[n->a] / v[n->b]; break;
  case RIR_POW: {
    const long double x = v[n->a], y = v[n->b];
    v[id] = y == 0.0L ? 1.0L : (isnan(y) || (isinf(y) && fabsl(x) == 1.0L)) ? (long double)NAN : powl(x, y);
    break;
  }
  case RIR_COMPARE: v[id] = v[n->a] > v[n->b] ? 1.0L : (v[n->a] == v[n->b] ? 0.0L : -1.0L); break;
  case RIR_EXP: v[id] = expl(v[n->a]); break;
  case RIR_LOG: v[id] = logl(v[n->a]); break;
  case RIR_ABS: v[id] = fabsl(v[n->a]); break;
  case RIR_NOOP: v[id] = v[n->a]; break;
  case RIR_SIN: v[id] = sinl(v[n->a]); break;
  case RIR_COS: v[id] = cosl(v[n->a]); break;
  case RIR_TAN: v[id] = tanl(v[n->a]); break;
  case RIR_ASIN: v[id] = asinl(v[n->a]); break;
  case RIR_ACOS: v[id] = acosl(v[n->a]); break;
  case RIR_ATAN: v[id] = atanl(v[n->a]); break;
  case RIR_LOOKUP: {
    int64_t k = (int64_t)java_d2i((double)v[n->a]) - (int64_t)n->low;
    if (k < 0 || k >= (int64_t)n->count) { *lookup_error = 1; v[id] = (long double)NAN; }
    else v[id] = v[n->table[k]];
    break;
  }
  case RIR_SEQ: v[id] = v[n->b]; break;
  }
}
int rir_density_update_x(rir_density *d, const double *q, double *out) {
  const rir_prog *p = d->prog;
  const uint32_t nout = p->n_params + 1;
  long double *in = calloc(p->n_inputs ? p->n_inputs : 1, sizeof(long double));
  long double *v = calloc(p->n_nodes ? p->n_nodes : 1, sizeof(long double));
  long double *acc = calloc(nout, sizeof(long double));
  for (uint32_t i = 0; i < p->n_params; i++) in[i] = (long double)q[i];
  int lookup_error = 0;
  size_t colbase = 0;
  for (uint32_t t = 0; t < p->n_targets; t++) {
    const rir_target *tg = &p->targets[t];
    for (uint32_t i = 0; i < tg->n_once; i++) eval_node_x(p, v, tg->once_nodes[i], in, &lookup_error);
    int64_t rows = tg->n_cols ? d->nrows[t] : 1;
    for (int64_t k = 0; k < rows; k++) {
      for (uint32_t j = 0; j < tg->n_cols; j++) in[tg->input_start + j] = (long double)d->cols[colbase + j][k];
      for (uint32_t i = 0; i < tg->n_row; i++) eval_node_x(p, v, tg->row_nodes[i], in, &lookup_error);
      for (uint32_t o = 0; o < nout; o++) acc[o] += v[tg->outputs[o]];
    }
    colbase += tg->n_cols;
  }
  for (uint32_t o = 0; o < nout; o++) out[o] = (double)acc[o];
  free(in); free(v); free(acc);
  return lookup_error;
}

/* Generator.prepare (core/Generator.scala:76-84): the compiled requirements evaluated at one draw.
 * A requirements program has one data-free target per requirement, outputs[0] = the requirement. */
int rir_requirements_eval(rir_density *d, const double *q, double *out) {
  const rir_prog *p = d->prog;
  double *in = calloc(p->n_inputs ? p->n_inputs : 1, sizeof(double));
  memcpy(in, q, sizeof(double) * p->n_params);
  d->lookup_error = 0;
  for (uint32_t t = 0; t < p->n_targets; t++) {
    const rir_target *tg = &p->targets[t];
    for (uint32_t i = 0; i < tg->n_once; i++) eval_node(d, tg->once_nodes[i], in);
    out[t] = d->val[tg->outputs[0]];
  }
  free(in);
  return d->lookup_error;
}
