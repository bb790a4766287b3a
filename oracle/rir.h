/* oracle/rir.h -- TEST INFRASTRUCTURE ONLY.
 *
 * "RIR" is this repo's flat wire encoding of the reference's compiled-function input:
 *   inputs: Seq[Param], exprs: Seq[(String, Expr)]   (rainier-compute/.../ir/CompiledFunction.scala:42-45)
 *   + DataFunction(cf, numParamInputs, numOutputs, data) (rainier-compute/.../ir/DataFunction.scala:13-16)
 * with the node set of rainier-compute/.../ir/IR.scala:3-41 and ir/Ops.scala:3-37.
 * The byte layout is documented in include/rainier_hip_rir.h; this file is an independent
 * reader + interpreter that follows the reference's evaluation semantics:
 *   row loop / accumulation order : ir/DataFunction.scala:32-84
 *   op semantics                  : ir/MethodGenerator.scala:56-94,134-167, ir/ExprMethodGenerator.scala:48-70
 */
#ifndef ORACLE_RIR_H
#define ORACLE_RIR_H
#include <stddef.h>
#include <stdint.h>

enum {
  RIR_CONST = 0, RIR_INPUT = 1,
  RIR_ADD = 2, RIR_SUB = 3, RIR_MUL = 4, RIR_DIV = 5, RIR_POW = 6, RIR_COMPARE = 7,
  RIR_EXP = 8, RIR_LOG = 9, RIR_ABS = 10, RIR_NOOP = 11, RIR_SIN = 12, RIR_COS = 13, RIR_TAN = 14,
  RIR_ASIN = 15, RIR_ACOS = 16, RIR_ATAN = 17,
  RIR_LOOKUP = 18, RIR_SEQ = 19
};

typedef struct {
  uint32_t op;
  uint32_t a, b;       /* operands (node ids); LOOKUP: a = index node; SEQ: a=first b=second */
  double cval;         /* CONST */
  uint32_t input;      /* INPUT: index into the DataFunction input layout */
  int32_t low;         /* LOOKUP */
  uint32_t count;      /* LOOKUP */
  uint32_t *table;     /* LOOKUP */
} rir_node;

typedef struct {
  uint32_t n_cols;
  uint32_t input_start; /* first input slot of this target's columns */
  uint32_t *outputs;    /* [n_params+1] node ids */
  /* evaluation plan */
  uint32_t n_once, n_row;
  uint32_t *once_nodes; /* nodes reachable from outputs that do not depend on a column */
  uint32_t *row_nodes;  /* nodes reachable from outputs that do */
} rir_target;

typedef struct {
  uint32_t n_params, n_targets, n_nodes, n_inputs;
  rir_node *nodes;
  rir_target *targets;
  uint8_t *dep_target; /* per node: 0 = params only, t+1 = reads a column of target t, 255 = mixed(error) */
} rir_prog;

rir_prog *rir_parse(const void *buf, size_t len, char *err, size_t errlen);
void rir_free(rir_prog *p);

typedef struct {
  const rir_prog *prog;
  const double *const *cols; /* flattened: target0 cols..., target1 cols... */
  const int64_t *nrows;      /* per target (ignored when n_cols == 0) */
  double *val;               /* [n_nodes] scratch */
  int math_mode;             /* JM_LIBM / JM_DET for EXP and LOG nodes */
  int lookup_error;          /* set when a LOOKUP index was out of range (reference: NPE) */
  int64_t n_updates;
} rir_density;

rir_density *rir_density_new(const rir_prog *p, const double *const *cols, const int64_t *nrows, int math_mode);
void rir_density_free(rir_density *d);
/* DataFunction.apply + Model.density(): out[0] = density, out[1+i] = gradient(i). returns 0 / nonzero on lookup error */
int rir_density_update(void *d, const double *q, double *out);
/* sum over rows of |per-row contribution| per output (for condition-aware tolerances) */
int rir_density_abs_sums(rir_density *d, const double *q, double *abs_out);
/* both of the above in one pass over the rows */
int rir_density_update_both(rir_density *d, const double *q, double *out, double *abs_out);
/* long-double accumulation variant (which side is closer to the true sum) */
int rir_density_update_ld(rir_density *d, const double *q, double *out);
/* the same program evaluated node by node and summed in extended precision (long double), rounded to double at the end: the
 * yardstick for "whose rounding is it" when device and oracle differ by more than the stated tolerance -- NOT the reference's
 * arithmetic */
int rir_density_update_x(rir_density *d, const double *q, double *out);
/* requirements program (header kind 1): out[t] = value of target t's outputs[0] at q */
int rir_requirements_eval(rir_density *d, const double *q, double *out);
#endif
