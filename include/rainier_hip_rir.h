/* rainier_hip_rir.h -- "RIR": the wire format of the engine's input language.
 *
 * RIR is a flat little-endian encoding of exactly what the reference hands its bytecode back end:
 *
 *   ir.CompiledFunction(inputs: Seq[Param], exprs: Seq[(String, Expr)], ...)
 *       rainier-compute/src/main/scala/com/stripe/rainier/ir/CompiledFunction.scala:42-45
 *   ir.DataFunction(cf, numParamInputs, numOutputs, data)
 *       rainier-compute/src/main/scala/com/stripe/rainier/ir/DataFunction.scala:3-30
 *
 * as produced by Compiler.compileTargets (compute/Compiler.scala:14-30).  The node set is the
 * reference's (ir/IR.scala:3-41, ir/Ops.scala:3-37).  VarDef/VarRef sharing becomes node-id
 * sharing: a VarDef(sym, rhs) is the node that holds rhs, every VarRef(sym) is its id.
 *
 * All fields are uint32 unless noted; a double is stored as its IEEE-754 bits, low word first.
 *
 *   header   : magic 0x31524952 ("RIR1"), version = 1, n_params, n_targets, n_nodes, kind (0 = density program,
 *              1 = requirements program for rh_requirements_eval: data-free targets, outputs[0] = one requirement each)
 *   targets  : n_targets x { n_cols, reserved = 0, outputs[n_params + 1] }
 *              outputs are node ids: [value, d/d theta_0 .. d/d theta_{n-1}]  (compute/Target.scala:50-56)
 *              target order is the DataFunction's: "prior" first, then one per likelihood.
 *   nodes    : n_nodes variable-length records, topologically ordered (operands have smaller ids)
 *                RIR_CONST  : op, f64 value            (never NaN: compute/ToReal.scala:16-17)
 *                RIR_INPUT  : op, input_index          (DataFunction input layout, below)
 *                binary     : op, a, b                 ADD SUB MUL DIV POW COMPARE
 *                unary      : op, a                    EXP LOG ABS NOOP SIN COS TAN ASIN ACOS ATAN
 *                RIR_LOOKUP : op, index, low (int32), count, table[count]
 *                RIR_SEQ    : op, first, second        value = second (ir/ExprMethodGenerator.scala:64-67)
 *
 * Input layout (ir/DataFunction.scala:3-12): inputs [0, n_params) are the chain parameters; then
 * target 0's n_cols column slots, then target 1's, ...  A node may read column slots of at most one
 * target, and a target's outputs may only reach its own column slots.
 *
 * Semantics: IEEE-754 double, round to nearest; COMPARE = DCMPL;I2D
 * (+1 / 0 / -1, NaN -> -1); LOOKUP = D2I truncation then select table[k - low], out of range is an
 * error (the reference throws); density = sum over targets (in order) and rows of outputs[0],
 * gradient(i) = the same sum of outputs[1+i]  (ir/DataFunction.scala:32-84, core/Model.scala:48-49).
 *
 * Observation data are NOT part of the blob: they are passed beside it as column pointers
 * (rh_model_create), flattened in target order then column order, with one row count per target.
 */
#ifndef RAINIER_HIP_RIR_H
#define RAINIER_HIP_RIR_H

#define RH_RIR_MAGIC 0x31524952u
#define RH_RIR_VERSION 1u
#define RH_RIR_MAX_PARAMS (1u << 24) /* header sanity bound; far above any model the engine can hold */
#define RH_RIR_MAX_COLS (1u << 20)  /* data columns over all targets (the engine keeps their pointers in a device table) */
#define RH_RIR_MAX_TARGETS 65536u    /* targets in a blob; the engine merges runs of data-free targets and then holds at most 64 */

enum rh_rir_op {
  RH_RIR_CONST = 0, RH_RIR_INPUT = 1,
  RH_RIR_ADD = 2, RH_RIR_SUB = 3, RH_RIR_MUL = 4, RH_RIR_DIV = 5, RH_RIR_POW = 6, RH_RIR_COMPARE = 7,
  RH_RIR_EXP = 8, RH_RIR_LOG = 9, RH_RIR_ABS = 10, RH_RIR_NOOP = 11,
  RH_RIR_SIN = 12, RH_RIR_COS = 13, RH_RIR_TAN = 14, RH_RIR_ASIN = 15, RH_RIR_ACOS = 16, RH_RIR_ATAN = 17,
  RH_RIR_LOOKUP = 18, RH_RIR_SEQ = 19
};

#endif
