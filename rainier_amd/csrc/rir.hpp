// rir.hpp -- in-memory form of an RIR program (include/rainier_hip_rir.h) and the IR -> HIP emitter.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rh {
// ---- diagnostic knobs ------------------------------------------------------------------------------------------------------------
// The engine reads its experiment / test switches (RH_FUSE, RH_COMPACT, RH_GATHER_*, RH_GLM_*, the *_WHY traces, ...: DESIGN appendix)
// through knob(): they exist only in a process that has RH_DIAG=1 in its environment -- tests/conftest.py and the tools/ scripts set
// it; nothing else does, so in normal use no environment variable changes which kernels run, how they are built or how results
// round.  (What stays readable without it is operational: where the kernel cache lives, RH_NO_KERNEL_CACHE, RH_COMM_TIMEOUT_S.)
// unsafe_knob(): the two switches that turn the correctness guards off (RH_ALLOW_UNHEALTHY: launch kernels kernel_health refused;
// RH_SELFCHECK=0: skip the create-time self-checks) are compiled out of the library altogether unless it is built with
// -DRH_DIAG_UNSAFE (`make unsafe-diag`: a separate file for reproducing compiler faults on a GPU, never the one that ships).
const char *knob(const char *name);
inline const char *unsafe_knob(const char *name) {
#ifdef RH_DIAG_UNSAFE
  return knob(name);
#else
  (void)name;
  return nullptr;
#endif
}


struct Node {
  uint32_t op = 0, a = 0, b = 0;
  double cval = 0.0;
  uint32_t input = 0;
  int32_t low = 0;
  std::vector<uint32_t> table;
  uint32_t dep = 0;  // 0: parameters only; t+1: reads a column of target t
};

struct Target {
  uint32_t n_cols = 0;
  uint32_t input_start = 0;  // first DataFunction input slot of this target's columns
  uint32_t col0 = 0;         // index of its first column in the flattened column list
  std::vector<uint32_t> outputs;  // [n_params + 1]
};

struct Program {
  uint32_t n_params = 0, n_inputs = 0, n_cols_total = 0, kind = 0;  // kind 1 = requirements program
  std::vector<Target> targets;
  std::vector<Node> nodes;
  // per data column (flattened order): its distinct values when there are at most 8 of them, else empty; empty vector = not
  // analysed.  Filled by canonicalize_columns from the data the model is created with.
  std::vector<std::vector<double>> col_domain;
};

// Parses and validates a blob; returns false and sets err on malformed input.
bool parse_rir(const void *buf, size_t len, Program &out, std::string &err);

std::vector<unsigned char> write_rir(const Program &p);

// When the program has more targets than the engine holds, merges every run of consecutive data-free targets into one (exact);
// old_target_of[new target] = index of the first original target it stands for (row targets keep their identity).
bool merge_data_free_targets(Program &P, std::vector<uint32_t> &old_target_of);

// lift.cpp: when the program has more targets than the engine holds, the largest group (>= 32) of data-free targets of one shape
// -- the same expression with different constants folded in: one Model.observe per observation -- becomes ONE row target whose
// columns (returned in synth, one row per member) are the constants that differ.  old_target_of as above; 0xFFFFFFFF marks the
// synthesised target (appended last; its columns follow the caller's).
bool lift_constants(Program &P, std::vector<std::vector<double>> &synth, std::vector<uint32_t> &old_target_of);
void recompute_deps(Program &P);
// A table of >= gather_min trailing parameters indexed by a data column whose PRIOR sits in a data-free target (where the
// reference's front end puts it): the per-entry prior terms become one more row target over a synthesised index column 0..G-1
// (appended to `synth`; the target is appended to the list), so that every gradient with respect to a table entry comes from
// row targets -- the precondition of gather mode (emit.cpp).  Returns false (and changes nothing) when the shape is not there.
// rederive_ok (fast builds): a prior that ties the entries to SHARED parameters -- the centred parameterisation, entries
// alpha_k ~ Normal(mu, sigma) written with Real.parameter (compute/Real.scala:63-78) -- is lifted too: the per-entry value terms
// become the row target as before, and because the reference hands the shared parameters' gradients over as sums that run over
// all entries (not per-entry expressions), both targets' gradients are derived again from their values and the rewrite is
// accepted only if, at random points, the new outputs add up to the original ones.
bool lift_table_priors(Program &P, std::vector<std::vector<double>> &synth, int gather_min, bool rederive_ok = false);
// Lookup(column, [f(z_0), f(z_1), ...]) -> f(Lookup(column, [z_0, z_1, ...])) when every entry is the same function of one of
// >= gather_min consecutive trailing parameters (Normal(mu, sd).latentVec(G)(site): z_k * sd + mu) and it is the program's only
// column-indexed table: the same arithmetic on the selected entry (bit-identical), and the shape gather mode reads.
bool hoist_table_maps(Program &P, int gather_min);
// A target whose expression is data-free and reaches exactly ONE entry of a gather-shaped table -- Model.observe's initial chunk when
// it has a single row: the front end folds a single observation into constants, Lookup(constant, table) into the entry -- becomes a
// one-row target over a synthesised index column, so that the entry is read through the gather again.  Appends to `synth`, returns
// the number of targets appended (each after the existing ones).
int lift_single_entry_targets(Program &P, std::vector<std::vector<double>> &synth, int gather_min);

// Column canonicalisation (columns.cpp): derived columns (copies, negations, products, affine images of earlier columns,
// constants) are replaced by expressions over the base columns.  kept[new global column] = caller's column index.  Returns
// true when the program was rewritten.
bool canonicalize_columns(Program &P, const double *const *columns, const int64_t *nrows, bool fast, std::vector<uint32_t> &kept,
                          std::string &err, bool allow_unroll = true);
// RH_INDEX_MASKS=1/0: strict builds recognise the per-entry mask columns of a Lookup's index column (columns.cpp) and gather mode
// reads a table gradient through the NOOP the front end may have wrapped it in (emit.cpp).  Off until it has run on the device.
bool index_masks_on();

// Fast-mode re-association of row targets after canonicalize_columns (refactor.cpp): products are merged into monomials
// and the factor common to every term of an output is pulled out, so that x_k * w shapes reappear.
// With `parts`, targets written by Model.observe's 8-way split are also rolled back into rows: parts[new column] = the old
// columns (flattened index) whose data is concatenated into it, in order (0xFFFFFFFF = a block of zeros as long as the same
// block of the target's first column); a target's row count is the sum of its parts'.
Program refactor(const Program &p, std::vector<std::vector<uint32_t>> *parts = nullptr);

// Strict builds (rollstrict.cpp): Model.observe's 8 slots rolled back into rows without re-association -- the per-slot terms
// are kept operation for operation, only the shared (parameter-only) terms are scaled by the exact 1 / S.  parts as for refactor.
bool roll_strict(Program &P, std::vector<std::vector<uint32_t>> &parts);

// Fast mode (rederive.cpp): the gradient outputs of every streamed target re-derived from its value output by reverse-mode
// differentiation and VERIFIED against the supplied ones on sample rows (cols[c] = host data of column c, nrows per target);
// targets that do not verify keep their outputs.
Program rederive_gradients(const Program &p, const std::vector<const double *> &cols, const int64_t *nrows, bool *changed = nullptr);

// rederive.cpp, for the loader's own rewrites: the gradient of `value` with respect to every parameter as new nodes of P, and the
// extended-precision block interpreter the rewrites are verified with (inputs[i * B + r] = input i at block row r)
std::vector<uint32_t> derive_gradient(Program &P, uint32_t value);
struct BlockEvaluator {   // val[node * B + r]; ok[r] = 0 where a Lookup index left its table; P must outlive the evaluator
  BlockEvaluator(const Program &P, const std::vector<uint32_t> &roots);
  ~BlockEvaluator();
  BlockEvaluator(const BlockEvaluator &) = delete;
  BlockEvaluator &operator=(const BlockEvaluator &) = delete;
  bool run(const std::vector<long double> &inputs, int B, std::vector<long double> &val, std::vector<char> &ok) const;
 private:
  void *impl;
};

// Exact clean-up of the DAG (select-of-select folding, select sinking, constant selects, CSE): simplify.cpp
Program simplify(const Program &p, bool fast = false);

struct EmitOptions {
  bool fast_log = true;      // fast mode: rh_fast_log (<= 1 ulp, ~40 instructions) instead of the device library's log (RH_FAST_LOG=0)
  bool pack = true;          // data-free models with <= 32 parameters: several chains per wavefront (RH_PACK=0 switches it off)
  bool simplify = true;      // run simplify() before lowering (RH_SIMPLIFY=0 switches it off)
  bool strict_math = false;  // EXP/LOG -> fdlibm
  bool fp_contract = false;  // allow FMA contraction in model code
  int rows_unroll = 4;
  bool factor_outputs = false;  // peel invariant affine wrappers off the accumulated outputs (fast mode)
  int grad_chains = 0;  // chains per wavefront in the batched gradient kernel (0 = default)
  int grad_unroll = 0;
  bool force_bign = false;  // tests: HBM-resident chain vectors (big mode) even for small models
  int gather_min = 65;   // Lookup tables of at least this many trailing parameters switch the model to gather mode
  bool glm_mfma = true;  // with factor_outputs: lower dense linear predictors to the fp64 MFMA kernel
  bool logit_link = true;  // fast mode: a VERIFIED Bernoulli-logit scalar part is emitted in closed form (RH_LOGIT_LINK=0 switches it off)
  bool fma_adds = false; // opt-in (RH_FMA_ADDS=1), per-row code: every fp64 add/sub as v_fma_f64(x, +-1.0, y) (same rounding).  Measured: no gain on
                         // cfg 2 -- the kernel already sits at ~88 % of the fp64 issue ceiling (profiles/r1_d_fp64_ceiling)
  bool xfuse = true;     // with fp_contract: the row code of streamed targets fuses mul+add explicitly and is compiled with contraction off, so that every
                         // inlined copy of it rounds alike (emit.cpp: TargetEmitter::xfuse; RH_XFUSE=0 gives the compiler's own contraction back)
  int chunk = 0;          // > 0: memory-resident lowering (emit.cpp chunk_body): generated functions are cut into chunks of at most this many
                          // statement groups and values travel between chunks through a per-lane scratch array; the engine's last resort for heavy models
  int big_unroll = 16;    // big mode (chain vectors in HBM): slots in flight per lane in the vector loops (RH_BIGU); halved by the engine while a sampler kernel does not fit
  int chain_waves = 2;    // wavefronts per SIMD rh_chain_kernel asks for (2: 256 registers; the engine falls back to 1 when the kernel does not fit)
  int grad_pipeline = 2;  // row loop of the batched gradient kernel: 0 plain, 1 double-buffered, 2 rolling (a tile's registers are reloaded as soon as it is consumed)
  bool const_pool = true; // the non-trivial constants of data-free targets are read from a per-model buffer, not spelled in the source (RH_CONST_POOL=0 switches it off)
};

// What the engine needs to know about the lowered program (besides the source text)
struct EmitInfo {
  bool gather_mode = false, bign = false;
  int n_shared = 0, grad_k = 4, nacc_max = 1, glm_target = -1;
  int pack_l = 64;  // lanes per chain in the chain / density kernels (64 = one chain per wavefront)
  bool glm_small = false;
  struct TargetInfo { bool has_rows = false, has_gather = false; int g_col = -1, g_count = 0, g_low = 0; };
  std::vector<TargetInfo> targets;
  // The constant pool (round 6: compile once per model shape).  Model.observe's initial chunk (core/Model.scala:71-132) and every
  // single observation arrive folded into data-free targets: their observations are CONSTANT nodes.  Spelled as literals they
  // made the generated source -- the key of the code-object cache -- a function of the data; now a data-free target reads every
  // constant that is not a small dyadic number from this buffer (`c[i]` in its row(): rh_model_data.kpool on the device), so two
  // data sets of one model share a translation unit whenever their programs have the same structure.
  std::vector<double> kpool;
};

// Lowers the program to the per-model part of the HIP translation unit (defines + rh_target<t> structs).
// The replacement for the reference's ASM generators (ir/ExprMethodGenerator.scala, ir/OutputClassGenerator.scala).
bool emit_hip(const Program &p, const EmitOptions &o, std::string &defines, std::string &targets, std::string &err,
              EmitInfo *info = nullptr);

// Lowers a requirements program (kind 1) to  rh_req_eval(th, out, err)  + defines RH_NVARS / RH_NREQ.
bool emit_requirements(const Program &p, const EmitOptions &o, std::string &defines, std::string &body, std::string &err);

// ---- isacheck.cpp: what the engine reads out of a code object before it agrees to launch one of its kernels -----------------
struct KernelMeta {
  bool found = false;
  long vgprs = -1, sgprs = -1, vgpr_spills = -1, sgpr_spills = -1, scratch_bytes = -1;
};
// the kernel's entry of the AMDGPU metadata note; false (found = false) when any of the fields is missing
bool kernel_meta(const std::vector<char> &code, const std::string &name, KernelMeta &out);
// the kernels of a code object (STT_FUNC symbols of .text with a kernel descriptor; kernels_only = false: device functions too)
bool list_kernels(const std::vector<char> &code, std::vector<std::string> &names, bool kernels_only = true);
// 1 / 0: the kernel's own code does / does not access private (scratch) memory; -1: it cannot be walked
int kernel_touches_scratch(const std::vector<char> &code, const std::string &name);
// instruction offsets of one kernel, relative to its first byte (the decoder's own test compares them with llvm-objdump)
bool kernel_instruction_offsets(const std::vector<char> &code, const std::string &name, std::vector<uint32_t> &offs);
// static check for vector instructions ahead of a join block's exec restore (this toolchain's register-allocator fault, see
// isacheck.cpp); only_kernel empty = every kernel.  Returns false when the code could not be walked; findings are text lines.
// *unproven (optional): exec restores behind vector instructions of their own block whose block could NOT be proven to be a join
// block (no s_cbranch_execz / execnz witness) -- not findings, but what the rule cannot see
bool check_code_object(const std::vector<char> &code, const std::string &only_kernel, std::vector<std::string> &findings, int *unproven = nullptr);

}  // namespace rh
