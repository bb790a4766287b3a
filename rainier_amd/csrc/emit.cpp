// emit.cpp -- IR -> HIP lowering.  One straight-line fp64 function pair per target:
//   invariants(th, inv) : the sub-DAG that depends on parameters only (hoisted out of the row loop; the
//                         reference re-evaluates it for every row inside its generated methods)
//   row(th, inv, c, acc): the per-row sub-DAG; acc[o] += output_o, o = 0..n   (DataFunction.scala:64-71)
// The whole DAG is evaluated once per row -- the reference's "outputs must be evaluated in index order so
// that a VarDef precedes its VarRefs" constraint disappears in a fused kernel.
#include <cmath>
#include <cstdio>
#include <map>
#include <sstream>

#include "../../include/rainier_hip_rir.h"
#include "rir.hpp"

namespace rh {
namespace {

std::string lit(double v) {
  if (std::isinf(v)) return v > 0 ? "RH_INF" : "(-RH_INF)";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%a", v);
  std::string s(buf);
  return v < 0 || std::signbit(v) ? "(" + s + ")" : s;
}

struct TargetEmitter {
  const Program &P;
  uint32_t t;
  std::vector<char> reach;
  std::map<uint32_t, int> inv_slot;  // non-trivial invariant node -> index in inv[]
  TargetEmitter(const Program &p, uint32_t ti) : P(p), t(ti), reach(p.nodes.size(), 0) {
    for (uint32_t o : P.targets[t].outputs) reach[o] = 1;
    for (size_t n = P.nodes.size(); n-- > 0;) {
      if (!reach[n]) continue;
      const Node &nd = P.nodes[n];
      switch (nd.op) {
        case RH_RIR_CONST: case RH_RIR_INPUT: break;
        case RH_RIR_LOOKUP: reach[nd.a] = 1; for (uint32_t e : nd.table) reach[e] = 1; break;
        case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_POW: case RH_RIR_COMPARE:
        case RH_RIR_SEQ: reach[nd.a] = 1; reach[nd.b] = 1; break;
        default: reach[nd.a] = 1;
      }
    }
  }
  bool trivial(uint32_t id) const { return P.nodes[id].op == RH_RIR_CONST || P.nodes[id].op == RH_RIR_INPUT; }
  bool has_rows() const { return P.targets[t].n_cols > 0; }
  // how an operand is spelled inside row() (in_row) or invariants()
  std::string ref(uint32_t id, bool in_row) const {
    const Node &nd = P.nodes[id];
    if (nd.op == RH_RIR_CONST) return lit(nd.cval);
    if (nd.op == RH_RIR_INPUT) {
      if (nd.input < P.n_params) return "th[" + std::to_string(nd.input) + "]";
      return "c[" + std::to_string(nd.input - P.targets[t].input_start) + "]";
    }
    if (in_row && has_rows() && nd.dep == 0) return "inv[" + std::to_string(inv_slot.at(id)) + "]";
    return "n" + std::to_string(id);
  }
  bool emit_node(std::ostringstream &os, uint32_t id, bool in_row, std::string &err) const {
    const Node &nd = P.nodes[id];
    auto R = [&](uint32_t x) { return ref(x, in_row); };
    const std::string lhs = "    const double n" + std::to_string(id) + " = ";
    switch (nd.op) {
      case RH_RIR_ADD: os << lhs << R(nd.a) << " + " << R(nd.b) << ";\n"; break;
      case RH_RIR_SUB: os << lhs << R(nd.a) << " - " << R(nd.b) << ";\n"; break;
      case RH_RIR_MUL: os << lhs << R(nd.a) << " * " << R(nd.b) << ";\n"; break;
      case RH_RIR_DIV: os << lhs << R(nd.a) << " / " << R(nd.b) << ";\n"; break;
      case RH_RIR_POW: os << lhs << "rh_java_pow(" << R(nd.a) << ", " << R(nd.b) << ");\n"; break;
      case RH_RIR_COMPARE: os << lhs << "rh_compare(" << R(nd.a) << ", " << R(nd.b) << ");\n"; break;
      case RH_RIR_EXP: os << lhs << "RH_EXP(" << R(nd.a) << ");\n"; break;
      case RH_RIR_LOG: os << lhs << "RH_LOG(" << R(nd.a) << ");\n"; break;
      case RH_RIR_ABS: os << lhs << "__builtin_fabs(" << R(nd.a) << ");\n"; break;
      case RH_RIR_NOOP: os << lhs << R(nd.a) << ";\n"; break;
      case RH_RIR_SIN: os << lhs << "sin(" << R(nd.a) << ");\n"; break;
      case RH_RIR_COS: os << lhs << "cos(" << R(nd.a) << ");\n"; break;
      case RH_RIR_TAN: os << lhs << "tan(" << R(nd.a) << ");\n"; break;
      case RH_RIR_ASIN: os << lhs << "asin(" << R(nd.a) << ");\n"; break;
      case RH_RIR_ACOS: os << lhs << "acos(" << R(nd.a) << ");\n"; break;
      case RH_RIR_ATAN: os << lhs << "atan(" << R(nd.a) << ");\n"; break;
      case RH_RIR_SEQ: os << lhs << R(nd.b) << ";\n"; break;  // evaluate first, POP2, evaluate second
      case RH_RIR_LOOKUP: {
        if (nd.table.size() > 64) { err = "Lookup tables with more than 64 entries are not supported yet"; return false; }
        const std::string k = "k" + std::to_string(id);
        os << "    const int " << k << " = rh_d2i(" << R(nd.a) << ") - (" << nd.low << ");\n";
        os << lhs;
        for (size_t e = 0; e + 1 < nd.table.size(); e++) os << "(" << k << " == " << e << ") ? " << R(nd.table[e]) << " : ";
        os << R(nd.table.back()) << ";\n";
        os << "    if ((unsigned)" << k << " >= " << nd.table.size() << "u) err = 1;\n";
        break;
      }
      default: err = "emit: unexpected opcode"; return false;
    }
    return true;
  }
  bool emit(std::ostringstream &os, std::string &err) {
    const Target &T = P.targets[t];
    const bool rows = has_rows();
    // invariant frontier: non-trivial parameter-only nodes that row nodes (or outputs) read
    if (rows) {
      auto want = [&](uint32_t x) {
        if (P.nodes[x].dep == 0 && !trivial(x) && !inv_slot.count(x)) { int s = (int)inv_slot.size(); inv_slot[x] = s; }
      };
      for (size_t n = 0; n < P.nodes.size(); n++) {
        if (!reach[n] || P.nodes[n].dep == 0) continue;
        const Node &nd = P.nodes[n];
        switch (nd.op) {
          case RH_RIR_INPUT: break;
          case RH_RIR_LOOKUP: want(nd.a); for (uint32_t e : nd.table) want(e); break;
          case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_POW: case RH_RIR_COMPARE:
          case RH_RIR_SEQ: want(nd.a); want(nd.b); break;
          default: want(nd.a);
        }
      }
      for (uint32_t o : T.outputs) want(o);
    }
    os << "template <> struct rh_target<" << t << "> {\n";
    os << "  static constexpr int NCOLS = " << T.n_cols << ", COL0 = " << T.col0 << ", NINV = " << inv_slot.size() << ";\n";
    os << "  static constexpr bool HAS_ROWS = " << (rows ? "true" : "false") << ";\n";
    int rowt = -1;
    if (rows) { rowt = 0; for (uint32_t u = 0; u < t; u++) if (P.targets[u].n_cols) rowt++; }
    os << "  static constexpr int ROWT = " << rowt << ";\n";
    os << "  static RH_DEV void invariants(const double (&th)[RH_NVARS], double *inv, int &err) {\n    (void)th; (void)inv; (void)err;\n";
    if (rows) {
      for (size_t n = 0; n < P.nodes.size(); n++)
        if (reach[n] && P.nodes[n].dep == 0 && !trivial((uint32_t)n))
          if (!emit_node(os, (uint32_t)n, false, err)) return false;
      for (auto &kv : inv_slot) os << "    inv[" << kv.second << "] = n" << kv.first << ";\n";
    }
    os << "  }\n";
    os << "  static RH_DEV void row(const double (&th)[RH_NVARS], const double *inv, const double *c, double (&acc)[RH_NOUT], int &err) {\n"
          "    (void)th; (void)inv; (void)c; (void)err;\n";
    for (size_t n = 0; n < P.nodes.size(); n++) {
      if (!reach[n] || trivial((uint32_t)n)) continue;
      if (rows && P.nodes[n].dep == 0) continue;  // hoisted
      if (!emit_node(os, (uint32_t)n, true, err)) return false;
    }
    for (size_t o = 0; o < T.outputs.size(); o++) {
      const Node &on = P.nodes[T.outputs[o]];
      if (on.op == RH_RIR_CONST && on.cval == 0.0 && !std::signbit(on.cval)) continue;  // += +0.0 is the identity
      os << "    acc[" << o << "] += " << ref(T.outputs[o], true) << ";\n";
    }
    os << "  }\n};\n";
    return true;
  }
};

}  // namespace

bool emit_hip(const Program &P, const EmitOptions &o, std::string &defines, std::string &targets, std::string &err) {
  std::ostringstream d;
  d << "#define RH_NVARS " << P.n_params << "\n#define RH_NOUT " << (P.n_params + 1) << "\n#define RH_SLOTS "
    << ((P.n_params + 63) / 64) << "\n#define RH_NTARGETS " << P.targets.size() << "\n#define RH_FP_CONTRACT "
    << (o.fp_contract ? 1 : 0) << "\n#define RH_ROWS_UNROLL " << (o.rows_unroll > 0 ? o.rows_unroll : 4) << "\n";
  int nrowt = 0;
  for (auto &T : P.targets) if (T.n_cols) nrowt++;
  d << "#define RH_NROWTARGETS " << nrowt << "\n#define RH_GRAD_K " << (o.grad_chains > 0 ? o.grad_chains : 4)
    << "\n#define RH_GRAD_U " << (o.grad_unroll > 0 ? o.grad_unroll : 2) << "\n";
  if (o.strict_math) d << "#define RH_EXP(x) rh_strict_exp(x)\n#define RH_LOG(x) rh_strict_log(x)\n";
  else d << "#define RH_EXP(x) exp(x)\n#define RH_LOG(x) log(x)\n";
  defines = d.str();
  std::ostringstream os;
  os << "template <int T> struct rh_target;\n";
  if (o.fp_contract) os << "#pragma clang fp contract(fast)\n";
  for (uint32_t t = 0; t < P.targets.size(); t++) {
    TargetEmitter te(P, t);
    if (!te.emit(os, err)) return false;
  }
  os << "#pragma clang fp contract(off)\n";
  targets = os.str();
  return true;
}
}  // namespace rh
