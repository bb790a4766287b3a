// simplify.cpp -- exact, value-preserving clean-up of an RIR program before it is lowered to HIP.
//
// The reference hands its back end the DAG that symbolic differentiation produced (compute/Gradient.scala): piecewise
// terms arrive as Lookup(Compare(..)) selects (Real.eq / Real.gt, compute/Real.scala:83-99) and their derivatives as
// selects of selects.  The JVM back end emits them literally (ir/ExprMethodGenerator.scala:57-63); on a GPU every fp64
// log / divide / select costs issue slots of the pipe that bounds the kernel, so the obvious redundancies are removed here.
// Every rewrite returns bit-identical values for every input (NaN and infinities included) and keeps the out-of-range
// Lookup error; the only exception is F1, applied in fast mode alone (where FMA contraction already changes last places).
//
//   R1  Lookup(Compare(Compare(p, q), c), T)      -> Lookup(Compare(p, q), T')   the inner compare is -1/0/+1: re-index T
//   R2  Lookup(Compare(..), [x, x, x]) (low = -1) -> x
//   R3  Lookup(k, [f(a), f(b), ..])               -> f(Lookup(k, [a, b, ..]))    f one pure unary op, entries not used elsewhere
//   R4  Lookup(k, C1) (+|-|*|/) Lookup(k, C2)     -> Lookup(k, C1 op C2)         all-constant tables; also against a constant
//   E1  x * 1.0                                   -> x                           exact (the Translator's LogLine products start with 1.0)
//   F1  1 / (1 / x)                               -> x                           fast mode only (<= 1 ulp apart)
//   F2  pow(x, c), c in {-1, 2, -2, 3, -3, 4}     -> multiplications / one division   fast mode only (the Translator's x^c terms)
//   CSE identical (op, a, b) nodes are built once (what VarDef/VarRef sharing already guarantees in the reference)
#include "rir.hpp"
#include "../../include/rainier_hip_rir.h"
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>

namespace rh {
namespace {

bool unary_op(uint32_t op) { return (op >= RH_RIR_EXP && op <= RH_RIR_ATAN); }
bool arith_op(uint32_t op) { return op >= RH_RIR_ADD && op <= RH_RIR_DIV; }

struct Builder {
  Program Q;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;
  std::map<uint64_t, uint32_t> consts;
  std::map<uint32_t, uint32_t> inputs;

  uint32_t push(const Node &n) { Q.nodes.push_back(n); return (uint32_t)Q.nodes.size() - 1; }
  uint32_t dep2(uint32_t a, uint32_t b) const { return Q.nodes[a].dep ? Q.nodes[a].dep : Q.nodes[b].dep; }
  uint32_t constant(double v) {
    uint64_t bits; std::memcpy(&bits, &v, 8);
    auto it = consts.find(bits);
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    return consts[bits] = push(n);
  }
  uint32_t input(const Node &old) {
    auto it = inputs.find(old.input);
    if (it != inputs.end()) return it->second;
    Node n; n.op = RH_RIR_INPUT; n.input = old.input; n.dep = old.dep;
    return inputs[old.input] = push(n);
  }
  uint32_t op2(uint32_t op, uint32_t a, uint32_t b) {
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = dep2(a, b);
    return cons[key] = push(n);
  }
  uint32_t op1(uint32_t op, uint32_t a) {
    auto key = std::make_tuple(op, a, 0xffffffffu);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = Q.nodes[a].dep;
    return cons[key] = push(n);
  }
  uint32_t lookup_raw(uint32_t idx, int32_t low, const std::vector<uint32_t> &table) {
    Node n; n.op = RH_RIR_LOOKUP; n.a = idx; n.low = low; n.table = table; n.dep = Q.nodes[idx].dep;
    for (uint32_t e : table) if (Q.nodes[e].dep) n.dep = Q.nodes[e].dep;
    return push(n);
  }
  bool is_const(uint32_t id) const { return Q.nodes[id].op == RH_RIR_CONST; }
  bool all_const(const Node &n) const {
    for (uint32_t e : n.table) if (!is_const(e)) return false;
    return true;
  }
};

double fold(uint32_t op, double x, double y) {
  switch (op) {
    case RH_RIR_ADD: return x + y;
    case RH_RIR_SUB: return x - y;
    case RH_RIR_MUL: return x * y;
    default: return x / y;
  }
}
int dcmpl(double x, double y) { return x > y ? 1 : (x == y ? 0 : -1); }  // DCMPL: NaN -> -1

}  // namespace

Program simplify(const Program &P, bool fast) {
  // how often each node of the input program is referenced (sinking a unary op only pays when its operands die with it)
  std::vector<uint32_t> uses(P.nodes.size(), 0);
  for (const Node &n : P.nodes) {
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    if (n.op == RH_RIR_LOOKUP) { uses[n.a]++; for (uint32_t e : n.table) uses[e]++; continue; }
    uses[n.a]++;
    if (!unary_op(n.op)) uses[n.b]++;
  }
  for (const Target &t : P.targets) for (uint32_t o : t.outputs) uses[o]++;

  Builder B;
  B.Q.n_params = P.n_params; B.Q.n_inputs = P.n_inputs; B.Q.n_cols_total = P.n_cols_total; B.Q.kind = P.kind;
  B.Q.targets = P.targets; B.Q.col_domain = P.col_domain;
  std::vector<uint32_t> m(P.nodes.size(), 0);
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST) { m[i] = B.constant(n.cval); continue; }
    if (n.op == RH_RIR_INPUT) { m[i] = B.input(n); continue; }
    if (unary_op(n.op)) { m[i] = B.op1(n.op, m[n.a]); continue; }
    if (n.op != RH_RIR_LOOKUP) {
      const uint32_t a = m[n.a], b = m[n.b];
      if (arith_op(n.op)) {  // R4
        const Node na = B.Q.nodes[a], nb = B.Q.nodes[b];  // copies: the node vector grows below
        const bool la = na.op == RH_RIR_LOOKUP && B.all_const(na), lb = nb.op == RH_RIR_LOOKUP && B.all_const(nb);
        if (la && lb && na.a == nb.a && na.low == nb.low && na.table.size() == nb.table.size() && uses[n.a] == 1 && uses[n.b] == 1) {
          // a folded entry that is NaN (inf - inf, 0 * inf, 0 / 0) cannot be written as an RIR constant: leave the arithmetic
          // to the device, which evaluates it to the same NaN the reference's bytecode would
          std::vector<double> fv;
          bool nan = false;
          for (size_t e = 0; e < na.table.size(); e++) {
            fv.push_back(fold(n.op, B.Q.nodes[na.table[e]].cval, B.Q.nodes[nb.table[e]].cval));
            nan = nan || fv.back() != fv.back();
          }
          if (!nan) {
            std::vector<uint32_t> t;
            for (double v : fv) t.push_back(B.constant(v));
            const uint32_t idx = na.a; const int32_t low = na.low;
            m[i] = B.lookup_raw(idx, low, t);
            continue;
          }
        }
        if ((la && B.is_const(b) && uses[n.a] == 1) || (lb && B.is_const(a) && uses[n.b] == 1)) {
          const Node &lk = la ? na : nb;  // (na / nb are local copies)
          std::vector<double> fv;
          bool nan = false;
          for (uint32_t e : lk.table) {
            fv.push_back(la ? fold(n.op, B.Q.nodes[e].cval, nb.cval) : fold(n.op, na.cval, B.Q.nodes[e].cval));
            nan = nan || fv.back() != fv.back();
          }
          if (!nan) {
            std::vector<uint32_t> t;
            for (double v : fv) t.push_back(B.constant(v));
            const uint32_t idx = lk.a; const int32_t low = lk.low;
            m[i] = B.lookup_raw(idx, low, t);
            continue;
          }
        }
      }
      // E1 (exact, both modes): x * 1.0 -> x.  IEEE multiplication by one returns its operand bit for bit (NaN payloads, signed
      // zeros and infinities included).  The reference's Translator starts EVERY LogLine product with the factor 1.0
      // (makeLine(ax, Constant.One, powRing): `if (b.isZero) terms else (b, Const(1.0)) :: terms`, compute/Translator.scala:67-89)
      if (n.op == RH_RIR_MUL && B.is_const(a) && B.Q.nodes[a].cval == 1.0) { m[i] = b; continue; }
      if (n.op == RH_RIR_MUL && B.is_const(b) && B.Q.nodes[b].cval == 1.0) { m[i] = a; continue; }
      // F2 (fast mode only): Math.pow with the constant exponents the Translator emits for LogLine terms other than 1 and 2
      // (compute/Translator.scala:101-114: x^-1 for every division, x^-2, x^3 ...) -> multiplications and ONE division; within
      // 1-2 ulp of the correctly rounded power, against ~100 instructions for a general fp64 pow.  (+-0.5 -> sqrt at emission.)
      if (fast && n.op == RH_RIR_POW && B.is_const(b)) {
        const double e = B.Q.nodes[b].cval;
        if (e == -1.0) { m[i] = B.op2(RH_RIR_DIV, B.constant(1.0), a); continue; }
        if (e == 2.0) { m[i] = B.op2(RH_RIR_MUL, a, a); continue; }
        if (e == -2.0) { m[i] = B.op2(RH_RIR_DIV, B.constant(1.0), B.op2(RH_RIR_MUL, a, a)); continue; }
        if (e == 3.0) { m[i] = B.op2(RH_RIR_MUL, B.op2(RH_RIR_MUL, a, a), a); continue; }
        if (e == -3.0) { m[i] = B.op2(RH_RIR_DIV, B.constant(1.0), B.op2(RH_RIR_MUL, B.op2(RH_RIR_MUL, a, a), a)); continue; }
        if (e == 4.0) { const uint32_t sq = B.op2(RH_RIR_MUL, a, a); m[i] = B.op2(RH_RIR_MUL, sq, sq); continue; }
      }
      // fast mode only (results may differ in the last place, like FMA contraction does): 1 / (1 / x) -> x
      if (fast && n.op == RH_RIR_DIV && B.is_const(a) && B.Q.nodes[a].cval == 1.0 && B.Q.nodes[b].op == RH_RIR_DIV &&
          B.is_const(B.Q.nodes[b].a) && B.Q.nodes[B.Q.nodes[b].a].cval == 1.0) {
        m[i] = B.Q.nodes[b].b;
        continue;
      }
      m[i] = B.op2(n.op, a, b);
      continue;
    }
    // ---- LOOKUP
    uint32_t idx = m[n.a];
    int32_t low = n.low;
    std::vector<uint32_t> table;
    for (uint32_t e : n.table) table.push_back(m[e]);
    for (;;) {  // R1
      const Node I = B.Q.nodes[idx];
      if (I.op != RH_RIR_COMPARE || B.Q.nodes[I.a].op != RH_RIR_COMPARE || !B.is_const(I.b)) break;
      const double c = B.Q.nodes[I.b].cval;
      std::vector<uint32_t> t;
      bool ok = true;
      for (int s = -1; s <= 1 && ok; s++) {
        const long pos = (long)dcmpl((double)s, c) - low;
        if (pos < 0 || pos >= (long)table.size()) ok = false; else t.push_back(table[(size_t)pos]);
      }
      if (!ok) break;
      idx = I.a; low = -1; table = t;
    }
    const bool cmp3 = B.Q.nodes[idx].op == RH_RIR_COMPARE && low == -1 && table.size() == 3;
    if (cmp3 && table[0] == table[1] && table[1] == table[2]) { m[i] = table[0]; continue; }  // R2
    {  // R3
      const uint32_t f = B.Q.nodes[table[0]].op;
      bool sink = unary_op(f);
      for (size_t e = 0; e < n.table.size() && sink; e++) {
        if (B.Q.nodes[table[e]].op != f) { sink = false; break; }
        uint32_t occ = 0;
        for (uint32_t e2 : n.table) occ += e2 == n.table[e];
        if (P.nodes[n.table[e]].op != f || uses[n.table[e]] != occ) sink = false;  // the entry must die with this select
      }
      if (sink) {
        std::vector<uint32_t> args;
        for (uint32_t e : table) args.push_back(B.Q.nodes[e].a);
        m[i] = B.op1(f, B.lookup_raw(idx, low, args));
        continue;
      }
    }
    m[i] = B.lookup_raw(idx, low, table);
  }
  for (Target &t : B.Q.targets) for (uint32_t &o : t.outputs) o = m[o];
  // drop what the rewrites orphaned (operands always precede their users, so one backward sweep marks everything live)
  Program &Q = B.Q;
  std::vector<char> live(Q.nodes.size(), 0);
  for (const Target &t : Q.targets) for (uint32_t o : t.outputs) live[o] = 1;
  for (size_t i = Q.nodes.size(); i-- > 0;) {
    if (!live[i]) continue;
    const Node &n = Q.nodes[i];
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    live[n.a] = 1;
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) live[e] = 1; }
    else if (!unary_op(n.op)) live[n.b] = 1;
  }
  std::vector<uint32_t> renum(Q.nodes.size(), 0);
  std::vector<Node> kept;
  for (size_t i = 0; i < Q.nodes.size(); i++) {
    if (!live[i]) continue;
    Node n = Q.nodes[i];
    if (n.op != RH_RIR_CONST && n.op != RH_RIR_INPUT) {
      n.a = renum[n.a];
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : n.table) e = renum[e]; }
      else if (!unary_op(n.op)) n.b = renum[n.b];
    }
    renum[i] = (uint32_t)kept.size();
    kept.push_back(std::move(n));
  }
  Q.nodes.swap(kept);
  for (Target &t : Q.targets) for (uint32_t &o : t.outputs) o = renum[o];
  return Q;
}

}  // namespace rh
