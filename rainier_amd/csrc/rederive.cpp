// rederive.cpp -- fast mode: the gradient outputs of a streamed target, derived again from its value output.
//
// What the reference hands over as "the gradient" is the result of its own symbolic differentiation AFTER its Line / LogLine
// algebra has distributed every data-only factor into a column of its own (compute/Gradient.scala, compute/Target.scala:27-31):
// masks of a Lookup's index column per table entry, x_k * y products, ...  The value output, in contrast, still reads the
// observation columns themselves: Lookup(site, alphas), a + x . b, ...  Reverse-mode differentiation of THAT expression
// (the same rules as compute/Gradient.scala:60-160: d Lookup / d entry_j = eq(index, low + j, g, 0), piecewise-constant compares,
// ...) gives the gradient in its natural form -- x_k * w, eq(index, k, g, 0) -- which is what the GLM, closed-form-link and
// gather lowerings recognise, makes the derived columns dead (they are not uploaded), and lets Model.observe's 8 slots be rolled
// back into rows (refactor.cpp) because every slot's part is then the same function of its own columns.
//
// The replacement is VERIFIED: on up to 24 sample rows of the target and 3 random parameter vectors, every supplied gradient
// output must agree with the re-derived one (1e-6 of the row's largest output); a program whose "gradient" is not the derivative
// of its value, or that cannot be evaluated on the host, keeps the outputs it came with.  Rounding changes only: fast mode.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>

#include "../../include/rainier_hip_rir.h"
#include "rir.hpp"

namespace rh {
namespace {

bool binary_op(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }

struct Deriver {
  Program Q;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;
  std::map<uint64_t, uint32_t> consts;

  explicit Deriver(const Program &p) : Q(p) {
    for (uint32_t i = 0; i < Q.nodes.size(); i++) {
      const Node &n = Q.nodes[i];
      if (n.op == RH_RIR_CONST) { uint64_t b; std::memcpy(&b, &n.cval, 8); consts.emplace(b, i); }
      else if (n.op >= RH_RIR_ADD && n.op <= RH_RIR_ATAN) cons.emplace(std::make_tuple(n.op, n.a, binary_op(n.op) ? n.b : 0xffffffffu), i);
    }
  }
  uint32_t k(double v) {
    uint64_t b; std::memcpy(&b, &v, 8);
    auto it = consts.find(b);
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    Q.nodes.push_back(n);
    return consts[b] = (uint32_t)Q.nodes.size() - 1;
  }
  bool is_c(uint32_t id, double v) const { return Q.nodes[id].op == RH_RIR_CONST && Q.nodes[id].cval == v; }
  uint32_t op2(uint32_t op, uint32_t a, uint32_t b) {
    if (op >= RH_RIR_ADD && op <= RH_RIR_DIV && Q.nodes[a].op == RH_RIR_CONST && Q.nodes[b].op == RH_RIR_CONST) {
      const double x = Q.nodes[a].cval, y = Q.nodes[b].cval;
      const double v = op == RH_RIR_ADD ? x + y : op == RH_RIR_SUB ? x - y : op == RH_RIR_MUL ? x * y : x / y;
      if (v == v) return k(v);
    }
    if (op == RH_RIR_MUL) { if (is_c(a, 0.0) || is_c(b, 0.0)) return k(0.0); if (is_c(a, 1.0)) return b; if (is_c(b, 1.0)) return a; }
    if (op == RH_RIR_ADD) { if (is_c(a, 0.0)) return b; if (is_c(b, 0.0)) return a; }
    if (op == RH_RIR_SUB && is_c(b, 0.0)) return a;
    if (op == RH_RIR_DIV) { if (is_c(a, 0.0)) return k(0.0); if (is_c(b, 1.0)) return a; }
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = Q.nodes[a].dep ? Q.nodes[a].dep : Q.nodes[b].dep;
    Q.nodes.push_back(n);
    return cons[key] = (uint32_t)Q.nodes.size() - 1;
  }
  uint32_t op1(uint32_t op, uint32_t a) {
    auto key = std::make_tuple(op, a, 0xffffffffu);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = Q.nodes[a].dep;
    Q.nodes.push_back(n);
    return cons[key] = (uint32_t)Q.nodes.size() - 1;
  }
  uint32_t neg(uint32_t a) { return op2(RH_RIR_MUL, k(-1.0), a); }
  // Real.eq(l, r, t, f) (compute/Real.scala:83-99): Lookup(Compare(l, r), [f, t, f], low = -1)
  uint32_t eq(uint32_t l, uint32_t r, uint32_t t, uint32_t f) {
    if (t == f) return t;
    Node n; n.op = RH_RIR_LOOKUP; n.a = op2(RH_RIR_COMPARE, l, r); n.low = -1; n.table = {f, t, f};
    n.dep = Q.nodes[n.a].dep;
    for (uint32_t e : n.table) if (Q.nodes[e].dep) n.dep = Q.nodes[e].dep;
    Q.nodes.push_back(n);
    return (uint32_t)Q.nodes.size() - 1;
  }

  // d root / d theta_p for every parameter, as new nodes (constants 0 where root does not depend on theta_p)
  std::vector<uint32_t> gradient(uint32_t root) {
    const uint32_t N0 = (uint32_t)Q.nodes.size();
    std::vector<char> live(N0, 0), hasp(N0, 0);
    live[root] = 1;
    for (uint32_t i = N0; i-- > 0;) {
      if (!live[i]) continue;
      const Node &n = Q.nodes[i];
      if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
      live[n.a] = 1;
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) live[e] = 1; }
      else if (binary_op(n.op)) live[n.b] = 1;
    }
    for (uint32_t i = 0; i < N0; i++) {
      const Node &n = Q.nodes[i];
      if (n.op == RH_RIR_INPUT) hasp[i] = n.input < Q.n_params;
      else if (n.op == RH_RIR_CONST) hasp[i] = 0;
      else if (n.op == RH_RIR_LOOKUP) { hasp[i] = hasp[n.a]; for (uint32_t e : n.table) hasp[i] = hasp[i] || hasp[e]; }
      else hasp[i] = hasp[n.a] || (binary_op(n.op) && hasp[n.b]);
    }
    std::map<uint32_t, uint32_t> adj;
    auto acc = [&](uint32_t i, uint32_t g) {
      if (!hasp[i]) return;
      auto it = adj.find(i);
      if (it == adj.end()) adj[i] = g; else it->second = op2(RH_RIR_ADD, it->second, g);
    };
    adj[root] = k(1.0);
    const uint32_t zero = k(0.0), one = k(1.0);
    for (uint32_t i = N0; i-- > 0;) {
      if (!live[i] || !hasp[i]) continue;
      auto ai = adj.find(i);
      if (ai == adj.end()) continue;
      const uint32_t g = ai->second;
      const Node n = Q.nodes[i];  // copy: the node vector grows
      switch (n.op) {
        case RH_RIR_CONST: case RH_RIR_INPUT: case RH_RIR_COMPARE: break;
        case RH_RIR_ADD: acc(n.a, g); acc(n.b, g); break;
        case RH_RIR_SUB: acc(n.a, g); acc(n.b, neg(g)); break;
        case RH_RIR_MUL: acc(n.a, op2(RH_RIR_MUL, g, n.b)); acc(n.b, op2(RH_RIR_MUL, g, n.a)); break;
        case RH_RIR_DIV:
          acc(n.a, op2(RH_RIR_DIV, g, n.b));
          if (hasp[n.b]) acc(n.b, neg(op2(RH_RIR_DIV, op2(RH_RIR_MUL, g, i), n.b)));   // -g * (a / b) / b
          break;
        case RH_RIR_POW:
          if (hasp[n.a]) acc(n.a, op2(RH_RIR_MUL, op2(RH_RIR_MUL, g, n.b), op2(RH_RIR_POW, n.a, op2(RH_RIR_SUB, n.b, one))));
          if (hasp[n.b]) acc(n.b, op2(RH_RIR_MUL, op2(RH_RIR_MUL, g, i), op1(RH_RIR_LOG, eq(n.a, zero, one, n.a))));
          break;
        case RH_RIR_SEQ: acc(n.b, g); break;
        case RH_RIR_EXP: acc(n.a, op2(RH_RIR_MUL, g, i)); break;
        case RH_RIR_LOG: acc(n.a, op2(RH_RIR_DIV, g, n.a)); break;
        case RH_RIR_ABS: acc(n.a, eq(n.a, zero, zero, op2(RH_RIR_DIV, op2(RH_RIR_MUL, g, n.a), i))); break;
        case RH_RIR_NOOP: acc(n.a, g); break;
        case RH_RIR_SIN: acc(n.a, op2(RH_RIR_MUL, g, op1(RH_RIR_COS, n.a))); break;
        case RH_RIR_COS: acc(n.a, neg(op2(RH_RIR_MUL, g, op1(RH_RIR_SIN, n.a)))); break;
        case RH_RIR_TAN: { const uint32_t c = op1(RH_RIR_COS, n.a); acc(n.a, op2(RH_RIR_DIV, g, op2(RH_RIR_MUL, c, c))); break; }
        case RH_RIR_ASIN: acc(n.a, op2(RH_RIR_DIV, g, op2(RH_RIR_POW, op2(RH_RIR_SUB, one, op2(RH_RIR_MUL, n.a, n.a)), k(0.5)))); break;
        case RH_RIR_ACOS: acc(n.a, neg(op2(RH_RIR_DIV, g, op2(RH_RIR_POW, op2(RH_RIR_SUB, one, op2(RH_RIR_MUL, n.a, n.a)), k(0.5))))); break;
        case RH_RIR_ATAN: acc(n.a, op2(RH_RIR_DIV, g, op2(RH_RIR_ADD, one, op2(RH_RIR_MUL, n.a, n.a)))); break;
        case RH_RIR_LOOKUP:
          for (size_t j = 0; j < n.table.size(); j++)
            if (hasp[n.table[j]]) acc(n.table[j], eq(n.a, k((double)n.low + (double)j), g, zero));
          break;
        default: break;
      }
    }
    std::vector<uint32_t> out(Q.n_params, zero);
    for (uint32_t i = 0; i < N0; i++)
      if (Q.nodes[i].op == RH_RIR_INPUT && Q.nodes[i].input < Q.n_params) { auto it = adj.find(i); if (it != adj.end()) out[Q.nodes[i].input] = it->second; }
    return out;
  }
};

// host evaluation of the nodes needed for `roots` (ascending ids are a topological order); false if a lookup goes out of range
bool eval(const Program &P, const std::vector<double> &inputs, const std::vector<uint32_t> &roots, std::vector<double> &val) {
  std::vector<char> live(P.nodes.size(), 0);
  for (uint32_t r : roots) live[r] = 1;
  for (size_t i = P.nodes.size(); i-- > 0;) {
    if (!live[i]) continue;
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    live[n.a] = 1;
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) live[e] = 1; }
    else if (binary_op(n.op)) live[n.b] = 1;
  }
  val.assign(P.nodes.size(), 0.0);
  for (size_t i = 0; i < P.nodes.size(); i++) {
    if (!live[i]) continue;
    const Node &n = P.nodes[i];
    double v = 0.0;
    switch (n.op) {
      case RH_RIR_CONST: v = n.cval; break;
      case RH_RIR_INPUT: v = inputs[n.input]; break;
      case RH_RIR_ADD: v = val[n.a] + val[n.b]; break;
      case RH_RIR_SUB: v = val[n.a] - val[n.b]; break;
      case RH_RIR_MUL: v = val[n.a] * val[n.b]; break;
      case RH_RIR_DIV: v = val[n.a] / val[n.b]; break;
      case RH_RIR_POW: v = std::pow(val[n.a], val[n.b]); break;
      case RH_RIR_COMPARE: v = val[n.a] > val[n.b] ? 1.0 : (val[n.a] == val[n.b] ? 0.0 : -1.0); break;
      case RH_RIR_SEQ: v = val[n.b]; break;
      case RH_RIR_EXP: v = std::exp(val[n.a]); break;
      case RH_RIR_LOG: v = std::log(val[n.a]); break;
      case RH_RIR_ABS: v = std::fabs(val[n.a]); break;
      case RH_RIR_NOOP: v = val[n.a]; break;
      case RH_RIR_SIN: v = std::sin(val[n.a]); break;
      case RH_RIR_COS: v = std::cos(val[n.a]); break;
      case RH_RIR_TAN: v = std::tan(val[n.a]); break;
      case RH_RIR_ASIN: v = std::asin(val[n.a]); break;
      case RH_RIR_ACOS: v = std::acos(val[n.a]); break;
      case RH_RIR_ATAN: v = std::atan(val[n.a]); break;
      case RH_RIR_LOOKUP: {
        const double ix = val[n.a];
        const long long kk = (ix != ix ? 0LL : (long long)ix) - (long long)n.low;
        if (kk < 0 || kk >= (long long)n.table.size()) return false;
        v = val[n.table[(size_t)kk]];
        break;
      }
      default: return false;
    }
    val[i] = v;
  }
  return true;
}

}  // namespace

Program rederive_gradients(const Program &P, const std::vector<const double *> &cols, const int64_t *nrows, bool *changed) {
  if (changed) *changed = false;
  Deriver D(P);
  std::vector<std::vector<uint32_t>> fresh(P.targets.size());
  bool any = false;
  for (size_t t = 0; t < P.targets.size(); t++) {
    const Target &T = P.targets[t];
    if (!T.n_cols || nrows[t] <= 0) continue;
    fresh[t] = D.gradient(T.outputs[0]);
    // ---- verification on sample rows
    const int64_t nr = nrows[t];
    const int S = (int)std::min<int64_t>(24, nr);
    std::vector<uint32_t> roots(T.outputs.begin() + 1, T.outputs.end());
    roots.insert(roots.end(), fresh[t].begin(), fresh[t].end());
    bool ok = true;
    int checked = 0;
    uint64_t lcg = 0x9E3779B97F4A7C15ull + (uint64_t)t;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0 - 0.5; };
    for (int trial = 0; trial < 3 && ok; trial++) {
      std::vector<double> in(D.Q.n_inputs, 0.0);
      for (uint32_t p = 0; p < P.n_params; p++) in[p] = 1.2 * uni();
      for (int s = 0; s < S && ok; s++) {
        const int64_t r = (int64_t)s * nr / S;
        for (uint32_t j = 0; j < T.n_cols; j++) in[T.input_start + j] = cols[T.col0 + j][r];
        std::vector<double> val;
        if (!eval(D.Q, in, roots, val)) continue;   // a row the reference itself would refuse
        double scale = 1.0;
        for (uint32_t p = 0; p < P.n_params; p++) { const double a = val[T.outputs[1 + p]]; if (std::isfinite(a)) scale = std::max(scale, std::fabs(a)); }
        for (uint32_t p = 0; p < P.n_params && ok; p++) {
          const double a = val[T.outputs[1 + p]], b = val[fresh[t][p]];
          if (!std::isfinite(a) || !std::isfinite(b)) continue;
          if (std::fabs(a - b) > 1e-6 * scale) ok = false;
          checked++;
        }
      }
    }
    if (!ok || checked == 0) fresh[t].clear(); else any = true;
  }
  if (!any) return P;
  for (size_t t = 0; t < P.targets.size(); t++)
    if (!fresh[t].empty()) for (uint32_t p = 0; p < P.n_params; p++) D.Q.targets[t].outputs[1 + p] = fresh[t][p];
  if (changed) *changed = true;
  return D.Q;
}

}  // namespace rh
