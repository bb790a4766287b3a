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
// The replacement is VERIFIED at the parity contract's own bound: on up to 256 sample rows of the target x 8 random parameter
// vectors, every supplied gradient output must agree with the re-derived one, row by row and in the sum over the rows, to
// 1e-11 * (that output's sum of magnitudes over the sample), evaluated in extended precision; a point where only one side is
// finite rejects.  A program whose "gradient" is not the derivative of its value, or that cannot be evaluated on the host, keeps
// the outputs it came with.  Rounding changes only: fast mode.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
    for (uint32_t i = 0; i < N0; i++)   // (a parameter may be read through several INPUT nodes: the Translator does not share them)
      if (Q.nodes[i].op == RH_RIR_INPUT && Q.nodes[i].input < Q.n_params) {
        auto it = adj.find(i);
        if (it == adj.end()) continue;
        uint32_t &o = out[Q.nodes[i].input];
        o = (o == zero) ? it->second : op2(RH_RIR_ADD, o, it->second);
      }
    return out;
  }
};

// Host evaluation of the nodes `roots` need, for a BLOCK of rows at once (ascending ids are a topological order) and in extended
// precision (x87 long double: 64-bit significand) -- the two gradients are the same function written two ways, and what is to be
// told apart is a wrong derivative from a differently rounded one: with 11 more bits the rounding noise of either spelling sits
// three orders below the acceptance bound even where the reference's masked forms cancel (1 - p next to p -> 1).
// inputs[i * B + r] = input i at block row r.  ok[r] is cleared where a Lookup index leaves its table (a row the reference itself
// would refuse).  The live set is computed once per target (Evaluator).
typedef long double xreal;
struct Evaluator {
  const Program &P;
  std::vector<uint32_t> order;   // live nodes, ascending
  Evaluator(const Program &p, const std::vector<uint32_t> &roots) : P(p) {
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
    for (size_t i = 0; i < P.nodes.size(); i++) if (live[i]) order.push_back((uint32_t)i);
  }
  // val[node * B + r]; returns false on an opcode the interpreter does not know
  bool run(const std::vector<xreal> &inputs, int B, std::vector<xreal> &val, std::vector<char> &ok) const {
    val.resize(P.nodes.size() * (size_t)B);
    ok.assign((size_t)B, 1);
    for (uint32_t i : order) {
      const Node &n = P.nodes[i];
      xreal *v = &val[(size_t)i * B];
      const xreal *a = &val[(size_t)n.a * B], *b = &val[(size_t)n.b * B];
      switch (n.op) {
        case RH_RIR_CONST: for (int r = 0; r < B; r++) v[r] = (xreal)n.cval; break;
        case RH_RIR_INPUT: for (int r = 0; r < B; r++) v[r] = inputs[(size_t)n.input * B + r]; break;
        case RH_RIR_ADD: for (int r = 0; r < B; r++) v[r] = a[r] + b[r]; break;
        case RH_RIR_SUB: for (int r = 0; r < B; r++) v[r] = a[r] - b[r]; break;
        case RH_RIR_MUL: for (int r = 0; r < B; r++) v[r] = a[r] * b[r]; break;
        case RH_RIR_DIV: for (int r = 0; r < B; r++) v[r] = a[r] / b[r]; break;
        case RH_RIR_POW: for (int r = 0; r < B; r++) v[r] = std::pow(a[r], b[r]); break;
        case RH_RIR_COMPARE: for (int r = 0; r < B; r++) v[r] = a[r] > b[r] ? 1.0L : (a[r] == b[r] ? 0.0L : -1.0L); break;
        case RH_RIR_SEQ: for (int r = 0; r < B; r++) v[r] = b[r]; break;
        case RH_RIR_EXP: for (int r = 0; r < B; r++) v[r] = std::exp(a[r]); break;
        case RH_RIR_LOG: for (int r = 0; r < B; r++) v[r] = std::log(a[r]); break;
        case RH_RIR_ABS: for (int r = 0; r < B; r++) v[r] = std::fabs(a[r]); break;
        case RH_RIR_NOOP: for (int r = 0; r < B; r++) v[r] = a[r]; break;
        case RH_RIR_SIN: for (int r = 0; r < B; r++) v[r] = std::sin(a[r]); break;
        case RH_RIR_COS: for (int r = 0; r < B; r++) v[r] = std::cos(a[r]); break;
        case RH_RIR_TAN: for (int r = 0; r < B; r++) v[r] = std::tan(a[r]); break;
        case RH_RIR_ASIN: for (int r = 0; r < B; r++) v[r] = std::asin(a[r]); break;
        case RH_RIR_ACOS: for (int r = 0; r < B; r++) v[r] = std::acos(a[r]); break;
        case RH_RIR_ATAN: for (int r = 0; r < B; r++) v[r] = std::atan(a[r]); break;
        case RH_RIR_LOOKUP:
          for (int r = 0; r < B; r++) {
            const xreal ix = a[r];
            const long long kk = (ix != ix ? 0LL : (long long)ix) - (long long)n.low;
            if (kk < 0 || kk >= (long long)n.table.size()) { ok[(size_t)r] = 0; v[r] = 0.0L; continue; }
            v[r] = val[(size_t)n.table[(size_t)kk] * B + r];
          }
          break;
        default: return false;
      }
    }
    return true;
  }
};

}  // namespace

// exported for lift.cpp: d value / d theta_p for every parameter as new nodes of P (reverse mode, compute/Gradient.scala:60-160)
std::vector<uint32_t> derive_gradient(Program &P, uint32_t value) {
  Deriver D(P);
  std::vector<uint32_t> g = D.gradient(value);
  P.nodes.swap(D.Q.nodes);
  return g;
}
// exported for lift.cpp: the block interpreter above behind an opaque handle (the live set of `roots` is computed once)
BlockEvaluator::BlockEvaluator(const Program &P, const std::vector<uint32_t> &roots) : impl(new Evaluator(P, roots)) {}
BlockEvaluator::~BlockEvaluator() { delete static_cast<Evaluator *>(impl); }
bool BlockEvaluator::run(const std::vector<long double> &inputs, int B, std::vector<long double> &val, std::vector<char> &ok) const {
  return static_cast<const Evaluator *>(impl)->run(inputs, B, val, ok);
}

Program rederive_gradients(const Program &P, const std::vector<const double *> &cols, const int64_t *nrows, bool *changed) {
  if (changed) *changed = false;
  Deriver D(P);
  std::vector<std::vector<uint32_t>> fresh(P.targets.size());
  bool any = false;
  for (size_t t = 0; t < P.targets.size(); t++) {
    const Target &T = P.targets[t];
    if (!T.n_cols || nrows[t] <= 0) continue;
    fresh[t] = D.gradient(T.outputs[0]);
    // ---- verification: the contract is |sum_rows (supplied - rederived)| <= 1e-11 * sum_rows |term| PER OUTPUT, so that is
    // what is asked of the sample -- up to 256 rows spread over the target x 8 parameter vectors (four at the scale of a standard
    // normal start, two near the origin, two three times further out), output by output against that output's own sum of
    // magnitudes, row by row and in the sum.  A point where exactly one side is non-finite (0 * inf out of a DIV / POW rule), or
    // where the two sides are different infinities, rejects the rewrite; so does an interpreter failure.
    const int64_t nr = nrows[t];
    const int S = (int)std::min<int64_t>(256, nr);
    std::vector<uint32_t> roots(T.outputs.begin() + 1, T.outputs.end());
    roots.insert(roots.end(), fresh[t].begin(), fresh[t].end());
    const Evaluator EV(D.Q, roots);
    bool ok = true;
    int64_t checked = 0;
    uint64_t lcg = 0x9E3779B97F4A7C15ull + (uint64_t)t;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0 - 0.5; };
    static const double kScale[8] = {1.2, 1.2, 0.2, 3.6, 1.2, 1.2, 0.2, 3.6};
    constexpr int B = 32;
    std::vector<xreal> in((size_t)D.Q.n_inputs * B, 0.0L), val;
    std::vector<char> rok;
    std::vector<xreal> sa(P.n_params), sb(P.n_params), mag(P.n_params), worst(P.n_params);
    for (int trial = 0; trial < 8 && ok; trial++) {
      std::vector<double> th(P.n_params);
      for (uint32_t p = 0; p < P.n_params; p++) th[p] = kScale[trial] * uni();
      std::fill(sa.begin(), sa.end(), 0.0L); std::fill(sb.begin(), sb.end(), 0.0L);
      std::fill(mag.begin(), mag.end(), 0.0L); std::fill(worst.begin(), worst.end(), 0.0L);
      for (int s0 = 0; s0 < S && ok; s0 += B) {
        const int nb = std::min(B, S - s0);
        for (int r = 0; r < B; r++) {
          const int64_t row = (int64_t)std::min(s0 + r, S - 1) * nr / S;
          for (uint32_t p = 0; p < P.n_params; p++) in[(size_t)p * B + r] = (xreal)th[p];
          for (uint32_t j = 0; j < T.n_cols; j++) in[(size_t)(T.input_start + j) * B + r] = (xreal)cols[T.col0 + j][row];
        }
        if (!EV.run(in, B, val, rok)) { ok = false; break; }
        for (uint32_t p = 0; p < P.n_params && ok; p++) {
          const xreal *a = &val[(size_t)T.outputs[1 + p] * B], *b = &val[(size_t)fresh[t][p] * B];
          for (int r = 0; r < nb; r++) {
            if (!rok[(size_t)r]) continue;
            const bool fa = std::isfinite(a[r]), fb = std::isfinite(b[r]);
            if (!fa || !fb) {
              if (fa != fb || (std::isinf(a[r]) && std::isinf(b[r]) && a[r] != b[r]) || (std::isnan(a[r]) != std::isnan(b[r]))) ok = false;
              continue;   // the same NaN / infinity on both sides
            }
            sa[p] += a[r]; sb[p] += b[r];
            mag[p] += std::max(std::fabs(a[r]), std::fabs(b[r]));
            worst[p] = std::max(worst[p], std::fabs(a[r] - b[r]));
            checked++;
          }
        }
      }
      for (uint32_t p = 0; p < P.n_params && ok; p++) {
        const xreal bound = 1e-11L * mag[p];
        if (worst[p] > bound || std::fabs(sa[p] - sb[p]) > bound) {
          ok = false;
          if (rh::knob("RH_REDERIVE_WHY"))
            std::fprintf(stderr, "rederive: target %zu output %u trial %d: worst row difference %.3Lg, sum difference %.3Lg, bound %.3Lg\n", t, p, trial,
                         worst[p], std::fabs(sa[p] - sb[p]), bound);
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
