// columns.cpp -- column canonicalisation: what the reference's front end hands its back end, brought back to its base data.
//
// rainier-compute pushes every data-only factor of an un-inlined target into a DERIVED column (`gradientColumns`,
// compute/Target.scala:27-31; Column.map / mapWith in compute/Real.scala): a 50-covariate logistic regression arrives with
// 255 columns -- the outcome y, -x_k, duplicates of y, y - 1, x_k * y (twice), x_k * (1 - y), and all-zero columns left over
// from `eq(index, k, g, 0)` gradients.  Streaming them literally costs 5x the HBM footprint and hides the model's structure
// (one design matrix, one adjoint) from the lowering.  At rh_model_create the data is in hand, so each column is compared
// with the columns before it and, when it is one of
//      a constant                   (every row the same value)
//      +-c_a                        (a copy / negated copy)
//      +-(c_a * c_b)                (an elementwise product of two earlier columns)
//      c_a + alpha, alpha - c_a, beta * c_a   (an affine image -- beta = -0.0 is how the signed zeros of x_k * (-0.0) survive
//                                              strict mode --; kept atomic -- wrapped in NOOP -- so that later algebra never
//                                              distributes over it: (y - 1) * S must stay a masked term, not S*y - S)
// its INPUT node is replaced by that expression over the BASE columns, which alone are uploaded.  Every relation is
// verified on ALL rows (candidates are pre-filtered on 32 sample rows).  In strict mode "equal" means bit-identical, so the
// recomputed value is the very double the reference stored (IEEE negation / multiplication / addition are deterministic);
// fast mode also accepts value equality (+0 == -0).  The pass also records, for every base column with at most 8 distinct
// values, that set (Program::col_domain): the emitter may verify a closed form on the values that actually occur.
//
// Second job, for programs recognised this way as the reference's lowering: Model.observe's INITIAL CHUNK (1-8 observations in a
// row target of its own, core/Model.scala:84-96) is unrolled -- its rows substituted as constants and summed in row order -- so
// that the model is left with one streamed target.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <thread>
#include <tuple>

#include "../../include/rainier_hip_rir.h"
#include "rir.hpp"

namespace rh {
namespace {

struct CExpr {  // expression of one original column over base columns
  enum Kind { BASE, ALIAS, CONST, NEG, MUL, ADDC, SUBC, MULC } kind = BASE;
  int a = -1, b = -1;   // operand original-column indices (local to the target)
  double c = 0.0;
};

inline uint64_t bits(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b; }

struct Same {
  bool fast;
  bool operator()(double x, double y) const { return fast ? x == y : bits(x) == bits(y); }
};

}  // namespace

bool canonicalize_columns(Program &P, const double *const *columns, const int64_t *nrows, bool fast, std::vector<uint32_t> &kept,
                          std::string &err, bool allow_unroll) {
  (void)err;
  kept.clear();
  P.col_domain.clear();
  const Same same{fast};
  bool changed = false;
  std::vector<std::vector<CExpr>> exprs(P.targets.size());
  for (size_t t = 0; t < P.targets.size(); t++) {
    const Target &T = P.targets[t];
    if (!T.n_cols) continue;
    const int64_t nr = nrows[t];
    const int nc = (int)T.n_cols;
    std::vector<const double *> col((size_t)nc);
    for (int j = 0; j < nc; j++) col[(size_t)j] = columns[T.col0 + (uint32_t)j];
    auto &ex = exprs[t];
    ex.assign((size_t)nc, CExpr());
    // sample rows for the candidate filter
    std::vector<int64_t> sr;
    if (nr > 0) {
      const int S = (int)std::min<int64_t>(32, nr);
      for (int i = 0; i < S; i++) sr.push_back((int64_t)i * nr / S);
      sr.push_back(nr - 1);
    }
    auto verify = [&](int j, auto &&f) {  // every row: col[j][r] == f(r); the sample rows first, the full scan on several threads
      const double *c = col[(size_t)j];
      for (int64_t r : sr) if (!same(c[r], f(r))) return false;
      const int nth = nr >= 400000 ? (int)std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), nr / 100000) : 1;
      if (nth <= 1) {
        for (int64_t r = 0; r < nr; r++) if (!same(c[r], f(r))) return false;
        return true;
      }
      std::vector<char> ok((size_t)nth, 1);
      std::vector<std::thread> th;
      for (int w = 0; w < nth; w++)
        th.emplace_back([&, w] {
          const int64_t r0 = nr * w / nth, r1 = nr * (w + 1) / nth;
          for (int64_t r = r0; r < r1; r++) if (!same(c[r], f(r))) { ok[(size_t)w] = 0; return; }
        });
      for (auto &t : th) t.join();
      for (char o : ok) if (!o) return false;
      return true;
    };
    // on a handful of rows every column is an affine image of every other: a small target (Model.observe's initial chunk, when it
    // is not unrolled) is only searched for copies, negations and products -- the relations its big sibling's slots have too
    const bool small = nr < 16;
    const bool search = nr >= 16 || (nr >= 3 && !allow_unroll);
    // constant columns first (a NaN column is left alone: RIR has no NaN constants); the search below skips them as operands
    std::vector<char> is_const((size_t)nc, 0);
    for (int j = 0; j < nc && search && !small; j++) {
      const double *c = col[(size_t)j];
      if (c[0] == c[0] && verify(j, [&](int64_t) { return c[0]; })) is_const[(size_t)j] = 1;
    }
    // every column against the columns before it: independent of each other, so models with thousands of columns (a mask per table
    // entry and slot) are searched on several threads -- unless the row scans themselves are threaded (verify, big targets)
    auto process = [&](int j) {
      const double *c = col[(size_t)j];
      CExpr e;
      bool found = false;
      if (is_const[(size_t)j]) { e.kind = CExpr::CONST; e.c = c[0]; found = true; }
      for (int a = 0; a < j && !found; a++) {
        const double *ca = col[(size_t)a];
        if (verify(j, [&](int64_t r) { return ca[r]; })) { e.kind = CExpr::ALIAS; e.a = a; found = true; break; }
        if (verify(j, [&](int64_t r) { return -ca[r]; })) { e.kind = CExpr::NEG; e.a = a; found = true; break; }
      }
      for (int a = 0; a < j && !found && !small; a++) {
        if (is_const[(size_t)a]) continue;
        const double *ca = col[(size_t)a];
        {
          const double al = c[0] - ca[0];
          if (al == al && std::isfinite(al) && al != 0.0 && verify(j, [&](int64_t r) { return ca[r] + al; })) { e.kind = CExpr::ADDC; e.a = a; e.c = al; found = true; break; }
        }
        {
          const double al = c[0] + ca[0];
          if (al == al && std::isfinite(al) && verify(j, [&](int64_t r) { return al - ca[r]; })) { e.kind = CExpr::SUBC; e.a = a; e.c = al; found = true; break; }
        }
        for (int64_t r : sr)
          if (ca[r] != 0.0 && std::isfinite(ca[r])) {
            const double be = c[r] / ca[r];
            if (be == be && std::isfinite(be) && be != 1.0 && be != -1.0 && verify(j, [&](int64_t q) { return be * ca[q]; })) { e.kind = CExpr::MULC; e.a = a; e.c = be; found = true; }
            break;
          }
      }
      // (quadratic in the number of earlier columns: bounded, so that a model with thousands of columns is not held up here)
      for (int a = 0; a < j && !found && (int64_t)j * j <= 4000000; a++) {
        if (is_const[(size_t)a]) continue;
        const double *ca = col[(size_t)a];
        for (int b = a; b < j && !found; b++) {
          if (is_const[(size_t)b]) continue;
          const double *cb = col[(size_t)b];
          if (verify(j, [&](int64_t r) { return ca[r] * cb[r]; })) { e.kind = CExpr::MUL; e.a = a; e.b = b; found = true; }
          else if (verify(j, [&](int64_t r) { return -(ca[r] * cb[r]); })) { e.kind = CExpr::MUL; e.a = a; e.b = b; e.c = -1.0; found = true; }
        }
      }
      if (found) ex[(size_t)j] = e;
      return found;
    };
    if (search) {
      const int nth = (nr < 400000 && nc >= 64) ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
      if (nth <= 1) {
        for (int j = 0; j < nc; j++) if (process(j)) changed = true;
      } else {
        std::atomic<int> next{0};
        std::atomic<bool> any{false};
        std::vector<std::thread> th;
        for (int w = 0; w < nth; w++)
          th.emplace_back([&] { for (int j = next++; j < nc; j = next++) if (process(j)) any = true; });
        for (auto &t : th) t.join();
        if (any) changed = true;
      }
    }
  }
  // Model.observe's initial chunk (core/Model.scala:84-96): a row target with at most 8 rows next to a big one.  Its rows are
  // substituted as constants and summed in row order (the order the reference's row loop adds them): a data-free target, so that
  // the model keeps ONE streamed target for the row kernels.
  std::vector<char> unroll(P.targets.size(), 0);
  {
    // only for programs that are recognisably the reference's lowering (derived columns were found) and not in gather mode,
    // whose data-free targets may not read the parameter table
    bool big = false;
    for (size_t t = 0; t < P.targets.size(); t++) big = big || (P.targets[t].n_cols && nrows[t] >= 16);
    big = big && changed && allow_unroll;
    for (size_t t = 0; t < P.targets.size() && big; t++)
      if (P.targets[t].n_cols && nrows[t] >= 1 && nrows[t] <= 8) { unroll[t] = 1; changed = true; }
  }
  // new column numbering: base columns only, original order
  std::vector<Target> nt = P.targets;
  uint32_t in = P.n_params, colc = 0;
  std::vector<std::vector<int>> base_new(P.targets.size());
  for (size_t t = 0; t < P.targets.size(); t++) {
    const Target &T = P.targets[t];
    nt[t].input_start = in; nt[t].col0 = colc;
    base_new[t].assign(T.n_cols, -1);
    uint32_t nb = 0;
    for (uint32_t j = 0; j < T.n_cols; j++) {
      const bool base = !unroll[t] && (exprs[t].empty() || exprs[t][j].kind == CExpr::BASE);
      if (!base) continue;
      base_new[t][j] = (int)nb++;
      kept.push_back(T.col0 + j);
      // the distinct values of the column, if few
      std::vector<double> dom;
      const double *c = columns[T.col0 + j];
      bool small = nrows[t] > 0;
      for (int64_t r = 0; r < nrows[t] && small; r++) {
        const double v = c[r];
        if (v != v) { small = false; break; }
        bool seen = false;
        for (double d : dom) if (d == v) { seen = true; break; }
        if (!seen) { if (dom.size() == 8) small = false; else dom.push_back(v); }
      }
      if (!small) dom.clear();
      std::sort(dom.begin(), dom.end());
      P.col_domain.push_back(dom);
    }
    // a row target keeps at least one column (its row count hangs on it): all-derived cannot happen (column 0 is never derived
    // unless constant) -- keep column 0 as a base then
    if (T.n_cols && nb == 0 && !unroll[t]) {
      base_new[t][0] = 0; nb = 1; kept.push_back(T.col0); P.col_domain.push_back({});
      exprs[t][0] = CExpr();
    }
    nt[t].n_cols = nb;
    in += nb; colc += nb;
  }
  if (!changed) { kept.clear(); for (uint32_t c = 0; c < P.n_cols_total; c++) kept.push_back(c); return false; }

  // rebuild the node list: INPUT nodes of derived columns become expressions (hash-consed), everything else is renumbered
  Program Q;
  Q.n_params = P.n_params; Q.kind = P.kind; Q.n_inputs = in; Q.n_cols_total = colc; Q.col_domain = P.col_domain;
  Q.targets = nt;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;
  std::map<uint64_t, uint32_t> consts;
  std::map<uint32_t, uint32_t> inputs;
  auto push = [&](const Node &n) { Q.nodes.push_back(n); return (uint32_t)Q.nodes.size() - 1; };
  auto constant = [&](double v) {
    auto it = consts.find(bits(v));
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    return consts[bits(v)] = push(n);
  };
  auto input = [&](uint32_t idx, uint32_t dep) {
    auto it = inputs.find(idx);
    if (it != inputs.end()) return it->second;
    Node n; n.op = RH_RIR_INPUT; n.input = idx; n.dep = dep;
    return inputs[idx] = push(n);
  };
  auto op2 = [&](uint32_t op, uint32_t a, uint32_t b, uint32_t dep) {
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = dep;
    return cons[key] = push(n);
  };
  auto op1 = [&](uint32_t op, uint32_t a, uint32_t dep) {
    auto key = std::make_tuple(op, a, 0xffffffffu);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = dep;
    return cons[key] = push(n);
  };
  std::vector<std::map<int, uint32_t>> col_node(P.targets.size());
  std::function<uint32_t(size_t, int)> column = [&](size_t t, int j) -> uint32_t {
    auto it = col_node[t].find(j);
    if (it != col_node[t].end()) return it->second;
    const uint32_t dep = (uint32_t)(t + 1);
    const CExpr &e = exprs[t][(size_t)j];
    uint32_t id = 0;
    switch (e.kind) {
      case CExpr::BASE: id = input(nt[t].input_start + (uint32_t)base_new[t][(size_t)j], dep); break;
      case CExpr::ALIAS: id = column(t, e.a); break;
      case CExpr::CONST: id = constant(e.c); break;
      case CExpr::NEG: id = op2(RH_RIR_MUL, constant(-1.0), column(t, e.a), dep); break;
      case CExpr::MUL: {
        const uint32_t p = op2(RH_RIR_MUL, column(t, e.a), column(t, e.b), dep);
        id = e.c == -1.0 ? op2(RH_RIR_MUL, constant(-1.0), p, dep) : p;
        break;
      }
      case CExpr::ADDC: id = op1(RH_RIR_NOOP, op2(RH_RIR_ADD, column(t, e.a), constant(e.c), dep), dep); break;
      case CExpr::SUBC: id = op1(RH_RIR_NOOP, op2(RH_RIR_SUB, constant(e.c), column(t, e.a), dep), dep); break;
      case CExpr::MULC: id = op1(RH_RIR_NOOP, op2(RH_RIR_MUL, constant(e.c), column(t, e.a), dep), dep); break;
    }
    return col_node[t][j] = id;
  };
  std::vector<uint32_t> m(P.nodes.size(), 0);
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    if (n.dep && unroll[(size_t)n.dep - 1]) { m[i] = constant(0.0); continue; }   // rebuilt per row below
    switch (n.op) {
      case RH_RIR_CONST: m[i] = constant(n.cval); break;
      case RH_RIR_INPUT:
        if (n.input < P.n_params) m[i] = input(n.input, 0);
        else {
          const size_t t = (size_t)n.dep - 1;
          m[i] = column(t, (int)(n.input - P.targets[t].input_start));
        }
        break;
      case RH_RIR_LOOKUP: {
        Node q; q.op = RH_RIR_LOOKUP; q.a = m[n.a]; q.low = n.low;
        q.dep = Q.nodes[q.a].dep;
        for (uint32_t e : n.table) { q.table.push_back(m[e]); if (Q.nodes[m[e]].dep) q.dep = Q.nodes[m[e]].dep; }
        m[i] = push(q);
        break;
      }
      case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_POW: case RH_RIR_COMPARE: case RH_RIR_SEQ: {
        const uint32_t a = m[n.a], b = m[n.b];
        m[i] = op2(n.op, a, b, Q.nodes[a].dep ? Q.nodes[a].dep : Q.nodes[b].dep);
        break;
      }
      default: m[i] = op1(n.op, m[n.a], Q.nodes[m[n.a]].dep);
    }
  }
  for (size_t t = 0; t < Q.targets.size(); t++) if (!unroll[t]) for (uint32_t &o : Q.targets[t].outputs) o = m[o];
  // the unrolled targets: per row, the target's nodes again with the row's values as constants (exact folding of + - * /,
  // compares and constant-index lookups only: transcendental nodes stay for the device)
  auto cval = [&](uint32_t id, double &v) { if (Q.nodes[id].op != RH_RIR_CONST) return false; v = Q.nodes[id].cval; return true; };
  for (size_t t = 0; t < P.targets.size(); t++) {
    if (!unroll[t]) continue;
    const Target &T = P.targets[t];
    std::vector<char> live(P.nodes.size(), 0);
    for (uint32_t o : T.outputs) live[o] = 1;
    for (size_t i = P.nodes.size(); i-- > 0;) {
      if (!live[i]) continue;
      const Node &n = P.nodes[i];
      if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
      live[n.a] = 1;
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) live[e] = 1; }
      else if ((n.op >= RH_RIR_ADD && n.op <= RH_RIR_COMPARE) || n.op == RH_RIR_SEQ) live[n.b] = 1;
    }
    std::vector<uint32_t> acc(T.outputs.size(), 0);
    for (int64_t r = 0; r < nrows[t]; r++) {
      std::vector<uint32_t> mr(P.nodes.size(), 0);
      for (uint32_t i = 0; i < P.nodes.size(); i++) {
        if (!live[i]) continue;
        const Node &n = P.nodes[i];
        if (n.dep != t + 1) { mr[i] = m[i]; continue; }
        double x, y;
        switch (n.op) {
          case RH_RIR_INPUT: mr[i] = constant(columns[T.col0 + (n.input - T.input_start)][r]); break;
          case RH_RIR_LOOKUP: {
            const uint32_t ix = mr[n.a];
            if (cval(ix, x)) {
              const long long k = (x != x ? 0LL : (long long)x) - (long long)n.low;
              if (k >= 0 && k < (long long)n.table.size()) { mr[i] = mr[n.table[(size_t)k]]; break; }
            }
            Node q; q.op = RH_RIR_LOOKUP; q.a = ix; q.low = n.low;
            for (uint32_t e : n.table) q.table.push_back(mr[e]);
            mr[i] = push(q);
            break;
          }
          case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_COMPARE: {
            const uint32_t a = mr[n.a], b = mr[n.b];
            if (cval(a, x) && cval(b, y)) {
              const double v = n.op == RH_RIR_ADD ? x + y : n.op == RH_RIR_SUB ? x - y : n.op == RH_RIR_MUL ? x * y : n.op == RH_RIR_DIV ? x / y
                                                          : (x > y ? 1.0 : (x == y ? 0.0 : -1.0));
              if (v == v) { mr[i] = constant(v); break; }
            }
            mr[i] = op2(n.op, a, b, 0);
            break;
          }
          case RH_RIR_POW: case RH_RIR_SEQ: mr[i] = op2(n.op, mr[n.a], mr[n.b], 0); break;
          default: mr[i] = op1(n.op, mr[n.a], 0);
        }
      }
      for (size_t o = 0; o < T.outputs.size(); o++) acc[o] = r == 0 ? mr[T.outputs[o]] : op2(RH_RIR_ADD, acc[o], mr[T.outputs[o]], 0);
    }
    Q.targets[t].outputs = acc;
  }
  P = std::move(Q);
  return true;
}

}  // namespace rh
