// rollstrict.cpp -- strict builds: Model.observe's 8 slots rolled back into rows WITHOUT re-association.
//
// Fast builds roll the second target Model.observe writes (core/Model.scala:71-132: one expression over the columns of 8
// observations per "row") in polynomial normal form (refactor.cpp), which re-associates products and sums.  Strict builds keep
// the reference's arithmetic operation for operation, so here the expression is only cut along its top-level additions (the
// Translator's left fold of a Line, compute/Translator.scala:91-125): every output is  sum_i term_i ; a term belongs to the slot
// whose columns it reads, or to nobody (parameter-only: the Line algebra has merged the 8 copies of such a term into ONE with
// 8 x the coefficient).  If, after renaming slot s's columns onto slot 1's, every term of slot s IS a term of slot 1 -- the very
// same expression tree, checked by hash-consing the renamed tree into the node table -- the target becomes
//        out = sum of slot 1's terms, in their original order  +  sum of (shared term) * (1 / S)
// over S x the rows.  1 / S is exact for S = 8 and the shared terms are the only arithmetic that changes (by a power of two);
// the order in which the rows are added up changes as it does in any parallel reduction.  Anything that does not match keeps the
// 8-slot expression.
//
// With RH_INDEX_MASKS=1 (rir.hpp index_masks_on: the strict lowering of hierarchical models, off until it has run on the device) the
// roll also reads through what keeps the reference's 8 slots from being ONE expression although they are one function:
//   * SEQ(definition, use) -- the Translator's sequencing of a shared sub-expression, placed inside whichever slot used it first --
//     is read as `use` (what the emitter evaluates anyway);
//   * a term (sum over the slots) * (parameter-only factors) -- the chain rule through a transformed parameter -- has the factors
//     distributed over the addends of different column groups: ONE extra rounding per addend;  (x + x) * f is written x f + x f
//     (the same double: a doubling is exact);
//   * if the slots still differ, once more with the addends of every data-dependent sum in ONE canonical order (SumNormaliser: the
//     reference's Line is a map, its addends are folded in a different order in every slot): the ORDER of the additions inside a
//     row's expression changes, nothing is merged or cancelled;  a * b and b * a are one node (the same double);
//   * a loose column times parameter-only factors (a * (y_1 + ... + y_8) arrives as one pre-summed column) is kept on the first
//     slot's rows and zero-padded for the others, like a bare loose column.
// These are the only places where the per-row arithmetic of a rolled strict target is not the reference's operation for operation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <tuple>

#include "../../include/rainier_hip_rir.h"
#include "rir.hpp"

namespace rh {
namespace {

bool binary_op(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }
uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xBF58476D1CE4E5B9ull; }

struct Roller {
  Program &P;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;                       // (op, a, b | ~0)
  std::map<std::tuple<uint32_t, int32_t, std::vector<uint32_t>>, uint32_t> lookups;        // (index, low, table)
  std::map<uint64_t, uint32_t> consts;
  std::map<uint32_t, uint32_t> inputs;
  std::vector<std::vector<uint32_t>> cols_memo;
  std::vector<char> cols_done;
  std::vector<uint64_t> hmemo;
  std::vector<char> hdone, hasp;

  explicit Roller(Program &p) : P(p) {
    for (uint32_t i = 0; i < P.nodes.size(); i++) index(i);
    hasp.assign(P.nodes.size(), 0);
    for (uint32_t i = 0; i < P.nodes.size(); i++) {
      const Node &n = P.nodes[i];
      if (n.op == RH_RIR_INPUT) hasp[i] = n.input < P.n_params;
      else if (n.op == RH_RIR_CONST) hasp[i] = 0;
      else if (n.op == RH_RIR_LOOKUP) { hasp[i] = hasp[n.a]; for (uint32_t e : n.table) hasp[i] = hasp[i] || hasp[e]; }
      else hasp[i] = hasp[n.a] || (binary_op(n.op) && hasp[n.b]);
    }
  }
  void index(uint32_t i) {
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST) { uint64_t b; std::memcpy(&b, &n.cval, 8); consts.emplace(b, i); }
    else if (n.op == RH_RIR_INPUT) inputs.emplace(n.input, i);
    else if (n.op == RH_RIR_LOOKUP) lookups.emplace(std::make_tuple(n.a, n.low, n.table), i);
    else cons.emplace(std::make_tuple(n.op, n.a, binary_op(n.op) ? n.b : 0xffffffffu), i);
  }
  uint32_t push(const Node &n) {
    P.nodes.push_back(n);
    const uint32_t i = (uint32_t)P.nodes.size() - 1;
    index(i);
    if (!hasp.empty()) {      // (nodes made after the constructor: renamed trees, distributed products)
      char h = 0;
      if (n.op == RH_RIR_INPUT) h = n.input < P.n_params;
      else if (n.op == RH_RIR_LOOKUP) { h = hasp[n.a]; for (uint32_t e : n.table) h = h || hasp[e]; }
      else if (n.op != RH_RIR_CONST) h = hasp[n.a] || (binary_op(n.op) && hasp[n.b]);
      hasp.resize(P.nodes.size(), 0);
      hasp[i] = h;
    }
    return i;
  }
  uint32_t constant(double v) {
    uint64_t b; std::memcpy(&b, &v, 8);
    auto it = consts.find(b);
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    return push(n);
  }
  uint32_t op2(uint32_t op, uint32_t a, uint32_t b) {
    auto it = cons.find(std::make_tuple(op, a, b));
    if (it != cons.end()) return it->second;
    if ((op == RH_RIR_MUL || op == RH_RIR_ADD) && commutative_mul) {      // a * b and b * a (a + b and b + a) are the same double
      it = cons.find(std::make_tuple(op, b, a));
      if (it != cons.end()) return it->second;
    }
    Node n; n.op = op; n.a = a; n.b = b; n.dep = P.nodes[a].dep ? P.nodes[a].dep : P.nodes[b].dep;
    return push(n);
  }
  bool commutative_mul = index_masks_on();
  uint32_t op1(uint32_t op, uint32_t a) {
    auto it = cons.find(std::make_tuple(op, a, 0xffffffffu));
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = P.nodes[a].dep;
    return push(n);
  }
  const std::vector<uint32_t> &cols(uint32_t id) {   // the column inputs a node reaches (sorted); operands have smaller ids
    if (cols_done.size() < P.nodes.size()) { cols_done.resize(P.nodes.size(), 0); cols_memo.resize(P.nodes.size()); }
    if (cols_done[id]) return cols_memo[id];
    std::vector<uint32_t> stack{id};
    while (!stack.empty()) {   // iterative post-order over the not-yet-done part
      const uint32_t x = stack.back();
      if (cols_done[x]) { stack.pop_back(); continue; }
      const Node &n = P.nodes[x];
      std::vector<uint32_t> ops;
      if (n.op != RH_RIR_CONST && n.op != RH_RIR_INPUT) { ops.push_back(n.a); if (n.op == RH_RIR_LOOKUP) ops.insert(ops.end(), n.table.begin(), n.table.end()); else if (binary_op(n.op)) ops.push_back(n.b); }
      bool ready = true;
      for (uint32_t o : ops) if (!cols_done[o]) { stack.push_back(o); ready = false; }
      if (!ready) continue;
      std::vector<uint32_t> r;
      if (n.op == RH_RIR_INPUT && n.input >= P.n_params) r.push_back(n.input);
      for (uint32_t o : ops) r.insert(r.end(), cols_memo[o].begin(), cols_memo[o].end());
      std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end());
      cols_memo[x] = r; cols_done[x] = 1;
      stack.pop_back();
    }
    return cols_memo[id];
  }
  uint64_t hash(uint32_t id) {   // column-blind structure
    if (hdone.size() < P.nodes.size()) { hdone.resize(P.nodes.size(), 0); hmemo.resize(P.nodes.size(), 0); }
    if (hdone[id]) return hmemo[id];
    for (uint32_t x = 0; x <= id; x++) {   // operands first: ascending ids
      if (hdone[x]) continue;
      const Node &n = P.nodes[x];
      uint64_t h = mix(0x51, n.op);
      if (n.op == RH_RIR_CONST) { uint64_t b; std::memcpy(&b, &n.cval, 8); h = mix(h, b); }
      else if (n.op == RH_RIR_INPUT) h = n.input >= P.n_params ? mix(h, 0xC01) : mix(mix(h, 0x9A7), n.input);
      else {
        h = mix(h, hmemo[n.a]);
        if (n.op == RH_RIR_LOOKUP) { h = mix(h, (uint64_t)(int64_t)n.low); for (uint32_t e : n.table) h = mix(h, hmemo[e]); }
        else if ((n.op == RH_RIR_MUL || n.op == RH_RIR_ADD) && commutative_mul) h = mix(mix(0x51, n.op), hmemo[n.a] + hmemo[n.b]);   // blind to the operand order
        else if (binary_op(n.op)) h = mix(h, hmemo[n.b]);
      }
      hmemo[x] = h; hdone[x] = 1;
    }
    return hmemo[id];
  }
  // the tree of `id` with columns renamed, hash-consed into the node table (an identical tree comes back as its existing id)
  uint32_t rename(uint32_t id, const std::map<uint32_t, uint32_t> &cmap, std::map<uint32_t, uint32_t> &memo) {
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    const Node n = P.nodes[id];
    uint32_t r;
    if (n.dep == 0) r = id;
    else if (n.op == RH_RIR_INPUT) {
      auto c = cmap.find(n.input);
      if (c == cmap.end()) r = id;
      else { auto ii = inputs.find(c->second); if (ii != inputs.end()) r = ii->second; else { Node q = n; q.input = c->second; r = push(q); } }
    } else if (n.op == RH_RIR_LOOKUP) {
      Node q = n; q.a = rename(n.a, cmap, memo);
      for (uint32_t &e : q.table) e = rename(e, cmap, memo);
      auto li = lookups.find(std::make_tuple(q.a, q.low, q.table));
      r = li != lookups.end() ? li->second : push(q);
    } else if (binary_op(n.op)) { const uint32_t a = rename(n.a, cmap, memo), b = rename(n.b, cmap, memo); r = op2(n.op, a, b); }
    else r = op1(n.op, rename(n.a, cmap, memo));
    return memo[id] = r;
  }
  void walk(uint32_t id, std::vector<uint32_t> &order, std::map<uint32_t, char> &seen_col, std::map<uint32_t, char> &seen_node) {
    if (P.nodes[id].dep == 0 || seen_node.count(id)) return;
    seen_node[id] = 1;
    const Node &n = P.nodes[id];
    if (n.op == RH_RIR_INPUT) { if (n.input >= P.n_params && !seen_col.count(n.input)) { seen_col[n.input] = 1; order.push_back(n.input); } return; }
    walk(n.a, order, seen_col, seen_node);
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) walk(e, order, seen_col, seen_node); }
    else if (binary_op(n.op)) walk(n.b, order, seen_col, seen_node);
  }
};

struct Term { uint32_t node; bool neg; };
void flatten(const Program &P, uint32_t id, bool neg, std::vector<Term> &out, bool seq_through) {
  std::vector<std::pair<uint32_t, bool>> stack{{id, neg}};
  std::vector<Term> rev;
  while (!stack.empty()) {      // left operand first -> original left-to-right order
    auto [x, ng] = stack.back(); stack.pop_back();
    const Node &n = P.nodes[x];
    if (n.dep != 0 && (n.op == RH_RIR_ADD || n.op == RH_RIR_SUB)) { stack.push_back({n.b, n.op == RH_RIR_SUB ? !ng : ng}); stack.push_back({n.a, ng}); }
    else if (n.dep != 0 && n.op == RH_RIR_SEQ && seq_through) stack.push_back({n.b, ng});   // SEQ(definition, use): the value is `use`
    else rev.push_back({x, ng});
  }
  out.insert(out.end(), rev.begin(), rev.end());
}

// Every sum of data-dependent addends re-written with its addends in ONE canonical order (sorted by a column-blind, order-blind
// structural hash; products likewise, which changes no bit).  The reference's Line is a map from terms to coefficients whose
// iteration order -- hence the order in which the Translator folds the addends -- differs from slot to slot of Model.observe's
// split; with the addends in one order the 8 slots are the same expression again.  The order of additions inside a row's
// expression changes (rounding only); nothing is distributed, merged or cancelled.
struct SumNormaliser {
  const Program &P;
  Program Q;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;
  std::map<std::tuple<uint32_t, int32_t, std::vector<uint32_t>>, uint32_t> lookups;
  std::map<uint64_t, uint32_t> consts;
  std::map<uint32_t, uint32_t> inputs;
  std::vector<uint32_t> m;          // P node -> Q node
  std::vector<char> done;
  std::vector<uint64_t> qh;         // Q node -> hash
  explicit SumNormaliser(const Program &p) : P(p) {
    Q.n_params = P.n_params; Q.n_inputs = P.n_inputs; Q.n_cols_total = P.n_cols_total; Q.kind = P.kind; Q.targets = P.targets; Q.col_domain = P.col_domain;
    m.assign(P.nodes.size(), 0); done.assign(P.nodes.size(), 0);
  }
  uint32_t push(const Node &n, uint64_t h) { Q.nodes.push_back(n); qh.push_back(h); return (uint32_t)Q.nodes.size() - 1; }
  uint32_t constant(double v) {
    uint64_t b; std::memcpy(&b, &v, 8);
    auto it = consts.find(b);
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    return consts[b] = push(n, mix(mix(0x51, RH_RIR_CONST), b));
  }
  uint32_t op2(uint32_t op, uint32_t a, uint32_t b) {
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = Q.nodes[a].dep ? Q.nodes[a].dep : Q.nodes[b].dep;
    uint64_t h;
    if (op == RH_RIR_MUL) h = mix(mix(0x51, op), qh[a] + qh[b]);
    else h = mix(mix(mix(0x51, op), qh[a]), qh[b]);       // (sums are hashed by their flattened addends in norm())
    return cons[key] = push(n, h);
  }
  uint32_t op1(uint32_t op, uint32_t a) {
    auto key = std::make_tuple(op, a, 0xffffffffu);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = Q.nodes[a].dep;
    return cons[key] = push(n, mix(mix(0x51, op), qh[a]));
  }
  void addends(uint32_t id, bool neg, std::vector<std::pair<uint32_t, bool>> &out) {   // P ids, left to right
    std::vector<std::pair<uint32_t, bool>> stack{{id, neg}};
    while (!stack.empty()) {
      auto [x, ng] = stack.back(); stack.pop_back();
      const Node &n = P.nodes[x];
      if (n.dep != 0 && (n.op == RH_RIR_ADD || n.op == RH_RIR_SUB)) { stack.push_back({n.b, n.op == RH_RIR_SUB ? !ng : ng}); stack.push_back({n.a, ng}); }
      else out.push_back({x, ng});
    }
  }
  uint32_t norm(uint32_t root) {
    std::vector<uint32_t> stack{root};
    std::vector<uint32_t> ops;
    while (!stack.empty()) {
      const uint32_t x = stack.back();
      if (done[x]) { stack.pop_back(); continue; }
      const Node &n = P.nodes[x];
      ops.clear();
      std::vector<std::pair<uint32_t, bool>> ad;
      const bool sum = n.dep != 0 && (n.op == RH_RIR_ADD || n.op == RH_RIR_SUB);
      if (sum) { addends(x, false, ad); for (auto &a : ad) ops.push_back(a.first); }
      else if (n.op != RH_RIR_CONST && n.op != RH_RIR_INPUT) { ops.push_back(n.a); if (n.op == RH_RIR_LOOKUP) ops.insert(ops.end(), n.table.begin(), n.table.end()); else if (binary_op(n.op)) ops.push_back(n.b); }
      bool ready = true;
      for (uint32_t o : ops) if (!done[o]) { stack.push_back(o); ready = false; }
      if (!ready) continue;
      uint32_t r;
      if (n.op == RH_RIR_CONST) r = constant(n.cval);
      else if (n.op == RH_RIR_INPUT) {
        auto it = inputs.find(n.input);
        if (it != inputs.end()) r = it->second;
        else { Node q = n; r = inputs[n.input] = push(q, n.input >= P.n_params ? mix(mix(0x51, RH_RIR_INPUT), 0xC01) : mix(mix(mix(0x51, RH_RIR_INPUT), 0x9A7), n.input)); }
      } else if (sum) {
        std::vector<std::tuple<int, uint64_t, size_t>> key;     // positive addends first, then by hash, then by original position
        for (size_t i = 0; i < ad.size(); i++) key.push_back(std::make_tuple(ad[i].second ? 1 : 0, qh[m[ad[i].first]], i));
        std::sort(key.begin(), key.end());
        uint32_t acc = 0;
        uint64_t hs = 0;
        for (size_t i = 0; i < key.size(); i++) {
          const auto &a = ad[std::get<2>(key[i])];
          const uint32_t q = m[a.first];
          hs += mix(qh[q], a.second ? 1 : 0);
          if (i == 0) acc = a.second ? op2(RH_RIR_SUB, constant(0.0), q) : q;
          else acc = op2(a.second ? RH_RIR_SUB : RH_RIR_ADD, acc, q);
        }
        r = acc;
        qh[r] = mix(mix(0x51, RH_RIR_ADD), hs);               // order-blind: the same for every order the addends came in
      } else if (n.op == RH_RIR_LOOKUP) {
        Node q; q.op = RH_RIR_LOOKUP; q.a = m[n.a]; q.low = n.low; q.dep = Q.nodes[q.a].dep;
        uint64_t h = mix(mix(mix(0x51, RH_RIR_LOOKUP), qh[q.a]), (uint64_t)(int64_t)n.low);
        for (uint32_t e : n.table) { q.table.push_back(m[e]); if (Q.nodes[m[e]].dep) q.dep = Q.nodes[m[e]].dep; h = mix(h, qh[m[e]]); }
        auto key = std::make_tuple(q.a, q.low, q.table);
        auto it = lookups.find(key);
        r = it != lookups.end() ? it->second : (lookups[key] = push(q, h));
      } else if (n.op == RH_RIR_MUL) {
        uint32_t a = m[n.a], b = m[n.b];
        if (n.dep != 0 && std::make_pair(qh[b], b) < std::make_pair(qh[a], a)) std::swap(a, b);   // a * b == b * a, bit for bit
        r = op2(RH_RIR_MUL, a, b);
      } else if (binary_op(n.op)) r = op2(n.op, m[n.a], m[n.b]);
      else r = op1(n.op, m[n.a]);
      m[x] = r; done[x] = 1;
      stack.pop_back();
    }
    return m[root];
  }
};

bool roll_strict_impl(Program &P, std::vector<std::vector<uint32_t>> &parts, int &not_rolled);

}  // namespace

bool roll_strict(Program &P, std::vector<std::vector<uint32_t>> &parts) {
  Program A = P;
  std::vector<std::vector<uint32_t>> pa;
  int left_a = 0;
  const bool ok_a = roll_strict_impl(A, pa, left_a);
  if (left_a > 0 && index_masks_on()) {   // a target kept its slots: once more with every sum's addends in one canonical order
    SumNormaliser N(P);
    for (Target &T : N.Q.targets) for (uint32_t &o : T.outputs) o = N.norm(o);
    std::vector<std::vector<uint32_t>> pb;
    int left_b = 0;
    Program B = N.Q;
    if (roll_strict_impl(B, pb, left_b) && left_b < left_a) { P = std::move(B); parts = std::move(pb); return true; }
  }
  if (ok_a) { P = std::move(A); parts = std::move(pa); }
  else { parts.clear(); for (uint32_t c = 0; c < P.n_cols_total; c++) parts.push_back({c}); }
  return ok_a;
}

namespace {
bool roll_strict_impl(Program &P, std::vector<std::vector<uint32_t>> &parts, int &not_rolled) {
  not_rolled = 0;
  parts.clear();
  for (uint32_t c = 0; c < P.n_cols_total; c++) parts.push_back({c});
  if (index_masks_on()) {
    // SEQ(definition, use) -- the Translator's sequencing of a shared sub-expression's first use -- has the value of `use` and sits
    // wherever that first use happened to be (inside ONE slot's term): read through it, so that the slots are the same expression
    std::vector<uint32_t> fwd(P.nodes.size());
    for (uint32_t i = 0; i < P.nodes.size(); i++) fwd[i] = P.nodes[i].op == RH_RIR_SEQ ? fwd[P.nodes[i].b] : i;
    for (Node &n : P.nodes) {
      if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
      n.a = fwd[n.a];
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : n.table) e = fwd[e]; }
      else if (binary_op(n.op)) n.b = fwd[n.b];
    }
    for (Target &T : P.targets) for (uint32_t &o : T.outputs) o = fwd[o];
  }
  Roller R(P);
  bool any = false;
  const size_t NT = P.targets.size();
  std::vector<std::vector<uint32_t>> keep(NT);
  std::vector<std::vector<std::vector<uint32_t>>> src(NT);
  for (size_t t = 0; t < NT; t++) for (uint32_t j = 0; j < P.targets[t].n_cols; j++) { keep[t].push_back(P.targets[t].input_start + j); src[t].push_back({P.targets[t].input_start + j}); }
  for (size_t t = 0; t < NT; t++) {
    if (!P.targets[t].n_cols) continue;
    const size_t no = P.targets[t].outputs.size();
    std::vector<std::vector<Term>> terms(no);
    for (size_t o = 0; o < no; o++) flatten(P, P.targets[t].outputs[o], false, terms[o], index_masks_on());
    // components of the columns: joined when one term reads both
    std::map<uint32_t, uint32_t> parent;
    std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { auto it = parent.find(x); if (it == parent.end()) { parent[x] = x; return x; } if (it->second == x) return x; const uint32_t r = find(it->second); parent[x] = r; return r; };
    if (index_masks_on()) {
      // A term  (sum over the slots) * (parameter-only factor)  -- the chain rule through a transformed parameter, e.g. d/d log(tau) of
      // tau * z(site): (sum_s g_s * z(site_s)) * exp(.) -- ties all the slots together as it stands.  The factor is distributed over
      // the addends that belong to different column groups (one extra rounding per addend: the only arithmetic of a rolled target
      // that is not the reference's, next to the exact 1 / S of the shared terms); a product inside one group is left as it is.
      // (sum) * f1 * f2 ...: the chain of parameter-only factors around a data-dependent sum, outermost first; side = the sum is
      // the left operand of that product
      struct Wrap { uint32_t fac; bool sum_left; };
      auto split_of = [&](const Term &tm, uint32_t &sum, std::vector<Wrap> &chain) {
        chain.clear();
        uint32_t x = tm.node;
        while (true) {
          const Node &n = P.nodes[x];
          if (n.dep == 0) return false;
          if (n.op == RH_RIR_ADD || n.op == RH_RIR_SUB) { sum = x; return !chain.empty(); }
          if (n.op != RH_RIR_MUL) return false;
          const bool da = P.nodes[n.a].dep != 0, db = P.nodes[n.b].dep != 0;
          if (da == db) return false;
          chain.push_back({da ? n.b : n.a, da});
          x = da ? n.a : n.b;
        }
      };
      uint32_t sum = 0;
      std::vector<Wrap> chain;
      for (auto &to : terms) for (const Term &tm : to) {
        if (split_of(tm, sum, chain)) continue;
        const std::vector<uint32_t> c = R.cols(tm.node);
        for (size_t i = 1; i < c.size(); i++) parent[find(c[i])] = find(c[0]);
        if (!c.empty()) find(c[0]);
      }
      for (auto &to : terms) {
        std::vector<Term> out;
        std::vector<Term> work(to.rbegin(), to.rend());
        while (!work.empty()) {
          const Term tm = work.back(); work.pop_back();
          bool spans = false;
          if (split_of(tm, sum, chain)) {
            const std::vector<uint32_t> c = R.cols(tm.node);
            for (size_t i = 1; i < c.size() && !spans; i++) spans = find(c[i]) != find(c[0]);
            // (x + x) * f, the Translator's 2 x f: written x f + x f for an entry whose rows straddle two slots -- the same double
            // either way (a doubling is exact), so it is brought to the second form everywhere
            if (!spans && P.nodes[sum].op == RH_RIR_ADD && P.nodes[sum].a == P.nodes[sum].b) spans = true;
          }
          if (!spans) { out.push_back(tm); continue; }
          std::vector<Term> addends;
          flatten(P, sum, tm.neg, addends, true);
          const std::vector<Wrap> ch = chain;
          for (size_t i = addends.size(); i-- > 0;) {
            uint32_t prod = addends[i].node;
            for (size_t w = ch.size(); w-- > 0;) prod = ch[w].sum_left ? R.op2(RH_RIR_MUL, prod, ch[w].fac) : R.op2(RH_RIR_MUL, ch[w].fac, prod);   // innermost factor first
            work.push_back({prod, addends[i].neg});
          }
        }
        to.swap(out);
      }
    }
    std::map<uint32_t, char> param_comp;
    for (auto &to : terms) for (const Term &tm : to) {
      const std::vector<uint32_t> c = R.cols(tm.node);
      for (size_t i = 1; i < c.size(); i++) parent[find(c[i])] = find(c[0]);
      if (!c.empty()) find(c[0]);
    }
    for (auto &to : terms) for (const Term &tm : to) { const std::vector<uint32_t> c = R.cols(tm.node); if (!c.empty() && R.hasp[tm.node]) param_comp[find(c[0])] = 1; }
    std::map<uint32_t, std::vector<uint32_t>> comp;
    for (auto &kv : parent) comp[find(kv.first)].push_back(kv.first);
    // every component gets a structural class; classes that parameters reach fix the slot count S; a parameter-free class with S
    // members rolls like any other; a parameter-free component outside such a class is a loose data term and may only be kept
    // (zero-padded for the other slots) when its terms are  column  or  constant * column  (they vanish on the padding)
    std::vector<std::vector<uint32_t>> comps;
    std::vector<char> comp_param;
    std::map<uint32_t, char> loose;
    for (auto &kv : comp) { std::sort(kv.second.begin(), kv.second.end()); comps.push_back(kv.second); comp_param.push_back(param_comp.count(kv.first) ? 1 : 0); }
    {
      std::vector<size_t> idx(comps.size());
      for (size_t c = 0; c < comps.size(); c++) idx[c] = c;
      std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return comps[a][0] < comps[b][0]; });
      std::vector<std::vector<uint32_t>> cs; std::vector<char> cp;
      for (size_t c : idx) { cs.push_back(comps[c]); cp.push_back(comp_param[c]); }
      comps.swap(cs); comp_param.swap(cp);
    }
    const bool loose_param = index_masks_on();
    const bool say = rh::knob("RH_ROLL_WHY") != nullptr;   // RH_ROLL_WHY=1: why a target with several column groups keeps them
    if (comps.size() < 2) continue;
    not_rolled++;                                   // (taken back at the end of the loop body when the target does roll)
    std::map<uint32_t, size_t> comp_of;
    for (size_t c = 0; c < comps.size(); c++) for (uint32_t col : comps[c]) comp_of[col] = c;
    std::vector<uint64_t> ch(comps.size(), 0x5107);
    std::vector<char> linear(comps.size(), 1);
    auto is_col = [&](uint32_t x) { return P.nodes[x].op == RH_RIR_INPUT && P.nodes[x].input >= P.n_params; };
    for (size_t o = 0; o < no; o++) {
      std::vector<std::vector<uint64_t>> hs(comps.size());
      for (const Term &tm : terms[o]) {
        const std::vector<uint32_t> c = R.cols(tm.node);
        if (c.empty()) continue;
        hs[comp_of[c[0]]].push_back(mix(R.hash(tm.node), tm.neg));
        const Node &n = P.nodes[tm.node];
        bool lin = is_col(tm.node) || (n.op == RH_RIR_MUL && ((is_col(n.a) && P.nodes[n.b].op == RH_RIR_CONST) || (is_col(n.b) && P.nodes[n.a].op == RH_RIR_CONST)));
        if (!lin && loose_param) {
          // column * (parameter-only factors), in any nesting: a * (y_1 + ... + y_8) arrives as ONE column times the parameter (the
          // Line algebra has added the 8 slots' columns up front); it vanishes on zero padding like a bare column does
          uint32_t x = tm.node;
          while (P.nodes[x].op == RH_RIR_MUL && P.nodes[x].dep != 0) {
            const Node &m = P.nodes[x];
            if (P.nodes[m.a].dep == 0) x = m.b; else if (P.nodes[m.b].dep == 0) x = m.a; else break;
          }
          lin = is_col(x);
        }
        if (!lin) linear[comp_of[c[0]]] = 0;
      }
      for (size_t c = 0; c < comps.size(); c++) { std::sort(hs[c].begin(), hs[c].end()); uint64_t h = mix(ch[c], o); for (uint64_t x : hs[c]) h = mix(h, x); ch[c] = h; }
    }
    std::map<uint64_t, std::vector<size_t>> classes;
    for (size_t c = 0; c < comps.size(); c++) classes[ch[c]].push_back(c);
    if (say && classes.size() > 2)     // which output tells the column groups apart: the first one on which two of them differ
      for (size_t o = 0; o < no; o++) {
        std::vector<std::vector<uint64_t>> hs(comps.size());
        for (const Term &tm : terms[o]) { const std::vector<uint32_t> c = R.cols(tm.node); if (!c.empty()) hs[comp_of[c[0]]].push_back(mix(R.hash(tm.node), tm.neg)); }
        size_t c0 = comps.size(), cd = comps.size();
        for (size_t c = 0; c < comps.size() && cd == comps.size(); c++) {
          if (!comp_param[c] || comps[c].size() < 2) continue;
          std::sort(hs[c].begin(), hs[c].end());
          if (c0 == comps.size()) c0 = c; else if (hs[c] != hs[c0]) cd = c;
        }
        if (cd != comps.size()) { std::fprintf(stderr, "rainier-hip: strict roll, target %zu: output %zu tells column group %zu (%zu terms) from group %zu (%zu terms)\n", t, o, c0, hs[c0].size(), cd, hs[cd].size()); break; }
      }
    size_t S = 0; bool ok = true;
    for (auto &kv : classes) {
      if (!comp_param[kv.second[0]]) continue;
      if (loose_param && kv.second.size() == 1 && linear[kv.second[0]] && comps[kv.second[0]].size() == 1) continue;   // a loose column times a parameter
      if (S == 0) S = kv.second.size(); else if (kv.second.size() != S) ok = false;
    }
    if (!ok || S < 2 || (S & (S - 1)) != 0) {                     // 1 / S must be exact
      if (say) std::fprintf(stderr, "rainier-hip: strict roll, target %zu: %zu column groups in %zu structural classes, no common slot count (S = %zu)\n", t, comps.size(), classes.size(), S);
      continue;
    }
    for (auto &kv : classes) {
      if (kv.second.size() == S) continue;
      for (size_t c : kv.second) { if ((comp_param[c] && !loose_param) || !linear[c]) ok = false; for (uint32_t col : comps[c]) loose[col] = 1; }
    }
    if (!ok) { if (say) std::fprintf(stderr, "rainier-hip: strict roll, target %zu: a column group outside the %zu slots is not a loose linear data term\n", t, S); continue; }
    std::vector<std::map<uint32_t, char>> slot_cols(S);
    for (auto &kv : classes) if (kv.second.size() == S) for (size_t s = 0; s < S; s++) for (uint32_t col : comps[kv.second[s]]) slot_cols[s][col] = 1;
    std::map<uint32_t, size_t> slot_of;
    for (size_t s = 0; s < S; s++) for (auto &kv : slot_cols[s]) slot_of[kv.first] = s;
    // corresponding columns: a walk over the slot's terms in column-blind hash order
    std::vector<std::vector<uint32_t>> order(S);
    {
      std::vector<std::map<uint32_t, char>> seen_col(S), seen_node(S);
      for (size_t o = 0; o < no; o++) {
        std::vector<std::vector<std::pair<uint64_t, uint32_t>>> st(S);
        for (const Term &tm : terms[o]) { const std::vector<uint32_t> c = R.cols(tm.node); if (!c.empty() && !loose.count(c[0])) st[slot_of[c[0]]].push_back({mix(R.hash(tm.node), tm.neg), tm.node}); }
        for (size_t s = 0; s < S; s++) { std::stable_sort(st[s].begin(), st[s].end(), [](auto &a, auto &b) { return a.first < b.first; }); for (auto &x : st[s]) R.walk(x.second, order[s], seen_col[s], seen_node[s]); }
      }
    }
    const size_t mcols = order[0].size();
    for (size_t s = 0; s < S && ok; s++) ok = order[s].size() == mcols && mcols == slot_cols[s].size();
    if (!ok) { if (say) std::fprintf(stderr, "rainier-hip: strict roll, target %zu: the slots read different numbers of columns\n", t); continue; }
    // verify: slot s's terms, renamed, are slot 0's terms (as multisets, with their signs)
    std::vector<std::vector<Term>> slot0(no);
    for (size_t o = 0; o < no && ok; o++) {
      std::vector<std::vector<std::pair<uint32_t, bool>>> st(S);
      for (const Term &tm : terms[o]) { const std::vector<uint32_t> c = R.cols(tm.node); if (!c.empty() && !loose.count(c[0])) st[slot_of[c[0]]].push_back({tm.node, tm.neg}); }
      std::vector<std::pair<uint32_t, bool>> want = st[0];
      std::sort(want.begin(), want.end());
      for (size_t s = 1; s < S && ok; s++) {
        std::map<uint32_t, uint32_t> cmap, memo;
        for (size_t i = 0; i < mcols; i++) cmap[order[s][i]] = order[0][i];
        std::vector<std::pair<uint32_t, bool>> got;
        for (auto &x : st[s]) got.push_back({R.rename(x.first, cmap, memo), x.second});
        std::sort(got.begin(), got.end());
        ok = got == want;
        if (!ok && say) std::fprintf(stderr, "rainier-hip: strict roll, target %zu: output %zu, slot %zu has %zu terms, slot 0 %zu, and they differ after renaming\n", t, o, s, got.size(), want.size());
      }
    }
    if (!ok) continue;
    // (RH_INDEX_MASKS) a select with parameter-only factors around it, eq(index, k, T, 0) * f, carries them inside: eq(index, k, T * f, 0)
    // -- the selected row computes the same product, the other rows 0 instead of 0 * f -- which is the one-scatter-value shape gather
    // mode reads (emit.cpp detect_gather); hash-consing makes T * f one node for all the entries
    auto sink = [&](uint32_t term) {
      std::vector<std::pair<uint32_t, bool>> chain;     // (factor, select on the left), outermost first
      uint32_t y = term;
      while (P.nodes[y].op == RH_RIR_MUL && P.nodes[y].dep != 0) {
        const Node &n = P.nodes[y];
        const bool da = P.nodes[n.a].dep != 0, db = P.nodes[n.b].dep != 0;
        if (da == db) return term;
        chain.push_back({da ? n.b : n.a, da});
        y = da ? n.a : n.b;
      }
      const Node lk = P.nodes[y];
      if (chain.empty() || lk.op != RH_RIR_LOOKUP || lk.low != -1 || lk.table.size() != 3) return term;
      const Node &z0 = P.nodes[lk.table[0]], &z2 = P.nodes[lk.table[2]];
      if (!(z0.op == RH_RIR_CONST && z0.cval == 0.0 && !std::signbit(z0.cval) && z2.op == RH_RIR_CONST && z2.cval == 0.0 && !std::signbit(z2.cval))) return term;
      if (P.nodes[lk.a].op != RH_RIR_COMPARE) return term;
      uint32_t v = lk.table[1];
      for (size_t w = chain.size(); w-- > 0;) v = chain[w].second ? R.op2(RH_RIR_MUL, v, chain[w].first) : R.op2(RH_RIR_MUL, chain[w].first, v);
      Node q = lk; q.table[1] = v;
      auto li = R.lookups.find(std::make_tuple(q.a, q.low, q.table));
      return li != R.lookups.end() ? li->second : R.push(q);
    };
    // rebuild: original order of the terms that stay (slot 0, loose), shared terms scaled by the exact 1 / S
    const uint32_t inv_s = R.constant(1.0 / (double)S);
    std::vector<uint32_t> outs(no);
    for (size_t o = 0; o < no; o++) {
      uint32_t acc = 0xFFFFFFFFu;
      uint32_t last_sel = 0xFFFFFFFFu;      // (RH_INDEX_MASKS) the select just added: a second copy of it is folded into the first
      for (const Term &tm : terms[o]) {
        const std::vector<uint32_t> c = R.cols(tm.node);
        uint32_t x;
        if (c.empty()) x = R.op2(RH_RIR_MUL, tm.node, inv_s);
        else if (loose.count(c[0]) || slot_of[c[0]] == 0) x = index_masks_on() ? sink(tm.node) : tm.node;
        else continue;
        if (index_masks_on() && !tm.neg && acc == last_sel && x == last_sel && P.nodes[x].op == RH_RIR_LOOKUP && P.nodes[x].low == -1 && P.nodes[x].table.size() == 3) {
          // eq(index, k, V, 0) + eq(index, k, V, 0) -- the Translator's 2 x -- is eq(index, k, V + V, 0): the same sum on the selected
          // row, 0 + 0 elsewhere; ONE select per entry again (the shape scatter families and gather mode read)
          Node q = P.nodes[x];
          q.table[1] = R.op2(RH_RIR_ADD, q.table[1], q.table[1]);
          auto li = R.lookups.find(std::make_tuple(q.a, q.low, q.table));
          acc = li != R.lookups.end() ? li->second : R.push(q);
          last_sel = 0xFFFFFFFFu;
          continue;
        }
        if (acc == 0xFFFFFFFFu) { acc = tm.neg ? R.op2(RH_RIR_SUB, R.constant(0.0), x) : x; last_sel = tm.neg ? 0xFFFFFFFFu : x; }
        else { acc = R.op2(tm.neg ? RH_RIR_SUB : RH_RIR_ADD, acc, x); last_sel = 0xFFFFFFFFu; }
      }
      outs[o] = acc == 0xFFFFFFFFu ? R.constant(0.0) : acc;
    }
    P.targets[t].outputs = outs;
    keep[t] = order[0];
    src[t].assign(mcols, {});
    for (size_t i = 0; i < mcols; i++) for (size_t s = 0; s < S; s++) src[t][i].push_back(order[s][i]);
    for (auto &kv : loose) {
      if (kv.first < P.targets[t].input_start || kv.first >= P.targets[t].input_start + P.targets[t].n_cols) continue;
      keep[t].push_back(kv.first);
      std::vector<uint32_t> cs{kv.first};
      for (size_t s = 1; s < S; s++) cs.push_back(0xFFFFFFFFu);
      src[t].push_back(cs);
    }
    any = true;
    not_rolled--;
  }
  if (!any) return false;
  // renumber the columns that are left
  std::map<uint32_t, uint32_t> renum;
  std::vector<std::vector<double>> dom;
  uint32_t in = P.n_params, colc = 0;
  const uint32_t np = P.n_params;
  parts.clear();
  for (size_t t = 0; t < NT; t++) {
    P.targets[t].input_start = in; P.targets[t].col0 = colc; P.targets[t].n_cols = (uint32_t)keep[t].size();
    for (size_t i = 0; i < keep[t].size(); i++) {
      renum[keep[t][i]] = in + (uint32_t)i;
      std::vector<uint32_t> cs;
      for (uint32_t old : src[t][i]) cs.push_back(old == 0xFFFFFFFFu ? old : old - np);
      parts.push_back(cs);
      dom.push_back({});
    }
    in += P.targets[t].n_cols; colc += P.targets[t].n_cols;
  }
  P.n_inputs = in; P.n_cols_total = colc; P.col_domain = dom;
  for (Node &n : P.nodes)
    if (n.op == RH_RIR_INPUT && n.input >= np) {
      auto it = renum.find(n.input);
      if (it != renum.end()) n.input = it->second;
      else { n.op = RH_RIR_CONST; n.cval = 0.0; n.input = 0; n.dep = 0; }   // only dead nodes still name a dropped column
    }
  return true;
}
}  // namespace

}  // namespace rh
