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
//      compare(c_a, k)              (strict builds, RH_INDEX_MASKS: the per-entry mask of a Lookup's index column, see index_masks below)
// its INPUT node is replaced by that expression over the BASE columns, which alone are uploaded.  Every relation is
// verified on ALL rows (candidates are pre-filtered on 32 sample rows).  In strict mode "equal" means bit-identical, so the
// recomputed value is the very double the reference stored (IEEE negation / multiplication / addition are deterministic);
// fast mode also accepts value equality (+0 == -0).  The pass also records, for every base column with at most 8 distinct
// values, that set (Program::col_domain): the emitter may verify a closed form on the values that actually occur.
//
// Strict builds with RH_INDEX_MASKS=1 (off until it has run on the device) also undo, with the data in hand, what the reference's
// front end did to the gradient of Lookup(index column, table) -- index_masks (the per-entry Compare columns recognised),
// complete_scatter_terms (terms folded for entries no row selects written back; parameter-only factors carried inside the select),
// fold_select_sums (sum_k eq(index, k, T, 0) = T, sum_k eq(index, k, F(z_k), 0) = F(Lookup(index, z))) -- so that hierarchical
// models reach gather mode (emit.cpp detect_gather) and the strict roll (rollstrict.cpp) with the reference's own arithmetic on
// every row.  Each rewrite is exact on the data it was verified on: the selected row computes the same operations on the same
// values, every other row contributes +0.0 either way (a factor that is not finite would have made it NaN: the one difference).
//
// Second job, for programs recognised this way as the reference's lowering: Model.observe's INITIAL CHUNK (1-8 observations in a
// row target of its own, core/Model.scala:84-96) is unrolled -- its rows substituted as constants and summed in row order -- so
// that the model is left with one streamed target.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
  enum Kind { BASE, ALIAS, CONST, NEG, MUL, ADDC, SUBC, MULC, CMPK } kind = BASE;
  int a = -1, b = -1;   // operand original-column indices (local to the target)
  double c = 0.0;
};

inline uint64_t bits(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b; }

struct Same {
  bool fast;
  bool operator()(double x, double y) const { return fast ? x == y : bits(x) == bits(y); }
};

// Strict builds, with the masks recognised: a parameter that every entry of a table depends on (mu in alpha_k = mu + sd * z_k, a
// common scale, ...) gets from the reference the gradient  sum_k eq(index, k, T, 0)  -- one select per entry and row, O(rows x
// entries), with the terms of entries that no row of the slot selects folded away by its front end.  On every row exactly one of
// the selects yields T and the others +0.0, so the sum IS T: the terms of one index column with one common T (and one common chain
// of parameter-only factors around the select, e.g. the 10 of a Normal(0, 10) prior's scale) are replaced by T once per
// multiplicity -- after checking on the data that every row's index is among the entries the sum names.
struct SelectSumFolder {
  std::vector<Node> &N;
  explicit SelectSumFolder(std::vector<Node> &n) : N(n) {}
  static bool binary(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }
  std::map<std::pair<uint32_t, uint32_t>, char> eq_memo;
  bool same(uint32_t a, uint32_t b) {    // structural equality, products in either operand order
    while (N[a].op == RH_RIR_SEQ) a = N[a].b;      // SEQ(definition, use) has the value of `use`
    while (N[b].op == RH_RIR_SEQ) b = N[b].b;
    if (a == b) return true;
    auto key = std::make_pair(std::min(a, b), std::max(a, b));
    auto it = eq_memo.find(key);
    if (it != eq_memo.end()) return it->second != 0;
    const Node &x = N[a], &y = N[b];
    bool r = false;
    if (x.op == y.op) {
      if (x.op == RH_RIR_CONST) r = bits(x.cval) == bits(y.cval);
      else if (x.op == RH_RIR_INPUT) r = x.input == y.input;
      else if (x.op == RH_RIR_LOOKUP) {
        r = x.low == y.low && x.table.size() == y.table.size() && same(x.a, y.a);
        for (size_t i = 0; r && i < x.table.size(); i++) r = same(x.table[i], y.table[i]);
      } else if (x.op == RH_RIR_MUL || x.op == RH_RIR_ADD) r = (same(x.a, y.a) && same(x.b, y.b)) || (same(x.a, y.b) && same(x.b, y.a));   // (a op b = b op a, bit for bit)
      else if (binary(x.op)) r = same(x.a, y.a) && same(x.b, y.b);
      else r = same(x.a, y.a);
    }
    eq_memo[key] = r ? 1 : 0;
    return r;
  }
  // a and b are the same expression except that parameter `pa` of a stands where parameter `pb` stands in b (the first parameter
  // pair that differs fixes pa -> pb; every other occurrence must agree): T_k = F(z_k) against T_0 = F(z_0)
  bool same_but(uint32_t a, uint32_t b, long &pa, long &pb, std::map<std::pair<uint32_t, uint32_t>, char> &memo) {
    while (N[a].op == RH_RIR_SEQ) a = N[a].b;
    while (N[b].op == RH_RIR_SEQ) b = N[b].b;
    const Node &x = N[a], &y = N[b];
    if (x.dep != 0 || y.dep != 0) {       // parts that read data do not hold the entry's parameter: plain equality
      if (a == b) return true;
    }
    auto key = std::make_pair(a, b);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second != 0;
    bool r = false;
    if (x.op == y.op) {
      if (x.op == RH_RIR_CONST) r = bits(x.cval) == bits(y.cval);
      else if (x.op == RH_RIR_INPUT) {
        if (x.input == y.input) r = pa < 0 || (long)x.input != pa;      // (the substituted parameter itself may not appear unsubstituted)
        else if (x.dep == 0 && y.dep == 0) {
          if (pa < 0) { pa = (long)x.input; pb = (long)y.input; r = true; }
          else if (pb < 0) { if ((long)x.input == pa) { pb = (long)y.input; r = true; } }     // the reference's parameter is known, its partner not yet
          else r = (long)x.input == pa && (long)y.input == pb;
        }
      } else if (x.op == RH_RIR_LOOKUP) {
        r = x.low == y.low && x.table.size() == y.table.size() && same_but(x.a, y.a, pa, pb, memo);
        for (size_t i = 0; r && i < x.table.size(); i++) r = same_but(x.table[i], y.table[i], pa, pb, memo);
      } else if (x.op == RH_RIR_MUL || x.op == RH_RIR_ADD) {
        long qa = pa, qb = pb;
        r = same_but(x.a, y.a, qa, qb, memo) && same_but(x.b, y.b, qa, qb, memo);
        if (r) { pa = qa; pb = qb; }
        else { qa = pa; qb = pb; r = same_but(x.a, y.b, qa, qb, memo) && same_but(x.b, y.a, qa, qb, memo); if (r) { pa = qa; pb = qb; } }
      } else if (binary(x.op)) r = same_but(x.a, y.a, pa, pb, memo) && same_but(x.b, y.b, pa, pb, memo);
      else r = same_but(x.a, y.a, pa, pb, memo);
    }
    if (pa >= 0 && pb >= 0) memo[key] = r ? 1 : 0;       // (only once the pair is fixed is the answer final)
    return r;
  }
  // a with parameter pa replaced by node `by`, following the walk that found a = b but for pa -> pb: parts where the two are the SAME
  // node are shared by every entry (the row's own Lookup over the table holds pa as an entry -- it must stay) and are kept
  uint32_t substitute_along(uint32_t a, uint32_t b, long pa, long pb, uint32_t by, uint32_t dep, std::map<std::pair<uint32_t, uint32_t>, uint32_t> &memo) {
    while (N[a].op == RH_RIR_SEQ) a = N[a].b;
    while (N[b].op == RH_RIR_SEQ) b = N[b].b;
    if (a == b) return a;
    auto key = std::make_pair(a, b);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    const Node x = N[a], y = N[b];
    uint32_t r = a;
    if (x.op == RH_RIR_INPUT) r = (x.dep == 0 && (long)x.input == pa) ? by : a;
    else if (x.op != RH_RIR_CONST) {
      Node q = x;
      if (x.op == RH_RIR_MUL || x.op == RH_RIR_ADD) {
        long qa = pa, qb = pb;
        std::map<std::pair<uint32_t, uint32_t>, char> m1;
        const bool straight = same_but(x.a, y.a, qa, qb, m1) && same_but(x.b, y.b, qa, qb, m1);
        q.a = substitute_along(x.a, straight ? y.a : y.b, pa, pb, by, dep, memo);
        q.b = substitute_along(x.b, straight ? y.b : y.a, pa, pb, by, dep, memo);
      } else {
        q.a = substitute_along(x.a, y.a, pa, pb, by, dep, memo);
        if (x.op == RH_RIR_LOOKUP) { for (size_t i = 0; i < x.table.size(); i++) q.table[i] = substitute_along(x.table[i], y.table[i], pa, pb, by, dep, memo); }
        else if (binary(x.op)) q.b = substitute_along(x.b, y.b, pa, pb, by, dep, memo);
      }
      q.dep = dep; N.push_back(q); r = (uint32_t)N.size() - 1;
    }
    return memo[key] = r;
  }
  // `id` with parameter p replaced by node `by` (new nodes appended; parts that do not reach p are shared)
  uint32_t substitute(uint32_t id, long p, uint32_t by, uint32_t dep, std::map<uint32_t, uint32_t> &memo) {
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    const Node n = N[id];
    uint32_t r = id;
    if (n.op == RH_RIR_INPUT) r = (n.dep == 0 && (long)n.input == p) ? by : id;
    else if (n.op != RH_RIR_CONST) {
      Node q = n;
      bool ch = false;
      q.a = substitute(n.a, p, by, dep, memo); ch = ch || q.a != n.a;
      if (n.op == RH_RIR_LOOKUP) { for (size_t i = 0; i < n.table.size(); i++) { q.table[i] = substitute(n.table[i], p, by, dep, memo); ch = ch || q.table[i] != n.table[i]; } }
      else if (binary(n.op)) { q.b = substitute(n.b, p, by, dep, memo); ch = ch || q.b != n.b; }
      if (ch) { q.dep = dep; N.push_back(q); r = (uint32_t)N.size() - 1; }
    }
    return memo[id] = r;
  }
};

// Strict builds, after the masks of a Lookup's index column have been recognised: the gradient of the parameter behind table entry k
// is a sum of  w(eq(index_s, low + k, T_s, 0))  terms, one per Lookup of the target that reads the table (8 in a target written by
// Model.observe's split; w = the parameter-only factors around the select) -- except where the front end has folded one away: a
// mask that is -1 or +1 on every row of its slot is a constant to it, and Lookup(constant, [0, g, 0]) is 0.  Gather mode reads every
// entry's gradient as THE scatter of one value (emit.cpp detect_gather) and the strict roll needs the 8 slots to be the same
// expression (rollstrict.cpp), so the folded terms are written back where the data confirms the folding: no row of that index
// column selects the entry, hence the term adds +0.0 on every row.  The table may be any Lookup of the target whose entries are
// one function of one parameter each (raw parameters, z_k sd + mu, ...), trailing or not.
void complete_scatter_terms(Program &P, const double *const *columns, const int64_t *nrows, const std::vector<uint32_t> &kept) {
  SelectSumFolder F(P.nodes);
  for (size_t t = 0; t < P.targets.size(); t++) {
    Target &T = P.targets[t];
    if (!T.n_cols) continue;
    auto local_col = [&](uint32_t id) { const Node &x = P.nodes[id]; return (x.op == RH_RIR_INPUT && x.input >= P.n_params && x.dep == t + 1) ? (int)(x.input - T.input_start) : -1; };
    struct Wrap { uint32_t fac; bool inner_left; };
    struct Slot { uint32_t ix; int col; int32_t low; std::vector<long> param; bool have = false, bad = false, nested = false; uint32_t value = 0; std::vector<Wrap> chain; std::vector<char> present;
                  uint32_t table_node = 0; bool raw = false, own_param = false; size_t ref_k = 0; };
    std::vector<Slot> slots;
    const size_t n0 = P.nodes.size();
    for (size_t i = 0; i < n0; i++) {
      if (P.nodes[i].op != RH_RIR_LOOKUP || P.nodes[i].table.size() <= 3 || local_col(P.nodes[i].a) < 0) continue;
      const Node nd = P.nodes[i];
      bool known = false;
      for (const Slot &sl : slots) known = known || sl.ix == nd.a;
      if (known) continue;
      // every entry the same function of ONE parameter of its own
      Slot sl; sl.ix = nd.a; sl.col = local_col(nd.a); sl.low = nd.low;
      long pa = -1;
      bool ok = true;
      for (size_t k = 1; k < nd.table.size() && ok; k++) {
        long qa = pa, qb = -1;
        std::map<std::pair<uint32_t, uint32_t>, char> memo;
        ok = F.same_but(nd.table[0], nd.table[k], qa, qb, memo) && qa >= 0 && qb >= 0 && qa != qb && (pa < 0 || pa == qa);
        if (ok) { if (pa < 0) { pa = qa; sl.param.push_back(pa); } sl.param.push_back(qb); }
      }
      if (!ok || sl.param.size() != nd.table.size()) continue;
      { std::vector<long> u = sl.param; std::sort(u.begin(), u.end()); if (std::adjacent_find(u.begin(), u.end()) != u.end()) continue; }
      sl.present.assign(nd.table.size(), 0);
      sl.table_node = (uint32_t)i;
      sl.raw = true;
      for (size_t k = 0; k < nd.table.size(); k++) sl.raw = sl.raw && P.nodes[nd.table[k]].op == RH_RIR_INPUT && (long)P.nodes[nd.table[k]].input == sl.param[k];
      slots.push_back(sl);
    }
    if (slots.empty()) continue;
    // the select terms that are there: the output of an entry's parameter, cut along its top-level additions
    auto peel = [&](uint32_t y, std::vector<Wrap> &chain, uint32_t &lookup) {
      chain.clear();
      while (true) {
        const Node &n = P.nodes[y];
        if (n.op == RH_RIR_SEQ) { y = n.b; continue; }
        if (n.op == RH_RIR_NOOP) { y = n.a; continue; }
        if (n.op == RH_RIR_ADD && n.dep != 0 && n.a == n.b) { chain.push_back({0xFFFFFFFFu, true}); y = n.a; continue; }   // x + x: the Translator's 2 x
        if (!(n.op == RH_RIR_MUL && n.dep != 0)) break;
        const bool da = P.nodes[n.a].dep != 0, db = P.nodes[n.b].dep != 0;
        if (da == db) return false;
        chain.push_back({da ? n.b : n.a, da});
        y = da ? n.a : n.b;
      }
      const Node &lk = P.nodes[y];
      if (lk.op != RH_RIR_LOOKUP || lk.low != -1 || lk.table.size() != 3) return false;
      const Node &z0 = P.nodes[lk.table[0]], &z2 = P.nodes[lk.table[2]];
      if (!(z0.op == RH_RIR_CONST && z0.cval == 0.0 && z2.op == RH_RIR_CONST && z2.cval == 0.0)) return false;
      const Node &cm = P.nodes[lk.a];
      if (cm.op != RH_RIR_COMPARE || P.nodes[cm.b].op != RH_RIR_CONST) return false;
      lookup = y;
      return true;
    };
    std::vector<uint32_t> stack;
    std::vector<Wrap> chain;
    // (the selects may sit below a sum that a parameter-only factor multiplies -- (sum over the slots of eq(site_s, k, g_s, 0)) * sd --:
    //  the walk goes through sums and such products alike and keeps the factors met on the way, outermost first; `nested` = a
    //  select was found below a product of a sum, where its term cannot be rewritten in place)
    for (Slot &sl : slots) {
      for (size_t k = 0; k < sl.param.size(); k++) {
        int steps = 0;
        std::function<void(uint32_t, bool)> walk = [&](uint32_t y, bool under_product) {
          if (++steps > 65536) return;
          const Node &x = P.nodes[y];
          if (x.op == RH_RIR_SEQ) { walk(x.b, under_product); return; }
          if (x.op == RH_RIR_NOOP) { walk(x.a, under_product); return; }
          if (x.dep != 0 && x.op == RH_RIR_ADD) {
            if (x.a == x.b) { chain.push_back({0xFFFFFFFFu, true}); walk(x.a, under_product); chain.pop_back(); }
            else { if (!chain.empty()) sl.nested = true; walk(x.a, under_product); walk(x.b, under_product); }
            return;
          }
          if (x.dep != 0 && x.op == RH_RIR_MUL) {
            const bool da = P.nodes[x.a].dep != 0, db = P.nodes[x.b].dep != 0;
            if (da == db) return;
            chain.push_back({da ? x.b : x.a, da}); walk(da ? x.a : x.b, true); chain.pop_back();
            return;
          }
          if (x.op != RH_RIR_LOOKUP || x.low != -1 || x.table.size() != 3) return;
          const Node &z0 = P.nodes[x.table[0]], &z2 = P.nodes[x.table[2]];
          if (!(z0.op == RH_RIR_CONST && z0.cval == 0.0 && z2.op == RH_RIR_CONST && z2.cval == 0.0)) return;
          const Node &cm = P.nodes[x.a];
          if (cm.op != RH_RIR_COMPARE || P.nodes[cm.b].op != RH_RIR_CONST) return;
          if (cm.a != sl.ix || P.nodes[cm.b].cval != (double)sl.low + (double)k) return;
          sl.present[k] = 1;
          const uint32_t val = x.table[1];
          if (!sl.have) { sl.have = true; sl.value = val; sl.chain = chain; sl.ref_k = k; }
          else {
            bool eq = F.same(sl.value, val) && sl.chain.size() == chain.size();
            for (size_t w = 0; eq && w < chain.size(); w++) {
              eq = chain[w].inner_left == sl.chain[w].inner_left;
              if (!eq) break;
              if (chain[w].fac == 0xFFFFFFFFu || sl.chain[w].fac == 0xFFFFFFFFu) { eq = chain[w].fac == sl.chain[w].fac; continue; }
              if (F.same(chain[w].fac, sl.chain[w].fac)) continue;
              // the factor holds the entry's own parameter (entries exp(z_k): d/d z_k = eq(site, k, g, 0) * exp(z_k)): the template's
              // factor with the template entry's parameter replaced by this entry's
              long pa = sl.param[sl.ref_k], pb = sl.param[k];
              std::map<std::pair<uint32_t, uint32_t>, char> memo;
              eq = pa != pb && F.same_but(sl.chain[w].fac, chain[w].fac, pa, pb, memo);
              if (eq) sl.own_param = true;
            }
            if (!eq) sl.bad = true;
          }
        };
        chain.clear();
        walk(T.outputs[1 + (size_t)sl.param[k]], false);
      }
    }
    uint32_t zero = 0xFFFFFFFFu;
    for (Slot &sl : slots) {
      if (sl.own_param && !sl.raw) sl.bad = true;      // (the entry's parameter is put in through the table: its entries must BE the parameters)
      if (!sl.have || sl.bad) continue;
      const size_t cnt = sl.param.size();
      std::vector<char> selected(cnt, 0);        // which entries do the rows of this index column select?
      const double *c = columns[kept[T.col0 + (uint32_t)sl.col]];
      for (int64_t r = 0; r < nrows[t]; r++) {
        const double kk = c[r] - (double)sl.low;
        if (kk >= 0.0 && kk < (double)cnt && kk == std::floor(kk)) selected[(size_t)kk] = 1;   // (a NaN index compares as -1 with everything)
      }
      for (size_t k = 0; k < cnt; k++) {
        if (sl.present[k] || selected[k]) continue;
        if (zero == 0xFFFFFFFFu) { Node z; z.op = RH_RIR_CONST; z.cval = 0.0; P.nodes.push_back(z); zero = (uint32_t)P.nodes.size() - 1; }
        Node kc; kc.op = RH_RIR_CONST; kc.cval = (double)sl.low + (double)k; P.nodes.push_back(kc);
        Node cm; cm.op = RH_RIR_COMPARE; cm.a = sl.ix; cm.b = (uint32_t)P.nodes.size() - 1; cm.dep = (uint32_t)t + 1; P.nodes.push_back(cm);
        Node lk; lk.op = RH_RIR_LOOKUP; lk.a = (uint32_t)P.nodes.size() - 1; lk.low = -1; lk.table = {zero, sl.value, zero}; lk.dep = (uint32_t)t + 1; P.nodes.push_back(lk);
        uint32_t term = (uint32_t)P.nodes.size() - 1;
        std::map<uint32_t, uint32_t> smemo;
        for (size_t w = sl.chain.size(); w-- > 0;) {
          Node mu; mu.op = RH_RIR_MUL; mu.dep = (uint32_t)t + 1;
          uint32_t fac = sl.chain[w].fac;
          if (fac != 0xFFFFFFFFu && sl.own_param) fac = F.substitute(fac, sl.param[sl.ref_k], P.nodes[sl.table_node].table[k], 0, smemo);   // (raw tables only, see below)
          if (fac == 0xFFFFFFFFu) { mu.op = RH_RIR_ADD; mu.a = term; mu.b = term; }
          else if (sl.chain[w].inner_left) { mu.a = term; mu.b = fac; } else { mu.a = fac; mu.b = term; }
          P.nodes.push_back(mu); term = (uint32_t)P.nodes.size() - 1;
        }
        uint32_t &out = T.outputs[1 + (size_t)sl.param[k]];
        const Node &o = P.nodes[out];
        if (o.op == RH_RIR_CONST && o.cval == 0.0) out = term;
        else { Node ad; ad.op = RH_RIR_ADD; ad.a = out; ad.b = term; ad.dep = (uint32_t)t + 1; P.nodes.push_back(ad); out = (uint32_t)P.nodes.size() - 1; }
      }
    }
    // The mask carried as a select, not as a product: w(eq(index, k, T, 0)) with parameter-only factors w around the select becomes
    // eq(index, k, w(T), 0) -- on the selected row the same operations on the same values, elsewhere 0 instead of 0 * factors (equal
    // unless a factor is not finite).  The entries of a slot share ONE w(T): the scatter value gather mode reads.
    for (Slot &sl : slots) {
      if (!sl.have || sl.bad || sl.chain.empty() || sl.nested) continue;     // (nested: the strict roll carries the factors in, rollstrict.cpp)
      if (zero == 0xFFFFFFFFu) { Node z; z.op = RH_RIR_CONST; z.cval = 0.0; P.nodes.push_back(z); zero = (uint32_t)P.nodes.size() - 1; }
      uint32_t v = sl.value;
      std::map<uint32_t, uint32_t> vmemo;
      for (size_t w = sl.chain.size(); w-- > 0;) {
        Node mu; mu.op = RH_RIR_MUL; mu.dep = (uint32_t)t + 1;
        uint32_t fac = sl.chain[w].fac;
        // a factor that holds the entry's own parameter reads it through the table: on the selected row Lookup(index, z) IS z_k
        if (fac != 0xFFFFFFFFu && sl.own_param) fac = F.substitute(fac, sl.param[sl.ref_k], sl.table_node, (uint32_t)t + 1, vmemo);
        if (fac == 0xFFFFFFFFu) { mu.op = RH_RIR_ADD; mu.a = v; mu.b = v; }
        else if (sl.chain[w].inner_left) { mu.a = v; mu.b = fac; } else { mu.a = fac; mu.b = v; }
        P.nodes.push_back(mu); v = (uint32_t)P.nodes.size() - 1;
      }
      for (size_t k = 0; k < sl.param.size(); k++) {
        std::vector<uint32_t> terms;
        stack.assign(1, T.outputs[1 + (size_t)sl.param[k]]);
        while (!stack.empty() && terms.size() < 65536) {
          const uint32_t y = stack.back(); stack.pop_back();
          const Node &x = P.nodes[y];
          if (x.dep != 0 && x.op == RH_RIR_ADD) { stack.push_back(x.b); stack.push_back(x.a); continue; }
          if (x.op == RH_RIR_NOOP && P.nodes[x.a].op == RH_RIR_ADD) { stack.push_back(x.a); continue; }
          if (x.op == RH_RIR_SEQ) { stack.push_back(x.b); continue; }
          terms.push_back(y);
        }
        bool touched = false;
        for (uint32_t &y : terms) {
          uint32_t lk = 0;
          if (!peel(y, chain, lk) || chain.empty()) continue;
          const Node cm = P.nodes[P.nodes[lk].a];
          if (cm.a != sl.ix || P.nodes[cm.b].cval != (double)sl.low + (double)k) continue;
          Node nl; nl.op = RH_RIR_LOOKUP; nl.a = P.nodes[lk].a; nl.low = -1; nl.table = {zero, v, zero}; nl.dep = (uint32_t)t + 1;
          P.nodes.push_back(nl); y = (uint32_t)P.nodes.size() - 1;
          touched = true;
        }
        if (!touched) continue;
        uint32_t acc = terms[0];
        for (size_t i = 1; i < terms.size(); i++) { Node ad; ad.op = RH_RIR_ADD; ad.a = acc; ad.b = terms[i]; ad.dep = (uint32_t)t + 1; P.nodes.push_back(ad); acc = (uint32_t)P.nodes.size() - 1; }
        T.outputs[1 + (size_t)sl.param[k]] = acc;
      }
    }
  }
}

void fold_select_sums(Program &P, const double *const *columns, const int64_t *nrows, const std::vector<uint32_t> &kept) {
  const size_t n_old = P.nodes.size();
  // a sum's root: an ADD over data that something other than an ADD reads (or an output names); the ADDs below it are its chain
  std::vector<char> root(n_old, 0);
  for (size_t x = 0; x < n_old; x++) {
    const Node &n = P.nodes[x];
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    auto use = [&](uint32_t o) { if (n.op != RH_RIR_ADD || n.dep == 0) root[o] = 1; };
    use(n.a);
    if (n.op == RH_RIR_LOOKUP) for (uint32_t e : n.table) use(e);
    else if ((n.op >= RH_RIR_ADD && n.op <= RH_RIR_COMPARE) || n.op == RH_RIR_SEQ) use(n.b);
  }
  for (const Target &T : P.targets) for (uint32_t o : T.outputs) root[o] = 1;
  std::vector<Node> Q;
  Q.reserve(n_old + n_old / 8);
  std::vector<uint32_t> m(n_old, 0xFFFFFFFFu);
  SelectSumFolder F(Q);
  bool any = false;
  struct Wrap { uint32_t fac; bool inner_left; };
  struct Cand { size_t pos; uint32_t ix; double c; uint32_t T; std::vector<Wrap> chain; };
  for (size_t x = 0; x < n_old; x++) {
    const Node &n = P.nodes[x];
    if (!(n.op == RH_RIR_ADD && n.dep != 0)) {
      Node q = n;
      if (n.op != RH_RIR_CONST && n.op != RH_RIR_INPUT) {
        q.a = m[n.a];
        if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = m[e]; }
        else if ((n.op >= RH_RIR_ADD && n.op <= RH_RIR_COMPARE) || n.op == RH_RIR_SEQ) q.b = m[n.b];
      }
      Q.push_back(q); m[x] = (uint32_t)Q.size() - 1;
      continue;
    }
    if (!root[x]) continue;     // read only through the chain of the sum above it
    const size_t t = (size_t)n.dep - 1;
    const Target &T = P.targets[t];
    // the addends, left to right (old ids; through SEQ(definition, use), whose value is `use`)
    std::vector<uint32_t> terms;
    {
      std::vector<uint32_t> stack{(uint32_t)x};
      while (!stack.empty()) {
        const uint32_t y = stack.back(); stack.pop_back();
        const Node &w = P.nodes[y];
        if (w.dep != 0 && w.op == RH_RIR_ADD) { stack.push_back(w.b); stack.push_back(w.a); }
        else terms.push_back(m[y]);
      }
    }
    auto local_col = [&](uint32_t id) { const Node &c = Q[id]; return (c.op == RH_RIR_INPUT && c.input >= P.n_params && c.dep == t + 1) ? (int)(c.input - T.input_start) : -1; };
    std::vector<Cand> cands;
    for (size_t i = 0; i < terms.size(); i++) {
      Cand c; c.pos = i;
      uint32_t y = terms[i];
      while (true) {      // parameter-only factors around the select, and SEQ(definition, use) read as `use`, in any nesting
        if (Q[y].op == RH_RIR_SEQ) { y = Q[y].b; continue; }
        if (!(Q[y].op == RH_RIR_MUL && Q[y].dep != 0)) break;
        const Node &mu = Q[y];
        const bool da = Q[mu.a].dep != 0, db = Q[mu.b].dep != 0;
        if (da == db) break;
        c.chain.push_back({da ? mu.b : mu.a, da});
        y = da ? mu.a : mu.b;
      }
      const Node &lk = Q[y];
      if (lk.op != RH_RIR_LOOKUP || lk.low != -1 || lk.table.size() != 3) continue;
      const Node &z0 = Q[lk.table[0]], &z2 = Q[lk.table[2]];
      if (!(z0.op == RH_RIR_CONST && bits(z0.cval) == bits(0.0) && z2.op == RH_RIR_CONST && bits(z2.cval) == bits(0.0))) continue;
      const Node &cm = Q[lk.a];
      if (cm.op != RH_RIR_COMPARE || local_col(cm.a) < 0 || Q[cm.b].op != RH_RIR_CONST) continue;
      c.ix = cm.a; c.c = Q[cm.b].cval; c.T = lk.table[1];
      cands.push_back(c);
    }
    std::vector<char> used(cands.size(), 0), drop(terms.size(), 0);
    std::map<size_t, std::vector<uint32_t>> insert_at;   // position of a group's first term -> the folded terms
    for (size_t i = 0; i < cands.size(); i++) {
      if (used[i]) continue;
      std::vector<size_t> grp{i};
      for (size_t j = i + 1; j < cands.size(); j++) {
        if (used[j] || cands[j].ix != cands[i].ix || cands[j].chain.size() != cands[i].chain.size() || !F.same(cands[j].T, cands[i].T)) continue;
        bool ch = true;
        for (size_t w = 0; ch && w < cands[i].chain.size(); w++) ch = cands[j].chain[w].inner_left == cands[i].chain[w].inner_left && F.same(cands[j].chain[w].fac, cands[i].chain[w].fac);
        if (ch) grp.push_back(j);
      }
      // ... or one common FUNCTION of the entry's own parameter, T_k = F(z_k) (the gradient with respect to a scale the entries
      // share: sum_k eq(index, k, g * z_k, 0)): on every row the sum is F of the selected entry's parameter, i.e. F(Lookup(index, z))
      std::map<double, long> entry_param;
      long ref_param = -1;
      if (grp.size() == 1) {
        for (size_t j = i + 1; j < cands.size(); j++) {
          if (used[j] || cands[j].ix != cands[i].ix || cands[j].chain.size() != cands[i].chain.size()) continue;
          long pa = ref_param, pb = -1;
          std::map<std::pair<uint32_t, uint32_t>, char> memo;
          bool ch = true;      // (the entry's parameter may sit in the factors around the select as well as in its value)
          for (size_t w = 0; ch && w < cands[i].chain.size(); w++) ch = cands[j].chain[w].inner_left == cands[i].chain[w].inner_left && F.same_but(cands[i].chain[w].fac, cands[j].chain[w].fac, pa, pb, memo);
          if (!ch) continue;
          if (!F.same_but(cands[i].T, cands[j].T, pa, pb, memo) || pa < 0 || pb < 0 || pa == pb) continue;
          if (ref_param < 0) ref_param = pa; else if (ref_param != pa) continue;
          auto it = entry_param.find(cands[j].c);
          if (it != entry_param.end() && it->second != pb) continue;
          entry_param[cands[j].c] = pb;
          grp.push_back(j);
        }
        if (grp.size() > 1) entry_param[cands[i].c] = ref_param;
      }
      for (size_t g : grp) used[g] = 1;
      std::map<double, int> mult;
      for (size_t g : grp) mult[cands[g].c]++;
      if (mult.size() < 2) continue;
      const int mm = mult.begin()->second;
      bool even = true;
      for (auto &kv : mult) even = even && kv.second == mm;
      if (!even) continue;
      // every row's index must be one of the entries the sum names (then exactly one select yields T on that row)
      const double *col = columns[kept[T.col0 + (uint32_t)local_col(cands[i].ix)]];
      bool covered = true;
      for (int64_t r = 0; r < nrows[t] && covered; r++) covered = mult.count(col[r]) != 0;
      if (!covered) continue;
      uint32_t value = cands[i].T;
      if (ref_param >= 0) {
        // the table: a Lookup of this index column whose entries hold those parameters, entry for entry -- taken from the program (the
        // value output reads it), so that every slot of a split target gets the same table; its entries reduced to their parameters
        uint32_t lv = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < Q.size() && lv == 0xFFFFFFFFu; q++) {
          const Node &c = Q[q];
          if (c.op != RH_RIR_LOOKUP || c.a != cands[i].ix || c.table.size() <= 3) continue;
          bool fits = true;
          for (auto &kv : entry_param) { const double kk = kv.first - (double)c.low; fits = fits && kk >= 0.0 && kk < (double)c.table.size() && kk == std::floor(kk); }
          if (fits) lv = q;
        }
        std::map<long, uint32_t> param_node;
        for (uint32_t q = 0; q < Q.size(); q++) if (Q[q].op == RH_RIR_INPUT && Q[q].dep == 0) param_node[(long)Q[q].input] = q;
        Node nl; nl.op = RH_RIR_LOOKUP; nl.a = cands[i].ix; nl.dep = (uint32_t)t + 1;
        bool ok_tab = lv != 0xFFFFFFFFu;
        if (ok_tab) {
          const Node tab = Q[lv];
          const uint32_t ref_entry = tab.table[(size_t)(cands[i].c - (double)tab.low)];
          nl.low = tab.low;
          for (size_t k = 0; k < tab.table.size() && ok_tab; k++) {
            long pa = ref_param, pb = -1;
            std::map<std::pair<uint32_t, uint32_t>, char> memo;
            if (tab.table[k] == ref_entry) pb = ref_param;
            else if (!F.same_but(ref_entry, tab.table[k], pa, pb, memo) || pb < 0) ok_tab = false;
            auto known = entry_param.find((double)tab.low + (double)k);
            if (ok_tab && known != entry_param.end() && known->second != pb) ok_tab = false;     // the sum's own pairing must agree
            if (ok_tab && !param_node.count(pb)) ok_tab = false;
            if (ok_tab) nl.table.push_back(param_node[pb]);
          }
        }
        if (!ok_tab) {
          // ... or, when the program's table holds more than the parameter (year effects y b1 + y^2 b2 + y^3 b3 + eps_year), the table
          // the sum itself spells out -- if it names a gap-free run of entries (in a split target every slot must see every entry then,
          // or the slots end up with different tables and are not rolled)
          nl.table.clear();
          const double lo = entry_param.begin()->first, hi = entry_param.rbegin()->first;
          ok_tab = lo == std::floor(lo) && hi - lo + 1.0 == (double)entry_param.size() && std::fabs(lo) < 1e9;
          nl.low = (int32_t)lo;
          for (auto &kv : entry_param) { ok_tab = ok_tab && kv.first == std::floor(kv.first) && param_node.count(kv.second); if (ok_tab) nl.table.push_back(param_node[kv.second]); }
        }
        if (!ok_tab) continue;
        uint32_t lnew = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < Q.size() && lnew == 0xFFFFFFFFu; q++)      // the program's own Lookup when it already is that table
          if (Q[q].op == RH_RIR_LOOKUP && Q[q].a == nl.a && Q[q].low == nl.low && Q[q].table == nl.table) lnew = q;
        if (lnew == 0xFFFFFFFFu) { Q.push_back(nl); lnew = (uint32_t)Q.size() - 1; }
        const Cand &other = cands[grp[1]];          // any other member: its tree shows where the entry's parameter stands
        const long other_param = entry_param[other.c];
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> smemo;
        value = F.substitute_along(cands[i].T, other.T, ref_param, other_param, lnew, (uint32_t)t + 1, smemo);
        for (size_t w = 0; w < cands[i].chain.size(); w++) cands[i].chain[w].fac = F.substitute_along(cands[i].chain[w].fac, other.chain[w].fac, ref_param, other_param, lnew, (uint32_t)t + 1, smemo);
      }
      for (size_t g : grp) drop[cands[g].pos] = 1;
      std::vector<uint32_t> folded;
      for (int rep = 0; rep < mm; rep++) {
        uint32_t y = value;
        for (size_t w = cands[i].chain.size(); w-- > 0;) {
          Node mu; mu.op = RH_RIR_MUL; mu.dep = (uint32_t)t + 1;
          if (cands[i].chain[w].inner_left) { mu.a = y; mu.b = cands[i].chain[w].fac; } else { mu.a = cands[i].chain[w].fac; mu.b = y; }
          Q.push_back(mu); y = (uint32_t)Q.size() - 1;
        }
        folded.push_back(y);
      }
      insert_at[cands[grp[0]].pos] = folded;
      any = true;
    }
    uint32_t acc = 0xFFFFFFFFu;
    auto add = [&](uint32_t y) {
      if (acc == 0xFFFFFFFFu) { acc = y; return; }
      Node ad; ad.op = RH_RIR_ADD; ad.a = acc; ad.b = y; ad.dep = (uint32_t)t + 1;
      Q.push_back(ad); acc = (uint32_t)Q.size() - 1;
    };
    for (size_t i = 0; i < terms.size(); i++) {
      auto it = insert_at.find(i);
      if (it != insert_at.end()) for (uint32_t y : it->second) add(y);
      if (!drop[i]) add(terms[i]);
    }
    m[x] = acc;
  }
  if (!any) return;          // nothing folded: the program stays as it is, node for node
  for (Target &T : P.targets) for (uint32_t &o : T.outputs) o = m[o];
  P.nodes.swap(Q);
}

// Do two programs over the same inputs compute the same outputs?  Sample rows of every row target x 3 parameter vectors, extended
// precision (rederive.cpp's block interpreter); "same" = equal to 1e-13 relative, or the same NaN / infinity on both sides.
bool same_outputs(const Program &A, const Program &Bq, const double *const *columns, const int64_t *nrows, const std::vector<uint32_t> &kept) {
  if (A.targets.size() != Bq.targets.size() || A.n_inputs != Bq.n_inputs) return false;
  constexpr int B = 32;
  static const double kScale[3] = {1.2, 0.2, 3.6};
  for (size_t t = 0; t < A.targets.size(); t++) {
    const Target &TA = A.targets[t], &TB = Bq.targets[t];
    if (!TA.n_cols || nrows[t] <= 0) continue;
    if (TA.outputs == TB.outputs && A.nodes.size() == Bq.nodes.size()) continue;     // untouched
    const BlockEvaluator EA(A, TA.outputs), EB(Bq, TB.outputs);
    const int64_t nr = nrows[t];
    const int S = (int)std::min<int64_t>(96, nr);
    uint64_t lcg = 0x9E3779B97F4A7C15ull + (uint64_t)t;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0 - 0.5; };
    std::vector<long double> in((size_t)A.n_inputs * B, 0.0L), va, vb;
    std::vector<char> oka, okb;
    for (int trial = 0; trial < 3; trial++) {
      std::vector<double> th(A.n_params);
      for (uint32_t p = 0; p < A.n_params; p++) th[p] = kScale[trial] * uni();
      for (int s0 = 0; s0 < S; s0 += B) {
        const int nb = std::min(B, S - s0);
        for (int r = 0; r < B; r++) {
          const int64_t row = (int64_t)std::min(s0 + r, S - 1) * nr / S;
          for (uint32_t p = 0; p < A.n_params; p++) in[(size_t)p * B + r] = (long double)th[p];
          for (uint32_t j = 0; j < TA.n_cols; j++) in[(size_t)(TA.input_start + j) * B + r] = (long double)columns[kept[TA.col0 + j]][row];
        }
        if (!EA.run(in, B, va, oka) || !EB.run(in, B, vb, okb)) return false;
        for (size_t o = 0; o < TA.outputs.size(); o++) {
          const long double *a = &va[(size_t)TA.outputs[o] * B], *b = &vb[(size_t)TB.outputs[o] * B];
          for (int r = 0; r < nb; r++) {
            if (oka[(size_t)r] != okb[(size_t)r]) return false;
            if (!oka[(size_t)r]) continue;
            const bool fa = std::isfinite(a[r]), fb = std::isfinite(b[r]);
            if (!fa || !fb) { if (fa != fb || std::isnan(a[r]) != std::isnan(b[r]) || (std::isinf(a[r]) && a[r] != b[r])) return false; continue; }
            if (std::fabs(a[r] - b[r]) > 1e-13L * std::max(std::fabs(a[r]), std::fabs(b[r]))) return false;
          }
        }
      }
    }
  }
  return true;
}

}  // namespace

bool index_masks_on() {
  const char *e = rh::knob("RH_INDEX_MASKS");
  return e ? std::atoi(e) != 0 : true;
}

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
    // Strict builds: the per-entry masks of a Lookup's index column.  The reference's gradient of Lookup(index, table) with respect to
    // entry k is eq(index, low + k, g, 0) = Lookup(Compare(index, low + k), [0, g, 0], -1) (compute/Gradient.scala:146-152,
    // compute/Real.scala:42), and with `index` a data column its front end evaluates Compare(index, low + k) ahead of time: one column
    // of -1 / 0 / +1 per entry (and per slot of Model.observe's split).  Fast builds get the natural form back by deriving the gradient
    // again (rederive.cpp); strict builds keep the reference's expression, so here the columns themselves are recognised: column j IS
    // compare(c_a, k) on every row, c_a the index column of a Lookup of this target.  k comes from a row where the mask is 0; a mask
    // without such a row (an entry no row of this target selects) takes the k of the table entry whose gradient output reads it --
    // any k that reproduces the column is exact on this data, that one keeps the shape gather mode reads (emit.cpp detect_gather).
    std::vector<int> ix_cols;                                   // index columns of this target's Lookups (local indices)
    std::vector<char> hinted((size_t)nc, 0);                    // 1: hint_k valid, 2: contradictory hints
    std::vector<double> hint_k((size_t)nc, 0.0);
    if (index_masks_on() && !fast) {
      auto local_col = [&](uint32_t id) { const Node &x = P.nodes[id]; return (x.op == RH_RIR_INPUT && x.input >= P.n_params && x.dep == t + 1) ? (int)(x.input - T.input_start) : -1; };
      bool shaped = false, consistent = true;
      uint32_t first = 0; int32_t low = 0;
      for (const Node &nd : P.nodes) {
        if (nd.op != RH_RIR_LOOKUP || nd.table.size() <= 3) continue;
        const int a = local_col(nd.a);
        if (a < 0) continue;
        if (std::find(ix_cols.begin(), ix_cols.end(), a) == ix_cols.end()) ix_cols.push_back(a);
        // a table that is the run of trailing parameters (the shape of gather mode): its entries' gradient outputs name their masks
        const Node &t0 = P.nodes[nd.table[0]];
        if (!(t0.op == RH_RIR_INPUT && t0.input < P.n_params && t0.input + nd.table.size() == P.n_params)) continue;
        bool run = true;
        for (size_t k = 0; k < nd.table.size() && run; k++) { const Node &tk = P.nodes[nd.table[k]]; run = tk.op == RH_RIR_INPUT && tk.input == t0.input + k; }
        if (!run) continue;
        if (shaped && (first != t0.input || low != nd.low)) consistent = false;
        shaped = true; first = t0.input; low = nd.low;
      }
      if (shaped && consistent) {
        std::vector<uint32_t> stack;
        for (uint32_t p = first; p < P.n_params; p++) {
          stack.assign(1, T.outputs[1 + p]);
          for (int steps = 0; !stack.empty() && steps < 4096; steps++) {   // the output cut along its top-level additions
            const Node &x = P.nodes[stack.back()]; stack.pop_back();
            if (x.op == RH_RIR_ADD) { stack.push_back(x.a); stack.push_back(x.b); continue; }
            if (x.op == RH_RIR_NOOP) { stack.push_back(x.a); continue; }
            if (x.op == RH_RIR_SEQ) { stack.push_back(x.b); continue; }     // (the Translator's SEQ(definition, use): the value is `use`)
            if (x.op == RH_RIR_MUL && x.dep != 0 && (P.nodes[x.a].dep == 0) != (P.nodes[x.b].dep == 0)) {   // a parameter-only factor around the select
              stack.push_back(P.nodes[x.a].dep == 0 ? x.b : x.a);
              continue;
            }
            if (x.op != RH_RIR_LOOKUP || x.low != -1 || x.table.size() != 3) continue;
            const int j = local_col(x.a);
            if (j < 0) continue;
            const double k = (double)low + (double)(p - first);
            if (hinted[(size_t)j] == 1 && hint_k[(size_t)j] != k) hinted[(size_t)j] = 2;
            else if (!hinted[(size_t)j]) { hinted[(size_t)j] = 1; hint_k[(size_t)j] = k; }
          }
        }
      }
    }
    auto cmp3 = [](double l, double r) { return l > r ? 1.0 : (l == r ? 0.0 : -1.0); };   // DCMPL; I2D (ir/MethodGenerator.scala:56-94)
    auto try_mask = [&](int j, CExpr &e) {
      const double *c = col[(size_t)j];
      for (int64_t r : sr) if (!(bits(c[r]) == bits(1.0) || bits(c[r]) == bits(0.0) || bits(c[r]) == bits(-1.0))) return false;
      for (int a : ix_cols) {
        if (a == j) continue;
        const double *ca = col[(size_t)a];
        if (a > j) {   // a later column as the operand: not one that could itself be recognised as derived from this one (no cycles)
          bool tri = true;
          for (int64_t r : sr) tri = tri && (ca[r] == 1.0 || ca[r] == 0.0 || ca[r] == -1.0);
          if (tri) continue;
        }
        double k = 0.0;
        bool have = false;
        if (hinted[(size_t)j] == 1) { k = hint_k[(size_t)j]; have = true; }
        else {
          double lo = -HUGE_VAL, hi = HUGE_VAL;
          for (int64_t r = 0; r < nr && !have; r++) {
            if (c[r] == 0.0) { k = ca[r]; have = true; }
            else if (c[r] < 0.0) { if (ca[r] > lo) lo = ca[r]; }
            else if (ca[r] < hi) hi = ca[r];
          }
          // no row selects the entry: the smallest integer above every "-1" row, if it is below every "+1" row (a column that is
          // the same on every row is left to the constant rule)
          if (!have && std::isfinite(lo) && std::isfinite(hi)) { k = std::floor(lo) + 1.0; have = lo < k && k < hi; }
        }
        if (have && k == k && verify(j, [&](int64_t r) { return cmp3(ca[r], k); })) { e.kind = CExpr::CMPK; e.a = a; e.c = k; return true; }
      }
      return false;
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
      if (!ix_cols.empty() && try_mask(j, e)) { ex[(size_t)j] = e; return true; }
      if (!search) return false;
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
    if (search || !ix_cols.empty()) {
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
      case CExpr::CMPK: id = op2(RH_RIR_COMPARE, column(t, e.a), constant(e.c), dep); break;
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
  if (index_masks_on() && !fast) {
    // the two rewrites below are exact by construction; like every other rewrite of the loader they are also VERIFIED before they
    // are kept: every output of every row target, before against after, on up to 96 rows x 3 parameter vectors in extended precision
    const Program before = P;
    complete_scatter_terms(P, columns, nrows, kept);
    fold_select_sums(P, columns, nrows, kept);
    if (!same_outputs(before, P, columns, nrows, kept)) {
      if (rh::knob("RH_ROLL_WHY")) std::fprintf(stderr, "rainier-hip: the select rewrites of a strict build did not reproduce the original outputs: dropped\n");
      P = before;
    }
  }
  return true;
}

}  // namespace rh
