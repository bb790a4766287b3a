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
// Second job, for programs recognised this way as the reference's lowering: Model.observe's INITIAL CHUNK (1-8 observations in a
// row target of its own, core/Model.scala:84-96) is unrolled -- its rows substituted as constants and summed in row order -- so
// that the model is left with one streamed target.
#include <algorithm>
#include <atomic>
#include <cmath>
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

// Strict builds, after the masks of a Lookup's index column have been recognised (index masks, below): the gradient of table entry k
// is a sum of eq(index_s, low + k, g_s, 0) terms, one per Lookup of the target that reads the table (8 in a target written by
// Model.observe's split) -- except where the front end has folded one away: a mask that is -1 or +1 on every row of its slot is a
// constant to it, and Lookup(constant, [0, g, 0]) is 0.  Gather mode reads every entry's gradient as THE scatter of one value
// (emit.cpp detect_gather) and the strict roll needs the 8 slots to be the same expression (rollstrict.cpp), so the folded terms
// are written back where the data confirms the folding: no row of that index column selects the entry, hence the term adds
// +0.0 on every row.
void complete_scatter_terms(Program &P, const double *const *columns, const int64_t *nrows, const std::vector<uint32_t> &kept) {
  for (size_t t = 0; t < P.targets.size(); t++) {
    Target &T = P.targets[t];
    if (!T.n_cols) continue;
    auto local_col = [&](uint32_t id) { const Node &x = P.nodes[id]; return (x.op == RH_RIR_INPUT && x.input >= P.n_params && x.dep == t + 1) ? (int)(x.input - T.input_start) : -1; };
    struct Slot { uint32_t ix; int col; uint32_t g; bool have_g, bad; std::vector<char> present; };
    std::vector<Slot> slots;
    uint32_t first = 0; int32_t low = 0; size_t cnt = 0;
    bool consistent = true;
    const size_t n0 = P.nodes.size();
    for (size_t i = 0; i < n0; i++) {
      const Node &nd = P.nodes[i];
      if (nd.op != RH_RIR_LOOKUP || nd.table.size() <= 3 || local_col(nd.a) < 0) continue;
      const Node &t0 = P.nodes[nd.table[0]];
      if (!(t0.op == RH_RIR_INPUT && t0.input < P.n_params && t0.input + nd.table.size() == P.n_params)) continue;
      bool run = true;
      for (size_t k = 0; k < nd.table.size() && run; k++) { const Node &tk = P.nodes[nd.table[k]]; run = tk.op == RH_RIR_INPUT && tk.input == t0.input + k; }
      if (!run) continue;
      if (cnt && (first != t0.input || low != nd.low)) consistent = false;
      first = t0.input; low = nd.low; cnt = nd.table.size();
      bool known = false;
      for (const Slot &sl : slots) known = known || sl.ix == nd.a;
      if (!known) slots.push_back({nd.a, local_col(nd.a), 0, false, false, std::vector<char>(cnt, 0)});
    }
    if (slots.empty() || !consistent) continue;
    // the eq terms that are there: output of entry k, cut along its top-level additions
    std::vector<uint32_t> stack;
    for (size_t k = 0; k < cnt; k++) {
      stack.assign(1, T.outputs[1 + first + k]);
      for (int steps = 0; !stack.empty() && steps < 4096; steps++) {
        const Node &x = P.nodes[stack.back()]; stack.pop_back();
        if (x.op == RH_RIR_ADD) { stack.push_back(x.a); stack.push_back(x.b); continue; }
        if (x.op == RH_RIR_NOOP) { stack.push_back(x.a); continue; }
        if (x.op == RH_RIR_SEQ) { stack.push_back(x.b); continue; }
        if (x.op != RH_RIR_LOOKUP || x.low != -1 || x.table.size() != 3) continue;
        const Node &cm = P.nodes[x.a];
        if (cm.op != RH_RIR_COMPARE || P.nodes[cm.b].op != RH_RIR_CONST || P.nodes[cm.b].cval != (double)low + (double)k) continue;
        const Node &z0 = P.nodes[x.table[0]], &z2 = P.nodes[x.table[2]];
        if (!(z0.op == RH_RIR_CONST && z0.cval == 0.0 && z2.op == RH_RIR_CONST && z2.cval == 0.0)) continue;
        for (Slot &sl : slots) {
          if (sl.ix != cm.a) continue;
          sl.present[k] = 1;
          if (!sl.have_g) { sl.g = x.table[1]; sl.have_g = true; } else if (sl.g != x.table[1]) sl.bad = true;
        }
      }
    }
    uint32_t zero = 0xFFFFFFFFu;
    for (Slot &sl : slots) {
      if (!sl.have_g || sl.bad) continue;
      // which entries do the rows of this index column select?
      std::vector<char> selected(cnt, 0);
      const double *c = columns[kept[T.col0 + (uint32_t)sl.col]];
      for (int64_t r = 0; r < nrows[t]; r++) {
        const double v = c[r];
        const double kk = v - (double)low;
        if (kk >= 0.0 && kk < (double)cnt && kk == std::floor(kk)) selected[(size_t)kk] = 1;
        else if (!(v == v)) selected[0] = selected[0];   // (a NaN index compares as -1 with everything: selects nothing)
      }
      for (size_t k = 0; k < cnt; k++) {
        if (sl.present[k] || selected[k]) continue;
        if (zero == 0xFFFFFFFFu) { Node z; z.op = RH_RIR_CONST; z.cval = 0.0; P.nodes.push_back(z); zero = (uint32_t)P.nodes.size() - 1; }
        Node kc; kc.op = RH_RIR_CONST; kc.cval = (double)low + (double)k; P.nodes.push_back(kc);
        Node cm; cm.op = RH_RIR_COMPARE; cm.a = sl.ix; cm.b = (uint32_t)P.nodes.size() - 1; cm.dep = (uint32_t)t + 1; P.nodes.push_back(cm);
        Node lk; lk.op = RH_RIR_LOOKUP; lk.a = (uint32_t)P.nodes.size() - 1; lk.low = -1; lk.table = {zero, sl.g, zero}; lk.dep = (uint32_t)t + 1; P.nodes.push_back(lk);
        const uint32_t term = (uint32_t)P.nodes.size() - 1;
        uint32_t &out = T.outputs[1 + first + k];
        const Node &o = P.nodes[out];
        if (o.op == RH_RIR_CONST && o.cval == 0.0) out = term;
        else { Node ad; ad.op = RH_RIR_ADD; ad.a = out; ad.b = term; ad.dep = (uint32_t)t + 1; P.nodes.push_back(ad); out = (uint32_t)P.nodes.size() - 1; }
      }
    }
  }
}

}  // namespace

bool index_masks_on() {
  const char *e = std::getenv("RH_INDEX_MASKS");
  return e ? std::atoi(e) != 0 : false;
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
        if (a >= j) continue;
        const double *ca = col[(size_t)a];
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
  if (index_masks_on() && !fast) complete_scatter_terms(P, columns, nrows, kept);
  return true;
}

}  // namespace rh
