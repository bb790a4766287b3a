// emit.cpp -- IR -> HIP lowering.  Per target three straight-line fp64 functions:
//   invariants(th, inv, err)          : the sub-DAG that depends on parameters only (hoisted out of the row loop;
//                                       the reference re-evaluates it for every row inside its generated methods)
//   row(th, inv, c, acc, err)         : the per-row sub-DAG; acc[j] += basis term j
//   finish(th, inv, S, nrows, tot)    : tot[o] += (row-summed) output o, from the reduced basis sums S
// The whole DAG is evaluated once per row -- the reference's "outputs must be evaluated in index order so that a
// VarDef precedes its VarRefs" constraint disappears in a fused kernel.
//
// Default (JVM-faithful) mode: the basis terms ARE the outputs (acc[o] += output_o per row, as
// ir/DataFunction.scala:64-71 does), finish() just adds the sums.
// factor_outputs mode: every output is peeled into  alpha * t + beta  with alpha, beta parameter-only and t a
// row-dependent basis term (through +,-,*,/ by invariants and products of single-term forms).  The row loop then
// accumulates the distinct basis terms only and finish() applies  alpha * sum(t) + nrows * beta  once per gradient.
// For the README regression this turns 6 outputs x ~20 fp64 ops per row into 5 basis sums (r*r, r, r*x_k) x 9 ops.
// Rows are still streamed for every evaluation (no data-only sums are hoisted: every basis term involves the
// parameters or is consumed as a product with one that does); only rounding changes, so it is a fast-mode option.
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <map>
#include <sstream>
#include <tuple>

#include "../../include/rainier_hip_rir.h"
#include "rir.hpp"

namespace rh {
namespace {

constexpr uint32_t ZERO = 0xFFFFFFFFu;  // the invariant constant 0 (no node)
constexpr uint32_t ONE = 0xFFFFFFFEu;   // the invariant constant 1 (no node)
constexpr uint32_t NONE = 0xFFFFFFFDu;  // no basis term

std::string lit(double v) {
  if (std::isinf(v)) return v > 0 ? "RH_INF" : "(-RH_INF)";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%a", v);
  std::string s(buf);
  return v < 0 || std::signbit(v) ? "(" + s + ")" : s;
}

bool is_binary(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }

struct Lin {  // alpha * term + beta
  uint32_t term = NONE, alpha = ONE, beta = ZERO;
};

// ---- bounded live state (the analogue of the reference's method splitting, ir/Packer.scala:10-71) --------------------------------
// The reference's back end cuts every expression tree into methods of at most 200 nodes whose results travel through the JVM's
// locals and fields, so no model is too heavy for it.  Here a straight-line function of thousands of statements meets a register
// file: what does not fit is spilled by the compiler, and this toolchain's spill placement is not to be trusted.  In
// the memory-resident lowering (EmitOptions.chunk > 0) the body of a generated function is cut into CHUNKS of at most `chunk`
// statement groups, each in its own scope; a value that is used in a later chunk than the one that defines it is stored to a slot
// of a per-lane scratch array when it is defined and loaded from there -- under a fresh name -- by every later chunk that uses it.
// Every access goes through an opaque index (rh_oz(): a volatile asm that yields 0), so the array cannot be split into registers,
// nothing is forwarded from a store to a load, and a value's register dies with its chunk.  (A volatile array would say the same,
// but this toolchain mis-selects the flat accesses volatile private memory is left with: "Illegal instruction detected".)  The arithmetic and its order are untouched (the results are the same bits); the
// live state of the function is one chunk's temporaries, whatever the size of the model.  Slots are reused when a value's last
// reader has passed.
// `parts` (the data-free target's row(), round 5): beyond kPartChunks chunks the chunks are also grouped into PARTS, each a noinline
// function of its own (a static member of a local struct; theta, the accumulators and the scratch array reach it by pointer).  hiprtc's
// time grows faster than linearly in the size of a function -- a 701-parameter state-space prior is ONE function of 16 000 statements:
// 52 s -- and the chunks already pass every value that crosses them through memory, so cutting along chunk borders changes nothing
// but the unit the compiler's passes run on (the reference: ir/Packer.scala:10-71 splits methods at 200 nodes for the same reason).
static const size_t kPartChunks = 10;
static std::string chunk_body(const std::string &body, int chunk, const std::string &indent = "    ", bool parts = false) {
  std::vector<std::string> lines;
  { std::istringstream is(body); std::string ln; while (std::getline(is, ln)) lines.push_back(ln); }
  struct Group { std::vector<std::string> text; std::vector<long> defs, uses; };
  std::vector<Group> groups;
  std::string head;   // declarations that stay at function scope (casts to void, rh1)
  auto ids_of = [](const std::string &ln, std::vector<long> &out) {
    for (size_t i = 0; i + 1 < ln.size(); i++) {
      if (ln[i] != 'n' || !std::isdigit((unsigned char)ln[i + 1])) continue;
      if (i > 0 && (std::isalnum((unsigned char)ln[i - 1]) || ln[i - 1] == '_')) continue;
      size_t j = i + 1; long v = 0;
      while (j < ln.size() && std::isdigit((unsigned char)ln[j])) { v = v * 10 + (ln[j] - '0'); j++; }
      if (j < ln.size() && (std::isalnum((unsigned char)ln[j]) || ln[j] == '_')) continue;
      out.push_back(v);
      i = j - 1;
    }
  };
  for (size_t i = 0; i < lines.size(); i++) {
    const std::string &ln = lines[i];
    if (ln.find("(void)") != std::string::npos || ln.find("const double rh1 = ") != std::string::npos) { head += ln + "\n"; continue; }
    Group g;
    g.text.push_back(ln);
    if (ln.find("const int k") != std::string::npos) {            // a Lookup: index, (table,) select, range check
      while (i + 1 < lines.size() && lines[i].find("err = 1;") == std::string::npos) g.text.push_back(lines[++i]);
    } else if (ln.find("{ const double ix") != std::string::npos) {   // an eq-family accumulation: two lines inside braces
      while (i + 1 < lines.size() && lines[i].find('}') == std::string::npos) g.text.push_back(lines[++i]);
    }
    for (const std::string &t : g.text) {
      const size_t d = t.find("const double n");
      std::vector<long> all;
      ids_of(t, all);
      long def = -1;
      if (d != std::string::npos && t.find(" = ", d) != std::string::npos) { std::vector<long> dd; ids_of(t.substr(d, t.find(" = ", d) - d), dd); if (!dd.empty()) def = dd[0]; }
      if (def >= 0) g.defs.push_back(def);
      for (long v : all) if (v != def) g.uses.push_back(v);
    }
    groups.push_back(g);
  }
  const size_t nch = (groups.size() + (size_t)chunk - 1) / (size_t)chunk;
  if (nch <= 1) return body;
  std::map<long, size_t> def_chunk, last_use;
  for (size_t gi = 0; gi < groups.size(); gi++) {
    const size_t c = gi / (size_t)chunk;
    for (long v : groups[gi].defs) def_chunk[v] = c;
    for (long v : groups[gi].uses) { auto it = last_use.find(v); if (it == last_use.end() || it->second < c) last_use[v] = c; }
  }
  std::map<long, int> slot;
  std::vector<int> free_slots;
  int nslots = 0;
  std::vector<std::vector<long>> dies(nch);
  for (auto &kv : last_use) if (def_chunk.count(kv.first) && kv.second > def_chunk[kv.first]) dies[kv.second].push_back(kv.first);
  std::ostringstream os, out;
  bool split = parts && nch > kPartChunks;
  if (split) {
    // a part is a static member of a LOCAL struct: it sees its arguments (th, inv, c, acc, err, rh_sp) and nothing of the enclosing
    // function's scope.  The data-free row() this was written for references nothing else; a body that does -- rh1 of the fma
    // spelling, gz / sv of gather mode, the closed-form link's lk_* -- keeps its chunks in one function
    auto has_token = [](const std::string &ln, const char *tok) {
      const size_t L = std::strlen(tok);
      for (size_t i = ln.find(tok); i != std::string::npos; i = ln.find(tok, i + 1)) {
        const bool left = i == 0 || !(std::isalnum((unsigned char)ln[i - 1]) || ln[i - 1] == '_');
        const bool right = i + L >= ln.size() || !(std::isalnum((unsigned char)ln[i + L]) || ln[i + L] == '_');
        if (left && right) return true;
      }
      return false;
    };
    for (const Group &g : groups)
      for (const std::string &t : g.text)
        if (has_token(t, "rh1") || has_token(t, "gz") || has_token(t, "sv") || t.find("lk_") != std::string::npos) split = false;
  }
  size_t npart = 0;
  for (size_t c = 0; c < nch; c++) {
    if (split && c % kPartChunks == 0) {
      if (c == 0) os << "#ifndef RH_DEV_NOINLINE\n#define RH_DEV_NOINLINE __device__ __attribute__((noinline))\n#endif\n";
      os << indent << "struct rh_part" << npart << " { static RH_DEV_NOINLINE void run(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, "
            "rh_acc_t *acc, int &err, double *rh_sp) {\n" << indent << "(void)th; (void)inv; (void)c; (void)acc; (void)err; (void)rh_sp;\n";
    }
    os << indent << "{\n";
    std::vector<long> loads, stores;
    std::map<long, char> seen;
    for (size_t gi = c * (size_t)chunk; gi < std::min(groups.size(), (c + 1) * (size_t)chunk); gi++) {
      for (long v : groups[gi].uses) if (def_chunk.count(v) && def_chunk[v] < c && !seen[v]) { seen[v] = 1; loads.push_back(v); }
      for (long v : groups[gi].defs) if (last_use.count(v) && last_use[v] > c) stores.push_back(v);
    }
    for (long v : loads) os << indent << "const double n" << v << " = rh_sp[" << slot.at(v) << " + rh_oz()];\n";
    for (size_t gi = c * (size_t)chunk; gi < std::min(groups.size(), (c + 1) * (size_t)chunk); gi++) {
      for (const std::string &t : groups[gi].text) os << t << "\n";
      for (long v : groups[gi].defs)
        if (last_use.count(v) && last_use[v] > c) {
          int sl;
          if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); } else sl = nslots++;
          slot[v] = sl;
          os << indent << "rh_sp[" << sl << " + rh_oz()] = n" << v << ";\n";
        }
    }
    os << indent << "}\n";
    if (split && (c % kPartChunks == kPartChunks - 1 || c + 1 == nch)) {
      os << indent << "} };\n" << indent << "rh_part" << npart << "::run(th, inv, c, acc, err, rh_sp);\n";
      npart++;
    }
    for (long v : dies[c]) free_slots.push_back(slot.at(v));
  }
  out << head << indent << "double rh_sp[" << std::max(1, nslots) << "];\n" << os.str();
  return out.str();
}

struct TargetEmitter {
  Program P;  // private copy: factoring appends synthesized nodes
  uint32_t t;
  bool factor;
  bool fast_div = false;       // fast mode: x / const -> x * (1/const)
  bool fma_adds = false;       // per-row code: adds/subs as fma(x, +-1.0, y); with fast_div (= contraction allowed) mul+add is fused here
  // Round 6: where contraction is allowed (fast builds) the row code of a streamed target fuses its mul+add pairs HERE, explicitly, and
  // is compiled with contraction OFF.  The gradient kernels inline row() once per chain of the wavefront's group; left to the compiler
  // (`#pragma clang fp contract(fast)`), the K copies were contracted differently (rh_grad_gather_kernel, centred cfg 5: 22 / 22 / 22 /
  // 28 fused operations in the four copies of one tile), so a chain's sums depended by an ulp on the SLOT it was served in -- on which
  // other chains were live (tests/test_gpu_live_chains.py, case 3; tools/r6_live_diag.py).  One spelling, one rounding sequence,
  // whatever the slot, the kernel, or row() / row_g().  (Memory-resident lowerings run one chain per wavefront and keep the old form.)
  bool xfuse = false;
  uint32_t run_end = 0;        // data-free targets t..run_end are emitted together (shared sub-expressions once)
  bool merged_away = false;    // this data-free target was emitted by an earlier one of its run
  int chunk = 0;               // > 0: memory-resident lowering, at most this many statement groups per chunk (chunk_body)
  // memory-resident lowering: the accumulators and invariants are reached through an OPAQUE index (rh_oz() is a volatile asm that
  // yields 0), so the arrays cannot be split into registers and no load is forwarded from a store or hoisted out of the row loop
  std::string oz() const { return chunk > 0 ? " + rh_oz()" : ""; }
  // gather mode (models with a big parameter table indexed by a data column -- cfg 5):
  //   value side   : LOOKUP(index column, [theta_first .. theta_first+count-1], low)   -> `gz`, the gathered parameter
  //   gradient side: output(1 + first + k) = eq(index, low + k, Gv, 0) for every k (compute/Gradient.scala:148-152) -> one
  //                  scatter value Gv per row, summed per group by rh_grad_gather_kernel instead of `count` eq-lookups
  bool gmode = false;
  uint32_t n_shared = 0;       // parameters [0, n_shared) are ordinary; [n_shared, n_params) are the gathered table
  struct Gather { bool ok = false; uint32_t node = 0, sv = 0; int col = -1, first = 0, count = 0, low = 0; } gather;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;  // hash-consing of synthesized nodes
  std::map<uint32_t, Lin> memo;
  std::vector<uint32_t> basis;            // basis term node ids, accumulator order
  // outputs that are sums of several basis terms (alpha_1 t_1 + alpha_2 t_2 + ... + beta): accepted only when every t_i is
  // accumulated anyway for another output -- the gradient w.r.t. a hyper-parameter of a Lookup table, sum_k c_k eq(index, k, g, 0)
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> outs_multi;   // per output: (basis index, alpha); empty = single-term (outs)
  // scatter families: basis terms  eq(column, k, g, 0)  for many k with one g (what differentiating Lookup(column, table) leaves,
  // compute/Gradient.scala:148-152).  Their accumulators are laid out consecutively and the row code does ONE
  // acc[base + column - kmin] += g  instead of one select per table entry
  struct Family { uint32_t col = 0, g = 0; int kmin = 0, size = 0, base = 0; };
  std::vector<Family> families;
  std::vector<char> in_family;            // per basis index
  std::vector<Lin> outs;                  // per output: (basis index in .term, alpha, beta)
  std::map<uint32_t, int> inv_slot;       // non-trivial invariant node -> index in inv[]
  std::map<uint32_t, int> inv_table;      // row-level LOOKUP whose >= 8 entries are all non-trivial invariants -> first of their CONSECUTIVE inv[] slots
  std::vector<char> reach_row, reach_inv;
  // the constant pool of data-free targets (EmitInfo::kpool): value bits -> index, shared by the targets of a program
  std::vector<double> *kpool = nullptr;
  std::map<uint64_t, int> *kpool_index = nullptr;
  bool pooled() const { return kpool != nullptr && !has_rows(); }
  static bool structural(double v) {   // small dyadic numbers (and +-inf) stay literals: they are what a model's TEXT contains, and the compiler folds them
    return std::isinf(v) || (std::fabs(v) <= 4096.0 && v * 8.0 == std::nearbyint(v * 8.0));
  }
  std::string pool_ref(double v) const {
    uint64_t bits; std::memcpy(&bits, &v, sizeof bits);
    auto it = kpool_index->find(bits);
    if (it == kpool_index->end()) { it = kpool_index->emplace(bits, (int)kpool->size()).first; kpool->push_back(v); }
    return "c[" + std::to_string(it->second) + "]";
  }

  TargetEmitter(const Program &p, uint32_t ti, bool f) : P(p), t(ti), factor(f), run_end(ti) {
    for (uint32_t i = 0; i < P.nodes.size(); i++)  // let synthesized nodes reuse identical existing ones
      if (P.nodes[i].op >= RH_RIR_ADD && P.nodes[i].op <= RH_RIR_DIV) cons.emplace(std::make_tuple(P.nodes[i].op, P.nodes[i].a, P.nodes[i].b), i);
  }

  bool has_rows() const { return P.targets[t].n_cols > 0; }
  // Math.pow; in fast mode x^0.5 / x^-0.5 (the Translator's sqrt spellings) become one v_sqrt_f64-based call
  std::string pow_call(const std::string &x, uint32_t e_id, const std::string &e) const {
    if (fast_div && P.nodes[e_id].op == RH_RIR_CONST) {
      if (P.nodes[e_id].cval == 0.5) return "rh_pow_half(" + x + ")";
      if (P.nodes[e_id].cval == -0.5) return "(0x1p+0 / rh_pow_half(" + x + "))";
    }
    return "rh_java_pow(" + x + ", " + e + ")";
  }
  bool trivial(uint32_t id) const { return P.nodes[id].op == RH_RIR_CONST || P.nodes[id].op == RH_RIR_INPUT; }
  bool is_const(uint32_t id, double v) const { return id < P.nodes.size() && P.nodes[id].op == RH_RIR_CONST && P.nodes[id].cval == v; }

  uint32_t mk_const(double v) {
    for (uint32_t i = 0; i < P.nodes.size(); i++)
      if (P.nodes[i].op == RH_RIR_CONST && P.nodes[i].cval == v && std::signbit(P.nodes[i].cval) == std::signbit(v)) return i;
    Node n; n.op = RH_RIR_CONST; n.cval = v; n.dep = 0;
    P.nodes.push_back(n);
    return (uint32_t)P.nodes.size() - 1;
  }
  uint32_t mk(uint32_t op, uint32_t a, uint32_t b, uint32_t dep) {
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = dep;
    P.nodes.push_back(n);
    return cons[key] = (uint32_t)P.nodes.size() - 1;
  }
  // invariant arithmetic on {ZERO, ONE, node}
  uint32_t iadd(uint32_t a, uint32_t b) { if (a == ZERO) return b; if (b == ZERO) return a; return mk(RH_RIR_ADD, real(a), real(b), 0); }
  uint32_t isub(uint32_t a, uint32_t b) { if (b == ZERO) return a; return mk(RH_RIR_SUB, a == ZERO ? mk_const(0.0) : real(a), real(b), 0); }
  uint32_t imul(uint32_t a, uint32_t b) { if (a == ZERO || b == ZERO) return ZERO; if (a == ONE) return b; if (b == ONE) return a; return mk(RH_RIR_MUL, a, b, 0); }
  uint32_t idiv(uint32_t a, uint32_t b) { if (a == ZERO) return ZERO; if (b == ONE) return a; return mk(RH_RIR_DIV, real(a), b, 0); }
  uint32_t real(uint32_t a) { return a == ONE ? mk_const(1.0) : (a == ZERO ? mk_const(0.0) : a); }

  Lin opaque(uint32_t id) { Lin l; l.term = id; return l; }
  Lin lin(uint32_t id) {
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    Lin r = lin_impl(id);
    memo[id] = r;
    return r;
  }
  Lin lin_impl(uint32_t id) {
    const Node nd = P.nodes[id];
    if (nd.dep == 0) {
      Lin l; l.term = NONE; l.alpha = ZERO; l.beta = id;
      if (nd.op == RH_RIR_CONST && nd.cval == 0.0) l.beta = ZERO;
      if (nd.op == RH_RIR_CONST && nd.cval == 1.0) l.beta = ONE;
      return l;
    }
    switch (nd.op) {
      case RH_RIR_NOOP: return lin(nd.a);
      case RH_RIR_SEQ: return lin(nd.b);
      case RH_RIR_ADD: case RH_RIR_SUB: {
        const Lin a = lin(nd.a), b = lin(nd.b);
        const bool sub = nd.op == RH_RIR_SUB;
        Lin r;
        if (a.term != NONE && b.term != NONE) {
          if (a.term != b.term) return opaque(id);
          r.term = a.term; r.alpha = sub ? isub(a.alpha, b.alpha) : iadd(a.alpha, b.alpha);
        } else if (a.term != NONE) { r.term = a.term; r.alpha = a.alpha; }
        else { r.term = b.term; r.alpha = sub ? isub(ZERO, b.alpha) : b.alpha; }
        r.beta = sub ? isub(a.beta, b.beta) : iadd(a.beta, b.beta);
        return r;
      }
      case RH_RIR_MUL: {
        const Lin a = lin(nd.a), b = lin(nd.b);
        if (a.term == NONE) { Lin r; r.term = b.term; r.alpha = imul(b.alpha, a.beta); r.beta = imul(b.beta, a.beta); return r; }
        if (b.term == NONE) { Lin r; r.term = a.term; r.alpha = imul(a.alpha, b.beta); r.beta = imul(a.beta, b.beta); return r; }
        if (a.beta == ZERO && b.beta == ZERO) {
          Lin r;
          const uint32_t lo = std::min(a.term, b.term), hi = std::max(a.term, b.term);
          r.term = (a.alpha == ONE && b.alpha == ONE) ? id : mk(RH_RIR_MUL, lo, hi, (uint32_t)(t + 1));
          r.alpha = imul(a.alpha, b.alpha);
          return r;
        }
        return opaque(id);
      }
      case RH_RIR_DIV: {
        const Lin a = lin(nd.a), b = lin(nd.b);
        if (b.term == NONE && a.term != NONE) { Lin r; r.term = a.term; r.alpha = idiv(a.alpha, b.beta); r.beta = idiv(a.beta, b.beta); return r; }
        return opaque(id);
      }
      default: return opaque(id);
    }
  }

  // id = sum_i alpha_i * t_i + beta with single-term leaves t_i (through +, -, and * or / by invariants)
  bool lin_multi(uint32_t id, uint32_t scale, std::vector<std::pair<uint32_t, uint32_t>> &terms, uint32_t &beta, int depth = 0) {
    if (depth > 4096 || terms.size() > 4096) return false;
    const Node nd = P.nodes[id];
    if (nd.dep == 0) {
      uint32_t v = id;
      if (nd.op == RH_RIR_CONST && nd.cval == 0.0) v = ZERO;
      if (nd.op == RH_RIR_CONST && nd.cval == 1.0) v = ONE;
      beta = iadd(beta, imul(scale, v));
      return true;
    }
    switch (nd.op) {
      case RH_RIR_NOOP: return lin_multi(nd.a, scale, terms, beta, depth + 1);
      case RH_RIR_SEQ: return lin_multi(nd.b, scale, terms, beta, depth + 1);
      case RH_RIR_ADD: return lin_multi(nd.a, scale, terms, beta, depth + 1) && lin_multi(nd.b, scale, terms, beta, depth + 1);
      case RH_RIR_SUB: return lin_multi(nd.a, scale, terms, beta, depth + 1) && lin_multi(nd.b, isub(ZERO, scale), terms, beta, depth + 1);
      case RH_RIR_MUL:
        if (P.nodes[nd.a].dep == 0) return lin_multi(nd.b, imul(scale, lin(nd.a).beta), terms, beta, depth + 1);
        if (P.nodes[nd.b].dep == 0) return lin_multi(nd.a, imul(scale, lin(nd.b).beta), terms, beta, depth + 1);
        break;
      case RH_RIR_DIV:
        if (P.nodes[nd.b].dep == 0) return lin_multi(nd.a, idiv(scale, lin(nd.b).beta), terms, beta, depth + 1);
        break;
      default: break;
    }
    const Lin l = lin(id);
    if (l.term != NONE) terms.push_back({l.term, imul(scale, l.alpha)});
    beta = iadd(beta, imul(scale, l.beta));
    return true;
  }

  void operands(const Node &nd, std::vector<uint32_t> &out) const {
    out.clear();
    switch (nd.op) {
      case RH_RIR_CONST: case RH_RIR_INPUT: break;
      case RH_RIR_LOOKUP: out.push_back(nd.a); for (uint32_t e : nd.table) out.push_back(e); break;
      default: out.push_back(nd.a); if (is_binary(nd.op)) out.push_back(nd.b);
    }
  }
  void sweep(std::vector<char> &reach) const {
    std::vector<uint32_t> ops;
    for (size_t n = P.nodes.size(); n-- > 0;) {
      if (!reach[n]) continue;
      operands(P.nodes[n], ops);
      for (uint32_t o : ops) reach[o] = 1;
    }
  }

  bool is_param(uint32_t x) const { return P.nodes[x].op == RH_RIR_INPUT && P.nodes[x].input < P.n_params; }
  int col_index(uint32_t x) const {
    const Node &nd = P.nodes[x];
    return (nd.op == RH_RIR_INPUT && nd.input >= P.n_params) ? (int)(nd.input - P.targets[t].input_start) : -1;
  }
  // gather mode: find this target's gather node and check that every table parameter's gradient output is the
  // matching eq-lookup of one common scatter value
  bool detect_gather(std::string &err) {
    const Target &T = P.targets[t];
    const uint32_t ns = n_shared, cnt = P.n_params - n_shared;
    if (!has_rows()) {
      for (uint32_t k = 0; k < cnt; k++)
        if (!(is_const(T.outputs[1 + ns + k], 0.0))) { err = "gather mode: a data-free target depends on a table parameter"; return false; }
      return true;
    }
    std::vector<char> reach(P.nodes.size(), 0);
    for (uint32_t o = 0; o <= ns; o++) reach[T.outputs[o]] = 1;
    sweep(reach);
    for (size_t n = 0; n < P.nodes.size(); n++) {
      if (!reach[n]) continue;
      const Node &nd = P.nodes[n];
      if (nd.op == RH_RIR_LOOKUP && col_index(nd.a) >= 0 && nd.table.size() == cnt && is_param(nd.table[0]) && P.nodes[nd.table[0]].input == ns) {
        bool consecutive = true;
        for (uint32_t k = 0; k < cnt; k++) consecutive = consecutive && is_param(nd.table[k]) && P.nodes[nd.table[k]].input == ns + k;
        if (!consecutive) continue;
        if (gather.ok) { err = "gather mode: more than one gather per target"; return false; }
        gather.ok = true; gather.node = (uint32_t)n; gather.col = col_index(nd.a); gather.first = (int)ns; gather.count = (int)cnt; gather.low = nd.low;
      }
    }
    // table parameters must not be used directly by any value/shared-gradient node except through the table
    for (size_t n = 0; n < P.nodes.size(); n++) {
      if (!reach[n] || (gather.ok && n == gather.node)) continue;
      std::vector<uint32_t> ops; operands(P.nodes[n], ops);
      for (uint32_t o : ops) if (is_param(o) && P.nodes[o].input >= ns) { err = "gather mode: a table parameter is used outside the gather"; return false; }
    }
    bool have_sv = false;
    for (uint32_t k = 0; k < cnt; k++) {
      uint32_t oid = T.outputs[1 + ns + k];
      // (the front end's wrappers around a whole output: a NOOP marker, or the Translator's SEQ(definition, use) whose value is `use`
      //  -- the definition is a sub-expression of the scatter value, evaluated with it)
      while (index_masks_on() && (P.nodes[oid].op == RH_RIR_NOOP || P.nodes[oid].op == RH_RIR_SEQ)) oid = P.nodes[oid].op == RH_RIR_NOOP ? P.nodes[oid].a : P.nodes[oid].b;
      if (!gather.ok) { if (!is_const(oid, 0.0)) { err = "gather mode: table gradient without a gather"; return false; } continue; }
      const Node &on = P.nodes[oid];
      if (index_masks_on() && is_const(oid, 0.0)) {
        // an entry that no row of this target selects (Model.observe's initial chunk has at most 8 rows): the front end has folded its
        // all-(-1) / all-(+1) mask and with it the gradient to 0, which is what the scatter adds to that entry -- accepted when the
        // data says so (the index column's distinct values are known and low + k is not among them)
        const size_t gc = (size_t)T.col0 + (size_t)gather.col;
        bool absent = gc < P.col_domain.size() && !P.col_domain[gc].empty();
        for (size_t i = 0; absent && i < P.col_domain[gc].size(); i++) absent = P.col_domain[gc][i] != (double)(gather.low + (int)k);
        if (absent) continue;
      }
      bool ok = on.op == RH_RIR_LOOKUP && on.low == -1 && on.table.size() == 3 && is_const(on.table[0], 0.0) && is_const(on.table[2], 0.0);
      if (ok) {
        const Node &cmp = P.nodes[on.a];
        ok = cmp.op == RH_RIR_COMPARE && cmp.a == P.nodes[gather.node].a && P.nodes[cmp.b].op == RH_RIR_CONST &&
             P.nodes[cmp.b].cval == (double)(gather.low + (int)k);
      }
      if (ok) { if (!have_sv) { gather.sv = on.table[1]; have_sv = true; } else ok = on.table[1] == gather.sv; }
      if (!ok) { err = "gather mode: gradient output of table parameter " + std::to_string(k) + " is not eq(index, k, g, 0)"; return false; }
    }
    if (gather.ok && !have_sv) { err = "gather mode: no table gradient in a target that reads the table"; return false; }
    return true;
  }

  // what the row code must compute: the basis terms, with every scatter family replaced by its scatter value and index column
  std::vector<uint32_t> row_roots() const {
    std::vector<uint32_t> r;
    for (size_t j = 0; j < basis.size(); j++) if (basis[j] != NONE && !(j < in_family.size() && in_family[j])) r.push_back(basis[j]);
    for (const Family &f : families) { r.push_back(f.g); r.push_back(f.col); }
    return r;
  }
  // ---- gradient-only row code (round 6) -------------------------------------------------------------------------------------
  // LeapFrog.twoFullSteps (sampler/LeapFrog.scala:175-184) overwrites pqBuf(potentialIndex) at every step of a trajectory and
  // only the last step's value is read (finishIteration's energy, :61-82; EHMC's snapshot at step minSteps, EHMC.scala:38-41, is
  // taken through takeSteps(1), whose only step is its last).  For a mid-trajectory gradient request the tick engine therefore
  // runs row_g(): row() without the basis terms that only output 0 -- the log-density -- needs, and without what only they
  // reach.  Their accumulators stay 0; every other sum is computed by the same statements in the same order (same bits).
  std::vector<char> value_only() const {   // per basis index
    std::vector<char> v(basis.size(), 1);
    for (size_t o = 1; o < outs.size(); o++) {
      if (outs[o].term != NONE && outs[o].alpha != ZERO) v[outs[o].term] = 0;
      for (auto &tm : outs_multi[o]) if (tm.second != ZERO) v[tm.first] = 0;
    }
    for (size_t j = 0; j < basis.size(); j++) if ((j < in_family.size() && in_family[j]) || basis[j] == NONE) v[j] = 0;
    return v;
  }
  // liveness of the row code from `roots` (+ the scatter value in gather mode), with the closed-form link's value and cores as leaves
  std::vector<char> reach_from(const std::vector<uint32_t> &roots) const {
    std::vector<char> r(P.nodes.size(), 0);
    for (uint32_t b : roots) r[b] = 1;
    if (gather.ok) r[gather.sv] = 1;
    if (!link.ok) { sweep(r); return r; }
    std::vector<uint32_t> ops;
    bool any_link = false;
    for (size_t n = P.nodes.size(); n-- > 0;) {
      if (!r[n]) continue;
      if (n == link.V) { for (auto &c : link.cterms) r[c.first] = 1; any_link = true; continue; }
      if (link.cores.count((uint32_t)n)) { any_link = true; continue; }
      operands(P.nodes[n], ops);
      for (uint32_t o : ops) r[o] = 1;
    }
    if (any_link) {   // the prelude: t from L, the alpha / beta sums
      std::vector<char> pre(P.nodes.size(), 0);
      pre[link.L] = 1;
      for (auto &c : link.alpha) if (c.first < NONE) pre[c.first] = 1;
      for (auto &c : link.beta) if (c.first < NONE) pre[c.first] = 1;
      sweep(pre);
      for (size_t n = 0; n < P.nodes.size(); n++) if (pre[n]) r[n] = 1;
    }
    return r;
  }
  std::vector<uint32_t> row_roots_g() const {   // row_roots() without the value-only basis terms
    const std::vector<char> vo = value_only();
    std::vector<uint32_t> r;
    for (size_t j = 0; j < basis.size(); j++) if (basis[j] != NONE && !(j < in_family.size() && in_family[j]) && !vo[j]) r.push_back(basis[j]);
    for (const Family &f : families) { r.push_back(f.g); r.push_back(f.col); }
    return r;
  }
  void find_families() {
    in_family.assign(basis.size(), 0);
    families.clear();
    // (strict builds accumulate every output itself; with RH_INDEX_MASKS their eq(column, k, g, 0) outputs -- the reference's masks,
    //  recognised by the loader -- form families too: `acc[k] += g` on the selected row instead of `+= g or +0.0` on every accumulator)
    if (!has_rows() || (!factor && !index_masks_on()) || gmode) return;
    std::map<std::pair<uint32_t, uint32_t>, std::map<int, int>> cand;   // (column node, g) -> k -> basis index
    for (size_t j = 0; j < basis.size(); j++) {
      const Node &nd = P.nodes[basis[j]];
      if (nd.op != RH_RIR_LOOKUP || nd.low != -1 || nd.table.size() != 3 || nd.table[0] != nd.table[2] || !is_const(nd.table[0], 0.0)) continue;
      const Node &cm = P.nodes[nd.a];
      if (cm.op != RH_RIR_COMPARE || col_index(cm.a) < 0 || P.nodes[cm.b].op != RH_RIR_CONST) continue;
      const double kv = P.nodes[cm.b].cval;
      if (!(kv == std::floor(kv)) || std::fabs(kv) > 1e9) continue;
      cand[{cm.a, nd.table[1]}][(int)kv] = (int)j;
    }
    std::vector<char> moved(basis.size(), 0);
    std::vector<std::pair<Family, std::map<int, int>>> fams;
    for (auto &kv : cand) {
      const int kmin = kv.second.begin()->first, kmax = kv.second.rbegin()->first;
      if (kv.second.size() < 8 || (size_t)(kmax - kmin + 1) > 2 * kv.second.size()) continue;
      Family f; f.col = kv.first.first; f.g = kv.first.second; f.kmin = kmin; f.size = kmax - kmin + 1;
      fams.push_back({f, kv.second});
      for (auto &m : kv.second) moved[(size_t)m.second] = 1;
    }
    if (fams.empty()) return;
    // new accumulator order: everything else first (old order), then one block per family (gaps keep an unused slot)
    std::vector<uint32_t> nb;
    std::vector<int> renum(basis.size(), -1);
    for (size_t j = 0; j < basis.size(); j++) if (!moved[j]) { renum[j] = (int)nb.size(); nb.push_back(basis[j]); }
    std::vector<char> fam(nb.size(), 0);
    for (auto &fm : fams) {
      fm.first.base = (int)nb.size();
      for (int k = 0; k < fm.first.size; k++) {
        auto it = fm.second.find(fm.first.kmin + k);
        if (it != fm.second.end()) { renum[(size_t)it->second] = (int)nb.size(); nb.push_back(basis[(size_t)it->second]); } else nb.push_back(NONE);
        fam.push_back(1);
      }
      families.push_back(fm.first);
    }
    basis.swap(nb);
    in_family = fam;
    for (Lin &l : outs) if (l.term != NONE) l.term = (uint32_t)renum[l.term];
    for (auto &mo : outs_multi) for (auto &tm : mo) tm.first = (uint32_t)renum[tm.first];
  }
  // synthesized nodes are appended after their operands, so ascending id stays a topological order
  void plan() {
    const Target &T = P.targets[t];
    const size_t n_out = gmode ? (size_t)n_shared + 1 : T.outputs.size();
    outs.resize(n_out);
    std::map<uint32_t, int> bidx;
    auto basis_of = [&](uint32_t term) { auto it = bidx.find(term); if (it != bidx.end()) return it->second; int i = (int)basis.size(); basis.push_back(term); bidx[term] = i; return i; };
    outs_multi.assign(n_out, {});
    std::vector<Lin> first(n_out);
    std::map<uint32_t, char> single_terms;   // terms some output accumulates on its own
    auto sum_like = [&](const Lin &l) { return l.term != NONE && (P.nodes[l.term].op == RH_RIR_ADD || P.nodes[l.term].op == RH_RIR_SUB); };
    for (size_t o = 0; o < n_out; o++) {
      const uint32_t id = T.outputs[o];
      Lin l;
      if (!has_rows()) { l.term = NONE; l.alpha = ZERO; l.beta = id; }
      else if (factor) l = lin(id);
      else if (is_const(id, 0.0) && !std::signbit(P.nodes[id].cval)) { l.term = NONE; l.alpha = ZERO; l.beta = ZERO; }  // += +0.0 is the identity
      else l = opaque(id);  // JVM-faithful: accumulate the output itself (also when it is parameter-only)
      first[o] = l;
      if (has_rows() && factor && l.term != NONE && !sum_like(l)) single_terms[l.term] = 1;
    }
    for (size_t o = 0; o < n_out; o++) {
      Lin l = first[o];
      if (has_rows() && factor && sum_like(l)) {
        std::vector<std::pair<uint32_t, uint32_t>> terms;
        uint32_t beta = l.beta;
        bool ok = lin_multi(l.term, l.alpha, terms, beta) && terms.size() >= 2;
        for (auto &tm : terms) ok = ok && single_terms.count(tm.first);
        if (ok) {
          std::map<uint32_t, uint32_t> merged;   // one entry per basis term
          std::vector<uint32_t> order;
          for (auto &tm : terms) { auto it = merged.find(tm.first); if (it == merged.end()) { merged[tm.first] = tm.second; order.push_back(tm.first); } else it->second = iadd(it->second, tm.second); }
          for (uint32_t tn : order) outs_multi[o].push_back({(uint32_t)basis_of(tn), merged[tn]});
          Lin r; r.term = NONE; r.alpha = ZERO; r.beta = beta;
          outs[o] = r;
          continue;
        }
      }
      if (l.term != NONE) l.term = (uint32_t)basis_of(l.term);
      outs[o] = l;
    }
    find_families();
    reach_row.assign(P.nodes.size(), 0);
    reach_inv.assign(P.nodes.size(), 0);
    if (has_rows()) {
      for (uint32_t b : row_roots()) reach_row[b] = 1;
      if (gather.ok) reach_row[gather.sv] = 1;
      sweep(reach_row);
      // invariants needed: parameter-only operands of row nodes, plus every alpha / beta
      std::vector<uint32_t> ops;
      auto want = [&](uint32_t x) {
        if (x >= NONE) return;
        if (P.nodes[x].dep == 0 && !trivial(x) && !inv_slot.count(x)) { int s = (int)inv_slot.size(); inv_slot[x] = s; }
      };
      // parameter-only tables of a row-level Lookup (alphas(site) with alphas_k = mu + z_k sigma: the reference's GLMMs) get
      // consecutive slots, so that the row code reads inv[first + k] instead of building the table per row or walking a select chain
      for (size_t n = 0; n < P.nodes.size(); n++) {
        const Node &nd = P.nodes[n];
        if (!reach_row[n] || nd.dep == 0 || nd.op != RH_RIR_LOOKUP || nd.table.size() < 8 || (gather.ok && n == gather.node)) continue;
        bool all_inv = true, fresh = true;
        std::map<uint32_t, char> distinct;
        for (uint32_t e : nd.table) { all_inv = all_inv && P.nodes[e].dep == 0 && !trivial(e); fresh = fresh && !inv_slot.count(e); distinct[e] = 1; }
        if (!all_inv || !fresh || distinct.size() != nd.table.size()) continue;
        inv_table[(uint32_t)n] = (int)inv_slot.size();
        for (uint32_t e : nd.table) { const int sl = (int)inv_slot.size(); inv_slot[e] = sl; }
      }
      for (size_t n = 0; n < P.nodes.size(); n++) {
        if (!reach_row[n]) continue;
        if (P.nodes[n].dep == 0) {   // a parameter-only row root: a basis term, or the scatter value of a table that enters with a parameter-only adjoint
          const std::vector<uint32_t> rr = row_roots();
          if (std::find(rr.begin(), rr.end(), (uint32_t)n) != rr.end() || (gather.ok && n == gather.sv)) want((uint32_t)n);
          continue;
        }
        operands(P.nodes[n], ops);
        for (uint32_t o : ops) want(o);
      }
      for (const Lin &l : outs) { want(l.alpha); want(l.beta); }
      for (auto &mo : outs_multi) for (auto &tm : mo) want(tm.second);
      for (auto &kv : inv_slot) reach_inv[kv.first] = 1;
      sweep(reach_inv);
    } else {
      for (uint32_t tt = t; tt <= run_end; tt++)
        for (size_t o = 0; o < (gmode ? (size_t)n_shared + 1 : P.targets[tt].outputs.size()); o++) reach_row[P.targets[tt].outputs[o]] = 1;
      sweep(reach_row);
    }
  }

  // ---- fast mode: the logit family of streamed likelihood terms in closed form (cfg 5's negative binomial) -------------------
  // core/Discrete.scala writes NegativeBinomial / Binomial / Geometric densities as  C + alpha log(1 - p) + beta log p  with the
  // caller's p; with p = 1 / (1 + m e^{-L}) (a logistic link, m > 0 constant: cfg 5 has m = n) symbolic differentiation leaves
  // one exp, two logs and three divisions per row-chain evaluation.  With t = log m - L:
  //       log p = -softplus(t),  log(1 - p) = t - softplus(t)
  //       value = C + alpha t - (alpha + beta) softplus(t),      g = d value / d L = (alpha + beta) sigmoid(t) - alpha
  // The VALUE is matched structurally (it is the distribution's own formula, not a differentiated one): a sum whose E-dependent
  // terms are coef * log(P) / coef * log(1 - P) with P = 1 / (m E + 1), E = exp(-L) the target's only exponential.  Every OTHER
  // E-dependent quantity the row needs (the adjoint in whatever association the differentiation produced it) is peeled of its
  // E-independent factors down to a "core" and each core is VERIFIED numerically to equal kappa * g with a constant kappa
  // (9 values of L x 6 random assignments of all parameters / columns, 1e-8).  Only if the value and every core verify, and
  // nothing else still needs E, are those nodes emitted in closed form (one rh_logit_link per evaluation); otherwise the
  // literal lowering stays.  Fast mode only.
  struct Link {
    bool ok = false;
    uint32_t L = 0, E = 0, V = 0;
    double logm = 0.0;
    std::vector<std::pair<uint32_t, double>> alpha, beta;   // sum of (coefficient node or ONE, sign): value terms on log(1-P) / log P
    std::vector<std::pair<uint32_t, double>> cterms;        // E-independent terms of the value (node, sign)
    std::map<uint32_t, double> cores;                       // core node -> kappa
    uint32_t first = 0;                                     // smallest overridden node id: the prelude is emitted before it
  } link;

  double host_eval2(uint32_t id, const std::vector<double> &inputs, double gz, std::map<uint32_t, double> &memo, bool &bad) const {
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    if (gather.ok && id == gather.node) return memo[id] = gz;
    const Node &nd = P.nodes[id];
    auto E = [&](uint32_t x) { return host_eval2(x, inputs, gz, memo, bad); };
    double v = 0.0;
    switch (nd.op) {
      case RH_RIR_CONST: v = nd.cval; break;
      case RH_RIR_INPUT: v = inputs[nd.input]; break;
      case RH_RIR_ADD: v = E(nd.a) + E(nd.b); break;
      case RH_RIR_SUB: v = E(nd.a) - E(nd.b); break;
      case RH_RIR_MUL: v = E(nd.a) * E(nd.b); break;
      case RH_RIR_DIV: v = E(nd.a) / E(nd.b); break;
      case RH_RIR_POW: v = std::pow(E(nd.a), E(nd.b)); break;
      case RH_RIR_COMPARE: { const double l = E(nd.a), r = E(nd.b); v = l > r ? 1.0 : (l == r ? 0.0 : -1.0); break; }
      case RH_RIR_EXP: v = std::exp(E(nd.a)); break;
      case RH_RIR_LOG: v = std::log(E(nd.a)); break;
      case RH_RIR_ABS: v = std::fabs(E(nd.a)); break;
      case RH_RIR_NOOP: v = E(nd.a); break;
      case RH_RIR_SEQ: (void)E(nd.a); v = E(nd.b); break;
      case RH_RIR_LOOKUP: {
        const double ix = E(nd.a);
        const long long k = (ix != ix ? 0LL : (long long)ix) - (long long)nd.low;
        if (k < 0 || k >= (long long)nd.table.size()) { bad = true; v = 0.0; } else v = E(nd.table[(size_t)k]);
        break;
      }
      default: bad = true;
    }
    return memo[id] = v;
  }

  void detect_link() {
    if (!has_rows() || !fast_div || !factor || basis.empty()) return;
    const size_t N = P.nodes.size();
    // the target's only exponential, of the form exp(-L)
    uint32_t E = NONE;
    for (size_t n = 0; n < N; n++)
      if (reach_row[n] && P.nodes[n].dep != 0 && P.nodes[n].op == RH_RIR_EXP) { if (E != NONE) return; E = (uint32_t)n; }
    if (E == NONE) return;
    uint32_t L = NONE;
    { const Node &a = P.nodes[P.nodes[E].a];
      if (a.op == RH_RIR_MUL && is_const(a.b, -1.0)) L = a.a;
      else if (a.op == RH_RIR_MUL && is_const(a.a, -1.0)) L = a.b;
      else if (a.op == RH_RIR_SUB && is_const(a.a, 0.0)) L = a.b;
      else return; }
    std::vector<char> depE(N, 0);
    depE[E] = 1;
    { std::vector<uint32_t> ops;
      for (size_t n = E + 1; n < N; n++) { operands(P.nodes[n], ops); for (uint32_t o : ops) if (depE[o]) { depE[n] = 1; break; } } }
    // P = 1 / (m E + 1), Q = 1 - P
    auto match_D = [&](uint32_t d, double &m) -> bool {   // d == m E + 1
      const Node &nd = P.nodes[d];
      if (nd.op != RH_RIR_ADD) return false;
      for (int sw = 0; sw < 2; sw++) {
        const uint32_t x = sw ? nd.b : nd.a, one = sw ? nd.a : nd.b;
        if (!is_const(one, 1.0)) continue;
        if (x == E) { m = 1.0; return true; }
        const Node &xn = P.nodes[x];
        if (xn.op == RH_RIR_MUL) {
          if (xn.a == E && P.nodes[xn.b].op == RH_RIR_CONST && P.nodes[xn.b].cval > 0 && std::isfinite(P.nodes[xn.b].cval)) { m = P.nodes[xn.b].cval; return true; }
          if (xn.b == E && P.nodes[xn.a].op == RH_RIR_CONST && P.nodes[xn.a].cval > 0 && std::isfinite(P.nodes[xn.a].cval)) { m = P.nodes[xn.a].cval; return true; }
        }
      }
      return false;
    };
    double m = 0.0;
    auto is_P = [&](uint32_t x) { const Node &nd = P.nodes[x]; double mm; if (nd.op == RH_RIR_DIV && is_const(nd.a, 1.0) && match_D(nd.b, mm)) { if (m == 0.0) m = mm; return mm == m; } return false; };
    auto is_Q = [&](uint32_t x) { const Node &nd = P.nodes[x]; return nd.op == RH_RIR_SUB && is_const(nd.a, 1.0) && is_P(nd.b); };
    // the value: the one basis term whose E-dependent summands are all coef * log(P | Q)
    Link lk;
    bool found = false;
    for (uint32_t b : row_roots()) {
      if (!depE[b]) continue;
      Link cand; cand.L = L; cand.E = E; cand.V = b;
      bool good = true, any = false;
      std::vector<std::pair<uint32_t, double>> stack{{b, 1.0}};
      while (!stack.empty() && good) {
        auto [x, sg] = stack.back(); stack.pop_back();
        const Node &nd = P.nodes[x];
        if (!depE[x]) { cand.cterms.push_back({x, sg}); continue; }
        if (nd.op == RH_RIR_ADD) { stack.push_back({nd.b, sg}); stack.push_back({nd.a, sg}); continue; }
        if (nd.op == RH_RIR_SUB) { stack.push_back({nd.b, -sg}); stack.push_back({nd.a, sg}); continue; }
        uint32_t lg = NONE, coef = ONE;
        if (nd.op == RH_RIR_LOG) lg = x;
        else if (nd.op == RH_RIR_MUL && P.nodes[nd.a].op == RH_RIR_LOG && depE[nd.a] && !depE[nd.b]) { lg = nd.a; coef = nd.b; }
        else if (nd.op == RH_RIR_MUL && P.nodes[nd.b].op == RH_RIR_LOG && depE[nd.b] && !depE[nd.a]) { lg = nd.b; coef = nd.a; }
        if (lg == NONE) { good = false; break; }
        const uint32_t arg = P.nodes[lg].a;
        if (is_P(arg)) cand.beta.push_back({coef, sg});
        else if (is_Q(arg)) cand.alpha.push_back({coef, sg});
        else { good = false; break; }
        any = true;
      }
      if (good && any) { if (found) return; lk = cand; found = true; }
    }
    if (!found || m <= 0.0) return;
    lk.logm = std::log(m);
    // every other E-dependent root -> its core (peel E-independent factors and negations)
    std::vector<uint32_t> roots;
    for (uint32_t b : row_roots()) if (b != lk.V && depE[b]) roots.push_back(b);
    if (gather.ok && depE[gather.sv]) roots.push_back(gather.sv);
    std::vector<char> above(N, 0);          // E-dependent nodes that stay (products of a core with E-independent factors)
    for (uint32_t r : roots) {
      uint32_t x = r;
      for (;;) {
        const Node &nd = P.nodes[x];
        if (nd.op == RH_RIR_MUL && depE[nd.a] != depE[nd.b]) { above[x] = 1; x = depE[nd.a] ? nd.a : nd.b; continue; }
        if (nd.op == RH_RIR_SUB && is_const(nd.a, 0.0) && depE[nd.b]) { above[x] = 1; x = nd.b; continue; }
        break;
      }
      if (x == lk.V) return;
      lk.cores[x] = 0.0;
    }
    // nothing else may need E once V and the cores are closed forms
    { std::vector<char> need(N, 0);
      for (uint32_t b : row_roots()) need[b] = 1;
      if (gather.ok) need[gather.sv] = 1;
      std::vector<uint32_t> ops;
      for (size_t n = N; n-- > 0;) {
        if (!need[n]) continue;
        if (n == lk.V) { for (auto &c : lk.cterms) need[c.first] = 1; for (auto &c : lk.alpha) if (c.first < NONE) need[c.first] = 1; for (auto &c : lk.beta) if (c.first < NONE) need[c.first] = 1; need[L] = 1; continue; }
        if (lk.cores.count((uint32_t)n)) continue;
        operands(P.nodes[n], ops);
        for (uint32_t o : ops) need[o] = 1;
      }
      for (size_t n = 0; n < N; n++) if (need[n] && depE[n] && n != lk.V && !lk.cores.count((uint32_t)n) && !above[n]) return;
      if (need[E]) return; }
    // the prelude (L, alpha, beta nodes) must be defined before the first overridden node
    lk.first = lk.V;
    for (auto &c : lk.cores) lk.first = std::min(lk.first, c.first);
    auto before = [&](uint32_t x) { return x >= NONE || trivial(x) || P.nodes[x].dep == 0 || x < lk.first; };
    if (!before(L)) return;
    for (auto &c : lk.alpha) if (!before(c.first)) return;
    for (auto &c : lk.beta) if (!before(c.first)) return;
    // numerical verification
    auto softplus = [](double x) { return x > 0 ? x + std::log1p(std::exp(-x)) : std::log1p(std::exp(x)); };
    auto sigmoid = [](double x) { return x >= 0 ? 1.0 / (1.0 + std::exp(-x)) : std::exp(x) / (1.0 + std::exp(x)); };
    static const double Ls[] = {-6.0, -3.1, -1.3, -0.4, 0.2, 0.9, 2.1, 4.4, 7.0};
    unsigned long long rs = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; };
    // pass 0 estimates every core's kappa = core / g (mean over the probes, then snapped to 13 significant digits: the true
    // factors are small rationals like -1 or 1/10 and one quotient carries rounding noise); pass 1 verifies value and cores
    std::map<uint32_t, std::pair<double, int>> acc;
    for (int pass = 0; pass < 2; pass++) {
      rs = 0x9E3779B97F4A7C15ULL;
      for (int probe = 0; probe < 6; probe++) {
        std::vector<double> inputs(P.n_inputs);
        for (uint32_t i = 0; i < P.n_inputs; i++) inputs[i] = i < P.n_params ? (rnd() - 0.5) * 1.6 : (probe % 2 ? std::floor(rnd() * 6.0) : rnd() * 3.0 + 0.1);
        const double gz = (rnd() - 0.5) * 1.2;
        for (double Lv : Ls) {
          std::map<uint32_t, double> memo; bool bad = false;
          memo[L] = Lv + lk.logm;                       // t = log m - L = -Lv
          auto sum = [&](const std::vector<std::pair<uint32_t, double>> &ts) { double a = 0; for (auto &c : ts) a += c.second * (c.first == ONE ? 1.0 : host_eval2(c.first, inputs, gz, memo, bad)); return a; };
          const double al = sum(lk.alpha), be = sum(lk.beta), cc = sum(lk.cterms);
          const double tt = -Lv, cv = cc + al * tt - (al + be) * softplus(tt), g = (al + be) * sigmoid(tt) - al;
          const double v0 = host_eval2(lk.V, inputs, gz, memo, bad);
          if (bad || !(std::fabs(v0 - cv) <= 1e-8 * std::max(1.0, std::fabs(cv)))) return;
          for (auto &c : lk.cores) {
            const double k0 = host_eval2(c.first, inputs, gz, memo, bad);
            if (bad) return;
            if (pass == 0) { if (std::fabs(g) > 1e-2) { acc[c.first].first += k0 / g; acc[c.first].second++; } }
            else if (!(std::fabs(k0 - c.second * g) <= 1e-8 * std::max(1.0, std::fabs(c.second * g)))) return;
          }
        }
      }
      if (pass == 0)
        for (auto &c : lk.cores) {
          auto &a = acc[c.first];
          if (a.second < 8) return;
          double k = a.first / a.second;
          if (!std::isfinite(k) || k == 0.0) return;
          char buf[64]; std::snprintf(buf, sizeof buf, "%.12e", k);
          const double snapped = std::strtod(buf, nullptr);
          c.second = std::fabs(snapped - k) <= 1e-12 * std::fabs(k) ? snapped : k;
        }
    }
    lk.ok = true;
    link = lk;
    // liveness again, with V and the cores as leaves
    reach_row.assign(N, 0);
    for (uint32_t b : row_roots()) reach_row[b] = 1;
    if (gather.ok) reach_row[gather.sv] = 1;
    { std::vector<uint32_t> ops;
      for (size_t n = N; n-- > 0;) {
        if (!reach_row[n]) continue;
        if (n == link.V) { for (auto &c : link.cterms) reach_row[c.first] = 1; for (auto &c : link.alpha) if (c.first < NONE) reach_row[c.first] = 1; for (auto &c : link.beta) if (c.first < NONE) reach_row[c.first] = 1; reach_row[L] = 1; continue; }
        if (link.cores.count((uint32_t)n)) continue;
        operands(P.nodes[n], ops);
        for (uint32_t o : ops) reach_row[o] = 1;
      } }
    // invariants: recompute the slots from the new liveness
    inv_slot.clear();
    { std::vector<uint32_t> ops;
      auto want = [&](uint32_t x) { if (x >= NONE) return; if (P.nodes[x].dep == 0 && !trivial(x) && !inv_slot.count(x)) { int sl = (int)inv_slot.size(); inv_slot[x] = sl; } };
      for (size_t n = 0; n < N; n++) {
        if (!reach_row[n]) continue;
        if (P.nodes[n].dep == 0) { want((uint32_t)n); continue; }
        if (n == link.V || link.cores.count((uint32_t)n)) continue;
        operands(P.nodes[n], ops);
        for (uint32_t o : ops) want(o);
      }
      for (const Lin &l : outs) { want(l.alpha); want(l.beta); }
      for (auto &mo : outs_multi) for (auto &tm : mo) want(tm.second);
      reach_inv.assign(N, 0);
      for (auto &kv : inv_slot) reach_inv[kv.first] = 1;
      sweep(reach_inv); }
  }

  std::string link_sum(const std::vector<std::pair<uint32_t, double>> &ts) const {
    std::string e;
    for (auto &c : ts) {
      const std::string term = c.first == ONE ? "0x1p+0" : ref(c.first, 1);
      e += (c.second < 0 ? " - " : (e.empty() ? "" : " + ")) + term;
    }
    return e.empty() ? "0x0p+0" : "(" + e + ")";
  }
  void emit_link_prelude(std::ostringstream &os) const {
    os << "    const double lk_t = " << lit(link.logm) << " - " << ref(link.L, 1) << ";\n"
       << "    double lk_sp, lk_sg;\n    rh_logit_link(lk_t, lk_sp, lk_sg);\n"
       << "    const double lk_a = " << link_sum(link.alpha) << ", lk_ab = lk_a + " << link_sum(link.beta) << ";\n"
       << (xfuse_row(1) ? "    const double lk_g = __builtin_fma(lk_ab, lk_sg, -lk_a);\n" : "    const double lk_g = lk_ab * lk_sg - lk_a;\n");
  }

  // how an operand is spelled: ctx 0 = invariants(), 1 = row(), 2 = finish()
  std::string ref(uint32_t id, int ctx) const {
    if (id == ONE) return "0x1p+0";
    if (id == ZERO) return "0x0p+0";
    if (gather.ok && id == gather.node) return "gz";
    const Node &nd = P.nodes[id];
    if (nd.op == RH_RIR_CONST) return (pooled() && ctx == 1 && !structural(nd.cval)) ? pool_ref(nd.cval) : lit(nd.cval);
    if (nd.op == RH_RIR_INPUT) {
      if (nd.input < P.n_params) return "th[" + std::to_string(nd.input) + "]";
      return "c[" + std::to_string(nd.input - P.targets[t].input_start) + oz() + "]";   // (memory-resident lowering: the row's values stay in scratch too)
    }
    if (ctx != 0 && has_rows() && nd.dep == 0) {
      auto it = inv_slot.find(id);
      if (it == inv_slot.end()) throw std::out_of_range("emit: parameter-only node " + std::to_string(id) + " (op " + std::to_string((int)nd.op) +
                                                         ") of target " + std::to_string(t) + " is used in context " + std::to_string(ctx) + " but has no invariant slot");
      return "inv[" + std::to_string(it->second) + oz() + "]";
    }
    return "n" + std::to_string(id);
  }
  // fast mode only: division by a finite non-zero constant becomes a multiplication by its reciprocal (<= 1 ulp apart)
  bool recip_const(uint32_t b, std::string &out) const {
    if (!fast_div || b >= P.nodes.size()) return false;
    const Node &nb = P.nodes[b];
    if (nb.op != RH_RIR_CONST || nb.cval == 0.0 || !std::isfinite(nb.cval) || !std::isfinite(1.0 / nb.cval)) return false;
    out = (pooled() && !structural(1.0 / nb.cval)) ? pool_ref(1.0 / nb.cval) : lit(1.0 / nb.cval);
    return true;
  }
  // LookupIR: D2I truncation, select table[k - low], out of range is an error (ExprMethodGenerator.scala:57-63).
  // <= 64 entries: a select chain in registers; all-constant tables of any size: a read-only array; otherwise a
  // per-evaluation local array (correct for any table, slow for large ones -- parameter tables belong in gather mode).
  template <class RefFn>
  bool emit_lookup(std::ostringstream &os, uint32_t id, RefFn R, std::string &err, bool use_inv = true) const {
    const Node &nd = P.nodes[id];
    const std::string k = "k" + std::to_string(id), lhs = "    const double n" + std::to_string(id) + " = ";
    const Node &ix = P.nodes[nd.a];
    if (ix.op == RH_RIR_COMPARE && nd.low == -1 && nd.table.size() == 3) {
      // the index is DCMPL's -1/0/+1 itself: select on it directly, no D2I, no range check (after inlining rh_compare the
      // compiler folds these tests into the original comparisons)
      os << lhs << "(" << R(nd.a) << " > 0x0p+0) ? " << R(nd.table[2]) << " : ((" << R(nd.a) << " == 0x0p+0) ? " << R(nd.table[1]) << " : "
         << R(nd.table[0]) << ");\n";
      return true;
    }
    // unsigned difference: a saturated index (INT_MIN / INT_MAX) minus `low` must wrap, not overflow a signed int;
    // every use below compares or indexes k as unsigned
    os << "    const int " << k << " = (int)((unsigned)rh_d2i(" << R(nd.a) << ") - (unsigned)(" << nd.low << "));\n";
    auto it_tab = use_inv ? inv_table.find(id) : inv_table.end();   // (the GLM scalar part has no inv[]: it keeps its own copies)
    bool all_params = chunk > 0 && nd.table.size() > 2;   // memory-resident lowering: theta is in memory (RH_BIGTH), so a table of parameters is ONE indexed load
    for (uint32_t e : nd.table) all_params = all_params && P.nodes[e].op == RH_RIR_INPUT && P.nodes[e].input < P.n_params && !(gather.ok && e == gather.node);
    if (it_tab != inv_table.end()) {
      os << lhs << "((unsigned)" << k << " < " << nd.table.size() << "u) ? inv[" << it_tab->second << " + " << k << "] : RH_NAN;\n";
    } else if (all_params) {
      os << "    static const int p" << id << "[" << nd.table.size() << "] = {";
      for (size_t e = 0; e < nd.table.size(); e++) os << (e ? ", " : "") << P.nodes[nd.table[e]].input;
      os << "};\n" << lhs << "((unsigned)" << k << " < " << nd.table.size() << "u) ? th[p" << id << "[" << k << "]] : RH_NAN;\n";
    } else if (nd.table.size() <= (chunk > 0 ? 4u : 64u)) {
      os << lhs;
      for (size_t e = 0; e + 1 < nd.table.size(); e++) os << "(" << k << " == " << e << ") ? " << R(nd.table[e]) << " : ";
      os << R(nd.table.back()) << ";\n";
    } else {
      if (nd.table.size() > 65536) { err = "Lookup tables with more than 65536 entries are only supported in gather mode"; return false; }
      bool all_const = true;
      for (uint32_t e : nd.table) all_const = all_const && P.nodes[e].op == RH_RIR_CONST && !(pooled() && !structural(P.nodes[e].cval));   // (a pooled entry is a load, not a literal)
      os << "    " << (all_const ? "static const double" : "const double") << " t" << id << "[" << nd.table.size() << "] = {";
      for (size_t e = 0; e < nd.table.size(); e++) os << (e ? ", " : "") << R(nd.table[e]);
      os << "};\n" << lhs << "((unsigned)" << k << " < " << nd.table.size() << "u) ? t" << id << "[" << k << "] : RH_NAN;\n";
    }
    os << "    if ((unsigned)" << k << " >= " << nd.table.size() << "u) err = 1;\n";
    return true;
  }
  // the fma-only spelling applies to the per-row code of streamed targets (the hot loop of every gradient kernel)
  bool fma_row(int ctx) const { return fma_adds && ctx == 1 && has_rows(); }
  // a product computed per row (not a hoisted invariant) that contraction may fold into the add consuming it
  bool fusable(uint32_t id, int ctx) const {
    return fast_div && fma_row(ctx) && id < P.nodes.size() && P.nodes[id].op == RH_RIR_MUL && P.nodes[id].dep != 0 &&
           !(gather.ok && id == gather.node);
  }
  // explicit fusion without the fma-only spelling (xfuse): the row code of a streamed target in a register lowering
  bool xfuse_row(int ctx) const { return xfuse && !fma_adds && ctx == 1 && has_rows() && chunk == 0; }
  bool xfusable(uint32_t id, int ctx) const {
    if (!(xfuse_row(ctx) && id < P.nodes.size() && P.nodes[id].op == RH_RIR_MUL && P.nodes[id].dep != 0 && !(gather.ok && id == gather.node) &&
          !(link.ok && (id == link.V || link.cores.count(id))))) return false;   // (a closed-form node is not its literal product)
    // x * +-1 is a sign, not a product: the compiler folds it into an operand modifier, and an fma would keep an instruction alive for it
    return !(is_const(P.nodes[id].a, 1.0) || is_const(P.nodes[id].a, -1.0) || is_const(P.nodes[id].b, 1.0) || is_const(P.nodes[id].b, -1.0));
  }
  std::string accumulate(const std::string &dst, uint32_t v, int ctx) const {
    if (xfusable(v, ctx)) return "    " + dst + " = __builtin_fma(" + ref(P.nodes[v].a, ctx) + ", " + ref(P.nodes[v].b, ctx) + ", " + dst + ");\n";
    if (!fma_row(ctx)) return "    " + dst + " += " + ref(v, ctx) + ";\n";
    if (fusable(v, ctx)) return "    " + dst + " = __builtin_fma(" + ref(P.nodes[v].a, ctx) + ", " + ref(P.nodes[v].b, ctx) + ", " + dst + ");\n";
    return "    " + dst + " = __builtin_fma(" + ref(v, ctx) + ", rh1, " + dst + ");\n";
  }
  bool emit_node(std::ostringstream &os, uint32_t id, int ctx, std::string &err) const {
    const Node &nd = P.nodes[id];
    auto R = [&](uint32_t x) { return ref(x, ctx); };
    const std::string lhs = "    const double n" + std::to_string(id) + " = ";
    if (xfuse_row(ctx) && (nd.op == RH_RIR_ADD || nd.op == RH_RIR_SUB) && (xfusable(nd.a, ctx) || xfusable(nd.b, ctx))) {
      // (a b) + y -> fma(a, b, y);  (a b) - y -> fma(a, b, -y);  x + (a b) -> fma(a, b, x);  x - (a b) -> fma(-a, b, x): the first operand
      // that is a row-level product is the one that is fused (the rule the compiler's own combiner applies)
      const bool sub = nd.op == RH_RIR_SUB;
      if (xfusable(nd.a, ctx)) os << lhs << "__builtin_fma(" << R(P.nodes[nd.a].a) << ", " << R(P.nodes[nd.a].b) << ", " << (sub ? "-(" + R(nd.b) + ")" : R(nd.b)) << ");\n";
      else os << lhs << "__builtin_fma(" << (sub ? "-(" + R(P.nodes[nd.b].a) + ")" : R(P.nodes[nd.b].a)) << ", " << R(P.nodes[nd.b].b) << ", " << R(nd.a) << ");\n";
      return true;
    }
    if (fma_row(ctx) && (nd.op == RH_RIR_ADD || nd.op == RH_RIR_SUB)) {
      // x + y == fma(x, 1.0, y) and x - y == fma(y, -1.0, x) exactly; where contraction is allowed a row-level product
      // feeding the sum is fused here, explicitly, instead of by the compiler
      const bool sub = nd.op == RH_RIR_SUB;
      if (fusable(nd.a, ctx)) os << lhs << "__builtin_fma(" << R(P.nodes[nd.a].a) << ", " << R(P.nodes[nd.a].b) << ", " << (sub ? "-(" + R(nd.b) + ")" : R(nd.b)) << ");\n";
      else if (fusable(nd.b, ctx)) os << lhs << "__builtin_fma(" << (sub ? "-(" + R(P.nodes[nd.b].a) + ")" : R(P.nodes[nd.b].a)) << ", " << R(P.nodes[nd.b].b) << ", " << R(nd.a) << ");\n";
      else os << lhs << "__builtin_fma(" << R(nd.b) << ", " << (sub ? "-rh1" : "rh1") << ", " << R(nd.a) << ");\n";
      return true;
    }
    switch (nd.op) {
      case RH_RIR_ADD: os << lhs << R(nd.a) << " + " << R(nd.b) << ";\n"; break;
      case RH_RIR_SUB: os << lhs << R(nd.a) << " - " << R(nd.b) << ";\n"; break;
      case RH_RIR_MUL: os << lhs << R(nd.a) << " * " << R(nd.b) << ";\n"; break;
      case RH_RIR_DIV: { std::string rc; if (recip_const(nd.b, rc)) os << lhs << R(nd.a) << " * " << rc << ";\n"; else os << lhs << R(nd.a) << " / " << R(nd.b) << ";\n"; break; }
      case RH_RIR_POW: os << lhs << pow_call(R(nd.a), nd.b, R(nd.b)) << ";\n"; break;
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
      case RH_RIR_LOOKUP: if (!emit_lookup(os, id, [&](uint32_t x) { return R(x); }, err)) return false; break;
      default: err = "emit: unexpected opcode"; return false;
    }
    return true;
  }


  // ---- GLM pattern: a dense linear predictor feeding a scalar nonlinearity -------------------------------------
  // eta = theta_p0 + sum_k theta_pk * x_k (>= 8 terms) and, after output factoring, basis terms {w * x_k for every
  // predictor column, w itself for the bare intercept, plus a few "other" terms}.  Then eta = X.B and the gradient
  // sums X^T.w are genuine dense contractions: rh_grad_glm_kernel runs them on the fp64 matrix cores
  // (v_mfma_f64_16x16x4_f64, 16 chains per wavefront) and only the scalar part elem(eta, ...) -> (w, others) stays on
  // the VALU.  This is the only place MFMA is used.
  struct Glm {
    bool ok = false;
    uint32_t L = 0, w = 0;
    std::vector<int> pred_param, pred_col, pred_acc;  // pred_col = -1: the constant 1 column (bare intercept)
    std::vector<double> pred_scale;                   // eta = sum_k pred_scale[k] * theta[pred_param[k]] * x[pred_col[k]]
    std::vector<std::pair<uint32_t, int>> others;     // (node, accumulator index)
    std::vector<uint32_t> thu;                        // parameters the scalar part reads (compact order)
    std::vector<uint32_t> elem_nodes;                 // row nodes of the scalar part, ascending
    std::vector<uint32_t> inv_nodes;                  // invariant nodes of the scalar part, ascending
  } glm;

  // leaf := [const *] param [* column] in any association;  tree := leaf | tree + tree | tree - tree.  A term is (parameter,
  // column or -1 for the bare intercept, scale): eta = sum scale * theta_param * x_column
  struct LinTerm { int param, col; double scale; };
  bool linleaf(uint32_t id, LinTerm &lt) const {
    const Node &nd = P.nodes[id];
    if (nd.op == RH_RIR_CONST) { lt.scale *= nd.cval; return std::isfinite(lt.scale) && lt.scale != 0.0; }
    if (nd.op == RH_RIR_INPUT) {
      if (nd.input < P.n_params) { if (lt.param >= 0) return false; lt.param = (int)nd.input; }
      else { if (lt.col >= 0) return false; lt.col = (int)(nd.input - P.targets[t].input_start); }
      return true;
    }
    if (nd.op == RH_RIR_MUL) return linleaf(nd.a, lt) && linleaf(nd.b, lt);
    return false;
  }
  bool lincomb(uint32_t id, double sign, std::vector<LinTerm> &out) const {
    const Node &nd = P.nodes[id];
    if (nd.op == RH_RIR_ADD) return lincomb(nd.a, sign, out) && lincomb(nd.b, sign, out);
    if (nd.op == RH_RIR_SUB) return lincomb(nd.a, sign, out) && lincomb(nd.b, -sign, out);
    LinTerm lt{-1, -1, sign};
    if (!linleaf(id, lt) || lt.param < 0) return false;
    out.push_back(lt);
    return true;
  }

  void detect_glm() {
    if (!has_rows() || !factor || basis.empty() || !families.empty()) return;
    // the widest linear predictor among the row nodes
    size_t best = 0;
    for (size_t n = 0; n < P.nodes.size(); n++) {
      if (!reach_row[n] || P.nodes[n].dep == 0 || (P.nodes[n].op != RH_RIR_ADD && P.nodes[n].op != RH_RIR_SUB)) continue;
      std::vector<LinTerm> terms;
      if (lincomb((uint32_t)n, 1.0, terms) && terms.size() >= 2 && terms.size() > best) { best = terms.size(); glm.L = (uint32_t)n; }
    }
    if (!best) return;
    std::vector<LinTerm> terms;
    lincomb(glm.L, 1.0, terms);
    std::map<int, int> col_pred;
    int bare = -1;
    for (auto &pc : terms) {
      if (pc.col < 0) { if (bare >= 0) return; bare = (int)glm.pred_param.size(); }
      else if (col_pred.count(pc.col)) return;
      else col_pred[pc.col] = (int)glm.pred_param.size();
      for (int pp : glm.pred_param) if (pp == pc.param) return;   // one coefficient per predictor
      glm.pred_param.push_back(pc.param); glm.pred_col.push_back(pc.col); glm.pred_acc.push_back(-1); glm.pred_scale.push_back(pc.scale);
    }
    auto col_of = [&](uint32_t x) -> int {
      const Node &nd = P.nodes[x];
      return (nd.op == RH_RIR_INPUT && nd.input >= P.n_params) ? (int)(nd.input - P.targets[t].input_start) : -1;
    };
    // classify the basis terms
    bool have_w = false;
    std::vector<int> other_idx;
    for (size_t j = 0; j < basis.size(); j++) {
      const Node &nd = P.nodes[basis[j]];
      bool matched = false;
      if (nd.op == RH_RIR_MUL) {
        for (int sw = 0; sw < 2 && !matched; sw++) {
          const uint32_t x = sw ? nd.b : nd.a, y = sw ? nd.a : nd.b;
          const int c = col_of(y);
          if (c >= 0 && col_pred.count(c) && col_of(x) < 0 && P.nodes[x].dep != 0) {
            if (have_w && glm.w != x) continue;
            glm.w = x; have_w = true; glm.pred_acc[col_pred[c]] = (int)j; matched = true;
          }
        }
      }
      if (!matched) other_idx.push_back((int)j);
    }
    if (!have_w) return;
    for (int j : other_idx) {
      if (basis[j] == glm.w && bare >= 0 && glm.pred_acc[bare] < 0) glm.pred_acc[bare] = j;
      else glm.others.push_back({basis[j], j});
    }
    for (int a : glm.pred_acc) if (a < 0) return;
    if (glm.others.size() > 8) return;
    // the scalar part: everything reachable from {w, others} with L as a leaf; it must not look inside L
    std::vector<char> in_L(P.nodes.size(), 0), need(P.nodes.size(), 0);
    in_L[glm.L] = 1;
    { std::vector<uint32_t> ops;
      for (size_t n = P.nodes.size(); n-- > 0;) { if (!in_L[n]) continue; operands(P.nodes[n], ops); for (uint32_t o : ops) if (P.nodes[o].op != RH_RIR_INPUT && P.nodes[o].op != RH_RIR_CONST) in_L[o] = 1; } }
    need[glm.w] = 1;
    for (auto &o : glm.others) need[o.first] = 1;
    { std::vector<uint32_t> ops;
      for (size_t n = P.nodes.size(); n-- > 0;) {
        if (!need[n] || n == glm.L) continue;
        operands(P.nodes[n], ops);
        for (uint32_t o : ops) need[o] = 1;
      } }
    std::map<uint32_t, int> thu_idx;
    for (size_t n = 0; n < P.nodes.size(); n++) {
      if (!need[n]) continue;
      if (in_L[n] && n != glm.L) return;  // a partial sum of eta is used elsewhere
      const Node &nd = P.nodes[n];
      if (nd.op == RH_RIR_INPUT && nd.input < P.n_params) { if (!thu_idx.count(nd.input)) { thu_idx[nd.input] = (int)glm.thu.size(); glm.thu.push_back(nd.input); } continue; }
      if (trivial((uint32_t)n) || n == glm.L) continue;
      if (nd.dep == 0) glm.inv_nodes.push_back((uint32_t)n); else glm.elem_nodes.push_back((uint32_t)n);
    }
    if (glm.thu.size() > 8) return;
    glm.ok = true;
  }

  // ---- fast mode: the scalar part of a Bernoulli-logit GLM in closed form ------------------------------------------------
  // The reference writes the logistic likelihood naively -- p = 1 / (1 + exp(-eta)), Bernoulli.logDensity =
  // Real.eq(y, 0, log(1 - p), log p) (core/Discrete.scala:38-52, compute/Real.scala:42) -- and differentiates it symbolically
  // (compute/Gradient.scala), which leaves one exp, one log, three divisions and several selects per row-chain evaluation:
  // ~190 VALU instructions, more than the two dense contractions around them.  Mathematically the pair is
  //       value = -softplus(s * eta),   w = d value / d eta = -s * sigmoid(s * eta),   s = +1 if y == c else -1  (or the reverse)
  // and in that form it costs one exp, one reciprocal and one short log series (rh_logit_link).  The rewrite is VERIFIED, not
  // pattern-matched on the differentiated DAG: the shape requirements below make the scalar part a function of (eta, one data
  // column) alone, the candidate is fixed by the value at two probe points, and both value AND w of the original DAG
  // must agree with the closed form (1e-8 relative: the naive form itself loses digits through 1 - p) at 15 eta in [-12, 12]
  // x 5 values of y, evaluated here on the host.  Anything that does not verify keeps its literal lowering.  Fast mode only:
  // strict builds keep the reference's arithmetic operation for operation.
  struct Logit { bool ok = false; int ycol = -1; double c = 0.0, s_hit = 1.0, s_miss = -1.0, kappa = 1.0; } logit;

  double host_eval(uint32_t id, double eta, int ycol, double yv, std::map<uint32_t, double> &memo, bool &bad) const {
    if (id == glm.L) return eta;
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    const Node &nd = P.nodes[id];
    auto E = [&](uint32_t x) { return host_eval(x, eta, ycol, yv, memo, bad); };
    double v = 0.0;
    switch (nd.op) {
      case RH_RIR_CONST: v = nd.cval; break;
      case RH_RIR_INPUT:
        if (nd.input >= P.n_params && (int)(nd.input - P.targets[t].input_start) == ycol) v = yv; else { bad = true; v = 0.0; }
        break;
      case RH_RIR_ADD: v = E(nd.a) + E(nd.b); break;
      case RH_RIR_SUB: v = E(nd.a) - E(nd.b); break;
      case RH_RIR_MUL: v = E(nd.a) * E(nd.b); break;
      case RH_RIR_DIV: v = E(nd.a) / E(nd.b); break;
      case RH_RIR_POW: v = std::pow(E(nd.a), E(nd.b)); break;
      case RH_RIR_COMPARE: { const double l = E(nd.a), r = E(nd.b); v = l > r ? 1.0 : (l == r ? 0.0 : -1.0); break; }
      case RH_RIR_EXP: v = std::exp(E(nd.a)); break;
      case RH_RIR_LOG: v = std::log(E(nd.a)); break;
      case RH_RIR_ABS: v = std::fabs(E(nd.a)); break;
      case RH_RIR_NOOP: v = E(nd.a); break;
      case RH_RIR_SEQ: (void)E(nd.a); v = E(nd.b); break;
      case RH_RIR_LOOKUP: {
        const double ix = E(nd.a);
        const long long k = (ix != ix ? 0LL : (long long)ix) - (long long)nd.low;
        for (uint32_t e : nd.table) (void)E(e);
        if (k < 0 || k >= (long long)nd.table.size()) { bad = true; v = 0.0; } else v = E(nd.table[(size_t)k]);
        break;
      }
      default: bad = true;  // trigonometric nodes do not occur in a logit link
    }
    memo[id] = v;
    return v;
  }

  void detect_logit() {
    if (!glm.ok || !fast_div || !glm.thu.empty() || glm.others.size() != 1 || !glm.inv_nodes.empty()) return;
    // the scalar part may read exactly one data column.  Either that column only feeds compare(y, c) with one constant c -- then
    // the closed form is checked for y = c and four other values -- or (the reference's own lowering: y indexes a Lookup and
    // multiplies masked terms) the column's distinct values are known from the data (Program::col_domain, at most two of
    // them) and the closed form is checked on exactly those.
    int ycol = -1; double c = 0.0; bool have_c = false, only_compare = true;
    for (uint32_t n : glm.elem_nodes) {
      const Node &nd = P.nodes[n];
      std::vector<uint32_t> ops; operands(nd, ops);
      for (uint32_t o : ops) {
        const Node &on = P.nodes[o];
        if (on.op == RH_RIR_INPUT && on.input >= P.n_params) {
          const int col = (int)(on.input - P.targets[t].input_start);
          if (ycol >= 0 && col != ycol) return;
          ycol = col;
          if (nd.op != RH_RIR_COMPARE || o != nd.a || P.nodes[nd.b].op != RH_RIR_CONST) { only_compare = false; continue; }
          if (have_c && P.nodes[nd.b].cval != c) only_compare = false;
          c = P.nodes[nd.b].cval; have_c = true;
        }
      }
    }
    if (ycol < 0) return;
    std::vector<double> ys;
    if (only_compare && have_c && std::isfinite(c)) ys = {c, c + 1.0, c - 1.0, c + 0.5, c + 2.0};
    else {
      const size_t g = (size_t)P.targets[t].col0 + (size_t)ycol;
      if (g >= P.col_domain.size() || P.col_domain[g].empty() || P.col_domain[g].size() > 2) return;
      ys = P.col_domain[g];
      c = ys[0];
    }
    auto original = [&](double eta, double yv, double &val, double &w) -> bool {
      std::map<uint32_t, double> memo; bool bad = false;
      val = host_eval(glm.others[0].first, eta, ycol, yv, memo, bad);
      w = host_eval(glm.w, eta, ycol, yv, memo, bad);
      return !bad;
    };
    auto softplus = [](double x) { return x > 0 ? x + std::log1p(std::exp(-x)) : std::log1p(std::exp(x)); };
    auto sigmoid = [](double x) { return x >= 0 ? 1.0 / (1.0 + std::exp(-x)) : std::exp(x) / (1.0 + std::exp(x)); };
    // fix the signs from one probe per branch: value(eta = 1) is -softplus(1) = -1.3133 or -softplus(-1) = -0.3133
    double s[2] = {1.0, 1.0};
    for (int b = 0; b < 2 && b < (int)ys.size(); b++) {
      double val, w;
      if (!original(1.0, ys[(size_t)b], val, w)) return;
      if (std::fabs(val + softplus(1.0)) < 1e-9) s[b] = 1.0;
      else if (std::fabs(val + softplus(-1.0)) < 1e-9) s[b] = -1.0;
      else return;
    }
    if (ys.size() == 1) s[1] = s[0];
    // the adjoint basis term may carry a constant factor (the output factoring peels signs into alpha): w = kappa * d value / d eta
    double kappa = 1.0;
    { double val, w;
      if (!original(1.0, ys[0], val, w)) return;
      kappa = w / (-s[0] * sigmoid(s[0] * 1.0));
      if (!std::isfinite(kappa) || kappa == 0.0) return;
      if (std::fabs(kappa - std::nearbyint(kappa)) < 1e-9) kappa = std::nearbyint(kappa); }
    static const double etas[] = {-12.0, -7.3, -3.1, -1.7, -0.9, -0.31, -1e-3, 0.0, 1e-3, 0.22, 0.8, 1.9, 3.7, 8.1, 12.0};
    for (double yv : ys)
      for (double eta : etas) {
        double val, w;
        if (!original(eta, yv, val, w)) return;
        const double sg = yv == c ? s[0] : s[1];
        const double cv = -softplus(sg * eta), cw = kappa * (-sg * sigmoid(sg * eta));
        if (!(std::fabs(val - cv) <= 1e-8 * std::max(1.0, std::fabs(cv))) || !(std::fabs(w - cw) <= 1e-8 * std::max(1.0, std::fabs(kappa)))) return;
      }
    logit.ok = true; logit.ycol = ycol; logit.c = c; logit.s_hit = s[0]; logit.s_miss = s[1]; logit.kappa = kappa;
  }

  std::string glm_ref(uint32_t id) const {
    if (id == glm.L) return "eta";
    const Node &nd = P.nodes[id];
    if (nd.op == RH_RIR_CONST) return lit(nd.cval);
    if (nd.op == RH_RIR_INPUT) {
      if (nd.input < P.n_params) {
        for (size_t k = 0; k < glm.thu.size(); k++) if (glm.thu[k] == nd.input) return "thu[" + std::to_string(k) + "]";
        return "0x0p+0";
      }
      return "RH_GLM_COL(" + std::to_string(nd.input - P.targets[t].input_start) + ")";
    }
    return "n" + std::to_string(id);
  }
  bool emit_glm(std::ostringstream &os, std::string &err) const {
    const size_t Pn = glm.pred_param.size();
    os << "template <> struct rh_glm<" << t << "> {\n  static constexpr int P = " << Pn << ", NOTHER = " << glm.others.size()
       << ", NTHU = " << glm.thu.size() << ", NCOLS = " << P.targets[t].n_cols << ";\n";
    auto arr = [&](const char *name, const std::vector<int> &v) {
      os << "  static constexpr int " << name << "[" << (v.empty() ? 1 : v.size()) << "] = {";
      for (size_t i = 0; i < v.size(); i++) os << (i ? "," : "") << v[i];
      if (v.empty()) os << "0";
      os << "};\n";
    };
    arr("pred_param", glm.pred_param); arr("pred_col", glm.pred_col); arr("pred_acc", glm.pred_acc);
    os << "  static constexpr double pred_scale[" << Pn << "] = {";
    for (size_t i = 0; i < Pn; i++) os << (i ? "," : "") << lit(glm.pred_scale[i]);
    os << "};\n";
    std::vector<int> oacc, thu;
    for (auto &o : glm.others) oacc.push_back(o.second);
    for (uint32_t x : glm.thu) thu.push_back((int)x);
    arr("other_acc", oacc); arr("thu_param", thu);
    // the scalar part; RH_GLM_COL(j) reads column j of the current row from the LDS tile.  elem_g(): the same without the terms only
    // the log-density needs (their `other` sums stay 0), for the mid-trajectory gradient requests -- see value_only()
    const std::vector<char> vo = value_only();
    for (int pass = 0; pass < 2; pass++) {
      const bool g_only = pass == 1;
      os << "  template <class ColFn>\n  static RH_DEV void " << (g_only ? "elem_g" : "elem") << "(const double *thu, const double eta, ColFn RH_GLM_COL, double &w, double *other, int &err) {\n"
            "    (void)thu; (void)eta; (void)other; (void)err;\n";
      if (logit.ok) {  // verified closed form of the Bernoulli-logit scalar part (detect_logit)
        const bool drop = g_only && !glm.others.empty() && vo[(size_t)glm.others[0].second];
        os << "    const double s = (RH_GLM_COL(" << logit.ycol << ") == " << lit(logit.c) << ") ? " << lit(logit.s_hit) << " : " << lit(logit.s_miss) << ";\n"
           << "    double sp, sg;\n    rh_logit_link(s * eta, sp, sg);\n"
           << (logit.kappa == 1.0 ? std::string("    w = -(s * sg);\n") : "    w = " + lit(-logit.kappa) + " * (s * sg);\n")
           << (drop ? "    other[0] = 0x0p+0; (void)sp;\n  }\n" : "    other[0] = -sp;\n  }\n");   // (dropped: the softplus half of the link is dead code)
        continue;
      }
      // emit_node spells operands through ref(); the GLM scalar part needs its own spelling
      std::vector<char> need(P.nodes.size(), 1);
      if (g_only) {
        need.assign(P.nodes.size(), 0);
        need[glm.w] = 1;
        for (auto &o : glm.others) if (!vo[(size_t)o.second]) need[o.first] = 1;
        sweep(need);
      }
      for (uint32_t n : glm.inv_nodes) if (need[n] && !emit_glm_node(os, n, err)) return false;
      for (uint32_t n : glm.elem_nodes) if (need[n] && !emit_glm_node(os, n, err)) return false;
      os << "    w = " << glm_ref(glm.w) << ";\n";
      for (size_t k = 0; k < glm.others.size(); k++)
        os << "    other[" << k << "] = " << (g_only && vo[(size_t)glm.others[k].second] ? std::string("0x0p+0") : glm_ref(glm.others[k].first)) << ";\n";
      os << "  }\n";
    }
    os << "};\n";
    return true;
  }
  bool emit_glm_node(std::ostringstream &os, uint32_t id, std::string &err) const {
    const Node &nd = P.nodes[id];
    auto R = [&](uint32_t x) { return glm_ref(x); };
    const std::string lhs = "    const double n" + std::to_string(id) + " = ";
    switch (nd.op) {
      case RH_RIR_ADD: os << lhs << R(nd.a) << " + " << R(nd.b) << ";\n"; break;
      case RH_RIR_SUB: os << lhs << R(nd.a) << " - " << R(nd.b) << ";\n"; break;
      case RH_RIR_MUL: os << lhs << R(nd.a) << " * " << R(nd.b) << ";\n"; break;
      case RH_RIR_DIV: { std::string rc; if (recip_const(nd.b, rc)) os << lhs << R(nd.a) << " * " << rc << ";\n"; else os << lhs << R(nd.a) << " / " << R(nd.b) << ";\n"; break; }
      case RH_RIR_POW: os << lhs << pow_call(R(nd.a), nd.b, R(nd.b)) << ";\n"; break;
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
      case RH_RIR_SEQ: os << lhs << R(nd.b) << ";\n"; break;
      case RH_RIR_LOOKUP: if (!emit_lookup(os, id, [&](uint32_t x) { return R(x); }, err, false)) return false; break;
      default: err = "emit: unexpected opcode"; return false;
    }
    return true;
  }

  int nacc() const { return has_rows() ? (int)basis.size() : 0; }

  bool emit(std::ostringstream &os, std::string &err) {
    const Target &T = P.targets[t];
    const bool rows = has_rows();
    int rowt = -1;
    if (rows) { rowt = 0; for (uint32_t u = 0; u < t; u++) if (P.targets[u].n_cols) rowt++; }
    os << "template <> struct rh_target<" << t << "> {\n";
    os << "  static constexpr int NCOLS = " << T.n_cols << ", COL0 = " << T.col0 << ", NINV = " << inv_slot.size()
       << ", NACC = " << nacc() << ", ROWT = " << rowt << ";\n";
    os << "  static constexpr bool HAS_ROWS = " << (rows ? "true" : "false") << ";\n";
    { bool any_vo = false;
      if (rows) for (char c : value_only()) any_vo = any_vo || c;
      os << "  static constexpr bool HAS_VALUE_ONLY = " << (any_vo ? "true" : "false") << ";   // row_g() differs from row()\n"; }
    os << "  static constexpr bool HAS_GATHER = " << (gather.ok ? "true" : "false") << ";\n  static constexpr int G_COL = " << gather.col
       << ", G_FIRST = " << gather.first << ", G_COUNT = " << gather.count << ", G_LOW = " << gather.low << ";\n";
    // ---- invariants
    os << "  static RH_DEV void invariants(const double (&th)[RH_NTH], rh_acc_t *inv, int &err) {\n    (void)th; (void)inv; (void)err;\n";
    if (rows) {
      std::ostringstream b;
      for (size_t n = 0; n < P.nodes.size(); n++)
        if (reach_inv[n] && P.nodes[n].dep == 0 && !trivial((uint32_t)n))
          if (!emit_node(b, (uint32_t)n, 0, err)) return false;
      for (auto &kv : inv_slot) b << "    inv[" << kv.second << oz() << "] = n" << kv.first << ";\n";
      os << (chunk > 0 ? chunk_body(b.str(), chunk) : b.str());
    }
    os << "  }\n";
    // ---- row
    if (rows) {
      // (xfuse: what is not fused explicitly below stays unfused -- in every copy of the row code alike)
      const std::string xpragma = xfuse_row(1) ? "#pragma clang fp contract(off)\n" : "";
      if (gmode)
        os << "  static RH_DEV void row(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, const double gz, rh_acc_t *acc, double &sv, int &err) {\n"
              << xpragma << "    (void)th; (void)inv; (void)c; (void)gz; (void)acc; (void)sv; (void)err;\n    const double rh1 = rh_one(); (void)rh1;\n";
      else
        os << "  static RH_DEV void row(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, rh_acc_t *acc, int &err) {\n"
              << xpragma << "    (void)th; (void)inv; (void)c; (void)acc; (void)err;\n    const double rh1 = rh_one(); (void)rh1;\n";
      // row() and, for the tick engine's mid-trajectory gradient requests, row_g(): the same statements without the value-only part
      const std::vector<char> vo = value_only();
      bool any_vo = false;
      for (char c : vo) any_vo = any_vo || c;
      for (int pass = 0; pass < (any_vo ? 2 : 1); pass++) {   // (no value-only part: row() serves both kinds of request)
        const bool g_only = pass == 1;
        const std::vector<char> reach = g_only ? reach_from(row_roots_g()) : reach_row;
        if (g_only) {
          if (gmode)
            os << "  static RH_DEV void row_g(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, const double gz, rh_acc_t *acc, double &sv, int &err) {\n"
                  << xpragma << "    (void)th; (void)inv; (void)c; (void)gz; (void)acc; (void)sv; (void)err;\n    const double rh1 = rh_one(); (void)rh1;\n";
          else
            os << "  static RH_DEV void row_g(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, rh_acc_t *acc, int &err) {\n"
                  << xpragma << "    (void)th; (void)inv; (void)c; (void)acc; (void)err;\n    const double rh1 = rh_one(); (void)rh1;\n";
        }
        bool link_open = false;
        std::ostringstream b;
        for (size_t n = 0; n < P.nodes.size(); n++) {
          if (!reach[n] || trivial((uint32_t)n) || P.nodes[n].dep == 0) continue;
          if (gather.ok && n == gather.node) continue;  // the kernel supplies the gathered parameter
          if (link.ok && (n == link.V || link.cores.count((uint32_t)n))) {   // verified closed forms (detect_link)
            if (!link_open) { emit_link_prelude(b); link_open = true; }      // (row_g: the softplus half of rh_logit_link is dead code there)
            if (n == link.V) {
              if (xfuse_row(1)) b << "    const double n" << n << " = __builtin_fma(-lk_ab, lk_sp, __builtin_fma(lk_a, lk_t, " << link_sum(link.cterms) << "));\n";
              else b << "    const double n" << n << " = " << link_sum(link.cterms) << " + (lk_a * lk_t - lk_ab * lk_sp);\n";
            }
            else b << "    const double n" << n << " = " << lit(link.cores.at((uint32_t)n)) << " * lk_g;\n";
            continue;
          }
          if (!emit_node(b, (uint32_t)n, 1, err)) return false;
        }
        for (size_t j = 0; j < basis.size(); j++)
          if (basis[j] != NONE && !in_family[j] && !(g_only && vo[j])) b << accumulate("acc[" + std::to_string(j) + oz() + "]", basis[j], 1);
        for (const Family &f : families) {   // eq(column, k, g, 0) for every k of the block: the column's value picks the one accumulator
          b << "    { const double ix = " << ref(f.col, 1) << "; const int kk = (int)ix - (" << f.kmin << ");\n"
            << "      if (ix == (double)(int)ix && (unsigned)kk < " << f.size << "u) acc[" << f.base << " + kk] += " << ref(f.g, 1) << "; }\n";
        }
        if (gather.ok) b << accumulate("sv", gather.sv, 1);
        os << (chunk > 0 && !link_open ? chunk_body(b.str(), chunk) : b.str());   // (the closed-form link is a light row by construction)
        os << "  }\n";
      }
      // ---- finish: tot[o] += alpha * S[j] + nrows * beta
      os << "  static RH_DEV void finish(const double (&th)[RH_NTH], const rh_acc_t *inv, const rh_acc_t *S, const double nrows, double (&tot_)[RH_NOUT]) {\n"
            "    (void)th; (void)inv; (void)S; (void)nrows; rh_acc_t *const tot = tot_; (void)tot;\n";
      for (size_t o = 0; o < outs.size(); o++) {
        const Lin &l = outs[o];
        std::string e;
        if (l.term != NONE && l.alpha != ZERO) e = l.alpha == ONE ? "S[" + std::to_string(l.term) + oz() + "]" : ref(l.alpha, 2) + " * S[" + std::to_string(l.term) + oz() + "]";
        for (auto &tm : outs_multi[o]) {
          if (tm.second == ZERO) continue;
          e += (e.empty() ? "" : " + ") + (tm.second == ONE ? "S[" + std::to_string(tm.first) + oz() + "]" : ref(tm.second, 2) + " * S[" + std::to_string(tm.first) + oz() + "]");
        }
        if (l.beta != ZERO) e += (e.empty() ? "" : " + ") + ("nrows * " + ref(l.beta, 2));
        if (!e.empty()) os << "    tot[" << o << oz() << "] += " << e << ";\n";
      }
      os << "  }\n";
    } else {
      // data-free target: evaluated once, outputs(o) += f_o(theta)  (DataFunction.scala:73-83)
      // (a data-free target of a model with hundreds of parameters is thousands of statements: one copy per translation unit, not
      //  one per kernel that evaluates it)
      if (P.n_params > 512)
        os << "#ifndef RH_DEV_NOINLINE\n#define RH_DEV_NOINLINE __device__ __attribute__((noinline))\n#endif\n";
      os << "  static " << (P.n_params > 512 ? "RH_DEV_NOINLINE" : "RH_DEV")
         << " void row(const double (&th)[RH_NTH], const rh_acc_t *inv, const double *c, double (&acc_)[RH_NOUT], int &err) {\n"
            "    (void)th; (void)inv; (void)c; (void)err; rh_acc_t *const acc = acc_; (void)acc;\n";
      if (!merged_away) {
        std::ostringstream b;
        for (size_t n = 0; n < P.nodes.size(); n++) {
          if (!reach_row[n] || trivial((uint32_t)n)) continue;
          if (!emit_node(b, (uint32_t)n, 1, err)) return false;
        }
        // accumulate target by target, in DataFunction order; every node above was evaluated exactly once
        for (uint32_t tt = t; tt <= run_end; tt++)
          for (size_t o = 0; o < (gmode ? (size_t)n_shared + 1 : P.targets[tt].outputs.size()); o++) {
            const Node &on = P.nodes[P.targets[tt].outputs[o]];
            if (on.op == RH_RIR_CONST && on.cval == 0.0 && !std::signbit(on.cval)) continue;  // += +0.0 is the identity
            b << "    acc[" << o << oz() << "] += " << ref(P.targets[tt].outputs[o], 1) << ";\n";
          }
        os << (chunk > 0 ? chunk_body(b.str(), chunk, "    ", true) : b.str());
      }
      os << "  }\n";
    }
    os << "};\n";
    return true;
  }
};

}  // namespace

static bool emit_hip_impl(const Program &P, const EmitOptions &o, std::string &defines, std::string &targets, std::string &err,
                          EmitInfo *info) {
  EmitInfo local_info; EmitInfo &I = info ? *info : local_info;
  I = EmitInfo();
  std::ostringstream os;
  os << "template <int T> struct rh_target;\ntemplate <int T> struct rh_glm;\n";
  if (o.fp_contract) os << "#pragma clang fp contract(fast)\n";
  int nacc_max = 1, nrowt = 0, glm_target = -1;
  bool glm_small = false;
  // gather mode: some LOOKUP over >= `gather_min` consecutive trailing parameters indexed by a data column
  bool gmode = false; uint32_t n_shared = P.n_params;
  for (const Node &nd : P.nodes) {
    if (nd.op != RH_RIR_LOOKUP || (int)nd.table.size() < o.gather_min) continue;
    const Node &ix = P.nodes[nd.a], &t0 = P.nodes[nd.table[0]];
    if (ix.op == RH_RIR_INPUT && ix.input >= P.n_params && t0.op == RH_RIR_INPUT && t0.input < P.n_params &&
        t0.input + nd.table.size() == P.n_params) { gmode = true; n_shared = std::min(n_shared, t0.input); }
  }
  int ngather = 0;
  std::map<uint64_t, int> kpool_index;
  for (uint32_t t = 0; t < P.targets.size(); t++) {
    TargetEmitter te(P, t, o.factor_outputs);
    te.fast_div = o.fp_contract;
    te.fma_adds = o.fma_adds;
    te.xfuse = o.fp_contract && o.xfuse;
    te.chunk = o.chunk;
    if (P.targets[t].n_cols == 0) {  // runs of consecutive data-free targets share one evaluation
      if (t > 0 && P.targets[t - 1].n_cols == 0) te.merged_away = true;
      else { uint32_t e = t; while (e + 1 < P.targets.size() && P.targets[e + 1].n_cols == 0) e++; te.run_end = e; }
    }
    te.gmode = gmode; te.n_shared = n_shared;
    if (o.const_pool) { te.kpool = &I.kpool; te.kpool_index = &kpool_index; }
    if (gmode) { if (!te.detect_gather(err)) return false; if (te.gather.ok) ngather++; }
    te.plan();
    if (o.logit_link) te.detect_link();
    if (!te.emit(os, err)) return false;
    { EmitInfo::TargetInfo ti; ti.has_rows = P.targets[t].n_cols > 0; ti.has_gather = te.gather.ok; ti.g_col = te.gather.col;
      ti.g_count = te.gather.count; ti.g_low = te.gather.low; I.targets.push_back(ti); }
    if (glm_target < 0 && o.glm_mfma && !gmode) {
      te.detect_glm();
      if (o.logit_link) te.detect_logit();
      if (te.glm.ok) { if (!te.emit_glm(os, err)) return false; glm_target = (int)t; glm_small = te.glm.pred_param.size() <= 8; }
    }
    nacc_max = std::max(nacc_max, te.nacc());
    if (P.targets[t].n_cols) nrowt++;
  }
  os << "#pragma clang fp contract(off)\n";
  targets = os.str();
  std::ostringstream d;
  d << "#define RH_HAS_GATHER " << (gmode ? 1 : 0) << "\n#define RH_NSHARED " << n_shared << "\n#define RH_NTH " << n_shared
    << "\n#define RH_NGATHER " << ngather << "\n";
  const int slots = (int)((P.n_params + 63) / 64);
  const bool bign = o.force_bign || slots > 8 || o.chunk > 0;   // (the memory-resident lowering keeps the chain vectors in HBM, theta and the outputs in memory: RH_BIGTH)
  // (generic models beyond 512 parameters read theta in place and accumulate their outputs in a per-chain scratch area: RH_BIGTH)
  I.bign = bign;
  d << "#define RH_BIGN " << (bign ? 1 : 0) << "\n#define RH_HEAVY " << (o.chunk > 0 ? 1 : 0) << "\n#define RH_BIGU " << o.big_unroll << "\n";
  // chain packing: without observation rows a chain only needs as many lanes as it has parameters
  if (o.pack && nrowt == 0 && !bign && !gmode && P.n_params <= 32) I.pack_l = P.n_params <= 8 ? 8 : (P.n_params <= 16 ? 16 : 32);
  d << "#ifndef RH_PACK_L\n#define RH_PACK_L " << I.pack_l << "\n#endif\n";
  d << "#define RH_NVARS " << P.n_params << "\n#define RH_NOUT " << (n_shared + 1) << "\n#define RH_SLOTS "
    << ((P.n_params + 63) / 64) << "\n#define RH_NTARGETS " << P.targets.size() << "\n#define RH_FP_CONTRACT "
    << (o.fp_contract ? 1 : 0) << "\n#define RH_ROWS_UNROLL " << (o.rows_unroll > 0 ? o.rows_unroll : 4) << "\n";
  // chains per wavefront in the gradient kernels: K * NACC fp64 accumulators per lane must fit the register budget
  // chains per wavefront in the batched gradient kernels: as many as the accumulators leave room for; the gather kernel
  // also carries the segmented scatter state per chain and is best at 4 (cfg 5: 8.4 ms vs 12.4 ms per gradient at 8)
  const int grad_k = o.grad_chains > 0 ? o.grad_chains : std::max(1, std::min(gmode ? 4 : 8, 48 / std::max(1, nacc_max + (gmode ? 1 : 0))));
  I.gather_mode = gmode; I.n_shared = (int)n_shared; I.grad_k = grad_k; I.nacc_max = nacc_max; I.glm_target = glm_target; I.glm_small = glm_small;
  d << "#define RH_NROWTARGETS " << nrowt << "\n#define RH_NACC_MAX " << nacc_max << "\n#define RH_GRAD_K "
    << grad_k << "\n#define RH_GRAD_U " << (o.grad_unroll > 0 ? o.grad_unroll : 2)
    << "\n#define RH_GRAD_PIPELINE " << o.grad_pipeline << "\n";
  d << "#define RH_KPOOL " << I.kpool.size() << "\n";   // doubles in the constant pool the data-free targets read (rh_model_data.kpool)
  if (glm_target >= 0) d << "#define RH_GLM_TARGET " << glm_target << "\n#define RH_GLM_SMALL " << (glm_small ? 1 : 0) << "\n";
  if (o.strict_math) d << "#define RH_EXP(x) rh_strict_exp(x)\n#define RH_LOG(x) rh_strict_log(x)\n";
  else d << "#define RH_EXP(x) exp(x)\n#define RH_LOG(x) " << (o.fast_log ? "rh_fast_log(x)" : "log(x)") << "\n";
  defines = d.str();
  return true;
}
bool emit_hip(const Program &P, const EmitOptions &o, std::string &defines, std::string &targets, std::string &err,
              EmitInfo *info) {
  const Program Q = o.simplify ? simplify(P, o.fp_contract) : P;
  if (emit_hip_impl(Q, o, defines, targets, err, info)) return true;
  // a model that has the shape of gather mode (a Lookup over a long run of trailing parameters indexed by a column) but not its
  // preconditions -- e.g. the table's prior sits in a data-free target -- is lowered on the generic path (the table as a
  // per-evaluation array: correct, slow) as long as it has few enough parameters for it
  if (err.rfind("gather mode:", 0) != 0) return false;
  if (rh::knob("RH_ROLL_WHY")) std::fprintf(stderr, "rainier-hip: generic path instead of gather mode: %s\n", err.c_str());
  EmitOptions o2 = o;
  o2.gather_min = 0x7fffffff;
  std::string err2;
  if (!emit_hip_impl(Q, o2, defines, targets, err2, info)) return false;   // (err keeps the gather-mode reason)
  err.clear();
  return true;
}
static bool emit_requirements_impl(const Program &P, const EmitOptions &o, std::string &defines, std::string &body, std::string &err);
bool emit_requirements(const Program &P, const EmitOptions &o, std::string &defines, std::string &body, std::string &err) {
  if (!o.simplify) return emit_requirements_impl(P, o, defines, body, err);
  return emit_requirements_impl(simplify(P, o.fp_contract), o, defines, body, err);
}
static bool emit_requirements_impl(const Program &P, const EmitOptions &o, std::string &defines, std::string &body, std::string &err) {
  // one shared evaluation of the union DAG (like a run of data-free targets), then out[m] = requirement m
  TargetEmitter te(P, 0, false);
  te.fast_div = o.fp_contract;
  te.run_end = (uint32_t)P.targets.size() - 1;
  for (uint32_t t = 0; t < P.targets.size(); t++)
    if (P.targets[t].n_cols) { err = "requirements program with data columns"; return false; }
  te.reach_row.assign(P.nodes.size(), 0);
  for (auto &T : P.targets) te.reach_row[T.outputs[0]] = 1;
  te.sweep(te.reach_row);
  std::ostringstream os;
  if (o.fp_contract) os << "#pragma clang fp contract(fast)\n";
  os << "RH_DEV void rh_req_eval(const double (&th)[RH_NTH], double (&out)[RH_NREQ], int &err) {\n  (void)th; (void)err;\n"
        "  const double *inv = nullptr, *c = nullptr; (void)inv; (void)c;\n";
  for (size_t n = 0; n < P.nodes.size(); n++) {
    if (!te.reach_row[n] || te.trivial((uint32_t)n)) continue;
    if (!te.emit_node(os, (uint32_t)n, 1, err)) return false;
  }
  for (size_t m = 0; m < P.targets.size(); m++) os << "  out[" << m << "] = " << te.ref(P.targets[m].outputs[0], 1) << ";\n";
  os << "}\n#pragma clang fp contract(off)\n";
  body = os.str();
  std::ostringstream d;
  d << "#define RH_NVARS " << P.n_params << "\n#define RH_NTH " << P.n_params << "\n#define RH_SLOTS " << ((P.n_params + 63) / 64) << "\n#define RH_NREQ " << P.targets.size() << "\n";
  if (o.strict_math) d << "#define RH_EXP(x) rh_strict_exp(x)\n#define RH_LOG(x) rh_strict_log(x)\n";
  else d << "#define RH_EXP(x) exp(x)\n#define RH_LOG(x) " << (o.fast_log ? "rh_fast_log(x)" : "log(x)") << "\n";
  defines = d.str();
  return true;
}
}  // namespace rh
