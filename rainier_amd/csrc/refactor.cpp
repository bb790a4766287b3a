// refactor.cpp -- fast-mode re-association of the row targets of a program whose derived columns were just folded back
// into products of base columns (columns.cpp).
//
// The reference differentiates symbolically and distributes every data-only factor into its own column, so a GLM's gradient
// arrives as  out_k = A*(x_k*y) + B*(x_k*y) + C*(-x_k*(y-1))  instead of  x_k * w.  After canonicalize_columns the x_k are
// visible again, but they sit inside separate product terms.  This pass brings every row-dependent node into a normal form
//        sum of  coefficient * monomial,   monomial = product of atoms with integer exponents
// (atoms: inputs, and every non-polynomial node -- exp, log, general pow, compare, lookup, NOOP-wrapped affine columns, and
// "sum atoms" for sums that are multiplied by other sums or raised to a power: SUM x SUM IS NEVER EXPANDED, only
// monomial x sum distributes), and rebuilds it as
//        s * F * (plain sum of the remaining monomials)
// with F the monomial common to ALL terms and s a common coefficient.  Nothing is factored partially (no Horner splitting):
// masked terms such as S*(y-1) stay separate products, so a branch that is switched off by a 0 mask never meets a large
// term of the other branch in a subtraction.  Equal sums are built once (hash-consed on the normal form, sign-normalised), so
// the residual / linear predictor shared by the value and all gradients is computed once per row.
//
// Rounding changes (products are re-associated, x/y becomes x*y^-1 grouped into one division per monomial), exactly like FMA
// contraction or output factoring: fast mode only, and only for programs that arrived with derived columns.  If the
// rebuilt row code of a target is not cheaper than the original, the original is kept for that target.
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

typedef std::vector<std::pair<uint32_t, int>> Mono;   // (atom id, exponent != 0), ascending atom id
typedef std::vector<std::pair<Mono, double>> Poly;    // (monomial, coefficient != 0), ascending monomial


struct Atom {
  uint32_t op = 0;                 // RH_RIR_INPUT, a unary op, POW, COMPARE, LOOKUP, or 0xFFFFFFFF = a sum atom
  uint32_t input = 0;
  int32_t low = 0;
  std::vector<uint32_t> kids;      // polynomial ids
  uint32_t dep = 0;
  bool operator<(const Atom &o) const { return std::tie(op, input, low, kids) < std::tie(o.op, o.input, o.low, o.kids); }
};
constexpr uint32_t SUM_ATOM = 0xFFFFFFFFu;
constexpr uint32_t ZERO_PART = 0xFFFFFFFFu;  // in a column's part list: a block of zeros as long as the same block of the target's first column

struct Refactor {
  const Program &P;
  std::vector<Atom> atoms;
  std::map<Atom, uint32_t> atom_ids;
  std::vector<Poly> polys;
  std::map<Poly, uint32_t> poly_ids;
  std::vector<int> node_poly;  // per input node: polynomial id (-1 = not yet)

  explicit Refactor(const Program &p) : P(p), node_poly(p.nodes.size(), -1) {}

  uint32_t intern(const Poly &p) {
    auto it = poly_ids.find(p);
    if (it != poly_ids.end()) return it->second;
    work += p.size() + 1;
    if (work > (size_t)1 << 24) bad = true;   // a budget, so that no program can hold model creation up: the caller keeps its input
    polys.push_back(p);
    return poly_ids[p] = (uint32_t)polys.size() - 1;
  }
  uint32_t atom(const Atom &a) {
    auto it = atom_ids.find(a);
    if (it != atom_ids.end()) return it->second;
    atoms.push_back(a);
    return atom_ids[a] = (uint32_t)atoms.size() - 1;
  }
  uint32_t poly_dep(const Poly &p) const {
    for (auto &t : p) for (auto &f : t.first) if (atoms[f.first].dep) return atoms[f.first].dep;
    return 0;
  }
  static Poly constant(double v) { Poly p; if (v != 0.0) p.push_back({Mono(), v}); return p; }
  Poly of_atom(uint32_t a, int e = 1) const { Poly p; p.push_back({Mono{{a, e}}, 1.0}); return p; }

  static void normalize(Poly &p) {
    std::sort(p.begin(), p.end(), [](const std::pair<Mono, double> &x, const std::pair<Mono, double> &y) { return x.first < y.first; });
    Poly q;
    for (auto &t : p) {
      if (!q.empty() && q.back().first == t.first) q.back().second += t.second; else q.push_back(t);
    }
    p.clear();
    for (auto &t : q) if (t.second != 0.0) p.push_back(t);
  }
  static Poly add(const Poly &a, const Poly &b, double sb) {
    Poly r = a;
    for (auto &t : b) r.push_back({t.first, sb * t.second});
    normalize(r);
    return r;
  }
  static Mono mono_mul(const Mono &a, const Mono &b) {
    Mono r;
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
      if (j == b.size() || (i < a.size() && a[i].first < b[j].first)) r.push_back(a[i++]);
      else if (i == a.size() || b[j].first < a[i].first) r.push_back(b[j++]);
      else { const int e = a[i].second + b[j].second; if (e) r.push_back({a[i].first, e}); i++; j++; }
    }
    return r;
  }
  // P = s * Q with Q sign-/scale-normalised: leading coefficient +1 when all |coefficients| agree, else positive
  static double content_scale(Poly &p) {
    if (p.empty()) return 1.0;
    bool equal = true;
    const double c0 = p[0].second;
    for (auto &t : p) if (std::fabs(t.second) != std::fabs(c0)) equal = false;
    const double s = equal ? c0 : (c0 < 0 ? -1.0 : 1.0);
    if (s != 1.0) for (auto &t : p) t.second = equal ? (t.second == c0 ? 1.0 : -1.0) : -t.second;
    return s;
  }
  // a multi-term polynomial as ONE atom (sign-normalised): returns the polynomial  s * atom^e
  Poly sum_atom(Poly p, int e) {
    const double s = content_scale(p);
    Atom a; a.op = SUM_ATOM; a.kids.push_back(intern(p)); a.dep = poly_dep(p);
    Poly r = of_atom(atom(a), e);
    double se = 1.0;
    for (int i = 0; i < std::abs(e); i++) se *= s;
    r[0].second = e >= 0 ? se : 1.0 / se;
    return r;
  }
  Poly as_factor(const Poly &p, int e) {  // p^e as a polynomial without expanding a sum (callers exclude 0^negative)
    if (p.empty()) return Poly();
    if (p.size() == 1) {
      Mono m;
      for (auto &f : p[0].first) m.push_back({f.first, f.second * e});
      double c = 1.0;
      for (int i = 0; i < std::abs(e); i++) c *= p[0].second;
      Poly r; r.push_back({m, e >= 0 ? c : 1.0 / c});
      return r;
    }
    return sum_atom(p, e);
  }
  Poly mul(const Poly &a, const Poly &b) {
    if (a.empty() || b.empty()) return Poly();
    if (a.size() > 1 && b.size() > 1) {  // sum x sum: never expanded ...
      // ... except a parameter-only sum against a row-level one: the invariant becomes ONE atom and multiplies every row term
      // (the common-factor extraction pulls it out again; the row terms -- e.g. the 8 slots of Model.observe -- stay apart)
      const bool ra = poly_dep(a) != 0, rb = poly_dep(b) != 0;
      if (ra != rb) return ra ? mul(a, sum_atom(b, 1)) : mul(sum_atom(a, 1), b);
      const Poly fa = sum_atom(a, 1), fb = sum_atom(b, 1);
      Poly r; r.push_back({mono_mul(fa[0].first, fb[0].first), fa[0].second * fb[0].second});
      return r;
    }
    Poly r;
    for (auto &x : a) for (auto &y : b) r.push_back({mono_mul(x.first, y.first), x.second * y.second});
    normalize(r);
    return r;
  }

  // operands first: nodes are in topological order and compute_polys() walks them ascending, so no recursion happens here
  Poly poly_of(uint32_t id) {
    if (node_poly[id] >= 0) return polys[(size_t)node_poly[id]];
    const Node &nd = P.nodes[id];
    Poly r;
    switch (nd.op) {
      case RH_RIR_CONST: r = constant(nd.cval); break;
      case RH_RIR_INPUT: { Atom a; a.op = RH_RIR_INPUT; a.input = nd.input; a.dep = nd.dep; r = of_atom(atom(a)); break; }
      case RH_RIR_ADD: r = add(poly_of(nd.a), poly_of(nd.b), 1.0); break;
      case RH_RIR_SUB: r = add(poly_of(nd.a), poly_of(nd.b), -1.0); break;
      case RH_RIR_MUL: { const Poly a = poly_of(nd.a), b = poly_of(nd.b); r = mul(a, b); break; }
      case RH_RIR_DIV: {
        const Poly a = poly_of(nd.a), b = poly_of(nd.b);
        if (b.empty()) r = opaque2(nd); else r = mul(a, as_factor(b, -1));
        break;
      }
      case RH_RIR_POW: {
        const Node &e = P.nodes[nd.b];
        const double ev = e.cval;
        if (e.op == RH_RIR_CONST && ev == std::floor(ev) && std::fabs(ev) <= 8.0 && ev != 0.0 && !poly_of(nd.a).empty()) {
          const Poly a = poly_of(nd.a);
          r = as_factor(a, (int)ev);
        } else r = opaque2(nd);
        break;
      }
      case RH_RIR_SEQ: r = poly_of(nd.b); break;   // the VarDef side is ordinary sharing in a DAG
      case RH_RIR_COMPARE: r = opaque2(nd); break;
      case RH_RIR_LOOKUP: {
        Atom a; a.op = RH_RIR_LOOKUP; a.low = nd.low;
        const Poly ix = poly_of(nd.a);
        a.kids.push_back(intern(ix)); a.dep = poly_dep(ix);
        std::vector<Poly> ent;
        bool row_entries = false;
        for (uint32_t e : nd.table) { ent.push_back(poly_of(e)); row_entries = row_entries || poly_dep(ent.back()) != 0; }
        if (a.dep == 0 && row_entries) {
          // a select on a parameter-only index between row-level sums (what differentiating |sigma| or a bound check leaves:
          // eq(sigma, 0, 0, sum over all slots ...)) is the sum of its terms, each times a select between CONSTANTS:
          //     select(k, [sum_m c_0m m, sum_m c_1m m, ...]) = sum_m m * select(k, [c_0m, c_1m, ...])
          // -- exact, and the row terms of different slots stay apart instead of being locked into one atom
          std::map<Mono, std::vector<double>> coef;
          for (size_t j = 0; j < ent.size(); j++) for (auto &t : ent[j]) { auto &v = coef[t.first]; v.resize(ent.size(), 0.0); v[j] = t.second; }
          for (auto &kv : coef) {
            bool same = true;
            for (double c : kv.second) same = same && c == kv.second[0];
            if (same) { r.push_back({kv.first, kv.second[0]}); continue; }
            Atom sel = a;
            for (double c : kv.second) sel.kids.push_back(intern(constant(c)));
            r.push_back({mono_mul(kv.first, Mono{{atom(sel), 1}}), 1.0});
          }
          normalize(r);
          break;
        }
        for (const Poly &k : ent) { a.kids.push_back(intern(k)); if (poly_dep(k)) a.dep = poly_dep(k); }
        r = of_atom(atom(a));
        break;
      }
      default: {  // unary ops, NOOP included (an affine column image stays atomic)
        Atom a; a.op = nd.op;
        const Poly k = poly_of(nd.a);
        a.kids.push_back(intern(k)); a.dep = poly_dep(k);
        r = of_atom(atom(a));
      }
    }
    for (auto &t : r) if (!std::isfinite(t.second)) bad = true;
    node_poly[id] = (int)intern(r);
    return r;
  }
  bool bad = false;  // a folded coefficient overflowed / became NaN, or the work budget ran out: the caller keeps the original program
  size_t work = 0;
  void compute_polys(const std::vector<char> &live) {
    for (uint32_t i = 0; i < P.nodes.size(); i++) if (live[i]) (void)poly_of(i);
  }
  Poly opaque2(const Node &nd) {
    Atom a; a.op = nd.op;
    const Poly x = poly_of(nd.a), y = poly_of(nd.b);
    a.kids.push_back(intern(x)); a.kids.push_back(intern(y));
    a.dep = poly_dep(x) ? poly_dep(x) : poly_dep(y);
    return of_atom(atom(a));
  }

  // ---- slot rolling ----------------------------------------------------------------------------------------------------
  // Model.observe cuts the observations into an initial chunk and 8 contiguous splits and sums the splits' log-densities as
  // ONE expression (core/Model.scala:71-132): a "row" of that target carries the columns of 8 observations, and the
  // reference's Line algebra has merged the parameter-only terms of the 8 copies (8 * log sigma ...).  In normal form the
  // structure is plain to see: the monomials of every output fall into S groups that touch disjoint column sets of equal
  // size, plus column-free ("shared") monomials.  If renaming group s's columns onto group 1's (in column order) turns
  // every output's group-s part into its group-1 part, the target is rolled back:  S times the rows, 1/S of the columns,
  //      out = shared / S + part_1          (S = 8: the division is exact)
  // and the initial chunk Model.observe writes as a row target of its own (1-8 observations), whose outputs are that same
  // function of its own columns, is appended as further rows -- no observation ends up in the generated source.
  std::map<uint32_t, std::vector<uint32_t>> acols;   // atom -> the column inputs it reaches (sorted)
  const std::vector<uint32_t> &atom_cols(uint32_t a) {
    auto it = acols.find(a);
    if (it != acols.end()) return it->second;
    std::vector<uint32_t> r;
    const Atom A = atoms[a];
    if (A.op == RH_RIR_INPUT) { if (A.input >= P.n_params) r.push_back(A.input); }
    else for (uint32_t kid : A.kids) { const std::vector<uint32_t> v = poly_cols(polys[kid]); r.insert(r.end(), v.begin(), v.end()); }
    std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end());
    return acols[a] = r;
  }
  std::vector<uint32_t> mono_cols(const Mono &m) {
    std::vector<uint32_t> r;
    for (auto &f : m) if (atoms[f.first].dep) { const std::vector<uint32_t> &v = atom_cols(f.first); r.insert(r.end(), v.begin(), v.end()); }
    std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end());
    return r;
  }
  std::vector<uint32_t> poly_cols(const Poly p) {
    std::vector<uint32_t> r;
    for (auto &t : p) { const std::vector<uint32_t> v = mono_cols(t.first); r.insert(r.end(), v.begin(), v.end()); }
    std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end());
    return r;
  }
  std::map<uint32_t, bool> aparam;   // atom -> does it read a parameter
  bool atom_has_param(uint32_t a) {
    auto it = aparam.find(a);
    if (it != aparam.end()) return it->second;
    const Atom A = atoms[a];
    bool r = false;
    if (A.op == RH_RIR_INPUT) r = A.input < P.n_params;
    else for (uint32_t kid : A.kids) { const Poly kp = polys[kid]; for (auto &t : kp) for (auto &f : t.first) if (atom_has_param(f.first)) r = true; }
    return aparam[a] = r;
  }
  bool mono_has_param(const Mono &m) { for (auto &f : m) if (atom_has_param(f.first)) return true; return false; }
  // column-blind structural hashes (every data column hashes alike): they order a slot's monomials and atoms the same way in
  // every slot, so that a walk in that order meets corresponding columns in the same sequence
  std::map<uint32_t, uint64_t> hatom;
  static uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xBF58476D1CE4E5B9ull; }
  uint64_t h_atom(uint32_t a) {
    auto it = hatom.find(a);
    if (it != hatom.end()) return it->second;
    const Atom A = atoms[a];
    uint64_t h = mix(0x1234, A.op);
    if (A.op == RH_RIR_INPUT) h = A.input >= P.n_params ? mix(h, 0xC01) : mix(mix(h, 0x9A7), A.input);
    else { h = mix(h, (uint64_t)(int64_t)A.low); for (uint32_t kid : A.kids) h = mix(h, h_poly(polys[kid])); }
    return hatom[a] = h;
  }
  uint64_t h_mono(const Mono &m) {
    std::vector<uint64_t> v;
    for (auto &f : m) v.push_back(mix(h_atom(f.first), (uint64_t)(int64_t)f.second));
    std::sort(v.begin(), v.end());
    uint64_t h = 0x77;
    for (uint64_t x : v) h = mix(h, x);
    return h;
  }
  uint64_t h_poly(const Poly p) {
    std::vector<uint64_t> v;
    for (auto &t : p) { uint64_t b; std::memcpy(&b, &t.second, 8); v.push_back(mix(h_mono(t.first), b)); }
    std::sort(v.begin(), v.end());
    uint64_t h = 0x99;
    for (uint64_t x : v) h = mix(h, x);
    return h;
  }
  void walk_atom(uint32_t a, std::vector<uint32_t> &order, std::map<uint32_t, char> &seen) {
    const Atom A = atoms[a];
    if (A.op == RH_RIR_INPUT) { if (A.input >= P.n_params && !seen.count(A.input)) { seen[A.input] = 1; order.push_back(A.input); } return; }
    if (A.dep == 0) return;
    const uint32_t key = 0x80000000u | a;            // atoms are visited once per walk (shared sub-atoms would be re-walked 2^depth times)
    if (seen.count(key)) return;
    seen[key] = 1;
    for (uint32_t kid : A.kids) walk_poly(polys[kid], order, seen);
  }
  void walk_poly(const Poly p, std::vector<uint32_t> &order, std::map<uint32_t, char> &seen) {
    std::vector<std::pair<uint64_t, size_t>> ms;
    for (size_t i = 0; i < p.size(); i++) ms.push_back({h_mono(p[i].first), i});
    std::sort(ms.begin(), ms.end());
    for (auto &mi : ms) {
      std::vector<std::pair<uint64_t, uint32_t>> as;
      for (auto &f : p[mi.second].first) as.push_back({h_atom(f.first), f.first});
      std::sort(as.begin(), as.end());
      for (auto &x : as) walk_atom(x.second, order, seen);
    }
  }
  typedef std::map<uint32_t, uint32_t> ColMap;
  uint32_t rename_atom(uint32_t a, const ColMap &cmap, std::map<uint32_t, uint32_t> &memo) {
    auto it = memo.find(a);
    if (it != memo.end()) return it->second;
    Atom A = atoms[a];
    if (A.dep == 0) return a;
    if (A.op == RH_RIR_INPUT) { auto c = cmap.find(A.input); if (c != cmap.end()) A.input = c->second; }
    else for (uint32_t &kid : A.kids) kid = intern(rename_poly(polys[kid], cmap, memo));
    const uint32_t r = atom(A);
    return memo[a] = r;
  }
  Poly rename_poly(const Poly p, const ColMap &cmap, std::map<uint32_t, uint32_t> &memo) {
    Poly r;
    for (auto &t : p) {
      Mono m;
      for (auto &f : t.first) m.push_back({rename_atom(f.first, cmap, memo), f.second});
      std::sort(m.begin(), m.end());
      r.push_back({m, t.second});
    }
    normalize(r);
    return r;
  }
  static bool approx_equal(const Poly &a, const Poly &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
      if (a[i].first != b[i].first || std::fabs(a[i].second - b[i].second) > 1e-13 * std::fabs(b[i].second)) return false;
    return true;
  }
  struct Rolled {
    bool ok = false;
    std::vector<std::vector<uint32_t>> slots;   // [slot][i]: column input ids, position i corresponds across slots
    std::vector<Poly> outs;                     // per output: shared / S + part_1 + loose data terms
    std::vector<uint32_t> loose;                // columns of parameter-free terms outside every slot (the reference's Line keeps the
                                                // sum of all data-only terms as ONE column): kept, padded with zeros for the other slots
    int init_target = -1;
    std::vector<uint32_t> init_cols;            // the initial chunk's columns, same positions
  };
  // RH_ROLL_WHY=1: say on stderr why a target with several column groups was not rolled
  static Rolled why(Rolled R, const char *reason) {
    if (rh::knob("RH_ROLL_WHY")) std::fprintf(stderr, "rainier-hip: target not rolled back into rows: %s\n", reason);
    return R;
  }
  Rolled try_roll(size_t t, const std::vector<char> &taken) {
    Rolled R;
    const Target &T = P.targets[t];
    const size_t no = T.outputs.size();
    std::vector<Poly> po(no);
    for (size_t o = 0; o < no; o++) po[o] = poly_of(T.outputs[o]);
    // union-find over the target's columns
    std::map<uint32_t, uint32_t> parent;
    std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { auto it = parent.find(x); if (it == parent.end()) { parent[x] = x; return x; } if (it->second == x) return x; const uint32_t r = find(it->second); parent[x] = r; return r; };
    for (auto &p : po) for (auto &m : p) { const std::vector<uint32_t> c = mono_cols(m.first); for (size_t i = 1; i < c.size(); i++) parent[find(c[i])] = find(c[0]); if (!c.empty()) find(c[0]); }
    std::map<uint32_t, std::vector<uint32_t>> comp;
    for (auto &kv : parent) comp[find(kv.first)].push_back(kv.first);
    // every component belongs to a structural class (column-blind hash of its monomials in every output).  A slot is one
    // connected component when the observation's density ties its columns together (the usual case), or one component of every
    // class when it is additively separable (sum_s A(x_s) + B(z_s): any pairing of the A and B terms into rows gives the same
    // sum; the s-th of each class, in column order, is taken).  Classes that parameters reach fix the slot count S; a
    // parameter-free class with S members rolls like any other; a parameter-free component outside such a class is a LOOSE
    // data term -- the reference's Line keeps the sum of all data-only terms as one column -- and may only be kept, zero-padded
    // for the other slots, when it is linear in its column (coefficient * column: it vanishes on the padding)
    std::map<uint32_t, char> has_param_comp;
    for (auto &p : po) for (auto &m : p) { const std::vector<uint32_t> c = mono_cols(m.first); if (!c.empty() && mono_has_param(m.first)) has_param_comp[find(c[0])] = 1; }
    std::map<uint32_t, char> is_loose;
    std::vector<std::vector<uint32_t>> comps;
    std::vector<char> comp_param;
    for (auto &kv : comp) { std::sort(kv.second.begin(), kv.second.end()); comps.push_back(kv.second); comp_param.push_back(has_param_comp.count(kv.first) ? 1 : 0); }
    {
      std::vector<size_t> idx(comps.size());
      for (size_t c = 0; c < comps.size(); c++) idx[c] = c;
      std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return comps[a][0] < comps[b][0]; });
      std::vector<std::vector<uint32_t>> cs; std::vector<char> cp;
      for (size_t c : idx) { cs.push_back(comps[c]); cp.push_back(comp_param[c]); }
      comps.swap(cs); comp_param.swap(cp);
    }
    {
      std::map<uint32_t, size_t> comp_of;
      for (size_t c = 0; c < comps.size(); c++) for (uint32_t col : comps[c]) comp_of[col] = c;
      std::vector<uint64_t> ch(comps.size(), 0x5107);
      std::vector<char> linear(comps.size(), 1);    // all its monomials are  coefficient * column
      for (size_t o = 0; o < no; o++) {
        std::vector<Poly> part(comps.size());
        for (auto &m : po[o]) {
          const std::vector<uint32_t> c = mono_cols(m.first);
          if (c.empty()) continue;
          part[comp_of[c[0]]].push_back(m);
          if (!(m.first.size() == 1 && m.first[0].second == 1 && atoms[m.first[0].first].op == RH_RIR_INPUT)) linear[comp_of[c[0]]] = 0;
        }
        for (size_t c = 0; c < comps.size(); c++) ch[c] = mix(mix(ch[c], o), h_poly(part[c]));
      }
      std::map<uint64_t, std::vector<size_t>> classes;
      for (size_t c = 0; c < comps.size(); c++) classes[ch[c]].push_back(c);
      size_t S0 = 0;
      if (rh::knob("RH_ROLL_WHY"))
        for (auto &kv : classes)
          std::fprintf(stderr, "rainier-hip:   class of %zu component(s), %zu column(s) each, %s, first column input %u\n", kv.second.size(),
                       comps[kv.second[0]].size(), comp_param[kv.second[0]] ? "reached by parameters" : "data only", comps[kv.second[0]][0]);
      for (auto &kv : classes) {
        if (!comp_param[kv.second[0]]) continue;
        if (S0 == 0) S0 = kv.second.size(); else if (kv.second.size() != S0) return why(R, "structural classes of unequal size");
      }
      if (S0 < 2) return why(R, "fewer than two structurally equal components");
      R.slots.assign(S0, {});
      for (auto &kv : classes) {
        if (kv.second.size() == S0) { for (size_t s = 0; s < S0; s++) for (uint32_t col : comps[kv.second[s]]) R.slots[s].push_back(col); continue; }
        for (size_t c : kv.second) {
          if (comp_param[c] || !linear[c]) return why(R, "a data-only term is neither one per slot nor linear in its column");
          for (uint32_t col : comps[c]) { R.loose.push_back(col); is_loose[col] = 1; }
        }
      }
    }
    if (R.slots.size() < 2) return why(R, "fewer than two slots");
    std::sort(R.slots.begin(), R.slots.end(), [](const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) { return a[0] < b[0]; });
    std::sort(R.loose.begin(), R.loose.end());
    const size_t S = R.slots.size(), mcols = R.slots[0].size();
    for (auto &sl : R.slots) if (sl.size() != mcols) return why(R, "slots of unequal width");
    std::map<uint32_t, size_t> slot_of;
    for (size_t s = 0; s < S; s++) for (uint32_t c : R.slots[s]) slot_of[c] = s;
    {  // corresponding columns by structure, not by position in the column list
      std::vector<std::vector<uint32_t>> order(S);
      std::vector<std::map<uint32_t, char>> seen(S);
      for (size_t o = 0; o < no; o++) {
        std::vector<Poly> part(S);
        for (auto &m : po[o]) { const std::vector<uint32_t> c = mono_cols(m.first); if (!c.empty() && !is_loose.count(c[0])) part[slot_of[c[0]]].push_back(m); }
        for (size_t s = 0; s < S; s++) walk_poly(part[s], order[s], seen[s]);
      }
      for (size_t s = 0; s < S; s++) if (order[s].size() != mcols) return why(R, "the structural walk did not reach every column of a slot");
      R.slots = order;
    }
    std::vector<ColMap> cmap(S);
    for (size_t s = 1; s < S; s++) for (size_t i = 0; i < mcols; i++) cmap[s][R.slots[s][i]] = R.slots[0][i];
    std::vector<std::map<uint32_t, uint32_t>> memo(S);
    for (size_t o = 0; o < no; o++) {
      Poly shared, loose; std::vector<Poly> part(S);
      for (auto &m : po[o]) {
        const std::vector<uint32_t> c = mono_cols(m.first);
        if (c.empty()) shared.push_back(m); else if (is_loose.count(c[0])) loose.push_back(m); else part[slot_of[c[0]]].push_back(m);
      }
      for (size_t s = 1; s < S; s++) if (!approx_equal(rename_poly(part[s], cmap[s], memo[s]), part[0])) return why(R, "an output's part of one slot is not the renamed part of the first slot");
      Poly out = part[0];
      for (auto &m : shared) out.push_back({m.first, m.second / (double)S});
      for (auto &m : loose) out.push_back(m);
      normalize(out);
      R.outs.push_back(out);
    }
    // an initial chunk: another row target whose outputs are the same function of its own columns
    for (size_t t0 = 0; t0 < P.targets.size() && R.init_target < 0 && R.loose.empty(); t0++) {
      const Target &T0 = P.targets[t0];
      if (t0 == t || !T0.n_cols || taken[t0] || T0.outputs.size() != no) continue;
      std::vector<Poly> p0(no);
      std::vector<uint32_t> used;                 // its columns in the order the same structural walk meets them
      {
        std::map<uint32_t, char> seen;
        for (size_t o = 0; o < no; o++) {
          p0[o] = poly_of(T0.outputs[o]);
          Poly withcols;
          for (auto &m : p0[o]) if (!mono_cols(m.first).empty()) withcols.push_back(m);
          walk_poly(withcols, used, seen);
        }
      }
      if (used.size() != mcols) continue;
      ColMap c0; for (size_t i = 0; i < mcols; i++) c0[used[i]] = R.slots[0][i];
      std::map<uint32_t, uint32_t> mm;
      bool same = true;
      for (size_t o = 0; o < no && same; o++) same = approx_equal(rename_poly(p0[o], c0, mm), R.outs[o]);
      if (same) { R.init_target = (int)t0; R.init_cols = used; }
    }
    R.ok = true;
    return R;
  }

  // ---- rebuilding -------------------------------------------------------------------------------------------------------
  Program Q;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cons;
  std::map<uint64_t, uint32_t> consts;
  std::map<uint32_t, uint32_t> inputs;
  std::map<uint32_t, uint32_t> built_poly, built_atom;

  uint32_t push(const Node &n) { Q.nodes.push_back(n); return (uint32_t)Q.nodes.size() - 1; }
  uint32_t k(double v) {
    uint64_t b; std::memcpy(&b, &v, 8);
    auto it = consts.find(b);
    if (it != consts.end()) return it->second;
    Node n; n.op = RH_RIR_CONST; n.cval = v;
    return consts[b] = push(n);
  }
  uint32_t op2(uint32_t op, uint32_t a, uint32_t b) {
    auto key = std::make_tuple(op, a, b);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.b = b; n.dep = Q.nodes[a].dep ? Q.nodes[a].dep : Q.nodes[b].dep;
    return cons[key] = push(n);
  }
  uint32_t op1(uint32_t op, uint32_t a) {
    auto key = std::make_tuple(op, a, 0xffffffffu);
    auto it = cons.find(key);
    if (it != cons.end()) return it->second;
    Node n; n.op = op; n.a = a; n.dep = Q.nodes[a].dep;
    return cons[key] = push(n);
  }
  uint32_t build_atom(uint32_t ai) {
    auto it = built_atom.find(ai);
    if (it != built_atom.end()) return it->second;
    const Atom a = atoms[ai];
    uint32_t id;
    if (a.op == RH_RIR_INPUT) {
      auto ii = inputs.find(a.input);
      if (ii != inputs.end()) id = ii->second;
      else { Node n; n.op = RH_RIR_INPUT; n.input = a.input; n.dep = a.dep; id = inputs[a.input] = push(n); }
    } else if (a.op == SUM_ATOM) id = build_poly(a.kids[0]);
    else if (a.op == RH_RIR_LOOKUP) {
      Node n; n.op = RH_RIR_LOOKUP; n.low = a.low;
      std::vector<uint32_t> ks;
      for (uint32_t kid : a.kids) ks.push_back(build_poly(kid));
      n.a = ks[0]; n.table.assign(ks.begin() + 1, ks.end());
      n.dep = 0;
      for (uint32_t x : ks) if (Q.nodes[x].dep) n.dep = Q.nodes[x].dep;
      id = push(n);
    } else if (a.kids.size() == 2) { const uint32_t x = build_poly(a.kids[0]), y = build_poly(a.kids[1]); id = op2(a.op, x, y); }
    else id = op1(a.op, build_poly(a.kids[0]));
    return built_atom[ai] = id;
  }
  uint32_t power(uint32_t x, int e) {  // e >= 1
    uint32_t r = x;
    if (e == 2) return op2(RH_RIR_MUL, x, x);
    if (e == 4) { const uint32_t sq = op2(RH_RIR_MUL, x, x); return op2(RH_RIR_MUL, sq, sq); }
    for (int i = 1; i < e; i++) r = op2(RH_RIR_MUL, r, x);
    return r;
  }
  // product of the atoms of m selected by (row-dependent?, positive exponent?); 0xFFFFFFFF = empty product
  uint32_t part(const Mono &m, bool row, bool pos) {
    uint32_t r = 0xFFFFFFFFu;
    for (auto &f : m) {
      if ((atoms[f.first].dep != 0) != row || (f.second > 0) != pos) continue;
      const uint32_t x = power(build_atom(f.first), std::abs(f.second));
      r = r == 0xFFFFFFFFu ? x : op2(RH_RIR_MUL, r, x);
    }
    return r;
  }
  uint32_t build_mono(double coef, const Mono &m) {
    uint32_t grp[2];
    for (int row = 0; row < 2; row++) {
      const uint32_t num = part(m, row != 0, true), den = part(m, row != 0, false);
      if (den == 0xFFFFFFFFu) grp[row] = num;
      else grp[row] = op2(RH_RIR_DIV, num == 0xFFFFFFFFu ? k(1.0) : num, den);
    }
    uint32_t r = grp[0];
    if (coef != 1.0) r = r == 0xFFFFFFFFu ? k(coef) : op2(RH_RIR_MUL, k(coef), r);
    if (grp[1] != 0xFFFFFFFFu) r = r == 0xFFFFFFFFu ? grp[1] : op2(RH_RIR_MUL, r, grp[1]);
    return r == 0xFFFFFFFFu ? k(1.0) : r;
  }
  uint32_t build_poly(uint32_t pi) {
    auto it = built_poly.find(pi);
    if (it != built_poly.end()) return it->second;
    Poly p = polys[pi];
    uint32_t id;
    if (p.empty()) id = k(0.0);
    else if (p.size() == 1) id = build_mono(p[0].second, p[0].first);
    else {
      // the monomial common to every term: atoms present everywhere with exponents of one sign, the smallest magnitude
      Mono F;
      for (auto &f : p[0].first) {
        int e = f.second;
        bool all = true;
        for (size_t t = 1; t < p.size() && all; t++) {
          int et = 0;
          for (auto &g : p[t].first) if (g.first == f.first) et = g.second;
          if (et == 0 || (et > 0) != (e > 0)) all = false; else if (std::abs(et) < std::abs(e)) e = et;
        }
        if (all) F.push_back({f.first, e});
      }
      if (!F.empty()) {
        Mono Finv;
        for (auto &f : F) Finv.push_back({f.first, -f.second});
        for (auto &t : p) t.first = mono_mul(t.first, Finv);
        normalize(p);
      }
      const double s = content_scale(p);
      if (!F.empty() || s != 1.0) {
        const uint32_t body = build_poly(intern(p));
        // data columns outermost, then the invariants, then the other row-level factors:  c_k * (alpha * (f * body)) -- the output
        // factoring peels alpha, and the x_k * w shape of a GLM gradient (w = f * body, the same node for every k) is in the open
        Mono Finvar, Fcol, Frow;
        for (auto &f : F) {
          const Atom &a = atoms[f.first];
          (a.dep == 0 ? Finvar : (a.op == RH_RIR_INPUT && f.second > 0 ? Fcol : Frow)).push_back(f);
        }
        uint32_t r = body;
        if (!Frow.empty()) r = op2(RH_RIR_MUL, build_mono(1.0, Frow), r);
        if (!Finvar.empty() || s != 1.0) r = op2(RH_RIR_MUL, build_mono(s, Finvar), r);
        if (!Fcol.empty()) r = op2(RH_RIR_MUL, build_mono(1.0, Fcol), r);
        id = r;
      } else {
        // plain sum in normal-form order, positive terms first so that the fold never starts with a negation
        std::vector<size_t> order;
        for (size_t t = 0; t < p.size(); t++) if (p[t].second > 0) order.push_back(t);
        for (size_t t = 0; t < p.size(); t++) if (!(p[t].second > 0)) order.push_back(t);
        uint32_t acc = 0xFFFFFFFFu;
        for (size_t t : order) {
          const double c = p[t].second;
          if (acc == 0xFFFFFFFFu) acc = build_mono(c, p[t].first);
          else if (c < 0) acc = op2(RH_RIR_SUB, acc, build_mono(-c, p[t].first));
          else acc = op2(RH_RIR_ADD, acc, build_mono(c, p[t].first));
        }
        id = acc;
      }
    }
    return built_poly[pi] = id;
  }
};

void mark_live(const Program &P, const std::vector<uint32_t> &roots, std::vector<char> &live) {
  live.assign(P.nodes.size(), 0);
  for (uint32_t o : roots) live[o] = 1;
  for (size_t i = P.nodes.size(); i-- > 0;) {
    if (!live[i]) continue;
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    live[n.a] = 1;
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) live[e] = 1; }
    else if ((n.op >= RH_RIR_ADD && n.op <= RH_RIR_COMPARE) || n.op == RH_RIR_SEQ) live[n.b] = 1;
  }
}
double row_cost(const Program &P, const Target &T) {
  std::vector<char> live;
  mark_live(P, T.outputs, live);
  double c = 0;
  for (size_t i = 0; i < P.nodes.size(); i++) {
    if (!live[i] || P.nodes[i].dep == 0) continue;
    switch (P.nodes[i].op) {
      case RH_RIR_CONST: case RH_RIR_INPUT: case RH_RIR_NOOP: case RH_RIR_SEQ: break;
      case RH_RIR_DIV: c += 8; break;
      case RH_RIR_POW: c += 60; break;
      case RH_RIR_EXP: case RH_RIR_LOG: c += 30; break;
      case RH_RIR_LOOKUP: c += (double)P.nodes[i].table.size(); break;
      default: c += (P.nodes[i].op >= RH_RIR_SIN) ? 60 : 1;
    }
  }
  return c;
}

}  // namespace

Program refactor(const Program &P, std::vector<std::vector<uint32_t>> *parts) {
  Refactor R(P);
  R.Q.n_params = P.n_params; R.Q.n_inputs = P.n_inputs; R.Q.n_cols_total = P.n_cols_total; R.Q.kind = P.kind;
  R.Q.targets = P.targets; R.Q.col_domain = P.col_domain;
  if (parts) { parts->clear(); for (uint32_t c = 0; c < P.n_cols_total; c++) parts->push_back({c}); }
  // 1) every node of the input program is copied (so that data-free targets and fall-backs keep their exact structure) ...
  std::vector<uint32_t> m(P.nodes.size());
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    switch (n.op) {
      case RH_RIR_CONST: m[i] = R.k(n.cval); break;
      case RH_RIR_INPUT: {
        auto it = R.inputs.find(n.input);
        if (it != R.inputs.end()) m[i] = it->second;
        else { Node q; q.op = RH_RIR_INPUT; q.input = n.input; q.dep = n.dep; m[i] = R.inputs[n.input] = R.push(q); }
        break;
      }
      case RH_RIR_LOOKUP: {
        Node q = n; q.a = m[n.a];
        for (uint32_t &e : q.table) e = m[e];
        m[i] = R.push(q);
        break;
      }
      case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_POW: case RH_RIR_COMPARE: case RH_RIR_SEQ:
        m[i] = R.op2(n.op, m[n.a], m[n.b]); break;
      default: m[i] = R.op1(n.op, m[n.a]);
    }
  }
  // 2) ... and the row targets are rebuilt from their normal forms behind them
  {
    std::vector<uint32_t> roots;
    for (const Target &T : P.targets) if (T.n_cols) for (uint32_t o : T.outputs) roots.push_back(o);
    std::vector<char> live;
    mark_live(P, roots, live);
    R.compute_polys(live);
    if (R.bad) return P;
  }
  const size_t NT = P.targets.size();
  std::vector<char> merged(NT, 0);                                 // initial-chunk targets appended to a rolled target
  std::vector<std::vector<uint32_t>> keep_cols(NT);                // per target: the old input ids it keeps, in order
  std::vector<std::vector<std::vector<uint32_t>>> col_src(NT);     // per kept column: the old input ids concatenated into it
  for (size_t t = 0; t < NT; t++) {
    Target &T = R.Q.targets[t];
    for (uint32_t &o : T.outputs) o = m[o];
    for (uint32_t j = 0; j < P.targets[t].n_cols; j++) { keep_cols[t].push_back(P.targets[t].input_start + j); col_src[t].push_back({P.targets[t].input_start + j}); }
  }
  for (size_t t = 0; t < NT; t++) {
    Target &T = R.Q.targets[t];
    if (!T.n_cols || merged[t]) continue;
    if (parts) {
      const Refactor::Rolled rr = R.try_roll(t, merged);
      if (rr.ok) {
        std::vector<uint32_t> rebuilt;
        for (const Poly &p : rr.outs) rebuilt.push_back(R.build_poly(R.intern(p)));
        T.outputs = rebuilt;
        keep_cols[t] = rr.slots[0];
        col_src[t].assign(rr.slots[0].size(), {});
        for (size_t i = 0; i < rr.slots[0].size(); i++) {
          if (rr.init_target >= 0) col_src[t][i].push_back(rr.init_cols[i]);
          for (auto &sl : rr.slots) col_src[t][i].push_back(sl[i]);
        }
        for (uint32_t c : rr.loose) {   // [its data, zeros for the rows of the other slots]
          keep_cols[t].push_back(c);
          std::vector<uint32_t> cs{c};
          for (size_t sl = 1; sl < rr.slots.size(); sl++) cs.push_back(ZERO_PART);
          col_src[t].push_back(cs);
        }
        if (rr.init_target >= 0) {
          const size_t t0 = (size_t)rr.init_target;
          merged[t0] = 1; keep_cols[t0].clear(); col_src[t0].clear();
          for (uint32_t &o : R.Q.targets[t0].outputs) o = R.k(0.0);
        }
        continue;
      }
    }
    std::vector<uint32_t> rebuilt;
    for (uint32_t o : P.targets[t].outputs) rebuilt.push_back(R.build_poly(R.intern(R.poly_of(o))));
    Target cand = T; cand.outputs = rebuilt;
    if (row_cost(R.Q, cand) <= row_cost(R.Q, T)) T.outputs = rebuilt;
  }
  if (R.bad) return P;     // (parts still holds the identity)
  if (!parts) return R.Q;
  // 3) renumber the columns that are left (dropped ones only survive in dead nodes, which the clean-up pass sweeps); columns no
  //    output reads any more (the reference's gradient-only columns after re-derivation) go as well
  Program &Q = R.Q;
  {
    std::vector<uint32_t> roots;
    for (const Target &T : Q.targets) for (uint32_t o : T.outputs) roots.push_back(o);
    std::vector<char> live;
    mark_live(Q, roots, live);
    std::vector<char> used(P.n_inputs, 0);
    for (size_t i = 0; i < Q.nodes.size(); i++) if (live[i] && Q.nodes[i].op == RH_RIR_INPUT && Q.nodes[i].input < P.n_inputs) used[Q.nodes[i].input] = 1;
    for (size_t t = 0; t < NT; t++) {
      if (keep_cols[t].empty()) continue;
      std::vector<uint32_t> kc; std::vector<std::vector<uint32_t>> cs;
      for (size_t i = 0; i < keep_cols[t].size(); i++) if (used[keep_cols[t][i]]) { kc.push_back(keep_cols[t][i]); cs.push_back(col_src[t][i]); }
      if (kc.empty()) { kc.push_back(keep_cols[t][0]); cs.push_back(col_src[t][0]); }   // the row count hangs on a column
      keep_cols[t] = kc; col_src[t] = cs;
    }
  }
  std::map<uint32_t, uint32_t> renum;
  std::vector<std::vector<double>> dom;
  uint32_t in = P.n_params, colc = 0;
  parts->clear();
  for (size_t t = 0; t < NT; t++) {
    Q.targets[t].input_start = in; Q.targets[t].col0 = colc; Q.targets[t].n_cols = (uint32_t)keep_cols[t].size();
    for (size_t i = 0; i < keep_cols[t].size(); i++) {
      renum[keep_cols[t][i]] = in + (uint32_t)i;
      std::vector<uint32_t> src; std::vector<double> d; bool known = !P.col_domain.empty();
      for (uint32_t old : col_src[t][i]) {
        if (old == ZERO_PART) { src.push_back(ZERO_PART); if (std::find(d.begin(), d.end(), 0.0) == d.end()) d.push_back(0.0); continue; }
        const uint32_t g = old - P.n_params;
        src.push_back(g);
        if (known && g < P.col_domain.size() && !P.col_domain[g].empty()) { for (double v : P.col_domain[g]) if (std::find(d.begin(), d.end(), v) == d.end()) d.push_back(v); } else known = false;
      }
      if (!known || d.size() > 8) d.clear();
      std::sort(d.begin(), d.end());
      parts->push_back(src); dom.push_back(d);
    }
    in += Q.targets[t].n_cols; colc += Q.targets[t].n_cols;
  }
  Q.n_inputs = in; Q.n_cols_total = colc;
  if (!P.col_domain.empty()) Q.col_domain = dom;
  for (Node &n : Q.nodes)
    if (n.op == RH_RIR_INPUT && n.input >= P.n_params) {
      auto it = renum.find(n.input);
      if (it != renum.end()) n.input = it->second;
      else { n.op = RH_RIR_CONST; n.cval = 0.0; n.input = 0; n.dep = 0; }   // a dropped column: only dead nodes still name it
    }
  return Q;
}

}  // namespace rh
