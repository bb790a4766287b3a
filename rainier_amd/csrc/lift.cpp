// lift.cpp -- many data-free targets of one shape -> ONE streamed target (the inverse of the reference's constant folding).
//
// A model that calls Model.observe(y, dist) once per observation (bench/stan/ARK.scala: 195 times) hands over one data-free
// target per observation: the same expression 195 times with different constants folded in (y_t, y_{t-1} ... as literals).  As
// data-free code that is a 20 000-line straight-line function which every chain evaluates on one wavefront; as a streamed
// target it is 195 rows that the lanes share.  When a program arrives with more targets than the engine holds
// (RH_MAX_TARGETS), the data-free targets are grouped by their constant-blind structure; for the largest group (>= 32
// members) every member is matched against the first one node by node -- same operation, same parameter, constants may
// differ, and the node-to-node mapping must be a function (so shared sub-expressions are shared alike) -- and the constants
// that differ become COLUMNS of a new row target (one row per member) whose expression is the first member's with those
// constants replaced by column reads.  Per-row arithmetic is the member's own, operation for operation; only the order in
// which the members are added up becomes that of a row reduction.  Members that do not match stay data-free.  The step repeats
// while another group of >= 32 qualifies (two time series observed one value at a time become two streamed targets).  When no
// group differs in constants alone, families whose members differ in a parameter are looked for (lift_one, second half).
//
// The second half of the file prepares parameter TABLES for gather mode (emit.cpp detect_gather), again by undoing
// what the reference's front end folded: hoist_table_maps (the entries' common map moved behind the lookup), lift_table_priors
// (the table's prior, data-free as it comes, becomes a row target over the group index) and lift_single_entry_targets (a
// data-free expression over one entry -- Model.observe's one-row initial chunk -- becomes a one-row target).  Everything here
// runs in rh_model_create's loader (engine.cpp load_program), before the data-dependent passes; synthesised columns follow the
// caller's.
#include <stdexcept>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>

#include "../../include/rainier_hip_rir.h"
#include "device/rh_shared.h"
#include "rir.hpp"

namespace {
// std::map::at with the caller's line in the message (an internal invariant that did not hold: the model is then loaded without the lift)
template <class M, class K>
auto &map_at(M &m, const K &k, int line) {
  auto it = m.find(k);
  if (it == m.end()) throw std::out_of_range("lift.cpp:" + std::to_string(line) + ": node is not in the member's map");
  return it->second;
}
}  // namespace
#define MAP_AT(m, k) map_at(m, k, __LINE__)

namespace rh {
namespace {

bool binary_op(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }
uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xBF58476D1CE4E5B9ull; }

// member node b against template node a; memo: template node -> member node (must be consistent)
bool match(const Program &P, uint32_t a, uint32_t b, std::map<uint32_t, uint32_t> &memo, bool any_param = false) {
  std::vector<std::pair<uint32_t, uint32_t>> stack{{a, b}};
  while (!stack.empty()) {
    auto [x, y] = stack.back(); stack.pop_back();
    auto it = memo.find(x);
    if (it != memo.end()) { if (it->second != y) return false; continue; }
    const Node &nx = P.nodes[x], &ny = P.nodes[y];
    if (nx.op != ny.op) return false;
    memo[x] = y;
    switch (nx.op) {
      case RH_RIR_CONST: break;
      case RH_RIR_INPUT: if (nx.input != ny.input && !any_param) return false; break;
      case RH_RIR_LOOKUP:
        if (nx.low != ny.low || nx.table.size() != ny.table.size()) return false;
        stack.push_back({nx.a, ny.a});
        for (size_t e = 0; e < nx.table.size(); e++) stack.push_back({nx.table[e], ny.table[e]});
        break;
      default:
        stack.push_back({nx.a, ny.a});
        if (binary_op(nx.op)) stack.push_back({nx.b, ny.b});
    }
  }
  return true;
}

// The same for the per-entry terms of a parameter table: table parameter `from` of the template must correspond to `to` of the
// member, every other parameter to itself -- and the operands of + and * may come in either order (the reference's Line / LogLine
// algebra orders the factors of a product by its own bookkeeping: alpha_0 * mu next to mu * alpha_1; IEEE + and * commute exactly)
bool match_entry(const Program &P, uint32_t a, uint32_t b, std::map<uint32_t, uint32_t> &memo, uint32_t from, uint32_t to, uint32_t t0) {
  auto it = memo.find(a);
  if (it != memo.end()) return it->second == b;
  const Node &na = P.nodes[a], &nb = P.nodes[b];
  if (na.op != nb.op) return false;
  switch (na.op) {
    case RH_RIR_CONST: memo[a] = b; return true;
    case RH_RIR_INPUT:
      if (!(na.input == from ? nb.input == to : (na.input == nb.input && (na.input < t0 || na.input >= P.n_params)))) return false;
      memo[a] = b; return true;
    case RH_RIR_LOOKUP: {
      if (na.low != nb.low || na.table.size() != nb.table.size()) return false;
      memo[a] = b;
      if (!match_entry(P, na.a, nb.a, memo, from, to, t0)) return false;
      for (size_t e = 0; e < na.table.size(); e++) if (!match_entry(P, na.table[e], nb.table[e], memo, from, to, t0)) return false;
      return true;
    }
    default: {
      if (!binary_op(na.op)) { memo[a] = b; return match_entry(P, na.a, nb.a, memo, from, to, t0); }
      const std::map<uint32_t, uint32_t> saved = memo;
      memo[a] = b;
      if (match_entry(P, na.a, nb.a, memo, from, to, t0) && match_entry(P, na.b, nb.b, memo, from, to, t0)) return true;
      if (na.op != RH_RIR_ADD && na.op != RH_RIR_MUL) return false;
      memo = saved; memo[a] = b;
      return match_entry(P, na.a, nb.b, memo, from, to, t0) && match_entry(P, na.b, nb.a, memo, from, to, t0);
    }
  }
}

bool lift_one(Program &P, std::vector<std::vector<double>> &synth, std::vector<uint32_t> &old_target_of);

}  // namespace

void recompute_deps(Program &P) {
  for (Node &n : P.nodes) {
    uint32_t dep = 0;
    if (n.op == RH_RIR_INPUT) {
      if (n.input >= P.n_params)
        for (size_t t = 0; t < P.targets.size(); t++)
          if (n.input >= P.targets[t].input_start && n.input < P.targets[t].input_start + P.targets[t].n_cols) dep = (uint32_t)(t + 1);
    } else if (n.op != RH_RIR_CONST) {
      dep = P.nodes[n.a].dep;
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) if (P.nodes[e].dep) dep = P.nodes[e].dep; }
      else if (binary_op(n.op) && P.nodes[n.b].dep) dep = P.nodes[n.b].dep;
    }
    n.dep = dep;
  }
}

bool lift_constants(Program &P, std::vector<std::vector<double>> &synth, std::vector<uint32_t> &old_target_of) {
  synth.clear();
  old_target_of.clear();
  for (uint32_t t = 0; t < P.targets.size(); t++) old_target_of.push_back(t);
  if (P.kind != 0) return false;
  bool any = false;
  while (any || P.targets.size() > RH_MAX_TARGETS) {   // once a program needs it, every group that qualifies is lifted
    std::vector<uint32_t> step;                    // target after the step -> target before it, 0xFFFFFFFF = the new row target
    if (!lift_one(P, synth, step)) break;
    for (uint32_t &o : step) if (o != 0xFFFFFFFFu) o = old_target_of[o];
    old_target_of.swap(step);
    any = true;
  }
  return any;
}

namespace {
// one group; the new target's columns are appended to `synth` and the target itself to the end of the target list
bool lift_one(Program &P, std::vector<std::vector<double>> &synth, std::vector<uint32_t> &old_target_of) {
  // constant-blind structural hash of every node
  std::vector<uint64_t> h(P.nodes.size());
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    uint64_t x = mix(0x71, n.op);
    if (n.op == RH_RIR_CONST) x = mix(x, 0xC0);
    else if (n.op == RH_RIR_INPUT) x = mix(x, n.input);
    else {
      x = mix(x, h[n.a]);
      if (n.op == RH_RIR_LOOKUP) { x = mix(x, (uint64_t)(int64_t)n.low); for (uint32_t e : n.table) x = mix(x, h[e]); }
      else if (binary_op(n.op)) x = mix(x, h[n.b]);
    }
    h[i] = x;
  }
  std::map<uint64_t, std::vector<uint32_t>> groups;
  for (uint32_t t = 0; t < P.targets.size(); t++) {
    if (P.targets[t].n_cols) continue;
    uint64_t x = 0x3A;
    for (uint32_t o : P.targets[t].outputs) x = mix(x, h[o]);
    groups[x].push_back(t);
  }
  // candidate groups, largest first; the first one whose members really match (and differ in some constant) is lifted
  std::vector<const std::vector<uint32_t> *> cands;
  for (auto &kv : groups) if (kv.second.size() >= 32) cands.push_back(&kv.second);
  std::sort(cands.begin(), cands.end(), [](auto *a, auto *b) { return a->size() != b->size() ? a->size() > b->size() : (*a)[0] < (*b)[0]; });
  uint32_t tmpl = 0;
  std::vector<uint32_t> members, slots;
  std::vector<std::map<uint32_t, uint32_t>> maps;
  bool found = false;
  for (const std::vector<uint32_t> *cand : cands) {
    tmpl = (*cand)[0];
    const size_t no = P.targets[tmpl].outputs.size();
    members.clear(); maps.clear(); slots.clear();
    for (uint32_t g : *cand) {
      std::map<uint32_t, uint32_t> memo;
      bool ok = true;
      for (size_t o = 0; o < no && ok; o++) ok = match(P, P.targets[tmpl].outputs[o], P.targets[g].outputs[o], memo);
      if (ok) { members.push_back(g); maps.push_back(std::move(memo)); }
    }
    if (members.size() < 32) continue;
    // the template's constants that differ between members -> columns
    for (auto &kv : maps[0]) {
      if (P.nodes[kv.first].op != RH_RIR_CONST) continue;
      bool differs = false;
      const double v0 = P.nodes[kv.first].cval;
      for (size_t g = 1; g < members.size() && !differs; g++) {
        const double v = P.nodes[MAP_AT(maps[g], kv.first)].cval;
        differs = std::memcmp(&v, &v0, 8) != 0;
      }
      if (differs) slots.push_back(kv.first);
    }
    if (slots.empty() || slots.size() > 4096) continue;
    found = true;
    break;
  }
  // no family that differs in constants only: families whose members also differ in a PARAMETER (one Model.observe per group of
  // a hierarchical model).  Grouped by the parameter-blind structure of the value output; the parameter becomes a Lookup over a
  // lifted index column, and the gradient outputs -- which sit at different output positions from member to member -- become
  // eq(index, k, g, 0) terms (compute/Gradient.scala's derivative of a Lookup), which is what the emitter's scatter families read.
  std::vector<uint32_t> pslots;                                  // template parameters that vary (one INPUT node of each)
  std::vector<std::vector<uint32_t>> pnodes;                     // ... all INPUT nodes of each (seen from the value output)
  std::vector<std::vector<uint32_t>> ptable;                     // per parameter slot: the distinct parameters, ascending
  if (!found) {
    std::vector<uint64_t> h2(P.nodes.size());
    for (uint32_t i = 0; i < P.nodes.size(); i++) {
      const Node &n = P.nodes[i];
      uint64_t x = mix(0x72, n.op);
      if (n.op == RH_RIR_CONST) x = mix(x, 0xC0);
      else if (n.op == RH_RIR_INPUT) x = mix(x, 0x9A7);
      else {
        x = mix(x, h2[n.a]);
        if (n.op == RH_RIR_LOOKUP) { x = mix(x, (uint64_t)(int64_t)n.low); for (uint32_t e : n.table) x = mix(x, h2[e]); }
        else if (binary_op(n.op)) x = mix(x, h2[n.b]);
      }
      h2[i] = x;
    }
    std::map<uint64_t, std::vector<uint32_t>> groups2;
    for (uint32_t t = 0; t < P.targets.size(); t++) if (!P.targets[t].n_cols) groups2[h2[P.targets[t].outputs[0]]].push_back(t);
    cands.clear();
    for (auto &kv : groups2) if (kv.second.size() >= 32) cands.push_back(&kv.second);
    std::sort(cands.begin(), cands.end(), [](auto *a, auto *b) { return a->size() != b->size() ? a->size() > b->size() : (*a)[0] < (*b)[0]; });
    auto is_zero = [&](uint32_t id) { return P.nodes[id].op == RH_RIR_CONST && P.nodes[id].cval == 0.0; };
    const uint32_t np = P.n_params;
    const bool why = rh::knob("RH_LIFT_WHY") != nullptr;
    if (why) std::fprintf(stderr, "lift: %zu parameter-blind candidate groups\n", cands.size());
    for (const std::vector<uint32_t> *cand : cands) {
      tmpl = (*cand)[0];
      const std::vector<uint32_t> &to = P.targets[tmpl].outputs;
      // pass 1: value outputs -> which parameter nodes vary
      std::vector<uint32_t> ok1;
      std::vector<std::map<uint32_t, uint32_t>> m1;
      for (uint32_t g : *cand) {
        std::map<uint32_t, uint32_t> memo;
        if (match(P, to[0], P.targets[g].outputs[0], memo, true)) { ok1.push_back(g); m1.push_back(std::move(memo)); }
      }
      if (why) std::fprintf(stderr, "lift: group of %zu, %zu match the template's value\n", cand->size(), ok1.size());
      if (ok1.size() < 32) continue;
      pslots.clear(); pnodes.clear();
      std::vector<char> fixed_param(np, 0), varies(np, 0);
      for (auto &kv : m1[0]) {   // (a parameter may have several INPUT nodes)
        if (P.nodes[kv.first].op != RH_RIR_INPUT) continue;
        for (size_t g = 1; g < ok1.size() && !varies[P.nodes[kv.first].input]; g++)
          varies[P.nodes[kv.first].input] = P.nodes[MAP_AT(m1[g], kv.first)].input != P.nodes[kv.first].input;
      }
      for (auto &kv : m1[0]) {
        if (P.nodes[kv.first].op != RH_RIR_INPUT) continue;
        const uint32_t q = P.nodes[kv.first].input;
        if (!varies[q]) { fixed_param[q] = 1; continue; }
        size_t j = 0;
        while (j < pslots.size() && P.nodes[pslots[j]].input != q) j++;
        if (j == pslots.size()) { pslots.push_back(kv.first); pnodes.push_back({}); }
        pnodes[j].push_back(kv.first);
      }
      if (why) for (uint32_t x : pslots) std::fprintf(stderr, "lift:   template parameter %u varies\n", P.nodes[x].input);
      if (pslots.empty() || pslots.size() > 64) continue;
      std::vector<char> in_s0(np, 0);
      for (uint32_t x : pslots) in_s0[P.nodes[x].input] = 1;
      // pass 2: per member, the gradient outputs -- by slot for its own varying parameters, by position for the others
      members.clear(); maps.clear(); slots.clear();
      for (size_t k = 0; k < ok1.size(); k++) {
        const std::vector<uint32_t> &go = P.targets[ok1[k]].outputs;
        std::map<uint32_t, uint32_t> &memo = m1[k];
        std::vector<char> in_sg(np, 0);
        bool ok = true;
        for (size_t j = 0; j < pslots.size() && ok; j++) {
          const uint32_t q = P.nodes[MAP_AT(memo, pslots[j])].input;
          if (in_sg[q] || fixed_param[q]) { ok = false; break; }   // one parameter in two roles: its gradient is a sum, not a slot
          for (uint32_t x : pnodes[j]) if (P.nodes[MAP_AT(memo, x)].input != q) ok = false;
          in_sg[q] = 1;
        }
        for (size_t j = 0; j < pslots.size() && ok; j++)
          ok = match(P, to[1 + P.nodes[pslots[j]].input], go[1 + P.nodes[MAP_AT(memo, pslots[j])].input], memo, true);
        for (uint32_t q = 0; q < np && ok; q++) {
          if (in_sg[q]) { if (!in_s0[q]) ok = is_zero(to[1 + q]); continue; }
          if (in_s0[q]) ok = is_zero(go[1 + q]);
          else ok = match(P, to[1 + q], go[1 + q], memo, true);
        }
        // the renaming may only touch the slots
        for (auto &kv : memo) if (ok && P.nodes[kv.first].op == RH_RIR_INPUT && P.nodes[kv.second].input != P.nodes[kv.first].input) {
          size_t j = 0;
          while (j < pslots.size() && P.nodes[pslots[j]].input != P.nodes[kv.first].input) j++;
          if (j == pslots.size() || P.nodes[kv.second].input != P.nodes[MAP_AT(memo, pslots[j])].input) ok = false;
        }
        if (ok) { members.push_back(ok1[k]); maps.push_back(std::move(memo)); }
      }
      if (why) std::fprintf(stderr, "lift: %zu varying parameters, %zu members with matching gradients\n", pslots.size(), members.size());
      if (members.size() < 32 || members[0] != tmpl) continue;
      for (auto &kv : maps[0]) {
        if (P.nodes[kv.first].op != RH_RIR_CONST) continue;
        bool differs = false;
        const double v0 = P.nodes[kv.first].cval;
        for (size_t g = 1; g < members.size() && !differs; g++) {
          auto it = maps[g].find(kv.first);          // (a zero some member never reaches: an output position it does not use)
          if (it == maps[g].end()) continue;
          const double v = P.nodes[it->second].cval;
          differs = std::memcmp(&v, &v0, 8) != 0;
        }
        if (differs) slots.push_back(kv.first);
      }
      if (slots.size() > 4096) continue;
      ptable.assign(pslots.size(), {});
      for (size_t j = 0; j < pslots.size(); j++) {
        for (size_t g = 0; g < members.size(); g++) ptable[j].push_back(P.nodes[MAP_AT(maps[g], pslots[j])].input);
        std::sort(ptable[j].begin(), ptable[j].end());
        ptable[j].erase(std::unique(ptable[j].begin(), ptable[j].end()), ptable[j].end());
      }
      found = true;
      break;
    }
    if (!found) return false;
  }
  for (uint32_t s : slots) {
    std::vector<double> col;
    for (size_t g = 0; g < members.size(); g++) { auto it = maps[g].find(s); col.push_back(P.nodes[it != maps[g].end() ? it->second : s].cval); }
    synth.push_back(col);
  }
  for (size_t j = 0; j < pslots.size(); j++) {   // index columns: the position of the member's parameter in the slot's table
    std::vector<double> col;
    for (size_t g = 0; g < members.size(); g++)
      col.push_back((double)(std::lower_bound(ptable[j].begin(), ptable[j].end(), P.nodes[MAP_AT(maps[g], pslots[j])].input) - ptable[j].begin()));
    synth.push_back(col);
  }
  // the new row target: the template's expression with the slot constants replaced by column reads (and the varying parameters by
  // Lookups over their index columns)
  const uint32_t in0 = P.n_inputs;
  auto push = [&](const Node &q) { P.nodes.push_back(q); return (uint32_t)P.nodes.size() - 1; };
  auto input_node = [&](uint32_t input) { Node q; q.op = RH_RIR_INPUT; q.input = input; return push(q); };
  auto const_node = [&](double v) { Node q; q.op = RH_RIR_CONST; q.cval = v; return push(q); };
  std::map<uint32_t, uint32_t> copy;
  std::vector<char> touched(P.nodes.size(), 0);   // reaches a slot
  std::map<uint32_t, uint32_t> slot_input;        // (their INPUT nodes are created where the constant stood, in node order)
  for (size_t j = 0; j < slots.size(); j++) slot_input[slots[j]] = in0 + (uint32_t)j;
  std::vector<uint32_t> pcol(pslots.size());
  for (size_t j = 0; j < pslots.size(); j++) {
    pcol[j] = input_node(in0 + (uint32_t)(slots.size() + j));
    Node L; L.op = RH_RIR_LOOKUP; L.a = pcol[j]; L.low = 0;
    for (uint32_t q : ptable[j]) L.table.push_back(input_node(q));
    const uint32_t lk = push(L);
    for (auto &kv : maps[0])
      if (P.nodes[kv.first].op == RH_RIR_INPUT && kv.first < touched.size() && P.nodes[kv.first].input == P.nodes[pslots[j]].input) { copy[kv.first] = lk; touched[kv.first] = 1; }
  }
  std::vector<uint32_t> order;                    // template nodes, ascending
  for (auto &kv : maps[0]) order.push_back(kv.first);
  std::sort(order.begin(), order.end());
  touched.resize(P.nodes.size(), 0);
  for (uint32_t x : order) {
    const Node n = P.nodes[x];
    auto si = slot_input.find(x);
    if (si != slot_input.end()) { copy[x] = input_node(si->second); touched[x] = 1; continue; }
    if (copy.count(x) || n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    bool t = touched[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) t = t || touched[e]; } else if (binary_op(n.op)) t = t || touched[n.b];
    if (!t) continue;
    touched[x] = 1;
    Node q = n;
    auto cp = [&](uint32_t y) { auto it = copy.find(y); return it != copy.end() ? it->second : y; };
    q.a = cp(n.a);
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = cp(e); } else if (binary_op(n.op)) q.b = cp(n.b);
    copy[x] = push(q);
  }
  Target R;
  R.n_cols = (uint32_t)(slots.size() + pslots.size()); R.input_start = in0; R.col0 = P.n_cols_total;
  auto cpo = [&](uint32_t o) { auto it = copy.find(o); return it != copy.end() ? it->second : o; };
  const std::vector<uint32_t> tout = P.targets[tmpl].outputs;
  if (pslots.empty()) {
    for (uint32_t o : tout) R.outputs.push_back(cpo(o));
  } else {
    R.outputs.push_back(cpo(tout[0]));
    std::vector<char> in_s0(P.n_params, 0);
    for (uint32_t x : pslots) in_s0[P.nodes[x].input] = 1;
    const uint32_t zero = const_node(0.0);
    for (uint32_t q = 0; q < P.n_params; q++) {
      uint32_t acc = in_s0[q] ? zero : cpo(tout[1 + q]);         // a fixed parameter's gradient (or the zero everybody has there)
      for (size_t j = 0; j < pslots.size(); j++) {
        auto it = std::lower_bound(ptable[j].begin(), ptable[j].end(), q);
        if (it == ptable[j].end() || *it != q) continue;
        Node c; c.op = RH_RIR_COMPARE; c.a = pcol[j]; c.b = const_node((double)(it - ptable[j].begin()));
        Node e; e.op = RH_RIR_LOOKUP; e.a = push(c); e.low = -1;
        const uint32_t gslot = cpo(tout[1 + P.nodes[pslots[j]].input]);
        e.table = {zero, gslot, zero};                           // eq(index, k, g, 0): compute/Real.scala:83-99
        const uint32_t term = push(e);
        if (P.nodes[acc].op == RH_RIR_CONST && P.nodes[acc].cval == 0.0) acc = term;
        else { Node a; a.op = RH_RIR_ADD; a.a = acc; a.b = term; acc = push(a); }
      }
      R.outputs.push_back(acc);
    }
  }
  // the target list: members removed, the new row target appended
  std::vector<char> is_member(P.targets.size(), 0);
  for (uint32_t g : members) is_member[g] = 1;
  std::vector<Target> nt;
  old_target_of.clear();
  for (uint32_t t = 0; t < P.targets.size(); t++) if (!is_member[t]) { nt.push_back(P.targets[t]); old_target_of.push_back(t); }
  nt.push_back(R);
  old_target_of.push_back(0xFFFFFFFFu);           // the synthesised target: its row count is the number of members
  P.targets.swap(nt);
  P.n_inputs = in0 + R.n_cols; P.n_cols_total += R.n_cols;
  recompute_deps(P);
  return true;
}
}  // namespace

// ---- table priors -> a row target over the group index ---------------------------------------------------------------------
// (see rir.hpp) The data-free target's value is cut along its top-level additions (the Translator's left fold of a Line); a term
// belongs to table entry k when k is the only table parameter it reaches.  Every entry must have the same terms as the first one
// up to the renaming z_first -> z_k (constants that differ become columns), its gradient output likewise, and no other output of
// the target may reach a table parameter.  Per-row arithmetic is the entry's own; the order of the additions changes as in any
// row reduction.
namespace {
struct STerm { uint32_t node; bool neg; };
void flatten_sum(const Program &P, uint32_t id, std::vector<STerm> &out) {
  std::vector<std::pair<uint32_t, bool>> stack{{id, false}};
  while (!stack.empty()) {      // left operand first -> original left-to-right order
    auto [x, ng] = stack.back(); stack.pop_back();
    const Node &n = P.nodes[x];
    if (n.op == RH_RIR_ADD || n.op == RH_RIR_SUB) { stack.push_back({n.b, n.op == RH_RIR_SUB ? !ng : ng}); stack.push_back({n.a, ng}); }
    else out.push_back({x, ng});
  }
}
}  // namespace

bool lift_table_priors(Program &P, std::vector<std::vector<double>> &synth, int gather_min, bool rederive_ok) {
  auto no = [](int where) { if (rh::knob("RH_LIFT_WHY")) std::fprintf(stderr, "lift: table prior not lifted (check %d)\n", where); return false; };
  if (P.kind != 0) return false;
  const uint32_t np = P.n_params;
  uint32_t t0 = np;
  for (const Node &nd : P.nodes) {
    if (nd.op != RH_RIR_LOOKUP || (int)nd.table.size() < gather_min) continue;
    const Node &ix = P.nodes[nd.a], &e0 = P.nodes[nd.table[0]];
    if (ix.op == RH_RIR_INPUT && ix.input >= np && e0.op == RH_RIR_INPUT && e0.input < np && e0.input + nd.table.size() == np) t0 = std::min(t0, e0.input);
  }
  if (t0 >= np) return false;
  const uint32_t G = np - t0;
  auto is_zero = [&](uint32_t id) { return P.nodes[id].op == RH_RIR_CONST && P.nodes[id].cval == 0.0; };
  int tp = -1;                                    // the one data-free target with a gradient w.r.t. a table entry
  for (size_t t = 0; t < P.targets.size(); t++) {
    if (P.targets[t].n_cols) continue;
    bool dep = false;
    for (uint32_t k = t0; k < np && !dep; k++) dep = !is_zero(P.targets[t].outputs[1 + k]);
    if (dep) { if (tp >= 0) return no(1); tp = (int)t; }
  }
  if (tp < 0 || P.targets.size() + 1 > RH_MAX_TARGETS) return no(2);
  // table support of every node: -1 none, k one entry, -2 several
  std::vector<int64_t> sup(P.nodes.size(), -1);
  auto join = [](int64_t a, int64_t b) { return a == -1 ? b : (b == -1 || a == b) ? a : (int64_t)-2; };
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST) continue;
    if (n.op == RH_RIR_INPUT) { if (n.input >= t0 && n.input < np) sup[i] = n.input; continue; }
    int64_t s = sup[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) s = join(s, sup[e]); } else if (binary_op(n.op)) s = join(s, sup[n.b]);
    sup[i] = s;
  }
  const std::vector<uint32_t> outs = P.targets[(size_t)tp].outputs;
  std::vector<STerm> vterms, rest;
  flatten_sum(P, outs[0], vterms);
  std::vector<std::vector<STerm>> val(G), grad(G);
  for (const STerm &tm : vterms) {
    if (sup[tm.node] == -2) return no(3);
    if (sup[tm.node] == -1) rest.push_back(tm); else val[(size_t)(sup[tm.node] - t0)].push_back(tm);
  }
  // `simple`: the supplied gradient is per-entry too (a standard prior on the entries); otherwise the prior ties shared parameters
  // to the table (centred parameterisation) and the gradients are derived again below (fast builds only)
  bool simple = true;
  for (uint32_t q = 0; q < t0; q++) if (sup[outs[1 + q]] != -1) simple = false;
  for (uint32_t k = t0; k < np; k++) {
    flatten_sum(P, outs[1 + k], grad[k - t0]);
    for (const STerm &tm : grad[k - t0]) if (sup[tm.node] != -1 && sup[tm.node] != (int64_t)k) simple = false;
    if (val[k - t0].empty()) return no(6);
  }
  if (!simple && !rederive_ok) return no(4);
  // every entry against the first: same terms in the same order, the only renaming z_first -> z_k
  std::vector<std::map<uint32_t, uint32_t>> maps(G);
  for (uint32_t g = 0; g < G; g++) {
    if (val[g].size() != val[0].size() || (simple && grad[g].size() != grad[0].size())) return no(7);
    std::map<uint32_t, uint32_t> &memo = maps[g];
    for (size_t i = 0; i < val[0].size(); i++) if (val[g][i].neg != val[0][i].neg || !match_entry(P, val[0][i].node, val[g][i].node, memo, t0, t0 + g, t0)) return no(8);
    if (simple) for (size_t i = 0; i < grad[0].size(); i++) if (grad[g][i].neg != grad[0][i].neg || !match_entry(P, grad[0][i].node, grad[g][i].node, memo, t0, t0 + g, t0)) return no(9);
  }
  std::vector<uint32_t> slots;                    // constants that differ between entries -> columns
  for (auto &kv : maps[0]) {
    if (P.nodes[kv.first].op != RH_RIR_CONST) continue;
    const double v0 = P.nodes[kv.first].cval;
    bool differs = false;
    for (uint32_t g = 1; g < G && !differs; g++) { const double v = P.nodes[MAP_AT(maps[g], kv.first)].cval; differs = std::memcmp(&v, &v0, 8) != 0; }
    if (differs) slots.push_back(kv.first);
  }
  if (slots.size() > 4096) return no(11);
  for (uint32_t s : slots) { std::vector<double> col; for (uint32_t g = 0; g < G; g++) col.push_back(P.nodes[MAP_AT(maps[g], s)].cval); synth.push_back(col); }
  { std::vector<double> col; for (uint32_t g = 0; g < G; g++) col.push_back((double)g); synth.push_back(col); }
  // the row target
  const uint32_t in0 = P.n_inputs;
  auto push = [&](const Node &q) { P.nodes.push_back(q); return (uint32_t)P.nodes.size() - 1; };
  auto input_node = [&](uint32_t input) { Node q; q.op = RH_RIR_INPUT; q.input = input; return push(q); };
  auto const_node = [&](double v) { Node q; q.op = RH_RIR_CONST; q.cval = v; return push(q); };
  auto op2 = [&](uint32_t op, uint32_t a, uint32_t b) { Node q; q.op = op; q.a = a; q.b = b; return push(q); };
  const size_t n_old = P.nodes.size();
  std::map<uint32_t, uint32_t> copy;
  std::vector<char> touched(n_old, 0);
  for (size_t j = 0; j < slots.size(); j++) { copy[slots[j]] = input_node(in0 + (uint32_t)j); touched[slots[j]] = 1; }
  const uint32_t idx = input_node(in0 + (uint32_t)slots.size());
  {
    Node L; L.op = RH_RIR_LOOKUP; L.a = idx; L.low = 0;
    for (uint32_t k = t0; k < np; k++) L.table.push_back(input_node(k));
    const uint32_t lk = push(L);
    for (auto &kv : maps[0]) if (P.nodes[kv.first].op == RH_RIR_INPUT && P.nodes[kv.first].input == t0) { copy[kv.first] = lk; touched[kv.first] = 1; }
  }
  std::vector<uint32_t> order;
  for (auto &kv : maps[0]) order.push_back(kv.first);
  std::sort(order.begin(), order.end());
  auto cp = [&](uint32_t y) { auto it = copy.find(y); return it != copy.end() ? it->second : y; };
  for (uint32_t x : order) {
    const Node n = P.nodes[x];
    if (copy.count(x) || n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    bool t = touched[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) t = t || touched[e]; } else if (binary_op(n.op)) t = t || touched[n.b];
    if (!t) continue;
    touched[x] = 1;
    Node q = n;
    q.a = cp(n.a);
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = cp(e); } else if (binary_op(n.op)) q.b = cp(n.b);
    copy[x] = push(q);
  }
  const uint32_t zero = const_node(0.0);
  auto chain = [&](const std::vector<STerm> &ts, bool copied) {
    uint32_t acc = 0xFFFFFFFFu;
    for (const STerm &tm : ts) {
      const uint32_t x = copied ? cp(tm.node) : tm.node;
      if (acc == 0xFFFFFFFFu) acc = tm.neg ? op2(RH_RIR_SUB, zero, x) : x;
      else acc = op2(tm.neg ? RH_RIR_SUB : RH_RIR_ADD, acc, x);
    }
    return acc == 0xFFFFFFFFu ? zero : acc;
  };
  Target R;
  R.n_cols = (uint32_t)slots.size() + 1; R.input_start = in0; R.col0 = P.n_cols_total;
  R.outputs.push_back(chain(val[0], true));
  if (simple) {
    const uint32_t gnode = chain(grad[0], true);
    for (uint32_t q = 0; q < t0; q++) R.outputs.push_back(zero);
    for (uint32_t g = 0; g < G; g++) {
      Node e; e.op = RH_RIR_LOOKUP; e.a = op2(RH_RIR_COMPARE, idx, const_node((double)g)); e.low = -1;
      e.table = {zero, gnode, zero};                               // eq(index, g, f'(z), 0): compute/Real.scala:83-99
      R.outputs.push_back(push(e));
    }
    // the data-free target keeps the rest
    Target &T = P.targets[(size_t)tp];
    T.outputs[0] = chain(rest, false);
    for (uint32_t k = t0; k < np; k++) T.outputs[1 + k] = zero;
    P.targets.push_back(R);
    P.n_inputs = in0 + R.n_cols; P.n_cols_total += R.n_cols;
    recompute_deps(P);
    return true;
  }
  // ---- centred: both gradients derived again from the two values, then checked against the outputs the program came with
  const size_t synth0 = synth.size() - (slots.size() + 1);
  const uint32_t rest_value = chain(rest, false);
  Program Q = P;                                           // built on a copy: the check below decides
  Q.n_inputs = in0 + R.n_cols; Q.n_cols_total += R.n_cols;  // the new inputs exist before nodes over them are differentiated
  {
    const std::vector<uint32_t> gr = derive_gradient(Q, R.outputs[0]);
    for (uint32_t q = 0; q < np; q++) R.outputs.push_back(gr[q]);
    const std::vector<uint32_t> gd = derive_gradient(Q, rest_value);
    Target &T = Q.targets[(size_t)tp];
    T.outputs[0] = rest_value;
    for (uint32_t q = 0; q < np; q++) T.outputs[1 + q] = gd[q];
  }
  Q.targets.push_back(R);
  recompute_deps(Q);
  {
    // original outputs against  data-free part + sum over the G rows of the row target, at 6 random points; per output:
    // |difference| <= 1e-11 * (sum of magnitudes), evaluated in extended precision; one-sided non-finite values reject
    bool ok = true;
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0 - 0.5; };
    static const double kScale[6] = {1.0, 1.0, 0.2, 2.5, 1.0, 0.2};
    constexpr int B = 32;
    const Target &T = Q.targets[(size_t)tp], &RT = Q.targets.back();
    // The row target's table-entry outputs are eq(index, k, g, 0) (the Lookup rule of the derivation): row g contributes to entry g
    // alone, so those outputs are checked row by row -- own entry only -- on up to 8 blocks of 32 entries spread over the table (every
    // entry is the same expression over the looked-up value); the value and the shared parameters' outputs are summed over ALL rows.
    const std::vector<uint32_t> roots_old(outs.begin(), outs.end()), roots_df(T.outputs.begin(), T.outputs.end());
    const std::vector<uint32_t> roots_sh(RT.outputs.begin(), RT.outputs.begin() + 1 + t0);
    std::vector<long double> val_o, val_d, val_r;
    std::vector<char> rok;
    const BlockEvaluator ev_old(P, roots_old), ev_df(Q, roots_df), ev_sh(Q, roots_sh);
    for (int trial = 0; trial < 6 && ok; trial++) {
      std::vector<double> th(np);
      for (uint32_t q = 0; q < np; q++) th[q] = kScale[trial] * uni();
      std::vector<long double> in1((size_t)P.n_inputs, 0.0L);
      for (uint32_t q = 0; q < np; q++) in1[q] = th[q];
      if (!ev_old.run(in1, 1, val_o, rok) || !rok[0]) { ok = false; break; }
      std::vector<long double> in2((size_t)Q.n_inputs, 0.0L);
      for (uint32_t q = 0; q < np; q++) in2[q] = th[q];
      if (!ev_df.run(in2, 1, val_d, rok) || !rok[0]) { ok = false; break; }
      std::vector<long double> sum(t0 + 1), mag(t0 + 1);
      for (uint32_t o = 0; o <= t0; o++) { sum[o] = val_d[T.outputs[o]]; mag[o] = std::fabs(sum[o]); }
      std::vector<long double> inb((size_t)Q.n_inputs * B, 0.0L);
      for (uint32_t q = 0; q < np; q++) for (int r = 0; r < B; r++) inb[(size_t)q * B + r] = th[q];   // the parameters: once per point
      auto fill = [&](uint32_t g0) {                                                                   // the block's rows
        for (int r = 0; r < B; r++) {
          const uint32_t g = std::min(g0 + (uint32_t)r, G - 1);
          for (uint32_t j = 0; j < RT.n_cols; j++) inb[(size_t)(RT.input_start + j) * B + r] = synth[synth0 + j][g];
        }
      };
      auto differs = [&](long double a, long double b, long double m) {
        if (!std::isfinite(a) || !std::isfinite(b)) return std::isfinite(a) != std::isfinite(b) || std::isnan(a) != std::isnan(b) || (std::isinf(a) && a != b);
        return std::fabs(a - b) > 1e-11L * std::max(m, std::fabs(a)) + 1e-300L;
      };
      for (uint32_t g0 = 0; g0 < G && ok; g0 += B) {
        const int nb = (int)std::min<uint32_t>(B, G - g0);
        fill(g0);
        if (!ev_sh.run(inb, B, val_r, rok)) { ok = false; break; }
        for (int r = 0; r < nb && ok; r++) {
          if (!rok[(size_t)r]) { ok = false; break; }
          for (uint32_t o = 0; o <= t0; o++) { const long double v = val_r[(size_t)RT.outputs[o] * B + r]; sum[o] += v; mag[o] += std::fabs(v); }
        }
      }
      for (uint32_t o = 0; o <= t0 && ok; o++)
        if (differs(val_o[outs[o]], sum[o], mag[o])) {
          ok = false;
          if (rh::knob("RH_LIFT_WHY")) std::fprintf(stderr, "lift: centred table prior: output %u differs (%.6Lg vs %.6Lg; data-free part %.6Lg)\n", o, val_o[outs[o]], sum[o], val_d[T.outputs[o]]);
        }
      const uint32_t nblk = (G + B - 1) / B, stride = std::max<uint32_t>(1, nblk / 8);
      for (uint32_t blk = 0; blk < nblk && ok; blk += stride) {
        const uint32_t g0 = blk * B;
        const int nb = (int)std::min<uint32_t>(B, G - g0);
        std::vector<uint32_t> roots_e;
        for (int r = 0; r < nb; r++) roots_e.push_back(RT.outputs[1 + t0 + g0 + (uint32_t)r]);
        const BlockEvaluator ev_e(Q, roots_e);
        fill(g0);
        if (!ev_e.run(inb, B, val_r, rok)) { ok = false; break; }
        for (int r = 0; r < nb && ok; r++) {
          const uint32_t o = 1 + t0 + g0 + (uint32_t)r;
          const long double a = val_o[outs[o]], b = val_d[T.outputs[o]] + val_r[(size_t)RT.outputs[o] * B + r];
          if (!rok[(size_t)r] || differs(a, b, std::fabs(b))) {
            ok = false;
            if (rh::knob("RH_LIFT_WHY")) std::fprintf(stderr, "lift: centred table prior: entry output %u differs (%.6Lg vs %.6Lg)\n", o, a, b);
          }
        }
      }
    }
    // (the variable is a debugging aid.)  Nothing of the attempt is left behind: the row target's nodes were appended to P itself,
    // among them INPUT nodes beyond P.n_inputs that a later lift would hand the same indices to
    if (!ok && !rh::knob("RH_LIFT_NOVERIFY")) { synth.resize(synth0); P.nodes.resize(n_old); return no(12); }
  }
  P = std::move(Q);
  return true;
}

// ---- the affine map of a parameter table moved behind the lookup (see rir.hpp) ----------------------------------------------
bool hoist_table_maps(Program &P, int gather_min) {
  auto no = [](int where) { if (rh::knob("RH_LIFT_WHY")) std::fprintf(stderr, "lift: table map not hoisted (check %d)\n", where); return false; };
  if (P.kind != 0) return false;
  const uint32_t np = P.n_params;
  std::vector<char> reaches_param(P.nodes.size(), 0);
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST) continue;
    if (n.op == RH_RIR_INPUT) { reaches_param[i] = n.input < np; continue; }
    char r = reaches_param[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) r = r || reaches_param[e]; } else if (binary_op(n.op)) r = r || reaches_param[n.b];
    reaches_param[i] = r;
  }
  std::vector<uint32_t> cands;
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &nd = P.nodes[i];
    if (nd.op != RH_RIR_LOOKUP) continue;
    const Node &ix = P.nodes[nd.a];
    if (!(ix.op == RH_RIR_INPUT && ix.input >= np)) continue;        // (selects on a compare, parameter-indexed tables: not ours)
    bool params_only = true, any_param = false;                       // a PARAMETER table: entries free of data, not all constant
    for (uint32_t e : nd.table) { params_only = params_only && P.nodes[e].dep == 0; any_param = any_param || reaches_param[e]; }
    if (!params_only || !any_param) continue;                         // (a select between row-level terms: Real.eq over a column)
    if ((int)nd.table.size() < gather_min) return no(1);              // a second, small parameter table: gather mode has one
    cands.push_back(i);
  }
  if (cands.empty()) return no(2);
  const Node L0 = P.nodes[cands[0]];
  for (uint32_t c : cands) {                       // the same table everywhere: the very nodes, or copies of them (8 slots)
    const Node &Lc = P.nodes[c];
    if (Lc.low != L0.low || Lc.table.size() != L0.table.size()) return no(3);
    if (Lc.table == L0.table) continue;
    std::map<uint32_t, uint32_t> same;
    for (size_t k = 0; k < L0.table.size(); k++) if (!match(P, L0.table[k], Lc.table[k], same, false)) return no(3);
    for (auto &kv : same) if (P.nodes[kv.first].op == RH_RIR_CONST && std::memcmp(&P.nodes[kv.first].cval, &P.nodes[kv.second].cval, 8) != 0) return no(3);
  }
  const uint32_t K = (uint32_t)L0.table.size();
  if (K > np || P.nodes[L0.table[0]].op == RH_RIR_INPUT) return no(4);                // already raw
  for (uint32_t e : L0.table) if (P.nodes[e].dep != 0) return no(5);
  const uint32_t t0 = np - K;
  // every entry against the first; the one parameter that varies must be t0 + k
  std::map<uint32_t, uint32_t> memo0;
  if (!match(P, L0.table[0], L0.table[0], memo0, true)) return no(6);
  std::vector<uint32_t> zin(K, 0xFFFFFFFFu);       // the INPUT node of z_k
  for (uint32_t k = 0; k < K; k++) {
    std::map<uint32_t, uint32_t> memo;
    if (!match(P, L0.table[0], L0.table[k], memo, true)) return no(7);
    for (auto &kv : memo) {
      const Node &a = P.nodes[kv.first], &b = P.nodes[kv.second];
      if (a.op == RH_RIR_CONST) { if (std::memcmp(&a.cval, &b.cval, 8) != 0) return no(8); continue; }
      if (a.op != RH_RIR_INPUT) continue;
      if (a.input == t0) { if (b.input != t0 + k) return no(9); zin[k] = kv.second; }
      else if (a.input != b.input || a.input >= t0) return no(10);
    }
    if (zin[k] == 0xFFFFFFFFu) return no(11);
  }
  std::vector<uint32_t> tmpl;                      // the first entry's nodes, ascending
  for (auto &kv : memo0) tmpl.push_back(kv.first);
  std::sort(tmpl.begin(), tmpl.end());
  std::vector<char> is_cand(P.nodes.size(), 0);
  for (uint32_t c : cands) is_cand[c] = 1;
  std::vector<char> is_entry(P.nodes.size(), 0);   // entries of the tables that are hoisted
  for (uint32_t c : cands) for (uint32_t e : P.nodes[c].table) is_entry[e] = 1;
  for (uint32_t i = 0; i < P.nodes.size(); i++)    // ... and the balanced tree of SeqIRs over them (ir/IR.scala:25-39)
    if (P.nodes[i].op == RH_RIR_SEQ && is_entry[P.nodes[i].a] && is_entry[P.nodes[i].b]) is_entry[i] = 1;
  // rebuild the node list (operands keep smaller ids than their users)
  std::vector<Node> Q;
  std::vector<uint32_t> m(P.nodes.size(), 0);
  auto push = [&](const Node &q) { Q.push_back(q); return (uint32_t)Q.size() - 1; };
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    // the Translator's VarDef chain that evaluates every entry before the lookup (SeqIR(defs :+ LookupIR)): value = second
    // operand; the hoisted table's entries are no longer evaluated one by one.  Any other SEQ of the program stays.
    if (n.op == RH_RIR_SEQ && is_entry[n.a]) { m[i] = m[n.b]; continue; }
    if (!is_cand[i]) {
      Node q = n;
      if (n.op != RH_RIR_CONST && n.op != RH_RIR_INPUT) {
        q.a = m[n.a];
        if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = m[e]; } else if (binary_op(n.op)) q.b = m[n.b];
      }
      m[i] = push(q);
      continue;
    }
    Node L; L.op = RH_RIR_LOOKUP; L.a = m[n.a]; L.low = n.low;
    for (uint32_t k = 0; k < K; k++) L.table.push_back(m[zin[k]]);
    const uint32_t lk = push(L);
    std::map<uint32_t, uint32_t> copy;             // template node -> its copy over the looked-up entry
    for (uint32_t x : tmpl) {
      const Node &t = P.nodes[x];
      if (t.op == RH_RIR_INPUT) { if (t.input == t0) copy[x] = lk; continue; }
      if (t.op == RH_RIR_CONST) continue;
      auto cp = [&](uint32_t y) { auto it = copy.find(y); return it != copy.end() ? it->second : m[y]; };
      bool touched = copy.count(t.a) != 0;
      if (t.op == RH_RIR_LOOKUP) { for (uint32_t e : t.table) touched = touched || copy.count(e); } else if (binary_op(t.op)) touched = touched || copy.count(t.b);
      if (!touched) continue;
      Node q = t;
      q.a = cp(t.a);
      if (t.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = cp(e); } else if (binary_op(t.op)) q.b = cp(t.b);
      copy[x] = push(q);
    }
    auto it = copy.find(L0.table[0]);
    m[i] = it != copy.end() ? it->second : m[L0.table[0]];
  }
  P.nodes.swap(Q);
  for (Target &T : P.targets) for (uint32_t &o : T.outputs) o = m[o];
  recompute_deps(P);
  return true;
}

// ---- a data-free expression over one table entry -> a one-row target (see rir.hpp) ----------------------------------------------
int lift_single_entry_targets(Program &P, std::vector<std::vector<double>> &synth, int gather_min) {
  if (P.kind != 0) return 0;
  const uint32_t np = P.n_params;
  uint32_t t0 = np; int32_t low = 0;
  for (const Node &nd : P.nodes) {
    if (nd.op != RH_RIR_LOOKUP || (int)nd.table.size() < gather_min) continue;
    const Node &ix = P.nodes[nd.a], &e0 = P.nodes[nd.table[0]];
    if (ix.op == RH_RIR_INPUT && ix.input >= np && e0.op == RH_RIR_INPUT && e0.input < np && e0.input + nd.table.size() == np && e0.input < t0) { t0 = e0.input; low = nd.low; }
  }
  if (t0 >= np) return 0;
  const uint32_t G = np - t0;
  std::vector<int64_t> sup(P.nodes.size(), -1);   // table support: -1 none, k one entry, -2 several
  auto join = [](int64_t a, int64_t b) { return a == -1 ? b : (b == -1 || a == b) ? a : (int64_t)-2; };
  for (uint32_t i = 0; i < P.nodes.size(); i++) {
    const Node &n = P.nodes[i];
    if (n.op == RH_RIR_CONST) continue;
    if (n.op == RH_RIR_INPUT) { if (n.input >= t0 && n.input < np) sup[i] = n.input; continue; }
    int64_t s = sup[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) s = join(s, sup[e]); } else if (binary_op(n.op)) s = join(s, sup[n.b]);
    sup[i] = s;
  }
  auto is_zero = [&](uint32_t id) { return P.nodes[id].op == RH_RIR_CONST && P.nodes[id].cval == 0.0; };
  int lifted = 0;
  const size_t NT = P.targets.size();
  for (size_t t = 0; t < NT && lifted < 4 && P.targets.size() < RH_MAX_TARGETS; t++) {
    const std::vector<uint32_t> outs = P.targets[t].outputs;
    int64_t k = -1;
    bool ok = true;
    for (uint32_t o : outs) { ok = ok && P.nodes[o].dep == 0; k = join(k, sup[o]); }
    if (!ok || k < 0) continue;
    for (uint32_t q = t0; q < np && ok; q++) if ((int64_t)q != k) ok = is_zero(outs[1 + q]);
    if (!ok) continue;
    // the one-row target: the entry read through a Lookup over the new column
    const uint32_t in0 = P.n_inputs;
    auto push = [&](const Node &q) { P.nodes.push_back(q); return (uint32_t)P.nodes.size() - 1; };
    auto input_node = [&](uint32_t input) { Node q; q.op = RH_RIR_INPUT; q.input = input; return push(q); };
    auto const_node = [&](double v) { Node q; q.op = RH_RIR_CONST; q.cval = v; return push(q); };
    const size_t n_old = P.nodes.size();
    const uint32_t idx = input_node(in0);
    Node L; L.op = RH_RIR_LOOKUP; L.a = idx; L.low = low;
    for (uint32_t q = t0; q < np; q++) L.table.push_back(input_node(q));
    const uint32_t lk = push(L);
    std::vector<uint32_t> cpy(n_old, 0xFFFFFFFFu);
    std::vector<char> reach(n_old, 0);
    for (uint32_t o : outs) reach[o] = 1;
    for (size_t i = n_old; i-- > 0;) {
      if (!reach[i]) continue;
      const Node &n = P.nodes[i];
      if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
      reach[n.a] = 1;
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) reach[e] = 1; } else if (binary_op(n.op)) reach[n.b] = 1;
    }
    for (uint32_t i = 0; i < n_old; i++) {
      if (!reach[i] || sup[i] != k) continue;
      const Node n = P.nodes[i];
      if (n.op == RH_RIR_INPUT) { cpy[i] = lk; continue; }
      auto cp = [&](uint32_t y) { return cpy[y] != 0xFFFFFFFFu ? cpy[y] : y; };
      Node q = n;
      q.a = cp(n.a);
      if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = cp(e); } else if (binary_op(n.op)) q.b = cp(n.b);
      cpy[i] = push(q);
    }
    auto cpo = [&](uint32_t o) { return cpy[o] != 0xFFFFFFFFu ? cpy[o] : o; };
    const uint32_t zero = const_node(0.0);
    Target R;
    R.n_cols = 1; R.input_start = in0; R.col0 = P.n_cols_total;
    R.outputs.push_back(cpo(outs[0]));
    for (uint32_t q = 0; q < t0; q++) R.outputs.push_back(cpo(outs[1 + q]));
    const uint32_t gnode = cpo(outs[1 + (uint32_t)k]);
    for (uint32_t g = 0; g < G; g++) {
      Node c; c.op = RH_RIR_COMPARE; c.a = idx; c.b = const_node((double)((int64_t)g + low));
      Node e; e.op = RH_RIR_LOOKUP; e.a = push(c); e.low = -1;
      e.table = {zero, gnode, zero};                               // eq(index, low + g, d/dz, 0)
      R.outputs.push_back(push(e));
    }
    for (uint32_t &o : P.targets[t].outputs) o = zero;
    P.targets.push_back(R);
    P.n_inputs = in0 + 1; P.n_cols_total += 1;
    synth.push_back({(double)((int64_t)((uint32_t)k - t0) + low)});
    lifted++;
    sup.resize(P.nodes.size(), -1);
  }
  if (lifted) recompute_deps(P);
  return lifted;
}

}  // namespace rh
