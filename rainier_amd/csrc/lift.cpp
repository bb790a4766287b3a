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
// while another group of >= 32 qualifies (two time series observed one value at a time become two streamed targets).
#include <algorithm>
#include <cstring>
#include <map>

#include "../../include/rainier_hip_rir.h"
#include "device/rh_shared.h"
#include "rir.hpp"

namespace rh {
namespace {

bool binary_op(uint32_t op) { return (op >= RH_RIR_ADD && op <= RH_RIR_COMPARE) || op == RH_RIR_SEQ; }
uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xBF58476D1CE4E5B9ull; }

// member node b against template node a; memo: template node -> member node (must be consistent)
bool match(const Program &P, uint32_t a, uint32_t b, std::map<uint32_t, uint32_t> &memo) {
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
      case RH_RIR_INPUT: if (nx.input != ny.input) return false; break;
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

bool lift_one(Program &P, std::vector<std::vector<double>> &synth, std::vector<uint32_t> &old_target_of);

}  // namespace

void recompute_deps(Program &P) {
  for (Node &n : P.nodes) {
    uint8_t dep = 0;
    if (n.op == RH_RIR_INPUT) {
      if (n.input >= P.n_params)
        for (size_t t = 0; t < P.targets.size(); t++)
          if (n.input >= P.targets[t].input_start && n.input < P.targets[t].input_start + P.targets[t].n_cols) dep = (uint8_t)(t + 1);
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
        const double v = P.nodes[maps[g].at(kv.first)].cval;
        differs = std::memcmp(&v, &v0, 8) != 0;
      }
      if (differs) slots.push_back(kv.first);
    }
    if (slots.empty() || slots.size() > 4096) continue;
    found = true;
    break;
  }
  if (!found) return false;
  for (uint32_t s : slots) {
    std::vector<double> col;
    for (size_t g = 0; g < members.size(); g++) col.push_back(P.nodes[maps[g].at(s)].cval);
    synth.push_back(col);
  }
  // the new row target: the template's expression with the slot constants replaced by column reads
  const uint32_t in0 = P.n_inputs;
  std::map<uint32_t, uint32_t> slot_input;
  for (size_t j = 0; j < slots.size(); j++) slot_input[slots[j]] = in0 + (uint32_t)j;
  std::vector<char> touched(P.nodes.size(), 0);   // reaches a slot
  std::vector<uint32_t> order;                    // template nodes, ascending
  for (auto &kv : maps[0]) order.push_back(kv.first);
  std::sort(order.begin(), order.end());
  std::map<uint32_t, uint32_t> copy;
  for (uint32_t x : order) {
    const Node n = P.nodes[x];
    auto si = slot_input.find(x);
    if (si != slot_input.end()) { Node q; q.op = RH_RIR_INPUT; q.input = si->second; P.nodes.push_back(q); copy[x] = (uint32_t)P.nodes.size() - 1; touched[x] = 1; continue; }
    if (n.op == RH_RIR_CONST || n.op == RH_RIR_INPUT) continue;
    bool t = touched[n.a];
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t e : n.table) t = t || touched[e]; } else if (binary_op(n.op)) t = t || touched[n.b];
    if (!t) continue;
    touched[x] = 1;
    Node q = n;
    auto cp = [&](uint32_t y) { auto it = copy.find(y); return it != copy.end() ? it->second : y; };
    q.a = cp(n.a);
    if (n.op == RH_RIR_LOOKUP) { for (uint32_t &e : q.table) e = cp(e); } else if (binary_op(n.op)) q.b = cp(n.b);
    P.nodes.push_back(q);
    copy[x] = (uint32_t)P.nodes.size() - 1;
  }
  Target R;
  R.n_cols = (uint32_t)slots.size(); R.input_start = in0; R.col0 = P.n_cols_total;
  for (uint32_t o : P.targets[tmpl].outputs) { auto it = copy.find(o); R.outputs.push_back(it != copy.end() ? it->second : o); }
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

}  // namespace rh
