// rir.cpp -- RIR reader/validator (format: include/rainier_hip_rir.h).
#include "rir.hpp"

#include <cstdlib>

#include <cstring>

#include "../../include/rainier_hip_rir.h"
#include "device/rh_shared.h"

namespace rh {
const char *knob(const char *name) {
  static const bool diag = [] { const char *e = std::getenv("RH_DIAG"); return e && std::atoi(e) != 0; }();
  return diag ? std::getenv(name) : nullptr;
}

namespace {
struct Reader {
  const uint8_t *p;
  size_t n, pos = 0;
  bool bad = false;
  uint32_t u32() {
    if (pos + 4 > n) { bad = true; return 0; }
    uint32_t v; std::memcpy(&v, p + pos, 4); pos += 4; return v;
  }
  double f64() {
    if (pos + 8 > n) { bad = true; return 0; }
    double v; std::memcpy(&v, p + pos, 8); pos += 8; return v;
  }
};
}  // namespace

bool parse_rir(const void *buf, size_t len, Program &P, std::string &err) {
  if (!buf || len < 24 || (len & 3)) { err = "RIR: blob too short or not 4-byte sized"; return false; }
  Reader r{static_cast<const uint8_t *>(buf), len};
  if (r.u32() != RH_RIR_MAGIC) { err = "RIR: bad magic"; return false; }
  if (r.u32() != RH_RIR_VERSION) { err = "RIR: unsupported version"; return false; }
  P.n_params = r.u32();
  const uint32_t n_targets = r.u32(), n_nodes = r.u32();
  P.kind = r.u32();
  if (P.kind > 1) { err = "RIR: unknown program kind"; return false; }
  if (P.n_params == 0 || n_targets == 0 || n_targets > (P.kind == 1 ? 4096u : RH_RIR_MAX_TARGETS)) { err = "RIR: bad header (params/targets)"; return false; }
  if ((size_t)n_nodes * 8 > len) { err = "RIR: node count exceeds blob"; return false; }
  // every target carries n_params + 1 output ids: a header whose target table cannot fit the blob is rejected before
  // anything is sized from it (n_params = 0xFFFFFFFF would wrap n_params + 1 to 0, 2^31 would allocate gigabytes)
  if (P.n_params > RH_RIR_MAX_PARAMS) { err = "RIR: n_params exceeds RH_RIR_MAX_PARAMS"; return false; }
  if ((uint64_t)n_targets * ((uint64_t)P.n_params + 3) * 4 > (uint64_t)len) { err = "RIR: target table exceeds blob"; return false; }
  P.targets.resize(n_targets);
  uint32_t in = P.n_params, col = 0;
  for (auto &t : P.targets) {
    t.n_cols = r.u32(); (void)r.u32();
    t.input_start = in; t.col0 = col;
    in += t.n_cols; col += t.n_cols;
    t.outputs.resize(P.n_params + 1);
    for (auto &o : t.outputs) {
      o = r.u32();
      if (r.bad) { err = "RIR: truncated target table"; return false; }
      if (o >= n_nodes) { err = "RIR: output node id out of range"; return false; }
    }
    if (r.bad) { err = "RIR: truncated target table"; return false; }
  }
  if (col > RH_RIR_MAX_COLS) { err = "RIR: too many data columns"; return false; }
  if (P.kind == 1 && col != 0) { err = "RIR: a requirements program cannot have data columns"; return false; }
  P.n_inputs = in; P.n_cols_total = col;
  P.nodes.resize(n_nodes);
  for (uint32_t i = 0; i < n_nodes; i++) {
    Node &nd = P.nodes[i];
    nd.op = r.u32();
    uint32_t dep = 0;
    auto use = [&](uint32_t x) -> bool {
      if (x >= i) { err = "RIR: node " + std::to_string(i) + " references a later node"; return false; }
      const uint32_t d = P.nodes[x].dep;
      if (d) { if (dep && dep != d) { err = "RIR: node " + std::to_string(i) + " mixes columns of two targets"; return false; } dep = d; }
      return true;
    };
    switch (nd.op) {
      case RH_RIR_CONST:
        nd.cval = r.f64();
        if (nd.cval != nd.cval) { err = "RIR: NaN constant"; return false; }
        break;
      case RH_RIR_INPUT:
        nd.input = r.u32();
        if (nd.input >= P.n_inputs) { err = "RIR: input index out of range"; return false; }
        if (nd.input >= P.n_params)
          for (uint32_t t = 0; t < n_targets; t++)
            if (nd.input >= P.targets[t].input_start && nd.input < P.targets[t].input_start + P.targets[t].n_cols) dep = (uint32_t)(t + 1);
        break;
      case RH_RIR_ADD: case RH_RIR_SUB: case RH_RIR_MUL: case RH_RIR_DIV: case RH_RIR_POW: case RH_RIR_COMPARE:
      case RH_RIR_SEQ:
        nd.a = r.u32(); nd.b = r.u32();
        if (!use(nd.a) || !use(nd.b)) return false;
        break;
      case RH_RIR_EXP: case RH_RIR_LOG: case RH_RIR_ABS: case RH_RIR_NOOP: case RH_RIR_SIN: case RH_RIR_COS:
      case RH_RIR_TAN: case RH_RIR_ASIN: case RH_RIR_ACOS: case RH_RIR_ATAN:
        nd.a = r.u32();
        if (!use(nd.a)) return false;
        break;
      case RH_RIR_LOOKUP: {
        nd.a = r.u32(); nd.low = (int32_t)r.u32();
        const uint32_t cnt = r.u32();
        if (cnt == 0 || (size_t)cnt * 4 > len) { err = "RIR: bad lookup table size"; return false; }
        if (!use(nd.a)) return false;
        nd.table.resize(cnt);
        for (auto &e : nd.table) { e = r.u32(); if (!use(e)) return false; }
        break;
      }
      default: err = "RIR: unknown opcode " + std::to_string(nd.op); return false;
    }
    if (r.bad) { err = "RIR: truncated node table"; return false; }
    nd.dep = dep;
  }
  if (r.pos != r.n) { err = "RIR: trailing bytes"; return false; }
  // a target's outputs may only reach its own columns
  for (uint32_t t = 0; t < n_targets; t++)
    for (uint32_t o : P.targets[t].outputs) {
      const uint32_t d = P.nodes[o].dep;
      if (d && d != t + 1) { err = "RIR: target " + std::to_string(t) + " reads a column of another target"; return false; }
    }
  return true;
}
// More targets than the engine holds (RH_MAX_TARGETS): every run of consecutive data-free targets becomes one target whose
// outputs are the left-fold sums of the run's outputs in target order -- the order in which the kernels add the targets up, so
// the model's values do not change by a bit.  (A model that observes one value at a time -- the reference's ARK benchmark:
// `Model.observe(ys(t), ...)` merged 195 times -- arrives with one data-free target per observation.)
bool merge_data_free_targets(Program &P, std::vector<uint32_t> &old_target_of) {
  old_target_of.clear();
  for (uint32_t t = 0; t < P.targets.size(); t++) old_target_of.push_back(t);
  if (P.kind != 0 || P.targets.size() <= RH_MAX_TARGETS) return false;
  std::vector<Target> nt;
  std::vector<uint32_t> first;
  for (uint32_t t = 0; t < P.targets.size(); t++) {
    const Target &T = P.targets[t];
    if (T.n_cols == 0 && !nt.empty() && nt.back().n_cols == 0) {
      Target &M = nt.back();
      for (size_t o = 0; o < M.outputs.size(); o++) {
        Node n; n.op = RH_RIR_ADD; n.a = M.outputs[o]; n.b = T.outputs[o]; n.dep = 0;
        P.nodes.push_back(n);
        M.outputs[o] = (uint32_t)P.nodes.size() - 1;
      }
    } else { nt.push_back(T); first.push_back(t); }
  }
  P.targets.swap(nt);
  old_target_of = first;
  recompute_deps(P);   // Node::dep names a target by its index
  return true;
}

// Program -> RIR bytes (the inverse of parse_rir; used by the rh_simplify_rir test hook)
std::vector<unsigned char> write_rir(const Program &P) {
  std::vector<unsigned char> out;
  auto u32 = [&](uint32_t v) { for (int b = 0; b < 4; b++) out.push_back((unsigned char)(v >> (8 * b))); };
  u32(RH_RIR_MAGIC); u32(RH_RIR_VERSION); u32(P.n_params); u32((uint32_t)P.targets.size()); u32((uint32_t)P.nodes.size()); u32(P.kind);
  for (const Target &t : P.targets) { u32(t.n_cols); u32(0); for (uint32_t o : t.outputs) u32(o); }
  for (const Node &n : P.nodes) {
    u32(n.op);
    switch (n.op) {
      case RH_RIR_CONST: { uint64_t bits; std::memcpy(&bits, &n.cval, 8); u32((uint32_t)bits); u32((uint32_t)(bits >> 32)); break; }
      case RH_RIR_INPUT: u32(n.input); break;
      case RH_RIR_LOOKUP: u32(n.a); u32((uint32_t)n.low); u32((uint32_t)n.table.size()); for (uint32_t e : n.table) u32(e); break;
      case RH_RIR_EXP: case RH_RIR_LOG: case RH_RIR_ABS: case RH_RIR_NOOP: case RH_RIR_SIN: case RH_RIR_COS: case RH_RIR_TAN:
      case RH_RIR_ASIN: case RH_RIR_ACOS: case RH_RIR_ATAN: u32(n.a); break;
      default: u32(n.a); u32(n.b); break;
    }
  }
  return out;
}
}  // namespace rh
