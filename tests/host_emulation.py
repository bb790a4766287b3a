"""TEST INFRASTRUCTURE: compiles the per-model code the emitter generates (the rh_target<t> structs between the "generated from RIR"
marker and the hand-written engine) with the host g++ and evaluates it over the columns row by row, DataFunction-style -- a CPU check
of the emitter's algebra against the oracle.  Not used by the product path (which only runs on the device)."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_HARNESS = r'''
// rh_host_gonly != 0: the rows are walked through row_g() / elem_g() -- the gradient-only row code of the tick engine's mid-trajectory
// requests (csrc/emit.cpp: value_only) -- where the target has one; outputs 1.. must then equal the full evaluation's bit for bit
static int rh_host_gonly = 0;
static const double *rh_host_kpool = nullptr;   // the constant pool the data-free targets read (EmitInfo::kpool; set from the test hook's trailing comment)
template <class TG, class TH, class INV, class CP, class ACC>
static void rh_host_row(const TH &th, const INV &inv, const CP &c, ACC &acc, int &err) {
  if constexpr (TG::HAS_VALUE_ONLY) { if (rh_host_gonly) { TG::row_g(th, inv, c, acc, err); return; } }
  TG::row(th, inv, c, acc, err);
}
template <class TG, class TH, class INV, class CP, class ACC>
static void rh_host_row_gz(const TH &th, const INV &inv, const CP &c, const double gz, ACC &acc, double &sv, int &err) {
  if constexpr (TG::HAS_VALUE_ONLY) { if (rh_host_gonly) { TG::row_g(th, inv, c, gz, acc, sv, err); return; } }
  TG::row(th, inv, c, gz, acc, sv, err);
}
template <int T> static void rh_host_target(const double (&th)[RH_NTH], const double *const *cols, const long long *nrows,
                                            double (&tot)[RH_NOUT], int &err) {
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    double inv[TG::NINV > 0 ? TG::NINV : 1];
    TG::invariants(th, inv, err);
    if constexpr (!TG::HAS_ROWS) {
      TG::row(th, inv, rh_host_kpool, tot, err);
    } else {
      double S[TG::NACC > 0 ? TG::NACC : 1];
      for (int j = 0; j < TG::NACC; j++) S[j] = 0.0;
      double c[TG::NCOLS > 0 ? TG::NCOLS : 1];
      for (long long r = 0; r < nrows[T]; r++) {
        for (int j = 0; j < TG::NCOLS; j++) c[j] = cols[TG::COL0 + j][r];
        rh_host_row<TG>(th, inv, c, S, err);
      }
      TG::finish(th, inv, S, (double)nrows[T], tot);
    }
    rh_host_target<T + 1>(th, cols, nrows, tot, err);
  }
}
#if RH_HAS_GATHER
// gather mode (a parameter table indexed by a data column, cfg 5): the row code takes the gathered table entry `gz` from the
// kernel and hands back the scatter value `sv`; the table parameters' gradients are the per-group sums of sv (rh_grad_gather_kernel)
template <int T> static void rh_host_gather_target(const double *q, const double (&th)[RH_NTH], const double *const *cols, const long long *nrows,
                                                   double (&tot)[RH_NOUT], double *tg, int &err) {
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    double inv[TG::NINV > 0 ? TG::NINV : 1];
    TG::invariants(th, inv, err);
    if constexpr (!TG::HAS_ROWS) {
      TG::row(th, inv, rh_host_kpool, tot, err);
    } else {
      double S[TG::NACC > 0 ? TG::NACC : 1];
      for (int j = 0; j < TG::NACC; j++) S[j] = 0.0;
      double c[TG::NCOLS > 0 ? TG::NCOLS : 1];
      for (long long r = 0; r < nrows[T]; r++) {
        for (int j = 0; j < TG::NCOLS; j++) c[j] = cols[TG::COL0 + j][r];
        double gz = 0.0, sv = 0.0;
        int g = 0;
        if constexpr (TG::HAS_GATHER) {
          g = (int)c[TG::G_COL] - TG::G_LOW;
          if (g < 0 || g >= TG::G_COUNT) { err = 1; continue; }
          gz = q[TG::G_FIRST + g];
        }
        rh_host_row_gz<TG>(th, inv, c, gz, S, sv, err);
        if constexpr (TG::HAS_GATHER) tg[g] += sv;
      }
      TG::finish(th, inv, S, (double)nrows[T], tot);
    }
    rh_host_gather_target<T + 1>(q, th, cols, nrows, tot, tg, err);
  }
}
extern "C" int rh_host_eval(const double *q, const double *const *cols, const long long *nrows, double *out) {
  double th[RH_NTH], tot[RH_NOUT];
  for (int i = 0; i < RH_NTH; i++) th[i] = q[i];
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  for (int i = 0; i < 1 + RH_NVARS; i++) out[i] = 0.0;
  int err = 0;
  rh_host_gather_target<0>(q, th, cols, nrows, tot, out + 1 + RH_NSHARED, err);
  for (int o = 0; o < RH_NOUT; o++) out[o] = tot[o];
  return err;
}
#else
#ifdef RH_GLM_TARGET
// the GLM lowering of the same target (what rh_grad_glm_kernel contracts on the matrix cores): eta = sum_k scale_k theta_k x_k,
// the scalar part elem(eta, columns) -> (w, others), basis sums  S[pred_acc_k] += w x_k,  S[other_acc_j] += other_j, then finish()
extern "C" int rh_host_eval_glm(const double *q, const double *const *cols, const long long *nrows, double *out) {
  typedef rh_glm<RH_GLM_TARGET> GL;
  typedef rh_target<RH_GLM_TARGET> TG;
  double th[RH_NTH], tot[RH_NOUT];
  for (int i = 0; i < RH_NTH; i++) th[i] = q[i];
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  int err = 0;
  double S[TG::NACC > 0 ? TG::NACC : 1];
  for (int j = 0; j < TG::NACC; j++) S[j] = 0.0;
  double thu[GL::NTHU > 0 ? GL::NTHU : 1];
  for (int k = 0; k < GL::NTHU; k++) thu[k] = th[GL::thu_param[k]];
  double c[TG::NCOLS > 0 ? TG::NCOLS : 1];
  for (long long r = 0; r < nrows[RH_GLM_TARGET]; r++) {
    for (int j = 0; j < TG::NCOLS; j++) c[j] = cols[TG::COL0 + j][r];
    double eta = 0.0;
    for (int k = 0; k < GL::P; k++) eta += GL::pred_scale[k] * th[GL::pred_param[k]] * (GL::pred_col[k] >= 0 ? c[GL::pred_col[k]] : 1.0);
    double w = 0.0, other[GL::NOTHER > 0 ? GL::NOTHER : 1];
    if (rh_host_gonly) GL::elem_g(thu, eta, [&](int j) { return c[j]; }, w, other, err);
    else GL::elem(thu, eta, [&](int j) { return c[j]; }, w, other, err);
    for (int k = 0; k < GL::P; k++) S[GL::pred_acc[k]] += w * (GL::pred_col[k] >= 0 ? c[GL::pred_col[k]] : 1.0);
    for (int k = 0; k < GL::NOTHER; k++) S[GL::other_acc[k]] += other[k];
  }
  double inv[TG::NINV > 0 ? TG::NINV : 1];
  TG::invariants(th, inv, err);
  TG::finish(th, inv, S, (double)nrows[RH_GLM_TARGET], tot);
  for (int o = 0; o < RH_NOUT; o++) out[o] = tot[o];
  return err;
}
#endif
extern "C" int rh_host_eval(const double *q, const double *const *cols, const long long *nrows, double *out) {
  double th[RH_NTH], tot[RH_NOUT];
  for (int i = 0; i < RH_NTH; i++) th[i] = q[i];
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  int err = 0;
  rh_host_target<0>(th, cols, nrows, tot, err);
  for (int o = 0; o < RH_NOUT; o++) out[o] = tot[o];
  return err;
}
#endif
'''


class HostTargets:
    """The generated rh_target<t> code of one lowered model, compiled for the host."""

    def __init__(self, hip_source: str):
        # the constant pool's values: the lowering test hook appends them to the source as a comment ("// rh_kpool n: v0 v1 ...")
        self.kpool = np.zeros(1)
        if "\n// rh_kpool " in hip_source:
            hip_source, tail = hip_source.rsplit("\n// rh_kpool ", 1)
            self.kpool = np.array([float.fromhex(x) for x in tail.split(":", 1)[1].split()], dtype=np.float64)
        head = hip_source[:hip_source.index("// rh_shared.h")]
        i = hip_source.index("// ---- generated from RIR")
        gen = hip_source[i:hip_source.index("// rh_engine.hip.h", i)]
        # an anonymous namespace: every model defines rh_target<0>, rh_glm<1>, ... and several models are loaded into one process
        harness = _HARNESS.replace('extern "C" int rh_host_eval_glm(', 'static int host_eval_glm_impl(').replace('extern "C" int rh_host_eval(', 'static int host_eval_impl(')
        text = (head + '#include "host_target_shim.hpp"\nnamespace {\n' + gen + harness + '}\n'
                'extern "C" int rh_host_eval(const double *q, const double *const *cols, const long long *nrows, double *out) { return host_eval_impl(q, cols, nrows, out); }\n'
                'extern "C" void rh_host_set_gonly(int v) { rh_host_gonly = v; }\n'
                'extern "C" void rh_host_set_kpool(const double *p) { rh_host_kpool = p; }\n'
                '#ifdef RH_GLM_TARGET\nextern "C" int rh_host_eval_glm(const double *q, const double *const *cols, const long long *nrows, double *out) { return host_eval_glm_impl(q, cols, nrows, out); }\n#endif\n')
        # (-fno-builtin-sin/cos: g++ would merge sin(x) and cos(x) of a row() into one sincos() call, whose sine may differ by an ulp from the
        #  lone sin(x) of row_g() -- an artefact of the host compiler, not of the generated code)
        flags = ["-std=c++17", "-O1", "-w", "-ffp-contract=off", "-fno-builtin-sin", "-fno-builtin-cos", "-fno-gnu-unique", "-shared", "-fPIC"]
        key = hashlib.sha256((" ".join(flags) + text).encode()).hexdigest()[:16]
        d = os.path.join(tempfile.gettempdir(), "rh_host_targets")
        os.makedirs(d, exist_ok=True)
        so = os.path.join(d, key + ".so")
        if not os.path.exists(so):
            src = os.path.join(d, key + ".cpp")
            open(src, "w").write(text)
            subprocess.check_call(["g++"] + flags + ["-I", os.path.join(HERE, "stubs"), src, "-o", so])
        self.lib = C.CDLL(so)
        self.lib.rh_host_set_kpool(self.kpool.ctypes.data_as(C.POINTER(C.c_double)))
        self.n_out = 1 + int(head.split("#define RH_NVARS ")[1].split("\n")[0])    # (gather mode: RH_NOUT covers the shared outputs only)

    def eval_glm(self, q, columns, nrows):
        """the GLM target alone, through rh_glm<t> (predictor tables + scalar part) instead of its row()"""
        return self._call(self.lib.rh_host_eval_glm, q, columns, nrows)

    def eval_glm_gradient_only(self, q, columns, nrows):
        self.lib.rh_host_set_gonly(1)
        try:
            return self._call(self.lib.rh_host_eval_glm, q, columns, nrows)
        finally:
            self.lib.rh_host_set_gonly(0)

    def eval(self, q, columns, nrows, gradient_only=False):
        """gradient_only: through row_g() / elem_g() where the targets have them (the log-density, output 0, is then not computed)"""
        self.lib.rh_host_set_gonly(1 if gradient_only else 0)
        try:
            return self._call(self.lib.rh_host_eval, q, columns, nrows)
        finally:
            self.lib.rh_host_set_gonly(0)

    def _call(self, fn, q, columns, nrows):
        cols = [np.ascontiguousarray(c, dtype=np.float64) for c in columns]
        arr = (C.POINTER(C.c_double) * max(1, len(cols)))(*[c.ctypes.data_as(C.POINTER(C.c_double)) for c in cols])
        nr = (C.c_longlong * max(1, len(nrows)))(*[int(x) for x in nrows])
        qq = np.ascontiguousarray(q, dtype=np.float64)
        out = np.zeros(self.n_out)
        err = fn(qq.ctypes.data_as(C.POINTER(C.c_double)), arr, nr, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out, err
