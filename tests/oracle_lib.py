"""ctypes loader for oracle/liboracle.so (the CPU restatement of the reference).  Tests only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
JM_LIBM, JM_DET = 0, 1
HMC, EHMC, NUTS = 0, 1, 2
STEP_DUALAVG, STEP_STATIC = 0, 1
MASS_IDENTITY, MASS_DIAG_WINDOWED, MASS_STATIC_DIAG, MASS_DENSE_WINDOWED = 0, 1, 2, 3

DENSITY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


class JRandom(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("have_next", C.c_int), ("next_next", C.c_double)]


class OrcConfig(C.Structure):
    _fields_ = [
        ("sampler", C.c_int), ("n_steps", C.c_int),
        ("max_steps", C.c_int), ("min_steps", C.c_int), ("buf_size", C.c_int), ("p_count", C.c_double),
        ("step_tuner", C.c_int), ("delta", C.c_double), ("static_step", C.c_double),
        ("mass_tuner", C.c_int), ("init_window", C.c_int), ("expansion", C.c_double),
        ("skip_first", C.c_int), ("skip_last", C.c_int), ("static_mass", C.POINTER(C.c_double)),
        ("nuts_max_depth", C.c_int), ("iterations", C.c_int), ("warmup", C.c_int), ("math_mode", C.c_int), ("dense_out", C.POINTER(C.c_double)),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("gradient_evaluations", C.c_int64), ("leapfrog_steps", C.c_int64),
        ("warmup_leapfrog_steps", C.c_int64), ("warmup_gradient_evaluations", C.c_int64),
        ("accepted", C.c_int64), ("mean_accept_prob", C.c_double), ("step_size", C.c_double),
        ("density_error", C.c_int), ("bfmi", C.c_double),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR])
    return os.path.join(ODIR, "liboracle.so")


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    dp = C.POINTER(C.c_double)
    lib.jm_strict_log.restype = C.c_double; lib.jm_strict_log.argtypes = [C.c_double]
    lib.jm_strict_exp.restype = C.c_double; lib.jm_strict_exp.argtypes = [C.c_double]
    lib.jm_pow_neg075.restype = C.c_double; lib.jm_pow_neg075.argtypes = [C.c_int, C.c_double]
    lib.jrandom_init.argtypes = [C.POINTER(JRandom), C.c_int64]
    lib.jrandom_next_int.restype = C.c_int32; lib.jrandom_next_int.argtypes = [C.POINTER(JRandom)]
    lib.jrandom_next_double.restype = C.c_double; lib.jrandom_next_double.argtypes = [C.POINTER(JRandom)]
    lib.jrandom_next_gaussian.restype = C.c_double; lib.jrandom_next_gaussian.argtypes = [C.POINTER(JRandom)]
    lib.rir_parse.restype = C.c_void_p; lib.rir_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.rir_free.argtypes = [C.c_void_p]
    lib.rir_density_new.restype = C.c_void_p
    lib.rir_density_new.argtypes = [C.c_void_p, C.POINTER(dp), C.POINTER(C.c_int64), C.c_int]
    lib.rir_density_free.argtypes = [C.c_void_p]
    for f in (lib.rir_density_update, lib.rir_density_abs_sums, lib.rir_density_update_ld, lib.rir_density_update_x):
        f.restype = C.c_int; f.argtypes = [C.c_void_p, dp, dp]
    lib.rir_requirements_eval.restype = C.c_int; lib.rir_requirements_eval.argtypes = [C.c_void_p, dp, dp]
    lib.orc_sample_chain.restype = C.c_int
    lib.orc_sample_chain.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_int64, dp, dp,
                                     C.POINTER(OrcStats)]
    lib.orc_sample_chain_state.restype = C.c_int
    lib.orc_sample_chain_state.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(JRandom), dp, dp,
                                           C.POINTER(OrcStats)]
    lib.orc_lf_new.restype = C.c_void_p
    lib.orc_lf_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(JRandom), C.c_int]
    lib.orc_lf_free.argtypes = [C.c_void_p]
    lib.orc_lf_initialize.argtypes = [C.c_void_p, dp, dp]
    lib.orc_lf_start_iteration.argtypes = [C.c_void_p, dp, dp]
    lib.orc_lf_take_steps.argtypes = [C.c_void_p, C.c_int, C.c_double, dp]
    lib.orc_lf_finish_iteration.restype = C.c_double; lib.orc_lf_finish_iteration.argtypes = [C.c_void_p, dp, dp]
    lib.orc_cholesky_upper.argtypes = [dp, C.c_int, dp]
    lib.orc_upper_triangular_solve.argtypes = [dp, dp, C.c_int, dp]
    lib.orc_square_multiply.argtypes = [dp, dp, C.c_int, dp]
    lib.orc_diagnostics.argtypes = [dp, C.c_int, C.c_int, dp, dp]
    lib.orc_linreg_streamed.argtypes = [dp, dp, dp, dp, C.c_long, dp, dp]
    lib.orc_linreg_streamed_reps.restype = C.c_double; lib.orc_linreg_streamed_reps.argtypes = [dp, dp, dp, dp, C.c_long, C.c_int]
    lib.orc_lbfgs_new.restype = C.c_void_p; lib.orc_lbfgs_new.argtypes = [dp, C.c_int, C.c_int, C.c_double]
    lib.orc_lbfgs_free.argtypes = [C.c_void_p]
    lib.orc_lbfgs_apply.restype = C.c_int; lib.orc_lbfgs_apply.argtypes = [C.c_void_p, C.c_double, dp]
    lib.orc_optimize.restype = C.c_int; lib.orc_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, dp, C.c_int, dp]
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class JavaRandom:
    def __init__(self, seed):
        self.lib = load(); self.r = JRandom(); self.lib.jrandom_init(C.byref(self.r), seed)

    def next_int(self): return self.lib.jrandom_next_int(C.byref(self.r))
    def next_double(self): return self.lib.jrandom_next_double(C.byref(self.r))
    def next_gaussian(self): return self.lib.jrandom_next_gaussian(C.byref(self.r))


class OracleDensity:
    """DataFunction + Model.density() over an RIR program (oracle/rir.c)."""

    def __init__(self, spec, math_mode=JM_LIBM):
        self.lib = load()
        self.spec = spec
        err = C.create_string_buffer(256)
        self._blob = C.create_string_buffer(spec.rir, len(spec.rir))
        self.prog = self.lib.rir_parse(self._blob, len(spec.rir), err, 256)
        if not self.prog:
            raise ValueError(err.value.decode())
        self._cols = [np.ascontiguousarray(c, dtype=np.float64) for c in spec.columns]
        arr = (C.POINTER(C.c_double) * max(1, len(self._cols)))(*[_dp(c) for c in self._cols])
        self._colarr = arr
        self._nrows = (C.c_int64 * len(spec.nrows))(*spec.nrows)
        self.handle = self.lib.rir_density_new(self.prog, arr, self._nrows, math_mode)
        self.n = spec.n_params
        self.fn_ptr = C.cast(self.lib.rir_density_update, C.c_void_p)

    def _call(self, f, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = np.zeros(self.n + 1)
        rc = f(self.handle, _dp(q), _dp(out))
        return rc, out

    def update(self, q):
        rc, out = self._call(self.lib.rir_density_update, q)
        if rc:
            raise RuntimeError("lookup index out of range")
        return out

    def requirements(self, q, n_req):
        q = np.ascontiguousarray(q, dtype=np.float64); out = np.zeros(n_req)
        if self.lib.rir_requirements_eval(self.handle, _dp(q), _dp(out)):
            raise RuntimeError("lookup index out of range")
        return out

    def abs_sums(self, q): return self._call(self.lib.rir_density_abs_sums, q)[1]

    def update_both(self, q):
        """(sequential-sum outputs, sums of |term|) in one pass over the rows."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        out, ab = np.zeros(self.n + 1), np.zeros(self.n + 1)
        self.lib.rir_density_update_both.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if self.lib.rir_density_update_both(self.handle, _dp(q), _dp(out), _dp(ab)):
            raise RuntimeError("lookup index out of range")
        return out, ab

    def update_ld(self, q): return self._call(self.lib.rir_density_update_ld, q)[1]

    def update_x(self, q):
        """the program evaluated and summed in extended precision (rir_density_update_x): a yardstick, not the reference's arithmetic"""
        return self._call(self.lib.rir_density_update_x, q)[1]

    def __del__(self):
        try:
            self.lib.rir_density_free(self.handle); self.lib.rir_free(self.prog)
        except Exception:
            pass


def make_config(sampler=HMC, n_steps=1, max_steps=1024, min_steps=1, buf_size=100, p_count=0.1,
                step_tuner=STEP_DUALAVG, delta=0.8, static_step=0.1,
                mass_tuner=MASS_IDENTITY, init_window=50, expansion=1.5, skip_first=50, skip_last=50,
                static_mass=None, iterations=100, warmup=100, math_mode=JM_LIBM, nuts_max_depth=10):
    cfg = OrcConfig()
    cfg.sampler, cfg.n_steps = sampler, n_steps
    cfg.max_steps, cfg.min_steps, cfg.buf_size, cfg.p_count = max_steps, min_steps, buf_size, p_count
    cfg.step_tuner, cfg.delta, cfg.static_step = step_tuner, delta, static_step
    cfg.mass_tuner, cfg.init_window, cfg.expansion = mass_tuner, init_window, expansion
    cfg.skip_first, cfg.skip_last = skip_first, skip_last
    cfg._static_mass_keep = None
    if static_mass is not None:
        sm = np.ascontiguousarray(static_mass, dtype=np.float64)
        cfg._static_mass_keep = sm
        cfg.static_mass = _dp(sm)
    cfg.iterations, cfg.warmup, cfg.math_mode = iterations, warmup, math_mode
    cfg.nuts_max_depth = nuts_max_depth
    return cfg


def sample_chain(density_fn_ptr, ctx, nvars, cfg, seed):
    """Driver.sample for one chain with ScalaRNG(seed). Returns (draws[iters][n], mass[n], stats)."""
    lib = load()
    draws = np.zeros((cfg.iterations, nvars))
    mass = np.zeros(nvars)
    st = OrcStats()
    rc = lib.orc_sample_chain(C.byref(cfg), density_fn_ptr, ctx, nvars, seed, _dp(draws), _dp(mass), C.byref(st))
    return draws, mass, st, rc


def sample_chain_state(density_fn_ptr, ctx, nvars, cfg, rstate):
    """Driver.sample continuing an existing java.util.Random state (JRandom struct)."""
    lib = load()
    draws = np.zeros((cfg.iterations, nvars)); mass = np.zeros(nvars); st = OrcStats()
    rc = lib.orc_sample_chain_state(C.byref(cfg), density_fn_ptr, ctx, nvars, C.byref(rstate), _dp(draws), _dp(mass), C.byref(st))
    return draws, mass, st, rc


def sample_model(spec, cfg, seed, math_mode=None):
    d = OracleDensity(spec, cfg.math_mode if math_mode is None else math_mode)
    draws, mass, st, rc = sample_chain(d.fn_ptr, d.handle, spec.n_params, cfg, seed)
    return draws, mass, st


def optimize_model(spec, x0=None, max_evals=100000, math_mode=JM_LIBM):
    """Optimizer.lbfgs over the RIR interpreter: (x, evals) -- evals < 0: -1 dginit, -2 max_evals, -3 density error."""
    d = OracleDensity(spec, math_mode)
    x = np.zeros(spec.n_params)
    x0p = None if x0 is None else _dp(np.ascontiguousarray(x0, dtype=np.float64))
    rc = load().orc_optimize(d.fn_ptr, d.handle, spec.n_params, x0p, max_evals, _dp(x))
    return x, rc


class Lbfgs:
    """new LBFGS(x, m, eps) driven from Python: apply(f, g) -> 1 converged / 0 evaluate again / -1 dginit."""

    def __init__(self, n, m=5, eps=0.1):
        self.x = np.zeros(n)
        self._h = load().orc_lbfgs_new(_dp(self.x), n, m, eps)

    def apply(self, f, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        return load().orc_lbfgs_apply(self._h, float(f), _dp(g))

    def __del__(self):
        try: load().orc_lbfgs_free(self._h)
        except Exception: pass


def diagnostics(traces):
    """traces: [m chains][n draws] -> (rhat, ess)  (core/Trace.scala:52-120)."""
    lib = load()
    t = np.ascontiguousarray(traces, dtype=np.float64)
    r, e = C.c_double(), C.c_double()
    lib.orc_diagnostics(_dp(t), t.shape[0], t.shape[1], C.byref(r), C.byref(e))
    return r.value, e.value
