"""Host-side mirror of rainier-sampler's plugin surface over the C ABI (include/rainier_hip.h).

Same names, argument meaning and defaults as the reference (rainier-sampler/.../sampler/):
  SamplerConfig / DefaultConfig            Sampler.scala:3-27
  HMCSampler(nSteps), HMC(warmIt, it, n)   HMC.scala:3-33
  EHMCSampler(maxSteps, minSteps, bufSize, pCount), EHMC(...)   EHMC.scala:3-74
  DualAvgTuner(delta), StaticStepSize      DualAvg.scala:3-25, Sampler.scala:36-40
  IdentityMassMatrixTuner, DiagonalMassMatrixTuner(50,1.5,50,50), StaticMassMatrix   MassMatrix.scala:120-173
  DensityFunction { nVars, update, density, gradient }         DensityFunction.scala:3-8
  Model.sample(config, nChains) -> Trace; Trace.diagnostics    core/Model.scala:13-24, core/Trace.scala:11-21

All compute happens in librainier_hip.so on the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from ._capi import RainierHipError  # noqa: F401  (re-export)


# ---- plugin classes (pure configuration, like the reference's) -------------------------------------
@dataclass
class HMCSampler:
    nSteps: int


@dataclass
class EHMCSampler:
    maxSteps: int
    minSteps: int = 1
    bufSize: int = 100
    pCount: float = 0.1


@dataclass
class NUTSSampler:
    """Extension (not in the reference): iterative multinomial NUTS behind the Sampler plugin point."""
    maxDepth: int = 10


@dataclass
class DualAvgTuner:
    delta: float


@dataclass
class StaticStepSize:
    stepSize: float


@dataclass
class IdentityMassMatrixTuner:
    pass


@dataclass
class DiagonalMassMatrixTuner:
    initialWindowSize: int = 50
    windowExpansion: float = 1.5
    skipFirst: int = 50
    skipLast: int = 50


@dataclass
class DenseMassMatrixTuner:
    """DenseMassMatrixTuner (MassMatrix.scala:175-181): CovarianceEstimator + packed Cholesky; at most 64 parameters."""
    initialWindowSize: int = 50
    windowExpansion: float = 1.5
    skipFirst: int = 50
    skipLast: int = 50


@dataclass
class DiagonalMassMatrix:
    elements: Sequence[float]

    def __post_init__(self):
        if any(float(x) == 0.0 for x in self.elements):   # require(!elements.contains(0.0)) MassMatrix.scala:8
            raise ValueError("requirement failed")


@dataclass
class StaticMassMatrix:
    mass: DiagonalMassMatrix


class SamplerConfig:
    """trait SamplerConfig (Sampler.scala:3-11); defaults = DefaultConfig (Sampler.scala:17-27)."""

    iterations = 1000
    warmupIterations = 1000
    statsWindow = 100
    engine = _capi.ENGINE_AUTO   # engine extension: device mapping (rh_engine_kind); not part of the reference trait
    gradSplits = 0

    def stepSizeTuner(self): return DualAvgTuner(0.8)
    def massMatrixTuner(self): return DiagonalMassMatrixTuner(50, 1.5, 50, 50)
    def sampler(self): return EHMCSampler(1024)


DefaultConfig = SamplerConfig


def make_config(iterations=1000, warmupIterations=1000, sampler=None, stepSizeTuner=None, massMatrixTuner=None,
                engine=_capi.ENGINE_AUTO, gradSplits=0):
    cfg = SamplerConfig()
    cfg.iterations, cfg.warmupIterations = iterations, warmupIterations
    cfg.engine, cfg.gradSplits = engine, gradSplits
    if sampler is not None: cfg.sampler = lambda: sampler
    if stepSizeTuner is not None: cfg.stepSizeTuner = lambda: stepSizeTuner
    if massMatrixTuner is not None: cfg.massMatrixTuner = lambda: massMatrixTuner
    return cfg


def HMC(warmIt: int, it: int, nSteps: int) -> SamplerConfig:          # HMC.scala:26-33
    return make_config(it, warmIt, sampler=HMCSampler(nSteps))


def EHMC(warmIt: int, it: int, minSteps: int = 1, numLengths: int = 100) -> SamplerConfig:  # EHMC.scala:64-74
    return make_config(it, warmIt, sampler=EHMCSampler(1000, minSteps, numLengths, 0.1))


def to_c_config(config: SamplerConfig, nvars: int):
    c = _capi.Config()
    _capi.lib().rh_config_default(C.byref(c))
    c.iterations, c.warmup = int(config.iterations), int(config.warmupIterations)
    c.engine, c.grad_splits = int(getattr(config, 'engine', 0)), int(getattr(config, 'gradSplits', 0))
    s, st, mt = config.sampler(), config.stepSizeTuner(), config.massMatrixTuner()
    keep = None
    if isinstance(s, HMCSampler):
        c.sampler, c.hmc_steps = _capi.SAMPLER_HMC, int(s.nSteps)
    elif isinstance(s, EHMCSampler):
        c.sampler = _capi.SAMPLER_EHMC
        c.ehmc_max_steps, c.ehmc_min_steps, c.ehmc_buf_size, c.ehmc_p_count = s.maxSteps, s.minSteps, s.bufSize, s.pCount
    elif isinstance(s, NUTSSampler):
        c.sampler, c.nuts_max_depth = _capi.SAMPLER_NUTS, int(s.maxDepth)
    else:
        raise TypeError("unsupported Sampler %r" % (s,))
    if isinstance(st, DualAvgTuner):
        c.step_tuner, c.dualavg_delta = _capi.STEP_DUALAVG, float(st.delta)
    elif isinstance(st, StaticStepSize):
        c.step_tuner, c.static_step = _capi.STEP_STATIC, float(st.stepSize)
    else:
        raise TypeError("unsupported StepSizeTuner %r" % (st,))
    if isinstance(mt, IdentityMassMatrixTuner):
        c.mass_tuner = _capi.MASS_IDENTITY
    elif isinstance(mt, DiagonalMassMatrixTuner):
        c.mass_tuner = _capi.MASS_DIAG_WINDOWED
        c.mass_init_window, c.mass_expansion = mt.initialWindowSize, mt.windowExpansion
        c.mass_skip_first, c.mass_skip_last = mt.skipFirst, mt.skipLast
    elif isinstance(mt, DenseMassMatrixTuner):
        c.mass_tuner = _capi.MASS_DENSE_WINDOWED
        c.mass_init_window, c.mass_expansion = mt.initialWindowSize, mt.windowExpansion
        c.mass_skip_first, c.mass_skip_last = mt.skipFirst, mt.skipLast
    elif isinstance(mt, StaticMassMatrix):
        keep = np.ascontiguousarray(mt.mass.elements, dtype=np.float64)
        assert keep.shape == (nvars,)
        c.mass_tuner, c.static_mass = _capi.MASS_STATIC_DIAG, _capi.dptr(keep)
    else:
        raise TypeError("unsupported MassMatrixTuner %r" % (mt,))
    return c, keep


# ---- model / density / trace ---------------------------------------------------------------------------
class DensityFunction:
    """trait DensityFunction (DensityFunction.scala:3-8) as built by Model.density() (core/Model.scala:38-50)."""

    def __init__(self, model: "Model"):
        self._m = model
        self.nVars = model.nVars
        self._out = np.zeros(self.nVars + 1)

    def update(self, vars: Sequence[float]) -> None:
        lp, g = self._m.density_batch(np.asarray(vars, dtype=np.float64).reshape(1, self.nVars))
        self._out[0], self._out[1:] = lp[0], g[0]

    @property
    def density(self) -> float: return float(self._out[0])
    def gradient(self, index: int) -> float: return float(self._out[index + 1])


@dataclass
class Stats:
    leapfrogSteps: int
    warmupLeapfrogSteps: int
    gradientEvaluations: int
    accepted: int
    meanAcceptProb: float
    stepSize: float
    bfmi: float = float("nan")   # Stats.bfmi (sampler/Stats.scala:14-16)


class Trace:
    """case class Trace(chains, mass, stats, model) (core/Trace.scala:6-9); chains [nChains][iterations][nVars]."""

    def __init__(self, chains: np.ndarray, mass: np.ndarray, stats: List[Stats]):
        self.chains, self.mass, self.stats = chains, mass, stats

    def diagnostics(self):
        """List of (rHat, effectiveSampleSize) per parameter -- Trace.diagnostics (core/Trace.scala:11-21)."""
        return diagnostics(self.chains)


def diagnostics(chains: np.ndarray):
    ch = np.ascontiguousarray(chains, dtype=np.float64)
    m, n, k = ch.shape
    if m < 2:
        raise ValueError("requirement failed: diagnostics requires multiple chains")
    rhat, ess = np.zeros(k), np.zeros(k)
    _capi.check(_capi.lib().rh_diagnostics(_capi.dptr(ch), m, n, k, _capi.dptr(rhat), _capi.dptr(ess)))
    return list(zip(rhat.tolist(), ess.tolist()))


def predict(requirements_rir: bytes, draws: np.ndarray, n_requirements: int, device: int = -1,
            math_mode: int = _capi.MATH_FAST) -> np.ndarray:
    """Trace.predict's compiled part (core/Trace.scala:34-41, core/Generator.scala:59-94): evaluate the requirements
    program for every draw on the device.  draws [..., nVars] -> [..., n_requirements]."""
    d = np.ascontiguousarray(draws, dtype=np.float64)
    flat = d.reshape(-1, d.shape[-1])
    out = np.zeros((flat.shape[0], n_requirements))
    blob = C.create_string_buffer(requirements_rir, len(requirements_rir))
    opts = _capi.compile_opts(device, math_mode)
    _capi.check(_capi.lib().rh_requirements_eval(blob, len(requirements_rir), C.byref(opts), _capi.dptr(flat), flat.shape[0], _capi.dptr(out)))
    return out.reshape(d.shape[:-1] + (n_requirements,))


class Sampler:
    """Device-resident chains: split form of Driver.sample used by bench.py (create -> warmup -> run)."""

    def __init__(self, model: "Model", config: SamplerConfig, seeds: Sequence[int] = None, rng_states=None):
        """seeds: chain c runs on ScalaRNG(seeds[c]).  rng_states: instead, continue existing java.util.Random streams --
        one (internal 48-bit state, pending nextNextGaussian or None) per chain (what Driver.sample does with the caller's rng)."""
        self.model = model
        self._nn = None
        if rng_states is not None:
            seeds = [int(st) ^ 0x5DEECE66D for st, _ in rng_states]
            self._nn = np.array([np.nan if g is None else float(g) for _, g in rng_states], dtype=np.float64)
        self.chains = len(seeds)
        self.iterations = int(config.iterations)
        self._cfg, self._keep = to_c_config(config, model.nVars)
        if self._nn is not None:
            self._cfg.rng_next_gaussian = _capi.dptr(self._nn)
        self._seeds = (C.c_int64 * self.chains)(*[int(s) for s in seeds])
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rh_sampler_create(model._h, C.byref(self._cfg), self._seeds, self.chains, C.byref(self._h)), model._h)

    def warmup(self): _capi.check(_capi.lib().rh_sampler_warmup(self._h), self.model._h)
    def run(self, n: int): _capi.check(_capi.lib().rh_sampler_run(self._h, int(n)), self.model._h)

    def draws(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.iterations - first if count is None else count
        out = np.zeros((self.chains, count, self.model.nVars))
        _capi.check(_capi.lib().rh_sampler_draws(self._h, first, count, _capi.dptr(out)), self.model._h)
        return out

    def draws_device_ptr(self) -> int:
        p = C.c_void_p()
        _capi.check(_capi.lib().rh_sampler_draws_device(self._h, C.byref(p)), self.model._h)
        return p.value

    def stats(self):
        st = (_capi.ChainStats * self.chains)()
        mass = np.zeros((self.chains, self.model.nVars))
        _capi.check(_capi.lib().rh_sampler_stats(self._h, st, _capi.dptr(mass)), self.model._h)
        return [Stats(s.leapfrog_steps, s.warmup_leapfrog_steps, s.gradient_evaluations, s.accepted,
                      s.mean_accept_prob, s.step_size, s.bfmi) for s in st], mass

    def progress(self):
        """(warmed, sampling iterations done): what a Progress callback would be told (sampler/Driver.scala:7-11), polled."""
        w, it = C.c_int32(0), C.c_int32(0)
        _capi.check(_capi.lib().rh_sampler_progress(self._h, C.byref(w), C.byref(it)), self.model._h)
        return bool(w.value), int(it.value)

    def mass_dense(self) -> np.ndarray:
        """DenseMassMatrix.elements of every chain: [chains][nVars][nVars] (DenseMassMatrixTuner only)."""
        n = self.model.nVars
        out = np.zeros((self.chains, n, n))
        _capi.check(_capi.lib().rh_sampler_mass_dense(self._h, _capi.dptr(out)), self.model._h)
        return out

    def timing(self, reset: bool = False):
        t = _capi.Timing()
        _capi.check(_capi.lib().rh_sampler_timing(self._h, C.byref(t), int(reset)), self.model._h)
        return {"kernel_ms": t.kernel_ms, "total_ms": t.total_ms, "launches": t.launches, "density_evals": t.density_evals,
                "row_chain_evals": t.row_chain_evals, "dominant_kernel": t.dominant_kernel.decode(), "chain_slots": t.chain_slots,
                "steady_kernel_ms": t.steady_kernel_ms, "steady_launches": t.steady_launches, "steady_density_evals": t.steady_density_evals}

    def close(self):
        if self._h:
            _capi.lib().rh_sampler_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def sample_multi(models, config: SamplerConfig = None, seeds: Sequence[int] = None) -> "Trace":
    """Model.sample fanned out over several devices behind the C ABI (rh_sample_multi): models[g] is the same program
    compiled for device g; chains are cut into contiguous shards by GLOBAL chain id, so the trace does not depend on
    len(models)."""
    config = config or SamplerConfig()
    nv = models[0].nVars
    chains = len(seeds)
    cfg, _keep = to_c_config(config, nv)
    draws = np.zeros((chains, int(config.iterations), nv))
    mass = np.zeros((chains, nv))
    st = (_capi.ChainStats * chains)()
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    sd = (C.c_int64 * chains)(*[int(x) for x in seeds])
    # a shard's failure is reported through the calling thread's error slot (rh_last_error(NULL)), not through models[0]
    _capi.check(_capi.lib().rh_sample_multi(handles, len(models), C.byref(cfg), sd, chains, _capi.dptr(draws), _capi.dptr(mass), st))
    stats = [Stats(s.leapfrog_steps, s.warmup_leapfrog_steps, s.gradient_evaluations, s.accepted, s.mean_accept_prob, s.step_size, s.bfmi)
             for s in st]
    return Trace(draws, mass, stats)


class Model:
    """A compiled model: Compiler.compileTargets' replacement (compute/Compiler.scala:14-30) + Model.sample."""

    def __init__(self, spec, device: int = -1, math_mode: int = _capi.MATH_FAST, fp_contract: bool = False,
                 rows_unroll: int = 0, grad_chains: int = 0, grad_unroll: int = 0, factor_outputs: bool = False,
                 with_nuts: bool = False):
        L = _capi.lib()
        self.spec = spec
        self.nVars = spec.n_params
        self._cols = [np.ascontiguousarray(c, dtype=np.float64) for c in spec.columns]
        colarr = (C.POINTER(C.c_double) * max(1, len(self._cols)))(*[_capi.dptr(c) for c in self._cols])
        nrows = (C.c_int64 * len(spec.nrows))(*spec.nrows)
        opts = _capi.compile_opts(device, math_mode, fp_contract, rows_unroll, grad_chains, grad_unroll, factor_outputs,
                                   with_nuts)
        blob = C.create_string_buffer(spec.rir, len(spec.rir))
        self._h = C.c_void_p()
        _capi.check(L.rh_model_create(blob, len(spec.rir), colarr, nrows, C.byref(opts), C.byref(self._h)))

    def clone(self, device: int = -1) -> "Model":
        """the same compiled model on another device of this process (rh_model_clone): code object reused, columns copied
        device to device -- one per GPU for sample_multi"""
        other = object.__new__(Model)
        other.spec, other.nVars, other._cols = self.spec, self.nVars, self._cols
        other._h = C.c_void_p()
        _capi.check(_capi.lib().rh_model_clone(self._h, int(device), C.byref(other._h)))
        return other

    @property
    def hip_source(self) -> str: return _capi.lib().rh_model_hip_source(self._h).decode()

    def engines(self) -> dict:
        """rh_model_engines: which engines the engine agrees to run for this model on this toolchain and why not
        ({"chain": bool, "tick": bool, "density": bool, "compile_attempts": int, "why": str})."""
        c, t, d, a = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        why = C.create_string_buffer(2048)
        _capi.check(_capi.lib().rh_model_engines(self._h, C.byref(c), C.byref(t), C.byref(d), C.byref(a), why, len(why)), self._h)
        return {"chain": bool(c.value), "tick": bool(t.value), "density": bool(d.value), "compile_attempts": a.value,
                "why": why.value.decode(errors="replace")}

    def density(self) -> DensityFunction: return DensityFunction(self)

    def density_batch(self, q: np.ndarray, engine: int = _capi.ENGINE_AUTO, grad_splits: int = 0):
        """Batched DensityFunction.update.  engine = ENGINE_TICK evaluates through the sampler's batched gradient kernels
        (rh_grad_kernel / rh_grad_glm_kernel / rh_grad_gather_kernel + the tick combine) instead of one chain per wavefront."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        chains = q.shape[0]
        lp, g = np.zeros(chains), np.zeros((chains, self.nVars))
        _capi.check(_capi.lib().rh_density_eval_ex(self._h, _capi.dptr(q), chains, int(engine), int(grad_splits),
                                                   _capi.dptr(lp), _capi.dptr(g)), self._h)
        return lp, g

    def sample(self, config: SamplerConfig = None, nChains: int = 4, seeds: Sequence[int] = None, rng_states=None) -> Trace:
        """Model.sample(config, nChains) (core/Model.scala:13-24).  Chain c is the reference run with
        nChains = 1 and ScalaRNG(seeds[c])."""
        config = config or SamplerConfig()
        seeds = list(range(1, nChains + 1)) if seeds is None else list(seeds)
        s = Sampler(self, config, seeds, rng_states)
        try:
            s.warmup(); s.run(config.iterations)
            draws = s.draws() if config.iterations > 0 else np.zeros((s.chains, 0, self.nVars))
            stats, mass = s.stats()
        finally:
            s.close()
        return Trace(draws, mass, stats)

    def optimize(self, starts: np.ndarray = None, max_evals: int = 0):
        """Model.optimize's numeric part = Optimizer.lbfgs(density()) (core/Model.scala:26-30, optimizer/Optimizer.scala:6-24).
        starts None: the reference's single start at 0 -> x [nVars]; starts [k][nVars]: k independent searches sharing
        each batched density launch -> (x [k][nVars], evals [k], status [k])"""
        single = starts is None
        x0 = np.zeros((1, self.nVars)) if single else np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, self.nVars)
        k = x0.shape[0]
        x, evals, status = np.zeros((k, self.nVars)), np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        _capi.check(_capi.lib().rh_optimize(self._h, _capi.dptr(x0), k, max_evals, _capi.dptr(x), ip(evals), ip(status)), self._h)
        if single:
            if status[0] == _capi.OPT_NOT_DESCENT:
                raise RuntimeError("dginit")       # LBFGS.java:236-237
            return x[0]
        return x, evals, status

    def selftest(self, mode: int, seed: int = 0, x: np.ndarray = None, n: int = None) -> np.ndarray:
        x = np.zeros(1) if x is None else np.ascontiguousarray(x, dtype=np.float64)
        n = (x.size // 2 if mode in (5, 14) else x.size) if n is None else n
        out = np.zeros(n)
        _capi.check(_capi.lib().rh_selftest(self._h, mode, seed, _capi.dptr(x), _capi.dptr(out), n), self._h)
        return out

    def close(self):
        if self._h:
            _capi.lib().rh_model_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass
