"""The JNI shim (rainier_amd/jni/rainier_hip_jni.c) EXECUTED without a JDK.

The shim is compiled against tests/stubs/jni.h and driven through tests/stubs/fake_jni.c -- a JNIEnv function table over
malloc'ed arrays with a copying JVM's semantics (Get = copy, Release(0) = copy back, JNI_ABORT = discard) -- from Python via
ctypes.  What a JVM caller would observe is therefore what these tests observe:
  * CPU: array marshalling, release modes, exception mapping, leak-free error paths, Trace.diagnostics end to end;
         the Scala side's flat-array layouts (integration/scala/*.scala) against the shim's X-macro lists, textually.
  * GPU: modelCreate -> sample / densityEval / optimize / requirementsEval through the shim, bit-identical to the ctypes
         path, incl. NUTS, the fast build, two device handles (rh_sample_multi) and a shared java.util.Random stream.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from rainier_amd import _capi, models
import rainier_amd as R
from tests import oracle_lib as O
from tests.test_gpu_parity import _oracle_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = "Java_com_stripe_rainier_hip_Native_00024_"
FJ_BYTE, FJ_INT, FJ_LONG, FJ_DOUBLE, FJ_OBJECT = 1, 2, 3, 4, 5


@pytest.fixture(scope="module")
def fj(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fj") / "librainier_hip_jni_fake.so")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tests", "stubs"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "rainier_amd", "jni", "rainier_hip_jni.c"),
                           os.path.join(ROOT, "tests", "stubs", "fake_jni.c"), "-L", os.path.join(ROOT, "rainier_amd"), "-lrainier_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "rainier_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", so])
    L = C.CDLL(so)
    vp = C.c_void_p
    L.fj_env.restype = vp
    L.fj_new_array.restype = vp; L.fj_new_array.argtypes = [C.c_int, C.c_int32, vp]
    L.fj_set_object.argtypes = [vp, C.c_int32, vp]
    L.fj_array_data.restype = vp; L.fj_array_data.argtypes = [vp]
    L.fj_free.argtypes = [vp]
    L.fj_exception_class.restype = C.c_char_p; L.fj_exception_message.restype = C.c_char_p
    L.fj_set_pinning.argtypes = [C.c_int]
    for name, res, args in [
            ("abiVersion", C.c_int32, []), ("deviceCount", C.c_int32, []),
            ("modelCreate", C.c_int64, [vp, vp, vp, vp]), ("modelDestroy", None, [C.c_int64]), ("modelNVars", C.c_int32, [C.c_int64]),
            ("modelClone", C.c_int64, [C.c_int64, C.c_int32]), ("modelEngines", C.c_int32, [C.c_int64]),
            ("densityEval", None, [C.c_int64, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]),
            ("optimize", None, [C.c_int64, vp, C.c_int32, C.c_int32, vp, vp, vp]),
            ("sample", None, [vp] * 9), ("requirementsEval", None, [vp, vp, vp, C.c_int64, vp]),
            ("diagnostics", None, [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp])]:
        f = getattr(L, P + name)
        f.restype = res
        f.argtypes = [vp, vp] + args          # JNIEnv *, jobject self
    return L


class JArr:
    """a "Java" primitive array living in the fake JVM"""
    KIND = {np.dtype(np.int8): FJ_BYTE, np.dtype(np.int32): FJ_INT, np.dtype(np.int64): FJ_LONG, np.dtype(np.float64): FJ_DOUBLE}

    def __init__(self, L, a):
        self.L = L
        a = np.ascontiguousarray(a)
        self.dtype, self.shape = a.dtype, a.shape
        self.h = L.fj_new_array(self.KIND[a.dtype], a.size, a.ctypes.data_as(C.c_void_p))

    def get(self):
        n = int(np.prod(self.shape))
        buf = (C.c_char * (n * self.dtype.itemsize)).from_address(self.L.fj_array_data(self.h))
        return np.frombuffer(buf, dtype=self.dtype).reshape(self.shape).copy()


def call(L, name, *args):
    L.fj_clear()
    r = getattr(L, P + name)(L.fj_env(), None, *[a.h if isinstance(a, JArr) else a for a in args])
    assert L.fj_outstanding() == 0, "Get<T>ArrayElements without Release (%s)" % name
    assert L.fj_bad_releases() == 0, "wrong Release call (%s)" % name
    exc = L.fj_exception_class().decode()
    if exc:
        raise RuntimeError("%s: %s" % (exc, L.fj_exception_message().decode()))
    return r


def copts(**kw):
    o = dict(device=-1, math_mode=0, fp_contract=0, rows_unroll=0, grad_chains=0, grad_unroll=0, factor_outputs=0, with_nuts=0)
    o.update(kw)
    return np.array([o[k] for k in ("device", "math_mode", "fp_contract", "rows_unroll", "grad_chains", "grad_unroll", "factor_outputs", "with_nuts")],
                    dtype=np.int32)


def flat_cfg(cfg, nvars=1):
    """sampler.py SamplerConfig -> (icfg, dcfg) in the shim's RH_JNI_ICFG / RH_JNI_DCFG order (what HipConfig.flatten builds)"""
    c, _keep = R.sampler.to_c_config(cfg, nvars)
    ic = [c.iterations, c.warmup, c.sampler, c.hmc_steps, c.ehmc_max_steps, c.ehmc_min_steps, c.ehmc_buf_size, c.step_tuner,
          c.mass_tuner, c.mass_init_window, c.mass_skip_first, c.mass_skip_last, c.nuts_max_depth, c.engine, c.grad_splits]
    dc = [c.ehmc_p_count, c.dualavg_delta, c.static_step, c.mass_expansion]
    return np.array(ic, dtype=np.int32), np.array(dc, dtype=np.float64)


# ---- static: the Scala side agrees with the shim ---------------------------------------------------------------------
def _macro(src, name):
    body = re.search(r"#define %s\(X\)((?:.*\\\n)*.*)\n" % name, src).group(1)
    return re.findall(r"X\((\w+)\)", body)


def test_scala_flat_array_layouts_match_the_shim():
    shim = open(os.path.join(ROOT, "rainier_amd", "jni", "rainier_hip_jni.c")).read()
    scala = open(os.path.join(ROOT, "integration", "scala", "HipModel.scala")).read()
    header = open(os.path.join(ROOT, "include", "rainier_hip.h")).read()

    def scala_array(decl):
        body = re.search(re.escape(decl) + r"\s*=\s*Array\(([^)]*)\)", scala).group(1)
        return [x.strip() for x in body.replace("\n", " ").split(",")]

    assert scala_array("val icfg: Array[Int]") == _macro(shim, "RH_JNI_ICFG")
    assert scala_array("val dcfg: Array[Double]") == _macro(shim, "RH_JNI_DCFG")
    assert scala_array("def copts: Array[Int]") == _macro(shim, "RH_JNI_COPTS")
    # ... and the shim's lists cover every field of the C structs (pointer fields travel as their own arrays)
    def struct_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [m.group(1) for m in re.finditer(r"(\w+)(?:\[\d+\])?;", body)]
    co = struct_fields("rh_compile_opts")
    assert co[0] == "struct_size" and co[-1] == "reserved" and co[1:-1] == _macro(shim, "RH_JNI_COPTS")
    cf = struct_fields("rh_config")
    scalars = [f for f in cf if f not in ("struct_size", "static_mass", "rng_next_gaussian", "reserved")]
    assert sorted(scalars) == sorted(_macro(shim, "RH_JNI_ICFG") + _macro(shim, "RH_JNI_DCFG"))
    st = struct_fields("rh_chain_stats")
    assert [f for f in st if f not in ("error", "reserved")] == _macro(shim, "RH_JNI_STATS")
    # every @native method of Native.scala has its JNI symbol in the shim, and vice versa
    native = open(os.path.join(ROOT, "integration", "scala", "Native.scala")).read()
    methods = re.findall(r"@native def (\w+)\(", native)
    symbols = re.findall(r"Java_com_stripe_rainier_hip_Native_00024_(\w+)\(", shim)
    assert sorted(methods) == sorted(symbols) and len(methods) == 12
    # argument counts agree (JNIEnv*, jobject + the Scala parameters)
    for mth in methods:
        sc = re.search(r"@native def %s\((.*?)\)\s*:" % mth, native, re.S).group(1)
        n_scala = 0 if not sc.strip() else len([a for a in sc.split(",") if ":" in a])
        cc = re.search(r"Native_00024_%s\(\s*(.*?)\)\s*\{" % mth, shim, re.S).group(1)
        n_c = len([a for a in cc.split(",") if a.strip()]) - 2
        assert n_scala == n_c, (mth, n_scala, n_c)


# ---- CPU: executed marshalling ------------------------------------------------------------------------------------------
@pytest.fixture(params=["copying", "pinning"])
def jvm(fj, request):
    """both array-access behaviours the JNI specification allows: a JVM that copies (isCopy = JNI_TRUE, Release(0) copies back,
    JNI_ABORT discards) and one that pins (the native code works on the array's own storage)"""
    fj.fj_set_pinning(1 if request.param == "pinning" else 0)
    yield fj
    fj.fj_set_pinning(0)


def test_shim_runs_without_a_jvm_diagnostics_end_to_end(jvm):
    fj = jvm
    assert call(fj, "abiVersion") == _capi.lib().rh_abi_version()
    rng = np.random.default_rng(0)
    draws = rng.normal(size=(4, 50, 3)).cumsum(axis=1) * 0.1
    jd, jr, je = JArr(fj, draws.ravel()), JArr(fj, np.zeros(3)), JArr(fj, np.zeros(3))
    call(fj, "diagnostics", jd, 4, 50, 3, jr, je)
    want = R.diagnostics(draws)
    assert np.array_equal(jr.get(), [r for r, _ in want]) and np.array_equal(je.get(), [e for _, e in want])
    assert np.array_equal(jd.get(), draws.ravel())           # the input array was released with JNI_ABORT, untouched
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*multiple chains"):
        call(fj, "diagnostics", jd, 1, 200, 3, jr, je)        # Trace.scala:12 require -> IllegalArgumentException


def test_shim_rejects_malformed_arrays_without_leaking(fj):
    spec = models.funnel(10)
    rir = JArr(fj, np.frombuffer(spec.rir, dtype=np.int8))
    cols = fj.fj_new_array(FJ_OBJECT, 0, None)
    nrows = JArr(fj, np.zeros(2, dtype=np.int64))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*copts"):
        call(fj, "modelCreate", rir, cols, nrows, JArr(fj, np.zeros(3, dtype=np.int32)))
    bad = JArr(fj, np.frombuffer(b"\0" * 64, dtype=np.int8))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*RIR"):
        call(fj, "modelCreate", bad, cols, nrows, JArr(fj, copts()))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*math_mode"):
        call(fj, "modelCreate", rir, cols, nrows, JArr(fj, copts(math_mode=7)))
    if _capi.lib().rh_device_count() == 0:
        with pytest.raises(RuntimeError, match="RuntimeException.*no CPU fallback"):
            call(fj, "modelCreate", rir, cols, nrows, JArr(fj, copts()))
    ic, dc = flat_cfg(R.HMC(5, 5, 2))
    seeds, draws, mass = JArr(fj, np.arange(2, dtype=np.int64)), JArr(fj, np.zeros(2 * 5 * 10)), JArr(fj, np.zeros(20))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*wrong length"):
        call(fj, "sample", JArr(fj, np.zeros(1, dtype=np.int64)), JArr(fj, ic[:-1]), JArr(fj, dc), None, None, seeds, draws, mass, None)
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*stats"):
        call(fj, "sample", JArr(fj, np.zeros(1, dtype=np.int64)), JArr(fj, ic), JArr(fj, dc), None, None, seeds, draws, mass, JArr(fj, np.zeros(3)))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*not a model handle"):   # a null handle never reaches the device
        call(fj, "sample", JArr(fj, np.zeros(2, dtype=np.int64)), JArr(fj, ic), JArr(fj, dc), None, None, seeds, draws, mass, None)


# ---- GPU: the shim end to end -----------------------------------------------------------------------------------------------
def _create(fj, spec, **kw):
    rir = JArr(fj, np.frombuffer(spec.rir, dtype=np.int8))
    cols = fj.fj_new_array(FJ_OBJECT, len(spec.columns), None)
    keep = [JArr(fj, c) for c in spec.columns]
    for i, c in enumerate(keep):
        fj.fj_set_object(cols, i, c.h)
    h = call(fj, "modelCreate", rir, cols, JArr(fj, np.array(spec.nrows, dtype=np.int64)), JArr(fj, copts(**kw)))
    assert h != 0 and call(fj, "modelNVars", h) == spec.n_params
    return h


def _sample(fj, handles, cfg, seeds, n, static_mass=None, nn=None):
    ic, dc = flat_cfg(cfg, n)
    chains = len(seeds)
    draws, mass, stats = JArr(fj, np.zeros(chains * cfg.iterations * n)), JArr(fj, np.zeros(chains * n)), JArr(fj, np.zeros(chains * 7))
    call(fj, "sample", JArr(fj, np.array(handles, dtype=np.int64)), JArr(fj, ic), JArr(fj, dc),
         None if static_mass is None else JArr(fj, static_mass), None if nn is None else JArr(fj, nn),
         JArr(fj, np.array(seeds, dtype=np.int64)), draws, mass, stats)
    return draws.get().reshape(chains, cfg.iterations, n), mass.get().reshape(chains, n), stats.get().reshape(chains, 7)


@pytest.mark.gpu
def test_shim_sample_is_bit_identical_to_the_ctypes_path(jvm):
    fj = jvm
    spec = models.eight_schools()
    seeds = [11, 12, 13, 14, 15]
    for cfg, kw in [(R.make_config(20, 60), dict(math_mode=1)),                                        # DefaultConfig: EHMC + DualAvg + diag mass
                    (R.make_config(15, 40, R.NUTSSampler(6)), dict(math_mode=1, with_nuts=1)),         # NUTS reachable through the glue
                    (R.HMC(30, 10, 7), dict(fp_contract=1, factor_outputs=1))]:                        # the fast build
        h = _create(fj, spec, **kw)
        d, m, s = _sample(fj, [h], cfg, seeds, spec.n_params)
        # the ORACLE first (oracle/sampler.c: Driver.sample restated), so that a wrong-but-consistent engine cannot pass: strict
        # builds bit for bit -- draws, mass matrix, step size, leapfrog count --; the fast build (FMA, fast log) on its first
        # iterations, before rounding differences have been amplified by the trajectories
        for c in (0, len(seeds) - 1):
            want, wmass, wst = O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), seeds[c])
            if kw.get("math_mode") == 1:
                assert np.array_equal(d[c], want) and np.array_equal(m[c], wmass), ("shim vs oracle", kw, c)
                assert s[c, 0] == wst.leapfrog_steps and s[c, 5] == wst.step_size
            else:
                # (measured: 1e-4 absolute after the 30 adaptation iterations -- last-place differences of the fast build's FMAs and
                #  log, amplified by the trajectories; a wrong engine is off by O(1))
                np.testing.assert_allclose(d[c][:3], want[:3], rtol=2e-2, atol=2e-3, err_msg="shim (fast build) vs oracle")
        ref = R.Model(spec, device=0, math_mode=kw.get("math_mode", 0), fp_contract=bool(kw.get("fp_contract")),
                      factor_outputs=bool(kw.get("factor_outputs"))).sample(cfg, seeds=seeds)
        assert np.array_equal(d, ref.chains) and np.array_equal(m, ref.mass)
        assert np.array_equal(s[:, 0], [x.leapfrogSteps for x in ref.stats]) and np.array_equal(s[:, 5], [x.stepSize for x in ref.stats])
        assert np.array_equal(s[:, 4], [x.meanAcceptProb for x in ref.stats])
        # two handles = rh_sample_multi: same trace
        h2 = call(fj, "modelClone", h, -1)                 # what hipSample(devices = ...) does: lowered once, cloned per device
        assert h2 != 0 and h2 != h and call(fj, "modelNVars", h2) == spec.n_params
        eng = call(fj, "modelEngines", h)                  # rh_model_engines: a light model has every engine it could have
        assert eng & 4 and eng & (1 | 2) and (eng >> 8) >= 1
        d2, m2, _ = _sample(fj, [h, h2], cfg, seeds, spec.n_params)
        assert np.array_equal(d2, d) and np.array_equal(m2, m)
        call(fj, "modelDestroy", h); call(fj, "modelDestroy", h2)
    # StaticMassMatrix + StaticStepSize through the nullable arrays; a zero mass element is the reference's require failure
    h = _create(fj, spec, math_mode=1)
    sm = np.linspace(0.5, 2.0, 10)
    cfg = R.make_config(10, 10, R.HMCSampler(3), R.StaticStepSize(0.05), R.StaticMassMatrix(R.DiagonalMassMatrix(sm)))
    d, m, _ = _sample(fj, [h], cfg, seeds, 10, static_mass=sm)
    want, _, _ = O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), seeds[0])
    assert np.array_equal(d[0], want), "shim (static mass / static step) vs oracle"
    ref = R.Model(spec, device=0, math_mode=1).sample(cfg, seeds=seeds)
    assert np.array_equal(d, ref.chains) and np.array_equal(m, np.tile(sm, (5, 1)))
    # a shared java.util.Random stream: seeds = state ^ multiplier, pending nextNextGaussian per chain
    nn = np.array([0.25, np.nan, -1.5, np.nan, 0.0])
    states = [(s * 2654435761) & ((1 << 48) - 1) for s in seeds]
    d, _, _ = _sample(fj, [h], R.HMC(10, 5, 2), [st ^ 0x5DEECE66D for st in states], 10, nn=nn)
    ref = R.Model(spec, device=0, math_mode=1).sample(R.HMC(10, 5, 2), rng_states=[(st, None if np.isnan(g) else g) for st, g in zip(states, nn)])
    assert np.array_equal(d, ref.chains)
    # wrongly sized result arrays become exceptions before anything is written (they were native heap overflows)
    ic, dc = flat_cfg(R.HMC(10, 5, 2), 10)
    args = lambda draws, mass: (JArr(fj, np.array([h], dtype=np.int64)), JArr(fj, ic), JArr(fj, dc), None, None,
                                JArr(fj, np.array(seeds, dtype=np.int64)), JArr(fj, np.zeros(draws)), JArr(fj, np.zeros(mass)), None)
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*draws must hold"):
        call(fj, "sample", *args(5 * 5 * 10 - 1, 50))
    with pytest.raises(RuntimeError, match="IllegalArgumentException.*mass must hold"):
        call(fj, "sample", *args(5 * 5 * 10, 49))
    call(fj, "modelDestroy", h)


@pytest.mark.gpu
def test_shim_density_optimize_and_requirements(fj):
    spec = models.linreg(n=70000, k=3)
    h = _create(fj, spec, fp_contract=1, factor_outputs=1, grad_chains=8)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True, grad_chains=8)
    q = np.random.default_rng(1).normal(size=(9, 5)) * 0.4
    od = O.OracleDensity(spec)
    refs = [od.update_both(qq) for qq in q]                 # the oracle: (logp, gradient) and sum|term| per output
    for engine in (_capi.ENGINE_AUTO, _capi.ENGINE_CHAIN, _capi.ENGINE_TICK):
        jl, jg = JArr(fj, np.zeros(9)), JArr(fj, np.zeros(45))
        call(fj, "densityEval", h, JArr(fj, q.ravel()), 9, engine, 0, jl, jg)
        for c, (ref, ab) in enumerate(refs):                # the shim against the ORACLE first
            got = np.concatenate([[jl.get()[c]], jg.get().reshape(9, 5)[c]])
            assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300), ("shim densityEval vs oracle", engine, c)
        lp, g = m.density_batch(q, engine=engine)
        assert np.array_equal(jl.get(), lp) and np.array_equal(jg.get().reshape(9, 5), g)
    with pytest.raises(RuntimeError, match="IllegalArgumentException"):
        call(fj, "densityEval", h, JArr(fj, q.ravel()), 9, 9, 0, JArr(fj, np.zeros(9)), JArr(fj, np.zeros(45)))
    x, ev, st = JArr(fj, np.zeros(10)), JArr(fj, np.zeros(2, dtype=np.int32)), JArr(fj, np.zeros(2, dtype=np.int32))
    x0 = np.concatenate([np.zeros(5), np.full(5, 0.1)])
    call(fj, "optimize", h, JArr(fj, x0), 2, 0, x, ev, st)
    ox, oev = O.optimize_model(spec, x0[:5])                # the oracle's L-BFGS (optimizer/LBFGS.java restated) from the first start
    assert oev > 0
    np.testing.assert_allclose(x.get().reshape(2, 5)[0], ox, rtol=1e-6, atol=1e-8, err_msg="shim optimize vs the oracle's L-BFGS")
    xr, er, sr = m.optimize(x0.reshape(2, 5))
    assert np.array_equal(x.get().reshape(2, 5), xr) and np.array_equal(ev.get(), er) and np.array_equal(st.get(), sr)
    call(fj, "modelDestroy", h)
    rir, nreq = models.funnel_predict(10)
    dr = np.random.default_rng(2).normal(size=(33, 10))
    out = JArr(fj, np.zeros(33 * nreq))
    call(fj, "requirementsEval", JArr(fj, np.frombuffer(rir, dtype=np.int8)), JArr(fj, copts(math_mode=1)), JArr(fj, dr.ravel()), 33, out)
    orq = O.OracleDensity(models.ModelSpec("req", rir, [], [0] * nreq, 10), O.JM_DET)
    assert np.array_equal(out.get().reshape(33, nreq), np.array([orq.requirements(qq, nreq) for qq in dr])), "shim requirementsEval vs oracle"
    assert np.array_equal(out.get().reshape(33, nreq), R.predict(rir, dr, nreq, math_mode=_capi.MATH_STRICT))
    with pytest.raises(RuntimeError, match="RuntimeException|IllegalArgumentException"):
        call(fj, "requirementsEval", JArr(fj, np.frombuffer(models.funnel().rir, dtype=np.int8)), JArr(fj, copts()), JArr(fj, dr.ravel()), 33, out)
