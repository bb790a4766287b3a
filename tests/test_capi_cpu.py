"""CPU-side checks of the C-ABI library: it loads, exports what include/rainier_hip.h declares, lowers every
model to a gfx950 code object, rejects malformed input -- and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rainier_amd import _capi, models
from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as G
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rainier_amd", "csrc")])
    return _capi.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "rainier_hip.h")).read()
    declared = set(re.findall(r"\b(rh_[a-z_]+)\s*\(", hdr))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rh_abi_version() == 6


def test_struct_layouts_match_header(lib):
    c = _capi.Config()
    lib.rh_config_default(C.byref(c))
    assert c.struct_size == C.sizeof(_capi.Config)
    # DefaultConfig (sampler/Sampler.scala:17-27)
    assert (c.iterations, c.warmup, c.sampler, c.ehmc_max_steps, c.ehmc_min_steps, c.ehmc_buf_size) == (1000, 1000, 1, 1024, 1, 100)
    assert (c.ehmc_p_count, c.dualavg_delta, c.step_tuner, c.mass_tuner) == (0.1, 0.8, 0, 1)
    assert (c.mass_init_window, c.mass_expansion, c.mass_skip_first, c.mass_skip_last) == (50, 1.5, 50, 50)


@pytest.mark.parametrize("builder", [models.normal_1d, models.funnel, models.eight_schools,
                                     lambda: models.linreg(n=8), lambda: models.logistic(n=8, k=5)])
def test_lowering_cross_compiles_for_gfx950(lib, builder):
    spec = builder()
    for mode in (_capi.MATH_FAST, _capi.MATH_STRICT):
        src, size = _capi.lower_only(spec.rir, _capi.compile_opts(math_mode=mode))
        assert size > 1000
        assert "#define RH_NVARS %d" % spec.n_params in src
        assert src.count("template <> struct rh_target<") == len(spec.nrows)
        for kernel in ("rh_chain_kernel", "rh_density_kernel", "rh_selftest_kernel"):
            assert kernel in src


def test_invariants_are_hoisted_out_of_the_row_loop(lib):
    src, _ = _capi.lower_only(models.linreg(n=8).rir)
    tgt = src[src.index("template <> struct rh_target<1>"):]
    inv, row = tgt[:tgt.index("static RH_DEV void row")], tgt[tgt.index("static RH_DEV void row"):tgt.index("};")]
    assert "RH_EXP" in inv and "RH_EXP" not in row          # exp(-2s) evaluated once per gradient, not per row
    assert "c[0]" in row and "c[3]" in row


def test_malformed_rir_is_rejected(lib):
    good = models.funnel().rir
    for bad, what in [(b"", "short"), (good[:-4], "truncated"), (b"XXXX" + good[4:], "magic"), (good + b"\0\0\0\0", "trailing")]:
        with pytest.raises(_capi.RainierHipError) as e:
            _capi.lower_only(bad)
        assert e.value.code == _capi.RH_E_INVALID, what
    # forward reference
    import struct
    blob = struct.pack("<6I", 0x31524952, 1, 1, 1, 1, 0) + struct.pack("<4I", 0, 0, 0, 0) + struct.pack("<3I", 2, 0, 5)
    with pytest.raises(_capi.RainierHipError):
        _capi.lower_only(blob)


def test_no_cpu_fallback(lib):
    import rainier_amd as R
    if lib.rh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(R.RainierHipError) as e:
        R.Model(models.funnel())
    assert e.value.code == _capi.RH_E_DEVICE and "no CPU fallback" in str(e.value)


def test_diagnostics_match_oracle(lib):
    import rainier_amd as R
    rng = np.random.default_rng(2)
    m, n, k = 5, 300, 3
    x = np.zeros((m, n, k))
    for i in range(1, n):
        x[:, i, :] = np.array([0.9, 0.5, 0.0]) * x[:, i - 1, :] + rng.normal(size=(m, k))
    got = R.diagnostics(x)
    for p in range(k):
        rhat, ess = O.diagnostics(x[:, :, p])
        assert got[p][0] == pytest.approx(rhat, rel=1e-13) and got[p][1] == pytest.approx(ess, rel=1e-12)
    with pytest.raises(ValueError):
        R.diagnostics(x[:1])


def test_requirements_program_lowers(lib, monkeypatch):
    import ctypes as C
    rir, nreq = models.funnel_predict(10)
    monkeypatch.setenv("RH_LOWER_ONLY", "1")
    draws = np.zeros((2, 10)); out = np.zeros((2, nreq))
    blob = C.create_string_buffer(rir, len(rir))
    rc = lib.rh_requirements_eval(blob, len(rir), None, _capi.dptr(draws), 2, _capi.dptr(out))
    assert rc == (_capi.RH_OK if lib.rh_device_count() == 0 else rc)
    # a density program is rejected by the predict entry point and vice versa
    bad = C.create_string_buffer(models.funnel().rir, len(models.funnel().rir))
    assert lib.rh_requirements_eval(bad, len(models.funnel().rir), None, _capi.dptr(draws), 2, _capi.dptr(out)) == _capi.RH_E_INVALID
    with pytest.raises(_capi.RainierHipError):
        _capi.lower_only(rir)


def test_headers_are_c_and_library_links_from_c(tmp_path):
    """include/*.h as C99 (-pedantic -Werror), linked against librainier_hip.so from plain C; without a device the model
    cannot be created (RH_E_DEVICE), i.e. no CPU fallback hides behind the ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "stubs", "abi_check.c"), "-o", exe, "-L", os.path.join(root, "rainier_amd"),
                           "-lrainier_hip", "-Wl,-rpath," + os.path.join(root, "rainier_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe]).decode()
    assert "abi 6 header 6" in out
    assert "sizeof rh_config %d rh_chain_stats %d" % (C.sizeof(_capi.Config), C.sizeof(_capi.ChainStats)) in out
    assert "default 1000 1000 sampler 1 ehmc 1024 mass 1 50 1.5" in out          # DefaultConfig, sampler/Sampler.scala:17-27
    # (no `import torch` in this process: torch brings its own bundled comgr / LLVM under the system library's soname, and every model
    #  this process compiled afterwards would be built by that other compiler and land in the shared kernel cache)
    if not os.path.exists("/dev/kfd"):
        assert "create rc %d devices" % _capi.RH_E_DEVICE in out and "no CPU fallback" in out


def test_jni_shim_typechecks():
    """rainier_amd/jni/rainier_hip_jni.c against a declaration-only jni.h (tests/stubs): the shim that a Rainier maintainer
    builds next to the library at least parses and type-checks against the current C ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "tests", "stubs"),
                           "-I", os.path.join(root, "include"), os.path.join(root, "rainier_amd", "jni", "rainier_hip_jni.c")])


def test_rir_parser_survives_mutated_blobs():
    """Bit flips, truncations, random words and trailing bytes on valid programs: parse_rir + simplify must either accept
    the blob or reject it with a status -- never crash (the JVM side hands this library bytes from another process space)."""
    import random
    seeds = [models.funnel().rir, models.eight_schools().rir, models.linreg(n=4).rir, models.logistic(n=4, k=8).rir,
             models.hier_negbin(70, 2).rir]
    rng = random.Random(2026)
    accepted = rejected = 0
    for _ in range(2500):
        b = bytearray(rng.choice(seeds))
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            del b[rng.randrange(len(b)):]
        elif kind == 2:
            i = rng.randrange(0, len(b) - 4, 4); b[i:i + 4] = rng.randrange(2 ** 32).to_bytes(4, "little")
        elif kind == 3:
            i = rng.randrange(0, len(b) - 4, 4); b[i:i + 4] = rng.choice([0, 1, 0xffffffff, 0x7fffffff, 18, 19, 20]).to_bytes(4, "little")
        else:
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        try:
            _capi.simplify_rir(bytes(b)); accepted += 1
        except _capi.RainierHipError as e:
            assert e.code == _capi.RH_E_INVALID
            rejected += 1
    assert rejected > 1500 and accepted > 50


def test_rir_header_overflow_is_rejected_before_allocation():
    """n_params is bounded against the blob before anything is sized from it -- 0xFFFFFFFF (n_params + 1 wraps to 0),
    2^31 (gigabytes) and "large but plausible" values all come back as RH_E_INVALID, quickly, from every entry point."""
    import struct, time
    good = models.funnel().rir
    for n_params in (0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 1 << 24, (1 << 24) + 1, 100000):
        b = bytearray(good)
        b[8:12] = struct.pack("<I", n_params)
        t0 = time.time()
        for fn in (lambda: _capi.simplify_rir(bytes(b)), lambda: _capi.lower_only(bytes(b))):
            with pytest.raises(_capi.RainierHipError) as ei:
                fn()
            assert ei.value.code == _capi.RH_E_INVALID
        assert time.time() - t0 < 2.0


def test_committed_bench_line_honours_the_contract():
    """profiles/r3_b_cfg2/bench_default.json is the output of `python bench.py` on an MI355X: one JSON line with the keys the
    driver and the judge read (metric/value/...; roofline{bound, achieved, peak, unit, frac, traffic}; cpu_baseline{...}), the two
    ESS legs with their R-hat, and the inlined GPU figure beside the inlined CPU one."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = [l for l in open(os.path.join(root, "profiles", "r3_b_cfg2", "bench_default.json")).read().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("leapfrog steps/sec") and d["unit"] == "leapfrog steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "cfg2" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "fp64_valu") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["inlined_sufficient_statistics"]["value"] > 0 and c["nproc"] >= c["cores"]
    assert d["ess_leg"]["warmup"] >= 128 and d["ess_leg"]["iterations"] >= 64 and 0.6 < d["ess_leg"]["mean_accept_prob"] < 0.99
    assert abs(d["value"] - d["config"]["chains"] * d["config"]["leapfrog_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["ess_leg"]["converged"] and d["ess_leg"]["rhat_max"] < 1.01       # the leg ess_per_s quotes is a converged run
    assert r["traffic_source"].startswith("measured in this run") and 0.5 * 32e6 < r["traffic"] < 2 * 32e6   # the 32 MB data set, about once
    for leg in ("ess_leg", "ess_leg_identity_mass"):
        e = d[leg]
        assert len(e["rhat"]) == len(e["ess"]) == 5 and e["rhat_max"] == max(e["rhat"]) and e["ess_min"] == min(e["ess"])
        assert e["iterations"] >= 1024 and abs(e["ess_per_s"] - e["ess_min"] / e["seconds"]) < 1e-6 * e["ess_per_s"]
    assert d["ess_per_s"] == d["ess_leg"]["ess_per_s"]
    g = d["gpu_inlined"]
    assert g["chains_32768"] > g["chains_1024"] > 1e8


def test_logit_family_links_are_lowered_in_closed_form_only_when_verified():
    """emit.cpp detect_logit / detect_link: fast builds of cfg 4 (Bernoulli-logit, MFMA GLM kernel), cfg 5 (negative binomial in
    gather mode) and a plain negative-binomial GLM call rh_logit_link once per evaluation; strict builds and RH_LOGIT_LINK=0
    keep the literal lowering; a model whose adjoint does NOT match the closed form (a gradient output scaled by hand) is left
    alone -- the rewrite is verified numerically, never assumed."""
    fast = _capi.compile_opts(fp_contract=True, factor_outputs=True)
    gen = lambda src: src.split("// ---- generated from RIR")[1].split("// rh_engine.hip.h")[0]   # the per-model part only
    for spec in (models.logistic(n=8, k=50), models.hier_negbin(200, 2), models.negbin_glm(n=8, k=3)):
        src, _ = _capi.lower_only(spec.rir, fast)
        assert "rh_logit_link(" in gen(src), spec.name
        src, _ = _capi.lower_only(spec.rir, _capi.compile_opts(math_mode=_capi.MATH_STRICT))
        assert "rh_logit_link(" not in gen(src), spec.name
    os.environ["RH_LOGIT_LINK"] = "0"
    try:
        src, _ = _capi.lower_only(models.hier_negbin(200, 2).rir, fast)
        assert "rh_logit_link(" not in gen(src)
    finally:
        del os.environ["RH_LOGIT_LINK"]
    # a wrong adjoint: the same value, but "d/d a" replaced by its square -> that core is not kappa * g -> literal lowering
    # (a constant factor would not do: output factoring peels it off and the row-level core still verifies, correctly)
    from rainier_amd.frontend import Graph
    import numpy as np
    g = Graph(2, [0, 3])
    a, b = g.param(0), g.param(1)
    eta = a + b * g.col(1, 2)
    p = 1.0 / ((eta * -1.0).exp() * 5.0 + 1.0)
    row = g.col(1, 1) + (1.0 - p).log() * 5.0 + g.col(1, 0) * p.log()
    good = g.compile([a * a * -0.5, row])
    src, _ = _capi.lower_only(good, fast)
    assert "rh_logit_link(" in gen(src)
    gr = g.gradient(row)
    bad = g.compile([a * a * -0.5, row], gradients=[g.gradient(a * a * -0.5), [gr[0] * gr[0], gr[1]]])
    src, _ = _capi.lower_only(bad, fast)
    assert "rh_logit_link(" not in gen(src)


def test_lowering_is_thread_safe():
    """`rh_model_create`'s host side (parser, data-dependent passes, emitter) from 6 threads at once: the same source as serially
    (the JVM calls it from whatever thread builds a model; nothing but the error slot and the kernel cache is shared)"""
    import hashlib
    import json
    import threading
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    load = lambda n: json.load(open(os.path.join(G, n)))
    specs = [models.eight_schools_reference(), models.ark_reference(load("ark.json")), models.glmm_poisson2_reference(100, 40, load("glmm_poisson2.json")),
             models.lowdim_gaussmix_reference(load("lowdim_gaussmix.json")), models.linreg(n=20000, k=3), models.hier_negbin(40, 6, seed=2)]
    opts = [dict(fp_contract=True, factor_outputs=True), dict(math_mode=_capi.MATH_STRICT)]

    def run(s, o):
        kw = dict(columns=s.columns, nrows=s.nrows) if s.columns else {}
        return hashlib.sha256(_capi.lower_only(s.rir, _capi.compile_opts(**o), compile=False, **kw)[0].encode()).hexdigest()

    serial = {(i, j): run(s, o) for i, s in enumerate(specs) for j, o in enumerate(opts)}
    wrong = []

    def worker(k):
        try:
            for i in np.random.default_rng(k).permutation(len(specs)):
                for j, o in enumerate(opts):
                    if run(specs[i], o) != serial[(i, j)]:
                        wrong.append((k, i, j))
        except Exception as e:  # noqa: BLE001
            wrong.append((k, repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not wrong, wrong[:3]


def test_lower_only_without_data_keeps_working_for_gather_shaped_models(lib):
    """rh_lower_only has no columns: the loader's preparation of a parameter table for gather mode (lift.cpp: it synthesises
    columns next to the caller's) waits for rh_model_create, and the model lowers on the generic path as it arrives -- a
    latentVec(K >= 65) hierarchical model from the reference's front end (it used to fail with 'a column pointer is
    NULL'); with the data in hand the same program reaches gather mode"""
    from rainier_amd import compute as CC, modeling as M
    rng = np.random.default_rng(4)
    K, n = 100, 400
    a = M.Normal(0, 1).latent; b = M.Normal(0, 1).latent; tau = M.Exponential(1).latent
    zs = M.Normal(0, 1).latentVec(K)
    site = rng.integers(0, K, n).astype(float); x = rng.normal(size=n); ys = rng.poisson(3.0, n).astype(float)
    fn = lambda s, u: M.NegativeBinomial((a + tau * CC.Lookup.apply(s, zs) + b * u).logistic, 5.0)
    spec = M.Model.observe_vec(ys, [site, x], fn, split=False).compile("raw_table_lower_only", inline=False)
    fast = _capi.compile_opts(fp_contract=True, factor_outputs=True)
    for opts in (fast, _capi.compile_opts(math_mode=_capi.MATH_STRICT)):
        src, _ = _capi.lower_only(spec.rir, opts, compile=False)
        assert "rh_chain_kernel" in src and "#define RH_HAS_GATHER 1\n" not in src
    src, _ = _capi.lower_only(spec.rir, fast, compile=False, columns=spec.columns, nrows=spec.nrows)
    assert "#define RH_HAS_GATHER 1\n" in src


def test_a_row_target_behind_more_than_255_data_free_targets(lib):
    """Node::dep names a target by index + 1: with an 8-bit field a row target at index >= 255 wrapped to 'parameters only' and
    the parser's cross-target checks went wrong modulo 256.  299 data-free targets followed by one row target parse,
    merge (runs of data-free targets become one) and lower; a row target that reads another target's column is still refused."""
    from rainier_amd.frontend import Graph
    ntar = 300
    g = Graph(1, [0] * (ntar - 1) + [2])
    a = g.param(0)
    terms = [a * a * (-0.5 / ntar) for _ in range(ntar - 1)] + [(g.col(ntar - 1, 0) - a * g.col(ntar - 1, 1)) * a]
    rir = g.compile(terms)
    src, _ = _capi.lower_only(rir, _capi.compile_opts(math_mode=_capi.MATH_STRICT), compile=False)
    assert "#define RH_NROWTARGETS 1\n" in src
    _capi.simplify_rir(rir)                                         # parses, cleans up and serialises again
    # ... and the check that a target only reaches its OWN columns still bites at index 299 (it compared indices modulo 256)
    g = Graph(1, [2] + [0] * (ntar - 2) + [2])
    a = g.param(0)
    bad = g.compile([g.col(0, 0) * a] + [a * a * -0.5 for _ in range(ntar - 2)] + [g.col(0, 1) * a + g.col(ntar - 1, 0)])
    with pytest.raises(_capi.RainierHipError, match="two targets|another target"):
        _capi.lower_only(bad, compile=False)


def _kernel_meta(code: bytes, kernel: str, key: str) -> int:
    """one integer field of a kernel's metadata in a gfx950 code object (msgpack note; a kernel's keys are in alphabetical order)"""
    def mstr(v):
        v = v.encode()
        return (bytes([0xa0 | len(v)]) if len(v) < 32 else bytes([0xd9, len(v)])) + v
    at = code.find(mstr(".name") + mstr(kernel))
    assert at >= 0, kernel
    # .name sorts before .private_segment_fixed_size / .sgpr_* / .vgpr_*; keys that sort before it belong to the NEXT kernel's map
    k = code.find(mstr(key), at)
    assert k >= 0, key
    p = code[k + len(mstr(key)):]
    if p[0] <= 0x7f:
        return p[0]
    return {0xcc: lambda: p[1], 0xcd: lambda: (p[1] << 8) | p[2], 0xce: lambda: int.from_bytes(p[1:5], "big")}[p[0]]()

def _lower_bench_model_in_a_fresh_process(cache_dir) -> bytes:
    """the cfg-2 build's code object, compiled by a python process that has loaded nothing but the engine: a process that imported
    torch first compiles with torch's BUNDLED comgr / LLVM (another ROCm version, loaded under the same soname) and gets different
    code -- the in-tree cache is filled by build(), which runs without torch"""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from rainier_amd import _capi, models\n"
            "src, size = _capi.lower_only(models.linreg(n=8, k=3).rir, _capi.compile_opts(grad_chains=8, fp_contract=True, factor_outputs=True))\n"
            "assert '#define RH_GRAD_U 8\\n' in src and '#define RH_GRAD_PIPELINE 2\\n' in src and 'torch' not in sys.modules\n") % root
    subprocess.check_call([sys.executable, "-c", prog], env=dict(os.environ, RH_KERNEL_CACHE=str(cache_dir)))
    files = glob.glob(os.path.join(str(cache_dir), "*.hsaco"))
    assert len(files) == 1
    return files[0]


def test_the_bench_models_gradient_kernels_keep_their_register_budget(tmp_path, monkeypatch):
    """a guard on what the compiler makes of the cfg-2 build: both batched gradient kernels fit two wavefronts per SIMD with
    room to spare (<= 200 of 256 VGPRs), spill nothing, use no scratch -- a toolchain or source change that breaks this costs the
    headline number before any test notices"""
    code = open(_lower_bench_model_in_a_fresh_process(tmp_path), "rb").read()
    for kernel in ("rh_grad_kernel", "rh_grad_fused_kernel"):
        assert _kernel_meta(code, kernel, ".vgpr_spill_count") == 0 and _kernel_meta(code, kernel, ".sgpr_spill_count") == 0
        assert _kernel_meta(code, kernel, ".private_segment_fixed_size") == 0
        assert 128 < _kernel_meta(code, kernel, ".vgpr_count") <= 200
    assert _kernel_meta(code, "rh_absorb_kernel", ".vgpr_count") <= 32


def test_the_bench_models_row_loop_is_nine_fp64_instructions_per_evaluation(tmp_path, monkeypatch):
    """the hot loop of cfg 2 at ISA level: per chunk of 8 tiles x 8 chains, 9 fp64 instructions per 64 row-chain
    evaluations (3 fma for the predictor, the residual, 5 accumulations) and 32 global -- not flat -- loads, no spill traffic"""
    import glob
    import re
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    dis = subprocess.check_output([objdump, "-d", "--mcpu=gfx950", _lower_bench_model_in_a_fresh_process(tmp_path)]).decode()
    body = dis[dis.index("<rh_grad_fused_kernel>:"):]
    body = body[:body.index("\n\n", 10)] if "\n\n" in body[10:] else body
    ins = []
    for line in body.splitlines():
        m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        tok = args.split()[-1] if args.split() else ""
        if not tok.isdigit():
            continue
        off = int(tok) - (65536 if int(tok) >= 32768 else 0)
        tgt = a + 4 + 4 * off
        if tgt < a and tgt in addr:
            loops.append([o for _, o, _ in ins[addr[tgt]:i + 1]])
    fp64 = lambda ops: sum(o.startswith(("v_fma_f64", "v_fmac_f64", "v_add_f64", "v_mul_f64")) for o in ops)   # noqa: E731
    hot = [ops for ops in loops if fp64(ops) == 8 * 8 * 9]                    # K x U x 9: the chunk loop (innermost: nothing nests in it)
    assert hot, sorted(fp64(ops) for ops in loops)
    ops = min(hot, key=len)
    assert sum(o == "global_load_dwordx2" for o in ops) == 8 * 4 and not any(o.startswith(("flat_load", "scratch_")) for o in ops)
    assert sum(o.startswith("v_accvgpr") for o in ops) == 0 and len(ops) <= 1.15 * 8 * 8 * 9   # <= 15 % of the loop is not arithmetic
