"""What the engine reads out of a code object before it launches one of its kernels (csrc/isacheck.cpp, csrc/engine.cpp
kernel_health) -- no GPU needed.

The reference never meets this problem: its back end cuts every expression tree into methods of at most 200 nodes
(rainier-compute/.../ir/Packer.scala:10-71, compute/Compiler.scala:33-35), so no model is too heavy for the JVM.  Here a heavy row
function meets a register file, and this toolchain's register allocator can place spill code, AGPR copies and live-range copies
AHEAD of a join block's exec restore (profiles/r4_spill_rootcause): silent wrong sums.  The engine therefore launches a kernel only
if it has no spilled vector register and its machine code shows no such join block."""
import glob
import lzma
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from rainier_amd import _capi, models

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KCACHE = os.path.join(ROOT, "rainier_amd", "kcache")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _cache_objects():
    files = sorted(glob.glob(os.path.join(KCACHE, "*.hsaco")))
    if not files:
        pytest.skip("the kernel cache is empty: run __graft_entry__.build() first")
    return files


def _fault_fixture():
    return lzma.decompress(open(os.path.join(HERE, "golden", "r4_join_fault.hsaco.xz"), "rb").read())


def test_the_code_object_that_returned_wrong_draws_is_rejected(tmp_path):
    # rh_chain_kernel of hier_negbin(6, 7) as round 3 ran it on the driver's box: 58 spilled VGPRs, chain-state vectors saved with
    # 6 of 64 lanes active at the exit of the 6-row target's loop
    code = _fault_fixture()
    rep = _capi.code_object_report(code)
    ck = rep[("object", "rh_chain_kernel")]
    assert ck["fit"] == 0 and ck["vgpr_spills"] == 58 and ck["scratch"] == 236
    assert rep[("object", "rh_density_kernel")]["fit"] == 1 and rep[("object", "rh_grad_kernel")]["fit"] == 1
    # ... and by the machine-code walk alone, whatever the metadata says (the same fault shows with zero spills as v_accvgpr_write)
    import isa_check
    f = tmp_path / "fault.hsaco"
    f.write_bytes(code)
    bad = isa_check.check_file(str(f))
    assert [(k, r) for k, _, _, r in bad] == [("rh_chain_kernel", "s_or_b64 exec, exec, s[12:13]")]
    assert bad[0][2].startswith("scratch_store_dwordx2 off, v[136:137]")


def test_instruction_walk_agrees_with_llvm_objdump_on_every_cached_kernel():
    # the engine's own decoder only needs instruction LENGTHS and a handful of opcodes; every instruction boundary of every kernel
    # build() left in the cache must be where llvm-objdump puts it
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    files = _cache_objects()
    step = max(1, len(files) // 24)            # ~24 code objects spread over the cache (each one: 6..10 kernels)
    nk = 0
    for f in files[::step]:
        code = open(f, "rb").read()
        out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], capture_output=True, text=True, check=True).stdout
        cur, addrs, base = None, {}, {}
        for ln in out.splitlines():
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", ln)
            if m:
                cur = m.group(2); addrs[cur] = []; base[cur] = int(m.group(1), 16); continue
            m = re.search(r"// ([0-9A-F]{12}):", ln)
            if m and cur:
                addrs[cur].append(int(m.group(1), 16) - base[cur])
        for k, a in addrs.items():
            mine = _capi.code_object_offsets(code, k)
            assert mine and mine == a[:len(mine)], (os.path.basename(f), k)     # (objdump also prints the padding behind the symbol)
            nk += 1
    assert nk >= 50


def test_engine_and_objdump_rule_agree_on_every_cached_kernel():
    # two statements of the join-block rule: the engine's walk of the machine code and tools/isa_check.py over objdump's text
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    import isa_check
    files = _cache_objects()
    step = max(1, len(files) // 30)
    for f in files[::step]:
        code = open(f, "rb").read()
        rep = _capi.code_object_report(code)
        flagged_py = {k for k, _, _, _ in isa_check.check_file(f)}
        flagged_engine = {k for (_, k), v in rep.items() if not v["fit"] and "exec restore" in v["why"]}
        spilled = {k for (_, k), v in rep.items() if v["vgpr_spills"] > 0}
        assert flagged_engine == flagged_py - spilled or flagged_engine | spilled >= flagged_py, (os.path.basename(f), flagged_engine, flagged_py)


def test_no_cached_code_object_is_an_abandoned_attempt():
    # an attempt the engine abandoned (it lowered the model again with a lighter shape) leaves a small marker, not a code object: what
    # ships in the cache is what can be launched, and the next process takes the same decision without compiling
    files = _cache_objects()
    markers = glob.glob(os.path.join(KCACHE, "*.unfit"))
    stems = {os.path.basename(f)[:-6] for f in files}
    assert not stems & {os.path.basename(m)[:-6] for m in markers}
    for m in markers:
        head, _, rest = open(m).read().partition("\n")
        assert head.startswith("rules=") and "allow_unhealthy=0" in head, (m, head)   # the verdict's rules version (csrc/engine.cpp marker_header)
        names = rest.split()
        assert names and all(n.startswith("rh_") for n in names), m


def test_exec_restores_the_rule_cannot_classify_stay_within_the_census():
    # The join-block rule flags an exec restore behind vector instructions only when the block is PROVEN to be the join block of the
    # region the restore closes; a restore it cannot classify (mostly a then / else arm that ends in the restore) is counted per kernel
    # (`unproven` of rh_code_object_report).  tests/golden/unproven_census.json holds, per kernel name, the most such blocks any code
    # object of the cache shows: a kernel that exceeds it -- new generated control flow, a new compiler -- fails here and is looked at
    # (profiles/r5_parity has the disassembly of the ones counted today) before the ceiling moves.
    import json
    census = json.load(open(os.path.join(HERE, "golden", "unproven_census.json")))
    seen = 0
    for f in _cache_objects():
        for (_, k), v in _capi.code_object_report(open(f, "rb").read()).items():
            assert k in census, (os.path.basename(f), k, "a kernel the census does not know")
            assert v["unproven"] <= census[k]["unproven_max"], (os.path.basename(f), k, v["unproven"], census[k])
            seen += 1
    assert seen >= 500


def test_copies_ahead_of_an_exec_restore_in_row_streaming_kernels_carry_the_arm_s_own_value():
    """VERDICT r5 next #6c.  Among the blocks the join-block rule cannot classify, the ones that END in a register / AGPR copy right ahead
    of the exec restore are textually what the allocator's fault would leave too.  What tells them apart: the fault copies a value that is
    live ACROSS the region (defined before it); an arm copies the value IT computed.  tools/unproven_census.py's witness -- every source
    of the trailing copy run is written earlier in the same block by a non-copy instruction -- must hold for every such block of the
    row-streaming kernels the engine launches, over the whole cache (the shapes that returned wrong sums in round 3; the sampler kernels'
    multi-block arms are outside a single-block witness and stay with the on-device bit comparisons)."""
    import sys
    from concurrent.futures import ProcessPoolExecutor
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import unproven_census as U
    files = _cache_objects()
    with ProcessPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        rows = [r for rs in ex.map(U.one, files, chunksize=8) for r in rs]
    row_kernels = ("rh_grad_kernel", "rh_grad_fused_kernel", "rh_grad_glm_kernel", "rh_grad_gather_kernel", "rh_grad_gather_scan_kernel",
                   "rh_density_kernel", "rh_density_fin_kernel")
    ending_in_copy = [r for r in rows if r[1] in row_kernels and r[6] in ("copy", "agpr") and r[7] == 1]
    # single-block arms (`arm-ool`, `arm`: the whole arm is in the block, so a value it copies out must have been computed in it).  A
    # `nested` block -- it opens with an INNER region's restore and ends with the outer one's -- may legitimately copy the inner region's
    # result, which other blocks define: outside this witness (strict builds of the fuzz models show a few; they run against the oracle
    # on the device, tools/gpu_fuzz_sweep.py), counted by the census, not asserted here
    bad = [r for r in ending_in_copy if r[3] in ("arm-ool", "arm") and not (r[8] > 0 and r[9] == r[8])]
    assert not bad, [(r[0], r[1], r[2], r[3], r[8], r[9]) for r in bad[:5]]
    assert all(r[3] in ("arm-ool", "arm", "nested") for r in ending_in_copy), sorted({r[3] for r in ending_in_copy})
    # (and nothing of a launched row-streaming kernel ends in a scratch access there)
    assert not [r for r in rows if r[1] in row_kernels and r[6] == "scratch" and r[7] == 1]
    # the witness itself, on a fault-shaped block: the copied value comes from outside the block
    assert U.arm_value_witness(["v_add_f64 v[2:3], v[4:5], v[6:7]", "v_accvgpr_write_b32 a3, v10", "s_or_b64 exec, exec, s[0:1]"], 2) == (1, 0)
    assert U.arm_value_witness(["v_add_f64 v[2:3], v[4:5], v[6:7]", "v_accvgpr_write_b32 a3, v2", "v_accvgpr_write_b32 a4, v3", "s_or_b64 exec, exec, s[0:1]"], 3) == (2, 2)
    assert U.arm_value_witness(["v_mov_b32_e32 v2, v9", "v_accvgpr_write_b32 a3, v2", "s_or_b64 exec, exec, s[0:1]"], 2) == (2, 0)   # a copy of a copy from outside


def _build_report():
    import json
    path = os.path.join(KCACHE, "build_report.json")
    if not os.path.exists(path):
        pytest.skip("no build report: run __graft_entry__.build() first")
    return json.load(open(path))


def test_every_model_of_build_keeps_a_usable_engine():
    # build() lowers ~100 models (the BASELINE configurations, the reference's own lowerings, the GPU fuzz cases, sampler-kernel
    # variants) and asserts for each that it keeps a density path and a sampling engine; what the engine settled on is in the
    # build report beside the code objects.  Here: no kernel the engine would launch reports a spilled vector register -- but
    # for the two sampler kernels, whose count may be the allocator's VGPR -> AGPR copies (csrc/engine.cpp kernel_health)
    rep = _build_report()
    assert len(rep) >= 80
    for name, r in rep.items():
        ks = {k.split(":", 1)[1]: v for k, v in r["kernels"].items() if k.startswith("base:")}
        gather = "rh_grad_gather_kernel" in ks
        tick = ks.get("rh_tick_kernel", {}).get("fit") and ks.get("rh_density_fin_kernel", {}).get("fit") and \
            (ks.get("rh_grad_gather_kernel" if gather else "rh_grad_kernel", {}).get("fit") or ks.get("rh_grad_glm_kernel", {}).get("fit"))
        chain = ks.get("rh_chain_kernel", {}).get("fit") and ks.get("rh_density_kernel", {}).get("fit")
        assert tick or chain, name
        for k, v in r["kernels"].items():
            if v["fit"] and v["vgpr_spills"] != 0:
                assert k.split(":", 1)[1] in ("rh_chain_kernel", "rh_tick_kernel"), (name, k, v)


def test_heavy_model_is_lowered_memory_resident_and_light_models_are_not():
    # a 134-parameter table-prior model with 134 accumulators per lane does not fit the register file in any shape: the engine
    # ends at the memory-resident lowering (chunks, opaque-index scratch arrays, theta and the outputs in memory) and keeps the tick
    # engine; the README model keeps its bench shape.  (From build()'s report: lowering the heavy one is minutes of compilation.)
    rep = _build_report()
    heavy = [r for n, r in rep.items() if "fuzz_table_3[strict]" in n]
    assert heavy and all(r["heavy"] and r["shape"]["chunk"] > 0 for r in heavy)
    for r in heavy:
        fit = {k.split(":", 1)[1]: v["fit"] for k, v in r["kernels"].items()}
        assert fit["rh_grad_kernel"] and fit["rh_tick_kernel"] and fit["rh_density_fin_kernel"] and fit["rh_density_kernel"]
    assert sum(r["heavy"] for r in rep.values()) < len(rep) // 3        # the last resort, not the rule
    spec = models.linreg(n=8, k=3)
    src, rep1 = _capi.lower_report(spec.rir, _capi.compile_opts(fp_contract=True, factor_outputs=True, grad_chains=8))
    assert rep1["shape"]["chunk"] == 0 and rep1["shape"]["grad_k"] == 8 and rep1["shape"]["grad_unroll"] == 8 and rep1["shape"]["chain_waves"] == 2
    assert all(v["fit"] for v in rep1["kernels"].values()) and "#define RH_HEAVY 0" in src


def test_memory_resident_lowering_is_bit_identical_on_the_host(monkeypatch):
    # the chunked form of the generated functions (values through a volatile scratch array) computes the same bits as the plain
    # form: checked with the host compiler on the generated code itself (tests/host_emulation.py)
    from tests.host_emulation import HostTargets
    from tests.fuzz_models import GPU_FUZZ_CASES, gpu_fuzz_case
    cases = [c for c in GPU_FUZZ_CASES if c[0] == "table"][:2] + [c for c in GPU_FUZZ_CASES if c[0] != "table"][:2]
    for kind, seed, kw in cases:
        spec, qs = gpu_fuzz_case(kind, seed, dict(kw, npoints=3))[:2]
        outs = []
        for chunk in ("0", "5", "40"):
            monkeypatch.setenv("RH_CHUNK", chunk)
            src, _ = _capi.lower_only(spec.rir, _capi.compile_opts(math_mode=_capi.MATH_STRICT), columns=spec.columns, nrows=spec.nrows, compile=False)
            assert ("#define RH_HEAVY 1" in src) == (chunk != "0")
            if chunk == "5":
                assert "double rh_sp[" in src and "rh_oz()" in src       # values do travel through the scratch array
            h = HostTargets(src)
            outs.append([h.eval(q, spec.columns, spec.nrows) for q in qs])
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert np.array_equal(a[0], b[0], equal_nan=True) and a[1] == b[1], (kind, seed)


def test_cache_key_carries_the_compiler_identity(tmp_path):
    # a process that imported torch first binds torch's bundled hiprtc / comgr (another LLVM under the same hiprtc version number):
    # its code objects must not share keys with the ROCm compiler's.  Same model, same options, two fresh processes, two temp caches.
    prog = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "%s"
            "from rainier_amd import _capi, models\n"
            "spec = models.normal_1d()\n"
            "_capi.lower_only(spec.rir, _capi.compile_opts(math_mode=_capi.MATH_STRICT))\n"
            "print(sorted(os.listdir(os.environ['RH_KERNEL_CACHE'])))\n")
    keys = []
    for tag, pre in (("plain", ""), ("torch", "import torch\n")):
        d = tmp_path / tag
        d.mkdir()
        env = dict(os.environ, RH_KERNEL_CACHE=str(d))
        out = subprocess.run([sys.executable, "-c", prog % (ROOT, pre)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        keys.append(eval(out.stdout.strip().splitlines()[-1]))
    assert keys[0] and keys[1] and not set(keys[0]) & set(keys[1]), keys
