"""ctypes binding of include/rainier_hip.h (the C-ABI drop-in boundary).

The shared library is built in-tree (rainier_amd/librainier_hip.so) by `make -C rainier_amd/csrc`
(see __graft_entry__.build).  There is no CPU fallback: without the library, or without a HIP
device, every compute entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librainier_hip.so")

RH_OK, RH_E_INVALID, RH_E_COMPILE, RH_E_DEVICE, RH_E_LOOKUP, RH_E_UNSUPPORTED = range(6)
MATH_FAST, MATH_STRICT = 0, 1
SAMPLER_HMC, SAMPLER_EHMC, SAMPLER_NUTS = 0, 1, 2
STEP_DUALAVG, STEP_STATIC = 0, 1
OPT_CONVERGED, OPT_NOT_DESCENT, OPT_MAX_EVALS = 0, 1, 2
MASS_IDENTITY, MASS_DIAG_WINDOWED, MASS_STATIC_DIAG, MASS_DENSE_WINDOWED = 0, 1, 2, 3
ENGINE_AUTO, ENGINE_CHAIN, ENGINE_TICK = 0, 1, 2

# every symbol include/rainier_hip.h declares (checked by tests/test_capi_cpu.py)
EXPORTS = [
    "rh_model_create", "rh_model_clone", "rh_model_destroy", "rh_model_nvars", "rh_model_hip_source", "rh_last_error",
    "rh_model_engines", "rh_compile_count",
    "rh_density_eval", "rh_density_eval_ex", "rh_config_default", "rh_sample", "rh_sample_multi", "rh_sampler_create", "rh_sampler_destroy",
    "rh_sampler_warmup", "rh_sampler_run", "rh_sampler_draws", "rh_sampler_draws_device", "rh_sampler_stats",
    "rh_sampler_timing", "rh_sampler_progress", "rh_sampler_mass_dense", "rh_optimize", "rh_diagnostics", "rh_abi_version", "rh_device_count", "rh_requirements_eval",
    "rh_comm_unique_id", "rh_comm_create", "rh_comm_destroy", "rh_comm_allgather_draws", "rh_comm_allreduce_max", "rh_device_synchronize",
]


class CompileOpts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("math_mode", C.c_int32),
                ("fp_contract", C.c_int32), ("rows_unroll", C.c_int32), ("grad_chains", C.c_int32),
                ("grad_unroll", C.c_int32), ("factor_outputs", C.c_int32), ("with_nuts", C.c_int32), ("reserved", C.c_int32)]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("iterations", C.c_int32), ("warmup", C.c_int32), ("sampler", C.c_int32),
        ("hmc_steps", C.c_int32), ("ehmc_max_steps", C.c_int32), ("ehmc_min_steps", C.c_int32),
        ("ehmc_buf_size", C.c_int32), ("ehmc_p_count", C.c_double), ("step_tuner", C.c_int32),
        ("mass_tuner", C.c_int32), ("dualavg_delta", C.c_double), ("static_step", C.c_double),
        ("mass_init_window", C.c_int32), ("mass_skip_first", C.c_int32), ("mass_skip_last", C.c_int32),
        ("nuts_max_depth", C.c_int32), ("mass_expansion", C.c_double), ("static_mass", C.POINTER(C.c_double)),
        ("engine", C.c_int32), ("grad_splits", C.c_int32), ("rng_next_gaussian", C.POINTER(C.c_double)), ("reserved", C.c_int64 * 2),
    ]


class ChainStats(C.Structure):
    _fields_ = [("leapfrog_steps", C.c_int64), ("warmup_leapfrog_steps", C.c_int64),
                ("gradient_evaluations", C.c_int64), ("accepted", C.c_int64), ("mean_accept_prob", C.c_double),
                ("step_size", C.c_double), ("error", C.c_int32), ("reserved", C.c_int32), ("bfmi", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("launches", C.c_int64), ("density_evals", C.c_int64),
                ("row_chain_evals", C.c_int64), ("dominant_kernel", C.c_char * 64), ("chain_slots", C.c_int64),
                ("steady_kernel_ms", C.c_double), ("steady_launches", C.c_int64), ("steady_density_evals", C.c_int64)]


class RainierHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rainier_hip status %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load librainier_hip.so; raises if the native library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RainierHipError(RH_E_DEVICE, "native library missing: %s (run __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    L.rh_model_create.restype = C.c_int
    L.rh_model_create.argtypes = [vp, C.c_size_t, C.POINTER(dp), C.POINTER(C.c_int64), C.POINTER(CompileOpts), C.POINTER(vp)]
    L.rh_model_clone.restype = C.c_int
    L.rh_model_clone.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.rh_model_destroy.argtypes = [vp]
    L.rh_model_nvars.argtypes = [vp]
    L.rh_model_hip_source.restype = C.c_char_p; L.rh_model_hip_source.argtypes = [vp]
    L.rh_last_error.restype = C.c_char_p; L.rh_last_error.argtypes = [vp]
    L.rh_model_engines.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
    L.rh_compile_count.restype = C.c_int64; L.rh_compile_count.argtypes = []
    L.rh_density_eval.argtypes = [vp, dp, C.c_int32, dp, dp]
    L.rh_density_eval_ex.argtypes = [vp, dp, C.c_int32, C.c_int32, C.c_int32, dp, dp]
    L.rh_config_default.argtypes = [C.POINTER(Config)]
    L.rh_sample.argtypes = [vp, C.POINTER(Config), C.POINTER(C.c_int64), C.c_int32, dp, dp, C.POINTER(ChainStats)]
    L.rh_sample_multi.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(Config), C.POINTER(C.c_int64), C.c_int32, dp, dp, C.POINTER(ChainStats)]
    L.rh_sampler_create.argtypes = [vp, C.POINTER(Config), C.POINTER(C.c_int64), C.c_int32, C.POINTER(vp)]
    L.rh_sampler_destroy.argtypes = [vp]
    L.rh_sampler_warmup.argtypes = [vp]
    L.rh_sampler_run.argtypes = [vp, C.c_int32]
    L.rh_sampler_draws.argtypes = [vp, C.c_int32, C.c_int32, dp]
    L.rh_sampler_draws_device.argtypes = [vp, C.POINTER(vp)]
    L.rh_sampler_stats.argtypes = [vp, C.POINTER(ChainStats), dp]
    L.rh_sampler_mass_dense.argtypes = [vp, dp]
    L.rh_optimize.argtypes = [vp, dp, C.c_int32, C.c_int32, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rh_sampler_timing.argtypes = [vp, C.POINTER(Timing), C.c_int]
    L.rh_sampler_progress.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rh_diagnostics.argtypes = [dp, C.c_int32, C.c_int32, C.c_int32, dp, dp]
    L.rh_requirements_eval.argtypes = [vp, C.c_size_t, C.POINTER(CompileOpts), dp, C.c_int64, dp]
    L.rh_comm_unique_id.argtypes = [C.c_char_p]
    L.rh_comm_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.rh_comm_destroy.argtypes = [vp]
    L.rh_comm_allgather_draws.argtypes = [vp, vp, dp, C.POINTER(vp)]
    L.rh_comm_allreduce_max.argtypes = [vp, dp]
    L.rh_device_synchronize.argtypes = [C.c_int32]
    L.rh_lower_only.argtypes = [vp, C.c_size_t, C.POINTER(CompileOpts), C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.rh_lower_only_data.argtypes = [vp, C.c_size_t, C.POINTER(dp), C.POINTER(C.c_int64), C.POINTER(CompileOpts), C.c_char_p,
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.rh_free.argtypes = [vp]
    L.rh_simplify_rir.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rh_canonicalize_rir.argtypes = [vp, C.c_size_t, C.POINTER(dp), C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
    L.rh_lift_rir.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(dp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                              C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.rh_selftest.argtypes = [vp, C.c_int32, C.c_int64, dp, dp, C.c_int32]
    _lib = L
    return L


def check(rc, model=None):
    if rc != RH_OK:
        msg = lib().rh_last_error(model)
        raise RainierHipError(rc, msg.decode(errors="replace") if msg else "")


def dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def compile_opts(device=-1, math_mode=MATH_FAST, fp_contract=False, rows_unroll=0, grad_chains=0, grad_unroll=0,
                 factor_outputs=False, with_nuts=False):
    o = CompileOpts()
    o.struct_size = C.sizeof(CompileOpts)
    o.device, o.math_mode, o.fp_contract, o.rows_unroll = device, math_mode, int(fp_contract), rows_unroll
    o.grad_chains, o.grad_unroll, o.factor_outputs = grad_chains, grad_unroll, int(factor_outputs)
    o.with_nuts = int(with_nuts)
    return o


def canonicalize_rir(rir: bytes, columns, nrows, fast: bool = False, refactor: bool = False):
    """What rh_model_create does to a program and its data before lowering (csrc/columns.cpp; with refactor -- fast builds --
    also csrc/refactor.cpp: re-association and Model.observe's 8-way split rolled back into rows), returned as
    (RIR, parts, nrows): parts[j] = [(index into `columns` or 0xFFFFFFFF for zeros, block length), ...] whose data,
    concatenated, is column j of the rewritten program; nrows = its row count per target (test hook, no device needed)."""
    L = lib()
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in columns]
    arr = (C.POINTER(C.c_double) * max(1, len(cols)))(*[dptr(c) for c in cols])
    nr = (C.c_int64 * max(1, len(nrows)))(*[int(x) for x in nrows])
    cap = 64 + 3 * len(cols) * 20          # a rolled column has one block per slot (8, or 9 with the initial chunk)
    words = (C.c_uint32 * cap)()
    nw = C.c_uint32(cap)
    nr_out = (C.c_int64 * max(1, len(nrows)))()
    out, n = C.c_void_p(), C.c_size_t(0)
    buf = C.create_string_buffer(rir, len(rir))
    check(L.rh_canonicalize_rir(buf, len(rir), arr, nr, int(fast), int(refactor), C.byref(out), C.byref(n), words, C.byref(nw), nr_out))
    try:
        parts, i = [], 0
        while i < nw.value:
            k = words[i]; parts.append([(int(words[i + 1 + 2 * j]), int(words[i + 2 + 2 * j])) for j in range(k)]); i += 1 + 2 * k
        return C.string_at(out, n.value), parts, [int(nr_out[t]) for t in range(len(nrows))]
    finally:
        L.rh_free(out)


def lift_rir(rir: bytes, nrows=None, fast: bool = False):
    """What rh_model_create's loader makes of a program with more than 64 targets (csrc/lift.cpp + the merge of data-free runs):
    (RIR, synthesised columns -- they follow the caller's in the rewritten program --, rows of the synthesised target, and with
    `nrows` the row count of every target of the rewritten program) -- test hook, no device needed."""
    L = lib()
    out, n = C.c_void_p(), C.c_size_t(0)
    cols, nc, nr = C.POINTER(C.c_double)(), C.c_uint32(0), C.c_uint32(0)
    buf = C.create_string_buffer(rir, len(rir))
    nr_in = (C.c_int64 * max(1, len(nrows)))(*[int(x) for x in nrows]) if nrows is not None else None
    nr_out = (C.c_int64 * 64)() if nrows is not None else None
    check(L.rh_lift_rir(buf, len(rir), C.byref(out), C.byref(n), C.byref(cols), C.byref(nc), C.byref(nr), nr_in, nr_out, int(fast)))
    try:
        arr = np.ctypeslib.as_array(cols, shape=(max(1, nc.value * nr.value),)).copy()
        rir2 = C.string_at(out, n.value)
        cols2 = [arr[c * nr.value:(c + 1) * nr.value] for c in range(nc.value)]   # zero-padded to the longest lifted group
        res = rir2, cols2, nr.value
        if nrows is not None:
            import struct
            w = struct.unpack_from("<%dI" % (len(rir2) // 4), rir2)
            rows_t = [int(nr_out[t]) for t in range(w[3])]
            per_col, pos = [], 6
            for t in range(w[3]):                           # targets: { n_cols, reserved, outputs[n_params + 1] }
                per_col += [rows_t[t]] * w[pos]; pos += 3 + w[2]
            first = len(per_col) - nc.value                 # the synthesised columns follow the caller's
            res = rir2, [c[:per_col[first + j]] for j, c in enumerate(cols2)], nr.value, rows_t
        return res
    finally:
        L.rh_free(out); L.rh_free(cols)


def simplify_rir(rir: bytes, fast: bool = False) -> bytes:
    """The emitter's clean-up pass (csrc/simplify.cpp) applied to an RIR blob, returned as RIR (test hook)."""
    L = lib()
    out, n = C.c_void_p(), C.c_size_t(0)
    buf = C.create_string_buffer(rir, len(rir))
    check(L.rh_simplify_rir(buf, len(rir), int(fast), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        L.rh_free(out)


def lower_only(rir: bytes, opts: CompileOpts = None, arch: str = "gfx950", columns=None, nrows=None, compile: bool = True):
    """RIR -> HIP source -> gfx950 code object, without a device.  Returns (source, code_size).  With columns / nrows the
    data-dependent passes of rh_model_create (column canonicalisation) run as well; compile = False stops after the lowering
    (source only, code_size 0)."""
    L = lib()
    src = C.c_char_p()
    size = C.c_size_t(0)
    buf = C.create_string_buffer(rir, len(rir))
    o = opts if opts is not None else compile_opts()
    if columns is not None:
        cols = [np.ascontiguousarray(c, dtype=np.float64) for c in columns]
        arr = (C.POINTER(C.c_double) * max(1, len(cols)))(*[dptr(c) for c in cols])
        nr = (C.c_int64 * max(1, len(nrows)))(*[int(x) for x in nrows])
        rc = L.rh_lower_only_data(buf, len(rir), arr, nr, C.byref(o), arch.encode(), C.byref(src), C.byref(size) if compile else None)
    else:
        rc = L.rh_lower_only_data(buf, len(rir), None, None, C.byref(o), arch.encode(), C.byref(src), C.byref(size) if compile else None)
    text = src.value.decode() if src.value else ""
    if src:
        L.rh_free(src)
    check(rc)
    return text, size.value


def lower_report(rir: bytes, opts: CompileOpts = None, arch: str = "gfx950", columns=None, nrows=None):
    """lower_only plus what the engine makes of the result (test hook, no device): returns (source, report) where report is
    {"shape": {attempts, rows_unroll, grad_unroll, grad_k, chain_waves, ...}, "kernels": {(tag, kernel): {vgprs, sgprs, vgpr_spills,
    sgpr_spills, scratch, fit, why}}} -- tag "base" or "variantN"; fit = the engine would launch it (csrc/engine.cpp kernel_health)."""
    L = lib()
    L.rh_lower_report_data.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64), C.POINTER(CompileOpts),
                                       C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
    src, size, rep = C.c_char_p(), C.c_size_t(0), C.c_void_p()
    buf = C.create_string_buffer(rir, len(rir))
    o = opts if opts is not None else compile_opts()
    arr = nr = None
    if columns is not None:
        cols = [np.ascontiguousarray(c, dtype=np.float64) for c in columns]
        arr = (C.POINTER(C.c_double) * max(1, len(cols)))(*[dptr(c) for c in cols])
        nr = (C.c_int64 * max(1, len(nrows)))(*[int(x) for x in nrows])
    rc = L.rh_lower_report_data(buf, len(rir), arr, nr, C.byref(o), arch.encode(), C.byref(src), C.byref(size), C.byref(rep))
    text = src.value.decode() if src.value else ""
    if src:
        L.rh_free(src)
    check(rc)
    try:
        return text, parse_report(C.string_at(rep).decode())
    finally:
        L.rh_free(rep)


def parse_report(text: str):
    out = {"shape": {}, "kernels": {}}
    for ln in text.splitlines():
        head, _, why = ln.partition(" why=")
        f = head.split()
        if f and f[0].startswith("attempts="):
            out["shape"] = {k: int(v) for k, v in (x.split("=") for x in f)}
            continue
        kv = dict(x.split("=", 1) for x in f[1:])
        name = kv.pop("kernel")
        out["kernels"][(f[0], name)] = dict({k: int(v) for k, v in kv.items()}, why=why)
    return out


def code_object_report(code: bytes):
    """Every kernel of a code object: registers, spills, scratch, and whether the engine would launch it (test hook)."""
    L = lib()
    L.rh_code_object_report.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    rep = C.c_void_p()
    check(L.rh_code_object_report(code, len(code), C.byref(rep)))
    try:
        return parse_report(C.string_at(rep).decode())["kernels"]
    finally:
        L.rh_free(rep)


def code_object_offsets(code: bytes, kernel: str):
    """The engine's own instruction walk of one kernel (offsets from the kernel's first byte): checked against llvm-objdump."""
    L = lib()
    L.rh_code_object_offsets.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]
    offs, n = C.POINTER(C.c_uint32)(), C.c_size_t(0)
    check(L.rh_code_object_offsets(code, len(code), kernel.encode(), C.byref(offs), C.byref(n)))
    try:
        return [int(offs[i]) for i in range(n.value)]
    finally:
        L.rh_free(offs)
