"""One-off wide sweep of the seeded fuzz models through the kernels (tests/test_gpu_fuzz.py keeps 24 of them in the suite).
  python tools/gpu_fuzz_sweep.py prebuild LO HI   -- cross-compile the cases' code objects into the in-tree cache (no GPU needed)
  python tools/gpu_fuzz_sweep.py run LO HI        -- on the GPU: every case, both math modes, both engines, vs the oracle at 1e-12"""
import os, sys
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rainier_amd import _capi
from tests.fuzz_models import gpu_fuzz_case

FAST = dict(fp_contract=True, factor_outputs=True)
STRICT = dict(math_mode=_capi.MATH_STRICT)


def cases(lo, hi):
    out = []
    for s in range(lo, hi):
        out.append(("slots", s, dict(n=(4096, 1000, 70, 513, 64, 8192)[s % 6], npoints=(11, 3, 8)[s % 3])))
        if s % 3 == 0:
            out.append(("table", s, dict(npoints=5, per_range=((2, 6), (64, 80), (20, 40))[(s // 3) % 3])))
    return out


def prebuild(lo, hi, worker, nworkers):
    for i, (kind, seed, kw) in enumerate(cases(lo, hi)):
        if i % nworkers != worker:
            continue
        spec = gpu_fuzz_case(kind, seed, dict(kw, npoints=0))[0]
        for opts in (STRICT, FAST):
            _capi.lower_only(spec.rir, _capi.compile_opts(**opts), columns=spec.columns, nrows=spec.nrows)


def run(lo, hi):
    import rainier_amd as R
    from tests import oracle_lib as O
    bad = done = chains_done = 0
    for kind, seed, kw in cases(lo, hi):
        spec, qs, mode = gpu_fuzz_case(kind, seed, kw)
        if not qs:
            continue
        d = O.OracleDensity(spec)
        refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
        for name, opts in (("strict", STRICT), ("fast", FAST)):
            try:
                if os.environ.get("SWEEP_VERBOSE"): print("case", kind, seed, kw, name, flush=True)
                m = R.Model(spec, device=0, **opts)
                gather = "#define RH_HAS_GATHER 1\n" in m.hip_source
                runs = [(_capi.ENGINE_TICK, 0), (_capi.ENGINE_TICK, 3)] + ([] if gather or os.environ.get("SWEEP_NO_CHAIN") else [(_capi.ENGINE_CHAIN, 0)])
                if os.environ.get("SWEEP_ONLY_CHAIN"):
                    runs = [] if gather else [(_capi.ENGINE_CHAIN, 0)]
                for engine, splits in runs:
                    if os.environ.get("SWEEP_VERBOSE"): print("  engine", engine, "splits", splits, flush=True)
                    lp, g = m.density_batch(np.asarray(qs), engine=engine, grad_splits=splits)
                    for c, (ref, ab) in enumerate(refs):
                        got = np.concatenate([[lp[c]], g[c]])
                        ratio = np.abs(got - ref) / (ab + 1e-300)
                        ok = np.all((ratio <= 1e-12) | (np.isnan(got) & np.isnan(ref)))
                        if not ok:
                            bad += 1
                            import re
                            print("FAIL", kind, seed, kw, name, "engine", engine, "splits", splits, "point", c, "worst", float(np.nanmax(ratio)),
                                  re.findall(r"#define RH_GRAD_[UK] \d+", m.hip_source), flush=True)
                            break
                # SWEEP_CHAINS=1: also through the SAMPLER kernels (they carry their own inlined copy of the density): 4 iterations of
                # tame static HMC from every engine in use against the oracle's chains, strict builds (as tests/test_gpu_fuzz.py)
                if os.environ.get("SWEEP_CHAINS") and name == "strict" and np.all(np.isfinite(O.OracleDensity(spec).update(np.asarray(qs[0], dtype=np.float64)))):
                    from tests.test_gpu_parity import _oracle_cfg
                    cfg = lambda e: R.make_config(4, 0, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner(), engine=e)
                    seeds = [4000 + seed, 4100 + seed]
                    want = np.array([O.sample_model(spec, _oracle_cfg(cfg(0), O.JM_DET), sd)[0] for sd in seeds])
                    eng = m.engines()
                    for e in ([_capi.ENGINE_TICK] if eng["tick"] else []) + ([_capi.ENGINE_CHAIN] if eng["chain"] else []):
                        got = m.sample(cfg(e), seeds=seeds).chains
                        chains_done += 1
                        if np.all(np.isfinite(want)) and not np.allclose(got, want, rtol=1e-8, atol=1e-10):
                            bad += 1
                            print("FAIL-CHAIN", kind, seed, kw, "engine", e, "max abs diff", float(np.nanmax(np.abs(got - want))), flush=True)
                m.close()
                done += 1
            except Exception as e:
                bad += 1
                print("ERROR", kind, seed, kw, name, repr(e)[:200], flush=True)
    print("sweep", lo, hi, "model builds", done, "sampler runs vs the oracle's chains", chains_done, "failures", bad, flush=True)


if __name__ == "__main__":
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if mode == "prebuild":
        import multiprocessing as mp
        nw = 8
        ps = [mp.get_context("fork").Process(target=prebuild, args=(lo, hi, w, nw)) for w in range(nw)]
        [p.start() for p in ps]; [p.join() for p in ps]
        print("prebuilt", len(cases(lo, hi)), "cases; exit codes", [p.exitcode for p in ps])
    else:
        run(lo, hi)
