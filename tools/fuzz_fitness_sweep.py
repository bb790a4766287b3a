"""The wide fuzz sweep of tools/gpu_fuzz_sweep.py (seeds LO..HI: eight-slot programs at six row counts, table-prior models at three
group-size ranges; both math modes) through the ENGINE'S LOWERING ONLY -- no GPU: what shape does every model settle on, how many
attempts did that take, does it keep a usable engine, which kernels are out of use?  (VERDICT r3 item 4's "done" criterion: no
launched kernel with vector spills anywhere in the sweep.)      python tools/fuzz_fitness_sweep.py LO HI [workers] > report.txt
The code objects go to RH_KERNEL_CACHE if set (use a scratch directory: the sweep is ~650 compilations)."""
import collections
import json
import multiprocessing as mp
import os
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(args):
    kind, seed, kw, mode = args
    import __graft_entry__ as G
    from rainier_amd import _capi
    from tests.fuzz_models import gpu_fuzz_case
    spec = gpu_fuzz_case(kind, seed, dict(kw, npoints=0))[0]
    opts = _capi.compile_opts(math_mode=_capi.MATH_STRICT) if mode == "strict" else _capi.compile_opts(fp_contract=True, factor_outputs=True)
    t0 = time.time()
    try:
        _, rep = _capi.lower_report(spec.rir, opts, columns=spec.columns, nrows=spec.nrows)
    except Exception as e:      # noqa: BLE001
        return dict(kind=kind, seed=seed, mode=mode, n_params=spec.n_params, error=str(e)[:200])
    unfit = sorted(k for (tag, k), v in rep["kernels"].items() if not v["fit"] and k.startswith("rh_"))
    launched_spills = [k for (tag, k), v in rep["kernels"].items() if v["fit"] and v["vgpr_spills"] != 0 and k not in ("rh_chain_kernel", "rh_tick_kernel")]
    return dict(kind=kind, seed=seed, mode=mode, n_params=spec.n_params, seconds=round(time.time() - t0, 1), usable=G._usable(rep), unfit=unfit,
                launched_with_spills=launched_spills, **{k: rep["shape"][k] for k in ("attempts", "rows_unroll", "grad_unroll", "grad_k", "chain_waves", "chunk")})


def main():
    from tools.gpu_fuzz_sweep import cases
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    nw = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 2) - 1)
    jobs = [(k, s, kw, mode) for k, s, kw in cases(lo, hi) for mode in ("strict", "fast")]
    t0 = time.time()
    with mp.get_context("fork").Pool(nw) as pool:
        res = pool.map(work, jobs, chunksize=1)
    for r in res:
        print(json.dumps(r))
    ok = [r for r in res if "error" not in r]
    summary = dict(builds=len(res), errors=len(res) - len(ok), usable=sum(r["usable"] for r in ok), memory_resident=sum(r["chunk"] > 0 for r in ok),
                   chain_engine_out_of_use=sum("rh_chain_kernel" in r["unfit"] for r in ok), tick_engine_out_of_use=sum("rh_tick_kernel" in r["unfit"] or "rh_grad_kernel" in r["unfit"] and "rh_grad_gather_kernel" not in r["unfit"] for r in ok),
                   launched_kernels_with_spills=sum(bool(r["launched_with_spills"]) for r in ok),
                   attempts=dict(collections.Counter(r["attempts"] for r in ok)), wall_seconds=round(time.time() - t0))
    print("SUMMARY", json.dumps(summary))


if __name__ == "__main__":
    main()
