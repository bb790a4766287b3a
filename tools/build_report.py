"""What the engine makes of every model build() lowers (no GPU): the shape each one settles on and which of its kernels are fit to
run (csrc/engine.cpp kernel_health: no spilled vector registers, no vector instruction ahead of a join block's exec restore).

usage: python tools/build_report.py [--all]      (default: only the lines that are not 'everything fit at the first attempt')"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)




def main():
    import __graft_entry__ as G
    from rainier_amd import _capi
    show_all = "--all" in sys.argv
    n = nbad = 0
    for name, rir, opts, check, kw in G.build_jobs():
        _, rep = G.lower_job(rir, opts, kw)
        n += 1
        unfit = {k: v for k, v in rep["kernels"].items() if not v["fit"]}
        sh = rep["shape"]
        if show_all or unfit or sh["attempts"] > 1 + (1 if opts.with_nuts else 0):
            print("%-44s attempts=%d U_rows=%d U=%d K=%d waves=%d pipe=%d" % (name, sh["attempts"], sh["rows_unroll"], sh["grad_unroll"], sh["grad_k"],
                                                                                sh["chain_waves"], sh["grad_pipeline"]))
            for (tag, k), v in unfit.items():
                print("      UNFIT %s %s: %s" % (tag, k, v["why"]))
            nbad += bool(unfit)
    print("%d models lowered, %d with a kernel out of use" % (n, nbad))


if __name__ == "__main__":
    main()
