"""is "NUTS(10) does not leave max depth and R-hat is far from 1" on cfg 5 a property of the POSTERIOR under the
configuration BASELINE.json names, or of the engine?  The CPU oracle (oracle/sampler.c: the same algorithm statement the GPU is
bit-compared with on data-free models) runs the same configuration -- non-centred hierarchical NegBin, 100 observations per group,
NUTS max depth 10, DualAvg(0.8), windowed diagonal mass (10, 1.5, 10, 6), 36 + 6 iterations -- on a reduced number of groups, and the
GPU runs the SAME instance with the SAME seeds:   python tools/cfg5_oracle_nuts.py oracle G [chains]   (CPU, no GPU needed)
                                                  python tools/cfg5_oracle_nuts.py gpu G [chains]      (on the GPU box)
Both print mean leapfrog steps per iteration (tree depth), acceptance, step size, and R-hat of the four shared parameters."""
import json
import os
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rainier_amd import models  # noqa: E402

WARM, ITERS = 36, 6


def rhat(draws):           # [chains][iters] -> split-free R-hat as core/Trace.scala:52-75 computes it
    m, n = draws.shape
    means = draws.mean(axis=1); var = draws.var(axis=1, ddof=1)
    B = n * means.var(ddof=1); W = var.mean()
    return float(np.sqrt(((n - 1) / n * W + B / n) / W))


def summary(tag, draws, lf, wlf, acc, eps, secs):
    print(json.dumps({"who": tag, "chains": draws.shape[0], "mean_leapfrog_per_iteration": float(np.mean(lf)) / ITERS,
                      "tree_depth": float(np.log2(np.mean(lf) / ITERS + 1)), "warmup_leapfrog_per_iteration": float(np.mean(wlf)) / WARM,
                      "mean_accept": float(np.mean(acc)), "step_size": [float(np.min(eps)), float(np.max(eps))],
                      "rhat_m_s_b0_b1": [rhat(draws[:, :, j]) for j in range(4)], "seconds": secs}), flush=True)


def main():
    who, G = sys.argv[1], int(sys.argv[2])
    chains = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    spec = models.hier_negbin(G, 100, seed=5)
    seeds = [9000 + c for c in range(chains)]
    if who == "oracle":
        import multiprocessing as mp
        from tests import oracle_lib as O
        global _SPEC
        _SPEC = spec                         # (inherited by the forked workers: ctypes configs cannot be pickled)
        t0 = time.time()
        with mp.get_context("fork").Pool(min(chains, os.cpu_count() or 1)) as pool:
            res = pool.map(_one, seeds)
        draws = np.array([r[0] for r in res])
        summary("oracle (CPU)", draws, [r[1] for r in res], [r[2] for r in res], [r[3] for r in res], [r[4] for r in res], time.time() - t0)
    else:
        import rainier_amd as R
        m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
        assert "rh_grad_gather_kernel" in m.hip_source
        cfg = R.make_config(ITERS, WARM, R.NUTSSampler(10), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(10, 1.5, 10, 6))
        t0 = time.time()
        tr = m.sample(cfg, seeds=seeds)
        summary("engine (MI355X, gather mode)", tr.chains, [s.leapfrogSteps for s in tr.stats], [s.warmupLeapfrogSteps for s in tr.stats],
                [s.meanAcceptProb for s in tr.stats], [s.stepSize for s in tr.stats], time.time() - t0)


_SPEC = None


def _one(seed):
    from tests import oracle_lib as O
    ocfg = O.make_config(sampler=O.NUTS, nuts_max_depth=10, iterations=ITERS, warmup=WARM, step_tuner=O.STEP_DUALAVG, delta=0.8,
                         mass_tuner=O.MASS_DIAG_WINDOWED, init_window=10, expansion=1.5, skip_first=10, skip_last=6, math_mode=O.JM_DET)
    d, _, st = O.sample_model(_SPEC, ocfg, seed)
    return d, st.leapfrog_steps, st.warmup_leapfrog_steps, st.mean_accept_prob if hasattr(st, "mean_accept_prob") else float("nan"), st.step_size


if __name__ == "__main__":
    main()
