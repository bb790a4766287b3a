"""Round-6 diagnostic: the centred hierarchical model's chains depended on which other chains shared a launch (tests/test_gpu_live_chains.py,
case 3).  Where does the dependency come from -- the gradient path (q -> (logp, grad) through the row-streaming kernels) or the sampler side?
usage (GPU box): python tools/r6_centred_diag.py [groups per_group]"""
import os, sys
os.environ.setdefault("RH_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import _capi, models

G, per = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (700, 100)
fast = dict(fp_contract=True, factor_outputs=True)


def grad_level(spec, label):
    m = R.Model(spec, device=0, **fast)
    rng = np.random.default_rng(3)
    qs = rng.normal(size=(8, spec.n_params)) * 0.3
    base = m.density_batch(qs, engine=_capi.ENGINE_TICK)      # (gather mode: 64 row splits whatever the chain count)
    ref = np.concatenate([base[0][:, None], base[1]], axis=1)
    worst = {}
    for name, order in (("reversed", list(range(7, -1, -1))), ("pairs swapped", [1, 0, 3, 2, 5, 4, 7, 6]), ("alone", None), ("tiled x5 permuted", list(np.random.default_rng(1).permutation(np.arange(40) % 8)))):
        if order is None:
            got = np.stack([np.concatenate([[lp[0]], g[0]]) for lp, g in (m.density_batch(qs[i:i + 1], engine=_capi.ENGINE_TICK) for i in range(8))])
            idx = list(range(8))
        else:
            lp, g = m.density_batch(qs[order], engine=_capi.ENGINE_TICK)
            got = np.concatenate([lp[:, None], g], axis=1); idx = order
        bad = [(j, int(np.sum(got[j] != ref[i])), float(np.max(np.abs(got[j] - ref[i]) / (np.abs(ref[i]) + 1e-300)))) for j, i in enumerate(idx) if not np.array_equal(got[j], ref[i])]
        worst[name] = bad[:4]
    print(label, "gradient path, same q in other company:", {k: (v if v else "identical") for k, v in worst.items()})
    m.close()


def chain_level(spec, label, sampler):
    m = R.Model(spec, device=0, **fast)
    cfg = lambda: R.make_config(4, 10, sampler, R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK, gradSplits=16)
    seeds = [5200 + c for c in range(11)]
    runs = {}
    for name, env, sd in (("all", {}, seeds), ("all again", {}, seeds), ("uncompacted", {"RH_COMPACT": "0"}, seeds), ("first 3", {}, seeds[:3]), ("chain 0 alone", {}, seeds[:1]),
                          ("scan walk everywhere", {"RH_GATHER_SCAN": "1"}, seeds), ("scan walk, first 3", {"RH_GATHER_SCAN": "1"}, seeds[:3]),
                          ("no value-free", {"RH_VALUE_FREE": "0"}, seeds)):
        os.environ.update(env)
        try:
            runs[name] = m.sample(cfg(), seeds=sd).chains
        finally:
            for k in env: os.environ.pop(k, None)
    ref = runs["all"]
    out = {}
    for name, ch in runs.items():
        n = ch.shape[0]
        diff = [c for c in range(n) if not np.array_equal(ch[c], ref[c])]
        first = None
        if diff:
            c = diff[0]; it = int(np.argwhere(np.any(ch[c] != ref[c], axis=1))[0][0])
            first = (c, it, float(np.max(np.abs(ch[c][it] - ref[c][it]))))
        out[name] = "identical" if not diff else "chains %s differ; first: chain %d iteration %d max |d| %.3g" % (diff[:6], *first)
    print(label, type(sampler).__name__, out)
    m.close()


for label, spec in (("centred", models.hier_negbin_centred(G, per)), ("non-centred", models.hier_negbin(G, per, seed=3))):
    grad_level(spec, label)
    for smp in (R.HMCSampler(4), R.NUTSSampler(4)):
        chain_level(spec, label, smp)
