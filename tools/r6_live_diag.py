"""Round-6 diagnostic (second session): on the centred form of cfg 5, NUTS chains served through compacted launches and through launches that
serve every chain (RH_COMPACT=0) differed in 2 of 11 chains (gpurun_out/r6_c/centred_diag.txt).  Two questions:
  1. gradient level -- does a chain's (logp, grad) change when the launch serves it through a compacted list (RH_EVAL_LIVE, a subset of
     the chains, other slot / group / pad pattern) instead of the identity list?   q taken wide (NUTS warm-up visits extreme points).
  2. sampler level -- compacted vs uncompacted under variants that take one suspect out at a time (K = 1, the fallback walks, the scan walk
     everywhere, the general tick), EHMC and NUTS.
usage (GPU box): python tools/r6_live_diag.py [groups per_group]"""
import os, sys
os.environ.setdefault("RH_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import _capi, models

G, per = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (700, 100)
fast = dict(fp_contract=True, factor_outputs=True)


def setenv(env):
    os.environ.update(env)


def clearenv(env):
    for k in env:
        os.environ.pop(k, None)


def grad_level(spec, label, scale):
    m = R.Model(spec, device=0, **fast)
    rng = np.random.default_rng(11)
    nc = 23
    qs = rng.normal(size=(nc, spec.n_params)) * scale
    lp, g = m.density_batch(qs, engine=_capi.ENGINE_TICK, grad_splits=64)
    ref = np.concatenate([lp[:, None], g], axis=1)
    out = {}
    subsets = [[7], [8], [7, 8], [3, 7, 8], [0, 1, 2, 7], [7, 8, 9, 10], [1, 7, 8, 9, 10], [0, 5, 6, 7, 8, 20, 22], list(range(0, 23, 2)), list(range(1, 23, 2)),
               list(range(5, 23)), list(range(0, 22))]
    for sub in subsets:
        env = {"RH_EVAL_LIVE": ",".join(str(c) for c in sub)}
        setenv(env)
        try:
            lp2, g2 = m.density_batch(qs, engine=_capi.ENGINE_TICK, grad_splits=64)
        finally:
            clearenv(env)
        got = np.concatenate([lp2[:, None], g2], axis=1)
        bad = []
        for c in sub:
            if not np.array_equal(got[c], ref[c], equal_nan=True):
                w = np.flatnonzero(~((got[c] == ref[c]) | (np.isnan(got[c]) & np.isnan(ref[c]))))
                bad.append((c, len(w), [int(i) for i in w[:5]], float(np.nanmax(np.abs(got[c][w] - ref[c][w]) / (np.abs(ref[c][w]) + 1e-300)))))
        out[str(sub) if len(sub) < 8 else "%d chains from %d" % (len(sub), sub[0])] = bad or "identical"
    print(label, "scale", scale, "finite rows:", int(np.sum(np.all(np.isfinite(ref), axis=1))), "of", nc, "| listed vs identity:", out, flush=True)
    m.close()


def chain_level(spec, label, sampler, variants, build=None, nchains=11):
    out = {}
    for vname, env, bkw in variants:
        setenv(env)
        try:
            m = R.Model(spec, device=0, **dict(build or fast, **bkw))
            cfg = lambda: R.make_config(4, 10, sampler, R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK, gradSplits=16)
            seeds = [5200 + c for c in range(nchains)]
            a = m.sample(cfg(), seeds=seeds)
            os.environ["RH_COMPACT"] = "0"
            b = m.sample(cfg(), seeds=seeds)
            os.environ.pop("RH_COMPACT")
            m.close()
        finally:
            clearenv(env)
        diff = [c for c in range(nchains) if not np.array_equal(a.chains[c], b.chains[c])]
        lf = [(int(sa.leapfrogSteps + sa.warmupLeapfrogSteps), int(sb.leapfrogSteps + sb.warmupLeapfrogSteps)) for sa, sb in zip(a.stats, b.stats)]
        if diff:
            c = diff[0]
            it = int(np.argwhere(np.any(a.chains[c] != b.chains[c], axis=1))[0][0])
            out[vname] = "chains %s differ (first: chain %d draw %d max|d| %.3g; leapfrogs compact/every %s)" % (diff, c, it, float(np.max(np.abs(a.chains[c][it] - b.chains[c][it]))), [lf[c] for c in diff[:4]])
        else:
            out[vname] = "identical"
    print(label, type(sampler).__name__, "compacted vs every-chain launches:", out, flush=True)


cen = models.hier_negbin_centred(G, per)
for scale in (0.3, 1.5, 4.0):
    grad_level(cen, "centred", scale)
grad_level(models.hier_negbin(G, per, seed=3), "non-centred", 1.5)
variants = [("default", {}, {}), ("K=1", {}, dict(grad_chains=1)), ("K=2", {}, dict(grad_chains=2)), ("fallback walks (RH_GATHER_V2=0)", {"RH_GATHER_V2": "0"}, {}),
            ("scan walk everywhere", {"RH_GATHER_SCAN": "1"}, {}), ("general tick", {"RH_TICK_FAST": "0"}, {}), ("no value-free", {"RH_VALUE_FREE": "0"}, {})]
for smp in (R.NUTSSampler(4), R.EHMCSampler(16, 2)):
    chain_level(cen, "centred", smp, variants)
chain_level(cen, "centred 23 chains", R.NUTSSampler(4), variants[:1], nchains=23)


def company_matrix(spec, label, sampler, nchains=11, watch=(7, 8)):
    """chains `watch` in different company, compacted and not: which runs agree with which (equivalence classes of the draws)"""
    m = R.Model(spec, device=0, **fast)
    cfg = lambda: R.make_config(4, 10, sampler, R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK, gradSplits=16)
    seeds = [5200 + c for c in range(nchains)]
    subsets = {"all": list(range(nchains)), "alone": None, "4..10": list(range(4, nchains)), "0..8": list(range(0, 9)), "7,8": [7, 8], "8,7 (swapped)": [8, 7], "6,7,8,9": [6, 7, 8, 9]}
    res = {w: {} for w in watch}
    for mode, env in (("compact", {}), ("every", {"RH_COMPACT": "0"})):
        setenv(env)
        try:
            for sname, sub in subsets.items():
                for w in watch:
                    ids = [w] if sub is None else sub
                    if w not in ids:
                        continue
                    key = (mode, sname)
                    if sub is not None and key in res.get("_cache", {}):
                        ch = res["_cache"][key]
                    else:
                        ch = m.sample(cfg(), seeds=[seeds[c] for c in ids]).chains
                        if sub is not None:
                            res.setdefault("_cache", {})[key] = ch
                    res[w]["%s/%s" % key] = ch[ids.index(w)]
        finally:
            clearenv(env)
    m.close()
    for w in watch:
        classes = []
        for name, arr in res[w].items():
            for cl in classes:
                if np.array_equal(cl[0], arr):
                    cl[1].append(name); break
            else:
                classes.append((arr, [name]))
        print(label, type(sampler).__name__, "chain", w, "-> %d distinct results:" % len(classes), [cl[1] for cl in classes], flush=True)


company_matrix(cen, "centred", R.NUTSSampler(4))
company_matrix(cen, "centred", R.EHMCSampler(16, 2))
