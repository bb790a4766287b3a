"""Cross-compile (no device needed) the code objects the GPU probes of tools/ load, variant by variant, into the in-tree kernel cache:
the generated source does not depend on the row count, so a probe run on the GPU box finds them and spends no GPU time in hiprtc.
usage: python tools/prebuild_probes.py cfg5|cfg4|cfg2 ["ENV=V ENV=V" ...]   (one lowering per environment string; "" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rainier_amd import _capi, models

what, envs = sys.argv[1], (sys.argv[2:] or [""])
for env in envs:
    kv = dict(x.split("=", 1) for x in env.split())
    K = int(kv.pop("K", 0))
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    t = time.time()
    fast = dict(fp_contract=True, factor_outputs=True)
    if what == "cfg5":
        spec = models.hier_negbin(10_000, 1); opts = _capi.compile_opts(grad_chains=K, **fast)
    elif what == "cfg4":
        spec = models.logistic(n=8, k=50); opts = _capi.compile_opts(grad_chains=K, **fast)
    else:
        spec = models.linreg(n=8, k=3); opts = _capi.compile_opts(grad_chains=K, **fast)
    src, rep = _capi.lower_report(spec.rir, opts)
    fit = {k[1]: (v["vgprs"], v["vgpr_spills"], v["fit"]) for k, v in rep["kernels"].items() if k[0] == "base" and ("grad" in k[1] or "tick" in k[1])}
    print("%-6s %-40s %.1fs %s %s" % (what, env, time.time() - t, rep["shape"], fit), flush=True)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
