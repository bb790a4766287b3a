import sys, os, time, json
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import models, _capi
G, per, chains = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nuts = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t = time.time(); spec = models.hier_negbin(G, per); print("build %.1fs rir %d bytes" % (time.time() - t, len(spec.rir)), flush=True)
t = time.time(); m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True, grad_chains=K); print("model %.1fs" % (time.time() - t), flush=True)
# closed form check (numpy) at one point
q = np.random.default_rng(0).normal(size=(2, spec.n_params)) * 0.3
t = time.time(); lp, g = m.density_batch(q); print("density_batch %.3fs" % (time.time() - t), flush=True)
v, crow, gid, x0, x1 = spec.columns[1:]
gi = gid.astype(int); nf = 10.0
for c in range(2):
    mm, s, b0, b1 = q[c, :4]; z = q[c, 4:]
    eta = 10 * mm + np.exp(s) * z[gi] + b0 * x0 + b1 * x1
    p = 1 / (1 + nf * np.exp(-eta))
    ll = crow + nf * np.log(1 - p) + v * np.log(p)
    ref = (-0.5 * mm * mm - models.HALF_LOG_2PI) + (s - np.exp(s)) + (-0.5 * b0 * b0 - models.HALF_LOG_2PI) + (-0.5 * b1 * b1 - models.HALF_LOG_2PI) \
        + np.sum(-0.5 * z * z - models.HALF_LOG_2PI) + ll.sum()
    w = v * (1 - p) - nf * p            # d ll / d eta
    gz = -z + np.exp(s) * np.bincount(gi, weights=w, minlength=G)
    print("logp rel err %.2e" % abs((lp[c] - ref) / ref), "grad_z max abs err %.2e" % np.max(np.abs(g[c, 4:] - gz)),
          "grad_b0 err %.2e" % abs(g[c, 2] - (-b0 + np.sum(w * x0))), flush=True)
# RH_PROBE_SPLITS="8,16,24": row splits per chain group to time (0 = the engine's default)
for splits in [int(x) for x in os.environ.get("RH_PROBE_SPLITS", "0").split(",")]:
    cfg = R.make_config(4, 0, R.HMCSampler(8), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner(), gradSplits=splits)
    s = R.Sampler(m, cfg, list(range(chains)))
    s.warmup(); s.timing(reset=True)
    t = time.time(); s.run(4); dt = time.time() - t
    tim = s.timing()
    print(json.dumps({"G": G, "per": per, "chains": chains, "K": K, "splits": splits, "env": {k: v for k, v in os.environ.items() if k.startswith("RH_")},
                      "s_per_tick": dt / 32, "row_chain_evals_per_s": G * per * chains * 32 / dt,
                      "grad_kernel_ms": tim["kernel_ms"] / max(1, tim["launches"]), "all_ms": tim["total_ms"] / 32, "kernel": tim["dominant_kernel"]}), flush=True)
    s.close()

if nuts:
    cfg = R.make_config(nuts, nuts, R.NUTSSampler(10))
    s2 = R.Sampler(m, cfg, [2000 + c for c in range(chains)])
    t = time.time(); s2.warmup(); tw = time.time() - t
    t = time.time(); s2.run(nuts); dt = time.time() - t
    st, _ = s2.stats()
    steps = sum(x.leapfrogSteps for x in st)
    d = s2.draws()
    ess = min(e for _, e in R.diagnostics(d[:, :, :8])) if nuts >= 4 else None
    print(json.dumps({"nuts_iters": nuts, "warmup_s": tw, "run_s": dt, "leapfrog_steps_per_s": steps / dt, "mean_steps_per_iter": steps / (nuts * chains),
                      "mean_accept": float(np.mean([x.meanAcceptProb for x in st])), "ess_min_first8_per_s": ess / dt if ess else None}))
