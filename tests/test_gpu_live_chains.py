"""Live-chain compaction of the tick engine (rh_compact_kernel + the `list` / `nlive` arguments of the gradient and tick kernels,
csrc/device/rh_engine.hip.h "live chains"): under the reference's dynamic samplers -- EHMCSampler is DefaultConfig's
(sampler/Sampler.scala:17-27, sampler/EHMC.scala:15-61), NUTS is the one BASELINE.json names -- the chains of a run end their
trajectories and iterations at different launches; the gradient launches and ticks serve the listed chains only.

What must hold:
  * a chain's draws do not change by a bit whether the launches are compacted or serve every chain every time (RH_COMPACT=0), on
    each of the row-streaming kernels (plain VALU, fp64 MFMA GLM, gather mode);
  * they do not depend on the other chains of the run either (how many there are, which ones share its chain group), given the row
    split count -- which is what makes sharded runs reproduce the unsharded one;
  * the compacted schedule spends its launches on live chains: far fewer chain slots are computed than launches x chains;
  * and, before any of this product-against-product evidence, the compacted engine agrees with the ORACLE's chains."""
import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models

pytestmark = pytest.mark.gpu


def _run(model, cfg, seeds, pieces=None):
    s = R.Sampler(model, cfg, seeds)
    s.warmup()
    s.timing(reset=True)
    for n in (pieces or [cfg.iterations]):
        s.run(n)
    tim = s.timing()
    stats, mass = s.stats()
    out = dict(draws=s.draws(), mass=mass, tim=tim,
               stats=[(st.leapfrogSteps, st.warmupLeapfrogSteps, st.stepSize, st.meanAcceptProb, st.gradientEvaluations) for st in stats])
    s.close()
    return out


def _cases():
    fast = dict(fp_contract=True, factor_outputs=True)
    return [
        ("linreg/plain VALU kernel", lambda: models.linreg(n=70_001, k=3), fast, "rh_grad_kernel"),
        ("logistic/fp64 MFMA kernel", lambda: models.logistic(n=66_000, k=20, seed=5), fast, "rh_grad_glm_kernel"),
        ("hier_negbin/gather kernel", lambda: models.hier_negbin(700, 100, seed=3), fast, "rh_grad_gather_kernel"),
        # cfg 5's centred form: a second row target -- the lifted prior, ONE row per group -- walked by the segmented-scan path, every
        # split ragged (the shape on which compacted and uncompacted launches disagreed until the ragged tile lost its `if (live)`)
        ("hier_negbin_centred/gather kernel, two row targets", lambda: models.hier_negbin_centred(700, 100), fast, "rh_grad_gather_kernel"),
    ]


@pytest.mark.parametrize("sampler", ["ehmc", "nuts"])
@pytest.mark.parametrize("case", range(4))
def test_compaction_leaves_every_chain_bit_identical(case, sampler, monkeypatch):
    name, mk, build, kernel = _cases()[case]
    spec = mk()
    m = R.Model(spec, device=0, **build)
    smp = R.EHMCSampler(64, 2) if sampler == "ehmc" else R.NUTSSampler(5)
    cfg = R.make_config(6, 25, smp, R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(8, 1.5, 4, 4), engine=_capi.ENGINE_TICK)
    seeds = [4100 + c for c in range(37)]         # ragged against every chain-group size (4, 8, 16 x 4)
    live = _run(m, cfg, seeds)                    # default: compacted launches
    assert live["tim"]["dominant_kernel"] == kernel, name
    pieces = _run(m, cfg, seeds, pieces=[1, 3, 2])
    monkeypatch.setenv("RH_COMPACT", "0")
    every = _run(m, cfg, seeds)                   # every launch serves every chain (paused ones included), same row splits
    monkeypatch.delenv("RH_COMPACT")
    for got in (live, pieces):
        assert np.array_equal(got["draws"], every["draws"]), name
        assert np.array_equal(got["mass"], every["mass"]) and got["stats"] == every["stats"], name
    # the compacted schedule computes the slots it needs (+ group padding), the other one launches x chains of them
    assert live["tim"]["launches"] == every["tim"]["launches"]
    assert live["tim"]["chain_slots"] == live["tim"]["density_evals"] < every["tim"]["chain_slots"] == every["tim"]["launches"] * len(seeds), name
    m.close()


@pytest.mark.parametrize("case", range(4))
def test_a_chain_does_not_depend_on_its_neighbours(case):
    """the same seeds as chains 0..4 of a larger run, alone and in another order of company: identical draws, given the split count"""
    name, mk, build, _ = _cases()[case]
    spec = mk()
    m = R.Model(spec, device=0, **build)
    cfg = lambda: R.make_config(5, 20, R.NUTSSampler(5), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(8, 1.5, 4, 4), engine=_capi.ENGINE_TICK,
                                gradSplits=16)
    seeds = [5200 + c for c in range(23)]
    whole = _run(m, cfg(), seeds)
    part = _run(m, cfg(), seeds[3:8])
    mixed = _run(m, cfg(), seeds[5:8] + seeds[:2] + seeds[20:])
    assert np.array_equal(part["draws"], whole["draws"][3:8]), name
    assert np.array_equal(mixed["draws"][:3], whole["draws"][5:8]) and np.array_equal(mixed["draws"][3:5], whole["draws"][:2]), name
    assert np.array_equal(mixed["draws"][5:], whole["draws"][20:]), name
    m.close()


@pytest.mark.parametrize("case", [2, 3])
def test_gradient_through_a_compacted_list_does_not_depend_on_the_slot(case, monkeypatch):
    """The sharpest form of the statement above, at the gradient level: (logp, gradient) of a chain served through a compacted list
    (RH_EVAL_LIVE: another slot of its chain group, other company, padded groups) equals the identity-list launch bit for bit -- at
    |q| ~ 4 too, where single rows dominate the sums and one row's ulp survives them.  Until round 6's explicit-fusion row code the
    compiler fused the K inlined copies of row() differently and this failed by one ulp for a chain whose slot changed
    (profiles/r6_parity/live_diag.txt)."""
    name, mk, build, _ = _cases()[case]
    spec = mk()
    m = R.Model(spec, device=0, **build)
    nc = 23
    for scale in (0.3, 4.0):
        qs = np.random.default_rng(11).normal(size=(nc, spec.n_params)) * scale      # (the points of tools/r6_live_diag.py)
        lp, g = m.density_batch(qs, engine=_capi.ENGINE_TICK, grad_splits=64)
        ref = np.concatenate([lp[:, None], g], axis=1)
        for sub in ([7], [7, 8], [3, 7, 8], [1, 7, 8, 9, 10], list(range(0, 23, 2)), list(range(1, 23, 2)), list(range(5, 23))):
            monkeypatch.setenv("RH_EVAL_LIVE", ",".join(str(c) for c in sub))
            lp2, g2 = m.density_batch(qs, engine=_capi.ENGINE_TICK, grad_splits=64)
            monkeypatch.delenv("RH_EVAL_LIVE")
            got = np.concatenate([lp2[:, None], g2], axis=1)
            for c in sub:
                assert np.array_equal(got[c], ref[c], equal_nan=True), (name, scale, sub, c, np.flatnonzero(got[c] != ref[c])[:4])
    m.close()


def test_compacted_nuts_and_ehmc_against_the_oracle():
    """oracle first: the compacted tick engine's chains against oracle/sampler.c (tame dynamics: a static step, identity mass, short
    trees, so that the rounding of the row sums does not grow)"""
    from tests import oracle_lib as O
    from tests.test_gpu_parity import _oracle_cfg
    spec = models.linreg(n=70_001, k=3)
    m = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT, grad_chains=4)
    seeds = [6300 + c for c in range(11)]
    for smp in (R.NUTSSampler(3), R.EHMCSampler(8, 2)):
        cfg = R.make_config(3, 3, smp, R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
        got = _run(m, cfg, seeds)
        for c in (0, 5, 10):
            want, _, _ = O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), seeds[c])
            np.testing.assert_allclose(got["draws"][c], want, rtol=1e-9, atol=1e-11, err_msg="%s, chain %d" % (type(smp).__name__, c))
    m.close()


@pytest.mark.parametrize("case", [1, 2])
@pytest.mark.parametrize("sampler", ["hmc", "ehmc"])
def test_gradient_only_requests_leave_every_chain_bit_identical(case, sampler, monkeypatch):
    """Mid-trajectory gradient requests skip what only the log-density needs (row_g() / elem_g(): csrc/emit.cpp value_only; the
    potential of such a step is overwritten before it is read, sampler/LeapFrog.scala:175-184).  Same chains, bit for bit, as with
    every launch computing the value (RH_VALUE_FREE=0) -- on the MFMA GLM kernel (softplus half of the logit link) and in gather mode."""
    name, mk, build, kernel = _cases()[case]
    spec = mk()
    m = R.Model(spec, device=0, **build)
    assert "row_g(" in m.hip_source and "HAS_VALUE_ONLY = true" in m.hip_source, name
    smp = R.HMCSampler(6) if sampler == "hmc" else R.EHMCSampler(32, 2)
    cfg = R.make_config(5, 20, smp, R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(6, 1.5, 3, 3), engine=_capi.ENGINE_TICK)
    seeds = [7300 + c for c in range(21)]
    lean = _run(m, cfg, seeds)
    assert lean["tim"]["dominant_kernel"] == kernel
    monkeypatch.setenv("RH_VALUE_FREE", "0")
    full = _run(m, cfg, seeds)
    monkeypatch.delenv("RH_VALUE_FREE")
    assert np.array_equal(lean["draws"], full["draws"]) and np.array_equal(lean["mass"], full["mass"]) and lean["stats"] == full["stats"], name
    m.close()


def test_big_mode_gather_tick_fast_path_is_bit_identical_to_the_general_path(monkeypatch):
    """rh_tick_kernel's fast path for gather-mode models in big mode (one loop per element instead of six vector passes; it does not
    store pend_g / Bg -- the invariant: whichever tick follows overwrites both before reading them, rh_engine.hip.h) against the
    general path (RH_TICK_FAST=0, another build of the same model): adapted diagonal mass, 13 chains (a ragged last chain group),
    704 parameters = 11 slots of 64 (not a multiple of the 8 slots the loop keeps in flight), piecewise runs."""
    spec = models.hier_negbin(700, 100, seed=3)
    build = dict(fp_contract=True, factor_outputs=True)
    cfg = R.make_config(5, 14, R.HMCSampler(5), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(4, 1.5, 2, 2), engine=_capi.ENGINE_TICK)
    seeds = [8400 + c for c in range(13)]
    m = R.Model(spec, device=0, **build)
    assert "#define RH_BIGN 1" in m.hip_source and "#define RH_SLOTS 11" in m.hip_source
    fast = _run(m, cfg, seeds, pieces=[2, 3])
    m.close()
    monkeypatch.setenv("RH_TICK_FAST", "0")
    m0 = R.Model(spec, device=0, **build)
    assert "#define RH_TICK_FAST 0" in m0.hip_source
    general = _run(m0, cfg, seeds, pieces=[2, 3])
    m0.close()
    monkeypatch.delenv("RH_TICK_FAST")
    assert np.array_equal(fast["draws"], general["draws"]) and np.array_equal(fast["mass"], general["mass"]) and fast["stats"] == general["stats"]
