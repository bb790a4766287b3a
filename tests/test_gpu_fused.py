"""The mid-trajectory leapfrog update fused into the gradient launch (rh_grad_fused_kernel, csrc/device/rh_engine.hip.h): static
HMC in the sampling phase issues ONE launch per leapfrog step instead of gradient + tick -- the update runs as the PROLOGUE of the
next gradient launch, on the partial sums the previous launch left, and the advanced chains live in per-chain records until the
tick that ends the trajectory.  The chains must not change by a bit against the two-launch schedule (RH_FUSE=0) --
the prologue sums the row splits in the tick kernel's order and applies `twoFullSteps` (sampler/LeapFrog.scala:175-184) as the
automaton spells it -- and, through it, against everything the tick engine is already checked against:
tests/test_gpu_parity.py::test_tick_engine_matches_chain_engine_and_oracle runs static HMC on the tick engine, i.e. through this
path, against the oracle's chains and the chain-per-wavefront engine."""
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
    out = (s.draws(), mass, [(st.leapfrogSteps, st.stepSize, st.meanAcceptProb, st.gradientEvaluations) for st in stats], tim["dominant_kernel"])
    s.close()
    return out


@pytest.mark.parametrize("build,tuner", [
    (dict(fp_contract=True, factor_outputs=True, grad_chains=8), "identity"),       # bench.py's cfg-2 build
    (dict(fp_contract=True, factor_outputs=True, grad_chains=8), "diag"),           # non-identity mass: q += eps * (p * M)
    (dict(math_mode=_capi.MATH_STRICT, grad_chains=4), "identity"),                 # JVM-faithful arithmetic, 16 lanes per chain
    (dict(fp_contract=True, factor_outputs=True, grad_chains=2), "diag")])
def test_fused_launches_leave_the_chains_bit_identical(build, tuner, monkeypatch):
    spec = models.linreg(n=70_001, k=3)           # ragged rows; tick engine (>= 65 536 rows)
    m = R.Model(spec, device=0, **build)
    mt = R.IdentityMassMatrixTuner() if tuner == "identity" else R.DiagonalMassMatrixTuner(10, 1.5, 5, 5)
    cfg = R.make_config(9, 40, R.HMCSampler(7), R.DualAvgTuner(0.8), mt, engine=_capi.ENGINE_TICK)
    seeds = [500 + c for c in range(21)]          # 21 chains: a ragged last chain group for every K
    fused = _run(m, cfg, seeds)                       # the default schedule
    assert fused[3] == "rh_grad_fused_kernel"
    pieces = _run(m, cfg, seeds, pieces=[2, 1, 6])    # rh_sampler_run called piecewise: every call starts a fresh lock-step schedule
    monkeypatch.setenv("RH_FUSE", "0")
    plain = _run(m, cfg, seeds)
    monkeypatch.delenv("RH_FUSE")
    assert plain[3] == "rh_grad_kernel"
    for got in (fused, pieces):
        assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1]) and got[2] == plain[2]
    assert all(st[0] == 9 * 7 for st in fused[2])


def test_fused_schedule_against_the_oracle():
    # the fused launches themselves against the ORACLE's chains (tame dynamics: static step, identity mass, so that rounding does
    # not grow), before anything is compared product against product
    from tests import oracle_lib as O
    from tests.test_gpu_parity import _oracle_cfg
    spec = models.linreg(n=70_001, k=3)
    cfg = R.make_config(3, 0, R.HMCSampler(5), R.StaticStepSize(2e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    seeds = [900 + c for c in range(9)]
    for build in (dict(math_mode=_capi.MATH_STRICT, grad_chains=4), dict(fp_contract=True, factor_outputs=True, grad_chains=8)):
        got = _run(R.Model(spec, device=0, **build), cfg, seeds)
        assert got[3] == "rh_grad_fused_kernel"
        for c in (0, 8):
            want, _, _ = O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), seeds[c])
            np.testing.assert_allclose(got[0][c], want, rtol=1e-9, atol=1e-11, err_msg="fused schedule differs from the ORACLE (%s, chain %d)" % (build, c))
