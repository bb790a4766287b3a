"""The seeded random models of the host-side emitter fuzz (tests/fuzz_models.py, tests/test_emitter_host.py) through the KERNELS:
rh_model_create's data-dependent passes (constant lifting, table-prior lifting incl. the centred form, column canonicalisation,
gradient re-derivation, re-association, slot rolling), hiprtc, then (logp, gradient) from the chain-per-wavefront density kernel AND
from the tick engine's row-streaming kernels (generic / gather) + the tick combine, against the oracle's interpreter on the
ORIGINAL program: 1e-12 * sum|term| per output, both math modes."""
import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi
from tests import oracle_lib as O
from tests.fuzz_models import eight_slot_model, table_prior_model

pytestmark = pytest.mark.gpu

FAST = dict(fp_contract=True, factor_outputs=True)
STRICT = dict(math_mode=_capi.MATH_STRICT)
TABLE_SEEDS, EIGHT_SLOT_SEEDS, EIGHT_SLOT_ROWS = range(12), range(24), 4096


def _against_oracle(spec, model, qs, tol, engines):
    d = O.OracleDensity(spec)
    worst = 0.0
    for engine in engines:
        lp, g = model.density_batch(np.asarray(qs), engine=engine)
        for c, q in enumerate(qs):
            ref, ab = d.update_both(np.asarray(q, dtype=np.float64))
            got = np.concatenate([[lp[c]], g[c]])
            ratio = np.abs(got - ref) / (ab + 1e-300)
            assert np.all((ratio <= tol) | (np.isnan(got) & np.isnan(ref))), (spec.name, engine, float(np.nanmax(ratio)))
            worst = max(worst, float(np.nanmax(ratio)))
    return worst


@pytest.mark.parametrize("seed", TABLE_SEEDS)
def test_random_table_priors_on_the_device(seed):
    spec, qs, mode = table_prior_model(seed)
    if not qs:
        pytest.skip("no finite evaluation point")
    for opts in (STRICT, FAST):
        m = R.Model(spec, device=0, **opts)
        gather = "#define RH_HAS_GATHER 1\n" in m.hip_source
        assert gather == (mode != 3 or opts is FAST)       # the centred prior is lifted in fast builds only
        # gather-mode models run on the tick engine only; the generic ones on both
        _against_oracle(spec, m, qs, 1e-12, [_capi.ENGINE_TICK] if gather else [_capi.ENGINE_CHAIN, _capi.ENGINE_TICK])
        m.close()


@pytest.mark.parametrize("seed", EIGHT_SLOT_SEEDS)
def test_random_eight_slot_programs_on_the_device(seed):
    spec, qs = eight_slot_model(seed, n=EIGHT_SLOT_ROWS)              # enough rows for several row splits
    if not qs:
        pytest.skip("no finite evaluation point")
    for opts in (STRICT, FAST):
        m = R.Model(spec, device=0, **opts)
        _against_oracle(spec, m, qs, 1e-12, [_capi.ENGINE_CHAIN, _capi.ENGINE_TICK])
        m.close()
