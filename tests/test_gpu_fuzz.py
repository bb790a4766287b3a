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
from tests.fuzz_models import GPU_FUZZ_CASES, gpu_fuzz_case

pytestmark = pytest.mark.gpu

FAST = dict(fp_contract=True, factor_outputs=True)
STRICT = dict(math_mode=_capi.MATH_STRICT)


def _against_oracle(spec, model, qs, tol, runs):
    d = O.OracleDensity(spec)
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
    for engine, splits in runs:
        lp, g = model.density_batch(np.asarray(qs), engine=engine, grad_splits=splits)
        for c, (ref, ab) in enumerate(refs):
            got = np.concatenate([[lp[c]], g[c]])
            ratio = np.abs(got - ref) / (ab + 1e-300)
            assert np.all((ratio <= tol) | (np.isnan(got) & np.isnan(ref))), (spec.name, engine, splits, c, float(np.nanmax(ratio)))


@pytest.mark.parametrize("kind,seed,kw", GPU_FUZZ_CASES, ids=["%s-%d-%s" % (k, s, "-".join(str(v) for v in kw.values())) for k, s, kw in GPU_FUZZ_CASES])
def test_random_models_on_the_device(kind, seed, kw):
    spec, qs, mode = gpu_fuzz_case(kind, seed, kw)
    if not qs:
        pytest.skip("no finite evaluation point")
    for opts in (STRICT, FAST):
        m = R.Model(spec, device=0, **opts)
        gather = "#define RH_HAS_GATHER 1\n" in m.hip_source
        if kind == "table":
            assert gather == (mode != 3 or opts is FAST)       # the centred prior is lifted in fast builds only
        # gather-mode models run on the tick engine only; the others on both, the tick engine with its default and with 3 row splits
        runs = [(_capi.ENGINE_TICK, 0), (_capi.ENGINE_TICK, 3)] + ([] if gather else [(_capi.ENGINE_CHAIN, 0)])
        _against_oracle(spec, m, qs, 1e-12, runs)
        m.close()
