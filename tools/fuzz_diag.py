"""diagnostic: one eight-slot fuzz model through both engines, per build / row-loop variant / split count"""
import os, sys
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rainier_amd as R
from rainier_amd import _capi
from tests import oracle_lib as O
from tests.fuzz_models import eight_slot_model
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
GU = int(os.environ.get("GU", "0")); GK = int(os.environ.get("GK", "0"))
NP = int(os.environ.get('NP', '3'))
spec, qs = eight_slot_model(seed, n=n, npoints=NP)
d = O.OracleDensity(spec)
for name, opts in (("strict", dict(math_mode=_capi.MATH_STRICT)), ("fast", dict(fp_contract=True, factor_outputs=True))):
    m = R.Model(spec, device=0, grad_unroll=GU, grad_chains=GK, **opts)
    for engine, splits in ((_capi.ENGINE_CHAIN, 0), (_capi.ENGINE_TICK, 0), (_capi.ENGINE_TICK, 1), (_capi.ENGINE_TICK, 3)):
        lp, g = m.density_batch(np.asarray(qs), engine=engine, grad_splits=splits)
        worst = []
        for c, q in enumerate(qs):
            ref, ab = d.update_both(np.asarray(q, dtype=np.float64))
            got = np.concatenate([[lp[c]], g[c]])
            worst.append(np.abs(got - ref) / (ab + 1e-300))
        import re
        print(re.search(r'#define RH_GRAD_U (\d+)', m.hip_source).group(0), re.search(r'#define RH_GRAD_K (\d+)', m.hip_source).group(0), 'nrows', [int(x) for x in spec.nrows])
        print(name, "GU", GU, "GK", GK, "pipeline", os.environ.get("RH_GRAD_PIPELINE", "default"), "engine", engine, "splits", splits, "ratio per output", np.max(np.array(worst), axis=0))
    m.close()
