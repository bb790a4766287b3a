"""Extended seeds for the host-side fuzz generators of tests/test_emitter_host.py (the suite keeps ~100 of them): every generator that
takes a seed, run on [LO, HI) in this process; failures are printed and counted, nothing stops.  No GPU needed (the generated code is
compiled for the host and compared with the oracle on the original program).
    python tools/host_fuzz_more.py 2000 2100
(The harness lifts once, as rh_model_create does: an already lifted program is lowered with the loader's lifting passes off.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest  # noqa: E402
import tests.test_emitter_host as T  # noqa: E402

NAMES = ["test_random_eight_slot_programs", "test_random_lookup_table_models", "test_random_single_observation_models",
         "test_random_single_observation_models_in_the_reference_text", "test_random_groups_of_single_observation_targets",
         "test_random_hierarchical_models_in_the_reference_text", "test_random_table_priors",
         "test_random_families_that_differ_in_a_parameter", "test_random_glms_through_the_glm_lowering",
         "test_random_initial_chunk_plus_eight_slots", "test_random_reference_text_models"]

if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(lo, hi):
        for nm in NAMES:
            try:
                getattr(T, nm)(seed)
            except pytest.skip.Exception:
                pass
            except BaseException as e:   # noqa: BLE001 -- report and go on
                bad += 1
                print("FAIL", nm, seed, repr(e)[:300], flush=True)
    print("done", lo, hi, "failures", bad, flush=True)
