"""Mirror of the reference's compute/RealTest.scala (rainier-test/.../compute/RealTest.scala:8-56) at the IR boundary:
every expression of that suite, at its nine evaluation points (including 0 and +-Infinity), must agree between
plain-double evaluation ("constant folding"), the math reference where the test names one, the RIR interpreter of the
oracle ("compiled IR"), and -- in tests/test_gpu_parity.py -- the generated HIP code; derivatives are checked against
central differences exactly as the reference does (dx = 1e-5)."""
import math

import numpy as np
import pytest

from rainier_amd import models
from rainier_amd.frontend import Graph
from tests import oracle_lib as O
from tests.realtest_cases import CASES, POINTS, Alg, constant, within_epsilon


def build(fn):
    g = Graph(1, [0])
    return models.ModelSpec("realtest", g.compile([fn(Alg(g), g.param(0))]), [], [0], 1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_realtest_expression_on_oracle(oracle, case):
    name, fn, defined, derivable, reference = case
    d = O.OracleDensity(build(fn))
    checked = 0
    for v in POINTS:
        if defined is not None and not defined(v):
            continue
        const = constant(fn, v)
        if reference is not None:
            assert within_epsilon(const, reference(v)), ("c/ref", name, v, const, reference(v))
        out = d.update(np.array([v]))
        assert within_epsilon(const, out[0]), ("ev/ir", name, v, const, out[0])
        if (derivable is None or derivable(v)) and not math.isinf(v):
            dx = 10e-6
            num = (constant(fn, v + dx) - constant(fn, v - dx)) / (dx * 2)
            assert within_epsilon(num, out[1]), ("numDiff/diffCompiled", name, v, num, out[1])
        checked += 1
    assert checked >= 2
