"""Random model generators shared by the host-side emitter fuzz tests (tests/test_emitter_host.py) and the GPU fuzz tests
(tests/test_gpu_fuzz.py): the same seeded models go through the generated code compiled for the host and through the kernels."""
import numpy as np

from rainier_amd.frontend import Graph
from rainier_amd.models import ModelSpec


def _random_expr(rng, g, leaves, depth):
    """a random smooth-enough expression over `leaves` (domain-safe: logs of 1 + u^2, divisions by 1 + u^2, small exponents)"""
    if depth == 0 or rng.random() < 0.15:
        return leaves[int(rng.integers(len(leaves)))]
    op = int(rng.integers(9))
    a = _random_expr(rng, g, leaves, depth - 1)
    if op <= 1:
        return a + _random_expr(rng, g, leaves, depth - 1)
    if op == 2:
        return a - _random_expr(rng, g, leaves, depth - 1)
    if op == 3:
        return a * _random_expr(rng, g, leaves, depth - 1)
    if op == 4:
        b = _random_expr(rng, g, leaves, depth - 1)
        return a / (b * b + 1.0)
    if op == 5:
        return (a * 0.3).exp()
    if op == 6:
        return (a * a + 1.0).log()
    if op == 7:
        return a ** float(rng.integers(2, 4))
    b = _random_expr(rng, g, leaves, depth - 1)
    return g.lookup(a.compare(b), [a, a + b, b * 0.5], -1)          # a select on a row-level compare


def table_prior_model(seed, npoints=2, per_range=(2, 6)):
    """a Lookup over 65-139 trailing parameters indexed by a column, with the table's prior folded into the data-free target the way
    the reference's front end leaves it (mode = seed % 4: standard | random per-entry shape with per-entry constants | two terms per
    entry | tied to a shared parameter, the centred parameterisation); returns (spec, finite evaluation points, mode)"""
    rng = np.random.default_rng(90000 + seed)
    G, per = int(rng.integers(65, 140)), int(rng.integers(*per_range))
    n, nsh = G * per, int(rng.integers(2, 4))
    P = nsh + G
    site = rng.permutation(np.repeat(np.arange(G), per)).astype(float); x = rng.normal(size=n); y = rng.normal(size=n)
    g = Graph(P, [0, 3])
    th = [g.param(i) for i in range(P)]
    mode = seed % 4
    prior = th[0] * th[0] * -0.5
    for i in range(1, nsh):
        prior = prior + th[i] * th[i] * -0.5
    st, depth = int(rng.integers(1 << 30)), int(rng.integers(1, 4))
    for k in range(G):
        z = th[nsh + k]
        if mode == 0:
            prior = prior + z * z * -0.5
        elif mode == 1:
            c1, c2 = g.const(float(rng.uniform(0.5, 2.0))), g.const(float(rng.normal()))
            prior = prior + _random_expr(np.random.default_rng(st), g, [z, c1, c2, z * c1], depth) + z * z * -0.5
        elif mode == 2:
            prior = prior + z * z * -0.5 - (z * z + 1.0).log() * 0.5
        else:
            prior = prior + (z - th[0]) * (z - th[0]) * -0.5
        if k == G // 2:
            prior = prior + th[1] * float(rng.normal())
    r = g.col(1, 2) - (th[0] + th[1] * g.col(1, 1) + g.lookup(g.col(1, 0), th[nsh:], 0))
    spec = ModelSpec("fuzz_table_prior_%d" % seed, g.compile([prior, r * r * -0.5]), [site, x, y], [0, n], P, {})
    qs = []
    if npoints > 0:
        from tests import oracle_lib as O          # (lazy: build() lowers these specs without touching the oracle)
        d = O.OracleDensity(spec)
        qs = [q for q in rng.normal(size=(3 * npoints, P)) * 0.5 if np.all(np.isfinite(d.update(q)))][:npoints]
    return spec, qs, mode


def eight_slot_model(seed, n=48, npoints=3):
    """a random expression as the per-observation term of an 8-slot Model.observe-shaped target (with a derived column per slot),
    gradient by the authoring DSL; returns (spec, finite evaluation points)"""
    rng = np.random.default_rng(1000 + seed)
    S, P = 8, 4
    cols = []
    for s in range(S):
        x = rng.uniform(-1, 1, n)
        cols += [x, rng.uniform(-1, 1, n), -x]
    g = Graph(P, [3 * S])
    th = [g.param(i) for i in range(P)]
    shape_rng_state = rng.integers(1 << 30)
    val = None
    for s in range(S):
        x, z, mx = g.col(0, 3 * s), g.col(0, 3 * s + 1), g.col(0, 3 * s + 2)
        leaves = th + [x, z, mx * 0.5, g.const(0.7), th[0] * x + th[1], th[2] * z]
        term = _random_expr(np.random.default_rng(shape_rng_state), g, leaves, 4) + th[3] * z      # the same shape in every slot
        val = term if val is None else val + term
    val = val + th[0] * th[1] * 8.0                                     # a shared, parameter-only term (8 copies merged)
    spec = ModelSpec("fuzz_%d" % seed, g.compile([val]), cols, [n], P, {})
    qs = []
    if npoints > 0:
        from tests import oracle_lib as O
        d = O.OracleDensity(spec)
        qs = [q for q in rng.normal(size=(2 * npoints, P)) * 0.6 if np.all(np.isfinite(d.update(q)))][:npoints]
    return spec, qs


# the cases tests/test_gpu_fuzz.py runs through the kernels (tests/test_emitter_host.py cross-compiles the same ones for gfx950)
GPU_FUZZ_CASES = ([("table", s, dict(npoints=5)) for s in range(8)] +                             # 2-5 rows per group: segmented scan
                  [("table", s, dict(npoints=5, per_range=(64, 80))) for s in range(4)] +          # >= 64 rows per group: two-accumulator walk
                  [("slots", s, dict(n=(4096, 1000, 70)[s % 3], npoints=11)) for s in range(12)])  # full / ragged / tiny row counts, ragged chain groups


def gpu_fuzz_case(kind, seed, kw):
    """-> (spec, evaluation points, mode | None)"""
    if kind == "table":
        return table_prior_model(seed, **kw)
    spec, qs = eight_slot_model(seed, **kw)
    return spec, qs, None
