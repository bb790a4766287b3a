"""The control flow of rh_gather_walk_a (csrc/device/rh_engine.hip.h, round 5) restated lane for lane in numpy and run over random group
structures: rows sorted by group, every non-empty group >= 64 rows, EMPTY groups anywhere, ragged split ends, splits cut at group
borders.  What the device code decides on the scalar unit -- which group is open (A), which comes next (B), when A is complete, which
buffer holds the ragged tile -- is checked against a direct per-group sum: every non-empty group of every split is flushed exactly once
with the sum of exactly its rows (in tile order), empty groups are never written.  (The arithmetic itself runs on the device against the
oracle: tests/test_gpu_parity.py, tests/test_gpu_baseline_sizes.py, tests/test_gpu_strict_gather.py.)"""
import numpy as np
import pytest


def walk(goff, g0, g1, sv_of_row):
    """one split [g0, g1) of one chain: returns {group: sum} in flush order, emulating the two-slot rolling pipeline"""
    r0, r1 = goff[g0], goff[g1]
    out = {}
    if r0 >= r1:
        return out
    lanes = np.arange(64)
    gA = g0
    while gA < g1 - 1 and goff[gA + 1] == goff[gA]:
        gA += 1
    endA = goff[gA + 1]
    gB = gA + 1
    while gB < g1 and goff[gB + 1] == goff[gB]:
        gB += 1
    accA = np.zeros(64); accB = np.zeros(64)
    rlast = r1 - 1

    def load_tile(tb):
        r = tb + lanes
        return np.minimum(r, rlast)            # the rows a slot holds (clamped into the split)

    def close_tile(tend):
        nonlocal gA, gB, endA, accA, accB
        if endA <= tend:
            assert gA not in out, "a group was flushed twice"
            out[gA] = accA.copy()              # (the wave reduction is the subject of test_device_math_host.py)
            accA, accB = accB, np.zeros(64)
            gA = gB
            endA = goff[gA + 1] if gA < g1 else 0x7fffffff
            if gB < g1:
                gB += 1
            while gB < g1 and goff[gB + 1] == goff[gB]:
                gB += 1

    c = [load_tile(r0), load_tile(r0 + 64)]
    tb = r0
    done = False
    while not done:
        for u in range(2):
            if tb + 64 > r1:
                done = True
                break
            inA = tb + lanes < endA
            rows = c[u]
            assert np.array_equal(rows, tb + lanes), "the slot does not hold this tile's rows"
            sv = sv_of_row[rows]
            accA += np.where(inA, sv, 0.0); accB += np.where(inA, 0.0, sv)
            c[u] = load_tile(tb + 128)
            close_tile(tb + 64)
            tb += 64
    if tb < r1:
        u = ((tb - r0) >> 6) & 1
        inA = tb + lanes < endA
        live = tb + lanes < r1
        rows = c[u]
        assert np.array_equal(rows[live], (tb + lanes)[live])
        sv = np.where(live, sv_of_row[rows], 0.0)
        accA += np.where(inA, sv, 0.0); accB += np.where(inA, 0.0, sv)
        close_tile(r1)
    return out


@pytest.mark.parametrize("seed", range(40))
def test_group_major_walk_flushes_every_group_once_with_its_own_rows(seed):
    rng = np.random.default_rng(seed)
    ngroups = int(rng.integers(1, 40))
    sizes = rng.integers(64, 400, ngroups)
    sizes[rng.random(ngroups) < 0.25] = 0                       # empty groups anywhere (also first / last)
    if seed % 5 == 0:
        sizes = np.where(sizes > 0, 64 * rng.integers(1, 4, ngroups), 0)   # groups that end exactly on tile borders
    if not sizes.any():
        sizes[int(rng.integers(0, ngroups))] = 64
    goff = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    nrows = int(goff[-1])
    sv = rng.standard_normal(nrows)
    nsplit = int(rng.integers(1, 6))
    # the host's cut (engine.cpp GatherBufs::build): balanced by rows, at group borders only
    gs = [0]
    g = 0
    for s in range(1, nsplit):
        want = nrows * s // nsplit
        while g < ngroups and goff[g] < want:
            g += 1
        gs.append(g)
    gs.append(ngroups)
    seen = {}
    for s in range(nsplit):
        for grp, lanesum in walk(goff, gs[s], gs[s + 1], sv).items():
            assert grp not in seen
            seen[grp] = lanesum
    for grp in range(ngroups):
        lo, hi = goff[grp], goff[grp + 1]
        if lo == hi:
            assert grp not in seen                              # an empty group is never stored (its sum stays the buffer's zero)
            continue
        # lane l of the accumulator holds the rows r of the group with (r - r0_of_split) % 64 == l, in ascending order
        assert grp in seen, grp
        assert abs(seen[grp].sum() - sv[lo:hi].sum()) <= 1e-12 * np.abs(sv[lo:hi]).sum()
        split = max(s for s in range(nsplit) if gs[s] <= grp)
        r0 = goff[gs[split]]
        want = np.zeros(64)
        for r in range(lo, hi):
            want[(r - r0) % 64] += sv[r]
        assert np.array_equal(seen[grp], want), grp            # bit for bit: the same additions in the same order
