"""Multi-GPU path, as far as one MI355X can show it.

  * rh_sample_multi (sharding behind the C ABI): the SAME chains cut into 1, 2, 3 and 5 shards -- every shard here runs on
    device 0 -- give bit-identical draws, stats and mass matrices: the result is a function of the global chain id only.
  * shard_seeds + per-rank samplers (what bench.py does per process) reproduce the unsharded run on real engine draws.
  * the device-resident draws -> torch tensor -> RCCL all-gather path of bench.py (gather_draws_from_device), executed on
    a world-size-1 NCCL(=RCCL) process group.
"""
import os
import socket

import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models
from rainier_amd import distributed as D

pytestmark = pytest.mark.gpu


def _cfg(engine=_capi.ENGINE_AUTO):
    return R.make_config(12, 30, R.EHMCSampler(64), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(10, 1.5, 5, 5), engine=engine)


@pytest.mark.parametrize("builder,engine,strict", [
    (lambda: models.eight_schools(), _capi.ENGINE_AUTO, dict(math_mode=_capi.MATH_STRICT)),
    (lambda: models.linreg(n=70000, k=3), _capi.ENGINE_TICK, dict(math_mode=_capi.MATH_STRICT)),
    # the bench build with automatic row splits: the split count follows the TOTAL chain count, not the shard's
    (lambda: models.linreg(n=600000, k=3), _capi.ENGINE_AUTO, dict(fp_contract=True, factor_outputs=True, grad_chains=8))])
def test_rh_sample_multi_is_independent_of_the_shard_count(builder, engine, strict):
    spec = builder()
    seeds = [7000 + c for c in range(11)]                 # 11 chains: uneven shards
    m0 = R.Model(spec, device=0, **strict)
    base = m0.sample(_cfg(engine), seeds=seeds)
    for nshards in (1, 2, 3, 5, 16):
        ms = [m0] + [m0.clone(0) for _ in range(min(nshards, 3) - 1)]    # rh_model_clone: code object reused, columns copied device to device
        ms = (ms * nshards)[:nshards]                     # several shards may share a model handle (calls serialise on it)
        tr = R.sample_multi(ms, _cfg(engine), seeds)
        assert np.array_equal(tr.chains, base.chains), nshards
        assert np.array_equal(tr.mass, base.mass)
        assert [s.leapfrogSteps for s in tr.stats] == [s.leapfrogSteps for s in base.stats]
        assert [s.stepSize for s in tr.stats] == [s.stepSize for s in base.stats]


def test_a_clone_takes_the_lowering_decisions_of_its_source():
    """rh_model_clone carries the lowering state (row-target count, which unrolls were the engine's choice): a clone that builds a
    sampler-kernel variant by itself -- NUTS asked of the CLONE first -- lightens the shape like its source would, keeps the same
    engines usable and returns the same chains (a clone with n_row_targets_hint = 0 accepted an unfit tick kernel)."""
    spec = models.hier_negbin(6, 7, seed=6)              # the round-3 reproducer: its first shapes do not fit
    seeds = [7100 + c for c in range(5)]
    for engine in (_capi.ENGINE_AUTO, _capi.ENGINE_TICK):
        cfg = R.make_config(6, 12, R.NUTSSampler(5), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(4, 1.5, 2, 2), engine=engine)
        m0 = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
        c = m0.clone(0)
        first = c.sample(cfg, seeds=seeds)                # the clone builds / loads the NUTS variant on its own
        base = m0.sample(cfg, seeds=seeds)
        e0, e1 = m0.engines(), c.engines()
        assert {k: e0[k] for k in ("chain", "tick", "density")} == {k: e1[k] for k in ("chain", "tick", "density")}, (e0, e1)
        assert np.array_equal(first.chains, base.chains)
        tr = R.sample_multi([c, m0], cfg, seeds)
        assert np.array_equal(tr.chains, base.chains)
        c.close(); m0.close()


def test_rh_sample_multi_rejects_bad_arguments():
    m = R.Model(models.funnel(10), device=0)
    m2 = R.Model(models.eight_schools(), device=0)
    m3 = R.Model(models.normal_1d(), device=0)
    with pytest.raises(R.RainierHipError):
        R.sample_multi([m, m3], R.HMC(5, 5, 2), [1, 2, 3])      # different programs (nvars differ)
    tr = R.sample_multi([m, m], R.HMC(5, 5, 2), [1])            # more shards than chains: the surplus shards stay idle
    assert tr.chains.shape == (1, 5, 10)
    del m2


def test_per_rank_samplers_with_global_seeds_match_the_unsharded_run():
    """bench.py's protocol on real draws: rank r owns the global chains [r*cpr, (r+1)*cpr) with seeds base + global id."""
    spec = models.linreg(n=70000, k=3)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True, grad_chains=8)
    cpr, world = 6, 3
    cfg = R.make_config(6, 10, R.HMCSampler(4), R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    whole = m.sample(cfg, seeds=[1000 + g for g in range(cpr * world)]).chains
    for rank in range(world):
        part = m.sample(cfg, seeds=D.shard_seeds(1000, cpr, rank)).chains
        assert np.array_equal(part, whole[rank * cpr:(rank + 1) * cpr])


def test_rccl_all_gather_of_device_draws_world_size_one():
    """rh_comm_*: the engine's own RCCL communicator (dlopen librccl), one rank: unique id -> communicator -> all-gather of
    the device-resident draws -> host, and the max-reduction used for the timing."""
    spec = models.linreg(n=70000, k=3)
    m = R.Model(spec, device=0)
    cfg = R.make_config(5, 5, R.HMCSampler(3), R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    s = R.Sampler(m, cfg, D.shard_seeds(1000, 4, 0))
    s.warmup(); s.run(5)
    comm = D.Comm(D.Comm.unique_id(), 1, 0, 0)
    out = comm.allgather_draws(s)
    assert out.shape == (4, 5, spec.n_params) and np.array_equal(out, s.draws())
    assert comm.allgather_draws(s, to_host=False) != 0
    assert comm.allreduce_max(3.25) == 3.25
    D.device_synchronize(0)
    comm.close()


def test_bench_under_torch_distributed_run_one_rank():
    """bench.py exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), with one
    rank: gloo bootstrap + the engine's RCCL all-gather inside the timed region; ONE JSON line on stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
           "--chains-per-gpu", "64", "--rows", "100000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["config"]["chains"] == 64
    # roofline.traffic is measured in the run (two rocprofv3 --pmc child runs); if the profiler is unusable on this box the line says so
    r = d["roofline"]
    assert (r["traffic"] is not None and r["traffic"] > 0 and r["traffic_source"].startswith("measured in this run")) or "live_traffic_failed" in r


@pytest.mark.parametrize("workload,extra", [
    ("cfg1", ["--steps", "5", "--warmup", "5", "--chains-per-gpu", "64"]),
    ("cfg3", ["--steps", "5", "--warmup", "20", "--chains-per-gpu", "64", "--sampler", "nuts"]),
    ("cfg4", ["--steps", "2", "--warmup", "6", "--chains-per-gpu", "32", "--rows", "100000"]),
    ("cfg5", ["--steps", "2", "--warmup", "4", "--chains-per-gpu", "16", "--rows", "20000"])])
def test_side_workloads_under_torch_distributed_run_one_rank(workload, extra):
    """the other BASELINE configurations through the same launch contract (`--workload cfgN --gpus N`), one rank: sharded seeds,
    the engine's RCCL all-gather of the draws, max-over-ranks timing, ONE JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--workload", workload] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["workload"].startswith(workload) and d["scaling"] == "weak"


def test_a_missing_rank_is_a_clean_error_not_a_hang():
    """rh_comm_create for a world of two with only one rank present: ncclCommInitRank would block for ever; the engine gives up after
    RH_COMM_TIMEOUT_S and says which rank waited (run in a child process: the helper thread stays inside RCCL)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from rainier_amd import _capi, distributed as D\n"
            "_capi.lib(); uid = D.Comm.unique_id(); t = time.time()\n"
            "try:\n    D.Comm(uid, 2, 0, 0); print('NO ERROR')\n"
            "except _capi.RainierHipError as e:\n    print('ERR', e.code, round(time.time() - t), str(e))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RH_COMM_TIMEOUT_S="4"))
    assert "ERR %d" % _capi.RH_E_DEVICE in out.stdout and "never reached rh_comm_create" in out.stdout, out.stdout + out.stderr[-2000:]
