"""N > 1 path on CPU: world_size-2 gloo run of the sharding + final all-gather used by bench.py."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from rainier_amd import distributed as D
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cpr, iters, n = 3, 4, 5
    seeds = D.shard_seeds(1000, cpr, rank)
    assert seeds == [1000 + g for g in D.global_chain_ids(cpr, rank)]
    # stand-in for the device draws: a pure function of the chain's seed, as the engine guarantees
    local = torch.tensor([[[s * 100.0 + i * 10 + v for v in range(n)] for i in range(iters)] for s in seeds], dtype=torch.float64)
    dist.barrier()
    out = D.gather_draws(local, world)
    want = torch.tensor([[[(1000 + g) * 100.0 + i * 10 + v for v in range(n)] for i in range(iters)] for g in range(world * cpr)], dtype=torch.float64)
    assert out.shape == (world * cpr, iters, n) and torch.equal(out, want)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the max-over-ranks timing reduction of bench.py
    assert t.item() == world
    dist.destroy_process_group()
    print("rank%%d_ok" %% rank, flush=True)
""") % ROOT


def test_two_rank_gloo_sharding_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0_ok" in out.stdout and "rank1_ok" in out.stdout
