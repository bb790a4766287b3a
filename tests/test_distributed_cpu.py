"""N > 1 path on CPU: world_size-2 and world_size-8 gloo runs of the HOST side of bench.py's multi-GPU protocol -- chain sharding by global
id, the broadcast that hands rank 0's RCCL unique id to the other ranks, the barrier and the host reductions.  The device
side (rh_comm_*: RCCL all-gather of real engine draws) is covered on the GPU by tests/test_gpu_multi.py."""
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
    cpr = 3
    seeds = D.shard_seeds(1000, cpr, rank)
    assert seeds == [1000 + g for g in D.global_chain_ids(cpr, rank)]
    # every rank ends up with rank 0's 128-byte id (on a GPU box make_id is rh_comm_unique_id)
    uid = D.exchange_unique_id(dist, lambda: bytes([7 + rank]) * 128, rank)
    assert uid == bytes([7]) * 128
    dist.barrier()
    # the host reductions of bench.py (HostGroup.sum; the max-over-ranks timing goes through RCCL on the GPU)
    t = torch.tensor([float(rank + 1), 10.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    assert t.tolist() == [sum(range(1, world + 1)), 10.0 * sum(range(1, world + 1))]
    # union of the shards = all global chains, in rank order
    ids = [None] * world
    dist.all_gather_object(ids, list(D.global_chain_ids(cpr, rank)))
    assert [g for part in ids for g in part] == list(range(world * cpr))
    dist.destroy_process_group()
    print("rank%%d_ok" %% rank, flush=True)
""") % ROOT


import pytest


@pytest.mark.parametrize("world", [2, 8])     # 8: the node shape the driver's scaling runs use (one rank per MI355X)
def test_gloo_sharding_and_bootstrap(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert all("rank%d_ok" % r in out.stdout for r in range(world))


def test_comm_entry_points_fail_cleanly_without_a_device():
    """no GPU here: the RCCL entry points return a status (never crash), and bad arguments are rejected first"""
    import ctypes as C
    from rainier_amd import _capi
    L = _capi.lib()
    if L.rh_device_count() > 0:
        return
    buf = C.create_string_buffer(128)
    # (RH_OK when a torch import earlier in this process already brought an RCCL that hands out ids without a device)
    assert L.rh_comm_unique_id(buf) in (_capi.RH_OK, _capi.RH_E_DEVICE, _capi.RH_E_UNSUPPORTED)
    h = C.c_void_p()
    assert L.rh_comm_create(buf, 0, 0, 0, C.byref(h)) == _capi.RH_E_INVALID
    assert L.rh_comm_create(buf, 2, 2, 0, C.byref(h)) == _capi.RH_E_INVALID
    assert L.rh_comm_create(buf, 1, 0, 0, C.byref(h)) in (_capi.RH_E_DEVICE, _capi.RH_E_UNSUPPORTED) and not h.value
    assert L.rh_comm_allgather_draws(None, None, None, None) == _capi.RH_E_INVALID
