"""Multi-GPU sharding of chains: one process per GPU, no data-path collective, ONE all-gather of the draws.

Chains never interact during warm-up or sampling (everything in Driver.sample is per chain,
rainier-sampler/.../sampler/Driver.scala:13-17), so GPU g simply owns the global chains
[g*C, (g+1)*C) with seeds indexed by GLOBAL chain id -- results do not depend on the number of GPUs.
Observation columns and the compiled kernel are replicated per GPU.  The only exchange is the final
all-gather of the draws (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List


def shard_seeds(base_seed: int, chains_per_rank: int, rank: int) -> List[int]:
    """Seeds of the chains owned by `rank`: base + global chain id."""
    return [base_seed + rank * chains_per_rank + c for c in range(chains_per_rank)]


def global_chain_ids(chains_per_rank: int, rank: int) -> range:
    return range(rank * chains_per_rank, (rank + 1) * chains_per_rank)


def gather_draws(local, world_size: int):
    """local: torch tensor [chains_per_rank][iterations][nvars] on this rank's device.
    Returns [world*chains_per_rank][iterations][nvars], rank-major == global chain id order."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world_size * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_draws_from_device(dev_ptr: int, shape, world_size: int):
    """The engine's device-resident draws (rh_sampler_draws_device: [chains_per_rank][iterations][nvars] fp64 on this
    rank's GPU) -> all ranks' draws, without touching the host: one device-to-device copy into a torch tensor (the
    engine's buffer belongs to the sampler handle) and ONE all-gather (RCCL over xGMI)."""
    import ctypes as C
    import torch
    n = 1
    for d in shape:
        n *= int(d)
    local = torch.empty(tuple(int(d) for d in shape), dtype=torch.float64, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.restype = C.c_int
    rc = hip.hipMemcpy(C.c_void_p(local.data_ptr()), C.c_void_p(int(dev_ptr)), C.c_size_t(n * 8), C.c_int(3))  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError("hipMemcpy(device draws -> torch tensor) failed: hipError %d" % rc)
    return gather_draws(local, world_size)

