"""Multi-GPU sharding of chains: one process per GPU, no data-path collective, ONE all-gather of the draws.

Chains never interact during warm-up or sampling (everything in Driver.sample is per chain,
rainier-sampler/.../sampler/Driver.scala:13-17), so GPU g simply owns the global chains
[g*C, (g+1)*C) with seeds indexed by GLOBAL chain id -- results do not depend on the number of GPUs.
Observation columns and the compiled kernel are replicated per GPU.  The only exchange is the final
all-gather of the device-resident draws: RCCL over xGMI, issued by the engine itself behind the C ABI
(rh_comm_*, csrc/comm.cpp).  The host program only bootstraps it: rank 0's 128-byte RCCL unique id is
handed to the other ranks through whatever channel it has -- here a torch.distributed *gloo* (CPU)
broadcast, so that torch never initialises its own bundled HIP runtime next to the engine's.

(The single-process form -- one host thread per device inside one rh_sample_multi call -- is in sampler.sample_multi.)
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _capi


def shard_seeds(base_seed: int, chains_per_rank: int, rank: int) -> List[int]:
    """Seeds of the chains owned by `rank`: base + global chain id."""
    return [base_seed + rank * chains_per_rank + c for c in range(chains_per_rank)]


def global_chain_ids(chains_per_rank: int, rank: int) -> range:
    return range(rank * chains_per_rank, (rank + 1) * chains_per_rank)


def exchange_unique_id(dist, make_id, rank: int) -> bytes:
    """rank 0 creates the communicator id (make_id() -> 128 bytes), every rank receives it: a broadcast over the host
    program's own process group (gloo in bench.py and in the CPU test)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    assert isinstance(box[0], (bytes, bytearray)) and len(box[0]) == 128
    return bytes(box[0])


class Comm:
    """rh_comm: the RCCL communicator of one rank (include/rainier_hip.h, multi-process multi-GPU)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int):
        self.world, self.rank, self.device = world, rank, device
        self._h = C.c_void_p()
        _capi.check(_capi.lib().rh_comm_create(unique_id, world, rank, device, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _capi.check(_capi.lib().rh_comm_unique_id(buf))
        return buf.raw

    def allgather_draws(self, sampler, to_host: bool = True):
        """every rank's draws [chains][iterations][nvars] -> [world*chains][iterations][nvars] (global chain id order);
        to_host = False leaves them on the device and returns the device pointer."""
        shape = (self.world * sampler.chains, sampler.iterations, sampler.model.nVars)
        if to_host:
            out = np.zeros(shape)
            _capi.check(_capi.lib().rh_comm_allgather_draws(self._h, sampler._h, _capi.dptr(out), None))
            return out
        p = C.c_void_p()
        _capi.check(_capi.lib().rh_comm_allgather_draws(self._h, sampler._h, None, C.byref(p)))
        return p.value

    def allreduce_max(self, v: float) -> float:
        a = np.array([float(v)])
        _capi.check(_capi.lib().rh_comm_allreduce_max(self._h, _capi.dptr(a)))
        return float(a[0])

    def close(self):
        if self._h:
            _capi.lib().rh_comm_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def device_synchronize(device: int) -> None:
    _capi.check(_capi.lib().rh_device_synchronize(int(device)))
