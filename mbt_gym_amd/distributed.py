"""Sharding of the trajectory axis over the GPUs of one node, and the only collective on the path.

Trajectories are independent (no operation of the step mixes lanes; the reference's only cross-lane reads are the
lane-invariant time and done flag, TradingEnvironment.py:218-220), so each rank owns a contiguous range of GLOBAL
lane ids and steps it with no communication at all.  Noise is a function of the global lane id, so results do
not depend on the number of ranks.  What a run reports - the mean (and spread) of the episode return - needs three
doubles per rank, [sum R, sum R^2, count]: ONE all-reduce over RCCL/xGMI per episode, 24 bytes, latency bound
(never inside the step loop).  Two transports: the C ABI's own RCCL binding (`RcclCommunicator` +
`TradingEnvironment.set_communicator` / `allreduce_return_sums`: the reduction is enqueued on the environment's stream
and never touches Python) and `torch.distributed` (`allreduce_return_sums` / `PendingReturnSums` below; backend "nccl" =
RCCL on ROCm, "gloo" on CPU - the route the CPU tests and single-device multi-rank tests take).
"""
from typing import Tuple

import numpy as np


SHARD_ALIGN = 1024  # noise is drawn per tile of 512 lanes (order book, csrc/philox.hpp) or 1024 lanes (speed dynamics,
# csrc/speed_kernel.hpp): shards start on tile boundaries


def shard_bounds(total_lanes: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous lane range; offsets are multiples of 1024."""
    assert 0 <= rank < world_size
    per = -(-total_lanes // world_size)
    per = -(-per // SHARD_ALIGN) * SHARD_ALIGN
    offset = rank * per  # an empty shard (count 0) still sits on a tile boundary
    return offset, max(0, min(per, total_lanes - offset))


class RcclCommunicator:
    """An RCCL communicator created through the C ABI (mbt_comm_*), one rank per GPU.  The 128-byte unique id is made on
    rank 0 and handed to the others by `exchange` - any callable bytes -> bytes that broadcasts rank 0's value (the
    default uses the torch.distributed process group the launcher already set up; the collective itself never goes
    through torch).  `handle` is the ncclComm_t for mbt_env_allreduce_returns / mbt_env_set_communicator."""

    def __init__(self, rank: int, world_size: int, device: int, exchange=None):
        import ctypes as C

        from mbt_gym_amd import _native

        _native.preload_torch_rccl()
        lib = _native.load_library()
        ident = C.create_string_buffer(_native.COMM_ID_BYTES)
        if rank == 0:
            _native.check(lib.mbt_comm_unique_id(ident))
        raw = ident.raw if world_size == 1 else (exchange or _broadcast_bytes)(ident.raw)  # (a world of one needs no launcher, and no torch)
        assert len(raw) == _native.COMM_ID_BYTES
        handle = C.c_void_p()
        _native.check(lib.mbt_comm_init_rank(int(device), int(world_size), C.create_string_buffer(raw, _native.COMM_ID_BYTES), int(rank), C.byref(handle)))
        self.handle, self.rank, self.world_size, self._lib = handle, rank, world_size, lib

    def count(self) -> int:
        """The number of ranks RCCL itself says the communicator spans (ncclCommCount)."""
        import ctypes as C

        from mbt_gym_amd import _native

        out = C.c_int(0)
        _native.check(self._lib.mbt_comm_count(self.handle, C.byref(out)))
        return int(out.value)

    def close(self):
        if self.handle is not None:
            self._lib.mbt_comm_destroy(self.handle)
            self.handle = None


def _broadcast_bytes(payload: bytes) -> bytes:
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return payload
    box = [payload]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def allreduce_return_sums(sums, device=None) -> np.ndarray:
    """Sum the per-rank [sum R, sum R^2, count] over the default process group (no-op without one)."""
    import torch
    import torch.distributed as dist

    local = np.asarray(sums, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    t = torch.tensor(np.nan_to_num(local, nan=0.0), dtype=torch.float64, device=device or "cpu")
    flag = torch.tensor([float(np.isnan(local[1]))], dtype=torch.float64, device=t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    out = t.cpu().numpy()
    if flag.item() > 0:  # some rank did not track per-lane returns: the second moment is unknown
        out[1] = np.nan
    return out


class PendingReturnSums:
    """An all-reduce of [sum R, sum R^2, count] that has been started (`async_op=True`) but not waited for: the 24-byte
    collective of one episode overlaps the stepping of the next.  `result()` waits and returns the global sums."""

    def __init__(self, sums, device=None):
        import torch
        import torch.distributed as dist

        self._local = np.asarray(sums, dtype=np.float64)
        self._work = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            nan = float(np.isnan(self._local[1]))  # a rank that does not track per-lane returns makes the second moment unknown
            self._t = torch.tensor(list(np.nan_to_num(self._local, nan=0.0)) + [nan], dtype=torch.float64, device=device or "cpu")
            self._work = dist.all_reduce(self._t, op=dist.ReduceOp.SUM, async_op=True)

    def result(self) -> np.ndarray:
        if self._work is None:
            return self._local
        self._work.wait()
        out = self._t.cpu().numpy()
        sums = out[:3].copy()
        if out[3] > 0:
            sums[1] = np.nan
        return sums


def return_statistics(sums) -> Tuple[float, float]:
    """(mean, population std) of the episode return from [sum R, sum R^2, count] (plotting.py:104-105)."""
    total, total_sq, count = (float(x) for x in sums)
    mean = total / count
    var = total_sq / count - mean * mean
    return mean, float(np.sqrt(max(var, 0.0))) if not np.isnan(var) else float("nan")
