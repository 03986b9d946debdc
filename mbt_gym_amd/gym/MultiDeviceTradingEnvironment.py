"""ONE logical TradingEnvironment whose trajectory axis is sharded over several GPUs of a node, driven from one host process.

The reference scales out with `MultiprocessTradingEnv` (gym/MultiprocessTradingEnv.py:72-116: M worker processes, each a
vectorised environment, observations concatenated for one Stable-Baselines3 learner).  The device counterpart of that
consumer - a single learner process that wants every GPU of the node behind one `VecEnv` - is this class: one shard
(`TradingEnvironment(device=g, trajectory_offset=...)`) per device, one host thread per shard (the C ABI's threading
contract; ctypes releases the GIL for the duration of each call, so the devices work concurrently), rows concatenated in
global lane order.  Noise is a function of the GLOBAL lane id and the host-side draws (initial inventories, start times)
are made once for all lanes, so the result does not depend on how many devices there are: it is what one environment with
all the lanes returns (tests/test_gpu_multi_device.py shards one GPU three ways and compares bit for bit).

For throughput-bound work prefer one PROCESS per GPU (bench.py, mbt_gym_amd/distributed.py): nothing crosses the host there.
This class is for consumers that are a single process by construction.
"""
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.distributed import shard_bounds
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment


class MultiDeviceTradingEnvironment:
    """`make_shard(num_trajectories, device, trajectory_offset) -> TradingEnvironment` builds one shard (the processes and
    the dynamics of the reference's API carry their own `num_trajectories`, so the caller's construction code is called
    once per device with the shard's size).  `devices` defaults to every visible GPU; listing a device twice puts two
    shards on it (testing)."""

    def __init__(self, make_shard: Callable[[int, int, int], TradingEnvironment], num_trajectories: int,
                 devices: Optional[Sequence[int]] = None, seed: Optional[int] = None):
        devices = list(range(_native.device_count())) if devices is None else list(devices)
        assert devices, "no device"
        self._num_trajectories = int(num_trajectories)
        self.shards: List[TradingEnvironment] = []
        self.bounds = []
        for rank, device in enumerate(devices):
            offset, count = shard_bounds(self._num_trajectories, rank, len(devices))
            if count == 0:
                continue
            shard = make_shard(count, device, offset)
            assert shard.num_trajectories == count and shard.trajectory_offset == offset and shard.device == device, \
                "make_shard must pass num_trajectories, device and trajectory_offset on to TradingEnvironment"
            self.shards.append(shard)
            self.bounds.append((offset, count))
        first = self.shards[0]
        # host-side decisions are made ONCE for all lanes, with the single environment's own protocol (TE:72, TE:257-281)
        self.initial_inventory, self.start_time = first.initial_inventory, first.start_time
        self.seed_ = seed if seed is not None else first.seed_
        self.rng = np.random.default_rng(self.seed_)
        self._draw_initial_inventories()  # the reference materialises the initial state in its constructor (TE:74): one draw
        self._pool = ThreadPoolExecutor(max_workers=len(self.shards), thread_name_prefix="mbt-shard")
        self._empty_infos = None
        for name in ("terminal_time", "n_steps", "observation_space", "action_space", "original_observation_space", "original_action_space",
                     "max_inventory", "max_cash", "initial_cash", "observation_dim", "action_dim", "reward_function", "normalise_action_space_",
                     "normalise_observation_space_", "reward_scaling"):
            setattr(self, name, getattr(first, name))

    # ---- the reference's environment protocol ---------------------------------------------------------------------------
    @property
    def num_trajectories(self) -> int:
        return self._num_trajectories

    @property
    def step_size(self) -> float:
        return self.shards[0].step_size

    @step_size.setter
    def step_size(self, value: float):
        for shard in self.shards:
            shard.step_size = value

    @property
    def model_dynamics(self):
        return self.shards[0].model_dynamics  # the market description (identical in every shard but for the batch size)

    def _draw_initial_inventories(self):
        q0 = self.initial_inventory
        if isinstance(q0, tuple) and len(q0) == 2:
            return self.rng.integers(*q0, size=self._num_trajectories).astype(np.float32)
        if callable(q0):  # evaluated ONCE for every lane, like the reference (TE:275-279) - not once per shard
            value = q0()
            if self.shards[0].model_dynamics.round_initial_inventory:
                value = int(np.round(value))
            return np.full((self._num_trajectories,), value, dtype=np.float32)
        return None

    def _map(self, fn, *per_shard):
        return list(self._pool.map(fn, self.shards, *per_shard))

    def _slices(self, array):
        return [array[offset:offset + count] for offset, count in self.bounds]

    def reset(self) -> np.ndarray:
        first = self.shards[0]
        first.start_time = self.start_time
        start = first._get_start_time()  # a callable start time is evaluated once, for every lane (TE:257-268)
        q0 = self._draw_initial_inventories()
        for shard, (offset, count) in zip(self.shards, self.bounds):
            shard.start_time = start
            if q0 is not None:
                shard.initial_inventory = q0[offset:offset + count]
        return np.concatenate(self._map(lambda shard: shard.reset()), axis=0)

    def step(self, action: np.ndarray):
        action = _native.as_f32(action, (self._num_trajectories, self.action_dim))
        results = self._map(lambda shard, a: shard.step(a), self._slices(action))
        obs = np.concatenate([r[0] for r in results], axis=0)
        rewards = np.concatenate([r[1] for r in results], axis=0)
        done = bool(results[0][2][0])  # the clock is shared (TE:218-220)
        return obs, rewards, np.full((self._num_trajectories,), done, dtype=bool), self._infos()

    def _infos(self):
        if self._empty_infos is None:  # TE:320-321
            n = self._num_trajectories
            self._empty_infos = [{} for _ in range(n)] if n > 1 else {}
        return self._empty_infos

    def seed(self, seed: int = None):
        self.seed_ = seed
        self.rng = np.random.default_rng(seed)
        self._map(lambda shard: shard.seed(seed))

    def close(self):
        for shard in self.shards:
            shard.close()
        self._pool.shutdown(wait=True)

    # ---- beyond the step loop ---------------------------------------------------------------------------------------------
    def rollout(self, policy, max_steps: int = None, record: bool = True):
        """Every shard's fused rollout at once; recordings concatenated along the lane axis."""
        results = self._map(lambda shard: shard.rollout(policy, max_steps, record))
        steps, done = results[0][3], results[0][4]
        parts = [None if results[0][k] is None else np.concatenate([r[k] for r in results], axis=1) for k in range(3)]
        return parts[0], parts[1], parts[2], steps, done

    def episode_return_sums(self) -> np.ndarray:
        """[sum R, sum R^2, lanes] over every shard (the multi-process path all-reduces the same three numbers)."""
        sums = np.array(self._map(lambda shard: shard.episode_return_sums()), dtype=np.float64)
        return sums.sum(axis=0)

    def track_lane_returns(self, on: bool = True):
        self._map(lambda shard: shard.track_lane_returns(on))

    @property
    def state(self) -> np.ndarray:
        return np.concatenate([shard.state for shard in self.shards], axis=0)
