"""Stable-Baselines3 VecEnv face of a TradingEnvironment: num_envs = num_trajectories
(reference: mbt_gym/gym/StableBaselinesTradingEnvironment.py:11-66).

`step_wait` steps every lane with one kernel launch; when the (lane-invariant) episode ends it stores each lane's
last observation under infos[i]["terminal_observation"] and returns the observation of the automatic reset, with
the terminal step's rewards and dones (SBE:28-37).  stable_baselines3 is optional: without it the class derives
from a minimal stand-in with the same attributes, so the adapter's semantics can be tested anywhere.
"""
import collections.abc
from typing import Any, List, Optional, Sequence

import numpy as np

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase

    HAVE_SB3 = True
except Exception:  # noqa: BLE001
    HAVE_SB3 = False

    class _VecEnvBase:
        """num_envs / observation_space / action_space and the step = step_async + step_wait split of SB3's VecEnv."""

        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()


LAZY_INFOS_ABOVE = 1 << 16  # lanes: beyond this the terminal infos are a lazy sequence instead of a list of dicts


class TerminalObservationInfos(collections.abc.Sequence):
    """infos of a terminal step for a large batch: behaves like the list of dicts SB3 expects - `len`, iteration, `infos[i]`,
    `infos[i].get("terminal_observation")`, slicing - but builds a lane's dict only when it is asked for.  The reference
    writes one dict entry per lane in a Python loop at every episode end (SBE:31-35): 0.4 s per episode at 2^20 lanes before
    any consumer has looked at them.  (A consumer that touches every lane - SB3's VecMonitor and collect_rollouts do, in
    Python, every step - is O(num_envs) on its own side; at that scale the device-resident interface is the intended one:
    `step_device`, `rollout_device`, `episode_log_pop`.)"""

    def __init__(self, terminal_observation: np.ndarray):
        self._obs = terminal_observation

    def __len__(self):
        return len(self._obs)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [{"terminal_observation": row} for row in self._obs[index]]
        return {"terminal_observation": self._obs[index, :]}


class StableBaselinesTradingEnvironment(_VecEnvBase):
    """All trajectories of one TradingEnvironment as the sub-environments of a VecEnv (they share the episode clock)."""

    def __init__(self, trading_env, store_terminal_observation_info: bool = True):
        self.env = trading_env
        self.store_terminal_observation_info = store_terminal_observation_info
        self.actions: np.ndarray = trading_env.action_space.sample()  # SBE:18: a placeholder until the first step_async
        _VecEnvBase.__init__(self, trading_env.num_trajectories, trading_env.observation_space, trading_env.action_space)

    # ---- the stepping protocol ----------------------------------------------------------------------------------
    def reset(self):
        return self.env.reset()

    def step_async(self, actions: np.ndarray) -> None:
        self.actions = actions  # nothing is launched yet (SBE:25-26)

    def step_wait(self):
        obs, rewards, dones, infos = self.env.step(self.actions)
        if not dones[0]:  # (the clock is shared - TE:218-220 fills `dones` with ONE flag - so lane 0 speaks for `dones.min()`, SBE:30)
            return obs, rewards, dones, infos
        # episode over in every lane (the clock is shared): hand out the terminal observation, then auto-reset (SBE:28-37)
        if self.store_terminal_observation_info:
            if len(obs) > LAZY_INFOS_ABOVE:
                infos = TerminalObservationInfos(obs)
            else:
                infos = list(infos) if isinstance(infos, list) else [infos]
                for info, row in zip(infos, obs):
                    info["terminal_observation"] = row
        return self.env.reset(), rewards, dones, infos

    def seed(self, seed: Optional[int] = None):
        return self.env.seed(seed)

    def close(self) -> None:
        self.env.close()

    # ---- the attribute / method plumbing SB3 expects of a VecEnv (SBE:39-58) ------------------------------------
    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        value = getattr(self.env, attr_name)
        return [value for _ in range(self.num_trajectories)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        return None  # not supported upstream either

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        return None

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False] * self.num_trajectories

    def get_images(self) -> Sequence[np.ndarray]:
        return None

    # ---- conveniences the reference's training scripts read -----------------------------------------------------
    @property
    def num_trajectories(self) -> int:
        return self.env.num_trajectories

    @property
    def n_steps(self) -> int:
        return self.env.n_steps
