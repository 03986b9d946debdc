"""Stable-Baselines3 VecEnv face of a TradingEnvironment: num_envs = num_trajectories
(reference: mbt_gym/gym/StableBaselinesTradingEnvironment.py:11-66).

`step_wait` steps every lane with one kernel launch; when the (lane-invariant) episode ends it stores each lane's
last observation under infos[i]["terminal_observation"] and returns the observation of the automatic reset, with
the terminal step's rewards and dones (SBE:28-37).  stable_baselines3 is optional: without it the class derives
from a minimal stand-in with the same attributes, so the adapter's semantics can be tested anywhere.
"""
from typing import Any, List, Optional, Sequence

import numpy as np

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase

    HAVE_SB3 = True
except Exception:  # noqa: BLE001
    HAVE_SB3 = False

    class _VecEnvBase:
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space = observation_space
            self.action_space = action_space

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()


class StableBaselinesTradingEnvironment(_VecEnvBase):
    def __init__(self, trading_env, store_terminal_observation_info: bool = True):
        self.env = trading_env
        self.store_terminal_observation_info = store_terminal_observation_info
        self.actions: np.ndarray = self.env.action_space.sample()
        super().__init__(self.env.num_trajectories, self.env.observation_space, self.env.action_space)

    def reset(self):
        return self.env.reset()

    def step_async(self, actions: np.ndarray) -> None:
        self.actions = actions

    def step_wait(self):
        obs, rewards, dones, infos = self.env.step(self.actions)
        if dones.min():
            if self.store_terminal_observation_info:
                infos = infos.copy() if isinstance(infos, list) else [infos]
                for lane, info in enumerate(infos):
                    info["terminal_observation"] = obs[lane, :]
            obs = self.env.reset()
        return obs, rewards, dones, infos

    def close(self) -> None:
        self.env.close()

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        return [getattr(self.env, attr_name)] * self.env.num_trajectories

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        pass

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        pass

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False for _ in range(self.env.num_trajectories)]

    def seed(self, seed: Optional[int] = None):
        return self.env.seed(seed)

    def get_images(self) -> Sequence[np.ndarray]:
        pass

    @property
    def num_trajectories(self):
        return self.env.num_trajectories

    @property
    def n_steps(self):
        return self.env.n_steps
