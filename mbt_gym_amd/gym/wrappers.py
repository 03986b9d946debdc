"""Observation / reward wrappers around a TradingEnvironment (reference: mbt_gym/gym/wrappers.py).

The wrappers are host-side views of what the HIP step produced: nothing here computes on the path, so the three
classes only re-express the reference's slicing / scaling on the arrays `env.step()` returns.  Reference behaviour
that is reproduced on purpose:

* `ReduceStateSizeWrapper` (wrappers.py:10-39): column subset of the observation, Box bounds subset as float64.
* `NormaliseASObservation` (wrappers.py:46-76): `reset()` returns `(obs - offset) * factor` but `step()` returns
  `obs / factor` (wrappers.py:68 vs :76) - the asymmetry is the reference's and is kept.
* `RemoveTerminalRewards` (wrappers.py:79-105): on the terminal step the reward is scaled by
  `per_step_inventory_aversion / terminal_inventory_aversion`; infos become `{}`.  The reference tests `if done:` on
  the dones array, which only works for one trajectory; dones are identical across trajectories (TE:218-220), so
  the first entry decides here.
"""
import numpy as np

from mbt_gym_amd.gym.index_names import INVENTORY_INDEX, TIME_INDEX
from mbt_gym_amd.spaces import Box

try:  # pragma: no cover - gym is not installed in the build image
    from gym import Wrapper  # type: ignore
except Exception:  # noqa: BLE001

    class Wrapper:
        """The part of gym.Wrapper the three wrappers rely on: attribute forwarding to the wrapped environment."""

        def __init__(self, env):
            self.env = env
            self.observation_space = env.observation_space
            self.action_space = env.action_space

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return getattr(self.env, "unwrapped", self.env)

        @property
        def metadata(self):
            return getattr(self.env, "metadata", {})

        def seed(self, seed=None):
            return self.env.seed(seed)

        def reset(self):
            return self.env.reset()

        def step(self, action):
            return self.env.step(action)

        def close(self):
            close = getattr(self.env, "close", None)
            return close() if close is not None else None


class _ObservationView(Wrapper):
    """Shared plumbing of the observation wrappers: `_on_reset` / `_on_step` map the observation the environment hands
    back; rewards, dones and infos pass through untouched."""

    def _on_reset(self, obs):
        return obs

    def _on_step(self, obs):
        return obs

    def reset(self):
        return self._on_reset(self.env.reset())

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self._on_step(obs), reward, done, info

    @property
    def spec(self):
        return getattr(self.env, "spec", None)


class ReduceStateSizeWrapper(_ObservationView):
    """Keeps the listed observation columns (default: inventory and time), wrappers.py:15-39."""

    def __init__(self, env, list_of_state_indices: list = [INVENTORY_INDEX, TIME_INDEX]):
        super().__init__(env)
        assert isinstance(env.observation_space, Box)
        self.list_of_state_indices = list_of_state_indices
        box = env.observation_space
        self.observation_space = Box(low=box.low[list_of_state_indices], high=box.high[list_of_state_indices], dtype=np.float64)

    def _on_reset(self, obs):
        return obs[:, self.list_of_state_indices]

    _on_step = _on_reset


class NormaliseASObservation(_ObservationView):
    """Affine map of the observation box onto [-1, 1]^D (wrappers.py:51-76), including the reference's step/reset asymmetry."""

    def __init__(self, env):
        super().__init__(env)
        assert isinstance(env.observation_space, Box)
        box = env.observation_space
        self.normalisation_factor = 2 / (box.high - box.low)
        self.normalisation_offset = (box.high + box.low) / 2
        self.observation_space = Box(low=-np.ones(box.shape), high=np.ones(box.shape), dtype=np.float64)

    def _on_reset(self, obs):
        return (obs - self.normalisation_offset) * self.normalisation_factor  # wrappers.py:68

    def _on_step(self, obs):
        return obs / self.normalisation_factor  # wrappers.py:76 (sic)


class RemoveTerminalRewards(Wrapper):
    """Rescales the terminal reward by per-step / terminal inventory aversion (wrappers.py:84-105)."""

    def __init__(self, env, num_final_steps: int = 5):
        super().__init__(env)

    def reset(self):
        return self.env.reset()

    def step(self, action):
        state, reward, done, _ = self.env.step(action)
        if np.asarray(done).reshape(-1)[0]:
            criterion = self.env.reward_function
            reward = reward * (criterion.per_step_inventory_aversion / criterion.terminal_inventory_aversion)
        return state, reward, done, {}
