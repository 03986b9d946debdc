"""Observation/action space containers.

`gym` (and `gymnasium`) are used when importable so that Stable-Baselines3 sees the types it expects; this image
has neither, so a minimal Box with the same attributes (`low`, `high`, `shape`, `dtype`, `sample`, `seed`)
stands in.  Bounds are float32, like the reference's (gym/TradingEnvironment.py:241)."""
import numpy as np

try:  # pragma: no cover - not installed in the build image
    from gym.spaces import Box, MultiBinary, Space  # type: ignore
    HAVE_GYM = True
except Exception:  # noqa: BLE001
    HAVE_GYM = False

    class Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._np_random = np.random.default_rng()

        def seed(self, seed=None):
            self._np_random = np.random.default_rng(seed)
            return [seed]

    class Box(Space):
        """Closed box in R^n with float32 bounds."""

        def __init__(self, low, high, shape=None, dtype=np.float32):
            low, high = np.asarray(low), np.asarray(high)
            if shape is None:
                shape = np.broadcast(low, high).shape
            super().__init__(shape, dtype)
            self.low = np.array(np.broadcast_to(low, self.shape), dtype=self.dtype)
            self.high = np.array(np.broadcast_to(high, self.shape), dtype=self.dtype)

        def sample(self):
            return self._np_random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class MultiBinary(Space):
        """{0, 1}^n (the action space of at-the-touch dynamics, ModelDynamics.py:165-167)."""

        def __init__(self, n):
            super().__init__((n,), np.int8)
            self.n = n

        def sample(self):
            return self._np_random.integers(0, 2, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))
