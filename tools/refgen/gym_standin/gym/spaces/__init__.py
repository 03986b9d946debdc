import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low)
        high = np.asarray(high)
        if shape is None:
            shape = np.broadcast(low, high).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(low, self.shape).astype(self.dtype)
        self.high = np.broadcast_to(high, self.shape).astype(self.dtype)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class MultiBinary(Space):
    def __init__(self, n):
        super().__init__((n,), np.int8)
        self.n = n

    def sample(self):
        return self._rng.integers(0, 2, size=self.shape).astype(self.dtype)


from . import box  # noqa: E402,F401
