"""Throw-away stand-in for the `gym` package (absent from this image, no network).

Used ONLY by tools/refgen/*.py, in the build container, so that the Python reference at
/root/reference can be imported to emit golden vectors.  It carries no numerics: it provides the
few container types the reference's constructors touch (Env, Wrapper, spaces.Box, MultiBinary).
Nothing under mbt_gym_amd/, oracle/, tests/ or bench.py imports it.
"""
from . import spaces  # noqa: F401


class Env:
    metadata = {}

    def __init__(self, *args, **kwargs):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)


class ObservationWrapper(Wrapper):
    pass
