"""`_native.OutputPool` - the host logic that lets env.step() re-use its pinned output buffers without ever overwriting an array
the caller still holds (the reference returns fresh arrays, TE:101, TE:110).  Runs without a device: the pinned allocator is
replaced by ordinary memory, the reference counting is the real thing."""
import ctypes as C
import gc

import numpy as np
import pytest

from mbt_gym_amd import _native


class _FakePinned:
    """mbt_host_alloc / mbt_host_free on ordinary memory, counting live blocks."""

    live = 0

    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self._memory = C.create_string_buffer(max(int(np.prod(self.shape)) * self.dtype.itemsize, 1))
        self.ptr = C.addressof(self._memory)
        type(self).live += 1

    @property
    def __array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3, "strides": None}

    def array(self):
        return np.asarray(self)

    def __del__(self):
        type(self).live -= 1


@pytest.fixture()
def pool_factory(monkeypatch):
    monkeypatch.setattr(_native, "PinnedBuffer", _FakePinned)
    monkeypatch.delenv("MBT_FRESH_OUTPUTS", raising=False)
    _FakePinned.live = 0
    return _native.OutputPool


def test_the_usual_loop_alternates_between_two_buffers(pool_factory):
    pool = pool_factory((100, 4))
    seen = set()
    obs = None
    for k in range(20):
        obs, pointer = pool.acquire_with_pointer()  # the previous array dies when the name is rebound - AFTER the call, like `obs, ... = env.step()`
        assert pointer == obs.ctypes.data and obs.shape == (100, 4) and obs.dtype == np.float32
        obs[:] = k
        seen.add(pointer)
    assert len(seen) == 2 and len(pool.buffers) == 2 and _FakePinned.live == 2 and pool.pinned_bytes == 2 * 1600


def test_an_array_a_view_or_a_view_of_a_view_keeps_its_buffer_out_of_circulation(pool_factory):
    pool = pool_factory((50, 4))
    first, _ = pool.acquire_with_pointer()
    first[:] = 7.0
    column = first[:, 1]       # a view of what was handed out
    corner = column[3:9]       # a view of a view
    del first, column
    for k in range(6):
        other, _ = pool.acquire_with_pointer()
        other[:] = -1.0
        del other
    np.testing.assert_array_equal(corner, 7.0)  # never written over
    assert len(pool.buffers) == 2               # the held one + ONE that the loop kept re-using
    address = corner.ctypes.data
    del corner
    again = [pool.acquire_with_pointer() for _ in range(2)]
    assert any(p <= address < p + 800 for _, p in again)  # back in circulation once the last view is gone


def test_a_caller_that_holds_many_outputs_grows_the_pool_up_to_its_cap_then_gets_ordinary_arrays(pool_factory):
    pool = pool_factory((10,), max_bytes=4 * 40)
    held = [pool.acquire_with_pointer() for _ in range(6)]
    assert [p is not None for _, p in held] == [True] * 4 + [False] * 2 and _FakePinned.live == 4
    for k, (array, _) in enumerate(held):
        array[:] = k
    for k, (array, _) in enumerate(held):
        np.testing.assert_array_equal(array, k)


def test_idle_buffers_beyond_the_minimum_are_freed_and_release_returns_everything(pool_factory):
    pool = pool_factory((10,), idle_seconds=0.0)
    held = [pool.acquire_with_pointer()[0] for _ in range(5)]
    assert _FakePinned.live == 5
    del held
    pool.acquire_with_pointer()  # trims what nobody references and nobody used for `idle_seconds`
    gc.collect()
    assert len(pool.buffers) == 2 and _FakePinned.live == 2
    kept = pool.acquire_with_pointer()[0]
    kept[:] = 3.0
    pool.release()
    gc.collect()
    assert pool.pinned_bytes == 0 and _FakePinned.live == 1  # the block under `kept` goes with `kept`
    np.testing.assert_array_equal(kept, 3.0)
    del kept
    gc.collect()
    assert _FakePinned.live == 0


def test_fresh_outputs_mode_hands_out_a_new_array_every_time(pool_factory, monkeypatch):
    monkeypatch.setenv("MBT_FRESH_OUTPUTS", "1")
    pool = pool_factory((10,))
    a, pa = pool.acquire_with_pointer()
    b, pb = pool.acquire_with_pointer()
    assert pa is None and pb is None and a is not b and _FakePinned.live == 0


def test_without_pinned_memory_the_pool_falls_back_to_ordinary_arrays(monkeypatch):
    class NoPinnedMemory:
        def __init__(self, *args, **kwargs):
            raise _native.NativeError(-3, "no device")

    monkeypatch.setattr(_native, "PinnedBuffer", NoPinnedMemory)
    pool = _native.OutputPool((10,))
    array, pointer = pool.acquire_with_pointer()
    assert pointer is None and array.shape == (10,) and pool.acquire()[1] is False and pool.pinned_bytes == 0
