import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture()
def no_device(monkeypatch):
    """Host-logic tests: TradingEnvironment without a device handle (no numerics are exercised)."""
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment

    resets = []
    monkeypatch.setattr(TradingEnvironment, "_create_handle", lambda self, n, scale, offset=None: None)

    def fake_reset(self, obs_out=None):
        resets.append((self._get_start_time(), self._get_initial_inventories()))

    monkeypatch.setattr(TradingEnvironment, "_reset_device", fake_reset)
    monkeypatch.setattr(TradingEnvironment, "close", lambda self: None)
    return resets
