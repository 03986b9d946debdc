"""The device-resident interface: torch wraps the library's buffers without a copy (__cuda_array_interface__),
an on-GPU policy writes actions in place, the environment runs on torch's stream."""
import numpy as np
import pytest

from tests.env_factory import make_env
from tests.golden_io import load_case

pytestmark = pytest.mark.gpu


def test_torch_wraps_device_buffers_without_copies():
    torch = pytest.importorskip("torch")
    cfg, _ = load_case("as_limit_pnl")
    cfg.num_trajectories, cfg.seed = 2048, 77
    env = make_env(cfg)
    env.set_stream(torch.cuda.current_stream().cuda_stream)
    host_obs = env.reset()
    obs = torch.as_tensor(env.obs_device, device="cuda")
    assert obs.data_ptr() == env.obs_device.ptr and obs.shape == (2048, 4) and obs.dtype == torch.float32
    np.testing.assert_array_equal(obs.cpu().numpy(), host_obs)
    act = torch.as_tensor(env.action_device, device="cuda")
    act[:, 0] = 0.5
    act[:, 1] = 0.9  # an "on-GPU policy" writing its action in place, ordered before the step on the same stream
    env.step_device()
    rew = torch.as_tensor(env.reward_device, device="cuda")
    obs1 = torch.as_tensor(env.obs_device, device="cuda")
    torch.cuda.synchronize()
    # the same step through the host path on a twin environment
    twin = make_env(cfg)
    twin.reset()
    o, r, _, _ = twin.step(np.tile(np.array([[0.5, 0.9]], np.float32), (2048, 1)))
    np.testing.assert_array_equal(obs1.cpu().numpy(), o)
    np.testing.assert_array_equal(rew.cpu().numpy(), r)
    # the regime is a stated property of the handle (mbt_env_state_in_place), not something a consumer has to probe: at this size two
    # buffers alternate - the previous rows outlive the step
    assert env.obs_device_aliases_next is False and obs1.data_ptr() != obs.data_ptr()
    np.testing.assert_array_equal(obs.cpu().numpy(), host_obs)
    env.close()
    twin.close()


@pytest.mark.parametrize("ping_pong", ["0", "1"])
def test_each_regime_of_the_state_buffers_is_what_the_getter_says(monkeypatch, ping_pong):
    """MBT_PING_PONG_STATE forces a regime: in place, `obs_device` keeps its address and a tensor over it shows the NEXT observation
    after a step (the documented aliasing a replay buffer has to copy around); with two buffers the address alternates and the old
    rows stay as they were until the step after next."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("MBT_PING_PONG_STATE", ping_pong)
    cfg, _ = load_case("as_limit_pnl")
    cfg.num_trajectories, cfg.seed = 2048, 77
    env = make_env(cfg)
    env.set_stream(torch.cuda.current_stream().cuda_stream)
    host_obs = env.reset()
    env.set_action_host(np.tile(np.array([[0.5, 0.9]], np.float32), (2048, 1)))
    before = torch.as_tensor(env.obs_device, device="cuda")
    env.step_device()
    after = torch.as_tensor(env.obs_device, device="cuda")
    torch.cuda.synchronize()
    if ping_pong == "0":
        assert env.obs_device_aliases_next is True and after.data_ptr() == before.data_ptr()
        assert not np.array_equal(before.cpu().numpy(), host_obs)  # the tensor made before the step shows the step's result
    else:
        assert env.obs_device_aliases_next is False and after.data_ptr() != before.data_ptr()
        np.testing.assert_array_equal(before.cpu().numpy(), host_obs)
        env.step_device()
        torch.cuda.synchronize()
        assert env.obs_device.ptr == before.data_ptr()  # ... and is overwritten by the step after next
    env.close()
