"""The summary table of one batch of episodes (reference: mbt_gym/gym/helpers/plotting.py:96-108,
`generate_results_table_and_hist`): mean quoted spread, mean / std of the total reward, mean / std of the terminal
inventory.  The reference builds it from a host-side trajectory; here the episode is one fused rollout launch and, when
PyTorch sees the GPU, the five statistics are reduced on the device from the recording, so nothing but five numbers (and
the per-trajectory totals the reference also returns) crosses PCIe.  The histogram figure needs matplotlib (helpers/plotting.py), which
is optional: without it the second return value is None."""
import numpy as np

from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory, generate_trajectory_on_device
from mbt_gym_amd.gym.index_names import INVENTORY_INDEX

COLUMNS = ["Mean spread", "Mean PnL", "Std PnL", "Mean terminal inventory", "Std terminal inventory"]


def episode_statistics(vec_env, agent, on_device: bool = None):
    """(dict of the five statistics, total reward per trajectory as a float64 NumPy array)."""
    if on_device is None:
        on_device = hasattr(agent, "device_policy") and _torch_sees_the_gpu()
    if on_device:
        import torch

        obs, act, rew = generate_trajectory_on_device(vec_env, agent)
        total = rew.sum(dim=-1).reshape(-1).double()
        q_t = obs[:, INVENTORY_INDEX, -1].double()
        half_spreads = act.double().mean(dim=(-1, -2))
        stats = [2 * half_spreads.mean(), total.mean(), total.std(unbiased=False), q_t.mean(), q_t.std(unbiased=False)]
        stats = [float(v) for v in torch.stack(stats).cpu()]
        total = total.cpu().numpy()
    else:
        obs, act, rew = generate_trajectory(vec_env, agent)
        total = rew.sum(axis=-1, dtype=np.float64).reshape(-1)
        q_t = obs[:, INVENTORY_INDEX, -1].astype(np.float64)
        half_spreads = act.mean(axis=(-1, -2), dtype=np.float64)
        stats = [2 * np.mean(half_spreads), np.mean(total), np.std(total), np.mean(q_t), np.std(q_t)]
    return dict(zip(COLUMNS, stats)), total


def generate_results_table_and_hist(vec_env, agent, n_episodes: int = 1000):
    """(results DataFrame with the reference's row and columns, histogram figure or None, total rewards)."""
    import pandas as pd

    assert vec_env.num_trajectories > 1, "To generate a results table and hist, vec_env must roll out > 1 trajectory."
    stats, total_rewards = episode_statistics(vec_env, agent)
    results = pd.DataFrame(index=["Inventory"], columns=COLUMNS, dtype=np.float64)
    for name, value in stats.items():
        results.loc["Inventory", name] = value
    return results, _plot_pnl(total_rewards), total_rewards


def _torch_sees_the_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001 - torch is plumbing, never a requirement
        return False


def _plot_pnl(rewards):
    from mbt_gym_amd.gym.helpers.plotting import plot_pnl  # (matplotlib only; None without it)

    return plot_pnl(rewards)
