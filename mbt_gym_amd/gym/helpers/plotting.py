"""The import path the reference's notebooks use for the episode summary (mbt_gym/gym/helpers/plotting.py:94-114).

`generate_results_table_and_hist` - the results table, the histogram figure, the total reward per trajectory - is on the
path's far side and lives in `results.py`, reduced on the device; `get_timestamps` is the time grid of an
episode.  The figure uses matplotlib alone (seaborn is optional upstream styling): None when matplotlib is absent.  The
single-trajectory line plots of the reference (plot_trajectory, plot_stable_baselines_actions) are notebook conveniences
and are not rebuilt."""
import numpy as np

from mbt_gym_amd.gym.helpers.results import COLUMNS, episode_statistics, generate_results_table_and_hist  # noqa: F401


def get_timestamps(env):
    return np.linspace(0, env.terminal_time, env.n_steps + 1)


def plot_pnl(rewards, symmetric_rewards=None):
    """Density histogram(s) of total rewards, 50 bins (plotting.py:84-91)."""
    try:
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001
        return None
    fig, ax = plt.subplots(1, 1, figsize=(20, 10))
    if symmetric_rewards is not None:
        ax.hist(np.asarray(symmetric_rewards), bins=50, density=True, alpha=0.6, label="Rewards of symmetric strategy")
    ax.hist(np.asarray(rewards), bins=50, density=True, alpha=0.6, color="red", label="Rewards")
    ax.legend()
    plt.close(fig)
    return fig
