"""A trained Stable-Baselines3 model as an Agent (reference: mbt_gym/agents/SbAgent.py:8-26).

A consumer of the environment: `get_action` is `model.predict` on the (optionally column-reduced) observation, on the
host, like the reference.  stable_baselines3 itself is not imported here - any object with `predict(obs,
deterministic=True) -> (actions, state)`, `action_space` and (for `train`) `learn(total_timesteps=...)` serves.

Beyond the reference: when the model's actor is the shape the in-kernel policy evaluates - SB3's `MlpPolicy` with
`net_arch` of two equal hidden layers of width <= 64 and Tanh or ReLU activations, i.e. `policy.mlp_extractor.policy_net`
= [Linear, act, Linear, act] and `policy.action_net` = Linear - `device_policy()` hands its weights to the fused rollout
(csrc/policy_mlp.hpp: matrix cores, fp16 operands), so `generate_trajectory(env, SbAgent(model))` and the results table
run a whole episode in one launch instead of one `predict` + one `step` per time step.  The deterministic action is the
actor's mean clipped to the action space, which is what `predict(deterministic=True)` returns for a Box action space."""
import numpy as np

from mbt_gym_amd.agents.Agent import Agent


def _to_numpy(tensor) -> np.ndarray:
    return np.asarray(tensor.detach().cpu().numpy() if hasattr(tensor, "detach") else tensor, dtype=np.float32)


FP16_OBSERVATION_BOUND = 4.0  # csrc/policy_mlp.hpp: kMlpMaxObservationBound


def observations_fit_fp16(env) -> bool:
    """True when the in-kernel MLP may read `env`'s observations: they are normalised to [-1, 1] (TE:112-118), or every column
    of the raw Box is bounded by FP16_OBSERVATION_BOUND (the library refuses anything else: mbt_env.hip, prepare_learned_policy).
    An object that is not one of our environments (no observation Box to inspect) is given the benefit of the doubt - the
    library checks again when the policy is handed over."""
    if env is None or not hasattr(env, "original_observation_space"):
        return True
    if getattr(env, "normalise_observation_space_", False):
        return True
    space = env.original_observation_space
    return bool(np.all(np.maximum(np.abs(space.low), np.abs(space.high)) <= FP16_OBSERVATION_BOUND))


class SbAgent(Agent):
    def __init__(self, model, reduced_training_indices: list = None, num_trajectories: int = None):
        self.model = model
        self.num_trajectories = num_trajectories or model.env.num_trajectories
        self.num_actions = model.action_space.shape[0]
        self.reduced_training = reduced_training_indices is not None
        if self.reduced_training:
            self.reduced_training_indices = reduced_training_indices

    def get_action(self, state: np.ndarray) -> np.ndarray:
        observed = state[:, self.reduced_training_indices] if self.reduced_training else state
        actions, _ = self.model.predict(observed, deterministic=True)
        return np.asarray(actions).reshape(observed.shape[0], self.num_actions)

    def train(self, total_timesteps: int = 100000):
        self.model.learn(total_timesteps=total_timesteps)

    # ---- the device route --------------------------------------------------------------------------------------
    def actor_layers(self, observation_dim: int = None):
        """[(W1, b1), (W2, b2), (W3, b3)], activation name - or raises ValueError when the actor is not a
        [Linear, act, Linear, act] + Linear network of width <= 64.  With `reduced_training_indices` the first layer is
        widened to the full observation (zero columns for what the model does not see)."""
        policy = getattr(self.model, "policy", None)
        net = getattr(getattr(policy, "mlp_extractor", None), "policy_net", None)
        head = getattr(policy, "action_net", None)
        if net is None or head is None:
            raise ValueError("the model has no policy.mlp_extractor.policy_net / policy.action_net (not an SB3 MlpPolicy actor)")
        modules = list(net)
        if len(modules) != 4 or not all(hasattr(m, "weight") for m in (modules[0], modules[2], head)):
            raise ValueError("the in-kernel policy evaluates two hidden layers: net_arch must be [H, H]")
        kinds = {type(modules[1]).__name__, type(modules[3]).__name__}
        if kinds == {"Tanh"}:
            activation = "tanh"
        elif kinds == {"ReLU"}:
            activation = "relu"
        else:
            raise ValueError(f"activations {sorted(kinds)}: the in-kernel policy evaluates Tanh or ReLU")
        layers = [(_to_numpy(m.weight), _to_numpy(m.bias)) for m in (modules[0], modules[2], head)]
        hidden = layers[0][0].shape[0]
        if hidden > 64 or layers[1][0].shape != (hidden, hidden):
            raise ValueError(f"hidden layers {layers[0][0].shape[0]}, {layers[1][0].shape[0]}: the in-kernel policy evaluates [H, H] with H <= 64")
        if self.reduced_training:
            if observation_dim is None:
                raise ValueError("reduced_training_indices: pass the environment's observation_dim")
            w1 = np.zeros((hidden, observation_dim), np.float32)
            w1[:, self.reduced_training_indices] = layers[0][0]
            layers[0] = (w1, layers[0][1])
        return layers, activation

    @property
    def has_device_policy(self) -> bool:
        """Can the fused rollout evaluate this actor AND read this environment's observations?  The matrix cores take the
        observation row as fp16 (11 bits, nothing beyond 65504): fine for normalised observations, not for a raw midprice of
        100 or raw cash - there `generate_trajectory` and the results table keep the host loop (`model.predict` in float32)."""
        try:
            self.actor_layers(observation_dim=max(self.reduced_training_indices) + 1 if self.reduced_training else None)
        except ValueError:
            return False
        env = getattr(self.model, "env", None)
        return observations_fit_fp16(getattr(env, "env", env))

    def device_policy(self, deterministic: bool = True):
        """deterministic=False adds the policy's exploration noise (SB3: std = exp(policy.log_std), state independent) in the
        kernel - what `model.predict(deterministic=False)` / rollout collection samples - for data collection at device speed."""
        from mbt_gym_amd import _native

        env = getattr(self.model, "env", None)
        dim = getattr(getattr(env, "env", env), "observation_dim", None) or (getattr(env, "observation_space", None).shape[0] if env is not None else None)
        layers, activation = self.actor_layers(observation_dim=dim)
        std = None
        if not deterministic:
            log_std = getattr(getattr(self.model, "policy", None), "log_std", None)
            if log_std is None:
                raise ValueError("the model's policy has no log_std: a stochastic device policy needs the exploration std")
            std = np.exp(_to_numpy(log_std)).reshape(-1)
        return _native.mlp_policy(layers, activation, action_std=std, clip=True)
