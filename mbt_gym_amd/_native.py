"""ctypes binding of include/mbt_env.h (libmbtenv.so).  This is the only module that touches the C ABI.

There is deliberately no alternative implementation behind it: if the shared library is missing or no gfx950
device is visible, environment construction raises - it never degrades to a CPU path.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

# Kernel arguments in device memory: this ROCm stack's default, spelled out for one whose default differs - and only if the HIP runtime has not
# started yet and the user has not said otherwise.  With the arguments in host memory every step kernel waits 1.2-3 us longer for its 1.3 KB of
# them (profiles/r06_kernarg_layout.txt).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

# MBT_LIBRARY_VARIANT=asan / tsan: the sanitizer build of the host side (mbt_gym_amd/build.py, MBT_SANITIZE; README "Sanitizers")
_VARIANT = os.environ.get("MBT_LIBRARY_VARIANT", "").strip()
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libmbtenv.{_VARIANT}.so" if _VARIANT else "libmbtenv.so")
if _VARIANT and "MBT_EXTRA_HIPCC_FLAGS" not in os.environ:
    # an experiment build (tools/dbg/build_variant.py) leaves the flags it was compiled with beside it: they are part of the source hash the
    # staleness check below compares, so `MBT_LIBRARY_VARIANT=name` alone selects such a library
    _flags = os.path.join(os.path.dirname(LIB_PATH), f"libmbtenv.{_VARIANT}.flags")
    if os.path.exists(_flags):
        os.environ["MBT_EXTRA_HIPCC_FLAGS"] = open(_flags).read().strip()
ABI_VERSION = 8

MID_BROWNIAN, MID_OU, MID_GBM, MID_BROWNIAN_JUMP, MID_OU_JUMP, MID_CONSTANT, MID_LINEAR_SDE, MID_USER, MID_HOST = 0, 1, 2, 3, 4, 5, 6, 7, 8
ARR_POISSON, ARR_HAWKES, ARR_POISSON_NONLINEAR, ARR_NONE, ARR_USER, ARR_HOST = 0, 1, 2, 3, 4, 5
FILL_EXPONENTIAL, FILL_NONE, FILL_EXOGENOUS_MM, FILL_USER, FILL_HOST = 0, 1, 2, 3, 4
DYN_LIMIT, DYN_LIMIT_AND_MARKET, DYN_AT_THE_TOUCH, DYN_SPEED = 0, 1, 2, 3
REW_PNL, REW_RUNNING_PENALTY, REW_CJ_MM, REW_EXP_UTILITY, REW_CJ_OE, REW_USER, REW_HOST = 0, 1, 2, 3, 4, 5, 6
IMPACT_NONE, IMPACT_TEMPORARY_POWER, IMPACT_TEMPORARY_AND_PERMANENT, IMPACT_TEMPORARY_AND_TRANSIENT, IMPACT_TRANSIENT, IMPACT_HOST, IMPACT_HOST_STATE = -1, 0, 1, 2, 3, 4, 5
NOISE_PHILOX, NOISE_INJECTED = 0, 1


class MbtConfig(C.Structure):
    """struct mbt_config (include/mbt_env.h) - field order and types must match the header."""

    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32),
        ("num_trajectories", C.c_uint64), ("trajectory_offset", C.c_uint64),
        ("n_steps", C.c_uint32), ("reserved0", C.c_uint32),
        ("terminal_time", C.c_double),
        ("midprice_kind", C.c_int32), ("arrival_kind", C.c_int32), ("fill_kind", C.c_int32),
        ("dynamics_kind", C.c_int32), ("reward_kind", C.c_int32), ("noise_mode", C.c_int32),
        ("drift", C.c_double), ("volatility", C.c_double), ("initial_price", C.c_double),
        ("ou_level", C.c_double), ("ou_speed", C.c_double),
        ("intensity", C.c_double * 2),
        ("hawkes_jump", C.c_double), ("hawkes_speed", C.c_double),
        ("fill_exponent", C.c_double), ("market_half_spread", C.c_double),
        ("phi", C.c_double), ("alpha", C.c_double), ("inventory_exponent", C.c_double),
        ("initial_cash", C.c_double), ("initial_inventory", C.c_double),
        ("max_inventory", C.c_double), ("max_cash", C.c_double), ("reward_scale", C.c_double),
        ("seed", C.c_uint64),
        ("normalise_observation", C.c_int32), ("normalise_action", C.c_int32),
        ("obs_lo", C.c_float * 8), ("obs_hi", C.c_float * 8),
        ("act_lo", C.c_float * 4), ("act_hi", C.c_float * 4),
        ("midprice_step_size", C.c_double), ("arrival_step_size", C.c_double),
        ("jump_size", C.c_double), ("risk_aversion", C.c_double),
        ("impact_kind", C.c_int32), ("reserved1", C.c_int32),
        ("temporary_impact", C.c_double), ("impact_exponent", C.c_double), ("permanent_impact", C.c_double),
        ("transient_impact", C.c_double), ("resilience", C.c_double), ("initial_transient_impact", C.c_double),
        ("kernel_coefficient", C.c_double), ("impact_step_size", C.c_double),
        ("exogenous_depth", C.c_double * 2), ("base_fill_probability", C.c_double),
        ("reward_terminal_time", C.c_double), ("mid_coef_add", C.c_double), ("mid_coef_mul", C.c_double),
        ("precise_state", C.c_int32), ("allow_stiff_hawkes", C.c_int32),
        ("hawkes_float32_intensities", C.c_int32), ("resident_step", C.c_int32),  # ABI 7
    ]


class MbtUserCode(C.Structure):
    """struct mbt_user_code (include/mbt_env.h): device expressions of user-defined plugins."""

    _fields_ = [("fill_probability", C.c_char_p), ("fill_param_names", C.c_char_p), ("fill_params", C.c_double * 8),
                ("reward", C.c_char_p), ("reward_param_names", C.c_char_p), ("reward_params", C.c_double * 8),
                ("arrival_probability", C.c_char_p), ("arrival_param_names", C.c_char_p), ("arrival_params", C.c_double * 8),
                ("midprice_increment", C.c_char_p), ("midprice_param_names", C.c_char_p), ("midprice_params", C.c_double * 8),
                ("state_columns", C.c_int32), ("extra_normals", C.c_int32), ("state_update", C.c_char_p * 2),
                ("state_param_names", C.c_char_p), ("state_params", C.c_double * 8), ("state_initial", C.c_double * 2), ("state_owner", C.c_int32 * 2)]


def user_code(fill=None, reward=None, arrival=None, midprice=None, state=None) -> MbtUserCode:
    """(expression, {name: value}) pairs -> struct mbt_user_code.  `state`: (list of 1-2 update expressions, {name: value},
    initial values, uses_extra_normals, owners (0 midprice / 1 arrival model per column)) for the state columns user processes own."""
    code = MbtUserCode()
    if state is not None:
        updates, params, initial, extra_normals, owners = state
        if not 1 <= len(updates) <= 2 or len(initial) != len(updates):
            raise ValueError("user processes own one or two state columns, each with an update expression and an initial value")
        if len(params) > 8:
            raise ValueError("a device expression takes at most 8 parameters")
        code.state_columns, code.extra_normals = len(updates), int(bool(extra_normals))
        for j, (expression, x0) in enumerate(zip(updates, initial)):
            code.state_update[j] = expression.encode()
            code.state_initial[j] = float(x0)
            code.state_owner[j] = int(owners[j])
        code.state_param_names = ",".join(params).encode()
        for j, value in enumerate(params.values()):
            code.state_params[j] = float(value)
    for prefix, part in (("fill", fill), ("reward", reward), ("arrival", arrival), ("midprice", midprice)):
        if part is None:
            continue
        expression, params = part
        if len(params) > 8:
            raise ValueError("a device expression takes at most 8 parameters")
        setattr(code, {"fill": "fill_probability", "reward": "reward", "arrival": "arrival_probability", "midprice": "midprice_increment"}[prefix], expression.encode())
        setattr(code, prefix + "_param_names", ",".join(params).encode())
        for j, value in enumerate(params.values()):
            getattr(code, prefix + "_params")[j] = float(value)
    return code


CLOCK_AUTO_RESET, CLOCK_TERMINAL_OBSERVATION = 1, 2


class MbtDeviceClock(C.Structure):
    """struct mbt_device_clock (include/mbt_env.h): the first 32 bytes of the device's clock block."""

    _fields_ = [("time", C.c_double), ("episode_step", C.c_uint32), ("philox_step", C.c_uint32), ("steps", C.c_uint32),
                ("episodes", C.c_uint32), ("done", C.c_int32), ("log_count", C.c_uint32)]


POLICY_FIXED, POLICY_AVELLANEDA_STOIKOV, POLICY_TIME_INVENTORY_TABLE, POLICY_TIME_TABLE, POLICY_ACTION_BUFFER = 0, 1, 2, 3, 4
POLICY_LINEAR, POLICY_MLP = 5, 6
ACTIVATION_TANH, ACTIVATION_RELU = 0, 1


class MbtPolicy(C.Structure):
    """struct mbt_policy (include/mbt_env.h)."""

    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("params", C.c_double * 8),
                ("table", C.POINTER(C.c_float)), ("table_rows", C.c_uint32), ("table_cols", C.c_uint32),
                ("table_q_offset", C.c_int32), ("reserved1", C.c_int32)]


def table_policy(table: np.ndarray, q_offset: int) -> MbtPolicy:
    """(rows = time steps, cols = inventory levels, 2) float32 depths -> policy descriptor (keeps the array alive)."""
    arr = np.ascontiguousarray(table, dtype=np.float32)
    assert arr.ndim == 3 and arr.shape[2] == 2
    pol = MbtPolicy(kind=POLICY_TIME_INVENTORY_TABLE, table=arr.ctypes.data_as(C.POINTER(C.c_float)),
                    table_rows=arr.shape[0], table_cols=arr.shape[1], table_q_offset=int(q_offset))
    pol._keepalive = arr
    return pol


def _exploration(pol: MbtPolicy, n_actions: int, action_std, clip: bool):
    """params[1] = clip to the action space; params[2..2+A) = exploration std per action component (scalar or vector)."""
    pol.params[1] = 1.0 if clip else 0.0
    if action_std is not None:
        std = np.broadcast_to(np.asarray(action_std, dtype=np.float64), (n_actions,))
        for j in range(n_actions):
            pol.params[2 + j] = float(std[j])


def linear_policy(weight: np.ndarray, bias: np.ndarray, action_std=None, clip: bool = True) -> MbtPolicy:
    """action = clip(weight @ obs + bias [+ action_std * N(0, 1)]): weight (A, D), bias (A) in torch.nn.Linear layout (keeps
    the blob alive)."""
    w, b = np.asarray(weight, dtype=np.float32), np.asarray(bias, dtype=np.float32)
    assert w.ndim == 2 and b.shape == (w.shape[0],)
    blob = np.ascontiguousarray(np.concatenate([w.reshape(-1), b]))
    pol = MbtPolicy(kind=POLICY_LINEAR, table=blob.ctypes.data_as(C.POINTER(C.c_float)), table_rows=0, table_cols=blob.size)
    _exploration(pol, w.shape[0], action_std, clip)
    pol._keepalive = blob
    return pol


def mlp_policy(layers, activation: str = "tanh", action_std=None, clip: bool = True) -> MbtPolicy:
    """A two-hidden-layer MLP actor evaluated inside the kernels on the matrix cores: `layers` = [(W1, b1), (W2, b2), (W3, b3)]
    in torch.nn.Linear layout (W: out x in) with W1 (H, D), W2 (H, H), W3 (A, H), H <= 64 - e.g. the `mlp_extractor.policy_net`
    and `action_net` weights of a Stable-Baselines3 MlpPolicy with net_arch [64, 64] (keeps the blob alive)."""
    assert len(layers) == 3, "two hidden layers + the output layer"
    (w1, b1), (w2, b2), (w3, b3) = [(np.asarray(w, dtype=np.float32), np.asarray(b, dtype=np.float32)) for w, b in layers]
    hidden = w1.shape[0]
    assert w2.shape == (hidden, hidden) and w3.shape[1] == hidden and b1.shape == b2.shape == (hidden,) and b3.shape == (w3.shape[0],)
    blob = np.ascontiguousarray(np.concatenate([a.reshape(-1) for a in (w1, b1, w2, b2, w3, b3)]))
    pol = MbtPolicy(kind=POLICY_MLP, table=blob.ctypes.data_as(C.POINTER(C.c_float)), table_rows=hidden, table_cols=blob.size)
    pol.params[0] = {"tanh": ACTIVATION_TANH, "relu": ACTIVATION_RELU}[activation]
    _exploration(pol, w3.shape[0], action_std, clip)  # action_std: exploration noise for data collection (None = deterministic)
    pol._keepalive = blob
    return pol


def schedule_policy(schedule: np.ndarray) -> MbtPolicy:
    """(rows = time steps, A) float32 open-loop actions -> policy descriptor (keeps the array alive)."""
    arr = np.ascontiguousarray(schedule, dtype=np.float32)
    assert arr.ndim == 2
    pol = MbtPolicy(kind=POLICY_TIME_TABLE, table=arr.ctypes.data_as(C.POINTER(C.c_float)), table_rows=arr.shape[0], table_cols=arr.shape[1])
    pol._keepalive = arr
    return pol


class NativeError(RuntimeError):
    """A libmbtenv call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"libmbtenv error {code}: {message}")
        self.code = code


_F = C.POINTER(C.c_float)
_ENV = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/mbt_env.h
SIGNATURES = {
    "mbt_abi_version": (C.c_uint32, []),
    "mbt_config_sizeof": (C.c_size_t, []),
    "mbt_source_hash": (C.c_char_p, []),
    "mbt_last_error": (C.c_char_p, []),
    "mbt_device_count": (C.c_int, []),
    "mbt_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "mbt_host_alloc": (C.c_void_p, [C.c_size_t]),
    "mbt_host_free": (None, [C.c_void_p]),
    "mbt_env_create": (C.c_int, [C.POINTER(MbtConfig), C.POINTER(_ENV)]),
    "mbt_env_destroy": (None, [_ENV]),
    "mbt_env_create_jit": (C.c_int, [C.POINTER(MbtConfig), C.POINTER(MbtUserCode), C.POINTER(_ENV)]),
    "mbt_jit_log": (C.c_char_p, []),
    "mbt_jit_check": (C.c_int, [C.POINTER(MbtConfig), C.POINTER(MbtUserCode)]),
    "mbt_env_set_stream": (C.c_int, [_ENV, C.c_void_p]),
    "mbt_env_synchronize": (C.c_int, [_ENV]),
    "mbt_env_set_step_size": (C.c_int, [_ENV, C.c_double]),
    "mbt_env_seed": (C.c_int, [_ENV, C.c_uint64]),
    "mbt_env_reset": (C.c_int, [_ENV, C.c_double, _F]),
    "mbt_env_reset_host": (C.c_int, [_ENV, C.c_double, _F, _F]),
    "mbt_env_step_host": (C.c_int, [_ENV, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),  # (float*: addresses are passed as they are)
    "mbt_env_step_device": (C.c_int, [_ENV, C.c_void_p, C.POINTER(C.c_int32)]),
    "mbt_env_set_launch_gate": (C.c_int, [_ENV, C.c_uint32]),
    "mbt_env_device_clock_begin": (C.c_int, [_ENV, C.c_uint32]),
    "mbt_env_step_device_captured": (C.c_int, [_ENV, C.c_void_p]),
    "mbt_env_device_clock_read": (C.c_int, [_ENV, C.c_void_p]),
    "mbt_env_device_clock_ptr": (C.c_void_p, [_ENV]),
    "mbt_env_terminal_obs_ptr": (C.c_void_p, [_ENV]),
    "mbt_env_device_clock_end": (C.c_int, [_ENV]),
    "mbt_env_host_depths": (C.c_int, [_ENV, _F, C.POINTER(C.c_double)]),
    "mbt_env_set_host_fill_probabilities": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_env_set_host_arrivals": (C.c_int, [_ENV, _F]),
    "mbt_env_set_host_impacts": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_env_set_host_rewards": (C.c_int, [_ENV, C.POINTER(C.c_double), _F]),
    "mbt_env_host_step_outputs": (C.c_int, [_ENV, C.POINTER(C.c_double), C.POINTER(C.c_uint8)]),
    "mbt_env_set_host_state_columns": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_env_step_many_device": (C.c_int, [_ENV, C.c_uint32, C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mbt_env_rollout_device": (C.c_int, [_ENV, C.POINTER(MbtPolicy), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "mbt_env_rollout_host": (C.c_int, [_ENV, C.POINTER(MbtPolicy), C.c_uint32, _F, _F, _F, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "mbt_env_release_staging": (C.c_int, [_ENV]),
    "mbt_env_policy_device": (C.c_int, [_ENV, C.POINTER(MbtPolicy)]),
    "mbt_env_padded_lanes": (C.c_uint64, [_ENV]),
    "mbt_env_record_floor_device": (C.c_int, [_ENV, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mbt_env_set_noise_host": (C.c_int, [_ENV, _F, _F, _F]),
    "mbt_env_set_user_noise_host": (C.c_int, [_ENV, _F]),
    "mbt_env_action_ptr": (C.c_void_p, [_ENV]),
    "mbt_env_obs_ptr": (C.c_void_p, [_ENV]),
    "mbt_env_state_in_place": (C.c_int, [_ENV]),
    "mbt_env_reward_ptr": (C.c_void_p, [_ENV]),
    "mbt_env_obs_dim": (C.c_int, [_ENV]),
    "mbt_env_action_dim": (C.c_int, [_ENV]),
    "mbt_env_get_state_host": (C.c_int, [_ENV, _F]),
    "mbt_env_get_state_f64_host": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_exact_split": (None, [C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "mbt_power_f32_device": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.c_double, C.POINTER(C.c_float), C.c_uint32]),
    "mbt_exact_join": (C.c_double, [C.c_float, C.c_int32]),
    "mbt_env_get_obs_host": (C.c_int, [_ENV, _F]),
    "mbt_env_set_action_host": (C.c_int, [_ENV, _F]),
    "mbt_env_set_state_host": (C.c_int, [_ENV, _F, C.c_double, C.c_uint32]),
    "mbt_env_get_clock": (C.c_int, [_ENV, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mbt_env_record_events": (C.c_int, [_ENV, C.c_int]),
    "mbt_env_get_events_host": (C.c_int, [_ENV, C.POINTER(C.c_uint8)]),
    "mbt_env_clip_count": (C.c_int, [_ENV, C.POINTER(C.c_uint64)]),
    "mbt_env_track_lane_returns": (C.c_int, [_ENV, C.c_int]),
    "mbt_env_return_sums": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_env_return_sums_begin": (C.c_int, [_ENV]),
    "mbt_env_return_sums_end": (C.c_int, [_ENV, C.POINTER(C.c_double)]),
    "mbt_env_episode_log_pop": (C.c_int, [_ENV, C.POINTER(C.c_double), C.c_int32]),
    "mbt_env_allreduce_returns": (C.c_int, [_ENV, C.c_void_p, C.POINTER(C.c_double)]),
    "mbt_env_set_communicator": (C.c_int, [_ENV, C.c_void_p]),
    "mbt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mbt_comm_init_rank": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mbt_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mbt_comm_destroy": (C.c_int, [C.c_void_p]),
    "mbt_reward_calculate_host": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]),
    "mbt_process_evaluate_host": (C.c_int, [C.c_int, C.c_int, C.POINTER(MbtConfig), C.c_uint64] + [C.POINTER(C.c_double)] * 5),
    "mbt_rng_fill_host": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, _F, _F, _F]),
    "mbt_rng_fill_quad_host": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, _F]),
    "mbt_rng_fill_user_host": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, _F]),
    "mbt_philox4x32_10_host": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mbt_env_timer_begin": (C.c_int, [_ENV]),
    "mbt_env_timer_end": (C.c_int, [_ENV, C.POINTER(C.c_float)]),
    "mbt_env_timer_stop": (C.c_int, [_ENV]),
    "mbt_env_timer_elapsed": (C.c_int, [_ENV, C.POINTER(C.c_float)]),
}

_lib = None


COMM_ID_BYTES = 128


def _torch_lib(name):
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
    return path if os.path.exists(path) else None


def preload_torch_rccl():
    """Same reasoning as for the HIP runtime below, for RCCL: PyTorch bundles its own librccl.so; libmbtenv binds RCCL with
    dlopen at first use and must find THAT copy when torch is installed (two RCCL copies over one HIP runtime do not mix).
    Called before the first communicator is made; harmless without torch (the system's /opt/rocm copy is used)."""
    path = _torch_lib("librccl.so")
    if path is None:
        return None
    os.environ.setdefault("MBT_RCCL_LIBRARY", path)
    # local scope (the default mode): with RTLD_GLOBAL the librocm_smi64 that librccl depends on lends its amd::smi globals
    # to /opt/rocm/lib/libamd_smi.so, which RCCL opens later, and the process aborts at exit in a double free
    return C.CDLL(path)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two copies of the HIP
    runtime in one process cannot share the GPU ("No HIP GPUs are available" in whichever initialises second), so
    when torch is installed its copy is loaded first and libmbtenv.so binds to it - regardless of import order.
    torch itself is not imported."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(bundled):
        return None
    return C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def load_library():
    """dlopen libmbtenv.so and bind every symbol of the header.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m mbt_gym_amd.build). "
            "mbt_gym_amd has no CPU fallback."
        )
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mbt_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libmbtenv ABI {lib.mbt_abi_version()} != binding {ABI_VERSION}")
    if lib.mbt_config_sizeof() != C.sizeof(MbtConfig):
        raise RuntimeError(f"struct mbt_config is {lib.mbt_config_sizeof()} bytes in the library, {C.sizeof(MbtConfig)} in the binding")
    from mbt_gym_amd import build as _build

    if os.path.isdir(_build.CSRC):  # sources present (a checkout, not an installed wheel): the library must be THEIR build
        built_from, present = lib.mbt_source_hash().decode(), _build.source_hash()
        if built_from != present:
            raise RuntimeError(
                f"{LIB_PATH} is stale: built from sources {built_from[:12]}..., the tree holds {present[:12]}...  "
                "Rebuild it (python -m mbt_gym_amd.build); a stale library is refused rather than silently used.")
    _lib = lib
    return lib


def check(code):
    if code < 0:
        lib = load_library()
        message = lib.mbt_last_error().decode("utf-8", "replace")
        if "mbt_jit_log" in message:  # a user expression that does not compile: show the compiler's diagnostics
            message += "\n" + lib.mbt_jit_log().decode("utf-8", "replace")
        raise NativeError(code, message)
    return code


def fptr(array):
    """float32 C-contiguous numpy array -> float*; None -> NULL."""
    if array is None:
        return None
    assert array.dtype == np.float32 and array.flags["C_CONTIGUOUS"]
    return array.ctypes.data_as(_F)


def as_f32(a, shape=None):
    out = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {out.shape}")
    return out


class PinnedBuffer:
    """A block of pinned host memory from libmbtenv (mbt_host_alloc) shaped as a float32 array: what the host API's DMA
    copies read and write directly.  `array()` makes a NumPy array over it whose base chain ends HERE, so the number of
    references to this object tells whether any array (or view, or torch tensor made from one) still looks at the block -
    which is how TradingEnvironment.step() re-uses its output buffers without ever overwriting one a caller still holds."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = load_library().mbt_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise NativeError(-3, load_library().mbt_last_error().decode("utf-8", "replace"))

    @property
    def __array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3, "strides": None}

    def array(self) -> np.ndarray:
        return np.asarray(self)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr and _lib is not None:
            try:
                _lib.mbt_host_free(ptr)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass


def _reference_counts_are_exact() -> bool:
    """OutputPool decides that a buffer is idle from `sys.getrefcount`.  That is exact on standard (GIL) CPython only: on
    PyPy, or a free-threaded build with deferred / biased counts, it says nothing reliable - there every output is a fresh
    array, like the reference's.  MBT_FRESH_OUTPUTS=1 asks for that behaviour everywhere (see TradingEnvironment.step)."""
    if os.environ.get("MBT_FRESH_OUTPUTS", "0") not in ("", "0"):
        return False
    if sys.implementation.name != "cpython" or not hasattr(sys, "getrefcount"):
        return False
    gil_enabled = getattr(sys, "_is_gil_enabled", None)
    return True if gil_enabled is None else bool(gil_enabled())


class OutputPool:
    """Output buffers of the host API (observations, rewards, dones) in pinned memory, re-used across steps WITHOUT changing
    what the caller sees: a buffer is handed out again only once no array over it is referenced any more (the reference
    returns fresh arrays, TE:101, TE:110 - a caller that keeps one keeps its values).  In the usual loop
    `obs, rew, done, info = env.step(action)` the previous step's arrays die when the names are rebound, so two buffers
    alternate and nothing is allocated after the first two steps (a fresh 16 MB np.empty per step - and the page faults of
    the DMA that fills it - were most of a 2^20-lane step).  A caller that holds on to many outputs makes the pool grow, up to
    `max_bytes` of pinned memory; beyond that it gets ordinary (pageable) arrays, as before.

    What "referenced" means: PYTHON references - arrays, views, torch tensors made from them - which is what the count sees.
    A consumer that keeps only a raw ADDRESS (`obs.ctypes.data`, a cffi / C-extension pointer cache) and drops the array is
    invisible to it and will see the buffer overwritten two steps later; such a caller keeps a reference to the array, or
    sets MBT_FRESH_OUTPUTS=1 (every output a fresh array, the reference's behaviour to the letter, at the cost described
    above).  Where reference counts are not exact (not standard GIL CPython) that mode is the only one.

    Memory: page-locked host memory is returned - `release()`, called by TradingEnvironment.close() - and does not pile up:
    buffers beyond `min_buffers` that stayed idle for `idle_seconds` are freed the next time the pool is used."""

    def __init__(self, shape, dtype=np.float32, max_bytes=1 << 30, min_buffers=2, idle_seconds=30.0):
        self.shape, self.dtype, self.max_bytes, self.min_buffers = tuple(shape), np.dtype(dtype), max_bytes, min_buffers
        self.idle_seconds = idle_seconds
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._pinned_unavailable = not _reference_counts_are_exact()
        # One MASTER array per pinned block: its base is the PinnedBuffer (which frees the block when the last array over it is
        # gone), and every array handed out is a view of it - NumPy points the base of a view, of a view of a view, ... at the
        # master, so the master's reference count says whether anything still looks at the block.  (A view costs 0.13 us; a new
        # array over the block through __array_interface__ 0.8 us - three of those were 2 of the ~16 us of a small step.)
        self.buffers = [_RefProbe()]
        self._idle_refs = self._refs(0) if not self._pinned_unavailable else 0  # references a master has when only this pool looks at it (counted exactly as acquire() counts)
        self.buffers = []
        self.pointers = []
        self._last_used = []

    def _refs(self, index):
        return sys.getrefcount(self.buffers[index])

    def acquire(self):
        """(array, is_pinned): an array nobody else references, preferably over pinned memory."""
        array, pointer = self.acquire_with_pointer()
        return array, pointer is not None

    def acquire_with_pointer(self):
        """(array, address of its pinned block or None for an ordinary array)."""
        buffers, idle_refs, getrefcount = self.buffers, self._idle_refs, sys.getrefcount
        for index in range(len(buffers)):  # the first idle one: the usual loop then alternates between buffers 0 and 1
            if getrefcount(buffers[index]) <= idle_refs:
                if len(buffers) > self.min_buffers:
                    self._last_used[index] = time.monotonic()
                    self._trim()  # (only ever removes buffers ABOVE the first idle one)
                return buffers[index].view(), self.pointers[index]
        if not self._pinned_unavailable and (len(buffers) + 1) * self.nbytes <= max(self.max_bytes, self.min_buffers * self.nbytes):
            try:
                block = PinnedBuffer(self.shape, self.dtype)
                master = block.array()
            except (NativeError, RuntimeError, OSError):  # no pinned memory to be had (no device, a locked-memory limit):
                self._pinned_unavailable = True          # pageable arrays work everywhere, only slower
            else:
                buffers.append(master)
                self.pointers.append(block.ptr)
                self._last_used.append(time.monotonic())
                del block  # (the master's base keeps it)
                return master.view(), self.pointers[-1]
        return np.empty(self.shape, dtype=self.dtype), None

    def _trim(self):
        """Frees buffers beyond `min_buffers` that nobody references and that have not been handed out for `idle_seconds`
        (a caller that once held many outputs - two recorded episodes, say - does not pin that memory for good)."""
        now = time.monotonic()
        for index in range(len(self.buffers) - 1, self.min_buffers - 1, -1):
            if index < len(self.buffers) and self._refs(index) <= self._idle_refs and now - self._last_used[index] > self.idle_seconds:
                del self.buffers[index], self.pointers[index], self._last_used[index]  # PinnedBuffer.__del__ returns the block (mbt_host_free)

    def release(self):
        """Gives back every buffer no caller references any more (the others go with their last array)."""
        self.buffers, self.pointers, self._last_used = [], [], []

    @property
    def pinned_bytes(self) -> int:
        return len(self.buffers) * self.nbytes


class _RefProbe:
    """(calibrates OutputPool's reference count without allocating anything)"""


class DeviceView:
    """A borrowed (N, D) / (N,) float32 device buffer owned by libmbtenv, exposed through
    `__cuda_array_interface__` so that `torch.as_tensor(view, device="cuda")` wraps it without a copy."""

    def __init__(self, ptr, shape, owner, typestr="<f4"):
        self.ptr, self.shape, self._owner, self.typestr = int(ptr), tuple(shape), owner, typestr

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.typestr, "data": (self.ptr, False), "version": 2, "strides": None}


def reward_calculate(kind, phi, alpha, exponent, current_state, next_state, is_terminal, q_init=None, episode_length=None,
                     action=None, risk_aversion=0.0, device=0):
    """RewardFunction.calculate on float64 state matrices, evaluated on the device (mbt_reward_calculate_host)."""
    cur = np.ascontiguousarray(current_state, dtype=np.float64)
    nxt = np.ascontiguousarray(next_state, dtype=np.float64)
    assert cur.ndim == 2 and cur.shape == nxt.shape, "Reward functions must be calculated on state matrices."
    n, dim = cur.shape
    out = np.empty((n,), dtype=np.float64)
    dptr = lambda a: None if a is None else np.ascontiguousarray(np.broadcast_to(a, (n,)), dtype=np.float64)  # noqa: E731
    qi, ln = dptr(q_init), dptr(episode_length)
    act = None if action is None else dptr(np.squeeze(np.asarray(action, dtype=np.float64)))
    as_p = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    check(load_library().mbt_reward_calculate_host(device, kind, float(phi), float(alpha), float(exponent), as_p(cur), as_p(nxt),
                                                   dim, n, int(bool(is_terminal)), as_p(qi), as_p(ln), as_p(act), float(risk_aversion),
                                                   as_p(out)))
    return out


PROCESS_MIDPRICE_UPDATE, PROCESS_HAWKES_UPDATE, PROCESS_ARRIVALS, PROCESS_FILLS = 0, 1, 2, 3


def process_evaluate(op, params: dict, a, b=None, c=None, d=None, device=0):
    """One update() / get_arrivals() / get_fills() of a plugin object on host arrays, evaluated on the device in double
    (mbt_process_evaluate_host).  `params`: the mbt_config fields of the process (its `device_params()`)."""
    cfg = MbtConfig()
    cfg.abi_version = ABI_VERSION
    for key, value in params.items():
        if value is None:
            continue
        if key in ("intensity", "exogenous_depth"):
            getattr(cfg, key)[0], getattr(cfg, key)[1] = value
        else:
            setattr(cfg, key, value)
    arrays = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (a, b, c, d)]
    n = arrays[0].shape[0]
    out = np.empty_like(arrays[0])
    ptr = lambda x: None if x is None else x.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    check(load_library().mbt_process_evaluate_host(device, op, C.byref(cfg), n, ptr(arrays[0]), ptr(arrays[1]), ptr(arrays[2]), ptr(arrays[3]), ptr(out)))
    return out


def device_count():
    return int(load_library().mbt_device_count())


def device_name(device=0):
    buf = C.create_string_buffer(64)
    check(load_library().mbt_device_name(device, buf, 64))
    return buf.value.decode()


def philox4x32_10(ctr, key, device=0):
    lib = load_library()
    c = (C.c_uint32 * 4)(*[int(x) for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) for x in key])
    out = (C.c_uint32 * 4)()
    check(lib.mbt_philox4x32_10_host(device, c, k, out))
    return [int(x) for x in out]


def rng_fill_quad(seed, trajectory_offset, step, n, device=0):
    """The normals lanes [offset, offset+n) of a speed-dynamics environment draw at one step."""
    z = np.empty((n,), np.float32)
    check(load_library().mbt_rng_fill_quad_host(device, int(seed), int(trajectory_offset), int(step), int(n), fptr(z)))
    return z


def rng_fill_user(seed, trajectory_offset, step, n, device=0):
    """The two extra normals (n, 2) user processes of lanes [offset, offset+n) draw at one step."""
    z = np.empty((n, 2), np.float32)
    check(load_library().mbt_rng_fill_user_host(device, int(seed), int(trajectory_offset), int(step), int(n), fptr(z)))
    return z


def rng_fill(seed, trajectory_offset, step, n, device=0):
    """The production generator's draws for lanes [offset, offset+n) at one step: (u_arr, u_fill, z)."""
    lib = load_library()
    u_arr, u_fill, z = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32), np.empty((n,), np.float32)
    check(lib.mbt_rng_fill_host(device, int(seed), int(trajectory_offset), int(step), int(n), fptr(u_arr), fptr(u_fill), fptr(z)))
    return u_arr, u_fill, z
