#!/usr/bin/env python3
"""A / B of the step kernel's three ways of getting its clock, Avellaneda-Stoikov, HIP-event time per step on the environment's stream:
  plain      mbt_env_step_many_device (kernel arguments from the host), state buffers as tune_for_size picks them
  in_place   the same with MBT_PING_PONG_STATE=0 (what device-clock mode forces)
  captured   mbt_env_step_device_captured, one library call per step (the clock on the device)
  graph      a HIP graph of 50 captured steps, replayed
python tools/dbg/r06_captured_ab.py [lanes ...]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mbt_gym_amd import _native  # noqa: E402
from tools.bench_device_loop import _env  # noqa: E402

lib = _native.load_library()
rt = C.CDLL(_native.LIB_PATH)


def hip(name, *args):
    code = getattr(rt, name)(*args)
    assert code == 0, (name, code)


def timed(env, run, steps):
    ms = C.c_float(0)
    run()
    env.synchronize()
    best = 1e9
    for _ in range(5):
        _native.check(lib.mbt_env_timer_begin(env._handle))
        run()
        _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
        best = min(best, ms.value * 1e3 / steps)
    return best


def measure(n, workload="as"):
    out = {"lanes": n}
    steps = max(200, min(4000, (1 << 31) // (n * 8)))
    for name, ping in (("plain", None), ("in_place", "0")):
        if ping is None:
            os.environ.pop("MBT_PING_PONG_STATE", None)
        else:
            os.environ["MBT_PING_PONG_STATE"] = ping
        env = _env(n, 0, False)
        env.reset_device()
        env.set_action_host(np.tile(np.array([[0.7, 0.7]], dtype=np.float32), (n, 1)))
        out[name] = timed(env, lambda: env.step_many_device(steps, auto_reset=True), steps)
        env.close()
    os.environ.pop("MBT_PING_PONG_STATE", None)
    env = _env(n, 0, False)
    env.reset_device()
    env.set_action_host(np.tile(np.array([[0.7, 0.7]], dtype=np.float32), (n, 1)))
    stream = C.c_void_p()
    hip("hipStreamCreate", C.byref(stream))
    env.set_stream(stream.value)
    env.device_clock_begin(auto_reset=True)
    step = lib.mbt_env_step_device_captured

    def eager():
        for _ in range(steps):
            step(env._handle, None)

    out["captured"] = timed(env, eager, steps)
    graph, executable = C.c_void_p(), C.c_void_p()
    hip("hipStreamBeginCapture", stream, 0)
    for _ in range(50):
        step(env._handle, None)
    hip("hipStreamEndCapture", stream, C.byref(graph))
    hip("hipGraphInstantiate", C.byref(executable), graph, None, None, C.c_size_t(0))

    def replay():
        for _ in range(steps // 50):
            hip("hipGraphLaunch", executable, stream)

    out["graph"] = timed(env, replay, steps // 50 * 50)
    env.device_clock_end()
    env.close()
    return out


if __name__ == "__main__":
    lanes = [int(x) for x in sys.argv[1:]] or [1000, 1 << 16, 1 << 18, 1 << 20, 1 << 22]
    for n in lanes:
        print(json.dumps(measure(n)), flush=True)
