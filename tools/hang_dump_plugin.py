"""pytest plugin for tools/soak_watchdog.sh (tools/run_sanitizers.sh, tools/dbg/r06_asan_soak_repro.sh): where is a test that hangs?

Every process (the xdist workers too) lets any process of the same user attach a debugger (prctl PR_SET_PTRACER_ANY: the watchdog's rocgdb is a
sibling, not an ancestor) and arms faulthandler before each test: after MBT_HANG_DUMP_AFTER seconds (default 25) the Python stacks of all threads
go to $MBT_HANG_DUMP_DIR/py_stack.<pid>.txt, the test's id in front, and the test carries on (the watchdog takes the native stacks and ends it).

    PYTHONPATH=tools python -m pytest -p hang_dump_plugin ..."""
import ctypes
import faulthandler
import os

_file = None


def pytest_configure(config):
    global _file
    directory = os.environ.get("MBT_HANG_DUMP_DIR")
    if not directory:
        return
    libc = ctypes.CDLL(None, use_errno=True)
    libc.prctl(0x59616D61, ctypes.c_ulong(-1 & (2**64 - 1)), 0, 0, 0)  # PR_SET_PTRACER, PR_SET_PTRACER_ANY
    _file = open(os.path.join(directory, f"py_stack.{os.getpid()}.txt"), "w")


def pytest_runtest_setup(item):
    if _file is None:
        return
    _file.write(f"\n== {item.nodeid}\n")
    _file.flush()
    faulthandler.dump_traceback_later(float(os.environ.get("MBT_HANG_DUMP_AFTER", "25")), repeat=False, file=_file, exit=False)


def pytest_runtest_teardown(item):
    if _file is not None:
        faulthandler.cancel_dump_traceback_later()
