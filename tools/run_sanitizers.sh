#!/bin/bash
# Sanitizer builds of the host side of libmbtenv over the GPU tests that exercise it hardest (README "Sanitizers").
#   bash tools/run_sanitizers.sh [asan|tsan|both] [out_dir]       (on a box with a gfx950 device: gpurun -- 'bash tools/run_sanitizers.sh')
# Builds libmbtenv.asan.so / libmbtenv.tsan.so if they are missing or stale (host objects only: -fno-gpu-sanitize), preloads the
# sanitizer's runtime into the uninstrumented python, runs the tests, and collects every report into <out_dir>/<variant>_reports.txt.
set -u
WHICH=${1:-both}
OUT=${2:-gpurun_out/sanitizers}
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
TESTS="tests/test_gpu_lifecycle.py tests/test_gpu_host_buffers.py tests/test_gpu_host_callbacks.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_output_pool.py tests/test_gpu_graph_step.py tests/test_gpu_zero_copy.py tests/test_cabi.py"
run_variant() {
  local variant=$1 sanitize=$2
  MBT_SANITIZE=$sanitize python -c "from mbt_gym_amd.build import build_native; print(build_native())" > "$OUT/${variant}_build.log" 2>&1 || { echo "$variant: build failed"; tail -5 "$OUT/${variant}_build.log"; return 1; }
  local runtime
  runtime=$(python -c "from mbt_gym_amd.build import sanitizer_runtime; print(sanitizer_runtime('$variant'))")
  rm -f "$OUT/${variant}_report".*
  # ASan: leaks are not this run's subject (python and the HIP runtime hold memory until exit); the HIP runtime maps its apertures
  #       where ASan's shadow gap would sit (protect_shadow_gap=0).  UBSan prints and carries on, so that every finding is collected.
  # TSan: the python interpreter and the HIP runtime are not instrumented - their own synchronisation is invisible to it, so only
  #       reports whose stack passes through libmbtenv count (ignore_noninstrumented_modules=1); history_size for long-lived threads.
  local -x ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:log_path=$ROOT/$OUT/${variant}_report:detect_stack_use_after_return=0"
  local -x UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$ROOT/$OUT/${variant}_report"
  local -x TSAN_OPTIONS="halt_on_error=0:log_path=$ROOT/$OUT/${variant}_report:ignore_noninstrumented_modules=1:history_size=4:report_signal_unsafe=0"
  local -x MBT_LIBRARY_VARIANT=$variant
  # (a sanitizer's dlopen interceptor makes ITS runtime the caller, so torch's lazily loaded libraries are no longer found through
  # torch's own RUNPATH - "libcaffe2_nvrtc.so: cannot open shared object file": name the directory)
  local torch_lib
  torch_lib=$(python -c "import importlib.util, os; s = importlib.util.find_spec('torch'); print(os.path.join(list(s.submodule_search_locations)[0], 'lib') if s else '')")
  local -x LD_LIBRARY_PATH="$torch_lib${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
  # (ROCm's ASan runtime quarantines DEVICE allocations too - it intercepts hsa_amd_memory_pool_free - so "the memory of 60 destroyed
  # environments is free again" cannot hold under it: 116 MiB were still quarantined when the test looked)
  local skip=""
  # (... and the tests that run bench.py at 2^20 lanes and more in a subprocess die inside the HIP runtime: "AddressSanitizer: out of
  # memory: allocator is trying to allocate 0x400000 bytes", frame libamdhip64.so - the runtime's own allocations through ASan's
  # allocator; they measure bench.py's output format, not the host library)
  [ "$variant" = asan ] && skip="--deselect tests/test_gpu_lifecycle.py::test_create_use_destroy_returns_device_memory --deselect tests/test_gpu_lifecycle.py::test_user_plugin_environments_come_and_go \
    --deselect tests/test_gpu_round4.py::test_the_bench_line_measures_every_baseline_config_and_the_contract_tier --deselect tests/test_gpu_round4.py::test_cfg4_sharded_block_of_the_multi_rank_line \
    --deselect tests/test_gpu_round5.py::test_the_bench_line_carries_the_fused_rollout --deselect tests/test_gpu_round5.py::test_ranks_rehearsed_on_one_device"
  # is the run what it claims to be?  The library the binding loads, how many sanitizer call sites it holds, and - for ASan - a canary:
  # mbt_exact_split handed a 2-byte heap block for its int32 result must be reported (and is kept out of the findings below)
  echo "== $variant: $(LD_PRELOAD=$runtime python -c "from mbt_gym_amd import _native; _native.load_library(); print('binding loads', _native.LIB_PATH)" 2>/dev/null), $(nm -D --undefined-only "mbt_gym_amd/libmbtenv.$variant.so" | grep -c "__${variant}_\|__ubsan_") sanitizer entry points referenced"
  if [ "$variant" = asan ]; then  # (the report lands with the others; tools/sanitizer_summary.py recognises it and counts it as the canary)
    LD_PRELOAD=$runtime python -c "
import ctypes as C
from mbt_gym_amd import _native
lib = _native.load_library()
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
hi, small = C.c_float(), libc.malloc(2)  # (libc's malloc: python's own small-object allocator is invisible to ASan)
lib.mbt_exact_split(1.5, C.byref(hi), C.cast(small, C.POINTER(C.c_int32)))" > /dev/null 2>&1
  fi
  echo "== $variant ($sanitize): LD_PRELOAD=$runtime python -m pytest $TESTS $skip"
  LD_PRELOAD=$runtime timeout 1500 python -m pytest -m "gpu or not gpu" $TESTS $skip -q -p no:cacheprovider > "$OUT/${variant}_pytest.log" 2>&1
  echo "   pytest rc=$?: $(grep -E " passed| failed| error" "$OUT/${variant}_pytest.log" | tail -1)"
  # the eight-process soak of round 5 (resident small-batch stepping: mailbox, spin flags, completion flags), shortened - under a watchdog
  # (tools/soak_watchdog.sh) that takes the stacks of a run that hangs and says what the stuck thread was doing.  Under ASan about one run
  # in ten hangs INSIDE ROCm's ASan runtime (it quarantines device allocations, and recycling one from within ROCr's own allocator waits
  # for a lock the thread already holds: profiles/r06_sanitizers.txt); such a run says nothing about libmbtenv and is repeated, anything
  # else is reported as it is.  quarantine_size_mb=2048 (default 256) keeps the quarantine from filling up - and so from recycling - within a
  # nine-second run: 0 hangs in 25 runs with it, 5 in 68 without (tools/dbg/r06_asan_soak_repro.sh).
  local attempt rc_soak=98
  for attempt in 1 2 3; do
    ASAN_OPTIONS="$ASAN_OPTIONS:quarantine_size_mb=2048" LD_PRELOAD_FOR_PYTHON=$runtime MBT_RESIDENT_STEP=1 MBT_FUZZ_SCALE=2 MBT_FUZZ_SEED=$((3100000 + attempt - 1)) HANG_AFTER=90 bash tools/soak_watchdog.sh "$OUT" "${variant}_soak"
    rc_soak=$?
    [ $rc_soak -ne 98 ] && break
    echo "   resident soak, attempt $attempt: $(cat "$OUT/${variant}_soak.verdict" 2>/dev/null | cut -c1-400)"
    grep -q "not in libmbtenv" "$OUT/${variant}_soak.verdict" 2>/dev/null || break
    cp "$OUT/${variant}_soak.verdict" "$OUT/${variant}_soak.attempt$attempt.verdict"
  done
  echo "   resident soak (8 processes) rc=$rc_soak: $(tail -1 "$OUT/${variant}_soak.log" | cut -c1-200)"
  cat "$OUT/${variant}_report".* > "$OUT/${variant}_reports.txt" 2>/dev/null
  python tools/sanitizer_summary.py "$OUT/${variant}_reports.txt" | tee "$OUT/${variant}_summary.txt"
}
case "$WHICH" in
  asan) run_variant asan address,undefined ;;
  tsan) run_variant tsan thread ;;
  *) run_variant asan address,undefined; run_variant tsan thread ;;
esac
