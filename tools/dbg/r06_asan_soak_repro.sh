#!/bin/bash
# The eight-process resident soak under a sanitizer, again and again, until one run hangs; tools/soak_watchdog.sh then says where.
#   REPS=40 bash tools/dbg/r06_asan_soak_repro.sh      VARIANT=asan|tsan|"" (the plain library)   RESIDENT=1|0   ASAN_EXTRA=quarantine_size_mb=1024
# Found with it (profiles/r06_sanitizers.txt): the hang is ROCm's ASan runtime deadlocking on ROCr's region lock, in mbt_env_create's hipMalloc.
set -u
OUT=gpurun_out/r06_asan_repro; mkdir -p "$OUT"; ROOT=$(pwd)
VARIANT=${VARIANT-asan}
torch_lib=$(python -c "import importlib.util, os; s = importlib.util.find_spec('torch'); print(os.path.join(list(s.submodule_search_locations)[0], 'lib') if s else '')")
export LD_LIBRARY_PATH="$torch_lib${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:log_path=$ROOT/$OUT/asan_report:detect_stack_use_after_return=0${ASAN_EXTRA:+:$ASAN_EXTRA}"
export TSAN_OPTIONS="halt_on_error=0:log_path=$ROOT/$OUT/tsan_report:ignore_noninstrumented_modules=1:history_size=4:report_signal_unsafe=0"
preload=""
if [ -n "$VARIANT" ]; then
  export MBT_LIBRARY_VARIANT=$VARIANT
  preload=$(python -c "from mbt_gym_amd.build import sanitizer_runtime; print(sanitizer_runtime('$VARIANT'))")
fi
hangs=0
for i in $(seq 1 ${REPS:-8}); do
  t0=$(date +%s)
  rm -f "$OUT"/asan_report.*  # (one per process at exit, see profiles/r06_sanitizers.txt: not this script's subject)
  LD_PRELOAD_FOR_PYTHON=$preload MBT_RESIDENT_STEP=${RESIDENT-1} MBT_FUZZ_SCALE=2 MBT_FUZZ_SEED=$((3100000 + i)) HANG_AFTER=${HANG_AFTER:-40} bash tools/soak_watchdog.sh "$OUT" "run$i"
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 )) s: $(tail -1 "$OUT/run$i.log" | cut -c1-150)"
  if [ $rc -eq 98 ]; then hangs=$((hangs + 1)); [ -n "${STOP_AT_FIRST_HANG-1}" ] && break; continue; fi
  [ $rc -ne 0 ] && break
  rm -f "$OUT/run$i.log"
done
echo "$hangs hangs in $i runs (VARIANT=$VARIANT RESIDENT=${RESIDENT-1} ASAN_EXTRA=${ASAN_EXTRA-})"
