#!/bin/bash
# The eight-process resident soak under ASan, again and again, until one run hangs; then WHERE: the Python stacks of the hanging worker
# (tools/dbg/hang_dump_plugin.py: faulthandler after 25 s) and the native stacks of every worker (rocgdb attached by this script, 40 s into a
# run that normally takes 9 s), then the run is ended through its own process ids.
#   REPS=40 bash tools/dbg/r06_asan_soak_repro.sh          (VARIANT=asan|tsan|"" for the plain library)
set -u
OUT=gpurun_out/r06_asan_repro; mkdir -p "$OUT"; ROOT=$(pwd)
VARIANT=${VARIANT-asan}
torch_lib=$(python -c "import importlib.util, os; s = importlib.util.find_spec('torch'); print(os.path.join(list(s.submodule_search_locations)[0], 'lib') if s else '')")
export LD_LIBRARY_PATH="$torch_lib${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:log_path=$ROOT/$OUT/asan_report:detect_stack_use_after_return=0"
export TSAN_OPTIONS="report_bugs=1:halt_on_error=0:log_path=$ROOT/$OUT/tsan_report"
preload=""
if [ -n "$VARIANT" ]; then
  export MBT_LIBRARY_VARIANT=$VARIANT
  preload=$(python -c "from mbt_gym_amd.build import sanitizer_runtime; print(sanitizer_runtime('$VARIANT'))")
fi
export MBT_HANG_DUMP_DIR=$ROOT/$OUT PYTHONPATH=$ROOT/tools/dbg${PYTHONPATH:+:$PYTHONPATH}
for i in $(seq 1 ${REPS:-8}); do
  t0=$(date +%s)
  rm -f "$OUT"/py_stack.*.txt
  LD_PRELOAD=$preload MBT_RESIDENT_STEP=${RESIDENT-1} MBT_FUZZ_SCALE=2 MBT_FUZZ_SEED=$((3100000 + i)) python -m pytest tests/test_gpu_random_configs.py -q -n 8 \
      -p no:cacheprovider -p hang_dump_plugin > "$OUT/soak_$i.log" 2>&1 &
  pid=$!
  hung=0
  while kill -0 $pid 2>/dev/null; do
    sleep 1
    if [ $(( $(date +%s) - t0 )) -ge ${HANG_AFTER:-40} ]; then
      hung=1
      workers=$(pgrep -P $pid)
      echo "run $i: still running after ${HANG_AFTER:-40} s; workers: $(echo $workers)"
      for w in $workers; do
        timeout 90 rocgdb -batch -p $w -ex "set pagination off" -ex "thread apply all bt 30" > "$OUT/native_stack_run${i}.$w.txt" 2>&1
      done
      cp "$OUT"/py_stack.*.txt "$OUT"/ 2>/dev/null
      for f in "$OUT"/py_stack.*.txt; do mv "$f" "${f%.txt}.run$i.keep"; done
      for w in $workers; do kill -9 $w 2>/dev/null; done
      kill -9 $pid 2>/dev/null
      break
    fi
  done
  wait $pid 2>/dev/null; rc=$?
  echo "run $i rc=$rc hung=$hung $(( $(date +%s) - t0 )) s: $(tail -1 "$OUT/soak_$i.log" | cut -c1-150)"
  [ $hung -ne 0 ] && break
  [ $rc -ne 0 ] && break
done
ls "$OUT" | grep -c asan_report
