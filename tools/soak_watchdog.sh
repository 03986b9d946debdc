#!/bin/bash
# One run of the eight-process soak (tests/test_gpu_random_configs.py -n 8) under a watchdog that says WHERE a run hangs.
#   bash tools/soak_watchdog.sh <out_dir> <tag> [pytest arguments ...]     (environment: LD_PRELOAD_FOR_PYTHON, HANG_AFTER [60], the MBT_* knobs)
# exit code: pytest's own; 98 = the run hung and was ended, its stacks are in <out_dir>:
#   hung_py_stack.<pid>.<tag>.txt  Python stacks of every worker (tools/hang_dump_plugin.py: faulthandler, 25 s into a test), the test's id in front
#   native_stack.<pid>.<tag>.txt   native stacks of every worker (rocgdb attached here, HANG_AFTER seconds into a run that normally takes 9)
#   <tag>.verdict                  one line: what the stuck thread was doing (classify below)
# The run is ended through its own process ids.  The sanitizer's runtime is preloaded into python ONLY (LD_PRELOAD_FOR_PYTHON), not into rocgdb.
set -u
OUT=$1; TAG=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export MBT_HANG_DUMP_DIR=$(cd "$OUT" && pwd) PYTHONPATH=$ROOT/tools${PYTHONPATH:+:$PYTHONPATH}
rm -f "$OUT"/py_stack.*.txt
t0=$(date +%s)
LD_PRELOAD=${LD_PRELOAD_FOR_PYTHON:-} python -m pytest tests/test_gpu_random_configs.py -q -n 8 -p no:cacheprovider -p hang_dump_plugin "$@" > "$OUT/$TAG.log" 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  sleep 1
  [ $(( $(date +%s) - t0 )) -lt ${HANG_AFTER:-60} ] && continue
  workers=$(pgrep -P $pid)
  for w in $workers; do
    timeout 90 rocgdb -batch -p $w -ex "set pagination off" -ex "thread apply all bt 40" > "$OUT/native_stack.$w.$TAG.txt" 2>&1
  done
  for f in "$OUT"/py_stack.*.txt; do b=$(basename "$f" .txt); mv "$f" "$OUT/hung_$b.$TAG.txt"; done  # (py_stack.<pid>.txt -> hung_py_stack.<pid>.<tag>.txt: kept)
  for w in $workers; do kill -9 $w 2>/dev/null; done
  kill -9 $pid 2>/dev/null
  wait $pid 2>/dev/null
  python - "$OUT" "$TAG" <<'PY' | tee "$OUT/$TAG.verdict"
import glob, re, sys
out, tag = sys.argv[1:3]
lines = []
for path in sorted(glob.glob(f"{out}/hung_py_stack.*.{tag}.txt")):
    text = open(path).read()
    if "most recent call first" not in text:
        continue  # (this worker's tests all finished within the bound)
    pid = path.split("hung_py_stack.")[1].split(".")[0]
    test = re.findall(r"^== (.*)$", text, re.M)[-1]
    frames = [f for f in re.findall(r'File "([^"]+)", line (\d+) in (\w+)', text) if "mbt_gym_amd" in f[0] or "/tests/" in f[0]]
    where = f"{frames[0][0].split('/')[-1]}:{frames[0][1]} {frames[0][2]}" if frames else "?"
    native = open(f"{out}/native_stack.{pid}.{tag}.txt").read() if glob.glob(f"{out}/native_stack.{pid}.{tag}.txt") else ""
    main = native.split("\nThread 1 ")[-1]
    inner = [m for m in re.findall(r"^#\d+\s+(?:0x[0-9a-f]+ in )?(.+?) \(", main, re.M)]
    # ROCm's ASan runtime against itself: ROCr's allocator (MemoryRegion::Allocate, holding the region's lock) frees a host object, ASan's operator
    # delete recycles its quarantine, the quarantine holds DEVICE allocations (the runtime intercepts hsa_amd_memory_pool_allocate / free), freeing
    # one re-enters ROCr (MemoryRegion::Free) and waits for the lock this very thread holds
    own_lock = any("MemoryRegion::Free" in f for f in inner) and any("MemoryRegion::Allocate" in f for f in inner) and any("Recycle" in f for f in inner)
    in_library = any("libmbtenv" in l for l in main.splitlines()[:12])
    kind = ("ROCm ASan runtime self-deadlock: quarantine recycling of a device allocation (hsa_memory_free) inside ROCr's own allocation, on ROCr's region lock - not in libmbtenv"
            if own_lock else ("stuck with libmbtenv among the innermost frames" if in_library else "stuck outside libmbtenv"))
    lines.append(f"hung: pid {pid}, {test}, python at {where}; innermost native frames: {' <- '.join(inner[:4])}; {kind}")
print("\n".join(lines) if lines else "hung: no worker had a test running for 25 s (the session itself did not end)")
PY
  exit 98
done
wait $pid
rc=$?
rm -f "$OUT"/py_stack.*.txt  # (nothing hung: the lists of test ids are of no further use)
exit $rc
