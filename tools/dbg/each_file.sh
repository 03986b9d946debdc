#!/bin/bash
# Every GPU test file in its own interpreter: import-order dependence and exit-time aborts show up here, not in the full run.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out/dbg
for f in tests/test_*.py; do
  timeout 600 python -m pytest "$f" -q -m gpu -x > /tmp/one.log 2>&1; rc=$?
  echo "$rc $(basename $f) $(grep -E 'passed|failed|no tests ran|deselected' /tmp/one.log | tail -1)"
  if [ $rc -ne 0 ] && [ $rc -ne 5 ]; then tail -30 /tmp/one.log; fi
done 2>&1 | tee gpurun_out/dbg/each_file.txt
