#!/bin/bash
# The recorded rollout against its write-only floor, fixed placement (bench.py's rollout block: kernel and floor alternately in one process, on
# the same buffers), FIVE processes - a new allocation each: medians of the five
set -u
OUT=gpurun_out/r06f; mkdir -p "$OUT"
for p in 1 2 3 4 5; do
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-hbm-resident --no-configs --no-device-loop 2>/dev/null | python -c "
import json, sys
line = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = line['rollout']
out = {k: {a: v[a] for a in ('us_per_env_step_of_all_lanes', 'write_only_floor_us', 'floor_over_kernel', 'write_GBps', 'us_per_step_min_max') if a in v} for k, v in r.items() if k.startswith('recorded')}
out['returns_only_as'] = r['returns_only_avellaneda_stoikov_policy']['env_steps_per_s']
print(json.dumps(out))" | tee -a "$OUT/rollout_floor.txt"
done
