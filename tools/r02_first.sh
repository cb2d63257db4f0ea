#!/bin/bash
# round 2, first GPU call: poisoned GPU suite once, then the placement-probe de-confounding runs
set -o pipefail
mkdir -p gpurun_out/r02a
BGS_DEBUG_POISON=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_poison.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest_poison.log
tail -5 gpurun_out/r02a/pytest_poison.log
for i in 1 2 3; do
  BGS_DEBUG_PROBE=2 timeout -k 10 300 python bench.py --steps 200 --warmup 50 --main-only > gpurun_out/r02a/probe_on_$i.json 2> gpurun_out/r02a/probe_on_$i.err || exit 1
  BGS_PLACEMENT_PROBE=0 timeout -k 10 300 python bench.py --steps 200 --warmup 50 --main-only > gpurun_out/r02a/probe_off_$i.json 2> gpurun_out/r02a/probe_off_$i.err || exit 1
done
grep -h "placement" gpurun_out/r02a/probe_on_*.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a/probe_o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])
P
