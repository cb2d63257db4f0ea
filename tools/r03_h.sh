# placement study on the round-3 production kernel (filter path): 6 fresh processes with and without the probe
set -o pipefail
mkdir -p gpurun_out/r03h
for i in 1 2 3 4 5 6; do
  for v in "BGS_PLACEMENT_PROBE=0" "BGS_PLACEMENT_PROBE=20"; do
    env $v timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100 > gpurun_out/r03h/pl_${v}_$i.json 2>/dev/null || true
    python - "gpurun_out/r03h/pl_${v}_$i.json" "$v" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['placement_probe']['candidates_ms_per_dense_launch'])
except Exception as e: print(sys.argv[2], 'failed', e)
P
  done
done
