set -o pipefail
t0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_live.json 2> gpurun_out/bench_live.err; echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_live.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'], r['traffic_read_write'])
print(r['algorithmic_bytes_per_launch'], r['traffic']/r['algorithmic_bytes_per_launch'] if r['traffic'] else None)
P
