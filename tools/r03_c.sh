set -o pipefail
mkdir -p gpurun_out/r03c
SECONDS=0
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err; echo "bench rc=$? in $SECONDS s"
tail -5 gpurun_out/r03c/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03c/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','streams_1080p30')}))
print(json.dumps(d['clip'], indent=1))
print(json.dumps(d['configs'], indent=1))
print(d['cpu_baseline'])
P
