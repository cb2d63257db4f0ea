set -o pipefail
mkdir -p gpurun_out/r03d
timeout -k 10 600 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_03_clip.py tests/test_gpu_parity.py -m gpu -x -q -k "mog2 or MOG2 or Mixture or bench_geometry or 1080p_mog2" > gpurun_out/r03d/pytest_mog2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03d/pytest_mog2.log
[ $rc -eq 0 ] || exit 1
for v in "BGS_DEBUG_STAT=1" "BGS_MOG2_SPARSE=1" "BGS_MOG2_SPARSE=2" "BGS_MOG2_SPARSE=4"; do
  n=$(echo $v | tr '=' '_')
  env $v timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 > gpurun_out/r03d/ab_$n.json 2> gpurun_out/r03d/ab_$n.err || true; grep "mog2 auto" gpurun_out/r03d/ab_$n.err | sort | uniq -c | sort -rn | head -5
  python - "$n" <<'P'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r03d/ab_%s.json'%v).read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['placement_probe']['candidates_ms_per_dense_launch'])
except Exception as e: print(v, 'failed', e)
P
done
SECONDS=0
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err; echo "bench rc=$? in $SECONDS s"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03d/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','streams_1080p30')}))
print(json.dumps(d['roofline'], indent=1))
print(json.dumps(d['s_surv'], indent=1))
print(json.dumps(d['clip'], indent=1))
P
