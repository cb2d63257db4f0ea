# round 3, first GPU contact of the slot-layout MOG2 kernel: parity subset, bench, A/B knobs
set -o pipefail
mkdir -p gpurun_out/r03a
timeout -k 10 600 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_03_clip.py tests/test_gpu_parity.py -m gpu -x -q -k "mog2 or MOG2 or Mixture or bench_geometry or 1080p_mog2" > gpurun_out/r03a/pytest_mog2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03a/pytest_mog2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
for v in "BGS_MOG2_COMPLETE=0" "BGS_LIB_PATH=$PWD/tracking_amd/lib/libbgs_hip_t64.so" "BGS_PLACEMENT_PROBE=0" "BGS_MOG2_SPARSE=4" "BGS_XCD_SWIZZLE=0"; do
  n=$(echo $v | cut -d= -f1)
  env $v timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 > gpurun_out/r03a/ab_$n.json 2>/dev/null || true
  python - "$n" <<'P'
import json,sys,glob
v=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r03a/ab_%s.json'%v).read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['placement_probe'])
except Exception as e: print(v, 'failed', e)
P
done
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03a/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','streams_1080p30')}), json.dumps(d['roofline']), d['s_surv'], d['clip'])
P
