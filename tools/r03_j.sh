# deterministic placement: three fresh processes of the default build, then the memory-return test and the big-model tests
set -o pipefail
mkdir -p gpurun_out/r03j
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100 > gpurun_out/r03j/d_$i.json 2>gpurun_out/r03j/d_$i.err || true
  python - gpurun_out/r03j/d_$i.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['model_placement']['chunk_MiB'], d['model_placement']['chunks'])
except Exception as e: print('failed', e, open(sys.argv[1].replace('.json','.err')).read()[-400:])
P
done
timeout -k 10 900 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "large_batches or bench_geometry or memory or returns or 1080p" > gpurun_out/r03j/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03j/pytest.log
python tools/bench_configs.py --only dp 2>&1 | grep -i "Zivkovic\|Grimson" | cut -c1-200
