# Validation on a GPU box (gpurun -- bash tools/validate_gpu.sh): GPU suite (plain, then once more with poisoned allocations), smoke, the driver's bench command
set -o pipefail
mkdir -p gpurun_out/validate
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/validate/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/validate/pytest.log
[ $rc -eq 0 ] || exit 1
BGS_DEBUG_POISON=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/validate/pytest_poison.log 2>&1; rc=$?; echo "pytest poison rc=$rc"; tail -3 gpurun_out/validate/pytest_poison.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/validate/bench_k20.json 2> gpurun_out/validate/bench_k20.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/validate/bench_k20.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained'], d['cpu_baseline']['value'], d['s_surv']['default']['mpixels_per_s'])
P
