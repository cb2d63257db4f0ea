set -o pipefail
mkdir -p gpurun_out/r03g
timeout -k 10 600 python -m pytest tests/test_gpu_06_group.py tests/test_gpu_parity.py tests/test_gpu_00_configs.py -m gpu -x -q -k "group or Weighted or wmv or 4k" > gpurun_out/r03g/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03g/pytest.log
[ $rc -eq 0 ] || exit 1
python - <<'P' > gpurun_out/r03g/configs.json
import json, sys
sys.path.insert(0, '.')
from tools import bench_configs
print(json.dumps(bench_configs.configs_block(cpu=False)))
P
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03g/configs.json').read().strip().splitlines()[-1])
c=d['configs2_wmv_abl_4k']
for k in ('wmv','abl','both_kernels','fused_group'): print(k, json.dumps(c[k])[:330])
P
python tools/bench_configs.py --only byte 2>&1 | grep -i "Variance"
