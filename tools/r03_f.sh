set -o pipefail
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_01_host_cpp.py -m gpu -x -q -k "registered or demo or host" > gpurun_out/r03f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r03f/pytest.log
[ $rc -eq 0 ] || exit 1
python - <<'P'
import sys, json
sys.path.insert(0,'.')
import torch, bench
pool = bench.make_pool("sat", 1, 10, torch.device("cuda",0), 1234)
print(json.dumps(bench.host_leg(0, pool), indent=1))
P
