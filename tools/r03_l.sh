set -o pipefail
mkdir -p gpurun_out/r03l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_00_configs.py tests/test_gpu_04_bench.py -m gpu -x -q -k "submit_wait or registered or mog2 or MOG2 or bench" > gpurun_out/r03l/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r03l/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only clip 2>&1 | grep "T=1\|T=4" | cut -c1-200
python - <<'P'
import sys, json
sys.path.insert(0,'.')
import torch, bench
pool = bench.make_pool("sat", 1, 10, torch.device("cuda",0), 1234)
print(json.dumps(bench.host_leg(0, pool), indent=1))
P
timeout -k 10 400 python tools/fuzz_parity.py 150 31000 > gpurun_out/r03l/fuzz_small.log 2>&1; echo "fuzz small rc=$?"; tail -2 gpurun_out/r03l/fuzz_small.log | cut -c1-300
timeout -k 10 400 python tools/fuzz_parity.py 150 32000 big > gpurun_out/r03l/fuzz_big.log 2>&1; echo "fuzz big rc=$?"; tail -2 gpurun_out/r03l/fuzz_big.log | cut -c1-300
