set -o pipefail
mkdir -p gpurun_out/r02r
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lbsp or subsense_golden or lobster_golden or subsense_ragged or subsense_gray" > gpurun_out/r02r/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02r/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only lbsp 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02r/bench.txt
