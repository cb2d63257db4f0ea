set -o pipefail
for ov in 1 0; do echo "BGS_SS_OVERLAP=$ov"; for k in subsense8 subsense; do BGS_SS_OVERLAP=$ov timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids; done; done
BGS_SS_OVERLAP=0 bash tools/trace_ss_step.sh
