timeout -k 10 300 python tools/bench_configs.py --only clipfd 2>&1 | grep -v amdgpu.ids | sed 's/.*clip T=\([0-9]*\) (\(.*\)) 3840.*wall \(.*\)/\2 T=\1 wall \3/'
