python tools/diag_submit.py 2>&1 | grep cams
timeout -k 10 300 python tools/bench_configs.py --only clip 2>&1 | grep "T=1\|T=4 (sat" | cut -c1-160
