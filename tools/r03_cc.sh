#!/bin/bash
# round-3 scratch: per-kernel times of the SuBSENSE -> components pipeline
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only pipeline > $O/pipe.log 2>&1
grep -v "^W2026\|^E2026" $O/pipe.log | tail -3
python3 - "$O/pipe/t_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %6s avg %10.1f us total %10.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
