#!/bin/bash
# Timeline of one 8-stream SuBSENSE step: rocprofv3 --kernel-trace, then the dispatches of the last complete step with their start offsets.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_ss
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only ${LEG:-subsense8} > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "bgs::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "ss_phase_a_kernel" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bgs::", "")
    print("%9.1f us  +%7.1f us  %s" % (s / 1e3, (e - s) / 1e3, name[:60]))
print("step (phase A start to next phase A start): %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -rf $OUT/raw
