#!/bin/bash
# PMC counters of EVERY bgs:: kernel of a run: tools/pmc_all.sh <tag> "<counters>" -- <python script and args>
# (counters in their own run with --kernel-trace only; mean per dispatch of each counter, per kernel)
set -e
TAG=$1; CTRS=$2; shift 2; [ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -k 5 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -o pmc -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bgs::" in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-44s %-22s mean/dispatch %16.1f  (n=%d)" % (k, c, s / n, n))
PY
rm -rf $OUT/raw
