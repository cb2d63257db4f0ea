#!/bin/bash
# PMC counters of one kernel: tools/pmc_kernel.sh <tag> <kernel substring> "<counters>" -- <python script and args>
# (counters in their own run with --kernel-trace only; summarised on the box: mean per dispatch of each counter)
set -e
TAG=$1; KERN=$2; CTRS=$3; shift 3; [ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("Request exceeds the capabilities of the hardware": rocprofv3 aborts and then
# hangs in its finaliser) - pass them in separate invocations; the timeout keeps a bad counter set from eating the whole call
timeout -k 5 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -o pmc -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-28s mean/dispatch %16.1f  (n=%d)" % (k, s / n, n))
PY
rm -rf $OUT/raw
